// TEST INFRASTRUCTURE ONLY (see wave_emu.h).  The scheduler of the lock-step wave model.  tests/emu/libspmx_emu.so =
// this + emu_launch.cc (csrc/launch.h on the model) + the product's own host code (csrc/api.cc, model.cc, dat.cc,
// tables.cc) compiled against fakehip/: it exports the C ABI of include/spmx.h, so the tests drive the product's
// launch sequence and device bodies on the CPU.
#include "wave_emu.h"

#include <sys/mman.h>

#include <mutex>
#include <string>
#include <vector>


namespace spmx {
namespace emu {

thread_local Wave g_wave;

asm(R"(
.text
.globl spmx_emu_switch
.type spmx_emu_switch,@function
spmx_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
)");

namespace {
constexpr size_t kStack = 256 * 1024;

void LaneEntry() {
  Lane &l = g_wave.lanes[g_wave.cur];
  g_wave.body();
  l.done = true;
  l.op = kNone;
  spmx_emu_switch(&l.sp, g_wave.sched_sp);
  abort();  // never resumed
}

void Fail(const char *msg) {
  fprintf(stderr, "wave_emu: %s (block %d)\n", msg, g_wave.block);
  abort();
}
}  // namespace

void RunWave(int block, int grid, unsigned char *smem, const std::function<void()> &body) {
  Wave &w = g_wave;
  w.block = block; w.grid = grid; w.smem = smem; w.body = body;
  for (int i = 0; i < 64; ++i) {
    Lane &l = w.lanes[i];
    if (!l.stack) {
      l.stack = static_cast<unsigned char *>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
      if (l.stack == MAP_FAILED) Fail("mmap failed");
    }
    l.done = false; l.op = kNone;
    uintptr_t top = reinterpret_cast<uintptr_t>(l.stack + kStack) & ~uintptr_t(15);
    void **sp = reinterpret_cast<void **>(top - 16);
    *sp = reinterpret_cast<void *>(&LaneEntry);   // return address popped by `ret`
    sp -= 6;                                      // rbp rbx r12 r13 r14 r15
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    l.sp = sp;
  }
  for (;;) {
    int n_done = 0;
    for (int i = 0; i < 64; ++i) {
      Lane &l = w.lanes[i];
      if (l.done) { ++n_done; continue; }
      w.cur = i;
      spmx_emu_switch(&w.sched_sp, l.sp);
      if (l.done) ++n_done;
    }
    if (n_done == 64) break;
    if (n_done != 0) Fail("divergent collective: some lanes exited while others wait");
    const Op op = w.lanes[0].op;
    for (int i = 1; i < 64; ++i) if (w.lanes[i].op != op) {
      fprintf(stderr, "wave_emu: ops per lane:");
      for (int k = 0; k < 64; ++k) fprintf(stderr, " %d", static_cast<int>(w.lanes[k].op));
      fprintf(stderr, "\n");
      Fail("divergent collective: lanes wait at different operations");
    }
    ++w.n_collectives;
    switch (op) {
      case kBallot: {
        uint64_t m = 0;
        for (int i = 0; i < 64; ++i) if (w.lanes[i].a) m |= 1ull << i;
        for (int i = 0; i < 64; ++i) w.lanes[i].out = m;
        break;
      }
      case kShfl:
        for (int i = 0; i < 64; ++i) w.lanes[i].out = w.lanes[w.lanes[i].b & 63].a;
        break;
      case kShflUp:
        for (int i = 0; i < 64; ++i) {
          const int d = static_cast<int>(w.lanes[i].b);
          w.lanes[i].out = i >= d ? w.lanes[i - d].a : w.lanes[i].a;
        }
        break;
      case kSync: break;
      case kBlockSync:
        if (!w.block_barrier) Fail("wv::block_sync() in a workgroup whose wavefronts run one after another");
        w.block_barrier();
        break;
      case kLaneUp1:
        for (int i = 63; i >= 0; --i) w.lanes[i].out = i > 0 ? w.lanes[i - 1].a : w.lanes[i].b;
        break;
      case kLaneDown1:
        for (int i = 0; i < 64; ++i) w.lanes[i].out = i < 63 ? w.lanes[i + 1].a : w.lanes[i].b;
        break;
      case kScanAdd: {
        uint32_t acc = 0;
        for (int i = 0; i < 64; ++i) { acc += static_cast<uint32_t>(w.lanes[i].a); w.lanes[i].out = acc; }
        break;
      }
      case kScanMax: {
        uint32_t acc = 0;
        for (int i = 0; i < 64; ++i) { const uint32_t v = static_cast<uint32_t>(w.lanes[i].a); if (v > acc) acc = v; w.lanes[i].out = acc; }
        break;
      }
      default: Fail("unknown collective");
    }
  }
}

}  // namespace emu
}  // namespace spmx

extern "C" {
// collectives executed so far (a cheap progress / determinism probe for the tests)
uint64_t emu_collectives() { return spmx::emu::g_wave.n_collectives; }
}
