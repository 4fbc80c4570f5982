// TEST INFRASTRUCTURE ONLY (see wave_emu.h).  Builds tests/emu/libspmx_emu.so:
// the product's host code (model.cc, dat.cc, tables.cc) + the product's device
// bodies (kernels.h) compiled for the CPU against the lock-step wave model, and
// a C entry point that runs the same launch sequence as csrc/api.cc.
#include "wave_emu.h"

#include <sys/mman.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../sentencepiece_amd/csrc/kernels.h"
#include "../../sentencepiece_amd/csrc/model.h"
#include "../../sentencepiece_amd/csrc/tables.h"

namespace spmx {
namespace emu {

Wave g_wave;

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
      default: Fail("unknown collective");
    }
  }
}

}  // namespace emu
}  // namespace spmx

using namespace spmx;

struct EmuHandle {
  ModelData model;
  HostTables tables;
  std::string error;
};

namespace {
struct LengthClass { uint32_t rcap, ncap; };
// keep in sync with csrc/launch.h (the test compares the emulated pipeline with
// the oracle, not with these numbers; small classes are added to exercise the
// escalation path on short inputs)
// (the last class stands for the document-length classes: FAST kernel only, no staging)
const LengthClass kUniCls[] = {{24, 40}, {192, 448}, {576, 1280}, {1536, 3328}, {4096, 8704}, {8192, 20480}, {65536, 98304}};
const uint32_t kEmuMaxStagedRaw = 8192;
const LengthClass kBpeCls[] = {{24, 40}, {192, 448}, {576, 1280}, {1536, 3328}, {4096, 6400}, {65536, 98304}};
}  // namespace

extern "C" {
extern uint64_t g_fast_kept, g_fast_handed, g_wave_handed;

void *emu_load(const void *bytes, uint64_t n, char *err, uint64_t errcap) {
  auto *h = new EmuHandle;
  Status st = ParseModelProto(bytes, n, &h->model);
  if (st.ok()) st = InitializeModel(&h->model);
  if (st.ok()) st = CompileTables(h->model, &h->tables);
  if (st.ok()) st = CompileExtraOptions(h->model, "", &h->tables);
  if (!st.ok()) {
    if (err && errcap) snprintf(err, errcap, "%s", st.message.c_str());
    delete h;
    return nullptr;
  }
  BindHostPointers(&h->tables);
  return h;
}

void emu_free(void *h) { delete static_cast<EmuHandle *>(h); }

int emu_set_encode_extra_options(void *hv, const char *opts) {
  auto *h = static_cast<EmuHandle *>(hv);
  Status st = CompileExtraOptions(h->model, opts, &h->tables);
  return st.code;
}

int emu_set_vocabulary(void *hv, const char *pieces, uint64_t len) {
  auto *h = static_cast<EmuHandle *>(hv);
  std::vector<std::string> v;
  const char *p = pieces, *end = pieces + len;
  while (p < end) {
    const char *q = static_cast<const char *>(memchr(p, '\n', end - p));
    if (!q) q = end;
    v.emplace_back(p, q - p);
    p = q + 1;
  }
  Status st = SetVocabulary(&h->model, v);
  RefreshTypeFlags(h->model, &h->tables);
  return st.code;
}

int emu_reset_vocabulary(void *hv) {
  auto *h = static_cast<EmuHandle *>(hv);
  ResetVocabulary(&h->model);
  RefreshTypeFlags(h->model, &h->tables);
  return 0;
}

// Runs classify -> encode (every class) -> scan -> compact with `grid` waves
// per launch.  Returns total ids, or -(needed) - 2 if cap is too small; the
// device status word is returned in *status.
static uint32_t *g_span_begin = nullptr, *g_span_end = nullptr;   // set by emu_encode_spans_batch around its call
static uint32_t *g_nspan_begin = nullptr, *g_nspan_end = nullptr;

int64_t emu_encode_batch(void *hv, const uint8_t *text, const uint64_t *offs, uint64_t n, int32_t *ids, uint64_t cap,
                         uint64_t *id_offs, int grid, uint32_t *status_out) {
  auto *h = static_cast<EmuHandle *>(hv);
  const bool spans = g_span_begin != nullptr;
  const SpmxDev &dev = h->tables.scalars;
  const bool bpe = dev.model_type == 2;
  const LengthClass *cls = bpe ? kBpeCls : kUniCls;
  const int ncls = bpe ? static_cast<int>(sizeof(kBpeCls) / sizeof(kBpeCls[0])) : static_cast<int>(sizeof(kUniCls) / sizeof(kUniCls[0]));
  if (grid < 1) grid = 1;
  g_fast_kept = g_fast_handed = g_wave_handed = 0;
  std::vector<uint32_t> lists(static_cast<size_t>(ncls) * (n ? n : 1)), list_counts(kMaxClasses, 0), counts(n + 1, 0);
  std::vector<uint64_t> tmp_off(n + 1, 0);
  ClassifyArgs ca{};
  ca.offs = offs; ca.n = static_cast<uint32_t>(n); ca.n_classes = static_cast<uint32_t>(ncls);
  for (int c = 0; c < ncls; ++c) ca.rcap[c] = cls[c].rcap;
  ca.lists = lists.data(); ca.list_counts = list_counts.data();
  std::vector<uint32_t> key_totals(kSortKeys, 0), key_cursor(kSortKeys, 0), hist(3 * kSortKeys, 0);
  ca.key_totals = key_totals.data(); ca.key_cursor = key_cursor.data();
  ca.sub_buckets = getenv("SPMX_SUB_BUCKETS") ? static_cast<uint32_t>(atoi(getenv("SPMX_SUB_BUCKETS"))) : static_cast<uint32_t>(kSubBuckets);
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { classify_block<0>(ca, hist.data()); });
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { classify_block<1>(ca, hist.data()); });
  const uint64_t text_bytes = offs[n];
  std::vector<int32_t> arena(12 * text_bytes + (8 + dev.n_prefix + dev.n_suffix) * n + 64);
  std::vector<int32_t> arena_tb(spans ? arena.size() : 0, -7);
  unsigned long long arena_head = 0;
  uint32_t status = 0;
  unsigned long long stats[kStatsPerClass * kMaxClasses] = {0};
  for (int c = 0; c < ncls; ++c) {
    EncodeArgs a{};
    a.dev = dev; a.text = text; a.offs = offs;
    a.list = lists.data() + static_cast<size_t>(c) * n; a.list_count = &list_counts[c];
    const bool has_next = c + 1 < ncls && cls[c + 1].rcap <= kEmuMaxStagedRaw;
    a.next_list = has_next ? lists.data() + static_cast<size_t>(c + 1) * n : nullptr;
    a.next_count = has_next ? &list_counts[c + 1] : nullptr;
    a.arena = arena.data(); a.arena_head = &arena_head; a.arena_cap = arena.size();
    a.tmp_off = tmp_off.data(); a.counts = counts.data(); a.status = &status; a.stats = &stats[kStatsPerClass * c];
    a.rcap = cls[c].rcap; a.ncap = cls[c].ncap;
    a.no_lane_general = getenv("SPMX_NO_LANE_GENERAL") ? 1u : 0u;
    a.lane_general_max_raw = getenv("SPMX_LANE_GENERAL_MAX_RAW") ? static_cast<uint32_t>(atoi(getenv("SPMX_LANE_GENERAL_MAX_RAW"))) : kLaneGeneralMaxRaw;
    a.lane_general_min_lanes = getenv("SPMX_LANE_GENERAL_MIN_LANES") ? static_cast<uint32_t>(atoi(getenv("SPMX_LANE_GENERAL_MIN_LANES"))) : (a.rcap <= 576 ? 16u : 4u);
    a.arena_tb = spans ? arena_tb.data() : nullptr;
    a.ring = 16;
    while (a.ring < static_cast<uint32_t>(h->tables.max_piece_len) + 1) a.ring <<= 1;
    // streaming form, as in csrc/api.cc: the FAST kernel first (when the model allows it), then the GENERAL
    // kernel on what it left over.
    const bool bpe_stream = bpe && (dev.flags & kNfBpeWordwise) && !(dev.flags & kNfHasUnused);
    if (!bpe || (bpe_stream && !getenv("SPMX_NO_STREAM"))) {
      std::vector<uint32_t> hard(n ? n : 1), wavel(n ? n : 1);
      uint32_t hard_count = 0, wave_count = 0;
      const int waves = grid;   // one wave per block in the emulator
      const int model = bpe ? 2 : 1;
      a.wave_list = wavel.data(); a.wave_count = &wave_count;
      const bool staged = a.rcap <= kEmuMaxStagedRaw;
      if (!staged && !(StreamFastEligible(dev.flags) && !getenv("SPMX_NO_FAST"))) {
        if (list_counts[c]) status |= kStTooLong;      // csrc/api.cc fails the call here
        continue;
      }
      if (StreamFastEligible(dev.flags) && !getenv("SPMX_NO_FAST")) {
        a.hard_list = staged ? hard.data() : nullptr; a.hard_count = &hard_count;
        a.stream_tcap = staged ? a.rcap + 1 : a.ncap;
        std::vector<uint32_t> st(StreamTextDwords(a.stream_tcap, a.ring) * waves, 0xCDCDCDCDu), sb(StreamBpWords(a.stream_tcap) * waves, 0xCDCDCDCDu);
        a.stream_text = st.data(); a.stream_bp = sb.data();
        std::vector<unsigned char> fsmem(StreamLdsBytes(true, model, a.rcap, a.ncap, a.ring, 1) + 64, 0xCD);
        uint32_t fast_cursor = 0;
        a.tile_cursor = getenv("SPMX_STATIC_TILES") ? nullptr : &fast_cursor;
        std::vector<uint8_t> bpe_long((bpe && !staged) ? static_cast<size_t>(waves) * 64u * kBpeLongBytes : 0, 0xCD);
        a.bpe_long = (bpe && !staged) ? bpe_long.data() : nullptr;
        for (int b = 0; b < grid; ++b) {
          if (bpe) emu::RunWave(b, grid, fsmem.data(), [&] { encode_stream_block<true, 2>(a, fsmem.data()); });
          else if (a.ring == 16) emu::RunWave(b, grid, fsmem.data(), [&] { encode_stream_block<true, 1, 16>(a, fsmem.data()); });   // as LaunchEncodeStream picks
          else emu::RunWave(b, grid, fsmem.data(), [&] { encode_stream_block<true, 1>(a, fsmem.data()); });
        }
        g_fast_kept += list_counts[c] - hard_count;
        g_fast_handed += hard_count;
        a.list = hard.data(); a.list_count = &hard_count;
        a.hard_list = nullptr; a.hard_count = nullptr;
      }
      if (!staged) continue;
      a.stream_tcap = a.ncap;
      std::vector<uint32_t> st(StreamTextDwords(a.stream_tcap, a.ring) * waves, 0xCDCDCDCDu), sb(StreamBpWords(a.stream_tcap) * waves, 0xCDCDCDCDu);
      a.stream_text = st.data(); a.stream_bp = sb.data();
      std::vector<unsigned char> gsmem(StreamLdsBytes(false, model, a.rcap, a.ncap, a.ring, 1) + 64, 0xCD);
      uint32_t general_cursor = 0;
      a.tile_cursor = getenv("SPMX_STATIC_TILES") ? nullptr : &general_cursor;
      for (int b = 0; b < grid; ++b) {
        if (bpe) emu::RunWave(b, grid, gsmem.data(), [&] { encode_stream_block<false, 2>(a, gsmem.data()); });
        else emu::RunWave(b, grid, gsmem.data(), [&] { encode_stream_block<false, 1>(a, gsmem.data()); });
      }
      if (bpe) {   // sentences with a word too long for the lane form: sentence per wave
        g_wave_handed += wave_count;
        a.list = wavel.data(); a.list_count = &wave_count;
        std::vector<unsigned char> smem(EncodeLdsBytes(dev.model_type, a.rcap, a.ncap) + 64, 0xCD);
        for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, smem.data(), [&] { encode_block<2>(a, smem.data()); });
      }
      continue;
    }
    // BPE, sentence-per-wave form (models that are not word-wise, or SPMX_NO_STREAM)
    if (a.rcap > kEmuMaxStagedRaw) { if (list_counts[c]) status |= kStTooLong; continue; }   // csrc/api.cc fails the call
    std::vector<unsigned char> smem(EncodeLdsBytes(dev.model_type, a.rcap, a.ncap) + 64, 0xCD);
    for (int b = 0; b < grid; ++b) {
      emu::RunWave(b, grid, smem.data(), [&] { encode_block<2>(a, smem.data()); });
    }
  }
  std::vector<uint64_t> tile_sums((n + kScanTile - 1) / kScanTile + 2, 0);
  ScanArgs sa{counts.data(), static_cast<uint32_t>(n), tile_sums.data(), id_offs};
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_tiles_block(sa); });
  emu::RunWave(0, 1, nullptr, [&] { scan_sums_block(sa); });
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_final_block(sa); });
  CompactArgs pa{arena.data(), tmp_off.data(), counts.data(), id_offs, ids, cap, static_cast<uint32_t>(n)};
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { compact_block(pa); });
  if (status_out) *status_out = status;
  const uint64_t total = id_offs[n];
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  if (spans && status == 0 && total > 0) {   // as csrc/api.cc: token begins to CSR order, then the align kernel per class
    std::vector<int32_t> tokb(total, -9);
    CompactArgs pb{arena_tb.data(), tmp_off.data(), counts.data(), id_offs, tokb.data(), total, static_cast<uint32_t>(n)};
    for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { compact_block(pb); });
    std::vector<uint32_t> alists(static_cast<size_t>(ncls + 1) * (n ? n : 1)), acounts(kMaxClasses + 1, 0);   // align escalation lists
    for (int c = 0; c < ncls; ++c) {
      if (cls[c].rcap > kEmuMaxStagedRaw) { if (list_counts[c]) status |= kStTooLong; continue; }
      const bool has_next = c + 1 < ncls && cls[c + 1].rcap <= kEmuMaxStagedRaw;
      AlignArgs aa{};
      aa.dev = dev; aa.text = text; aa.offs = offs;
      aa.id_offs = id_offs; aa.tok_begin = tokb.data(); aa.begin = g_span_begin; aa.end = g_span_end;
      aa.nbegin = g_nspan_begin; aa.nend = g_nspan_end;
      aa.status = &status; aa.rcap = cls[c].rcap; aa.ncap = cls[c].ncap;
      aa.next_list = has_next ? alists.data() + static_cast<size_t>(c + 1) * n : nullptr;
      aa.next_count = has_next ? &acounts[c + 1] : nullptr;
      aa.list_cap = static_cast<uint32_t>(n);
      std::vector<unsigned char> smem(AlignLdsBytes(aa.rcap, aa.ncap, aa.nbegin != nullptr) + 64, 0xCD);
      aa.list = lists.data() + static_cast<size_t>(c) * n; aa.list_count = &list_counts[c];
      for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, smem.data(), [&] { align_block(aa, smem.data()); });
      aa.list = alists.data() + static_cast<size_t>(c) * n; aa.list_count = &acounts[c];
      for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, smem.data(), [&] { align_block(aa, smem.data()); });
    }
    if (status_out) *status_out = status;
    if (status) return -1;
  }
  return static_cast<int64_t>(total);
}

// The spans form (kernels_align.h): begin / end hold cap entries.
int64_t emu_encode_spans_batch(void *hv, const uint8_t *text, const uint64_t *offs, uint64_t n, int32_t *ids, uint32_t *begin,
                               uint32_t *end, uint64_t cap, uint64_t *id_offs, int grid, uint32_t *status_out,
                               uint32_t *nbegin, uint32_t *nend) {
  g_span_begin = begin; g_span_end = end; g_nspan_begin = nbegin; g_nspan_end = nbegin ? nend : nullptr;
  const int64_t r = emu_encode_batch(hv, text, offs, n, ids, cap, id_offs, grid, status_out);
  g_span_begin = g_span_end = g_nspan_begin = g_nspan_end = nullptr;
  return r;
}

// Batch Normalize as csrc/api.cc runs it: classify -> count pass per class -> scan -> write pass per class.
// Returns total normalized bytes, -(needed) - 2 if cap is too small, -1 with the status on a failure.
static uint32_t g_device_text = 0;    // set by emu_nbest_batch around its normalize call

int64_t emu_normalize_batch(void *hv, const uint8_t *text, const uint64_t *offs, uint64_t n, uint8_t *norm, uint64_t cap,
                            uint64_t *norm_offs, uint32_t *n2o, int grid, uint32_t *status_out) {
  auto *h = static_cast<EmuHandle *>(hv);
  const SpmxDev &dev = h->tables.scalars;
  const bool bpe = dev.model_type == 2;
  const LengthClass *cls = bpe ? kBpeCls : kUniCls;
  const int ncls = bpe ? static_cast<int>(sizeof(kBpeCls) / sizeof(kBpeCls[0])) : static_cast<int>(sizeof(kUniCls) / sizeof(kUniCls[0]));
  if (grid < 1) grid = 1;
  std::vector<uint32_t> lists(static_cast<size_t>(ncls) * (n ? n : 1)), list_counts(kMaxClasses, 0), counts(n + 1, 0);
  ClassifyArgs ca{};
  ca.offs = offs; ca.n = static_cast<uint32_t>(n); ca.n_classes = static_cast<uint32_t>(ncls);
  for (int c = 0; c < ncls; ++c) ca.rcap[c] = cls[c].rcap;
  ca.lists = lists.data(); ca.list_counts = list_counts.data();
  std::vector<uint32_t> key_totals(kSortKeys, 0), key_cursor(kSortKeys, 0), hist(3 * kSortKeys, 0);
  ca.key_totals = key_totals.data(); ca.key_cursor = key_cursor.data();
  ca.sub_buckets = getenv("SPMX_SUB_BUCKETS") ? static_cast<uint32_t>(atoi(getenv("SPMX_SUB_BUCKETS"))) : static_cast<uint32_t>(kSubBuckets);
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { classify_block<0>(ca, hist.data()); });
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { classify_block<1>(ca, hist.data()); });
  uint32_t status = 0;
  auto pass = [&](bool write) {
    for (int c = 0; c < ncls; ++c) {
      if (cls[c].rcap > kEmuMaxStagedRaw) { if (list_counts[c]) status |= kStTooLong; continue; }
      NormalizeArgs a{};
      a.dev = dev; a.text = text; a.offs = offs;
      a.list = lists.data() + static_cast<size_t>(c) * n; a.list_count = &list_counts[c];
      const bool has_next = c + 1 < ncls && cls[c + 1].rcap <= kEmuMaxStagedRaw;
      a.next_list = has_next ? lists.data() + static_cast<size_t>(c + 1) * n : nullptr;
      a.next_count = has_next ? &list_counts[c + 1] : nullptr;
      a.counts = counts.data(); a.norm_offs = norm_offs; a.norm = norm; a.n2o = n2o; a.status = &status;
      a.rcap = cls[c].rcap; a.ncap = cls[c].ncap;
      a.device_text = g_device_text;
      std::vector<unsigned char> smem(NormalizeLdsBytes(a.rcap, a.ncap) + 64, 0xCD);
      for (int b = 0; b < grid; ++b) {
        if (write) emu::RunWave(b, grid, smem.data(), [&] { normalize_block<true>(a, smem.data()); });
        else emu::RunWave(b, grid, smem.data(), [&] { normalize_block<false>(a, smem.data()); });
      }
    }
  };
  pass(false);
  std::vector<uint64_t> tile_sums((n + kScanTile - 1) / kScanTile + 2, 0);
  ScanArgs sa{counts.data(), static_cast<uint32_t>(n), tile_sums.data(), norm_offs};
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_tiles_block(sa); });
  emu::RunWave(0, 1, nullptr, [&] { scan_sums_block(sa); });
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_final_block(sa); });
  if (status_out) *status_out = status;
  if (status) return -1;
  const uint64_t total = norm_offs[n];
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  pass(true);
  if (status_out) *status_out = status;
  return status ? -1 : static_cast<int64_t>(total);
}

// Batch Decode as csrc/api.cc runs it: count pass -> scan -> write pass.  Returns total bytes, -(needed) - 2 if cap
// is too small, -1 with the device status in *status_out on a bad id.
int64_t emu_decode_batch(void *hv, const int32_t *ids, const uint64_t *id_offs, uint64_t n, uint8_t *text, uint64_t cap,
                         uint64_t *text_offs, int grid, uint32_t *status_out) {
  auto *h = static_cast<EmuHandle *>(hv);
  if (grid < 1) grid = 1;
  std::vector<uint32_t> counts(n + 1, 0);
  uint32_t status = 0;
  unsigned long long bad_key = ~0ull;
  DecodeArgs a{};
  a.dev = h->tables.scalars; a.ids = ids; a.id_offs = id_offs; a.n = static_cast<uint32_t>(n);
  a.counts = counts.data(); a.text_offs = text_offs; a.text = text; a.text_cap = cap;
  a.status = &status; a.bad_key = &bad_key;
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { decode_block<false>(a); });
  std::vector<uint64_t> tile_sums((n + kScanTile - 1) / kScanTile + 2, 0);
  ScanArgs sa{counts.data(), static_cast<uint32_t>(n), tile_sums.data(), text_offs};
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_tiles_block(sa); });
  emu::RunWave(0, 1, nullptr, [&] { scan_sums_block(sa); });
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_final_block(sa); });
  if (status_out) *status_out = status;
  if (status) return -1;
  const uint64_t total = text_offs[n];
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { decode_block<true>(a); });
  return static_cast<int64_t>(total);
}

// The corpus packer as csrc/api.cc runs it: count pass -> scan -> write pass.  `file` must be 16-byte aligned and
// readable to the next multiple of 16.  Returns the number of lines; *text_bytes = bytes of packed text.
int64_t emu_split_lines(const uint8_t *file, uint64_t bytes, uint8_t *text, uint64_t *offsets, int grid, uint64_t *text_bytes) {
  if (grid < 1) grid = 1;
  if (bytes == 0) { offsets[0] = 0; *text_bytes = 0; return 0; }
  const uint64_t chunks = (bytes + kSplitChunk - 1) / kSplitChunk;
  std::vector<uint32_t> counts(chunks + 1, 0);
  std::vector<uint64_t> base(chunks + 1, 0), tile_sums((chunks + kScanTile - 1) / kScanTile + 2, 0);
  SplitArgs a{};
  a.file = file; a.bytes = bytes; a.counts = counts.data(); a.chunk_base = base.data(); a.text = text; a.offsets = offsets;
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { split_block<false>(a, nullptr); });
  ScanArgs sa{counts.data(), static_cast<uint32_t>(chunks), tile_sums.data(), base.data()};
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_tiles_block(sa); });
  emu::RunWave(0, 1, nullptr, [&] { scan_sums_block(sa); });
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { scan_final_block(sa); });
  const uint64_t nl = base[chunks];
  alignas(16) static unsigned char stage[kSplitLdsBytes];
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, stage, [&] { split_block<true>(a, stage); });
  *text_bytes = bytes - nl;
  return static_cast<int64_t>(nl + (file[bytes - 1] != 0x0A ? 1 : 0));
}

// NBestEncode as csrc/api.cc runs it: Normalize kernels (device text) -> NBest kernel -> host CSR.
// Outputs: result r of the batch has ids out[id_offs[r], id_offs[r + 1]) and scores[r]; sentence s owns results
// [res_offs[s], res_offs[s + 1]).  id_offs / scores hold n * nbest (+ 1) entries.  Returns the number of results,
// -1 with the status on a failure, -(ids needed) - 2 if cap is too small.
int64_t emu_nbest_batch(void *hv, const uint8_t *text, const uint64_t *offs, uint64_t n, int nbest, int32_t *out, uint64_t cap,
                        uint64_t *id_offs, float *scores, uint64_t *res_offs, int grid, uint32_t *status_out) {
  auto *h = static_cast<EmuHandle *>(hv);
  if (grid < 1) grid = 1;
  const uint64_t tbytes = offs[n] - offs[0];
  std::vector<uint8_t> norm(tbytes * 20 + 64);
  std::vector<uint64_t> norm_offs(n + 1, 0);
  uint32_t status = 0;
  g_device_text = 1;
  const int64_t tot = emu_normalize_batch(hv, text, offs, n, norm.data(), norm.size(), norm_offs.data(), nullptr, grid, &status);
  g_device_text = 0;
  if (status_out) *status_out = status;
  if (tot < 0) return -1;
  NBestArgs a{};
  a.dev = h->tables.scalars; a.norm = norm.data(); a.norm_offs = norm_offs.data(); a.n = static_cast<uint32_t>(n);
  a.nbest = static_cast<uint32_t>(nbest);
  a.max_hyps = 65536;
  a.lane_bytes = (NbestLaneBytes(a.max_hyps) + 15) / 16 * 16;
  std::vector<uint8_t> scratch(static_cast<size_t>(grid) * 64 * a.lane_bytes, 0xCD);
  a.scratch = scratch.data();
  std::vector<int32_t> arena(static_cast<size_t>(tot + 8 * n + 64) * nbest + 1024);
  unsigned long long head = 0;
  a.arena = arena.data(); a.arena_head = &head; a.arena_cap = arena.size();
  std::vector<unsigned long long> roff(n * nbest + 1, 0);
  std::vector<uint32_t> rlen(n * nbest + 1, 0), rcount(n + 1, 0);
  std::vector<float> rscore(n * nbest + 1, 0.f);
  a.res_off = roff.data(); a.res_len = rlen.data(); a.res_score = rscore.data(); a.res_count = rcount.data();
  a.status = &status;
  for (int b = 0; b < grid; ++b) emu::RunWave(b, grid, nullptr, [&] { nbest_block(a); });
  if (status_out) *status_out = status;
  if (status) return -1;
  uint64_t r = 0, total = 0;
  for (uint64_t s = 0; s < n; ++s) {
    res_offs[s] = r;
    for (uint32_t k = 0; k < rcount[s]; ++k) {
      id_offs[r] = total;
      scores[r] = rscore[s * nbest + k];
      const uint32_t len = rlen[s * nbest + k];
      if (total + len <= cap) for (uint32_t i = 0; i < len; ++i) out[total + i] = arena[roff[s * nbest + k] + i];
      total += len;
      ++r;
    }
  }
  res_offs[n] = r;
  id_offs[r] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  return static_cast<int64_t>(r);
}

uint64_t emu_collectives() { return emu::g_wave.n_collectives; }
uint32_t emu_flags(void *hv) { return static_cast<EmuHandle *>(hv)->tables.scalars.flags; }
// sentences the FAST tile kernel kept / handed to the GENERAL kernel in the last emu_encode_batch
uint64_t g_fast_kept = 0, g_fast_handed = 0, g_wave_handed = 0;
uint64_t emu_wave_handed() { return g_wave_handed; }
uint64_t emu_fast_kept() { return g_fast_kept; }
uint64_t emu_fast_handed() { return g_fast_handed; }

}  // extern "C"
