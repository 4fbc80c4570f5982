// TEST INFRASTRUCTURE ONLY: a lock-step CPU model of the wv:: wave API
// (sentencepiece_amd/csrc/wave.h) so that the device bodies in kernels.h can
// be executed, stepped through and checked against the oracle in a container
// without a GPU.  It is not a product back end: the product library
// (libspmx.so) is compiled by hipcc from wave.h and fails loudly without a
// GPU.  Nothing under sentencepiece_amd/ includes this file.
//
// Model: one wavefront = 64 fibers (one per lane) on one OS thread.  A lane
// runs until it reaches a wv:: collective, then parks; when all 64 lanes have
// parked at the SAME collective the scheduler computes the result and resumes
// them.  A lane that exits while others wait in a collective, or lanes that
// meet at different collectives, abort the test ("divergent collective") --
// exactly the discipline the real kernels rely on.  Between collectives lanes
// run one after another, so cross-lane LDS traffic without a wv::sync() in
// between is caught by the optional LDS race check.
#ifndef SPMX_WAVE_EMU_H_
#define SPMX_WAVE_EMU_H_
#define SPMX_WAVE_API 1
#define SPMX_DEVICE inline
#define SPMX_DEVICE_CALL inline

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#include "dev.h"

namespace spmx {
namespace emu {

enum Op { kNone = 0, kBallot, kShfl, kShflUp, kSync, kLaneUp1, kLaneDown1, kScanAdd, kScanMax, kBlockSync };

struct Lane {
  void *sp = nullptr;        // saved stack pointer
  unsigned char *stack = nullptr;
  bool done = false;
  Op op = kNone;
  uint64_t a = 0, b = 0, out = 0;
};

struct Wave {
  Lane lanes[64];
  void *sched_sp = nullptr;
  int cur = 0;
  int block = 0, grid = 1;
  int wib = 0, wpb = 1;      // wave index within its workgroup, waves per workgroup
  std::function<void()> body;
  std::function<void()> block_barrier;   // (workgroups whose wavefronts run side by side, emu_launch.cc RunGridTogether) what wv::block_sync() waits at
  unsigned char *smem = nullptr;
  uint64_t n_collectives = 0;
};

extern thread_local Wave g_wave;   // (one emulated wavefront per host thread: the pipelined host form has workers)
extern "C" void spmx_emu_switch(void **save_sp, void *load_sp);

void RunWave(int block, int grid, unsigned char *smem, const std::function<void()> &body);

inline uint64_t Collective(Op op, uint64_t a, uint64_t b) {
  Lane &l = g_wave.lanes[g_wave.cur];
  l.op = op; l.a = a; l.b = b;
  spmx_emu_switch(&l.sp, g_wave.sched_sp);
  return l.out;
}

}  // namespace emu

namespace wv {

inline int lane() { return emu::g_wave.cur; }
inline int block_id() { return emu::g_wave.block; }
inline int grid_size() { return emu::g_wave.grid; }
inline int wave_in_block() { return emu::g_wave.wib; }
inline int waves_per_block() { return emu::g_wave.wpb; }

inline uint64_t ballot(bool p) { return emu::Collective(emu::kBallot, p ? 1 : 0, 0); }
inline bool any(bool p) { return ballot(p) != 0; }

inline uint32_t shfl(uint32_t v, int src) { return static_cast<uint32_t>(emu::Collective(emu::kShfl, v, static_cast<uint64_t>(src & 63))); }
inline int shfl(int v, int src) { return static_cast<int>(shfl(static_cast<uint32_t>(v), src)); }
inline float shfl(float v, int src) { uint32_t u; memcpy(&u, &v, 4); u = shfl(u, src); memcpy(&v, &u, 4); return v; }
inline double shfl(double v, int src) {
  uint64_t u; memcpy(&u, &v, 8);
  u = emu::Collective(emu::kShfl, u, static_cast<uint64_t>(src & 63));
  memcpy(&v, &u, 8);
  return v;
}
inline int shfl_up(int v, int delta) {
  return static_cast<int>(static_cast<uint32_t>(emu::Collective(emu::kShflUp, static_cast<uint32_t>(v), static_cast<uint64_t>(delta))));
}
inline uint32_t lane_up1(uint32_t v, uint32_t fill) { return static_cast<uint32_t>(emu::Collective(emu::kLaneUp1, v, fill)); }
inline uint32_t lane_down1(uint32_t v, uint32_t fill) { return static_cast<uint32_t>(emu::Collective(emu::kLaneDown1, v, fill)); }
inline uint32_t scan_add(uint32_t v) { return static_cast<uint32_t>(emu::Collective(emu::kScanAdd, v, 0)); }
inline uint32_t scan_max(uint32_t v) { return static_cast<uint32_t>(emu::Collective(emu::kScanMax, v, 0)); }
inline void opaque(uint32_t &) {}
inline uint32_t uniform(uint32_t v) { return v; }
inline uint64_t uniform64(uint64_t v) { return v; }
inline uint32_t read_lane(uint32_t v, int src) { return static_cast<uint32_t>(emu::Collective(emu::kShfl, v, static_cast<uint64_t>(src & 63))); }
template <int N> inline void keep_apart() {}
inline void sync() { emu::Collective(emu::kSync, 0, 0); }
inline void block_sync() { emu::Collective(emu::kBlockSync, 0, 0); }
inline void sync_global() { emu::Collective(emu::kSync, 0, 0); }

inline uint32_t atomic_add(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline unsigned long long atomic_add(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
inline void atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
inline uint32_t lds_atomic_add(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline void lds_atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
inline void lds_atomic_min(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
inline void atomic_min(unsigned long long *p, unsigned long long v) { if (v < *p) *p = v; }
inline void atomic_max(unsigned long long *p, unsigned long long v) { if (v > *p) *p = v; }
inline void atomic_and(uint32_t *p, uint32_t v) { *p &= v; }
inline unsigned long long atomic_cas(unsigned long long *p, unsigned long long expect, unsigned long long v) {
  const unsigned long long o = *p;
  if (o == expect) *p = v;
  return o;
}
inline uint32_t atomic_load(const uint32_t *p) { return *p; }
inline unsigned long long atomic_load64(const unsigned long long *p) { return *p; }

inline void store_stream(U2 *p, const U2 &v) { *p = v; }
inline Q4 load_stream(const Q4 *p) { return *p; }
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n) { return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (8u * (n & 3u))); }
inline unsigned long long clock() { return 0; }

inline int popc64(uint64_t x) { return __builtin_popcountll(x); }
inline int ffs64(uint64_t x) { return __builtin_ffsll(static_cast<long long>(x)); }
inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline double bits_to_double(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
#ifndef SPMX_EXP
#define SPMX_EXP 0
#endif

}  // namespace wv
}  // namespace spmx
#endif
