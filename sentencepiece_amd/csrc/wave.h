// Wave-level primitives for gfx950 (64-lane wavefronts).  Everything the
// kernels need across lanes goes through this small API so that every
// cross-lane operation sits in wave-uniform control flow (all 64 lanes call
// it together).  tests/emu/ provides a lock-step CPU model of the same API to
// run the kernel bodies under a debugger without a GPU (test seam only).
#ifndef SPMX_WAVE_H_
#define SPMX_WAVE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev.h"

#define SPMX_DEVICE __device__ __forceinline__
#define SPMX_DEVICE_CALL __device__ __attribute__((noinline))   // a real call: rare paths that sit inside unrolled code

namespace spmx {
namespace wv {

SPMX_DEVICE int lane() { return static_cast<int>(threadIdx.x) & 63; }
SPMX_DEVICE int block_id() { return static_cast<int>(blockIdx.x); }
SPMX_DEVICE int grid_size() { return static_cast<int>(gridDim.x); }
// workgroups of several wavefronts (tile kernels): this wave's index in its workgroup, waves per workgroup
SPMX_DEVICE int wave_in_block() { return static_cast<int>(threadIdx.x) >> 6; }
SPMX_DEVICE int waves_per_block() { return static_cast<int>(blockDim.x) >> 6; }

SPMX_DEVICE uint64_t ballot(bool p) { return __ballot(p ? 1 : 0); }
SPMX_DEVICE bool any(bool p) { return __ballot(p ? 1 : 0) != 0ull; }

SPMX_DEVICE uint32_t shfl(uint32_t v, int src) { return static_cast<uint32_t>(__shfl(static_cast<int>(v), src, 64)); }
SPMX_DEVICE int shfl(int v, int src) { return __shfl(v, src, 64); }
SPMX_DEVICE float shfl(float v, int src) { return __shfl(v, src, 64); }
SPMX_DEVICE double shfl(double v, int src) {
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t lo = shfl(static_cast<uint32_t>(u), src), hi = shfl(static_cast<uint32_t>(u >> 32), src);
  return __longlong_as_double(static_cast<long long>(static_cast<uint64_t>(hi) << 32 | lo));
}
// value of lane (lane - delta); lanes < delta keep their own value
SPMX_DEVICE int shfl_up(int v, int delta) { return __shfl_up(v, static_cast<unsigned>(delta), 64); }

// ---- cross-lane moves and scans on the DPP path (no LDS traffic; gfx9 row_shr / row_bcast / wave_shr controls) ----
// value of lane - 1; lane 0 gets `fill`
SPMX_DEVICE uint32_t lane_up1(uint32_t v, uint32_t fill) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(fill), static_cast<int>(v), 0x138, 0xF, 0xF, false));   // wave_shr:1
}
// value of lane + 1; lane 63 gets `fill`
SPMX_DEVICE uint32_t lane_down1(uint32_t v, uint32_t fill) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(fill), static_cast<int>(v), 0x130, 0xF, 0xF, false));   // wave_shl:1
}
template <int CTRL, int ROWS>
SPMX_DEVICE uint32_t dpp_or0(uint32_t v) {      // the moved value, 0 where the control names no source lane / the row is masked
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROWS, 0xF, false));
}
// inclusive prefix sum / prefix maximum over the 64 lanes (unsigned; six DPP steps: row_shr 1, 2, 4, 8, row_bcast 15, 31)
SPMX_DEVICE uint32_t scan_add(uint32_t v) {
  v += dpp_or0<0x111, 0xF>(v);
  v += dpp_or0<0x112, 0xF>(v);
  v += dpp_or0<0x114, 0xF>(v);
  v += dpp_or0<0x118, 0xF>(v);
  v += dpp_or0<0x142, 0xA>(v);
  v += dpp_or0<0x143, 0xC>(v);
  return v;
}
SPMX_DEVICE uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
SPMX_DEVICE uint32_t scan_max(uint32_t v) {
  v = umax(v, dpp_or0<0x111, 0xF>(v));
  v = umax(v, dpp_or0<0x112, 0xF>(v));
  v = umax(v, dpp_or0<0x114, 0xF>(v));
  v = umax(v, dpp_or0<0x118, 0xF>(v));
  v = umax(v, dpp_or0<0x142, 0xA>(v));
  v = umax(v, dpp_or0<0x143, 0xC>(v));
  return v;
}
// A marker the optimizer cannot merge with another one: at the end of the arms of an `if` whose arms run the same code on
// DIFFERENT register sets, it keeps the arms from being folded into one block that picks its operands by address (which
// would put both sets into scratch memory).
template <int N>
SPMX_DEVICE void keep_apart() { asm volatile("; keep_apart %0" ::"n"(N)); }
// the value of ONE lane as a scalar (src wave-uniform)
SPMX_DEVICE uint32_t read_lane(uint32_t v, int src) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), src)); }

// the optimizer forgets what it knows about v (no instruction): what is computed from v behind this is computed here
SPMX_DEVICE void opaque(uint32_t &v) { asm volatile("" : "+v"(v)); }
// a value every lane holds alike, as a scalar (v_readfirstlane): loops over it run on the scalar unit
SPMX_DEVICE uint32_t uniform(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
SPMX_DEVICE uint64_t uniform64(uint64_t v) { return static_cast<uint64_t>(uniform(static_cast<uint32_t>(v >> 32))) << 32 | uniform(static_cast<uint32_t>(v)); }

// Orders this wave's LDS traffic: writes before the call are visible to every
// lane's reads after it.  A wave executes in lock step, so only the compiler
// and the LDS queue need to be fenced -- no s_barrier.
SPMX_DEVICE void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier of a kernel whose wavefronts work TOGETHER (kernels_uniwave.h uni_long_pipe_block: a walker and a folder):
// every wavefront of the workgroup arrives, LDS writes before it are visible behind it.
SPMX_DEVICE void block_sync() { __syncthreads(); }

// As sync(), for HBM: this wave's global stores before the call are visible to every lane's loads after it
// (all lanes of a wave share one vector L1; the fence drains the store queue).
SPMX_DEVICE void sync_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

SPMX_DEVICE uint32_t atomic_add(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
SPMX_DEVICE unsigned long long atomic_add(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
SPMX_DEVICE void atomic_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }
SPMX_DEVICE uint32_t lds_atomic_add(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }   // p in LDS
SPMX_DEVICE void lds_atomic_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }                  // p in LDS
SPMX_DEVICE void lds_atomic_min(uint32_t *p, uint32_t v) { atomicMin(p, v); }                // p in LDS
SPMX_DEVICE void atomic_min(unsigned long long *p, unsigned long long v) { atomicMin(p, v); }
SPMX_DEVICE void atomic_max(unsigned long long *p, unsigned long long v) { atomicMax(p, v); }
SPMX_DEVICE void atomic_and(uint32_t *p, uint32_t v) { atomicAnd(p, v); }
// compare-and-swap on a 64-bit word in HBM (the call-local word memo's tags, kernels_word.h): returns the old value
SPMX_DEVICE unsigned long long atomic_cas(unsigned long long *p, unsigned long long expect, unsigned long long v) { return atomicCAS(p, expect, v); }
// a load that sees what other workgroups' atomics wrote (tile queue of the streaming kernels)
SPMX_DEVICE uint32_t atomic_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SPMX_DEVICE unsigned long long atomic_load64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


// Streaming accesses: data written once and read once much later (the split form's candidate streams, kernels_matchfold.h)
// should not push the tables the probes live on out of L2 -- the non-temporal hint (slc / nt on gfx9).
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
SPMX_DEVICE void store_stream(U2 *p, const U2 &v) {                    // one 8-byte store (p is 8-byte aligned)
  u32x2_t w;
  w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<u32x2_t *>(p));
}
SPMX_DEVICE Q4 load_stream(const Q4 *p) {                              // one 16-byte load
  const u32x4_t w = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
  return Q4{w.x, w.y, w.z, w.w};
}

// the 64-bit value hi:lo shifted right by n & 3 bytes, its low 32 bits (v_alignbyte_b32)
SPMX_DEVICE uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }
SPMX_DEVICE unsigned long long clock() { return __builtin_amdgcn_s_memtime(); }   // shader cycles

SPMX_DEVICE int popc64(uint64_t x) { return __popcll(x); }
SPMX_DEVICE int ffs64(uint64_t x) { return __ffsll(static_cast<unsigned long long>(x)); }   // 1-based, 0 if none
SPMX_DEVICE int clz64(uint64_t x) { return __clzll(static_cast<long long>(x)); }
SPMX_DEVICE float bits_to_float(uint32_t u) { return __uint_as_float(u); }
SPMX_DEVICE uint32_t float_to_bits(float f) { return __float_as_uint(f); }
SPMX_DEVICE double bits_to_double(uint64_t u) { return __longlong_as_double(static_cast<long long>(u)); }

// Experiment builds (-DSPMX_EXP=<bits>, csrc/Makefile `variants`; results are WRONG, only the counters mean something):
// which store stream writes how much -- 1 drops the id stores into the arena, 2 the back-pointer block stores, 4 the
// text-column stores of the ASCII normalizer; word kernels (kernels_word.h): 8 one more text gather per iteration, 16 no
// id bursts, 32 one more memo gather per iteration.
#ifndef SPMX_EXP
#define SPMX_EXP 0
#endif

}  // namespace wv
}  // namespace spmx
#endif
