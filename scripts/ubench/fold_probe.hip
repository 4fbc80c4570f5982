// Probe: what does ONE STEP of the wave-cooperative fold (kernels_uniwave.h: the relaxations of one character start, scores in
// registers, the start's pieces one LDS matrix row) cost on gfx950, and which of its parts is it -- the issue slots, the
// chain from one start's score to the next, the LDS read?  One wavefront per workgroup, W workgroups per CU; every
// variant runs the same 64 steps per chunk over a synthetic matrix (3-8 entries reached per row, a third of them
// pieces) and reports shader cycles per step (s_memtime).
//   V0  the step as kernels_uniwave.h wrote it when this probe was made (entry fetched one step ahead)
//   V1  V0 with the score's conversion, the user-defined adjustment and the `has` mask prepared one step ahead too
//   V2  V1 without the double arithmetic (float add / compare: WRONG results, shows what f64 costs)
//   V3  V1 without the LDS read (the entry is a register: shows what the read costs)
//   V4  V1, two steps per loop trip (no register moves between trips)
//   V5  float arithmetic made exact (the sum of two floats whose exponents are within 28 of each other IS the double sum;
//       "greater" by the float sum, a tie decided by the sum's rounding error), "none" = NaN score, "unreached" = -inf:
//       one add, two compares, two selects a step
//   V7  V5 restated for the compiler: lane j holds position c + j (the start's own score is a lane too: no special first
//       step), the 64 steps unrolled with the start a constant (readlane of a fixed lane, the row's LDS offset an
//       immediate), lanes at or below the start masked by a NaN score, ties only RECORDED (a chunk with one is folded
//       again the slow way), the window beyond the chunk only in the last 8 steps
//   V6  V5, the 64 steps unrolled (kept for the record: the optimizer proves its result constant and removes it -- prints 0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
struct U2 { uint32_t x, y; };
constexpr uint32_t kNone = 0xFFFFFFFFu, kUnr = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// the rounding error of sum = a + b (TwoSum): the rare tie's decider, kept out of line so that the step does not pay for it
__device__ __attribute__((noinline)) float sum_err(float a, float b, float sum) {
  const float bb = sum - a;
  return (a - (sum - bb)) + (b - bb);
}
__device__ __forceinline__ double adj_of(uint32_t x) {
  return __longlong_as_double((long long)(0xBFB999999999999Aull & (uint64_t)(int64_t)((int32_t)x >> 31)));
}
template <int V>
__global__ __launch_bounds__(64) void fold(const U2 *init, uint32_t ML, uint32_t chunks, unsigned long long *cyc, float *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  U2 *M = reinterpret_cast<U2 *>(smem);
  const int lane = threadIdx.x & 63;
  for (uint32_t i = lane; i < 64u * ML; i += 64u) M[i] = init[i];
  __builtin_amdgcn_s_barrier();
  const uint32_t reach = 3u + ((uint32_t)lane * 7u) % 6u;
  float cur_s = V >= 5 ? -__builtin_inff() : 0.f, nxt_s = cur_s, s_c = 0.f;
  uint32_t cur_b = kUnr, nxt_b = kUnr;
  const uint32_t ML1 = ML - 1u;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t c = 0; c < chunks; ++c) {
    asm volatile("" ::: "memory");   // (the matrix is another one every chunk: nothing of it stays in registers)
    uint64_t m = ~0ull;
    m = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    int l = 0;
    U2 ent = M[(uint32_t)l * ML1 + (uint32_t)lane];
    uint32_t r = rl(reach, l);
    if (V == 0) {
      while (m != 0) {
        m &= m - 1;
        const int l2 = m ? __ffsll((unsigned long long)m) - 1 : l;
        const U2 ent2 = M[(uint32_t)l2 * ML1 + (uint32_t)lane];
        const uint32_t r2 = rl(reach, l2);
        const float bs = l == 0 ? s_c : __uint_as_float(rl(__float_as_uint(cur_s), l - 1));
        const double dbs = (double)bs;
        const uint32_t k = (uint32_t)(lane - l);
        {
          const bool has = k < r && ent.x != kNone;
          const double score = (double)__uint_as_float(ent.y) + adj_of(ent.x);
          const double cand = score + dbs;
          const bool win = has && (cur_b == kUnr || cand > (double)cur_s);
          cur_s = win ? (float)cand : cur_s;
          cur_b = win ? (ent.x & 0x7FFFFFFFu) : cur_b;
        }
        if ((uint32_t)l + r > 64u) {
          const uint32_t k2 = (uint32_t)(64 + lane - l);
          U2 e2{kNone, 0u};
          if (k2 < r) e2 = M[(uint32_t)l * ML + k2];
          const bool has = k2 < r && e2.x != kNone;
          const double cand = (double)__uint_as_float(e2.y) + adj_of(e2.x) + dbs;
          const bool win = has && (nxt_b == kUnr || cand > (double)nxt_s);
          nxt_s = win ? (float)cand : nxt_s;
          nxt_b = win ? (e2.x & 0x7FFFFFFFu) : nxt_b;
        }
        l = l2; ent = ent2; r = r2;
      }
    } else if (V == 5) {
      while (m != 0) {
        m &= m - 1;
        const int l2 = m ? __ffsll((unsigned long long)m) - 1 : l;
        const U2 ent2 = M[(uint32_t)l2 * ML1 + (uint32_t)lane];
        const float bs = l == 0 ? s_c : __uint_as_float(rl(__float_as_uint(cur_s), l - 1));
        const float a = __uint_as_float(ent.y);
        const float sum = a + bs;
        bool gt = sum > cur_s;
        if (__ballot(sum == cur_s) != 0ull) gt = gt || (sum == cur_s && sum_err(a, bs, sum) > 0.f);
        cur_s = gt ? sum : cur_s;
        cur_b = gt ? ent.x : cur_b;
        if ((c >> l) & 1u & (c >> 9)) {   // (rare, wave-uniform) the other window
          const U2 e2 = M[(uint32_t)l * ML + (uint32_t)lane];
          const float s2 = __uint_as_float(e2.y) + bs;
          const bool g2 = s2 > nxt_s;
          nxt_s = g2 ? s2 : nxt_s;
          nxt_b = g2 ? e2.x : nxt_b;
        }
        l = l2; ent = ent2;
      }
    } else if (V == 7) {
      uint32_t tie = 0;
      const float qnan = __uint_as_float(0x7FC00000u);
#pragma unroll
      for (int ll = 0; ll < 64; ++ll) {
        const U2 e = M[ll * 15 + lane];                               // entry [ll][lane - ll - 1] (+ 1: the pad in front)
        const float bs = __uint_as_float(rl(__float_as_uint(cur_s), ll));
        const float bsv = lane > ll ? bs : qnan;
        const float sum = __uint_as_float(e.y) + bsv;
        const bool gt = sum > cur_s;
        tie |= sum == cur_s ? 1u : 0u;
        cur_s = gt ? sum : cur_s;
        cur_b = gt ? e.x : cur_b;
        if (ll >= 56) {
          const U2 e2 = M[ll * 15 + 64 + lane];
          const float s2 = __uint_as_float(e2.y) + bs;
          const bool g2 = s2 > nxt_s;
          tie |= s2 == nxt_s ? 1u : 0u;
          nxt_s = g2 ? s2 : nxt_s;
          nxt_b = g2 ? e2.x : nxt_b;
        }
      }
      acc += tie;
    } else if (V == 6) {
#pragma unroll
      for (int ll = 0; ll < 64; ++ll) {
        if (!((m >> ll) & 1ull)) continue;
        const U2 e = M[(uint32_t)ll * 15u + (uint32_t)lane];
        const float bs = ll == 0 ? s_c : __uint_as_float(rl(__float_as_uint(cur_s), ll - 1));
        const float a = __uint_as_float(e.y);
        const float sum = a + bs;
        bool gt = sum > cur_s;
        if (__ballot(sum == cur_s) != 0ull) gt = gt || (sum == cur_s && sum_err(a, bs, sum) > 0.f);
        cur_s = gt ? sum : cur_s;
        cur_b = gt ? e.x : cur_b;
        if ((c >> ll) & 1u & (c >> 9)) {
          const U2 e2 = M[(uint32_t)ll * 16u + (uint32_t)lane];
          const float s2 = __uint_as_float(e2.y) + bs;
          const bool g2 = s2 > nxt_s;
          nxt_s = g2 ? s2 : nxt_s;
          nxt_b = g2 ? e2.x : nxt_b;
        }
      }
    } else if (V == 4) {
      // two steps a trip: A holds the even step's prepared operands, B the odd one's
      auto prep = [&](int ll, double *sc, bool *has, uint32_t *x, uint32_t *rr) __attribute__((always_inline)) {
        const U2 e = M[(uint32_t)ll * ML1 + (uint32_t)lane];
        *rr = rl(reach, ll);
        *sc = (double)__uint_as_float(e.y) + adj_of(e.x);
        *has = (uint32_t)(lane - ll) < *rr && e.x != kNone;
        *x = e.x & 0x7FFFFFFFu;
      };
      auto step = [&](int ll, double sc, bool has, uint32_t x, uint32_t rr) __attribute__((always_inline)) {
        const float bs = ll == 0 ? s_c : __uint_as_float(rl(__float_as_uint(cur_s), ll - 1));
        const double dbs = (double)bs;
        const double cand = sc + dbs;
        const bool win = has && (cur_b == kUnr || cand > (double)cur_s);
        cur_s = win ? (float)cand : cur_s;
        cur_b = win ? x : cur_b;
        if ((uint32_t)ll + rr > 64u) {
          const uint32_t k2 = (uint32_t)(64 + lane - ll);
          U2 e2{kNone, 0u};
          if (k2 < rr) e2 = M[(uint32_t)ll * ML + k2];
          const bool h2 = k2 < rr && e2.x != kNone;
          const double c2 = (double)__uint_as_float(e2.y) + adj_of(e2.x) + dbs;
          const bool w2 = h2 && (nxt_b == kUnr || c2 > (double)nxt_s);
          nxt_s = w2 ? (float)c2 : nxt_s;
          nxt_b = w2 ? (e2.x & 0x7FFFFFFFu) : nxt_b;
        }
      };
      double scA, scB; bool hA, hB; uint32_t xA, xB, rA, rB;
      prep(0, &scA, &hA, &xA, &rA);
      for (int ll = 0; ll < 64; ll += 2) {
        prep(ll + 1, &scB, &hB, &xB, &rB);
        step(ll, scA, hA, xA, rA);
        if (ll + 2 < 64) prep(ll + 2, &scA, &hA, &xA, &rA);
        step(ll + 1, scB, hB, xB, rB);
      }
    } else {
      double sc = (double)__uint_as_float(ent.y) + adj_of(ent.x);
      float scf = __uint_as_float(ent.y);
      bool has = (uint32_t)(lane - l) < r && ent.x != kNone;
      uint32_t x = ent.x & 0x7FFFFFFFu;
      while (m != 0) {
        m &= m - 1;
        const int l2 = m ? __ffsll((unsigned long long)m) - 1 : l;
        U2 ent2;
        if (V == 3) { ent2.x = x + 1u; ent2.y = __float_as_uint(scf * 0.5f); }
        else ent2 = M[(uint32_t)l2 * ML1 + (uint32_t)lane];
        const uint32_t r2 = rl(reach, l2);
        const float bs = l == 0 ? s_c : __uint_as_float(rl(__float_as_uint(cur_s), l - 1));
        if (V == 2) {
          const float cand = scf + bs;
          const bool win = has && (cur_b == kUnr || cand > cur_s);
          cur_s = win ? cand : cur_s;
          cur_b = win ? x : cur_b;
        } else {
          const double dbs = (double)bs;
          const double cand = sc + dbs;
          const bool win = has && (cur_b == kUnr || cand > (double)cur_s);
          cur_s = win ? (float)cand : cur_s;
          cur_b = win ? x : cur_b;
          if ((uint32_t)l + r > 64u) {
            const uint32_t k2 = (uint32_t)(64 + lane - l);
            U2 e2{kNone, 0u};
            if (k2 < r) e2 = M[(uint32_t)l * ML + k2];
            const bool h2 = k2 < r && e2.x != kNone;
            const double c2 = (double)__uint_as_float(e2.y) + adj_of(e2.x) + dbs;
            const bool w2 = h2 && (nxt_b == kUnr || c2 > (double)nxt_s);
            nxt_s = w2 ? (float)c2 : nxt_s;
            nxt_b = w2 ? (e2.x & 0x7FFFFFFFu) : nxt_b;
          }
        }
        // the next step's operands (nothing here waits for this step's result)
        sc = (double)__uint_as_float(ent2.y) + adj_of(ent2.x);
        scf = __uint_as_float(ent2.y);
        has = (uint32_t)(lane - l2) < r2 && ent2.x != kNone;
        x = ent2.x & 0x7FFFFFFFu;
        l = l2; r = r2;
      }
    }
    s_c = __uint_as_float(rl(__float_as_uint(cur_s), 63));
    acc ^= __float_as_uint(s_c) + cur_b;
    cur_s = nxt_s; cur_b = nxt_b; nxt_b = kUnr;
    if (V >= 5) nxt_s = -__builtin_inff();
    if ((c & 15u) == 15u) { s_c = 0.f; cur_s = V >= 5 ? -__builtin_inff() : 0.f; }    // (keep the magnitudes bounded)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(cyc, t1 - t0);
  out[blockIdx.x * 64 + lane] = __uint_as_float(acc ^ cur_b ^ __float_as_uint(s_c));
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  const uint32_t ML = 16, chunks = 4096;
  std::vector<U2> h(64 * ML);
  uint32_t s = 12345;
  size_t idx = 0;
  for (auto &e : h) {
    s = s * 1664525u + 1013904223u;
    const bool piece = (s >> 28) < 6 || (idx++ % ML) == 0;   // (every start has a one-byte piece: every position is reached)
    const float sc = -3.0f - (float)((s >> 8) & 0xFFFF) / 4096.0f;
    e.x = piece ? ((s >> 4) & 0x7FFFu) | (3u << 24) : kNone;
    memcpy(&e.y, &sc, 4);
    if (!piece) e.y = 0x7FC00000u;
  }
  U2 *d; unsigned long long *cyc; float *out;
  hipMalloc(&d, h.size() * 8); hipMalloc(&cyc, 8); hipMalloc(&out, (size_t)cus * 16 * 64 * 4);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  for (int w : {1, 4, 8, 16}) for (int v = 0; v < 8; ++v) {
    const int grid = cus * w;
    float best = 1e30f; unsigned long long c = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, 8);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      const size_t lds = 64 * ML * 8 + 1024;
      switch (v) {
        case 0: fold<0><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 1: fold<1><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 2: fold<2><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 3: fold<3><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 4: fold<4><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 5: fold<5><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 6: fold<6><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
        case 7: fold<7><<<grid, 64, lds>>>(d, ML, chunks, cyc, out); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("waves/CU %2d  V%d: %8.3f ms  %7.1f cycles/step/wave (s_memtime)  %6.1f ns/step wall\n", w, v, best,
           (double)c / ((double)grid * chunks * 64.0), best * 1e6 / (chunks * 64.0));
  }
  return 0;
}
