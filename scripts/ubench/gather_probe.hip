// Probe: what does a 64-lane GATHER cost on the vector memory path of a gfx950 CU, by width (4 / 16 bytes per lane),
// by the number of active lanes, and by where the lines live (L1-sized, L2-sized, beyond)?  16 waves per CU, every
// lane reads table[(random index) * stride]; reports cycles per wave-instruction per CU at saturation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct alignas(16) Q4 { uint32_t x, y, z, w; };
template <int WIDTH>
__global__ __launch_bounds__(1024) void gather(const uint32_t *tab, uint32_t mask_entries, uint32_t iters, uint32_t active, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  // `active` lanes take part: spread over the wave (every 64/active-th lane)
  const bool on = (lane * active) / 64u != ((lane + 1u) * active) / 64u || active == 64u;
  if (on) {
    for (uint32_t i = 0; i < iters; ++i) {
      s = s * 1664525u + 1013904223u;
      const uint32_t idx = (s >> 8) & mask_entries;
      if (WIDTH == 16) { const Q4 v = reinterpret_cast<const Q4 *>(tab)[idx]; acc += v.x ^ v.w; }
      else acc += tab[idx * 4u];
    }
  }
  out[blockIdx.x * 1024u + threadIdx.x] = acc;
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  const size_t bytes = 256u << 20;
  uint32_t *tab, *out;
  hipMalloc(&tab, bytes); hipMalloc(&out, (size_t)cus * 1024 * 4);
  hipMemset(tab, 1, bytes);
  const uint32_t iters = 2000;
  for (int width : {4, 16}) for (uint32_t kb : {16u, 2048u, 131072u}) for (uint32_t active : {64u, 32u, 16u, 8u, 4u}) {
    const uint32_t entries = kb * 1024u / 16u;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (width == 16) gather<16><<<cus, 1024>>>(tab, entries - 1, iters, active, out);
      else gather<4><<<cus, 1024>>>(tab, entries - 1, iters, active, out);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / (16.0 * iters);     // cycles per wave-instruction per CU (16 waves share the CU)
    printf("width %2d B  table %6u KB  active %2u lanes: %.3f ms, %.1f cycles / wave-gather / CU\n", width, kb, active, ms, cyc);
  }
  return 0;
}
