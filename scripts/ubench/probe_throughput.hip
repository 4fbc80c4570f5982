// Throughput of fully scattered per-lane loads on gfx950 as occupancy grows
// (no LDS, so up to 32 waves / CU are resident): how many CU-cycles does one
// wave-wide "every lane its own cache line" load cost once latency is hidden?
//   hipcc --offload-arch=gfx950 -O3 -o probe_throughput probe_throughput.hip && ./probe_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct U4 { uint32_t x, y, z, w; };

template <int BYTES, int CHAINS>
__global__ __launch_bounds__(64) void chase(const unsigned char *tab, uint32_t mask, int iters, uint32_t *sink) {
  const int lane = threadIdx.x;
  uint32_t idx[CHAINS];
  for (int k = 0; k < CHAINS; ++k) idx[k] = (lane * 2654435761u + blockIdx.x * 40503u + k * 977u) & mask;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
      uint32_t v;
      if (BYTES == 1) v = tab[idx[k] * 16u] * 2654435761u;
      else if (BYTES == 4) v = *reinterpret_cast<const uint32_t *>(tab + idx[k] * 16u);
      else { const U4 u = *reinterpret_cast<const U4 *>(tab + idx[k] * 16u); v = u.x ^ u.y ^ u.z; }
      idx[k] = (v + acc) & mask;
      acc += v >> 7;
    }
  }
  if (acc == 12345u) sink[0] = acc;
}

int main() {
  const int iters = 4000;
  for (uint32_t units : {4096u, 65536u, 262144u}) {
    std::vector<U4> h(units);
    srand(1);
    for (auto &u : h) { u.x = rand(); u.y = rand(); u.z = rand(); u.w = 0; }
    unsigned char *d; uint32_t *sink;
    hipMalloc(&d, units * sizeof(U4)); hipMemcpy(d, h.data(), units * sizeof(U4), hipMemcpyHostToDevice);
    hipMalloc(&sink, 4);
    for (int waves_per_cu : {1, 4, 8, 16, 32}) {
      const int grid = 256 * waves_per_cu;
      auto run = [&](auto kern, const char *name, int chains) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, units - 1, iters, sink);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, units - 1, iters, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double loads = double(grid) * iters * chains;     // wave-wide load instructions
        printf("table %7u units (%5.1f MB)  %2d waves/CU  %-22s %8.3f ms  %7.1f ns per dependent step   %6.2f ns of CU time per wave-load  (%.1f G lane-loads/s)\n",
               units, units * 16 / 1048576.0, waves_per_cu, name, ms, ms * 1e6 / iters, ms * 1e6 * 256 / loads, loads * 64 / ms / 1e6);
      };
      run(chase<12, 1>, "12B x1 chain", 1);
      run(chase<12, 2>, "12B x2 chains", 2);
      run(chase<12, 4>, "12B x4 chains", 4);
      run(chase<4, 1>, "4B x1 chain", 1);
      run(chase<1, 1>, "1B x1 chain", 1);
    }
    hipFree(d); hipFree(sink);
  }
  return 0;
}
