// Microbenchmark behind DESIGN.md's cost model of the per-lane search loop: what
// does ONE iteration of "64 lanes each chase their own pointer through a table"
// cost on gfx950, and what do the LDS byte accesses add?
//   hipcc --offload-arch=gfx950 -O3 -o probe_latency probe_latency.hip && ./probe_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct U4 { uint32_t x, y, z, w; };

template <int MODE>
__global__ __launch_bounds__(64) void chase(const U4 *tab, uint32_t mask, int iters, unsigned long long *out, uint32_t *sink) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x;
  uint8_t *mine = lds + lane * 403;          // per-lane byte region, odd stride like the real kernel
  float *ring = reinterpret_cast<float *>(lds + 64 * 403 + 64) + lane;
  for (int i = lane; i < 64 * 403 + 64 + 64 * 16 * 4; i += 64) lds[i] = (unsigned char)(i * 7);
  __syncthreads();
  uint32_t idx = (lane * 2654435761u + blockIdx.x * 40503u) & mask;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE & 1) {                          // the dependent, scattered 12-byte load
      const U4 u = tab[idx];
      idx = (u.x + acc) & mask;
      acc += u.y ^ u.z;
    } else {
      idx = (idx * 1664525u + 1013904223u) & mask;
    }
    if (MODE & 2) {                          // LDS traffic of one relax: 3 byte reads, 1 dword read, 2 conditional writes
      const uint32_t p = idx % 400u;
      const uint32_t a = mine[p], b = mine[(p + 1) % 400u], c = mine[(p + 2) % 400u];
      float *slot = ring + ((p & 15u) << 6);
      const float rv = *slot;
      if ((a + b + c + (uint32_t)rv) & 1u) { *slot = rv + 1.0f; mine[p] = (uint8_t)(a + 1); }
      acc += a ^ b ^ c;
    }
    if (MODE & 4) {                          // the loop's wave-wide vote
      if (__ballot(acc == 0xFFFFFFFFu) == ~0ull) break;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345u) sink[0] = acc + idx;
}

int main() {
  const int iters = 2000;
  for (uint32_t units : {4096u, 262144u}) {
    std::vector<U4> h(units);
    srand(1);
    for (auto &u : h) { u.x = rand(); u.y = rand(); u.z = rand(); u.w = 0; }
    U4 *d; unsigned long long *out; uint32_t *sink;
    hipMalloc(&d, units * sizeof(U4)); hipMemcpy(d, h.data(), units * sizeof(U4), hipMemcpyHostToDevice);
    hipMalloc(&out, 8192 * 8); hipMalloc(&sink, 4);
    for (int waves_per_cu : {1, 4, 8, 16}) {
      const int grid = 256 * waves_per_cu;
      const size_t lds = 64 * 403 + 64 + 64 * 16 * 4;
      auto run = [&](auto kern, const char *name) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, 0, d, units - 1, iters, out, sink);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, 0, d, units - 1, iters, out, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(grid);
        hipMemcpy(t.data(), out, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : t) s += v;
        printf("table %7u units  grid %5d  %-28s  %8.1f ticks/iter/wave   %8.3f ms  -> %7.1f ns/iter wall per wave\n", units, grid, name,
               s / grid / iters, ms, ms * 1e6 / iters);
      };
      run(chase<1>, "load only");
      run(chase<2>, "lds only");
      run(chase<3>, "load + lds");
      run(chase<7>, "load + lds + ballot");
    }
    hipFree(d); hipFree(out); hipFree(sink);
  }
  return 0;
}
