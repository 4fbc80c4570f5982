// Probe: do ds_read_b128 / ds_read_b32 take UNALIGNED LDS addresses on gfx950 (alignment mode "unaligned"), and at what
// cost?  Every lane reads 16 + 4 bytes at its own byte offset of a per-lane 96-byte LDS row; result checked on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
struct Q4v { uint32_t x, y, z, w; };
__device__ __forceinline__ Q4v lds_read16(uint32_t addr) {
  Q4v v;
  asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__global__ __launch_bounds__(64) void probe(const uint8_t *src, uint32_t shift, uint32_t iters, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t rows[64 * 96];
  const uint32_t lane = threadIdx.x;
  for (uint32_t k = 0; k < 96; ++k) rows[lane * 96 + k] = src[(blockIdx.x * 64 + lane) * 96 + k];
  __syncthreads();
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(rows)) + lane * 96;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t o = (shift + i * 5u + lane) % 60u;
    const Q4v v = lds_read16(base + o);
    const uint32_t w = lds_read4(base + o + 16);
    acc += v.x ^ v.y ^ v.z ^ v.w ^ w;
  }
  out[blockIdx.x * 64 + lane] = acc;
}
int main() {
  const uint32_t blocks = 4096, n = blocks * 64, iters = 256;
  std::vector<uint8_t> h((size_t)n * 96);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 7));
  uint8_t *d; uint32_t *o;
  hipMalloc(&d, h.size()); hipMalloc(&o, n * 4);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  std::vector<uint32_t> r(n);
  for (uint32_t shift : {0u, 1u, 2u, 3u}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<<<blocks, 64>>>(d, shift, iters, o);
    hipEventRecord(a);
    probe<<<blocks, 64>>>(d, shift, iters, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    uint32_t bad = 0;
    for (uint32_t l = 0; l < n; l += 97) {
      uint32_t acc = 0;
      for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t off = (shift + i * 5u + (l & 63)) % 60u;
        uint32_t w[5]; memcpy(w, h.data() + (size_t)l * 96 + off, 20);
        acc += w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4];
      }
      bad += acc != r[l];
    }
    printf("shift %u: %.3f ms, %u mismatches (%s)\n", shift, ms, bad, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
