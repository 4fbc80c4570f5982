// Probe: do unaligned 16-byte global loads work on gfx950 (SH_MEM_CONFIG alignment mode), and what do they cost
// against aligned ones?  Each lane reads 16 B at base + lane * stride + shift.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u1u __attribute__((aligned(1)));
__global__ void probe(const uint8_t *base, uint32_t stride, uint32_t shift, uint32_t iters, uint32_t *out) {
  const uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  const uint8_t *p = base + (uint64_t)lane * stride + shift;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < iters; ++i) {
    const u4u v = *reinterpret_cast<const u4u *>(p + (i & 7) * 16);
    const uint32_t w = *reinterpret_cast<const u1u *>(p + (i & 7) * 16 + 16);
    acc += v.x ^ v.y ^ v.z ^ v.w ^ w;
  }
  out[lane] = acc;
}
int main() {
  const uint32_t n = 256 * 1024, stride = 160;
  std::vector<uint8_t> h((size_t)n * stride + 4096);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8));
  uint8_t *d; uint32_t *o;
  hipMalloc(&d, h.size()); hipMalloc(&o, n * 4);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  std::vector<uint32_t> r(n);
  for (uint32_t shift : {0u, 1u, 3u, 5u, 13u}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<<<n / 256, 256>>>(d, stride, shift, 64, o);
    hipEventRecord(a);
    probe<<<n / 256, 256>>>(d, stride, shift, 64, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    uint32_t bad = 0;
    for (uint32_t l = 0; l < n; l += 997) {
      uint32_t acc = 0;
      const uint8_t *p = h.data() + (size_t)l * stride + shift;
      for (uint32_t i = 0; i < 64; ++i) { uint32_t w[5]; memcpy(w, p + (i & 7) * 16, 20); acc += w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4]; }
      bad += acc != r[l];
    }
    printf("shift %2u: %.3f ms, %u mismatches, err=%s\n", shift, ms, bad, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
