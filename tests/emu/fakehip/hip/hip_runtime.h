// TEST INFRASTRUCTURE ONLY: a host-memory stand-in for the handful of HIP runtime calls csrc/api.cc makes, so that
// the product's own launch sequence (api.cc, unchanged) can run against the lock-step wave model of wave_emu.h in a
// container without a GPU.  "Device" memory is malloc'd host memory, streams and events are inert, a kernel launch
// (tests/emu/emu_launch.cc) runs the wavefronts of the grid one after another.  Nothing under sentencepiece_amd/
// includes this file; libspmx.so is compiled by hipcc against the real <hip/hip_runtime.h>.
#ifndef SPMX_FAKE_HIP_RUNTIME_H_
#define SPMX_FAKE_HIP_RUNTIME_H_
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct FakeHipStream *hipStream_t;
struct FakeHipEvent { double t_ms; };
typedef FakeHipEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum : unsigned { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostRegisterDefault = 0 };
struct hipDeviceProp_t { int multiProcessorCount; size_t totalGlobalMem; };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "fake hip error"; }
inline hipError_t hipGetDeviceCount(int *n) { *n = getenv("SPMX_EMU_NO_DEVICE") ? 0 : 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  const char *e = getenv("SPMX_EMU_CUS");
  p->multiProcessorCount = e ? atoi(e) : 2;
  p->totalGlobalMem = 8ull << 30;
  return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) {
  // 64 bytes of slack on either side, poisoned: the kernels' aligned 16-byte loads may over-read inside an allocation
  unsigned char *b = static_cast<unsigned char *>(malloc(n + 128));
  if (!b) return hipErrorOutOfMemory;
  memset(b, 0xCD, n + 128 < (256u << 20) ? n + 128 : (256u << 20));   // poison (the first 256 MiB of a huge block)
  *p = b + 64;
  return hipSuccess;
}
inline hipError_t hipFree(void *p) { if (p) free(static_cast<unsigned char *>(p) - 64); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new FakeHipEvent{0.0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new FakeHipEvent{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = static_cast<float>(b->t_ms - a->t_ms); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = 8ull << 30; *t = 8ull << 30; return hipSuccess; }
#endif
