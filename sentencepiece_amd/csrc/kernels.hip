// __global__ entry points for gfx950.  Grids are persistent (grid-stride over work lists
// whose lengths live in device memory, so no host round trip sits between the
// classify, encode, scan and compact launches).
#include "launch.h"

namespace spmx {

// CLS only gives every length class its own kernel symbol (rocprofv3 lists and
// times them separately); the capacities travel in EncodeArgs.
template <int MODEL, int CLS>
__global__ __launch_bounds__(64) void EncodeKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_block<MODEL>(a, smem);
}

// Streaming form (kernels_stream.h): workgroups of up to 16 wavefronts, each wave on its own tiles.
template <int CLS, bool FAST>
__global__ __launch_bounds__(FAST ? 1024 : 512) void EncodeStreamKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<FAST, 1>(a, smem);
}
// The FAST kernel specialized for a score ring of 16 entries (models whose longest piece is <= 15 bytes).
template <int CLS>
__global__ __launch_bounds__(1024) void EncodeStreamKernelR16(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<true, 1, 16>(a, smem);
}
template <int CLS, bool FAST>
__global__ __launch_bounds__(FAST ? 1024 : 512) void EncodeBpeStreamKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<FAST, 2>(a, smem);
}

__global__ __launch_bounds__(64) void AlignKernel(AlignArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  align_block(a, smem);
}
__global__ __launch_bounds__(64) void NormalizeCountKernel(NormalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  normalize_block<false>(a, smem);
}
__global__ __launch_bounds__(64) void NormalizeWriteKernel(NormalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  normalize_block<true>(a, smem);
}
__global__ __launch_bounds__(64) void NBestKernel(NBestArgs a) { nbest_block(a); }
__global__ __launch_bounds__(64) void SplitCountKernel(SplitArgs a) { split_block<false>(a, nullptr); }
__global__ __launch_bounds__(64) void SplitWriteKernel(SplitArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[kSplitLdsBytes];
  split_block<true>(a, stage);
}
__global__ __launch_bounds__(64) void DecodeCountKernel(DecodeArgs a) { decode_block<false>(a); }
__global__ __launch_bounds__(64) void DecodeWriteKernel(DecodeArgs a) { decode_block<true>(a); }
__global__ __launch_bounds__(64) void ClassifyCountKernel(ClassifyArgs a) {
  __shared__ uint32_t hist[3 * kSortKeys];
  classify_block<0>(a, hist);
}
__global__ __launch_bounds__(64) void ClassifyScatterKernel(ClassifyArgs a) {
  __shared__ uint32_t hist[3 * kSortKeys];
  classify_block<1>(a, hist);
}
__global__ __launch_bounds__(64) void ScanTilesKernel(ScanArgs a) { scan_tiles_block(a); }
__global__ __launch_bounds__(64) void ScanSumsKernel(ScanArgs a) { scan_sums_block(a); }
__global__ __launch_bounds__(64) void ScanFinalKernel(ScanArgs a) { scan_final_block(a); }
__global__ __launch_bounds__(64) void CompactKernel(CompactArgs a) { compact_block(a); }

namespace {
using EncodeFn = void (*)(EncodeArgs);

template <int MODEL>
EncodeFn PickEncode(int cls) {
  switch (cls) {
    case 0: return EncodeKernel<MODEL, 0>;
    case 1: return EncodeKernel<MODEL, 1>;
    case 2: return EncodeKernel<MODEL, 2>;
    case 3: return EncodeKernel<MODEL, 3>;
    default: return EncodeKernel<MODEL, 4>;
  }
}
}  // namespace

hipError_t LaunchEncode(int model_type, int cls, const EncodeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream) {
  (void)model_type;   // the sentence-per-wave form serves BPE only
  EncodeFn fn = PickEncode<2>(cls);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds_bytes, stream, a);
  return hipGetLastError();
}

namespace {
template <bool FAST>
EncodeFn PickStream(int cls) {
  switch (cls) {
    case 0: return EncodeStreamKernel<0, FAST>;
    case 1: return EncodeStreamKernel<1, FAST>;
    case 2: return EncodeStreamKernel<2, FAST>;
    case 3: return EncodeStreamKernel<3, FAST>;
    case 4: return EncodeStreamKernel<4, FAST>;
    case 5: return EncodeStreamKernel<5, FAST>;
    default: return EncodeStreamKernel<6, FAST>;
  }
}
EncodeFn PickStreamR16(int cls) {
  switch (cls) {
    case 0: return EncodeStreamKernelR16<0>;
    case 1: return EncodeStreamKernelR16<1>;
    case 2: return EncodeStreamKernelR16<2>;
    case 3: return EncodeStreamKernelR16<3>;
    case 4: return EncodeStreamKernelR16<4>;
    case 5: return EncodeStreamKernelR16<5>;
    default: return EncodeStreamKernelR16<6>;
  }
}
template <bool FAST>
EncodeFn PickBpeStream(int cls) {
  switch (cls) {
    case 0: return EncodeBpeStreamKernel<0, FAST>;
    case 1: return EncodeBpeStreamKernel<1, FAST>;
    case 2: return EncodeBpeStreamKernel<2, FAST>;
    case 3: return EncodeBpeStreamKernel<3, FAST>;
    case 4: return EncodeBpeStreamKernel<4, FAST>;
    default: return EncodeBpeStreamKernel<5, FAST>;
  }
}
}  // namespace

hipError_t LaunchEncodeStream(int model_type, int cls, bool fast, const EncodeArgs &a, int grid, int waves,
                              uint32_t lds_bytes, hipStream_t stream) {
  EncodeFn fn = model_type == 2 ? (fast ? PickBpeStream<true>(cls) : PickBpeStream<false>(cls))
                                : (fast ? (a.ring == 16 ? PickStreamR16(cls) : PickStream<true>(cls)) : PickStream<false>(cls));
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t LaunchAlign(const AlignArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream) {
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(AlignKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(AlignKernel, dim3(grid), dim3(64), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t LaunchNormalize(bool write, const NormalizeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream) {
  void (*fn)(NormalizeArgs) = write ? NormalizeWriteKernel : NormalizeCountKernel;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t LaunchNBest(const NBestArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(NBestKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchSplit(bool write, const SplitArgs &a, int grid, hipStream_t stream) {
  if (write) hipLaunchKernelGGL(SplitWriteKernel, dim3(grid), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(SplitCountKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchDecode(bool write, const DecodeArgs &a, int grid, hipStream_t stream) {
  if (write) hipLaunchKernelGGL(DecodeWriteKernel, dim3(grid), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(DecodeCountKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchClassify(const ClassifyArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(ClassifyCountKernel, dim3(grid), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(ClassifyScatterKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchScan(const ScanArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(ScanTilesKernel, dim3(grid), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(ScanSumsKernel, dim3(1), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(ScanFinalKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchCompact(const CompactArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(CompactKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace spmx
