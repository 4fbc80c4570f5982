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

// Streaming form (kernels_stream.h): workgroups of up to 16 wavefronts, each wave on its own tiles; one launch
// serves every length class.  RING 16: the score ring's size is a compile-time constant (models whose longest piece
// is <= 15 bytes); 0: taken from the arguments.
template <int RING, bool UDS>
__global__ __launch_bounds__(1024) void EncodeStreamKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<1, RING, UDS>(a, smem);
}
// no user-defined pieces, 16-bit back-pointer entries (BpShort): 16 wavefronts per CU at a ring of 16
template <int RING>
__global__ __launch_bounds__(1024) void EncodeStreamShortKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<1, RING, false, BpShort>(a, smem);
}
// every tile in the split form (kernels_matchfold.h): a launch of its own for the long length classes of a unigram model
__global__ __launch_bounds__(1024) void EncodeSplitKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<1, 0, false, BpWord, true>(a, smem);
}
__global__ __launch_bounds__(1024) void EncodeSplitShortKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<1, 0, false, BpShort, true>(a, smem);
}
__global__ __launch_bounds__(1024) void EncodeBpeStreamKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_stream_block<2, 0, false>(a, smem);
}
__global__ __launch_bounds__(1024) void EncodeWordKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_word_block<false, kWmPlain>(a, smem);
}
__global__ __launch_bounds__(1024) void EncodeWordCollectKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_word_block<false, kWmCollect>(a, smem);
}
__global__ __launch_bounds__(1024) void EncodeWordAgainKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_word_block<false, kWmDyn>(a, smem);
}
// word per lane (kernels_wordwave.h); H16: 16-bit ids in the arena slots (EncodeArgs::ids16)
template <bool H16>
__global__ __launch_bounds__(1024) void EncodeWordWaveKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_wordwave_block<kWmPlain, H16>(a, smem);
}
template <bool H16>
__global__ __launch_bounds__(1024) void EncodeWordWaveCollectKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_wordwave_block<kWmCollect, H16>(a, smem);
}
template <bool H16>
__global__ __launch_bounds__(1024) void EncodeWordWaveAgainKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_wordwave_block<kWmDyn, H16>(a, smem);
}
__global__ __launch_bounds__(512) void EncodeWordDpKernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  encode_word_block<true, kWmPlain>(a, smem);
}
__global__ __launch_bounds__(64) void WordResolveKernel(ResolveArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kResolveLdsBytesAll];
  word_resolve_block(a, smem);
}
__global__ __launch_bounds__(64) void BpeLongKernel(LongArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * kRawWinBytes];
  bpe_long_block(a, smem);
}

// (four wavefronts per SIMD: at most 128 vector registers -- 8192 documents are then two rounds of 4096 wavefronts)
template <uint32_t ML>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void UniLongKernel(LongArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uni_long_block<ML>(a, smem);
}

template <uint32_t ML>
__global__ __launch_bounds__(128) void UniLongPipeKernel(LongArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uni_long_pipe_block<ML>(a, smem);
}

__global__ __launch_bounds__(64) void NormalizeLongCountKernel(NormalizeArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * kRawWinBytes];
  norm_long_block<false>(a, smem);
}
__global__ __launch_bounds__(64) void NormalizeLongWriteKernel(NormalizeArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * kRawWinBytes];
  norm_long_block<true>(a, smem);
}
__global__ __launch_bounds__(64) void AlignLongKernel(AlignLongArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * kRawWinBytes];
  align_long_block(a, smem);
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
__global__ __launch_bounds__(64) void NBestKernel(NBestArgs a) { nbest_block<uint16_t>(a); }
__global__ __launch_bounds__(64) void NBestWideKernel(NBestArgs a) { nbest_block<uint32_t>(a); }
__global__ __launch_bounds__(64) void SplitCountKernel(SplitArgs a) { split_block<false>(a, nullptr); }
__global__ __launch_bounds__(64) void SplitWriteKernel(SplitArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[kSplitLdsBytes];
  split_block<true>(a, stage);
}
__global__ __launch_bounds__(64) void DecodeCountKernel(DecodeArgs a) { decode_block<false>(a); }
__global__ __launch_bounds__(64) void DecodeWriteKernel(DecodeArgs a) { decode_block<true>(a); }
__global__ __launch_bounds__(256) void PlainScanKernel(PlainScanArgs a) { plain_scan_block<false>(a); }
__global__ __launch_bounds__(256) void PlainScanKeepWsKernel(PlainScanArgs a) { plain_scan_block<true>(a); }
__global__ __launch_bounds__(64) void ClassifyCountKernel(ClassifyArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[kClassifyLdsWords];
  classify_block<0>(a, lds);
}
__global__ __launch_bounds__(64) void ClassifyScatterKernel(ClassifyArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[kClassifyLdsWords];
  classify_block<1>(a, lds);
}
__global__ __launch_bounds__(64) void ScanTilesKernel(ScanArgs a) { scan_tiles_block(a); }
__global__ __launch_bounds__(64) void ScanSumsKernel(ScanArgs a) { scan_sums_block(a); }
__global__ __launch_bounds__(64) void ScanFinalKernel(ScanArgs a) { scan_final_block(a); }
__global__ __launch_bounds__(64) void CompactKernel(CompactArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char compact_image[];
  compact_block(a, reinterpret_cast<uint16_t *>(compact_image));
}
__global__ __launch_bounds__(64) void CompactBigKernel(CompactArgs a) { compact_big_block(a); }
__global__ __launch_bounds__(64) void RebaseOffsetsKernel(RebaseArgs a) { rebase_block(a); }

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

hipError_t LaunchEncodeStream(int model_type, bool uds, const EncodeArgs &a, int grid, int waves,
                              uint32_t lds_bytes, hipStream_t stream) {
  EncodeFn fn = model_type == 2 ? EncodeBpeStreamKernel : a.bp_short ? (a.ring == 16 ? EncodeStreamShortKernel<16> : EncodeStreamShortKernel<0>)
                                : (a.ring == 16 ? (uds ? EncodeStreamKernel<16, true> : EncodeStreamKernel<16, false>)
                                                : (uds ? EncodeStreamKernel<0, true> : EncodeStreamKernel<0, false>));
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t LaunchEncodeSplit(const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t stream) {
  EncodeFn fn = a.bp_short ? EncodeSplitShortKernel : EncodeSplitKernel;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t LaunchEncodeWord(int mode, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t stream) {
  void (*fn)(EncodeArgs) = mode == 3 ? EncodeWordDpKernel : mode == 2 ? EncodeWordAgainKernel : mode == 1 ? EncodeWordCollectKernel : EncodeWordKernel;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), lds_bytes, stream, a);
  return hipGetLastError();
}
hipError_t LaunchEncodeWordWave(int mode, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t stream) {
  void (*fn)(EncodeArgs) = a.ids16 ? (mode == 2 ? EncodeWordWaveAgainKernel<true> : mode == 1 ? EncodeWordWaveCollectKernel<true> : EncodeWordWaveKernel<true>)
                                   : (mode == 2 ? EncodeWordWaveAgainKernel<false> : mode == 1 ? EncodeWordWaveCollectKernel<false> : EncodeWordWaveKernel<false>);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), lds_bytes, stream, a);
  return hipGetLastError();
}
hipError_t LaunchWordResolve(const ResolveArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(WordResolveKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchBpeLong(const LongArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(BpeLongKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchUniLong(const LongArgs &a, uint32_t cands, int grid, hipStream_t stream) {
  if (cands == 16u) hipLaunchKernelGGL(UniLongKernel<16>, dim3(grid), dim3(64), UniWaveLdsBytes(16), stream, a);
  else if (cands == 32u) hipLaunchKernelGGL(UniLongKernel<32>, dim3(grid), dim3(64), UniWaveLdsBytes(32), stream, a);
  else hipLaunchKernelGGL(UniLongKernel<64>, dim3(grid), dim3(64), UniWaveLdsBytes(64), stream, a);
  return hipGetLastError();
}

hipError_t LaunchNormalizeLong(bool write, const NormalizeArgs &a, int grid, hipStream_t stream) {
  if (write) hipLaunchKernelGGL(NormalizeLongWriteKernel, dim3(grid), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(NormalizeLongCountKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}
hipError_t LaunchAlignLong(const AlignLongArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(AlignLongKernel, dim3(grid), dim3(64), 0, stream, a);
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

hipError_t LaunchNBest(bool wide, const NBestArgs &a, int grid, hipStream_t stream) {
  if (wide) hipLaunchKernelGGL(NBestWideKernel, dim3(grid), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(NBestKernel, dim3(grid), dim3(64), 0, stream, a);
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

hipError_t LaunchPlainScan(const PlainScanArgs &a, int grid, hipStream_t stream) {
  if (a.keep_ws) hipLaunchKernelGGL(PlainScanKeepWsKernel, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(PlainScanKernel, dim3(grid), dim3(256), 0, stream, a);      // (four wavefronts per workgroup)
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

hipError_t LaunchRebase(const RebaseArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(RebaseOffsetsKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t LaunchCompact(const CompactArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(CompactKernel, dim3(grid), dim3(64), CompactLdsBytes(a.staged), stream, a);
  return hipGetLastError();
}
hipError_t LaunchUniLongPipe(const LongArgs &a, uint32_t cands, int grid, hipStream_t stream) {
  if (cands == 16u) hipLaunchKernelGGL(UniLongPipeKernel<16>, dim3(grid), dim3(128), UniPipeLdsBytes(16), stream, a);
  else if (cands == 32u) hipLaunchKernelGGL(UniLongPipeKernel<32>, dim3(grid), dim3(128), UniPipeLdsBytes(32), stream, a);
  else hipLaunchKernelGGL(UniLongPipeKernel<64>, dim3(grid), dim3(128), UniPipeLdsBytes(64), stream, a);
  return hipGetLastError();
}
hipError_t LaunchCompactBig(const CompactArgs &a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(CompactBigKernel, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace spmx
