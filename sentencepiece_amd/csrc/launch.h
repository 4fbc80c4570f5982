// Host-callable launchers of the kernels in kernels.hip.
#ifndef SPMX_LAUNCH_H_
#define SPMX_LAUNCH_H_
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace spmx {

struct LengthClass { uint32_t rcap, ncap; };
// Length classes: raw-byte capacity of a class (the classify kernels sort by it) and normalized-byte capacity.
// Streaming kernels (unigram; BPE that segments word by word): ncap is only the stride of a tile's HBM scratch -- a
// sentence whose normalized form overflows it goes to the call's overflow launch, which sizes its columns exactly.
// The classes up to kStreamMainMaxRaw share ONE launch; the document classes above run in a second one (few, long
// sentences: its own grid and slab).  Sentences longer than the last class go through the overflow launch as well.
// Sentence-per-wave BPE kernel (models that are not word-wise): rcap / ncap are its LDS workspace; only the classes up
// to kMaxStagedRaw run there, longer sentences -- and sentences whose normalized form overflows the last staged
// class -- take the long form (kernels_long.h).
constexpr int kNumClasses = 7;
constexpr uint32_t kMaxStagedRaw = 4096;       // sentence-per-wave BPE / align / Normalize kernels stage one sentence in LDS
constexpr uint32_t kStreamMainMaxRaw = 16384;  // streaming: classes up to this size share the main launch
constexpr LengthClass kClasses[kNumClasses] = {
    {192, 448}, {576, 1280}, {1536, 3328}, {4096, 8704}, {16384, 32768}, {65536, 98304}, {1048576, 1572864}};
// BPE, sentence-per-wave: the normalized capacity of the last staged class is what fits the LDS of a CU
constexpr uint32_t kBpeWaveNcap3 = 6400;

// score ring entries for a model whose longest piece has max_piece_len bytes: one more than that -- 16 (the
// compile-time specialization) when that is enough, else exactly what the model needs (the ring is what bounds the
// wavefronts per CU: 8 bytes per entry per lane of LDS)
inline uint32_t ScoreRing(int max_piece_len) {
  const uint32_t need = static_cast<uint32_t>(max_piece_len) + 1u;
  return need <= 16u ? 16u : ((need + 1u) & ~1u);
}

// One streaming launch (kernels_stream.h): model_type 1 unigram / 2 BPE; uds: the model has USER_DEFINED pieces
hipError_t LaunchEncodeStream(int model_type, bool uds, const EncodeArgs &a, int grid, int waves,
                              uint32_t lds_bytes, hipStream_t stream);
// the same launch with EVERY tile in the split form (kernels_matchfold.h): unigram, no user-defined pieces
hipError_t LaunchEncodeSplit(const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t stream);
// The word kernel (kernels_word.h): unigram models with kNfUniWordwise
// mode: 0 plain first pass, 1 collecting first pass, 2 second round over the call-local memo, 3 the DP pass
hipError_t LaunchEncodeWord(int mode, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t stream);
// the word-per-lane form of the same rounds (kernels_wordwave.h): mode 0 plain, 1 collecting, 2 second round
hipError_t LaunchEncodeWordWave(int mode, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchWordResolve(const ResolveArgs &a, int grid, hipStream_t stream);
hipError_t LaunchBpeLong(const LongArgs &a, int grid, hipStream_t stream);
// The wave-cooperative unigram form (kernels_uniwave.h): one sentence per 64-thread workgroup; cands = entries of a matrix row (UniWaveRow: the longest piece in bytes)
hipError_t LaunchUniLong(const LongArgs &a, uint32_t cands, int grid, hipStream_t stream);
// ... a document per workgroup of two wavefronts, a walker and a folder (kernels_uniwave.h uni_long_pipe_block): few, long documents
hipError_t LaunchUniLongPipe(const LongArgs &a, uint32_t cands, int grid, hipStream_t stream);
hipError_t LaunchNormalizeLong(bool write, const NormalizeArgs &a, int grid, hipStream_t stream);
hipError_t LaunchAlignLong(const AlignLongArgs &a, int grid, hipStream_t stream);
hipError_t LaunchEncode(int model_type, int cls, const EncodeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchAlign(const AlignArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchNormalize(bool write, const NormalizeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchNBest(bool wide, const NBestArgs &a, int grid, hipStream_t stream);
hipError_t LaunchSplit(bool write, const SplitArgs &a, int grid, hipStream_t stream);
hipError_t LaunchDecode(bool write, const DecodeArgs &a, int grid, hipStream_t stream);
hipError_t LaunchPlainScan(const PlainScanArgs &a, int grid, hipStream_t stream);
hipError_t LaunchClassify(const ClassifyArgs &a, int grid, hipStream_t stream);
hipError_t LaunchScan(const ScanArgs &a, int grid, hipStream_t stream);
hipError_t LaunchCompact(const CompactArgs &a, int grid, hipStream_t stream);
hipError_t LaunchCompactBig(const CompactArgs &a, int grid, hipStream_t stream);   // the document blocks LaunchCompact listed
hipError_t LaunchRebase(const RebaseArgs &a, int grid, hipStream_t stream);   // kernels_gather.h

}  // namespace spmx
#endif
