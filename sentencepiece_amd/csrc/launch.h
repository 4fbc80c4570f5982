// Host-callable launchers of the kernels in kernels.hip.
#ifndef SPMX_LAUNCH_H_
#define SPMX_LAUNCH_H_
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace spmx {

struct LengthClass { uint32_t rcap, ncap; };
// Length classes per model type: raw-byte capacity and normalized-byte capacity
// (the stride of the streaming kernels' HBM scratch; the LDS workspace of the
// sentence-per-wave BPE kernel).  A sentence whose normalized form overflows
// ncap is handed to the next class; past the last class it is an error.
// Unigram: the last two classes are for document-length inputs (up to 1 MiB per sentence).  Only the FAST
// kernel runs there (per-lane normalizers; its working set does not depend on the length), so they need a model
// it can take (kernels_stream.h StreamFastEligible); ncap is the capacity of a text column there.
constexpr int kNumClassesUnigram = 7;
constexpr int kNumClassesBpe = 6;
constexpr uint32_t kMaxStagedRaw = 8192;   // GENERAL kernels stage one sentence in LDS: classes up to this raw size
constexpr LengthClass kClassesUnigram[kNumClassesUnigram] = {
    {192, 448}, {576, 1280}, {1536, 3328}, {4096, 8704}, {8192, 20480}, {65536, 98304}, {1048576, 1572864}};
// BPE: the last two classes are document-length as well (word-wise models the FAST kernel can take: the lane form's
// working set is one word; a word that outgrows the LDS slots is merged in HBM, kernels_bpe_stream.h).
constexpr LengthClass kClassesBpe[kNumClassesBpe] = {{192, 448}, {576, 1280}, {1536, 3328}, {4096, 6400},
                                                     {65536, 98304}, {1048576, 1572864}};

// score ring entries for a model whose longest piece has max_piece_len bytes
inline uint32_t ScoreRing(int max_piece_len) {
  uint32_t r = 16;
  while (r < static_cast<uint32_t>(max_piece_len) + 1) r <<= 1;
  return r;
}

hipError_t LaunchEncodeStream(int model_type, int cls, bool fast, const EncodeArgs &a, int grid, int waves,
                              uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchEncode(int model_type, int cls, const EncodeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchAlign(const AlignArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchNormalize(bool write, const NormalizeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchNBest(const NBestArgs &a, int grid, hipStream_t stream);
hipError_t LaunchSplit(bool write, const SplitArgs &a, int grid, hipStream_t stream);
hipError_t LaunchDecode(bool write, const DecodeArgs &a, int grid, hipStream_t stream);
hipError_t LaunchClassify(const ClassifyArgs &a, int grid, hipStream_t stream);
hipError_t LaunchScan(const ScanArgs &a, int grid, hipStream_t stream);
hipError_t LaunchCompact(const CompactArgs &a, int grid, hipStream_t stream);

}  // namespace spmx
#endif
