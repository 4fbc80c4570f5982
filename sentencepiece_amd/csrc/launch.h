// Host-callable launchers of the kernels in kernels.hip.
#ifndef SPMX_LAUNCH_H_
#define SPMX_LAUNCH_H_
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace spmx {

struct LengthClass { uint32_t rcap, ncap; };
// Length classes per model type: raw-byte capacity and normalized-byte capacity
// of the per-wave LDS workspace.  A sentence whose normalized form overflows
// ncap is handed to the next class; past the last class it is an error.
constexpr int kNumClassesUnigram = 5;
constexpr int kNumClassesBpe = 4;
constexpr LengthClass kClassesUnigram[kNumClassesUnigram] = {
    {192, 448}, {576, 1280}, {1536, 3328}, {4096, 8704}, {8192, 20480}};
constexpr LengthClass kClassesBpe[kNumClassesBpe] = {{192, 448}, {576, 1280}, {1536, 3328}, {4096, 6400}};

// Unigram classes that run in the tile form (kernels_tile.h): a wave takes 64
// sentences, one per lane.  area = LDS bytes for the round's text + back-pointer
// bytes (>= 2 * ncap + 1); the per-wave classes above remain for longer sentences
// and for BPE.
struct TileClass { uint32_t area, fast_area; };   // GENERAL kernel, FAST kernel (slots of 2 (L + 1) + 1 bytes)
constexpr int kNumTileClasses = 2;
constexpr TileClass kTileClasses[kNumTileClasses] = {{24 * 1024, 15616}, {40 * 1024, 27 * 1024}};
// score ring entries for a model whose longest piece has max_piece_len bytes
inline uint32_t TileRing(int max_piece_len) {
  uint32_t r = 16;
  while (r < static_cast<uint32_t>(max_piece_len) + 1) r <<= 1;
  return r;
}

hipError_t LaunchEncodeTile(int cls, bool fast, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes,
                            hipStream_t stream);
hipError_t LaunchEncodeStream(int model_type, int cls, bool fast, const EncodeArgs &a, int grid, int waves,
                              uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchEncode(int model_type, int cls, const EncodeArgs &a, int grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t LaunchClassify(const ClassifyArgs &a, int grid, hipStream_t stream);
hipError_t LaunchScan(const ScanArgs &a, int grid, hipStream_t stream);
hipError_t LaunchCompact(const CompactArgs &a, int grid, hipStream_t stream);

}  // namespace spmx
#endif
