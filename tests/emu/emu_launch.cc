// TEST INFRASTRUCTURE ONLY (see wave_emu.h, fakehip/hip/hip_runtime.h): csrc/launch.h implemented on the lock-step
// wave model, standing in for csrc/kernels.hip so that csrc/api.cc -- the product's launch sequence, unchanged --
// runs on the CPU.  A launch runs the wavefronts of the grid one after another (workgroup by workgroup, wave by wave);
// the waves of a workgroup share one LDS block, as on the device.
#include "wave_emu.h"

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../sentencepiece_amd/csrc/launch.h"

namespace spmx {
namespace {
template <typename F>
void RunGrid(int grid, int waves, uint32_t lds_bytes, F body) {
  std::vector<unsigned char> raw(lds_bytes + 128);
  unsigned char *smem = raw.data() + ((64 - (reinterpret_cast<uintptr_t>(raw.data()) & 63)) & 63);
  for (int b = 0; b < grid; ++b) {
    memset(smem, 0xCD, lds_bytes + 32);
    for (int w = 0; w < waves; ++w) {
      emu::g_wave.wib = w;
      emu::g_wave.wpb = waves;
      emu::RunWave(b, grid, smem, [&] { body(smem); });
    }
  }
  emu::g_wave.wib = 0;
  emu::g_wave.wpb = 1;
}
// Workgroups of TWO wavefronts that run side by side and meet at wv::block_sync() (the documents' walker / folder pair):
// the second wavefront on a helper thread that lives as long as the process (a wavefront's 64 lane stacks belong to its
// thread), a two-party barrier between them.
class Together {
 public:
  void Arrive() {
    std::unique_lock<std::mutex> l(mu_);
    const uint64_t gen = gen_;
    if (++waiting_ == 2) { waiting_ = 0; ++gen_; cv_.notify_all(); }
    else cv_.wait(l, [&] { return gen_ != gen; });
  }
  void RunOnHelper(std::function<void()> f) {
    std::unique_lock<std::mutex> l(mu_);
    if (!started_) { started_ = true; std::thread([this] { Loop(); }).detach(); }
    task_ = std::move(f); has_task_ = true; task_done_ = false;
    cv_.notify_all();
  }
  void WaitHelper() {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return task_done_; });
  }
 private:
  void Loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return has_task_; });
        f = std::move(task_); has_task_ = false;
      }
      f();
      { std::unique_lock<std::mutex> l(mu_); task_done_ = true; cv_.notify_all(); }
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  int waiting_ = 0;
  uint64_t gen_ = 0;
  bool started_ = false, has_task_ = false, task_done_ = true;
  std::function<void()> task_;
};
Together &g_together = *new Together;              // (never destroyed: its helper thread waits on it until the process ends)
std::mutex &g_together_user = *new std::mutex;   // (one launch of this kind at a time: the pipelined host form has several workers)

template <typename F>
void RunGridTogether(int grid, uint32_t lds_bytes, F body) {
  std::lock_guard<std::mutex> user(g_together_user);
  std::vector<unsigned char> raw(lds_bytes + 128);
  unsigned char *smem = raw.data() + ((64 - (reinterpret_cast<uintptr_t>(raw.data()) & 63)) & 63);
  for (int b = 0; b < grid; ++b) {
    memset(smem, 0xCD, lds_bytes + 32);
    g_together.RunOnHelper([&, b] {
      emu::g_wave.wib = 1; emu::g_wave.wpb = 2;
      emu::g_wave.block_barrier = [] { g_together.Arrive(); };
      emu::RunWave(b, grid, smem, [&] { body(smem); });
    });
    emu::g_wave.wib = 0; emu::g_wave.wpb = 2;
    emu::g_wave.block_barrier = [] { g_together.Arrive(); };
    emu::RunWave(b, grid, smem, [&] { body(smem); });
    g_together.WaitHelper();
  }
  emu::g_wave.wib = 0; emu::g_wave.wpb = 1;
  emu::g_wave.block_barrier = nullptr;
}
}  // namespace

hipError_t LaunchEncodeStream(int model_type, bool uds, const EncodeArgs &a, int grid, int waves,
                              uint32_t lds_bytes, hipStream_t) {
  if (model_type == 2) {
    RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<2, 0, false>(a, s); });
  } else if (a.bp_short) {
    if (a.ring == 16) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 16, false, BpShort>(a, s); });
    else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 0, false, BpShort>(a, s); });
  } else if (a.ring == 16) {
    if (uds) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 16, true>(a, s); });
    else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 16, false>(a, s); });
  } else {
    if (uds) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 0, true>(a, s); });
    else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 0, false>(a, s); });
  }
  return hipSuccess;
}
hipError_t LaunchEncodeSplit(const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t) {
  if (a.bp_short) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 0, false, BpShort, true>(a, s); });
  else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_stream_block<1, 0, false, BpWord, true>(a, s); });
  return hipSuccess;
}
hipError_t LaunchEncodeWord(int mode, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t) {
  if (mode == 3) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_word_block<true, kWmPlain>(a, s); });
  else if (mode == 2) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_word_block<false, kWmDyn>(a, s); });
  else if (mode == 1) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_word_block<false, kWmCollect>(a, s); });
  else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_word_block<false, kWmPlain>(a, s); });
  return hipSuccess;
}
hipError_t LaunchEncodeWordWave(int mode, const EncodeArgs &a, int grid, int waves, uint32_t lds_bytes, hipStream_t) {
  if (a.ids16) {
    if (mode == 2) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_wordwave_block<kWmDyn, true>(a, s); });
    else if (mode == 1) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_wordwave_block<kWmCollect, true>(a, s); });
    else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_wordwave_block<kWmPlain, true>(a, s); });
  } else {
    if (mode == 2) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_wordwave_block<kWmDyn, false>(a, s); });
    else if (mode == 1) RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_wordwave_block<kWmCollect, false>(a, s); });
    else RunGrid(grid, waves, lds_bytes, [&](unsigned char *s) { encode_wordwave_block<kWmPlain, false>(a, s); });
  }
  return hipSuccess;
}
hipError_t LaunchWordResolve(const ResolveArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, ResolveLdsBytesAll(), [&](unsigned char *s) { word_resolve_block(a, s); });
  return hipSuccess;
}
hipError_t LaunchBpeLong(const LongArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, 64 * kRawWinBytes, [&](unsigned char *s) { bpe_long_block(a, s); });
  return hipSuccess;
}

hipError_t LaunchUniLong(const LongArgs &a, uint32_t cands, int grid, hipStream_t) {
  if (cands == 16u) RunGrid(grid, 1, UniWaveLdsBytes(16), [&](unsigned char *s) { uni_long_block<16>(a, s); });
  else if (cands == 32u) RunGrid(grid, 1, UniWaveLdsBytes(32), [&](unsigned char *s) { uni_long_block<32>(a, s); });
  else RunGrid(grid, 1, UniWaveLdsBytes(64), [&](unsigned char *s) { uni_long_block<64>(a, s); });
  return hipSuccess;
}

hipError_t LaunchUniLongPipe(const LongArgs &a, uint32_t cands, int grid, hipStream_t) {
  if (cands == 16u) RunGridTogether(grid, UniPipeLdsBytes(16), [&](unsigned char *s) { uni_long_pipe_block<16>(a, s); });
  else if (cands == 32u) RunGridTogether(grid, UniPipeLdsBytes(32), [&](unsigned char *s) { uni_long_pipe_block<32>(a, s); });
  else RunGridTogether(grid, UniPipeLdsBytes(64), [&](unsigned char *s) { uni_long_pipe_block<64>(a, s); });
  return hipSuccess;
}

hipError_t LaunchNormalizeLong(bool write, const NormalizeArgs &a, int grid, hipStream_t) {
  if (write) RunGrid(grid, 1, 64 * kRawWinBytes, [&](unsigned char *s) { norm_long_block<true>(a, s); });
  else RunGrid(grid, 1, 64 * kRawWinBytes, [&](unsigned char *s) { norm_long_block<false>(a, s); });
  return hipSuccess;
}
hipError_t LaunchAlignLong(const AlignLongArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, 64 * kRawWinBytes, [&](unsigned char *s) { align_long_block(a, s); });
  return hipSuccess;
}
hipError_t LaunchEncode(int, int, const EncodeArgs &a, int grid, uint32_t lds_bytes, hipStream_t) {
  RunGrid(grid, 1, lds_bytes, [&](unsigned char *s) { encode_block<2>(a, s); });
  return hipSuccess;
}
hipError_t LaunchAlign(const AlignArgs &a, int grid, uint32_t lds_bytes, hipStream_t) {
  RunGrid(grid, 1, lds_bytes, [&](unsigned char *s) { align_block(a, s); });
  return hipSuccess;
}
hipError_t LaunchNormalize(bool write, const NormalizeArgs &a, int grid, uint32_t lds_bytes, hipStream_t) {
  if (write) RunGrid(grid, 1, lds_bytes, [&](unsigned char *s) { normalize_block<true>(a, s); });
  else RunGrid(grid, 1, lds_bytes, [&](unsigned char *s) { normalize_block<false>(a, s); });
  return hipSuccess;
}
hipError_t LaunchNBest(bool wide, const NBestArgs &a, int grid, hipStream_t) {
  if (wide) RunGrid(grid, 1, 0, [&](unsigned char *) { nbest_block<uint32_t>(a); });
  else RunGrid(grid, 1, 0, [&](unsigned char *) { nbest_block<uint16_t>(a); });
  return hipSuccess;
}
hipError_t LaunchSplit(bool write, const SplitArgs &a, int grid, hipStream_t) {
  if (write) RunGrid(grid, 1, kSplitLdsBytes, [&](unsigned char *s) { split_block<true>(a, s); });
  else RunGrid(grid, 1, 0, [&](unsigned char *) { split_block<false>(a, nullptr); });
  return hipSuccess;
}
hipError_t LaunchDecode(bool write, const DecodeArgs &a, int grid, hipStream_t) {
  if (write) RunGrid(grid, 1, 0, [&](unsigned char *) { decode_block<true>(a); });
  else RunGrid(grid, 1, 0, [&](unsigned char *) { decode_block<false>(a); });
  return hipSuccess;
}
hipError_t LaunchPlainScan(const PlainScanArgs &a, int grid, hipStream_t) {
  if (a.keep_ws) RunGrid(grid, 4, 0, [&](unsigned char *) { plain_scan_block<true>(a); });
  else RunGrid(grid, 4, 0, [&](unsigned char *) { plain_scan_block<false>(a); });
  return hipSuccess;
}
hipError_t LaunchClassify(const ClassifyArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, kClassifyLdsWords * 4, [&](unsigned char *s) { classify_block<0>(a, reinterpret_cast<uint32_t *>(s)); });
  RunGrid(grid, 1, kClassifyLdsWords * 4, [&](unsigned char *s) { classify_block<1>(a, reinterpret_cast<uint32_t *>(s)); });
  return hipSuccess;
}
hipError_t LaunchScan(const ScanArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, 0, [&](unsigned char *) { scan_tiles_block(a); });
  RunGrid(1, 1, 0, [&](unsigned char *) { scan_sums_block(a); });
  RunGrid(grid, 1, 0, [&](unsigned char *) { scan_final_block(a); });
  return hipSuccess;
}
hipError_t LaunchCompact(const CompactArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, CompactLdsBytes(a.staged), [&](unsigned char *s) { compact_block(a, reinterpret_cast<uint16_t *>(s)); });
  return hipSuccess;
}
hipError_t LaunchCompactBig(const CompactArgs &a, int grid, hipStream_t) {
  RunGrid(grid > 3 ? 3 : grid, 1, 0, [&](unsigned char *) { compact_big_block(a); });
  return hipSuccess;
}
hipError_t LaunchRebase(const RebaseArgs &a, int grid, hipStream_t) {
  RunGrid(grid, 1, 0, [&](unsigned char *) { rebase_block(a); });
  return hipSuccess;
}
}  // namespace spmx
