// libspmx.so: the C ABI of include/spmx.h over the HIP kernels of kernels.hip.
//
// One handle = one loaded model on one GPU: the compiled tables live in HBM for
// the life of the handle, a grow-only workspace (class lists, id arena, scan
// scratch) is reused from call to call, and one encode call is a fixed
// sequence of launches on the caller's stream with a single small read-back at
// the end:
//
//   memset ctrl -> classify -> encode[class 0..k) -> scan x3 -> compact -> D2H {ctrl, total}
//
// There is no CPU path: without a usable HIP device spmx_create fails with
// UNAVAILABLE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/spmx.h"
#include "launch.h"
#include "model.h"
#include "tables.h"

using namespace spmx;

namespace {

constexpr uint32_t kLdsPerCu = 160u * 1024u;   // gfx950 (MI355X_MICROARCH.md)

std::mutex g_err_mu;
std::string g_create_error;

// ctrl block layout (device + pinned host mirror), zeroed before every call
struct Ctrl {
  uint32_t list_counts[kMaxClasses];
  uint32_t hard_counts[kMaxClasses];   // sentences a FAST kernel left to the GENERAL kernel of its class
  uint32_t wave_counts[kMaxClasses];
  uint32_t key_totals[kSortKeys], key_cursor[kSortKeys];   // classify: counting sort by (class, length sub-bucket)   // BPE: sentences the streaming kernels left to the sentence-per-wave kernel
  uint32_t status;
  uint32_t pad;
  uint32_t tile_cursor[2 * kMaxClasses];   // per kernel slot: the streaming kernels' tile queue
  unsigned long long arena_head;
  unsigned long long stats[kStatsPerClass * 2 * kMaxClasses];   // per kernel slot (see Profile)
  unsigned long long bad_key;   // decode: min over offending (sentence << 32 | id)
  uint64_t total_ids;   // copied from id_offs[n] by the final D2H
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;   // elements
  hipError_t Reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    const size_t want = n + n / 4 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void Free() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// One entry per kernel slot: slot c < kMaxClasses is length class c's encode kernel (the FAST kernel where one
// runs); slot kSlotGeneral + c is the GENERAL kernel that follows a FAST kernel of class c.
constexpr int kSlotGeneral = kMaxClasses;
constexpr int kMaxSlots = 2 * kMaxClasses;
struct Profile {
  int n = 0;
  char name[kMaxSlots][40] = {{0}};
  float kernel_ms[kMaxSlots] = {0};
  uint64_t sentences[kMaxSlots] = {0}, raw_bytes[kMaxSlots] = {0}, ids[kMaxSlots] = {0};
  uint32_t rcap[kMaxSlots] = {0};
  uint64_t cycles[kMaxSlots][5] = {{0}};
  float total_ms = 0.f;
};

}  // namespace

struct spmx_handle {
  std::mutex mu;
  std::string error;
  ModelData model;
  HostTables tables;
  std::string extra_options;
  int device = 0;
  int n_cu = 256;
  bool no_fast = false;   // SPMX_NO_FAST=1: GENERAL kernels only (A/B measurements)
  int tile_waves_override = 0;   // SPMX_TILE_WAVES: cap on wavefronts per workgroup of the streaming kernels
  // device copies of the tables
  DevBuf<uint32_t> d_ndarts, d_npair, d_sym_final, d_dec_info, d_dec_off;
  DevBuf<uint8_t> d_dec_bytes;
  DevBuf<uint8_t> d_nblob;
  DevBuf<U4> d_ptrie, d_chartab, d_pairtab;
  DevBuf<U2> d_utrie;
  DevBuf<uint16_t> d_sym_len;
  DevBuf<int32_t> d_byte_ids;
  SpmxDev dev{};   // scalars + device pointers
  // workspace
  DevBuf<uint32_t> d_lists, d_counts;
  DevBuf<uint64_t> d_tmp_off, d_tile_sums, d_chunk_base;
  DevBuf<int32_t> d_arena_tb, d_tok_begin;      // spans form
  DevBuf<uint32_t> d_span_begin, d_span_end, d_nspan_begin, d_nspan_end;
  DevBuf<uint8_t> d_norm, d_bpe_long, d_nbest_scratch;
  DevBuf<unsigned long long> d_res_off;
  DevBuf<float> d_res_score;
  DevBuf<int32_t> d_arena;
  Ctrl *d_ctrl = nullptr;
  Ctrl *h_ctrl = nullptr;   // pinned
  // host-form staging
  DevBuf<uint8_t> d_text;
  DevBuf<uint64_t> d_offs, d_id_offs;
  DevBuf<int32_t> d_ids;
  // profiling
  bool profiling = false;
  hipEvent_t ev[kMaxSlots + 1][2] = {};   // per kernel slot (see Profile) + the whole call
  char slot_name[kMaxSlots][40] = {{0}};
  bool slot_used[kMaxSlots] = {false};
  bool no_lane_general = false;      // SPMX_NO_LANE_GENERAL=1: FAST kernels hand every non-ASCII sentence to GENERAL
  uint32_t lane_general_max_raw = kLaneGeneralMaxRaw, lane_general_min_lanes = 0;   // SPMX_LANE_GENERAL_MAX_RAW / _MIN_LANES (0: per class)
  bool tiles_ascending = false;      // SPMX_TILE_ORDER=asc
  uint32_t sub_buckets = kSubBuckets;   // SPMX_SUB_BUCKETS: length sub-buckets per class in the classify sort (1..64)
  bool static_tiles = false;         // SPMX_STATIC_TILES=1: fixed-stride tiles in the streaming kernels (A/B measurements)
  bool no_merge_general = false;     // SPMX_NO_MERGE_GENERAL=1: a GENERAL launch per class (A/B measurements)
  bool no_stream = false;            // SPMX_NO_STREAM=1: BPE in the sentence-per-wave form only (A/B measurements)
  uint32_t ring_override = 0;        // SPMX_FORCE_RING: score-ring entries (A/B measurements; must exceed the longest piece)
  uint64_t stream_scratch_limit = 4ull << 30;   // SPMX_STREAM_SCRATCH_MB: cap on the streaming kernels' HBM scratch
  DevBuf<uint32_t> d_stream;         // scratch of the streaming kernels (text columns + back-pointer words)
  bool ev_ready = false;
  Profile prof;
};

namespace {

int Fail(spmx_handle *h, int code, const std::string &msg) {
  if (h) h->error = msg;
  else { std::lock_guard<std::mutex> l(g_err_mu); g_create_error = msg; }
  return code;
}
int FailHip(spmx_handle *h, hipError_t e, const char *what) {
  return Fail(h, kInternal, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_OR_RETURN(h, expr)                                   \
  do {                                                           \
    hipError_t e_ = (expr);                                      \
    if (e_ != hipSuccess) return FailHip((h), e_, #expr);        \
  } while (0)

template <typename T>
hipError_t Upload(DevBuf<T> *b, const std::vector<T> &v) {
  hipError_t e = b->Reserve(v.size() ? v.size() : 1);
  if (e != hipSuccess) return e;
  if (v.empty()) return hipSuccess;
  return hipMemcpy(b->p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

// Uploads every table and binds h->dev to the device copies.
int UploadTables(spmx_handle *h) {
  HostTables &t = h->tables;
  HIP_OR_RETURN(h, Upload(&h->d_ndarts, t.ndarts));
  HIP_OR_RETURN(h, Upload(&h->d_nblob, t.nblob));
  HIP_OR_RETURN(h, Upload(&h->d_npair, t.npair));
  HIP_OR_RETURN(h, Upload(&h->d_ptrie, t.ptrie));
  HIP_OR_RETURN(h, Upload(&h->d_utrie, t.utrie));
  HIP_OR_RETURN(h, Upload(&h->d_chartab, t.chartab));
  HIP_OR_RETURN(h, Upload(&h->d_pairtab, t.pairtab));
  HIP_OR_RETURN(h, Upload(&h->d_sym_final, t.sym_final));
  HIP_OR_RETURN(h, Upload(&h->d_sym_len, t.sym_len));
  HIP_OR_RETURN(h, Upload(&h->d_byte_ids, t.byte_ids));
  HIP_OR_RETURN(h, Upload(&h->d_dec_info, t.dec_info));
  HIP_OR_RETURN(h, Upload(&h->d_dec_off, t.dec_off));
  HIP_OR_RETURN(h, Upload(&h->d_dec_bytes, t.dec_bytes));
  h->dev = t.scalars;
  h->dev.ndarts = h->d_ndarts.p;
  h->dev.nblob = h->d_nblob.p;
  h->dev.npair = h->d_npair.p;
  h->dev.ptrie = h->d_ptrie.p;
  h->dev.utrie = h->d_utrie.p;
  h->dev.chartab = h->d_chartab.p;
  h->dev.pairtab = h->d_pairtab.p;
  h->dev.sym_final = h->d_sym_final.p;
  h->dev.sym_len = h->d_sym_len.p;
  h->dev.byte_ids = h->d_byte_ids.p;
  h->dev.dec_info = h->d_dec_info.p;
  h->dev.dec_off = h->d_dec_off.p;
  h->dev.dec_bytes = h->d_dec_bytes.p;
  return kOk;
}

// After SetVocabulary / ResetVocabulary / SetEncodeExtraOptions: only the
// type-dependent words and the scalars change.
int RefreshDevice(spmx_handle *h, bool types_changed) {
  HostTables &t = h->tables;
  if (types_changed) {
    if (h->model.model_type == kUnigram) HIP_OR_RETURN(h, Upload(&h->d_ptrie, t.ptrie));
    else HIP_OR_RETURN(h, Upload(&h->d_sym_final, t.sym_final));
  }
  SpmxDev d = t.scalars;
  d.ndarts = h->dev.ndarts; d.nblob = h->dev.nblob; d.npair = h->dev.npair; d.ptrie = h->d_ptrie.p; d.utrie = h->dev.utrie;
  d.chartab = h->dev.chartab; d.pairtab = h->dev.pairtab; d.sym_final = h->d_sym_final.p;
  d.sym_len = h->dev.sym_len; d.byte_ids = h->dev.byte_ids;
  d.dec_info = h->dev.dec_info; d.dec_off = h->dev.dec_off; d.dec_bytes = h->dev.dec_bytes;
  h->dev = d;
  return kOk;
}

void DestroyHandle(spmx_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  h->d_ndarts.Free(); h->d_npair.Free(); h->d_sym_final.Free(); h->d_nblob.Free(); h->d_ptrie.Free(); h->d_chartab.Free();
  h->d_pairtab.Free(); h->d_utrie.Free(); h->d_sym_len.Free(); h->d_byte_ids.Free();
  h->d_dec_info.Free(); h->d_dec_off.Free(); h->d_dec_bytes.Free();
  h->d_stream.Free(); h->d_lists.Free(); h->d_counts.Free(); h->d_tmp_off.Free(); h->d_tile_sums.Free(); h->d_chunk_base.Free(); h->d_arena_tb.Free(); h->d_tok_begin.Free(); h->d_span_begin.Free(); h->d_span_end.Free(); h->d_nspan_begin.Free(); h->d_nspan_end.Free(); h->d_norm.Free(); h->d_bpe_long.Free(); h->d_nbest_scratch.Free(); h->d_res_off.Free(); h->d_res_score.Free(); h->d_arena.Free();
  h->d_text.Free(); h->d_offs.Free(); h->d_id_offs.Free(); h->d_ids.Free();
  if (h->d_ctrl) (void)hipFree(h->d_ctrl);
  if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
  if (h->ev_ready)
    for (auto &pair : h->ev) { (void)hipEventDestroy(pair[0]); (void)hipEventDestroy(pair[1]); }
  delete h;
}

int NumClasses(const spmx_handle *h) { return h->model.model_type == kBpe ? kNumClassesBpe : kNumClassesUnigram; }
const LengthClass *Classes(const spmx_handle *h) { return h->model.model_type == kBpe ? kClassesBpe : kClassesUnigram; }

// score-ring entries of the streaming unigram kernels for this handle's model
uint32_t HandleRing(const spmx_handle *h) {
  const uint32_t r = ScoreRing(h->tables.max_piece_len);
  return h->ring_override > r ? h->ring_override : r;
}

// Launch shape of one streaming kernel (kernels_stream.h) on a class list of `known` sentences: as many
// wavefronts per workgroup as the LDS of a CU holds (one workgroup per CU), fewer workgroups when the list is
// short (at least one sentence per wave) or when the HBM scratch would pass the handle's limit.
struct StreamPlan {
  int grid = 1, waves = 1;
  uint32_t lds = 0, tcap = 0;
  uint64_t text_words = 0, scratch_words = 0;
};
StreamPlan PlanStream(const spmx_handle *h, const LengthClass &lc, bool fast, uint64_t known) {
  const int model = h->model.model_type;
  StreamPlan sp;
  const uint32_t ring = HandleRing(h);
  // a FAST text column never exceeds raw length + 1 (one-byte space symbol); GENERAL: the class's normalized capacity
  sp.tcap = fast ? (lc.rcap > kMaxStagedRaw ? lc.ncap : lc.rcap + 1) : lc.ncap;
  const uint32_t priv = StreamPrivateBytes(fast, model, lc.rcap, lc.ncap, ring);
  int waves = static_cast<int>((kLdsPerCu - kStreamSharedBytes) / priv);
  const int wmax = fast ? 16 : 8;   // __launch_bounds__ of the two kernels
  if (waves > wmax) waves = wmax;
  if (waves < 1) waves = 1;
  if (h->tile_waves_override > 0 && h->tile_waves_override < waves) waves = h->tile_waves_override;
  uint64_t grid = static_cast<uint64_t>(h->n_cu);
  if (grid * waves > known) grid = (known + waves - 1) / waves;
  if (grid < 1) grid = 1;
  const uint64_t per_wave = StreamTextDwords(sp.tcap, ring) + StreamBpWords(sp.tcap);
  const uint64_t max_waves = h->stream_scratch_limit / (per_wave * 4);
  if (grid * waves > max_waves) grid = max_waves / waves;
  if (grid < 1) grid = 1;
  sp.grid = static_cast<int>(grid);
  sp.waves = waves;
  sp.lds = StreamLdsBytes(fast, model, lc.rcap, lc.ncap, ring, static_cast<uint32_t>(waves));
  sp.text_words = grid * waves * StreamTextDwords(sp.tcap, ring);
  sp.scratch_words = grid * waves * per_wave;
  return sp;
}

// The launch sequence.  Caller holds h->mu and has set the device.
// d_begin / d_end (both or neither): the spans form (kernels_align.h).
int EncodeDevice(spmx_handle *h, const uint8_t *d_text, uint64_t text_bytes, const uint64_t *d_offsets, uint64_t n,
                 int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets, hipStream_t stream,
                 uint64_t *total_ids, uint32_t *d_begin = nullptr, uint32_t *d_end = nullptr,
                 uint32_t *d_nbegin = nullptr, uint32_t *d_nend = nullptr) {
  const bool spans = d_begin != nullptr && d_end != nullptr;
  if (total_ids) *total_ids = 0;
  if (n >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "more than 2^32 - 64 sentences in one batch");
  if (!d_offsets || !d_id_offsets) return Fail(h, kInvalidArgument, "null offsets");
  if (n == 0) {
    HIP_OR_RETURN(h, hipMemsetAsync(d_id_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  const int ncls = NumClasses(h);
  const LengthClass *cls = Classes(h);
  HIP_OR_RETURN(h, h->d_lists.Reserve(static_cast<size_t>(3 * ncls) * n));   // class lists + two hand-over lists each
  HIP_OR_RETURN(h, h->d_counts.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_tmp_off.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_tile_sums.Reserve((n + kScanTile - 1) / kScanTile + 2));
  // ids are at most one per normalized byte; the streaming kernels reserve a sentence's slot by that bound
  uint64_t expand = (h->dev.flags & kNfCompressSp) || !(h->dev.flags & kNfEscapeWs) ? 1 : 3;
  if ((h->dev.flags & kNfCompressSp) && (h->dev.flags & kNfByteFallback)) expand = 2;   // slots: bytes + 2 per space symbol
  uint64_t arena_need = expand * text_bytes + (4 + static_cast<uint64_t>(h->dev.n_prefix + h->dev.n_suffix)) * n + 64;
  if (h->profiling && !h->ev_ready) {
    for (auto &pair : h->ev) {
      HIP_OR_RETURN(h, hipEventCreate(&pair[0]));
      HIP_OR_RETURN(h, hipEventCreate(&pair[1]));
    }
    h->ev_ready = true;
  }
  const bool prof = h->profiling;
  for (int attempt = 0; attempt < 3; ++attempt) {
    HIP_OR_RETURN(h, h->d_arena.Reserve(arena_need));
    if (spans) HIP_OR_RETURN(h, h->d_arena_tb.Reserve(h->d_arena.cap));
    if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[kMaxSlots][0], stream));
    HIP_OR_RETURN(h, hipMemsetAsync(h->d_ctrl, 0, sizeof(Ctrl), stream));
    for (bool &u : h->slot_used) u = false;
    const uint32_t n32 = static_cast<uint32_t>(n);
    const int wide = h->n_cu * 8;
    {
      ClassifyArgs ca{};
      ca.offs = d_offsets; ca.n = n32; ca.n_classes = static_cast<uint32_t>(ncls);
      for (int c = 0; c < ncls; ++c) ca.rcap[c] = cls[c].rcap;
      ca.lists = h->d_lists.p; ca.list_counts = h->d_ctrl->list_counts;
      ca.key_totals = h->d_ctrl->key_totals; ca.key_cursor = h->d_ctrl->key_cursor;
      ca.sub_buckets = h->sub_buckets;
      const uint32_t chunks = (n32 + 64 * kClassifyChunk - 1) / (64 * kClassifyChunk);
      HIP_OR_RETURN(h, LaunchClassify(ca, static_cast<int>(chunks < static_cast<uint32_t>(wide) ? chunks : wide), stream));
    }
    // streaming (lane-per-sentence) kernels: every unigram model; BPE models that can be segmented word by word
    const bool bpe_stream = h->model.model_type == kBpe && (h->dev.flags & kNfBpeWordwise) && !(h->dev.flags & kNfHasUnused);
    const bool streaming = h->model.model_type == kUnigram || (bpe_stream && !h->no_stream);
    uint32_t known[kMaxClasses] = {0};   // class sizes after classify (escalations from a GENERAL kernel come on top)
    // The GENERAL kernel of a class is latency-bound (a few thousand leftover sentences, one serial recurrence each:
    // 0.7 ms for class 0 of the C2 bench whatever the count), so the two short classes share one: class 0's FAST
    // kernel appends to the same hand-over list as class 1's and the GENERAL launch of class 1 (whose capacities
    // cover both) takes them all.
    const bool merge01 = streaming && StreamFastEligible(h->dev.flags) && !h->no_fast && !h->no_merge_general && ncls >= 2 &&
                         cls[1].rcap <= kMaxStagedRaw;
    HIP_OR_RETURN(h, hipMemcpyAsync(h->h_ctrl->list_counts, h->d_ctrl->list_counts, sizeof(h->h_ctrl->list_counts),
                                    hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    for (int c = 0; c < ncls; ++c) known[c] = h->h_ctrl->list_counts[c];
    if (!streaming)      // the sentence-per-wave kernels stage a whole sentence in LDS
      for (int c = 0; c < ncls; ++c)
        if (cls[c].rcap > kMaxStagedRaw && known[c] > 0)
          return Fail(h, kOutOfRange, "a sentence is longer than 4096 bytes: this BPE model (not segmentable word by word, or with "
                                      "unused pieces) is limited to that on the device path");
    if (streaming) {
      // one scratch buffer serves every launch of the call (they run one after another): size it for the largest
      uint64_t need = 0;
      bool prev_general = false;
      for (int c = 0; c < ncls; ++c) {
        const bool shared01 = merge01 && c == 1 && known[0] > 0;   // class 1's GENERAL launch also takes class 0's leftovers
        if (known[c] == 0 && !prev_general && !shared01) continue;
        const bool fast = StreamFastEligible(h->dev.flags) && !h->no_fast && known[c] > 0;
        const bool staged = cls[c].rcap <= kMaxStagedRaw;   // document-length classes: FAST kernel only
        if (!staged && known[c] > 0 && !fast)
          return Fail(h, kOutOfRange, h->model.model_type == kBpe
                          ? "a sentence is longer than 4096 bytes: this BPE model (user-defined symbols, whitespace-as-suffix or "
                            "unescaped U+2581 rules) is limited to that on the device path"
                          : "a sentence is longer than 8192 bytes: this model (user-defined symbols, whitespace-as-suffix "
                            "or unescaped U+2581 rules) is limited to that on the device path");
        if (!staged && known[c] > 0 && spans)
          return Fail(h, kOutOfRange, "a sentence is longer than 8192 bytes: the spans form is limited to that");
        if (!staged && !fast) continue;
        const bool general = staged && !(merge01 && c == 0);
        for (int pass = fast ? 0 : 1; pass < (general ? 2 : 1); ++pass) {
          StreamPlan sp = PlanStream(h, cls[c], pass == 0, pass == 1 && shared01 ? known[0] + known[1] : known[c]);
          if (sp.scratch_words > need) need = sp.scratch_words;
        }
        prev_general = general;
      }
      HIP_OR_RETURN(h, h->d_stream.Reserve(need));
      if (bpe_stream) {      // document-length classes: HBM slices for words that outgrow the LDS slots
        uint64_t long_waves = 0;
        for (int c = 0; c < ncls; ++c) {
          if (cls[c].rcap <= kMaxStagedRaw || known[c] == 0) continue;
          const StreamPlan sp = PlanStream(h, cls[c], true, known[c]);
          const uint64_t w = static_cast<uint64_t>(sp.grid) * sp.waves;
          if (w > long_waves) long_waves = w;
        }
        if (long_waves) HIP_OR_RETURN(h, h->d_bpe_long.Reserve(long_waves * 64u * kBpeLongBytes));
      }
    }
    bool prev_general = false;
    for (int c = 0; c < ncls; ++c) {
      EncodeArgs a{};
      a.dev = h->dev; a.text = d_text; a.offs = d_offsets;
      a.list = h->d_lists.p + static_cast<size_t>(c) * n; a.list_count = &h->d_ctrl->list_counts[c];
      // a sentence whose normalized form overflows its class goes to the next one -- among the classes whose
      // GENERAL kernel can stage it; past the last of those it fails the call
      const bool has_next = c + 1 < ncls && cls[c + 1].rcap <= kMaxStagedRaw;
      a.next_list = has_next ? h->d_lists.p + static_cast<size_t>(c + 1) * n : nullptr;
      a.next_count = has_next ? &h->d_ctrl->list_counts[c + 1] : nullptr;
      a.arena = h->d_arena.p; a.arena_head = &h->d_ctrl->arena_head; a.arena_cap = h->d_arena.cap;
      a.tmp_off = h->d_tmp_off.p; a.counts = h->d_counts.p; a.status = &h->d_ctrl->status;
      a.stats = &h->d_ctrl->stats[kStatsPerClass * c];
      a.rcap = cls[c].rcap; a.ncap = cls[c].ncap;
      a.no_lane_general = h->no_lane_general ? 1u : 0u;
      a.lane_general_max_raw = h->lane_general_max_raw;
      // short classes: a stray non-ASCII sentence would hold 63 ASCII lanes up, so a tile needs 16 of them to keep
      // them; long classes: the position-parallel normalizer of the GENERAL kernel takes one sentence per wave at
      // a time, so 4 are enough (C5, 1 M sentences: 51.8 -> 45.2 ms per step)
      a.lane_general_min_lanes = h->lane_general_min_lanes ? h->lane_general_min_lanes : (cls[c].rcap <= 576 ? 16u : 4u);
      a.arena_tb = spans ? h->d_arena_tb.p : nullptr;
      if (streaming) {
        const bool shared01 = merge01 && c == 1 && known[0] > 0;
        if (known[c] == 0 && !prev_general && !shared01) continue;
        a.ring = HandleRing(h);
        const bool fast = StreamFastEligible(h->dev.flags) && !h->no_fast && known[c] > 0;
        const bool staged = cls[c].rcap <= kMaxStagedRaw;
        if (!staged && !fast) continue;
        const bool general = staged && !(merge01 && c == 0);
        const int hl = (merge01 && c <= 1) ? 0 : c;          // which hand-over list this class uses
        for (int pass = fast ? 0 : 1; pass < (general ? 2 : 1); ++pass) {
          const bool is_fast = pass == 0;
          const StreamPlan sp = PlanStream(h, cls[c], is_fast, !is_fast && shared01 ? known[0] + known[1] : known[c]);
          if (is_fast) {
            a.hard_list = staged ? h->d_lists.p + static_cast<size_t>(ncls + hl) * n : nullptr;   // no GENERAL kernel to hand over to
            a.hard_count = &h->d_ctrl->hard_counts[hl];
          } else if (fast || shared01) {
            a.list = h->d_lists.p + static_cast<size_t>(ncls + hl) * n;
            a.list_count = &h->d_ctrl->hard_counts[hl];
            a.hard_list = nullptr; a.hard_count = nullptr;
          }
          a.stream_tcap = sp.tcap;
          a.stream_text = h->d_stream.p;
          a.stream_bp = h->d_stream.p + sp.text_words;
          const int slot = (fast && !is_fast) ? kSlotGeneral + c : c;
          a.stats = &h->d_ctrl->stats[kStatsPerClass * slot];
          a.tile_cursor = h->static_tiles ? nullptr : &h->d_ctrl->tile_cursor[slot];
          a.bpe_long = (bpe_stream && !staged) ? h->d_bpe_long.p : nullptr;
          a.tiles_ascending = h->tiles_ascending ? 1u : 0u;
          a.wave_list = h->d_lists.p + static_cast<size_t>(2 * ncls + c) * n;
          a.wave_count = &h->d_ctrl->wave_counts[c];
          if (!bpe_stream && is_fast && a.ring == 16)
            snprintf(h->slot_name[slot], sizeof(h->slot_name[slot]), "EncodeStreamKernelR16<%d>", c);
          else
            snprintf(h->slot_name[slot], sizeof(h->slot_name[slot]), "Encode%sStreamKernel<%d, %s>", bpe_stream ? "Bpe" : "", c,
                     is_fast ? "true" : "false");
          if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[slot][0], stream));
          HIP_OR_RETURN(h, LaunchEncodeStream(h->model.model_type, c, is_fast, a, sp.grid, sp.waves, sp.lds, stream));
          if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[slot][1], stream));
          h->slot_used[slot] = true;
        }
        prev_general = general;
        if (bpe_stream && staged) {   // what the lane form could not take (a word longer than kBpeWordMax): sentence per wave
          a.list = h->d_lists.p + static_cast<size_t>(2 * ncls + c) * n;
          a.list_count = &h->d_ctrl->wave_counts[c];
          const int slot = kSlotGeneral + 4 + c;     // (staged BPE classes are 0..3; GENERAL slots end at kSlotGeneral + 3)
          a.stats = &h->d_ctrl->stats[kStatsPerClass * slot];
          snprintf(h->slot_name[slot], sizeof(h->slot_name[slot]), "EncodeKernel<2, %d>", c);
          const uint32_t lds = EncodeLdsBytes(kBpe, a.rcap, a.ncap);
          int per_cu = static_cast<int>(kLdsPerCu / lds);
          if (per_cu > 32) per_cu = 32;
          if (per_cu < 1) per_cu = 1;
          uint64_t grid = static_cast<uint64_t>(h->n_cu) * per_cu;
          const uint64_t most = known[c] > 64 ? known[c] : 64;
          if (grid > most) grid = most;
          if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[slot][0], stream));
          HIP_OR_RETURN(h, LaunchEncode(kBpe, c, a, static_cast<int>(grid), lds, stream));
          if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[slot][1], stream));
          h->slot_used[slot] = true;
        }
        continue;
      }
      // BPE, sentence-per-wave form (models that are not word-wise, or SPMX_NO_STREAM)
      if (cls[c].rcap > kMaxStagedRaw) continue;      // (empty: checked above)
      snprintf(h->slot_name[c], sizeof(h->slot_name[c]), "EncodeKernel<%d, %d>", h->model.model_type, c);
      const uint32_t lds = EncodeLdsBytes(h->model.model_type, a.rcap, a.ncap);
      int per_cu = static_cast<int>(kLdsPerCu / lds);
      if (per_cu > 32) per_cu = 32;
      if (per_cu < 1) per_cu = 1;
      uint64_t grid = static_cast<uint64_t>(h->n_cu) * per_cu;
      if (grid > n) grid = n;
      if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[c][0], stream));
      HIP_OR_RETURN(h, LaunchEncode(h->model.model_type, c, a, static_cast<int>(grid), lds, stream));
      if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[c][1], stream));
      h->slot_used[c] = true;
    }
    {
      ScanArgs sa{h->d_counts.p, n32, h->d_tile_sums.p, d_id_offsets};
      const uint32_t tiles = (n32 + kScanTile - 1) / kScanTile;
      HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(wide) ? tiles : wide), stream));
      CompactArgs pa{h->d_arena.p, h->d_tmp_off.p, h->d_counts.p, d_id_offsets, d_ids, d_ids ? ids_capacity : 0, n32};
      const uint64_t cblocks = (n + 63) / 64;
      const uint64_t cgrid = cblocks < static_cast<uint64_t>(h->n_cu) * 32 ? cblocks : static_cast<uint64_t>(h->n_cu) * 32;
      HIP_OR_RETURN(h, LaunchCompact(pa, static_cast<int>(cgrid), stream));
    }
    if (prof) HIP_OR_RETURN(h, hipEventRecord(h->ev[kMaxSlots][1], stream));
    HIP_OR_RETURN(h, hipMemcpyAsync(h->h_ctrl, h->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipMemcpyAsync(&h->h_ctrl->total_ids, d_id_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    const uint32_t st = h->h_ctrl->status;
    if (st & kStArenaOverflow) {   // rare: byte fallback of multi-byte unknowns; arena_head holds what was asked for
      arena_need = h->h_ctrl->arena_head + 64;
      continue;
    }
    if (st & kStTooLong) return Fail(h, kOutOfRange, "a sentence is too long for the device path (more than 1 MiB, or its normalized form exceeds the largest length class)");
    if (st & kStRevMergeOverflow) return Fail(h, kResourceExhausted, "BPE: more than 64 distinct unused merged pieces in one sentence");
    if (st & kStInternal) return Fail(h, kInternal, "all normalized characters are not consumed.");   // sentencepiece_processor.cc:628
    if (prof) {
      Profile &p = h->prof;
      p = Profile();
      p.n = kMaxSlots;
      for (int c = 0; c < kMaxSlots; ++c) {
        if (!h->slot_used[c]) continue;
        memcpy(p.name[c], h->slot_name[c], sizeof(p.name[c]));
        HIP_OR_RETURN(h, hipEventElapsedTime(&p.kernel_ms[c], h->ev[c][0], h->ev[c][1]));
        const unsigned long long *s = &h->h_ctrl->stats[kStatsPerClass * c];
        p.sentences[c] = s[0];
        p.raw_bytes[c] = s[1];
        p.ids[c] = s[2];
        for (int k = 0; k < 5; ++k) p.cycles[c][k] = s[3 + k];
        p.rcap[c] = cls[(c < kSlotGeneral ? c : c - kSlotGeneral) % ncls].rcap;
      }
      HIP_OR_RETURN(h, hipEventElapsedTime(&p.total_ms, h->ev[kMaxSlots][0], h->ev[kMaxSlots][1]));
    }
    if (total_ids) *total_ids = h->h_ctrl->total_ids;
    if (h->h_ctrl->total_ids > ids_capacity || !d_ids) {
      if (h->h_ctrl->total_ids == 0) return kOk;
      return Fail(h, kResourceExhausted, "ids_capacity is too small");
    }
    if (spans && h->h_ctrl->total_ids > 0) {
      // token begins to CSR order, then one align launch per length class over the encode's own lists
      const uint64_t total = h->h_ctrl->total_ids;
      HIP_OR_RETURN(h, h->d_tok_begin.Reserve(total));
      CompactArgs pa{h->d_arena_tb.p, h->d_tmp_off.p, h->d_counts.p, d_id_offsets, h->d_tok_begin.p, total, n32};
      const uint64_t cblocks = (n + 63) / 64;
      const uint64_t cgrid = cblocks < static_cast<uint64_t>(h->n_cu) * 32 ? cblocks : static_cast<uint64_t>(h->n_cu) * 32;
      HIP_OR_RETURN(h, LaunchCompact(pa, static_cast<int>(cgrid), stream));
      HIP_OR_RETURN(h, hipMemsetAsync(&h->d_ctrl->status, 0, sizeof(uint32_t), stream));
      // the hand-over lists of the encode are dead by now: they serve as the align kernels' escalation lists
      HIP_OR_RETURN(h, hipMemsetAsync(h->d_ctrl->hard_counts, 0, sizeof(h->d_ctrl->hard_counts), stream));
      bool prev = false;
      for (int c = 0; c < ncls; ++c) {
        const uint32_t cnt = h->h_ctrl->list_counts[c];
        if (cls[c].rcap > kMaxStagedRaw) {
          if (cnt) return Fail(h, kOutOfRange, "a sentence is longer than 8192 bytes: the spans form is limited to that");
          continue;
        }
        if (cnt == 0 && !prev) continue;
        const bool has_next = c + 1 < ncls && cls[c + 1].rcap <= kMaxStagedRaw;
        AlignArgs aa{};
        aa.dev = h->dev; aa.text = d_text; aa.offs = d_offsets;
        aa.id_offs = d_id_offsets; aa.tok_begin = h->d_tok_begin.p; aa.begin = d_begin; aa.end = d_end;
        aa.nbegin = d_nbegin; aa.nend = d_nbegin ? d_nend : nullptr;
        aa.status = &h->d_ctrl->status; aa.rcap = cls[c].rcap; aa.ncap = cls[c].ncap;
        aa.next_list = has_next ? h->d_lists.p + static_cast<size_t>(ncls + c + 1) * n : nullptr;
        aa.next_count = has_next ? &h->d_ctrl->hard_counts[c + 1] : nullptr;
        aa.list_cap = n32;
        const uint32_t lds = AlignLdsBytes(aa.rcap, aa.ncap, aa.nbegin != nullptr && aa.nend != nullptr);
        int per_cu = static_cast<int>(kLdsPerCu / lds);
        if (per_cu > 32) per_cu = 32;
        if (per_cu < 1) per_cu = 1;
        const uint64_t full = static_cast<uint64_t>(h->n_cu) * per_cu;
        if (cnt) {                       // the class's own sentences
          aa.list = h->d_lists.p + static_cast<size_t>(c) * n; aa.list_count = &h->d_ctrl->list_counts[c];
          HIP_OR_RETURN(h, LaunchAlign(aa, static_cast<int>(full < cnt ? full : cnt), lds, stream));
        }
        if (prev) {                      // what the previous class's align kernels could not hold (count on the device)
          aa.list = h->d_lists.p + static_cast<size_t>(ncls + c) * n; aa.list_count = &h->d_ctrl->hard_counts[c];
          HIP_OR_RETURN(h, LaunchAlign(aa, static_cast<int>(full < 256 ? full : 256), lds, stream));
        }
        prev = true;
      }
      uint32_t st2 = 0;
      HIP_OR_RETURN(h, hipMemcpyAsync(&st2, &h->d_ctrl->status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      HIP_OR_RETURN(h, hipStreamSynchronize(stream));
      if (st2 & kStTooLong) return Fail(h, kOutOfRange, "a sentence is too long for the spans form");
      if (st2) return Fail(h, kInternal, "token boundaries do not tile the normalized text");
    }
    return kOk;
  }
  return Fail(h, kInternal, "id arena kept overflowing");
}

// Batch Normalize on the device (kernels_normalize.h): classify -> count pass per class -> scan -> write pass per class.
// Caller holds h->mu and has set the device.
int NormalizeDevice(spmx_handle *h, const uint8_t *d_text, const uint64_t *d_offsets, uint64_t n, uint8_t *d_norm,
                    uint64_t norm_capacity, uint64_t *d_norm_offsets, uint32_t *d_n2o, hipStream_t stream,
                    uint64_t *total_bytes, bool device_text = false) {
  if (total_bytes) *total_bytes = 0;
  if (n >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "more than 2^32 - 64 sentences in one batch");
  if (!d_offsets || !d_norm_offsets) return Fail(h, kInvalidArgument, "null offsets");
  if (n == 0) {
    HIP_OR_RETURN(h, hipMemsetAsync(d_norm_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  const int ncls = NumClasses(h);
  const LengthClass *cls = Classes(h);
  HIP_OR_RETURN(h, h->d_lists.Reserve(static_cast<size_t>(3 * ncls) * n));
  HIP_OR_RETURN(h, h->d_counts.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_tile_sums.Reserve((n + kScanTile - 1) / kScanTile + 2));
  HIP_OR_RETURN(h, hipMemsetAsync(h->d_ctrl, 0, sizeof(Ctrl), stream));
  const uint32_t n32 = static_cast<uint32_t>(n);
  const int wide = h->n_cu * 8;
  {
    ClassifyArgs ca{};
    ca.offs = d_offsets; ca.n = n32; ca.n_classes = static_cast<uint32_t>(ncls);
    for (int c = 0; c < ncls; ++c) ca.rcap[c] = cls[c].rcap;
    ca.lists = h->d_lists.p; ca.list_counts = h->d_ctrl->list_counts;
    ca.key_totals = h->d_ctrl->key_totals; ca.key_cursor = h->d_ctrl->key_cursor;
      ca.sub_buckets = h->sub_buckets;
    const uint32_t chunks = (n32 + 64 * kClassifyChunk - 1) / (64 * kClassifyChunk);
    HIP_OR_RETURN(h, LaunchClassify(ca, static_cast<int>(chunks < static_cast<uint32_t>(wide) ? chunks : wide), stream));
  }
  auto pass = [&](bool write) -> int {
    for (int c = 0; c < ncls; ++c) {
      if (cls[c].rcap > kMaxStagedRaw) continue;             // checked below
      NormalizeArgs a{};
      a.dev = h->dev; a.text = d_text; a.offs = d_offsets;
      a.list = h->d_lists.p + static_cast<size_t>(c) * n; a.list_count = &h->d_ctrl->list_counts[c];
      const bool has_next = c + 1 < ncls && cls[c + 1].rcap <= kMaxStagedRaw;
      a.next_list = has_next ? h->d_lists.p + static_cast<size_t>(c + 1) * n : nullptr;
      a.next_count = has_next ? &h->d_ctrl->list_counts[c + 1] : nullptr;
      a.counts = h->d_counts.p; a.norm_offs = d_norm_offsets; a.norm = d_norm; a.n2o = d_n2o;
      a.status = &h->d_ctrl->status; a.rcap = cls[c].rcap; a.ncap = cls[c].ncap;
      a.device_text = device_text ? 1u : 0u;
      const uint32_t lds = NormalizeLdsBytes(a.rcap, a.ncap);
      int per_cu = static_cast<int>(kLdsPerCu / lds);
      if (per_cu > 32) per_cu = 32;
      if (per_cu < 1) per_cu = 1;
      uint64_t grid = static_cast<uint64_t>(h->n_cu) * per_cu;
      if (grid > n) grid = n;
      HIP_OR_RETURN(h, LaunchNormalize(write, a, static_cast<int>(grid), lds, stream));
    }
    return kOk;
  };
  if (int rc = pass(false); rc != kOk) return rc;
  {
    ScanArgs sa{h->d_counts.p, n32, h->d_tile_sums.p, d_norm_offsets};
    const uint32_t tiles = (n32 + kScanTile - 1) / kScanTile;
    HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(wide) ? tiles : wide), stream));
  }
  HIP_OR_RETURN(h, hipMemcpyAsync(h->h_ctrl, h->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipMemcpyAsync(&h->h_ctrl->total_ids, d_norm_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  for (int c = 0; c < ncls; ++c)
    if (cls[c].rcap > kMaxStagedRaw && h->h_ctrl->list_counts[c])
      return Fail(h, kOutOfRange, "a sentence is longer than 8192 bytes: Normalize on the device is limited to that");
  if (h->h_ctrl->status & kStTooLong) return Fail(h, kOutOfRange, "the normalized form of a sentence exceeds the largest length class");
  const uint64_t total = h->h_ctrl->total_ids;
  if (total_bytes) *total_bytes = total;
  if (total == 0 && !d_n2o) return kOk;
  if ((!d_norm && total) || total > norm_capacity) return Fail(h, kResourceExhausted, "norm_capacity is too small");
  if (int rc = pass(true); rc != kOk) return rc;
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  return kOk;
}

// Batch Decode on the device (kernels_decode.h): count pass -> scan -> (host checks status / capacity) -> write pass.
// Caller holds h->mu and has set the device.
int DecodeDevice(spmx_handle *h, const int32_t *d_ids, const uint64_t *d_id_offsets, uint64_t n, uint8_t *d_text,
                 uint64_t text_capacity, uint64_t *d_text_offsets, hipStream_t stream, uint64_t *total_bytes) {
  if (total_bytes) *total_bytes = 0;
  if (h->model.has_denormalizer)
    return Fail(h, kUnimplemented, "the model has a denormalizer_spec; Decode with a denormalizer is not on the device path");
  if (n >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "more than 2^32 - 64 sentences in one batch");
  if (!d_id_offsets || !d_text_offsets) return Fail(h, kInvalidArgument, "null offsets");
  if (n == 0) {
    HIP_OR_RETURN(h, hipMemsetAsync(d_text_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  HIP_OR_RETURN(h, h->d_counts.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_tile_sums.Reserve((n + kScanTile - 1) / kScanTile + 2));
  HIP_OR_RETURN(h, hipMemsetAsync(h->d_ctrl, 0, sizeof(Ctrl), stream));
  HIP_OR_RETURN(h, hipMemsetAsync(&h->d_ctrl->bad_key, 0xFF, sizeof(unsigned long long), stream));
  DecodeArgs a{};
  a.dev = h->dev; a.ids = d_ids; a.id_offs = d_id_offsets; a.n = static_cast<uint32_t>(n);
  a.counts = h->d_counts.p; a.text_offs = d_text_offsets; a.text = d_text; a.text_cap = d_text ? text_capacity : 0;
  a.status = &h->d_ctrl->status; a.bad_key = &h->d_ctrl->bad_key;
  const uint64_t wide = static_cast<uint64_t>(h->n_cu) * 32;
  const int grid = static_cast<int>(n < wide ? n : wide);
  HIP_OR_RETURN(h, LaunchDecode(false, a, grid, stream));
  {
    ScanArgs sa{h->d_counts.p, static_cast<uint32_t>(n), h->d_tile_sums.p, d_text_offsets};
    const uint32_t tiles = (static_cast<uint32_t>(n) + kScanTile - 1) / kScanTile;
    HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(h->n_cu * 8) ? tiles : h->n_cu * 8), stream));
  }
  HIP_OR_RETURN(h, hipMemcpyAsync(h->h_ctrl, h->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipMemcpyAsync(&h->h_ctrl->total_ids, d_text_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  if (h->h_ctrl->status & kStBadId) {   // sentencepiece_processor.cc:913-917 (the id reported is one of the batch's first failing sentence)
    const int id = static_cast<int>(static_cast<uint32_t>(h->h_ctrl->bad_key));
    return Fail(h, kOutOfRange, "Invalid id: " + std::to_string(id));
  }
  const uint64_t total = h->h_ctrl->total_ids;
  if (total_bytes) *total_bytes = total;
  if (total == 0) return kOk;
  if (!d_text || total > text_capacity) return Fail(h, kResourceExhausted, "text_capacity is too small");
  HIP_OR_RETURN(h, LaunchDecode(true, a, grid, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  return kOk;
}

bool ReadFile(const char *path, std::string *out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

}  // namespace

extern "C" {

int spmx_create(const void *model_bytes, uint64_t n_bytes, int device, spmx_handle **out) {
  if (!out) return Fail(nullptr, kInvalidArgument, "null output handle");
  *out = nullptr;
  if (!model_bytes || n_bytes == 0) return Fail(nullptr, kInvalidArgument, "empty model");   // "model file is empty" analogue
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return Fail(nullptr, kUnavailable, std::string("no HIP device is available (libspmx has no CPU path): ") +
                                           (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
  if (device < 0 || device >= n_dev) return Fail(nullptr, kInvalidArgument, "device ordinal out of range");
  spmx_handle *h = new spmx_handle;
  h->device = device;
  Status st = ParseModelProto(model_bytes, n_bytes, &h->model);
  if (st.ok()) st = InitializeModel(&h->model);
  if (st.ok()) st = CompileTables(h->model, &h->tables);
  if (st.ok()) st = CompileExtraOptions(h->model, "", &h->tables);
  if (!st.ok()) { const int c = Fail(nullptr, st.code, st.message); delete h; return c; }
  auto bail = [&](hipError_t err, const char *what) {
    const int c = FailHip(nullptr, err, what);
    DestroyHandle(h);
    return c;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char *e = getenv("SPMX_NO_FAST")) h->no_fast = e[0] == '1';
  if (const char *e = getenv("SPMX_NO_STREAM")) h->no_stream = e[0] == '1';
  if (const char *e = getenv("SPMX_NO_LANE_GENERAL")) h->no_lane_general = e[0] == '1';
  if (const char *e = getenv("SPMX_NO_MERGE_GENERAL")) h->no_merge_general = e[0] == '1';
  if (const char *e = getenv("SPMX_STATIC_TILES")) h->static_tiles = e[0] == '1';
  if (const char *e = getenv("SPMX_SUB_BUCKETS")) {
    const int v = atoi(e);
    h->sub_buckets = static_cast<uint32_t>(v < 1 ? 1 : (v > kMaxSubBuckets ? kMaxSubBuckets : v));
  }
  if (const char *e = getenv("SPMX_TILE_ORDER")) h->tiles_ascending = e[0] == 'a';
  if (const char *e = getenv("SPMX_LANE_GENERAL_MAX_RAW")) h->lane_general_max_raw = static_cast<uint32_t>(atoi(e));
  if (const char *e = getenv("SPMX_LANE_GENERAL_MIN_LANES")) h->lane_general_min_lanes = static_cast<uint32_t>(atoi(e));
  if (const char *e = getenv("SPMX_STREAM_SCRATCH_MB")) h->stream_scratch_limit = static_cast<uint64_t>(atoll(e)) << 20;
  if (const char *e = getenv("SPMX_TILE_WAVES")) h->tile_waves_override = atoi(e);
  if (const char *e = getenv("SPMX_FORCE_RING")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) h->ring_override = static_cast<uint32_t>(v); }
  if ((e = hipMalloc(reinterpret_cast<void **>(&h->d_ctrl), sizeof(Ctrl))) != hipSuccess) return bail(e, "hipMalloc(ctrl)");
  if ((e = hipHostMalloc(reinterpret_cast<void **>(&h->h_ctrl), sizeof(Ctrl), hipHostMallocDefault)) != hipSuccess)
    return bail(e, "hipHostMalloc(ctrl)");
  if (UploadTables(h) != kOk) {
    const std::string msg = h->error;
    DestroyHandle(h);
    return Fail(nullptr, kInternal, msg);
  }
  *out = h;
  return kOk;
}

int spmx_create_from_file(const char *filename, int device, spmx_handle **out) {
  if (out) *out = nullptr;
  std::string blob;
  if (!filename || !ReadFile(filename, &blob))   // io::LoadModelProto (sentencepiece_processor.cc:1131-1149)
    return Fail(nullptr, kNotFound, std::string("\"") + (filename ? filename : "") + "\": No such file or directory");
  return spmx_create(blob.data(), blob.size(), device, out);
}

void spmx_destroy(spmx_handle *h) { DestroyHandle(h); }

const char *spmx_last_error(const spmx_handle *h) {
  if (h) return h->error.c_str();
  static thread_local std::string copy;
  std::lock_guard<std::mutex> l(g_err_mu);
  copy = g_create_error;
  return copy.c_str();
}

int spmx_set_encode_extra_options(spmx_handle *h, const char *options) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  Status st = CompileExtraOptions(h->model, options ? options : "", &h->tables);
  if (!st.ok()) return Fail(h, st.code, st.message);
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return RefreshDevice(h, false);
}

int spmx_set_vocabulary(spmx_handle *h, const char *const *pieces, const uint64_t *piece_lens, uint64_t n) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  std::vector<std::string> v;
  v.reserve(n);
  for (uint64_t i = 0; i < n; ++i) v.emplace_back(pieces[i], piece_lens[i]);
  Status st = SetVocabulary(&h->model, v);
  if (!st.ok()) return Fail(h, st.code, st.message);
  RefreshTypeFlags(h->model, &h->tables);
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return RefreshDevice(h, true);
}

int spmx_reset_vocabulary(spmx_handle *h) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  ResetVocabulary(&h->model);
  RefreshTypeFlags(h->model, &h->tables);
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return RefreshDevice(h, true);
}

int spmx_piece_size(const spmx_handle *h) { return h ? static_cast<int>(h->model.pieces.size()) : 0; }
int spmx_piece_to_id(const spmx_handle *h, const char *piece, uint64_t len) {
  if (!h) return 0;
  return h->model.PieceToId(std::string(piece ? piece : "", piece ? len : 0));
}
int64_t spmx_id_to_piece(const spmx_handle *h, int id, char *out, uint64_t cap) {
  if (!h || id < 0 || id >= static_cast<int>(h->model.pieces.size())) return -1;
  const std::string &p = h->model.pieces[id].piece;
  if (out && cap) memcpy(out, p.data(), p.size() < cap ? p.size() : cap);
  return static_cast<int64_t>(p.size());
}
int spmx_unk_id(const spmx_handle *h) { return h ? h->model.unk_id : -1; }
int spmx_piece_type(const spmx_handle *h, int id) {
  if (!h || id < 0 || id >= static_cast<int>(h->model.pieces.size())) return -1;
  return h->model.pieces[id].type;
}
// bos_id / eos_id / pad_id (src/sentencepiece_processor.cc:1002-1017): PieceToId, -1 if it resolves to unk
static int ReservedId(const spmx_handle *h, const std::string &piece) {
  if (!h) return -1;
  const int id = h->model.PieceToId(std::string(piece.c_str()));
  return h->model.pieces[id].type == kUnknown_ ? -1 : id;
}
int spmx_bos_id(const spmx_handle *h) { return h ? ReservedId(h, h->model.bos_piece) : -1; }
int spmx_eos_id(const spmx_handle *h) { return h ? ReservedId(h, h->model.eos_piece) : -1; }
int spmx_pad_id(const spmx_handle *h) { return h ? ReservedId(h, h->model.pad_piece) : -1; }
int spmx_model_type(const spmx_handle *h) { return h ? h->model.model_type : 0; }

int spmx_encode_batch_device(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                             uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets, void *stream,
                             uint64_t *total_ids) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return EncodeDevice(h, static_cast<const uint8_t *>(d_text), text_bytes, d_offsets, n, d_ids, ids_capacity,
                      d_id_offsets, static_cast<hipStream_t>(stream), total_ids);
}

namespace {
// Host-buffer form of the batch encode, with (begin / end non-null) or without the spans.
int EncodeBatchHost(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                    uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin = nullptr,
                    uint32_t **nend = nullptr) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  const bool spans = begin != nullptr;
  if (!ids || !id_offsets || (spans && !end)) return Fail(h, kInternal, "output container is null");   // sentencepiece_processor.cc:367-370
  *ids = nullptr; *id_offsets = nullptr;
  if (spans) { *begin = nullptr; *end = nullptr; }
  const bool nspans = spans && nbegin && nend;
  if (nspans) { *nbegin = nullptr; *nend = nullptr; }
  if (n && !offsets) return Fail(h, kInvalidArgument, "null offsets");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  uint64_t *ho = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
  if (!ho) return Fail(h, kResourceExhausted, "out of host memory");
  if (n == 0) {
    ho[0] = 0; *id_offsets = ho; *ids = static_cast<int32_t *>(malloc(sizeof(int32_t)));
    if (spans) { *begin = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); *end = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); }
    if (nspans) { *nbegin = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); *nend = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); }
    return kOk;
  }
  const uint64_t base = offsets[0], text_bytes = offsets[n] - base;
  HIP_OR_RETURN(h, h->d_text.Reserve(text_bytes + 16));
  HIP_OR_RETURN(h, h->d_offs.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_id_offs.Reserve(n + 1));
  if (text_bytes) HIP_OR_RETURN(h, hipMemcpyAsync(h->d_text.p, text + base, text_bytes, hipMemcpyHostToDevice, nullptr));
  HIP_OR_RETURN(h, hipMemcpyAsync(h->d_offs.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, nullptr));
  // offsets are used as given: the kernels address text + offs[i], so rebase the text pointer instead
  const uint8_t *d_text = h->d_text.p - base;
  uint64_t cap = text_bytes / 2 + 4 * n + 64, total = 0;
  int rc = kOk;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (hipError_t e = h->d_ids.Reserve(cap); e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(ids)"); }
    if (spans) {
      hipError_t e = h->d_span_begin.Reserve(h->d_ids.cap);
      if (e == hipSuccess) e = h->d_span_end.Reserve(h->d_ids.cap);
      if (e == hipSuccess && nspans) e = h->d_nspan_begin.Reserve(h->d_ids.cap);
      if (e == hipSuccess && nspans) e = h->d_nspan_end.Reserve(h->d_ids.cap);
      if (e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(spans)"); }
    }
    rc = EncodeDevice(h, d_text, text_bytes, h->d_offs.p, n, h->d_ids.p, h->d_ids.cap, h->d_id_offs.p, nullptr, &total,
                      spans ? h->d_span_begin.p : nullptr, spans ? h->d_span_end.p : nullptr,
                      nspans ? h->d_nspan_begin.p : nullptr, nspans ? h->d_nspan_end.p : nullptr);
    if (rc != kResourceExhausted || total <= h->d_ids.cap) break;
    cap = total;
  }
  if (rc != kOk) { free(ho); return rc; }
  int32_t *hi = static_cast<int32_t *>(malloc((total ? total : 1) * sizeof(int32_t)));
  if (!hi) { free(ho); return Fail(h, kResourceExhausted, "out of host memory"); }
  hipError_t e = hipMemcpy(ho, h->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && total) e = hipMemcpy(hi, h->d_ids.p, total * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(ho); free(hi); return FailHip(h, e, "hipMemcpy(ids)"); }
  if (spans) {
    uint32_t *hb = static_cast<uint32_t *>(malloc((total ? total : 1) * sizeof(uint32_t)));
    uint32_t *he = static_cast<uint32_t *>(malloc((total ? total : 1) * sizeof(uint32_t)));
    if (!hb || !he) { free(ho); free(hi); free(hb); free(he); return Fail(h, kResourceExhausted, "out of host memory"); }
    if (total) e = hipMemcpy(hb, h->d_span_begin.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && total) e = hipMemcpy(he, h->d_span_end.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(ho); free(hi); free(hb); free(he); return FailHip(h, e, "hipMemcpy(spans)"); }
    *begin = hb;
    *end = he;
    if (nspans) {
      uint32_t *nb = static_cast<uint32_t *>(malloc((total ? total : 1) * sizeof(uint32_t)));
      uint32_t *ne = static_cast<uint32_t *>(malloc((total ? total : 1) * sizeof(uint32_t)));
      if (nb && ne && total) e = hipMemcpy(nb, h->d_nspan_begin.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost);
      if (nb && ne && e == hipSuccess && total) e = hipMemcpy(ne, h->d_nspan_end.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost);
      if (!nb || !ne || e != hipSuccess) {
        free(ho); free(hi); free(hb); free(he); free(nb); free(ne);
        *begin = nullptr; *end = nullptr;
        return e != hipSuccess ? FailHip(h, e, "hipMemcpy(spans)") : Fail(h, kResourceExhausted, "out of host memory");
      }
      *nbegin = nb;
      *nend = ne;
    }
  }
  *ids = hi;
  *id_offsets = ho;
  return kOk;
}
}  // namespace

int spmx_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                      uint64_t **id_offsets) {
  return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, nullptr, nullptr);
}

int spmx_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                            uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend) {
  if (h && (!begin || !end || (nbegin != nullptr) != (nend != nullptr))) return Fail(h, kInternal, "output container is null");
  return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, begin, end, nbegin, nend);
}

int spmx_encode_batch_spans_device(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                                   uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets,
                                   uint32_t *d_begin, uint32_t *d_end, uint32_t *d_nbegin, uint32_t *d_nend, void *stream,
                                   uint64_t *total_ids) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  if (d_ids && (!d_begin || !d_end)) return Fail(h, kInvalidArgument, "null span buffers");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return EncodeDevice(h, static_cast<const uint8_t *>(d_text), text_bytes, d_offsets, n, d_ids, ids_capacity,
                      d_id_offsets, static_cast<hipStream_t>(stream), total_ids, d_begin, d_end, d_nbegin, d_nend);
}

// NBestEncode (kernels_nbest.h), host-buffer form.
int spmx_nbest_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                            int32_t **ids, uint64_t **id_offsets, float **scores, uint64_t **result_offsets) {
  if (!h) return kInvalidArgument;
  if (!ids || !id_offsets || !scores || !result_offsets) {
    std::lock_guard<std::mutex> l(h->mu);
    return Fail(h, kInternal, "output container is null");
  }
  *ids = nullptr; *id_offsets = nullptr; *scores = nullptr; *result_offsets = nullptr;
  if (h->model.model_type != kUnigram) {
    std::lock_guard<std::mutex> l(h->mu);
    return Fail(h, kInternal, "NBestEncode is not available for the current model.");   // sentencepiece_processor.cc:662
  }
  if (nbest_size > 1024) nbest_size = 1024;                                              // unigram_model.cc:692
  if (nbest_size < 1) nbest_size = 1;
  if (nbest_size == 1 || n == 0) {          // :694-696 the plain encoder, score 0.0; one result per sentence
    int32_t *i1 = nullptr;
    uint64_t *o1 = nullptr;
    const int rc = EncodeBatchHost(h, text, offsets, n, &i1, &o1, nullptr, nullptr);
    if (rc != kOk) return rc;
    uint64_t *ro = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
    float *sc = static_cast<float *>(calloc(n + 1, sizeof(float)));
    if (!ro || !sc) { free(i1); free(o1); free(ro); free(sc); std::lock_guard<std::mutex> l(h->mu); return Fail(h, kResourceExhausted, "out of host memory"); }
    for (uint64_t s = 0; s <= n; ++s) ro[s] = s;
    *ids = i1; *id_offsets = o1; *scores = sc; *result_offsets = ro;
    return kOk;
  }
  std::lock_guard<std::mutex> l(h->mu);
  if (!offsets) return Fail(h, kInvalidArgument, "null offsets");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  const uint64_t base = offsets[0], text_bytes = offsets[n] - base;
  HIP_OR_RETURN(h, h->d_text.Reserve(text_bytes + 16));
  HIP_OR_RETURN(h, h->d_offs.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_id_offs.Reserve(n + 1));
  if (text_bytes) HIP_OR_RETURN(h, hipMemcpyAsync(h->d_text.p, text + base, text_bytes, hipMemcpyHostToDevice, nullptr));
  HIP_OR_RETURN(h, hipMemcpyAsync(h->d_offs.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, nullptr));
  const uint8_t *d_text = h->d_text.p - base;
  uint64_t ncap = 2 * text_bytes + 4 * n + 64, ntotal = 0;
  int rc = kOk;
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIP_OR_RETURN(h, h->d_norm.Reserve(ncap));
    rc = NormalizeDevice(h, d_text, h->d_offs.p, n, h->d_norm.p, h->d_norm.cap, h->d_id_offs.p, nullptr, nullptr, &ntotal, true);
    if (rc != kResourceExhausted || ntotal <= h->d_norm.cap) break;
    ncap = ntotal;
  }
  if (rc != kOk) return rc;
  const uint32_t K = static_cast<uint32_t>(nbest_size);
  NBestArgs a{};
  a.dev = h->dev; a.norm = h->d_norm.p; a.norm_offs = h->d_id_offs.p; a.n = static_cast<uint32_t>(n); a.nbest = K;
  uint64_t hyps = static_cast<uint64_t>(K) * 2048;
  a.max_hyps = static_cast<uint32_t>(hyps < 16384 ? 16384 : (hyps > 262144 ? 262144 : hyps));
  a.lane_bytes = (NbestLaneBytes(a.max_hyps) + 15) / 16 * 16;
  uint64_t waves = (n + 63) / 64;
  const uint64_t budget = 8ull << 30;                       // HBM for the lanes' slices
  if (waves * 64 * a.lane_bytes > budget) waves = budget / (64 * a.lane_bytes);
  if (waves > static_cast<uint64_t>(h->n_cu) * 8) waves = static_cast<uint64_t>(h->n_cu) * 8;
  if (waves < 1) waves = 1;
  HIP_OR_RETURN(h, h->d_nbest_scratch.Reserve(waves * 64 * a.lane_bytes));
  HIP_OR_RETURN(h, h->d_res_off.Reserve(n * K + 1));
  HIP_OR_RETURN(h, h->d_span_begin.Reserve(n * K + 1));     // result lengths
  HIP_OR_RETURN(h, h->d_res_score.Reserve(n * K + 1));
  HIP_OR_RETURN(h, h->d_counts.Reserve(n + 1));
  a.scratch = h->d_nbest_scratch.p;
  a.res_off = h->d_res_off.p; a.res_len = h->d_span_begin.p; a.res_score = h->d_res_score.p; a.res_count = h->d_counts.p;
  a.status = &h->d_ctrl->status; a.arena_head = &h->d_ctrl->arena_head;
  uint64_t arena_need = (ntotal + (4 + static_cast<uint64_t>(h->dev.n_prefix + h->dev.n_suffix)) * n) * (K < 8 ? K : 8) + 1024;
  for (int attempt = 0; attempt < 3; ++attempt) {
    HIP_OR_RETURN(h, h->d_arena.Reserve(arena_need));
    a.arena = h->d_arena.p; a.arena_cap = h->d_arena.cap;
    HIP_OR_RETURN(h, hipMemsetAsync(h->d_ctrl, 0, sizeof(Ctrl), nullptr));
    HIP_OR_RETURN(h, LaunchNBest(a, static_cast<int>(waves), nullptr));
    HIP_OR_RETURN(h, hipMemcpyAsync(h->h_ctrl, h->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, nullptr));
    HIP_OR_RETURN(h, hipStreamSynchronize(nullptr));
    const uint32_t st = h->h_ctrl->status;
    if (st & kStTooLong) return Fail(h, kOutOfRange, "NBestEncode on the device is limited to 1024 normalized bytes per sentence");
    if (st & kStNbestOverflow) return Fail(h, kResourceExhausted, "NBestEncode: the lattice or the agenda of a sentence exceeds the device capacities");
    if (st & kStArenaOverflow) { arena_need = h->h_ctrl->arena_head + 1024; continue; }
    // results -> host CSR
    std::vector<uint32_t> cnt(n), len(n * K);
    std::vector<unsigned long long> off(n * K);
    std::vector<float> sc(n * K);
    const uint64_t used = h->h_ctrl->arena_head;
    std::vector<int32_t> arena(used ? used : 1);
    HIP_OR_RETURN(h, hipMemcpy(cnt.data(), h->d_counts.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_OR_RETURN(h, hipMemcpy(len.data(), h->d_span_begin.p, n * K * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_OR_RETURN(h, hipMemcpy(off.data(), h->d_res_off.p, n * K * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_OR_RETURN(h, hipMemcpy(sc.data(), h->d_res_score.p, n * K * sizeof(float), hipMemcpyDeviceToHost));
    if (used) HIP_OR_RETURN(h, hipMemcpy(arena.data(), h->d_arena.p, used * sizeof(int32_t), hipMemcpyDeviceToHost));
    uint64_t R = 0, total = 0;
    for (uint64_t s = 0; s < n; ++s) for (uint32_t k = 0; k < cnt[s]; ++k) { ++R; total += len[s * K + k]; }
    int32_t *hi = static_cast<int32_t *>(malloc((total ? total : 1) * sizeof(int32_t)));
    uint64_t *ho = static_cast<uint64_t *>(malloc((R + 1) * sizeof(uint64_t)));
    float *hs = static_cast<float *>(malloc((R ? R : 1) * sizeof(float)));
    uint64_t *hr = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
    if (!hi || !ho || !hs || !hr) { free(hi); free(ho); free(hs); free(hr); return Fail(h, kResourceExhausted, "out of host memory"); }
    uint64_t r = 0, t = 0;
    for (uint64_t s = 0; s < n; ++s) {
      hr[s] = r;
      for (uint32_t k = 0; k < cnt[s]; ++k) {
        ho[r] = t;
        hs[r] = sc[s * K + k];
        const uint32_t ln = len[s * K + k];
        if (ln) memcpy(hi + t, arena.data() + off[s * K + k], ln * sizeof(int32_t));
        t += ln;
        ++r;
      }
    }
    hr[n] = r;
    ho[r] = t;
    *ids = hi; *id_offsets = ho; *scores = hs; *result_offsets = hr;
    return kOk;
  }
  return Fail(h, kInternal, "id arena kept overflowing");
}

int spmx_normalize_batch_device(spmx_handle *h, const void *d_text, const uint64_t *d_offsets, uint64_t n, void *d_norm,
                                uint64_t norm_capacity, uint64_t *d_norm_offsets, uint32_t *d_norm_to_orig, void *stream,
                                uint64_t *total_bytes) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return NormalizeDevice(h, static_cast<const uint8_t *>(d_text), d_offsets, n, static_cast<uint8_t *>(d_norm), norm_capacity,
                         d_norm_offsets, d_norm_to_orig, static_cast<hipStream_t>(stream), total_bytes);
}

int spmx_normalize_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, char **norm,
                         uint64_t **norm_offsets, uint32_t **norm_to_orig) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  if (!norm || !norm_offsets) return Fail(h, kInternal, "output container is null");
  *norm = nullptr; *norm_offsets = nullptr;
  if (norm_to_orig) *norm_to_orig = nullptr;
  if (n && !offsets) return Fail(h, kInvalidArgument, "null offsets");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  uint64_t *ho = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
  if (!ho) return Fail(h, kResourceExhausted, "out of host memory");
  if (n == 0) {
    ho[0] = 0; *norm_offsets = ho; *norm = static_cast<char *>(malloc(1));
    if (norm_to_orig) *norm_to_orig = static_cast<uint32_t *>(malloc(sizeof(uint32_t)));
    return kOk;
  }
  const uint64_t base = offsets[0], text_bytes = offsets[n] - base;
  HIP_OR_RETURN(h, h->d_text.Reserve(text_bytes + 16));
  HIP_OR_RETURN(h, h->d_offs.Reserve(n + 1));
  HIP_OR_RETURN(h, h->d_id_offs.Reserve(n + 1));
  if (text_bytes) HIP_OR_RETURN(h, hipMemcpyAsync(h->d_text.p, text + base, text_bytes, hipMemcpyHostToDevice, nullptr));
  HIP_OR_RETURN(h, hipMemcpyAsync(h->d_offs.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, nullptr));
  const uint8_t *d_text = h->d_text.p - base;
  uint64_t cap = 2 * text_bytes + 4 * n + 64, total = 0;
  int rc = kOk;
  for (int attempt = 0; attempt < 2; ++attempt) {
    hipError_t e = h->d_norm.Reserve(cap);
    if (e == hipSuccess && norm_to_orig) e = h->d_span_begin.Reserve(cap + n + 1);
    if (e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(norm)"); }
    rc = NormalizeDevice(h, d_text, h->d_offs.p, n, h->d_norm.p, h->d_norm.cap, h->d_id_offs.p,
                         norm_to_orig ? h->d_span_begin.p : nullptr, nullptr, &total);
    if (rc != kResourceExhausted || total <= h->d_norm.cap) break;
    cap = total;
  }
  if (rc != kOk) { free(ho); return rc; }
  char *ht = static_cast<char *>(malloc(total ? total : 1));
  uint32_t *hn = norm_to_orig ? static_cast<uint32_t *>(malloc((total + n + 1) * sizeof(uint32_t))) : nullptr;
  if (!ht || (norm_to_orig && !hn)) { free(ho); free(ht); free(hn); return Fail(h, kResourceExhausted, "out of host memory"); }
  hipError_t e = hipMemcpy(ho, h->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && total) e = hipMemcpy(ht, h->d_norm.p, total, hipMemcpyDeviceToHost);
  if (e == hipSuccess && hn) e = hipMemcpy(hn, h->d_span_begin.p, (total + n) * sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(ho); free(ht); free(hn); return FailHip(h, e, "hipMemcpy(norm)"); }
  *norm = ht;
  *norm_offsets = ho;
  if (norm_to_orig) *norm_to_orig = hn;
  return kOk;
}

void spmx_free(void *p) { free(p); }

int spmx_encode(spmx_handle *h, const char *text, uint64_t len, int32_t *ids, uint64_t cap, uint64_t *n_ids) {
  if (!h) return kInvalidArgument;
  if (!n_ids || (!ids && cap)) return Fail(h, kInternal, "output container is null");
  const uint64_t offs[2] = {0, len};
  int32_t *out = nullptr;
  uint64_t *oo = nullptr;
  const int rc = spmx_encode_batch(h, text ? text : "", offs, 1, &out, &oo);
  if (rc != kOk) return rc;
  const uint64_t total = oo[1];
  *n_ids = total;
  int ret = kOk;
  if (total > cap) ret = Fail(h, kResourceExhausted, "ids buffer is too small");
  else if (total) memcpy(ids, out, total * sizeof(int32_t));
  free(out);
  free(oo);
  return ret;
}

int spmx_decode_batch_device(spmx_handle *h, const int32_t *d_ids, const uint64_t *d_id_offsets, uint64_t n, void *d_text,
                             uint64_t text_capacity, uint64_t *d_text_offsets, void *stream, uint64_t *total_bytes) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  return DecodeDevice(h, d_ids, d_id_offsets, n, static_cast<uint8_t *>(d_text), text_capacity, d_text_offsets,
                      static_cast<hipStream_t>(stream), total_bytes);
}

int spmx_decode_batch(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, char **text,
                      uint64_t **text_offsets) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  if (!text || !text_offsets) return Fail(h, kInternal, "output container is null");
  *text = nullptr; *text_offsets = nullptr;
  if (n && !id_offsets) return Fail(h, kInvalidArgument, "null offsets");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  uint64_t *ho = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
  if (!ho) return Fail(h, kResourceExhausted, "out of host memory");
  if (n == 0) { ho[0] = 0; *text_offsets = ho; *text = static_cast<char *>(malloc(1)); return kOk; }
  const uint64_t base = id_offsets[0], n_ids = id_offsets[n] - base;
  if (hipError_t e = h->d_ids.Reserve(n_ids + 16); e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(ids)"); }
  if (hipError_t e = h->d_offs.Reserve(n + 1); e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(offsets)"); }
  if (hipError_t e = h->d_id_offs.Reserve(n + 1); e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(offsets)"); }
  hipError_t e = hipSuccess;
  if (n_ids) e = hipMemcpyAsync(h->d_ids.p, ids + base, n_ids * sizeof(int32_t), hipMemcpyHostToDevice, nullptr);
  if (e == hipSuccess) e = hipMemcpyAsync(h->d_offs.p, id_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, nullptr);
  if (e != hipSuccess) { free(ho); return FailHip(h, e, "hipMemcpy(ids)"); }
  const int32_t *d_ids = h->d_ids.p - base;     // the kernels address ids + id_offsets[i]
  uint64_t cap = n_ids * 6 + 64, total = 0;
  int rc = kOk;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (hipError_t e2 = h->d_text.Reserve(cap); e2 != hipSuccess) { free(ho); return FailHip(h, e2, "hipMalloc(text)"); }
    rc = DecodeDevice(h, d_ids, h->d_offs.p, n, h->d_text.p, h->d_text.cap, h->d_id_offs.p, nullptr, &total);
    if (rc != kResourceExhausted || total <= h->d_text.cap) break;
    cap = total;
  }
  if (rc != kOk) { free(ho); return rc; }
  char *ht = static_cast<char *>(malloc(total ? total : 1));
  if (!ht) { free(ho); return Fail(h, kResourceExhausted, "out of host memory"); }
  e = hipMemcpy(ho, h->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && total) e = hipMemcpy(ht, h->d_text.p, total, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(ho); free(ht); return FailHip(h, e, "hipMemcpy(text)"); }
  *text = ht;
  *text_offsets = ho;
  return kOk;
}

int spmx_decode(spmx_handle *h, const int32_t *ids, uint64_t n_ids, char *out, uint64_t cap, uint64_t *len) {
  if (!h) return kInvalidArgument;
  if (!len || (!out && cap)) return Fail(h, kInternal, "output container is null");
  const uint64_t offs[2] = {0, n_ids};
  char *t = nullptr;
  uint64_t *to = nullptr;
  const int32_t dummy = 0;
  const int rc = spmx_decode_batch(h, ids ? ids : &dummy, offs, 1, &t, &to);
  if (rc != kOk) return rc;
  const uint64_t total = to[1];
  *len = total;
  int ret = kOk;
  if (total > cap) ret = Fail(h, kResourceExhausted, "text buffer is too small");
  else if (total) memcpy(out, t, total);
  free(t);
  free(to);
  return ret;
}

int spmx_split_lines_device(spmx_handle *h, const void *d_file, uint64_t bytes, void *d_text, uint64_t text_capacity,
                            uint64_t *d_offsets, uint64_t offsets_capacity, void *stream_v, uint64_t *n_lines,
                            uint64_t *text_bytes) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  if (n_lines) *n_lines = 0;
  if (text_bytes) *text_bytes = 0;
  if (!n_lines || !text_bytes) return Fail(h, kInternal, "output container is null");
  if (bytes && (!d_file || (reinterpret_cast<uintptr_t>(d_file) & 15u))) return Fail(h, kInvalidArgument, "d_file must be 16-byte aligned");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (bytes == 0) {
    if (d_offsets && offsets_capacity) HIP_OR_RETURN(h, hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  const uint64_t chunks = (bytes + kSplitChunk - 1) / kSplitChunk;
  if (chunks >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "file image too large for one call");
  HIP_OR_RETURN(h, h->d_counts.Reserve(chunks + 1));
  HIP_OR_RETURN(h, h->d_chunk_base.Reserve(chunks + 1));
  HIP_OR_RETURN(h, h->d_tile_sums.Reserve((chunks + kScanTile - 1) / kScanTile + 2));
  SplitArgs a{};
  a.file = static_cast<const uint8_t *>(d_file); a.bytes = bytes; a.counts = h->d_counts.p;
  a.chunk_base = h->d_chunk_base.p; a.text = static_cast<uint8_t *>(d_text); a.offsets = d_offsets;
  const uint64_t wide = static_cast<uint64_t>(h->n_cu) * 16;
  const int grid = static_cast<int>(chunks < wide ? chunks : wide);
  HIP_OR_RETURN(h, LaunchSplit(false, a, grid, stream));
  {
    ScanArgs sa{h->d_counts.p, static_cast<uint32_t>(chunks), h->d_tile_sums.p, h->d_chunk_base.p};
    const uint32_t tiles = (static_cast<uint32_t>(chunks) + kScanTile - 1) / kScanTile;
    HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(h->n_cu * 8) ? tiles : h->n_cu * 8), stream));
  }
  uint8_t last = 0;
  HIP_OR_RETURN(h, hipMemcpyAsync(&h->h_ctrl->total_ids, h->d_chunk_base.p + chunks, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipMemcpyAsync(&last, static_cast<const uint8_t *>(d_file) + bytes - 1, 1, hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  const uint64_t nl = h->h_ctrl->total_ids;
  const uint64_t lines = nl + (last != 0x0A ? 1 : 0);      // std::getline: a last line without '\n' counts
  *n_lines = lines;
  *text_bytes = bytes - nl;
  if (!d_text || !d_offsets || text_capacity < bytes - nl || offsets_capacity < lines + 1)
    return Fail(h, kResourceExhausted, "text_capacity / offsets_capacity is too small");
  HIP_OR_RETURN(h, LaunchSplit(true, a, grid, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  return kOk;
}

int spmx_set_profiling(spmx_handle *h, int enabled) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  h->profiling = enabled != 0;
  return kOk;
}

int spmx_last_phase_cycles(const spmx_handle *h, uint64_t *cycles) {
  if (!h) return 0;
  for (int c = 0; c < h->prof.n; ++c)
    for (int k = 0; k < 5; ++k) cycles[5 * c + k] = h->prof.cycles[c][k];
  return h->prof.n;
}

int spmx_last_profile_name(const spmx_handle *h, int slot, char *out, uint64_t cap) {
  if (!h || slot < 0 || slot >= h->prof.n || !out || !cap) return 0;
  snprintf(out, cap, "%s", h->prof.name[slot]);
  return static_cast<int>(strlen(h->prof.name[slot]));
}

int spmx_last_profile(const spmx_handle *h, float *kernel_ms, uint64_t *sentences, uint64_t *raw_bytes, uint64_t *ids,
                      uint64_t *bytes, uint32_t *rcap, float *total_ms) {
  if (!h) return 0;
  const Profile &p = h->prof;
  for (int c = 0; c < p.n; ++c) {
    if (kernel_ms) kernel_ms[c] = p.kernel_ms[c];
    if (sentences) sentences[c] = p.sentences[c];
    if (raw_bytes) raw_bytes[c] = p.raw_bytes[c];
    if (ids) ids[c] = p.ids[c];
    // L + 8 + 4 T' + 8 per sentence
    if (bytes) bytes[c] = p.raw_bytes[c] + 16 * p.sentences[c] + 4 * p.ids[c];
    if (rcap) rcap[c] = p.rcap[c];
  }
  if (total_ms) *total_ms = p.total_ms;
  return p.n;
}

}  // extern "C"
