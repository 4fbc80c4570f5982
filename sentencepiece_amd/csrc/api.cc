// libspmx.so: the C ABI of include/spmx.h over the HIP kernels of kernels.hip.
//
// One handle = one loaded model on one GPU: the compiled tables live in HBM for the life of the handle.  Every call
// leases a WORKSPACE from the handle's pool (grow-only device buffers: class lists, id arena, scan scratch, host
// staging; a control block mirrored in pinned host memory; a stream of its own for the host-buffer forms), so calls
// from several host threads run concurrently, as SentencePieceProcessor's const methods do.  One encode call is a
// fixed sequence of launches on one stream with two small read-backs:
//
//   memset ctrl -> classify -> D2H class sizes -> streaming launch(es) -> scan x3 -> compact -> D2H {ctrl, total}
//
// and, only when the fast kernels set sentences aside (a normalized form that overflows its class's column, a BPE word
// longer than the lane form's slots ...): an overflow launch with exact capacities / the long form, then scan and
// compact again.  No sentence fails for its length; a sentence that does fail (a control piece among BPE symbols)
// yields no ids and a status byte, and the rest of the batch is unaffected.
//
// There is no CPU path: without a usable HIP device spmx_create fails with UNAVAILABLE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <new>
#include <sstream>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/spmx.h"
#include "launch.h"
#include "model.h"
#include "tables.h"

using namespace spmx;

namespace {

constexpr uint32_t kLdsPerCu = 160u * 1024u;   // gfx950 (MI355X_MICROARCH.md)
constexpr uint32_t kDynSlotsDefault = 1u << 20;       // call-local word memo: slots (64 B each) and words per call it can take
constexpr uint32_t kDynListCapDefault = 1u << 18;     // (handle fields dyn_slots / dyn_list_cap; tests shrink them)

thread_local std::string t_error;              // text of the calling thread's last failing call

// kernel slots of the per-call profile
enum { kSlotMain = 0, kSlotDoc = 1, kSlotExact = 2, kSlotWave = 3, kSlotLong = 4, kSlotWord = 5, kSlotWord2 = 6, kSlotSplit = 7, kNumSlots = 8 };

// ctrl block layout (device + pinned host mirror), zeroed before every call
struct Ctrl {
  uint32_t list_counts[kMaxClasses];
  uint32_t gen_counts[kMaxClasses];   // classify's plain scan: the sentences set aside for the general kernels, per class (read together with list_counts)
  uint32_t key_totals[kSortKeys], key_cursor[kSortKeys];   // classify: counting sort by (class, length sub-bucket)
  uint32_t status;
  uint32_t retry_count[2];            // long form: sentences that found the pool exhausted (ping-pong)
  uint32_t pad;
  // The words the kernels update with atomics all the time -- tile cursors, list counters, the arena's head -- each on a
  // 128-byte line of their own.  Found by accident in round 6: the control block grew by one 16-byte queue, which moved the
  // word kernels' list counters off the line of their tile cursor, and the first word round went from 3.84 to 3.1 ms with
  // its code unchanged (profiles/README_r06.md).
  StreamQueue q[7];                   // tile queues of the main / document / overflow launches; [3], [4]: the word kernels; [5]: the tail launch; [6]: the split launch
  alignas(128) uint32_t left_counts[3][32];   // word kernels: second-round input / general input / what the second round left, per class ([kMaxClasses] of a row used; a row = a line)
  uint32_t dyn_count;                     // ... words entered into the call-local memo (must follow left_counts: read together)
  alignas(128) uint32_t align_counts[kMaxClasses]; // spans form: escalation lists of the staged align kernels
  alignas(128) SideLists side;
  alignas(128) unsigned long long arena_head;
  alignas(128) unsigned long long pool_head;       // long form: bytes of slices asked for
  alignas(128) unsigned long long stats[kStatsPerClass * kNumSlots];
  unsigned long long bad_key;         // decode: min over offending (sentence << 32 | id)
  uint32_t big_count[2];              // CompactKernel: document blocks listed for CompactBigKernel (ids; token begins of the spans form)
  uint64_t total_ids;                 // copied from id_offs[n] by the final D2H
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;   // elements
  hipError_t Reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    const size_t want = (n + n / 4 + 64 + 15) & ~static_cast<size_t>(15);   // (whole 16-byte units, whatever T: CompactKernel reads the arena by them)
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void Free() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Pinned host memory NEAR THE GPU: hipHostMalloc places its pages by the calling thread's memory policy, and on a
// two-socket host a staging buffer on the other socket makes every H2D / D2H copy cross the inter-socket link.  While one
// of these is alive the thread PREFERS the NUMA node the current device hangs off (its PCI function's numa_node in
// sysfs; set_mempolicy through the raw system call: no libnuma); anything that fails leaves the default policy alone.
// SPMX_NO_NUMA=1 switches it off.
struct PreferGpuNode {
  bool set = false;
  PreferGpuNode() {
#if defined(__linux__) && !defined(SPMX_EMULATED)
    static const bool off = getenv("SPMX_NO_NUMA") != nullptr;
    if (off) return;
    int dev = 0;
    char bus[64] = {0};
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) return;
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = static_cast<char>(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0 || node >= 1024) return;
    // (a caller that runs under its own policy -- numactl --membind / --interleave, set_mempolicy -- keeps it: the override is
    // only for threads on the default policy, which is also what the destructor puts back)
    int cur = 0;
    if (syscall(SYS_get_mempolicy, &cur, nullptr, 0ul, nullptr, 0ul) != 0 || cur != 0 /* MPOL_DEFAULT */) return;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    set = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, sizeof(mask) * 8 + 1) == 0;
#endif
  }
  ~PreferGpuNode() {
#if defined(__linux__) && !defined(SPMX_EMULATED)
    if (set) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0);
#endif
  }
};

// pinned host staging (host-buffer forms)
template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t Reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = (n + n / 4 + 64 + 15) & ~static_cast<size_t>(15);   // (whole 16-byte units, whatever T: CompactKernel reads the arena by them)
    PreferGpuNode near;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void Free() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Output arrays of the big host-buffer calls are PINNED (the D2H copies land in them directly) and recycled: spmx_free()
// of such a pointer returns it to this process-wide pool instead of unpinning ~GBs per call; plain malloc'd outputs
// are not in the registry and go to free().
struct PinnedPool {
  std::mutex mu;
  std::map<void *, size_t> live;                   // handed to a caller
  std::multimap<size_t, void *> idle;              // by capacity
  size_t idle_bytes = 0;
  void *Get(size_t bytes) {
    if (bytes == 0) bytes = 1;
    {
      std::lock_guard<std::mutex> l(mu);
      auto it = idle.lower_bound(bytes);
      if (it != idle.end() && it->first <= 2 * bytes + (1u << 20)) {
        void *p = it->second;
        const size_t cap = it->first;
        idle.erase(it);
        idle_bytes -= cap;
        live[p] = cap;
        return p;
      }
    }
    void *p = nullptr;
    const size_t cap = bytes + bytes / 8;
    PreferGpuNode near;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    live[p] = cap;
    return p;
  }
  bool Put(void *p) {                              // false: not ours
    void *drop = nullptr;
    {
      std::lock_guard<std::mutex> l(mu);
      auto it = live.find(p);
      if (it == live.end()) return false;
      const size_t cap = it->second;
      live.erase(it);
      if (idle_bytes + cap > (24ull << 30)) drop = p;         // keep at most 24 GiB around
      else { idle.emplace(cap, p); idle_bytes += cap; }
    }
    if (drop) (void)hipHostFree(drop);
    return true;
  }
};
PinnedPool g_pinned;

struct Profile {
  int n = 0;
  char name[kNumSlots][48] = {{0}};
  float kernel_ms[kNumSlots] = {0};
  uint64_t sentences[kNumSlots] = {0}, raw_bytes[kNumSlots] = {0}, ids[kNumSlots] = {0};
  uint64_t cycles[kNumSlots][5] = {{0}};
  uint64_t path[8] = {0};     // {hard-list sentences, overflow-list sentences, long-form sentences, failed sentences}
  float total_ms = 0.f;
};

// Everything one call needs besides the model: leased from the handle's pool for the duration of the call.
struct Workspace {
  DevBuf<uint32_t> d_lists, d_counts;
  DevBuf<uint64_t> d_tmp_off, d_tile_sums, d_chunk_base;
  DevBuf<uint32_t> d_big_list;   // CompactKernel's list of document blocks (kernels.h compact_big_block)
  DevBuf<int32_t> d_arena, d_arena_tb, d_tok_begin;
  DevBuf<uint32_t> d_span_begin, d_span_end, d_nspan_begin, d_nspan_end;
  DevBuf<uint8_t> d_norm, d_nbest_scratch, d_slab, d_pool, d_sent_status, d_flags;
  DevBuf<unsigned long long> d_res_off, d_dyn_tag;     // d_dyn_*: the call-local word memo (kernels_word.h)
  DevBuf<U4> d_dyn_ent, d_resume;
  DevBuf<uint32_t> d_dyn_list;
  DevBuf<float> d_res_score;
  Ctrl *d_ctrl = nullptr;
  Ctrl *h_ctrl = nullptr;   // pinned
  // host-buffer forms
  DevBuf<uint8_t> d_text, d_dn_text;       // d_dn_*: the decoded text before the denormalizer
  DevBuf<uint64_t> d_offs, d_id_offs, d_dn_offs;
  DevBuf<int32_t> d_ids;
  DevBuf<uint8_t> d_lit_bytes;           // Decode(pieces): the pieces outside the vocabulary (kernels_decode.h DecodeArgs::lit_*)
  DevBuf<uint32_t> d_lit_offs;
  const uint8_t *lit_bytes = nullptr;    // set for the duration of one spmx_decode_batch_pieces call
  const uint32_t *lit_offs = nullptr;
  uint32_t n_lit = 0;
  PinBuf<uint8_t> h_text;       // pinned staging of the pipelined host form
  PinBuf<uint64_t> h_offs, h_id_offs;
  // set around its calls by the pipelined host form: the LARGEST chunk of the batch (bytes, sentences).  A workspace sizes its
  // arena for that, not for the chunk it happens to get: a length-bucketed batch has small chunks and large ones, a worker leases
  // whichever workspace is free, and a workspace that grows in the second or third call of a handle costs that call a
  // hipFree + hipMalloc of half a gigabyte in the middle of the pipeline (10 M C2 sentences: calls 2 and 3 took 115 and 60 ms
  // against 50 from the fourth on, scripts/host_calls_probe.py)
  uint64_t reserve_text_bytes = 0, reserve_n = 0;
  float bpe_dropout = 0.f;      // set around a call by spmx_sample_encode_batch: BPE-dropout through the long form
  uint64_t sample_seed = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;          // the general launches run here while the second word round runs on the call's stream
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev[kNumSlots + 1][2] = {};   // per kernel slot + the whole call
  bool ev_ready = false;
  bool slot_used[kNumSlots] = {false};
  char slot_name[kNumSlots][48] = {{0}};
  Profile prof;

  ~Workspace() {
    d_lists.Free(); d_counts.Free(); d_tmp_off.Free(); d_tile_sums.Free(); d_chunk_base.Free(); d_big_list.Free(); d_arena.Free();
    d_arena_tb.Free(); d_tok_begin.Free(); d_span_begin.Free(); d_span_end.Free(); d_nspan_begin.Free(); d_nspan_end.Free();
    d_norm.Free(); d_nbest_scratch.Free(); d_slab.Free(); d_pool.Free(); d_sent_status.Free(); d_flags.Free(); d_res_off.Free();
    d_res_score.Free(); d_dyn_tag.Free(); d_dyn_ent.Free(); d_dyn_list.Free(); d_resume.Free(); d_text.Free(); d_offs.Free(); d_id_offs.Free(); d_ids.Free(); d_dn_text.Free(); d_dn_offs.Free();
    h_text.Free(); h_offs.Free(); h_id_offs.Free();
    if (d_ctrl) (void)hipFree(d_ctrl);
    if (h_ctrl) (void)hipHostFree(h_ctrl);
    if (stream) (void)hipStreamDestroy(stream);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_ready) for (auto &pair : ev) { (void)hipEventDestroy(pair[0]); (void)hipEventDestroy(pair[1]); }
  }
};

}  // namespace

struct spmx_handle {
  std::mutex mu;                 // the workspace pool, the last profile
  ModelData model;
  HostTables tables;
  std::string extra_options;
  std::string serialized;        // the ModelProto as it was loaded (serialized_model_proto, src/sentencepiece_processor.h:694)
  int device = 0;
  int n_cu = 256;
  // device copies of the tables
  DevBuf<uint32_t> d_ndarts, d_npair, d_sym_final, d_dec_info, d_dec_off;
  DevBuf<uint8_t> d_dec_bytes;
  DevBuf<uint8_t> d_nblob, d_plen;
  DevBuf<U4> d_ptrie, d_chartab, d_pairtab, d_wordtab, d_umemo, d_umemo16, d_uhot, d_uall, d_uhot2, d_cfirst;
  DevBuf<uint16_t> d_udisp;
  DevBuf<float> d_pscore;
  DevBuf<U2> d_utrie;
  DevBuf<uint16_t> d_sym_len;
  DevBuf<int32_t> d_byte_ids;
  SpmxDev dev{};   // scalars + device pointers
  // the denormalizer (denormalizer_spec with a charsmap): normalizer tables of its own
  HostTables dn_tables;
  DevBuf<uint32_t> dn_ndarts, dn_npair;
  DevBuf<uint8_t> dn_nblob;
  DevBuf<U2> dn_utrie;
  SpmxDev dn_dev{};
  std::vector<std::unique_ptr<Workspace>> pool;   // idle workspaces
  bool profiling = false;
  Profile prof;                  // of the last profiled encode call
  // A/B switches (environment, read once at load)
  bool no_fast = false;          // SPMX_NO_FAST=1: every tile runs the general normalizer
  bool no_lane_general = false;  // SPMX_NO_LANE_GENERAL=1: main tiles set every non-ASCII sentence aside
  uint32_t char_norm_mode = 0;   // SPMX_NO_CHAR_NORM=1 -> 1: no tile takes the character-stepping normalizer; SPMX_CHAR_NORM_ALWAYS=1 -> 2 (test seam): every tile
  bool no_stream = false;        // SPMX_NO_STREAM=1: BPE in the sentence-per-wave form only
  uint64_t arena_first = 0;      // SPMX_ARENA_FIRST: cap on the first attempt's id arena (tests: the overflow-and-retry path)
  bool no_bp_short = false;      // SPMX_NO_BP_SHORT=1: 32-bit back-pointer entries for every unigram model
  bool no_wave = false;          // SPMX_NO_WAVE=1: BPE models that are not word-wise use the long form only
  std::atomic<int> word_backoff{0};   // calls that leave the word rounds out (they did not pay on the last batch that tried)
  bool no_word = false;          // SPMX_NO_WORD_KERNEL=1: unigram models skip the word kernels (kernels_word.h)
  bool no_word_dp = false;       // SPMX_NO_WORD_DP=1: ... skip the second pass only
  uint32_t dyn_slots = kDynSlotsDefault;        // SPMX_DYN_SLOTS_LOG2: slots of the call-local word memo (a power of two)
  uint32_t dyn_list_cap = kDynListCapDefault;   // SPMX_DYN_LIST_CAP: words it takes per call; what it cannot take stays with the general kernels
  uint64_t nbest_budget = 32ull << 30;  // SPMX_NBEST_BUDGET_GB: HBM the lattice slices of one launch may take (200 k sentences, n-best 5: 0.83 M sentences/s at 8 GB, 1.15 M at 32, 1.19 M at 96)
  uint32_t nbest_hyps_min = 4096;    // SPMX_NBEST_HYPS_MIN: hypotheses a lane's A* may hold in the first launch (what outgrows it runs again)
  uint32_t tile_min_lanes = 1;   // SPMX_TILE_MIN_LANES: a main tile has at least this many sentences even when that leaves wavefronts without a tile of the class
  // SetDecodeExtraOptions: the net effect of the options on a sentence's ids (kernels_decode.h DecodeArgs::x_*)
  int32_t dx_npre = 0, dx_nsuf = 0, dx_pre[kMaxExtra] = {0}, dx_suf[kMaxExtra] = {0};
  bool dx_reverse = false;
  bool dx_unk = false;           // ... holds `unk` / `unk_piece`: Decode(pieces) turns a piece outside the vocabulary into the unknown piece (.cc:1050-1058)
  uint32_t compact_staged = 2048;  // ids a CompactKernel wave's LDS image holds (SPMX_COMPACT_STAGED=<ids>; 0: the search form for every block; results identical)
  bool no_direct = false;        // SPMX_NO_DIRECT=1: the word rounds take classify's lists even where they could do without
  int fork_cus = 0;              // SPMX_FORK_CUS: the general launch beside the word rounds takes at most this many CUs (0: every CU)
  int fork_waves = 4;            // SPMX_FORK_WAVES: wavefronts per workgroup of the general launch while it runs next to the first word round (0: by its size)
  bool early_tail = false;       // SPMX_EARLY_TAIL=1: a direct call's first-round give-ups get a tail launch of their own beside round 2 (below: measured, not the default)
  bool no_overlap = false;       // SPMX_NO_OVERLAP=1: the general launches do not run next to the word rounds
  bool no_ids16 = false;         // SPMX_NO_IDS16=1: the word kernels write 32-bit ids into the arena whatever the vocabulary's size
  bool no_scan = false;          // SPMX_NO_SCAN=1: classify does not set the non-plain sentences aside (the word rounds find them)
  bool no_word_dyn = false;      // SPMX_NO_WORD_DYN=1: no call-local word memo (one word round, then the DP pass)
  bool memo_unsafe = false;      // TEST SEAM of the emulator build (SPMX_TEST_SEAMS + SPMX_WORDMEMO_UNSAFE=1): the call-local memo takes no margin either
  bool force_word_dp = false;    // SPMX_FORCE_WORD_DP=1: the second pass runs whatever the first one left (tests)
  bool no_split = false;         // SPMX_NO_SPLIT=1: no class takes the split form (kernels_matchfold.h)
  uint32_t split_min_raw = 576;  // SPMX_SPLIT_MIN: classes of MORE than this many raw bytes (up to kMfMaxRaw) take the split form
  bool split_own_launch = false; // SPMX_SPLIT_LAUNCH=1: the split classes get a launch of their own (EncodeSplitKernel)
  uint32_t split_tiles = 1;      // SPMX_SPLIT_TILES: split tiles per wavefront and class the planner aims at (their match phase is sequential: small tiles balance, large tiles fold more lanes at once)
  uint32_t split_per_byte = 4;   // SPMX_SPLIT_CANDS: candidates per normalized byte a sentence's stream holds before the overflow launch takes the sentence
  uint32_t compact_big = kCompactBigIds;   // CompactKernel: blocks with more ids go to CompactBigKernel (0: none do)
  bool uw_exact = false;
  int uw_pipe = 1;               // SPMX_UW_PIPE: 0 never / 1 few long documents (default) / 2 always the two-wavefront form of the wave-cooperative kernel
  bool no_uni_wave = false;      // SPMX_NO_UNI_WAVE=1: unigram models never take the wave-cooperative form (kernels_uniwave.h)
  uint32_t uni_wave_max = 0;     // SPMX_UNI_WAVE_MAX: a staged class with fewer sentences than this takes the wave-cooperative form
  int word_wgs = 1;              // SPMX_WORD_WGS: workgroups per CU of the word kernel's first pass
  uint64_t table_bytes = 0;      // device bytes of the tables uploaded at load (spmx_handle_info)
  double load_ms = 0.0;          // parse + table build + upload, wall clock
  int word_form = 3;             // SPMX_WORD_WAVE: which word rounds take the word-per-lane form (kernels_wordwave.h): bit 0 the first, bit 1 the second; 0: the sentence-per-lane loops
  int wordwave_waves = 14;       // SPMX_WORDWAVE_WAVES: wavefronts per workgroup of the word-per-lane kernels (C2's first round: 8 -> 3.71 ms, 10 -> 3.31, 12 -> 3.08, 13 -> 3.02, 14 -> 2.96, 15 -> 2.94 with a worse step; 14 x 10 KB + the shared tables = 153 KB of LDS)
  int word_waves = 12;           // SPMX_WORD_WAVES: wavefronts per workgroup of the word kernels (C2 step: 16 -> 8.60 ms, 14 -> 8.39, 12 -> 8.37, 10 -> 8.52, 8 -> 8.90)
  int tile_waves_override = 0;   // SPMX_TILE_WAVES: cap on wavefronts per workgroup of the streaming kernels
  uint32_t lane_general_min_lanes = 0;   // SPMX_LANE_GENERAL_MIN_LANES (0: per class)
  uint32_t sub_buckets = kSubBuckets;    // SPMX_SUB_BUCKETS: length sub-buckets per class in the classify sort (1..64)
  int host_threads = 24;         // SPMX_HOST_THREADS: workers of the pipelined host form (chunks in flight)
  uint64_t host_chunk = 0;       // SPMX_HOST_CHUNK: sentences per chunk of the pipelined host form (0: by batch size)
  int reserve_cus = 0;           // SPMX_RESERVE_CUS: CUs the persistent encode grids leave free (an RCCL gather in flight)
  uint32_t ring_override = 0;    // SPMX_FORCE_RING: score-ring entries (must exceed the longest piece)
  uint64_t stream_scratch_limit = 16ull << 30;   // SPMX_STREAM_SCRATCH_MB: cap on the streaming kernels' HBM scratch
  bool wide_tcap = false;        // SPMX_WIDE_TCAP=1: text columns of the class's full normalized capacity (default: 1.25 x
                                 // its raw size -- ASCII and CJK text do not grow; what does takes the overflow launch)
  uint32_t main_max_raw = kStreamMainMaxRaw;     // SPMX_MAIN_MAX_RAW: classes up to this size share the main launch
  LengthClass classes[kNumClasses];              // SPMX_CLASSES="r:n,r:n,...": the class table (tests shrink it)
};

namespace {

int Fail(spmx_handle *, int code, const std::string &msg) {
  t_error = msg;
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

// RAII lease of a workspace
struct Lease {
  spmx_handle *h;
  std::unique_ptr<Workspace> ws;
  explicit Lease(spmx_handle *hh) : h(hh) {
    std::lock_guard<std::mutex> l(h->mu);
    if (!h->pool.empty()) { ws = std::move(h->pool.back()); h->pool.pop_back(); }
  }
  // creates the workspace's fixed parts on first use; the device must be current
  int Ready() {
    if (ws) return kOk;
    ws.reset(new (std::nothrow) Workspace);
    if (!ws) return Fail(h, kResourceExhausted, "out of host memory");
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ws->d_ctrl), sizeof(Ctrl));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ws->h_ctrl), sizeof(Ctrl), hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ws->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {          // a half-built workspace must not reach the pool (~Lease): the next call would lease it
      ws.reset();
      return FailHip(h, e, "creating a workspace");
    }
    return kOk;
  }
  ~Lease() {
    if (!ws) return;
    std::lock_guard<std::mutex> l(h->mu);
    h->pool.push_back(std::move(ws));
  }
};

thread_local uint64_t t_upload_bytes = 0;   // device bytes the tables of the handle being loaded take (spmx_handle_info)
template <typename T>
hipError_t Upload(DevBuf<T> *b, const std::vector<T> &v) {
  hipError_t e = b->Reserve(v.size() ? v.size() : 1);
  if (e != hipSuccess) return e;
  t_upload_bytes += b->cap * sizeof(T);
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
  HIP_OR_RETURN(h, Upload(&h->d_cfirst, t.cfirst));
  HIP_OR_RETURN(h, Upload(&h->d_plen, t.plen));
  HIP_OR_RETURN(h, Upload(&h->d_utrie, t.utrie));
  HIP_OR_RETURN(h, Upload(&h->d_chartab, t.chartab));
  HIP_OR_RETURN(h, Upload(&h->d_pairtab, t.pairtab));
  HIP_OR_RETURN(h, Upload(&h->d_wordtab, t.wordtab));
  HIP_OR_RETURN(h, Upload(&h->d_umemo, t.umemo));
  HIP_OR_RETURN(h, Upload(&h->d_umemo16, t.umemo16));
  HIP_OR_RETURN(h, Upload(&h->d_uall, t.uall));
  HIP_OR_RETURN(h, Upload(&h->d_udisp, t.udisp));
  HIP_OR_RETURN(h, Upload(&h->d_uhot2, t.uhot2));
  HIP_OR_RETURN(h, Upload(&h->d_uhot, t.uhot));
  HIP_OR_RETURN(h, Upload(&h->d_pscore, t.pscore));
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
  h->dev.cfirst = t.cfirst.empty() ? nullptr : h->d_cfirst.p;
  h->dev.plen = h->d_plen.p;
  h->dev.utrie = h->d_utrie.p;
  h->dev.chartab = h->d_chartab.p;
  h->dev.pairtab = h->d_pairtab.p;
  h->dev.wordtab = h->d_wordtab.p;
  h->dev.umemo = h->d_umemo.p;
  h->dev.umemo16 = h->d_umemo16.p;
  h->dev.uall = h->d_uall.p;
  h->dev.udisp = h->d_udisp.p;
  h->dev.uhot2 = h->d_uhot2.p;
  h->dev.uhot = h->d_uhot.p;
  h->dev.pscore = h->d_pscore.p;
  h->dev.sym_final = h->d_sym_final.p;
  h->dev.sym_len = h->d_sym_len.p;
  h->dev.byte_ids = h->d_byte_ids.p;
  h->dev.dec_info = h->d_dec_info.p;
  h->dev.dec_off = h->d_dec_off.p;
  h->dev.dec_bytes = h->d_dec_bytes.p;
  if (h->model.has_denormalizer) {
    const HostTables &d = h->dn_tables;
    HIP_OR_RETURN(h, Upload(&h->dn_ndarts, d.ndarts));
    HIP_OR_RETURN(h, Upload(&h->dn_nblob, d.nblob));
    HIP_OR_RETURN(h, Upload(&h->dn_npair, d.npair));
    HIP_OR_RETURN(h, Upload(&h->dn_utrie, d.utrie));
    h->dn_dev = d.scalars;
    h->dn_dev.ndarts = h->dn_ndarts.p;
    h->dn_dev.nblob = h->dn_nblob.p;
    h->dn_dev.npair = h->dn_npair.p;
    h->dn_dev.utrie = h->dn_utrie.p;
  }
  return kOk;
}

// After SetVocabulary / ResetVocabulary / SetEncodeExtraOptions: only the type-dependent words and the scalars
// change (the decode tables follow the piece types too: a BYTE piece turned UNUSED decodes as its literal text).
int RefreshDevice(spmx_handle *h, bool types_changed) {
  HostTables &t = h->tables;
  if (types_changed) {
    if (h->model.model_type == kUnigram) {
      HIP_OR_RETURN(h, Upload(&h->d_ptrie, t.ptrie));
      HIP_OR_RETURN(h, Upload(&h->d_cfirst, t.cfirst));
    } else {
      HIP_OR_RETURN(h, Upload(&h->d_sym_final, t.sym_final));
    }
    // the word memo follows the live piece types -- of BPE models too (tables.cc BuildWordMemo: an UNUSED piece
    // switches it off, ResetVocabulary switches it back on with full-size tables and masks)
    HIP_OR_RETURN(h, Upload(&h->d_umemo, t.umemo));
    HIP_OR_RETURN(h, Upload(&h->d_umemo16, t.umemo16));
    HIP_OR_RETURN(h, Upload(&h->d_uall, t.uall));
    HIP_OR_RETURN(h, Upload(&h->d_udisp, t.udisp));
    HIP_OR_RETURN(h, Upload(&h->d_uhot2, t.uhot2));
    HIP_OR_RETURN(h, Upload(&h->d_uhot, t.uhot));
    HIP_OR_RETURN(h, Upload(&h->d_pscore, t.pscore));
    HIP_OR_RETURN(h, Upload(&h->d_dec_info, t.dec_info));
    HIP_OR_RETURN(h, Upload(&h->d_dec_off, t.dec_off));
    HIP_OR_RETURN(h, Upload(&h->d_dec_bytes, t.dec_bytes));
  }
  SpmxDev d = t.scalars;
  d.ndarts = h->dev.ndarts; d.nblob = h->dev.nblob; d.npair = h->dev.npair; d.ptrie = h->d_ptrie.p; d.cfirst = t.cfirst.empty() ? nullptr : h->d_cfirst.p; d.plen = h->d_plen.p; d.utrie = h->dev.utrie;
  d.chartab = h->dev.chartab; d.pairtab = h->dev.pairtab; d.wordtab = h->dev.wordtab; d.sym_final = h->d_sym_final.p;
  d.sym_len = h->dev.sym_len; d.byte_ids = h->dev.byte_ids;
  d.umemo = h->d_umemo.p; d.umemo16 = h->d_umemo16.p; d.uall = h->d_uall.p; d.udisp = h->d_udisp.p; d.uhot2 = h->d_uhot2.p; d.uhot = h->d_uhot.p; d.pscore = h->d_pscore.p;
  d.dec_info = h->d_dec_info.p; d.dec_off = h->d_dec_off.p; d.dec_bytes = h->d_dec_bytes.p;
  h->dev = d;
  return kOk;
}

void DestroyHandle(spmx_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  h->d_ndarts.Free(); h->d_npair.Free(); h->d_sym_final.Free(); h->d_nblob.Free(); h->d_ptrie.Free(); h->d_chartab.Free();
  h->d_pairtab.Free(); h->d_wordtab.Free(); h->d_utrie.Free(); h->d_sym_len.Free(); h->d_byte_ids.Free();
  h->d_dec_info.Free(); h->d_dec_off.Free(); h->d_dec_bytes.Free(); h->d_plen.Free(); h->d_cfirst.Free(); h->d_umemo.Free(); h->d_umemo16.Free(); h->d_uall.Free(); h->d_udisp.Free(); h->d_uhot2.Free(); h->d_uhot.Free(); h->d_pscore.Free();
  h->dn_ndarts.Free(); h->dn_npair.Free(); h->dn_nblob.Free(); h->dn_utrie.Free();
  h->pool.clear();
  delete h;
}

// score-ring entries of the streaming unigram kernels for this handle's model
uint32_t HandleRing(const spmx_handle *h) {
  const uint32_t r = ScoreRing(h->tables.max_piece_len);
  return h->ring_override > r ? h->ring_override : r;
}

hipError_t EnsureEvents(Workspace *ws) {
  if (ws->ev_ready) return hipSuccess;
  for (auto &pair : ws->ev) {
    hipError_t e = hipEventCreate(&pair[0]);
    if (e == hipSuccess) e = hipEventCreate(&pair[1]);
    if (e != hipSuccess) return e;
  }
  ws->ev_ready = true;
  return hipSuccess;
}

// d_text / gen_lists (both or neither): the plain scan (kernels.h ClassifyArgs) -- sentences with a byte outside
// 0x20 .. 0x7E go to gen_lists, counted in Ctrl::gen_counts
int RunClassify(spmx_handle *h, Workspace *ws, const uint64_t *d_offsets, uint32_t n32, hipStream_t stream,
                const uint8_t *d_text = nullptr, uint64_t text_bytes = 0, uint32_t *gen_lists = nullptr) {
  ClassifyArgs ca{};
  ca.offs = d_offsets; ca.n = n32; ca.n_classes = static_cast<uint32_t>(kNumClasses);
  if (d_text && gen_lists) {
    HIP_OR_RETURN(h, hipMemsetAsync(ws->d_flags.p, 0, n32, stream));
    PlainScanArgs pa{d_text, text_bytes, d_offsets, n32, (h->dev.flags & kNfRemoveExtraWs) ? 0u : 1u, ws->d_flags.p};
    // a workgroup (four wavefronts) takes 16 KB per step; one step each up to 2^20 workgroups (16 GB of text)
    uint64_t g = (text_bytes + 16383) / 16384;
    if (g > (1u << 20)) g = 1u << 20;
    if (g < 1) g = 1;
    HIP_OR_RETURN(h, LaunchPlainScan(pa, static_cast<int>(g), stream));
    ca.flags = ws->d_flags.p; ca.lists2 = gen_lists; ca.list2_counts = ws->d_ctrl->gen_counts;
    ca.scan_max_rcap = kMaxStagedRaw;
  }
  for (int c = 0; c < kNumClasses; ++c) ca.rcap[c] = h->classes[c].rcap;
  ca.lists = ws->d_lists.p; ca.list_counts = ws->d_ctrl->list_counts;
  ca.key_totals = ws->d_ctrl->key_totals; ca.key_cursor = ws->d_ctrl->key_cursor;
  ca.sub_buckets = h->sub_buckets;
  const uint32_t chunks = (n32 + 64 * kClassifyChunk - 1) / (64 * kClassifyChunk);
  const uint32_t wide = static_cast<uint32_t>(h->n_cu) * 8u;
  HIP_OR_RETURN(h, LaunchClassify(ca, static_cast<int>(chunks < wide ? chunks : wide), stream));
  return kOk;
}

// ---- planning of one streaming launch ---------------------------------------------------------------------------
struct StreamPlan {
  int grid = 0, waves = 1;
  uint32_t lds = 0;
  uint64_t slab_bytes = 0;       // per wavefront
  uint32_t open = 0;             // non-empty classes
  bool split_only = false;       // every tile takes the split form: LaunchEncodeSplit
};

// Fills a->cls / n_classes / total_main / ring / slab_bytes for the classes [c_lo, c_hi) whose sizes are `counts`
// (classes outside the range get no tiles); tcap_of(c) gives a class's text-column capacity.
template <typename TcapFn>
StreamPlan PlanStream(const spmx_handle *h, EncodeArgs *a, const uint32_t *counts, int c_lo, int c_hi, int n_classes,
                      const uint32_t *rcaps, TcapFn tcap_of, bool all_general, int waves_cap = 0, int cus_cap = 0, int allow_split = 0) {
  StreamPlan sp;
  const int model = h->model.model_type;
  const uint32_t ring = HandleRing(h);
  a->ring = ring;
  // the short back-pointer form (kernels_stream.h BpShort): unigram, no user-defined pieces, ids that fit
  const bool bp_short = model == kUnigram && !(h->dev.flags & kNfHasUserDefined) &&
                        h->model.pieces.size() <= kBpShortMaxVocab && !h->no_bp_short;
  a->bp_short = bp_short ? 1u : 0u;
  const uint32_t bpsz = bp_short ? 2u : 4u;
  uint32_t priv = StreamPrivateBytes(model, ring, bpsz);
  // the split form (kernels_matchfold.h) for the classes of long sentences of a unigram model: its match phase keeps ONE
  // sentence's raw and normalized image in the wavefront's LDS
  // allow_split: 0 no class takes the split form; 1 the eligible ones do, beside lane-per-sentence tiles of other classes
  // (a tail launch); 2 EVERY class of the launch does (the caller passes only eligible ones): the wavefront's slice then
  // holds the match phase's image or the fold's character rings, not the lane form's byte rings -- 12 instead of 9 per CU
  const bool can_split = allow_split != 0 && model == kUnigram && h->tables.split_ok && !h->no_split && !(h->dev.flags & kNfHasUserDefined);
  auto split_class = [&](int c) { return can_split && counts[c] > 0 && rcaps[c] > h->split_min_raw && rcaps[c] <= kMfMaxRaw && tcap_of(c) < 65000u; };
  a->match_rows = static_cast<uint32_t>(h->tables.max_prefixes);
  a->split_ring = FoldRing(h->tables.max_piece_chars);
  if (allow_split == 2)                        // (only if every class with sentences is eligible: else as mode 1)
    for (int c = c_lo; c < c_hi; ++c) if (counts[c] > 0 && !split_class(c)) allow_split = 1;
  sp.split_only = allow_split == 2;
  if (allow_split == 2) priv = ((FoldLdsBytes(a->split_ring, bpsz) + 15u) & ~15u) + 256u;
  if (allow_split == 2 && priv < 16u * 64u * 4u + 256u) priv = 16u * 64u * 4u + 256u;      // (emit_stream_lane's staging column of 16 ids a lane)
  for (int c = c_lo; c < c_hi; ++c)
    if (split_class(c)) {
      const uint32_t need = ((MatchLdsBytes(rcaps[c], tcap_of(c), a->match_rows) + 15u) & ~15u) + 256u;
      if (need > priv) priv = need;
    }
  int waves = static_cast<int>((kLdsPerCu - kStreamSharedBytes) / priv);
  if (waves > 16) waves = 16;            // __launch_bounds__(1024)
  if (waves < 1) waves = 1;
  if (h->tile_waves_override > 0 && h->tile_waves_override < waves) waves = h->tile_waves_override;
  if (waves_cap > 0 && waves_cap < waves) waves = waves_cap;     // (a launch that shares the CUs with another kernel)
  uint64_t total = 0;
  for (int c = c_lo; c < c_hi; ++c) total += counts[c];
  uint64_t grid = static_cast<uint64_t>(h->n_cu - h->reserve_cus);
  // (cus_cap is NOT applied -- it never was, although round 5's notes say otherwise -- and round 6 measured that it should
  // not be: the general launch beside the word rounds takes a workgroup on every CU, the word-per-lane workgroups (162 KB of
  // LDS: nothing shares a CU with them) start as those end, and that order is the fast one.  On the Llama-style model
  // (12 % of the sentences set aside, a 13 ms general launch) real partitions cost more: 96 / 128 / 160 / 192 CUs for the
  // general launch -> 62.9 / 49.4 / 41.0 / 35.3 ms a step against 29.9 -- the lane-per-sentence BPE kernel runs 1.7 times
  // slower with the word rounds streaming beside it.  This is also what the "17 ms first round" of that model was: 4.7 ms
  // of work behind the 13 ms it waited for its CUs, both inside the events that time it; profiles/r06_llama_fork_cus.txt)
  (void)cus_cap;
  if (grid * waves > total) grid = (total + waves - 1) / waves;
  if (grid < 1) grid = 1;
  // the slab of a wavefront must hold one lane of the largest class present: fewer wavefronts if the limit says so
  uint64_t need1 = 0;
  auto slab_of = [&](int c, uint32_t sh) -> uint64_t {
    if (!split_class(c)) return StreamSlabBytes(tcap_of(c), ring, sh, bpsz);
    return StreamSplitBase(tcap_of(c), ring, sh, bpsz) + (MatchStreamBytes(MatchStreamCap(tcap_of(c), h->split_per_byte)) << sh);
  };
  for (int c = c_lo; c < c_hi; ++c)
    if (counts[c]) { const uint64_t b = slab_of(c, 0); if (b > need1) need1 = b; }
  if (need1 && grid * waves * need1 > h->stream_scratch_limit) {
    uint64_t w = h->stream_scratch_limit / need1;
    if (w < 1) w = 1;
    if (w < static_cast<uint64_t>(waves)) { waves = static_cast<int>(w); grid = 1; }
    else grid = w / waves;
  }
  uint64_t n_waves = grid * waves;
  // A THIN launch (fewer full tiles than wavefronts) takes full tiles: a wavefront's time is the sum over its tiles of
  // the longest sentence in each, so a tile per class per wavefront (what spreading every class over every wavefront
  // gives) costs the sum of the classes' longest sentences, one full tile per wavefront only the longest one's.
  uint64_t full_tiles = 0;
  for (int c = c_lo; c < c_hi; ++c) full_tiles += (static_cast<uint64_t>(counts[c]) + 63) / 64;
  const bool thin = full_tiles <= n_waves;
  // no more wavefronts than tiles: a first pass at full tiles tells how many there can be
  {
    uint64_t tiles = 0;
    for (int c = c_lo; c < c_hi; ++c) {
      if (!counts[c]) continue;
      uint64_t tw = (static_cast<uint64_t>(counts[c]) + n_waves - 1) / n_waves;
      if (tw > 64 || thin) tw = 64;
      if (split_class(c)) { tw = (static_cast<uint64_t>(counts[c]) + h->split_tiles * n_waves - 1) / (h->split_tiles * n_waves); if (tw > 64) tw = 64; if (tw < 1) tw = 1; }   // (below)
      tiles += (static_cast<uint64_t>(counts[c]) + tw - 1) / tw;
    }
    if (tiles < n_waves) {
      if (grid > 1) { grid = (tiles + waves - 1) / waves; if (grid < 1) grid = 1; }
      if (grid == 1 && tiles < static_cast<uint64_t>(waves)) waves = static_cast<int>(tiles < 1 ? 1 : tiles);
      n_waves = grid * waves;
    }
  }
  const uint64_t budget = h->stream_scratch_limit / n_waves;
  a->n_classes = static_cast<uint32_t>(n_classes);
  uint32_t tile_base = 0;
  for (int c = n_classes - 1; c >= 0; --c) {          // tiles are handed out longest class first
    StreamClass &sc = a->cls[c];
    sc = StreamClass{};
    sc.rcap = rcaps[c];
    if (c < c_lo || c >= c_hi || counts[c] == 0) continue;
    sc.tcap = tcap_of(c);
    uint64_t tw = (static_cast<uint64_t>(counts[c]) + n_waves - 1) / n_waves;   // sentences per main tile
    if (thin) tw = 64;
    // (a split tile's match phase takes its sentences ONE AFTER ANOTHER: small tiles, two or three to a wavefront, so that
    // the queue evens the wavefronts out -- a "thin" launch of 64-sentence tiles left 60 % of them idle, round 6)
    if (split_class(c)) tw = (static_cast<uint64_t>(counts[c]) + h->split_tiles * n_waves - 1) / (h->split_tiles * n_waves);
    if (tw < h->tile_min_lanes) tw = h->tile_min_lanes;
    if (tw > 64) tw = 64;
    if (tw < 1) tw = 1;
    uint32_t sh = 0;                                   // lanes of a tile: enough for tw, as many as the budget allows
    while ((1ull << sh) < tw) ++sh;
    while (sh > 0 && slab_of(c, sh) > budget) --sh;
    if (tw > (1ull << sh)) tw = 1ull << sh;
    sc.lane_shift = sh;
    sc.split = split_class(c) ? 1u : 0u;
    sc.ccap = sc.split ? MatchStreamCap(sc.tcap, h->split_per_byte) : 0u;
    const uint64_t slab = slab_of(c, sh);
#ifdef SPMX_TEST_SEAMS
    if (getenv("SPMX_DEBUG_PLAN")) fprintf(stderr, "spmx plan: class %d rcap %u count %u tw %llu lanes %u split %u ccap %u slab %llu priv %u waves %d\n", c, sc.rcap, counts[c], (unsigned long long)tw, 1u << sh, sc.split, sc.ccap, (unsigned long long)slab, priv, waves);
#endif
    if (slab > sp.slab_bytes) sp.slab_bytes = slab;
    sc.count = counts[c];
    sc.tw = static_cast<uint32_t>(tw);
    sc.main_tiles = static_cast<uint32_t>((static_cast<uint64_t>(counts[c]) + tw - 1) / tw);
    sc.tile_base = tile_base;
    tile_base += sc.main_tiles;
    sc.general = all_general ? 1u : 0u;
    // short classes: a stray non-ASCII sentence would hold 63 ASCII lanes up, so a tile needs 16 of them to keep
    // them; longer classes: 4; tiles of a few lanes keep everything
    sc.min_lanes = h->lane_general_min_lanes ? h->lane_general_min_lanes : (sh < 6 ? 0u : (sc.rcap <= 576 ? 16u : 4u));
    ++sp.open;
  }
  a->total_main = tile_base;
  sp.slab_bytes = (sp.slab_bytes + 255u) & ~static_cast<uint64_t>(255);
  a->slab_bytes = sp.slab_bytes;
  sp.grid = static_cast<int>(grid);
  sp.waves = waves;
  sp.lds = kStreamSharedBytes + static_cast<uint32_t>(waves) * priv;
  a->private_bytes = priv;
  return sp;
}

// ---- the encode launch sequence ----------------------------------------------------------------------------------
// d_begin / d_end (both or neither): the spans form (kernels_align.h).  d_status (optional): n status bytes.
int EncodeDevice(spmx_handle *h, Workspace *ws, const uint8_t *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                 uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets, uint8_t *d_status,
                 hipStream_t stream, uint64_t *total_ids, uint64_t *n_failed, uint32_t *d_begin = nullptr,
                 uint32_t *d_end = nullptr, uint32_t *d_nbegin = nullptr, uint32_t *d_nend = nullptr) {
  const bool spans = d_begin != nullptr && d_end != nullptr;
  if (total_ids) *total_ids = 0;
  if (n_failed) *n_failed = 0;
  if (n >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "more than 2^32 - 64 sentences in one batch");
  if (!d_offsets || !d_id_offsets) return Fail(h, kInvalidArgument, "null offsets");
  if (n == 0) {
    HIP_OR_RETURN(h, hipMemsetAsync(d_id_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  const LengthClass *cls = h->classes;
  const int ncls = kNumClasses;
  // lists: kMaxClasses class lists (the last one is the overflow list), kMaxClasses escalation lists of the align
  // kernels, the long list, two retry lists
  // the word kernel (kernels_word.h): unigram models whose pieces never reach across a word boundary; not for the spans
  // form (it records no token begins) nor under the `reverse` option
  bool word_ok = (h->dev.flags & kNfUniWordwise) && !(h->dev.flags & kNfReverse) && !spans && !h->no_word &&
                 !(h->model.model_type == kBpe && (ws->bpe_dropout > 0.f || h->no_stream));
  // The word kernels pay on text that is mostly plain ASCII words.  On other text (CJK, byte soup) their rounds only
  // cost: a handle remembers when a batch went almost entirely to the general kernels and leaves the word rounds out
  // of its next few calls, then tries again (results are the same either way).
  if (word_ok && !h->force_word_dp && h->word_backoff.load(std::memory_order_relaxed) > 0) {
    h->word_backoff.fetch_sub(1, std::memory_order_relaxed);
    word_ok = false;
  }
  const bool is_bpe = h->model.model_type == kBpe;
  // streaming (lane-per-sentence) kernels: every unigram model; BPE models that can be segmented word by word
  const bool dropout = is_bpe && ws->bpe_dropout > 0.f;      // BPE-dropout: every sentence takes the long form
  const bool bpe_stream = is_bpe && (h->dev.flags & kNfBpeWordwise) && !(h->dev.flags & kNfHasUnused) && !h->no_stream && !dropout;
  const bool streaming = !is_bpe || bpe_stream;
  // With the word rounds, classify also reads the text once and sets the sentences that are not plain ASCII aside
  // (kernels.h ClassifyArgs): the general launch over them runs NEXT TO the first word round, not after it.
  // A model whose words normalize by themselves (dev.h kNfWordLocalNorm) needs no scan: the word-per-lane rounds take the
  // words that are not plain ASCII through the call-local memo, and the sentences stay with them.
  const bool any_word = (h->dev.flags & kNfWordLocalNorm) && (h->word_form & 1) && !h->no_word_dyn;
  const bool scanned = word_ok && streaming && !h->no_scan && !any_word;
  HIP_OR_RETURN(h, ws->d_lists.Reserve(static_cast<size_t>(2 * kMaxClasses + 3 + (word_ok ? 3 * kMaxClasses : 0) + (scanned ? kMaxClasses : 0)) * n));
  if (scanned) HIP_OR_RETURN(h, ws->d_flags.Reserve(n));
  HIP_OR_RETURN(h, ws->d_counts.Reserve(n + 1));
  HIP_OR_RETURN(h, ws->d_tmp_off.Reserve(n + 1));
  HIP_OR_RETURN(h, ws->d_tile_sums.Reserve((n + kScanTile - 1) / kScanTile + 2));
  HIP_OR_RETURN(h, ws->d_big_list.Reserve((n + 63) / 64 + 1));
  if (!d_status) { HIP_OR_RETURN(h, ws->d_sent_status.Reserve(n)); d_status = ws->d_sent_status.p; }
  uint32_t *const class_lists = ws->d_lists.p;
  uint32_t *const over_list = class_lists + static_cast<size_t>(kMaxClasses - 1) * n;
  uint32_t *const hard_lists = class_lists + static_cast<size_t>(kMaxClasses) * n;   // (escalation lists of the align kernels)
  uint32_t *const long_list = class_lists + static_cast<size_t>(2 * kMaxClasses) * n;
  uint32_t *const retry_lists[2] = {long_list + n, long_list + 2 * n};
  uint32_t *const left_lists[3] = {long_list + 3 * n, long_list + (3 + static_cast<size_t>(kMaxClasses)) * n,
                                   long_list + (3 + 2 * static_cast<size_t>(kMaxClasses)) * n};   // (word_ok only)
  uint32_t *const gen_lists = long_list + (3 + 3 * static_cast<size_t>(kMaxClasses)) * n;           // (scanned only)
  // ids are at most one per normalized byte; the streaming kernels reserve a sentence's slot by that bound
  uint64_t expand = (h->dev.flags & kNfCompressSp) || !(h->dev.flags & kNfEscapeWs) ? 1 : 3;
  if ((h->dev.flags & kNfCompressSp) && (h->dev.flags & kNfByteFallback)) expand = 2;   // slots: bytes + 2 per space symbol
  // (a slot per sentence: its normalized length + the extra ids, rounded to groups of 4 with 3 ids of slack for alignment)
  uint64_t arena_need = expand * text_bytes + (10 + static_cast<uint64_t>(h->dev.n_prefix + h->dev.n_suffix)) * n + 4096;
  const bool prof = h->profiling;
  if (prof) HIP_OR_RETURN(h, EnsureEvents(ws));
  const uint32_t n32 = static_cast<uint32_t>(n);
  const int wide = h->n_cu * 8;
  const bool fast_ok = StreamFastEligible(h->dev.flags) && !h->no_fast;
  const bool uds = (h->dev.flags & kNfHasUserDefined) != 0;
  uint32_t rcaps[kMaxClasses] = {0};
  for (int c = 0; c < ncls; ++c) rcaps[c] = cls[c].rcap;
  rcaps[ncls - 1] = cls[ncls - 1].rcap;
  // the largest raw sentence any launch can take: its normalized form must stay below 2^31 bytes
  const uint64_t max_raw = (0x7FFFFF00ull - 16) / h->dev.expand_max;
  auto record = [&](int slot, int which) -> hipError_t {
    return prof ? hipEventRecord(ws->ev[slot][which], stream) : hipSuccess;
  };
  auto scan_compact = [&]() -> int {
    ScanArgs sa{ws->d_counts.p, n32, ws->d_tile_sums.p, d_id_offsets};
    const uint32_t tiles = (n32 + kScanTile - 1) / kScanTile;
    HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(wide) ? tiles : wide), stream));
    // After an arena overflow some kernels (the long form, the sentence-per-wave BPE) leave the range they asked for in
    // tmp_off / counts, beyond the arena's end -- compacting that would read past the allocation (seen as a memory fault on
    // the GPU, found again under ASAN on the emulator).  The compaction looks at the status word ITSELF (round 6: the host
    // used to read it back first -- one more synchronisation per call); the caller re-runs the batch with the arena
    // arena_head asks for.
    // (blocks of documents go to a second launch, by the whole chip; a batch of ten million short sentences has none and
    // does not pay for that launch)
    const uint32_t big = (n < 65536 || text_bytes > 160ull * n) ? h->compact_big : 0u;
    CompactArgs pa{ws->d_arena.p, ws->d_tmp_off.p, ws->d_counts.p, d_id_offsets, d_ids, d_ids ? ids_capacity : 0, n32, &ws->d_ctrl->status, h->compact_staged,
                   big, ws->d_big_list.p, &ws->d_ctrl->big_count[0]};
    const uint64_t cblocks = (n + 63) / 64;
    const uint64_t cgrid = cblocks < static_cast<uint64_t>(h->n_cu) * 32 ? cblocks : static_cast<uint64_t>(h->n_cu) * 32;
    HIP_OR_RETURN(h, LaunchCompact(pa, static_cast<int>(cgrid), stream));
    if (big) HIP_OR_RETURN(h, LaunchCompactBig(pa, h->n_cu * 4, stream));   // (returns at once when CompactKernel listed none)
    HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl, ws->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->total_ids, d_id_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));      // (the call returns with its outputs complete)
    return kOk;
  };
  // SPMX_ARENA_FIRST=<ids>: the first attempt's arena is no larger than this (tests force the overflow-and-retry path)
  uint64_t arena_cap_limit = 0;
  if (h->arena_first && arena_need > h->arena_first) { arena_need = h->arena_first; arena_cap_limit = h->arena_first; }
  // (the pipelined host form's largest chunk: Workspace::reserve_text_bytes -- capacity only, the call's own need decides the rest)
  const uint64_t arena_reserve = (h->arena_first || ws->reserve_text_bytes <= text_bytes) ? 0 :
      expand * ws->reserve_text_bytes + (10 + static_cast<uint64_t>(h->dev.n_prefix + h->dev.n_suffix)) * (ws->reserve_n > n ? ws->reserve_n : n) + 4096;
  for (int attempt = 0; attempt < 4; ++attempt) {
    HIP_OR_RETURN(h, ws->d_arena.Reserve(arena_need > arena_reserve ? arena_need : arena_reserve));
    if (spans) HIP_OR_RETURN(h, ws->d_arena_tb.Reserve(ws->d_arena.cap));
    if (prof) HIP_OR_RETURN(h, hipEventRecord(ws->ev[kNumSlots][0], stream));
    HIP_OR_RETURN(h, hipMemsetAsync(ws->d_ctrl, 0, sizeof(Ctrl), stream));
    HIP_OR_RETURN(h, hipMemsetAsync(d_status, 0, n, stream));
    for (bool &u : ws->slot_used) u = false;
    // DIRECT: both word rounds in the word-per-lane form over a model whose words normalize by themselves -- the first
    // round takes the sentences in input order, 64 to a tile, and finds a sentence's length class itself where it matters
    // (kernels.h EncodeArgs::direct): no classify pass, no read-back of its counts
    const bool direct = word_ok && any_word && h->word_form == 3 && !h->no_direct && streaming;
    uint32_t known[kMaxClasses] = {0};       // the class lists: every sentence, or (scanned) the plain ones
    uint32_t gen_known[kMaxClasses] = {0};   // (scanned) the sentences set aside for the general kernels
    uint64_t gen_total = 0;
    if (!direct) {
      if (int rc = RunClassify(h, ws, d_offsets, n32, stream, scanned ? d_text : nullptr, text_bytes, scanned ? gen_lists : nullptr); rc != kOk) return rc;
      HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl->list_counts, ws->d_ctrl->list_counts,
                                      sizeof(ws->h_ctrl->list_counts) + sizeof(ws->h_ctrl->gen_counts),       // (gen_counts follows list_counts in Ctrl)
                                      hipMemcpyDeviceToHost, stream));
      HIP_OR_RETURN(h, hipStreamSynchronize(stream));
      for (int c = 0; c < ncls; ++c) { known[c] = ws->h_ctrl->list_counts[c]; gen_known[c] = ws->h_ctrl->gen_counts[c]; gen_total += gen_known[c]; }
    } else {
      known[0] = n32;                        // (one run of tiles; the classes are the kernel's business)
    }
    // what every encode launch shares
    EncodeArgs a{};
    a.dev = h->dev; a.text = d_text; a.offs = d_offsets;
    a.arena = ws->d_arena.p; a.arena_head = &ws->d_ctrl->arena_head;
    a.arena_cap = attempt == 0 && arena_cap_limit && arena_cap_limit < ws->d_arena.cap ? arena_cap_limit : ws->d_arena.cap;
    a.tmp_off = ws->d_tmp_off.p; a.counts = ws->d_counts.p; a.sent_status = d_status; a.status = &ws->d_ctrl->status;
    a.arena_tb = spans ? ws->d_arena_tb.p : nullptr;
    a.long_list = long_list; a.side = &ws->d_ctrl->side;
    a.lists = class_lists; a.over_list = over_list;
    a.n = n32;
    a.fast_ok = fast_ok ? 1u : 0u;
    a.no_lane_general = h->no_lane_general ? 1u : 0u;
    a.no_char_norm = h->char_norm_mode;
    // one streaming launch over the classes [c_lo, c_hi) (or, exact: over the overflow list with exact capacities)
    int stream_waves_cap = 0;            // (set while a launch has to share the CUs with the second word round)
    int stream_cus_cap = 0;
    // smode: PlanStream's allow_split -- 1: eligible classes take the split form beside lane tiles; 2: a launch of split tiles only
    auto stream_launch = [&](int slot, int qi, int c_lo, int c_hi, const uint32_t *counts, bool exact, uint64_t exact_raw, int smode = 1) -> int {
      EncodeArgs la = a;
      uint32_t rc2[kMaxClasses];
      for (int c = 0; c < kMaxClasses; ++c) rc2[c] = rcaps[c];
      const bool esc3 = (h->dev.flags & kNfEscapeWs) && !(h->dev.flags & kNfCompressSp);
      StreamPlan sp;
      if (exact) {
        rc2[kMaxClasses - 1] = static_cast<uint32_t>(exact_raw < max_raw ? exact_raw : max_raw);
        const uint32_t tc = static_cast<uint32_t>(static_cast<uint64_t>(rc2[kMaxClasses - 1]) * h->dev.expand_max + 16);
        sp = PlanStream(h, &la, counts, kMaxClasses - 1, kMaxClasses, kMaxClasses, rc2, [&](int) { return tc; }, true, stream_waves_cap);
        la.over_list = nullptr;
        la.lists = class_lists;            // (the overflow list is the last of the classify lists, whatever `a.lists` names)
      } else {
        // the last class of the table takes every longer sentence too: those go straight to the overflow list
        sp = PlanStream(h, &la, counts, c_lo, c_hi, ncls, rc2,
                        [&](int c) { return esc3 ? 2u * cls[c].rcap + 64u : (h->wide_tcap ? cls[c].ncap : cls[c].rcap + cls[c].rcap / 4u + 16u); }, !fast_ok,
                        stream_waves_cap, stream_cus_cap, /*allow_split=*/smode);
      }
      if (la.total_main == 0) return kOk;
      la.q = &ws->d_ctrl->q[qi];
      la.stats = &ws->d_ctrl->stats[kStatsPerClass * slot];
      const uint64_t slab_total = static_cast<uint64_t>(sp.grid) * sp.waves * sp.slab_bytes;
      HIP_OR_RETURN(h, ws->d_slab.Reserve(slab_total));
      la.slab = ws->d_slab.p;
      snprintf(ws->slot_name[slot], sizeof(ws->slot_name[slot]), "%s", is_bpe ? "EncodeBpeStreamKernel" : la.bp_short ? (la.ring == 16 ? "EncodeStreamShortKernel<16>" : "EncodeStreamShortKernel<0>")
               : (la.ring == 16 ? (uds ? "EncodeStreamKernel<16, true>" : "EncodeStreamKernel<16, false>")
                                : (uds ? "EncodeStreamKernel<0, true>" : "EncodeStreamKernel<0, false>")));
      HIP_OR_RETURN(h, record(slot, 0));
      if (sp.split_only) {
        snprintf(ws->slot_name[slot], sizeof(ws->slot_name[slot]), "%s", la.bp_short ? "EncodeSplitShortKernel" : "EncodeSplitKernel");
        HIP_OR_RETURN(h, LaunchEncodeSplit(la, sp.grid, sp.waves, sp.lds, stream));
      } else {
        HIP_OR_RETURN(h, LaunchEncodeStream(h->model.model_type, uds, la, sp.grid, sp.waves, sp.lds, stream));
      }
      HIP_OR_RETURN(h, record(slot, 1));
      ws->slot_used[slot] = true;
      return kOk;
    };
    // which classes a launch of split tiles takes (PlanStream decides again, from the same rules)
    auto split_launch_class = [&](int c) -> bool {
      return !is_bpe && h->tables.split_ok && !h->no_split && !uds && cls[c].rcap > h->split_min_raw && cls[c].rcap <= kMfMaxRaw &&
             !((h->dev.flags & kNfEscapeWs) && !(h->dev.flags & kNfCompressSp) && 2u * cls[c].rcap + 64u >= 65000u);
    };
    // the long form over one device-side list (BPE), growing the slice pool until every sentence has had its turn
    // (uni: the wave-cooperative unigram form over the same pool, kernels_uniwave.h)
    auto long_launch = [&](const uint32_t *list, const uint32_t *d_count, uint32_t count, bool uni = false) -> int {
      if (count == 0) return kOk;
      LongArgs la{};
      la.dev = h->dev; if (h->uw_exact) la.dev.uw_f32_limit = 0.f; la.text = d_text; la.offs = d_offsets;
      la.arena = ws->d_arena.p; la.arena_head = &ws->d_ctrl->arena_head; la.arena_cap = a.arena_cap;
      la.tmp_off = ws->d_tmp_off.p; la.counts = ws->d_counts.p; la.sent_status = d_status; la.status = &ws->d_ctrl->status;
      la.side = &ws->d_ctrl->side; la.arena_tb = spans ? ws->d_arena_tb.p : nullptr;
      la.stack_cap = static_cast<uint32_t>(h->tables.max_piece_len) + 8u;
      la.dropout = ws->bpe_dropout; la.seed = ws->sample_seed;
      la.stats = uni ? &ws->d_ctrl->stats[kStatsPerClass * kSlotLong] : nullptr;
      la.pool_head = &ws->d_ctrl->pool_head;
      uint64_t want = 96ull * text_bytes / (n > count ? n / count : 1) + 4096ull * count + (1ull << 20);
      if (uni) want = 10ull * text_bytes / (n > count ? n / count : 1) + 1024ull * count + (1ull << 20);
      if (want > (4ull << 30)) want = 4ull << 30;
      int turn = 0;
      uint32_t left = count;
      for (int round = 0; round < 40 && left > 0; ++round) {
        HIP_OR_RETURN(h, ws->d_pool.Reserve(want));
        la.pool = ws->d_pool.p; la.pool_cap = ws->d_pool.cap;
        la.list = list; la.list_count = d_count;
        la.retry_list = retry_lists[turn]; la.retry_count = &ws->d_ctrl->retry_count[turn];
        HIP_OR_RETURN(h, hipMemsetAsync(&ws->d_ctrl->pool_head, 0, sizeof(unsigned long long), stream));
        HIP_OR_RETURN(h, hipMemsetAsync(&ws->d_ctrl->retry_count[turn], 0, sizeof(uint32_t), stream));
        uint64_t g = uni ? static_cast<uint64_t>(left) : (static_cast<uint64_t>(left) + 63) / 64;   // a sentence per wave / per lane
        if (g > static_cast<uint64_t>(h->n_cu) * 16) g = static_cast<uint64_t>(h->n_cu) * 16;
        snprintf(ws->slot_name[kSlotLong], sizeof(ws->slot_name[kSlotLong]), uni ? "UniLongKernel" : "BpeLongKernel");
        if (!ws->slot_used[kSlotLong]) HIP_OR_RETURN(h, record(kSlotLong, 0));
        // few, long documents: a workgroup of two wavefronts each (a walker and a folder side by side); many: a wavefront each
        const bool pipe = uni && h->uw_pipe != 0 && (h->uw_pipe == 2 || (left <= static_cast<uint32_t>(h->n_cu) * 4u && text_bytes / n >= 4096u));
        if (pipe) snprintf(ws->slot_name[kSlotLong], sizeof(ws->slot_name[kSlotLong]), "UniLongPipeKernel");
        if (pipe) HIP_OR_RETURN(h, LaunchUniLongPipe(la, UniWaveRow(h->tables.max_piece_len), static_cast<int>(left < static_cast<uint32_t>(h->n_cu) * 8u ? left : static_cast<uint32_t>(h->n_cu) * 8u), stream));
        else if (uni) HIP_OR_RETURN(h, LaunchUniLong(la, UniWaveRow(h->tables.max_piece_len), static_cast<int>(g), stream));
        else HIP_OR_RETURN(h, LaunchBpeLong(la, static_cast<int>(g), stream));
        HIP_OR_RETURN(h, record(kSlotLong, 1));
        ws->slot_used[kSlotLong] = true;
        HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->pool_head, &ws->d_ctrl->pool_head, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->retry_count[turn], &ws->d_ctrl->retry_count[turn], sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_OR_RETURN(h, hipStreamSynchronize(stream));
        const uint32_t again = ws->h_ctrl->retry_count[turn];
        if (again == 0) return kOk;
        // the pool was too small for `again` sentences; pool_head is what all of this round's asked for
        want = ws->h_ctrl->pool_head + (1ull << 20);
        list = retry_lists[turn]; d_count = &ws->d_ctrl->retry_count[turn];
        left = again;
        turn ^= 1;
      }
      return left ? Fail(h, kResourceExhausted, "the long form's slice pool kept overflowing") : kOk;
    };
    const uint32_t *d_list_counts = ws->d_ctrl->list_counts;      // the device-side counts of a.lists
    uint32_t again_counts[kMaxClasses] = {0};                     // (call-local memo) the second word round's input, per class
    uint64_t again_total = 0;
    uint32_t again_words = 0;
    std::function<int(int, int, int, uint32_t *, uint32_t *, uint32_t *, uint32_t *)> word_pass;
    // Which classes the wave-cooperative unigram form takes (kernels_uniwave.h, a sentence per wavefront): documents --
    // the classes beyond the main streaming launch's (16 KiB) always; the classes between 4 and 16 KiB when they are
    // most of the batch (a batch OF documents: one lane per document would leave the chip idle), not when they are a
    // tail of a batch of sentences (there they hide behind the main launch's other tiles).
    const bool uni_wave = !is_bpe && !spans && h->tables.max_prefixes >= 1 &&
                          h->tables.max_piece_len <= static_cast<int>(kUwMaxPiece) && !h->no_uni_wave;
    bool uni_class[kMaxClasses] = {false};
    if (uni_wave) {
      uint64_t vol_staged = 0, vol_mid = 0;
      for (int c = 0; c < ncls; ++c) {
        const uint64_t v = (static_cast<uint64_t>(known[c]) + gen_known[c]) * cls[c].rcap;
        if (cls[c].rcap <= kMaxStagedRaw) vol_staged += v;
        else if (cls[c].rcap <= h->main_max_raw) vol_mid += v;
      }
      for (int c = 0; c < ncls; ++c) {
        if (cls[c].rcap > h->main_max_raw) uni_class[c] = true;
        else if (cls[c].rcap > kMaxStagedRaw) uni_class[c] = vol_mid > vol_staged;
        else uni_class[c] = known[c] + gen_known[c] > 0 && known[c] + gen_known[c] < h->uni_wave_max;
      }
    }
    // ---- the general kernels over one set of class lists (`cnt` sentences per class; cnt is consumed) ----
    // tail: what the word rounds left (thin lists: a sentence per wavefront where the model allows it; its streaming launch
    // has a tile queue of its own, the early launches' may still be running on the second stream)
    int tail_qi = 5;                        // the tile queue of a tail launch (a second tail launch of the call takes another: a queue is used once)
    auto general_pass = [&](const uint32_t *lists, const uint32_t *d_counts, uint32_t *cnt, bool tail) -> int {
      a.lists = lists;
      d_list_counts = d_counts;
      uint64_t total = 0;
      for (int c = 0; c < ncls; ++c) total += cnt[c];
      if (total == 0) return kOk;
      if (streaming) {
        // unigram: the wave-cooperative form (a sentence per wavefront, kernels_uniwave.h) takes the classes where the
        // lane-per-sentence kernels are the wrong tool -- documents (one lane would walk them alone: 2.5 us per byte),
        // and classes with too few sentences to fill 64 lanes of every wavefront
        if (uni_wave) {
          // A tail launch (what the word rounds handed on) is bound by the LATENCY of a lane-per-sentence tile's longest
          // sentence -- about 4.5 us per byte of the largest class present, whatever the count (C2's classes: 2.3 ms) -- while a
          // sentence per wavefront costs by the VOLUME: about 0.27 ms per MB of class capacity over the ~4000 wavefronts in
          // flight (open-vocabulary text: 50 k sentences of C2's lengths in 0.7 ms against 2.7; botchan's short lines: 3.2 ms
          // against 0.4 -- gpurun_out r05v / r05w).  The cheaper estimate wins; documents of a tail (no main launch to hide
          // behind) always take the wavefront form.
          uint64_t tail_vol = 0, rcap_max = 0;
          for (int c = 0; c < ncls; ++c) {
            tail_vol += static_cast<uint64_t>(cnt[c]) * cls[c].rcap;
            if (cnt[c] && cls[c].rcap <= kMaxStagedRaw && cls[c].rcap > rcap_max) rcap_max = cls[c].rcap;
          }
          const double est_wave_ms = 2.7e-7 * static_cast<double>(tail_vol), est_lane_ms = 4.5e-3 * static_cast<double>(rcap_max);
          const bool few = tail && (total < 4096 || est_wave_ms < est_lane_ms);
          for (int c = 0; c < ncls; ++c) {
            if (cnt[c] == 0 || !(uni_class[c] || few || (tail && cls[c].rcap > kMaxStagedRaw))) continue;
            if (int rc = long_launch(lists + static_cast<size_t>(c) * n, &d_counts[c], cnt[c], true); rc != kOk) return rc;
            cnt[c] = 0;
          }
        }
        if (tail) return stream_launch(kSlotDoc, tail_qi, 0, ncls, cnt, false, 0);
        int c_doc = ncls;                    // first class of the document launch
        for (int c = 0; c < ncls; ++c) if (cls[c].rcap > h->main_max_raw) { c_doc = c; break; }
        // The long classes of a unigram model the split form takes in a launch of their OWN (EncodeSplitKernel: 12 wavefronts
        // per CU instead of 10): measured and not the default -- 8.8 ms for the long classes + 6.0 ms for the short ones
        // (latency-bound, half idle) against 11.2 ms for both in one launch, where the short classes' lane tiles fill
        // the gaps of the long classes' split tiles (SPMX_SPLIT_LAUNCH=1; C5, profiles/README_r06.md)
        if (h->split_own_launch) {
          // (copies: the caller's counts also say which classes the spans form's align launches visit)
          uint32_t cnt_split[kMaxClasses] = {0}, cnt_main[kMaxClasses] = {0};
          bool any_split = false;
          for (int c = 0; c < ncls; ++c) {
            if (c < c_doc && cnt[c] && split_launch_class(c)) { cnt_split[c] = cnt[c]; any_split = true; }
            else cnt_main[c] = cnt[c];
          }
          if (any_split) {
            if (int rc = stream_launch(kSlotSplit, 6, 0, c_doc, cnt_split, false, 0, 2); rc != kOk) return rc;
            if (int rc = stream_launch(kSlotMain, 0, 0, c_doc, cnt_main, false, 0); rc != kOk) return rc;
            return stream_launch(kSlotDoc, 1, c_doc, ncls, cnt, false, 0);
          }
        }
        if (int rc = stream_launch(kSlotMain, 0, 0, c_doc, cnt, false, 0); rc != kOk) return rc;
        return stream_launch(kSlotDoc, 1, c_doc, ncls, cnt, false, 0);
      }
      // BPE, sentence-per-wave form (models that are not word-wise, or with UNUSED pieces; never after word rounds): the
      // staged classes; a sentence whose normalized form overflows its class escalates to the next staged one, then to
      // the long form
      int c_staged = 0;
      while (c_staged < ncls && cls[c_staged].rcap <= kMaxStagedRaw && !h->no_wave && !dropout) ++c_staged;
      bool first = true;
      for (int c = 0; c < c_staged; ++c) {
        EncodeArgs la = a;
        la.list = class_lists + static_cast<size_t>(c) * n; la.list_count = &ws->d_ctrl->list_counts[c];
        const bool has_next = c + 1 < c_staged;
        la.next_list = has_next ? class_lists + static_cast<size_t>(c + 1) * n : nullptr;
        la.next_count = has_next ? &ws->d_ctrl->list_counts[c + 1] : nullptr;
        la.rcap = cls[c].rcap; la.ncap = cls[c].ncap;
        if (la.rcap == kMaxStagedRaw && la.ncap > kBpeWaveNcap3) la.ncap = kBpeWaveNcap3;
        la.stats = &ws->d_ctrl->stats[kStatsPerClass * kSlotWave];
        const uint32_t lds = EncodeLdsBytes(kBpe, la.rcap, la.ncap);
        int per_cu = static_cast<int>(kLdsPerCu / lds);
        if (per_cu > 32) per_cu = 32;
        if (per_cu < 1) per_cu = 1;
        uint64_t grid = static_cast<uint64_t>(h->n_cu) * per_cu;
        if (grid > n) grid = n;
        snprintf(ws->slot_name[kSlotWave], sizeof(ws->slot_name[kSlotWave]), "EncodeKernel<2, *>");
        if (first) HIP_OR_RETURN(h, record(kSlotWave, 0));
        first = false;
        HIP_OR_RETURN(h, LaunchEncode(kBpe, c, la, static_cast<int>(grid), lds, stream));
        HIP_OR_RETURN(h, record(kSlotWave, 1));
        ws->slot_used[kSlotWave] = true;
      }
      for (int c = c_staged; c < ncls; ++c)
        if (int rc = long_launch(class_lists + static_cast<size_t>(c) * n, &ws->d_ctrl->list_counts[c], cnt[c]); rc != kOk) return rc;
      return kOk;
    };
    // ---- fork (scanned): the general kernels over the sentences classify set aside go to the workspace's second stream
    // and run NEXT TO the first word round -- they work on disjoint sentences.  The general launch is bound by the latency
    // of its longest sentences, not by the chip: it is held to a few wavefronts per workgroup (SPMX_FORK_WAVES), so that
    // its workgroups and the word kernel's fit a CU together.  Plain DOCUMENTS never take the word kernels of a unigram
    // model (a sentence per wavefront is their form): they start here too. ----
    hipStream_t main_stream = stream;
    bool forked = false;
    if (scanned) {
      uint32_t doc_known[kMaxClasses] = {0};
      uint64_t doc_total = 0;
      if (uni_wave)
        for (int c = 0; c < ncls; ++c)
          if (known[c] && ((uni_class[c] && cls[c].rcap > kMaxStagedRaw) || cls[c].rcap > h->main_max_raw)) { doc_known[c] = known[c]; doc_total += known[c]; known[c] = 0; }
      if (gen_total + doc_total > 0) {
        uint64_t plain_total = 0;
        for (int c = 0; c < ncls; ++c) plain_total += known[c];
        // (when what was set aside is the bulk of the batch -- CJK text -- it runs first and at full width instead)
        forked = plain_total > 0 && !h->no_overlap && (gen_total + doc_total) * 2 <= plain_total;
        if (forked) {
          HIP_OR_RETURN(h, hipEventRecord(ws->ev_fork, stream));
          HIP_OR_RETURN(h, hipStreamWaitEvent(ws->stream2, ws->ev_fork, 0));
          stream = ws->stream2;
          // wavefronts per workgroup of the general launch: 4 (SPMX_FORK_WAVES) -- a CU holds 16 wavefronts of the two
          // kernels together, the word kernel's workgroup has 12; with more the general launch runs as fast as alone and the
          // word round waits for its CUs (C2: 4 -> general 5.4 ms / word round 4.5 ms; 5 -> 3.2 / 6.1; profiles/
          // r04_ab_kernel_stats.txt).  SPMX_FORK_WAVES=0: one full tile per wavefront, by the launch's size.
          uint64_t gt = 0;
          for (int c = 0; c < ncls; ++c) gt += (static_cast<uint64_t>(gen_known[c]) + 63) / 64;
          const uint64_t per_cu = (gt + static_cast<uint64_t>(h->n_cu) - 1) / static_cast<uint64_t>(h->n_cu);
          stream_waves_cap = h->fork_waves ? h->fork_waves : static_cast<int>(per_cu < 4 ? 4 : (per_cu > 12 ? 12 : per_cu));
          stream_cus_cap = h->fork_cus;
          // Beside the word-per-lane round (kernels_wordwave.h: 12 wavefronts and 156 KB of LDS per workgroup) nothing
          // else fits a CU: the general launch gets CUs of its own instead -- 32 of them at full width (it is bound by
          // the latency of its longest sentences: 2.7 ms there against 2.3 ms on all 256), the word round the others and,
          // when those workgroups end, these (C2: 6.65 ms a step against 8.4 with 4 wavefronts on every CU;
          // gpurun_out r05h, profiles/r05_fork_ab.txt)
          if ((h->word_form & 1) && !h->no_word_dyn && h->fork_waves == 4 && h->fork_cus == 0) {
            stream_waves_cap = 16;
            stream_cus_cap = h->n_cu >= 64 ? 32 : (h->n_cu / 8 > 0 ? h->n_cu / 8 : 1);
          }
        }
        int rc = kOk;
        for (int c = 0; c < ncls && rc == kOk; ++c)
          if (doc_known[c]) rc = long_launch(class_lists + static_cast<size_t>(c) * n, &ws->d_ctrl->list_counts[c], doc_known[c], true);
        if (rc == kOk) rc = general_pass(gen_lists, ws->d_ctrl->gen_counts, gen_known, false);
        if (forked) {
          // (on any error while forked: nothing of this call may still be queued on the second stream when the
          // workspace goes back to the pool)
          if (rc != kOk) { (void)hipStreamSynchronize(ws->stream2); return rc; }
          if (hipError_t e = hipEventRecord(ws->ev_join, stream); e != hipSuccess) { (void)hipStreamSynchronize(ws->stream2); return FailHip(h, e, "hipEventRecord(join)"); }
          stream = main_stream;
          stream_waves_cap = 0;
          stream_cus_cap = 0;
        } else if (rc != kOk) {
          return rc;
        }
        a.lists = class_lists;
        d_list_counts = ws->d_ctrl->list_counts;
      }
    }
    // (from here to the join, an error return must not leave work of this call queued on the second stream)
    auto fail_forked = [&](int rc) -> int {
      if (forked) (void)hipStreamSynchronize(ws->stream2);
      return rc;
    };
#define FORKED_OR_RETURN(expr) do { if (int rc_ = (expr); rc_ != kOk) return fail_forked(rc_); } while (0)
#define FORKED_HIP_OR_RETURN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail_forked(FailHip(h, e_, #expr)); } while (0)
    if (word_ok) {
      // ---- the word kernels (kernels_word.h): every class from one queue, longest first.
      //   round 1   takes the sentences whose words are all in the load-time memo, and enters every other plain word
      //             it meets into the call-local memo;
      //   resolve   segments the collected words, one lane per word;
      //   round 2   over the sentences round 1 kept for it, looking the collected words up;
      //   (without the call-local memo, SPMX_NO_WORD_DYN=1: round 1, then the DP pass over what it left.)
      // What is left then -- a word of more than 16 bytes, a margin that does not hold, and without the plain scan
      // anything that is not plain ASCII words -- comes back as per-class lists for the tail launch below.
      const bool dyn = !h->no_word_dyn;
      word_pass = [&](int mode, int slot, int qi, uint32_t *out_lists, uint32_t *d_out_counts, uint32_t *out2_lists,
                           uint32_t *d_out2_counts) -> int {
        const bool dp = mode == 3;
        EncodeArgs wa = a;
        const bool wform = !dp && ((mode == 2 ? h->word_form & 2 : h->word_form & 1) != 0);   // word per lane
        int waves = dp ? 8 : wform ? h->wordwave_waves : h->word_waves;
        while (wform && waves > 1 && WordWaveLdsBytes(static_cast<uint32_t>(waves)) > 160u * 1024u) --waves;   // (a CU's LDS: an A/B build with a larger hot table)
        uint64_t total = 0;
        for (int c = 0; c < ncls; ++c) total += known[c];
        if (total == 0) return kOk;
        uint64_t grid = static_cast<uint64_t>(h->n_cu - h->reserve_cus) * static_cast<uint64_t>(dp ? 1 : h->word_wgs);
        if (grid * waves * 64 > total) grid = (total + waves * 64 - 1) / (waves * 64);
        if (grid < 1) grid = 1;
        const uint64_t n_waves = grid * waves;
        wa.n_classes = static_cast<uint32_t>(ncls);
        uint32_t tile_base = 0;
        wa.direct = direct ? 1u : 0u;
        if (direct && mode != 2) wa.lists = nullptr;             // (the first round: sentence = tile's first + lane)
        for (int c = ncls - 1; c >= 0; --c) {
          StreamClass &sc = wa.cls[c];
          sc = StreamClass{};
          sc.rcap = rcaps[c];
          if (direct) {                                          // every tile from row 0; a class only names documents and lists
            sc.lane_shift = 6;
            // (without classify's counts nobody knows whether documents are the bulk of the batch: beyond the staged classes a
            // sentence is handed on -- the bound of |score| a few thousand words add up leaves the margins of a long text's
            // later words behind anyway -- and the tail launch gives it a wavefront)
            sc.general = cls[c].rcap > kMaxStagedRaw ? 1u : 0u;
            if (c == 0) {
              sc.count = static_cast<uint32_t>(total);
              sc.tw = 64;
              sc.main_tiles = static_cast<uint32_t>((total + 63) / 64);
              sc.tile_base = 0;
              tile_base = sc.main_tiles;
            }
            continue;
          }
          if (known[c] == 0) continue;
          uint64_t tw = (static_cast<uint64_t>(known[c]) + n_waves - 1) / n_waves;
          if (tw > 64) tw = 64;
          if (tw < 1) tw = 1;
          sc.lane_shift = 6;
          sc.general = ((uni_class[c] && cls[c].rcap > kMaxStagedRaw) || cls[c].rcap > h->main_max_raw) ? 1u : 0u;   // documents pass through
          sc.count = known[c];
          sc.tw = static_cast<uint32_t>(tw);
          sc.main_tiles = static_cast<uint32_t>((static_cast<uint64_t>(known[c]) + tw - 1) / tw);
          sc.tile_base = tile_base;
          tile_base += sc.main_tiles;
        }
        wa.total_main = tile_base;
        wa.q = &ws->d_ctrl->q[qi];
        wa.stats = &ws->d_ctrl->stats[kStatsPerClass * slot];
        wa.left_lists = out_lists;
        wa.left_counts = d_out_counts;
        wa.left2_lists = out2_lists;
        wa.left2_counts = d_out2_counts;
        wa.dyn_tag = ws->d_dyn_tag.p;
        wa.dyn_ent = ws->d_dyn_ent.p;
        wa.dyn_list = ws->d_dyn_list.p;
        wa.dyn_count = &ws->d_ctrl->dyn_count;
        wa.dyn_mask = h->dyn_slots - 1u;
        wa.dyn_cap = h->dyn_list_cap;
        wa.resume = ws->d_resume.p;
        wa.ids16 = (h->model.pieces.size() <= 65536 && !h->no_ids16) ? 1u : 0u;
        snprintf(ws->slot_name[slot], sizeof(ws->slot_name[slot]), "%s",
                 wform ? (wa.ids16 ? (mode == 2 ? "EncodeWordWaveAgainKernel<true>" : mode == 1 ? "EncodeWordWaveCollectKernel<true>" : "EncodeWordWaveKernel<true>")
                                   : (mode == 2 ? "EncodeWordWaveAgainKernel<false>" : mode == 1 ? "EncodeWordWaveCollectKernel<false>" : "EncodeWordWaveKernel<false>"))
                       : mode == 3 ? "EncodeWordDpKernel" : mode == 2 ? "EncodeWordAgainKernel" : mode == 1 ? "EncodeWordCollectKernel" : "EncodeWordKernel");
        HIP_OR_RETURN(h, record(slot, 0));
        if (wform) HIP_OR_RETURN(h, LaunchEncodeWordWave(mode, wa, static_cast<int>(grid), waves, WordWaveLdsBytes(waves), stream));
        else HIP_OR_RETURN(h, LaunchEncodeWord(mode, wa, static_cast<int>(grid), waves, WordLdsBytes(waves, dp), stream));
        HIP_OR_RETURN(h, record(slot, 1));
        ws->slot_used[slot] = true;
        return kOk;
      };
      auto read_counts = [&]() -> int {
        HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl->left_counts, ws->d_ctrl->left_counts, sizeof(ws->h_ctrl->left_counts) + sizeof(uint32_t),
                                        hipMemcpyDeviceToHost, stream));      // (dyn_count follows left_counts in Ctrl)
        HIP_OR_RETURN(h, hipStreamSynchronize(stream));
        return kOk;
      };
      uint64_t eligible = 0;                 // sentences the first round was given (the backoff rule below)
      for (int c = 0; c < ncls; ++c) if (cls[c].rcap <= h->main_max_raw) eligible += known[c];
      int left_at = 0;                       // which of left_lists holds what the word rounds leave
      if (dyn) {
        FORKED_HIP_OR_RETURN(ws->d_dyn_tag.Reserve(h->dyn_slots));
        FORKED_HIP_OR_RETURN(ws->d_dyn_ent.Reserve(static_cast<size_t>(h->dyn_slots) * 4));
        FORKED_HIP_OR_RETURN(ws->d_dyn_list.Reserve(h->dyn_list_cap));
        FORKED_HIP_OR_RETURN(ws->d_resume.Reserve(n));
        FORKED_HIP_OR_RETURN(hipMemsetAsync(ws->d_dyn_tag.p, 0, static_cast<size_t>(h->dyn_slots) * sizeof(unsigned long long), stream));
        FORKED_OR_RETURN(word_pass(1, kSlotWord, 3, left_lists[0], ws->d_ctrl->left_counts[0], left_lists[1], ws->d_ctrl->left_counts[1]));
        FORKED_OR_RETURN(read_counts());
        for (int c = 0; c < ncls; ++c) { again_counts[c] = ws->h_ctrl->left_counts[0][c]; again_total += again_counts[c]; }
        again_words = ws->h_ctrl->dyn_count < h->dyn_list_cap ? ws->h_ctrl->dyn_count : h->dyn_list_cap;
        left_at = 1;
        if (again_total > 0) {
          // DIRECT (nothing else uses the second stream): what round 1 gave up for good is known NOW, and its tail launch
          // COULD go to the second stream beside resolve and round 2 (SPMX_EARLY_TAIL=1; the fork recorded here, before they
          // are enqueued -- recorded behind them, as it was until round 6, the second stream waited for round 2 anyway).
          // Not the default: round 2's workgroups take a CU's whole LDS, so nothing runs BESIDE them whatever the streams
          // say, and the extra launch with its read-back costs 0.03 ms (C2: 4.68 against 4.65 ms a step).  The give-ups of
          // both rounds then share one tail launch behind round 2.
          uint32_t gone1[kMaxClasses] = {0};
          uint64_t gone1_total = 0;
          for (int c = 0; c < ncls; ++c) { gone1[c] = ws->h_ctrl->left_counts[1][c]; gone1_total += gone1[c]; }
          const bool early_tail = h->early_tail && direct && !forked && !h->no_overlap && gone1_total > 0;
          if (early_tail) {
            HIP_OR_RETURN(h, hipEventRecord(ws->ev_fork, stream));
            HIP_OR_RETURN(h, hipStreamWaitEvent(ws->stream2, ws->ev_fork, 0));
          }
          if (again_words) {      // the collected words, segmented once each (a few workgroups)
            ResolveArgs ra{};
            ra.dev = h->dev; ra.dyn_ent = ws->d_dyn_ent.p; ra.dyn_list = ws->d_dyn_list.p; ra.dyn_count = &ws->d_ctrl->dyn_count;
            ra.dyn_cap = h->dyn_list_cap; ra.unsafe = h->memo_unsafe ? 1u : 0u;
            uint64_t g = (static_cast<uint64_t>(again_words) + 63) / 64;
            if (g > static_cast<uint64_t>(h->n_cu) * 8) g = static_cast<uint64_t>(h->n_cu) * 8;
            FORKED_HIP_OR_RETURN(LaunchWordResolve(ra, static_cast<int>(g), stream));
          }
          // round 2 over what round 1 kept for it; what it cannot take either (a margin that does not hold, a word of
          // more than 8 pieces) is APPENDED to the lists of what round 1 gave up for good: one tail launch takes both
          // (round 2 then keeps a list of its own for what it cannot take)
          if (early_tail) left_at = 2;
          for (int c = 0; c < ncls; ++c) known[c] = again_counts[c];
          a.lists = left_lists[0];
          FORKED_OR_RETURN(word_pass(2, kSlotWord2, 4, left_lists[left_at], ws->d_ctrl->left_counts[left_at], nullptr, nullptr));
          if (early_tail) {
            forked = true;
            stream = ws->stream2;
            tail_qi = 0;                                                     // (no main launch in a direct call: its queue is free)
            const int rc = general_pass(left_lists[1], ws->d_ctrl->left_counts[1], gone1, true);
            tail_qi = 5;
            hipError_t ej = rc == kOk ? hipEventRecord(ws->ev_join, stream) : hipSuccess;
            stream = main_stream;
            if (rc != kOk) return fail_forked(rc);
            if (ej != hipSuccess) return fail_forked(FailHip(h, ej, "hipEventRecord(join)"));
            a.lists = left_lists[0];
          }
          // (round 6 tried to take these counts with the control block after the compaction instead -- one synchronisation
          // less: C2's second round does hand a few sentences on in every call, and their tail then costs a second scan +
          // compaction: 6.11 ms a step against 5.38)
          FORKED_OR_RETURN(read_counts());
        }
      } else {
        FORKED_OR_RETURN(word_pass(0, kSlotWord, 3, left_lists[0], ws->d_ctrl->left_counts[0], nullptr, nullptr));
        FORKED_OR_RETURN(read_counts());
        uint64_t total = 0;
        for (int c = 0; c < ncls; ++c) { known[c] = ws->h_ctrl->left_counts[0][c]; total += known[c]; }
        a.lists = left_lists[0];
        // the DP pass pays when the first one's misses are sparse (a rare word here and there)
        if (!is_bpe && !h->no_word_dp && total && (total * 4 <= n || h->force_word_dp)) {
          FORKED_OR_RETURN(word_pass(3, kSlotWord2, 4, left_lists[1], ws->d_ctrl->left_counts[1], nullptr, nullptr));
          FORKED_OR_RETURN(read_counts());
          left_at = 1;
        }
      }
      uint64_t leftover = 0;
      for (int c = 0; c < ncls; ++c) {
        known[c] = ws->h_ctrl->left_counts[left_at][c];
        if (cls[c].rcap <= h->main_max_raw) leftover += known[c];
        // (the first round's give-ups went to their own tail launch already -- left_at == 2 -- and count all the same)
        if (left_at == 2 && cls[c].rcap <= h->main_max_raw) leftover += ws->h_ctrl->left_counts[1][c];
      }
      // The word rounds pay on text that is mostly plain ASCII words: a handle remembers when a batch they were given
      // went almost entirely on to the general kernels and leaves them out of its next few calls (see word_ok above).
      if (n >= 4096 && eligible >= 4096 && leftover * 8 > eligible * 7) h->word_backoff.store(15, std::memory_order_relaxed);
      // ... and when most of the batch was set aside by the plain scan (CJK text): the scan, the word rounds over the
      // plain minority and their tail only add to the one general launch that does the work
      if (scanned && n >= 4096 && (gen_total * 8 > static_cast<uint64_t>(n) * 5)) h->word_backoff.store(15, std::memory_order_relaxed);
      // ---- the tail: what the word rounds left ----
      // (its launches share the workspace's streaming scratch and slice pool with the launches on the second stream: they
      // wait for those to END -- on the host, because sizing the scratch may free what a running kernel still uses.  The
      // tail is empty on batches of plain text; the join below then costs nothing more.)
      uint64_t tail_total = 0;
      for (int c = 0; c < ncls; ++c) tail_total += known[c];
      if (forked && tail_total > 0) FORKED_HIP_OR_RETURN(hipStreamSynchronize(ws->stream2));
      FORKED_OR_RETURN(general_pass(left_lists[left_at], ws->d_ctrl->left_counts[left_at], known, true));
    } else {
      FORKED_OR_RETURN(general_pass(class_lists, ws->d_ctrl->list_counts, known, false));
    }
    // ---- join ----
    if (forked) FORKED_HIP_OR_RETURN(hipStreamWaitEvent(stream, ws->ev_join, 0));
#undef FORKED_OR_RETURN
#undef FORKED_HIP_OR_RETURN
    if (int rc = scan_compact(); rc != kOk) return rc;
    if (ws->h_ctrl->status & kStArenaOverflow) {   // rare: byte fallback of multi-byte unknowns; arena_head holds what was asked for
      arena_need = ws->h_ctrl->arena_head + ws->h_ctrl->arena_head / 8 + 64;
      continue;
    }
    bool extra = false;
    if (ws->h_ctrl->side.over_count) {             // sentences that fit no column of their launch: exact capacities
      uint32_t counts[kMaxClasses] = {0};
      counts[kMaxClasses - 1] = ws->h_ctrl->side.over_count;
      if (int rc = stream_launch(kSlotExact, 2, 0, 0, counts, true, ws->h_ctrl->side.over_max_raw); rc != kOk) return rc;
      extra = true;
      if (is_bpe) {                                // it may have added to the long list
        HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->side, &ws->d_ctrl->side, sizeof(SideLists), hipMemcpyDeviceToHost, stream));
        HIP_OR_RETURN(h, hipStreamSynchronize(stream));
      }
    }
    if (ws->h_ctrl->side.long_count) {             // BPE: what the lane / sentence-per-wave forms handed to the long form
      if (int rc = long_launch(long_list, &ws->d_ctrl->side.long_count, ws->h_ctrl->side.long_count); rc != kOk) return rc;
      extra = true;
    }
    if (extra) {
      if (int rc = scan_compact(); rc != kOk) return rc;
      if (ws->h_ctrl->status & kStArenaOverflow) {
        arena_need = ws->h_ctrl->arena_head + ws->h_ctrl->arena_head / 8 + 64;
        continue;
      }
    }
    if (prof) {
      HIP_OR_RETURN(h, hipEventRecord(ws->ev[kNumSlots][1], stream));
      HIP_OR_RETURN(h, hipStreamSynchronize(stream));
      Profile &p = ws->prof;
      p = Profile();
      p.n = kNumSlots;
      for (int c = 0; c < kNumSlots; ++c) {
        if (!ws->slot_used[c]) continue;
        memcpy(p.name[c], ws->slot_name[c], sizeof(p.name[c]));
        HIP_OR_RETURN(h, hipEventElapsedTime(&p.kernel_ms[c], ws->ev[c][0], ws->ev[c][1]));
        const unsigned long long *s = &ws->h_ctrl->stats[kStatsPerClass * c];
        p.sentences[c] = s[0];
        p.raw_bytes[c] = s[1];
        p.ids[c] = s[2];
        for (int k = 0; k < 5; ++k) p.cycles[c][k] = s[3 + k];
      }
      p.path[0] = ws->h_ctrl->side.n_backlog; p.path[1] = ws->h_ctrl->side.over_count; p.path[2] = ws->h_ctrl->side.long_count;
      p.path[3] = ws->h_ctrl->side.n_failed;
      HIP_OR_RETURN(h, hipEventElapsedTime(&p.total_ms, ws->ev[kNumSlots][0], ws->ev[kNumSlots][1]));
      std::lock_guard<std::mutex> l(h->mu);
      h->prof = p;
    }
    if (total_ids) *total_ids = ws->h_ctrl->total_ids;
    if (n_failed) *n_failed = ws->h_ctrl->side.n_failed;
    if (ws->h_ctrl->total_ids > ids_capacity || !d_ids) {
      if (ws->h_ctrl->total_ids == 0) return kOk;
      return Fail(h, kResourceExhausted, "ids_capacity is too small");
    }
    if (spans && ws->h_ctrl->total_ids > 0) {
      // token begins to CSR order, then one align launch per staged length class over the classify lists
      const uint64_t total = ws->h_ctrl->total_ids;
      HIP_OR_RETURN(h, ws->d_tok_begin.Reserve(total));
      CompactArgs pa{ws->d_arena_tb.p, ws->d_tmp_off.p, ws->d_counts.p, d_id_offsets, ws->d_tok_begin.p, total, n32, nullptr, h->compact_staged,
                     h->compact_big, ws->d_big_list.p, &ws->d_ctrl->big_count[1]};
      const uint64_t cblocks = (n + 63) / 64;
      const uint64_t cgrid = cblocks < static_cast<uint64_t>(h->n_cu) * 32 ? cblocks : static_cast<uint64_t>(h->n_cu) * 32;
      HIP_OR_RETURN(h, LaunchCompact(pa, static_cast<int>(cgrid), stream));
      if (h->compact_big) HIP_OR_RETURN(h, LaunchCompactBig(pa, h->n_cu * 4, stream));
      HIP_OR_RETURN(h, hipMemsetAsync(&ws->d_ctrl->status, 0, sizeof(uint32_t), stream));
      HIP_OR_RETURN(h, hipMemsetAsync(ws->d_ctrl->align_counts, 0, sizeof(ws->d_ctrl->align_counts), stream));
      HIP_OR_RETURN(h, hipMemsetAsync(&ws->d_ctrl->side.long_count, 0, sizeof(uint32_t), stream));
      AlignArgs base{};
      base.dev = h->dev; base.text = d_text; base.offs = d_offsets;
      base.id_offs = d_id_offsets; base.tok_begin = ws->d_tok_begin.p; base.begin = d_begin; base.end = d_end;
      base.nbegin = d_nbegin; base.nend = d_nbegin ? d_nend : nullptr;
      base.status = &ws->d_ctrl->status; base.list_cap = n32;
      bool prev = false;
      int c_staged = 0;
      while (c_staged < ncls && cls[c_staged].rcap <= kMaxStagedRaw) ++c_staged;
      for (int c = 0; c < c_staged; ++c) {
        const uint32_t cnt = known[c];
        if (cnt == 0 && !prev) continue;
        const bool has_next = c + 1 < c_staged;
        AlignArgs aa = base;
        aa.rcap = cls[c].rcap; aa.ncap = cls[c].ncap;
        // past the last staged class the sentence goes to the lane-per-sentence align kernel's list
        aa.next_list = has_next ? hard_lists + static_cast<size_t>(c + 1) * n : long_list;
        aa.next_count = has_next ? &ws->d_ctrl->align_counts[c + 1] : &ws->d_ctrl->side.long_count;
        const uint32_t lds = AlignLdsBytes(aa.rcap, aa.ncap, aa.nbegin != nullptr && aa.nend != nullptr);
        int per_cu = static_cast<int>(kLdsPerCu / lds);
        if (per_cu > 32) per_cu = 32;
        if (per_cu < 1) per_cu = 1;
        const uint64_t full = static_cast<uint64_t>(h->n_cu) * per_cu;
        if (cnt) {                       // the class's own sentences
          aa.list = class_lists + static_cast<size_t>(c) * n; aa.list_count = &ws->d_ctrl->list_counts[c];
          HIP_OR_RETURN(h, LaunchAlign(aa, static_cast<int>(full < cnt ? full : cnt), lds, stream));
        }
        if (prev) {                      // what the previous class's align kernels could not hold (count on the device)
          aa.list = hard_lists + static_cast<size_t>(c) * n; aa.list_count = &ws->d_ctrl->align_counts[c];
          HIP_OR_RETURN(h, LaunchAlign(aa, static_cast<int>(full < 256 ? full : 256), lds, stream));
        }
        prev = true;
      }
      // lane per sentence (kernels_long.h): the classes beyond the staged ones, and what the staged kernels passed on
      auto align_long = [&](const uint32_t *list, const uint32_t *d_count, uint64_t most) -> int {
        if (most == 0) return kOk;
        AlignLongArgs la{};
        la.dev = h->dev; la.text = d_text; la.offs = d_offsets; la.list = list; la.list_count = d_count;
        la.id_offs = d_id_offsets; la.tok_begin = ws->d_tok_begin.p; la.begin = d_begin; la.end = d_end;
        la.nbegin = d_nbegin; la.nend = d_nbegin ? d_nend : nullptr;
        la.status = &ws->d_ctrl->status;
        uint64_t g = (most + 63) / 64;
        if (g > static_cast<uint64_t>(h->n_cu) * 16) g = static_cast<uint64_t>(h->n_cu) * 16;
        HIP_OR_RETURN(h, LaunchAlignLong(la, static_cast<int>(g), stream));
        return kOk;
      };
      for (int c = c_staged; c < ncls; ++c)
        if (int rc = align_long(class_lists + static_cast<size_t>(c) * n, &ws->d_ctrl->list_counts[c], known[c]); rc != kOk) return rc;
      if (prev) if (int rc = align_long(long_list, &ws->d_ctrl->side.long_count, 256 * 64); rc != kOk) return rc;
      uint32_t st2 = 0;
      HIP_OR_RETURN(h, hipMemcpyAsync(&st2, &ws->d_ctrl->status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      HIP_OR_RETURN(h, hipStreamSynchronize(stream));
      if (st2) return Fail(h, kInternal, "token boundaries do not tile the normalized text");
    }
    return kOk;
  }
  return Fail(h, kInternal, "id arena kept overflowing");
}

// Batch Normalize on the device: classify -> count pass -> scan -> write pass.  The staged classes run the
// position-parallel kernels (kernels_normalize.h); longer sentences, and what overflows the last staged class, the
// lane-per-sentence ones (kernels_long.h).
int NormalizeDevice(spmx_handle *h, Workspace *ws, const uint8_t *d_text, const uint64_t *d_offsets, uint64_t n, uint8_t *d_norm,
                    uint64_t norm_capacity, uint64_t *d_norm_offsets, uint32_t *d_n2o, hipStream_t stream,
                    uint64_t *total_bytes, bool device_text = false, const SpmxDev *dev = nullptr) {
  if (total_bytes) *total_bytes = 0;
  if (n >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "more than 2^32 - 64 sentences in one batch");
  if (!d_offsets || !d_norm_offsets) return Fail(h, kInvalidArgument, "null offsets");
  if (n == 0) {
    HIP_OR_RETURN(h, hipMemsetAsync(d_norm_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  const int ncls = kNumClasses;
  const LengthClass *cls = h->classes;
  HIP_OR_RETURN(h, ws->d_lists.Reserve(static_cast<size_t>(2 * kMaxClasses + 3) * n));
  HIP_OR_RETURN(h, ws->d_counts.Reserve(n + 1));
  HIP_OR_RETURN(h, ws->d_tile_sums.Reserve((n + kScanTile - 1) / kScanTile + 2));
  HIP_OR_RETURN(h, hipMemsetAsync(ws->d_ctrl, 0, sizeof(Ctrl), stream));
  const uint32_t n32 = static_cast<uint32_t>(n);
  const int wide = h->n_cu * 8;
  uint32_t *const class_lists = ws->d_lists.p;
  uint32_t *const long_list = class_lists + static_cast<size_t>(2 * kMaxClasses) * n;
  if (int rc = RunClassify(h, ws, d_offsets, n32, stream); rc != kOk) return rc;
  int c_staged = 0;
  while (c_staged < ncls && cls[c_staged].rcap <= kMaxStagedRaw) ++c_staged;
  auto pass = [&](bool write) -> int {
    NormalizeArgs base{};
    base.dev = dev ? *dev : h->dev; base.text = d_text; base.offs = d_offsets;
    base.counts = ws->d_counts.p; base.norm_offs = d_norm_offsets; base.norm = d_norm; base.n2o = d_n2o;
    base.status = &ws->d_ctrl->status;
    base.device_text = device_text ? 1u : 0u;
    for (int c = 0; c < c_staged; ++c) {
      NormalizeArgs a = base;
      a.list = class_lists + static_cast<size_t>(c) * n; a.list_count = &ws->d_ctrl->list_counts[c];
      const bool has_next = c + 1 < c_staged;
      a.next_list = has_next ? class_lists + static_cast<size_t>(c + 1) * n : long_list;
      a.next_count = has_next ? &ws->d_ctrl->list_counts[c + 1] : &ws->d_ctrl->side.long_count;
      a.rcap = cls[c].rcap; a.ncap = cls[c].ncap;
      const uint32_t lds = NormalizeLdsBytes(a.rcap, a.ncap);
      int per_cu = static_cast<int>(kLdsPerCu / lds);
      if (per_cu > 32) per_cu = 32;
      if (per_cu < 1) per_cu = 1;
      uint64_t grid = static_cast<uint64_t>(h->n_cu) * per_cu;
      if (grid > n) grid = n;
      HIP_OR_RETURN(h, LaunchNormalize(write, a, static_cast<int>(grid), lds, stream));
    }
    uint64_t g = (n + 63) / 64;
    if (g > static_cast<uint64_t>(h->n_cu) * 16) g = static_cast<uint64_t>(h->n_cu) * 16;
    for (int c = c_staged; c <= ncls; ++c) {            // c == ncls: what the last staged class passed on
      NormalizeArgs a = base;
      a.list = c < ncls ? class_lists + static_cast<size_t>(c) * n : long_list;
      a.list_count = c < ncls ? &ws->d_ctrl->list_counts[c] : &ws->d_ctrl->side.long_count;
      HIP_OR_RETURN(h, LaunchNormalizeLong(write, a, static_cast<int>(g), stream));
    }
    return kOk;
  };
  if (int rc = pass(false); rc != kOk) return rc;
  {
    ScanArgs sa{ws->d_counts.p, n32, ws->d_tile_sums.p, d_norm_offsets};
    const uint32_t tiles = (n32 + kScanTile - 1) / kScanTile;
    HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(wide) ? tiles : wide), stream));
  }
  HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl, ws->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->total_ids, d_norm_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  if (ws->h_ctrl->status & kStTooLong) return Fail(h, kOutOfRange, "the normalized form of a sentence would exceed 2^31 bytes");
  const uint64_t total = ws->h_ctrl->total_ids;
  if (total_bytes) *total_bytes = total;
  if (total == 0 && !d_n2o) return kOk;
  if ((!d_norm && total) || total > norm_capacity) return Fail(h, kResourceExhausted, "norm_capacity is too small");
  if (int rc = pass(true); rc != kOk) return rc;
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  return kOk;
}

// Batch Decode on the device (kernels_decode.h): count pass -> scan -> (host checks status / capacity) -> write pass.
int DecodeRaw(spmx_handle *h, Workspace *ws, const int32_t *d_ids, const uint64_t *d_id_offsets, uint64_t n, uint8_t *d_text,
              uint64_t text_capacity, uint64_t *d_text_offsets, hipStream_t stream, uint64_t *total_bytes) {
  if (total_bytes) *total_bytes = 0;
  if (n >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "more than 2^32 - 64 sentences in one batch");
  if (!d_id_offsets || !d_text_offsets) return Fail(h, kInvalidArgument, "null offsets");
  if (n == 0) {
    HIP_OR_RETURN(h, hipMemsetAsync(d_text_offsets, 0, sizeof(uint64_t), stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  }
  HIP_OR_RETURN(h, ws->d_counts.Reserve(n + 1));
  HIP_OR_RETURN(h, ws->d_tile_sums.Reserve((n + kScanTile - 1) / kScanTile + 2));
  HIP_OR_RETURN(h, hipMemsetAsync(ws->d_ctrl, 0, sizeof(Ctrl), stream));
  HIP_OR_RETURN(h, hipMemsetAsync(&ws->d_ctrl->bad_key, 0xFF, sizeof(unsigned long long), stream));
  DecodeArgs a{};
  a.dev = h->dev; a.ids = d_ids; a.id_offs = d_id_offsets; a.n = static_cast<uint32_t>(n);
  a.counts = ws->d_counts.p; a.text_offs = d_text_offsets; a.text = d_text; a.text_cap = d_text ? text_capacity : 0;
  a.status = &ws->d_ctrl->status; a.bad_key = &ws->d_ctrl->bad_key;
  {
    std::lock_guard<std::mutex> l(h->mu);
    a.lit_bytes = ws->lit_bytes; a.lit_offs = ws->lit_offs; a.n_lit = ws->n_lit;
    a.x_npre = h->dx_npre; a.x_nsuf = h->dx_nsuf; a.x_reverse = h->dx_reverse ? 1 : 0;
    for (int i = 0; i < kMaxExtra; ++i) { a.x_pre[i] = h->dx_pre[i]; a.x_suf[i] = h->dx_suf[i]; }
  }
  const uint64_t wide = static_cast<uint64_t>(h->n_cu) * 32;
  const int grid = static_cast<int>(n < wide ? n : wide);
  HIP_OR_RETURN(h, LaunchDecode(false, a, grid, stream));
  {
    ScanArgs sa{ws->d_counts.p, static_cast<uint32_t>(n), ws->d_tile_sums.p, d_text_offsets};
    const uint32_t tiles = (static_cast<uint32_t>(n) + kScanTile - 1) / kScanTile;
    HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(h->n_cu * 8) ? tiles : h->n_cu * 8), stream));
  }
  HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl, ws->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->total_ids, d_text_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  if (ws->h_ctrl->status & kStBadId) {   // sentencepiece_processor.cc:913-917 (the id reported is one of the batch's first failing sentence)
    const int id = static_cast<int>(static_cast<uint32_t>(ws->h_ctrl->bad_key));
    return Fail(h, kOutOfRange, "Invalid id: " + std::to_string(id));
  }
  const uint64_t total = ws->h_ctrl->total_ids;
  if (total_bytes) *total_bytes = total;
  if (total == 0) return kOk;
  if (!d_text || total > text_capacity) return Fail(h, kResourceExhausted, "text_capacity is too small");
  HIP_OR_RETURN(h, LaunchDecode(true, a, grid, stream));
  HIP_OR_RETURN(h, hipStreamSynchronize(stream));
  return kOk;
}

// Decode, then the denormalizer over each decoded sentence when the model carries one (`*text = denormalizer_->
// Normalize(*text)`, sentencepiece_processor.cc:905-907): the batch Normalize kernels with the denormalizer's tables.
int DecodeDevice(spmx_handle *h, Workspace *ws, const int32_t *d_ids, const uint64_t *d_id_offsets, uint64_t n, uint8_t *d_text,
                 uint64_t text_capacity, uint64_t *d_text_offsets, hipStream_t stream, uint64_t *total_bytes) {
  if (!h->model.has_denormalizer)
    return DecodeRaw(h, ws, d_ids, d_id_offsets, n, d_text, text_capacity, d_text_offsets, stream, total_bytes);
  if (total_bytes) *total_bytes = 0;
  if (!d_id_offsets || !d_text_offsets) return Fail(h, kInvalidArgument, "null offsets");
  HIP_OR_RETURN(h, ws->d_dn_offs.Reserve(n + 1));
  uint64_t cap = ws->d_dn_text.cap ? ws->d_dn_text.cap : (1u << 20), total = 0;
  int rc = kOk;
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIP_OR_RETURN(h, ws->d_dn_text.Reserve(cap));
    rc = DecodeRaw(h, ws, d_ids, d_id_offsets, n, ws->d_dn_text.p, ws->d_dn_text.cap, ws->d_dn_offs.p, stream, &total);
    if (rc != kResourceExhausted || total <= ws->d_dn_text.cap) break;
    cap = total + 64;
  }
  if (rc != kOk) return rc;
  rc = NormalizeDevice(h, ws, ws->d_dn_text.p, ws->d_dn_offs.p, n, d_text, d_text ? text_capacity : 0, d_text_offsets, nullptr, stream,
                       total_bytes, false, &h->dn_dev);
  if (rc == kResourceExhausted) t_error = "text_capacity is too small";
  return rc;
}

bool ReadFile(const char *path, std::string *out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

const char *StatusText(int code) {
  switch (code) {
    case kOutOfRange: return "the sentence is too long for the device path (its normalized form would exceed 2^31 bytes)";
    case kResourceExhausted: return "out of device memory for the sentence's working set";
    default: return "all normalized characters are not consumed.";   // sentencepiece_processor.cc:628
  }
}

// Every extern "C" body runs under this: no C++ exception crosses the boundary (include/spmx.h).
template <typename F>
int Guard(spmx_handle *h, F f) {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    return Fail(h, kResourceExhausted, "out of host memory");
  } catch (const std::exception &e) {
    return Fail(h, kInternal, e.what());
  } catch (...) {
    return Fail(h, kInternal, "unknown exception");
  }
}

}  // namespace

extern "C" {

namespace { int RunSelfTest(spmx_handle *h); }

int spmx_create(const void *model_bytes, uint64_t n_bytes, int device, spmx_handle **out) {
  if (!out) return Fail(nullptr, kInvalidArgument, "null output handle");
  *out = nullptr;
  return Guard(nullptr, [&]() -> int {
    if (!model_bytes || n_bytes == 0) return Fail(nullptr, kInvalidArgument, "empty model");   // "model file is empty" analogue
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
      return Fail(nullptr, kUnavailable, std::string("no HIP device is available (libspmx has no CPU path): ") +
                                             (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= n_dev) return Fail(nullptr, kInvalidArgument, "device ordinal out of range");
    const auto t_load0 = std::chrono::steady_clock::now();
    std::unique_ptr<spmx_handle, void (*)(spmx_handle *)> h(new spmx_handle, DestroyHandle);
    h->device = device;
    h->serialized.assign(static_cast<const char *>(model_bytes), n_bytes);
    Status st = ParseModelProto(model_bytes, n_bytes, &h->model);
    if (st.ok()) st = InitializeModel(&h->model);
    if (st.ok()) st = CompileTables(h->model, &h->tables);
    if (st.ok()) st = CompileExtraOptions(h->model, "", &h->tables);
    if (st.ok() && h->model.has_denormalizer) {   // std::make_unique<normalizer::Normalizer>(denormalizer_spec) (:248-252)
      ModelData dn;
      dn.normalizer_only = true;
      dn.charsmap = h->model.dn_charsmap;
      dn.add_dummy_prefix = h->model.dn_add_dummy_prefix;
      dn.remove_extra_ws = h->model.dn_remove_extra_ws;
      dn.escape_ws = h->model.dn_escape_ws;
      st = CompileTables(dn, &h->dn_tables);
    }
    if (!st.ok()) return Fail(nullptr, st.code, st.message);
    HIP_OR_RETURN(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_OR_RETURN(nullptr, hipGetDeviceProperties(&prop, device));
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (int c = 0; c < kNumClasses; ++c) h->classes[c] = kClasses[c];
    if (const char *e = getenv("SPMX_NO_FAST")) h->no_fast = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_STREAM")) h->no_stream = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_WAVE")) h->no_wave = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_WORD_KERNEL")) h->no_word = e[0] == '1';
    if (const char *e = getenv("SPMX_FORCE_WORD_DP")) h->force_word_dp = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_WORD_DYN")) h->no_word_dyn = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_OVERLAP")) h->no_overlap = e[0] == '1';
    if (const char *e = getenv("SPMX_EARLY_TAIL")) h->early_tail = e[0] == '1';
    if (const char *e = getenv("SPMX_FORK_WAVES")) h->fork_waves = atoi(e);
    if (const char *e = getenv("SPMX_FORK_CUS")) h->fork_cus = atoi(e);
    if (const char *e = getenv("SPMX_NO_DIRECT")) h->no_direct = e[0] == '1';
    if (const char *e = getenv("SPMX_COMPACT_STAGED")) {
      const long v = atol(e);
      h->compact_staged = v <= 0 ? 0u : v < 256 ? 256u : v > static_cast<long>(kCompactLdsIdsMax) ? kCompactLdsIdsMax : static_cast<uint32_t>(v) & ~7u;
    }
    if (const char *e = getenv("SPMX_NO_SCAN")) h->no_scan = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_IDS16")) h->no_ids16 = e[0] == '1';
    if (const char *e = getenv("SPMX_DYN_SLOTS_LOG2")) { const int v = atoi(e); if (v >= 4 && v <= 26) h->dyn_slots = 1u << v; }
    if (const char *e = getenv("SPMX_DYN_LIST_CAP")) { const long v = atol(e); if (v >= 1 && v <= (1l << 26)) h->dyn_list_cap = static_cast<uint32_t>(v); }
    if (const char *e = getenv("SPMX_NBEST_BUDGET_GB")) { const long v = atol(e); if (v >= 1 && v <= 200) h->nbest_budget = static_cast<uint64_t>(v) << 30; }
#ifdef SPMX_TEST_SEAMS   // (the emulator build, tests/emu/Makefile: the release library reads none of these)
    if (getenv("SPMX_WORDMEMO_UNSAFE")) h->memo_unsafe = true;
    if (getenv("SPMX_UW_EXACT")) h->uw_exact = true;   // the wave-cooperative form folds in double arithmetic only (kernels_uniwave.h uw_fold_exact)
    if (const char *e = getenv("SPMX_COMPACT_BIG")) h->compact_big = static_cast<uint32_t>(atoll(e));   // ids of a block that goes to CompactBigKernel (0: none does)
    // A/B switches of settled experiments (their measurements: DESIGN.md section 4, profiles/)
    if (const char *e = getenv("SPMX_NO_WORD_DP")) h->no_word_dp = e[0] == '1';
    if (const char *e = getenv("SPMX_NBEST_HYPS_MIN")) { const long v = atol(e); if (v >= 1024 && v <= 262144) h->nbest_hyps_min = static_cast<uint32_t>(v); }
    if (const char *e = getenv("SPMX_TILE_MIN_LANES")) h->tile_min_lanes = static_cast<uint32_t>(atoi(e));
    if (const char *e = getenv("SPMX_NO_UNI_WAVE")) h->no_uni_wave = e[0] == '1';
    if (const char *e = getenv("SPMX_WORD_WGS")) { const int v = atoi(e); if (v >= 1 && v <= 4) h->word_wgs = v; }
    if (const char *e = getenv("SPMX_NO_BP_SHORT")) h->no_bp_short = e[0] == '1';
    if (const char *e = getenv("SPMX_WIDE_TCAP")) h->wide_tcap = e[0] == '1';
    if (const char *e = getenv("SPMX_SUB_BUCKETS")) {
      const int v = atoi(e);
      h->sub_buckets = static_cast<uint32_t>(v < 1 ? 1 : (v > kMaxSubBuckets ? kMaxSubBuckets : v));
    }
    if (const char *e = getenv("SPMX_LANE_GENERAL_MIN_LANES")) h->lane_general_min_lanes = static_cast<uint32_t>(atoi(e));
    if (const char *e = getenv("SPMX_STREAM_SCRATCH_MB")) h->stream_scratch_limit = static_cast<uint64_t>(atoll(e)) << 20;
    if (const char *e = getenv("SPMX_MAIN_MAX_RAW")) h->main_max_raw = static_cast<uint32_t>(atoll(e));
    if (const char *e = getenv("SPMX_FORCE_RING")) { const int v = atoi(e); if (v >= 16 && v <= 122) h->ring_override = static_cast<uint32_t>(v); }
#else
    {   // (an A/B script that sets one of these against the release library would compare a configuration with itself: say so, once)
      static const char *const kSeamOnly[] = {"SPMX_NO_WORD_DP", "SPMX_NBEST_HYPS_MIN", "SPMX_TILE_MIN_LANES",
          "SPMX_NO_UNI_WAVE", "SPMX_WORD_WGS", "SPMX_NO_BP_SHORT", "SPMX_WIDE_TCAP", "SPMX_SUB_BUCKETS", "SPMX_LANE_GENERAL_MIN_LANES",
          "SPMX_STREAM_SCRATCH_MB", "SPMX_MAIN_MAX_RAW", "SPMX_FORCE_RING", "SPMX_MEMO16_ONE"};
      static std::atomic<bool> warned{false};
      for (const char *v : kSeamOnly)
        if (getenv(v) && !warned.exchange(true))
          fprintf(stderr, "libspmx: %s is read by test-seam builds only (-DSPMX_TEST_SEAMS); this library ignores it\n", v);
    }
#endif
    if (const char *e = getenv("SPMX_NO_SPLIT")) h->no_split = e[0] == '1';
    if (const char *e = getenv("SPMX_SPLIT_MIN")) h->split_min_raw = static_cast<uint32_t>(atoll(e));
    if (const char *e = getenv("SPMX_SPLIT_LAUNCH")) h->split_own_launch = e[0] == '1';
    if (const char *e = getenv("SPMX_SPLIT_TILES")) { const int v = atoi(e); if (v >= 1 && v <= 16) h->split_tiles = static_cast<uint32_t>(v); }
    if (const char *e = getenv("SPMX_SPLIT_CANDS")) { const int v = atoi(e); if (v >= 1 && v <= 16) h->split_per_byte = static_cast<uint32_t>(v); }
    if (const char *e = getenv("SPMX_UW_PIPE")) h->uw_pipe = atoi(e);
    if (const char *e = getenv("SPMX_UNI_WAVE_MAX")) h->uni_wave_max = static_cast<uint32_t>(atoll(e));
    if (const char *e = getenv("SPMX_WORD_WAVE")) { const int v = atoi(e); if (v >= 0 && v <= 3) h->word_form = v; }
    if (const char *e = getenv("SPMX_WORDWAVE_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 16) h->wordwave_waves = v; }
    if (const char *e = getenv("SPMX_WORD_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 16) h->word_waves = v; }
    if (const char *e = getenv("SPMX_ARENA_FIRST")) h->arena_first = static_cast<uint64_t>(atoll(e));
    if (const char *e = getenv("SPMX_NO_LANE_GENERAL")) h->no_lane_general = e[0] == '1';
    if (const char *e = getenv("SPMX_NO_CHAR_NORM")) h->char_norm_mode = e[0] == '1' ? 1u : 0u;
    if (const char *e = getenv("SPMX_CHAR_NORM_ALWAYS")) { if (e[0] == '1') h->char_norm_mode = 2u; }
    if (const char *e = getenv("SPMX_TILE_WAVES")) h->tile_waves_override = atoi(e);
    if (const char *e = getenv("SPMX_RESERVE_CUS")) { const int v = atoi(e); if (v >= 0 && v < h->n_cu) h->reserve_cus = v; }
    if (const char *e = getenv("SPMX_HOST_THREADS")) { const int v = atoi(e); h->host_threads = v < 1 ? 1 : (v > 64 ? 64 : v); }
    if (const char *e = getenv("SPMX_HOST_CHUNK")) { const long long v = atoll(e); if (v >= 1024) h->host_chunk = static_cast<uint64_t>(v); }
    if (const char *e = getenv("SPMX_CLASSES")) {          // "raw:norm,raw:norm,..." (ascending; tests shrink the table)
      int c = 0;
      const char *p = e;
      while (*p && c < kNumClasses) {
        char *q = nullptr;
        const unsigned long r = strtoul(p, &q, 10);
        if (q == p || *q != ':') break;
        p = q + 1;
        const unsigned long nn = strtoul(p, &q, 10);
        if (q == p) break;
        h->classes[c++] = LengthClass{static_cast<uint32_t>(r), static_cast<uint32_t>(nn)};
        p = *q == ',' ? q + 1 : q;
      }
    }
    t_upload_bytes = 0;
    if (int rc = UploadTables(h.get()); rc != kOk) return rc;
    h->table_bytes = t_upload_bytes;
    h->load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_load0).count();
    if (!h->model.self_test.empty())                        // "Running self-testing." (sentencepiece_processor.cc:259-278)
      if (int rc = RunSelfTest(h.get()); rc != kOk) return rc;
    *out = h.release();
    return kOk;
  });
}

int spmx_create_from_file(const char *filename, int device, spmx_handle **out) {
  if (out) *out = nullptr;
  return Guard(nullptr, [&]() -> int {
    std::string blob;
    if (!filename || !ReadFile(filename, &blob))   // io::LoadModelProto (sentencepiece_processor.cc:1131-1149)
      return Fail(nullptr, kNotFound, std::string("\"") + (filename ? filename : "") + "\": No such file or directory");
    return spmx_create(blob.data(), blob.size(), device, out);
  });
}

void spmx_destroy(spmx_handle *h) { DestroyHandle(h); }

const char *spmx_last_error(const spmx_handle *) { return t_error.c_str(); }

int spmx_set_encode_extra_options(spmx_handle *h, const char *options) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    Status st = CompileExtraOptions(h->model, options ? options : "", &h->tables);
    if (!st.ok()) return Fail(h, st.code, st.message);
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    return RefreshDevice(h, false);
  });
}

int spmx_set_decode_extra_options(spmx_handle *h, const char *options) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    HostTables scratch;                  // (only the scalars the option compiler writes are read back)
    Status st = CompileExtraOptions(h->model, options ? options : "", &scratch);
    if (!st.ok()) return Fail(h, st.code, st.message);
    std::lock_guard<std::mutex> l(h->mu);
    h->dx_npre = scratch.scalars.n_prefix; h->dx_nsuf = scratch.scalars.n_suffix;
    for (int i = 0; i < kMaxExtra; ++i) { h->dx_pre[i] = scratch.scalars.prefix_ids[i]; h->dx_suf[i] = scratch.scalars.suffix_ids[i]; }
    h->dx_reverse = (scratch.scalars.flags & kNfReverse) != 0;
    h->dx_unk = false;
    for (std::string rest = options ? options : ""; !rest.empty();) {
      const size_t q = rest.find(':');
      const std::string o = rest.substr(0, q);
      if (o == "unk" || o == "unk_piece") h->dx_unk = true;
      rest = q == std::string::npos ? std::string() : rest.substr(q + 1);
    }
    return kOk;
  });
}

int spmx_set_vocabulary(spmx_handle *h, const char *const *pieces, const uint64_t *piece_lens, uint64_t n) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    std::vector<std::string> v;
    v.reserve(n);
    for (uint64_t i = 0; i < n; ++i) v.emplace_back(pieces[i], piece_lens[i]);
    Status st = SetVocabulary(&h->model, v);
    if (!st.ok()) return Fail(h, st.code, st.message);
    RefreshTypeFlags(h->model, &h->tables);
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    return RefreshDevice(h, true);
  });
}

int spmx_reset_vocabulary(spmx_handle *h) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    ResetVocabulary(&h->model);
    RefreshTypeFlags(h->model, &h->tables);
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    return RefreshDevice(h, true);
  });
}

int spmx_piece_size(const spmx_handle *h) { return h ? static_cast<int>(h->model.pieces.size()) : 0; }
int spmx_piece_to_id(const spmx_handle *h, const char *piece, uint64_t len) {
  if (!h) return 0;
  try { return h->model.PieceToId(std::string(piece ? piece : "", piece ? len : 0)); } catch (...) { return h->model.unk_id; }
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
// bos_id / eos_id / pad_id (src/sentencepiece_processor.cc:1000-1017): PieceToId, -1 unless that piece IsControl
static int ReservedId(const spmx_handle *h, const std::string &piece) {
  if (!h) return -1;
  try {
    const int id = h->model.PieceToId(std::string(piece.c_str()));
    return h->model.pieces[id].type == kControl ? id : -1;
  } catch (...) { return -1; }
}
int spmx_bos_id(const spmx_handle *h) { return h ? ReservedId(h, h->model.bos_piece) : -1; }
int spmx_eos_id(const spmx_handle *h) { return h ? ReservedId(h, h->model.eos_piece) : -1; }
int spmx_pad_id(const spmx_handle *h) { return h ? ReservedId(h, h->model.pad_piece) : -1; }
int spmx_model_type(const spmx_handle *h) { return h ? h->model.model_type : 0; }
uint32_t spmx_model_flags(const spmx_handle *h) { return h ? h->dev.flags : 0u; }
int64_t spmx_unk_piece(const spmx_handle *h, char *out, uint64_t cap) {
  if (!h) return -1;
  const std::string &p = h->model.unk_piece;
  if (out && cap) memcpy(out, p.data(), p.size() < cap ? p.size() : cap);
  return static_cast<int64_t>(p.size());
}

int spmx_encode_batch_device_ex(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                                uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets,
                                uint8_t *d_status, void *stream, uint64_t *total_ids, uint64_t *n_failed) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    return EncodeDevice(h, L.ws.get(), static_cast<const uint8_t *>(d_text), text_bytes, d_offsets, n, d_ids, ids_capacity,
                        d_id_offsets, d_status, static_cast<hipStream_t>(stream), total_ids, n_failed);
  });
}

int spmx_encode_batch_device(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                             uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets, void *stream,
                             uint64_t *total_ids) {
  return spmx_encode_batch_device_ex(h, d_text, text_bytes, d_offsets, n, d_ids, ids_capacity, d_id_offsets, nullptr, stream,
                                     total_ids, nullptr);
}

namespace {
// Host-buffer form of the batch encode, with (begin / end non-null) or without the spans.
int EncodeBatchHost(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                    uint64_t **id_offsets, uint8_t **status, uint64_t *n_failed, uint32_t **begin, uint32_t **end,
                    uint32_t **nbegin = nullptr, uint32_t **nend = nullptr, float bpe_dropout = 0.f, uint64_t seed = 0) {
  if (!h) return kInvalidArgument;
  const bool spans = begin != nullptr;
  if (!ids || !id_offsets || (spans && !end)) return Fail(h, kInternal, "output container is null");   // sentencepiece_processor.cc:367-370
  *ids = nullptr; *id_offsets = nullptr;
  if (status) *status = nullptr;
  if (n_failed) *n_failed = 0;
  if (spans) { *begin = nullptr; *end = nullptr; }
  const bool nspans = spans && nbegin && nend;
  if (nspans) { *nbegin = nullptr; *nend = nullptr; }
  if (n && !offsets) return Fail(h, kInvalidArgument, "null offsets");
  HIP_OR_RETURN(h, hipSetDevice(h->device));
  Lease L(h);
  if (int rc = L.Ready(); rc != kOk) return rc;
  Workspace *ws = L.ws.get();
  hipStream_t st = ws->stream;
  ws->bpe_dropout = bpe_dropout; ws->sample_seed = seed;
  struct Reset { Workspace *w; ~Reset() { w->bpe_dropout = 0.f; w->sample_seed = 0; } } reset{ws};
  // everything the caller gets is malloc'd here and freed on any failure
  void *outs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  auto drop = [&]() { for (void *&p : outs) { free(p); p = nullptr; } };
  uint64_t *ho = static_cast<uint64_t *>(outs[0] = malloc((n + 1) * sizeof(uint64_t)));
  if (!ho) return Fail(h, kResourceExhausted, "out of host memory");
  if (n == 0) {
    ho[0] = 0; *id_offsets = ho; *ids = static_cast<int32_t *>(malloc(sizeof(int32_t)));
    if (status) *status = static_cast<uint8_t *>(malloc(1));
    if (spans) { *begin = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); *end = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); }
    if (nspans) { *nbegin = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); *nend = static_cast<uint32_t *>(malloc(sizeof(uint32_t))); }
    return kOk;
  }
  const uint64_t base = offsets[0], text_bytes = offsets[n] - base;
  auto hip_fail = [&](hipError_t e, const char *what) { drop(); return FailHip(h, e, what); };
  hipError_t e = ws->d_text.Reserve(text_bytes + 32);
  if (e == hipSuccess) e = ws->d_offs.Reserve(n + 1);
  if (e == hipSuccess) e = ws->d_id_offs.Reserve(n + 1);
  if (e == hipSuccess && status) e = ws->d_sent_status.Reserve(n);
  if (e == hipSuccess && text_bytes) e = hipMemcpyAsync(ws->d_text.p, text + base, text_bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(ws->d_offs.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return hip_fail(e, "staging the batch");
  // offsets are used as given: the kernels address text + offs[i], so rebase the text pointer instead (they read
  // aligned 16-byte blocks of the absolute address: nothing before the staging buffer, at most 15 bytes of its slack after)
  const uint8_t *d_text = ws->d_text.p - base;
  uint64_t cap = text_bytes / 2 + 4 * n + 64, total = 0, failed = 0;
  int rc = kOk;
  for (int attempt = 0; attempt < 2; ++attempt) {
    e = ws->d_ids.Reserve(cap);
    if (e == hipSuccess && spans) e = ws->d_span_begin.Reserve(ws->d_ids.cap);
    if (e == hipSuccess && spans) e = ws->d_span_end.Reserve(ws->d_ids.cap);
    if (e == hipSuccess && nspans) e = ws->d_nspan_begin.Reserve(ws->d_ids.cap);
    if (e == hipSuccess && nspans) e = ws->d_nspan_end.Reserve(ws->d_ids.cap);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(ids)");
    rc = EncodeDevice(h, ws, d_text, text_bytes, ws->d_offs.p, n, ws->d_ids.p, ws->d_ids.cap, ws->d_id_offs.p,
                      status ? ws->d_sent_status.p : nullptr, st, &total, &failed,
                      spans ? ws->d_span_begin.p : nullptr, spans ? ws->d_span_end.p : nullptr,
                      nspans ? ws->d_nspan_begin.p : nullptr, nspans ? ws->d_nspan_end.p : nullptr);
    if (rc != kResourceExhausted || total <= ws->d_ids.cap) break;
    cap = total;
  }
  if (rc != kOk) { drop(); return rc; }
  const size_t tn = total ? total : 1;
  int32_t *hi = static_cast<int32_t *>(outs[1] = malloc(tn * sizeof(int32_t)));
  uint8_t *hs = status ? static_cast<uint8_t *>(outs[2] = malloc(n)) : nullptr;
  uint32_t *hb = spans ? static_cast<uint32_t *>(outs[3] = malloc(tn * sizeof(uint32_t))) : nullptr;
  uint32_t *he = spans ? static_cast<uint32_t *>(outs[4] = malloc(tn * sizeof(uint32_t))) : nullptr;
  uint32_t *nb = nspans ? static_cast<uint32_t *>(outs[5] = malloc(tn * sizeof(uint32_t))) : nullptr;
  uint32_t *ne = nspans ? static_cast<uint32_t *>(outs[6] = malloc(tn * sizeof(uint32_t))) : nullptr;
  if (!hi || (status && !hs) || (spans && (!hb || !he)) || (nspans && (!nb || !ne))) { drop(); return Fail(h, kResourceExhausted, "out of host memory"); }
  e = hipMemcpyAsync(ho, ws->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && total) e = hipMemcpyAsync(hi, ws->d_ids.p, total * sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && hs) e = hipMemcpyAsync(hs, ws->d_sent_status.p, n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && hb && total) e = hipMemcpyAsync(hb, ws->d_span_begin.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && he && total) e = hipMemcpyAsync(he, ws->d_span_end.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && nb && total) e = hipMemcpyAsync(nb, ws->d_nspan_begin.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && ne && total) e = hipMemcpyAsync(ne, ws->d_nspan_end.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(ids)");
  *id_offsets = ho; *ids = hi;
  if (status) *status = hs;
  if (n_failed) *n_failed = failed;
  if (spans) { *begin = hb; *end = he; }
  if (nspans) { *nbegin = nb; *nend = ne; }
  return kOk;
}
}  // namespace

namespace {
// Load()'s self-test (src/sentencepiece_processor.cc:259-278): every sample of self_test_data is encoded to pieces,
// joined with ' ' and compared with the expected string through the model's VerifyOutputsEquivalent -- equality for
// BPE (model_interface.h:194-197), equal path scores within kEpsilon for unigram (unigram_model.cc:857-888).
int RunSelfTest(spmx_handle *h) {
  const auto &samples = h->model.self_test;
  std::string text;
  std::vector<uint64_t> offs(1, 0);
  for (const auto &sm : samples) { text += sm.first; offs.push_back(text.size()); }
  const uint64_t n = samples.size();
  int32_t *ids = nullptr;
  uint64_t *io = nullptr, *no = nullptr;
  uint32_t *b = nullptr, *e = nullptr, *nb = nullptr, *ne = nullptr;
  char *norm = nullptr;
  uint8_t *st = nullptr;
  uint64_t failed = 0;
  int rc = EncodeBatchHost(h, text.data(), offs.data(), n, &ids, &io, &st, &failed, &b, &e, &nb, &ne);
  if (rc == kOk) rc = spmx_normalize_batch(h, text.data(), offs.data(), n, &norm, &no, nullptr);
  auto done = [&](int code, const std::string &msg) {
    free(ids); free(io); free(st); free(b); free(e); free(nb); free(ne); free(norm); free(no);
    return code == kOk ? kOk : Fail(h, code, msg);
  };
  if (rc != kOk) return done(rc, t_error);
  const ModelData &m = h->model;
  auto score_of = [&](const std::string &joined) -> float {          // compute_unigram_model_score
    float total = 0;
    const float unk_score = m.min_score - 10.0f;
    size_t p = 0;
    for (;;) {                                                       // absl::StrSplit(s, ' '): empty fields are kept
      const size_t q = joined.find(' ', p);
      const std::string piece = joined.substr(p, q == std::string::npos ? std::string::npos : q - p);
      const int id = m.PieceToId(piece);
      if (id == m.unk_id) total += unk_score;
      else total += m.pieces[id].type == kUserDefined ? static_cast<float>(static_cast<int>(piece.size()) * m.max_score - 0.1) : m.pieces[id].score;
      if (q == std::string::npos) break;
      p = q + 1;
    }
    return total;
  };
  size_t errors = 0;
  std::string first;
  for (uint64_t s = 0; s < n; ++s) {
    if (st[s]) return done(st[s], StatusText(st[s]));                // RETURN_IF_ERROR(Encode(s.input(), &sps))
    std::string result;
    for (uint64_t k = io[s]; k < io[s + 1]; ++k) {
      if (k > io[s]) result += ' ';
      const int type = m.pieces[ids[k]].type;
      if (type == kByte || type == kControl) result += m.pieces[ids[k]].piece;
      else result.append(norm + no[s] + nb[k], ne[k] - nb[k]);
    }
    const std::string &expected = samples[s].second;
    bool ok = expected == result;
    if (!ok && m.model_type == kUnigram) {
      const float d = score_of(expected) - score_of(result);
      ok = !((d < 0 ? -d : d) > 1e-7f);
    }
    if (!ok && errors++ == 0) first = samples[s].first + "\t" + expected + "\t" + result;
  }
  if (errors) return done(kInternal, "Self-test failures. See LOG(INFO). " + std::to_string(errors) + "/" + std::to_string(n) +
                                         " samples did not pass the test; first: " + first);
  return done(kOk, "");
}
}  // namespace

namespace {
// The host-buffer encode of a BIG batch as a pipeline: the batch is cut into chunks of host_chunk sentences and
// host_threads workers take them round-robin, each with a workspace and a stream of its own:
//   gather the chunk into pinned staging (memcpy) -> H2D -> the encode launches -> D2H straight into the (pinned,
//   pooled) output arrays at the chunk's place in the CSR -> id offsets rebased on the way out.
// The H2D of one chunk, the kernels of another and the D2H of a third overlap on the GPU, and the staging copies run
// on as many host threads.  A chunk's place in the id array is the sum of the earlier chunks' id counts: a worker
// waits for its predecessor's count (not for its copies) before it issues the D2H.
// `views` (optional, instead of text / offsets): n (pointer, length) pairs, gathered the same way.
struct spmx_view_ { const char *data; uint64_t len; };
// sentences per chunk: about eight chunks per GPU, between 128 k and 1 M sentences (measured on 10 M C2 sentences: 1 M
// chunks on 24 workers 52 ms, 256 k chunks 57 - 120 ms: the fixed cost of a call's launches and its two host syncs)
uint64_t HostChunk(const spmx_handle *h, uint64_t n, int n_h) {
  if (h->host_chunk) return h->host_chunk;
  uint64_t c = (n + 8ull * n_h - 1) / (8ull * n_h);
  // several handles (GPUs): every GPU must get chunks, so the floor drops with the batch (a 128 k floor left a batch
  // below 128 k sentences on handles[0] alone)
  const uint64_t floor_c = n_h > 1 ? (4u << 10) : (128u << 10);
  if (c < floor_c) c = floor_c;
  if (c > (1u << 20)) c = 1u << 20;
  return c;
}
// the pipeline pays from a few chunks on
bool HostPipelined(const spmx_handle *h, uint64_t n) { return n >= (h->host_chunk ? 2 * h->host_chunk : (512u << 10)); }

// `hs`: one handle per GPU of the node (the same model on each): worker w runs on hs[w % n_h], so the chunks are dealt
// round-robin over the GPUs and all of them fill ONE output CSR -- the single-process form of the multi-GPU encode
// (no collective: every GPU copies its chunks' ids to their place in the host array).
int EncodeBatchPipelined(spmx_handle *const *hs, int n_h, const char *text, const uint64_t *offsets, const spmx_view_ *views,
                         uint64_t n, int32_t **ids, uint64_t **id_offsets, uint8_t **status, uint64_t *n_failed) {
  spmx_handle *h = hs[0];
  const uint64_t chunk = HostChunk(h, n, n_h);
  const uint64_t n_chunks = (n + chunk - 1) / chunk;
  int T = h->host_threads * n_h;
  if (static_cast<uint64_t>(T) > n_chunks) T = static_cast<int>(n_chunks);
  // byte offset of every chunk's first sentence
  std::vector<uint64_t> cbeg(n_chunks + 1, 0);
  if (views) {
    for (uint64_t k = 0; k < n_chunks; ++k) {
      uint64_t b = 0;
      const uint64_t s1 = (k + 1) * chunk < n ? (k + 1) * chunk : n;
      for (uint64_t i = k * chunk; i < s1; ++i) b += views[i].len;
      cbeg[k + 1] = cbeg[k] + b;
    }
  } else {
    for (uint64_t k = 0; k <= n_chunks; ++k) cbeg[k] = offsets[k * chunk < n ? k * chunk : n] - offsets[0];
  }
  const uint64_t text_bytes = cbeg[n_chunks];
  uint64_t max_bytes = 0;                                        // the largest chunk: what every worker's workspace is sized for
  for (uint64_t k = 0; k < n_chunks; ++k) if (cbeg[k + 1] - cbeg[k] > max_bytes) max_bytes = cbeg[k + 1] - cbeg[k];
  uint64_t cap = text_bytes / 2 + 4 * n + 64;                    // ids the output array holds (more: the plain path takes over)
  int32_t *out_ids = static_cast<int32_t *>(g_pinned.Get(cap * sizeof(int32_t)));
  uint64_t *out_offs = static_cast<uint64_t *>(g_pinned.Get((n + 1) * sizeof(uint64_t)));
  uint8_t *out_st = status ? static_cast<uint8_t *>(g_pinned.Get(n)) : nullptr;
  auto drop = [&]() { if (out_ids) g_pinned.Put(out_ids); if (out_offs) g_pinned.Put(out_offs); if (out_st) g_pinned.Put(out_st); };
  if (!out_ids || !out_offs || (status && !out_st)) { drop(); return Fail(h, kResourceExhausted, "out of pinned host memory"); }
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint64_t> base(n_chunks + 1, 0);                   // ids before chunk k
  uint64_t known = 0;                                            // base[0 .. known] are final
  int first_rc = kOk;
  std::string first_err;
  bool overflow = false;
  uint64_t failed_total = 0;
  auto worker = [&](int w) {
    int rc = kOk;
    std::string err;
    spmx_handle *h = hs[w % n_h];                     // (shadows the first handle: this worker's GPU)
    auto body = [&]() -> int {
      HIP_OR_RETURN(h, hipSetDevice(h->device));
      Lease L(h);
      if (int r = L.Ready(); r != kOk) return r;
      Workspace *ws = L.ws.get();
      hipStream_t st = ws->stream;
      const uint64_t max_cnt = chunk < n ? chunk : n;
      ws->reserve_text_bytes = max_bytes; ws->reserve_n = max_cnt;
      struct ResetHint { Workspace *w; ~ResetHint() { w->reserve_text_bytes = 0; w->reserve_n = 0; } } reset_hint{ws};
      // (staging, text, offsets and ids for the LARGEST chunk, once: see Workspace::reserve_text_bytes)
      HIP_OR_RETURN(h, ws->h_text.Reserve(max_bytes + 32));
      HIP_OR_RETURN(h, ws->h_offs.Reserve(max_cnt + 1));
      HIP_OR_RETURN(h, ws->h_id_offs.Reserve(max_cnt + 1));
      HIP_OR_RETURN(h, ws->d_text.Reserve(max_bytes + 32));
      HIP_OR_RETURN(h, ws->d_offs.Reserve(max_cnt + 1));
      HIP_OR_RETURN(h, ws->d_id_offs.Reserve(max_cnt + 1));
      HIP_OR_RETURN(h, ws->d_sent_status.Reserve(max_cnt));
      HIP_OR_RETURN(h, ws->d_ids.Reserve(max_bytes / 2 + 4 * max_cnt + 64));
      for (uint64_t k = static_cast<uint64_t>(w); k < n_chunks; k += static_cast<uint64_t>(T)) {
        const uint64_t s0 = k * chunk, s1 = (k + 1) * chunk < n ? (k + 1) * chunk : n, cnt = s1 - s0;
        const uint64_t bytes = cbeg[k + 1] - cbeg[k];
        if (views) {
          uint64_t at = 0;
          for (uint64_t i = 0; i < cnt; ++i) {
            ws->h_offs.p[i] = at;
            if (views[s0 + i].len) memcpy(ws->h_text.p + at, views[s0 + i].data, views[s0 + i].len);
            at += views[s0 + i].len;
          }
          ws->h_offs.p[cnt] = at;
        } else {
          const uint64_t b0 = offsets[s0];
          if (bytes) memcpy(ws->h_text.p, text + b0, bytes);
          for (uint64_t i = 0; i <= cnt; ++i) ws->h_offs.p[i] = offsets[s0 + i] - b0;
        }
        if (bytes) HIP_OR_RETURN(h, hipMemcpyAsync(ws->d_text.p, ws->h_text.p, bytes, hipMemcpyHostToDevice, st));
        HIP_OR_RETURN(h, hipMemcpyAsync(ws->d_offs.p, ws->h_offs.p, (cnt + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        uint64_t want = bytes / 2 + 4 * cnt + 64, total = 0, failed = 0;
        int r = kOk;
        for (int attempt = 0; attempt < 2; ++attempt) {
          HIP_OR_RETURN(h, ws->d_ids.Reserve(want));
          r = EncodeDevice(h, ws, ws->d_text.p, bytes, ws->d_offs.p, cnt, ws->d_ids.p, ws->d_ids.cap, ws->d_id_offs.p,
                           ws->d_sent_status.p, st, &total, &failed);
          if (r != kResourceExhausted || total <= ws->d_ids.cap) break;
          want = total;
        }
        if (r != kOk) return r;
        uint64_t my_base = 0;
        {                                                        // the chunk's place in the CSR
          std::unique_lock<std::mutex> l(mu);
          cv.wait(l, [&] { return known >= k || first_rc != kOk; });
          if (first_rc != kOk) return kOk;                       // (another worker failed: stop quietly)
          my_base = base[k];
          base[k + 1] = my_base + total;
          known = k + 1;
          failed_total += failed;
          if (my_base + total > cap) overflow = true;
          cv.notify_all();
          if (overflow) continue;                                // (keep the counts flowing; the plain path redoes the call)
        }
        if (total) HIP_OR_RETURN(h, hipMemcpyAsync(out_ids + my_base, ws->d_ids.p, total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_id_offs.p, ws->d_id_offs.p, (cnt + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        if (out_st) HIP_OR_RETURN(h, hipMemcpyAsync(out_st + s0, ws->d_sent_status.p, cnt, hipMemcpyDeviceToHost, st));
        HIP_OR_RETURN(h, hipStreamSynchronize(st));
        for (uint64_t i = 0; i < cnt; ++i) out_offs[s0 + i] = ws->h_id_offs.p[i] + my_base;
      }
      return kOk;
    };
    try { rc = body(); } catch (const std::bad_alloc &) { rc = kResourceExhausted; t_error = "out of host memory"; } catch (...) { rc = kInternal; t_error = "unknown exception"; }
    if (rc != kOk) {
      err = t_error;
      std::lock_guard<std::mutex> l(mu);
      if (first_rc == kOk) { first_rc = rc; first_err = err; }
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  for (int w = 1; w < T; ++w) pool.emplace_back(worker, w);
  worker(0);
  for (auto &t : pool) t.join();
  if (first_rc != kOk) { drop(); return Fail(h, first_rc, first_err); }
  if (overflow) { drop(); return -1; }                           // more ids than the estimate: the caller takes the plain path
  out_offs[n] = base[n_chunks];
  *ids = out_ids;
  *id_offsets = out_offs;
  if (status) *status = out_st;
  if (n_failed) *n_failed = failed_total;
  return kOk;
}
}  // namespace

int spmx_encode_batch_ex(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                         uint64_t **id_offsets, uint8_t **status, uint64_t *n_failed) {
  return Guard(h, [&]() -> int {
    if (h && ids && id_offsets && offsets && HostPipelined(h, n)) {     // big batches: the chunk pipeline
      *ids = nullptr; *id_offsets = nullptr;
      if (status) *status = nullptr;
      if (n_failed) *n_failed = 0;
      const int rc = EncodeBatchPipelined(&h, 1, text, offsets, nullptr, n, ids, id_offsets, status, n_failed);
      if (rc != -1) return rc;
    }
    return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, status, n_failed, nullptr, nullptr);
  });
}

/* One batch over several GPUs of the node from ONE process: handles[g] holds the same model on GPU g; the batch is cut
 * into chunks that the GPUs take round-robin (several in flight per GPU), and every GPU copies its chunks' ids to their
 * place in the one output CSR.  No collective is involved: the host array is the meeting point. */
int spmx_encode_batch_multi(spmx_handle *const *handles, int n_handles, const char *text, const uint64_t *offsets, uint64_t n,
                            int32_t **ids, uint64_t **id_offsets, uint8_t **status, uint64_t *n_failed) {
  if (!handles || n_handles < 1 || !handles[0]) return kInvalidArgument;
  spmx_handle *h = handles[0];
  if (!ids || !id_offsets) return Fail(h, kInternal, "output container is null");
  for (int g = 0; g < n_handles; ++g)
    if (!handles[g] || handles[g]->model.pieces.size() != h->model.pieces.size() || handles[g]->dev.flags != h->dev.flags)
      return Fail(h, kInvalidArgument, "the handles do not hold the same model");
  return Guard(h, [&]() -> int {
    *ids = nullptr; *id_offsets = nullptr;
    if (status) *status = nullptr;
    if (n_failed) *n_failed = 0;
    if (n && !offsets) return Fail(h, kInvalidArgument, "null offsets");
    if (n >= 2) {
      // chunks small enough that every GPU gets a few
      const int rc = EncodeBatchPipelined(handles, n_handles, text, offsets, nullptr, n, ids, id_offsets, status, n_failed);
      if (rc != -1) return rc;
    }
    return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, status, n_failed, nullptr, nullptr);
  });
}

/* The same batch given as n (pointer, length) pairs -- what a std::vector<absl::string_view> holds: the sentences are
 * gathered into the staging buffers directly, chunk by chunk, without an intermediate packed copy. */
int spmx_encode_batch_views(spmx_handle *h, const spmx_view *views, uint64_t n, int32_t **ids, uint64_t **id_offsets,
                            uint8_t **status, uint64_t *n_failed) {
  if (!h) return kInvalidArgument;
  if (!ids || !id_offsets) return Fail(h, kInternal, "output container is null");
  return Guard(h, [&]() -> int {
    *ids = nullptr; *id_offsets = nullptr;
    if (status) *status = nullptr;
    if (n_failed) *n_failed = 0;
    if (n && !views) return Fail(h, kInvalidArgument, "null views");
    if (HostPipelined(h, n)) {
      const int rc = EncodeBatchPipelined(&h, 1, nullptr, nullptr, reinterpret_cast<const spmx_view_ *>(views), n, ids, id_offsets, status, n_failed);
      if (rc != -1) return rc;
    }
    std::string text;                                     // small batches (and the fallback): pack, then the plain form
    std::vector<uint64_t> offs(n + 1, 0);
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) total += views[i].len;
    text.reserve(total);
    for (uint64_t i = 0; i < n; ++i) { text.append(views[i].data, views[i].len); offs[i + 1] = text.size(); }
    return EncodeBatchHost(h, text.data(), offs.data(), n, ids, id_offsets, status, n_failed, nullptr, nullptr);
  });
}

int spmx_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                      uint64_t **id_offsets) {
  return spmx_encode_batch_ex(h, text, offsets, n, ids, id_offsets, nullptr, nullptr);
}

int spmx_encode_batch_spans_ex(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                               uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend,
                               uint8_t **status, uint64_t *n_failed) {
  if (h && (!begin || !end || (nbegin != nullptr) != (nend != nullptr))) return Fail(h, kInternal, "output container is null");
  return Guard(h, [&]() -> int { return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, status, n_failed, begin, end, nbegin, nend); });
}
int spmx_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                            uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend) {
  return spmx_encode_batch_spans_ex(h, text, offsets, n, ids, id_offsets, begin, end, nbegin, nend, nullptr, nullptr);
}

const char *spmx_status_message(int code) { return code == kOk ? "" : StatusText(code); }

int spmx_encode_batch_spans_device(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                                   uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets,
                                   uint32_t *d_begin, uint32_t *d_end, uint32_t *d_nbegin, uint32_t *d_nend, void *stream,
                                   uint64_t *total_ids) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    if (d_ids && (!d_begin || !d_end)) return Fail(h, kInvalidArgument, "null span buffers");
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    return EncodeDevice(h, L.ws.get(), static_cast<const uint8_t *>(d_text), text_bytes, d_offsets, n, d_ids, ids_capacity,
                        d_id_offsets, nullptr, static_cast<hipStream_t>(stream), total_ids, nullptr, d_begin, d_end, d_nbegin, d_nend);
  });
}

namespace {
// The lattice kernels (kernels_nbest.h), host-buffer form.  mode 0: NBestEncode; 1: Lattice::Sample(inv_theta); 2: the
// kOriginal encoder (Lattice::Viterbi).  Modes 1 and 2 give one result per sentence.
// begin / end / nbegin / nend (all or none): the spans form -- per id of every result the byte range of the input and of
// the normalized text its token covers, as spmx_encode_batch_spans gives them for the best path.
int LatticeBatchHost(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int mode, int nbest_size,
                     float inv_theta, uint64_t seed, int32_t **ids, uint64_t **id_offsets, float **scores,
                     uint64_t **result_offsets, uint32_t **begin = nullptr, uint32_t **end = nullptr,
                     uint32_t **nbegin = nullptr, uint32_t **nend = nullptr) {
  if (!ids || !id_offsets || !scores || !result_offsets) return Fail(h, kInternal, "output container is null");
  *ids = nullptr; *id_offsets = nullptr; *scores = nullptr; *result_offsets = nullptr;
  const bool spans = begin != nullptr;
  if (spans && (!end || !nbegin || !nend)) return Fail(h, kInternal, "output container is null");
  if (spans) { *begin = nullptr; *end = nullptr; *nbegin = nullptr; *nend = nullptr; }
  if (h->model.model_type != kUnigram)
    return Fail(h, kInternal, mode == 0 ? "NBestEncode is not available for the current model."     // sentencepiece_processor.cc:662
                                        : "SampleEncode is not available for the current model.");  // :690
  if (nbest_size > 1024) nbest_size = 1024;                                              // unigram_model.cc:692
  if (nbest_size < 1) nbest_size = 1;
  if (mode != 0) nbest_size = 1;
  {
    if ((mode == 0 && nbest_size == 1) || n == 0) {          // :694-696 the plain encoder, score 0.0; one result per sentence
      int32_t *i1 = nullptr;
      uint64_t *o1 = nullptr;
      const int rc = EncodeBatchHost(h, text, offsets, n, &i1, &o1, nullptr, nullptr, spans ? begin : nullptr, spans ? end : nullptr,
                                     spans ? nbegin : nullptr, spans ? nend : nullptr);
      if (rc != kOk) return rc;
      uint64_t *ro = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
      float *sc = static_cast<float *>(calloc(n + 1, sizeof(float)));
      if (!ro || !sc) {
        spmx_free(i1); spmx_free(o1); free(ro); free(sc);
        if (spans) { spmx_free(*begin); spmx_free(*end); spmx_free(*nbegin); spmx_free(*nend); *begin = *end = *nbegin = *nend = nullptr; }
        return Fail(h, kResourceExhausted, "out of host memory");
      }
      for (uint64_t s = 0; s <= n; ++s) ro[s] = s;
      *ids = i1; *id_offsets = o1; *scores = sc; *result_offsets = ro;
      return kOk;
    }
    if (!offsets) return Fail(h, kInvalidArgument, "null offsets");
    // spans form: the kernel reports token ranges in the DEVICE form of the normalized text; the reference's form of
    // it and its norm_to_orig (Normalizer::Normalize) turn them into the ranges PopulateSentencePieceText reports
    struct HostNorm {
      char *text = nullptr; uint64_t *offs = nullptr; uint32_t *n2o = nullptr;
      ~HostNorm() { spmx_free(text); spmx_free(offs); spmx_free(n2o); }
    } hn;
    if (spans) {
      const int rc = spmx_normalize_batch(h, text, offsets, n, &hn.text, &hn.offs, &hn.n2o);
      if (rc != kOk) return rc;
    }
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    Workspace *ws = L.ws.get();
    hipStream_t st = ws->stream;
    const uint64_t base = offsets[0], text_bytes = offsets[n] - base;
    HIP_OR_RETURN(h, ws->d_text.Reserve(text_bytes + 32));
    HIP_OR_RETURN(h, ws->d_offs.Reserve(n + 1));
    HIP_OR_RETURN(h, ws->d_id_offs.Reserve(n + 1));
    if (text_bytes) HIP_OR_RETURN(h, hipMemcpyAsync(ws->d_text.p, text + base, text_bytes, hipMemcpyHostToDevice, st));
    HIP_OR_RETURN(h, hipMemcpyAsync(ws->d_offs.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const uint8_t *d_text = ws->d_text.p - base;
    uint64_t ncap = 2 * text_bytes + 4 * n + 64, ntotal = 0;
    int rc = kOk;
    for (int attempt = 0; attempt < 2; ++attempt) {
      HIP_OR_RETURN(h, ws->d_norm.Reserve(ncap));
      rc = NormalizeDevice(h, ws, d_text, ws->d_offs.p, n, ws->d_norm.p, ws->d_norm.cap, ws->d_id_offs.p, nullptr, st, &ntotal, true);
      if (rc != kResourceExhausted || ntotal <= ws->d_norm.cap) break;
      ncap = ntotal;
    }
    if (rc != kOk) return rc;
    const uint32_t K = static_cast<uint32_t>(nbest_size);
    if (n * K + 1 > (1ull << 32)) return Fail(h, kInvalidArgument, "too many results for one batch");
    NBestArgs a{};
    a.dev = h->dev; a.norm = ws->d_norm.p; a.norm_offs = ws->d_id_offs.p; a.n = static_cast<uint32_t>(n); a.nbest = K;
    a.mode = static_cast<uint32_t>(mode); a.inv_theta = inv_theta; a.seed = seed;
    const uint64_t hyps0 = static_cast<uint64_t>(K) * 512;   // (a C2 sentence's n-best 5 makes ~1,000; what outgrows the slice runs again)
    // The first launch's capacities follow the batch: a slice sized for 1024 bytes and 16384 nodes is 0.7 MB a lane, of
    // which a 126-byte sentence touches a twentieth, and the lanes in flight are what the budget holds (200 k sentences,
    // n-best 5: 108 ms a call at 697 wavefronts, 94 at 2048).  A sentence that normalizes to more than the guess is set
    // aside like one beyond the fixed capacities and runs in the wide launch.
    uint64_t max_raw = 0;
    for (uint64_t s = 0; s < n; ++s) max_raw = std::max<uint64_t>(max_raw, offsets[s + 1] - offsets[s]);
    uint64_t len0 = (max_raw + max_raw / 4 + 16 + 63) & ~static_cast<uint64_t>(63);
    if (len0 > kNbMaxLen) len0 = kNbMaxLen;
    uint64_t nodes0 = (len0 + 2) * (static_cast<uint64_t>(h->tables.max_prefixes) + 1) + 2;
    if (nodes0 > kNbMaxNodes) nodes0 = kNbMaxNodes;
    const uint64_t hyps_min = h->nbest_hyps_min;
    const uint32_t max_hyps0 = mode != 0 ? 512u : static_cast<uint32_t>(hyps0 < hyps_min ? hyps_min : (hyps0 > 262144 ? 262144 : hyps0));
    const uint64_t budget = h->nbest_budget;                  // HBM for the lanes' slices
    HIP_OR_RETURN(h, ws->d_res_off.Reserve(n * K + 1));
    HIP_OR_RETURN(h, ws->d_span_begin.Reserve(n * K + 1));     // result lengths
    HIP_OR_RETURN(h, ws->d_res_score.Reserve(n * K + 1));
    HIP_OR_RETURN(h, ws->d_counts.Reserve(n + 1));
    HIP_OR_RETURN(h, ws->d_lists.Reserve(2 * n + 2));          // the sentences beyond a launch's capacities (two lists, in turn)
    a.res_off = ws->d_res_off.p; a.res_len = ws->d_span_begin.p; a.res_score = ws->d_res_score.p; a.res_count = ws->d_counts.p;
    a.status = &ws->d_ctrl->status; a.arena_head = &ws->d_ctrl->arena_head;
    uint64_t arena_need = (ntotal + (4 + static_cast<uint64_t>(h->dev.n_prefix + h->dev.n_suffix)) * n) * (K < 8 ? K : 8) + 1024;
    for (int attempt = 0; attempt < 12; ++attempt) {
      HIP_OR_RETURN(h, ws->d_arena.Reserve(arena_need));
      a.arena = ws->d_arena.p; a.arena_cap = ws->d_arena.cap;
      if (spans) {
        HIP_OR_RETURN(h, ws->d_arena_tb.Reserve(ws->d_arena.cap));
        HIP_OR_RETURN(h, ws->d_tok_begin.Reserve(ws->d_arena.cap));
        a.arena_nb = ws->d_arena_tb.p; a.arena_ne = ws->d_tok_begin.p;
      }
      HIP_OR_RETURN(h, hipMemsetAsync(ws->d_ctrl, 0, sizeof(Ctrl), st));
      // first launch: every sentence, 16-bit lattice indices, capacities for the batch (at most kNbMaxLen / kNbMaxNodes)
      a.max_len = static_cast<uint32_t>(len0); a.max_nodes = static_cast<uint32_t>(nodes0); a.max_hyps = max_hyps0;
      a.lane_bytes = NbestLaneBytes(a.max_len, a.max_nodes, a.max_hyps, 2);
      a.list = nullptr; a.n_list = 0;
      a.retry_list = ws->d_lists.p; a.retry_count = &ws->d_ctrl->retry_count[0]; a.retry_max_len = &ws->d_ctrl->side.over_max_raw;
      uint64_t waves = (n + 63) / 64;
      if (waves * 64 * a.lane_bytes > budget) waves = budget / (64 * a.lane_bytes);
      if (waves > static_cast<uint64_t>(h->n_cu) * 8) waves = static_cast<uint64_t>(h->n_cu) * 8;
      if (waves < 1) waves = 1;
      HIP_OR_RETURN(h, ws->d_nbest_scratch.Reserve(waves * 64 * a.lane_bytes));
      a.scratch = ws->d_nbest_scratch.p;
      HIP_OR_RETURN(h, LaunchNBest(false, a, static_cast<int>(waves), st));
      HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl, ws->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, st));
      HIP_OR_RETURN(h, hipStreamSynchronize(st));
      // further launches: 32-bit indices, capacities for the longest sentence set aside; a sentence whose A* needs more
      // hypotheses than its slice holds is set aside again and the slice grows (the reference's agenda is unbounded too)
      uint64_t hy_scale = 1;
      for (int round = 0; ws->h_ctrl->retry_count[round & 1] && !(ws->h_ctrl->status & kStArenaOverflow); ++round) {
        const uint32_t n_retry = ws->h_ctrl->retry_count[round & 1];
        // A position starts at most max_prefixes pieces (the deepest chain of pieces that are prefixes of one another)
        // and one UNK node.
        const uint64_t L = ws->h_ctrl->side.over_max_raw > kNbMaxLen ? ws->h_ctrl->side.over_max_raw : kNbMaxLen;
        const uint64_t nodes = (L + 2) * (static_cast<uint64_t>(h->tables.max_prefixes) + 1) + 2;
        uint64_t hy = mode != 0 ? L / 4 + 512 : (static_cast<uint64_t>(K) * (L + 2) * 4 + 16384) * hy_scale;   // mode 1: alpha[L + 1] floats
        if (hy < max_hyps0 * hy_scale) hy = max_hyps0 * hy_scale;
        if (L >= (1ull << 31) || nodes >= (1ull << 32) || hy >= (1ull << 32)) return Fail(h, kOutOfRange, "a sentence is too long for the lattice");
        a.max_len = static_cast<uint32_t>(L); a.max_nodes = static_cast<uint32_t>(nodes); a.max_hyps = static_cast<uint32_t>(hy);
        a.lane_bytes = NbestLaneBytes(a.max_len, a.max_nodes, a.max_hyps, 4);
        uint64_t lanes = budget / a.lane_bytes;
        if (lanes < 1) return Fail(h, kResourceExhausted, "out of device memory for the lattice of a sentence of " + std::to_string(L) + " normalized bytes");
        if (lanes > n_retry) lanes = n_retry;
        uint64_t w2 = lanes / 64;                           // whole waves within the budget ...
        const bool partial = w2 == 0;                       // ... or one partial wave: only its first `lanes` lanes own a slice
        if (partial) w2 = 1;
        if (w2 > static_cast<uint64_t>(h->n_cu) * 8) w2 = static_cast<uint64_t>(h->n_cu) * 8;
        a.list = ws->d_lists.p + static_cast<size_t>(round & 1) * n; a.n_list = n_retry;
        a.retry_list = ws->d_lists.p + static_cast<size_t>((round + 1) & 1) * n;
        a.retry_count = &ws->d_ctrl->retry_count[(round + 1) & 1];
        a.retry_max_len = &ws->d_ctrl->side.over_max_raw;
        HIP_OR_RETURN(h, hipMemsetAsync(a.retry_count, 0, sizeof(uint32_t), st));
        const uint64_t slices = partial ? lanes : w2 * 64;
        HIP_OR_RETURN(h, ws->d_nbest_scratch.Reserve(slices * a.lane_bytes));
        a.scratch = ws->d_nbest_scratch.p;
        a.live_lanes = static_cast<uint32_t>(slices);
        HIP_OR_RETURN(h, LaunchNBest(true, a, static_cast<int>(w2), st));
        a.live_lanes = 0;
        HIP_OR_RETURN(h, hipMemcpyAsync(ws->h_ctrl, ws->d_ctrl, offsetof(Ctrl, total_ids), hipMemcpyDeviceToHost, st));
        HIP_OR_RETURN(h, hipStreamSynchronize(st));
        hy_scale *= 4;
        if (round == 9 && ws->h_ctrl->retry_count[(round + 1) & 1])
          return Fail(h, kResourceExhausted, "NBestEncode: the agenda of a sentence exceeds the device capacities");
      }
      const uint32_t stw = ws->h_ctrl->status;
      if (stw & kStNbestOverflow) return Fail(h, kResourceExhausted, "NBestEncode: the agenda of a sentence exceeds the device capacities");
      // (a lane stops at its first result that does not fit, so arena_head is a lower bound: grow geometrically)
      if (stw & kStArenaOverflow) { arena_need = 4 * ws->d_arena.cap > ws->h_ctrl->arena_head + 1024 ? 4 * ws->d_arena.cap : ws->h_ctrl->arena_head + 1024; continue; }
      // results -> host CSR.  The result arrays land in ONE pinned staging block (the workspace's h_text, idle in this
      // call) by asynchronous copies behind one synchronisation -- copies into fresh pageable vectors and a single-thread
      // assembly were 100 of a 178 ms call of 200 k sentences x 5 results beside 76 ms of kernel -- and the CSR is put
      // together by a few threads over ranges of sentences (where a sentence's results go follows from a prefix pass).
      const uint64_t used = ws->h_ctrl->arena_head;
      auto al16 = [](uint64_t x) { return (x + 15u) & ~static_cast<uint64_t>(15); };
      const uint64_t b_cnt = 0, b_len = b_cnt + al16(n * 4), b_off = b_len + al16(n * K * 4), b_sc = b_off + al16(n * K * 8),
                     b_ar = b_sc + al16(n * K * 4), b_nb = b_ar + al16(used * 4), b_ne = b_nb + (spans ? al16(used * 4) : 0),
                     b_do = b_ne + (spans ? al16(used * 4) : 0), b_end = b_do + (spans ? al16((n + 1) * 8) : 0);
      HIP_OR_RETURN(h, ws->h_text.Reserve(b_end + 16));
      uint8_t *stg = ws->h_text.p;
      HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_cnt, ws->d_counts.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_len, ws->d_span_begin.p, n * K * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_off, ws->d_res_off.p, n * K * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_sc, ws->d_res_score.p, n * K * sizeof(float), hipMemcpyDeviceToHost, st));
      if (used) HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_ar, ws->d_arena.p, used * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      if (spans) {
        if (used) {
          HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_nb, ws->d_arena_tb.p, used * sizeof(int32_t), hipMemcpyDeviceToHost, st));
          HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_ne, ws->d_tok_begin.p, used * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        }
        HIP_OR_RETURN(h, hipMemcpyAsync(stg + b_do, ws->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      }
      HIP_OR_RETURN(h, hipStreamSynchronize(st));
      const uint32_t *cnt = reinterpret_cast<const uint32_t *>(stg + b_cnt), *len = reinterpret_cast<const uint32_t *>(stg + b_len);
      const unsigned long long *off = reinterpret_cast<const unsigned long long *>(stg + b_off);
      const float *sc = reinterpret_cast<const float *>(stg + b_sc);
      const int32_t *arena = reinterpret_cast<const int32_t *>(stg + b_ar);
      const int32_t *arena_nb = reinterpret_cast<const int32_t *>(stg + b_nb), *arena_ne = reinterpret_cast<const int32_t *>(stg + b_ne);
      const uint64_t *dev_offs = reinterpret_cast<const uint64_t *>(stg + b_do);   // device-form normalized offsets (a consistency check of the mapping)
      uint64_t *hr = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
      std::vector<uint64_t> t_of(n + 1);                    // where sentence s's ids start
      if (!hr) return Fail(h, kResourceExhausted, "out of host memory");
      uint64_t R = 0, total = 0;
      for (uint64_t s = 0; s < n; ++s) {
        hr[s] = R; t_of[s] = total;
        for (uint32_t k = 0; k < cnt[s]; ++k) total += len[s * K + k];
        R += cnt[s];
      }
      hr[n] = R; t_of[n] = total;
      // (the ids: pinned and recycled through spmx_free when they are many -- fresh pageable pages cost a fault each)
      const size_t hi_bytes = (total ? total : 1) * sizeof(int32_t);
      int32_t *hi = static_cast<int32_t *>(hi_bytes >= (8u << 20) ? g_pinned.Get(hi_bytes) : malloc(hi_bytes));
      uint64_t *ho = static_cast<uint64_t *>(malloc((R + 1) * sizeof(uint64_t)));
      float *hs = static_cast<float *>(malloc((R ? R : 1) * sizeof(float)));
      uint32_t *sb = nullptr, *se = nullptr, *snb = nullptr, *sne = nullptr;
      if (spans) {
        const size_t bytes = (total ? total : 1) * sizeof(uint32_t);
        sb = static_cast<uint32_t *>(malloc(bytes)); se = static_cast<uint32_t *>(malloc(bytes));
        snb = static_cast<uint32_t *>(malloc(bytes)); sne = static_cast<uint32_t *>(malloc(bytes));
      }
      if (!hi || !ho || !hs || (spans && (!sb || !se || !snb || !sne))) {
        spmx_free(hi); free(ho); free(hs); free(hr); free(sb); free(se); free(snb); free(sne);
        return Fail(h, kResourceExhausted, "out of host memory");
      }
      const bool one = (h->dev.flags & kNfCompressSp) != 0;
      std::atomic<bool> map_bad{false};
      auto do_range = [&](uint64_t s0, uint64_t s1) {
        std::vector<uint32_t> ref_of_dev;                   // position in the reference's normalized text of device byte x
        bool map_ok = true;
        for (uint64_t s = s0; s < s1; ++s) {
          uint64_t r = hr[s], t = t_of[s];
          const char *R_ = nullptr;
          const uint32_t *n2o = nullptr;
          uint64_t rlen = 0;
          if (spans) {
            R_ = hn.text + hn.offs[s];
            rlen = hn.offs[s + 1] - hn.offs[s];
            n2o = hn.n2o + hn.offs[s] + s;
            ref_of_dev.clear();
            for (uint64_t i = 0; i < rlen;) {
              ref_of_dev.push_back(static_cast<uint32_t>(i));
              const bool sp3 = one && i + 2 < rlen && static_cast<unsigned char>(R_[i]) == 0xE2u &&
                               static_cast<unsigned char>(R_[i + 1]) == 0x96u && static_cast<unsigned char>(R_[i + 2]) == 0x81u;
              i += sp3 ? 3 : 1;
            }
            ref_of_dev.push_back(static_cast<uint32_t>(rlen));
            if (ref_of_dev.size() - 1 != dev_offs[s + 1] - dev_offs[s]) map_ok = false;
          }
          const uint32_t in_len = static_cast<uint32_t>(offsets[s + 1] - offsets[s]);
          for (uint32_t k = 0; k < cnt[s]; ++k) {
            ho[r] = t;
            hs[r] = sc[s * K + k];
            const uint32_t ln = len[s * K + k];
            if (ln) memcpy(hi + t, arena + off[s * K + k], ln * sizeof(int32_t));
            if (spans && map_ok) {
              const int32_t *pnb = arena_nb + off[s * K + k], *pne = arena_ne + off[s * K + k];
              for (uint32_t j = 0; j < ln; ++j) {
                if (pnb[j] < 0) {                           // bos / eos (sentencepiece_processor.cc:1029-1048)
                  sb[t + j] = se[t + j] = pnb[j] == -1 ? in_len : 0u;
                  snb[t + j] = sne[t + j] = 0u;
                  continue;
                }
                if (static_cast<size_t>(pne[j]) >= ref_of_dev.size() || pnb[j] > pne[j]) { map_ok = false; break; }
                const uint32_t rb = ref_of_dev[pnb[j]], re = ref_of_dev[pne[j]];
                snb[t + j] = rb; sne[t + j] = re;
                sb[t + j] = n2o[rb]; se[t + j] = n2o[re];   // :566-574
              }
            }
            t += ln;
            ++r;
          }
          if (!map_ok) break;
        }
        if (!map_ok) map_bad.store(true);
      };
#ifdef SPMX_TEST_SEAMS
      const uint64_t kThreadedFrom = 64;                    // (the emulated library takes the threaded form in its tests)
#else
      const uint64_t kThreadedFrom = 1u << 20;
#endif
      unsigned T = total + R >= kThreadedFrom ? std::thread::hardware_concurrency() : 1u;
      if (T > 16u) T = 16u;
      if (T < 1u) T = 1u;
      if (T == 1u) {
        do_range(0, n);
      } else {                                              // ranges of about equal id counts
        std::vector<std::thread> pool;
        uint64_t s0 = 0;
        for (unsigned w = 0; w < T; ++w) {
          const uint64_t want = total / T * (w + 1);
          uint64_t s1 = w + 1 == T ? n : static_cast<uint64_t>(std::upper_bound(t_of.begin(), t_of.begin() + n, want) - t_of.begin());
          if (s1 < s0) s1 = s0;
          if (s1 > s0) pool.emplace_back(do_range, s0, s1);
          s0 = s1;
        }
        for (auto &t : pool) t.join();
      }
      ho[R] = total;
      const bool map_ok = !map_bad.load();
      if (!map_ok) {
        spmx_free(hi); free(ho); free(hs); free(hr); free(sb); free(se); free(snb); free(sne);
        return Fail(h, kInternal, "token ranges do not map onto the normalized text");
      }
      *ids = hi; *id_offsets = ho; *scores = hs; *result_offsets = hr;
      if (spans) { *begin = sb; *end = se; *nbegin = snb; *nend = sne; }
      return kOk;
    }
    return Fail(h, kInternal, "id arena kept overflowing");
  }
}
}  // namespace

// NBestEncode (src/sentencepiece_processor.h:323-324; unigram::Model::NBestEncode src/unigram_model.cc:686-717).
int spmx_nbest_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                            int32_t **ids, uint64_t **id_offsets, float **scores, uint64_t **result_offsets) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int { return LatticeBatchHost(h, text, offsets, n, 0, nbest_size, 0.f, 0, ids, id_offsets, scores, result_offsets); });
}

// NBestEncode(input, nbest_size, NBestSentencePieceText *) (src/sentencepiece_processor.cc:653-676): every result with the
// byte ranges of its pieces.
int spmx_nbest_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                                  int32_t **ids, uint64_t **id_offsets, float **scores, uint64_t **result_offsets,
                                  uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend) {
  if (!h) return kInvalidArgument;
  if (!begin || !end || !nbegin || !nend) return Fail(h, kInternal, "output container is null");
  return Guard(h, [&]() -> int {
    return LatticeBatchHost(h, text, offsets, n, 0, nbest_size, 0.f, 0, ids, id_offsets, scores, result_offsets, begin, end, nbegin, nend);
  });
}

// The kOriginal unigram encoder (unigram::Model::Encode with EncoderVersion::kOriginal, src/unigram_model.cc:674-692:
// Lattice::SetSentence + PopulateNodes + Lattice::Viterbi), per sentence.
int spmx_encode_batch_original(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                               uint64_t **id_offsets) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    float *sc = nullptr;
    uint64_t *ro = nullptr;
    const int rc = LatticeBatchHost(h, text, offsets, n, 2, 1, 0.f, 0, ids, id_offsets, &sc, &ro);
    free(sc);
    free(ro);
    return rc;
  });
}

// SampleEncode(input, nbest_size, alpha, &ids) (src/sentencepiece_processor.cc:678-720) per sentence:
//   nbest_size < 0      unigram: Lattice::Sample(alpha) (forward filtering, backward sampling, :511-542); BPE: BPE-dropout
//                       with probability alpha (src/bpe_model.cc:131-156)
//   nbest_size 0 or 1   the plain encoder
//   nbest_size > 1      one of the nbest_size best segmentations, drawn with probability proportional to
//                       exp(alpha * score) (:700-716; unigram only)
// The draws come from generators keyed by (seed, sentence index): the reference's thread-local mt19937 stream is not
// reproduced (its own tests pin SampleEncode statistically, src/unigram_model_test.cc:429-470, bpe_model_test.cc:252-295).
namespace {
int SampleEncodeImpl(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size, float alpha,
                     uint64_t seed, int32_t **ids, uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin,
                     uint32_t **nend);
}
int spmx_sample_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                             float alpha, uint64_t seed, int32_t **ids, uint64_t **id_offsets) {
  return SampleEncodeImpl(h, text, offsets, n, nbest_size, alpha, seed, ids, id_offsets, nullptr, nullptr, nullptr, nullptr);
}
// SampleEncode(input, nbest_size, alpha, SentencePieceText *) (src/sentencepiece_processor.cc:678-720): the drawn
// segmentation with the byte ranges of its pieces.
int spmx_sample_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                                   float alpha, uint64_t seed, int32_t **ids, uint64_t **id_offsets, uint32_t **begin,
                                   uint32_t **end, uint32_t **nbegin, uint32_t **nend) {
  if (h && (!begin || !end || !nbegin || !nend)) return Fail(h, kInternal, "output container is null");
  return SampleEncodeImpl(h, text, offsets, n, nbest_size, alpha, seed, ids, id_offsets, begin, end, nbegin, nend);
}
namespace {
int SampleEncodeImpl(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size, float alpha,
                     uint64_t seed, int32_t **ids, uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin,
                     uint32_t **nend) {
  if (!h) return kInvalidArgument;
  if (!ids || !id_offsets) return Fail(h, kInternal, "output container is null");
  const bool spans = begin != nullptr;
  return Guard(h, [&]() -> int {
    *ids = nullptr; *id_offsets = nullptr;
    if (spans) { *begin = nullptr; *end = nullptr; *nbegin = nullptr; *nend = nullptr; }
    if (nbest_size > 512) return Fail(h, kInternal, "nbest_size must be nbest_size <= 512");   // sentencepiece_processor.cc:684
    if (h->model.model_type == kBpe) {
      // !IsNBestEncodeAvailable(): every nbest_size goes to bpe::Model::SampleEncode(normalized, alpha) (:688-693),
      // BPE-dropout with merge-skip probability alpha; alpha <= 0 is the plain merge order (bpe_model.cc:131-156)
      if (n && !offsets) return Fail(h, kInvalidArgument, "null offsets");
      if (n == 0 || !(alpha > 0.f)) return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, nullptr, nullptr, begin, end, nbegin, nend);
      return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, nullptr, nullptr, begin, end, nbegin, nend, alpha, seed);
    }
    if (nbest_size == 0 || nbest_size == 1) return EncodeBatchHost(h, text, offsets, n, ids, id_offsets, nullptr, nullptr, begin, end, nbegin, nend);
    float *sc = nullptr;
    uint64_t *ro = nullptr;
    if (nbest_size < 0) {
      const int rc = LatticeBatchHost(h, text, offsets, n, 1, 1, alpha, seed, ids, id_offsets, &sc, &ro, begin, end, nbegin, nend);
      free(sc);
      free(ro);
      return rc;
    }
    int32_t *nids = nullptr;
    uint64_t *nio = nullptr;
    uint32_t *nb4[4] = {nullptr, nullptr, nullptr, nullptr};     // the spans of every n-best result
    const int rc = spans ? LatticeBatchHost(h, text, offsets, n, 0, nbest_size, 0.f, 0, &nids, &nio, &sc, &ro, &nb4[0], &nb4[1], &nb4[2], &nb4[3])
                         : LatticeBatchHost(h, text, offsets, n, 0, nbest_size, 0.f, 0, &nids, &nio, &sc, &ro);
    if (rc != kOk) return rc;
    struct Release {                               // (the n-best arrays, whatever way this function is left)
      void *p[8];
      ~Release() { for (void *q : p) spmx_free(q); }
    } release{{nids, nio, sc, ro, nb4[0], nb4[1], nb4[2], nb4[3]}};
    // one of each sentence's results, with probability exp(alpha * score) / Z
    std::vector<uint64_t> pick(n);
    uint64_t *oo = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
    uint64_t total = 0;
    for (uint64_t s2 = 0; oo && s2 < n; ++s2) {
      const uint64_t r0 = ro[s2], r1 = ro[s2 + 1];
      double mx = -1e300, z = 0.0;
      for (uint64_t r = r0; r < r1; ++r) mx = std::max(mx, static_cast<double>(alpha) * sc[r]);
      for (uint64_t r = r0; r < r1; ++r) z += exp(static_cast<double>(alpha) * sc[r] - mx);
      // keyed by (seed, sentence) with separate multipliers, as the device generators are (kernels_nbest.h): the
      // draws of (seed, i) and (seed + 1, i - 1) are unrelated
      unsigned long long x = seed * 0x9E3779B97F4A7C15ull + (s2 + 1) * 0xD1B54A32D192ED03ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      x ^= x >> 31;
      double u = static_cast<double>(x >> 11) * (1.0 / 9007199254740992.0) * z;
      uint64_t c = r1 > r0 ? r1 - 1 : r0;
      for (uint64_t r = r0; r < r1; ++r) { u -= exp(static_cast<double>(alpha) * sc[r] - mx); if (u < 0.0) { c = r; break; } }
      pick[s2] = c;
      oo[s2] = total;
      if (r1 > r0) total += nio[c + 1] - nio[c];
    }
    int32_t *oi = static_cast<int32_t *>(malloc((total ? total : 1) * sizeof(int32_t)));
    uint32_t *os[4] = {nullptr, nullptr, nullptr, nullptr};
    bool os_ok = true;
    if (spans) for (auto &q : os) { q = static_cast<uint32_t *>(malloc((total ? total : 1) * sizeof(uint32_t))); os_ok = os_ok && q; }
    if (!oo || !oi || !os_ok) { free(oo); free(oi); for (auto q : os) free(q); return Fail(h, kResourceExhausted, "out of host memory"); }
    oo[n] = total;
    for (uint64_t s2 = 0; s2 < n; ++s2) {
      if (ro[s2 + 1] <= ro[s2]) continue;
      const uint64_t from = nio[pick[s2]], cnt = nio[pick[s2] + 1] - from;
      memcpy(oi + oo[s2], nids + from, cnt * sizeof(int32_t));
      if (spans) for (int q = 0; q < 4; ++q) memcpy(os[q] + oo[s2], nb4[q] + from, cnt * sizeof(uint32_t));
    }
    *ids = oi;
    *id_offsets = oo;
    if (spans) { *begin = os[0]; *end = os[1]; *nbegin = os[2]; *nend = os[3]; }
    return kOk;
  });
}
}  // namespace

int spmx_normalize_batch_device(spmx_handle *h, const void *d_text, const uint64_t *d_offsets, uint64_t n, void *d_norm,
                                uint64_t norm_capacity, uint64_t *d_norm_offsets, uint32_t *d_norm_to_orig, void *stream,
                                uint64_t *total_bytes) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    return NormalizeDevice(h, L.ws.get(), static_cast<const uint8_t *>(d_text), d_offsets, n, static_cast<uint8_t *>(d_norm),
                           norm_capacity, d_norm_offsets, d_norm_to_orig, static_cast<hipStream_t>(stream), total_bytes);
  });
}

int spmx_normalize_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, char **norm,
                         uint64_t **norm_offsets, uint32_t **norm_to_orig) {
  if (!h) return kInvalidArgument;
  if (!norm || !norm_offsets) return Fail(h, kInternal, "output container is null");
  *norm = nullptr; *norm_offsets = nullptr;
  if (norm_to_orig) *norm_to_orig = nullptr;
  if (n && !offsets) return Fail(h, kInvalidArgument, "null offsets");
  return Guard(h, [&]() -> int {
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    Workspace *ws = L.ws.get();
    hipStream_t st = ws->stream;
    uint64_t *ho = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
    if (!ho) return Fail(h, kResourceExhausted, "out of host memory");
    if (n == 0) {
      ho[0] = 0; *norm_offsets = ho; *norm = static_cast<char *>(malloc(1));
      if (norm_to_orig) *norm_to_orig = static_cast<uint32_t *>(malloc(sizeof(uint32_t)));
      return kOk;
    }
    const uint64_t base = offsets[0], text_bytes = offsets[n] - base;
    hipError_t e = ws->d_text.Reserve(text_bytes + 32);
    if (e == hipSuccess) e = ws->d_offs.Reserve(n + 1);
    if (e == hipSuccess) e = ws->d_id_offs.Reserve(n + 1);
    if (e == hipSuccess && text_bytes) e = hipMemcpyAsync(ws->d_text.p, text + base, text_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ws->d_offs.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { free(ho); return FailHip(h, e, "staging the batch"); }
    const uint8_t *d_text = ws->d_text.p - base;
    uint64_t cap = 2 * text_bytes + 4 * n + 64, total = 0;
    int rc = kOk;
    for (int attempt = 0; attempt < 2; ++attempt) {
      e = ws->d_norm.Reserve(cap);
      if (e == hipSuccess && norm_to_orig) e = ws->d_span_begin.Reserve(cap + n + 1);
      if (e != hipSuccess) { free(ho); return FailHip(h, e, "hipMalloc(norm)"); }
      rc = NormalizeDevice(h, ws, d_text, ws->d_offs.p, n, ws->d_norm.p, ws->d_norm.cap, ws->d_id_offs.p,
                           norm_to_orig ? ws->d_span_begin.p : nullptr, st, &total);
      if (rc != kResourceExhausted || total <= ws->d_norm.cap) break;
      cap = total;
    }
    if (rc != kOk) { free(ho); return rc; }
    char *ht = static_cast<char *>(malloc(total ? total : 1));
    uint32_t *hn = norm_to_orig ? static_cast<uint32_t *>(malloc((total + n + 1) * sizeof(uint32_t))) : nullptr;
    if (!ht || (norm_to_orig && !hn)) { free(ho); free(ht); free(hn); return Fail(h, kResourceExhausted, "out of host memory"); }
    e = hipMemcpyAsync(ho, ws->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && total) e = hipMemcpyAsync(ht, ws->d_norm.p, total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && hn) e = hipMemcpyAsync(hn, ws->d_span_begin.p, (total + n) * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(ho); free(ht); free(hn); return FailHip(h, e, "hipMemcpy(norm)"); }
    *norm = ht;
    *norm_offsets = ho;
    if (norm_to_orig) *norm_to_orig = hn;
    return kOk;
  });
}

void spmx_free(void *p) {
  if (!p) return;
  if (!g_pinned.Put(p)) free(p);
}

int spmx_encode(spmx_handle *h, const char *text, uint64_t len, int32_t *ids, uint64_t cap, uint64_t *n_ids) {
  if (!h) return kInvalidArgument;
  if (!n_ids || (!ids && cap)) return Fail(h, kInternal, "output container is null");
  const uint64_t offs[2] = {0, len};
  int32_t *out = nullptr;
  uint64_t *oo = nullptr;
  uint8_t *st = nullptr;
  uint64_t failed = 0;
  const int rc = spmx_encode_batch_ex(h, text ? text : "", offs, 1, &out, &oo, &st, &failed);
  if (rc != kOk) return rc;
  const uint64_t total = oo[1];
  *n_ids = total;
  int ret = kOk;
  if (failed && st[0]) ret = Fail(h, st[0], StatusText(st[0]));           // Encode(input, &ids) returns the Status
  else if (total > cap) ret = Fail(h, kResourceExhausted, "ids buffer is too small");
  else if (total) memcpy(ids, out, total * sizeof(int32_t));
  free(out);
  free(oo);
  free(st);
  return ret;
}

int spmx_decode_batch_device(spmx_handle *h, const int32_t *d_ids, const uint64_t *d_id_offsets, uint64_t n, void *d_text,
                             uint64_t text_capacity, uint64_t *d_text_offsets, void *stream, uint64_t *total_bytes) {
  if (!h) return kInvalidArgument;
  return Guard(h, [&]() -> int {
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    return DecodeDevice(h, L.ws.get(), d_ids, d_id_offsets, n, static_cast<uint8_t *>(d_text), text_capacity, d_text_offsets,
                        static_cast<hipStream_t>(stream), total_bytes);
  });
}

namespace {
int DecodeBatchHost(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, const char *lit_bytes,
                    const uint64_t *lit_offsets, uint64_t n_lit, char **text, uint64_t **text_offsets);
}
int spmx_decode_batch(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, char **text,
                      uint64_t **text_offsets) {
  return DecodeBatchHost(h, ids, id_offsets, n, nullptr, nullptr, 0, text, text_offsets);
}
int spmx_decode_batch_pieces(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, const char *lit_bytes,
                             const uint64_t *lit_offsets, uint64_t n_lit, char **text, uint64_t **text_offsets) {
  if (h && n_lit && (!lit_offsets || (lit_offsets[n_lit] && !lit_bytes))) return Fail(h, kInvalidArgument, "null literal pieces");
  return DecodeBatchHost(h, ids, id_offsets, n, lit_bytes, lit_offsets, n_lit, text, text_offsets);
}
int spmx_piece_score(const spmx_handle *h, int id, float *score) {
  if (!h || !score || id < 0 || static_cast<size_t>(id) >= h->model.pieces.size()) return kOutOfRange;
  *score = h->model.pieces[static_cast<size_t>(id)].score;
  return kOk;
}
int spmx_decode_unk_option(const spmx_handle *h) { return h && h->dx_unk ? 1 : 0; }
int spmx_serialized_model(const spmx_handle *h, const char **data, uint64_t *n_bytes) {
  if (!h || !data || !n_bytes) return kInvalidArgument;
  *data = h->serialized.data();
  *n_bytes = h->serialized.size();
  return kOk;
}
namespace {
int DecodeBatchHost(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, const char *lit_bytes,
                    const uint64_t *lit_offsets, uint64_t n_lit, char **text, uint64_t **text_offsets) {
  if (!h) return kInvalidArgument;
  if (!text || !text_offsets) return Fail(h, kInternal, "output container is null");
  *text = nullptr; *text_offsets = nullptr;
  if (n && !id_offsets) return Fail(h, kInvalidArgument, "null offsets");
  return Guard(h, [&]() -> int {
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    Workspace *ws = L.ws.get();
    hipStream_t st = ws->stream;
    uint64_t *ho = static_cast<uint64_t *>(malloc((n + 1) * sizeof(uint64_t)));
    if (!ho) return Fail(h, kResourceExhausted, "out of host memory");
    if (n == 0) { ho[0] = 0; *text_offsets = ho; *text = static_cast<char *>(malloc(1)); return kOk; }
    const uint64_t base = id_offsets[0], n_ids = id_offsets[n] - base;
    hipError_t e = ws->d_ids.Reserve(n_ids + 16);
    if (e == hipSuccess) e = ws->d_offs.Reserve(n + 1);
    if (e == hipSuccess) e = ws->d_id_offs.Reserve(n + 1);
    if (e == hipSuccess && n_ids) e = hipMemcpyAsync(ws->d_ids.p, ids + base, n_ids * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ws->d_offs.p, id_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    // the pieces outside the vocabulary (Decode(pieces)): their bytes and 32-bit offsets, for this call only
    struct LitGuard { Workspace *w; ~LitGuard() { w->lit_bytes = nullptr; w->lit_offs = nullptr; w->n_lit = 0; } } lit_guard{ws};
    if (e == hipSuccess && n_lit) {
      if (n_lit >= (1ull << 31) || lit_offsets[n_lit] >= (1ull << 32)) { free(ho); return Fail(h, kInvalidArgument, "too many pieces outside the vocabulary in one batch"); }
      std::vector<uint32_t> lo(n_lit + 1);
      for (uint64_t k = 0; k <= n_lit; ++k) lo[k] = static_cast<uint32_t>(lit_offsets[k]);
      for (uint64_t k = 0; k < n_lit; ++k)
        if (lo[k + 1] < lo[k] || lo[k + 1] - lo[k] > 0xFFFFu) { free(ho); return Fail(h, kInvalidArgument, "a piece outside the vocabulary is longer than 65535 bytes"); }
      e = ws->d_lit_offs.Reserve(n_lit + 1);
      if (e == hipSuccess) e = ws->d_lit_bytes.Reserve(lo[n_lit] + 16);
      if (e == hipSuccess) e = hipMemcpyAsync(ws->d_lit_offs.p, lo.data(), (n_lit + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st);
      if (e == hipSuccess && lo[n_lit]) e = hipMemcpyAsync(ws->d_lit_bytes.p, lit_bytes, lo[n_lit], hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);       // (lo is a local)
      ws->lit_bytes = ws->d_lit_bytes.p; ws->lit_offs = ws->d_lit_offs.p; ws->n_lit = static_cast<uint32_t>(n_lit);
    }
    if (e != hipSuccess) { free(ho); return FailHip(h, e, "staging the ids"); }
    const int32_t *d_ids = ws->d_ids.p - base;     // the kernels address ids + id_offsets[i]
    uint64_t cap = n_ids * 6 + 64, total = 0;
    int rc = kOk;
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (hipError_t e2 = ws->d_text.Reserve(cap); e2 != hipSuccess) { free(ho); return FailHip(h, e2, "hipMalloc(text)"); }
      rc = DecodeDevice(h, ws, d_ids, ws->d_offs.p, n, ws->d_text.p, ws->d_text.cap, ws->d_id_offs.p, st, &total);
      if (rc != kResourceExhausted || total <= ws->d_text.cap) break;
      cap = total;
    }
    if (rc != kOk) { free(ho); return rc; }
    char *ht = static_cast<char *>(malloc(total ? total : 1));
    if (!ht) { free(ho); return Fail(h, kResourceExhausted, "out of host memory"); }
    e = hipMemcpyAsync(ho, ws->d_id_offs.p, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && total) e = hipMemcpyAsync(ht, ws->d_text.p, total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(ho); free(ht); return FailHip(h, e, "hipMemcpy(text)"); }
    *text = ht;
    *text_offsets = ho;
    return kOk;
  });
}
}  // namespace

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
  if (n_lines) *n_lines = 0;
  if (text_bytes) *text_bytes = 0;
  if (!n_lines || !text_bytes) return Fail(h, kInternal, "output container is null");
  if (bytes && (!d_file || (reinterpret_cast<uintptr_t>(d_file) & 15u))) return Fail(h, kInvalidArgument, "d_file must be 16-byte aligned");
  return Guard(h, [&]() -> int {
    HIP_OR_RETURN(h, hipSetDevice(h->device));
    Lease L(h);
    if (int rc = L.Ready(); rc != kOk) return rc;
    Workspace *ws = L.ws.get();
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (bytes == 0) {
      if (d_offsets && offsets_capacity) HIP_OR_RETURN(h, hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), stream));
      HIP_OR_RETURN(h, hipStreamSynchronize(stream));
      return kOk;
    }
    const uint64_t chunks = (bytes + kSplitChunk - 1) / kSplitChunk;
    if (chunks >= (1ull << 32) - 64) return Fail(h, kInvalidArgument, "file image too large for one call");
    HIP_OR_RETURN(h, ws->d_counts.Reserve(chunks + 1));
    HIP_OR_RETURN(h, ws->d_chunk_base.Reserve(chunks + 1));
    HIP_OR_RETURN(h, ws->d_tile_sums.Reserve((chunks + kScanTile - 1) / kScanTile + 2));
    SplitArgs a{};
    a.file = static_cast<const uint8_t *>(d_file); a.bytes = bytes; a.counts = ws->d_counts.p;
    a.chunk_base = ws->d_chunk_base.p; a.text = static_cast<uint8_t *>(d_text); a.offsets = d_offsets;
    const uint64_t wide = static_cast<uint64_t>(h->n_cu) * 16;
    const int grid = static_cast<int>(chunks < wide ? chunks : wide);
    HIP_OR_RETURN(h, LaunchSplit(false, a, grid, stream));
    {
      ScanArgs sa{ws->d_counts.p, static_cast<uint32_t>(chunks), ws->d_tile_sums.p, ws->d_chunk_base.p};
      const uint32_t tiles = (static_cast<uint32_t>(chunks) + kScanTile - 1) / kScanTile;
      HIP_OR_RETURN(h, LaunchScan(sa, static_cast<int>(tiles < static_cast<uint32_t>(h->n_cu * 8) ? tiles : h->n_cu * 8), stream));
    }
    HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->total_ids, ws->d_chunk_base.p + chunks, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipMemcpyAsync(&ws->h_ctrl->pad, static_cast<const uint8_t *>(d_file) + bytes - 1, 1, hipMemcpyDeviceToHost, stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    const uint8_t last = *reinterpret_cast<const uint8_t *>(&ws->h_ctrl->pad);
    const uint64_t nl = ws->h_ctrl->total_ids;
    const uint64_t lines = nl + (last != 0x0A ? 1 : 0);      // std::getline: a last line without '\n' counts
    *n_lines = lines;
    *text_bytes = bytes - nl;
    if (!d_text || !d_offsets || text_capacity < bytes - nl || offsets_capacity < lines + 1)
      return Fail(h, kResourceExhausted, "text_capacity / offsets_capacity is too small");
    HIP_OR_RETURN(h, LaunchSplit(true, a, grid, stream));
    HIP_OR_RETURN(h, hipStreamSynchronize(stream));
    return kOk;
  });
}

/* ---- corpus file -> ids (the caller-side loop of spm_encode, src/spm_encode_main.cc:159-165) -------------------------
 * mmap the input, cut it into chunks that end at a line end, and run them through a pipeline of worker threads (each
 * with a workspace and a stream): pinned staging copy -> H2D -> spmx_split_lines (getline semantics, on the device)
 * -> encode -> D2H -> format.  A writer keeps the chunks in order.  format "id": one line of space-separated ids per
 * input line, as `spm_encode --output_format=id` writes; "bin": out_path gets the ids (int32, flat), out_path + ".idx"
 * the n + 1 uint64 offsets. */
int spmx_encode_file(spmx_handle *h, const char *in_path, const char *out_path, const char *format, uint64_t *n_sentences,
                     uint64_t *n_ids) {
  if (!h) return kInvalidArgument;
  if (n_sentences) *n_sentences = 0;
  if (n_ids) *n_ids = 0;
  return Guard(h, [&]() -> int {
    const bool bin = format && std::string(format) == "bin";
    if (format && !bin && std::string(format) != "id") return Fail(h, kInvalidArgument, "format must be \"id\" or \"bin\"");
    const int fd = open(in_path ? in_path : "", O_RDONLY);
    if (fd < 0) return Fail(h, kNotFound, std::string("\"") + (in_path ? in_path : "") + "\": No such file or directory");
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return Fail(h, kInternal, "fstat failed"); }
    const uint64_t size = static_cast<uint64_t>(sb.st_size);
    const uint8_t *file = nullptr;
    if (size) {
      void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); return Fail(h, kInternal, "mmap failed"); }
      file = static_cast<const uint8_t *>(m);
    }
    close(fd);
    FILE *out = fopen(out_path ? out_path : "", "wb");
    FILE *idx = nullptr;
    if (out && bin) idx = fopen((std::string(out_path) + ".idx").c_str(), "wb");
    if (!out || (bin && !idx)) {
      if (out) fclose(out);
      if (file) munmap(const_cast<uint8_t *>(file), size);
      return Fail(h, kPermissionDenied, std::string("cannot write \"") + (out_path ? out_path : "") + "\"");
    }
    // chunks of about 64 MiB that end after a line end
    std::vector<uint64_t> cut(1, 0);
    const uint64_t target = 64ull << 20;
    while (cut.back() < size) {
      uint64_t e = cut.back() + target;
      if (e >= size) e = size;
      else {
        const void *nl = memchr(file + e, '\n', size - e);
        e = nl ? static_cast<uint64_t>(static_cast<const uint8_t *>(nl) - file) + 1 : size;
      }
      cut.push_back(e);
    }
    const uint64_t n_chunks = cut.size() - 1;
    int T = h->host_threads < 4 ? h->host_threads : 4;
    if (static_cast<uint64_t>(T) > n_chunks) T = static_cast<int>(n_chunks ? n_chunks : 1);
    std::mutex mu;
    std::condition_variable cv;
    uint64_t next_write = 0, sent_total = 0, id_total = 0;
    int first_rc = kOk;
    std::string first_err;
    auto worker = [&](int w) {
      auto body = [&]() -> int {
        HIP_OR_RETURN(h, hipSetDevice(h->device));
        Lease L(h);
        if (int r = L.Ready(); r != kOk) return r;
        Workspace *ws = L.ws.get();
        hipStream_t st = ws->stream;
        DevBuf<uint8_t> d_file;
        struct Guard2 { DevBuf<uint8_t> &b; ~Guard2() { b.Free(); } } g2{d_file};
        std::vector<int32_t> ids;
        std::vector<uint64_t> io;
        std::string formatted;
        for (uint64_t k = static_cast<uint64_t>(w); k < n_chunks; k += static_cast<uint64_t>(T)) {
          const uint64_t bytes = cut[k + 1] - cut[k];
          const uint8_t *src = file + cut[k];
          uint64_t nl = 0;
          for (const uint8_t *p = src, *e = src + bytes; p < e;) {
            const void *q = memchr(p, '\n', static_cast<size_t>(e - p));
            if (!q) break;
            ++nl;
            p = static_cast<const uint8_t *>(q) + 1;
          }
          const uint64_t lines = nl + (bytes && src[bytes - 1] != '\n' ? 1 : 0);
          HIP_OR_RETURN(h, ws->h_text.Reserve(bytes + 32));
          HIP_OR_RETURN(h, d_file.Reserve(bytes + 32));
          HIP_OR_RETURN(h, ws->d_text.Reserve(bytes + 32));
          HIP_OR_RETURN(h, ws->d_offs.Reserve(lines + 2));
          HIP_OR_RETURN(h, ws->d_id_offs.Reserve(lines + 2));
          memcpy(ws->h_text.p, src, bytes);
          HIP_OR_RETURN(h, hipMemcpyAsync(d_file.p, ws->h_text.p, bytes, hipMemcpyHostToDevice, st));
          uint64_t n_lines = 0, text_bytes = 0;
          int r = spmx_split_lines_device(h, d_file.p, bytes, ws->d_text.p, ws->d_text.cap, ws->d_offs.p, ws->d_offs.cap, st, &n_lines, &text_bytes);
          if (r != kOk) return r;
          uint64_t want = text_bytes / 2 + 4 * n_lines + 64, total = 0;
          for (int attempt = 0; attempt < 2; ++attempt) {
            HIP_OR_RETURN(h, ws->d_ids.Reserve(want));
            r = EncodeDevice(h, ws, ws->d_text.p, text_bytes, ws->d_offs.p, n_lines, ws->d_ids.p, ws->d_ids.cap, ws->d_id_offs.p,
                             nullptr, st, &total, nullptr);
            if (r != kResourceExhausted || total <= ws->d_ids.cap) break;
            want = total;
          }
          if (r != kOk) return r;
          ids.resize(total);
          io.resize(n_lines + 1);
          if (total) HIP_OR_RETURN(h, hipMemcpyAsync(ids.data(), ws->d_ids.p, total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
          HIP_OR_RETURN(h, hipMemcpyAsync(io.data(), ws->d_id_offs.p, (n_lines + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
          HIP_OR_RETURN(h, hipStreamSynchronize(st));
          if (!bin) {                                  // absl::StrJoin(ids, " ") per line (spm_encode_main.cc:116-119)
            formatted.clear();
            formatted.reserve(total * 6 + n_lines);
            char tmp[16];
            for (uint64_t s2 = 0; s2 < n_lines; ++s2) {
              for (uint64_t x = io[s2]; x < io[s2 + 1]; ++x) {
                if (x > io[s2]) formatted.push_back(' ');
                int v = ids[x], len = 0;
                do { tmp[len++] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
                while (len) formatted.push_back(tmp[--len]);
              }
              formatted.push_back('\n');
            }
          }
          std::unique_lock<std::mutex> l(mu);
          cv.wait(l, [&] { return next_write == k || first_rc != kOk; });
          if (first_rc != kOk) return kOk;
          bool ok = true;
          if (bin) {
            if (total) ok = fwrite(ids.data(), sizeof(int32_t), total, out) == total;
            for (uint64_t s2 = 0; s2 < n_lines; ++s2) io[s2] += id_total;
            if (n_lines) ok = ok && fwrite(io.data(), sizeof(uint64_t), n_lines, idx) == n_lines;
          } else if (!formatted.empty()) {
            ok = fwrite(formatted.data(), 1, formatted.size(), out) == formatted.size();
          }
          sent_total += n_lines;
          id_total += total;
          next_write = k + 1;
          cv.notify_all();
          if (!ok) return Fail(h, kDataLoss, "short write");
        }
        return kOk;
      };
      int rc = kOk;
      try { rc = body(); } catch (const std::bad_alloc &) { rc = kResourceExhausted; t_error = "out of host memory"; } catch (...) { rc = kInternal; t_error = "unknown exception"; }
      if (rc != kOk) {
        const std::string err = t_error;
        std::lock_guard<std::mutex> l(mu);
        if (first_rc == kOk) { first_rc = rc; first_err = err; }
        cv.notify_all();
      }
    };
    std::vector<std::thread> pool;
    for (int w = 1; w < T; ++w) pool.emplace_back(worker, w);
    worker(0);
    for (auto &t : pool) t.join();
    if (bin && first_rc == kOk) fwrite(&id_total, sizeof(uint64_t), 1, idx);      // the closing offset
    if (idx) fclose(idx);
    fclose(out);
    if (file) munmap(const_cast<uint8_t *>(file), size);
    if (first_rc != kOk) return Fail(h, first_rc, first_err);
    if (n_sentences) *n_sentences = sent_total;
    if (n_ids) *n_ids = id_total;
    return kOk;
  });
}

int spmx_set_profiling(spmx_handle *h, int enabled) {
  if (!h) return kInvalidArgument;
  std::lock_guard<std::mutex> l(h->mu);
  h->profiling = enabled != 0;
  return kOk;
}

int spmx_handle_info(const spmx_handle *h, uint64_t *table_bytes, double *load_ms) {
  if (!h) return kInvalidArgument;
  if (table_bytes) *table_bytes = h->table_bytes;
  if (load_ms) *load_ms = h->load_ms;
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
                      uint64_t *bytes, uint64_t *path, float *total_ms) {
  if (!h) return 0;
  const Profile &p = h->prof;
  for (int c = 0; c < p.n; ++c) {
    if (kernel_ms) kernel_ms[c] = p.kernel_ms[c];
    if (sentences) sentences[c] = p.sentences[c];
    if (raw_bytes) raw_bytes[c] = p.raw_bytes[c];
    if (ids) ids[c] = p.ids[c];
    // L + 8 + 4 T' + 8 per sentence
    if (bytes) bytes[c] = p.raw_bytes[c] + 16 * p.sentences[c] + 4 * p.ids[c];
  }
  if (path) for (int k = 0; k < 4; ++k) path[k] = p.path[k];
  if (total_ms) *total_ms = p.total_ms;
  return p.n;
}

}  // extern "C"
