// Device bodies of the encode path: one sentence per 64-lane wavefront, the
// whole sentence resident in LDS between the raw-byte load and the id store.
//
//   normalize_wave   Normalizer::Normalize          (src/normalizer.cc:71-253), position-parallel
//   bpe_wave         bpe::Model::SampleEncode(a=0)   (src/bpe_model.cc:38-203), sentence per wave
//   (the lane-per-sentence kernels -- unigram EncodeOptimized, word-wise BPE -- are in kernels_stream.h /
//    kernels_bpe_stream.h)
//   emit_wave        PopulateSentencePieceText + ApplyExtraOptions, ids only
//                    (src/sentencepiece_processor.cc:547-636, :1019-1064)
//
// All cross-lane traffic goes through wv:: (wave.h) and sits in wave-uniform
// control flow.  No MFMA: this is byte / index work bound by dependent table
// lookups, not by FLOPs.
#ifndef SPMX_KERNELS_H_
#define SPMX_KERNELS_H_

#ifndef SPMX_WAVE_API   // tests/emu/ supplies a lock-step CPU model of wv:: (test seam)
#include "wave.h"
#endif
#include <math.h>

#include "dev.h"

namespace spmx {

// call-level status bits written to EncodeArgs::status
enum : uint32_t {
  kStArenaOverflow = 1u << 0,   // the scratch id arena was too small: caller retries with a bigger one
  kStTooLong = 1u << 1,         // (Normalize / align kernels) a sentence does not fit the largest staged class
  kStInternal = 1u << 2,        // "all normalized characters are not consumed" and friends
};
// per-sentence status bytes (EncodeArgs::sent_status): util::StatusCode numbers (src/sentencepiece_processor.h:34-52).
// A sentence that fails yields no ids, as the reference's batch form does for a failing element (EncodeAsIds swallows
// the Status, python/src/sentencepiece/sentencepiece.i:249-265); the other sentences of the batch are unaffected.
enum : uint32_t { kSsOk = 0, kSsResourceExhausted = 8, kSsOutOfRange = 11, kSsInternal = 13 };

constexpr int kMaxClasses = 8;

// One length class of the streaming launch (host-planned after classify; kernels_stream.h)
struct StreamClass {
  uint32_t rcap;         // sentences of the class have at most rcap raw bytes
  uint32_t tcap;         // bytes a text column of its tiles holds (> rcap + 4)
  uint32_t lane_shift;   // log2 of the lanes (sentences) of a tile: 6, or less when a column is too big for 64 per slab
  uint32_t tw;           // sentences per main tile (<= 1 << lane_shift; fewer when the class is too short to give every wave a full tile)
  uint32_t main_tiles;   // ceil(count / tw)
  uint32_t tile_base;    // main tiles handed out before this class's (longest class first)
  uint32_t count;        // sentences in the class list
  uint32_t general;      // every lane of its main tiles runs norm_lane_any (models / classes the ASCII fast path cannot take)
  uint32_t min_lanes;    // an ASCII tile normalizes its non-ASCII sentences itself when at least this many lanes have one
  uint32_t split;        // the class's tiles take the SPLIT form (kernels_matchfold.h): match wave-cooperatively, fold per lane
  uint32_t ccap;         // split: entries a sentence's candidate stream holds (MatchStreamCap)
  uint32_t pad[1];
};
// Device-side state of the tile queue of ONE streaming launch, zeroed before it
// (a cache line of its own: every tile a launch takes is an atomic on this word, and what else lived on its line -- the
// word kernels' list counters did until round 6 -- waited behind them)
struct alignas(128) StreamQueue {
  uint32_t main_cursor;               // next main tile
  uint32_t pad[31];
};
// Lists that outlive a launch (one set per call): what the streaming launches could not take
struct SideLists {
  uint32_t over_count;                // overflow list: sentences that fit no text column of their launch
  uint32_t long_count;                // long list (BPE): sentences for the long form (kernels_long.h)
  unsigned long long over_max_raw;    // longest raw sentence on the overflow list
  unsigned long long long_raw;        // raw bytes on the long list
  unsigned long long n_failed;        // sentences with a non-zero status byte
  unsigned long long n_backlog;       // sentences that waited in a wave's backlog (kernels_stream.h)
};

struct EncodeArgs {
  SpmxDev dev;
  const uint8_t *text;          // packed sentences
  const uint64_t *offs;         // n + 1 byte offsets
  int32_t *arena;               // scratch id arena, filled in completion order
  unsigned long long *arena_head;
  uint64_t arena_cap;
  uint64_t *tmp_off;            // per sentence: where its ids sit in the arena
  uint32_t *counts;             // per sentence: number of ids
  uint8_t *sent_status;         // per sentence: kSs* (zeroed before the first launch of a call)
  uint32_t *status;             // call-level kSt* bits
  unsigned long long *stats;    // kStatsPerClass words: {sentences, raw bytes, ids, cycles load, normalize, segment, emit, search iterations}
  int32_t *arena_tb;            // spans form (kernels_align.h), else null: next to every body id in `arena`, the
                                // position in the normalized (device) text where its token begins
  uint32_t *long_list;          // BPE: sentences for the long form (kernels_long.h): a word too long for the lane form, a
                                // sentence too long for the sentence-per-wave form, too many UNUSED merges
  SideLists *side;
  // ---- streaming launch (kernels_stream.h) ----
  const uint32_t *lists;        // n_classes x n: the class lists of classify
  uint32_t *over_list;          // sentences that fit no text column of this launch (null in the overflow launch itself)
  StreamQueue *q;
  uint8_t *slab;                // HBM scratch: slab_bytes per wavefront of the launch
  uint64_t slab_bytes;
  uint32_t n;                   // sentences of the batch = stride of lists
  uint32_t n_classes;
  uint32_t total_main;          // sum of main_tiles
  uint32_t bp_short;      // back-pointer entries are uint16 (kernels_stream.h BpShort)
  uint32_t ring;                // score ring entries (power of two > longest piece)
  uint32_t fast_ok;             // the model meets fast_norm_stream's preconditions
  uint32_t no_lane_general;     // A/B switch: ASCII tiles put every non-ASCII sentence into the backlog
  uint32_t private_bytes;       // LDS of one wavefront of the streaming launch (StreamPrivateBytes, or more for the split form's image)
  uint32_t split_ring;          // split form: entries of the fold's character-indexed rings (8 or 16, kernels_matchfold.h FoldRing)
  uint32_t match_rows;          // split form (kernels_matchfold.h): candidate-row entries = the trie's deepest chain of prefixes
  uint32_t no_char_norm;        // A/B switch: 1: no tile takes the character-stepping normalizer (kernels_normlane.h char_norm_stream); 2 (test seam): every tile does
  StreamClass cls[kMaxClasses];
  // ---- word kernel (kernels_word.h): what it cannot take, per length class, for the general launch that follows ----
  uint32_t *left_lists;         // n_classes x n
  uint32_t *left_counts;        // n_classes (zeroed before the launch)
  uint32_t *left2_lists;        // (collecting pass) sentences no later word pass can take: for the general launches
  uint32_t *left2_counts;
  // the call-local word memo (kernels_word.h): words the load-time memo lacks, segmented once per call
  unsigned long long *dyn_tag;  // [dyn_mask + 1] 64-bit hash of the word, 0: free
  U4 *dyn_ent;                  // [dyn_mask + 1][4] {key} {state, n_ids | flags, bound, bmax} {ids 0-3} {ids 4-7} (kDynWide: 16 x 16 bits)
  uint32_t *dyn_list;           // slots taken, in order of arrival
  uint32_t *dyn_count;
  uint32_t dyn_mask;
  uint32_t dyn_cap;             // entries dyn_list holds
  U4 *resume;                   // per sentence the first round keeps for the second: {position, ids written, bound, 0}
  uint32_t ids16;               // the word kernels write 16-bit ids into their arena slots (vocabularies of up to 65536 pieces)
  // word-per-lane rounds WITHOUT classify (kernels_wordwave.h): tiles are runs of 64 sentences in input order (lists ==
  // null: sentence first + lane) or of one list (row 0 of lists); a sentence's length class -- which only says whether it
  // is a document (cls[].general) and which list it goes to when the word form hands it on -- is found from cls[].rcap
  uint32_t direct;
  // ---- sentence-per-wave launch (BPE models that are not word-wise; kernels_bpe.h) ----
  const uint32_t *list;         // sentence indices of this length class
  const uint32_t *list_count;   // number of entries in list (device resident)
  uint32_t *next_list;          // escalation: sentences whose normalized form overflowed ncap (null: they go to long_list)
  uint32_t *next_count;
  uint32_t rcap, ncap;          // LDS capacities of this class: raw bytes, normalized bytes
};

constexpr int kStatsPerClass = 8;
constexpr uint32_t kTokEnd = 0x8000u;  // blen[] flag: a token of the best path ends here

// Darts::DoubleArrayUnit::offset() (third_party/darts_clone/darts.h:72-74)
SPMX_DEVICE uint32_t DartsOffset(uint32_t u) { return (u >> 10) << ((u & (1u << 9)) >> 6); }

SPMX_DEVICE int OneCharLenDev(uint32_t c) {  // src/util.h:151-153
  const uint32_t h = c >> 4;
  return h < 12 ? 1 : (h < 14 ? 2 : (h == 14 ? 3 : 4));
}
// The byte that stands for a whole character U+2581 in the normalized text, or a value no byte equals.
SPMX_DEVICE uint32_t SpByteOf(const SpmxDev &d) { return (d.flags & kNfCompressSp) ? kSpByte : 0x100u; }

// ---------------------------------------------------------------- helpers --
// Inclusive -> exclusive prefix sum over the 64 lanes; *total gets the wave sum.
SPMX_DEVICE int wave_excl_scan(int v, int lane, int *total) {
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = wv::shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  *total = wv::shfl(incl, 63);
  return incl - v;
}

// Resolves which positions of a 64-position window are reached by the chain
//   start -> start + step[start] -> ...
// given the (uniform) first start inside or beyond the window.  step is each
// lane's own hop length (>= 1).  Returns the bitmask of reached positions and
// advances *next_start (absolute) past the window.  Hops of 1 are flooded with
// one 64-bit add (carry propagation); 2/3/4 with shifts; longer hops one by one.
SPMX_DEVICE uint64_t resolve_chain(int base, int step, bool valid, int *next_start) {
  const uint64_t m1 = wv::ballot(valid && step == 1);
  const uint64_t m2 = wv::ballot(valid && step == 2);
  const uint64_t m3 = wv::ballot(valid && step == 3);
  const uint64_t m4 = wv::ballot(valid && step == 4);
  const uint64_t mo = wv::ballot(valid && step > 4);
  const uint64_t mv = wv::ballot(valid);
  uint64_t S = 0;
  const int rel = *next_start - base;
  if (rel < 64) S = 1ull << rel;
  S &= mv;
  uint64_t done_long = 0;
  for (;;) {
    const uint64_t S0 = S;
    const uint64_t x = S & m1;
    S |= ((x + m1) ^ m1);
    S |= (S & m2) << 2;
    S |= (S & m3) << 3;
    S |= (S & m4) << 4;
    uint64_t lg = S & mo & ~done_long;
    done_long |= lg;
    while (lg) {  // uniform: S and mo are wave-uniform
      const int i = wv::ffs64(lg) - 1;
      lg &= lg - 1;
      const int st = wv::shfl(step, i);
      if (i + st < 64) S |= 1ull << (i + st);
    }
    S &= mv;
    if (S == S0) break;
  }
  if (S) {
    const int hi = 63 - wv::clz64(S);
    *next_start = base + hi + wv::shfl(step, hi);
  }
  return S;
}

}  // namespace spmx
#include "kernels_normlane.h"
namespace spmx {

// ------------------------------------------------------------- normalizer --
// Returns the normalized length, or -1 if it does not fit in ncap bytes.
// raw[0, L) and norm[] live in LDS.  Restates Normalize()'s per-prefix loop as
// 64-position sweeps: (1) every position speculatively computes its own
// NormalizePrefix result (longest user-defined symbol, else longest charsmap
// rule, else one UTF-8 char or U+FFFD for a malformed byte); (2) the chain of
// real prefix starts is resolved on wave-uniform bitmasks; (3) the whitespace
// state machine (is_prev_space) becomes a "last writer" lookup on ballot masks;
// (4) a prefix sum places every prefix's output.
// `orig` (optional, LDS, ncap entries): norm_to_orig (:105-120, :142-152) -- for every normalized byte the raw offset
// at which the prefix that produced it starts; *orig_end receives the closing entry (:181), -1 where the reference
// returns before pushing one (all-whitespace input, :96-99).
// HBM: raw / norm are global memory (the wave-cooperative unigram form, kernels_uniwave.h, normalizes documents in
// place): the fences between the sweeps and the read-back of the text's tail then cover global stores too.
template <bool HBM = false>
SPMX_DEVICE int normalize_wave(const SpmxDev &d, const uint8_t *raw, int L, uint8_t *norm, int ncap, int lane,
                               uint16_t *orig = nullptr, int *orig_end = nullptr) {
  const uint32_t F = d.flags;
  const bool rm = (F & kNfRemoveExtraWs) != 0;
  const bool one = (F & kNfCompressSp) != 0;             // U+2581 is the single byte kSpByte (dev.h)
  const bool esc = (F & kNfEscapeWs) != 0 && !one;       // escaped spaces take three bytes
  const int spw = esc ? 3 : 1;
  const uint32_t sp1 = one ? kSpByte : 0x20u;            // the one-byte space symbol when !esc
  int out = 0;
  if ((F & kNfAddDummyPrefix) && !(F & kNfWsSuffix)) {   // :128
    if (spw > ncap) return -1;
    if (lane < spw) norm[lane] = static_cast<uint8_t>(esc ? (lane == 0 ? 0xE2u : (lane == 1 ? 0x96u : 0x81u)) : sp1);
    out = spw;
  }
  bool P = rm;                       // is_prev_space (:130), wave-uniform between sweeps
  bool any_other = false;            // some prefix normalizes to something other than exactly " "
  int first_other = -1;              // raw offset of the first such prefix: `consumed` after the leading loop (:85-94)
  int next_start = 0;
  const bool has_map = (F & kNfHasCharsmap) != 0, has_uds = (F & kNfHasUserDefined) != 0;
  const uint32_t droot = has_map ? DartsOffset(d.ndarts[0]) : 0u;
  const uint32_t uroot = has_uds ? (d.utrie[0].x >> kDatBaseShiftDev) : 0u;
  const bool try_fast = !has_uds && (!has_map || d.nfilt != 0u);
  // What the common sweep (0) knows about a position from the text alone -- the character there, the one behind it, and the
  // three filter words -- is prepared ONE SWEEP AHEAD: the words come from global memory, and a sweep that asks for them and
  // waits pays an L2 round trip per 64 bytes (a third of the sweep, round 6).
  struct FastPre {
    uint32_t w0 = 0, nx = 0, cp = 0, cq = 0, nq = 0, ws = 0, wa = 0, w2 = 0;
    int mb = 1;
    bool is_cont = false, okc = false, after_ok = false, has_next = false, big = false, nbig = false;
  };
  auto prepare = [&](int pp) __attribute__((always_inline)) -> FastPre {
    FastPre f;
    if (pp >= L) return f;
    struct __attribute__((packed, aligned(1))) U32B { uint32_t v; };
    uint32_t w0 = 0, w1 = 0;
    const int rem = L - pp;
    if (!HBM || rem >= 8) { w0 = reinterpret_cast<const U32B *>(raw + pp)->v; w1 = reinterpret_cast<const U32B *>(raw + pp + 4)->v; }
    else for (int k = 0; k < rem; ++k) { if (k < 4) w0 |= static_cast<uint32_t>(raw[pp + k]) << (8 * k); else w1 |= static_cast<uint32_t>(raw[pp + k]) << (8 * (k - 4)); }
    if (rem < 8) {                                     // bytes past the text read as 0: they continue nothing
      if (rem <= 4) { w1 = 0; if (rem < 4) w0 &= (1u << (8 * rem)) - 1u; }
      else w1 &= (1u << (8 * (rem - 4))) - 1u;
    }
    const uint32_t b0 = w0 & 0xFFu, b1 = (w0 >> 8) & 0xFFu, b2 = (w0 >> 16) & 0xFFu, b3 = w0 >> 24;
    f.w0 = w0;
    f.is_cont = (b0 & 0xC0u) == 0x80u;
    const int mb = b0 < 0x80u ? 1 : (b0 < 0xE0u ? 2 : (b0 < 0xF0u ? 3 : 4));
    f.mb = mb;
    const bool c1 = (b1 & 0xC0u) == 0x80u, c2 = (b2 & 0xC0u) == 0x80u, c3 = (b3 & 0xC0u) == 0x80u;
    if (mb == 1) { f.okc = true; f.cp = b0; }
    else if (mb == 2) { f.okc = b0 >= 0xC2u && c1; f.cp = (b0 & 0x1Fu) << 6 | (b1 & 0x3Fu); }
    else if (mb == 3) { f.okc = c1 && c2 && (b0 != 0xE0u || b1 >= 0xA0u) && (b0 != 0xEDu || b1 < 0xA0u); f.cp = (b0 & 0x0Fu) << 12 | (b1 & 0x3Fu) << 6 | (b2 & 0x3Fu); }
    else { f.okc = b0 <= 0xF4u && c1 && c2 && c3 && (b0 != 0xF0u || b1 >= 0x90u) && (b0 != 0xF4u || b1 < 0x90u);
           f.cp = (b0 & 0x07u) << 18 | (b1 & 0x3Fu) << 12 | (b2 & 0x3Fu) << 6 | (b3 & 0x3Fu); }
    const uint64_t x8 = static_cast<uint64_t>(w1) << 32 | w0;
    const uint32_t nx = static_cast<uint32_t>(x8 >> (8 * mb));               // the four bytes behind the character
    f.nx = nx;
    const uint32_t n0 = nx & 0xFFu;
    f.after_ok = (n0 & 0xC0u) != 0x80u;
    if (has_map && !f.is_cont && f.okc) {
      // (the three filter words are asked for together, whatever the first one says: one trip to the cache, not three)
      const uint32_t n1 = (nx >> 8) & 0xFFu, n2 = (nx >> 16) & 0xFFu, n3 = nx >> 24;
      uint32_t ncp;
      bool nok = true;
      if (n0 < 0x80u) ncp = n0;
      else if (n0 < 0xE0u) { ncp = (n0 & 0x1Fu) << 6 | (n1 & 0x3Fu); nok = n0 >= 0xC2u && (n1 & 0xC0u) == 0x80u; }
      else if (n0 < 0xF0u) { ncp = (n0 & 0x0Fu) << 12 | (n1 & 0x3Fu) << 6 | (n2 & 0x3Fu); nok = (n1 & 0xC0u) == 0x80u && (n2 & 0xC0u) == 0x80u; }
      else { ncp = (n0 & 0x07u) << 18 | (n1 & 0x3Fu) << 12 | (n2 & 0x3Fu) << 6 | (n3 & 0x3Fu); nok = (n1 & 0xC0u) == 0x80u && (n2 & 0xC0u) == 0x80u && (n3 & 0xC0u) == 0x80u; }
      f.has_next = mb < rem;                                                   // the character behind it, if any
      f.big = f.cp >= kNfiltCps;
      f.nbig = f.has_next && (!nok || ncp >= kNfiltCps);
      const uint32_t *ft = d.npair + 2048;
      f.cq = f.big ? 0u : f.cp;
      f.nq = (f.has_next && !f.nbig) ? ncp : 0u;
      f.ws = ft[f.cq >> 5]; f.wa = ft[kNfiltWords + (f.cq >> 5)]; f.w2 = ft[2u * kNfiltWords + (f.nq >> 5)];
    }
    return f;
  };
  FastPre cur_pre, nxt_pre;
  if (try_fast) cur_pre = prepare(lane);
  for (int b = 0; b < L; b += 64, cur_pre = nxt_pre) {
    const int p = b + lane;
    const bool valid = p < L;
    // (0) THE COMMON SWEEP: every position from the chain's next start on is a well-formed character (DecodeUTF8's rule,
    // util.cc:51-84) at which no charsmap key can match (the code-point filters, dev.h nfilt) and that is no literal
    // U+2581 -- every prefix is its own character (:231-244), so the chain of starts is the bytes that continue no
    // character, and what remains of the loop is the whitespace state machine and a prefix sum.  One test per sweep; any
    // other sweep takes the general steps below.  (The eight bytes a lane looks at may lie past the text: an LDS image has
    // the slack; in HBM the last lanes of a text gather theirs byte by byte.)
    if (try_fast) {
      nxt_pre = b + 64 < L ? prepare(b + 64 + lane) : FastPre();
      const FastPre &f = cur_pre;
      const uint32_t b0 = f.w0 & 0xFFu, b1 = (f.w0 >> 8) & 0xFFu, b2 = (f.w0 >> 16) & 0xFFu, b3 = f.w0 >> 24;
      const int mb = f.mb;
      const uint32_t cp = f.cp;
      const bool is_cont = f.is_cont;
      const bool from = valid && p >= next_start;
      bool cx = from && (is_cont ? p == next_start : !(f.okc && f.after_ok));
      const bool st = from && !is_cont;
      uint32_t rule = 0;                                                       // a one-character rule: nblob offset | length << 24
      if (st && !cx) {
        if (one && cp == 0x2581u) cx = true;                                   // a literal U+2581 (kind 3 below)
        if (has_map) {
          if (f.big) cx = true;
          else if ((f.ws >> (f.cq & 31u)) & 1u) {
            const bool alone = (f.wa >> (f.cq & 31u)) & 1u;
            const bool second = f.nbig || (f.has_next && ((f.w2 >> (f.nq & 31u)) & 1u));
            if (second) cx = true;
            else if (alone) { rule = d.npair[2048u + 3u * kNfiltWords + cp]; if (rule == 0u) cx = true; }
          }
        }
      }
      if (!wv::any(cx)) {
        const bool is_sp = st && b0 == 0x20u;
        {
          const uint64_t om = wv::ballot(st && !is_sp);
          if (om && first_other < 0) first_other = b + wv::ffs64(om) - 1;
          any_other = any_other || om != 0;
        }
        bool Pl = false;
        if (rm) {                                          // (3) below with clsA = the spaces, clsN = the other characters
          const uint64_t mA = wv::ballot(is_sp), setters = wv::ballot(st);
          const uint64_t below = setters & ((1ull << lane) - 1ull);
          Pl = below ? (((mA >> (63 - wv::clz64(below))) & 1ull) != 0) : P;
          if (setters) P = ((mA >> (63 - wv::clz64(setters))) & 1ull) != 0;
        }
        const bool drop = is_sp && Pl;                     // :137-138
        const uint32_t out_len = st && !drop ? (rule ? rule >> 24 : static_cast<uint32_t>(is_sp ? spw : mb)) : 0u;
        const uint32_t incl = wv::scan_add(out_len);
        const int total = static_cast<int>(wv::read_lane(incl, 63));
        if (out + total > ncap) return -1;
        if (out_len) {
          int w = out + static_cast<int>(incl - out_len);
          if (orig) for (uint32_t k = 0; k < out_len; ++k) orig[w + static_cast<int>(k)] = static_cast<uint16_t>(p);
          if (rule) {                                      // :245-250 the key's replacement (no space in it)
            const uint8_t *rs = d.nblob + (rule & 0x00FFFFFFu);
            for (uint32_t k = 0; k < out_len; ++k) norm[w + static_cast<int>(k)] = rs[k];
          } else if (is_sp) {
            if (esc) { norm[w] = 0xE2; norm[w + 1] = 0x96; norm[w + 2] = 0x81; }
            else norm[w] = static_cast<uint8_t>(sp1);
          } else {
            norm[w] = static_cast<uint8_t>(b0);
            if (mb > 1) norm[w + 1] = static_cast<uint8_t>(b1);
            if (mb > 2) norm[w + 2] = static_cast<uint8_t>(b2);
            if (mb > 3) norm[w + 3] = static_cast<uint8_t>(b3);
          }
        }
        out += total;
        const uint64_t S = wv::ballot(st);
        if (S) {
          const int hi = 63 - wv::clz64(S);
          next_start = b + hi + wv::shfl(mb, hi);
        }
        continue;
      }
    }
    // (1) NormalizePrefix at every position (:195-253)
    int uds_len = 0, rule_len = 0;
    uint32_t rule_off = 0;
    if (has_uds) {                         // matcher_->PrefixMatch (:201-205, :324-346): longest user-defined symbol
      bool alive = valid;
      uint32_t nb = uroot;
      int depth = 0;
      while (wv::any(alive)) {
        if (alive) {
          const int q = p + depth;
          if (q < L) {
            const uint32_t c = raw[q];
            const U2 u = d.utrie[nb ^ c];
            if ((u.x & 0x1FFu) == (0x100u | c)) {
              ++depth;
              nb = u.x >> kDatBaseShiftDev;
              if (u.x & kDatTerminalDev) uds_len = depth;
            } else {
              alive = false;
            }
          } else {
            alive = false;
          }
        }
      }
    }
    if (has_map) {                         // trie_->commonPrefixSearch (:218-228; darts.h:467-513), longest key
      bool alive = valid;
      if (valid) {                         // no key starts with these two bytes (tables.cc npair): nothing to walk
        const uint32_t b0 = raw[p];
        const uint32_t b1 = p + 1 < L ? raw[p + 1] : 0u;
        if (!((d.npair[(b0 << 8 | b1) >> 5] >> (b1 & 31u)) & 1u)) alive = false;
      }
      uint32_t pos = droot;
      int depth = 0;
      while (wv::any(alive)) {
        if (alive) {
          const int q = p + depth;
          alive = false;
          if (q < L) {
            const uint32_t c = raw[q];
            pos ^= c;
            if (pos < d.ndarts_n) {
              const uint32_t u = d.ndarts[pos];
              if ((u & 0x800000FFu) == c) {          // unit.label() == c
                pos ^= DartsOffset(u);
                ++depth;
                alive = true;
                if ((u >> 8) & 1u) {                 // has_leaf: value sits in the unit at pos
                  if (pos < d.ndarts_n) { rule_len = depth; rule_off = d.ndarts[pos] & 0x7FFFFFFFu; }
                  else alive = false;
                }
              }
            }
          }
        }
      }
    }
    // kind: 0 copy raw span, 1 rule string, 2 U+FFFD, 3 a literal U+2581 written as kSpByte
    int kind = 0, consumed = 1;
    int len = 1, lead = 0, nsp = 0;
    bool ends_sp = false;
    uint32_t src = 0;
    if (valid) {
      if (uds_len > 0) {                       // :201-205 user-defined symbols pass through
        consumed = len = uds_len;
      } else if (rule_len > 0) {               // :222-228 longest rule, :245-250
        kind = 1;
        consumed = rule_len;
        src = rule_off;
      } else {                                 // :231-244 one UTF-8 char (DecodeUTF8, util.cc:51-84)
        const uint32_t b0 = raw[p];
        const int rem = L - p;
        int mb = 1;
        bool ok = b0 < 0x80;
        if (!ok) {
          const uint32_t b1 = rem >= 2 ? raw[p + 1] : 0u, b2 = rem >= 3 ? raw[p + 2] : 0u, b3 = rem >= 4 ? raw[p + 3] : 0u;
          const bool t1 = (b1 & 0xC0u) == 0x80u, t2 = (b2 & 0xC0u) == 0x80u, t3 = (b3 & 0xC0u) == 0x80u;
          if (rem >= 2 && (b0 & 0xE0u) == 0xC0u) {
            const uint32_t cp = (b0 & 0x1Fu) << 6 | (b1 & 0x3Fu);
            if (t1 && cp >= 0x80u) { ok = true; mb = 2; }
          } else if (rem >= 3 && (b0 & 0xF0u) == 0xE0u) {
            const uint32_t cp = (b0 & 0x0Fu) << 12 | (b1 & 0x3Fu) << 6 | (b2 & 0x3Fu);
            if (t1 && t2 && cp >= 0x800u && (cp < 0xD800u || cp >= 0xE000u)) { ok = true; mb = 3; }
          } else if (rem >= 4 && (b0 & 0xF8u) == 0xF0u) {
            const uint32_t cp = (b0 & 0x07u) << 18 | (b1 & 0x3Fu) << 12 | (b2 & 0x3Fu) << 6 | (b3 & 0x3Fu);
            if (t1 && t2 && t3 && cp >= 0x10000u && cp <= 0x10FFFFu) { ok = true; mb = 4; }
          }
        }
        if (ok) {
          consumed = len = mb;
          if (b0 == 0x20) { lead = nsp = 1; ends_sp = true; }
          if (one && mb == 3 && b0 == 0xE2u && raw[p + 1] == 0x96u && raw[p + 2] == 0x81u) { kind = 3; len = 1; }
        } else {
          kind = 2; consumed = 1; len = 3;     // U+FFFD, one byte consumed
        }
      }
    }
    // (2) which positions are real prefix starts
    const uint64_t S = resolve_chain(b, consumed, valid, &next_start);
    const bool is_start = ((S >> lane) & 1ull) != 0;
    if (is_start && uds_len > 0) {             // the matched span itself, spaces and all
      for (int k = 0; k < len; ++k) nsp += raw[p + k] == 0x20;
      while (lead < len && raw[p + lead] == 0x20) ++lead;
      ends_sp = raw[p + len - 1] == 0x20;
    } else if (is_start && kind == 1) {        // replacement = C string at normalized_[value] (:249)
      len = 0;
      bool leading = true;
      while (src + static_cast<uint32_t>(len) < d.nblob_n) {
        const uint32_t ch = d.nblob[src + len];
        if (ch == 0) break;
        ++len;
        if (ch == 0x20u) { ++nsp; if (leading) ++lead; } else { leading = false; }
        ends_sp = ch == 0x20u;
      }
    }
    // (3) whitespace state machine (:131-163)
    const bool single_sp = len == 1 && lead == 1;                 // p.first == " "
    {
      const uint64_t om = wv::ballot(is_start && !single_sp);
      if (om && first_other < 0) first_other = b + wv::ffs64(om) - 1;
      any_other = any_other || om != 0;
    }
    const bool clsA = is_start && len > 0 && lead == len;          // all spaces: leaves is_prev_space true
    const bool clsN = is_start && lead < len;                      // has a non-space byte
    bool Pl = false;
    if (rm) {
      const uint64_t mA = wv::ballot(clsA), mN = wv::ballot(clsN), mNe = wv::ballot(clsN && ends_sp);
      const uint64_t setters = mA | mN, val = mA | mNe;
      const uint64_t below = setters & ((1ull << lane) - 1ull);
      Pl = below ? (((val >> (63 - wv::clz64(below))) & 1ull) != 0) : P;
      if (setters) P = ((val >> (63 - wv::clz64(setters))) & 1ull) != 0;
    }
    const int strip = Pl ? lead : 0;                               // :137-138
    const int eff_len = is_start ? len - strip : 0;
    const int eff_sp = is_start ? nsp - strip : 0;
    const int out_len = eff_len + (esc ? 2 * eff_sp : 0);
    // (4) placement
    int total = 0;
    int w = out + wave_excl_scan(out_len, lane, &total);
    if (out + total > ncap) return -1;
    int k = strip;
    const int kend = is_start ? len : 0;
    while (wv::any(k < kend)) {
      if (k < kend) {
        uint32_t ch;
        if (kind == 0) ch = raw[p + k];
        else if (kind == 1) ch = d.nblob[src + k];
        else if (kind == 2) ch = k == 0 ? 0xEFu : (k == 1 ? 0xBFu : 0xBDu);
        else ch = kSpByte;
        if (one && ch == 0x20u) ch = kSpByte;
        if (esc && ch == 0x20u) {                                  // :143-148
          norm[w] = 0xE2; norm[w + 1] = 0x96; norm[w + 2] = 0x81;
          if (orig) orig[w] = orig[w + 1] = orig[w + 2] = static_cast<uint16_t>(p);
          w += 3;
        } else {
          if (orig) orig[w] = static_cast<uint16_t>(p);
          norm[w++] = static_cast<uint8_t>(ch);
        }
        ++k;
      }
    }
    out += total;
  }
  if (HBM) wv::sync_global(); else wv::sync();
  if (rm && !any_other) {              // :86-100 every prefix was " ": empty result, no dummy prefix,
    if (orig_end) *orig_end = -1;      // and no closing entry in norm_to_orig either
    return 0;
  }
  int fin = L;                         // `consumed` when the closing entry is pushed (:181)
  if (orig && (F & kNfAddDummyPrefix) && !(F & kNfWsSuffix)) {
    // the dummy prefix maps to `consumed` after the leading-space loop (:85-94, :128)
    if (lane < spw) orig[lane] = static_cast<uint16_t>(rm && first_other > 0 ? first_other : 0);
    if (HBM) wv::sync_global(); else wv::sync();
  }
  if (rm) {                            // :166-176 trailing space symbols
    for (;;) {
      if (out < spw) break;
      const bool is_sp = esc ? (norm[out - 3] == 0xE2 && norm[out - 2] == 0x96 && norm[out - 1] == 0x81)
                             : (norm[out - 1] == sp1);
      if (!is_sp) break;
      out -= spw;
      if (orig) fin = orig[out];       // :172
    }
  }
  if ((F & kNfAddDummyPrefix) && (F & kNfWsSuffix)) {   // :179
    if (out + spw > ncap) return -1;
    if (lane < spw) {
      norm[out + lane] = static_cast<uint8_t>(esc ? (lane == 0 ? 0xE2u : (lane == 1 ? 0x96u : 0x81u)) : sp1);
      if (orig) orig[out + lane] = static_cast<uint16_t>(fin);
    }
    out += spw;
    if (HBM) wv::sync_global(); else wv::sync();
  }
  if (orig_end) *orig_end = fin;
  return out;
}

// ------------------------------------------------------------------- emit --
// ids of the marked tokens in forward order, with the unknown-run merge or the
// byte-fallback expansion (sentencepiece_processor.cc:581-613) and the net
// effect of the extra options (:1019-1064).
// (Args: EncodeArgs, or LongArgs of the wave-cooperative unigram form -- the fields used here have the same names)
template <typename Args>
SPMX_DEVICE int emit_wave(const Args &a, uint32_t sid, const uint8_t *norm, int nlen, const int32_t *bid,
                           const uint16_t *blen, int lane) {
  const SpmxDev &d = a.dev;
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  int total = 0;
  // (both sweeps ask for the NEXT 64 positions' entries before they work on these: in HBM -- the long forms -- an
  // iteration is otherwise one memory latency long)
  auto entry = [&](int e, uint32_t *l, int32_t *id) __attribute__((always_inline)) {
    *l = 0; *id = 0;
    if (e <= nlen) { *l = blen[e]; *id = bid[e]; }
  };
  uint32_t l_nx; int32_t id_nx;
  entry(lane + 1, &l_nx, &id_nx);
  for (int b = 0; b < nlen; b += 64) {
    const int e = b + lane + 1;
    int cnt = 0;
    const uint32_t l = l_nx;
    const int32_t id_e = id_nx;
    entry(e + 64, &l_nx, &id_nx);
    if (e <= nlen) {
      if (l & kTokEnd) {
        const int len = static_cast<int>(l & (kTokEnd - 1));
        if (id_e == d.unk_id) {
          if (bf) cnt = norm[e - len] == spb ? 3 : len;   // an unknown piece is one character; U+2581 has 3 bytes
          else cnt = (e - len > 0 && bid[e - len] == d.unk_id) ? 0 : 1;
        } else {
          cnt = 1;
        }
      }
    }
    int t = 0;
    wave_excl_scan(cnt, lane, &t);
    total += t;
  }
  const int n_out = d.n_prefix + total + d.n_suffix;
  unsigned long long off = 0;
  if (lane == 0) off = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(n_out));
  off = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(off >> 32), 0)) << 32) |
        wv::shfl(static_cast<uint32_t>(off), 0);
  if (lane == 0) {
    a.counts[sid] = static_cast<uint32_t>(n_out);
    a.tmp_off[sid] = off;
  }
  if (off + static_cast<unsigned long long>(n_out) > a.arena_cap) {
    if (lane == 0) wv::atomic_or(a.status, kStArenaOverflow);
    return n_out;
  }
  int32_t *dst = a.arena + off;
  int32_t *dtb = a.arena_tb ? a.arena_tb + off : nullptr;
  if (lane < d.n_prefix) dst[lane] = d.prefix_ids[lane];
  if (lane < d.n_suffix) dst[d.n_prefix + total + lane] = d.suffix_ids[lane];
  int done = 0;
  entry(lane + 1, &l_nx, &id_nx);
  for (int b = 0; b < nlen; b += 64) {
    const int e = b + lane + 1;
    int cnt = 0, len = 0;
    int32_t id = 0;
    const uint32_t l = l_nx;
    const int32_t id_e = id_nx;
    entry(e + 64, &l_nx, &id_nx);
    if (e <= nlen) {
      if (l & kTokEnd) {
        len = static_cast<int>(l & (kTokEnd - 1));
        id = id_e;
        if (id == d.unk_id) {
          if (bf) cnt = norm[e - len] == spb ? 3 : len;
          else cnt = (e - len > 0 && bid[e - len] == d.unk_id) ? 0 : 1;
        } else {
          cnt = 1;
        }
      }
    }
    int t = 0;
    const int pos = done + wave_excl_scan(cnt, lane, &t);
    done += t;
    if (cnt == 1 && !(bf && id == d.unk_id)) {
      dst[d.n_prefix + (reverse ? total - 1 - pos : pos)] = id;
      if (dtb) dtb[d.n_prefix + (reverse ? total - 1 - pos : pos)] = e - len;
    } else if (cnt > 0) {          // byte fallback: one BYTE id per byte of the unknown piece
      const bool sp = norm[e - len] == spb;
      for (int k = 0; k < cnt; ++k) {
        const int j = pos + k;
        const uint32_t byte = sp ? (k == 0 ? 0xE2u : (k == 1 ? 0x96u : 0x81u)) : norm[e - len + k];
        dst[d.n_prefix + (reverse ? total - 1 - j : j)] = d.byte_ids[byte];
        if (dtb) dtb[d.n_prefix + (reverse ? total - 1 - j : j)] = e - len;
      }
    }
  }
  return n_out;
}

// ------------------------------------------------------- one work item -----
struct WaveLds {
  uint8_t *raw, *norm;
  int32_t *bid;
  uint16_t *blen;
  unsigned char *extra;   // model-specific scratch (BPE symbol tables)
};

SPMX_DEVICE WaveLds carve_lds(unsigned char *base, uint32_t rcap, uint32_t ncap) {
  WaveLds w;
  const uint32_t r = (rcap + 16 + 15) & ~15u, n = (ncap + 16 + 15) & ~15u;
  const uint32_t n4 = ((ncap + 4) * 4 + 15) & ~15u, n2 = ((ncap + 8) * 2 + 15) & ~15u;
  w.raw = base;
  w.norm = base + r;
  w.bid = reinterpret_cast<int32_t *>(base + r + n);
  w.blen = reinterpret_cast<uint16_t *>(base + r + n + n4);
  w.extra = base + r + n + n4 + n2;
  return w;
}

SPMX_DEVICE void fail_sentence(const EncodeArgs &a, uint32_t sid, uint32_t code, int lane) {
  if (lane == 0) {
    a.counts[sid] = 0;
    a.tmp_off[sid] = 0;
    a.sent_status[sid] = static_cast<uint8_t>(code);
    wv::atomic_add(&a.side->n_failed, 1ull);
  }
}
// the sentence leaves the sentence-per-wave kernel for the long form (kernels_long.h)
SPMX_DEVICE void to_long_list(const EncodeArgs &a, uint32_t sid, uint64_t raw_len, int lane) {
  if (lane == 0) {
    a.long_list[wv::atomic_add(&a.side->long_count, 1u)] = sid;
    wv::atomic_add(&a.side->long_raw, static_cast<unsigned long long>(raw_len));
    a.counts[sid] = 0u;                          // (until the long form has had it)
  }
}

}  // namespace spmx

#include "kernels_bpe.h"

namespace spmx {

// Encodes sentence `sid` with this wave. MODEL: 2 (BPE).
// Returns the number of ids written, or -1 if the sentence was handed on / failed.
template <int MODEL>
SPMX_DEVICE int encode_sentence(const EncodeArgs &a, uint32_t sid, const WaveLds &w, int lane, int *raw_len,
                                unsigned long long *cyc) {
  const unsigned long long c0 = wv::clock();
  const uint64_t beg = a.offs[sid];
  const uint64_t L64 = a.offs[sid + 1] - beg;
  if (L64 > a.rcap) {   // only reachable for the last staged class
    to_long_list(a, sid, L64, lane);
    return -1;
  }
  const int L = static_cast<int>(L64);
  *raw_len = L;
  const uint8_t *src = a.text + beg;
  for (int p = lane; p < L; p += 64) w.raw[p] = src[p];
  wv::sync();
  const unsigned long long c1 = wv::clock();
  int nlen = 0;
  if (L > 0) nlen = normalize_wave(a.dev, w.raw, L, w.norm, static_cast<int>(a.ncap), lane);
  const unsigned long long c2 = wv::clock();
  if (nlen < 0) {
    if (a.next_list) {
      if (lane == 0) a.next_list[wv::atomic_add(a.next_count, 1u)] = sid;
    } else {
      to_long_list(a, sid, L64, lane);
    }
    return -1;
  }
  int ok = 1;
  if (nlen > 0) {
    for (int e = lane; e <= nlen; e += 64) w.blen[e] = 0;
    wv::sync();
    static_assert(MODEL == 2, "the sentence-per-wave form serves BPE only");
    ok = bpe_wave(a, w.norm, nlen, w.bid, w.blen, carve_bpe(w.extra, a.ncap), lane);
  }
  if (ok == 0) {                   // a control piece among the symbols: "all normalized characters are not consumed."
    fail_sentence(a, sid, kSsInternal, lane);
    return -1;
  }
  if (ok < 0) {                    // more distinct UNUSED merges than the LDS table holds: the long form has no such bound
    to_long_list(a, sid, L64, lane);
    return -1;
  }
  const unsigned long long c3 = wv::clock();
  const int n_out = emit_wave(a, sid, w.norm, nlen, w.bid, w.blen, lane);
  const unsigned long long c4 = wv::clock();
  cyc[0] += c1 - c0; cyc[1] += c2 - c1; cyc[2] += c3 - c2; cyc[3] += c4 - c3;
  return n_out;
}

// Persistent block body: 64-thread workgroups, grid-stride over the class list.
template <int MODEL>
SPMX_DEVICE void encode_block(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const WaveLds w = carve_lds(smem, a.rcap, a.ncap);
  const uint32_t count = *a.list_count;
  unsigned long long n_sent = 0, n_raw = 0, n_ids = 0;
  unsigned long long cyc[4] = {0, 0, 0, 0};
  for (uint32_t item = static_cast<uint32_t>(wv::block_id()); item < count; item += static_cast<uint32_t>(wv::grid_size())) {
    int raw_len = 0;
    const int n_out = encode_sentence<MODEL>(a, a.list[item], w, lane, &raw_len, cyc);
    if (n_out >= 0) { ++n_sent; n_raw += static_cast<unsigned long long>(raw_len); n_ids += static_cast<unsigned long long>(n_out); }
  }
  if (a.stats && lane == 0 && n_sent) {
    wv::atomic_add(&a.stats[0], n_sent);
    wv::atomic_add(&a.stats[1], n_raw);
    wv::atomic_add(&a.stats[2], n_ids);
    for (int k = 0; k < 4; ++k) wv::atomic_add(&a.stats[3 + k], cyc[k]);
  }
}

// LDS bytes one wave needs for a (rcap, ncap) class.  Host and device agree on it.
inline uint32_t EncodeLdsBytes(int model_type, uint32_t rcap, uint32_t ncap) {
  const uint32_t r = (rcap + 16 + 15) & ~15u, n = (ncap + 16 + 15) & ~15u;
  const uint32_t n4 = ((ncap + 4) * 4 + 15) & ~15u, n2 = ((ncap + 8) * 2 + 15) & ~15u;
  uint32_t total = r + n + n4 + n2;
  if (model_type == 2) total += 3 * n4 + 2 * n2 + kRevCap * 12;
  return total;
}

// ---------------------------------------------------- bookkeeping kernels --
// Length classification = a two-pass counting sort of the sentence indices by (length class, length sub-bucket):
// the class decides which kernels / scratch stride a sentence gets, the sub-bucket only orders the class list so
// that the 64 sentences of a tile have similar lengths whatever the order of the input (the lanes of a tile run
// in lock step: an unsorted corpus cost 1.6x the search iterations of a length-bucketed one, profiles/).
constexpr int kSubBuckets = 16;                        // default number of length sub-buckets per class
constexpr int kMaxSubBuckets = 32;                     // ClassifyArgs::sub_buckets may go up to this
constexpr int kSortKeys = 2 * kMaxClasses * kMaxSubBuckets;   // capacity of the key tables (plain + set-aside keys)
constexpr int kClassifyChunk = 16;   // a wave takes chunks of 16 x 64 sentences
constexpr uint32_t kClassifyLdsWords = 3u * kSortKeys;

// The PLAIN scan (round 4).  The word kernels (kernels_word.h) take sentences that are plain ASCII words; a sentence
// with any other byte (below 0x20, 0x7F and above: control characters, UTF-8) leaves them for the general kernels
// anyway -- after costing the word round an iteration per word, and with the general launch waiting for the word round
// to name it.  PlainScanKernel (below) therefore reads the text once and flags every sentence holding such a byte; with
// `flags` set the two classify passes send the flagged sentences to a second set of class lists (lists2): the general
// launch over them starts NEXT TO the first word round instead of after it.  The flag only routes: both kernel families
// encode any sentence they are given.
struct ClassifyArgs {
  const uint64_t *offs;
  uint32_t n;
  uint32_t n_classes;
  uint32_t rcap[kMaxClasses];   // ascending; sentences longer than the last go to the last class
  uint32_t *lists;              // n_classes x n
  uint32_t *list_counts;        // n_classes (zeroed before launch; written by the scatter pass)
  uint32_t *key_totals;         // kSortKeys (zeroed): sentences per (class, sub-bucket), from the count pass
  uint32_t *key_cursor;         // kSortKeys (zeroed): scatter pass progress
  uint32_t sub_buckets;         // length sub-buckets per class (1 .. kMaxSubBuckets): the tiles of the encode kernels
                                // are the more homogeneous the finer the sort
  // the plain scan (all null / unused without it)
  const uint8_t *flags;         // n bytes from PlainScanKernel: 1 = the sentence holds a byte outside 0x20 .. 0x7E
  uint32_t *lists2;             // n_classes x n: the flagged sentences, by class
  uint32_t *list2_counts;       // n_classes
  uint32_t scan_max_rcap;       // classes beyond this size are never split (documents: the few there are go through one launch)
};

// (class << 4 | sub-bucket) of a sentence of len raw bytes
SPMX_DEVICE uint32_t classify_key(const ClassifyArgs &a, uint64_t len) {
  int cls = static_cast<int>(a.n_classes) - 1;
  for (int c = static_cast<int>(a.n_classes) - 2; c >= 0; --c) if (len <= a.rcap[c]) cls = c;
  const uint64_t lo = cls > 0 ? a.rcap[cls - 1] : 0, hi = a.rcap[cls];
  const uint64_t nsub = a.sub_buckets;
  uint64_t sub = len > lo ? (len - lo - 1) * nsub / (hi - lo) : 0;
  if (sub > nsub - 1) sub = nsub - 1;
  return static_cast<uint32_t>(cls) * a.sub_buckets + static_cast<uint32_t>(sub);
}

// bit 7 of every byte of v that is below 0x20 or above 0x7E (a carry / borrow may also flag the byte above a true one:
// harmless, the flag only routes)
SPMX_DEVICE uint32_t nonplain_bits(uint32_t v) {
  return ((((v + 0x01010101u) | v) | ((v - 0x20202020u) & ~v)) & 0x80808080u);
}

// bit 7 of every byte of v that equals 0x20 (exact for every byte: no borrow between bytes)
SPMX_DEVICE uint32_t space_bits(uint32_t v) {
  const uint32_t x = v ^ 0x20202020u;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// (keep_ws) bit 7 of both bytes of every pair of neighbouring 0x20 inside the 16 bytes x, y, z, w, added to m[0 .. 3]
SPMX_DEVICE void doubled_space_bits(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t *m) {
  const uint32_t s[4] = {space_bits(x), space_bits(y), space_bits(z), space_bits(w)};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t in = s[d] & (s[d] >> 8);              // byte i: bytes i and i + 1 of the dword are both spaces
    m[d] |= in | (in << 8);
    if (d < 3 && (s[d] >> 31) && (s[d + 1] & 0x80u)) { m[d] |= 0x80000000u; m[d + 1] |= 0x80u; }
  }
}

// The plain scan proper: the batch's text read ONCE in address order -- the wavefronts of the launch march through it
// side by side like a copy kernel (4 KB per wavefront and step, a wavefront a step when the text is large: the hardware's
// own dispatch order keeps the reads in address order; a scan chunked by classify's 1024-sentence blocks, the
// first version, had 2560 streams 128 KB apart in flight and ran at 2.5 TB/s: profiles/r04_ab_kernel_stats.txt) --
// and flags[k] = 1 for every sentence k that holds a byte outside 0x20 .. 0x7E (flags zeroed before the launch).  The
// owner of a flagged byte is found by a binary search over the offsets (rare: such bytes are).
// keep_ws (a model that keeps extra whitespace, kernels_word.h): a sentence with a leading, a doubled or a trailing
// space is not the word form's either -- flagged as well (a pair of spaces that straddles two 16-byte units is not seen
// here: the word loop meets it and hands the sentence on, one launch later).
constexpr int kPlainScanFlight = 4;                   // a wavefront takes 4 KB per step: 4 units of 16 bytes per lane
struct PlainScanArgs {
  const uint8_t *text;
  uint64_t text_bytes;          // (only sizes the launch: the kernel reads [offs[0], offs[n]))
  const uint64_t *offs;         // n + 1
  uint32_t n;
  uint32_t keep_ws;             // the model does not remove extra whitespace
  uint8_t *flags;               // n
};
// KEEP_WS = a.keep_ws: a kernel of its own, so that the plain form keeps its registers (64: eight wavefronts per SIMD)
template <bool KEEP_WS>
SPMX_DEVICE void plain_scan_block(const PlainScanArgs &a) {
  const int lane = wv::lane();
  if (a.n == 0) return;
  const uint64_t base_addr = reinterpret_cast<uint64_t>(a.text);
  // What is read is the text of the batch's sentences, [offs[0], offs[n]) from the text pointer -- the pointer itself may
  // lie before the buffer (the host forms rebase it by offs[0]).  16-byte units aligned in MEMORY: a unit that holds a
  // sentence's byte lies inside the buffer's pages; what it holds before the first or beyond the last such byte belongs
  // to no sentence and is ignored.
  const uint64_t t0 = a.offs[0], t1 = a.offs[a.n];
  // The word kernels read 16 + 4 bytes from a word's start whatever its length and therefore leave the sentences that
  // end within 20 bytes of the batch's last byte alone (kernels_word.h): those few are set aside HERE, so that they
  // ride with the general launch beside the first word round instead of costing a tail launch of their own after it.
  if (wv::block_id() == 0 && wv::wave_in_block() == 0 && lane == 0) {
    uint32_t k = a.n;
    for (int step = 0; step < 64 && k > 0u && a.offs[k] + 20u > t1; ++step) a.flags[--k] = 1;
  }
  if (t1 <= t0) return;
  const uint64_t u0 = (base_addr + t0) & ~15ull, u1 = base_addr + t1;
  constexpr int kFlight = kPlainScanFlight;           // 16-byte units in flight per lane
  const uint64_t n_waves = static_cast<uint64_t>(wv::grid_size()) * static_cast<uint64_t>(wv::waves_per_block());
  const uint64_t my_wave = static_cast<uint64_t>(wv::block_id()) * static_cast<uint64_t>(wv::waves_per_block()) + static_cast<uint64_t>(wv::wave_in_block());
  constexpr bool keep_ws = KEEP_WS;
  if (keep_ws) {                                      // a leading / trailing space: a lane per sentence
    for (uint64_t i = my_wave * 64u + static_cast<uint64_t>(lane); i < a.n; i += n_waves * 64u) {
      const uint64_t b = a.offs[i], e = a.offs[i + 1];
      if (e > b && (a.text[b] == 0x20u || a.text[e - 1] == 0x20u)) a.flags[i] = 1;
    }
  }
  const uint64_t step = n_waves * 1024u * kFlight;
  auto owner = [&](uint64_t off) -> uint32_t {       // the last sentence k with offs[k] <= off (offs[0] <= off < offs[n])
    uint32_t lo = 0, hi = a.n;
    while (hi - lo > 1u) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (a.offs[mid] <= off) lo = mid; else hi = mid;
    }
    return lo;
  };
  for (uint64_t u = u0 + my_wave * 1024u * kFlight + static_cast<uint64_t>(lane) * 16u; u < u1; u += step) {
    Q4 v[kFlight];
    uint32_t m[kFlight];
#pragma unroll
    for (int j = 0; j < kFlight; ++j) {
      const uint64_t uj = u + static_cast<uint64_t>(j) * 1024u;
      // (an offset from the kernel's text pointer, not a bare address: a global load, not a flat one)
      v[j] = uj < u1 ? *reinterpret_cast<const Q4 *>(a.text + static_cast<long long>(uj - base_addr))
                     : Q4{0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
    }
    uint32_t any_m = 0;
#pragma unroll
    for (int j = 0; j < kFlight; ++j) {
      m[j] = nonplain_bits(v[j].x) | nonplain_bits(v[j].y) | nonplain_bits(v[j].z) | nonplain_bits(v[j].w);
      if (keep_ws) {
        uint32_t dz[4] = {0u, 0u, 0u, 0u};
        doubled_space_bits(v[j].x, v[j].y, v[j].z, v[j].w, dz);
        m[j] |= dz[0] | dz[1] | dz[2] | dz[3];
      }
      any_m |= m[j];
    }
    if (any_m == 0u) continue;
#pragma unroll
    for (int j = 0; j < kFlight; ++j) {
      if (m[j] == 0u) continue;
      const uint64_t uj = u + static_cast<uint64_t>(j) * 1024u;
      uint32_t z[4] = {nonplain_bits(v[j].x), nonplain_bits(v[j].y), nonplain_bits(v[j].z), nonplain_bits(v[j].w)};
      if (keep_ws) doubled_space_bits(v[j].x, v[j].y, v[j].z, v[j].w, z);
      int b_lo = -1, b_hi = -1;                       // first and last flagged byte of the unit
      for (int d = 0; d < 4; ++d)
        for (int q = 0; q < 4; ++q)
          if (z[d] & (0x80u << (8 * q))) { if (b_lo < 0) b_lo = 4 * d + q; b_hi = 4 * d + q; }
      // (a flag raised by a carry sits one byte above a true one: still inside the same or the next sentence -- a
      // sentence flagged for nothing only takes the general kernels)
      if (uj + static_cast<uint64_t>(b_hi) < base_addr + t0) continue;                 // all of it before the first sentence
      uint64_t o_lo = uj + static_cast<uint64_t>(b_lo) < base_addr + t0 ? t0 : uj + static_cast<uint64_t>(b_lo) - base_addr;
      uint64_t o_hi = uj + static_cast<uint64_t>(b_hi) - base_addr;
      if (o_hi >= t1) o_hi = t1 - 1;
      if (t1 == t0 || o_lo >= t1 || o_lo > o_hi) continue;
      const uint32_t k_lo = owner(o_lo);
      const uint32_t k_hi = o_hi < a.offs[k_lo + 1] ? k_lo : owner(o_hi);      // (mostly the same sentence)
      a.flags[k_lo] = 1;
      a.flags[k_hi] = 1;
      if (k_hi - k_lo <= 17u) for (uint32_t k = k_lo + 1; k < k_hi; ++k) a.flags[k] = 1;   // (more than that between two bytes of a unit: empty sentences)
    }
  }
}

// PASS 0 counts, PASS 1 scatters.  lds: kClassifyLdsWords words (per wave).
template <int PASS>
SPMX_DEVICE void classify_block(const ClassifyArgs &a, uint32_t *lds) {
  const int lane = wv::lane();
  uint32_t *hist = lds, *start = lds + kSortKeys, *base = lds + 2 * kSortKeys;
  const bool scan = a.flags != nullptr;
  const int nsub = static_cast<int>(a.sub_buckets);
  const int n_plain_keys = static_cast<int>(a.n_classes) * nsub;
  const int n_keys = scan ? 2 * n_plain_keys : n_plain_keys;
  if (PASS == 1) {
    // where every key's run begins: its class list + the keys of the same list before it
    for (int k = lane; k < n_keys; k += 64) {
      const int c = k / nsub;
      uint32_t before = 0;
      for (int j = c * nsub; j < k; ++j) before += a.key_totals[j];
      base[k] = before;
      if (wv::block_id() == 0 && k % nsub == nsub - 1) {
        if (c < static_cast<int>(a.n_classes)) a.list_counts[c] = before + a.key_totals[k];
        else a.list2_counts[c - static_cast<int>(a.n_classes)] = before + a.key_totals[k];
      }
    }
  }
  const uint32_t per_chunk = 64u * kClassifyChunk;
  const uint32_t chunks = (a.n + per_chunk - 1) / per_chunk;
  for (uint32_t ch = static_cast<uint32_t>(wv::block_id()); ch < chunks; ch += static_cast<uint32_t>(wv::grid_size())) {
    const uint32_t first = ch * per_chunk;
    for (int k = lane; k < n_keys; k += 64) hist[k] = 0;
    wv::sync();
    uint32_t keys[kClassifyChunk];
#pragma unroll
    for (int k = 0; k < kClassifyChunk; ++k) {
      const uint32_t i = first + static_cast<uint32_t>(k) * 64u + static_cast<uint32_t>(lane);
      keys[k] = 0xFFFFFFFFu;
      if (i < a.n) {
        keys[k] = classify_key(a, a.offs[i + 1] - a.offs[i]);
        if (scan) {
          const uint32_t f = a.flags[i];
          if (f && a.rcap[keys[k] / a.sub_buckets] <= a.scan_max_rcap) keys[k] += static_cast<uint32_t>(n_plain_keys);
        }
        wv::lds_atomic_add(&hist[keys[k]], 1u);
      }
    }
    wv::sync();
    if (PASS == 0) {
      for (int k = lane; k < n_keys; k += 64) if (hist[k]) wv::atomic_add(&a.key_totals[k], hist[k]);
    } else {
      // reserve this chunk's slice of every key's run, then hand out the slots (order inside a slice is arbitrary)
      for (int k = lane; k < n_keys; k += 64) {
        start[k] = hist[k] ? wv::atomic_add(&a.key_cursor[k], hist[k]) : 0u;
        hist[k] = 0;
      }
      wv::sync();
#pragma unroll
      for (int k = 0; k < kClassifyChunk; ++k) {
        const uint32_t i = first + static_cast<uint32_t>(k) * 64u + static_cast<uint32_t>(lane);
        if (keys[k] != 0xFFFFFFFFu) {
          const uint32_t key = keys[k], c = key / a.sub_buckets;
          const uint32_t r = wv::lds_atomic_add(&hist[key], 1u);
          uint32_t *list = c < a.n_classes ? a.lists + static_cast<uint64_t>(c) * a.n
                                           : a.lists2 + static_cast<uint64_t>(c - a.n_classes) * a.n;
          list[base[key] + start[key] + r] = i;
        }
      }
    }
    wv::sync();
  }
}

// counts[n] -> exclusive prefix sums id_offs[n + 1] (uint64), in three passes
// over tiles of kScanTile sentences: tile sums, scan of tile sums (one wave),
// final offsets.
constexpr uint32_t kScanTile = 2048;   // 64 lanes x 32 consecutive counts

struct ScanArgs {
  const uint32_t *counts;
  uint32_t n;
  uint64_t *tile_sums;    // ceil(n / kScanTile) + 1
  uint64_t *id_offs;      // n + 1
};

SPMX_DEVICE uint64_t wave_excl_scan64(uint64_t v, int lane, uint64_t *total) {
  uint64_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t lo = static_cast<uint32_t>(wv::shfl_up(static_cast<int>(static_cast<uint32_t>(incl)), d));
    const uint32_t hi = static_cast<uint32_t>(wv::shfl_up(static_cast<int>(static_cast<uint32_t>(incl >> 32)), d));
    if (lane >= d) incl += (static_cast<uint64_t>(hi) << 32 | lo);
  }
  const uint32_t tl = wv::shfl(static_cast<uint32_t>(incl), 63), th = wv::shfl(static_cast<uint32_t>(incl >> 32), 63);
  *total = static_cast<uint64_t>(th) << 32 | tl;
  return incl - v;
}

// Whole tiles take the coalesced form below when the two arrays are aligned for it (they are, wherever the caller's offsets
// come from an allocator): a lane owns TWO consecutive counts of a 128-count row, 16 rows to the tile, so a wave's load is
// 512 contiguous bytes and its store of offsets 1 KB of whole lines -- the lane-owns-32-consecutive form wrote every
// 32-byte granule in four visits of 8 bytes (rocprofv3 WRITE_SIZE: 610 MB for the 80 MB of C2's offsets, 0.17 ms).  The
// row's prefix sums are two DPP scans of 32-bit halves (26 + 7 bits: 64 lanes of them cannot carry out).
SPMX_DEVICE bool scan_rows_ok(const ScanArgs &a) {
  return ((reinterpret_cast<uintptr_t>(a.counts) & 7u) | (reinterpret_cast<uintptr_t>(a.id_offs) & 15u)) == 0;
}
constexpr uint32_t kScanRows = kScanTile / 128;

SPMX_DEVICE void scan_tiles_block(const ScanArgs &a) {     // pass 1
  const int lane = wv::lane();
  const uint32_t tiles = (a.n + kScanTile - 1) / kScanTile;
  const bool rows_ok = scan_rows_ok(a);
  for (uint32_t tile = static_cast<uint32_t>(wv::block_id()); tile < tiles; tile += static_cast<uint32_t>(wv::grid_size())) {
    uint64_t s = 0;
    if (rows_ok && tile * kScanTile + kScanTile <= a.n) {
      const D2 *src = reinterpret_cast<const D2 *>(a.counts + static_cast<size_t>(tile) * kScanTile) + lane;
#pragma unroll
      for (uint32_t k = 0; k < kScanRows; ++k) { const D2 c = src[k * 64]; s += static_cast<uint64_t>(c.x) + c.y; }
    } else {
      const uint32_t first = tile * kScanTile + static_cast<uint32_t>(lane) * 32;
      for (uint32_t k = 0; k < 32; ++k) if (first + k < a.n) s += a.counts[first + k];
    }
    uint64_t total = 0;
    wave_excl_scan64(s, lane, &total);
    if (lane == 0) a.tile_sums[tile] = total;
  }
}

SPMX_DEVICE void scan_sums_block(const ScanArgs &a) {      // pass 2, ONE wave
  const int lane = wv::lane();
  const uint32_t tiles = (a.n + kScanTile - 1) / kScanTile;
  uint64_t carry = 0;
  for (uint32_t b = 0; b < tiles; b += 64) {
    const uint32_t t = b + static_cast<uint32_t>(lane);
    const uint64_t v = t < tiles ? a.tile_sums[t] : 0;
    uint64_t total = 0;
    const uint64_t ex = wave_excl_scan64(v, lane, &total);
    if (t < tiles) a.tile_sums[t] = carry + ex;
    carry += total;
  }
  if (lane == 0) { a.tile_sums[tiles] = carry; a.id_offs[a.n] = carry; }
}

SPMX_DEVICE void scan_final_block(const ScanArgs &a) {     // pass 3
  const int lane = wv::lane();
  const uint32_t tiles = (a.n + kScanTile - 1) / kScanTile;
  const bool rows_ok = scan_rows_ok(a);
  for (uint32_t tile = static_cast<uint32_t>(wv::block_id()); tile < tiles; tile += static_cast<uint32_t>(wv::grid_size())) {
    if (rows_ok && tile * kScanTile + kScanTile <= a.n) {
      const D2 *src = reinterpret_cast<const D2 *>(a.counts + static_cast<size_t>(tile) * kScanTile) + lane;
      L2 *dst = reinterpret_cast<L2 *>(a.id_offs + static_cast<size_t>(tile) * kScanTile) + lane;
      uint64_t carry = a.tile_sums[tile];
      D2 c[kScanRows];
#pragma unroll
      for (uint32_t k = 0; k < kScanRows; ++k) c[k] = src[k * 64];
#pragma unroll
      for (uint32_t k = 0; k < kScanRows; ++k) {
        const uint64_t v = static_cast<uint64_t>(c[k].x) + c[k].y;
        const uint32_t lo = wv::scan_add(static_cast<uint32_t>(v) & 0x3FFFFFFu), hi = wv::scan_add(static_cast<uint32_t>(v >> 26));
        const uint64_t before = carry + lo + (static_cast<uint64_t>(hi) << 26) - v;
        L2 o;
        o.x = before;
        o.y = before + c[k].x;
        dst[k * 64] = o;
        carry += static_cast<uint64_t>(wv::read_lane(lo, 63)) + (static_cast<uint64_t>(wv::read_lane(hi, 63)) << 26);
      }
      continue;
    }
    const uint32_t first = tile * kScanTile + static_cast<uint32_t>(lane) * 32;
    uint64_t s = 0;
    for (uint32_t k = 0; k < 32; ++k) if (first + k < a.n) s += a.counts[first + k];
    uint64_t total = 0;
    uint64_t run = a.tile_sums[tile] + wave_excl_scan64(s, lane, &total);
    for (uint32_t k = 0; k < 32; ++k) {
      if (first + k < a.n) { a.id_offs[first + k] = run; run += a.counts[first + k]; }
    }
  }
}

// tmp_off[s] with this bit: sentence s's ids are 16-bit values, the rest of the word is their offset in 16-BIT units from
// the arena's start (the word kernels, EncodeArgs::ids16); without it 32-bit values at an offset in 32-bit units
constexpr unsigned long long kTmpOffHalf = 1ull << 63;
struct CompactArgs {
  const int32_t *arena;
  const uint64_t *tmp_off;
  const uint32_t *counts;
  const uint64_t *id_offs;
  int32_t *ids;
  uint64_t ids_cap;
  uint32_t n;
  const uint32_t *status = nullptr;   // the call's kSt* bits (null: not looked at): after an arena overflow nothing is moved -- some
                                      // kernels leave the range they ASKED for in tmp_off / counts, beyond the arena's end
  uint32_t staged = 2048;   // ids the LDS image of a wave holds: blocks of 16-bit ids go through it (0: the search form for every block)
  // blocks of more than big_ids ids (documents) are not moved by their wave: it lists them, and the launch behind
  // (compact_big_block) moves them with every wave of its grid (0 / null: no such list, every block is moved here)
  uint32_t big_ids = 0;
  uint32_t *big_list = nullptr;     // [(n + 63) / 64]
  uint32_t *big_count = nullptr;
};
constexpr uint32_t kCompactBigIds = 32768;     // a block of 64 sentences with more ids than this is a block of documents
constexpr uint32_t kCompactBigChunk = 8192;    // ids of one work item of compact_big_block

// LDS of a CompactKernel wave: the 16-bit ids of its 64 sentences (or of a half, a quarter of them), in CSR order
constexpr uint32_t kCompactLdsIdsMax = 16384;
constexpr uint32_t CompactLdsBytes(uint32_t ids) { return ids ? (ids + 8u) * 2u : 0u; }

// Moves every sentence's ids from where its wave happened to put them in the arena to their place in the
// caller's CSR.  A wave takes 64 consecutive sentences: their CSR range is contiguous, so the output is written
// as one coalesced stream; every output element finds its sentence by a 6-step binary search over the 64
// per-lane start offsets (cross-lane reads) and gathers from that sentence's arena slot -- 32-bit ids, or 16-bit ones
// where the word kernels wrote them (kTmpOffHalf).  (One wave per sentence, the first version, used 28 of 64 lanes and
// re-read three offsets per sentence: 1.1 TB/s.  Round 4 tried FOUR ids per lane and round -- one search per quad, a
// 16-byte read at the slot's alignment, a 16-byte store: 1.06 ms against this form's 0.81 on the same 10 M sentences,
// profiles/r04_ab_kernel_stats.txt: the quads that cross a sentence boundary, one in seven, run a second path in
// almost every round.)
//
// Blocks whose sentences all hold 16-bit ids (the word kernels' output: C2, C3) and fit the wave's LDS take the STAGED form
// (round 5): a lane copies ITS sentence from the arena into LDS at the sentence's place in the block's CSR range -- 16-byte
// reads at the slot's own alignment, eight ids each, four or five of them for a C2 sentence -- and the wave then streams
// the LDS image out, four ids a lane as one aligned 16-byte store.  No search at all: about 300 instructions for a block
// of 1900 ids against 1350 (30 rounds x (9 cross-lane reads + a 2-byte load + a 4-byte store)).
SPMX_DEVICE void compact_block(const CompactArgs &a, uint16_t *lds) {
  const int lane = wv::lane();
  if (a.id_offs[a.n] > a.ids_cap) return;   // caller sees the needed size in id_offs[n]
  if (a.status != nullptr && (*a.status & kStArenaOverflow)) return;   // the caller encodes the batch again with a larger arena
  const uint32_t blocks = (a.n + 63) / 64;
  const uint16_t *arena16 = reinterpret_cast<const uint16_t *>(a.arena);
  const uint32_t cap = (reinterpret_cast<uintptr_t>(a.arena) & 15u) == 0 ? a.staged : 0u;   // ids the LDS image holds
  // (a block's three reads are asked for one block ahead: they are in while the block before is being moved)
  uint64_t my_dst = 0, my_src = 0, end_dst = 0;
  auto ask = [&](uint32_t b, uint64_t *dst, uint64_t *src, uint64_t *end) __attribute__((always_inline)) {
    const uint32_t s = b * 64 + static_cast<uint32_t>(lane);
    *dst = a.id_offs[s < a.n ? s : a.n];                       // id_offs[n] closes the last block
    *src = a.tmp_off[s < a.n ? s : a.n - 1u];
    *end = a.id_offs[(b * 64 + 64 <= a.n) ? b * 64 + 64 : a.n];
  };
  uint32_t b = static_cast<uint32_t>(wv::block_id());
  if (b < blocks) ask(b, &my_dst, &my_src, &end_dst);
  for (; b < blocks; b += static_cast<uint32_t>(wv::grid_size())) {
    const uint32_t nb = b + static_cast<uint32_t>(wv::grid_size());
    uint64_t n_dst = 0, n_src = 0, n_end = 0;
    if (nb < blocks) ask(nb, &n_dst, &n_src, &n_end);
    const uint64_t dst0 = (static_cast<uint64_t>(wv::shfl(static_cast<uint32_t>(my_dst >> 32), 0)) << 32) |
                          wv::shfl(static_cast<uint32_t>(my_dst), 0);
    const uint32_t total = static_cast<uint32_t>(end_dst - dst0);
    const uint32_t rel = static_cast<uint32_t>(my_dst - dst0);
    const uint32_t src_lo = static_cast<uint32_t>(my_src), src_hi = static_cast<uint32_t>(my_src >> 32);
    my_dst = n_dst; my_src = n_src; end_dst = n_end;
    if (a.big_ids != 0u && total > a.big_ids) {   // documents: for every wave of the launch behind
      if (lane == 0) a.big_list[wv::atomic_add(a.big_count, 1u)] = b;
      continue;
    }
    // the block as one part, two halves or four quarters of its lanes: the fewest whose ids each fit the image
    uint32_t parts = 0u;
    if (cap) {
      const uint32_t r16 = wv::shfl(rel, 16), r32 = wv::shfl(rel, 32), r48 = wv::shfl(rel, 48);
      const uint32_t h0 = r32, h1 = total - r32;
      const uint32_t q0 = r16, q1 = r32 - r16, q2 = r48 - r32, q3 = total - r48;
      const uint32_t hmax = h0 > h1 ? h0 : h1;
      const uint32_t qa = q0 > q1 ? q0 : q1, qb = q2 > q3 ? q2 : q3, qmax = qa > qb ? qa : qb;
      parts = total + 3u <= cap ? 1u : hmax + 3u <= cap ? 2u : qmax + 3u <= cap ? 4u : 0u;
    }
    const uint32_t nxt = wv::shfl(rel, (lane + 1) & 63);
    const uint32_t cnt_all = (lane == 63 ? total : nxt) - rel;   // (lanes past the last sentence sit at `total`: 0)
    if (parts && !wv::any(cnt_all != 0u && (src_hi >> 31) == 0u)) {
      const uint64_t base16 = (static_cast<uint64_t>(src_hi & 0x7FFFFFFFu) << 32) | src_lo;
      const uint32_t mis = static_cast<uint32_t>(base16) & 7u;
      const Q4 *q = reinterpret_cast<const Q4 *>(arena16 + (base16 - mis));
      const uint32_t lanes_per = 64u / parts;
      for (uint32_t part = 0; part < parts; ++part) {
        const uint32_t l0 = part * lanes_per, l1 = l0 + lanes_per;
        const uint32_t rel0 = wv::shfl(rel, static_cast<int>(l0));
        const uint32_t ptotal = (l1 == 64u ? total : wv::shfl(rel, static_cast<int>(l1 & 63u))) - rel0;
        const bool in = static_cast<uint32_t>(lane) >= l0 && static_cast<uint32_t>(lane) < l1;
        const uint32_t cnt = in ? cnt_all : 0u;
        // where the part's first id lands within its 16-byte unit of the output: LDS entry e <-> a.ids[dst0 + rel0 - m + e]
        const uint32_t m = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(a.ids + dst0 + rel0) >> 2) & 3u;
        const uint32_t groups = cnt ? (mis + cnt + 7u) >> 3 : 0u;
        uint16_t *mine = lds + m + (rel - rel0);                 // (only dereferenced where cnt != 0: a lane of the part)
        // (four reads in flight a step; a lane that has run out re-reads its last unit -- or the arena's first one if it
        // has no ids: its tmp_off may be anything -- and places nothing: every position is past its count)
        const Q4 *qs = groups ? q : reinterpret_cast<const Q4 *>(arena16);
        const uint32_t glast = groups ? groups - 1u : 0u;
        auto place = [&](const Q4 &v, uint32_t g) __attribute__((always_inline)) {
          const int p0 = static_cast<int>(g * 8u) - static_cast<int>(mis);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int pp = p0 + i;
            if (pp >= 0 && pp < static_cast<int>(cnt)) mine[pp] = static_cast<uint16_t>(w[i >> 1] >> ((i & 1) * 16));
          }
        };
        for (uint32_t g = 0; wv::any(g < groups); g += 4) {
          const Q4 v0 = qs[g < groups ? g : glast];
          const Q4 v1 = qs[g + 1u < groups ? g + 1u : glast];
          const Q4 v2 = qs[g + 2u < groups ? g + 2u : glast];
          const Q4 v3 = qs[g + 3u < groups ? g + 3u : glast];
          place(v0, g);
          place(v1, g + 1u);
          place(v2, g + 2u);
          place(v3, g + 3u);
        }
        wv::sync();
        const uint32_t end = m + ptotal;
        int32_t *out = a.ids + dst0 + rel0 - m;
        for (uint32_t e0 = 0; e0 < end; e0 += 256u) {
          const uint32_t e = e0 + static_cast<uint32_t>(lane) * 4u;
          if (e < end) {
            const D2 h = *reinterpret_cast<const D2 *>(lds + e);
            const uint32_t i0 = h.x & 0xFFFFu, i1 = h.x >> 16, i2 = h.y & 0xFFFFu, i3 = h.y >> 16;
            if (e >= m && e + 4u <= end) {
              Q4 o;
              o.x = i0; o.y = i1; o.z = i2; o.w = i3;
              *reinterpret_cast<Q4 *>(out + e) = o;
            } else {
              if (e + 0u >= m && e + 0u < end) out[e + 0u] = static_cast<int32_t>(i0);
              if (e + 1u >= m && e + 1u < end) out[e + 1u] = static_cast<int32_t>(i1);
              if (e + 2u >= m && e + 2u < end) out[e + 2u] = static_cast<int32_t>(i2);
              if (e + 3u >= m && e + 3u < end) out[e + 3u] = static_cast<int32_t>(i3);
            }
          }
        }
        wv::sync();                                              // (the next part's sentences overwrite the image)
      }
      continue;
    }
    const uint32_t rounds = (total + 63) / 64;
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint32_t j = r * 64 + static_cast<uint32_t>(lane);
      int lo = 0;                                              // the last sentence of the block that starts at or before j
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) {
        const uint32_t v = wv::shfl(rel, lo + step);
        if (v <= j) lo += step;
      }
      const uint32_t r0 = wv::shfl(rel, lo);
      const uint32_t hi = wv::shfl(src_hi, lo);
      const uint64_t base = (static_cast<uint64_t>(hi & 0x7FFFFFFFu) << 32) | wv::shfl(src_lo, lo);
      if (j < total) a.ids[dst0 + j] = (hi >> 31) ? static_cast<int32_t>(arena16[base + (j - r0)]) : a.arena[base + (j - r0)];
    }
  }
}

// The blocks compact_block listed (documents: 64 sentences of 230 K ids each would keep ONE wave busy for 40 ms while the
// chip idles -- 158 of the 611 ms of round 5's 256 x 1 MiB step): their sentences, cut into items of kCompactBigChunk
// ids, dealt round-robin to every wave of the grid; an item is one coalesced stream.
SPMX_DEVICE void compact_big_block(const CompactArgs &a) {
  const int lane = wv::lane();
  const uint32_t listed = *a.big_count;
  if (listed == 0u) return;
  if (a.id_offs[a.n] > a.ids_cap) return;
  if (a.status != nullptr && (*a.status & kStArenaOverflow)) return;
  const uint16_t *arena16 = reinterpret_cast<const uint16_t *>(a.arena);
  // a block's items are dealt to a GROUP of waves: the whole grid when few blocks are listed (four blocks of megabyte
  // documents), one wave per block when many are (the long sentences of a mixed batch: every wave scanning every listed
  // block cost C5 2 ms)
  const uint32_t n_waves = static_cast<uint32_t>(wv::grid_size()), wave = static_cast<uint32_t>(wv::block_id());
  const uint32_t per_block = listed < n_waves ? n_waves / listed : 1u;
  const uint32_t groups = n_waves / per_block;
  if (wave / per_block >= groups) return;
  const uint64_t nw = per_block;
  for (uint32_t i = wave / per_block; i < listed; i += groups) {
    uint64_t w = 0, mine = wave % per_block;                           // the block's items so far; this wave's next item
    const uint32_t b = a.big_list[i];
    const uint32_t s = b * 64u + static_cast<uint32_t>(lane);
    const bool have = s < a.n;
    const uint64_t dst = have ? a.id_offs[s] : 0u;
    const uint64_t src = have ? a.tmp_off[s] : 0u;
    const uint32_t cnt = have ? static_cast<uint32_t>(a.id_offs[s + 1u] - dst) : 0u;
    for (int k = 0; k < 64; ++k) {
      const uint32_t c = wv::shfl(cnt, k);
      const uint64_t items = (static_cast<uint64_t>(c) + kCompactBigChunk - 1u) / kCompactBigChunk;
      if (w + items <= mine) { w += items; continue; }
      const uint64_t d0 = (static_cast<uint64_t>(wv::shfl(static_cast<uint32_t>(dst >> 32), k)) << 32) | wv::shfl(static_cast<uint32_t>(dst), k);
      const uint32_t s_hi = wv::shfl(static_cast<uint32_t>(src >> 32), k), s_lo = wv::shfl(static_cast<uint32_t>(src), k);
      const uint64_t base = (static_cast<uint64_t>(s_hi & 0x7FFFFFFFu) << 32) | s_lo;
      const bool half = (s_hi >> 31) != 0u;
      for (; mine < w + items; mine += nw) {
        const uint32_t j0 = static_cast<uint32_t>(mine - w) * kCompactBigChunk;
        const uint32_t j1 = j0 + kCompactBigChunk < c ? j0 + kCompactBigChunk : c;
        for (uint32_t j = j0 + static_cast<uint32_t>(lane); j < j1; j += 256u) {   // four loads in flight
          int32_t v[4];
#pragma unroll
          for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t jj = j + 64u * u;
            v[u] = jj < j1 ? (half ? static_cast<int32_t>(arena16[base + jj]) : a.arena[base + jj]) : 0;
          }
#pragma unroll
          for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t jj = j + 64u * u;
            if (jj < j1) a.ids[d0 + jj] = v[u];
          }
        }
      }
      w += items;
    }
  }
}

}  // namespace spmx

#include "kernels_bpe_stream.h"
#include "kernels_stream.h"
#include "kernels_word.h"
#include "kernels_wordwave.h"
#include "kernels_decode.h"
#include "kernels_split.h"
#include "kernels_align.h"
#include "kernels_normalize.h"
#include "kernels_nbest.h"
#include "kernels_long.h"
#include "kernels_uniwave.h"
#include "kernels_gather.h"

#endif
