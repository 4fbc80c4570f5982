// Unigram segmentation, WORD form: one sentence per lane, one WORD per iteration, straight from the raw text.
// Reference: unigram::Model::EncodeOptimized (src/unigram_model.cc:889-1020) behind Normalizer::Normalize
// (src/normalizer.cc:71-186).
//
// Why a word at a time is legal.  In a model whose pieces carry the space symbol only as their FIRST character (dev.h
// kNfUniWordwise; the trainer's default split_by_whitespace guarantees it) no piece and no UNK candidate (one character,
// :995-1005) spans the boundary between two words, a WORD being a space symbol and the characters up to the next one.
// Every path of the lattice therefore passes through every word boundary, and the best path of the sentence is the
// concatenation of the best paths of its words -- in exact arithmetic.  The reference does not compute in exact
// arithmetic: a candidate is (double)score + (double)best[start], compared against and stored as a float (:979-989), so
// WHICH of two close candidates wins may depend on the magnitude of the score accumulated before the word.  The word
// memo (dev.h umemo16 / umemo, tables.cc BuildWordMemo) therefore stores, next to a word's best segmentation, the
// largest magnitude of the accumulated score up to which the decision is provably the exact-arithmetic one:
//
//   * all float values inside the word lie within M = |B| + wmag of zero (B = best_path_score at the word's start, wmag
//     = the largest |partial sum| of any candidate inside the word), so every stored best[e] differs from its exact
//     value by at most k(e) * U, U = ulp_float(M) / 2, k(e) <= the characters before e (one rounding per piece, by
//     induction over :982-989; the double add itself is exact for operands a float apart);
//   * the candidate that is best at e in exact arithmetic wins the fold at e whatever the order of the candidates if it
//     leads the second best by more than (2 * nchar + 2) * U;
//   * tables.cc computes, in double, the smallest such lead `gap` over the positions ON the word's best path and stores
//     bmax = the largest |B| for which ulp_float(|B| + wmag) <= gap / (4 * nchar + 4) -- four times the bound above.
//
// A memo entry is taken only while |B| < bmax.  The FIRST pass (DP = false) does not even keep B: it keeps an upper
// bound of |B| -- every emitted piece adds ceil(|score|) + 1, which also covers the float roundings of the running sum
// -- and compares that with the power of two below bmax; the entry is 16 bytes (12 key bytes, id, those two small
// numbers), found in a table of the likeliest words held in LDS or else by one probe of the table in HBM.  Anything the
// first pass cannot take (a word that is not in the memo, a bound beyond its margin, a byte outside 0x21-0x7E, a word
// of more than 16 bytes, text within 20 bytes of the end of the buffer) sends the WHOLE sentence on, through per-class
// leftover lists: a pass either produces the reference's ids for a sentence or produces nothing for it.
//
// The SECOND pass (DP = true) runs over the first one's leftovers and keeps B exactly as the reference has it --
// B = (float)((double)score + (double)B) per emitted piece, the very operation of :982-989 along the winning path.  A
// word that is not in the memo is then segmented by the lane itself with the reference's own recurrence over the word's
// characters from that B (uni_word_dp: exact, no margin involved); lanes wait at such a word until no lane of the wave
// can go on, and all of them run their DP together.  What the second pass cannot take either (non-ASCII text, more than
// kWordDpMax such words in a sentence) goes to the general kernels (kernels_stream.h).
//
// The CALL-LOCAL memo.  The load-time memo knows the vocabulary's own whole words; a corpus has others (rarer words
// that split into several pieces).  A sentence with ONE such word would otherwise leave the word form altogether, and
// on natural text most sentences have one.  So the first pass COLLECTS: a word that is not in the memo is entered --
// once, by a 64-bit compare-and-swap on a hash of its bytes -- into a table in HBM, the lane goes on through its
// sentence looking for more, and the sentence is kept for a second round.  A small kernel then segments every
// collected word ONCE, one lane per word (word_resolve_block): BPE by the merge loop (exact); unigram by
// EncodeOptimized of the word from score 0 in float with the margin analysis above done on the device (the float
// roundings of this DP are charged to the margin).  The second round (the same word loop, now looking a missed word
// up in the call-local table; up to 8 pieces per word, or 9 .. 16 as 16-bit ids: kDynWide) then takes those sentences.  Two different words with the same
// 64-bit hash: the second finds the first's bytes in the entry, which is a miss -- its sentence takes the general
// kernels; nothing wrong can come out.
//
// What the normalizer contributes is implicit: the model must add a dummy prefix and escape whitespace with the
// one-byte space symbol, and every byte 0x20-0x7E must be a character no charsmap rule starts with (tables.cc checks
// all of it), so Normalize() of such a sentence is "words joined by single space symbols, one in front" -- a word's
// normalized form is the space symbol plus its raw bytes, which is how the memo is keyed (by the raw bytes alone).
// Leading, trailing and doubled spaces: a model that removes extra whitespace drops them -- empty words, an iteration
// each; a model that KEEPS it (Llama style) has runs of space symbols there, which are not words -- such a sentence is
// left to the general kernels (keep_ws below).
//
// No scratch in HBM, no back-pointers, no backtrack: ids leave in forward order through an 8-id LDS staging column as
// 32-byte bursts.
#ifndef SPMX_KERNELS_WORD_H_
#define SPMX_KERNELS_WORD_H_

namespace spmx {

constexpr uint32_t kWordMaskBytes = 18u * 16u + 32u;            // mask rows 0 .. 17 (+ padding)
constexpr uint32_t kWordLdsShared = kWordMaskBytes + kWordHotSlots * 16u;
constexpr uint32_t kWordStage = 8;                              // ids per burst
constexpr uint32_t kWordDpPos = 18;                             // positions of a word in the DP: space symbol + 16 bytes + end
constexpr int kWordDpMax = 4;                                   // words per sentence the second pass segments itself
constexpr uint32_t kDynMaxIds = 8;                              // pieces per word of the call-local memo (32-bit ids) ...
constexpr uint32_t kDynMaxWide = 16;                            // ... or 9 .. 16 of them as 16-bit values (a WIDE entry: kDynWide in its count word)
constexpr uint32_t kDynFirstUnk = 0x100u, kDynLastUnk = 0x200u, kDynWide = 0x400u;   // flags next to the count
constexpr uint32_t kDynProbes = 24;                             // slots tried before a word is given up
// word modes of uni_word_lane: plain; collecting (first round of a call that has a call-local memo); looking the
// call-local memo up (second round)
enum { kWmPlain = 0, kWmCollect = 1, kWmDyn = 2 };
SPMX_DEVICE unsigned long long DynTag(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
  const uint32_t h1 = HashWordKey(k0, k1, k2, k3);
  const unsigned long long t = (static_cast<unsigned long long>(h1) << 32) | UallHash2(k0, k1, k2, k3, h1);
  return t | 1ull;                                              // (0 means "free")
}
SPMX_HD inline uint32_t WordLdsPerWave(bool dp) {
  return 64u * kWordStage * 4u + (dp ? 64u * (kWordDpPos * 8u + 20u) : 0u);
}
SPMX_HD inline uint32_t WordLdsBytes(uint32_t waves, bool dp) { return kWordLdsShared + waves * WordLdsPerWave(dp); }

// 16 / 4 bytes at any address (gfx950 runs with unaligned vector memory access enabled; scripts/ubench/unaligned_probe.hip)
struct __attribute__((packed, aligned(1))) Q4U { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U1U { uint32_t x; };

// bit 7 of every byte of v that equals 0x20, exact for every byte (no borrow between bytes): the word loop asks for the
// first AND the second 0x20 of its window
SPMX_DEVICE uint32_t space_flags(uint32_t v) {
  const uint32_t x = v ^ 0x20202020u;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// A memo KEY is the word's L raw bytes padded with 0x20 to 12 / 16 bytes.  A word holds no 0x20 (it ends at the first
// one), so the padding can be told from every byte a word may consist of -- 0x00 included, which the reference keeps
// as a character of its own (src/normalizer.cc:231-244) -- and equal keys mean equal bytes AND equal length.
// (one v_bfi_b32: the mask's bits from the text, the others from the padding)
constexpr uint32_t kWordKeyPad = 0x20202020u;
SPMX_DEVICE uint32_t key_dword(uint32_t text, uint32_t mask) { return (text & mask) | (kWordKeyPad & ~mask); }

struct WordLds {
  const Q4 *masks;      // [18] key masks (shared)
  const U4 *hot;        // [kWordHotSlots] the likeliest words, umemo16 format (shared)
  int32_t *stage;       // this lane's id staging column: entry k at stage[k << 6]
  float *dp_best;       // (DP) this lane's best_path_score column: position i at dp_best[i << 6]
  uint32_t *dp_bp;      // (DP) back-pointer words  id | length << 24 | unk << 31  (0: not reached)
  uint8_t *dp_bytes;    // (DP) the word in device form: [20] bytes of this lane
};

// EncodeOptimized (src/unigram_model.cc:957-1008) of ONE word -- the space symbol and the L raw bytes in key[] -- from
// best_path_score B at its start, by this lane alone: the reference's loops flattened, one trie probe per iteration.
// Every byte is one character (ASCII, or the one-byte space symbol).  On return T.dp_bp holds the back-pointers and
// T.dp_best[n] the score at the word's end (n = L + 1).  `active`: this lane has a word to do.
SPMX_DEVICE void uni_word_dp(const SpmxDev &d, const WordLds &T, int L, float B, bool active) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  const int n = L + 1;
  if (active) {
    for (int i = 0; i <= n; ++i) T.dp_bp[i << 6] = 0u;
    T.dp_best[0] = B;
  }
  int s = 0, dep = 0;
  uint32_t node = root;
  float bs = B;
  bool single = false;
  while (wv::any(active)) {
    if (active) {
      const uint32_t c = T.dp_bytes[s + dep];
      const U4 u = ptrie[node ^ c];
      bool match = (u.x & 0x1FFu) == (0x100u | c);                                  // :969-971
      if (match) {
        ++dep;
        node = u.x >> kDatBaseShiftDev;
        if ((u.x & kDatTerminalDev) && !(u.y & kPtUnused)) {                         // :973-974
          const int e = s + dep;
          const double cand = static_cast<double>(wv::bits_to_float(u.z)) + static_cast<double>(bs);   // :982-983
          if (T.dp_bp[e << 6] == 0u || cand > static_cast<double>(T.dp_best[e << 6])) {                // :984-989
            T.dp_best[e << 6] = static_cast<float>(cand);
            T.dp_bp[e << 6] = (u.y & kBwIdMask) | (static_cast<uint32_t>(dep) << kBwLenShift);
          }
          if (dep == 1) single = true;                                               // :990 (every character is one byte)
        }
        if (s + dep >= n) match = false;                                             // the word ends: this start is done
      }
      if (!match) {
        if (!single) {                                                               // :995-1005 UNK, float arithmetic
          const float cand = d.unk_score + bs;
          if (T.dp_bp[(s + 1) << 6] == 0u || cand > T.dp_best[(s + 1) << 6]) {
            T.dp_best[(s + 1) << 6] = cand;
            T.dp_bp[(s + 1) << 6] = (1u << kBwLenShift) | kBwUnk;
          }
        }
        ++s;                                                                         // :1007
        if (s >= n) active = false;
        else { bs = T.dp_best[s << 6]; node = root; dep = 0; single = false; }
      }
    }
  }
}

// The words of this lane's sentence (raw bytes gtext[beg, beg + len)) -> ids in slot[0, n), forward order.
// Returns n >= 0, or -1: the sentence is not for this pass (nothing usable was written).
// MODE (kWm*): the call-local memo of `a` (dyn_*) is filled (collect) / consulted (dyn); -2: "try again in the second round".
// `rs` of uni_word_lane: (collect) out -- where the lane stood when it met its first missing word: {position, ids
// written, bound}; (dyn) in -- where to take the sentence up again: the ids before that point are in `slot` already.
struct WordResume { int p; int n; float B; };
// H16: the ids leave as 16-bit values (vocabularies of up to 65536 pieces): a burst of 8 ids is ONE 16-byte store, and
// CompactKernel reads half the bytes.  `slot` is then an array of uint16_t at the same (16-byte aligned) address.
template <bool DP, int MODE, bool H16>
SPMX_DEVICE int uni_word_lane(const EncodeArgs &a, const uint8_t *gtext, uint64_t beg, int len, int32_t *slot, int cap,
                              const WordLds &T, bool active_in, int *n_steps, WordResume *rs) {
  const SpmxDev &d = a.dev;
  const U4 *__restrict__ memo16 = d.umemo16;
  const U4 *__restrict__ memo32 = d.umemo;
  const uint32_t m16 = d.umemo16_mask, m32 = d.umemo_mask;
  const uint8_t *text = gtext + beg;
  int32_t *stage = T.stage;
  bool active = active_in && len > 0;
  bool bad = false;
  bool again = false;                              // (collect) every word this lane could not take is in the call-local memo now
  int p = 0, n = 0, steps = 0;
  float B = 0.f;                                   // DP: best_path_score at the start of the current word; else a bound of its magnitude
  if (MODE == kWmDyn && active) {                  // take the sentence up where the first round left it
    p = rs->p; n = rs->n; B = rs->B;
    for (int k = n & ~7; k < n; ++k)                 // the incomplete group goes back into the staging column
      T.stage[(k & 7) << 6] = H16 ? static_cast<int32_t>(reinterpret_cast<const uint16_t *>(slot)[k]) : slot[k];
    if (p >= len) active = false;                  // (cannot happen: the first round stopped AT a word)
  }
  int n_dp = 0;
  // a word waiting for the DP (second pass): its length; the lane goes on once the wave has run uni_word_dp
  bool stalled = false, stall_last = false;
  int stall_L = 0;
  bool prev_unk = false;                           // the last piece emitted was unknown (a run of them is ONE id, :609-613)
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool keep_ws = (d.flags & kNfRemoveExtraWs) == 0;
  Q4U w{0, 0, 0, 0};
  int wvalid = 16;                                 // bytes of `w` that are text (what a shift brought in behind them is zero)
  if (active) w = *reinterpret_cast<const Q4U *>(text + p);
#if SPMX_EXP & (8 | 32)
  uint32_t exp_acc = 0;
#endif
  auto put = [&](uint32_t id) __attribute__((always_inline)) {
    stage[(n & 7) << 6] = static_cast<int32_t>(id);
    ++n;
    if (H16 && (n & 7) == 0 && !(SPMX_EXP & 16)) {
      const uint32_t s0 = static_cast<uint32_t>(stage[0]), s1 = static_cast<uint32_t>(stage[1 << 6]), s2 = static_cast<uint32_t>(stage[2 << 6]),
                     s3 = static_cast<uint32_t>(stage[3 << 6]), s4 = static_cast<uint32_t>(stage[4 << 6]), s5 = static_cast<uint32_t>(stage[5 << 6]),
                     s6 = static_cast<uint32_t>(stage[6 << 6]), s7 = static_cast<uint32_t>(stage[7 << 6]);
      *reinterpret_cast<Q4 *>(reinterpret_cast<uint16_t *>(slot) + (n - 8)) = Q4{s0 | s1 << 16, s2 | s3 << 16, s4 | s5 << 16, s6 | s7 << 16};
    }
    if (!H16 && (n & 7) == 0 && !(SPMX_EXP & 16)) {   // (16: experiment build without the id bursts)
      int32_t *q = slot + (n - 8);
      *reinterpret_cast<Q4 *>(q) = Q4{static_cast<uint32_t>(stage[0]), static_cast<uint32_t>(stage[1 << 6]),
                                      static_cast<uint32_t>(stage[2 << 6]), static_cast<uint32_t>(stage[3 << 6])};
      *reinterpret_cast<Q4 *>(q + 4) = Q4{static_cast<uint32_t>(stage[4 << 6]), static_cast<uint32_t>(stage[5 << 6]),
                                          static_cast<uint32_t>(stage[6 << 6]), static_cast<uint32_t>(stage[7 << 6])};
    }
  };
  while (wv::any(active)) {
    if (DP && !wv::any(active && !stalled)) {
      // ---- no lane can go on: every waiting lane segments its word (exact: from the true B) ----
      uni_word_dp(d, T, stall_L, B, active && stalled);
      if (active && stalled) {
        const int nn = stall_L + 1;
        // backtrack (:1010-1018): count first, then piece by piece in forward order, with the id post-processing of
        // sentencepiece_processor.cc:581-613 -- a run of unknown pieces is one id (the run may continue from the previous
        // word: prev_unk), or every unknown character's bytes under byte fallback
        int e = nn, cnt = 0, need = 0;
        bool broken = false;
        while (e > 0) {
          const uint32_t bw = T.dp_bp[e << 6];
          const int bl = static_cast<int>((bw >> kBwLenShift) & kBwLenMask);
          if (bw == 0u || bl == 0 || bl > e) { broken = true; break; }
          ++cnt;
          need += (bw & kBwUnk) ? (bf ? (T.dp_bytes[e - bl] == kSpByte ? 3 : 1) : 1) : 1;
          e -= bl;
        }
        if (broken || n + need > cap) { bad = true; active = false; }
        else {
          B = T.dp_best[nn << 6];
          for (int k = 0; k < cnt; ++k) {            // (cnt <= 17: piece k is found by walking back from the end again)
            int e2 = nn;
            for (int j = cnt - 1; j > k; --j) e2 -= static_cast<int>((T.dp_bp[e2 << 6] >> kBwLenShift) & kBwLenMask);
            const uint32_t bw = T.dp_bp[e2 << 6];
            if (bw & kBwUnk) {
              const uint32_t ch = T.dp_bytes[e2 - 1];
              if (bf) {
                if (ch == kSpByte) { put(static_cast<uint32_t>(d.byte_ids[0xE2])); put(static_cast<uint32_t>(d.byte_ids[0x96])); put(static_cast<uint32_t>(d.byte_ids[0x81])); }
                else put(static_cast<uint32_t>(d.byte_ids[ch]));
              } else if (!prev_unk) {
                put(static_cast<uint32_t>(d.unk_id));
              }
              prev_unk = true;
            } else {
              put(bw & kBwIdMask);
              prev_unk = false;
            }
          }
        }
        stalled = false;
        if (stall_last) active = false;              // it was the sentence's last word
      }
      continue;
    }
    ++steps;
    const bool run = active && !stalled;
    // ---- the word that starts at p: its length = the distance to the next 0x20 (or to the end of the sentence) ----
    const uint32_t z0 = space_flags(w.x), z1 = space_flags(w.y), z2 = space_flags(w.z), z3 = space_flags(w.w);
    const uint64_t zlo = static_cast<uint64_t>(z0) | static_cast<uint64_t>(z1) << 32, zhi = static_cast<uint64_t>(z2) | static_cast<uint64_t>(z3) << 32;
    const int fa = wv::ffs64(zlo);
    const int fb = wv::ffs64(zhi);
    int L = fa ? (fa - 1) >> 3 : (fb ? 8 + ((fb - 1) >> 3) : 16);      // 16: no space among the 16 bytes
    const int rem = len - p;
    if (L > rem) L = rem;
    // ---- where the next word starts; its text is made ready now and used in the next iteration.  The 16 bytes the lane
    // holds usually hold the next word too (a word and its space are 5.7 bytes on average): when they hold ALL of it --
    // another 0x20 behind this word's, or the sentence's end -- the window is shifted down in registers instead of asked
    // for again; a gather costs by the lane (section 4.0 of DESIGN.md), and this takes more than half the lanes out of it. ----
    const int pn = p + L + 1;
    const bool more = run && pn < len;
    const int held = wvalid - (L + 1);                // bytes of the window behind this word and its space
    const bool second = fa ? ((zlo & (zlo - 1ull)) != 0ull || zhi != 0ull) : ((zhi & (zhi - 1ull)) != 0ull);
    const bool reuse = more && held > 0 && (second || pn + held >= len);
    Q4U wn = w;
    int wvn = wvalid;
    if (reuse) {
      const uint32_t sh = static_cast<uint32_t>(L + 1);        // 1 .. 15 bytes
      uint32_t a0 = w.x, a1 = w.y, a2 = w.z, a3 = w.w;
      if (sh & 8u) { a0 = a2; a1 = a3; a2 = 0u; a3 = 0u; }
      if (sh & 4u) { a0 = a1; a1 = a2; a2 = a3; a3 = 0u; }
      wn = Q4U{wv::alignbyte(a1, a0, sh), wv::alignbyte(a2, a1, sh), wv::alignbyte(a3, a2, sh), wv::alignbyte(0u, a3, sh)};
      wvn = held;
    }
    if (more && !reuse) { wn = *reinterpret_cast<const Q4U *>(text + pn); wvn = 16; }
#if SPMX_EXP & 8      // (experiment build: what ONE more 64-lane text gather per iteration costs)
    if (more) { const Q4U x = *reinterpret_cast<const Q4U *>(text + (pn + 160 < len ? pn + 160 : pn)); exp_acc ^= x.x ^ x.y ^ x.z ^ x.w; }
#endif
    U4 ent{0, 0, 0, 0xFFFFFFFFu};                    // the entry taken: umemo16 format, or {id0, id1, bound, bmax} of umemo
    bool hit16 = false, hit32 = false;
    uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    uint32_t wx4 = 0x20202020u;                      // bytes 16 .. 19 of the word's text (only words of 16 bytes and more read them)
    const bool word = run && L > 0;                  // L == 0: a space (leading, doubled): skip it
    bool shortw = word && L <= 12;
    if (shortw) {
      const Q4 mk = T.masks[L];
      k0 = key_dword(w.x, mk.x); k1 = key_dword(w.y, mk.y); k2 = key_dword(w.z, mk.z);
    }
    const uint32_t h = HashWordKey(k0, k1, k2, 0u);
    // which form of 16-byte entry a word of L bytes has (dev.h umemo16): up to 10 bytes the two-piece form -- the third
    // dword's high half is the second id, not key bytes
    const uint32_t zmask = L <= 10 ? 0xFFFFu : 0xFFFFFFFFu, form = L <= 10 ? kMemo16TwoPiece : 0u;
    auto same16 = [&](const U4 &e) __attribute__((always_inline)) -> bool {
      return e.x == k0 && e.y == k1 && ((e.z ^ k2) & zmask) == 0u && (e.w & kMemo16TwoPiece) == form && e.w != 0xFFFFFFFFu;
    };
    {   // the likeliest words: LDS
      const U4 e = T.hot[h & (kWordHotSlots - 1u)];
      hit16 = shortw && same16(e);
      if (hit16) ent = e;
    }
#if SPMX_EXP & 32     // (experiment build: one more memo16 gather per iteration, every lane with a short word)
    { const U4 x = memo16[shortw ? ((h * 0x9E3779B1u) >> 7) & m16 : 0u]; exp_acc ^= x.x ^ x.w; }
#endif
    bool look = shortw && !hit16;
    if (wv::any(look)) {                             // the rest of the one-piece words of up to 12 bytes: one probe, HBM / L2
      uint32_t sl = h & m16;
      U4 e = memo16[look ? sl : 0u];
      hit16 = hit16 || (look && same16(e));
      bool walk = look && !hit16 && e.w != 0xFFFFFFFFu;
      while (wv::any(walk)) {                        // a collision: walk on (rare: the table is half empty)
        if (walk) {
          sl = (sl + 1u) & m16;
          e = memo16[sl];
          hit16 = same16(e);
          walk = !hit16 && e.w != 0xFFFFFFFFu;
        }
      }
      if (look && hit16) ent = e;
    }
    bool lng = false;                                // a word of more than 16 bytes
    if (wv::any(word && !hit16)) {
      // ---- the other words of the memo (two pieces, 13 .. 16 bytes): 32-byte entries ----
      const bool need = word && !hit16;
      int L2 = L;
      if (need && L == 16 && rem > 16) {             // the 17th byte decides whether the word ends here
        wx4 = reinterpret_cast<const U1U *>(text + p + 16)->x;
        if ((wx4 & 0xFFu) != 0x20u) lng = true;
      }
      const bool probe = need && !lng;
      const Q4 mk = T.masks[probe ? L2 : 0];
      k0 = key_dword(w.x, mk.x); k1 = key_dword(w.y, mk.y); k2 = key_dword(w.z, mk.z); k3 = key_dword(w.w, mk.w);
      uint32_t sl = HashWordKey(k0, k1, k2, k3) & m32;
      U4 e0 = memo32[2u * (probe ? sl : 0u)], e1 = memo32[2u * (probe ? sl : 0u) + 1u];
      hit32 = probe && e0.x == k0 && e0.y == k1 && e0.z == k2 && e0.w == k3 && e1.x != 0xFFFFFFFFu;
      bool walk = probe && !hit32 && e1.x != 0xFFFFFFFFu;
      while (wv::any(walk)) {
        if (walk) {
          sl = (sl + 1u) & m32;
          e0 = memo32[2u * sl];
          e1 = memo32[2u * sl + 1u];
          hit32 = e0.x == k0 && e0.y == k1 && e0.z == k2 && e0.w == k3 && e1.x != 0xFFFFFFFFu;
          walk = !hit32 && e1.x != 0xFFFFFFFFu;
        }
      }
      if (hit32) ent = e1;
    }
    // ---- (second round) the call-local memo: words collected by the first round, segmented by word_resolve_block ----
    bool hitd = false;
    uint32_t dn = 0;
    U4 dia{0, 0, 0, 0}, dib{0, 0, 0, 0};            // the word's ids
    if (MODE == kWmDyn && wv::any(word && !hit16 && !hit32 && !lng)) {
      if (word && !hit16 && !hit32 && !lng) {
        const unsigned long long tag = DynTag(k0, k1, k2, k3);
        uint32_t sl = static_cast<uint32_t>(tag >> 32) & a.dyn_mask;
        for (uint32_t t = 0; t < kDynProbes; ++t) {
          // the slot's tag AND its 64-byte entry are asked for together: one round trip per probe (a lane's wait is the
          // whole wavefront's; round 3 waited for the tag, then the key, then the ids: uni32k_w16's second round 3.1 ms)
          const unsigned long long g = a.dyn_tag[sl];
          const U4 e0 = a.dyn_ent[4u * sl], e1 = a.dyn_ent[4u * sl + 1u], e2 = a.dyn_ent[4u * sl + 2u], e3 = a.dyn_ent[4u * sl + 3u];
          if (g == 0ull) break;
          if (g == tag) {
            if (e0.x == k0 && e0.y == k1 && e0.z == k2 && e0.w == k3 && e1.x == 1u) { hitd = true; dn = e1.y; ent = U4{0, 0, e1.z, e1.w}; dia = e2; dib = e3; }
            break;                                   // (same hash, other bytes or an unusable word: a miss)
          }
          sl = (sl + 1u) & a.dyn_mask;
        }
      }
    }
    // ---- take the entry while its margin holds ----
    const uint32_t id0 = hit16 ? (ent.w & 0xFFFFu) : ent.x;
    const uint32_t id1 = hit32 ? ent.y : ((hit16 && (ent.w & kMemo16TwoPiece) && (ent.z >> 16) != 0xFFFFu) ? ent.z >> 16 : 0xFFFFFFFFu);
    // valid while |B| < lim: 2^e (16-byte entry: the power of two below bmax) or bmax itself
    const float lim = hit16 ? wv::bits_to_float((((ent.w >> 16) & 0x7Fu) + 127u) << 23) : wv::bits_to_float(ent.w);
    const bool hit = hit16 || hit32 || hitd;
    // (a collecting lane that has already given its sentence up only scouts for more words: no margin to check)
    const bool ok = hit && ((MODE == kWmCollect && bad) || fabsf(B) < lim);
    if (MODE == kWmCollect && word && !hit && !lng) {
      // ---- a word the memo lacks: into the call-local memo (once per word per call), if it is plain ----
      const Q4 mk = T.masks[L];
      const uint32_t pad = 0x41414141u;
      const uint32_t a0 = (w.x & mk.x) | (pad & ~mk.x), a1 = (w.y & mk.y) | (pad & ~mk.y), a2 = (w.z & mk.z) | (pad & ~mk.z),
                     a3 = (w.w & mk.w) | (pad & ~mk.w);
      auto plain = [](uint32_t x) -> bool {
        return (((x + 0x01010101u) | x) & 0x80808080u) == 0u && (((x - 0x21212121u) & ~x) & 0x80808080u) == 0u;
      };
      bool kept = false;
      if (plain(a0) && plain(a1) && plain(a2) && plain(a3)) {
        const unsigned long long tag = DynTag(k0, k1, k2, k3);
        uint32_t sl = static_cast<uint32_t>(tag >> 32) & a.dyn_mask;
        for (uint32_t t = 0; t < kDynProbes && !kept; ++t) {
          // (a look first: most occurrences of a word find it entered already, and a load is cheaper than an atomic)
          unsigned long long g = wv::atomic_load64(&a.dyn_tag[sl]);
          if (g == 0ull) g = wv::atomic_cas(&a.dyn_tag[sl], 0ull, tag);
          if (g == 0ull) {                           // ours: the word's bytes, and a place in the list of words to segment
            const uint32_t at = wv::atomic_add(a.dyn_count, 1u);
            a.dyn_ent[4u * sl] = U4{k0, k1, k2, k3};
            a.dyn_ent[4u * sl + 1u] = U4{at < a.dyn_cap ? 0u : 2u, 0u, 0u, 0u};     // (2: no room on the list: never usable)
            if (at < a.dyn_cap) { a.dyn_list[at] = sl; kept = true; }
            break;
          }
          if (g == tag) { kept = true; break; }      // another lane has entered it
          sl = (sl + 1u) & a.dyn_mask;
        }
      }
      if (!bad) { rs->p = p; rs->n = n; rs->B = B; }   // (the first missing word: the second round starts here)
      bad = true;
      if (kept) again = true; else { again = false; active = false; }
    } else if (word && !ok) {
      if (MODE == kWmCollect) again = false;         // (a margin that does not hold, a word of more than 16 bytes: no second round)
      bool dp_ok = false;
      if (DP && !lng && n_dp < kWordDpMax) {
        // the lane segments the word itself if it is plain: L bytes 0x21 .. 0x7E
        const Q4 mk = T.masks[L];
        const uint32_t pad = 0x41414141u;
        const uint32_t a0 = (w.x & mk.x) | (pad & ~mk.x), a1 = (w.y & mk.y) | (pad & ~mk.y), a2 = (w.z & mk.z) | (pad & ~mk.z),
                       a3 = (w.w & mk.w) | (pad & ~mk.w);
        auto plain = [](uint32_t x) -> bool {
          return (((x + 0x01010101u) | x) & 0x80808080u) == 0u && (((x - 0x21212121u) & ~x) & 0x80808080u) == 0u;
        };
        dp_ok = plain(a0) && plain(a1) && plain(a2) && plain(a3);
        if (dp_ok) {
          T.dp_bytes[0] = static_cast<uint8_t>(kSpByte);
          for (int i = 0; i < 4; ++i) {
            T.dp_bytes[1 + i] = static_cast<uint8_t>(a0 >> (8 * i));
            T.dp_bytes[5 + i] = static_cast<uint8_t>(a1 >> (8 * i));
            T.dp_bytes[9 + i] = static_cast<uint8_t>(a2 >> (8 * i));
            T.dp_bytes[13 + i] = static_cast<uint8_t>(a3 >> (8 * i));
          }
          stalled = true;
          stall_L = L;
          stall_last = !more;
          ++n_dp;
        }
      }
      if (!dp_ok) { bad = true; active = false; }
    }
    if (ok && hitd) {
      // a word of the call-local memo: dn = count | kDynFirstUnk (begins with an unknown piece) | kDynLastUnk | kDynWide
      const uint32_t cntd = dn & 0xFFu;
      if (n + static_cast<int>(cntd) > cap) { bad = true; active = false; }
      else if (cntd == 0u) {
        // a word Normalize drops altogether: the words around it are neighbours in the normalized text; an unknown-piece
        // run that would have to continue across it is left to the general kernels (no such run: nothing to do)
        if (prev_unk) { bad = true; active = false; }
      } else {
        B += wv::bits_to_float(ent.z);
        const uint32_t di[kDynMaxIds] = {dia.x, dia.y, dia.z, dia.w, dib.x, dib.y, dib.z, dib.w};
        const uint32_t k0d = (prev_unk && (dn & kDynFirstUnk)) ? 1u : 0u;                       // (:609-613 the run goes on)
        if (dn & kDynWide) {                         // 9 .. 16 pieces, two to a dword (small vocabularies split rare words finely)
          for (uint32_t k = k0d; k < cntd; ++k) put((di[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu);
        } else {
          for (uint32_t k = k0d; k < cntd; ++k) put(di[k]);
        }
        prev_unk = (dn & kDynLastUnk) != 0u;
      }
    } else if (ok && !(MODE == kWmCollect && bad)) {
      const bool two = id1 != 0xFFFFFFFFu;
      if (n + (two ? 2 : 1) > cap) { bad = true; active = false; }
      else {
        if (DP) {
          B = static_cast<float>(static_cast<double>(d.pscore[id0]) + static_cast<double>(B));   // :982-989 along the path
          if (two) B = static_cast<float>(static_cast<double>(d.pscore[id1]) + static_cast<double>(B));
        } else {
          // a bound of |B|: ceil(|score|) + 1 per piece (the + 1 covers the roundings of the sum)
          B += hit16 ? static_cast<float>(ent.w >> 24) : wv::bits_to_float(ent.z);
        }
        put(id0);
        if (two) put(id1);
        prev_unk = false;
      }
    }
    // a model that KEEPS extra whitespace (remove_extra_whitespaces off, src/normalizer.cc:88-110,160-176): a leading, a
    // doubled or a trailing space is a space symbol of its own in the normalized text, next to other space symbols --
    // pieces made of space symbols only may match there.  Not a word: the sentence takes the general kernels.
    if (keep_ws && run && (L == 0 || pn == len)) { bad = true; again = false; active = false; }
    if (run && !more && !(DP && stalled)) active = false;     // the sentence is done
    if (run) { p = pn; w = wn; wvalid = wvn; }
  }
  *n_steps = steps;
#if SPMX_EXP & (8 | 32)
  if (exp_acc == 0x9E3779B1u) ++steps, *n_steps = steps;     // (keeps the experiment's loads alive)
#endif
  auto flush_tail = [&]() __attribute__((always_inline)) {
    for (int k = n & ~7; k < n; ++k) {
      if (H16) reinterpret_cast<uint16_t *>(slot)[k] = static_cast<uint16_t>(stage[(k & 7) << 6]);
      else slot[k] = stage[(k & 7) << 6];
    }
  };
  if (bad && MODE == kWmCollect && again) {
    flush_tail();                                  // what is staged belongs to the ids the second round keeps
    return -2;
  }
  if (bad) return -1;
  if (active_in && len > 0) flush_tail();          // the last, incomplete group
  return n;
}

// Persistent body of the word kernels: tiles of up to 64 sentences from the launch's queue (kernels_stream.h
// next_tile), one sentence per lane.  What a lane cannot take goes to the leftover list of its class.
template <bool DP, int MODE, bool H16>
SPMX_DEVICE void encode_word_block_as(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  Q4 *masks = reinterpret_cast<Q4 *>(smem);
  U4 *hot = reinterpret_cast<U4 *>(smem + kWordMaskBytes);
  unsigned char *mine_lds = smem + kWordLdsShared + static_cast<uint32_t>(wv::wave_in_block()) * WordLdsPerWave(DP);
  WordLds T;
  T.masks = masks;
  T.hot = hot;
  T.stage = reinterpret_cast<int32_t *>(mine_lds) + lane;
  T.dp_best = reinterpret_cast<float *>(mine_lds + 64u * kWordStage * 4u) + lane;
  T.dp_bp = reinterpret_cast<uint32_t *>(mine_lds + 64u * kWordStage * 4u + 64u * kWordDpPos * 4u) + lane;
  T.dp_bytes = mine_lds + 64u * kWordStage * 4u + 64u * kWordDpPos * 8u + static_cast<uint32_t>(lane) * 20u;
  {   // shared read-only tables (every wave writes the same values: no workgroup barrier)
    if (lane < 18) {
      const uint32_t L = static_cast<uint32_t>(lane);
      auto m = [&](uint32_t i) -> uint32_t { return L >= 4u * i + 4u ? 0xFFFFFFFFu : (L <= 4u * i ? 0u : (1u << (8u * (L - 4u * i))) - 1u); };
      masks[lane] = Q4{m(0), m(1), m(2), m(3)};
    }
    for (uint32_t k = static_cast<uint32_t>(lane); k < kWordHotSlots; k += 64u) hot[k] = d.uhot[k];
    wv::sync();
  }
  const int n_extra = d.n_prefix + d.n_suffix;
  const uint64_t text_end = a.offs[a.n];
  WaveCounters tc;
  for (;;) {
    uint32_t c = 0, first = 0, ucnt = 0, got = 0;
    if (lane == 0) got = next_tile(a, &c, &first, &ucnt) ? 1u : 0u;
    got = wv::shfl(got, 0);
    if (!got) break;
    c = wv::shfl(c, 0); first = wv::shfl(first, 0); ucnt = wv::shfl(ucnt, 0);
    const unsigned long long c0 = wv::clock();
    const uint32_t *list = a.lists + static_cast<uint64_t>(c) * a.n;
    const bool have = static_cast<uint32_t>(lane) < ucnt;
    uint32_t sid = 0;
    uint64_t beg = 0, l64 = 0;
    if (have) {
      sid = list[first + static_cast<uint32_t>(lane)];
      beg = a.offs[sid];
      l64 = a.offs[sid + 1] - beg;
    }
    // not for this kernel: beyond the int range of the lane's counters, or text that ends within the over-read of the
    // last sentences of the buffer
    // (a class marked `general` passes through: documents belong to the wave-cooperative form, kernels_uniwave.h)
    const bool mine = have && !a.cls[c].general && l64 < (1ull << 30) && beg + l64 + 20u <= text_end;
    const int len = mine ? static_cast<int>(l64) : 0;
    // ---- a slot of cap ids in the arena (at most one id per byte of the normalized form: the bytes + 1) ----
    const int cap = mine ? len + 1 : 0;
    // (second round: the sentence keeps the slot -- and the ids -- the first round gave it)
    const int room = (mine && MODE != kWmDyn) ? (cap + n_extra + 3 + 3) & ~3 : 0;
    int total = 0;
    const int excl = wave_excl_scan(room, lane, &total);
    unsigned long long base = 0;
    if (lane == 0 && total > 0) base = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(total + 3));
    base = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(base >> 32), 0)) << 32) |
           wv::shfl(static_cast<uint32_t>(base), 0);
    const bool overflow = base + static_cast<unsigned long long>(total + 3) > a.arena_cap;
    if (overflow && lane == 0) wv::atomic_or(a.status, kStArenaOverflow);
    base = (base + 3ull) & ~3ull;
    const int at = excl + d.n_prefix;
    const int shift = (4 - (at & 3)) & 3;
    int32_t *slot = a.arena + base + static_cast<unsigned long long>(excl + shift) + d.n_prefix;
    WordResume rs{0, 0, 0.f};
    if (MODE == kWmDyn && mine) {
      slot = a.arena + a.tmp_off[sid] + d.n_prefix;
      const U4 r = a.resume[sid];
      rs.p = static_cast<int>(r.x); rs.n = static_cast<int>(r.y); rs.B = wv::bits_to_float(r.z);
    }
    int steps = 0;
    int n = uni_word_lane<DP, MODE, H16>(a, a.text, beg, len, slot, cap, T, mine && !overflow, &steps, &rs);
    const unsigned long long c1 = wv::clock();
    if (overflow) n = -1;
    const bool done = mine && n >= 0;
    if (done) {
      if (H16) {
        // (16-bit ids: the extra ids sit right before / behind the body in 16-bit units too; the sentence's place is told
        // to CompactKernel in 16-bit units from the arena's start, flagged by kTmpOffHalf)
        uint16_t *s16 = reinterpret_cast<uint16_t *>(slot);
        for (int x = 0; x < d.n_prefix; ++x) s16[x - d.n_prefix] = static_cast<uint16_t>(d.prefix_ids[x]);
        for (int x = 0; x < d.n_suffix; ++x) s16[n + x] = static_cast<uint16_t>(d.suffix_ids[x]);
        a.tmp_off[sid] = kTmpOffHalf | (2ull * static_cast<unsigned long long>(slot - a.arena) - static_cast<unsigned long long>(d.n_prefix));
      } else {
        for (int x = 0; x < d.n_prefix; ++x) slot[x - d.n_prefix] = d.prefix_ids[x];
        for (int x = 0; x < d.n_suffix; ++x) slot[n + x] = d.suffix_ids[x];
        a.tmp_off[sid] = static_cast<unsigned long long>(slot - d.n_prefix - a.arena);
      }
      a.counts[sid] = static_cast<uint32_t>(n + n_extra);
    }
    const bool left = have && !done;
    if (left) a.counts[sid] = 0u;                    // (until a later pass has had it)
    if (MODE == kWmCollect) {
      // the second round takes what is only short of words the call-local memo now holds; the rest goes straight to
      // the general launches
      const bool again = left && n == -2;
      const bool gone = left && !again;
      if (again) {                                   // where the second round takes the sentence up again
        a.resume[sid] = U4{static_cast<uint32_t>(rs.p), static_cast<uint32_t>(rs.n), wv::float_to_bits(rs.B), 0u};
        a.tmp_off[sid] = static_cast<unsigned long long>(slot - d.n_prefix - a.arena);
      }
      append_lanes(wv::ballot(again), again, sid, a.left_lists + static_cast<uint64_t>(c) * a.n, &a.left_counts[c], lane);
      append_lanes(wv::ballot(gone), gone, sid, a.left2_lists + static_cast<uint64_t>(c) * a.n, &a.left2_counts[c], lane);
    } else {
      append_lanes(wv::ballot(left), left, sid, a.left_lists + static_cast<uint64_t>(c) * a.n, &a.left_counts[c], lane);
    }
    if (done) { ++tc.n_sent; tc.n_raw += static_cast<unsigned long long>(len); tc.n_ids += static_cast<unsigned long long>(n + n_extra); }
    tc.n_trips += static_cast<unsigned long long>(steps);
    tc.cyc[2] += c1 - c0;
    tc.cyc[3] += wv::clock() - c1;
  }
  if (a.stats) {
    unsigned long long v[3] = {tc.n_sent, tc.n_raw, tc.n_ids};
    for (int k = 0; k < 3; ++k) {
      uint64_t tot = 0;
      wave_excl_scan64(v[k], lane, &tot);
      if (lane == 0 && tot) wv::atomic_add(&a.stats[k], static_cast<unsigned long long>(tot));
    }
    if (lane == 0) {
      for (int k = 0; k < 4; ++k) wv::atomic_add(&a.stats[3 + k], tc.cyc[k]);
      wv::atomic_add(&a.stats[7], tc.n_trips);
    }
  }
}

template <bool DP, int MODE>
SPMX_DEVICE void encode_word_block(const EncodeArgs &a, unsigned char *smem) {
  if (a.ids16) encode_word_block_as<DP, MODE, true>(a, smem);      // (wave-uniform: one form per launch)
  else encode_word_block_as<DP, MODE, false>(a, smem);
}

// ---- the call-local memo's words, segmented once each: one LANE per collected word ------------------------------------
struct ResolveArgs {
  SpmxDev dev;
  U4 *dyn_ent;
  const uint32_t *dyn_list;
  const uint32_t *dyn_count;
  uint32_t dyn_cap;
  uint32_t unsafe;              // TEST SEAM (SPMX_WORDMEMO_UNSAFE): no margin, as in tables.cc
};
constexpr uint32_t kResolvePos = 18;
SPMX_HD inline uint32_t ResolveLdsBytes() { return 64u * (kResolvePos * 12u + 20u); }

// unigram: EncodeOptimized (src/unigram_model.cc:957-1018) of the word from score 0, in float, keeping the second best
// candidate of every position: the margin analysis of tables.cc BuildWordMemo on the device.  Every value here is
// within n * ulp(wmag) / 2 of its exact counterpart (one rounding per piece), so the exact lead of the best candidate at
// a position is at least (best - second) - n * ulp(wmag); that is what the margin is computed from.
SPMX_DEVICE void resolve_unigram_lane(const ResolveArgs &a, uint32_t slot, float *best, float *second, uint32_t *bp, uint8_t *wb, bool active) {
  const SpmxDev &d = a.dev;
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  int L = 0;
  if (active) {
    const U4 k = a.dyn_ent[4u * slot];
    const uint32_t kw[4] = {k.x, k.y, k.z, k.w};
    wb[0] = static_cast<uint8_t>(kSpByte);
    for (int i = 0; i < 16; ++i) {
      const uint32_t b = (kw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      wb[1 + i] = static_cast<uint8_t>(b);
      if (b != 0x20u && L == i) L = i + 1;               // (the key's padding: key_dword)
    }
  }
  const int n = L + 1;
  const float kNone = -3.0e38f;
  if (active) {
    for (int i = 0; i <= n; ++i) { bp[i << 6] = 0u; best[i << 6] = kNone; second[i << 6] = kNone; }
    best[0] = 0.f;
  }
  float wmag = 0.f;
  int s = 0, dep = 0;
  uint32_t node = root;
  float bs = 0.f;
  bool single = false;
  bool run = active;
  auto relax = [&](int e, float cand, uint32_t word) __attribute__((always_inline)) {
    if (fabsf(cand) > wmag) wmag = fabsf(cand);
    if (bp[e << 6] == 0u || cand > best[e << 6]) { second[e << 6] = bp[e << 6] == 0u ? kNone : best[e << 6]; best[e << 6] = cand; bp[e << 6] = word; }
    else if (cand > second[e << 6]) second[e << 6] = cand;
  };
  while (wv::any(run)) {
    if (run) {
      const uint32_t c = wb[s + dep];
      const U4 u = ptrie[node ^ c];
      bool match = (u.x & 0x1FFu) == (0x100u | c);
      if (match) {
        ++dep;
        node = u.x >> kDatBaseShiftDev;
        if ((u.x & kDatTerminalDev) && !(u.y & kPtUnused)) {
          relax(s + dep, wv::bits_to_float(u.z) + bs, (u.y & kBwIdMask) | (static_cast<uint32_t>(dep) << kBwLenShift));
          if (dep == 1) single = true;
        }
        if (s + dep >= n) match = false;
      }
      if (!match) {
        if (!single) relax(s + 1, d.unk_score + bs, (1u << kBwLenShift) | kBwUnk);
        ++s;
        if (s >= n) run = false;
        else { bs = best[s << 6]; node = root; dep = 0; single = false; }
      }
    }
  }
  if (!active) return;
  // the best path, its smallest lead, its ids
  // (ids in backtrack order = last piece first; an unknown piece is unk_id, a run of them ONE id, or under byte fallback
  // the bytes of every unknown character -- sentencepiece_processor.cc:581-613; whether the word begins / ends with an
  // unknown piece is kept, so that a run can continue across words)
  uint32_t ids[kDynMaxWide];
  int cnt = 0;
  uint32_t id_or = 0;
  float gap = 3.0e38f, bound = 0.f;
  bool good = true;
  const bool bf = (d.flags & kNfByteFallback) != 0;
  bool last_unk = false, first_unk = false, right_unk = false;
  auto push = [&](uint32_t id) __attribute__((always_inline)) { if (cnt < static_cast<int>(kDynMaxWide)) { ids[cnt++] = id; id_or |= id; } else good = false; };
  for (int e = n; e > 0 && good;) {
    const uint32_t bw = bp[e << 6];
    const int bl = static_cast<int>((bw >> kBwLenShift) & kBwLenMask);
    if (bw == 0u || bl == 0 || bl > e) { good = false; break; }
    if (second[e << 6] > kNone) { const float g = best[e << 6] - second[e << 6]; if (g < gap) gap = g; }
    const bool unk = (bw & kBwUnk) != 0;
    if (e == n) last_unk = unk;
    if (e - bl == 0) first_unk = unk;
    if (unk) {
      bound += ceilf(fabsf(d.unk_score)) + 1.f;
      if (bf) {
        const uint32_t ch = wb[e - 1];
        if (ch == kSpByte) { push(static_cast<uint32_t>(d.byte_ids[0x81])); push(static_cast<uint32_t>(d.byte_ids[0x96])); push(static_cast<uint32_t>(d.byte_ids[0xE2])); }
        else push(static_cast<uint32_t>(d.byte_ids[ch]));
      } else if (!right_unk) {
        push(static_cast<uint32_t>(d.unk_id));
      }
      right_unk = true;
    } else {
      push(bw & kBwIdMask);
      bound += ceilf(fabsf(d.pscore[bw & kBwIdMask])) + 1.f;
      right_unk = false;
    }
    e -= bl;
  }
  if (bf) { first_unk = false; last_unk = false; }     // (no runs under byte fallback)
  float bmax = 0.f;
  if (good) {
    // ulp of a float of magnitude wmag; the exact lead is at least gap - n * ulp (see above; twice the bound needed)
    const uint32_t we = (wv::float_to_bits(wmag) >> 23) & 0xFFu;
    const float ulp_w = we > 23u ? wv::bits_to_float((we - 23u) << 23) : 0.f;
    const float g2 = gap >= 3.0e38f ? gap : gap - 2.f * static_cast<float>(n) * ulp_w;
    if (!(g2 > 0.f)) good = false;
    else if (a.unsafe || g2 >= 3.0e38f) bmax = 3.0e38f;
    else {
      const float thr = g2 / (4.f * static_cast<float>(n) + 4.f);
      const int te = static_cast<int>((wv::float_to_bits(thr) >> 23) & 0xFFu) - 127;      // floor(log2 thr)
      const int k = te + 23;                                  // the largest k with 2^(k - 23) <= thr
      if (k + 1 > 126) bmax = 3.0e38f;
      else if (k + 1 < -120) good = false;
      else {
        const float lim = wv::bits_to_float(static_cast<uint32_t>(k + 1 + 127) << 23);
        bmax = (lim - wmag * 1.000001f) * 0.999999f;
        if (!(bmax > 0.f)) good = false;
      }
    }
  }
  const bool wide = cnt > static_cast<int>(kDynMaxIds);
  if (wide && id_or > 0xFFFFu) good = false;           // (more than 8 pieces AND a vocabulary beyond 65536: not kept)
  if (good) {
    uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < cnt; ++k) {
      const uint32_t id = ids[cnt - 1 - k];
      if (wide) o[k >> 1] |= id << (16 * (k & 1)); else o[k] = id;
    }
    a.dyn_ent[4u * slot + 2u] = U4{o[0], o[1], o[2], o[3]};
    a.dyn_ent[4u * slot + 3u] = U4{o[4], o[5], o[6], o[7]};
    a.dyn_ent[4u * slot + 1u] = U4{1u, static_cast<uint32_t>(cnt) | (first_unk ? kDynFirstUnk : 0u) | (last_unk ? kDynLastUnk : 0u) | (wide ? kDynWide : 0u),
                                  wv::float_to_bits(bound), wv::float_to_bits(bmax)};
  } else {
    a.dyn_ent[4u * slot + 1u] = U4{2u, 0u, 0u, 0u};
  }
}

// BPE: bpe::Model::SampleEncode(alpha = 0) (src/bpe_model.cc:38-203) of the word: symbols from the char table, then the
// best adjacent pair again and again -- highest score, then leftmost (:53-56) -- until none is a piece.  Exact.
SPMX_DEVICE void resolve_bpe_lane(const ResolveArgs &a, uint32_t slot, uint32_t *sym, float *pscore, uint32_t *pmerged, bool active) {
  const SpmxDev &d = a.dev;
  if (!active) return;
  const U4 kk = a.dyn_ent[4u * slot];
  const uint32_t kw[4] = {kk.x, kk.y, kk.z, kk.w};
  int n = 1;
  bool good = true;
  sym[0] = char_lookup(d, kSpByte, 1u);
  for (int i = 0; i < 16; ++i) {
    const uint32_t b = (kw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
    if (b == 0x20u) break;                            // (the key's padding: key_dword)
    sym[n << 6] = char_lookup(d, b, 1u);
    ++n;
  }
  // a character without a symbol never merges (kernels_bpe.h pair_lookup) and is the unknown piece in the end: under byte
  // fallback its byte's piece (sentencepiece_processor.cc:581-601) -- kept; else a run of them is one id -- not kept
  const bool bf = (d.flags & kNfByteFallback) != 0;
  for (int i = 0; i < n; ++i) if (sym[i << 6] >= kSsUnknown && (!bf || i == 0)) good = false;
  uint32_t alive = n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u, pmask = 0u;
  if (good) {
    for (int i = 0; i + 1 < n; ++i) {
      uint32_t mg = 0; float sc = 0.f;
      if (pair_lookup(d, sym[i << 6], sym[(i + 1) << 6], &mg, &sc)) { pmask |= 1u << i; pscore[i << 6] = sc; pmerged[i << 6] = mg; }
    }
    for (;;) {
      int bi = -1;
      float bs = 0.f;
      for (uint32_t m = pmask; m != 0u; m &= m - 1u) {
        const int i = wv::ffs64(static_cast<uint64_t>(m)) - 1;
        if (bi < 0 || pscore[i << 6] > bs) { bi = i; bs = pscore[i << 6]; }
      }
      if (bi < 0) break;
      const uint32_t above = alive & ~((2u << bi) - 1u);
      const int j = wv::ffs64(static_cast<uint64_t>(above)) - 1;                    // the right symbol (the pair is live)
      const uint32_t bm = pmerged[bi << 6];
      sym[bi << 6] = bm;
      alive &= ~(1u << j);
      pmask &= ~((1u << bi) | (1u << j));
      const uint32_t below = alive & ((1u << bi) - 1u);
      if (below) {
        const int q = 31 - (wv::clz64(static_cast<uint64_t>(below)) - 32);
        pmask &= ~(1u << q);
        uint32_t mg = 0; float sc = 0.f;
        if (pair_lookup(d, sym[q << 6], bm, &mg, &sc)) { pmask |= 1u << q; pscore[q << 6] = sc; pmerged[q << 6] = mg; }
      }
      const uint32_t after = alive & ~((2u << bi) - 1u);
      if (after) {
        const int r = wv::ffs64(static_cast<uint64_t>(after)) - 1;
        uint32_t mg = 0; float sc = 0.f;
        if (pair_lookup(d, bm, sym[r << 6], &mg, &sc)) { pmask |= 1u << bi; pscore[bi << 6] = sc; pmerged[bi << 6] = mg; }
      }
    }
  }
  uint32_t ids[kDynMaxWide];
  uint32_t id_or = 0;
  int cnt = 0;
  for (uint32_t m = alive; good && m != 0u; m &= m - 1u) {
    const int i = wv::ffs64(static_cast<uint64_t>(m)) - 1;
    const uint32_t sy = sym[i << 6];
    uint32_t f = sy;
    if (sy >= kSsUnknown) {                          // (byte fallback, i >= 1: character i is byte i - 1 of the key)
      f = static_cast<uint32_t>(d.byte_ids[(kw[(i - 1) >> 2] >> (8 * ((i - 1) & 3))) & 0xFFu]);
    } else {
      if (sy >= d.n_pieces) {                        // PieceToId (:178): only the extra characters go through sym_final
        f = d.sym_final[sy];
        if (f & kSfControl) { good = false; break; }
        f &= kSfIdMask;
      }
      if (static_cast<int32_t>(f) == d.unk_id) { good = false; break; }
    }
    if (cnt >= static_cast<int>(kDynMaxWide)) { good = false; break; }
    ids[cnt++] = f;
    id_or |= f;
  }
  const bool wide = cnt > static_cast<int>(kDynMaxIds);
  if (wide && id_or > 0xFFFFu) good = false;
  if (good && cnt > 0) {
    uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < cnt; ++k) {
      if (wide) o[k >> 1] |= ids[k] << (16 * (k & 1)); else o[k] = ids[k];
    }
    a.dyn_ent[4u * slot + 2u] = U4{o[0], o[1], o[2], o[3]};
    a.dyn_ent[4u * slot + 3u] = U4{o[4], o[5], o[6], o[7]};
    a.dyn_ent[4u * slot + 1u] = U4{1u, static_cast<uint32_t>(cnt) | (wide ? kDynWide : 0u), 0u, wv::float_to_bits(3.0e38f)};
  } else {
    a.dyn_ent[4u * slot + 1u] = U4{2u, 0u, 0u, 0u};
  }
}

// ---- collected words that are NOT plain ASCII (dev.h kNfWordLocalNorm) --------------------------------------------------
// The word is normalized BY ITSELF, as a sentence of its own (norm_lane_any: dummy prefix, charsmap rules, U+FFFD for a
// malformed byte, spaces a rule or a tab brings escaped / collapsed / cut at the ends) -- which is what Normalize makes of it
// inside its sentence, one space symbol in front included (tables.cc: no charsmap key reaches across a 0x20, extra
// whitespace is removed).  The normalized string may hold several words of the model (a tab inside the raw word); no
// piece reaches across their boundaries (kNfUniWordwise / kNfBpeWordwise), so the segmentation of the string is the
// concatenation of theirs, and the margin analysis of resolve_unigram_lane holds for the string as a whole.
constexpr int kRnBytes = 96;                   // normalized bytes of a word that is kept (longer: not usable)
constexpr int kRnChars = 32;                   // ... and characters
constexpr uint32_t kRnLdsPerLane = (kRnChars + 2) * 12u + kRnBytes + 16u + kRnChars + 4u + kRawWinBytes;
struct RnText { uint8_t *nb; uint8_t *cb; int n, nch; };     // bytes, byte offset of every character (cb[nch] = n)

// raw key of `slot` -> its normalized text in the lane's LDS; false: not usable (empty, too long)
SPMX_DEVICE bool resolve_normalize(const ResolveArgs &a, uint32_t slot, uint8_t *nb, uint8_t *cb, uint8_t *rawwin, RnText *out) {
  const SpmxDev &d = a.dev;
  const U4 k = a.dyn_ent[4u * slot];
  const uint32_t kw[4] = {k.x, k.y, k.z, k.w};
  int L = 0;
  for (int i = 0; i < 16; ++i) if (((kw[i >> 2] >> (8 * (i & 3))) & 0xFFu) != 0x20u && L == i) L = i + 1;   // (the key's padding: key_dword)
  // A literal U+2581 in the raw word is not for this path: Normalize cuts space symbols at the END OF THE SENTENCE whether
  // they were spaces or literals (src/normalizer.cc:166-176), so what such a word normalizes to depends on whether it is
  // the sentence's last.
  for (int i = 0; i + 2 < L; ++i) {
    auto kb = [&](int q) -> uint32_t { return (kw[q >> 2] >> (8 * (q & 3))) & 0xFFu; };
    if (kb(i) == 0xE2u && kb(i + 1) == 0x96u && kb(i + 2) == 0x81u) return false;
  }
  FlatSink sink{nb, nullptr, kRnBytes};
  int nsp = 0;
  const int n = norm_lane_any(d, reinterpret_cast<const uint8_t *>(a.dyn_ent + 4u * slot), 0, L, sink, rawwin, &nsp);
  if (n < 0 || n > kRnBytes) return false;
  int nch = 0;                                   // (n == 0: the word is nothing but what Normalize drops -- a tab, U+3000: no ids)
  for (int p = 0; p < n;) {
    if (nch >= kRnChars) return false;
    cb[nch++] = static_cast<uint8_t>(p);
    const uint32_t c = nb[p];
    int l = c == SpByteOf(d) ? 1 : OneCharLenDev(c);
    if (l > n - p) l = n - p;
    p += l;
  }
  cb[nch] = static_cast<uint8_t>(n);
  out->nb = nb; out->cb = cb; out->n = n; out->nch = nch;
  return true;
}

// unigram: resolve_unigram_lane over the CHARACTERS of the normalized text (positions = character indices)
SPMX_DEVICE void resolve_norm_unigram_lane(const ResolveArgs &a, uint32_t slot, float *best, float *second, uint32_t *bp, const RnText &T) {
  const SpmxDev &d = a.dev;
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  const int n = T.nch;
  const float kNone = -3.0e38f;
  for (int i = 0; i <= n; ++i) { bp[i << 6] = 0u; best[i << 6] = kNone; second[i << 6] = kNone; }
  best[0] = 0.f;
  float wmag = 0.f;
  auto relax = [&](int e, float cand, uint32_t word) __attribute__((always_inline)) {
    if (fabsf(cand) > wmag) wmag = fabsf(cand);
    if (bp[e << 6] == 0u || cand > best[e << 6]) { second[e << 6] = bp[e << 6] == 0u ? kNone : best[e << 6]; best[e << 6] = cand; bp[e << 6] = word; }
    else if (cand > second[e << 6]) second[e << 6] = cand;
  };
  for (int s = 0; s < n; ++s) {                                    // :957-1008, every character start in turn
    const float bs = best[s << 6];
    bool single = false;
    uint32_t node = root;
    int e = s;                                                     // characters matched so far end at character index e
    for (int q = T.cb[s]; q < T.n;) {
      const uint32_t c = T.nb[q];
      const U4 u = ptrie[node ^ c];
      if ((u.x & 0x1FFu) != (0x100u | c)) break;                  // :969-971
      ++q;
      node = u.x >> kDatBaseShiftDev;
      if (q == T.cb[e + 1]) {                                      // a character boundary (a piece ends at one)
        ++e;
        if ((u.x & kDatTerminalDev) && !(u.y & kPtUnused)) {       // :973-974
          relax(e, wv::bits_to_float(u.z) + bs, (u.y & kBwIdMask) | (static_cast<uint32_t>(e - s) << kBwLenShift));
          if (e == s + 1) single = true;                           // :990
        }
      }
    }
    if (!single) relax(s + 1, d.unk_score + bs, (1u << kBwLenShift) | kBwUnk);      // :995-1005
  }
  // the best path, its smallest lead, its ids (as resolve_unigram_lane)
  uint32_t ids[kDynMaxWide];
  int cnt = 0;
  uint32_t id_or = 0;
  float gap = 3.0e38f, bound = 0.f;
  bool good = true;
  const bool bf = (d.flags & kNfByteFallback) != 0;
  bool last_unk = false, first_unk = false, right_unk = false;
  auto push = [&](uint32_t id) __attribute__((always_inline)) { if (cnt < static_cast<int>(kDynMaxWide)) { ids[cnt++] = id; id_or |= id; } else good = false; };
  for (int e = n; e > 0 && good;) {
    const uint32_t bw = bp[e << 6];
    const int bl = static_cast<int>((bw >> kBwLenShift) & kBwLenMask);
    if (bw == 0u || bl == 0 || bl > e) { good = false; break; }
    if (second[e << 6] > kNone) { const float g = best[e << 6] - second[e << 6]; if (g < gap) gap = g; }
    const bool unk = (bw & kBwUnk) != 0;
    if (e == n) last_unk = unk;
    if (e - bl == 0) first_unk = unk;
    if (unk) {
      bound += ceilf(fabsf(d.unk_score)) + 1.f;
      if (bf) {                                                    // the bytes of the unknown character, last first (the list is reversed below)
        const int b0 = T.cb[e - 1], b1 = T.cb[e];
        if (b1 - b0 == 1 && T.nb[b0] == SpByteOf(d)) { push(static_cast<uint32_t>(d.byte_ids[0x81])); push(static_cast<uint32_t>(d.byte_ids[0x96])); push(static_cast<uint32_t>(d.byte_ids[0xE2])); }
        else for (int q = b1 - 1; q >= b0; --q) push(static_cast<uint32_t>(d.byte_ids[T.nb[q]]));
      } else if (!right_unk) {
        push(static_cast<uint32_t>(d.unk_id));
      }
      right_unk = true;
    } else {
      push(bw & kBwIdMask);
      bound += ceilf(fabsf(d.pscore[bw & kBwIdMask])) + 1.f;
      right_unk = false;
    }
    e -= bl;
  }
  if (bf) { first_unk = false; last_unk = false; }
  float bmax = 0.f;
  if (good) {
    const uint32_t we = (wv::float_to_bits(wmag) >> 23) & 0xFFu;
    const float ulp_w = we > 23u ? wv::bits_to_float((we - 23u) << 23) : 0.f;
    const float g2 = gap >= 3.0e38f ? gap : gap - 2.f * static_cast<float>(n) * ulp_w;
    if (!(g2 > 0.f)) good = false;
    else if (a.unsafe || g2 >= 3.0e38f) bmax = 3.0e38f;
    else {
      const float thr = g2 / (4.f * static_cast<float>(n) + 4.f);
      const int te = static_cast<int>((wv::float_to_bits(thr) >> 23) & 0xFFu) - 127;
      const int k = te + 23;
      if (k + 1 > 126) bmax = 3.0e38f;
      else if (k + 1 < -120) good = false;
      else {
        const float lim = wv::bits_to_float(static_cast<uint32_t>(k + 1 + 127) << 23);
        bmax = (lim - wmag * 1.000001f) * 0.999999f;
        if (!(bmax > 0.f)) good = false;
      }
    }
  }
  const bool wide = cnt > static_cast<int>(kDynMaxIds);
  if (wide && id_or > 0xFFFFu) good = false;
  if (good && cnt > 0) {
    uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < cnt; ++k) {
      const uint32_t id = ids[cnt - 1 - k];
      if (wide) o[k >> 1] |= id << (16 * (k & 1)); else o[k] = id;
    }
    a.dyn_ent[4u * slot + 2u] = U4{o[0], o[1], o[2], o[3]};
    a.dyn_ent[4u * slot + 3u] = U4{o[4], o[5], o[6], o[7]};
    a.dyn_ent[4u * slot + 1u] = U4{1u, static_cast<uint32_t>(cnt) | (first_unk ? kDynFirstUnk : 0u) | (last_unk ? kDynLastUnk : 0u) | (wide ? kDynWide : 0u),
                                  wv::float_to_bits(bound), wv::float_to_bits(bmax)};
  } else {
    a.dyn_ent[4u * slot + 1u] = U4{2u, 0u, 0u, 0u};
  }
}

// BPE: resolve_bpe_lane over the characters of the normalized text
SPMX_DEVICE void resolve_norm_bpe_lane(const ResolveArgs &a, uint32_t slot, uint32_t *sym, float *pscore, uint32_t *pmerged, const RnText &T) {
  const SpmxDev &d = a.dev;
  const int n = T.nch;                                             // (<= 32: the masks below)
  bool good = true;
  const bool bf = (d.flags & kNfByteFallback) != 0;
  for (int i = 0; i < n; ++i) {
    uint32_t bytes = 0;
    const int b0 = T.cb[i], len = T.cb[i + 1] - b0;
    for (int q = 0; q < len && q < 4; ++q) bytes |= static_cast<uint32_t>(T.nb[b0 + q]) << (8 * q);
    sym[i << 6] = char_lookup(d, bytes, static_cast<uint32_t>(len));
    // a character without a symbol never merges and is the unknown piece in the end: under byte fallback its bytes'
    // pieces; else a run of them is ONE id (sentencepiece_processor.cc:609-613), which may continue into the next word:
    // kDynFirstUnk / kDynLastUnk.  (The space symbol always has a symbol.)
    if (sym[i << 6] >= kSsUnknown && len == 1 && T.nb[b0] == SpByteOf(d)) good = false;
  }
  uint32_t alive = n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u, pmask = 0u;
  if (good) {
    for (int i = 0; i + 1 < n; ++i) {
      uint32_t mg = 0; float sc = 0.f;
      if (pair_lookup(d, sym[i << 6], sym[(i + 1) << 6], &mg, &sc)) { pmask |= 1u << i; pscore[i << 6] = sc; pmerged[i << 6] = mg; }
    }
    for (;;) {
      int bi = -1;
      float bs = 0.f;
      for (uint32_t m = pmask; m != 0u; m &= m - 1u) {
        const int i = wv::ffs64(static_cast<uint64_t>(m)) - 1;
        if (bi < 0 || pscore[i << 6] > bs) { bi = i; bs = pscore[i << 6]; }
      }
      if (bi < 0) break;
      const uint32_t above = bi >= 31 ? 0u : alive & ~((2u << bi) - 1u);
      const int j = wv::ffs64(static_cast<uint64_t>(above)) - 1;
      if (j < 0) { pmask &= ~(1u << bi); continue; }
      const uint32_t bm = pmerged[bi << 6];
      sym[bi << 6] = bm;
      alive &= ~(1u << j);
      pmask &= ~((1u << bi) | (1u << j));
      const uint32_t below = alive & ((1u << bi) - 1u);
      if (below) {
        const int q = 31 - (wv::clz64(static_cast<uint64_t>(below)) - 32);
        pmask &= ~(1u << q);
        uint32_t mg = 0; float sc = 0.f;
        if (pair_lookup(d, sym[q << 6], bm, &mg, &sc)) { pmask |= 1u << q; pscore[q << 6] = sc; pmerged[q << 6] = mg; }
      }
      const uint32_t after = bi >= 31 ? 0u : alive & ~((2u << bi) - 1u);
      if (after) {
        const int r = wv::ffs64(static_cast<uint64_t>(after)) - 1;
        uint32_t mg = 0; float sc = 0.f;
        if (pair_lookup(d, bm, sym[r << 6], &mg, &sc)) { pmask |= 1u << bi; pscore[bi << 6] = sc; pmerged[bi << 6] = mg; }
      }
    }
  }
  uint32_t ids[kDynMaxWide];
  uint32_t id_or = 0;
  int cnt = 0;
  auto push = [&](uint32_t id) __attribute__((always_inline)) { if (cnt < static_cast<int>(kDynMaxWide)) { ids[cnt++] = id; id_or |= id; } else good = false; };
  bool first_unk = false, last_unk = false, any_piece = false;
  for (uint32_t m = alive; good && m != 0u; m &= m - 1u) {
    const int i = wv::ffs64(static_cast<uint64_t>(m)) - 1;
    const uint32_t sy = sym[i << 6];
    uint32_t f = sy;
    bool unk = sy >= kSsUnknown;                     // a character without a symbol: never merged, the unknown piece
    if (!unk && sy >= d.n_pieces) {                  // PieceToId (:178): only the extra characters go through sym_final
      f = d.sym_final[sy];
      if (f & kSfControl) { good = false; break; }
      f &= kSfIdMask;
    }
    if (!unk && static_cast<int32_t>(f) == d.unk_id) unk = true;      // (a character whose piece is not in the vocabulary: never merged either)
    if (unk) {
      if (bf) {                                      // its bytes' pieces (sentencepiece_processor.cc:581-601)
        for (int q = T.cb[i]; q < T.cb[i + 1]; ++q) {
          if (T.nb[q] == SpByteOf(d)) { push(static_cast<uint32_t>(d.byte_ids[0xE2])); push(static_cast<uint32_t>(d.byte_ids[0x96])); push(static_cast<uint32_t>(d.byte_ids[0x81])); }
          else push(static_cast<uint32_t>(d.byte_ids[T.nb[q]]));
        }
        last_unk = false;
      } else {                                       // a run of them is ONE id (:609-613)
        if (!last_unk) push(static_cast<uint32_t>(d.unk_id));
        if (!any_piece) first_unk = true;
        last_unk = true;
      }
      any_piece = true;
      continue;
    }
    any_piece = true;
    last_unk = false;
    push(f);
  }
  const bool wide = cnt > static_cast<int>(kDynMaxIds);
  if (wide && id_or > 0xFFFFu) good = false;
  if (good && cnt > 0) {
    uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < cnt; ++k) {
      if (wide) o[k >> 1] |= ids[k] << (16 * (k & 1)); else o[k] = ids[k];
    }
    a.dyn_ent[4u * slot + 2u] = U4{o[0], o[1], o[2], o[3]};
    a.dyn_ent[4u * slot + 3u] = U4{o[4], o[5], o[6], o[7]};
    a.dyn_ent[4u * slot + 1u] = U4{1u, static_cast<uint32_t>(cnt) | (first_unk ? kDynFirstUnk : 0u) | (last_unk ? kDynLastUnk : 0u) | (wide ? kDynWide : 0u),
                                  0u, wv::float_to_bits(3.0e38f)};
  } else {
    a.dyn_ent[4u * slot + 1u] = U4{2u, 0u, 0u, 0u};
  }
}

constexpr uint32_t kResolveLdsBytesAll = 64u * (kResolvePos * 12u + 20u) > 64u * kRnLdsPerLane ? 64u * (kResolvePos * 12u + 20u) : 64u * kRnLdsPerLane;
SPMX_HD inline uint32_t ResolveLdsBytesAll() { return kResolveLdsBytesAll; }
SPMX_DEVICE void word_resolve_block(const ResolveArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  uint32_t count = *a.dyn_count;
  if (count > a.dyn_cap) count = a.dyn_cap;
  float *best = reinterpret_cast<float *>(smem) + lane;
  float *second = reinterpret_cast<float *>(smem + 64u * kResolvePos * 4u) + lane;
  uint32_t *bp = reinterpret_cast<uint32_t *>(smem + 64u * kResolvePos * 8u) + lane;
  uint8_t *wb = smem + 64u * kResolvePos * 12u + static_cast<uint32_t>(lane) * 20u;
  // (words that are not plain ASCII: the same LDS, laid out for up to kRnChars + 1 positions, the normalized bytes and
  // norm_lane_any's raw window)
  float *nbest = reinterpret_cast<float *>(smem) + lane;
  float *nsecond = reinterpret_cast<float *>(smem + 64u * (kRnChars + 2) * 4u) + lane;
  uint32_t *nbp = reinterpret_cast<uint32_t *>(smem + 64u * (kRnChars + 2) * 8u) + lane;
  uint8_t *ntext = smem + 64u * (kRnChars + 2) * 12u + static_cast<uint32_t>(lane) * (kRnBytes + 16u + kRnChars + 4u);
  uint8_t *nraw = smem + 64u * ((kRnChars + 2) * 12u + kRnBytes + 16u + kRnChars + 4u) + static_cast<uint32_t>(lane) * kRawWinBytes;
  const bool norm_ok = (a.dev.flags & kNfWordLocalNorm) != 0;
  const uint32_t lanes = static_cast<uint32_t>(wv::grid_size()) * 64u;
  for (uint32_t base = static_cast<uint32_t>(wv::block_id()) * 64u; base < count; base += lanes) {
    const uint32_t i = base + static_cast<uint32_t>(lane);
    const bool active = i < count;
    const uint32_t slot = active ? a.dyn_list[i] : 0u;
    // plain: every key byte 0x20 .. 0x7E (the word's bytes 0x21 .. 0x7E and the padding)
    bool plain = true;
    if (active) {
      const U4 k = a.dyn_ent[4u * slot];
      auto pl = [](uint32_t v) -> bool {
        return (((v + 0x01010101u) | v) & 0x80808080u) == 0u && (((v - 0x20202020u) & ~v) & 0x80808080u) == 0u;
      };
      plain = pl(k.x) && pl(k.y) && pl(k.z) && pl(k.w);
    }
    // (BPE: the normalizing path also keeps characters the model lacks -- a plain "?!" or "3.14159" in a vocabulary without
    // those characters -- so every word takes it where the model allows)
    if (norm_ok && a.dev.model_type == 2) plain = false;
    if (wv::any(active && plain)) {
      if (a.dev.model_type == 2) resolve_bpe_lane(a, slot, reinterpret_cast<uint32_t *>(best), second, bp, active && plain);
      else resolve_unigram_lane(a, slot, best, second, bp, wb, active && plain);
    }
    wv::sync();
    if (wv::any(active && !plain)) {
      if (active && !plain) {
        RnText T{nullptr, nullptr, 0, 0};
        if (!norm_ok || !resolve_normalize(a, slot, ntext, ntext + kRnBytes + 16u, nraw, &T)) a.dyn_ent[4u * slot + 1u] = U4{2u, 0u, 0u, 0u};
        else if (T.n == 0) {                           // no ids, no share of the bound, valid whatever came before
          a.dyn_ent[4u * slot + 2u] = U4{0, 0, 0, 0};
          a.dyn_ent[4u * slot + 3u] = U4{0, 0, 0, 0};
          a.dyn_ent[4u * slot + 1u] = U4{1u, 0u, 0u, wv::float_to_bits(3.0e38f)};
        }
        else if (a.dev.model_type == 2) resolve_norm_bpe_lane(a, slot, reinterpret_cast<uint32_t *>(nbest), nsecond, nbp, T);
        else resolve_norm_unigram_lane(a, slot, nbest, nsecond, nbp, T);
      }
    }
    wv::sync();
  }
}

}  // namespace spmx
#endif
