// Unigram segmentation, WAVE-COOPERATIVE form: one SENTENCE PER WAVEFRONT, exact for any length.
// Reference: unigram::Model::EncodeOptimized (src/unigram_model.cc:889-1020).
//
// The lane-per-sentence kernels (kernels_stream.h, kernels_word.h) need tens of thousands of sentences to fill the
// chip and give one sentence a single lane: a document of a megabyte is 2.6 s of dependent iterations there.  This form
// gives a sentence the whole wavefront, 64 consecutive byte positions at a time:
//
//   walk   (parallel) lane l walks the piece trie from start c + l -- the reference's inner loop (:965-993) -- and
//          writes what it finds into row l of an LDS matrix indexed by LENGTH: entry [l][k - 1] is the piece of k bytes
//          that begins at c + l (none: a NaN score).  Which pieces match at a start does not depend on any score, so the
//          64 walks are independent.  The UNK candidate of a start without a one-character piece (:995-1005) is the
//          entry at the character's length (no piece is there);
//   fold   the relaxations of best_path_ends_at in the reference's order, starts ascending one after another, with the
//          scores in REGISTERS (round 6): lane j holds best_path_ends_at[c + j] (-inf: no candidate yet) and, in a
//          second pair of registers, [c + 64 + j]: what a piece that begins in this chunk can reach beyond it.  A step
//          takes the start's final score with one v_readlane, and lane j relaxes ITS position with the start's piece
//          that ends there -- entry [l][j - l - 1], one conflict-free LDS read that does not depend on the step before;
//          the pieces of one start end at different positions, so the order among them does not matter (:973-993).
//          The arithmetic is the reference's -- double add, double compare against the float stored, float store
//          (:979-989), the UNK candidate in float -- in one of two forms:
//            exact   (uw_fold_exact) as written: ~45 instructions a step, of which the seven 64-bit ones alone cost
//                    150 cycles (scripts/ubench/fold_probe.hip: 316 cycles a step);
//            float   (uw_fold_float) while |best_path_score| stays below the model's uw_f32_limit (dev.h) the double
//                    sum of the two floats is EXACT, so what the reference stores is the float sum, and its comparison
//                    is the float sum's except on a tie of the rounded values, where the sign of the rounding error
//                    decides.  The step is one add, two compares, three selects with the start a compile-time
//                    constant (64 steps unrolled; 88 cycles a step in the probe); a tie is only RECORDED, and a chunk
//                    that saw one is folded again the exact way from the scores it began with;
//   then id and length of every position's best piece go to the sentence's bid / blen arrays, the backtrack
//   (:1010-1018) marks the token ends and emit_wave (kernels.h) writes the ids.
//
// The sentence's arrays (normalized text, bid, blen) live in a slice of the long form's HBM pool (kernels_long.h), so
// nothing here depends on the sentence length.  For models whose longest piece has at most kUwMaxPiece bytes (a piece
// that begins in a chunk ends in it or in the next); the others take the lane-per-sentence long form (kernels_long.h).
// Used for documents (length classes beyond 4 KiB), and for what the word kernels leave when that is too little to
// fill the lane-per-sentence kernel.
#ifndef SPMX_KERNELS_UNIWAVE_H_
#define SPMX_KERNELS_UNIWAVE_H_

namespace spmx {

constexpr uint32_t kUwRing = 256;       // the backtrack's window of blen entries
constexpr uint32_t kUwWindow = 256;     // bytes of text staged per chunk: 64 starts + the longest piece
constexpr uint32_t kUwMaxPiece = 64;    // the longest piece (bytes) this form takes: matrix rows have that many entries
constexpr uint32_t kUwUnreached = 0xFFFFFFFFu;
constexpr uint32_t kUwNone = 0xFFFFFFFFu;       // matrix entry, word x: no piece of this length begins here
constexpr uint32_t kUwNan = 0x7FC00000u;        // ... and its word y: a score that wins no comparison
// matrix rows: one entry per piece length -- 16, 32 or 64 (the kernel is compiled for each)
SPMX_HD inline uint32_t UniWaveRow(int max_piece_bytes) { return max_piece_bytes <= 16 ? 16u : max_piece_bytes <= 32 ? 32u : 64u; }
// the matrix with one entry in front of it and 64 behind: what the lanes a step does not concern read
SPMX_HD constexpr uint32_t UniWaveMatrixEntries(uint32_t ML) { return 2u + 64u * ML + 64u; }
// (the backtrack's window and the fallback normalizer's raw window lie over the matrix: it is dead before the first
// chunk and behind the last)
SPMX_HD inline uint32_t UniWaveLdsBytes(uint32_t ML) {
  return UniWaveMatrixEntries(ML) * 8u + kUwWindow + 16u;
}
static_assert(UniWaveMatrixEntries(16) * 8u >= kUwRing * 16u && kUwRing * 12u >= kUwRing * 4u + kRawWinBytes, "the aliased windows fit the smallest matrix");

struct UniWaveLds {
  uint32_t *ring_b;   // [4][kUwRing] the backtrack's window: lengths, two predecessor maps, marks
  U2 *cands;          // entry [l][k] at cands[l * ML + k], two entries in front of [0][0] and 64 behind [63][ML - 1]: {id | length << 24 | user-defined << 31, score bits} of
                      // the piece of k + 1 bytes that begins at c + l; {kUwNone, kUwNan}: none.  A user-defined piece carries
                      // (float)length * max_score_ for a score.  (16-byte aligned: the rows are cleared two entries a store)
  uint8_t *win;       // [kUwWindow + 16] text window
  uint8_t *rawwin;    // [kRawWinBytes] lane 0's raw-text window of norm_lane_any
};
SPMX_DEVICE UniWaveLds carve_uniwave(unsigned char *smem, uint32_t ML) {
  UniWaveLds T;
  T.ring_b = reinterpret_cast<uint32_t *>(smem);                      // (over the matrix)
  T.rawwin = smem + kUwRing * 4u;                                     // (over the matrix)
  T.cands = reinterpret_cast<U2 *>(smem) + 2;                         // (entry [0][0] sits at a 16-byte boundary)
  T.win = smem + UniWaveMatrixEntries(ML) * 8u;
  return T;
}

// ---- the fold of one chunk.  M = entry [0][-1]; S: the chunk's character starts; cur / nxt: best_path_ends_at of
// positions c + lane / c + 64 + lane (score; piece: id | length << 24, kUwUnreached) ----

// EXACT form: the reference's arithmetic as written.
template <uint32_t ML>
SPMX_DEVICE void uw_fold_exact(const SpmxDev &d, const U2 *M, uint64_t S, int lane, float &cur_s, uint32_t &cur_b, float &nxt_s,
                               uint32_t &nxt_b) {
  const uint32_t unk = static_cast<uint32_t>(d.unk_id) & 0x00FFFFFFu;
  auto relax = [&](float &s, uint32_t &b, const U2 &ent, bool mine, double dbs, float fu) __attribute__((always_inline)) {
    const bool has = mine && ent.x != kUwNone;
    const bool isunk = (ent.x & 0x00FFFFFFu) == unk;                  // (no piece of the trie has that id; its bit 31 is not "user-defined")
    // a user-defined piece (bit 31) carries (float)length * max_score_: its score is that - 0.1 in double (:979-981)
    const double adj = wv::bits_to_double(0xBFB999999999999Aull & static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(ent.x) >> 31)));   // -0.1 or 0.0
    const double score = static_cast<double>(wv::bits_to_float(ent.y)) + adj;    // (+ 0.0 otherwise: the same value)
    const double cand = score + dbs;                                  // :982-983
    const float nv = isunk ? fu : static_cast<float>(cand);           // (:997-1001: the UNK candidate in float)
    const bool gt = isunk ? fu > s : cand > static_cast<double>(s);   // :984-985 (s = -inf: no candidate yet)
    const bool win = has && gt;
    s = win ? nv : s;
    b = win ? (ent.x & 0x7FFFFFFFu) : b;
  };
  for (uint64_t m = S; m != 0; m &= m - 1) {
    const int l = wv::ffs64(m) - 1;
    const float bs = wv::bits_to_float(wv::read_lane(wv::float_to_bits(cur_s), l));
    const double dbs = static_cast<double>(bs);
    const float fu = d.unk_score + bs;
    relax(cur_s, cur_b, M[static_cast<uint32_t>(l) * (ML - 1u) + static_cast<uint32_t>(lane)], static_cast<uint32_t>(lane - l - 1) < ML, dbs, fu);
    if (static_cast<uint32_t>(l) + ML >= 64u)                          // (wave-uniform) its row reaches beyond the chunk
      relax(nxt_s, nxt_b, M[static_cast<uint32_t>(l) * (ML - 1u) + 64u + static_cast<uint32_t>(lane)],
            static_cast<uint32_t>(lane + 63 - l) < ML, dbs, fu);
  }
}

// One relaxation of the float form: the position's score / piece s / b, the candidate's score a (NaN: none) and word x,
// the start's score bs.
template <bool TIES>
SPMX_DEVICE void uw_relax_float(float &s, uint32_t &b, uint32_t &tie, float a, float bs, uint32_t x) {
  const float sum = a + bs;
  if (!TIES) {
    const bool gt = sum > s;
    tie = sum == s ? 1u : tie;
    s = gt ? sum : s;
    b = gt ? x : b;
  } else {
    // (a + bs exactly) > s  <=>  sum > s, or sum == s and the rounding error (a + bs) - sum is above 0
    const float bb = sum - a;
    const float err = (a - (sum - bb)) + (bs - bb);
    const float dif = sum - s;                                        // (its sign is exact; +inf over "no candidate yet", NaN for "none")
    const float key = dif == 0.f ? (static_cast<int32_t>(x) < 0 ? -1.f : err) : dif;
    const bool gt = key > 0.f;
    s = gt ? sum : s;
    b = gt ? x : b;
  }
}

// FLOAT form (see the top).  D: the longest piece (bytes) any start of the chunk has.  ALL: every position of the chunk
// is a character start (plain ASCII: no test per step).  The matrix entries are asked for three steps ahead (their
// addresses depend on nothing).
//   TIES = false  a comparison that meets a tie of the rounded values is only recorded: returns nonzero in some lane if
//                 there was one, and the caller folds the chunk again;
//   TIES = true   a tie is decided as the reference's double comparison decides it -- by the sign of the float sum's
//                 rounding error (TwoSum: the error of a float addition is a float); never for the UNK candidate, whose
//                 comparison IS the float one (:999-1000; its entry has bit 31 set).  Eight more instructions a step:
//                 for the chunks deep inside a long document, where |best_path_score| has grown to 10^5 .. 10^6, a float
//                 holds 1/64 .. 1/4 and different paths tie all the time.  Returns 0.
template <uint32_t ML, bool ALL, bool TIES>
SPMX_DEVICE uint32_t uw_fold_float(const U2 *M, uint64_t S, uint32_t D, int lane, float &cur_s, uint32_t &cur_b, float &nxt_s,
                                   uint32_t &nxt_b) {
  uint32_t tie = 0;
  const float qnan = wv::bits_to_float(kUwNan);
  const U2 *mine = M + lane;
  uint32_t lm1 = static_cast<uint32_t>(lane) - 1u;
  U2 eb[4];
#pragma unroll
  for (int p = 0; p < 3; ++p) eb[p] = mine[p * static_cast<int>(ML - 1u)];
#pragma unroll
  for (int l = 0; l < 64; ++l) {
    if (l + 3 < 64) eb[(l + 3) & 3] = mine[(l + 3) * static_cast<int>(ML - 1u)];            // [l + 3][lane - l - 4]
    if (!ALL && !((S >> l) & 1ull)) continue;                         // (wave-uniform)
    // [l][lane - l - 1]; the lanes the row does not concern (they read other rows) get a score that wins nothing -- before
    // the start's score is asked for: the chain from one start to the next is readlane, add, compare, select
    wv::opaque(lm1);                                                  // (the lane masks are made where they are used: 64 of them do not fit the scalar registers)
    const U2 e = eb[l & 3];
    const float ey = lm1 - static_cast<uint32_t>(l) < ML ? wv::bits_to_float(e.y) : qnan;
    const float bs = wv::bits_to_float(wv::read_lane(wv::float_to_bits(cur_s), l));
    uw_relax_float<TIES>(cur_s, cur_b, tie, ey, bs, e.x);
    wv::opaque(cur_b); wv::opaque(tie);                               // (here, not at the end of the 64 steps with every mask kept until then)
  }
  // what the chunk's last starts reach beyond it, behind the main loop: their scores are final now, so nothing here
  // waits for a step before it but the position's own comparisons (per position the candidates still come in the order
  // of their starts); the entries are asked for together
#pragma unroll
  for (int l = 64 - static_cast<int>(ML); l < 64; ++l) {
    if (static_cast<uint32_t>(l) + D < 64u) continue;                 // (wave-uniform)
    if (!ALL && !((S >> l) & 1ull)) continue;
    const U2 e = mine[l * static_cast<int>(ML - 1u) + 64];           // [l][64 + lane - l - 1]
    const float bs = wv::bits_to_float(wv::read_lane(wv::float_to_bits(cur_s), l));
    uw_relax_float<TIES>(nxt_s, nxt_b, tie, static_cast<uint32_t>(lane + 63 - l) < ML ? wv::bits_to_float(e.y) : qnan, bs, e.x);
  }
  return tie;
}

// ---- the two halves of a chunk: the WALKER's (text window, character starts, trie walks into the matrix) and the FOLDER's
// (fold, the chunk's row of results to HBM, the matrix rows "none" again).  One wavefront does both in turn
// (unigram_wave), or two wavefronts of a workgroup one each, the walker a chunk ahead (uni_long_pipe_block). ----
struct UwWalker {
  uint32_t text_v;     // this lane's dword of the NEXT chunk's text window
  int next_start;      // the next character start (absolute), across chunks
};
struct UwFolder {
  float cur_s, nxt_s;              // best_path_ends_at[c + lane], [c + 64 + lane]: score (-inf: no candidate yet)
  uint32_t cur_b, nxt_b;           // ... piece: id | length << 24 (kUwUnreached)
  uint32_t tie_credit;             // (wave-uniform) grows with every chunk that met a tie, shrinks with every one that did not
};

SPMX_DEVICE void uw_walker_begin(UwWalker &w, const uint8_t *nt, int nlen, int lane) {
  w.next_start = 0;
  w.text_v = 0;
  if (4 * lane < nlen) w.text_v = *reinterpret_cast<const uint32_t *>(nt + 4 * lane);
}

// The chunk of positions [c, c + 64): its text into `win`, its character starts (*S, wave-uniform), every piece that begins
// at start c + lane into row `lane` of the matrix `cands` (entry [0][0]).  *deep: the row's last entry that is not "none";
// *D (wave-uniform): the largest of them.  cyc += the cycles of the walks.
template <uint32_t ML>
SPMX_DEVICE void uw_walk_chunk(const SpmxDev &d, const uint8_t *nt, int nlen, int c, uint8_t *win, U2 *cands, int lane, UwWalker &w,
                               uint64_t *S_out, uint32_t *D_out, uint32_t *deep_out, unsigned long long *cyc) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  const uint32_t root_w = d.ptrie[0].w;
  const uint32_t spb = SpByteOf(d);
  U2 *row = cands + static_cast<uint32_t>(lane) * ML;
  // ---- the text of this chunk's walks: positions [c, c + kUwWindow), asked for one chunk ahead (the load's latency
  // sits behind the chunk before) ----
  wv::sync();                                                        // (the previous chunk's walks are done with the window)
  *reinterpret_cast<uint32_t *>(win + 4 * lane) = w.text_v;
  {
    const int q = c + 64 + 4 * lane;
    w.text_v = 0;
    if (q < nlen) w.text_v = *reinterpret_cast<const uint32_t *>(nt + q);   // (the slice is padded: a whole dword is readable)
  }
  wv::sync();
  const int s = c + lane;
  const bool valid = s < nlen;
  int step = 1;
  if (valid) {
    const uint32_t b0 = win[lane];
    step = b0 == spb ? 1 : OneCharLenDev(b0);                         // :962-963
    if (step > nlen - s) step = nlen - s;
  }
  const uint64_t S = wv::uniform64(resolve_chain(c, step, valid, &w.next_start));   // which positions are character starts (:1007)
  const unsigned long long tw0 = wv::clock();
  // ---- walk (:965-993): every piece that starts at s, by length, into row `lane` ----
  uint32_t deep = 0;
  {
    const bool is_start = ((S >> lane) & 1ull) != 0;
    bool alive = is_start;
    uint32_t node = root, dep = 0;
    uint32_t wsum = root_w;                                           // child-label summary of the node the walk stands on (dev.h ChildBit)
    bool single = false;                                              // a piece of exactly one character matched (:990)
    // (one level of conditions: the loop is 64 lanes' worth of straight code around one LDS read and one probe)
    while (wv::any(alive)) {
      const int q = s + static_cast<int>(dep);
      const uint32_t cb = win[(q - c) & static_cast<int>(kUwWindow - 1u)];   // (a dead lane reads what it likes)
      // the next byte has a child below the node the walk stands on (dev.h ChildBit: a failing probe is not issued)
      const bool go = alive && q < nlen && dep < ML && ((wsum >> ChildBit(cb)) & 1u) != 0u;   // (no piece is longer than ML bytes)
      U4 u{0u, 0u, 0u, 0u};
      if (go) u = ptrie[node ^ cb];
      alive = go && (u.x & 0x1FFu) == (0x100u | cb);                  // :969-971
      node = alive ? u.x >> kDatBaseShiftDev : node;
      wsum = alive ? u.w : wsum;
      dep += alive ? 1u : 0u;
      if (alive && (u.x & kDatTerminalDev) && !(u.y & kPtUnused)) {   // :973-974
        const bool ud = (u.y & kPtUserDefined) != 0u;
        row[dep - 1u] = U2{(u.y & 0x00FFFFFFu) | (dep << 24) | (ud ? 0x80000000u : 0u),
                           ud ? wv::float_to_bits(static_cast<float>(static_cast<int>(dep)) * d.max_score) : u.z};
        deep = dep;
        single = single || static_cast<int>(dep) == step;
      }
    }
    if (is_start && !single) {                                        // :995-1005: the UNK candidate, `step` bytes long
      const uint32_t us = static_cast<uint32_t>(step);
      row[us - 1u] = U2{(static_cast<uint32_t>(d.unk_id) & 0x00FFFFFFu) | (us << 24) | 0x80000000u, wv::float_to_bits(d.unk_score)};   // (bit 31: uw_relax_float)
      if (deep < us) deep = us;
    }
  }
  *S_out = S;
  *D_out = wv::uniform(wv::read_lane(wv::scan_max(deep), 63));
  *deep_out = deep;
  *cyc += wv::clock() - tw0;
}

// Two chunks at once -- positions [c, c + 64) and [c + 64, c + 128) -- for the walker of uni_long_pipe_block: one text
// window, the character starts of one chunk after the other, and BOTH chunks' walks in one loop (a lane walks from c + lane
// and from c + 64 + lane side by side: two probes in flight, the chain of dependent L2 round trips is shared).
template <uint32_t ML>
SPMX_DEVICE void uw_walk_pair(const SpmxDev &d, const uint8_t *nt, int nlen, int c, uint8_t *win, U2 *cands0, U2 *cands1, int lane,
                              UwWalker &w, uint64_t *S_out, uint32_t *D_out, uint32_t *deep_out, unsigned long long *cyc) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  const uint32_t root_w = d.ptrie[0].w;
  const uint32_t spb = SpByteOf(d);
  U2 *row[2] = {cands0 + static_cast<uint32_t>(lane) * ML, cands1 + static_cast<uint32_t>(lane) * ML};
  wv::sync();                                                        // (the pair before is done with the window)
  *reinterpret_cast<uint32_t *>(win + 4 * lane) = w.text_v;
  {
    const int q = c + 128 + 4 * lane;
    w.text_v = 0;
    if (q < nlen) w.text_v = *reinterpret_cast<const uint32_t *>(nt + q);   // (the slice is padded: a whole dword is readable)
  }
  wv::sync();
  int s[2], step[2];
  uint64_t S[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    s[h] = c + 64 * h + lane;
    const bool valid = s[h] < nlen;
    step[h] = 1;
    if (valid) {
      const uint32_t b0 = win[64 * h + lane];
      step[h] = b0 == spb ? 1 : OneCharLenDev(b0);                    // :962-963
      if (step[h] > nlen - s[h]) step[h] = nlen - s[h];
    }
    S[h] = c + 64 * h > nlen ? 0ull : wv::uniform64(resolve_chain(c + 64 * h, step[h], valid, &w.next_start));   // (:1007; a chunk beyond the text has none)
  }
  const unsigned long long tw0 = wv::clock();
  uint32_t deep[2] = {0u, 0u}, node[2] = {root, root}, dep[2] = {0u, 0u}, wsum[2] = {root_w, root_w};
  bool alive[2], single[2] = {false, false}, is_start[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { is_start[h] = ((S[h] >> lane) & 1ull) != 0; alive[h] = is_start[h]; }
  while (wv::any(alive[0] || alive[1])) {
    uint32_t cb[2];
    bool go[2];
    U4 u[2] = {U4{0u, 0u, 0u, 0u}, U4{0u, 0u, 0u, 0u}};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = s[h] + static_cast<int>(dep[h]);
      cb[h] = win[(q - c) & static_cast<int>(kUwWindow - 1u)];
      go[h] = alive[h] && q < nlen && dep[h] < ML && ((wsum[h] >> ChildBit(cb[h])) & 1u) != 0u;
    }
    if (go[0]) u[0] = ptrie[node[0] ^ cb[0]];                          // (both probes asked for before either is looked at)
    if (go[1]) u[1] = ptrie[node[1] ^ cb[1]];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      alive[h] = go[h] && (u[h].x & 0x1FFu) == (0x100u | cb[h]);      // :969-971
      node[h] = alive[h] ? u[h].x >> kDatBaseShiftDev : node[h];
      wsum[h] = alive[h] ? u[h].w : wsum[h];
      dep[h] += alive[h] ? 1u : 0u;
      if (alive[h] && (u[h].x & kDatTerminalDev) && !(u[h].y & kPtUnused)) {   // :973-974
        const bool ud = (u[h].y & kPtUserDefined) != 0u;
        row[h][dep[h] - 1u] = U2{(u[h].y & 0x00FFFFFFu) | (dep[h] << 24) | (ud ? 0x80000000u : 0u),
                                 ud ? wv::float_to_bits(static_cast<float>(static_cast<int>(dep[h])) * d.max_score) : u[h].z};
        deep[h] = dep[h];
        single[h] = single[h] || static_cast<int>(dep[h]) == step[h];
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (is_start[h] && !single[h]) {                                  // :995-1005: the UNK candidate, `step` bytes long
      const uint32_t us = static_cast<uint32_t>(step[h]);
      row[h][us - 1u] = U2{(static_cast<uint32_t>(d.unk_id) & 0x00FFFFFFu) | (us << 24) | 0x80000000u, wv::float_to_bits(d.unk_score)};
      if (deep[h] < us) deep[h] = us;
    }
    S_out[h] = S[h];
    D_out[h] = wv::uniform(wv::read_lane(wv::scan_max(deep[h]), 63));
    deep_out[h] = deep[h];
  }
  *cyc += wv::clock() - tw0;
}

SPMX_DEVICE void uw_folder_begin(UwFolder &f, int lane) {
  const float ninf = -__builtin_inff();
  f.cur_s = lane == 0 ? 0.f : ninf;                                   // best_path_ends_at[0] = {0, nothing}
  f.nxt_s = ninf;
  f.cur_b = kUwUnreached; f.nxt_b = kUwUnreached;
  f.tie_credit = 0;
}

// The fold of the chunk whose matrix is `cands` (entry [0][0]): the starts one after another (a start's score must be
// final before its candidates are scored).  cyc += its cycles.
template <uint32_t ML>
SPMX_DEVICE void uw_fold_chunk(const SpmxDev &d, const U2 *cands, uint64_t S, uint32_t D, int lane, UwFolder &f, unsigned long long *cyc) {
  const unsigned long long t0 = wv::clock();
  const float ninf = -__builtin_inff();
  const U2 *M = cands - 1;
  if (S != 0) {
    // float arithmetic while every score the chunk starts from is above -uw_f32_limit (nxt has none yet)
    const bool f32 = d.uw_f32_limit > 0.f && !wv::any(f.cur_s <= -d.uw_f32_limit && f.cur_s > ninf);
    if (f32) {
      // which float flavour: the one that decides ties itself while ties are frequent (tie_credit), else the shorter
      // one, and the chunk again with the other if it met one
      const bool all = S == ~0ull;
      bool ties = !all || f.tie_credit > 8u;
      if (!ties) {
        const float s0 = f.cur_s;
        const uint32_t b0 = f.cur_b;
        if (wv::any(uw_fold_float<ML, true, false>(M, S, D, lane, f.cur_s, f.cur_b, f.nxt_s, f.nxt_b) != 0u)) {
          f.cur_s = s0; f.cur_b = b0; f.nxt_s = ninf; f.nxt_b = kUwUnreached;
          f.tie_credit = f.tie_credit + 8u > 64u ? 64u : f.tie_credit + 8u;
          ties = true;
        } else if (f.tie_credit != 0u) {
          --f.tie_credit;
        }
      } else if (all) {
        --f.tie_credit;
      }
      if (ties) {
        if (all) (void)uw_fold_float<ML, true, true>(M, S, D, lane, f.cur_s, f.cur_b, f.nxt_s, f.nxt_b);
        else (void)uw_fold_float<ML, false, true>(M, S, D, lane, f.cur_s, f.cur_b, f.nxt_s, f.nxt_b);
      }
    }
    else uw_fold_exact<ML>(d, M, S, lane, f.cur_s, f.cur_b, f.nxt_s, f.nxt_b);
  }
  *cyc += wv::clock() - t0;
}

// Positions [c, c + 64) are final: one coalesced row to HBM; what reached beyond the chunk becomes the next chunk's start;
// the matrix is "none" again (deep: what the walker said of this lane's row; nothing to do if no row holds anything).
template <uint32_t ML>
SPMX_DEVICE void uw_flush_chunk(int c, int nlen, int32_t *bid, uint16_t *blen, U2 *cands, uint32_t deep, int lane, UwFolder &f) {
  const int p = c + lane;
  if (p <= nlen) {
    bid[p] = f.cur_b == kUwUnreached ? 0 : static_cast<int32_t>(f.cur_b & 0x00FFFFFFu);
    blen[p] = f.cur_b == kUwUnreached ? static_cast<uint16_t>(0) : static_cast<uint16_t>((f.cur_b >> 24) & 0x7Fu);
  }
  f.cur_s = f.nxt_s; f.cur_b = f.nxt_b;
  f.nxt_s = -__builtin_inff(); f.nxt_b = kUwUnreached;
  wv::sync();                                                        // (every lane has read the matrix)
  // the matrix "none" again, 64 consecutive entries a store (a lane clearing its OWN row writes at a stride of ML * 8 bytes:
  // every lane on the same LDS banks -- 57 % of the kernel's LDS cycles were bank conflicts, profiles/r06_docs_16k_pmc_sq.txt)
  if (wv::any(deep != 0u)) {
#pragma unroll
    for (uint32_t k = 0; k < ML; ++k) cands[k * 64u + static_cast<uint32_t>(lane)] = U2{kUwNone, kUwNan};
  }
}

// The backtrack (:1010-1018) through windows of 256 positions staged in LDS (`ws`: 4 * kUwRing words): on return blen[e] has
// kTokEnd | length at every token end of the best path.  The chain from the window's last token end is not walked token by
// token (a dependent LDS read each: 350 cycles a token, a sixth of a megabyte document's time in round 6's first profile)
// but MARKED by pointer doubling: round k marks what the marked positions' 2^k-th predecessors are and squares the
// predecessor map -- eight rounds for 256 positions, whatever the number of tokens.  A position whose token begins at or
// below the window's first position ends the window: the next one begins there.  false: a broken chain (cannot happen).
SPMX_DEVICE bool uw_backtrack(uint32_t *ws, uint16_t *blen, int nlen, int lane) {
  uint32_t ok = 1;
  {
    uint32_t *len_w = ws, *j_a = len_w + kUwRing, *j_b = j_a + kUwRing, *mk = j_b + kUwRing;
    int e = nlen;                                                     // wave-uniform
    while (e > 0) {
      const int lo = e > static_cast<int>(kUwRing) - 1 ? e - (static_cast<int>(kUwRing) - 1) : 0;   // window [lo, e]
      const int W = e - lo + 1;
      wv::sync();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, p = lo + i;
        uint32_t len = 0;
        if (i < W) len = blen[p] & 0x7FFFu;
        len_w[i] = len;
        const bool inner = i < W && len != 0u && static_cast<int>(len) < p - lo;       // its token begins inside the window
        j_a[i] = inner ? static_cast<uint32_t>(i) - len : static_cast<uint32_t>(i);
        mk[i] = i == W - 1 ? 1u : 0u;
      }
      wv::sync();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t *src = (r & 1) ? j_b : j_a;
        uint32_t *dst = (r & 1) ? j_a : j_b;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = lane + 64 * k;
          if (mk[i]) mk[src[i]] = 1u;
        }
        wv::sync();
        if (r < 7) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            dst[i] = src[src[i]];
          }
          wv::sync();
        }
      }
      int e2 = e;
      bool bad = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, p = lo + i;
        int next = -1;
        if (i < W && mk[i]) {                                         // a token end of the best path
          const int len = static_cast<int>(len_w[i]);
          if (len == 0 || len > p) bad = true;                        // (unreached / a length beyond the text: cannot happen)
          else {
            blen[p] = static_cast<uint16_t>(kTokEnd | static_cast<uint32_t>(len));
            if (p - len <= lo) next = p - len;
          }
        }
        const uint64_t m = wv::ballot(next >= 0);
        if (m) e2 = wv::shfl(next, wv::ffs64(m) - 1);
      }
      if (wv::any(bad) || e2 >= e) { ok = 0; break; }                 // broken chain / no progress
      e = e2;
    }
  }
  ok = wv::shfl(ok, 0);
  return ok != 0;
}

// EncodeOptimized of the normalized text nt[0, nlen) (device form, HBM) by one wavefront.  bid / blen: nlen + 2
// entries in HBM, written here.  On return blen[e] has kTokEnd | length at every token end of the best path and bid[e]
// the token's id (unk_id for an unknown character).  false: a broken chain (cannot happen).  cyc: [0] += the cycles of
// the walks, [1] += of the folds.
template <uint32_t ML>
SPMX_DEVICE bool unigram_wave(const SpmxDev &d, const uint8_t *nt, int nlen, const UniWaveLds &T, int32_t *bid, uint16_t *blen,
                              int lane, unsigned long long *cyc) {
  for (uint32_t k = static_cast<uint32_t>(lane); k < UniWaveMatrixEntries(ML); k += 64u) T.cands[static_cast<int>(k) - 2] = U2{kUwNone, kUwNan};
  UwWalker w;
  UwFolder f;
  uw_walker_begin(w, nt, nlen, lane);
  uw_folder_begin(f, lane);
  for (int c = 0; c <= nlen; c += 64) {                              // (c == nlen: only position nlen is left to store)
    uint64_t S;
    uint32_t D, deep;
    uw_walk_chunk<ML>(d, nt, nlen, c, T.win, T.cands, lane, w, &S, &D, &deep, &cyc[0]);
    wv::sync();
    uw_fold_chunk<ML>(d, T.cands, S, D, lane, f, &cyc[1]);
    uw_flush_chunk<ML>(c, nlen, bid, blen, T.cands, deep, lane, f);
  }
  wv::sync_global();
  return uw_backtrack(T.ring_b, blen, nlen, lane);                    // (over the matrix, dead now)
}


// One sentence per wavefront over a device-side list; slices from the long form's pool (LongArgs, kernels_long.h).
template <uint32_t ML>
SPMX_DEVICE void uni_long_block(const LongArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const UniWaveLds T = carve_uniwave(smem, ML);
  const uint32_t wave_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block());
  const uint32_t n_waves = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block());
  const uint32_t count = *a.list_count;
  const int n_extra = d.n_prefix + d.n_suffix;
  unsigned long long st_sent = 0, st_raw = 0, st_ids = 0, st_cyc[3] = {0, 0, 0}, st_seg[2] = {0, 0};
  for (uint32_t i = wave_id; i < count; i += n_waves) {
    const unsigned long long t0 = wv::clock();
    const uint32_t sid = a.list[i];
    const uint64_t beg = a.offs[sid];
    const uint64_t L64 = a.offs[sid + 1] - beg;
    if (L64 > 0x7FFFFFF0ull / (d.expand_max ? d.expand_max : 1u)) {   // its normalized form could pass 2^31 bytes
      if (lane == 0) {
        a.counts[sid] = 0; a.tmp_off[sid] = 0; a.sent_status[sid] = static_cast<uint8_t>(kSsOutOfRange);
        wv::atomic_add(&a.side->n_failed, 1ull);
      }
      continue;
    }
    const int L = static_cast<int>(L64);
    auto empty_out = [&]() {                                          // (empty, or nothing but whitespace)
      if (lane == 0) {
        const unsigned long long at = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(n_extra));
        a.tmp_off[sid] = at;
        a.counts[sid] = static_cast<uint32_t>(n_extra);
        if (at + static_cast<unsigned long long>(n_extra) > a.arena_cap) wv::atomic_or(a.status, kStArenaOverflow);
        else {
          for (int x = 0; x < d.n_prefix; ++x) a.arena[at + x] = d.prefix_ids[x];
          for (int x = 0; x < d.n_suffix; ++x) a.arena[at + d.n_prefix + x] = d.suffix_ids[x];
        }
      }
    };
    // a slice for a normalized form of up to `cap` bytes; false: the pool is exhausted (the sentence is on the retry list)
    uint8_t *norm = nullptr;
    int32_t *bid = nullptr;
    uint16_t *blen = nullptr;
    auto take_slice = [&](uint64_t cap) -> bool {
      const uint64_t b_text = Align16(cap + kUwWindow + 16);
      const uint64_t b_bid = Align16((cap + 2) * 4);
      const uint64_t b_len = Align16((cap + 2) * 2);
      const uint64_t need = b_text + b_bid + b_len;
      unsigned long long at = 0;
      if (lane == 0) at = wv::atomic_add(a.pool_head, static_cast<unsigned long long>(need));
      at = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(at >> 32), 0)) << 32) | wv::shfl(static_cast<uint32_t>(at), 0);
      if (at + need > a.pool_cap) {                                   // the host grows the pool and launches again
        if (lane == 0) { a.retry_list[wv::atomic_add(a.retry_count, 1u)] = sid; a.counts[sid] = 0u; }
        return false;
      }
      norm = a.pool + at;
      bid = reinterpret_cast<int32_t *>(norm + b_text);
      blen = reinterpret_cast<uint16_t *>(norm + b_text + b_bid);
      return true;
    };
    // ---- normalize: the position-parallel normalizer (kernels.h normalize_wave) straight from / to HBM, into a slice
    // sized for text that does not grow (ASCII, CJK: 1.5 x + 64; three-byte space symbols: 3 x); the rare sentence
    // that outgrows it (NFKC expansions) is normalized again by one lane with norm_lane_any -- a count-only pass, a
    // slice of exactly the size it takes, a write pass ----
    int nlen = 0;
    if (L > 0) {
      const bool esc3 = (d.flags & kNfEscapeWs) && !(d.flags & kNfCompressSp);
      uint64_t cap1 = esc3 ? 3ull * static_cast<uint64_t>(L) + 64u : static_cast<uint64_t>(L) + static_cast<uint64_t>(L) / 2u + 64u;
      const uint64_t bound = static_cast<uint64_t>(L) * d.expand_max + 16u;
      if (cap1 > bound) cap1 = bound;
      if (!take_slice(cap1)) continue;
      nlen = normalize_wave<true>(d, a.text + beg, L, norm, static_cast<int>(cap1), lane);
      if (nlen < 0) {
        int n2 = 0;
        if (lane == 0) {
          int nsp = 0;
          FlatSink cs{nullptr, nullptr, 0};
          n2 = norm_lane_any(d, a.text, beg, L, cs, T.rawwin, &nsp);
        }
        n2 = wv::shfl(n2, 0);
        if (n2 > 0) {
          if (!take_slice(static_cast<uint64_t>(n2))) continue;
          if (lane == 0) {
            FlatSink ws{norm, nullptr, n2};
            int nsp2 = 0;
            norm_lane_any(d, a.text, beg, L, ws, T.rawwin, &nsp2);
          }
        }
        nlen = n2;
      }
    }
    if (nlen == 0) { empty_out(); ++st_sent; st_raw += static_cast<unsigned long long>(L); st_ids += static_cast<unsigned long long>(n_extra); continue; }
    const unsigned long long t1 = wv::clock();
    if (lane == 0) { blen[0] = 0; bid[0] = 0; }
    wv::sync_global();
    const bool ok = unigram_wave<ML>(d, norm, nlen, T, bid, blen, lane, st_seg);
    wv::sync_global();
    if (!ok) {                                                        // "all normalized characters are not consumed."
      if (lane == 0) {
        a.counts[sid] = 0; a.tmp_off[sid] = 0; a.sent_status[sid] = static_cast<uint8_t>(kSsInternal);
        wv::atomic_add(&a.side->n_failed, 1ull);
      }
      continue;
    }
    const unsigned long long t2 = wv::clock();
    const int n_out = emit_wave(a, sid, norm, nlen, bid, blen, lane);
    ++st_sent; st_raw += static_cast<unsigned long long>(L); st_ids += static_cast<unsigned long long>(n_out);
    st_cyc[0] += t1 - t0; st_cyc[1] += t2 - t1; st_cyc[2] += wv::clock() - t2;
  }
  if (a.stats && lane == 0 && st_sent) {
    wv::atomic_add(&a.stats[0], st_sent);
    wv::atomic_add(&a.stats[1], st_raw);
    wv::atomic_add(&a.stats[2], st_ids);
    wv::atomic_add(&a.stats[3], st_seg[0]);       // (of the segment cycles) the walks
    wv::atomic_add(&a.stats[7], st_seg[1]);       // (of the segment cycles) the fold
    wv::atomic_add(&a.stats[4], st_cyc[0]);       // normalize
    wv::atomic_add(&a.stats[5], st_cyc[1]);       // segment
    wv::atomic_add(&a.stats[6], st_cyc[2]);       // emit
  }
}

// ---- a document per WORKGROUP OF TWO wavefronts (round 6): wavefront 0 normalizes and then WALKS, a pair of chunks ahead of
// wavefront 1, which FOLDS, stores, and at the end backtracks and emits.  The hand-over is the matrix (four of them: two
// buffers of a pair of chunks) and three words beside each (the chunk's character starts, its longest piece, every row's
// depth); one workgroup barrier per pair of chunks; the walker walks both chunks of its pair in ONE loop (uw_walk_pair).  For batches of FEW long documents: a document that has a CU to itself spends 130 + 40 cycles a
// byte in the walker's half and 130 + 30 in the folder's, and this form runs them side by side.  (A batch of thousands of
// documents fills the chip with one wavefront each -- uni_long_block -- and gains nothing from pairs.) ----
struct UniPipeHdr {            // wavefront 0 -> wavefront 1, per document
  unsigned long long norm, bid, blen;
  int32_t nlen;
  uint32_t go;                 // 1: fold it; 0: the document is done with (empty, failed, on the retry list)
};
struct UniPipeMeta {           // per buffer and chunk of the pair
  uint32_t s_lo, s_hi, d, pad;
  uint8_t deep[64];
};
SPMX_HD constexpr uint32_t UniPipeLdsBytes(uint32_t ML) {
  return 4u * UniWaveMatrixEntries(ML) * 8u + kUwWindow + 16u + ((kRawWinBytes + 15u) & ~15u) + 64u + 4u * 96u;
}

template <uint32_t ML>
SPMX_DEVICE void uni_long_pipe_block(const LongArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const bool walker = wv::wave_in_block() == 0;
  const SpmxDev &d = a.dev;
  constexpr uint32_t kM = UniWaveMatrixEntries(ML) * 8u;
  // entry [0][0] of each matrix: [buffer][chunk of the pair]
  U2 *cands[2][2] = {{reinterpret_cast<U2 *>(smem) + 2, reinterpret_cast<U2 *>(smem + kM) + 2},
                     {reinterpret_cast<U2 *>(smem + 2u * kM) + 2, reinterpret_cast<U2 *>(smem + 3u * kM) + 2}};
  uint8_t *win = smem + 4u * kM;
  uint8_t *rawwin = win + kUwWindow + 16u;
  UniPipeHdr *hdr = reinterpret_cast<UniPipeHdr *>(rawwin + ((kRawWinBytes + 15u) & ~15u));
  UniPipeMeta *meta = reinterpret_cast<UniPipeMeta *>(reinterpret_cast<unsigned char *>(hdr) + 64);   // [buffer][chunk of the pair], 96 bytes apart
  auto meta_of = [&](int k, int hh) -> UniPipeMeta * { return reinterpret_cast<UniPipeMeta *>(reinterpret_cast<unsigned char *>(meta) + 96 * (2 * (k & 1) + hh)); };
  const uint32_t count = *a.list_count;
  const int n_extra = d.n_prefix + d.n_suffix;
  unsigned long long st_sent = 0, st_raw = 0, st_ids = 0, st_cyc[3] = {0, 0, 0}, st_seg[2] = {0, 0};
  for (uint32_t i = static_cast<uint32_t>(wv::block_id()); i < count; i += static_cast<uint32_t>(wv::grid_size())) {
    const unsigned long long t0 = wv::clock();
    const uint32_t sid = a.list[i];
    const uint64_t beg = a.offs[sid];
    const uint64_t L64 = a.offs[sid + 1] - beg;
    const int L = static_cast<int>(L64 > 0x7FFFFFF0ull ? 0x7FFFFFF0ull : L64);
    if (walker) {
      // ---- wavefront 0: the slice and the normalized text (as uni_long_block) ----
      uint32_t go = 0;
      uint8_t *norm = nullptr;
      int32_t *bid = nullptr;
      uint16_t *blen = nullptr;
      int nlen = 0;
      auto take_slice = [&](uint64_t cap) -> bool {
        const uint64_t b_text = Align16(cap + kUwWindow + 16);
        const uint64_t b_bid = Align16((cap + 2) * 4);
        const uint64_t b_len = Align16((cap + 2) * 2);
        const uint64_t need = b_text + b_bid + b_len;
        unsigned long long at = 0;
        if (lane == 0) at = wv::atomic_add(a.pool_head, static_cast<unsigned long long>(need));
        at = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(at >> 32), 0)) << 32) | wv::shfl(static_cast<uint32_t>(at), 0);
        if (at + need > a.pool_cap) {                                 // the host grows the pool and launches again
          if (lane == 0) { a.retry_list[wv::atomic_add(a.retry_count, 1u)] = sid; a.counts[sid] = 0u; }
          return false;
        }
        norm = a.pool + at;
        bid = reinterpret_cast<int32_t *>(norm + b_text);
        blen = reinterpret_cast<uint16_t *>(norm + b_text + b_bid);
        return true;
      };
      bool done = false;                                              // nothing for the folder: failed, on the retry list, empty
      if (L64 > 0x7FFFFFF0ull / (d.expand_max ? d.expand_max : 1u)) {   // its normalized form could pass 2^31 bytes
        if (lane == 0) {
          a.counts[sid] = 0; a.tmp_off[sid] = 0; a.sent_status[sid] = static_cast<uint8_t>(kSsOutOfRange);
          wv::atomic_add(&a.side->n_failed, 1ull);
        }
        done = true;
      } else if (L > 0) {
        const bool esc3 = (d.flags & kNfEscapeWs) && !(d.flags & kNfCompressSp);
        uint64_t cap1 = esc3 ? 3ull * static_cast<uint64_t>(L) + 64u : static_cast<uint64_t>(L) + static_cast<uint64_t>(L) / 2u + 64u;
        const uint64_t bound = static_cast<uint64_t>(L) * d.expand_max + 16u;
        if (cap1 > bound) cap1 = bound;
        if (!take_slice(cap1)) done = true;
        else {
          nlen = normalize_wave<true>(d, a.text + beg, L, norm, static_cast<int>(cap1), lane);
          if (nlen < 0) {                                             // it outgrew the slice: count, a slice of that size, write
            int n2 = 0;
            if (lane == 0) {
              int nsp = 0;
              FlatSink cs{nullptr, nullptr, 0};
              n2 = norm_lane_any(d, a.text, beg, L, cs, rawwin, &nsp);
            }
            n2 = wv::shfl(n2, 0);
            if (n2 > 0) {
              if (!take_slice(static_cast<uint64_t>(n2))) done = true;
              else if (lane == 0) {
                FlatSink ws{norm, nullptr, n2};
                int nsp2 = 0;
                norm_lane_any(d, a.text, beg, L, ws, rawwin, &nsp2);
              }
            }
            nlen = n2;
          }
        }
      }
      if (!done && nlen == 0) {                                       // (empty, or nothing but whitespace)
        if (lane == 0) {
          const unsigned long long at = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(n_extra));
          a.tmp_off[sid] = at;
          a.counts[sid] = static_cast<uint32_t>(n_extra);
          if (at + static_cast<unsigned long long>(n_extra) > a.arena_cap) wv::atomic_or(a.status, kStArenaOverflow);
          else {
            for (int x = 0; x < d.n_prefix; ++x) a.arena[at + x] = d.prefix_ids[x];
            for (int x = 0; x < d.n_suffix; ++x) a.arena[at + d.n_prefix + x] = d.suffix_ids[x];
          }
        }
        ++st_sent; st_raw += static_cast<unsigned long long>(L); st_ids += static_cast<unsigned long long>(n_extra);
        done = true;
      }
      go = done ? 0u : 1u;
      if (lane == 0) {
        if (go) { blen[0] = 0; bid[0] = 0; }
        hdr->norm = reinterpret_cast<unsigned long long>(norm); hdr->bid = reinterpret_cast<unsigned long long>(bid);
        hdr->blen = reinterpret_cast<unsigned long long>(blen); hdr->nlen = nlen; hdr->go = go;
      }
      wv::sync_global();                                              // (the normalized text is in HBM before the walks read it)
      st_cyc[0] += wv::clock() - t0;
    } else {
      // ---- wavefront 1 meanwhile: the four matrices "none" ----
      for (uint32_t k = static_cast<uint32_t>(lane); k < 4u * UniWaveMatrixEntries(ML); k += 64u) reinterpret_cast<U2 *>(smem)[k] = U2{kUwNone, kUwNan};
    }
    wv::block_sync();
    const uint32_t go = hdr->go;
    const int nlen = hdr->nlen;
    const uint8_t *norm = reinterpret_cast<const uint8_t *>(hdr->norm);
    int32_t *bid = reinterpret_cast<int32_t *>(hdr->bid);
    uint16_t *blen = reinterpret_cast<uint16_t *>(hdr->blen);
    if (go) {
      // ---- the chunks, a PAIR per iteration: iteration k walks chunks 2k, 2k + 1 and folds chunks 2k - 2, 2k - 1 ----
      const unsigned long long t1 = wv::clock();
      const int n_pairs = (nlen / 64 + 2) / 2;                        // (chunks: nlen / 64 + 1; a chunk beyond the text is nothing)
      UwWalker w;
      UwFolder f;
      if (walker) uw_walker_begin(w, norm, nlen, lane); else uw_folder_begin(f, lane);
      for (int k = 0; k <= n_pairs; ++k) {
        if (walker) {
          if (k < n_pairs) {
            uint64_t S[2];
            uint32_t D[2], deep[2];
            uw_walk_pair<ML>(d, norm, nlen, 128 * k, win, cands[k & 1][0], cands[k & 1][1], lane, w, S, D, deep, &st_seg[0]);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              UniPipeMeta *m = meta_of(k, hh);
              m->deep[lane] = static_cast<uint8_t>(deep[hh]);
              if (lane == 0) { m->s_lo = static_cast<uint32_t>(S[hh]); m->s_hi = static_cast<uint32_t>(S[hh] >> 32); m->d = D[hh]; }
            }
          }
        } else if (k >= 1) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const UniPipeMeta *m = meta_of(k - 1, hh);
            const uint64_t S = wv::uniform64(static_cast<uint64_t>(m->s_hi) << 32 | m->s_lo);
            const uint32_t D = wv::uniform(m->d);
            const uint32_t deep = m->deep[lane];
            const int c = 128 * (k - 1) + 64 * hh;
            if (c <= nlen) {
              uw_fold_chunk<ML>(d, cands[(k - 1) & 1][hh], S, D, lane, f, &st_seg[1]);
              uw_flush_chunk<ML>(c, nlen, bid, blen, cands[(k - 1) & 1][hh], deep, lane, f);
            }
          }
        }
        wv::block_sync();
      }
      if (!walker) {
        wv::sync_global();
        const bool ok = uw_backtrack(reinterpret_cast<uint32_t *>(smem), blen, nlen, lane);      // (over the first matrix, dead now)
        wv::sync_global();
        const unsigned long long t2 = wv::clock();
        if (!ok) {                                                    // "all normalized characters are not consumed."
          if (lane == 0) {
            a.counts[sid] = 0; a.tmp_off[sid] = 0; a.sent_status[sid] = static_cast<uint8_t>(kSsInternal);
            wv::atomic_add(&a.side->n_failed, 1ull);
          }
        } else {
          const int n_out = emit_wave(a, sid, norm, nlen, bid, blen, lane);
          ++st_sent; st_raw += static_cast<unsigned long long>(L); st_ids += static_cast<unsigned long long>(n_out);
          st_cyc[1] += t2 - t1; st_cyc[2] += wv::clock() - t2;
        }
      }
    }
    wv::block_sync();                                                 // (the next document's header and matrices are this one's)
  }
  // (one wavefront after the other: both add to the same three counters)
  for (int turn = 0; turn < 2; ++turn) {
    if (a.stats && lane == 0 && turn == (walker ? 0 : 1)) {
      if (st_sent) {
        wv::atomic_add(&a.stats[0], st_sent);
        wv::atomic_add(&a.stats[1], st_raw);
        wv::atomic_add(&a.stats[2], st_ids);
      }
      if (walker) {
        wv::atomic_add(&a.stats[3], st_seg[0]);       // the walks
        wv::atomic_add(&a.stats[4], st_cyc[0]);       // normalize
      } else {
        wv::atomic_add(&a.stats[7], st_seg[1]);       // the fold
        wv::atomic_add(&a.stats[5], st_cyc[1]);       // segment: the chunks' pipeline and the backtrack, by the folder's clock
        wv::atomic_add(&a.stats[6], st_cyc[2]);       // emit
      }
    }
    wv::block_sync();
  }
}

}  // namespace spmx
#endif
