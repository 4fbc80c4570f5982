// Word form, WORD PER LANE (round 5): the word rounds of kernels_word.h with the lanes of a wavefront on the WORDS of a
// few sentences instead of on 64 different sentences.
// Reference: unigram::Model::EncodeOptimized (src/unigram_model.cc:889-1020) / bpe::Model::SampleEncode (src/bpe_model.cc:
// 38-203) behind Normalizer::Normalize (src/normalizer.cc:71-186), one WORD at a time through the word memo -- the legality
// argument, the margin guard and the tables are those of kernels_word.h (top of that file); nothing about WHAT is computed
// changes here, only which lane does it.
//
// Why.  The sentence-per-lane loop (uni_word_lane) pays 4 - 5 vector memory instructions per iteration with 64 DIFFERENT
// addresses each (a text window, a memo probe, two id bursts per lane, every lane in its own sentence); its cost is the
// address unit's, by the lane, and every 128-byte line of text is fetched about three times
// because a lane walks its line over ~28 iterations while the XCD's L2 turns over.  Here
//
//   units   the tile's sentences are cut into the 16-byte units (aligned in memory) that hold their bytes; a ROUND is 64
//           consecutive units, one per lane: the text of ~7 sentences arrives as one load instruction over ~8 cache
//           lines, each byte of the batch exactly once, two rounds ahead of its use, and is kept in a 4 KB LDS ring
//           together with a 16-bit mask of its spaces per unit;
//   starts  a lane finds the word starts of its unit (a byte that is not 0x20 behind a 0x20 or at the sentence's start:
//           SWAR space flags, the neighbour's last byte by a DPP shift) and appends them to a QUEUE of word records in LDS
//           at the place a wave prefix sum of the counts names -- in text order;
//   words   64 records at a time whatever round they came from (the queue keeps every lane busy), a WORD per lane: its
//           length from the space masks, its 16-byte window from the ring (unaligned LDS reads), then the memo: the LDS
//           table of the likeliest words, else ONE probe of `uall` (dev.h: every word of the memo in one table), the
//           call-local memo in its two modes;
//   stages  a software pipeline of two stages: a batch's probe is in flight while the wavefront finishes the batch before
//           it and prepares the one behind it;
//   margin  what chains the words of a sentence in the sentence-per-lane loop is only the running bound of |score| (a sum
//           of per-word constants) and the number of ids written before the word: both are SEGMENTED wave prefix sums (one
//           add scan over the packed pair, one max scan that hands every lane its sentence's first lane's value, a scalar
//           carry for the sentence that continues from the previous 64 words) -- exact integers, compared with the entry's
//           limit; so is the unknown-piece run across words (the neighbour lane's flag);
//   ids     consecutive words are consecutive ids: a word's ids go straight to its sentence's slot at the offset the scan
//           gave it (neighbouring lanes write neighbouring addresses).
//
// A sentence either gets the reference's ids or nothing (status bits per sentence in LDS: "again" = every word it lacks is
// in the call-local memo now, "gone" = not for the word form); the lists, the resume record, the arena slots, tmp_off and
// counts are exactly what encode_word_block_as leaves, so resolve, the other round, the tail launch, scan and compact do
// not care which of the two forms ran a round.  (The second round in this form starts its sentences over: a lane's work
// is a word, not the rest of a sentence.)
#ifndef SPMX_KERNELS_WORDWAVE_H_
#define SPMX_KERNELS_WORDWAVE_H_

namespace spmx {

constexpr uint32_t kWwRing = 4096u;                 // text bytes a wave holds: four rounds of 64 units
constexpr uint32_t kWwRingPad = 32u;                // the ring's first bytes again behind its end: a window that starts in the last unit reads on
constexpr uint32_t kWwUnits = kWwRing / 16u;        // 256
constexpr uint32_t kWwMaskBytes = (kWwUnits + 4u) * 2u + 8u;   // a 16-bit space mask per unit (+ the first four again behind the end)
constexpr uint32_t kWwRecs = 128u + 512u;           // the word queue: what the last round left (< 128) + at most 8 starts per unit
constexpr uint32_t kWwMaxLen = 65536u;              // longer sentences are not taken (record positions have 26 bits)
constexpr uint32_t kWwPerWave = kWwRing + kWwRingPad + kWwMaskBytes + kWwRecs * 4u + 64u * 16u + 64u * 8u + 5u * 64u * 4u;
static_assert(kWwPerWave % 16u == 0u, "per-wave LDS blocks keep 16-byte alignment");
constexpr uint32_t kWwAgain = 1u, kWwGone = 2u;     // per-sentence status bits
constexpr uint32_t kWwKeyMaskBytes = 640u;          // 18 rows {key mask, padding under the mask's zeros} of 32 bytes (+ slack)
// + the LDS table of the likeliest words (kWordHotSlots entries of uhot2), the displacements of uall's perfect hash.
// The LDS table is the SECOND round's only (kWwHotIn): there a word it answers does not ask the call-local memo in HBM (64
// bytes a word).  The first round dropped it in round 6: a word the table lacks is answered by `uall` from L2, every lane
// issues that probe anyway, and the table's lookup was ~25 instructions per 64 words of an issue-bound loop -- 2048 / 1024 /
// 512 slots: 3.085 / 3.076 / 3.067 ms, none: 2 % faster still; with all words asking the call-local memo the second round
// is 12 % slower on open-vocabulary text (profiles/r06_hot_slots_waves.txt).  Same LDS layout for every mode.
constexpr uint32_t kWwShared = kWwKeyMaskBytes + kWordHotSlots * 16u + kUallBuckets * 2u;
SPMX_HD constexpr bool kWwHotIn(int mode) { return mode == 2; }      // (kWmDyn)
SPMX_HD inline uint32_t WordWaveLdsBytes(uint32_t waves) { return kWwShared + waves * kWwPerWave; }

struct __attribute__((packed, aligned(2))) U64H { uint32_t lo, hi; };   // 8 bytes at any 2-byte address

// 16-bit mask of the bytes of the 16-byte unit v that equal 0x20
SPMX_DEVICE uint32_t space_mask16(const Q4 &v) {
  auto m4 = [](uint32_t z) -> uint32_t { return (((z >> 7) & 0x01010101u) * 0x10204080u) >> 28; };
  return m4(space_flags(v.x)) | (m4(space_flags(v.y)) << 4) | (m4(space_flags(v.z)) << 8) | (m4(space_flags(v.w)) << 12);
}
// the low n bytes of `text`, the other bytes of the key dword padded with 0x20 (kernels_word.h key_dword); n may be <= 0 or >= 4
SPMX_DEVICE uint32_t key_dword_n(uint32_t text, int n) {
  const uint32_t m = n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : (1u << (8 * n)) - 1u);
  return key_dword(text, m);
}

// what a batch of 64 words carries from one pipeline stage to the next (registers)
struct WwStage {
  uint32_t cnt;                  // words of the batch (wave-uniform; 0: an empty slot of the pipeline)
  uint32_t P, j;                 // the word's position in the tile's unit stream, its sentence (lane of the tile)
  uint32_t L;                    // the word's length, 17 = more than 16 bytes; bit 8: found in the LDS table
  uint32_t k0, k1, k2, k3;       // its key
  uint32_t hz, hw;               // the LDS table's entry at the word's hash: second id, {id, limit exponent, bound share}
  uint32_t e0x, e0y, e0z, e0w, e1x, e1y, e1z, e1w;   // what `uall` holds at the word's hash: {key} {ids, bound share, limit}
  // the call-local memo at the word's hash (collecting / second round): the slot's tag; (second round) its key and state
  uint32_t tg_lo, tg_hi;
  uint32_t warm;                 // (wave-uniform) the tag was asked for in stage A
  uint32_t d0x, d0y, d0z, d0w, d1x, d1y, d1z, d1w, d2x, d2y, d2z, d2w;
};
constexpr uint32_t kWwHit16 = 0x100u;

template <int MODE, bool H16>
SPMX_DEVICE void encode_wordwave_block(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  Q4 *masks = reinterpret_cast<Q4 *>(smem);                         // [18][2] by word length: the key's bytes that are the word's; 0x20 in the others
  U4 *hot = reinterpret_cast<U4 *>(smem + kWwKeyMaskBytes);
  uint16_t *disp = reinterpret_cast<uint16_t *>(smem + kWwKeyMaskBytes + kWordHotSlots * 16u);
  unsigned char *mine_lds = smem + kWwShared + static_cast<uint32_t>(wv::wave_in_block()) * kWwPerWave;
  uint8_t *ring = mine_lds;
  uint16_t *smask = reinterpret_cast<uint16_t *>(mine_lds + kWwRing + kWwRingPad);
  uint32_t *recq = reinterpret_cast<uint32_t *>(mine_lds + kWwRing + kWwRingPad + kWwMaskBytes);
  U4 *sent = reinterpret_cast<U4 *>(recq + kWwRecs);          // per sentence of the tile: {start, end (positions in the tile's unit stream), arena slot}
  U2 *sent_ua = reinterpret_cast<U2 *>(sent + 64);            // ... the address of its first unit
  uint32_t *s_nids = reinterpret_cast<uint32_t *>(sent_ua + 64), *s_stat = s_nids + 64, *s_first = s_stat + 64;
  uint32_t *s_n0 = s_first + 64, *s_x0 = s_n0 + 64;            // (second round) ids / bound in front of where the sentence is taken up
  {   // the shared read-only table (every wave writes the same values: no workgroup barrier)
    if (lane < 18) {
      const uint32_t L = static_cast<uint32_t>(lane);
      auto m = [&](uint32_t i) -> uint32_t { return L >= 4u * i + 4u ? 0xFFFFFFFFu : (L <= 4u * i ? 0u : (1u << (8u * (L - 4u * i))) - 1u); };
      masks[2 * lane] = Q4{m(0), m(1), m(2), m(3)};
      masks[2 * lane + 1] = Q4{kWordKeyPad & ~m(0), kWordKeyPad & ~m(1), kWordKeyPad & ~m(2), kWordKeyPad & ~m(3)};
    }
    if (kWwHotIn(MODE)) for (uint32_t k = static_cast<uint32_t>(lane); k < kWordHotSlots; k += 64u) hot[k] = d.uhot2[k];
    for (uint32_t k = static_cast<uint32_t>(lane); k < kUallBuckets; k += 64u) disp[k] = d.udisp[k];
    wv::sync();
  }
  const U4 *__restrict__ uall = d.uall;
  const uint32_t mall = d.uall_mask;
  const bool perfect = d.uall_perfect != 0u;
  const int n_extra = d.n_prefix + d.n_suffix;
  const bool keep_ws = (d.flags & kNfRemoveExtraWs) == 0;
  const bool any_word = (d.flags & kNfWordLocalNorm) != 0;
  const uint64_t tbase = reinterpret_cast<uint64_t>(a.text);
  WaveCounters tc;
  WwStage SA{}, SB{};                                               // the two batches of the word pipeline (below)
  uint32_t dyn_warm = 0u;                                           // (collecting round) batches for which stage A still asks the call-local memo's tags
  for (;;) {
    const unsigned long long cs = wv::clock();
    uint32_t c = 0, first = 0, ucnt = 0, got = 0;
    if (lane == 0) got = next_tile(a, &c, &first, &ucnt) ? 1u : 0u;
    got = wv::shfl(got, 0);
    if (!got) break;
    c = wv::shfl(c, 0); first = wv::shfl(first, 0); ucnt = wv::shfl(ucnt, 0);
    const bool direct = a.direct != 0u;
    const uint32_t *list = a.lists ? a.lists + static_cast<uint64_t>(c) * a.n : nullptr;
    const bool have = static_cast<uint32_t>(lane) < ucnt;
    uint32_t sid = 0;
    uint64_t beg = 0, l64 = 0;
    if (have) {
      sid = list ? list[first + static_cast<uint32_t>(lane)] : first + static_cast<uint32_t>(lane);
      beg = a.offs[sid];
      l64 = a.offs[sid + 1] - beg;
    }
    uint32_t cl = c;                                 // the sentence's length class
    if (direct) {
      cl = a.n_classes - 1u;
      for (int k = static_cast<int>(a.n_classes) - 2; k >= 0; --k) if (l64 <= a.cls[k].rcap) cl = static_cast<uint32_t>(k);
    }
    // (a class marked `general` passes through: documents belong to the wave-cooperative form, kernels_uniwave.h)
    const bool mine = have && !a.cls[cl].general && l64 <= kWwMaxLen;
    // ---- a slot of cap ids in the arena per sentence, as encode_word_block_as lays them out ----
    const int cap = mine ? static_cast<int>(l64) + 1 : 0;
    // (round 6 tried half-size slots for 16-bit ids -- lines twice as full for the compaction: no change in the step, 5.43 ms
    // against 5.38: CompactKernel is bound by its 1.1 GB of 32-bit output, not by the sparse reads)
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
    // (second round) the sentence keeps the slot the first round gave it -- and the ids in it: it is taken up at its
    // first missing word (a.resume: where that word starts, the ids in front of it, the bound of |score| there)
    uint32_t p0 = 0u, n0 = 0u, x0 = 0u;
    if (MODE == kWmDyn && mine) {
      slot = a.arena + a.tmp_off[sid] + d.n_prefix;
      const U4 rs = a.resume[sid];
      const float bf0 = wv::bits_to_float(rs.z);
      if (rs.x < l64 && rs.y <= rs.x && bf0 >= 0.f && bf0 < 16777216.f) { p0 = rs.x; n0 = rs.y; x0 = static_cast<uint32_t>(bf0); }
    }
    const bool work = mine && !overflow;
    const uint32_t len = work ? static_cast<uint32_t>(l64) - p0 : 0u;
    beg += p0;
    // ---- the tile's unit stream: sentence j's units follow sentence j - 1's ----
    const uint32_t begmod = static_cast<uint32_t>((tbase + beg) & 15ull);
    const uint32_t nun = len ? (begmod + len + 15u) >> 4 : 0u;
    const uint32_t pu_incl = wv::scan_add(nun);
    const uint32_t pu = pu_incl - nun;                              // units before this sentence's (lanes without one: the total)
    const uint32_t U = wv::read_lane(pu_incl, 63);
    {
      const uint32_t start = pu * 16u + begmod;
      const uint64_t ua = tbase + beg - begmod;                    // address of the unit that holds the sentence's first byte
      const uint64_t so = static_cast<uint64_t>(slot - a.arena);
      sent[lane] = U4{start, start + len, static_cast<uint32_t>(so), static_cast<uint32_t>(so >> 32)};
      sent_ua[lane] = U2{static_cast<uint32_t>(ua), static_cast<uint32_t>(ua >> 32)};
      s_nids[lane] = n0;
      s_stat[lane] = 0u;
      s_first[lane] = 0xFFFFFFFFu;
      if (MODE == kWmDyn) { s_n0[lane] = n0; s_x0[lane] = x0; }
    }
    wv::sync();
    const unsigned long long c0 = wv::clock();
    tc.cyc[0] += c0 - cs;                                           // (the tile's start-up: queue, list, offsets, arena slot)
    const uint32_t rounds = (U + 63u) >> 6;
    // the unit of round q this lane loads: its sentence (a search over the unit prefix sums, cross-lane) and its text
    auto issue = [&](uint32_t q, Q4 *v, uint32_t *jq) __attribute__((always_inline)) {
      const uint32_t u = q * 64u + static_cast<uint32_t>(lane);
      const bool ok = u < U;
      int lo = 0;
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) {
        const uint32_t pv = wv::shfl(pu, lo + step);
        if (pv <= u) lo += step;
      }
      *jq = static_cast<uint32_t>(lo);
      const uint32_t s_start = sent[lo].x;
      const U2 a2 = sent_ua[lo];
      const uint64_t ua = (static_cast<uint64_t>(a2.y) << 32 | a2.x) + (static_cast<uint64_t>(u) * 16u - (s_start & ~15u));
      // (an offset from the kernel's text pointer, not a bare address: a global load, not a flat one)
      *v = ok ? *reinterpret_cast<const Q4 *>(a.text + static_cast<long long>(ua - tbase)) : Q4{0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
    };
    uint32_t prev_last = 0x20u;                                     // the last byte of the unit before lane 0's
    // the unit into the ring; returns the word starts among its bytes
    auto commit = [&](uint32_t q, const Q4 &v, uint32_t j) __attribute__((always_inline)) -> uint32_t {
      const uint32_t u = q * 64u + static_cast<uint32_t>(lane);
      const bool ok = u < U;
      const uint32_t ui = u & (kWwUnits - 1u);
      const uint32_t spm = space_mask16(v);
      *reinterpret_cast<Q4 *>(ring + ui * 16u) = v;
      smask[ui] = static_cast<uint16_t>(spm);
      if (ui < 4u) {
        smask[kWwUnits + ui] = static_cast<uint16_t>(spm);
        if (ui < 2u) *reinterpret_cast<Q4 *>(ring + kWwRing + ui * 16u) = v;
      }
      const U4 s = sent[j];
      const uint32_t last = v.w >> 24;
      const uint32_t prevb = wv::lane_up1(last, prev_last);
      prev_last = wv::read_lane(last, 63);
      const uint32_t ub = u * 16u;
      const bool firstu = s.x >= ub;                                // the sentence starts in this unit
      const uint32_t lo = firstu ? s.x - ub : 0u;
      const uint32_t hi = s.y - ub < 16u ? s.y - ub : 16u;
      const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);   // the unit's bytes that are the sentence's
      uint32_t before = (spm << 1) | ((!firstu && prevb == 0x20u) ? 1u : 0u);   // bytes behind a 0x20 ...
      if (firstu) before |= 1u << lo;                                            // ... or at the sentence's start
      if (keep_ws && ok) {
        // a model that KEEPS extra whitespace (kernels_word.h keep_ws): a leading, a doubled or a trailing space is a space
        // symbol next to another one in the normalized text -- not a word boundary: the sentence is not the word form's
        const uint32_t lead = firstu ? (spm >> lo) & 1u : 0u;
        const uint32_t dbl = spm & before & vm & ~(firstu ? 1u << lo : 0u);
        const uint32_t trail = (s.y - ub <= 16u) ? (spm >> (hi - 1u)) & 1u : 0u;
        if (lead | dbl | trail) wv::lds_atomic_or(&s_stat[j], kWwGone);
      }
      return ok ? (~spm & before & vm) : 0u;
    };

    // ---------------- the word pipeline: two batches of 64 words in flight ----------------
    SA.cnt = 0u; SB.cnt = 0u;
    // what the sentence that continues from the previous 64 words has behind it
    uint32_t carry_j = 0xFFFFFFFFu, carry_x = 0u, carry_n = 0u, carry_unk = 0u;
    unsigned long long steps = 0;

    // stage A: the words' records, windows, lengths, keys; the LDS table's answer; every other word asks `uall`.
    // qcnt = 0: an empty batch (nothing is read that matters; its loads keep the instruction stream the same)
    auto stageA = [&](uint32_t qbase, uint32_t qcnt, WwStage &S) __attribute__((always_inline)) {
      S.cnt = qcnt;
      const uint32_t l32 = static_cast<uint32_t>(lane);
      const bool valid = l32 < qcnt;
      const uint32_t rec = recq[qbase + (valid ? l32 : (qcnt ? qcnt - 1u : 0u))];   // (the lanes beyond the batch repeat its last word)
      const uint32_t P = rec & 0x03FFFFFFu, j = rec >> 26;
      const uint32_t s_end = sent[j].y;
      const Q4U w = *reinterpret_cast<const Q4U *>(ring + (P & (kWwRing - 1u)));
      const U64H sm = *reinterpret_cast<const U64H *>(smask + ((P >> 4) & (kWwUnits - 1u)));
      // the word ends at the first 0x20 from its start, or where the sentence does
      const uint64_t bits = (static_cast<uint64_t>(sm.hi) << 32 | sm.lo) >> (P & 15u);
      const uint32_t rem = s_end - P;
      uint32_t L = static_cast<uint32_t>(wv::ffs64(bits | (1ull << 17))) - 1u;    // 17: no 0x20 among the next 17 bytes
      if (L > rem) L = rem;
      S.P = P; S.j = j;
      const Q4 mk = masks[2u * (L < 16u ? L : 16u)], pd = masks[2u * (L < 16u ? L : 16u) + 1u];
      S.k0 = (w.x & mk.x) | pd.x; S.k1 = (w.y & mk.y) | pd.y; S.k2 = (w.z & mk.z) | pd.z; S.k3 = (w.w & mk.w) | pd.w;   // (kernels_word.h key_dword)
      const uint32_t h1 = HashWordKey(S.k0, S.k1, S.k2, S.k3), h2 = UallHash2(S.k0, S.k1, S.k2, S.k3, h1);
      const U4 e = kWwHotIn(MODE) ? hot[h1 & (kWordHotSlots - 1u)] : U4{0u, 0u, 0u, 0xFFFFFFFFu};   // (words of up to 12 bytes: k3 is all padding, as in the table's keys)
      const uint32_t dsp = disp[UallBucket(h2)];
      const uint32_t zmask = L <= 10u ? 0xFFFFu : 0xFFFFFFFFu, form = L <= 10u ? kMemo16TwoPiece : 0u;
      // (bitwise: no branches, every load above is asked for at once)
      const bool hit16 = kWwHotIn(MODE) & (valid & (L <= 12u)) & ((e.x == S.k0) & (e.y == S.k1)) & ((((e.z ^ S.k2) & zmask) == 0u) &
                         ((e.w & kMemo16TwoPiece) == form) & (e.w != 0xFFFFFFFFu));
      S.L = L | (hit16 ? kWwHit16 : 0u);
      S.hz = e.z; S.hw = e.w;
      // (every lane loads -- the ones that do not ask, slot 0: a load under a divergent `if` sits behind a branch, and the
      // wait for the PREVIOUS batch's probe could then not count on this one's two loads being younger)
      const uint32_t slp = perfect ? UallSlot(h1, h2, dsp, mall) : (h1 & mall);
      const uint32_t sl = (valid & !hit16 & (L <= 16u)) ? slp : 0u;
      const U4 e0 = uall[2u * sl], e1 = uall[2u * sl + 1u];
      S.e0x = e0.x; S.e0y = e0.y; S.e0z = e0.z; S.e0w = e0.w; S.e1x = e1.x; S.e1y = e1.y; S.e1z = e1.z; S.e1w = e1.w;
      if (MODE != kWmPlain) {
        // the call-local memo is asked at the same time (most words that come this far are in `uall`: what this brings is
        // then not looked at -- but a word that needs it would otherwise wait a second round trip, and its wavefront with it).
        // Collecting round: only while the wavefront keeps meeting words the memo lacks (dyn_warm: set by stage B, counted
        // down here) -- on text the load-time memo fits, the slot's tag is fetched when a word needs it.
        const bool pre = MODE == kWmDyn || dyn_warm != 0u;
        if (MODE == kWmCollect && dyn_warm != 0u) --dyn_warm;
        const uint32_t dsl = (pre & valid & !hit16 & (L <= 16u)) ? (h1 & a.dyn_mask) : 0u;
        const unsigned long long g = MODE == kWmCollect ? wv::atomic_load64(&a.dyn_tag[dsl]) : a.dyn_tag[dsl];
        S.tg_lo = static_cast<uint32_t>(g); S.tg_hi = static_cast<uint32_t>(g >> 32);
        S.warm = pre ? 1u : 0u;
        if (MODE == kWmDyn) {
          const U4 d0 = a.dyn_ent[4u * dsl], d1 = a.dyn_ent[4u * dsl + 1u], d2 = a.dyn_ent[4u * dsl + 2u];
          S.d0x = d0.x; S.d0y = d0.y; S.d0z = d0.z; S.d0w = d0.w; S.d1x = d1.x; S.d1y = d1.y; S.d1z = d1.z; S.d1w = d1.w;
          S.d2x = d2.x; S.d2y = d2.y; S.d2z = d2.z; S.d2w = d2.w;
        }
      }
    };
    // stage B: the probe's answer; the margin, the ids, the sentence's state
    auto stageB = [&](WwStage &S) __attribute__((always_inline)) {
      if (S.cnt == 0u) return;
      ++steps;
      const bool word = static_cast<uint32_t>(lane) < S.cnt;
      const uint32_t k0 = S.k0, k1 = S.k1, k2 = S.k2, k3 = S.k3, j = S.j, P = S.P;
      const bool lng = (S.L & 0xFFu) > 16u;                          // a word of more than 16 bytes
      const bool hit16 = kWwHotIn(MODE) && (S.L & kWwHit16) != 0u;
      const bool probe = word && !hit16 && !lng;
      U4 e0{S.e0x, S.e0y, S.e0z, S.e0w}, e1{S.e1x, S.e1y, S.e1z, S.e1w};
      bool hit32 = probe & ((e0.x == k0) & (e0.y == k1)) & ((e0.z == k2) & (e0.w == k3)) & (e1.x != 0xFFFFFFFFu);
      // (a perfect hash: the word sits there or nowhere; else open addressing: a collision walks on)
      bool walk = !perfect && probe && !hit32 && e1.x != 0xFFFFFFFFu;
      if (wv::any(walk)) {
        uint32_t sl = HashWordKey(k0, k1, k2, k3) & mall;
        while (wv::any(walk)) {
          ++tc.cyc[1];                                               // (profiling: steps of collision walks)
          if (walk) {
            sl = (sl + 1u) & mall;
            e0 = uall[2u * sl];
            e1 = uall[2u * sl + 1u];
            hit32 = e0.x == k0 && e0.y == k1 && e0.z == k2 && e0.w == k3 && e1.x != 0xFFFFFFFFu;
            walk = !hit32 && e1.x != 0xFFFFFFFFu;
          }
        }
      }
      // ---- (second round) the call-local memo: words collected by the first round, segmented by word_resolve_block ----
      bool hitd = false;
      uint32_t dn = 0;
      U4 dia{0, 0, 0, 0}, dib{0, 0, 0, 0};
      uint32_t dbound = 0u, dlim = 0u;
      if (MODE == kWmDyn && wv::any(probe & !hit32)) {
        if (probe & !hit32) {
          const unsigned long long tag = DynTag(k0, k1, k2, k3);
          // the slot the word's hash names came with the probe (stage A); the ids are asked for now; another slot (a
          // collision in the table) is walked to -- rare: the table is sparse
          uint32_t sl = static_cast<uint32_t>(tag >> 32) & a.dyn_mask;
          unsigned long long g = static_cast<unsigned long long>(S.tg_hi) << 32 | S.tg_lo;
          U4 d0{S.d0x, S.d0y, S.d0z, S.d0w}, d1{S.d1x, S.d1y, S.d1z, S.d1w}, d2{S.d2x, S.d2y, S.d2z, S.d2w};
          for (uint32_t t = 0; t < kDynProbes; ++t) {
            if (g == 0ull) break;
            if (g == tag) {
              if (d0.x == k0 && d0.y == k1 && d0.z == k2 && d0.w == k3 && d1.x == 1u) {
                hitd = true; dn = d1.y; dbound = d1.z; dlim = d1.w;
                dia = d2;
                // (the second half of the ids: only a word of more than four pieces -- or more than eight under kDynWide -- has it)
                const uint32_t cn = d1.y & 0xFFu;
                if ((d1.y & kDynWide) ? cn > 8u : cn > 4u) dib = a.dyn_ent[4u * sl + 3u];
              }
              break;                                     // (same hash, other bytes or an unusable word: a miss)
            }
            sl = (sl + 1u) & a.dyn_mask;
            g = a.dyn_tag[sl];
            d0 = a.dyn_ent[4u * sl]; d1 = a.dyn_ent[4u * sl + 1u]; d2 = a.dyn_ent[4u * sl + 2u];
          }
        }
      }
      const bool hit = hit16 || hit32 || hitd;
      bool gone = word && lng;
      bool miss = false;
      if (MODE == kWmCollect) {
        if (wv::any(probe && !hit32)) {
          dyn_warm = 16u;
          if (probe && !hit32) {
            // ---- a word the memo lacks: into the call-local memo (once per word per call), if it is plain: the key is
            // the word's bytes and 0x20s, so every byte of it is within 0x20 .. 0x7E ----
            auto plain = [](uint32_t v) -> bool {
              return (((v + 0x01010101u) | v) & 0x80808080u) == 0u && (((v - 0x20202020u) & ~v) & 0x80808080u) == 0u;
            };
            bool kept = false;
            // (dev.h kNfWordLocalNorm: any word goes in -- word_resolve_block normalizes the ones that are not plain)
            if (any_word || (plain(k0) && plain(k1) && plain(k2) && plain(k3))) {
              const unsigned long long tag = DynTag(k0, k1, k2, k3);
              uint32_t sl = static_cast<uint32_t>(tag >> 32) & a.dyn_mask;
              // (the first slot's tag came with the probe, stage A, while the wavefront is warm: most occurrences of a word
              // find it entered)
              unsigned long long g0 = static_cast<unsigned long long>(S.tg_hi) << 32 | S.tg_lo;
              const bool have0 = S.warm != 0u;
              for (uint32_t t = 0; t < kDynProbes && !kept; ++t) {
                unsigned long long g = (t == 0u && have0) ? g0 : wv::atomic_load64(&a.dyn_tag[sl]);
                if (g == 0ull) g = wv::atomic_cas(&a.dyn_tag[sl], 0ull, tag);
                if (g == 0ull) {                         // ours: the word's bytes, and a place in the list of words to segment
                  const uint32_t at2 = wv::atomic_add(a.dyn_count, 1u);
                  a.dyn_ent[4u * sl] = U4{k0, k1, k2, k3};
                  a.dyn_ent[4u * sl + 1u] = U4{at2 < a.dyn_cap ? 0u : 2u, 0u, 0u, 0u};     // (2: no room on the list: never usable)
                  if (at2 < a.dyn_cap) { a.dyn_list[at2] = sl; kept = true; }
                  break;
                }
                if (g == tag) { kept = true; break; }    // another lane has entered it
                sl = (sl + 1u) & a.dyn_mask;
              }
            }
            if (kept) miss = true; else gone = true;
          }
        }
      } else {
        if (probe && !hit) gone = true;
      }
      // ---- what the word adds: ids, and its share of the bound of |score| ----
      uint32_t id0, id1, x;
      float lim;
      if (hit16) {
        id0 = S.hw & 0xFFFFu;
        id1 = ((S.hw & kMemo16TwoPiece) && (S.hz >> 16) != 0xFFFFu) ? S.hz >> 16 : 0xFFFFFFFFu;
        x = S.hw >> 24;
        lim = wv::bits_to_float((((S.hw >> 16) & 0x7Fu) + 127u) << 23);              // 2^e: the power of two below bmax
      } else {
        id0 = e1.x; id1 = e1.y;
        const float xf = wv::bits_to_float(hitd ? dbound : e1.z);    // (an integer: a sum of ceil(|piece score|) + 1, tables.cc / resolve_unigram_lane)
        x = xf < 8192.f ? static_cast<uint32_t>(xf) : 8192u;
        lim = wv::bits_to_float(hitd ? dlim : e1.w);
      }
      if (!hit) x = 0u;
      if (hit && x >= 8192u) { gone = true; x = 0u; }
      // (:609-613) a run of unknown pieces is ONE id and may continue from the previous word of the sentence
      const uint32_t jprev = wv::lane_up1(j, 0xFFFFFFFFu);
      const bool head = jprev != j;                                  // the first of this sentence's words among the 64
      const uint32_t j0 = wv::read_lane(j, 0);
      const bool cont = j == j0 && j0 == carry_j;                    // ... of a sentence that continues from the previous 64
      uint32_t skip = 0u;
      const uint32_t lastunk = (word && hitd && (dn & kDynLastUnk)) ? 1u : 0u;
      if (MODE == kWmDyn) {
        const uint32_t pl = wv::lane_up1(lastunk, 0u);
        const bool prev_unk = head ? (cont && carry_unk != 0u) : pl != 0u;
        if (hitd && (dn & kDynFirstUnk) && prev_unk) skip = 1u;
      }
      const uint32_t cntd = dn & 0xFFu;
      if (MODE == kWmDyn) {
        // a word Normalize drops altogether (no ids): its neighbours are neighbours in the normalized text; an unknown-piece
        // run that would have to continue across it is left to the general kernels
        const uint32_t pl = wv::lane_up1(lastunk, 0u);
        const bool prev_unk = head ? (cont && carry_unk != 0u) : pl != 0u;
        if (word && hitd && cntd == 0u && prev_unk) gone = true;
      }
      const uint32_t cnt = !word ? 0u : (hitd ? cntd - skip : (hit ? (id1 != 0xFFFFFFFFu ? 2u : 1u) : 0u));
      const uint32_t packed = (x << 12) | cnt;
      const uint32_t Sc = wv::scan_add(packed);
      const uint32_t E = Sc - packed;
      const uint32_t hE = wv::scan_max(head ? E : 0u);
      const uint32_t rel = E - hE;
      uint32_t xb = rel >> 12, nb = rel & 0xFFFu;
      uint32_t n0j = 0u;
      if (MODE == kWmDyn) n0j = s_n0[j];
      if (cont) { xb += carry_x; nb += carry_n; }
      else if (MODE == kWmDyn) { xb += s_x0[j]; nb += n0j; }       // (the sentence's first words here: what the first round left in front of them)
      {   // the carry for the next 64 words: what the last word's sentence has behind it now
        const int lv = static_cast<int>(S.cnt - 1u);
        carry_j = wv::read_lane(j, lv);
        carry_x = wv::read_lane(xb + x, lv);
        carry_n = wv::read_lane(nb + cnt, lv);
        carry_unk = wv::read_lane(lastunk, lv);
      }
      // take the entry while its margin holds: |score before the word| <= xb < lim (exact integers; a float(xb) that
      // rounds can only round towards refusing: lim is a float)
      if (word && hit && !(xb < (1u << 24) && static_cast<float>(xb) < lim)) gone = true;
      const U4 sj = sent[j];                                         // {start, end, arena slot}
      if (MODE == kWmDyn && word && hit && nb + cnt > sj.y - sj.x + 1u + n0j) gone = true;   // (more ids than the slot holds: byte fallback of a finely split word)
      const bool emit = word && hit && !gone;
      // ---- ids ----
      const uint64_t so = static_cast<uint64_t>(sj.w) << 32 | sj.z;
      if (emit && !hitd) {
        if (H16) {
          uint16_t *q = reinterpret_cast<uint16_t *>(a.arena + so) + nb;
          q[0] = static_cast<uint16_t>(id0);
          if (cnt == 2u) q[1] = static_cast<uint16_t>(id1);
        } else {
          int32_t *q = a.arena + so + nb;
          q[0] = static_cast<int32_t>(id0);
          if (cnt == 2u) q[1] = static_cast<int32_t>(id1);
        }
      }
      if (MODE == kWmDyn && wv::any(emit && hitd)) {
        const bool wide = (dn & kDynWide) != 0u;
        auto idc = [&](uint32_t t) __attribute__((always_inline)) -> uint32_t {   // piece t of the entry (t a constant after unrolling)
          const uint32_t di[8] = {dia.x, dia.y, dia.z, dia.w, dib.x, dib.y, dib.z, dib.w};
          if (wide) return t < 16u ? (di[(t >> 1) & 7u] >> (16u * (t & 1u))) & 0xFFFFu : 0u;
          return t < 8u ? di[t & 7u] : 0u;
        };
        const uint32_t maxc = wv::read_lane(wv::scan_max((emit && hitd) ? cnt : 0u), 63);     // (most such words are two or three pieces)
#pragma unroll
        for (uint32_t t = 0; t < kDynMaxWide; ++t) {
          if (t >= maxc) break;
          if (emit && hitd && t < cnt) {
            const uint32_t id = skip ? idc(t + 1u) : idc(t);
            if (H16) reinterpret_cast<uint16_t *>(a.arena + so)[nb + t] = static_cast<uint16_t>(id);
            else (a.arena + so)[nb + t] = static_cast<int32_t>(id);
          }
        }
      }
      // ---- per sentence: ids so far (the last of its words among the 64 knows), status ----
      const uint32_t jnext = wv::lane_down1(word ? j : 0xFFFFFFFEu, 0xFFFFFFFEu);
      if (word && jnext != j) s_nids[j] = nb + cnt;
      if (wv::any(gone || miss)) {
        if (gone) wv::lds_atomic_or(&s_stat[j], kWwGone);
        if (miss) { wv::lds_atomic_or(&s_stat[j], kWwAgain); wv::lds_atomic_min(&s_first[j], P); }
        wv::sync();
        if (MODE == kWmCollect) {
          // where a sentence-per-lane second round takes the sentence up again: at its FIRST missing word (the sentence's
          // index is the sentence lane's: fetched for the lanes that write a resume record)
          const bool wr = miss && s_first[j] == P;
          const uint32_t sidj = wv::shfl(sid, static_cast<int>(j));
          if (wr) a.resume[sidj] = U4{P - sj.x, nb, wv::float_to_bits(static_cast<float>(xb)), 0u};
        }
      }
    };
    // One step of the pipeline takes TWO batches (q1 >= 1 and q2 >= 0 words of the queue from qbase): the first starts and
    // asks its probe, the batch that was in flight -- its probe has had a whole step to land -- finishes, the second starts,
    // the first finishes.  Always the same instruction stream: the batches take turns in SA / SB without copies (a copy
    // would wait for the loads), and the waits in front of a stage B can count on the two loads of the stage A behind it
    // being younger than what they wait for.
    auto tick2 = [&](uint32_t qbase, uint32_t q1, uint32_t q2) __attribute__((always_inline)) {
      stageA(qbase, q1, SA);
      stageB(SB);
      stageA(qbase + 64u, q2, SB);
      stageB(SA);
    };

    Q4 nx{0, 0, 0, 0};
    uint32_t nj = 0, ws_cur = 0, j_cur = 0, ws_next = 0, j_next = 0;
    if (rounds > 0u) {
      issue(0u, &nx, &nj);
      ws_cur = commit(0u, nx, nj);
      j_cur = nj;
      if (rounds > 1u) issue(1u, &nx, &nj);
    }
    uint32_t qn = 0u;                                               // words waiting in the queue (< 128 between rounds)
    bool aged = false;                                              // ... some of them since the round before the last one
    for (uint32_t r = 0; r < rounds; ++r) {
      if (r + 1u < rounds) {                                        // round r + 1 into the ring; round r + 2 asked for
        ws_next = commit(r + 1u, nx, nj);
        j_next = nj;
        if (r + 2u < rounds) issue(r + 2u, &nx, &nj);
      }
      // ---- this round's word starts behind what the queue holds, in text order ----
      const uint32_t cw = static_cast<uint32_t>(__builtin_popcount(ws_cur));
      const uint32_t cw_incl = wv::scan_add(cw);
      uint32_t wb = qn + cw_incl - cw;
      const uint32_t avail = qn + wv::read_lane(cw_incl, 63);
      {
        uint32_t m = ws_cur;
        const uint32_t recbase = (j_cur << 26) | ((r * 64u + static_cast<uint32_t>(lane)) * 16u);
        while (wv::any(m != 0u)) {
          if (m != 0u) {
            const uint32_t b = static_cast<uint32_t>(__builtin_ctz(m));
            m &= m - 1u;
            recq[wb++] = recbase + b;
          }
        }
      }
      wv::sync();
      // Pairs of full batches go; what is left (< 128 words) waits for the next round's -- unless some of it has waited
      // two rounds (the ring keeps the text of the current round and the two before it) or there is no next round.
      const bool last = r + 1u == rounds;
      const bool all = last || (avail < 128u && aged);
      const uint32_t take = all ? avail : (avail & ~127u);
      for (uint32_t qb = 0; qb < take; qb += 128u) {
        const uint32_t q1 = take - qb < 64u ? take - qb : 64u;
        const uint32_t q2 = take - qb < 128u ? take - qb - q1 : 64u;
        tick2(qb, q1, q2);
      }
      const uint32_t left = avail - take;
      aged = take == 0u && qn != 0u;                                // (nothing went: what the queue held is a round older now)
      if (left != 0u && take != 0u) {                               // to the front of the queue
        const uint32_t l32 = static_cast<uint32_t>(lane);
        const uint32_t v0 = recq[take + (l32 < left ? l32 : 0u)];
        const uint32_t v1 = recq[take + (l32 + 64u < left ? l32 + 64u : 0u)];
        wv::sync();
        if (l32 < left) recq[l32] = v0;
        if (l32 + 64u < left) recq[l32 + 64u] = v1;
      }
      qn = left;
      wv::sync();                                                   // (the queue and the ring's oldest round are rewritten next)
      ws_cur = ws_next;
      j_cur = j_next;
    }
    stageB(SB);                                                     // the tile's last batch
    SB.cnt = 0u;
    wv::sync();
    const unsigned long long c1 = wv::clock();
    // ---- per sentence: done, kept for the second round, or handed on ----
    const uint32_t st = s_stat[lane];
    const int n = static_cast<int>(s_nids[lane]);
    const bool done = work && st == 0u;
    if (done) {
      if (H16) {
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
    // (direct: what is handed on goes to the list of its LENGTH class -- the general kernels plan by those; what the second
    // round takes is one list, row 0)
    auto hand_on = [&](bool pred, uint32_t *lists, uint32_t *counts) __attribute__((always_inline)) {
      if (!direct) { append_lanes(wv::ballot(pred), pred, sid, lists + static_cast<uint64_t>(c) * a.n, &counts[c], lane); return; }
      if (!wv::any(pred)) return;
      for (uint32_t k = 0; k < a.n_classes; ++k) {
        const bool pk = pred && cl == k;
        append_lanes(wv::ballot(pk), pk, sid, lists + static_cast<uint64_t>(k) * a.n, &counts[k], lane);
      }
    };
    if (MODE == kWmCollect) {
      const bool again = left && work && st == kWwAgain;
      const bool gonel = left && !again;
      if (again) a.tmp_off[sid] = static_cast<unsigned long long>(slot - d.n_prefix - a.arena);
      const uint32_t ca = direct ? 0u : c;
      append_lanes(wv::ballot(again), again, sid, a.left_lists + static_cast<uint64_t>(ca) * a.n, &a.left_counts[ca], lane);
      hand_on(gonel, a.left2_lists, a.left2_counts);
    } else {
      hand_on(left, a.left_lists, a.left_counts);
    }
    if (done) { ++tc.n_sent; tc.n_raw += static_cast<unsigned long long>(len); tc.n_ids += static_cast<unsigned long long>(n + n_extra); }
    tc.n_trips += steps;
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

}  // namespace spmx
#endif
