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
// memo (dev.h umemo, tables.cc BuildWordMemo) therefore stores, next to a word's best segmentation, the largest
// magnitude of the accumulated score up to which the decision is provably the exact-arithmetic one:
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
// At run time a lane keeps B exactly as the reference has it -- B = (float)((double)score + (double)B) per emitted piece,
// the very operation of :982-989 along the winning path -- and takes a memo entry only while |B| < bmax.  Anything else
// (a word that is not in the memo, |B| too large for its margin, a byte outside 0x21-0x7E, a word of more than 16 bytes,
// text within 20 bytes of the end of the buffer) sends the WHOLE sentence to the general kernels (kernels_stream.h)
// through the call's leftover lists: this kernel either produces the reference's ids or produces nothing.
//
// What the normalizer contributes is implicit: the model must add a dummy prefix, remove extra whitespace and escape
// whitespace with the one-byte space symbol, and every byte 0x20-0x7E must be a character no charsmap rule starts with
// (tables.cc checks all of it), so Normalize() of such a sentence is "words joined by single space symbols, one in
// front" -- a word's normalized form is the space symbol plus its raw bytes, which is how the memo is keyed (by the raw
// bytes alone).  Leading, trailing and doubled spaces are empty words and cost an iteration each.
//
// Per iteration and lane: one unaligned 20-byte read of the text (issued one word ahead), one 32-byte probe of the memo.
// No scratch in HBM, no back-pointers, no backtrack: ids leave in forward order through a 16-id LDS staging column as
// 64-byte bursts.
#ifndef SPMX_KERNELS_WORD_H_
#define SPMX_KERNELS_WORD_H_

namespace spmx {

constexpr uint32_t kWordLdsShared = 18u * 16u + 32u;            // mask rows 0 .. 17 (+ padding)
constexpr uint32_t kWordLdsPerWave = 64u * 16u * 4u;            // id staging: [16][64] int32
SPMX_HD inline uint32_t WordLdsBytes(uint32_t waves) { return kWordLdsShared + waves * kWordLdsPerWave; }

// 16 bytes at any address (gfx950 runs with unaligned vector memory access enabled; scripts/ubench/unaligned_probe.hip)
struct __attribute__((packed, aligned(1))) Q4U { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U1U { uint32_t x; };

// bit 7 of every byte of v that equals 0x20 -- exact for the LOWEST such byte, which is all that is used
// (a borrow can only flag the byte above a true one)
SPMX_DEVICE uint32_t space_bits(uint32_t v) {
  const uint32_t x = v ^ 0x20202020u;
  return (x - 0x01010101u) & ~x & 0x80808080u;
}

// The words of this lane's sentence (raw bytes gtext[beg, beg + len)) -> ids in slot[0, n), forward order.
// Returns n >= 0, or -1: the sentence is not for this kernel (nothing usable was written).
// `stage`: this lane's column of the wave's id staging (entry k at stage[k << 6]); `masks`: the 18 key masks in LDS.
SPMX_DEVICE int uni_word_lane(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int len, int32_t *slot, int cap,
                              int32_t *stage, const Q4 *masks, bool active_in, int *n_steps) {
  const U4 *__restrict__ memo = d.umemo;
  const uint32_t mmask = d.umemo_mask;
  const uint8_t *text = gtext + beg;
  bool active = active_in && len > 0;
  bool bad = false;
  int p = 0, n = 0, steps = 0;
  float B = 0.f;                                   // best_path_score at the start of the current word
  Q4U w{0, 0, 0, 0};
  uint32_t w4 = 0;
  if (active) {
    w = *reinterpret_cast<const Q4U *>(text);
    w4 = reinterpret_cast<const U1U *>(text + 16)->x;
  }
  while (wv::any(active)) {
    ++steps;
    // ---- the word that starts at p: its length = the distance to the next 0x20 (or to the end of the sentence) ----
    const uint32_t z0 = space_bits(w.x), z1 = space_bits(w.y), z2 = space_bits(w.z), z3 = space_bits(w.w), z4 = space_bits(w4) & 0x80u;
    const int fa = wv::ffs64(static_cast<uint64_t>(z0) | static_cast<uint64_t>(z1) << 32);
    const int fb = wv::ffs64(static_cast<uint64_t>(z2) | static_cast<uint64_t>(z3) << 32);
    int L = fa ? (fa - 1) >> 3 : (fb ? 8 + ((fb - 1) >> 3) : (z4 ? 16 : 17));
    const int rem = len - p;
    if (L > rem) L = rem;
    const bool word = active && L > 0;               // L == 0: a space (leading, doubled): skip it
    const bool lng = word && L > 16;
    const int pn = p + L + 1;
    const bool more = active && !lng && pn < len;
    // ---- the next word's text, one iteration ahead of its use ----
    Q4U wn = w;
    uint32_t wn4 = w4;
    if (more) {
      wn = *reinterpret_cast<const Q4U *>(text + pn);
      wn4 = reinterpret_cast<const U1U *>(text + pn + 16)->x;
    }
    // ---- memo probe ----
    const Q4 mk = masks[word && !lng ? L : 0];
    const uint32_t k0 = w.x & mk.x, k1 = w.y & mk.y, k2 = w.z & mk.z, k3 = w.w & mk.w;
    uint32_t sl = HashWordKey(k0, k1, k2, k3) & mmask;
    const bool probing = word && !lng;
    U4 e0 = memo[2 * (probing ? sl : 0u)], e1 = memo[2 * (probing ? sl : 0u) + 1];
    bool hit = probing && e0.x == k0 && e0.y == k1 && e0.z == k2 && e0.w == k3 && e1.x != 0xFFFFFFFFu;
    bool walk = probing && !hit && e1.x != 0xFFFFFFFFu;
    while (wv::any(walk)) {                          // a collision: walk on (rare: the table is half empty)
      if (walk) {
        sl = (sl + 1u) & mmask;
        e0 = memo[2 * sl];
        e1 = memo[2 * sl + 1];
        hit = e0.x == k0 && e0.y == k1 && e0.z == k2 && e0.w == k3 && e1.x != 0xFFFFFFFFu;
        walk = !hit && e1.x != 0xFFFFFFFFu;
      }
    }
    // ---- take the entry while the margin holds: |B| < bmax ----
    const float bmax = wv::bits_to_float(e1.w);
    const bool ok = hit && fabsf(B) < bmax;
    if (word && !ok) { bad = true; active = false; }
    if (ok) {
      const bool two = e1.y != 0xFFFFFFFFu;
      if (n + (two ? 2 : 1) > cap) { bad = true; active = false; }
      else {
        B = static_cast<float>(static_cast<double>(wv::bits_to_float(e1.z)) + static_cast<double>(B));   // :982-989
        stage[(n & 15) << 6] = static_cast<int32_t>(e1.x);
        ++n;
        if ((n & 15) == 0) {
          int32_t *q = slot + (n - 16);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<Q4 *>(q + 4 * r) = Q4{static_cast<uint32_t>(stage[(4 * r) << 6]), static_cast<uint32_t>(stage[(4 * r + 1) << 6]),
                                                   static_cast<uint32_t>(stage[(4 * r + 2) << 6]), static_cast<uint32_t>(stage[(4 * r + 3) << 6])};
        }
        if (two) {                                   // (rare) the second piece: its score from the per-id table
          B = static_cast<float>(static_cast<double>(d.pscore[e1.y]) + static_cast<double>(B));
          stage[(n & 15) << 6] = static_cast<int32_t>(e1.y);
          ++n;
          if ((n & 15) == 0) {
            int32_t *q = slot + (n - 16);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<Q4 *>(q + 4 * r) = Q4{static_cast<uint32_t>(stage[(4 * r) << 6]), static_cast<uint32_t>(stage[(4 * r + 1) << 6]),
                                                     static_cast<uint32_t>(stage[(4 * r + 2) << 6]), static_cast<uint32_t>(stage[(4 * r + 3) << 6])};
          }
        }
      }
    }
    if (active && !more) active = false;             // the sentence is done
    p = pn;
    w = wn;
    w4 = wn4;
  }
  *n_steps = steps;
  if (bad) return -1;
  if (active_in && len > 0)
    for (int k = n & ~15; k < n; ++k) slot[k] = stage[(k & 15) << 6];   // the last, incomplete group
  return n;
}

// Persistent body of the word kernel: tiles of up to 64 sentences from the launch's queue (kernels_stream.h next_tile),
// one sentence per lane.  What a lane cannot take goes to the leftover list of its class.
SPMX_DEVICE void encode_word_block(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  Q4 *masks = reinterpret_cast<Q4 *>(smem);
  int32_t *stage = reinterpret_cast<int32_t *>(smem + kWordLdsShared + static_cast<uint32_t>(wv::wave_in_block()) * kWordLdsPerWave) + lane;
  if (lane < 18) {                                   // (every wave writes the same rows: no workgroup barrier)
    const uint32_t L = static_cast<uint32_t>(lane);
    auto m = [&](uint32_t i) -> uint32_t { return L >= 4u * i + 4u ? 0xFFFFFFFFu : (L <= 4u * i ? 0u : (1u << (8u * (L - 4u * i))) - 1u); };
    masks[lane] = Q4{m(0), m(1), m(2), m(3)};
  }
  wv::sync();
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
    bool mine = have && l64 < (1ull << 30) && beg + l64 + 20u <= text_end;
    const int len = mine ? static_cast<int>(l64) : 0;
    // ---- a slot of cap ids in the arena (at most two ids per word, a word per two bytes) ----
    const int cap = mine ? len + 1 : 0;
    const int room = mine ? (cap + n_extra + 3 + 3) & ~3 : 0;
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
    int steps = 0;
    int n = uni_word_lane(d, a.text, beg, len, slot, cap, stage, masks, mine && !overflow, &steps);
    const unsigned long long c1 = wv::clock();
    if (overflow) n = -1;
    const bool done = mine && n >= 0;
    if (done) {
      for (int x = 0; x < d.n_prefix; ++x) slot[x - d.n_prefix] = d.prefix_ids[x];
      for (int x = 0; x < d.n_suffix; ++x) slot[n + x] = d.suffix_ids[x];
      a.tmp_off[sid] = static_cast<unsigned long long>(slot - d.n_prefix - a.arena);
      a.counts[sid] = static_cast<uint32_t>(n + n_extra);
    }
    const bool left = have && !done;
    if (left) a.counts[sid] = 0u;                    // (until the general kernels have had it)
    append_lanes(wv::ballot(left), left, sid, a.left_lists + static_cast<uint64_t>(c) * a.n, &a.left_counts[c], lane);
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

}  // namespace spmx
#endif
