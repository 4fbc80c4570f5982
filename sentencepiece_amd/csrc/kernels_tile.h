// Unigram segmentation, tile form: one wavefront takes a TILE of up to 64
// sentences.  Normalization stays position-parallel (one sentence at a time,
// 64 byte positions per sweep, normalize_wave); the Viterbi search, the
// backtrack and the id store run one SENTENCE PER LANE, so that all 64 lanes
// of the wave carry an independent EncodeOptimized recurrence
// (src/unigram_model.cc:889-1020) instead of one recurrence using a handful of
// lanes.  Measured motivation (profiles/, round 1): in the sentence-per-wave
// form 91 % of the wave cycles went into the serial end-position loop.
//
// Per lane, in LDS: the normalized text (nlen bytes), one back-pointer byte
// per position (piece length | unknown flag) and a ring of the best scores of
// the last R positions (R > longest piece).  Ids are NOT stored: (start, end)
// determines the id (a piece of the trie or the unknown id, never both:
// :990-1005), so the backtrack re-walks the trie over each chosen piece.
//
// The two nested loops of the reference (for each start: for each prefix) are
// flattened into one loop in which every lane does exactly one trie probe per
// iteration; a lane whose walk dies relaxes UNK and moves to its next start in
// the same iteration.  Lanes therefore stay busy regardless of how deep their
// neighbours' walks go.
#ifndef SPMX_KERNELS_TILE_H_
#define SPMX_KERNELS_TILE_H_

namespace spmx {

constexpr uint32_t kBpUnk = 0x80u;     // back-pointer byte: the UNK candidate won this position
constexpr uint32_t kBpLen = 0x7Fu;

struct TileLds {
  U4 *roottab;       // [256] trie units of the one-byte prefixes (first level of every walk)
  uint8_t *raw;      // staging of one raw sentence (rcap + 16)
  float *ring;       // [R][64] best scores, lane-interleaved
  uint8_t *area;     // text + back-pointer regions of the round's sentences
};

SPMX_HD inline uint32_t TileRawBytes(uint32_t rcap) { return (rcap + 16 + 15) & ~15u; }
SPMX_HD inline uint32_t TileLdsBytes(uint32_t rcap, uint32_t ring, uint32_t area) {
  return 256u * 16u + TileRawBytes(rcap) + 64u * ring * 4u + ((area + 15) & ~15u);
}

SPMX_DEVICE TileLds carve_tile(unsigned char *base, uint32_t rcap, uint32_t ring) {
  TileLds t;
  t.roottab = reinterpret_cast<U4 *>(base);
  base += 256u * 16u;
  t.raw = base;
  t.ring = reinterpret_cast<float *>(base + TileRawBytes(rcap));
  t.area = base + TileRawBytes(rcap) + 64u * ring * 4u;
  return t;
}

// :979-983 score of the piece in unit u (byte length len) as the reference's double
SPMX_DEVICE double piece_score(const U4 &u, int len, float max_score) {
  double score = static_cast<double>(wv::bits_to_float(u.z));
  if (u.y & kPtUserDefined) {                                      // (length * max_score_ - 0.1)
    const float prod = static_cast<float>(len) * max_score;
    score = static_cast<double>(prod) - 0.1;
  }
  return score;
}

// EncodeOptimized for this lane's sentence.  text/bp are this lane's regions, ring is &ring[lane]; slot of
// position e is ring[(e & rm) * 64]; roottab[c] is the trie unit of the one-byte prefix c (LDS copy).
//
// The reference's two nested loops (for each start: for each prefix, :960-1008) are flattened so that one
// iteration costs ONE global trie probe per lane, split into a control half and a data half:
//   control  (registers only) -- consume the probe in flight (:969-971); the child-label summary in the unit
//            tells whether the next byte can match at all, so a walk's last, failing probe is usually never
//            issued; if the walk of this start is over, move to the next start (:1007) and take its first
//            trie level from the root-table unit that was fetched from LDS when the previous start began;
//            issue the next probe;
//   data     (LDS, under the shadow of that probe) -- up to three relaxations of best_path_ends_at, in the
//            reference's order: (A) the piece just matched (double add, double compare against the
//            float-rounded best, :979-989), (B) UNK for the start that is over unless a one-character piece
//            was seen (float add, :990-1005), (C) a one-byte piece of the next start.  Their six LDS reads
//            are issued together; where two of them hit the same position the later one is forwarded the
//            earlier one's result in registers.
// Returns the number of iterations (wave-uniform) for the profiling counters.
SPMX_DEVICE int unigram_lane(const SpmxDev &d, const uint8_t *text, uint8_t *bp, int nlen, float *ring, uint32_t rm,
                             const U4 *roottab, bool active_in) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const float unk_score = d.unk_score, max_score = d.max_score;
  int trips = 0;
  bool active = active_in && nlen > 0;
  if (!active) nlen = 0;                          // all indices of an idle lane stay 0
  // a finished pseudo-start in front of position 0: single = true (no UNK), mb = 0 (the next start is 0)
  int s = 0, mb = 0, dep = 0;
  bool walking = false, single = true;
  uint32_t c = 0;
  float sbest = 0.f;
  U4 u{0, 0, 0, 0};
  // fetched when a start begins, for the start that follows it: first byte, the byte after, root-table unit
  uint32_t cs = 0, cs1 = 0;
  U4 rn{0, 0, 0, 0};
  uint32_t cq = 0;                                // text[s + dep + 1]: the byte after the one being probed
  if (active) {
    ring[0] = 0.f;                                // best_path_ends_at[0].best_path_score = 0
    cs = text[0];
    cs1 = text[1];
    rn = roottab[cs];
  }
  while (wv::any(active)) {
    ++trips;
    // ---------------- control ----------------
    const int dep1 = dep + 1;
    const bool matchA = active && walking && (u.x & 0x1FFu) == (0x100u | c);           // :969-971
    const bool termA = matchA && (u.x & kDatTerminalDev) && !(u.y & kPtUnused);          // :973-974
    const bool cont = matchA && s + dep1 < nlen && ((u.w >> ChildBit(cq)) & 1u);
    const bool ended = active && !cont;           // not walking, mismatch, or no child can match the next byte
    const int s2 = s + mb;                        // :1007 the next start
    const bool begin = ended && s2 < nlen;
    int mb2 = OneCharLenDev(cs);                  // :962-963
    if (mb2 > nlen - s2) mb2 = nlen - s2;
    const U4 r = rn;
    const bool rootC = begin && (r.x & 0x1FFu) == (0x100u | cs);                         // first trie level
    const bool termC = rootC && (r.x & kDatTerminalDev) && !(r.y & kPtUnused);
    const bool contC = rootC && s2 + 1 < nlen && ((r.w >> ChildBit(cs1)) & 1u);
    const bool nwalking = cont || contC;
    const uint32_t nnode = cont ? (u.x >> kDatBaseShiftDev) : (r.x >> kDatBaseShiftDev);
    const uint32_t nc = cont ? cq : cs1;
    const U4 uA = u;                              // the unit the data half scores
    if (nwalking) u = ptrie[nnode ^ nc];          // next probe
    // ---------------- data ----------------
    const int eA = s + dep1;                      // <= nlen when matchA
    const int eB = s2;                            // <= nlen
    const int eC = s2 + 1 <= nlen ? s2 + 1 : nlen;
    float *slotA = ring + ((static_cast<uint32_t>(eA) & rm) << 6);
    float *slotB = ring + ((static_cast<uint32_t>(eB) & rm) << 6);
    float *slotC = ring + ((static_cast<uint32_t>(eC) & rm) << 6);
    const int iA = matchA ? eA : 0;
    uint32_t bpA = bp[iA], bpB = bp[eB], bpC = bp[eC];
    float rA = *(matchA ? slotA : ring), rB = *slotB, rC = *slotC;
    // (A) the piece that just matched
    const double candA = piece_score(uA, dep1, max_score) + static_cast<double>(sbest);  // :982-983
    const bool updA = termA && (bpA == 0 || candA > static_cast<double>(rA));            // :984-989
    const float nvA = static_cast<float>(candA);
    if (updA && eB == eA) { rB = nvA; bpB = static_cast<uint32_t>(dep1); }
    if (updA && eC == eA) { rC = nvA; bpC = static_cast<uint32_t>(dep1); }
    const bool single2 = single || (termA && dep1 == mb);                                // :990
    // (B) UNK for the start that is over
    const float candB = unk_score + sbest;                                               // :997-1001, float
    const bool updB = ended && !single2 && (bpB == 0 || candB > rB);
    const float sbest2 = updB ? candB : rB;       // best score at the next start, after (A) and (B)
    // (C) a one-byte piece of the next start
    const double candC = piece_score(r, 1, max_score) + static_cast<double>(sbest2);
    const bool updC = termC && (bpC == 0 || candC > static_cast<double>(rC));
    if (updA) { *slotA = nvA; bp[eA] = static_cast<uint8_t>(dep1); }
    if (updB) { *slotB = candB; bp[eB] = static_cast<uint8_t>(static_cast<uint32_t>(mb) | kBpUnk); }
    if (updC) { *slotC = static_cast<float>(candC); bp[eC] = 1; }
    // ---------------- commit ----------------
    if (ended) {
      active = begin;
      s = s2;
      mb = begin ? mb2 : 0;
      sbest = sbest2;
      single = termC && mb2 == 1;
      dep = rootC ? 1 : 0;
      if (begin) {                                // what the start after this one will need
        cs = text[s2 + mb2];
        cs1 = text[s2 + mb2 + 1];
        rn = roottab[cs];
      }
    } else {
      dep = dep1;
      single = single2;
    }
    walking = nwalking;
    c = nc;
    if (nwalking) cq = text[s + dep + 1];
  }
  return trips;
}

// Number of ids this lane's sentence produces (:1010-1018 backtrack + the
// unknown-run merge / byte-fallback expansion of sentencepiece_processor.cc:
// 581-613).  -1 on a broken chain.
SPMX_DEVICE int count_lane(const SpmxDev &d, const uint8_t *bp, int nlen, bool active) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  int n = 0;
  if (!active) return 0;
  int e = nlen;
  bool right_unk = false;
  while (e > 0) {
    const uint32_t b = bp[e];
    const int len = static_cast<int>(b & kBpLen);
    if (len == 0 || len > e) return -1;
    if (b & kBpUnk) {
      if (bf) n += len;
      else if (!right_unk) n += 1;
      right_unk = true;
    } else {
      n += 1;
      right_unk = false;
    }
    e -= len;
  }
  return n;
}

// Writes this lane's n ids to dst[0, n) (forward order, or reversed).  One trie
// probe per iteration per lane, pieces visited from the last to the first.
SPMX_DEVICE void write_lane(const SpmxDev &d, const uint8_t *text, const uint8_t *bp, int nlen, int n, int32_t *dst,
                            bool active) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  int e = nlen, j = n;        // j: forward index one past the next id to write
  int k = 0, len = 0, tb = 0;
  uint32_t node = root, last_y = 0;
  bool right_unk = false;
  active = active && nlen > 0 && n > 0;
  while (wv::any(active)) {
    if (active) {
      if (k == 0) {                                   // open the piece that ends at e
        const uint32_t b = bp[e];
        len = static_cast<int>(b & kBpLen);
        tb = e - len;
        if (b & kBpUnk) {
          if (bf) {                                   // one BYTE id per byte of the unknown piece (:581-603)
            for (int x = len - 1; x >= 0; --x) {
              --j;
              dst[reverse ? n - 1 - j : j] = d.byte_ids[text[tb + x]];
            }
          } else if (!right_unk) {                    // a run of unknown pieces yields one id (:609-613)
            --j;
            dst[reverse ? n - 1 - j : j] = d.unk_id;
          }
          right_unk = true;
          e = tb;
          len = 0;                                    // nothing to walk
          if (e <= 0) active = false;
        } else {
          right_unk = false;
          node = root;
        }
      }
      if (len > 0) {                                  // one byte of the piece's trie path
        const uint32_t c = text[tb + k];
        const U4 u = d.ptrie[node ^ c];
        node = u.x >> kDatBaseShiftDev;
        last_y = u.y;
        if (++k == len) {
          --j;
          dst[reverse ? n - 1 - j : j] = static_cast<int32_t>(last_y & kPtIdMask);
          e = tb;
          k = 0;
          if (e <= 0) active = false;
        }
      }
    }
  }
}

// Persistent block body of the tile form.
SPMX_DEVICE void encode_tile_block(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const TileLds T = carve_tile(smem, a.rcap, a.ring);
  const uint32_t rm = a.ring - 1;
  float *my_ring = T.ring + lane;
  {   // LDS copy of the trie's first level: unit of byte c, or an empty unit if no piece starts with c
    const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
    for (uint32_t cb = static_cast<uint32_t>(lane); cb < 256u; cb += 64u) {
      U4 r = d.ptrie[root ^ cb];
      if ((r.x & 0x1FFu) != (0x100u | cb)) r = U4{0, 0, 0, 0};
      T.roottab[cb] = r;
    }
    wv::sync();
  }
  const uint32_t count = *a.list_count;
  const uint32_t tiles = (count + 63) / 64;
  const int n_extra = d.n_prefix + d.n_suffix;
  unsigned long long n_sent = 0, n_raw = 0, n_ids = 0;
  unsigned long long cyc[4] = {0, 0, 0, 0};
  unsigned long long n_trips = 0;     // iterations of the per-lane search loop (wave-uniform)
  for (uint32_t tile = static_cast<uint32_t>(wv::block_id()); tile < tiles; tile += static_cast<uint32_t>(wv::grid_size())) {
    const uint32_t first = tile * 64;
    const int cnt = static_cast<int>(count - first < 64 ? count - first : 64);
    // this lane's sentence
    uint32_t my_sid = 0;
    uint64_t my_beg = 0;
    uint32_t my_len = 0;
    if (lane < cnt) {
      my_sid = a.list[first + lane];
      my_beg = a.offs[my_sid];
      my_len = static_cast<uint32_t>(a.offs[my_sid + 1] - my_beg);
      if (a.offs[my_sid + 1] - my_beg > a.rcap) my_len = 0xFFFFFFFFu;   // only reachable in the last class
    }
    int i0 = 0;
    while (i0 < cnt) {
      // ---- round: pack sentences i0.. into the area until it is full ----
      const unsigned long long c0 = wv::clock();
      uint32_t used = 0;
      int my_off = -1, my_nlen = 0;     // my_off < 0: this lane has no sentence in this round
      int i = i0;
      unsigned long long t_load = 0;
      for (; i < cnt; ++i) {
        const unsigned long long l0 = wv::clock();
        const uint32_t L = wv::shfl(my_len, i);
        const uint32_t sid = wv::shfl(my_sid, i);
        if (L == 0xFFFFFFFFu) { fail_sentence(a, sid, kStTooLong, lane); continue; }
        const uint64_t beg = static_cast<uint64_t>(wv::shfl(static_cast<uint32_t>(my_beg >> 32), i)) << 32 |
                             wv::shfl(static_cast<uint32_t>(my_beg), i);
        const uint8_t *src = a.text + beg;
        for (uint32_t p = static_cast<uint32_t>(lane); p < L; p += 64) T.raw[p] = src[p];
        wv::sync();
        t_load += wv::clock() - l0;
        const uint32_t room = a.tile_area - used;
        uint32_t cap = room >= 3 ? (room - 1) / 2 : 0;
        const bool class_bound = cap >= a.ncap;
        if (class_bound) cap = a.ncap;
        int nlen = 0;
        if (L > 0) nlen = cap > 0 ? normalize_wave(d, T.raw, static_cast<int>(L), T.area + used, static_cast<int>(cap), lane) : -1;
        wv::sync();   // every lane has read the tail of the text (trailing-space trim) before anyone clears it
        if (nlen < 0) {
          if (!class_bound && i > i0) break;             // the area is full: this sentence opens the next round
          // does not fit this class at all: hand it on (or fail in the last class)
          if (a.next_list) { if (lane == 0) a.next_list[wv::atomic_add(a.next_count, 1u)] = sid; }
          else fail_sentence(a, sid, kStTooLong, lane);
          continue;
        }
        if (lane == i) { my_off = static_cast<int>(used); my_nlen = nlen; }
        // back-pointer bytes of this sentence: [used + nlen, used + 2 nlen + 1) (0 = position not reached yet).
        // Whatever an earlier normalize_wave left there (trimmed trailing spaces, a failed attempt) is cleared here.
        for (uint32_t p = used + static_cast<uint32_t>(nlen) + static_cast<uint32_t>(lane);
             p < used + 2u * static_cast<uint32_t>(nlen) + 1u; p += 64) T.area[p] = 0;
        used += 2u * static_cast<uint32_t>(nlen) + 1u;
      }
      wv::sync();
      const unsigned long long c1 = wv::clock();
      const bool mine = my_off >= 0;
      const uint8_t *text = T.area + (mine ? my_off : 0);
      uint8_t *bp = T.area + (mine ? my_off + my_nlen : 0);
      // ---- segment: one sentence per lane ----
      n_trips += static_cast<unsigned long long>(unigram_lane(d, text, bp, my_nlen, my_ring, rm, T.roottab, mine));
      const unsigned long long c2 = wv::clock();
      // ---- ids ----
      int n = count_lane(d, bp, my_nlen, mine);
      const bool broken = n < 0;
      if (broken) n = 0;
      const int n_out = mine ? n + n_extra : 0;
      int total = 0;
      const int excl = wave_excl_scan(n_out, lane, &total);
      unsigned long long base = 0;
      if (lane == 0 && total > 0) base = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(total));
      base = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(base >> 32), 0)) << 32) |
             wv::shfl(static_cast<uint32_t>(base), 0);
      const bool overflow = base + static_cast<unsigned long long>(total) > a.arena_cap;
      if (mine) {
        a.counts[my_sid] = broken ? 0u : static_cast<uint32_t>(n_out);
        a.tmp_off[my_sid] = base + static_cast<unsigned long long>(excl);
      }
      if (wv::any(broken) && lane == 0) wv::atomic_or(a.status, kStInternal);
      if (overflow) {
        if (lane == 0) wv::atomic_or(a.status, kStArenaOverflow);
      } else {
        int32_t *dst = a.arena + base + static_cast<unsigned long long>(excl);
        if (mine && !broken) {
          for (int x = 0; x < d.n_prefix; ++x) dst[x] = d.prefix_ids[x];
          for (int x = 0; x < d.n_suffix; ++x) dst[d.n_prefix + n + x] = d.suffix_ids[x];
        }
        write_lane(d, text, bp, my_nlen, n, dst + d.n_prefix, mine && !broken);
      }
      const unsigned long long c3 = wv::clock();
      if (mine && !broken) { ++n_sent; n_raw += my_len; n_ids += static_cast<unsigned long long>(n_out); }
      cyc[0] += t_load; cyc[1] += (c1 - c0) - t_load; cyc[2] += c2 - c1; cyc[3] += c3 - c2;
      i0 = i;
    }
  }
  if (a.stats) {
    // per-lane sentence counters -> wave totals
    unsigned long long v[3] = {n_sent, n_raw, n_ids};
    for (int k = 0; k < 3; ++k) {
      uint64_t tot = 0;
      wave_excl_scan64(v[k], lane, &tot);
      if (lane == 0 && tot) wv::atomic_add(&a.stats[k], static_cast<unsigned long long>(tot));
    }
    if (lane == 0) {
      for (int k = 0; k < 4; ++k) wv::atomic_add(&a.stats[3 + k], cyc[k]);
      wv::atomic_add(&a.stats[7], n_trips);
    }
  }
}

}  // namespace spmx
#endif
