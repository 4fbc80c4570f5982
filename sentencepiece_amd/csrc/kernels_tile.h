// Unigram segmentation, tile form: one wavefront takes a TILE of up to 64
// sentences and runs the Viterbi search, the backtrack and the id store one
// SENTENCE PER LANE, so that all 64 lanes of the wave carry an independent
// EncodeOptimized recurrence (src/unigram_model.cc:889-1020) instead of one
// recurrence using a handful of lanes.  Measured motivation (profiles/, round
// 1): in the sentence-per-wave form 91 % of the wave cycles went into the
// serial end-position loop.
//
// Two kernels share the search and the id phases and differ in how the tile's
// sentences are normalized (src/normalizer.cc:71-186):
//
//   FAST     each lane normalizes its own sentence straight from HBM
//            (fast_norm_lane): possible when every byte is an ASCII byte that
//            no charsmap rule can start at (SpmxDev::ascii_safe) and the space
//            symbol is one byte wide.  A lane that meets any other byte hands
//            its sentence to the GENERAL kernel through a device-side list.
//   GENERAL  position-parallel normalize_wave (kernels.h), one sentence at a
//            time, 64 byte positions per sweep: charsmap rules, user-defined
//            symbols, malformed UTF-8.
//
// Per lane, in LDS: the normalized text (nlen bytes), one back-pointer byte
// per position (piece length | unknown flag) and a ring of the best scores of
// the last R positions (R > longest piece).  Ids are NOT stored: (start, end)
// determines the id (a piece of the trie or the unknown id, never both:
// :990-1005); the backtrack fingerprints the bytes of each chosen piece and
// looks the id up in SpmxDev::idtab (one probe per piece).
//
// The two nested loops of the reference (for each start: for each prefix) are
// flattened into one loop in which every lane does exactly one trie probe per
// iteration; a lane whose walk dies relaxes UNK and moves to its next start in
// the same iteration.  Lanes therefore stay busy regardless of how deep their
// neighbours' walks go.
//
// A workgroup is W wavefronts that share nothing but two read-only LDS tables
// (first trie level, byte classes); each wave owns a private slice of LDS, so
// no workgroup barrier is ever needed.
#ifndef SPMX_KERNELS_TILE_H_
#define SPMX_KERNELS_TILE_H_

namespace spmx {

constexpr uint32_t kBpUnk = 0x80u;     // back-pointer byte: the UNK candidate won this position
constexpr uint32_t kBpLen = 0x7Fu;

// byte classes of the FAST normalizer (TileLds::bcls)
constexpr uint32_t kBcComplex = 1u;    // not handled by fast_norm_lane: non-ASCII, or a charsmap rule may start here

struct TileLds {
  U4 *roottab;       // [256] trie units of the one-byte prefixes (first level of every walk); shared by the workgroup
  uint8_t *bcls;     // [256] byte classes (kBc*); shared by the workgroup
  uint8_t *raw;      // GENERAL: staging of one raw sentence (rcap + 16)
  float *ring;       // [R][64] best scores, lane-interleaved
  uint8_t *area;     // text + back-pointer regions of the round's sentences
};

constexpr uint32_t kTileSharedBytes = 256u * 16u + 256u;
SPMX_HD inline uint32_t TileRawBytes(uint32_t rcap) { return (rcap + 16 + 15) & ~15u; }
// LDS bytes private to one wave
SPMX_HD inline uint32_t TilePrivateBytes(bool fast, uint32_t rcap, uint32_t ring, uint32_t area) {
  return (fast ? 0u : TileRawBytes(rcap)) + 64u * ring * 4u + ((area + 15) & ~15u);
}
SPMX_HD inline uint32_t TileLdsBytes(bool fast, uint32_t rcap, uint32_t ring, uint32_t area, uint32_t waves) {
  return kTileSharedBytes + waves * TilePrivateBytes(fast, rcap, ring, area);
}

SPMX_DEVICE TileLds carve_tile(unsigned char *base, bool fast, uint32_t rcap, uint32_t ring, uint32_t area, int wave) {
  TileLds t;
  t.roottab = reinterpret_cast<U4 *>(base);
  t.bcls = base + 256u * 16u;
  unsigned char *mine = base + kTileSharedBytes + static_cast<uint32_t>(wave) * TilePrivateBytes(fast, rcap, ring, area);
  t.raw = mine;
  if (!fast) mine += TileRawBytes(rcap);
  t.ring = reinterpret_cast<float *>(mine);
  t.area = mine + 64u * ring * 4u;
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
  const uint32_t spb = SpByteOf(d);
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
    int mb2 = cs == spb ? 1 : OneCharLenDev(cs);  // :962-963
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

// Ids an unknown piece [tb, tb + len) yields under byte fallback: one per byte of the ORIGINAL text
// (src/sentencepiece_processor.cc:581-603); an unknown piece is one character, and kSpByte stands for three bytes.
SPMX_DEVICE int unk_bytes(const uint8_t *text, int tb, int len, uint32_t spb) { return text[tb] == spb ? 3 : len; }

// Number of ids this lane's sentence produces (:1010-1018 backtrack + the
// unknown-run merge / byte-fallback expansion of sentencepiece_processor.cc:
// 581-613).  -1 on a broken chain.
SPMX_DEVICE int count_lane(const SpmxDev &d, const uint8_t *text, const uint8_t *bp, int nlen, bool active) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const uint32_t spb = SpByteOf(d);
  int n = 0;
  if (!active) return 0;
  int e = nlen;
  bool right_unk = false;
  while (e > 0) {
    const uint32_t b = bp[e];
    const int len = static_cast<int>(b & kBpLen);
    if (len == 0 || len > e) return -1;
    if (b & kBpUnk) {
      if (bf) n += unk_bytes(text, e - len, len, spb);
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

// Writes this lane's n ids to dst[0, n) (forward order, or reversed).  Pieces are visited from the last to
// the first, one per iteration per lane: the piece's bytes are fingerprinted out of LDS (dev.h PieceHash*) and
// its id comes from ONE probe of idtab, which is consumed an iteration later, after the next piece has been
// fingerprinted under its shadow.  Returns false if a chosen piece is not in idtab (cannot happen: it was
// matched in the trie built from the same keys).
SPMX_DEVICE bool write_lane(const SpmxDev &d, const uint8_t *text, const uint8_t *bp, int nlen, int n, int32_t *dst,
                            bool active) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  const U4 *__restrict__ idtab = d.idtab;
  const uint32_t mask = d.idtab_mask;
  int e = nlen, j = n;        // j: forward index one past the next id to write
  bool right_unk = false, ok = true;
  bool pend = false;          // a probe is in flight for the id at forward index pj
  uint32_t pa = 0, pb = 0, pslot = 0;
  int pj = 0;
  U4 pent{0, 0, 0, 0};
  active = active && nlen > 0 && n > 0;
  while (wv::any(active || pend)) {
    bool issue = false;
    uint32_t ha = 0, hb = 0;
    int ij = 0;
    if (active) {
      const uint32_t b = bp[e];
      const int len = static_cast<int>(b & kBpLen);
      const int tb = e - len;
      if (b & kBpUnk) {
        if (bf) {                                   // one BYTE id per byte of the unknown piece (:581-603)
          const bool sp = text[tb] == spb;
          const int nb = sp ? 3 : len;
          for (int x = nb - 1; x >= 0; --x) {
            --j;
            const uint32_t byte = sp ? (x == 0 ? 0xE2u : (x == 1 ? 0x96u : 0x81u)) : text[tb + x];
            dst[reverse ? n - 1 - j : j] = d.byte_ids[byte];
          }
        } else if (!right_unk) {                    // a run of unknown pieces yields one id (:609-613)
          --j;
          dst[reverse ? n - 1 - j : j] = d.unk_id;
        }
        right_unk = true;
      } else {
        right_unk = false;
        PieceHashInit(d.id_seed, &ha, &hb);
        for (int k = tb; k < e; ++k) PieceHashStep(&ha, &hb, text[k]);
        issue = true;
        ij = --j;
      }
      e = tb;
      if (e <= 0) active = false;
    }
    if (pend) {                                     // the probe issued one iteration ago
      int guard = 0;
      while (pent.x != pa || pent.y != pb || pent.z == kSymNone) {
        if (pent.z == kSymNone || ++guard > 64) { ok = false; break; }
        pslot = (pslot + 1) & mask;
        pent = idtab[pslot];
      }
      dst[reverse ? n - 1 - pj : pj] = static_cast<int32_t>(pent.z);
    }
    pend = issue;
    if (issue) {
      pa = ha; pb = hb; pj = ij;
      pslot = PieceHashSlot(ha, hb) & mask;
      pent = idtab[pslot];
    }
  }
  return ok;
}

// Normalize() of one all-ASCII sentence by ONE lane (src/normalizer.cc:71-186 with every NormalizePrefix result
// being the byte itself, :231-244): src[0, L) in HBM -> out[] in LDS (capacity L + 1).  Valid when the space symbol
// is one byte wide (kNfCompressSp, or no whitespace escaping) and the model has no user-defined symbols; bytes
// whose bcls entry says kBcComplex make the lane give up (returns -1) and the sentence goes to the GENERAL kernel.
// Reads 16-byte aligned blocks; a block that holds at least one byte of the sentence lies in the same page as
// that byte, so the over-read at either end stays inside the caller's mapping.
SPMX_DEVICE int fast_norm_lane(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int L, uint8_t *out,
                               const uint8_t *bcls) {
  const uint32_t F = d.flags;
  const bool rm = (F & kNfRemoveExtraWs) != 0;
  const uint32_t sp = (F & kNfCompressSp) ? kSpByte : 0x20u;
  int w = 0;
  if ((F & kNfAddDummyPrefix) && !(F & kNfWsSuffix)) { out[0] = static_cast<uint8_t>(sp); w = 1; }   // :128
  bool P = rm;                    // is_prev_space (:130)
  int wl = w;                     // output length up to the last non-space byte (:166-176 trailing spaces)
  bool seen = false;              // some prefix is not " " (:86-100)
  uint32_t bad = 0;
  const uint64_t q0 = beg & ~15ull;
  const uint8_t *blk = gtext + q0;
  int rel = static_cast<int>(q0 - beg);         // index of the block's first byte within the sentence (<= 0 at first)
  Q4 cur = *reinterpret_cast<const Q4 *>(blk);
  while (rel < L) {
    Q4 nxt = cur;
    if (rel + 16 < L) nxt = *reinterpret_cast<const Q4 *>(blk + 16);
    const uint32_t wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t c = (wd[k >> 2] >> (8 * (k & 3))) & 0xFFu;
      if (static_cast<uint32_t>(rel + k) < static_cast<uint32_t>(L)) {
        bad |= bcls[c];
        const bool is_sp = c == 0x20u;
        if (!is_sp || !P) {                     // :137-138 a space after a space is dropped
          out[w] = static_cast<uint8_t>(is_sp ? sp : c);
          ++w;
        }
        P = is_sp && rm;                        // :154-162
        if (!is_sp) { wl = w; seen = true; }
      }
    }
    cur = nxt;
    blk += 16;
    rel += 16;
  }
  if (bad & kBcComplex) return -1;
  if (rm) {
    if (!seen) return 0;                        // :86-100 nothing but spaces
    w = wl;
  }
  if ((F & kNfAddDummyPrefix) && (F & kNfWsSuffix)) { out[w] = static_cast<uint8_t>(sp); ++w; }   // :179
  return w;
}

// True when fast_norm_lane's preconditions hold for this model (host and device agree on it).
SPMX_HD inline bool TileFastEligible(uint32_t flags) {
  return !(flags & kNfHasUserDefined) && ((flags & kNfCompressSp) || !(flags & kNfEscapeWs));
}

// Segment + ids of one round: every lane with `mine` holds a normalized sentence text[0, my_nlen) whose
// back-pointer bytes bp[0, my_nlen] are zero.
struct TileCounters {
  unsigned long long n_sent = 0, n_raw = 0, n_ids = 0, n_trips = 0;
  unsigned long long cyc[4] = {0, 0, 0, 0};
};

SPMX_DEVICE void tile_segment_and_emit(const EncodeArgs &a, const TileLds &T, float *my_ring, uint32_t rm, bool mine,
                                       const uint8_t *text, uint8_t *bp, int my_nlen, uint32_t my_sid,
                                       uint32_t my_len, int lane, TileCounters *tc) {
  const SpmxDev &d = a.dev;
  const int n_extra = d.n_prefix + d.n_suffix;
  const unsigned long long c1 = wv::clock();
  tc->n_trips += static_cast<unsigned long long>(unigram_lane(d, text, bp, my_nlen, my_ring, rm, T.roottab, mine));
  const unsigned long long c2 = wv::clock();
  int n = count_lane(d, text, bp, my_nlen, mine);
  bool broken = n < 0;
  if (broken) n = 0;
  const int n_out = mine ? n + n_extra : 0;
  int total = 0;
  const int excl = wave_excl_scan(n_out, lane, &total);
  unsigned long long base = 0;
  if (lane == 0 && total > 0) base = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(total));
  base = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(base >> 32), 0)) << 32) |
         wv::shfl(static_cast<uint32_t>(base), 0);
  const bool overflow = base + static_cast<unsigned long long>(total) > a.arena_cap;
  if (overflow) {
    if (lane == 0) wv::atomic_or(a.status, kStArenaOverflow);
  } else {
    int32_t *dst = a.arena + base + static_cast<unsigned long long>(excl);
    if (mine && !broken) {
      for (int x = 0; x < d.n_prefix; ++x) dst[x] = d.prefix_ids[x];
      for (int x = 0; x < d.n_suffix; ++x) dst[d.n_prefix + n + x] = d.suffix_ids[x];
    }
    if (!write_lane(d, text, bp, my_nlen, n, dst + d.n_prefix, mine && !broken)) broken = true;
  }
  if (mine) {
    a.counts[my_sid] = broken ? 0u : static_cast<uint32_t>(n_out);
    a.tmp_off[my_sid] = base + static_cast<unsigned long long>(excl);
  }
  if (wv::any(broken) && lane == 0) wv::atomic_or(a.status, kStInternal);
  const unsigned long long c3 = wv::clock();
  if (mine && !broken) { ++tc->n_sent; tc->n_raw += my_len; tc->n_ids += static_cast<unsigned long long>(n_out); }
  tc->cyc[2] += c2 - c1; tc->cyc[3] += c3 - c2;
}

// Persistent body of both tile kernels; FAST selects the normalizer (see the head of this file).
template <bool FAST>
SPMX_DEVICE void encode_tile_block(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const TileLds T = carve_tile(smem, FAST, a.rcap, a.ring, a.tile_area, wv::wave_in_block());
  const uint32_t rm = a.ring - 1;
  float *my_ring = T.ring + lane;
  {   // Shared read-only tables.  Every wave of the workgroup writes all of both (same values), so a wave only
      // has to wait for its own stores: no workgroup barrier.
    const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
    for (uint32_t cb = static_cast<uint32_t>(lane); cb < 256u; cb += 64u) {
      U4 r = d.ptrie[root ^ cb];      // unit of byte cb, or an empty unit if no piece starts with cb
      if ((r.x & 0x1FFu) != (0x100u | cb)) r = U4{0, 0, 0, 0};
      T.roottab[cb] = r;
      const bool safe = cb < 128u && ((d.ascii_safe[cb >> 5] >> (cb & 31u)) & 1u);
      T.bcls[cb] = static_cast<uint8_t>(safe ? 0u : kBcComplex);
    }
    wv::sync();
  }
  const uint32_t count = *a.list_count;
  const uint32_t wave_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block());
  const uint32_t n_waves = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block());
  // Sentences per tile: 64, or fewer when the list is too short to give every wave a full tile (a short list
  // -- typically what the FAST kernel handed over -- is latency-bound per tile, so it is spread over all waves).
  uint32_t tw = (count + n_waves - 1) / n_waves;
  tw = tw < 1u ? 1u : (tw > 64u ? 64u : tw);
  const uint32_t tiles = (count + tw - 1) / tw;
  TileCounters tc;
  for (uint32_t tile = wave_id; tile < tiles; tile += n_waves) {
    const uint32_t first = tile * tw;
    const int cnt = static_cast<int>(count - first < tw ? count - first : tw);
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
      const unsigned long long c0 = wv::clock();
      unsigned long long t_load = 0;
      int my_off = -1, my_nlen = 0;     // my_off < 0: this lane has no sentence in this round
      int i1 = cnt;
      if (FAST) {
        // ---- round: the longest run of lanes i0.. whose slots (2 (L + 1) + 1 bytes) fit the area ----
        const bool cand = lane >= i0 && lane < cnt;
        const bool too_long = cand && (my_len == 0xFFFFFFFFu || 2u * (my_len + 1u) + 1u > a.tile_area);
        const uint32_t need = (cand && !too_long) ? 2u * (my_len + 1u) + 1u : 0u;
        int total = 0;
        const uint32_t off = static_cast<uint32_t>(wave_excl_scan(static_cast<int>(need), lane, &total));
        const uint64_t fits = wv::ballot(!cand || off + need <= a.tile_area);
        const uint64_t nofit = ~fits & ~((1ull << i0) - 1ull);
        if (nofit) i1 = wv::ffs64(nofit) - 1;   // > i0: a single slot always fits (api.cc sizes the area)
        const bool in_round = cand && lane < i1;
        const uint32_t used = static_cast<uint32_t>(wv::shfl(static_cast<int>(off), i1 < 64 ? i1 : 63)) +
                              (i1 < 64 ? 0u : static_cast<uint32_t>(wv::shfl(static_cast<int>(need), 63)));
        if (wv::any(in_round && too_long)) {
          uint64_t m = wv::ballot(in_round && too_long);
          while (m) { const int i = wv::ffs64(m) - 1; m &= m - 1; fail_sentence(a, wv::shfl(my_sid, i), kStTooLong, lane); }
        }
        for (uint32_t p = static_cast<uint32_t>(lane) * 4u; p < used; p += 256u)
          *reinterpret_cast<uint32_t *>(T.area + p) = 0u;
        wv::sync();
        int nlen = 0;
        const bool go = in_round && !too_long;
        if (go && my_len > 0) nlen = fast_norm_lane(d, a.text, my_beg, static_cast<int>(my_len), T.area + off, T.bcls);
        const bool hard = go && nlen < 0;
        if (go && nlen >= 0) {
          my_off = static_cast<int>(off);
          my_nlen = nlen;
          T.area[off + static_cast<uint32_t>(nlen)] = 0;   // bp[0]: a trimmed trailing space symbol may sit here
        }
        const uint64_t hm = wv::ballot(hard);
        if (hm) {                               // hand the sentence to the GENERAL kernel of this class
          const int leader = wv::ffs64(hm) - 1;
          uint32_t hb = 0;
          if (lane == leader) hb = wv::atomic_add(a.hard_count, static_cast<uint32_t>(wv::popc64(hm)));
          hb = wv::shfl(hb, leader);
          if (hard) a.hard_list[hb + static_cast<uint32_t>(wv::popc64(hm & ((1ull << lane) - 1ull)))] = my_sid;
        }
        wv::sync();
      } else {
        // ---- round: pack sentences i0.. into the area until it is full ----
        uint32_t used = 0;
        int i = i0;
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
        i1 = i;
        wv::sync();
      }
      const unsigned long long c1 = wv::clock();
      tc.cyc[0] += t_load; tc.cyc[1] += (c1 - c0) - t_load;
      const bool mine = my_off >= 0;
      const uint8_t *text = T.area + (mine ? my_off : 0);
      uint8_t *bp = T.area + (mine ? my_off + my_nlen : 0);
      tile_segment_and_emit(a, T, my_ring, rm, mine, text, bp, my_nlen, my_sid, my_len, lane, &tc);
      i0 = i1;
    }
  }
  if (a.stats) {
    // per-lane sentence counters -> wave totals
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
