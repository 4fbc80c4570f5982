// Unigram segmentation, WAVE-COOPERATIVE form: one SENTENCE PER WAVEFRONT, exact for any model and any length.
// Reference: unigram::Model::EncodeOptimized (src/unigram_model.cc:889-1020).
//
// The lane-per-sentence kernels (kernels_stream.h, kernels_word.h) need tens of thousands of sentences to fill the
// chip and give one sentence a single lane: a document of a megabyte is 2.6 s of dependent iterations there.  This form
// gives a sentence the whole wavefront, 64 consecutive character starts at a time:
//
//   walk   (parallel) lane l walks the piece trie from start c + l -- the reference's inner loop (:965-993) -- and
//          writes what it finds, piece by piece in order of length, into its row of an LDS candidate list.  Which
//          pieces match at a start does not depend on any score, so the 64 walks are independent;
//   fold   (one lane) the relaxations of best_path_ends_at in the reference's order: starts ascending, for a start its
//          candidates by length, then the UNK candidate (:995-1005); same arithmetic -- double add, double compare
//          against the float stored, float store (:979-989).  The scores of the last 256 positions live in an LDS
//          ring; id and length of every position's best piece go to the sentence's bid / blen arrays;
//   then the backtrack (:1010-1018) marks the token ends and emit_wave (kernels.h) writes the ids.
//
// The sentence's arrays (normalized text, bid, blen) live in a slice of the long form's HBM pool (kernels_long.h), so
// nothing here depends on the sentence length; normalization is norm_lane_any (any normalizer_spec) by one lane.
// Used for documents (length classes beyond 4 KiB) of every unigram model, and for what the word kernels leave when
// that is too little to fill the lane-per-sentence kernel.
#ifndef SPMX_KERNELS_UNIWAVE_H_
#define SPMX_KERNELS_UNIWAVE_H_

namespace spmx {

constexpr uint32_t kUwRing = 256;       // score ring: 64 starts + the longest piece (<= kMaxPieceBytes) + slack
constexpr uint32_t kUwWindow = 256;     // bytes of text staged per chunk: 64 starts + the longest piece
constexpr uint32_t kUwMaxCands = 32;    // candidate rows hold SpmxDev-independent max_prefixes entries (<= this)
SPMX_HD inline uint32_t UniWaveLdsBytes(uint32_t J) {
  return kUwRing * 4u + 64u * J * 8u + 64u + kUwWindow + 16u + kRawWinBytes;
}

struct UniWaveLds {
  float *ring_s;      // [kUwRing] best_path_score of position p at ring_s[p % kUwRing]
  U2 *cands;          // [64][J]  {id | length << 24 | user-defined << 31, score bits}
  uint8_t *ncand;     // [64]
  uint8_t *win;       // [kUwWindow + 16] text window
  uint8_t *rawwin;    // [kRawWinBytes] lane 0's raw-text window of norm_lane_any
};
SPMX_DEVICE UniWaveLds carve_uniwave(unsigned char *smem, uint32_t J) {
  UniWaveLds T;
  T.ring_s = reinterpret_cast<float *>(smem);
  T.cands = reinterpret_cast<U2 *>(smem + kUwRing * 4u);
  T.ncand = smem + kUwRing * 4u + 64u * J * 8u;
  T.win = T.ncand + 64u;
  T.rawwin = T.win + kUwWindow + 16u;
  return T;
}

// EncodeOptimized of the normalized text nt[0, nlen) (device form, HBM) by one wavefront.  bid / blen: nlen + 1
// entries, blen zeroed by the caller.  On return blen[e] has kTokEnd | length at every token end of the best path and
// bid[e] the token's id (unk_id for an unknown character).  false: a broken chain (cannot happen).
SPMX_DEVICE bool unigram_wave(const SpmxDev &d, const uint8_t *nt, int nlen, const UniWaveLds &T, uint32_t J, int32_t *bid,
                              uint16_t *blen, int lane) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  const uint32_t spb = SpByteOf(d);
  if (lane == 0) T.ring_s[0] = 0.f;                                   // best_path_ends_at[0].best_path_score = 0
  int next_start = 0;                                                // the next character start (absolute), across chunks
  for (int c = 0; c < nlen; c += 64) {
    // ---- the text of this chunk's walks: positions [c, c + kUwWindow) ----
    wv::sync();                                                      // (the previous chunk's walks and fold are done)
    {
      const int q = c + 4 * lane;
      uint32_t v = 0;
      if (q < nlen) v = *reinterpret_cast<const uint32_t *>(nt + q);   // (the slice is padded: a whole dword is readable)
      *reinterpret_cast<uint32_t *>(T.win + 4 * lane) = v;
    }
    wv::sync();
    const int s = c + lane;
    const bool valid = s < nlen;
    int step = 1;
    if (valid) {
      const uint32_t b0 = T.win[lane];
      step = b0 == spb ? 1 : OneCharLenDev(b0);                       // :962-963
      if (step > nlen - s) step = nlen - s;
    }
    const uint64_t S = resolve_chain(c, step, valid, &next_start);   // which positions are character starts (:1007)
    // ---- walk (:965-993): every piece that starts at s, in order of length ----
    {
      bool alive = ((S >> lane) & 1ull) != 0;
      uint32_t node = root, k = 0;
      int dep = 0;
      while (wv::any(alive)) {
        if (alive) {
          const int q = s + dep;
          if (q >= nlen) { alive = false; }
          else {
            const uint32_t cb = T.win[q - c];
            const U4 u = ptrie[node ^ cb];
            if ((u.x & 0x1FFu) == (0x100u | cb)) {                    // :969-971
              ++dep;
              node = u.x >> kDatBaseShiftDev;
              if ((u.x & kDatTerminalDev) && !(u.y & kPtUnused) && k < J) {   // :973-974
                T.cands[static_cast<uint32_t>(lane) * J + k] =
                    U2{(u.y & 0x00FFFFFFu) | (static_cast<uint32_t>(dep) << 24) | ((u.y & kPtUserDefined) ? 0x80000000u : 0u), u.z};
                ++k;
              }
              if (dep >= static_cast<int>(kUwWindow) - 64) alive = false;   // (no piece is that long)
            } else {
              alive = false;
            }
          }
        }
      }
      T.ncand[lane] = static_cast<uint8_t>(k);
    }
    wv::sync();
    // ---- fold: the relaxations in the reference's order ----
    if (lane == 0) {
      for (uint64_t m = S; m != 0; m &= m - 1) {
        const int l = wv::ffs64(m) - 1;
        const int ss = c + l;
        const float bs = T.ring_s[static_cast<uint32_t>(ss) & (kUwRing - 1u)];
        const uint32_t b0 = T.win[l];
        int mb = b0 == spb ? 1 : OneCharLenDev(b0);
        if (mb > nlen - ss) mb = nlen - ss;
        bool single = false;
        const uint32_t nk = T.ncand[l];
        for (uint32_t k = 0; k < nk; ++k) {
          const U2 cw = T.cands[static_cast<uint32_t>(l) * J + k];
          const int len = static_cast<int>((cw.x >> 24) & 0x7Fu);
          const int e = ss + len;
          double score = static_cast<double>(wv::bits_to_float(cw.y));
          if (cw.x & 0x80000000u) {                                   // (length * max_score_ - 0.1), :979-981
            const float prod = static_cast<float>(len) * d.max_score;
            score = static_cast<double>(prod) - 0.1;
          }
          const double cand = score + static_cast<double>(bs);       // :982-983
          float *slot = &T.ring_s[static_cast<uint32_t>(e) & (kUwRing - 1u)];
          if (blen[e] == 0 || cand > static_cast<double>(*slot)) {    // :984-989
            *slot = static_cast<float>(cand);
            bid[e] = static_cast<int32_t>(cw.x & 0x00FFFFFFu);
            blen[e] = static_cast<uint16_t>(len);
          }
          if (len == mb) single = true;                               // :990
        }
        if (!single) {                                                // :995-1005, float arithmetic
          const int e = ss + mb;
          const float cand = d.unk_score + bs;
          float *slot = &T.ring_s[static_cast<uint32_t>(e) & (kUwRing - 1u)];
          if (blen[e] == 0 || cand > *slot) {
            *slot = cand;
            bid[e] = d.unk_id;
            blen[e] = static_cast<uint16_t>(mb);
          }
        }
      }
    }
  }
  wv::sync();
  // ---- backtrack (:1010-1018) ----
  uint32_t ok = 1;
  if (lane == 0) {
    int e = nlen;
    while (e > 0) {
      const int len = blen[e];
      if (len == 0 || len > e) { ok = 0; break; }
      blen[e] = static_cast<uint16_t>(kTokEnd | static_cast<uint32_t>(len));
      e -= len;
    }
  }
  return wv::shfl(ok, 0) != 0;
}

// One sentence per wavefront over a device-side list; slices from the long form's pool (LongArgs, kernels_long.h).
SPMX_DEVICE void uni_long_block(const LongArgs &a, unsigned char *smem, uint32_t J) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const UniWaveLds T = carve_uniwave(smem, J);
  const uint32_t wave_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block());
  const uint32_t n_waves = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block());
  const uint32_t count = *a.list_count;
  const int n_extra = d.n_prefix + d.n_suffix;
  for (uint32_t i = wave_id; i < count; i += n_waves) {
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
    // ---- normalized length first (a count-only pass by lane 0), then a slice of the size it takes ----
    int nlen = 0;
    if (lane == 0) {
      int nsp = 0;
      FlatSink cs{nullptr, nullptr, 0};
      nlen = norm_lane_any(d, a.text, beg, L, cs, T.rawwin, &nsp);
    }
    nlen = wv::shfl(nlen, 0);
    if (nlen == 0) {                                                  // (empty, or nothing but whitespace)
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
      continue;
    }
    const uint64_t b_text = Align16(static_cast<uint64_t>(nlen) + kUwWindow + 16);
    const uint64_t b_bid = Align16((static_cast<uint64_t>(nlen) + 2) * 4);
    const uint64_t b_len = Align16((static_cast<uint64_t>(nlen) + 2) * 2);
    const uint64_t need = b_text + b_bid + b_len;
    unsigned long long at = 0;
    if (lane == 0) at = wv::atomic_add(a.pool_head, static_cast<unsigned long long>(need));
    at = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(at >> 32), 0)) << 32) | wv::shfl(static_cast<uint32_t>(at), 0);
    if (at + need > a.pool_cap) {                                     // the host grows the pool and launches again
      if (lane == 0) { a.retry_list[wv::atomic_add(a.retry_count, 1u)] = sid; a.counts[sid] = 0u; }
      continue;
    }
    uint8_t *norm = a.pool + at;
    int32_t *bid = reinterpret_cast<int32_t *>(norm + b_text);
    uint16_t *blen = reinterpret_cast<uint16_t *>(norm + b_text + b_bid);
    if (lane == 0) {
      FlatSink ws{norm, nullptr, nlen};
      int nsp2 = 0;
      norm_lane_any(d, a.text, beg, L, ws, T.rawwin, &nsp2);
    }
    for (int e = lane; e <= nlen + 1; e += 64) blen[e] = 0;
    wv::sync_global();
    const bool ok = unigram_wave(d, norm, nlen, T, J, bid, blen, lane);
    wv::sync_global();
    if (!ok) {                                                        // "all normalized characters are not consumed."
      if (lane == 0) {
        a.counts[sid] = 0; a.tmp_off[sid] = 0; a.sent_status[sid] = static_cast<uint8_t>(kSsInternal);
        wv::atomic_add(&a.side->n_failed, 1ull);
      }
      continue;
    }
    emit_wave(a, sid, norm, nlen, bid, blen, lane);
  }
}

}  // namespace spmx
#endif
