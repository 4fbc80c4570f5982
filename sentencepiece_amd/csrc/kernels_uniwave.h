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
//   fold   the relaxations of best_path_ends_at in the reference's order: starts ascending one after another, a start's
//          candidates (distinct end positions) side by side, then the UNK candidate (:995-1005); same arithmetic -- double add, double compare
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

constexpr uint32_t kUwRing = 256;       // score / back-pointer rings: 64 starts + the longest piece (<= kMaxPieceBytes) + slack
constexpr uint32_t kUwWindow = 256;     // bytes of text staged per chunk: 64 starts + the longest piece
constexpr uint32_t kUwMaxCands = 32;    // candidate rows hold max_prefixes entries (<= this)
constexpr uint32_t kUwUnreached = 0xFFFFFFFFu;
SPMX_HD inline uint32_t UniWaveLdsBytes(uint32_t J) {
  return kUwRing * 8u + 64u * J * 8u + 64u + kUwWindow + 16u + kRawWinBytes;
}

struct UniWaveLds {
  float *ring_s;      // [kUwRing] best_path_score of position p at ring_s[p % kUwRing]
  uint32_t *ring_b;   // [kUwRing] its best piece: id | length << 24 (kUwUnreached: no candidate yet); doubles as the
                      //           backtrack's window of blen entries
  U2 *cands;          // [64][J]  {id | length << 24 | user-defined << 31, score bits}
  uint8_t *ncand;     // [64]
  uint8_t *win;       // [kUwWindow + 16] text window
  uint8_t *rawwin;    // [kRawWinBytes] lane 0's raw-text window of norm_lane_any
};
SPMX_DEVICE UniWaveLds carve_uniwave(unsigned char *smem, uint32_t J) {
  UniWaveLds T;
  T.ring_s = reinterpret_cast<float *>(smem);
  T.ring_b = reinterpret_cast<uint32_t *>(smem + kUwRing * 4u);
  T.cands = reinterpret_cast<U2 *>(smem + kUwRing * 8u);
  T.ncand = smem + kUwRing * 8u + 64u * J * 8u;
  T.win = T.ncand + 64u;
  T.rawwin = T.win + kUwWindow + 16u;
  return T;
}

// EncodeOptimized of the normalized text nt[0, nlen) (device form, HBM) by one wavefront.  bid / blen: nlen + 2
// entries in HBM, written here.  On return blen[e] has kTokEnd | length at every token end of the best path and bid[e]
// the token's id (unk_id for an unknown character).  false: a broken chain (cannot happen).
// The fold touches LDS only: scores and back-pointers of the 256 most recent positions live in rings; after the chunk
// of starts [c, c + 64) has been folded the positions up to c + 64 are final and leave for HBM as one coalesced row.
SPMX_DEVICE bool unigram_wave(const SpmxDev &d, const uint8_t *nt, int nlen, const UniWaveLds &T, uint32_t J, int32_t *bid,
                              uint16_t *blen, int lane) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  const uint32_t root_w = d.ptrie[0].w;
  const uint32_t spb = SpByteOf(d);
  for (uint32_t k = static_cast<uint32_t>(lane); k < kUwRing; k += 64u) T.ring_b[k] = kUwUnreached;
  if (lane == 0) T.ring_s[0] = 0.f;                                   // best_path_ends_at[0].best_path_score = 0
  int next_start = 0;                                                // the next character start (absolute), across chunks
  for (int c = 0; c < nlen; c += 64) {
    // ---- the text of this chunk's walks: positions [c, c + kUwWindow) ----
    wv::sync();                                                      // (the previous chunk's walks, fold and flush are done)
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
      uint32_t wsum = root_w;                                         // child-label summary of the node the walk stands on (dev.h ChildBit)
      int dep = 0;
      while (wv::any(alive)) {
        if (alive) {
          const int q = s + dep;
          if (q >= nlen) { alive = false; }
          else if (!((wsum >> ChildBit(T.win[q - c])) & 1u)) { alive = false; }   // no child with this label: the failing probe is not issued
          else {
            const uint32_t cb = T.win[q - c];
            const U4 u = ptrie[node ^ cb];
            if ((u.x & 0x1FFu) == (0x100u | cb)) {                    // :969-971
              ++dep;
              node = u.x >> kDatBaseShiftDev;
              wsum = u.w;
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
    // ---- fold: the relaxations in the reference's order.  The starts one after another (a start's score must be
    // final before its candidates are scored); the candidates of ONE start end at different positions, so lane k takes
    // candidate k and lane 63 the UNK candidate (:995-1005, only when no piece of one character matched, so its end
    // position is no candidate's either) ----
    for (uint64_t m = S; m != 0; m &= m - 1) {
      const int l = wv::ffs64(m) - 1;
      const int ss = c + l;
      const float bs = T.ring_s[static_cast<uint32_t>(ss) & (kUwRing - 1u)];
      const uint32_t b0 = T.win[l];
      int mb = b0 == spb ? 1 : OneCharLenDev(b0);
      if (mb > nlen - ss) mb = nlen - ss;
      const uint32_t nk = T.ncand[l];
      const bool has = static_cast<uint32_t>(lane) < nk;
      int len = 0;
      if (has) {
        const U2 cw = T.cands[static_cast<uint32_t>(l) * J + static_cast<uint32_t>(lane)];
        len = static_cast<int>((cw.x >> 24) & 0x7Fu);
        const uint32_t es = static_cast<uint32_t>(ss + len) & (kUwRing - 1u);
        double score = static_cast<double>(wv::bits_to_float(cw.y));
        if (cw.x & 0x80000000u) {                                     // (length * max_score_ - 0.1), :979-981
          const float prod = static_cast<float>(len) * d.max_score;
          score = static_cast<double>(prod) - 0.1;
        }
        const double cand = score + static_cast<double>(bs);         // :982-983
        if (T.ring_b[es] == kUwUnreached || cand > static_cast<double>(T.ring_s[es])) {      // :984-989
          T.ring_s[es] = static_cast<float>(cand);
          T.ring_b[es] = cw.x & 0x7FFFFFFFu;
        }
      }
      const bool single = wv::any(has && len == mb);                  // :990
      if (!single && lane == 63) {                                    // :995-1005, float arithmetic
        const uint32_t es = static_cast<uint32_t>(ss + mb) & (kUwRing - 1u);
        const float cand = d.unk_score + bs;
        if (T.ring_b[es] == kUwUnreached || cand > T.ring_s[es]) {
          T.ring_s[es] = cand;
          T.ring_b[es] = (static_cast<uint32_t>(d.unk_id) & 0x00FFFFFFu) | (static_cast<uint32_t>(mb) << 24);
        }
      }
      wv::sync();
    }
    wv::sync();
    // ---- positions (c, c + 64] are final: one coalesced row to HBM, their ring slots are free for c + 256 ... ----
    {
      const int p = c + 1 + lane;
      if (p <= nlen) {
        const uint32_t sl = static_cast<uint32_t>(p) & (kUwRing - 1u);
        const uint32_t w = T.ring_b[sl];
        bid[p] = w == kUwUnreached ? 0 : static_cast<int32_t>(w & 0x00FFFFFFu);
        blen[p] = w == kUwUnreached ? static_cast<uint16_t>(0) : static_cast<uint16_t>((w >> 24) & 0x7Fu);
        T.ring_b[sl] = kUwUnreached;
      }
    }
  }
  wv::sync_global();
  // ---- backtrack (:1010-1018) through windows of 256 blen entries staged in LDS (the idle back-pointer ring) ----
  uint32_t ok = 1;
  {
    int e = nlen;                                                     // wave-uniform: broadcast from lane 0 after every window
    while (e > 0) {
      const int lo = e > static_cast<int>(kUwRing) - 1 ? e - (static_cast<int>(kUwRing) - 1) : 0;   // window [lo, e]
      wv::sync();
      for (int p = lo + lane; p <= e; p += 64) T.ring_b[p - lo] = blen[p];
      wv::sync();
      int e2 = e;
      if (lane == 0) {
        while (e2 > lo) {
          const int len = static_cast<int>(T.ring_b[e2 - lo] & 0x7FFFu);
          if (len == 0 || len > e2) { ok = 0; e2 = 0; break; }
          blen[e2] = static_cast<uint16_t>(kTokEnd | static_cast<uint32_t>(len));
          if (e2 - len < lo) { e2 -= len; break; }                    // (cannot happen: len < 256 - 64; kept for safety)
          e2 -= len;
        }
      }
      e2 = wv::shfl(e2, 0);
      if (e2 >= e) { ok = 0; break; }                                 // no progress: broken
      e = e2;
    }
  }
  ok = wv::shfl(ok, 0);
  return ok != 0;
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
  unsigned long long st_sent = 0, st_raw = 0, st_ids = 0, st_cyc[3] = {0, 0, 0};
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
    const bool ok = unigram_wave(d, norm, nlen, T, J, bid, blen, lane);
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
    wv::atomic_add(&a.stats[4], st_cyc[0]);       // normalize
    wv::atomic_add(&a.stats[5], st_cyc[1]);       // segment
    wv::atomic_add(&a.stats[6], st_cyc[2]);       // emit
  }
}

}  // namespace spmx
#endif
