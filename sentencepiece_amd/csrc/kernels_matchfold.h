// Unigram segmentation, SPLIT form: the two halves of EncodeOptimized (src/unigram_model.cc:957-1008) taken apart.
//
// The reference's loop does two things per character start: it WALKS the piece trie along the text (:965-971; which
// pieces match at a start depends on the text only) and it FOLDS every match into best_path_ends_at (:973-1005; a
// start's score must be final before its matches are scored).  The lane-per-sentence kernel (kernels_stream.h) does
// both per lane, so a sentence is one chain of dependent trie probes: 4 KB of mixed-script text is ~10,000 probes of
// ~1.5 us behind each other, whatever else the chip is doing (BASELINE configs[4]: the 4 % longest sentences took the
// whole step).  Here:
//
//   match  (wave-cooperative, one sentence at a time) the sentence is normalized position-parallel in LDS
//          (normalize_wave, kernels.h), its character starts are queued, and 64 x kMfWalks starts walk the trie side by
//          side, a start per lane -- the chain of dependent probes is one PIECE long, not one sentence.  What a walk
//          finds goes, in the reference's order (starts ascending, a start's pieces by length, then its UNK candidate
//          :995-1005), to the sentence's CANDIDATE STREAM in HBM: 8 bytes per candidate {id, length, flags; score};
//   fold   (a sentence per lane) reads its stream front to back and applies the reference's relaxations with the
//          reference's arithmetic (double add, double compare against the float stored, float store :979-989; the UNK
//          candidate in float :997-1001) to the same LDS rings and back-pointer blocks as the lane-per-sentence kernel.
//          No probe sits in this chain: the stream is read ahead in 64-byte blocks;
//   then the backtrack and the ids as before (emit_stream_lane, kernels_stream.h).
//
// Exact for the models it takes (the folds are the reference's, in the reference's order); the loader decides
// (tables.cc SplitEligible): unigram, no user-defined pieces, every charsmap replacement valid UTF-8 (so that the
// character starts of the normalized text are its non-continuation bytes), ids below 2^21.  A sentence whose stream
// outgrows its slab share goes to the call's overflow launch (the lane-per-sentence kernel), like one whose text does.
#ifndef SPMX_KERNELS_MATCHFOLD_H_
#define SPMX_KERNELS_MATCHFOLD_H_

namespace spmx {

// candidate word x: id (19 bits) | the piece's length in CHARACTERS << 19 (4 bits) | (bytes of the start's character - 1) << 23
// | the piece's length in bytes << 25 (6 bits) | first-of-its-start << 31; y: score bits
constexpr uint32_t kCsIdMask = 0x0007FFFFu;
constexpr int kCsNchShift = 19;
constexpr int kCsMbShift = 23;
constexpr int kCsLenShift = 25;
constexpr uint32_t kCsFirst = 0x80000000u;
constexpr uint32_t kMfMaxVocab = 1u << 19;
constexpr int kMfMaxPieceBytes = 63;     // (six bits)
constexpr int kMfMaxPieceChars = 15;     // (four bits); the fold's rings are indexed by CHARACTER: 8 or 16 entries
constexpr uint32_t kMfMaxCands = 16;     // candidate rows: the deepest chain of pieces that are prefixes of one another
constexpr uint32_t kMfQueue = 512;       // queued character starts (a sweep adds up to 256 to fewer than 64 * kMfWalks)
constexpr int kMfWalks = 2;              // starts a lane walks side by side: two probes in flight per lane (four: the same cycles per start -- the gathers are bound by their throughput, scripts/ubench/gather_probe -- and fewer wavefronts for the LDS rows)
constexpr uint32_t kMfMaxRaw = 4096;     // sentences beyond this do not fit the LDS image (normalize_wave's LDS form)

// LDS of the match phase for sentences of up to rcap raw / tcap normalized bytes, rows of J candidates
// (the queue and the candidate rows lie OVER the raw image: it is dead once the sentence is normalized)
SPMX_HD inline uint32_t MatchLdsBytes(uint32_t rcap, uint32_t tcap, uint32_t J) {
  const uint32_t raw = (rcap + 48u + 15u) & ~15u, walk = kMfQueue * 2u + 64u * kMfWalks * J * 8u;
  return ((tcap + 16u + 15u) & ~15u) + (raw > walk ? raw : walk);
}
// LDS of the fold phase: the two rings of RC characters and the staging block of 8 back-pointer entries, per lane
SPMX_HD inline uint32_t FoldLdsBytes(uint32_t rc, uint32_t bpsz) { return 64u * rc * (4u + bpsz) + 64u * 8u * bpsz; }
// ring entries for pieces of up to max_chars characters
SPMX_HD inline uint32_t FoldRing(int max_chars) { return max_chars < 8 ? 8u : 16u; }
// candidates a sentence of tcap normalized bytes may put into its stream (entries), and the stream's bytes in the slab
// (the fold reads whole blocks of 8 entries, two blocks ahead)
SPMX_HD inline uint32_t MatchStreamCap(uint32_t tcap, uint32_t per_byte) { return (tcap * per_byte + 64u + 7u) & ~7u; }
SPMX_HD inline uint64_t MatchStreamBytes(uint32_t ccap) { return (static_cast<uint64_t>(ccap) + 32u) * 8u; }

struct MatchLds {
  uint8_t *norm;      // [tcap + 16]
  uint8_t *raw;       // [rcap + 48]: whole 16-byte units of the text; the sentence begins at raw + (its address & 15)
  uint16_t *queue;    // [kMfQueue] byte positions of character starts, in text order (over raw)
  U2 *cands;          // [kMfWalks][64][J] (over raw, behind the queue)
};
SPMX_DEVICE MatchLds carve_match(unsigned char *mine, uint32_t rcap, uint32_t tcap, uint32_t) {
  MatchLds m;
  (void)rcap;
  m.norm = mine;
  m.raw = mine + ((tcap + 16u + 15u) & ~15u);
  m.queue = reinterpret_cast<uint16_t *>(m.raw);             // (over the raw image, see MatchLdsBytes)
  m.cands = reinterpret_cast<U2 *>(m.raw + kMfQueue * 2u);
  return m;
}

// The candidate stream of the normalized text norm[0, nlen) (LDS, device form, readable as dwords up to nlen + 3):
// entries in the reference's order into out[0, ccap).  Returns their number, -1 when they do not fit.  *nsp: bytes of
// the text that are the one-byte space symbol.
SPMX_DEVICE int match_wave(const SpmxDev &d, const uint8_t *norm, int nlen, const U4 *roottab, const MatchLds &M, uint32_t J,
                           U2 *out, int ccap, int lane, int *nsp) {
  const U4 *__restrict__ ptrie = d.ptrie;
  const U4 *__restrict__ cfirst = d.cfirst;
  const uint32_t spb = SpByteOf(d);
  const uint32_t unk_word = static_cast<uint32_t>(d.unk_id) & kCsIdMask;
  const uint32_t unk_bits = wv::float_to_bits(d.unk_score);
  int n_out = 0, scan_pos = 0, n_sp = 0;
  uint32_t qhead = 0, qn = 0;                    // wave-uniform
  for (;;) {
    // ---- queue the character starts of the next 256 bytes: the bytes that continue no character (the text is valid
    // UTF-8 plus the one-byte space symbol; unigram_model.cc:962-963, :1007 hop by the lead byte's length) ----
    while (qn < 64u * kMfWalks && scan_pos < nlen) {
      const int p = scan_pos + 4 * lane;
      uint32_t v = 0;
      if (p < nlen) v = *reinterpret_cast<const uint32_t *>(norm + p);
      uint32_t m = 0, sp = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t c = (v >> (8 * b)) & 0xFFu;
        if (p + b < nlen) {
          if ((c & 0xC0u) != 0x80u) m |= 1u << b;
          if (c == spb) ++sp;
        }
      }
      const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(m));
      const uint32_t packed = wv::scan_add(cnt | (sp << 16));
      const uint32_t total = wv::read_lane(packed, 63);
      uint32_t at = qhead + qn + (packed & 0xFFFFu) - cnt;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if ((m >> b) & 1u) { M.queue[at & (kMfQueue - 1u)] = static_cast<uint16_t>(p + b); ++at; }
      qn += total & 0xFFFFu;
      n_sp += static_cast<int>(total >> 16);
      scan_pos += 256;
      wv::sync();
    }
    if (qn == 0) break;
    // ---- kMfWalks x 64 starts, a start per lane and walk (:965-993): every piece that begins there, by length ----
    uint32_t take[kMfWalks];
    int p0[kMfWalks], dep[kMfWalks], mb[kMfWalks];
    uint32_t nch[kMfWalks];                          // characters of the text the walk has consumed
    uint32_t k[kMfWalks];
    bool alive[kMfWalks], single[kMfWalks], has[kMfWalks];
    U4 u[kMfWalks];
#pragma unroll
    for (int w = 0; w < kMfWalks; ++w) {
      take[w] = qn < 64u ? qn : 64u;
      has[w] = static_cast<uint32_t>(lane) < take[w];
      p0[w] = has[w] ? static_cast<int>(M.queue[(qhead + static_cast<uint32_t>(lane)) & (kMfQueue - 1u)]) : 0;
      qhead += take[w];
      qn -= take[w];
      k[w] = 0; dep[w] = 0; mb[w] = 1; nch[w] = 1; single[w] = false; alive[w] = false;
      u[w] = U4{0, 0, 0, 0};
      if (has[w]) {
        const uint32_t b0 = norm[p0[w]];
        const U4 r = roottab[b0];                  // first byte's unit; its label byte carries the character's length
        int m1 = static_cast<int>(r.x & 7u);
        if (m1 < 1) m1 = b0 == spb ? 1 : OneCharLenDev(b0);     // (bytes no piece starts with have an all-zero entry)
        if (m1 > nlen - p0[w]) m1 = nlen - p0[w];
        mb[w] = m1;
        u[w] = r;
        dep[w] = 1;
        alive[w] = (r.x & 0x100u) != 0;
        if (cfirst != nullptr && b0 >= 0xC2u && b0 < 0xF0u) {    // the whole first character in one probe (dev.h cfirst)
          const bool three = b0 >= 0xE0u;
          const int dch = three ? 3 : 2;
          if (p0[w] + dch <= nlen) {
            const uint32_t b1 = norm[p0[w] + 1], b2 = norm[p0[w] + 2];
            const uint32_t cp = three ? ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu) : ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
            const bool wf = (b1 & 0xC0u) == 0x80u && (!three || ((b2 & 0xC0u) == 0x80u && cp >= 0x800u));
            if (wf) { u[w] = cfirst[cp]; dep[w] = dch; alive[w] = (u[w].x & 0x100u) != 0; }
          }
        }
      }
    }
    U2 *row[kMfWalks];
#pragma unroll
    for (int w = 0; w < kMfWalks; ++w) row[w] = M.cands + (static_cast<uint32_t>(w) * 64u + static_cast<uint32_t>(lane)) * J;
    bool any_alive = false;
#pragma unroll
    for (int w = 0; w < kMfWalks; ++w) any_alive = any_alive || alive[w];
    while (wv::any(any_alive)) {
      // control: consume the unit each walk stands on, name the next probe; then the probes of all walks fly together
      uint32_t nxt[kMfWalks], cb[kMfWalks];
      bool go[kMfWalks];
#pragma unroll
      for (int w = 0; w < kMfWalks; ++w) {
        go[w] = false; nxt[w] = 0; cb[w] = 0;
        if (alive[w]) {
          const U4 uu = u[w];
          const int q = p0[w] + dep[w];
          const uint32_t cq = q < nlen ? norm[q] : 0u;                               // the byte behind what has matched (0: the text ends)
          // a piece that ends INSIDE a character (a model whose pieces are not whole characters) only ever relaxes a position
          // no start reaches (:1007 hops by characters) and the backtrack never visits: it is left out, and the fold may
          // count positions in characters
          if ((uu.x & kDatTerminalDev) && !(uu.y & kPtUnused) && (cq & 0xC0u) != 0x80u) {   // :973-974
            if (k[w] < J)
              row[w][k[w]] = U2{(uu.y & kCsIdMask) | (nch[w] << kCsNchShift) | (static_cast<uint32_t>(dep[w]) << kCsLenShift), uu.z};
            ++k[w];
            if (dep[w] == mb[w]) single[w] = true;                                   // :990
          }
          if (q < nlen && dep[w] < kMfMaxPieceBytes) {
            cb[w] = cq;
            if ((uu.w >> ChildBit(cq)) & 1u) { go[w] = true; nxt[w] = (uu.x >> kDatBaseShiftDev) ^ cq; }
          }
          alive[w] = go[w];
        }
      }
      U4 v[kMfWalks];
#pragma unroll
      for (int w = 0; w < kMfWalks; ++w) if (go[w]) v[w] = ptrie[nxt[w]];
      any_alive = false;
#pragma unroll
      for (int w = 0; w < kMfWalks; ++w) {
        if (go[w]) {
          if ((v[w].x & 0x1FFu) == (0x100u | cb[w])) { u[w] = v[w]; ++dep[w]; if ((cb[w] & 0xC0u) != 0x80u) ++nch[w]; }     // :969-971
          else alive[w] = false;
        }
        any_alive = any_alive || alive[w];
      }
    }
    // ---- to the stream, walk group by walk group (text order): a start's pieces, then UNK unless a piece of exactly
    // one character matched (:995-1005) ----
#pragma unroll
    for (int w = 0; w < kMfWalks; ++w) {
      if (take[w] == 0) continue;                // (wave-uniform)
      const bool over_row = wv::any(has[w] && k[w] > J);     // (cannot happen: J is the trie's deepest prefix chain)
      const uint32_t cnt = has[w] ? k[w] + (single[w] ? 0u : 1u) : 0u;
      const uint32_t incl = wv::scan_add(cnt);
      const uint32_t total = wv::read_lane(incl, 63);
      if (over_row || n_out + static_cast<int>(total) > ccap) return -1;
      U2 *o = out + n_out + (incl - cnt);
      const uint32_t head = kCsFirst | (static_cast<uint32_t>(mb[w] - 1) << kCsMbShift);
      wv::sync();
      if (has[w]) {
        for (uint32_t i = 0; i < k[w]; ++i) {
          U2 e = row[w][i];
          if (i == 0) e.x |= head;
          wv::store_stream(&o[i], e);
        }
        if (!single[w]) wv::store_stream(&o[k[w]], U2{unk_word | (1u << kCsNchShift) | (static_cast<uint32_t>(mb[w]) << kCsLenShift) | (k[w] == 0 ? head : 0u), unk_bits});
      }
      n_out += static_cast<int>(total);
      wv::sync();
    }
  }
  *nsp = n_sp;
  return n_out;
}

// EncodeOptimized's fold (:960-1008) for this lane's sentence from its candidate stream cs[0, n_ent).  best_path_ends_at
// lives in two LDS rings of RC entries INDEXED BY CHARACTER (a candidate carries its length in characters; positions inside
// a character are never relaxed -- match_wave leaves such pieces out): RC = 8 or 16 instead of one entry per byte of the
// longest piece, which is what lets a launch of split tiles keep 12 wavefronts on a CU.  The staging block and the
// back-pointer blocks are unigram_stream_lane's (by BYTE position: the backtrack walks bytes).  The stream is read in
// blocks of 8 entries (64 bytes), two blocks ahead of their use; all lanes of the wave step through their streams together.
// Returns the number of blocks the wave went through.
template <int RC, typename BP>
SPMX_DEVICE int fold_stream_lane(const SpmxDev &d, const U2 *cs, int n_ent, const BpCol<typename BP::T> &gb, int nlen, float *ring_s,
                                 typename BP::T *ring_b, typename BP::T *st, bool active_in) {
  typedef typename BP::T BT;
  constexpr bool kShort = sizeof(BT) == 2;
  constexpr uint32_t rm = static_cast<uint32_t>(RC - 1);
  const float unk_score = d.unk_score;
  const uint32_t unk_word = static_cast<uint32_t>(d.unk_id) & kCsIdMask;
  const bool active = active_in && nlen > 0 && n_ent > 0;
  if (!active) n_ent = 0;
  int s = 0, mb = 0;                               // the start's byte position, its character's bytes
  uint32_t sc = 0;                                 // ... its position in characters
  float sbest = 0.f;
  if (active) {
    for (uint32_t k = 0; k <= rm; ++k) ring_b[k * 64] = 0;
    ring_s[0] = 0.f;                               // best_path_ends_at[0].best_path_score = 0
  }
  // (the staging block is written entry by entry -- 16-bit entries in the short form -- and moved as 16-byte units: by
  // memcpy, which may alias anything -- a typed 16-byte read may legally be scheduled before the last entry's store, and
  // g++ -O2 did exactly that in the emulator build; both compilers turn these into single 16-byte moves)
  auto put_block = [&](int blk_index) __attribute__((always_inline)) {
    BT *blk = gb.blk(blk_index);
    if (kShort) {
      __builtin_memcpy(__builtin_assume_aligned(blk, 16), __builtin_assume_aligned(st, 16), 16);
    } else {
      __builtin_memcpy(__builtin_assume_aligned(blk, 16), __builtin_assume_aligned(st, 16), 16);
      __builtin_memcpy(__builtin_assume_aligned(blk + 4, 16), __builtin_assume_aligned(st + 256, 16), 16);
    }
  };
  // the start moves on by its character (:1007): the position behind it is final -- its entry goes to the staging block,
  // the block of 8 byte positions left behind to HBM, its ring slot is free for the character RC further on
  auto advance = [&]() __attribute__((always_inline)) {
    const int s2 = s + mb;
    const uint32_t slB = ((sc + 1u) & rm) << 6;
    const BT finB = ring_b[slB];
    sbest = ring_s[slB];
    if ((s2 >> 3) != (s >> 3)) put_block(s >> 3);
    if (kShort) st[static_cast<uint32_t>(s2) & 7u] = finB;
    else st[((static_cast<uint32_t>(s2) >> 2) & 1u) * 256u + (static_cast<uint32_t>(s2) & 3u)] = finB;
    ring_b[slB] = 0;
    s = s2;
    ++sc;
  };
  auto step = [&](uint32_t x, uint32_t y, int k) __attribute__((always_inline)) {
    if (k >= n_ent) return;
    if (x & kCsFirst) {
      if (k > 0) advance();
      mb = static_cast<int>((x >> kCsMbShift) & 3u) + 1;
    }
    const int len = static_cast<int>((x >> kCsLenShift) & 0x3Fu);
    const uint32_t id = x & kCsIdMask;
    const uint32_t sl = ((sc + ((x >> kCsNchShift) & 15u)) & rm) << 6;
    const BT b = ring_b[sl];
    const float r = ring_s[sl];
    if (id == unk_word) {                                                            // :995-1005, float arithmetic
      const float cand = unk_score + sbest;
      if (b == 0 || cand > r) { ring_s[sl] = cand; ring_b[sl] = BP::unk(len); }
    } else {
      const double cand = static_cast<double>(wv::bits_to_float(y)) + static_cast<double>(sbest);   // :982-983
      if (b == 0 || cand > static_cast<double>(r)) {                                  // :984-989
        ring_s[sl] = static_cast<float>(cand);
        ring_b[sl] = BP::piece(id, len);
      }
    }
  };
  const Q4 *blocks = reinterpret_cast<const Q4 *>(cs);     // a block of 8 entries = 4 x Q4
  Q4 A0{0, 0, 0, 0}, A1 = A0, A2 = A0, A3 = A0, B0 = A0, B1 = A0, B2 = A0, B3 = A0;
  const int n_blocks = (n_ent + 7) >> 3;
  if (n_blocks > 0) { A0 = wv::load_stream(blocks); A1 = wv::load_stream(blocks + 1); A2 = wv::load_stream(blocks + 2); A3 = wv::load_stream(blocks + 3); }
  if (n_blocks > 1) { B0 = wv::load_stream(blocks + 4); B1 = wv::load_stream(blocks + 5); B2 = wv::load_stream(blocks + 6); B3 = wv::load_stream(blocks + 7); }
  int trips = 0;
  for (int kb = 0; wv::any(kb < n_blocks); kb += 2) {
    ++trips;
    {
      const int k0 = kb * 8;
      step(A0.x, A0.y, k0); step(A0.z, A0.w, k0 + 1); step(A1.x, A1.y, k0 + 2); step(A1.z, A1.w, k0 + 3);
      step(A2.x, A2.y, k0 + 4); step(A2.z, A2.w, k0 + 5); step(A3.x, A3.y, k0 + 6); step(A3.z, A3.w, k0 + 7);
      if (kb + 2 < n_blocks) { const Q4 *q = blocks + 4 * (kb + 2); A0 = wv::load_stream(q); A1 = wv::load_stream(q + 1); A2 = wv::load_stream(q + 2); A3 = wv::load_stream(q + 3); }
    }
    {
      const int k0 = kb * 8 + 8;
      step(B0.x, B0.y, k0); step(B0.z, B0.w, k0 + 1); step(B1.x, B1.y, k0 + 2); step(B1.z, B1.w, k0 + 3);
      step(B2.x, B2.y, k0 + 4); step(B2.z, B2.w, k0 + 5); step(B3.x, B3.y, k0 + 6); step(B3.z, B3.w, k0 + 7);
      if (kb + 3 < n_blocks) { const Q4 *q = blocks + 4 * (kb + 3); B0 = wv::load_stream(q); B1 = wv::load_stream(q + 1); B2 = wv::load_stream(q + 2); B3 = wv::load_stream(q + 3); }
    }
  }
  if (active) {
    advance();                                     // the last start's character ends the text: position nlen is final
    put_block(nlen >> 3);                          // the last block (it holds position nlen)
  }
  return trips;
}

}  // namespace spmx
#endif
