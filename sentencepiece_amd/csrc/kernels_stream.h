// Unigram (and word-wise BPE) segmentation, streaming form: a wavefront takes a
// tile of up to 64 sentences and runs ONE SENTENCE PER LANE, so that all 64
// lanes carry an independent EncodeOptimized recurrence
// (src/unigram_model.cc:889-1020), with a per-lane working set in LDS that does
// not depend on the sentence length.
// (History, profiles/: a sentence-per-wave form spent 91 % of its cycles in a
// serial loop over end positions; a form with text + back-pointers in LDS was
// held to one wave per SIMD by its 2 B per byte per sentence.)
//
// The two nested loops of the reference (for each start: for each prefix) are
// flattened into one loop in which every lane does exactly one trie probe per
// iteration; a lane whose walk dies relaxes UNK and moves to its next start in
// the same iteration, so lanes stay busy regardless of how deep their
// neighbours' walks go.
//
// Per lane, in LDS (lane-interleaved or lane-strided, see StreamLds):
//   * a ring of the R most recent positions' best_path_ends_at entries
//     (src/unigram_model.cc:944-952): score (float) and a packed back-pointer
//     word  id | piece length << 24 | unknown << 31  (0 = position not reached);
//   * a W-byte window of the normalized text around the current start.
// Per lane, in HBM scratch (one slab per wavefront, position-major so that the
// lanes of a wave, which advance at similar speeds, touch the same lines):
//   * the normalized text, as dwords   text[pos >> 2][lane];
//   * the FINAL back-pointer word of every character start   bp[lane][pos]: a
//     position is final when the start reaches it (all candidates into a
//     position are folded before any piece starting there is scored,
//     :960-1008); final words are staged in LDS and leave for HBM as whole
//     32-byte blocks of 8 positions (a 4-byte store per position dirtied a
//     32-byte sector each: measured 9x write amplification, profiles/).
// The backtrack (:1010-1018) follows bp[] from the end and writes ids straight
// into the arena, last piece first.
//
// ONE persistent launch serves every length class of a call (encode_stream_block): the waves take tiles from a
// queue, longest class first.  A lane normalizes its own sentence from HBM into its text column -- fast_norm_stream
// for ASCII, norm_lane_any (kernels_normlane.h) for everything else.  A stray non-ASCII sentence in an ASCII tile
// would hold the other 63 lanes up for its whole length, so the wave keeps it in a backlog of its own and runs the
// backlog as a tile (all lanes in norm_lane_any) when it has filled up or the queue is empty; waves share nothing but
// the queue's cursor.  A sentence whose normalized form does not fit its class's text column goes to the call's
// overflow list (a second, small launch with exact capacities); nothing fails for its length.
// A workgroup is W wavefronts that share nothing but two read-only LDS tables (first trie level, byte classes;
// every wave writes identical copies, so no workgroup barrier is ever needed); each wave owns a private slice of LDS.
#ifndef SPMX_KERNELS_STREAM_H_
#define SPMX_KERNELS_STREAM_H_

namespace spmx {

constexpr uint32_t kStreamSharedBytes = 256u * 16u + 256u;   // roottab + bcls

// per-wave profiling counters
struct WaveCounters {
  unsigned long long n_sent = 0, n_raw = 0, n_ids = 0, n_trips = 0, n_backlog = 0;
  unsigned long long cyc[4] = {0, 0, 0, 0};
};

// :979-983 score of the piece in unit u (byte length len) as the reference's double
SPMX_DEVICE double piece_score(const U4 &u, int len, float max_score) {
  double score = static_cast<double>(wv::bits_to_float(u.z));
  if (u.y & kPtUserDefined) {                                      // (length * max_score_ - 0.1)
    const float prod = static_cast<float>(len) * max_score;
    score = static_cast<double>(prod) - 0.1;
  }
  return score;
}

constexpr uint32_t kBwUnk = 0x80000000u;   // back-pointer word: the UNK candidate won this position
constexpr int kBwLenShift = 24;
constexpr uint32_t kBwLenMask = 0x7Fu;
constexpr uint32_t kBwIdMask = 0x00FFFFFFu;

// Back-pointer entry of a position: which piece ends there.  Two forms, chosen per model at plan time:
//   BpWord  (uint32)  id | length << 24 | UNK flag; 0 = not reached
//   BpShort (uint16)  id + 1, or 0xFFFB + length for UNK (length 1..4); 0 = not reached.  The piece's length is not
//                     stored: the backtrack reads it from SpmxDev::plen.  For vocabularies below 65531 ids and rings of
//                     16: half the LDS of ring_b and of the staging block (16 instead of 12 wavefronts per CU) and half
//                     the back-pointer bytes through HBM.
struct BpWord {
  typedef uint32_t T;
  static SPMX_DEVICE T piece(uint32_t id_word, int len) { return (id_word & kBwIdMask) | (static_cast<uint32_t>(len) << kBwLenShift); }
  static SPMX_DEVICE T unk(int len) { return (static_cast<uint32_t>(len) << kBwLenShift) | kBwUnk; }
  static SPMX_DEVICE bool is_unk(T w) { return (w & kBwUnk) != 0; }
  static SPMX_DEVICE int len(const SpmxDev &, T w) { return static_cast<int>((w >> kBwLenShift) & kBwLenMask); }
  static SPMX_DEVICE int32_t id(T w) { return static_cast<int32_t>(w & kBwIdMask); }
};
struct BpShort {
  typedef uint16_t T;
  static SPMX_DEVICE T piece(uint32_t id_word, int) { return static_cast<T>((id_word & kBwIdMask) + 1u); }
  static SPMX_DEVICE T unk(int len) { return static_cast<T>(0xFFFBu + static_cast<uint32_t>(len)); }
  static SPMX_DEVICE bool is_unk(T w) { return w >= 0xFFFCu; }
  static SPMX_DEVICE int len(const SpmxDev &d, T w) {
    if (w == 0) return 0;
    return w >= 0xFFFCu ? static_cast<int>(w) - 0xFFFB : static_cast<int>(d.plen[static_cast<uint32_t>(w) - 1u]);
  }
  static SPMX_DEVICE int32_t id(T w) { return static_cast<int32_t>(w) - 1; }
};
constexpr uint32_t kBpShortMaxVocab = 0xFFFAu;
// A lane's view of its tile's back-pointer array: blocks of 8 positions, block k of lane l at [(k << sh) + l] --
// position-major like the text, because the lanes of a tile advance together: the 64 blocks k of a tile are one
// contiguous run (1 KB short form, 2 KB word form) that the lanes write within a few iterations of each other and the
// backtrack reads the same way.  (Lane-major rows kept one open cache line per LANE for 64 positions: the L2 of an XCD
// is about as large as the open lines of its resident lanes, so lines left for memory half written, several times.)
template <typename BT>
struct BpCol {
  BT *p;            // + lane * 8 already
  uint32_t sh;
  SPMX_DEVICE BT *blk(int k) const { return p + ((static_cast<uint64_t>(static_cast<uint32_t>(k)) << sh) << 3); }
  SPMX_DEVICE BT at(int pos) const { return blk(pos >> 3)[pos & 7]; }
};

struct StreamLds {
  U4 *roottab;        // [256] first trie level (shared by the workgroup, read-only)
  uint8_t *bcls;      // [256] byte classes of the ASCII fast path (shared, read-only)
  float *ring_s;      // [R][64]
  uint32_t *ring_b;   // [R][64] back-pointer entries (uint16 in the short form)
  uint8_t *win;       // [64][W + 4]: lane l's window starts at win + l * (W + 4)
  uint32_t *stage;    // final back-pointer entries of the lane's current block of 8 positions: [2][64][4] words, or
                      // [64][8] uint16 in the short form
  uint32_t *backlog;  // [64] sentences waiting for a tile of their own (encode_stream_block)
  uint8_t *rawwin;    // [64][kRawWinBytes] raw-text windows of norm_lane_any (aliases the rings / the BPE word: idle
                      // while a tile is normalized)
  // BPE (kernels_bpe_stream.h) instead of the rings / window / staging block:
  uint32_t *asym;     // [256] symbol of every one-byte character (shared, aliases roottab)
  BpeWordLds bw;      // the lane's current word
  uint8_t *bwin;      // [64][kBpeWindow + 4] text windows
};

// bytes of the text window: a power of two that holds the longest piece (< ring), the two bytes looked at beyond it
// and the dword in flight
SPMX_HD inline uint32_t StreamWindow(uint32_t ring) { uint32_t w = 32; while (w < ring + 6u) w <<= 1; return w; }
// model: 1 unigram, 2 BPE
// bpsz: bytes of a back-pointer entry (4; 2 in the short form, see BpShort)
SPMX_HD inline uint32_t StreamPrivateBytes(int model, uint32_t ring, uint32_t bpsz = 4u) {
  uint32_t work = model == 2 ? BpeWordLdsBytes() + 64u * (kBpeWindow + 4u)
                             : 64u * ring * (4u + bpsz) + 64u * (StreamWindow(ring) + 4u) + 64u * 8u * bpsz;
  if (work < 64u * kRawWinBytes) work = 64u * kRawWinBytes;
  return ((work + 15u) & ~15u) + 256u;                 // + the backlog
}
SPMX_HD inline uint32_t StreamLdsBytes(int model, uint32_t ring, uint32_t waves, uint32_t bpsz = 4u) {
  return kStreamSharedBytes + waves * StreamPrivateBytes(model, ring, bpsz);
}
// HBM scratch of one tile whose text columns hold tcap bytes, for 1 << lane_shift lanes:
//   uint32 text[StreamTextDwords(tcap, ring)][lanes]   then   bp[StreamBpStride(tcap) / 8][lanes][8] (entries of bpsz bytes)
SPMX_HD inline uint64_t StreamTextDwords(uint32_t tcap, uint32_t ring) {
  return static_cast<uint64_t>(tcap + 3) / 4 + StreamWindow(ring) / 4 + 4;
}
SPMX_HD inline uint32_t StreamBpStride(uint32_t tcap) { return (tcap + 16u) & ~7u; }   // words per lane, whole blocks of 8
SPMX_HD inline uint64_t StreamSlabBytes(uint32_t tcap, uint32_t ring, uint32_t lane_shift, uint32_t bpsz = 4u) {
  const uint64_t text = ((StreamTextDwords(tcap, ring) << lane_shift) * 4u + 31u) & ~static_cast<uint64_t>(31);
  return text + (static_cast<uint64_t>(StreamBpStride(tcap)) << lane_shift) * bpsz + 32u;
}

// split form: where the candidate streams of a tile begin in its slab, and the slab's size with them
SPMX_HD inline uint64_t StreamSplitBase(uint32_t tcap, uint32_t ring, uint32_t lane_shift, uint32_t bpsz) {
  return (StreamSlabBytes(tcap, ring, lane_shift, bpsz) + 255u) & ~static_cast<uint64_t>(255);
}

// priv: bytes of one wavefront's slice (EncodeArgs::private_bytes; 0: StreamPrivateBytes)
SPMX_DEVICE StreamLds carve_stream(unsigned char *base, int model, uint32_t ring, int wave, uint32_t bpsz = 4u, uint32_t priv = 0u) {
  StreamLds t;
  if (priv == 0u) priv = StreamPrivateBytes(model, ring, bpsz);
  t.roottab = reinterpret_cast<U4 *>(base);
  t.asym = reinterpret_cast<uint32_t *>(base);
  t.bcls = base + 256u * 16u;
  unsigned char *mine = base + kStreamSharedBytes + static_cast<uint32_t>(wave) * priv;
  t.backlog = reinterpret_cast<uint32_t *>(mine + priv - 256u);
  t.rawwin = mine;
  t.ring_s = reinterpret_cast<float *>(mine);
  t.ring_b = reinterpret_cast<uint32_t *>(mine + 64u * ring * 4u);
  t.win = mine + 64u * ring * (4u + bpsz);
  t.stage = reinterpret_cast<uint32_t *>(t.win + 64u * (StreamWindow(ring) + 4u));
  t.bw.sym = reinterpret_cast<uint32_t *>(mine);
  t.bw.score = reinterpret_cast<float *>(mine + kBpeWordMax * 64u * 4u);
  t.bw.merged = reinterpret_cast<uint32_t *>(mine + kBpeWordMax * 64u * 8u);
  t.bw.len = mine + kBpeWordMax * 64u * 12u;
  t.bwin = mine + BpeWordLdsBytes();
  return t;
}

// piece_score; UDS = false drops the user-defined branch (models without USER_DEFINED pieces)
template <bool UDS>
SPMX_DEVICE double piece_score_u(const U4 &u, int len, float max_score) {
  if (UDS) return piece_score(u, len, max_score);
  return static_cast<double>(wv::bits_to_float(u.z));
}

// EncodeOptimized for this lane's sentence.  One iteration costs ONE global trie probe per lane, split into a
// control half and a data half:
//   control  (registers only) -- consume the probe in flight (:969-971); the child-label summary in the unit
//            tells whether the next byte can match at all, so a walk's last, failing probe is usually never
//            issued; if the walk of this start is over, move to the next start (:1007) and take its first
//            trie level from the root-table unit that was fetched from LDS when the previous start began;
//            issue the next probe;
//   data     (LDS, under the shadow of that probe) -- up to three relaxations of best_path_ends_at, in the
//            reference's order: (A) the piece just matched (double add, double compare against the
//            float-rounded best, :979-989), (B) UNK for the start that is over unless a one-character piece
//            was seen (float add, :990-1005), (C) a one-byte piece of the next start.  Their LDS reads are
//            issued together; where two of them hit the same position the later one is forwarded the earlier
//            one's result in registers.
// State:
//   * text comes from the W-byte LDS window `win` (position p at win[p & wmask]), refilled one dword per
//     iteration from the lane's text column gt -- the load is issued next to the trie probe and lands in the
//     window at the top of the next iteration, by which time the probe wait has covered it;
//   * best_path_ends_at lives in the rings only: ring_s / ring_b slot of position e is [(e mod R) * 64];
//     ring_b == 0 means "not reached" (:984);
//   * when the start moves from s to s2, position s2's back-pointer word is final: it goes to the lane's staging
//     block st[] (position p at st[((p >> 2) & 1) * 256 + (p & 3)]), the block of 8 positions that s2 leaves behind
//     is written to gb[] as two 16-byte stores, and the ring slots of the positions (s, s2] are cleared for the
//     positions that will reuse them R later.  gb is this lane's view of the tile's back-pointer blocks (BpCol).
// Returns the number of iterations (wave-uniform).
// RING > 0: the ring size is a compile-time power of two (index masks and the distances between the LDS arrays fold
// into immediates); RING == 0: ANY size rm_in + 1 > the longest piece (slots by a running position-mod-R: the
// distances involved are below R, so one conditional subtraction wraps them); the window mask is wmask_in.  UDS: the
// model may have USER_DEFINED pieces.
template <int RING, bool UDS, typename BP>
SPMX_DEVICE int unigram_stream_lane(const SpmxDev &d, const TextCol &gt, const BpCol<typename BP::T> &gb, int nlen, float *ring_s,
                                    typename BP::T *ring_b, uint32_t rm_in, uint8_t *win, uint32_t wmask_in,
                                    typename BP::T *st, const U4 *roottab, bool active_in) {
  typedef typename BP::T BT;
  constexpr bool kShort = sizeof(BT) == 2;
  const uint32_t rm = RING ? static_cast<uint32_t>(RING - 1) : rm_in;
  const uint32_t wmask = RING ? StreamWindow(RING) - 1u : wmask_in;
  const uint32_t R = rm + 1u;
  uint32_t s_slot = 0;                            // RING == 0: s mod R
  // slot of the position base_slot's position + d (0 <= d < R)
  auto wrap = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t { return x >= R ? x - R : x; };
  const U4 *__restrict__ ptrie = d.ptrie;
  const float unk_score = d.unk_score, max_score = d.max_score;
  const int W = static_cast<int>(wmask) + 1;
  int trips = 0;
  bool active = active_in && nlen > 0;
  if (!active) nlen = 0;                          // all indices of an idle lane stay 0
  int s = 0, mb = 0, dep = 0;
  bool walking = false, single = true;
  uint32_t c = 0;
  float sbest = 0.f;
  U4 u{0, 0, 0, 0};
  uint32_t cs = 0, cs1 = 0;
  U4 rn{0, 0, 0, 0};
  int rd = 1;                                     // how many bytes of the next start rn stands for: 1 (root table), or the whole
                                                  // first character (cfirst)
  uint32_t cq = 0;
  // The unit a walk from position q starts with, and the byte the walk looks at next.  The root table (LDS) holds the
  // first BYTE's unit.  cfirst (dev.h; models with many pieces in multi-byte scripts) holds, by code point, the unit the
  // byte trie reaches after a whole two- or three-byte CHARACTER -- no piece ends inside a character (checked when the
  // table is built), so the two or three dependent probes that spell a CJK character are one probe, and that one is asked
  // for a whole start ahead: off the lane's latency chain (C5: 4.8 dependent probes per character, most of them these).
  const U4 *__restrict__ cfirst = d.cfirst;
  auto first_unit = [&](int q, int lim) __attribute__((always_inline)) {
    cs = win[static_cast<uint32_t>(q) & wmask];
    cs1 = win[static_cast<uint32_t>(q + 1) & wmask];
    rn = roottab[cs];
    rd = 1;
    if (cfirst != nullptr && cs >= 0xC2u && cs < 0xF0u) {
      const uint32_t b2 = win[static_cast<uint32_t>(q + 2) & wmask];
      const bool three = cs >= 0xE0u;
      const uint32_t cp = three ? ((cs & 0x0Fu) << 12) | ((cs1 & 0x3Fu) << 6) | (b2 & 0x3Fu) : ((cs & 0x1Fu) << 6) | (cs1 & 0x3Fu);
      const bool wf = (cs1 & 0xC0u) == 0x80u && (!three || ((b2 & 0xC0u) == 0x80u && cp >= 0x800u));   // the canonical bytes of cp
      const int dch = three ? 3 : 2;
      if (wf && q + dch <= lim) {
        rn = cfirst[cp];
        rd = dch;
        cs1 = three ? win[static_cast<uint32_t>(q + 3) & wmask] : b2;       // the byte behind the character
      }
    }
  };
  int nf = 0;                                     // next text dword to fetch; the window holds dwords [nf - W/4, nf)
  bool pf_pend = false;
  uint32_t pf = 0;
  if (active) {
    for (uint32_t k = 0; k <= rm; ++k) ring_b[k * 64] = 0;
    ring_s[0] = 0.f;                              // best_path_ends_at[0].best_path_score = 0
    for (int k = 0; k < W / 4; ++k) *reinterpret_cast<uint32_t *>(win + 4 * k) = gt.dw(k);
    nf = W / 4;
    first_unit(0, nlen);
  }
  while (wv::any(active)) {
    ++trips;
    if (pf_pend) *reinterpret_cast<uint32_t *>(win + ((4u * static_cast<uint32_t>(nf - 1)) & wmask)) = pf;
    // ---------------- control ----------------
    const int dep1 = dep + 1;
    const bool matchA = active && walking && (u.x & 0x1FFu) == (0x100u | c);           // :969-971
    const bool termA = matchA && (u.x & kDatTerminalDev) && !(u.y & kPtUnused);          // :973-974
    const bool cont = matchA && s + dep1 < nlen && ((u.w >> ChildBit(cq)) & 1u);
    const bool ended = active && !cont;
    const int s2 = s + mb;                        // :1007 the next start
    const bool begin = ended && s2 < nlen;
    const U4 r = rn;
    int mb2 = static_cast<int>(r.x & 7u);         // :962-963 the character's byte length rides in the root-table entry
    if (mb2 > nlen - s2) mb2 = nlen - s2;
    const bool rootC = begin && (r.x & 0x100u);   // a piece starts with cs (entries without one have the bit clear)
    const int dC = rd;                            // bytes of the start at s2 that r stands for
    const bool termC = rootC && (r.x & kDatTerminalDev) && !(r.y & kPtUnused);
    const bool contC = rootC && s2 + dC < nlen && ((r.w >> ChildBit(cs1)) & 1u);
    const bool nwalking = cont || contC;
    const uint32_t nnode = cont ? (u.x >> kDatBaseShiftDev) : (r.x >> kDatBaseShiftDev);
    const uint32_t nc = cont ? cq : cs1;
    const U4 uA = u;
    // window refill: dword nf may replace positions [4 nf - W, 4 nf - W + 4), which are dead once they lie below s
    pf_pend = active && 4 * nf + 4 <= s + W && 4 * nf < nlen + 8;
    if (pf_pend) { pf = gt.dw(nf); ++nf; }
    if (nwalking) u = ptrie[nnode ^ nc];          // next probe
    // ---------------- data ----------------
    const int eA = s + dep1;
    const int eB = s2;
    const int eC = s2 + dC <= nlen ? s2 + dC : nlen;
    const uint32_t slB = RING ? (static_cast<uint32_t>(eB) & rm) : wrap(s_slot + static_cast<uint32_t>(mb));
    const uint32_t oA = (RING ? (static_cast<uint32_t>(matchA ? eA : 0) & rm) : (matchA ? wrap(s_slot + static_cast<uint32_t>(dep1)) : 0u)) << 6;
    const uint32_t oB = slB << 6;
    const uint32_t oC = (RING ? (static_cast<uint32_t>(eC) & rm) : wrap(slB + static_cast<uint32_t>(eC - eB))) << 6;
    BT bA = ring_b[oA], bB = ring_b[oB], bC = ring_b[oC];
    float rA = ring_s[oA], rB = ring_s[oB], rC = ring_s[oC];
    // (A) the piece that just matched
    const double candA = piece_score_u<UDS>(uA, dep1, max_score) + static_cast<double>(sbest);  // :982-983
    const bool updA = termA && (bA == 0 || candA > static_cast<double>(rA));             // :984-989
    const float nvA = static_cast<float>(candA);
    const BT wA = BP::piece(uA.y, dep1);
    if (updA && eB == eA) { rB = nvA; bB = wA; }
    if (updA && eC == eA) { rC = nvA; bC = wA; }
    const bool single2 = single || (termA && dep1 == mb);                                // :990
    // (B) UNK for the start that is over
    const float candB = unk_score + sbest;                                               // :997-1001, float
    const bool updB = ended && !single2 && (bB == 0 || candB > rB);
    const float sbest2 = updB ? candB : rB;
    const BT finB = updB ? BP::unk(mb) : bB;
    // (C) the piece of the next start that r stands for: its first byte, or its first character (cfirst)
    const double candC = piece_score_u<UDS>(r, dC, max_score) + static_cast<double>(sbest2);
    const bool updC = termC && (bC == 0 || candC > static_cast<double>(rC));
    if (updA) { ring_s[oA] = nvA; ring_b[oA] = wA; }
    if (updC) { ring_s[oC] = static_cast<float>(candC); ring_b[oC] = BP::piece(r.y, dC); }
    if (ended) {
      if ((s2 >> 3) != (s >> 3)) {                // the block of 8 positions behind s2 is complete
        BT *blk = gb.blk(s >> 3);
        if (SPMX_EXP & 2) {
        } else if (kShort) {
          *reinterpret_cast<Q4 *>(blk) = *reinterpret_cast<const Q4 *>(st);
        } else {
          const Q4 lo = *reinterpret_cast<const Q4 *>(st), hi = *reinterpret_cast<const Q4 *>(st + 256);
          *reinterpret_cast<Q4 *>(blk) = lo;
          *reinterpret_cast<Q4 *>(blk + 4) = hi;
        }
      }
      // position s2 is final
      if (kShort) st[static_cast<uint32_t>(eB) & 7u] = finB;
      else st[((static_cast<uint32_t>(eB) >> 2) & 1u) * 256u + (static_cast<uint32_t>(eB) & 3u)] = finB;
      // the positions just passed, (s, s2], are dead: free their ring slots (after this iteration's reads and writes)
      if (mb > 0) ring_b[oB] = 0;
      if (mb > 1)                                  // a multi-byte character: its inner positions too (rare in ASCII text)
        for (int k = 1; k < mb; ++k)
          ring_b[(RING ? (static_cast<uint32_t>(s2 - k) & rm) : (slB >= static_cast<uint32_t>(k) ? slB - static_cast<uint32_t>(k) : slB + R - static_cast<uint32_t>(k))) << 6] = 0;
    }
    // ---------------- commit ----------------
    if (ended) {
      active = begin;
      s = s2;
      s_slot = slB;
      mb = begin ? mb2 : 0;
      sbest = sbest2;
      single = termC && mb2 == dC;
      dep = rootC ? dC : 0;
      if (begin) first_unit(s2 + mb2, nlen);
    } else {
      dep = dep1;
      single = single2;
    }
    walking = nwalking;
    c = nc;
    if (nwalking) cq = win[static_cast<uint32_t>(s + dep + 1) & wmask];
  }
  if (active_in && nlen > 0) {                    // the last block (it holds position nlen)
    BT *blk = gb.blk(nlen >> 3);
    if (kShort) {
      *reinterpret_cast<Q4 *>(blk) = *reinterpret_cast<const Q4 *>(st);
    } else {
      const Q4 lo = *reinterpret_cast<const Q4 *>(st), hi = *reinterpret_cast<const Q4 *>(st + 256);
      *reinterpret_cast<Q4 *>(blk) = lo;
      *reinterpret_cast<Q4 *>(blk + 4) = hi;
    }
  }
  return trips;
}

// Backtrack (:1010-1018) + id post-processing (sentencepiece_processor.cc:581-613) of this lane's sentence:
// follows the lane's back-pointer entries gb from position nlen to 0 and writes the ids as it goes, last piece first, into slot[0, cap):
// forward order fills the slot from its END (ids end up in slot[cap - n, cap)), `reverse` fills it from the start.
// Returns n, or -1 on a broken chain / overflow.
// `tslot` (spans form, else null): the slot's twin in EncodeArgs::arena_tb, receives every token's begin.
template <typename BP>
SPMX_DEVICE int emit_stream_lane(const SpmxDev &d, const TextCol &gt, const BpCol<typename BP::T> &gb, int nlen, int32_t *slot,
                                 int32_t *tslot, int cap, int32_t *stage, bool active) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  int e = nlen, n = 0;
  bool right_unk = false, ok = true;
  active = active && nlen > 0;
  // The ids leave in bursts of 16 (64 bytes, four aligned 16-byte stores): id k waits in the lane's LDS staging column
  // stage[(k & 15) * 64] (the score ring, idle now) until its group is complete.  Single 4-byte stores kept one open
  // cache line per lane for the whole backtrack -- an XCD's L2 is about the size of the open lines of its resident
  // lanes -- and lines left for memory part written, several times over (4.8 bytes written per id byte).
  // The caller aligns the slot so that the groups are 16-byte aligned: slot + cap (forward order) / slot (reverse).
  auto put = [&](int32_t id, int tb) __attribute__((always_inline)) {
    if (tslot) {                                   // the spans form keeps the simple path
      slot[reverse ? n : cap - 1 - n] = id;
      tslot[reverse ? n : cap - 1 - n] = tb;
      ++n;
      return;
    }
    stage[(n & 15) << 6] = id;
    ++n;
    if ((n & 15) == 0) {
      if (SPMX_EXP & 1) return;
      int32_t *p = reverse ? slot + (n - 16) : slot + (cap - n);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Q4 v;
        if (reverse) v = Q4{static_cast<uint32_t>(stage[(4 * q) << 6]), static_cast<uint32_t>(stage[(4 * q + 1) << 6]),
                            static_cast<uint32_t>(stage[(4 * q + 2) << 6]), static_cast<uint32_t>(stage[(4 * q + 3) << 6])};
        else v = Q4{static_cast<uint32_t>(stage[(15 - 4 * q) << 6]), static_cast<uint32_t>(stage[(14 - 4 * q) << 6]),
                    static_cast<uint32_t>(stage[(13 - 4 * q) << 6]), static_cast<uint32_t>(stage[(12 - 4 * q) << 6])};
        *reinterpret_cast<Q4 *>(p + 4 * q) = v;
      }
    }
  };
  while (wv::any(active)) {
    if (active) {
      const typename BP::T w = gb.at(e);
      const int len = BP::len(d, w);
      if (len == 0 || len > e) { ok = false; active = false; continue; }
      const int tb = e - len;
      if (BP::is_unk(w)) {
        if (bf) {                                   // one BYTE id per byte of the unknown piece (:581-603)
          const bool sp = col_byte(gt, tb) == spb;
          const int nb = sp ? 3 : len;
          if (n + nb > cap) { ok = false; active = false; continue; }
          for (int x = nb - 1; x >= 0; --x) {
            const uint32_t byte = sp ? (x == 0 ? 0xE2u : (x == 1 ? 0x96u : 0x81u)) : col_byte(gt, tb + x);
            put(d.byte_ids[byte], tb);
          }
        } else if (!right_unk) {                    // a run of unknown pieces yields one id (:609-613)
          if (n >= cap) { ok = false; active = false; continue; }
          put(d.unk_id, tb);
        } else if (tslot) {                         // the run grows to the left: so does the merged token
          tslot[reverse ? n - 1 : cap - n] = tb;
        }
        right_unk = true;
      } else {
        right_unk = false;
        if (n >= cap) { ok = false; active = false; continue; }
        put(BP::id(w), tb);
      }
      e = tb;
      if (e <= 0) active = false;
    }
  }
  if (!tslot && ok)                                 // the last, incomplete group
    for (int k = n & ~15; k < n; ++k) slot[reverse ? k : cap - 1 - k] = stage[(k & 15) << 6];
  return ok ? n : -1;
}

}  // namespace spmx
#include "kernels_matchfold.h"
namespace spmx {

// ---- the tile queue of one launch ------------------------------------------------------------------------------
// Lane 0 of a wave asks for its next main tile: one atomic; the classes are laid out longest first in the cursor's range.
SPMX_DEVICE bool next_tile(const EncodeArgs &a, uint32_t *cls, uint32_t *first, uint32_t *cnt) {
  const uint32_t t = wv::atomic_add(&a.q->main_cursor, 1u);
  if (t >= a.total_main) return false;
  int c = static_cast<int>(a.n_classes) - 1;
  while (c > 0 && !(t >= a.cls[c].tile_base && t - a.cls[c].tile_base < a.cls[c].main_tiles)) --c;
  const uint32_t k = t - a.cls[c].tile_base;
  *cls = static_cast<uint32_t>(c);
  *first = k * a.cls[c].tw;
  *cnt = a.cls[c].count - *first < a.cls[c].tw ? a.cls[c].count - *first : a.cls[c].tw;
  return true;
}

// appends the sentences of the lanes in `m` to list[*count ...] (one atomic per wave)
SPMX_DEVICE void append_lanes(uint64_t m, bool mine, uint32_t sid, uint32_t *list, uint32_t *count, int lane) {
  if (!m) return;
  const int leader = wv::ffs64(m) - 1;
  uint32_t base = 0;
  if (lane == leader) base = wv::atomic_add(count, static_cast<uint32_t>(wv::popc64(m)));
  base = wv::shfl(base, leader);
  if (mine) list[base + static_cast<uint32_t>(wv::popc64(m & ((1ull << lane) - 1ull)))] = sid;
}

// Persistent body of the streaming kernels.  MODEL: 1 unigram, 2 BPE (word-wise models only).  RING: see
// unigram_stream_lane (0 = a.ring).  UDS: the model may have USER_DEFINED pieces.
// SPLIT_ONLY: every tile of the launch takes the split form (kernels_matchfold.h; EncodeSplitKernel): the lane-per-sentence
// normalizers and search are not even compiled in, and the wavefront's LDS slice is the split form's own (12 per CU).
template <int MODEL, int RING, bool UDS, typename BP = BpWord, bool SPLIT_ONLY = false>
SPMX_DEVICE void encode_stream_block(const EncodeArgs &a, unsigned char *smem) {
  typedef typename BP::T BT;
  constexpr uint32_t kBpSz = sizeof(BT);
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const uint32_t ring = RING ? static_cast<uint32_t>(RING) : a.ring;
  const StreamLds T = carve_stream(smem, MODEL, ring, wv::wave_in_block(), kBpSz, a.private_bytes);
  const uint32_t rm = ring - 1;
  const uint32_t W = StreamWindow(ring);
  float *my_rs = T.ring_s + lane;
  BT *my_rb = reinterpret_cast<BT *>(T.ring_b) + lane;
  uint8_t *my_win = T.win + static_cast<uint32_t>(lane) * (W + 4u);
  uint8_t *my_raw = T.rawwin + static_cast<uint32_t>(lane) * kRawWinBytes;
  {   // shared read-only tables; every wave writes all of both (same values): no workgroup barrier
    const uint32_t root = MODEL == 1 ? d.ptrie[0].x >> kDatBaseShiftDev : 0u;
    for (uint32_t cb = static_cast<uint32_t>(lane); cb < 256u; cb += 64u) {
      if (MODEL == 1) {
        U4 r = d.ptrie[root ^ cb];
        if ((r.x & 0x1FFu) != (0x100u | cb)) r = U4{0, 0, 0, 0};
        // the label byte is redundant here (it is the index): it carries the byte length of a character that
        // starts with cb instead (src/util.h:151-153; the one-byte space symbol is one character)
        r.x = (r.x & ~0xFFu) | static_cast<uint32_t>(cb == SpByteOf(d) ? 1 : OneCharLenDev(cb));
        T.roottab[cb] = r;
      } else {
        T.asym[cb] = char_lookup(d, cb, 1u);
      }
      const bool safe = cb < 128u && ((d.ascii_safe[cb >> 5] >> (cb & 31u)) & 1u);
      T.bcls[cb] = static_cast<uint8_t>(safe ? 0u : kBcComplex);
    }
    wv::sync();
  }
  const uint32_t wave_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block());
  uint8_t *slab = a.slab + static_cast<uint64_t>(wave_id) * a.slab_bytes;
  BT *my_st = kBpSz == 2 ? reinterpret_cast<BT *>(T.stage) + static_cast<uint32_t>(lane) * 8u
                         : reinterpret_cast<BT *>(T.stage + static_cast<uint32_t>(lane) * 4u);
  const int n_extra = d.n_prefix + d.n_suffix;
  const bool bf_sp = (d.flags & kNfByteFallback) && (d.flags & kNfCompressSp);
  WaveCounters tc;
  // The wave's BACKLOG: sentences of its ASCII tiles that need norm_lane_any.  A stray one would hold the other 63 lanes
  // of its tile up for its whole length, so it waits here (sentence index in LDS) until the wave has a tile's worth of
  // them -- or has run out of main tiles -- and they run together, every lane in norm_lane_any.  Nothing is shared
  // between waves.  (Only classes whose tiles have all 64 lanes put sentences here: a backlog tile uses the slab the
  // same way.)
  uint32_t bl_n = 0, bl_cls = 0;           // entries; the longest class among them (its text columns fit them all)
  bool drained = false;                     // the queue has no main tile left
  for (;;) {
    uint32_t c = 0, first = 0, ucnt = 0;
    bool backlog = bl_n >= 48u || (drained && bl_n > 0u);
    if (!backlog && !drained) {
      uint32_t got = 0;
      if (lane == 0) got = next_tile(a, &c, &first, &ucnt) ? 1u : 0u;
      got = wv::shfl(got, 0);
      if (!got) { drained = true; backlog = bl_n > 0u; }
      else { c = wv::shfl(c, 0); first = wv::shfl(first, 0); ucnt = wv::shfl(ucnt, 0); }
    }
    if (drained && !backlog) break;
    if (backlog) { c = bl_cls; ucnt = bl_n; }
    const StreamClass sc = a.cls[c];
    const int cnt = static_cast<int>(ucnt);
    const uint32_t tcap = sc.tcap;
    const uint32_t lane_shift = backlog ? 6u : sc.lane_shift;
    // this tile's view of the wave's slab
    const uint64_t text_bytes = ((StreamTextDwords(tcap, ring) << lane_shift) * 4u + 31u) & ~static_cast<uint64_t>(31);
    const TextCol gt{reinterpret_cast<uint32_t *>(slab) + lane, lane_shift};
    const BpCol<BT> gb{reinterpret_cast<BT *>(slab + text_bytes) + static_cast<uint32_t>(lane) * 8u, lane_shift};
    // split form (kernels_matchfold.h): the lanes' candidate streams lie behind the tile's text and back-pointer blocks
    const bool split = SPLIT_ONLY || (MODEL == 1 && !UDS && !backlog && sc.split != 0u);
    const uint64_t cs_stride = MatchStreamBytes(sc.ccap);
    const U2 *my_cs = reinterpret_cast<const U2 *>(slab + StreamSplitBase(tcap, ring, lane_shift, kBpSz) + static_cast<uint64_t>(lane) * cs_stride);
    int my_nent = 0;
    const uint32_t *list = a.lists + static_cast<uint64_t>(c) * a.n;
    uint32_t my_sid = 0;
    uint64_t my_beg = 0;
    uint32_t my_len = 0;
    bool over = false;               // goes to the overflow list
    if (lane < cnt) {
      my_sid = backlog ? T.backlog[lane] : list[first + lane];
      my_beg = a.offs[my_sid];
      const uint64_t l64 = a.offs[my_sid + 1] - my_beg;
      my_len = static_cast<uint32_t>(l64);
      over = l64 > sc.rcap;                                              // (the last class takes every length)
    }
    if (backlog) { wv::sync(); bl_n = 0; bl_cls = 0; }                   // (the entries are in registers now)
    const unsigned long long c0 = wv::clock();
    bool mine = false;
    int my_nlen = 0, my_nsp = 0;     // normalized length; how many of its bytes are the space symbol
    {
      const bool go = lane < cnt && !over;
      int nlen = 0;
      bool need_any = go && my_len > 0;
      if (split) {
        // ---- match: the tile's sentences one after another, each by the whole wavefront ----
        need_any = false;
        nlen = -2;                               // (this lane's result is set when its sentence's turn comes)
        const MatchLds ML = carve_match(T.rawwin, sc.rcap, tcap, a.match_rows);
        wv::sync();                              // (the image aliases the rings the previous tile's fold used)
        for (int j = 0; j < cnt; ++j) {
          const uint32_t j_len = wv::shfl(my_len, j);
          if (wv::shfl(over ? 1u : 0u, j) != 0u) continue;
          int nl = 0, nsp = 0, nent = 0;
          if (j_len > 0) {
            const uint64_t j_beg = (static_cast<uint64_t>(wv::shfl(static_cast<uint32_t>(my_beg >> 32), j)) << 32) | wv::shfl(static_cast<uint32_t>(my_beg), j);
            const uint8_t *src = a.text + j_beg;
            const unsigned long long m0 = wv::clock();
            // the sentence's image: the aligned 16-byte units that hold its bytes (as the word-per-lane kernel reads them:
            // include/spmx.h on the bytes that share a unit with the text's first and last byte), four loads in flight a lane
            const uintptr_t s_addr = reinterpret_cast<uintptr_t>(src);
            const uint32_t s_sh = static_cast<uint32_t>(s_addr & 15u);
            const Q4 *g = reinterpret_cast<const Q4 *>(s_addr - s_sh);
            Q4 *img = reinterpret_cast<Q4 *>(ML.raw);
            const uint32_t units = (s_sh + j_len + 15u) >> 4;
            for (uint32_t u0 = static_cast<uint32_t>(lane); u0 < units; u0 += 256u) {
              Q4 v0{0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
              v0 = g[u0];
              if (u0 + 64u < units) v1 = g[u0 + 64u];
              if (u0 + 128u < units) v2 = g[u0 + 128u];
              if (u0 + 192u < units) v3 = g[u0 + 192u];
              img[u0] = v0;
              if (u0 + 64u < units) img[u0 + 64u] = v1;
              if (u0 + 128u < units) img[u0 + 128u] = v2;
              if (u0 + 192u < units) img[u0 + 192u] = v3;
            }
            wv::sync();
            nl = normalize_wave(d, ML.raw + s_sh, static_cast<int>(j_len), ML.norm, static_cast<int>(tcap), lane);
            tc.cyc[0] += wv::clock() - m0;       // (of the match phase: the sentence's image and its normalization)
            if (nl > 0) {
              const TextCol gj{reinterpret_cast<uint32_t *>(slab) + j, lane_shift};      // (the backtrack's byte fallback reads it)
              for (int k = lane; 4 * k < nl; k += 64) gj.dw(k) = *reinterpret_cast<const uint32_t *>(ML.norm + 4 * k);
              U2 *out = reinterpret_cast<U2 *>(slab + StreamSplitBase(tcap, ring, lane_shift, kBpSz) + static_cast<uint64_t>(j) * cs_stride);
              nent = match_wave(d, ML.norm, nl, T.roottab, ML, a.match_rows, out, static_cast<int>(sc.ccap), lane, &nsp);
            }
            wv::sync();
          }
          if (lane == j) {
            if (nl < 0 || nent < 0) { over = true; }
            else { nlen = nl; my_nsp = nsp; my_nent = nent; }
          }
        }
        wv::sync_global();                       // (a lane folds what the whole wavefront wrote)
      } else if (!backlog && a.fast_ok && !sc.general) {
        // which form: a tile where one sentence in eight begins with non-ASCII text (CJK ...) steps character by
        // character; the byte-stepping form is for ASCII with the odd accent
        bool na = false;
        if (go && my_len > 0) {
          uint32_t v0 = 0, v1 = 0;
          if (my_len >= 4) __builtin_memcpy(&v0, a.text + my_beg, 4); else v0 = a.text[my_beg];
          if (my_len >= 8) __builtin_memcpy(&v1, a.text + my_beg + 4, 4);
          na = ((v0 | v1) & 0x80808080u) != 0u;
        }
        const bool by_char = a.no_char_norm == 2u || (a.no_char_norm == 0u && 8 * wv::popc64(wv::ballot(na)) >= cnt);
        if (go && my_len > 0) {
          if (by_char) nlen = char_norm_stream(d, a.text, my_beg, static_cast<int>(my_len), gt, T.bcls, static_cast<int>(tcap), &my_nsp);
          else nlen = fast_norm_stream(d, a.text, my_beg, static_cast<int>(my_len), gt, T.bcls, static_cast<int>(tcap), &my_nsp);
        }
        need_any = nlen < 0;
        // Not plain ASCII.  A tile that is mostly such sentences (CJK text ...) normalizes them here, one per lane; a
        // few stray ones wait in the backlog
        const uint64_t hm = wv::ballot(need_any);
        const uint32_t k = static_cast<uint32_t>(wv::popc64(hm));
        if (hm && (k < sc.min_lanes || a.no_lane_general) && lane_shift == 6u && bl_n + k <= 64u) {
          if (need_any) T.backlog[bl_n + static_cast<uint32_t>(wv::popc64(hm & ((1ull << lane) - 1ull)))] = my_sid;
          bl_n += k;
          if (c > bl_cls) bl_cls = c;
          tc.n_backlog += k;
          need_any = false;
          if (nlen < 0) nlen = -2;
          wv::sync();
        }
      }
      if (wv::any(need_any)) {
        wv::sync();                              // the raw windows alias LDS the previous tile's search used
        if (need_any) {
          ColSink sink{gt, static_cast<int>(tcap)};
          nlen = norm_lane_any(d, a.text, my_beg, static_cast<int>(my_len), sink, my_raw, &my_nsp);
          if (nlen < 0) over = true;
        }
        wv::sync();
      }
      if (go && nlen >= 0) { mine = true; my_nlen = nlen; }
    }
    {   // sentences that fit no column of this launch: the overflow launch takes them with exact capacities
      const uint64_t om = wv::ballot(over);
      if (om && a.over_list) {
        append_lanes(om, over, my_sid, a.over_list, &a.side->over_count, lane);
        if (over) {
          wv::atomic_max(&a.side->over_max_raw, static_cast<unsigned long long>(a.offs[my_sid + 1] - my_beg));
          a.counts[my_sid] = 0u;                 // (the scan that runs before the overflow launch sees no ids yet)
        }
      } else if (over) {                         // the overflow launch itself: longer than any launch can take
        a.counts[my_sid] = 0u;
        a.tmp_off[my_sid] = 0;
        a.sent_status[my_sid] = static_cast<uint8_t>(kSsOutOfRange);
        wv::atomic_add(&a.side->n_failed, 1ull);
      }
    }
    const unsigned long long c1 = wv::clock();
    tc.cyc[1] += c1 - c0;
    // ---- a slot of cap ids per sentence in the arena ----
    int cap = 0;
    // at most one id per normalized byte -- three for a one-byte space symbol that falls back to its bytes
    if (mine) cap = bf_sp ? my_nlen + 2 * my_nsp : my_nlen;
    // rooms are whole groups of 4 ids with 3 ids of slack, so that the lane can shift its slot to the alignment
    // emit_stream_lane's 16-byte stores want (slot + cap for the forward order, slot when reversing)
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
    const int at = excl + d.n_prefix + ((d.flags & kNfReverse) ? 0 : cap);
    const int shift = (4 - (at & 3)) & 3;
    int32_t *slot = a.arena + base + static_cast<unsigned long long>(excl + shift) + d.n_prefix;
    int32_t *tslot = a.arena_tb ? a.arena_tb + base + static_cast<unsigned long long>(excl + shift) + d.n_prefix : nullptr;
    bool broken = false;
    int n = 0;
    bool at_end = false;      // the ids sit at the end of the slot
    unsigned long long c2 = c1;
    if (MODEL == 1) {
      // ---- segment, then backtrack: the slot is filled from its end (or from its start when reversing) ----
      if (split) {
        // the fold's rings are indexed by character (a.split_ring = 8 or 16 entries): its own layout of the wave's slice
        wv::sync();
        const uint32_t rc = a.split_ring;
        float *f_rs = reinterpret_cast<float *>(T.rawwin) + lane;
        BT *f_rb = reinterpret_cast<BT *>(T.rawwin + 64u * rc * 4u) + lane;
        BT *f_st = reinterpret_cast<BT *>(T.rawwin + 64u * rc * (4u + kBpSz)) + (kBpSz == 2 ? static_cast<uint32_t>(lane) * 8u : static_cast<uint32_t>(lane) * 4u);
        if (rc == 8u) tc.n_trips += static_cast<unsigned long long>(fold_stream_lane<8, BP>(d, my_cs, my_nent, gb, my_nlen, f_rs, f_rb, f_st, mine));
        else tc.n_trips += static_cast<unsigned long long>(fold_stream_lane<16, BP>(d, my_cs, my_nent, gb, my_nlen, f_rs, f_rb, f_st, mine));
        wv::sync();
      } else {
        tc.n_trips += static_cast<unsigned long long>(
            unigram_stream_lane<RING, UDS, BP>(d, gt, gb, my_nlen, my_rs, my_rb, rm, my_win, W - 1u, my_st, T.roottab, mine));
      }
      c2 = wv::clock();
      if (!overflow) {
        n = emit_stream_lane<BP>(d, gt, gb, my_nlen, slot, tslot, cap, reinterpret_cast<int32_t *>(my_rs), mine);
        at_end = (d.flags & kNfReverse) == 0;
      }
    } else {
      // ---- word by word, ids written as the words complete: the slot is filled from its start (end when reversing) ----
      n = bpe_stream_lane(d, gt, my_nlen, slot, tslot, cap, T.bw, T.asym, T.bwin + static_cast<uint32_t>(lane) * (kBpeWindow + 4u),
                          kBpeWindow - 1u, lane, mine && !overflow);
      c2 = wv::clock();
      at_end = (d.flags & kNfReverse) != 0;
      const bool handed = n == -2;              // a word too long for the lane form: the sentence goes to the long form
      append_lanes(wv::ballot(handed), handed, my_sid, a.long_list, &a.side->long_count, lane);
      if (handed) {
        wv::atomic_add(&a.side->long_raw, static_cast<unsigned long long>(my_len));
        a.counts[my_sid] = 0u;                   // (until the long form has had it)
        mine = false; n = 0;
      }
    }
    broken = n < 0;
    if (broken) n = 0;
    if (mine && !broken && !overflow) {
      int32_t *ids = at_end ? slot + (cap - n) : slot;
      for (int x = 0; x < d.n_prefix; ++x) ids[x - d.n_prefix] = d.prefix_ids[x];
      for (int x = 0; x < d.n_suffix; ++x) ids[n + x] = d.suffix_ids[x];
      a.tmp_off[my_sid] = static_cast<unsigned long long>(ids - d.n_prefix - a.arena);
      a.counts[my_sid] = static_cast<uint32_t>(n + n_extra);
    } else if (mine) {
      a.counts[my_sid] = 0u;
      a.tmp_off[my_sid] = 0;
      if (broken) {                              // "all normalized characters are not consumed." (sentencepiece_processor.cc:628)
        a.sent_status[my_sid] = static_cast<uint8_t>(kSsInternal);
        wv::atomic_add(&a.side->n_failed, 1ull);
      }
    }
    const unsigned long long c3 = wv::clock();
    if (mine && !broken) { ++tc.n_sent; tc.n_raw += my_len; tc.n_ids += static_cast<unsigned long long>(n + n_extra); }
    tc.cyc[2] += c2 - c1; tc.cyc[3] += c3 - c2;
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
      if (tc.n_backlog) wv::atomic_add(&a.side->n_backlog, tc.n_backlog);
    }
  }
}

}  // namespace spmx
#endif
