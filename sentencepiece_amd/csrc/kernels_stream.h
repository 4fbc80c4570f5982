// Unigram (and word-wise BPE) segmentation, streaming form: a wavefront takes a
// tile of up to 64 sentences and runs ONE SENTENCE PER LANE, so that all 64
// lanes carry an independent EncodeOptimized recurrence
// (src/unigram_model.cc:889-1020), with a per-lane working set in LDS that does
// not depend on the sentence length: 12 wavefronts fit a CU whatever the length
// class, and sentences of any length up to the class capacity run lane-parallel.
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
// Two kernels per length class: FAST (each lane normalizes its own sentence from
// HBM into its text column: fast_norm_stream for ASCII, norm_lane_general for
// tiles of mostly non-ASCII text) and GENERAL (normalize_wave into an LDS buffer,
// one sentence at a time, then a copy into the lane's column) for what FAST
// hands over through a device-side list and for models FAST cannot take.
// A workgroup is W wavefronts that share nothing but two read-only LDS tables
// (first trie level, byte classes; every wave writes identical copies, so no
// workgroup barrier is ever needed); each wave owns a private slice of LDS.
#ifndef SPMX_KERNELS_STREAM_H_
#define SPMX_KERNELS_STREAM_H_

namespace spmx {

// byte classes of the FAST normalizer (StreamLds::bcls)
constexpr uint32_t kBcComplex = 1u;    // not handled by fast_norm_stream: non-ASCII, or a charsmap rule may start here
constexpr uint32_t kStreamSharedBytes = 256u * 16u + 256u;   // roottab + bcls

// per-wave profiling counters
struct WaveCounters {
  unsigned long long n_sent = 0, n_raw = 0, n_ids = 0, n_trips = 0;
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

struct StreamLds {
  U4 *roottab;        // [256] first trie level (shared by the workgroup, read-only)
  uint8_t *bcls;      // [256] byte classes of the FAST normalizer (shared, read-only)
  uint8_t *raw;       // GENERAL: one raw sentence (rcap + 16)
  uint8_t *norm;      // GENERAL: its normalized form (ncap + 16)
  float *ring_s;      // [R][64]
  uint32_t *ring_b;   // [R][64]
  uint8_t *win;       // [64][W + 4]: lane l's window starts at win + l * (W + 4)
  uint32_t *stage;    // [2][64][4]: final back-pointer words of the lane's current block of 8 positions
  // BPE (kernels_bpe_stream.h) instead of the rings / window / staging block:
  uint32_t *asym;     // [256] symbol of every one-byte character (shared, aliases roottab)
  BpeWordLds bw;      // the lane's current word
  uint8_t *bwin;      // [64][kBpeWindow + 4] text windows
};

SPMX_HD inline uint32_t StreamWindow(uint32_t ring) { return 2u * ring; }
// model: 1 unigram, 2 BPE
SPMX_HD inline uint32_t StreamPrivateBytes(bool fast, int model, uint32_t rcap, uint32_t ncap, uint32_t ring) {
  const uint32_t stage = fast ? 0u : (((rcap + 16 + 15) & ~15u) + ((ncap + 16 + 15) & ~15u));
  const uint32_t work = model == 2 ? BpeWordLdsBytes() + 64u * (kBpeWindow + 4u)
                                   : 64u * ring * 8u + 64u * (StreamWindow(ring) + 4u) + 2u * 64u * 16u;
  return stage + work;
}
SPMX_HD inline uint32_t StreamLdsBytes(bool fast, int model, uint32_t rcap, uint32_t ncap, uint32_t ring, uint32_t waves) {
  return kStreamSharedBytes + waves * StreamPrivateBytes(fast, model, rcap, ncap, ring);
}
// HBM scratch of one wavefront for a class whose normalized sentences have at most tcap bytes
SPMX_HD inline uint64_t StreamTextDwords(uint32_t tcap, uint32_t ring) {      // uint32 [dw][64]
  return (static_cast<uint64_t>(tcap + 3) / 4 + StreamWindow(ring) / 4 + 4) * 64u;
}
SPMX_HD inline uint32_t StreamBpStride(uint32_t tcap) { return (tcap + 16u) & ~7u; }   // words per lane, whole blocks of 8
SPMX_HD inline uint64_t StreamBpWords(uint32_t tcap) { return static_cast<uint64_t>(StreamBpStride(tcap)) * 64u; }   // uint32 [lane][stride]

SPMX_DEVICE StreamLds carve_stream(unsigned char *base, bool fast, int model, uint32_t rcap, uint32_t ncap, uint32_t ring,
                                   int wave) {
  StreamLds t;
  t.roottab = reinterpret_cast<U4 *>(base);
  t.asym = reinterpret_cast<uint32_t *>(base);
  t.bcls = base + 256u * 16u;
  unsigned char *mine = base + kStreamSharedBytes + static_cast<uint32_t>(wave) * StreamPrivateBytes(fast, model, rcap, ncap, ring);
  t.raw = mine;
  t.norm = mine + ((rcap + 16 + 15) & ~15u);
  if (!fast) mine += ((rcap + 16 + 15) & ~15u) + ((ncap + 16 + 15) & ~15u);
  t.ring_s = reinterpret_cast<float *>(mine);
  t.ring_b = reinterpret_cast<uint32_t *>(mine + 64u * ring * 4u);
  t.win = mine + 64u * ring * 8u;
  t.stage = reinterpret_cast<uint32_t *>(t.win + 64u * (StreamWindow(ring) + 4u));
  t.bw.sym = reinterpret_cast<uint32_t *>(mine);
  t.bw.score = reinterpret_cast<float *>(mine + kBpeWordMax * 64u * 4u);
  t.bw.merged = reinterpret_cast<uint32_t *>(mine + kBpeWordMax * 64u * 8u);
  t.bw.len = mine + kBpeWordMax * 64u * 12u;
  t.bwin = mine + BpeWordLdsBytes();
  return t;
}

// Normalize() of one all-ASCII sentence by ONE lane (src/normalizer.cc:71-186 with every NormalizePrefix result
// being the byte itself, :231-244): raw text in HBM (16-byte aligned loads; a block that holds a byte of the
// sentence lies in the same page as that byte, so the over-read at either end stays inside the caller's mapping)
// -> the lane's text column gt[dw * 64], four bytes at a time.  Valid when the space symbol is one byte wide
// (kNfCompressSp, or no whitespace escaping) and the model has no user-defined symbols; not for
// whitespace-as-suffix models (the suffix would have to be patched into a dword that is already stored).  A byte
// whose bcls entry says kBcComplex makes the lane give up (-1).
// *n_sp: how many bytes of the result are the space symbol (sizes the id slot under byte fallback).
SPMX_DEVICE int fast_norm_stream(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int L, uint32_t *gt,
                                 const uint8_t *bcls, int tcap, int *n_sp) {
  const uint32_t F = d.flags;
  const bool rm = (F & kNfRemoveExtraWs) != 0;
  const uint32_t sp = (F & kNfCompressSp) ? kSpByte : 0x20u;
  const bool has_map = (F & kNfHasCharsmap) != 0;
  int w = 0, nsp = 0;
  uint32_t acc = 0;
  if (F & kNfAddDummyPrefix) { acc = sp; w = 1; nsp = 1; }   // :128
  bool P = rm;                    // is_prev_space (:130)
  int wl = w;                     // output length up to the last non-space byte (:166-176 trailing spaces)
  bool seen = false;              // some prefix is not " " (:86-100)
  uint32_t bad = 0;
  uint32_t carry = 0x80u;         // the last byte of the previous block
  int skip = 0;                   // continuation bytes of a validated character still to copy
  const uint64_t q0 = beg & ~15ull;
  const uint8_t *blk = gtext + q0;
  int rel = static_cast<int>(q0 - beg);         // index of the block's first byte within the sentence (<= 0 at first)
  Q4 cur = *reinterpret_cast<const Q4 *>(blk);
  while (rel < L) {
    Q4 nxt = cur;
    if (rel + 16 < L) nxt = *reinterpret_cast<const Q4 *>(blk + 16);
    const uint32_t wd[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((wd[q] & 0x80808080u) == 0u) {
        // ---- four ASCII bytes: every NormalizePrefix result is the byte itself ----
#pragma unroll
        for (int k = 4 * q; k < 4 * q + 4; ++k) {
          const uint32_t c = (wd[q] >> (8 * (k & 3))) & 0xFFu;
          if (static_cast<uint32_t>(rel + k) < static_cast<uint32_t>(L)) {
            bad |= bcls[c];
            const bool is_sp = c == 0x20u;
            if (!is_sp || !P) {                     // :137-138 a space after a space is dropped
              acc |= (is_sp ? sp : c) << (8 * (w & 3));
              ++w;
              nsp += is_sp ? 1 : 0;
              if ((w & 3) == 0) { gt[((w >> 2) - 1) * 64] = acc; acc = 0; }
            }
            P = is_sp && rm;                        // :154-162
            if (!is_sp) { wl = w; seen = true; }
          }
        }
      } else {
        // ---- a dword with a non-ASCII byte (rare in this kernel's tiles).  A character whose first two bytes
        // start no charsmap key (tables.cc npair -- the filter of the position-parallel normalizer) normalizes to
        // itself (:231-244), a malformed byte to U+FFFD (util.cc:51-84); a possible rule, or a literal U+2581,
        // leaves the sentence to the general normalizers. ----
#pragma unroll
        for (int k = 4 * q; k < 4 * q + 4; ++k) {
          const uint32_t c = (wd[q] >> (8 * (k & 3))) & 0xFFu;
          if (static_cast<uint32_t>(rel + k) < static_cast<uint32_t>(L)) {
            if (c < 0x80u) {
              bad |= bcls[c];
              const bool is_sp = c == 0x20u;
              if (!is_sp || !P) {
                acc |= (is_sp ? sp : c) << (8 * (w & 3));
                ++w;
                nsp += is_sp ? 1 : 0;
                if ((w & 3) == 0) { gt[((w >> 2) - 1) * 64] = acc; acc = 0; }
              }
              P = is_sp && rm;
              if (!is_sp) { wl = w; seen = true; }
            } else {
              uint32_t o0 = c, o1 = 0, o2 = 0;
              int n_out = 1;
              const int rem = L - (rel + k);
              if (skip > 0) {
                --skip;
              } else {
                const uint32_t b1 = rem >= 2 ? (wd[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xFFu : 0u;
                const uint32_t b2 = rem >= 3 ? (wd[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 0xFFu : 0u;
                const uint32_t b3 = rem >= 4 ? (wd[(k + 3) >> 2] >> (8 * ((k + 3) & 3))) & 0xFFu : 0u;
                const uint32_t pb = k > 0 ? (wd[(k > 0 ? k - 1 : 0) >> 2] >> (8 * ((k > 0 ? k - 1 : 0) & 3))) & 0xFFu : carry;
                const uint32_t prevc = rel + k > 0 ? pb : 0x80u;      // nothing before the first byte of the sentence
                const bool t1 = (b1 & 0xC0u) == 0x80u, t2 = (b2 & 0xC0u) == 0x80u, t3 = (b3 & 0xC0u) == 0x80u;
                int mb = 0;
                if (rem >= 2 && (c & 0xE0u) == 0xC0u) {
                  if (t1 && ((c & 0x1Fu) << 6 | (b1 & 0x3Fu)) >= 0x80u) mb = 2;
                } else if (rem >= 3 && (c & 0xF0u) == 0xE0u) {
                  const uint32_t cp = (c & 0x0Fu) << 12 | (b1 & 0x3Fu) << 6 | (b2 & 0x3Fu);
                  if (t1 && t2 && cp >= 0x800u && (cp < 0xD800u || cp >= 0xE000u)) mb = 3;
                  if (c == 0xE2u && b1 == 0x96u && b2 == 0x81u) bad |= kBcComplex;
                } else if (rem >= 4 && (c & 0xF8u) == 0xF0u) {
                  const uint32_t cp = (c & 0x07u) << 18 | (b1 & 0x3Fu) << 12 | (b2 & 0x3Fu) << 6 | (b3 & 0x3Fu);
                  if (t1 && t2 && t3 && cp >= 0x10000u && cp <= 0x10FFFFu) mb = 4;
                }
                if (has_map) {
                  if ((d.npair[(c << 8 | b1) >> 5] >> (b1 & 31u)) & 1u) bad |= kBcComplex;          // a key may start here
                  if (prevc < 0x80u && ((d.npair[(prevc << 8 | c) >> 5] >> (c & 31u)) & 1u)) bad |= kBcComplex;   // or at the ASCII byte before
                }
                if (mb) skip = mb - 1;
                else { o0 = 0xEFu; o1 = 0xBFu; o2 = 0xBDu; n_out = 3; }
              }
              // a malformed byte grows into three: keep "what is written + what is left to read" within the
              // column (all other bytes produce at most one), else leave the sentence to the general normalizers
              if (w + rem + 2 > tcap) bad |= kBcComplex;
              else {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                  if (j < n_out) {
                    acc |= (j == 0 ? o0 : (j == 1 ? o1 : o2)) << (8 * (w & 3));
                    ++w;
                    if ((w & 3) == 0) { gt[((w >> 2) - 1) * 64] = acc; acc = 0; }
                  }
                }
              }
              P = false;
              wl = w;
              seen = true;
            }
          }
        }
      }
    }
    carry = cur.w >> 24;
    cur = nxt;
    blk += 16;
    rel += 16;
  }
  gt[(w >> 2) * 64] = acc;                      // the last, partial dword
  if (bad & kBcComplex) return -1;
  if (rm) {
    if (!seen) return 0;                        // :86-100 nothing but spaces
    nsp -= w - wl;                              // the trimmed tail is nothing but space symbols
    w = wl;
  }
  *n_sp = nsp;
  return w;
}

// ---- Normalize() of one sentence of ANY content by ONE lane (src/normalizer.cc:71-253) ------------------------
// The same loop as the reference's: one NormalizePrefix result after another -- longest charsmap rule by a Darts
// walk (darts.h:467-513), else one UTF-8 character, else U+FFFD for a malformed byte (util.cc:51-84) -- through
// the whitespace state machine, with the output going to the lane's text column four bytes at a time.  Same
// preconditions as fast_norm_stream (one-byte space symbol, no user-defined symbols, no whitespace-as-suffix).
// Raw bytes come through a kRawWin-byte LDS window filled 16 bytes at a time; a rule walk that would outrun the
// window, or an output longer than tcap, makes the lane give up (-1) and the sentence goes to the GENERAL kernel.
constexpr int kRawWin = 64;
constexpr uint32_t kLaneGeneralMaxRaw = 4096;  // length classes whose sentences norm_lane_general takes (EncodeArgs::lane_general_max_raw)

SPMX_DEVICE int norm_lane_general(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int L, uint32_t *gt, int tcap,
                                  uint8_t *rawwin, int *n_sp) {
  const uint32_t F = d.flags;
  const bool rm = (F & kNfRemoveExtraWs) != 0;
  const bool one = (F & kNfCompressSp) != 0;
  const uint32_t sp1 = one ? kSpByte : 0x20u;              // the space symbol (one byte wide here)
  const bool has_map = (F & kNfHasCharsmap) != 0;
  const uint32_t droot = has_map ? DartsOffset(d.ndarts[0]) : 0u;
  // raw byte i of the sentence (0 <= i < L, within kRawWin - 16 of every byte still needed) through the LDS window,
  // which is indexed by the low bits of the absolute address and filled one aligned 16-byte block at a time
  int hi;                                                  // raw bytes [0, hi) have been loaded
  {
    const uint64_t q0 = beg & ~15ull;
    *reinterpret_cast<Q4 *>(rawwin + (q0 & (kRawWin - 1))) = *reinterpret_cast<const Q4 *>(gtext + q0);
    hi = static_cast<int>(q0 + 16 - beg);
  }
  auto raw = [&](int i) __attribute__((always_inline)) -> uint32_t {
    while (i >= hi) {
      const uint64_t q = beg + static_cast<uint64_t>(hi);                // 16-byte aligned
      *reinterpret_cast<Q4 *>(rawwin + (q & (kRawWin - 1))) = *reinterpret_cast<const Q4 *>(gtext + q);
      hi += 16;
    }
    return rawwin[(beg + static_cast<uint64_t>(i)) & (kRawWin - 1)];
  };
  int w = 0, wl = 0, nsp = 0;
  uint32_t acc = 0;
  bool giveup = false;
  auto emit = [&](uint32_t b) __attribute__((always_inline)) {
    if (w >= tcap) { giveup = true; return; }
    acc |= b << (8 * (w & 3));
    ++w;
    if ((w & 3) == 0) { gt[((w >> 2) - 1) * 64] = acc; acc = 0; }
    if (b != sp1) wl = w;                                  // :166-176 trailing space symbols are cut at the end
    else ++nsp;
  };
  // NormalizePrefix at raw offset p (:195-253): kind 0 raw bytes [src, src + len), 1 rule string
  // nblob[src, src + len), 2 U+FFFD, 3 the space symbol (a literal U+2581 under kNfCompressSp)
  struct Pfx { int kind, len, consumed; uint32_t src; };
  auto prefix = [&](int p) __attribute__((always_inline)) -> Pfx {
    const uint32_t b0 = raw(p);
    const int rem = L - p;
    int rule_len = 0;
    uint32_t rule_off = 0;
    bool walk = has_map;
    if (walk) {                                            // no key starts with these two bytes (tables.cc npair)
      const uint32_t b1 = rem >= 2 ? raw(p + 1) : 0u;
      walk = ((d.npair[(b0 << 8 | b1) >> 5] >> (b1 & 31u)) & 1u) != 0;
    }
    if (walk) {                                            // commonPrefixSearch, longest key (:218-228)
      uint32_t pos = droot;
      for (int depth = 0; p + depth < L;) {
        if (depth >= kRawWin - 20) { giveup = true; break; }
        const uint32_t c = raw(p + depth);
        pos ^= c;
        if (pos >= d.ndarts_n) break;
        const uint32_t u = d.ndarts[pos];
        if ((u & 0x800000FFu) != c) break;                 // unit.label() == c
        pos ^= DartsOffset(u);
        ++depth;
        if ((u >> 8) & 1u) {                               // has_leaf: the value sits in the unit at pos
          if (pos >= d.ndarts_n) break;
          rule_len = depth;
          rule_off = d.ndarts[pos] & 0x7FFFFFFFu;
        }
      }
    }
    Pfx r{0, 0, 0, 0};
    if (rule_len > 0) {                                    // :245-250 the C string at normalized_[value]
      int n = 0;
      while (rule_off + static_cast<uint32_t>(n) < d.nblob_n && d.nblob[rule_off + n] != 0) ++n;
      r = Pfx{1, n, rule_len, rule_off};
    } else {
      // :231-244 one UTF-8 character (DecodeUTF8, util.cc:51-84)
      int mb = 1;
      bool ok = b0 < 0x80u, lit_sp = false;
      if (!ok) {
        const uint32_t b1 = rem >= 2 ? raw(p + 1) : 0u, b2 = rem >= 3 ? raw(p + 2) : 0u, b3 = rem >= 4 ? raw(p + 3) : 0u;
        const bool t1 = (b1 & 0xC0u) == 0x80u, t2 = (b2 & 0xC0u) == 0x80u, t3 = (b3 & 0xC0u) == 0x80u;
        if (rem >= 2 && (b0 & 0xE0u) == 0xC0u) {
          const uint32_t cp = (b0 & 0x1Fu) << 6 | (b1 & 0x3Fu);
          if (t1 && cp >= 0x80u) { ok = true; mb = 2; }
        } else if (rem >= 3 && (b0 & 0xF0u) == 0xE0u) {
          const uint32_t cp = (b0 & 0x0Fu) << 12 | (b1 & 0x3Fu) << 6 | (b2 & 0x3Fu);
          if (t1 && t2 && cp >= 0x800u && (cp < 0xD800u || cp >= 0xE000u)) { ok = true; mb = 3; }
          lit_sp = ok && one && b0 == 0xE2u && b1 == 0x96u && b2 == 0x81u;
        } else if (rem >= 4 && (b0 & 0xF8u) == 0xF0u) {
          const uint32_t cp = (b0 & 0x07u) << 18 | (b1 & 0x3Fu) << 12 | (b2 & 0x3Fu) << 6 | (b3 & 0x3Fu);
          if (t1 && t2 && t3 && cp >= 0x10000u && cp <= 0x10FFFFu) { ok = true; mb = 4; }
        }
      }
      if (lit_sp) r = Pfx{3, 1, 3, 0};
      else if (ok) r = Pfx{0, mb, mb, static_cast<uint32_t>(p)};
      else r = Pfx{2, 3, 1, 0};
    }
    return r;
  };
  auto sp_byte = [&](const Pfx &x, int k) __attribute__((always_inline)) -> uint32_t {
    if (x.kind == 0) return raw(static_cast<int>(x.src) + k);
    if (x.kind == 1) return d.nblob[x.src + static_cast<uint32_t>(k)];
    if (x.kind == 2) return k == 0 ? 0xEFu : (k == 1 ? 0xBFu : 0xBDu);
    return kSpByte;
  };
  int p = 0;
  if (rm) {                                                // :84-95 prefixes that normalize to exactly " "
    while (p < L && !giveup) {
      const Pfx x = prefix(p);
      if (!(x.len == 1 && x.kind != 3 && sp_byte(x, 0) == 0x20u)) break;
      p += x.consumed;
    }
  }
  if (giveup) return -1;
  if (p >= L) { gt[0] = 0; return 0; }                     // :98-100 nothing but whitespace
  if (F & kNfAddDummyPrefix) emit(sp1);                    // :128
  bool is_prev_space = rm;                                 // :130
  while (p < L && !giveup) {
    const Pfx x = prefix(p);
    if (giveup) break;
    int k = 0;
    if (x.kind != 3) while (is_prev_space && k < x.len && sp_byte(x, k) == 0x20u) ++k;      // :137-138
    if (k < x.len) {
      uint32_t last = 0;
      for (; k < x.len; ++k) {
        last = sp_byte(x, k);
        emit((x.kind != 3 && last == 0x20u) ? sp1 : last);  // :143-152 (whitespace escaping = the one-byte symbol)
      }
      is_prev_space = x.kind != 3 && last == 0x20u;        // :154
    }
    p += x.consumed;
    if (!rm) is_prev_space = false;                        // :160-162
  }
  if (giveup) return -1;
  gt[(w >> 2) * 64] = acc;                                 // the last, partial dword
  if (rm) { nsp -= w - wl; w = wl; }
  *n_sp = nsp;
  return w;
}

// True when the preconditions of the per-lane normalizers hold for this model (host and device agree on it).
SPMX_HD inline bool StreamFastEligible(uint32_t flags) {
  return !(flags & kNfHasUserDefined) && ((flags & kNfCompressSp) || !(flags & kNfEscapeWs)) &&
         !((flags & kNfAddDummyPrefix) && (flags & kNfWsSuffix));
}

// piece_score; UDS = false drops the user-defined branch (FAST kernels: such models never reach them)
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
//     iteration from the lane's text column gt[] -- the load is issued next to the trie probe and lands in the
//     window at the top of the next iteration, by which time the probe wait has covered it;
//   * best_path_ends_at lives in the rings only: ring_s / ring_b slot of position e is [(e & rm) * 64];
//     ring_b == 0 means "not reached" (:984);
//   * when the start moves from s to s2, position s2's back-pointer word is final: it goes to the lane's staging
//     block st[] (position p at st[((p >> 2) & 1) * 256 + (p & 3)]), the block of 8 positions that s2 leaves behind
//     is written to gb[] as two 16-byte stores, and the ring slots of the positions (s, s2] are cleared for the
//     positions that will reuse them R later.  gb[] is this lane's row: gb[p] for position p.
// Returns the number of iterations (wave-uniform).
// RING > 0: the ring size is a compile-time constant (index masks and the distances between the LDS arrays fold
// into immediates); RING == 0: taken from rm_in / wmask_in.  UDS: the model may have USER_DEFINED pieces.
template <int RING, bool UDS>
SPMX_DEVICE int unigram_stream_lane(const SpmxDev &d, const uint32_t *gt, uint32_t *gb, int nlen, float *ring_s,
                                    uint32_t *ring_b, uint32_t rm_in, uint8_t *win, uint32_t wmask_in, uint32_t *st,
                                    const U4 *roottab, bool active_in) {
  const uint32_t rm = RING ? static_cast<uint32_t>(RING - 1) : rm_in;
  const uint32_t wmask = RING ? static_cast<uint32_t>(2 * RING - 1) : wmask_in;
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
  uint32_t cq = 0;
  int nf = 0;                                     // next text dword to fetch; the window holds dwords [nf - W/4, nf)
  bool pf_pend = false;
  uint32_t pf = 0;
  if (active) {
    for (uint32_t k = 0; k <= rm; ++k) ring_b[k * 64] = 0u;
    ring_s[0] = 0.f;                              // best_path_ends_at[0].best_path_score = 0
    for (int k = 0; k < W / 4; ++k) *reinterpret_cast<uint32_t *>(win + 4 * k) = gt[k * 64];
    nf = W / 4;
    cs = win[0];
    cs1 = win[1];
    rn = roottab[cs];
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
    const bool termC = rootC && (r.x & kDatTerminalDev) && !(r.y & kPtUnused);
    const bool contC = rootC && s2 + 1 < nlen && ((r.w >> ChildBit(cs1)) & 1u);
    const bool nwalking = cont || contC;
    const uint32_t nnode = cont ? (u.x >> kDatBaseShiftDev) : (r.x >> kDatBaseShiftDev);
    const uint32_t nc = cont ? cq : cs1;
    const U4 uA = u;
    // window refill: dword nf may replace positions [4 nf - W, 4 nf - W + 4), which are dead once they lie below s
    pf_pend = active && 4 * nf + 4 <= s + W && 4 * nf < nlen + 8;
    if (pf_pend) { pf = gt[nf * 64]; ++nf; }
    if (nwalking) u = ptrie[nnode ^ nc];          // next probe
    // ---------------- data ----------------
    const int eA = s + dep1;
    const int eB = s2;
    const int eC = s2 + 1 <= nlen ? s2 + 1 : nlen;
    const uint32_t oA = (static_cast<uint32_t>(matchA ? eA : 0) & rm) << 6;
    const uint32_t oB = (static_cast<uint32_t>(eB) & rm) << 6;
    const uint32_t oC = (static_cast<uint32_t>(eC) & rm) << 6;
    uint32_t bA = ring_b[oA], bB = ring_b[oB], bC = ring_b[oC];
    float rA = ring_s[oA], rB = ring_s[oB], rC = ring_s[oC];
    // (A) the piece that just matched
    const double candA = piece_score_u<UDS>(uA, dep1, max_score) + static_cast<double>(sbest);  // :982-983
    const bool updA = termA && (bA == 0 || candA > static_cast<double>(rA));             // :984-989
    const float nvA = static_cast<float>(candA);
    const uint32_t wA = (uA.y & kBwIdMask) | (static_cast<uint32_t>(dep1) << kBwLenShift);
    if (updA && eB == eA) { rB = nvA; bB = wA; }
    if (updA && eC == eA) { rC = nvA; bC = wA; }
    const bool single2 = single || (termA && dep1 == mb);                                // :990
    // (B) UNK for the start that is over
    const float candB = unk_score + sbest;                                               // :997-1001, float
    const bool updB = ended && !single2 && (bB == 0 || candB > rB);
    const float sbest2 = updB ? candB : rB;
    const uint32_t finB = updB ? ((static_cast<uint32_t>(mb) << kBwLenShift) | kBwUnk) : bB;
    // (C) a one-byte piece of the next start
    const double candC = piece_score_u<UDS>(r, 1, max_score) + static_cast<double>(sbest2);
    const bool updC = termC && (bC == 0 || candC > static_cast<double>(rC));
    if (updA) { ring_s[oA] = nvA; ring_b[oA] = wA; }
    if (updC) { ring_s[oC] = static_cast<float>(candC); ring_b[oC] = (r.y & kBwIdMask) | (1u << kBwLenShift); }
    if (ended) {
      if ((s2 >> 3) != (s >> 3)) {                // the block of 8 positions behind s2 is complete
        const Q4 lo = *reinterpret_cast<const Q4 *>(st), hi = *reinterpret_cast<const Q4 *>(st + 256);
        uint32_t *blk = gb + ((s >> 3) << 3);
        *reinterpret_cast<Q4 *>(blk) = lo;
        *reinterpret_cast<Q4 *>(blk + 4) = hi;
      }
      st[((static_cast<uint32_t>(eB) >> 2) & 1u) * 256u + (static_cast<uint32_t>(eB) & 3u)] = finB;   // position s2 is final
      // the positions just passed, (s, s2], are dead: free their ring slots (after this iteration's reads and writes)
      if (mb > 0) ring_b[oB] = 0u;
      if (mb > 1)                                  // a multi-byte character: its inner positions too (rare in ASCII text)
        for (int k = 1; k < mb; ++k) ring_b[(static_cast<uint32_t>(s2 - k) & rm) << 6] = 0u;
    }
    // ---------------- commit ----------------
    if (ended) {
      active = begin;
      s = s2;
      mb = begin ? mb2 : 0;
      sbest = sbest2;
      single = termC && mb2 == 1;
      dep = rootC ? 1 : 0;
      if (begin) {
        cs = win[static_cast<uint32_t>(s2 + mb2) & wmask];
        cs1 = win[static_cast<uint32_t>(s2 + mb2 + 1) & wmask];
        rn = roottab[cs];
      }
    } else {
      dep = dep1;
      single = single2;
    }
    walking = nwalking;
    c = nc;
    if (nwalking) cq = win[static_cast<uint32_t>(s + dep + 1) & wmask];
  }
  if (active_in && nlen > 0) {                    // the last block (it holds position nlen)
    const Q4 lo = *reinterpret_cast<const Q4 *>(st), hi = *reinterpret_cast<const Q4 *>(st + 256);
    uint32_t *blk = gb + ((nlen >> 3) << 3);
    *reinterpret_cast<Q4 *>(blk) = lo;
    *reinterpret_cast<Q4 *>(blk + 4) = hi;
  }
  return trips;
}

// Backtrack (:1010-1018) + id post-processing (sentencepiece_processor.cc:581-613) of this lane's sentence:
// follows the lane's row gb[] from position nlen to 0 and writes the ids as it goes, last piece first, into slot[0, cap):
// forward order fills the slot from its END (ids end up in slot[cap - n, cap)), `reverse` fills it from the start.
// Returns n, or -1 on a broken chain / overflow.
// `tslot` (spans form, else null): the slot's twin in EncodeArgs::arena_tb, receives every token's begin.
SPMX_DEVICE int emit_stream_lane(const SpmxDev &d, const uint32_t *gt, const uint32_t *gb, int nlen, int32_t *slot,
                                 int32_t *tslot, int cap, bool active) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  int e = nlen, n = 0;
  bool right_unk = false, ok = true;
  active = active && nlen > 0;
  while (wv::any(active)) {
    if (active) {
      const uint32_t w = gb[e];
      const int len = static_cast<int>((w >> kBwLenShift) & kBwLenMask);
      if (len == 0 || len > e) { ok = false; active = false; continue; }
      const int tb = e - len;
      if (w & kBwUnk) {
        if (bf) {                                   // one BYTE id per byte of the unknown piece (:581-603)
          const bool sp = stream_text_byte(gt, tb) == spb;
          const int nb = sp ? 3 : len;
          if (n + nb > cap) { ok = false; active = false; continue; }
          for (int x = nb - 1; x >= 0; --x) {
            const uint32_t byte = sp ? (x == 0 ? 0xE2u : (x == 1 ? 0x96u : 0x81u)) : stream_text_byte(gt, tb + x);
            slot[reverse ? n : cap - 1 - n] = d.byte_ids[byte];
            if (tslot) tslot[reverse ? n : cap - 1 - n] = tb;
            ++n;
          }
        } else if (!right_unk) {                    // a run of unknown pieces yields one id (:609-613)
          if (n >= cap) { ok = false; active = false; continue; }
          slot[reverse ? n : cap - 1 - n] = d.unk_id;
          if (tslot) tslot[reverse ? n : cap - 1 - n] = tb;
          ++n;
        } else if (tslot) {                         // the run grows to the left: so does the merged token
          tslot[reverse ? n - 1 : cap - n] = tb;
        }
        right_unk = true;
      } else {
        right_unk = false;
        if (n >= cap) { ok = false; active = false; continue; }
        slot[reverse ? n : cap - 1 - n] = static_cast<int32_t>(w & kBwIdMask);
        if (tslot) tslot[reverse ? n : cap - 1 - n] = tb;
        ++n;
      }
      e = tb;
      if (e <= 0) active = false;
    }
  }
  return ok ? n : -1;
}

// Persistent body of the streaming kernels.  MODEL: 1 unigram, 2 BPE (word-wise models only).  RING: see
// unigram_stream_lane (0 = a.ring).
template <bool FAST, int MODEL, int RING = 0>
SPMX_DEVICE void encode_stream_block(const EncodeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const uint32_t ring = RING ? static_cast<uint32_t>(RING) : a.ring;
  const StreamLds T = carve_stream(smem, FAST, MODEL, a.rcap, a.ncap, ring, wv::wave_in_block());
  const uint32_t rm = ring - 1;
  const uint32_t W = StreamWindow(ring);
  float *my_rs = T.ring_s + lane;
  uint32_t *my_rb = T.ring_b + lane;
  uint8_t *my_win = T.win + static_cast<uint32_t>(lane) * (W + 4u);
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
  const uint32_t count = *a.list_count;
  const uint32_t wave_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block());
  const uint32_t n_waves = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block());
  // this wave's scratch slab; tcap = capacity of a text column in bytes
  const uint32_t tcap = a.stream_tcap;
  uint32_t *gt = a.stream_text + static_cast<uint64_t>(wave_id) * StreamTextDwords(tcap, ring) + static_cast<uint32_t>(lane);
  uint32_t *gb = a.stream_bp + static_cast<uint64_t>(wave_id) * StreamBpWords(tcap) + static_cast<uint32_t>(lane) * StreamBpStride(tcap);
  uint32_t *my_st = T.stage + static_cast<uint32_t>(lane) * 4u;
  // sentences per tile: 64, or fewer when the list is too short to give every wave a full tile
  uint32_t tw = (count + n_waves - 1) / n_waves;
  tw = tw < 1u ? 1u : (tw > 64u ? 64u : tw);
  const uint32_t tiles = (count + tw - 1) / tw;
  const int n_extra = d.n_prefix + d.n_suffix;
  WaveCounters tc;
  // Tiles come from a queue (one atomic per tile), not from a fixed stride: a wave that starts late -- its CU was
  // held by another stream's kernel, e.g. the RCCL all-gather of the previous batch -- takes fewer tiles instead of
  // finishing its fixed share after everybody else, and uneven tiles even out.
  for (uint32_t tile = wave_id;; tile += n_waves) {
    if (a.tile_cursor) {
      uint32_t t = 0;
      if (lane == 0) t = wv::atomic_add(a.tile_cursor, 1u);
      tile = wv::shfl(t, 0);
      // the class lists are sorted by length (classify's sub-buckets): longest tiles first, so that the tiles
      // still running when the queue empties are the short ones
      if (tile < tiles && !a.tiles_ascending) tile = tiles - 1u - tile;
    }
    if (tile >= tiles) break;
    const uint32_t first = tile * tw;
    const int cnt = static_cast<int>(count - first < tw ? count - first : tw);
    uint32_t my_sid = 0;
    uint64_t my_beg = 0;
    uint32_t my_len = 0;
    bool too_long = false;
    if (lane < cnt) {
      my_sid = a.list[first + lane];
      my_beg = a.offs[my_sid];
      my_len = static_cast<uint32_t>(a.offs[my_sid + 1] - my_beg);
      too_long = a.offs[my_sid + 1] - my_beg > a.rcap;                    // only reachable in the last class
    }
    const unsigned long long c0 = wv::clock();
    unsigned long long t_load = 0;
    bool mine = false;
    int my_nlen = 0, my_nsp = 0;     // normalized length; how many of its bytes are the space symbol
    if (wv::any(too_long)) {
      uint64_t m = wv::ballot(too_long);
      while (m) { const int i = wv::ffs64(m) - 1; m &= m - 1; fail_sentence(a, wv::shfl(my_sid, i), kStTooLong, lane); }
    }
    if (FAST) {
      const bool go = lane < cnt && !too_long;
      int nlen = 0;
      if (go && my_len > 0) nlen = fast_norm_stream(d, a.text, my_beg, static_cast<int>(my_len), gt, T.bcls, static_cast<int>(tcap), &my_nsp);
      // Not plain ASCII.  A tile that is mostly such sentences (CJK text ...) normalizes them here, one per lane
      // (the raw window borrows the rings, idle until the search); a stray one in an ASCII tile would hold the
      // other 63 lanes up for its whole length, and long sentences are better off position-parallel: both go to
      // the GENERAL kernel.
      // Document-length classes have no GENERAL kernel (its LDS staging holds one whole sentence): every
      // non-ASCII sentence is normalized here, and one that cannot be fails the call.
      const bool no_general = a.hard_list == nullptr;
      const bool many = no_general ||
                        (wv::popc64(wv::ballot(nlen < 0)) >= static_cast<int>(a.lane_general_min_lanes) &&
                         a.rcap <= a.lane_general_max_raw && !a.no_lane_general);
      if (many && nlen < 0)
        nlen = norm_lane_general(d, a.text, my_beg, static_cast<int>(my_len), gt, static_cast<int>(tcap),
                                 reinterpret_cast<uint8_t *>(T.ring_s) + static_cast<uint32_t>(lane) * (kRawWin + 16), &my_nsp);
      const bool hard = go && nlen < 0;
      if (go && nlen >= 0) { mine = true; my_nlen = nlen; }
      const uint64_t hm = wv::ballot(hard);
      if (hm && no_general) {
        uint64_t m = hm;
        while (m) { const int i = wv::ffs64(m) - 1; m &= m - 1; fail_sentence(a, wv::shfl(my_sid, i), kStTooLong, lane); }
      } else if (hm) {                          // hand the sentence to the GENERAL kernel of this class
        const int leader = wv::ffs64(hm) - 1;
        uint32_t hb = 0;
        if (lane == leader) hb = wv::atomic_add(a.hard_count, static_cast<uint32_t>(wv::popc64(hm)));
        hb = wv::shfl(hb, leader);
        if (hard) a.hard_list[hb + static_cast<uint32_t>(wv::popc64(hm & ((1ull << lane) - 1ull)))] = my_sid;
      }
    } else {
      for (int i = 0; i < cnt; ++i) {
        const unsigned long long l0 = wv::clock();
        if (wv::shfl(too_long ? 1 : 0, i)) continue;
        const uint32_t L = wv::shfl(my_len, i);
        const uint32_t sid = wv::shfl(my_sid, i);
        const uint64_t beg = static_cast<uint64_t>(wv::shfl(static_cast<uint32_t>(my_beg >> 32), i)) << 32 |
                             wv::shfl(static_cast<uint32_t>(my_beg), i);
        const uint8_t *src = a.text + beg;
        for (uint32_t p = static_cast<uint32_t>(lane); p < L; p += 64) T.raw[p] = src[p];
        wv::sync();
        t_load += wv::clock() - l0;
        int nlen = 0;
        if (L > 0) nlen = normalize_wave(d, T.raw, static_cast<int>(L), T.norm, static_cast<int>(a.ncap < tcap ? a.ncap : tcap), lane);
        wv::sync();
        if (nlen < 0) {                         // does not fit this class: hand it on (or fail in the last class)
          if (a.next_list) { if (lane == 0) a.next_list[wv::atomic_add(a.next_count, 1u)] = sid; }
          else fail_sentence(a, sid, kStTooLong, lane);
          continue;
        }
        // norm[0, nlen) -> lane i's text column, a dword per lane per step
        uint32_t *col = gt - lane + i;
        for (int p4 = lane; p4 * 4 < nlen; p4 += 64) col[p4 * 64] = *reinterpret_cast<const uint32_t *>(T.norm + 4 * p4);
        int nsp = 0;
        if ((d.flags & kNfByteFallback) && (d.flags & kNfCompressSp)) {          // sizes the id slot below
          int cnt_sp = 0;
          for (int p = lane; p < nlen; p += 64) cnt_sp += T.norm[p] == kSpByte ? 1 : 0;
          wave_excl_scan(cnt_sp, lane, &nsp);
        }
        if (lane == i) { mine = true; my_nlen = nlen; my_nsp = nsp; }
        wv::sync();                             // norm is rewritten by the next sentence
      }
    }
    wv::sync_global();                          // text columns written by other lanes are read below
    const unsigned long long c1 = wv::clock();
    tc.cyc[0] += t_load; tc.cyc[1] += (c1 - c0) - t_load;
    // ---- a slot of cap ids per sentence in the arena ----
    int cap = 0;
    // at most one id per normalized byte -- three for a space symbol that falls back to its bytes
    if (mine) cap = ((d.flags & kNfByteFallback) && (d.flags & kNfCompressSp)) ? my_nlen + 2 * my_nsp : my_nlen;
    const int room = mine ? cap + n_extra : 0;
    int total = 0;
    const int excl = wave_excl_scan(room, lane, &total);
    unsigned long long base = 0;
    if (lane == 0 && total > 0) base = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(total));
    base = (static_cast<unsigned long long>(wv::shfl(static_cast<uint32_t>(base >> 32), 0)) << 32) |
           wv::shfl(static_cast<uint32_t>(base), 0);
    const bool overflow = base + static_cast<unsigned long long>(total) > a.arena_cap;
    if (overflow && lane == 0) wv::atomic_or(a.status, kStArenaOverflow);
    int32_t *slot = a.arena + base + static_cast<unsigned long long>(excl) + d.n_prefix;
    int32_t *tslot = a.arena_tb ? a.arena_tb + base + static_cast<unsigned long long>(excl) + d.n_prefix : nullptr;
    bool broken = false, handed = false;
    int n = 0;
    bool at_end = false;      // the ids sit at the end of the slot
    unsigned long long c2 = c1;
    if (MODEL == 1) {
      // ---- segment, then backtrack: the slot is filled from its end (or from its start when reversing) ----
      tc.n_trips += static_cast<unsigned long long>(
          unigram_stream_lane<RING, !FAST>(d, gt, gb, my_nlen, my_rs, my_rb, rm, my_win, W - 1u, my_st, T.roottab, mine));
      c2 = wv::clock();
      if (!overflow) {
        n = emit_stream_lane(d, gt, gb, my_nlen, slot, tslot, cap, mine);
        at_end = (d.flags & kNfReverse) == 0;
      }
    } else {
      // ---- word by word, ids written as the words complete: the slot is filled from its start (end when reversing) ----
      n = bpe_stream_lane(d, gt, my_nlen, slot, tslot, cap, T.bw, T.asym, T.bwin + static_cast<uint32_t>(lane) * (kBpeWindow + 4u),
                          kBpeWindow - 1u, lane, mine && !overflow,
                          a.bpe_long ? a.bpe_long + (static_cast<uint64_t>(wave_id) * 64u + static_cast<uint32_t>(lane)) * kBpeLongBytes : nullptr);
      c2 = wv::clock();
      at_end = (d.flags & kNfReverse) != 0;
      handed = n == -2;
      if (wv::any(n == -4)) {                     // a word of more than kBpeLongMax characters: OUT_OF_RANGE for the call
        if (lane == 0) wv::atomic_or(a.status, kStTooLong);
        if (n == -4) { n = 0; mine = false; a.counts[my_sid] = 0; a.tmp_off[my_sid] = 0; }
      }
      const uint64_t wm = wv::ballot(handed);
      if (wm) {                                 // words too long for the lane form: the sentence-per-wave kernel takes it
        const int leader = wv::ffs64(wm) - 1;
        uint32_t wb = 0;
        if (lane == leader) wb = wv::atomic_add(a.wave_count, static_cast<uint32_t>(wv::popc64(wm)));
        wb = wv::shfl(wb, leader);
        if (handed) a.wave_list[wb + static_cast<uint32_t>(wv::popc64(wm & ((1ull << lane) - 1ull)))] = my_sid;
      }
      if (handed) { mine = false; n = 0; }
    }
    broken = n < 0;
    if (broken) n = 0;
    if (mine && !broken && !overflow) {
      int32_t *ids = at_end ? slot + (cap - n) : slot;
      for (int x = 0; x < d.n_prefix; ++x) ids[x - d.n_prefix] = d.prefix_ids[x];
      for (int x = 0; x < d.n_suffix; ++x) ids[n + x] = d.suffix_ids[x];
      a.tmp_off[my_sid] = static_cast<unsigned long long>(ids - d.n_prefix - a.arena);
    }
    if (mine) {
      a.counts[my_sid] = (broken || overflow) ? 0u : static_cast<uint32_t>(n + n_extra);
      if (broken || overflow) a.tmp_off[my_sid] = 0;
    }
    if (wv::any(broken) && lane == 0) wv::atomic_or(a.status, kStInternal);
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
    }
  }
}

}  // namespace spmx
#endif
