// BPE segmentation, streaming form: one SENTENCE PER LANE, one WORD at a time.
// Reference: bpe::Model::SampleEncode with alpha = 0 (src/bpe_model.cc:38-203).
//
// When every piece is either a run of space symbols or has none after its
// first character (dev.h kNfBpeWordwise; the default split_by_whitespace
// training guarantees it, with or without allow_whitespace_only_pieces), a pair
// (x, U+2581...) whose left symbol ends in another character can never be in
// the vocabulary (:88-94).  With a WORD = a run of U+2581 and the characters
// that follow it up to the next U+2581, the agenda therefore never joins two
// words and the merges of different words do not interact: the reference's
// global best-first order restricted to one word is that word's own
// best-first order.  A lane therefore segments its sentence word by word with a working
// set of kBpeWordMax symbols in LDS, whatever the sentence length:
//
//   read      one character per iteration: symbol id from an LDS table (ASCII)
//             or the char table, and the pair (previous, this) from the pair
//             table (:127-129) -- until the next word begins;
//   merge     one merge per iteration: argmax over the word's live pairs
//             (highest score, then leftmost, :53-56), replace, close the gap,
//             look up the two new neighbours (:159-172);
//   output    the word's symbols -> ids (PieceToId, :178), with the
//             unknown-run merge / byte fallback of
//             sentencepiece_processor.cc:581-613 carried across words.
//
// A sentence with a word of more than kBpeWordMax characters (a URL, a hash) is
// handed to the long form (kernels_long.h) through a device-side list; models
// that are not word-wise, or with UNUSED pieces (resegmentation, :175-200), do
// not come here at all.
#ifndef SPMX_KERNELS_BPE_STREAM_H_
#define SPMX_KERNELS_BPE_STREAM_H_

namespace spmx {

constexpr int kBpeWordMax = 24;
constexpr uint32_t kBpeWindow = 32;   // bytes of text a lane keeps in LDS

struct BpeWordLds {
  uint32_t *sym;     // [kBpeWordMax][64] symbol at character slot k (kSsUnknown: a character without a symbol)
  float *score;      // [kBpeWordMax][64] score of the live pair whose left symbol sits at slot k
  uint32_t *merged;  // [kBpeWordMax][64] the symbol that pair merges into
  uint8_t *len;      // [kBpeWordMax][64] byte length of the symbol at slot k
};
SPMX_HD inline uint32_t BpeWordLdsBytes() { return kBpeWordMax * 64u * (4u + 4u + 4u + 1u); }

// pieces_.find(left + right) (src/bpe_model.cc:88-94) split in two so that several probes are in flight at once:
// issue computes the slot and starts the load, resolve consumes it (and walks on after a hash collision).
struct PairProbe {
  uint32_t a = 0, b = 0, slot = 0;
  int idx = 0;        // character slot of the left symbol
  bool live = false;
  U4 e{0, 0, 0, 0};
};
// request: remember the pair; the load itself is issued by pair_issue at the end of the iteration, for both probe
// objects unconditionally (a dead probe reads entry 0), so that the memory pipeline is the same on every path and
// the wait at the top of the next iteration is the only one.
SPMX_DEVICE void pair_request(const SpmxDev &d, PairProbe *p, uint32_t a, uint32_t b, int idx) {
  p->live = a < kSsUnknown && b < kSsUnknown;       // an unknown character (or frozen symbol) never merges
  p->a = a; p->b = b; p->idx = idx;
  p->slot = p->live ? (HashPair(a, b) & d.pairtab_mask) : 0u;
}
SPMX_DEVICE void pair_issue(const SpmxDev &d, PairProbe *p) { p->e = d.pairtab[p->slot]; }
// Returns true (and the merged symbol / score) if the pair is a piece.
SPMX_DEVICE bool pair_resolve(const SpmxDev &d, PairProbe *p, uint32_t *merged, float *score) {
  U4 e = p->e;
  bool hit = (e.x == p->a) & (e.y == p->b), none = e.x == kSymNone;   // consumes the load on every path
  const bool live = p->live;
  uint32_t slot = p->slot;
  p->live = false;
  p->slot = 0;
  if (!live) return false;
  while (!hit && !none) {                           // hash collision: walk on
    slot = (slot + 1) & d.pairtab_mask;
    e = d.pairtab[slot];
    hit = (e.x == p->a) & (e.y == p->b);
    none = e.x == kSymNone;
  }
  if (!hit) return false;
  *merged = e.z;
  *score = wv::bits_to_float(e.w);
  return true;
}

// ---- the word table (dev.h wordtab): a whole word's pieces by ONE probe ---------------------------------------------
// At the start of a word the lane takes the next 17 bytes of its text window, finds the word's end (the first space
// symbol after the word's own leading ones, or the end of the text), and looks the word up; the probe is in flight
// like the pair probes and lands at the top of the next iteration.  A hit writes the ids and moves on to the next
// word: no characters are read, no merges run.  A miss (a word that is no vocabulary string, or longer than 16 bytes)
// takes the merge loop below.
struct WordProbe {
  uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0, len = 0, slot = 0;
  bool live = false;
  U4 e0{0, 0, 0, 0}, e1{0, 0, 0, 0};
};
// bit b of the result: byte b of x is kSpByte (0xFF never occurs in UTF-8 otherwise)
SPMX_DEVICE uint32_t sp_mask4(uint32_t x) {
  const uint32_t m = x & 0x80808080u & ((x & 0x7F7F7F7Fu) + 0x01010101u);
  return (((m >> 7) * 0x00204081u) >> 21) & 0xFu;
}
SPMX_DEVICE uint32_t keep_bytes(uint32_t x, int keep) {      // the low `keep` bytes of x (keep may be <= 0 or >= 4)
  return keep >= 4 ? x : (keep <= 0 ? 0u : x & ((1u << (8 * keep)) - 1u));
}
SPMX_DEVICE void word_issue(const SpmxDev &d, WordProbe *p) {
  p->e0 = d.wordtab[2 * p->slot];
  p->e1 = d.wordtab[2 * p->slot + 1];
}

// Segments this lane's sentence (text column gt, nlen bytes) and writes its ids into slot[0, cap): forward order
// fills the slot from its START, `reverse` from its end.  Returns the number of ids, -1 on an error status
// (control piece, overflow), -2 if the sentence has a word of more than kBpeWordMax characters: it goes to the long
// form (kernels_long.h), which has no such bound.
//
// The word lives in character slots 0 .. n0-1 of the LDS arrays; `alive` has a bit per slot that still starts a
// symbol, `pmask` a bit per slot whose pair with the next live slot is a piece.  A merge clears the right slot's
// bit: nothing is moved.  Every iteration first lands the (at most two) pair probes issued by the previous one.
//
// Text comes through a W-byte LDS window (position p at win[p & wmask]) refilled one dword per iteration from the
// lane's text column; that load and the pair probes are issued at the END of an iteration and land at the top of
// the next one, so an iteration waits for memory once.
SPMX_DEVICE int bpe_stream_lane(const SpmxDev &d, const TextCol &gt, int nlen, int32_t *slot, int32_t *tslot, int cap,
                                const BpeWordLds &B, const uint32_t *asym, uint8_t *win, uint32_t wmask, int lane,
                                bool active_in) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  uint32_t *sym = B.sym + lane;
  float *scr = B.score + lane;
  uint32_t *mrg = B.merged + lane;
  uint8_t *ln = B.len + lane;
  bool active = active_in && nlen > 0;
  int pos = 0;          // next byte of the text to read
  int n0 = 0;           // character slots of the current word
  int wstart = 0;       // byte offset of the current word
  uint32_t alive = 0, pmask = 0;
  bool merging = false; // the current word has been read completely
  bool prev_sp = false; // the last character read was U+2581
  bool right_unk = false;
  int n_out = 0, ret = 0;
  auto slot_out = [&](uint32_t id, int off) __attribute__((always_inline)) {
    slot[reverse ? cap - 1 - n_out : n_out] = static_cast<int32_t>(id);
    if (tslot) tslot[reverse ? cap - 1 - n_out : n_out] = off;
    ++n_out;
  };
  PairProbe p0, p1;
  WordProbe wq;
  const bool use_words = d.wordtab_mask != 0u;
  int word_tried = -1;   // the word that starts at this byte missed the table (or cannot be looked up): merge loop
  const int W = static_cast<int>(wmask) + 1;
  int nf = 0;                                     // next text dword to fetch; the window holds dwords [nf - W/4, nf)
  bool pf_pend = false;
  uint32_t pf = 0, pf2 = 0;
  int pf_n = 0;                                   // text dwords in flight (the word table consumes a word per iteration)
  if (active) {
    for (int k = 0; k < W / 4; ++k) *reinterpret_cast<uint32_t *>(win + 4 * k) = gt.dw(k);
    nf = W / 4;
  }
  while (wv::any(active)) {
    if (!active) continue;
    {   // land what the previous iteration issued: a text dword and up to two pair probes
      if (pf_pend) *reinterpret_cast<uint32_t *>(win + ((4u * static_cast<uint32_t>(nf - pf_n)) & wmask)) = pf;
      if (pf_n == 2) *reinterpret_cast<uint32_t *>(win + ((4u * static_cast<uint32_t>(nf - 1)) & wmask)) = pf2;
      pf_pend = false;
      pf_n = 0;
      uint32_t m = 0;
      float sc = 0.f;
      if (pair_resolve(d, &p0, &m, &sc)) { mrg[p0.idx * 64] = m; scr[p0.idx * 64] = sc; pmask |= 1u << p0.idx; }
      if (pair_resolve(d, &p1, &m, &sc)) { mrg[p1.idx * 64] = m; scr[p1.idx * 64] = sc; pmask |= 1u << p1.idx; }
      if (use_words) {
        // the word probe: consumes the load on every path; walks on after a collision
        U4 e0 = wq.e0, e1 = wq.e1;
        const bool live = wq.live;
        uint32_t slot = wq.slot;
        wq.live = false;
        wq.slot = 0;
        if (live) {
          bool hit = e0.x == wq.k0 && e0.y == wq.k1 && e0.z == wq.k2 && e0.w == wq.k3 && (e1.x & 0xFFu) == wq.len;
          while (!hit && (e1.x & 0xFFu) != 0u) {
            slot = (slot + 1) & d.wordtab_mask;
            e0 = d.wordtab[2 * slot];
            e1 = d.wordtab[2 * slot + 1];
            hit = e0.x == wq.k0 && e0.y == wq.k1 && e0.z == wq.k2 && e0.w == wq.k3 && (e1.x & 0xFFu) == wq.len;
          }
          if (!hit) {
            word_tried = pos;
          } else {                                    // the word's pieces (never an unknown or a control piece)
            const int nid = static_cast<int>((e1.x >> 8) & 0xFFu);
            if (n_out + nid > cap) { ret = -1; active = false; }
            else {
              int off = pos;
              for (int k = 0; k < nid; ++k) {
                const uint32_t id = k == 0 ? e1.y : (k == 1 ? e1.z : e1.w);
                slot_out(id, off);
                off += static_cast<int>((e1.x >> (16 + 5 * k)) & 31u);
              }
              right_unk = false;
              pos += static_cast<int>(wq.len);
              prev_sp = false;
              if (pos >= nlen) active = false;
            }
          }
        }
      }
    }
    bool looking = false;                      // a word probe goes out this iteration: no character is read
    if (use_words && active && !merging && n0 == 0 && word_tried != pos) {
      if ((pos & ~3) + 20 > 4 * nf && 4 * nf < nlen) {
        looking = true;                          // its bytes have not landed yet
      } else {
        // bytes [pos, pos + 17) of the text through the window
        const uint32_t a0 = static_cast<uint32_t>(pos) & ~3u;
        const uint32_t d0 = *reinterpret_cast<const uint32_t *>(win + (a0 & wmask));
        const uint32_t d1 = *reinterpret_cast<const uint32_t *>(win + ((a0 + 4u) & wmask));
        const uint32_t d2 = *reinterpret_cast<const uint32_t *>(win + ((a0 + 8u) & wmask));
        const uint32_t d3 = *reinterpret_cast<const uint32_t *>(win + ((a0 + 12u) & wmask));
        const uint32_t d4 = *reinterpret_cast<const uint32_t *>(win + ((a0 + 16u) & wmask));
        const uint32_t sh = 8u * (static_cast<uint32_t>(pos) & 3u);
        auto funnel = [&](uint32_t hi, uint32_t lo) __attribute__((always_inline)) -> uint32_t {
          return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        };
        const uint32_t k0 = funnel(d1, d0), k1 = funnel(d2, d1), k2 = funnel(d3, d2), k3 = funnel(d4, d3);
        const uint32_t b16 = (d4 >> sh) & 0xFFu;
        const uint32_t ff = sp_mask4(k0) | sp_mask4(k1) << 4 | sp_mask4(k2) << 8 | sp_mask4(k3) << 12 | (b16 == kSpByte ? 1u << 16 : 0u);
        const int lead = wv::ffs64(static_cast<uint64_t>(~ff)) - 1;                 // the word's own leading space symbols
        const uint32_t rest = ff & ~((1u << lead) - 1u);
        int L = rest ? wv::ffs64(static_cast<uint64_t>(rest)) - 1 : 17;
        if (L > nlen - pos) L = nlen - pos;
        if (L > static_cast<int>(kWordKeyBytes) || L <= lead) {
          word_tried = pos;                      // too long for the table / nothing but space symbols: merge loop
        } else {
          wq.k0 = keep_bytes(k0, L); wq.k1 = keep_bytes(k1, L - 4); wq.k2 = keep_bytes(k2, L - 8); wq.k3 = keep_bytes(k3, L - 12);
          wq.len = static_cast<uint32_t>(L);
          wq.slot = HashWord(wq.k0, wq.k1, wq.k2, wq.k3) & d.wordtab_mask;
          wq.live = true;
          looking = true;
        }
      }
    }
    bool do_merge = merging;
    if (!merging && !looking) {
      // ---- read up to two characters of the current word (:109-129) ----
      // (two explicit calls: the probe objects must stay in registers, so no run-time choice between them)
      auto read_char = [&](PairProbe *pp) {
        if (merging || ret != 0) return;
        if (pos + 4 > 4 * nf && 4 * nf < nlen) return;                       // its bytes have not landed yet
        const uint32_t c0 = win[static_cast<uint32_t>(pos) & wmask];
        if (n0 > 0 && c0 == spb && !prev_sp) { merging = true; return; }     // a U+2581 after another character: the next word
        if (n0 == kBpeWordMax) { ret = -2; return; }                         // too long for the LDS working set
        int mb = c0 == spb ? 1 : OneCharLenDev(c0);
        if (mb > nlen - pos) mb = nlen - pos;
        uint32_t s;
        if (mb == 1) {
          s = asym[c0];
        } else {
          uint32_t bytes = c0;
          for (int k = 1; k < mb; ++k) bytes |= static_cast<uint32_t>(win[static_cast<uint32_t>(pos + k) & wmask]) << (8 * k);
          s = char_lookup(d, bytes, static_cast<uint32_t>(mb));
        }
        if (n0 == 0) wstart = pos;
        sym[n0 * 64] = s;
        ln[n0 * 64] = static_cast<uint8_t>(mb);
        if (n0 > 0) pair_request(d, pp, sym[(n0 - 1) * 64], s, n0 - 1);
        alive |= 1u << n0;
        ++n0;
        pos += mb;
        prev_sp = c0 == spb;
        if (pos >= nlen) merging = true;
      };
      read_char(&p0);
      read_char(&p1);
      if (ret != 0) active = false;
    }
    if (do_merge) {
    // ---- one merge (:142-173): best live pair, highest score first, then leftmost ----
    int best = -1;
    float bs = 0.f;
    for (int base = 0; base < kBpeWordMax; base += 8) {
      const uint32_t mm = (pmask >> base) & 0xFFu;
      if (mm == 0) continue;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = scr[(base + k) * 64];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (((mm >> k) & 1u) && (best < 0 || v[k] > bs)) { best = base + k; bs = v[k]; }
    }
    if (best >= 0) {
      const uint32_t above = alive & ~((2u << best) - 1u);                   // live slots right of `best`
      const int j = wv::ffs64(above) - 1;                                     // the right symbol (exists: the pair is live)
      const uint32_t bm = mrg[best * 64];
      sym[best * 64] = bm;
      ln[best * 64] = static_cast<uint8_t>(ln[best * 64] + ln[j * 64]);
      alive &= ~(1u << j);
      pmask &= ~((1u << best) | (1u << j));
      // :171-172 the two new neighbours
      const uint32_t below = alive & ((1u << best) - 1u);
      if (below) {
        const int p = 31 - static_cast<int>(wv::clz64(static_cast<uint64_t>(below)) - 32);
        pmask &= ~(1u << p);
        pair_request(d, &p0, sym[p * 64], bm, p);
      }
      const uint32_t after = alive & ~((2u << best) - 1u);
      if (after) pair_request(d, &p1, bm, sym[(wv::ffs64(after) - 1) * 64], best);
    } else {
    // ---- no pair left: the word's symbols are its pieces (:175-200, no UNUSED pieces here) ----
    int off = wstart;
    for (uint32_t m = alive; m != 0 && ret == 0; m &= m - 1) {
      const int k = wv::ffs64(m) - 1;
      const uint32_t s = sym[k * 64];
      const int len = ln[k * 64];
      // PieceToId (:178): a symbol below n_pieces is a member of pieces_ and its own id (never a control piece);
      // only the extra characters (split parts that are no pieces themselves) go through sym_final
      uint32_t f = s;
      if (s >= d.n_pieces) {
        f = static_cast<uint32_t>(d.unk_id);
        if (s != kSsUnknown) {
          f = d.sym_final[s];
          if (f & kSfControl) { ret = -1; break; }
          f &= kSfIdMask;
        }
      }
      if (static_cast<int32_t>(f) == d.unk_id) {
        if (bf) {                                   // one BYTE id per byte of the unknown piece (:581-603)
          for (int x = 0; x < len && ret == 0; ++x) {
            const uint32_t b = col_byte(gt, off + x);
            const int nb = b == spb ? 3 : 1;
            if (n_out + nb > cap) { ret = -1; break; }
            for (int y = 0; y < nb; ++y) {
              const uint32_t byte = b == spb ? (y == 0 ? 0xE2u : (y == 1 ? 0x96u : 0x81u)) : b;
              slot[reverse ? cap - 1 - n_out : n_out] = d.byte_ids[byte];
              if (tslot) tslot[reverse ? cap - 1 - n_out : n_out] = off;
              ++n_out;
            }
          }
        } else if (!right_unk) {                    // a run of unknown pieces yields one id (:609-613)
          if (n_out >= cap) { ret = -1; break; }
          slot[reverse ? cap - 1 - n_out : n_out] = d.unk_id;
          if (tslot) tslot[reverse ? cap - 1 - n_out : n_out] = off;
          ++n_out;
        }
        right_unk = true;
      } else {
        if (n_out >= cap) { ret = -1; break; }
        slot[reverse ? cap - 1 - n_out : n_out] = static_cast<int32_t>(f);
        if (tslot) tslot[reverse ? cap - 1 - n_out : n_out] = off;
        ++n_out;
        right_unk = false;
      }
      off += len;
    }
    n0 = 0;
    alive = 0;
    pmask = 0;
    merging = false;
    if (ret != 0 || pos >= nlen) active = false;
    }
    }
    // window refill: dword nf may replace positions [4 nf - W, 4 nf - W + 4), dead once they lie below pos
    if (active && 4 * nf + 4 <= pos + W && 4 * nf < nlen + 8) { pf = gt.dw(nf); ++nf; pf_pend = true; pf_n = 1; }
    if (use_words && pf_pend && 4 * nf + 4 <= pos + W && 4 * nf < nlen + 8) { pf2 = gt.dw(nf); ++nf; pf_n = 2; }
    pair_issue(d, &p0);
    pair_issue(d, &p1);
    if (use_words) word_issue(d, &wq);
  }
  return ret != 0 ? ret : n_out;
}

}  // namespace spmx
#endif
