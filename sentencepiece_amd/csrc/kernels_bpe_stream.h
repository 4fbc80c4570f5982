// BPE segmentation, streaming form: one SENTENCE PER LANE, one WORD at a time.
// Reference: bpe::Model::SampleEncode with alpha = 0 (src/bpe_model.cc:38-203).
//
// When no piece has the space symbol after its first character (dev.h
// kNfBpeWordwise; the default split_by_whitespace training guarantees it), a
// pair whose right symbol starts a word -- i.e. starts with U+2581 -- can never
// be in the vocabulary (:88-94), so the agenda never joins two words and the
// merges of different words do not interact: the reference's global
// best-first order restricted to one word is that word's own best-first
// order.  A lane therefore segments its sentence word by word with a working
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
// A sentence with a word of more than kBpeWordMax characters is handed to the
// sentence-per-wave kernel (kernels_bpe.h) through a device-side list; so is
// everything when pieces are UNUSED (resegmentation, :175-200) or the model is
// not word-wise.
#ifndef SPMX_KERNELS_BPE_STREAM_H_
#define SPMX_KERNELS_BPE_STREAM_H_

namespace spmx {

constexpr int kBpeWordMax = 24;

// Byte pos of a lane's text column (kernels_stream.h: dwords text[pos >> 2][lane]).
SPMX_DEVICE uint32_t stream_text_byte(const uint32_t *gt, int pos) {
  return (gt[(pos >> 2) * 64] >> (8 * (pos & 3))) & 0xFFu;
}

struct BpeWordLds {
  uint32_t *sym;     // [kBpeWordMax][64] symbol ids (kSsUnknown: a character without a symbol)
  U2 *pair;          // [kBpeWordMax][64] {merged symbol | kSymNone, score bits} of (k, k + 1)
  uint8_t *len;      // [kBpeWordMax][64] byte length of symbol k
};
SPMX_HD inline uint32_t BpeWordLdsBytes() { return kBpeWordMax * 64u * (4u + 8u + 1u); }

// Byte `pos` of the lane's text column through a two-dword register window [4 q, 4 q + 8).
struct TextCursor {
  const uint32_t *gt;
  int q;
  uint32_t cur, nxt;
};
SPMX_DEVICE void cursor_init(TextCursor *t, const uint32_t *gt) {
  t->gt = gt; t->q = 0; t->cur = gt[0]; t->nxt = gt[64];
}
SPMX_DEVICE void cursor_seek(TextCursor *t, int pos) {            // pos never moves back
  while (pos >= 4 * t->q + 4) { t->cur = t->nxt; ++t->q; t->nxt = t->gt[(t->q + 1) * 64]; }
}
SPMX_DEVICE uint32_t cursor_byte(const TextCursor &t, int pos) {  // 4 q <= pos < 4 q + 8
  const int i = pos - 4 * t.q;
  return ((i < 4 ? t.cur : t.nxt) >> (8 * (i & 3))) & 0xFFu;
}

// Segments this lane's sentence (text column gt, nlen bytes) and writes its ids into slot[0, cap): forward order
// fills the slot from its START, `reverse` from its end.  Returns the number of ids, -1 on an error status
// (control piece, overflow), -2 if the sentence has to go to the sentence-per-wave kernel (a word too long).
SPMX_DEVICE int bpe_stream_lane(const SpmxDev &d, const uint32_t *gt, int nlen, int32_t *slot, int cap,
                                const BpeWordLds &B, const uint32_t *asym, int lane, bool active_in) {
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  uint32_t *sym = B.sym + lane;
  U2 *pr = B.pair + lane;
  uint8_t *ln = B.len + lane;
  bool active = active_in && nlen > 0;
  int pos = 0;          // next byte of the text to read
  int n = 0;            // symbols of the current word
  int wstart = 0;       // byte offset of the current word
  bool merging = false; // the current word has been read completely
  bool right_unk = false;
  int n_out = 0, ret = 0;
  TextCursor tc{gt, 0, 0, 0};
  if (active) cursor_init(&tc, gt);
  while (wv::any(active)) {
    if (!active) continue;
    if (!merging) {
      // ---- read one character of the current word (:109-129) ----
      cursor_seek(&tc, pos);
      const uint32_t c0 = cursor_byte(tc, pos);
      if (n > 0 && c0 == spb) { merging = true; continue; }                 // the next word begins here
      if (n == kBpeWordMax) { ret = -2; active = false; continue; }          // too long for the LDS working set
      int mb = c0 == spb ? 1 : OneCharLenDev(c0);
      if (mb > nlen - pos) mb = nlen - pos;
      uint32_t s;
      if (mb == 1) {
        s = asym[c0];
      } else {
        uint32_t bytes = c0;
        for (int k = 1; k < mb; ++k) bytes |= cursor_byte(tc, pos + k) << (8 * k);
        s = char_lookup(d, bytes, static_cast<uint32_t>(mb));
      }
      if (n == 0) wstart = pos;
      sym[n * 64] = s;
      ln[n * 64] = static_cast<uint8_t>(mb);
      pr[n * 64] = U2{kSymNone, 0};
      if (n > 0) {
        uint32_t merged = 0;
        float sc = 0.f;
        if (pair_lookup(d, sym[(n - 1) * 64], s, &merged, &sc)) pr[(n - 1) * 64] = U2{merged, __builtin_bit_cast(uint32_t, sc)};
      }
      ++n;
      pos += mb;
      if (pos >= nlen) merging = true;
      continue;
    }
    // ---- one merge (:142-173): best live pair, highest score first, then leftmost ----
    int best = -1;
    float bs = 0.f;
    uint32_t bm = 0;
    for (int k = 0; k + 1 < n; ++k) {
      const U2 e = pr[k * 64];
      if (e.x != kSymNone) {
        const float sc = wv::bits_to_float(e.y);
        if (best < 0 || sc > bs) { best = k; bs = sc; bm = e.x; }
      }
    }
    if (best >= 0) {
      sym[best * 64] = bm;
      ln[best * 64] = static_cast<uint8_t>(ln[best * 64] + ln[(best + 1) * 64]);
      for (int k = best + 1; k + 1 < n; ++k) {                               // close the gap
        sym[k * 64] = sym[(k + 1) * 64];
        ln[k * 64] = ln[(k + 1) * 64];
        pr[k * 64] = pr[(k + 1) * 64];
      }
      --n;
      // :171-172 the two new neighbours
      uint32_t merged = 0;
      float sc = 0.f;
      if (best > 0) {
        if (pair_lookup(d, sym[(best - 1) * 64], bm, &merged, &sc)) pr[(best - 1) * 64] = U2{merged, __builtin_bit_cast(uint32_t, sc)};
        else pr[(best - 1) * 64] = U2{kSymNone, 0};
      }
      if (best + 1 < n) {
        if (pair_lookup(d, bm, sym[(best + 1) * 64], &merged, &sc)) pr[best * 64] = U2{merged, __builtin_bit_cast(uint32_t, sc)};
        else pr[best * 64] = U2{kSymNone, 0};
      } else {
        pr[best * 64] = U2{kSymNone, 0};
      }
      continue;
    }
    // ---- no pair left: the word's symbols are its pieces (:175-200, no UNUSED pieces here) ----
    int off = wstart;
    for (int k = 0; k < n && ret == 0; ++k) {
      const uint32_t s = sym[k * 64];
      const int len = ln[k * 64];
      uint32_t f = static_cast<uint32_t>(d.unk_id);
      if (s != kSsUnknown) {
        f = d.sym_final[s];
        if (f & kSfControl) { ret = -1; break; }
        f &= kSfIdMask;
      }
      if (static_cast<int32_t>(f) == d.unk_id) {
        if (bf) {                                   // one BYTE id per byte of the unknown piece (:581-603)
          for (int x = 0; x < len; ++x) {
            const uint32_t b = stream_text_byte(gt, off + x);
            const int nb = b == spb ? 3 : 1;
            if (n_out + nb > cap) { ret = -1; break; }
            for (int y = 0; y < nb; ++y) {
              const uint32_t byte = b == spb ? (y == 0 ? 0xE2u : (y == 1 ? 0x96u : 0x81u)) : b;
              slot[reverse ? cap - 1 - n_out : n_out] = d.byte_ids[byte];
              ++n_out;
            }
          }
        } else if (!right_unk) {                    // a run of unknown pieces yields one id (:609-613)
          if (n_out >= cap) { ret = -1; break; }
          slot[reverse ? cap - 1 - n_out : n_out] = d.unk_id;
          ++n_out;
        }
        right_unk = true;
      } else {
        if (n_out >= cap) { ret = -1; break; }
        slot[reverse ? cap - 1 - n_out : n_out] = static_cast<int32_t>(f);
        ++n_out;
        right_unk = false;
      }
      off += len;
    }
    n = 0;
    merging = false;
    if (ret != 0 || pos >= nlen) active = false;
  }
  return ret != 0 ? ret : n_out;
}

}  // namespace spmx
#endif
