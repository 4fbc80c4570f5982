// Batch Decode: token ids -> text, one sentence per wavefront, 64 pieces per sweep.
// Reference: SentencePieceProcessor::Decode(const std::vector<int>& ids, std::string*)
// (src/sentencepiece_processor.cc:761-925): IdToPiece, then per piece
//   control piece   -> nothing (:780-781)
//   unknown piece   -> unk_surface (:782-788)
//   byte pieces     -> runs are reassembled into UTF-8 characters; a structurally invalid byte becomes
//                      U+FFFD (ProcessBytePieces, :825-880)
//   anything else   -> the piece with U+2581 -> ' ' (:807), minus its leading U+2581 while the text is still
//                      empty and the model adds a dummy prefix / removes extra whitespace (:791-805)
// The per-id work is folded into tables at load (tables.cc: decoded bytes, kind, "starts with U+2581", byte
// value), so the kernel is a gather: for every piece of a sweep its output length, an exclusive scan, and a
// coalesced copy in which every output byte finds its piece by a binary search over the 64 lane offsets.
//
// Sequential state of the reference, restated for a sweep:
//   is_bos_ws (:884-899)  true until a piece has produced text or consumed a leading U+2581 (the latter only
//                         without remove_extra_whitespaces): a ballot of the lanes that would end it;
//   byte runs             characters of 1-4 byte pieces chain from the start of the run: resolve_chain
//                         (kernels.h) over "byte pieces step by their character length, other pieces by 1";
//                         a character that would straddle the end of the sweep opens the next sweep instead.
// The kernel runs twice per call (count, then write) with a scan of the per-sentence lengths in between.
#ifndef SPMX_KERNELS_DECODE_H_
#define SPMX_KERNELS_DECODE_H_

namespace spmx {

// dec_info word per id
constexpr uint32_t kDkMask = 3u;
constexpr uint32_t kDkText = 0u;     // decoded bytes dec_bytes[dec_off[id], dec_off[id + 1]) (U+2581 already -> ' ')
constexpr uint32_t kDkEmpty = 1u;    // control piece
constexpr uint32_t kDkByte = 2u;     // byte piece, value in bits 8..15
constexpr uint32_t kDkLiteral = 3u;  // unknown piece: dec_bytes verbatim, never stripped
constexpr uint32_t kDiStartsSp = 1u << 2;   // kDkText: the piece starts with U+2581 (its decoded form with ' ')
constexpr int kDiLenShift = 16;             // bits 16..31: number of decoded bytes (so the count pass needs this word only)

constexpr uint32_t kStBadId = 1u << 4;   // DecodeArgs::status: an id outside [0, GetPieceSize())

struct DecodeArgs {
  SpmxDev dev;
  const int32_t *ids;           // CSR ids
  const uint64_t *id_offs;      // n + 1
  uint32_t n;
  uint32_t *counts;             // per sentence: bytes of text (written by the count pass)
  const uint64_t *text_offs;    // n + 1 (write pass)
  uint8_t *text;
  uint64_t text_cap;
  uint32_t *status;
  unsigned long long *bad_key;  // min over offending (sentence << 32 | id as uint32)
  // SetDecodeExtraOptions (src/sentencepiece_processor.cc:288-291): ApplyExtraOptions(decode_extra_options_) on the piece
  // list before it is decoded (:819, :1019-1064) -- its net effect on the ids: bos / eos ids in front / behind, the body
  // reversed (tables.cc CompileExtraOptions; "unk" only rewrites piece strings of ids that decode to unk_surface anyway)
  int32_t x_npre, x_nsuf, x_reverse;
  int32_t x_pre[kMaxExtra], x_suf[kMaxExtra];
  // Decode(pieces) (src/sentencepiece_processor.cc:761-769, :784-790): a piece that is not in the vocabulary has the id of
  // the unknown piece and another string -- it goes through as it is.  Such a piece travels as the id -(k + 1), its bytes
  // lit_bytes[lit_offs[k], lit_offs[k + 1]) (spmx_decode_batch_pieces; null: every negative id is an error)
  const uint8_t *lit_bytes;
  const uint32_t *lit_offs;
  uint32_t n_lit;
};

template <bool WRITE>
SPMX_DEVICE void decode_block(const DecodeArgs &a) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  const bool rm = (d.flags & kNfRemoveExtraWs) != 0;
  const bool strip = rm || (d.flags & kNfAddDummyPrefix) != 0;
  if (WRITE && a.text_offs[a.n] > a.text_cap) return;      // caller sees the needed size in text_offs[n]
  for (uint32_t s = static_cast<uint32_t>(wv::block_id()); s < a.n; s += static_cast<uint32_t>(wv::grid_size())) {
    const uint64_t beg = a.id_offs[s];
    const int n_body = static_cast<int>(a.id_offs[s + 1] - beg);
    const int n_pieces = n_body + a.x_npre + a.x_nsuf;
    uint8_t *dst = WRITE ? a.text + a.text_offs[s] : nullptr;
    uint32_t out = 0;            // bytes of this sentence so far
    bool bos = true;             // is_bos_ws
    bool bad = false;
    int base = 0;
    int next_start = 0;
    while (base < n_pieces) {
      const int i = base + lane;
      const bool valid = i < n_pieces;
      int32_t id = 0;
      uint32_t info = kDkEmpty, off = 0, full = 0, lit = 0;
      if (valid) {
        if (i < a.x_npre) id = a.x_pre[i];
        else if (i < a.x_npre + n_body) {
          const int k = i - a.x_npre;
          id = a.ids[beg + static_cast<uint64_t>(a.x_reverse ? n_body - 1 - k : k)];
        } else id = a.x_suf[i - a.x_npre - n_body];
        if (id < 0 && a.lit_offs != nullptr && static_cast<uint32_t>(-(id + 1)) < a.n_lit) {
          const uint32_t k = static_cast<uint32_t>(-(id + 1));
          off = a.lit_offs[k];
          full = a.lit_offs[k + 1] - off;
          info = kDkLiteral | (full << kDiLenShift);
          lit = 1u;
        } else if (id < 0 || static_cast<uint32_t>(id) >= d.n_pieces) {
          bad = true;
          wv::atomic_min(a.bad_key, (static_cast<unsigned long long>(s) << 32) | static_cast<uint32_t>(id));
        } else {
          info = d.dec_info[id];
          full = info >> kDiLenShift;
          if (WRITE) off = d.dec_off[id];
        }
      }
      const bool is_byte = valid && (info & kDkMask) == kDkByte;
      const uint32_t bv = (info >> 8) & 0xFFu;
      // ---- byte runs: speculative character length at every byte piece (IsValidDecodeUTF8, util.h:173-176) ----
      const uint64_t bm = wv::ballot(is_byte);
      int step = 1;
      bool ch_ok = true;
      {
        // bytes of the run that follow this lane (0 past the run or the sweep)
        const uint32_t b1 = wv::shfl(bv, lane + 1), b2 = wv::shfl(bv, lane + 2), b3 = wv::shfl(bv, lane + 3);
        const bool h1 = lane + 1 < 64 && ((bm >> (lane + 1)) & 1ull), h2 = h1 && lane + 2 < 64 && ((bm >> (lane + 2)) & 1ull),
                   h3 = h2 && lane + 3 < 64 && ((bm >> (lane + 3)) & 1ull);
        if (is_byte) {
          const uint32_t b0 = bv;
          int mb = 1;
          bool ok = b0 < 0x80u;
          if (!ok) {
            const bool t1 = h1 && (b1 & 0xC0u) == 0x80u, t2 = h2 && (b2 & 0xC0u) == 0x80u, t3 = h3 && (b3 & 0xC0u) == 0x80u;
            if ((b0 & 0xE0u) == 0xC0u) {
              const uint32_t cp = (b0 & 0x1Fu) << 6 | (b1 & 0x3Fu);
              if (t1 && cp >= 0x80u) { ok = true; mb = 2; }
            } else if ((b0 & 0xF0u) == 0xE0u) {
              const uint32_t cp = (b0 & 0x0Fu) << 12 | (b1 & 0x3Fu) << 6 | (b2 & 0x3Fu);
              if (t1 && t2 && cp >= 0x800u && (cp < 0xD800u || cp >= 0xE000u)) { ok = true; mb = 3; }
            } else if ((b0 & 0xF8u) == 0xF0u) {
              const uint32_t cp = (b0 & 0x07u) << 18 | (b1 & 0x3Fu) << 12 | (b2 & 0x3Fu) << 6 | (b3 & 0x3Fu);
              if (t1 && t2 && t3 && cp >= 0x10000u && cp <= 0x10FFFFu) { ok = true; mb = 4; }
            }
          }
          step = mb;
          ch_ok = ok;
        }
      }
      // ---- which pieces start an output unit (a non-byte piece, or the first byte piece of a character) ----
      if (next_start < base) next_start = base;
      int ns = next_start;
      const uint64_t S = resolve_chain(base, step, valid, &ns);
      // a character that may need byte pieces beyond this sweep opens the next one
      int proc_end = base + 64 < n_pieces ? base + 64 : n_pieces;
      if (base + 64 < n_pieces) {
        uint64_t tail = 0;                       // byte pieces that reach the end of the sweep, last three positions
        if ((bm >> 63) & 1ull) { tail = 1ull << 63; if ((bm >> 62) & 1ull) { tail |= 1ull << 62; if ((bm >> 61) & 1ull) tail |= 1ull << 61; } }
        const uint64_t defer = S & tail;
        if (defer) proc_end = base + wv::ffs64(defer) - 1;
      }
      const bool live = valid && i < proc_end && ((S >> lane) & 1ull);
      // ---- is_bos_ws: ends at the first piece that produces text or consumes a leading U+2581 (:791-805, :893) ----
      const bool starts_sp = (info & kDkMask) == kDkText && (info & kDiStartsSp);
      const uint32_t len_if_bos = is_byte ? (ch_ok ? static_cast<uint32_t>(step) : 3u)
                                          : (full - ((strip && starts_sp) ? 1u : 0u));
      const bool ends_bos = live && (len_if_bos > 0 || (strip && starts_sp && !rm));
      const uint64_t em = wv::ballot(ends_bos);
      const bool my_bos = bos && (em & ((1ull << lane) - 1ull)) == 0;
      if (em) bos = false;
      // ---- output length of every live piece ----
      uint32_t olen = 0, skip = 0;
      if (live) {
        if (is_byte) olen = ch_ok ? static_cast<uint32_t>(step) : 3u;
        else { skip = (my_bos && strip && starts_sp) ? 1u : 0u; olen = full - skip; }
      }
      int total = 0;
      const uint32_t rel = static_cast<uint32_t>(wave_excl_scan(static_cast<int>(olen), lane, &total));
      if (WRITE && total > 0) {
        // every output byte of the sweep finds its piece: the last lane whose offset is <= j (lanes without
        // output share the offset of the next one and lose to it).  All cross-lane reads sit in uniform flow.
        const uint32_t src = off + skip;
        const uint32_t okc = ch_ok ? 1u : 0u;
        for (uint32_t r = 0; r * 64u < static_cast<uint32_t>(total); ++r) {
          const uint32_t j = r * 64u + static_cast<uint32_t>(lane);
          int lo = 0;
#pragma unroll
          for (int st = 32; st >= 1; st >>= 1) {
            const uint32_t v = wv::shfl(rel, lo + st);
            if (v <= j) lo += st;
          }
          const uint32_t r0 = wv::shfl(rel, lo), s0 = wv::shfl(src, lo), inf = wv::shfl(info, lo), ok0 = wv::shfl(okc, lo), lit0 = wv::shfl(lit, lo);
          const uint32_t k = j - r0;
          // a character of byte pieces: byte k of the character is the value of piece lo + k
          const uint32_t pv = wv::shfl(bv, (lo + static_cast<int>(k & 3u)) & 63);
          if (j < static_cast<uint32_t>(total)) {
            uint32_t byte;
            if ((inf & kDkMask) == kDkByte) byte = ok0 ? pv : (k == 0 ? 0xEFu : (k == 1 ? 0xBFu : 0xBDu));   // invalid -> U+FFFD
            else byte = lit0 ? a.lit_bytes[s0 + k] : d.dec_bytes[s0 + k];
            dst[out + j] = static_cast<uint8_t>(byte);
          }
        }
      }
      out += static_cast<uint32_t>(total);
      next_start = proc_end;     // no unit straddles sweeps, so the next sweep starts on a unit
      base = proc_end;
    }
    if (wv::any(bad)) { if (lane == 0) wv::atomic_or(a.status, kStBadId); out = 0; }   // the host stops before the write pass
    if (!WRITE && lane == 0) a.counts[s] = out;
  }
}

}  // namespace spmx
#endif
