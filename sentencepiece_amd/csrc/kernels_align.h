// Spans form of the batch encode: next to every id, the byte range [begin, end) of the
// INPUT sentence it covers -- SentencePieceText.pieces[i].begin / .end of the reference
// (PopulateSentencePieceText, src/sentencepiece_processor.cc:547-636):
//   token over normalized bytes [b, e)      -> begin = norm_to_orig[b], end = norm_to_orig[e]       (:573-583)
//   byte-fallback pieces of one character   -> all but the last are empty at norm_to_orig[b]        (:586-610)
//   a run of unknown pieces                 -> one piece from the first begin to the last end       (:612-619)
//   bos / eos of the extra options          -> empty at 0 / at the end of the input                 (:1029-1048)
// Tokens tile the normalized text, so e of a token is b of the next one (the end of the text for the last): the
// encode kernels record b only (EncodeArgs::arena_tb; the byte pieces of a character share theirs), and this
// kernel -- a second, cold pass, one sentence per wavefront -- runs the position-parallel normalizer again with its
// norm_to_orig output switched on and translates.  The hot kernels carry no alignment state.
// It walks the length-class lists of the classify kernels with an escalation list of its own (a sentence the
// encode moved to a later class is simply seen twice: the writes are idempotent).
#ifndef SPMX_KERNELS_ALIGN_H_
#define SPMX_KERNELS_ALIGN_H_

namespace spmx {

struct AlignArgs {
  SpmxDev dev;
  const uint8_t *text;          // packed sentences
  const uint64_t *offs;         // n + 1
  const uint32_t *list;         // the length class's sentences (the encode's own lists)
  const uint32_t *list_count;
  const uint64_t *id_offs;      // n + 1: CSR of the ids
  const int32_t *tok_begin;     // CSR, same shape as the ids: token begins in the normalized (device) text
  uint32_t *begin, *end;        // CSR out
  uint32_t *nbegin, *nend;      // optional CSR out: the same tokens as ranges of the normalized text (as Normalize()
                                // returns it: U+2581 takes three bytes there); 0, 0 for a bos / eos
  uint32_t *status;
  uint32_t rcap, ncap;          // LDS capacities, those of the encode class
  uint32_t *next_list;          // a sentence whose normalized form overflows ncap goes to the next class's align
  uint32_t *next_count;         // list (null past the last staged class: the call fails)
  uint32_t list_cap;            // entries a list holds
};

inline uint32_t AlignLdsBytes(uint32_t rcap, uint32_t ncap, bool norm_spans) {
  const uint32_t n2 = ((ncap + 8) * 2 + 15) & ~15u;
  return ((rcap + 16 + 15) & ~15u) + ((ncap + 16 + 15) & ~15u) + n2 + (norm_spans ? n2 : 0u);
}

SPMX_DEVICE void align_block(const AlignArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  uint8_t *raw = smem;
  uint8_t *norm = smem + ((a.rcap + 16 + 15) & ~15u);
  uint16_t *orig = reinterpret_cast<uint16_t *>(norm + ((a.ncap + 16 + 15) & ~15u));
  uint16_t *spc = orig + (((a.ncap + 8) * 2 + 15) & ~15u) / 2;     // (norm spans) one-byte space symbols before p
  const bool nspans = a.nbegin != nullptr && a.nend != nullptr;
  const bool one = (d.flags & kNfCompressSp) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t count = *a.list_count < a.list_cap ? *a.list_count : a.list_cap;
  for (uint32_t item = static_cast<uint32_t>(wv::block_id()); item < count; item += static_cast<uint32_t>(wv::grid_size())) {
    const uint32_t sid = a.list[item];
    const uint64_t beg = a.offs[sid];
    const uint64_t L64 = a.offs[sid + 1] - beg;
    const uint64_t ib = a.id_offs[sid];
    const int T = static_cast<int>(a.id_offs[sid + 1] - ib);
    const int body = T - d.n_prefix - d.n_suffix;
    if (body < 0) continue;                       // the encode failed this sentence (its status says why)
    if (L64 > a.rcap) {
      if (lane == 0) wv::atomic_or(a.status, kStTooLong);
      continue;
    }
    const int L = static_cast<int>(L64);
    const uint8_t *src = a.text + beg;
    for (int p = lane; p < L; p += 64) raw[p] = src[p];
    wv::sync();
    int fin = L, nlen = 0;
    if (L > 0) nlen = normalize_wave(d, raw, L, norm, static_cast<int>(a.ncap), lane, orig, &fin);
    if (nlen < 0) {
      if (lane == 0) {
        if (a.next_list) {
          const uint32_t at = wv::atomic_add(a.next_count, 1u);
          if (at < a.list_cap) a.next_list[at] = sid;
          else wv::atomic_or(a.status, kStInternal);
        } else {
          wv::atomic_or(a.status, kStTooLong);
        }
      }
      wv::sync();
      continue;
    }
    if (lane == 0) orig[nlen] = static_cast<uint16_t>(fin < 0 ? L : fin);
    if (nspans && one) {
      int run = 0;
      for (int p0 = 0; p0 <= nlen; p0 += 64) {
        const int p = p0 + lane;
        const int c = (p < nlen && norm[p] == kSpByte) ? 1 : 0;
        int t = 0;
        const int before = run + wave_excl_scan(c, lane, &t);
        if (p <= nlen) spc[p] = static_cast<uint16_t>(before);
        run += t;
      }
    }
    wv::sync();
    if (lane < d.n_prefix) {
      const uint32_t v = ((d.extra_eos >> lane) & 1u) ? static_cast<uint32_t>(L) : 0u;
      a.begin[ib + lane] = v; a.end[ib + lane] = v;
      if (nspans) { a.nbegin[ib + lane] = 0; a.nend[ib + lane] = 0; }
    }
    if (lane < d.n_suffix) {
      const uint32_t v = ((d.extra_eos >> (kMaxExtra + lane)) & 1u) ? static_cast<uint32_t>(L) : 0u;
      a.begin[ib + d.n_prefix + body + lane] = v; a.end[ib + d.n_prefix + body + lane] = v;
      if (nspans) { a.nbegin[ib + d.n_prefix + body + lane] = 0; a.nend[ib + d.n_prefix + body + lane] = 0; }
    }
    bool bad = false;
    for (int i0 = 0; i0 < body; i0 += 64) {
      const int i = i0 + lane;                    // token index in text order
      if (i < body) {
        const uint64_t slot = ib + static_cast<uint64_t>(d.n_prefix + (reverse ? body - 1 - i : i));
        const int b = a.tok_begin[slot];
        const int e = i + 1 < body ? a.tok_begin[reverse ? slot - 1 : slot + 1] : nlen;
        if (b < 0 || e < b || e > nlen) { bad = true; }
        else {
          a.begin[slot] = orig[b]; a.end[slot] = orig[e];
          if (nspans) {
            a.nbegin[slot] = static_cast<uint32_t>(b) + (one ? 2u * spc[b] : 0u);
            a.nend[slot] = static_cast<uint32_t>(e) + (one ? 2u * spc[e] : 0u);
          }
        }
      }
    }
    if (wv::any(bad) && lane == 0) wv::atomic_or(a.status, kStInternal);
    wv::sync();                                   // raw / norm / orig are rewritten by the next sentence
  }
}

}  // namespace spmx
#endif
