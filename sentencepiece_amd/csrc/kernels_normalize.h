// Batch Normalize on the device: SentencePieceProcessor::Normalize(input, &normalized, &norm_to_orig)
// (src/sentencepiece_processor.cc:933-945 -> Normalizer::Normalize, src/normalizer.cc:71-186) for a packed batch:
// the normalized text as the reference returns it (U+2581 in its three bytes) and, optionally, the alignment vector.
// One sentence per wavefront over the length-class lists of the classify kernels, the position-parallel normalizer
// of kernels.h; count pass -> scan -> write pass, as the decode kernels.  A cold path: it feeds the piece strings
// of the spans form (the piece of an unknown token is its normalized text, sentencepiece_processor.cc:614-617).
#ifndef SPMX_KERNELS_NORMALIZE_H_
#define SPMX_KERNELS_NORMALIZE_H_

namespace spmx {

constexpr uint32_t kNoClosingEntry = 0xFFFFFFFFu;   // n2o: the reference's vector is EMPTY for this sentence (:77-79, :96-99)

struct NormalizeArgs {
  SpmxDev dev;
  const uint8_t *text;          // packed sentences
  const uint64_t *offs;         // n + 1
  const uint32_t *list;
  const uint32_t *list_count;
  uint32_t *next_list;          // count pass: a sentence whose normalized form overflows ncap goes to the next class
  uint32_t *next_count;
  uint32_t *counts;             // per sentence: normalized bytes (count pass)
  const uint64_t *norm_offs;    // n + 1 (write pass)
  uint8_t *norm;                // packed normalized text
  uint32_t *n2o;                // optional: sentence s owns entries [norm_offs[s] + s, norm_offs[s + 1] + s + 1): one
                                // per normalized byte + the closing one (kNoClosingEntry where the reference has none)
  uint32_t *status;
  uint32_t rcap, ncap;
  uint32_t device_text;         // keep the device form (U+2581 as one byte under kNfCompressSp): the input of kernels_nbest.h
};

inline uint32_t NormalizeLdsBytes(uint32_t rcap, uint32_t ncap) {
  return ((rcap + 16 + 15) & ~15u) + ((ncap + 16 + 15) & ~15u) + (((ncap + 8) * 2 + 15) & ~15u);
}

template <bool WRITE>
SPMX_DEVICE void normalize_block(const NormalizeArgs &a, unsigned char *smem) {
  const int lane = wv::lane();
  const SpmxDev &d = a.dev;
  uint8_t *raw = smem;
  uint8_t *norm = smem + ((a.rcap + 16 + 15) & ~15u);
  uint16_t *orig = reinterpret_cast<uint16_t *>(norm + ((a.ncap + 16 + 15) & ~15u));
  const bool one = (d.flags & kNfCompressSp) != 0 && !a.device_text;
  const bool want_n2o = WRITE && a.n2o != nullptr;
  const uint32_t count = *a.list_count;
  for (uint32_t item = static_cast<uint32_t>(wv::block_id()); item < count; item += static_cast<uint32_t>(wv::grid_size())) {
    const uint32_t sid = a.list[item];
    const uint64_t beg = a.offs[sid];
    const uint64_t L64 = a.offs[sid + 1] - beg;
    if (L64 > a.rcap) {                            // only reachable in the last class
      if (lane == 0) { wv::atomic_or(a.status, kStTooLong); if (!WRITE) a.counts[sid] = 0; }
      continue;
    }
    const int L = static_cast<int>(L64);
    const uint8_t *src = a.text + beg;
    for (int p = lane; p < L; p += 64) raw[p] = src[p];
    wv::sync();
    int fin = -1, nlen = 0;
    if (L > 0) nlen = normalize_wave(d, raw, L, norm, static_cast<int>(a.ncap), lane, want_n2o ? orig : nullptr, &fin);
    if (nlen < 0) {
      if (!WRITE) {
        if (a.next_list) { if (lane == 0) a.next_list[wv::atomic_add(a.next_count, 1u)] = sid; }
        else if (lane == 0) { wv::atomic_or(a.status, kStTooLong); a.counts[sid] = 0; }
      }
      wv::sync();
      continue;                                    // (write pass: the sentence is in a later list too)
    }
    // position in the reference's text: every one-byte space symbol before p adds two bytes
    uint8_t *dst = WRITE ? a.norm + a.norm_offs[sid] : nullptr;
    uint32_t *dn = want_n2o ? a.n2o + a.norm_offs[sid] + sid : nullptr;
    int run = 0;
    for (int p0 = 0; p0 < nlen; p0 += 64) {
      const int p = p0 + lane;
      const bool sp = one && p < nlen && norm[p] == kSpByte;
      int t = 0;
      const int o = p + 2 * (run + wave_excl_scan(sp ? 1 : 0, lane, &t));
      run += t;
      if (WRITE && p < nlen) {
        if (sp) { dst[o] = 0xE2; dst[o + 1] = 0x96; dst[o + 2] = 0x81; }
        else dst[o] = norm[p];
        if (dn) { dn[o] = orig[p]; if (sp) { dn[o + 1] = orig[p]; dn[o + 2] = orig[p]; } }
      }
    }
    const int true_len = nlen + 2 * run;
    if (!WRITE && lane == 0) a.counts[sid] = static_cast<uint32_t>(true_len);
    if (dn && lane == 0) dn[true_len] = fin < 0 ? kNoClosingEntry : static_cast<uint32_t>(fin);
    wv::sync();                                    // raw / norm / orig are rewritten by the next sentence
  }
}

}  // namespace spmx
#endif
