// BPE segmentation of one normalized sentence by one wavefront.
// Reference: bpe::Model::SampleEncode with alpha = 0 (src/bpe_model.cc:38-203).
//
// The reference keeps an agenda (priority queue) of adjacent symbol pairs whose
// concatenation is a vocabulary piece and repeatedly pops the best
// (max score, then smallest left index), skipping stale entries.  Stale
// entries never win, and a live pair is unique per left symbol, so each pop is
// exactly "argmax over the currently adjacent, currently mergeable pairs".
// Here that argmax is a wave reduction over per-position pair slots in LDS;
// the string hash lookup pieces_.find(left + right) (:91) becomes an integer
// lookup (symL, symR) -> (merged symbol, score) in a table compiled at load
// from every two-way split of every piece (tables.cc).
#ifndef SPMX_KERNELS_BPE_H_
#define SPMX_KERNELS_BPE_H_

namespace spmx {

constexpr uint32_t kSsDead = 0xFFFFFFFFu;     // not (or no longer) the start of a symbol
constexpr uint32_t kSsUnknown = 0x7FFFFFFFu;  // a character with no symbol id: PieceToId -> unk
constexpr uint32_t kSsFrozen = 0x80000000u;   // user-defined symbol: never merged (:85-87)
constexpr int kRevCap = 64;                   // distinct UNUSED merged pieces tracked per sentence

struct BpeLds {
  uint32_t *ssym, *pmerg;
  float *pscore;
  uint16_t *snext, *sprev;
  uint32_t *rev;   // kRevCap x {merged sym, left sym, right sym}
};

SPMX_DEVICE uint32_t bpe_lds_bytes(uint32_t ncap) {
  const uint32_t n4 = ((ncap + 4) * 4 + 15) & ~15u, n2 = ((ncap + 8) * 2 + 15) & ~15u;
  return 3 * n4 + 2 * n2 + kRevCap * 12;
}

SPMX_DEVICE BpeLds carve_bpe(unsigned char *base, uint32_t ncap) {
  const uint32_t n4 = ((ncap + 4) * 4 + 15) & ~15u, n2 = ((ncap + 8) * 2 + 15) & ~15u;
  BpeLds b;
  b.ssym = reinterpret_cast<uint32_t *>(base);
  b.pmerg = reinterpret_cast<uint32_t *>(base + n4);
  b.pscore = reinterpret_cast<float *>(base + 2 * n4);
  b.snext = reinterpret_cast<uint16_t *>(base + 3 * n4);
  b.sprev = reinterpret_cast<uint16_t *>(base + 3 * n4 + n2);
  b.rev = reinterpret_cast<uint32_t *>(base + 3 * n4 + 2 * n2);
  return b;
}

// pieces_.find(left.piece + right.piece) (:88-94) as an integer probe.
SPMX_DEVICE bool pair_lookup(const SpmxDev &d, uint32_t a, uint32_t b, uint32_t *merged, float *score) {
  if (a >= kSsUnknown || b >= kSsUnknown) return false;   // unknown char or frozen symbol
  uint32_t s = HashPair(a, b) & d.pairtab_mask;
  for (;;) {
    const U4 e = d.pairtab[s];
    if (e.x == kSymNone) return false;
    if (e.x == a && e.y == b) {
      *merged = e.z;
      *score = wv::bits_to_float(e.w);
      return true;
    }
    s = (s + 1) & d.pairtab_mask;
  }
}

SPMX_DEVICE uint32_t char_lookup(const SpmxDev &d, uint32_t bytes, uint32_t len) {
  uint32_t s = HashChar(bytes, len) & d.chartab_mask;
  for (;;) {
    const U4 e = d.chartab[s];
    if (e.y == 0) return kSsUnknown;
    if (e.x == bytes && e.y == len) return e.z;
    s = (s + 1) & d.chartab_mask;
  }
}

// rev_merge[piece] = (left, right) (:103-106): last registration wins.  Called
// by ONE lane at a time, in the reference's registration order.
SPMX_DEVICE bool rev_put(uint32_t *rev, int *n_rev, uint32_t merged, uint32_t l, uint32_t r) {
  for (int i = 0; i < *n_rev; ++i)
    if (rev[3 * i] == merged) { rev[3 * i + 1] = l; rev[3 * i + 2] = r; return true; }
  if (*n_rev >= kRevCap) return false;
  rev[3 * *n_rev] = merged; rev[3 * *n_rev + 1] = l; rev[3 * *n_rev + 2] = r;
  ++*n_rev;
  return true;
}

// Returns 0 on an error status (control id), -1 when the UNUSED bookkeeping overflowed (the caller hands the sentence
// to the long form), 1 on success: bid[e] / blen[e] | kTokEnd mark every output piece's end e.
SPMX_DEVICE int bpe_wave(const EncodeArgs &a, const uint8_t *norm, int nlen, int32_t *bid, uint16_t *blen,
                          const BpeLds &B, int lane) {
  const SpmxDev &d = a.dev;
  const bool has_uds = (d.flags & kNfHasUserDefined) != 0;
  const bool track_unused = (d.flags & kNfHasUnused) != 0;
  const uint32_t uroot = has_uds ? (d.utrie[0].x >> kDatBaseShiftDev) : 0u;
  int n_rev = 0;          // meaningful in lane 0 only... kept uniform by broadcasting updates
  bool ok = true;
  // ---- split into symbols (:109-120): PrefixMatch = longest user-defined symbol, else one char
  int next_start = 0;
  for (int b = 0; b < nlen; b += 64) {
    const int p = b + lane;
    const bool valid = p < nlen;
    int uds_len = 0;
    uint32_t uds_id = 0;
    if (has_uds) {
      bool alive = valid;
      uint32_t nb = uroot;
      int depth = 0;
      while (wv::any(alive)) {
        if (alive) {
          const int q = p + depth;
          if (q < nlen) {
            const uint32_t c = norm[q];
            const U2 u = d.utrie[nb ^ c];
            if ((u.x & 0x1FFu) == (0x100u | c)) {
              ++depth;
              nb = u.x >> kDatBaseShiftDev;
              if (u.x & kDatTerminalDev) { uds_len = depth; uds_id = u.y; }
            } else {
              alive = false;
            }
          } else {
            alive = false;
          }
        }
      }
    }
    int step = 1;
    if (valid) {
      if (uds_len > 0) step = uds_len;
      else { step = norm[p] == SpByteOf(d) ? 1 : OneCharLenDev(norm[p]); if (step > nlen - p) step = nlen - p; }
    }
    const uint64_t S = resolve_chain(b, step, valid, &next_start);
    if (valid) {
      uint32_t sym = kSsDead;
      if ((S >> lane) & 1ull) {
        if (uds_len > 0) {
          sym = kSsFrozen | uds_id;
        } else {
          uint32_t bytes = 0;
          for (int k = 0; k < step; ++k) bytes |= static_cast<uint32_t>(norm[p + k]) << (8 * k);
          sym = char_lookup(d, bytes, static_cast<uint32_t>(step));
        }
        B.snext[p] = static_cast<uint16_t>(p + step);
        if (p + step < nlen) B.sprev[p + step] = static_cast<uint16_t>(p);
      }
      B.ssym[p] = sym;
      B.pmerg[p] = kSymNone;
    }
  }
  if (lane == 0) B.sprev[0] = 0xFFFFu;
  wv::sync();
  // ---- all bigrams (:127-129), in increasing left position
  for (int b = 0; b < nlen; b += 64) {
    const int p = b + lane;
    bool found = false, unused = false;
    uint32_t merged = 0, l = 0, r = 0;
    if (p < nlen && B.ssym[p] != kSsDead) {
      const int q = B.snext[p];
      if (q < nlen) {
        float sc = 0.f;
        l = B.ssym[p]; r = B.ssym[q];
        found = pair_lookup(d, l, r, &merged, &sc);
        if (found) {
          B.pmerg[p] = merged;
          B.pscore[p] = sc;
          unused = (d.sym_final[merged] & kSfUnused) != 0;   // IsUnusedInlined(it->second) (:103)
        }
      }
    }
    if (track_unused) {
      uint64_t mu = wv::ballot(found && unused);
      while (mu) {
        const int i = wv::ffs64(mu) - 1;
        mu &= mu - 1;
        const uint32_t mm = wv::shfl(merged, i), ll = wv::shfl(l, i), rr = wv::shfl(r, i);
        if (lane == 0 && !rev_put(B.rev, &n_rev, mm, ll, rr)) ok = false;
        wv::sync();
      }
    }
  }
  wv::sync();
  // ---- main loop (:142-173)
  for (;;) {
    float bs = 0.f;
    int bp = -1;
    for (int p = lane; p < nlen; p += 64) {
      if (B.pmerg[p] != kSymNone) {
        const float s = B.pscore[p];
        if (bp < 0 || s > bs) { bs = s; bp = p; }     // ties keep the smaller position (:53-56)
      }
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const float os = wv::shfl(bs, lane ^ dlt);
      const int op = wv::shfl(bp, lane ^ dlt);
      if (op >= 0 && (bp < 0 || os > bs || (os == bs && op < bp))) { bs = os; bp = op; }
    }
    if (bp < 0) break;
    const int left = bp;
    const int right = B.snext[left];
    const uint32_t msym = B.pmerg[left];
    const int nn = B.snext[right];
    const int pv = B.sprev[left];
    wv::sync();
    if (lane == 0) {                                   // :159-168
      B.ssym[left] = msym;
      B.snext[left] = static_cast<uint16_t>(nn);
      if (nn < nlen) B.sprev[nn] = static_cast<uint16_t>(left);
      B.ssym[right] = kSsDead;
      B.pmerg[right] = kSymNone;
      B.pmerg[left] = kSymNone;
    }
    wv::sync();
    // :171-172 the two new neighbours; lane 0 = (prev, left), lane 1 = (left, next)
    bool found = false, unused = false;
    uint32_t merged = 0, l = 0, r = 0;
    if (lane == 0 && pv != 0xFFFF) {
      float sc = 0.f;
      l = B.ssym[pv]; r = msym;
      found = pair_lookup(d, l, r, &merged, &sc);
      if (found) { B.pmerg[pv] = merged; B.pscore[pv] = sc; }
      else B.pmerg[pv] = kSymNone;
    } else if (lane == 1 && nn < nlen) {
      float sc = 0.f;
      l = msym; r = B.ssym[nn];
      found = pair_lookup(d, l, r, &merged, &sc);
      if (found) { B.pmerg[left] = merged; B.pscore[left] = sc; }
    }
    if (track_unused) {
      if (found) unused = (d.sym_final[merged] & kSfUnused) != 0;
      uint64_t mu = wv::ballot(found && unused);
      while (mu) {
        const int i = wv::ffs64(mu) - 1;
        mu &= mu - 1;
        const uint32_t mm = wv::shfl(merged, i), ll = wv::shfl(l, i), rr = wv::shfl(r, i);
        if (lane == 0 && !rev_put(B.rev, &n_rev, mm, ll, rr)) ok = false;
        wv::sync();
      }
    }
    wv::sync();
  }
  n_rev = wv::shfl(n_rev, 0);
  ok = !wv::any(!ok);
  if (!ok) return -1;
  // ---- output pieces (:175-200): every live symbol, UNUSED ones resegmented through rev_merge
  bool bad = false, deep = false;
  for (int b = 0; b < nlen; b += 64) {
    const int p = b + lane;
    if (p < nlen && B.ssym[p] != kSsDead) {
      const uint32_t top = B.ssym[p];
      const int end = B.snext[p];
      if (top == kSsUnknown) {                          // PieceToId of an unseen character -> unk_id
        bid[end] = d.unk_id;
        blen[end] = static_cast<uint16_t>((end - p) | kTokEnd);
      } else {
        const uint32_t sym0 = top & ~kSsFrozen;
        const uint32_t f0 = d.sym_final[sym0];
        if (!(f0 & kSfUnused) || !track_unused) {
          if (f0 & kSfControl) bad = true;
          bid[end] = static_cast<int32_t>(f0 & kSfIdMask);
          blen[end] = static_cast<uint16_t>((end - p) | kTokEnd);
        } else {
          // resegment(w) (:176-193), depth-first, left part first
          uint32_t stack[kMaxResegDepth];
          int sp = 0;
          int pos = p;
          stack[sp++] = sym0;
          while (sp > 0) {
            const uint32_t s = stack[--sp];
            const uint32_t f = d.sym_final[s];
            int ri = -1;
            if (f & kSfUnused)
              for (int i = 0; i < n_rev; ++i) if (B.rev[3 * i] == s) { ri = i; break; }
            if (ri < 0 || sp + 2 > kMaxResegDepth) {
              if (ri >= 0) deep = true;              // the recursion outgrew the stack: the long form has none
              if (f & kSfControl) bad = true;
              const int len = d.sym_len[s];
              bid[pos + len] = static_cast<int32_t>(f & kSfIdMask);
              blen[pos + len] = static_cast<uint16_t>(len | kTokEnd);
              pos += len;
            } else {
              stack[sp++] = B.rev[3 * ri + 2];
              stack[sp++] = B.rev[3 * ri + 1];
            }
          }
        }
      }
    }
  }
  wv::sync();
  if (wv::any(bad)) return 0;
  if (wv::any(deep)) return -1;
  return 1;
}

}  // namespace spmx
#endif
