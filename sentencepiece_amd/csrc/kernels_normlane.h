// Normalizer::Normalize (src/normalizer.cc:71-186) of ONE sentence by ONE lane -- the normalizers of the
// lane-per-sentence kernels (kernels_stream.h, kernels_long.h).  Two forms:
//
//   fast_norm_stream   the ASCII fast path: every NormalizePrefix result is the byte itself (:231-244); also keeps
//                      characters that start no charsmap key and malformed bytes; gives up (-1) on anything else
//   norm_lane_any      the reference's loop verbatim, one NormalizePrefix result after another (:195-253): longest
//                      user-defined symbol (PrefixMatcher, :324-346), else longest charsmap rule by a Darts walk
//                      (darts.h:467-513), else one UTF-8 character, else U+FFFD for a malformed byte
//                      (util.cc:51-84); every normalizer_spec / trainer_spec switch (dummy prefix, extra
//                      whitespace, escaping with the one-byte or the three-byte space symbol, whitespace as
//                      suffix), any length, optional norm_to_orig.  Nothing is staged per sentence: raw bytes come
//                      through a 64-byte LDS window (evicted bytes are read from HBM again), output goes to a sink.
#ifndef SPMX_KERNELS_NORMLANE_H_
#define SPMX_KERNELS_NORMLANE_H_

namespace spmx {

// A lane's text column in the streaming kernels' HBM scratch: dword k of the column is p[k << sh] (1 << sh lanes
// share a slab, position-major: lanes advance together, so they share lines).
struct TextCol {
  uint32_t *p;
  uint32_t sh;
  SPMX_DEVICE uint32_t &dw(int k) const { return p[static_cast<uint32_t>(k) << sh]; }
};
SPMX_DEVICE uint32_t col_byte(const TextCol &c, int pos) { return (c.dw(pos >> 2) >> (8 * (pos & 3))) & 0xFFu; }

// byte classes of the ASCII fast path (StreamLds::bcls)
constexpr uint32_t kBcComplex = 1u;    // not handled by fast_norm_stream: a charsmap rule may start here

// ---- sinks --------------------------------------------------------------------------------------------------
// put(b, o): append the normalized byte b; o = raw offset at which the prefix that produced it starts
// (norm_to_orig, :142-152).  A sink stores nothing past its capacity but keeps counting: the text may pass the
// capacity only to fall back below it (trailing space symbols are cut at the end, :166-176); overflowed() is asked
// once the sentence is done.
struct ColSink {            // a text column, four bytes per store
  TextCol c;
  int cap;
  int w = 0;                // bytes of the text so far
  int sw = 0;               // ... of which stored or in acc (min(w, cap))
  uint32_t acc = 0;
  SPMX_DEVICE void put(uint32_t b, int) {
    if (w < cap) {
      acc |= b << (8 * (sw & 3));
      ++sw;
      if ((sw & 3) == 0) { c.dw((sw >> 2) - 1) = acc; acc = 0; }
    }
    ++w;
  }
  SPMX_DEVICE void flush() { c.dw(sw >> 2) = acc; }                 // the last, partial dword
  // the text ends at n <= w: later puts continue from there
  SPMX_DEVICE void truncate(int n) {
    if (n < sw) {
      if ((n >> 2) < (sw >> 2)) acc = c.dw(n >> 2);
      acc &= (1u << (8 * (n & 3))) - 1u;
      sw = n;
    }
    w = n;
  }
  SPMX_DEVICE bool overflowed() const { return w > cap; }
};
// flat bytes dst[0, cap) (+ norm_to_orig o[0, cap)) in HBM; a null dst only counts
struct FlatSink {
  uint8_t *dst;
  uint32_t *o;
  long long cap;
  int w = 0;
  SPMX_DEVICE void put(uint32_t b, int orig) {
    if (dst && w < cap) {
      dst[w] = static_cast<uint8_t>(b);
      if (o) o[w] = static_cast<uint32_t>(orig);
    }
    ++w;
  }
  SPMX_DEVICE void flush() {}
  SPMX_DEVICE void truncate(int n) { w = n; }
  SPMX_DEVICE bool overflowed() const { return dst != nullptr && w > cap; }
};

// trie_->commonPrefixSearch, longest key (src/normalizer.cc:218-228; darts.h:467-513) at s[0, rem): the key's length
// in the low word and its value (offset of the replacement in the blob) in the high word; length 0 when no key matches,
// 0xFFFFFFFF when the longest key holds an ASCII byte (fast_norm_stream's bookkeeping covers keys of non-ASCII bytes
// only).  A real call (it sits inside fast_norm_stream's unrolled byte loop), results in registers.
SPMX_DEVICE_CALL unsigned long long charsmap_longest(const uint32_t *ndarts, uint32_t ndarts_n, const uint8_t *s, int rem) {
  uint32_t off = 0;
  uint32_t pos = DartsOffset(ndarts[0]);
  int best = 0;
  bool ascii = false, best_ascii = false;
  for (int depth = 0; depth < rem;) {
    const uint32_t c = s[depth];
    pos ^= c;
    if (pos >= ndarts_n) break;
    const uint32_t u = ndarts[pos];
    if ((u & 0x800000FFu) != c) break;                   // unit.label() == c
    pos ^= DartsOffset(u);
    ascii = ascii || c < 0x80u;
    ++depth;
    if ((u >> 8) & 1u) {                                 // has_leaf: the value sits in the unit at pos
      if (pos >= ndarts_n) break;
      best = depth;
      best_ascii = ascii;
      off = ndarts[pos] & 0x7FFFFFFFu;
    }
  }
  return static_cast<unsigned long long>(off) << 32 | (best_ascii ? 0xFFFFFFFFu : static_cast<uint32_t>(best));
}

// The replacement string of a matched charsmap key (src/normalizer.cc:245-250: the C string at normalized_[value]) goes
// to a text column as the reference's loop writes a NormalizePrefix result (:133-163): leading spaces dropped after a
// space, every other ' ' written as the (one-byte) space symbol.  State of the caller's loop by reference: acc (the
// partial dword), w (bytes written), nsp (space symbols written), wl (length up to the last byte that is no space
// symbol), P (is_prev_space), seen (something other than a space was written).  `left`: raw bytes after the key.
// false: the replacement does not fit the column (the caller leaves the sentence to norm_lane_any).
SPMX_DEVICE bool write_rule(const SpmxDev &d, uint32_t rule_off, int left, uint32_t sp, bool rm, const TextCol &gt, int tcap,
                            uint32_t &acc, int &w, int &nsp, int &wl, bool &P, bool &seen) {
  int n = 0;
  while (rule_off + static_cast<uint32_t>(n) < d.nblob_n && d.nblob[rule_off + n] != 0) ++n;
  if (w + n + left > tcap) return false;
  int j = 0;
  while (P && j < n && d.nblob[rule_off + j] == 0x20u) ++j;          // :137-138
  if (j < n) {
    uint32_t last = 0;
    for (; j < n; ++j) {
      last = d.nblob[rule_off + j];
      const uint32_t b = last == 0x20u ? sp : last;                  // :143-148
      acc |= b << (8 * (w & 3));
      ++w;
      if ((w & 3) == 0) { gt.dw((w >> 2) - 1) = acc; acc = 0; }
      if (b == sp) ++nsp;
      else { wl = w; seen = true; }
    }
    P = last == 0x20u && rm;                                         // :154
  }
  if (!rm) P = false;                                                // :160-162
  return true;
}

// Normalize() of one all-ASCII sentence by ONE lane (every NormalizePrefix result is the byte itself, :231-244): raw
// text in HBM, read as aligned 16-byte blocks of the ABSOLUTE address (a block that holds a byte of the sentence lies
// in the same page as that byte, so the over-read at either end stays inside the caller's mapping whatever the
// alignment of the buffer) -> the lane's text column, four bytes at a time.  Valid when the space symbol is one byte
// wide (kNfCompressSp, or no whitespace escaping) and the model has no user-defined symbols; not for
// whitespace-as-suffix models.  A byte whose bcls entry says kBcComplex makes the lane give up (-1).
// Non-ASCII characters: one whose first two bytes start no charsmap key is itself (:231-244); otherwise the longest
// key is looked up right here (charsmap_longest) and its replacement written as the reference's loop would (:133-163:
// leading spaces dropped after a space, every other ' ' escaped); what is left to norm_lane_any is a key that starts
// at an ASCII byte and continues with a non-ASCII one (a letter and a combining mark), a literal U+2581, a
// replacement that does not fit the column.
// *n_sp: how many bytes of the result are the space symbol (sizes the id slot under byte fallback).
// (Tried: appending a dword of four plain bytes 0x21 .. 0x7E whole.  The lanes of a wave disagree on it dword by
// dword, so the wave runs both paths: 3 % slower end to end.)
SPMX_DEVICE int fast_norm_stream(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int L, const TextCol &gt,
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
  int drop = 0;                   // bytes of a matched charsmap key still to pass over (its replacement is written)
  const uint8_t *first = gtext + beg;
  const uintptr_t a0 = reinterpret_cast<uintptr_t>(first) & ~static_cast<uintptr_t>(15);
  const uint8_t *blk = reinterpret_cast<const uint8_t *>(a0);
  int rel = static_cast<int>(static_cast<long long>(a0) - static_cast<long long>(reinterpret_cast<uintptr_t>(first)));   // index of the block's first byte within the sentence (<= 0 at first)
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
              if ((w & 3) == 0) { if (!(SPMX_EXP & 4)) gt.dw((w >> 2) - 1) = acc; acc = 0; }
            }
            P = is_sp && rm;                        // :154-162
            if (!is_sp) { wl = w; seen = true; }
          }
        }
      } else {
        // ---- a dword with a non-ASCII byte (rare in this kernel's tiles).  A character whose first two bytes
        // start no charsmap key (tables.cc npair) normalizes to itself (:231-244), a malformed byte to U+FFFD
        // (util.cc:51-84); a possible rule, or a literal U+2581, leaves the sentence to norm_lane_any. ----
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
                if ((w & 3) == 0) { if (!(SPMX_EXP & 4)) gt.dw((w >> 2) - 1) = acc; acc = 0; }
              }
              P = is_sp && rm;
              if (!is_sp) { wl = w; seen = true; }
            } else {
              uint32_t o0 = c, o1 = 0, o2 = 0;
              int n_out = 1;
              bool plain = true;                    // this byte contributes n_out bytes that are not spaces
              const int rem = L - (rel + k);
              if (drop > 0) {
                --drop;
                plain = false;
              } else if (skip > 0) {
                --skip;
              } else {
                const uint32_t b1 = rem >= 2 ? (wd[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xFFu : 0u;
                const uint32_t b2 = rem >= 3 ? (wd[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 0xFFu : 0u;
                const uint32_t b3 = rem >= 4 ? (wd[(k + 3) >> 2] >> (8 * ((k + 3) & 3))) & 0xFFu : 0u;
                const uint32_t pb = k > 0 ? (wd[(k > 0 ? k - 1 : 0) >> 2] >> (8 * ((k > 0 ? k - 1 : 0) & 3))) & 0xFFu : carry;
                const uint32_t prevc = rel + k > 0 ? pb : 0x80u;      // nothing before the first byte of the sentence
                int rule_len = 0;
                uint32_t rule_off = 0;
                if (has_map) {
                  if (prevc < 0x80u && ((d.npair[(prevc << 8 | c) >> 5] >> (c & 31u)) & 1u)) bad |= kBcComplex;   // a key may start at the ASCII byte before
                  if ((d.npair[(c << 8 | b1) >> 5] >> (b1 & 31u)) & 1u) {                            // ... or here
                    const unsigned long long hit = charsmap_longest(d.ndarts, d.ndarts_n, first + (rel + k), rem);
                    rule_len = static_cast<int>(static_cast<uint32_t>(hit));
                    rule_off = static_cast<uint32_t>(hit >> 32);
                    if (rule_len < 0) { bad |= kBcComplex; rule_len = 0; }
                  }
                }
                if (rule_len > 0) {
                  plain = false;
                  if (!write_rule(d, rule_off, rem - rule_len, sp, rm, gt, tcap, acc, w, nsp, wl, P, seen)) bad |= kBcComplex;
                  drop = rule_len - 1;
                } else {
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
                  if (mb) skip = mb - 1;
                  else { o0 = 0xEFu; o1 = 0xBFu; o2 = 0xBDu; n_out = 3; }
                }
              }
              if (plain) {
                // a malformed byte grows into three: keep "what is written + what is left to read" within the
                // column (all other bytes produce at most one), else leave the sentence to norm_lane_any
                if (w + rem + 2 > tcap) bad |= kBcComplex;
                else {
#pragma unroll
                  for (int j = 0; j < 3; ++j) {
                    if (j < n_out) {
                      acc |= (j == 0 ? o0 : (j == 1 ? o1 : o2)) << (8 * (w & 3));
                      ++w;
                      if ((w & 3) == 0) { if (!(SPMX_EXP & 4)) gt.dw((w >> 2) - 1) = acc; acc = 0; }
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
    }
    carry = cur.w >> 24;
    cur = nxt;
    blk += 16;
    rel += 16;
  }
  gt.dw(w >> 2) = acc;                          // the last, partial dword
  if (bad & kBcComplex) return -1;
  if (rm) {
    if (!seen) return 0;                        // :86-100 nothing but spaces
    nsp -= w - wl;                              // the trimmed tail is nothing but space symbols
    w = wl;
  }
  *n_sp = nsp;
  return w;
}

// The same contract as fast_norm_stream for text that is mostly NOT ASCII (CJK ...): one NormalizePrefix result per
// iteration instead of one byte -- the raw bytes come through a two-dword register window of aligned loads, a
// character that starts no charsmap key is validated and appended whole (:231-244), a possible key is looked up
// (charsmap_longest) and its replacement written (write_rule).  Leaves to norm_lane_any (-1) what fast_norm_stream
// leaves to it.  A tile takes this form when enough of its sentences begin with non-ASCII text (kernels_stream.h).
SPMX_DEVICE int char_norm_stream(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int L, const TextCol &gt,
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
  bool bad = false;
  const uint8_t *first = gtext + beg;
  const uintptr_t fa = reinterpret_cast<uintptr_t>(first);
  // window: the aligned dwords at sentence offsets wb and wb + 4 (a dword that holds a byte of the sentence lies in
  // that byte's page: no over-read beyond the caller's mapping)
  const uint32_t *wp = reinterpret_cast<const uint32_t *>(fa & ~static_cast<uintptr_t>(3));
  int wb = -static_cast<int>(fa & 3u);
  uint32_t wlo = wp[0];
  uint32_t whi = wb + 4 < L ? wp[1] : 0u;
  uint32_t prevc = 0x80u;         // the raw byte before p when it is ASCII (a key may start there), else 0x80
  int p = 0;
  while (p < L && !bad) {
    while (p - wb >= 4) { wlo = whi; wb += 4; ++wp; whi = wb + 4 < L ? wp[1] : 0u; }
    const int rem = L - p;
    uint32_t v = static_cast<uint32_t>((static_cast<unsigned long long>(whi) << 32 | wlo) >> (8 * (p - wb)));   // raw bytes p .. p + 3
    if (rem < 4) v &= (1u << (8 * rem)) - 1u;                // (what follows the sentence is the next sentence's)
    const uint32_t c = v & 0xFFu;
    if (c < 0x80u) {
      // ---- an ASCII byte: the prefix is the byte itself unless bcls says a key may start here ----
      bad = (bcls[c] & kBcComplex) != 0;
      const bool is_sp = c == 0x20u;
      if (!is_sp || !P) {                                    // :137-138 a space after a space is dropped
        acc |= (is_sp ? sp : c) << (8 * (w & 3));
        ++w;
        nsp += is_sp ? 1 : 0;
        if ((w & 3) == 0) { gt.dw((w >> 2) - 1) = acc; acc = 0; }
      }
      P = is_sp && rm;                                       // :154-162
      if (!is_sp) { wl = w; seen = true; }
      prevc = c;
      ++p;
      continue;
    }
    const uint32_t b1 = (v >> 8) & 0xFFu, b2 = (v >> 16) & 0xFFu, b3 = v >> 24;
    int rule_len = 0;
    uint32_t rule_off = 0;
    if (has_map) {
      if (prevc < 0x80u && ((d.npair[(prevc << 8 | c) >> 5] >> (c & 31u)) & 1u)) bad = true;   // a key may start at the ASCII byte before
      if ((d.npair[(c << 8 | b1) >> 5] >> (b1 & 31u)) & 1u) {                                  // ... or here
        const unsigned long long hit = charsmap_longest(d.ndarts, d.ndarts_n, first + p, rem);
        rule_len = static_cast<int>(static_cast<uint32_t>(hit));
        rule_off = static_cast<uint32_t>(hit >> 32);
        if (rule_len < 0) { bad = true; rule_len = 0; }
      }
    }
    prevc = 0x80u;
    if (rule_len > 0) {
      if (!write_rule(d, rule_off, rem - rule_len, sp, rm, gt, tcap, acc, w, nsp, wl, P, seen)) bad = true;
      p += rule_len;
      continue;
    }
    // :231-244 one UTF-8 character (DecodeUTF8, util.cc:51-84), else U+FFFD for one malformed byte
    const bool t1 = (b1 & 0xC0u) == 0x80u, t2 = (b2 & 0xC0u) == 0x80u, t3 = (b3 & 0xC0u) == 0x80u;
    int mb = 0;
    if (rem >= 2 && (c & 0xE0u) == 0xC0u) {
      if (t1 && ((c & 0x1Fu) << 6 | (b1 & 0x3Fu)) >= 0x80u) mb = 2;
    } else if (rem >= 3 && (c & 0xF0u) == 0xE0u) {
      const uint32_t cp = (c & 0x0Fu) << 12 | (b1 & 0x3Fu) << 6 | (b2 & 0x3Fu);
      if (t1 && t2 && cp >= 0x800u && (cp < 0xD800u || cp >= 0xE000u)) mb = 3;
      if (c == 0xE2u && b1 == 0x96u && b2 == 0x81u) bad = true;            // a literal U+2581
    } else if (rem >= 4 && (c & 0xF8u) == 0xF0u) {
      const uint32_t cp = (c & 0x07u) << 18 | (b1 & 0x3Fu) << 12 | (b2 & 0x3Fu) << 6 | (b3 & 0x3Fu);
      if (t1 && t2 && t3 && cp >= 0x10000u && cp <= 0x10FFFFu) mb = 4;
    }
    const int n_out = mb ? mb : 3;
    const int consumed = mb ? mb : 1;
    const uint32_t out = mb ? (mb == 4 ? v : v & ((1u << (8 * mb)) - 1u)) : 0x00BDBFEFu;
    // "what is written + what is left to read" stays within the column (a malformed byte grows into three)
    if (w + n_out + rem - consumed > tcap) { bad = true; break; }
    {
      const unsigned long long a64 = static_cast<unsigned long long>(acc) | (static_cast<unsigned long long>(out) << (8 * (w & 3)));
      if ((w & 3) + n_out >= 4) { gt.dw(w >> 2) = static_cast<uint32_t>(a64); acc = static_cast<uint32_t>(a64 >> 32); }
      else acc = static_cast<uint32_t>(a64);
      w += n_out;
    }
    P = false;
    wl = w;
    seen = true;
    p += consumed;
  }
  if (bad) return -1;
  gt.dw(w >> 2) = acc;                          // the last, partial dword
  if (rm) {
    if (!seen) return 0;                        // :86-100 nothing but spaces
    nsp -= w - wl;                              // the trimmed tail is nothing but space symbols
    w = wl;
  }
  *n_sp = nsp;
  return w;
}

// True when the preconditions of fast_norm_stream hold for this model (host and device agree on it).
SPMX_HD inline bool StreamFastEligible(uint32_t flags) {
  return !(flags & kNfHasUserDefined) && ((flags & kNfCompressSp) || !(flags & kNfEscapeWs)) &&
         !((flags & kNfAddDummyPrefix) && (flags & kNfWsSuffix));
}

constexpr int kRawWin = 64;         // bytes of raw text a lane keeps in LDS (norm_lane_any)
constexpr int kRawWinBytes = kRawWin + 16;   // per lane, 16-byte aligned

// Normalize() of one sentence of ANY content under ANY normalizer_spec by ONE lane.  `rawwin`: this lane's
// kRawWinBytes of LDS (16-byte aligned).  Returns the normalized length (sink.w), or -1 when it exceeds the sink's capacity.
// *n_sp: space symbols in the result (device form: one byte each under kNfCompressSp; under three-byte escaping the
// count is not used).  *orig_end (optional): the closing entry of norm_to_orig (:181), -1 where the reference's
// vector is empty (:77-79, :96-99).
template <typename Sink>
SPMX_DEVICE int norm_lane_any(const SpmxDev &d, const uint8_t *gtext, uint64_t beg, int L, Sink &out, uint8_t *rawwin,
                              int *n_sp, int *orig_end = nullptr) {
  const uint32_t F = d.flags;
  const bool rm = (F & kNfRemoveExtraWs) != 0;
  const bool one = (F & kNfCompressSp) != 0;               // U+2581 is the single byte kSpByte (dev.h)
  const bool esc3 = (F & kNfEscapeWs) != 0 && !one;        // escaped spaces take three bytes
  const uint32_t sp1 = one ? kSpByte : 0x20u;              // the one-byte space symbol when !esc3
  const bool has_map = (F & kNfHasCharsmap) != 0, has_uds = (F & kNfHasUserDefined) != 0;
  const bool dummy = (F & kNfAddDummyPrefix) != 0, suffix = (F & kNfWsSuffix) != 0;
  const uint32_t droot = has_map ? DartsOffset(d.ndarts[0]) : 0u;
  const uint32_t uroot = has_uds ? (d.utrie[0].x >> kDatBaseShiftDev) : 0u;
  if (orig_end) *orig_end = -1;
  *n_sp = 0;
  if (L <= 0) return 0;                                    // :77-79
  // raw byte i of the sentence through the LDS window, which is indexed by the low bits of the absolute address and
  // filled one aligned 16-byte block at a time, forward only; a byte the window no longer holds (a long rule or
  // user-defined symbol was walked past it) comes from HBM again
  const uint8_t *first = gtext + beg;
  const uintptr_t fa = reinterpret_cast<uintptr_t>(first);
  int hi;                                                  // raw bytes [max(0, hi - kRawWin), hi) are in the window
  {
    const uintptr_t q0 = fa & ~static_cast<uintptr_t>(15);
    *reinterpret_cast<Q4 *>(rawwin + (q0 & (kRawWin - 1))) = *reinterpret_cast<const Q4 *>(q0);
    hi = static_cast<int>(q0 + 16 - fa);
  }
  auto raw = [&](int i) __attribute__((always_inline)) -> uint32_t {
    while (i >= hi) {
      const uintptr_t q = fa + static_cast<uintptr_t>(hi);               // 16-byte aligned
      *reinterpret_cast<Q4 *>(rawwin + (q & (kRawWin - 1))) = *reinterpret_cast<const Q4 *>(q);
      hi += 16;
    }
    if (i < hi - kRawWin) return first[i];
    return rawwin[(fa + static_cast<uintptr_t>(i)) & (kRawWin - 1)];
  };
  int wl = 0, nsp = 0;           // output length up to the last byte that is not part of a trailing space symbol
  int o_wl = 0;                  // while wl < out.w: norm_to_orig of the byte at wl (the first trailing space symbol, :172)
  int save = 0, save_o = 0, o_e2 = 0;   // esc3: wl / o_wl before the character that starts with the last 0xE2 written, its orig
  bool save_run = false;         // esc3: ... and whether a trailing run was already open then
  uint32_t tail = 0;             // esc3: the last three bytes written
  auto emit = [&](uint32_t b, int orig) __attribute__((always_inline)) {
    const int before = out.w;
    out.put(b, orig);
    if (esc3) {                                            // EndsWith(normalized, "\xe2\x96\x81") (:166-170)
      if (b == 0xE2u) { save = wl; save_run = wl < before; save_o = o_wl; o_e2 = orig; }
      tail = (tail << 8 | b) & 0xFFFFFFu;
      if (tail == 0xE29681u) { wl = save; o_wl = save_run ? save_o : o_e2; ++nsp; }
      else wl = out.w;
    } else if (b != sp1) {
      wl = out.w;
    } else {
      if (wl == before) o_wl = orig;
      ++nsp;
    }
  };
  auto emit_space = [&](int orig) __attribute__((always_inline)) {      // add_ws / an escaped ' ' (:112-122, :143-148)
    if (esc3) { emit(0xE2u, orig); emit(0x96u, orig); emit(0x81u, orig); }
    else emit(sp1, orig);
  };
  // NormalizePrefix at raw offset p (:195-253): kind 0 raw bytes [src, src + len), 1 rule string
  // nblob[src, src + len), 2 U+FFFD, 3 the space symbol (a literal U+2581 under kNfCompressSp)
  struct Pfx { int kind, len, consumed; uint32_t src; };
  auto prefix = [&](int p) __attribute__((always_inline)) -> Pfx {
    const int rem = L - p;
    if (has_uds) {                                         // matcher_->PrefixMatch (:201-205, :324-346): longest symbol
      uint32_t nb = uroot;
      int uds_len = 0;
      for (int depth = 0; depth < rem;) {
        const uint32_t c = raw(p + depth);
        const U2 u = d.utrie[nb ^ c];
        if ((u.x & 0x1FFu) != (0x100u | c)) break;
        ++depth;
        nb = u.x >> kDatBaseShiftDev;
        if (u.x & kDatTerminalDev) uds_len = depth;
      }
      if (uds_len > 0) return Pfx{0, uds_len, uds_len, static_cast<uint32_t>(p)};
    }
    const uint32_t b0 = raw(p);
    int rule_len = 0;
    uint32_t rule_off = 0;
    bool walk = has_map;
    if (walk) {                                            // no key starts with these two bytes (tables.cc npair)
      const uint32_t b1 = rem >= 2 ? raw(p + 1) : 0u;
      walk = ((d.npair[(b0 << 8 | b1) >> 5] >> (b1 & 31u)) & 1u) != 0;
    }
    if (walk) {                                            // commonPrefixSearch, longest key (:218-228)
      uint32_t pos = droot;
      for (int depth = 0; depth < rem;) {
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
    while (p < L) {
      const Pfx x = prefix(p);
      if (!(x.len == 1 && x.kind != 3 && sp_byte(x, 0) == 0x20u)) break;
      p += x.consumed;
    }
  }
  if (p >= L) { out.flush(); return 0; }                   // :98-100 nothing but whitespace
  if (dummy && !suffix) emit_space(p);                     // :128
  bool is_prev_space = rm;                                 // :130
  while (p < L) {
    const Pfx x = prefix(p);
    int k = 0;
    if (x.kind != 3) while (is_prev_space && k < x.len && sp_byte(x, k) == 0x20u) ++k;      // :137-138
    if (k < x.len) {
      uint32_t last = 0;
      for (; k < x.len; ++k) {
        last = sp_byte(x, k);
        if (x.kind != 3 && last == 0x20u && (F & kNfEscapeWs)) emit_space(p);    // :143-148
        else emit(last, p);
      }
      is_prev_space = x.kind != 3 && last == 0x20u;        // :154
    }
    p += x.consumed;
    if (!rm) is_prev_space = false;                        // :160-162
  }
  int fin = L;                                             // `consumed` when the closing entry is pushed (:181)
  if (rm && wl < out.w) {                                  // :166-176 trailing space symbols
    nsp -= (out.w - wl) / (esc3 ? 3 : 1);
    fin = o_wl;
    out.truncate(wl);
  }
  if (dummy && suffix) emit_space(fin);                    // :179
  if (out.overflowed()) return -1;
  out.flush();
  if (orig_end) *orig_end = fin;
  *n_sp = nsp;
  return out.w;
}

}  // namespace spmx
#endif
