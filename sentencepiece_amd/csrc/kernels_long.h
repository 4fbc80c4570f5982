// The long form: ONE SENTENCE PER LANE with every per-sentence structure in the lane's own slice of an HBM pool,
// sized from the sentence itself -- so nothing here has a length limit.  It takes what the fast forms set aside:
//
//   bpe_long_block     bpe::Model::SampleEncode(alpha = 0) (src/bpe_model.cc:38-203) with the reference's own data
//                      structures: a doubly linked symbol list and a binary-heap agenda of (score, left) with lazy
//                      deletion (:142-173), rev_merge as a hash table (:103-106, :175-200).  For sentences of models
//                      that cannot be segmented word by word (pieces spanning words, user-defined symbols,
//                      whitespace-as-suffix) beyond the sentence-per-wave form's LDS, for words longer than the lane
//                      form's slots (URLs, hashes), for sentences with more UNUSED merges than the LDS table holds.
//   norm_long_block    Normalizer::Normalize (src/normalizer.cc:71-186) + norm_to_orig for sentences beyond the
//                      position-parallel normalizer's LDS staging (spmx_normalize_batch), count / write passes.
//   align_long_block   token begins -> input byte ranges (kernels_align.h) for the same sentences.
//
// A cold path: no LDS beyond the normalizer's raw-text windows, no cross-lane step, the lanes of a wave simply diverge.
// Work is list-driven (device-side lists filled by the fast kernels or by classify); slices come from a bump
// allocator, and a sentence that finds the pool exhausted is put on a retry list (the host grows the pool and
// launches again): nothing fails for its size short of the device's memory.
#ifndef SPMX_KERNELS_LONG_H_
#define SPMX_KERNELS_LONG_H_

namespace spmx {

struct LongArgs {
  SpmxDev dev;
  const uint8_t *text;          // packed sentences
  const uint64_t *offs;         // n + 1
  const uint32_t *list;         // sentences to take
  const uint32_t *list_count;
  uint32_t *retry_list;         // sentences that found the pool exhausted (next launch's list)
  uint32_t *retry_count;
  uint8_t *pool;                // slices
  unsigned long long *pool_head;   // bytes asked for so far (keeps counting past pool_cap: the size the host needs)
  uint64_t pool_cap;
  int32_t *arena;               // as EncodeArgs
  unsigned long long *arena_head;
  uint64_t arena_cap;
  uint64_t *tmp_off;
  uint32_t *counts;
  uint8_t *sent_status;
  uint32_t *status;
  SideLists *side;
  int32_t *arena_tb;            // spans form, else null
  uint32_t stack_cap;           // entries of the resegmentation stack (> longest piece in characters)
  float dropout;                // BPE-dropout (src/bpe_model.cc:131-156): a valid merge is skipped with this probability
  uint64_t seed;                // ... by a generator keyed by (seed, sentence index)
  unsigned long long *stats;    // (wave-cooperative unigram form, nullable) kStatsPerClass words as EncodeArgs::stats
};

SPMX_HD inline uint64_t Align16(uint64_t x) { return (x + 15u) & ~static_cast<uint64_t>(15); }

// ---- agenda: binary max-heap of {score bits, left, size, merged symbol} ordered by (score, then smaller left)
// (src/bpe_model.cc:53-56).  Entries with equal (score, left) are interchangeable: at most one of them is live (:147-151).
SPMX_DEVICE bool agenda_below(const U4 &a, const U4 &b) {          // a pops after b
  const float sa = wv::bits_to_float(a.x), sb = wv::bits_to_float(b.x);
  return sa < sb || (sa == sb && a.y > b.y);
}
SPMX_DEVICE void agenda_push(U4 *heap, uint32_t *hn, const U4 &e) {
  uint32_t hole = (*hn)++;
  while (hole > 0) {
    const uint32_t parent = (hole - 1) / 2;
    const U4 pe = heap[parent];
    if (!agenda_below(pe, e)) break;
    heap[hole] = pe;
    hole = parent;
  }
  heap[hole] = e;
}
SPMX_DEVICE U4 agenda_pop(U4 *heap, uint32_t *hn) {
  const U4 top = heap[0];
  const uint32_t n = --(*hn);
  if (n == 0) return top;
  const U4 e = heap[n];
  uint32_t hole = 0;
  for (;;) {
    uint32_t child = 2 * hole + 1;
    if (child >= n) break;
    U4 ce = heap[child];
    if (child + 1 < n) {
      const U4 ce2 = heap[child + 1];
      if (agenda_below(ce, ce2)) { ce = ce2; ++child; }
    }
    if (!agenda_below(e, ce)) break;
    heap[hole] = ce;
    hole = child;
  }
  heap[hole] = e;
  return top;
}

// rev_merge[piece] = (left, right) (:103-106), last registration wins: open addressing on the merged symbol
SPMX_DEVICE void rev_store(uint32_t *tab, uint32_t mask, uint32_t merged, uint32_t l, uint32_t r) {
  uint32_t s = (merged * 0x9E3779B1u) & mask;
  for (;;) {
    const uint32_t k = tab[3 * s];
    if (k == kSymNone || k == merged) { tab[3 * s] = merged; tab[3 * s + 1] = l; tab[3 * s + 2] = r; return; }
    s = (s + 1) & mask;
  }
}
SPMX_DEVICE bool rev_find(const uint32_t *tab, uint32_t mask, uint32_t merged, uint32_t *l, uint32_t *r) {
  uint32_t s = (merged * 0x9E3779B1u) & mask;
  for (;;) {
    const uint32_t k = tab[3 * s];
    if (k == kSymNone) return false;
    if (k == merged) { *l = tab[3 * s + 1]; *r = tab[3 * s + 2]; return true; }
    s = (s + 1) & mask;
  }
}

// bytes of the slice for a sentence of nlen normalized bytes
SPMX_HD inline uint64_t BpeLongSliceBytes(uint32_t nlen, bool unused, uint32_t stack_cap, uint32_t *rev_mask) {
  uint64_t b = Align16(static_cast<uint64_t>(nlen) + 16);          // normalized text
  b += 3 * Align16((static_cast<uint64_t>(nlen) + 2) * 4);         // sym, next, prev (by byte position)
  b += Align16((3ull * nlen + 4) * 16);                            // agenda: <= n - 1 seeds + 2 per merge
  uint32_t mask = 0;
  if (unused) {                                                    // rev_merge: one registration per agenda entry at most
    uint64_t cap = 16;
    while (cap < 2 * (3ull * nlen + 4)) cap <<= 1;
    mask = static_cast<uint32_t>(cap - 1);
    b += Align16(cap * 12) + Align16(static_cast<uint64_t>(stack_cap) * 4);
  }
  if (rev_mask) *rev_mask = mask;
  return b;
}

constexpr uint32_t kLongDead = 0xFFFFFFFFu;      // sym[]: not (or no longer) the start of a symbol (= kSsDead)

SPMX_DEVICE void bpe_long_lane(const LongArgs &a, uint32_t sid, uint8_t *rawwin) {
  const SpmxDev &d = a.dev;
  const uint64_t beg = a.offs[sid];
  const uint64_t L64 = a.offs[sid + 1] - beg;
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const bool has_uds = (d.flags & kNfHasUserDefined) != 0;
  const bool track_unused = (d.flags & kNfHasUnused) != 0;
  const uint32_t spb = SpByteOf(d);
  const int n_extra = d.n_prefix + d.n_suffix;
  auto fail = [&](uint32_t code) {
    a.counts[sid] = 0; a.tmp_off[sid] = 0; a.sent_status[sid] = static_cast<uint8_t>(code);
    wv::atomic_add(&a.side->n_failed, 1ull);
  };
  if (L64 > 0x7FFFFFF0ull / (d.expand_max ? d.expand_max : 1u)) { fail(kSsOutOfRange); return; }
  const int L = static_cast<int>(L64);
  // ---- normalized length first (a count-only pass), then a slice of exactly the size it takes ----
  int nsp = 0;
  FlatSink cs{nullptr, nullptr, 0};
  const int nlen = norm_lane_any(d, a.text, beg, L, cs, rawwin, &nsp);
  int32_t *dst = nullptr;
  int32_t *tdst = nullptr;
  auto alloc_ids = [&](int n_ids) -> bool {
    const unsigned long long at = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(n_ids + n_extra));
    a.tmp_off[sid] = at;
    a.counts[sid] = static_cast<uint32_t>(n_ids + n_extra);
    if (at + static_cast<unsigned long long>(n_ids + n_extra) > a.arena_cap) { wv::atomic_or(a.status, kStArenaOverflow); return false; }
    int32_t *p = a.arena + at;
    for (int x = 0; x < d.n_prefix; ++x) p[x] = d.prefix_ids[x];
    for (int x = 0; x < d.n_suffix; ++x) p[d.n_prefix + n_ids + x] = d.suffix_ids[x];
    dst = p + d.n_prefix;
    if (a.arena_tb) tdst = a.arena_tb + at + d.n_prefix;
    return true;
  };
  if (nlen == 0) { alloc_ids(0); return; }
  uint32_t rev_mask = 0;
  const uint64_t need = BpeLongSliceBytes(static_cast<uint32_t>(nlen), track_unused, a.stack_cap, &rev_mask);
  const unsigned long long at = wv::atomic_add(a.pool_head, static_cast<unsigned long long>(need));
  if (at + need > a.pool_cap) {                                    // the host grows the pool and launches again
    a.retry_list[wv::atomic_add(a.retry_count, 1u)] = sid;
    a.counts[sid] = 0u;
    return;
  }
  uint8_t *mine = a.pool + at;
  uint8_t *norm = mine;
  const uint64_t n4 = Align16((static_cast<uint64_t>(nlen) + 2) * 4);
  uint32_t *sym = reinterpret_cast<uint32_t *>(mine + Align16(static_cast<uint64_t>(nlen) + 16));
  uint32_t *nxt = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(sym) + n4);
  uint32_t *prv = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(nxt) + n4);
  U4 *heap = reinterpret_cast<U4 *>(reinterpret_cast<uint8_t *>(prv) + n4);
  uint32_t *rev = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(heap) + Align16((3ull * nlen + 4) * 16));
  uint32_t *stack = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(rev) + Align16((static_cast<uint64_t>(rev_mask) + 1) * 12));
  {
    FlatSink ws{norm, nullptr, nlen};
    int nsp2 = 0;
    norm_lane_any(d, a.text, beg, L, ws, rawwin, &nsp2);
  }
  if (track_unused) for (uint32_t i = 0; i <= rev_mask; ++i) rev[3 * i] = kSymNone;
  const uint32_t N = static_cast<uint32_t>(nlen);
  // ---- split into symbols (:109-120): PrefixMatch = longest user-defined symbol, else one character ----
  const uint32_t uroot = has_uds ? (d.utrie[0].x >> kDatBaseShiftDev) : 0u;
  uint32_t last = kLongDead;
  for (uint32_t p = 0; p < N;) {
    uint32_t step = 0, s = 0;
    if (has_uds) {
      uint32_t nb = uroot, uds_len = 0, uds_id = 0;
      for (uint32_t q = p; q < N;) {
        const uint32_t c = norm[q];
        const U2 u = d.utrie[nb ^ c];
        if ((u.x & 0x1FFu) != (0x100u | c)) break;
        ++q;
        nb = u.x >> kDatBaseShiftDev;
        if (u.x & kDatTerminalDev) { uds_len = q - p; uds_id = u.y; }
      }
      if (uds_len) { step = uds_len; s = kSsFrozen | uds_id; }
    }
    if (!step) {
      const uint32_t c0 = norm[p];
      step = c0 == spb ? 1u : static_cast<uint32_t>(OneCharLenDev(c0));
      if (step > N - p) step = N - p;
      uint32_t bytes = 0;
      for (uint32_t k = 0; k < step; ++k) bytes |= static_cast<uint32_t>(norm[p + k]) << (8 * k);
      s = char_lookup(d, bytes, step);
    }
    sym[p] = s;
    for (uint32_t k = 1; k < step; ++k) sym[p + k] = kLongDead;
    prv[p] = last;
    if (last != kLongDead) nxt[last] = p;
    last = p;
    p += step;
  }
  nxt[last] = N;
  uint32_t hn = 0;
  auto add_pair = [&](uint32_t l) {                                 // MaybeAddNewSymbolPair (:85-107)
    if (l == kLongDead) return;
    const uint32_t r = nxt[l];
    if (r >= N) return;
    uint32_t m = 0;
    float sc = 0.f;
    if (!pair_lookup(d, sym[l], sym[r], &m, &sc)) return;
    const U4 e{wv::float_to_bits(sc), l, nxt[r] - l, m};
    agenda_push(heap, &hn, e);
    if (track_unused && (d.sym_final[m] & kSfUnused)) rev_store(rev, rev_mask, m, sym[l], sym[r]);   // :103-106
  };
  for (uint32_t p = 0; p < N; p = nxt[p]) add_pair(p);               // :127-129
  // ---- main loop (:142-173) ----
  unsigned long long rng = a.seed * 0x9E3779B97F4A7C15ull + (static_cast<unsigned long long>(sid) + 1ull) * 0xD1B54A32D192ED03ull;
  while (hn > 0) {
    const U4 top = agenda_pop(heap, &hn);
    const uint32_t l = top.y;
    if (sym[l] == kLongDead) continue;                               // :147-151 stale entries
    const uint32_t r = nxt[l];
    if (r >= N || nxt[r] - l != top.z) continue;
    if (a.dropout > 0.f) {                                           // skip_merge (:131-138, :153-156)
      bool skip = a.dropout >= 1.f;
      if (!skip) {
        rng += 0x9E3779B97F4A7C15ull;
        unsigned long long z = rng;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        skip = static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0) < static_cast<double>(a.dropout);
      }
      if (skip) continue;
    }
    sym[l] = top.w;                                                  // :159-168
    const uint32_t nr = nxt[r];
    nxt[l] = nr;
    if (nr < N) prv[nr] = l;
    sym[r] = kLongDead;
    add_pair(prv[l]);                                                // :171-172
    add_pair(l);
  }
  // ---- output pieces (:175-200) and their ids (sentencepiece_processor.cc:581-613), twice: count, then write ----
  bool bad = false;
  int n_ids = 0;
  for (int pass = 0; pass < 2 && !bad; ++pass) {
    int k = 0;
    bool right_unk = false;
    // one output piece: symbol s (kSsUnknown: a character without one) over bytes [off, off + len)
    auto piece = [&](uint32_t s, uint32_t off, uint32_t len) {
      uint32_t f = static_cast<uint32_t>(d.unk_id);
      if (s != kSsUnknown) {
        f = d.sym_final[s];
        if (f & kSfControl) { bad = true; return; }
        f &= kSfIdMask;
      }
      if (static_cast<int32_t>(f) == d.unk_id) {
        if (bf) {                                                    // one BYTE id per byte of the unknown piece
          for (uint32_t x = 0; x < len; ++x) {
            const uint32_t b = norm[off + x];
            const int nb = b == spb ? 3 : 1;
            for (int y = 0; y < nb; ++y) {
              if (pass) {
                const uint32_t byte = b == spb ? (y == 0 ? 0xE2u : (y == 1 ? 0x96u : 0x81u)) : b;
                dst[reverse ? n_ids - 1 - k : k] = d.byte_ids[byte];
                if (tdst) tdst[reverse ? n_ids - 1 - k : k] = static_cast<int32_t>(off);
              }
              ++k;
            }
          }
        } else if (!right_unk) {                                     // a run of unknown pieces yields one id
          if (pass) { dst[reverse ? n_ids - 1 - k : k] = d.unk_id; if (tdst) tdst[reverse ? n_ids - 1 - k : k] = static_cast<int32_t>(off); }
          ++k;
        }
        right_unk = true;
      } else {
        if (pass) { dst[reverse ? n_ids - 1 - k : k] = static_cast<int32_t>(f); if (tdst) tdst[reverse ? n_ids - 1 - k : k] = static_cast<int32_t>(off); }
        ++k;
        right_unk = false;
      }
    };
    for (uint32_t p = 0; p < N && !bad; p = nxt[p]) {
      const uint32_t top = sym[p];
      const uint32_t end = nxt[p];
      if (top == kSsUnknown) { piece(kSsUnknown, p, end - p); continue; }
      const uint32_t sym0 = top & ~kSsFrozen;
      if (!track_unused || !(d.sym_final[sym0] & kSfUnused)) { piece(sym0, p, end - p); continue; }
      // resegment(w) (:176-193), depth-first, left part first
      uint32_t sp = 0, pos = p;
      stack[sp++] = sym0;
      while (sp > 0 && !bad) {
        const uint32_t s = stack[--sp];
        uint32_t l = 0, r = 0;
        if ((d.sym_final[s] & kSfUnused) && rev_find(rev, rev_mask, s, &l, &r) && sp + 2 <= a.stack_cap) {
          stack[sp++] = r;
          stack[sp++] = l;
        } else {
          const uint32_t len = d.sym_len[s];
          piece(s, pos, len);
          pos += len;
        }
      }
    }
    if (pass == 0) {
      n_ids = k;
      if (bad || !alloc_ids(n_ids)) break;
    }
  }
  if (bad) { fail(kSsInternal); }        // a control piece among the symbols: "all normalized characters are not consumed."
}

// Persistent body: lane l of wave w takes list entries w * 64 + l, + lanes of the launch, ...
SPMX_DEVICE void bpe_long_block(const LongArgs &a, unsigned char *smem) {
  const uint32_t lane_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block()) * 64u +
                           static_cast<uint32_t>(wv::lane());
  const uint32_t lanes = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block()) * 64u;
  uint8_t *rawwin = smem + (static_cast<uint32_t>(wv::wave_in_block()) * 64u + static_cast<uint32_t>(wv::lane())) * kRawWinBytes;
  const uint32_t count = *a.list_count;
  for (uint32_t i = lane_id; i < count; i += lanes) bpe_long_lane(a, a.list[i], rawwin);
}

// ---- Normalize(input, &normalized, &norm_to_orig) for sentences beyond the staged kernels (kernels_normalize.h) -----
// The text leaves in the reference's form: under kNfCompressSp every one-byte space symbol is written as the three
// bytes of U+2581 (unless the caller asked for the device form).  dst == null only counts.
struct ExpandSink {
  uint8_t *dst;
  uint32_t *o;
  bool expand;
  long long limit;    // write pass: the sentence's final length (from the count pass); nothing is stored past it --
                      // the text passes it only by trailing space symbols that are cut again (:166-176)
  int w = 0;          // device-form bytes (what norm_lane_any counts in)
  long long ow = 0;   // bytes written / counted in the output form
  SPMX_DEVICE void put(uint32_t b, int orig) {
    if (expand && b == kSpByte) {
      if (dst && ow + 3 <= limit) { dst[ow] = 0xE2; dst[ow + 1] = 0x96; dst[ow + 2] = 0x81; if (o) { o[ow] = o[ow + 1] = o[ow + 2] = static_cast<uint32_t>(orig); } }
      ow += 3;
    } else {
      if (dst && ow + 1 <= limit) { dst[ow] = static_cast<uint8_t>(b); if (o) o[ow] = static_cast<uint32_t>(orig); }
      ow += 1;
    }
    ++w;
  }
  SPMX_DEVICE void flush() {}
  // only ever cuts trailing space symbols (:166-176)
  SPMX_DEVICE void truncate(int n) { ow -= static_cast<long long>(w - n) * (expand ? 3 : 1); w = n; }
  SPMX_DEVICE bool overflowed() const { return false; }
};

template <bool WRITE>
SPMX_DEVICE void norm_long_lane(const NormalizeArgs &a, uint32_t sid, uint8_t *rawwin) {
  const SpmxDev &d = a.dev;
  const uint64_t beg = a.offs[sid];
  const uint64_t L64 = a.offs[sid + 1] - beg;
  if (L64 > 0x7FFFFFF0ull / (d.expand_max ? d.expand_max : 1u)) {      // its normalized form could pass 2^31 bytes
    if (!WRITE) { a.counts[sid] = 0; wv::atomic_or(a.status, kStTooLong); }
    return;
  }
  const bool expand = (d.flags & kNfCompressSp) != 0 && !a.device_text;
  int nsp = 0, fin = -1;
  if (!WRITE) {
    ExpandSink cs{nullptr, nullptr, expand, 0};
    norm_lane_any(d, a.text, beg, static_cast<int>(L64), cs, rawwin, &nsp);
    a.counts[sid] = static_cast<uint32_t>(cs.ow);
  } else {
    uint32_t *dn = a.n2o ? a.n2o + a.norm_offs[sid] + sid : nullptr;
    ExpandSink ws{a.norm + a.norm_offs[sid], dn, expand, static_cast<long long>(a.norm_offs[sid + 1] - a.norm_offs[sid])};
    norm_lane_any(d, a.text, beg, static_cast<int>(L64), ws, rawwin, &nsp, &fin);
    if (dn) dn[ws.ow] = fin < 0 ? kNoClosingEntry : static_cast<uint32_t>(fin);
  }
}
template <bool WRITE>
SPMX_DEVICE void norm_long_block(const NormalizeArgs &a, unsigned char *smem) {
  const uint32_t lane_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block()) * 64u +
                           static_cast<uint32_t>(wv::lane());
  const uint32_t lanes = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block()) * 64u;
  uint8_t *rawwin = smem + (static_cast<uint32_t>(wv::wave_in_block()) * 64u + static_cast<uint32_t>(wv::lane())) * kRawWinBytes;
  const uint32_t count = *a.list_count;
  for (uint32_t i = lane_id; i < count; i += lanes) norm_long_lane<WRITE>(a, a.list[i], rawwin);
}

// ---- spans form for the same sentences (kernels_align.h): token begins -> input byte ranges -------------------------
// The normalizer runs again with its norm_to_orig on; nothing is stored: tokens tile the normalized text in increasing
// order, so the sink hands every byte's origin to the tokens that begin there as the bytes go by.
struct AlignLongArgs {
  SpmxDev dev;
  const uint8_t *text;
  const uint64_t *offs;
  const uint32_t *list;
  const uint32_t *list_count;
  const uint64_t *id_offs;
  const int32_t *tok_begin;
  uint32_t *begin, *end, *nbegin, *nend;
  uint32_t *status;
};
struct AlignSink {
  const int32_t *tb;      // token begins, CSR slot of the sentence's first body token
  uint32_t *begin, *end, *nbegin, *nend;   // same slot
  int body;
  bool reverse, one;
  int w = 0, i = 0;       // device-form bytes so far; next token (text order) waiting for its begin
  int spc = 0;            // one-byte space symbols before w
  SPMX_DEVICE bool overflowed() const { return false; }
  SPMX_DEVICE int slot(int k) const { return reverse ? body - 1 - k : k; }
  SPMX_DEVICE void at(int pos, int orig) {          // the byte at device position pos has origin orig
    while (i < body && tb[slot(i)] == pos) {
      begin[slot(i)] = static_cast<uint32_t>(orig);
      if (nbegin) nbegin[slot(i)] = static_cast<uint32_t>(pos + 2 * spc);
      if (i > 0) { end[slot(i - 1)] = static_cast<uint32_t>(orig); if (nend) nend[slot(i - 1)] = static_cast<uint32_t>(pos + 2 * spc); }
      ++i;
    }
  }
  SPMX_DEVICE void put(uint32_t b, int orig) {
    at(w, orig);
    if (one && b == kSpByte) ++spc;
    ++w;
  }
  SPMX_DEVICE void flush() {}
  SPMX_DEVICE void truncate(int n) {                // only ever cuts trailing one-byte / three-byte space symbols
    if (one) spc -= w - n;
    w = n;
    while (i > 0 && tb[slot(i - 1)] >= n) --i;      // tokens that began in the cut tail wait again (whitespace as suffix)
  }
};

SPMX_DEVICE void align_long_lane(const AlignLongArgs &a, uint32_t sid, uint8_t *rawwin) {
  const SpmxDev &d = a.dev;
  const uint64_t beg = a.offs[sid];
  const uint64_t L64 = a.offs[sid + 1] - beg;
  const uint64_t ib = a.id_offs[sid];
  const int T = static_cast<int>(a.id_offs[sid + 1] - ib);
  const int body = T - d.n_prefix - d.n_suffix;
  if (body < 0) return;                            // the encode failed this sentence (its status byte says why)
  const uint32_t L = static_cast<uint32_t>(L64);
  for (int x = 0; x < d.n_prefix; ++x) {
    const uint32_t v = ((d.extra_eos >> x) & 1u) ? L : 0u;
    a.begin[ib + x] = v; a.end[ib + x] = v;
    if (a.nbegin) { a.nbegin[ib + x] = 0; a.nend[ib + x] = 0; }
  }
  for (int x = 0; x < d.n_suffix; ++x) {
    const uint32_t v = ((d.extra_eos >> (kMaxExtra + x)) & 1u) ? L : 0u;
    a.begin[ib + d.n_prefix + body + x] = v; a.end[ib + d.n_prefix + body + x] = v;
    if (a.nbegin) { a.nbegin[ib + d.n_prefix + body + x] = 0; a.nend[ib + d.n_prefix + body + x] = 0; }
  }
  if (body == 0) return;
  const uint64_t s0 = ib + static_cast<uint64_t>(d.n_prefix);
  AlignSink sk{a.tok_begin + s0, a.begin + s0, a.end + s0, a.nbegin ? a.nbegin + s0 : nullptr, a.nend ? a.nend + s0 : nullptr,
               body, (d.flags & kNfReverse) != 0, (d.flags & kNfCompressSp) != 0};
  int nsp = 0, fin = -1;
  const int nlen = norm_lane_any(d, a.text, beg, static_cast<int>(L64), sk, rawwin, &nsp, &fin);
  if (nlen < 0 || sk.i != body) { wv::atomic_or(a.status, kStInternal); return; }
  const uint32_t last = static_cast<uint32_t>(fin < 0 ? static_cast<int>(L) : fin);          // the closing entry (:181)
  sk.end[sk.slot(body - 1)] = last;
  if (sk.nend) sk.nend[sk.slot(body - 1)] = static_cast<uint32_t>(nlen + 2 * sk.spc);
}
SPMX_DEVICE void align_long_block(const AlignLongArgs &a, unsigned char *smem) {
  const uint32_t lane_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block()) * 64u +
                           static_cast<uint32_t>(wv::lane());
  const uint32_t lanes = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block()) * 64u;
  uint8_t *rawwin = smem + (static_cast<uint32_t>(wv::wave_in_block()) * 64u + static_cast<uint32_t>(wv::lane())) * kRawWinBytes;
  const uint32_t count = *a.list_count;
  for (uint32_t i = lane_id; i < count; i += lanes) align_long_lane(a, a.list[i], rawwin);
}

}  // namespace spmx
#endif
