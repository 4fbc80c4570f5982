// NBestEncode on the device: unigram::Model::NBestEncode
// (src/unigram_model.cc:686-717) = Lattice::SetSentence (:114-152) + Model::PopulateNodes (:547-596) +
// Lattice::NBest (:345-515: float Viterbi for the exact heuristic, then A* from EOS), followed per result by the id
// rules of PopulateSentencePieceText (src/sentencepiece_processor.cc:547-636) and the extra options (:1019-1064).
//
// One SENTENCE PER LANE, everything in the lane's own slice of an HBM scratch buffer: the lattice (character starts,
// nodes, the nodes ending at every position), the hypotheses and the agenda.  The agenda is a binary heap that
// performs exactly the element moves of libstdc++'s push_heap / pop_heap (bits/stl_heap.h __push_heap /
// __adjust_heap): std::priority_queue's order among equal keys is part of the reference's output.  A cold path --
// no LDS, no cross-lane step, the lanes of a wave simply diverge; its input is the normalized text in device form
// (U+2581 as one byte under kNfCompressSp) written by the Normalize kernels.
#ifndef SPMX_KERNELS_NBEST_H_
#define SPMX_KERNELS_NBEST_H_

namespace spmx {

// The first launch runs every sentence with 16-bit lattice indices under these capacities; what exceeds them is set
// aside (NBestArgs::retry_list) and runs in a second launch with 32-bit indices and capacities sized for the longest
// of them (api.cc LatticeBatchHost): no length limit.
constexpr uint32_t kNbMaxLen = 1024;        // normalized (device) bytes per sentence, first launch
constexpr uint32_t kNbMaxNodes = 16384;     // lattice nodes per sentence, first launch
constexpr uint32_t kNbAgendaCap = 10000 + 512;   // :487 kMaxAgendaSize + the largest fan-in handled
constexpr uint32_t kStNbestOverflow = 1u << 5;   // status: a per-sentence capacity was exceeded

template <typename IX>
struct NbNodeT { IX pos, length, byte_begin, byte_len; int32_t id; float score, backtrace; uint32_t prev; };   // 24 / 32 bytes
struct NbHyp { uint32_t next; uint32_t node; float fx, gx; };                                      // 16 bytes

SPMX_HD inline uint64_t NbAlign16(uint64_t x) { return (x + 15u) & ~static_cast<uint64_t>(15); }
// bytes of one lane's slice; ix = sizeof(IX)
SPMX_HD inline uint64_t NbestLaneBytes(uint64_t max_len, uint64_t max_nodes, uint64_t max_hyps, uint32_t ix) {
  return NbAlign16((max_len + 2) * ix)                       // surf: character starts
       + NbAlign16(max_nodes * (4ull * ix + 16))             // nodes
       + NbAlign16((max_len + 3) * 4ull)                     // end_off: CSR offsets of the nodes ending at a position
       + NbAlign16(max_nodes * ix)                           // end_idx
       + NbAlign16((max_len + 2) * ix)                       // cursor per position
       + NbAlign16(max_hyps * sizeof(NbHyp))                 // hypotheses (mode 1: alpha, one float per position)
       + kNbAgendaCap * 4ull + 64;
}

struct NBestArgs {
  SpmxDev dev;
  const uint8_t *norm;          // packed normalized text, device form
  const uint64_t *norm_offs;    // n + 1
  uint32_t n;
  uint32_t nbest;               // 2 .. 1024 (mode 0); 1 otherwise
  uint32_t mode;                // 0 Lattice::NBest (:345-515); 1 Lattice::Sample(inv_theta) (:511-542); 2 Lattice::Viterbi
                                // (:161-198), the kOriginal encoder (:674-692)
  float inv_theta;              // mode 1
  uint64_t seed;                // mode 1: the lanes' generators are keyed by (seed, sentence index)
  uint8_t *scratch;             // [lanes of the launch][lane_bytes]
  uint64_t lane_bytes;
  uint32_t max_hyps;
  uint32_t max_len, max_nodes;  // this launch's capacities (normalized bytes, lattice nodes)
  const uint32_t *list;         // the sentences of this launch (null: 0 .. n - 1)
  uint32_t n_list;
  uint32_t live_lanes;          // lanes of the launch that own a slice (0: all)
  uint32_t *retry_list;         // sentences beyond the capacities go here (null: they fail with kStNbestOverflow)
  uint32_t *retry_count;
  unsigned long long *retry_max_len;   // the longest of them (normalized bytes)
  int32_t *arena;               // ids of all results, allocated by atomics
  int32_t *arena_nb, *arena_ne; // spans form, else null: next to every id in `arena` the byte range [nb, ne) of the normalized
                                // DEVICE text its token covers (PopulateSentencePieceText :566-617: a run of unknown
                                // characters is one token; of a character's byte-fallback pieces the last one carries
                                // the range, the others are empty at its begin); an eos is -1, -1 (the end of the INPUT, :1033-1034), a bos -2, -2 (0)
  unsigned long long *arena_head;
  uint64_t arena_cap;
  unsigned long long *res_off;  // [n][nbest] where result k of sentence s sits in the arena
  uint32_t *res_len;            // [n][nbest]
  float *res_score;             // [n][nbest]
  uint32_t *res_count;          // [n]
  uint32_t *status;
};

// std::push_heap with comp(a, b) = a->fx < b->fx
SPMX_DEVICE void nb_heap_push(uint32_t *heap, uint32_t *hn, const NbHyp *hy, uint32_t value) {
  int hole = static_cast<int>((*hn)++);
  const float fx = hy[value].fx;
  while (hole > 0) {
    const int parent = (hole - 1) / 2;
    if (!(hy[heap[parent]].fx < fx)) break;
    heap[hole] = heap[parent];
    hole = parent;
  }
  heap[hole] = value;
}
// top(), then std::pop_heap + pop_back
SPMX_DEVICE uint32_t nb_heap_pop(uint32_t *heap, uint32_t *hn, const NbHyp *hy) {
  const uint32_t top = heap[0];
  const uint32_t value = heap[*hn - 1];
  const int len = static_cast<int>(--(*hn));
  if (len == 0) return top;
  int hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (hy[heap[child]].fx < hy[heap[child - 1]].fx) --child;
    heap[hole] = heap[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    heap[hole] = heap[child - 1];
    hole = child - 1;
  }
  const float fx = hy[value].fx;
  while (hole > 0) {
    const int parent = (hole - 1) / 2;
    if (!(hy[heap[parent]].fx < fx)) break;
    heap[hole] = heap[parent];
    hole = parent;
  }
  heap[hole] = value;
  return top;
}

template <typename IX>
SPMX_DEVICE void nbest_lane(const NBestArgs &a, uint32_t s, uint8_t *mine) {
  typedef NbNodeT<IX> NbNode;
  const SpmxDev &d = a.dev;
  const uint8_t *norm = a.norm + a.norm_offs[s];
  const int size = static_cast<int>(a.norm_offs[s + 1] - a.norm_offs[s]);
  const bool bf = (d.flags & kNfByteFallback) != 0;
  const bool reverse = (d.flags & kNfReverse) != 0;
  const uint32_t spb = SpByteOf(d);
  const int n_extra = d.n_prefix + d.n_suffix;
  unsigned long long *res_off = a.res_off + static_cast<uint64_t>(s) * a.nbest;
  uint32_t *res_len = a.res_len + static_cast<uint64_t>(s) * a.nbest;
  float *res_score = a.res_score + static_cast<uint64_t>(s) * a.nbest;
  a.res_count[s] = 0;
  // one result = the extra ids around `body` ids produced by `emit`
  auto alloc = [&](int n_ids) -> int32_t * {
    const unsigned long long at = wv::atomic_add(a.arena_head, static_cast<unsigned long long>(n_ids));
    const uint32_t k = a.res_count[s];
    res_off[k] = at;
    res_len[k] = static_cast<uint32_t>(n_ids);
    if (at + static_cast<unsigned long long>(n_ids) > a.arena_cap) { wv::atomic_or(a.status, kStArenaOverflow); return nullptr; }
    return a.arena + at;
  };
  // the extra ids around a result's `body` ids (ApplyExtraOptions, sentencepiece_processor.cc:1019-1064); returns
  // where the body goes
  auto put_extras = [&](int32_t *dst, int body) -> int32_t * {
    for (int x = 0; x < d.n_prefix; ++x) dst[x] = d.prefix_ids[x];
    for (int x = 0; x < d.n_suffix; ++x) dst[d.n_prefix + body + x] = d.suffix_ids[x];
    if (a.arena_nb) {
      int32_t *nb = a.arena_nb + (dst - a.arena), *ne = a.arena_ne + (dst - a.arena);
      for (int x = 0; x < d.n_prefix; ++x) nb[x] = ne[x] = ((d.extra_eos >> x) & 1u) ? -1 : -2;                       // (reverse after eos puts an eos in front)
      for (int x = 0; x < d.n_suffix; ++x) nb[d.n_prefix + body + x] = ne[d.n_prefix + body + x] = ((d.extra_eos >> (kMaxExtra + x)) & 1u) ? -1 : -2;
    }
    return dst + d.n_prefix;
  };
  if (size == 0) {                               // unigram_model.cc:688-690: one empty result, score 0
    int32_t *dst = alloc(n_extra);
    if (dst) put_extras(dst, 0);
    res_score[0] = 0.f;
    a.res_count[s] = 1;
    return;
  }
  // beyond this launch's capacities: to the second launch, or a failure when this is it
  auto set_aside = [&]() {
    if (a.retry_list) {
      a.retry_list[wv::atomic_add(a.retry_count, 1u)] = s;
      wv::atomic_max(a.retry_max_len, static_cast<unsigned long long>(size));
    } else {
      wv::atomic_or(a.status, kStNbestOverflow);
    }
  };
  if (static_cast<uint32_t>(size) > a.max_len) { set_aside(); return; }
  const uint64_t ML = a.max_len, MN = a.max_nodes;
  IX *surf = reinterpret_cast<IX *>(mine);
  NbNode *nodes = reinterpret_cast<NbNode *>(mine + NbAlign16((ML + 2) * sizeof(IX)));
  uint32_t *end_off = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(nodes) + NbAlign16(MN * sizeof(NbNode)));
  IX *end_idx = reinterpret_cast<IX *>(reinterpret_cast<uint8_t *>(end_off) + NbAlign16((ML + 3) * 4));
  IX *cursor = reinterpret_cast<IX *>(reinterpret_cast<uint8_t *>(end_idx) + NbAlign16(MN * sizeof(IX)));
  NbHyp *hy = reinterpret_cast<NbHyp *>(reinterpret_cast<uint8_t *>(cursor) + NbAlign16((ML + 2) * sizeof(IX)));
  uint32_t *heap = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(hy) + NbAlign16(static_cast<uint64_t>(a.max_hyps) * sizeof(NbHyp)));
  // ---- SetSentence (:114-152): character starts ----
  int len = 0;
  for (int p = 0; p < size;) {
    surf[len++] = static_cast<IX>(p);
    const uint32_t c = norm[p];
    int mb = c == spb ? 1 : OneCharLenDev(c);
    if (mb > size - p) mb = size - p;
    p += mb;
  }
  surf[len] = static_cast<IX>(size);
  // ---- PopulateNodes (:547-596).  Nodes 0 = BOS, 1 = EOS, then in insertion order (begin position, then length):
  // the order of end_nodes(pos) that Viterbi's "first best wins" and the A* expansion follow ----
  uint32_t n_nodes = 2;
  nodes[0] = NbNode{0, 0, 0, 0, -1, 0.f, 0.f, 0u};
  nodes[1] = NbNode{static_cast<IX>(len), 0, 0, 0, -1, 0.f, 0.f, 0u};
  const float unk_score = d.unk_score;                         // :555 min_score() - kUnkPenalty
  const uint32_t root = d.ptrie[0].x >> kDatBaseShiftDev;
  bool over = false;
  for (int bp = 0; bp < len && !over; ++bp) {
    const int b0 = surf[bp];
    bool single = false;
    uint32_t nb = root;
    int pos_c = bp;                                            // get_chars_length (:548-552), monotone in the key length
    for (int kp = b0; kp < size;) {                            // commonPrefixSearch: matches in increasing length
      const uint32_t c = norm[kp];
      const U4 u = d.ptrie[nb ^ c];
      if ((u.x & 0x1FFu) != (0x100u | c)) break;
      ++kp;
      nb = u.x >> kDatBaseShiftDev;
      if (!(u.x & kDatTerminalDev)) continue;
      while (surf[pos_c] < kp) ++pos_c;
      const int length = pos_c - bp;
      if (u.y & kPtUnused) continue;                           // :576
      if (n_nodes >= a.max_nodes) { over = true; break; }
      float sc = wv::bits_to_float(u.z);
      if (u.y & kPtUserDefined) sc = static_cast<float>(static_cast<double>(static_cast<float>(length) * d.max_score) - 0.1);   // :580
      nodes[n_nodes++] = NbNode{static_cast<IX>(bp), static_cast<IX>(length), static_cast<IX>(b0),
                                static_cast<IX>(surf[bp + length] - b0), static_cast<int32_t>(u.y & kPtIdMask), sc, 0.f, 0u};
      if (length == 1) single = true;
    }
    if (!single && !over) {                                    // :589-593 the UNK node
      if (n_nodes >= a.max_nodes) { over = true; break; }
      nodes[n_nodes++] = NbNode{static_cast<IX>(bp), 1, static_cast<IX>(b0), static_cast<IX>(surf[bp + 1] - b0),
                                d.unk_id, unk_score, 0.f, 0u};
    }
  }
  if (over) { set_aside(); return; }
  // end_nodes as CSR over positions (insertion order within a position): BOS ends at 0
  for (int p = 0; p <= len + 1; ++p) end_off[p] = 0;
  end_off[0 + 1] = 1;
  for (uint32_t i = 2; i < n_nodes; ++i) ++end_off[nodes[i].pos + nodes[i].length + 1];
  for (int p = 0; p <= len; ++p) end_off[p + 1] += end_off[p];
  for (int p = 0; p <= len; ++p) cursor[p] = 0;
  end_idx[end_off[0] + cursor[0]++] = 0;
  for (uint32_t i = 2; i < n_nodes; ++i) {
    const int e = nodes[i].pos + nodes[i].length;
    end_idx[end_off[e] + cursor[e]++] = static_cast<IX>(i);
  }
  // ---- Viterbi (:167-198): backtrace_score of every node, begin positions in order (nodes are sorted by pos) ----
  auto best_into = [&](int pos, float score, uint32_t *prev) -> float {      // first best left node wins (:171)
    float best = 0.f;
    bool have = false;
    for (uint32_t l = end_off[pos]; l < end_off[pos + 1]; ++l) {
      const float sc = nodes[end_idx[l]].backtrace + score;
      if (!have || sc > best) { best = sc; have = true; *prev = end_idx[l]; }
    }
    return best;
  };
  if (a.mode != 1u) {
    for (uint32_t i = 2; i < n_nodes; ++i) nodes[i].backtrace = best_into(nodes[i].pos, nodes[i].score, &nodes[i].prev);
    nodes[1].backtrace = best_into(len, 0.f, &nodes[1].prev);
  }
  // ids of a path given as a chain of node indices, left to right through `next_of` (PopulateSentencePieceText
  // :581-613: a run of unknown pieces yields one id, byte fallback one id per byte), twice: count, then write
  // the ids of ONE lattice node of a path (dst null: count only).  k: ids so far; last: where the latest token that
  // is not a byte-fallback piece went (a run of unknown pieces extends it)
  auto put_node = [&](const NbNode &x, int32_t *dst, int body, int &k, bool &prev_unk, int &last) {
    const bool unk = x.id == d.unk_id;
    const int32_t nb0 = static_cast<int32_t>(x.byte_begin), ne0 = static_cast<int32_t>(x.byte_begin) + static_cast<int32_t>(x.byte_len);
    int32_t *snb = dst && a.arena_nb ? a.arena_nb + (dst - a.arena) : nullptr;
    int32_t *sne = dst && a.arena_nb ? a.arena_ne + (dst - a.arena) : nullptr;
    if (unk && bf) {
      int total = 0, j = 0;
      for (uint32_t t = 0; t < x.byte_len; ++t) total += norm[x.byte_begin + t] == spb ? 3 : 1;
      for (uint32_t t = 0; t < x.byte_len; ++t) {
        const uint32_t b = norm[x.byte_begin + t];
        const int nbt = b == spb ? 3 : 1;
        for (int y = 0; y < nbt; ++y, ++j) {
          const int at = reverse ? body - 1 - k : k;
          if (dst) dst[at] = d.byte_ids[b == spb ? (y == 0 ? 0xE2u : (y == 1 ? 0x96u : 0x81u)) : b];
          if (snb) { snb[at] = nb0; sne[at] = j == total - 1 ? ne0 : nb0; }       // :595-606
          ++k;
        }
      }
    } else if (!(unk && prev_unk)) {
      const int at = reverse ? body - 1 - k : k;
      if (dst) dst[at] = x.id;
      if (snb) { snb[at] = nb0; sne[at] = ne0; }
      last = at;
      ++k;
    } else if (sne) {
      sne[last] = ne0;                                                             // :612-616 the run's token grows
    }
    prev_unk = unk;
  };
  // ids of a path given as a chain of node indices, left to right through `next_of` (PopulateSentencePieceText
  // :581-613: a run of unknown pieces yields one id, byte fallback one id per byte), twice: count, then write
  auto emit_path = [&](auto first_of, auto next_of, auto done_of, float score) -> bool {
    int body = 0;
    for (int pass = 0; pass < 2; ++pass) {
      int32_t *dst = nullptr;
      if (pass == 1) {
        dst = alloc(body + n_extra);
        if (!dst) return false;
        dst = put_extras(dst, body);
      }
      int k = 0, last = 0;
      bool prev_unk = false;
      for (uint32_t h = first_of(); !done_of(h); h = next_of(h)) put_node(nodes[h], dst, body, k, prev_unk, last);
      body = k;
    }
    res_score[a.res_count[s]] = score;
    a.res_count[s] = a.res_count[s] + 1;
    return true;
  };
  if (a.mode == 2u) {
    // ---- Lattice::Viterbi (:161-198): the path is the prev chain from EOS; emitted left to right by reversing it in
    // place (prev -> next) ----
    uint32_t nxt = 1u, cur = nodes[1].prev;             // walk from EOS's prev towards BOS, turning the links around
    while (cur != 0u) {
      const uint32_t pv = nodes[cur].prev;
      nodes[cur].prev = nxt;
      nxt = cur;
      cur = pv;
    }
    emit_path([&] { return nxt; }, [&](uint32_t h) { return nodes[h].prev; }, [&](uint32_t h) { return h == 1u; }, nodes[1].backtrace);
    return;
  }
  if (a.mode == 1u) {
    // ---- Lattice::Sample(inv_theta) (:511-542): ForwardAlgorithm (:200-216) -- alpha of a node depends on its begin
    // position only: kept per position in the (idle) hypothesis slice -- then one left node after another from EOS,
    // each drawn with probability exp(alpha[l] + inv_theta * score[l] - Z) (std::discrete_distribution's weights) ----
    float *alpha = reinterpret_cast<float *>(hy);
    auto lse = [](float x, float y, bool init) -> float {                   // LogSumExp (:47-59)
      if (init) return y;
      const float vmin = x < y ? x : y, vmax = x < y ? y : x;
      if (vmax > vmin + 50.f) return vmax;
      return static_cast<float>(static_cast<double>(vmax) + log(exp(static_cast<double>(vmin - vmax)) + 1.0));
    };
    alpha[0] = 0.f;
    for (int pos = 1; pos <= len; ++pos) {
      float acc = 0.f;
      for (uint32_t l = end_off[pos]; l < end_off[pos + 1]; ++l) {
        const NbNode &ln = nodes[end_idx[l]];
        acc = lse(acc, a.inv_theta * ln.score + alpha[ln.pos], l == end_off[pos]);
      }
      alpha[pos] = acc;
    }
    // a generator per sentence: splitmix64 over (seed, sentence); 53 random bits per draw
    unsigned long long st = a.seed * 0x9E3779B97F4A7C15ull + (static_cast<unsigned long long>(s) + 1ull) * 0xD1B54A32D192ED03ull;
    auto uniform = [&]() -> double {
      st += 0x9E3779B97F4A7C15ull;
      unsigned long long z = st;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);
    };
    float Z = alpha[len];
    uint32_t nxt = 1u;                                  // the chain is built right to left: prev = the node to the right
    int pos = len;
    for (;;) {
      double sum = 0.0;
      for (uint32_t l = end_off[pos]; l < end_off[pos + 1]; ++l) {
        const NbNode &ln = nodes[end_idx[l]];
        sum += exp(static_cast<double>(alpha[ln.pos] + a.inv_theta * ln.score - Z));
      }
      double r = uniform() * sum;
      uint32_t pick = end_idx[end_off[pos + 1] - 1];
      for (uint32_t l = end_off[pos]; l < end_off[pos + 1]; ++l) {
        const NbNode &ln = nodes[end_idx[l]];
        r -= exp(static_cast<double>(alpha[ln.pos] + a.inv_theta * ln.score - Z));
        if (r < 0.0) { pick = end_idx[l]; break; }
      }
      if (pick == 0u) break;                            // reached BOS
      nodes[pick].prev = nxt;
      nxt = pick;
      pos = nodes[pick].pos;
      Z = alpha[pos];
    }
    emit_path([&] { return nxt; }, [&](uint32_t h) { return nodes[h].prev; }, [&](uint32_t h) { return h == 1u; }, 0.f);
    return;
  }
  // ---- A* (:375-515) ----
  uint32_t n_hyp = 0, hn = 0;
  hy[0] = NbHyp{0xFFFFFFFFu, 1u, nodes[1].backtrace, 0.f};
  n_hyp = 1;
  nb_heap_push(heap, &hn, hy, 0);
  uint32_t n_res = 0;
  while (hn > 0) {
    const uint32_t top = nb_heap_pop(heap, &hn, hy);
    const uint32_t node = hy[top].node;
    if (node == 0) {                                           // reached BOS: a complete path
      // ids of the path, left to right (PopulateSentencePieceText :581-613)
      int body = 0;
      for (int pass = 0; pass < 2; ++pass) {
        int32_t *dst = nullptr;
        if (pass == 1) {
          dst = alloc(body + n_extra);
          if (!dst) return;
          dst = put_extras(dst, body);
        }
        int k = 0, last = 0;
        bool prev_unk = false;
        for (uint32_t h = hy[top].next; hy[h].next != 0xFFFFFFFFu; h = hy[h].next) put_node(nodes[hy[h].node], dst, body, k, prev_unk, last);
        body = k;
      }
      res_score[n_res] = hy[top].fx;
      a.res_count[s] = ++n_res;
      if (n_res == a.nbest) break;
      continue;
    }
    const int pos = nodes[node].pos;
    const uint32_t fan = end_off[pos + 1] - end_off[pos];
    if (hn + fan > kNbAgendaCap) { wv::atomic_or(a.status, kStNbestOverflow); return; }
    if (n_hyp + fan > a.max_hyps) { a.res_count[s] = 0; set_aside(); return; }      // again, with a larger hypothesis slice
    const float gx = hy[top].gx;
    for (uint32_t l = end_off[pos]; l < end_off[pos + 1]; ++l) {
      const uint32_t ln = end_idx[l];
      hy[n_hyp] = NbHyp{top, ln, nodes[ln].backtrace + gx, nodes[ln].score + gx};
      nb_heap_push(heap, &hn, hy, n_hyp);
      ++n_hyp;
    }
    if (hn >= 10000u) {                                        // :487-514 keep the best min(512, 10 nbest)
      const uint32_t keep = a.nbest * 10u < 512u ? a.nbest * 10u : 512u;
      // pop the best `keep` into the (free) tail of the hypothesis slice, then rebuild the heap from them in that order
      if (n_hyp + keep > a.max_hyps) { a.res_count[s] = 0; set_aside(); return; }
      uint32_t *tmp = reinterpret_cast<uint32_t *>(hy + n_hyp);
      for (uint32_t i = 0; i < keep; ++i) tmp[i] = nb_heap_pop(heap, &hn, hy);
      hn = 0;
      for (uint32_t i = 0; i < keep; ++i) nb_heap_push(heap, &hn, hy, tmp[i]);
    }
  }
}

// Persistent body: lane l of wave w takes sentences w * 64 + l, + lanes of the launch, ...
template <typename IX>
SPMX_DEVICE void nbest_block(const NBestArgs &a) {
  const uint32_t lane_id = static_cast<uint32_t>(wv::block_id() * wv::waves_per_block() + wv::wave_in_block()) * 64u +
                           static_cast<uint32_t>(wv::lane());
  const uint32_t lanes_all = static_cast<uint32_t>(wv::grid_size() * wv::waves_per_block()) * 64u;
  uint8_t *mine = a.scratch + static_cast<uint64_t>(lane_id) * a.lane_bytes;
  const uint32_t count = a.list ? a.n_list : a.n;
  if (a.live_lanes) { if (lane_id >= a.live_lanes) return; }
  const uint32_t lanes = a.live_lanes ? a.live_lanes : lanes_all;
  for (uint32_t i = lane_id; i < count; i += lanes) nbest_lane<IX>(a, a.list ? a.list[i] : i, mine);
}

}  // namespace spmx
#endif
