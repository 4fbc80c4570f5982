#include "tables.h"

#include <algorithm>
#include <cstring>
#include <map>

#include "dat.h"

namespace spmx {
namespace {

// Darts unit accessors (reference: third_party/darts_clone/darts.h:50-80).
inline uint32_t DuOffset(uint32_t u) { return (u >> 10) << ((u & (1u << 9)) >> 6); }
inline uint32_t DuLabel(uint32_t u) { return u & ((1u << 31) | 0xFF); }
inline bool DuHasLeaf(uint32_t u) { return (u >> 8) & 1; }
inline uint32_t DuValue(uint32_t u) { return u & ((1u << 31) - 1); }

struct NormKey {
  bool uds = false, rule = false;
  uint32_t off = 0, len = 0;
};

// Enumerates every (key, value) of the serialized Darts trie inside a
// precompiled_charsmap (layout: src/normalizer.cc:274-309) by walking it the
// way commonPrefixSearch does (darts.h:467-513).  Also validates it: every
// reachable unit index is in range.  Darts-clone builds its array from a DAWG,
// so units are legitimately shared between keys (merged suffixes); termination
// on a malformed (cyclic) blob is guaranteed by the key-length and step caps.
Status EnumerateCharsmap(const std::string &blob, std::map<std::string, NormKey> *keys, std::string *strings) {
  if (blob.size() <= 4) return Status::Error(kInternal, "Blob for normalization rule is broken.");
  uint32_t trie_bytes = 0;
  memcpy(&trie_bytes, blob.data(), 4);
  if (trie_bytes >= blob.size()) return Status::Error(kInternal, "Trie data size exceeds the input blob size.");
  const size_t n = trie_bytes / 4;
  std::vector<uint32_t> units(n);
  memcpy(units.data(), blob.data() + 4, n * 4);
  strings->assign(blob.data() + 4 + trie_bytes, blob.size() - 4 - trie_bytes);
  if (n == 0) return Status::OK();
  uint64_t steps = 0;
  struct Frame { uint32_t pos; int next_c; };
  std::vector<Frame> stack;
  std::string key;
  stack.push_back({DuOffset(units[0]), 1});
  while (!stack.empty()) {
    Frame &f = stack.back();
    if (f.next_c > 255) {
      stack.pop_back();
      if (!key.empty()) key.pop_back();
      continue;
    }
    const int c = f.next_c++;
    const uint32_t p2 = f.pos ^ static_cast<uint32_t>(c);
    if (p2 >= n) continue;
    const uint32_t u = units[p2];
    if (DuLabel(u) != static_cast<uint32_t>(c)) continue;
    if (++steps > (1ull << 26)) return Status::Error(kInternal, "precompiled_charsmap trie is malformed (too many paths).");
    const uint32_t child = p2 ^ DuOffset(u);
    key.push_back(static_cast<char>(c));
    if (key.size() > 1024) return Status::Error(kInternal, "precompiled_charsmap trie is malformed (key too long).");
    if (DuHasLeaf(u)) {
      if (child >= n) return Status::Error(kInternal, "precompiled_charsmap trie is malformed (leaf out of range).");
      const uint32_t off = DuValue(units[child]);
      if (off >= strings->size()) return Status::Error(kInternal, "precompiled_charsmap value out of range.");
      NormKey &k = (*keys)[key];
      k.rule = true;
      k.off = off;
      k.len = static_cast<uint32_t>(strnlen(strings->data() + off, strings->size() - off));
    }
    stack.push_back({child, 1});
  }
  return Status::OK();
}

void PackTrie2(const DatTrie &d, const std::vector<uint32_t> &payload, std::vector<U2> *out) {
  out->resize(d.w0.size());
  for (size_t i = 0; i < d.w0.size(); ++i) {
    (*out)[i].x = d.w0[i];
    (*out)[i].y = d.value[i] == 0xFFFFFFFFu ? 0u : payload[d.value[i]];
  }
}

uint32_t FloatBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

bool IsCharLike(const std::string &s) {
  return !s.empty() && s.size() <= 4 && s.size() <= static_cast<size_t>(OneCharLen(static_cast<unsigned char>(s[0])));
}
uint32_t PackChar(const std::string &s) {
  uint32_t v = 0;
  for (size_t i = 0; i < s.size(); ++i) v |= static_cast<uint32_t>(static_cast<unsigned char>(s[i])) << (8 * i);
  return v;
}
uint32_t NextPow2(size_t n) { uint32_t p = 16; while (p < n) p <<= 1; return p; }

}  // namespace

Status CompileTables(const ModelData &m, HostTables *t) {
  std::string err;
  SpmxDev &sc = t->scalars;
  sc = SpmxDev{};
  sc.model_type = m.model_type;
  if (m.model_type != kUnigram && m.model_type != kBpe)
    return Status::Error(kUnimplemented, "only unigram and bpe models are on the device path");
  if (m.pieces.size() >= (1u << 30)) return Status::Error(kResourceExhausted, "vocabulary too large");
  uint32_t flags = 0;
  if (m.add_dummy_prefix) flags |= kNfAddDummyPrefix;
  if (m.remove_extra_ws) flags |= kNfRemoveExtraWs;
  if (m.escape_ws) flags |= kNfEscapeWs;
  if (m.ws_suffix) flags |= kNfWsSuffix;
  if (m.byte_fallback) flags |= kNfByteFallback;

  // ---------------------------------------------------------- normalizer ---
  std::map<std::string, NormKey> nkeys;
  std::string strings;
  if (!m.charsmap.empty()) {
    Status st = EnumerateCharsmap(m.charsmap, &nkeys, &strings);
    if (!st.ok()) return st;
  }
  std::vector<std::pair<std::string, uint32_t>> uds_keys;
  for (size_t i = 0; i < m.pieces.size(); ++i) {
    if (m.pieces[i].load_type != kUserDefined) continue;
    nkeys[m.pieces[i].piece].uds = true;
    uds_keys.emplace_back(m.pieces[i].piece, static_cast<uint32_t>(i));
    flags |= kNfHasUserDefined;
  }
  if (strings.size() >= (1u << 24)) return Status::Error(kResourceExhausted, "precompiled_charsmap strings exceed 16 MiB");
  t->nblob.assign(strings.begin(), strings.end());
  if (t->nblob.empty()) t->nblob.push_back(0);
  t->ninfo.clear();
  t->max_norm_key_len = 4;
  t->max_expansion_num = 3;
  t->max_expansion_den = 1;
  if (!nkeys.empty()) {
    std::vector<std::pair<std::string, uint32_t>> keys;
    std::vector<uint32_t> payload;
    for (const auto &kv : nkeys) {
      const NormKey &k = kv.second;
      uint32_t lead = 0, nsp = 0, f = 0;
      if (k.rule) {
        const char *s = strings.data() + k.off;
        while (lead < k.len && s[lead] == ' ') ++lead;
        for (uint32_t i = 0; i < k.len; ++i) nsp += s[i] == ' ';
        if (k.len > 0 && s[k.len - 1] == ' ') f |= kNiEndsSpace;
        if (k.len >= 4096 || lead >= 256 || nsp >= 4096)
          return Status::Error(kUnimplemented, "normalization rule replacement too long for the device path");
        const uint64_t out_bytes = k.len + (m.escape_ws ? 2ull * nsp : 0);
        if (out_bytes * t->max_expansion_den > static_cast<uint64_t>(t->max_expansion_num) * kv.first.size()) {
          t->max_expansion_num = static_cast<int>(out_bytes);
          t->max_expansion_den = static_cast<int>(kv.first.size());
        }
      }
      const uint32_t idx = static_cast<uint32_t>(t->ninfo.size());
      t->ninfo.push_back(U2{k.off << 8 | f, k.len | lead << 12 | nsp << 20});
      payload.push_back(idx | (k.uds ? kNkUds : 0) | (k.rule ? kNkRule : 0));
      keys.emplace_back(kv.first, idx);
      t->max_norm_key_len = std::max<int>(t->max_norm_key_len, static_cast<int>(kv.first.size()));
    }
    DatTrie d;
    if (!BuildDat(keys, &d, &err)) return Status::Error(kInternal, "normalizer trie: " + err);
    PackTrie2(d, payload, &t->ntrie);
    flags |= kNfHasTrie;
  } else {
    t->ntrie.assign(256, U2{0, 0});
    t->ninfo.assign(1, U2{0, 0});
  }
  {  // user-defined symbols alone, for PrefixMatch over normalized text (BPE)
    if (!uds_keys.empty()) {
      DatTrie d;
      std::vector<std::pair<std::string, uint32_t>> keys;
      std::vector<uint32_t> payload;
      for (auto &kv : uds_keys) { keys.emplace_back(kv.first, static_cast<uint32_t>(payload.size())); payload.push_back(kv.second); }
      if (!BuildDat(keys, &d, &err)) return Status::Error(kInternal, "user-defined symbol trie: " + err);
      PackTrie2(d, payload, &t->utrie);
    } else {
      t->utrie.assign(256, U2{0, 0});
    }
  }

  // ------------------------------------------------------- id post-process --
  t->byte_ids.assign(m.byte_ids, m.byte_ids + 256);
  sc.unk_id = m.unk_id;
  sc.unk_score = m.min_score - 10.0f;  // kUnkPenalty (src/unigram_model.cc:39, :955), float arithmetic
  sc.max_score = m.max_score;

  // --------------------------------------------------------------- unigram --
  t->ptrie.clear();
  t->max_piece_len = 0;
  if (m.model_type == kUnigram) {
    std::vector<std::pair<std::string, uint32_t>> keys;
    for (const auto &kv : m.pieces_map) keys.emplace_back(kv.first, static_cast<uint32_t>(kv.second));
    DatTrie d;
    if (!BuildDat(keys, &d, &err)) return Status::Error(kInternal, "piece trie: " + err);
    t->max_piece_len = d.max_key_len;
    t->max_prefixes = d.max_prefixes;
    if (d.max_key_len > kMaxPieceBytes)
      return Status::Error(kUnimplemented, "a piece is longer than 64 bytes; unsupported by the device unigram path");
    t->ptrie.resize(d.w0.size());
    for (size_t i = 0; i < d.w0.size(); ++i) {
      U4 u{d.w0[i], 0, 0, 0};
      if (d.value[i] != 0xFFFFFFFFu) {
        u.y = d.value[i];
        u.z = FloatBits(m.pieces[d.value[i]].score);
      }
      t->ptrie[i] = u;
    }
  } else {
    t->ptrie.assign(256, U4{0, 0, 0, 0});
  }

  // ------------------------------------------------------------------- BPE --
  t->chartab.clear();
  t->pairtab.clear();
  t->sym_final.clear();
  t->sym_len.clear();
  if (m.model_type == kBpe) {
    const uint32_t V = static_cast<uint32_t>(m.pieces.size());
    // symbol universe: piece ids [0, V) for strings in pieces_, then extra
    // char-like strings that are split parts of a piece or reserved names.
    std::map<std::string, uint32_t> extra;
    t->sym_final.assign(V, 0);
    t->sym_len.assign(V, 0);
    for (uint32_t i = 0; i < V; ++i) {
      // PieceToId(piece string) (src/bpe_model.cc:178, src/model_interface.cc:51-61)
      t->sym_final[i] = static_cast<uint32_t>(m.PieceToId(m.pieces[i].piece));
      if (m.pieces[i].piece.size() > 0xFFFF) return Status::Error(kUnimplemented, "piece longer than 65535 bytes");
      t->sym_len[i] = static_cast<uint16_t>(m.pieces[i].piece.size());
    }
    auto sym_of = [&](const std::string &s, bool create) -> uint32_t {
      auto it = m.pieces_map.find(s);
      if (it != m.pieces_map.end()) return static_cast<uint32_t>(it->second);
      if (!IsCharLike(s)) return kSymNone;
      auto e = extra.find(s);
      if (e != extra.end()) return e->second;
      if (!create) return kSymNone;
      const uint32_t id = V + static_cast<uint32_t>(extra.size());
      extra.emplace(s, id);
      return id;
    };
    // chars that only exist as reserved names still need their PieceToId
    for (const auto &kv : m.reserved_map)
      if (IsCharLike(kv.first)) sym_of(kv.first, true);
    struct PairEnt { uint32_t a, b, merged; float score; };
    std::vector<PairEnt> pairs;
    for (const auto &kv : m.pieces_map) {
      const std::string &p = kv.first;
      for (size_t k = 1; k < p.size(); ++k) {
        const std::string a = p.substr(0, k), b = p.substr(k);
        const bool a_ok = m.pieces_map.count(a) || IsCharLike(a);
        const bool b_ok = m.pieces_map.count(b) || IsCharLike(b);
        if (!a_ok || !b_ok) continue;
        pairs.push_back({sym_of(a, true), sym_of(b, true), static_cast<uint32_t>(kv.second), m.pieces[kv.second].score});
      }
    }
    t->sym_final.resize(V + extra.size());
    t->sym_len.resize(V + extra.size());
    for (const auto &kv : extra) {
      t->sym_final[kv.second] = static_cast<uint32_t>(m.PieceToId(kv.first));
      t->sym_len[kv.second] = static_cast<uint16_t>(kv.first.size());
    }
    // char table: every char-like string that has a symbol
    std::vector<std::pair<std::string, uint32_t>> chars;
    for (const auto &kv : m.pieces_map) if (IsCharLike(kv.first)) chars.emplace_back(kv.first, static_cast<uint32_t>(kv.second));
    for (const auto &kv : extra) chars.emplace_back(kv.first, kv.second);
    const uint32_t csz = NextPow2(chars.size() * 2 + 16);
    t->chartab.assign(csz, U4{0, 0, kSymNone, 0});
    for (const auto &kv : chars) {
      const uint32_t bytes = PackChar(kv.first), len = static_cast<uint32_t>(kv.first.size());
      uint32_t s = HashChar(bytes, len) & (csz - 1);
      while (t->chartab[s].y != 0) s = (s + 1) & (csz - 1);
      t->chartab[s] = U4{bytes, len, kv.second, 0};
    }
    sc.chartab_mask = csz - 1;
    const uint32_t psz = NextPow2(pairs.size() * 2 + 16);
    t->pairtab.assign(psz, U4{kSymNone, kSymNone, kSymNone, 0});
    for (const PairEnt &e : pairs) {
      uint32_t s = HashPair(e.a, e.b) & (psz - 1);
      while (t->pairtab[s].x != kSymNone) s = (s + 1) & (psz - 1);
      t->pairtab[s] = U4{e.a, e.b, e.merged, FloatBits(e.score)};
    }
    sc.pairtab_mask = psz - 1;
    for (uint32_t i = 0; i < t->sym_final.size(); ++i) {
      const uint32_t fid = t->sym_final[i] & kSfIdMask;
      if (m.pieces[fid].load_type == kControl || m.pieces[fid].type == kControl) t->sym_final[i] |= kSfControl;
    }
  } else {
    t->chartab.assign(16, U4{0, 0, kSymNone, 0});
    t->pairtab.assign(16, U4{kSymNone, kSymNone, kSymNone, 0});
    t->sym_final.assign(1, 0);
    t->sym_len.assign(1, 0);
  }
  sc.flags = flags;
  RefreshTypeFlags(m, t);
  return Status::OK();
}

void RefreshTypeFlags(const ModelData &m, HostTables *t) {
  bool any_unused = false;
  for (const PieceRec &p : m.pieces) any_unused |= p.type == kUnused;
  t->scalars.flags = (t->scalars.flags & ~kNfHasUnused) | (any_unused ? kNfHasUnused : 0);
  if (m.model_type == kUnigram) {
    for (U4 &u : t->ptrie) {
      if (!(u.x & kDatTerminal)) continue;
      const uint32_t id = u.y & kPtIdMask;
      const int type = m.pieces[id].type;  // IsUnusedInlined / IsUserDefinedInlined read the live type
      u.y = id | (type == kUnused ? kPtUnused : 0) | (type == kUserDefined ? kPtUserDefined : 0);
    }
  } else if (m.model_type == kBpe) {
    // IsUnusedInlined(id) on the *final* id (src/bpe_model.cc:179) and on the
    // merged id when registering rev_merge (:103).
    for (uint32_t i = 0; i < t->sym_final.size(); ++i) {
      const uint32_t fid = t->sym_final[i] & kSfIdMask;
      t->sym_final[i] = (t->sym_final[i] & ~kSfUnused) | (m.pieces[fid].type == kUnused ? kSfUnused : 0);
    }
  }
}

Status CompileExtraOptions(const ModelData &m, const std::string &opts, HostTables *t) {
  // Simulate ApplyExtraOptions on a symbolic list: ids < 0 stand for the
  // sentence body; the net effect is always prefix + (body | reversed body) + suffix.
  std::vector<int> pre, suf;
  bool reversed = false;
  size_t pos = 0;
  if (!opts.empty()) {
    for (;;) {
      const size_t q = opts.find(':', pos);
      const std::string o = opts.substr(pos, q == std::string::npos ? std::string::npos : q - pos);
      if (o == "bos" || o == "eos") {
        // PieceToId(string_view(piece.data())) : the C-string prefix of the piece name
        const std::string &name = o == "bos" ? m.bos_piece : m.eos_piece;
        const int id = m.PieceToId(std::string(name.c_str()));
        if (m.pieces[id].type == kUnknown_)
          return Status::Error(kInternal, "id for `" + name + "` is not defined.");
        if (o == "bos") pre.insert(pre.begin(), id); else suf.push_back(id);
      } else if (o == "reverse") {
        std::vector<int> np(suf.rbegin(), suf.rend()), ns(pre.rbegin(), pre.rend());
        pre.swap(np);
        suf.swap(ns);
        reversed = !reversed;
      } else if (o == "unk" || o == "unk_piece") {
        // only rewrites piece strings; ids unchanged
      } else {
        return Status::Error(kInternal, "option \"" + o + "\" is not available.");
      }
      if (q == std::string::npos) break;
      pos = q + 1;
    }
  }
  if (pre.size() > kMaxExtra || suf.size() > kMaxExtra)
    return Status::Error(kUnimplemented, "more than 4 bos/eos ids on one side");
  SpmxDev &sc = t->scalars;
  sc.n_prefix = static_cast<int32_t>(pre.size());
  sc.n_suffix = static_cast<int32_t>(suf.size());
  for (size_t i = 0; i < pre.size(); ++i) sc.prefix_ids[i] = pre[i];
  for (size_t i = 0; i < suf.size(); ++i) sc.suffix_ids[i] = suf[i];
  sc.flags = (sc.flags & ~kNfReverse) | (reversed ? kNfReverse : 0);
  return Status::OK();
}

void BindHostPointers(HostTables *t) {
  SpmxDev &sc = t->scalars;
  sc.ntrie = t->ntrie.data();
  sc.ninfo = t->ninfo.data();
  sc.nblob = t->nblob.data();
  sc.ptrie = t->ptrie.data();
  sc.byte_ids = t->byte_ids.data();
  sc.utrie = t->utrie.data();
  sc.chartab = t->chartab.data();
  sc.pairtab = t->pairtab.data();
  sc.sym_final = t->sym_final.data();
  sc.sym_len = t->sym_len.data();
}

}  // namespace spmx
