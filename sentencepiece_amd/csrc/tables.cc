#include "tables.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "dat.h"

namespace spmx {
namespace {

void PackTrie2(const DatTrie &d, const std::vector<uint32_t> &payload, std::vector<U2> *out) {
  out->resize(d.w0.size());
  for (size_t i = 0; i < d.w0.size(); ++i) {
    (*out)[i].x = d.w0[i];
    (*out)[i].y = d.value[i] == 0xFFFFFFFFu ? 0u : payload[d.value[i]];
  }
}

uint32_t FloatBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// Strings the BPE symbol splitter can produce for one character: up to OneCharLen(first byte) bytes
// (src/normalizer.cc:336-344); under kNfCompressSp the byte kSpByte is a whole character.
bool IsCharLike(const std::string &s, bool compress) {
  if (s.empty() || s.size() > 4) return false;
  const unsigned char c = static_cast<unsigned char>(s[0]);
  const size_t mb = (compress && c == kSpByte) ? 1 : static_cast<size_t>(OneCharLen(c));
  return s.size() <= mb;
}
uint32_t PackChar(const std::string &s) {
  uint32_t v = 0;
  for (size_t i = 0; i < s.size(); ++i) v |= static_cast<uint32_t>(static_cast<unsigned char>(s[i])) << (8 * i);
  return v;
}
uint32_t NextPow2(size_t n) { uint32_t p = 16; while (p < n) p <<= 1; return p; }

const char kSpaceSymbol[] = "\xE2\x96\x81";   // U+2581 (src/normalizer.cc:109)

// kNfCompressSp: every U+2581 of a piece becomes the single byte kSpByte.
std::string ExpandSp(const std::string &s) {
  std::string o;
  for (unsigned char c : s) {
    if (c == kSpByte) o += kSpaceSymbol; else o.push_back(static_cast<char>(c));
  }
  return o;
}

std::string CompressSp(const std::string &s) {
  std::string o;
  o.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    if (s.compare(i, 3, kSpaceSymbol) == 0) { o.push_back(static_cast<char>(kSpByte)); i += 3; }
    else o.push_back(s[i++]);
  }
  return o;
}

}  // namespace

// Per id, what Decode(ids) appends for it (src/sentencepiece_processor.cc:776-808): the kind is taken from
// PieceToId(IdToPiece(id)) and the CURRENT piece types as the reference does (:812, IsByte / IsControl / IsUnknown read
// the live type: after SetVocabulary a BYTE piece that became UNUSED decodes as its literal "<0xNN>" text); U+2581 ->
// ' ' is applied here, the leading-whitespace rule at run time.  Rebuilt by RefreshTypeFlags.
Status BuildDecodeTables(const ModelData &m, HostTables *t) {
  const size_t V = m.pieces.size();
  t->dec_info.assign(V, 0);
  t->dec_off.assign(V + 1, 0);
  t->dec_bytes.clear();
  for (size_t i = 0; i < V; ++i) {
    const std::string &piece = m.pieces[i].piece;
    const int id = m.PieceToId(piece);
    const int type = m.pieces[id].type;
    uint32_t info = 0;
    t->dec_off[i] = static_cast<uint32_t>(t->dec_bytes.size());
    if (type == kControl) {
      info = 1u;                                                     // kDkEmpty
    } else if (type == kByte) {
      unsigned v = 0;                                                // PieceToByte("<0xHH>") (model_interface.cc:214-229)
      if (piece.size() == 6) v = static_cast<unsigned>(strtoul(piece.substr(3, 2).c_str(), nullptr, 16));
      info = 2u | (v << 8);                                          // kDkByte
    } else if (type == kUnknown_) {
      const std::string &sfc = (m.pieces[id].piece == piece) ? m.unk_surface : piece;
      t->dec_bytes.insert(t->dec_bytes.end(), sfc.begin(), sfc.end());
      info = 3u;                                                     // kDkLiteral
    } else {
      if (piece.compare(0, 3, kSpaceSymbol) == 0) info |= 1u << 2;   // kDiStartsSp
      for (size_t k = 0; k < piece.size();) {
        if (piece.compare(k, 3, kSpaceSymbol) == 0) { t->dec_bytes.push_back(' '); k += 3; }
        else t->dec_bytes.push_back(static_cast<uint8_t>(piece[k++]));
      }
    }
    const size_t dl = t->dec_bytes.size() - t->dec_off[i];
    if (dl > 0xFFFF) return Status::Error(kUnimplemented, "piece longer than 65535 bytes");
    t->dec_info[i] = info | static_cast<uint32_t>(dl) << 16;        // kDiLenShift
  }
  t->dec_off[V] = static_cast<uint32_t>(t->dec_bytes.size());
  if (t->dec_bytes.empty()) t->dec_bytes.push_back(0);
  return Status::OK();
}

// dev.h uw_f32_limit: the bound on |best_path_score| below which the wave-cooperative fold may add in float.
// Every score the fold adds (the pieces', the UNK candidate's) must be 0 or in [2^lo, 2^(hi+1)) with hi - lo <= 27, and
// none above 0 (then a best_path_score is 0 or at least 2^lo in magnitude: sums of non-positive floats only grow);
// a best_path_score below 2^(lo + 28) is then within 28 binades of every score.  One chunk of 128 positions adds at
// most 128 * 2^(hi+1) to the largest score it starts from: that is taken off.
static float UwFloatLimit(const ModelData &m, float unk_score) {
  int lo = 1000, hi = -1000;
  auto take = [&](float v) -> bool {
    if (!(v <= 0.0f) || v < -3.0e38f) return false;      // above 0, NaN, -inf
    if (v == 0.0f) return true;
    int e = 0;
    (void)std::frexp(static_cast<double>(-v), &e);       // -v = f * 2^e, f in [0.5, 1): exponent e - 1
    lo = std::min(lo, e - 1); hi = std::max(hi, e - 1);
    return true;
  };
  if (!take(unk_score)) return 0.0f;
  for (const PieceRec &p : m.pieces) {
    if (p.type == kUserDefined || p.load_type == kUserDefined) return 0.0f;
    if (p.load_type == kNormal || p.load_type == kUnused || p.load_type == kByte) { if (!take(p.score)) return 0.0f; }
  }
  if (lo > hi || hi - lo > 27 || lo < -60 || hi > 60) return 0.0f;
  const double limit = std::ldexp(1.0, lo + 28) - 128.0 * std::ldexp(1.0, hi + 1);
  return limit > 0.0 ? static_cast<float>(limit * 0.999) : 0.0f;
}

Status CompileTables(const ModelData &m, HostTables *t) {
  std::string err;
  SpmxDev &sc = t->scalars;
  sc = SpmxDev{};
  sc.model_type = m.model_type;
  if (!m.normalizer_only && m.model_type != kUnigram && m.model_type != kBpe)
    return Status::Error(kUnimplemented, "only unigram and bpe models are on the device path");
  if (m.pieces.size() >= (1u << 30)) return Status::Error(kResourceExhausted, "vocabulary too large");
  uint32_t flags = 0;
  if (m.add_dummy_prefix) flags |= kNfAddDummyPrefix;
  if (m.remove_extra_ws) flags |= kNfRemoveExtraWs;
  if (m.escape_ws) flags |= kNfEscapeWs;
  if (m.ws_suffix) flags |= kNfWsSuffix;
  if (m.byte_fallback) flags |= kNfByteFallback;

  // ---------------------------------------------------------- normalizer ---
  // DecodePrecompiledCharsMap (src/normalizer.cc:274-309): <u32 trie bytes><Darts units><strings>.
  // The units are used on the device exactly as serialized; every probe is
  // bounds-checked there, so a malformed blob cannot read out of range.
  t->ndarts.clear();
  t->nblob.clear();
  if (!m.charsmap.empty()) {
    const std::string &blob = m.charsmap;
    if (blob.size() <= 4) return Status::Error(kInternal, "Blob for normalization rule is broken.");
    uint32_t trie_bytes = 0;
    memcpy(&trie_bytes, blob.data(), 4);
    if (trie_bytes >= blob.size()) return Status::Error(kInternal, "Trie data size exceeds the input blob size.");
    t->ndarts.resize(trie_bytes / 4);
    memcpy(t->ndarts.data(), blob.data() + 4, t->ndarts.size() * 4);
    t->nblob.assign(blob.begin() + 4 + trie_bytes, blob.end());
    if (!t->ndarts.empty()) flags |= kNfHasCharsmap;
  }
  // ASCII fast path of the normalizer: which ASCII first bytes can only start a rule when a non-ASCII byte
  // follows (in nmt_nfkc / nfkc: letters that take combining marks).  Walk the model's own Darts units
  // (darts.h:50-80, :467-513) two levels deep.
  for (uint32_t &w : sc.ascii_safe) w = 0;
  {
    const std::vector<uint32_t> &u = t->ndarts;
    auto offset = [](uint32_t x) { return (x >> 10) << ((x & (1u << 9)) >> 6); };
    for (uint32_t b = 0; b < 128; ++b) {
      bool safe = true;
      if (!u.empty()) {
        uint32_t pos = offset(u[0]) ^ b;
        if (pos < u.size() && (u[pos] & 0x800000FFu) == b) {       // a key starts with b
          if ((u[pos] >> 8) & 1u) safe = false;                    // b alone is a key
          const uint32_t base = pos ^ offset(u[pos]);
          for (uint32_t c2 = 0; c2 < 128 && safe; ++c2) {
            const uint32_t p2 = base ^ c2;
            if (p2 < u.size() && (u[p2] & 0x800000FFu) == c2) safe = false;   // b followed by ASCII continues a key
          }
        }
      }
      if (safe) sc.ascii_safe[b >> 5] |= 1u << (b & 31);
    }
  }
  // two-byte prefix filter of the charsmap keys (dev.h npair), from the same two-level walk
  t->npair.assign(2048, 0);
  {
    const std::vector<uint32_t> &u = t->ndarts;
    auto offset = [](uint32_t x) { return (x >> 10) << ((x & (1u << 9)) >> 6); };
    for (uint32_t b0 = 1; b0 < 256 && !u.empty(); ++b0) {
      const uint32_t pos = offset(u[0]) ^ b0;
      if (pos >= u.size() || (u[pos] & 0x800000FFu) != b0) continue;       // no key starts with b0
      const bool alone = (u[pos] >> 8) & 1u;                               // b0 alone is a key
      const uint32_t base = pos ^ offset(u[pos]);
      for (uint32_t b1 = 0; b1 < 256; ++b1) {
        bool hit = alone;
        if (!hit && b1 != 0) { const uint32_t p2 = base ^ b1; hit = p2 < u.size() && (u[p2] & 0x800000FFu) == b1; }
        if (hit) t->npair[(b0 << 8 | b1) >> 5] |= 1u << (b1 & 31u);
      }
    }
  }
  // code-point filters of the keys (dev.h nfilt): every key by a depth-first walk that carries the key's first 8 bytes
  sc.nfilt = 0;
  if (!t->ndarts.empty()) {
    const std::vector<uint32_t> &u = t->ndarts;
    auto offset = [](uint32_t x) { return (x >> 10) << ((x & (1u << 9)) >> 6); };
    // one character of s[0, n): its length (0: not valid UTF-8 by DecodeUTF8's rule, src/util.cc:51-84) and code point
    auto decode = [](const unsigned char *s, size_t n, uint32_t *cp) -> int {
      if (n == 0) return 0;
      const unsigned b0 = s[0];
      auto ct = [&](size_t k) { return k < n && (s[k] & 0xC0u) == 0x80u; };
      if (b0 < 0x80) { *cp = b0; return 1; }
      if (b0 >= 0xC2 && b0 < 0xE0 && ct(1)) { *cp = (b0 & 0x1Fu) << 6 | (s[1] & 0x3Fu); return 2; }
      if (b0 >= 0xE0 && b0 < 0xF0 && ct(1) && ct(2)) {
        *cp = (b0 & 0x0Fu) << 12 | (s[1] & 0x3Fu) << 6 | (s[2] & 0x3Fu);
        return *cp >= 0x800 && (*cp < 0xD800 || *cp >= 0xE000) ? 3 : 0;
      }
      if (b0 >= 0xF0 && b0 < 0xF8 && ct(1) && ct(2) && ct(3)) {
        *cp = (b0 & 0x07u) << 18 | (s[1] & 0x3Fu) << 12 | (s[2] & 0x3Fu) << 6 | (s[3] & 0x3Fu);
        return *cp >= 0x10000 && *cp <= 0x10FFFF ? 4 : 0;
      }
      return 0;
    };
    std::vector<uint32_t> f(3 * kNfiltWords + kNfiltCps, 0);     // the three bitmaps, then the one-character rules (dev.h)
    auto mark = [&](int which, uint32_t cp) {
      if (cp >= kNfiltCps) return;                        // (the kernels take such characters the general way)
      f[static_cast<size_t>(which) * kNfiltWords + (cp >> 5)] |= 1u << (cp & 31u);
    };
    struct Node { uint32_t pos, depth; unsigned char key[8]; };
    std::vector<Node> todo;
    Node root{offset(u[0]), 0, {0}};
    todo.push_back(root);
    size_t visited = 0;
    bool ok = true;
    while (!todo.empty() && ok && visited < 256 * u.size() + 1024) {
      const Node nd = todo.back();
      todo.pop_back();
      ++visited;
      for (uint32_t c = 1; c < 256 && ok; ++c) {
        const uint32_t p = nd.pos ^ c;
        if (p >= u.size() || (u[p] & 0x800000FFu) != c) continue;
        Node ch = nd;
        ch.pos = p ^ offset(u[p]);
        ch.depth = nd.depth + 1;
        if (nd.depth < 8) ch.key[nd.depth] = static_cast<unsigned char>(c);
        if ((u[p] >> 8) & 1u) {                             // a key ends here
          const size_t have = ch.depth < 8 ? ch.depth : 8;
          uint32_t c1 = 0, c2 = 0;
          const int l1 = decode(ch.key, have, &c1);
          if (l1 == 0) {
            if (getenv("SPMX_DEBUG_TABLES")) { fprintf(stderr, "spmx: charsmap key whose first character does not decode:"); for (size_t q = 0; q < have; ++q) fprintf(stderr, " %02X", ch.key[q]); fprintf(stderr, " (depth %u)\n", ch.depth); }
            ok = false; break;
          }
          mark(0, c1);
          if (static_cast<uint32_t>(l1) == ch.depth) {
            mark(1, c1);
            // the rule of a key that is ONE character: offset | length << 24 of its replacement when that is 1 .. 255 bytes
            // without a space (a space goes through the whitespace state machine: the general steps take those)
            if (c1 < kNfiltCps && ch.pos < u.size()) {
              const uint32_t off = u[ch.pos] & 0x7FFFFFFFu;
              size_t n = 0;
              bool plain = off < (1u << 24);
              while (off + n < t->nblob.size() && t->nblob[off + n] != 0) { if (t->nblob[off + n] == ' ') plain = false; ++n; }
              if (plain && n >= 1 && n <= 255) f[3 * kNfiltWords + c1] = off | static_cast<uint32_t>(n) << 24;
            }
          }
          else {
            const int l2 = decode(ch.key + l1, have - static_cast<size_t>(l1), &c2);
            if (l2 == 0) {
              if (getenv("SPMX_DEBUG_TABLES")) { fprintf(stderr, "spmx: charsmap key whose second character does not decode:"); for (size_t q = 0; q < have; ++q) fprintf(stderr, " %02X", ch.key[q]); fprintf(stderr, " (depth %u)\n", ch.depth); }
              ok = false; break;
            }
            mark(2, c2);
          }
        }
        if (ch.depth < 256) todo.push_back(ch);
      }
    }
    if (ok && todo.empty()) {
      t->npair.insert(t->npair.end(), f.begin(), f.end());
      sc.nfilt = 1;
    }
    if (getenv("SPMX_DEBUG_TABLES") && sc.nfilt) {
      size_t n[3] = {0, 0, 0};
      for (int w = 0; w < 3; ++w) for (uint32_t q = 0; q < kNfiltWords; ++q) n[w] += static_cast<size_t>(__builtin_popcount(f[static_cast<size_t>(w) * kNfiltWords + q]));
      fprintf(stderr, "spmx: characters that start a key %zu, are a key alone %zu, are a key's second character %zu; second characters below U+0300:", n[0], n[1], n[2]);
      for (uint32_t cp = 0; cp < 0x300; ++cp) if ((f[2 * kNfiltWords + (cp >> 5)] >> (cp & 31u)) & 1u) fprintf(stderr, " %04X", cp);
      fprintf(stderr, "\n");
    }
    if (getenv("SPMX_DEBUG_TABLES")) fprintf(stderr, "spmx: code-point filters of the charsmap keys: %s (%zu nodes visited, %zu left)\n", sc.nfilt ? "built" : "NOT built", visited, todo.size());
  }
  // Largest growth of a NormalizePrefix result over the bytes it consumes (dev.h expand_max): every key of the
  // charsmap trie by a depth-first walk of its units, the replacement's length with every ' ' counted as a
  // three-byte U+2581; at least 3 (U+FFFD for one malformed byte, an escaped space).
  sc.expand_max = 3;
  if (!t->ndarts.empty()) {
    const std::vector<uint32_t> &u = t->ndarts;
    auto offset = [](uint32_t x) { return (x >> 10) << ((x & (1u << 9)) >> 6); };
    struct Node { uint32_t pos, depth; };
    std::vector<Node> todo;
    todo.push_back({offset(u[0]), 0});                      // pos = node index after its own offset was applied
    size_t visited = 0;
    // (the trie is a DAWG: a unit is reached once per key prefix that leads to it -- nmt_nfkc: 44,800 units, 262,841 prefixes;
    // the bound only stops a malformed blob's cycles)
    while (!todo.empty() && visited < 256 * u.size() + 1024) {
      const Node nd = todo.back();
      todo.pop_back();
      ++visited;
      for (uint32_t c = 1; c < 256; ++c) {
        const uint32_t p = nd.pos ^ c;
        if (p >= u.size() || (u[p] & 0x800000FFu) != c) continue;
        const uint32_t child = p ^ offset(u[p]);
        const uint32_t depth = nd.depth + 1;
        if (c == 0x20u && nd.depth >= 1) t->charsmap_inner_space = true;   // (every edge of the trie lies on the way to a key)
        if ((u[p] >> 8) & 1u) {                             // a key ends here: its value sits in the unit at child
          if (child < u.size()) {
            const uint32_t off = u[child] & 0x7FFFFFFFu;
            uint64_t out = 0;
            for (size_t k = off; k < t->nblob.size() && t->nblob[k] != 0; ++k) out += t->nblob[k] == ' ' ? 3 : 1;
            const uint64_t ratio = (out + depth - 1) / depth;
            if (ratio > sc.expand_max) sc.expand_max = static_cast<uint32_t>(ratio);
          }
        }
        if (depth < 256) todo.push_back({child, depth});
      }
    }
    if (getenv("SPMX_DEBUG_TABLES")) fprintf(stderr, "spmx: expand_max walk: %zu prefixes visited, %zu left, expand_max %u\n", visited, todo.size(), sc.expand_max);
  }
  // every replacement string valid UTF-8 by the reference's own rule (DecodeUTF8, src/util.cc:51-84)?  (kernels_matchfold.h)
  bool blob_utf8 = true;
  for (size_t i = 0; i < t->nblob.size() && blob_utf8;) {
    const unsigned b0 = t->nblob[i];
    auto cont = [&](size_t k) { return i + k < t->nblob.size() && (t->nblob[i + k] & 0xC0u) == 0x80u; };
    if (b0 < 0x80) { i += 1; continue; }
    if (b0 >= 0xC2 && b0 < 0xE0 && cont(1)) { i += 2; continue; }
    if (b0 >= 0xE0 && b0 < 0xF0 && cont(1) && cont(2)) {
      const unsigned cp = (b0 & 0x0Fu) << 12 | (t->nblob[i + 1] & 0x3Fu) << 6 | (t->nblob[i + 2] & 0x3Fu);
      if (cp >= 0x800 && (cp < 0xD800 || cp >= 0xE000)) { i += 3; continue; }
    }
    if (b0 >= 0xF0 && b0 < 0xF8 && cont(1) && cont(2) && cont(3)) {
      const unsigned cp = (b0 & 0x07u) << 18 | (t->nblob[i + 1] & 0x3Fu) << 12 | (t->nblob[i + 2] & 0x3Fu) << 6 | (t->nblob[i + 3] & 0x3Fu);
      if (cp >= 0x10000 && cp <= 0x10FFFF) { i += 4; continue; }
    }
    blob_utf8 = false;
  }
  if (t->ndarts.empty()) t->ndarts.push_back(0);
  if (t->nblob.empty()) t->nblob.push_back(0);
  sc.ndarts_n = static_cast<uint32_t>(t->ndarts.size());
  sc.nblob_n = static_cast<uint32_t>(t->nblob.size());
  std::vector<std::pair<std::string, uint32_t>> uds_keys;
  for (size_t i = 0; i < m.pieces.size(); ++i) {
    if (m.pieces[i].load_type != kUserDefined) continue;
    uds_keys.emplace_back(m.pieces[i].piece, static_cast<uint32_t>(i));
    flags |= kNfHasUserDefined;
  }
  {  // PrefixMatcher over USER_DEFINED pieces (src/normalizer.cc:311-346): raw text in the normalizer, normalized text in BPE
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

  // One-byte space symbol (dev.h kNfCompressSp).  Sound when 0xFF cannot reach the normalized text by any other
  // route and every U+2581 in it is produced where the device normalizer can see it: escaped spaces and literal
  // U+2581 characters of the raw text.  So: no user-defined symbols (their raw spans are copied verbatim), no
  // 0xFF / U+2581 inside the charsmap's replacement strings, no 0xFF inside a piece.
  bool compress = m.escape_ws && uds_keys.empty() && !getenv("SPMX_NO_COMPRESS");
  if (compress) {
    const std::string blob(t->nblob.begin(), t->nblob.end());
    if (blob.find(static_cast<char>(kSpByte)) != std::string::npos || blob.find(kSpaceSymbol) != std::string::npos)
      compress = false;
    for (const auto &kv : m.pieces_map)
      if (kv.first.find(static_cast<char>(kSpByte)) != std::string::npos) { compress = false; break; }
  }
  if (compress) flags |= kNfCompressSp;
  if (m.normalizer_only) {      // a Normalizer(spec) without a model behind it: the denormalizer
    sc.flags = flags;
    return Status::OK();
  }

  // ------------------------------------------------------- id post-process --
  t->byte_ids.assign(m.byte_ids, m.byte_ids + 256);
  sc.unk_id = m.unk_id;
  sc.unk_score = m.min_score - 10.0f;  // kUnkPenalty (src/unigram_model.cc:39, :955), float arithmetic
  sc.max_score = m.max_score;

  // --------------------------------------------------------------- unigram --
  t->ptrie.clear();
  t->plen.assign(m.pieces.size() + 1, 0);
  t->max_piece_len = 0;
  if (m.model_type == kUnigram) {
    std::vector<std::pair<std::string, uint32_t>> keys;
    for (const auto &kv : m.pieces_map)
      keys.emplace_back(compress ? CompressSp(kv.first) : kv.first, static_cast<uint32_t>(kv.second));
    for (const auto &kv : keys) t->plen[kv.second] = static_cast<uint8_t>(kv.first.size() > 255 ? 255 : kv.first.size());
    DatTrie d;
    if (!BuildDat(keys, &d, &err)) return Status::Error(kInternal, "piece trie: " + err);
    t->max_piece_len = d.max_key_len;
    t->max_prefixes = d.max_prefixes;
    t->max_piece_chars = 0;
    for (const auto &kv : keys) {
      int nc = 0;
      for (unsigned char ch : kv.first) nc += (ch & 0xC0u) != 0x80u;
      if (nc > t->max_piece_chars) t->max_piece_chars = nc;
    }
    sc.uw_f32_limit = UwFloatLimit(m, sc.unk_score);
    // (the candidate word of the split form holds 19 bits of id, 6 of byte length, 4 of character length)
    t->split_ok = blob_utf8 && uds_keys.empty() && m.pieces.size() < (1u << 19) && d.max_prefixes >= 1 && d.max_prefixes <= 16 &&
                  d.max_key_len <= 63 && t->max_piece_chars <= 15;
    if (d.max_key_len > kMaxPieceBytes)
      return Status::Error(kUnimplemented, "a piece is longer than 120 bytes; unsupported by the device unigram path");
    t->ptrie.resize(d.w0.size());
    for (size_t i = 0; i < d.w0.size(); ++i) {
      U4 u{d.w0[i], 0, 0, 0};
      if (d.value[i] != 0xFFFFFFFFu) {
        u.y = d.value[i];
        u.z = FloatBits(m.pieces[d.value[i]].score);
      }
      t->ptrie[i] = u;
    }
    // child-label summaries (dev.h ChildBit): a unit at index j with label c is the child of the node whose
    // base is j ^ c; bases are unique, so map base -> node unit first
    {
      std::vector<uint32_t> owner(d.w0.size(), 0xFFFFFFFFu);   // base -> unit that owns it
      for (size_t i = 0; i < d.w0.size(); ++i)
        if (i == 0 || (d.w0[i] & kDatOccupied)) {
          const uint32_t base = d.w0[i] >> kDatBaseShift;
          if (base < owner.size() && (i == 0 || base != 0)) owner[base] = static_cast<uint32_t>(i);
        }
      for (size_t j = 1; j < d.w0.size(); ++j) {
        if (!(d.w0[j] & kDatOccupied)) continue;
        const uint32_t c = d.w0[j] & 0xFFu;
        const uint32_t base = static_cast<uint32_t>(j) ^ c;
        if (base < owner.size() && owner[base] != 0xFFFFFFFFu) t->ptrie[owner[base]].w |= 1u << ChildBit(c);
      }
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
    // Everything below works on the strings as the device sees them: with kNfCompressSp every U+2581 of a piece
    // is the byte kSpByte.  tmap = pieces_ (src/bpe_model.cc:88-94) keyed by those strings.
    auto dev_str = [&](const std::string &p) { return compress ? CompressSp(p) : p; };
    std::map<std::string, int> tmap;
    bool wordwise = compress && !m.ws_suffix && !getenv("SPMX_NO_WORDWISE");
    for (const auto &kv : m.pieces_map) {
      const std::string t = dev_str(kv.first);
      tmap.emplace(t, kv.second);
      // a piece is either a run of space symbols (allow_whitespace_only_pieces) or has none after its first character
      const bool all_sp = t.find_first_not_of(static_cast<char>(kSpByte)) == std::string::npos;
      if (!all_sp && t.find(static_cast<char>(kSpByte), 1) != std::string::npos) wordwise = false;
    }
    if (wordwise) flags |= kNfBpeWordwise;
    // symbol universe: piece ids [0, V) for strings in pieces_, then extra
    // char-like strings that are split parts of a piece or reserved names.
    std::map<std::string, uint32_t> extra;
    t->sym_final.assign(V, 0);
    t->sym_len.assign(V, 0);
    for (uint32_t i = 0; i < V; ++i) {
      // PieceToId(piece string) (src/bpe_model.cc:178, src/model_interface.cc:51-61)
      t->sym_final[i] = static_cast<uint32_t>(m.PieceToId(m.pieces[i].piece));
      const size_t dl = dev_str(m.pieces[i].piece).size();
      if (dl > 0xFFFF) return Status::Error(kUnimplemented, "piece longer than 65535 bytes");
      t->sym_len[i] = static_cast<uint16_t>(dl);
    }
    auto sym_of = [&](const std::string &s, bool create) -> uint32_t {
      auto it = tmap.find(s);
      if (it != tmap.end()) return static_cast<uint32_t>(it->second);
      if (!IsCharLike(s, compress)) return kSymNone;
      auto e = extra.find(s);
      if (e != extra.end()) return e->second;
      if (!create) return kSymNone;
      const uint32_t id = V + static_cast<uint32_t>(extra.size());
      extra.emplace(s, id);
      return id;
    };
    // chars that only exist as reserved names still need their PieceToId
    for (const auto &kv : m.reserved_map) {
      const std::string t = dev_str(kv.first);
      if (IsCharLike(t, compress)) sym_of(t, true);
    }
    struct PairEnt { uint32_t a, b, merged; float score; };
    std::vector<PairEnt> pairs;
    for (const auto &kv : tmap) {
      const std::string &p = kv.first;
      for (size_t k = 1; k < p.size(); ++k) {
        const std::string a = p.substr(0, k), b = p.substr(k);
        const bool a_ok = tmap.count(a) || IsCharLike(a, compress);
        const bool b_ok = tmap.count(b) || IsCharLike(b, compress);
        if (!a_ok || !b_ok) continue;
        pairs.push_back({sym_of(a, true), sym_of(b, true), static_cast<uint32_t>(kv.second), m.pieces[kv.second].score});
      }
    }
    t->sym_final.resize(V + extra.size());
    t->sym_len.resize(V + extra.size());
    for (const auto &kv : extra) {
      t->sym_final[kv.second] = static_cast<uint32_t>(m.PieceToId(compress ? ExpandSp(kv.first) : kv.first));
      t->sym_len[kv.second] = static_cast<uint16_t>(kv.first.size());
    }
    // char table: every char-like string that has a symbol
    std::vector<std::pair<std::string, uint32_t>> chars;
    for (const auto &kv : tmap) if (IsCharLike(kv.first, compress)) chars.emplace_back(kv.first, static_cast<uint32_t>(kv.second));
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
    // Word table (dev.h wordtab): SampleEncode(alpha = 0) of every vocabulary string that is a whole word -- at most 16
    // bytes, an optional run of space symbols and then characters without one -- by the merge loop itself on the
    // compiled tables (:142-173: best pair = highest score, then leftmost).  Only for models that segment word by word.
    t->wordtab.clear();
    sc.wordtab_mask = 0;
    if (wordwise && !getenv("SPMX_NO_WORDTAB")) {
      auto host_char = [&](uint32_t bytes, uint32_t len) -> uint32_t {
        uint32_t sl = HashChar(bytes, len) & sc.chartab_mask;
        for (;;) {
          const U4 &e = t->chartab[sl];
          if (e.y == 0) return kSymNone;
          if (e.x == bytes && e.y == len) return e.z;
          sl = (sl + 1) & sc.chartab_mask;
        }
      };
      auto host_pair = [&](uint32_t a, uint32_t b, uint32_t *mg, float *score) -> bool {
        uint32_t sl = HashPair(a, b) & sc.pairtab_mask;
        for (;;) {
          const U4 &e = t->pairtab[sl];
          if (e.x == kSymNone) return false;
          if (e.x == a && e.y == b) { *mg = e.z; memcpy(score, &e.w, 4); return true; }
          sl = (sl + 1) & sc.pairtab_mask;
        }
      };
      struct WordEnt { uint32_t k[4]; uint32_t meta, id[3]; };
      std::vector<WordEnt> words;
      for (const auto &kv : tmap) {
        const std::string &w = kv.first;
        if (w.empty() || w.size() > kWordKeyBytes) continue;
        size_t lead = 0;
        while (lead < w.size() && static_cast<unsigned char>(w[lead]) == kSpByte) ++lead;
        if (lead == w.size() || w.find(static_cast<char>(kSpByte), lead) != std::string::npos) continue;
        if (w.find('\0') != std::string::npos) continue;
        std::vector<uint32_t> sym;
        std::vector<uint32_t> len;
        bool ok = true;
        for (size_t p = 0; p < w.size() && ok;) {
          const unsigned char c = static_cast<unsigned char>(w[p]);
          size_t mb = c == kSpByte ? 1 : static_cast<size_t>(OneCharLen(c));
          if (mb > w.size() - p) mb = w.size() - p;
          const uint32_t sy = host_char(PackChar(w.substr(p, mb)), static_cast<uint32_t>(mb));
          if (sy == kSymNone) ok = false;
          sym.push_back(sy);
          len.push_back(static_cast<uint32_t>(mb));
          p += mb;
        }
        while (ok && sym.size() > 1) {
          int best = -1;
          float bs = 0.f;
          uint32_t bm = 0;
          for (size_t i = 0; i + 1 < sym.size(); ++i) {
            uint32_t mg = 0;
            float scv = 0.f;
            if (host_pair(sym[i], sym[i + 1], &mg, &scv) && (best < 0 || scv > bs)) { best = static_cast<int>(i); bs = scv; bm = mg; }
          }
          if (best < 0) break;
          sym[best] = bm;
          len[best] += len[best + 1];
          sym.erase(sym.begin() + best + 1);
          len.erase(len.begin() + best + 1);
        }
        if (!ok || sym.size() > kWordMaxIds) continue;
        WordEnt e{};
        unsigned char key[kWordKeyBytes] = {0};
        memcpy(key, w.data(), w.size());
        memcpy(e.k, key, sizeof(key));
        e.meta = static_cast<uint32_t>(w.size()) | static_cast<uint32_t>(sym.size()) << 8;
        for (size_t i = 0; i < sym.size() && ok; ++i) {
          const uint32_t f = t->sym_final[sym[i]];
          const uint32_t fid = f & kSfIdMask;
          if ((f & kSfControl) || static_cast<int>(fid) == m.unk_id || m.pieces[fid].type == kUnknown_) ok = false;
          e.id[i] = fid;
          e.meta |= len[i] << (16 + 5 * i);
        }
        if (ok) words.push_back(e);
      }
      // a small vocabulary holds few whole words: most lookups would miss and only cost (measured: a 1k-piece model
      // ran at 94 M sentences/s with the table and 121 M without); SPMX_WORDTAB_MIN overrides the threshold
      size_t min_words = 4096;
      if (const char *e = getenv("SPMX_WORDTAB_MIN")) min_words = static_cast<size_t>(atoll(e));
      if (!words.empty() && words.size() >= min_words) {
        const uint32_t wsz = NextPow2(words.size() * 2 + 16);
        t->wordtab.assign(static_cast<size_t>(wsz) * 2, U4{0, 0, 0, 0});
        for (const WordEnt &e : words) {
          uint32_t sl = HashWord(e.k[0], e.k[1], e.k[2], e.k[3]) & (wsz - 1);
          while (t->wordtab[2 * sl + 1].x != 0) sl = (sl + 1) & (wsz - 1);
          t->wordtab[2 * sl] = U4{e.k[0], e.k[1], e.k[2], e.k[3]};
          t->wordtab[2 * sl + 1] = U4{e.meta, e.id[0], e.id[1], e.id[2]};
        }
        sc.wordtab_mask = wsz - 1;
      }
    }
  } else {
    t->chartab.assign(16, U4{0, 0, kSymNone, 0});
    t->pairtab.assign(16, U4{kSymNone, kSymNone, kSymNone, 0});
    t->sym_final.assign(1, 0);
    t->sym_len.assign(1, 0);
    t->wordtab.clear();
    sc.wordtab_mask = 0;
  }
  if (Status ds = BuildDecodeTables(m, t); !ds.ok()) return ds;
  if (t->wordtab.empty()) t->wordtab.assign(2, U4{0, 0, 0, 0});
  sc.n_pieces = static_cast<uint32_t>(m.pieces.size());
  sc.flags = flags;
  RefreshTypeFlags(m, t);
  return Status::OK();
}

// The word memo of the unigram word form (dev.h umemo, kernels_word.h): for every vocabulary string that is a whole word
// -- U+2581 and then 1 .. 16 bytes 0x21 .. 0x7E -- EncodeOptimized (src/unigram_model.cc:889-1020) of that word alone, in
// double, together with the margin analysis kernels_word.h states: `gap` = the smallest lead of the best candidate over
// the second best at the positions on the word's best path, `wmag` = the largest magnitude of any candidate.  The entry
// is valid while |best_path_score before the word| < bmax, bmax = 2^(k + 1) - wmag with 2^(k - 23) the largest float
// ulp not above gap / (4 * nchar + 4).  Words whose best path holds an unknown piece, more than two pieces, or an exact
// tie are left out: they take the general kernels.
// The first-character table of the piece trie (dev.h SpmxDev::cfirst).  Follows the live piece types (the units carry the
// UNUSED flag): rebuilt with them.
void BuildFirstCharTable(const ModelData &m, HostTables *t) {
  t->cfirst.clear();
  t->scalars.cfirst = nullptr;
  if (m.model_type != kUnigram || t->ptrie.empty() || getenv("SPMX_NO_CFIRST")) return;
  size_t multi = 0;
  const bool one_byte_sp = (t->scalars.flags & kNfCompressSp) != 0;      // (the space symbol is then the byte 0xFF in the trie's keys)
  for (const auto &kv : m.pieces_map) {
    const unsigned char c0 = kv.first.empty() ? 0 : static_cast<unsigned char>(kv.first[0]);
    if (one_byte_sp && kv.first.compare(0, 3, kSpaceSymbol) == 0) continue;
    if (c0 >= 0xC2 && c0 < 0xF0) ++multi;
  }
  if (multi < 1024) return;                              // (an ASCII vocabulary: the table would only cost cache)
  const uint32_t root = t->ptrie[0].x >> kDatBaseShiftDev;
  std::vector<U4> tab(65536, U4{1, 0, 0, 0});
  for (uint32_t cp = 0x80; cp < 0x10000; ++cp) {
    unsigned char b[3];
    int D;
    if (cp < 0x800) { b[0] = static_cast<unsigned char>(0xC0 | (cp >> 6)); b[1] = static_cast<unsigned char>(0x80 | (cp & 0x3F)); D = 2; }
    else { b[0] = static_cast<unsigned char>(0xE0 | (cp >> 12)); b[1] = static_cast<unsigned char>(0x80 | ((cp >> 6) & 0x3F));
           b[2] = static_cast<unsigned char>(0x80 | (cp & 0x3F)); D = 3; }
    uint32_t node = root;
    U4 u{0, 0, 0, 0};
    bool ok = true;
    for (int k = 0; k < D && ok; ++k) {
      const uint32_t at = node ^ b[k];
      if (at >= t->ptrie.size()) { ok = false; break; }
      u = t->ptrie[at];
      if ((u.x & 0x1FFu) != (0x100u | b[k])) { ok = false; break; }
      if (k + 1 < D && (u.x & kDatTerminalDev)) {        // a piece that ends inside a character: no table for this model
        if (getenv("SPMX_DEBUG_TABLES")) fprintf(stderr, "spmx: no first-character table: a piece ends inside U+%04X\n", cp);
        return;
      }
      node = u.x >> kDatBaseShiftDev;
    }
    tab[cp] = ok ? U4{(u.x & ~0xFFu) | static_cast<uint32_t>(D), u.y, u.z, u.w} : U4{static_cast<uint32_t>(D), 0, 0, 0};
  }
  t->cfirst.swap(tab);
  t->scalars.cfirst = t->cfirst.data();
  if (getenv("SPMX_DEBUG_TABLES")) fprintf(stderr, "spmx: first-character table built (%zu pieces start with a multi-byte character)\n", multi);
}

void BuildWordMemo(const ModelData &m, HostTables *t) {
  SpmxDev &sc = t->scalars;
  t->umemo.assign(2, U4{0, 0, 0, 0});
  t->umemo[1].x = 0xFFFFFFFFu;
  t->umemo16.assign(1, U4{0, 0, 0, 0xFFFFFFFFu});
  t->uall.assign(2, U4{0, 0, 0, 0});
  t->uall[1].x = 0xFFFFFFFFu;
  sc.uall_mask = 0;
  sc.uall_perfect = 0;
  t->udisp.assign(kUallBuckets, 0);
  t->uhot.assign(kWordHotSlots, U4{0, 0, 0, 0xFFFFFFFFu});
  t->uhot2.assign(kWordHotSlots, U4{0, 0, 0, 0xFFFFFFFFu});
  sc.umemo_mask = 0;
  sc.umemo16_mask = 0;
  sc.flags &= ~(kNfUniWordwise | kNfWordLocalNorm);
  t->memo_words = t->memo_candidates = 0;
  t->pscore.assign(m.pieces.size() + 1, 0.f);
  for (size_t i = 0; i < m.pieces.size(); ++i) t->pscore[i] = m.pieces[i].score;
  if ((m.model_type != kUnigram && m.model_type != kBpe) || getenv("SPMX_NO_WORD")) return;
  const uint32_t F = sc.flags;
  // BPE: the word-wise models only (dev.h kNfBpeWordwise), and no UNUSED piece (resegmentation, src/bpe_model.cc:175-200)
  if (m.model_type == kBpe && (!(F & kNfBpeWordwise) || (F & kNfHasUnused))) return;
  // (remove_extra_whitespaces may be off -- the Llama-style models: the word loop then leaves every sentence with a leading, a
  // doubled or a trailing space to the general kernels, kernels_word.h keep_ws)
  if (!(F & kNfCompressSp) || !(F & kNfAddDummyPrefix) || (F & kNfWsSuffix) || (F & kNfHasUserDefined)) return;
  for (uint32_t b = 0x20; b < 0x7F; ++b)
    if (!((sc.ascii_safe[b >> 5] >> (b & 31u)) & 1u)) return;     // a charsmap rule may start with an ASCII byte
  const std::string sp(kSpaceSymbol);
  // no piece may reach across a word boundary -- pieces of space symbols only (allow_whitespace_only_pieces) cannot: they
  // match inside a run of space symbols, which is never part of a word here; no USER_DEFINED piece (its score is not a
  // float, :979-981)
  for (const auto &kv : m.pieces_map) {
    bool all_sp = kv.first.size() % sp.size() == 0;
    for (size_t q = 0; all_sp && q < kv.first.size(); q += sp.size()) all_sp = kv.first.compare(q, sp.size(), sp) == 0;
    if (!all_sp && kv.first.find(sp, 1) != std::string::npos) return;
    if (m.pieces[kv.second].type == kUserDefined) return;
  }
  const double unk_score = static_cast<double>(m.min_score - 10.0f);
  struct Ent { uint32_t k[4]; uint32_t id0, id1; float s0, bmax; float order; };
  std::vector<Ent> ents;
  // BPE: bpe::Model::SampleEncode(alpha = 0) (src/bpe_model.cc:38-203) of one word on the compiled tables -- the merge
  // loop itself (:142-173: best pair = highest score, then leftmost); exact, no margin involved (bmax = "always")
  auto bpe_word = [&](const std::string &w, std::vector<uint32_t> *ids) -> bool {
    auto host_char = [&](uint32_t bytes, uint32_t len) -> uint32_t {
      uint32_t sl = HashChar(bytes, len) & sc.chartab_mask;
      for (;;) {
        const U4 &e = t->chartab[sl];
        if (e.y == 0) return kSymNone;
        if (e.x == bytes && e.y == len) return e.z;
        sl = (sl + 1) & sc.chartab_mask;
      }
    };
    auto host_pair = [&](uint32_t a, uint32_t b, uint32_t *mg, float *score) -> bool {
      uint32_t sl = HashPair(a, b) & sc.pairtab_mask;
      for (;;) {
        const U4 &e = t->pairtab[sl];
        if (e.x == kSymNone) return false;
        if (e.x == a && e.y == b) { *mg = e.z; memcpy(score, &e.w, 4); return true; }
        sl = (sl + 1) & sc.pairtab_mask;
      }
    };
    std::vector<uint32_t> sym;
    for (size_t p = 0; p < w.size(); ++p) {             // (the space symbol and ASCII: one byte per character)
      const uint32_t sy = host_char(static_cast<unsigned char>(w[p]), 1u);
      if (sy == kSymNone) return false;
      sym.push_back(sy);
    }
    while (sym.size() > 1) {
      int best = -1;
      float bs = 0.f;
      uint32_t bm = 0;
      for (size_t i = 0; i + 1 < sym.size(); ++i) {
        uint32_t mg = 0;
        float scv = 0.f;
        if (host_pair(sym[i], sym[i + 1], &mg, &scv) && (best < 0 || scv > bs)) { best = static_cast<int>(i); bs = scv; bm = mg; }
      }
      if (best < 0) break;
      sym[best] = bm;
      sym.erase(sym.begin() + best + 1);
    }
    ids->clear();
    for (uint32_t sy : sym) {
      const uint32_t f = t->sym_final[sy];
      const uint32_t fid = f & kSfIdMask;
      if ((f & kSfControl) || static_cast<int>(fid) == m.unk_id || m.pieces[fid].type == kUnknown_) return false;
      ids->push_back(fid);
    }
    return true;
  };
  for (const auto &kv : m.pieces_map) {
    const std::string &pc = kv.first;
    if (pc.size() <= sp.size() || pc.size() > sp.size() + kWordKeyBytes || pc.compare(0, sp.size(), sp) != 0) continue;
    const std::string body = pc.substr(sp.size());
    bool plain = true;
    for (unsigned char c : body) plain = plain && c >= 0x21 && c <= 0x7E;
    if (!plain) continue;
    ++t->memo_candidates;
    if (m.model_type == kBpe) {
      std::vector<uint32_t> ids;
      if (!bpe_word(std::string(1, static_cast<char>(kSpByte)) + body, &ids) || ids.empty() || ids.size() > 2) continue;
      Ent en{};
      unsigned char key[kWordKeyBytes];
      memset(key, 0x20, sizeof(key));                     // (kernels_word.h key_dword: padded with 0x20, which no word holds)
      memcpy(key, body.data(), body.size());
      memcpy(en.k, key, sizeof(key));
      en.id0 = ids[0];
      en.id1 = ids.size() > 1 ? ids[1] : 0xFFFFFFFFu;
      en.s0 = 0.f;
      en.bmax = 3.0e38f;
      en.order = m.pieces[kv.second].score;
      ents.push_back(en);
      continue;
    }
    // characters of the word: [0] = U+2581, then one byte each; cb[i] = byte offset of character i
    const int nchar = 1 + static_cast<int>(body.size());
    auto cb = [&](int i) -> size_t { return i == 0 ? 0 : sp.size() + static_cast<size_t>(i - 1); };
    const double kNone = -1e300;
    std::vector<double> best(nchar + 1, kNone), second(nchar + 1, kNone);
    std::vector<int> bp(nchar + 1, -1), bid(nchar + 1, -1);
    best[0] = 0.0;
    double wmag = 0.0;
    auto relax = [&](int e, double cand, int s, int id) {
      if (fabs(cand) > wmag) wmag = fabs(cand);
      if (cand > best[e]) { second[e] = best[e]; best[e] = cand; bp[e] = s; bid[e] = id; }
      else if (cand > second[e]) second[e] = cand;
    };
    for (int s = 0; s < nchar; ++s) {                     // every character start is reachable (a piece or UNK, :995-1005)
      bool single = false;
      for (int e = s + 1; e <= nchar; ++e) {
        const auto it = m.pieces_map.find(pc.substr(cb(s), cb(e) - cb(s)));
        if (it == m.pieces_map.end()) continue;
        const PieceRec &r = m.pieces[it->second];
        if (r.type == kUnused) continue;                  // :974
        relax(e, best[s] + static_cast<double>(r.score), s, it->second);
        if (e == s + 1) single = true;                    // :990 a piece of exactly one character
      }
      if (!single) relax(s + 1, best[s] + unk_score, s, -1);
    }
    // the best path and its smallest lead
    std::vector<int> ids;
    double gap = 1e300;
    bool ok = true;
    for (int e = nchar; e > 0 && ok; e = bp[e]) {
      if (bp[e] < 0 || bid[e] < 0) { ok = false; break; }   // unreachable (cannot happen) / an unknown piece on the path
      if (second[e] > kNone) gap = std::min(gap, best[e] - second[e]);
      ids.push_back(bid[e]);
    }
    if (!ok || ids.size() > 2 || !(gap > 0.0)) continue;
    std::reverse(ids.begin(), ids.end());
    const double thr = gap / (4.0 * nchar + 4.0);
    int ex = 0;
    (void)frexp(thr, &ex);                                // thr = f * 2^ex, f in [0.5, 1): floor(log2 thr) = ex - 1
    const int k = (ex - 1) + 23;                          // the largest k with 2^(k - 23) <= thr
    double lim = k + 1 > 126 ? 3.0e38 : ldexp(1.0, k + 1);
    double bmax = lim - wmag * (1.0 + 1e-6) - 1e-30;
    if (!(bmax > 0.0)) continue;
#ifdef SPMX_TEST_SEAMS   // (the emulator build only, tests/emu/Makefile: the release library has no such switch)
    if (getenv("SPMX_WORDMEMO_UNSAFE")) bmax = 3.1e38;    // no margin guard -- shows that tests/test_word_form.py's near-tie fuzz has teeth
#endif
    float bf = bmax > 3.0e38 ? 3.0e38f : static_cast<float>(bmax);
    if (static_cast<double>(bf) > bmax) bf = nextafterf(bf, 0.f);
    if (!(bf > 0.f)) continue;
    Ent en{};
    unsigned char key[kWordKeyBytes];
    memset(key, 0x20, sizeof(key));
    memcpy(key, body.data(), body.size());
    memcpy(en.k, key, sizeof(key));
    en.id0 = static_cast<uint32_t>(ids[0]);
    en.id1 = ids.size() > 1 ? static_cast<uint32_t>(ids[1]) : 0xFFFFFFFFu;
    en.s0 = m.pieces[ids[0]].score;
    en.bmax = bf;
    en.order = m.pieces[kv.second].score;
    ents.push_back(en);
  }
  size_t min_words = 64;                                  // a handful of whole words: every sentence would miss anyway
  if (const char *e = getenv("SPMX_WORDMEMO_MIN")) min_words = static_cast<size_t>(atoll(e));
  if (ents.size() < min_words) return;
  // likelier words first: they get the slots their hash names, the rest walk (the kernel's lanes wait for the longest walk)
  std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.order > b.order; });
  auto bound_of = [&](uint32_t id) -> double {          // (BPE keeps no score: its entries are valid whatever came before)
    return m.model_type == kBpe ? 0.0 : ceil(fabs(static_cast<double>(m.pieces[id].score))) + 1.0;
  };
  // 16-byte entries (dev.h umemo16): words of up to 12 bytes that are ONE piece, and -- the TWO-PIECE form -- words of up
  // to 10 bytes that are one or two pieces: their key needs only the low half of the third key dword, the high half holds
  // the second id (0xFFFF: none); bit 23 of the meta word tells the forms apart (a lookup knows which one its word's
  // length asks for).  Everything else takes the 32-byte entries.
  std::vector<const Ent *> small, big;
  auto len_of = [](const Ent &e) -> int {
    const unsigned char *kb = reinterpret_cast<const unsigned char *>(e.k);
    int l = 0;
    while (l < static_cast<int>(kWordKeyBytes) && kb[l] != 0x20) ++l;
    return l;
  };
  auto bound2 = [&](const Ent &e) -> double { return bound_of(e.id0) + (e.id1 != 0xFFFFFFFFu ? bound_of(e.id1) : 0.0); };
  for (const Ent &e : ents) {
    const bool one = e.id1 == 0xFFFFFFFFu;
    const int len = len_of(e);
    const bool fits = e.id0 < 65535u && (one || e.id1 < 65535u) && bound2(e) <= 255.0 && e.bmax >= 1.0f;
#ifdef SPMX_TEST_SEAMS
    static const bool one_only = getenv("SPMX_MEMO16_ONE") != nullptr;   // A/B (emulator build): only one-piece words in the 16-byte entries, as in round 3
#else
    constexpr bool one_only = false;
#endif
    if (fits && ((one && len <= 12) || (len <= 10 && !one_only))) small.push_back(&e);
    else big.push_back(&e);
  }
  auto key2_16 = [&](const Ent &e) -> uint32_t {        // the third dword of a 16-byte entry
    if (len_of(e) > 10) return e.k[2];
    return (e.k[2] & 0xFFFFu) | (e.id1 == 0xFFFFFFFFu ? 0xFFFFu : e.id1) << 16;
  };
  auto meta16 = [&](const Ent &e) -> uint32_t {
    int ex = 0;
    (void)frexpf(e.bmax, &ex);                            // bmax = f * 2^ex, f in [0.5, 1): 2^(ex - 1) <= bmax
    int pw = ex - 1;
    if (pw > 126) pw = 126;
    return e.id0 | static_cast<uint32_t>(pw) << 16 | (len_of(e) <= 10 ? kMemo16TwoPiece : 0u) | static_cast<uint32_t>(bound2(e)) << 24;
  };
  // The word-per-lane kernels (kernels_wordwave.h, the default) read uhot2 and uall only.  The open-addressed tiers of the
  // sentence-per-lane kernels (kernels_word.h: umemo16, uhot, umemo) are built -- and uploaded -- only for a handle that
  // asks for those kernels (SPMX_WORD_WAVE=0 / 1 / 2, SPMX_NO_WORD_DYN=1, SPMX_FORCE_WORD_DP=1: read here as api.cc reads them); otherwise they stay the empty
  // sentinels set above (uni32k: 4 MB + 2 MB of HBM per handle and their inserts at load left out).
  const bool legacy_tiers = [] {
    auto on = [](const char *name) { const char *e = getenv(name); return e != nullptr && e[0] == '1'; };
    const char *e = getenv("SPMX_WORD_WAVE");
    return (e != nullptr && atoi(e) >= 0 && atoi(e) <= 2) || on("SPMX_NO_WORD_DYN") || on("SPMX_FORCE_WORD_DP");   // (the DP pass is a sentence-per-lane kernel too)
  }();
  for (const Ent *e : small) {   // the LDS table of the likeliest words (they come likeliest first)
    U4 &h2s = t->uhot2[HashWordKey(e->k[0], e->k[1], e->k[2], e->k[3]) & (static_cast<uint32_t>(t->uhot2.size()) - 1u)];
    if (h2s.w == 0xFFFFFFFFu) h2s = U4{e->k[0], e->k[1], key2_16(*e), meta16(*e)};
  }
  if (legacy_tiers) {
    const uint32_t wsz = NextPow2(small.size() * 6 + 16);     // sparse: a collision costs the whole wave another probe
    t->umemo16.assign(wsz, U4{0, 0, 0, 0xFFFFFFFFu});
    for (const Ent *e : small) {
      const uint32_t h = HashWordKey(e->k[0], e->k[1], e->k[2], 0u);
      U4 &hs = t->uhot[h & (static_cast<uint32_t>(t->uhot.size()) - 1u)];
      if (hs.w == 0xFFFFFFFFu) hs = U4{e->k[0], e->k[1], key2_16(*e), meta16(*e)};
      uint32_t sl = h & (wsz - 1);
      while (t->umemo16[sl].w != 0xFFFFFFFFu) sl = (sl + 1) & (wsz - 1);
      t->umemo16[sl] = U4{e->k[0], e->k[1], key2_16(*e), meta16(*e)};
    }
    sc.umemo16_mask = wsz - 1;
  }
  if (legacy_tiers) {
    const uint32_t wsz = NextPow2(big.size() * 2 + 16);
    t->umemo.assign(static_cast<size_t>(wsz) * 2, U4{0, 0, 0, 0});
    for (uint32_t i = 0; i < wsz; ++i) t->umemo[2 * i + 1].x = 0xFFFFFFFFu;
    for (const Ent *e : big) {
      uint32_t sl = HashWordKey(e->k[0], e->k[1], e->k[2], e->k[3]) & (wsz - 1);
      while (t->umemo[2 * sl + 1].x != 0xFFFFFFFFu) sl = (sl + 1) & (wsz - 1);
      const double b = bound_of(e->id0) + (e->id1 != 0xFFFFFFFFu ? bound_of(e->id1) : 0.0);
      t->umemo[2 * sl] = U4{e->k[0], e->k[1], e->k[2], e->k[3]};
      t->umemo[2 * sl + 1] = U4{e->id0, e->id1, FloatBits(static_cast<float>(b)), FloatBits(e->bmax)};
    }
    sc.umemo_mask = wsz - 1;
  }
  {   // every word in the 32-byte format (dev.h uall) behind a perfect hash: hash, displace -- the buckets largest first,
      // each takes the first displacement that puts all of its keys on free slots
    auto put = [&](uint32_t sl, const Ent &e) {
      t->uall[2 * sl] = U4{e.k[0], e.k[1], e.k[2], e.k[3]};
      t->uall[2 * sl + 1] = U4{e.id0, e.id1, FloatBits(static_cast<float>(bound2(e))), FloatBits(e.bmax)};
    };
    std::vector<uint32_t> h1(ents.size()), h2(ents.size());
    std::vector<std::vector<uint32_t>> buckets(kUallBuckets);
    for (size_t i = 0; i < ents.size(); ++i) {
      h1[i] = HashWordKey(ents[i].k[0], ents[i].k[1], ents[i].k[2], ents[i].k[3]);
      h2[i] = UallHash2(ents[i].k[0], ents[i].k[1], ents[i].k[2], ents[i].k[3], h1[i]);
      buckets[UallBucket(h2[i])].push_back(static_cast<uint32_t>(i));
    }
    std::vector<uint32_t> order(kUallBuckets);
    for (uint32_t b = 0; b < kUallBuckets; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return buckets[x].size() > buckets[y].size(); });
    bool perfect = false;
    uint32_t wsz = NextPow2(ents.size() * 2 + 16);
    for (int attempt = 0; attempt < 4 && !perfect && wsz <= (1u << 22); ++attempt, wsz <<= 1) {
      t->uall.assign(static_cast<size_t>(wsz) * 2, U4{0, 0, 0, 0});
      for (uint32_t i = 0; i < wsz; ++i) t->uall[2 * i + 1].x = 0xFFFFFFFFu;
      std::vector<uint8_t> taken(wsz, 0);
      std::vector<uint32_t> sl;
      perfect = true;
      for (uint32_t b : order) {
        const std::vector<uint32_t> &ks = buckets[b];
        if (ks.empty()) break;
        bool placed = false;
        for (uint32_t dd = 0; dd < 65536u && !placed; ++dd) {
          sl.clear();
          bool ok = true;
          for (uint32_t i : ks) {
            const uint32_t q = UallSlot(h1[i], h2[i], dd, wsz - 1);
            if (taken[q] || std::find(sl.begin(), sl.end(), q) != sl.end()) { ok = false; break; }
            sl.push_back(q);
          }
          if (!ok) continue;
          for (size_t x = 0; x < ks.size(); ++x) { taken[sl[x]] = 1; put(sl[x], ents[ks[x]]); }
          t->udisp[b] = static_cast<uint16_t>(dd);
          placed = true;
        }
        if (!placed) { perfect = false; break; }
      }
      if (perfect) { sc.uall_mask = wsz - 1; sc.uall_perfect = 1; }
    }
    if (!perfect) {   // (a vocabulary too large for the bucket table: open addressing, likelier words first)
      wsz = NextPow2(ents.size() * 2 + 16);
      t->uall.assign(static_cast<size_t>(wsz) * 2, U4{0, 0, 0, 0});
      for (uint32_t i = 0; i < wsz; ++i) t->uall[2 * i + 1].x = 0xFFFFFFFFu;
      for (const Ent &e : ents) {
        uint32_t q = HashWordKey(e.k[0], e.k[1], e.k[2], e.k[3]) & (wsz - 1);
        while (t->uall[2 * q + 1].x != 0xFFFFFFFFu) q = (q + 1) & (wsz - 1);
        put(q, e);
      }
      std::fill(t->udisp.begin(), t->udisp.end(), 0);
      sc.uall_mask = wsz - 1;
      sc.uall_perfect = 0;
    }
  }
  sc.flags |= kNfUniWordwise;
  // words that are not plain ASCII through the call-local memo (dev.h kNfWordLocalNorm)
  if ((F & kNfRemoveExtraWs) && !t->charsmap_inner_space && !getenv("SPMX_NO_WORD_NORM")) sc.flags |= kNfWordLocalNorm;
  t->memo_words = static_cast<uint32_t>(ents.size());
}

void RefreshTypeFlags(const ModelData &m, HostTables *t) {
  (void)BuildDecodeTables(m, t);        // (cannot fail here: the same pieces passed at load)
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
  BuildWordMemo(m, t);                  // (follows the live piece types: an UNUSED piece is no candidate)
  BuildFirstCharTable(m, t);
}

Status CompileExtraOptions(const ModelData &m, const std::string &opts, HostTables *t) {
  // Simulate ApplyExtraOptions on a symbolic list: ids < 0 stand for the
  // sentence body; the net effect is always prefix + (body | reversed body) + suffix.
  std::vector<int> pre, suf;
  std::vector<int> pre_eos, suf_eos;      // parallel: 1 for an eos
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
        if (o == "bos") { pre.insert(pre.begin(), id); pre_eos.insert(pre_eos.begin(), 0); }
        else { suf.push_back(id); suf_eos.push_back(1); }
      } else if (o == "reverse") {
        std::vector<int> np(suf.rbegin(), suf.rend()), ns(pre.rbegin(), pre.rend());
        pre.swap(np);
        suf.swap(ns);
        std::vector<int> npe(suf_eos.rbegin(), suf_eos.rend()), nse(pre_eos.rbegin(), pre_eos.rend());
        pre_eos.swap(npe);
        suf_eos.swap(nse);
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
  sc.extra_eos = 0;
  for (size_t i = 0; i < pre.size(); ++i) sc.extra_eos |= static_cast<uint32_t>(pre_eos[i]) << i;
  for (size_t i = 0; i < suf.size(); ++i) sc.extra_eos |= static_cast<uint32_t>(suf_eos[i]) << (kMaxExtra + i);
  sc.flags = (sc.flags & ~kNfReverse) | (reversed ? kNfReverse : 0);
  return Status::OK();
}

void BindHostPointers(HostTables *t) {
  SpmxDev &sc = t->scalars;
  sc.plen = t->plen.data();
  sc.ndarts = t->ndarts.data();
  sc.nblob = t->nblob.data();
  sc.npair = t->npair.data();
  sc.ptrie = t->ptrie.data();
  sc.cfirst = t->cfirst.empty() ? nullptr : t->cfirst.data();
  sc.byte_ids = t->byte_ids.data();
  sc.dec_info = t->dec_info.data();
  sc.dec_off = t->dec_off.data();
  sc.dec_bytes = t->dec_bytes.data();
  sc.utrie = t->utrie.data();
  sc.chartab = t->chartab.data();
  sc.pairtab = t->pairtab.data();
  sc.sym_final = t->sym_final.data();
  sc.sym_len = t->sym_len.data();
  sc.wordtab = t->wordtab.data();
  sc.umemo = t->umemo.data();
  sc.umemo16 = t->umemo16.data();
  sc.uall = t->uall.data();
  sc.uhot2 = t->uhot2.data();
  sc.udisp = t->udisp.data();
  sc.uhot = t->uhot.data();
  sc.pscore = t->pscore.data();
}

}  // namespace spmx
