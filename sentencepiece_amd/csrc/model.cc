#include "model.h"

#include <cfloat>
#include <cstdio>
#include <cstring>
#include <set>

namespace spmx {
namespace {

// proto2 wire format cursor.  Unknown fields are skipped by wire type.
struct Cursor {
  const uint8_t *p, *end;
  bool bad = false;
  uint64_t Varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      const uint8_t c = *p++;
      v |= static_cast<uint64_t>(c & 0x7F) << shift;
      if (!(c & 0x80)) return v;
    }
    bad = true;
    return 0;
  }
  // Next field: number (0 at end / on error), wire type, scalar or bytes view.
  int Next(int *wt, uint64_t *scalar, const uint8_t **bytes, size_t *len) {
    if (bad || p >= end) return 0;
    const uint64_t key = Varint();
    if (bad) return 0;
    *wt = static_cast<int>(key & 7);
    switch (*wt) {
      case 0: *scalar = Varint(); break;
      case 1:
        if (end - p < 8) { bad = true; return 0; }
        memcpy(scalar, p, 8); p += 8; break;
      case 5: {
        if (end - p < 4) { bad = true; return 0; }
        uint32_t t; memcpy(&t, p, 4); *scalar = t; p += 4; break;
      }
      case 2: {
        const uint64_t n = Varint();
        if (bad || static_cast<uint64_t>(end - p) < n) { bad = true; return 0; }
        *bytes = p; *len = static_cast<size_t>(n); p += n; break;
      }
      default: bad = true; return 0;
    }
    return static_cast<int>(key >> 3);
  }
};

bool ParsePiece(const uint8_t *b, size_t n, PieceRec *out) {
  Cursor c{b, b + n};
  int wt; uint64_t v = 0; const uint8_t *s = nullptr; size_t sl = 0;
  while (int f = c.Next(&wt, &v, &s, &sl)) {
    if (f == 1 && wt == 2) out->piece.assign(reinterpret_cast<const char *>(s), sl);
    else if (f == 2 && wt == 5) { const uint32_t u = static_cast<uint32_t>(v); memcpy(&out->score, &u, 4); }
    else if (f == 3 && wt == 0) out->load_type = out->type = static_cast<int>(v);
  }
  return !c.bad;
}

}  // namespace

Status ParseModelProto(const void *data, size_t n, ModelData *m) {
  const uint8_t *b = static_cast<const uint8_t *>(data);
  Cursor c{b, b + n};
  int wt; uint64_t v = 0; const uint8_t *s = nullptr; size_t sl = 0;
  while (int f = c.Next(&wt, &v, &s, &sl)) {
    if (wt != 2) continue;
    if (f == 1) {
      m->pieces.emplace_back();
      if (!ParsePiece(s, sl, &m->pieces.back())) c.bad = true;
    } else if (f == 2) {  // trainer_spec
      Cursor t{s, s + sl};
      const uint8_t *ts = nullptr; size_t tl = 0;
      while (int g = t.Next(&wt, &v, &ts, &tl)) {
        auto str = [&]() { return std::string(reinterpret_cast<const char *>(ts), tl); };
        if (g == 3 && wt == 0) m->model_type = static_cast<int>(v);
        else if (g == 24 && wt == 0) m->ws_suffix = v != 0;
        else if (g == 35 && wt == 0) m->byte_fallback = v != 0;
        // RETURN_PIECE (src/model_interface.cc:29-31): an empty field means the default.
        else if (g == 44 && wt == 2) m->unk_surface = std::string(str().c_str());   // used as a C string (:772-773)
        else if (g == 45 && wt == 2 && tl) m->unk_piece = str();
        else if (g == 46 && wt == 2 && tl) m->bos_piece = str();
        else if (g == 47 && wt == 2 && tl) m->eos_piece = str();
        else if (g == 48 && wt == 2 && tl) m->pad_piece = str();
      }
      if (t.bad) c.bad = true;
    } else if (f == 3) {  // normalizer_spec
      Cursor t{s, s + sl};
      const uint8_t *ts = nullptr; size_t tl = 0;
      while (int g = t.Next(&wt, &v, &ts, &tl)) {
        if (g == 2 && wt == 2) m->charsmap.assign(reinterpret_cast<const char *>(ts), tl);
        else if (g == 3 && wt == 0) m->add_dummy_prefix = v != 0;
        else if (g == 4 && wt == 0) m->remove_extra_ws = v != 0;
        else if (g == 5 && wt == 0) m->escape_ws = v != 0;
      }
      if (t.bad) c.bad = true;
    } else if (f == 5) {  // denormalizer_spec
      Cursor t{s, s + sl};
      const uint8_t *ts = nullptr; size_t tl = 0;
      while (int g = t.Next(&wt, &v, &ts, &tl)) {
        if (g == 2 && wt == 2) m->dn_charsmap.assign(reinterpret_cast<const char *>(ts), tl);
        else if (g == 3 && wt == 0) m->dn_add_dummy_prefix = v != 0;
        else if (g == 4 && wt == 0) m->dn_remove_extra_ws = v != 0;
        else if (g == 5 && wt == 0) m->dn_escape_ws = v != 0;
      }
      m->has_denormalizer = !m->dn_charsmap.empty();
      if (t.bad) c.bad = true;
    } else if (f == 4) {  // self_test_data { repeated Sample samples = 1 { input = 1; expected = 2 } }
      Cursor t{s, s + sl};
      const uint8_t *ts = nullptr; size_t tl = 0;
      while (int g = t.Next(&wt, &v, &ts, &tl)) {
        if (g != 1 || wt != 2) continue;
        Cursor u{ts, ts + tl};
        const uint8_t *us = nullptr; size_t ul = 0;
        std::pair<std::string, std::string> sample;
        while (int h = u.Next(&wt, &v, &us, &ul)) {
          if (h == 1 && wt == 2) sample.first.assign(reinterpret_cast<const char *>(us), ul);
          else if (h == 2 && wt == 2) sample.second.assign(reinterpret_cast<const char *>(us), ul);
        }
        if (u.bad) t.bad = true;
        m->self_test.push_back(std::move(sample));
      }
      if (t.bad) c.bad = true;
    }
  }
  if (c.bad) return Status::Error(kInternal, "could not parse ModelProto");
  return Status::OK();
}

int ModelData::PieceToId(const std::string &piece) const {
  auto it = reserved_map.find(piece);
  if (it != reserved_map.end()) return it->second;
  auto it2 = pieces_map.find(piece);
  if (it2 != pieces_map.end()) return it2->second;
  return unk_id;
}

Status InitializeModel(ModelData *m) {
  m->pieces_map.clear();
  m->reserved_map.clear();
  m->unk_id = -1;
  bool byte_found[256] = {false};
  for (int i = 0; i < static_cast<int>(m->pieces.size()); ++i) {
    const PieceRec &sp = m->pieces[i];
    if (sp.piece.empty()) return Status::Error(kInternal, "piece must not be empty.");
    const bool is_normal = sp.load_type == kNormal || sp.load_type == kUserDefined || sp.load_type == kUnused;
    auto &map = is_normal ? m->pieces_map : m->reserved_map;
    if (!map.emplace(sp.piece, i).second) return Status::Error(kInternal, sp.piece + " is already defined.");
    if (sp.load_type == kUnknown_) {
      if (m->unk_id >= 0) return Status::Error(kInternal, "unk is already defined.");
      m->unk_id = i;
    }
    if (sp.load_type == kByte) {
      if (!m->byte_fallback)
        return Status::Error(kInternal, "byte piece " + sp.piece + " is found although `byte_fallback` is false.");
      int byte = -1;  // PieceToByte (src/model_interface.cc:214-229)
      for (int b = 0; b < 256 && byte < 0; ++b) {
        char name[8];
        snprintf(name, sizeof(name), "<0x%02X>", b);
        if (sp.piece == name) byte = b;
      }
      if (byte < 0) return Status::Error(kInternal, "byte piece " + sp.piece + " is invalid.");
      byte_found[byte] = true;
    }
  }
  if (m->unk_id == -1) return Status::Error(kInternal, "unk is not defined.");
  if (m->byte_fallback)
    for (int b = 0; b < 256; ++b)
      if (!byte_found[b])
        return Status::Error(kInternal, "there are not 256 byte pieces although `byte_fallback` is true.");
  // PieceToId(ByteToPiece(b)) resolved once (src/sentencepiece_processor.cc:587-588).
  for (int b = 0; b < 256; ++b) {
    char name[8];
    snprintf(name, sizeof(name), "<0x%02X>", b);
    m->byte_ids[b] = m->PieceToId(name);
  }
  m->min_score = FLT_MAX;
  m->max_score = FLT_MIN;
  for (const PieceRec &sp : m->pieces) {
    if (sp.load_type == kNormal) {
      if (sp.score < m->min_score) m->min_score = sp.score;
      if (sp.score > m->max_score) m->max_score = sp.score;
    }
  }
  if (m->model_type == kUnigram && m->pieces_map.empty()) return Status::Error(kInternal, "no pieces are loaded.");
  return Status::OK();
}

Status SetVocabulary(ModelData *m, const std::vector<std::string> &valid) {
  if (m->model_type != kUnigram && m->model_type != kBpe)
    return Status::Error(kInternal, "Vocabulary constraint is only enabled in subword units.");
  const std::set<std::string> vocab(valid.begin(), valid.end());
  for (PieceRec &p : m->pieces) {
    if (p.type == kControl || p.type == kUnknown_ || p.type == kUserDefined) continue;
    if (vocab.count(p.piece) || static_cast<size_t>(OneCharLen(static_cast<unsigned char>(p.piece[0]))) == p.piece.size())
      p.type = kNormal;
    else
      p.type = kUnused;
  }
  return Status::OK();
}

Status ResetVocabulary(ModelData *m) {
  for (PieceRec &p : m->pieces)
    if (p.type == kUnused) p.type = kNormal;
  return Status::OK();
}

}  // namespace spmx
