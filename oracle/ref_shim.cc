// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from
// the product path (sentencepiece_amd/).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load the library built from this file.
//
// A thin extern "C" shim over the *compiled upstream reference*
// (sentencepiece::SentencePieceProcessor, /root/reference/src/
// sentencepiece_processor.h:245-300, :458-460, :622-631).  It is linked
// against objects built from the reference sources where they lie (see
// oracle/Makefile); no reference source is copied here.  It exists so that
// Python tests can (1) pin the C restatement in spm_oracle.c against the
// real thing and (2) time the reference CPU path as bench.py's cpu_baseline.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "builtin_pb/sentencepiece.pb.h"
#include "sentencepiece_processor.h"
#include "normalizer.h"
#include "unigram_model.h"

namespace {
struct RefHandle {
  sentencepiece::SentencePieceProcessor sp;
  std::string last_error;
};
}  // namespace

extern "C" {

void *spmref_load(const void *model_bytes, uint64_t n) {
  auto *h = new RefHandle;
  const auto st = h->sp.LoadFromSerializedProto(
      absl::string_view(static_cast<const char *>(model_bytes), n));
  if (!st.ok()) {
    delete h;
    return nullptr;
  }
  return h;
}

void spmref_free(void *handle) { delete static_cast<RefHandle *>(handle); }

const char *spmref_last_error(void *handle) {
  return static_cast<RefHandle *>(handle)->last_error.c_str();
}

// SetEncodeExtraOptions (sentencepiece_processor.h:267). 0 on success.
int spmref_set_encode_extra_options(void *handle, const char *opts) {
  auto *h = static_cast<RefHandle *>(handle);
  const auto st = h->sp.SetEncodeExtraOptions(opts);
  if (!st.ok()) h->last_error = st.ToString();
  return st.ok() ? 0 : static_cast<int>(st.code());
}

// SetDecodeExtraOptions (sentencepiece_processor.h:270). 0 on success.
int spmref_set_decode_extra_options(void *handle, const char *opts) {
  auto *h = static_cast<RefHandle *>(handle);
  const auto st = h->sp.SetDecodeExtraOptions(opts);
  if (!st.ok()) h->last_error = st.ToString();
  return st.ok() ? 0 : static_cast<int>(st.code());
}

// SetVocabulary / ResetVocabulary (sentencepiece_processor.h:279-283).
// `pieces` is a '\n'-joined list.
int spmref_set_vocabulary(void *handle, const char *pieces, uint64_t len) {
  auto *h = static_cast<RefHandle *>(handle);
  std::vector<std::string> store;
  const char *p = pieces, *end = pieces + len;
  while (p < end) {
    const char *q = static_cast<const char *>(memchr(p, '\n', end - p));
    if (!q) q = end;
    store.emplace_back(p, q - p);
    p = q + 1;
  }
  std::vector<absl::string_view> v(store.begin(), store.end());
  const auto st = h->sp.SetVocabulary(v);
  if (!st.ok()) h->last_error = st.ToString();
  return st.ok() ? 0 : static_cast<int>(st.code());
}

int spmref_reset_vocabulary(void *handle) {
  auto *h = static_cast<RefHandle *>(handle);
  return h->sp.ResetVocabulary().ok() ? 0 : 1;
}

// Encode(input, vector<int>*) (sentencepiece_processor.h:299-300).
// Returns the id count (>= 0), -1 on a Status error, or -(needed) - 2 if
// `cap` is too small.
int64_t spmref_encode(void *handle, const char *text, uint64_t len,
                      int32_t *out, uint64_t cap) {
  auto *h = static_cast<RefHandle *>(handle);
  std::vector<int> ids;
  const auto st = h->sp.Encode(absl::string_view(text, len), &ids);
  if (!st.ok()) {
    h->last_error = st.ToString();
    return -1;
  }
  if (ids.size() > cap) return -static_cast<int64_t>(ids.size()) - 2;
  for (size_t i = 0; i < ids.size(); ++i) out[i] = ids[i];
  return static_cast<int64_t>(ids.size());
}

// Normalize(input, &normalized) (sentencepiece_processor.h:622-623).
int64_t spmref_normalize(void *handle, const char *text, uint64_t len,
                         char *out, uint64_t cap) {
  auto *h = static_cast<RefHandle *>(handle);
  std::string normalized;
  const auto st = h->sp.Normalize(absl::string_view(text, len), &normalized);
  if (!st.ok()) {
    h->last_error = st.ToString();
    return -1;
  }
  if (normalized.size() > cap) return -static_cast<int64_t>(normalized.size()) - 2;
  memcpy(out, normalized.data(), normalized.size());
  return static_cast<int64_t>(normalized.size());
}

// The batch form of python/src/sentencepiece/sentencepiece.i:245-267: a pool
// of `num_threads` workers pulling sentence indices from one atomic counter,
// each calling Encode(ins[i], &ids).  Ids are written to a flat CSR:
// id_offsets[n + 1], ids[cap].  Two passes are avoided by letting every
// worker keep its results and copying at the end.  Returns total ids, -1 on
// a Status error, or -(needed) - 2 if cap is too small.
int64_t spmref_encode_batch(void *handle, const char *text,
                            const uint64_t *offsets, uint64_t n,
                            int32_t *ids, uint64_t cap, uint64_t *id_offsets,
                            int num_threads) {
  auto *h = static_cast<RefHandle *>(handle);
  if (num_threads < 1) num_threads = 1;
  std::vector<std::vector<int>> outs(n);
  std::atomic<uint64_t> index{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    for (;;) {
      const uint64_t i = index.fetch_add(1);
      if (i >= n) return;
      const auto st = h->sp.Encode(
          absl::string_view(text + offsets[i], offsets[i + 1] - offsets[i]),
          &outs[i]);
      if (!st.ok()) failed = true;
    }
  };
  if (num_threads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < num_threads; ++t) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
  }
  if (failed) return -1;
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    id_offsets[i] = total;
    total += outs[i].size();
  }
  id_offsets[n] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  for (uint64_t i = 0; i < n; ++i)
    for (size_t k = 0; k < outs[i].size(); ++k) ids[id_offsets[i] + k] = outs[i][k];
  return static_cast<int64_t>(total);
}

// Timed loop for cpu_baseline: encodes every sentence, keeps only the count.
// Same worker scheme as above, but without materialising the CSR.
int64_t spmref_encode_count(void *handle, const char *text,
                            const uint64_t *offsets, uint64_t n,
                            int num_threads) {
  auto *h = static_cast<RefHandle *>(handle);
  if (num_threads < 1) num_threads = 1;
  std::atomic<uint64_t> index{0};
  std::atomic<int64_t> total{0};
  auto worker = [&]() {
    std::vector<int> ids;
    int64_t local = 0;
    for (;;) {
      const uint64_t i = index.fetch_add(1);
      if (i >= n) break;
      if (h->sp.Encode(absl::string_view(text + offsets[i],
                                         offsets[i + 1] - offsets[i]),
                       &ids)
              .ok())
        local += static_cast<int64_t>(ids.size());
    }
    total += local;
  };
  if (num_threads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < num_threads; ++t) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
  }
  return total;
}

// SentencePieceProcessor::Decode(const std::vector<int>&, std::string*) per sentence of a CSR id buffer.
// Returns total bytes, -1 on a Status error, -(needed) - 2 if cap is too small.
int64_t spmref_decode_batch(void *handle, const int32_t *ids, const uint64_t *id_offsets, uint64_t n,
                            char *text, uint64_t cap, uint64_t *text_offsets) {
  auto *h = static_cast<RefHandle *>(handle);
  std::string all;
  for (uint64_t i = 0; i < n; ++i) {
    text_offsets[i] = all.size();
    std::vector<int> v(ids + id_offsets[i], ids + id_offsets[i + 1]);
    std::string out;
    const auto st = h->sp.Decode(v, &out);
    if (!st.ok()) { h->last_error = st.ToString(); return -1; }
    all += out;
  }
  text_offsets[n] = all.size();
  if (all.size() > cap) return -static_cast<int64_t>(all.size()) - 2;
  memcpy(text, all.data(), all.size());
  return static_cast<int64_t>(all.size());
}

// Encode(input, SentencePieceText *) per sentence (sentencepiece_processor.h:401-402): ids and
// pieces(i).begin() / .end().  Returns total pieces, -1 on a Status error, -(needed) - 2 if cap is too small.
int64_t spmref_encode_spans_batch(void *handle, const char *text, const uint64_t *offsets, uint64_t n, int32_t *ids,
                                  uint32_t *begin, uint32_t *end, uint64_t cap, uint64_t *id_offsets) {
  auto *h = static_cast<RefHandle *>(handle);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    id_offsets[i] = total;
    sentencepiece::SentencePieceText spt;
    const auto st = h->sp.Encode(absl::string_view(text + offsets[i], offsets[i + 1] - offsets[i]), &spt);
    if (!st.ok()) { h->last_error = st.ToString(); return -1; }
    for (int k = 0; k < spt.pieces_size(); ++k) {
      if (total < cap) {
        ids[total] = static_cast<int32_t>(spt.pieces(k).id());
        begin[total] = spt.pieces(k).begin();
        end[total] = spt.pieces(k).end();
      }
      ++total;
    }
  }
  id_offsets[n] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  return static_cast<int64_t>(total);
}

// The same with pieces(i).piece(), packed (piece k at [piece_offsets[k], piece_offsets[k + 1]), cap + 1 entries).
int64_t spmref_encode_pieces_batch(void *handle, const char *text, const uint64_t *offsets, uint64_t n, int32_t *ids,
                                   uint32_t *begin, uint32_t *end, uint64_t cap, uint64_t *id_offsets, char *pieces,
                                   uint64_t pieces_cap, uint64_t *piece_offsets) {
  auto *h = static_cast<RefHandle *>(handle);
  uint64_t total = 0, pbytes = 0;
  for (uint64_t i = 0; i < n; ++i) {
    id_offsets[i] = total;
    sentencepiece::SentencePieceText spt;
    const auto st = h->sp.Encode(absl::string_view(text + offsets[i], offsets[i + 1] - offsets[i]), &spt);
    if (!st.ok()) { h->last_error = st.ToString(); return -1; }
    for (int k = 0; k < spt.pieces_size(); ++k) {
      const std::string &pc = spt.pieces(k).piece();
      if (total < cap) {
        ids[total] = static_cast<int32_t>(spt.pieces(k).id());
        begin[total] = spt.pieces(k).begin();
        end[total] = spt.pieces(k).end();
        piece_offsets[total] = pbytes;
        if (pbytes + pc.size() <= pieces_cap) memcpy(pieces + pbytes, pc.data(), pc.size());
        // surface must be the input slice (:577-578)
        if (spt.pieces(k).surface() != std::string(text + offsets[i] + spt.pieces(k).begin(), spt.pieces(k).end() - spt.pieces(k).begin())) {
          h->last_error = "surface is not input[begin, end)";
          return -1;
        }
      }
      pbytes += pc.size();
      ++total;
    }
  }
  id_offsets[n] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  piece_offsets[total] = pbytes;
  if (pbytes > pieces_cap) return -static_cast<int64_t>(pbytes) - 3;
  return static_cast<int64_t>(total);
}

// Normalize(input, &normalized, &norm_to_orig) per sentence (sentencepiece_processor.h:627-629); layout as
// oracle_normalize_batch.
int64_t spmref_normalize_batch(void *handle, const char *text, const uint64_t *offsets, uint64_t n, char *out,
                               uint64_t cap, uint64_t *norm_offsets, uint32_t *n2o) {
  auto *h = static_cast<RefHandle *>(handle);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    norm_offsets[i] = total;
    std::string norm;
    std::vector<size_t> a;
    const auto st = h->sp.Normalize(absl::string_view(text + offsets[i], offsets[i + 1] - offsets[i]), &norm, &a);
    if (!st.ok()) { h->last_error = st.ToString(); return -1; }
    if (total + norm.size() <= cap) {
      if (!norm.empty()) memcpy(out + total, norm.data(), norm.size());
      if (n2o) {
        if (a.empty()) n2o[total + i] = 0xFFFFFFFFu;
        else for (size_t k = 0; k < a.size(); ++k) n2o[total + i + k] = static_cast<uint32_t>(a[k]);
      }
    }
    total += norm.size();
  }
  norm_offsets[n] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  return static_cast<int64_t>(total);
}

// NBestEncode(input, nbest_size, &ids) + the scores of NBestEncode(input, nbest_size, NBestSentencePieceText*)
// (sentencepiece_processor.h:323-324, :404-405).  Layout as oracle_nbest_encode.
int64_t spmref_nbest_encode(void *handle, const char *text, uint64_t len, int nbest_size, int32_t *ids, uint64_t cap,
                            uint64_t *offs, float *scores) {
  auto *h = static_cast<RefHandle *>(handle);
  sentencepiece::NBestSentencePieceText nb;
  const auto st = h->sp.NBestEncode(absl::string_view(text, len), nbest_size, &nb);
  if (!st.ok()) { h->last_error = st.ToString(); return -1; }
  uint64_t total = 0;
  for (int k = 0; k < nb.nbests_size(); ++k) {
    offs[k] = total;
    scores[k] = nb.nbests(k).score();
    for (int i = 0; i < nb.nbests(k).pieces_size(); ++i) {
      if (total < cap) ids[total] = static_cast<int32_t>(nb.nbests(k).pieces(i).id());
      ++total;
    }
  }
  offs[nb.nbests_size()] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  return nb.nbests_size();
}

// Switches a unigram model to EncoderVersion::kOriginal (unigram_model.h:176-186): a fresh unigram::Model over the
// processor's own ModelProto, installed with SetModel (sentencepiece_processor.h:683).  -1 if the model is not unigram.
int spmref_set_encoder_original(void *handle) {
  auto *h = static_cast<RefHandle *>(handle);
  const auto &mp = h->sp.model_proto();
  if (mp.trainer_spec().model_type() != sentencepiece::TrainerSpec::UNIGRAM) return -1;
  auto m = std::make_unique<sentencepiece::unigram::Model>(mp);
  m->SetEncoderVersion(sentencepiece::unigram::Model::kOriginal);
  // the normalizer points at the model's prefix matcher (sentencepiece_processor.cc Load): rebuild it on the new model
  auto nz = std::make_unique<sentencepiece::normalizer::Normalizer>(mp.normalizer_spec(), mp.trainer_spec());
  nz->SetPrefixMatcher(m->prefix_matcher());
  h->sp.SetNormalizer(std::move(nz));
  h->sp.SetModel(std::move(m));
  return 0;
}

// SampleEncode(input, nbest_size, alpha, &ids) (sentencepiece_processor.h:333-334).  Returns the id count.
int64_t spmref_sample_encode(void *handle, const char *text, uint64_t len, int nbest_size, float alpha, int32_t *ids,
                             uint64_t cap) {
  auto *h = static_cast<RefHandle *>(handle);
  std::vector<int> v;
  const auto st = h->sp.SampleEncode(absl::string_view(text, len), nbest_size, alpha, &v);
  if (!st.ok()) { h->last_error = st.ToString(); return -1; }
  for (size_t i = 0; i < v.size() && i < cap; ++i) ids[i] = v[i];
  return static_cast<int64_t>(v.size());
}

// EncodeAsSerializedProto(input) per sentence (sentencepiece_processor.h:528-531): the serialized SentencePieceText
// messages back to back, message i at out[out_offs[i], out_offs[i + 1]).
int64_t spmref_encode_serialized_batch(void *handle, const char *text, const uint64_t *offsets, uint64_t n, char *out,
                                       uint64_t cap, uint64_t *out_offs) {
  auto *h = static_cast<RefHandle *>(handle);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    out_offs[i] = total;
    const std::string s = h->sp.EncodeAsSerializedProto(absl::string_view(text + offsets[i], offsets[i + 1] - offsets[i]));
    if (total + s.size() <= cap && !s.empty()) memcpy(out + total, s.data(), s.size());
    total += s.size();
  }
  out_offs[n] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  return static_cast<int64_t>(total);
}

// EncodeAsImmutableProto(input) + ConvertToUnicodeSpans() (sentencepiece_processor.h:573-576, .cc:63-89, :168-170):
// what the reference's Python wrapper returns for out_type="immutable_proto": begin / end in characters.
int64_t spmref_encode_unicode_spans_batch(void *handle, const char *text, const uint64_t *offsets, uint64_t n,
                                          uint32_t *begin, uint32_t *end, uint64_t cap, uint64_t *id_offsets) {
  auto *h = static_cast<RefHandle *>(handle);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    id_offsets[i] = total;
    auto spt = h->sp.EncodeAsImmutableProto(absl::string_view(text + offsets[i], offsets[i + 1] - offsets[i]));
    spt.ConvertToUnicodeSpans();
    for (size_t k = 0; k < spt.pieces_size(); ++k) {
      if (total < cap) { begin[total] = spt.pieces(k).begin(); end[total] = spt.pieces(k).end(); }
      ++total;
    }
  }
  id_offsets[n] = total;
  if (total > cap) return -static_cast<int64_t>(total) - 2;
  return static_cast<int64_t>(total);
}

// Decode(const std::vector<std::string>& pieces, std::string*) (sentencepiece_processor.h:303-305): the pieces of ONE
// sentence packed as pieces[offsets[i], offsets[i + 1]).  Returns the text's length, -1 on a Status error,
// -(needed) - 2 if cap is too small.
int64_t spmref_decode_pieces(void *handle, const char *pieces, const uint64_t *offsets, uint64_t n, char *out, uint64_t cap) {
  auto *h = static_cast<RefHandle *>(handle);
  std::vector<std::string> v;
  v.reserve(n);
  for (uint64_t i = 0; i < n; ++i) v.emplace_back(pieces + offsets[i], offsets[i + 1] - offsets[i]);
  std::string text;
  const auto st = h->sp.Decode(v, &text);
  if (!st.ok()) {
    h->last_error = st.ToString();
    return -1;
  }
  if (text.size() > cap) return -static_cast<int64_t>(text.size()) - 2;
  memcpy(out, text.data(), text.size());
  return static_cast<int64_t>(text.size());
}
// GetScore(id) (sentencepiece_processor.h:650)
float spmref_get_score(void *handle, int id) { return static_cast<RefHandle *>(handle)->sp.GetScore(id); }

int spmref_piece_size(void *handle) {
  return static_cast<RefHandle *>(handle)->sp.GetPieceSize();
}

}  // extern "C"
