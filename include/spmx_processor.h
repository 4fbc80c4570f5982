// C++ host facade over the C ABI (spmx.h): the reference's
// sentencepiece::SentencePieceProcessor, restricted to the text -> ids path,
// with the same method names, argument meaning and error convention
// (util::Status values, no exceptions; reference: src/sentencepiece_processor.h
// :34-76 Status, :245 Load, :261 LoadFromSerializedProto, :264 status,
// :267 SetEncodeExtraOptions, :279-283 SetVocabulary / ResetVocabulary,
// :299-300 Encode(input, vector<int>*), :303-312 Decode from pieces / ids, :458-460 EncodeAsIds, :528
// EncodeAsSerializedProto, :288 LoadVocabulary, :638-677 vocabulary accessors incl. GetScore, :694
// serialized_model_proto; the methods the reference declares virtual (:245-677) are virtual here), plus the batch form the reference only has in its
// Python wrapper (python/src/sentencepiece/sentencepiece.i:439-446
// _EncodeAsIdsBatch): EncodeBatch == element-wise Encode.
//
// Header-only; link with libspmx.so.  An application written against the
// reference switches by including this header and
//   namespace sentencepiece = sentencepiece_amd;
#ifndef SPMX_PROCESSOR_H_
#define SPMX_PROCESSOR_H_
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <random>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "spmx.h"

namespace sentencepiece_amd {
namespace util {

enum class StatusCode : int {   // src/sentencepiece_processor.h:34-52
  kOk = 0, kCancelled = 1, kUnknown = 2, kInvalidArgument = 3, kDeadlineExceeded = 4, kNotFound = 5,
  kAlreadyExists = 6, kPermissionDenied = 7, kResourceExhausted = 8, kFailedPrecondition = 9, kAborted = 10,
  kOutOfRange = 11, kUnimplemented = 12, kInternal = 13, kUnavailable = 14, kDataLoss = 15, kUnauthenticated = 16,
};

class Status {
 public:
  Status() = default;
  Status(StatusCode code, std::string_view msg) : code_(code), msg_(msg) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const char *error_message() const { return msg_.c_str(); }
  std::string ToString() const {   // src/error.cc:62-
    if (ok()) return "OK";
    static const char *const kNames[] = {"OK", "Cancelled", "Unknown", "Invalid argument", "Deadline exceeded",
                                         "Not found", "Already exists", "Permission denied", "Resource exhausted",
                                         "Failed precondition", "Aborted", "Out of range", "Unimplemented",
                                         "Internal", "Unavailable", "Data loss", "Unauthenticated"};
    const int c = static_cast<int>(code_);
    return std::string(c >= 0 && c <= 16 ? kNames[c] : "Unknown") + ": " + msg_;
  }

 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};

}  // namespace util

// The fields of the reference's SentencePieceText message (src/sentencepiece.proto) that Encode(input, spt) fills,
// as a plain struct: text, and per piece its string, id, surface and byte range of the input.
struct SentencePieceText {
  struct SentencePiece {
    std::string piece;
    uint32_t id = 0;
    std::string surface;
    uint32_t begin = 0, end = 0;
    // proto2 presence of `surface`: PopulateSentencePieceText sets it on every piece except the byte-fallback pieces before
    // the last one of their character (src/sentencepiece_processor.cc:598-603); a bos / eos carries none (:1029-1048)
    bool has_surface = true;
  };
  std::string text;
  std::vector<SentencePiece> pieces;
  float score = 0.f;       // set by NBestEncode only (src/sentencepiece_processor.cc:670)
  bool has_score = false;
  // The message's wire format (src/sentencepiece.proto:25-65), what SerializeAsString() of the reference's proto gives:
  // text = 1, pieces = 2 {piece = 1, id = 2, surface = 3, begin = 4, end = 5}, score = 3 -- fields in number order.
  std::string SerializeAsString() const {
    auto varint = [](std::string *o, uint64_t v) { while (v >= 0x80) { o->push_back(static_cast<char>(v | 0x80)); v >>= 7; } o->push_back(static_cast<char>(v)); };
    auto bytes_field = [&](std::string *o, int num, const std::string &b) { varint(o, static_cast<uint64_t>(num) << 3 | 2); varint(o, b.size()); o->append(b); };
    auto uint_field = [&](std::string *o, int num, uint32_t v) { varint(o, static_cast<uint64_t>(num) << 3); varint(o, v); };
    std::string out;
    bytes_field(&out, 1, text);
    for (const SentencePiece &p : pieces) {
      std::string m;
      bytes_field(&m, 1, p.piece);
      uint_field(&m, 2, p.id);
      if (p.has_surface) bytes_field(&m, 3, p.surface);
      uint_field(&m, 4, p.begin);
      uint_field(&m, 5, p.end);
      bytes_field(&out, 2, m);
    }
    if (has_score) {
      out.push_back(static_cast<char>(3 << 3 | 5));
      char f[4];
      memcpy(f, &score, 4);
      out.append(f, 4);
    }
    return out;
  }
};
// NBestSentencePieceText (src/sentencepiece.proto): the results of NBestEncode, best first
struct NBestSentencePieceText {
  std::vector<SentencePieceText> nbests;
};

// An encoded batch as the library hands it over: the CSR arrays themselves (pinned host memory for big batches), owned
// by this object and released with spmx_free -- no copy into containers.  Sentence i is ids[id_offsets[i] .. id_offsets[i + 1]).
// (The vector forms below copy: 10 M sentences into std::vector<std::vector<int>> cost more host time than the device path.)
class EncodedBatch {
 public:
  EncodedBatch() = default;
  ~EncodedBatch() { reset(); }
  EncodedBatch(const EncodedBatch &) = delete;
  EncodedBatch &operator=(const EncodedBatch &) = delete;
  EncodedBatch(EncodedBatch &&o) noexcept : ids_(o.ids_), offs_(o.offs_), n_(o.n_) { o.ids_ = nullptr; o.offs_ = nullptr; o.n_ = 0; }
  EncodedBatch &operator=(EncodedBatch &&o) noexcept {
    if (this != &o) { reset(); ids_ = o.ids_; offs_ = o.offs_; n_ = o.n_; o.ids_ = nullptr; o.offs_ = nullptr; o.n_ = 0; }
    return *this;
  }
  uint64_t size() const { return n_; }                                  // sentences
  uint64_t total_ids() const { return offs_ ? offs_[n_] : 0; }
  const int32_t *ids() const { return ids_; }
  const uint64_t *id_offsets() const { return offs_; }
  const int32_t *begin(uint64_t i) const { return ids_ + offs_[i]; }
  const int32_t *end(uint64_t i) const { return ids_ + offs_[i + 1]; }
  uint64_t length(uint64_t i) const { return offs_[i + 1] - offs_[i]; }
  void reset() { spmx_free(ids_); spmx_free(offs_); ids_ = nullptr; offs_ = nullptr; n_ = 0; }
  void adopt(int32_t *ids, uint64_t *offs, uint64_t n) { reset(); ids_ = ids; offs_ = offs; n_ = n; }

 private:
  int32_t *ids_ = nullptr;
  uint64_t *offs_ = nullptr;
  uint64_t n_ = 0;
};

class SentencePieceProcessor {
 public:
  explicit SentencePieceProcessor(int device = 0) : device_(device) {}
  virtual ~SentencePieceProcessor() { spmx_destroy(h_); }
  SentencePieceProcessor(const SentencePieceProcessor &) = delete;
  SentencePieceProcessor &operator=(const SentencePieceProcessor &) = delete;

  virtual util::Status Load(std::string_view filename) {
    Reset();
    const std::string f(filename);
    return Created(spmx_create_from_file(f.c_str(), device_, &h_));
  }
  virtual util::Status LoadFromSerializedProto(std::string_view serialized) {
    Reset();
    return Created(spmx_create(serialized.data(), serialized.size(), device_, &h_));
  }
  void LoadOrDie(std::string_view filename) {
    if (!Load(filename).ok()) std::abort();
  }
  util::Status status() const {
    return h_ ? util::Status() : util::Status(util::StatusCode::kInternal, "Model is not initialized.");
  }

  // :270 SetDecodeExtraOptions (.cc:288-291): applied to the pieces of every later Decode before they become text (.cc:819)
  virtual util::Status SetDecodeExtraOptions(std::string_view extra_option) {
    if (!h_) return status();
    const std::string o(extra_option);
    return FromHandle(spmx_set_decode_extra_options(h_, o.c_str()));
  }
  virtual util::Status SetEncodeExtraOptions(std::string_view extra_option) {
    if (!h_) return status();
    const std::string o(extra_option);
    const int rc = spmx_set_encode_extra_options(h_, o.c_str());
    if (rc == 0) {
      unk_piece_option_ = false;
      reverse_option_ = false;
      for (size_t p = 0; p <= o.size();) {
        const size_t q = o.find(':', p);
        const std::string f = o.substr(p, q == std::string::npos ? std::string::npos : q - p);
        if (f == "unk" || f == "unk_piece") unk_piece_option_ = true;
        if (f == "reverse") reverse_option_ = !reverse_option_;
        if (q == std::string::npos) break;
        p = q + 1;
      }
    }
    return FromHandle(rc);
  }
  virtual util::Status SetVocabulary(const std::vector<std::string_view> &valid_vocab) {
    if (!h_) return status();
    std::vector<const char *> p;
    std::vector<uint64_t> l;
    for (auto v : valid_vocab) { p.push_back(v.data()); l.push_back(v.size()); }
    return FromHandle(spmx_set_vocabulary(h_, p.data(), l.data(), p.size()));
  }
  virtual util::Status ResetVocabulary() { return h_ ? FromHandle(spmx_reset_vocabulary(h_)) : status(); }
  // LoadVocabulary (sentencepiece_processor.h:288, .cc:341-362): "<token> TAB <freq>" lines; the tokens whose frequency
  // reaches `threshold` (1 where the line has no second column) become the valid vocabulary
  virtual util::Status LoadVocabulary(std::string_view filename, int threshold) {
    const std::string fn(filename);
    FILE *f = fopen(fn.c_str(), "rb");
    if (!f) return util::Status(util::StatusCode::kNotFound, "\"" + fn + "\": No such file or directory");
    std::string data;
    char buf[65536];
    for (size_t r; (r = fread(buf, 1, sizeof(buf), f)) > 0;) data.append(buf, r);
    fclose(f);
    std::vector<std::string> vocab;
    for (size_t p = 0; p < data.size();) {
      size_t q = data.find('\n', p);
      if (q == std::string::npos) q = data.size();
      std::string line = data.substr(p, q - p);
      if (!line.empty() && line.back() == '\r') line.pop_back();
      p = q + 1;
      const size_t tab = line.find('\t');
      const std::string tok = line.substr(0, tab);
      if (tok.empty()) return util::Status(util::StatusCode::kInternal, "LoadVocabulary: an empty token");   // CHECK_OR_RETURN(!v[0].empty())
      long freq = 1;
      if (tab != std::string::npos) {
        const size_t tab2 = line.find('\t', tab + 1);
        const std::string num = line.substr(tab + 1, tab2 == std::string::npos ? std::string::npos : tab2 - tab - 1);
        char *end = nullptr;
        freq = strtol(num.c_str(), &end, 10);
        if (num.empty() || !end || *end != 0) return util::Status(util::StatusCode::kInternal, "Could not parse the frequency");
      }
      if (freq >= threshold) vocab.push_back(tok);
    }
    std::vector<std::string_view> v(vocab.begin(), vocab.end());
    return SetVocabulary(v);
  }

  // ---- encode ----
  virtual util::Status Encode(std::string_view input, std::vector<int> *ids) const {
    if (!h_) return status();
    if (!ids) return util::Status(util::StatusCode::kInternal, "output container is null");
    ids->clear();
    int32_t *out = nullptr;
    uint64_t *offs = nullptr;
    uint8_t *st = nullptr;
    const uint64_t io[2] = {0, input.size()};
    const int rc = spmx_encode_batch_ex(h_, input.data() ? input.data() : "", io, 1, &out, &offs, &st, nullptr);
    if (rc != 0) return FromHandle(rc);
    // the sentence's own Status, as the reference returns it (sentencepiece_processor.cc:392-403 -> :628)
    const int code = st ? st[0] : 0;
    if (code == 0) ids->assign(out, out + offs[1]);
    spmx_free(out);
    spmx_free(offs);
    spmx_free(st);
    return code == 0 ? util::Status() : util::Status(static_cast<util::StatusCode>(code), spmx_status_message(code));
  }
  virtual std::vector<int> EncodeAsIds(std::string_view input) const {   // errors are swallowed, as in the reference (:427-436)
    std::vector<int> ids;
    (void)Encode(input, &ids);
    return ids;
  }
  // Flat form: packed text + offsets in, CSR out (ids and id_offsets are resized).
  util::Status EncodeBatchFlat(const char *text, const uint64_t *offsets, uint64_t n, std::vector<int32_t> *ids,
                               std::vector<uint64_t> *id_offsets) const {
    if (!h_) return status();
    if (!ids || !id_offsets) return util::Status(util::StatusCode::kInternal, "output container is null");
    ids->clear();
    id_offsets->clear();
    int32_t *out = nullptr;
    uint64_t *offs = nullptr;
    const int rc = spmx_encode_batch(h_, text, offsets, n, &out, &offs);
    if (rc != 0) return FromHandle(rc);
    id_offsets->assign(offs, offs + n + 1);
    ids->assign(out, out + offs[n]);
    spmx_free(out);
    spmx_free(offs);
    return util::Status();
  }
  // The same, zero-copy: the library's own CSR arrays, owned by *out (the C ABI's rate: no container is filled).
  util::Status EncodeBatchFlat(const char *text, const uint64_t *offsets, uint64_t n, EncodedBatch *out) const {
    if (!h_) return status();
    if (!out) return util::Status(util::StatusCode::kInternal, "output container is null");
    out->reset();
    int32_t *ids = nullptr;
    uint64_t *offs = nullptr;
    const int rc = spmx_encode_batch(h_, text, offsets, n, &ids, &offs);
    if (rc != 0) return FromHandle(rc);
    out->adopt(ids, offs, n);
    return util::Status();
  }
  util::Status EncodeBatch(const std::vector<std::string_view> &ins, EncodedBatch *out) const {
    if (!h_) return status();
    if (!out) return util::Status(util::StatusCode::kInternal, "output container is null");
    out->reset();
    std::vector<spmx_view> views(ins.size());
    for (size_t i = 0; i < ins.size(); ++i) views[i] = spmx_view{ins[i].data(), ins[i].size()};
    int32_t *ids = nullptr;
    uint64_t *io = nullptr;
    const int rc = spmx_encode_batch_views(h_, views.data(), views.size(), &ids, &io, nullptr, nullptr);
    if (rc != 0) return FromHandle(rc);
    out->adopt(ids, io, ins.size());
    return util::Status();
  }
  // Element-wise identical to Encode() per input (sentencepiece.i:245-267): a failing element yields an empty id list
  // (the reference's workers call EncodeAsIds, which drops the Status).  The sentences are handed over as (pointer,
  // length) pairs: the library gathers them into its pinned staging buffers chunk by chunk, no packed copy is made here.
  util::Status EncodeBatch(const std::vector<std::string_view> &ins, std::vector<std::vector<int>> *outs) const {
    if (!h_) return status();
    if (!outs) return util::Status(util::StatusCode::kInternal, "output container is null");
    outs->clear();
    std::vector<spmx_view> views(ins.size());
    for (size_t i = 0; i < ins.size(); ++i) views[i] = spmx_view{ins[i].data(), ins[i].size()};
    int32_t *ids = nullptr;
    uint64_t *io = nullptr;
    const int rc = spmx_encode_batch_views(h_, views.data(), views.size(), &ids, &io, nullptr, nullptr);
    if (rc != 0) return FromHandle(rc);
    outs->resize(ins.size());
    for (size_t i = 0; i < ins.size(); ++i) (*outs)[i].assign(ids + io[i], ids + io[i + 1]);
    spmx_free(ids);
    spmx_free(io);
    return util::Status();
  }
  // One batch over several GPUs of the node: procs[g] holds the same model on GPU g (constructed with that device
  // ordinal).  The library deals the batch's chunks round-robin over the GPUs and every GPU writes its ids into the one
  // CSR returned here (spmx_encode_batch_multi): the single-process form of the sharded encode.
  static util::Status EncodeBatchSharded(const std::vector<const SentencePieceProcessor *> &procs, const char *text,
                                         const uint64_t *offsets, uint64_t n, std::vector<int32_t> *ids,
                                         std::vector<uint64_t> *id_offsets) {
    if (procs.empty() || !procs[0] || !procs[0]->h_) return util::Status(util::StatusCode::kInternal, "Model is not initialized.");
    if (!ids || !id_offsets) return util::Status(util::StatusCode::kInternal, "output container is null");
    std::vector<spmx_handle *> hs;
    for (const SentencePieceProcessor *p : procs) {
      if (!p || !p->h_) return util::Status(util::StatusCode::kInternal, "Model is not initialized.");
      hs.push_back(p->h_);
    }
    int32_t *out = nullptr;
    uint64_t *offs = nullptr;
    const int rc = spmx_encode_batch_multi(hs.data(), static_cast<int>(hs.size()), text, offsets, n, &out, &offs, nullptr, nullptr);
    if (rc != 0) return procs[0]->FromHandle(rc);
    id_offsets->assign(offs, offs + n + 1);
    ids->assign(out, out + offs[n]);
    spmx_free(out);
    spmx_free(offs);
    return util::Status();
  }
  // Device-resident form (HIP pointers, see spmx_encode_batch_device).
  util::Status EncodeBatchDevice(const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets, uint64_t n,
                                 int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets, void *stream,
                                 uint64_t *total_ids) const {
    if (!h_) return status();
    return FromHandle(spmx_encode_batch_device(h_, d_text, text_bytes, d_offsets, n, d_ids, ids_capacity,
                                               d_id_offsets, stream, total_ids));
  }

  // One process per GPU: every rank's device CSR (what EncodeBatchDevice wrote for its contiguous shard of the job's
  // sentences) on every rank, over RCCL (spmx_all_gather_ids: counts all-gather, exact-size grouped sends and receives,
  // offsets rebased to the whole job).  comm: an ncclComm_t (the caller's, or spmx_rccl_comm_init's); d_scratch:
  // GatherScratchWords(world) uint64 of device memory; rank_sentences / rank_ids (nullable): prefix sums over the ranks, world + 1 entries.
  static uint64_t GatherScratchWords(int world) { return spmx_gather_scratch_words(world); }
  static util::Status AllGatherIds(void *comm, int rank, int world, const int32_t *d_ids, uint64_t n_ids,
                                   const uint64_t *d_id_offsets, uint64_t n_sentences, int32_t *d_all_ids,
                                   uint64_t all_ids_capacity, uint64_t *d_all_id_offsets, uint64_t all_offsets_capacity,
                                   uint64_t *d_scratch, std::vector<uint64_t> *rank_sentences, std::vector<uint64_t> *rank_ids,
                                   void *stream) {
    if (rank_sentences) rank_sentences->assign(static_cast<size_t>(world > 0 ? world : 0) + 1, 0);
    if (rank_ids) rank_ids->assign(static_cast<size_t>(world > 0 ? world : 0) + 1, 0);
    const int rc = spmx_all_gather_ids(comm, rank, world, d_ids, n_ids, d_id_offsets, n_sentences, d_all_ids, all_ids_capacity,
                                       d_all_id_offsets, all_offsets_capacity, d_scratch,
                                       rank_sentences ? rank_sentences->data() : nullptr, rank_ids ? rank_ids->data() : nullptr, stream);
    if (rc == 0) return util::Status();
    return util::Status(static_cast<util::StatusCode>(rc), spmx_gather_last_error());
  }

  // ---- pieces / SentencePieceText (sentencepiece_processor.h:295-296, :401-402, :453-456) ----
  // The device returns ids, input spans and normalized-text spans (spmx_encode_batch_spans) and the normalized text
  // (spmx_normalize_batch); a piece is its normalized text, the piece name for a byte-fallback piece and a bos / eos,
  // or unk_piece for an unknown token under the `unk_piece` extra option (sentencepiece_processor.cc:547-636, :1019-1064).
  virtual util::Status Encode(std::string_view input, SentencePieceText *spt) const {
    if (!h_) return status();
    if (!spt) return util::Status(util::StatusCode::kInternal, "output proto is null");   // CHECK_OR_RETURN_STATUS_PROTO
    spt->text.clear();
    spt->pieces.clear();
    const uint64_t offs[2] = {0, input.size()};
    int32_t *ids = nullptr;
    uint64_t *io = nullptr, *no = nullptr;
    uint32_t *b = nullptr, *e = nullptr, *nb = nullptr, *ne = nullptr;
    uint8_t *st = nullptr;
    char *norm = nullptr;
    int rc = spmx_encode_batch_spans_ex(h_, input.data() ? input.data() : "", offs, 1, &ids, &io, &b, &e, &nb, &ne, &st, nullptr);
    const int code = rc == 0 && st ? st[0] : 0;     // the sentence's own Status (sentencepiece_processor.cc:638-651 -> :628)
    spmx_free(st);
    if (code != 0) {
      spmx_free(ids); spmx_free(io); spmx_free(b); spmx_free(e); spmx_free(nb); spmx_free(ne);
      return util::Status(static_cast<util::StatusCode>(code), spmx_status_message(code));
    }
    if (rc == 0) rc = spmx_normalize_batch(h_, input.data() ? input.data() : "", offs, 1, &norm, &no, nullptr);
    if (rc == 0) {
      FillPieces(input, norm, ids, b, e, nb, ne, 0, io[1], spt);
    }
    spmx_free(ids); spmx_free(io); spmx_free(b); spmx_free(e); spmx_free(nb); spmx_free(ne); spmx_free(norm); spmx_free(no);
    return FromHandle(rc);
  }
  virtual util::Status Encode(std::string_view input, std::vector<std::string> *pieces) const {
    if (!h_) return status();
    if (!pieces) return util::Status(util::StatusCode::kInternal, "output container is null");
    pieces->clear();
    SentencePieceText spt;
    const util::Status st = Encode(input, &spt);
    if (!st.ok()) return st;
    for (auto &p : spt.pieces) pieces->push_back(std::move(p.piece));
    return util::Status();
  }
  virtual std::vector<std::string> EncodeAsPieces(std::string_view input) const {   // errors are swallowed, as in the reference
    std::vector<std::string> pieces;
    (void)Encode(input, &pieces);
    return pieces;
  }

  // ---- n-best (sentencepiece_processor.h:323-324; unigram models) ----
  virtual util::Status NBestEncode(std::string_view input, int nbest_size, std::vector<std::vector<int>> *ids) const {
    if (!h_) return status();
    if (!ids) return util::Status(util::StatusCode::kInternal, "output container is null");
    ids->clear();
    const uint64_t offs[2] = {0, input.size()};
    int32_t *i = nullptr;
    uint64_t *io = nullptr, *ro = nullptr;
    float *sc = nullptr;
    const int rc = spmx_nbest_encode_batch(h_, input.data() ? input.data() : "", offs, 1, nbest_size, &i, &io, &sc, &ro);
    if (rc == 0)
      for (uint64_t r = ro[0]; r < ro[1]; ++r) ids->emplace_back(i + io[r], i + io[r + 1]);
    spmx_free(i); spmx_free(io); spmx_free(sc); spmx_free(ro);
    return FromHandle(rc);
  }
  std::vector<std::vector<int>> NBestEncodeAsIds(std::string_view input, int nbest_size) const {   // errors are swallowed
    std::vector<std::vector<int>> ids;
    (void)NBestEncode(input, nbest_size, &ids);
    return ids;
  }
  // NBestEncode(input, nbest_size, NBestSentencePieceText *) (sentencepiece_processor.h:323-324, .cc:653-676): every
  // result with its score and the pieces / surfaces / byte ranges PopulateSentencePieceText gives it
  virtual util::Status NBestEncode(std::string_view input, int nbest_size, NBestSentencePieceText *nbest_spt) const {
    if (!h_) return status();
    if (!nbest_spt) return util::Status(util::StatusCode::kInternal, "output proto is null");
    nbest_spt->nbests.clear();
    const uint64_t offs[2] = {0, input.size()};
    int32_t *ids = nullptr;
    uint64_t *io = nullptr, *ro = nullptr, *no = nullptr;
    float *sc = nullptr;
    uint32_t *b = nullptr, *e = nullptr, *nb = nullptr, *ne = nullptr;
    char *norm = nullptr;
    const char *txt = input.data() ? input.data() : "";
    int rc = spmx_nbest_encode_batch_spans(h_, txt, offs, 1, nbest_size, &ids, &io, &sc, &ro, &b, &e, &nb, &ne);
    if (rc == 0) rc = spmx_normalize_batch(h_, txt, offs, 1, &norm, &no, nullptr);
    if (rc == 0) {
      for (uint64_t r = ro[0]; r < ro[1]; ++r) {
        nbest_spt->nbests.emplace_back();
        nbest_spt->nbests.back().score = sc[r];
        nbest_spt->nbests.back().has_score = true;
        FillPieces(input, norm, ids, b, e, nb, ne, io[r], io[r + 1], &nbest_spt->nbests.back());
      }
    }
    spmx_free(ids); spmx_free(io); spmx_free(sc); spmx_free(ro); spmx_free(b); spmx_free(e); spmx_free(nb); spmx_free(ne);
    spmx_free(norm); spmx_free(no);
    return FromHandle(rc);
  }
  virtual util::Status NBestEncode(std::string_view input, int nbest_size, std::vector<std::vector<std::string>> *pieces) const {
    if (!h_) return status();
    if (!pieces) return util::Status(util::StatusCode::kInternal, "output container is null");
    pieces->clear();
    NBestSentencePieceText nb;
    const util::Status st = NBestEncode(input, nbest_size, &nb);
    if (!st.ok()) return st;
    for (auto &spt : nb.nbests) {
      pieces->emplace_back();
      for (auto &p : spt.pieces) pieces->back().push_back(std::move(p.piece));
    }
    return util::Status();
  }
  std::vector<std::vector<std::string>> NBestEncodeAsPieces(std::string_view input, int nbest_size) const {   // errors are swallowed
    std::vector<std::vector<std::string>> pieces;
    (void)NBestEncode(input, nbest_size, &pieces);
    return pieces;
  }

  // ---- sampling (sentencepiece_processor.h:346-353, .cc:678-720): lattice sampling / n-best sampling (unigram),
  // BPE-dropout (BPE).  The draws are keyed by (seed, sentence); each call without a seed of its own takes the next
  // value of a per-process counter, so repeated calls draw afresh as the reference's thread-local generator does ----
  virtual util::Status SampleEncode(std::string_view input, int nbest_size, float alpha, std::vector<int> *ids) const {
    static std::atomic<uint64_t> calls{std::random_device{}()};   // (the reference seeds from std::random_device, src/util.cc:202-204)
    return SampleEncode(input, nbest_size, alpha, ++calls, ids);
  }
  util::Status SampleEncode(std::string_view input, int nbest_size, float alpha, uint64_t seed, std::vector<int> *ids) const {
    if (!h_) return status();
    if (!ids) return util::Status(util::StatusCode::kInternal, "output container is null");
    ids->clear();
    const uint64_t offs[2] = {0, input.size()};
    int32_t *i = nullptr;
    uint64_t *io = nullptr;
    const int rc = spmx_sample_encode_batch(h_, input.data() ? input.data() : "", offs, 1, nbest_size, alpha, seed, &i, &io);
    if (rc == 0) ids->assign(i + io[0], i + io[1]);
    spmx_free(i); spmx_free(io);
    return FromHandle(rc);
  }
  std::vector<int> SampleEncodeAsIds(std::string_view input, int nbest_size, float alpha) const {   // errors are swallowed
    std::vector<int> ids;
    (void)SampleEncode(input, nbest_size, alpha, &ids);
    return ids;
  }
  // SampleEncode(input, nbest_size, alpha, SentencePieceText *) (sentencepiece_processor.h:346-348, .cc:678-720)
  virtual util::Status SampleEncode(std::string_view input, int nbest_size, float alpha, SentencePieceText *spt) const {
    static std::atomic<uint64_t> calls{std::random_device{}()};
    return SampleEncode(input, nbest_size, alpha, ++calls, spt);
  }
  util::Status SampleEncode(std::string_view input, int nbest_size, float alpha, uint64_t seed, SentencePieceText *spt) const {
    if (!h_) return status();
    if (!spt) return util::Status(util::StatusCode::kInternal, "output proto is null");
    spt->text.clear();
    spt->pieces.clear();
    const uint64_t offs[2] = {0, input.size()};
    int32_t *ids = nullptr;
    uint64_t *io = nullptr, *no = nullptr;
    uint32_t *b = nullptr, *e = nullptr, *nb = nullptr, *ne = nullptr;
    char *norm = nullptr;
    const char *txt = input.data() ? input.data() : "";
    int rc = spmx_sample_encode_batch_spans(h_, txt, offs, 1, nbest_size, alpha, seed, &ids, &io, &b, &e, &nb, &ne);
    if (rc == 0) rc = spmx_normalize_batch(h_, txt, offs, 1, &norm, &no, nullptr);
    if (rc == 0) FillPieces(input, norm, ids, b, e, nb, ne, io[0], io[1], spt);
    spmx_free(ids); spmx_free(io); spmx_free(b); spmx_free(e); spmx_free(nb); spmx_free(ne); spmx_free(norm); spmx_free(no);
    return FromHandle(rc);
  }
  virtual util::Status SampleEncode(std::string_view input, int nbest_size, float alpha, std::vector<std::string> *pieces) const {
    if (!h_) return status();
    if (!pieces) return util::Status(util::StatusCode::kInternal, "output container is null");
    pieces->clear();
    SentencePieceText spt;
    const util::Status st = SampleEncode(input, nbest_size, alpha, &spt);
    if (!st.ok()) return st;
    for (auto &p : spt.pieces) pieces->push_back(std::move(p.piece));
    return util::Status();
  }
  std::vector<std::string> SampleEncodeAsPieces(std::string_view input, int nbest_size, float alpha) const {   // errors are swallowed
    std::vector<std::string> pieces;
    (void)SampleEncode(input, nbest_size, alpha, &pieces);
    return pieces;
  }
  // batch form: sentence i draws from the generator keyed by (seed, i)
  util::Status SampleEncodeBatchFlat(const char *text, const uint64_t *offsets, uint64_t n, int nbest_size, float alpha,
                                     uint64_t seed, std::vector<int32_t> *ids, std::vector<uint64_t> *id_offsets) const {
    if (!h_) return status();
    if (!ids || !id_offsets) return util::Status(util::StatusCode::kInternal, "output container is null");
    int32_t *i = nullptr;
    uint64_t *io = nullptr;
    const int rc = spmx_sample_encode_batch(h_, text, offsets, n, nbest_size, alpha, seed, &i, &io);
    if (rc == 0) { id_offsets->assign(io, io + n + 1); ids->assign(i, i + io[n]); }
    spmx_free(i); spmx_free(io);
    return FromHandle(rc);
  }
  // The reference's kOriginal unigram encoder (Lattice::Viterbi, src/unigram_model.cc:161-198, :674-692; selected
  // there by unigram::Model::SetEncoderVersion, src/unigram_model.h:176-186).
  util::Status EncodeOriginal(std::string_view input, std::vector<int> *ids) const {
    if (!h_) return status();
    if (!ids) return util::Status(util::StatusCode::kInternal, "output container is null");
    ids->clear();
    const uint64_t offs[2] = {0, input.size()};
    int32_t *i = nullptr;
    uint64_t *io = nullptr;
    const int rc = spmx_encode_batch_original(h_, input.data() ? input.data() : "", offs, 1, &i, &io);
    if (rc == 0) ids->assign(i + io[0], i + io[1]);
    spmx_free(i); spmx_free(io);
    return FromHandle(rc);
  }

  // ---- Normalize (sentencepiece_processor.h:622-631) ----
  virtual util::Status Normalize(std::string_view input, std::string *normalized, std::vector<size_t> *norm_to_orig) const {
    if (!h_) return status();
    if (!normalized || !norm_to_orig) return util::Status(util::StatusCode::kInternal, "output container is null");
    normalized->clear();
    norm_to_orig->clear();
    const uint64_t offs[2] = {0, input.size()};
    char *norm = nullptr;
    uint64_t *no = nullptr;
    uint32_t *a = nullptr;
    const int rc = spmx_normalize_batch(h_, input.data() ? input.data() : "", offs, 1, &norm, &no, &a);
    if (rc == 0) {
      normalized->assign(norm, no[1]);
      if (!(no[1] == 0 && a[0] == 0xFFFFFFFFu)) norm_to_orig->assign(a, a + no[1] + 1);
    }
    spmx_free(norm); spmx_free(no); spmx_free(a);
    return FromHandle(rc);
  }
  virtual util::Status Normalize(std::string_view input, std::string *normalized) const {
    std::vector<size_t> a;
    return Normalize(input, normalized, &a);
  }
  std::string Normalize(std::string_view input) const {
    std::string out;
    (void)Normalize(input, &out);
    return out;
  }

  // ---- decode (sentencepiece_processor.h:311-312, :515-517) ----
  virtual util::Status Decode(const std::vector<int> &ids, std::string *detokenized) const {
    if (!h_) return status();
    if (!detokenized) return util::Status(util::StatusCode::kInternal, "output container is null");
    detokenized->clear();
    char *text = nullptr;
    uint64_t *offs = nullptr;
    const uint64_t io[2] = {0, ids.size()};
    static_assert(sizeof(int) == sizeof(int32_t), "ids are 32-bit");
    const int32_t none = 0;
    const int rc = spmx_decode_batch(h_, ids.empty() ? &none : reinterpret_cast<const int32_t *>(ids.data()), io, 1, &text, &offs);
    if (rc != 0) return FromHandle(rc);
    detokenized->assign(text, text + offs[1]);
    spmx_free(text);
    spmx_free(offs);
    return util::Status();
  }
  std::string DecodeIds(const std::vector<int> &ids) const {   // errors are swallowed, as in the reference
    std::string out;
    (void)Decode(ids, &out);
    return out;
  }
  // Decode from pieces (sentencepiece_processor.h:303-309, .cc:761-769): every piece by PieceToId; a piece that is not in
  // the vocabulary is copied through as it is (.cc:784-790) -- it travels to the decode kernels as a literal beside the ids
  // (spmx_decode_batch_pieces) -- unless a decode extra option `unk` turns it into the unknown piece first (.cc:1050-1058)
  virtual util::Status Decode(const std::vector<std::string_view> &pieces, std::string *detokenized) const {
    if (!h_) return status();
    if (!detokenized) return util::Status(util::StatusCode::kInternal, "output container is null");
    detokenized->clear();
    const int unk = unk_id();
    const std::string unk_name = IdToPiece(unk);
    const bool to_unk = spmx_decode_unk_option(h_) != 0;
    std::vector<int32_t> ids;
    std::string lit;
    std::vector<uint64_t> lo{0};
    ids.reserve(pieces.size());
    for (std::string_view p : pieces) {
      int32_t t = spmx_piece_to_id(h_, p.data(), p.size());
      if (t == unk && p != unk_name && !to_unk) {
        lit.append(p.data(), p.size());
        lo.push_back(lit.size());
        t = -static_cast<int32_t>(lo.size() - 1);
      }
      ids.push_back(t);
    }
    char *text = nullptr;
    uint64_t *offs = nullptr;
    const uint64_t io[2] = {0, ids.size()};
    const int32_t none = 0;
    const int rc = spmx_decode_batch_pieces(h_, ids.empty() ? &none : ids.data(), io, 1, lit.data(), lo.data(), lo.size() - 1, &text, &offs);
    if (rc != 0) return FromHandle(rc);
    detokenized->assign(text, text + offs[1]);
    spmx_free(text);
    spmx_free(offs);
    return util::Status();
  }
  virtual util::Status Decode(const std::vector<std::string> &pieces, std::string *detokenized) const {
    std::vector<std::string_view> v(pieces.begin(), pieces.end());
    return Decode(v, detokenized);
  }
  std::string DecodePieces(const std::vector<std::string> &pieces) const {   // sentencepiece_processor.h:510-513; errors are swallowed
    std::string out;
    (void)Decode(pieces, &out);
    return out;
  }
  // EncodeAsSerializedProto (sentencepiece_processor.h:528-531): the serialized SentencePieceText of Encode(input, &spt)
  virtual std::string EncodeAsSerializedProto(std::string_view input) const {
    SentencePieceText spt;
    if (!Encode(input, &spt).ok()) return std::string();
    return spt.SerializeAsString();
  }
  // Flat form: CSR ids in, packed text + offsets out.
  util::Status DecodeBatchFlat(const int32_t *ids, const uint64_t *id_offsets, uint64_t n, std::string *text,
                               std::vector<uint64_t> *text_offsets) const {
    if (!h_) return status();
    if (!text || !text_offsets) return util::Status(util::StatusCode::kInternal, "output container is null");
    text->clear();
    text_offsets->clear();
    char *t = nullptr;
    uint64_t *offs = nullptr;
    const int rc = spmx_decode_batch(h_, ids, id_offsets, n, &t, &offs);
    if (rc != 0) return FromHandle(rc);
    text_offsets->assign(offs, offs + n + 1);
    text->assign(t, t + offs[n]);
    spmx_free(t);
    spmx_free(offs);
    return util::Status();
  }

  // ---- vocabulary ----
  virtual int GetPieceSize() const { return h_ ? spmx_piece_size(h_) : 0; }
  virtual int PieceToId(std::string_view piece) const { return h_ ? spmx_piece_to_id(h_, piece.data(), piece.size()) : 0; }
  virtual std::string IdToPiece(int id) const {
    static const std::string kEmpty;
    if (!h_) return kEmpty;
    const int64_t n = spmx_id_to_piece(h_, id, nullptr, 0);
    if (n < 0) return kEmpty;
    std::string s(static_cast<size_t>(n), '\0');
    spmx_id_to_piece(h_, id, s.data(), s.size());
    return s;
  }
  // GetScore (sentencepiece_processor.h:650): the score the ModelProto holds; 0 for an id out of range, as the reference's
  // CHECK_STATUS_OR_RETURN_DEFAULT does for a processor without a model
  virtual float GetScore(int id) const {
    float s = 0.f;
    if (!h_ || spmx_piece_score(h_, id, &s) != 0) return 0.f;
    return s;
  }
  // serialized_model_proto (sentencepiece_processor.h:694): the bytes the model was loaded from
  std::string serialized_model_proto() const {
    const char *p = nullptr;
    uint64_t n = 0;
    if (!h_ || spmx_serialized_model(h_, &p, &n) != 0) return std::string();
    return std::string(p, n);
  }
  virtual bool IsUnknown(int id) const { return h_ && spmx_piece_type(h_, id) == 2; }   // sentencepiece_processor.h:653-662
  virtual bool IsControl(int id) const { return h_ && spmx_piece_type(h_, id) == 3; }
  virtual bool IsUnused(int id) const { return h_ && spmx_piece_type(h_, id) == 5; }
  virtual bool IsByte(int id) const { return h_ && spmx_piece_type(h_, id) == 6; }
  // trainer_spec.unk_piece: what the `unk_piece` extra option writes for unknown tokens
  std::string UnkPiece() const {
    if (!h_) return std::string();
    const int64_t n = spmx_unk_piece(h_, nullptr, 0);
    std::string s(static_cast<size_t>(n > 0 ? n : 0), '\0');
    if (n > 0) spmx_unk_piece(h_, s.data(), s.size());
    return s;
  }
  virtual int unk_id() const { return h_ ? spmx_unk_id(h_) : 0; }
  virtual int bos_id() const { return h_ ? spmx_bos_id(h_) : 0; }
  virtual int eos_id() const { return h_ ? spmx_eos_id(h_) : 0; }
  virtual int pad_id() const { return h_ ? spmx_pad_id(h_) : 0; }

  spmx_handle *handle() const { return h_; }

 private:
  void Reset() {
    spmx_destroy(h_);
    h_ = nullptr;
  }
  util::Status Created(int rc) {
    if (rc == 0) return util::Status();
    h_ = nullptr;
    return util::Status(static_cast<util::StatusCode>(rc), spmx_last_error(nullptr));
  }
  util::Status FromHandle(int rc) const {
    if (rc == 0) return util::Status();
    return util::Status(static_cast<util::StatusCode>(rc), spmx_last_error(h_));
  }
  // pieces [lo, hi) of a spans-form result -> spt (PopulateSentencePieceText, sentencepiece_processor.cc:547-636): a piece
  // is its normalized text, the piece name for a byte-fallback piece and a bos / eos, unk_piece for an unknown token
  // under the `unk_piece` extra option (:1050-1058)
  void FillPieces(std::string_view input, const char *norm, const int32_t *ids, const uint32_t *b, const uint32_t *e,
                  const uint32_t *nb, const uint32_t *ne, uint64_t lo, uint64_t hi, SentencePieceText *spt) const {
    spt->text.assign(input.data() ? input.data() : "", input.size());
    for (uint64_t k = lo; k < hi; ++k) {
      SentencePieceText::SentencePiece p;
      p.id = static_cast<uint32_t>(ids[k]);
      p.begin = b[k];
      p.end = e[k];
      p.surface.assign(input.data() + b[k], e[k] - b[k]);
      const int type = spmx_piece_type(h_, ids[k]);
      if (type == 2 && unk_piece_option_) p.piece = UnkPiece();
      else if (type == 6 || type == 3) p.piece = IdToPiece(ids[k]);
      else p.piece.assign(norm + nb[k], ne[k] - nb[k]);
      // presence of `surface`: not on a bos / eos; of the byte pieces of one unknown character (consecutive, same
      // normalized begin) only on the one that is last in TEXT order
      if (type == 3) p.has_surface = false;
      else if (type == 6) {
        const bool has_next = reverse_option_ ? k > lo : k + 1 < hi;
        const uint64_t nx = reverse_option_ ? k - 1 : k + 1;
        if (has_next && spmx_piece_type(h_, ids[nx]) == 6 && nb[nx] == nb[k]) p.has_surface = false;
      }
      spt->pieces.push_back(std::move(p));
    }
  }
  int device_ = 0;
  bool unk_piece_option_ = false;   // the `unk` / `unk_piece` extra option: piece strings only (:1050-1058)
  bool reverse_option_ = false;     // an odd number of `reverse` options: the pieces come out last first
  spmx_handle *h_ = nullptr;
};

}  // namespace sentencepiece_amd
#endif
