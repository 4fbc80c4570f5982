// The reference-side binding of the encode path: what a maintainer of google/sentencepiece would add to the reference
// tree -- a subclass of sentencepiece::SentencePieceProcessor that forwards the virtuals the
// text -> ids path goes through (src/sentencepiece_processor.h:245-312) to the C ABI of libspmx (include/spmx.h), plus
// the batch entry point the Python wrapper's _EncodeAsIdsBatch (python/src/sentencepiece/sentencepiece.i:439-446) would
// call instead of its thread pool.  Compiled only INSIDE the reference tree (it includes the reference's own header);
// tests/cpp/ref_binding_test.cc builds it against /root/reference where that exists and drives it through a
// base-class pointer with spm_encode's loop (src/spm_encode_main.cc:115-119).
#ifndef SPMX_REFERENCE_BINDING_H_
#define SPMX_REFERENCE_BINDING_H_
#include <string>
#include <vector>

#include "sentencepiece_processor.h"   // the reference's: -I<reference>/src
#include "spmx.h"

namespace sentencepiece {

class AmdSentencePieceProcessor : public SentencePieceProcessor {
 public:
  explicit AmdSentencePieceProcessor(int device = 0) : device_(device) {}
  ~AmdSentencePieceProcessor() override { spmx_destroy(h_); }

  // Load(filename) (:245) reads the file and ends here too (sentencepiece_processor.cc:201-206 -> :242)
  util::Status LoadFromSerializedProto(absl::string_view serialized) override {                     // :261
    const util::Status st = SentencePieceProcessor::LoadFromSerializedProto(serialized);            // (keeps every other method working)
    if (!st.ok()) return st;
    spmx_destroy(h_);
    h_ = nullptr;
    return ToStatus(spmx_create(serialized.data(), serialized.size(), device_, &h_), nullptr);
  }
  util::Status Load(absl::string_view filename) override {                                           // :245
    const util::Status st = SentencePieceProcessor::Load(filename);
    if (!st.ok()) return st;
    const std::string blob = serialized_model_proto();                                               // :694
    spmx_destroy(h_);
    h_ = nullptr;
    return ToStatus(spmx_create(blob.data(), blob.size(), device_, &h_), nullptr);
  }
  util::Status SetEncodeExtraOptions(absl::string_view o) override {                                 // :267
    const util::Status st = SentencePieceProcessor::SetEncodeExtraOptions(o);
    if (!st.ok()) return st;
    return ToStatus(spmx_set_encode_extra_options(h_, std::string(o.data(), o.size()).c_str()), h_);
  }
  util::Status SetDecodeExtraOptions(absl::string_view o) override {                                 // :270
    const util::Status st = SentencePieceProcessor::SetDecodeExtraOptions(o);
    if (!st.ok()) return st;
    return ToStatus(spmx_set_decode_extra_options(h_, std::string(o.data(), o.size()).c_str()), h_);
  }
  util::Status SetVocabulary(const std::vector<absl::string_view> &valid_vocab) override {           // :279
    const util::Status st = SentencePieceProcessor::SetVocabulary(valid_vocab);
    if (!st.ok()) return st;
    std::vector<const char *> p;
    std::vector<uint64_t> l;
    for (const auto &v : valid_vocab) { p.push_back(v.data()); l.push_back(v.size()); }
    return ToStatus(spmx_set_vocabulary(h_, p.data(), l.data(), p.size()), h_);
  }
  util::Status ResetVocabulary() override {                                                          // :283
    const util::Status st = SentencePieceProcessor::ResetVocabulary();
    if (!st.ok()) return st;
    return ToStatus(spmx_reset_vocabulary(h_), h_);
  }
  // the path itself: Encode(input, vector<int>*) (:299-300, .cc:392-403) on the device
  util::Status Encode(absl::string_view input, std::vector<int> *ids) const override {
    const util::Status st = status();                      // CHECK_OR_RETURN_STATUS_STL (.cc:364-370)
    if (!st.ok()) return st;
    if (!ids) return util::Status(util::StatusCode::kInternal, "output container is null");
    ids->clear();
    uint64_t n = 0;
    ids->resize(input.size() + 8);
    int rc = spmx_encode(h_, input.data(), input.size(), ids->data(), ids->size(), &n);
    if (rc == 8 /* RESOURCE_EXHAUSTED: n says what it takes */) {
      ids->resize(n);
      rc = spmx_encode(h_, input.data(), input.size(), ids->data(), ids->size(), &n);
    }
    ids->resize(rc == 0 ? n : 0);
    return ToStatus(rc, h_);
  }
  util::Status Decode(const std::vector<int> &ids, std::string *detokenized) const override {       // :311-312
    const util::Status st = status();
    if (!st.ok()) return st;
    if (!detokenized) return util::Status(util::StatusCode::kInternal, "output container is null");
    uint64_t n = 0;
    detokenized->resize(ids.size() * 8 + 16);
    int rc = spmx_decode(h_, ids.data(), ids.size(), &(*detokenized)[0], detokenized->size(), &n);
    if (rc == 8) {
      detokenized->resize(n);
      rc = spmx_decode(h_, ids.data(), ids.size(), &(*detokenized)[0], detokenized->size(), &n);
    }
    detokenized->resize(rc == 0 ? n : 0);
    return ToStatus(rc, h_);
  }
  // NEW: the batch form, element-wise Encode (sentencepiece.i:245-267): packed sentences in, CSR out (spmx_free)
  util::Status EncodeBatch(const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids, uint64_t **id_offsets) const {
    return ToStatus(spmx_encode_batch(h_, text, offsets, n, ids, id_offsets), h_);
  }

 private:
  static util::Status ToStatus(int rc, const spmx_handle *h) {
    if (rc == 0) return util::Status();
    return util::Status(static_cast<util::StatusCode>(rc), spmx_last_error(h));
  }
  int device_ = 0;
  spmx_handle *h_ = nullptr;
};

}  // namespace sentencepiece
#endif
