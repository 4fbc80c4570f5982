// Host-side model: a minimal proto2 wire reader for the ModelProto fields the
// encode path reads (reference: src/sentencepiece_model.proto:293-332 pieces,
// :54/:151/:194/:220-223 trainer_spec, :245-274 normalizer_spec, :277-283
// self_test_data) and the load-time bookkeeping of
// ModelInterface::InitializePieces (src/model_interface.cc:63-151).
#ifndef SPMX_MODEL_H_
#define SPMX_MODEL_H_
#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace spmx {

// util::StatusCode values (reference: src/sentencepiece_processor.h:34-52).
enum StatusCode {
  kOk = 0, kCancelled = 1, kUnknown = 2, kInvalidArgument = 3, kDeadlineExceeded = 4,
  kNotFound = 5, kAlreadyExists = 6, kPermissionDenied = 7, kResourceExhausted = 8,
  kFailedPrecondition = 9, kAborted = 10, kOutOfRange = 11, kUnimplemented = 12,
  kInternal = 13, kUnavailable = 14, kDataLoss = 15, kUnauthenticated = 16
};

struct Status {
  int code = kOk;
  std::string message;
  bool ok() const { return code == kOk; }
  static Status OK() { return Status(); }
  static Status Error(int c, std::string m) { Status s; s.code = c; s.message = std::move(m); return s; }
};

enum PieceType { kNormal = 1, kUnknown_ = 2, kControl = 3, kUserDefined = 4, kUnused = 5, kByte = 6 };
enum ModelType { kUnigram = 1, kBpe = 2, kWord = 3, kChar = 4 };

struct PieceRec {
  std::string piece;
  float score = 0.f;
  int load_type = kNormal;  // type when the model was loaded (structure is built from this)
  int type = kNormal;       // current type (SetVocabulary / ResetVocabulary mutate it)
};

struct ModelData {
  std::vector<PieceRec> pieces;
  int model_type = kUnigram;
  bool byte_fallback = false;
  bool ws_suffix = false;  // treat_whitespace_as_suffix
  std::string unk_piece = "<unk>", bos_piece = "<s>", eos_piece = "</s>", pad_piece = "<pad>";
  std::string unk_surface = " \xE2\x81\x87 ";   // trainer_spec.unk_surface (sentencepiece_model.proto:228), for Decode
  bool has_denormalizer = false;              // denormalizer_spec with a charsmap (sentencepiece_processor.cc:248-252)
  std::string dn_charsmap;                    // its precompiled_charsmap and flags: a Normalizer(spec) of its own, run over
  bool dn_add_dummy_prefix = true, dn_remove_extra_ws = true, dn_escape_ws = true;   // the decoded text (:905-907)
  bool normalizer_only = false;               // CompileTables: stop after the normalizer tables (the denormalizer's)
  std::string charsmap;    // normalizer_spec.precompiled_charsmap
  bool add_dummy_prefix = true, remove_extra_ws = true, escape_ws = true;
  std::vector<std::pair<std::string, std::string>> self_test;

  // InitializePieces state
  std::unordered_map<std::string, int> pieces_map;    // NORMAL / USER_DEFINED / UNUSED
  std::unordered_map<std::string, int> reserved_map;  // UNKNOWN / CONTROL / BYTE
  int unk_id = -1;
  int byte_ids[256];
  // unigram::Model::Model (src/unigram_model.cc:652-670)
  float min_score = 0.f, max_score = 0.f;

  // ModelInterface::PieceToId (src/model_interface.cc:51-61).
  int PieceToId(const std::string &piece) const;
};

Status ParseModelProto(const void *data, size_t n, ModelData *out);
// InitializePieces + unigram min/max.  Error text follows the reference.
Status InitializeModel(ModelData *m);
// SetVocabulary / ResetVocabulary (src/sentencepiece_processor.cc:301-340).
Status SetVocabulary(ModelData *m, const std::vector<std::string> &valid);
Status ResetVocabulary(ModelData *m);

inline int OneCharLen(unsigned char c) {  // src/util.h:151-153
  return "\1\1\1\1\1\1\1\1\1\1\1\1\2\2\3\4"[c >> 4];
}

}  // namespace spmx
#endif
