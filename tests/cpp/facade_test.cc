// Exercises include/spmx_processor.h the way a C++ caller of the reference
// would (cf. src/spm_encode_main.cc:115-119): Load, SetEncodeExtraOptions,
// Encode per line and EncodeBatch over all lines, ids printed one line per
// sentence.  `--expect-unavailable` checks the no-GPU error path instead.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/spmx_processor.h"

namespace sentencepiece = sentencepiece_amd;   // the one-line switch INTEGRATION.md describes

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: facade_test MODEL TEXTFILE|--expect-unavailable [extra_options]\n"); return 2; }
  sentencepiece::SentencePieceProcessor sp;
  if (sp.status().ok()) { fprintf(stderr, "status() must fail before Load\n"); return 1; }
  const sentencepiece::util::Status st = sp.Load(argv[1]);
  if (std::string(argv[2]) == "--expect-unavailable") {
    if (st.ok() || st.code() != sentencepiece::util::StatusCode::kUnavailable) { fprintf(stderr, "expected kUnavailable, got %s\n", st.ToString().c_str()); return 1; }
    std::vector<int> ids{1, 2, 3};
    if (sp.Encode("abc", &ids).ok() || !sp.EncodeAsIds("abc").empty()) { fprintf(stderr, "Encode must fail without a model\n"); return 1; }
    if (!sp.Load("/nonexistent/x.model").ok()) { printf("unavailable ok: %s\n", st.ToString().c_str()); return 0; }
    return 1;
  }
  if (!st.ok()) { fprintf(stderr, "%s\n", st.ToString().c_str()); return 1; }
  if (argc > 3 && !sp.SetEncodeExtraOptions(argv[3]).ok()) { fprintf(stderr, "bad options\n"); return 1; }
  if (sp.SetEncodeExtraOptions("nonsense").ok()) { fprintf(stderr, "unknown option accepted\n"); return 1; }
  if (argc > 3) sp.SetEncodeExtraOptions(argv[3]);
  std::ifstream f(argv[2], std::ios::binary);
  std::vector<std::string> lines;
  for (std::string line; std::getline(f, line);) lines.push_back(line);
  std::vector<std::string_view> views(lines.begin(), lines.end());
  std::vector<std::vector<int>> batch;
  if (!sp.EncodeBatch(views, &batch).ok() || batch.size() != lines.size()) { fprintf(stderr, "EncodeBatch failed\n"); return 1; }
  for (size_t i = 0; i < lines.size(); ++i) {
    if (i < 40) {   // batch == per-sentence Encode (python/test/sentencepiece_test.py:745-760)
      std::vector<int> one;
      if (!sp.Encode(lines[i], &one).ok() || one != batch[i]) { fprintf(stderr, "Encode != EncodeBatch at line %zu\n", i); return 1; }
    }
    std::ostringstream os;
    for (size_t k = 0; k < batch[i].size(); ++k) os << (k ? " " : "") << batch[i][k];
    std::cout << os.str() << "\n";
  }
  if (sp.GetPieceSize() <= 0 || sp.IdToPiece(sp.unk_id()).empty() || sp.PieceToId(sp.IdToPiece(5)) != 5) { fprintf(stderr, "vocab accessors\n"); return 1; }
  if (sp.Encode("x", nullptr).ok()) { fprintf(stderr, "null output accepted\n"); return 1; }
  return 0;
}
