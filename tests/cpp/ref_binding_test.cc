// TEST: include/spmx_reference_binding.h compiled against the reference's own header and objects, driven through a
// sentencepiece::SentencePieceProcessor* exactly as spm_encode drives the reference (src/spm_encode_main.cc:115-119:
// per line `sp.Encode(line, &ids)`), next to the unmodified base class on the same lines.
//   ref_binding_test <model> <text file> [extra options]
// exit 0 and "OK <lines> <ids>" when every line's ids (and Decode of them) agree.
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "spmx_reference_binding.h"

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s model text [options]\n", argv[0]); return 2; }
  sentencepiece::SentencePieceProcessor base;
  std::unique_ptr<sentencepiece::SentencePieceProcessor> amd(new sentencepiece::AmdSentencePieceProcessor(0));
  sentencepiece::SentencePieceProcessor *sp = amd.get();          // everything below goes through the base-class pointer
  auto st = base.Load(argv[1]);
  if (!st.ok()) { fprintf(stderr, "base.Load: %s\n", st.ToString().c_str()); return 1; }
  st = sp->Load(argv[1]);
  if (!st.ok()) { fprintf(stderr, "amd.Load: %s\n", st.ToString().c_str()); return 1; }
  if (argc > 3) {
    st = base.SetEncodeExtraOptions(argv[3]);
    if (st.ok()) st = sp->SetEncodeExtraOptions(argv[3]);
    if (!st.ok()) { fprintf(stderr, "SetEncodeExtraOptions: %s\n", st.ToString().c_str()); return 1; }
  }
  std::ifstream in(argv[2]);
  std::string line, packed;
  std::vector<uint64_t> offs{0};
  std::vector<std::vector<int>> want;
  size_t lines = 0, total = 0;
  while (std::getline(in, line)) {
    std::vector<int> a, b;
    const auto s1 = base.Encode(line, &a), s2 = sp->Encode(line, &b);
    if (s1.ok() != s2.ok() || a != b) { fprintf(stderr, "line %zu: ids differ (%zu vs %zu)\n", lines, a.size(), b.size()); return 1; }
    std::string t1, t2;
    if (!base.Decode(a, &t1).ok() || !sp->Decode(b, &t2).ok() || t1 != t2) { fprintf(stderr, "line %zu: Decode differs\n", lines); return 1; }
    packed += line;
    offs.push_back(packed.size());
    want.push_back(std::move(a));
    total += b.size();
    ++lines;
  }
  // the batch entry point of the subclass: the same ids, sentence by sentence
  int32_t *ids = nullptr;
  uint64_t *io = nullptr;
  st = static_cast<sentencepiece::AmdSentencePieceProcessor *>(sp)->EncodeBatch(packed.data(), offs.data(), lines, &ids, &io);
  if (!st.ok()) { fprintf(stderr, "EncodeBatch: %s\n", st.ToString().c_str()); return 1; }
  for (size_t i = 0; i < lines; ++i) {
    if (io[i + 1] - io[i] != want[i].size()) { fprintf(stderr, "EncodeBatch: line %zu has %llu ids, not %zu\n", i, (unsigned long long)(io[i + 1] - io[i]), want[i].size()); return 1; }
    for (size_t k = 0; k < want[i].size(); ++k)
      if (ids[io[i] + k] != want[i][k]) { fprintf(stderr, "EncodeBatch: line %zu id %zu differs\n", i, k); return 1; }
  }
  spmx_free(ids);
  spmx_free(io);
  // vocabulary restriction goes to both sides (:279-283)
  std::vector<absl::string_view> vocab;
  const std::vector<std::string> keep = {"\xE2\x96\x81the", "\xE2\x96\x81a", "s", "e", "t"};
  for (const auto &k : keep) vocab.push_back(k);
  if (base.SetVocabulary(vocab).ok() && sp->SetVocabulary(vocab).ok()) {
    std::vector<int> a, b;
    (void)base.Encode("the theatre of the absurd", &a);
    (void)sp->Encode("the theatre of the absurd", &b);
    if (a != b) { fprintf(stderr, "SetVocabulary: ids differ\n"); return 1; }
    (void)base.ResetVocabulary();
    (void)sp->ResetVocabulary();
  }
  printf("OK %zu %zu\n", lines, total);
  return 0;
}
