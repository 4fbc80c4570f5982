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
  if (argc < 3) { fprintf(stderr, "usage: facade_test MODEL TEXTFILE|--expect-unavailable [extra_options [--pieces|--status]]\n"); return 2; }
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
  if (argc > 4 && std::string(argv[4]) == "--status") {
    // one line per sentence: the Status of Encode(input, &ids), of Encode(input, &spt), of Encode(input, &pieces) as
    // "code|message", then how many ids the error-swallowing EncodeAsIds returns (sentencepiece_processor.cc:392-403, :628)
    for (const std::string &line : lines) {
      std::vector<int> ids{7, 7, 7};
      const sentencepiece::util::Status a = sp.Encode(line, &ids);
      if (!a.ok() && !ids.empty()) { fprintf(stderr, "ids of a failing sentence\n"); return 1; }
      sentencepiece::SentencePieceText spt;
      const sentencepiece::util::Status b = sp.Encode(line, &spt);
      std::vector<std::string> pcs{"x"};
      const sentencepiece::util::Status c = sp.Encode(line, &pcs);
      if (!c.ok() && !pcs.empty()) { fprintf(stderr, "pieces of a failing sentence\n"); return 1; }
      std::cout << static_cast<int>(a.code()) << "|" << a.error_message() << "\t" << static_cast<int>(b.code()) << "|" << b.error_message()
                << "\t" << static_cast<int>(c.code()) << "|" << c.error_message() << "\t" << sp.EncodeAsIds(line).size() << "\t" << ids.size() << "\n";
    }
    return 0;
  }
  if (argc > 4 && std::string(argv[4]) == "--extras") {
    // The facade's newer methods, driven through a BASE-CLASS POINTER to a subclass that overrides nothing (they are virtual,
    // as the reference's are).  One line each:
    //   D <hex of Decode(pieces) of the piece list of line i, damaged by pieces outside the vocabulary>   (per input line)
    //   P <hex of EncodeAsSerializedProto(line)>                                                          (per input line)
    //   S <GetScore(id)> for ids 0 .. 63 as %.9g
    //   M <size of serialized_model_proto()> <its FNV-1a>
    //   V <ids of line 0 after LoadVocabulary(argv[5], 2)> | <ids after ResetVocabulary>
    struct Sub : sentencepiece::SentencePieceProcessor {};
    auto hex = [](const std::string &x) { static const char *d = "0123456789abcdef"; std::string o; for (unsigned char c : x) { o += d[c >> 4]; o += d[c & 15]; } return o.empty() ? std::string("-") : o; };
    Sub sub;
    sentencepiece::SentencePieceProcessor *p = &sub;
    if (!p->Load(argv[1]).ok() || !p->SetEncodeExtraOptions(argv[3]).ok()) { fprintf(stderr, "Load through the base pointer\n"); return 1; }
    const char *decode_opts = argc > 6 ? argv[6] : "";
    if (!p->SetDecodeExtraOptions(decode_opts).ok()) { fprintf(stderr, "SetDecodeExtraOptions\n"); return 1; }
    size_t li = 0;
    for (const std::string &line : lines) {
      std::vector<std::string> pcs = p->EncodeAsPieces(line);
      if (li % 3 == 0) pcs.insert(pcs.begin() + static_cast<long>(pcs.size() / 2), "zzqq-not-a-piece");
      if (li % 3 == 1) { pcs.push_back("\xE2\x96\x81outside"); pcs.insert(pcs.begin(), ""); }
      if (li % 5 == 2) pcs.push_back(p->IdToPiece(p->unk_id()));
      std::string text;
      if (!p->Decode(pcs, &text).ok() || sub.DecodePieces(pcs) != text) { fprintf(stderr, "Decode(pieces) at line %zu\n", li); return 1; }
      std::cout << "D " << hex(text) << "\n";
      std::cout << "P " << hex(p->EncodeAsSerializedProto(line)) << "\n";
      ++li;
    }
    std::cout << "S";
    for (int id = 0; id < 64 && id < p->GetPieceSize(); ++id) { char b[32]; snprintf(b, sizeof(b), " %.9g", static_cast<double>(p->GetScore(id))); std::cout << b; }
    std::cout << "\n";
    const std::string blob = sub.serialized_model_proto();
    uint64_t fnv = 1469598103934665603ull;
    for (unsigned char c : blob) { fnv ^= c; fnv *= 1099511628211ull; }
    std::cout << "M " << blob.size() << " " << fnv << "\n";
    if (argc > 5 && argv[5][0]) {
      if (!p->LoadVocabulary(argv[5], 2).ok()) { fprintf(stderr, "LoadVocabulary\n"); return 1; }
      std::cout << "V";
      for (int t : p->EncodeAsIds(lines[0])) std::cout << " " << t;
      if (!p->ResetVocabulary().ok()) { fprintf(stderr, "ResetVocabulary\n"); return 1; }
      std::cout << " |";
      for (int t : p->EncodeAsIds(lines[0])) std::cout << " " << t;
      std::cout << "\n";
      if (p->LoadVocabulary("/nonexistent/vocab.tsv", 1).ok()) { fprintf(stderr, "LoadVocabulary of a missing file\n"); return 1; }
    }
    return 0;
  }
  if (argc > 4 && std::string(argv[4]) == "--pieces") {
    // one line per sentence: hex(piece):id:begin:end ..., then "N " + hex(normalized) + the norm_to_orig entries
    auto hex = [](const std::string &x) { static const char *d = "0123456789abcdef"; std::string o; for (unsigned char c : x) { o += d[c >> 4]; o += d[c & 15]; } return o.empty() ? std::string("-") : o; };
    for (const std::string &line : lines) {
      sentencepiece::SentencePieceText spt;
      if (!sp.Encode(line, &spt).ok() || spt.text != line) { fprintf(stderr, "Encode(spt) failed\n"); return 1; }
      std::ostringstream os;
      for (const auto &p : spt.pieces) {
        if (p.surface != line.substr(p.begin, p.end - p.begin)) { fprintf(stderr, "surface\n"); return 1; }
        os << hex(p.piece) << ":" << p.id << ":" << p.begin << ":" << p.end << " ";
      }
      std::vector<std::string> pcs = sp.EncodeAsPieces(line);
      if (pcs.size() != spt.pieces.size()) { fprintf(stderr, "EncodeAsPieces\n"); return 1; }
      std::string norm;
      std::vector<size_t> n2o;
      if (!sp.Normalize(line, &norm, &n2o).ok() || sp.Normalize(line) != norm) { fprintf(stderr, "Normalize failed\n"); return 1; }
      os << "N " << hex(norm);
      for (size_t v : n2o) os << " " << v;
      std::cout << os.str() << "\n";
    }
    return 0;
  }
  std::vector<std::string_view> views(lines.begin(), lines.end());
  std::vector<std::vector<int>> batch;
  if (!sp.EncodeBatch(views, &batch).ok() || batch.size() != lines.size()) { fprintf(stderr, "EncodeBatch failed\n"); return 1; }
  {   // the zero-copy forms hold the same ids
    sentencepiece::EncodedBatch eb;
    if (!sp.EncodeBatch(views, &eb).ok() || eb.size() != lines.size()) { fprintf(stderr, "EncodeBatch(EncodedBatch) failed\n"); return 1; }
    for (size_t i = 0; i < lines.size(); ++i)
      if (std::vector<int>(eb.begin(i), eb.end(i)) != batch[i]) { fprintf(stderr, "EncodedBatch differs at line %zu\n", i); return 1; }
    std::string packed;
    std::vector<uint64_t> po(1, 0);
    for (const std::string &l : lines) { packed += l; po.push_back(packed.size()); }
    sentencepiece::EncodedBatch fb;
    if (!sp.EncodeBatchFlat(packed.data(), po.data(), lines.size(), &fb).ok() || fb.total_ids() != eb.total_ids()) { fprintf(stderr, "EncodeBatchFlat(EncodedBatch) failed\n"); return 1; }
    sentencepiece::EncodedBatch moved(std::move(fb));
    if (moved.size() != lines.size() || fb.size() != 0) { fprintf(stderr, "EncodedBatch move\n"); return 1; }
  }
  for (size_t i = 0; i < lines.size(); ++i) {
    if (i < 40) {   // batch == per-sentence Encode (python/test/sentencepiece_test.py:745-760)
      std::vector<int> one;
      if (!sp.Encode(lines[i], &one).ok() || one != batch[i]) { fprintf(stderr, "Encode != EncodeBatch at line %zu\n", i); return 1; }
    }
    std::ostringstream os;
    for (size_t k = 0; k < batch[i].size(); ++k) os << (k ? " " : "") << batch[i][k];
    std::cout << os.str() << "\n";
  }
  {   // sampling: nbest_size 0 / 1 is Encode (unigram; BPE: dropout at alpha = 0); a sampled segmentation decodes to the same text; the kOriginal encoder
    const std::string &probe = lines.size() > 7 ? lines[7] : lines[0];
    std::vector<int> plain, same, drawn, orig;
    if (!sp.Encode(probe, &plain).ok() || !sp.SampleEncode(probe, 1, 0.0f, &same).ok() || same != plain) { fprintf(stderr, "SampleEncode(nbest 1)\n"); return 1; }
    if (!sp.SampleEncode(probe, -1, 0.2f, 99, &drawn).ok()) { fprintf(stderr, "SampleEncode\n"); return 1; }
    std::string t1, t2;
    if (!sp.Decode(plain, &t1).ok() || !sp.Decode(drawn, &t2).ok() || t1 != t2) { fprintf(stderr, "sampled ids decode differently\n"); return 1; }
    std::vector<int> again;
    if (!sp.SampleEncode(probe, -1, 0.2f, 99, &again).ok() || again != drawn) { fprintf(stderr, "seeded draw is not reproducible\n"); return 1; }
    const sentencepiece::util::Status so = sp.EncodeOriginal(probe, &orig);   // unigram only (INTERNAL for BPE, like NBestEncode)
    if (so.ok()) { std::string t3; if (!sp.Decode(orig, &t3).ok() || t3 != t1) { fprintf(stderr, "EncodeOriginal\n"); return 1; } }
  }
  {   // piece / proto forms of NBestEncode and SampleEncode (sentencepiece_processor.h:318-324, :346-348, :404-408)
    const std::string &probe = lines.size() > 7 ? lines[7] : lines[0];
    std::vector<std::vector<int>> nids;
    sentencepiece::NBestSentencePieceText nb;
    std::vector<std::vector<std::string>> npc;
    const sentencepiece::util::Status a = sp.NBestEncode(probe, 3, &nids), b2 = sp.NBestEncode(probe, 3, &nb), c2 = sp.NBestEncode(probe, 3, &npc);
    if (a.ok() != b2.ok() || a.ok() != c2.ok()) { fprintf(stderr, "NBestEncode forms disagree on the status\n"); return 1; }
    if (a.ok()) {
      if (nb.nbests.size() != nids.size() || npc.size() != nids.size() || sp.NBestEncodeAsPieces(probe, 3) != npc) { fprintf(stderr, "NBestEncode forms disagree on the count\n"); return 1; }
      for (size_t r = 0; r < nids.size(); ++r) {
        const auto &spt = nb.nbests[r];
        if (spt.text != probe || spt.pieces.size() != nids[r].size() || npc[r].size() != nids[r].size()) { fprintf(stderr, "NBestEncode(spt) shape\n"); return 1; }
        if (r > 0 && spt.score > nb.nbests[r - 1].score) { fprintf(stderr, "NBestEncode(spt) scores are not descending\n"); return 1; }
        for (size_t k = 0; k < spt.pieces.size(); ++k) {
          const auto &p = spt.pieces[k];
          if (static_cast<int>(p.id) != nids[r][k] || p.piece != npc[r][k] || p.begin > p.end || p.end > probe.size() ||
              p.surface != probe.substr(p.begin, p.end - p.begin)) { fprintf(stderr, "NBestEncode(spt) piece %zu of result %zu\n", k, r); return 1; }
        }
      }
    }
    sentencepiece::SentencePieceText plain, s1, drawn;
    std::vector<std::string> dp;
    if (!sp.Encode(probe, &plain).ok() || !sp.SampleEncode(probe, 1, 0.0f, &s1).ok() || s1.pieces.size() != plain.pieces.size()) { fprintf(stderr, "SampleEncode(spt, nbest 1)\n"); return 1; }
    for (size_t k = 0; k < plain.pieces.size(); ++k)
      if (s1.pieces[k].piece != plain.pieces[k].piece || s1.pieces[k].id != plain.pieces[k].id || s1.pieces[k].begin != plain.pieces[k].begin ||
          s1.pieces[k].end != plain.pieces[k].end) { fprintf(stderr, "SampleEncode(spt, nbest 1) piece %zu\n", k); return 1; }
    std::vector<int> di;
    if (!sp.SampleEncode(probe, -1, 0.2f, 99, &drawn).ok() || !sp.SampleEncode(probe, -1, 0.2f, 99, &di).ok() || drawn.pieces.size() != di.size()) { fprintf(stderr, "SampleEncode(spt)\n"); return 1; }
    for (size_t k = 0; k < di.size(); ++k) if (static_cast<int>(drawn.pieces[k].id) != di[k]) { fprintf(stderr, "SampleEncode(spt) ids\n"); return 1; }
    if (!sp.SampleEncode(probe, -1, 0.2f, &dp).ok() || dp.empty() != probe.empty()) { fprintf(stderr, "SampleEncode(pieces)\n"); return 1; }
  }
  if (sp.GetPieceSize() <= 0 || sp.IdToPiece(sp.unk_id()).empty() || sp.PieceToId(sp.IdToPiece(5)) != 5) { fprintf(stderr, "vocab accessors\n"); return 1; }
  if (sp.Encode("x", static_cast<std::vector<int> *>(nullptr)).ok()) { fprintf(stderr, "null output accepted\n"); return 1; }
  return 0;
}
