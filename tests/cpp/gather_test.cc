// The multi-GPU form a C++ host of the reference would write, with ranks = THREADS of this
// process: every rank loads the model, encodes its contiguous shard of the file's lines with EncodeBatchDevice and calls
// SentencePieceProcessor::AllGatherIds; every rank must end up with the CSR a single processor gives for the whole file.
// TEST ONLY: linked against tests/emu/libspmx_emu.so (the product's api.cc and kernels over the CPU model of the
// wavefront -- "device" memory is host memory) with SPMX_RCCL_LIB pointing at tests/emu/libfake_rccl.so.
//   usage: gather_test MODEL TEXTFILE WORLD          prints "ok <sentences> <ids>" or a message and exits 1
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spmx_processor.h"

namespace sentencepiece = sentencepiece_amd;

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: gather_test MODEL TEXTFILE WORLD\n"); return 2; }
  const int world = atoi(argv[3]);
  std::ifstream f(argv[2], std::ios::binary);
  std::string text;
  std::vector<uint64_t> offs{0};
  for (std::string line; std::getline(f, line) && offs.size() <= 600;) { text += line; offs.push_back(text.size()); }
  const uint64_t n = offs.size() - 1;
  // the whole file on one processor: what every rank must end up with
  sentencepiece::SentencePieceProcessor one;
  if (!one.Load(argv[1]).ok()) { fprintf(stderr, "load failed\n"); return 1; }
  std::vector<int32_t> want_ids(text.size() + 16 * n + 64);
  std::vector<uint64_t> want_offs(n + 1);
  uint64_t want_total = 0;
  if (!one.EncodeBatchDevice(text.data(), text.size(), offs.data(), n, want_ids.data(), want_ids.size(), want_offs.data(), nullptr, &want_total).ok()) {
    fprintf(stderr, "single-processor encode failed\n"); return 1;
  }
  char id[128];
  if (spmx_rccl_unique_id(id) != 0) { fprintf(stderr, "unique id: %s\n", spmx_gather_last_error()); return 1; }
  std::atomic<int> bad{0};
  std::vector<std::thread> ranks;
  for (int rank = 0; rank < world; ++rank) {
    ranks.emplace_back([&, rank]() {
      sentencepiece::SentencePieceProcessor sp;
      if (!sp.Load(argv[1]).ok()) { ++bad; return; }
      void *comm = nullptr;
      if (spmx_rccl_comm_init(&comm, world, rank, id) != 0) { ++bad; return; }
      const uint64_t lo = n * rank / world, hi = n * (rank + 1) / world, m = hi - lo;       // contiguous shards by sentence
      std::vector<uint64_t> my_offs(m + 1);
      for (uint64_t i = 0; i <= m; ++i) my_offs[i] = offs[lo + i] - offs[lo];
      std::vector<int32_t> ids(text.size() + 16 * m + 64);
      std::vector<uint64_t> io(m + 1, 0);
      uint64_t total = 0;
      if (m && !sp.EncodeBatchDevice(text.data() + offs[lo], offs[hi] - offs[lo], my_offs.data(), m, ids.data(), ids.size(), io.data(), nullptr, &total).ok()) { ++bad; return; }
      std::vector<int32_t> all_ids(want_total + 8, -7);
      std::vector<uint64_t> all_offs(n + 2, 0xCDCD), scratch(sentencepiece::SentencePieceProcessor::GatherScratchWords(world)), rs, ri;
      const sentencepiece::util::Status st = sentencepiece::SentencePieceProcessor::AllGatherIds(
          comm, rank, world, ids.data(), total, io.data(), m, all_ids.data(), all_ids.size(), all_offs.data(), all_offs.size(),
          scratch.data(), &rs, &ri, nullptr);
      if (!st.ok()) { fprintf(stderr, "rank %d: %s\n", rank, st.ToString().c_str()); ++bad; spmx_rccl_comm_destroy(comm); return; }
      for (uint64_t i = 0; i <= n; ++i) if (all_offs[i] != want_offs[i]) { ++bad; break; }
      for (uint64_t i = 0; i < want_total; ++i) if (all_ids[i] != want_ids[i]) { ++bad; break; }
      if (all_ids[want_total] != -7 || all_offs[n + 1] != 0xCDCD || rs[world] != n || ri[world] != want_total) ++bad;
      spmx_rccl_comm_destroy(comm);
    });
  }
  for (std::thread &t : ranks) t.join();
  if (bad.load()) { fprintf(stderr, "%d rank(s) differ\n", bad.load()); return 1; }
  printf("ok %llu %llu\n", static_cast<unsigned long long>(n), static_cast<unsigned long long>(want_total));
  return 0;
}
