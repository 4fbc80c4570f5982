// A few seconds of the long-input paths through the C ABI, no Python: does anything fault?  usage: crash_probe <golden dir>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "spmx.h"

static std::string Slurp(const std::string &p) { std::ifstream f(p, std::ios::binary); std::ostringstream s; s << f.rdbuf(); return s.str(); }

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden";
  const std::string bot = Slurp(dir + "/botchan.txt");
  std::string eng;
  while (eng.size() < (1u << 20)) eng += bot;
  eng.resize(1u << 20);
  for (char &c : eng) if (c == '\n') c = ' ';
  std::vector<std::string> docs = {eng.substr(0, 100000), std::string(349525 * 3, ' '), std::string((1u << 20), 'z') + "!", eng,
                                   std::string(12500, 'x') + " 0123456789", "", "\xef\xb7\xba\xef\xb7\xba\xef\xb7\xba", std::string(5000, '\xff') + "a", "short one"};
  for (size_t i = 0; i < docs[1].size(); i += 3) { docs[1][i] = 'a'; docs[1][i + 1] = 'b'; }
  std::string text;
  std::vector<uint64_t> offs{0};
  for (const auto &d : docs) { text += d; offs.push_back(text.size()); }
  std::vector<const char *> models = {"test_model", "uni1k_uds", "uni1k_suffix", "uni32k", "bpe1k_noesc", "bpe1k"};
  if (argc > 2) models.assign(argv + 2, argv + argc);
  for (const char *m : models) {
    const auto t0 = std::chrono::steady_clock::now();
    spmx_handle *h = nullptr;
    int rc = spmx_create_from_file((dir + "/" + m + ".model").c_str(), 0, &h);
    if (rc) { printf("%s create rc=%d %s\n", m, rc, spmx_last_error(nullptr)); return 2; }
    int32_t *ids = nullptr; uint64_t *io = nullptr; uint8_t *st = nullptr; uint64_t nf = 0;
    rc = spmx_encode_batch_ex(h, text.data(), offs.data(), docs.size(), &ids, &io, &st, &nf);
    printf("%s encode rc=%d ids=%llu failed=%llu", m, rc, rc ? 0ull : (unsigned long long)io[docs.size()], (unsigned long long)nf);
    spmx_free(ids); spmx_free(io); spmx_free(st);
    // spans + normalize on the first (100 KB) document and the short ones
    const uint64_t o2[3] = {0, offs[1], offs[1]};
    int32_t *sid = nullptr; uint32_t *sb = nullptr, *se = nullptr; uint64_t *sio = nullptr;
    rc = spmx_encode_batch_spans(h, text.data(), o2, 2, &sid, &sio, &sb, &se, nullptr, nullptr);
    printf(" spans rc=%d", rc);
    spmx_free(sid); spmx_free(sb); spmx_free(se); spmx_free(sio);
    const auto t1 = std::chrono::steady_clock::now();
    printf(" %.2fs\n", std::chrono::duration<double>(t1 - t0).count());
    fflush(stdout);
    spmx_destroy(h);
  }
  printf("probe done\n");
  return 0;
}
