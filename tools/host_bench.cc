// Rates of the C++ facade's batch forms on a corpus file (one sentence per line), timed around the calls only:
//   flat    EncodeBatchFlat(text, offsets)                     packed input, CSR output
//   nested  EncodeBatch(vector<string_view>, vector<vector<int>>*)   the reference-shaped batch form
// usage: host_bench model corpus.txt      (prints one JSON line)
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../include/spmx_processor.h"

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  sentencepiece_amd::SentencePieceProcessor sp;
  if (!sp.Load(argv[1]).ok()) { fprintf(stderr, "load failed\n"); return 1; }
  std::ifstream f(argv[2], std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string data = ss.str();
  std::vector<std::string_view> views;
  std::vector<uint64_t> offs(1, 0);
  std::string packed;
  packed.reserve(data.size());
  for (size_t p = 0; p < data.size();) {
    size_t q = data.find('\n', p);
    if (q == std::string::npos) q = data.size();
    views.emplace_back(data.data() + p, q - p);
    packed.append(data, p, q - p);
    offs.push_back(packed.size());
    p = q + 1;
  }
  const size_t n = views.size();
  auto now = [] { return std::chrono::steady_clock::now(); };
  double flat_ms = 1e30, nested_ms = 1e30;
  size_t total = 0;
  for (int it = 0; it < 3; ++it) {
    std::vector<int32_t> ids;
    std::vector<uint64_t> io;
    const auto t0 = now();
    if (!sp.EncodeBatchFlat(packed.data(), offs.data(), n, &ids, &io).ok()) return 1;
    const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
    if (it > 0 && ms < flat_ms) flat_ms = ms;
    total = ids.size();
  }
  // the zero-copy forms: the library's own CSR arrays in an EncodedBatch
  double own_ms = 1e30, own_views_ms = 1e30, single_us = 1e30;
  for (int it = 0; it < 4; ++it) {
    sentencepiece_amd::EncodedBatch eb;
    const auto t0 = now();
    if (!sp.EncodeBatchFlat(packed.data(), offs.data(), n, &eb).ok()) return 1;
    const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
    if (it > 0 && ms < own_ms) own_ms = ms;
    if (eb.total_ids() != total) return 1;
  }
  for (int it = 0; it < 3; ++it) {
    sentencepiece_amd::EncodedBatch eb;
    const auto t0 = now();
    if (!sp.EncodeBatch(views, &eb).ok()) return 1;
    const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
    if (it > 0 && ms < own_views_ms) own_views_ms = ms;
  }
  {   // one sentence per call: Encode(input, &ids) -- the latency of a whole call (H2D, classify, launches, D2H) for one sentence
    std::vector<int> one;
    for (int it = 0; it < 5; ++it) (void)sp.Encode(views[it % n], &one);
    const int reps = 200;
    const auto t0 = now();
    for (int it = 0; it < reps; ++it) (void)sp.Encode(views[(it * 7919) % n], &one);
    single_us = std::chrono::duration<double, std::micro>(now() - t0).count() / reps;
  }
  size_t total2 = 0;
  for (int it = 0; it < 3; ++it) {
    std::vector<std::vector<int>> outs;
    const auto t0 = now();
    if (!sp.EncodeBatch(views, &outs).ok()) return 1;
    const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
    if (it > 0 && ms < nested_ms) nested_ms = ms;
    total2 = 0;
    for (const auto &v : outs) total2 += v.size();
  }
  printf("{\"sentences\": %zu, \"ids\": %zu, \"ids_nested\": %zu, \"flat_ms\": %.2f, \"flat_sentences_per_s\": %.0f, "
         "\"nested_ms\": %.2f, \"nested_sentences_per_s\": %.0f, \"flat_owned_ms\": %.2f, \"flat_owned_sentences_per_s\": %.0f, "
         "\"views_owned_ms\": %.2f, \"views_owned_sentences_per_s\": %.0f, \"single_encode_us\": %.1f}\n",
         n, total, total2, flat_ms, n / flat_ms * 1e3, nested_ms, n / nested_ms * 1e3, own_ms, n / own_ms * 1e3, own_views_ms,
         n / own_views_ms * 1e3, single_us);
  return 0;
}
