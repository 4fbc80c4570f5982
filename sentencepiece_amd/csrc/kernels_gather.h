// Multi-GPU gather of the token ids (spmx_all_gather_ids, gather.cc): after the ids and the per-sentence offsets of
// every rank have arrived, the offsets of rank r's sentences are moved from "ids before me on rank r" to "ids before
// me in the whole job" -- + the ids of the ranks before r.  One launch over all sentences; the rank of a sentence by a
// search over the (at most kMaxRanks + 1) sentence prefix sums held in the kernel's arguments.
#ifndef SPMX_KERNELS_GATHER_H_
#define SPMX_KERNELS_GATHER_H_

namespace spmx {

constexpr int kMaxRanks = 64;

struct RebaseArgs {
  uint64_t *offs;                      // [total_sentences + 1]: every rank's LOCAL offsets back to back; [total] gets the total
  uint32_t world;
  uint64_t sent_before[kMaxRanks + 1]; // sentences on the ranks before r (prefix sums, [world] = total)
  uint64_t ids_before[kMaxRanks + 1];  // ids on the ranks before r ([world] = total)
};

SPMX_DEVICE void rebase_block(const RebaseArgs &a) {
  const uint64_t total = a.sent_before[a.world];
  const uint64_t stride = static_cast<uint64_t>(wv::grid_size()) * 64u;
  for (uint64_t i = static_cast<uint64_t>(wv::block_id()) * 64u + static_cast<uint64_t>(wv::lane()); i <= total; i += stride) {
    if (i == total) { a.offs[i] = a.ids_before[a.world]; continue; }
    uint32_t lo = 0, hi = a.world;                 // sent_before[lo] <= i < sent_before[hi]
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a.sent_before[mid] <= i) lo = mid; else hi = mid;
    }
    a.offs[i] += a.ids_before[lo];
  }
}

}  // namespace spmx
#endif
