// spmx_all_gather_ids and the spmx_rccl_* helpers (include/spmx.h): the token ids of every rank on every rank, over
// RCCL, from a C / C++ host -- what BASELINE.json's north_star names ("shards the input corpus across the 8 GPUs of one
// node with a RCCL all-gatherv of the token-id output over xGMI"; the per-sentence semantics are the Python wrapper's
// batch form, python/src/sentencepiece/sentencepiece.i:245-267: the job's sentences in order, each with its own ids).
//
// One process per GPU.  librccl is NOT a link-time dependency of libspmx.so: its entry points are looked up at the
// first call (dlopen of SPMX_RCCL_LIB or "librccl.so"), so single-GPU users never load it.  The gather is
//
//   counts   ncclAllGather of {sentences, ids} of every rank (two uint64 per rank, through device memory), read back;
//   payload  one ncclGroup of exact-size point-to-point transfers: to every peer my ids and my per-sentence offsets,
//            from every peer theirs, straight into their places in the gathered CSR -- xGMI is point to point, so
//            world - 1 concurrent sends per rank is the pattern the links are built for, and nothing is padded to the
//            largest rank (the all-gather of equal-sized blocks sentencepiece_amd/sharding.py also offers is);
//   rebase   one small launch moves every rank's offsets from "ids before me on my rank" to "in the whole job".
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/spmx.h"
#include "launch.h"

using namespace spmx;

namespace {

// the part of rccl.h this file needs (ncclResult_t 0 = ncclSuccess; ncclDataType_t numbers: rccl.h:459-470)
struct NcclUniqueId { char internal[128]; };
enum { kNcclInt32 = 2, kNcclUint64 = 5 };
struct RcclApi {
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string error;
  bool ok = false;
};

RcclApi &Rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *path = getenv("SPMX_RCCL_LIB");
    void *lib = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib && !(path && *path)) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { api.error = std::string("RCCL is not loadable: ") + dlerror(); return; }
    auto sym = [&](const char *name) -> void * {
      void *p = dlsym(lib, name);
      if (!p && api.error.empty()) api.error = std::string("RCCL lacks ") + name;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.ok = api.error.empty();
  });
  return api;
}

thread_local std::string t_gather_error;

int FailG(int code, const std::string &msg) {
  t_gather_error = msg;
  return code;
}

#define RCCL_OR_RETURN(api, expr)                                                                              \
  do {                                                                                                         \
    const int r_ = (expr);                                                                                     \
    if (r_ != 0) return FailG(13, std::string(#expr) + ": " + ((api).GetErrorString ? (api).GetErrorString(r_) : "RCCL error")); \
  } while (0)
#define HIPG_OR_RETURN(expr)                                                                 \
  do {                                                                                       \
    const hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) return FailG(13, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace

constexpr int kGatherWords = 5;      // per rank in the counts all-gather: sentences, ids, id capacity, offset capacity, arguments valid

extern "C" {

const char *spmx_gather_last_error(void) { return t_gather_error.c_str(); }

int spmx_rccl_unique_id(void *id128) {
  RcclApi &api = Rccl();
  if (!api.ok) return FailG(14, api.error);
  if (!id128) return FailG(3, "null id buffer");
  NcclUniqueId id;
  RCCL_OR_RETURN(api, api.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int spmx_rccl_comm_init(void **comm, int world, int rank, const void *id128) {
  RcclApi &api = Rccl();
  if (!api.ok) return FailG(14, api.error);
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return FailG(3, "bad communicator arguments");
  NcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  RCCL_OR_RETURN(api, api.CommInitRank(comm, world, id, rank));
  return 0;
}

int spmx_rccl_comm_destroy(void *comm) {
  RcclApi &api = Rccl();
  if (!api.ok) return FailG(14, api.error);
  if (comm) RCCL_OR_RETURN(api, api.CommDestroy(comm));
  return 0;
}

uint64_t spmx_gather_scratch_words(int world) {
  return world >= 1 && world <= kMaxRanks ? static_cast<uint64_t>(kGatherWords) * (1u + static_cast<uint64_t>(world)) : 0u;
}

int spmx_all_gather_ids(void *nccl_comm, int rank, int world, const int32_t *d_ids, uint64_t n_ids,
                        const uint64_t *d_id_offsets, uint64_t n_sentences, int32_t *d_all_ids, uint64_t all_ids_capacity,
                        uint64_t *d_all_id_offsets, uint64_t all_offsets_capacity, uint64_t *d_scratch,
                        uint64_t *rank_sentences, uint64_t *rank_ids, void *stream_) {
  RcclApi &api = Rccl();
  if (!api.ok) return FailG(14, api.error);
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return FailG(3, "world must be 1 .. 64 and rank inside it");
  if (!nccl_comm || !d_scratch) return FailG(3, "null communicator or scratch");   // (the only errors decided by one rank alone)
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // ---- counts: {sentences, ids} of every rank -- and what its output buffers hold: whether the gathered CSR fits is
  // decided by ALL ranks from the same numbers, so that either every rank goes on to the transfers or none does (a rank
  // that returned early on a capacity only it knew to be too small would leave its peers waiting in ncclSend / ncclRecv) ----
  const bool args_ok = !(n_sentences && !d_id_offsets) && !(n_ids && !d_ids);
  const uint64_t mine[kGatherWords] = {n_sentences, n_ids, d_all_ids ? all_ids_capacity : 0, d_all_id_offsets ? all_offsets_capacity : 0,
                                       args_ok ? 1u : 0u};
  uint64_t *d_mine = d_scratch, *d_all = d_scratch + kGatherWords;     // scratch: kGatherWords * (1 + world) uint64
  HIPG_OR_RETURN(hipMemcpyAsync(d_mine, mine, sizeof(mine), hipMemcpyHostToDevice, stream));
  RCCL_OR_RETURN(api, api.AllGather(d_mine, d_all, kGatherWords, kNcclUint64, nccl_comm, stream));
  uint64_t got[kGatherWords * kMaxRanks];
  HIPG_OR_RETURN(hipMemcpyAsync(got, d_all, sizeof(uint64_t) * kGatherWords * static_cast<size_t>(world), hipMemcpyDeviceToHost, stream));
  HIPG_OR_RETURN(hipStreamSynchronize(stream));
  uint64_t all[2 * kMaxRanks];
  RebaseArgs ra{};
  ra.world = static_cast<uint32_t>(world);
  for (int r = 0; r < world; ++r) {
    all[2 * r] = got[kGatherWords * r];
    all[2 * r + 1] = got[kGatherWords * r + 1];
    ra.sent_before[r + 1] = ra.sent_before[r] + all[2 * r];
    ra.ids_before[r + 1] = ra.ids_before[r] + all[2 * r + 1];
  }
  if (rank_sentences) memcpy(rank_sentences, ra.sent_before, sizeof(uint64_t) * static_cast<size_t>(world + 1));
  if (rank_ids) memcpy(rank_ids, ra.ids_before, sizeof(uint64_t) * static_cast<size_t>(world + 1));
  const uint64_t total_s = ra.sent_before[world], total_i = ra.ids_before[world];
  for (int r = 0; r < world; ++r)
    if (!got[kGatherWords * r + 4])
      return FailG(3, "rank " + std::to_string(r) + " passed a null id / offset buffer with a non-zero size: no rank transfers anything");
  for (int r = 0; r < world; ++r) {
    const uint64_t cap_i = got[kGatherWords * r + 2], cap_o = got[kGatherWords * r + 3];
    if (total_i > cap_i || total_s + 1 > cap_o)
      return FailG(8, "the gathered CSR needs " + std::to_string(total_i) + " ids and " + std::to_string(total_s + 1) + " offsets; rank " +
                          std::to_string(r) + "'s buffers hold " + std::to_string(cap_i) + " / " + std::to_string(cap_o) +
                          " (every rank returns this: no rank transfers anything)");
  }
  // ---- payload: exact sizes, point to point, one group ----
  RCCL_OR_RETURN(api, api.GroupStart());
  int in_group = 0;                       // (a call that fails inside the group must not leave it open: the first failure is kept, the group closed)
  for (int k = 1; k < world && in_group == 0; ++k) {
    const int to = (rank + k) % world, from = (rank - k + world) % world;    // (every rank a different peer per step)
    if (n_ids && in_group == 0) in_group = api.Send(d_ids, n_ids, kNcclInt32, to, nccl_comm, stream);
    if (n_sentences && in_group == 0) in_group = api.Send(d_id_offsets, n_sentences, kNcclUint64, to, nccl_comm, stream);
    if (all[2 * from + 1] && in_group == 0) in_group = api.Recv(d_all_ids + ra.ids_before[from], all[2 * from + 1], kNcclInt32, from, nccl_comm, stream);
    if (all[2 * from] && in_group == 0) in_group = api.Recv(d_all_id_offsets + ra.sent_before[from], all[2 * from], kNcclUint64, from, nccl_comm, stream);
  }
  const int ended = api.GroupEnd();
  if (in_group != 0) return FailG(13, std::string("ncclSend / ncclRecv: ") + (api.GetErrorString ? api.GetErrorString(in_group) : "RCCL error"));
  RCCL_OR_RETURN(api, ended);
  if (n_ids) HIPG_OR_RETURN(hipMemcpyAsync(d_all_ids + ra.ids_before[rank], d_ids, n_ids * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  if (n_sentences) HIPG_OR_RETURN(hipMemcpyAsync(d_all_id_offsets + ra.sent_before[rank], d_id_offsets, n_sentences * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
  // ---- rebase ----
  ra.offs = d_all_id_offsets;
  uint64_t grid = (total_s + 1 + 63) / 64;
  if (grid > 2048) grid = 2048;
  HIPG_OR_RETURN(LaunchRebase(ra, static_cast<int>(grid), stream));
  return 0;
}

}  // extern "C"
