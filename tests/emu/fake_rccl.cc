// TEST INFRASTRUCTURE ONLY: a stand-in for librccl whose ranks are THREADS of one process and whose "device" buffers
// are host memory (the emulated libspmx's fake HIP runtime) -- what spmx_all_gather_ids (csrc/gather.cc) is pointed at
// through SPMX_RCCL_LIB by tests/test_gather.py, so that its world > 1 logic (counts all-gather, exact-size grouped
// sends and receives, peer order, rebasing) runs on the CPU.  It implements the nine entry points gather.cc looks up,
// with RCCL's semantics where they matter here: collectives and grouped point-to-point calls complete only when every
// rank has made the matching call; a receive larger than the matching send is an error.
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct UniqueId { char internal[128]; };
struct World {
  int nranks = 0, arrived = 0, generation = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<const void *> slot;                     // all-gather: every rank's send buffer
  struct Msg { const void *p; size_t bytes; };
  std::map<std::pair<int, int>, std::vector<Msg>> box;  // (from, to) -> messages in posting order
  void Barrier() {
    std::unique_lock<std::mutex> l(mu);
    const int gen = generation;
    if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(l, [&] { return generation != gen; });
  }
};
struct Comm { World *w; int rank; };
std::mutex g_mu;
std::map<uint64_t, World *> g_worlds;
uint64_t g_next = 1;
struct Op { bool send; void *p; size_t bytes; int peer; Comm *c; };
thread_local int t_group = 0;
thread_local std::vector<Op> t_ops;
size_t SizeOf(int dtype) { return dtype <= 1 ? 1 : (dtype <= 3 ? 4 : (dtype <= 5 ? 8 : (dtype == 6 ? 2 : (dtype == 7 ? 4 : 8)))); }

int Flush() {
  if (t_ops.empty()) return 0;
  Comm *c = t_ops[0].c;
  World *w = c->w;
  {
    std::lock_guard<std::mutex> l(w->mu);
    for (const Op &o : t_ops) if (o.send) w->box[{c->rank, o.peer}].push_back({o.p, o.bytes});
  }
  w->Barrier();                                       // every rank has posted its sends
  int rc = 0;
  {
    std::lock_guard<std::mutex> l(w->mu);
    std::map<int, size_t> taken;
    for (const Op &o : t_ops) {
      if (o.send) continue;
      auto &q = w->box[{o.peer, c->rank}];
      size_t &k = taken[o.peer];
      if (k >= q.size() || q[k].bytes != o.bytes) { rc = 5; continue; }     // (ncclInvalidUsage)
      memcpy(o.p, q[k].p, o.bytes);
      ++k;
    }
  }
  w->Barrier();                                       // every rank has read: the boxes may go
  {
    std::lock_guard<std::mutex> l(w->mu);
    for (auto it = w->box.begin(); it != w->box.end();) it = it->first.first == c->rank ? w->box.erase(it) : ++it;
  }
  w->Barrier();
  t_ops.clear();
  return rc;
}
}  // namespace

extern "C" {
int ncclGetUniqueId(UniqueId *id) {
  std::lock_guard<std::mutex> l(g_mu);
  memset(id, 0, sizeof(*id));
  const uint64_t k = g_next++;
  memcpy(id->internal, &k, sizeof(k));
  g_worlds[k] = new World;
  return 0;
}
int ncclCommInitRank(void **comm, int nranks, UniqueId id, int rank) {
  uint64_t k;
  memcpy(&k, id.internal, sizeof(k));
  World *w;
  {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_worlds.find(k);
    if (it == g_worlds.end()) return 4;
    w = it->second;
    std::lock_guard<std::mutex> l2(w->mu);
    if (w->nranks == 0) { w->nranks = nranks; w->slot.assign(nranks, nullptr); }
    if (w->nranks != nranks || rank < 0 || rank >= nranks) return 4;
  }
  *comm = new Comm{w, rank};
  w->Barrier();
  return 0;
}
int ncclCommDestroy(void *comm) { delete static_cast<Comm *>(comm); return 0; }
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, void *) {
  Comm *c = static_cast<Comm *>(comm);
  World *w = c->w;
  const size_t bytes = count * SizeOf(dtype);
  { std::lock_guard<std::mutex> l(w->mu); w->slot[c->rank] = send; }
  w->Barrier();
  for (int r = 0; r < w->nranks; ++r) memcpy(static_cast<char *>(recv) + bytes * r, w->slot[r], bytes);
  w->Barrier();
  return 0;
}
int ncclSend(const void *p, size_t count, int dtype, int peer, void *comm, void *) {
  t_ops.push_back({true, const_cast<void *>(p), count * SizeOf(dtype), peer, static_cast<Comm *>(comm)});
  return t_group ? 0 : Flush();
}
int ncclRecv(void *p, size_t count, int dtype, int peer, void *comm, void *) {
  t_ops.push_back({false, p, count * SizeOf(dtype), peer, static_cast<Comm *>(comm)});
  return t_group ? 0 : Flush();
}
int ncclGroupStart() { ++t_group; return 0; }
int ncclGroupEnd() { return --t_group == 0 ? Flush() : 0; }
const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 5 ? "invalid usage (fake RCCL: a receive without its send, or of another size)" : "error (fake RCCL)"); }
}
