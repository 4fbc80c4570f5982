#include "dat.h"

#include <algorithm>
#include <cstring>

namespace spmx {
namespace {

struct TmpNode {
  int32_t first_child = -1, next_sibling = -1;
  uint32_t value = 0xFFFFFFFFu;
  uint8_t label = 0;
};

}  // namespace

bool BuildDat(const std::vector<std::pair<std::string, uint32_t>> &keys_in, DatTrie *out, std::string *error) {
  std::vector<std::pair<std::string, uint32_t>> keys = keys_in;
  std::sort(keys.begin(), keys.end());
  // 1. plain trie; children kept in ascending label order because keys are sorted.
  std::vector<TmpNode> nodes(1);
  std::vector<int32_t> last_child(1, -1);
  out->max_key_len = 0;
  for (size_t k = 0; k < keys.size(); ++k) {
    const std::string &key = keys[k].first;
    if (key.empty() || key.find('\0') != std::string::npos) {
      if (error) *error = "trie key is empty or contains NUL";
      return false;
    }
    if (k > 0 && key == keys[k - 1].first) {
      if (error) *error = "duplicate trie key: " + key;
      return false;
    }
    out->max_key_len = std::max<int>(out->max_key_len, static_cast<int>(key.size()));
    int32_t cur = 0;
    for (unsigned char c : key) {
      int32_t lc = last_child[cur];
      if (lc >= 0 && nodes[lc].label == c) {
        cur = lc;
        continue;
      }
      const int32_t nn = static_cast<int32_t>(nodes.size());
      nodes.emplace_back();
      last_child.push_back(-1);
      nodes[nn].label = c;
      if (lc >= 0) nodes[lc].next_sibling = nn; else nodes[cur].first_child = nn;
      last_child[cur] = nn;
      cur = nn;
    }
    nodes[cur].value = keys[k].second;
  }
  // 2. place nodes breadth-first.  Free slots are kept on a doubly linked list
  //    (index 0 is the list head sentinel = the root slot, never free).
  std::vector<uint32_t> &w0 = out->w0;
  std::vector<uint32_t> &val = out->value;
  w0.assign(256, 0);
  val.assign(256, 0xFFFFFFFFu);
  std::vector<uint8_t> occupied(256, 0), base_used(256, 0);
  std::vector<uint32_t> nxt(256), prv(256);
  for (uint32_t i = 0; i < 256; ++i) { nxt[i] = (i + 1) & 255; prv[i] = (i + 255) & 255; }
  auto grow = [&]() {
    const uint32_t old = static_cast<uint32_t>(w0.size());
    w0.resize(old + 256, 0);
    val.resize(old + 256, 0xFFFFFFFFu);
    occupied.resize(old + 256, 0);
    base_used.resize(old + 256, 0);
    nxt.resize(old + 256);
    prv.resize(old + 256);
    const uint32_t tail = prv[0];
    for (uint32_t i = old; i < old + 256; ++i) { nxt[i] = i + 1; prv[i] = i - 1; }
    nxt[tail] = old; prv[old] = tail;
    nxt[old + 255] = 0; prv[0] = old + 255;
  };
  auto take = [&](uint32_t slot) {
    occupied[slot] = 1;
    nxt[prv[slot]] = nxt[slot];
    prv[nxt[slot]] = prv[slot];
  };
  occupied[0] = 1;   // root; stays linked as the sentinel
  base_used[0] = 1;  // base 0 is what childless units carry: never a real base, so their probes always mismatch
  w0[0] = 0;  // the root carries no label: a NUL byte probe from a childless unit (base 0) must not match it
  std::vector<std::pair<int32_t, uint32_t>> queue;  // (tmp node, unit index)
  queue.emplace_back(0, 0u);
  std::vector<uint8_t> labels;
  uint32_t cursor = 0;
  uint32_t hint[256];
  for (uint32_t &x : hint) x = 1;
  for (size_t qi = 0; qi < queue.size(); ++qi) {
    const int32_t tn = queue[qi].first;
    const uint32_t unit = queue[qi].second;
    labels.clear();
    for (int32_t ch = nodes[tn].first_child; ch >= 0; ch = nodes[ch].next_sibling) labels.push_back(nodes[ch].label);
    if (nodes[tn].value != 0xFFFFFFFFu) {
      w0[unit] |= kDatTerminal;
      val[unit] = nodes[tn].value;
    }
    if (labels.empty()) continue;
    uint32_t base = 0;
    bool found = false;
    auto fits = [&](uint32_t f) {
      const uint32_t b = f ^ labels[0];
      if (base_used[b]) return false;
      for (size_t i = 1; i < labels.size(); ++i) if (occupied[b ^ labels[i]]) return false;
      base = b;
      return true;
    };
    // Single-child nodes (the bulk of a trie) can sit in any hole: first fit
    // from the head of the free list.  Branching nodes need several holes in
    // one 256-slot block: next fit from a cursor that only moves forward.
    int tries = 0;
    if (labels.size() == 1) {
      // A free slot f that was refused for label c (base f ^ c already in use) stays refused for c: bases are never
      // released.  The free list is in ascending order, so the scan for c resumes at the first free slot at or after
      // the last one it looked at instead of walking the refused prefix again (that walk was 8 s of a 250k-piece
      // model's load; where the old scan found a slot within its 8192 tries this finds the same one).
      uint32_t &h = hint[labels[0]];
      while (h < occupied.size() && occupied[h]) ++h;
      for (uint32_t f = h < occupied.size() ? h : 0; f != 0 && tries < 8192 && !found; f = nxt[f], ++tries) {
        found = fits(f);
        h = f;
      }
    } else {
      if (cursor == 0 || occupied[cursor]) cursor = nxt[0];
      for (uint32_t f = cursor; f != 0 && tries < 2048 && !found; f = nxt[f], ++tries) {
        found = fits(f);
        if (found) cursor = f;
      }
    }
    while (!found) {       // a fresh block always fits
      const uint32_t old = static_cast<uint32_t>(w0.size());
      if (old + 256 > kDatMaxUnits) {
        if (error) *error = "trie needs more than 4M units";
        return false;
      }
      grow();
      for (uint32_t f = old; f < old + 256 && !found; ++f) found = fits(f);
      if (labels.size() > 1) cursor = old;
    }
    base_used[base] = 1;
    w0[unit] |= base << kDatBaseShift;
    size_t li = 0;
    for (int32_t ch = nodes[tn].first_child; ch >= 0; ch = nodes[ch].next_sibling, ++li) {
      const uint32_t slot = base ^ labels[li];
      take(slot);
      w0[slot] = kDatOccupied | labels[li];
      queue.emplace_back(ch, slot);
    }
  }
  // 3. max number of keys that prefix one key (Darts' trie_results_size_ analogue).
  out->max_prefixes = 0;
  for (const auto &kv : keys) {
    int cnt = 0;
    uint32_t node = 0;
    for (unsigned char c : kv.first) {
      const uint32_t idx = (w0[node] >> kDatBaseShift) ^ c;
      node = idx;
      if (w0[node] & kDatTerminal) ++cnt;
    }
    out->max_prefixes = std::max(out->max_prefixes, cnt);
  }
  return true;
}

int64_t DatFind(const DatTrie &t, const std::string &key) {
  uint32_t node = 0;
  for (unsigned char c : key) {
    const uint32_t idx = (t.w0[node] >> kDatBaseShift) ^ c;
    if (idx >= t.w0.size() || (t.w0[idx] & 0x1FF) != (kDatOccupied | c)) return -1;
    node = idx;
  }
  return node;
}

}  // namespace spmx
