// XOR double-array trie in the engine's own device layout.
//
// The reference rebuilds its piece trie at every load (src/unigram_model.cc:
// 608-650) and only the *search semantics* of Darts (third_party/darts_clone/
// darts.h:467-547: prefixes in increasing length) are part of the path, so the
// device index is free to use its own unit format.  Ours folds Darts' separate
// leaf unit into the node itself so that one load answers "does the edge
// exist", "where are the children" and "does a key end here":
//
//   w0 = base << 10 | terminal << 9 | occupied << 8 | label
//   child(node, c) = units[base(node) ^ c], valid iff (w0 & 0x1FF) == (0x100 | c)
//
// As in Darts, every node gets a unique base, so the label test alone is
// sufficient (a probe from another parent lands on a different label).  The
// array length is a multiple of 256, so base ^ c never leaves the array.
// Unit 0 is the root.  Payload words (value, score, ...) are attached by the
// caller per unit index.
#ifndef SPMX_DAT_H_
#define SPMX_DAT_H_
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace spmx {

constexpr uint32_t kDatOccupied = 1u << 8;
constexpr uint32_t kDatTerminal = 1u << 9;
constexpr int kDatBaseShift = 10;
constexpr uint32_t kDatMaxUnits = 1u << 22;

struct DatTrie {
  std::vector<uint32_t> w0;     // transition word per unit
  std::vector<uint32_t> value;  // key index for terminal units, 0xFFFFFFFF otherwise
  int max_key_len = 0;
  int max_prefixes = 0;  // max number of keys that are prefixes of one key (>= 1 if any key)
};

// keys must be unique and non-empty, without NUL bytes.  value[i] is stored at
// the terminal unit of keys[i].  Returns false if the trie needs more than
// kDatMaxUnits units or a key is invalid (error filled).
bool BuildDat(const std::vector<std::pair<std::string, uint32_t>> &keys, DatTrie *out, std::string *error);

// Host-side walk used by the table compiler and by load-time checks.
// Returns the unit index reached after consuming `key` or -1.
int64_t DatFind(const DatTrie &t, const std::string &key);

}  // namespace spmx
#endif
