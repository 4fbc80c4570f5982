// Table compiler: ModelData -> flat arrays in the device layout of dev.h.
#ifndef SPMX_TABLES_H_
#define SPMX_TABLES_H_
#include <cstdint>
#include <string>
#include <vector>

#include "dev.h"
#include "model.h"

namespace spmx {

struct HostTables {
  // normalizer
  std::vector<uint32_t> ndarts;
  std::vector<uint8_t> nblob;
  bool charsmap_inner_space = false;      // some charsmap key holds a 0x20 behind its first byte (dev.h kNfWordLocalNorm)
  std::vector<uint32_t> npair;   // 65536 bits, see SpmxDev::npair
  // unigram
  std::vector<U4> ptrie;
  std::vector<uint8_t> plen;     // per id: byte length of the piece as the device sees it (SpmxDev::plen)
  std::vector<U4> umemo, umemo16, uhot, uall, uhot2;
  std::vector<U4> cfirst;                 // first-character table of the piece trie (SpmxDev::cfirst; empty: none)
  std::vector<uint16_t> udisp;            // displacements of uall's perfect hash (SpmxDev::udisp)   // word memo of the word form (SpmxDev::umemo, umemo16, uhot)
  std::vector<float> pscore;     // per id: score (SpmxDev::pscore)
  uint32_t memo_words = 0, memo_candidates = 0;   // entries in the memo / vocabulary strings that are whole words
  // bpe
  std::vector<U2> utrie;
  std::vector<U4> chartab, pairtab, wordtab;
  std::vector<uint32_t> sym_final;
  std::vector<uint16_t> sym_len;
  std::vector<int32_t> byte_ids;
  // decode (kernels_decode.h)
  std::vector<uint32_t> dec_info, dec_off;
  std::vector<uint8_t> dec_bytes;
  // scalars (pointers are filled in by whoever owns the memory)
  SpmxDev scalars{};
  int max_piece_len = 0;
  int max_prefixes = 0;
  // the split form (kernels_matchfold.h) may take this model: unigram, no user-defined pieces, every replacement string of
  // the charsmap valid UTF-8 (then so is the normalized text, and its character starts are its non-continuation bytes)
  bool split_ok = false;
  int max_piece_chars = 0;       // the longest piece in characters (bytes that continue no character), of the device form
};

// Builds everything that depends only on load-time structure.
Status CompileTables(const ModelData &m, HostTables *t);
// The per-id decode tables (kernels_decode.h) from the current piece types.
Status BuildDecodeTables(const ModelData &m, HostTables *t);
// Refreshes the type-dependent bits (UNUSED / USER_DEFINED flags) after
// SetVocabulary / ResetVocabulary without rebuilding tries.
void RefreshTypeFlags(const ModelData &m, HostTables *t);
// The word memo of the unigram word form (dev.h umemo) from the current piece types; sets / clears kNfUniWordwise.
void BuildWordMemo(const ModelData &m, HostTables *t);
// Net effect of ApplyExtraOptions (src/sentencepiece_processor.cc:1019-1064)
// for an option string such as "bos:eos:reverse".
Status CompileExtraOptions(const ModelData &m, const std::string &opts, HostTables *t);
// Points t->scalars at t's own vectors (host execution / emulation).
void BindHostPointers(HostTables *t);

}  // namespace spmx
#endif
