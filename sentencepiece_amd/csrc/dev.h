// Device-visible model description shared by the host table compiler and the
// HIP kernels.  Plain pointers + scalars; passed to kernels by value (kernarg).
#ifndef SPMX_DEV_H_
#define SPMX_DEV_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define SPMX_HD __host__ __device__
#else
#define SPMX_HD
#endif

namespace spmx {

struct U2 { uint32_t x, y; };
struct U4 { uint32_t x, y, z, w; };
struct alignas(16) Q4 { uint32_t x, y, z, w; };   // one aligned 16-byte load
struct alignas(8) D2 { uint32_t x, y; };           // one aligned 8-byte load
struct alignas(16) L2 { uint64_t x, y; };          // one aligned 16-byte store of two offsets

// normalizer flags
enum : uint32_t {
  kNfAddDummyPrefix = 1u << 0,
  kNfRemoveExtraWs = 1u << 1,
  kNfEscapeWs = 1u << 2,
  kNfWsSuffix = 1u << 3,
  kNfHasCharsmap = 1u << 4,  // precompiled_charsmap present (Darts units used verbatim)
  kNfByteFallback = 1u << 5,
  kNfReverse = 1u << 6,      // net effect of the extra options (see tables.cc)
  kNfHasUserDefined = 1u << 7,
  kNfHasUnused = 1u << 8,    // some piece is currently UNUSED (BPE resegmentation armed)
  // the space symbol U+2581 (E2 96 81) is ONE byte, kSpByte, in the normalized text held in LDS and
  // in the keys of the piece trie (tables.cc decides; ids do not depend on the encoding of the text)
  kNfCompressSp = 1u << 9,
  // BPE only: every piece is either a run of U+2581 or has no U+2581 after its first character (and kNfCompressSp
  // holds, no whitespace-as-suffix, no user-defined symbols), so no merge ever joins a character that is not
  // U+2581 with a U+2581 to its right: a sentence can be segmented word by word, a word being a run of U+2581
  // plus what follows up to the next one (kernels_bpe_stream.h)
  kNfBpeWordwise = 1u << 10,
  // unigram only: the word form (kernels_word.h) applies -- no piece has a space symbol after its first character, the
  // normalizer adds a dummy prefix, removes extra whitespace and escapes whitespace with the one-byte space symbol, no
  // charsmap rule starts with a byte 0x20 .. 0x7E, and the word memo (umemo) is not empty
  kNfUniWordwise = 1u << 11,
  // the word form also takes words that are NOT plain ASCII (kernels_word.h word_resolve_block normalizes a collected word
  // by itself): the normalizer removes extra whitespace, and no charsmap key holds a 0x20 behind its first byte -- so what
  // Normalize makes of a word (a run of bytes other than 0x20) does not depend on its neighbours, and the sentence's
  // normalized text is the words' own, each behind one space symbol (tables.cc BuildWordMemo)
  kNfWordLocalNorm = 1u << 12,
};

constexpr uint32_t kNfiltCps = 0x20000u;
constexpr uint32_t kNfiltWords = kNfiltCps / 32u;

// One-byte stand-in for U+2581 under kNfCompressSp.  0xFF never occurs in valid UTF-8, and the normalizer's
// output is valid UTF-8 whenever the model's own strings are (checked at load).
constexpr uint32_t kSpByte = 0xFFu;


// ptrie payload (U4: w0, idflags, score bits, 0)
constexpr uint32_t kPtUnused = 1u << 31;
constexpr uint32_t kPtUserDefined = 1u << 30;
constexpr uint32_t kPtIdMask = (1u << 30) - 1;

constexpr uint32_t kSymNone = 0xFFFFFFFFu;
// sym_final word: final id | flags
constexpr uint32_t kSfUnused = 1u << 31;
constexpr uint32_t kSfControl = 1u << 30;
constexpr uint32_t kSfIdMask = (1u << 30) - 1;

// unit word of the device tries (dat.h): base << 10 | terminal << 9 | occupied << 8 | label
constexpr int kDatBaseShiftDev = 10;
constexpr uint32_t kDatTerminalDev = 1u << 9;
constexpr int kMaxResegDepth = 64;

constexpr int kMaxExtra = 4;       // bos/eos ids on either side
constexpr int kMaxPieceBytes = 120; // unigram: the back-pointer word holds the piece length in 7 bits; the score ring lives in LDS

struct SpmxDev {
  // ---- normalizer (reference: src/normalizer.cc:71-253) ----
  const uint32_t *ndarts; // the model's own Darts double-array units (part of the .model format)
  const uint8_t *nblob;   // NUL-terminated replacement strings
  uint32_t ndarts_n, nblob_n;
  // bit b (b < 128): no charsmap key is `b` alone or `b` followed by another ASCII byte, so an ASCII byte b
  // whose successor is ASCII (or the end of the input) cannot begin a rule and needs no trie probe
  uint32_t ascii_safe[4];
  // bit (b0 << 8 | b1): some charsmap key starts with the bytes b0 b1, or is b0 alone (then the whole row b0 is
  // set); b1 = 0 stands for "no second byte".  A clear bit proves that no rule starts at a position without a
  // single Darts probe -- most CJK ideographs and ASCII pairs (tables.cc)
  const uint32_t *npair;   // [2048] (+ the code-point filters below when nfilt != 0)
  // code-point filters of the charsmap keys behind npair's 2048 words (tables.cc), three bitmaps of kNfiltWords words over
  // the code points below kNfiltCps: [0] some key STARTS with the character, [1] the character ALONE is a key, [2] the
  // character is the SECOND character of some key.  NormalizePrefix's longest-key search (normalizer.cc:218-228) finds
  // nothing at a character c followed by d unless starts[c] and (alone[c] or second[d]) -- so most kana, Cyrillic and
  // accented letters (keys only together with a combining mark) need no trie walk.  0: not built (a key that is not UTF-8).
  // Behind the bitmaps, kNfiltCps words: the rule of a key that is exactly ONE character -- offset into nblob | length << 24
  // of a replacement of 1 .. 255 bytes without a space; 0: none such (normalize_wave's common sweep applies these itself).
  uint32_t nfilt;
  uint32_t flags;
  // normalized length <= expand_max * raw length + 3: the largest growth of any NormalizePrefix result (a charsmap
  // rule's replacement over its key, U+FFFD for one malformed byte, a space escaped to U+2581), tables.cc
  uint32_t expand_max;
  // ---- unigram (reference: src/unigram_model.cc:889-1020) ----
  const U4 *ptrie;        // piece trie with inline id / flags / score
  const uint8_t *plen;    // per id: the piece's byte length in the device form of the text (the short back-pointer form)
  // first-CHARACTER table (null: none): by code point U+0080 .. U+FFFF the unit of ptrie reached after the character's two
  // or three UTF-8 bytes, its label byte replaced by the character's byte length (0x100 clear: no piece starts with the
  // character).  Built for vocabularies with many pieces in multi-byte scripts when no piece ends inside a character
  // (tables.cc BuildFirstCharTable); the streaming kernel starts a walk there instead of spelling the character
  // (kernels_stream.h unigram_stream_lane).
  const U4 *cfirst;
  float unk_score;        // min_score - 10.0f
  float max_score;
  // The wave-cooperative form's fold in FLOAT arithmetic (kernels_uniwave.h): while every best_path_score it adds to
  // stays above -uw_f32_limit, the reference's (double)score + (double)best is EXACT (two floats whose exponents differ
  // by at most 28 have a sum of at most 53 bits), so the float it stores is the float sum and its comparison is the float
  // sum's plus, on a tie, the sign of the sum's rounding error.  0: not for this model (a score above 0, user-defined
  // pieces -- their score is a double --, scores too far apart).  tables.cc UwFloatLimit.
  float uw_f32_limit;
  int32_t unk_id;
  // ---- id post-processing (reference: src/sentencepiece_processor.cc:547-636) ----
  const int32_t *byte_ids;   // [256]
  int32_t n_prefix, n_suffix;
  int32_t prefix_ids[kMaxExtra], suffix_ids[kMaxExtra];
  uint32_t extra_eos;               // bit i: prefix_ids[i] is an eos (its span is the end of the input, a bos's is 0);
                                    // bit kMaxExtra + i: the same for suffix_ids[i]
  // ---- decode (reference: src/sentencepiece_processor.cc:761-925; kernels_decode.h) ----
  const uint32_t *dec_info;  // per id: kind | flags | byte value
  const uint32_t *dec_off;   // per id + 1: offsets into dec_bytes
  const uint8_t *dec_bytes;  // decoded piece bytes (U+2581 -> ' ', unknown -> unk_surface)
  // ---- BPE (reference: src/bpe_model.cc:38-203) ----
  const U2 *utrie;        // user-defined symbols only (PrefixMatcher; normalizer and BPE both use it)
  const U4 *chartab;      // {bytes, len, sym, 0}   open addressing, empty: len == 0
  const U4 *pairtab;      // {symL, symR, merged sym, score bits}   empty: symL == kSymNone
  const uint32_t *sym_final;  // per symbol: final id | flags
  const uint16_t *sym_len;    // per symbol: byte length
  uint32_t chartab_mask, pairtab_mask;
  // word table (BPE, word-wise models without UNUSED pieces): the segmentation of every vocabulary string that is a
  // whole word, computed at load by the same merges -- a word's pieces are a pure function of its bytes, so a word found
  // here needs no merge loop.  Slot = two U4: {16 key bytes, zero padded} {len | n_ids << 8 | piece lengths << 16 (5 bits
  // each), id0, id1, id2}; empty: len == 0.  Open addressing on HashWord.
  const U4 *wordtab;
  uint32_t wordtab_mask;      // 0: no table
  // word memo (unigram, kNfUniWordwise; kernels_word.h): EncodeOptimized of every vocabulary string that is a whole
  // word -- the space symbol and then 1 .. 16 bytes 0x21 .. 0x7E -- computed at load in double, with the magnitude bmax
  // of the accumulated score below which the float arithmetic of the reference provably takes the same decisions.
  //   umemo16  16-byte entries, ids below 65535.  Words of 11 / 12 bytes that are ONE piece: {12 raw bytes of the word
  //            WITHOUT its space symbol, padded with 0x20 (kernels_word.h key_dword); id | e << 16 | sc << 24}.  Words of
  //            up to 10 bytes that are one or TWO pieces: {8 key bytes; key bytes 8, 9 | second id << 16 (0xFFFF: none);
  //            id | e << 16 | kMemo16TwoPiece | sc << 24}.  Valid while |score| < 2^e (the power of two below bmax; e <= 126);
  //            sc = the sum over the pieces of ceil(|piece score|) + 1, what the word adds to the first pass's bound of
  //            |score|.  empty: w == 0xFFFFFFFF.  uhot: the kWordHotSlots likeliest of them, direct-mapped (the kernels
  //            keep it in LDS).
  //   umemo    the other words (two pieces, 13 .. 16 bytes): two U4 {16 key bytes, 0x20 padded} {id0, id1 or 0xFFFFFFFF, sc0 + sc1
  //            (float bits), bmax (float bits)}; empty: id0 == 0xFFFFFFFF.
  // Open addressing on HashWordKey.  pscore: score per piece id (the second pass replays the exact score).
  //   uhot2    the kWordHotSlots likeliest words of umemo16 again, direct-mapped on HashWordKey of the whole 16-byte key
  //            (the word-per-lane kernels need that hash anyway).
  //   uall     EVERY word of the memo in umemo's 32-byte format behind a PERFECT hash (hash, displace: UallSlot below):
  //            the bucket UallHash2(key) & (kUallBuckets - 1) names a displacement udisp[bucket] (the kernels keep the
  //            8 KB of them in LDS), and the word -- if the memo has it -- sits at exactly one slot: the word-per-lane
  //            kernels (kernels_wordwave.h) find a word that is not in the LDS table by ONE probe, and a word the memo
  //            lacks costs one probe too (no walk: a wavefront's lanes wait for its longest).  uall_perfect = 0 (a
  //            vocabulary too large for 4096 buckets): open addressing on HashWordKey, walked.
  const U4 *umemo16;
  const U4 *uhot;
  const U4 *umemo;
  const U4 *uall;
  const U4 *uhot2;
  const uint16_t *udisp;   // [kUallBuckets]
  const float *pscore;
  uint32_t umemo16_mask, umemo_mask, uall_mask, uall_perfect;
  uint32_t n_pieces;      // symbols below this are piece ids (their own final id); the rest are extra characters
  int32_t model_type;     // 1 unigram, 2 bpe
};

// ptrie unit word w: 32-bit summary of the node's child labels, bit ChildBit(c) set for every child byte c.
// A clear bit proves "no child c" without a probe (the walk's last, failing probe is usually predictable:
// 97 % on the round-1 bench corpus); a set bit means "probe".
SPMX_HD inline uint32_t ChildBit(uint32_t c) { return c & 31u; }

SPMX_HD inline uint32_t HashPair(uint32_t a, uint32_t b) {
  uint64_t h = (static_cast<uint64_t>(a) << 32 | b) * 0x9E3779B97F4A7C15ull;
  return static_cast<uint32_t>(h >> 32);
}
SPMX_HD inline uint32_t HashWord(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
  uint32_t h = k0 * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + k1 * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + k2 * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + k3 * 0x27D4EB2Fu;
  return h ^ (h >> 15);
}
// word memo of the unigram word form (kernels_word.h): cheap on the vector ALU -- three rotates, one multiply
SPMX_HD inline uint32_t HashWordKey(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
  const uint32_t h = k0 ^ ((k1 << 9) | (k1 >> 23)) ^ ((k2 << 18) | (k2 >> 14)) ^ ((k3 << 27) | (k3 >> 5));
  const uint32_t g = h * 0x9E3779B1u;
  return g ^ (g >> 15);
}
#ifndef SPMX_HOT_SLOTS
#define SPMX_HOT_SLOTS 256
#endif
// The LDS copy of the likeliest words (uhot / uhot2), a power of two (A/B builds: -DSPMX_HOT_SLOTS=).  2048 slots until round
// 6; measured then on C2 (profiles/r06_hot_slots_waves.txt): the first word round takes 3.085 / 3.076 / 3.067 ms with 2048 /
// 1024 / 512 slots at 12 wavefronts per CU -- a word the table lacks is answered by `uall` from L2, whose probe every lane
// issues anyway -- and the 28 KB the table gives back are two more wavefronts per CU: 2.96 ms at 14.
constexpr uint32_t kWordHotSlots = SPMX_HOT_SLOTS;
// the perfect hash of `uall` (dev.h SpmxDev): where the key with hashes h1 = HashWordKey, h2 = UallHash2 sits under its
// bucket's displacement d.  UallHash2 mixes the bytes another way than HashWordKey does (two keys that agree in one
// differ in the other), at the cost of a few rotates: no second multiply.
constexpr uint32_t kUallBuckets = 4096;
SPMX_HD inline uint32_t UallHash2(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t h1) {
  const uint32_t m = k0 ^ ((k1 << 7) | (k1 >> 25)) ^ ((k2 << 13) | (k2 >> 19)) ^ ((k3 << 21) | (k3 >> 11));
  const uint32_t t = m ^ ((h1 << 11) | (h1 >> 21));
  return t ^ (t >> 16) ^ (t >> 7);
}
SPMX_HD inline uint32_t UallBucket(uint32_t h2) { return h2 & (kUallBuckets - 1u); }
SPMX_HD inline uint32_t UallSlot(uint32_t h1, uint32_t h2, uint32_t d, uint32_t mask) { return (h1 + d * ((h2 >> 12) | 1u)) & mask; }
constexpr uint32_t kWordKeyBytes = 16;   // words longer than this are not looked up
constexpr uint32_t kMemo16TwoPiece = 1u << 23;   // umemo16 meta word: the entry is of the two-piece form (a word of up to 10 bytes)
constexpr uint32_t kWordMaxIds = 3;
SPMX_HD inline uint32_t HashChar(uint32_t bytes, uint32_t len) {
  uint64_t h = (static_cast<uint64_t>(len) << 32 | bytes) * 0xD6E8FEB86659FD93ull;
  return static_cast<uint32_t>(h >> 32);
}

}  // namespace spmx
#endif
