// Corpus packer on the device: a file image with '\n'-terminated lines ->
// the packed form the encode calls take (text without the terminators + n + 1 offsets).
// Line semantics are std::getline's, as in the reference's spm_encode loop (src/spm_encode_main.cc:159-165):
// '\n' ends a line and is dropped, '\r' is kept, a last line without '\n' counts, "a\n" is one line.
//
// Two passes over chunks of kSplitChunk bytes around the scan of kernels.h:
//   count   newlines per chunk                                   -> counts[chunk]
//   (scan)  exclusive prefix                                     -> chunk_base[chunk] = newlines before the chunk
//   write   every '\n' at byte p with rank r (0-based, global)   -> offsets[r + 1] = p - r
//           every other byte at p with r newlines before it      -> text[p - r]
// HBM-bound: B bytes read twice, B - n written, 8 n offsets.
#ifndef SPMX_KERNELS_SPLIT_H_
#define SPMX_KERNELS_SPLIT_H_

namespace spmx {

constexpr uint32_t kSplitChunk = 16384;   // bytes per chunk: 64 lanes x 16 bytes x 16 steps

struct SplitArgs {
  const uint8_t *file;       // 16-byte aligned
  uint64_t bytes;
  uint32_t *counts;          // per chunk (count pass out)
  const uint64_t *chunk_base;   // per chunk + 1 (scan out): newlines before the chunk; [n_chunks] = total
  uint8_t *text;             // bytes - total
  uint64_t *offsets;         // lines + 1
};

// bit k of the result: byte k of the 16-byte block is '\n' (bytes at or past `limit` do not count)
SPMX_DEVICE uint32_t newline_mask(const Q4 &q, uint64_t pos, uint64_t limit) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
    if (c == 0x0Au && pos + static_cast<uint64_t>(k) < limit) m |= 1u << k;
  }
  return m;
}

constexpr uint32_t kSplitStep = 1024;          // bytes per wave step: 64 lanes x 16
constexpr uint32_t kSplitLdsBytes = kSplitStep + 32;

// `stage` (write pass only): kSplitLdsBytes of LDS, 16-byte aligned.  The kept bytes of a step are packed there and
// leave as 16-byte stores aligned on the destination (the packed text of a step is contiguous: every byte moves
// left by the number of newlines before it).
template <bool WRITE>
SPMX_DEVICE void split_block(const SplitArgs &a, uint8_t *stage) {
  const int lane = wv::lane();
  const uint64_t chunks = (a.bytes + kSplitChunk - 1) / kSplitChunk;
  for (uint64_t ch = static_cast<uint64_t>(wv::block_id()); ch < chunks; ch += static_cast<uint64_t>(wv::grid_size())) {
    const uint64_t c0 = ch * kSplitChunk;
    uint64_t run = WRITE ? a.chunk_base[ch] : 0;     // newlines before the current step
    uint32_t total = 0;
    for (uint32_t step = 0; step < kSplitChunk / kSplitStep; ++step) {
      const uint64_t s0 = c0 + step * kSplitStep;
      if (s0 >= a.bytes) break;
      const uint64_t pos = s0 + static_cast<uint64_t>(lane) * 16u;
      Q4 q{0, 0, 0, 0};
      if (pos < a.bytes) q = *reinterpret_cast<const Q4 *>(a.file + pos);   // (the allocation is padded to 16 bytes)
      const uint32_t m = newline_mask(q, pos, a.bytes);
      const int cnt = wv::popc64(static_cast<uint64_t>(m));
      int step_total = 0;
      const int before = wave_excl_scan(cnt, lane, &step_total);
      if (WRITE) {
        const uint64_t left = a.bytes - s0;
        const uint32_t n_in = left < kSplitStep ? static_cast<uint32_t>(left) : kSplitStep;
        const uint32_t n_out = n_in - static_cast<uint32_t>(step_total);
        uint8_t *dst = a.text + (s0 - run);
        if (pos < a.bytes) {
          uint64_t r = run + static_cast<uint64_t>(before);          // newlines before this lane's block
          uint32_t o = static_cast<uint32_t>(lane) * 16u - static_cast<uint32_t>(before);
          const uint32_t w[4] = {q.x, q.y, q.z, q.w};
          if (m == 0 && pos + 16 <= a.bytes && (o & 3u) == 0) {
            uint32_t *s32 = reinterpret_cast<uint32_t *>(stage + o);
            s32[0] = w[0]; s32[1] = w[1]; s32[2] = w[2]; s32[3] = w[3];
          } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              const uint64_t p = pos + static_cast<uint64_t>(k);
              if (p < a.bytes) {
                if ((m >> k) & 1u) { a.offsets[r + 1] = p - r; ++r; }
                else stage[o++] = static_cast<uint8_t>((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
              }
            }
          }
        }
        wv::sync();
        uint32_t head = static_cast<uint32_t>((16u - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
        if (head > n_out) head = n_out;
        const uint32_t nq = (n_out - head) >> 4;                      // <= 64
        const uint32_t tail = n_out - head - (nq << 4);
        if (static_cast<uint32_t>(lane) < head) dst[lane] = stage[lane];
        if (static_cast<uint32_t>(lane) < nq) {
          const uint32_t s = head + static_cast<uint32_t>(lane) * 16u;
          const uint32_t *s32 = reinterpret_cast<const uint32_t *>(stage) + (s >> 2);
          const uint32_t sh = (s & 3u) * 8u;
          const uint32_t v0 = s32[0], v1 = s32[1], v2 = s32[2], v3 = s32[3];
          Q4 out{v0, v1, v2, v3};
          if (sh) {
            const uint32_t v4 = s32[4];
            out.x = (v0 >> sh) | (v1 << (32u - sh)); out.y = (v1 >> sh) | (v2 << (32u - sh));
            out.z = (v2 >> sh) | (v3 << (32u - sh)); out.w = (v3 >> sh) | (v4 << (32u - sh));
          }
          *reinterpret_cast<Q4 *>(dst + s) = out;
        }
        if (static_cast<uint32_t>(lane) < tail) {
          const uint32_t s = head + (nq << 4) + static_cast<uint32_t>(lane);
          dst[s] = stage[s];
        }
        wv::sync();
      }
      run += static_cast<uint64_t>(step_total);
      total += static_cast<uint32_t>(step_total);
    }
    if (!WRITE && lane == 0) a.counts[ch] = total;
  }
  if (WRITE && wv::block_id() == 0 && lane == 0) {
    const uint64_t nl = a.chunk_base[chunks];
    a.offsets[0] = 0;
    // a last line without its '\n' (std::getline returns it)
    if (a.bytes > 0 && a.file[a.bytes - 1] != 0x0Au) a.offsets[nl + 1] = a.bytes - nl;
  }
}

}  // namespace spmx
#endif
