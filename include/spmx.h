/* spmx -- C ABI of the MI355X batch tokenization engine (libspmx.so).
 *
 * Drop-in boundary for ONE path of google/sentencepiece: text -> token ids
 * (normalize -> unigram Viterbi / BPE merge -> id post-processing).  Plain
 * pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference tree).  Return values are util::StatusCode numbers
 * (src/sentencepiece_processor.h:34-52): 0 = OK, 13 = INTERNAL, ...; the text
 * of the last error is available from spmx_last_error().  No C++ exception
 * crosses the boundary, as in the reference (only Status values).
 *
 * Thread safety: like SentencePieceProcessor's const methods, a loaded handle
 * may be used for encoding / decoding from several host threads at once: every
 * call leases its own workspace and stream, so the calls overlap on the GPU.
 * The mutators (set_encode_extra_options, set_vocabulary, ...) must not race
 * with them, as in the reference.  spmx_last_error() is per calling thread.
 */
#ifndef SPMX_H_
#define SPMX_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmx_handle spmx_handle;

/* ---- load ---------------------------------------------------------------
 * SentencePieceProcessor::LoadFromSerializedProto(serialized)
 *   (src/sentencepiece_processor.h:261, .cc:234-240) followed by the Load()
 *   body (.cc:242-281): model factory (unigram | bpe), normalizer, prefix
 *   matcher.  The model's tables are compiled into the device layout and
 *   uploaded to GPU `device` (hip device ordinal).  Fails loudly (UNAVAILABLE)
 *   when no HIP device is usable: there is no CPU fallback. */
int spmx_create(const void *model_bytes, uint64_t n_bytes, int device, spmx_handle **out);
/* SentencePieceProcessor::Load(filename) (src/sentencepiece_processor.h:245). */
int spmx_create_from_file(const char *filename, int device, spmx_handle **out);
void spmx_destroy(spmx_handle *h);

/* util::Status::ToString() of the calling thread's last failing call (h is ignored and may be NULL). */
const char *spmx_last_error(const spmx_handle *h);

/* ---- configuration ------------------------------------------------------
 * SetEncodeExtraOptions("bos:eos:reverse") (src/sentencepiece_processor.h:267). */
int spmx_set_encode_extra_options(spmx_handle *h, const char *options);
/* SetDecodeExtraOptions("bos:eos:reverse") (src/sentencepiece_processor.h:270, .cc:288-291): applied to the pieces of
 * every later Decode before they are turned into text (.cc:819).  Same option syntax and error text as the encode side. */
int spmx_set_decode_extra_options(spmx_handle *h, const char *options);
/* SetVocabulary / ResetVocabulary (src/sentencepiece_processor.h:279-283). */
int spmx_set_vocabulary(spmx_handle *h, const char *const *pieces, const uint64_t *piece_lens, uint64_t n);
int spmx_reset_vocabulary(spmx_handle *h);

/* ---- vocabulary accessors (src/sentencepiece_processor.h:638-677) -------- */
int spmx_piece_size(const spmx_handle *h);                                  /* GetPieceSize */
int spmx_piece_to_id(const spmx_handle *h, const char *piece, uint64_t len);/* PieceToId */
/* IdToPiece: copies up to cap bytes, returns the piece length (or -1). */
int64_t spmx_id_to_piece(const spmx_handle *h, int id, char *out, uint64_t cap);
int spmx_unk_id(const spmx_handle *h);
/* SentencePiece::Type of a piece (src/sentencepiece_model.proto:296-303): 1 NORMAL, 2 UNKNOWN, 3 CONTROL,
 * 4 USER_DEFINED, 5 UNUSED, 6 BYTE -- IsUnknown / IsControl / IsUnused / IsByte (sentencepiece_processor.h:653-662);
 * -1 for an id out of range. */
int spmx_piece_type(const spmx_handle *h, int id);
int spmx_bos_id(const spmx_handle *h);
int spmx_eos_id(const spmx_handle *h);
int spmx_pad_id(const spmx_handle *h);
int spmx_model_type(const spmx_handle *h);   /* 1 unigram, 2 bpe */
/* Diagnostic: the kNf* bits the table compiler derived for this model (csrc/dev.h): which normalizer switches are on,
 * whether the one-byte space symbol / the word-wise BPE form apply. */
uint32_t spmx_model_flags(const spmx_handle *h);
/* trainer_spec.unk_piece (src/sentencepiece_model.proto:220), the string the `unk_piece` extra option writes
 * (src/sentencepiece_processor.cc:1050-1058): copies up to cap bytes, returns its length. */
int64_t spmx_unk_piece(const spmx_handle *h, char *out, uint64_t cap);

/* ---- encode -------------------------------------------------------------
 * All three are element-wise identical to calling
 *   SentencePieceProcessor::Encode(input, std::vector<int>* ids)
 *     (src/sentencepiece_processor.h:299-300, .cc:392-403)
 * per sentence, which is what the reference's only batch form --
 *   _EncodeAsIdsBatch (python/src/sentencepiece/sentencepiece.i:439-446) --
 * computes with a thread pool.
 *
 * No length limits: a sentence of any size is encoded (the fast kernels take what fits their length classes; what
 * does not -- documents, words of thousands of characters, BPE models whose pieces span words -- runs in kernels
 * whose working set lives in HBM, sized from the sentence).  The only bound is the reference's own int arithmetic:
 * the NORMALIZED form of one sentence must stay below 2^31 bytes.
 *
 * Per-sentence status: a sentence the reference's Encode would fail (kInternal "all normalized characters are not
 * consumed", e.g. a CONTROL piece among BPE symbols), or one beyond the bound above (OUT_OF_RANGE), yields NO ids;
 * the other sentences of the batch are unaffected -- what the reference's batch form does, whose workers call
 * EncodeAsIds and drop the Status (python/src/sentencepiece/sentencepiece.i:249-265).  The batch calls return OK;
 * the _ex forms report a util::StatusCode byte per sentence (0 = OK) and the number of failed sentences.
 *
 * Sentences are passed packed: `text` holds the bytes of all sentences back to
 * back, offsets[i] .. offsets[i+1] delimit sentence i (n + 1 entries).
 * Ids come back as CSR: ids[id_offsets[i] .. id_offsets[i+1]). */

/* Device-resident form: every pointer is HIP device memory on the handle's
 * GPU; `stream` is a hipStream_t (NULL = default stream).  The call enqueues
 * its kernels on `stream`, waits for them, and returns the number of ids in
 * *total_ids.  If ids_capacity is too small, returns RESOURCE_EXHAUSTED (8)
 * with the required capacity in *total_ids (id_offsets is still valid).
 * d_text may have any alignment; the kernels read it in ALIGNED 16-byte units, so the bytes that share a unit with the
 * text's first or last byte may be read (never interpreted) -- a unit does not cross a page, and a device allocation
 * covers whole pages. */
int spmx_encode_batch_device(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                             uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets, void *stream,
                             uint64_t *total_ids);

/* As above, plus d_status (nullable): n status bytes in device memory; *n_failed (nullable): sentences with a
 * non-zero status. */
int spmx_encode_batch_device_ex(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                                uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets,
                                uint8_t *d_status, void *stream, uint64_t *total_ids, uint64_t *n_failed);

/* Host-buffer form: copies text to the GPU, encodes, copies ids back.  Big batches (from 2^19 sentences) run as a
 * chunk pipeline on several host threads: staging copy, H2D, kernels and D2H of different chunks overlap.
 * *ids (total ids) and *id_offsets (n + 1) are allocated by the library and
 * released with spmx_free(). */
int spmx_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                      uint64_t **id_offsets);
/* As above, plus *status (nullable): n status bytes, released with spmx_free(); *n_failed (nullable). */
int spmx_encode_batch_ex(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                         uint64_t **id_offsets, uint8_t **status, uint64_t *n_failed);
/* The same batch given as n (pointer, length) pairs -- the layout of a std::vector<absl::string_view>'s elements as
 * this struct spells them: no packed copy is made, the sentences are gathered into the staging buffers chunk by chunk. */
typedef struct spmx_view { const char *data; uint64_t len; } spmx_view;
int spmx_encode_batch_views(spmx_handle *h, const spmx_view *views, uint64_t n, int32_t **ids, uint64_t **id_offsets,
                            uint8_t **status, uint64_t *n_failed);
/* One batch over several GPUs of the node from ONE process: handles[g] = the same model on GPU g.  The chunks of the
 * batch go round-robin over the GPUs (several in flight on each) and all land in ONE output CSR; no collective (the
 * multi-process form with an RCCL all-gather of the ids is sentencepiece_amd/sharding.py). */
int spmx_encode_batch_multi(spmx_handle *const *handles, int n_handles, const char *text, const uint64_t *offsets, uint64_t n,
                            int32_t **ids, uint64_t **id_offsets, uint8_t **status, uint64_t *n_failed);
void spmx_free(void *p);

/* ---- multi-GPU, one process per GPU: the ids of every rank on every rank, over RCCL ----------------------------------
 * The reference has no multi-device form; BASELINE.json's north_star asks for "a RCCL all-gatherv of the token-id output
 * over xGMI" behind the batch semantics of python/src/sentencepiece/sentencepiece.i:245-267 (the job's sentences in order,
 * each with its own ids).  Every rank encodes its contiguous shard of the job's sentences with
 * spmx_encode_batch_device and then calls spmx_all_gather_ids with its CSR; every rank gets the job's CSR:
 *   d_all_ids[rank_ids[r] ...)            rank r's ids (rank order = sentence order),
 *   d_all_id_offsets[0 .. total + 1)      offsets into d_all_ids, rebased to the whole job.
 * nccl_comm: the caller's ncclComm_t (one rank per GPU), or one made by spmx_rccl_comm_init.  librccl is looked up at the
 * first call (SPMX_RCCL_LIB names another library), it is no link-time dependency of libspmx.so.  d_scratch:
 * spmx_gather_scratch_words(world) uint64 of device memory (today 5 * (1 + world); ask, do not hard-code).  rank_sentences / rank_ids (host, world + 1 entries each, nullable): prefix sums over the
 * ranks.  Stream-ordered except for one read-back of the counts; collective: every rank of the communicator calls it.
 * Returns 0, or a util::StatusCode number (8: a capacity is too small -- the message names what is needed; 14: RCCL is
 * not loadable) with the text in spmx_gather_last_error().  Whether the gathered CSR fits is decided from EVERY rank's
 * capacities (they travel with the counts): when one rank's buffers are too small, all ranks return 8 and none starts a
 * transfer -- the ranks may size their buffers independently and retry together. */
int spmx_all_gather_ids(void *nccl_comm, int rank, int world, const int32_t *d_ids, uint64_t n_ids,
                        const uint64_t *d_id_offsets, uint64_t n_sentences, int32_t *d_all_ids, uint64_t all_ids_capacity,
                        uint64_t *d_all_id_offsets, uint64_t all_offsets_capacity, uint64_t *d_scratch,
                        uint64_t *rank_sentences, uint64_t *rank_ids, void *stream);
/* uint64 words of device scratch spmx_all_gather_ids needs for `world` ranks (0 for a world outside 1 .. 64). */
uint64_t spmx_gather_scratch_words(int world);
/* A communicator without linking RCCL oneself: rank 0 makes the 128-byte id and hands it to the other ranks by whatever
 * means the job has (a file, MPI, a socket); every rank then calls spmx_rccl_comm_init with its GPU current. */
int spmx_rccl_unique_id(void *id128);
int spmx_rccl_comm_init(void **nccl_comm, int world, int rank, const void *id128);
int spmx_rccl_comm_destroy(void *nccl_comm);
const char *spmx_gather_last_error(void);

/* Single sentence, caller-provided buffer (Encode(input, &ids)): returns the sentence's own Status, as the
 * reference does.  RESOURCE_EXHAUSTED with the needed size in *n_ids if cap is too small. */
int spmx_encode(spmx_handle *h, const char *text, uint64_t len, int32_t *ids, uint64_t cap, uint64_t *n_ids);

/* ---- decode -------------------------------------------------------------
 * Element-wise identical to SentencePieceProcessor::Decode(const std::vector<int>& ids, std::string*)
 *   (src/sentencepiece_processor.h:311-312, .cc:761-925): control pieces vanish, the unknown piece becomes
 *   trainer_spec.unk_surface, byte pieces are reassembled into UTF-8 (a structurally invalid byte -> U+FFFD),
 *   U+2581 -> ' ', leading whitespace handled as the normalizer_spec asks.  An id outside [0, GetPieceSize())
 *   fails the call with OUT_OF_RANGE (11) "Invalid id: N"; a model with a denormalizer_spec has its rules applied to
 *   the decoded text (.cc:905-907).
 * Ids are CSR (as produced by the encode calls); text comes back packed with text_offsets (n + 1 entries). */

/* Device-resident form: every pointer is HIP device memory.  If text_capacity is too small (or d_text is NULL)
 * returns RESOURCE_EXHAUSTED (8) with the required capacity in *total_bytes (d_text_offsets is valid). */
int spmx_decode_batch_device(spmx_handle *h, const int32_t *d_ids, const uint64_t *d_id_offsets, uint64_t n, void *d_text,
                             uint64_t text_capacity, uint64_t *d_text_offsets, void *stream, uint64_t *total_bytes);
/* Decode from PIECES: Decode(const std::vector<std::string>& pieces, ...) (src/sentencepiece_processor.h:303-309,
 * .cc:761-769; python/src/sentencepiece/sentencepiece.i:547 _DecodePiecesBatch).  The caller maps every piece to its id
 * (spmx_piece_to_id); a piece that is NOT in the vocabulary -- PieceToId gives the unknown id for a string other than the
 * unknown piece's -- goes through as it is (.cc:784-790): it travels as the id -(k + 1) with its bytes
 * lit_bytes[lit_offsets[k], lit_offsets[k + 1]) (k < n_lit; at most 65535 bytes each).  Under a decode extra option `unk`
 * (spmx_decode_unk_option() != 0) the reference rewrites such a piece to the unknown piece first (.cc:1050-1058): the
 * caller then passes the unknown id instead.  Everything else as spmx_decode_batch. */
int spmx_decode_batch_pieces(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, const char *lit_bytes,
                             const uint64_t *lit_offsets, uint64_t n_lit, char **text, uint64_t **text_offsets);
int spmx_decode_unk_option(const spmx_handle *h);
/* GetScore(id) (src/sentencepiece_processor.h:650): the piece's score as the ModelProto holds it; 11 for an id out of range. */
int spmx_piece_score(const spmx_handle *h, int id, float *score);
/* serialized_model_proto() (src/sentencepiece_processor.h:694): the bytes the handle was created from; owned by the handle. */
int spmx_serialized_model(const spmx_handle *h, const char **data, uint64_t *n_bytes);
/* Host-buffer form; *text (total bytes) and *text_offsets (n + 1) are released with spmx_free(). */
int spmx_decode_batch(spmx_handle *h, const int32_t *ids, const uint64_t *id_offsets, uint64_t n, char **text,
                      uint64_t **text_offsets);
/* Single sentence, caller-provided buffer (Decode(ids, &text)); RESOURCE_EXHAUSTED with the needed size in *len. */
int spmx_decode(spmx_handle *h, const int32_t *ids, uint64_t n_ids, char *out, uint64_t cap, uint64_t *len);

/* ---- spans form ---------------------------------------------------------
 * The ids plus, for every id, the byte range [begin, end) of its sentence that it covers: pieces(i).begin() /
 * .end() of the SentencePieceText that Encode(absl::string_view, SentencePieceText *) fills
 * (src/sentencepiece_processor.cc:638-651, PopulateSentencePieceText :547-636, bos / eos spans :1029-1048).
 * Offsets are relative to the start of the sentence, in bytes (the C++ convention; the Python wrapper converts to
 * characters, sentencepiece.i ConvertToUnicodeSpans).  surface = input[begin, end); the piece of a known id is
 * IdToPiece(id).  Offsets are 32-bit: sentences below 4 GiB.
 * nbegin / nend (optional, both or neither): the same tokens as byte ranges of the NORMALIZED sentence
 * (spmx_normalize_batch): the piece of an unknown token is that text (:614-617); 0, 0 for a bos / eos.
 * d_begin / d_end / d_nbegin / d_nend: ids_capacity entries each; the arrays of the host form are released with
 * spmx_free(). */
int spmx_encode_batch_spans_device(spmx_handle *h, const void *d_text, uint64_t text_bytes, const uint64_t *d_offsets,
                                   uint64_t n, int32_t *d_ids, uint64_t ids_capacity, uint64_t *d_id_offsets,
                                   uint32_t *d_begin, uint32_t *d_end, uint32_t *d_nbegin, uint32_t *d_nend, void *stream,
                                   uint64_t *total_ids);
int spmx_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                            uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend);
/* As above, plus *status (nullable): n status bytes, released with spmx_free(); *n_failed (nullable): what
 * Encode(input, SentencePieceText *) returns per sentence (src/sentencepiece_processor.cc:638-651 -> :628). */
int spmx_encode_batch_spans_ex(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                               uint64_t **id_offsets, uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend,
                               uint8_t **status, uint64_t *n_failed);
/* The message of a per-sentence status byte (the _ex forms): for INTERNAL (13) the reference's "all normalized
 * characters are not consumed." (src/sentencepiece_processor.cc:628); "" for 0. */
const char *spmx_status_message(int code);

/* ---- batch Normalize ----------------------------------------------------
 * SentencePieceProcessor::Normalize(input, &normalized, &norm_to_orig) (src/sentencepiece_processor.cc:933-945 ->
 * Normalizer::Normalize, src/normalizer.cc:71-186) per sentence: the packed normalized text + n + 1 offsets and,
 * optionally, the alignment vectors: sentence s owns entries [norm_offsets[s] + s, norm_offsets[s + 1] + s + 1) of
 * norm_to_orig -- one per normalized byte plus the closing one, which is 0xFFFFFFFF where the reference's vector
 * is empty (empty or all-whitespace input).  No length limit (OUT_OF_RANGE only where the normalized form of a
 * sentence would exceed 2^31 bytes).
 * Device form: d_norm_to_orig (nullable) holds norm_capacity + n + 1 entries. */
int spmx_normalize_batch_device(spmx_handle *h, const void *d_text, const uint64_t *d_offsets, uint64_t n, void *d_norm,
                                uint64_t norm_capacity, uint64_t *d_norm_offsets, uint32_t *d_norm_to_orig, void *stream,
                                uint64_t *total_bytes);
int spmx_normalize_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, char **norm,
                         uint64_t **norm_offsets, uint32_t **norm_to_orig);

/* ---- n-best -------------------------------------------------------------
 * NBestEncode(input, nbest_size, std::vector<std::vector<int>>*) (src/sentencepiece_processor.h:323-324;
 * unigram::Model::NBestEncode src/unigram_model.cc:686-717, Lattice::NBest :345-515) per sentence, unigram models
 * only (INTERNAL otherwise, as the reference).  nbest_size is clamped to [1, 1024]; 1 is the plain encoder with
 * score 0.  Result r of the batch: ids[id_offsets[r], id_offsets[r + 1]) and scores[r]; sentence s owns the
 * results [result_offsets[s], result_offsets[s + 1]) (n + 1 entries), best first.  The four arrays are released
 * with spmx_free().  No length limit: a sentence beyond the first launch's lattice capacities (1024 normalized bytes,
 * 16384 nodes) runs in a second launch sized for it; RESOURCE_EXHAUSTED only where the device memory for one
 * sentence's lattice and agenda runs out. */
int spmx_nbest_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                            int32_t **ids, uint64_t **id_offsets, float **scores, uint64_t **result_offsets);
/* NBestEncode(input, nbest_size, NBestSentencePieceText *) / (…, std::vector<std::vector<std::string>> *)
 * (src/sentencepiece_processor.h:318-324, .cc:653-676): the same results, plus for every id of every result the byte
 * range of the input (begin / end) and of the normalized text (nbegin / nend) its piece covers -- the four arrays of
 * the spans form above, indexed like ids; released with spmx_free().  A run of unknown characters is one piece; of a
 * character's byte-fallback pieces the last carries the range (sentencepiece_processor.cc:581-617). */
int spmx_nbest_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                                  int32_t **ids, uint64_t **id_offsets, float **scores, uint64_t **result_offsets,
                                  uint32_t **begin, uint32_t **end, uint32_t **nbegin, uint32_t **nend);

/* ---- sampling and the original encoder -----------------------------------
 * SampleEncode(input, nbest_size, alpha, std::vector<int>*) (src/sentencepiece_processor.h:346-353, .cc:678-720) per
 * sentence, the subword-regularization entry:
 *   unigram models (IsNBestEncodeAvailable):
 *     nbest_size < 0      one segmentation drawn from the lattice, Lattice::Sample(alpha) (forward filtering /
 *                         backward sampling, src/unigram_model.cc:511-542)
 *     nbest_size 0 or 1   the plain encoder
 *     nbest_size > 1      one of the nbest_size best, drawn with probability ~ exp(alpha * score) (:700-716)
 *   BPE models: EVERY nbest_size is BPE-dropout with merge-skip probability alpha (.cc:688-693 sends the call to
 *     bpe::Model::SampleEncode, src/bpe_model.cc:131-156); alpha <= 0 is the plain encoder
 *   nbest_size > 512    INTERNAL "nbest_size must be nbest_size <= 512" (.cc:684)
 * Draws come from generators keyed by (seed, sentence index) -- reproducible per call; the reference's thread-local
 * mt19937 stream is not (and cannot be) reproduced, its own tests pin the DISTRIBUTION (unigram_model_test.cc:429-470,
 * bpe_model_test.cc:252-295), as tests/test_sampling.py does.  alpha = 0 under BPE is bit-equal to Encode. */
int spmx_sample_encode_batch(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                             float alpha, uint64_t seed, int32_t **ids, uint64_t **id_offsets);
/* SampleEncode(input, nbest_size, alpha, SentencePieceText *) / (…, std::vector<std::string> *)
 * (src/sentencepiece_processor.h:346-353, :404-408): the drawn segmentation with the four span arrays (as above). */
int spmx_sample_encode_batch_spans(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int nbest_size,
                                   float alpha, uint64_t seed, int32_t **ids, uint64_t **id_offsets, uint32_t **begin,
                                   uint32_t **end, uint32_t **nbegin, uint32_t **nend);
/* The reference's ORIGINAL unigram encoder (EncoderVersion::kOriginal, src/unigram_model.cc:674-692): the lattice of
 * Lattice::SetSentence / Model::PopulateNodes and Lattice::Viterbi (:161-198, all-float, first best left node wins)
 * instead of EncodeOptimized.  Same ids except where float and double arithmetic break a tie differently. */
int spmx_encode_batch_original(spmx_handle *h, const char *text, const uint64_t *offsets, uint64_t n, int32_t **ids,
                               uint64_t **id_offsets);

/* ---- corpus packer ------------------------------------------------------
 * The caller-side step of the reference's spm_encode (src/spm_encode_main.cc:159-165: std::getline over the input
 * file, one Encode per line) on the device: a file image with '\n'-terminated lines -> the packed text (without
 * the terminators; '\r' is kept, as getline keeps it) + n_lines + 1 offsets that the encode calls take.  A last line
 * without '\n' counts; "a\n" is one line.  d_file must be 16-byte aligned device memory.  Too small a capacity ->
 * RESOURCE_EXHAUSTED (8) with the needed sizes in *n_lines (+ 1 offsets) and *text_bytes. */
int spmx_split_lines_device(spmx_handle *h, const void *d_file, uint64_t bytes, void *d_text, uint64_t text_capacity,
                            uint64_t *d_offsets, uint64_t offsets_capacity, void *stream, uint64_t *n_lines,
                            uint64_t *text_bytes);

/* ---- corpus file -> ids --------------------------------------------------
 * What the reference's `spm_encode --output_format=id < in > out` does (src/spm_encode_main.cc:115-119, :159-165: getline,
 * Encode, StrJoin(ids, " ")), as one call: the file is mmap'ed and goes through a pipeline of worker threads -- pinned
 * staging, H2D, the device line splitter, the encode kernels, D2H, formatting -- chunk by chunk, in order.
 * format "id": the reference's text output, byte for byte; "bin": out_path receives the ids (int32, flat) and
 * out_path + ".idx" the n + 1 uint64 offsets. */
int spmx_encode_file(spmx_handle *h, const char *in_path, const char *out_path, const char *format, uint64_t *n_sentences,
                     uint64_t *n_ids);

/* ---- measurement --------------------------------------------------------
 * Per-kernel timing of the encode kernels of the LAST profiled encode call on the handle, measured with hipEvents
 * on the call's stream (enable first).  Arrays hold 7 entries ("kernel slots": 0 the streaming launch over the
 * classes up to 16 KiB, 1 the streaming launch over the document classes, 2 the overflow launch, 3 the
 * sentence-per-wave BPE launches, 4 the long form (the wave-cooperative unigram form included), 5 the first round of the
 * word form, 6 its second round; unused slots are zero); returns the number of slots (callers size for 8).
 * spmx_last_profile_name() gives the kernel symbol of a slot as rocprofv3 prints it.  bytes[] is the algorithmic
 * byte count SURVEY.md section 8d defines (raw bytes + 8 + 4 * ids + 8 per sentence).  path[4]: sentences the main
 * tiles set aside on hard lists, sentences on the overflow list, sentences that took the long form, failed sentences. */
int spmx_set_profiling(spmx_handle *h, int enabled);
int spmx_last_profile_name(const spmx_handle *h, int slot, char *out, uint64_t cap);
int spmx_last_profile(const spmx_handle *h, float *kernel_ms, uint64_t *sentences, uint64_t *raw_bytes,
                      uint64_t *ids, uint64_t *bytes, uint64_t *path, float *total_ms);

/* Shader-clock cycles the waves of the LAST profiled call spent per phase, summed over waves:
 * cycles[5 * slot + {0 load, 1 normalize, 2 segment, 3 emit}] (35 entries; callers size for 40); entry 4 is the number of search-loop
 * iterations the waves of the lane-per-sentence forms executed. */
int spmx_last_phase_cycles(const spmx_handle *h, uint64_t *cycles);

/* What loading this handle cost: the device bytes of its tables (normalizer tries, piece trie, word memo, decode tables --
 * not the per-call workspaces, which grow with the batches a handle sees and are kept: up to SPMX_STREAM_SCRATCH_MB, default
 * 16 GB, for the streaming kernels' text columns) and the wall-clock milliseconds of spmx_create: parse + table build + upload.
 * No reference counterpart (the reference's Load builds its tries on the host, sentencepiece_processor.cc:231-275); either
 * pointer may be null. */
int spmx_handle_info(const spmx_handle *h, uint64_t *table_bytes, double *load_ms);

#ifdef __cplusplus
}
#endif
#endif
