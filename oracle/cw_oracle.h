/*
 * cw_oracle.h -- CPU oracle for the hash+compress hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the arithmetic
 * the reference's `hashandcompress` path performs on the CPU.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported baseline -- never as the product path.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   Skein-512 / Skein-256 : pinned by the reference's own NIST KAT files
 *                           (reference_code/skein/KAT_MCT/{Short,Long}MsgKAT_{256,512}.txt,
 *                           MonteCarlo_512.txt) and by oracle/_ref (the reference's
 *                           Optimized_64bit C sources compiled where they lie).
 *   SHA-256               : pinned by FIPS 180-4 known answers + OpenSSL (hashlib),
 *                           the library the reference calls (src/hashing_perf/hash.cpp:35).
 *   LZ4 1.8.2 / liblzf    : third-party code that ships in the reference only as
 *                           prebuilt archives (never linked or run here).  Pinned by the
 *                           reference outputs SURVEY.md 8(c) recorded (sizes, SHA-256 of
 *                           the compressed bytes, corpus ratios); otherwise "parity
 *                           unpinned by the reference itself".
 */
#ifndef CW_ORACLE_H
#define CW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Skein (2008 NIST round-1 submission, "v1.1" constants) ------------- */
/* follows reference_code/skein/Reference_Implementation/skein.c + skein_block.c */
int cw_oracle_skein512(const uint8_t *msg, size_t msg_bits, unsigned hash_bits, uint8_t *out);
int cw_oracle_skein256(const uint8_t *msg, size_t msg_bits, unsigned hash_bits, uint8_t *out);
/* initial chaining value for (state words, hash_bits): the config-block UBI */
void cw_oracle_skein_iv(int state_words, unsigned hash_bits, uint64_t *iv);
/* tree hashing, follows reference_code/skein/Additional_Implementations/skein_test.c:616-680 (state_words 4 or 8) */
int cw_oracle_skein_tree(int state_words, const uint8_t *msg, size_t len, unsigned hash_bits, unsigned leaf, unsigned node,
                         unsigned max_level, uint8_t *out);

/* debug aid (the reference's SKEIN_DEBUG callouts, skein.h:246-254): one Threefish call + feed-forward with the state after
 * the initial key injection, after each of the 72 rounds and after each of the 18 later key injections, in execution
 * order -- (1 + 72 + 18) * state_words words; checked against KAT_MCT/skein_golden_kat_short_internals.txt */
void cw_oracle_threefish_trace(int state_words, uint64_t *key, const uint64_t *tweak, const uint8_t *block, uint64_t *trace);

/* ---- SHA-256 (FIPS 180-4) ---------------------------------------------- */
void cw_oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);

/* ---- LZ4 block, "fast" parser as LZ4_compress_default(src,dst,n,cap) of v1.8.2 */
/* returns compressed size, 0 if it does not fit */
size_t cw_oracle_lz4_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
/* decoder (LZ4_decompress_safe semantics); returns decoded size or -1 */
long   cw_oracle_lz4_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
size_t cw_oracle_lz4_bound(size_t n);

/* ---- LZF, liblzf 3.x compressor, HLOG 16 / VERY_FAST / zeroed table ------ */
size_t cw_oracle_lzf_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
long   cw_oracle_lzf_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);

/* ---- synthetic block generator shared by bench/tests -------- */
/* word w (u64, little-endian) of block b = splitmix64(seed ^ (b<<13 | w))  */
void cw_oracle_gen_random_blocks(uint64_t seed, uint64_t first_block, size_t nblocks,
                                 size_t block_bytes, uint8_t *dst);

/* compressible mix: even blocks as above, odd blocks = 64-byte motif with 1/16 of the bytes mutated */
void cw_oracle_gen_mixed_blocks(uint64_t seed, uint64_t first_block, size_t nblocks,
                                size_t block_bytes, uint8_t *dst);

/* ---- batched per-block drivers (the CPU "hashandcompress" worker loop) --- */
enum { CW_OR_HASH_SKEIN512 = 0, CW_OR_HASH_SKEIN256_128 = 1, CW_OR_HASH_SHA256 = 2, CW_OR_HASH_NONE = 3 };
enum { CW_OR_COMP_LZ4 = 0, CW_OR_COMP_LZF = 1, CW_OR_COMP_NONE = 2 };
size_t cw_oracle_digest_bytes(int hash_alg);
/*
 * Process nblocks blocks of block_bytes at src with `threads` pthreads that pull
 * block indices from a shared counter (the reference's PopAndProcessBlocks,
 * src/hashandcompress/HashAndCompress.cpp:263-272): compress each block into
 * dst + i*dst_stride (sizes[i] = bytes, 0 = did not fit), then hash it into
 * digests + i*digest_bytes.  Any of digests/dst/sizes may be NULL.
 * Returns elapsed seconds of the worker phase (the reference's timed window).
 */
double cw_oracle_hash_and_compress(const uint8_t *src, size_t nblocks, size_t block_bytes,
                                   int hash_alg, int comp_alg, int threads,
                                   uint8_t *digests, uint8_t *dst, size_t dst_stride,
                                   uint32_t *sizes);

#ifdef __cplusplus
}
#endif
#endif
