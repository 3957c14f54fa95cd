/*
 * hc_oracle.c -- TEST INFRASTRUCTURE: the CPU "hashandcompress" worker phase and the
 * synthetic block generator, over the oracle's own arithmetic.
 *
 * Mirrors the reference's timed window (src/hashandcompress/HashAndCompress.cpp:391-406):
 * all input resident in memory, N worker threads each pulling the next unit of work
 * (PopAndProcessBlocks :263-272) and, per block, compressing then hashing it
 * (ProcessBlock :231-261 -- without its two data bugs).
 * Used by tests as the checker and by bench.py only for the reported cpu_baseline.
 */
#include "cw_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void cw_oracle_gen_random_blocks(uint64_t seed, uint64_t first_block, size_t nblocks,
                                 size_t block_bytes, uint8_t *dst)
{
    const size_t words = block_bytes / 8;
    for (size_t b = 0; b < nblocks; b++) {
        uint8_t *p = dst + b * block_bytes;
        const uint64_t blk = first_block + b;
        for (size_t w = 0; w < words; w++) {
            uint64_t v = splitmix64(seed ^ ((blk << 13) | (uint64_t)w));
            for (int k = 0; k < 8; k++) p[8 * w + k] = (uint8_t)(v >> (8 * k));
        }
    }
}

/* host twin of gen_mixed_kernel (compute_war_amd/csrc/misc_kernels.hip): SURVEY.md 8(d)'s compressible mix */
static uint64_t mixed_word(uint64_t seed, uint64_t blk, uint64_t w)
{
    const uint64_t salt_motif = 0x6D6F746966ULL, salt_mutate = 0x6D7574617465ULL;
    uint64_t m, r, r2, mask = 0;
    if ((blk & 1) == 0) return splitmix64(seed ^ ((blk << 13) | w));
    m = splitmix64(seed ^ salt_motif ^ ((blk << 13) | (w & 7)));
    r = splitmix64(seed ^ salt_mutate ^ ((blk << 13) | w));
    r2 = splitmix64(r);
    for (int k = 0; k < 8; k++) if (((r >> (4 * k)) & 15) == 0) mask |= 0xFFULL << (8 * k);
    return (m & ~mask) | (r2 & mask);
}

void cw_oracle_gen_mixed_blocks(uint64_t seed, uint64_t first_block, size_t nblocks,
                                size_t block_bytes, uint8_t *dst)
{
    const size_t words = block_bytes / 8;
    for (size_t b = 0; b < nblocks; b++) {
        uint8_t *p = dst + b * block_bytes;
        for (size_t w = 0; w < words; w++) {
            uint64_t v = mixed_word(seed, first_block + b, (uint64_t)w);
            for (int k = 0; k < 8; k++) p[8 * w + k] = (uint8_t)(v >> (8 * k));
        }
    }
}

size_t cw_oracle_digest_bytes(int hash_alg)
{
    switch (hash_alg) {
    case CW_OR_HASH_SKEIN512: return 64;
    case CW_OR_HASH_SKEIN256_128: return 16;
    case CW_OR_HASH_SHA256: return 32;
    default: return 0;
    }
}

typedef struct {
    const uint8_t *src;
    size_t nblocks, block_bytes;
    int hash_alg, comp_alg;
    uint8_t *digests, *dst;
    size_t dst_stride;
    uint32_t *sizes;
    size_t next;              /* shared work counter */
    pthread_mutex_t lock;
} job_t;

static void process_one(job_t *j, size_t i, uint8_t *scratch, size_t scratch_cap)
{
    const uint8_t *blk = j->src + i * j->block_bytes;
    const size_t n = j->block_bytes;
    if (j->comp_alg != CW_OR_COMP_NONE) {
        uint8_t *out = j->dst ? j->dst + i * j->dst_stride : scratch;
        size_t cap = j->dst ? j->dst_stride : scratch_cap, c;
        if (j->comp_alg == CW_OR_COMP_LZ4) {
            c = cw_oracle_lz4_compress(blk, n, out, cap);   /* reference passes 2*n (:353) */
        } else {
            size_t lim = n - 1 < cap ? n - 1 : cap;         /* reference passes n-1 (:346) */
            c = cw_oracle_lzf_compress(blk, n, out, lim);
        }
        if (j->sizes) j->sizes[i] = (uint32_t)c;
    }
    if (j->hash_alg != CW_OR_HASH_NONE) {
        uint8_t tmp[64];
        const size_t db = cw_oracle_digest_bytes(j->hash_alg);
        uint8_t *d = j->digests ? j->digests + i * db : tmp;
        switch (j->hash_alg) {
        case CW_OR_HASH_SKEIN512: cw_oracle_skein512(blk, n * 8, 512, d); break;
        case CW_OR_HASH_SKEIN256_128: cw_oracle_skein256(blk, n * 8, 128, d); break;
        default: cw_oracle_sha256(blk, n, d); break;
        }
    }
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const size_t cap = 2 * j->block_bytes + 64;
    uint8_t *scratch = (uint8_t *)malloc(cap);
    for (;;) {
        size_t i;
        pthread_mutex_lock(&j->lock);
        i = j->next++;
        pthread_mutex_unlock(&j->lock);
        if (i >= j->nblocks) break;
        process_one(j, i, scratch, cap);
    }
    free(scratch);
    return NULL;
}

double cw_oracle_hash_and_compress(const uint8_t *src, size_t nblocks, size_t block_bytes,
                                   int hash_alg, int comp_alg, int threads,
                                   uint8_t *digests, uint8_t *dst, size_t dst_stride,
                                   uint32_t *sizes)
{
    job_t j;
    struct timespec t0, t1;
    pthread_t *tid;
    if (threads < 1) threads = 1;
    memset(&j, 0, sizeof j);
    j.src = src; j.nblocks = nblocks; j.block_bytes = block_bytes;
    j.hash_alg = hash_alg; j.comp_alg = comp_alg;
    j.digests = digests; j.dst = dst; j.dst_stride = dst_stride; j.sizes = sizes;
    pthread_mutex_init(&j.lock, NULL);
    tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);

    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) pthread_create(&tid[t], NULL, worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);

    free(tid);
    pthread_mutex_destroy(&j.lock);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
