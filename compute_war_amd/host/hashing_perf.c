/*
 * hashing_perf.c -- the reference's per-block hash timing harness (src/hashing_perf/test.cpp:7-92,
 * hash.cpp:5-77) re-stated in C over the C ABI of libcwhc.so, same log format so the reference's notebook
 * (notebooks/hash-perf.ipynb cells 2-4,15) can plot the device numbers:
 *
 *   RunHashingSB  (test.cpp:7-29)   per 4 KiB block:  file|index|Skein256|us|   then   file|index|Sha256|us|
 *                                   (blockIndex is post-incremented after each line, as in the reference)
 *   RunHashingMB  (test.cpp:31-66)  for window = 1..64: per window of N blocks copied contiguous,
 *                                   file|windowIndex|Sha256MB|us|window|          (hash.cpp:48-77)
 *
 * Each timed region is one synchronous call through the slot-compatible entry points (H2D + kernel + D2H), i.e.
 * the latency a caller of the reference's functions would see; blocks are whole 4 KiB blocks, a partial tail is
 * dropped (file.cpp:18-60).  Usage: hashing_perf [--verify] <data-dir>   (test.cpp:68-73)
 *
 * --verify (not in the reference, whose harness never looks at a digest: test.cpp:7-29): after the log, one line per algorithm
 *   verify|<alg>|<digests>|<xor-fold of all digests as 64-bit words>
 * the fold hashandcompress -v prints (Skein256: the single-block calls; Sha256: the single-block calls; Sha256MB: every window
 * of every window size), so the harness's results can be checked against the oracle, not only its format.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "../../include/cw_hashcompress.h"

#define LOG_SEPARATOR "|"
enum { kBlockSize = 4096 }; /* shared.h:25 */

static int verify = 0;
static uint64_t fold_sk = 0, fold_sha = 0, fold_mb = 0, n_sk = 0, n_sha = 0, n_mb = 0;
static void fold_digests(const uint8_t *d, size_t bytes, uint64_t *acc)
{
    for (size_t i = 0; i + 8 <= bytes; i += 8) {
        uint64_t v;
        memcpy(&v, d + i, 8);
        *acc ^= v;
    }
}

static uint64_t now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (uint64_t)t.tv_sec * 1000000u + (uint64_t)t.tv_nsec / 1000u;
}

static uint8_t *read_blocks(const char *path, size_t *nblocks)
{
    FILE *f = fopen(path, "rb");
    *nblocks = 0;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    size_t n = sz > 0 ? (size_t)sz / kBlockSize : 0;
    uint8_t *buf = (uint8_t *)malloc(n ? n * kBlockSize : 1);
    if (n && fread(buf, kBlockSize, n, f) != n) n = 0;
    fclose(f);
    *nblocks = n;
    return buf;
}

static void run_sb(const char *file, const uint8_t *blocks, size_t n)
{
    uint8_t digest[64];
    uint64_t index = 0;
    for (size_t i = 0; i < n; i++) {
        const char *b = (const char *)blocks + i * kBlockSize;
        uint64_t t0 = now_us();
        cw_hash_skein(b, (char *)digest, 1); /* HashBlockSkein256 (hash.cpp:5-26) */
        uint64_t t1 = now_us();
        printf("%s" LOG_SEPARATOR "%llu" LOG_SEPARATOR "Skein256" LOG_SEPARATOR "%llu" LOG_SEPARATOR "\n", file,
               (unsigned long long)index++, (unsigned long long)(t1 - t0));
        if (verify) { fold_digests(digest, 16, &fold_sk); n_sk++; }
        t0 = now_us();
        cw_hash_sha256mb(b, (char *)digest, 1); /* HashBlockSHA256 (hash.cpp:28-46) */
        t1 = now_us();
        printf("%s" LOG_SEPARATOR "%llu" LOG_SEPARATOR "Sha256" LOG_SEPARATOR "%llu" LOG_SEPARATOR "\n", file,
               (unsigned long long)index++, (unsigned long long)(t1 - t0));
        if (verify) { fold_digests(digest, 32, &fold_sha); n_sha++; }
    }
}

static void run_mb(const char *file, const uint8_t *blocks, size_t n, size_t window)
{
    uint8_t *digests = (uint8_t *)malloc(32 * window);
    const size_t windows = n / window;
    for (size_t w = 0; w < windows; w++) {
        const char *p = (const char *)blocks + w * window * kBlockSize; /* already contiguous (test.cpp:45-53) */
        uint64_t t0 = now_us();
        cw_hash_sha256mb(p, (char *)digests, (int)window); /* HashBlockSHA256MB (hash.cpp:48-77) */
        uint64_t t1 = now_us();
        printf("%s" LOG_SEPARATOR "%llu" LOG_SEPARATOR "Sha256MB" LOG_SEPARATOR "%llu" LOG_SEPARATOR "%llu" LOG_SEPARATOR "\n", file,
               (unsigned long long)w, (unsigned long long)(t1 - t0), (unsigned long long)window);
        if (verify) { fold_digests(digests, 32 * window, &fold_mb); n_mb += window; }
    }
    free(digests);
}

static void walk(const char *dir)
{
    struct dirent **names;
    int n = scandir(dir, &names, NULL, alphasort);
    for (int i = 0; i < n; i++) {
        char path[4096];
        struct stat st;
        if (names[i]->d_name[0] == '.') { free(names[i]); continue; }
        snprintf(path, sizeof path, "%s/%s", dir, names[i]->d_name);
        free(names[i]);
        if (stat(path, &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) { walk(path); continue; }
        size_t nb;
        uint8_t *blocks = read_blocks(path, &nb);
        if (blocks && nb) {
            run_sb(path, blocks, nb);
            for (size_t window = 1; window <= 64; window++) run_mb(path, blocks, nb, window); /* test.cpp:87-90 */
        }
        free(blocks);
    }
    if (n >= 0) free(names);
}

int main(int argc, char **argv)
{
    if (argc == 3 && strcmp(argv[1], "--verify") == 0) { verify = 1; argv++; argc--; }
    if (argc != 2) { /* ASSERT_OP(argc, ==, 2) (test.cpp:70) */
        fprintf(stderr, "Usage: %s [--verify] <data-dir>\n", argv[0]);
        return 1;
    }
    if (cw_init(0) != CW_OK) {
        fprintf(stderr, "libcwhc: %s\n", cw_last_error());
        return 2;
    }
    cw_set_block_size(kBlockSize);
    walk(argv[1]);
    if (verify) {
        printf("verify" LOG_SEPARATOR "Skein256" LOG_SEPARATOR "%llu" LOG_SEPARATOR "%016llx\n", (unsigned long long)n_sk, (unsigned long long)fold_sk);
        printf("verify" LOG_SEPARATOR "Sha256" LOG_SEPARATOR "%llu" LOG_SEPARATOR "%016llx\n", (unsigned long long)n_sha, (unsigned long long)fold_sha);
        printf("verify" LOG_SEPARATOR "Sha256MB" LOG_SEPARATOR "%llu" LOG_SEPARATOR "%016llx\n", (unsigned long long)n_mb, (unsigned long long)fold_mb);
    }
    cw_shutdown();
    return 0;
}
