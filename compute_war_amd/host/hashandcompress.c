/*
 * hashandcompress.c -- the reference's `hashAndCompress` driver re-stated in C (pthreads + getopt_long)
 * over the C ABI of libcwhc.so.  Same command line, same timed window, same report line as
 * src/hashandcompress/HashAndCompress.cpp:
 *
 *   options   -c/--c-threads  -g/--gpu-offload  -r/--read-blocks  -G/--hash-blocks (parsed, ignored: :334-335)
 *             -C/--comp-alg {lzf|lz4}   -H/--hash-alg {skein|sha256mb}  [+ skein512]      (:301-379)
 *             defaults from HashAndCompress.h:12-33 (offload false, read-blocks 8, threads 8, lz4, skein)
 *   input     every file is cut into read units of blockSize*readBlockFactor bytes; a partial tail is
 *             dropped (ReadFile :188-218); reading is OUTSIDE the timed window (:391-397)
 *   work      N worker threads pop read units; per unit each block is compressed, then the unit's
 *             blocks are hashed (ProcessBlock :231-261, PopAndProcessBlocks :263-272)
 *   report    hash|comp|totalTimeMS|throughputMBPS with MB = 2^20 and integer division (:406-412)
 *
 * Differences, all deliberate: blocks are kept as raw bytes (the reference's
 * std::string(rawData) truncates at the first NUL, :213) and every block of a unit is compressed (the
 * reference re-compresses block 0, :245-246).  Added flags: --block-size/-b (default 4096, :89) and
 * --verify/-v (print total compressed bytes and an XOR-fold of all digests).
 *
 * Two ways onto the GPU, both through include/cw_hashcompress.h:
 *   --gpu-offload=false  workers call the slot-compatible functions per read unit, exactly where the
 *                        reference calls doCompression/doHashing (:250,:257);
 *   --gpu-offload=true   workers hand whole spans of read units to the batched entry point
 *                        cw_hash_and_compress_blocks (what HashOffload::Start/Complete were meant to be).
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/cw_hashcompress.h"

#define LOG_SEPARATOR "|"

static size_t block_size = 4096;
static int read_block_factor = 8, n_threads = 8, gpu_offload = 0, verify = 0;
static const char *comp_name = "lz4", *hash_name = "skein";
static int comp_alg = CW_COMP_LZ4, hash_alg = CW_HASH_SKEIN256_128;

/* all read units, contiguous: unit u = data + u * unit_bytes */
static uint8_t *data = NULL;
static size_t n_units = 0, cap_units = 0, unit_bytes = 0;

static size_t next_unit = 0; /* the queue: workers take the next unit (or span of units) */
static pthread_mutex_t q_lock = PTHREAD_MUTEX_INITIALIZER;

static uint64_t total_comp = 0, digest_fold = 0;
static pthread_mutex_t r_lock = PTHREAD_MUTEX_INITIALIZER;

static void usage(const char *n, const char *msg)
{
    if (msg) fprintf(stderr, "%s\n", msg);
    fprintf(stderr,
            "Usage: %s [Options] [input-file]...\n"
            "  -h, --help             usage\n"
            "  -c, --c-threads N      compression threads (default 8)\n"
            "  -g, --gpu-offload B    use GPU offload (batched) path? (default false)\n"
            "  -r, --read-blocks N    read blocking factor (default 8)\n"
            "  -G, --hash-blocks N    hash grouping factor (ignored, as in the reference)\n"
            "  -C, --comp-alg A       lzf | lz4 (default lz4)\n"
            "  -H, --hash-alg A       skein | sha256mb | skein512 (default skein)\n"
            "  -b, --block-size N     block size in bytes (default 4096)\n"
            "  -v, --verify           also print total compressed bytes and a digest fold\n",
            n);
    exit(msg ? 1 : 0);
}

static int parse_bool(const char *s)
{
    return !(strcmp(s, "0") == 0 || strcmp(s, "false") == 0 || strcmp(s, "no") == 0 || strcmp(s, "off") == 0);
}

static void read_file(const char *path)
{
    FILE *f = strcmp(path, "-") == 0 ? stdin : fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "Unable to open file: %s\n", path);
        return;
    }
    for (;;) {
        if (n_units == cap_units) {
            cap_units = cap_units ? cap_units * 2 : 1024;
            data = (uint8_t *)realloc(data, cap_units * unit_bytes);
            if (!data) { fprintf(stderr, "out of memory\n"); exit(1); }
        }
        size_t got = fread(data + n_units * unit_bytes, 1, unit_bytes, f);
        if (got != unit_bytes) break; /* truncate partial last reads (:207-210) */
        n_units++;
    }
    if (f != stdin) fclose(f);
}

static void fold_digests(const uint8_t *d, size_t bytes, uint64_t *acc)
{
    for (size_t i = 0; i + 8 <= bytes; i += 8) {
        uint64_t v;
        memcpy(&v, d + i, 8);
        *acc ^= v;
    }
}

/* ProcessBlock through the two slots, one read unit at a time */
static void process_unit_slots(const uint8_t *unit, uint8_t *hashes, uint8_t *compressed, uint64_t *comp, uint64_t *fold)
{
    const size_t db = cw_digest_bytes(hash_alg);
    for (int b = 0; b < read_block_factor; b++) {
        const char *blk = (const char *)unit + (size_t)b * block_size;
        char *out = (char *)compressed + (size_t)b * 2 * block_size;
        size_t c = comp_alg == CW_COMP_LZ4 ? cw_compress_lz4(blk, out, block_size) : cw_compress_lzf(blk, out, block_size);
        *comp += c ? c : block_size;
    }
    switch (hash_alg) {
    case CW_HASH_SKEIN256_128: cw_hash_skein((const char *)unit, (char *)hashes, read_block_factor); break;
    case CW_HASH_SKEIN512: cw_hash_skein512((const char *)unit, (char *)hashes, read_block_factor); break;
    default: cw_hash_sha256mb((const char *)unit, (char *)hashes, read_block_factor); break;
    }
    fold_digests(hashes, db * (size_t)read_block_factor, fold);
}

static void *worker(void *arg)
{
    (void)arg;
    const size_t db = cw_digest_bytes(hash_alg);
    const size_t bound = cw_compress_bound(comp_alg, block_size);
    /* offload path: a span of units per call keeps the device busy; slot path: one unit, like the reference */
    size_t span = gpu_offload ? ((size_t)64 << 20) / unit_bytes : 1;
    if (span == 0) span = 1;
    const size_t span_blocks = span * (size_t)read_block_factor;
    uint8_t *hashes = (uint8_t *)malloc(db * span_blocks);
    uint8_t *compressed = (uint8_t *)malloc((gpu_offload ? bound : 2 * block_size) * span_blocks);
    uint32_t *sizes = (uint32_t *)malloc(sizeof(uint32_t) * span_blocks);
    uint64_t comp = 0, fold = 0;
    if (!hashes || !compressed || !sizes) { fprintf(stderr, "out of memory\n"); exit(1); }

    for (;;) {
        pthread_mutex_lock(&q_lock);
        size_t first = next_unit;
        size_t n = n_units - first < span ? n_units - first : span;
        next_unit += n;
        pthread_mutex_unlock(&q_lock);
        if (n == 0) break;

        if (!gpu_offload) {
            process_unit_slots(data + first * unit_bytes, hashes, compressed, &comp, &fold);
        } else {
            const size_t nb = n * (size_t)read_block_factor;
            int rc = cw_hash_and_compress_blocks(hash_alg, comp_alg, data + first * unit_bytes, block_size, nb, hashes,
                                                 compressed, bound, sizes);
            if (rc != CW_OK) { fprintf(stderr, "libcwhc: %s\n", cw_last_error()); exit(2); }
            for (size_t i = 0; i < nb; i++) comp += sizes[i] ? sizes[i] : block_size;
            fold_digests(hashes, db * nb, &fold);
        }
    }
    pthread_mutex_lock(&r_lock);
    total_comp += comp;
    digest_fold ^= fold;
    pthread_mutex_unlock(&r_lock);
    free(hashes); free(compressed); free(sizes);
    return NULL;
}

int main(int argc, char **argv)
{
    static const struct option opts[] = {
        {"help", no_argument, 0, 'h'},           {"c-threads", required_argument, 0, 'c'},
        {"gpu-offload", required_argument, 0, 'g'}, {"read-blocks", required_argument, 0, 'r'},
        {"hash-blocks", required_argument, 0, 'G'}, {"comp-alg", required_argument, 0, 'C'},
        {"hash-alg", required_argument, 0, 'H'},  {"block-size", required_argument, 0, 'b'},
        {"verify", no_argument, 0, 'v'},         {0, 0, 0, 0}};
    int o;
    while ((o = getopt_long(argc, argv, "hc:g:r:G:C:H:b:v", opts, NULL)) != -1) {
        switch (o) {
        case 'h': usage(argv[0], NULL); break;
        case 'c': n_threads = atoi(optarg); break;
        case 'g': gpu_offload = parse_bool(optarg); break;
        case 'r': read_block_factor = atoi(optarg); break;
        case 'G': break; /* hashBlockFactor = readBlockFactor (:334-335) */
        case 'C': comp_name = optarg; break;
        case 'H': hash_name = optarg; break;
        case 'b': block_size = (size_t)atol(optarg); break;
        case 'v': verify = 1; break;
        default: usage(argv[0], "invalid option");
        }
    }
    if (strcmp(comp_name, "lzf") == 0) comp_alg = CW_COMP_LZF;
    else if (strcmp(comp_name, "lz4") == 0) comp_alg = CW_COMP_LZ4;
    else usage(argv[0], "invalid compression algorithm specified; please use either \"lzf\" or \"lz4\"");
    if (strcmp(hash_name, "skein") == 0) hash_alg = CW_HASH_SKEIN256_128;
    else if (strcmp(hash_name, "sha256mb") == 0) hash_alg = CW_HASH_SHA256;
    else if (strcmp(hash_name, "skein512") == 0) hash_alg = CW_HASH_SKEIN512;
    else usage(argv[0], "invalid hashing algorithm specified; please use either \"skein\" or \"sha256mb\"");
    if (n_threads < 1 || read_block_factor < 1 || block_size < 1 || block_size > CW_MAX_BLOCK_BYTES)
        usage(argv[0], "threads and read-blocks must be >= 1, block-size in 1..65536");

    if (cw_init(0) != CW_OK) { /* initializeGpu() (:95-98); there is no CPU path to fall back to */
        fprintf(stderr, "libcwhc: %s\n", cw_last_error());
        return 2;
    }
    cw_set_block_size(block_size);
    unit_bytes = block_size * (size_t)read_block_factor;

    /* Read all files into memory and chunk them into read units (:385-389) */
    if (optind >= argc) read_file("-");
    for (int i = optind; i < argc; i++) read_file(argv[i]);

    const uint64_t total_data = (uint64_t)unit_bytes * n_units;
    struct timespec t0, t1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) pthread_create(&tid[t], NULL, worker, NULL);
    for (int t = 0; t < n_threads; t++) pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);

    uint64_t ms = (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_nsec - t0.tv_nsec) / 1000000);
    uint64_t mbps = ms ? (total_data * 1000) / (ms * 1024 * 1024) : 0;
    printf("%s" LOG_SEPARATOR "%s" LOG_SEPARATOR "%llu" LOG_SEPARATOR "%llu\n", hash_name, comp_name,
           (unsigned long long)ms, (unsigned long long)mbps);
    if (verify)
        printf("blocks=%llu in=%llu out=%llu fold=%016llx\n", (unsigned long long)(n_units * (size_t)read_block_factor),
               (unsigned long long)total_data, (unsigned long long)total_comp, (unsigned long long)digest_fold);
    cw_shutdown();
    free(tid);
    free(data);
    return 0;
}
