/*
 * mgpu_stream.c -- BASELINE.json configs[4] from C: Skein-512 + LZ4 (or any hash/codec pair) over a synthetic block
 * stream that is sharded contiguously over the GPUs of one node, device-resident, with the per-pass result
 * gather over RCCL (cw_mgpu_gather: ncclAllGather of the digests, ncclAllReduce of the byte totals).
 *
 * The reference has no multi-device path at all -- its only parallelism is worker threads over independent blocks
 * (src/hashandcompress/HashAndCompress.cpp:398-403) and its --gpu-offload seam (:305,331) is dormant; this program is
 * that worker loop with "thread" replaced by "device": one host thread per device, each processing its own shard with
 * the single-device entry points of include/cw_hashcompress.h, no collective on the data path.  Weak scaling: the
 * blocks per GPU are fixed.  The timed window covers `steps` passes including their gathers (inputs resident in HBM,
 * like the reference's window that starts after the files are read, :391-397).
 *
 *   mgpu_stream --devices 8 --blocks-per-gpu 1048576 --block-size 65536 --steps 5 --warmup 1 -H skein512 -C lz4
 * prints the reference's report line hash|comp|ms|MB/s (:409-412), then one JSON object (GB/s, ratio, digest fold).
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

static int n_dev = 1, steps = 3, warmup = 1, hash_alg = CW_HASH_SKEIN512, comp_alg = CW_COMP_LZ4, mixed = 0;
static size_t blocks_per_gpu = 16384, block_size = 65536;
static const char *hash_name = "skein512", *comp_name = "lz4";
static const uint64_t seed = 0xC0FFEE;

typedef struct {
    int rank, device, rc;
    size_t first, n;
    void *d_src, *d_dst, *d_digests, *d_all;
    uint32_t *d_sizes;
    uint64_t *d_totals;
    uint64_t fold;
} rank_t;

static rank_t *ranks;
static cw_mgpu_t *mg;
static pthread_barrier_t bar;
static size_t db, stride, shard_max;
static int gather_rc = CW_OK;

static void die_rank(rank_t *r, const char *what)
{
    fprintf(stderr, "mgpu_stream: rank %d (device %d): %s: %s\n", r->rank, r->device, what, cw_last_error());
    exit(2);
}

static void one_pass(rank_t *r)
{
    const uint64_t zero[2] = {0, 0};
    if (cw_dev_upload(r->d_totals, zero, sizeof zero) != CW_OK) die_rank(r, "totals");
    if (cw_dev_hash_and_compress(hash_alg, comp_alg, r->d_src, block_size, block_size, r->n, r->d_digests, r->d_dst, stride, r->d_sizes, NULL) != CW_OK)
        die_rank(r, "hash_and_compress");
    if (cw_dev_sum_sizes(r->d_sizes, r->n, (uint32_t)block_size, r->d_totals, NULL) != CW_OK) die_rank(r, "sum_sizes");
    if (cw_dev_synchronize() != CW_OK) die_rank(r, "synchronize");
    pthread_barrier_wait(&bar);
    if (r->rank == 0) { /* the one exchange of the pass: digests + totals of every shard to every device */
        const void **loc = (const void **)malloc(sizeof(void *) * (size_t)n_dev);
        void **all = (void **)malloc(sizeof(void *) * (size_t)n_dev);
        uint64_t **tot = (uint64_t **)malloc(sizeof(uint64_t *) * (size_t)n_dev);
        for (int g = 0; g < n_dev; g++) { loc[g] = ranks[g].d_digests; all[g] = ranks[g].d_all; tot[g] = ranks[g].d_totals; }
        gather_rc = cw_mgpu_gather(mg, loc, shard_max * db, all, tot, 2);
        free(loc); free(all); free(tot);
    }
    pthread_barrier_wait(&bar);
    if (gather_rc != CW_OK) { if (r->rank == 0) fprintf(stderr, "mgpu_stream: %s\n", cw_mgpu_last_error()); exit(2); }
}

static void *rank_main(void *arg)
{
    rank_t *r = (rank_t *)arg;
    if (cw_set_device(r->device) != CW_OK) die_rank(r, "set_device");
    r->d_src = cw_dev_alloc(r->n * block_size);
    r->d_dst = cw_dev_alloc(r->n * stride);
    r->d_sizes = (uint32_t *)cw_dev_alloc(r->n * 4);
    r->d_digests = cw_dev_alloc(shard_max * db); /* padded to the largest shard: ncclAllGather wants equal contributions */
    r->d_all = cw_dev_alloc((size_t)n_dev * shard_max * db);
    r->d_totals = (uint64_t *)cw_dev_alloc(16);
    if (!r->d_src || !r->d_dst || !r->d_sizes || !r->d_digests || !r->d_all || !r->d_totals) die_rank(r, "device memory");
    if ((mixed ? cw_dev_gen_mixed : cw_dev_gen_random)(seed, r->first, r->n, block_size, r->d_src, NULL) != CW_OK) die_rank(r, "generate");
    if (cw_dev_synchronize() != CW_OK) die_rank(r, "synchronize");
    for (int i = 0; i < warmup; i++) one_pass(r);
    pthread_barrier_wait(&bar); /* rank 0 stamps t0 behind this barrier, t1 behind the last pass's */
    for (int i = 0; i < steps; i++) one_pass(r);
    pthread_barrier_wait(&bar);
    { /* every device now holds every shard's digests: fold its copy, the folds must agree */
        const size_t bytes = (size_t)n_dev * shard_max * db;
        uint8_t *h = (uint8_t *)malloc(bytes);
        if (cw_dev_download(h, r->d_all, bytes) != CW_OK) die_rank(r, "download");
        uint64_t f = 0;
        for (int g = 0; g < n_dev; g++)
            for (size_t i = 0; i + 8 <= ranks[g].n * db; i += 8) {
                uint64_t v;
                memcpy(&v, h + (size_t)g * shard_max * db + i, 8);
                f ^= v;
            }
        r->fold = f;
        free(h);
    }
    return NULL;
}

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv)
{
    static const struct option opts[] = {
        {"devices", required_argument, 0, 'D'}, {"blocks-per-gpu", required_argument, 0, 'n'}, {"block-size", required_argument, 0, 'b'},
        {"steps", required_argument, 0, 's'},   {"warmup", required_argument, 0, 'w'},         {"hash-alg", required_argument, 0, 'H'},
        {"comp-alg", required_argument, 0, 'C'}, {"data", required_argument, 0, 'd'},          {"help", no_argument, 0, 'h'},
        {"blocks", required_argument, 0, 'N'}, {0, 0, 0, 0}};
    size_t total_given = 0; /* --blocks: the whole stream's length; shards then differ by one block when devices do not divide it */
    int o;
    while ((o = getopt_long(argc, argv, "D:n:N:b:s:w:H:C:d:h", opts, NULL)) != -1) {
        switch (o) {
        case 'D': n_dev = atoi(optarg); break;
        case 'n': blocks_per_gpu = (size_t)atol(optarg); break;
        case 'N': total_given = (size_t)atol(optarg); break;
        case 'b': block_size = (size_t)atol(optarg); break;
        case 's': steps = atoi(optarg); break;
        case 'w': warmup = atoi(optarg); break;
        case 'H': hash_name = optarg; break;
        case 'C': comp_name = optarg; break;
        case 'd': mixed = strcmp(optarg, "mixed") == 0; break;
        default:
            fprintf(stderr, "Usage: %s [--devices N] [--blocks-per-gpu B | --blocks TOTAL] [--block-size S] [--steps K] [--warmup W] [-H skein|skein512|sha256mb] "
                            "[-C lz4|lzf] [--data random|mixed]\n", argv[0]);
            return o == 'h' ? 0 : 1;
        }
    }
    if (strcmp(comp_name, "lzf") == 0) comp_alg = CW_COMP_LZF;
    else if (strcmp(comp_name, "lz4") == 0) comp_alg = CW_COMP_LZ4;
    else { fprintf(stderr, "invalid compression algorithm specified; please use either \"lzf\" or \"lz4\"\n"); return 1; }
    if (strcmp(hash_name, "skein") == 0) hash_alg = CW_HASH_SKEIN256_128;
    else if (strcmp(hash_name, "sha256mb") == 0) hash_alg = CW_HASH_SHA256;
    else if (strcmp(hash_name, "skein512") == 0) hash_alg = CW_HASH_SKEIN512;
    else { fprintf(stderr, "invalid hashing algorithm specified; please use either \"skein\" or \"sha256mb\"\n"); return 1; }
    if (n_dev < 1 || n_dev > 16 || steps < 1 || warmup < 0 || blocks_per_gpu < 1 || block_size < 16 || block_size % 16 || block_size > CW_MAX_BLOCK_BYTES) {
        fprintf(stderr, "devices 1..16, steps >= 1, block-size a multiple of 16 in 16..65536\n");
        return 1;
    }
    const int have = cw_device_count();
    if (have < n_dev) {
        fprintf(stderr, "libcwhc: %d device(s) asked for, %d usable%s\n", n_dev, have, have ? "" : " (no HIP device)");
        return 2;
    }
    int devs[16];
    for (int g = 0; g < n_dev; g++) devs[g] = g;
    mg = cw_mgpu_create(devs, n_dev);
    if (!mg) { fprintf(stderr, "libcwhc: %s\n", cw_mgpu_last_error()); return 2; }

    db = cw_digest_bytes(hash_alg);
    stride = (cw_compress_bound(comp_alg, block_size) + 15) / 16 * 16;
    if (total_given && total_given < (size_t)n_dev) { fprintf(stderr, "--blocks: at least one block per device\n"); return 1; }
    const size_t total_blocks = total_given ? total_given : blocks_per_gpu * (size_t)n_dev;
    if (total_given) blocks_per_gpu = total_blocks / (size_t)n_dev; /* (reported; the shards are [g*N/G, (g+1)*N/G)) */
    ranks = (rank_t *)calloc((size_t)n_dev, sizeof(rank_t));
    shard_max = 0;
    for (int g = 0; g < n_dev; g++) {
        size_t a, b;
        cw_shard_range(total_blocks, g, n_dev, &a, &b);
        ranks[g].rank = g; ranks[g].device = devs[g]; ranks[g].first = a; ranks[g].n = b - a;
        if (b - a > shard_max) shard_max = b - a;
    }
    pthread_barrier_init(&bar, NULL, (unsigned)n_dev + 1);
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_dev);
    for (int g = 0; g < n_dev; g++) pthread_create(&tid[g], NULL, rank_main, &ranks[g]);
    /* the main thread joins every barrier of the ranks: 2 per pass, 1 before the timed passes, 1 after */
    for (int i = 0; i < 2 * warmup; i++) pthread_barrier_wait(&bar);
    pthread_barrier_wait(&bar);
    const double t0 = now_s();
    for (int i = 0; i < 2 * steps; i++) pthread_barrier_wait(&bar);
    const double t1 = now_s();
    pthread_barrier_wait(&bar);
    for (int g = 0; g < n_dev; g++) pthread_join(tid[g], NULL);

    uint64_t totals[2];
    if (cw_set_device(devs[0]) != CW_OK || cw_dev_download(totals, ranks[0].d_totals, sizeof totals) != CW_OK) {
        fprintf(stderr, "libcwhc: %s\n", cw_last_error());
        return 2;
    }
    int agree = 1;
    for (int g = 1; g < n_dev; g++) agree &= ranks[g].fold == ranks[0].fold;
    const double secs = t1 - t0, in_bytes = (double)total_blocks * (double)block_size * steps;
    const uint64_t ms = (uint64_t)(secs * 1000.0);
    printf("%s|%s|%llu|%llu\n", hash_name, comp_name, (unsigned long long)ms, (unsigned long long)(ms ? in_bytes / 1048576.0 * 1000.0 / (double)ms : 0));
    printf("{\"n_gpus\": %d, \"blocks\": %zu, \"blocks_per_gpu\": %zu, \"block_bytes\": %zu, \"steps\": %d, \"data\": \"%s\", \"GBps\": %.2f, \"ms_per_step\": %.3f, "
           "\"bytes_out\": %llu, \"ratio\": %.4f, \"digest_fold\": \"%016llx\", \"all_devices_agree\": %s, \"gather\": \"RCCL ncclAllGather + ncclAllReduce\"}\n",
           n_dev, total_blocks, blocks_per_gpu, block_size, steps, mixed ? "mixed" : "random", in_bytes / secs / 1e9, secs / steps * 1e3,
           (unsigned long long)totals[0], (double)total_blocks * (double)block_size / (double)totals[0], (unsigned long long)ranks[0].fold,
           agree ? "true" : "false");
    for (int g = 0; g < n_dev; g++) {
        (void)cw_set_device(devs[g]);
        cw_dev_free(ranks[g].d_src); cw_dev_free(ranks[g].d_dst); cw_dev_free(ranks[g].d_sizes);
        cw_dev_free(ranks[g].d_digests); cw_dev_free(ranks[g].d_all); cw_dev_free(ranks[g].d_totals);
    }
    cw_mgpu_destroy(mg);
    cw_shutdown();
    pthread_barrier_destroy(&bar);
    free(tid);
    free(ranks);
    return agree ? 0 : 3;
}
