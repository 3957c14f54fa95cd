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
 *   --gpu-offload=true   workers hand whole spans of read units to the pipelined batch entry point
 *                        cw_hash_and_compress_packed (what HashOffload::Start/Complete were meant to be); the read
 *                        units are page-locked once, outside the timed window, so the copy engines use them in place.
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
static int data_pinned = 0; /* data came from cw_host_alloc */

/* the queue: device g owns the units [shard_next[g], shard_end[g]) (contiguous shards); a worker takes the
 * next unit (or span of units) of its device's shard */
#define MAX_DEVICES 16
static int n_devices = 1, devices_given = 0;
static size_t shard_next[MAX_DEVICES], shard_end[MAX_DEVICES];
static uint64_t dev_in[MAX_DEVICES], dev_out[MAX_DEVICES];
static pthread_mutex_t q_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_barrier_t start_bar; /* workers set up (device context, buffers) in front of it; the timed window opens behind it */
static pthread_barrier_t end_bar;   /* ... and closes when every worker has handed in its totals: releasing gigabytes of page-locked and
                                     * device memory (hundreds of milliseconds) is teardown, like the setup in front of start_bar */

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
            "  -D, --devices N        GPUs to shard the read units over (default 1)\n"
            "  -v, --verify           also print total compressed bytes and a digest fold\n",
            n);
    exit(msg ? 1 : 0);
}

static int parse_bool(const char *s)
{
    return !(strcmp(s, "0") == 0 || strcmp(s, "false") == 0 || strcmp(s, "no") == 0 || strcmp(s, "off") == 0);
}

/* Whole read units of every regular file, counted first so that the units can live in ONE page-locked allocation the copy
 * engines read in place (a buffer that is page-locked after the fact, cw_host_register, measured at half the rate). */
static size_t units_of(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    size_t n = 0;
    if (fseek(f, 0, SEEK_END) == 0) { long e = ftell(f); if (e > 0) n = (size_t)e / unit_bytes; }
    fclose(f);
    return n;
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
            if (data_pinned) break; /* sized exactly for the files' whole units */
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
    const int dev = (int)(intptr_t)arg;
    const size_t db = cw_digest_bytes(hash_alg);
    const size_t bound = cw_compress_bound(comp_alg, block_size);
    /* offload path: a span of units per call keeps the device's pipeline busy; slot path: one unit, like the reference */
    size_t span = gpu_offload ? ((size_t)8 << 30) / unit_bytes : 1; /* 16 chunks of the library's pipeline per call: its fill and drain (~25 ms) once per 8 GiB */
    if (span == 0) span = 1;
    {   /* never more than this worker's device has to offer: a 226 MB dataset does not page-lock 8.6 GiB per worker */
        const size_t shard = shard_end[dev] - shard_next[dev]; /* (set before the workers start; read-only until the start barrier) */
        if (gpu_offload && span > shard) span = shard ? shard : 1;
    }
    const size_t span_blocks = span * (size_t)read_block_factor;
    if (cw_set_device(dev) != CW_OK) { fprintf(stderr, "libcwhc: %s\n", cw_last_error()); exit(2); }
    uint8_t *hashes = (uint8_t *)malloc(db * span_blocks);
    /* offload: one packed stream in page-locked memory (the device-to-host copy lands in place); slots: the reference's 2*l each */
    uint8_t *compressed = gpu_offload ? (uint8_t *)cw_host_alloc(bound * span_blocks) : (uint8_t *)malloc(2 * block_size * span_blocks);
    uint32_t *sizes = (uint32_t *)malloc(sizeof(uint32_t) * span_blocks);
    uint64_t *offsets = (uint64_t *)malloc(sizeof(uint64_t) * (span_blocks + 1));
    uint64_t comp = 0, fold = 0, in = 0;
    if (!hashes || !compressed || !sizes || !offsets) { fprintf(stderr, "out of memory\n"); exit(1); }
    /* initializeGpu() (:95-98) per worker: context and batch buffers exist before the clock starts, as the reference's
     * device setup would (it sits in front of the reads, :381-383) */
    if (gpu_offload && cw_prepare(hash_alg, comp_alg, block_size, span_blocks, 1) != CW_OK) { fprintf(stderr, "libcwhc: %s\n", cw_last_error()); exit(2); }
    pthread_barrier_wait(&start_bar);

    for (;;) {
        pthread_mutex_lock(&q_lock);
        size_t first = shard_next[dev];
        size_t n = shard_end[dev] - first < span ? shard_end[dev] - first : span;
        shard_next[dev] += n;
        pthread_mutex_unlock(&q_lock);
        if (n == 0) break;
        in += n * unit_bytes;

        if (!gpu_offload) {
            process_unit_slots(data + first * unit_bytes, hashes, compressed, &comp, &fold);
        } else {
            const size_t nb = n * (size_t)read_block_factor;
            int rc = cw_hash_and_compress_packed(hash_alg, comp_alg, data + first * unit_bytes, block_size, nb, hashes, compressed,
                                                 bound * span_blocks, offsets, sizes);
            if (rc != CW_OK) { fprintf(stderr, "libcwhc: %s\n", cw_last_error()); exit(2); }
            for (size_t i = 0; i < nb; i++) comp += sizes[i] ? sizes[i] : block_size;
            fold_digests(hashes, db * nb, &fold);
        }
    }
    pthread_mutex_lock(&r_lock);
    total_comp += comp;
    digest_fold ^= fold;
    dev_in[dev] += in;
    dev_out[dev] += comp;
    pthread_mutex_unlock(&r_lock);
    pthread_barrier_wait(&end_bar);
    free(hashes); free(sizes); free(offsets);
    if (gpu_offload) cw_host_free(compressed); else free(compressed);
    return NULL;
}

/* the per-device byte totals, summed over the devices with RCCL (ncclAllReduce over xGMI): every device ends up with
 * the node's totals, device 0's copy is reported */
static int gather_totals(uint64_t out[2])
{
    int devs[MAX_DEVICES];
    uint64_t *d_tot[MAX_DEVICES];
    for (int g = 0; g < n_devices; g++) devs[g] = g;
    cw_mgpu_t *mg = cw_mgpu_create(devs, n_devices);
    if (!mg) { fprintf(stderr, "libcwhc: %s\n", cw_mgpu_last_error()); return -1; }
    for (int g = 0; g < n_devices; g++) {
        const uint64_t t[2] = {dev_in[g], dev_out[g]};
        if (cw_set_device(g) != CW_OK || !(d_tot[g] = (uint64_t *)cw_dev_alloc(sizeof t)) || cw_dev_upload(d_tot[g], t, sizeof t) != CW_OK) {
            fprintf(stderr, "libcwhc: %s\n", cw_last_error());
            return -1;
        }
    }
    int rc = cw_mgpu_gather(mg, NULL, 0, NULL, d_tot, 2);
    if (rc != CW_OK) fprintf(stderr, "libcwhc: %s\n", cw_mgpu_last_error());
    if (rc == CW_OK && (cw_set_device(0) != CW_OK || cw_dev_download(out, d_tot[0], 2 * sizeof(uint64_t)) != CW_OK)) rc = -1;
    for (int g = 0; g < n_devices; g++) { (void)cw_set_device(g); cw_dev_free(d_tot[g]); }
    cw_mgpu_destroy(mg);
    return rc;
}

int main(int argc, char **argv)
{
    static const struct option opts[] = {
        {"help", no_argument, 0, 'h'},           {"c-threads", required_argument, 0, 'c'},
        {"gpu-offload", required_argument, 0, 'g'}, {"read-blocks", required_argument, 0, 'r'},
        {"hash-blocks", required_argument, 0, 'G'}, {"comp-alg", required_argument, 0, 'C'},
        {"hash-alg", required_argument, 0, 'H'},  {"block-size", required_argument, 0, 'b'},
        {"verify", no_argument, 0, 'v'},         {"devices", required_argument, 0, 'D'},
        {0, 0, 0, 0}};
    int o;
    while ((o = getopt_long(argc, argv, "hc:g:r:G:C:H:b:vD:", opts, NULL)) != -1) {
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
        case 'D': n_devices = atoi(optarg); devices_given = 1; break;
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
    if (n_devices < 1 || n_devices > MAX_DEVICES) usage(argv[0], "devices must be 1..16");
    {   /* fail early and loudly when the node has fewer devices than asked for: before any file is read or buffer pinned */
        const int have = cw_device_count();
        if (have < n_devices) {
            fprintf(stderr, "libcwhc: %d device(s) asked for, %d usable%s\n", n_devices, have, have ? "" : " (no HIP device)");
            return 2;
        }
    }
    if (n_threads < n_devices) n_threads = n_devices; /* every device needs a worker */
    /* --c-threads counts the reference's compute threads.  With the work on the device a worker only feeds one pipeline (three
     * 512 MiB slots of device memory, an 8.6 GiB page-locked output span), and a second pipeline on the same device shares the same
     * link: 38.9 GB/s with one worker, 27.1 with two.  run_tests' -c 14 therefore means 14 workers only on the CPU path. */
    if (gpu_offload && n_threads > n_devices && !getenv("CW_DRIVER_ALL_THREADS")) n_threads = n_devices; /* (the variable: tests of the worker logic) */

    for (int g = 0; g < n_devices; g++)
        if (cw_init(g) != CW_OK) { /* initializeGpu() (:95-98), per device; there is no CPU path to fall back to */
            fprintf(stderr, "libcwhc: %s\n", cw_last_error());
            return 2;
        }
    cw_set_block_size(block_size);
    unit_bytes = block_size * (size_t)read_block_factor;

    /* Read all files into memory and chunk them into read units (:385-389) */
    if (gpu_offload && optind < argc) { /* named files: their units fit one page-locked allocation */
        size_t total = 0;
        for (int i = optind; i < argc; i++) total += units_of(argv[i]);
        if (total && (data = (uint8_t *)cw_host_alloc(total * unit_bytes)) != NULL) { data_pinned = 1; cap_units = total; }
    }
    if (optind >= argc) read_file("-");
    for (int i = optind; i < argc; i++) read_file(argv[i]);

    const uint64_t total_data = (uint64_t)unit_bytes * n_units;
    for (int g = 0; g < n_devices; g++) cw_shard_range(n_units, g, n_devices, &shard_next[g], &shard_end[g]);
    /* like the reads, page-locking the read units is preparation, not work: outside the timed window (:391-397) */
    const int locked = gpu_offload && !data_pinned && n_units && cw_host_register(data, n_units * unit_bytes) == CW_OK;
    struct timespec t0, t1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    pthread_barrier_init(&start_bar, NULL, (unsigned)n_threads + 1);
    pthread_barrier_init(&end_bar, NULL, (unsigned)n_threads + 1);
    for (int t = 0; t < n_threads; t++) pthread_create(&tid[t], NULL, worker, (void *)(intptr_t)(t % n_devices));
    pthread_barrier_wait(&start_bar); /* every worker is set up */
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&end_bar);   /* every block is processed and accounted for (the reference joins here: its workers own nothing to release) */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < n_threads; t++) pthread_join(tid[t], NULL);

    uint64_t ms = (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_nsec - t0.tv_nsec) / 1000000);
    uint64_t mbps = ms ? (total_data * 1000) / (ms * 1024 * 1024) : 0;
    printf("%s" LOG_SEPARATOR "%s" LOG_SEPARATOR "%llu" LOG_SEPARATOR "%llu\n", hash_name, comp_name,
           (unsigned long long)ms, (unsigned long long)mbps);
    if (verify)
        printf("blocks=%llu in=%llu out=%llu fold=%016llx\n", (unsigned long long)(n_units * (size_t)read_block_factor),
               (unsigned long long)total_data, (unsigned long long)total_comp, (unsigned long long)digest_fold);
    if (devices_given) { /* the node's totals as the devices hold them after the RCCL reduction */
        uint64_t t[2] = {0, 0};
        if (gather_totals(t) != CW_OK) return 2;
        printf("devices=%d in=%llu out=%llu (ncclAllReduce over the per-device totals)\n", n_devices, (unsigned long long)t[0], (unsigned long long)t[1]);
        if (t[0] != total_data || t[1] != total_comp) { fprintf(stderr, "device totals disagree with the host's\n"); return 3; }
    }
    if (locked) (void)cw_host_unregister(data);
    if (data_pinned) cw_host_free(data); else free(data);
    cw_shutdown();
    free(tid);
    return 0;
}
