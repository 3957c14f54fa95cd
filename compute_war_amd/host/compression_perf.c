/*
 * compression_perf.c -- the lz4 / lzf subset of the reference's per-block codec experiment
 * (src/compression_perf/src/experiment.cpp) re-stated in C over the C ABI of libcwhc.so, same log format, so the
 * reference's notebooks can plot the device numbers and the corpus ratios can be compared line by line:
 *
 *   for every FULL 4 KiB block of every file (experiment.cpp:94-103; BLKSIZ 4096, :35), per selected codec
 *       alg|csize|comp_us|decomp_us|file|block          (lzf :105-125, lz4 :243-267; block counts from 1)
 *   --best / -B prints only the smallest line of each block (:507-510), lzf winning ties because lz4 replaces it
 *   only when strictly smaller (:261).
 *
 * Options as in the reference (:540-566): -4/--lz4, -f/--lzf, -B/--best, -v/--verbose; arguments are files or
 * directories (:515-537).  The other seven codecs of the experiment are outside this build's scope (SURVEY.md 8
 * "next" row N3 is the lz4/lzf front-end only) and asking for one is an error, not a silent skip.
 *
 * Default mode times ONE synchronous slot call per block (H2D + kernel + D2H), i.e. the latency a caller of
 * LZ4_compress_default / lzf_compress would see from the device.  --batch compresses and decodes each file in one
 * call per codec and reports every block with the per-block average of that call (the throughput view); csize is
 * exact in both modes.  The decoded bytes are compared with the input in both modes (the reference does not check).
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "../../include/cw_hashcompress.h"

enum { BLKSIZ = 4096 }; /* experiment.cpp:35 */

static struct { int lzf, lz4, best, verbose, batch; } flags;

static uint64_t now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (uint64_t)t.tv_sec * 1000000u + (uint64_t)t.tv_nsec / 1000u;
}

static void die_cw(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, cw_last_error());
    exit(2);
}

static void mismatch(const char *alg, const char *fname, size_t block)
{
    fprintf(stderr, "%s: decoded bytes differ from the input: %s block %zu\n", alg, fname, block);
    exit(3);
}

/* one slot call per block, as the reference times them */
static void per_block(const char *fname, const uint8_t *data, size_t nblocks)
{
    static char outbuf[BLKSIZ * 2], vbuf[BLKSIZ * 2];
    for (size_t i = 0; i < nblocks; i++) {
        const char *inbuf = (const char *)data + i * BLKSIZ;
        unsigned best = BLKSIZ;
        char line[4608], bname[4608] = "";
        if (flags.lzf) {
            uint64_t t0 = now_us();
            unsigned csize = (unsigned)cw_compress_lzf(inbuf, outbuf, BLKSIZ); /* lzf_compress(in, BLKSIZ, out, BLKSIZ - 1) */
            uint64_t t1 = now_us();
            unsigned vsize = cw_decompress_lzf(outbuf, csize, vbuf, sizeof vbuf);
            uint64_t t2 = now_us();
            if (csize && (vsize != BLKSIZ || memcmp(vbuf, inbuf, BLKSIZ))) mismatch("lzf", fname, i + 1);
            snprintf(line, sizeof line, "lzf|%u|%llu|%llu|%s|%zu\n", csize, (unsigned long long)(t1 - t0),
                     (unsigned long long)(t2 - t1), fname, i + 1);
            best = csize; /* :115 -- unconditionally, also when lzf returned 0 */
            strcpy(bname, line);
            if (!flags.best) fputs(line, stdout);
        }
        if (flags.lz4) {
            uint64_t t0 = now_us();
            unsigned csize = (unsigned)cw_compress_lz4(inbuf, outbuf, BLKSIZ); /* LZ4_compress_default(in, out, BLKSIZ, 2*BLKSIZ) */
            uint64_t t1 = now_us();
            int ret = cw_decompress_lz4(outbuf, vbuf, (int)csize, (int)sizeof vbuf);
            uint64_t t2 = now_us();
            if (ret != BLKSIZ || memcmp(vbuf, inbuf, BLKSIZ)) mismatch("lz4", fname, i + 1);
            snprintf(line, sizeof line, "lz4|%u|%llu|%llu|%s|%zu\n", csize, (unsigned long long)(t1 - t0),
                     (unsigned long long)(t2 - t1), fname, i + 1);
            if (csize < best) { best = csize; strcpy(bname, line); }
            if (!flags.best) fputs(line, stdout);
        }
        if (flags.best) fputs(bname, stdout);
    }
}

/* one call per codec and file; every block reports the call's per-block average */
static void batched(const char *fname, const uint8_t *data, size_t nblocks)
{
    const size_t stride = cw_compress_bound(CW_COMP_LZ4, BLKSIZ);
    uint8_t *comp[2] = {NULL, NULL}, *plain = (uint8_t *)malloc(nblocks * BLKSIZ);
    uint32_t *sizes[2] = {NULL, NULL}, *status = (uint32_t *)malloc(nblocks * sizeof(uint32_t));
    uint64_t c_us[2] = {0, 0}, d_us[2] = {0, 0};
    const int algs[2] = {CW_COMP_LZF, CW_COMP_LZ4}, on[2] = {flags.lzf, flags.lz4};
    const char *names[2] = {"lzf", "lz4"};
    for (int a = 0; a < 2; a++) {
        if (!on[a]) continue;
        comp[a] = (uint8_t *)malloc(nblocks * stride);
        sizes[a] = (uint32_t *)malloc(nblocks * sizeof(uint32_t));
        uint64_t t0 = now_us();
        if (cw_compress_blocks(algs[a], data, BLKSIZ, nblocks, comp[a], stride, sizes[a]) != CW_OK) die_cw("cw_compress_blocks");
        uint64_t t1 = now_us();
        if (cw_decompress_blocks(algs[a], comp[a], stride, sizes[a], nblocks, plain, BLKSIZ, status) != CW_OK) die_cw("cw_decompress_blocks");
        uint64_t t2 = now_us();
        c_us[a] = (t1 - t0) / nblocks;
        d_us[a] = (t2 - t1) / nblocks;
        for (size_t i = 0; i < nblocks; i++) {
            if (sizes[a][i] == 0) continue; /* lzf: did not fit, nothing to decode */
            if (status[i] != 0 || memcmp(plain + i * BLKSIZ, data + i * BLKSIZ, BLKSIZ)) mismatch(names[a], fname, i + 1);
        }
    }
    for (size_t i = 0; i < nblocks; i++) {
        unsigned best = BLKSIZ;
        int best_alg = -1;
        for (int a = 0; a < 2; a++) {
            if (!on[a]) continue;
            const unsigned csize = sizes[a][i];
            if (a == 0 || csize < best) { best = csize; best_alg = a; }
            if (!flags.best)
                printf("%s|%u|%llu|%llu|%s|%zu\n", names[a], csize, (unsigned long long)c_us[a], (unsigned long long)d_us[a], fname, i + 1);
        }
        if (flags.best && best_alg >= 0)
            printf("%s|%u|%llu|%llu|%s|%zu\n", names[best_alg], best, (unsigned long long)c_us[best_alg],
                   (unsigned long long)d_us[best_alg], fname, i + 1);
    }
    for (int a = 0; a < 2; a++) { free(comp[a]); free(sizes[a]); }
    free(plain);
    free(status);
}

static void process_file(const char *fname)
{
    if (flags.verbose) printf("Processing file: %s\n", fname);
    FILE *f = fopen(fname, "rb");
    if (!f) { fprintf(stderr, "Unable to open %s\n", fname); return; }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    const size_t nblocks = sz > 0 ? (size_t)sz / BLKSIZ : 0; /* only full blocks are compressed (:101-102) */
    uint8_t *data = (uint8_t *)malloc(nblocks ? nblocks * BLKSIZ : 1);
    if (nblocks && fread(data, BLKSIZ, nblocks, f) != nblocks) { fprintf(stderr, "Short read on %s\n", fname); fclose(f); free(data); return; }
    fclose(f);
    if (nblocks) (flags.batch ? batched : per_block)(fname, data, nblocks);
    free(data);
}

static void process(const char *path);

static void process_directory(const char *dirname)
{
    if (flags.verbose) printf("Processing directory: %s\n", dirname);
    struct dirent **names;
    const int n = scandir(dirname, &names, NULL, alphasort); /* the reference takes readdir order; sorted here so runs compare */
    if (n < 0) { fprintf(stderr, "Unable to open directory %s\n", dirname); return; }
    for (int i = 0; i < n; i++) {
        char path[4096];
        if (names[i]->d_name[0] != '.') { /* also skips . and .. (:54-57) */
            snprintf(path, sizeof path, "%s/%s", dirname, names[i]->d_name);
            process(path);
        }
        free(names[i]);
    }
    free(names);
}

static void process(const char *path)
{
    struct stat st;
    if (stat(path, &st) < 0) { fprintf(stderr, "Cannot stat %s\n", path); return; }
    if (S_ISDIR(st.st_mode)) process_directory(path);
    else if (S_ISREG(st.st_mode)) process_file(path);
    else fprintf(stderr, "Cannot process %s: not a regular file or directory\n", path);
}

int main(int argc, char **argv)
{
    static const struct option opts[] = {
        {"best", 0, NULL, 'B'}, {"bzip", 0, NULL, 'b'}, {"gzip", 0, NULL, 'g'}, {"lz4", 0, NULL, '4'},
        {"lzf", 0, NULL, 'f'},  {"lzo", 0, NULL, 'o'},  {"lzma", 0, NULL, 'a'}, {"snappy", 0, NULL, 's'},
        {"isal", 0, NULL, 'i'}, {"verbose", 0, NULL, 'v'}, {"batch", 0, NULL, 'T'}, {NULL, 0, NULL, 0}};
    int opt;
    while ((opt = getopt_long(argc, argv, "Bbg4foasivT", opts, NULL)) != -1) {
        switch (opt) {
        case 'B': flags.best = 1; break;
        case '4': flags.lz4 = 1; break;
        case 'f': flags.lzf = 1; break;
        case 'v': flags.verbose = 1; break;
        case 'T': flags.batch = 1; break;
        case 'b': case 'g': case 'o': case 'a': case 's': case 'i':
            fprintf(stderr, "%s: codec -%c is not part of this build (lz4 and lzf only)\n", argv[0], opt);
            return 1;
        default:
            fprintf(stderr, "Usage: %s [-4|--lz4] [-f|--lzf] [-B|--best] [-v] [--batch] <file-or-dir>...\n", argv[0]);
            return 1;
        }
    }
    if (flags.verbose) printf("best:   %d\nLZ4:    %d\nLZF:    %d\nVerbose:%d\n", flags.best, flags.lz4, flags.lzf, flags.verbose);
    if (cw_init(0) != CW_OK) die_cw("libcwhc");
    cw_set_block_size(BLKSIZ);
    for (int i = optind; i < argc; i++) process(argv[i]);
    cw_shutdown();
    return 0;
}
