/*
 * cwhc_stub.c -- TEST INFRASTRUCTURE: a CPU stand-in for libcwhc.so's C ABI (include/cw_hashcompress.h), built on the
 * oracle.  It exists so that the HOST programs of this repository (the C files under compute_war_amd/host: worker threads, unit queue,
 * shards over several devices, barriers, buffers) can run under ThreadSanitizer / AddressSanitizer / UBSan in the CPU
 * test suite, where there is no GPU -- and so that the multi-device logic can be driven with several FAKE devices
 * (CW_STUB_DEVICES=N).  It is never built into, linked with or loaded by the product: only tests/ builds it
 * (tests/stub/Makefile), statically, into sanitizer builds of the host programs.  "Device" memory is host memory, the
 * "RCCL gather" is memcpy + a sum, every compute entry point calls the oracle.
 *
 * The reference has no sanitizer builds at all (src/hashandcompress/Makefile:31, src/hashing_perf/Makefile:4-17).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/cw_hashcompress.h"
#include "../../oracle/cw_oracle.h"

static __thread char t_err[256] = "";
static __thread int t_device = -1;
static int g_default = -1;
static size_t g_block_size = 4096;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static int fail(int rc, const char *msg) { snprintf(t_err, sizeof t_err, "%s", msg); return rc; }

int cw_device_count(void)
{
    const char *e = getenv("CW_STUB_DEVICES");
    return e ? atoi(e) : 1;
}
int cw_init(int device)
{
    if (device < 0 || device >= cw_device_count()) return fail(CW_ERR_BAD_ARG, "device out of range (stub)");
    pthread_mutex_lock(&g_lock);
    if (g_default < 0) g_default = device;
    pthread_mutex_unlock(&g_lock);
    t_device = device;
    return CW_OK;
}
int cw_set_device(int device) { return cw_init(device); }
int cw_get_device(void) { return t_device >= 0 ? t_device : g_default; }
void cw_shutdown(void) { pthread_mutex_lock(&g_lock); g_default = -1; pthread_mutex_unlock(&g_lock); t_device = -1; }
const char *cw_last_error(void) { return t_err; }
const char *cw_version(void) { return "cwhc stub (oracle on the CPU; test infrastructure)"; }
size_t cw_digest_bytes(int h) { return cw_oracle_digest_bytes(h); }
size_t cw_compress_bound(int c, size_t l) { return c == CW_COMP_LZ4 ? l + l / 255 + 16 : c == CW_COMP_LZF ? (l ? l : 1) : 0; }
void cw_set_block_size(size_t b) { g_block_size = b; }
size_t cw_get_block_size(void) { return g_block_size; }

static void hash_one(int alg, const uint8_t *p, size_t n, uint8_t *d)
{
    if (alg == CW_HASH_SKEIN512) cw_oracle_skein512(p, n * 8, 512, d);
    else if (alg == CW_HASH_SKEIN256_128) cw_oracle_skein256(p, n * 8, 128, d);
    else cw_oracle_sha256(p, n, d);
}
static size_t comp_one(int alg, const uint8_t *p, size_t n, uint8_t *out, size_t cap)
{
    if (alg == CW_COMP_LZ4) return cw_oracle_lz4_compress(p, n, out, cap);
    return cw_oracle_lzf_compress(p, n, out, n - 1 < cap ? n - 1 : cap);
}

void cw_hash_skein(const char *s, char *d, int c) { for (int i = 0; i < c; i++) hash_one(CW_HASH_SKEIN256_128, (const uint8_t *)s + (size_t)i * g_block_size, g_block_size, (uint8_t *)d + 16 * i); }
void cw_hash_skein512(const char *s, char *d, int c) { for (int i = 0; i < c; i++) hash_one(CW_HASH_SKEIN512, (const uint8_t *)s + (size_t)i * g_block_size, g_block_size, (uint8_t *)d + 64 * i); }
void cw_hash_sha256mb(const char *s, char *d, int c) { for (int i = 0; i < c; i++) hash_one(CW_HASH_SHA256, (const uint8_t *)s + (size_t)i * g_block_size, g_block_size, (uint8_t *)d + 32 * i); }
size_t cw_compress_lz4(const char *s, char *d, size_t l) { return l ? cw_oracle_lz4_compress((const uint8_t *)s, l, (uint8_t *)d, l + l / 255 + 16 > 2 * l ? l + l / 255 + 16 : 2 * l) : 0; }
size_t cw_compress_lzf(const char *s, char *d, size_t l) { return l < 2 ? 0 : cw_oracle_lzf_compress((const uint8_t *)s, l, (uint8_t *)d, l - 1); }
int cw_decompress_lz4(const char *s, char *d, int c, int cap) { return (int)cw_oracle_lz4_decompress((const uint8_t *)s, (size_t)c, (uint8_t *)d, (size_t)cap); }
unsigned cw_decompress_lzf(const void *s, unsigned c, void *d, unsigned cap)
{
    long r = cw_oracle_lzf_decompress((const uint8_t *)s, c, (uint8_t *)d, cap);
    return r < 0 ? 0u : (unsigned)r;
}

int cw_hash_and_compress_blocks(int h, int c, const void *src, size_t bb, size_t n, void *digests, void *dst, size_t dst_stride, uint32_t *sizes)
{
    if (cw_get_device() < 0 && cw_init(0) != CW_OK) return CW_ERR_NO_DEVICE;
    const size_t db = cw_digest_bytes(h);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = (const uint8_t *)src + i * bb;
        if (dst && sizes && c != CW_COMP_NONE) sizes[i] = (uint32_t)comp_one(c, p, bb, (uint8_t *)dst + i * dst_stride, dst_stride);
        if (digests && h != CW_HASH_NONE) hash_one(h, p, bb, (uint8_t *)digests + i * db);
    }
    return CW_OK;
}
int cw_hash_blocks(int h, const void *src, size_t bb, size_t n, void *digests) { return cw_hash_and_compress_blocks(h, CW_COMP_NONE, src, bb, n, digests, NULL, 0, NULL); }
int cw_compress_blocks(int c, const void *src, size_t bb, size_t n, void *dst, size_t stride, uint32_t *sizes) { return cw_hash_and_compress_blocks(CW_HASH_NONE, c, src, bb, n, NULL, dst, stride, sizes); }
int cw_hash_and_compress_packed(int h, int c, const void *src, size_t bb, size_t n, void *digests, void *packed, size_t cap, uint64_t *offsets, uint32_t *sizes)
{
    if (cw_get_device() < 0 && cw_init(0) != CW_OK) return CW_ERR_NO_DEVICE;
    const size_t db = cw_digest_bytes(h), bound = cw_compress_bound(c, bb);
    uint8_t *tmp = (uint8_t *)malloc(bound + 16);
    uint64_t off = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = (const uint8_t *)src + i * bb;
        const size_t z = comp_one(c, p, bb, tmp, bound);
        if (off + z > cap) { free(tmp); return fail(CW_ERR_BAD_ARG, "packed stream too small (stub)"); }
        memcpy((uint8_t *)packed + off, tmp, z);
        offsets[i] = off; sizes[i] = (uint32_t)z; off += z;
        if (digests && h != CW_HASH_NONE) hash_one(h, p, bb, (uint8_t *)digests + i * db);
    }
    offsets[n] = off;
    free(tmp);
    return CW_OK;
}
int cw_decompress_blocks(int c, const void *comp, size_t stride, const uint32_t *sizes, size_t n, void *dst, size_t bb, uint32_t *status)
{
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = (const uint8_t *)comp + i * stride;
        long r = c == CW_COMP_LZ4 ? cw_oracle_lz4_decompress(p, sizes[i], (uint8_t *)dst + i * bb, bb) : cw_oracle_lzf_decompress(p, sizes[i], (uint8_t *)dst + i * bb, bb);
        status[i] = r == (long)bb ? 0u : 1u;
    }
    return CW_OK;
}
int cw_prepare(int h, int c, size_t bb, size_t n, int pinned) { (void)h; (void)c; (void)bb; (void)n; (void)pinned; return cw_get_device() < 0 ? cw_init(0) : CW_OK; }
void *cw_host_alloc(size_t b) { return malloc(b ? b : 1); }
void cw_host_free(void *p) { free(p); }
int cw_host_register(void *p, size_t b) { (void)p; (void)b; return CW_OK; }
int cw_host_unregister(void *p) { (void)p; return CW_OK; }

/* "device" memory and the device-resident entry points the C programs use */
void *cw_dev_alloc(size_t b) { return calloc(b ? b : 1, 1); }
void cw_dev_free(void *p) { free(p); }
int cw_dev_upload(void *d, const void *s, size_t b) { memcpy(d, s, b); return CW_OK; }
int cw_dev_download(void *d, const void *s, size_t b) { memcpy(d, s, b); return CW_OK; }
int cw_dev_synchronize(void) { return CW_OK; }
int cw_dev_gen_random(uint64_t seed, uint64_t first, size_t n, size_t bb, void *d, void *s) { (void)s; cw_oracle_gen_random_blocks(seed, first, n, bb, (uint8_t *)d); return CW_OK; }
int cw_dev_gen_mixed(uint64_t seed, uint64_t first, size_t n, size_t bb, void *d, void *s) { (void)s; cw_oracle_gen_mixed_blocks(seed, first, n, bb, (uint8_t *)d); return CW_OK; }
int cw_dev_hash_and_compress(int h, int c, const void *src, size_t bb, size_t stride, size_t n, void *dig, void *dst, size_t dstride, uint32_t *sizes, void *s)
{
    (void)s;
    const size_t db = cw_digest_bytes(h);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = (const uint8_t *)src + i * stride;
        sizes[i] = (uint32_t)comp_one(c, p, bb, (uint8_t *)dst + i * dstride, dstride);
        hash_one(h, p, bb, (uint8_t *)dig + i * db);
    }
    return CW_OK;
}
int cw_dev_sum_sizes(const uint32_t *sizes, size_t n, uint32_t raw, uint64_t *tot, void *s)
{
    (void)s;
    for (size_t i = 0; i < n; i++) { tot[0] += sizes[i] ? sizes[i] : raw; tot[1] += sizes[i] == 0; }
    return CW_OK;
}

/* several fake devices */
void cw_shard_range(size_t n, int g, int G, size_t *first, size_t *last)
{
    if (G < 1) G = 1;
    if (g < 0) g = 0;
    if (g >= G) g = G - 1;
    if (first) *first = (size_t)((unsigned __int128)n * (unsigned)g / (unsigned)G);
    if (last) *last = (size_t)((unsigned __int128)n * (unsigned)(g + 1) / (unsigned)G);
}
struct cw_mgpu { int n; };
const char *cw_mgpu_last_error(void) { return t_err; }
cw_mgpu_t *cw_mgpu_create(const int *devs, int n)
{
    for (int i = 0; i < n; i++) if (cw_init(devs[i]) != CW_OK) return NULL;
    cw_mgpu_t *m = (cw_mgpu_t *)malloc(sizeof *m);
    m->n = n;
    return m;
}
void cw_mgpu_destroy(cw_mgpu_t *m) { free(m); }
int cw_mgpu_ndev(const cw_mgpu_t *m) { return m ? m->n : 0; }
int cw_mgpu_device(const cw_mgpu_t *m, int r) { return m && r >= 0 && r < m->n ? r : -1; }
int cw_mgpu_gather(cw_mgpu_t *m, const void *const *loc, size_t each, void *const *all, uint64_t *const *tot, size_t nt)
{
    if (each && loc && all)
        for (int g = 0; g < m->n; g++)
            for (int r = 0; r < m->n; r++) memcpy((uint8_t *)all[g] + (size_t)r * each, loc[r], each);
    if (nt && tot) {
        uint64_t sum[8] = {0};
        for (int r = 0; r < m->n; r++) for (size_t k = 0; k < nt && k < 8; k++) sum[k] += tot[r][k];
        for (int g = 0; g < m->n; g++) for (size_t k = 0; k < nt && k < 8; k++) tot[g][k] = sum[k];
    }
    return CW_OK;
}
