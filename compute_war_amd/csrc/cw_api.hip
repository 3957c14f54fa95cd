// cw_api.hip -- host side of libcwhc.so: the C ABI of include/cw_hashcompress.h over the HIP kernels.
//
// Structure: a process-wide device selection (cw_init = the reference's empty initializeGpu(),
// src/hashandcompress/HashAndCompress.cpp:95-98), one lazily created context per calling host thread
// (own HIP stream + growable device staging buffers, because the reference invokes its slots from
// --c-threads workers with no locking, :398-402), the HashOffload batch object (HashOffload.h:13-64)
// and the single consumer thread that drains it (hashing_offload_entry_point, :160-183).
//
// No CPU fallback exists: every compute path ends in a kernel launch or an error.

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/cw_hashcompress.h"
#include "cw_device.h"

namespace {

thread_local char t_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(CW_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

// ---- optional per-thread kernel timing (cw_profile_*): HIP events on the stream each kernel is launched on ----
enum { PROF_CODEC = 0, PROF_HASH = 1, PROF_OTHER = 2, PROF_KINDS = 3 };
struct ProfSpan { int kind; hipEvent_t a, b; };
thread_local bool t_prof_on = false;
thread_local std::vector<ProfSpan> t_prof;

struct ProfScope { // brackets the launches made during its lifetime on `stream`
    int kind; hipStream_t stream; hipEvent_t a = nullptr;
    ProfScope(int k, hipStream_t s) : kind(k), stream(s)
    {
        if (t_prof_on && hipEventCreate(&a) == hipSuccess) (void)hipEventRecord(a, stream);
    }
    ~ProfScope()
    {
        if (!a) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) == hipSuccess) { (void)hipEventRecord(b, stream); t_prof.push_back({kind, a, b}); }
        else (void)hipEventDestroy(a);
    }
};

// ---- process-wide state -----------------------------------------------------------------------
std::mutex g_lock;
std::atomic<int> g_device{-1};
cw::SkeinIV g_iv512_512, g_iv256_128;
std::atomic<size_t> g_block_size{4096}; // the reference's global blockSize (:89)

int ensure_init()
{
    if (g_device.load(std::memory_order_acquire) >= 0) return CW_OK;
    return cw_init(0);
}

// ---- per-thread context -------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return CW_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n < (1u << 20) ? (1u << 20) : n;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(CW_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        cap = want;
        return CW_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinnedBuf { // host staging for the packed stream (one large D2H instead of one per block)
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return CW_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n < (1u << 20) ? (1u << 20) : n;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(CW_ERR_NOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
        cap = want;
        return CW_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// fork/join helper for cw_dev_hash_and_compress: the codec runs on a side stream beside the hash
struct SideStream {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
    int open()
    {
        const int dev = g_device.load();
        if (side && device == dev) return CW_OK;
        HIP_TRY(hipSetDevice(dev));
        int least = 0, greatest = 0; // the hash is the long, ALU-bound kernel: it yields dispatch slots to the codec
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least));
        HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        device = dev;
        return CW_OK;
    }
    ~SideStream()
    {
        if (!side) return;
        (void)hipEventDestroy(fork); (void)hipEventDestroy(join); (void)hipStreamDestroy(side);
    }
};
thread_local SideStream t_side;

struct ThreadCtx {
    hipStream_t stream = nullptr;
    DevBuf src, dst, dig, sizes, pack, offs;
    PinnedBuf hpack;
    bool ready = false;
    int open()
    {
        if (ready) return CW_OK;
        int rc = ensure_init();
        if (rc != CW_OK) return rc;
        HIP_TRY(hipSetDevice(g_device.load()));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        ready = true;
        return CW_OK;
    }
    ~ThreadCtx()
    {
        if (!ready) return;
        src.release(); dst.release(); dig.release(); sizes.release(); pack.release(); offs.release(); hpack.release();
        (void)hipStreamDestroy(stream);
    }
};
thread_local ThreadCtx t_ctx;

const size_t kMaxChunkBytes = (size_t)256 << 20; // host-API staging granularity

int check_block(size_t block_bytes)
{
    if (block_bytes > CW_MAX_BLOCK_BYTES) return fail(CW_ERR_BAD_ARG, "block_bytes %zu > %u", block_bytes, CW_MAX_BLOCK_BYTES);
    return CW_OK;
}

// sliced: long Skein messages (>= 256 steps, >= 4096 blocks) are hashed in several launches of short-lived wavefronts
int dev_hash(int alg, const uint8_t *d_src, size_t bb, size_t stride, size_t n, uint8_t *d_dig, hipStream_t s, bool lean = false,
             bool sliced = false)
{
    hipError_t e;
    ProfScope prof(PROF_HASH, s);
    static const char *slice_env = getenv("CW_SKEIN_SLICED"); // CW_SKEIN_SLICED=0: always the one-launch hash kernel (profiling knob)
    if (sliced && !(slice_env && slice_env[0] == '0') && (alg == CW_HASH_SKEIN512 || alg == CW_HASH_SKEIN256_128)) {
        const int nw = alg == CW_HASH_SKEIN512 ? 8 : 4;
        if (cw::skein_sliced_applies(nw, d_src, bb, stride, n, d_dig)) {
            e = cw::skein_sliced_launch(nw, d_src, bb, stride, n, nw == 8 ? g_iv512_512 : g_iv256_128, d_dig, nw == 8 ? 64 : 16, s);
            if (e != hipSuccess) return fail(CW_ERR_HIP, "hash launch: %s", hipGetErrorString(e));
            return CW_OK;
        }
    }
    switch (alg) {
    case CW_HASH_SKEIN512: e = cw::skein512_launch(d_src, bb, stride, n, g_iv512_512, d_dig, 64, s, lean); break;
    case CW_HASH_SKEIN256_128: e = cw::skein256_launch(d_src, bb, stride, n, g_iv256_128, d_dig, 16, s, lean); break;
    case CW_HASH_SHA256: e = cw::sha256_launch(d_src, bb, stride, n, d_dig, s); break;
    case CW_HASH_NONE: return CW_OK;
    default: return fail(CW_ERR_BAD_ARG, "unknown hash algorithm %d", alg);
    }
    if (e != hipSuccess) return fail(CW_ERR_HIP, "hash launch: %s", hipGetErrorString(e));
    return CW_OK;
}

int dev_compress(int alg, const uint8_t *d_src, size_t bb, size_t stride, size_t n, uint8_t *d_dst, size_t dst_stride,
                 uint32_t *d_sizes, hipStream_t s)
{
    hipError_t e;
    if (alg == CW_COMP_NONE) return CW_OK;
    if (alg != CW_COMP_LZ4 && alg != CW_COMP_LZF) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", alg);
    if (bb == 0) return fail(CW_ERR_BAD_ARG, "compression needs block_bytes > 0");
    if (dst_stride < cw_compress_bound(alg, bb))
        return fail(CW_ERR_BAD_ARG, "dst_stride %zu < bound %zu", dst_stride, cw_compress_bound(alg, bb));
    ProfScope prof(PROF_CODEC, s);
    e = alg == CW_COMP_LZ4 ? cw::lz4_launch(d_src, bb, stride, n, d_dst, dst_stride, d_sizes, s)
                           : cw::lzf_launch(d_src, bb, stride, n, d_dst, dst_stride, d_sizes, s);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "compress launch: %s", hipGetErrorString(e));
    return CW_OK;
}

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "libcwhc: %s failed: %s\n", what, t_err);
    abort();
}

} // namespace

extern "C" {

// ---- lifecycle ------------------------------------------------------------------------------------
int cw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cw_init(int device)
{
    std::lock_guard<std::mutex> g(g_lock);
    if (g_device.load() >= 0) return CW_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(CW_ERR_NO_DEVICE, "no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(CW_ERR_BAD_ARG, "device %d out of range (count %d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CW_ERR_NO_DEVICE, "device %d is %s; libcwhc is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    cw::skein_compute_iv(8, 512, &g_iv512_512);
    cw::skein_compute_iv(4, 128, &g_iv256_128);
    g_device.store(device, std::memory_order_release);
    return CW_OK;
}

void cw_shutdown(void)
{
    cw_offload_thread_stop();
    std::lock_guard<std::mutex> g(g_lock);
    if (g_device.load() < 0) return;
    (void)hipDeviceSynchronize();
    cw::skein_release_workspaces();
    cw::lz4_release_workspaces();
    cw::lzf_release_workspaces();
    cw::pack_release_workspaces();
    g_device.store(-1);
}

const char *cw_last_error(void) { return t_err; }
const char *cw_version(void) { return "compute-war_amd 0.1 (gfx950)"; }

size_t cw_digest_bytes(int hash_alg)
{
    switch (hash_alg) {
    case CW_HASH_SKEIN512: return 64;
    case CW_HASH_SKEIN256_128: return 16;
    case CW_HASH_SHA256: return 32;
    default: return 0;
    }
}

size_t cw_compress_bound(int comp_alg, size_t l)
{
    if (comp_alg == CW_COMP_LZ4) return l + l / 255 + 16; // LZ4_compressBound (lz4.h:156); the reference's 2*l covers it for l >= 17
    if (comp_alg == CW_COMP_LZF) return l ? l : 1; // out_len = l-1 usable (:346)
    return 0;
}

void cw_set_block_size(size_t b) { g_block_size.store(b); }
size_t cw_get_block_size(void) { return g_block_size.load(); }

// ---- device-resident API ------------------------------------------------------------------------------
int cw_dev_hash(int hash_alg, const void *d_src, size_t block_bytes, size_t src_stride, size_t nblocks, void *d_digests,
                void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if (!d_src || !d_digests) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if ((rc = check_block(block_bytes)) != CW_OK) return rc;
    if (src_stride < block_bytes) return fail(CW_ERR_BAD_ARG, "src_stride < block_bytes");
    // long Skein messages in sliced launches here too: alone they are as fast as the one-launch kernel (46.6 ms per Mi blocks
    // either way, slightly ahead on small batches), and the hot kernel is then the same with and without a codec beside it
    return dev_hash(hash_alg, (const uint8_t *)d_src, block_bytes, src_stride, nblocks, (uint8_t *)d_digests, (hipStream_t)stream, false, true);
}

int cw_dev_compress(int comp_alg, const void *d_src, size_t block_bytes, size_t src_stride, size_t nblocks, void *d_dst,
                    size_t dst_stride, uint32_t *d_sizes, void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if (!d_src || !d_dst || !d_sizes) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if ((rc = check_block(block_bytes)) != CW_OK) return rc;
    if (src_stride < block_bytes) return fail(CW_ERR_BAD_ARG, "src_stride < block_bytes");
    return dev_compress(comp_alg, (const uint8_t *)d_src, block_bytes, src_stride, nblocks, (uint8_t *)d_dst, dst_stride, d_sizes,
                        (hipStream_t)stream);
}

int cw_dev_hash_and_compress(int hash_alg, int comp_alg, const void *d_src, size_t block_bytes, size_t src_stride,
                             size_t nblocks, void *d_digests, void *d_dst, size_t dst_stride, uint32_t *d_sizes, void *stream)
{
    // ProcessBlock (:243-257) compresses, then hashes; the two only share their read-only input, so they run
    // side by side: the codec (latency/memory bound, few issue slots, persistent grid) goes first on the
    // caller's stream so its workgroups are resident before the hash (pure integer VALU, one long-lived
    // wavefront per 64 blocks) fills the rest of every CU from a low-priority side stream; the side stream
    // is joined before this call's work counts as done.
    if (comp_alg == CW_COMP_NONE) return cw_dev_hash(hash_alg, d_src, block_bytes, src_stride, nblocks, d_digests, stream);
    if (hash_alg == CW_HASH_NONE)
        return cw_dev_compress(comp_alg, d_src, block_bytes, src_stride, nblocks, d_dst, dst_stride, d_sizes, stream);
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    static const char *serial = getenv("CW_SERIAL"); // CW_SERIAL=1: both kernels on the caller's stream (profiling knob)
    if (serial && serial[0] == '1') {
        rc = cw_dev_compress(comp_alg, d_src, block_bytes, src_stride, nblocks, d_dst, dst_stride, d_sizes, stream);
        return rc == CW_OK ? cw_dev_hash(hash_alg, d_src, block_bytes, src_stride, nblocks, d_digests, stream) : rc;
    }
    if ((rc = t_side.open()) != CW_OK) return rc;
    hipStream_t main_s = (hipStream_t)stream;
    HIP_TRY(hipEventRecord(t_side.fork, main_s));
    HIP_TRY(hipStreamWaitEvent(t_side.side, t_side.fork, 0));
    rc = cw_dev_compress(comp_alg, d_src, block_bytes, src_stride, nblocks, d_dst, dst_stride, d_sizes, stream);
    if (rc == CW_OK) {
        if (!d_digests) rc = fail(CW_ERR_BAD_ARG, "NULL device pointer");
        else rc = dev_hash(hash_alg, (const uint8_t *)d_src, block_bytes, src_stride ? src_stride : block_bytes, nblocks, (uint8_t *)d_digests,
                           t_side.side, false, true);
    }
    HIP_TRY(hipEventRecord(t_side.join, t_side.side));
    HIP_TRY(hipStreamWaitEvent(main_s, t_side.join, 0));
    return rc;
}

int cw_dev_decompress(int comp_alg, const void *d_comp, size_t comp_stride, const uint32_t *d_sizes, size_t nblocks, void *d_dst,
                      size_t block_bytes, uint32_t *d_status, void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if (!d_comp || !d_sizes || !d_dst || !d_status) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if (comp_alg != CW_COMP_LZ4 && comp_alg != CW_COMP_LZF) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", comp_alg);
    if (block_bytes == 0 || (rc = check_block(block_bytes)) != CW_OK) return rc ? rc : fail(CW_ERR_BAD_ARG, "block_bytes == 0");
    hipError_t e = cw::decompress_launch(comp_alg == CW_COMP_LZ4 ? 0 : 1, (const uint8_t *)d_comp, comp_stride, d_sizes, nblocks,
                                         (uint8_t *)d_dst, block_bytes, d_status, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "decompress launch: %s", hipGetErrorString(e));
    return CW_OK;
}

int cw_dev_hash_tree(int hash_alg, const void *d_src, size_t block_bytes, size_t src_stride, size_t nblocks, unsigned leaf, unsigned node,
                     unsigned max_level, void *d_digests, void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if (!d_src || !d_digests) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if ((rc = check_block(block_bytes)) != CW_OK) return rc;
    if (hash_alg != CW_HASH_SKEIN512 && hash_alg != CW_HASH_SKEIN256_128) return fail(CW_ERR_BAD_ARG, "tree hashing is defined for the Skein algorithms only");
    hipError_t e = cw::skein_tree_launch(hash_alg == CW_HASH_SKEIN512 ? 8 : 4, (const uint8_t *)d_src, block_bytes,
                                         src_stride ? src_stride : block_bytes, nblocks, hash_alg == CW_HASH_SKEIN512 ? 512u : 128u, leaf,
                                         node, max_level, (uint8_t *)d_digests, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(CW_ERR_BAD_ARG, "tree parameters leaf=%u node=%u maxLevel=%u not usable for %zu-byte blocks", leaf, node, max_level, block_bytes);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "tree hash launch: %s", hipGetErrorString(e));
    return CW_OK;
}

int cw_dev_pack(const void *d_slots, size_t slot_stride, const uint32_t *d_sizes, size_t nblocks, void *d_packed, uint64_t *d_offsets,
                void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (!d_offsets || (nblocks && (!d_sizes || (d_packed && !d_slots)))) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    hipError_t e = cw::pack_launch((const uint8_t *)d_slots, slot_stride, d_sizes, nblocks, (uint8_t *)d_packed, d_offsets,
                                   (hipStream_t)stream);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "pack launch: %s", hipGetErrorString(e));
    return CW_OK;
}

int cw_dev_gen_random(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes, void *d_dst, void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (!d_dst) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if (block_bytes % 16 || block_bytes > CW_MAX_BLOCK_BYTES) return fail(CW_ERR_BAD_ARG, "block_bytes must be a multiple of 16, <= 65536");
    hipError_t e = cw::gen_random_launch(seed, first_block, nblocks, block_bytes, (uint8_t *)d_dst, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "gen launch: %s", hipGetErrorString(e));
    return CW_OK;
}

int cw_dev_sum_sizes(const uint32_t *d_sizes, size_t nblocks, uint32_t raw_bytes, uint64_t *d_totals, void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (!d_sizes || !d_totals) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    hipError_t e = cw::sum_sizes_launch(d_sizes, nblocks, raw_bytes, d_totals, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "sum launch: %s", hipGetErrorString(e));
    return CW_OK;
}

void cw_profile_enable(int on) { t_prof_on = on != 0; }

int cw_profile_read(double ms_sum[3], unsigned count[3], int reset)
{
    for (int k = 0; k < PROF_KINDS; k++) { ms_sum[k] = 0; count[k] = 0; }
    for (ProfSpan &p : t_prof) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(p.b));
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        ms_sum[p.kind] += ms;
        count[p.kind]++;
    }
    if (reset) {
        for (ProfSpan &p : t_prof) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        t_prof.clear();
    }
    return CW_OK;
}

// ---- batched host API ------------------------------------------------------------------------------------
int cw_hash_and_compress_blocks(int hash_alg, int comp_alg, const void *src, size_t block_bytes, size_t nblocks, void *digests,
                                void *dst, size_t dst_stride, uint32_t *sizes)
{
    ThreadCtx &c = t_ctx;
    int rc = c.open();
    if (rc != CW_OK) return rc;
    const bool do_hash = hash_alg != CW_HASH_NONE && digests != nullptr;
    const bool do_comp = comp_alg != CW_COMP_NONE && dst != nullptr && sizes != nullptr;
    if (nblocks == 0 || (!do_hash && !do_comp)) return CW_OK;
    if (!src && block_bytes) return fail(CW_ERR_BAD_ARG, "NULL src");
    if ((rc = check_block(block_bytes)) != CW_OK) return rc;
    const size_t db = cw_digest_bytes(hash_alg);
    if (do_hash && db == 0) return fail(CW_ERR_BAD_ARG, "unknown hash algorithm %d", hash_alg);
    const size_t bound = do_comp ? cw_compress_bound(comp_alg, block_bytes) : 0;
    if (do_comp && bound == 0) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", comp_alg);
    // The caller's slot may be the reference's (2*l for lz4, l-1 for lzf, :234-239); device slots use the bound.
    const size_t d_stride = (bound + 15) & ~(size_t)15;

    const size_t per_block = block_bytes + (do_comp ? 2 * d_stride + 8 : 0) + 64; // input, slots, packed stream, digest/size/offset
    size_t chunk = kMaxChunkBytes / (per_block ? per_block : 1);
    if (chunk == 0) chunk = 1;
    if (chunk > nblocks) chunk = nblocks;

    if ((rc = c.src.reserve(chunk * block_bytes + 16)) != CW_OK) return rc;
    if (do_hash && (rc = c.dig.reserve(chunk * db)) != CW_OK) return rc;
    if (do_comp && ((rc = c.dst.reserve(chunk * d_stride)) != CW_OK || (rc = c.sizes.reserve(chunk * 4)) != CW_OK ||
                    (rc = c.pack.reserve(chunk * d_stride)) != CW_OK || (rc = c.offs.reserve((chunk + 1) * 8)) != CW_OK))
        return rc;

    for (size_t first = 0; first < nblocks; first += chunk) {
        const size_t n = nblocks - first < chunk ? nblocks - first : chunk;
        const uint8_t *h_src = (const uint8_t *)src + first * block_bytes;
        if (block_bytes) HIP_TRY(hipMemcpyAsync(c.src.p, h_src, n * block_bytes, hipMemcpyHostToDevice, c.stream));
        if (do_comp) {
            rc = dev_compress(comp_alg, (const uint8_t *)c.src.p, block_bytes, block_bytes, n, (uint8_t *)c.dst.p, d_stride,
                              (uint32_t *)c.sizes.p, c.stream);
            if (rc != CW_OK) return rc;
        }
        if (do_hash) {
            rc = dev_hash(hash_alg, (const uint8_t *)c.src.p, block_bytes, block_bytes, n, (uint8_t *)c.dig.p, c.stream);
            if (rc != CW_OK) return rc;
            HIP_TRY(hipMemcpyAsync((uint8_t *)digests + first * db, c.dig.p, n * db, hipMemcpyDeviceToHost, c.stream));
        }
        if (do_comp) {
            // the slots are packed into one stream on the device (pack_kernels.hip) so that the payload of the whole
            // batch crosses the bus in one copy; the caller's slots (which may be smaller than the bound) are filled
            // from the pinned staging buffer
            hipError_t pe = cw::pack_launch((const uint8_t *)c.dst.p, d_stride, (const uint32_t *)c.sizes.p, n, (uint8_t *)c.pack.p,
                                            (uint64_t *)c.offs.p, c.stream);
            if (pe != hipSuccess) return fail(CW_ERR_HIP, "pack launch: %s", hipGetErrorString(pe));
            HIP_TRY(hipMemcpyAsync(sizes + first, c.sizes.p, n * 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipStreamSynchronize(c.stream));
            size_t total = 0;
            for (size_t i = 0; i < n; i++) {
                const uint32_t sz = sizes[first + i];
                if (sz > dst_stride) return fail(CW_ERR_BAD_ARG, "block %zu: %u bytes exceed dst_stride %zu", first + i, sz, dst_stride);
                total += sz;
            }
            if (total) {
                if ((rc = c.hpack.reserve(total)) != CW_OK) return rc;
                HIP_TRY(hipMemcpyAsync(c.hpack.p, c.pack.p, total, hipMemcpyDeviceToHost, c.stream));
                HIP_TRY(hipStreamSynchronize(c.stream));
                const uint8_t *from = (const uint8_t *)c.hpack.p;
                for (size_t i = 0; i < n; i++) {
                    const uint32_t sz = sizes[first + i];
                    memcpy((uint8_t *)dst + (first + i) * dst_stride, from, sz);
                    from += sz;
                }
            }
        }
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    return CW_OK;
}

int cw_hash_tree_blocks(int hash_alg, const void *src, size_t block_bytes, size_t nblocks, unsigned leaf, unsigned node, unsigned max_level,
                        void *digests)
{
    ThreadCtx &c = t_ctx;
    int rc = c.open();
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if ((!src && block_bytes) || !digests) return fail(CW_ERR_BAD_ARG, "NULL pointer");
    if ((rc = check_block(block_bytes)) != CW_OK) return rc;
    const size_t db = cw_digest_bytes(hash_alg);
    if (db == 0) return fail(CW_ERR_BAD_ARG, "unknown hash algorithm %d", hash_alg);
    size_t chunk = kMaxChunkBytes / (block_bytes + 64);
    if (chunk == 0) chunk = 1;
    if (chunk > nblocks) chunk = nblocks;
    if ((rc = c.src.reserve(chunk * block_bytes + 16)) != CW_OK || (rc = c.dig.reserve(chunk * db)) != CW_OK) return rc;
    for (size_t first = 0; first < nblocks; first += chunk) {
        const size_t n = nblocks - first < chunk ? nblocks - first : chunk;
        if (block_bytes) HIP_TRY(hipMemcpyAsync(c.src.p, (const uint8_t *)src + first * block_bytes, n * block_bytes, hipMemcpyHostToDevice, c.stream));
        if ((rc = cw_dev_hash_tree(hash_alg, c.src.p, block_bytes, block_bytes, n, leaf, node, max_level, c.dig.p, c.stream)) != CW_OK) return rc;
        HIP_TRY(hipMemcpyAsync((uint8_t *)digests + first * db, c.dig.p, n * db, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    return CW_OK;
}

int cw_hash_blocks(int hash_alg, const void *src, size_t block_bytes, size_t nblocks, void *digests)
{
    if (!digests && nblocks) return fail(CW_ERR_BAD_ARG, "NULL digests");
    if (hash_alg == CW_HASH_NONE || cw_digest_bytes(hash_alg) == 0) return fail(CW_ERR_BAD_ARG, "unknown hash algorithm %d", hash_alg);
    return cw_hash_and_compress_blocks(hash_alg, CW_COMP_NONE, src, block_bytes, nblocks, digests, nullptr, 0, nullptr);
}

int cw_compress_blocks(int comp_alg, const void *src, size_t block_bytes, size_t nblocks, void *dst, size_t dst_stride,
                       uint32_t *sizes)
{
    if ((!dst || !sizes) && nblocks) return fail(CW_ERR_BAD_ARG, "NULL dst/sizes");
    if (comp_alg != CW_COMP_LZ4 && comp_alg != CW_COMP_LZF) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", comp_alg);
    return cw_hash_and_compress_blocks(CW_HASH_NONE, comp_alg, src, block_bytes, nblocks, nullptr, dst, dst_stride, sizes);
}

int cw_decompress_blocks(int comp_alg, const void *comp, size_t comp_stride, const uint32_t *sizes, size_t nblocks, void *dst,
                         size_t block_bytes, uint32_t *status)
{
    ThreadCtx &c = t_ctx;
    int rc = c.open();
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if (!comp || !sizes || !dst || !status) return fail(CW_ERR_BAD_ARG, "NULL pointer");
    if (comp_alg != CW_COMP_LZ4 && comp_alg != CW_COMP_LZF) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", comp_alg);
    if (block_bytes == 0 || (rc = check_block(block_bytes)) != CW_OK) return rc ? rc : fail(CW_ERR_BAD_ARG, "block_bytes == 0");
    const size_t d_stride = (comp_stride + 15) & ~(size_t)15;
    size_t chunk = kMaxChunkBytes / (d_stride + block_bytes + 8);
    if (chunk == 0) chunk = 1;
    if (chunk > nblocks) chunk = nblocks;
    if ((rc = c.src.reserve(chunk * d_stride + 16)) != CW_OK || (rc = c.dst.reserve(chunk * block_bytes)) != CW_OK ||
        (rc = c.sizes.reserve(chunk * 4)) != CW_OK || (rc = c.dig.reserve(chunk * 4)) != CW_OK)
        return rc;
    for (size_t first = 0; first < nblocks; first += chunk) {
        const size_t n = nblocks - first < chunk ? nblocks - first : chunk;
        for (size_t i = 0; i < n; i++) { // only the bytes each slot holds cross the bus
            const uint32_t sz = sizes[first + i];
            if (sz > comp_stride) return fail(CW_ERR_BAD_ARG, "block %zu: %u bytes exceed comp_stride %zu", first + i, sz, comp_stride);
            if (sz) HIP_TRY(hipMemcpyAsync((uint8_t *)c.src.p + i * d_stride, (const uint8_t *)comp + (first + i) * comp_stride, sz,
                                           hipMemcpyHostToDevice, c.stream));
        }
        HIP_TRY(hipMemcpyAsync(c.sizes.p, sizes + first, n * 4, hipMemcpyHostToDevice, c.stream));
        rc = cw_dev_decompress(comp_alg, c.src.p, d_stride, (const uint32_t *)c.sizes.p, n, c.dst.p, block_bytes, (uint32_t *)c.dig.p,
                               c.stream);
        if (rc != CW_OK) return rc;
        HIP_TRY(hipMemcpyAsync((uint8_t *)dst + first * block_bytes, c.dst.p, n * block_bytes, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(status + first, c.dig.p, n * 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    return CW_OK;
}

// ---- the reference's slots ----------------------------------------------------------------------------------
void cw_hash_skein(const char *src, char *dst, int count)
{
    if (count > 0 && cw_hash_blocks(CW_HASH_SKEIN256_128, src, cw_get_block_size(), (size_t)count, dst) != CW_OK) die("cw_hash_skein");
}

void cw_hash_skein512(const char *src, char *dst, int count)
{
    if (count > 0 && cw_hash_blocks(CW_HASH_SKEIN512, src, cw_get_block_size(), (size_t)count, dst) != CW_OK) die("cw_hash_skein512");
}

void cw_hash_sha256mb(const char *src, char *dst, int count)
{
    if (count > 0 && cw_hash_blocks(CW_HASH_SHA256, src, cw_get_block_size(), (size_t)count, dst) != CW_OK) die("cw_hash_sha256mb");
}

size_t cw_compress_lz4(const char *src, char *dst, size_t len)
{
    uint32_t sz = 0;
    if (len == 0) return 0;
    // the reference's caller provides 2*len (:353); output that cannot fit it reports 0 like LZ4 would
    int rc = cw_compress_blocks(CW_COMP_LZ4, src, len, 1, dst, 2 * len, &sz);
    if (rc == CW_ERR_BAD_ARG && strstr(t_err, "exceed dst_stride")) return 0;
    if (rc != CW_OK) die("cw_compress_lz4");
    return sz;
}

size_t cw_compress_lzf(const char *src, char *dst, size_t len)
{
    uint32_t sz = 0;
    if (len < 2) return 0; // lzf_compress(in, len, out, len-1) cannot fit anything
    if (cw_compress_blocks(CW_COMP_LZF, src, len, 1, dst, len - 1, &sz) != CW_OK) die("cw_compress_lzf");
    return sz;
}

int cw_decompress_lz4(const char *src, char *dst, int csize, int dst_cap)
{
    const size_t bb = cw_get_block_size();
    uint32_t sz = csize > 0 ? (uint32_t)csize : 0, st = 1;
    if (csize <= 0 || dst_cap < 0 || (size_t)dst_cap < bb) return -1;
    if (cw_decompress_blocks(CW_COMP_LZ4, src, sz, &sz, 1, dst, bb, &st) != CW_OK) die("cw_decompress_lz4");
    return st == 0 ? (int)bb : -1;
}

unsigned cw_decompress_lzf(const void *src, unsigned csize, void *dst, unsigned dst_cap)
{
    const size_t bb = cw_get_block_size();
    uint32_t sz = csize, st = 1;
    if (csize == 0 || dst_cap < bb) return 0;
    if (cw_decompress_blocks(CW_COMP_LZF, src, sz, &sz, 1, dst, bb, &st) != CW_OK) die("cw_decompress_lzf");
    return st == 0 ? (unsigned)bb : 0;
}

// ---- HashOffload -------------------------------------------------------------------------------------------------
struct cw_offload {
    int hash_alg;
    int n_blocks;           // offloadCount
    size_t block_bytes;
    char *data = nullptr;   // host
    char *results = nullptr;
    void (*on_complete)(void *) = nullptr;
    void *arg = nullptr;
    std::atomic<int> state{CW_OFFLOAD_INIT};
    hipStream_t stream = nullptr;
    DevBuf d_src, d_dig;
};

cw_offload_t *cw_offload_create(int hash_alg, int n_blocks, size_t block_bytes)
{
    if (ensure_init() != CW_OK) return nullptr;
    if (n_blocks <= 0 || cw_digest_bytes(hash_alg) == 0 || check_block(block_bytes) != CW_OK) {
        fail(CW_ERR_BAD_ARG, "cw_offload_create: bad arguments");
        return nullptr;
    }
    cw_offload *h = new cw_offload;
    h->hash_alg = hash_alg; h->n_blocks = n_blocks; h->block_bytes = block_bytes;
    if (hipSetDevice(g_device.load()) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        h->d_src.reserve((size_t)n_blocks * block_bytes + 16) != CW_OK || h->d_dig.reserve((size_t)n_blocks * cw_digest_bytes(hash_alg)) != CW_OK) {
        cw_offload_destroy(h);
        fail(CW_ERR_HIP, "cw_offload_create: device resources");
        return nullptr;
    }
    return h;
}

void cw_offload_destroy(cw_offload_t *h)
{
    if (!h) return;
    h->d_src.release(); h->d_dig.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int cw_offload_reset(cw_offload_t *h, char *data, char *results, void (*on_complete)(void *), void *arg)
{
    if (!h) return fail(CW_ERR_BAD_ARG, "NULL offload");
    h->data = data; h->results = results; h->on_complete = on_complete; h->arg = arg;
    h->state.store(CW_OFFLOAD_INIT);
    return CW_OK;
}

int cw_offload_enqueue(cw_offload_t *h)
{
    if (!h) return fail(CW_ERR_BAD_ARG, "NULL offload");
    int want = CW_OFFLOAD_INIT;
    if (!h->state.compare_exchange_strong(want, CW_OFFLOAD_QUEUED)) return fail(CW_ERR_STATE, "Enqueue: state %d != hInit", want);
    return CW_OK;
}

int cw_offload_start(cw_offload_t *h)
{
    if (!h) return fail(CW_ERR_BAD_ARG, "NULL offload");
    int want = CW_OFFLOAD_QUEUED;
    if (!h->state.compare_exchange_strong(want, CW_OFFLOAD_OFFLOADED)) return fail(CW_ERR_STATE, "Start: state %d != hQueued", want);
    if (!h->data || !h->results) return fail(CW_ERR_BAD_ARG, "Start: Reset() gave no data/results");
    const size_t bytes = (size_t)h->n_blocks * h->block_bytes, db = cw_digest_bytes(h->hash_alg);
    HIP_TRY(hipSetDevice(g_device.load()));
    if (bytes) HIP_TRY(hipMemcpyAsync(h->d_src.p, h->data, bytes, hipMemcpyHostToDevice, h->stream));
    int rc = dev_hash(h->hash_alg, (const uint8_t *)h->d_src.p, h->block_bytes, h->block_bytes, (size_t)h->n_blocks, (uint8_t *)h->d_dig.p, h->stream);
    if (rc != CW_OK) return rc;
    HIP_TRY(hipMemcpyAsync(h->results, h->d_dig.p, (size_t)h->n_blocks * db, hipMemcpyDeviceToHost, h->stream));
    return CW_OK;
}

int cw_offload_complete(cw_offload_t *h)
{
    if (!h) return fail(CW_ERR_BAD_ARG, "NULL offload");
    if (h->state.load() != CW_OFFLOAD_OFFLOADED) return fail(CW_ERR_STATE, "Complete: state %d != hOffloaded", h->state.load());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->state.store(CW_OFFLOAD_COMPLETE);
    if (h->on_complete) h->on_complete(h->arg);
    return CW_OK;
}

int cw_offload_completed(const cw_offload_t *h) { return h && h->state.load() == CW_OFFLOAD_COMPLETE; }
int cw_offload_state(const cw_offload_t *h) { return h ? h->state.load() : CW_ERR_BAD_ARG; }

int cw_offload_do(cw_offload_t *h)
{
    int rc = cw_offload_start(h);
    return rc == CW_OK ? cw_offload_complete(h) : rc;
}

// ---- the offload thread (:160-183) ---------------------------------------------------------------------------
namespace {
std::mutex q_lock;               // hashLock
std::condition_variable q_cv;    // hashCV
std::deque<cw_offload *> q_work; // hashQueue
bool q_finished = false;         // allWorkFinished
std::thread q_thread;
bool q_running = false;

void offload_entry_point()
{
    std::unique_lock<std::mutex> lk(q_lock);
    for (;;) {
        if (q_work.empty()) {
            if (q_finished) break; // drain before exiting
            q_cv.wait(lk);
            continue;
        }
        cw_offload *h = q_work.front();
        q_work.pop_front();
        lk.unlock();
        if (cw_offload_do(h) != CW_OK) fprintf(stderr, "libcwhc: offload failed: %s\n", t_err);
        lk.lock();
    }
}
} // namespace

int cw_offload_thread_start(void)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    std::lock_guard<std::mutex> g(q_lock);
    if (q_running) return CW_OK;
    q_finished = false;
    q_thread = std::thread(offload_entry_point);
    q_running = true;
    return CW_OK;
}

int cw_offload_submit(cw_offload_t *h)
{
    int rc = cw_offload_enqueue(h);
    if (rc != CW_OK) return rc;
    {
        std::lock_guard<std::mutex> g(q_lock);
        if (!q_running) return fail(CW_ERR_STATE, "offload thread not started");
        q_work.push_back(h);
    }
    q_cv.notify_one();
    return CW_OK;
}

void cw_offload_thread_stop(void)
{
    {
        std::lock_guard<std::mutex> g(q_lock);
        if (!q_running) return;
        q_finished = true;
    }
    q_cv.notify_all();
    q_thread.join();
    std::lock_guard<std::mutex> g(q_lock);
    q_running = false;
}

} // extern "C"
