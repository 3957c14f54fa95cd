// cw_api.hip -- host side of libcwhc.so: the C ABI of include/cw_hashcompress.h over the HIP kernels.
//
// Structure: a set of initialised devices (cw_init = the reference's empty initializeGpu(),
// src/hashandcompress/HashAndCompress.cpp:95-98) with one "current device" per calling host thread (cw_set_device;
// threads that never choose use the first initialised device), one lazily created context per (thread, device) --
// own HIP streams + growable device/pinned staging buffers, because the reference invokes its slots from --c-threads
// workers with no locking, :398-402 --, the pipelined host batch path (what HashOffload::Start()/Complete() were meant
// to be, HashOffload.h:26-40), the HashOffload batch object itself (HashOffload.h:13-64) and the single consumer
// thread that drains it (hashing_offload_entry_point, :160-183).
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
#include <memory>
#include <map>
#include <mutex>
#include <string>
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

// ---- devices ----------------------------------------------------------------------------------------
constexpr int kMaxDevices = 16;
std::mutex g_lock;
std::atomic<int> g_default{-1};        // first initialised device: what threads use that never chose one
std::atomic<uint32_t> g_mask{0};       // initialised devices
thread_local int t_device = -1;        // the calling thread's device (cw_init / cw_set_device), -1 = g_default
cw::SkeinIV g_iv512_512, g_iv256_128;  // host-computed, device independent
std::atomic<size_t> g_block_size{4096}; // the reference's global blockSize (:89)

int current_device()
{
    const int d = t_device;
    return d >= 0 && ((g_mask.load(std::memory_order_acquire) >> d) & 1u) ? d : g_default.load(std::memory_order_acquire);
}

// every entry point: initialise on first use and make the library's device the calling thread's HIP device
int ensure_init()
{
    int d = current_device();
    if (d < 0) {
        const int rc = cw_init(t_device >= 0 ? t_device : 0);
        if (rc != CW_OK) return rc;
        d = current_device();
    }
    HIP_TRY(hipSetDevice(d));
    return CW_OK;
}

// ---- per-(thread, device) context ---------------------------------------------------------------------
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
struct PinnedBuf { // page-locked host staging: the only kind of host memory a copy engine reads or writes at bus speed
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

// one chunk of the host batch path in flight: device buffers, pinned staging, the events of its three stages
constexpr int kSlots = 3;
struct Slot {
    DevBuf src, dst, pack, sizes, offs, dig;
    PinnedBuf h_src, h_meta, h_pack;
    hipStream_t stream = nullptr, side = nullptr; // this chunk's kernels: codec on `stream`, hash beside it on `side`
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_meta = nullptr, ev_payload = nullptr;
    size_t first = 0, n = 0;
    uint64_t total = 0;
    bool borrowed = false; // stream and side belong to the thread's first slot (CW_HOST_SHARED_STREAMS)
    int open(const Slot *lender = nullptr)
    {
        if (ev_h2d) return CW_OK;
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (lender && lender->stream) {
            stream = lender->stream; side = lender->side; borrowed = true;
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least));
        }
        HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_meta, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_payload, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_h2d, hipEventDisableTiming));
        return CW_OK;
    }
    void release()
    {
        src.release(); dst.release(); pack.release(); sizes.release(); offs.release(); dig.release();
        h_src.release(); h_meta.release(); h_pack.release();
        if (stream && !borrowed) { cw::release_stream_workspaces(stream); (void)hipStreamDestroy(stream); }
        if (side && !borrowed) { cw::release_stream_workspaces(side); (void)hipStreamDestroy(side); }
        borrowed = false;
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        if (ev_meta) (void)hipEventDestroy(ev_meta);
        if (ev_payload) (void)hipEventDestroy(ev_payload);
        if (ev_h2d) (void)hipEventDestroy(ev_h2d);
        stream = side = nullptr; fork = join = ev_meta = ev_payload = ev_h2d = nullptr;
    }
};

// what the last fused LZ4 call of this thread left behind for the next one: the share of its blocks that the scan queued for the parsers
// (copied back asynchronously; read when it has arrived, never waited for)
struct FusedHint { uint32_t *h_queued = nullptr; hipEvent_t ev = nullptr; bool pending = false; size_t blocks_of_copy = 0; float queued_share = 0.f; };

struct ThreadCtx {
    int device = -1;
    FusedHint hint;
    hipStream_t stream = nullptr;            // kernels of the host-buffer entry points
    hipStream_t s_h2d = nullptr, s_d2h = nullptr; // the two copy directions of the pipelined batch path
    // fork/join for the fused call: the hash runs on `side` beside the codec on the caller's stream
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    DevBuf src, dst, dig, sizes;             // unpipelined helpers (decompress, tree hash)
    Slot slot[kSlots];
    int open(int dev)
    {
        if (stream) return CW_OK;
        device = dev;
        HIP_TRY(hipSetDevice(dev));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s_h2d, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s_d2h, hipStreamNonBlocking));
        int least = 0, greatest = 0; // the hash is the long, ALU-bound kernel: it yields dispatch slots to the codec
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least));
        HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        return CW_OK;
    }
    ~ThreadCtx()
    {
        if (!stream) return;
        (void)hipSetDevice(device);
        src.release(); dst.release(); dig.release(); sizes.release();
        for (Slot &s : slot) s.release();
        (void)hipEventDestroy(fork); (void)hipEventDestroy(join);
        if (hint.ev) (void)hipEventDestroy(hint.ev);
        if (hint.h_queued) (void)hipHostFree(hint.h_queued);
        cw::release_stream_workspaces(side); cw::release_stream_workspaces(stream);
        (void)hipStreamDestroy(side); (void)hipStreamDestroy(s_h2d); (void)hipStreamDestroy(s_d2h); (void)hipStreamDestroy(stream);
    }
};
struct ThreadCtxSet { std::unique_ptr<ThreadCtx> of[kMaxDevices]; };
thread_local ThreadCtxSet t_ctxs;

// the calling thread's context on its current device (created on first use)
int thread_ctx(ThreadCtx **out)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    const int d = current_device();
    std::unique_ptr<ThreadCtx> &c = t_ctxs.of[d];
    if (!c) c.reset(new ThreadCtx);
    if ((rc = c->open(d)) != CW_OK) return rc;
    *out = c.get();
    return CW_OK;
}

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
    const char *slice_env = cw::tune("CW_SKEIN_SLICED"); // CW_SKEIN_SLICED=0: always the one-launch hash kernel (profiling knob)
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
                 uint32_t *d_sizes, hipStream_t s, const cw::AfterScan *after_scan = nullptr)
{
    hipError_t e;
    if (alg == CW_COMP_NONE) return CW_OK;
    if (alg != CW_COMP_LZ4 && alg != CW_COMP_LZF) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", alg);
    if (bb == 0) return fail(CW_ERR_BAD_ARG, "compression needs block_bytes > 0");
    if (dst_stride < cw_compress_bound(alg, bb))
        return fail(CW_ERR_BAD_ARG, "dst_stride %zu < bound %zu", dst_stride, cw_compress_bound(alg, bb));
    ProfScope prof(PROF_CODEC, s);
    e = alg == CW_COMP_LZ4 ? cw::lz4_launch(d_src, bb, stride, n, d_dst, dst_stride, d_sizes, s, after_scan)
                           : cw::lzf_launch(d_src, bb, stride, n, d_dst, dst_stride, d_sizes, s);
    if (e != hipSuccess) return fail(CW_ERR_HIP, "compress launch: %s", hipGetErrorString(e));
    return CW_OK;
}

thread_local char t_kernels[2][320] = {"", ""};

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "libcwhc: %s failed: %s\n", what, t_err);
    abort();
}

} // namespace

// Tuning and test knobs.  A knob's value is what cw_tune_set gave it, else the environment variable of the same name (read once),
// else unset.  Every launch function asks per call, so a test can run two settings in one process.
namespace {
std::mutex tune_lock;
std::map<std::string, std::string> tune_over;                 // cw_tune_set
std::map<std::string, std::pair<bool, std::string>> tune_env; // getenv, cached (present?, value)
} // namespace

const char *cw::tune(const char *key)
{
    std::lock_guard<std::mutex> g(tune_lock);
    auto o = tune_over.find(key);
    if (o != tune_over.end()) return o->second.c_str();
    auto e = tune_env.find(key);
    if (e == tune_env.end()) {
        const char *v = getenv(key);
        e = tune_env.emplace(key, std::make_pair(v != nullptr, std::string(v ? v : ""))).first;
    }
    return e->second.first ? e->second.second.c_str() : nullptr;
}

void cw::release_stream_workspaces(hipStream_t stream)
{
    (void)hipStreamSynchronize(stream); // nothing of the stream's may still use what is freed here
    cw::lz4_release_stream(stream);
    cw::lzf_release_stream(stream);
    cw::pack_release_stream(stream);
    cw::skein_release_stream(stream);
}

void cw::note_kernels(int kind, const char *names)
{
    if (kind < 0 || kind > 1) return;
    strncpy(t_kernels[kind], names, sizeof t_kernels[kind] - 1);
    t_kernels[kind][sizeof t_kernels[kind] - 1] = 0;
}

extern "C" {

// ---- knobs (CW_TESTING section of the header) -------------------------------------------------------
int cw_tune_set(const char *key, const char *value)
{
    if (!key || strncmp(key, "CW_", 3) != 0) return fail(CW_ERR_BAD_ARG, "knob names start with CW_");
    std::lock_guard<std::mutex> g(tune_lock);
    if (value) tune_over[key] = value;
    else tune_over.erase(key);
    return CW_OK;
}
void cw_tune_reset(void)
{
    std::lock_guard<std::mutex> g(tune_lock);
    tune_over.clear();
}

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
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(CW_ERR_NO_DEVICE, "no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(CW_ERR_BAD_ARG, "device %d out of range (count %d)", device, n);
    if (!((g_mask.load() >> device) & 1u)) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(CW_ERR_NO_DEVICE, "device %d is %s; libcwhc is built for gfx950 only", device, prop.gcnArchName);
        if (g_mask.load() == 0) {
            cw::skein_compute_iv(8, 512, &g_iv512_512);
            cw::skein_compute_iv(4, 128, &g_iv256_128);
        }
        g_mask.fetch_or(1u << device, std::memory_order_release);
        if (g_default.load() < 0) g_default.store(device, std::memory_order_release);
    }
    HIP_TRY(hipSetDevice(device));
    t_device = device; // the calling thread works on this device from now on
    return CW_OK;
}

int cw_set_device(int device) { return cw_init(device); }
int cw_get_device(void) { return current_device(); }

void cw_shutdown(void)
{
    cw_offload_thread_stop();
    std::lock_guard<std::mutex> g(g_lock);
    const uint32_t mask = g_mask.load();
    if (!mask) return;
    for (int d = 0; d < kMaxDevices; d++)
        if ((mask >> d) & 1u) {
            (void)hipSetDevice(d);
            (void)hipDeviceSynchronize();
        }
    cw::skein_release_workspaces();
    cw::lz4_release_workspaces();
    cw::lzf_release_workspaces();
    cw::pack_release_workspaces();
    g_mask.store(0);
    g_default.store(-1);
    t_device = -1;
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

// ProcessBlock (:243-257) compresses, then hashes; the two only share their read-only input, so they run side by
// side: the codec (latency/memory bound, few issue slots, persistent grid) goes first on the caller's stream so its
// workgroups are resident before the hash (pure integer VALU, one long-lived wavefront per 64 blocks) fills the rest of
// every CU from the context's low-priority side stream; the side stream is joined before the call's work counts as done.
static int dev_fused(hipStream_t side, hipEvent_t fork, hipEvent_t join, int hash_alg, int comp_alg, const uint8_t *d_src, size_t block_bytes, size_t src_stride, size_t nblocks,
                     uint8_t *d_digests, uint8_t *d_dst, size_t dst_stride, uint32_t *d_sizes, hipStream_t main_s)
{
    int rc;
    const char *serial = cw::tune("CW_SERIAL"); // CW_SERIAL=1: both kernels on the caller's stream (profiling knob)
    if (serial && serial[0] == '1') {
        rc = dev_compress(comp_alg, d_src, block_bytes, src_stride, nblocks, d_dst, dst_stride, d_sizes, main_s);
        return rc == CW_OK ? dev_hash(hash_alg, d_src, block_bytes, src_stride, nblocks, d_digests, main_s, false, true) : rc;
    }
    HIP_TRY(hipEventRecord(fork, main_s));
    HIP_TRY(hipStreamWaitEvent(side, fork, 0));
    // Side by side pays when the codec is light -- incompressible input, where it is the scan alone (the headline: 61.7 ms against 72.9 one after
    // the other).  On compressible input the parsers fill every CU with wavefronts that stay until the queue is empty; a hash kernel that arrives
    // beside them trickles in behind, ends with the call and costs the parsers more than it takes alone (Skein-256 over 1 Mi blocks of 4 KiB:
    // 3.5 ms alone, but the fused call 97.7 ms against 84.8 one after the other; 51,728 blocks 8.7 against 6.8 ms).  Which input a call has is known
    // on the device only -- but calls come in streams of alike ones, so the LAST LZ4 call's queued share decides (copied back asynchronously, used
    // when it has arrived): a quarter or more of the blocks queued, at least 16 Ki blocks (fewer leave CUs idle for the hash anyway), and blocks of
    // at most 4 KiB => the hash is enqueued right behind the scan, runs beside it, and the PARSERS WAIT FOR IT.  Measured, fused call gated / not:
    // Skein-256 + LZ4, 1 Mi blocks of 4 KiB 93.3 / 104.2 ms, SHA-256 + LZ4 92.2 / 101.6, 51,728 blocks 7.15 / 7.55; blocks of 64 KiB, where the
    // scalar-thread parsers leave the vector ALUs to the hash: 64 Ki blocks 94.8 / 91.1 ms, the mix 43.4 / 42.2, 256 Ki blocks 370.6 / 372.0 -- so
    // those stay side by side.  (Always gating costs the headline 2.7 ms: its empty-queue parser and redo launches then run behind the hash.)
    struct Gate { int hash_alg; const uint8_t *src; size_t bb, stride, n; uint8_t *dig; hipStream_t side, main_s; hipEvent_t join; int rc; bool called; };
    Gate gate = {hash_alg, d_src, block_bytes, src_stride, nblocks, d_digests, side, main_s, join, CW_OK, false};
    const cw::AfterScan hook = {[](void *p) -> hipError_t {
                                    Gate *g = static_cast<Gate *>(p);
                                    g->called = true;
                                    g->rc = dev_hash(g->hash_alg, g->src, g->bb, g->stride, g->n, g->dig, g->side, false, true);
                                    if (g->rc != CW_OK) return hipErrorUnknown;
                                    hipError_t e = hipEventRecord(g->join, g->side);
                                    return e == hipSuccess ? hipStreamWaitEvent(g->main_s, g->join, 0) : e;
                                }, &gate};
    ThreadCtx *tc = nullptr;
    FusedHint *hint = comp_alg == CW_COMP_LZ4 && thread_ctx(&tc) == CW_OK ? &tc->hint : nullptr;
    if (hint && hint->pending && hipEventQuery(hint->ev) == hipSuccess) {
        hint->pending = false;
        if (hint->blocks_of_copy) hint->queued_share = (float)*hint->h_queued / (float)hint->blocks_of_copy;
    }
    (void)hipGetLastError(); // (hipErrorNotReady of the query is not an error)
    const char *gate_env = cw::tune("CW_FUSED_GATE"); // 0 = never, 1 = always (profiling knob)
    const bool gated = hint && (gate_env ? gate_env[0] == '1' : block_bytes <= 4096 && nblocks >= 16384 && hint->queued_share >= 0.25f);
    rc = dev_compress(comp_alg, d_src, block_bytes, src_stride, nblocks, d_dst, dst_stride, d_sizes, main_s, gated ? &hook : nullptr);
    if (gate.called && gate.rc != CW_OK) return gate.rc;
    if (rc == CW_OK && !gate.called) rc = dev_hash(hash_alg, d_src, block_bytes, src_stride, nblocks, d_digests, side, false, true);
    HIP_TRY(hipEventRecord(join, side));
    HIP_TRY(hipStreamWaitEvent(main_s, join, 0));
    if (rc == CW_OK && hint && !hint->pending) { // this call's queued share, for the next call
        const uint32_t *word = cw::lz4_queued_blocks_word(main_s);
        if (word) {
            if (!hint->h_queued) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&hint->h_queued), 64, hipHostMallocDefault));
            if (!hint->ev) HIP_TRY(hipEventCreateWithFlags(&hint->ev, hipEventDisableTiming));
            HIP_TRY(hipMemcpyAsync(hint->h_queued, word, sizeof(uint32_t), hipMemcpyDeviceToHost, main_s));
            HIP_TRY(hipEventRecord(hint->ev, main_s));
            hint->pending = true;
            hint->blocks_of_copy = nblocks;
        }
    }
    return rc;
}

int cw_dev_hash_and_compress(int hash_alg, int comp_alg, const void *d_src, size_t block_bytes, size_t src_stride,
                             size_t nblocks, void *d_digests, void *d_dst, size_t dst_stride, uint32_t *d_sizes, void *stream)
{
    if (comp_alg == CW_COMP_NONE) return cw_dev_hash(hash_alg, d_src, block_bytes, src_stride, nblocks, d_digests, stream);
    if (hash_alg == CW_HASH_NONE)
        return cw_dev_compress(comp_alg, d_src, block_bytes, src_stride, nblocks, d_dst, dst_stride, d_sizes, stream);
    ThreadCtx *c;
    int rc = thread_ctx(&c);
    if (rc != CW_OK) return rc;
    if (nblocks == 0) return CW_OK;
    if (!d_src || !d_dst || !d_sizes || !d_digests) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if ((rc = check_block(block_bytes)) != CW_OK) return rc;
    if (src_stride < block_bytes) return fail(CW_ERR_BAD_ARG, "src_stride < block_bytes");
    return dev_fused(c->side, c->fork, c->join, hash_alg, comp_alg, (const uint8_t *)d_src, block_bytes, src_stride, nblocks, (uint8_t *)d_digests, (uint8_t *)d_dst,
                     dst_stride, d_sizes, (hipStream_t)stream);
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

int cw_dev_gen_mixed(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes, void *d_dst, void *stream)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    if (!d_dst) return fail(CW_ERR_BAD_ARG, "NULL device pointer");
    if (block_bytes % 16 || block_bytes > CW_MAX_BLOCK_BYTES) return fail(CW_ERR_BAD_ARG, "block_bytes must be a multiple of 16, <= 65536");
    hipError_t e = cw::gen_mixed_launch(seed, first_block, nblocks, block_bytes, (uint8_t *)d_dst, (hipStream_t)stream);
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

// plain device memory for C callers of the cw_dev_* functions (the host programs link no HIP runtime themselves)
void *cw_dev_alloc(size_t bytes)
{
    if (ensure_init() != CW_OK) return nullptr;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { fail(CW_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
void cw_dev_free(void *d_p) { if (d_p) (void)hipFree(d_p); }
int cw_dev_upload(void *d_dst, const void *src, size_t bytes)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return CW_OK;
}
int cw_dev_download(void *dst, const void *d_src, size_t bytes)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return CW_OK;
}
int cw_dev_synchronize(void)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return CW_OK;
}

#ifdef CW_CLOCK_STAMP
// diagnostic build only (tools/clock_probe.py): the in-kernel clock stamps of the last launches
int cw_debug_clock_read(int which, unsigned long long *out)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(which == 0 ? cw::skein_clock_read(out) : cw::lz4_clock_read(out));
    return CW_OK;
}
#endif

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

int cw_profile_kernels(int kind, char *buf, size_t cap)
{
    if (kind < 0 || kind > 1 || !buf || cap == 0) return fail(CW_ERR_BAD_ARG, "cw_profile_kernels: kind 0 (codec) or 1 (hash), a buffer");
    strncpy(buf, t_kernels[kind], cap - 1);
    buf[cap - 1] = 0;
    return CW_OK;
}

// ---- batched host API ------------------------------------------------------------------------------------
// The pipelined batch path: what HashOffload::Start() ("xfer data, load kernel") and Complete() ("wait for and reap the
// results", HashOffload.h:26-40) were meant to be, for a whole batch.  The batch is cut into chunks and three of them are
// in flight at any time, each in a Slot with its own device buffers, pinned staging and streams:
//   s_h2d      chunk k+1 crosses the bus host -> device
//   slot.stream / slot.side   chunk k: codec and hash side by side, then the slots are packed into one stream on the device
//              (pack_kernels.hip) so that only the compressed bytes cross the bus back; sizes, the total and the digests
//              follow on the same stream
//   s_d2h      chunk k-1's packed payload crosses device -> host
//   host       meanwhile the calling thread moves chunk k-2's payload from pinned staging into the caller's slots.
// Host memory the copy engines touch directly must be page-locked: a caller's buffer that is (cw_host_alloc /
// cw_host_register) is used in place, anything else goes through the slot's pinned staging with one memcpy.
namespace {
struct HostJob {
    int hash_alg, comp_alg;
    const uint8_t *src; size_t bb, nblocks;
    uint8_t *digests; uint8_t *dst; size_t dst_stride; uint32_t *sizes;
    uint8_t *packed; size_t packed_cap; uint64_t *offsets; // packed-output form (dst == NULL)
    bool do_hash, do_comp, src_pinned, packed_pinned;
    size_t db, d_stride;
    uint64_t packed_off;
};

bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

size_t meta_off_total(size_t n) { return (n * 4 + 7) & ~(size_t)7; }
size_t meta_off_dig(size_t n) { return meta_off_total(n) + 8; }

int slot_reserve(HostJob &j, Slot &s, size_t n, bool stage_src, bool stage_pack, const Slot *lender = nullptr);

// The slots of a thread's pipeline share ONE pair of kernel streams (the first slot's) unless the job's codec is LZF (CW_HOST_SHARED_STREAMS=0|1
// forces either; decided when a slot is first opened).  HIP multiplexes streams onto four hardware queues per priority level, and kernels -- and the
// small device-to-host copies of a chunk's sizes and digests -- that land on one queue wait for each other: with a stream pair per slot (plus the
// codec's side streams per slot stream) the pipeline had sixteen streams, and its timeline showed a chunk's copies and kernels waiting behind
// another chunk's long kernels.  The chunks' kernels do not gain from overlapping each other anyway.  8 GiB through
// cw_hash_and_compress_packed, own streams -> shared: corpus, Skein-512 + LZ4, 64 KiB 24.1 -> 26.0 GB/s; Skein-256 + LZ4, 4 KiB 22.6 -> 33.4;
// SHA-256 + LZ4, 4 KiB 23.0 -> 33.5; noise 41.4 -> 43.6; SHA-256 + LZF, 4 KiB 25.5 -> 25.6, 64 KiB 17.6 -> 15.9 (its rounds of link and parse
// kernels do overlap across chunks).
const Slot *shared_lender(ThreadCtx &c, const HostJob &j, const Slot &s)
{
    const char *sh_env = cw::tune("CW_HOST_SHARED_STREAMS");
    const bool share = sh_env ? sh_env[0] == '1' : j.comp_alg != CW_COMP_LZF;
    if (!share || &s == &c.slot[0]) return nullptr;
    return c.slot[0].open() == CW_OK ? &c.slot[0] : nullptr;
}

int pipe_issue(ThreadCtx &c, HostJob &j, Slot &s, size_t first, size_t n)
{
    int rc = slot_reserve(j, s, n, !j.src_pinned, !(j.packed && j.packed_pinned), shared_lender(c, j, s));
    if (rc != CW_OK) return rc;
    s.first = first; s.n = n; s.total = 0;
    const size_t bytes = n * j.bb;
    const uint8_t *hsrc = j.src + first * j.bb;
    if (!j.src_pinned && bytes) {
        memcpy(s.h_src.p, hsrc, bytes);
        hsrc = (const uint8_t *)s.h_src.p;
    }
    if (bytes) HIP_TRY(hipMemcpyAsync(s.src.p, hsrc, bytes, hipMemcpyHostToDevice, c.s_h2d));
    HIP_TRY(hipEventRecord(s.ev_h2d, c.s_h2d));
    HIP_TRY(hipStreamWaitEvent(s.stream, s.ev_h2d, 0));
    uint8_t *meta = (uint8_t *)s.h_meta.p;
    if (j.do_comp && j.do_hash)
        rc = dev_fused(s.side, s.fork, s.join, j.hash_alg, j.comp_alg, (const uint8_t *)s.src.p, j.bb, j.bb, n, (uint8_t *)s.dig.p, (uint8_t *)s.dst.p,
                       j.d_stride, (uint32_t *)s.sizes.p, s.stream);
    else if (j.do_comp)
        rc = dev_compress(j.comp_alg, (const uint8_t *)s.src.p, j.bb, j.bb, n, (uint8_t *)s.dst.p, j.d_stride, (uint32_t *)s.sizes.p, s.stream);
    else
        rc = dev_hash(j.hash_alg, (const uint8_t *)s.src.p, j.bb, j.bb, n, (uint8_t *)s.dig.p, s.stream, false, true);
    if (rc != CW_OK) return rc;
    if (j.do_comp) {
        hipError_t pe = cw::pack_launch((const uint8_t *)s.dst.p, j.d_stride, (const uint32_t *)s.sizes.p, n, (uint8_t *)s.pack.p, (uint64_t *)s.offs.p,
                                        s.stream);
        if (pe != hipSuccess) return fail(CW_ERR_HIP, "pack launch: %s", hipGetErrorString(pe));
        HIP_TRY(hipMemcpyAsync(meta, s.sizes.p, n * 4, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(meta + meta_off_total(n), (const uint64_t *)s.offs.p + n, 8, hipMemcpyDeviceToHost, s.stream));
    }
    if (j.do_hash) HIP_TRY(hipMemcpyAsync(meta + meta_off_dig(n), s.dig.p, n * j.db, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(hipEventRecord(s.ev_meta, s.stream));
    return CW_OK;
}

// sizes, total and digests of the slot's chunk are on the host: hand them out and start the payload's way back
int pipe_reap(ThreadCtx &c, HostJob &j, Slot &s)
{
    HIP_TRY(hipEventSynchronize(s.ev_meta));
    const uint8_t *meta = (const uint8_t *)s.h_meta.p;
    const size_t n = s.n;
    if (j.do_hash) memcpy(j.digests + s.first * j.db, meta + meta_off_dig(n), n * j.db);
    if (j.do_comp) {
        const uint32_t *sz = (const uint32_t *)meta;
        memcpy(j.sizes + s.first, sz, n * 4);
        memcpy(&s.total, meta + meta_off_total(n), 8);
        if (j.dst) {
            for (size_t i = 0; i < n; i++)
                if (sz[i] > j.dst_stride) return fail(CW_ERR_BAD_ARG, "block %zu: %u bytes exceed dst_stride %zu", s.first + i, sz[i], j.dst_stride);
        } else {
            if (j.packed_off + s.total > j.packed_cap)
                return fail(CW_ERR_BAD_ARG, "packed stream needs more than the %zu bytes provided", j.packed_cap);
            uint64_t o = j.packed_off;
            for (size_t i = 0; i < n; i++) { j.offsets[s.first + i] = o; o += sz[i]; }
        }
        HIP_TRY(hipStreamWaitEvent(c.s_d2h, s.ev_meta, 0));
        if (s.total) {
            void *to = j.packed && j.packed_pinned ? (void *)(j.packed + j.packed_off) : s.h_pack.p;
            HIP_TRY(hipMemcpyAsync(to, s.pack.p, s.total, hipMemcpyDeviceToHost, c.s_d2h));
        }
        HIP_TRY(hipEventRecord(s.ev_payload, c.s_d2h));
        if (!j.dst) { s.first = (size_t)j.packed_off; j.packed_off += s.total; } // first now = the chunk's place in the packed stream
    }
    return CW_OK;
}

// the payload of the slot's chunk is in pinned staging: move it to where the caller wants it
int pipe_finish(HostJob &j, Slot &s)
{
    if (!j.do_comp) return CW_OK;
    HIP_TRY(hipEventSynchronize(s.ev_payload));
    if (!s.total) return CW_OK;
    const uint8_t *from = (const uint8_t *)s.h_pack.p;
    if (j.dst) {
        const uint32_t *sz = (const uint32_t *)s.h_meta.p; // still this chunk's: the slot is not reused before this returns
        for (size_t i = 0; i < s.n; i++) {
            memcpy(j.dst + (s.first + i) * j.dst_stride, from, sz[i]);
            from += sz[i];
        }
    } else if (!j.packed_pinned) {
        memcpy(j.packed + s.first, from, s.total);
    }
    return CW_OK;
}

void pipe_drain(ThreadCtx &c)
{
    (void)hipStreamSynchronize(c.s_h2d);
    for (Slot &s : c.slot) { if (s.stream) { (void)hipStreamSynchronize(s.stream); (void)hipStreamSynchronize(s.side); } }
    (void)hipStreamSynchronize(c.s_d2h);
}

// chunk: a chunk's kernels cost 8-10 ms almost whatever its size (a 64 KiB Skein block is a serial chain of 1,025 steps, 8 hash
// launches of ~0.5 ms each even for a few thousand blocks, plus scan / parse / pack launches), and the kernels of successive
// chunks do not overlap each other in practice (rocprofv3 timeline; more hardware queues -- GPU_MAX_HW_QUEUES -- did not
// change that and made 4 KiB blocks 7x slower).  So a chunk of large blocks must carry enough bytes for that time to hide
// behind its own host->device copy: 64 / 128 / 256 / 512 MiB measured 21.7 / 27.8 / 28.0 / 45.7 GB/s on random 64 KiB blocks
// (the last is the link's duplex rate).  Three slots x (input + slots + packed stream) = 4.6 GiB of device memory per calling
// thread at 512 MiB.  CW_HOST_CHUNK_MB overrides.
size_t pipeline_chunk(size_t bb, size_t nblocks)
{
    const char *ck_env = cw::tune("CW_HOST_CHUNK_MB");
    // 512 MiB: the chunk's kernels cost 8-10 ms whatever its size and do not overlap across chunks, and a chunk of 4 KiB blocks has
    // to be large enough for the lane parsers to run beside the LDS-resident ones (8 GiB of 4 KiB corpus blocks, Skein-256 + LZ4 /
    // SHA-256 + LZF: chunks of 64 MiB 29.5 / 18.8 GB/s, 512 MiB 38.6 / 27.2, 1 GiB 40.0 / 20.8; random data 47 -> 45-47 GB/s)
    size_t chunk_bytes = ck_env && atol(ck_env) > 0 ? (size_t)atol(ck_env) << 20 : (size_t)512 << 20;
    size_t chunk = chunk_bytes / (bb ? bb : 1);
    if (chunk == 0) chunk = 1;
    if (chunk > nblocks) chunk = nblocks;
    if (nblocks > chunk && nblocks < 3 * chunk) chunk = (nblocks + 2) / 3; // a small batch still gets three stages
    return chunk ? chunk : 1;
}

// device buffers, pinned staging, streams and events of one slot for chunks of n blocks (idempotent; grows only)
int slot_reserve(HostJob &j, Slot &s, size_t n, bool stage_src, bool stage_pack, const Slot *lender)
{
    int rc = s.open(lender);
    if (rc != CW_OK) return rc;
    if ((rc = s.src.reserve(n * j.bb + 16)) != CW_OK) return rc;
    if (j.do_hash && (rc = s.dig.reserve(n * j.db)) != CW_OK) return rc;
    if (j.do_comp && ((rc = s.dst.reserve(n * j.d_stride)) != CW_OK || (rc = s.sizes.reserve(n * 4)) != CW_OK ||
                      (rc = s.pack.reserve(n * j.d_stride)) != CW_OK || (rc = s.offs.reserve((n + 1) * 8)) != CW_OK))
        return rc;
    if ((rc = s.h_meta.reserve(meta_off_dig(n) + n * j.db)) != CW_OK) return rc;
    if (j.do_comp && stage_pack && (rc = s.h_pack.reserve(n * j.d_stride)) != CW_OK) return rc;
    if (stage_src && n * j.bb && (rc = s.h_src.reserve(n * j.bb)) != CW_OK) return rc;
    return CW_OK;
}

// Chunks of compressible blocks > 4 KiB grow once the first results are in.  A 512 MiB chunk is 8,192 blocks of 64 KiB: the regime of the
// wavefront-per-block parsers (13 GB/s LZ4, 8.5 GB/s LZF on text), and the codec, not the link, bounds the call.  2 GiB are 32 Ki blocks,
// which the lane parsers take (the number of blocks in a call): 8 GiB of the corpus in page-locked memory, Skein-512 + LZ4
// 23.0 -> 31.0 GB/s, SHA-256 + LZF 11.4 -> 24.5.  Noise stays with 512 MiB (45.7 GB/s; 40.5 with 2 GiB chunks: fill and drain).  Only
// with the caller's buffers page-locked (no 2 GiB of staging per slot) and CW_HOST_CHUNK_MB unset; 18.5 GiB of device memory per calling
// thread once grown.
constexpr size_t kBigChunkBytes = (size_t)2 << 30;

// blocks of a grown chunk, or 0 when chunks of this job never grow
size_t grown_chunk(const HostJob &j, size_t chunk, bool pinned_io)
{
    const char *ck_env = cw::tune("CW_HOST_CHUNK_MB");
    const char *bk_env = cw::tune("CW_HOST_BIG_CHUNK_MB"); // test knob: the grown chunk's size (and growth although CW_HOST_CHUNK_MB is set)
    const bool bk_set = bk_env && atol(bk_env) > 0;
    const size_t big = (bk_set ? (size_t)atol(bk_env) << 20 : kBigChunkBytes) / (j.bb ? j.bb : 1);
    const bool may = (bk_set || !(ck_env && atol(ck_env) > 0)) && j.do_comp && j.bb > 4096 && pinned_io && big > chunk;
    return may ? big : 0;
}

// grown chunks need 3 x (input + slots + packed stream) of device memory per calling thread: only when the device has it to spare
// (many host threads on one device each run a pipeline of their own)
bool room_to_grow(const ThreadCtx &c, const HostJob &j, size_t big)
{
    if (c.slot[0].src.cap >= big * j.bb && c.slot[kSlots - 1].src.cap >= big * j.bb) return true; // grown before
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    // + the lane parsers' tables on each slot stream (LZ4 4 GiB, LZF 8 GiB at most; lz4_kernel.hip / lzf_kernel.hip allocate them on the first
    //   large chunk and fall back to the parsers without lanes when the device cannot give them)
    const size_t lane_tabs = kSlots * (j.comp_alg == CW_COMP_LZF ? (size_t)8 << 30 : (size_t)4 << 30);
    const size_t need = kSlots * big * (j.bb + 2 * j.d_stride + 64) + lane_tabs;
    return free_b > need + ((size_t)16 << 30);
}

int host_pipeline(HostJob &j)
{
    ThreadCtx *cp;
    int rc = thread_ctx(&cp);
    if (rc != CW_OK) return rc;
    ThreadCtx &c = *cp;
    if (j.nblocks == 0 || (!j.do_hash && !j.do_comp)) { if (j.offsets) j.offsets[0] = 0; return CW_OK; }
    const size_t chunk = pipeline_chunk(j.bb, j.nblocks);
    j.src_pinned = is_pinned(j.src);
    j.packed_pinned = j.packed && is_pinned(j.packed);
    j.packed_off = 0;
    const size_t big = grown_chunk(j, chunk, j.src_pinned && j.packed_pinned);
    const bool may_grow = big != 0 && j.nblocks > 2 * chunk && room_to_grow(c, j, big);
    size_t next = 0, issued = 0, seen_in = 0, seen_out = 0; // blocks handed out; chunks issued; bytes in / out of the chunks reaped so far
    for (size_t k = 0; rc == CW_OK && (next < j.nblocks || k < issued + 2); k++) {
        if (next < j.nblocks) {
            const size_t left = j.nblocks - next;
            size_t n = left < chunk ? left : chunk;
            if (may_grow && seen_in && seen_in / 10 * 9 >= seen_out && left > chunk) // compressible so far (>= 10 % saved)
                n = left >= 2 * big ? big : left > big ? (left + 1) / 2 : left;
            const char *dbg_env = cw::tune("CW_DEBUG_HOST"); // prints the chunks of a call (tests)
            if (dbg_env && dbg_env[0] == '1') fprintf(stderr, "cw host pipeline: chunk %zu = %zu blocks of %zu B\n", issued, n, j.bb);
            rc = pipe_issue(c, j, c.slot[k % kSlots], next, n);
            next += n;
            issued++;
        }
        if (rc == CW_OK && k >= 1 && k - 1 < issued) {
            Slot &s = c.slot[(k - 1) % kSlots];
            rc = pipe_reap(c, j, s);
            seen_in += s.n * j.bb;
            seen_out += s.total;
            if (may_grow) { // a block that did not fit (LZF: size 0) is kept raw by the caller: it counts as not compressed
                const uint32_t *sz = (const uint32_t *)s.h_meta.p;
                size_t raw = 0;
                for (size_t i = 0; i < s.n; i++) raw += sz[i] == 0;
                seen_out += raw * j.bb;
            }
        }
        if (rc == CW_OK && k >= 2 && k - 2 < issued) rc = pipe_finish(j, c.slot[(k - 2) % kSlots]);
    }
    if (rc != CW_OK) { pipe_drain(c); return rc; }
    if (j.offsets) j.offsets[j.nblocks] = j.packed_off;
    return CW_OK;
}

int host_job_init(HostJob &j, int hash_alg, int comp_alg, const void *src, size_t block_bytes, size_t nblocks, void *digests, bool want_comp)
{
    memset(&j, 0, sizeof j);
    j.hash_alg = hash_alg; j.comp_alg = comp_alg; j.src = (const uint8_t *)src; j.bb = block_bytes; j.nblocks = nblocks;
    j.digests = (uint8_t *)digests;
    j.do_hash = hash_alg != CW_HASH_NONE && digests != nullptr;
    j.do_comp = comp_alg != CW_COMP_NONE && want_comp;
    if (nblocks == 0 || (!j.do_hash && !j.do_comp)) return CW_OK;
    if (!src && block_bytes) return fail(CW_ERR_BAD_ARG, "NULL src");
    int rc = check_block(block_bytes);
    if (rc != CW_OK) return rc;
    j.db = cw_digest_bytes(hash_alg);
    if (j.do_hash && j.db == 0) return fail(CW_ERR_BAD_ARG, "unknown hash algorithm %d", hash_alg);
    const size_t bound = j.do_comp ? cw_compress_bound(comp_alg, block_bytes) : 0;
    if (j.do_comp && bound == 0) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", comp_alg);
    if (j.do_comp && block_bytes == 0) return fail(CW_ERR_BAD_ARG, "compression needs block_bytes > 0");
    // The caller's slot may be the reference's (2*l for lz4, l-1 for lzf, :234-239); device slots use the bound.
    j.d_stride = (bound + 15) & ~(size_t)15;
    return CW_OK;
}
} // namespace

int cw_hash_and_compress_blocks(int hash_alg, int comp_alg, const void *src, size_t block_bytes, size_t nblocks, void *digests,
                                void *dst, size_t dst_stride, uint32_t *sizes)
{
    HostJob j;
    int rc = host_job_init(j, hash_alg, comp_alg, src, block_bytes, nblocks, digests, dst != nullptr && sizes != nullptr);
    if (rc != CW_OK) return rc;
    j.dst = (uint8_t *)dst; j.dst_stride = dst_stride; j.sizes = sizes;
    return host_pipeline(j);
}

int cw_hash_and_compress_packed(int hash_alg, int comp_alg, const void *src, size_t block_bytes, size_t nblocks, void *digests,
                                void *packed, size_t packed_cap, uint64_t *offsets, uint32_t *sizes)
{
    if (comp_alg != CW_COMP_LZ4 && comp_alg != CW_COMP_LZF) return fail(CW_ERR_BAD_ARG, "unknown compression algorithm %d", comp_alg);
    if (nblocks && (!packed || !offsets || !sizes)) return fail(CW_ERR_BAD_ARG, "NULL packed/offsets/sizes");
    HostJob j;
    int rc = host_job_init(j, hash_alg, comp_alg, src, block_bytes, nblocks, digests, true);
    if (rc != CW_OK) return rc;
    j.packed = (uint8_t *)packed; j.packed_cap = packed_cap; j.offsets = offsets; j.sizes = sizes;
    return host_pipeline(j);
}

// initializeGpu() (HashAndCompress.cpp:95-98) for the calling thread: its context on its device and everything the batch
// path allocates on first use for batches of up to nblocks blocks, so that a timed run starts with warm buffers
int cw_prepare(int hash_alg, int comp_alg, size_t block_bytes, size_t nblocks, int pinned_io)
{
    ThreadCtx *c;
    int rc = thread_ctx(&c);
    if (rc != CW_OK || nblocks == 0) return rc;
    HostJob j;
    int dummy = 0;
    if ((rc = host_job_init(j, hash_alg, comp_alg, &dummy, block_bytes, nblocks, &dummy, true)) != CW_OK) return rc;
    if (!j.do_hash && !j.do_comp) return CW_OK; // (CW_HASH_NONE, CW_COMP_NONE): nothing a batch would allocate
    size_t chunk = pipeline_chunk(block_bytes, nblocks);
    // a batch whose chunks may grow (host_pipeline): buffers, workspaces and lane tables for the grown chunk, so that the growth costs
    // no allocation inside a timed window (18.5 GiB of device memory per calling thread)
    const size_t big = grown_chunk(j, chunk, pinned_io != 0);
    if (big && nblocks > 2 * chunk && room_to_grow(*c, j, big)) chunk = nblocks - 2 * chunk < big ? nblocks - 2 * chunk : big;
    for (Slot &s : c->slot)
        if ((rc = slot_reserve(j, s, chunk, !pinned_io, !pinned_io, shared_lender(*c, j, s))) != CW_OK) return rc;
    // One chunk's kernels on every slot (over whatever its device buffers hold): the codecs' per-stream workspaces -- queues, link
    // arrays, the lane parsers' tables -- are allocated here instead of inside the first timed batch, and the device leaves its
    // idle clocks.  Then a few copies each way to wake the link (the first pass after idle ran at 25-29 GB/s against 45.7).
    // CW_PREPARE_COLD=1 skips both.
    const char *cold = cw::tune("CW_PREPARE_COLD");
    if (cold && cold[0] == '1') return CW_OK;
    for (Slot &s : c->slot) { // the same predicates as pipe_issue: a hash-only job has no slots, sizes or packed stream to touch
        if (j.do_comp && j.do_hash)
            rc = dev_fused(s.side, s.fork, s.join, hash_alg, comp_alg, (const uint8_t *)s.src.p, block_bytes, block_bytes, chunk, (uint8_t *)s.dig.p,
                           (uint8_t *)s.dst.p, j.d_stride, (uint32_t *)s.sizes.p, s.stream);
        else if (j.do_comp)
            rc = dev_compress(comp_alg, (const uint8_t *)s.src.p, block_bytes, block_bytes, chunk, (uint8_t *)s.dst.p, j.d_stride, (uint32_t *)s.sizes.p, s.stream);
        else
            rc = dev_hash(hash_alg, (const uint8_t *)s.src.p, block_bytes, block_bytes, chunk, (uint8_t *)s.dig.p, s.stream, false, true);
        if (rc != CW_OK) return rc;
        if (j.do_comp) {
            hipError_t pe = cw::pack_launch((const uint8_t *)s.dst.p, j.d_stride, (const uint32_t *)s.sizes.p, chunk, (uint8_t *)s.pack.p, (uint64_t *)s.offs.p,
                                            s.stream);
            if (pe != hipSuccess) return fail(CW_ERR_HIP, "pack launch: %s", hipGetErrorString(pe));
        }
    }
    void *h = nullptr;
    const size_t wb = (size_t)64 << 20 < chunk * block_bytes ? (size_t)64 << 20 : chunk * block_bytes;
    if (hipHostMalloc(&h, 2 * wb, hipHostMallocPortable) == hipSuccess) {
        for (int k = 0; k < 6; k++) {
            (void)hipMemcpyAsync(c->slot[1].src.p, h, wb, hipMemcpyHostToDevice, c->s_h2d);
            (void)hipMemcpyAsync((uint8_t *)h + wb, c->slot[2].src.p, wb, hipMemcpyDeviceToHost, c->s_d2h);
        }
    }
    pipe_drain(*c);
    if (h) (void)hipHostFree(h);
    (void)hipGetLastError();
    return CW_OK;
}

// page-locked host memory for the batch path (hipHostMalloc / hipHostRegister): buffers the copy engines use in place
void *cw_host_alloc(size_t bytes)
{
    if (ensure_init() != CW_OK) return nullptr;
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { fail(CW_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
void cw_host_free(void *p) { if (p) (void)hipHostFree(p); }
int cw_host_register(void *p, size_t bytes)
{
    int rc = ensure_init();
    if (rc != CW_OK) return rc;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return CW_OK;
}
int cw_host_unregister(void *p)
{
    HIP_TRY(hipHostUnregister(p));
    return CW_OK;
}

int cw_hash_tree_blocks(int hash_alg, const void *src, size_t block_bytes, size_t nblocks, unsigned leaf, unsigned node, unsigned max_level,
                        void *digests)
{
    ThreadCtx *cp;
    int rc = thread_ctx(&cp);
    if (rc != CW_OK) return rc;
    ThreadCtx &c = *cp;
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
    ThreadCtx *cp;
    int rc = thread_ctx(&cp);
    if (rc != CW_OK) return rc;
    ThreadCtx &c = *cp;
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
    int error = CW_OK;      // why the object is in CW_OFFLOAD_FAILED
    char error_msg[256] = "";
    int device = -1;        // the device the object was created on
    hipStream_t stream = nullptr;
    DevBuf d_src, d_dig;
};

namespace {
int offload_fail(cw_offload *h, int rc) // record the failure on the object, so that waiters and Complete() see it
{
    h->error = rc;
    strncpy(h->error_msg, t_err, sizeof h->error_msg - 1);
    h->state.store(CW_OFFLOAD_FAILED);
    return rc;
}
} // namespace

cw_offload_t *cw_offload_create(int hash_alg, int n_blocks, size_t block_bytes)
{
    if (ensure_init() != CW_OK) return nullptr;
    if (n_blocks <= 0 || cw_digest_bytes(hash_alg) == 0 || check_block(block_bytes) != CW_OK) {
        fail(CW_ERR_BAD_ARG, "cw_offload_create: bad arguments");
        return nullptr;
    }
    cw_offload *h = new cw_offload;
    h->hash_alg = hash_alg; h->n_blocks = n_blocks; h->block_bytes = block_bytes;
    h->device = current_device();
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
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
    h->error = CW_OK; h->error_msg[0] = 0;
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
    if (h->state.load() != CW_OFFLOAD_QUEUED) return fail(CW_ERR_STATE, "Start: state %d != hQueued", h->state.load());
    // everything that can fail is checked or attempted BEFORE the object counts as offloaded; a failure leaves it in
    // CW_OFFLOAD_FAILED with the reason on the object (cw_offload_error), never in hOffloaded with nothing in flight
    if (!h->data || !h->results) return offload_fail(h, fail(CW_ERR_BAD_ARG, "Start: Reset() gave no data/results"));
    const size_t bytes = (size_t)h->n_blocks * h->block_bytes, db = cw_digest_bytes(h->hash_alg);
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(h->d_src.p, h->data, bytes, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return offload_fail(h, fail(CW_ERR_HIP, "Start: %s", hipGetErrorString(e)));
    int rc = dev_hash(h->hash_alg, (const uint8_t *)h->d_src.p, h->block_bytes, h->block_bytes, (size_t)h->n_blocks, (uint8_t *)h->d_dig.p, h->stream);
    if (rc != CW_OK) { (void)hipStreamSynchronize(h->stream); return offload_fail(h, rc); }
    e = hipMemcpyAsync(h->results, h->d_dig.p, (size_t)h->n_blocks * db, hipMemcpyDeviceToHost, h->stream);
    if (e != hipSuccess) { (void)hipStreamSynchronize(h->stream); return offload_fail(h, fail(CW_ERR_HIP, "Start: %s", hipGetErrorString(e))); }
    h->state.store(CW_OFFLOAD_OFFLOADED);
    return CW_OK;
}

int cw_offload_complete(cw_offload_t *h)
{
    if (!h) return fail(CW_ERR_BAD_ARG, "NULL offload");
    if (h->state.load() == CW_OFFLOAD_FAILED) return fail(h->error, "Complete: the offload failed: %s", h->error_msg);
    if (h->state.load() != CW_OFFLOAD_OFFLOADED) return fail(CW_ERR_STATE, "Complete: state %d != hOffloaded", h->state.load());
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return offload_fail(h, fail(CW_ERR_HIP, "Complete: %s", hipGetErrorString(e)));
    h->state.store(CW_OFFLOAD_COMPLETE);
    if (h->on_complete) h->on_complete(h->arg);
    return CW_OK;
}

int cw_offload_completed(const cw_offload_t *h) { return h && h->state.load() == CW_OFFLOAD_COMPLETE; }
int cw_offload_state(const cw_offload_t *h) { return h ? h->state.load() : CW_ERR_BAD_ARG; }
int cw_offload_error(const cw_offload_t *h) { return h ? h->error : CW_ERR_BAD_ARG; }

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
        if (cw_offload_do(h) != CW_OK) {
            // the reason is on the object (CW_OFFLOAD_FAILED, cw_offload_error); whoever waits for the callback is
            // still woken, and finds Completed() false
            fprintf(stderr, "libcwhc: offload failed: %s\n", t_err);
            if (h->state.load() != CW_OFFLOAD_FAILED) offload_fail(h, CW_ERR_STATE);
            if (h->on_complete) h->on_complete(h->arg);
        }
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
