// cw_device.h -- internal declarations shared by the HIP translation units of libcwhc.so.
// Not part of the public boundary (that is include/cw_hashcompress.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#ifndef CW_SKEIN_THREADS
#define CW_SKEIN_THREADS 64 // one wavefront per workgroup: lanes never communicate, small groups spread evenly
#endif

namespace cw {

// Diagnostic build only (-DCW_CLOCK_STAMP, tools/clock_probe.py): the clock a kernel actually runs at.  Lane 0 of a workgroup
// reads s_memtime (shader cycles) and s_memrealtime (a constant 100 MHz counter) when the workgroup starts and when it ends;
// the four values go to a buffer nothing else reads, and the host takes, per workgroup, d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  In the product build no stamp executes.
#ifdef CW_CLOCK_STAMP
constexpr unsigned kClockSlots = 1024;
struct ClockScope {
    unsigned long long m0 = 0, r0 = 0;
    unsigned long long *rec;
    bool on;
    __device__ ClockScope(unsigned long long *buf, unsigned wg) : rec(buf + 4 * (wg % kClockSlots)), on(threadIdx.x == 0)
    {
        if (on) { m0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ ~ClockScope()
    {
        if (on) {
            const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            rec[0] = m0; rec[1] = r0; rec[2] = m1; rec[3] = r1;
        }
    }
};
#define CW_CLOCK_SCOPE(buf) ClockScope clock_scope_(buf, blockIdx.x)
// keyed: the kClockSlots records are cut into 8 groups of 128, one per `key` (the slice kernel's launch index within a pass), so
// that the launches that overlap the codec and those that run after it are reported apart
#define CW_CLOCK_SCOPE_KEYED(buf, key) ClockScope clock_scope_(buf, ((unsigned)(key) % 8u) * 128u + blockIdx.x % 128u)
// which: 0 = the Skein slice kernel, 1 = the LZ4 span scan; out = kClockSlots x {m0, r0, m1, r1}
hipError_t skein_clock_read(unsigned long long *out);
hipError_t lz4_clock_read(unsigned long long *out);
#else
#define CW_CLOCK_SCOPE(buf) do { } while (0)
#define CW_CLOCK_SCOPE_KEYED(buf, key) do { } while (0)
#endif

// Key of the per-(device, stream) scratch the launch sequences keep (the NULL stream exists once per device).  Every
// entry also holds a launch mutex: a sequence of launches that shares scratch -- memset counters, scan, parse, redo; the
// chained slices of one hash -- is queued under it, so that two host threads using the same stream cannot interleave
// their sequences (stream order then keeps each sequence atomic).
static inline uint64_t ws_key(hipStream_t s)
{
    int d = 0;
    (void)hipGetDevice(&d);
    return ((uint64_t)reinterpret_cast<uintptr_t>(s) << 4) | (uint64_t)(d & 15);
}

struct SkeinIV { uint64_t w[8]; };

// The launch functions note which kernels they used, per calling thread (kind 0 = codec, 1 = hash): cw_profile_kernels
// hands the names to the caller so that a benchmark reports what ran instead of guessing it from its arguments.
void note_kernels(int kind, const char *names);
// Value of a tuning / test knob: what cw_tune_set gave it, else the environment variable of that name, else nullptr.  Asked per call
// (never cached in a static), so tests sweep settings in one process.  The pointer stays valid until the knob is set again.
const char *tune(const char *key);

// host: chaining value after the configuration block (Skein_*_Init)
void skein_compute_iv(int state_words, unsigned hash_bits, SkeinIV *iv, uint64_t tree_info = 0);
// sliced Skein for the fused call: the steps of every block in 8 launches (see skein_kernels.hip)
bool skein_sliced_applies(int state_words, const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks,
                          const uint8_t *digests);
hipError_t skein_sliced_launch(int state_words, const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                               uint8_t *digests, unsigned digest_bytes, hipStream_t stream);
// tree hashing of every block (one wavefront per block, lane = leaf/node); digest = hash_bits / 8 bytes per block
hipError_t skein_tree_launch(int state_words, const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks,
                             unsigned hash_bits, unsigned leaf, unsigned node, unsigned max_level, uint8_t *digests, hipStream_t stream);

// device launches (async on `stream`); src_stride = distance between consecutive blocks in bytes
// lean: the caller runs codec wavefronts beside the hash kernel and wants its low-register variant
hipError_t skein512_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                           uint8_t *digests, unsigned digest_bytes, hipStream_t stream, bool lean = false);
hipError_t skein256_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                           uint8_t *digests, unsigned digest_bytes, hipStream_t stream, bool lean = false);
hipError_t sha256_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, uint8_t *digests,
                         hipStream_t stream);
// after_scan (optional): called ONCE, right behind the launch of the scan and in front of everything else, if the call gets that far (the caller
// checks): what it enqueues on other streams runs beside the scan; what it makes `stream` wait for, the parsers wait for
struct AfterScan { hipError_t (*fn)(void *ctx); void *ctx; };
hipError_t lz4_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, uint8_t *dst,
                      size_t dst_stride, uint32_t *sizes, hipStream_t stream, const AfterScan *after_scan = nullptr);
const uint32_t *lz4_queued_blocks_word(hipStream_t stream);
hipError_t lzf_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, uint8_t *dst,
                      size_t dst_stride, uint32_t *sizes, hipStream_t stream);
// LZ4 parser with the table in vector registers (lz4_vtab_kernel.hip): parses blocks of the scan's queue (counters[0] = head,
// counters[1] = length) while more than `reserve` are left, if the queue's length lies in [min_queued, max_queued)
hipError_t lz4_vtab_launch(const uint8_t *src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *dst, size_t dst_stride, uint32_t *sizes,
                           const uint32_t *queue, uint32_t *counters, uint32_t min_queued, uint32_t max_queued, uint32_t reserve,
                           unsigned waves_per_cu, hipStream_t stream, const char **kernel_name);
// the same scalar-thread parser with its table in LDS (ten wavefronts per CU): takes what is left of the queue
hipError_t lz4_ltab_launch(const uint8_t *src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *dst, size_t dst_stride, uint32_t *sizes,
                           const uint32_t *queue, uint32_t *counters, unsigned waves_per_cu, hipStream_t stream);
hipError_t decompress_launch(int alg, const uint8_t *comp, size_t comp_stride, const uint32_t *sizes, size_t nblocks, uint8_t *dst,
                             size_t block_bytes, uint32_t *status, hipStream_t stream);
// packed stream: offsets[i] = sum sizes[0..i) (nblocks + 1 entries); slot i copied to packed + offsets[i] (packed may be NULL)
hipError_t pack_launch(const uint8_t *slots, size_t slot_stride, const uint32_t *sizes, size_t nblocks, uint8_t *packed,
                       uint64_t *offsets, hipStream_t stream);
// per-stream scratch of the codec / pack launches (queues, link arrays, scan partials): freed by cw_shutdown
void skein_release_workspaces();
void lz4_release_workspaces();
void lzf_release_workspaces();
void pack_release_workspaces();
// the same for ONE stream of the current device: called by whoever owns the stream before destroying it, so that a short-lived
// calling thread does not leave gigabytes of lane tables behind and a recycled stream handle does not inherit a stale entry
void release_stream_workspaces(hipStream_t stream);
void skein_release_stream(hipStream_t stream);
void lz4_release_stream(hipStream_t stream);
void lzf_release_stream(hipStream_t stream);
void pack_release_stream(hipStream_t stream);
hipError_t sum_sizes_launch(const uint32_t *sizes, size_t n, uint32_t raw_bytes, uint64_t *totals, hipStream_t stream);
hipError_t gen_random_launch(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes, uint8_t *dst,
                             hipStream_t stream);
hipError_t gen_mixed_launch(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes, uint8_t *dst,
                            hipStream_t stream);

} // namespace cw
