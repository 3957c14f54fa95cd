/*
 * cw_hashcompress.h -- C ABI of libcwhc.so: the MI355X (gfx950) back end for the reference's
 * per-block hash + front-end compression hot path.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  Plain pointers and sizes only; no C++/torch types cross this line.
 *
 *   slots      doHashing / doCompression          src/hashandcompress/HashAndCompress.cpp:111,119
 *   hashes     doSkeinHashing                     src/hashandcompress/HashAndCompress.cpp:121-134
 *              doSHA256MBHashing                  src/hashandcompress/HashAndCompress.cpp:136-158
 *              HashBlockSkein256/SHA256/SHA256MB  src/hashing_perf/hash.cpp:5-77
 *   codecs     lz4 / lzf lambdas                  src/hashandcompress/HashAndCompress.cpp:342-355
 *   GPU seam   initializeGpu()                    src/hashandcompress/HashAndCompress.cpp:95-98
 *              class HashOffload                  src/hashandcompress/HashOffload.h:13-64
 *              hashing_offload_entry_point        src/hashandcompress/HashAndCompress.cpp:160-183
 *
 * Threading: every function may be called concurrently from any number of host threads
 * (the reference calls its slots from --c-threads workers, :398-402); each calling thread
 * gets its own HIP streams and staging buffers per device.  The cw_dev_* functions may also be
 * called from several threads on the SAME stream: each call's launch sequence is queued
 * atomically (a per-(device, stream) mutex), so the calls execute in some serial order.
 *
 * Devices: cw_init(d) / cw_set_device(d) initialise device d (if needed) and make it the calling
 * thread's device; threads that never choose use the first initialised device.  Device pointers
 * and streams passed to cw_dev_* must belong to the calling thread's device.
 *
 * There is NO CPU fallback: every compute entry point fails (CW_ERR_NO_DEVICE / abort in the
 * void slot wrappers) when no gfx950 device is usable.
 */
#ifndef CW_HASHCOMPRESS_H
#define CW_HASHCOMPRESS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------------------------- */
enum {
    CW_OK = 0,
    CW_ERR_NO_DEVICE = -1,   /* no usable HIP device / cw_init not possible */
    CW_ERR_BAD_ARG = -2,     /* unknown algorithm, NULL pointer, size out of the supported range */
    CW_ERR_HIP = -3,         /* a HIP runtime call failed; see cw_last_error() */
    CW_ERR_STATE = -4,       /* offload object used out of order (HashOffload's asserts) */
    CW_ERR_NOMEM = -5
};

/* --hash-alg values of the driver (:361-370) plus the north-star Skein-512-512 */
typedef enum cw_hash_alg {
    CW_HASH_SKEIN512 = 0,      /* Skein-512, 512-bit digest (reference_code/skein, skein.c:226-408) */
    CW_HASH_SKEIN256_128 = 1,  /* "skein": Skein-256, 128-bit digest (kHashSizeBytesSkein=16, hash.h:16) */
    CW_HASH_SHA256 = 2,        /* "sha256mb": FIPS 180-4 SHA-256 of each block (kHashSizeBytesSHA=32) */
    CW_HASH_NONE = 3
} cw_hash_alg;

/* --comp-alg values of the driver (:342-355) */
typedef enum cw_comp_alg {
    CW_COMP_LZ4 = 0,           /* LZ4_compress_default(s, d, l, 2*l), LZ4 v1.8.2 */
    CW_COMP_LZF = 1,           /* lzf_compress(s, l, d, l-1), liblzf HLOG 16 / VERY_FAST */
    CW_COMP_NONE = 2
} cw_comp_alg;

/* ---- lifecycle: initializeGpu() (:95-98) and the shutdown hook (:182) ----------------------- */
int  cw_init(int device);            /* initialise `device` (idempotent) and make it the calling thread's device */
void cw_shutdown(void);              /* every initialised device: synchronise, free the library's scratch */
int  cw_device_count(void);          /* usable gfx950 devices (0 when none) */
int  cw_set_device(int device);      /* = cw_init: the device the calling thread's next calls run on (--devices) */
int  cw_get_device(void);            /* the calling thread's device, -1 before any cw_init */
const char *cw_last_error(void);     /* message of the calling thread's last failure */
const char *cw_version(void);

/* ---- sizes ---------------------------------------------------------------------------------- */
size_t cw_digest_bytes(int hash_alg);                        /* 64 / 16 / 32 / 0 */
size_t cw_compress_bound(int comp_alg, size_t block_bytes);  /* output slot a block may need:
                                                                lz4: l + l/255 + 16 (LZ4_compressBound); lzf: l */
#define CW_MAX_BLOCK_BYTES 65536u   /* LZ4's 16-bit-table regime of the reference (< 65547) */

/* ---- the two slots, host pointers, synchronous -- drop-in for the reference's function objects --
 * void doHashing(const char* src, char* dst, int count)  hashes `count` consecutive blocks of
 * cw_get_block_size() bytes at src into count consecutive digests at dst (:121-133).
 * size_t doCompression(const char* src, char* dst, size_t len) compresses one block; returns the
 * compressed size, 0 = did not fit (lzf, lzf.h:59-64).  dst capacity is 2*len (lz4) / len-1 (lzf),
 * as the reference's callers provide (:234-239, :346, :353).
 * The void slots abort() with a message on a device error, since the reference's slots cannot
 * report one (SURVEY.md 8b "Errors").                                                            */
void   cw_set_block_size(size_t block_bytes);   /* the reference's global blockSize (:89), default 4096 */
size_t cw_get_block_size(void);
void   cw_hash_skein(const char *src, char *dst, int count);      /* doSkeinHashing: Skein-256-128 */
void   cw_hash_skein512(const char *src, char *dst, int count);   /* Skein-512-512 */
void   cw_hash_sha256mb(const char *src, char *dst, int count);   /* doSHA256MBHashing (digests ARE returned) */
size_t cw_compress_lz4(const char *src, char *dst, size_t len);
size_t cw_compress_lzf(const char *src, char *dst, size_t len);
/* The decoders the reference times beside the compressors (src/compression_perf/src/experiment.cpp:118,256):
 *   int LZ4_decompress_safe(src, dst, csize, dst_cap)           -> decoded bytes, < 0 = malformed
 *   unsigned lzf_decompress(src, csize, dst, dst_cap)           -> decoded bytes, 0 = error (lzf.h:83-97)
 * One difference: the device decoder is told the decoded size, so a slot must decode to exactly
 * cw_get_block_size() bytes (what every caller in the reference expects); anything else is "malformed". */
int      cw_decompress_lz4(const char *src, char *dst, int csize, int dst_cap);
unsigned cw_decompress_lzf(const void *src, unsigned csize, void *dst, unsigned dst_cap);

/* ---- batched host API: many blocks per call (H2D, kernels, D2H on the caller's stream) --------
 * src: nblocks * block_bytes contiguous; digests: nblocks * cw_digest_bytes(); dst: nblocks slots of
 * dst_stride >= cw_compress_bound(); sizes[i] = compressed bytes of block i (0 = did not fit).
 * Any of digests / (dst,sizes) may be NULL to skip that half.  This is ProcessBlock (:231-261) for
 * a whole read-unit or file at once.                                                              */
int cw_hash_blocks(int hash_alg, const void *src, size_t block_bytes, size_t nblocks, void *digests);
int cw_compress_blocks(int comp_alg, const void *src, size_t block_bytes, size_t nblocks,
                       void *dst, size_t dst_stride, uint32_t *sizes);
int cw_hash_and_compress_blocks(int hash_alg, int comp_alg, const void *src, size_t block_bytes,
                                size_t nblocks, void *digests, void *dst, size_t dst_stride,
                                uint32_t *sizes);
/* Same work, packed output: the compressed blocks arrive as ONE stream (block i at packed + offsets[i], sizes[i] bytes;
 * offsets has nblocks + 1 entries, the last one the total; a block that did not fit occupies nothing) -- the device
 * packs the slots (cw_dev_pack) so only compressed bytes cross the bus, and with a page-locked `packed` buffer they land
 * in place with no copy on the host.  packed_cap must cover the total (nblocks * cw_compress_bound() always does). */
int cw_hash_and_compress_packed(int hash_alg, int comp_alg, const void *src, size_t block_bytes,
                                size_t nblocks, void *digests, void *packed, size_t packed_cap,
                                uint64_t *offsets, uint32_t *sizes);
/* Both forms run as a three-stage pipeline (host->device copy of chunk k+1 | kernels of chunk k | device->host copy of
 * chunk k-1: what HashOffload::Start()/Complete() were meant to be, HashOffload.h:26-40).  The copy engines read and
 * write page-locked host memory in place; other buffers go through pinned staging with one memcpy.  Chunks are 512 MiB
 * (4.6 GiB of device memory per calling thread and device, plus the codecs' per-stream scratch: up to 4 GiB (LZ4) / 8 GiB (LZF)
 * of lane-parser tables on each of the three slot streams once a chunk is large enough for the lanes -- a device that cannot give
 * them runs the call without the lanes instead of failing it); with page-locked buffers on both sides, chunks of blocks
 * above 4 KiB grow to 2 GiB once the first results show that the data compresses (18.5 GiB then), because the codecs
 * reach their rate only on tens of thousands of blocks at a time.  To get buffers the engines can use directly:       */
/* initializeGpu() (:95-98) for the calling thread: its context on its device plus everything the batch path would allocate
 * on first use for batches of up to nblocks blocks (pinned_io: the caller's buffers are page-locked, no staging needed) */
int   cw_prepare(int hash_alg, int comp_alg, size_t block_bytes, size_t nblocks, int pinned_io);
void *cw_host_alloc(size_t bytes);               /* page-locked host memory (NULL on failure) */
void  cw_host_free(void *p);
int   cw_host_register(void *p, size_t bytes);   /* page-lock an existing buffer (e.g. the driver's read units) */
int   cw_host_unregister(void *p);
/* decode nblocks slots (comp_stride apart, sizes[i] bytes each) into nblocks * block_bytes at dst;
 * status[i] = 0 iff slot i is well formed and yields exactly block_bytes (see cw_dev_decompress).  */
/* host-buffer form of cw_dev_hash_tree */
int cw_hash_tree_blocks(int hash_alg, const void *src, size_t block_bytes, size_t nblocks, unsigned leaf, unsigned node,
                        unsigned max_level, void *digests);
int cw_decompress_blocks(int comp_alg, const void *comp, size_t comp_stride, const uint32_t *sizes,
                         size_t nblocks, void *dst, size_t block_bytes, uint32_t *status);

/* ---- device-resident API: every pointer is device memory, work is queued on `stream`
 *      (a hipStream_t passed as void*; NULL = the default stream) and NOT synchronised -----------
 * src_stride = bytes between consecutive blocks (>= block_bytes).                                 */
int cw_dev_hash(int hash_alg, const void *d_src, size_t block_bytes, size_t src_stride, size_t nblocks,
                void *d_digests, void *stream);
int cw_dev_compress(int comp_alg, const void *d_src, size_t block_bytes, size_t src_stride,
                    size_t nblocks, void *d_dst, size_t dst_stride, uint32_t *d_sizes, void *stream);
int cw_dev_hash_and_compress(int hash_alg, int comp_alg, const void *d_src, size_t block_bytes,
                             size_t src_stride, size_t nblocks, void *d_digests, void *d_dst,
                             size_t dst_stride, uint32_t *d_sizes, void *stream);
/* Decoders (the reference calls LZ4_decompress_safe / lzf_decompress only to time them,
 * src/compression_perf/src/experiment.cpp:118,256): decode nblocks compressed slots (comp_stride apart, d_sizes[i]
 * bytes each) into nblocks * block_bytes at d_dst; d_status[i] = 0 iff slot i is well formed and yields exactly
 * block_bytes.  Used as the reference-independent round-trip verifier of the compressors.                     */
int cw_dev_decompress(int comp_alg, const void *d_comp, size_t comp_stride, const uint32_t *d_sizes, size_t nblocks,
                      void *d_dst, size_t block_bytes, uint32_t *d_status, void *stream);
/* Skein tree hashing of every block (the reference's Skein_TreeHash,
 * reference_code/skein/Additional_Implementations/skein_test.c:616-680, tree fields skein.h:209-210): leaves of
 * state_bytes << leaf bytes, nodes of state_bytes << node bytes, at most max_level levels (>= 2; 255 = unlimited).
 * NOT the digest of cw_dev_hash -- tree mode changes the configuration block -- but one wavefront hashes a block with
 * lane-per-leaf parallelism, so a few large blocks already fill the GPU.  hash_alg: CW_HASH_SKEIN512 (64-byte digests)
 * or CW_HASH_SKEIN256_128 (16-byte digests).  The level buffers live in LDS: (block_bytes >> leaf) * 1.5 <= 64 KiB.   */
int cw_dev_hash_tree(int hash_alg, const void *d_src, size_t block_bytes, size_t src_stride, size_t nblocks,
                     unsigned leaf, unsigned node, unsigned max_level, void *d_digests, void *stream);
/* Packed output stream + block index: d_offsets[i] = sum of d_sizes[0..i) (u64, nblocks + 1
 * entries, the last one is the total); slot i's d_sizes[i] bytes are copied to d_packed + d_offsets[i].  A block that
 * did not fit (size 0) occupies nothing.  d_packed may be NULL: index only.  d_packed needs sum(d_sizes) bytes
 * (at most nblocks * slot_stride).                                                                              */
int cw_dev_pack(const void *d_slots, size_t slot_stride, const uint32_t *d_sizes, size_t nblocks,
                void *d_packed, uint64_t *d_offsets, void *stream);
/* synthetic input: u64 word w of block b = splitmix64(seed ^ (b << 13 | w)) */
int cw_dev_gen_random(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes,
                      void *d_dst, void *stream);
/* compressible synthetic mix: even blocks as cw_dev_gen_random, odd blocks = a 64-byte motif repeated with
 * every byte replaced by a random one with probability 1/16 (exact definition: csrc/misc_kernels.hip, host twin
 * oracle/hc_oracle.c) -- so that the codecs' match/emit loops are timed, not only their incompressible fast path */
int cw_dev_gen_mixed(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes,
                     void *d_dst, void *stream);
/* d_totals[0] += sum of sizes (a 0 counts as raw_bytes: stored uncompressed); d_totals[1] += #zeros */
int cw_dev_sum_sizes(const uint32_t *d_sizes, size_t nblocks, uint32_t raw_bytes, uint64_t *d_totals,
                     void *stream);

/* plain device memory on the calling thread's device, for C callers of cw_dev_* (the host programs link no HIP runtime) */
void *cw_dev_alloc(size_t bytes);                                   /* NULL on failure */
void  cw_dev_free(void *d_p);
int   cw_dev_upload(void *d_dst, const void *src, size_t bytes);    /* synchronous copies */
int   cw_dev_download(void *dst, const void *d_src, size_t bytes);
int   cw_dev_synchronize(void);                                     /* all work queued on the calling thread's device */

/* ---- kernel timing (the reference times with std::chrono around its calls, hash.cpp:11-18, HashAndCompress.cpp:397-406;
 *      device work is asynchronous, so the library brackets its own launches with HIP events on the stream each
 *      kernel is launched on).  Per calling thread.  Kinds: [0] codec kernels, [1] hash kernel, [2] reserved.       */
void cw_profile_enable(int on);
int  cw_profile_read(double ms_sum[3], unsigned count[3], int reset);   /* synchronises on the recorded events */
/* names of the kernels the calling thread's latest codec (kind 0) / hash (kind 1) launch used, as rocprofv3 prints them */
int  cw_profile_kernels(int kind, char *buf, size_t cap);

/* ---- CW_TESTING: tuning and test knobs -----------------------------------------------------------------------------
 * Every knob the launch policy reads (thresholds such as CW_LZ4_LANES / CW_LZF_LANES / CW_LZ4_VTAB, CW_LANES_*, CW_LZF_ROUND,
 * CW_LZ_FORCE_REDO, CW_DECODE_LANES, CW_HOST_*CHUNK_MB, ...; DESIGN.md lists them) is looked up PER CALL: the value given
 * here wins, the environment variable of the same name is the default.  value = NULL removes an override.  Not part of the
 * reference's interface (it has no tunables beyond its CLI); meant for tests and profiling, and not to be changed while
 * other threads are inside compute calls.                                                                             */
int  cw_tune_set(const char *key, const char *value);
void cw_tune_reset(void);

/* ---- HashOffload (HashOffload.h:13-64): batch object + the offload thread that drains it -------
 * Lifecycle  hInit --Enqueue--> hQueued --Start--> hOffloaded --Complete--> hComplete.
 * Start() = "xfer data, load kernel" (:26-31): async H2D + hash kernel + async D2H on the object's stream.
 * Complete() = "wait for and reap the results" (:33-40): blocks, then runs on_complete(arg).       */
typedef struct cw_offload cw_offload_t;
enum { CW_OFFLOAD_INIT = 0, CW_OFFLOAD_QUEUED = 1, CW_OFFLOAD_OFFLOADED = 2, CW_OFFLOAD_COMPLETE = 3,
       CW_OFFLOAD_FAILED = 4 /* Start()/Complete() failed: nothing is in flight, cw_offload_error() says why;
                                Reset() makes the object usable again */ };

cw_offload_t *cw_offload_create(int hash_alg, int n_blocks, size_t block_bytes);   /* HashOffload(int nBlocks) */
void cw_offload_destroy(cw_offload_t *h);
int  cw_offload_reset(cw_offload_t *h, char *data, char *results,
                      void (*on_complete)(void *), void *arg);                     /* Reset(d, r, f) */
int  cw_offload_enqueue(cw_offload_t *h);                                          /* Enqueue() */
int  cw_offload_start(cw_offload_t *h);                                            /* Start() */
int  cw_offload_complete(cw_offload_t *h);                                         /* Complete() */
int  cw_offload_completed(const cw_offload_t *h);                                  /* Completed() */
int  cw_offload_state(const cw_offload_t *h);
int  cw_offload_error(const cw_offload_t *h);                                      /* CW_OK, or why the state is hFailed */
int  cw_offload_do(cw_offload_t *h);                                               /* DoOffload() */

/* hashing_offload_entry_point (:160-183): one consumer thread popping a queue of HashOffload* */
int  cw_offload_thread_start(void);
int  cw_offload_submit(cw_offload_t *h);   /* Enqueue() + push + notify (the producer the reference never wrote);
                                              on_complete runs on the offload thread also when the offload FAILED --
                                              check cw_offload_completed() / cw_offload_error() in it */
void cw_offload_thread_stop(void);         /* allWorkFinished = true; join */

/* ---- several GPUs of one node (BASELINE.json configs[4]) ----------------------------------------------
 * The reference's only parallelism is worker threads over independent blocks (HashAndCompress.cpp:398-403); its dormant
 * --gpu-offload seam (:305,331) has one device at most.  Here the block index space is cut into contiguous shards, one
 * per device, each processed with the single-device entry points above (one host thread per device, cw_set_device);
 * the only exchange is the result gather after a pass, over RCCL (xGMI between the GPUs of the node):
 * ncclAllGather of the digests, ncclAllReduce(sum, u64) of the byte totals.  RCCL is loaded (dlopen) by cw_mgpu_create
 * only.                                                                                                             */
/* shard g of G over n units: [*first, *last) = [g*n/G, (g+1)*n/G) */
void cw_shard_range(size_t n, int g, int G, size_t *first, size_t *last);
typedef struct cw_mgpu cw_mgpu_t;
cw_mgpu_t *cw_mgpu_create(const int *devices, int ndev);   /* cw_init on each + ncclCommInitAll; NULL on failure */
void cw_mgpu_destroy(cw_mgpu_t *m);
int  cw_mgpu_ndev(const cw_mgpu_t *m);
int  cw_mgpu_device(const cw_mgpu_t *m, int rank);
const char *cw_mgpu_last_error(void);
/* The gather: rank g contributes bytes_each bytes at d_local[g] (its shard's digests, padded to the largest shard) and
 * ntotals u64 at d_totals[g]; afterwards d_all[g] on EVERY device holds all ranks' contributions in rank order
 * (ndev * bytes_each bytes) and d_totals[g] the element-wise sums.  The work that produced the inputs must be complete
 * (synchronise its streams first).  Either half may be skipped (bytes_each = 0 / ntotals = 0).  Blocking.            */
int  cw_mgpu_gather(cw_mgpu_t *m, const void *const *d_local, size_t bytes_each, void *const *d_all,
                    uint64_t *const *d_totals, size_t ntotals);

#ifdef __cplusplus
}
#endif
#endif /* CW_HASHCOMPRESS_H */
