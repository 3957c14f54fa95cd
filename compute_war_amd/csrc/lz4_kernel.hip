// lz4_kernel.hip -- bit-exact LZ4 block compression (the v1.8.2 "fast" greedy parser, acceleration 1,
// 16-bit position table) for gfx950, one storage block per wavefront.
//
// Replaces the reference's lz4 slot  LZ4_compress_default(s, d, l, 2*l)
// (src/hashandcompress/HashAndCompress.cpp:351-354, src/compression_perf/src/experiment.cpp:249;
// API src/compression_perf/include/lz4/lz4.h:126-139) for l < 65547, the only regime the
// reference's 4 KiB blocks (and the 64 KiB north-star blocks) ever enter.  The parser's semantics are
// those written down in SURVEY.md 8(a) row A5 and restated on the CPU in oracle/lz4_oracle.c.
//
// Mapping to the machine.  The parse is a serial greedy walk over one mutable hash table, so the unit of
// parallelism is the block: a 64-lane wavefront owns one block and its 8192 x u16 table in LDS, and uses its lanes
// for what is parallel inside a block -- the probes of a search (their positions depend only on the skip schedule,
// lane j runs probe j), match extension (byte compares + ballot), literal and length-byte stores.
// Output goes straight to the block's slot in HBM (dst + i*dst_stride); sizes[i] receives the length.
//
// Kernels, in launch order (DESIGN.md 4.3 has the measurements):
//   1. scan   -- for every block the probe sequence the serial parser performs while it finds NO match.  No probe
//                matched => the output is one literal run, written here (every incompressible block, at HBM speed);
//                first hit => the block index is queued.  lz4_scan_span_kernel (power-of-two sizes 4..64 KiB, 64 KiB
//                spans, straight-line memory operations), lz4_scan_stream_kernel (other aligned sizes and span
//                tails), lz4_scan_kernel (unaligned: gathers).
//   2. parse  -- lz4_lanes_kernel: one block per LANE, the serial parser as it stands, tables in global memory -- tens of
//                thousands of chains instead of the 2,560 that LDS admits.  Blocks > 4 KiB, from 10-14 Ki queued blocks on: it
//                takes the whole queue.  Blocks <= 4 KiB, from 60 Ki blocks on: it runs BESIDE lz4_parse_kernel on a second
//                stream, both pulling from the scan's queue (one is bound by LDS capacity and chain latency, the other by
//                random memory lines: the rates add).
//                lz4_parse_kernel (everything else, and what the lanes leave): full parse of the queued blocks, one block per
//                wavefront; the table operation of a batch of items is one ds_mskor_rtn_b32 exchange whose lanes the LDS
//                applies in ascending order (verified per batch).  lz4_parse_fp_kernel: a diagnostic variant
//                (CW_LZ4_PARSE=fp).
//   3. redo   -- lz4_blocks_kernel: the first-generation parser (write/read-back collision detection, batch cut,
//                rollback), run on the blocks the parse kernel hands back when its lane-order check fails (never
//                observed; forced in the tests).
// All of them produce the serial parser's bytes.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "cw_device.h"
#include "lz_device.h"

#ifndef HEADW_GLOBAL
#define HEADW_GLOBAL 16
#endif
#ifndef HEADW_STAGED
#define HEADW_STAGED 64
#endif

namespace cw {

using lz::Around;
using lz::around;
using lz::copy_g2g;
using lz::rd32;
using lz::rd32x;
using lz::tab_exchange;

namespace {

constexpr uint32_t kTabBytes = (1u << 13) * 2; // 8192 x u16
constexpr uint32_t kStageMax = 16384;          // parse kernel: blocks up to this size are staged in LDS
constexpr uint32_t kNeedsParse = 0xFFFFFFFFu;  // sizes[] marker: scan kernel -> parse kernel
constexpr int kScanGroup = 16;                 // probe batches in flight per wavefront in the generic scan kernel
constexpr uint32_t kMinMatch = 4, kLastLiterals = 5, kMFLimit = 12;

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t hash13(uint32_t v) { return (v * 2654435761u) >> 19; }
__device__ __forceinline__ uint32_t ctz64(unsigned long long m) { return m ? (uint32_t)__builtin_ctzll(m) : 64u; }

// offset of the k-th probe of a search from its first probe: the parser advances by
// step = (64 + probes_so_far) >> 6 after the first two probes (LZ4_skipTrigger = 6).
__device__ __forceinline__ uint32_t probe_delta(uint32_t k)
{
    const uint32_t t = 62 + k, q = t >> 6, r = t & 63;
    return k ? 1 + q * (32 * (q - 1) + r + 1) : 0; // k = 0 gives q = 0 and would come out as 1
}

// wavefront copy LDS -> global, any alignment; long runs go as 16 B per lane
__device__ __forceinline__ void copy_out(uint8_t *__restrict__ g, const uint8_t *lds, uint32_t s, uint32_t len, uint32_t lane)
{
    if (len < 256) {
        for (uint32_t i = lane; i < len; i += 64) g[i] = lds[s + i];
        return;
    }
    const uint32_t head = (uint32_t)(0 - reinterpret_cast<uintptr_t>(g)) & 15u;
    if (lane < head) g[lane] = lds[s + lane];
    g += head; s += head; len -= head;
    const uint32_t nvec = len >> 4;
    for (uint32_t i = lane; i < nvec; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, lds + s + 16 * i, 16);
        *reinterpret_cast<uint4 *>(g + 16 * i) = v;
    }
    const uint32_t done = nvec << 4, tail = len - done;
    if (lane < tail) g[done + lane] = lds[s + done + lane];
}

// LZ4 length continuation: `extra` as a run of 255s closed by one byte < 255; returns bytes written
__device__ __forceinline__ uint32_t put_len(uint8_t *__restrict__ g, uint32_t extra, uint32_t lane)
{
    const uint32_t n255 = extra / 255;
    for (uint32_t i = lane; i < n255; i += 64) g[i] = 255;
    if (lane == 0) g[n255] = (uint8_t)(extra - n255 * 255);
    return n255 + 1;
}


__device__ __forceinline__ uint32_t ld32g(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4); // unaligned global_load_dword
    return v;
}

} // namespace

// ---------------------------------------------------------------------------------------------------
// Scan kernels: the no-match walk of the serial parser, all probes of a block in flight.
//
// A table entry is one 32-bit LDS word  epoch:4 | position:16 | fingerprint:12  inserted with a
// returning atomic max.  Positions only grow during a walk, so within an epoch "max" is the parser's
// overwrite, and the returned old word is the entry the parser would have read -- also between lanes of
// one batch that hit the same slot, PROVIDED the LDS applies same-address atomics of one instruction in
// ascending lane order.  That order is not architecturally promised, so it is checked rather than
// assumed: any other order hands some lane a candidate >= its own position, which the serial parser can
// never see, and such a block is simply queued for the parse kernel.
// The fingerprint is 12 further bits of the probe value's multiplicative hash: the parser's test
// read32(candidate) == read32(position) can only hold when the fingerprints agree, so the candidate's
// bytes are fetched from memory only on a fingerprint hit (~0.2 per incompressible 64 KiB block) instead
// of 64 random cache lines per batch.  The epoch makes entries of earlier blocks read as "empty"
// (candidate = position 0, as in the parser's zeroed table) so the table is re-zeroed once per 15 blocks.
// ---------------------------------------------------------------------------------------------------
struct ScanState {
    bool hit = false, maybe = false;
    uint32_t mcand = 0, mv = 0;
};

// One probe per lane, written without branches so that the LDS operations of several batches pipeline:
// insert (pos, v) and test it the way the parser would.  An inactive lane issues max(tab[0], 0), which can never
// change an entry.  split: issue() returns the atomic's old word, judge() folds it into the lane's state.
__device__ __forceinline__ uint32_t scan_issue(uint32_t *tab, uint32_t tag, uint32_t pos, uint32_t v, bool active)
{
    const uint32_t h = v * 2654435761u, fp = (h >> 7) & 0xFFFu;
    return atomicMax(&tab[active ? h >> 19 : 0u], active ? tag | (pos << 12) | fp : 0u);
}
__device__ __forceinline__ void scan_judge(uint32_t old, uint32_t epoch, uint32_t v0, uint32_t pos, uint32_t v, bool active, ScanState &st)
{
    const uint32_t fp = ((v * 2654435761u) >> 7) & 0xFFFu, cand = (old >> 12) & 0xFFFFu;
    const bool same = (old >> 28) == epoch;               // else: empty slot, the parser's candidate is position 0
    const bool fpm = active && same && (old & 0xFFFu) == fp;
    st.hit |= active && (same ? cand >= pos : v == v0);   // cand >= pos: atomics applied out of lane order
    st.hit |= fpm && st.maybe;                            // a second fingerprint hit before settling: let the parser decide
    st.mcand = fpm ? cand : st.mcand;
    st.mv = fpm ? v : st.mv;
    st.maybe |= fpm;
}

// rare: a fingerprint agreed -- settle it on the candidate's actual bytes; returns the wave-wide verdict
__device__ __forceinline__ bool scan_settle(const uint8_t *g, ScanState &st)
{
    if (__ballot(st.maybe)) {
        if (st.maybe) st.hit |= ld32g(g + st.mcand) == st.mv;
        st.maybe = false;
    }
    return __ballot(st.hit) != 0;
}

__device__ __forceinline__ void scan_begin_block(uint32_t *tab, uint32_t &epoch, uint32_t lane)
{
    if (++epoch == 16) { // tags exhausted: start over on a clean table
        for (uint32_t i = lane; i < (1u << 13) / 4; i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0, 0, 0, 0);
        epoch = 1;
    }
}

// Next block for this wavefront from the shared counter (counters[2]).  Blocks are handed out dynamically because
// the scan's workgroups become resident at different times when a hash kernel is filling the same CUs: with a
// static partition a late workgroup still owns its full share and the kernel grows a tail.  Lane 0 pulls, the
// index reaches the other lanes through an LDS word (see the parse kernel for why not through readfirstlane).
// One shared word takes ~88 pulls per microsecond, so a pull hands out a run of blocks worth ~64 KiB of input.
struct BlockFeed {
    size_t next = 0, end = 0;
};
__device__ __forceinline__ bool scan_next_block(BlockFeed &f, size_t &blk, uint32_t *counters, volatile uint32_t *mailbox,
                                                uint32_t run, size_t nblocks, uint32_t lane)
{
    if (f.next == f.end) {
        __syncthreads();
        if (lane == 0) *mailbox = atomicAdd(&counters[2], run);
        __syncthreads();
        f.next = __builtin_amdgcn_readfirstlane(*mailbox);
        f.end = f.next + run;
    }
    blk = f.next++;
    return blk < nblocks;
}

__device__ __forceinline__ void scan_mark(uint32_t *sizes, size_t blk, uint32_t *queue, uint32_t *counters, uint32_t lane)
{
    if (lane == 0) {
        sizes[blk] = kNeedsParse;
        queue[atomicAdd(&counters[1], 1u)] = (uint32_t)blk; // counters[1] = queue tail
    }
}

// Probes gather their 4 bytes from global memory (kScanGroup batches in flight per wavefront); a block in which
// none matched is written as one literal run (header bytes + one global->global copy).
__global__ void __launch_bounds__(64)
lz4_scan_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                        uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes, uint32_t nprobes,
                        uint32_t *__restrict__ queue, uint32_t *__restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[1u << 13];
    __shared__ __attribute__((aligned(16))) uint32_t mailbox_word[4];
    volatile uint32_t *mailbox = mailbox_word;
    const uint32_t lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3); // latency-bound wavefront next to ALU-bound hash wavefronts
    uint32_t epoch = 15;           // forces a clean table before the first block

    BlockFeed feed;
    const uint32_t run = n >= 65536 ? 1u : 65536u / n; // blocks per pull
    for (size_t blk; scan_next_block(feed, blk, counters, mailbox, run, nblocks, lane);) {
        scan_begin_block(tab, epoch, lane);
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;
        const uint32_t tag = epoch << 28;
        ScanState st;
        bool marked = false;

        if (nprobes) {
            const uint32_t v0 = ld32g(g);
            {   // position 0 (LZ4_putPosition of the first bytes)
                const uint32_t h0 = v0 * 2654435761u;
                if (lane == 0) atomicMax(&tab[h0 >> 19], tag | ((h0 >> 7) & 0xFFFu));
            }
            for (uint32_t k0 = 0; k0 < nprobes && !marked; k0 += 64 * kScanGroup) {
                uint32_t pos[kScanGroup], v[kScanGroup];
                bool act[kScanGroup];
#pragma unroll
                for (int j = 0; j < kScanGroup; j++) {
                    const uint32_t k = k0 + 64 * j + lane;
                    pos[j] = 1 + probe_delta(k);
                    act[j] = k < nprobes;
                    v[j] = ld32g(g + (act[j] ? pos[j] : 0u)); // unconditional: the gathers of the whole group fly together
                }
                uint32_t old[kScanGroup];
#pragma unroll
                for (int j = 0; j < kScanGroup; j++) old[j] = scan_issue(tab, tag, pos[j], v[j], act[j]);
#pragma unroll
                for (int j = 0; j < kScanGroup; j++) scan_judge(old[j], epoch, v0, pos[j], v[j], act[j], st);
                marked = scan_settle(g, st);
            }
        }
        if (marked) {
            scan_mark(sizes, blk, queue, counters, lane);
            continue;
        }
        // no probe matched: the whole block is one literal run
        uint32_t op = 1;
        if (n >= 15) {
            if (lane == 0) out[0] = 15u << 4;
            op += put_len(out + 1, n - 15, lane);
        } else if (lane == 0) {
            out[0] = (uint8_t)(n << 4);
        }
        copy_g2g(out + op, g, n, lane);
        if (lane == 0) sizes[blk] = op + n;
    }
}

// ---------------------------------------------------------------------------------------------------
// Streaming scan (16-byte aligned source, n a multiple of 16): the block crosses the memory system ONCE.
// The wavefront walks the block in 4 KiB chunks (4 coalesced 16-byte loads per lane), three register sets in
// rotation: chunk c+2 is being loaded while chunk c is written into a two-chunk LDS ring and the probes
// that fall into chunk c-1 read their 4 bytes from that ring (an LDS round trip instead of a gather from
// memory; a probe may straddle into the next chunk, which is why probing lags staging by one chunk).  When
// Chunks are stored from their registers to their place in the literal run (the run's header shifts them by
// 258 bytes, hence unaligned 16-byte stores).  A block is abandoned and queued for the parser at its first
// hit, so compressible data costs a chunk or two of reads and at most one chunk of wasted stores here.
// LDS: 32 KiB table + 8 KiB ring = 40 KiB -> 4 wavefronts per CU.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kChunk = 4096, kPieces = kChunk / 1024, kRing = 2 * kChunk, kStreamGroup = 2;

struct Chunk { uint4 p[kPieces]; }; // 4 KiB of the block across the wavefront: piece j, lane L = bytes [j*1024 + L*16, +16)

__device__ __forceinline__ void chunk_load(Chunk &r, const uint8_t *g, uint32_t c, uint32_t nchunks, uint32_t n, uint32_t lane)
{
#pragma unroll
    for (uint32_t j = 0; j < kPieces; j++) {
        const uint32_t off = c * kChunk + j * 1024 + lane * 16;
        if (c < nchunks && off < n) r.p[j] = *reinterpret_cast<const uint4 *>(g + off);
    }
}
__device__ __forceinline__ void chunk_stage(const Chunk &r, uint32_t *ring32, uint32_t c, uint32_t n, uint32_t lane)
{
#pragma unroll
    for (uint32_t j = 0; j < kPieces; j++) {
        const uint32_t off = c * kChunk + j * 1024 + lane * 16;
        if (off < n) *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(ring32) + (off & (kRing - 1))) = r.p[j];
    }
}
__device__ __forceinline__ void chunk_store(const Chunk &r, uint8_t *lit, uint32_t c, uint32_t n, uint32_t lane)
{
#pragma unroll
    for (uint32_t j = 0; j < kPieces; j++) {
        const uint32_t off = c * kChunk + j * 1024 + lane * 16;
        if (off < n) { // four dword stores at a 2-byte-misaligned address: hipcc merges them into one global_store_dwordx4
            uint32_t *q = reinterpret_cast<uint32_t *>(lit + off);
            __builtin_nontemporal_store(r.p[j].x, q);
            __builtin_nontemporal_store(r.p[j].y, q + 1);
            __builtin_nontemporal_store(r.p[j].z, q + 2);
            __builtin_nontemporal_store(r.p[j].w, q + 3);
        }
    }
}

struct Walk { // progress of one block's no-match walk
    ScanState st;
    uint32_t knext = 0, v0 = 0; // first probe that has not run yet; bytes 0..3 of the block
    bool marked = false;
};

// NG batches of 64 probes: all ring reads, then all table exchanges, then all verdicts (so that the LDS round trips
// of the batches overlap)
template <int NG>
__device__ __forceinline__ void probe_groups(Walk &w, const uint32_t (&pos)[kStreamGroup], const bool (&act)[kStreamGroup], uint32_t *tab,
                                             const uint32_t *ring32, uint32_t tag, uint32_t epoch)
{
    uint32_t v[NG], old[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) {
        // unaligned 4 bytes out of the ring: two aligned dwords (the second may wrap) + byte align; read
        // unconditionally (any ring address is safe) so that the reads of the whole group pipeline
        const uint32_t a0 = (pos[j] & (kRing - 1)) >> 2, a1 = (a0 + 1) & (kRing / 4 - 1);
        v[j] = __builtin_amdgcn_alignbyte(ring32[a1], ring32[a0], pos[j] & 3u);
    }
#pragma unroll
    for (int j = 0; j < NG; j++) old[j] = scan_issue(tab, tag, pos[j], v[j], act[j]);
#pragma unroll
    for (int j = 0; j < NG; j++) scan_judge(old[j], epoch, w.v0, pos[j], v[j], act[j], w.st);
}

// run every probe whose position is < end (their bytes, incl. a straddle of up to 3, are in the ring)
__device__ __forceinline__ void probe_upto(Walk &w, uint32_t end, uint32_t *tab, const uint32_t *ring32, uint32_t tag, uint32_t epoch,
                                           uint32_t nprobes, const uint8_t *g, uint32_t lane)
{
    for (;;) {
        uint32_t pos[kStreamGroup], nact = 0, ngroups = 0;
        bool act[kStreamGroup];
#pragma unroll
        for (uint32_t j = 0; j < kStreamGroup; j++) {
            const uint32_t k = w.knext + 64 * j + lane;
            pos[j] = 1 + probe_delta(k);
            act[j] = k < nprobes && pos[j] < end;
            const unsigned long long m = __ballot(act[j]);
            nact += (uint32_t)__builtin_popcountll(m);
            ngroups += m != 0; // probes ascend with k: the batches that have any form a prefix
        }
        if (nact == 0) return;
        // late chunks hold fewer than 128 probes (the parser's step has grown): do not run empty batches
        if (ngroups == 1) probe_groups<1>(w, pos, act, tab, ring32, tag, epoch);
        else if (ngroups == 2) probe_groups<2>(w, pos, act, tab, ring32, tag, epoch);
        else probe_groups<kStreamGroup>(w, pos, act, tab, ring32, tag, epoch);
        w.marked = scan_settle(g, w.st);
        w.knext += nact;
        if (w.marked || nact < 64 * kStreamGroup) return;
    }
}

__global__ void __launch_bounds__(64)
lz4_scan_stream_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                       uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes, uint32_t nprobes,
                       uint32_t *__restrict__ queue, uint32_t *__restrict__ counters, size_t first_block, uint32_t feed_word)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[1u << 13];
    __shared__ __attribute__((aligned(16))) uint32_t ring32[kRing / 4];
    volatile uint32_t *mailbox = ring32; // free between two blocks
    const uint32_t lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    uint32_t epoch = 15;
    const uint32_t nchunks = (n + kChunk - 1) / kChunk;
    const uint32_t hdr = 1 + (n >= 15 ? (n - 15) / 255 + 1 : 0); // token + length bytes of the single literal run

    BlockFeed feed;
    const uint32_t run = n >= 65536 ? 1u : 65536u / n; // blocks per pull
    // blocks [first_block, nblocks): the span kernel may have taken the blocks in front; feed counter = counters[feed_word]
    for (size_t rel; scan_next_block(feed, rel, counters + feed_word - 2, mailbox, run, nblocks - first_block, lane);) {
        const size_t blk = first_block + rel;
        scan_begin_block(tab, epoch, lane);
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride, *lit = out + hdr;
        const uint32_t tag = epoch << 28;
        Walk w;
        Chunk r0, r1, r2; // chunk c lives in register set c % 3

        // One pipeline stage: chunk c is in CUR, chunk c-1 in PREV.  PREV is stored first -- speculatively, its probes
        // run at the end of this stage; a block that is queued after all is overwritten by the parser and wastes at
        // most this one chunk of writes -- so that its registers can take the loads of chunk c+2 now, two stages
        // before they are needed.
#define CW_STAGE(PREV, CUR, C)                                                                         \
        do {                                                                                           \
            const uint32_t c_ = (C);                                                                   \
            if (c_) chunk_store(PREV, lit, c_ - 1, n, lane);                                           \
            chunk_load(PREV, g, c_ + 2, nchunks, n, lane);                                             \
            chunk_stage(CUR, ring32, c_, n, lane);                                                     \
            if (c_ == 0) {                                                                             \
                w.v0 = ring32[0];                                                                      \
                if (nprobes && lane == 0) {                                                            \
                    const uint32_t h0 = w.v0 * 2654435761u;                                            \
                    atomicMax(&tab[h0 >> 19], tag | ((h0 >> 7) & 0xFFFu)); /* position 0 */            \
                }                                                                                      \
            } else {                                                                                   \
                probe_upto(w, c_ * kChunk, tab, ring32, tag, epoch, nprobes, g, lane);                 \
            }                                                                                          \
        } while (0)

        chunk_load(r0, g, 0, nchunks, n, lane);
        chunk_load(r1, g, 1, nchunks, n, lane);
        for (uint32_t c = 0; c < nchunks && !w.marked; c += 3) {
            CW_STAGE(r2, r0, c);
            if (c + 1 < nchunks && !w.marked) CW_STAGE(r0, r1, c + 1);
            if (c + 2 < nchunks && !w.marked) CW_STAGE(r1, r2, c + 2);
        }
#undef CW_STAGE
        if (!w.marked) { // the last chunk: its probes end 12 bytes before the block does, nothing straddles
            probe_upto(w, n, tab, ring32, tag, epoch, nprobes, g, lane);
            if (!w.marked) {
                const uint32_t last = (nchunks - 1) % 3;
                if (last == 0) chunk_store(r0, lit, nchunks - 1, n, lane);
                else if (last == 1) chunk_store(r1, lit, nchunks - 1, n, lane);
                else chunk_store(r2, lit, nchunks - 1, n, lane);
            }
        }
        if (w.marked) {
            scan_mark(sizes, blk, queue, counters, lane);
            continue;
        }
        if (n >= 15) {
            if (lane == 0) out[0] = 15u << 4;
            put_len(out + 1, n - 15, lane);
        } else if (lane == 0) {
            out[0] = (uint8_t)(n << 4);
        }
        if (lane == 0) sizes[blk] = hdr + n;
    }
}

// ---------------------------------------------------------------------------------------------------
// Span scan: the streaming scan for power-of-two block sizes 4 KiB .. 64 KiB, written so that hipcc can count its
// memory operations.  In lz4_scan_stream_kernel every chunk load/store sits in a bounds branch ("off < n"), and
// hipcc answers a load inside a branch with s_waitcnt vmcnt(0) at the next use of ANY loaded value: the wait in
// front of staging chunk c also waited for the loads of chunk c+2 issued a few instructions earlier, so nothing was
// ever in flight across a stage.  Here a wavefront pulls a SPAN of 16 chunks = 64 KiB (one 64 KiB block, or 16
// consecutive 4 KiB blocks, ...): every chunk is full, every load and store is unconditional (the two loads past the
// span re-read its last chunk), the chunk pipeline runs across the block boundaries inside a span (so 4 KiB blocks
// are pipelined at all), and the waits come out as vmcnt(N) with the next chunks still in flight.
// Step i: store chunk i-1 (speculatively, as before), load chunk i+2 into the registers that frees, finish the
// previous block if chunk i starts a new one, stage chunk i in the LDS ring, probe the positions below it.
// ---------------------------------------------------------------------------------------------------
#ifdef CW_CLOCK_STAMP
__device__ unsigned long long g_clock_scan[4 * kClockSlots];
hipError_t lz4_clock_read(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_scan), sizeof g_clock_scan); }
#endif

__global__ void __launch_bounds__(64)
lz4_scan_span_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint32_t nspans, uint8_t *__restrict__ dst,
                     size_t dst_stride, uint32_t *__restrict__ sizes, uint32_t nprobes, uint32_t *__restrict__ queue,
                     uint32_t *__restrict__ counters, uint32_t lg)
{
    CW_CLOCK_SCOPE(g_clock_scan);
    __shared__ __attribute__((aligned(16))) uint32_t tab[1u << 13];
    __shared__ __attribute__((aligned(16))) uint32_t ring32[kRing / 4];
    volatile uint32_t *mailbox = ring32; // free between two spans
    const uint32_t lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    uint32_t epoch = 15;
    const uint32_t cmask = (1u << lg) - 1, run = 16u >> lg; // chunks per block - 1, blocks per span
    const uint32_t hdr = 1 + (n - 15) / 255 + 1;            // token + length bytes of the single literal run (n >= 4096)
    const uint32_t lane16 = lane * 16;

    for (;;) {
        __syncthreads();
        if (lane == 0) *mailbox = atomicAdd(&counters[2], 1u);
        __syncthreads();
        const uint32_t span = __builtin_amdgcn_readfirstlane(*mailbox);
        if (span >= nspans) break;
        __syncthreads(); // the mailbox word is ring memory
        const size_t first = (size_t)span * run;
        const uint8_t *gs = src + first * src_stride;
        uint8_t *os = dst + first * dst_stride + hdr;

        Walk w;
        size_t blk = first;
        uint32_t tag = 0, marks = 0; // marks: bit b = block first + b of the span has a match
        Chunk r0, r1, r2;
#define CW_LD(R, I)                                                                                          \
        do {                                                                                                 \
            const uint32_t i_ = (I) < 15u ? (I) : 15u;                                                       \
            const uint8_t *p_ = gs + (size_t)(i_ >> lg) * src_stride + ((i_ & cmask) << 12) + lane16;        \
            _Pragma("unroll") for (uint32_t j = 0; j < kPieces; j++) R.p[j] = *reinterpret_cast<const uint4 *>(p_ + j * 1024); \
        } while (0)
#define CW_ST(R, I)                                                                                          \
        do {                                                                                                 \
            const uint32_t i_ = (I);                                                                         \
            uint8_t *p_ = os + (size_t)(i_ >> lg) * dst_stride + ((i_ & cmask) << 12) + lane16;              \
            _Pragma("unroll") for (uint32_t j = 0; j < kPieces; j++) {                                       \
                uint32_t *q = reinterpret_cast<uint32_t *>(p_ + j * 1024); /* 2-byte-misaligned: 4 dword stores, merged by hipcc */ \
                __builtin_nontemporal_store(R.p[j].x, q);     __builtin_nontemporal_store(R.p[j].y, q + 1);  \
                __builtin_nontemporal_store(R.p[j].z, q + 2); __builtin_nontemporal_store(R.p[j].w, q + 3);  \
            }                                                                                                \
        } while (0)
#define CW_FINISH()                                                                                          \
        do { /* the block's last probes (nothing straddles its end), then its verdict */                     \
            if (!w.marked) probe_upto(w, n, tab, ring32, tag, epoch, nprobes, src + blk * src_stride, lane); \
            if (w.marked) marks |= 1u << (uint32_t)(blk - first); /* queued at the end of the span */        \
            else {                                                                                           \
                uint8_t *out = dst + blk * dst_stride;                                                       \
                if (lane == 0) out[0] = 15u << 4;                                                            \
                put_len(out + 1, n - 15, lane);                                                              \
                if (lane == 0) sizes[blk] = hdr + n;                                                         \
            }                                                                                                \
        } while (0)
#define CW_BEGIN(I)                                                                                          \
        do {                                                                                                 \
            blk = first + ((I) >> lg);                                                                       \
            scan_begin_block(tab, epoch, lane);                                                              \
            tag = epoch << 28;                                                                               \
            w = Walk();                                                                                      \
        } while (0)
#define CW_STAGE_IN(R, C)                                                                                    \
        do {                                                                                                 \
            uint8_t *p_ = reinterpret_cast<uint8_t *>(ring32) + (((C) & 1u) << 12) + lane16;                 \
            _Pragma("unroll") for (uint32_t j = 0; j < kPieces; j++) *reinterpret_cast<uint4 *>(p_ + j * 1024) = R.p[j]; \
        } while (0)
#define CW_AFTER_STAGE(C)                                                                                    \
        do {                                                                                                 \
            if ((C) == 0) {                                                                                  \
                w.v0 = ring32[0];                                                                            \
                if (nprobes && lane == 0) {                                                                  \
                    const uint32_t h0 = w.v0 * 2654435761u;                                                  \
                    atomicMax(&tab[h0 >> 19], tag | ((h0 >> 7) & 0xFFFu)); /* position 0 */                  \
                }                                                                                            \
            } else if (!w.marked) {                                                                          \
                probe_upto(w, (C) * kChunk, tab, ring32, tag, epoch, nprobes, src + blk * src_stride, lane); \
            }                                                                                                \
        } while (0)
#define CW_STEP(PREV, CUR, I)                                                                                \
        do {                                                                                                 \
            const uint32_t s_ = (I), c_ = s_ & cmask;                                                        \
            CW_ST(PREV, s_ - 1);                                                                             \
            CW_LD(PREV, s_ + 2);                                                                             \
            if (c_ == 0) { CW_FINISH(); CW_BEGIN(s_); }                                                      \
            CW_STAGE_IN(CUR, c_);                                                                            \
            CW_AFTER_STAGE(c_);                                                                              \
        } while (0)

        CW_LD(r0, 0u);
        CW_LD(r1, 1u);
        CW_LD(r2, 2u);
        CW_BEGIN(0u);
        CW_STAGE_IN(r0, 0u);
        CW_AFTER_STAGE(0u);
        for (uint32_t i = 1; i < 16; i += 3) {
            CW_STEP(r0, r1, i);
            CW_STEP(r1, r2, i + 1);
            CW_STEP(r2, r0, i + 2);
        }
        CW_ST(r0, 15u);
        CW_FINISH();
        // The span's blocks with a match go onto the queue with ONE fetch-and-add (scan_mark's add per block returns the slot the block is
        // stored to, so the wavefront waits for it: 16 such round trips per span of 4 KiB blocks, all on one address -- 12.8 ms of scan per Mi
        // compressible blocks of 4 KiB, against 1.8 ms for the same bytes in 64 KiB blocks)
        if (marks) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&counters[1], (uint32_t)__builtin_popcount(marks)); // counters[1] = queue tail
            base = __builtin_amdgcn_readfirstlane(base);
            if (lane < 16 && ((marks >> lane) & 1u)) {
                sizes[first + lane] = kNeedsParse;
                queue[base + (uint32_t)__builtin_popcount(marks & ((1u << lane) - 1u))] = (uint32_t)(first + lane);
            }
        }
#undef CW_LD
#undef CW_ST
#undef CW_FINISH
#undef CW_BEGIN
#undef CW_STAGE_IN
#undef CW_AFTER_STAGE
#undef CW_STEP
    }
}

// number of probes of a no-match walk over n bytes (host)
static uint32_t scan_probes(uint32_t n)
{
    if (n < kMFLimit + 1) return 0;
    const uint32_t limit = n - 11;
    uint32_t k = 0, p = 1, step = 1, nb = 64; // probe k sits at p; it runs iff the next position <= limit
    while (p + step <= limit) { p += step; step = nb++ >> 6; k++; }
    return k;
}

// Parse kernel: the full greedy parse of the blocks the scan queued (queue[0 .. counters[1]); counters[0] is the
// shared head the workgroups pull from, so a handful of queued blocks spreads over as many workgroups).
template <bool STAGED>
__global__ void __launch_bounds__(64)
lz4_blocks_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                  uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes,
                  const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters)
{
    // LDS: the 16 KiB position table, then (STAGED) the block itself.  Small blocks are staged: every read of
    // the parse is then an LDS read.  Large blocks are read through L1/L2 instead: with only the table in LDS ten
    // blocks fit a CU instead of two, and the parse -- one wavefront per block, bound by its own instruction
    // issue and dependent round trips -- gains more from the extra wavefronts than it loses to the longer reads.
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    uint8_t *lds_in = smem + kTabBytes;
    const uint32_t lane = threadIdx.x;

    const uint32_t qcount = counters[1];
    volatile uint32_t *mailbox = reinterpret_cast<volatile uint32_t *>(tab); // table memory, before it is cleared
    for (uint32_t guard = 0; guard <= qcount; guard++) { // each pull advances the shared head; never more than qcount+1 pulls
        // one lane pulls the next queue index and hands it to the wavefront through LDS (a divergent branch
        // feeding readfirstlane directly was mis-structured by hipcc into a loop that never re-pulled)
        __syncthreads(); // previous block's LDS readers are done
        if (lane == 0) *mailbox = atomicAdd(&counters[0], 1u);
        __syncthreads();
        const uint32_t qi = __builtin_amdgcn_readfirstlane(*mailbox);
        if (qi >= qcount) break;
        const size_t blk = queue[qi];
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;

        // ---- stage the block into LDS (coalesced 16 B per lane) and clear the table ----
        __syncthreads(); // previous block's readers are done
        // `in` is purely an LDS pointer or purely a global one (template), so every read below is a ds_read or a
        // global_load, never a flat access
        const uint8_t *in = STAGED ? lds_in : g;
        if (STAGED) {
            if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
                const uint4 *g4 = reinterpret_cast<const uint4 *>(g);
                const uint32_t nvec = n >> 4;
                for (uint32_t i = lane; i < nvec; i += 64) reinterpret_cast<uint4 *>(lds_in)[i] = g4[i];
                for (uint32_t i = (nvec << 4) + lane; i < n; i += 64) lds_in[i] = g[i];
            } else {
                for (uint32_t i = lane; i < n; i += 64) lds_in[i] = g[i];
            }
        }
        for (uint32_t i = lane; i < kTabBytes / 16; i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();

        uint32_t ip = 0, anchor = 0, op = 0;

        if (n >= kMFLimit + 1) {
            const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals;
            // What the serial parser does between two sequences is a fixed list of table operations in
            // position order: [insert ip-2] [re-test ip] probe s0, probe s0+1, ... (skip schedule).  Item t of
            // that list goes to lane t - t0, so ONE round of LDS reads fetches every value the list needs, one
            // round does the table reads+writes, one the candidate compares:
            //   t = 0      insert `ins` (write only)         -- position 0 at the start of the block, else ip-2
            //   t = 1      re-test of position ip            -- only after a match (a hit = sequence with no literals)
            //   t >= 2     probe k = t-2 of the search starting at s0 (= ip+1 after a match, 1 at the start)
            uint32_t ins = 0, s0 = 1;
            bool has_retest = false;

            for (uint32_t seq = 0; seq < n; seq++) { // one iteration per emitted sequence (a block has < n of them)
                uint32_t match = 0, mpos = 0;
                bool found = false, exhausted = false;
                for (uint32_t t0 = 0;;) {
                    const uint32_t t = t0 + lane, k = t - 2;
                    const bool is_probe = t >= 2;
                    // the first 65 probes of a search advance by 1, so the head batch needs no closed form
                    const uint32_t dk = t0 == 0 ? k : probe_delta(k), stepk = k == 0 ? 1u : (63u + k) >> 6;
                    const uint32_t pos = t == 0 ? ins : t == 1 ? ip : s0 + dk;
                    // a probe runs iff the position after it stays <= mflimit + 1; later probes are dead
                    const bool dead = is_probe && s0 + dk + stepk > mflimit + 1;
                    const bool active = !dead && (is_probe || t == 0 || has_retest);
                    const uint32_t ndead = ctz64(__ballot(dead)); // first dead lane (64 = none)
                    const unsigned long long amask = __ballot(active);
                    if (!amask) { exhausted = true; break; }

                    uint32_t v = 0, h = 0, old = 0;
                    if (active) {
                        v = rd32x<STAGED>(in, pos);
                        h = hash13(v);
                        old = tab[h];
                        tab[h] = (uint16_t)pos;
                    }
                    // a lane whose slot was overwritten by another lane of this batch shares its hash
                    // (the barrier keeps hipcc from forwarding the lane's own store to the read-back)
                    __syncthreads();
                    const bool lost = active && (tab[h] != (uint16_t)pos);
                    const uint32_t fa = ctz64(amask);
                    uint32_t m = ctz64(__ballot(lost));
                    if (m == fa) m = fa + 1; // the first active lane read the table before any write of this batch
                    const uint32_t L = m < ndead ? m : ndead;

                    const bool eq = active && t != 0 && lane < L && rd32x<STAGED>(in, old) == v;
                    const unsigned long long eqmask = __ballot(eq);
                    const uint32_t ncommit = eqmask ? ctz64(eqmask) + 1 : L;
                    // undo speculative writes beyond the committed lanes, then re-assert the committed ones
                    if (__ballot(active && lane >= ncommit)) {
                        if (active && lane >= ncommit) tab[h] = (uint16_t)old;
                        if (active && lane < ncommit) tab[h] = (uint16_t)pos;
                    }
                    if (eqmask) {
                        const uint32_t w = ctz64(eqmask);
                        mpos = __builtin_amdgcn_readlane(pos, w);
                        match = __builtin_amdgcn_readlane(old, w);
                        found = true;
                        break;
                    }
                    if (ndead < 64 && L == ndead) { exhausted = true; break; } // the next probe would pass the end of the block
                    t0 += L;
                }
                (void)exhausted;
                if (!found) break; // -> last literals
                ip = mpos;

                // ---- one round for both directions: lanes 0..47 extend forwards, lanes 48..63 backwards ----
                uint32_t mc, back;
                {
                    bool ok;
                    if (lane < 48) {
                        const uint32_t i = ip + kMinMatch + lane;
                        ok = i < matchlimit && in[i] == in[match + kMinMatch + lane];
                    } else {
                        const uint32_t j = lane - 47;
                        ok = ip >= anchor + j && match >= j && in[ip - j] == in[match - j];
                    }
                    const unsigned long long okm = __ballot(ok);
                    mc = ctz64(~okm | (1ull << 48)); // trailing ones within lanes 0..47
                    back = ctz64(~(okm >> 48));
                }
                if (mc == 48) { // long match: keep counting, 64 bytes per round
                    for (;;) {
                        const uint32_t i = ip + kMinMatch + mc + lane;
                        const bool ok = i < matchlimit && in[i] == in[match + kMinMatch + mc + lane];
                        const uint32_t cnt = ctz64(~__ballot(ok));
                        mc += cnt;
                        if (cnt < 64) break;
                    }
                }
                if (back == 16) { // long catch-up (rare)
                    for (;;) {
                        const uint32_t j = back + lane + 1;
                        const bool ok = ip >= anchor + j && match >= j && in[ip - j] == in[match - j];
                        const uint32_t cnt = ctz64(~__ballot(ok));
                        back += cnt;
                        if (cnt < 64) break;
                    }
                }
                const uint32_t mend = ip + kMinMatch + mc; // first byte after the match
                ip -= back; match -= back; mc += back;     // LZ4_count restarts 4 bytes after the moved-back start

                // ---- emit: literals [anchor, ip), offset, match length ----
                if (ip < anchor || mend > n) { op = 0; anchor = n; break; } // cannot happen; never write out of bounds
                const uint32_t lit = ip - anchor, tok_pos = op;
                uint32_t token;
                op += 1;
                if (lit >= 15) { token = 15u << 4; op += put_len(out + op, lit - 15, lane); }
                else token = lit << 4;
                copy_out(out + op, in, anchor, lit, lane);
                op += lit;
                const uint32_t off = ip - match, off_pos = op;
                op += 2;
                if (mc >= 15) { token += 15; op += put_len(out + op, mc - 15, lane); }
                else token += mc;
                if (lane < 3) { // token and the two offset bytes: three lanes, one store instruction
                    const uint32_t where = lane == 0 ? tok_pos : off_pos + lane - 1;
                    const uint32_t what = lane == 0 ? token : lane == 1 ? off : off >> 8;
                    out[where] = (uint8_t)what;
                }

                ip = mend;
                anchor = ip;
                if (ip > mflimit) break; // end of parse: remaining bytes are literals
                ins = ip - 2; s0 = ip + 1; has_retest = true;
            }
        }

        // ---- last literals ----
        {
            const uint32_t run = n - anchor;
            const uint32_t tok_pos = op;
            op += 1;
            if (run >= 15) {
                if (lane == 0) out[tok_pos] = 15u << 4;
                op += put_len(out + op, run - 15, lane);
            } else if (lane == 0) {
                out[tok_pos] = (uint8_t)(run << 4);
            }
            copy_out(out + op, in, anchor, run, lane);
            op += run;
        }
        if (lane == 0) sizes[blk] = op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Parse kernel, second generation: the same item list per sequence, but the table operation of a batch is ONE
// LDS instruction -- ds_mskor_rtn_b32, a masked exchange of the slot's 16 bits inside its 32-bit word that returns
// the previous content.  The LDS applies the lanes of one such instruction in ascending lane order, so a lane gets
// back exactly what the serial parser would have read at its item, INCLUDING the writes of earlier lanes of the
// same batch: no write/read-back round, no cutting of batches at colliding lanes, one batch per sequence.  That
// order is measured (tools/mskor_order.hip: 0 of 1.5 M same-slot successors out of order) but not architecturally
// promised, so it is checked on every batch: any other order hands some lane a candidate >= its own position,
// which the serial parser can never see, and such a block is requeued for lz4_blocks_kernel above, which does not
// depend on it.  Speculative writes of the lanes behind the first match are undone by the first lane of each
// slot's group (its returned value is the slot's content before the group).
// Round trips per sequence: (1) the 4 bytes at every item of the next batch -- requested as soon as the match end
// is known, before the sequence is emitted; they also ARE the next literals, which are stored from registers;
// (2) the table exchange; (3) the candidates' 4 bytes together with 8 bytes after / 4 bytes before both the
// position and the candidate, so that matches up to 11 bytes and catch-ups up to 3 bytes (most of them) need no
// further round.
// ---------------------------------------------------------------------------------------------------
// raw head values of a search: two aligned dwords + byte shift (STAGED) or the value itself
struct HeadRaw { uint32_t lo, hi, sh; };
template <bool STAGED>
__device__ __forceinline__ HeadRaw head_request(const uint8_t *in, uint32_t n, uint32_t ins, uint32_t s0, uint32_t lane)
{
    uint32_t p = lane == 0 ? ins : s0 - 2 + lane;
    p = p <= n - 8 ? p : n - 8; // only positions <= n - 12 are ever used
    HeadRaw r;
    if (STAGED) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + (p >> 2);
        r.lo = w[0]; r.hi = w[1]; r.sh = p & 3u;
    } else {
        r.lo = rd32(in, p); r.hi = 0; r.sh = 0;
    }
    return r;
}

// result of one batch of items
struct BatchOut { uint32_t mpos, match, mc, back; bool stop, found, broken, flong, blong; };

// exchange, lane-order check, candidate test for one batch of items (lane = item)
template <bool STAGED>
__device__ __forceinline__ BatchOut run_batch(const uint8_t *in, uint16_t *tab, uint32_t tab_lds, uint32_t pos, uint32_t v, bool active,
                                             bool tested, uint32_t anchor, uint32_t matchlimit, uint32_t lane)
{
    BatchOut r;
    r.mpos = 0; r.match = 0; r.mc = 0; r.back = 0; r.stop = false; r.found = false; r.broken = false; r.flong = false; r.blong = false;
    const uint32_t h = hash13(v);
    uint32_t old = 0;
    if (active) old = tab_exchange(tab_lds, h, pos);
    if (__ballot(tested && old >= pos)) { r.broken = true; r.stop = true; return r; }
    // candidate bytes, and the neighbourhood that settles short extensions in the same round
    const bool near_start = pos < 4 || old < 4;
    Around ap, ac;
    ap.before = 0; ap.at = 0; ap.after = 0; ac.before = 0; ac.at = 0; ac.after = 1;
    if (tested) {
        ac = around<STAGED>(in, old, !near_start);
        ap = around<STAGED>(in, pos, !near_start); // pos <= n - 12
    }
    const unsigned long long eqmask = __ballot(tested && ac.at == v);
    if (!eqmask) return r;
    const uint32_t w = (uint32_t)__builtin_ctzll(eqmask);
    const uint32_t pm = __builtin_amdgcn_readlane(pos | (old << 16), w);
    r.mpos = pm & 0xFFFFu; r.match = pm >> 16;
    // undo the speculative writes behind the match: the first lane of each slot's group got the content from before
    // the group (a value <= mpos); later lanes of a group got a position > mpos
    if (active && lane > w && old <= r.mpos) tab[h] = (uint16_t)old;
    // forward: bytes 4..11; backward: up to 3 (4 = keep going)
    const uint64_t x = ap.after ^ ac.after;
    uint32_t nf = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
    const uint32_t lim = matchlimit - (pos + 4);
    bool fl = false;
    if (nf >= lim) nf = lim; else fl = nf == 8;
    const uint32_t room = pos - anchor < old ? pos - anchor : old; // bytes the match may move back
    const uint32_t y = ap.before ^ ac.before;
    uint32_t nb = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
    bool bl = false;
    if (near_start) { nb = 0; bl = room != 0; }
    else if (nb >= room) nb = room;
    else bl = nb == 4;
    const uint32_t packed = __builtin_amdgcn_readlane(nf | (nb << 8) | ((uint32_t)fl << 16) | ((uint32_t)bl << 17), w);
    r.mc = packed & 0xFFu; r.back = (packed >> 8) & 0xFFu;
    r.flong = (packed >> 16) & 1u; r.blong = (packed >> 17) & 1u;
    r.found = true; r.stop = true;
    return r;
}

template <bool STAGED>
__global__ void __launch_bounds__(64)
lz4_parse_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                 uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes,
                 const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters, uint32_t *__restrict__ requeue,
                 uint32_t force_redo)
{
    // Width of the head batch.  A wavefront waits for the slowest lane of a round, and when the block is read from
    // global memory the candidates of far lanes are arbitrary earlier positions (L2 misses): items beyond the first
    // few are rarely needed (text: 0.9 literals per sequence at 64 KiB), so only kHead of them are speculated on.
    constexpr uint32_t kHead = STAGED ? HEADW_STAGED : HEADW_GLOBAL;
    constexpr unsigned long long kHeadMask = kHead >= 64 ? ~0ull : (1ull << (kHead & 63)) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    uint8_t *lds_in = smem + kTabBytes;
    const uint32_t tab_lds = (uint32_t)reinterpret_cast<uintptr_t>(tab); // LDS byte address of the table
    const uint32_t lane = threadIdx.x;

    const uint32_t qcount = counters[1];
    volatile uint32_t *mailbox = reinterpret_cast<volatile uint32_t *>(tab);
    for (uint32_t guard = 0; guard <= qcount; guard++) {
        __syncthreads();
        if (lane == 0) *mailbox = atomicAdd(&counters[0], 1u);
        __syncthreads();
        const uint32_t qi = __builtin_amdgcn_readfirstlane(*mailbox);
        if (qi >= qcount) break;
        const size_t blk = queue[qi];
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;

        __syncthreads();
        const uint8_t *in = STAGED ? lds_in : g;
        if (STAGED) {
            if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
                const uint4 *g4 = reinterpret_cast<const uint4 *>(g);
                const uint32_t nvec = n >> 4;
                for (uint32_t i = lane; i < nvec; i += 64) reinterpret_cast<uint4 *>(lds_in)[i] = g4[i];
                for (uint32_t i = (nvec << 4) + lane; i < n; i += 64) lds_in[i] = g[i];
            } else {
                for (uint32_t i = lane; i < n; i += 64) lds_in[i] = g[i];
            }
        }
        for (uint32_t i = lane; i < kTabBytes / 16; i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();

        uint32_t anchor = 0, op = 0;
        bool broken = force_redo != 0; // lane-order check failed (or an impossible state): hand the block to the other parser

        if (!broken && n >= kMFLimit + 1) {
            const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals;
            // Head batch of a search (almost always the only one): lane 0 inserts `ins`, lane 1 re-tests s0-1 (= ip,
            // after a match), lane t >= 2 probes s0 + t - 2 (the first 65 probes advance by 1).  The literals of the
            // sequence start at s0-1, so literal i is the low byte of lane i+1's value.  The values are requested as
            // two aligned dwords per lane (STAGED) and only combined when the search starts, so that the request
            // overlaps the emission of the previous sequence.
            uint32_t ins = 0, s0 = 1;
            unsigned long long lane1 = 0; // bit 1 set once there is a position to re-test
            HeadRaw hr = head_request<STAGED>(in, n, ins, s0, lane);

            for (uint32_t seq = 0; seq < n; seq++) {
                const uint32_t vhead = STAGED ? __builtin_amdgcn_alignbyte(hr.hi, hr.lo, hr.sh) : hr.lo;
                BatchOut bo;
                {   // head batch: all steps are 1, so an item is dead iff its position is past mflimit
                    const uint32_t pos = lane == 0 ? ins : s0 - 2 + lane;
                    const bool live = pos <= mflimit;
                    const unsigned long long amask = __ballot(live) & (lane1 | ~2ull) & kHeadMask;
                    const bool active = (amask >> lane) & 1u;
                    bo = run_batch<STAGED>(in, tab, tab_lds, pos, vhead, active, active && lane != 0, anchor, matchlimit, lane);
                    if (!bo.stop && (__ballot(!live) & kHeadMask)) bo.stop = true; // the next probe would pass the end of the block
                }
                if (!bo.stop) { // rare: more than kHead - 2 probes without a match
                    for (uint32_t t0 = kHead;; t0 += 64) {
                        const uint32_t k = t0 + lane - 2;
                        const uint32_t dk = probe_delta(k), stepk = (63u + k) >> 6;
                        const uint32_t p2 = s0 + dk;
                        const bool act2 = p2 + stepk <= mflimit + 1;
                        if (!__ballot(act2)) break;
                        const uint32_t v2 = rd32x<STAGED>(in, act2 ? p2 : 0u);
                        bo = run_batch<STAGED>(in, tab, tab_lds, p2, v2, act2, act2, anchor, matchlimit, lane);
                        if (bo.stop || __ballot(!act2)) break;
                    }
                }
                if (bo.broken) { broken = true; break; }
                if (!bo.found) break;
                uint32_t mpos = bo.mpos, match = bo.match, mc = bo.mc, back = bo.back;
                const bool flong = bo.flong, blong = bo.blong;

                if (flong) { // long match: keep counting, 64 bytes per round
                    for (;;) {
                        const uint32_t i = mpos + kMinMatch + mc + lane;
                        const bool ok = i < matchlimit && in[i] == in[match + kMinMatch + mc + lane];
                        const uint32_t cnt = ctz64(~__ballot(ok));
                        mc += cnt;
                        if (cnt < 64) break;
                    }
                }
                if (blong) { // long catch-up (rare)
                    for (;;) {
                        const uint32_t j = back + lane + 1;
                        const bool ok = mpos >= anchor + j && match >= j && in[mpos - j] == in[match - j];
                        const uint32_t cnt = ctz64(~__ballot(ok));
                        back += cnt;
                        if (cnt < 64) break;
                    }
                }
                const uint32_t mend = mpos + kMinMatch + mc; // first byte after the match
                const uint32_t ip = mpos - back;
                match -= back; mc += back;                   // LZ4_count restarts 4 bytes after the moved-back start
                if (ip < anchor || mend > n) { broken = true; break; } // cannot happen; never write out of bounds

                // request the next search's head values now; the stores below do not wait for them
                const bool more = mend <= mflimit;
                if (more) hr = head_request<STAGED>(in, n, mend - 2, mend + 1, lane);

                // ---- emit: literals [anchor, ip), offset, match length ----
                const uint32_t lit = ip - anchor, tok_pos = op;
                uint32_t token;
                op += 1;
                if (lit >= 15) { token = 15u << 4; op += put_len(out + op, lit - 15, lane); }
                else token = lit << 4;
                if (lane - 1 < lit) out[op + lane - 1] = (uint8_t)vhead; // literal i sits in lane i+1 (lane 0 wraps to "no")
                if (lit > 63) copy_out(out + op + 63, in, anchor + 63, lit - 63, lane);
                op += lit;
                const uint32_t off = ip - match, off_pos = op;
                op += 2;
                if (mc >= 15) { token += 15; op += put_len(out + op, mc - 15, lane); }
                else token += mc;
                if (lane < 3) { // token and the two offset bytes: three lanes, one store instruction
                    const uint32_t where = lane == 0 ? tok_pos : off_pos + lane - 1;
                    const uint32_t what = lane == 0 ? token : lane == 1 ? off : off >> 8;
                    out[where] = (uint8_t)what;
                }

                anchor = mend;
                if (!more) break; // end of parse: remaining bytes are literals
                ins = mend - 2; s0 = mend + 1; lane1 = 2;
            }
        }
        if (broken) {
            if (lane == 0) requeue[atomicAdd(&counters[5], 1u)] = (uint32_t)blk; // counters[4..5]: second queue head, tail
            continue;
        }

        // ---- last literals ----
        {
            const uint32_t run = n - anchor;
            const uint32_t tok_pos = op;
            op += 1;
            if (run >= 15) {
                if (lane == 0) out[tok_pos] = 15u << 4;
                op += put_len(out + op, run - 15, lane);
            } else if (lane == 0) {
                out[tok_pos] = (uint8_t)(run << 4);
            }
            copy_out(out + op, in, anchor, run, lane);
            op += run;
        }
        if (lane == 0) sizes[blk] = op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Parse kernel, third generation, for blocks that are read from global memory (> 4 KiB).
//
// What bounded the second generation at 64 KiB was its own speculation: every item of a batch fetched its candidate's
// bytes, 16 arbitrary earlier positions = 16 cache lines per sequence, of which the parser needs 1.9 on text; 2,560
// resident blocks have a 160 MiB working set, so those lines come over the fabric (about 4 TB/s of line fills at
// 14 GB/s of input) and the load latency grows with every wavefront added (2 -> 10 wavefronts per CU: 4.7 -> 14.1 GB/s).
// Here a 4-bit fingerprint of the inserted value (bits 15..18 of the multiplicative hash whose bits 19..31 are the
// slot) sits beside every table slot, 4 KiB per block, and goes through the same one-instruction exchange as the
// position: a candidate whose fingerprint differs cannot hold the same 4 bytes and is never fetched (15 of 16 are
// not), candidates at positions 0..3 are compared against the block's first 8 bytes kept in registers.  20 KiB of LDS
// per block = 8 blocks per CU instead of 10.
// Also: the 16 bytes [p-4, p+12) around every item of a head batch and around every fetched candidate are ONE 16-byte
// load each (before: 4 + 3 + 3 loads per sequence), and the winner's extension (forward <= 8, backward <= 3 bytes) is
// scalar code on values read from its lane.
// The fingerprint exchange relies on the same lane order as the position exchange; it is checked the same way: a
// lane whose candidate was inserted by an earlier lane of the same batch must have received that lane's fingerprint.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kFpBytes = (1u << 13) / 2; // 4-bit fingerprint per table slot

// Diagnostic build only (-DCW_STAMP, tools/parse_stamp.hip): where a sequence's cycles go.  A stamp is s_memtime behind a
// drained LDS/scalar queue; the differences are summed per phase in scalar registers and added to g_stamp once per block.
#ifdef CW_STAMP
__device__ unsigned long long g_stamp[16];
#define CW_STAMP_DECL uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_seq = 0; unsigned long long st_last = stamp_now();
#define CW_STAMP_AT(i) do { const unsigned long long t_ = stamp_now(); st_acc[i] += (uint32_t)(t_ - st_last); st_last = t_; } while (0)
#define CW_STAMP_FLUSH() do { if (lane == 0) { for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_stamp[i_], (unsigned long long)st_acc[i_]); atomicAdd(&g_stamp[15], (unsigned long long)st_seq); } } while (0)
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define CW_STAMP_DECL
#define CW_STAMP_AT(i) do { } while (0)
#define CW_STAMP_FLUSH() do { } while (0)
#endif

__device__ __forceinline__ uint32_t fp4(uint32_t v) { return ((v * 2654435761u) >> 15) & 0xFu; }
__device__ __forceinline__ uint4 ld16g(const uint8_t *p)
{
    uint4 v;
    __builtin_memcpy(&v, p, 16); // unaligned global_load_dwordx4
    return v;
}

// any batch of items (positions ascending with the lane): run_batch<false> + fingerprint upkeep.  Used for the first
// search of a block and for searches that outlast their head batch; the loads are per field, as in the second generation.
__device__ __forceinline__ BatchOut run_batch_fp(const uint8_t *in, uint16_t *tab, uint32_t tab_lds, uint32_t fp_lds, uint32_t pos, uint32_t v,
                                                 bool active, bool tested, uint32_t anchor, uint32_t matchlimit, uint32_t lane)
{
    BatchOut r;
    r.mpos = 0; r.match = 0; r.mc = 0; r.back = 0; r.stop = false; r.found = false; r.broken = false; r.flong = false; r.blong = false;
    const uint32_t h = hash13(v), fp = fp4(v);
    uint32_t old = 0, fo = 0;
    if (active) old = lz::tab_fp_exchange(tab_lds, fp_lds, h, pos, fp, &fo);
    if (__ballot(tested && old >= pos)) { r.broken = true; r.stop = true; return r; }
    {   // fingerprint lane order: the lane that inserted `old` in this batch (positions ascend with the lane)
        const uint32_t first = __builtin_amdgcn_readfirstlane(pos);
        uint32_t j = 0;
#pragma unroll
        for (uint32_t s = 32; s; s >>= 1) {
            const uint32_t pj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j + s) << 2), (int)pos);
            if (pj <= old) j += s;
        }
        const uint32_t pfp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)fp);
        // (old == 0 may also be an empty slot; position 0 is compared by its bytes, never by its fingerprint)
        if (__ballot(tested && old >= first && old != 0 && pfp != fo)) { r.broken = true; r.stop = true; return r; }
    }
    const bool near_start = pos < 4 || old < 4;
    Around ap, ac;
    ap.before = 0; ap.at = 0; ap.after = 0; ac.before = 0; ac.at = 0; ac.after = 1;
    if (tested) {
        ac = around<false>(in, old, !near_start);
        ap = around<false>(in, pos, !near_start);
    }
    const unsigned long long eqmask = __ballot(tested && ac.at == v);
    if (!eqmask) return r;
    const uint32_t w = (uint32_t)__builtin_ctzll(eqmask);
    const uint32_t pm = __builtin_amdgcn_readlane(pos | (old << 16), w);
    r.mpos = pm & 0xFFFFu; r.match = pm >> 16;
    if (active && lane > w && old <= r.mpos) { tab[h] = (uint16_t)old; lz::fp_store(fp_lds, h, fo); }
    const uint64_t x = ap.after ^ ac.after;
    uint32_t nf = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
    const uint32_t lim = matchlimit - (pos + 4);
    bool fl = false;
    if (nf >= lim) nf = lim; else fl = nf == 8;
    const uint32_t room = pos - anchor < old ? pos - anchor : old;
    const uint32_t y = ap.before ^ ac.before;
    uint32_t nb = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
    bool bl = false;
    if (near_start) { nb = 0; bl = room != 0; }
    else if (nb >= room) nb = room;
    else bl = nb == 4;
    const uint32_t packed = __builtin_amdgcn_readlane(nf | (nb << 8) | ((uint32_t)fl << 16) | ((uint32_t)bl << 17), w);
    r.mc = packed & 0xFFu; r.back = (packed >> 8) & 0xFFu;
    r.flong = (packed >> 16) & 1u; r.blong = (packed >> 17) & 1u;
    r.found = true; r.stop = true;
    return r;
}

// Head batch of a search after a match, on the register window W: lane t holds the 4 bytes at base + t, base = mend - 2.
// Lane 0 inserts mend-2, lane 1 idles, lane 2 re-tests mend, lane t >= 3 probes mend + t - 2: items in lane order.
// The window makes the batch independent of memory: the only load is the fetch of candidates whose fingerprint agrees.
#ifdef CW_STAMP
#define CW_HEAD_STAMP_PARAMS , uint32_t (&st_acc)[8], unsigned long long &st_last
#define CW_HEAD_STAMP_ARGS , st_acc, st_last
#else
#define CW_HEAD_STAMP_PARAMS
#define CW_HEAD_STAMP_ARGS
#endif
template <uint32_t HEADW>
__device__ __forceinline__ BatchOut head_batch_fp(const uint8_t *g, uint16_t *tab, uint32_t tab_lds, uint32_t fp_lds, uint32_t W, uint32_t base,
                                                  uint32_t mflimit, uint32_t matchlimit, uint32_t first_lo, uint32_t first_hi, uint32_t lane
                                                  CW_HEAD_STAMP_PARAMS)
{
    constexpr unsigned long long kHeadMask = (HEADW >= 64 ? ~0ull : (1ull << (HEADW & 63)) - 1) & ~2ull;
    BatchOut r;
    r.mpos = 0; r.match = 0; r.mc = 0; r.back = 0; r.stop = false; r.found = false; r.broken = false; r.flong = false; r.blong = false;
    const uint32_t pos = base + lane, anchor = base + 2;
    const unsigned long long lmask = __ballot(pos <= mflimit);
    const bool active = ((lmask & kHeadMask) >> lane) & 1u, tested = active && lane != 0;
    const uint32_t hp = W * 2654435761u, h = hp >> 19, fp = (hp >> 15) & 0xFu;
    uint32_t old = 0, fo = 0;
    CW_STAMP_AT(1); // window ready (bpermute or reload landed), hash computed
    if (active) old = lz::tab_fp_exchange(tab_lds, fp_lds, h, pos, fp, &fo);
    CW_STAMP_AT(2); // exchange
    // candidates: positions 0..3 from the block's first 8 bytes, the others only if the fingerprint agrees
    const bool need = tested && old >= 4 && fo == fp;
    uint4 cd = make_uint4(0, 0, 0, 0);
    if (need) cd = ld16g(g + old - 4);
    // lane order of the two exchanges (checked while the candidates are on their way): a candidate that an earlier lane of
    // this batch inserted sits at base + its lane, and must have come with that lane's fingerprint
    const uint32_t pfp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((old - base) << 2), (int)fp);
    if (__ballot(tested && (old >= pos || (old >= base && pfp != fo)))) { r.broken = true; r.stop = true; return r; }
    const uint32_t c_small = __builtin_amdgcn_alignbyte(first_hi, first_lo, old & 3u);
    const bool eq = tested && (old < 4 ? c_small == W : need && cd.y == W);
    const unsigned long long eqmask = __ballot(eq);
    CW_STAMP_AT(3); // candidate fetch + lane-order check
    if (!eqmask) {
        r.stop = (~lmask & kHeadMask) != 0; // the next probe would pass the end of the block
        return r;
    }
    const uint32_t w = (uint32_t)__builtin_ctzll(eqmask);
    r.mpos = base + w;
    r.match = (uint32_t)__builtin_amdgcn_readlane(old, w);
    if (active && lane > w && old <= r.mpos) { tab[h] = (uint16_t)old; lz::fp_store(fp_lds, h, fo); }
    // the winner's extension, scalar: forward bytes 4..11 are the window 4 and 8 lanes up, the bytes before it 4 lanes down
    const uint32_t room = w - 2 < r.match ? w - 2 : r.match; // mpos - anchor = w - 2
    if (r.match >= 4) {
        const uint64_t pa = (uint32_t)__builtin_amdgcn_readlane(W, w + 4) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(W, w + 8) << 32);
        const uint64_t ca = (uint32_t)__builtin_amdgcn_readlane(cd.z, w) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(cd.w, w) << 32);
        const uint64_t x = pa ^ ca;
        uint32_t nf = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
        const uint32_t lim = matchlimit - (r.mpos + 4);
        if (nf >= lim) nf = lim; else r.flong = nf == 8;
        // bytes [mpos-4, mpos): for w >= 6 a window lane; below that only the bytes from the anchor (lane 2) on can count,
        // and they are the low bytes of lane 2's value moved to the top (room < 4 cuts the rest off)
        const uint32_t pb = w >= 6 ? (uint32_t)__builtin_amdgcn_readlane(W, w - 4) : (uint32_t)__builtin_amdgcn_readlane(W, 2) << (8 * ((6 - w) & 3));
        const uint32_t y = pb ^ (uint32_t)__builtin_amdgcn_readlane(cd.x, w);
        uint32_t nb = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
        if (nb >= room) nb = room; else r.blong = nb == 4;
        r.mc = nf; r.back = nb;
        (void)anchor;
    } else { // candidate in the first 4 bytes: both directions in the byte loops
        r.flong = true; r.blong = room != 0;
    }
    r.found = true; r.stop = true;
    CW_STAMP_AT(4); // undo + scalar extension
    return r;
}

template <uint32_t HEADW>
__global__ void __launch_bounds__(64)
lz4_parse_fp_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                    uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes,
                    const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters, uint32_t *__restrict__ requeue,
                    uint32_t force_redo)
{
    constexpr unsigned long long kHeadMask = HEADW >= 64 ? ~0ull : (1ull << (HEADW & 63)) - 1;
    constexpr uint32_t kWinNeed = HEADW + 8; // lanes a head batch may read: its items and the 8 bytes after the last of them
    static_assert(kWinNeed <= 64, "the window is one register: a head batch and its look-ahead must fit 64 lanes");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    const uint32_t tab_lds = (uint32_t)reinterpret_cast<uintptr_t>(tab), fp_lds = tab_lds + kTabBytes;
    const uint32_t lane = threadIdx.x;

    const uint32_t qcount = counters[1];
    volatile uint32_t *mailbox = reinterpret_cast<volatile uint32_t *>(tab);
    for (uint32_t guard = 0; guard <= qcount; guard++) {
        __syncthreads();
        if (lane == 0) *mailbox = atomicAdd(&counters[0], 1u);
        __syncthreads();
        const uint32_t qi = __builtin_amdgcn_readfirstlane(*mailbox);
        if (qi >= qcount) break;
        const size_t blk = queue[qi];
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;

        __syncthreads();
        for (uint32_t i = lane; i < (kTabBytes + kFpBytes) / 16; i += 64) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();

        uint32_t anchor = 0, op = 0;
        bool broken = force_redo != 0;

        if (!broken && n >= kMFLimit + 1) {
            const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals;
            const uint32_t first_lo = __builtin_amdgcn_readfirstlane(rd32(g, 0)), first_hi = __builtin_amdgcn_readfirstlane(rd32(g, 4));
            // The window: lane t holds the 4 bytes at wbase + t for t < wvalid.  A match moves it up by mend - wbase - 2 lanes
            // (one ds_bpermute); when fewer than kWinNeed lanes would stay valid it is reloaded (a 64-lane gather of 68 bytes).
            CW_STAMP_DECL
            uint32_t W, Wlit, wbase = 0, wvalid = 0, s0 = 1;
            uint32_t gen_t0 = HEADW; // first item of a search that its head batch did not cover (the first search has one probe more)
            bool head_found; // the match came out of the head batch: its literals are window bytes
            uint32_t lit_lane;  // window lane of the literal run's first byte
            BatchOut bo;
            {   // first search of the block: lane 0 inserts position 0, lane 1 idles, lane t probes t-1
                const uint32_t pos = lane == 0 ? 0u : lane - 1;
                const unsigned long long lmask = __ballot(pos <= mflimit);
                const bool active = ((lmask & ~2ull & kHeadMask) >> lane) & 1u;
                W = rd32(g, pos <= mflimit ? pos : 0u);
                bo = run_batch_fp(g, tab, tab_lds, fp_lds, pos, W, active, active && lane != 0, anchor, matchlimit, lane);
                if (!bo.stop && (~lmask & kHeadMask)) bo.stop = true;
                head_found = bo.found; lit_lane = 1;
            }
            for (uint32_t seq = 0; seq < n; seq++) {
                if (!bo.stop) { // rare: more than HEADW - 2 probes without a match
                    head_found = false;
                    for (uint32_t t0 = gen_t0;; t0 += 64) {
                        const uint32_t k = t0 + lane - 2;
                        const uint32_t dk = probe_delta(k), stepk = (63u + k) >> 6;
                        const uint32_t p2 = s0 + dk;
                        const bool act2 = p2 + stepk <= mflimit + 1;
                        if (!__ballot(act2)) break;
                        const uint32_t v2 = rd32(g, act2 ? p2 : 0u);
                        bo = run_batch_fp(g, tab, tab_lds, fp_lds, p2, v2, act2, act2, anchor, matchlimit, lane);
                        if (bo.stop || __ballot(!act2)) break;
                    }
                }
                CW_STAMP_AT(5); // searches that outlast their head batch
                if (bo.broken) { broken = true; break; }
                if (!bo.found) break;
                uint32_t mpos = bo.mpos, match = bo.match, mc = bo.mc, back = bo.back;
#ifdef CW_STAMP
                st_seq++;
#endif

                if (bo.flong) { // long match: keep counting, 64 bytes per round
                    for (;;) {
                        const uint32_t i = mpos + kMinMatch + mc + lane;
                        const bool ok = i < matchlimit && g[i] == g[match + kMinMatch + mc + lane];
                        const uint32_t cnt = ctz64(~__ballot(ok));
                        mc += cnt;
                        if (cnt < 64) break;
                    }
                }
                if (bo.blong) { // long catch-up (rare)
                    for (;;) {
                        const uint32_t j = back + lane + 1;
                        const bool ok = mpos >= anchor + j && match >= j && g[mpos - j] == g[match - j];
                        const uint32_t cnt = ctz64(~__ballot(ok));
                        back += cnt;
                        if (cnt < 64) break;
                    }
                }
                const uint32_t mend = mpos + kMinMatch + mc; // first byte after the match
                const uint32_t ip = mpos - back;
                match -= back; mc += back;
                if (ip < anchor || mend > n) { broken = true; break; } // cannot happen; never write out of bounds

                CW_STAMP_AT(6); // byte loops of long matches / catch-ups
                // ---- the next head batch's window first (its exchange of lanes overlaps the stores below): move it to mend - 2,
                //      or reload it; the literals of this sequence are bytes of the window as it stands ----
                const bool more = mend <= mflimit;
                Wlit = W;
                if (more) {
                    const uint32_t nbase = mend - 2, shift = nbase - wbase;
                    if (shift + kWinNeed <= wvalid) {
                        W = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + shift) << 2), (int)W);
                        wvalid -= shift;
                    } else {
                        const uint32_t p = nbase + lane;
                        W = rd32(g, p <= n - 4 ? p : n - 4);
                        wvalid = 64;
                    }
                    wbase = nbase;
                }

                // ---- emit: literals [anchor, ip), offset, match length ----
                const uint32_t lit = ip - anchor, tok_pos = op;
                uint32_t token;
                op += 1;
                if (lit >= 15) { token = 15u << 4; op += put_len(out + op, lit - 15, lane); }
                else token = lit << 4;
                if (head_found) { // at most HEADW - 3 literals, all of them window bytes
                    if (lane - lit_lane < lit) out[op + lane - lit_lane] = (uint8_t)Wlit;
                } else {
                    copy_out(out + op, g, anchor, lit, lane);
                }
                op += lit;
                const uint32_t off = ip - match, off_pos = op;
                op += 2;
                if (mc >= 15) { token += 15; op += put_len(out + op, mc - 15, lane); }
                else token += mc;
                if (lane < 3) {
                    const uint32_t where = lane == 0 ? tok_pos : off_pos + lane - 1;
                    const uint32_t what = lane == 0 ? token : lane == 1 ? off : off >> 8;
                    out[where] = (uint8_t)what;
                }

                anchor = mend;
                CW_STAMP_AT(7); // window move issued, sequence emitted
                if (!more) break; // end of parse: remaining bytes are literals
                s0 = mend + 1; gen_t0 = HEADW - 1;
                bo = head_batch_fp<HEADW>(g, tab, tab_lds, fp_lds, W, wbase, mflimit, matchlimit, first_lo, first_hi, lane CW_HEAD_STAMP_ARGS);
                head_found = true; lit_lane = 2;
            }
            CW_STAMP_FLUSH();
        }
        if (broken) {
            if (lane == 0) requeue[atomicAdd(&counters[5], 1u)] = (uint32_t)blk;
            continue;
        }

        // ---- last literals ----
        {
            const uint32_t run = n - anchor;
            const uint32_t tok_pos = op;
            op += 1;
            if (run >= 15) {
                if (lane == 0) out[tok_pos] = 15u << 4;
                op += put_len(out + op, run - 15, lane);
            } else if (lane == 0) {
                out[tok_pos] = (uint8_t)(run << 4);
            }
            copy_out(out + op, g, anchor, run, lane);
            op += run;
        }
        if (lane == 0) sizes[blk] = op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Lane-per-block parser for large batches of queued blocks.
//
// The wavefront-per-block parsers above keep the 16 KiB table in LDS, so a CU holds 8-10 blocks, and one sequence of a
// block is a serial chain of about 2,500 cycles of instruction and LDS/memory latency (tools/parse_stamp.hip): 2,560
// chains on the chip are 14-17 GB/s on text whatever is trimmed, and LDS capacity admits no more chains.  Here a LANE
// owns a block and runs the serial parser as it stands (the oracle's loop, one probe per iteration), its table in global
// memory (16 KiB per lane, zeroed by the lane when it takes a block): 64 chains per wavefront, tens of thousands per
// chip, bound by how many random table / candidate accesses the memory system retires, not by any one chain's latency.
// It only pays when there are that many blocks: lz4_launch uses it from kLaneMidBlocks queued blocks on.
//
// Every lane is in one of the states below; an iteration of the wavefront's loop runs one step of every lane:
//   PROBE   the parser's search loop body, or the re-test right after a match (same table traffic, different follow-up)
//   EMIT    a match was found: catch-up, literals, offset, match length; then PROBE (as a re-test) or TAIL
//   TAIL    last literals, size; then NEXT
//   NEXT    pull the next queued block, zero the table
// Per iteration a lane's dependent memory chain is: its 16 bytes around ip -> table slot -> the candidate's 16 bytes.
// ---------------------------------------------------------------------------------------------------
// blocks > 4 KiB: below kLaneMidBlocks queued blocks the wavefront-per-block parser's 13-14 GB/s win; [mid, wide): lanes with two
// positions per iteration (every lane holds one block: latency regime), from kLaneWideBlocks on one (random-line regime); lz4_launch
// Round 3: below kLaneMidBlocks the two scalar-thread parsers (table in vector registers / in LDS, lz4_vtab_kernel.hip) win (corpus, 64 KiB: 12 Ki /
// 16 Ki / 20 Ki blocks 23.9 / 27.3 / 29.0 GB/s against the lanes' 15.4 / 19.5 / ~22), so the lanes start later than in round 2 (10,240), and from there on
// they run BESIDE those two: in the one-block-per-lane regime there are no lanes for kLaneLeave blocks of the queue and the lanes leave kLaneShare
// blocks (two thirds of a small call) alone; from kLaneWideBlocks on they leave kLaneShareWide.  One launch for both regimes (lz4_lanes_ring_auto_kernel).
constexpr uint32_t kLaneMidBlocks = 22528, kLaneWideBlocks = 98304; // (blocks <= 32 KiB: higher lower thresholds, lz4_launch)
constexpr bool kLtabDefault = true; // corpus, 64 KiB, alone on the queue: 8 Ki / 16 Ki / 48 Ki blocks 16.1 / 17.5 / 18.6 GB/s against the wavefront parser's 14.6 / 15.8 / 16.6; beside the register form 23.3 / 26.7 against 22.2 / 25.4
constexpr size_t kLaneLeave = 18432;   // blocks > 4 KiB, calls below kLaneWideBlocks: this many blocks get no lane (lz4_launch has the measurements)
constexpr uint32_t kLaneShare = 24576, kLaneShareWide = 32768;     // blocks of the queue the lanes leave to the other parsers (K = 2 / K = 1 regime); K = 2: at most two thirds of
                                                                   // the call -- since the lanes no longer take the whole queue at once (kLaneLeave) they pay from 22 Ki blocks on: corpus,
                                                                   // 64 KiB, 20 Ki / 24 Ki / 28 Ki blocks without lanes 29.0 / 29.4 / 29.8 GB/s, with 28.2 / 31.7 / 35.3 (16 Ki left);
                                                                   // 56 Ki / 72 Ki blocks with 16 Ki left 39.9-43.9 / 43.2-48.1, with 24 Ki 47.1 / 48.2-48.7
constexpr uint32_t kLaneMinSmall = 61440;  // LDS-staged blocks: lanes beside the LDS-resident parser from 60 Ki blocks on (64 Ki blocks of text: 28.5 against 25.7 GB/s)
enum : uint32_t { LS_NEXT = 0, LS_PROBE = 1, LS_EMIT = 2, LS_TAIL = 3, LS_EXIT = 4 };

__device__ __forceinline__ void lane_put_len(uint8_t *__restrict__ out, uint32_t &op, uint32_t extra)
{
    while (extra >= 255) { out[op++] = 255; extra -= 255; }
    out[op++] = (uint8_t)extra;
}

// 4 bytes at byte offset s (4 <= s <= 12) of a 16-byte window held in (x, y, z, w)
__device__ __forceinline__ uint32_t win_at(const uint4 &q, uint32_t s)
{
    return s < 8 ? __builtin_amdgcn_alignbyte(q.z, q.y, s & 3u) : s < 12 ? __builtin_amdgcn_alignbyte(q.w, q.z, s & 3u) : q.w;
}

// Table entries:
//   kLaneTagged (blocks <= 4 KiB)  u16 epoch:4 | position:12; an entry of another epoch reads as empty (= position 0, as in the
//                                  parser's zeroed table) and the table is zeroed once per 15 blocks instead of per block
//   kLaneFp (blocks > 4 KiB)       u32 fingerprint:16 | position:16, the fingerprint being 16 further bits of the hash product of
//                                  the 4 bytes the position was entered for: a candidate whose fingerprint differs holds other
//                                  bytes and is not fetched (on text two candidates of three are such: the kernel is bound by
//                                  the random 64-byte lines it draws from memory, and those are a third of them)
//   kLanePlain                     u16 position (profiling: CW_LZ4_LANES_FP=0)
enum : int { kLanePlain = 0, kLaneTagged = 1, kLaneFp = 2 };
__device__ __forceinline__ uint32_t fp16(uint32_t v) { return ((v * 2654435761u) >> 3) & 0xFFFFu; }

template <int MODE>
__global__ void __launch_bounds__(64)
lz4_lanes_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                 uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters,
                 uint16_t *__restrict__ tables, uint32_t min_blocks, uint32_t reserve)
{
    const uint32_t qcount = counters[1];
    if (qcount < min_blocks) return; // too few chains to fill the chip: the wavefront-per-block parser takes them all
    // fewer queued blocks than lanes: the first qcount / 64 workgroups take them with every lane busy (all of the grid starting
    // at once leaves each wavefront a random half of its lanes: mixed data, 64 Ki queued blocks on 128 Ki lanes, 79 vs 99 GB/s)
    if ((size_t)blockIdx.x * 64 >= qcount) return;
    // reserve > 0: the wavefront-per-block parser runs BESIDE this kernel on another stream, pulling from the same queue (it
    // is bound by LDS capacity and its own latency, this kernel by random memory accesses: the rates add).  A block takes a lane
    // ~100 ms and a wavefront ~10 ms, so the lanes stop pulling while `reserve` blocks are left: the wavefronts finish those in
    // about the time the lanes need for the blocks they hold.
    constexpr bool TAGGED = MODE == kLaneTagged, FP = MODE == kLaneFp;
    using Entry = typename std::conditional<FP, uint32_t, uint16_t>::type;
    Entry *tab = reinterpret_cast<Entry *>(tables) + ((size_t)blockIdx.x * 64 + threadIdx.x) * (1u << 13);
    const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals; // n >= 13: a queued block had a match
    uint32_t epoch = 15; // TAGGED: forces a clean table before the first block
    // the slot's position, and whether the bytes there can be the 4 bytes v at all
    auto tab_get = [&](uint32_t h, uint32_t v, bool &maybe) -> uint32_t {
        const uint32_t e = tab[h];
        maybe = true;
        if (FP) { maybe = (e >> 16) == fp16(v); return e & 0xFFFFu; }
        return TAGGED ? ((e >> 12) == epoch ? e & 0xFFFu : 0u) : e;
    };
    auto tab_put = [&](uint32_t h, uint32_t v, uint32_t pos) {
        tab[h] = (Entry)(FP ? (fp16(v) << 16) | pos : TAGGED ? (epoch << 12) | pos : pos);
    };

    uint32_t state = LS_NEXT;
    const uint8_t *g = src;
    uint8_t *out = dst;
    uint32_t blk = 0, ip = 0, anchor = 0, op = 0, step = 1, nb = 64, match = 0, first_lo = 0, first_hi = 0;
    bool retest = false;
    // own = the 16 bytes [ip-4, ip+12), requested one iteration ahead; vcur = the 4 bytes at ip, cut out of the previous
    // window when it reached that far (have_v), so that the table lookup never waits for the request
    uint4 own = make_uint4(0, 0, 0, 0), cd = make_uint4(0, 0, 0, 0);
    uint32_t vcur = 0, v2cur = 0;
    bool have_v = false;
    // literals of the last sequence on their way from memory: stored one iteration later (their load is then long done)
    uint64_t pend_a = 0, pend_b = 0;
    uint8_t *pend_dst = nullptr;
    uint32_t pend_n = 0;

    while (__ballot(state != LS_EXIT)) {
        // everything requested during the previous iteration is waited for here, once
        if (pend_n) { // exactly pend_n (1..16) bytes: what follows them in the slot is already written
            uint64_t lo = pend_a;
            uint8_t *p = pend_dst;
            if (pend_n & 16) { __builtin_memcpy(p, &lo, 8); __builtin_memcpy(p + 8, &pend_b, 8); }
            else {
                if (pend_n & 8) { __builtin_memcpy(p, &lo, 8); lo = pend_b; p += 8; }
                if (pend_n & 4) { const uint32_t t = (uint32_t)lo; __builtin_memcpy(p, &t, 4); lo >>= 32; p += 4; }
                if (pend_n & 2) { const uint16_t t = (uint16_t)lo; __builtin_memcpy(p, &t, 2); lo >>= 16; p += 2; }
                if (pend_n & 1) *p = (uint8_t)lo;
            }
            pend_n = 0;
        }
        if (state == LS_NEXT) {
            uint32_t qi = qcount;
            if (!reserve || __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + reserve < qcount)
                qi = atomicAdd(&counters[0], 1u);
            if (qi >= qcount) {
                state = LS_EXIT;
            } else {
                blk = queue[qi];
                g = src + (size_t)blk * src_stride;
                out = dst + (size_t)blk * dst_stride;
                if (!TAGGED || ++epoch == 16) {
                    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
                    for (uint32_t i = 0; i < (1u << 13) * sizeof(Entry) / 16; i++) t4[i] = make_uint4(0, 0, 0, 0);
                    epoch = 1;
                }
                first_lo = rd32(g, 0); first_hi = rd32(g, 4);
                // tab[hash(first 4 bytes)] = 0 is what an empty table already says; with fingerprints the entry says whose 0 it is
                if (FP) tab_put(hash13(first_lo), first_lo, 0);
                ip = 1; anchor = 0; op = 0; step = 1; nb = 64; retest = false;
                own.x = 0; own.y = rd32(g, 1); own.z = rd32(g, 5); own.w = rd32(g, 9); // no "before" at the block's start
                have_v = false;
                state = LS_PROBE;
            }
        }

        if (state == LS_PROBE) {
            const uint32_t next = ip + step;
            if (!retest && next > mflimit + 1) {
                state = LS_TAIL;
            } else {
                const uint32_t v = have_v ? vcur : own.y;
                if (retest) { // LZ4_putPosition(ip - 2) in front of the re-test
                    const uint32_t v2 = have_v ? v2cur : (own.x >> 16) | (own.y << 16);
                    tab_put(hash13(v2), v2, ip - 2);
                }
                const uint32_t h = hash13(v);
                bool maybe;
                match = tab_get(h, v, maybe);
                tab_put(h, v, ip);
                uint32_t cat = ~v;
                if (maybe) {
                    if (match >= 4) { cd = ld16g(g + match - 4); cat = cd.y; }
                    else cat = __builtin_amdgcn_alignbyte(first_hi, first_lo, match);
                }
                if (cat == v) {
                    state = LS_EMIT; // (own was requested for this ip an iteration ago: it is here by now)
                } else {
                    uint32_t nip;
                    if (retest) { nip = ip + 1; step = 1; nb = 64; retest = false; }
                    else { nip = next; step = nb >> 6; nb++; }
                    // the next position's 4 bytes, from the window if it reaches (it is the window of `ip` only if that
                    // has arrived, which it has unless this iteration ran on vcur: then the request is still the one for ip)
                    const uint32_t s = nip - ip + 4;
                    have_v = s <= 12 && ip >= 4;
                    if (have_v) vcur = win_at(own, s);
                    ip = nip;
                    // (ip = mflimit + 1 is never probed, the next iteration sends it to TAIL: keep its request inside the block)
                    const uint32_t rp = ip <= mflimit ? ip : mflimit;
                    own = ld16g(g + rp - (rp >= 4 ? 4 : 0));
                    if (rp < 4) { own.w = own.z; own.z = own.y; own.y = own.x; own.x = 0; have_v = false; }
                }
            }
        }

        if (state == LS_EMIT) {
            // own = [ip-4, ip+12) and cd = [match-4, match+12) (match >= 4), both as found by the probe
            const uint32_t ip0 = ip;
            const bool windows = ip >= 4 && match >= 4;
            uint32_t nf = 0; // equal bytes behind the 4 that matched
            bool nf_open = true;
            if (windows) {
                const uint64_t x = ((uint64_t)own.w << 32 | own.z) ^ ((uint64_t)cd.w << 32 | cd.z);
                nf = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
                nf_open = nf == 8;
                const uint32_t lim = matchlimit - (ip0 + kMinMatch);
                if (nf >= lim) { nf = lim; nf_open = false; }
            }
            // ---- catch-up over the pending literals (a re-test has none: anchor == ip) ----
            if (!retest) {
                if (windows) {
                    const uint32_t room = ip - anchor < match ? ip - anchor : match;
                    const uint32_t y = own.x ^ cd.x;
                    uint32_t back = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
                    if (back > room) back = room;
                    ip -= back; match -= back;
                    if (back == 4) while (ip > anchor && match > 0 && g[ip - 1] == g[match - 1]) { ip--; match--; }
                } else {
                    while (ip > anchor && match > 0 && g[ip - 1] == g[match - 1]) { ip--; match--; }
                }
            }
            // ---- literals: 8 or 16 bytes requested now and stored next iteration; longer runs copied here ----
            const uint32_t lit = ip - anchor, tok = op++;
            uint32_t token;
            if (lit >= 15) { token = 15u << 4; lane_put_len(out, op, lit - 15); }
            else token = lit << 4;
            if (lit) {
                // (8 bytes from anchor stay inside the block: anchor + 8 <= ip + 7 <= n - 5; 16 only for runs of 9 and more)
                __builtin_memcpy(&pend_a, g + anchor, 8);
                if (lit > 8) __builtin_memcpy(&pend_b, g + anchor + 8, 8);
                pend_dst = out + op;
                pend_n = lit < 16 ? lit : 16;
                for (uint32_t k = 16; k < lit; k += 8) { // runs beyond 16 (0.3 % on text): 8 bytes at a time, the overshoot (< 8
                    uint64_t q;                          // bytes) lands where the offset and what follows are written next
                    __builtin_memcpy(&q, g + anchor + k, 8);
                    __builtin_memcpy(out + op + k, &q, 8);
                }
            }
            op += lit;
            // ---- offset, match length ----
            const uint32_t off = ip - match;
            out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8);
            op += 2;
            // the bytes taken back, the 4 that matched and the nf behind them are one run: mc = (ip0 - ip) + nf (+ what memory adds)
            uint32_t mc = ip0 - ip + nf;
            if (nf_open) {
                const uint32_t a = ip + kMinMatch, b = match + kMinMatch;
                while (a + mc + 8 <= matchlimit) {
                    uint64_t x, y;
                    __builtin_memcpy(&x, g + a + mc, 8);
                    __builtin_memcpy(&y, g + b + mc, 8);
                    const uint64_t d = x ^ y;
                    if (d) { mc += (uint32_t)__builtin_ctzll(d) >> 3; break; }
                    mc += 8;
                }
                if (a + mc + 8 > matchlimit) while (a + mc < matchlimit && g[a + mc] == g[b + mc]) mc++;
            }
            if (mc >= 15) { token += 15; lane_put_len(out, op, mc - 15); }
            else token += mc;
            out[tok] = (uint8_t)token;
            ip += kMinMatch + mc;
            anchor = ip;
            if (ip > mflimit) {
                state = LS_TAIL;
            } else {
                // the re-test's values out of the old window when the match was short enough (mend + 4 <= ip0 + 12)
                const uint32_t s = ip - ip0 + 4;
                have_v = windows && s <= 12;
                if (have_v) { vcur = win_at(own, s); v2cur = win_at(own, s - 2); }
                own = ld16g(g + ip - 4); // ip >= 5
                retest = true;
                state = LS_PROBE;
            }
        }

        if (state == LS_TAIL) {
            const uint32_t run = n - anchor;
            if (run >= 15) { out[op++] = 15u << 4; lane_put_len(out, op, run - 15); }
            else out[op++] = (uint8_t)(run << 4);
            uint32_t k = 0;
            for (; k + 16 <= run; k += 16) {
                uint4 q;
                __builtin_memcpy(&q, g + anchor + k, 16);
                __builtin_memcpy(out + op + k, &q, 16);
            }
            for (; k < run; k++) out[op + k] = g[anchor + k];
            op += run;
            sizes[blk] = op;
            state = LS_NEXT;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Lane-per-block parser with the lane's input in an LDS ring (blocks > 4 KiB).
//
// tools/random_line.hip: the chip retires about 17.5 G "probes" per second of the lanes' pattern (a table entry loaded and
// stored at a random slot of a private table, a candidate's 16 bytes for every third), whatever the occupancy and whether or
// not the addresses depend on loaded data -- the lane parsers are bound by lines, not by latency.  lz4_lanes_kernel spends
// lines on more than probes: the 16 bytes around every position and the literals of every sequence are global loads of their
// own.  Here the block's bytes reach the lane once, in 64-byte pieces requested an iteration and more ahead of their use and
// kept in a 256-byte ring per lane in LDS (16 KiB per wavefront); positions, windows and literals are read from the ring, and
// memory sees the table entry, the candidates the fingerprint lets through, and the output.
//
// An iteration of the wavefront's loop:
//   A  every searching lane hashes its next K positions (the parser's own sequence of steps; a re-test after a match is the
//      batch's first entry) out of the ring and requests the K table entries; lanes extending a long match request the
//      next 32 bytes of both sides
//   B  the piece requested an iteration ago goes into the ring; entries arrive; a position whose slot an earlier position of
//      the batch (or the re-test's ip-2) also hashes to takes that position instead, as it would have read it; the candidates
//      whose fingerprint fits are requested
//   C  the first position whose candidate matches wins, the table takes the positions up to it (the parser never looked at
//      the rest), the sequence is emitted from the ring and the candidate's window; without a winner the search goes on
//      behind the batch.
// K = 1 ships: more positions per iteration cut the round trips per sequence, but the entries and candidates behind the
// winner are extra lines (K = 4: 26 GB/s against 36 GB/s without the fingerprints).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kRingBytes = 256, kRingPiece = 64;
enum : uint32_t { LB_NEXT = 0, LB_PROBE = 1, LB_EXTEND = 2, LB_TAIL = 3, LB_EXIT = 4 };

// stores exactly cnt (1..16) bytes of (a, b)
__device__ __forceinline__ void lane_store_upto16(uint8_t *p, uint64_t a, uint64_t b, uint32_t cnt)
{
    if (cnt & 16) { __builtin_memcpy(p, &a, 8); __builtin_memcpy(p + 8, &b, 8); return; }
    if (cnt & 8) { __builtin_memcpy(p, &a, 8); a = b; p += 8; }
    if (cnt & 4) { const uint32_t t = (uint32_t)a; __builtin_memcpy(p, &t, 4); a >>= 32; p += 4; }
    if (cnt & 2) { const uint16_t t = (uint16_t)a; __builtin_memcpy(p, &t, 2); a >>= 16; p += 2; }
    if (cnt & 1) *p = (uint8_t)a;
}

// (the body of the two kernels below; ring: the workgroup's (kRingBytes / 4) * 64 dwords of LDS, dword d of lane l's ring at [d * 64 + l]: a lane only
// touches its column; qcount: the queue's length)
template <int K>
__device__ __forceinline__ void
lz4_lanes_ring_body(uint32_t *__restrict__ ring, const uint32_t qcount, const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst,
                    size_t dst_stride, uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters,
                    uint16_t *__restrict__ tables, uint32_t reserve)
{
    const uint32_t lane = threadIdx.x;
    uint32_t *tab = reinterpret_cast<uint32_t *>(tables) + ((size_t)blockIdx.x * 64 + lane) * (1u << 13); // fingerprint:16 | position:16
    const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals; // n >= 64

    // bytes [x, x+4) / [x, x+16) of the block out of the ring (the caller knows they are there)
    auto ring_dw = [&](uint32_t x) -> uint32_t { return ring[((x >> 2) & (kRingBytes / 4 - 1)) * 64 + lane]; };
    auto ring32 = [&](uint32_t x) -> uint32_t { return __builtin_amdgcn_alignbyte(ring_dw(x + 4), ring_dw(x), x & 3u); };
    auto ring16 = [&](uint32_t x) -> uint4 {
        const uint32_t d0 = ring_dw(x), d1 = ring_dw(x + 4), d2 = ring_dw(x + 8), d3 = ring_dw(x + 12), d4 = ring_dw(x + 16), r = x & 3u;
        return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, r), __builtin_amdgcn_alignbyte(d2, d1, r), __builtin_amdgcn_alignbyte(d3, d2, r),
                          __builtin_amdgcn_alignbyte(d4, d3, r));
    };

    uint32_t state = LB_NEXT;
    const uint8_t *g = src;
    uint8_t *out = dst;
    uint32_t blk = 0, ip = 0, anchor = 0, op = 0, step = 1, nb = 64, match = 0, first_lo = 0, first_hi = 0;
    bool retest = false;
    uint32_t rb = 0, re = 0;            // the ring holds the block's bytes [rb, re)
    uint4 fl0, fl1, fl2, fl3;           // the piece [re, re + 64) on its way
    fl0 = fl1 = fl2 = fl3 = make_uint4(0, 0, 0, 0);
    bool inflight = false;
    uint32_t mc = 0, tok = 0, token = 0; // a sequence whose match is still being extended

    while (__ballot(state != LB_EXIT)) {
        if (state == LB_NEXT) {
            uint32_t qi = qcount;
            if (!reserve || __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + reserve < qcount)
                qi = atomicAdd(&counters[0], 1u);
            if (qi >= qcount) {
                state = LB_EXIT;
            } else {
                blk = queue[qi];
                g = src + (size_t)blk * src_stride;
                out = dst + (size_t)blk * dst_stride;
                uint4 *t4 = reinterpret_cast<uint4 *>(tab);
                for (uint32_t i = 0; i < (1u << 13) * 4 / 16; i++) t4[i] = make_uint4(0, 0, 0, 0);
                first_lo = rd32(g, 0); first_hi = rd32(g, 4);
                tab[hash13(first_lo)] = fp16(first_lo) << 16; // position 0, by name (an empty entry has fingerprint 0)
                ip = 1; anchor = 0; op = 0; step = 1; nb = 64; retest = false;
                rb = 0; re = 0; inflight = false;
                state = LB_PROBE;
            }
        }
        // a lane that searches needs a piece on its way whenever its ring ends less than 104 bytes ahead (the most a batch
        // advances is 16 + 12, the piece requested now is usable two iterations on); a match that jumped over the ring's end
        // restarts the ring at the new position
        if (state == LB_PROBE && !inflight) {
            if (ip >= 4 + kRingPiece && ip - 4 - kRingPiece >= re) rb = re = (ip - 4) & ~(kRingPiece - 1);
            if (re < n && re < ip + 104) {
                const uint8_t *q = g + re;
                if (re + kRingPiece <= n) {
                    fl0 = ld16g(q); fl1 = ld16g(q + 16); fl2 = ld16g(q + 32); fl3 = ld16g(q + 48);
                } else { // the block's last piece: byte by byte, zeros behind the block
                    uint32_t w[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        uint32_t v = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const uint32_t x = re + 4 * i + b;
                            if (x < n) v |= (uint32_t)g[x] << (8 * b);
                        }
                        w[i] = v;
                    }
                    fl0 = make_uint4(w[0], w[1], w[2], w[3]); fl1 = make_uint4(w[4], w[5], w[6], w[7]);
                    fl2 = make_uint4(w[8], w[9], w[10], w[11]); fl3 = make_uint4(w[12], w[13], w[14], w[15]);
                }
                inflight = true;
            }
        }

        // ---------------- A: positions, hashes, table requests ----------------
        uint32_t pos[K], hsh[K], val[K], raw[K];
        uint32_t cnt = 0, h2 = 0xFFFFFFFFu, v2 = 0, after_ip = 0, after_step = 0, after_nb = 0;
        bool tail_after = false;
        const bool searching = state == LB_PROBE && re >= (ip + 24 < n ? ip + 24 : n); // (rb <= ip - 4 by construction)
        if (searching) {
            uint32_t p = ip, st = step, c = nb;
            if (retest) { // LZ4_putPosition(ip - 2) in front of the re-test
                v2 = ring32(ip - 2);
                h2 = hash13(v2);
                tab[h2] = fp16(v2) << 16 | (ip - 2);
            }
            bool building = true;
#pragma unroll
            for (int j = 0; j < K; j++) {
                pos[j] = hsh[j] = val[j] = raw[j] = 0;
                if (building) {
                    const bool is_rt = retest && j == 0;
                    if (!is_rt && p + st > mflimit + 1) { tail_after = true; building = false; }
                    else if (p - ip > 12) building = false;
                    else {
                        pos[j] = p; val[j] = ring32(p); hsh[j] = hash13(val[j]); cnt = j + 1;
                        raw[j] = tab[hsh[j]];
                        if (is_rt) { p = p + 1; st = 1; c = 64; }
                        else { p = p + st; st = c >> 6; c++; }
                    }
                }
            }
            after_ip = p; after_step = st; after_nb = c;
        }
        uint4 xa0, xb0, xa1, xb1;
        xa0 = xb0 = xa1 = xb1 = make_uint4(0, 0, 0, 0);
        uint32_t ext_vec = 0;
        if (state == LB_EXTEND) {
            const uint32_t a = ip + kMinMatch + mc, b = match + kMinMatch + mc;
            if (a + 32 <= matchlimit) { xa0 = ld16g(g + a); xb0 = ld16g(g + b); xa1 = ld16g(g + a + 16); xb1 = ld16g(g + b + 16); ext_vec = 2; }
            else if (a + 16 <= matchlimit) { xa0 = ld16g(g + a); xb0 = ld16g(g + b); ext_vec = 1; }
        }

        // ---------------- B: the piece requested an iteration ago goes into the ring; candidates ----------------
        if (inflight) {
            const uint32_t d = (re >> 2) & (kRingBytes / 4 - 1);
            uint32_t *r = ring + d * 64 + lane;
            r[0 * 64] = fl0.x; r[1 * 64] = fl0.y; r[2 * 64] = fl0.z; r[3 * 64] = fl0.w;
            r[4 * 64] = fl1.x; r[5 * 64] = fl1.y; r[6 * 64] = fl1.z; r[7 * 64] = fl1.w;
            r[8 * 64] = fl2.x; r[9 * 64] = fl2.y; r[10 * 64] = fl2.z; r[11 * 64] = fl2.w;
            r[12 * 64] = fl3.x; r[13 * 64] = fl3.y; r[14 * 64] = fl3.z; r[15 * 64] = fl3.w;
            re += kRingPiece;
            if (re > rb + kRingBytes) rb = re - kRingBytes;
            inflight = false;
        }
        uint32_t mt[K];
        bool maybe[K];
        uint4 cdw[K];
        if (searching) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                uint32_t m = raw[j] & 0xFFFFu;
                bool may = (raw[j] >> 16) == fp16(val[j]);
                if (hsh[j] == h2) { m = ip - 2; may = v2 == val[j]; }
#pragma unroll
                for (int i = 0; i < j; i++)
                    if ((uint32_t)i < cnt && hsh[i] == hsh[j]) { m = pos[i]; may = val[i] == val[j]; }
                mt[j] = m;
                maybe[j] = may && (uint32_t)j < cnt;
                cdw[j] = make_uint4(0, 0, 0, 0);
                if (maybe[j] && m >= 4) cdw[j] = ld16g(g + m - 4);
            }
        }

        // ---------------- C: winner, table, sequence ----------------
        bool emit = false;
        uint4 cd = make_uint4(0, 0, 0, 0);
        if (searching) {
            uint32_t win = K;
#pragma unroll
            for (int j = K - 1; j >= 0; j--) {
                const uint32_t cat = mt[j] >= 4 ? cdw[j].y : __builtin_amdgcn_alignbyte(first_hi, first_lo, mt[j]);
                if (maybe[j] && cat == val[j]) win = (uint32_t)j;
            }
#pragma unroll
            for (int j = 0; j < K; j++)
                if ((uint32_t)j < cnt && (uint32_t)j <= win) tab[hsh[j]] = fp16(val[j]) << 16 | pos[j];
            if (win < K) {
#pragma unroll
                for (int j = 0; j < K; j++)
                    if (win == (uint32_t)j) { ip = pos[j]; match = mt[j]; cd = cdw[j]; }
                emit = true;
            } else if (tail_after) {
                state = LB_TAIL;
            } else {
                ip = after_ip; step = after_step; nb = after_nb; retest = false;
            }
        }
        bool finish = false; // the sequence's match length is known: token, length bytes, advance
        if (emit) {
            const uint32_t ip0 = ip;
            const bool windows = ip >= 4 && match >= 4;
            uint32_t nf = 0; // equal bytes behind the 4 that matched
            bool nf_open = true;
            if (windows) {
                const uint4 own = ring16(ip - 4);
                const uint64_t x = ((uint64_t)own.w << 32 | own.z) ^ ((uint64_t)cd.w << 32 | cd.z);
                nf = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
                nf_open = nf == 8;
                const uint32_t lim = matchlimit - (ip0 + kMinMatch);
                if (nf >= lim) { nf = lim; nf_open = false; }
                // catch-up over the pending literals (none right after a match: anchor == ip)
                const uint32_t room = ip - anchor < match ? ip - anchor : match;
                const uint32_t y = own.x ^ cd.x;
                uint32_t back = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
                if (back > room) back = room;
                ip -= back; match -= back;
                if (back == 4) while (ip > anchor && match > 0 && g[ip - 1] == g[match - 1]) { ip--; match--; }
            } else {
                while (ip > anchor && match > 0 && g[ip - 1] == g[match - 1]) { ip--; match--; }
            }
            const uint32_t lit = ip - anchor;
            tok = op++;
            if (lit >= 15) { token = 15u << 4; lane_put_len(out, op, lit - 15); }
            else token = lit << 4;
            if (lit) {
                if (anchor >= rb) { // out of the ring (ip <= re: the run's bytes are all there)
                    for (uint32_t k = 0; k < lit; k += 16) {
                        const uint4 q = ring16(anchor + k);
                        lane_store_upto16(out + op + k, (uint64_t)q.y << 32 | q.x, (uint64_t)q.w << 32 | q.z, lit - k < 16 ? lit - k : 16);
                    }
                } else { // a run longer than the ring remembers
                    uint32_t k = 0;
                    for (; k + 8 <= lit; k += 8) {
                        uint64_t q;
                        __builtin_memcpy(&q, g + anchor + k, 8);
                        __builtin_memcpy(out + op + k, &q, 8);
                    }
                    for (; k < lit; k++) out[op + k] = g[anchor + k];
                }
            }
            op += lit;
            const uint32_t off = ip - match;
            out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8);
            op += 2;
            mc = ip0 - ip + nf; // the bytes taken back, the 4 that matched and the nf behind them are one run
            if (nf_open) state = LB_EXTEND;
            else finish = true;
        } else if (state == LB_EXTEND) {
            const uint32_t a = ip + kMinMatch, b = match + kMinMatch;
            bool open = true;
            if (ext_vec) {
                const uint64_t d0 = ((uint64_t)xa0.y << 32 | xa0.x) ^ ((uint64_t)xb0.y << 32 | xb0.x);
                const uint64_t d1 = ((uint64_t)xa0.w << 32 | xa0.z) ^ ((uint64_t)xb0.w << 32 | xb0.z);
                if (d0) { mc += (uint32_t)__builtin_ctzll(d0) >> 3; open = false; }
                else if (d1) { mc += 8 + ((uint32_t)__builtin_ctzll(d1) >> 3); open = false; }
                else {
                    mc += 16;
                    if (ext_vec == 2) {
                        const uint64_t d2 = ((uint64_t)xa1.y << 32 | xa1.x) ^ ((uint64_t)xb1.y << 32 | xb1.x);
                        const uint64_t d3 = ((uint64_t)xa1.w << 32 | xa1.z) ^ ((uint64_t)xb1.w << 32 | xb1.z);
                        if (d2) { mc += (uint32_t)__builtin_ctzll(d2) >> 3; open = false; }
                        else if (d3) { mc += 8 + ((uint32_t)__builtin_ctzll(d3) >> 3); open = false; }
                        else mc += 16;
                    }
                }
            } else { // fewer than 16 bytes to the match limit
                while (a + mc < matchlimit && g[a + mc] == g[b + mc]) mc++;
                open = false;
            }
            if (!open) finish = true;
        }
        if (finish) {
            if (mc >= 15) { token += 15; lane_put_len(out, op, mc - 15); }
            else token += mc;
            out[tok] = (uint8_t)token;
            ip += kMinMatch + mc;
            anchor = ip;
            if (ip > mflimit) state = LB_TAIL;
            else { retest = true; state = LB_PROBE; }
        }

        if (state == LB_TAIL) {
            const uint32_t run = n - anchor;
            if (run >= 15) { out[op++] = 15u << 4; lane_put_len(out, op, run - 15); }
            else out[op++] = (uint8_t)(run << 4);
            uint32_t k = 0;
            for (; k + 16 <= run; k += 16) {
                uint4 q;
                __builtin_memcpy(&q, g + anchor + k, 16);
                __builtin_memcpy(out + op + k, &q, 16);
            }
            for (; k < run; k++) out[op + k] = g[anchor + k];
            op += run;
            sizes[blk] = op;
            state = LB_NEXT;
        }
    }
}

// K positions of a lane's search per iteration, whatever the queue's length within [min_blocks, max_blocks)
template <int K>
__global__ void __launch_bounds__(64)
lz4_lanes_ring_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                      uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters,
                      uint16_t *__restrict__ tables, uint32_t min_blocks, uint32_t reserve, uint32_t max_blocks)
{
    __shared__ uint32_t ring[(kRingBytes / 4) * 64];
    const uint32_t qcount = counters[1];
    if (qcount < min_blocks || qcount >= max_blocks) return;
    if ((size_t)blockIdx.x * 64 >= qcount) return; // (as in lz4_lanes_kernel)
    lz4_lanes_ring_body<K>(ring, qcount, src, n, src_stride, dst, dst_stride, sizes, queue, counters, tables, reserve);
}

// The launch policy's form: ONE launch, the number of positions per iteration chosen on the device by the queue's length (two below wide_from,
// one from there on; see lz4_launch).  Until round 3 these were two launches on one stream, each returning at once outside its range -- and from
// kLaneWideBlocks queued blocks on the second one lost the race for the CUs: while the first launch's workgroups came and went, the register-table
// and LDS-table parsers of the other streams had filled every CU (the register-table parser alone takes a CU's whole vector register file), and the
// lanes only started when those ran out of queue: corpus, 64 KiB, 128 Ki / 256 Ki blocks 30.3 / 31.5 GB/s instead of 47.2 / 49.0.
__global__ void __launch_bounds__(64)
lz4_lanes_ring_auto_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                           uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters,
                           uint16_t *__restrict__ tables, uint32_t min_blocks, uint32_t reserve_mid, uint32_t wide_from, uint32_t reserve_wide,
                           uint32_t leave_mid)
{
    __shared__ uint32_t ring[(kRingBytes / 4) * 64];
    const uint32_t qcount = counters[1];
    if (qcount < min_blocks) return; // (the queue's length decides on the device whether the lanes parse at all)
    if ((size_t)blockIdx.x * 64 >= qcount) return;
    if (qcount >= wide_from) {
        lz4_lanes_ring_body<1>(ring, qcount, src, n, src_stride, dst, dst_stride, sizes, queue, counters, tables, reserve_wide);
    } else {
        // below wide_from every lane gets one block: no lanes for the leave_mid blocks the on-chip parsers get through meanwhile (lz4_launch)
        if ((size_t)blockIdx.x * 64 + leave_mid >= qcount && blockIdx.x > 0) return;
        lz4_lanes_ring_body<2>(ring, qcount, src, n, src_stride, dst, dst_stride, sizes, queue, counters, tables, reserve_mid);
    }
}

// per-stream workspace: counters[8] (parse queue head, tail; scan feed; -; second queue head, tail) + two queues
namespace {
struct Workspace {
    uint32_t *p = nullptr; size_t cap = 0;
    uint16_t *lane_tabs = nullptr; size_t lane_cap = 0; // tables of the lane-per-block parser: 16 KiB per lane
    hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; // the lane parser's stream beside the caller's
    hipStream_t side2 = nullptr; hipEvent_t fork2 = nullptr, join2 = nullptr; // the register-table parser's
    std::mutex launch;
};
std::mutex ws_lock;
std::unordered_map<uint64_t, Workspace> ws_map; // references stay valid across inserts

Workspace &find_workspace(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(ws_lock);
    return ws_map[ws_key(stream)];
}

// caller holds w.launch
hipError_t grow_workspace(Workspace &w, size_t nblocks, uint32_t **out, size_t *cap_out)
{
    if (w.cap < nblocks) { // only ever on the first (or a larger) call on this stream
        if (w.p) { hipError_t e = hipFree(w.p); if (e != hipSuccess) return e; }
        w.p = nullptr; w.cap = 0;
        size_t cap = nblocks < 4096 ? 4096 : nblocks;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&w.p), (2 * cap + 8) * sizeof(uint32_t));
        if (e != hipSuccess) return e;
        w.cap = cap;
    }
    *out = w.p;
    *cap_out = w.cap;
    return hipSuccess;
}
} // namespace

// device address of the word that holds the number of blocks the last call's scan queued for the parsers on this stream (nullptr: no call yet)
const uint32_t *lz4_queued_blocks_word(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(ws_lock);
    auto it = ws_map.find(ws_key(stream));
    return it == ws_map.end() || !it->second.p ? nullptr : it->second.p + 1;
}

void lz4_release_workspaces()
{
    std::lock_guard<std::mutex> g(ws_lock);
    for (auto &kv : ws_map) {
        if (kv.second.p) (void)hipFree(kv.second.p);
        if (kv.second.lane_tabs) (void)hipFree(kv.second.lane_tabs);
        if (kv.second.side) { (void)hipStreamDestroy(kv.second.side); (void)hipEventDestroy(kv.second.fork); (void)hipEventDestroy(kv.second.join); }
        if (kv.second.side2) { (void)hipStreamDestroy(kv.second.side2); (void)hipEventDestroy(kv.second.fork2); (void)hipEventDestroy(kv.second.join2); }
    }
    ws_map.clear();
}

void lz4_release_stream(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(ws_lock);
    auto it = ws_map.find(ws_key(stream));
    if (it == ws_map.end()) return;
    Workspace &w = it->second;
    if (w.p) (void)hipFree(w.p);
    if (w.lane_tabs) (void)hipFree(w.lane_tabs);
    if (w.side) { (void)hipStreamDestroy(w.side); (void)hipEventDestroy(w.fork); (void)hipEventDestroy(w.join); }
    if (w.side2) { (void)hipStreamDestroy(w.side2); (void)hipEventDestroy(w.fork2); (void)hipEventDestroy(w.join2); }
    ws_map.erase(it);
}

hipError_t lz4_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, uint8_t *dst,
                      size_t dst_stride, uint32_t *sizes, hipStream_t stream, const AfterScan *after_scan)
{
    if (nblocks == 0) return hipSuccess;
    if (block_bytes == 0 || block_bytes > 65536 || nblocks > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t n = (uint32_t)block_bytes;
    const char *sm_env = tune("CW_LZ4_STAGE_MAX"); // profiling knob: largest block parsed from an LDS copy
    // measured on text: 4 KiB 26.0 (staged) vs 22.4 GB/s (global); 8 KiB 18.1 vs 20.6; 16 KiB 11.5 vs 18.7 -- blocks per CU win
    const uint32_t stage_max = sm_env && atoi(sm_env) >= 0 ? (uint32_t)atoi(sm_env) : 4096u;
    const bool staged = n <= (stage_max < kStageMax ? stage_max : kStageMax);
    // staged bytes are read as aligned dwords: a size that is not a multiple of 4 gets 16 bytes of slack behind it
    uint32_t lds = kTabBytes + (staged ? ((n + 15u) & ~15u) + (n % 4 ? 16u : 0u) : 0u);
    static bool attr_set = false; // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lz4_blocks_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kStageMax + kTabBytes + 16);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(lz4_parse_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kStageMax + kTabBytes + 16);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    uint32_t *ws = nullptr;
    size_t cap = 0;
    Workspace &wsp = find_workspace(stream);
    std::lock_guard<std::mutex> sequence(wsp.launch); // counters/queues are shared by every launch below
    hipError_t e = grow_workspace(wsp, nblocks, &ws, &cap);
    if (e != hipSuccess) return e;
    uint32_t *counters = ws, *queue = ws + 8, *queue2 = ws + 8 + cap;
    // what this call launches, noted in the branch that launches it (cw_profile_kernels): names as rocprofv3 prints them; a kernel that
    // decides on the device whether the queue's length is in its range carries the range
    char launched[320] = "";
    auto note = [&](const char *fmt, auto... a) __attribute__((format(printf, 2, 0))) {
        const size_t used = strlen(launched);
        if (used && used + 3 < sizeof launched) strcat(launched, " + ");
        const size_t at = strlen(launched);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wformat-security"
        snprintf(launched + at, sizeof launched - at, fmt, a...);
#pragma clang diagnostic pop
    };

    if ((e = hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), stream)) != hipSuccess) return e;
    // CW_LZ4_MODE=scan stops after the scan kernel (queued blocks keep sizes[i] = 0xFFFFFFFF): a profiling knob
    const char *mode = tune("CW_LZ4_MODE");
    // scan: one wavefront per workgroup, 32 KiB of LDS each -> 5 per CU; the grid-stride loop walks the rest
    const size_t scan_grid = nblocks < 256 * 5 ? nblocks : 256 * 5;
    // CW_LZ4_MODE=generic forces the gather-based scan (profiling knob)
    const bool streamable = ((reinterpret_cast<uintptr_t>(src) | src_stride | n) & 15) == 0 && !(mode && strcmp(mode, "generic") == 0);
    if (streamable) {
        const char *wpc_env = tune("CW_SCAN_WPC"); // scan wavefronts per CU (profiling knob; 4 = all that fit)
        const size_t wpc = wpc_env && atoi(wpc_env) > 0 ? (size_t)atoi(wpc_env) : 4;
        // power-of-two sizes 4 KiB .. 64 KiB go through the span kernel, 64 KiB of whole blocks per pull; what does not
        // fill a span (and every other size) through the per-block streaming kernel.  CW_LZ4_MODE=stream: the latter only.
        const bool pow2 = n >= kChunk && (n & (n - 1)) == 0 && !(mode && strcmp(mode, "stream") == 0);
        uint32_t lg = 0;
        while (pow2 && (kChunk << lg) < n) lg++;
        const size_t run = pow2 ? (size_t)(16u >> lg) : 1;
        const size_t nspans = pow2 ? nblocks / run : 0, done = nspans * run;
        if (nspans) {
            const size_t g = nspans < 256 * wpc ? nspans : 256 * wpc; // 40 KiB of LDS each -> at most 4 per CU
            hipLaunchKernelGGL(lz4_scan_span_kernel, dim3((unsigned)g), dim3(64), 0, stream, src, n, src_stride, (uint32_t)nspans, dst,
                               dst_stride, sizes, scan_probes(n), queue, counters, lg);
            note("cw::lz4_scan_span_kernel");
        }
        if (done < nblocks) {
            const size_t rest = nblocks - done;
            const size_t sgrid = rest < 256 * wpc ? rest : 256 * wpc;
            hipLaunchKernelGGL(lz4_scan_stream_kernel, dim3((unsigned)sgrid), dim3(64), 0, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, scan_probes(n), queue, counters, done, 3u);
            note("cw::lz4_scan_stream_kernel");
        }
    } else {
        hipLaunchKernelGGL(lz4_scan_kernel, dim3((unsigned)scan_grid), dim3(64), 0, stream, src, n, src_stride, nblocks, dst,
                           dst_stride, sizes, scan_probes(n), queue, counters);
        note("cw::lz4_scan_kernel");
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // the fused call's hook (cw_api.hip, dev_fused): what it enqueues here runs beside the scan, and the parsers below wait for it
    if (after_scan && (e = after_scan->fn(after_scan->ctx)) != hipSuccess) return e;
    if (mode && strcmp(mode, "scan") == 0) { note_kernels(0, launched); return hipSuccess; }
    // parse: queued blocks only; LDS admits 160 KiB / lds workgroups per CU
    // CW_LZ4_PARSE=fp: blocks read from global memory go through the fingerprint parser (20 KiB of LDS, 8 blocks per CU).
    // Measured on text at 64 KiB: 11.9 GB/s against 14.2 GB/s for the second generation with its 10 blocks per CU -- both
    // are bound by the instruction latency of one sequence's serial chain (tools/parse_stamp.hip), not by candidate
    // traffic, so the extra blocks win; the second generation stays the default.
    const char *gen_env = tune("CW_LZ4_PARSE");
    const bool use_fp = !staged && gen_env && strcmp(gen_env, "fp") == 0;
    const char *hw_env = tune("CW_LZ4_HEADW"); // head batch width of the fingerprint parser (profiling knob: 8, 16, 32)
    const int headw = hw_env ? atoi(hw_env) : 16;
    if (use_fp) lds = kTabBytes + kFpBytes;
    const size_t per_cu = (160u * 1024u) / lds ? (160u * 1024u) / lds : 1;
    // Large batches: the lane-per-block parser.  Two regimes:
    //  * blocks read from global memory (> 4 KiB), from kLaneMidBlocks queued blocks on: the lanes take the whole queue, the
    //    wavefront-per-block parser only what they leave (running it beside the lanes gains nothing there: both end up waiting
    //    for the same memory system -- 33.1 vs 34.2 GB/s);
    //  * LDS-staged blocks (<= 4 KiB), from kLaneMinSmall blocks on: the lanes run BESIDE the LDS-resident parser on a second
    //    stream, both pulling from the scan's queue -- one is bound by LDS capacity and its chain latency, the other by random
    //    memory accesses, and the rates add (4 KiB text: 26.2 -> 40.1 GB/s).
    // The kernel looks at the queue length on the device and leaves everything to the wavefront parser below the threshold.
    // CW_LZ4_LANES=0 switches it off, =N sets the threshold (1: every queued block, in the tests); CW_LANES_WPC = its
    // wavefronts per CU, CW_LANES_CONCURRENT=0|1 forces the regime, CW_LANES_RESERVE the blocks left to the wavefronts.
    const char *lanes_env = tune("CW_LZ4_LANES");
    // measured break-even with the wavefront parser on text (GB/s, wavefront parser / lanes): 64 KiB 16 Ki blocks 14.2 / 20.9; 16 KiB 16 Ki
    // blocks 18.3 / 17.4, 24 Ki 18.3 / 20.7; 8 KiB 24 Ki blocks 20.9 / 18.5, 32 Ki 20.4 / 22.7 (on small blocks the wavefront parser
    // is faster and a lane slower per byte: every block starts on an empty table, and has one to zero)
    const uint32_t lane_min = lanes_env ? (uint32_t)atoi(lanes_env)
                              : staged ? kLaneMinSmall : n > 32768 ? kLaneMidBlocks : n > 16384 ? 40960u : n > 8192 ? 61440u : 98304u;
    // (16 KiB blocks: the register-table + wavefront parsers 25.7 / 27.9 / 29.3 GB/s at 16 Ki / 32 Ki / 64 Ki blocks against the lanes' 19.5 / 25.0 / 30.1;
    //  8 KiB blocks: 25.5 / 28.8 / 30.0 at 16 Ki / 48 Ki / 96 Ki blocks against 21.4 / 21.3 / 30.3)
    bool lanes_used = false, lanes_beside = false;
    const char *lf_env = tune("CW_LZ4_LANES_FP"); // profiling knob: 0 = 16-bit table entries without fingerprints for blocks > 4 KiB
    const bool lanes_fp = !(lf_env && lf_env[0] == '0');
    // CW_LZ4_LANES_RING: 0 = input from global memory (lz4_lanes_kernel); 1, 2, 4, 8 = the ring form with that many positions per
    // iteration whatever the queue's length.  Unset: the ring form, K chosen ON THE DEVICE by the queue's length -- two launches,
    // each of which returns at once unless the length lies in its range:
    //   [kLaneMidBlocks, kLaneWideBlocks)  K = 2.  Every lane holds one block and the call lasts as long as one lane needs for
    //       one block: latency, not lines, so the second position's table entry and candidate requested together with the
    //       first's pay (text, 64 KiB, 16 Ki / 24 Ki / 32 Ki blocks: 21.1 / 26.0 / 29.9 GB/s against 16.4 / 21.0 / 27.0 with K = 1
    //       and 14.2 for the wavefront parser, which keeps everything below ~10 Ki blocks: 8 Ki blocks 13.5 against 11.8);
    //   [kLaneWideBlocks, ...)             K = 1.  Enough chains to be bound by the memory system's random lines, where the
    //       lines of the speculative second position only cost (64 Ki blocks: 38.3 against 35.6 GB/s).
    //   K = 4 / 8 are never better (16 Ki blocks: 20.0 / 16.4 GB/s): each position adds instructions to every iteration.
    const char *lr_env = tune("CW_LZ4_LANES_RING");
    const int lanes_ring = lr_env ? atoi(lr_env) : -1; // -1: by queue length
    if (!use_fp && lane_min && nblocks >= lane_min && n >= 64) {
        const char *lw_env = tune("CW_LANES_WPC");
        const size_t lwpc = lw_env && atoi(lw_env) > 0 ? (size_t)atoi(lw_env) : 8;
        size_t lgrid = (nblocks + 63) / 64, lcap = 256 * lwpc;
        uint32_t lane_leave = 0;
        // LDS-staged blocks, lanes beside the wavefront parser: lanes for about half of the blocks (2 .. 8 wavefronts per CU).  With
        // fewer lanes each is faster (less traffic per probe in flight), and a batch of 64 Ki .. 256 Ki blocks is over before a lane
        // has parsed more than two or three (text, 4 KiB, 80 Ki / 128 Ki / 256 Ki blocks: 2 wavefronts per CU 34.4 / 33.6 / 35.6 GB/s,
        // 4: 27.7 / 39.6 / 36.5, 8: 24.9 / 26.0 / 39.0-41.0; the wavefront parser alone 25.8)
        if (staged && !(lw_env && atoi(lw_env) > 0)) lcap = nblocks / 128 < 512 ? 512 : nblocks / 128 > 2048 ? 2048 : nblocks / 128;
        if (lgrid > lcap) lgrid = lcap;
        // blocks > 4 KiB, lanes beside the on-chip parsers, calls below kLaneWideBlocks (every lane gets ONE block and the call lasts as long as a lane
        // needs for it, 60-110 ms depending on how many lanes run): no lanes for the ~18 Ki blocks the two on-chip parsers get through in that time.
        // A grid with a lane for every block takes the whole queue in its first microseconds (every lane passes the "leave `reserve` blocks" check
        // before any has drawn) and the on-chip parsers get nothing: 64 Ki blocks, the lanes' kernel 110 ms, the two scalar-thread kernels beside
        // it 15 ms each.  Corpus, 64 KiB, share of the blocks with a lane 100 / 85 / 72 / 60 / 50 %, GB/s: 32 Ki blocks 32.5 / 33.5 / 33.0 / 35.2 / 37.6,
        // 48 Ki 42.5 / 43.2 / 44.1 / 46.8 / 39.2, 64 Ki 41.1 / 42.9 / 47.3 / 45.2 / 42.1 (best: all but 16-19 Ki blocks); 128 Ki and 256 Ki blocks
        // (lanes take several blocks each, the reserve works): 47.8 / 42.6 / 46.5 / 44.6 / 44.7 and 47.6-49.1, no trend.
        // (the kernel applies the same rule to the queue's length, which may be shorter than the call: blocks the scan has dealt with are not queued)
        const char *ll_env = tune("CW_LANES_LEAVE"); // blocks of such a call that get no lane (profiling knob; 0 = a lane for every block)
        const size_t leave = ll_env ? (size_t)atoi(ll_env) : kLaneLeave;
        if (!staged && leave && !(tune("CW_LANES_CONCURRENT") && tune("CW_LANES_CONCURRENT")[0] == '0') && lane_min > 1) {
            lane_leave = (uint32_t)leave;
            const size_t want = nblocks > leave + 4096 ? (nblocks - leave + 63) / 64 : 64;
            if (nblocks < kLaneWideBlocks && lgrid > want) lgrid = want;
        }
        if (wsp.lane_cap < lgrid * 64) {
            if (wsp.lane_tabs) { e = hipFree(wsp.lane_tabs); if (e != hipSuccess) return e; }
            wsp.lane_tabs = nullptr; wsp.lane_cap = 0;
            e = hipMalloc(reinterpret_cast<void **>(&wsp.lane_tabs), lgrid * 64 * (size_t)kTabBytes * 2); // (entries of 4 bytes for blocks > 4 KiB)
            if (e != hipSuccess) { // up to 4 GiB: a nearly full device does without the lanes instead of failing the call
                (void)hipGetLastError();
                wsp.lane_tabs = nullptr;
                lgrid = 0;
            } else {
                wsp.lane_cap = lgrid * 64;
            }
        }
        if (lgrid) {
        // CW_LANES_CONCURRENT=0: one after the other on the caller's stream (the lanes take the whole queue); default: side by side
        const char *cc_env = tune("CW_LANES_CONCURRENT");
        const char *rs_env = tune("CW_LANES_RESERVE");
        lanes_beside = cc_env ? cc_env[0] != '0' : true;
        uint32_t reserve = 0, reserve_wide = 0, lmin = lane_min;
        if (lanes_beside) {
            if (!wsp.side) {
                {   // The lanes' and the register-table parser's streams come from the HIGH-PRIORITY pool of hardware queues (CW_SIDE_PRIO=0: the normal one,
                    // =1: the lanes' only).  HIP multiplexes its streams onto four hardware queues per priority level, and kernels of different streams
                    // that land on one queue run one after the other.  A device-resident call has four streams and is not affected; the host pipeline has
                    // three slots with four streams each plus two for copies, and its timeline (rocprofv3 --kernel-trace) showed a chunk's two scalar-thread
                    // kernels starting the moment ITS OWN lanes kernel had ended, 108 ms late.  With the side streams in another pool a chunk's kernels
                    // no longer share a queue with each other: host path over the corpus 20.5-21.0 -> 24.0-24.3 GB/s (GPU_MAX_HW_QUEUES=8 on top: 24.7-24.9);
                    // the device-resident legs and the headline are unchanged (16 GiB corpus leg 44-48 -> 49.6).
                    const char *sp_env = tune("CW_SIDE_PRIO");
                    int least = 0, greatest = 0;
                    if ((e = hipDeviceGetStreamPriorityRange(&least, &greatest)) != hipSuccess) return e;
                    e = !(sp_env && sp_env[0] == '0') ? hipStreamCreateWithPriority(&wsp.side, hipStreamNonBlocking, greatest)
                                                   : hipStreamCreateWithFlags(&wsp.side, hipStreamNonBlocking);
                    if (e != hipSuccess) return e;
                }
                if ((e = hipEventCreateWithFlags(&wsp.fork, hipEventDisableTiming)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&wsp.join, hipEventDisableTiming)) != hipSuccess) return e;
            }
            // what the wavefront parser gets through while a lane parses its last block: 4 KiB text, 1 Mi blocks: 8 Ki..40 Ki 40-43 GB/s, 48 Ki 39.8;
            // 256 Ki blocks: 16 Ki / 28 Ki / 40 Ki 37.4 / 39.0 / 41.0
            // blocks > 4 KiB (round 3; corpus, 64 KiB, lanes alone -> lanes beside the other two, GB/s): two positions per iteration, 32 Ki blocks
            // 29.4 -> 31.0 (reserve 24 Ki), 48 Ki 34.0 -> 40.7 (16-24 Ki), 64 Ki 37.2 -> 40.6 (32 Ki); one position: 128 Ki 44.9 -> 48.3 (32 Ki), 256 Ki 42.8 -> 46.7
            reserve = rs_env && atoi(rs_env) > 0 ? (uint32_t)atoi(rs_env) : (staged ? 32768u : kLaneShare);
            if (!staged && !(rs_env && atoi(rs_env) > 0) && reserve > nblocks / 3 * 2) reserve = (uint32_t)(nblocks / 3 * 2);
            reserve_wide = rs_env && atoi(rs_env) > 0 ? (uint32_t)atoi(rs_env) : kLaneShareWide;
            if (lane_min > 1 && lmin < reserve + reserve / 4) lmin = reserve + reserve / 4; // (CW_LZ4_LANES=1 in the tests: no reserve)
            if (lane_min == 1) reserve = reserve_wide = 0;
            if ((e = hipEventRecord(wsp.fork, stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(wsp.side, wsp.fork, 0)) != hipSuccess) return e;
        }
        hipStream_t ls = lanes_beside ? wsp.side : stream;
        const uint32_t no_max = 0xFFFFFFFFu;
        const char *side_tag = lanes_beside ? " [side stream]" : "";
#define CW_RING(K, LO, HI) do { \
            const uint32_t lo_ = (LO), hi_ = (HI); \
            hipLaunchKernelGGL(lz4_lanes_ring_kernel<K>, dim3((unsigned)lgrid), dim3(64), 0, ls, src, n, src_stride, dst, dst_stride, sizes, queue, counters, \
                               wsp.lane_tabs, lo_, hi_ == no_max && lanes_ring < 0 ? reserve_wide : reserve, hi_); \
            if (hi_ == no_max) note("cw::lz4_lanes_ring_kernel<" #K "> (queue >= %u)%s", lo_, side_tag); \
            else note("cw::lz4_lanes_ring_kernel<" #K "> (queue in [%u, %u))%s", lo_, hi_, side_tag); } while (0)
        if (n <= 4096)
        {
            hipLaunchKernelGGL(lz4_lanes_kernel<kLaneTagged>, dim3((unsigned)lgrid), dim3(64), 0, ls, src, n, src_stride, dst, dst_stride, sizes, queue,
                               counters, wsp.lane_tabs, lmin, reserve);
            note("cw::lz4_lanes_kernel<1> (queue >= %u)%s", lmin, side_tag);
        }
        else if (lanes_ring < 0) {
            const uint32_t wide_from = lmin < kLaneWideBlocks ? kLaneWideBlocks : lmin;
            hipLaunchKernelGGL(lz4_lanes_ring_auto_kernel, dim3((unsigned)lgrid), dim3(64), 0, ls, src, n, src_stride, dst, dst_stride, sizes, queue, counters,
                               wsp.lane_tabs, lmin, reserve, wide_from, reserve_wide, lane_leave);
            note("cw::lz4_lanes_ring_auto_kernel (queue >= %u: two positions per iteration, >= %u: one)%s", lmin, wide_from, side_tag);
        }
        else if (lanes_ring == 1) CW_RING(1, lmin, no_max);
        else if (lanes_ring == 2) CW_RING(2, lmin, no_max);
        else if (lanes_ring == 4) CW_RING(4, lmin, no_max);
        else if (lanes_ring == 8) CW_RING(8, lmin, no_max);
#undef CW_RING
        else if (lanes_fp) {
            hipLaunchKernelGGL(lz4_lanes_kernel<kLaneFp>, dim3((unsigned)lgrid), dim3(64), 0, ls, src, n, src_stride, dst, dst_stride, sizes, queue,
                               counters, wsp.lane_tabs, lmin, reserve);
            note("cw::lz4_lanes_kernel<2> (queue >= %u)%s", lmin, side_tag);
        } else {
            hipLaunchKernelGGL(lz4_lanes_kernel<kLanePlain>, dim3((unsigned)lgrid), dim3(64), 0, ls, src, n, src_stride, dst, dst_stride, sizes, queue,
                               counters, wsp.lane_tabs, lmin, reserve);
            note("cw::lz4_lanes_kernel<0> (queue >= %u)%s", lmin, side_tag);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        lanes_used = true;
        }
    }
    // The register-table parser (lz4_vtab_kernel.hip): 16 more chains per CU than the LDS admits, no table traffic.  It runs BESIDE the
    // wavefront parser on a second stream, both pulling from the scan's queue, whenever the lanes do not take the whole queue
    // (text, 64 KiB blocks, wavefront parser alone -> both: 3,233 blocks 11.8 -> 15.7 GB/s, 8 Ki 13.3 -> 21.6, 16 Ki 24.5 against the
    // lanes' 19.5; beside the lanes in their random-line regime it gains nothing -- they keep the memory system busy and its
    // candidate fetches wait).  CW_LZ4_VTAB: 0 = off, 1 = on the caller's stream AHEAD of the wavefront parser (it takes the whole
    // queue: tests), 2 = beside (default); CW_VTAB_MIN / CW_VTAB_MAX = queue lengths between which it runs (checked on the device),
    // CW_VTAB_RESERVE = blocks it leaves to the others, CW_VTAB_WPC = its wavefronts per CU (at most 16), CW_VTAB_GEN = kernel generation.
    const char *vt_env = tune("CW_LZ4_VTAB");
    const int vt_mode = vt_env ? atoi(vt_env) : 2;
    const bool cut_first = mode && strcmp(mode, "cut") == 0; // (CW_LZ4_MODE=cut: the first-generation parser only)
    bool vtab_used = false, vtab_beside = false;
    if (vt_mode > 0 && !use_fp && !cut_first && n >= 64 && nblocks >= 64 && ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) == 0) {
        const char *vm_env = tune("CW_VTAB_MIN"), *vx_env = tune("CW_VTAB_MAX"), *vr_env = tune("CW_VTAB_RESERVE"), *vw_env = tune("CW_VTAB_WPC");
        // LDS-staged blocks: a small queue is the LDS-resident parser's (4 Ki blocks of 4 KiB: 19.4 GB/s alone against 14.5 with the register-table
        // parser's 4,096 wavefronts taking a block each; 16 Ki blocks 23.8 -> 24.5, 32 Ki 25.1 -> 28.0, 51,728 25.6 -> 29.5)
        const uint32_t vmin = vm_env ? (uint32_t)atoi(vm_env) : staged ? 12288u : 1u, vres = vr_env ? (uint32_t)atoi(vr_env) : 0u;
        // lanes that take the whole queue (blocks > 4 KiB) start at lane_min queued blocks: the register-table parser stays below
        const uint32_t vmax = vx_env ? (uint32_t)atoi(vx_env) : (lanes_used && !lanes_beside ? lane_min : 0xFFFFFFFFu);
        const unsigned vwpc = vw_env && atoi(vw_env) > 0 ? (unsigned)atoi(vw_env) : 16u;
        hipStream_t vs = stream;
        if (vt_mode == 2) {
            if (!wsp.side2) {
                {
                    const char *sp_env = tune("CW_SIDE_PRIO");
                    int least = 0, greatest = 0;
                    if ((e = hipDeviceGetStreamPriorityRange(&least, &greatest)) != hipSuccess) return e;
                    e = !(sp_env && sp_env[0] != '2') ? hipStreamCreateWithPriority(&wsp.side2, hipStreamNonBlocking, greatest)
                                                   : hipStreamCreateWithFlags(&wsp.side2, hipStreamNonBlocking);
                    if (e != hipSuccess) return e;
                }
                if ((e = hipEventCreateWithFlags(&wsp.fork2, hipEventDisableTiming)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&wsp.join2, hipEventDisableTiming)) != hipSuccess) return e;
            }
            if ((e = hipEventRecord(wsp.fork2, stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(wsp.side2, wsp.fork2, 0)) != hipSuccess) return e;
            vs = wsp.side2;
            vtab_beside = true;
        }
        const char *vname = nullptr;
        if ((e = lz4_vtab_launch(src, n, src_stride, nblocks, dst, dst_stride, sizes, queue, counters, vmin, vmax, vres, vwpc, vs, &vname)) != hipSuccess) return e;
        if (vmax != 0xFFFFFFFFu) note("%s (queue < %u)%s", vname, vmax, vtab_beside ? " [side stream]" : "");
        else note("%s%s", vname, vtab_beside ? " [side stream]" : "");
        vtab_used = true;
    }
    const char *pwpc_env = tune("CW_PARSE_WPC"); // parse wavefronts per CU (profiling knob; default: all the LDS admits)
    const size_t pwpc = pwpc_env && atoi(pwpc_env) > 0 ? (size_t)atoi(pwpc_env) : 10;
    const size_t want = 256 * (per_cu > pwpc ? pwpc : per_cu);
    const size_t grid = nblocks < want ? nblocks : want;
    // CW_LZ4_MODE=cut parses with the first-generation (write/read-back) kernel only (profiling knob)
    const bool cut_only = mode && strcmp(mode, "cut") == 0;
    // CW_LZ_FORCE_REDO=1: the exchange kernel hands every block back, as if its lane-order check had failed (test knob)
    const char *redo_env = tune("CW_LZ_FORCE_REDO");
    const uint32_t force_redo = redo_env && atoi(redo_env) > 0 ? 1u : 0u;
    if (!cut_only) {
        // blocks read from global memory: the scalar-thread parser with its table in LDS (lz4_vtab3_kernel<true>) in the place of the round-2
        // wavefront parser; CW_LZ4_LTAB=0 keeps the latter (and the forced-redo test knob and unaligned sources need it)
        const char *lt_env = tune("CW_LZ4_LTAB");
        const bool use_ltab = !staged && !use_fp && !force_redo && n >= 64 && ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) == 0 &&
                              (lt_env ? atoi(lt_env) != 0 : kLtabDefault);
        if (use_ltab) {
            note("cw::lz4_vtab3_kernel<true>");
            if ((e = lz4_ltab_launch(src, n, src_stride, nblocks, dst, dst_stride, sizes, queue, counters, (unsigned)pwpc, stream)) != hipSuccess) return e;
        } else {
        note(staged ? "cw::lz4_parse_kernel<true>" : use_fp ? "cw::lz4_parse_fp_kernel<%d>" : "cw::lz4_parse_kernel<false>", headw == 32 || headw == 8 ? headw : 16);
        if (staged)
            hipLaunchKernelGGL(lz4_parse_kernel<true>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, queue, counters, queue2, force_redo);
        else if (use_fp && headw == 32)
            hipLaunchKernelGGL(lz4_parse_fp_kernel<32>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, queue, counters, queue2, force_redo);
        else if (use_fp && headw == 8)
            hipLaunchKernelGGL(lz4_parse_fp_kernel<8>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, queue, counters, queue2, force_redo);
        else if (use_fp)
            hipLaunchKernelGGL(lz4_parse_fp_kernel<16>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, queue, counters, queue2, force_redo);
        else
            hipLaunchKernelGGL(lz4_parse_kernel<false>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, queue, counters, queue2, force_redo);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    }
    if (vtab_used && vtab_beside) {
        if ((e = hipEventRecord(wsp.join2, wsp.side2)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, wsp.join2, 0)) != hipSuccess) return e;
    }
    if (lanes_used && lanes_beside) { // the redo pass and the caller's later work wait for the lanes too
        if ((e = hipEventRecord(wsp.join, wsp.side)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, wsp.join, 0)) != hipSuccess) return e;
    }
    if (cut_only) note(staged ? "cw::lz4_blocks_kernel<true>" : "cw::lz4_blocks_kernel<false>");
    note_kernels(0, launched); // (the redo pass below finds an empty list unless the LDS ever applied an exchange's lanes out of order)
    // blocks the exchange-based parser handed back (none, unless the LDS ever applies lanes out of order)
    const uint32_t *q = cut_only ? queue : queue2;
    uint32_t *c = cut_only ? counters : counters + 4;
    if (staged)
        hipLaunchKernelGGL(lz4_blocks_kernel<true>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                           dst_stride, sizes, q, c);
    else
        hipLaunchKernelGGL(lz4_blocks_kernel<false>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                           dst_stride, sizes, q, c);
    return hipGetLastError();
}

} // namespace cw
