// lzf_kernel.hip -- bit-exact LZF compression (liblzf 3.x compressor as the reference builds it: HLOG 16,
// VERY_FAST, offsets in the table; src/compression_perf/include/lzf/lzfP.h:54-92) for gfx950, one storage
// block per wavefront.
//
// Replaces the reference's lzf slot  lzf_compress(s, l, d, l - 1)
// (src/hashandcompress/HashAndCompress.cpp:344-347, src/compression_perf/src/experiment.cpp:110; API
// src/compression_perf/include/lzf/lzf.h:79-81).  Semantics are those of SURVEY.md 8(a) row A6 as restated on the
// CPU in oracle/lzf_oracle.c, with the hash table taken as zero-initialised (hard part 4).
//
// Mapping to the machine.  Like LZ4 the parse is a serial walk over a mutable hash table -- here 65,536 slots that
// see EVERY position -- so a wavefront owns one block; lane j speculatively handles position ip+j, the first lane
// whose reference passes the parser's match test ends the batch, the lanes before it become literals (run control
// bytes placed by closed form), match extension / emission are done wave-wide.  The reference's out_len = l - 1
// makes incompressible blocks return 0; every one of the parser's overflow checks is reproduced, so that verdict
// is exact too.
//
// Kernels (DESIGN.md 4.4 has the measurements):
//   lzf_lanes_kernel   one block per LANE, liblzf's loop as it stands, 128 KiB table per lane in global memory.  Blocks
//                      > 4 KiB, from 12 Ki blocks on: the whole batch.  Blocks <= 4 KiB, from 28 Ki blocks on: BESIDE the
//                      link/chain rounds on a second stream, the lanes pulling from the top of the batch while the rounds
//                      climb from the bottom (LaneShare);
//   lzf_links_kernel + lzf_chain_kernel   (everything else from 16 bytes on) per-position "previous position with my slot"
//                      (128 KiB table, throughput bound), then the parse on link chains + skip flags with no table: blocks
//                      <= 4 KiB with links and block in LDS (12.3 KiB, 13 blocks per CU), larger ones with only the skip
//                      bits in LDS -- and, since round 3, parsed by lzf_sthread_kernel: the same algorithm with the wavefront
//                      run as one scalar thread (the wavefront-wide lzf_chain_kernel<true> is kept as CW_LZF_STHREAD=0);
//   lzf_parse_kernel   (blocks < 16 bytes; CW_LZF_MODE=table) 128 KiB table in LDS, one block per CU; table operation of a
//                      batch = one ds_mskor_rtn_b32 exchange, lanes in ascending order, verified per batch;
//   redo:              lzf_blocks_kernel, the first-generation parser (write/read-back collision detection, batch
//                      cut, rollback) for blocks whose lane-order check failed (never observed; forced in the tests).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "cw_device.h"
#include "lz_device.h"
#include "scalar_thread.h"

namespace cw {

namespace {

constexpr uint32_t kLzfSlots = 1u << 16, kLzfTabBytes = kLzfSlots * 2;
constexpr uint32_t kMaxOff = 1u << 13, kMaxRef = (1u << 8) + (1u << 3), kMaxLit = 32;
constexpr uint32_t kInLdsMax = 16384; // blocks up to this size are staged in LDS next to the table
constexpr uint32_t kRedo = 0xFFFFFFFFu; // sizes[] marker: exchange kernel -> write/read-back kernel

// Small blocks, large batches: the lane-per-block parser runs BESIDE the link/chain rounds (second stream).  The rounds walk
// the batch from block 0 upwards, the lanes take blocks from the top downwards.  ONE 64-bit word W holds both frontiers -- low
// half: ranks the lanes have drawn (rank k is block total-1-k), high half: blocks the rounds have claimed -- and both sides
// move theirs with ONE fetch-and-add each, so each sees the other's frontier as of the instant of its own move:
//   * a round's first kernel adds the round's length to `claimed` (rounds run in order over contiguous ranges) before it
//     touches a block; the value returned is the lanes' `taken` AT THE CLAIM.  Workgroup 0 does it and publishes that number
//     with the round's sequence number; the other workgroups and the round's chain kernel use the published number.  The
//     round parses exactly the blocks of its range with rank >= that `taken`.
//   * a wavefront of lanes adds the number of its idle lanes to `taken`; a drawn rank is the lane's to parse if its block lay
//     above `claimed` as returned by the same add, and is dropped otherwise.
// A block drawn before a round's claim is counted in that round's `taken` and skipped by it; a block drawn after the claim
// is either above the round (kept) or inside a claimed range (dropped -- and its rank is >= the `taken` of the round that
// owns it, which therefore parses it).  Whatever the interleaving, every block is parsed by exactly one side, with no retry
// loop anywhere: a compare-and-swap version of this collapsed under its own retries (a thousand wavefronts and the rounds'
// workgroups on one word: 32 -> 5 GB/s), and a check followed by a separate add let a full grid of lanes pass the check at
// once and meet the rounds.  `reserve` (how many unclaimed blocks the lanes leave alone) is a matter of speed only.
// ctr == nullptr: no lanes beside (the rounds take everything).
// counter block of a stream's workspace (dword indices; each group on a 128-byte line of its own -- they are hammered by different
// parties): [0] the round's pull counter, [32..33] W, [64] poor, [65] fine, [66] handed back, [96..97] published, [98] claim ticket,
// [99] failed
constexpr uint32_t kCtrWord = 32, kCtrPoor = 64, kCtrFine = 65, kCtrHanded = 66, kCtrPublished = 96, kCtrTicket = 98, kCtrFailed = 99, kCtrBytes = 512;
constexpr uint32_t kShareGaveUp = 0xFFFFFFFFu; // share_round_taken: the published number never came (see there)
constexpr uint32_t kShareSpinCap = 1u << 20;   // polls of ~64 cycles: tens of milliseconds, against the microsecond a claim takes
struct LaneShare { uint32_t *ctr; size_t round_first, total; uint32_t seq, spin_cap; };
__device__ __forceinline__ unsigned long long *share_word(uint32_t *ctr) { return reinterpret_cast<unsigned long long *>(ctr + kCtrWord); }
__device__ __forceinline__ unsigned long long *share_published(uint32_t *ctr) { return reinterpret_cast<unsigned long long *>(ctr + kCtrPublished); }
__device__ __forceinline__ bool lanes_took(const LaneShare &sh, uint32_t taken, size_t blk)
{
    return sh.ctr && sh.round_first + blk >= sh.total - taken;
}
// The lanes' `taken` as of this round's claim (claim == true: the round's first kernel; every lane of the grid calls it).
// The claim is made by whichever workgroup of the round's first kernel ARRIVES first (a ticket moved from seq - 1 to seq with one
// compare-and-swap: rounds run in stream order, so the ticket of round seq - 1 is final when round seq starts) -- no assumption
// about which workgroup the dispatcher starts first: the others wait for a workgroup that is, by construction, running and a
// few instructions away from publishing.  The wait is bounded all the same: after spin_cap polls the caller gets kShareGaveUp,
// the batch's `failed` word is set and the caller leaves its blocks alone; the final lzf_blocks_kernel pass then parses EVERY
// block of the batch again (slow and exact), as it does for blocks whose lane-order check failed.  (spin_cap = 0 forces that path
// for every workgroup but the claimant: CW_LZF_SHARE_GIVE_UP=1, a test knob.)
__device__ __forceinline__ uint32_t share_round_taken(const LaneShare &sh, size_t nblocks, bool claim)
{
    unsigned long long *pub = share_published(sh.ctr);
    if (claim && threadIdx.x == 0 && atomicCAS(sh.ctr + kCtrTicket, sh.seq - 1, sh.seq) == sh.seq - 1) {
        const unsigned long long old = atomicAdd(share_word(sh.ctr), (unsigned long long)nblocks << 32);
        __hip_atomic_store(pub, ((unsigned long long)sh.seq << 32) | (uint32_t)old, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long p = __hip_atomic_load(pub, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t polls = 0; (uint32_t)(p >> 32) != sh.seq; polls++) {
        if (polls >= sh.spin_cap) {
            if (threadIdx.x == 0) __hip_atomic_store(sh.ctr + kCtrFailed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return kShareGaveUp;
        }
        __builtin_amdgcn_s_sleep(2);
        p = __hip_atomic_load(pub, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __builtin_amdgcn_readfirstlane((uint32_t)p);
}
// m ranks for a wavefront of lanes: returns the first rank and the rounds' `claimed` as of the draw; 0 ranks (first = total)
// when fewer than m + reserve unclaimed blocks are left
__device__ __forceinline__ void share_take(uint32_t *ctr, uint32_t m, size_t total, uint32_t reserve, uint32_t &first, uint32_t &claimed)
{
    unsigned long long *w = share_word(ctr);
    const unsigned long long seen = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    first = (uint32_t)total; claimed = 0;
    if ((uint64_t)(uint32_t)seen + (uint32_t)(seen >> 32) + reserve + m > total) return;
    const unsigned long long old = atomicAdd(w, (unsigned long long)m);
    first = (uint32_t)old; claimed = (uint32_t)(old >> 32);
}

// A round of the link/chain kernels over listed blocks (those the lanes handed back) instead of a contiguous range
struct BlockList {
    const uint32_t *queue, *count; uint32_t first;
    __device__ __forceinline__ size_t at(size_t blk) const { return queue ? queue[first + blk] : blk; }
};

__device__ __forceinline__ uint32_t ctz64(unsigned long long m) { return m ? (uint32_t)__builtin_ctzll(m) : 64u; }
__device__ __forceinline__ uint32_t lzf_slot(uint32_t b0, uint32_t b1, uint32_t b2)
{
    // IDX(hval) = ((hval >> 8) - hval*5) & 0xFFFF with hval = b0<<16 | b1<<8 | b2 (VERY_FAST, HLOG 16)
    return (((b0 << 8) | b1) - (((b1 << 8) | b2) * 5u)) & 0xFFFFu;
}

// m literals starting at in[ip]: bytes and completed-run control bytes; updates (op, lit)
__device__ __forceinline__ void put_literals(uint8_t *__restrict__ out, const uint8_t *in, uint32_t ip, uint32_t m,
                                             uint32_t &op, uint32_t &lit, uint32_t lane)
{
    for (uint32_t i = lane; i < m; i += 64) {
        const uint32_t t = lit + i, p = op + i + t / kMaxLit;
        out[p] = in[ip + i];
        if ((t + 1) % kMaxLit == 0) out[p - kMaxLit] = kMaxLit - 1; // this byte completed a run of 32
    }
    op += m + (lit + m) / kMaxLit; // every completed run also reserved the next control byte
    lit = (lit + m) % kMaxLit;
}

} // namespace

__global__ void __launch_bounds__(64)
lzf_blocks_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                  uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes, uint32_t in_lds, uint32_t only_marked, const uint32_t *__restrict__ failed)
{
    // failed: the batch's "a round's claim was never seen" word (share_round_taken): set => every block is parsed again here
    const bool everything = failed && __builtin_amdgcn_readfirstlane(*failed) != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    uint8_t *stage = smem + kLzfTabBytes;
    const uint32_t lane = threadIdx.x;
    const uint32_t cap = n - 1; // out_len of the reference's call

    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        // second pass behind lzf_parse_kernel: only the blocks it handed back (none, unless the LDS ever applies the
        // lanes of an exchange out of order)
        if (only_marked && !everything && __builtin_amdgcn_readfirstlane(sizes[blk]) != kRedo) continue;
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;

        __syncthreads();
        for (uint32_t i = lane; i < kLzfTabBytes / 16; i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0, 0, 0, 0);
        const uint8_t *in = g;
        if (in_lds) {
            for (uint32_t i = lane; i < n; i += 64) stage[i] = g[i];
            in = stage;
        }
        __syncthreads();

        uint32_t ip = 0, op = 1, lit = 0; // op = 1: the first literal run's control byte is reserved
        bool fail = (n == 0 || cap == 0);

        while (!fail && ip + 2 < n) {
            // ---- one batch: lane j takes position ip + j ----
            const uint32_t pos = ip + lane;
            const bool valid = pos + 2 < n;
            const uint32_t nvalid = ctz64(~__ballot(valid));
            uint32_t b0 = 0, b1 = 0, b2 = 0, slot = 0, old = 0;
            if (valid) {
                b0 = in[pos]; b1 = in[pos + 1]; b2 = in[pos + 2];
                slot = lzf_slot(b0, b1, b2);
                old = tab[slot];
                tab[slot] = (uint16_t)pos;
            }
            __syncthreads(); // keeps hipcc from forwarding the lane's own store to the read-back
            const bool lost = valid && tab[slot] != (uint16_t)pos;
            uint32_t m = ctz64(__ballot(lost));
            if (m == 0) m = 1;
            const uint32_t L = m < nvalid ? m : nvalid;

            // the parser's match test (ref < ip holds by construction; "ref > in_data" excludes the empty slot)
            bool is_match = false;
            if (lane < L && old > 0 && pos - old - 1 < kMaxOff)
                is_match = in[old + 2] == b2 && in[old] == b0 && in[old + 1] == b1;
            const unsigned long long mm = __ballot(is_match);
            const uint32_t w = ctz64(mm);
            const uint32_t ncommit = mm ? w + 1 : L;
            if (ncommit < nvalid) { // undo speculative table writes, then re-assert the committed ones
                if (valid && lane >= ncommit) tab[slot] = (uint16_t)old;
                if (lane < ncommit) tab[slot] = (uint16_t)pos;
            }

            // ---- the positions before the match (or the whole batch) are literals ----
            const uint32_t nlit = mm ? w : L;
            if (nlit) {
                // each literal checks op < out_end before it is stored; positions grow, so test the last one
                const uint32_t last = op + (nlit - 1) + (lit + nlit - 1) / kMaxLit;
                if (last >= cap) { fail = true; break; }
                put_literals(out, in, ip, nlit, op, lit, lane);
                ip += nlit;
            }
            if (!mm) continue;

            // ---- match at ip against ref ----
            const uint32_t ref = __builtin_amdgcn_readlane(old, w);
            uint32_t maxlen = n - ip - 2;
            if (maxlen > kMaxRef) maxlen = kMaxRef;
            if (op + 4 >= cap && op - (lit == 0) + 4 >= cap) { fail = true; break; }
            if (lit) { if (lane == 0) out[op - lit - 1] = (uint8_t)(lit - 1); } // stop run
            else op -= 1;                                                        // undo an empty run

            // eq = number of equal bytes from index 3 on (bounded by the block end)
            uint32_t eq = 0;
            for (;;) {
                const uint32_t t = 3 + eq + lane;
                const bool ok = ip + t < n && t < kMaxRef + 2 && in[ref + t] == in[ip + t];
                const uint32_t cnt = ctz64(~__ballot(ok));
                eq += cnt;
                if (cnt < 64) break;
            }
            uint32_t len; // matched octets, with the reference's unrolled-compare overshoot
            if (maxlen > 16) {
                if (eq < 16) len = 3 + eq;
                else { len = 3 + eq < maxlen ? 3 + eq : maxlen; if (len < 19) len = 19; }
            } else {
                len = 3 + eq < maxlen ? 3 + eq : maxlen;
                if (len < 3) len = 3;
            }
            const uint32_t off = ip - ref - 1, l2 = len - 2;
            if (lane == 0) {
                if (l2 < 7) {
                    out[op] = (uint8_t)((off >> 8) + (l2 << 5));
                    out[op + 1] = (uint8_t)off;
                } else {
                    out[op] = (uint8_t)((off >> 8) + (7u << 5));
                    out[op + 1] = (uint8_t)(l2 - 7);
                    out[op + 2] = (uint8_t)off;
                }
            }
            op += l2 < 7 ? 2 : 3;
            lit = 0; op += 1; // start run
            ip += len;
            if (ip + 2 >= n) break;
            // VERY_FAST: only the last two positions of the match enter the table
            for (uint32_t q = ip - 2; q < ip; q++) tab[lzf_slot(in[q], in[q + 1], in[q + 2])] = (uint16_t)q;
        }

        if (!fail) {
            if (op + 3 > cap) { // at most 3 bytes can be missing here
                fail = true;
            } else {
                if (ip < n) put_literals(out, in, ip, n - ip, op, lit, lane);
                if (lit) { if (lane == 0) out[op - lit - 1] = (uint8_t)(lit - 1); } // end run
                else op -= 1;
            }
        }
        if (lane == 0) sizes[blk] = fail ? 0u : op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Second generation (same idea as lz4_parse_kernel): the table operation of a batch is one ds_mskor_rtn_b32 -- a
// masked 16-bit exchange whose lanes the LDS applies in ascending order -- so every lane gets back exactly the
// reference the serial parser would have read at its position, including the stores of earlier lanes of the batch:
// no write/read-back round and no cutting of batches.  The order is verified on every batch (a returned reference
// >= the lane's own position cannot occur serially); a block where it fails is marked and redone by
// lzf_blocks_kernel.  Lanes 0 and 1 of a batch re-insert the last two positions of the previous match (VERY_FAST),
// lane j >= 2 takes position ip + j - 2.  Per batch: one round for the 4 bytes at every position (requested before
// the previous sequence is emitted; their low bytes are the literals, stored from registers), one exchange, one
// round for the 16 bytes at position and reference (match test + the first 9 bytes of the extension).
// Staged blocks are read as aligned dwords + v_alignbyte (unaligned ds_reads serialise the LDS).
// ---------------------------------------------------------------------------------------------------
template <bool STAGED>
__global__ void __launch_bounds__(64)
lzf_parse_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks,
                 uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ sizes, uint32_t force_redo)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    uint8_t *stage = smem + kLzfTabBytes;
    const uint32_t tab_lds = (uint32_t)reinterpret_cast<uintptr_t>(tab);
    const uint32_t lane = threadIdx.x;
    const uint32_t cap = n - 1; // out_len of the reference's call

    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        if (force_redo) { // test knob: behave as if the lane-order check had failed
            if (lane == 0) sizes[blk] = kRedo;
            continue;
        }
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;

        __syncthreads();
        for (uint32_t i = lane; i < kLzfTabBytes / 16; i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0, 0, 0, 0);
        const uint8_t *in = STAGED ? stage : g;
        if (STAGED) {
            for (uint32_t i = lane; i < n; i += 64) stage[i] = g[i];
            if (lane < 16) stage[n + lane] = 0; // slack read by the dword loads (never compared as data)
        }
        __syncthreads();

        uint32_t ip = 0, op = 1, lit = 0; // op = 1: the first literal run's control byte is reserved
        bool fail = (n == 0 || cap == 0), broken = false, has_q = false;

        // the 4 bytes at every position of a batch; a global read must not pass the end of the block
        auto request = [&](uint32_t ip_) __attribute__((always_inline)) -> uint32_t {
            const uint32_t pos = ip_ - 2 + lane;
            if (STAGED) return lz::rd32x<true>(in, (int32_t)pos < 0 ? 0u : pos + 2 < n ? pos : 0u);
            const uint32_t p = (int32_t)pos < 0 || pos + 2 >= n ? 0u : pos;
            const uint32_t q = p + 4 <= n ? p : p - 1; // pos = n - 3: read one byte earlier and shift
            return n >= 4 ? lz::rd32(in, q) >> ((p - q) * 8) : 0u;
        };
        uint32_t vnext = (!fail && n >= 4) ? request(0) : 0u;
        if (n < 4) { // too small for the dword reads (cannot compress anyway): byte-wise values
            const uint32_t pos = lane - 2;
            vnext = 0;
            if ((int32_t)pos >= 0 && pos + 2 < n) vnext = in[pos] | (in[pos + 1] << 8) | (in[pos + 2] << 16);
        }

        while (!fail && ip + 2 < n) {
            // ---- one batch: lanes 0,1 re-insert ip-2, ip-1 (after a match), lane j >= 2 takes position ip + j - 2 ----
            const uint32_t pos = ip - 2 + lane;
            const bool tested = lane >= 2 && pos + 2 < n;
            const bool active = tested || (lane < 2 && has_q);
            const uint32_t ntest = (uint32_t)__builtin_popcountll(__ballot(tested));
            const uint32_t v = vnext;
            const uint32_t b0 = v & 0xFFu, b1 = (v >> 8) & 0xFFu, b2 = (v >> 16) & 0xFFu;
            const uint32_t slot = lzf_slot(b0, b1, b2);
            uint32_t old = 0;
            if (active) old = lz::tab_exchange(tab_lds, slot, pos);
            if (__ballot(tested && old >= pos && (old | pos) != 0)) { broken = true; break; } // (position 0 finds the empty slot: 0)

            // the parser's match test ("ref > in_data" excludes the empty slot) + the neighbourhood for the extension
            const bool cand = tested && old > 0 && pos - old - 1 < kMaxOff;
            const bool wide = STAGED || pos + 12 <= n; // all 16 bytes readable (reference < position)
            lz::Around ap, ac;
            ap.before = 0; ap.at = 0; ap.after = 0; ac.before = 0; ac.at = 1u << 24; ac.after = 1;
            if (cand) {
                if (wide) {
                    ac = lz::around<STAGED>(in, old, false);
                    ap = lz::around<STAGED>(in, pos, false);
                } else {
                    ac.at = in[old] | (in[old + 1] << 8) | (in[old + 2] << 16) | (~v & 0xFF000000u); // 4th byte: "differs"
                }
            }
            const unsigned long long mm = __ballot(cand && ((ac.at ^ v) & 0xFFFFFFu) == 0);
            const uint32_t w = mm ? (uint32_t)__builtin_ctzll(mm) : 64u;

            // ---- the positions before the match (or the whole batch) are literals ----
            const uint32_t nlit = mm ? w - 2 : ntest;
            uint32_t ref = 0, mpos = 0, eqs = 0;
            bool more_eq = false;
            if (mm) {
                const uint32_t pm = __builtin_amdgcn_readlane(pos | (old << 16), w);
                mpos = pm & 0xFFFFu; ref = pm >> 16;
                // undo the speculative stores behind the match (first lane of each slot's group restores the slot)
                if (active && lane > w && old <= mpos) tab[slot] = (uint16_t)old;
                // equal bytes from index 3 on: byte 3 is the top byte of `at`, bytes 4..11 are `after`
                const uint64_t x = ap.after ^ ac.after;
                uint32_t e = (ac.at ^ v) >> 24 ? 0u : 1u + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u);
                bool me = wide ? e == 9 : true; // not wide: nothing was compared beyond the 3 bytes
                if (!wide) e = 0;
                const uint32_t packed = __builtin_amdgcn_readlane(e | ((uint32_t)me << 8), w);
                eqs = packed & 0xFFu; more_eq = (packed >> 8) & 1u;
            }
            if (nlit) {
                // each literal checks op < out_end before it is stored; positions grow, so test the last one
                const uint32_t last = op + (nlit - 1) + (lit + nlit - 1) / kMaxLit;
                if (last >= cap) { fail = true; break; }
                const uint32_t i = lane - 2; // literal i = low byte of lane i + 2
                if (i < nlit) {
                    const uint32_t t = lit + i, p = op + i + t / kMaxLit;
                    out[p] = (uint8_t)b0;
                    if ((t + 1) % kMaxLit == 0) out[p - kMaxLit] = kMaxLit - 1; // this byte completed a run of 32
                }
                op += nlit + (lit + nlit) / kMaxLit; // every completed run also reserved the next control byte
                lit = (lit + nlit) % kMaxLit;
                ip += nlit;
            }
            if (!mm) { // no match among the tested positions: next batch starts behind them, nothing to re-insert
                has_q = false;
                if (ip + 2 < n) vnext = request(ip);
                continue;
            }

            // ---- match at ip (= mpos) against ref ----
            uint32_t maxlen = n - ip - 2;
            if (maxlen > kMaxRef) maxlen = kMaxRef;
            if (op + 4 >= cap && op - (lit == 0) + 4 >= cap) { fail = true; break; }
            if (lit) { if (lane == 0) out[op - lit - 1] = (uint8_t)(lit - 1); } // stop run
            else op -= 1;                                                        // undo an empty run

            // eq = number of equal bytes from index 3 on (bounded by the block end and the longest reference)
            uint32_t eq = eqs;
            {
                const uint32_t room = (n - ip < kMaxRef + 2 ? n - ip : kMaxRef + 2) - 3; // indices 3 .. room+2 may be compared
                if (eq >= room) { eq = room; more_eq = false; }
            }
            while (more_eq) {
                const uint32_t t = 3 + eq + lane;
                const bool ok = ip + t < n && t < kMaxRef + 2 && in[ref + t] == in[ip + t];
                const uint32_t cnt = ctz64(~__ballot(ok));
                eq += cnt;
                if (cnt < 64) break;
            }
            uint32_t len; // matched octets, with the reference's unrolled-compare overshoot
            if (maxlen > 16) {
                if (eq < 16) len = 3 + eq;
                else { len = 3 + eq < maxlen ? 3 + eq : maxlen; if (len < 19) len = 19; }
            } else {
                len = 3 + eq < maxlen ? 3 + eq : maxlen;
                if (len < 3) len = 3;
            }
            const uint32_t off = ip - ref - 1, l2 = len - 2, at = op;
            op += l2 < 7 ? 2 : 3;
            lit = 0; op += 1; // start run
            ip += len;
            has_q = true;
            const bool go_on = ip + 2 < n;
            if (go_on) vnext = request(ip); // before the stores: they do not wait for it
            if (lane == 0) {
                if (l2 < 7) {
                    out[at] = (uint8_t)((off >> 8) + (l2 << 5));
                    out[at + 1] = (uint8_t)off;
                } else {
                    out[at] = (uint8_t)((off >> 8) + (7u << 5));
                    out[at + 1] = (uint8_t)(l2 - 7);
                    out[at + 2] = (uint8_t)off;
                }
            }
            if (!go_on) break;
        }

        if (broken) {
            if (lane == 0) sizes[blk] = kRedo;
            continue;
        }
        if (!fail) {
            if (op + 3 > cap) { // at most 3 bytes can be missing here
                fail = true;
            } else {
                if (ip < n) put_literals(out, in, ip, n - ip, op, lit, lane);
                if (lit) { if (lane == 0) out[op - lit - 1] = (uint8_t)(lit - 1); } // end run
                else op -= 1;
            }
        }
        if (lane == 0) sizes[blk] = fail ? 0u : op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Small blocks (<= 16 KiB; the reference's own block size is 4 KiB): parse WITHOUT the 128 KiB table, so that a CU
// holds a dozen blocks instead of one.  The table answers "latest INSERTED earlier position with my slot".  Which
// positions are inserted depends on the parse (the inside of a match is skipped, its last two positions are not),
// but "latest EARLIER position with my slot, inserted or not" does not: that is a per-position link that can be
// computed for the whole block up front.  The parser's reference is then the first INSERTED position on the chain
// link[p], link[link[p]], ..., and "skipped" is one flag per position set when a match is emitted.
//   lzf_links_kernel  (one wavefront and one 128 KiB table per CU, but throughput- not latency-bound: 256 positions
//                      per exchange round, loads one round ahead)  writes link[] (u16 per position) to a workspace;
//   lzf_chain_kernel  holds link[] + flags (2n bytes) and the block (n bytes) in LDS: 12.3 KiB per 4 KiB block.  There is no
//                      table write during the parse, hence no rollback and no ordering assumption in this kernel;
//                      lanes walk their chains in lockstep (one aligned u16 read per step); a batch speculates on
//                      kChainHead positions after a match (doubling while none is found) so that the slowest chain
//                      of a batch stays short.
// The lane-order check sits in lzf_links_kernel (a link >= its own position); a failing block is marked and parsed
// by lzf_blocks_kernel like everywhere else.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kChainMax = 16384, kChainHead = 8, kSkipFlag = 0x8000u;

// 3 bytes at pos (as the low 24 bits) from global memory without reading past the block
__device__ __forceinline__ uint32_t load3(const uint8_t *g, uint32_t n, uint32_t pos, bool ok)
{
    const uint32_t p = ok ? pos : 0u;
    const uint32_t q = p + 4 <= n ? p : p - 1; // pos = n - 3: read one byte earlier and shift
    return lz::rd32(g, q) >> ((p - q) * 8);
}

__global__ void __launch_bounds__(64)
lzf_links_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks, uint16_t *__restrict__ links,
                 uint32_t n2, uint32_t *__restrict__ sizes, uint32_t force_redo, LaneShare share, BlockList list)
{
    if (list.queue) { // a round over handed-back blocks: entries [list.first, list.first + nblocks) of the queue, as far as it is filled
        const uint32_t cnt = *list.count;
        if (list.first >= cnt) return;
        if (nblocks > cnt - list.first) nblocks = cnt - list.first;
    }
    uint32_t taken = 0;
    if (share.ctr) { // claim this round's blocks and learn what the lanes held at that instant
        taken = share_round_taken(share, nblocks, true);
        if (taken == kShareGaveUp || share.round_first >= share.total - taken) return; // (gave up: the final pass parses the batch) the whole round is theirs
    }
    // LDS: the 128 KiB table, then the block (coalesced copy; positions are then read as aligned dwords)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *stage = smem + kLzfTabBytes; // kChainMax + 48 bytes
    const uint32_t tab_lds = (uint32_t)reinterpret_cast<uintptr_t>(smem);
    const uint32_t lane = threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | src_stride) & 15) == 0 && n <= 8192;
    // the next block's bytes are requested while this block is linked (vec: up to 8 KiB in registers)
    uint4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7; // named, not an array: hipcc put the array in scratch
    pre0 = pre1 = pre2 = pre3 = pre4 = pre5 = pre6 = pre7 = make_uint4(0, 0, 0, 0);
    // unconditional loads (a load inside a branch is waited for on the spot): past the block or the batch they
    // re-read an in-range address and the value is never used
#define CW_PRE1(K, G4, LAST) { const uint32_t i_ = 64 * K + lane; pre##K = (G4)[i_ <= (LAST) ? i_ : (LAST)]; }
#define CW_PREFETCH(BLK)                                                                                   \
    do {                                                                                                   \
        const size_t b_ = (BLK) < nblocks ? (BLK) : nblocks - 1;                                           \
        const uint4 *g4 = reinterpret_cast<const uint4 *>(src + list.at(b_) * src_stride);                 \
        const uint32_t last_ = (n - 1) / 16;                                                               \
        CW_PRE1(0, g4, last_) CW_PRE1(1, g4, last_) CW_PRE1(2, g4, last_) CW_PRE1(3, g4, last_)            \
        CW_PRE1(4, g4, last_) CW_PRE1(5, g4, last_) CW_PRE1(6, g4, last_) CW_PRE1(7, g4, last_)            \
    } while (0)
#define CW_STAGE1(K) if ((64 * K + lane) * 16 < n) reinterpret_cast<uint4 *>(stage)[64 * K + lane] = pre##K;
    if (vec) CW_PREFETCH(blockIdx.x);

    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        if (lanes_took(share, taken, blk)) { // not ours: keep the prefetch pipeline in step and move on
            if (vec) CW_PREFETCH(blk + gridDim.x);
            continue;
        }
        const size_t gb = list.at(blk); // the block's index in src / sizes (links are per round)
        const uint8_t *g = src + gb * src_stride;
        uint16_t *out = links + blk * (size_t)n2;
        __syncthreads();
#pragma unroll 16
        for (uint32_t i = 0; i < kLzfTabBytes / 16 / 64; i++) reinterpret_cast<uint4 *>(smem)[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        bool bad = force_redo != 0;
        // the block passes through the staging area in pieces of kChainMax bytes (one piece for small blocks)
        for (uint32_t p0 = 0; p0 < n && !bad; p0 += kChainMax) {
            const uint32_t pn = n - p0 < kChainMax + 16 ? n - p0 : kChainMax + 16; // + the bytes the last positions hash
            if (p0) __syncthreads();
            if (vec) {
                CW_STAGE1(0) CW_STAGE1(1) CW_STAGE1(2) CW_STAGE1(3) CW_STAGE1(4) CW_STAGE1(5) CW_STAGE1(6) CW_STAGE1(7)
                CW_PREFETCH(blk + gridDim.x);
            } else if (((reinterpret_cast<uintptr_t>(g) | p0) & 15) == 0) {
                for (uint32_t i = lane; i < (pn + 15) / 16; i += 64)
                    if (i * 16 + 16 <= pn) reinterpret_cast<uint4 *>(stage)[i] = reinterpret_cast<const uint4 *>(g + p0)[i];
                    else for (uint32_t b = i * 16; b < pn; b++) stage[b] = g[p0 + b];
            } else {
                for (uint32_t i = lane; i < pn; i += 64) stage[i] = g[p0 + i];
            }
            if (lane < 16) stage[((pn + 15u) & ~15u) + lane] = 0; // slack read by the dword loads
            __syncthreads();

            const uint32_t pend = p0 + kChainMax < n ? p0 + kChainMax : n; // positions [p0, pend) belong to this piece
            // the dwords around the positions of a round are read one round ahead, so that a round has ONE LDS wait
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t pos = p0 + 64 * k + lane;
                const uint32_t *w = reinterpret_cast<const uint32_t *>(stage) + ((pos + 2 < n && pos < pend ? pos - p0 : 0u) >> 2);
                lo[k] = w[0]; hi[k] = w[1];
            }
            for (uint32_t base = p0; base < pend && base + 2 < n && !bad; base += 256) {
                uint32_t addr[4], mask[4], data[4], old[4], sh[4];
                bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t pos = base + 64 * k + lane;
                    ok[k] = pos + 2 < n && pos < pend;
                    const uint32_t v = __builtin_amdgcn_alignbyte(hi[k], lo[k], (ok[k] ? pos - p0 : 0u) & 3u);
                    const uint32_t slot = lzf_slot(v & 0xFFu, (v >> 8) & 0xFFu, (v >> 16) & 0xFFu);
                    sh[k] = (slot & 1u) * 16;
                    addr[k] = tab_lds + (slot >> 1) * 4;
                    mask[k] = ok[k] ? 0xFFFFu << sh[k] : 0u; // a lane without a position exchanges nothing
                    data[k] = ok[k] ? pos << sh[k] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t pos = base + 256 + 64 * k + lane;
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(stage) + ((pos + 2 < n && pos < pend ? pos - p0 : 0u) >> 2);
                    lo[k] = w[0]; hi[k] = w[1];
                }
                // four exchanges back to back: the LDS runs them in order, lanes ascending inside each
                asm volatile("ds_mskor_rtn_b32 %0, %4, %8, %12\n\t"
                             "ds_mskor_rtn_b32 %1, %5, %9, %13\n\t"
                             "ds_mskor_rtn_b32 %2, %6, %10, %14\n\t"
                             "ds_mskor_rtn_b32 %3, %7, %11, %15\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                             : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]),
                               "v"(data[0]), "v"(data[1]), "v"(data[2]), "v"(data[3])
                             : "memory");
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t pos = base + 64 * k + lane;
                    const uint32_t o = ok[k] ? (old[k] >> sh[k]) & 0xFFFFu : 0u;
                    if (__ballot(ok[k] && o >= pos && (o | pos) != 0)) bad = true; // not in lane order
                    if (pos < pend || (pend == n && pos < n2)) out[pos] = (uint16_t)o;
                }
            }
        }
        if (lane == 0) sizes[gb] = bad ? kRedo : 0u; // 0 = "links are valid" for lzf_chain_kernel
    }
#undef CW_PREFETCH
#undef CW_PRE1
#undef CW_STAGE1
}

// BIG = false: links + flags and the block in LDS (blocks <= 16 KiB).  BIG = true: the links stay in global memory, the
// block is read from global memory, only the skip flags (one bit per position) are in LDS -- 8 KiB for a 64 KiB block,
// so 16 blocks fit a CU where the table-based kernel fits one; a chain step then costs a global round trip.
template <bool BIG>
__global__ void __launch_bounds__(64)
lzf_chain_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *__restrict__ dst,
                 size_t dst_stride, uint32_t *__restrict__ sizes, uint16_t *__restrict__ links, uint32_t n2,
                 uint32_t *__restrict__ counter, LaneShare share, BlockList list)
{
    if (list.queue) {
        const uint32_t cnt = *list.count;
        if (list.first >= cnt) return;
        if (nblocks > cnt - list.first) nblocks = cnt - list.first;
    }
    uint32_t taken = 0;
    if (share.ctr) { // (the round is claimed: its links kernel ran and published)
        taken = share_round_taken(share, nblocks, false);
        if (taken == kShareGaveUp || share.round_first >= share.total - taken) return;
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *E = reinterpret_cast<uint16_t *>(smem);          // !BIG: link | kSkipFlag, n2 entries
    uint8_t *stage = smem + 2 * (size_t)n2;                     // !BIG: the block, + 16 bytes of slack
    uint32_t *skipmap = reinterpret_cast<uint32_t *>(smem);     // BIG: one bit per position
    __shared__ uint32_t mailbox;
    const uint32_t lane = threadIdx.x;
    const uint32_t cap = n - 1;

    for (;;) {
        // blocks are handed out dynamically: their parse times differ by an order of magnitude
        __syncthreads();
        if (lane == 0) mailbox = atomicAdd(counter, 1u);
        __syncthreads();
        const size_t blk = __builtin_amdgcn_readfirstlane(mailbox);
        if (blk >= nblocks) break;
        if (lanes_took(share, taken, blk)) continue;                       // a lane parses (or parsed) it
        const size_t gb = list.at(blk);
        if (__builtin_amdgcn_readfirstlane(sizes[gb]) == kRedo) continue; // links not valid: lzf_blocks_kernel parses it
        const uint8_t *g = src + gb * src_stride;
        uint8_t *out = dst + gb * dst_stride;
        uint16_t *lk = links + blk * (size_t)n2;
        if (BIG) {
            for (uint32_t i = lane; i < n2 / 32; i += 64) skipmap[i] = 0;
        } else {
            const uint4 *l4 = reinterpret_cast<const uint4 *>(lk);
            for (uint32_t i = lane; i < n2 / 8; i += 64) reinterpret_cast<uint4 *>(E)[i] = l4[i];
            for (uint32_t i = lane; i < n; i += 64) stage[i] = g[i];
            if (lane < 16) stage[n + lane] = 0;
        }
        __syncthreads();
        const uint8_t *in = BIG ? g : stage;

        uint32_t ip = 0, op = 1, lit = 0;
        bool fail = (n == 0 || cap == 0);
        auto request = [&](uint32_t ip_) __attribute__((always_inline)) -> uint32_t {
            const uint32_t pos = ip_ + lane;
            if (!BIG) return lz::rd32x<true>(in, pos + 2 < n ? pos : 0u);
            const uint32_t p = pos + 2 < n ? pos : 0u;
            const uint32_t q = p + 4 <= n ? p : p - 1; // pos = n - 3: read one byte earlier and shift (no read past the block)
            return lz::rd32(in, q) >> ((p - q) * 8);
        };
        // BIG: the link of every position of the next batch is requested together with its bytes (one round trip less)
        auto request_link = [&](uint32_t ip_) __attribute__((always_inline)) -> uint32_t {
            const uint32_t pos = ip_ + lane;
            return BIG ? (uint32_t)lk[pos + 2 < n ? pos : 0u] : 0u;
        };
        uint32_t vnext = fail ? 0u : request(0), lnext = fail ? 0u : request_link(0);
        uint32_t head = kChainHead;

        while (!fail && ip + 2 < n) {
            const uint32_t pos = ip + lane;
            const bool tested = lane < head && pos + 2 < n;
            const uint32_t ntest = (uint32_t)__builtin_popcountll(__ballot(tested));
            const uint32_t v = vnext;
            // the reference the serial parser would read: first position on the link chain that was inserted
            // A skipped position stays skipped and an inserted one inserted (the parser never goes back), so the answer for
            // a skipped position never changes: the first skipped position of a walk gets the walk's result as its link
            // (path compression).  Without it periodic data -- a spreadsheet's records, a bitmap's rows: every period holds a
            // same-slot position inside a long match -- walks back period by period through the whole block at every
            // lookup, one dependent memory round trip per step for BIG blocks (kennedy.xls at 64 KiB: 0.1 GB/s).  A later
            // walk enters at the previous walker's position and is home in three steps.  A reader that still sees the old
            // link only walks the longer, equally valid path.
            uint32_t cur, first_skipped = 0;
            if (BIG) {
                cur = tested ? lnext : 0u;
                for (;;) {
                    const bool skipped = cur && ((skipmap[cur >> 5] >> (cur & 31u)) & 1u);
                    if (skipped) {
                        if (!first_skipped) first_skipped = cur;
                        cur = lk[cur];
                    }
                    if (!__ballot(skipped)) break;
                }
                if (first_skipped) lk[first_skipped] = (uint16_t)cur;
            } else {
                cur = tested ? E[pos] & 0x7FFFu : 0u;
                for (;;) {
                    const uint32_t e = cur ? E[cur] : 0u;
                    const bool skipped = (e & kSkipFlag) != 0;
                    if (skipped) {
                        if (!first_skipped) first_skipped = cur;
                        cur = e & 0x7FFFu;
                    }
                    if (!__ballot(skipped)) break;
                }
                if (first_skipped) E[first_skipped] = (uint16_t)(kSkipFlag | cur);
            }
            const uint32_t old = cur;
            const bool cand = tested && old > 0 && pos - old - 1 < kMaxOff;
            lz::Around ap, ac;
            ap.before = 0; ap.at = 0; ap.after = 0; ac.before = 0; ac.at = 1u << 24; ac.after = 1;
            const bool wide = !BIG || pos + 12 <= n; // all 16 bytes readable (reference < position)
            if (cand) {
                if (wide) {
                    ac = lz::around<!BIG>(in, old, false);
                    ap = lz::around<!BIG>(in, pos, false);
                } else {
                    ac.at = in[old] | (in[old + 1] << 8) | (in[old + 2] << 16) | (~v & 0xFF000000u); // 4th byte: "differs"
                }
            }
            const unsigned long long mm = __ballot(cand && ((ac.at ^ v) & 0xFFFFFFu) == 0);
            const uint32_t w = mm ? (uint32_t)__builtin_ctzll(mm) : 64u;

            const uint32_t nlit = mm ? w : ntest;
            uint32_t ref = 0, eqs = 0;
            bool more_eq = false;
            if (mm) {
                const uint64_t x = ap.after ^ ac.after;
                uint32_t e = (ac.at ^ v) >> 24 ? 0u : 1u + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u);
                bool me = e == 9;
                if (!wide) { e = 0; me = true; } // nothing was compared beyond the 3 bytes
                const uint32_t packed = __builtin_amdgcn_readlane(e | ((uint32_t)me << 4), w);
                eqs = packed & 0xFu; more_eq = (packed >> 4) & 1u;
                ref = __builtin_amdgcn_readlane(old, w);
            }
            if (nlit) {
                const uint32_t last = op + (nlit - 1) + (lit + nlit - 1) / kMaxLit;
                if (last >= cap) { fail = true; break; }
                if (lane < nlit) { // literal i = low byte of lane i
                    const uint32_t t = lit + lane, p = op + lane + t / kMaxLit;
                    out[p] = (uint8_t)v;
                    if ((t + 1) % kMaxLit == 0) out[p - kMaxLit] = kMaxLit - 1;
                }
                op += nlit + (lit + nlit) / kMaxLit;
                lit = (lit + nlit) % kMaxLit;
                ip += nlit;
            }
            if (!mm) {
                if (ip + 2 < n) { vnext = request(ip); lnext = request_link(ip); }
                head = head < 64 ? head * 2 : 64;
                continue;
            }
            head = kChainHead;

            // ---- match at ip against ref ----
            uint32_t maxlen = n - ip - 2;
            if (maxlen > kMaxRef) maxlen = kMaxRef;
            if (op + 4 >= cap && op - (lit == 0) + 4 >= cap) { fail = true; break; }
            if (lit) { if (lane == 0) out[op - lit - 1] = (uint8_t)(lit - 1); }
            else op -= 1;
            uint32_t eq = eqs;
            {
                const uint32_t room = (n - ip < kMaxRef + 2 ? n - ip : kMaxRef + 2) - 3;
                if (eq >= room) { eq = room; more_eq = false; }
            }
            while (more_eq) {
                const uint32_t t = 3 + eq + lane;
                const bool ok = ip + t < n && t < kMaxRef + 2 && in[ref + t] == in[ip + t];
                const uint32_t cnt = ctz64(~__ballot(ok));
                eq += cnt;
                if (cnt < 64) break;
            }
            uint32_t len;
            if (maxlen > 16) {
                if (eq < 16) len = 3 + eq;
                else { len = 3 + eq < maxlen ? 3 + eq : maxlen; if (len < 19) len = 19; }
            } else {
                len = 3 + eq < maxlen ? 3 + eq : maxlen;
                if (len < 3) len = 3;
            }
            const uint32_t off = ip - ref - 1, l2 = len - 2, at = op;
            op += l2 < 7 ? 2 : 3;
            lit = 0; op += 1;
            // the inside of the match is never inserted (VERY_FAST re-inserts only its last two positions)
            for (uint32_t i = lane; i + 3 < len; i += 64) {
                const uint32_t q = ip + 1 + i;
                if (BIG) atomicOr(&skipmap[q >> 5], 1u << (q & 31u));
                else atomicOr(reinterpret_cast<uint32_t *>(E) + (q >> 1), kSkipFlag << ((q & 1u) * 16));
            }
            ip += len;
            const bool go_on = ip + 2 < n;
            if (go_on) { vnext = request(ip); lnext = request_link(ip); }
            if (lane == 0) {
                if (l2 < 7) {
                    out[at] = (uint8_t)((off >> 8) + (l2 << 5));
                    out[at + 1] = (uint8_t)off;
                } else {
                    out[at] = (uint8_t)((off >> 8) + (7u << 5));
                    out[at + 1] = (uint8_t)(l2 - 7);
                    out[at + 2] = (uint8_t)off;
                }
            }
            if (!go_on) break;
        }

        if (!fail) {
            if (op + 3 > cap) {
                fail = true;
            } else {
                if (ip < n) put_literals(out, in, ip, n - ip, op, lit, lane);
                if (lit) { if (lane == 0) out[op - lit - 1] = (uint8_t)(lit - 1); }
                else op -= 1;
            }
        }
        if (lane == 0) sizes[gb] = fail ? 0u : op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Scalar-thread chain parser (round 3; blocks > 4 KiB): lzf_chain_kernel<true>'s algorithm -- links from lzf_links_kernel, one skip bit per
// position in LDS, no table -- with the wavefront run as ONE scalar thread, like the LZ4 parsers of lz4_vtab_kernel.hip.
// The wavefront-wide chain kernel speculates on 8..64 positions per step, but on compressible data a match ends the step after two
// or three of them, and each step is a chain of VECTOR memory round trips (link, skip word, candidate bytes: 56 ms per 8 Ki blocks of
// text, 9.5 GB/s).  Here a position is ~20 scalar instructions, and what it needs from memory comes in ONE round trip through the
// scalar cache and the LDS together: the skip word of the candidate, the candidate's own link (the next step of the chain, should
// it turn out to be skipped) and the 32 bytes around it -- plus, every fourth position, the next 32 bytes of the block and the next four
// links.  A position whose link is empty or out of reach costs no memory access at all.  Sequences are recorded and written out 64 at a
// time, a lane per output byte (as lz4_vtab_kernel.hip's emit_batch); the parser's overflow checks run on a scalar copy of liblzf's
// (op, lit) arithmetic, so "returns 0" is exact.  8 KiB of LDS per 64 KiB block: 20 single-wavefront workgroups per CU.
// ---------------------------------------------------------------------------------------------------
namespace {
using namespace st;
constexpr uint32_t kNoMatch = 0xFFFFFFFFu; // rec_m of the block's last literals

// bytes of a literal run of L bytes in the stream (a control byte per 32), of a match with l2 = length - 2
__device__ __forceinline__ uint32_t lzf_lit_bytes(uint32_t L) { return L + ((L + 31u) >> 5); }

// writes sequences 0 .. nrec-1 of the batch (lane i: literals [a, a + L), then the match m = l2 | off << 16, or kNoMatch) at out + op0;
// returns the output position behind them
__device__ __forceinline__ uint32_t lzf_emit_batch(uint8_t *__restrict__ out, const uint8_t *__restrict__ g, uint32_t op0, uint32_t nrec, uint32_t rec_a,
                                                   uint32_t rec_L, uint32_t rec_m, uint32_t lane)
{
    const bool live = lane < nrec;
    const uint32_t L = live ? rec_L : 0u;
    const uint32_t mb = !live || rec_m == kNoMatch ? 0u : (rec_m & 0xFFFFu) < 7u ? 2u : 3u;
    const uint32_t full = lzf_lit_bytes(L) + mb;
    const uint32_t fend = wave_scan_add(full), fstart = fend - full;
    const uint32_t total = __builtin_amdgcn_readlane(fend, 63);
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t j = base + lane;
        uint32_t sq = 0; // number of sequences that end at or before byte j = the sequence of byte j
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1)
            if (lane_get(fend, sq + step - 1u) <= j) sq += step;
        const uint32_t q_fstart = lane_get(fstart, sq), q_a = lane_get(rec_a, sq), q_L = lane_get(L, sq), q_m = lane_get(rec_m, sq);
        const uint32_t r = j - q_fstart, lb = lzf_lit_bytes(q_L);
        if (j < total) {
            uint32_t byte;
            if (r < lb) { // run c of the literals: a control byte, then up to 32 bytes
                const uint32_t c = r / 33u, k = r - 33u * c;
                byte = k == 0 ? min(32u, q_L - 32u * c) - 1u : (uint32_t)g[q_a + 32u * c + k - 1u];
            } else {
                const uint32_t r2 = r - lb, l2 = q_m & 0xFFFFu, off = q_m >> 16;
                if (l2 < 7) byte = r2 == 0 ? (off >> 8) + (l2 << 5) : off;
                else byte = r2 == 0 ? (off >> 8) + (7u << 5) : r2 == 1 ? l2 - 7u : off;
            }
            out[op0 + j] = (uint8_t)byte;
        }
    }
    return op0 + total;
}

// One step of a position's chain, ONE wait: the skip word of `cur` (LDS), the dword of links that holds cur's own link, the 32 bytes
// from cur & ~3 on; NEXT: also the block's next 32 bytes and next four links (the position in hand is the last of its dword).
// (Loads and wait in one statement: scalar_thread.h.)
template <bool NEXT>
__device__ __forceinline__ void lzf_step_loads(uint32_t skip_addr, const u32x4 &rs, const u32x4 &rl, uint32_t coff, uint32_t loff, uint32_t noff,
                                               uint32_t nloff, uint32_t &skipword, uint32_t &lnk, u32x8 &wc, u32x8 &wq, u32x2 &lq)
{
    uint32_t sv;
    if constexpr (NEXT)
        asm volatile("ds_read_b32 %[sv], %[sa]\n\t"
                     "s_buffer_load_dword %[lnk], %[rl], %[loff]\n\t"
                     "s_buffer_load_dwordx8 %[wc], %[rs], %[coff]\n\t"
                     "s_buffer_load_dwordx8 %[wq], %[rs], %[noff]\n\t"
                     "s_buffer_load_dwordx2 %[lq], %[rl], %[nloff]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %[sw], %[sv]"
                     : [sv] "=&v"(sv), [lnk] "=&s"(lnk), [wc] "=&s"(wc), [wq] "=&s"(wq), [lq] "=&s"(lq), [sw] "=&s"(skipword)
                     : [sa] "v"(skip_addr), [rs] "s"(rs), [rl] "s"(rl), [coff] "s"(coff), [loff] "s"(loff), [noff] "s"(noff), [nloff] "s"(nloff)
                     : "memory");
    else
        asm volatile("ds_read_b32 %[sv], %[sa]\n\t"
                     "s_buffer_load_dword %[lnk], %[rl], %[loff]\n\t"
                     "s_buffer_load_dwordx8 %[wc], %[rs], %[coff]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %[sw], %[sv]"
                     : [sv] "=&v"(sv), [lnk] "=&s"(lnk), [wc] "=&s"(wc), [sw] "=&s"(skipword)
                     : [sa] "v"(skip_addr), [rs] "s"(rs), [rl] "s"(rl), [coff] "s"(coff), [loff] "s"(loff)
                     : "memory");
}
// the block's 32 bytes from byte offset noff on and the four links from byte offset nloff on
__device__ __forceinline__ void lzf_next_loads(const u32x4 &rs, const u32x4 &rl, uint32_t noff, uint32_t nloff, u32x8 &wq, u32x2 &lq)
{
    asm volatile("s_buffer_load_dwordx8 %[wq], %[rs], %[noff]\n\t"
                 "s_buffer_load_dwordx2 %[lq], %[rl], %[nloff]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [wq] "=&s"(wq), [lq] "=&s"(lq) : [rs] "s"(rs), [rl] "s"(rl), [noff] "s"(noff), [nloff] "s"(nloff));
}
} // namespace

__global__ void __launch_bounds__(64)
lzf_sthread_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *__restrict__ dst, size_t dst_stride,
                   uint32_t *__restrict__ sizes, uint16_t *__restrict__ links, uint32_t n2, uint32_t *__restrict__ counter, LaneShare share,
                   BlockList list)
{
    if (list.queue) {
        const uint32_t cnt = *list.count;
        if (list.first >= cnt) return;
        if (nblocks > cnt - list.first) nblocks = cnt - list.first;
    }
    uint32_t taken = 0;
    if (share.ctr) { // (the round is claimed: its links kernel ran and published)
        taken = share_round_taken(share, nblocks, false);
        if (taken == kShareGaveUp || share.round_first >= share.total - taken) return;
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *skipmap = reinterpret_cast<uint32_t *>(smem); // one bit per position
    const uint32_t map_lds = opaque_s((uint32_t)reinterpret_cast<uintptr_t>(smem));
    const uint32_t lane = threadIdx.x;
    const uint32_t cap = n - 1;

    for (;;) {
        // blocks are handed out dynamically: their parse times differ by an order of magnitude
        uint32_t drawn = 0;
        if (lane == 0) drawn = atomicAdd(counter, 1u);
        const size_t blk = __builtin_amdgcn_readfirstlane(drawn);
        if (blk >= nblocks) break;
        if (lanes_took(share, taken, blk)) continue;                       // a lane parses (or parsed) it
        const size_t gb = list.at(blk);
        if (__builtin_amdgcn_readfirstlane(sizes[gb]) == kRedo) continue; // links not valid: lzf_blocks_kernel parses it
        const uint8_t *g = src + gb * src_stride;
        uint8_t *out = dst + gb * dst_stride;
        uint16_t *lk = links + blk * (size_t)n2;
        const u32x4 rs = sc_descriptor(g, n), rl = sc_descriptor(lk, n2 * 2u);
        for (uint32_t i = lane; i < n2 / 32; i += 64) skipmap[i] = 0;
        asm volatile("" ::: "memory");

        // the parse: liblzf's loop (oracle/lzf_oracle.c), every value wave-uniform.  (op, lit): liblzf's output position and open literal
        // run, kept for its overflow checks only; the bytes are written by lzf_emit_batch from the records
        uint32_t ip = 0, anchor = 0, op = 1, lit = 0;
        uint32_t rec_a = 0, rec_L = 0, rec_m = 0, nrec = 0, s_op = 0;
        bool fail = false;
        u32x8 wp;   // the 32 bytes from ip & ~3 on
        u32x2 lwin; // the links of positions ip & ~3 .. + 3
        lzf_next_loads(rs, rl, 0, 0, wp, lwin);
        while (ip + 2 < n) {
            const uint32_t b = ip & 3u;
            const uint32_t v3 = cut32(wp[0], wp[1], b * 8u) & 0xFFFFFFu;
            uint32_t cur = (uint32_t)((((uint64_t)lwin[1] << 32) | lwin[0]) >> (b * 16u)) & 0xFFFFu;
            const bool cross = b == 3u; // the next position starts a new dword of the block and a new group of four links
            const uint32_t nip = ip + 1u;
            u32x8 wq, wc;
            u32x2 lq;
            bool have_next = false, match = false;
            if (cur != 0 && ip - cur - 1u < kMaxOff) {
                // the reference the serial parser would read: the first position on the link chain that was inserted (not skipped as the
                // inside of a match), as long as it is within reach -- a chain only moves further away.  Path compression as in
                // lzf_chain_kernel: the first skipped position of a walk gets the walk's end as its link.
                uint32_t first_skipped = 0;
                for (;;) {
                    uint32_t skipword, lnk;
                    const uint32_t sa = to_v(map_lds + (cur >> 5) * 4u);
                    if (cross && !have_next) {
                        lzf_step_loads<true>(sa, rs, rl, cur & ~3u, (cur * 2u) & ~3u, nip, nip * 2u, skipword, lnk, wc, wq, lq);
                        have_next = true;
                    } else {
                        lzf_step_loads<false>(sa, rs, rl, cur & ~3u, (cur * 2u) & ~3u, 0, 0, skipword, lnk, wc, wq, lq);
                    }
                    if (!((skipword >> (cur & 31u)) & 1u)) {
                        match = (cut32(wc[0], wc[1], (cur & 3u) * 8u) & 0xFFFFFFu) == v3;
                        break;
                    }
                    if (!first_skipped) first_skipped = cur;
                    cur = (lnk >> ((cur & 1u) * 16u)) & 0xFFFFu;
                    if (cur == 0 || ip - cur - 1u >= kMaxOff) break; // the chain has ended or left the window
                }
                if (first_skipped && lane == 0) lk[first_skipped] = (uint16_t)cur;
            }
            if (match) {
                // ---- length: equal bytes from the two windows (24 of them), then 64 at a time ----
                const uint32_t psh = b * 8u, csh = (cur & 3u) * 8u;
                uint32_t plo, phi, clo, chi;
                cut64_v(wp[0], wp[1], wp[2], psh, plo, phi);
                cut64_v(wc[0], wc[1], wc[2], csh, clo, chi);
                uint32_t T = equal_bytes_v(plo, phi, clo, chi); // >= 3
                if (__builtin_amdgcn_readfirstlane((uint32_t)(T == 8))) {
                    cut64_v(wp[2], wp[3], wp[4], psh, plo, phi);
                    cut64_v(wc[2], wc[3], wc[4], csh, clo, chi);
                    T = 8u + equal_bytes_v(plo, phi, clo, chi);
                    if (__builtin_amdgcn_readfirstlane((uint32_t)(T == 16))) {
                        cut64_v(wp[4], wp[5], wp[6], psh, plo, phi);
                        cut64_v(wc[4], wc[5], wc[6], csh, clo, chi);
                        T = 16u + equal_bytes_v(plo, phi, clo, chi);
                        if (__builtin_amdgcn_readfirstlane((uint32_t)(T == 24))) {
                            uint32_t t = 24;
                            for (;;) {
                                const uint32_t i = t + lane;
                                const bool ok = ip + i < n && i < kMaxRef + 2 && g[cur + i] == g[ip + i];
                                const uint32_t cnt = ctz64(~__ballot(ok));
                                t += cnt;
                                if (cnt < 64) break;
                            }
                            T = t;
                        }
                    }
                }
                uint32_t eq = __builtin_amdgcn_readfirstlane(T) - 3u;
                {
                    const uint32_t room = (n - ip < kMaxRef + 2 ? n - ip : kMaxRef + 2) - 3u; // (windows read zeros beyond the block)
                    if (eq > room) eq = room;
                }
                uint32_t maxlen = n - ip - 2u, len;
                if (maxlen > kMaxRef) maxlen = kMaxRef;
                if (maxlen > 16) { // liblzf's sixteen unrolled, unbounded compares
                    if (eq < 16) len = 3u + eq;
                    else { len = 3u + eq < maxlen ? 3u + eq : maxlen; if (len < 19) len = 19; }
                } else {
                    len = 3u + eq < maxlen ? 3u + eq : maxlen;
                    if (len < 3) len = 3;
                }
                if (op - (lit == 0) + 4u >= cap) { fail = true; break; }
                if (lit == 0) op -= 1;
                const uint32_t l2 = len - 2u, off = ip - cur - 1u;
                op += l2 < 7 ? 2u : 3u;
                lit = 0; op += 1;
                {
                    const bool mine = lane == nrec;
                    rec_a = mine ? anchor : rec_a;
                    rec_L = mine ? ip - anchor : rec_L;
                    rec_m = mine ? l2 | (off << 16) : rec_m;
                    if (++nrec == 64) { s_op = lzf_emit_batch(out, g, s_op, 64, rec_a, rec_L, rec_m, lane); nrec = 0; }
                }
                // the inside of the match is never inserted (VERY_FAST re-inserts only its last two positions): positions [ip + 1, ip + len - 2)
                if (len > 3) {
                    const uint32_t first = ip + 1u, last = ip + len - 2u; // [first, last): up to 261 bits, ten words at most
                    const uint32_t w = (first >> 5) + lane;
                    const uint32_t lo = max(first, w * 32u), hi = min(last, w * 32u + 32u);
                    if (lane < 10 && lo < hi) {
                        const uint32_t bits = hi - lo;
                        atomicOr(&skipmap[w], (bits == 32 ? 0xFFFFFFFFu : (1u << bits) - 1u) << (lo & 31u));
                    }
                }
                asm volatile("" ::: "memory");
                ip += len;
                anchor = ip;
                if (ip + 2 >= n) break;
                lzf_next_loads(rs, rl, ip & ~3u, (ip * 2u) & ~7u, wp, lwin);
                continue;
            }
            // ---- a literal ----
            if (op >= cap) { fail = true; break; }
            lit += 1; op += 1;
            if (lit == kMaxLit) { lit = 0; op += 1; }
            if (cross) {
                if (!have_next) lzf_next_loads(rs, rl, nip, nip * 2u, wq, lq);
                wp = wq; lwin = lq;
            }
            ip = nip;
        }
        if (!fail && op + 3 > cap) fail = true;
        if (!fail) {
            if (anchor < n) {
                const bool mine = lane == nrec;
                rec_a = mine ? anchor : rec_a;
                rec_L = mine ? n - anchor : rec_L;
                rec_m = mine ? kNoMatch : rec_m;
                nrec += 1;
            }
            if (nrec) s_op = lzf_emit_batch(out, g, s_op, nrec, rec_a, rec_L, rec_m, lane);
        }
        if (lane == 0) sizes[gb] = fail ? 0u : s_op;
    }
}

// ---------------------------------------------------------------------------------------------------
// Lane-per-block parser for large batches of blocks that do not fit the LDS-resident scheme (> 4 KiB): the counterpart
// of lz4_lanes_kernel (lz4_kernel.hip has the reasoning).  A lane runs liblzf's loop as it stands -- one position per
// iteration: hash the next three bytes, exchange the table slot, test the reference, emit a literal or a match -- with its
// 65,536 x u16 table in global memory (128 KiB per lane, zeroed by the lane when it takes a block).  No links, no skip
// flags, no lane-order assumption: the parse is the serial one.
// ---------------------------------------------------------------------------------------------------
// A lane needs ~66 ms for a 64 KiB block of text however few lanes there are, the link/chain kernels run at 8.5 GB/s: the lanes win
// from ~9 Ki compressible blocks on (text, 64 KiB, 12 Ki / 16 Ki / 20 Ki / 24 Ki blocks: 13.2 / 16.3 / 13.6 / 15.2 GB/s against 8.5; the dip
// is the second wavefront on some CUs' SIMDs).  A batch that is a third noise or more goes back to the chain kernels anyway.
constexpr uint32_t kLzfLaneMinBlocks = 12288;
constexpr uint32_t kLzfLaneMinSmall = 28672;  // blocks <= 4 KiB: lanes beside the rounds from 28 Ki blocks on (text, 32 Ki / 40 Ki / 64 Ki blocks: 16.1 / 19.5 / 21 against 13.5 GB/s)
constexpr size_t kLzfBesideRound = 16384;     // ... in rounds of 16 Ki blocks, the last 16 Ki unclaimed blocks left to the rounds.  4 KiB blocks, text /
                                              // 50 % noise / noise, GB/s -- 1 Mi blocks: rounds of 8 Ki 29.8 / 24.7 / 52.9, 16 Ki 33.1 / 29.0 / 57.7,
                                              // 32 Ki 32.6 / 30.9 / 59.2; 96 Ki blocks: 25.6 / 36.3 / 50.5, 26.2 / 39.3 / 53.6, 23.1 / 36.4 / 55.0
// blocks > 16 KiB: lanes beside the scalar-thread rounds from 52 Ki blocks on; from 96 Ki blocks on rounds of 8 Ki blocks and as many left to the rounds, below
// (every lane gets one block; the rounds get what they manage in that time) rounds of 4 Ki and 12 Ki blocks left.  Corpus, 64 KiB, lanes alone -> beside:
// 48 Ki blocks 27.0 -> 26.3-27.6 (not used), 56 Ki 27.0 -> 29.1, 64 Ki 27.3 -> 29.0, 80 Ki 26.4 -> 34.8, 128 Ki 27.3 -> 31.0, 256 Ki 29.4 -> 35.2 GB/s
constexpr size_t kLzfBigBesideMin = 53248, kLzfBigBesideWide = 98304, kLzfBigBesideRound = 8192, kLzfBigBesideRoundMid = 4096, kLzfBigBesideReserveMid = 12288;
constexpr uint32_t kLzfBesideReserve = 16384, kLzfBesideReserveFew = 8192; // blocks left to the rounds; below 48 Ki blocks (32 Ki blocks: 16.1 against 13.2 GB/s with 16 Ki)

// 4 bytes at ip (ip + 2 < n): the last position of a block is read one byte early and shifted (no read past the block)
__device__ __forceinline__ uint32_t lzf_rd(const uint8_t *g, uint32_t ip, uint32_t n)
{
    const uint32_t q = ip + 4 <= n ? ip : n - 4;
    return lz::rd32(g, q) >> ((ip - q) * 8);
}

// TAGGED (blocks <= 4 KiB): a table entry is epoch:4 | position:12, an entry of another epoch reads as empty, and the 128 KiB
// table is zeroed once per 15 blocks instead of per block -- a 4 KiB block is ~1,300 parse iterations, zeroing its table 8,192
// store iterations that the whole wavefront sits through whenever one of its lanes takes a new block.
template <bool TAGGED>
__global__ void __launch_bounds__(64)
lzf_lanes_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *__restrict__ dst, size_t dst_stride,
                 uint32_t *__restrict__ sizes, uint16_t *__restrict__ tables, uint32_t *__restrict__ counter, uint32_t reserve,
                 uint32_t *__restrict__ handback)
{
    // Blocks that do not compress are the lanes' worst case and the chain parser's best (a noise block is a serial walk of all
    // its positions here -- 61 -> 18 GB/s on random 64 KiB blocks when the lanes took them all -- and a 60 GB/s bail-out there).
    // So a lane looks at its block once, after `check_at` positions: less than 1/32 saved so far => the block goes to the
    // hand-back list (counter[kCtrHanded] entries), which the link/chain kernels parse after the lanes are done; and once more than a
    // third of the blocks looked at were such (counter[kCtrPoor], counter[kCtrFine]), the lanes stop parsing: beside the rounds they
    // take no more blocks, on their own they pass what is left straight to the list.
    const uint32_t check_at = n >= 2048 ? 512u : n / 4 < 128 ? 128u : n / 4;
    bool checked = false;
    // Slow start: the first 128 workgroups sample the batch; the others wait (bounded: ~3 ms) until 512 blocks have been looked at
    // and then start -- or, on noise, leave at once.  (Random 4 KiB blocks beside the rounds: 65,536 first looks cost 11 ms.)
    if (blockIdx.x >= 128) {
        for (uint32_t spin = 0; spin < 4096; spin++) {
            const uint32_t seen = __hip_atomic_load(&counter[kCtrPoor], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                                  __hip_atomic_load(&counter[kCtrFine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= 512) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
    uint32_t epoch = 15; // TAGGED: forces a clean table before the first block
    auto tab_get = [&](uint16_t *t, uint32_t slot) -> uint32_t {
        const uint32_t e = t[slot];
        return TAGGED ? ((e >> 12) == epoch ? e & 0xFFFu : 0u) : e;
    };
    auto tab_put = [&](uint16_t *t, uint32_t slot, uint32_t pos) { t[slot] = (uint16_t)(TAGGED ? (epoch << 12) | pos : pos); };
    // reserve == 0: the lanes take every block, pulled upwards from counter[0].  reserve > 0: beside the link/chain rounds --
    // blocks are taken from the top downwards while more than `reserve` unclaimed blocks are left (LaneShare)
    uint16_t *tab = tables + ((size_t)blockIdx.x * 64 + threadIdx.x) * kLzfSlots;
    const uint32_t cap = n - 1; // out_len of the reference's call (n >= 16 here)
    enum : uint32_t { NEXT = 0, STEP = 1, TAIL = 2, EXIT = 3 };
    uint32_t state = NEXT, ip = 0, op = 0, lit = 0, v = 0;
    size_t blk = 0;
    const uint8_t *g = src;
    uint8_t *out = dst;
    bool fail = false;

    while (__ballot(state != EXIT)) {
        if (state == NEXT) {
            const uint32_t poor = __hip_atomic_load(&counter[kCtrPoor], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t fine = __hip_atomic_load(&counter[kCtrFine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (beside the rounds the lanes also slow THEM down, so they give up sooner: a fifth of noise against a third)
            const bool noisy = reserve ? poor > 64 + fine / 4 : poor > 32 + fine / 2;
            if (reserve && noisy) {
                blk = nblocks; // the rounds take the rest
            } else if (reserve) { // the wavefront's idle lanes ask together (LaneShare): one compare-and-swap for all of them
                const unsigned long long idle = __ballot(true);
                const uint32_t m = (uint32_t)__builtin_popcountll(idle), leader = (uint32_t)__builtin_ctzll(idle);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                uint32_t first = 0, claimed = 0;
                if (threadIdx.x == leader) share_take(counter, m, nblocks, reserve, first, claimed);
                first = __builtin_amdgcn_readlane(first, leader);
                claimed = __builtin_amdgcn_readlane(claimed, leader);
                const size_t k = (size_t)first + rank; // my rank, if any: block nblocks-1-k, mine if it lay above `claimed` at the draw
                blk = k < nblocks && nblocks - 1 - k >= claimed ? nblocks - 1 - k : nblocks;
            } else {
                blk = atomicAdd(counter, 1u);
                while (noisy && blk < nblocks) { // unparsed, to the list
                    handback[atomicAdd(&counter[kCtrHanded], 1u)] = (uint32_t)blk;
                    blk = atomicAdd(counter, 1u);
                }
            }
            if (blk >= nblocks) {
                state = EXIT;
            } else {
                checked = false;
                g = src + blk * src_stride;
                out = dst + blk * dst_stride;
                if (!TAGGED || ++epoch == 16) {
                    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
                    for (uint32_t i = 0; i < kLzfTabBytes / 16; i++) t4[i] = make_uint4(0, 0, 0, 0);
                    epoch = 1;
                }
                ip = 0; op = 1; lit = 0; fail = false; // op = 1: the first literal run's control byte is reserved
                v = lzf_rd(g, 0, n);
                state = STEP;
            }
        }

        if (state == STEP && !checked && ip >= check_at) {
            checked = true;
            if (op + (ip >> 5) >= ip) { // (what the lane wrote so far is overwritten by the chain parser)
                atomicAdd(&counter[kCtrPoor], 1u);
                handback[atomicAdd(&counter[kCtrHanded], 1u)] = (uint32_t)blk;
                state = NEXT;
            } else {
                atomicAdd(&counter[kCtrFine], 1u);
            }
        }

        if (state == STEP) {
            // v = the 4 bytes at ip (requested an iteration ago)
            const uint32_t b0 = v & 0xFFu, b1 = (v >> 8) & 0xFFu, b2 = (v >> 16) & 0xFFu;
            const uint32_t slot = lzf_slot(b0, b1, b2);
            const uint32_t ref = tab_get(tab, slot);
            tab_put(tab, slot, ip);
            bool is_match = false;
            if (ref > 0 && ip - ref - 1 < kMaxOff) is_match = ((lz::rd32(g, ref) ^ v) & 0xFFFFFFu) == 0; // ref + 4 <= ip + 3 <= n
            if (is_match) {
                uint32_t maxlen = n - ip - 2;
                if (maxlen > kMaxRef) maxlen = kMaxRef;
                if (op + 4 >= cap && op - (lit == 0) + 4 >= cap) {
                    fail = true; state = TAIL;
                } else {
                    if (lit) out[op - lit - 1] = (uint8_t)(lit - 1);
                    else op -= 1;
                    // equal bytes from index 3 on, as far as the reference's loops can look
                    const uint32_t room = (n - ip < kMaxRef + 2 ? n - ip : kMaxRef + 2) - 3;
                    uint32_t eq = 0;
                    while (eq + 8 <= room) {
                        uint64_t x, y;
                        __builtin_memcpy(&x, g + ref + 3 + eq, 8);
                        __builtin_memcpy(&y, g + ip + 3 + eq, 8);
                        const uint64_t d = x ^ y;
                        if (d) { eq += (uint32_t)__builtin_ctzll(d) >> 3; break; }
                        eq += 8;
                    }
                    if (eq + 8 > room) while (eq < room && g[ref + 3 + eq] == g[ip + 3 + eq]) eq++;
                    uint32_t len;
                    if (maxlen > 16) { // 16 unrolled compares without a bound, then the bounded loop
                        if (eq < 16) len = 3 + eq;
                        else { len = 3 + eq < maxlen ? 3 + eq : maxlen; if (len < 19) len = 19; }
                    } else {
                        len = 3 + eq < maxlen ? 3 + eq : maxlen;
                        if (len < 3) len = 3;
                    }
                    const uint32_t off = ip - ref - 1, l2 = len - 2;
                    if (l2 < 7) {
                        out[op] = (uint8_t)((off >> 8) + (l2 << 5));
                        out[op + 1] = (uint8_t)off;
                        op += 2;
                    } else {
                        out[op] = (uint8_t)((off >> 8) + (7u << 5));
                        out[op + 1] = (uint8_t)(l2 - 7);
                        out[op + 2] = (uint8_t)off;
                        op += 3;
                    }
                    lit = 0; op += 1;
                    ip += len;
                    if (ip + 2 >= n) {
                        state = TAIL;
                    } else { // VERY_FAST: only the last two positions of the match are inserted
                        const uint32_t w = lz::rd32(g, ip - 2); // bytes ip-2 .. ip+1
                        tab_put(tab, lzf_slot(w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu), ip - 2);
                        tab_put(tab, lzf_slot((w >> 8) & 0xFFu, (w >> 16) & 0xFFu, w >> 24), ip - 1);
                        v = lzf_rd(g, ip, n);
                    }
                }
            } else {
                if (op >= cap) {
                    fail = true; state = TAIL;
                } else {
                    lit++;
                    out[op++] = (uint8_t)b0;
                    if (lit == kMaxLit) { out[op - lit - 1] = (uint8_t)(kMaxLit - 1); lit = 0; op++; }
                    ip++;
                    if (ip + 2 < n) v = (v >> 8) | ((uint32_t)(ip + 3 < n ? g[ip + 3] : 0u) << 24);
                    else state = TAIL;
                }
            }
        }

        if (state == TAIL) {
            if (!fail) {
                if (op + 3 > cap) {
                    fail = true;
                } else {
                    while (ip < n) {
                        lit++;
                        out[op++] = g[ip++];
                        if (lit == kMaxLit) { out[op - lit - 1] = (uint8_t)(kMaxLit - 1); lit = 0; op++; }
                    }
                    if (lit) out[op - lit - 1] = (uint8_t)(lit - 1);
                    else op -= 1;
                }
            }
            sizes[blk] = fail ? 0u : op;
            state = NEXT;
        }
    }
}

namespace {
struct LinkSpace {
    uint16_t *p = nullptr; size_t cap = 0; uint32_t *counter = nullptr;
    uint16_t *lane_tabs = nullptr; size_t lane_cap = 0; // tables of the lane-per-block parser: 128 KiB per lane
    uint32_t *handback = nullptr; size_t hb_cap = 0;    // blocks the lanes passed on to the link/chain kernels
    hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; // the lane parser's stream beside the rounds
};
struct LinkEntry { LinkSpace s; std::mutex launch; };
std::mutex link_lock;
std::unordered_map<uint64_t, LinkEntry> link_map; // references stay valid across inserts
}

void lzf_release_workspaces()
{
    std::lock_guard<std::mutex> g(link_lock);
    for (auto &kv : link_map) {
        if (kv.second.s.p) (void)hipFree(kv.second.s.p);
        if (kv.second.s.counter) (void)hipFree(kv.second.s.counter);
        if (kv.second.s.lane_tabs) (void)hipFree(kv.second.s.lane_tabs);
        if (kv.second.s.handback) (void)hipFree(kv.second.s.handback);
        if (kv.second.s.side) { (void)hipStreamDestroy(kv.second.s.side); (void)hipEventDestroy(kv.second.s.fork); (void)hipEventDestroy(kv.second.s.join); }
    }
    link_map.clear();
}

void lzf_release_stream(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(link_lock);
    auto it = link_map.find(ws_key(stream));
    if (it == link_map.end()) return;
    auto &w = it->second.s;
    if (w.p) (void)hipFree(w.p);
    if (w.counter) (void)hipFree(w.counter);
    if (w.lane_tabs) (void)hipFree(w.lane_tabs);
    if (w.handback) (void)hipFree(w.handback);
    if (w.side) { (void)hipStreamDestroy(w.side); (void)hipEventDestroy(w.fork); (void)hipEventDestroy(w.join); }
    link_map.erase(it);
}

hipError_t lzf_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, uint8_t *dst,
                      size_t dst_stride, uint32_t *sizes, hipStream_t stream)
{
    // what this call launches, noted where it is launched (cw_profile_kernels); names as rocprofv3 prints them
    char launched[320] = "";
    auto note = [&](const char *name) {
        if (strstr(launched, name)) return; // (rounds repeat their kernels)
        const size_t used = strlen(launched);
        if (used + strlen(name) + 4 < sizeof launched) { if (used) strcat(launched, " + "); strcat(launched, name); }
    };
    if (nblocks == 0) return hipSuccess;
    if (block_bytes == 0 || block_bytes > 65536) return hipErrorInvalidValue;
    const uint32_t n = (uint32_t)block_bytes;
    const uint32_t in_lds = n <= kInLdsMax ? 1u : 0u;
    const uint32_t lds = kLzfTabBytes + (in_lds ? ((n + 15u) & ~15u) + 16u : 0u); // 16 bytes of slack for dword reads
    static bool attr_set = false; // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lzf_blocks_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLzfTabBytes + kInLdsMax + 16);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(lzf_parse_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLzfTabBytes + kInLdsMax + 16);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(lzf_parse_kernel<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLzfTabBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t grid = nblocks < 256 ? nblocks : 256; // the 128 KiB table admits one workgroup per CU
    // CW_LZF_MODE=cut: write/read-back kernel only; =table: exchange kernel with the 128 KiB table also for small blocks
    const char *mode = tune("CW_LZF_MODE");
    const bool cut_only = mode && strcmp(mode, "cut") == 0, table_only = mode && strcmp(mode, "table") == 0;
    const char *redo_env = tune("CW_LZ_FORCE_REDO"); // test knob, see lz4_kernel.hip
    const uint32_t force_redo = redo_env && atoi(redo_env) > 0 ? 1u : 0u;
    if (!cut_only && !table_only && n >= 16) {
        // links for a round of blocks, then the chain parser over that round
        const char *lm_env = tune("CW_LZF_LDS_MAX"); // profiling knob: largest block parsed from LDS-resident links
        // measured (text): 4 KiB 13.2 (LDS) vs 12.4 GB/s (global links); 8 KiB 7.2 vs 10.6; 16 KiB 3.7 vs 9.3 -- blocks per CU win
        const uint32_t lds_max = lm_env && atoi(lm_env) > 0 ? (uint32_t)atoi(lm_env) : 4096u;
        const bool big = n > (lds_max < kChainMax ? lds_max : kChainMax);
        const uint32_t n2 = (n + 63u) & ~63u;
        const size_t ws_bytes = big ? (size_t)1 << 30 : (size_t)256 << 20; // links per round
        const char *round_env = tune("CW_LZF_ROUND"); // test knob: blocks per round (many rounds on small data)
        const char *lanes_env = tune("CW_LZF_LANES");
        // (blocks of 4-8 KiB: the chain kernels win up to ~18 Ki blocks -- text, 8 KiB, 16 Ki blocks 11.0 against 10.6 GB/s, 24 Ki 11.3 / 13.0)
        const size_t lane_min = lanes_env ? (size_t)atoi(lanes_env) : (big ? (n > 8192 ? kLzfLaneMinBlocks : 18432u) : kLzfLaneMinSmall);
        const char *cc_env = tune("CW_LANES_CONCURRENT");
        bool use_lanes = lane_min && nblocks >= lane_min;
        // blocks > 4 KiB: the scalar-thread form of the chain parser (CW_LZF_STHREAD=0: the wavefront-wide one); needs dword-aligned blocks
        const char *st_env = tune("CW_LZF_STHREAD");
        const char *stw_env = tune("CW_LZF_ST_WPC");
        const size_t st_wpc = stw_env && atoi(stw_env) > 0 ? (size_t)atoi(stw_env) : 20;
        const bool sthread = big && (st_env ? st_env[0] != '0' : true) && ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) == 0;
        // lanes BESIDE the rounds: blocks <= 4 KiB always; larger blocks from kLzfBigBesideMin blocks on, and only with the scalar-thread
        // parser in the rounds (with the wavefront-wide chain kernel in the rounds: 256 Ki blocks 27.7 -> 27.7 GB/s)
        const bool big_beside = sthread && n > 16384 && nblocks >= kLzfBigBesideMin;
        bool beside = use_lanes && (cc_env ? cc_env[0] != '0' : !big || big_beside);
        const size_t chunk_cap = round_env && atoi(round_env) > 0 ? (size_t)atoi(round_env) : beside ? (big ? (nblocks < kLzfBigBesideWide ? kLzfBigBesideRoundMid : kLzfBigBesideRound) : kLzfBesideRound) : ws_bytes / (2 * (size_t)n2);
        const size_t chunk_max = chunk_cap < ws_bytes / (2 * (size_t)n2) ? chunk_cap : ws_bytes / (2 * (size_t)n2);
        const size_t chunk = nblocks < chunk_max ? nblocks : chunk_max;
        LinkSpace ls;
        LinkEntry *entry;
        {
            std::lock_guard<std::mutex> g(link_lock);
            entry = &link_map[ws_key(stream)];
        }
        std::lock_guard<std::mutex> sequence(entry->launch); // the link array and the counter are shared by the launches below
        // Large batches: the lane-per-block parser (CW_LZF_LANES=0 off, =N threshold, 1 = every block, in the tests; CW_LANES_WPC
        // wavefronts per CU).  Blocks > 4 KiB from kLzfLaneMinBlocks on: the lanes take the whole batch.  Blocks that fit the
        // LDS-resident chain parser, from kLzfLaneMinSmall on: the lanes run BESIDE the link/chain rounds on a second stream,
        // pulling from the top of the batch while the rounds climb from the bottom (LaneShare) -- one side is bound by LDS
        // capacity and chain latency, the other by random memory accesses.
        uint32_t lane_reserve = 0;
        const size_t round_max = ws_bytes / (2 * (size_t)n2); // blocks per round that the link workspace admits
        const size_t hb_chunk = nblocks < round_max ? nblocks : round_max; // rounds of the hand-back pass
        hipError_t e;
        {
            LinkSpace &w = entry->s;
            const size_t need = use_lanes && hb_chunk > chunk ? hb_chunk : chunk;
            if (w.cap < need * n2) {
                if (w.p) { e = hipFree(w.p); if (e != hipSuccess) return e; }
                w.p = nullptr; w.cap = 0;
                e = hipMalloc(reinterpret_cast<void **>(&w.p), need * n2 * sizeof(uint16_t));
                if (e != hipSuccess) return e;
                w.cap = need * n2;
            }
            if (!w.counter && (e = hipMalloc(reinterpret_cast<void **>(&w.counter), kCtrBytes)) != hipSuccess) return e;
        }
        size_t lgrid = 0; // workgroups of the lane-per-block kernel
        if (use_lanes) {
            const char *lw_env = tune("CW_LANES_WPC");
            const size_t lwpc = lw_env && atoi(lw_env) > 0 ? (size_t)atoi(lw_env) : 4;
            lgrid = (nblocks + 63) / 64;
            if (lgrid > 256 * lwpc) lgrid = 256 * lwpc;
            const char *rs0_env = tune("CW_LANES_RESERVE");
            const size_t want_reserve = rs0_env && atoi(rs0_env) > 0 ? (size_t)atoi(rs0_env) : big ? (nblocks < kLzfBigBesideWide ? kLzfBigBesideReserveMid : kLzfBigBesideRound) : nblocks < 49152 ? kLzfBesideReserveFew : kLzfBesideReserve;
            if (beside && lgrid * 64 + want_reserve > nblocks) lgrid = nblocks > want_reserve + 64 ? (nblocks - want_reserve) / 64 : 1; // (no lane without a block)
            LinkSpace &w = entry->s;
            if (w.lane_cap < lgrid * 64) {
                if (w.lane_tabs) { e = hipFree(w.lane_tabs); if (e != hipSuccess) return e; }
                w.lane_tabs = nullptr; w.lane_cap = 0;
                e = hipMalloc(reinterpret_cast<void **>(&w.lane_tabs), lgrid * 64 * (size_t)kLzfTabBytes);
                if (e != hipSuccess) { // up to 8 GiB: a nearly full device does without the lanes instead of failing the call
                    (void)hipGetLastError();
                    w.lane_tabs = nullptr;
                    use_lanes = beside = false;
                } else {
                    w.lane_cap = lgrid * 64;
                }
            }
        }
        if (use_lanes) {
            LinkSpace &w = entry->s;
            if (w.hb_cap < nblocks) { // a block is handed back once at most
                if (w.handback) { e = hipFree(w.handback); if (e != hipSuccess) return e; }
                w.handback = nullptr; w.hb_cap = 0;
                e = hipMalloc(reinterpret_cast<void **>(&w.handback), nblocks * sizeof(uint32_t));
                if (e != hipSuccess) return e;
                w.hb_cap = nblocks;
            }
            if ((e = hipMemsetAsync(w.counter, 0, kCtrBytes, stream)) != hipSuccess) return e;
            hipStream_t lstream = stream;
            if (beside) {
                if (!w.side) {
                    if ((e = hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking)) != hipSuccess) return e;
                    if ((e = hipEventCreateWithFlags(&w.fork, hipEventDisableTiming)) != hipSuccess) return e;
                    if ((e = hipEventCreateWithFlags(&w.join, hipEventDisableTiming)) != hipSuccess) return e;
                }
                const char *rs_env = tune("CW_LANES_RESERVE");
                lane_reserve = rs_env && atoi(rs_env) > 0 ? (uint32_t)atoi(rs_env) : big ? (nblocks < kLzfBigBesideWide ? kLzfBigBesideReserveMid : kLzfBigBesideRound) : nblocks < 49152 ? kLzfBesideReserveFew : kLzfBesideReserve;
                if (lane_reserve < 1) lane_reserve = 1; // (0 means "on their own" to the kernel; the protocol itself needs no reserve)
                if ((e = hipEventRecord(w.fork, stream)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(w.side, w.fork, 0)) != hipSuccess) return e;
                lstream = w.side;
            }
            if (n <= 4096)
                { note(beside ? "cw::lzf_lanes_kernel<true> [side stream]" : "cw::lzf_lanes_kernel<true>");
                hipLaunchKernelGGL(lzf_lanes_kernel<true>, dim3((unsigned)lgrid), dim3(64), 0, lstream, src, n, src_stride, nblocks, dst, dst_stride,
                                   sizes, w.lane_tabs, w.counter, lane_reserve, w.handback); }
            else
                { note(beside ? "cw::lzf_lanes_kernel<false> [side stream]" : "cw::lzf_lanes_kernel<false>");
                hipLaunchKernelGGL(lzf_lanes_kernel<false>, dim3((unsigned)lgrid), dim3(64), 0, lstream, src, n, src_stride, nblocks, dst, dst_stride,
                                   sizes, w.lane_tabs, w.counter, lane_reserve, w.handback); }
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        ls = entry->s;
        static bool chain_attr = false;
        if (!chain_attr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(lzf_links_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLzfTabBytes + kChainMax + 48);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(lzf_chain_kernel<false>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kChainMax + 256);
            if (e != hipSuccess) return e;
            chain_attr = true;
        }
        const uint32_t links_lds = kLzfTabBytes + ((big ? kChainMax + 16 : n) + 15u) / 16u * 16u + 32u;
        const uint32_t chain_lds = big ? n2 / 8 + 16u : 2 * n2 + ((n + 15u) & ~15u) + 16u;
        size_t per_cu = (160u * 1024u) / (chain_lds + 64);
        if (per_cu > (big ? 20u : 16u)) per_cu = big ? 20 : 16;
        // one round of the link + chain kernels: blocks [first, first + nb) of the batch, or entries [first, first + nb) of the
        // hand-back list (as far as the lanes filled it: the kernels read its length on the device and return at once beyond it)
        const char *gu_env = tune("CW_LZF_SHARE_GIVE_UP"); // test knob: every workgroup but a round's claimant gives up at once
        const uint32_t spin_cap = gu_env && atoi(gu_env) > 0 ? 0u : kShareSpinCap;
        auto round = [&](size_t first, size_t nb, bool listed) -> hipError_t {
            hipError_t r = hipMemsetAsync(ls.counter, 0, sizeof(uint32_t), stream);
            if (r != hipSuccess) return r;
            const LaneShare share = {beside && !listed ? ls.counter : nullptr, first, nblocks, (uint32_t)(first / chunk + 1), spin_cap};
            const BlockList list = {listed ? ls.handback : nullptr, listed ? ls.counter + kCtrHanded : nullptr, (uint32_t)first};
            const size_t off = listed ? 0 : first; // listed blocks are addressed through the list, from the batch's base
            note(listed ? "cw::lzf_links_kernel (handed-back blocks)" : "cw::lzf_links_kernel");
            note(sthread ? "cw::lzf_sthread_kernel" : big ? "cw::lzf_chain_kernel<true>" : "cw::lzf_chain_kernel<false>");
            hipLaunchKernelGGL(lzf_links_kernel, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(64), links_lds, stream, src + off * src_stride, n,
                               src_stride, nb, ls.p, n2, sizes + off, force_redo, share, list);
            size_t cgrid = nb < 256 * per_cu ? nb : 256 * per_cu;
            if (sthread) { // (n2 / 8 bytes of LDS per workgroup: 20 per CU at 64 KiB; smaller blocks: as many wavefronts as measured to pay)
                const size_t st_cu = (160u * 1024u) / (n2 / 8) < st_wpc ? (160u * 1024u) / (n2 / 8) : st_wpc;
                cgrid = nb < 256 * st_cu ? nb : 256 * st_cu;
            }
            if (sthread)
                hipLaunchKernelGGL(lzf_sthread_kernel, dim3((unsigned)cgrid), dim3(64), n2 / 8, stream, src + off * src_stride, n, src_stride, nb,
                                   dst + off * dst_stride, dst_stride, sizes + off, ls.p, n2, ls.counter, share, list);
            else if (big)
                hipLaunchKernelGGL(lzf_chain_kernel<true>, dim3((unsigned)cgrid), dim3(64), chain_lds, stream, src + off * src_stride, n, src_stride,
                                   nb, dst + off * dst_stride, dst_stride, sizes + off, ls.p, n2, ls.counter, share, list);
            else
                hipLaunchKernelGGL(lzf_chain_kernel<false>, dim3((unsigned)cgrid), dim3(64), chain_lds, stream, src + off * src_stride, n, src_stride,
                                   nb, dst + off * dst_stride, dst_stride, sizes + off, ls.p, n2, ls.counter, share, list);
            return hipGetLastError();
        };
        if (!use_lanes || beside)
            for (size_t first = 0; first < nblocks; first += chunk)
                if ((e = round(first, nblocks - first < chunk ? nblocks - first : chunk, false)) != hipSuccess) return e;
        if (beside) { // the hand-back pass, the redo pass and the caller's later work wait for the lanes
            e = hipEventRecord(ls.join, ls.side);
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, ls.join, 0);
            if (e != hipSuccess) return e;
        }
        if (use_lanes && tune("CW_DEBUG_LZF")) {
            uint32_t h[kCtrBytes / 4];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(h, ls.counter, kCtrBytes, hipMemcpyDeviceToHost);
            fprintf(stderr, "lzf lanes: taken %u claimed %u poor %u fine %u handed back %u of %zu; claim ticket %u, gave up %u\n", h[kCtrWord],
                    h[kCtrWord + 1], h[kCtrPoor], h[kCtrFine], h[kCtrHanded], nblocks, h[kCtrTicket], h[kCtrFailed]);
        }
        if (use_lanes)
            for (size_t first = 0; first < nblocks; first += hb_chunk)
                if ((e = round(first, nblocks - first < hb_chunk ? nblocks - first : hb_chunk, true)) != hipSuccess) return e;
        hipLaunchKernelGGL(lzf_blocks_kernel, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                           dst_stride, sizes, in_lds, 1u, beside ? ls.counter + kCtrFailed : nullptr);
        note_kernels(0, launched);
        return hipGetLastError();
    }
    if (!cut_only) {
        if (in_lds)
            { note("cw::lzf_parse_kernel<true>");
            hipLaunchKernelGGL(lzf_parse_kernel<true>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, force_redo); }
        else
            { note("cw::lzf_parse_kernel<false>");
            hipLaunchKernelGGL(lzf_parse_kernel<false>, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                               dst_stride, sizes, force_redo); }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(lzf_blocks_kernel, dim3((unsigned)grid), dim3(64), lds, stream, src, n, src_stride, nblocks, dst,
                       dst_stride, sizes, in_lds, cut_only ? 0u : 1u, static_cast<const uint32_t *>(nullptr));
    if (cut_only) note("cw::lzf_blocks_kernel");
    note_kernels(0, launched);
    return hipGetLastError();
}

} // namespace cw
