// lz4_vtab_kernel.hip -- bit-exact LZ4 block compression (LZ4 v1.8.2 fast parser, byU16 table; the reference's lz4 slot
// LZ4_compress_default(s, d, l, 2*l), src/hashandcompress/HashAndCompress.cpp:351-354) with the parser's hash table held in
// VECTOR REGISTERS: one storage block per wavefront, the wavefront run as ONE scalar thread.
//
// Why.  An exact greedy LZ4 parse is one serial chain per block (table lookup -> candidate compare -> extension), so throughput is
// (chains in flight) / (latency of one chain step).  The table is 8192 x u16 = 16 KiB of randomly accessed state per chain:
//   * in LDS (lz4_parse_kernel): 160 KiB / 16 KiB = 10 chains per CU, 2,560 on the chip;
//   * in HBM (lz4_lanes_*_kernel): 65-131 k chains, but every probe moves whole memory lines -- 48-83x the algorithmic bytes, and
//     a chain step takes microseconds, so small and medium batches (a few thousand blocks) get nothing out of it.
// The largest fast SRAM of a CU is neither: the vector register file, 512 KiB per CU.  A 16 KiB table is 64 VGPRs of a wavefront
// (entry h lives in register h >> 7, lane (h >> 1) & 63, half h & 1), so four such wavefronts fit a SIMD: 16 chains per CU
// BESIDE the ten of the LDS -- with no memory traffic for the table at all.  A VGPR can only be indexed by a wave-uniform value
// (s_set_gpr_idx_on), which is exactly what a scalar thread has: the parse below is the serial parser as it stands (oracle/lz4_oracle.c
// restates it), every value wave-uniform, the table touched through v_readlane_b32 / a one-lane v_mov_b32 on the indexed register, the
// input read through the scalar data cache (s_buffer_load_dwordx8: bounds-checked by the buffer descriptor, so no load reaches outside
// the block), and the 64 lanes used only where a block has width to offer: long match extensions, and the OUTPUT -- the parse records
// its sequences and writes them out 64 at a time, a lane per output byte (emit_batch).
//
// The table registers are the physical VGPRs v64..v127, above the range the compiler may allocate (amdgpu_num_vgpr), named only in
// inline assembly and declared as clobbered there.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cw_device.h"
#include "lz_device.h"
#include "scalar_thread.h"

#pragma clang diagnostic ignored "-Winline-asm" // "clobber list contains reserved registers": that is the point (see below)

namespace cw {

using namespace st;

namespace {

constexpr uint32_t kMinMatch = 4, kLastLiterals = 5, kMFLimit = 12;


// ---- the table: v[64..127] -------------------------------------------------------------------------------------------
// amdgpu_num_vgpr(32): on gfx90a and later the attribute counts half of the unified file, so the compiler allocates v0..v63 (it cannot
// be held below 64: a smaller request is discarded as incompatible with eight wavefronts per SIMD) and v64.. are "reserved" to it --
// the table.  128 registers per wavefront = 4 wavefronts per SIMD, 16 blocks per CU.
#define CW_VT_COMPILER_VGPRS 32
#define CW_VT_BASE "v64"
#define CW_VT_CLOBBER \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
    "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", \
    "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", \
    "v124", "v125", "v126", "v127"

__device__ __forceinline__ void vt_zero()
{
    asm volatile(".irp r,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,1"
                 "00,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127\n\tv_mov_b32 v\\r, 0\n\t.endr"
                 ::: CW_VT_CLOBBER);
}

__device__ __forceinline__ uint32_t hash13(uint32_t v) { return (v * 2654435761u) >> 19; }
__device__ __forceinline__ uint32_t ctz64(unsigned long long m) { return m ? (uint32_t)__builtin_ctzll(m) : 64u; }

// LZ4 length continuation (wavefront-wide): `extra` as a run of 255s closed by one byte < 255; returns bytes written
__device__ __forceinline__ uint32_t put_len(uint8_t *__restrict__ g, uint32_t extra, uint32_t lane)
{
    const uint32_t n255 = extra / 255u;
    for (uint32_t i = lane; i < n255; i += 64) g[i] = 255;
    if (lane == 0) g[n255] = (uint8_t)(extra - n255 * 255u);
    return n255 + 1;
}

// wavefront copy global -> global, any alignment, 4 KiB in flight (lz::copy_g2g keeps 16 KiB in 64 registers: too many here)
__device__ __forceinline__ void copy_run(uint8_t *__restrict__ d, const uint8_t *__restrict__ s, uint32_t len, uint32_t lane)
{
    if (len < 64) {
        if (lane < len) d[lane] = s[lane];
        return;
    }
    const uint32_t head = (uint32_t)(0 - reinterpret_cast<uintptr_t>(d)) & 15u;
    if (lane < head) d[lane] = s[lane];
    d += head; s += head; len -= head;
    const uint32_t nvec = len >> 4;
    uint32_t i = lane;
    for (; i + 3 * 64 < nvec; i += 4 * 64) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) __builtin_memcpy(&v[u], s + 16 * (size_t)(i + 64 * u), 16);
#pragma unroll
        for (int u = 0; u < 4; u++) *reinterpret_cast<uint4 *>(d + 16 * (size_t)(i + 64 * u)) = v[u];
    }
    for (; i < nvec; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, s + 16 * (size_t)i, 16);
        *reinterpret_cast<uint4 *>(d + 16 * (size_t)i) = v;
    }
    const uint32_t done = nvec << 4, tail = len - done;
    if (lane < tail) d[done + lane] = s[done + lane];
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Blocks up to 4 KiB (lz4_vtab2_kernel): a batch of K items per memory round trip, the per-item arithmetic on the vector lanes.
//
// The first form of this parser (all scalar, removed in round 3: 18 GB/s on text at 64 KiB, and it carried the split load/wait pattern
// described above) was bound by SCALAR instruction issue (one scalar ALU per CU, shared by its four SIMDs): ~210 scalar instructions
// per sequence on text; 4 / 8 / 12 / 16 wavefronts per CU gave 9.7 / 14.1 / 16.1 / 18.4 GB/s.
// Here lane k of the wavefront owns item k of a batch -- after a match: insert(ip-2), re-test(ip), probe(ip+1), probe(ip+2); in a
// continuing search: K probes -- and does everything that is per item on the VALU, all items at once: the item's position from the
// skip schedule, its 4 + 8 + 4 bytes (one unaligned load each), its hash, and later the candidate's bytes, the 4-byte test and both
// extensions.  Only what touches the table stays serial and scalar: item after item in position order, each a read and a write of
// the indexed register (the writes of items behind the first match are taken back from saved words, in reverse order), ~16 scalar
// instructions per item.  One candidate round trip per batch instead of one per probe.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kItems = 4;
struct Saved { uint32_t r, l, w; };
// entry h <- pos; returns the previous entry and what vt_restore needs to take the write back
__device__ __forceinline__ uint32_t vt_exchange_s(uint32_t h, uint32_t pos, Saved &sv)
{
    const uint32_t sh = (h & 1u) << 4;
    uint32_t w, r, l;
    // (register and lane are worked out INSIDE the statement: a lane select that a VALU instruction wrote -- a spilled value coming back
    // through v_readlane_b32 -- needs four wait states in front of v_readlane_b32, and the compiler does not look into inline assembly)
    asm volatile("s_lshr_b32 %1, %3, 7\n\t"
                 "s_bfe_u32 %2, %3, 0x60001\n\t"
                 "s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "v_readlane_b32 %0, " CW_VT_BASE ", %2\n\t"
                 "s_set_gpr_idx_off"
                 : "=&s"(w), "=&s"(r), "=&s"(l) : "s"(h) : "scc", CW_VT_CLOBBER);
    const uint32_t nw = (w & ~(0xFFFFu << sh)) | (pos << sh);
    asm volatile("s_lshl_b64 exec, 1, %2\n\t"
                 "s_set_gpr_idx_on %0, gpr_idx(DST)\n\t"
                 "v_mov_b32 " CW_VT_BASE ", %1\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_mov_b64 exec, -1"
                 :: "s"(r), "s"(nw), "s"(l) : CW_VT_CLOBBER);
    sv.r = r; sv.l = l; sv.w = w;
    return (w >> sh) & 0xFFFFu;
}
__device__ __forceinline__ void vt_restore(const Saved &sv)
{
    asm volatile("s_lshl_b64 exec, 1, %2\n\t"
                 "s_set_gpr_idx_on %0, gpr_idx(DST)\n\t"
                 "v_mov_b32 " CW_VT_BASE ", %1\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_mov_b64 exec, -1"
                 :: "s"(sv.r), "s"(sv.w), "s"(sv.l) : CW_VT_CLOBBER);
}
template <int KLANE>
__device__ __forceinline__ void put_lane(uint32_t &v, uint32_t s)
{
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(KLANE));
}
__device__ __forceinline__ uint32_t probe_delta(uint32_t k)
{
    const uint32_t t = 62 + k, q = t >> 6, r = t & 63;
    return k ? 1 + q * (32 * (q - 1) + r + 1) : 0;
}
struct ItemBytes { uint32_t before, at, a0, a1; }; // [p-4, p), [p, p+4), [p+4, p+8), [p+8, p+12)
// the 16 bytes around position p of the block at g (p + 12 <= n); positions below 4 have their `before` bytes moved to the top
__device__ __forceinline__ ItemBytes item_bytes(const uint8_t *g, uint32_t p)
{
    ItemBytes b;
    uint32_t w[3];
    __builtin_memcpy(w, g + p, 12); // one unaligned global_load_dwordx3
    const uint32_t pb = p < 4 ? 0u : p - 4;
    uint32_t bw;
    __builtin_memcpy(&bw, g + pb, 4);
    b.before = p < 4 ? bw << ((4u - p) * 8u) : bw; // (p = 0: shift by 32 is never used: nothing lies before position 0)
    b.at = w[0]; b.a0 = w[1]; b.a1 = w[2];
    return b;
}
} // namespace

__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(CW_VT_COMPILER_VGPRS)))
lz4_vtab2_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                 uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters, uint32_t min_queued,
                 uint32_t max_queued, uint32_t reserve)
{
    constexpr int K = kItems;
    const uint32_t lane = threadIdx.x;
    const uint32_t kk = lane < (uint32_t)K ? lane : (uint32_t)K - 1; // lanes beyond the batch repeat its last item (no divergence, same lines)
    const uint32_t qcount = __builtin_amdgcn_readfirstlane(counters[1]);
    if (qcount < min_queued || qcount >= max_queued) return;
    const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals;

    for (;;) {
        uint32_t qi = qcount;
        if (lane == 0 && (!reserve || __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + reserve < qcount))
            qi = atomicAdd(&counters[0], 1u);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (qi >= qcount) break;
        const size_t blk = queue[qi];
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;
        vt_zero();

        uint32_t anchor = 0, op = 0;
        uint32_t pend_val = 0, pend_pos = 0, pend_cnt = 0; // literals on their way from memory, stored one sequence later

        if (n >= kMFLimit + 1) {
            // a batch: head (after a match at s): lane 0 inserts s - 2, lane 1 re-tests s, lanes 2.. are probes 0.. of the search that
            // starts at ip0 = s + 1; otherwise lanes 0.. are probes i0.. of the search that started at ip0
            bool head = false;
            uint32_t s = 0, ip0 = 1, i0 = 0;
            for (;;) {
                // ---- per lane: position, liveness, bytes, hash ----
                const uint32_t shift = head ? 2u : 0u;
                const uint32_t pi = i0 + kk - shift;                       // probe index of this lane (lanes below `shift`: not a probe)
                uint32_t pos, nxt;
                if (i0 + (uint32_t)K < 64u) { pos = ip0 + pi; nxt = pos + 1; }    // the first 65 probes of a search are consecutive
                else { pos = ip0 + probe_delta(pi); nxt = ip0 + probe_delta(pi + 1); }
                bool live = nxt <= mflimit + 1;                             // the parser stops before a probe whose successor passes the limit
                if (kk < shift) { pos = kk == 0 ? s - 2 : s; live = true; }
                const uint32_t nlive = ctz64(~__ballot(live) | (1ull << K));
                const uint32_t safe = live ? pos : 0u;                      // dead lanes read the block's start (their results are never used)
                const ItemBytes pb = item_bytes(g, safe);
                const uint32_t h = (pb.at * 2654435761u) >> 19;

                // ---- table, item after item ----
                uint32_t cand = 0;
                Saved sv[K];
#define CW_VT_ITEM(I)                                                                                              \
                if ((uint32_t)(I) < nlive) {                                                                                \
                    const uint32_t old_ = vt_exchange_s(__builtin_amdgcn_readlane(h, I), __builtin_amdgcn_readlane(pos, I), sv[I]); \
                    put_lane<I>(cand, old_);                                                                                \
                }
                CW_VT_ITEM(0) CW_VT_ITEM(1) CW_VT_ITEM(2) CW_VT_ITEM(3)
#undef CW_VT_ITEM
                static_assert(K == 4, "the item list above is written out for four items");

                // ---- candidates: bytes, the 4-byte test, both extensions (per lane) ----
                const bool tested = kk < nlive && kk >= (head ? 1u : 0u); // (lane 0 of a head batch only inserts)
                ItemBytes cb = item_bytes(g, tested ? cand : 0u);
                asm volatile("" : "+v"(cb.before), "+v"(cb.at)); // both loads in flight before the test (hipcc would sink the second one behind it)
                const unsigned long long hits = __ballot(tested && cb.at == pb.at) & ((1ull << K) - 1);
                if (!hits) {
                    if (nlive < (uint32_t)K) break;                        // the search ran into the end of the block: last literals
                    i0 += (uint32_t)K - shift;
                    head = false;
                    continue;
                }
                const uint32_t j = (uint32_t)__builtin_ctzll(hits);
                // take back the writes of the items behind the match, last first
                if (3 > j && 3 < nlive) vt_restore(sv[3]);
                if (2 > j && 2 < nlive) vt_restore(sv[2]);
                if (1 > j && 1 < nlive) vt_restore(sv[1]);
                // forward: bytes 4..11 of the match; backward: up to 4 bytes
                uint32_t fwd, bck;
                {
                    const uint32_t x0 = pb.a0 ^ cb.a0, x1 = pb.a1 ^ cb.a1;
                    const uint32_t f = x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : 8u;
                    const uint32_t y = pb.before ^ cb.before;
                    const uint32_t b = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
                    const uint32_t packed = __builtin_amdgcn_readlane(f | (b << 8), j);
                    fwd = packed & 0xFFu; bck = packed >> 8;
                }
                const uint32_t cur = __builtin_amdgcn_readlane(pos, j), match = __builtin_amdgcn_readlane(cand, j);
                uint32_t back = 0;
                if (!(head && j == 1)) { // (a re-test hit takes no literals back: the parser jumps straight to the match)
                    const uint32_t room = cur - anchor < match ? cur - anchor : match;
                    back = bck < room ? bck : room;
                    if (bck == 4 && room > 4) {
                        for (;;) {
                            const uint32_t jj = back + lane + 1;
                            const bool ok = jj <= room && g[cur - jj] == g[match - jj];
                            const uint32_t cnt = ctz64(~__ballot(ok));
                            back += cnt;
                            if (cnt < 64) break;
                        }
                    }
                }
                uint32_t mc = fwd;
                {
                    const uint32_t lim = matchlimit - (cur + kMinMatch);
                    if (mc == 8 && lim > 8) {
                        for (;;) {
                            const uint32_t i = cur + kMinMatch + mc + lane;
                            const bool ok = i < matchlimit && g[i] == g[match + kMinMatch + mc + lane];
                            const uint32_t cnt = ctz64(~__ballot(ok));
                            mc += cnt;
                            if (cnt < 64) break;
                        }
                    }
                    if (mc > lim) mc = lim;
                }
                const uint32_t mend = cur + kMinMatch + mc; // first byte after the match
                const uint32_t off = cur - match;
                const uint32_t lit = cur - back - anchor;
                mc += back;

                // ---- emit: token, literals [anchor, anchor + lit), offset, match length ----
                if (pend_cnt) {
                    if (lane < pend_cnt) out[pend_pos + lane] = (uint8_t)pend_val;
                    pend_cnt = 0;
                }
                const uint32_t tok_pos = op;
                uint32_t token;
                op += 1;
                if (lit >= 15) { token = 15u << 4; op += put_len(out + op, lit - 15, lane); }
                else token = lit << 4;
                if (lit) {
                    if (lit <= 64) {
                        pend_val = lane < lit ? g[anchor + lane] : 0u;
                        pend_pos = op; pend_cnt = lit;
                    } else {
                        copy_run(out + op, g + anchor, lit, lane);
                    }
                    op += lit;
                }
                const uint32_t off_pos = op;
                op += 2;
                if (mc >= 15) { token += 15; op += put_len(out + op, mc - 15, lane); }
                else token += mc;
                if (lane < 3) {
                    const uint32_t where = lane == 0 ? tok_pos : off_pos + lane - 1;
                    const uint32_t what = lane == 0 ? token : lane == 1 ? off : off >> 8;
                    out[where] = (uint8_t)what;
                }
                anchor = mend;
                if (mend > mflimit) break; // end of parse: the remaining bytes are literals
                head = true; s = mend; ip0 = mend + 1; i0 = 0;
            }
        }
        if (pend_cnt && lane < pend_cnt) out[pend_pos + lane] = (uint8_t)pend_val;

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
            copy_run(out + op, g + anchor, run, lane);
            op += run;
        }
        if (lane == 0) sizes[blk] = op;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Third generation: the first kernel's serial chain (scalar loads through the scalar cache: the lowest latency the chip offers, and
// one candidate round trip per probe as the parser needs it), with the scalar unit relieved.  The first kernel issues ~210 scalar
// instructions per sequence and is bound by that (16 wavefronts share one scalar ALU per CU); the vector ALUs are idle.  Here
//   * the half-word arithmetic of a table operation (extract the old entry, merge the new one into its 32-bit word) runs on the VALU
//     on wave-uniform operands, inside one block of inline assembly with the two indexed register accesses;
//   * windows are loaded from the position's own dword (no bytes in front of it, so no special case for positions below 4); the
//     bytes in front are fetched only when a match may actually move back (literals pending and candidate > 0);
//   * consecutive probes re-use the window in registers (a new one is loaded when the position crosses a dword);
//   * both extensions, the token and all output addresses are computed on the VALU in wave-uniform VGPRs (anchor and output
//     position live there); the scalar unit keeps what only it can do: scalar loads and their addresses, the hash (one s_mul),
//     register indices, EXEC masks and branches.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// entry h <- pos; returns the previous entry.  k_ffff: a VGPR holding 0xFFFF in every lane.
__device__ __forceinline__ uint32_t vt3_exchange(uint32_t h, uint32_t pos, uint32_t k_ffff)
{
    uint32_t old, w, t, ps, m, nw, r, l;
    asm volatile("s_lshr_b32 %[r], %[h], 7\n\t"              // register and lane of the entry, on the scalar ALU and in here (see vt_exchange_s)
                 "s_bfe_u32 %[l], %[h], 0x60001\n\t"
                 "s_set_gpr_idx_on %[r], gpr_idx(SRC0)\n\t"
                 "v_readlane_b32 %[w], " CW_VT_BASE ", %[l]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "v_and_b32 %[t], 1, %[h]\n\t"
                 "v_lshlrev_b32 %[t], 4, %[t]\n\t"             // 0 or 16: the half of the word
                 "v_lshlrev_b32_e64 %[ps], %[t], %[pos]\n\t"
                 "v_lshlrev_b32 %[m], %[t], %[kf]\n\t"
                 "v_bfi_b32 %[nw], %[m], %[ps], %[w]\n\t"      // (mask & pos << sh) | (~mask & word)
                 "v_lshrrev_b32_e64 %[t], %[t], %[w]\n\t"
                 "s_lshl_b64 exec, 1, %[l]\n\t"
                 "s_set_gpr_idx_on %[r], gpr_idx(DST)\n\t"
                 "v_mov_b32 " CW_VT_BASE ", %[nw]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 "v_readfirstlane_b32 %[old], %[t]\n\t"
                 "s_and_b32 %[old], %[old], 0xffff"
                 : [old] "=&s"(old), [w] "=&s"(w), [t] "=&v"(t), [ps] "=&v"(ps), [m] "=&v"(m), [nw] "=&v"(nw), [r] "=&s"(r), [l] "=&s"(l)
                 : [h] "s"(h), [pos] "s"(pos), [kf] "v"(k_ffff)
                 : "scc", CW_VT_CLOBBER);
    return old;
}
// entry h <- pos (the previous entry is of no interest: the insert in front of a re-test)
__device__ __forceinline__ void vt3_put(uint32_t h, uint32_t pos, uint32_t k_ffff)
{
    uint32_t w, t, ps, m, nw, r, l;
    asm volatile("s_lshr_b32 %[r], %[h], 7\n\t"
                 "s_bfe_u32 %[l], %[h], 0x60001\n\t"
                 "s_set_gpr_idx_on %[r], gpr_idx(SRC0)\n\t"
                 "v_readlane_b32 %[w], " CW_VT_BASE ", %[l]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "v_and_b32 %[t], 1, %[h]\n\t"
                 "v_lshlrev_b32 %[t], 4, %[t]\n\t"
                 "v_lshlrev_b32_e64 %[ps], %[t], %[pos]\n\t"
                 "v_lshlrev_b32 %[m], %[t], %[kf]\n\t"
                 "v_bfi_b32 %[nw], %[m], %[ps], %[w]\n\t"
                 "s_lshl_b64 exec, 1, %[l]\n\t"
                 "s_set_gpr_idx_on %[r], gpr_idx(DST)\n\t"
                 "v_mov_b32 " CW_VT_BASE ", %[nw]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_mov_b64 exec, -1"
                 : [w] "=&s"(w), [t] "=&v"(t), [ps] "=&v"(ps), [m] "=&v"(m), [nw] "=&v"(nw), [r] "=&s"(r), [l] "=&s"(l)
                 : [h] "s"(h), [pos] "s"(pos), [kf] "v"(k_ffff)
                 : "scc", CW_VT_CLOBBER);
}
// The same two operations on a table in LDS (LDSTAB form of the third kernel, below): slot h of 8192 x u16 at byte address
// tab_lds + 2 h, touched by wave-uniform addresses (a broadcast read; every lane writes the same value).
__device__ __forceinline__ uint32_t lt3_exchange(uint32_t tab_lds, uint32_t h, uint32_t pos)
{
    uint32_t old, t;
    const uint32_t addr = tab_lds + 2u * h;
    asm volatile("ds_read_u16 %[t], %[a]\n\t"
                 "ds_write_b16 %[a], %[p]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_readfirstlane_b32 %[old], %[t]"
                 : [old] "=s"(old), [t] "=&v"(t) : [a] "v"(addr), [p] "v"(pos) : "memory");
    return old;
}
__device__ __forceinline__ void lt3_put(uint32_t tab_lds, uint32_t h, uint32_t pos)
{
    asm volatile("ds_write_b16 %[a], %[p]" :: [a] "v"(tab_lds + 2u * h), [p] "v"(pos) : "memory");
}
// a candidate's window: 32 bytes in scalar registers
struct CandWin {
    u32x8 s;
    template <int I> __device__ __forceinline__ uint32_t dw() const { return to_v(s[I]); }
    template <int I> __device__ __forceinline__ void cut64(uint32_t sh, uint32_t &lo, uint32_t &hi) const { cut64_v(s[I], s[I + 1], s[I + 2], sh, lo, hi); }
};
// ---- emission in batches of 64 sequences ---------------------------------------------------------------------------------------
// The parse only RECORDS a sequence (start of its literals, their number, match length - 4, offset) in lane `number of the sequence in
// the batch` of four vector registers: five VALU instructions on the chain of a sequence instead of the ~80 that computing the token,
// the length bytes, the output addresses and the literal copy for ONE sequence on a whole wavefront came to.  Every 64 sequences (and
// at the end of the block) the batch is written out with the 64 lanes on 64 consecutive OUTPUT bytes: two prefix sums over the
// sequences' encoded sizes give every sequence its place, a lane finds the sequence its byte belongs to by a binary search over the
// prefix sums (six ds_bpermute: no LDS memory), works out which byte of the sequence that is -- token, literal length, literal,
// offset, match length -- and stores it: coalesced byte stores, ~60 instructions per 64 bytes of output whatever the sequences look
// like.  Literal runs longer than kFlatLit are left out of that byte space and copied 1 KiB per step afterwards.
constexpr uint32_t kFlatLit = 96;
// LZ4 length field: number of continuation bytes of the value v (a literal count, or a match length - 4), and the k-th of them
__device__ __forceinline__ uint32_t len_bytes(uint32_t v) { return v >= 15u ? 1u + (v - 15u) / 255u : 0u; }
__device__ __forceinline__ uint32_t len_byte(uint32_t v, uint32_t nx, uint32_t k) { return k + 1u < nx ? 255u : v - 15u - 255u * (nx - 1u); }

// writes sequences 0 .. nrec-1 of the batch (lane i: sequence i) at out + op0; returns the output position behind them
__device__ __forceinline__ uint32_t emit_batch(uint8_t *__restrict__ out, const uint8_t *__restrict__ g, uint32_t op0, uint32_t nrec, uint32_t rec_a,
                                               uint32_t rec_lit, uint32_t rec_ml, uint32_t rec_off, uint32_t lane)
{
    const bool live = lane < nrec;
    const uint32_t lit = live ? rec_lit : 0u, ml = live ? rec_ml : 0u;
    const uint32_t llx = len_bytes(lit), mlx = len_bytes(ml);
    const uint32_t flat = lit > kFlatLit ? 0u : lit;
    const uint32_t full = live ? 3u + llx + lit + mlx : 0u;  // token, literal length bytes, literals, offset (2), match length bytes
    const uint32_t comp = live ? 3u + llx + flat + mlx : 0u; // the same without a long literal run: the byte space of the loop below
    const uint32_t fend = wave_scan_add(full), cend = wave_scan_add(comp);
    const uint32_t fstart = fend - full, cstart = cend - comp;
    const uint32_t ctotal = __builtin_amdgcn_readlane(cend, 63), ftotal = __builtin_amdgcn_readlane(fend, 63);
    // (two or four windows per step, their searches and loads interleaved, change nothing: 12.5 ms per 4,096 blocks of text either way --
    // at 16 wavefronts per CU the other wavefronts fill the waits of this loop)
    for (uint32_t base = 0; base < ctotal; base += 64) {
        const uint32_t j = base + lane;
        uint32_t sq = 0; // number of sequences that end at or before byte j = the sequence of byte j
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1)
            if (lane_get(cend, sq + step - 1u) <= j) sq += step;
        const uint32_t q_cstart = lane_get(cstart, sq), q_fstart = lane_get(fstart, sq), q_a = lane_get(rec_a, sq), q_lit = lane_get(lit, sq),
                       q_ml = lane_get(ml, sq), q_off = lane_get(rec_off, sq);
        const uint32_t r = j - q_cstart;                      // byte r of the sequence (in the compacted space)
        const uint32_t x = len_bytes(q_lit), y = len_bytes(q_ml), fl = q_lit > kFlatLit ? 0u : q_lit;
        const uint32_t pre = 1u + x;                          // token + literal length bytes
        if (j < ctotal) {
            uint32_t byte, addr = q_fstart + r;
            if (r == 0) {
                byte = (min(q_lit, 15u) << 4) | min(q_ml, 15u);
            } else if (r < pre) {
                byte = len_byte(q_lit, x, r - 1u);
            } else if (r < pre + fl) {
                byte = g[q_a + (r - pre)];
            } else {
                const uint32_t r2 = r - pre - fl;
                addr += q_lit - fl;
                byte = r2 == 0 ? q_off : r2 == 1 ? q_off >> 8 : len_byte(q_ml, y, r2 - 2u);
            }
            out[op0 + addr] = (uint8_t)byte;
        }
    }
    for (unsigned long long m = __ballot(live && lit > kFlatLit); m; m &= m - 1) { // long literal runs
        const uint32_t q = (uint32_t)__builtin_ctzll(m);
        copy_run(out + op0 + __builtin_amdgcn_readlane(fstart, q) + 1u + __builtin_amdgcn_readlane(llx, q), g + __builtin_amdgcn_readlane(rec_a, q),
                 __builtin_amdgcn_readlane(lit, q), lane);
    }
    return op0 + ftotal;
}
} // namespace

// (Tried and dropped: two probes per memory round trip -- while probe A's candidate is on its way, probe B is carried out as if A had
// missed (its table exchange behind A's, both candidates and the next window requested together, B's write taken back when A hits):
// one wait per sequence instead of two on text, bit-exact, and slower: text, 64 KiB, 1,024 / 3,233 / 8,192 blocks 8.63 / 17.1 / 20.8 ->
// 6.77 / 12.7 / 15.1 GB/s.  A sequence's time is its ~190 dependent instructions, not its memory waits; the second exchange, the
// third window and the scalar registers they spill cost more than the wait they save.)
// (Tried and dropped for blocks up to 8 KiB: the block staged in LDS beside the register table (4 KiB of LDS per wavefront, so all 16
// wavefronts per CU fit), every byte of the parse an LDS read and the arithmetic on the VALU: bit-exact, 28 GB/s alone on the 4 KiB corpus --
// more than any other single parser there (LDS-resident wavefront parser 25.6, second-generation register-table kernel 17.8) and linear
// in its wavefronts per CU (8 / 12 / 16: 17 / 23.5 / 28), i.e. bound by the chain of one sequence -- but beside the parsers it would
// have to share the CU with it adds nothing: 51,728 blocks 30.0 against 29.5 GB/s, 1 Mi blocks 44.9 against 44.9.)
// (Tried and dropped: the candidate's bytes through the VECTOR memory path -- two bounds-checked buffer loads at a wave-uniform offset,
// compared on the VALU -- on the theory that the scalar cache's few outstanding misses were the queue: text, 64 KiB, 8 Ki blocks
// 20.8 -> 18.4 GB/s, 1,024 blocks 8.65 -> 7.26: the vector path's latency is simply longer.)
// Diagnostic build only (-DCW_VSTAMP, tools/vtab_stamp.hip): where a sequence's cycles go.  A stamp is s_memtime; the differences are summed
// per phase in scalar registers and added to g_vstamp once per block.  In the product build no stamp executes.
#ifdef CW_VSTAMP
__device__ unsigned long long g_vstamp[16];
#define CW_VS_DECL uint32_t vs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vs_seq = 0, vs_probe = 0; unsigned long long vs_last = __builtin_amdgcn_s_memtime();
#define CW_VS_AT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); vs_acc[i] += (uint32_t)(t_ - vs_last); vs_last = t_; } while (0)
#define CW_VS_SEQ() (vs_seq++)
#define CW_VS_PROBE() (vs_probe++)
#define CW_VS_FLUSH() do { if (lane == 0) { for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_vstamp[i_], (unsigned long long)vs_acc[i_]); \
                                            atomicAdd(&g_vstamp[14], (unsigned long long)vs_probe); atomicAdd(&g_vstamp[15], (unsigned long long)vs_seq); } } while (0)
#else
#define CW_VS_DECL
#define CW_VS_AT(i) do { } while (0)
#define CW_VS_SEQ() do { } while (0)
#define CW_VS_PROBE() do { } while (0)
#define CW_VS_FLUSH() do { } while (0)
#endif

// LDSTAB: the same scalar-thread parser with its table in LDS (16 KiB per wavefront, ten per CU) instead of in registers.  It takes
// the place of the round-2 wavefront parser (lz4_parse_kernel<false>) beside the register form for blocks read from global memory: a
// table operation is two LDS instructions at a wave-uniform address instead of an exchange over 64 lanes with its lane-order check and
// undo, and the chain of a sequence is this kernel's (~2,000 cycles) instead of that one's (2,500-3,100).
template <bool LDSTAB>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(CW_VT_COMPILER_VGPRS)))
lz4_vtab3_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                 uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters, uint32_t min_queued,
                 uint32_t max_queued, uint32_t reserve)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t qcount = __builtin_amdgcn_readfirstlane(counters[1]);
    if (qcount < min_queued || qcount >= max_queued) return;
    const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals;
    const uint32_t k_ffff = to_v(0xFFFFu);
    extern __shared__ __attribute__((aligned(16))) uint8_t ltab[]; // LDSTAB: the 8192 x u16 table
    const uint32_t tab_lds = opaque_s((uint32_t)reinterpret_cast<uintptr_t>(ltab));
    (void)tab_lds; (void)k_ffff;

    for (;;) {
        uint32_t qi = qcount;
        if (lane == 0 && (!reserve || __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + reserve < qcount))
            qi = atomicAdd(&counters[0], 1u);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (qi >= qcount) break;
        const size_t blk = queue[qi];
        const uint8_t *g = src + blk * src_stride;
        uint8_t *out = dst + blk * dst_stride;
        u32x4 rs;
        {
            const uint64_t a = reinterpret_cast<uint64_t>(g);
            rs.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
            rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xFFFFu);
            rs.z = n;
            rs.w = 0x00020000u;
        }
        if constexpr (LDSTAB) {
            __syncthreads(); // (one wavefront: orders the previous block's LDS traffic before the clearing stores)
            for (uint32_t i = lane; i < (1u << 13) * 2u / 16u; i += 64) reinterpret_cast<uint4 *>(ltab)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
        } else {
            vt_zero();
        }

        // anchor: wave-uniform, in a vector register.  The batch of recorded sequences: lane i of rec_* = sequence i, nrec of them; s_op = output
        // position behind the last batch written
        uint32_t anchor = to_v(0);
        uint32_t rec_a = 0, rec_lit = 0, rec_ml = 0, rec_off = 0;
        uint32_t nrec = 0, s_op = 0;

        if (n >= kMFLimit + 1) {
            // scalar state of the search: the probe position cur, its window wp = the 32 bytes from pb4 = cur & ~3 on
            uint32_t cur = 1, pb4 = 0;
            u32x8 wp = sc_load32_now(rs, 0);
            CW_VS_DECL
            // one probe: the table exchange, then the candidate's window and the next position's window in ONE round trip
#define CW_VT3_PROBE(FIP)                                                                                                      \
            const uint32_t psh = (cur - pb4) * 8u;                                                                             \
            const uint32_t v = cut32(wp[0], wp[1], psh);                                                                       \
            CW_VS_AT(0); CW_VS_PROBE();                                                                                        \
            const uint32_t cand = LDSTAB ? lt3_exchange(tab_lds, (v * 2654435761u) >> 19, cur)                                 \
                                         : vt3_exchange((v * 2654435761u) >> 19, cur, k_ffff);                                 \
            CW_VS_AT(1);                                                                                                       \
            const uint32_t cb4 = cand & ~3u, csh = (cand & 3u) * 8u, fb4 = (FIP) & ~3u;                                         \
            CandWin wc;                                                                                                        \
            u32x8 wq;                                                                                                          \
            /* (both loads always: with the next window requested only when the next position lies in another dword -- one probe in four at  \
               step 1 -- both forms were 4 % slower: 12.5 -> 13.1 and 17.3 -> 17.9 ms per 4,096 blocks of text) */                       \
            asm volatile("s_buffer_load_dwordx8 %0, %2, %3\n\ts_buffer_load_dwordx8 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"      \
                         : "=&s"(wc.s), "=&s"(wq) : "s"(rs), "s"(cb4), "s"(fb4));                                              \
            CW_VS_AT(2);                                                                                                       \
            const bool hit = cut32(wc.s[0], wc.s[1], csh) == v;
#define CW_VT3_ADVANCE(FIP)                                                                                                    \
            if (fb4 != pb4) { wp = wq; pb4 = fb4; }                                                                            \
            cur = (FIP);

            for (;;) { // one sequence per iteration
                // ---- search: probe cur, then positions with a stride that grows every 64 misses ----
                uint32_t mcur = 0, mcand = 0, mpsh = 0, mcsh = 0, mcb4 = 0;
                CandWin mwc;
                bool found = false;
                {
                    uint32_t step = 1, nb = 64;
                    for (;;) {
                        const uint32_t fip = cur + step;
                        step = nb++ >> 6;
                        if (fip > mflimit + 1) break;
                        CW_VT3_PROBE(fip)
                        if (hit) { found = true; mcur = cur; mcand = cand; mpsh = psh; mcsh = csh; mcb4 = cb4; mwc = wc; break; }
                        CW_VT3_ADVANCE(fip)
                    }
                }
                if (!found) break; // -> last literals
                bool first = true; // the match came out of the search (it may move back); false: out of the re-test after a match
                for (;;) { // next_match
                    CW_VS_AT(0); CW_VS_SEQ();
                    const uint32_t vcur = to_v(mcur), vcand = to_v(mcand);
                    uint32_t back = to_v(0);
                    if (first) {
                        // bytes the match may move back over: pending literals, and the candidate must stay >= 0
                        const uint32_t room = min(vcur - anchor, vcand);
                        if (__builtin_amdgcn_readfirstlane(room) != 0) {
                            // the 4 bytes in front of both positions (an offset below zero reads as zero; room < 4 then anyway)
                            uint32_t fp, fc;
                            sc_load4x2_now(rs, pb4 - 4u, mcb4 - 4u, fp, fc);
                            const uint32_t bp = __builtin_amdgcn_alignbit(to_v(wp[0]), to_v(fp), mpsh);
                            const uint32_t bc = __builtin_amdgcn_alignbit(mwc.template dw<0>(), to_v(fc), mcsh);
                            const uint32_t y = bp ^ bc;
                            const uint32_t b = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
                            back = min(b, room);
                            if (__builtin_amdgcn_readfirstlane((uint32_t)(b == 4 && room > 4))) { // rare: more than 4 bytes
                                uint32_t bk = 4;
                                const uint32_t rm = __builtin_amdgcn_readfirstlane(room);
                                for (;;) {
                                    const uint32_t jj = bk + lane + 1;
                                    const bool ok = jj <= rm && g[mcur - jj] == g[mcand - jj];
                                    const uint32_t cnt = ctz64(~__ballot(ok));
                                    bk += cnt;
                                    if (cnt < 64) break;
                                }
                                back = to_v(bk);
                            }
                        }
                    }
                    CW_VS_AT(3);
                    // forward: bytes 4..11 behind both positions, then 12..19, then the byte loop
                    uint32_t mc;
                    {
                        uint32_t plo, phi, clo, chi;
                        cut64_v(wp[1], wp[2], wp[3], mpsh, plo, phi);
                        mwc.template cut64<1>(mcsh, clo, chi);
                        mc = equal_bytes_v(plo, phi, clo, chi);
                        const uint32_t lim = to_v(matchlimit - kMinMatch) - vcur;
                        if (__builtin_amdgcn_readfirstlane((uint32_t)(mc == 8 && lim > 8))) {
                            cut64_v(wp[3], wp[4], wp[5], mpsh, plo, phi);
                            mwc.template cut64<3>(mcsh, clo, chi);
                            mc = 8u + equal_bytes_v(plo, phi, clo, chi);
                            if (__builtin_amdgcn_readfirstlane((uint32_t)(mc == 16 && lim > 16))) {
                                uint32_t m2 = 16;
                                for (;;) {
                                    const uint32_t i = mcur + kMinMatch + m2 + lane;
                                    const bool ok = i < matchlimit && g[i] == g[mcand + kMinMatch + m2 + lane];
                                    const uint32_t cnt = ctz64(~__ballot(ok));
                                    m2 += cnt;
                                    if (cnt < 64) break;
                                }
                                mc = to_v(m2);
                            }
                        }
                        mc = min(mc, lim);
                    }
                    const uint32_t vmend = vcur + kMinMatch + mc;     // first byte after the match
                    const uint32_t mend = __builtin_amdgcn_readfirstlane(vmend);
                    CW_VS_AT(4);
                    const bool more = mend <= mflimit;

                    // ---- record the sequence: literals [anchor, anchor + lit), offset, match length - 4 ----
                    {
                        const bool mine = lane == nrec;
                        rec_lit = mine ? vcur - back - anchor : rec_lit;
                        rec_a = mine ? anchor : rec_a;
                        rec_ml = mine ? mc + back : rec_ml;
                        rec_off = mine ? vcur - vcand : rec_off;
                        if (++nrec == 64) { s_op = emit_batch(out, g, s_op, 64, rec_a, rec_lit, rec_ml, rec_off, lane); nrec = 0; }
                    }
                    anchor = vmend;
                    CW_VS_AT(5);
                    if (!more) { cur = mend; break; }

                    // ---- table: insert mend - 2, then the immediate re-test at mend ----
                    u32x8 wnext; // the window at mend, and the 8 bytes around mend - 2
                    u32x2 wins;
                    const uint32_t nb4 = mend & ~3u;
                    sc_load32_8_now(rs, nb4, (mend - 2u) & ~3u, wnext, wins);
                    CW_VS_AT(6);
                    if constexpr (LDSTAB) lt3_put(tab_lds, (cut32(wins[0], wins[1], ((mend - 2u) & 3u) * 8u) * 2654435761u) >> 19, mend - 2u);
                    else vt3_put((cut32(wins[0], wins[1], ((mend - 2u) & 3u) * 8u) * 2654435761u) >> 19, mend - 2u, k_ffff);
                    wp = wnext; pb4 = nb4; cur = mend;
                    CW_VS_AT(7);
                    {
                        const uint32_t fip = cur + 1;
                        CW_VT3_PROBE(fip)
                        if (hit) { first = false; mcur = cur; mcand = cand; mpsh = psh; mcsh = csh; mcb4 = cb4; mwc = wc; continue; }
                        CW_VT3_ADVANCE(fip)
                    }
                    break;
                }
                if (cur > mflimit) break; // (cur = mend past the limit: the remaining bytes are literals)
            }
#undef CW_VT3_PROBE
#undef CW_VT3_ADVANCE
            CW_VS_FLUSH();
        }
        const uint32_t s_anchor = __builtin_amdgcn_readfirstlane(anchor);
        if (nrec) s_op = emit_batch(out, g, s_op, nrec, rec_a, rec_lit, rec_ml, rec_off, lane);

        // ---- last literals ----
        {
            const uint32_t run = n - s_anchor;
            const uint32_t tok_pos = s_op;
            s_op += 1;
            if (run >= 15) {
                if (lane == 0) out[tok_pos] = 15u << 4;
                s_op += put_len(out + s_op, run - 15, lane);
            } else if (lane == 0) {
                out[tok_pos] = (uint8_t)(run << 4);
            }
            copy_run(out + s_op, g + s_anchor, run, lane);
            s_op += run;
        }
        if (lane == 0) sizes[blk] = s_op;
    }
}

// grid: as many single-wavefront workgroups as the register file admits (128 VGPRs -> 4 per SIMD, 16 per CU), at most one per queued block
hipError_t lz4_vtab_launch(const uint8_t *src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *dst, size_t dst_stride, uint32_t *sizes,
                           const uint32_t *queue, uint32_t *counters, uint32_t min_queued, uint32_t max_queued, uint32_t reserve,
                           unsigned waves_per_cu, hipStream_t stream, const char **kernel_name)
{
    // CW_VTAB_GEN: 2 = batches of four items (vector loads), 3 = the scalar chain with the VALU's help.  Default by measurement (GB/s alone,
    // 16 wavefronts per CU; (1) = the all-scalar first form, removed): text, 64 KiB blocks, 8 Ki blocks 18.4 (1) / 13.7 (2) / 20.8 (3),
    // 3,233 blocks 13.6 / 11.3 / 17.1; corpus, 4 KiB blocks 15.4-16.0 (1) against 17.8-17.9 (2)
    const char *gen_env = tune("CW_VTAB_GEN");
    const int gen = gen_env && atoi(gen_env) == 2 ? 2 : gen_env && atoi(gen_env) == 3 ? 3 : n <= 4096 ? 2 : 3;
    if ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) return hipErrorInvalidValue; // the scalar loads are dword loads
    size_t grid = 256 * (size_t)(waves_per_cu ? waves_per_cu : 16);
    if (grid > nblocks) grid = nblocks;
    if (grid == 0) return hipSuccess;
    if (kernel_name) *kernel_name = gen == 3 ? "cw::lz4_vtab3_kernel<false>" : "cw::lz4_vtab2_kernel";
    if (gen == 3)
        hipLaunchKernelGGL(lz4_vtab3_kernel<false>, dim3((unsigned)grid), dim3(64), 0, stream, src, n, src_stride, dst, dst_stride, sizes, queue, counters,
                           min_queued, max_queued, reserve);
    else
        hipLaunchKernelGGL(lz4_vtab2_kernel, dim3((unsigned)grid), dim3(64), 0, stream, src, n, src_stride, dst, dst_stride, sizes, queue, counters,
                           min_queued, max_queued, reserve);
    return hipGetLastError();
}

// the scalar-thread parser with its table in LDS: ten single-wavefront workgroups per CU at most (16 KiB of LDS each)
hipError_t lz4_ltab_launch(const uint8_t *src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *dst, size_t dst_stride, uint32_t *sizes,
                           const uint32_t *queue, uint32_t *counters, unsigned waves_per_cu, hipStream_t stream)
{
    if ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) return hipErrorInvalidValue;
    size_t grid = 256 * (size_t)(waves_per_cu && waves_per_cu < 10 ? waves_per_cu : 10);
    if (grid > nblocks) grid = nblocks;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_vtab3_kernel<true>, dim3((unsigned)grid), dim3(64), (1u << 13) * 2u, stream, src, n, src_stride, dst, dst_stride, sizes, queue,
                       counters, 0u, 0xFFFFFFFFu, 0u);
    return hipGetLastError();
}

} // namespace cw
