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
// restates it), every value wave-uniform and kept in SGPRs, the table touched through v_readlane_b32 / a one-lane v_mov_b32 on the
// indexed register, the input read through the scalar data cache (s_buffer_load_dwordx8: bounds-checked by the buffer descriptor, so
// no load reaches outside the block), and the 64 lanes used only where a block has width to offer: literal copies, long match
// extensions, length bytes.
//
// The table registers are the physical VGPRs v64..v127, above the range the compiler may allocate (amdgpu_num_vgpr), named only in
// inline assembly and declared as clobbered there.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cw_device.h"
#include "lz_device.h"

#pragma clang diagnostic ignored "-Winline-asm" // "clobber list contains reserved registers": that is the point (see below)

namespace cw {

namespace {

constexpr uint32_t kMinMatch = 4, kLastLiterals = 5, kMFLimit = 12;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

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

// entry h <- pos; returns the previous entry.  All operands wave-uniform (SGPRs); EXEC is all ones on entry and on exit.
__device__ __forceinline__ uint32_t vt_exchange(uint32_t h, uint32_t pos)
{
    const uint32_t r = h >> 7, l = (h >> 1) & 63u, sh = (h & 1u) << 4;
    uint32_t w;
    // (the register index applies to v_readlane_b32's source and to a one-lane v_mov_b32's destination on gfx950: tools/idxmode.hip)
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "v_readlane_b32 %0, " CW_VT_BASE ", %2\n\t"
                 "s_set_gpr_idx_off"
                 : "=s"(w) : "s"(r), "s"(l) : CW_VT_CLOBBER);
    const uint32_t nw = (w & ~(0xFFFFu << sh)) | (pos << sh);
    asm volatile("s_lshl_b64 exec, 1, %2\n\t"
                 "s_set_gpr_idx_on %0, gpr_idx(DST)\n\t"
                 "v_mov_b32 " CW_VT_BASE ", %1\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_mov_b64 exec, -1"
                 :: "s"(r), "s"(nw), "s"(l) : CW_VT_CLOBBER);
    return (w >> sh) & 0xFFFFu;
}

// ---- the input through the scalar cache -------------------------------------------------------------------------------
// 32 bytes at byte offset `off` (a multiple of 4; may have wrapped below zero) of the block described by rs; dwords outside
// [0, num_records) read as zero.  The caller waits (sc_wait) before it uses them.
__device__ __forceinline__ u32x8 sc_load32(const u32x4 &rs, uint32_t off)
{
    u32x8 v;
    asm volatile("s_buffer_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(rs), "s"(off));
    return v;
}
__device__ __forceinline__ void sc_wait(u32x8 &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
__device__ __forceinline__ void sc_wait(u32x8 &a, u32x8 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

// A Window is the 32 bytes loaded for position q from byte offset (q & ~3) - 4: `before` = bytes [q-4, q), `at` = [q, q+4),
// `after` = [q+4, q+12), `after2` = [q+12, q+20).  For q < 4 the load starts at offset 0 (an offset that wrapped below zero makes the
// whole load read as zero: tools/sbuf.hip) and win_ready moves the dwords up by one, so that the accessors stay the same.
struct Window { u32x8 d; uint32_t sh; };
__device__ __forceinline__ Window win_load(const u32x4 &rs, uint32_t q)
{
    Window w;
    const uint32_t q4 = q & ~3u;
    w.d = sc_load32(rs, q4 ? q4 - 4u : 0u);
    w.sh = (q & 3u) * 8u;
    return w;
}
// after the wait: bring a window loaded for a position below 4 into the common layout (`before` then holds [0, q) in its top bytes)
__device__ __forceinline__ void win_ready(Window &w, uint32_t q)
{
    if (q < 4) { // rare (the block's first probes, empty table slots): a real branch, not seven conditional moves on every window
        asm volatile("" : "+s"(w.d));
        w.d[7] = w.d[6]; w.d[6] = w.d[5]; w.d[5] = w.d[4]; w.d[4] = w.d[3]; w.d[3] = w.d[2]; w.d[2] = w.d[1]; w.d[1] = w.d[0]; w.d[0] = 0;
    }
}
__device__ __forceinline__ uint32_t fun32(uint32_t lo, uint32_t hi, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
__device__ __forceinline__ uint32_t win_before(const Window &w) { return fun32(w.d[0], w.d[1], w.sh); }
__device__ __forceinline__ uint32_t win_at(const Window &w) { return fun32(w.d[1], w.d[2], w.sh); }
__device__ __forceinline__ uint64_t win_after(const Window &w)
{
    return (uint64_t)fun32(w.d[2], w.d[3], w.sh) | ((uint64_t)fun32(w.d[3], w.d[4], w.sh) << 32);
}
__device__ __forceinline__ uint64_t win_after2(const Window &w)
{
    return (uint64_t)fun32(w.d[4], w.d[5], w.sh) | ((uint64_t)fun32(w.d[5], w.d[6], w.sh) << 32);
}
// 4 bytes at q - 2 of the window loaded for q (q >= 2)
__device__ __forceinline__ uint32_t win_at_m2(const Window &w)
{
    // byte offset of q-2 inside the window: 2 + (q & 3) = 2..5
    const uint32_t s2 = w.sh + 16u;
    return s2 < 32u ? fun32(w.d[0], w.d[1], s2) : fun32(w.d[1], w.d[2], s2 - 32u);
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

__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(CW_VT_COMPILER_VGPRS)))
lz4_vtab_kernel(const uint8_t *__restrict__ src, uint32_t n, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                uint32_t *__restrict__ sizes, const uint32_t *__restrict__ queue, uint32_t *__restrict__ counters, uint32_t min_queued,
                uint32_t max_queued, uint32_t reserve)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t qcount = __builtin_amdgcn_readfirstlane(counters[1]);
    if (qcount < min_queued || qcount >= max_queued) return; // the launch policy's regime test, on the device: the queue's length is only known here
    const uint32_t mflimit = n - kMFLimit, matchlimit = n - kLastLiterals;

    for (;;) {
        // reserve > 0: other parsers pull from the same queue; stop pulling while `reserve` blocks are left (as the lane kernels do)
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
            rs.z = n;           // num_records (bytes; stride 0): dwords beyond read as zero
            rs.w = 0x00020000u; // raw buffer, 32-bit elements (gfx9 family)
        }
        vt_zero();

        uint32_t anchor = 0, op = 0, ip = 1;
        // a pending literal copy: bytes loaded one sequence ago, stored now (so the store never waits for its load)
        uint32_t pend_val = 0, pend_pos = 0, pend_cnt = 0;

        if (n >= kMFLimit + 1) {
            Window wp = win_load(rs, ip);
            sc_wait(wp.d);
            win_ready(wp, ip);
            for (;;) { // one sequence per iteration
                // ---- search: probe ip, ip+1, ... with a stride that grows every 64 misses ----
                uint32_t cur, cand;
                Window wc;
                bool found = false;
                {
                    uint32_t fip = ip, step = 1, nb = 64;
                    for (;;) {
                        cur = fip;
                        fip += step;
                        step = nb++ >> 6;
                        if (fip > mflimit + 1) break;
                        const uint32_t v = win_at(wp);
                        cand = vt_exchange(hash13(v), cur);
                        wc = win_load(rs, cand);
                        Window wn = win_load(rs, fip);
                        sc_wait(wc.d, wn.d);
                        win_ready(wc, cand);
                        win_ready(wn, fip);
                        if (win_at(wc) == v) { found = true; break; }
                        wp = wn;
                    }
                }
                if (!found) break; // -> last literals
                // ---- extend backwards over the pending literals (the windows hold 4 bytes; longer: the byte loop) ----
                uint32_t back = 0;
                {
                    const uint32_t room = cur - anchor < cand ? cur - anchor : cand;
                    if (room) {
                        const uint32_t y = win_before(wp) ^ win_before(wc);
                        back = y ? (uint32_t)__builtin_clz(y) >> 3 : 4u;
                        if (back >= room) back = room;
                        else if (back == 4) {
                            for (;;) {
                                const uint32_t j = back + lane + 1;
                                const bool ok = j <= room && g[cur - j] == g[cand - j];
                                const uint32_t cnt = ctz64(~__ballot(ok));
                                back += cnt;
                                if (cnt < 64) break;
                            }
                        }
                    }
                }
                uint32_t lit = cur - back - anchor;
                for (;;) { // next_match: entered again when the re-test after a match hits (no literals, no catch-up)
                    // ---- forward extension from cur + 4 ----
                    uint32_t mc;
                    {
                        const uint32_t lim = matchlimit - (cur + kMinMatch);
                        const uint64_t x = win_after(wp) ^ win_after(wc);
                        mc = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
                        if (mc == 8 && lim > 8) {
                            const uint64_t x2 = win_after2(wp) ^ win_after2(wc);
                            mc += x2 ? (uint32_t)__builtin_ctzll(x2) >> 3 : 8u;
                            if (mc == 16 && lim > 16) {
                                for (;;) {
                                    const uint32_t i = cur + kMinMatch + mc + lane;
                                    const bool ok = i < matchlimit && g[i] == g[cand + kMinMatch + mc + lane];
                                    const uint32_t cnt = ctz64(~__ballot(ok));
                                    mc += cnt;
                                    if (cnt < 64) break;
                                }
                            }
                        }
                        if (mc > lim) mc = lim;
                    }
                    const uint32_t mend = cur + kMinMatch + mc; // first byte after the match
                    const uint32_t off = cur - cand;
                    mc += back;
                    // the next search's window now: the emission below does not wait for it
                    const bool more = mend <= mflimit;
                    Window wnext;
                    if (more) wnext = win_load(rs, mend);

                    // ---- emit: token, literals [anchor, anchor + lit), offset, match length ----
                    if (pend_cnt) { // the previous sequence's literals (loaded an iteration ago)
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
                    if (lane < 3) { // token and the two offset bytes: three lanes, one store instruction
                        const uint32_t where = lane == 0 ? tok_pos : off_pos + lane - 1;
                        const uint32_t what = lane == 0 ? token : lane == 1 ? off : off >> 8;
                        out[where] = (uint8_t)what;
                    }
                    anchor = mend;
                    ip = mend;
                    if (!more) break;

                    // ---- table: insert ip - 2, then the immediate re-test at ip ----
                    sc_wait(wnext.d);
                    wp = wnext;
                    vt_exchange(hash13(win_at_m2(wp)), ip - 2);
                    const uint32_t v = win_at(wp);
                    cand = vt_exchange(hash13(v), ip);
                    wc = win_load(rs, cand);
                    Window wn = win_load(rs, ip + 1);
                    sc_wait(wc.d, wn.d);
                    win_ready(wc, cand);
                    if (win_at(wc) == v) { cur = ip; back = 0; lit = 0; continue; }
                    wp = wn;
                    ip += 1;
                    break;
                }
                if (anchor > mflimit) break; // end of parse: the remaining bytes are literals
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
// Second generation: a batch of K items per memory round trip, the per-item arithmetic on the vector lanes.
//
// The first kernel above is bound by SCALAR instruction issue (one scalar ALU per CU, shared by its four SIMDs): ~210 scalar
// instructions per sequence on text, 16 wavefronts per CU, 18 GB/s; 4 / 8 / 12 / 16 wavefronts per CU give 9.7 / 14.1 / 16.1 / 18.4 GB/s.
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
    const uint32_t r = h >> 7, l = (h >> 1) & 63u, sh = (h & 1u) << 4;
    uint32_t w;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "v_readlane_b32 %0, " CW_VT_BASE ", %2\n\t"
                 "s_set_gpr_idx_off"
                 : "=s"(w) : "s"(r), "s"(l) : CW_VT_CLOBBER);
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
    const uint32_t r = h >> 7, l = (h >> 1) & 63u;
    uint32_t old, w, t, ps, m, nw;
    asm volatile("s_set_gpr_idx_on %[r], gpr_idx(SRC0)\n\t"
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
                 : [old] "=s"(old), [w] "=&s"(w), [t] "=&v"(t), [ps] "=&v"(ps), [m] "=&v"(m), [nw] "=&v"(nw)
                 : [r] "s"(r), [l] "s"(l), [h] "s"(h), [pos] "s"(pos), [kf] "v"(k_ffff)
                 : "scc", CW_VT_CLOBBER);
    return old;
}
// entry h <- pos (the previous entry is of no interest: the insert in front of a re-test)
__device__ __forceinline__ void vt3_put(uint32_t h, uint32_t pos, uint32_t k_ffff)
{
    const uint32_t r = h >> 7, l = (h >> 1) & 63u;
    uint32_t w, t, ps, m, nw;
    asm volatile("s_set_gpr_idx_on %[r], gpr_idx(SRC0)\n\t"
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
                 : [w] "=&s"(w), [t] "=&v"(t), [ps] "=&v"(ps), [m] "=&v"(m), [nw] "=&v"(nw)
                 : [r] "s"(r), [l] "s"(l), [h] "s"(h), [pos] "s"(pos), [kf] "v"(k_ffff)
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
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 sc_load8(const u32x4 &rs, uint32_t off)
{
    u32x2 v;
    asm volatile("s_buffer_load_dwordx2 %0, %1, %2" : "=s"(v) : "s"(rs), "s"(off));
    return v;
}
__device__ __forceinline__ uint32_t sc_load4(const u32x4 &rs, uint32_t off)
{
    uint32_t v;
    asm volatile("s_buffer_load_dword %0, %1, %2" : "=s"(v) : "s"(rs), "s"(off));
    return v;
}
__device__ __forceinline__ void sc_wait3(u32x8 &a, u32x8 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
__device__ __forceinline__ void sc_wait3(u32x8 &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
__device__ __forceinline__ void sc_wait3(u32x8 &a, u32x2 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
__device__ __forceinline__ void sc_wait3(uint32_t &a, uint32_t &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
// a wave-uniform value moved to a vector register: what is computed from it runs on the VALU
__device__ __forceinline__ uint32_t to_v(uint32_t s)
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
// 4 bytes at bit offset sh (0, 8, 16 or 24) of the 8 bytes lo | hi << 32, on the scalar unit (lo, hi: an aligned register pair)
__device__ __forceinline__ uint32_t cut32(uint32_t lo, uint32_t hi, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
// 8 bytes at bit offset sh of the 12 bytes d1 | d2 << 32 | d3 << 64, on the VALU (wave-uniform vector registers)
__device__ __forceinline__ void cut64_v(uint32_t d1, uint32_t d2, uint32_t d3, uint32_t sh, uint32_t &lo, uint32_t &hi)
{
    const uint32_t v1 = to_v(d1), v2 = to_v(d2), v3 = to_v(d3);
    lo = __builtin_amdgcn_alignbit(v2, v1, sh);
    hi = __builtin_amdgcn_alignbit(v3, v2, sh);
}
// a candidate's window: 32 bytes in scalar registers
struct CandWin {
    u32x8 s;
    template <int I> __device__ __forceinline__ uint32_t dw() const { return to_v(s[I]); }
    template <int I> __device__ __forceinline__ void cut64(uint32_t sh, uint32_t &lo, uint32_t &hi) const { cut64_v(s[I], s[I + 1], s[I + 2], sh, lo, hi); }
};
// number of equal low bytes of two 8-byte values (0..8), on the VALU
__device__ __forceinline__ uint32_t equal_bytes_v(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi)
{
    const uint32_t x0 = alo ^ blo, x1 = ahi ^ bhi;
    return x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : 8u;
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
    const uint32_t tab_lds = (uint32_t)reinterpret_cast<uintptr_t>(ltab);
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

        // wave-uniform state in vector registers: anchor, output position; a pending literal copy (per lane: the byte)
        uint32_t anchor = to_v(0), op = to_v(0);
        uint32_t pend_val = 0, pend_pos = to_v(0);
        uint32_t pend_cnt = 0; // scalar

        if (n >= kMFLimit + 1) {
            // scalar state of the search: the probe position cur, its window wp = the 32 bytes from pb4 = cur & ~3 on
            uint32_t cur = 1, pb4 = 0;
            u32x8 wp = sc_load32(rs, 0);
            sc_wait3(wp);
            // one probe: the table exchange, then the candidate's window and the next position's window in ONE round trip
#define CW_VT3_PROBE(FIP)                                                                                                      \
            const uint32_t psh = (cur - pb4) * 8u;                                                                             \
            const uint32_t v = cut32(wp[0], wp[1], psh);                                                                       \
            const uint32_t cand = LDSTAB ? lt3_exchange(tab_lds, (v * 2654435761u) >> 19, cur)                                 \
                                         : vt3_exchange((v * 2654435761u) >> 19, cur, k_ffff);                                 \
            const uint32_t cb4 = cand & ~3u, csh = (cand & 3u) * 8u, fb4 = (FIP) & ~3u;                                         \
            CandWin wc;                                                                                                        \
            u32x8 wq;                                                                                                          \
            asm volatile("s_buffer_load_dwordx8 %0, %2, %3\n\ts_buffer_load_dwordx8 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"      \
                         : "=&s"(wc.s), "=&s"(wq) : "s"(rs), "s"(cb4), "s"(fb4));                                              \
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
                    const uint32_t vcur = to_v(mcur), vcand = to_v(mcand);
                    uint32_t back = to_v(0);
                    if (first) {
                        // bytes the match may move back over: pending literals, and the candidate must stay >= 0
                        const uint32_t room = min(vcur - anchor, vcand);
                        if (__builtin_amdgcn_readfirstlane(room) != 0) {
                            // the 4 bytes in front of both positions (an offset below zero reads as zero; room < 4 then anyway)
                            uint32_t fp = sc_load4(rs, pb4 - 4u), fc = sc_load4(rs, mcb4 - 4u);
                            sc_wait3(fp, fc);
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
                    const bool more = mend <= mflimit;
                    // the windows of what follows the match, now: the emission below does not wait for them
                    u32x8 wnext;
                    u32x2 wins;
                    const uint32_t nb4 = mend & ~3u, ib4 = (mend - 2u) & ~3u;
                    if (more) { wnext = sc_load32(rs, nb4); wins = sc_load8(rs, ib4); }

                    // ---- emit (VALU): token, literals [anchor, anchor + lit), offset, match length ----
                    const uint32_t lit = vcur - back - anchor;
                    const uint32_t off = vcur - vcand;
                    const uint32_t mlen = mc + back;
                    if (pend_cnt) { // the previous sequence's literals (loaded a sequence ago)
                        if (lane < pend_cnt) out[pend_pos + lane] = (uint8_t)pend_val;
                        pend_cnt = 0;
                    }
                    const uint32_t tok_pos = op;
                    const uint32_t token = (min(lit, 15u) << 4) | min(mlen, 15u);
                    op += 1;
                    const uint32_t slit = __builtin_amdgcn_readfirstlane(lit);
                    if (slit) {
                        if (slit >= 15) op += put_len(out + __builtin_amdgcn_readfirstlane(op), slit - 15, lane);
                        if (slit <= 64) {
                            pend_val = lane < slit ? g[__builtin_amdgcn_readfirstlane(anchor) + lane] : 0u;
                            pend_pos = op; pend_cnt = slit;
                        } else {
                            copy_run(out + __builtin_amdgcn_readfirstlane(op), g + __builtin_amdgcn_readfirstlane(anchor), slit, lane);
                        }
                        op += lit;
                    }
                    const uint32_t off_pos = op;
                    op += 2;
                    if (__builtin_amdgcn_readfirstlane((uint32_t)(mlen >= 15)))
                        op += put_len(out + __builtin_amdgcn_readfirstlane(op), __builtin_amdgcn_readfirstlane(mlen) - 15, lane);
                    if (lane < 3) { // token and the two offset bytes: three lanes, one store instruction
                        const uint32_t where = lane == 0 ? tok_pos : off_pos + lane - 1;
                        const uint32_t what = lane == 0 ? token : lane == 1 ? off : off >> 8;
                        out[where] = (uint8_t)what;
                    }
                    anchor = vmend;
                    if (!more) { cur = mend; break; }

                    // ---- table: insert mend - 2, then the immediate re-test at mend ----
                    sc_wait3(wnext, wins);
                    if constexpr (LDSTAB) lt3_put(tab_lds, (cut32(wins[0], wins[1], ((mend - 2u) & 3u) * 8u) * 2654435761u) >> 19, mend - 2u);
                    else vt3_put((cut32(wins[0], wins[1], ((mend - 2u) & 3u) * 8u) * 2654435761u) >> 19, mend - 2u, k_ffff);
                    wp = wnext; pb4 = nb4; cur = mend;
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
        }
        const uint32_t s_anchor = __builtin_amdgcn_readfirstlane(anchor);
        uint32_t s_op = __builtin_amdgcn_readfirstlane(op);
        if (pend_cnt && lane < pend_cnt) out[pend_pos + lane] = (uint8_t)pend_val;

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
    // CW_VTAB_GEN: 1 = the all-scalar kernel, 2 = batches of four items, 3 = the scalar chain with the VALU's help.  Default by measurement
    // (GB/s alone, 16 wavefronts per CU): text, 64 KiB blocks, 8 Ki blocks 18.4 (1) / 13.7 (2) / 20.8 (3), 3,233 blocks 13.6 / 11.3 / 17.1;
    // corpus, 4 KiB blocks 15.4-16.0 (1) against 17.8-17.9 (2)
    const char *gen_env = tune("CW_VTAB_GEN");
    const int gen = gen_env ? atoi(gen_env) : n <= 4096 ? 2 : 3;
    if ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) return hipErrorInvalidValue; // the scalar loads are dword loads
    size_t grid = 256 * (size_t)(waves_per_cu ? waves_per_cu : 16);
    if (grid > nblocks) grid = nblocks;
    if (grid == 0) return hipSuccess;
    if (kernel_name) *kernel_name = gen == 3 ? "cw::lz4_vtab3_kernel<false>" : gen == 1 ? "cw::lz4_vtab_kernel" : "cw::lz4_vtab2_kernel";
    if (gen == 3)
        hipLaunchKernelGGL(lz4_vtab3_kernel<false>, dim3((unsigned)grid), dim3(64), 0, stream, src, n, src_stride, dst, dst_stride, sizes, queue, counters,
                           min_queued, max_queued, reserve);
    else if (gen == 1)
        hipLaunchKernelGGL(lz4_vtab_kernel, dim3((unsigned)grid), dim3(64), 0, stream, src, n, src_stride, dst, dst_stride, sizes, queue, counters,
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
