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
                uint32_t reserve)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t qcount = __builtin_amdgcn_readfirstlane(counters[1]);
    if (qcount < min_queued) return; // the launch policy's regime test, on the device: the queue's length is only known here
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

// grid: as many single-wavefront workgroups as the register file admits (128 VGPRs -> 4 per SIMD, 16 per CU), at most one per queued block
hipError_t lz4_vtab_launch(const uint8_t *src, uint32_t n, size_t src_stride, size_t nblocks, uint8_t *dst, size_t dst_stride, uint32_t *sizes,
                           const uint32_t *queue, uint32_t *counters, uint32_t min_queued, uint32_t reserve, unsigned waves_per_cu, hipStream_t stream)
{
    if ((reinterpret_cast<uintptr_t>(src) | src_stride) & 3) return hipErrorInvalidValue; // the scalar loads are dword loads
    size_t grid = 256 * (size_t)(waves_per_cu ? waves_per_cu : 16);
    if (grid > nblocks) grid = nblocks;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_vtab_kernel, dim3((unsigned)grid), dim3(64), 0, stream, src, n, src_stride, dst, dst_stride, sizes, queue, counters,
                       min_queued, reserve);
    return hipGetLastError();
}

} // namespace cw
