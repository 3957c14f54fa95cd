// decompress_kernels.hip -- LZ4-block and LZF decoders for gfx950, one compressed block per wavefront.
//
// The reference decompresses only to time it (LZ4_decompress_safe / lzf_decompress in
// src/compression_perf/src/experiment.cpp:118,256).  Here the decoders are the
// reference-independent verifier of the compressors at full scale: encode -> decode -> compare on the device.
// Format-level decoders (LZ4 block format; LZF stream format, src/compression_perf/include/lzf/lzf.h:83-95),
// bounds-checked like the "safe" variants: a malformed or truncated input gives status 1, never an
// out-of-bounds access -- a size that exceeds the slot (comp_stride) is malformed too, and every length is checked
// against what is left before it is added, so no sum can wrap.
//
// A wavefront first copies its compressed block into LDS (coalesced; parsing token by token from global memory costs
// a dependent ~1 us round trip per token), decodes it into a second LDS buffer (sequences are serial, their byte copies
// are 64 lanes wide; an overlapping match -- offset < length -- is copied in rounds of `offset` bytes) and then streams
// the block out.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "cw_device.h"

namespace cw {

namespace {
__device__ __forceinline__ uint32_t bcast(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// out[op .. op+len) = out[op-off ..): overlapping copy, `off` bytes per round at most
__device__ __forceinline__ void copy_match(uint8_t *out, uint32_t op, uint32_t off, uint32_t len, uint32_t lane)
{
    const uint32_t round = off < 64 ? off : 64;
    for (uint32_t done = 0; done < len; done += round) {
        const uint32_t k = done + lane;
        uint8_t b = 0;
        const bool act = lane < round && k < len;
        if (act) b = out[op + k - off];
        __syncthreads(); // all reads of this round before its writes (and LDS ordering across rounds)
        if (act) out[op + k] = b;
        __syncthreads();
    }
}
} // namespace

// status[i]: 0 = ok and exactly block_bytes produced, 1 = malformed / wrong size (including sizes[i] == 0, LZF's "did not
// fit": such a block was stored raw and there is nothing to decode) or a size larger than the slot.
template <int ALG, bool STAGE_IN> // ALG: 0 = LZ4, 1 = LZF; STAGE_IN: compressed slot copied to LDS first
__global__ void __launch_bounds__(64)
decompress_kernel(const uint8_t *__restrict__ comp, size_t comp_stride, const uint32_t *__restrict__ sizes, size_t nblocks,
                  uint8_t *__restrict__ dst, uint32_t block_bytes, uint32_t *__restrict__ status, uint32_t in_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t out[]; // block_bytes (rounded to 16), then in_cap bytes of input
    uint8_t *cin = out + ((block_bytes + 15u) & ~15u);
    const uint32_t lane = threadIdx.x;
    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const uint8_t *gin = comp + blk * comp_stride;
        const uint32_t n = sizes[blk];
        uint32_t ip = 0, op = 0;
        // inside the slot; a valid slot never exceeds the codec's bound (~66 KB), and below 2^24 bytes no run of 255s can wrap a length
        bool bad = n == 0 || n > comp_stride || n > (1u << 24) || (STAGE_IN && n > in_cap);
        __syncthreads();
        if (STAGE_IN && !bad) {
            if ((reinterpret_cast<uintptr_t>(gin) & 15) == 0) {
                for (uint32_t i = lane; i < n / 16; i += 64) reinterpret_cast<uint4 *>(cin)[i] = reinterpret_cast<const uint4 *>(gin)[i];
                for (uint32_t i = (n & ~15u) + lane; i < n; i += 64) cin[i] = gin[i]; // never read past the slot's bytes
            } else {
                for (uint32_t i = lane; i < n; i += 64) cin[i] = gin[i];
            }
        }
        __syncthreads();
        const uint8_t *in = STAGE_IN ? cin : gin;
        if (ALG == 0) {
            while (!bad) {
                if (ip >= n) { bad = true; break; }
                const uint32_t tok = bcast(in[ip]); ip++;
                uint32_t lit = tok >> 4;
                if (lit == 15) {
                    uint32_t c;
                    do { if (ip >= n) { bad = true; break; } c = bcast(in[ip]); ip++; lit += c; } while (c == 255);
                    if (bad) break;
                }
                if (lit > n - ip || lit > block_bytes - op) { bad = true; break; } // ip <= n, op <= block_bytes: no wrap
                for (uint32_t i = lane; i < lit; i += 64) out[op + i] = in[ip + i];
                ip += lit; op += lit;
                if (ip == n) break; // last sequence: literals only
                if (n - ip < 2) { bad = true; break; }
                const uint32_t off = bcast((uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8)); ip += 2;
                if (off == 0 || off > op) { bad = true; break; }
                uint32_t ml = tok & 15;
                if (ml == 15) {
                    uint32_t c;
                    do { if (ip >= n) { bad = true; break; } c = bcast(in[ip]); ip++; ml += c; } while (c == 255);
                    if (bad) break;
                }
                if (ml > block_bytes || ml + 4 > block_bytes - op) { bad = true; break; }
                ml += 4;
                copy_match(out, op, off, ml, lane);
                op += ml;
            }
        } else {
            while (!bad && ip < n) {
                const uint32_t ctrl = bcast(in[ip]); ip++;
                if (ctrl < 32) {
                    const uint32_t run = ctrl + 1;
                    if (run > n - ip || run > block_bytes - op) { bad = true; break; }
                    if (lane < run) out[op + lane] = in[ip + lane];
                    ip += run; op += run;
                } else {
                    uint32_t len = ctrl >> 5;
                    if (ip >= n) { bad = true; break; }
                    if (len == 7) { len += bcast(in[ip]); ip++; if (ip >= n) { bad = true; break; } }
                    const uint32_t off = (((ctrl & 0x1f) << 8) | bcast(in[ip])) + 1; ip++;
                    len += 2;
                    if (off > op || len > block_bytes - op) { bad = true; break; }
                    copy_match(out, op, off, len, lane);
                    op += len;
                }
            }
        }
        __syncthreads();
        if (op != block_bytes) bad = true;
        if (!bad) {
            uint8_t *d = dst + blk * (size_t)block_bytes;
            for (uint32_t i = lane; i < block_bytes; i += 64) d[i] = out[i];
        }
        if (lane == 0) status[blk] = bad ? 1u : 0u;
    }
}

// ---------------------------------------------------------------------------------------------------
// Lane-per-block decoders for large batches of large blocks.  A 64 KiB block fills the LDS budget of the wavefront decoder
// with its output alone (two blocks per CU) and its token chain is serial: 7.7 / 6.0 GB/s.  Here a LANE decodes a block
// straight from its slot into its place in dst -- the decoder's loop as it stands, 64 per wavefront, bound like the
// lane-per-block parsers by the lines the memory system retires: per sequence one 16-byte window of the compressed stream
// (token, short literal run and offset in one load when they fit), the literals' store, the match's load(s) from the lane's
// own earlier output and its store(s).  A match closer than 16 bytes is copied at a multiple of its offset (the output is
// periodic there), doubling until 16-byte pieces go through.  Same checks as above: nothing is read outside [0, n) of the
// slot or written outside the block.
// ---------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void store_upto16(uint8_t *p, uint64_t a, uint64_t b, uint32_t cnt)
{
    if (cnt & 16) { __builtin_memcpy(p, &a, 8); __builtin_memcpy(p + 8, &b, 8); return; }
    if (cnt & 8) { __builtin_memcpy(p, &a, 8); a = b; p += 8; }
    if (cnt & 4) { const uint32_t t = (uint32_t)a; __builtin_memcpy(p, &t, 4); a >>= 32; p += 4; }
    if (cnt & 2) { const uint16_t t = (uint16_t)a; __builtin_memcpy(p, &t, 2); a >>= 16; p += 2; }
    if (cnt & 1) *p = (uint8_t)a;
}

// d[op .. op+len) = d[op-off ..), the format's overlapping copy; off >= 1, op - off >= 0, op + len <= cap (checked by the caller)
__device__ __forceinline__ void lane_copy_match(uint8_t *d, uint32_t op, uint32_t off, uint32_t len, uint32_t cap)
{
    const uint32_t base = op - off, end = op + len;
    uint32_t dist = off;
    while (op < end) {
        while (dist < 16 && 2 * dist <= op - base) dist *= 2; // any multiple of off that is already written is a period
        const uint32_t left = end - op, piece = left < 16 ? left : 16, cnt = piece < dist ? piece : dist;
        const uint32_t s = op - dist;
        if (s + 16 <= cap) {
            uint64_t a, b;
            __builtin_memcpy(&a, d + s, 8);
            __builtin_memcpy(&b, d + s + 8, 8);
            store_upto16(d + op, a, b, cnt);
        } else {
            for (uint32_t k = 0; k < cnt; k++) d[op + k] = d[s + k];
        }
        op += cnt;
    }
}

// d[op .. op+len) = in[ip ..): literal bytes; ip + len <= n and op + len <= cap checked by the caller
__device__ __forceinline__ void lane_copy_literals(uint8_t *d, uint32_t op, const uint8_t *in, uint32_t ip, uint32_t len, uint32_t n)
{
    uint32_t k = 0;
    for (; k < len; k += 16) {
        const uint32_t cnt = len - k < 16 ? len - k : 16;
        if (ip + k + 16 <= n) {
            uint64_t a, b;
            __builtin_memcpy(&a, in + ip + k, 8);
            __builtin_memcpy(&b, in + ip + k + 8, 8);
            store_upto16(d + op + k, a, b, cnt);
        } else {
            for (uint32_t j = 0; j < cnt; j++) d[op + k + j] = in[ip + k + j];
        }
    }
}
} // namespace

template <int ALG>
__global__ void __launch_bounds__(64)
decompress_lanes_kernel(const uint8_t *__restrict__ comp, size_t comp_stride, const uint32_t *__restrict__ sizes, size_t nblocks,
                        uint8_t *__restrict__ dst, uint32_t block_bytes, uint32_t *__restrict__ status)
{
    const size_t lanes = (size_t)gridDim.x * 64;
    for (size_t blk = (size_t)blockIdx.x * 64 + threadIdx.x; blk < nblocks; blk += lanes) {
        const uint8_t *in = comp + blk * comp_stride;
        uint8_t *d = dst + blk * (size_t)block_bytes;
        const uint32_t n = sizes[blk];
        uint32_t ip = 0, op = 0;
        bool bad = n == 0 || n > comp_stride || n > (1u << 24);
        if (ALG == 0) {
            while (!bad) {
                if (ip >= n) { bad = true; break; }
                // token, a literal run of up to 13 bytes and the offset in one window when the slot has 16 bytes left
                uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                const bool win = ip + 16 <= n;
                if (win) { uint4 q; __builtin_memcpy(&q, in + ip, 16); w0 = q.x; w1 = q.y; w2 = q.z; w3 = q.w; }
                const uint32_t tok = win ? w0 & 0xFFu : in[ip];
                uint32_t lit = tok >> 4, ml = tok & 15;
                uint32_t off;
                if (win && lit <= 13) {
                    if (lit > block_bytes - op) { bad = true; break; }
                    // literals = window bytes [1, 1 + lit)
                    const uint64_t lo = (uint64_t)w1 << 32 | w0, hi = (uint64_t)w3 << 32 | w2;
                    const uint64_t a = lo >> 8 | hi << 56, b = hi >> 8;
                    if (lit) store_upto16(d + op, a, b, lit);
                    op += lit;
                    ip += 1 + lit;
                    if (ip == n) break; // (a window means 16 bytes were left: not the last sequence unless lit == 15.. never here)
                    const uint32_t sh = (1 + lit) * 8; // offset = window bytes [1 + lit, 3 + lit)
                    const uint64_t o = sh < 64 ? (lo >> sh | (sh ? hi << (64 - sh) : 0)) : hi >> (sh - 64);
                    off = (uint32_t)o & 0xFFFFu;
                    ip += 2;
                } else {
                    ip++;
                    if (lit == 15) {
                        uint32_t c;
                        do { if (ip >= n) { bad = true; break; } c = in[ip]; ip++; lit += c; } while (c == 255);
                        if (bad) break;
                    }
                    if (lit > n - ip || lit > block_bytes - op) { bad = true; break; }
                    lane_copy_literals(d, op, in, ip, lit, n);
                    ip += lit; op += lit;
                    if (ip == n) break; // last sequence: literals only
                    if (n - ip < 2) { bad = true; break; }
                    off = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8);
                    ip += 2;
                }
                if (off == 0 || off > op) { bad = true; break; }
                if (ml == 15) {
                    uint32_t c;
                    do { if (ip >= n) { bad = true; break; } c = in[ip]; ip++; ml += c; } while (c == 255);
                    if (bad) break;
                }
                if (ml > block_bytes || ml + 4 > block_bytes - op) { bad = true; break; }
                ml += 4;
                lane_copy_match(d, op, off, ml, block_bytes);
                op += ml;
            }
        } else {
            while (!bad && ip < n) {
                const uint32_t ctrl = in[ip]; ip++;
                if (ctrl < 32) {
                    const uint32_t run = ctrl + 1;
                    if (run > n - ip || run > block_bytes - op) { bad = true; break; }
                    lane_copy_literals(d, op, in, ip, run, n);
                    ip += run; op += run;
                } else {
                    uint32_t len = ctrl >> 5;
                    if (ip >= n) { bad = true; break; }
                    if (len == 7) { len += in[ip]; ip++; if (ip >= n) { bad = true; break; } }
                    const uint32_t off = (((ctrl & 0x1f) << 8) | in[ip]) + 1; ip++;
                    len += 2;
                    if (off > op || len > block_bytes - op) { bad = true; break; }
                    lane_copy_match(d, op, off, len, block_bytes);
                    op += len;
                }
            }
        }
        if (op != block_bytes) bad = true;
        status[blk] = bad ? 1u : 0u;
    }
}

hipError_t decompress_launch(int alg, const uint8_t *comp, size_t comp_stride, const uint32_t *sizes, size_t nblocks, uint8_t *dst,
                             size_t block_bytes, uint32_t *status, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (block_bytes == 0 || block_bytes > 65536) return hipErrorInvalidValue;
    // Large batches: one block per lane (CW_DECODE_LANES=0 off / =N threshold in blocks).  A lane needs ~0.37 (LZ4) / 0.57 (LZF) us
    // per KiB of block whatever the batch, the wavefront decoders run at 75 / 44 / 22 / 8 GB/s for 4 / 8 / 16 / 64 KiB blocks of
    // text: the lanes win from about 128 MiB of blocks on, and never below ~3,000 blocks (measured: 64 KiB x 2,048 6.1 vs 7.9 GB/s,
    // x 4,096 11.1 vs 7.8, x 65,536 97.7 vs 7.9; 4 KiB x 512 Ki 109.6 vs 75.5).  Unlike the wavefront decoder, which assembles a
    // block in LDS and only stores it when it decoded cleanly, a lane writes as it goes: the bytes of a block with status 1 are
    // unspecified either way.
    const char *dl_env = tune("CW_DECODE_LANES");
    const size_t by_bytes = ((size_t)128 << 20) / block_bytes;
    const size_t lane_min = dl_env ? (size_t)atoi(dl_env) : (by_bytes > 4096 ? by_bytes : 4096);
    if (lane_min && nblocks >= lane_min) {
        const char *lw_env = tune("CW_LANES_WPC");
        const size_t wpc = lw_env && atoi(lw_env) > 0 ? (size_t)atoi(lw_env) : 8;
        size_t lgrid = (nblocks + 63) / 64;
        if (lgrid > 256 * wpc) lgrid = 256 * wpc;
        if (alg == 0)
            hipLaunchKernelGGL(decompress_lanes_kernel<0>, dim3((unsigned)lgrid), dim3(64), 0, stream, comp, comp_stride, sizes, nblocks, dst,
                               (uint32_t)block_bytes, status);
        else
            hipLaunchKernelGGL(decompress_lanes_kernel<1>, dim3((unsigned)lgrid), dim3(64), 0, stream, comp, comp_stride, sizes, nblocks, dst,
                               (uint32_t)block_bytes, status);
        return hipGetLastError();
    }
    // LDS: the decoded block, and -- if at least four blocks per CU still fit (blocks up to ~16 KiB: 4 KiB text decodes
    // 2.2x faster, 64 KiB 1.6x slower with it) -- the compressed slot (at most the codec's bound)
    const uint32_t out_bytes = (uint32_t)((block_bytes + 15) & ~(size_t)15);
    uint32_t in_cap = (uint32_t)(((alg == 0 ? block_bytes + block_bytes / 255 + 16 : block_bytes) + 15) & ~(size_t)15);
    const bool stage_in = out_bytes + in_cap <= 40 * 1024;
    if (!stage_in) in_cap = 0;
    const uint32_t lds = out_bytes + in_cap;
    const size_t per_cu = (160u * 1024u) / lds;
    const size_t want = 256 * (per_cu > 16 ? 16 : per_cu ? per_cu : 1), grid = nblocks < want ? nblocks : want;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(decompress_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(decompress_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
#define CW_DECODE(A, S) hipLaunchKernelGGL((decompress_kernel<A, S>), dim3((unsigned)grid), dim3(64), lds, stream, comp, comp_stride, sizes, \
                                           nblocks, dst, (uint32_t)block_bytes, status, in_cap)
    if (alg == 0 && stage_in) CW_DECODE(0, true);
    else if (alg == 0) CW_DECODE(0, false);
    else if (stage_in) CW_DECODE(1, true);
    else CW_DECODE(1, false);
#undef CW_DECODE
    return hipGetLastError();
}

} // namespace cw
