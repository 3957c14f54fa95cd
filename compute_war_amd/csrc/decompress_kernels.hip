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

hipError_t decompress_launch(int alg, const uint8_t *comp, size_t comp_stride, const uint32_t *sizes, size_t nblocks, uint8_t *dst,
                             size_t block_bytes, uint32_t *status, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (block_bytes == 0 || block_bytes > 65536) return hipErrorInvalidValue;
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
