// sha256_kernel.hip -- FIPS 180-4 SHA-256 of every storage block, one block per lane (gfx950).
//
// Replaces doSHA256MBHashing (src/hashandcompress/HashAndCompress.cpp:136-158) and
// HashBlockSHA256 / HashBlockSHA256MB (src/hashing_perf/hash.cpp:28-77).  The reference's "multi-buffer"
// manager runs N independent messages in the lanes of an AVX register; here the 64 lanes of a
// wavefront are the buffers: lane i hashes block i, all lanes in lockstep, pure 32-bit VALU
// (v_alignbit_b32 rotates, v_bfi_b32 choose/majority, v_xor3/v_add3).  Unlike the reference (which
// drops the digests, :151-155) the 32-byte big-endian digests are written out.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cw_device.h"

namespace cw {

__constant__ uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static __device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }
// a ^ b ^ c in one v_bitop3_b32 (truth table 0x96); hipcc emits two v_xor_b32 for the plain expression
static __device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// one 64-byte compression; w[] holds the 16 big-endian message words and is consumed as the rolling schedule
static __device__ __forceinline__ void sha256_compress(uint32_t (&h)[8], uint32_t (&w)[16])
{
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
            const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
        const uint32_t ch = (e & f) | (~e & g);
        const uint32_t t1 = hh + S1 + ch + K256[i] + w[i & 15];
        const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
        const uint32_t mj = (a & b) | (c & (a | b));
        const uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1;
        d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

template <bool ALIGNED16>
static __device__ __forceinline__ void load_chunk(uint32_t (&w)[16], const uint8_t *p)
{
    if (ALIGNED16) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = q[i];
            w[4 * i] = __builtin_bswap32(v.x); w[4 * i + 1] = __builtin_bswap32(v.y);
            w[4 * i + 2] = __builtin_bswap32(v.z); w[4 * i + 3] = __builtin_bswap32(v.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    }
}

// padding chunk `which` (0 or 1) of a message whose last partial chunk holds `rem` bytes at p (cold path)
static __device__ __noinline__ void load_padding(uint32_t (&w)[16], const uint8_t *p, unsigned rem, unsigned which, uint64_t bits, bool two)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) {
            const unsigned idx = which * 64 + 4 * i + b;
            uint32_t byte = 0;
            if (idx < rem) byte = p[idx];
            else if (idx == rem) byte = 0x80;
            v = (v << 8) | byte;
        }
        w[i] = v;
    }
    if (which == (two ? 1u : 0u)) { w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; }
}

// RAGGED=false: block_bytes is a multiple of 64, so the single padding chunk is a wave-uniform constant.
template <bool ALIGNED16, bool RAGGED>
__global__ void __launch_bounds__(CW_SKEIN_THREADS)
sha256_blocks_kernel(const uint8_t *__restrict__ src, size_t block_bytes, size_t src_stride, size_t nblocks,
                     uint8_t *__restrict__ digests)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nblocks) return;
    const uint8_t *p = src + gid * src_stride;
    const size_t nfull = block_bytes / 64;
    const unsigned rem = RAGGED ? (unsigned)(block_bytes % 64) : 0;
    const bool two = rem >= 56;                 // 0x80 + length do not fit the partial chunk
    const size_t nsteps = nfull + 1 + (two ? 1 : 0);
    const uint64_t bits = (uint64_t)block_bytes * 8;

    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint32_t w[16];

    if (RAGGED) {
        auto fetch = [&](uint32_t (&dst)[16], size_t step) __attribute__((always_inline)) {
            if (step < nfull) load_chunk<ALIGNED16>(dst, p + step * 64);
            else load_padding(dst, p + nfull * 64, rem, (unsigned)(step - nfull), bits, two);
        };
        fetch(w, 0);
#pragma unroll 1
        for (size_t step = 0; step < nsteps; step++) {
            uint32_t nx[16];
            if (step + 1 < nsteps) fetch(nx, step + 1);
            sha256_compress(h, w);
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = nx[i];
        }
    } else {
        // Hot path.  The next chunk's loads are issued unconditionally (the padding step re-reads the last chunk and
        // is replaced) and are only touched AFTER this chunk's 64 rounds: a branch around the loads, or a byte swap
        // right behind them, makes hipcc put s_waitcnt vmcnt(0) in front of the rounds and nothing overlaps.
        load_chunk<ALIGNED16>(w, p);
#pragma unroll 1
        for (size_t step = 0; step < nsteps; step++) {
            const size_t nxt = step + 1;
            const bool msg = nxt < nfull; // wave-uniform
            const uint8_t *q = p + (msg ? nxt : nfull - 1) * 64;
            uint32_t raw[16];
            if (ALIGNED16) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint4 v = reinterpret_cast<const uint4 *>(q)[i];
                    raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
                }
            } else {
                load_chunk<false>(raw, q);
            }
            sha256_compress(h, w);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t v = ALIGNED16 ? __builtin_bswap32(raw[i]) : raw[i];
                const uint32_t pad = i == 0 ? 0x80000000u : i == 14 ? (uint32_t)(bits >> 32) : i == 15 ? (uint32_t)bits : 0u;
                w[i] = msg ? v : pad;
            }
        }
    }

    uint8_t *o = digests + gid * 32;
    if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
        uint4 *out = reinterpret_cast<uint4 *>(o);
        out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
        out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
    } else { // any alignment: big-endian bytes
#pragma unroll
        for (int k = 0; k < 32; k++) o[k] = (uint8_t)(h[k >> 2] >> (24 - 8 * (k & 3)));
    }
}

hipError_t sha256_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, uint8_t *digests,
                         hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    const dim3 grid((unsigned)((nblocks + CW_SKEIN_THREADS - 1) / CW_SKEIN_THREADS)), block(CW_SKEIN_THREADS);
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | src_stride) & 15) == 0;
    // (an empty message is "ragged" too: its only chunk is the padding, and the hot path's first load would read 64 bytes that are not there)
    const bool ragged = block_bytes == 0 || (block_bytes % 64) != 0;
#define CW_LAUNCH(A, R) hipLaunchKernelGGL((sha256_blocks_kernel<A, R>), grid, block, 0, stream, src, block_bytes, src_stride, nblocks, digests)
    if (aligned && !ragged) CW_LAUNCH(true, false);
    else if (aligned) CW_LAUNCH(true, true);
    else if (!ragged) CW_LAUNCH(false, false);
    else CW_LAUNCH(false, true);
    note_kernels(1, aligned ? (ragged ? "cw::sha256_blocks_kernel<true, true>" : "cw::sha256_blocks_kernel<true, false>")
                            : (ragged ? "cw::sha256_blocks_kernel<false, true>" : "cw::sha256_blocks_kernel<false, false>"));
#undef CW_LAUNCH
    return hipGetLastError();
}

} // namespace cw
