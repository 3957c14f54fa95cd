// skein_kernels.hip -- Skein-512 / Skein-256 block hashing for gfx950 (CDNA4), one storage block per lane.
//
// Replaces, on the device, what the reference does per block on a CPU thread:
//   Skein_256_Init/Update/Final in doSkeinHashing   (src/hashandcompress/HashAndCompress.cpp:121-134,
//                                                    src/hashing_perf/hash.cpp:5-26)
//   Skein_512_Init/Update/Final                     (reference_code/skein/Optimized_64bit/skein.c:226-408)
//   Skein_{256,512}_Process_Block                   (reference_code/skein/Optimized_64bit/skein_block.c:42-210, :227-418)
//
// Mapping to the machine.  A Skein digest is a strictly serial chain of Threefish calls (1025 of them
// for a 64 KiB block), each with only 4-way instruction parallelism, so the parallel axis is the set of
// independent storage blocks: lane i of a wavefront owns block i and all 64 lanes run in lockstep.  Because
// every lane is at the same byte offset of its own block, the tweak words (bytes-so-far, block-type flags)
// are wave-uniform and live in SGPRs; the chaining value, key schedule and message words are 64-bit
// values in VGPR pairs.  The work is pure 32-bit integer VALU (add/addc, v_alignbit for the rotates, xor):
// about 2.2 k instructions per 64 message bytes, which -- not HBM -- is the roof (DESIGN.md "Rooflines").
// Message bytes are fetched straight from HBM, 64 B (one Threefish-512 block) per lane per step with the
// next step's loads issued before the current step's rounds so their latency hides under ~4 k cycles of ALU.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "cw_device.h"

namespace cw {

// ---- Threefish constants of the 2008 NIST submission the reference vendors (skein.h:275-292) ----
#define KS_PARITY 0x5555555555555555ULL

// 64-bit rotate-left by a compile-time amount as two v_alignbit_b32 on the register halves
// (left to itself hipcc emits v_lshlrev_b64 + v_lshrrev_b64 + 2 v_or for a 64-bit rotate).
template <int R>
static __device__ __forceinline__ uint64_t rotl(uint64_t x)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t nlo, nhi;
    if (R == 32) { nlo = hi; nhi = lo; }
    else if (R < 32) {
        nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
        nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
    } else {
        nlo = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
        nhi = __builtin_amdgcn_alignbit(lo, hi, 64 - R);
    }
    return ((uint64_t)nhi << 32) | nlo;
}

// 64-bit add.  CW_ADD64_PAIR=1 forces v_add_co_u32 + v_addc_co_u32; 0 lets hipcc pick (v_lshl_add_u64 on gfx950).
#ifndef CW_ADD64_PAIR
#define CW_ADD64_PAIR 0
#endif

static __device__ __forceinline__ uint64_t add64(uint64_t a, uint64_t b)
{
#if CW_ADD64_PAIR
    uint32_t lo, hi;
    asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
        : "=&v"(lo), "=v"(hi)
        : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32))
        : "vcc");
    return ((uint64_t)hi << 32) | lo;
#else
    return a + b;
#endif
}

#define MIX(a, b, r) do { a = add64(a, b); b = rotl<r>(b) ^ a; } while (0)

// ---------------------------------------------------------------------------------------------
// Threefish-512 in UBI/MMO form: X <- E_{X,tweak}(w) ^ w.   ks/ts follow skein_block.c:264-278.
// The word permutation of each round is folded into operand order exactly as the cipher defines it:
// round d uses pairs (p0,p1) (p2,p3) (p4,p5) (p6,p7) of the permuted state.
// ---------------------------------------------------------------------------------------------
#define R512(X, a0, a1, a2, a3, a4, a5, a6, a7, r0, r1, r2, r3) \
    MIX(X[a0], X[a1], r0); MIX(X[a2], X[a3], r1); MIX(X[a4], X[a5], r2); MIX(X[a6], X[a7], r3)

#define INJECT512(X, ks, ts, s)                                              \
    X[0] = add64(X[0], ks[((s) + 0) % 9]); X[1] = add64(X[1], ks[((s) + 1) % 9]); \
    X[2] = add64(X[2], ks[((s) + 2) % 9]); X[3] = add64(X[3], ks[((s) + 3) % 9]); \
    X[4] = add64(X[4], ks[((s) + 4) % 9]);                                   \
    X[5] = add64(add64(X[5], ks[((s) + 5) % 9]), ts[(s) % 3]);               \
    X[6] = add64(add64(X[6], ks[((s) + 6) % 9]), ts[((s) + 1) % 3]);         \
    X[7] = add64(add64(X[7], ks[((s) + 7) % 9]), (uint64_t)(s))

#define EIGHT_ROUNDS_512(X, ks, ts, q)                                       \
    R512(X, 0, 1, 2, 3, 4, 5, 6, 7, 38, 30, 50, 53);                         \
    R512(X, 2, 1, 4, 7, 6, 5, 0, 3, 48, 20, 43, 31);                         \
    R512(X, 4, 1, 6, 3, 0, 5, 2, 7, 34, 14, 15, 27);                         \
    R512(X, 6, 1, 0, 7, 2, 5, 4, 3, 26, 12, 58, 7);                          \
    INJECT512(X, ks, ts, 2 * (q) + 1);                                       \
    R512(X, 0, 1, 2, 3, 4, 5, 6, 7, 33, 49, 8, 42);                          \
    R512(X, 2, 1, 4, 7, 6, 5, 0, 3, 39, 27, 41, 14);                         \
    R512(X, 4, 1, 6, 3, 0, 5, 2, 7, 29, 26, 11, 9);                          \
    R512(X, 6, 1, 0, 7, 2, 5, 4, 3, 33, 51, 39, 35);                         \
    INJECT512(X, ks, ts, 2 * (q) + 2)

static __device__ __forceinline__ void ubi512(uint64_t (&chain)[8], const uint64_t (&w)[8], uint64_t t0, uint64_t t1)
{
    uint64_t ks[9], ts[3], X[8];
    ks[8] = KS_PARITY;
#pragma unroll
    for (int i = 0; i < 8; i++) { ks[i] = chain[i]; ks[8] ^= chain[i]; }
    ts[0] = t0; ts[1] = t1; ts[2] = t0 ^ t1;
#pragma unroll
    for (int i = 0; i < 8; i++) X[i] = add64(w[i], ks[i]);
    X[5] = add64(X[5], ts[0]);
    X[6] = add64(X[6], ts[1]);
    EIGHT_ROUNDS_512(X, ks, ts, 0); EIGHT_ROUNDS_512(X, ks, ts, 1); EIGHT_ROUNDS_512(X, ks, ts, 2);
    EIGHT_ROUNDS_512(X, ks, ts, 3); EIGHT_ROUNDS_512(X, ks, ts, 4); EIGHT_ROUNDS_512(X, ks, ts, 5);
    EIGHT_ROUNDS_512(X, ks, ts, 6); EIGHT_ROUNDS_512(X, ks, ts, 7); EIGHT_ROUNDS_512(X, ks, ts, 8);
#pragma unroll
    for (int i = 0; i < 8; i++) chain[i] = X[i] ^ w[i];
}

// ---- Threefish-256 (skein_block.c:42-210; rotations skein.h:275-282) ----
#define R256(X, a0, a1, a2, a3, r0, r1) MIX(X[a0], X[a1], r0); MIX(X[a2], X[a3], r1)
#define INJECT256(X, ks, ts, s)                                              \
    X[0] = add64(X[0], ks[((s) + 0) % 5]);                                   \
    X[1] = add64(add64(X[1], ks[((s) + 1) % 5]), ts[(s) % 3]);               \
    X[2] = add64(add64(X[2], ks[((s) + 2) % 5]), ts[((s) + 1) % 3]);         \
    X[3] = add64(add64(X[3], ks[((s) + 3) % 5]), (uint64_t)(s))
#define EIGHT_ROUNDS_256(X, ks, ts, q)                                       \
    R256(X, 0, 1, 2, 3, 5, 56);  R256(X, 0, 3, 2, 1, 36, 28);                \
    R256(X, 0, 1, 2, 3, 13, 46); R256(X, 0, 3, 2, 1, 58, 44);                \
    INJECT256(X, ks, ts, 2 * (q) + 1);                                       \
    R256(X, 0, 1, 2, 3, 26, 20); R256(X, 0, 3, 2, 1, 53, 35);                \
    R256(X, 0, 1, 2, 3, 11, 42); R256(X, 0, 3, 2, 1, 59, 50);                \
    INJECT256(X, ks, ts, 2 * (q) + 2)

static __device__ __forceinline__ void ubi256(uint64_t (&chain)[4], const uint64_t (&w)[4], uint64_t t0, uint64_t t1)
{
    uint64_t ks[5], ts[3], X[4];
    ks[4] = KS_PARITY;
#pragma unroll
    for (int i = 0; i < 4; i++) { ks[i] = chain[i]; ks[4] ^= chain[i]; }
    ts[0] = t0; ts[1] = t1; ts[2] = t0 ^ t1;
#pragma unroll
    for (int i = 0; i < 4; i++) X[i] = add64(w[i], ks[i]);
    X[1] = add64(X[1], ts[0]);
    X[2] = add64(X[2], ts[1]);
    EIGHT_ROUNDS_256(X, ks, ts, 0); EIGHT_ROUNDS_256(X, ks, ts, 1); EIGHT_ROUNDS_256(X, ks, ts, 2);
    EIGHT_ROUNDS_256(X, ks, ts, 3); EIGHT_ROUNDS_256(X, ks, ts, 4); EIGHT_ROUNDS_256(X, ks, ts, 5);
    EIGHT_ROUNDS_256(X, ks, ts, 6); EIGHT_ROUNDS_256(X, ks, ts, 7); EIGHT_ROUNDS_256(X, ks, ts, 8);
#pragma unroll
    for (int i = 0; i < 4; i++) chain[i] = X[i] ^ w[i];
}

template <int NW> struct Ubi;
template <> struct Ubi<8> {
    static __device__ __forceinline__ void run(uint64_t (&c)[8], const uint64_t (&w)[8], uint64_t t0, uint64_t t1) { ubi512(c, w, t0, t1); }
};
template <> struct Ubi<4> {
    static __device__ __forceinline__ void run(uint64_t (&c)[4], const uint64_t (&w)[4], uint64_t t0, uint64_t t1) { ubi256(c, w, t0, t1); }
};

// tweak T1 fields (skein.h:146-186)
#define T1_FIRST (1ULL << 62)
#define T1_FINAL (1ULL << 63)
#define T1_MSG   (48ULL << 56)
#define T1_OUT   (63ULL << 56)

// ---- message fetch: NW little-endian u64 words at p ----
template <int NW, bool ALIGNED16>
static __device__ __forceinline__ void load_words(uint64_t (&w)[NW], const uint8_t *p)
{
    if (ALIGNED16) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int i = 0; i < NW / 2; i++) {
            uint4 v = q[i];
            w[2 * i] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            w[2 * i + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NW; i++) {
            uint64_t v = 0;
#pragma unroll
            for (int b = 7; b >= 0; b--) v = (v << 8) | p[8 * i + b];
            w[i] = v;
        }
    }
}

// the last (1..NW*8-1 byte, or empty-message 0 byte) chunk, zero padded (skein.c:383-384); cold path
template <int NW>
static __device__ __noinline__ void load_tail(uint64_t (&w)[NW], const uint8_t *p, unsigned rem)
{
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; b--) {
            const unsigned idx = 8 * i + b;
            const uint64_t byte = (idx < rem) ? p[idx] : 0;
            v = (v << 8) | byte;
        }
        w[i] = v;
    }
}

// One storage block per lane.  digest_bytes <= NW*8 (one output block), iv = config-block UBI result.
// The message steps, the final step and the output transform all go through ONE Threefish body inside
// one loop (the step kind only changes wave-uniform tweak values), which keeps the unrolled 72-round
// body -- ~17 KiB of code -- single in the instruction cache and the register allocation tight.
// RAGGED=false is the hot instantiation: block_bytes is a positive multiple of the Threefish block, so
// no byte-granular tail code (and no scratch) exists in it.
template <int NW, bool ALIGNED16, bool RAGGED>
__global__ void __launch_bounds__(CW_SKEIN_THREADS)
skein_blocks_kernel(const uint8_t *__restrict__ src, size_t block_bytes, size_t src_stride, size_t nblocks,
                    SkeinIV iv, uint8_t *__restrict__ digests, unsigned digest_bytes)
{
    constexpr unsigned BB = NW * 8;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nblocks) return;
    const uint8_t *p = src + gid * src_stride;

    uint64_t X[NW], w[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) X[i] = iv.w[i];

    // Update(): all Threefish blocks but the last, which is held back so FINAL lands on data (skein.c:356)
    const size_t nfull = block_bytes ? (block_bytes - 1) / BB : 0;
    const unsigned rem = RAGGED ? (unsigned)(block_bytes - nfull * BB) : BB; // 1..BB, or 0 for an empty message
    const size_t nsteps = nfull + 2;                           // + Final() + output transform
    uint64_t t0 = 0, t1 = T1_FIRST | T1_MSG;

    if (!RAGGED || nfull || rem == BB) load_words<NW, ALIGNED16>(w, p);
    else load_tail<NW>(w, p, rem);

#pragma unroll 1
    for (size_t step = 0; step < nsteps; step++) {
        uint64_t nx[NW];
        // issue the NEXT step's message loads before this step's 72 rounds.  In the hot (!RAGGED) instantiation the
        // load is unconditional -- past the message it re-reads the last step and is masked to zero -- because hipcc
        // answers loads inside a branch with a vmcnt(0) right behind them, which would serialise fetch and rounds.
        if (!RAGGED) {
            const size_t nxt_step = step + 1;
            const uint64_t keep = nxt_step <= nfull ? ~0ull : 0ull; // steps 0..nfull carry message bytes
            load_words<NW, ALIGNED16>(nx, p + (nxt_step <= nfull ? nxt_step : nfull) * BB);
#pragma unroll
            for (int k = 0; k < NW; k++) nx[k] &= keep;
        } else if (step + 1 < nfull || (step + 1 == nfull && rem == BB)) {
            load_words<NW, ALIGNED16>(nx, p + (step + 1) * BB);
        } else if (step + 1 == nfull) {
            load_tail<NW>(nx, p + (step + 1) * BB, rem);
        } else {
#pragma unroll
            for (int k = 0; k < NW; k++) nx[k] = 0; // output block: counter 0 (skein.c:396)
        }
        if (step < nfull) {
            t0 += BB;
        } else if (step == nfull) {               // Final(): T0 += bytes present, FINAL flag (skein.c:381-386)
            t0 += rem;
            t1 |= T1_FINAL;
        } else {                                  // output transform (skein.c:391-405)
            t0 = 8;
            t1 = T1_FIRST | T1_FINAL | T1_OUT;
        }
        Ubi<NW>::run(X, w, t0, t1);
        t1 &= ~T1_FIRST;
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = nx[k];
    }

    uint8_t *out = digests + gid * digest_bytes;
    if (((digest_bytes | (unsigned)reinterpret_cast<uintptr_t>(out)) & 15) == 0) { // else: any alignment, byte stores
        uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
        for (int k = 0; k < NW / 2; k++)
            if ((unsigned)(16 * k) < digest_bytes)
                o4[k] = make_uint4((uint32_t)X[2 * k], (uint32_t)(X[2 * k] >> 32), (uint32_t)X[2 * k + 1], (uint32_t)(X[2 * k + 1] >> 32));
    } else {
        for (unsigned k = 0; k < digest_bytes; k++) out[k] = (uint8_t)(X[k >> 3] >> (8 * (k & 7)));
    }
}

// ---------------------------------------------------------------------------------------------------
// Hot instantiation for block sizes that are a positive multiple of the Threefish block (4096, 65536 ...).
// Each lane fetches its message one 128-byte cache line at a time (8 x global_load_dwordx4 issued back to
// back = 2 Threefish-512 steps or 4 Threefish-256 steps) so that every line crosses the memory system
// exactly once -- fetching 64 B per step let ~40 % of the lines fall out of L2 between their two halves
// (rocprofv3 FETCH_SIZE, profiles/).  Register budget: the line is held as two halves A and B plus one
// spare half S.  When the steps of half A are done its registers are free, so the NEXT line is requested
// right then -- first half into A, second half into S -- and lands while the steps of half B run; B <- S
// afterwards.  48 message VGPRs instead of 64 keeps the kernel at ~100 VGPRs, which matters when codec
// wavefronts share the SIMDs (4 hash wavefronts per SIMD still fit beside them).
// ---------------------------------------------------------------------------------------------------
template <int NW, bool ALIGNED16>
__global__ void __launch_bounds__(CW_SKEIN_THREADS)
skein_lines_kernel(const uint8_t *__restrict__ src, size_t block_bytes, size_t src_stride, size_t nblocks,
                   SkeinIV iv, uint8_t *__restrict__ digests, unsigned digest_bytes)
{
    constexpr unsigned BB = NW * 8, SPL = 128 / BB, HS = SPL / 2; // steps per line / per half line
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nblocks) return;
    const uint8_t *p = src + gid * src_stride;

    uint64_t X[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) X[i] = iv.w[i];

    const size_t nmsg = block_bytes / BB; // message steps; the last one carries FINAL (skein.c:356,381-386)
    const size_t total = nmsg + 1;        // + output transform (skein.c:391-405), whose "message" is counter 0
    uint64_t t0 = 0, t1 = T1_FIRST | T1_MSG;

    uint64_t A[HS][NW], B[HS][NW], S[HS][NW];
    // steps [first, first+HS) of the message into one half-line buffer; steps past the message read as zero.
    // The loads are UNCONDITIONAL (a step past the message re-reads the last step and is masked to zero): with the
    // loads inside a branch hipcc cannot count how many are outstanding and waits for all of them (vmcnt(0)) right
    // at the top of the loop, i.e. the prefetch would not overlap the rounds at all.
    auto fetch_half = [&](uint64_t (&dst)[HS][NW], size_t first) __attribute__((always_inline)) {
#pragma unroll
        for (unsigned j = 0; j < HS; j++) {
            const size_t s = first + j;
            const uint64_t keep = s < nmsg ? ~0ull : 0ull; // wave-uniform
            load_words<NW, ALIGNED16>(dst[j], p + (s < nmsg ? s : nmsg - 1) * BB);
#pragma unroll
            for (int k = 0; k < NW; k++) dst[j][k] &= keep;
        }
    };
    auto run_half = [&](uint64_t (&buf)[HS][NW], size_t first) {
#pragma unroll
        for (unsigned j = 0; j < HS; j++) {
            const size_t s = first + j;
            if (s < total) {
                if (s + 1 < nmsg) {
                    t0 += BB;
                } else if (s + 1 == nmsg) {
                    t0 += BB;
                    t1 |= T1_FINAL;
                } else {
                    t0 = 8;
                    t1 = T1_FIRST | T1_FINAL | T1_OUT;
                }
                Ubi<NW>::run(X, buf[j], t0, t1);
                t1 &= ~T1_FIRST;
            }
        }
    };
    fetch_half(A, 0);
    fetch_half(B, HS);

#pragma unroll 1
    for (size_t first = 0; first < total; first += SPL) {
        run_half(A, first);
        fetch_half(A, first + SPL);      // the whole next line, requested together:
        fetch_half(S, first + SPL + HS); //   first half into the freed A, second half into the spare
        run_half(B, first + HS);
#pragma unroll
        for (unsigned j = 0; j < HS; j++)
#pragma unroll
            for (int k = 0; k < NW; k++) B[j][k] = S[j][k];
    }

    uint8_t *out = digests + gid * digest_bytes;
    if (((digest_bytes | (unsigned)reinterpret_cast<uintptr_t>(out)) & 15) == 0) { // else: any alignment, byte stores
        uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
        for (int k = 0; k < NW / 2; k++)
            if ((unsigned)(16 * k) < digest_bytes)
                o4[k] = make_uint4((uint32_t)X[2 * k], (uint32_t)(X[2 * k] >> 32), (uint32_t)X[2 * k + 1], (uint32_t)(X[2 * k + 1] >> 32));
    } else {
        for (unsigned k = 0; k < digest_bytes; k++) out[k] = (uint8_t)(X[k >> 3] >> (8 * (k & 7)));
    }
}

// ---------------------------------------------------------------------------------------------------
// Sliced launches: the steps of a block are cut into kSkeinSlices launches of short-lived wavefronts
// that hand the chaining values on through a state array (kernel boundaries order and publish it: no flags, no
// spinning -- a persistent-grid version with in-kernel dependencies lost to the oldest-first issue order).
// Beside the codec this is 61.6 instead of 64.5 ms per Mi blocks of 64 KiB: the scan then finishes after 33 instead of
// 53 ms and the hash has the chip to itself for the rest.  The number of slices hardly matters (2..64: 61.6-63.0 ms);
// splitting the blocks over two streams to fill each launch's tail did not help.  Without a codec beside it the sliced
// hash is as fast as the one-launch line kernel (46.6 ms per Mi blocks, a little ahead on small batches), so it is used
// there too; the line kernel serves batches below 4,096 blocks and short messages.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kSkeinSlices = 8;

#ifdef CW_CLOCK_STAMP
__device__ unsigned long long g_clock_skein[4 * kClockSlots];
hipError_t skein_clock_read(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_skein), sizeof g_clock_skein); }
#endif

template <int NW, bool ALIGNED16>
__global__ void __launch_bounds__(CW_SKEIN_THREADS)
skein_slice_kernel(const uint8_t *__restrict__ src, size_t block_bytes, size_t src_stride, size_t nblocks, SkeinIV iv,
                   uint8_t *__restrict__ digests, unsigned digest_bytes, uint64_t *__restrict__ state, size_t s_begin, size_t s_end)
{
    CW_CLOCK_SCOPE_KEYED(g_clock_skein, s_begin / (s_end - s_begin ? s_end - s_begin : 1)); // (the launch's index within its pass)
    constexpr unsigned BB = NW * 8, SPL = 128 / BB, HS = SPL / 2;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nblocks) return;
    const uint8_t *p = src + gid * src_stride;
    const size_t nmsg = block_bytes / BB, total = nmsg + 1;
    uint64_t *st = state + gid * NW;

    uint64_t X[NW];
    if (s_begin == 0) {
#pragma unroll
        for (int i = 0; i < NW; i++) X[i] = iv.w[i];
    } else {
#pragma unroll
        for (int i = 0; i < NW; i++) X[i] = st[i];
    }
    uint64_t t0 = (uint64_t)s_begin * BB, t1 = (s_begin == 0 ? T1_FIRST : 0) | T1_MSG;

    uint64_t A[HS][NW], B[HS][NW], S[HS][NW];
    auto fetch_half = [&](uint64_t (&dst)[HS][NW], size_t first) __attribute__((always_inline)) {
#pragma unroll
        for (unsigned j = 0; j < HS; j++) {
            const size_t s = first + j;
            const uint64_t keep = s < nmsg ? ~0ull : 0ull; // wave-uniform
            load_words<NW, ALIGNED16>(dst[j], p + (s < nmsg ? s : nmsg - 1) * BB);
#pragma unroll
            for (int k = 0; k < NW; k++) dst[j][k] &= keep;
        }
    };
    auto run_half = [&](uint64_t (&buf)[HS][NW], size_t first) {
#pragma unroll
        for (unsigned j = 0; j < HS; j++) {
            const size_t s = first + j;
            if (s < s_end) {
                if (s + 1 < nmsg) {
                    t0 += BB;
                } else if (s + 1 == nmsg) {
                    t0 += BB;
                    t1 |= T1_FINAL;
                } else {
                    t0 = 8;
                    t1 = T1_FIRST | T1_FINAL | T1_OUT;
                }
                Ubi<NW>::run(X, buf[j], t0, t1);
                t1 &= ~T1_FIRST;
            }
        }
    };
    fetch_half(A, s_begin);
    fetch_half(B, s_begin + HS);
#pragma unroll 1
    for (size_t first = s_begin; first < s_end; first += SPL) {
        run_half(A, first);
        fetch_half(A, first + SPL);
        fetch_half(S, first + SPL + HS);
        run_half(B, first + HS);
#pragma unroll
        for (unsigned j = 0; j < HS; j++)
#pragma unroll
            for (int k = 0; k < NW; k++) B[j][k] = S[j][k];
    }

    if (s_end < total) {
#pragma unroll
        for (int i = 0; i < NW; i++) st[i] = X[i];
        return;
    }
    uint8_t *out = digests + gid * digest_bytes;
    if (((digest_bytes | (unsigned)reinterpret_cast<uintptr_t>(out)) & 15) == 0) { // else: any alignment, byte stores
        uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
        for (int k = 0; k < NW / 2; k++)
            if ((unsigned)(16 * k) < digest_bytes)
                o4[k] = make_uint4((uint32_t)X[2 * k], (uint32_t)(X[2 * k] >> 32), (uint32_t)X[2 * k + 1], (uint32_t)(X[2 * k + 1] >> 32));
    } else {
        for (unsigned k = 0; k < digest_bytes; k++) out[k] = (uint8_t)(X[k >> 3] >> (8 * (k & 7)));
    }
}

namespace {
struct SliceSpace { uint64_t *p = nullptr; size_t cap = 0; std::mutex launch; };
std::mutex slice_lock;
std::unordered_map<uint64_t, SliceSpace> slice_map; // references stay valid across inserts
}

void skein_release_workspaces()
{
    std::lock_guard<std::mutex> g(slice_lock);
    for (auto &kv : slice_map) if (kv.second.p) (void)hipFree(kv.second.p);
    slice_map.clear();
}

void skein_release_stream(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(slice_lock);
    auto it = slice_map.find(ws_key(stream));
    if (it == slice_map.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    slice_map.erase(it);
}

bool skein_sliced_applies(int nw, const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const uint8_t *digests)
{
    const size_t bb = (size_t)nw * 8;
    return nblocks >= 4096 && block_bytes % bb == 0 && block_bytes / bb + 1 >= 256 &&
           ((reinterpret_cast<uintptr_t>(src) | src_stride | reinterpret_cast<uintptr_t>(digests)) & 15) == 0;
}

hipError_t skein_sliced_launch(int nw, const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                               uint8_t *digests, unsigned digest_bytes, hipStream_t stream)
{
    const size_t bb = (size_t)nw * 8, total = block_bytes / bb + 1, spl = 128 / bb;
    const char *ns_env = tune("CW_SKEIN_NSLICES"); // profiling knob
    const size_t nsl = ns_env && atoi(ns_env) > 0 ? (size_t)atoi(ns_env) : kSkeinSlices;
    size_t slice_steps = (total + nsl - 1) / nsl;
    slice_steps = (slice_steps + spl - 1) / spl * spl;
    uint64_t *state = nullptr;
    SliceSpace *wsp;
    {
        std::lock_guard<std::mutex> g(slice_lock);
        wsp = &slice_map[ws_key(stream)];
    }
    std::lock_guard<std::mutex> sequence(wsp->launch); // the slices hand their chaining values on through w.p
    {
        SliceSpace &w = *wsp;
        if (w.cap < nblocks * (size_t)nw) {
            if (w.p) { hipError_t e = hipFree(w.p); if (e != hipSuccess) return e; }
            w.p = nullptr; w.cap = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&w.p), nblocks * (size_t)nw * sizeof(uint64_t));
            if (e != hipSuccess) return e;
            w.cap = nblocks * (size_t)nw;
        }
        state = w.p;
    }
    const dim3 grid((unsigned)((nblocks + CW_SKEIN_THREADS - 1) / CW_SKEIN_THREADS)), block(CW_SKEIN_THREADS);
    for (size_t b = 0; b < total; b += slice_steps) {
        const size_t e = b + slice_steps < total ? b + slice_steps : total;
        if (nw == 8)
            hipLaunchKernelGGL((skein_slice_kernel<8, true>), grid, block, 0, stream, src, block_bytes, src_stride, nblocks, iv, digests,
                               digest_bytes, state, b, e);
        else
            hipLaunchKernelGGL((skein_slice_kernel<4, true>), grid, block, 0, stream, src, block_bytes, src_stride, nblocks, iv, digests,
                               digest_bytes, state, b, e);
    }
    note_kernels(1, nw == 8 ? "cw::skein_slice_kernel<8, true>" : "cw::skein_slice_kernel<4, true>");
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Tree hashing (semantics of the reference's Skein_TreeHash, skein_test.c:616-680, restated on the
// CPU in oracle/skein_oracle.c and pinned to the reference's tree KAT vectors).  The digests differ from sequential
// Skein by construction, so this is an extra entry point, not a drop-in: what it buys is parallelism INSIDE a block --
// one wavefront per block, lane i hashes leaf i (then node i of the next level, ...), so a single 64 KiB block takes
// 1/64 of the serial chain and a handful of blocks already fills the chip.
// Level outputs ping-pong between two LDS buffers; a level is one pass of "lane i: UBI chain over node i".
// ---------------------------------------------------------------------------------------------------
template <int NW>
static __device__ __forceinline__ void load_node_words(uint64_t (&w)[NW], const uint8_t *p, size_t n)
{
    // n = bytes available (1..NW*8, or 0): whole 8-byte words are read as such, the rest byte-wise, zero padded
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint64_t v = 0;
        if ((size_t)(8 * i + 8) <= n) __builtin_memcpy(&v, p + 8 * i, 8);
        else
            for (int b = 0; b < 8; b++)
                if ((size_t)(8 * i + b) < n) v |= (uint64_t)p[8 * i + b] << (8 * b);
        w[i] = v;
    }
}

template <int NW>
__global__ void __launch_bounds__(64)
skein_tree_kernel(const uint8_t *__restrict__ src, size_t block_bytes, size_t src_stride, size_t nblocks, SkeinIV g,
                  uint8_t *__restrict__ digests, unsigned digest_bytes, unsigned leaf, unsigned node, unsigned max_level,
                  unsigned cap_a)
{
    constexpr unsigned BB = NW * 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t lvl[]; // level buffers: [0, cap_a) and [cap_a, ...)
    const uint32_t lane = threadIdx.x;

    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const uint8_t *m = src + blk * src_stride;
        size_t bcnt = block_bytes;
        uint64_t S[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) S[i] = 0;
        __syncthreads(); // the previous block's level buffers are free
        for (unsigned height = 0;; height++) {
            if (height && bcnt == BB) { // one block left: its bytes are the last chaining value
                load_node_words<NW>(S, m, BB);
                break;
            }
            const bool last_level = height + 1 == max_level; // hash whatever is left in one chain
            const size_t node_len = last_level ? bcnt : (size_t)BB << (height ? node : leaf);
            const size_t nnodes = bcnt == 0 || last_level ? 1 : (bcnt + node_len - 1) / node_len;
            uint8_t *out = lvl + ((height & 1) ? cap_a : 0);
            for (size_t i = lane; i < nnodes; i += 64) {
                const size_t off = i * node_len, n = bcnt - off < node_len ? bcnt - off : node_len;
                uint64_t X[NW], w[NW];
#pragma unroll
                for (int k = 0; k < NW; k++) X[k] = g.w[k];
                uint64_t t0 = off, t1 = T1_FIRST | T1_MSG | ((uint64_t)(height + 1) << 48); // tree level: tweak bits 112..118
                for (size_t pos = 0;;) {
                    const bool fin = n - pos <= BB;
                    const size_t take = fin ? n - pos : BB;
                    load_node_words<NW>(w, m + off + pos, take);
                    t0 += take;
                    Ubi<NW>::run(X, w, t0, fin ? t1 | T1_FINAL : t1);
                    t1 &= ~T1_FIRST;
                    pos += take;
                    if (fin) break;
                }
                if (last_level) {
#pragma unroll
                    for (int k = 0; k < NW; k++) S[k] = X[k];
                } else {
#pragma unroll
                    for (int k = 0; k < NW; k++) *reinterpret_cast<uint64_t *>(out + i * BB + 8 * k) = X[k];
                }
            }
            if (last_level) break; // (lane 0 holds the chain)
            __syncthreads();
            m = out;
            bcnt = nnodes * BB;
        }
        if (lane == 0) { // output transform (skein.c:391-405) on the last chaining value
            uint64_t w[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) w[k] = 0;
            Ubi<NW>::run(S, w, 8, T1_FIRST | T1_FINAL | T1_OUT);
            uint8_t *o = digests + blk * digest_bytes;
            for (unsigned k = 0; k < digest_bytes; k++) o[k] = (uint8_t)(S[k >> 3] >> (8 * (k & 7)));
        }
    }
}

hipError_t skein_tree_launch(int nw, const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, unsigned hash_bits,
                             unsigned leaf, unsigned node, unsigned max_level, uint8_t *digests, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if ((nw != 4 && nw != 8) || leaf == 0 || node == 0 || max_level < 2 || leaf > 20 || node > 20 || max_level > 255 ||
        hash_bits == 0 || hash_bits > (unsigned)nw * 64 || hash_bits % 8)
        return hipErrorInvalidValue;
    const size_t bb = (size_t)nw * 8;
    // level-1 output and level-2 output bound the two LDS buffers (later levels only shrink)
    const size_t n1 = block_bytes ? (block_bytes + (bb << leaf) - 1) / (bb << leaf) : 1;
    const size_t n2 = (n1 * bb + (bb << node) - 1) / (bb << node);
    const size_t cap_a = (n1 * bb + 15) & ~(size_t)15, cap_b = (n2 * bb + 15) & ~(size_t)15;
    if (cap_a + cap_b > 64 * 1024) return hipErrorInvalidValue; // leaves too small for this block size
    SkeinIV g;
    skein_compute_iv(nw, hash_bits, &g, (uint64_t)leaf | ((uint64_t)node << 8) | ((uint64_t)max_level << 16));
    const size_t grid = nblocks < 256 * 16 ? nblocks : 256 * 16; // 4 wavefronts per SIMD (VGPR bound)
    if (nw == 8)
        hipLaunchKernelGGL(skein_tree_kernel<8>, dim3((unsigned)grid), dim3(64), cap_a + cap_b, stream, src, block_bytes, src_stride,
                           nblocks, g, digests, hash_bits / 8, leaf, node, max_level, (unsigned)cap_a);
    else
        hipLaunchKernelGGL(skein_tree_kernel<4>, dim3((unsigned)grid), dim3(64), cap_a + cap_b, stream, src, block_bytes, src_stride,
                           nblocks, g, digests, hash_bits / 8, leaf, node, max_level, (unsigned)cap_a);
    return hipGetLastError();
}

template <int NW>
static hipError_t launch_skein(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                               uint8_t *digests, unsigned digest_bytes, hipStream_t stream, bool lean)
{
    if (nblocks == 0) return hipSuccess;
    const dim3 grid((unsigned)((nblocks + CW_SKEIN_THREADS - 1) / CW_SKEIN_THREADS)), block(CW_SKEIN_THREADS);
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | src_stride) & 15) == 0;
    const bool ragged = block_bytes == 0 || (block_bytes % (NW * 8)) != 0;
    // the name is noted by the macro that launches (cw_profile_kernels), spelled as rocprofv3 prints the instantiation
#define CW_LAUNCH(A, R) do { hipLaunchKernelGGL((skein_blocks_kernel<NW, A, R>), grid, block, 0, stream, \
                                                src, block_bytes, src_stride, nblocks, iv, digests, digest_bytes); \
                             note_kernels(1, NW == 8 ? "cw::skein_blocks_kernel<8, " #A ", " #R ">" : "cw::skein_blocks_kernel<4, " #A ", " #R ">"); } while (0)
#define CW_LAUNCH_LINES(A) do { hipLaunchKernelGGL((skein_lines_kernel<NW, A>), grid, block, 0, stream, \
                                                   src, block_bytes, src_stride, nblocks, iv, digests, digest_bytes); \
                                note_kernels(1, NW == 8 ? "cw::skein_lines_kernel<8, " #A ">" : "cw::skein_lines_kernel<4, " #A ">"); } while (0)
    // Two hot kernels for aligned, whole-step blocks: the line kernel (105 VGPRs, every cache line fetched once) and
    // the step kernel (92 VGPRs, 64 bytes per step, ~40 % of the lines fetched twice).  Alone they are equally fast;
    // beside codec wavefronts the step kernel keeps 4 instead of 3 hash wavefronts per SIMD, which helped at 512 Ki
    // blocks (32.7 vs 36-42 ms) and made no difference at 1 Mi blocks (68 vs 69 ms), so callers do not ask for it
    // (`lean` stays false); CW_SKEIN_MODE=steps|lines overrides (profiling knob).
    const char *mode = tune("CW_SKEIN_MODE");
    const bool steps = mode ? strcmp(mode, "steps") == 0 : lean;
    if (aligned && !ragged && steps) CW_LAUNCH(true, false);
    else if (aligned && !ragged) CW_LAUNCH_LINES(true);
    else if (aligned) CW_LAUNCH(true, true);
    else if (!ragged) CW_LAUNCH_LINES(false);
    else CW_LAUNCH(false, true);
#undef CW_LAUNCH
#undef CW_LAUNCH_LINES
    return hipGetLastError();
}

hipError_t skein512_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                           uint8_t *digests, unsigned digest_bytes, hipStream_t stream, bool lean)
{
    return launch_skein<8>(src, block_bytes, src_stride, nblocks, iv, digests, digest_bytes, stream, lean);
}

hipError_t skein256_launch(const uint8_t *src, size_t block_bytes, size_t src_stride, size_t nblocks, const SkeinIV &iv,
                           uint8_t *digests, unsigned digest_bytes, hipStream_t stream, bool lean)
{
    return launch_skein<4>(src, block_bytes, src_stride, nblocks, iv, digests, digest_bytes, stream, lean);
}

// ---- host-side config-block UBI (Skein_*_Init's "no precomputed IV" path, skein.c:245-259) ----
static uint64_t h_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

void skein_compute_iv(int nw, unsigned hash_bits, SkeinIV *iv, uint64_t tree_info)
{
    static const int rot8[8][4] = {{38, 30, 50, 53}, {48, 20, 43, 31}, {34, 14, 15, 27}, {26, 12, 58, 7},
                                   {33, 49, 8, 42},  {39, 27, 41, 14}, {29, 26, 11, 9},  {33, 51, 39, 35}};
    static const int rot4[8][2] = {{5, 56}, {36, 28}, {13, 46}, {58, 44}, {26, 20}, {53, 35}, {11, 42}, {59, 50}};
    static const int perm8[8] = {2, 1, 4, 7, 6, 5, 0, 3}, perm4[4] = {0, 3, 2, 1};
    uint64_t ks[9] = {0}, ts[3], w[8] = {0}, v[8], t[8];
    w[0] = (1ULL << 32) | 0x33414853ULL; // schema version 1, "SHA3"
    w[1] = hash_bits;
    w[2] = tree_info;                    // 0 = sequential; leaf | node << 8 | maxLevel << 16 (skein.h:209-210)
    ks[nw] = KS_PARITY;                  // chaining value is all zero
    ts[0] = 32;                          // config string length
    ts[1] = T1_FIRST | T1_FINAL | (4ULL << 56);
    ts[2] = ts[0] ^ ts[1];
    for (int i = 0; i < nw; i++) v[i] = w[i];
    for (int d = 0; d <= 72; d++) {
        if ((d & 3) == 0) {
            const int s = d >> 2;
            for (int i = 0; i < nw; i++) v[i] += ks[(s + i) % (nw + 1)];
            v[nw - 3] += ts[s % 3];
            v[nw - 2] += ts[(s + 1) % 3];
            v[nw - 1] += (uint64_t)s;
            if (d == 72) break;
        }
        for (int i = 0; i < nw / 2; i++) {
            const int r = nw == 8 ? rot8[d & 7][i] : rot4[d & 7][i];
            v[2 * i] += v[2 * i + 1];
            v[2 * i + 1] = h_rotl(v[2 * i + 1], r) ^ v[2 * i];
        }
        for (int i = 0; i < nw; i++) t[i] = v[nw == 8 ? perm8[i] : perm4[i]];
        for (int i = 0; i < nw; i++) v[i] = t[i];
    }
    for (int i = 0; i < 8; i++) iv->w[i] = i < nw ? (v[i] ^ w[i]) : 0;
}

} // namespace cw
