// scalar_thread.h -- helpers of the parsers that run a wavefront as ONE scalar thread (lz4_vtab_kernel.hip, the LZF scalar-thread
// parser in lzf_kernel.hip): input through the scalar data cache, wave-uniform values on the VALU, prefix sums and lane reads for
// the batched output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cw {
namespace st {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// buffer descriptor for s_buffer_load_*: `bytes` bytes at p, raw, 32-bit elements (gfx9 family); dwords beyond read as zero
__device__ __forceinline__ u32x4 sc_descriptor(const void *p, uint32_t bytes)
{
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
    rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xFFFFu);
    rs.z = bytes;
    rs.w = 0x00020000u;
    return rs;
}

// ---- the input through the scalar cache -------------------------------------------------------------------------------
// s_buffer_load_*: bytes at byte offset `off` (a multiple of 4) of the block described by rs; dwords outside [0, num_records) read as
// zero (an offset that wrapped below zero makes the WHOLE load read as zero: tools/sbuf.hip).
// RULE: a scalar load and the s_waitcnt that covers it are ONE asm statement.  The compiler does not know that the destination
// registers of an s_buffer_load in inline assembly are still in flight behind the statement: with the load in one statement and the
// wait in a later one (as lz4_vtab_kernel.hip had it, to run the emission of a sequence under the next windows' latency) it is free to copy or
// spill those registers in between -- and did, in the -DCW_VSTAMP build of the day: an `s_mov_b64` of the two dwords in front of the
// emission read them before they had arrived once in ~40,000 sequences, and half of all blocks came out a few bytes wrong.  The
// product build of the same source happened to have no such copy and passed every test.
__device__ __forceinline__ u32x8 sc_load32_now(const u32x4 &rs, uint32_t off)
{
    u32x8 v;
    asm volatile("s_buffer_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(rs), "s"(off));
    return v;
}

// (one statement per load group and its wait: see sc_load32_now)
__device__ __forceinline__ void sc_load4x2_now(const u32x4 &rs, uint32_t off_a, uint32_t off_b, uint32_t &a, uint32_t &b)
{
    asm volatile("s_buffer_load_dword %0, %2, %3\n\ts_buffer_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(rs), "s"(off_a), "s"(off_b));
}
__device__ __forceinline__ void sc_load32_8_now(const u32x4 &rs, uint32_t off_a, uint32_t off_b, u32x8 &a, u32x2 &b)
{
    asm volatile("s_buffer_load_dwordx8 %0, %2, %3\n\ts_buffer_load_dwordx2 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(rs), "s"(off_a), "s"(off_b));
}
// a value the compiler shall treat as computed once and opaque from here on (the LDS base address of a dynamic array: as an expression it is
// re-derived -- a null check of the address-space cast, three scalar instructions -- at every use inside the parse loops)
__device__ __forceinline__ uint32_t opaque_s(uint32_t v)
{
    asm volatile("" : "+s"(v));
    return v;
}
// a wave-uniform value moved to a vector register: what is computed from it runs on the VALU
__device__ __forceinline__ uint32_t to_v(uint32_t s)
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
// 4 bytes at bit offset sh (0, 8, 16 or 24) of the 8 bytes lo | hi << 32, on the scalar unit (lo, hi: an aligned register pair)
__device__ __forceinline__ uint32_t cut32(uint32_t lo, uint32_t hi, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_or0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false); // lanes without a source lane (or in a masked row): 0
}
// inclusive prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += dpp_or0<0x111, 0xF>(v);
    v += dpp_or0<0x112, 0xF>(v);
    v += dpp_or0<0x114, 0xF>(v);
    v += dpp_or0<0x118, 0xF>(v);
    v += dpp_or0<0x142, 0xA>(v);
    v += dpp_or0<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t from) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v); }

// 8 bytes at bit offset sh of the 12 bytes d1 | d2 << 32 | d3 << 64, on the VALU (wave-uniform vector registers)
__device__ __forceinline__ void cut64_v(uint32_t d1, uint32_t d2, uint32_t d3, uint32_t sh, uint32_t &lo, uint32_t &hi)
{
    const uint32_t v1 = to_v(d1), v2 = to_v(d2), v3 = to_v(d3);
    lo = __builtin_amdgcn_alignbit(v2, v1, sh);
    hi = __builtin_amdgcn_alignbit(v3, v2, sh);
}
// number of equal low bytes of two 8-byte values (0..8), on the VALU
__device__ __forceinline__ uint32_t equal_bytes_v(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi)
{
    const uint32_t x0 = alo ^ blo, x1 = ahi ^ bhi;
    return x0 ? (uint32_t)__builtin_ctz(x0) >> 3 : x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : 8u;
}

} // namespace st
} // namespace cw
