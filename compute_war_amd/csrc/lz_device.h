// lz_device.h -- device helpers shared by the LZ4 and LZF parse kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cw {
namespace lz {

__device__ __forceinline__ uint32_t rd32(const uint8_t *p, uint32_t pos)
{
    uint32_t v;
    __builtin_memcpy(&v, p + pos, 4); // unaligned load
    return v;
}

// 4 bytes at any offset of a block staged in LDS.  An unaligned ds_read is legal but the LDS handles it lane by lane
// (SQ_LDS_UNALIGNED_STALL was 92 % of the LDS-busy cycles of the LZ4 parse kernel, which it saturated), so: the two
// aligned dwords around the position (one ds_read2_b32) and a byte align; the caller guarantees that the second
// dword is inside the staged bytes.  Blocks read from global memory use plain unaligned loads.
template <bool STAGED>
__device__ __forceinline__ uint32_t rd32x(const uint8_t *in, uint32_t pos)
{
    if (STAGED) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + (pos >> 2);
        return __builtin_amdgcn_alignbyte(w[1], w[0], pos & 3u);
    }
    return rd32(in, pos);
}

// Masked exchange of the 16-bit slot h of a u16 table in LDS (byte address tab_lds): stores pos, returns the previous
// content.  ds_mskor_rtn_b32 applies the lanes of one instruction in ascending lane order (tools/mskor_order.hip),
// so a lane also sees the stores of earlier lanes of the same instruction; callers verify that on every use.
__device__ __forceinline__ uint32_t tab_exchange(uint32_t tab_lds, uint32_t h, uint32_t pos)
{
    const uint32_t addr = tab_lds + (h >> 1) * 4, sh = (h & 1) * 16;
    uint32_t old;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(old) : "v"(addr), "v"(0xFFFFu << sh), "v"(pos << sh) : "memory");
    return (old >> sh) & 0xFFFFu;
}

// Table exchange + fingerprint exchange of one batch of items: the 16-bit slot h of the position table (byte address
// tab_lds) and the 4-bit slot h of the fingerprint table beside it (byte address fp_lds, eight slots per 32-bit word),
// two ds_mskor_rtn_b32 behind one wait.  Returns the previous position; *fp_old receives the previous fingerprint.
__device__ __forceinline__ uint32_t tab_fp_exchange(uint32_t tab_lds, uint32_t fp_lds, uint32_t h, uint32_t pos, uint32_t fp,
                                                    uint32_t *fp_old)
{
    const uint32_t addr = tab_lds + (h >> 1) * 4, sh = (h & 1) * 16;
    const uint32_t faddr = fp_lds + (h >> 3) * 4, fsh = (h & 7) * 4;
    uint32_t old, fo;
    asm volatile("ds_mskor_rtn_b32 %0, %2, %3, %4\n\tds_mskor_rtn_b32 %1, %5, %6, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old), "=&v"(fo)
                 : "v"(addr), "v"(0xFFFFu << sh), "v"(pos << sh), "v"(faddr), "v"(0xFu << fsh), "v"(fp << fsh)
                 : "memory");
    *fp_old = (fo >> fsh) & 0xFu;
    return (old >> sh) & 0xFFFFu;
}
// put a fingerprint back (undo of a speculative exchange); no value returned, ordered with the wavefront's other LDS operations
__device__ __forceinline__ void fp_store(uint32_t fp_lds, uint32_t h, uint32_t fp)
{
    const uint32_t faddr = fp_lds + (h >> 3) * 4, fsh = (h & 7) * 4;
    asm volatile("ds_mskor_b32 %0, %1, %2" : : "v"(faddr), "v"(0xFu << fsh), "v"(fp << fsh) : "memory");
}

// the 16 bytes around a position: [p-4, p) (only if has_before), [p, p+4), [p+4, p+12)
struct Around { uint32_t before, at; uint64_t after; };
template <bool STAGED>
__device__ __forceinline__ Around around(const uint8_t *in, uint32_t p, bool has_before)
{
    Around a;
    if (STAGED) { // five aligned dwords from p-4 on; the last one only if the bytes are not dword aligned
        const uint32_t *w = reinterpret_cast<const uint32_t *>(in) + (p >> 2);
        const uint32_t sh = p & 3u;
        const uint32_t w0 = w[has_before ? -1 : 0], w1 = w[0], w2 = w[1], w3 = w[2], w4 = w[sh ? 3 : 2];
        a.before = __builtin_amdgcn_alignbyte(w1, w0, sh);
        a.at = __builtin_amdgcn_alignbyte(w2, w1, sh);
        a.after = __builtin_amdgcn_alignbyte(w3, w2, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(w4, w3, sh) << 32);
    } else {
        a.before = rd32(in, has_before ? p - 4 : 0u);
        a.at = rd32(in, p);
        __builtin_memcpy(&a.after, in + p + 4, 8);
    }
    return a;
}

// wavefront copy global -> global, any alignment: aligned 16 B stores fed by unaligned 16 B loads
__device__ __forceinline__ void copy_g2g(uint8_t *__restrict__ d, const uint8_t *__restrict__ s, uint32_t len, uint32_t lane)
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
    for (; i + 15 * 64 < nvec; i += 16 * 64) { // 16 KiB in flight per wavefront
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) __builtin_memcpy(&v[u], s + 16 * (size_t)(i + 64 * u), 16);
#pragma unroll
        for (int u = 0; u < 16; u++) *reinterpret_cast<uint4 *>(d + 16 * (size_t)(i + 64 * u)) = v[u];
    }
    for (; i < nvec; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, s + 16 * (size_t)i, 16);
        *reinterpret_cast<uint4 *>(d + 16 * (size_t)i) = v;
    }
    const uint32_t done = nvec << 4, tail = len - done;
    if (lane < tail) d[done + lane] = s[done + lane];
}

} // namespace lz
} // namespace cw
