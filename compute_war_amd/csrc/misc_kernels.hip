// misc_kernels.hip -- synthetic-input generator and compressed-size reduction (gfx950).
//
// gen_random: the benchmark stream of SURVEY.md 8(d): u64 word w of block b is
// splitmix64(seed ^ (b << 13 | w)), so host (oracle/hc_oracle.c) and device regenerate any block
// identically.  One lane writes 16 B; a wavefront writes 1 KiB contiguous (coalesced dwordx4 stores).
//
// sum_sizes: total compressed bytes for the compression ratio (the reference derives it from
// experiment.cpp's per-block csize column, src/compression_perf/src/experiment.cpp:243-267).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cw_device.h"

namespace cw {

static __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256)
gen_random_kernel(uint64_t seed, uint64_t first_block, size_t total_pairs, unsigned pairs_per_block, uint4 *__restrict__ dst)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pairs; i += stride) {
        const uint64_t blk = first_block + i / pairs_per_block;
        const uint64_t w = 2 * (i % pairs_per_block);
        const uint64_t a = splitmix64(seed ^ ((blk << 13) | w));
        const uint64_t b = splitmix64(seed ^ ((blk << 13) | (w + 1)));
        dst[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
}

// gen_mixed: the compressible synthetic mix of SURVEY.md 8(d) -- even blocks are the uniform-random stream above, odd
// blocks repeat a 64-byte motif (8 u64 words keyed by the block) in which every byte is replaced by a random one with
// probability 1/16, so that the LZ4/LZF match + literal emit loops run (short matches broken by literals, offsets of a
// few motif periods).  Word w of odd block b:
//   motif  m = splitmix64(seed ^ kSaltMotif ^ (b << 13 | (w & 7)))
//   draw   r = splitmix64(seed ^ kSaltMutate ^ (b << 13 | w)),  r2 = splitmix64(r)
//   byte k = nibble k of r is 0 ? byte k of r2 : byte k of m
// The checker regenerates sampled blocks on the host from this definition (the test infrastructure holds its own twin).
constexpr uint64_t kSaltMotif = 0x6D6F746966ULL, kSaltMutate = 0x6D7574617465ULL;

static __device__ __forceinline__ uint64_t mixed_word(uint64_t seed, uint64_t blk, uint64_t w)
{
    if ((blk & 1) == 0) return splitmix64(seed ^ ((blk << 13) | w));
    const uint64_t m = splitmix64(seed ^ kSaltMotif ^ ((blk << 13) | (w & 7)));
    const uint64_t r = splitmix64(seed ^ kSaltMutate ^ ((blk << 13) | w)), r2 = splitmix64(r);
    uint64_t mask = 0; // 0xFF in every byte whose nibble of r is zero
#pragma unroll
    for (int k = 0; k < 8; k++) mask |= ((r >> (4 * k)) & 15) == 0 ? 0xFFULL << (8 * k) : 0;
    return (m & ~mask) | (r2 & mask);
}

__global__ void __launch_bounds__(256)
gen_mixed_kernel(uint64_t seed, uint64_t first_block, size_t total_pairs, unsigned pairs_per_block, uint4 *__restrict__ dst)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pairs; i += stride) {
        const uint64_t blk = first_block + i / pairs_per_block;
        const uint64_t w = 2 * (i % pairs_per_block);
        const uint64_t a = mixed_word(seed, blk, w), b = mixed_word(seed, blk, w + 1);
        dst[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
}

hipError_t gen_mixed_launch(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes, uint8_t *dst,
                            hipStream_t stream)
{
    if (nblocks == 0 || block_bytes == 0) return hipSuccess;
    if (block_bytes % 16 || (reinterpret_cast<uintptr_t>(dst) & 15)) return hipErrorInvalidValue;
    const size_t total = nblocks * (block_bytes / 16);
    size_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(gen_mixed_kernel, dim3((unsigned)grid), dim3(256), 0, stream, seed, first_block, total,
                       (unsigned)(block_bytes / 16), reinterpret_cast<uint4 *>(dst));
    return hipGetLastError();
}

hipError_t gen_random_launch(uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_bytes, uint8_t *dst,
                             hipStream_t stream)
{
    if (nblocks == 0 || block_bytes == 0) return hipSuccess;
    if (block_bytes % 16 || (reinterpret_cast<uintptr_t>(dst) & 15)) return hipErrorInvalidValue;
    const size_t total = nblocks * (block_bytes / 16);
    size_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16; // grid-stride beyond 16 workgroups per CU
    hipLaunchKernelGGL(gen_random_kernel, dim3((unsigned)grid), dim3(256), 0, stream, seed, first_block, total,
                       (unsigned)(block_bytes / 16), reinterpret_cast<uint4 *>(dst));
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
sum_sizes_kernel(const uint32_t *__restrict__ sizes, size_t n, uint32_t raw_bytes, unsigned long long *__restrict__ totals)
{
    unsigned long long bytes = 0, zeros = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t s = sizes[i];
        bytes += s ? s : raw_bytes;
        zeros += s == 0;
    }
    // wavefront reduction (64 lanes), then one atomic per wavefront
    for (int off = 32; off > 0; off >>= 1) {
        bytes += __shfl_down(bytes, off, 64);
        zeros += __shfl_down(zeros, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&totals[0], bytes);
        atomicAdd(&totals[1], zeros);
    }
}

hipError_t sum_sizes_launch(const uint32_t *sizes, size_t n, uint32_t raw_bytes, uint64_t *totals, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    size_t grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(sum_sizes_kernel, dim3((unsigned)grid), dim3(256), 0, stream, sizes, n, raw_bytes,
                       reinterpret_cast<unsigned long long *>(totals));
    return hipGetLastError();
}

} // namespace cw
