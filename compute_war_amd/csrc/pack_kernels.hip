// pack_kernels.hip -- packed output stream: the compressed blocks leave the codec kernels in
// fixed-stride slots (slot i at dst + i * dst_stride, sizes[i] bytes used); this turns them into one contiguous
// stream plus a block index: offsets[i] = sum of sizes[0..i) as u64, offsets[nblocks] = total.  A block that did
// not fit (sizes[i] == 0, LZF) occupies no bytes in the stream -- its index entry is empty and the caller keeps the
// raw block, as the reference's callers do (HashAndCompress.cpp:247-252).
//
// Three small kernels for the exclusive scan (tile sums, scan of the tile sums by one workgroup, tile offsets) and
// one copy kernel: a wavefront per slot, 16-byte aligned stores fed by unaligned loads.  The host batch API uses
// the packed stream to bring a whole batch back in ONE device-to-host copy instead of one per block.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

#include "cw_device.h"
#include "lz_device.h"

namespace cw {

namespace {

constexpr unsigned kTile = 4096, kThreads = 256, kPerThread = kTile / kThreads;

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ void __launch_bounds__(kThreads)
tile_sums_kernel(const uint32_t *__restrict__ sizes, size_t n, unsigned long long *__restrict__ partial)
{
    __shared__ unsigned long long wsum[kThreads / 64];
    const size_t base = (size_t)blockIdx.x * kTile;
    unsigned long long s = 0;
    for (unsigned k = 0; k < kPerThread; k++) {
        const size_t i = base + k * kThreads + threadIdx.x;
        if (i < n) s += sizes[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup: exclusive scan of the tile sums in place
__global__ void __launch_bounds__(kThreads)
scan_partials_kernel(unsigned long long *__restrict__ partial, size_t ntiles)
{
    __shared__ unsigned long long buf[kThreads];
    unsigned long long carry = 0;
    for (size_t base = 0; base < ntiles; base += kThreads) {
        const size_t i = base + threadIdx.x;
        const unsigned long long v = i < ntiles ? partial[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (unsigned d = 1; d < kThreads; d <<= 1) { // Hillis-Steele inclusive scan
            const unsigned long long add = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < ntiles) partial[i] = carry + buf[threadIdx.x] - v;
        carry += buf[kThreads - 1];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kThreads)
tile_offsets_kernel(const uint32_t *__restrict__ sizes, size_t n, const unsigned long long *__restrict__ partial,
                    unsigned long long *__restrict__ offsets)
{
    __shared__ unsigned long long buf[kThreads];
    // thread t owns kPerThread consecutive entries of the tile
    const size_t first = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * kPerThread;
    uint32_t v[kPerThread];
    unsigned long long s = 0;
    for (unsigned k = 0; k < kPerThread; k++) {
        v[k] = first + k < n ? sizes[first + k] : 0;
        s += v[k];
    }
    buf[threadIdx.x] = s;
    __syncthreads();
    for (unsigned d = 1; d < kThreads; d <<= 1) {
        const unsigned long long add = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
        __syncthreads();
        buf[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long off = partial[blockIdx.x] + buf[threadIdx.x] - s;
    for (unsigned k = 0; k < kPerThread; k++) {
        if (first + k < n) offsets[first + k] = off;
        off += v[k];
        if (first + k + 1 == n) offsets[n] = off; // total
    }
}

__global__ void __launch_bounds__(64)
pack_copy_kernel(const uint8_t *__restrict__ slots, size_t slot_stride, const uint32_t *__restrict__ sizes,
                 const unsigned long long *__restrict__ offsets, size_t n, uint8_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x;
    for (size_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t len = sizes[i];
        if (len) lz::copy_g2g(out + offsets[i], slots + i * slot_stride, len, lane);
    }
}

struct Workspace { unsigned long long *p = nullptr; size_t cap = 0; std::mutex launch; };
std::mutex ws_lock;
std::unordered_map<uint64_t, Workspace> ws_map; // references stay valid across inserts

} // namespace

void pack_release_workspaces()
{
    std::lock_guard<std::mutex> g(ws_lock);
    for (auto &kv : ws_map) if (kv.second.p) (void)hipFree(kv.second.p);
    ws_map.clear();
}

void pack_release_stream(hipStream_t stream)
{
    std::lock_guard<std::mutex> g(ws_lock);
    auto it = ws_map.find(ws_key(stream));
    if (it == ws_map.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    ws_map.erase(it);
}

hipError_t pack_launch(const uint8_t *slots, size_t slot_stride, const uint32_t *sizes, size_t nblocks, uint8_t *packed,
                       uint64_t *offsets, hipStream_t stream)
{
    if (nblocks == 0) return hipMemsetAsync(offsets, 0, sizeof(uint64_t), stream);
    const size_t ntiles = (nblocks + kTile - 1) / kTile;
    unsigned long long *partial = nullptr;
    Workspace *wsp;
    {
        std::lock_guard<std::mutex> g(ws_lock);
        wsp = &ws_map[ws_key(stream)];
    }
    std::lock_guard<std::mutex> sequence(wsp->launch); // the tile partials are shared by the launches below
    {
        Workspace &w = *wsp;
        if (w.cap < ntiles) { // first (or a larger) call on this stream
            if (w.p) { hipError_t e = hipFree(w.p); if (e != hipSuccess) return e; }
            w.p = nullptr; w.cap = 0;
            const size_t cap = ntiles < 1024 ? 1024 : ntiles;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&w.p), cap * sizeof(unsigned long long));
            if (e != hipSuccess) return e;
            w.cap = cap;
        }
        partial = w.p;
    }
    unsigned long long *off = reinterpret_cast<unsigned long long *>(offsets);
    hipLaunchKernelGGL(tile_sums_kernel, dim3((unsigned)ntiles), dim3(kThreads), 0, stream, sizes, nblocks, partial);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kThreads), 0, stream, partial, ntiles);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((unsigned)ntiles), dim3(kThreads), 0, stream, sizes, nblocks, partial, off);
    if (packed) {
        const size_t grid = nblocks < 256 * 32 ? nblocks : 256 * 32;
        hipLaunchKernelGGL(pack_copy_kernel, dim3((unsigned)grid), dim3(64), 0, stream, slots, slot_stride, sizes, off, nblocks, packed);
    }
    return hipGetLastError();
}

} // namespace cw
