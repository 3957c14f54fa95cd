// random_line.hip -- how many random memory lines per second an MI355X retires for the access pattern of the lane-per-block
// LZ parsers: every lane owns a private table (T bytes) and a private window (W bytes) in global memory and
// per iteration loads one table entry at a random slot, stores to the same slot, and -- every third iteration, the share of
// candidates the fingerprint lets through on text -- loads 16 bytes at a random offset of its window.  "dep" makes the next
// slot depend on the loaded entry and the window load on the table load (the parser's chain); "ind" takes all addresses from
// a counter-based generator (no dependence on loaded data: what the memory system retires when latency is hidden).
//   hipcc --offload-arch=gfx950 -O3 -o tools/random_line tools/random_line.hip && tools/random_line
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// HOW: 0 plain load + store, 1 nontemporal load + store, 2 plain load + nontemporal store, 3 one returning atomic exchange,
// 4 the LZ4 parser's mix on text (tools/lz_probe_count.py: 2 probes per sequence): every second probe also stores an entry at
// another random slot (the ip-2 entry in front of a re-test) and fetches its candidate (it is the match)
template <bool DEP, int ENTRY, int HOW = 0>
__global__ void __launch_bounds__(64)
probe_kernel(uint8_t *tables, size_t tbytes, const uint8_t *windows, size_t wbytes, uint32_t iters, uint32_t *sink)
{
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint8_t *tab = tables + lane * tbytes;
    const uint8_t *win = windows + lane * wbytes;
    const uint32_t slots = (uint32_t)(tbytes / ENTRY);
    uint32_t x = mix((uint32_t)lane * 2654435761u + 1), acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t slot = x % slots;
        uint32_t e;
        if (HOW == 3) {
            e = __hip_atomic_exchange(reinterpret_cast<uint32_t *>(tab) + slot, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (ENTRY == 2 && HOW != 4) {
            uint16_t *p = reinterpret_cast<uint16_t *>(tab) + slot;
            e = HOW == 1 ? __builtin_nontemporal_load(p) : *p;
            if (HOW) __builtin_nontemporal_store((uint16_t)i, p); else *p = (uint16_t)i;
        } else {
            uint32_t *p = reinterpret_cast<uint32_t *>(tab) + slot;
            e = HOW == 1 ? __builtin_nontemporal_load(p) : *p;
            if (HOW == 1 || HOW == 2) __builtin_nontemporal_store(i, p); else *p = i;
        }
        uint32_t y = mix(x + 0x9E3779B9u);
        if (DEP) y ^= e & 1u; // (entries are small counters: the address now waits for the load)
        if (HOW == 4 && (i & 1)) reinterpret_cast<uint32_t *>(tab)[(y >> 7) % slots] = i;
        if (HOW == 4 ? (i & 1) : y % 3 == 0) {
            uint4 q;
            __builtin_memcpy(&q, win + (y >> 2) % (uint32_t)(wbytes - 16), 16);
            acc += q.x ^ q.w;
            if (DEP) y ^= q.y & 1u;
        }
        acc += e;
        x = DEP ? mix(y) : mix(x + i);
    }
    if (acc == 0xDEADBEEFu) sink[0] = acc;
}

template <bool DEP, int ENTRY, int HOW = 0>
static void run(const char *name, int wpc, size_t tbytes, size_t wbytes, uint32_t iters, uint8_t *tables, uint8_t *windows, uint32_t *sink)
{
    const unsigned grid = 256u * (unsigned)wpc;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe_kernel<DEP, ENTRY, HOW>), dim3(grid), dim3(64), 0, 0, tables, tbytes, windows, wbytes, iters / 8, sink); // warm
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((probe_kernel<DEP, ENTRY, HOW>), dim3(grid), dim3(64), 0, 0, tables, tbytes, windows, wbytes, iters, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)grid * 64 * iters;
    printf("%-4s entry=%d B table=%3zu KiB window=%3zu KiB wavefronts/CU=%2d lanes=%7u: %8.2f ms  %6.2f G probes/s  (%.2f us per probe and lane)\n", name,
           ENTRY, tbytes >> 10, wbytes >> 10, wpc, grid * 64, ms, probes / ms / 1e6, ms * 1e3 / iters);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const int max_wpc = 16;
    const size_t lanes = (size_t)256 * max_wpc * 64, tmax = 128 << 10, wmax = 64 << 10;
    uint8_t *tables, *windows;
    uint32_t *sink;
    CHECK(hipMalloc(reinterpret_cast<void **>(&tables), lanes * tmax));
    CHECK(hipMalloc(reinterpret_cast<void **>(&windows), lanes * wmax));
    CHECK(hipMalloc(reinterpret_cast<void **>(&sink), 64));
    CHECK(hipMemset(tables, 0, lanes * tmax));
    CHECK(hipMemset(windows, 1, lanes * wmax));
    const int wpcs[] = {2, 4, 8, 16};
    for (int wpc : wpcs) run<true, 4>("dep", wpc, 32 << 10, 64 << 10, iters, tables, windows, sink);   // LZ4 lanes, 64 KiB blocks
    for (int wpc : wpcs) run<false, 4>("ind", wpc, 32 << 10, 64 << 10, iters, tables, windows, sink);
    for (int wpc : wpcs) run<true, 2>("dep", wpc, 16 << 10, 4 << 10, iters, tables, windows, sink);    // LZ4 lanes, 4 KiB blocks
    for (int wpc : wpcs) run<true, 2>("dep", wpc, 128 << 10, 64 << 10, iters, tables, windows, sink);  // LZF lanes, 64 KiB blocks
    for (int wpc : wpcs) run<false, 2>("ind", wpc, 128 << 10, 64 << 10, iters, tables, windows, sink);
    for (int wpc : wpcs) run<true, 2>("dep", wpc, 128 << 10, 4 << 10, iters, tables, windows, sink);   // LZF lanes, 4 KiB blocks
    // other ways to touch the table (LZ4 lanes' shape, 64 KiB blocks)
    for (int wpc : {4, 8}) run<true, 4, 1>("dep nontemporal load+store", wpc, 32 << 10, 64 << 10, iters, tables, windows, sink);
    for (int wpc : {4, 8}) run<true, 4, 2>("dep nontemporal store", wpc, 32 << 10, 64 << 10, iters, tables, windows, sink);
    for (int wpc : {4, 8}) run<true, 4, 3>("dep atomic exchange", wpc, 32 << 10, 64 << 10, iters, tables, windows, sink);
    for (int wpc : {4, 8}) run<true, 2, 1>("dep nontemporal load+store", wpc, 16 << 10, 64 << 10, iters, tables, windows, sink);
    for (int wpc : {4, 6, 8}) run<true, 4, 4>("dep LZ4-on-text mix", wpc, 32 << 10, 64 << 10, iters, tables, windows, sink);
    return 0;
}
