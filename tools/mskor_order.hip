// Checks, on the device, the LDS behaviour the LZ4/LZF parse kernels build on: ds_mskor_rtn_b32 (masked 16-bit
// exchange inside a 32-bit LDS word) applied by the 64 lanes of ONE instruction behaves as if the lanes ran in
// ascending order -- each lane gets back the value left by the latest earlier lane that addressed the same
// 16-bit slot.  (The kernels do not rely on it blindly: a returned value >= the lane's own position reroutes the
// block to the cut-based parser.)   hipcc --offload-arch=gfx950 -O2 -o tools/mskor_order tools/mskor_order.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k(const unsigned *slots, unsigned *out, unsigned rounds)
{
    __shared__ unsigned tab[4096];
    for (unsigned r = 0; r < rounds; r++) {
        __syncthreads();
        for (unsigned i = threadIdx.x; i < 4096; i += 64) tab[i] = 0;
        __syncthreads();
        const unsigned h = slots[(blockIdx.x * rounds + r) * 64 + threadIdx.x];
        const unsigned addr = (unsigned)(uintptr_t)tab + (h >> 1) * 4, sh = (h & 1) * 16;
        unsigned old;
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(old) : "v"(addr), "v"(0xFFFFu << sh), "v"((threadIdx.x + 1) << sh) : "memory");
        out[(blockIdx.x * rounds + r) * 64 + threadIdx.x] = (old >> sh) & 0xFFFF;
    }
}

int main()
{
    const unsigned blocks = 1024, rounds = 64, n = blocks * rounds * 64;
    unsigned *h = (unsigned *)malloc(n * 4), *o = (unsigned *)malloc(n * 4), *ds, *dout;
    srand(1);
    for (unsigned i = 0; i < n; i++) h[i] = (unsigned)rand() % (8u << ((i / 64) % 8)); // 8 .. 1024 distinct slots
    hipMalloc(&ds, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(ds, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, ds, dout, rounds);
    if (hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    unsigned long bad = 0, coll = 0;
    for (unsigned b = 0; b < n / 64; b++)
        for (unsigned j = 0; j < 64; j++) {
            unsigned want = 0;
            for (unsigned i = 0; i < j; i++) if (h[b * 64 + i] == h[b * 64 + j]) want = i + 1;
            coll += want != 0;
            bad += o[b * 64 + j] != want;
        }
    printf("mskor_order: %u batches, %lu same-slot successors, %lu out of lane order\n", n / 64, coll, bad);
    return bad ? 1 : 0;
}
