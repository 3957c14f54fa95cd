// lds_bench.hip -- cost of the LDS operations the parse kernels are built on (gfx950): one wavefront per CU, random
// 16-bit slots in a 128 KiB table, ROUNDS x 4 operations with one wait per group of 4 (as lzf_links_kernel issues them).
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o tools/lds_bench tools/lds_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ROUNDS 4096

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t seed)
{
    extern __shared__ uint32_t tab[];
    for (uint32_t i = threadIdx.x; i < 32768; i += 64) tab[i] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)tab;
    uint32_t x = seed * (threadIdx.x + 1), acc = 0;
    for (int r = 0; r < ROUNDS; r++) {
        uint32_t a[4], m[4], d[4], o[4];
        for (int j = 0; j < 4; j++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t slot = x >> 16, sh = (slot & 1) * 16;
            a[j] = base + (slot >> 1) * 4; m[j] = 0xFFFFu << sh; d[j] = (r & 0xFFFF) << sh;
        }
        if (OP == 0)
            asm volatile("ds_mskor_rtn_b32 %0, %4, %8, %12\n\tds_mskor_rtn_b32 %1, %5, %9, %13\n\tds_mskor_rtn_b32 %2, %6, %10, %14\n\t"
                         "ds_mskor_rtn_b32 %3, %7, %11, %15\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        if (OP == 1)
            asm volatile("ds_max_rtn_u32 %0, %4, %8\n\tds_max_rtn_u32 %1, %5, %9\n\tds_max_rtn_u32 %2, %6, %10\n\t"
                         "ds_max_rtn_u32 %3, %7, %11\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        if (OP == 2)
            asm volatile("ds_wrxchg_rtn_b32 %0, %4, %8\n\tds_wrxchg_rtn_b32 %1, %5, %9\n\tds_wrxchg_rtn_b32 %2, %6, %10\n\t"
                         "ds_wrxchg_rtn_b32 %3, %7, %11\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        if (OP == 3)
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\t"
                         "ds_write_b32 %4, %8\n\tds_write_b32 %5, %9\n\tds_write_b32 %6, %10\n\tds_write_b32 %7, %11\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
        if (OP == 4) // one exchange per wait: the latency a parse batch sees
            for (int j = 0; j < 4; j++)
                asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o[j]) : "v"(a[j]), "v"(m[j]), "v"(d[j]) : "memory");
        acc += o[0] + o[1] + o[2] + o[3];
        x += acc & 1;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int OP>
void run(const char *name)
{
    uint32_t *out;
    hipMalloc(&out, 256 * 64 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64), 131072, 0, out, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64), 131072, 0, out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f ns per LDS instruction (64 lanes, incl. address arithmetic)\n", name, ms * 1e6 / (ROUNDS * 4.0));
    hipFree(out);
}

int main()
{
    run<0>("ds_mskor_rtn_b32 x4 per wait");
    run<1>("ds_max_rtn_u32 x4 per wait");
    run<2>("ds_wrxchg_rtn_b32 x4 per wait");
    run<3>("ds_read_b32 + ds_write_b32 (x4 each) per wait");
    run<4>("ds_mskor_rtn_b32, one per wait");
    return 0;
}
