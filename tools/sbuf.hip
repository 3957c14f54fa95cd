// s_buffer_load_dwordx8 bounds behaviour on gfx950: per-dword range check against num_records, negative (wrapped) offsets
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(64) k(const uint8_t *src, uint32_t n, uint32_t *out)
{
    const uint64_t a = (uint64_t)src;
    u32x4 rs;
    rs.x = (uint32_t)a; rs.y = (uint32_t)(a >> 32) & 0xFFFFu; rs.z = n; rs.w = 0x00020000u;
    uint32_t offs[4] = {0u, n - 16u, n - 8u, 0xFFFFFFFCu};
    for (int i = 0; i < 4; i++) {
        u32x8 v;
        uint32_t o = __builtin_amdgcn_readfirstlane(offs[i]);
        asm volatile("s_buffer_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(rs), "s"(o));
        if (threadIdx.x == 0) for (int j = 0; j < 8; j++) out[i * 8 + j] = v[j];
    }
}
int main()
{
    const uint32_t n = 4096 - 4; // num_records a multiple of 4
    uint8_t *d; uint32_t *o, h[32];
    hipMalloc(&d, 8192); hipMalloc(&o, 128);
    uint32_t *hb = (uint32_t *)malloc(8192);
    for (int i = 0; i < 2048; i++) hb[i] = 0xA0000000u + i;
    hipMemcpy(d, hb, 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, o);
    hipMemcpy(h, o, 128, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) { printf("sbuf case %d:", i); for (int j = 0; j < 8; j++) printf(" %08x", h[i * 8 + j]); printf("\n"); }
    printf("expect case1: dwords at n-16.. = %08x.. then zeros past n; case3 (offset -4): zero then a0000000..?\n", 0xA0000000u + (n - 16) / 4);
    return 0;
}
