// ubench.hip -- VALU issue-rate micro-benchmark for the integer ops the hash kernels are made of (gfx950).
// One workgroup on one CU, W waves per SIMD (blockDim = 256*W), each lane runs ITER x 32 independent-chain ops.
// Prints cycles per wave-instruction per SIMD (s_memtime ticks = shader cycles).
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"

#define ITER 2048

template <int OP>
__global__ void k(uint64_t *out, uint64_t *cycles, uint64_t seed)
{
    uint64_t a[8];
    uint32_t c[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1 + i); c[i] = (uint32_t)(a[i] >> 7); }
    const uint64_t inc = seed | 1;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] += a[(i + 1) & 7] ^ 0;                        // 64-bit add (hipcc: v_lshl_add_u64)
                if (OP == 1) c[i] = __builtin_amdgcn_alignbit(c[i], c[(i + 1) & 7], 13); // v_alignbit_b32
                if (OP == 2) c[i] ^= c[(i + 1) & 7];                            // v_xor_b32
                if (OP == 3) c[i] += c[(i + 1) & 7];                            // v_add_u32
                if (OP == 4) c[i] = c[i] * c[(i + 1) & 7];                         // v_mul_lo_u32
                if (OP == 5) {                                                  // explicit add_co/addc pair
                    uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32), bl = (uint32_t)a[(i + 1) & 7], bh = (uint32_t)(a[(i + 1) & 7] >> 32);
                    uint32_t rl, rh;
                    asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc" : "=&v"(rl), "=v"(rh) : "v"(lo), "v"(hi), "v"(bl), "v"(bh) : "vcc");
                    a[i] = ((uint64_t)rh << 32) | rl;
                }
                if (OP == 6) { a[i] = __builtin_rotateleft64(a[i], 13) ^ a[(i + 1) & 7]; } // hipcc's own 64-bit rotate + xor
                if (OP == 7) c[i] = __builtin_amdgcn_perm(c[i], c[(i + 1) & 7], 0x02010003u); // v_perm_b32
                if (OP == 8) c[i] = (c[i] & c[(i + 1) & 7]) | (~c[i] & c[(i + 2) & 7]);        // v_bfi_b32
                if (OP == 9) c[i] = c[i] + c[(i + 1) & 7] + c[(i + 2) & 7];                    // v_add3_u32
                if (OP == 10) c[i] = c[i] ^ c[(i + 1) & 7] ^ c[(i + 2) & 7];                   // v_xor3_b32
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int instr_per_op)
{
    uint64_t *out, *cyc;
    hipMalloc(&out, (size_t)1024 * 8 * 1024);
    hipMalloc(&cyc, 8 * 1024);
    // whole chip: 256 CUs x (w waves per SIMD); wall-clock -> absolute wave-instructions per second per SIMD
    for (int w = 2; w <= 8; w *= 2) {
        const int wg_threads = w >= 4 ? 1024 : 256 * w, wgs = 256 * (w >= 4 ? w / 4 : 1);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(wg_threads), 0, 0, out, cyc, 0x9E3779B97F4A7C15ULL);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(wg_threads), 0, 0, out, cyc, 0x9E3779B97F4A7C15ULL);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double instr_per_simd = (double)ITER * 32 * instr_per_op * w;
        printf("%-26s waves/SIMD=%d  ticks/instr/SIMD=%.2f  wall: %.3f ms -> %.2f G wave-instr/s/SIMD (ns per instr %.3f)\n", name, w,
               (double)c / instr_per_simd, ms, instr_per_simd / (ms * 1e-3) / 1e9, ms * 1e6 / instr_per_simd);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("u64 add (v_lshl_add_u64)", 1);
    run<5>("v_add_co+v_addc pair", 2);
    run<1>("v_alignbit_b32", 1);
    run<2>("v_xor_b32", 1);
    run<3>("v_add_u32", 1);
    run<4>("v_mul_lo_u32", 1);
    run<6>("hipcc rotl64^ (4-6 instr)", 6);
    run<7>("v_perm_b32", 1);
    run<8>("v_bfi_b32", 1);
    run<9>("v_add3_u32", 1);
    run<10>("v_xor3_b32", 1);
    return 0;
}
