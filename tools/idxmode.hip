// does s_set_gpr_idx_on apply to v_readlane_b32 src0 / v_writelane_b32 vdst on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define TCLOB "v64","v65","v66","v67","v68","v69","v70","v71"
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(64))) k(uint32_t *out)
{
    const uint32_t lane = threadIdx.x;
    // v64+i = 1000*i + lane
    for (uint32_t i = 0; i < 8; i++) {
        uint32_t val = 1000 * i + lane;
        asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v64, %1\n\ts_set_gpr_idx_off" :: "s"(i), "v"(val) : TCLOB);
    }
    uint32_t res = 0;
    // read (reg 5, lane 17) with indexed readlane
    uint32_t r = 5, l = 17, sw;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_readlane_b32 %0, v64, %2\n\ts_set_gpr_idx_off" : "=s"(sw) : "s"(r), "s"(l) : TCLOB);
    if (lane == 0) out[0] = sw;            // expect 5017 if indexing applies, 17 if not
    // indexed writelane: (reg 3, lane 9) = 777
    uint32_t r2 = 3, l2 = 9, nw = 777;
    asm volatile("s_lshl_b64 exec, 1, %2\n\ts_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v64, %1\n\ts_set_gpr_idx_off\n\ts_mov_b64 exec, -1" :: "s"(r2), "s"(nw), "s"(l2) : TCLOB);
    uint32_t a, b;
    asm volatile("v_mov_b32 %0, v67\n\tv_mov_b32 %1, v64" : "=v"(a), "=v"(b) :: TCLOB);
    if (lane == 9) { out[1] = a; out[2] = b; }  // expect a = 777, b = 9 if indexing applies; a = 3009, b = 777 if not
}
int main()
{
    uint32_t *d, h[3];
    hipMalloc(&d, 12);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("idxmode: readlane=%u (5017 = indexed) writelane a=%u b=%u (777, 9 = indexed)\n", h[0], h[1], h[2]);
    return 0;
}
