// vtab_stamp.hip -- diagnostic build of the scalar-thread LZ4 parser (lz4_vtab_kernel.hip, CW_VSTAMP) with in-kernel stamps: prints, per
// phase, the shader cycles a sequence spends there, for the register-table form and the LDS-table form.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCW_VSTAMP -o tools/vtab_stamp tools/vtab_stamp.hip && tools/vtab_stamp <file> [nblocks] [wavefronts per CU]
#include "../compute_war_amd/csrc/lz4_vtab_kernel.hip"

#include <stdio.h>
#include <vector>

const char *cw::tune(const char *) { return nullptr; }

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "tests/golden/corpus/canterbury/lcet10.txt";
    const size_t bs = 65536, nb = argc > 2 ? (size_t)atol(argv[2]) : 4096;
    const unsigned wpc = argc > 3 ? (unsigned)atoi(argv[3]) : 16;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    std::vector<uint8_t> text;
    for (int c; (c = fgetc(f)) != EOF;) text.push_back((uint8_t)c);
    fclose(f);
    std::vector<uint8_t> host(bs * nb);
    for (size_t i = 0; i < host.size(); i++) host[i] = text[i % text.size()];
    const size_t stride = (bs + bs / 255 + 16 + 15) / 16 * 16;
    uint8_t *src, *dst; uint32_t *sizes, *queue, *counters;
    hipMalloc(&src, host.size()); hipMalloc(&dst, stride * nb); hipMalloc(&sizes, nb * 4); hipMalloc(&queue, nb * 4); hipMalloc(&counters, 32);
    hipMemcpy(src, host.data(), host.size(), hipMemcpyHostToDevice);
    std::vector<uint32_t> q(nb);
    for (size_t i = 0; i < nb; i++) q[i] = (uint32_t)i;
    hipMemcpy(queue, q.data(), nb * 4, hipMemcpyHostToDevice);
    const char *names[8] = {"window value, hash, loop control", "table exchange", "candidate + next window: loads and wait", "back extension (incl. its loads)",
                            "forward extension, end of match", "record the sequence (+ a batch of 64 written out)", "load the next windows and wait", "insert + window move"};
    for (int form = 0; form < 2; form++) {
        for (int pass = 0; pass < 2; pass++) {
#ifdef CW_VSTAMP
            unsigned long long zero[16] = {0};
            hipMemcpyToSymbol(HIP_SYMBOL(cw::g_vstamp), zero, sizeof zero);
#endif
            const uint32_t c0[8] = {0, (uint32_t)nb, 0, 0, 0, 0, 0, 0};
            hipMemcpy(counters, c0, sizeof c0, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            size_t grid = 256 * (size_t)(form ? (wpc < 10 ? wpc : 10) : wpc);
            if (grid > nb) grid = nb;
            if (form) hipLaunchKernelGGL(cw::lz4_vtab3_kernel<true>, dim3((unsigned)grid), dim3(64), 16384, 0, src, (uint32_t)bs, bs, dst, stride, sizes, queue, counters, 0u, 0xFFFFFFFFu, 0u);
            else hipLaunchKernelGGL(cw::lz4_vtab3_kernel<false>, dim3((unsigned)grid), dim3(64), 0, 0, src, (uint32_t)bs, bs, dst, stride, sizes, queue, counters, 0u, 0xFFFFFFFFu, 0u);
            hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long st[16] = {0};
#ifdef CW_VSTAMP
            hipMemcpyFromSymbol(st, HIP_SYMBOL(cw::g_vstamp), sizeof st);
#endif
            if (pass == 0) continue;
            std::vector<uint32_t> hs(nb);
            hipMemcpy(hs.data(), sizes, nb * 4, hipMemcpyDeviceToHost);
            unsigned long long total = 0, mix = 0;
            for (size_t i = 0; i < nb; i++) { total += hs[i]; mix = mix * 1000003ull + hs[i]; }
            std::vector<uint8_t> hd(stride * 64);
            hipMemcpy(hd.data(), dst, hd.size(), hipMemcpyDeviceToHost);
            unsigned long long pay = 0;
            for (size_t i = 0; i < 64; i++) for (uint32_t k = 0; k < hs[i]; k++) pay = pay * 1000003ull + hd[i * stride + k];
            static std::vector<uint32_t> first_sizes;
            if (first_sizes.empty()) first_sizes = hs;
            else {
                size_t bad = 0;
                for (size_t i = 0; i < nb; i++)
                    if (hs[i] != first_sizes[i]) { if (bad++ < 12) printf("  block %zu: %u bytes, register-table form %u\n", i, hs[i], first_sizes[i]); }
                printf("blocks whose size differs between the forms: %zu\n", bad);
            }
            printf("output: %llu bytes, sizes fold %016llx, payload fold of the first 64 blocks %016llx\n", total, mix, pay);
            const double seqs = (double)st[15], probes = (double)st[14];
            double tot = 0;
            for (int i = 0; i < 8; i++) tot += (double)st[i];
            printf("%s form, %zu blocks of %zu B on %zu wavefronts: %.0f sequences, %.2f probes per sequence, %.2f ms (stamped build: %.1f GB/s), %.0f cycles per sequence\n",
                   form ? "LDS-table" : "register-table", nb, bs, grid, seqs, probes / seqs, ms, nb * bs / ms / 1e6, tot / seqs);
            for (int i = 0; i < 8; i++) printf("  %-42s %8.1f cycles/seq  %5.1f %%\n", names[i], st[i] / seqs, 100.0 * st[i] / tot);
        }
    }
    return 0;
}
