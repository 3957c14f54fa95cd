// parse_stamp.hip -- diagnostic build of the LZ4 fingerprint parser with in-kernel stamps (lz4_kernel.hip, CW_STAMP):
// prints, per phase, the shader cycles a sequence spends there.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCW_STAMP -o tools/parse_stamp tools/parse_stamp.hip && tools/parse_stamp <file> [block_bytes] [nblocks] [wpc]
#include "../compute_war_amd/csrc/lz4_kernel.hip"

#include <stdio.h>
#include <vector>

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "tests/golden/corpus/canterbury/lcet10.txt";
    const size_t bs = argc > 2 ? (size_t)atol(argv[2]) : 65536, nb = argc > 3 ? (size_t)atol(argv[3]) : 16384;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    std::vector<uint8_t> text;
    for (int c; (c = fgetc(f)) != EOF;) text.push_back((uint8_t)c);
    fclose(f);
    std::vector<uint8_t> host(bs * nb);
    for (size_t i = 0; i < host.size(); i++) host[i] = text[i % text.size()];
    const size_t stride = (bs + bs / 255 + 16 + 15) / 16 * 16;
    uint8_t *src, *dst; uint32_t *sizes;
    hipMalloc(&src, host.size()); hipMalloc(&dst, stride * nb); hipMalloc(&sizes, nb * 4);
    hipMemcpy(src, host.data(), host.size(), hipMemcpyHostToDevice);
    const char *names[8] = {"-", "window ready + hash", "exchange (2 x ds_mskor_rtn)", "candidate fetch + order check", "undo + scalar extension",
                            "generic search", "byte loops", "window move + emit"};
    for (int pass = 0; pass < 2; pass++) {
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(cw::g_stamp), zero, sizeof zero);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipError_t e = cw::lz4_launch(src, bs, bs, nb, dst, stride, sizes, 0);
        hipEventRecord(e1, 0);
        if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 2; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(cw::g_stamp), sizeof st);
        if (pass == 0) continue;
        const double seqs = (double)st[15];
        double tot = 0;
        for (int i = 1; i < 8; i++) tot += (double)st[i];
        printf("%zu blocks of %zu B, %.0f sequences, %.2f ms (stamped build), %.1f cycles per sequence in all\n", nb, bs, seqs, ms, tot / seqs);
        for (int i = 1; i < 8; i++) printf("  %-34s %8.1f cycles/seq  %5.1f %%\n", names[i], st[i] / seqs, 100.0 * st[i] / tot);
    }
    return 0;
}
