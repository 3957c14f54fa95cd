/* TEST INFRASTRUCTURE: drives every oracle entry point natively (for the ASan/UBSan and TSan builds of tests/stub/Makefile):
 * round trips over structured, random and degenerate inputs of many sizes, the threaded worker loop against the serial
 * one.  Prints one line with a checksum of everything it produced; tests/test_sanitizers.py compares the two builds' lines. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/cw_oracle.h"

static uint64_t fold(uint64_t a, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; i++) a = (a ^ p[i]) * 0x100000001B3ULL;
    return a;
}

int main(void)
{
    enum { MAXN = 70000 };
    uint8_t *in = (uint8_t *)malloc(MAXN), *out = (uint8_t *)malloc(2 * MAXN + 64), *back = (uint8_t *)malloc(MAXN);
    uint8_t dig[64];
    uint64_t acc = 0xCBF29CE484222325ULL;
    const size_t sizes[] = {0, 1, 2, 3, 11, 12, 13, 14, 31, 32, 33, 63, 64, 65, 255, 256, 4095, 4096, 4097, 16384, 65535, 65536};
    for (unsigned kind = 0; kind < 4; kind++)
        for (size_t si = 0; si < sizeof sizes / sizeof sizes[0]; si++) {
            const size_t n = sizes[si];
            for (size_t i = 0; i < n; i++)
                in[i] = kind == 0 ? 0 : kind == 1 ? (uint8_t)(i * 131 + (i >> 8)) : kind == 2 ? (uint8_t)((i * 2654435761u) >> 13) : (uint8_t)"abcab"[i % 5];
            cw_oracle_skein512(in, n * 8, 512, dig); acc = fold(acc, dig, 64);
            cw_oracle_skein256(in, n * 8, 128, dig); acc = fold(acc, dig, 16);
            cw_oracle_sha256(in, n, dig); acc = fold(acc, dig, 32);
            size_t c = cw_oracle_lz4_compress(in, n, out, 2 * n > cw_oracle_lz4_bound(n) ? 2 * n : cw_oracle_lz4_bound(n));
            acc = fold(acc, out, c);
            if (n && cw_oracle_lz4_decompress(out, c, back, n) != (long)n) { printf("lz4 round trip failed at %zu\n", n); return 1; }
            if (n && memcmp(in, back, n)) { printf("lz4 round trip differs at %zu\n", n); return 1; }
            if (n >= 2) {
                c = cw_oracle_lzf_compress(in, n, out, n - 1);
                acc = fold(acc, out, c);
                if (c && (cw_oracle_lzf_decompress(out, c, back, n) != (long)n || memcmp(in, back, n))) { printf("lzf round trip failed at %zu\n", n); return 1; }
            }
        }
    { /* tree mode + the generators + the threaded worker loop against the serial one */
        enum { NB = 48, BS = 4096 };
        uint8_t *data = (uint8_t *)malloc(NB * BS), *d1 = (uint8_t *)malloc(NB * 64), *d4 = (uint8_t *)malloc(NB * 64);
        uint32_t z1[NB], z4[NB];
        cw_oracle_gen_mixed_blocks(0xC0FFEE, 3, NB, BS, data);
        cw_oracle_skein_tree(8, data, 8192, 512, 1, 1, 255, dig); acc = fold(acc, dig, 64);
        cw_oracle_hash_and_compress(data, NB, BS, CW_OR_HASH_SKEIN512, CW_OR_COMP_LZ4, 1, d1, NULL, 0, z1);
        cw_oracle_hash_and_compress(data, NB, BS, CW_OR_HASH_SKEIN512, CW_OR_COMP_LZ4, 4, d4, NULL, 0, z4);
        if (memcmp(d1, d4, NB * 64) || memcmp(z1, z4, sizeof z1)) { printf("threaded worker loop differs from the serial one\n"); return 1; }
        acc = fold(acc, d4, NB * 64); acc = fold(acc, (const uint8_t *)z4, sizeof z4);
        cw_oracle_hash_and_compress(data, NB, BS, CW_OR_HASH_SHA256, CW_OR_COMP_LZF, 3, d4, NULL, 0, z4);
        acc = fold(acc, d4, NB * 32); acc = fold(acc, (const uint8_t *)z4, sizeof z4);
        free(data); free(d1); free(d4);
    }
    printf("oracle selftest ok %016llx\n", (unsigned long long)acc);
    free(in); free(out); free(back);
    return 0;
}
