/*
 * lz4_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of what
 * LZ4_compress_default(src, dst, n, 2*n) of LZ4 v1.8.2 computes for n < 65547
 * (the reference's call: src/hashandcompress/HashAndCompress.cpp:351-354,
 * src/compression_perf/src/experiment.cpp:249; declaration
 * src/compression_perf/include/lz4/lz4.h:126-139, version :94-96).
 *
 * The LZ4 source is NOT in the reference tree (only a prebuilt liblz4.a, which is
 * never linked or run here), so this follows the published LZ4 block format and
 * the v1.8.2 "fast" greedy parser as specified in SURVEY.md 8(a) row A5:
 * 16-bit position table of 8192 slots zeroed per call, acceleration 1, no
 * dictionary, unlimited output (2*n >= LZ4_compressBound(n)).
 *
 * Pinning: the reference holds no LZ4 test vectors ("parity unpinned by the
 * reference itself").  tests/test_oracle_lz4.py pins this file to the reference
 * outputs SURVEY.md 8(c) recorded from liblz4.a (5 sizes, SHA-256 of one 39,618-byte
 * output, 4 corpus totals via BASELINE.md section 2) and cross-checks round trips.
 */
#include "cw_oracle.h"
#include <string.h>

enum {
    MINMATCH = 4,
    LASTLITERALS = 5,
    MFLIMIT = 12,          /* a match may not start in the last 12 bytes */
    HASHLOG16 = 13,        /* byU16 table: 8192 slots */
    SKIP_TRIGGER = 6,
    LIMIT_64K = 65536 + MFLIMIT - 1
};

static inline uint32_t rd32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint32_t hash13(uint32_t v) { return (v * 2654435761u) >> (32 - HASHLOG16); }

size_t cw_oracle_lz4_bound(size_t n) { return n + n / 255 + 16; }

static uint8_t *put_len(uint8_t *op, size_t extra)
{
    while (extra >= 255) { *op++ = 255; extra -= 255; }
    *op++ = (uint8_t)extra;
    return op;
}

size_t cw_oracle_lz4_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
    uint16_t tab[1 << HASHLOG16];
    size_t ip = 0, anchor = 0, match = 0;
    uint8_t *op = dst, *token;
    uint32_t fh;

    if (n >= (size_t)LIMIT_64K) return 0;            /* oracle covers the byU16 regime only */
    if (cap < cw_oracle_lz4_bound(n)) return 0;       /* "notLimited" regime only */
    memset(tab, 0, sizeof tab);

    if (n < MFLIMIT + 1) goto last_literals;
    {
        const size_t mflimit = n - MFLIMIT;           /* last position a match may start */
        const size_t matchlimit = n - LASTLITERALS;

        tab[hash13(rd32(src))] = 0;
        ip = 1;
        fh = hash13(rd32(src + ip));

        for (;;) {
            /* search: probe ip, ip+1, ... with a stride that grows every 64 misses */
            {
                size_t fip = ip;
                unsigned step = 1, nb = 1u << SKIP_TRIGGER;
                do {
                    uint32_t h = fh;
                    ip = fip;
                    fip += step;
                    step = nb++ >> SKIP_TRIGGER;
                    if (fip > mflimit + 1) goto last_literals;  /* v1.8.2: forwardIp > mflimitPlusOne */
                    match = tab[h];                   /* empty slot == position 0 */
                    fh = hash13(rd32(src + fip));
                    tab[h] = (uint16_t)ip;
                } while (rd32(src + match) != rd32(src + ip));
            }
            /* extend backwards over the pending literals */
            while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }

            /* literal run */
            {
                size_t lit = ip - anchor;
                token = op++;
                if (lit >= 15) { *token = 15 << 4; op = put_len(op, lit - 15); }
                else *token = (uint8_t)(lit << 4);
                memcpy(op, src + anchor, lit);
                op += lit;
            }
        next_match:
            {
                size_t off = ip - match, mc = 0;
                const size_t a = ip + MINMATCH, b = match + MINMATCH;
                *op++ = (uint8_t)off; *op++ = (uint8_t)(off >> 8);
                while (a + mc < matchlimit && src[a + mc] == src[b + mc]) mc++;
                ip += MINMATCH + mc;
                if (mc >= 15) { *token += 15; op = put_len(op, mc - 15); }
                else *token += (uint8_t)mc;
            }
            anchor = ip;
            if (ip > mflimit) break;

            tab[hash13(rd32(src + ip - 2))] = (uint16_t)(ip - 2);

            /* immediate re-test at the new position */
            {
                uint32_t h = hash13(rd32(src + ip));
                match = tab[h];
                tab[h] = (uint16_t)ip;
                if (rd32(src + match) == rd32(src + ip)) { token = op++; *token = 0; goto next_match; }
            }
            fh = hash13(rd32(src + ++ip));
        }
    }
last_literals:
    {
        size_t run = n - anchor;
        if (run >= 15) { *op++ = 15 << 4; op = put_len(op, run - 15); }
        else *op++ = (uint8_t)(run << 4);
        memcpy(op, src + anchor, run);
        op += run;
    }
    return (size_t)(op - dst);
}

/* Straight LZ4 block decoder (format spec), bounds-checked like LZ4_decompress_safe. */
long cw_oracle_lz4_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
    size_t ip = 0, op = 0;
    if (n == 0) return -1;
    for (;;) {
        unsigned tok;
        size_t lit, ml, off;
        if (ip >= n) return -1;
        tok = src[ip++];
        lit = tok >> 4;
        if (lit == 15) {
            unsigned c;
            do { if (ip >= n) return -1; c = src[ip++]; lit += c; } while (c == 255);
        }
        if (ip + lit > n || op + lit > cap) return -1;
        memcpy(dst + op, src + ip, lit);
        ip += lit; op += lit;
        if (ip == n) break;                 /* last sequence: literals only */
        if (ip + 2 > n) return -1;
        off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        if (off == 0 || off > op) return -1;
        ml = tok & 15;
        if (ml == 15) {
            unsigned c;
            do { if (ip >= n) return -1; c = src[ip++]; ml += c; } while (c == 255);
        }
        ml += MINMATCH;
        if (op + ml > cap) return -1;
        for (size_t k = 0; k < ml; k++) dst[op + k] = dst[op + k - off];
        op += ml;
    }
    return (long)op;
}
