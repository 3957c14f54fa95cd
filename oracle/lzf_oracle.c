/*
 * lzf_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of what
 * lzf_compress(in, n, out, n-1) of the reference's liblzf computes
 * (call sites src/hashandcompress/HashAndCompress.cpp:344-347,
 * src/compression_perf/src/experiment.cpp:110; declaration
 * src/compression_perf/include/lzf/lzf.h:79-81; build configuration
 * src/compression_perf/include/lzf/lzfP.h:54-92: HLOG 16, VERY_FAST 1,
 * ULTRA_FAST 0, INIT_HTAB 0, offsets in the table).
 *
 * The compressor source is NOT in the reference tree (only a prebuilt liblzf.a,
 * never linked or run here), so this follows the LZF stream format and the
 * parser behaviour specified in SURVEY.md 8(a) row A6, with the hash table taken
 * as zero-initialised (hard part 4: the archive leaves it
 * uninitialised; stale entries are validated, and none of 2409 corpus blocks
 * differed between stale/zeroed tables when the survey probed it).
 *
 * Pinning: the reference holds no LZF vectors ("parity unpinned by the reference
 * itself"); tests/test_oracle_lzf.py pins this file to the outputs SURVEY.md 8(c)
 * recorded from liblzf.a (5 sizes, one SHA-256, 4 corpus totals) + round trips.
 */
#include "cw_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { LZF_HLOG = 16, LZF_MAX_LIT = 32, LZF_MAX_OFF = 8192, LZF_MAX_REF = 264 };

static inline uint32_t lzf_slot(uint32_t hval)
{
    /* VERY_FAST index: ((h >> (24 - HLOG)) - h*5) & (HSIZE-1) */
    return ((hval >> (24 - LZF_HLOG)) - hval * 5u) & ((1u << LZF_HLOG) - 1u);
}

size_t cw_oracle_lzf_compress(const uint8_t *in, size_t n, uint8_t *out, size_t cap)
{
    uint32_t *tab;
    size_t ip = 0, op = 0;
    uint32_t hval;
    int lit = 0;
    size_t result = 0;

    if (n == 0 || cap == 0) return 0;
    tab = (uint32_t *)calloc((size_t)1 << LZF_HLOG, sizeof(uint32_t));
    if (!tab) return 0;

    op++; /* reserve the literal-run control byte */
    if (n >= 2) {
        hval = ((uint32_t)in[0] << 8) | in[1];
        while (ip + 2 < n) {
            uint32_t slot, ref;
            size_t off;
            hval = (hval << 8) | in[ip + 2];
            slot = lzf_slot(hval);
            ref = tab[slot];
            tab[slot] = (uint32_t)ip;

            off = ip - ref - 1;
            if (ref < ip && off < LZF_MAX_OFF && ref > 0 &&
                in[ref + 2] == in[ip + 2] && in[ref] == in[ip] && in[ref + 1] == in[ip + 1]) {
                size_t len = 2, maxlen = n - ip - 2;
                if (maxlen > LZF_MAX_REF) maxlen = LZF_MAX_REF;

                if (op + 3 + 1 >= cap && op - (lit == 0) + 3 + 1 >= cap) goto done;

                out[op - (size_t)lit - 1] = (uint8_t)(lit - 1);  /* close the literal run */
                if (lit == 0) op--;                              /* ... or drop an empty one */

                if (maxlen > 16) {
                    /* 16 unrolled, un-bounded compares: may overshoot maxlen near the tail */
                    int k, stop = 0;
                    for (k = 0; k < 16; k++) {
                        len++;
                        if (in[ref + len] != in[ip + len]) { stop = 1; break; }
                    }
                    if (!stop) {
                        do len++; while (len < maxlen && in[ref + len] == in[ip + len]);
                    }
                } else {
                    do len++; while (len < maxlen && in[ref + len] == in[ip + len]);
                }

                len -= 2; /* now (match bytes - 1) - 1 ... i.e. octets - 2 */
                ip++;
                if (len < 7) {
                    out[op++] = (uint8_t)((off >> 8) + (len << 5));
                } else {
                    out[op++] = (uint8_t)((off >> 8) + (7u << 5));
                    out[op++] = (uint8_t)(len - 7);
                }
                out[op++] = (uint8_t)off;

                lit = 0; op++; /* open a fresh literal run */
                ip += len + 1;
                if (ip + 2 >= n) break;

                /* VERY_FAST: re-insert only the last two positions of the match */
                ip -= 2;
                hval = ((uint32_t)in[ip] << 8) | in[ip + 1];
                hval = (hval << 8) | in[ip + 2];
                tab[lzf_slot(hval)] = (uint32_t)ip;
                ip++;
                hval = (hval << 8) | in[ip + 2];
                tab[lzf_slot(hval)] = (uint32_t)ip;
                ip++;
            } else {
                if (op >= cap) goto done;
                lit++;
                out[op++] = in[ip++];
                if (lit == LZF_MAX_LIT) {
                    out[op - (size_t)lit - 1] = (uint8_t)(lit - 1);
                    lit = 0; op++;
                }
            }
        }
    }
    if (op + 3 > cap) goto done; /* at most 3 bytes can still be missing */
    while (ip < n) {
        lit++;
        out[op++] = in[ip++];
        if (lit == LZF_MAX_LIT) {
            out[op - (size_t)lit - 1] = (uint8_t)(lit - 1);
            lit = 0; op++;
        }
    }
    out[op - (size_t)lit - 1] = (uint8_t)(lit - 1);
    if (lit == 0) op--;
    result = op;
done:
    free(tab);
    return result;
}

/* LZF stream decoder (format: ctrl<32 -> ctrl+1 literals; else back-reference). */
long cw_oracle_lzf_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap)
{
    size_t ip = 0, op = 0;
    while (ip < n) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            size_t run = (size_t)ctrl + 1;
            if (ip + run > n || op + run > cap) return -1;
            memcpy(out + op, in + ip, run);
            ip += run; op += run;
        } else {
            size_t len = ctrl >> 5, off;
            if (ip >= n) return -1;
            if (len == 7) { len += in[ip++]; if (ip >= n) return -1; }
            off = ((size_t)(ctrl & 0x1f) << 8) | in[ip++];
            off += 1;
            len += 2;
            if (off > op || op + len > cap) return -1;
            for (size_t k = 0; k < len; k++) out[op + k] = out[op + k - off];
            op += len;
        }
    }
    return (long)op;
}
