/*
 * skein_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of Skein-256/-512 as the
 * reference vendors it (2008 NIST round-1 submission: rotation constants of
 * reference_code/skein/Optimized_64bit/skein.h:275-292, key-schedule parity
 * 0x5555555555555555 skein.h:196 -- NOT the later Skein 1.2/1.3 constants).
 *
 * Written in the textbook Threefish form (MIX on adjacent word pairs, then a word
 * permutation) rather than the reference's operand-renamed unrolled form; the
 * two are equivalent (reference loop form: Reference_Implementation/skein_block.c:127-219,
 * key injection :29-35; API: Optimized_64bit/skein.c:226-266 Init, :329-373 Update,
 * :377-408 Final; bit padding: KAT_MCT/src/SHA3api_ref.c Update()).
 *
 * Pinned by tests/test_oracle_skein.py against the reference's KAT files and
 * against oracle/_ref/libskein_ref.so (the reference's own C compiled in place).
 */
#include "cw_oracle.h"
#include <stdlib.h>
#include <string.h>

#define T1_FIRST   (1ULL << 62)
#define T1_FINAL   (1ULL << 63)
#define T1_BITPAD  (1ULL << 55)
#define T1_TYPE(t) ((uint64_t)(t) << 56)
enum { TYPE_CFG = 4, TYPE_MSG = 48, TYPE_OUT = 63 };

#define KS_PARITY 0x5555555555555555ULL
#define SCHEMA_VER ((1ULL << 32) | 0x33414853ULL) /* version 1, "SHA3" */

/* rotation amounts, [round mod 8][pair] */
static const unsigned char ROT4[8][2] = {
    {5, 56}, {36, 28}, {13, 46}, {58, 44}, {26, 20}, {53, 35}, {11, 42}, {59, 50}};
static const unsigned char ROT8[8][4] = {
    {38, 30, 50, 53}, {48, 20, 43, 31}, {34, 14, 15, 27}, {26, 12, 58, 7},
    {33, 49, 8, 42},  {39, 27, 41, 14}, {29, 26, 11, 9},  {33, 51, 39, 35}};
/* word permutation applied after each round: new[i] = old[PERM[i]] */
static const unsigned char PERM4[4] = {0, 3, 2, 1};
static const unsigned char PERM8[8] = {2, 1, 4, 7, 6, 5, 0, 3};

static inline uint64_t rotl64(uint64_t x, unsigned r) { return (x << r) | (x >> (64 - r)); }

static uint64_t load_le64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}

/*
 * One UBI step: chain <- Threefish_{chain, tweak}(block) XOR block.
 * nw = 4 or 8 state words, 72 rounds, a subkey every 4 rounds.
 */
/* trace (optional, the reference's SKEIN_DEBUG callouts skein.h:246-254 / skein_debug.h:13-44): the state after the
 * initial key injection, after every round and after every later key injection, in execution order: 1 + 72 + 18
 * records of nw words -- what KAT_MCT/skein_golden_kat_short_internals.txt lists per Threefish call */
static void ubi_block_trace(int nw, uint64_t *chain, uint64_t t0, uint64_t t1, const uint8_t *block, uint64_t *trace)
{
    uint64_t ks[9], ts[3], w[8], v[8], tmp[8];
    int i, d, s;

    ks[nw] = KS_PARITY;
    for (i = 0; i < nw; i++) { ks[i] = chain[i]; ks[nw] ^= chain[i]; }
    ts[0] = t0; ts[1] = t1; ts[2] = t0 ^ t1;
    for (i = 0; i < nw; i++) { w[i] = load_le64(block + 8 * i); v[i] = w[i]; }

    for (d = 0; d < 72; d++) {
        if ((d & 3) == 0) { /* subkey s = d/4 */
            s = d >> 2;
            for (i = 0; i < nw; i++) v[i] += ks[(s + i) % (nw + 1)];
            v[nw - 3] += ts[s % 3];
            v[nw - 2] += ts[(s + 1) % 3];
            v[nw - 1] += (uint64_t)s;
            if (trace) { memcpy(trace, v, sizeof(uint64_t) * (size_t)nw); trace += nw; }
        }
        for (i = 0; i < nw / 2; i++) {
            unsigned r = (nw == 4) ? ROT4[d & 7][i] : ROT8[d & 7][i];
            v[2 * i] += v[2 * i + 1];
            v[2 * i + 1] = rotl64(v[2 * i + 1], r) ^ v[2 * i];
        }
        for (i = 0; i < nw; i++) tmp[i] = v[(nw == 4) ? PERM4[i] : PERM8[i]];
        memcpy(v, tmp, sizeof(uint64_t) * (size_t)nw);
        if (trace) { memcpy(trace, v, sizeof(uint64_t) * (size_t)nw); trace += nw; }
    }
    s = 18;
    for (i = 0; i < nw; i++) v[i] += ks[(s + i) % (nw + 1)];
    v[nw - 3] += ts[s % 3];
    v[nw - 2] += ts[(s + 1) % 3];
    v[nw - 1] += (uint64_t)s;
    if (trace) memcpy(trace, v, sizeof(uint64_t) * (size_t)nw);

    for (i = 0; i < nw; i++) chain[i] = v[i] ^ w[i];
}

static void ubi_block(int nw, uint64_t *chain, uint64_t t0, uint64_t t1, const uint8_t *block)
{
    ubi_block_trace(nw, chain, t0, t1, block, NULL);
}

/* One Threefish call + feed-forward with its round-by-round state (debug aid; nw = 4 or 8).  key: nw words (updated to the
 * chaining output), tweak: 2 words, block: nw * 8 bytes, trace: (1 + 72 + 18) * nw words. */
void cw_oracle_threefish_trace(int nw, uint64_t *key, const uint64_t *tweak, const uint8_t *block, uint64_t *trace)
{
    ubi_block_trace(nw, key, tweak[0], tweak[1], block, trace);
}

static void skein_iv_tree(int nw, unsigned hash_bits, uint64_t tree_info, uint64_t *iv)
{
    uint8_t cfg[64];
    uint64_t words[3] = {SCHEMA_VER, hash_bits, tree_info /* 0 = sequential; leaf | node << 8 | maxLevel << 16 (skein.h:209-210) */};
    memset(cfg, 0, sizeof cfg);
    for (int k = 0; k < 3; k++)
        for (int b = 0; b < 8; b++) cfg[8 * k + b] = (uint8_t)(words[k] >> (8 * b));
    memset(iv, 0, sizeof(uint64_t) * (size_t)nw);
    /* config string is 32 bytes whatever the state size (skein.h SKEIN_CFG_STR_LEN) */
    ubi_block(nw, iv, 32, T1_FIRST | T1_FINAL | T1_TYPE(TYPE_CFG), cfg);
}

void cw_oracle_skein_iv(int nw, unsigned hash_bits, uint64_t *iv) { skein_iv_tree(nw, hash_bits, 0, iv); }

static int skein_generic(int nw, const uint8_t *msg, size_t msg_bits, unsigned hash_bits, uint8_t *out)
{
    const size_t bb = (size_t)nw * 8; /* block bytes */
    uint64_t chain[8], t0 = 0, t1 = T1_FIRST | T1_TYPE(TYPE_MSG);
    uint8_t last[64];
    size_t nbytes = (msg_bits + 7) >> 3, pos = 0, rem;
    unsigned out_bytes = (hash_bits + 7) >> 3, produced = 0;
    uint64_t ctr = 0;

    if (hash_bits == 0) return -1;
    cw_oracle_skein_iv(nw, hash_bits, chain);

    /* all blocks but the last: a full final block is held back so FINAL lands on data */
    while (nbytes - pos > bb) {
        t0 += bb;
        ubi_block(nw, chain, t0, t1, msg + pos);
        t1 &= ~T1_FIRST;
        pos += bb;
    }
    rem = nbytes - pos; /* 0 (empty message only) .. bb */
    memset(last, 0, sizeof last);
    if (rem) memcpy(last, msg + pos, rem);
    if (msg_bits & 7) { /* partial final byte: keep the top bits, append a 1 bit */
        uint8_t mask = (uint8_t)(1u << (7 - (msg_bits & 7)));
        last[rem - 1] = (uint8_t)((last[rem - 1] & (uint8_t)(0 - mask)) | mask);
        t1 |= T1_BITPAD;
    }
    t0 += rem;
    ubi_block(nw, chain, t0, t1 | T1_FINAL, last);

    /* output: Threefish in counter mode keyed by the final chaining value */
    while (produced < out_bytes) {
        uint64_t o[8];
        uint8_t cblk[64];
        unsigned n = out_bytes - produced;
        memcpy(o, chain, sizeof o);
        memset(cblk, 0, sizeof cblk);
        for (int b = 0; b < 8; b++) cblk[b] = (uint8_t)(ctr >> (8 * b));
        ubi_block(nw, o, 8, T1_FIRST | T1_FINAL | T1_TYPE(TYPE_OUT), cblk);
        if (n > bb) n = (unsigned)bb;
        for (unsigned k = 0; k < n; k++) out[produced + k] = (uint8_t)(o[k >> 3] >> (8 * (k & 7)));
        produced += n;
        ctr++;
    }
    return 0;
}

int cw_oracle_skein512(const uint8_t *msg, size_t msg_bits, unsigned hash_bits, uint8_t *out)
{
    return skein_generic(8, msg, msg_bits, hash_bits, out);
}

int cw_oracle_skein256(const uint8_t *msg, size_t msg_bits, unsigned hash_bits, uint8_t *out)
{
    return skein_generic(4, msg, msg_bits, hash_bits, out);
}

/*
 * Tree hashing.  Follows the reference's all-in-one Skein_TreeHash
 * (reference_code/skein/Additional_Implementations/skein_test.c:616-680; tree fields of the configuration block
 * skein.h:209-210, tree level in the tweak skein.h:148,161,241): leaves of blkBytes << leaf bytes are hashed as
 * independent UBI chains that start from the configuration result G with the leaf's byte offset as the initial
 * tweak position and tree level 1; their outputs are concatenated and hashed again in nodes of blkBytes << node
 * bytes at level 2, and so on, until one block is left -- or until level maxLevel, which hashes whatever is left in
 * one chain.  The output transform is applied to the last chaining value.  Byte-granular messages only.
 */
static void ubi_chain(int nw, uint64_t *chain, const uint64_t *g, const uint8_t *data, size_t n, uint64_t t0, unsigned level)
{
    const size_t bb = (size_t)nw * 8;
    uint64_t t1 = T1_FIRST | T1_TYPE(TYPE_MSG) | ((uint64_t)level << 48); /* level: tweak bits 112..118 */
    uint8_t last[64];
    size_t pos = 0, rem;
    memcpy(chain, g, sizeof(uint64_t) * (size_t)nw);
    while (n - pos > bb) {
        t0 += bb;
        ubi_block(nw, chain, t0, t1, data + pos);
        t1 &= ~T1_FIRST;
        pos += bb;
    }
    rem = n - pos;
    memset(last, 0, sizeof last);
    if (rem) memcpy(last, data + pos, rem);
    ubi_block(nw, chain, t0 + rem, t1 | T1_FINAL, last);
}

int cw_oracle_skein_tree(int nw, const uint8_t *msg, size_t len, unsigned hash_bits, unsigned leaf, unsigned node,
                         unsigned max_level, uint8_t *out)
{
    const size_t bb = (size_t)nw * 8;
    uint64_t g[8], s[8] = {0};
    size_t bcnt = len, cap = len + bb;
    uint8_t *m;
    unsigned height, out_bytes = (hash_bits + 7) >> 3, produced = 0;
    uint64_t ctr = 0;
    if ((nw != 4 && nw != 8) || hash_bits == 0 || leaf == 0 || node == 0 || max_level < 2 || leaf > 255 || node > 255 || max_level > 255)
        return -1;
    if (leaf > 40 || node > 40) return -1; /* shifts below */
    m = (uint8_t *)malloc(cap);
    if (!m) return -1;
    if (len) memcpy(m, msg, len);
    skein_iv_tree(nw, hash_bits, (uint64_t)leaf | ((uint64_t)node << 8) | ((uint64_t)max_level << 16), g);
    for (height = 0;; height++) {
        size_t node_len, src, dst;
        if (height && bcnt == bb) break;            /* one block left: its bytes are the last chaining value */
        if (height + 1 == max_level) {              /* last allowed level: one chain over everything that is left */
            ubi_chain(nw, s, g, m, bcnt, 0, height + 1);
            break;
        }
        node_len = bb << (height ? node : leaf);
        for (src = dst = 0; src <= bcnt;) {
            size_t n = bcnt - src;
            uint8_t blk[64];
            if (n > node_len) n = node_len;
            ubi_chain(nw, s, g, m + src, n, (uint64_t)src, height + 1);
            for (size_t k = 0; k < bb; k++) blk[k] = (uint8_t)(s[k >> 3] >> (8 * (k & 7)));
            memcpy(m + dst, blk, bb);               /* dst <= src: never overtakes unread input */
            dst += bb;
            src += n;
            if (src >= bcnt) break;                 /* (also ends the len == 0 case after one empty leaf) */
        }
        bcnt = dst;
    }
    while (produced < out_bytes) {
        uint64_t o[8];
        uint8_t cblk[64];
        unsigned n = out_bytes - produced;
        memcpy(o, s, sizeof o);
        memset(cblk, 0, sizeof cblk);
        for (int b = 0; b < 8; b++) cblk[b] = (uint8_t)(ctr >> (8 * b));
        ubi_block(nw, o, 8, T1_FIRST | T1_FINAL | T1_TYPE(TYPE_OUT), cblk);
        if (n > bb) n = (unsigned)bb;
        for (unsigned k = 0; k < n; k++) out[produced + k] = (uint8_t)(o[k >> 3] >> (8 * (k & 7)));
        produced += n;
        ctr++;
    }
    free(m);
    return 0;
}
