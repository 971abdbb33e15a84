/*
 * oracle/hashes.c — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 *  - Keccak-256 (pre-NIST padding 0x01): the EVM transcript hash the reference's
 *    verifier pins (/root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:105-124);
 *    KATs: /root/reference/crates/shielder-account/src/secrets.rs:75-106.
 *  - ChaCha20 block function (RFC 8439 state layout, 64-bit counter in words 12..13):
 *    the blinding-randomness expander of this build's ABI (Fr draw i = from_u512 of block i).
 *  - Poseidon2 permutation t=8, alpha=7, R_F=8, R_P=48 over BN254 Fr and the
 *    arity-N sponge `hash` with capacity N*2^64, restating the round structure of
 *    /root/reference/poseidon2-solidity/generate_t8.py:474-588.
 */
#include "oracle.h"
#include "poseidon2_constants.h"

/* ---------------------------------------------------------------- Keccak */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[5][5] = {   /* [x][y] */
    {0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
static inline uint64_t rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

static void keccak_f(uint64_t a[25]) {      /* a[x + 5y] */
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], KROT[x][y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KRC[rnd];
    }
}

void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint64_t st[25] = {0};
    const size_t rate = 136;
    uint8_t blk[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; }
        keccak_f(st);
        in += rate; len -= rate;
    }
    memset(blk, 0, rate);
    memcpy(blk, in, len);
    blk[len] ^= 0x01;
    blk[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
    keccak_f(st);
    memcpy(out, st, 32);
}

/* ---------------------------------------------------------------- ChaCha20 */
static inline uint32_t rol32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
#define QR(a, b, c, d) \
    a += b; d ^= a; d = rol32(d, 16); c += d; b ^= c; b = rol32(b, 12); \
    a += b; d ^= a; d = rol32(d, 8);  c += d; b ^= c; b = rol32(b, 7);

void orc_chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865; s[1] = 0x3320646e; s[2] = 0x79622d32; s[3] = 0x6b206574;
    memcpy(&s[4], key, 32);
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = 0; s[15] = 0;
    memcpy(x, s, 64);
    for (int i = 0; i < 10; i++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13])
        QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
        QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) x[i] += s[i];
    memcpy(out, x, 64);
}

/* Fr draw #counter of the stream keyed by `seed`: 64 bytes -> wide reduction */
void orc_fr_draw(const uint8_t seed[32], uint64_t counter, fe *out) {
    uint8_t blk[64];
    orc_chacha20_block(seed, counter, blk);
    fp_from_u512(out, blk, &FR);
}

/* ---------------------------------------------------------------- Poseidon2 */
static fe RCF[8][8], RCP[48], DIAG[8];
static int p2_ready = 0;
static void p2_init(void) {
    if (p2_ready) return;
    for (int r = 0; r < 8; r++)
        for (int i = 0; i < 8; i++) fp_from_raw(&RCF[r][i], P2_RC_FULL[r][i], &FR);
    for (int r = 0; r < 48; r++) fp_from_raw(&RCP[r], P2_RC_PART[r], &FR);
    for (int i = 0; i < 8; i++) fp_from_raw(&DIAG[i], P2_DIAG[i], &FR);
    p2_ready = 1;
}

static inline void sbox7(fe *x) {
    fe x2, x4, x6;
    fr_sqr(&x2, x);
    fr_sqr(&x4, &x2);
    fr_mul(&x6, &x4, &x2);
    fr_mul(x, &x6, x);
}
/* the 4x4 block of generate_t8.py:476-492 (mm4), written out additively */
static void mm4(fe *a, fe *b, fe *c, fe *d) {
    fe t0, t1, t2, t3, t4, t5;
    fr_add(&t0, a, b);
    fr_add(&t1, c, d);
    fr_add(&t2, b, b); fr_add(&t2, &t2, &t1);
    fr_add(&t3, d, d); fr_add(&t3, &t3, &t0);
    fr_add(&t4, &t1, &t1); fr_add(&t4, &t4, &t4); fr_add(&t4, &t4, &t3);
    fr_add(&t5, &t0, &t0); fr_add(&t5, &t5, &t5); fr_add(&t5, &t5, &t2);
    fr_add(a, &t3, &t5);
    *b = t5;
    fr_add(c, &t2, &t4);
    *d = t4;
}
/* external linear layer, generate_t8.py:494-515 (fr_mm) */
static void ext_layer(fe s[8]) {
    mm4(&s[0], &s[1], &s[2], &s[3]);
    mm4(&s[4], &s[5], &s[6], &s[7]);
    fe w[4];
    for (int i = 0; i < 4; i++) fr_add(&w[i], &s[i], &s[i + 4]);
    for (int i = 0; i < 4; i++) {
        fr_add(&s[i], &s[i], &w[i]);
        fr_add(&s[i + 4], &s[i + 4], &w[i]);
    }
}

void orc_poseidon2_permute(fe s[8]) {
    p2_init();
    ext_layer(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 8; i++) { fr_add(&s[i], &s[i], &RCF[r][i]); sbox7(&s[i]); }
        ext_layer(s);
    }
    for (int r = 0; r < 48; r++) {
        fr_add(&s[0], &s[0], &RCP[r]);
        sbox7(&s[0]);
        fe sum = s[0];
        for (int i = 1; i < 8; i++) fr_add(&sum, &sum, &s[i]);
        for (int i = 0; i < 8; i++) {
            fe t;
            fr_mul(&t, &DIAG[i], &s[i]);
            fr_add(&s[i], &t, &sum);
        }
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 8; i++) { fr_add(&s[i], &s[i], &RCF[r][i]); sbox7(&s[i]); }
        ext_layer(s);
    }
}

/* hash of `arity` (1..7) inputs: state = [x0..x_{arity-1}, 0.., arity * 2^64]
 * (generate_t8.py:521-531 for arity 7; for arity < 7 the same rule is ASSUMED -- halo2 poseidon's ConstantLength<N> padding --
 * and pinned by no reference vector), output lane 0 */
void orc_poseidon2_hash(const fe *in, int arity, fe *out) {
    fe s[8];
    memset(s, 0, sizeof(s));
    for (int i = 0; i < arity; i++) s[i] = in[i];
    uint64_t cap[4] = {0, (uint64_t)arity, 0, 0};
    fp_from_raw(&s[7], cap, &FR);
    orc_poseidon2_permute(s);
    *out = s[0];
}
