/*
 * oracle/field.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of BN254 base/scalar field arithmetic as used by the
 * reference's un-vendored dependency halo2curves 0.6.1 (Cargo.lock:2367-2368):
 * 4x64-bit little-endian limbs in Montgomery form (R = 2^256), the same
 * representation the reference's SRS files carry
 * (/root/reference/crates/powers-of-tau/lib.rs:92-109, 206-227).
 * Moduli: /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:222-223
 * and /root/reference/crates/shielder-account/src/secrets.rs:20-25.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;          /* one field element, Montgomery form */

typedef struct {
    uint64_t p[4];      /* modulus */
    uint64_t r[4];      /* R   mod p  (= Montgomery 1) */
    uint64_t r2[4];     /* R^2 mod p */
    uint64_t r3[4];     /* R^3 mod p */
    uint64_t inv;       /* -p^{-1} mod 2^64 */
} fparams;

static const fparams FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
    {0xb1cd6dafda1530dfULL, 0x62f210e6a7283db6ULL, 0xef7f0b0c0ada0afbULL, 0x20fd6e902d592544ULL},
    0x87d20782e4866389ULL};

static const fparams FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
    {0x5e94d8e1b4bf0040ULL, 0x2a489cbe1cfbb6b8ULL, 0x893cc664a19fcfedULL, 0x0cf8594b7fcc657cULL},
    0xc2e1f593efffffffULL};

static inline int fp_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fp_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }

/* a >= b on raw limbs */
static inline int raw_geq(const uint64_t *a, const uint64_t *b) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline uint64_t raw_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - b[i] - bw;
        r[i] = (uint64_t)t;
        bw = (t >> 64) & 1;
    }
    return (uint64_t)bw;
}
static inline uint64_t raw_add(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}

static inline void fp_add(fe *r, const fe *a, const fe *b, const fparams *P) {
    uint64_t t[4];
    uint64_t c = raw_add(t, a->l, b->l);
    if (c || raw_geq(t, P->p)) raw_sub(t, t, P->p);
    memcpy(r->l, t, 32);
}
static inline void fp_sub(fe *r, const fe *a, const fe *b, const fparams *P) {
    uint64_t t[4];
    if (raw_sub(t, a->l, b->l)) raw_add(t, t, P->p);
    memcpy(r->l, t, 32);
}
static inline void fp_neg(fe *r, const fe *a, const fparams *P) {
    if (fp_is_zero(a)) { memset(r, 0, sizeof(fe)); return; }
    uint64_t t[4];
    raw_sub(t, P->p, a->l);
    memcpy(r->l, t, 32);
}
static inline void fp_dbl(fe *r, const fe *a, const fparams *P) { fp_add(r, a, a, P); }

/* Montgomery product a*b/R mod p; schoolbook product then word-wise reduction
 * (deliberately not the CIOS interleaving the HIP path uses). Inputs may be any
 * 256-bit values with a*b < p*R. */
static inline void fp_mul(fe *r, const fe *a, const fe *b, const fparams *P) {
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[i] * b->l[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    for (int i = 0; i < 4; i++) {
        uint64_t m = t[i] * P->inv;
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)m * P->p[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int j = i + 4; c && j < 9; j++) {
            c += t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
    }
    uint64_t *hi = t + 4;
    if (t[8] || raw_geq(hi, P->p)) raw_sub(hi, hi, P->p);
    memcpy(r->l, hi, 32);
}
static inline void fp_sqr(fe *r, const fe *a, const fparams *P) { fp_mul(r, a, a, P); }

static inline void fp_one(fe *r, const fparams *P) { memcpy(r->l, P->r, 32); }
static inline void fp_zero(fe *r) { memset(r, 0, sizeof(fe)); }

/* canonical 4x64 LE integer -> Montgomery */
static inline void fp_from_raw(fe *r, const uint64_t raw[4], const fparams *P) {
    fe a, r2;
    memcpy(a.l, raw, 32);
    memcpy(r2.l, P->r2, 32);
    fp_mul(r, &a, &r2, P);
}
/* Montgomery -> canonical 4x64 LE integer */
static inline void fp_to_raw(uint64_t raw[4], const fe *a, const fparams *P) {
    fe one = {{1, 0, 0, 0}}, t;
    fp_mul(&t, a, &one, P);
    memcpy(raw, t.l, 32);
}
static inline void fp_from_u64(fe *r, uint64_t v, const fparams *P) {
    uint64_t raw[4] = {v, 0, 0, 0};
    fp_from_raw(r, raw, P);
}

/* a^e for a 256-bit exponent given as 4x64 LE limbs (square-and-multiply, MSB first) */
static inline void fp_pow(fe *r, const fe *a, const uint64_t e[4], const fparams *P) {
    fe acc;
    fp_one(&acc, P);
    for (int i = 255; i >= 0; i--) {
        fp_sqr(&acc, &acc, P);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, a, P);
    }
    *r = acc;
}
static inline void fp_pow_u64(fe *r, const fe *a, uint64_t e, const fparams *P) {
    uint64_t ee[4] = {e, 0, 0, 0};
    fp_pow(r, a, ee, P);
}
/* a^{-1} = a^{p-2} (0 -> 0, as halo2's invert().unwrap_or(0) in batch paths) */
static inline void fp_inv(fe *r, const fe *a, const fparams *P) {
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    raw_sub(e, P->p, two);
    fp_pow(r, a, e, P);
}

/* 32-byte little-endian canonical encoding <-> Montgomery
 * (/root/reference/crates/type-conversions/lib.rs:59-73). Returns 0 if >= p. */
static inline int fp_from_le32(fe *r, const uint8_t b[32], const fparams *P) {
    uint64_t raw[4];
    memcpy(raw, b, 32);
    if (raw_geq(raw, P->p)) return 0;
    fp_from_raw(r, raw, P);
    return 1;
}
static inline void fp_to_le32(uint8_t b[32], const fe *a, const fparams *P) {
    uint64_t raw[4];
    fp_to_raw(raw, a, P);
    memcpy(b, raw, 32);
}
static inline void fp_to_be32(uint8_t b[32], const fe *a, const fparams *P) {
    uint8_t le[32];
    fp_to_le32(le, a, P);
    for (int i = 0; i < 32; i++) b[i] = le[31 - i];
}
static inline int fp_from_be32(fe *r, const uint8_t b[32], const fparams *P) {
    uint8_t le[32];
    for (int i = 0; i < 32; i++) le[i] = b[31 - i];
    return fp_from_le32(r, le, P);
}
/* 64 uniform bytes (LE) -> field element: d0*R^2 + d1*R^3 in Montgomery arithmetic,
 * halo2curves' from_u512. */
static inline void fp_from_u512(fe *r, const uint8_t b[64], const fparams *P) {
    fe d0, d1, r2, r3, t0, t1;
    memcpy(d0.l, b, 32);
    memcpy(d1.l, b + 32, 32);
    memcpy(r2.l, P->r2, 32);
    memcpy(r3.l, P->r3, 32);
    fp_mul(&t0, &d0, &r2, P);
    fp_mul(&t1, &d1, &r3, P);
    fp_add(r, &t0, &t1, P);
}

#define fr_add(r, a, b) fp_add(r, a, b, &FR)
#define fr_sub(r, a, b) fp_sub(r, a, b, &FR)
#define fr_mul(r, a, b) fp_mul(r, a, b, &FR)
#define fr_sqr(r, a) fp_sqr(r, a, &FR)
#define fr_neg(r, a) fp_neg(r, a, &FR)
#define fr_inv(r, a) fp_inv(r, a, &FR)
#define fr_one(r) fp_one(r, &FR)
#define fq_add(r, a, b) fp_add(r, a, b, &FQ)
#define fq_sub(r, a, b) fp_sub(r, a, b, &FQ)
#define fq_mul(r, a, b) fp_mul(r, a, b, &FQ)
#define fq_sqr(r, a) fp_sqr(r, a, &FQ)
#define fq_neg(r, a) fp_neg(r, a, &FQ)
#define fq_inv(r, a) fp_inv(r, a, &FQ)
#define fq_one(r) fp_one(r, &FQ)

#endif
