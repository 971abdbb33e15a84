/*
 * oracle/curve.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * BN254 G1 (y^2 = x^3 + 3, /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:94)
 * in affine and Jacobian coordinates, restating what halo2curves 0.6.1's bn256::G1
 * computes (un-vendored dependency, Cargo.lock:2367-2368). The identity is encoded
 * as affine (0,0), as in the reference's RawBytes SRS files.
 */
#ifndef ORACLE_CURVE_H
#define ORACLE_CURVE_H

#include "field.h"

typedef struct { fe x, y; } g1a;        /* affine, Montgomery coords; (0,0) = identity */
typedef struct { fe x, y, z; } g1j;     /* Jacobian; z = 0 = identity */

static inline int g1a_is_inf(const g1a *p) { return fp_is_zero(&p->x) && fp_is_zero(&p->y); }
static inline int g1j_is_inf(const g1j *p) { return fp_is_zero(&p->z); }
static inline void g1j_set_inf(g1j *p) { memset(p, 0, sizeof(*p)); }
static inline void g1j_from_affine(g1j *r, const g1a *p) {
    if (g1a_is_inf(p)) { g1j_set_inf(r); return; }
    r->x = p->x; r->y = p->y; fq_one(&r->z);
}

void g1j_dbl(g1j *r, const g1j *p);
void g1j_add(g1j *r, const g1j *p, const g1j *q);
void g1j_madd(g1j *r, const g1j *p, const g1a *q);       /* mixed addition */
void g1j_neg(g1j *r, const g1j *p);
void g1j_to_affine(g1a *r, const g1j *p);
void g1j_batch_to_affine(g1a *r, const g1j *p, size_t n);
int  g1a_on_curve(const g1a *p);
/* scalar given as canonical 4x64 LE integer */
void g1j_mul_raw(g1j *r, const g1j *p, const uint64_t k[4]);
/* scalar given as Montgomery Fr */
void g1j_mul(g1j *r, const g1j *p, const fe *k);
int  g1j_eq(const g1j *a, const g1j *b);

/* 64-byte big-endian x||y (EVM / transcript encoding,
 * /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:89-99) */
void g1a_to_be64(uint8_t out[64], const g1a *p);

#endif
