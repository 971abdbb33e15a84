/*
 * oracle/vec.c — TEST INFRASTRUCTURE (CPU oracle), not product code.
 * Element-wise Fr vector primitives (OpenMP) used by oracle/plonk.py, the restatement of
 * halo2_proofs v0.3.0 create_proof: the Python side only sequences
 * them, every field operation runs here.
 */
#include "oracle.h"
#include <stdlib.h>

#define PAR _Pragma("omp parallel for schedule(static) if (n >= 2048)")

void orc_vec_add(fe *d, const fe *a, const fe *b, size_t n) { PAR for (size_t i = 0; i < n; i++) fr_add(&d[i], &a[i], &b[i]); }
void orc_vec_sub(fe *d, const fe *a, const fe *b, size_t n) { PAR for (size_t i = 0; i < n; i++) fr_sub(&d[i], &a[i], &b[i]); }
void orc_vec_mul(fe *d, const fe *a, const fe *b, size_t n) { PAR for (size_t i = 0; i < n; i++) fr_mul(&d[i], &a[i], &b[i]); }
void orc_vec_neg(fe *d, const fe *a, size_t n) { PAR for (size_t i = 0; i < n; i++) fr_neg(&d[i], &a[i]); }
void orc_vec_scale(fe *d, const fe *a, const fe *c, size_t n) { PAR for (size_t i = 0; i < n; i++) fr_mul(&d[i], &a[i], c); }
void orc_vec_addc(fe *d, const fe *a, const fe *c, size_t n) { PAR for (size_t i = 0; i < n; i++) fr_add(&d[i], &a[i], c); }
void orc_vec_fill(fe *d, const fe *c, size_t n) { PAR for (size_t i = 0; i < n; i++) d[i] = *c; }
/* d = a * y + b  (the quotient fold) */
void orc_vec_fold(fe *d, const fe *a, const fe *y, const fe *b, size_t n) {
    PAR for (size_t i = 0; i < n; i++) { fe t; fr_mul(&t, &a[i], y); fr_add(&d[i], &t, &b[i]); }
}
/* d[i] = a[i] * t[i % period] */
void orc_vec_mul_periodic(fe *d, const fe *a, const fe *t, size_t period, size_t n) {
    PAR for (size_t i = 0; i < n; i++) fr_mul(&d[i], &a[i], &t[i % period]);
}
/* out[0] = start; out[i+1] = out[i] * m[i]  for i < n-1  (n outputs) */
void orc_running_product(fe *out, const fe *start, const fe *m, size_t n) {
    out[0] = *start;
    for (size_t i = 1; i < n; i++) fr_mul(&out[i], &out[i - 1], &m[i - 1]);
}
/* out[i] = base^i * first */
void orc_powers(fe *out, const fe *first, const fe *base, size_t n) {
    if (!n) return;
    out[0] = *first;
    for (size_t i = 1; i < n; i++) fr_mul(&out[i], &out[i - 1], base);
}
/* synthetic division: q(X) = (p(X) - p(a)) / (X - a); out has n-1 coefficients (halo2 kate_division) */
void orc_kate_div(const fe *p, size_t n, const fe *a, fe *out) {
    if (n < 2) return;
    fe acc = p[n - 1];
    for (size_t i = n - 1; i-- > 0;) {
        out[i] = acc;
        fe t;
        fr_mul(&t, &acc, a);
        fr_add(&acc, &p[i], &t);
    }
}
/* count consecutive draws of the blinding stream */
void orc_fr_draw_many(const uint8_t seed[32], uint64_t counter0, size_t count, fe *out) {
#pragma omp parallel for schedule(static) if (count >= 1024)
    for (size_t i = 0; i < count; i++) orc_fr_draw(seed, counter0 + i, &out[i]);
}
/* canonical little-endian bytes of n elements (for sorting / transcript) */
void orc_vec_to_le32(uint8_t *out, const fe *a, size_t n) { PAR for (size_t i = 0; i < n; i++) fp_to_le32(out + 32 * i, &a[i], &FR); }
void orc_vec_from_le32(fe *out, const uint8_t *in, size_t n) { PAR for (size_t i = 0; i < n; i++) fp_from_le32(&out[i], in + 32 * i, &FR); }
