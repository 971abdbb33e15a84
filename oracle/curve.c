/* oracle/curve.c — TEST INFRASTRUCTURE (CPU oracle). See curve.h. */
#include "curve.h"
#include <stdlib.h>

/* dbl-2009-l (a = 0) */
void g1j_dbl(g1j *r, const g1j *p) {
    if (g1j_is_inf(p)) { g1j_set_inf(r); return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fq_sqr(&a, &p->x);
    fq_sqr(&b, &p->y);
    fq_sqr(&c, &b);
    fq_add(&t, &p->x, &b);
    fq_sqr(&t, &t);
    fq_sub(&t, &t, &a);
    fq_sub(&t, &t, &c);
    fq_add(&d, &t, &t);
    fq_add(&e, &a, &a);
    fq_add(&e, &e, &a);
    fq_sqr(&f, &e);
    fq_add(&t, &d, &d);
    fq_sub(&x3, &f, &t);
    fq_mul(&z3, &p->y, &p->z);
    fq_add(&z3, &z3, &z3);
    fq_sub(&t, &d, &x3);
    fq_mul(&y3, &e, &t);
    fq_add(&c, &c, &c);
    fq_add(&c, &c, &c);
    fq_add(&c, &c, &c);
    fq_sub(&y3, &y3, &c);
    r->x = x3; r->y = y3; r->z = z3;
}

/* add-2007-bl, with the doubling / inverse special cases */
void g1j_add(g1j *r, const g1j *p, const g1j *q) {
    if (g1j_is_inf(p)) { *r = *q; return; }
    if (g1j_is_inf(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &p->z);
    fq_sqr(&z2z2, &q->z);
    fq_mul(&u1, &p->x, &z2z2);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s1, &p->y, &q->z);
    fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &q->y, &p->z);
    fq_mul(&s2, &s2, &z1z1);
    if (fp_eq(&u1, &u2)) {
        if (fp_eq(&s1, &s2)) { g1j_dbl(r, p); return; }
        g1j_set_inf(r);
        return;
    }
    fq_sub(&h, &u2, &u1);
    fq_add(&i, &h, &h);
    fq_sqr(&i, &i);
    fq_mul(&j, &h, &i);
    fq_sub(&rr, &s2, &s1);
    fq_add(&rr, &rr, &rr);
    fq_mul(&v, &u1, &i);
    fq_sqr(&x3, &rr);
    fq_sub(&x3, &x3, &j);
    fq_sub(&x3, &x3, &v);
    fq_sub(&x3, &x3, &v);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &rr, &t);
    fq_mul(&t, &s1, &j);
    fq_add(&t, &t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, &p->z, &q->z);
    fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &z2z2);
    fq_mul(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

/* madd-2007-bl */
void g1j_madd(g1j *r, const g1j *p, const g1a *q) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (g1j_is_inf(p)) { g1j_from_affine(r, q); return; }
    fe z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &p->z);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s2, &q->y, &p->z);
    fq_mul(&s2, &s2, &z1z1);
    if (fp_eq(&p->x, &u2)) {
        if (fp_eq(&p->y, &s2)) { g1j_dbl(r, p); return; }
        g1j_set_inf(r);
        return;
    }
    fq_sub(&h, &u2, &p->x);
    fq_sqr(&hh, &h);
    fq_add(&i, &hh, &hh);
    fq_add(&i, &i, &i);
    fq_mul(&j, &h, &i);
    fq_sub(&rr, &s2, &p->y);
    fq_add(&rr, &rr, &rr);
    fq_mul(&v, &p->x, &i);
    fq_sqr(&x3, &rr);
    fq_sub(&x3, &x3, &j);
    fq_sub(&x3, &x3, &v);
    fq_sub(&x3, &x3, &v);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &rr, &t);
    fq_mul(&t, &p->y, &j);
    fq_add(&t, &t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, &p->z, &h);
    fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &hh);
    r->x = x3; r->y = y3; r->z = z3;
}

void g1j_neg(g1j *r, const g1j *p) {
    r->x = p->x; r->z = p->z;
    fq_neg(&r->y, &p->y);
}

void g1j_to_affine(g1a *r, const g1j *p) {
    if (g1j_is_inf(p)) { memset(r, 0, sizeof(*r)); return; }
    fe zi, zi2, zi3;
    fq_inv(&zi, &p->z);
    fq_sqr(&zi2, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(&r->x, &p->x, &zi2);
    fq_mul(&r->y, &p->y, &zi3);
}

/* Montgomery's trick over the z coordinates */
void g1j_batch_to_affine(g1a *r, const g1j *p, size_t n) {
    if (n == 0) return;
    fe *pre = (fe *)malloc(n * sizeof(fe));
    fe acc;
    fq_one(&acc);
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!g1j_is_inf(&p[i])) fq_mul(&acc, &acc, &p[i].z);
    }
    fe inv;
    fq_inv(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (g1j_is_inf(&p[i])) { memset(&r[i], 0, sizeof(g1a)); continue; }
        fe zi, zi2, zi3;
        fq_mul(&zi, &inv, &pre[i]);
        fq_mul(&inv, &inv, &p[i].z);
        fq_sqr(&zi2, &zi);
        fq_mul(&zi3, &zi2, &zi);
        fq_mul(&r[i].x, &p[i].x, &zi2);
        fq_mul(&r[i].y, &p[i].y, &zi3);
    }
    free(pre);
}

int g1a_on_curve(const g1a *p) {
    if (g1a_is_inf(p)) return 1;
    fe l, rr, three;
    fq_sqr(&l, &p->y);
    fq_sqr(&rr, &p->x);
    fq_mul(&rr, &rr, &p->x);
    fp_from_u64(&three, 3, &FQ);
    fq_add(&rr, &rr, &three);
    return fp_eq(&l, &rr);
}

void g1j_mul_raw(g1j *r, const g1j *p, const uint64_t k[4]) {
    g1j acc;
    g1j_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_dbl(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g1j_add(&acc, &acc, p);
    }
    *r = acc;
}

void g1j_mul(g1j *r, const g1j *p, const fe *k) {
    uint64_t raw[4];
    fp_to_raw(raw, k, &FR);
    g1j_mul_raw(r, p, raw);
}

int g1j_eq(const g1j *a, const g1j *b) {
    if (g1j_is_inf(a) || g1j_is_inf(b)) return g1j_is_inf(a) && g1j_is_inf(b);
    fe z1z1, z2z2, u1, u2, s1, s2;
    fq_sqr(&z1z1, &a->z);
    fq_sqr(&z2z2, &b->z);
    fq_mul(&u1, &a->x, &z2z2);
    fq_mul(&u2, &b->x, &z1z1);
    fq_mul(&s1, &a->y, &b->z);
    fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &b->y, &a->z);
    fq_mul(&s2, &s2, &z1z1);
    return fp_eq(&u1, &u2) && fp_eq(&s1, &s2);
}

void g1a_to_be64(uint8_t out[64], const g1a *p) {
    fp_to_be32(out, &p->x, &FQ);
    fp_to_be32(out + 32, &p->y, &FQ);
}

/* non-inline export for the ctypes binding */
void g1j_from_affine_export(g1j *r, const g1a *p) { g1j_from_affine(r, p); }
