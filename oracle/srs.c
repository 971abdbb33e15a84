/*
 * oracle/srs.c — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * SRS ingest restating /root/reference/crates/powers-of-tau/lib.rs:
 *   read_header_size/read_k   lib.rs:76-85     (u64 @16, k = u32 @ 24+hdr-8)
 *   g1_offset / read_g1       lib.rs:121-123, 190-231  (2x32-byte LE *Montgomery* Fq per point)
 *   g2_offset / read_g2       lib.rs:111-119, 125-188  (first two G2 points: g2, s*g2)
 *   Format::Raw               lib.rs:64  (halo2 ParamsKZG RawBytes: u32 k | g[n] | g_lagrange[n] | g2 | s_g2,
 *                                          coordinates already Montgomery limbs)
 * and ParamsKZG::from_parts' derivation of g_lagrange (lib.rs:71).
 */
#include "oracle.h"
#include <stdlib.h>

static uint64_t rd_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

void orc_srs_free(orc_srs *s) {
    free(s->g);
    free(s->g_lagrange);
    memset(s, 0, sizeof(*s));
}

/* The file holds x*R mod q as a little-endian integer; lib.rs:206-227 reads it with
 * from_repr (canonical) and multiplies by R^{-1}: the stored integer IS the Montgomery
 * limb vector of x, so it is taken over unchanged after a range check. */
static int read_mont_fq(fe *out, const uint8_t *p) {
    memcpy(out->l, p, 32);
    return !raw_geq(out->l, FQ.p);
}

int orc_srs_from_ptau(const uint8_t *b, size_t len, orc_srs *out) {
    memset(out, 0, sizeof(*out));
    if (len < 64) return -1;
    uint64_t hdr = rd_u64(b + 16);
    if (24 + hdr > len) return -1;
    uint32_t k = rd_u32(b + 24 + hdr - 8);
    if (k > 28) return -1;
    size_t n = (size_t)1 << k;
    size_t g1_off = 24 + hdr + 12;
    size_t g2_off = g1_off + 64 * (2 * n - 1) + 12;
    if (g2_off + 256 > len) return -1;
    out->k = k; out->n = n;
    out->g = (g1a *)malloc(n * sizeof(g1a));
    out->g_lagrange = (g1a *)malloc(n * sizeof(g1a));
    for (size_t i = 0; i < n; i++) {
        if (!read_mont_fq(&out->g[i].x, b + g1_off + 64 * i) || !read_mont_fq(&out->g[i].y, b + g1_off + 64 * i + 32) ||
            !g1a_on_curve(&out->g[i])) {
            orc_srs_free(out);
            return -2;
        }
    }
    /* G2: stored Montgomery; exported canonical LE (x.c0, x.c1, y.c0, y.c1) */
    for (int pt = 0; pt < 2; pt++) {
        uint8_t *dst = pt == 0 ? out->g2 : out->s_g2;
        for (int c = 0; c < 4; c++) {
            fe v;
            if (!read_mont_fq(&v, b + g2_off + 128 * pt + 32 * c)) { orc_srs_free(out); return -2; }
            fp_to_le32(dst + 32 * c, &v, &FQ);
        }
    }
    orc_g1_lagrange_from_monomial(out->g, (int)k, out->g_lagrange);
    return 0;
}

int orc_srs_from_halo2_raw(const uint8_t *b, size_t len, orc_srs *out) {
    memset(out, 0, sizeof(*out));
    if (len < 4) return -1;
    uint32_t k = rd_u32(b);
    if (k > 28) return -1;
    size_t n = (size_t)1 << k;
    if (len != 4 + 2 * n * 64 + 256) return -1;
    out->k = k; out->n = n;
    out->g = (g1a *)malloc(n * sizeof(g1a));
    out->g_lagrange = (g1a *)malloc(n * sizeof(g1a));
    const uint8_t *p = b + 4;
    for (size_t i = 0; i < 2 * n; i++) {
        g1a *dst = i < n ? &out->g[i] : &out->g_lagrange[i - n];
        if (!read_mont_fq(&dst->x, p + 64 * i) || !read_mont_fq(&dst->y, p + 64 * i + 32) || !g1a_on_curve(dst)) {
            orc_srs_free(out);
            return -2;
        }
    }
    p += 2 * n * 64;
    for (int pt = 0; pt < 2; pt++) {
        uint8_t *dst = pt == 0 ? out->g2 : out->s_g2;
        for (int c = 0; c < 4; c++) {
            fe v;
            if (!read_mont_fq(&v, p + 128 * pt + 32 * c)) { orc_srs_free(out); return -2; }
            fp_to_le32(dst + 32 * c, &v, &FQ);
        }
    }
    return 0;
}

/* fixed-base table for the generator (1,2): T[w][d] = d * 2^(8w) * G */
static void fixed_base_mul_many(const fe *scalars, size_t n, g1a *out) {
    g1a gen;
    fp_from_u64(&gen.x, 1, &FQ);
    fp_from_u64(&gen.y, 2, &FQ);
    g1j *tj = (g1j *)malloc(32 * 256 * sizeof(g1j));
    g1j base;
    g1j_from_affine(&base, &gen);
    for (int w = 0; w < 32; w++) {
        g1j_set_inf(&tj[w * 256]);
        for (int d = 1; d < 256; d++) g1j_add(&tj[w * 256 + d], &tj[w * 256 + d - 1], &base);
        for (int i = 0; i < 8; i++) g1j_dbl(&base, &base);
    }
    g1a *ta = (g1a *)malloc(32 * 256 * sizeof(g1a));
    g1j_batch_to_affine(ta, tj, 32 * 256);
    free(tj);
    g1j *res = (g1j *)malloc(n * sizeof(g1j));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint64_t raw[4];
        fp_to_raw(raw, &scalars[i], &FR);
        g1j acc;
        g1j_set_inf(&acc);
        for (int w = 0; w < 32; w++) {
            unsigned d = (unsigned)((raw[w / 8] >> (8 * (w % 8))) & 0xff);
            if (d) g1j_madd(&acc, &acc, &ta[w * 256 + d]);
        }
        res[i] = acc;
    }
    g1j_batch_to_affine(out, res, n);
    free(res);
    free(ta);
}

/* Synthetic SRS with a KNOWN trapdoor (stand-in for the ppot_0080_13.ptau the
 * reference downloads and does not ship): g[i] = tau^i G; g_lagrange[i] = L_i(tau) G with
 * L_i(tau) = (tau^n - 1)/n * omega^i / (tau - omega^i).  g2/s_g2 are not produced
 * (left zero): proofs over a synthetic SRS are checked with the trapdoor. */
void orc_srs_synthetic(uint32_t k, const fe *tau, orc_srs *out) {
    memset(out, 0, sizeof(*out));
    size_t n = (size_t)1 << k;
    out->k = k; out->n = n;
    out->g = (g1a *)malloc(n * sizeof(g1a));
    out->g_lagrange = (g1a *)malloc(n * sizeof(g1a));
    fe *s = (fe *)malloc(n * sizeof(fe));
    fr_one(&s[0]);
    for (size_t i = 1; i < n; i++) fr_mul(&s[i], &s[i - 1], tau);
    fixed_base_mul_many(s, n, out->g);
    fe tn, one, w, wi, nn, c;
    fr_mul(&tn, &s[n - 1], tau);
    fr_one(&one);
    fr_sub(&tn, &tn, &one);
    fp_from_u64(&nn, (uint64_t)n, &FR);
    fr_inv(&nn, &nn);
    fr_mul(&c, &tn, &nn);
    orc_fr_root_of_unity((int)k, &w);
    fe *den = (fe *)malloc(n * sizeof(fe));
    wi = one;
    for (size_t i = 0; i < n; i++) {
        fr_sub(&den[i], tau, &wi);
        fr_mul(&s[i], &c, &wi);
        fr_mul(&wi, &wi, &w);
    }
    orc_batch_invert(den, n);
    for (size_t i = 0; i < n; i++) fr_mul(&s[i], &s[i], &den[i]);
    fixed_base_mul_many(s, n, out->g_lagrange);
    free(den);
    free(s);
}
