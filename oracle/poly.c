/*
 * oracle/poly.c — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Evaluation-domain arithmetic restating halo2_proofs v0.3.0's EvaluationDomain /
 * best_fft (un-vendored; Cargo.lock:2332-2334;
 * constants pinned by /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:475
 * (delta) and by resources/ppot_0080_11_raw's g_lagrange (omega, direction)):
 *   ROOT_OF_UNITY = 7^((r-1)/2^28), omega_k = ROOT^(2^(28-k)), delta = 7^(2^28),
 *   zeta = 7^((r-1)/3) (coset generator of the extended domain).
 * NTT = bit-reversal + radix-2 decimation-in-time stages, natural order in/out.
 */
#include "oracle.h"
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static fe ROOT28, DELTA, ZETA, ZETA2;
static int poly_ready = 0;
static void poly_init(void) {
    if (poly_ready) return;
    fe seven;
    fp_from_u64(&seven, 7, &FR);
    /* (r-1) >> 28 */
    uint64_t e[4], one[4] = {1, 0, 0, 0}, rm1[4];
    raw_sub(rm1, FR.p, one);
    for (int i = 0; i < 4; i++) e[i] = (rm1[i] >> 28) | (i < 3 ? rm1[i + 1] << 36 : 0);
    fp_pow(&ROOT28, &seven, e, &FR);
    fp_pow_u64(&DELTA, &seven, 1ULL << 28, &FR);
    /* (r-1)/3 by schoolbook long division on limbs */
    uint64_t q[4];
    u128 rem = 0;
    for (int i = 3; i >= 0; i--) {
        u128 cur = (rem << 64) | rm1[i];
        q[i] = (uint64_t)(cur / 3);
        rem = cur % 3;
    }
    fp_pow(&ZETA, &seven, q, &FR);
    fr_sqr(&ZETA2, &ZETA);
    poly_ready = 1;
}

void orc_fr_root_of_unity(int k, fe *omega) {
    poly_init();
    fe w = ROOT28;
    for (int i = k; i < 28; i++) fr_sqr(&w, &w);
    *omega = w;
}
void orc_fr_delta(fe *d) { poly_init(); *d = DELTA; }
void orc_fr_zeta(fe *z) { poly_init(); *z = ZETA; }

static inline size_t bitrev(size_t x, int k) {
    size_t r = 0;
    for (int i = 0; i < k; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

void orc_ntt(fe *a, int k, const fe *omega) {
    size_t n = (size_t)1 << k;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, k);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    /* twiddles w^0..w^{n/2-1} */
    fe *tw = (fe *)malloc((n / 2 ? n / 2 : 1) * sizeof(fe));
    fr_one(&tw[0]);
    for (size_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], omega);
    for (int s = 1; s <= k; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n / m;
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t b = 0; b < n / 2; b++) {
            size_t grp = b / half, j = b % half;
            size_t lo = grp * m + j, hi = lo + half;
            fe t, u = a[lo];
            fr_mul(&t, &a[hi], &tw[j * step]);
            fr_add(&a[lo], &u, &t);
            fr_sub(&a[hi], &u, &t);
        }
    }
    free(tw);
}

void orc_fntt(fe *a, int k) {
    fe w;
    orc_fr_root_of_unity(k, &w);
    orc_ntt(a, k, &w);
}

void orc_intt(fe *a, int k) {
    fe w, wi, ninv, nn;
    orc_fr_root_of_unity(k, &w);
    fr_inv(&wi, &w);
    orc_ntt(a, k, &wi);
    fp_from_u64(&nn, 1ULL << k, &FR);
    fr_inv(&ninv, &nn);
    size_t n = (size_t)1 << k;
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
}

/* halo2 coeff_to_extended: coefficient i *= [1, zeta, zeta^2][i mod 3], zero-pad, forward NTT */
void orc_coeff_to_extended(const fe *coeff, int k, int ext_k, fe *out) {
    poly_init();
    size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
    for (size_t i = 0; i < n; i++) {
        switch (i % 3) {
            case 0: out[i] = coeff[i]; break;
            case 1: fr_mul(&out[i], &coeff[i], &ZETA); break;
            default: fr_mul(&out[i], &coeff[i], &ZETA2); break;
        }
    }
    memset(out + n, 0, (ne - n) * sizeof(fe));
    orc_fntt(out, ext_k);
}

/* halo2 extended_to_coeff: inverse NTT, then coefficient i *= [1, zeta^2, zeta][i mod 3] */
void orc_extended_to_coeff(fe *ext, int ext_k) {
    poly_init();
    size_t ne = (size_t)1 << ext_k;
    orc_intt(ext, ext_k);
    for (size_t i = 0; i < ne; i++) {
        if (i % 3 == 1) fr_mul(&ext[i], &ext[i], &ZETA2);
        else if (i % 3 == 2) fr_mul(&ext[i], &ext[i], &ZETA);
    }
}

void orc_eval_poly(const fe *coeff, size_t n, const fe *x, fe *out) {
    fe acc;
    fp_zero(&acc);
    for (size_t i = n; i-- > 0;) {
        fr_mul(&acc, &acc, x);
        fr_add(&acc, &acc, &coeff[i]);
    }
    *out = acc;
}

void orc_batch_invert(fe *a, size_t n) {
    if (!n) return;
    fe *pre = (fe *)malloc(n * sizeof(fe));
    fe acc;
    fr_one(&acc);
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!fp_is_zero(&a[i])) fr_mul(&acc, &acc, &a[i]);
    }
    fe inv;
    fr_inv(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fp_is_zero(&a[i])) continue;
        fe t;
        fr_mul(&t, &inv, &pre[i]);
        fr_mul(&inv, &inv, &a[i]);
        a[i] = t;
    }
    free(pre);
}
