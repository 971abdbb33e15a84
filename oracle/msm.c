/*
 * oracle/msm.c — TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Multi-scalar multiplication restating halo2curves 0.6.1 `best_multiexp` /
 * `multiexp_serial` (un-vendored): window c = 1 | 3 | ceil(ln m),
 * segments = 256/c + 1 processed high to low with c doublings between, 2^c - 1
 * buckets, running-sum reduction; the point range is split over the threads and the
 * per-chunk results added.  No signed digits, no GLV.  The result is a group
 * element, so it is bit-identical to any other correct MSM after to_affine.
 *
 * Also the group FFT used by ParamsKZG::from_parts to derive g_lagrange from g
 * (called at /root/reference/crates/powers-of-tau/lib.rs:71).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline unsigned get_at(int segment, int c, const uint64_t raw[4]) {
    int skip = segment * c;
    if (skip >= 256) return 0;
    int limb = skip / 64, off = skip % 64;
    uint64_t v = raw[limb] >> off;
    if (off + c > 64 && limb < 3) v |= raw[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ULL << c) - 1));
}

static void multiexp_serial(const g1a *bases, const fe *scalars, size_t m, g1j *acc) {
    int c;
    if (m < 4) c = 1;
    else if (m < 32) c = 3;
    else c = (int)ceil(log((double)m));
    uint64_t(*raw)[4] = malloc(m * sizeof(*raw));
    for (size_t i = 0; i < m; i++) fp_to_raw(raw[i], &scalars[i], &FR);
    size_t nb = ((size_t)1 << c) - 1;
    g1j *buckets = (g1j *)malloc(nb * sizeof(g1j));
    int segments = 256 / c + 1;
    for (int seg = segments - 1; seg >= 0; seg--) {
        for (int i = 0; i < c; i++) g1j_dbl(acc, acc);
        memset(buckets, 0, nb * sizeof(g1j));
        for (size_t i = 0; i < m; i++) {
            unsigned d = get_at(seg, c, raw[i]);
            if (d) g1j_madd(&buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        g1j running;
        g1j_set_inf(&running);
        for (size_t b = nb; b-- > 0;) {
            g1j_add(&running, &running, &buckets[b]);
            g1j_add(acc, acc, &running);
        }
    }
    free(buckets);
    free(raw);
}

void orc_msm_serial(const g1a *bases, const fe *scalars, size_t n, g1j *out) {
    g1j_set_inf(out);
    multiexp_serial(bases, scalars, n, out);
}

void orc_msm(const g1a *bases, const fe *scalars, size_t n, g1j *out) {
    size_t T = (size_t)orc_num_threads();
    if (n <= T) { orc_msm_serial(bases, scalars, n, out); return; }
    size_t chunk = n / T;
    size_t nchunks = (n + chunk - 1) / chunk;
    g1j *res = (g1j *)calloc(nchunks, sizeof(g1j));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t ci = 0; ci < nchunks; ci++) {
        size_t lo = ci * chunk, hi = lo + chunk > n ? n : lo + chunk;
        multiexp_serial(bases + lo, scalars + lo, hi - lo, &res[ci]);
    }
    g1j_set_inf(out);
    for (size_t ci = 0; ci < nchunks; ci++) g1j_add(out, out, &res[ci]);
    free(res);
}

static inline size_t bitrev(size_t x, int k) {
    size_t r = 0;
    for (int i = 0; i < k; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

void orc_g1_fft(g1j *a, int k, const fe *omega) {
    size_t n = (size_t)1 << k;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, k);
        if (i < j) { g1j t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    fe *tw = (fe *)malloc((n / 2 ? n / 2 : 1) * sizeof(fe));
    fr_one(&tw[0]);
    for (size_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], omega);
    for (int s = 1; s <= k; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n / m;
#pragma omp parallel for schedule(static)
        for (size_t b = 0; b < n / 2; b++) {
            size_t grp = b / half, j = b % half;
            size_t lo = grp * m + j, hi = lo + half;
            g1j t, u = a[lo], nt;
            if (j == 0) t = a[hi];
            else g1j_mul(&t, &a[hi], &tw[j * step]);
            g1j_add(&a[lo], &u, &t);
            g1j_neg(&nt, &t);
            g1j_add(&a[hi], &u, &nt);
        }
    }
    free(tw);
}

/* g_lagrange = n^{-1} * FFT_{omega^{-1}}(g) */
void orc_g1_lagrange_from_monomial(const g1a *g, int k, g1a *g_lagrange) {
    size_t n = (size_t)1 << k;
    g1j *a = (g1j *)malloc(n * sizeof(g1j));
    for (size_t i = 0; i < n; i++) g1j_from_affine(&a[i], &g[i]);
    fe w, wi, nn, ninv;
    orc_fr_root_of_unity(k, &w);
    fr_inv(&wi, &w);
    orc_g1_fft(a, k, &wi);
    fp_from_u64(&nn, (uint64_t)n, &FR);
    fr_inv(&ninv, &nn);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) g1j_mul(&a[i], &a[i], &ninv);
    g1j_batch_to_affine(g_lagrange, a, n);
    free(a);
}
