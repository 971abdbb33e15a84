/*
 * oracle/oracle.h — TEST INFRASTRUCTURE: the CPU oracle's exported surface.
 *
 * A plain-C restatement of the algorithms on the Shielder proving path
 * (SRS ingest, MSM, NTT, Poseidon2, Keccak transcript, the PLONK prover and verifier).  It exists to CHECK the HIP path (tests/, smoke()) and
 * to be TIMED as the CPU baseline (bench.py cpu_baseline).  Nothing in the
 * product (zkos-monorepo_amd/, include/) may include, link or call it.
 *
 * Parity status: "partially pinned" — every primitive is pinned against the
 * fixtures the reference holds (ptau/RawBytes SRS incl. g_lagrange, Poseidon2
 * parameter set and generated on-chain code, Keccak KATs, delta/omega literals);
 * whole-proof bytes are UNPINNED (the reference has no golden proof and its
 * prover crates are not vendored).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include "field.h"
#include "curve.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashes.c */
void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
void orc_chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]);
void orc_fr_draw(const uint8_t seed[32], uint64_t counter, fe *out);
void orc_poseidon2_permute(fe state[8]);
void orc_poseidon2_hash(const fe *in, int arity, fe *out);

/* ---- poly.c : evaluation domain, NTT */
void orc_fr_root_of_unity(int k, fe *omega);            /* primitive 2^k-th root */
void orc_fr_delta(fe *delta);
void orc_fr_zeta(fe *zeta);
void orc_ntt(fe *a, int k, const fe *omega);            /* in place, natural order in/out, no scaling */
void orc_intt(fe *a, int k);                            /* inverse with 1/n scaling (lagrange_to_coeff) */
void orc_fntt(fe *a, int k);                            /* forward with the domain's omega (coeff_to_lagrange) */
void orc_coeff_to_extended(const fe *coeff, int k, int ext_k, fe *out);   /* out has 2^ext_k */
void orc_extended_to_coeff(fe *ext, int ext_k);                           /* in place */
void orc_eval_poly(const fe *coeff, size_t n, const fe *x, fe *out);      /* Horner */
void orc_batch_invert(fe *a, size_t n);                                   /* zeros stay zero */
int  orc_num_threads(void);

/* ---- msm.c */
void orc_msm(const g1a *bases, const fe *scalars, size_t n, g1j *out);    /* multi-threaded, bucket method */
void orc_msm_serial(const g1a *bases, const fe *scalars, size_t n, g1j *out);
void orc_g1_fft(g1j *a, int k, const fe *omega);        /* in place group FFT (natural in/out) */
void orc_g1_lagrange_from_monomial(const g1a *g, int k, g1a *g_lagrange);

/* ---- srs.c */
typedef struct {
    uint32_t k;
    size_t n;
    g1a *g;             /* n, monomial basis */
    g1a *g_lagrange;    /* n */
    uint8_t g2[128];    /* raw bytes as stored: x.c0 x.c1 y.c0 y.c1, 32-byte LE each (canonical) */
    uint8_t s_g2[128];
} orc_srs;
int  orc_srs_from_ptau(const uint8_t *bytes, size_t len, orc_srs *out);          /* g_lagrange computed */
int  orc_srs_from_halo2_raw(const uint8_t *bytes, size_t len, orc_srs *out);
void orc_srs_synthetic(uint32_t k, const fe *tau, orc_srs *out);                 /* g[i] = tau^i * G */
void orc_srs_free(orc_srs *s);

#ifdef __cplusplus
}
#endif
#endif
