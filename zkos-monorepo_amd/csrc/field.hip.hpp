// BN254 prime-field arithmetic for the HIP path (device) and the host driver.
//
// Representation: 8 x u32 little-endian limbs, Montgomery form with R = 2^256 -- byte
// for byte the 4 x u64 Montgomery limbs of the reference's SRS files
// (/root/reference/crates/powers-of-tau/lib.rs:92-109, 206-227), so SRS tables, proving
// keys and witnesses are used in HBM exactly as they arrive.
//
// The multiplier is a CIOS Montgomery product on 32-bit limbs: every partial product is
// one v_mad_u64_u32 (32x32+64 -> 64) on CDNA4; there is no MFMA formulation of a 254-bit
// modular product worth its operand shuffling (BASELINE.json north_star).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SHP_HD __host__ __device__ __forceinline__
#else
#define SHP_HD inline
#endif

namespace shp {

struct FqParams {
    static SHP_HD constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    static SHP_HD constexpr uint32_t r1(int i) {   // R mod p  (Montgomery one)
        constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    static SHP_HD constexpr uint32_t r2(int i) {   // R^2 mod p
        constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return v[i];
    }
    static SHP_HD constexpr uint32_t r3(int i) {   // R^3 mod p
        constexpr uint32_t v[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u, 0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};
        return v[i];
    }
    static constexpr uint32_t INV = 0xe4866389u;   // -p^{-1} mod 2^32
};

struct FrParams {
    static SHP_HD constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    static SHP_HD constexpr uint32_t r1(int i) {
        constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    static SHP_HD constexpr uint32_t r2(int i) {
        constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return v[i];
    }
    static SHP_HD constexpr uint32_t r3(int i) {
        constexpr uint32_t v[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};
        return v[i];
    }
    static constexpr uint32_t INV = 0xefffffffu;
};


#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 building blocks (measured, tools/ubench/instr_rate.hip): v_mad_u64_u32 issues at the
// same 4 cycles per wave64 as a 32-bit add, so a 254-bit Montgomery product is priced by its
// instruction COUNT.  hipcc's code for the portable CIOS loop is 128 mads + ~490 moves and 64-bit
// adds; the product-scanning form below is 128 mads + 128 carry adds + ~50 moves.
// (hi:lo) is a 96-bit column accumulator: lo += a * b, the carry out of bit 63 goes to hi.
__device__ __forceinline__ void mac96(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
// same with a wave-uniform (compile-time constant) multiplier held in an SGPR
__device__ __forceinline__ void mac96_k(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t k) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "s"(k) : "vcc");
}
#endif

template <class PR>
struct alignas(16) Fp {
    uint32_t l[8];

    static SHP_HD Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    static SHP_HD Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = PR::r1(i);
        return r;
    }
    static SHP_HD Fp modulus() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = PR::p(i);
        return r;
    }
    SHP_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    SHP_HD bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    SHP_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // raw limb vector >= modulus ?
    SHP_HD bool geq_p() const {
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            if (l[i] > PR::p(i)) return true;
            if (l[i] < PR::p(i)) return false;
        }
        return true;
    }
    // raw a < b (as 256-bit integers)
    static SHP_HD bool raw_lt(const Fp& a, const Fp& b) {
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            if (a.l[i] < b.l[i]) return true;
            if (a.l[i] > b.l[i]) return false;
        }
        return false;
    }

    // r = a - p if a >= p (branch-free select)
    SHP_HD void reduce_once() {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7, keep;
        asm("v_subrev_co_u32 %0, vcc, %17, %9\n\t"
            "v_subbrev_co_u32 %1, vcc, %18, %10, vcc\n\t"
            "v_subbrev_co_u32 %2, vcc, %19, %11, vcc\n\t"
            "v_subbrev_co_u32 %3, vcc, %20, %12, vcc\n\t"
            "v_subbrev_co_u32 %4, vcc, %21, %13, vcc\n\t"
            "v_subbrev_co_u32 %5, vcc, %22, %14, vcc\n\t"
            "v_subbrev_co_u32 %6, vcc, %23, %15, vcc\n\t"
            "v_subbrev_co_u32 %7, vcc, %24, %16, vcc\n\t"
            "v_cndmask_b32 %8, 0, -1, vcc"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(keep)
            : "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(l[4]), "v"(l[5]), "v"(l[6]), "v"(l[7]),
              // the modulus sits in VGPRs: an SGPR source next to the VCC carry-in would be a second
              // constant-bus read, which gfx9 encodings do not allow
              "v"(PR::p(0)), "v"(PR::p(1)), "v"(PR::p(2)), "v"(PR::p(3)), "v"(PR::p(4)), "v"(PR::p(5)), "v"(PR::p(6)), "v"(PR::p(7))
            : "vcc");
        // keep = all ones where the subtraction borrowed (value < p)
        l[0] = keep ? l[0] : t0; l[1] = keep ? l[1] : t1; l[2] = keep ? l[2] : t2; l[3] = keep ? l[3] : t3;
        l[4] = keep ? l[4] : t4; l[5] = keep ? l[5] : t5; l[6] = keep ? l[6] : t6; l[7] = keep ? l[7] : t7;
#else
        uint32_t t[8];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)l[i] - PR::p(i) - bw;
            t[i] = (uint32_t)d;
            bw = (d >> 32) & 1;
        }
        if (!bw) {
#pragma unroll
            for (int i = 0; i < 8; i++) l[i] = t[i];
        }
#endif
    }

    static SHP_HD Fp add(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r = a;
        asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
            "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %7, %15, vcc"
            : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7])
            : "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
            : "vcc");
        r.reduce_once();     // p < 2^254: no carry out of limb 7
        return r;
#else
        return add_host64(a, b);
#endif
    }
    static SHP_HD Fp sub(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r = a;
        uint32_t mask, tmp;
        asm("v_sub_co_u32 %0, vcc, %0, %9\n\t"
            "v_subb_co_u32 %1, vcc, %1, %10, vcc\n\t"
            "v_subb_co_u32 %2, vcc, %2, %11, vcc\n\t"
            "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\t"
            "v_subb_co_u32 %4, vcc, %4, %13, vcc\n\t"
            "v_subb_co_u32 %5, vcc, %5, %14, vcc\n\t"
            "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\t"
            "v_subb_co_u32 %7, vcc, %7, %16, vcc\n\t"
            "v_cndmask_b32 %8, 0, -1, vcc"
            : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
              "=&v"(mask)
            : "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
            : "vcc");
        // + p where the subtraction borrowed
        asm("v_and_b32 %8, %10, %9\n\t"
            "v_add_co_u32 %0, vcc, %0, %8\n\t"
            "v_and_b32 %8, %11, %9\n\t"
            "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\t"
            "v_and_b32 %8, %12, %9\n\t"
            "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t"
            "v_and_b32 %8, %13, %9\n\t"
            "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
            "v_and_b32 %8, %14, %9\n\t"
            "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\t"
            "v_and_b32 %8, %15, %9\n\t"
            "v_addc_co_u32 %5, vcc, %5, %8, vcc\n\t"
            "v_and_b32 %8, %16, %9\n\t"
            "v_addc_co_u32 %6, vcc, %6, %8, vcc\n\t"
            "v_and_b32 %8, %17, %9\n\t"
            "v_addc_co_u32 %7, vcc, %7, %8, vcc"
            : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
              "=&v"(tmp)
            : "v"(mask), "s"(PR::p(0)), "s"(PR::p(1)), "s"(PR::p(2)), "s"(PR::p(3)), "s"(PR::p(4)), "s"(PR::p(5)), "s"(PR::p(6)),
              "s"(PR::p(7))
            : "vcc");
        return r;
#else
        return sub_host64(a, b);
#endif
    }
    static SHP_HD Fp neg(const Fp& a) {
        if (a.is_zero()) return a;
        Fp r;
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)PR::p(i) - a.l[i] - bw;
            r.l[i] = (uint32_t)d;
            bw = (d >> 32) & 1;
        }
        return r;
    }
    static SHP_HD Fp dbl(const Fp& a) { return add(a, a); }

    // Montgomery product a*b/R mod p. Operands may be any 256-bit values with a*b < p*R
    // (so raw, unreduced inputs such as ChaCha output are allowed against a reduced one).
    static SHP_HD Fp mul(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
        return mul_host64(a, b);
#else
        // finely integrated product scanning (FIPS): column k sums a_i b_{k-i} and m_i p_{k-i} in one
        // 96-bit accumulator; m_k clears the column's low word; the upper columns are the result.
        // One asm statement per column (tools/gen_field_mul.py).
#include "field_mul_gfx950.inc"
        // the 2^256 word is 0 for p < 2^254 with reduced inputs; with one raw 256-bit input the value is
        // still < 2p, and 2p < 2^256, so a single conditional subtraction is enough.
        r.reduce_once();
        return r;
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // The same product without its last conditional subtraction: below 2p whenever a b < p R (one operand below p, the
    // other any 256-bit value below 4p).  For code that keeps values lazily in [0, 4p) -- 4p < 2^256 -- such as the
    // butterflies of ntt_dit_pass_kernel.
    static __device__ __forceinline__ Fp mul_lazy(const Fp& a, const Fp& b) {
#include "field_mul_gfx950.inc"
        return r;
    }
#elif defined(__HIPCC__)
    static __device__ Fp mul_lazy(const Fp& a, const Fp& b);      // device code only (declared for hipcc's host pass)
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host arithmetic on 4 x u64 limbs (the driver's witness program, transcript and point normalisation run on it: a
    // lone proof waits for three dependent Poseidon2 permutations of ~900 products each before its first kernel).  On a
    // little-endian host the eight 32-bit limbs ARE four 64-bit ones; modulus and -p^-1 mod 2^64 are compile-time constants.
    static constexpr uint64_t p64(int i) { return (uint64_t)PR::p(2 * i) | ((uint64_t)PR::p(2 * i + 1) << 32); }
    static constexpr uint64_t inv64() {
        // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: x' = x (2 + p x) keeps x = -p^-1 and doubles the bits
        const uint64_t x = (uint64_t)PR::INV;
        return x * (2 + p64(0) * x);
    }
    // t (+ 2^256 t4) -> t - p where t >= p; the value is below 2p
    static inline void reduce_once64(uint64_t t[4]) {
        typedef unsigned __int128 u128;
        uint64_t d[4];
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            const u128 x = (u128)t[i] - p64(i) - (uint64_t)bw;
            d[i] = (uint64_t)x;
            bw = (x >> 64) & 1;
        }
        if (!bw) { t[0] = d[0]; t[1] = d[1]; t[2] = d[2]; t[3] = d[3]; }
    }
    // CIOS, product and reduction interleaved limb by limb; a fifth word keeps the one extra bit a raw (unreduced
    // 256-bit) operand can produce between iterations
    static inline Fp mul_host64(const Fp& a, const Fp& b) {
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4];
        memcpy(A, a.l, 32);
        memcpy(B, b.l, 32);
        constexpr uint64_t P0 = p64(0), P1 = p64(1), P2 = p64(2), P3 = p64(3), INV = inv64();
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma GCC unroll 4
        for (int i = 0; i < 4; i++) {
            const uint64_t bi = B[i];
            u128 c = (u128)A[0] * bi + t0;
            const uint64_t m = (uint64_t)c * INV;
            u128 r = (u128)m * P0 + (uint64_t)c;                       // low word becomes zero
            c = (u128)A[1] * bi + t1 + (uint64_t)(c >> 64);
            r = (u128)m * P1 + (uint64_t)c + (uint64_t)(r >> 64);
            t0 = (uint64_t)r;
            c = (u128)A[2] * bi + t2 + (uint64_t)(c >> 64);
            r = (u128)m * P2 + (uint64_t)c + (uint64_t)(r >> 64);
            t1 = (uint64_t)r;
            c = (u128)A[3] * bi + t3 + (uint64_t)(c >> 64);
            r = (u128)m * P3 + (uint64_t)c + (uint64_t)(r >> 64);
            t2 = (uint64_t)r;
            const u128 top = (u128)t4 + (uint64_t)(c >> 64) + (uint64_t)(r >> 64);
            t3 = (uint64_t)top;
            t4 = (uint64_t)(top >> 64);
        }
        uint64_t t[4] = {t0, t1, t2, t3};          // a b < p R leaves t4 = 0 and t < 2p
        reduce_once64(t);
        Fp out;
        memcpy(out.l, t, 32);
        return out;
    }
    static inline Fp add_host64(const Fp& a, const Fp& b) {
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], t[4];
        memcpy(A, a.l, 32);
        memcpy(B, b.l, 32);
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)A[i] + B[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
        reduce_once64(t);                          // p < 2^254: no carry out of the top limb
        Fp out;
        memcpy(out.l, t, 32);
        return out;
    }
    static inline Fp sub_host64(const Fp& a, const Fp& b) {
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], t[4];
        memcpy(A, a.l, 32);
        memcpy(B, b.l, 32);
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            const u128 x = (u128)A[i] - B[i] - (uint64_t)bw;
            t[i] = (uint64_t)x;
            bw = (x >> 64) & 1;
        }
        const uint64_t mask = (uint64_t)0 - (uint64_t)bw;       // + p where the subtraction borrowed
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + (p64(i) & mask);
            t[i] = (uint64_t)c;
            c >>= 64;
        }
        Fp out;
        memcpy(out.l, t, 32);
        return out;
    }
#endif
    static SHP_HD Fp sqr(const Fp& a) { return mul(a, a); }

    SHP_HD Fp operator+(const Fp& b) const { return add(*this, b); }
    SHP_HD Fp operator-(const Fp& b) const { return sub(*this, b); }
    SHP_HD Fp operator*(const Fp& b) const { return mul(*this, b); }

    // canonical integer (raw limbs) -> Montgomery
    static SHP_HD Fp from_raw(const Fp& raw) {
        Fp r2;
#pragma unroll
        for (int i = 0; i < 8; i++) r2.l[i] = PR::r2(i);
        return mul(raw, r2);
    }
    // Montgomery -> canonical integer
    SHP_HD Fp to_raw() const {
        Fp o = zero();
        o.l[0] = 1;
        return mul(*this, o);
    }
    static SHP_HD Fp from_u64(uint64_t v) {
        Fp r = zero();
        r.l[0] = (uint32_t)v;
        r.l[1] = (uint32_t)(v >> 32);
        return from_raw(r);
    }
    // 64 uniform bytes (two raw 256-bit halves) -> d0*R^2 + d1*R^3 (halo2curves from_u512)
    static SHP_HD Fp from_u512(const Fp& d0, const Fp& d1) {
        Fp r2, r3;
#pragma unroll
        for (int i = 0; i < 8; i++) { r2.l[i] = PR::r2(i); r3.l[i] = PR::r3(i); }
        return add(mul(d0, r2), mul(d1, r3));
    }

    // a^e, e a raw 256-bit exponent (MSB-first square and multiply)
    static SHP_HD Fp pow(const Fp& a, const Fp& e) {
        Fp acc = one();
        bool started = false;
        for (int i = 255; i >= 0; i--) {
            if (started) acc = sqr(acc);
            if ((e.l[i >> 5] >> (i & 31)) & 1) {
                acc = started ? mul(acc, a) : a;
                started = true;
            }
        }
        return acc;
    }
    static SHP_HD Fp pow_u64(const Fp& a, uint64_t e) {
        Fp ee = zero();
        ee.l[0] = (uint32_t)e;
        ee.l[1] = (uint32_t)(e >> 32);
        return pow(a, ee);
    }
    // a^(p-2); 0 -> 0   (Fermat: ~380 dependent products -- kept as the cross-check of inv())
    static SHP_HD Fp inv_fermat(const Fp& a) {
        if (a.is_zero()) return a;
        Fp e = modulus();
        e.l[0] -= 2;      // p is odd and p[0] >= 2
        return pow(a, e);
    }

    // raw 256-bit helpers of the binary inversion
    SHP_HD bool raw_is_one() const {
        uint32_t o = l[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 8; i++) o |= l[i];
        return o == 0;
    }
    SHP_HD void raw_shr1() {
#pragma unroll
        for (int i = 0; i < 7; i++) l[i] = (l[i] >> 1) | (l[i + 1] << 31);
        l[7] >>= 1;
    }
    SHP_HD void raw_sub(const Fp& b) {              // *this -= b (caller guarantees *this >= b)
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)l[i] - b.l[i] - bw;
            l[i] = (uint32_t)d;
            bw = (d >> 32) & 1;
        }
    }
    SHP_HD void raw_half_mod() {                    // *this = *this / 2 mod p (value < p)
        uint32_t mask = (uint32_t)0 - (l[0] & 1u);
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {               // p < 2^254 and value < p: the sum fits 256 bits
            c += (uint64_t)l[i] + (PR::p(i) & mask);
            l[i] = (uint32_t)c;
            c >>= 32;
        }
        raw_shr1();
    }

    // 1/a for a in Montgomery form; 0 -> 0.  Binary extended Euclid on the raw limbs (shift /
    // subtract only: ~1.5 * 254 short iterations instead of ~380 dependent Montgomery products),
    // then one product by R^3 to return to Montgomery form: (aR)^-1 * R^3 / R = a^-1 R.
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host inversion: Kaliski's almost-inverse on four 64-bit limbs, every run of trailing zero bits shifted out at
    // once (the same recurrence as inv_kaliski below), then the 2^k correction by Montgomery products with one-bit
    // constants.  ~3 us instead of ~9 for the bit-at-a-time extended Euclid it replaces: the prover inverts on the
    // host after every commitment phase (normalisation of the MSM sums) with the GPU idle.
    static Fp inv_host(const Fp& a) {
        typedef unsigned __int128 u128;
        uint64_t u[4], v[4], r[4] = {0, 0, 0, 0}, s[4] = {1, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u[i] = (uint64_t)PR::p(2 * i) | ((uint64_t)PR::p(2 * i + 1) << 32);
            v[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        }
        auto shr = [](uint64_t* x, unsigned t) {           // 1 <= t <= 63
            x[0] = (x[0] >> t) | (x[1] << (64 - t));
            x[1] = (x[1] >> t) | (x[2] << (64 - t));
            x[2] = (x[2] >> t) | (x[3] << (64 - t));
            x[3] >>= t;
        };
        auto shl = [](uint64_t* x, unsigned t) {
            x[3] = (x[3] << t) | (x[2] >> (64 - t));
            x[2] = (x[2] << t) | (x[1] >> (64 - t));
            x[1] = (x[1] << t) | (x[0] >> (64 - t));
            x[0] <<= t;
        };
        auto tz = [](uint64_t low) -> unsigned { return low ? (unsigned)__builtin_ctzll(low) : 63u; };
        auto sub = [](uint64_t* x, const uint64_t* y) {    // x -= y, returns the borrow
            u128 b = 0;
            for (int i = 0; i < 4; i++) {
                u128 d = (u128)x[i] - y[i] - b;
                x[i] = (uint64_t)d;
                b = (d >> 64) & 1;
            }
            return (uint64_t)b;
        };
        auto add = [](uint64_t* x, const uint64_t* y) {
            u128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (u128)x[i] + y[i];
                x[i] = (uint64_t)c;
                c >>= 64;
            }
        };
        auto geq = [](const uint64_t* x, const uint64_t* y) {
            for (int i = 3; i >= 0; i--)
                if (x[i] != y[i]) return x[i] > y[i];
            return true;
        };
        auto is_zero4 = [](const uint64_t* x) { return (x[0] | x[1] | x[2] | x[3]) == 0; };
        uint32_t k = 0;
        while (!is_zero4(v) && k < 600) {
            if (!(u[0] & 1)) {
                const unsigned t = tz(u[0]);
                shr(u, t); shl(s, t); k += t;
            } else if (!(v[0] & 1)) {
                const unsigned t = tz(v[0]);
                shr(v, t); shl(r, t); k += t;
            } else if (geq(v, u)) {
                sub(v, u);
                add(s, r);
                if (is_zero4(v)) {                         // u == v == 1: the last step of the plain algorithm
                    shl(r, 1);
                    k += 1;
                    break;
                }
                const unsigned t = tz(v[0]);
                shr(v, t); shl(r, t); k += t;
            } else {
                sub(u, v);
                add(r, s);
                const unsigned t = tz(u[0]);
                shr(u, t); shl(s, t); k += t;
            }
        }
        uint64_t pm[4];
        for (int i = 0; i < 4; i++) pm[i] = (uint64_t)PR::p(2 * i) | ((uint64_t)PR::p(2 * i + 1) << 32);
        if (geq(r, pm)) sub(r, pm);
        sub(pm, r);                                        // (aR)^-1 2^k mod p
        Fp x;
        for (int i = 0; i < 4; i++) { x.l[2 * i] = (uint32_t)pm[i]; x.l[2 * i + 1] = (uint32_t)(pm[i] >> 32); }
        Fp r3;
        for (int i = 0; i < 8; i++) r3.l[i] = PR::r3(i);
        x = mul(x, r3);
        const uint32_t k1 = k < 256 ? k : 256, k2 = k - k1;
        auto pow2 = [](uint32_t e) {
            Fp c = zero();
            c.l[e >> 5] = 1u << (e & 31);
            return c;
        };
        x = mul(x, pow2(256 - k1));
        if (k2) x = mul(x, pow2(256 - k2));
        return x;
    }
#endif

    static SHP_HD Fp inv(const Fp& a) {
        if (a.is_zero()) return a;
#if defined(__HIP_DEVICE_COMPILE__)
        return inv_kaliski(a);
#else
        return inv_host(a);
#endif
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // raw 256-bit carry chains (no reduction)
    static __device__ __forceinline__ void raw_add_to(Fp& r, const Fp& b) {
        asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
            "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %7, %15, vcc"
            : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7])
            : "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
            : "vcc");
    }
    // d = a - b (mod 2^256); returns all ones when it borrowed (a < b)
    static __device__ __forceinline__ uint32_t raw_sub_borrow(Fp& d, const Fp& a, const Fp& b) {
        d = a;
        uint32_t mask;
        asm("v_sub_co_u32 %0, vcc, %0, %9\n\t"
            "v_subb_co_u32 %1, vcc, %1, %10, vcc\n\t"
            "v_subb_co_u32 %2, vcc, %2, %11, vcc\n\t"
            "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\t"
            "v_subb_co_u32 %4, vcc, %4, %13, vcc\n\t"
            "v_subb_co_u32 %5, vcc, %5, %14, vcc\n\t"
            "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\t"
            "v_subb_co_u32 %7, vcc, %7, %16, vcc\n\t"
            "v_cndmask_b32 %8, 0, -1, vcc"
            : "+v"(d.l[0]), "+v"(d.l[1]), "+v"(d.l[2]), "+v"(d.l[3]), "+v"(d.l[4]), "+v"(d.l[5]), "+v"(d.l[6]), "+v"(d.l[7]),
              "=&v"(mask)
            : "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
            : "vcc");
        return mask;
    }
    // multi-bit shifts of the raw 256-bit value, 1 <= t <= 31 (one funnel shift per limb)
    __device__ __forceinline__ void raw_shr(uint32_t t) {
#pragma unroll
        for (int i = 0; i < 7; i++) l[i] = (l[i] >> t) | (l[i + 1] << (32 - t));
        l[7] >>= t;
    }
    __device__ __forceinline__ void raw_shl(uint32_t t) {
#pragma unroll
        for (int i = 7; i > 0; i--) l[i] = (l[i] << t) | (l[i - 1] >> (32 - t));
        l[0] <<= t;
    }
    static __device__ __forceinline__ uint32_t tz31(uint32_t low) {      // trailing zeros, capped at 31
        return low ? (uint32_t)__builtin_ctz(low) : 31u;
    }
    // Kaliski's almost-inverse (shift / add / subtract only, no modular halving) with every run of
    // trailing zero bits shifted out at once (~180 iterations instead of ~370), followed by the 2^k
    // correction as Montgomery products with one-bit constants:
    // x = (aR)^-1 2^k  ->  x 2^512 (by R^3)  ->  / 2^k1  ->  / 2^k2   (k1 + k2 = k)  =  a^-1 R.
    static __device__ Fp inv_kaliski(const Fp& a) {
        Fp u = modulus(), v = a, r = zero(), s = zero(), d;
        s.l[0] = 1;
        uint32_t k = 0;
        while (!v.is_zero() && k < 600) {
            if (!(u.l[0] & 1u)) {
                const uint32_t t = tz31(u.l[0]);
                u.raw_shr(t);
                s.raw_shl(t);
                k += t;
            } else if (!(v.l[0] & 1u)) {
                const uint32_t t = tz31(v.l[0]);
                v.raw_shr(t);
                r.raw_shl(t);
                k += t;
            } else if (raw_sub_borrow(d, v, u) == 0) {     // v >= u
                v = d;
                raw_add_to(s, r);
                if (v.is_zero()) {                         // u == v == 1: the last step of the plain algorithm
                    r.raw_shl(1);
                    k += 1;
                    break;
                }
                const uint32_t t = tz31(v.l[0]);
                v.raw_shr(t);
                r.raw_shl(t);
                k += t;
            } else {
                raw_sub_borrow(u, u, v);
                raw_add_to(r, s);
                const uint32_t t = tz31(u.l[0]);
                u.raw_shr(t);
                s.raw_shl(t);
                k += t;
            }
        }
        if (r.geq_p()) r.raw_sub(modulus());
        Fp x = modulus();
        x.raw_sub(r);                                      // (aR)^-1 2^k mod p
        Fp r3;
#pragma unroll
        for (int i = 0; i < 8; i++) r3.l[i] = PR::r3(i);
        x = mul(x, r3);
        const uint32_t k1 = k < 256 ? k : 256, k2 = k - k1;
        // one-bit constants 2^e built limb by limb (a dynamic limb index would put the value in scratch memory)
        auto pow2 = [](uint32_t e) {
            Fp c;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) c.l[i] = (e >> 5) == i ? 1u << (e & 31) : 0u;
            return c;
        };
        Fp c = pow2(256 - k1);
        if (k1 == 0) c = one();                            // unreachable for a != 0 (k >= 254)
        x = mul(x, c);
        if (k2) {
            c = pow2(256 - k2);
            x = mul(x, c);
        }
        return x;
    }
#endif
};

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

}  // namespace shp
