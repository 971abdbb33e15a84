// Lazy Fr values for the NTT butterflies (ntt_dit_pass_kernel): anything in [0, 4p) stands for its residue -- 4p < 2^256
// for the 254-bit modulus -- so that a butterfly (x, y) -> (x + t y, x - t y) spends ONE conditional subtraction where the
// reduced arithmetic spends three: x comes down to [0, 2p); Fp::mul_lazy leaves t y below 2p; the sum is a plain 256-bit
// addition and the difference adds 2p.  Also 32 x mod p by shifts.  The device versions are carry chains in inline
// assembly; the host versions compute the same values portably (tests/native/ntt_lazy_check.cpp checks the rules and
// bounds on the CPU).
#pragma once
#include "field.hip.hpp"

namespace shp {

SHP_HD constexpr uint32_t fr_2p(int i) {
    return (FrParams::p(i) << 1) | (i ? FrParams::p(i - 1) >> 31 : 0);
}

#if defined(__HIP_DEVICE_COMPILE__)
// v - 2p where that is not negative: [0, 4p) -> [0, 2p)
__device__ __forceinline__ void fr_reduce_2p(Fr& v) {
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
        : "v"(v.l[0]), "v"(v.l[1]), "v"(v.l[2]), "v"(v.l[3]), "v"(v.l[4]), "v"(v.l[5]), "v"(v.l[6]), "v"(v.l[7]),
          "v"(fr_2p(0)), "v"(fr_2p(1)), "v"(fr_2p(2)), "v"(fr_2p(3)), "v"(fr_2p(4)), "v"(fr_2p(5)), "v"(fr_2p(6)), "v"(fr_2p(7))
        : "vcc");
    v.l[0] = keep ? v.l[0] : t0; v.l[1] = keep ? v.l[1] : t1; v.l[2] = keep ? v.l[2] : t2; v.l[3] = keep ? v.l[3] : t3;
    v.l[4] = keep ? v.l[4] : t4; v.l[5] = keep ? v.l[5] : t5; v.l[6] = keep ? v.l[6] : t6; v.l[7] = keep ? v.l[7] : t7;
}
// a + b as 256-bit integers (both below 2p: the sum is below 4p)
__device__ __forceinline__ Fr fr_add_raw(const Fr& a, const Fr& b) {
    Fr r;
    asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %15, %23, vcc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
          "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
        : "vcc");
    return r;
}
// a - b + 2p (both below 2p: the result is positive and below 4p; the wrap of the first chain is undone by the second)
__device__ __forceinline__ Fr fr_sub_2p(const Fr& a, const Fr& b) {
    Fr r;
    asm("v_sub_co_u32 %0, vcc, %8, %16\n\t"
        "v_subb_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %15, %23, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %24\n\t"
        "v_addc_co_u32 %1, vcc, %1, %25, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %26, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %27, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %28, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %29, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %30, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %31, vcc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
          "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]),
          "v"(fr_2p(0)), "v"(fr_2p(1)), "v"(fr_2p(2)), "v"(fr_2p(3)), "v"(fr_2p(4)), "v"(fr_2p(5)), "v"(fr_2p(6)), "v"(fr_2p(7))
        : "vcc");
    return r;
}
#else
// v - 2p where that is not negative: [0, 4p) -> [0, 2p)
SHP_HD void fr_reduce_2p(Fr& v) {
    uint32_t t[8];
    uint64_t bw = 0;
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)v.l[i] - fr_2p(i) - bw;
        t[i] = (uint32_t)d;
        bw = (d >> 32) & 1;
    }
    if (!bw)
        for (int i = 0; i < 8; i++) v.l[i] = t[i];
}
// a + b as 256-bit integers (both below 2p: the sum is below 4p)
SHP_HD Fr fr_add_raw(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t c = 0;
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
// a - b + 2p (both below 2p: the result is positive and below 4p)
SHP_HD Fr fr_sub_2p(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t bw = 0, c = 0;
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a.l[i] - b.l[i] - bw;
        r.l[i] = (uint32_t)d;
        bw = (d >> 32) & 1;
    }
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)r.l[i] + fr_2p(i);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
#endif

// 32 x mod p without a product: x < p < 2^254 shifted up five bits, minus q p for q = (its top 32 bits) / (the modulus's top
// 27 bits + 1) -- never above floor(32 x / p) and at most one below it -- leaves less than 2 p
SHP_HD Fr fr_times32(const Fr& x) {
    uint32_t t[9];
    t[0] = x.l[0] << 5;
#pragma unroll
    for (int i = 1; i < 8; i++) t[i] = (x.l[i] << 5) | (x.l[i - 1] >> 27);
    t[8] = x.l[7] >> 27;
    constexpr uint32_t P27 = (FrParams::p(7) >> 3) + 1;
    const uint32_t q = ((t[8] << 29) | (t[7] >> 3)) / P27;
    Fr r;
    uint64_t carry = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t m = (uint64_t)q * FrParams::p(i) + carry;
        carry = m >> 32;
        const uint64_t d = (uint64_t)t[i] - (uint32_t)m - borrow;
        r.l[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    r.reduce_once();
    return r;
}

}  // namespace shp
