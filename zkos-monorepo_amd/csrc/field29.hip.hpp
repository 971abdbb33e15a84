// BN254 Fq in 9 UNSATURATED 29-bit limbs, for the point additions of the MSM kernels (device only).
//
// Why a second representation: on gfx950 every integer VALU instruction costs the same four cycles, so a Montgomery
// product is priced by its instruction count.  With saturated 32-bit limbs (field.hip.hpp) each of the 128 partial products is
// a v_mad_u64_u32 PLUS a v_addc_co_u32 for the carry out of the 64-bit accumulator.  With 29-bit limbs a column of 9 + 9
// partial products (< 2^58 each) cannot overflow 64 bits: a partial product is ONE v_mad_u64_u32, 162 in all, and the
// Montgomery radix 2^261 exceeds 4p by so much that no product needs a conditional subtraction.  Measured with every CU busy
// (tools/ubench/mul_variants.hip): 174 G products/s against 133 G for the saturated form (and 115 G for a 5 x 52-bit FP64
// formulation).
//
// Values are held LAZILY: an Fq29 spells an integer v = sum l[i] 2^(29 i) congruent to x * 2^261 (mod p), with limbs 0..7
// below 2^29 ("normalised") and v below a small multiple of p that every call site tracks:
//   mul29(a, b)  needs a < A p, b < B p with A B <= 254; returns < 2.5 p  (v = (a b + q p) / 2^261 < a b / 2^261 + p)
//   sub29<K>(a, b) = a - b + K p needs b < (K - 1/2) p; returns < a + K p
// Tables and results stay in the 8 x 32-bit Montgomery-2^256 form of field.hip.hpp; conversions happen once per gathered point
// and once per accumulator.
#pragma once
#include "curve.hip.hpp"


namespace shp {

constexpr uint32_t M29 = (1u << 29) - 1;
struct Fq29 { uint32_t l[9]; };

namespace f29 {
struct L8 { uint32_t l[8]; };
struct L9 { uint32_t l[9]; };

constexpr L8 modulus() {
    L8 r{};
    for (int i = 0; i < 8; i++) r.l[i] = FqParams::p(i);
    return r;
}
constexpr bool geq(const L8& a, const L8& b) {
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    }
    return true;
}
constexpr L8 dbl_mod_p(const L8& a) {                      // 2a mod p for a < p < 2^254
    L8 r{};
    uint32_t carry = 0;
    for (int i = 0; i < 8; i++) { r.l[i] = (a.l[i] << 1) | carry; carry = a.l[i] >> 31; }
    const L8 p = modulus();
    if (geq(r, p)) {
        uint64_t borrow = 0;
        for (int i = 0; i < 8; i++) {
            const uint64_t d = (uint64_t)r.l[i] - p.l[i] - borrow;
            r.l[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
    }
    return r;
}
constexpr L8 pow2_mod_p(int e) {                           // 2^e mod p, e >= 256 (from R mod p = 2^256 mod p)
    L8 r{};
    for (int i = 0; i < 8; i++) r.l[i] = FqParams::r1(i);
    for (int i = 256; i < e; i++) r = dbl_mod_p(r);
    return r;
}
constexpr L9 split29(const L8& v) {
    L9 r{};
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint64_t x = v.l[w];
        if (w + 1 < 8) x |= (uint64_t)v.l[w + 1] << 32;
        r.l[i] = (uint32_t)(x >> s) & M29;
    }
    return r;
}
// K p with limbs 0..7 in [S 2^29 - S, (S + 1) 2^29): added limb by limb it keeps a - b + K p non-negative in EVERY limb
// as long as the limbs of b stay below S 2^29 - S + 1, so the subtraction needs no signed arithmetic
constexpr L9 kp_spread(uint32_t K, int64_t S = 1) {
    const L9 p = split29(modulus());
    int64_t t[9] = {};
    uint64_t carry = 0;
    for (int i = 0; i < 9; i++) {
        const uint64_t acc = (uint64_t)K * p.l[i] + carry;
        t[i] = i < 8 ? (int64_t)(acc & M29) : (int64_t)acc;
        carry = acc >> 29;
    }
    for (int i = 0; i < 8; i++) { t[i] += S << 29; t[i + 1] -= S; }
    L9 r{};
    for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)t[i];
    return r;
}

constexpr L9 P29 = split29(modulus());
constexpr uint32_t PINV29 = FqParams::INV & M29;           // -p^-1 mod 2^29
constexpr uint32_t PPOSINV29 = (0u - FqParams::INV) & M29; // p^-1 mod 2^29
constexpr L9 ONE = split29(pow2_mod_p(261));               // Montgomery one for R = 2^261
constexpr L9 C266 = split29(pow2_mod_p(266));              // Montgomery-2^256 limbs (no shift) -> Montgomery-2^261, reduced
constexpr L9 C256 = split29(pow2_mod_p(256));              // Montgomery-2^261 -> Montgomery-2^256
constexpr L9 KP3 = kp_spread(3), KP6 = kp_spread(6), KP8W = kp_spread(8, 4), KP11 = kp_spread(11);
}  // namespace f29


// bits [29 i, 29 i + 29) of the 256-bit value shifted left by SHIFT bits (SHIFT = 5: Montgomery-2^256 -> 2^261, value < 32 p)
template <int SHIFT>
SHP_HD Fq29 split29_dev(const Fq& v) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i - SHIFT;                    // first source bit of the limb
        uint32_t x = 0;
        if (bit + 29 > 0) {
            const int b0 = bit < 0 ? 0 : bit;
            const int w = b0 >> 5, s = b0 & 31;
            uint64_t src = v.l[w];
            if (w + 1 < 8) src |= (uint64_t)v.l[w + 1] << 32;
            x = (uint32_t)(src >> s);
            if (bit < 0) x <<= -bit;
        }
        r.l[i] = x & M29;
    }
    return r;
}

// a b 2^-261 mod p (lazy); see the bounds in the header comment
SHP_HD Fq29 mul29(const Fq29& a, const Fq29& b) {
    uint64_t acc = 0;
    uint32_t q[9];
    Fq29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * f29::P29.l[k - i];
        q[k] = ((uint32_t)acc * f29::PINV29) & M29;
        acc += (uint64_t)q[k] * f29::P29.l[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * f29::P29.l[k - i];
        r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// a^2 2^-261: the 36 cross products once, against a doubled operand (2 a_i < 2^30: a column stays below 2^64)
SHP_HD Fq29 sqr29(const Fq29& a) {
    uint64_t acc = 0;
    uint32_t q[9], d[9];
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)d[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * f29::P29.l[k - i];
        q[k] = ((uint32_t)acc * f29::PINV29) & M29;
        acc += (uint64_t)q[k] * f29::P29.l[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)d[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * f29::P29.l[k - i];
        r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// limbs 0..7 back below 2^29 (every limb non-negative on entry)
SHP_HD void normalise29(Fq29& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.l[i + 1] += a.l[i] >> 29;
        a.l[i] &= M29;
    }
}

// a - b + K p, normalised; b normalised and below (K - 1/2) p
SHP_HD Fq29 sub29(const Fq29& a, const Fq29& b, const f29::L9& kp) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + kp.l[i] - b.l[i];
    normalise29(r);
    return r;
}

// Montgomery-2^261 lazy value -> canonical Montgomery-2^256 limbs of field.hip.hpp
SHP_HD Fq to_fq(const Fq29& a) {
    Fq29 c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.l[i] = f29::C256.l[i];
    Fq29 t = mul29(a, c);                                   // a 2^256 2^-261 = a 2^-5, below 1.1 p
    Fq r;
#pragma unroll
    for (int j = 0; j < 8; j++) {                           // word j = bits [32 j, 32 j + 32)
        const int i = (32 * j) / 29, s = (32 * j) % 29;
        uint32_t v = t.l[i] >> s;
        v |= t.l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) v |= t.l[i + 2] << (58 - s);
        r.l[j] = v;
    }
    r.reduce_once();
    return r;
}
// canonical Montgomery-2^256 -> Montgomery-2^261, reduced below 1.1 p (one product by 2^266)
SHP_HD Fq29 from_fq_reduced(const Fq& x) {
    Fq29 c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.l[i] = f29::C266.l[i];
    return mul29(split29_dev<0>(x), c);
}

// XYZZ accumulator in the lazy form.  Bounds between additions: x < 10.5 p, y < 5.5 p, zz, zzz < 2.5 p.
struct G1X29 {
    Fq29 x, y, zz, zzz;
    uint32_t inf;            // a full word: a bool leaves three padding bytes, and copying those goes through scratch memory
};

SHP_HD G1X to_g1x(const G1X29& a) {
    if (a.inf) return G1X::inf();
    G1X r;
    r.x = to_fq(a.x); r.y = to_fq(a.y); r.zz = to_fq(a.zz); r.zzz = to_fq(a.zzz);
    return r;
}
SHP_HD G1X29 from_g1x(const G1X& p) {
    G1X29 r;
    r.inf = p.is_inf();
    r.x = from_fq_reduced(p.x); r.y = from_fq_reduced(p.y); r.zz = from_fq_reduced(p.zz); r.zzz = from_fq_reduced(p.zzz);
    return r;
}

// acc += (qx, qy), an affine point that is not the identity, given in the 8 x 32 Montgomery-2^256 form of the tables
// (madd-2008-s, 8 products + 2 squarings).  The doubling / opposite-point case (u2 = x1) cannot be seen on lazy values
// without a full reduction; it needs P = u2 - x1 + 11 p to be one of the 14 multiples of p below 13.5 p: lanes whose low
// limb says so (about one in 2^25 otherwise) take the exact 8 x 32 path.
SHP_HD void g1_madd29(G1X29& acc, const Fq& qx, const Fq& qy) {
    if (acc.inf) {
        acc.x = from_fq_reduced(qx);
        acc.y = from_fq_reduced(qy);
#pragma unroll
        for (int i = 0; i < 9; i++) { acc.zz.l[i] = f29::ONE.l[i]; acc.zzz.l[i] = f29::ONE.l[i]; }
        acc.inf = false;
        return;
    }
    const Fq29 x2 = split29_dev<5>(qx), y2 = split29_dev<5>(qy);          // < 32 p
    const Fq29 u2 = mul29(x2, acc.zz), s2 = mul29(y2, acc.zzz);
    const Fq29 P = sub29(u2, acc.x, f29::KP11);                           // < 13.5 p
    // P = K p (K < 14) has the low limb K p mod 2^29, so K = l[0] p^-1 mod 2^29: anything else gives a value >= 14
    // except once in 2^29 / 14
    const bool suspicious = ((P.l[0] * f29::PPOSINV29) & M29) < 14u;
    if (suspicious) {                                                      // rare: exact arithmetic on canonical values
        acc = from_g1x(g1_madd(to_g1x(acc), qx, qy));
        return;
    }
    const Fq29 Rr = sub29(s2, acc.y, f29::KP6);                           // < 8.5 p
    const Fq29 PP = sqr29(P);
    const Fq29 PPP = mul29(P, PP);
    const Fq29 Q = mul29(acc.x, PP);
    const Fq29 R2 = sqr29(Rr);
    Fq29 X3;
#pragma unroll
    for (int i = 0; i < 9; i++) X3.l[i] = R2.l[i] + f29::KP8W.l[i] - PPP.l[i] - (Q.l[i] << 1);   // R^2 - PPP - 2Q + 8p; 8p spread by 4 x 2^29 a limb
    normalise29(X3);                                                       // < 10.5 p
    Fq29 T;
#pragma unroll
    for (int i = 0; i < 9; i++) T.l[i] = Q.l[i] + f29::KP11.l[i] - X3.l[i];  // < 13.5 p; limbs below 2^29 + 2^30: one lazy operand
    const Fq29 RT = mul29(Rr, T);
    const Fq29 YP = mul29(acc.y, PPP);
    acc.y = sub29(RT, YP, f29::KP3);                                      // < 5.5 p
    acc.x = X3;
    acc.zz = mul29(acc.zz, PP);
    acc.zzz = mul29(acc.zzz, PPP);
}

// acc + q, both in the lazy form with the accumulator bounds above (add-2008-s, 12 products + 2 squarings); the result
// keeps those bounds.  P = u2 - u1 + 3 p < 5.5 p is a multiple of p exactly when the points share their x: those lanes
// (and the one in 2^29 / 6 whose low limb says so by chance) take the exact 8 x 32 path.
SHP_HD G1X29 g1_add29(const G1X29& a, const G1X29& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    const Fq29 u1 = mul29(a.x, b.zz), u2 = mul29(b.x, a.zz);
    const Fq29 s1 = mul29(a.y, b.zzz), s2 = mul29(b.y, a.zzz);
    const Fq29 P = sub29(u2, u1, f29::KP3);                               // < 5.5 p
    if (((P.l[0] * f29::PPOSINV29) & M29) < 6u) return from_g1x(g1_add(to_g1x(a), to_g1x(b)));
    const Fq29 Rr = sub29(s2, s1, f29::KP3);                              // < 5.5 p
    const Fq29 PP = sqr29(P);
    const Fq29 PPP = mul29(P, PP);
    const Fq29 Q = mul29(u1, PP);
    const Fq29 R2 = sqr29(Rr);
    G1X29 r;
    r.inf = false;
#pragma unroll
    for (int i = 0; i < 9; i++) r.x.l[i] = R2.l[i] + f29::KP8W.l[i] - PPP.l[i] - (Q.l[i] << 1);
    normalise29(r.x);                                                      // < 10.5 p
    Fq29 T;
#pragma unroll
    for (int i = 0; i < 9; i++) T.l[i] = Q.l[i] + f29::KP11.l[i] - r.x.l[i];  // < 13.5 p, one lazy operand
    r.y = sub29(mul29(Rr, T), mul29(s1, PPP), f29::KP3);                  // < 5.5 p
    r.zz = mul29(mul29(a.zz, b.zz), PP);
    r.zzz = mul29(mul29(a.zzz, b.zzz), PPP);
    return r;
}

}  // namespace shp
