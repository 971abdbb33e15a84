// BN254 Fq in 9 unsaturated 29-bit limbs (lazy29.hip.hpp) for the point additions of the MSM kernels: the mixed and the
// general XYZZ addition with their bounds.  Tables and results stay in the 8 x 32-bit Montgomery-2^256 form of field.hip.hpp;
// conversions happen once per gathered point and once per accumulator.
#pragma once
#include "curve.hip.hpp"
#include "lazy29.hip.hpp"

namespace shp {

using LQ29 = Lazy29<FqParams>;
using Fq29 = LQ29::El;

namespace f29 {
using L9 = l29::L9;
constexpr L9 P29 = LQ29::P29;
constexpr uint32_t PINV29 = LQ29::PINV29, PPOSINV29 = LQ29::PPOSINV29;
constexpr L9 ONE = LQ29::ONE;
constexpr L9 KP3 = LQ29::kp_spread(3), KP6 = LQ29::kp_spread(6), KP8W = LQ29::kp_spread(8, 4), KP11 = LQ29::kp_spread(11);
}  // namespace f29

template <int SHIFT>
SHP_HD Fq29 split29_dev(const Fq& v) { return LQ29::split<SHIFT>(v); }
SHP_HD Fq29 mul29(const Fq29& a, const Fq29& b) { return LQ29::mul(a, b); }
SHP_HD Fq29 sqr29(const Fq29& a) { return LQ29::sqr(a); }
SHP_HD void normalise29(Fq29& a) { LQ29::normalise(a); }
SHP_HD Fq29 sub29(const Fq29& a, const Fq29& b, const f29::L9& kp) { return LQ29::sub(a, b, kp); }
SHP_HD Fq to_fq(const Fq29& a) { return LQ29::to_fp(a); }
SHP_HD Fq29 from_fq_reduced(const Fq& x) { return LQ29::from_fp_reduced(x); }

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

// ---- Homogeneous projective points with the complete addition of Renes, Costello and Batina (EUROCRYPT 2016, algorithm 7:
// a = 0) for the reduction trees of the MSM kernels.  An addition is twelve products in TWO rounds of six independent ones
// (add-2008-s: fourteen in four rounds) and has no exceptional cases on a curve of prime order -- the identity, equal and
// opposite points go through the same formulas -- so six lanes can share an addition without flags or a fallback.
//
// A point is held as (X : Y : Zs) with Zs = 3 Z: then 3b Z1 Z2 = 9 Z1 Z2 (b = 3) is simply Zs1 Zs2, the cross terms with Zs
// come out as 3 t4 and 3 t5, and the only constant left in the formulas is the 9 of x9 = 9 X1 X2:
//   round 1   t0 = X1 X2 | t1 = Y1 Y2 | t2 = Zs1 Zs2 | m3 = (X1+Y1)(X2+Y2) | m4 = (Y1+Zs1)(Y2+Zs2) | m5 = (X1+Zs1)(X2+Zs2)
//   linear    t3 = m3 - t0 - t1,  t4 = m4 - t1 - t2,  t5 = m5 - t0 - t2,  d = t1 - t2,  s = t1 + t2,  x9 = 9 t0
//   round 2   X3 = t3 d - t4 t5 | Y3 = d s + x9 t5 | Zs3 = s t4 + x9 t3
// The identity is (0 : anything but 0 : 0).  Bounds: coordinates < 4.5 p with limbs 0..7 below 2^29 going in and coming out
// (round 1: sums < 9 p with limbs < 2^30 in BOTH operands -- a column of nine such partial products, the nine q p terms and
// the carry stay below 11.3 * 2^60; products < 1.5 p; the linear forms add 3 p and stay below 5.3 p, x9 + 3 p < 13.2 p;
// round 2: products < 1.4 p; X3 < 4.2 p, Y3 and Zs3 < 2.6 p).
struct G1P29 {
    Fq29 x, y, z;
};

namespace f29 {
constexpr L9 KP3S2 = LQ29::kp_spread(3, 2);                  // 3 p for a - b - c (limbs of b + c below 2^30)
}

// a + (b & mask), limb by limb (round-1 operands: a coordinate, or the sum of two)
SHP_HD Fq29 rcb_pair(const Fq29& a, const Fq29& b, uint32_t mask) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (b.l[i] & mask);
    return r;
}
// 9 v for a normalised v < 1.2 p: limbs below 2^30
SHP_HD Fq29 rcb_times9(const Fq29& v) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = v.l[i] << 3;        // < 2^32; the carries below cannot overflow a limb
    normalise29(r);
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] += v.l[i];
    return r;
}
// a - (b & mb) - (c & mc) + 3 p, or with mp = mc = ~0: a - (b & mb) + c + 3 p; normalised.  b + c < 2.5 p.
SHP_HD Fq29 rcb_lin(const Fq29& a, const Fq29& b, uint32_t mb, const Fq29& c, uint32_t mc, uint32_t mp) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + f29::KP3S2.l[i] - (b.l[i] & mb) - (c.l[i] & mc) + ((c.l[i] & mp) << 1);
    normalise29(r);
    return r;
}
// p - q + 3 p (minus) or p + q; normalised
SHP_HD Fq29 rcb_out(const Fq29& p, const Fq29& q, bool minus) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p.l[i] + (minus ? f29::KP3.l[i] - q.l[i] : q.l[i]);
    normalise29(r);
    return r;
}

SHP_HD G1P29 g1p_identity() {
    G1P29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.l[i] = 0; r.y.l[i] = f29::ONE.l[i]; r.z.l[i] = 0; }
    return r;
}
// an XYZZ accumulator (x = X / ZZ, y = Y / ZZZ) as (X ZZZ : Y ZZ : 3 ZZ ZZZ)
SHP_HD G1P29 g1p_from_x29(const G1X29& a) {
    if (a.inf) return g1p_identity();
    G1P29 r;
    r.x = mul29(a.x, a.zzz);
    r.y = mul29(a.y, a.zz);
    const Fq29 z = mul29(a.zz, a.zzz);
#pragma unroll
    for (int i = 0; i < 9; i++) r.z.l[i] = z.l[i] * 3u;
    normalise29(r.z);                                          // < 3.2 p
    return r;
}
// one lane's whole addition: the same rounds and linear forms as the six-lane version in the MSM kernel
SHP_HD G1P29 g1p_add29(const G1P29& a, const G1P29& b) {
    const Fq29 t0 = mul29(a.x, b.x), t1 = mul29(a.y, b.y), t2 = mul29(a.z, b.z);
    const Fq29 m3 = mul29(rcb_pair(a.x, a.y, ~0u), rcb_pair(b.x, b.y, ~0u));
    const Fq29 m4 = mul29(rcb_pair(a.y, a.z, ~0u), rcb_pair(b.y, b.z, ~0u));
    const Fq29 m5 = mul29(rcb_pair(a.x, a.z, ~0u), rcb_pair(b.x, b.z, ~0u));
    const Fq29 t3 = rcb_lin(m3, t0, ~0u, t1, ~0u, 0u), t4 = rcb_lin(m4, t1, ~0u, t2, ~0u, 0u), t5 = rcb_lin(m5, t0, ~0u, t2, ~0u, 0u);
    const Fq29 d = rcb_lin(t1, t2, ~0u, t2, 0u, 0u), s = rcb_lin(t1, t2, 0u, t2, ~0u, ~0u);
    const Fq29 x9 = rcb_lin(rcb_times9(t0), t0, 0u, t0, 0u, 0u);
    G1P29 r;
    r.x = rcb_out(mul29(t3, d), mul29(t4, t5), true);
    r.y = rcb_out(mul29(d, s), mul29(x9, t5), false);
    r.z = rcb_out(mul29(s, t4), mul29(x9, t3), false);
    return r;
}
// (X : Y : Zs) -> XYZZ in the canonical 8 x 32-bit form: x = 3 X / Zs, y = 3 Y / Zs, so (3 X Zs, 3 Y Zs^2, Zs^2, Zs^3)
SHP_HD G1X g1p_to_g1x(const G1P29& a) {
    const Fq29 zz = sqr29(a.z);
    const Fq29 zzz = mul29(zz, a.z);
    Fq29 x3, y3;
#pragma unroll
    for (int i = 0; i < 9; i++) { x3.l[i] = a.x.l[i] * 3u; y3.l[i] = a.y.l[i] * 3u; }     // < 13.5 p, limbs < 2^31: the lazy operand
    G1X r;
    r.zz = to_fq(zz);
    if (r.zz.is_zero()) return G1X::inf();
    r.zzz = to_fq(zzz);
    r.x = to_fq(mul29(x3, a.z));
    r.y = to_fq(mul29(y3, zz));
    return r;
}

}  // namespace shp
