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

}  // namespace shp
