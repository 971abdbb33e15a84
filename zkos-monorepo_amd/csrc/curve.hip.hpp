// BN254 G1 for the HIP path: affine bases as they sit in the SRS tables (64 B, Montgomery,
// (0,0) = identity: /root/reference/crates/powers-of-tau/lib.rs:190-231 and the RawBytes
// layout of resources/ppot_0080_11_raw) and extended-Jacobian "XYZZ" accumulators
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity ZZ = 0) for bucket sums: a mixed addition is
// 8M + 2S against 7M + 4S for Jacobian, and needs no field inversion until the host
// normalises the handful of final sums.
#pragma once
#include "field.hip.hpp"

namespace shp {

struct alignas(16) G1Affine {
    Fq x, y;
    SHP_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

struct alignas(16) G1X {       // XYZZ
    Fq x, y, zz, zzz;
    static SHP_HD G1X inf() {
        G1X r;
        r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero();
        return r;
    }
    SHP_HD bool is_inf() const { return zz.is_zero(); }
    static SHP_HD G1X from_affine(const G1Affine& p) {
        if (p.is_inf()) return inf();
        G1X r;
        r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one();
        return r;
    }
    SHP_HD G1X negated() const {
        G1X r = *this;
        r.y = Fq::neg(y);
        return r;
    }
};

// 2 * (affine p)   (mdbl-2008-s-1, a = 0)
SHP_HD G1X g1_dbl_affine(const Fq& x1, const Fq& y1) {
    G1X r;
    Fq u = Fq::dbl(y1);
    Fq v = Fq::sqr(u);
    Fq w = u * v;
    Fq s = x1 * v;
    Fq xx = Fq::sqr(x1);
    Fq m = Fq::dbl(xx) + xx;
    r.x = Fq::sqr(m) - Fq::dbl(s);
    r.y = m * (s - r.x) - w * y1;
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * p   (dbl-2008-s-1, a = 0)
SHP_HD G1X g1_dbl(const G1X& p) {
    if (p.is_inf()) return p;
    G1X r;
    Fq u = Fq::dbl(p.y);
    Fq v = Fq::sqr(u);
    Fq w = u * v;
    Fq s = p.x * v;
    Fq xx = Fq::sqr(p.x);
    Fq m = Fq::dbl(xx) + xx;
    r.x = Fq::sqr(m) - Fq::dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}

// p + (affine q), q given as (qx, qy) and known not to be the identity   (madd-2008-s)
SHP_HD G1X g1_madd(const G1X& p, const Fq& qx, const Fq& qy) {
    if (p.is_inf()) {
        G1X r;
        r.x = qx; r.y = qy; r.zz = Fq::one(); r.zzz = Fq::one();
        return r;
    }
    Fq u2 = qx * p.zz;
    Fq s2 = qy * p.zzz;
    Fq pp_ = u2 - p.x;
    Fq rr = s2 - p.y;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return g1_dbl_affine(qx, qy);
        return G1X::inf();
    }
    G1X r;
    Fq pp = Fq::sqr(pp_);
    Fq ppp = pp_ * pp;
    Fq q = p.x * pp;
    r.x = Fq::sqr(rr) - ppp - Fq::dbl(q);
    r.y = rr * (q - r.x) - p.y * ppp;
    r.zz = p.zz * pp;
    r.zzz = p.zzz * ppp;
    return r;
}

// p + q   (add-2008-s)
SHP_HD G1X g1_add(const G1X& p, const G1X& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fq u1 = p.x * q.zz;
    Fq u2 = q.x * p.zz;
    Fq s1 = p.y * q.zzz;
    Fq s2 = q.y * p.zzz;
    Fq pp_ = u2 - u1;
    Fq rr = s2 - s1;
    if (pp_.is_zero()) {
        if (rr.is_zero()) return g1_dbl(p);
        return G1X::inf();
    }
    G1X r;
    Fq pp = Fq::sqr(pp_);
    Fq ppp = pp_ * pp;
    Fq qq = u1 * pp;
    r.x = Fq::sqr(rr) - ppp - Fq::dbl(qq);
    r.y = rr * (qq - r.x) - s1 * ppp;
    r.zz = p.zz * q.zz * pp;
    r.zzz = p.zzz * q.zzz * ppp;
    return r;
}

// host-side normalisation of a few final sums
SHP_HD G1Affine g1_to_affine(const G1X& p) {
    G1Affine r;
    if (p.is_inf()) { r.x = Fq::zero(); r.y = Fq::zero(); return r; }
    Fq zi = Fq::inv(p.zzz);          // 1/ZZZ
    Fq zz_inv = zi * zi * p.zz * p.zz; // (ZZ/ZZZ)^2 = 1/ZZ since ZZ^3 = ZZZ^2
    r.x = p.x * zz_inv;
    r.y = p.y * zi;
    return r;
}

}  // namespace shp
