// Prime fields of BN254 in 9 UNSATURATED 29-bit limbs with lazy reduction (device and host), generic over the field.
//
// Why a second representation: on gfx950 every integer VALU instruction costs the same four cycles, so a Montgomery
// product is priced by its instruction count.  With saturated 32-bit limbs (field.hip.hpp) each of the 128 partial products is
// a v_mad_u64_u32 PLUS a v_addc_co_u32 for the carry out of the 64-bit accumulator.  With 29-bit limbs a column of 9 + 9
// partial products (< 2^58 each) cannot overflow 64 bits: a partial product is ONE v_mad_u64_u32, 162 in all, and the
// Montgomery radix 2^261 exceeds 4p by so much that no product needs a conditional subtraction.  Measured with every CU busy
// (tools/ubench/mul_variants.hip): 174 G products/s against 133 G for the saturated form (and 115 G for a 5 x 52-bit FP64
// formulation).
//
// Values are held LAZILY: an element spells an integer v = sum l[i] 2^(29 i) congruent to x * 2^261 (mod p), with limbs
// 0..7 below 2^29 ("normalised") and v below a small multiple of p that every call site tracks:
//   mul(a, b)     needs a < A p, b < B p with A B <= 254 and at most one operand with limbs up to 2^31; returns a
//                 normalised value < (A B / 164 + 1) p   (v = (a b + q p) / 2^261 < a b / 2^261 + p, p / 2^261 < 1 / 164)
//   sub(a, b, kp) = a - b + K p with kp = kp_spread(K, S): needs b < (K - 1/2) p and the limbs of b below S 2^29 - S + 1;
//                 returns < a + K p, normalised
#pragma once
#include "field.hip.hpp"

namespace shp {

constexpr uint32_t M29 = (1u << 29) - 1;

namespace l29 {
struct L8 { uint32_t l[8]; };
struct L9 { uint32_t l[9]; };

template <class PR>
constexpr L8 modulus() {
    L8 r{};
    for (int i = 0; i < 8; i++) r.l[i] = PR::p(i);
    return r;
}
constexpr bool geq(const L8& a, const L8& b) {
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    }
    return true;
}
template <class PR>
constexpr L8 dbl_mod_p(const L8& a) {                      // 2a mod p for a < p < 2^254
    L8 r{};
    uint32_t carry = 0;
    for (int i = 0; i < 8; i++) { r.l[i] = (a.l[i] << 1) | carry; carry = a.l[i] >> 31; }
    const L8 p = modulus<PR>();
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
template <class PR>
constexpr L8 pow2_mod_p(int e) {                           // 2^e mod p, e >= 256 (from R mod p = 2^256 mod p)
    L8 r{};
    for (int i = 0; i < 8; i++) r.l[i] = PR::r1(i);
    for (int i = 256; i < e; i++) r = dbl_mod_p<PR>(r);
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
}  // namespace l29

template <class PR>
struct Lazy29 {
    struct El { uint32_t l[9]; };
    using L9 = l29::L9;
    static constexpr L9 P29 = l29::split29(l29::modulus<PR>());
    static constexpr uint32_t PINV29 = PR::INV & M29;               // -p^-1 mod 2^29
    static constexpr uint32_t PPOSINV29 = (0u - PR::INV) & M29;     // p^-1 mod 2^29
    static constexpr L9 ONE = l29::split29(l29::pow2_mod_p<PR>(261));    // Montgomery one for R = 2^261
    static constexpr L9 C266 = l29::split29(l29::pow2_mod_p<PR>(266));   // Montgomery-2^256 limbs (no shift) -> Montgomery-2^261, reduced
    static constexpr L9 C256 = l29::split29(l29::pow2_mod_p<PR>(256));   // Montgomery-2^261 -> Montgomery-2^256

    // K p with limbs 0..7 in [S 2^29 - S, (S + 1) 2^29): added limb by limb it keeps a - b + K p non-negative in EVERY
    // limb as long as the limbs of b stay below S 2^29 - S + 1, so the subtraction needs no signed arithmetic
    static constexpr L9 kp_spread(uint32_t K, int64_t S = 1) {
        const L9 p = P29;
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

    // bits [29 i, 29 i + 29) of the 256-bit value shifted left by SHIFT bits (SHIFT = 5: Montgomery-2^256 -> 2^261,
    // value < 32 p; SHIFT = 0: the integer itself)
    template <int SHIFT>
    static SHP_HD El split(const Fp<PR>& v) {
        El r;
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
    static SHP_HD El from_l9(const L9& c) {
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c.l[i];
        return r;
    }

    // a b 2^-261 mod p (lazy); see the bounds in the header comment
    static SHP_HD El mul(const El& a, const El& b) {
        uint64_t acc = 0;
        uint32_t q[9];
        El r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * P29.l[k - i];
            q[k] = ((uint32_t)acc * PINV29) & M29;
            acc += (uint64_t)q[k] * P29.l[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * P29.l[k - i];
            r.l[k - 9] = (uint32_t)acc & M29;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }

    // a^2 2^-261: the 36 cross products once, against a doubled operand (2 a_i < 2^30: a column stays below 2^64)
    static SHP_HD El sqr(const El& a) {
        uint64_t acc = 0;
        uint32_t q[9], d[9];
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; 2 * i < k; i++) acc += (uint64_t)d[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * P29.l[k - i];
            q[k] = ((uint32_t)acc * PINV29) & M29;
            acc += (uint64_t)q[k] * P29.l[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)d[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * P29.l[k - i];
            r.l[k - 9] = (uint32_t)acc & M29;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }

    // limbs 0..7 back below 2^29 (every limb non-negative on entry)
    static SHP_HD void normalise(El& a) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            a.l[i + 1] += a.l[i] >> 29;
            a.l[i] &= M29;
        }
    }

    // a - b + K p, normalised (kp = kp_spread(K, S))
    static SHP_HD El sub(const El& a, const El& b, const L9& kp) {
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + kp.l[i] - b.l[i];
        normalise(r);
        return r;
    }

    // Montgomery-2^261 lazy value -> canonical Montgomery-2^256 limbs of field.hip.hpp
    static SHP_HD Fp<PR> to_fp(const El& a) {
        const El t = mul(a, from_l9(C256));                       // a 2^256 2^-261 = a 2^-5, below 1.1 p
        Fp<PR> r;
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
    static SHP_HD El from_fp_reduced(const Fp<PR>& x) { return mul(split<0>(x), from_l9(C266)); }
};

}  // namespace shp
