// Poseidon2 t=8 permutation body shared by the device kernels and the host witness generator.
// Round structure: /root/reference/poseidon2-solidity/generate_t8.py:474-588.
#pragma once
#include "field.hip.hpp"
#include "poseidon2_constants.inc"

namespace shp {

SHP_HD Fr p2_sbox(const Fr& x) {
    Fr x2 = Fr::sqr(x);
    Fr x3 = x2 * x;
    Fr x6 = Fr::sqr(x3);
    return x6 * x;
}

// circ(2*M4, M4) with M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]], additions only
SHP_HD void p2_m4(Fr& a, Fr& b, Fr& c, Fr& d) {
    Fr ab = a + b, cd = c + d;
    Fr t2 = Fr::dbl(b) + cd;        //      2b +  c +  d
    Fr t3 = Fr::dbl(d) + ab;        //  a +  b      + 2d
    Fr cd4 = Fr::dbl(Fr::dbl(cd));
    Fr ab4 = Fr::dbl(Fr::dbl(ab));
    Fr t4 = cd4 + t3;               //  a +  b + 4c + 6d
    Fr t5 = ab4 + t2;               // 4a + 6b +  c +  d
    a = t3 + t5;
    b = t5;
    c = t2 + t4;
    d = t4;
}
SHP_HD void p2_external(Fr s[8]) {
    p2_m4(s[0], s[1], s[2], s[3]);
    p2_m4(s[4], s[5], s[6], s[7]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        Fr w = s[i] + s[i + 4];
        s[i] = s[i] + w;
        s[i + 4] = s[i + 4] + w;
    }
}

template <class RCF, class RCP, class DG>
SHP_HD void poseidon2_permute_impl(Fr s[8], const RCF& rc_full, const RCP& rc_part, const DG& diag) {
    p2_external(s);
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = p2_sbox(s[i] + rc_full[r][i]);
        p2_external(s);
    }
    for (int r = 0; r < 48; r++) {
        s[0] = p2_sbox(s[0] + rc_part[r]);
        Fr sum = s[0];
#pragma unroll
        for (int i = 1; i < 8; i++) sum = sum + s[i];
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = diag[i] * s[i] + sum;
    }
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = p2_sbox(s[i] + rc_full[r][i]);
        p2_external(s);
    }
}

void poseidon2_permute_host(Fr s[8]);
// The permutation's states row by row, as the circuits' Poseidon2 chip lays them out: cell (cols[i], r0) of the column-major
// grid (col_stride cells a column) holds the input state; row r0 + 1 receives the state after the first linear layer,
// then one row per round (4 full, 48 partial, 4 full): rows r0 + 1 .. r0 + 56 are written.
constexpr uint32_t P2_TRACE_ROWS = 56;
void poseidon2_trace_host(Fr* grid, size_t col_stride, const uint32_t cols[8], uint32_t r0);
Fr poseidon2_hash_host(const Fr* in, uint32_t arity);

}  // namespace shp
