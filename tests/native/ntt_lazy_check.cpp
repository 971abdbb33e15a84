// CPU check of the lazy Fr arithmetic of the NTT butterflies (zkos-monorepo_amd/csrc/ntt_lazy.hip.hpp; both it and the field
// compile for the host): 32 x by shifts against the product, and chains of butterflies on values in [0, 4p) against the
// reduced arithmetic -- every intermediate below 4p, every residue equal.  Exit code 0 and a line "ok ..." when every
// comparison holds; the first mismatch is printed and the exit code is 1.
#include <stdio.h>
#include <stdlib.h>
#include "../../zkos-monorepo_amd/csrc/ntt_lazy.hip.hpp"

using namespace shp;

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static Fr rnd_fr() {
    Fr lo, hi;
    for (int i = 0; i < 8; i++) { lo.l[i] = (uint32_t)rnd(); hi.l[i] = (uint32_t)rnd(); }
    return Fr::from_u512(lo, hi);
}
// raw 256-bit helpers
static bool raw_less(const Fr& a, const uint32_t* b) {
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] < b[i]) return true;
        if (a.l[i] > b[i]) return false;
    }
    return false;
}
static Fr raw_add_kp(const Fr& a, int k) {          // a + k p as 256-bit integers (the caller keeps it below 2^256)
    Fr r = a;
    for (int j = 0; j < k; j++) {
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) { c += (uint64_t)r.l[i] + FrParams::p(i); r.l[i] = (uint32_t)c; c >>= 32; }
    }
    return r;
}
static Fr canon(Fr v) {                              // the residue of a value below 4p
    fr_reduce_2p(v);
    v.reduce_once();
    return v;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    uint32_t p2[8], p4[8];
    for (int i = 0; i < 8; i++) p2[i] = fr_2p(i);
    for (int i = 0; i < 8; i++) p4[i] = (p2[i] << 1) | (i ? p2[i - 1] >> 31 : 0);
    const Fr c32 = Fr::from_u64(32);

    // 1. 32 x by shifts: random values, the ends of the range, and the values around every multiple of p / 32 (where the
    //    quotient estimate changes)
    long n32 = 0;
    auto check32 = [&](const Fr& x) {
        if (x.geq_p()) return true;
        n32++;
        const Fr got = fr_times32(x), want = x * c32;
        if (got != want) { printf("fr_times32 mismatch\n"); return false; }
        return true;
    };
    for (int it = 0; it < rounds; it++)
        if (!check32(rnd_fr())) return 1;
    {
        Fr z = Fr::zero(), one_raw = Fr::zero(), pm1 = Fr::modulus();
        one_raw.l[0] = 1;
        pm1.l[0] -= 1;
        if (!check32(z) || !check32(one_raw) || !check32(pm1)) return 1;
        // floor(m p / 32) + d for m = 1..31, d = -2..2, as raw integers
        for (int m = 1; m < 32; m++) {
            // m p as a 9-limb integer, then >> 5
            uint64_t c = 0;
            uint32_t w[9];
            for (int i = 0; i < 8; i++) { c += (uint64_t)m * FrParams::p(i); w[i] = (uint32_t)c; c >>= 32; }
            w[8] = (uint32_t)c;
            Fr base;
            for (int i = 0; i < 8; i++) base.l[i] = (w[i] >> 5) | (w[i + 1] << 27);
            for (int d = -2; d <= 2; d++) {
                Fr v = base;
                if (d >= 0) { uint64_t cc = (uint64_t)d; for (int i = 0; i < 8 && cc; i++) { cc += v.l[i]; v.l[i] = (uint32_t)cc; cc >>= 32; } }
                else { uint64_t bw = (uint64_t)(-d); for (int i = 0; i < 8 && bw; i++) { uint64_t dd = (uint64_t)v.l[i] - bw; v.l[i] = (uint32_t)dd; bw = (dd >> 32) & 1; } }
                if (!check32(v)) return 1;
            }
        }
    }

    // 2. butterflies on lazy values.  The product of the device leaves any value below 2p in the residue class: here it is
    //    the reduced product plus p or not, at random.  x and y enter anywhere in [0, 4p).
    long nb = 0;
    for (int it = 0; it < rounds; it++) {
        Fr x = rnd_fr(), y = rnd_fr();
        const Fr xr = x, yr = y;                       // reduced twins
        x = raw_add_kp(x, (int)(rnd() % 4));           // p < 2^254: x + 3p < 4p
        y = raw_add_kp(y, (int)(rnd() % 4));
        if (!raw_less(x, p4) || !raw_less(y, p4)) { printf("test input out of range\n"); return 1; }
        Fr xw = xr, yw = yr;
        for (int stage = 0; stage < 12; stage++) {
            const Fr w = rnd_fr();
            // lazy
            Fr ty = Fr::mul(y, w);                     // the host product accepts any 256-bit operand against a reduced one
            if (rnd() & 1) ty = raw_add_kp(ty, 1);
            if (!raw_less(ty, p2)) { printf("product out of range\n"); return 1; }
            fr_reduce_2p(x);
            if (!raw_less(x, p2)) { printf("reduce_2p out of range\n"); return 1; }
            const Fr u = fr_add_raw(x, ty), d = fr_sub_2p(x, ty);
            if (!raw_less(u, p4) || !raw_less(d, p4)) { printf("butterfly output not below 4p\n"); return 1; }
            // reduced
            const Fr tw = yw * w;
            const Fr uw = xw + tw, dw = xw - tw;
            if (canon(u) != uw || canon(d) != dw) { printf("butterfly mismatch at %d stage %d\n", it, stage); return 1; }
            // next stage: cross over so that both outputs are exercised as either operand
            if (stage & 1) { x = u; y = d; xw = uw; yw = dw; }
            else { x = d; y = u; xw = dw; yw = uw; }
            nb++;
        }
    }
    printf("ok %ld shifts by five, %ld lazy butterflies\n", n32, nb);
    return 0;
}
