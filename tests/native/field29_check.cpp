// CPU check of the 29-bit-limb Fq and its mixed addition against the 8 x 32-bit arithmetic (both compile for the host).
// Exit code 0 and a line "ok ..." when every comparison holds; the first mismatch is printed and the exit code is 1.
#include <stdio.h>
#include <stdlib.h>
#include <initializer_list>
#include "../../zkos-monorepo_amd/csrc/field29.hip.hpp"

using namespace shp;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static Fq rnd_fq() {
    Fq lo, hi;
    for (int i = 0; i < 8; i++) { lo.l[i] = (uint32_t)rnd(); hi.l[i] = (uint32_t)rnd(); }
    return Fq::from_u512(lo, hi);
}
static bool same_point(const G1X& a, const G1X& b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    const G1Affine x = g1_to_affine(a), y = g1_to_affine(b);
    return x.x == y.x && x.y == y.y;
}
// v (Montgomery-2^261, lazy) as a canonical Montgomery-2^256 value
static Fq canon(const Fq29& v) { return to_fq(v); }

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    // 1. products and squares, operands pushed to the top of their allowed ranges by adding multiples of p
    for (int it = 0; it < rounds; it++) {
        const Fq a = rnd_fq(), b = rnd_fq();
        Fq29 a29 = split29_dev<5>(a), b29 = from_fq_reduced(b);     // a29 < 32 p
        const int extra = (int)(rnd() % 6);                         // b29 + extra * p < 7.1 p: 32 * 7.1 < 254
        for (int e = 0; e < extra; e++) {
            for (int i = 0; i < 9; i++) b29.l[i] += f29::P29.l[i];
            normalise29(b29);
        }
        const Fq want = a * b;
        if (canon(mul29(a29, b29)) != want) { printf("mul29 mismatch at %d\n", it); return 1; }
        Fq29 s = from_fq_reduced(a);
        const int extra2 = (int)(rnd() % 13);                       // up to 13.1 p: 13.1^2 < 254
        for (int e = 0; e < extra2; e++) {
            for (int i = 0; i < 9; i++) s.l[i] += f29::P29.l[i];
            normalise29(s);
        }
        if (canon(sqr29(s)) != a * a) { printf("sqr29 mismatch at %d\n", it); return 1; }
        if (canon(mul29(s, s)) != a * a) { printf("mul29(s, s) mismatch at %d\n", it); return 1; }
        // a - b + K p
        Fq29 ra = from_fq_reduced(a), rb = from_fq_reduced(b);
        if (canon(sub29(ra, rb, f29::KP3)) != a - b) { printf("sub29 mismatch at %d\n", it); return 1; }
    }
    // 2. chains of mixed additions, with the exceptional cases spliced in
    Fq gx = Fq::one(), gy = Fq::one() + Fq::one();                  // the generator (1, 2)
    const int n_pts = 64;
    G1Affine pts[n_pts];
    G1X run = G1X::inf();
    for (int i = 0; i < n_pts; i++) {
        const int reps = 1 + (int)(rnd() % 5);
        for (int r = 0; r < reps; r++) run = g1_madd(run, gx, gy);
        if (i % 7 == 3) run = g1_dbl(run);
        pts[i] = g1_to_affine(run);
    }
    long adds = 0;
    for (int it = 0; it < rounds / 10 + 1; it++) {
        G1X ref = G1X::inf();
        G1X29 acc;
        acc.inf = true;
        G1Affine last = pts[0];
        const int len = 1 + (int)(rnd() % 40);
        for (int s = 0; s < len; s++) {
            G1Affine q = pts[rnd() % n_pts];
            const uint64_t kind = rnd() % 16;
            if (kind == 0 && s > 0) q = last;                        // often a doubling (when the accumulator equals q)
            if (kind == 1 && s > 0) {                                // accumulator == q exactly: force it
                if (!ref.is_inf()) q = g1_to_affine(ref);
            }
            if (kind == 2 && s > 0 && !ref.is_inf()) {               // accumulator == -q: the sum is the identity
                q = g1_to_affine(ref);
                q.y = Fq::neg(q.y);
            }
            if (kind == 3) q.y = Fq::neg(q.y);
            ref = g1_madd(ref, q.x, q.y);
            g1_madd29(acc, q.x, q.y);
            adds++;
            last = q;
            if (!same_point(to_g1x(acc), ref)) { printf("madd29 mismatch: chain %d step %d kind %d\n", it, s, (int)kind); return 1; }
        }
    }
    // 3. general additions: random pairs of partial sums, equal points (doubling), opposite points, the identity
    long gadds = 0;
    for (int it = 0; it < rounds / 4 + 1; it++) {
        G1X ra = G1X::inf(), rb = G1X::inf();
        G1X29 a29, b29;
        a29.inf = true; b29.inf = true;
        const int la = (int)(rnd() % 6), lb = (int)(rnd() % 6);       // 0 additions: the identity
        for (int s = 0; s < la; s++) { const G1Affine q = pts[rnd() % n_pts]; ra = g1_madd(ra, q.x, q.y); g1_madd29(a29, q.x, q.y); }
        for (int s = 0; s < lb; s++) { const G1Affine q = pts[rnd() % n_pts]; rb = g1_madd(rb, q.x, q.y); g1_madd29(b29, q.x, q.y); }
        const uint64_t kind = rnd() % 8;
        if (kind == 0) { rb = ra; b29 = a29; }                         // doubling (or identity + identity)
        if (kind == 1) { rb = ra.negated(); b29 = from_g1x(rb); }      // cancels
        if (kind == 2 && !ra.is_inf()) {                               // the same point under another representation
            const G1Affine q = g1_to_affine(ra);
            rb = G1X::from_affine(q); b29 = from_g1x(rb);
        }
        G1X want = g1_add(ra, rb);
        G1X29 got = g1_add29(a29, b29);
        gadds++;
        if (!same_point(to_g1x(got), want)) { printf("add29 mismatch: case %d kind %d\n", it, (int)kind); return 1; }
        // sums of sums keep the bounds: feed the result back in
        G1X want2 = g1_add(want, ra);
        G1X29 got2 = g1_add29(got, a29);
        if (!same_point(to_g1x(got2), want2)) { printf("add29 (second level) mismatch: case %d kind %d\n", it, (int)kind); return 1; }
    }
    // 3b. projective points with the complete addition: trees of partial sums with the exceptional cases spliced in, and the
    // coordinates pushed towards their bound (4.5 p) by adding multiples of p
    long padds = 0;
    auto lift = [&](Fq29& v, int times) {
        for (int e = 0; e < times; e++) {
            for (int i = 0; i < 9; i++) v.l[i] += f29::P29.l[i];
            normalise29(v);
        }
    };
    for (int it = 0; it < rounds / 4 + 1; it++) {
        G1X ra = G1X::inf(), rb = G1X::inf();
        G1X29 a29, b29;
        a29.inf = true; b29.inf = true;
        const int la = (int)(rnd() % 6), lb = (int)(rnd() % 6);
        for (int s = 0; s < la; s++) { const G1Affine q = pts[rnd() % n_pts]; ra = g1_madd(ra, q.x, q.y); g1_madd29(a29, q.x, q.y); }
        for (int s = 0; s < lb; s++) { const G1Affine q = pts[rnd() % n_pts]; rb = g1_madd(rb, q.x, q.y); g1_madd29(b29, q.x, q.y); }
        const uint64_t kind = rnd() % 8;
        if (kind == 0) { rb = ra; b29 = a29; }
        if (kind == 1) { rb = ra.negated(); b29 = from_g1x(rb); }
        if (kind == 2 && !ra.is_inf()) { rb = G1X::from_affine(g1_to_affine(ra)); b29 = from_g1x(rb); }
        G1P29 pa = g1p_from_x29(a29), pb = g1p_from_x29(b29);
        if (!same_point(g1p_to_g1x(pa), ra)) { printf("projective conversion mismatch: case %d\n", it); return 1; }
        lift(pa.x, (int)(rnd() % 4)); lift(pa.y, (int)(rnd() % 4)); lift(pa.z, (int)(rnd() % 2));      // < 4.5 p each
        lift(pb.x, (int)(rnd() % 4)); lift(pb.y, (int)(rnd() % 4)); lift(pb.z, (int)(rnd() % 2));
        const G1X want = g1_add(ra, rb);
        const G1P29 got = g1p_add29(pa, pb);
        padds++;
        if (!same_point(g1p_to_g1x(got), want)) { printf("projective add mismatch: case %d kind %d\n", it, (int)kind); return 1; }
        // sums of sums, the sum with itself, and the sum with its opposite
        const G1P29 got2 = g1p_add29(got, pa), dbl = g1p_add29(got, got);
        if (!same_point(g1p_to_g1x(got2), g1_add(want, ra))) { printf("projective add (second level) mismatch: case %d\n", it); return 1; }
        if (!same_point(g1p_to_g1x(dbl), g1_add(want, want))) { printf("projective doubling mismatch: case %d\n", it); return 1; }
        const G1P29 neg = g1p_from_x29(from_g1x(want.negated()));
        if (!g1p_to_g1x(g1p_add29(got, neg)).is_inf()) { printf("projective P + (-P) mismatch: case %d\n", it); return 1; }
        // coordinates stay below 4.5 p: the top limb of 4.5 p
        for (const Fq29* c : {&got2.x, &got2.y, &got2.z})
            if (c->l[8] > (uint32_t)((9ull * (f29::P29.l[8] + 1)) / 2)) { printf("projective bound exceeded: case %d\n", it); return 1; }
        padds += 4;
    }
    // 4. the same arithmetic over Fr (the quotient kernels): products, squares, subtractions, both conversions
    using LR = Lazy29<FrParams>;
    constexpr l29::L9 RKP3 = LR::kp_spread(3), RKP40 = LR::kp_spread(40, 2);
    for (int it = 0; it < rounds; it++) {
        Fr lo, hi, lo2, hi2;
        for (int i = 0; i < 8; i++) { lo.l[i] = (uint32_t)rnd(); hi.l[i] = (uint32_t)rnd(); lo2.l[i] = (uint32_t)rnd(); hi2.l[i] = (uint32_t)rnd(); }
        const Fr a = Fr::from_u512(lo, hi), b = Fr::from_u512(lo2, hi2);
        const LR::El a29 = LR::split<5>(a), b29 = LR::from_fp_reduced(b);          // < 32 p, < 1.1 p
        if (LR::to_fp(LR::mul(a29, b29)) != a * b) { printf("Fr mul mismatch at %d\n", it); return 1; }
        if (LR::to_fp(LR::sqr(b29)) != b * b) { printf("Fr sqr mismatch at %d\n", it); return 1; }
        if (LR::to_fp(LR::sub(LR::from_fp_reduced(a), b29, RKP3)) != a - b) { printf("Fr sub mismatch at %d\n", it); return 1; }
        // an un-normalised operand (three values added limb by limb) against a large multiple of p
        LR::El s3;
        for (int i = 0; i < 9; i++) s3.l[i] = a29.l[i] + b29.l[i] + b29.l[i];
        if (LR::to_fp(LR::mul(s3, b29)) != (a + b + b) * b) { printf("Fr lazy-operand mul mismatch at %d\n", it); return 1; }
        LR::El bb;
        for (int i = 0; i < 9; i++) bb.l[i] = a29.l[i] + b29.l[i];                 // < 33.1 p, limbs < 2^30
        if (LR::to_fp(LR::sub(b29, bb, RKP40)) != b - (a + b)) { printf("Fr wide sub mismatch at %d\n", it); return 1; }
    }
    printf("ok %d products, %ld additions, %ld general additions, %ld projective additions\n", rounds, adds, gadds, padds);
    return 0;
}
