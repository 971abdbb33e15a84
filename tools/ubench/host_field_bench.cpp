// Host field arithmetic as the witness program uses it (dependent chains of Fr additions and Montgomery products on
// 4 x 64-bit limbs).   clang++ -O3 -std=c++17 -Izkos-monorepo_amd/csrc -o tools/ubench/bin/host_field_bench tools/ubench/host_field_bench.cpp
// On the GPU box's EPYC: add 3.2 ns, product 17.8 ns in a dependent chain (a branch-free conditional subtraction and
// -mbmi2 -madx change neither): the 1772 cell operations of one Poseidon2 permutation hold ~19 us of arithmetic,
// the witness interpreter runs them in ~31 us.
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#include "field.hip.hpp"
using namespace shp;
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
    const int N = 4096;
    std::vector<Fr> v(N);
    for (auto& x : v) { Fr lo, hi; for (int i = 0; i < 8; i++) { lo.l[i] = (uint32_t)rnd(); hi.l[i] = (uint32_t)rnd(); } x = Fr::from_u512(lo, hi); }
    // dependent chains, as in a permutation
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        Fr acc = v[0];
        for (int it = 0; it < 200; it++) for (int i = 0; i < N; i++) acc = acc + v[i];
        auto t1 = std::chrono::steady_clock::now();
        Fr m = v[1];
        for (int it = 0; it < 200; it++) for (int i = 0; i < N; i++) m = m * v[i];
        auto t2 = std::chrono::steady_clock::now();
        Fr q = v[2];
        for (int it = 0; it < 100; it++) for (int i = 0; i < N; i++) { q = q * v[i]; q = q + v[(i + 7) & (N - 1)]; q = q + acc; }
        auto t3 = std::chrono::steady_clock::now();
        double na = 200.0 * N, nm = 200.0 * N;
        printf("add %.2f ns  mul %.2f ns  mix(mul+2add) %.2f ns   [%u %u %u]\n",
               std::chrono::duration<double, std::nano>(t1 - t0).count() / na,
               std::chrono::duration<double, std::nano>(t2 - t1).count() / nm,
               std::chrono::duration<double, std::nano>(t3 - t2).count() / (100.0 * N), acc.l[0], m.l[0], q.l[0]);
    }
}
