// Which 254-bit Montgomery multiplier should the MSM use on gfx950?  Three candidates, each validated against the
// library's own Fq::mul on the host and then timed with every CU busy (dependent chains, 8 waves per SIMD where the
// registers allow):
//   A  "fips32"   the shipped one: 8 x 32-bit saturated limbs, product scanning, one v_mad_u64_u32 + one v_addc_co_u32
//                 per partial product, R = 2^256, conditional subtraction at the end           (csrc/field_mul_gfx950.inc)
//   B  "lazy29"   9 x 29-bit UNSATURATED limbs, R = 2^261: a column of 18 partial products (< 2^58 each) fits a 64-bit
//                 accumulator, so a partial product is ONE v_mad_u64_u32 and nothing else; 2^261 > 4p leaves so much
//                 headroom that no conditional subtraction is needed (inputs and outputs stay below 2p)
//   C  "fp64x52"  5 x 52-bit limbs held as doubles, partial products split by round-toward-zero FMAs
//                 (hi = fma(a, b, 2^104), lo = fma(a, b, (2^104 + 2^52) - hi)), columns accumulated as integers on the
//                 bit patterns, R = 2^260
// Build: hipcc -O3 --offload-arch=gfx950 -I zkos-monorepo_amd/csrc -o tools/ubench/bin/mul_variants tools/ubench/mul_variants.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "field.hip.hpp"

using shp::Fq;

// ------------------------------------------------------------------------------------------------ B: 9 x 29 bits
constexpr uint32_t M29 = (1u << 29) - 1;
struct P29 {
    uint32_t p[9];
    uint32_t pinv;      // -p^-1 mod 2^29
};

__host__ __device__ inline void to29(const uint32_t in[8], uint32_t out[9]) {
    // bits [29 i, 29 i + 29) of the 256-bit value
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint64_t v = in[w];
        if (w + 1 < 8) v |= (uint64_t)in[w + 1] << 32;
        out[i] = (uint32_t)(v >> s) & M29;
    }
}
__host__ __device__ inline void from29(const uint32_t in[9], uint32_t out[8]) {     // limbs may exceed 29 bits slightly
    uint64_t acc = 0;
    int have = 0, o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (uint64_t)in[i] << have;
        have += 29;
        while (have >= 32 && o < 8) { out[o++] = (uint32_t)acc; acc >>= 32; have -= 32; }
    }
    while (o < 8) { out[o++] = (uint32_t)acc; acc >>= 32; }
}

// r = a b 2^-261 mod p, limbs of a and b below 2^29 (+ a few), values below 2p; so is r
__device__ __forceinline__ void mul29(const uint32_t a[9], const uint32_t b[9], uint32_t r[9], const P29& P) {
    uint64_t acc = 0;
    uint32_t q[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * P.p[k - i];
        q[k] = ((uint32_t)acc * P.pinv) & M29;
        acc += (uint64_t)q[k] * P.p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * P.p[k - i];
        r[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
}

// D: the same product by ROWS: 18 independent 64-bit column accumulators, so consecutive v_mad_u64_u32 never depend on
// each other (in B a column is one chain of up to 18 dependent instructions); costs a 64-bit carry addition per column
__device__ __forceinline__ void mul29_rows(const uint32_t a[9], const uint32_t b[9], uint32_t r[9], const P29& P) {
    uint64_t t[18];
#pragma unroll
    for (int i = 0; i < 18; i++) t[i] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[k + j] += (uint64_t)a[k] * b[j];
        const uint32_t q = ((uint32_t)t[k] * P.pinv) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[k + j] += (uint64_t)q * P.p[j];
        t[k + 1] += t[k] >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        t[k + 1] += t[k] >> 29;
        r[k - 9] = (uint32_t)t[k] & M29;
    }
    r[8] = (uint32_t)t[17];
}

// ------------------------------------------------------------------------------------------------ C: 5 x 52 bits in doubles
struct P52 {
    double p[5];
    uint64_t pinv;      // -p^-1 mod 2^52
};
constexpr uint64_t M52 = (1ull << 52) - 1;

__device__ __forceinline__ double fma_rz(double a, double b, double c) {
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));     // rounding mode: set once per kernel
    return d;
}
__device__ __forceinline__ double sub_rz(double a, double b) {
    double d;
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// column accumulators hold SUMS OF BIT PATTERNS: hi = 2^104 + h 2^52 has mantissa h, lo' = 2^52 + l has mantissa l, so adding
// the patterns as integers adds h and l (plus a known multiple of the exponent fields, removed at the end)
__device__ __forceinline__ void mac52(uint64_t& col_lo, uint64_t& col_hi, double a, double b) {
    const double C1 = 20282409603651670423947251286016.0;                  // 2^104
    const double C3 = 20282409603651674927546878656512.0;                  // 2^104 + 2^52
    const double hi = fma_rz(a, b, C1);
    const double lo = fma_rz(a, b, sub_rz(C3, hi));
    col_hi += (uint64_t)__double_as_longlong(hi);
    col_lo += (uint64_t)__double_as_longlong(lo);
}
__device__ __forceinline__ double u52_to_double(uint64_t v) {               // v < 2^52
    return __longlong_as_double((long long)(v | 0x4330000000000000ull)) - 4503599627370496.0;
}
// r = a b 2^-260 mod p (limbs < 2^52 as doubles, values < 2p)
__device__ __forceinline__ constexpr int pairs52(int c) { return c < 0 || c > 8 ? 0 : (c <= 4 ? c + 1 : 9 - c); }   // (i, j) with i + j = c
__device__ __forceinline__ void mul52(const double a[5], const double b[5], double r[5], const P52& P) {
    const uint64_t EHI = 0x4670000000000000ull, ELO = 0x4330000000000000ull;    // bit patterns of 2^104 and 2^52
    uint64_t col[11];
#pragma unroll
    for (int i = 0; i < 11; i++) col[i] = 0;
    // plain products: limb product (i, j) adds its low half to column i + j and its high half to column i + j + 1
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) mac52(col[i + j], col[i + j + 1], a[i], b[j]);
    // remove the exponent fields: column c received pairs52(c) low halves and pairs52(c - 1) high halves
#pragma unroll
    for (int c = 0; c < 10; c++) col[c] -= (uint64_t)pairs52(c) * ELO + (uint64_t)pairs52(c - 1) * EHI;
    // Montgomery reduction, one 52-bit limb at a time
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint64_t qk = (col[k] * P.pinv) & M52;
        const double qd = u52_to_double(qk);
#pragma unroll
        for (int j = 0; j < 5; j++) mac52(col[k + j], col[k + j + 1], qd, P.p[j]);
#pragma unroll
        for (int j = 0; j < 5; j++) { col[k + j] -= ELO; col[k + j + 1] -= EHI; }
        col[k + 1] += col[k] >> 52;                                           // the low 52 bits are zero now
    }
#pragma unroll
    for (int k = 5; k < 9; k++) { col[k + 1] += col[k] >> 52; col[k] &= M52; }
#pragma unroll
    for (int k = 0; k < 5; k++) r[k] = u52_to_double(col[5 + k] & M52);
}

// ------------------------------------------------------------------------------------------------ kernels
// every thread: x <- x * y, `iters` times; out = x
__global__ __launch_bounds__(256) void chain_fips32(const Fq* in, Fq* out, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = in[2 * t], y = in[2 * t + 1];
    for (uint32_t i = 0; i < iters; i++) x = x * y;
    out[t] = x;
}
__global__ __launch_bounds__(256) void chain_lazy29(const Fq* in, Fq* out, uint32_t iters, P29 P) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[9], y[9], r[9];
    to29(in[2 * t].l, x);
    to29(in[2 * t + 1].l, y);
    for (uint32_t i = 0; i < iters; i++) {
        mul29(x, y, r, P);
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = r[j];
    }
    Fq o;
    from29(x, o.l);
    out[t] = o;
}
__global__ __launch_bounds__(256) void chain_lazy29_rows(const Fq* in, Fq* out, uint32_t iters, P29 P) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[9], y[9], r[9];
    to29(in[2 * t].l, x);
    to29(in[2 * t + 1].l, y);
    for (uint32_t i = 0; i < iters; i++) {
        mul29_rows(x, y, r, P);
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = r[j];
    }
    Fq o;
    from29(x, o.l);
    out[t] = o;
}
__global__ __launch_bounds__(256) void chain_fp64x52(const Fq* in, Fq* out, uint32_t iters, P52 P) {
    // FP64 rounding mode of this wave: toward zero (MODE.FP_ROUND bits [3:2] govern f64)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    double x[5], y[5], r[5];
    auto load = [&](const Fq& v, double* d) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int bit = 52 * i, w = bit >> 5, s = bit & 31;
            uint64_t lo = v.l[w];
            if (w + 1 < 8) lo |= (uint64_t)v.l[w + 1] << 32;
            uint64_t val = lo >> s;
            if (w + 2 < 8 && s) val |= (uint64_t)v.l[w + 2] << (64 - s);
            d[i] = u52_to_double(val & M52);
        }
    };
    load(in[2 * t], x);
    load(in[2 * t + 1], y);
    for (uint32_t i = 0; i < iters; i++) {
        mul52(x, y, r, P);
#pragma unroll
        for (int j = 0; j < 5; j++) x[j] = r[j];
    }
    // back to 8 x 32 (limbs < 2^52)
    Fq o;
    uint64_t limb[5];
#pragma unroll
    for (int i = 0; i < 5; i++) limb[i] = (uint64_t)x[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {                                             // word j = bits [32 j, 32 j + 32)
        const int i = (32 * j) / 52, sh = (32 * j) % 52;
        uint64_t v = limb[i] >> sh;
        if (sh + 32 > 52 && i + 1 < 5) v |= limb[i + 1] << (52 - sh);
        o.l[j] = (uint32_t)v;
    }
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0");
    out[t] = o;
}

// ------------------------------------------------------------------------------------------------ host
static Fq fq_pow2(uint32_t e) {
    Fq two = Fq::from_u64(2), acc = Fq::one();
    for (uint32_t i = 0; i < e; i++) acc = acc * two;
    return acc;                                                              // Montgomery form of 2^e
}
static Fq reduce_raw(Fq v) {                                                 // raw value < 4p -> raw value mod p
    for (int i = 0; i < 3; i++) v.reduce_once();
    return v;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e-6;
    printf("device %s CUs %d clock %.2f GHz\n", prop.name, cus, clk);
    // constants
    P29 p29;
    uint32_t pl[8];
    for (int i = 0; i < 8; i++) pl[i] = shp::FqParams::p(i);
    to29(pl, p29.p);
    {
        uint32_t inv = 1, p0 = p29.p[0];                                     // Newton: inv = p0^-1 mod 2^32
        for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;
        p29.pinv = (0u - inv) & M29;
    }
    P52 p52;
    {
        uint64_t lim[5];
        for (int i = 0; i < 5; i++) {
            const int bit = 52 * i, w = bit >> 5, s = bit & 31;
            unsigned __int128 v = 0;
            for (int j = 0; j < 4 && w + j < 8; j++) v |= (unsigned __int128)pl[w + j] << (32 * j);
            lim[i] = (uint64_t)(v >> s) & M52;
            p52.p[i] = (double)lim[i];
        }
        uint64_t inv = 1;
        for (int i = 0; i < 7; i++) inv *= 2 - lim[0] * inv;
        p52.pinv = (0ull - inv) & M52;
    }
    const uint32_t threads = (uint32_t)cus * 8 * 256;                        // 8 waves per SIMD
    std::vector<Fq> h_in(2 * (size_t)threads), h_out(threads);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& v : h_in) {
        for (int i = 0; i < 8; i += 2) { uint64_t x = rnd(); v.l[i] = (uint32_t)x; v.l[i + 1] = (uint32_t)(x >> 32); }
        v.l[7] &= 0x1fffffffu;                                               // < 2^253 < p: canonical
    }
    Fq *d_in, *d_out;
    hipMalloc(&d_in, h_in.size() * sizeof(Fq));
    hipMalloc(&d_out, h_out.size() * sizeof(Fq));
    hipMemcpy(d_in, h_in.data(), h_in.size() * sizeof(Fq), hipMemcpyHostToDevice);
    dim3 grid(threads / 256), block(256);
    auto launch = [&](int which, uint32_t iters) {
        if (which == 0) hipLaunchKernelGGL(chain_fips32, grid, block, 0, 0, d_in, d_out, iters);
        else if (which == 1) hipLaunchKernelGGL(chain_lazy29, grid, block, 0, 0, d_in, d_out, iters, p29);
        else if (which == 2) hipLaunchKernelGGL(chain_fp64x52, grid, block, 0, 0, d_in, d_out, iters, p52);
        else hipLaunchKernelGGL(chain_lazy29_rows, grid, block, 0, 0, d_in, d_out, iters, p29);
    };
    // ---- validation: `iters` dependent products x <- x y 2^-w against the library's Fq::mul on the host
    const char* names[4] = {"A fips32", "B lazy29 (columns)", "C fp64x52", "D lazy29 (rows)"};
    const uint32_t rbits[4] = {256, 261, 260, 261};
    for (int which = 0; which < 4; which++) {
        const uint32_t iters = 7;
        launch(which, iters);
        hipDeviceSynchronize();
        hipMemcpy(h_out.data(), d_out, h_out.size() * sizeof(Fq), hipMemcpyDeviceToHost);
        const Fq rinv = Fq::inv(fq_pow2(rbits[which]));                       // Montgomery form of 2^-w
        size_t bad = 0;
        for (uint32_t t = 0; t < threads; t += 97) {
            Fq x = Fq::from_raw(h_in[2 * t]), y = Fq::from_raw(h_in[2 * t + 1]);      // the integers the limbs spell
            for (uint32_t i = 0; i < iters; i++) x = x * y * rinv;
            const Fq got = Fq::from_raw(reduce_raw(h_out[t]));
            if (!(got == x)) bad++;
        }
        printf("%-20s validation: %zu mismatches in %u sampled chains of %u products\n", names[which], bad, (threads + 96) / 97, iters);
    }
    // ---- timing: slope between two chain lengths (launch overhead and clock ramp cancel)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // at 8 waves per SIMD and then at the occupancies the MSM kernels really run at (their 130-160 VGPRs allow 3)
    for (int wps : {8, 4, 3, 2, 1})
    for (int which = 0; which < 4; which++) {
        grid = dim3((uint32_t)cus * wps);                                    // 4 waves a workgroup: wps waves per SIMD
        const uint32_t threads_now = grid.x * 256;
        auto timed = [&](uint32_t iters) {
            hipEventRecord(e0, 0);
            launch(which, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            return (double)ms;
        };
        const uint32_t n1 = 2000, n2 = 4000;
        double warm = 0;
        while (warm < (wps == 8 && which == 0 ? 50.0 : 5.0)) warm += timed(n2);
        double t1 = 1e30, t2 = 1e30;
        for (int rep = 0; rep < 3; rep++) { t1 = std::min(t1, timed(n1)); t2 = std::min(t2, timed(n2)); }
        const double per_s = (double)threads_now * (n2 - n1) / ((t2 - t1) * 1e-3);
        const double cyc = clk * 1e9 / (per_s / 64.0 / (cus * 4.0));          // cycles one SIMD spends per wave-product
        printf("%d waves/SIMD  %-20s %.3f ms @%u  %.3f ms @%u  -> %.1f G products/s, %.0f SIMD cycles per wave-product (= instructions at 4 cycles: %.0f)\n",
               wps, names[which], t1, n1, t2, n2, per_s * 1e-9, cyc, cyc / 4.0);
    }
    return 0;
}
