// The quotient's numerator in 29-bit lazy arithmetic: argument block and helpers of the generated kernels
// (quotient_kernels.inc, written by tools/gen_quotient_kernels.py).
//
// A key whose plan has a generated kernel keeps everything that kernel reads in Montgomery radix 2^261 -- the radix of
// Lazy29 -- so that a stored word is split into limbs and used, with no product to convert it: the twist table of the
// coset transform carries the factor 32 (every coset value of the proof's columns and of the key's fixed, sigma and
// l_0 / l_last / l_active columns inherits it, transforms being linear), and the points of the quotient domain, the
// plan's constants and the challenges are multiplied by 32 on the host.  The kernel's result is converted back
// (Lazy29::to_fp), so what follows the quotient phase sees the same values as with the separate radix-2^256 kernels.
#pragma once
#include "lazy29.hip.hpp"
#include "fr_mem.hip.hpp"

namespace shp {

using R29 = Lazy29<FrParams>;
using E29 = R29::El;
constexpr int QUOT29_THREADS = 64;     // most threads of a workgroup (one point of the quotient domain each)

struct Quot29Args {
    const Fr* cos;                 // [columns][ne]: coset values of the proof's columns (advice first), radix 2^261
    size_t ne;
    const Fr *fixed_cos, *sigma_cos, *l0, *l_last, *l_active;      // the key's tables, radix 2^261
    const Fr* dx_ext;              // [permutation column c][ne]: delta^c x at the points x of the quotient domain, radix 2^261
    const Fr* consts;              // the plan's constants, radix 2^261
    uint32_t i_pin, i_pz, i_lz, i_inst;      // first permuted-input, permutation-product, lookup-product, instance column of `cos`
    uint32_t mask;                 // n - 1: index = coset * n + row, rotations move the row
    int32_t rot_last;
    // challenges, radix 2^261; 1 / beta and beta^(columns of a full / of the last permutation set) come with them
    Fr y, theta, beta, gamma, beta_inv, beta_pow_full, beta_pow_last;
    Fr* h;                         // [ne] result, canonical, radix 2^256
};

// the generated kernel for a plan structure (hash: quotient_hash in prover.hip, tools/gen_quotient_kernels.py), or null
typedef void (*QuotientKernelFn)(Quot29Args);
QuotientKernelFn quotient_kernel_for(uint64_t hash);

__device__ __forceinline__ E29 ld29(const Fr* p) { return R29::split<0>(ld(p)); }
__device__ __forceinline__ E29 split29r(const Fr& v) { return R29::split<0>(v); }
__device__ __forceinline__ E29 add29(const E29& a, const E29& b) {
    E29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// K p - a, normalised (kp = kp_spread(K, S) as for Lazy29::sub)
__device__ __forceinline__ E29 neg29(const E29& a, const l29::L9& kp) {
    E29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = kp.l[i] - a.l[i];
    R29::normalise(r);
    return r;
}

}  // namespace shp
