// Poseidon2 permutation t = 8, alpha = 7, R_F = 8, R_P = 48 over BN254 Fr, batched: one lane per
// permutation, the 8-lane state (64 VGPRs) never leaves registers.
//
// Replaces shielder_circuits::poseidon::off_circuit::hash (called at
// /root/reference/crates/shielder_bindings/src/hash.rs:17-26, src/utils.rs:14-30); round
// structure and parameters follow /root/reference/poseidon2-solidity/generate_t8.py:474-588
// (initial external layer, 4 full rounds, 48 partial rounds with the round constant on lane 0
// only and the internal layer y_i = D_i x_i + sum(x), 4 full rounds, output lane 0; arity-N
// capacity element N * 2^64 in lane 7).
#include "ctx.hpp"
#include "poseidon2.cuh"

namespace shp {

__device__ const Fr P2D_RC_FULL[8][8] = P2_RC_FULL_INIT;
__device__ const Fr P2D_RC_PART[48] = P2_RC_PART_INIT;
__device__ const Fr P2D_DIAG[8] = P2_DIAG_INIT;

static const Fr P2H_RC_FULL[8][8] = P2_RC_FULL_INIT;
static const Fr P2H_RC_PART[48] = P2_RC_PART_INIT;
static const Fr P2H_DIAG[8] = P2_DIAG_INIT;

void poseidon2_permute_host(Fr s[8]) { poseidon2_permute_impl(s, P2H_RC_FULL, P2H_RC_PART, P2H_DIAG); }

Fr poseidon2_hash_host(const Fr* in, uint32_t arity) {
    Fr s[8];
    for (int i = 0; i < 8; i++) s[i] = Fr::zero();
    for (uint32_t i = 0; i < arity; i++) s[i] = in[i];
    Fr cap = Fr::zero();
    cap.l[2] = arity;          // arity * 2^64 as a raw integer
    s[7] = Fr::from_raw(cap);
    poseidon2_permute_host(s);
    return s[0];
}

__device__ __forceinline__ Fr p2_load(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void p2_store(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ __launch_bounds__(128) void poseidon2_hash_kernel(const Fr* in, size_t n, uint32_t arity, Fr* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = Fr::zero();
#pragma unroll
    for (int j = 0; j < 7; j++)
        if ((uint32_t)j < arity) s[j] = p2_load(in + i * arity + j);
    Fr cap = Fr::zero();
    cap.l[2] = arity;
    s[7] = Fr::from_raw(cap);
    poseidon2_permute_impl(s, P2D_RC_FULL, P2D_RC_PART, P2D_DIAG);
    p2_store(out + i, s[0]);
}

__global__ __launch_bounds__(128) void poseidon2_permute_kernel(Fr* states, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = p2_load(states + i * 8 + j);
    poseidon2_permute_impl(s, P2D_RC_FULL, P2D_RC_PART, P2D_DIAG);
#pragma unroll
    for (int j = 0; j < 8; j++) p2_store(states + i * 8 + j, s[j]);
}

}  // namespace shp

using namespace shp;

extern "C" {

int32_t shp_poseidon2_t8(shp_ctx* ctx, const void* d_in, size_t n_hashes, uint32_t arity, void* d_out, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_in && d_out, SHP_ERR_INVALID_ARG, "poseidon2: null argument");
        SHP_REQUIRE(arity >= 1 && arity <= 7, SHP_ERR_INVALID_ARG, "poseidon2: arity must be in 1..7");
        if (!n_hashes) return;
        hipLaunchKernelGGL(poseidon2_hash_kernel, dim3(ceil_div(n_hashes, 128)), dim3(128), 0, ctx->pick(stream),
                           (const Fr*)d_in, n_hashes, arity, (Fr*)d_out);
        SHP_HIP(hipGetLastError());
    });
}

int32_t shp_poseidon2_t8_permute(shp_ctx* ctx, void* d_states, size_t n_states, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_states, SHP_ERR_INVALID_ARG, "poseidon2: null argument");
        if (!n_states) return;
        hipLaunchKernelGGL(poseidon2_permute_kernel, dim3(ceil_div(n_states, 128)), dim3(128), 0, ctx->pick(stream),
                           (Fr*)d_states, n_states);
        SHP_HIP(hipGetLastError());
    });
}

}  // extern "C"
