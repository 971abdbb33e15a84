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
#include <memory>
#include <string.h>
#include <vector>

#include "poseidon2.hip.hpp"

namespace shp {

__device__ const Fr P2D_RC_FULL[8][8] = P2_RC_FULL_INIT;
__device__ const Fr P2D_RC_PART[48] = P2_RC_PART_INIT;
__device__ const Fr P2D_DIAG[8] = P2_DIAG_INIT;

static const Fr P2H_RC_FULL[8][8] = P2_RC_FULL_INIT;
static const Fr P2H_RC_PART[48] = P2_RC_PART_INIT;
static const Fr P2H_DIAG[8] = P2_DIAG_INIT;

void poseidon2_permute_host(Fr s[8]) { poseidon2_permute_impl(s, P2H_RC_FULL, P2H_RC_PART, P2H_DIAG); }

// (the rounds of poseidon2_permute_impl with a row stored behind every linear layer)
void poseidon2_trace_host(Fr* grid, size_t col_stride, const uint32_t cols[8], uint32_t r0) {
    Fr s[8];
    for (int i = 0; i < 8; i++) s[i] = grid[(size_t)cols[i] * col_stride + r0];
    uint32_t row = r0;
    auto put = [&] {
        row++;
        for (int i = 0; i < 8; i++) grid[(size_t)cols[i] * col_stride + row] = s[i];
    };
    p2_external(s);
    put();
    for (int r = 0; r < 8; r++) {
        if (r == 4) {
            for (int q = 0; q < 48; q++) {
                s[0] = p2_sbox(s[0] + P2H_RC_PART[q]);
                Fr sum = s[0];
                for (int i = 1; i < 8; i++) sum = sum + s[i];
                for (int i = 0; i < 8; i++) s[i] = P2H_DIAG[i] * s[i] + sum;
                put();
            }
        }
        for (int i = 0; i < 8; i++) s[i] = p2_sbox(s[i] + P2H_RC_FULL[r][i]);
        p2_external(s);
        put();
    }
}

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

// ---- arity-7 Poseidon2 Merkle tree (contracts/MerkleTree.sol) ------------------------------
// level l+1 node j = H(level l nodes 7j .. 7j+6), children beyond the populated prefix read as 0
// (the contract's unset `notes[]` slots), nodes above an empty subtree stay 0.
__global__ __launch_bounds__(128) void merkle_level_kernel(const Fr* cur, size_t n_cur, Fr* next, size_t n_next) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_next) return;
    Fr s[8];
#pragma unroll
    for (int c = 0; c < 7; c++) {
        size_t idx = 7 * j + c;
        s[c] = idx < n_cur ? p2_load(cur + idx) : Fr::zero();
    }
    Fr cap = Fr::zero();
    cap.l[2] = 7;
    s[7] = Fr::from_raw(cap);
    poseidon2_permute_impl(s, P2D_RC_FULL, P2D_RC_PART, P2D_DIAG);
    p2_store(next + j, s[0]);
}

// out[q] = | 7 siblings of level 0 | 7 of level 1 | ... | root |  for leaf index idx[q]
// (getMerklePath, MerkleTree.sol:95-118), canonical 32-byte little-endian elements
__global__ void merkle_paths_kernel(const Fr* const* levels, const size_t* counts, uint32_t height, const uint64_t* idx,
                                    size_t n_queries, Fr* out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t q = t / (7 * height + 1);
    uint32_t e = (uint32_t)(t % (7 * height + 1));
    if (q >= n_queries) return;
    Fr v;
    if (e == 7 * height) {
        v = p2_load(levels[height]);
    } else {
        uint32_t l = e / 7, c = e % 7;
        uint64_t node = idx[q];
        for (uint32_t i = 0; i < l; i++) node /= 7;
        uint64_t sib = (node / 7) * 7 + c;
        v = sib < counts[l] ? p2_load(levels[l] + sib) : Fr::zero();
    }
    p2_store(out + t, v.to_raw());
}

}  // namespace shp

using namespace shp;

struct shp_merkle {
    shp_ctx* ctx = nullptr;
    uint32_t height = 0;
    size_t n_leaves = 0;
    std::vector<Fr*> levels;          // device, levels[0] = leaves ... levels[height] = root
    std::vector<size_t> counts;
    Fr** d_levels = nullptr;
    size_t* d_counts = nullptr;
    ~shp_merkle() {
        for (Fr* p : levels) if (p) (void)hipFree(p);
        if (d_levels) (void)hipFree(d_levels);
        if (d_counts) (void)hipFree(d_counts);
    }
};

extern "C" {

int32_t shp_merkle_build(shp_ctx* ctx, const void* d_leaves, size_t n_leaves, uint32_t height, shp_merkle** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_leaves && out && n_leaves >= 1 && height >= 1 && height <= 32, SHP_ERR_INVALID_ARG,
                    "merkle_build: bad arguments");
        double cap = 1;
        for (uint32_t i = 0; i < height; i++) cap *= 7;
        SHP_REQUIRE((double)n_leaves <= cap, SHP_ERR_INVALID_ARG, "merkle_build: more leaves than 7^height");
        SHP_HIP(hipSetDevice(ctx->device));
        std::unique_ptr<shp_merkle> t(new shp_merkle);
        t->ctx = ctx;
        t->height = height;
        t->n_leaves = n_leaves;
        hipStream_t st = ctx->stream;
        size_t cnt = n_leaves;
        for (uint32_t l = 0; l <= height; l++) {
            Fr* p = nullptr;
            SHP_HIP(hipMalloc(&p, cnt * sizeof(Fr)));
            t->levels.push_back(p);
            t->counts.push_back(cnt);
            if (l == 0) {
                SHP_HIP(hipMemcpyAsync(p, d_leaves, cnt * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            } else {
                hipLaunchKernelGGL(merkle_level_kernel, dim3(ceil_div(cnt, 128)), dim3(128), 0, st, t->levels[l - 1],
                                   t->counts[l - 1], p, cnt);
            }
            cnt = (cnt + 6) / 7;
        }
        SHP_HIP(hipGetLastError());
        SHP_HIP(hipMalloc(&t->d_levels, (height + 1) * sizeof(Fr*)));
        SHP_HIP(hipMalloc(&t->d_counts, (height + 1) * sizeof(size_t)));
        SHP_HIP(hipMemcpyAsync(t->d_levels, t->levels.data(), (height + 1) * sizeof(Fr*), hipMemcpyHostToDevice, st));
        SHP_HIP(hipMemcpyAsync(t->d_counts, t->counts.data(), (height + 1) * sizeof(size_t), hipMemcpyHostToDevice, st));
        SHP_HIP(hipStreamSynchronize(st));
        *out = t.release();
    });
}

int32_t shp_merkle_destroy(shp_merkle* t) {
    return guarded([&] { delete t; });
}

int32_t shp_merkle_root(const shp_merkle* t, uint8_t root_le32[32]) {
    return guarded([&] {
        SHP_REQUIRE(t && root_le32, SHP_ERR_INVALID_ARG, "merkle_root: null argument");
        Fr m;
        SHP_HIP(hipMemcpy(&m, t->levels[t->height], sizeof(Fr), hipMemcpyDeviceToHost));
        Fr raw = m.to_raw();
        memcpy(root_le32, raw.l, 32);
    });
}

int32_t shp_merkle_paths(shp_merkle* t, const uint64_t* leaf_indices, size_t n_queries, uint8_t* paths_le32) {
    return guarded([&] {
        SHP_REQUIRE(t && leaf_indices && paths_le32, SHP_ERR_INVALID_ARG, "merkle_paths: null argument");
        if (!n_queries) return;
        for (size_t q = 0; q < n_queries; q++)
            SHP_REQUIRE(leaf_indices[q] < t->n_leaves, SHP_ERR_INVALID_ARG, "merkle_paths: leaf does not exist");   // LeafNotExisting
        SHP_HIP(hipSetDevice(t->ctx->device));
        const size_t per = (size_t)7 * t->height + 1, total = per * n_queries;
        uint64_t* d_idx = nullptr;
        Fr* d_out = nullptr;
        SHP_HIP(hipMalloc(&d_idx, n_queries * 8));
        hipError_t e = hipMalloc(&d_out, total * sizeof(Fr));
        if (e != hipSuccess) { (void)hipFree(d_idx); SHP_HIP(e); }
        hipStream_t st = t->ctx->stream;
        e = hipMemcpyAsync(d_idx, leaf_indices, n_queries * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(merkle_paths_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const Fr* const*)t->d_levels,
                               t->d_counts, t->height, d_idx, n_queries, d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(paths_le32, d_out, total * sizeof(Fr), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_idx);
        (void)hipFree(d_out);
        SHP_HIP(e);
    });
}

int32_t shp_poseidon2_t8(shp_ctx* ctx, const void* d_in, size_t n_hashes, uint32_t arity, void* d_out, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_in && d_out, SHP_ERR_INVALID_ARG, "poseidon2: null argument");
        SHP_REQUIRE(arity >= 1 && arity <= 7, SHP_ERR_INVALID_ARG, "poseidon2: arity must be in 1..7");
        if (!n_hashes) return;
        SHP_HIP(hipSetDevice(ctx->device));
        hipLaunchKernelGGL(poseidon2_hash_kernel, dim3(ceil_div(n_hashes, 128)), dim3(128), 0, ctx->pick(stream),
                           (const Fr*)d_in, n_hashes, arity, (Fr*)d_out);
        SHP_HIP(hipGetLastError());
    });
}

int32_t shp_poseidon2_t8_permute(shp_ctx* ctx, void* d_states, size_t n_states, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_states, SHP_ERR_INVALID_ARG, "poseidon2: null argument");
        if (!n_states) return;
        SHP_HIP(hipSetDevice(ctx->device));
        hipLaunchKernelGGL(poseidon2_permute_kernel, dim3(ceil_div(n_states, 128)), dim3(128), 0, ctx->pick(stream),
                           (Fr*)d_states, n_states);
        SHP_HIP(hipGetLastError());
    });
}

}  // extern "C"
