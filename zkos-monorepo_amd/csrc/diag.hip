// Diagnostics: the integer-ALU ceiling that actually bounds MSM and NTT on gfx950.
// Measures sustained Montgomery products per second (Fr, 8 x u32 CIOS) with every CU busy,
// 4 independent chains per lane so that the v_mad_u64_u32 pipeline is not latency bound.
#include "ctx.hpp"

namespace shp {
__global__ __launch_bounds__(256) void diag_mul_kernel(Fr* out, uint32_t iters) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr a = Fr::from_u64(gid + 3), b = Fr::from_u64(gid * 7 + 11), c = Fr::from_u64(gid ^ 0x5555), d = Fr::from_u64(gid + 99);
    Fr m = Fr::from_u64(0x9e3779b97f4a7c15ull);
    for (uint32_t i = 0; i < iters; i++) {
        a = a * m; b = b * a; c = c * m; d = d * c;
    }
    out[gid] = a + b + c + d;
}
}  // namespace shp
using namespace shp;

extern "C" int32_t shp_diag_fr_mul_rate(shp_ctx* ctx, uint32_t iters, double* muls_per_sec) {
    return guarded([&] {
        SHP_REQUIRE(ctx && muls_per_sec, SHP_ERR_INVALID_ARG, "diag: null argument");
        SHP_HIP(hipSetDevice(ctx->device));
        const uint32_t blocks = (uint32_t)ctx->num_cus * 8, threads = 256;
        Fr* d = nullptr;
        SHP_HIP(hipMalloc(&d, (size_t)blocks * threads * sizeof(Fr)));
        hipEvent_t e0, e1;
        SHP_HIP(hipEventCreate(&e0));
        SHP_HIP(hipEventCreate(&e1));
        hipLaunchKernelGGL(diag_mul_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, d, 16u);
        SHP_HIP(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(diag_mul_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, d, iters);
        SHP_HIP(hipEventRecord(e1, ctx->stream));
        SHP_HIP(hipEventSynchronize(e1));
        float ms = 0;
        SHP_HIP(hipEventElapsedTime(&ms, e0, e1));
        *muls_per_sec = 4.0 * iters * blocks * threads / (ms * 1e-3);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
    });
}
