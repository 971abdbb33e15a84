// Internal interfaces shared by the kernel files and the prover (not part of the C ABI).
#pragma once
#include "ctx.hpp"
#include "curve.cuh"

namespace shp {

// per-caller scratch of one MSM pipeline (a prover instance owns one per stream)
struct MsmWorkspace {
    DevBuf cnt, start, entries, pf, pl, buckets, out;
    std::vector<G1X> h_out;
    std::vector<uint32_t> h_totals;
    uint64_t last_adds = 0, last_bytes = 0;
    void release() {
        cnt.release(); start.release(); entries.release(); pf.release(); pl.release(); buckets.release(); out.release();
    }
};

}  // namespace shp

struct shp_msm {
    shp_ctx* ctx = nullptr;
    size_t n = 0;
    uint32_t c = 0, W = 0, NB = 0;
    shp::G1Affine* d_table = nullptr;
    shp::MsmWorkspace ws;
    std::mutex mu;
};

struct shp_srs {
    shp_ctx* ctx = nullptr;
    uint32_t k = 0;
    size_t n = 0;
    shp::G1Affine* d_g = nullptr;
    shp::G1Affine* d_gl = nullptr;
    shp_msm* msm_g = nullptr;
    shp_msm* msm_gl = nullptr;
    uint8_t g2[128] = {0}, s_g2[128] = {0};
    bool has_g2 = false;
};

// `batch` MSMs over the plan's bases; synchronises `st` and normalises on the host
void msm_run_ws(shp_msm* m, shp::MsmWorkspace& ws, const shp::Fr* d_scalars, size_t n_scalars, size_t stride,
                uint32_t batch, shp::G1Affine* h_out, hipStream_t st);
void msm_run(shp_msm* m, const shp::Fr* d_scalars, size_t n_scalars, size_t stride, uint32_t batch,
             shp::G1Affine* h_out, hipStream_t st);
