// Internal interfaces shared by the kernel files and the prover (not part of the C ABI).
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "ctx.hpp"
#include "curve.hip.hpp"

namespace shp {

// Optional HIP-event timing of individual kernel launches on the stream they run on (bench.py's
// `roofline` object needs the dominant kernel's average launch duration measured live).
enum KernelId {
    KID_MSM_DIGITS = 0, KID_MSM_ACCUMULATE, KID_MSM_FIXUP, KID_MSM_REDUCE, KID_NTT, KID_PROG, KID_GRAND_PRODUCT,
    KID_RANK_SORT, KID_BATCH_INVERT, KID_MSM_DIRECT, KID_COUNT
};
static const char* const KERNEL_NAMES[KID_COUNT] = {"msm_digits", "msm_accumulate", "msm_fixup", "msm_reduce", "ntt_pass",
                                                    "prog_fold", "grand_product", "rank_sort", "batch_invert", "msm_direct"};
struct KernelTimers {
    bool enabled = false;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[KID_COUNT] = {0};
    uint64_t launches[KID_COUNT] = {0};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        SHP_HIP(hipEventCreate(&e));
        return e;
    }
    void begin(int id, hipStream_t st) {
        if (!enabled) return;
        Rec r{id, get(), get()};
        SHP_HIP(hipEventRecord(r.a, st));
        pending.push_back(r);
    }
    void end(hipStream_t st) {
        if (!enabled) return;
        SHP_HIP(hipEventRecord(pending.back().b, st));
    }
    // call after the stream has been synchronised
    void collect() {
        for (auto& r : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.id] += ms; launches[r.id]++; }
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        pending.clear();
    }
    void reset() {
        for (int i = 0; i < KID_COUNT; i++) { total_ms[i] = 0; launches[i] = 0; }
    }
    ~KernelTimers() {
        for (auto& r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

// rows of a scalar column that can be non-zero: [0, prefix) and [tail_start, tail_start + tail_count)
struct MsmRows { uint32_t prefix, tail_start, tail_count; };

// per-caller scratch of one MSM pipeline (a prover instance owns one per stream)
struct MsmWorkspace {
    KernelTimers* timers = nullptr;
    DevBuf cnt, start, entries, pf, pl, buckets, out, wg_hist;
    DevBuf binned, cursor;                 // two-level scatter of large MSMs: entries grouped by bucket bin, bucket write cursors
    PinnedBuf h_out, h_totals;             // G1X sums / per-MSM entry counts, read back after each run
    DevBuf done;                           // direct mode: per-MSM count of finished workgroups (zero between launches)
    size_t done_zeroed = 0;                // bytes of `done` known to be zero
    DevBuf segs;                           // direct mode: segment tables (16 cached slots of 4 + scratch)
    PinnedBuf h_segs;
    uint32_t seg_cached = 0, seg_cache_n[16] = {0};
    std::vector<G1Affine> h_aff;
    uint64_t last_adds = 0, last_bytes = 0;
    uint64_t min_waves = 0;     // direct mode: waves a launch should have before chains get longer (0 = default)
    // direct mode, a lone proof's lane: wait for the sums themselves (stamps in pinned memory) instead of the stream
    bool poll_results = false;
    PinnedBuf h_stamp;
    PinnedBuf h_trace;                     // SHP_MSM_TRACE=1: the kernel's clock readings of the launch in flight
    uint64_t trace_wgs = 0;
    uint32_t trace_msms = 0;
    uint32_t stamp_seq = 0;
    // the launch in flight between msm_launch_jobs and msm_finish_jobs
    struct PendJob { G1Affine* h_out; uint32_t msm_base, batch; };
    std::vector<PendJob> pend;
    uint32_t pend_total = 0, pend_seq = 0;
    bool pend_poll = false;
    bool pend_projective = false;          // its sums are lazy projective points (field29.hip.hpp: G1P29), not XYZZ
    void release() {
        cnt.release(); start.release(); entries.release(); pf.release(); pl.release(); buckets.release(); out.release();
        wg_hist.release(); binned.release(); cursor.release(); h_out.release(); h_totals.release(); segs.release(); h_segs.release(); seg_cached = 0;
        h_stamp.release();
        h_trace.release();
        done.release(); done_zeroed = 0;
    }
};

}  // namespace shp

struct shp_msm {
    shp_ctx* ctx = nullptr;
    size_t n = 0;
    uint32_t c = 0, W = 0, NB = 0;
    shp::G1Affine* d_table = nullptr;     // base rows 2^(c w) * P_i  [W][n]
    bool direct = false;                  // direct-table mode: d_mult holds every digit multiple
    shp::G1Affine* d_mult = nullptr;      // [W][n][NB]  d * 2^(c w) * P_i, d = 1..NB
    shp::Fr bias;                         // signed-digit bias of the direct mode
    shp::MsmWorkspace ws;
    std::mutex mu;
    std::atomic<int> refs{1};             // an SRS downsized to its own k shares its parent's plans (shp_msm_destroy drops one)
    ~shp_msm() {                          // also runs when plan construction throws half way
        if (d_table) (void)hipFree(d_table);
        if (d_mult) (void)hipFree(d_mult);
        ws.release();
    }
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
    uint8_t fingerprint[32] = {0};        // Keccak over k, g[0], g[1], g_lagrange[0], g2, s_g2: what a saved key is bound to
};

// one entry of a multi-job direct-table launch: `batch` MSMs over m's bases, scalar column b at scalars + b * stride
namespace shp {
constexpr uint32_t MSM_MAX_JOBS = 64;
struct MsmJob {
    shp_msm* m;
    const Fr* scalars;
    size_t stride;
    uint32_t batch;
    MsmRows rows;
    G1Affine* h_out;      // batch affine points
};
}  // namespace shp
void msm_run_jobs(const shp::MsmJob* jobs, uint32_t n_jobs, shp::MsmWorkspace& ws, hipStream_t st);
void msm_launch_jobs(const shp::MsmJob* jobs, uint32_t n_jobs, shp::MsmWorkspace& ws, hipStream_t st);
void msm_finish_jobs(shp::MsmWorkspace& ws, hipStream_t st);
namespace shp {
// the MSM plan of one of an SRS's two bases: the primary one (built first) may take the larger multiples table
void msm_create_for_srs(shp_ctx* ctx, const void* d_bases, size_t n, bool primary, shp_msm** out);
}  // namespace shp

namespace shp {
// A proving pool's lanes reach their commitment phases at different times; each phase alone is a launch of a few
// hundred waves that cannot fill 1024 SIMDs.  The combiner takes whatever requests are queued when its previous
// launch has finished (natural batching: no waiting window, a lone lane pays nothing) and serves them with ONE
// msm_direct launch + one final-sum launch + one read-back on its own stream.  Lanes hand over device pointers and
// an event recorded on their stream; they sleep until their points are back.
struct MsmCombiner {
    struct Req {
        const MsmJob* jobs;
        uint32_t n_jobs;
        hipEvent_t ready;
        bool done = false;
        int32_t status = SHP_OK;
        std::string err;
    };
    int device = 0;
    hipStream_t st = nullptr;
    MsmWorkspace ws;
    KernelTimers timers;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_req, cv_done;
    std::deque<Req*> q;
    bool stop = false;
    uint64_t launches = 0, requests = 0;
    // blocks until the jobs' points are in their h_out; everything queued on lane_stream so far has completed then
    void run(const MsmJob* jobs, uint32_t n_jobs, hipStream_t lane_stream, hipEvent_t ready);
    void loop();
};
}  // namespace shp
// `batch` MSMs over the plan's bases; synchronises `st` and normalises on the host
void msm_run_ws(shp_msm* m, shp::MsmWorkspace& ws, const shp::Fr* d_scalars, size_t n_scalars, size_t stride,
                uint32_t batch, shp::G1Affine* h_out, hipStream_t st);
// `rows` (optional): the only rows of m's scalar columns that can be non-zero, [0, prefix) and
// [tail_start, tail_start + tail_count); the direct-table path skips the rest without reading them
void msm_run_ws2(shp_msm* m, const shp::Fr* d_scalars, size_t stride, uint32_t batch, shp_msm* m2, const shp::Fr* d_scalars2,
                 size_t stride2, uint32_t batch2, size_t n_scalars, shp::MsmWorkspace& ws, shp::G1Affine* h_out, hipStream_t st,
                 const shp::MsmRows* rows = nullptr);
void msm_run(shp_msm* m, const shp::Fr* d_scalars, size_t n_scalars, size_t stride, uint32_t batch,
             shp::G1Affine* h_out, hipStream_t st);
