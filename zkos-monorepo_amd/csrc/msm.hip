// BN254 G1 Pippenger MSM for gfx950 (fixed bases, batched scalar vectors).
//
// Replaces halo2curves 0.6.1 `best_multiexp` behind Params::commit / commit_lagrange
// (every commitment of keygen and create_proof; exercised by the reference at
// /root/reference/crates/powers-of-tau/lib.rs:247-264).  The result is a group element, so
// after normalisation it is bit-identical to the reference's bucket method.
//
// MI355X-first design (not a translation of the CPU algorithm):
//   * The SRS is fixed, HBM is 288 GB: the window tables 2^(c*w) * P_i are precomputed once
//     (n * W * 64 B), so ALL windows of one MSM share ONE bucket set -- no per-window
//     reduction and no doublings on the critical path.
//   * Signed c-bit digits -> 2^(c-1) buckets; entries (table index | sign) are grouped by
//     bucket with a counting sort (histogram / scan / scatter).
//   * Bucket accumulation is load-balanced by construction: each thread sums a fixed-size
//     CHUNK of the sorted entry list (exactly one mixed XYZZ addition per entry, 8M + 2S in
//     registers), whatever the bucket sizes -- witness columns full of 0/1 make single
//     buckets thousands of entries long.  Runs cut by a chunk boundary are emitted as
//     partials and merged per bucket, wave-cooperatively (lanes stride over the chunks, then a
//     __shfl_xor butterfly of XYZZ additions) when a bucket spans many chunks.
//   * sum_b b * B_b: per-thread running sums over g buckets, an LDS suffix scan and a tree.
//   * The handful of final XYZZ sums are normalised on the host (needed there anyway for the
//     Keccak transcript).
#include <string.h>
#include <algorithm>
#include <mutex>
#include <stdlib.h>

#include "engine.hpp"
#include "field29.hip.hpp"

namespace shp {

constexpr uint32_t BUCKET_NONE = 0xffffffffu;
constexpr int MSM_THREADS = 256;

struct alignas(16) Partial {
    G1X sum;
    uint32_t bucket;
    uint32_t pad[3];
};

__device__ __forceinline__ Fr load_fr(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

// Digits are taken from the low end of a value that is shifted right window by window: every limb index is a
// compile-time constant.  (A dynamic index -- even written as a select chain, which LLVM folds back into an indexed
// load -- puts the limbs in scratch memory: a private segment for the kernel and VMEM round trips on the digit path.)
__device__ __forceinline__ void raw_shr_bits(Fr& r, uint32_t t) {     // 1 <= t <= 31
#pragma unroll
    for (int i = 0; i < 7; i++) r.l[i] = (r.l[i] >> t) | (r.l[i + 1] << (32 - t));
    r.l[7] >>= t;
}
__device__ __forceinline__ void raw_shr_any(Fr& r, uint32_t s) {      // 0 <= s; >= 256 clears
    if (s >= 256) { r = Fr::zero(); return; }
    if (s & 128) {
#pragma unroll
        for (int i = 0; i < 4; i++) { r.l[i] = r.l[i + 4]; r.l[i + 4] = 0; }
    }
    if (s & 64) {
#pragma unroll
        for (int i = 0; i < 6; i++) r.l[i] = r.l[i + 2];
        r.l[6] = r.l[7] = 0;
    }
    if (s & 32) {
#pragma unroll
        for (int i = 0; i < 7; i++) r.l[i] = r.l[i + 1];
        r.l[7] = 0;
    }
    if (s & 31) raw_shr_bits(r, s & 31);
}
// the low c bits (c <= 16), then shift them out
__device__ __forceinline__ uint32_t raw_pop_bits(Fr& r, uint32_t c) {
    const uint32_t d = r.l[0] & ((1u << c) - 1);
    raw_shr_bits(r, c);
    return d;
}

// ---------------------------------------------------------------- table precompute
__global__ void msm_table_kernel(const G1Affine* bases, G1Affine* table, size_t n, uint32_t c, uint32_t W) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = bases[i];
    table[i] = p;
    G1X x = G1X::from_affine(p);
    for (uint32_t w = 1; w < W; w++) {
        for (uint32_t k = 0; k < c; k++) x = g1_dbl(x);
        table[(size_t)w * n + i] = g1_to_affine(x);
    }
}

// ---- counting sort by bucket without global atomics -----------------------------------------
// (2^24 scalars x 16 windows = 268 M increments on 32768 counters: as global atomics that was 43 %
// of the whole MSM.)  Each workgroup owns a contiguous range of scalars and keeps the bucket
// histogram in LDS (NB + 1 words: 128 KB at c = 16, one workgroup per CU):
//   msm_hist_kernel<0>     LDS histogram of my range                 -> wg_hist[m][wg][b]
//   msm_hist_prefix_kernel per bucket: exclusive prefix over the workgroups (in place), total -> cnt
//   msm_scan_kernel        bucket starts from the totals
//   msm_hist_kernel<1>     LDS cursors start[b] + wg_hist[m][wg][b]; digits again, entries written
//   (large inputs: msm_hist_kernel<2> + msm_bin_scatter_kernel instead of <1>, see below)
// The order of the entries inside a bucket depends on the LDS atomics' arrival order; the bucket
// SUM does not (exact group arithmetic), so results stay bit-identical.
constexpr int DIG_THREADS = 1024;

// Large MSMs scatter in TWO levels.  Written straight to its bucket, an entry is a 4-byte store to one of 2^15 cursors
// per workgroup: a write frontier of megabytes per workgroup, so nearly every store reaches HBM as a partial line
// (6.8 ms of a 30 ms MSM at 2^24).  Instead the entries are first grouped by BIN = 256 consecutive buckets
// (msm_hist_kernel<2>: 128 cursors per workgroup, each filling whole lines; 8-byte entries that carry the bucket), then
// msm_bin_scatter_kernel takes slices of that array, counts them per bucket in LDS, claims a range of each bucket's
// region with one global atomic per (slice, bucket) and writes runs of ~32 entries.
constexpr uint32_t BIN_LOG = 8;
constexpr uint32_t SLICE = 8192;             // binned entries per workgroup of the second level (8 per thread)

// MODE 0: count; 1: scatter straight to the buckets; 2: scatter to the bins
template <int MODE>
__global__ __launch_bounds__(DIG_THREADS) void msm_hist_kernel(const Fr* scalars, size_t n_scalars, size_t stride, size_t n_table,
                                                               uint32_t c, uint32_t W, uint32_t NB, uint32_t* wg_hist,
                                                               const uint32_t* start, uint32_t* entries, size_t entries_stride,
                                                               uint64_t* binned) {
    extern __shared__ uint32_t lds_cnt[];                       // NB + 1 words (MODE 2: one per bin)
    const uint32_t G = gridDim.x, wg = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
    uint32_t* row = wg_hist + ((size_t)m * G + wg) * (NB + 1);
    if (MODE == 1) {
        const uint32_t* st = start + (size_t)m * (NB + 2);
        for (uint32_t b = tid; b <= NB; b += DIG_THREADS) lds_cnt[b] = st[b] + row[b];
    } else if (MODE == 2) {
        // cursor of bin k = start of its first bucket (256 k + 1) + what the workgroups before this one put into the bin
        const uint32_t* st = start + (size_t)m * (NB + 2);
        const uint32_t bins = NB >> BIN_LOG;
        for (uint32_t k = tid; k < bins; k += DIG_THREADS) lds_cnt[k] = st[(k << BIN_LOG) + 1];
        __syncthreads();
        for (uint32_t b = tid; b < NB; b += DIG_THREADS) {
            const uint32_t before = row[b + 1];
            if (before) atomicAdd(&lds_cnt[b >> BIN_LOG], before);
        }
    } else {
        for (uint32_t b = tid; b <= NB; b += DIG_THREADS) lds_cnt[b] = 0;
    }
    __syncthreads();
    const size_t per = (n_scalars + G - 1) / G;
    const size_t i0 = (size_t)wg * per, i1 = i0 + per < n_scalars ? i0 + per : n_scalars;
    for (size_t i = i0 + tid; i < i1; i += DIG_THREADS) {
        Fr raw = load_fr(scalars + (size_t)m * stride + i).to_raw();
        if (raw.is_zero()) continue;
        uint32_t carry = 0;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t d = raw_pop_bits(raw, c) + carry;
            uint32_t sign = 0;
            carry = 0;
            if (d > NB) {
                d = (1u << c) - d;
                sign = 1;
                carry = 1;
            }
            if (d == 0) continue;
            const uint32_t ent = (uint32_t)((size_t)w * n_table + i) | (sign << 31);
            if (MODE == 2) {
                const uint32_t pos = atomicAdd(&lds_cnt[(d - 1) >> BIN_LOG], 1u);
                binned[(size_t)m * entries_stride + pos] = (uint64_t)ent | ((uint64_t)d << 32);
            } else {
                const uint32_t pos = atomicAdd(&lds_cnt[d], 1u);
                if (MODE == 1) entries[(size_t)m * entries_stride + pos] = ent;
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        for (uint32_t b = tid; b <= NB; b += DIG_THREADS) row[b] = lds_cnt[b];
    }
}

// second level: slice blockIdx.x of the binned array (sorted by bin, so the buckets of a slice lie between those of its
// first and its last entry's bins) -> the bucket regions of `entries`.  cursor[d] starts at start[d].
__global__ __launch_bounds__(DIG_THREADS) void msm_bin_scatter_kernel(const uint64_t* binned, size_t entries_stride,
                                                                      const uint32_t* start, uint32_t NB, uint32_t* cursor,
                                                                      uint32_t* entries) {
    extern __shared__ uint32_t lds_cnt[];                       // up to NB words: buckets [lo, hi) of this slice
    const uint32_t m = blockIdx.y, tid = threadIdx.x;
    const uint32_t total = start[(size_t)m * (NB + 2) + NB + 1];
    const size_t s0 = (size_t)blockIdx.x * SLICE;
    if (s0 >= total) return;
    const uint32_t cnt = (uint32_t)(total - s0 < SLICE ? total - s0 : SLICE);
    const uint64_t* src = binned + (size_t)m * entries_stride + s0;
    uint32_t* cur = cursor + (size_t)m * (NB + 2);
    uint32_t* dst = entries + (size_t)m * entries_stride;
    const uint32_t d_first = (uint32_t)(src[0] >> 32), d_last = (uint32_t)(src[cnt - 1] >> 32);
    const uint32_t lo = (((d_first - 1) >> BIN_LOG) << BIN_LOG) + 1;           // first bucket of the first bin
    const uint32_t hi = ((((d_last - 1) >> BIN_LOG) + 1) << BIN_LOG) + 1;      // one past the last bucket of the last bin
    for (uint32_t b = tid; b < hi - lo; b += DIG_THREADS) lds_cnt[b] = 0;
    __syncthreads();
    constexpr uint32_t PER = SLICE / DIG_THREADS;
    uint64_t e[PER];
    uint32_t rank[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t idx = k * DIG_THREADS + tid;
        e[k] = idx < cnt ? src[idx] : 0;
        if (idx < cnt) rank[k] = atomicAdd(&lds_cnt[(uint32_t)(e[k] >> 32) - lo], 1u);
    }
    __syncthreads();
    for (uint32_t b = tid; b < hi - lo; b += DIG_THREADS) {
        const uint32_t c = lds_cnt[b];
        if (c) lds_cnt[b] = atomicAdd(&cur[lo + b], c);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t idx = k * DIG_THREADS + tid;
        if (idx < cnt) dst[lds_cnt[(uint32_t)(e[k] >> 32) - lo] + rank[k]] = (uint32_t)e[k];
    }
}

// thread <-> bucket: exclusive prefix of wg_hist[m][.][b] over the workgroups, total into cnt[m][b]
__global__ void msm_hist_prefix_kernel(uint32_t* wg_hist, uint32_t G, uint32_t NB, uint32_t* cnt) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (b > NB) return;
    uint32_t* col = wg_hist + (size_t)m * G * (NB + 1) + b;
    uint32_t run = 0;
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t v = col[(size_t)g * (NB + 1)];
        col[(size_t)g * (NB + 1)] = run;
        run += v;
    }
    cnt[(size_t)m * (NB + 2) + b] = run;
}

// exclusive scan of cnt[m][1..NB] -> start[m][1..NB+1]; one workgroup per batch element
__global__ __launch_bounds__(MSM_THREADS) void msm_scan_kernel(const uint32_t* cnt, uint32_t* start, uint32_t NB) {
    __shared__ uint32_t sh[MSM_THREADS];
    uint32_t m = blockIdx.x, t = threadIdx.x;
    const uint32_t* c = cnt + (size_t)m * (NB + 2);
    uint32_t* s = start + (size_t)m * (NB + 2);
    uint32_t per = (NB + MSM_THREADS - 1) / MSM_THREADS;
    uint32_t lo = 1 + t * per, hi = lo + per;
    if (hi > NB + 1) hi = NB + 1;
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += c[b];
    sh[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < MSM_THREADS; off <<= 1) {
        uint32_t v = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    uint32_t run = sh[t] - sum;
    for (uint32_t b = lo; b < hi; b++) {
        s[b] = run;
        run += c[b];
    }
    if (t == MSM_THREADS - 1) s[NB + 1] = sh[t];
    if (t == 0) s[0] = 0;
}

__device__ __forceinline__ G1Affine load_affine(const G1Affine* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    G1Affine r;
    r.x.l[0] = a.x; r.x.l[1] = a.y; r.x.l[2] = a.z; r.x.l[3] = a.w;
    r.x.l[4] = b.x; r.x.l[5] = b.y; r.x.l[6] = b.z; r.x.l[7] = b.w;
    r.y.l[0] = c.x; r.y.l[1] = c.y; r.y.l[2] = c.z; r.y.l[3] = c.w;
    r.y.l[4] = d.x; r.y.l[5] = d.y; r.y.l[6] = d.z; r.y.l[7] = d.w;
    return r;
}

// ---------------------------------------------------------------- chunked bucket accumulation
__global__ __launch_bounds__(MSM_THREADS) void msm_accumulate_kernel(const G1Affine* __restrict__ table,
                                                                     const uint32_t* __restrict__ entries,
                                                                     size_t entries_stride, const uint32_t* start,
                                                                     uint32_t NB, uint32_t T, uint32_t chunks_per_m,
                                                                     G1X* buckets, Partial* part_first,
                                                                     Partial* part_last) {
    uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m = blockIdx.y;
    if (chunk >= chunks_per_m) return;
    const uint32_t* s = start + (size_t)m * (NB + 2);
    const uint32_t total = s[NB + 1];
    const uint32_t begin = chunk * T;
    if (begin >= total) return;
    const uint32_t end = begin + T < total ? begin + T : total;
    const uint32_t* ent = entries + (size_t)m * entries_stride;
    G1X* bk = buckets + (size_t)m * (NB + 1);
    Partial* pf = part_first + (size_t)m * chunks_per_m + chunk;
    Partial* pl = part_last + (size_t)m * chunks_per_m + chunk;

    // last bucket b in [1, NB] with s[b] <= begin
    uint32_t lo = 1, hi = NB;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (s[mid] <= begin) lo = mid;
        else hi = mid - 1;
    }
    uint32_t cur = lo;
    uint32_t boundary = s[cur + 1];
    // the running sum lives in the 29-bit-limb form (field29.hip.hpp); it is converted when a run ends
    G1X29 acc;
    acc.inf = true;
    bool first_run = true;
    pl->bucket = BUCKET_NONE;
    // the gather of entry pos + 1 is in flight while entry pos is added: a random 64 B read from a table of many GB
    // costs microseconds, and three waves per SIMD do not hide it behind a 2400-instruction addition on their own
    uint32_t e_next = ent[begin];
    G1Affine p_next = load_affine(table + (e_next & 0x7fffffffu));
    for (uint32_t pos = begin; pos < end; pos++) {
        if (pos >= boundary) {
            if (first_run) { pf->sum = to_g1x(acc); pf->bucket = cur; first_run = false; }
            else bk[cur] = to_g1x(acc);
            acc.inf = true;
            do { cur++; boundary = s[cur + 1]; } while (pos >= boundary);
        }
        const uint32_t e = e_next;
        const G1Affine p = p_next;
        if (pos + 1 < end) {
            e_next = ent[pos + 1];
            p_next = load_affine(table + (e_next & 0x7fffffffu));
        }
        if (p.is_inf()) continue;
        Fq y = (e >> 31) ? Fq::neg(p.y) : p.y;
        g1_madd29(acc, p.x, y);
    }
    if (first_run) { pf->sum = to_g1x(acc); pf->bucket = cur; }
    else { pl->sum = to_g1x(acc); pl->bucket = cur; }
}

__device__ __forceinline__ G1X shfl_xor_point(const G1X& p, int mask) {
    G1X r;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 32; i++) dst[i] = __shfl_xor(src[i], mask);
    return r;
}

// merge the partial sums of runs cut by chunk boundaries; one lane per bucket, heavy buckets
// (spanning >= 64 chunks) by the whole wave
__global__ __launch_bounds__(MSM_THREADS) void msm_fixup_kernel(const uint32_t* start, uint32_t NB, uint32_t T,
                                                                uint32_t chunks_per_m, G1X* buckets,
                                                                const Partial* part_first, const Partial* part_last) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t m = blockIdx.y;
    const uint32_t* s = start + (size_t)m * (NB + 2);
    const uint32_t total = s[NB + 1];
    const Partial* pf = part_first + (size_t)m * chunks_per_m;
    const Partial* pl = part_last + (size_t)m * chunks_per_m;
    G1X* bk = buckets + (size_t)m * (NB + 1);

    bool active = false, heavy = false;
    uint32_t c0 = 0, c1 = 0;
    if (b <= NB) {
        uint32_t sb = s[b], eb = s[b + 1];
        if (eb > sb) {
            c0 = sb / T;
            c1 = (eb - 1) / T;
            uint32_t cend = (c0 + 1) * T < total ? (c0 + 1) * T : total;
            bool middle = (c0 == c1) && (sb != c0 * T) && (eb != cend);
            active = !middle;
            heavy = active && (c1 - c0 >= 64);
        }
    }
    if (active && !heavy) {
        G1X acc = G1X::inf();
        for (uint32_t ch = c0; ch <= c1; ch++) {
            if (pf[ch].bucket == b) acc = g1_add(acc, pf[ch].sum);
            if (pl[ch].bucket == b) acc = g1_add(acc, pl[ch].sum);
        }
        bk[b] = acc;
    }
    unsigned long long hm = __ballot(heavy);
    const int lane = threadIdx.x & 63;
    while (hm) {
        int src = __ffsll((long long)hm) - 1;
        hm &= hm - 1;
        uint32_t hb = __shfl(b, src), h0 = __shfl(c0, src), h1 = __shfl(c1, src);
        G1X acc = G1X::inf();
        for (uint32_t ch = h0 + lane; ch <= h1; ch += 64) {
            if (pf[ch].bucket == hb) acc = g1_add(acc, pf[ch].sum);
            if (pl[ch].bucket == hb) acc = g1_add(acc, pl[ch].sum);
        }
        for (int mask = 32; mask >= 1; mask >>= 1) acc = g1_add(acc, shfl_xor_point(acc, mask));
        if (lane == 0) bk[hb] = acc;
    }
}

// ---------------------------------------------------------------- sum_b b * B_b
// The buckets are cut into slices of Q = 256 g consecutive buckets, one workgroup each: slice w
// returns L_w = sum_l l * B_(wQ+l) (running sums per thread, LDS suffix scan, tree) and its plain
// sum R_w; the combine kernel adds sum_w L_w + Q * sum_w w R_w (Q a power of two: doublings).
// A single workgroup over all 2^15 buckets of a c = 16 plan spent 2.5 ms here whatever the MSM size.
__global__ __launch_bounds__(MSM_THREADS) void msm_reduce_kernel(const G1X* buckets, uint32_t NB, uint32_t Q, G1X* outL, G1X* outR) {
    __shared__ G1X sh[2][MSM_THREADS];
    const uint32_t t = threadIdx.x, w = blockIdx.x, m = blockIdx.y;
    const G1X* bk = buckets + (size_t)m * (NB + 1) + (size_t)w * Q;
    const uint32_t A = Q < MSM_THREADS ? Q : MSM_THREADS;
    const uint32_t g = Q / A;
    G1X run = G1X::inf(), V = G1X::inf();
    if (t < A) {
        for (int j = (int)g - 1; j >= 0; j--) {
            run = g1_add(run, bk[t * g + j + 1]);
            V = g1_add(V, run);
        }
    }
    // inclusive suffix sums of the per-thread totals
    int cur = 0;
    sh[0][t] = run;
    __syncthreads();
    for (uint32_t off = 1; off < A; off <<= 1) {
        G1X v = sh[cur][t];
        if (t + off < A) v = g1_add(v, sh[cur][t + off]);
        sh[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    if (t == 0) outR[(size_t)m * gridDim.x + w] = sh[cur][0];
    G1X Y = V;
    if (t >= 1 && t < A) {
        G1X S = sh[cur][t];
        for (uint32_t k = g; k > 1; k >>= 1) S = g1_dbl(S);
        Y = g1_add(Y, S);
    }
    __syncthreads();
    sh[0][t] = Y;
    __syncthreads();
    for (uint32_t off = MSM_THREADS / 2; off >= 1; off >>= 1) {
        if (t < off) sh[0][t] = g1_add(sh[0][t], sh[0][t + off]);
        __syncthreads();
    }
    if (t == 0) outL[(size_t)m * gridDim.x + w] = sh[0][0];
}

// one wave per MSM: sum_w L_w + Q * sum_w w * R_w over nW <= 64 slices
__global__ __launch_bounds__(64) void msm_reduce_combine_kernel(const G1X* L, const G1X* R, uint32_t nW, uint32_t Q, G1X* out) {
    const uint32_t lane = threadIdx.x, m = blockIdx.x;
    G1X l = G1X::inf(), s = G1X::inf();
    if (lane < nW) {
        l = L[(size_t)m * nW + lane];
        s = R[(size_t)m * nW + lane];
    }
    // inclusive suffix sums S_w = sum_{w' >= w} R_w' ; sum_w w R_w = sum_{w >= 1} S_w
    for (int off = 1; off < 64; off <<= 1) {
        G1X o;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&s);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int i = 0; i < 32; i++) dst[i] = __shfl_down(src[i], off);
        if ((int)lane + off < 64 && (uint32_t)off < nW) s = g1_add(s, o);
    }
    if (lane == 0) s = G1X::inf();
    for (int mask = 32; mask >= 1; mask >>= 1) {
        if ((uint32_t)mask < nW || mask == 1) {
            s = g1_add(s, shfl_xor_point(s, mask));
            l = g1_add(l, shfl_xor_point(l, mask));
        }
    }
    for (uint32_t k = Q; k > 1; k >>= 1) s = g1_dbl(s);
    if (lane == 0) out[m] = g1_add(l, s);
}

// ================================================================== direct-table mode
// For the prover's SRS (n <= 2^16 fixed bases, 288 GB of HBM) the plan stores EVERY digit multiple
//   mult[(w * n + i) * NB + (d - 1)] = d * 2^(c w) * P_i,   d = 1 .. NB = 2^(c-1)
// (n * W * NB * 64 B: 23.6 GB at n = 8192, c = 12), so an MSM has no buckets at all: it is the
// plain sum of n * W gathered affine points.  No sort, no atomics, no bucket reduction; the
// critical path is T mixed additions per lane plus a log-depth tree of XYZZ additions, and the
// work is the minimum any window method can do (one mixed addition per non-zero digit).
//
// Signed digits without carry chains: with K = sum_{w < W-1} 2^(c(w+1)-1) added to the scalar once,
// digit_w = ((s + K) >> cw & (2^c - 1)) - 2^(c-1) for w < W-1 and the top window is taken as is.

// base rows -> all NB multiples, one wave per row, into XYZZ scratch + running zzz products
__global__ __launch_bounds__(64) void msm_multiples_kernel(const G1Affine* __restrict__ rows, size_t row0, uint32_t n_rows,
                                                           uint32_t NB, G1X* __restrict__ xyzz, Fq* __restrict__ pre,
                                                           G1Affine* __restrict__ mult) {
    const uint32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t K = NB / 64;
    const G1Affine B = rows[row0 + r];
    G1Affine* out = mult + (row0 + r) * (size_t)NB;
    if (B.is_inf()) {
        G1Affine z; z.x = Fq::zero(); z.y = Fq::zero();
        for (uint32_t t = 0; t < K; t++) out[(size_t)lane * K + t] = z;
        return;
    }
    G1X* xs = xyzz + (size_t)r * NB + (size_t)lane * K;
    Fq* ps = pre + (size_t)r * NB + (size_t)lane * K;
    // KB = K * B (K is a power of two), S = lane * KB + B
    G1X KB = G1X::from_affine(B);
    for (uint32_t k = K; k > 1; k >>= 1) KB = g1_dbl(KB);
    G1X S = G1X::inf();
    for (int bit = 5; bit >= 0; bit--) {
        S = g1_dbl(S);
        if ((lane >> bit) & 1u) S = g1_add(S, KB);
    }
    S = g1_madd(S, B.x, B.y);
    Fq acc = Fq::one();
    for (uint32_t t = 0; t < K; t++) {
        xs[t] = S;
        ps[t] = acc;
        acc = acc * S.zzz;               // d * B is never the identity for d < r
        S = g1_madd(S, B.x, B.y);
    }
    // wave-wide Montgomery trick: one inversion for the 64 * K points of the row
    Fq incl = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Fq o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = __shfl_up(incl.l[i], off);
        if ((int)lane >= off) incl = incl * o;
    }
    Fq suf = acc;                          // inclusive suffix products
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Fq o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(suf.l[i], off);
        if ((int)lane + off < 64) suf = suf * o;
    }
    Fq total;
#pragma unroll
    for (int i = 0; i < 8; i++) total.l[i] = __shfl(incl.l[i], 63);
    const Fq inv_total = Fq::inv(total);
    Fq pre_excl, suf_excl;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        pre_excl.l[i] = __shfl_up(incl.l[i], 1);
        suf_excl.l[i] = __shfl_down(suf.l[i], 1);
    }
    if (lane == 0) pre_excl = Fq::one();
    if (lane == 63) suf_excl = Fq::one();
    Fq inv = inv_total * pre_excl * suf_excl;     // 1 / (product of my K zzz)
    for (int t = (int)K - 1; t >= 0; t--) {
        const G1X P = xs[t];
        const Fq zi = inv * ps[t];                // 1 / zzz_t
        inv = inv * P.zzz;
        const Fq zz_inv = Fq::sqr(zi * P.zz);     // (zz / zzz)^2 = 1 / zz
        G1Affine a;
        a.x = P.x * zz_inv;
        a.y = P.y * zi;
        out[(size_t)lane * K + t] = a;
    }
}

// One launch serves any number of jobs (segments): `batch` MSMs each, over its own multiples table and scalar
// columns; the tables hold the same number of bases but may differ in window width (the two bases of an SRS do).
// The segment table lives in device memory; workgroup -> segment by
// binary search over wg_base (a by-value array indexed dynamically would be copied to scratch).
struct DirectSeg {
    const G1Affine* mult;
    const Fr* scalars;
    uint64_t stride;
    // rows a segment actually visits: [0, prefix) and [tail_start, tail_start + tail_count)
    // (a witness column is zero between its assigned rows and its blinding rows)
    uint32_t prefix, tail_start, tail_count;
    uint32_t gx;          // 256-row blocks per MSM
    uint32_t wg_base;     // first blockIdx.x of the segment
    uint32_t msm_base;    // index of its first MSM in the launch
    uint64_t part_base;   // its partials: [batch][gy][gx] from here
    // its table: window width, windows, entries per (window, base), windows per lane in this launch (a window group
    // beyond the table's last window adds nothing), the bias that makes the digits signed
    uint32_t c, W, NB, T;
    Fr bias;
};
struct DirectArgs {
    const DirectSeg* segs;
    uint32_t n_segs;
    size_t n_table;
    G1X* partials;
    uint32_t* done;       // per MSM of the launch: workgroups that have stored their partial sum
    G1X* out;             // per MSM: the sum (may be pinned host memory)
    uint32_t* stamp;      // per MSM, pinned host memory (null: none): set to `seq` once the sum is stored (wait_stamps)
    uint32_t seq;
    // diagnostics (SHP_MSM_TRACE=1, else null): per workgroup the constant-rate clock at its start, after its lanes' sums,
    // after its tree; then, behind those, four words per MSM: the clock when the last workgroup starts the merge, has the
    // sum (twice) and has stored it
    uint64_t* trace;
};

// Sum of one XYZZ point per thread over a 256-thread workgroup; the result is valid in thread 0.
//
// The points live in LDS in [coordinate][limb][slot] order (conflict free).  Level 256 -> 128 is one
// ordinary addition per lane on two waves.  From 64 additions down, FOUR lanes cooperate on each
// addition: the 14 field products of add-2008-s have a dependency depth of only four
//   R1  u1 = x1 zz2 | s1 = y1 zzz2 | u2 = x2 zz1 | s2 = y2 zzz1
//   R2  pp = (u2-u1)^2 | rr2 = (s2-s1)^2 | zz12 = zz1 zz2 | zzz12 = zzz1 zzz2
//   R3  qq = u1 pp | - | zz3 = zz12 pp | ppp = (u2-u1) pp
//   R4  x3 = rr2-ppp-2qq, t2 = rr (qq-x3) | t1 = s1 ppp | - | zzz3 = zzz12 ppp          y3 = t2 - t1
// so a level costs four product rounds instead of fourteen, which is what bounds the latency of a
// small MSM (a lone wave issues one instruction per ~8 cycles whatever its ILP).  Operands and
// intermediates travel between the four lanes through LDS.  Identity operands are selected
// around; the doubling / opposite-point cases (pp == 0) are rare and redone by lane 0 of the quad
// with the generic g1_add.
constexpr uint32_t COOP_SLOTS = 128;                          // points alive after the first level
constexpr uint32_t COOP_POINT_WORDS = 32 * COOP_SLOTS;        // [4 coords][8 limbs][128 slots]
constexpr uint32_t COOP_SCR_WORDS = 9 * 8 * 64;               // [9 values][8 limbs][64 quads]
constexpr uint32_t COOP_LDS_WORDS = COOP_POINT_WORDS + COOP_SCR_WORDS + 64;

// `count` (wave-uniform): threads >= count are known to hold the identity, so the levels that would only
// add identities are skipped.
__device__ __forceinline__ G1X block_sum_256(G1X acc, uint32_t* sh /* [COOP_LDS_WORDS] */, uint32_t count = MSM_THREADS) {
    const uint32_t t = threadIdx.x;
    uint32_t* scr = sh + COOP_POINT_WORDS;
    uint32_t* flg = scr + COOP_SCR_WORDS;                       // per quad: bit0 inf1, bit1 inf2, bit2 pp == 0
    auto ldc = [&](uint32_t coord, uint32_t slot) {
        Fq v;
#pragma unroll
        for (int i = 0; i < 8; i++) v.l[i] = sh[(coord * 8 + i) * COOP_SLOTS + slot];
        return v;
    };
    auto stc = [&](uint32_t coord, uint32_t slot, const Fq& v) {
#pragma unroll
        for (int i = 0; i < 8; i++) sh[(coord * 8 + i) * COOP_SLOTS + slot] = v.l[i];
    };
    auto lds = [&](uint32_t val, uint32_t quad) {
        Fq v;
#pragma unroll
        for (int i = 0; i < 8; i++) v.l[i] = scr[(val * 8 + i) * 64 + quad];
        return v;
    };
    auto sts = [&](uint32_t val, uint32_t quad, const Fq& v) {
#pragma unroll
        for (int i = 0; i < 8; i++) scr[(val * 8 + i) * 64 + quad] = v.l[i];
    };
    auto put = [&](uint32_t slot, const G1X& p) { stc(0, slot, p.x); stc(1, slot, p.y); stc(2, slot, p.zz); stc(3, slot, p.zzz); };
    auto get = [&](uint32_t slot) {
        G1X p;
        p.x = ldc(0, slot); p.y = ldc(1, slot); p.zz = ldc(2, slot); p.zzz = ldc(3, slot);
        return p;
    };
    // level 256 -> 128: one addition per lane (the upper half parks its points in the 128 slots)
    if (count > 128) {
        if (t >= 128) put(t - 128, acc);
        __syncthreads();
        if (t < 128) {
            acc = g1_add(acc, get(t));
            put(t, acc);                  // slot t is read and rewritten by thread t only
        }
    } else if (t < 128) {
        put(t, acc);
    }
    __syncthreads();
    uint32_t m0 = 64;
    while (m0 > 1 && m0 >= count) m0 >>= 1;               // slots >= count hold the identity: nothing to add there
    enum { V_U1 = 0, V_S1, V_U2, V_S2, V_PPU, V_PP, V_RR, V_RR2, V_PPP };
    const uint32_t q = t >> 2, j = t & 3;
    for (uint32_t m = m0; m >= 1; m >>= 1) {
        const bool act = t < 4 * m;
        // A wave none of whose lanes takes part still meets the barriers but skips the products (the branch is uniform
        // per wave: the whole wave jumps).  From 32 additions down that is most of the workgroup; executed by all four
        // waves the seven levels were a quarter of the vector instructions of a 20-window launch.
        const bool wave_act = (t & ~63u) < 4 * m;
        const uint32_t a = q, b = q + m;
        Fq mine = Fq::zero(), keep = Fq::zero();
        // R1: u1 | s1 | u2 | s2
        Fq A1 = Fq::zero(), B1 = Fq::zero();
        if (act) {
            A1 = ldc(j & 1, (j & 2) ? b : a);                    // x or y of P (lanes 0,1) / of Q (lanes 2,3)
            B1 = ldc(2 + (j & 1), (j & 2) ? a : b);              // zz or zzz of the OTHER point
        }
        if (wave_act) mine = A1 * B1;
        if (act) sts(V_U1 + j, q, mine);
        __syncthreads();
        // Every round is ONE product executed by all four lanes on lane-selected operands: the
        // selection diverges (a few LDS reads), the ~340-instruction product never does.
        // R2: pp_, pp | rr, rr2 | zz12 | zzz12
        Fq A2 = Fq::zero(), B2 = Fq::zero();
        if (act) {
            if (j < 2) {
                A2 = lds(V_U2 + j, q) - mine;                    // pp_ = u2 - u1 (lane 0), rr = s2 - s1 (lane 1)
                B2 = A2;
                sts(j == 0 ? V_PPU : V_RR, q, A2);
                if (j == 0) {
                    const Fq z1 = ldc(2, a), z2 = ldc(2, b);
                    flg[q] = (z1.is_zero() ? 1u : 0u) | (z2.is_zero() ? 2u : 0u) | (A2.is_zero() ? 4u : 0u);
                }
            } else {
                A2 = ldc(j, a);                                  // zz1, zz2 (lane 2) / zzz1, zzz2 (lane 3)
                B2 = ldc(j, b);
            }
        }
        if (wave_act) keep = A2 * B2;                                          // pp | rr2 | zz12 | zzz12
        if (act && j < 2) sts(j == 0 ? V_PP : V_RR2, q, keep);
        __syncthreads();
        // R3: qq = u1 pp | - | zz3 = zz12 pp | ppp = pp_ pp
        Fq A3 = Fq::zero(), B3 = Fq::zero();
        if (act) {
            B3 = lds(V_PP, q);
            A3 = j == 0 ? mine : keep;                           // lane 0: u1; lane 2: zz12; (lane 1: unused)
            if (j == 3) A3 = lds(V_PPU, q);
        }
        Fq r3 = Fq::zero();
        if (wave_act) r3 = A3 * B3;                                   // qq | - | zz3 | ppp
        if (act && j == 3) sts(V_PPP, q, r3);
        __syncthreads();
        // R4: x3, t2 = rr (qq - x3) | t1 = s1 ppp | - | zzz3 = zzz12 ppp
        Fq A4 = Fq::zero(), B4 = Fq::zero(), out = r3;           // lane 2: zz3
        if (act) {
            if (j == 0) {
                out = lds(V_RR2, q) - lds(V_PPP, q) - Fq::dbl(r3);   // x3
                A4 = lds(V_RR, q);
                B4 = r3 - out;
            } else if (j == 1) {
                A4 = mine;                                       // s1
                B4 = lds(V_PPP, q);
            } else if (j == 3) {
                A4 = keep;                                       // zzz12
                B4 = r3;                                         // ppp
            }
        }
        Fq r4 = Fq::zero();
        if (wave_act) r4 = A4 * B4;                                   // t2 | t1 | - | zzz3
        if (act && j == 0) sts(V_U2, q, r4);                     // t2 (the u2 slot is free by now)
        if (j == 3) out = r4;
        __syncthreads();
        // R5: y3, select around identities, write the sum into slot a
        bool special = false;
        if (act) {
            const uint32_t f = flg[q];
            if (j == 1) out = lds(V_U2, q) - r4;                 // y3 = t2 - t1
            if (f & 1u) stc(j, a, ldc(j, b));                    // P = O: the sum is Q
            else if (f & 2u) { /* Q = O: the sum is P, already in slot a */ }
            else if (f & 4u) special = true;                     // same x: doubling or opposite points
            else stc(j, a, out);
        }
        if (__syncthreads_or(special ? 1 : 0)) {
            if (act && j == 0 && special) put(a, g1_add(get(a), get(b)));
            __syncthreads();
        }
    }
    G1X r = G1X::inf();
    if (t == 0) r = get(0);
    return r;
}

// Sum of one projective point per thread (field29.hip.hpp: G1P29, complete addition) over a 256-thread workgroup; the result is
// valid in thread 0.  SIX lanes share an addition: round 1 is one product per lane, the six products go through LDS to the
// lanes of the group, each forms its two operands of round 2 (linear in the six), round 2 is again one product per lane,
// and the even lanes combine their product with their neighbour's into a coordinate of the sum.  Two product rounds and two
// barriers per pass instead of the four and five of the XYZZ addition, and no exceptional cases.  A wave holds ten groups
// (lanes 60..63 idle), a workgroup forty: a pass adds the last min(40, n / 2) live points onto the first ones, 256 points
// take eleven passes.
constexpr uint32_t RCB_PITCH = 272;                           // slots per limb row: 9 * 272 = 16 (mod 64), so the three
                                                              // coordinates of a slot sit in different banks
constexpr uint32_t RCB_GROUPS_PER_WAVE = 10, RCB_GROUPS = 4 * RCB_GROUPS_PER_WAVE;
constexpr uint32_t RCB_POINT_WORDS = 27 * RCB_PITCH;          // [3 coords][9 limbs][slots]
constexpr uint32_t RCB_SCR_WORDS = 9 * MSM_THREADS;           // [9 limbs][thread]: a round's products, then 9 X1 X2 likewise
constexpr uint32_t RCB_LDS_WORDS = RCB_POINT_WORDS + 2 * RCB_SCR_WORDS;

// `count` (workgroup-uniform, >= 1): threads >= count hold nothing (their point is ignored)
__device__ __forceinline__ G1P29 block_sum_rcb29(const G1P29& mine, uint32_t* sh /* [RCB_LDS_WORDS] */, uint32_t count = MSM_THREADS) {
    const uint32_t t = threadIdx.x;
    uint32_t* const scr = sh + RCB_POINT_WORDS;
    auto ldc = [&](uint32_t coord, uint32_t slot) {
        Fq29 v;
#pragma unroll
        for (int i = 0; i < 9; i++) v.l[i] = sh[(coord * 9 + i) * RCB_PITCH + slot];
        return v;
    };
    auto stc = [&](uint32_t coord, uint32_t slot, const Fq29& v) {
#pragma unroll
        for (int i = 0; i < 9; i++) sh[(coord * 9 + i) * RCB_PITCH + slot] = v.l[i];
    };
    auto lds = [&](uint32_t at) {                              // a value of the scratch rows: limb i at scr[at + 256 i]
        Fq29 v;
#pragma unroll
        for (int i = 0; i < 9; i++) v.l[i] = scr[at + i * MSM_THREADS];
        return v;
    };
    if (t < count) { stc(0, t, mine.x); stc(1, t, mine.y); stc(2, t, mine.z); }
    __syncthreads();
    const uint32_t lane = t & 63, wave = t >> 6;
    const uint32_t gl = lane / 6, j = lane - 6 * gl;           // group of the wave, lane of the group
    const uint32_t g = wave * RCB_GROUPS_PER_WAVE + gl;
    const uint32_t base = t - j;                               // the group's products of a round: scr[base + 0..5 + 256 i]
    // round 1, lane j: X1 X2 | Y1 Y2 | Zs1 Zs2 | (X1+Y1)(X2+Y2) | (Y1+Zs1)(Y2+Zs2) | (X1+Zs1)(X2+Zs2)
    const uint32_t c0 = (j == 1 || j == 4) ? 1u : j == 2 ? 2u : 0u;
    const uint32_t c1 = j == 3 ? 1u : 2u;
    const uint32_t pair_mask = j >= 3 ? ~0u : 0u;
    // round 2, lane j: t3 d | t4 t5 | d s | x9 t5 | s t4 | x9 t3, each operand  a - b - c + 3 p  or  a + c + 3 p  over the
    // round-1 products V0..V5 (scratch row offsets 0..5) and x9 (second scratch block): {a, b, c} offsets and masks
    enum : uint32_t { V0 = 0, V1, V2, V3, V4, V5, X9 = RCB_SCR_WORDS, MB = 1, MC = 2, MP = 4 };
    uint32_t aA, bA, cA, fA, aB, bB, cB, fB;
    switch (j) {
        case 0:  aA = V3; bA = V0; cA = V1; fA = MB | MC;      aB = V1; bB = V2; cB = V2; fB = MB;           break;   // t3, d
        case 1:  aA = V4; bA = V1; cA = V2; fA = MB | MC;      aB = V5; bB = V0; cB = V2; fB = MB | MC;      break;   // t4, t5
        case 2:  aA = V1; bA = V2; cA = V2; fA = MB;           aB = V1; bB = V2; cB = V2; fB = MC | MP;      break;   // d, s
        case 3:  aA = X9; bA = V0; cA = V0; fA = 0;            aB = V5; bB = V0; cB = V2; fB = MB | MC;      break;   // x9, t5
        case 4:  aA = V1; bA = V2; cA = V2; fA = MC | MP;      aB = V4; bB = V1; cB = V2; fB = MB | MC;      break;   // s, t4
        default: aA = X9; bA = V0; cA = V0; fA = 0;            aB = V3; bB = V0; cB = V1; fB = MB | MC;      break;   // x9, t3
    }
    aA += base; bA += base; cA += base; aB += base; bB += base; cB += base;
    const uint32_t mbA = 0u - (fA & 1u), mcA = 0u - ((fA >> 1) & 1u), mpA = 0u - ((fA >> 2) & 1u);
    const uint32_t mbB = 0u - (fB & 1u), mcB = 0u - ((fB >> 1) & 1u), mpB = 0u - ((fB >> 2) & 1u);
    // A group never spans two waves, so what its lanes exchange inside a pass needs no workgroup barrier: the LDS executes a
    // wave's instructions in order, the fence keeps the compiler from moving them across.
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint32_t n = count;
    while (n > 1) {
        const uint32_t adds = n / 2 < RCB_GROUPS ? n / 2 : RCB_GROUPS;
        // from ten additions down (a pass never has more than the one before) the first wave works alone: the others leave,
        // and it needs no workgroup barrier any more (the one that ended the previous pass made their sums visible)
        const bool solo = adds <= RCB_GROUPS_PER_WAVE;
        if (solo && wave != 0) break;
        const bool act = gl < RCB_GROUPS_PER_WAVE && g < adds;
        const bool wave_act = wave * RCB_GROUPS_PER_WAVE < adds;   // a wave with no group in the pass only meets the barriers
        const uint32_t a = g, b = n - adds + g;                // slot b is added onto slot a
        if (wave_act) {
            // (a lane with no addition in this pass reads slots of the shared array all the same -- b stays below 296 --
            // and multiplies what it finds: cheaper than selecting zeros, and it stores nothing)
            const Fq29 u = rcb_pair(ldc(c0, a), ldc(c1, a), pair_mask);
            const Fq29 v = rcb_pair(ldc(c0, b), ldc(c1, b), pair_mask);
            const Fq29 p1 = mul29(u, v);
            if (act) {
#pragma unroll
                for (int i = 0; i < 9; i++) scr[t + i * MSM_THREADS] = p1.l[i];
                if (j == 0) {
                    const Fq29 x9 = rcb_times9(p1);
#pragma unroll
                    for (int i = 0; i < 9; i++) scr[RCB_SCR_WORDS + t + i * MSM_THREADS] = x9.l[i];
                }
            }
        }
        wave_sync();
        if (wave_act) {
            const Fq29 A = rcb_lin(lds(aA), lds(bA), mbA, lds(cA), mcA, mpA);
            const Fq29 B = rcb_lin(lds(aB), lds(bB), mbB, lds(cB), mcB, mpB);
            const Fq29 p2 = mul29(A, B);
            Fq29 nb;                                           // the neighbour's product (lane + 1: same group for even j)
#pragma unroll
            for (int i = 0; i < 9; i++) nb.l[i] = __shfl_down(p2.l[i], 1);
            if (act && !(j & 1)) stc(j >> 1, a, rcb_out(p2, nb, j == 0));
        }
        if (solo) wave_sync(); else __syncthreads();
        n -= adds;
    }
    G1P29 r = g1p_identity();
    if (t == 0) { r.x = ldc(0, 0); r.y = ldc(1, 0); r.z = ldc(2, 0); }
    return r;
}

// lane <-> scalar i, blockIdx.y <-> group of T windows, blockIdx.x <-> (segment, MSM of its batch, 256-row block):
// only row blocks a segment visits are launched (an empty workgroup still costs ~0.1 us of dispatch)
template <bool LAZY29>
__global__ __launch_bounds__(MSM_THREADS, 2) void msm_direct_kernel(DirectArgs a) {
    __shared__ uint32_t sh[LAZY29 ? RCB_LDS_WORDS : COOP_LDS_WORDS];
    uint32_t lo = 0, hi = a.n_segs;                            // last segment with wg_base <= blockIdx.x
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.segs[mid].wg_base <= blockIdx.x) lo = mid; else hi = mid;
    }
    const DirectSeg sg = a.segs[lo];
    const uint32_t wg_id = blockIdx.y * gridDim.x + blockIdx.x;
    if (a.trace && threadIdx.x == 0) a.trace[3 * (size_t)wg_id] = wall_clock64();
    const uint32_t bx = blockIdx.x - sg.wg_base;
    const uint32_t m = bx / sg.gx, blk = bx % sg.gx;
    const uint32_t lane = blk * MSM_THREADS + threadIdx.x;
    const uint32_t visited = sg.prefix + sg.tail_count;
    const size_t part = sg.part_base + ((size_t)m * gridDim.y + blockIdx.y) * sg.gx + blk;
    const size_t i = lane < sg.prefix ? lane : (size_t)sg.tail_start + (lane - sg.prefix);
    const G1Affine* __restrict__ mult = sg.mult;
    const uint32_t w0 = blockIdx.y * sg.T;
    const uint32_t w1 = w0 + sg.T < sg.W ? w0 + sg.T : sg.W;
    G1X acc = G1X::inf();
    G1X29 acc29;
    acc29.inf = true;
    if (lane < visited && w0 < sg.W) {
        Fr raw = load_fr(sg.scalars + (size_t)m * sg.stride + i).to_raw();
        if (!raw.is_zero()) {
            uint64_t cy = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                cy += (uint64_t)raw.l[j] + sg.bias.l[j];
                raw.l[j] = (uint32_t)cy;
                cy >>= 32;
            }
            const uint32_t half = sg.NB;
            // next non-zero digit at or after window w: (table slot, sign), slot == NB means none
            raw_shr_any(raw, w0 * sg.c);               // the group's first window at bit 0
            auto next_digit = [&](uint32_t& w, uint32_t& neg) -> size_t {
                for (; w < w1; w++) {
                    uint32_t d = raw_pop_bits(raw, sg.c);
                    neg = 0;
                    if (w + 1 < sg.W) {
                        neg = d < half;
                        d = neg ? half - d : d - half;
                    }
                    if (d) return ((size_t)w * a.n_table + i) * sg.NB + (d - 1);
                }
                return (size_t)-1;
            };
            // the gather of entry k+1 is in flight while entry k is added (one 64 B random HBM read per
            // digit: ~2 us exposed per addition otherwise, with only a few waves per SIMD to hide it)
            uint32_t w = w0, neg = 0;
            size_t slot = next_digit(w, neg);
            G1Affine p;
            if (slot != (size_t)-1) p = load_affine(mult + slot);
            while (slot != (size_t)-1) {
                const G1Affine cur = p;
                const uint32_t cur_neg = neg;
                w++;
                slot = next_digit(w, neg);
                if (slot != (size_t)-1) p = load_affine(mult + slot);
                if (cur.is_inf()) continue;
                const Fq y = cur_neg ? Fq::neg(cur.y) : cur.y;
                if constexpr (LAZY29) g1_madd29(acc29, cur.x, y);
                else acc = g1_madd(acc, cur.x, y);
            }
        }
    }
    if (a.trace) {
        __syncthreads();
        if (threadIdx.x == 0) a.trace[3 * (size_t)wg_id + 1] = wall_clock64();
    }
    G1P29 accp;
    if constexpr (LAZY29) accp = block_sum_rcb29(g1p_from_x29(acc29), sh);
    else acc = block_sum_256(acc, sh);
    if (a.trace && threadIdx.x == 0) a.trace[3 * (size_t)wg_id + 2] = wall_clock64();
    // The workgroup that stores the last partial sum of an MSM adds them all up: no second dispatch (on a busy chip a
    // dispatch costs its stream tens of microseconds whatever its size).  Device-scope fences on both sides of the
    // count make the partials of workgroups on other XCDs (other L2s) visible to the one that reads them.
    __shared__ uint32_t is_last;
    const uint32_t P = sg.gx * gridDim.y;
    uint32_t* const done = a.done + sg.msm_base + m;
    // (the lazy form keeps its partial sums lazy and projective: they are converted once, after the merge)
    G1P29* const partials29 = reinterpret_cast<G1P29*>(a.partials);
    if (threadIdx.x == 0) {
        if constexpr (LAZY29) partials29[part] = accp;
        else a.partials[part] = acc;
        __threadfence();
        is_last = atomicAdd(done, 1u) + 1 == P;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    uint64_t* const trace_merge = a.trace ? a.trace + 3 * (size_t)gridDim.x * gridDim.y + 4 * (size_t)(sg.msm_base + m) : nullptr;
    if (trace_merge && threadIdx.x == 0) trace_merge[0] = wall_clock64();
    G1X sum = G1X::inf();
    if constexpr (LAZY29) {
        const G1P29* p = partials29 + sg.part_base + (size_t)m * P;
        G1P29 sump = g1p_identity();
        if (threadIdx.x < P) {
            sump = p[threadIdx.x];
            for (uint32_t j = threadIdx.x + MSM_THREADS; j < P; j += MSM_THREADS) sump = g1p_add29(sump, p[j]);
        }
        sump = block_sum_rcb29(sump, sh, P < MSM_THREADS ? P : MSM_THREADS);
        if (threadIdx.x == 0) {
            // the sum leaves as it is (lazy, projective): the host, which divides by Z anyway, converts it in a fraction
            // of the eight products that one lane here would spend on it
            if (trace_merge) trace_merge[1] = trace_merge[2] = wall_clock64();
            reinterpret_cast<G1P29*>(a.out)[sg.msm_base + m] = sump;
        }
    } else {
        const G1X* p = a.partials + sg.part_base + (size_t)m * P;
        for (uint32_t j = threadIdx.x; j < P; j += MSM_THREADS) sum = g1_add(sum, p[j]);
        sum = block_sum_256(sum, sh, P < MSM_THREADS ? P : MSM_THREADS);
    }
    if (threadIdx.x == 0) {
        if constexpr (!LAZY29) {
            if (trace_merge) trace_merge[1] = trace_merge[2] = wall_clock64();
            a.out[sg.msm_base + m] = sum;
        }
        *done = 0;                                             // ready for the next launch
        if (trace_merge) trace_merge[3] = wall_clock64();
        if (a.stamp) publish_stamp(a.stamp + sg.msm_base + m, a.seq);
    }
}

}  // namespace shp

using namespace shp;

static uint32_t pick_window(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = (int)lg - 1;
    if (c < 6) c = 6;
    if (c > 16) c = 16;
    return (uint32_t)c;
}

// Direct-table budget per plan (GiB); SHP_MSM_TABLE_GB=0 forces the bucket method.
// An SRS holds two bases.  The Lagrange basis carries EVERY commitment of a proof (the prover transforms the random
// polynomial and the quotient pieces to their values on the domain and commits to those), so it is the PRIMARY table:
// built first, it may take c = 15 (k = 13: 146 GB, 17 windows).  The coefficient basis serves shp_srs_commit callers
// only and gets a small table (8 GiB: c = 10 at k = 13); a stand-alone plan stops at c = 14.
enum class TableRole { Single, Primary, Secondary };

static double direct_budget_gib(TableRole role) {
    const char* e = getenv("SHP_MSM_TABLE_GB");
    if (e && *e) return role == TableRole::Secondary ? std::min(8.0, atof(e)) : atof(e);
    return role == TableRole::Primary ? 140.0 : role == TableRole::Secondary ? 8.0 : 84.0;
}

// scalars are canonical (< r < 2^254)
static uint32_t direct_windows(uint32_t c) {
    uint32_t W = (254 + c - 1) / c;
    if (254 - (int)(c * (W - 1)) > (int)c - 1) W++;       // the unsigned top digit (plus a carry) must be <= 2^(c-1)
    return W;
}

static double direct_table_bytes(size_t n, uint32_t c) {
    return (double)n * direct_windows(c) * (double)(1u << (c - 1)) * sizeof(G1Affine);
}

// HBM the library may still take on the current device: what hipMemGetInfo reports free, capped by the context's limit
// (shp_ctx_set_hbm_limit: bytes the whole process may hold on the device, 0 = no cap)
static double hbm_available(shp_ctx* ctx) {
    size_t free_b = 0, total_b = 0;
    SHP_HIP(hipMemGetInfo(&free_b, &total_b));
    double avail = (double)free_b;
    const uint64_t limit = ctx->hbm_limit.load();
    if (limit) {
        const double used = (double)total_b - (double)free_b;
        avail = std::min(avail, std::max(0.0, (double)limit - used));
    }
    return avail;
}

// largest c in [8, below) whose full multiples table fits both the per-table budget and the device, after 6 GiB of
// headroom (build scratch, proving workspaces).  A stand-alone plan may take half of what is free; of an SRS's two
// tables the primary one 55 % (the sibling follows) and the secondary one 85 % of what is then left.  0 = none fits: use
// buckets.
static uint32_t pick_direct_window(shp_ctx* ctx, size_t n, TableRole role, uint32_t below) {
    const double gib = 1073741824.0;
    const double share = role == TableRole::Primary ? 0.55 : role == TableRole::Secondary ? 0.85 : 0.5;
    const double budget = std::min(direct_budget_gib(role) * gib, (hbm_available(ctx) - 6.0 * gib) * share);
    const uint32_t cmax = role == TableRole::Primary ? 15 : 14;
    for (uint32_t c = std::min<uint32_t>(cmax, below - 1); c >= 8; c--)
        if (direct_table_bytes(n, c) <= budget) return c;
    return 0;
}

static void msm_build_direct(shp_ctx* ctx, shp_msm* m, const G1Affine* d_bases) {
    const size_t n = m->n;
    const uint32_t c = m->c, W = m->W, NB = m->NB;
    hipStream_t st = ctx->stream;
    SHP_HIP(hipMalloc(&m->d_table, (size_t)W * n * sizeof(G1Affine)));       // base rows 2^(cw) P_i
    hipLaunchKernelGGL(msm_table_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, st, d_bases, m->d_table, n, c, W);
    SHP_HIP(hipGetLastError());
    const size_t rows = (size_t)W * n;
    SHP_HIP(hipMalloc(&m->d_mult, rows * NB * sizeof(G1Affine)));
    // scratch for a batch of rows: XYZZ + running products, <= ~1.3 GB
    size_t batch_rows = ((size_t)1 << 23) / NB;
    if (batch_rows > rows) batch_rows = rows;
    DevBuf xyzz, pre;
    xyzz.reserve(batch_rows * NB * sizeof(G1X));
    pre.reserve(batch_rows * NB * sizeof(Fq));
    for (size_t r0 = 0; r0 < rows; r0 += batch_rows) {
        uint32_t nr = (uint32_t)std::min(batch_rows, rows - r0);
        hipLaunchKernelGGL(msm_multiples_kernel, dim3(nr), dim3(64), 0, st, m->d_table, r0, nr, NB, xyzz.as<G1X>(),
                           pre.as<Fq>(), m->d_mult);
    }
    SHP_HIP(hipGetLastError());
    SHP_HIP(hipStreamSynchronize(st));
    xyzz.release();
    pre.release();
    // bias K = sum_{w < W-1} 2^(c(w+1)-1)
    m->bias = Fr::zero();
    for (uint32_t w = 0; w + 1 < W; w++) {
        uint32_t bit = c * (w + 1) - 1;
        m->bias.l[bit >> 5] |= 1u << (bit & 31);
    }
}

static void msm_build(shp_ctx* ctx, const G1Affine* d_bases, size_t n, uint32_t window_bits, shp_msm** out,
                      TableRole role = TableRole::Single) {
    SHP_REQUIRE(ctx && out && n > 0, SHP_ERR_INVALID_ARG, "msm_create: bad arguments");
    SHP_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<shp_msm> m(new shp_msm);
    m->ctx = ctx;
    m->n = n;
    // window_bits == 0: direct multiples table when it fits the HBM budget, buckets otherwise;
    // an explicit window_bits always selects the bucket method with that window
    uint32_t cd = window_bits ? 0 : pick_direct_window(ctx, n, role, 16);
    while (cd) {
        m->direct = true;
        m->c = cd;
        m->W = direct_windows(cd);
        m->NB = 1u << (cd - 1);
        try {
            msm_build_direct(ctx, m.get(), d_bases);
            *out = m.release();
            return;
        } catch (const Error& e) {
            // another allocator got there first: give the partial tables back, step the window down (each step halves
            // the table; c = 12 costs 3.5 % of proving throughput) and, below c = 8, fall through to the bucket method
            if (e.hip_code != (int)hipErrorOutOfMemory) throw;
            (void)hipGetLastError();
            if (m->d_table) { (void)hipFree(m->d_table); m->d_table = nullptr; }
            if (m->d_mult) { (void)hipFree(m->d_mult); m->d_mult = nullptr; }
            m->direct = false;
            cd = pick_direct_window(ctx, n, role, cd);
        }
    }
    uint32_t c = window_bits ? window_bits : pick_window(n);
    SHP_REQUIRE(c >= 2 && c <= 20, SHP_ERR_INVALID_ARG, "msm_create: window_bits must be in [2,20]");
    uint32_t W = (255 + c - 1) / c;
    SHP_REQUIRE((uint64_t)n * W < (1ull << 31), SHP_ERR_INVALID_ARG, "msm_create: n * windows must be < 2^31");
    m->c = c;
    m->W = W;
    m->NB = 1u << (c - 1);
    SHP_HIP(hipMalloc(&m->d_table, (size_t)W * n * sizeof(G1Affine)));
    hipLaunchKernelGGL(msm_table_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, ctx->stream, d_bases, m->d_table, n, c, W);
    SHP_HIP(hipGetLastError());
    SHP_HIP(hipStreamSynchronize(ctx->stream));
    *out = m.release();
}

// host normalisation of `batch` XYZZ sums with one shared inversion (Montgomery's trick over ZZZ)
static void normalise_batch(const G1X* in, uint32_t batch, G1Affine* h_out) {
    std::vector<Fq> pre(batch);
    Fq acc = Fq::one();
    for (uint32_t b = 0; b < batch; b++) {
        pre[b] = acc;
        if (!in[b].is_inf()) acc = acc * in[b].zzz;
    }
    Fq inv = Fq::inv(acc);
    for (uint32_t b = batch; b-- > 0;) {
        const G1X& p = in[b];
        if (p.is_inf()) { h_out[b].x = Fq::zero(); h_out[b].y = Fq::zero(); continue; }
        Fq zi = inv * pre[b];
        inv = inv * p.zzz;
        Fq zz_inv = zi * zi * p.zz * p.zz;
        h_out[b].x = p.x * zz_inv;
        h_out[b].y = p.y * zi;
    }
}

// the same for the lazy projective sums (X : Y : 3 Z) of the direct kernel: x = 3 X / Zs, y = 3 Y / Zs
static void normalise_batch_p29(const G1P29* in, uint32_t batch, G1Affine* h_out) {
    std::vector<Fq> pre(batch), zs(batch);
    Fq acc = Fq::one();
    for (uint32_t b = 0; b < batch; b++) {
        zs[b] = to_fq(in[b].z);                                // canonical: zero exactly for the identity
        pre[b] = acc;
        if (!zs[b].is_zero()) acc = acc * zs[b];
    }
    Fq inv = Fq::inv(acc);
    for (uint32_t b = batch; b-- > 0;) {
        if (zs[b].is_zero()) { h_out[b].x = Fq::zero(); h_out[b].y = Fq::zero(); continue; }
        const Fq zi = inv * pre[b];
        inv = inv * zs[b];
        const Fq zi3 = zi + zi + zi;
        h_out[b].x = to_fq(in[b].x) * zi3;
        h_out[b].y = to_fq(in[b].y) * zi3;
    }
}

// All jobs in ONE launch of msm_direct_kernel (the final sums included), in two halves so that a caller can queue other
// work behind the launch before it waits: msm_launch_jobs returns once the kernel is queued, msm_finish_jobs waits for the
// sums -- for their stamps in pinned memory when the workspace polls (whatever else the stream still holds), for the
// stream otherwise -- normalises them and hands them out.  One launch per workspace at a time.
void msm_run_jobs(const shp::MsmJob* jobs, uint32_t n_jobs, shp::MsmWorkspace& ws, hipStream_t st) {
    msm_launch_jobs(jobs, n_jobs, ws, st);
    msm_finish_jobs(ws, st);
}

void msm_launch_jobs(const shp::MsmJob* jobs, uint32_t n_jobs, shp::MsmWorkspace& ws, hipStream_t st) {
    SHP_REQUIRE(!ws.pend_total, SHP_ERR_INTERNAL, "msm: the workspace already has a launch in flight");
    SHP_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= shp::MSM_MAX_JOBS, SHP_ERR_INVALID_ARG, "msm: bad job list");
    const shp_msm* m0 = jobs[0].m;
    SHP_REQUIRE(m0 && m0->direct, SHP_ERR_INVALID_ARG, "msm: job list needs direct-table plans");
    uint32_t W = 0;                                            // the most windows any of the tables has
    for (uint32_t k = 0; k < n_jobs; k++)
        if (jobs[k].m) W = std::max(W, jobs[k].m->W);
    std::vector<DirectSeg> segs(n_jobs);
    uint64_t busy_wgs = 0, total = 0, visited_all = 0;
    for (uint32_t k = 0; k < n_jobs; k++) {
        const shp::MsmJob& jb = jobs[k];
        SHP_REQUIRE(jb.m && jb.m->direct && jb.m->n == m0->n, SHP_ERR_INVALID_ARG,
                    "msm: the jobs of one launch need direct-table plans over the same number of bases");
        SHP_REQUIRE(jb.scalars && jb.h_out && jb.batch, SHP_ERR_INVALID_ARG, "msm: bad job");
        SHP_REQUIRE(jb.rows.prefix <= jb.m->n && (size_t)jb.rows.tail_start + jb.rows.tail_count <= jb.m->n &&
                        (jb.rows.tail_count == 0 || jb.rows.prefix <= jb.rows.tail_start),
                    SHP_ERR_INVALID_ARG, "msm: bad row ranges");
        const uint32_t visited = jb.rows.prefix + jb.rows.tail_count;
        DirectSeg& sg = segs[k];
        sg.mult = jb.m->d_mult; sg.scalars = jb.scalars; sg.stride = jb.stride;
        sg.prefix = jb.rows.prefix; sg.tail_start = jb.rows.tail_start; sg.tail_count = jb.rows.tail_count;
        sg.c = jb.m->c; sg.W = jb.m->W; sg.NB = jb.m->NB; sg.bias = jb.m->bias; sg.T = 0;
        sg.gx = ceil_div(std::max<uint32_t>(visited, 1), MSM_THREADS);
        sg.wg_base = (uint32_t)busy_wgs;
        sg.msm_base = (uint32_t)total;
        busy_wgs += (uint64_t)sg.gx * jb.batch;
        total += jb.batch;
        visited_all += (uint64_t)visited * jb.batch;
    }
    SHP_REQUIRE(busy_wgs < (1ull << 31), SHP_ERR_INVALID_ARG, "msm: launch too large");
    // T windows per lane: all of them when the launch fills the chip anyway (least tree overhead),
    // shorter chains for small launches (latency)
    // (measured: a lone proof's p50 is 1.87 / 1.82 / 1.82 / 1.79 / 1.84 / 1.88 ms at 384 / 512 / 640 / 768 / 1024 / 1536
    // waves -- a shallow optimum, 768 ahead in every later run as well; pool lanes set their own figure, 512)
    static const uint64_t base_waves = getenv("SHP_MSM_MIN_WAVES") ? (uint64_t)atol(getenv("SHP_MSM_MIN_WAVES")) : 768;
    const uint64_t min_waves = ws.min_waves ? ws.min_waves : base_waves;
    // the fewest window groups that reach min_waves, the windows spread evenly over them
    const uint64_t per_group = busy_wgs * (MSM_THREADS / 64);
    const uint32_t groups = (uint32_t)std::min<uint64_t>(W, std::max<uint64_t>(1, (min_waves + per_group - 1) / per_group));
    const uint32_t T = ceil_div(W, groups);
    const uint32_t gy = ceil_div(W, T);
    uint64_t parts = 0;
    for (uint32_t k = 0; k < n_jobs; k++) {
        segs[k].T = ceil_div(segs[k].W, gy);                  // a table's windows spread over the launch's groups
        segs[k].part_base = parts;
        parts += (uint64_t)jobs[k].batch * segs[k].gx * gy;
    }
    ws.pf.reserve(parts * std::max(sizeof(G1X), sizeof(G1P29)));   // (the lazy kernel stores its partial sums in its own form)
    ws.h_out.reserve(total * sizeof(G1X));                     // before the launch: the final sums are written
    ws.done.reserve(std::max<size_t>(total, 256) * sizeof(uint32_t));
    if (ws.done_zeroed != ws.done.cap) {                       // grown (or new): the counters start at zero, and every
        SHP_HIP(hipMemsetAsync(ws.done.p, 0, ws.done.cap, st));   // launch leaves them at zero again
        ws.done_zeroed = ws.done.cap;
    }
    // segment tables live in a small device arena: a prover's six commitment phases present the same tables
    // proof after proof (same buffers, same shapes), so short tables are uploaded once and found again by
    // content; longer or novel ones (a combiner's mixes) go through the scratch region behind the cache slots
    constexpr uint32_t SLOT_SEGS = 4, SLOTS = 16, SCRATCH_AT = SLOT_SEGS * SLOTS;
    ws.segs.reserve((SCRATCH_AT + shp::MSM_MAX_JOBS) * sizeof(DirectSeg));
    ws.h_segs.reserve((SCRATCH_AT + shp::MSM_MAX_JOBS) * sizeof(DirectSeg));
    const size_t seg_bytes = n_jobs * sizeof(DirectSeg);
    uint32_t seg_at = SCRATCH_AT;
    bool upload = true;
    if (n_jobs <= SLOT_SEGS) {
        for (uint32_t e = 0; e < ws.seg_cached && upload; e++)
            if (ws.seg_cache_n[e] == n_jobs && memcmp(ws.h_segs.as<DirectSeg>() + e * SLOT_SEGS, segs.data(), seg_bytes) == 0) {
                seg_at = e * SLOT_SEGS;
                upload = false;
            }
        if (upload && ws.seg_cached < SLOTS) {
            seg_at = ws.seg_cached * SLOT_SEGS;
            ws.seg_cache_n[ws.seg_cached++] = n_jobs;
        }
    }
    if (upload) {
        memcpy(ws.h_segs.as<DirectSeg>() + seg_at, segs.data(), seg_bytes);
        SHP_HIP(hipMemcpyAsync(ws.segs.as<DirectSeg>() + seg_at, ws.h_segs.as<DirectSeg>() + seg_at, seg_bytes,
                               hipMemcpyHostToDevice, st));
    }
    DirectArgs a;
    a.segs = ws.segs.as<DirectSeg>() + seg_at; a.n_segs = n_jobs;
    a.n_table = m0->n;
    a.partials = ws.pf.as<G1X>();
    a.done = ws.done.as<uint32_t>();
    a.out = ws.h_out.as<G1X>();                                // straight into pinned host memory: no copy dispatch
    KernelTimers none, *tm = ws.timers ? ws.timers : &none;
    const bool poll = ws.poll_results && !tm->enabled;         // (event timing needs the stream drained)
    a.stamp = nullptr;
    a.seq = 0;
    a.trace = nullptr;
    static const bool trace_on = getenv("SHP_MSM_TRACE") && atoi(getenv("SHP_MSM_TRACE")) != 0;
    if (trace_on) {
        ws.trace_wgs = busy_wgs * gy;
        ws.trace_msms = total;
        ws.h_trace.reserve((3 * ws.trace_wgs + 4 * total) * sizeof(uint64_t));
        memset(ws.h_trace.p, 0, (3 * ws.trace_wgs + 4 * total) * sizeof(uint64_t));
        a.trace = ws.h_trace.as<uint64_t>();
    }
    if (poll) {
        ws.h_stamp.reserve(std::max<size_t>(total, 64) * sizeof(uint32_t));
        a.stamp = ws.h_stamp.as<uint32_t>();
        a.seq = ++ws.stamp_seq ? ws.stamp_seq : ++ws.stamp_seq;   // never 0: a fresh buffer may hold anything but is unlikely to hold the next number
        for (size_t i = 0; i < total; i++) a.stamp[i] = a.seq - 1;
    }
    tm->begin(KID_MSM_DIRECT, st);
    static const bool lazy29 = !(getenv("SHP_MSM_LAZY29") && atoi(getenv("SHP_MSM_LAZY29")) == 0);
    if (lazy29) hipLaunchKernelGGL(msm_direct_kernel<true>, dim3((uint32_t)busy_wgs, gy), dim3(MSM_THREADS), 0, st, a);
    else hipLaunchKernelGGL(msm_direct_kernel<false>, dim3((uint32_t)busy_wgs, gy), dim3(MSM_THREADS), 0, st, a);
    tm->end(st);
    SHP_HIP(hipGetLastError());
    ws.pend.clear();
    for (uint32_t k = 0; k < n_jobs; k++) ws.pend.push_back({jobs[k].h_out, segs[k].msm_base, jobs[k].batch});
    ws.pend_total = (uint32_t)total;
    ws.pend_poll = poll;
    ws.pend_seq = a.seq;
    ws.pend_projective = lazy29;
    ws.last_adds = visited_all * W;                           // upper bound: one mixed add per non-zero digit
    ws.last_bytes = visited_all * 32 + (uint64_t)m0->n * 64;
}

void msm_finish_jobs(shp::MsmWorkspace& ws, hipStream_t st) {
    SHP_REQUIRE(ws.pend_total, SHP_ERR_INTERNAL, "msm: no launch in flight");
    const uint32_t total = ws.pend_total;
    ws.pend_total = 0;
    if (ws.pend_poll) shp::wait_stamps(ws.h_stamp.as<uint32_t>(), total, ws.pend_seq, st);
    else shp::sync_stream(st);
    if (ws.timers) ws.timers->collect();
    if (ws.trace_wgs) {                                            // SHP_MSM_TRACE: where a launch's time goes
        shp::sync_stream(st);
        const uint64_t* t = ws.h_trace.as<uint64_t>();
        const uint64_t* mg = t + 3 * ws.trace_wgs;
        uint64_t t0 = ~0ull, start_max = 0, sums_max = 0, tree_max = 0, merge0 = 0, merge1 = 0, merge2 = 0, merge3 = 0;
        for (uint64_t w = 0; w < ws.trace_wgs; w++) t0 = std::min(t0, t[3 * w]);
        for (uint64_t w = 0; w < ws.trace_wgs; w++) {
            start_max = std::max(start_max, t[3 * w] - t0);
            sums_max = std::max(sums_max, t[3 * w + 1] - t0);
            tree_max = std::max(tree_max, t[3 * w + 2] - t0);
        }
        for (uint32_t i = 0; i < ws.trace_msms; i++) {
            merge0 = std::max(merge0, mg[4 * i] - t0);
            merge1 = std::max(merge1, mg[4 * i + 1] - t0);
            merge2 = std::max(merge2, mg[4 * i + 2] - t0);
            merge3 = std::max(merge3, mg[4 * i + 3] - t0);
        }
        // the clock runs at 100 MHz
        fprintf(stderr, "[msm] %llu workgroups, %u sums: last start %.1f us, lanes' sums done %.1f, trees done %.1f, merge %.1f -> sum %.1f -> stored %.1f\n",
                (unsigned long long)ws.trace_wgs, ws.trace_msms, start_max / 100.0, sums_max / 100.0, tree_max / 100.0,
                merge0 / 100.0, merge1 / 100.0, merge3 / 100.0);
        (void)merge2;
        ws.trace_wgs = 0;
    }
    ws.h_aff.resize(total);
    if (ws.pend_projective) normalise_batch_p29(ws.h_out.as<G1P29>(), total, ws.h_aff.data());
    else normalise_batch(ws.h_out.as<G1X>(), total, ws.h_aff.data());
    for (const auto& pj : ws.pend) memcpy(pj.h_out, ws.h_aff.data() + pj.msm_base, (size_t)pj.batch * sizeof(G1Affine));
}

// ---------------------------------------------------------------- cross-lane combiner (engine.hpp)
void shp::MsmCombiner::run(const MsmJob* jobs, uint32_t n_jobs, hipStream_t lane_stream, hipEvent_t ready) {
    Req r;
    r.jobs = jobs;
    r.n_jobs = n_jobs;
    r.ready = ready;
    SHP_HIP(hipEventRecord(ready, lane_stream));
    {
        std::unique_lock<std::mutex> lk(mu);
        SHP_REQUIRE(!stop, SHP_ERR_INTERNAL, "msm combiner: shutting down");
        q.push_back(&r);
        cv_req.notify_one();
        cv_done.wait(lk, [&] { return r.done; });
    }
    if (r.status != SHP_OK) throw shp::Error(r.status, r.err);
}

void shp::MsmCombiner::loop() {
    (void)hipSetDevice(device);
    std::vector<Req*> batch;
    std::vector<MsmJob> all;
    for (;;) {
        batch.clear();
        all.clear();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_req.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;                                 // stopping and drained
            while (!q.empty() && all.size() + q.front()->n_jobs <= MSM_MAX_JOBS) {
                Req* r = q.front();
                q.pop_front();
                batch.push_back(r);
                all.insert(all.end(), r->jobs, r->jobs + r->n_jobs);
            }
        }
        auto serve = [&](const MsmJob* jobs, uint32_t n, Req* const* reqs, size_t n_reqs) {
            int32_t status = SHP_OK;
            std::string err;
            try {
                for (size_t i = 0; i < n_reqs; i++) SHP_HIP(hipStreamWaitEvent(st, reqs[i]->ready, 0));
                msm_run_jobs(jobs, n, ws, st);
                launches++;
            } catch (const shp::Error& e) {
                status = e.code;
                err = e.what();
            } catch (const std::exception& e) {
                status = SHP_ERR_INTERNAL;
                err = e.what();
            }
            for (size_t i = 0; i < n_reqs; i++) { reqs[i]->status = status; reqs[i]->err = err; }
            return status;
        };
        if (serve(all.data(), (uint32_t)all.size(), batch.data(), batch.size()) != SHP_OK && batch.size() > 1)
            for (Req* r : batch) serve(r->jobs, r->n_jobs, &r, 1);  // one bad request must not fail its neighbours
        {
            std::lock_guard<std::mutex> lk(mu);
            for (Req* r : batch) r->done = true;
            requests += batch.size();
        }
        cv_done.notify_all();
    }
}

shp::MsmCombiner* shp::msm_combiner_create(int device) {
    std::unique_ptr<MsmCombiner> c(new MsmCombiner);
    c->device = device;
    SHP_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    c->ws.timers = &c->timers;
    c->ws.min_waves = 512;                                         // throughput scheduling: long chains
    MsmCombiner* raw = c.get();
    c->th = std::thread([raw] { raw->loop(); });
    return c.release();
}

void shp::msm_combiner_destroy(MsmCombiner* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv_req.notify_all();
    if (c->th.joinable()) c->th.join();
    c->ws.release();
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

static void msm_run_direct(shp_msm* m, shp::MsmWorkspace& ws, const Fr* d_scalars, size_t n_scalars, size_t stride,
                           uint32_t batch, G1Affine* h_out, hipStream_t st, shp_msm* m2 = nullptr,
                           const Fr* d_scalars2 = nullptr, size_t stride2 = 0, uint32_t batch2 = 0,
                           const shp::MsmRows* rows = nullptr) {
    shp::MsmJob jobs[2];
    jobs[0] = shp::MsmJob{m, d_scalars, stride, batch, rows ? *rows : shp::MsmRows{(uint32_t)n_scalars, 0, 0}, h_out};
    if (rows) SHP_REQUIRE((size_t)rows->tail_start + rows->tail_count <= n_scalars, SHP_ERR_INVALID_ARG, "msm: bad row ranges");
    if (batch2) jobs[1] = shp::MsmJob{m2, d_scalars2, stride2, batch2, shp::MsmRows{(uint32_t)n_scalars, 0, 0}, h_out + batch};
    msm_run_jobs(jobs, batch2 ? 2 : 1, ws, st);
}

// `batch` MSMs over m and `batch2` over m2 (both direct-table plans of one shape) in ONE launch;
// h_out receives batch + batch2 points
void msm_run_ws2(shp_msm* m, const Fr* d_scalars, size_t stride, uint32_t batch, shp_msm* m2, const Fr* d_scalars2,
                 size_t stride2, uint32_t batch2, size_t n_scalars, shp::MsmWorkspace& ws, G1Affine* h_out, hipStream_t st,
                 const shp::MsmRows* rows) {
    SHP_REQUIRE(m && m2 && d_scalars && d_scalars2 && h_out && batch && batch2, SHP_ERR_INVALID_ARG, "msm2: bad argument");
    SHP_REQUIRE(n_scalars <= m->n, SHP_ERR_INVALID_ARG, "msm: more scalars than bases");
    if (m->direct && m2->direct) {
        msm_run_direct(m, ws, d_scalars, n_scalars, stride, batch, h_out, st, m2, d_scalars2, stride2, batch2, rows);
        return;
    }
    msm_run_ws(m, ws, d_scalars, n_scalars, stride, batch, h_out, st);
    msm_run_ws(m2, ws, d_scalars2, n_scalars, stride2, batch2, h_out + batch, st);
}

void msm_run_ws(shp_msm* m, shp::MsmWorkspace& ws, const Fr* d_scalars, size_t n_scalars, size_t stride,
                uint32_t batch, G1Affine* h_out, hipStream_t st) {
    SHP_REQUIRE(m && d_scalars && h_out, SHP_ERR_INVALID_ARG, "msm: null argument");
    SHP_REQUIRE(n_scalars <= m->n, SHP_ERR_INVALID_ARG, "msm: more scalars than bases");
    if (batch == 0) return;
    if (m->direct) {
        msm_run_direct(m, ws, d_scalars, n_scalars, stride, batch, h_out, st);
        return;
    }
    const uint32_t NB = m->NB, W = m->W;
    const size_t entries_stride = n_scalars * W;
    // entries per accumulation chunk: long chunks for huge inputs keep the number of partials per bucket
    // (merged by msm_fixup_kernel, one lane per bucket) small while every SIMD still has several waves
    const uint32_t T = entries_stride >= ((size_t)1 << 26) ? 256 : entries_stride >= ((size_t)1 << 22) ? 64 : 16;
    const uint32_t chunks = ceil_div(entries_stride, T);
    ws.cnt.reserve((size_t)batch * (NB + 2) * 4);
    ws.start.reserve((size_t)batch * (NB + 2) * 4);
    ws.entries.reserve((size_t)batch * entries_stride * 4);
    const uint32_t red_Q = NB < 2048 ? NB : 2048, red_nW = NB / red_Q;   // slices of the bucket reduction
    ws.pf.reserve(std::max((size_t)batch * chunks * sizeof(Partial), (size_t)batch * 2 * red_nW * sizeof(G1X)));
    ws.pl.reserve((size_t)batch * chunks * sizeof(Partial));
    ws.buckets.reserve((size_t)batch * (NB + 1) * sizeof(G1X));
    ws.out.reserve((size_t)batch * sizeof(G1X));
    uint32_t* cnt = ws.cnt.as<uint32_t>();
    uint32_t* start = ws.start.as<uint32_t>();
    uint32_t* entries = ws.entries.as<uint32_t>();
    G1X* buckets = ws.buckets.as<G1X>();

    KernelTimers none, *tm = ws.timers ? ws.timers : &none;
    // counting sort by bucket: LDS histograms per workgroup, no global atomics
    const uint32_t G = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)m->ctx->num_cus, ceil_div(n_scalars, DIG_THREADS)));
    const size_t lds_bytes = (size_t)(NB + 1) * 4;
    SHP_REQUIRE(lds_bytes <= 160 * 1024 - 512, SHP_ERR_INVALID_ARG, "msm: window too wide for the LDS histogram (c <= 16)");
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)msm_hist_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)msm_hist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void*)msm_bin_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    });
    ws.wg_hist.reserve((size_t)batch * G * (NB + 1) * 4);
    uint32_t* wg_hist = ws.wg_hist.as<uint32_t>();
    SHP_HIP(hipMemsetAsync(cnt, 0, (size_t)batch * (NB + 2) * 4, st));
    // two-level scatter for large inputs (whole bins of 256 buckets, and enough entries for a slice to bring a run to
    // each bucket); SHP_MSM_TWO_LEVEL=0/1 forces the choice (tuning and tests)
    const char* two_env = getenv("SHP_MSM_TWO_LEVEL");
    const int force_two = two_env && *two_env ? atoi(two_env) : -1;
    const bool two_level = (NB >> BIN_LOG) >= 2 && (force_two >= 0 ? force_two != 0 : entries_stride >= ((size_t)1 << 22));
    tm->begin(KID_MSM_DIGITS, st);
    hipLaunchKernelGGL(msm_hist_kernel<0>, dim3(G, batch), dim3(DIG_THREADS), lds_bytes, st, d_scalars, n_scalars, stride,
                       m->n, m->c, W, NB, wg_hist, (const uint32_t*)nullptr, (uint32_t*)nullptr, entries_stride, (uint64_t*)nullptr);
    hipLaunchKernelGGL(msm_hist_prefix_kernel, dim3(ceil_div(NB + 1, 256), batch), dim3(256), 0, st, wg_hist, G, NB, cnt);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(batch), dim3(MSM_THREADS), 0, st, cnt, start, NB);
    if (two_level) {
        ws.binned.reserve((size_t)batch * entries_stride * 8);
        ws.cursor.reserve((size_t)batch * (NB + 2) * 4);
        SHP_HIP(hipMemcpyAsync(ws.cursor.p, start, (size_t)batch * (NB + 2) * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(msm_hist_kernel<2>, dim3(G, batch), dim3(DIG_THREADS), (size_t)(NB >> BIN_LOG) * 4, st, d_scalars,
                           n_scalars, stride, m->n, m->c, W, NB, wg_hist, start, (uint32_t*)nullptr, entries_stride,
                           ws.binned.as<uint64_t>());
        hipLaunchKernelGGL(msm_bin_scatter_kernel, dim3(ceil_div(entries_stride, SLICE), batch), dim3(DIG_THREADS),
                           (size_t)NB * 4, st, ws.binned.as<uint64_t>(), entries_stride, start, NB, ws.cursor.as<uint32_t>(),
                           entries);
    } else {
        hipLaunchKernelGGL(msm_hist_kernel<1>, dim3(G, batch), dim3(DIG_THREADS), lds_bytes, st, d_scalars, n_scalars, stride,
                           m->n, m->c, W, NB, wg_hist, start, entries, entries_stride, (uint64_t*)nullptr);
    }
    tm->end(st);
    SHP_HIP(hipMemsetAsync(buckets, 0, (size_t)batch * (NB + 1) * sizeof(G1X), st));
    tm->begin(KID_MSM_ACCUMULATE, st);
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3(ceil_div(chunks, MSM_THREADS), batch), dim3(MSM_THREADS), 0, st,
                       m->d_table, entries, entries_stride, start, NB, T, chunks, buckets, ws.pf.as<Partial>(),
                       ws.pl.as<Partial>());
    tm->end(st);
    tm->begin(KID_MSM_FIXUP, st);
    hipLaunchKernelGGL(msm_fixup_kernel, dim3(ceil_div(NB, MSM_THREADS), batch), dim3(MSM_THREADS), 0, st, start, NB, T,
                       chunks, buckets, ws.pf.as<Partial>(), ws.pl.as<Partial>());
    tm->end(st);
    tm->begin(KID_MSM_REDUCE, st);
    {
        const uint32_t Q = red_Q, nW = red_nW;                           // powers of two, nW <= 16 (c <= 16)
        G1X* L = ws.pf.as<G1X>();                                        // the partials are dead by now: reuse as L | R
        G1X* Rr = L + (size_t)batch * nW;
        hipLaunchKernelGGL(msm_reduce_kernel, dim3(nW, batch), dim3(MSM_THREADS), 0, st, buckets, NB, Q, L, Rr);
        hipLaunchKernelGGL(msm_reduce_combine_kernel, dim3(batch), dim3(64), 0, st, L, Rr, nW, Q, ws.out.as<G1X>());
    }
    tm->end(st);
    SHP_HIP(hipGetLastError());
    ws.h_out.reserve((size_t)batch * sizeof(G1X));
    ws.h_totals.reserve((size_t)batch * 4);
    SHP_HIP(hipMemcpyAsync(ws.h_out.p, ws.out.p, (size_t)batch * sizeof(G1X), hipMemcpyDeviceToHost, st));
    SHP_HIP(hipMemcpy2DAsync(ws.h_totals.p, 4, start + NB + 1, (size_t)(NB + 2) * 4, 4, batch,
                             hipMemcpyDeviceToHost, st));
    SHP_HIP(hipStreamSynchronize(st));
    tm->collect();
    normalise_batch(ws.h_out.as<G1X>(), batch, h_out);
    ws.last_adds = 0;
    for (uint32_t b = 0; b < batch; b++) ws.last_adds += ws.h_totals.as<uint32_t>()[b];
    ws.last_bytes = (uint64_t)batch * n_scalars * 32 + (uint64_t)n_scalars * 64;
}

void msm_run(shp_msm* m, const Fr* d_scalars, size_t n_scalars, size_t stride, uint32_t batch, G1Affine* h_out,
             hipStream_t st) {
    SHP_REQUIRE(m, SHP_ERR_INVALID_ARG, "msm: null plan");
    std::lock_guard<std::mutex> g(m->mu);
    msm_run_ws(m, m->ws, d_scalars, n_scalars, stride, batch, h_out, st);
}

void shp::msm_create_for_srs(shp_ctx* ctx, const void* d_bases, size_t n, bool primary, shp_msm** out) {
    msm_build(ctx, (const G1Affine*)d_bases, n, 0, out, primary ? TableRole::Primary : TableRole::Secondary);
}

extern "C" {

int32_t shp_msm_create_dev(shp_ctx* ctx, const void* d_bases, size_t n, uint32_t window_bits, shp_msm** out) {
    return guarded([&] { msm_build(ctx, (const G1Affine*)d_bases, n, window_bits, out); });
}

int32_t shp_msm_create(shp_ctx* ctx, const void* h_bases, size_t n, uint32_t window_bits, shp_msm** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && h_bases && n > 0, SHP_ERR_INVALID_ARG, "msm_create: bad arguments");
        SHP_HIP(hipSetDevice(ctx->device));
        G1Affine* d = nullptr;
        SHP_HIP(hipMalloc(&d, n * sizeof(G1Affine)));
        hipError_t e = hipMemcpy(d, h_bases, n * sizeof(G1Affine), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); SHP_HIP(e); }
        try {
            msm_build(ctx, d, n, window_bits, out);
        } catch (...) {
            (void)hipFree(d);
            throw;
        }
        (void)hipFree(d);
    });
}

int32_t shp_msm_destroy(shp_msm* m) {
    return guarded([&] {
        if (m && m->refs.fetch_sub(1) == 1) delete m;      // ~shp_msm frees the tables and the workspace
    });
}

int32_t shp_msm_g1(shp_msm* m, const void* d_scalars, size_t n_scalars, size_t stride_elems, uint32_t batch,
                   void* h_out_affine, void* stream) {
    return guarded([&] {
        SHP_REQUIRE(m, SHP_ERR_INVALID_ARG, "msm: null plan");
        SHP_HIP(hipSetDevice(m->ctx->device));
        msm_run(m, (const Fr*)d_scalars, n_scalars, stride_elems ? stride_elems : n_scalars, batch,
                (G1Affine*)h_out_affine, m->ctx->pick(stream));
    });
}

int32_t shp_msm_last_stats(shp_msm* m, uint64_t* point_adds, uint64_t* bytes) {
    return guarded([&] {
        SHP_REQUIRE(m, SHP_ERR_INVALID_ARG, "msm: null plan");
        if (point_adds) *point_adds = m->ws.last_adds;
        if (bytes) *bytes = m->ws.last_bytes;
    });
}

int32_t shp_diag_multiples(shp_ctx* ctx, const void* d_bases, size_t n_rows, uint32_t multiples, void* d_out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && d_bases && d_out && n_rows >= 1, SHP_ERR_INVALID_ARG, "diag_multiples: null argument");
        SHP_REQUIRE(multiples >= 64 && multiples <= (1u << 16) && (multiples & (multiples - 1)) == 0, SHP_ERR_INVALID_ARG,
                    "diag_multiples: multiples must be a power of two in [64, 65536]");
        SHP_REQUIRE((uint64_t)n_rows * multiples <= (1ull << 26), SHP_ERR_INVALID_ARG, "diag_multiples: more than 2^26 points");
        SHP_HIP(hipSetDevice(ctx->device));
        size_t batch_rows = std::max<size_t>(1, ((size_t)1 << 23) / multiples);
        if (batch_rows > n_rows) batch_rows = n_rows;
        DevBuf xyzz, pre;
        xyzz.reserve(batch_rows * multiples * sizeof(G1X));
        pre.reserve(batch_rows * multiples * sizeof(Fq));
        for (size_t r0 = 0; r0 < n_rows; r0 += batch_rows) {
            const uint32_t nr = (uint32_t)std::min(batch_rows, n_rows - r0);
            hipLaunchKernelGGL(msm_multiples_kernel, dim3(nr), dim3(64), 0, ctx->stream, (const G1Affine*)d_bases, r0, nr, multiples,
                               xyzz.as<G1X>(), pre.as<Fq>(), (G1Affine*)d_out);
        }
        hipError_t e = hipGetLastError();
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        xyzz.release();
        pre.release();
        SHP_HIP(e);
        SHP_HIP(e2);
    });
}

int32_t shp_msm_info(const shp_msm* m, int32_t* direct, uint32_t* window_bits, uint64_t* table_bytes) {
    return guarded([&] {
        SHP_REQUIRE(m, SHP_ERR_INVALID_ARG, "msm: null plan");
        if (direct) *direct = m->direct ? 1 : 0;
        if (window_bits) *window_bits = m->c;
        if (table_bytes)
            *table_bytes = (uint64_t)m->W * m->n * sizeof(G1Affine) * (m->direct ? (uint64_t)m->NB + 1 : 1);
    });
}

}  // extern "C"
