// Device kernels of the PLONK prover (halo2 create_proof from the lookup
// permutation to the SHPLONK opening):
// expression-program interpreter, grand products, lookup permutation, quotient terms,
// evaluations and the evaluation-form SHPLONK quotients.  Everything is element-wise or a
// single-workgroup scan over n <= 2^16 rows: latency-, not bandwidth-bound at k = 13, so the
// design goal is few launches and no host round trips inside a phase.
#pragma once
#include "engine.hpp"
#include "hostcrypto.hpp"
#include "plan.hpp"
#include "fr_mem.hip.hpp"

namespace shp {

// ------------------------------------------------------------------ expression programs
// One lane per row.  The plan's three-address program runs over "slots" kept in LDS as
// [slot][16-byte plane][lane] (lane-contiguous: conflict free); the instruction stream is wave
// uniform, so the opcode switch never diverges.  OUT folds into acc = acc * mult + value:
// mult = y for the gate part of the quotient, theta for lookup compression.
struct ProgArgs {
    const Instr* prog;
    uint32_t n_instr, nslots;
    const Fr* consts;
    const Query* adv_q;
    const Query* fix_q;
    const Query* inst_q;
    const Fr* adv_base; size_t adv_stride;
    const Fr* fix_base; size_t fix_stride;
    const Fr* inst_base;
    // row r, rotation q.rot -> (r & hi_mask) | ((r + q.rot * rot_step) & size_mask): hi_mask = 0 on the Lagrange
    // domain; on the quotient domain (cosets c_j * H laid out coset-major) the coset index stays, the row moves
    uint32_t size_mask, rot_step, hi_mask;
    Fr mult;
    const Fr* acc_in;
    Fr* out;
    Fr* out_raw;      // optional second copy of the result as a plain integer (out of Montgomery form), for the sort keys
};

constexpr int PROG_THREADS = 64;

__device__ __forceinline__ void prog_fold_body(const ProgArgs& a) {
    // slots as two 16-byte planes per slot, [slot][plane][lane]: one ds_read_b128 / ds_write_b128 per plane,
    // lane-contiguous (conflict free) -- a quarter of the LDS instructions of a [slot][limb][lane] layout
    extern __shared__ uint4 slots[];
    const uint32_t tid = threadIdx.x;
    const uint32_t row = blockIdx.x * PROG_THREADS + tid;
    auto get = [&](uint32_t s) {
        const uint4 a0 = slots[(s * 2) * PROG_THREADS + tid], a1 = slots[(s * 2 + 1) * PROG_THREADS + tid];
        Fr v;
        v.l[0] = a0.x; v.l[1] = a0.y; v.l[2] = a0.z; v.l[3] = a0.w;
        v.l[4] = a1.x; v.l[5] = a1.y; v.l[6] = a1.z; v.l[7] = a1.w;
        return v;
    };
    auto put = [&](uint32_t s, const Fr& v) {
        slots[(s * 2) * PROG_THREADS + tid] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        slots[(s * 2 + 1) * PROG_THREADS + tid] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    };
    Fr acc = a.acc_in ? ld(a.acc_in + row) : Fr::zero();
    for (uint32_t i = 0; i < a.n_instr; i++) {
        const Instr in = a.prog[i];
        switch (in.op) {
            case OP_CONST: put(in.dst, ld(a.consts + in.a)); break;
            case OP_ADV: {
                Query q = a.adv_q[in.a];
                uint32_t r = (row & a.hi_mask) | ((row + (uint32_t)(q.rot * (int32_t)a.rot_step)) & a.size_mask);
                put(in.dst, ld(a.adv_base + (size_t)q.col * a.adv_stride + r));
                break;
            }
            case OP_FIX: {
                Query q = a.fix_q[in.a];
                uint32_t r = (row & a.hi_mask) | ((row + (uint32_t)(q.rot * (int32_t)a.rot_step)) & a.size_mask);
                put(in.dst, ld(a.fix_base + (size_t)q.col * a.fix_stride + r));
                break;
            }
            case OP_INST: {
                Query q = a.inst_q[in.a];
                uint32_t r = (row & a.hi_mask) | ((row + (uint32_t)(q.rot * (int32_t)a.rot_step)) & a.size_mask);
                put(in.dst, ld(a.inst_base + r));
                break;
            }
            case OP_ADD: put(in.dst, get(in.a) + get(in.b)); break;
            case OP_SUB: put(in.dst, get(in.a) - get(in.b)); break;
            case OP_MUL: put(in.dst, get(in.a) * get(in.b)); break;
            case OP_NEG: put(in.dst, Fr::neg(get(in.a))); break;
            case OP_OUT: acc = acc * a.mult + get(in.a); break;
            default: break;
        }
    }
    st(a.out + row, acc);
    if (a.out_raw) st(a.out_raw + row, acc.to_raw());
}

// The same program with the slot file in VECTOR REGISTERS (plans with <= 16 slots): one 16-wide register vector
// per limb, indexed by the wave-uniform slot number (gfx9 VGPR index mode / v_movrel) -- no LDS at all, so the
// launch is not held to five waves per CU by 30 KB of slots per wave and leaves the CU's LDS to the kernels of
// the other proving lanes.
typedef uint32_t slotvec_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void prog_fold_reg_body(const ProgArgs& a) {
    slotvec_t L[8];
#pragma unroll
    for (int k = 0; k < 8; k++) L[k] = (slotvec_t)(0u);
    const uint32_t row = blockIdx.x * PROG_THREADS + threadIdx.x;
    auto get = [&](uint32_t s) {
        s = __builtin_amdgcn_readfirstlane(s);
        Fr v;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = L[k][s];
        return v;
    };
    auto put = [&](uint32_t s, const Fr& v) {
        s = __builtin_amdgcn_readfirstlane(s);
#pragma unroll
        for (int k = 0; k < 8; k++) L[k][s] = v.l[k];
    };
    Fr acc = a.acc_in ? ld(a.acc_in + row) : Fr::zero();
    // one read site and one write site of the slot vectors per instruction (operands are fetched whether or not
    // the opcode uses them): the register allocator keeps the eight vectors in place instead of copying them
    // at every merge of a per-opcode branch
    for (uint32_t i = 0; i < a.n_instr; i++) {
        const Instr in = a.prog[i];
        const Fr x = get(in.a & 15u), y = get(in.b & 15u);
        Fr r = x;
        uint32_t dst = in.dst;
        switch (in.op) {
            case OP_CONST: r = ld(a.consts + in.a); break;
            case OP_ADV: {
                Query q = a.adv_q[in.a];
                uint32_t rr = (row & a.hi_mask) | ((row + (uint32_t)(q.rot * (int32_t)a.rot_step)) & a.size_mask);
                r = ld(a.adv_base + (size_t)q.col * a.adv_stride + rr);
                break;
            }
            case OP_FIX: {
                Query q = a.fix_q[in.a];
                uint32_t rr = (row & a.hi_mask) | ((row + (uint32_t)(q.rot * (int32_t)a.rot_step)) & a.size_mask);
                r = ld(a.fix_base + (size_t)q.col * a.fix_stride + rr);
                break;
            }
            case OP_INST: {
                Query q = a.inst_q[in.a];
                uint32_t rr = (row & a.hi_mask) | ((row + (uint32_t)(q.rot * (int32_t)a.rot_step)) & a.size_mask);
                r = ld(a.inst_base + rr);
                break;
            }
            case OP_ADD: r = x + y; break;
            case OP_SUB: r = x - y; break;
            case OP_MUL: r = x * y; break;
            case OP_NEG: r = Fr::neg(x); break;
            case OP_OUT: acc = acc * a.mult + x; dst = in.a; break;      // the slot keeps its value
            default: dst = in.a; break;
        }
        put(dst & 15u, r);
    }
    st(a.out + row, acc);
    if (a.out_raw) st(a.out_raw + row, acc.to_raw());
}

// One or two programs per launch (grid.y = 1 or 2, blockIdx.y selects -- by branch, not by index: a by-value argument
// indexed at run time would be copied to scratch): the input and table halves of a lookup's compression share a dispatch.
__global__ __launch_bounds__(PROG_THREADS) void prog_fold_kernel(ProgArgs a0, ProgArgs a1) {
    if (blockIdx.y == 0) prog_fold_body(a0);
    else prog_fold_body(a1);
}
__global__ __launch_bounds__(PROG_THREADS) void prog_fold_reg_kernel(ProgArgs a0, ProgArgs a1) {
    if (blockIdx.y == 0) prog_fold_reg_body(a0);
    else prog_fold_reg_body(a1);
}

// rows [r0, r1) of every column (blockIdx.y) to zero: what a previous proof assigned beyond this proof's rows
__global__ void zero_rows_kernel(Fr* cols, size_t stride, uint32_t r0, uint32_t r1) {
    const uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r < r1) st(cols + (size_t)blockIdx.y * stride + r, Fr::zero());
}

// ------------------------------------------------------------------ blinding stream
struct ChaKey { uint32_t w[8]; };     // by value in the kernel arguments: no upload dispatch for 32 bytes
__global__ void chacha_fr_kernel(ChaKey key8, uint64_t counter0, uint32_t count, Fr* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t key[8];
#pragma unroll
    for (int j = 0; j < 8; j++) key[j] = key8.w[j];
    st(out + i, fr_draw(key, counter0 + i));
}

// Everything a proof brings to the device, in ONE dispatch: the kernel reads the lane's pinned witness grid itself
// (three copies and a kernel before; a dispatch costs a busy stream tens of microseconds whatever its size).
//   blockIdx.y <  n_adv   advice column y: rows [0, rows_now) and the blinding rows [u, n)
//   blockIdx.y == n_adv   the instance column (n_inst values; the rows behind them stay zero)
//   blockIdx.y == n_adv+1 the n coefficients of the random polynomial: ChaCha20 draws, or the caller's (h_rand)
struct IngestArgs {
    const Fr* grid;          // pinned host memory, [n_adv + 1][n]
    Fr* vals;                // device, advice columns [n_adv][n]
    Fr* inst;                // device instance column
    Fr* rand_poly;
    const Fr* h_rand;        // pinned host memory or null
    uint32_t n, n_adv, rows_now, u, n_inst;
    uint32_t y0;             // first grid row of this launch: rows 0 .. n_adv - 1 advice, n_adv instances, n_adv + 1 random polynomial
    ChaKey key;
    uint64_t counter0;
};
__global__ void ingest_kernel(IngestArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y + a.y0;
    if (y < a.n_adv) {
        const uint32_t moved = a.rows_now + (a.n - a.u);
        if (i >= moved) return;
        const uint32_t r = i < a.rows_now ? i : a.u + (i - a.rows_now);
        st(a.vals + (size_t)y * a.n + r, ld(a.grid + (size_t)y * a.n + r));
    } else if (y == a.n_adv) {
        if (i < a.n_inst) st(a.inst + i, ld(a.grid + (size_t)a.n_adv * a.n + i));
    } else {
        if (i >= a.n) return;
        if (a.h_rand) {
            st(a.rand_poly + i, ld(a.h_rand + i));
        } else {
            uint32_t key[8];
#pragma unroll
            for (int j = 0; j < 8; j++) key[j] = a.key.w[j];
            st(a.rand_poly + i, fr_draw(key, a.counter0 + i));
        }
    }
}

// ------------------------------------------------------------------ grand products
constexpr int MAX_PERM_COLS = 64;
struct PermArgs {
    const Fr* col[MAX_PERM_COLS];       // per permutation column: values (lagrange) or coset
    Fr dbeta[MAX_PERM_COLS];            // delta^ci * beta
    const Fr* sigma;                    // [PC][size]
    size_t sigma_stride;
    const Fr* xs;                       // omega^i (lagrange) or zeta*omega_ext^i (extended)
    uint32_t n_cols, chunk;
    Fr beta, gamma;
};

// ratio numerators / denominators of permutation set s on the lagrange domain
struct LookupRatioArgs {
    const Fr *cin, *ctab, *pin, *ptab;
    size_t pstride;
    Fr beta, gamma;
};
__device__ __forceinline__ void lookup_ratio_body(const LookupRatioArgs& a, uint32_t l, uint32_t i, uint32_t n, Fr* num, Fr* den) {
    Fr nu = (ld(a.cin + (size_t)l * n + i) + a.beta) * (ld(a.ctab + (size_t)l * n + i) + a.gamma);
    Fr de = (ld(a.pin + (size_t)l * a.pstride + i) + a.beta) * (ld(a.ptab + (size_t)l * a.pstride + i) + a.gamma);
    st(num + (size_t)l * n + i, nu);
    st(den + (size_t)l * n + i, de);
}
// grid (n / 256, n_sets + n_lookups): rows of blockIdx.y below n_sets are permutation sets, the rest lookups (their
// ratios follow the sets' in num / den) -- one dispatch for every grand product of the proof
__global__ void perm_ratio_kernel(PermArgs a, LookupRatioArgs lk, uint32_t n_sets, uint32_t n, Fr* num, Fr* den) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = blockIdx.y;
    if (i >= n) return;
    if (s >= n_sets) {
        lookup_ratio_body(lk, s - n_sets, i, n, num + (size_t)n_sets * n, den + (size_t)n_sets * n);
        return;
    }
    Fr nu = Fr::one(), de = Fr::one();
    Fr x = ld(a.xs + i);
    uint32_t lo = s * a.chunk, hi = lo + a.chunk < a.n_cols ? lo + a.chunk : a.n_cols;
    for (uint32_t c = lo; c < hi; c++) {
        Fr v = ld(a.col[c] + i);
        de = de * (a.beta * ld(a.sigma + (size_t)c * a.sigma_stride + i) + v + a.gamma);
        nu = nu * (a.dbeta[c] * x + v + a.gamma);
    }
    st(num + (size_t)s * n + i, nu);
    st(den + (size_t)s * n + i, de);
}

// block-wide multiplicative scans over one value per thread (Hillis-Steele in LDS)
__device__ __forceinline__ Fr block_scan_fwd_excl(Fr v, Fr* sh, uint32_t t, uint32_t nt) {
    sh[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < nt; off <<= 1) {
        Fr x = sh[t];
        if (t >= off) x = x * sh[t - off];
        __syncthreads();
        sh[t] = x;
        __syncthreads();
    }
    Fr r = t ? sh[t - 1] : Fr::one();
    __syncthreads();
    return r;
}
__device__ __forceinline__ Fr block_scan_bwd_excl(Fr v, Fr* sh, uint32_t t, uint32_t nt) {
    sh[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < nt; off <<= 1) {
        Fr x = sh[t];
        if (t + off < nt) x = x * sh[t + off];
        __syncthreads();
        sh[t] = x;
        __syncthreads();
    }
    Fr r = t + 1 < nt ? sh[t + 1] : Fr::one();
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------ product scans over many workgroups
// Prefix products of `a` and suffix products of `b` over `cnt` products of `n` elements each, in two launches.  Every
// thread owns E CONSECUTIVE elements: their product is a serial chain (one product per element), only the per-thread
// products go through the wave-wide scan (twelve products per thread), and the apply kernels walk the thread's elements
// again -- about 7 products per element for E = 8 where one element per thread pays 18.  E = 1 keeps the shortest
// dependency chain and is what a lone proof uses; pool lanes, which fill the GPU between them, use E = 8.
//   scan_local   per thread: exclusive prefix of a and exclusive suffix of b over the threads of its workgroup, and the
//                workgroup's totals; the workgroup that stores a product's last totals goes on with scan_totals_body:
//                exclusive prefix / suffix of the block totals and the ONE inversion of the product of all b (Kaliski)
//   apply        grand_product_apply / batch_invert_apply
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_E_POOL = 8;

__device__ __forceinline__ Fr shfl_up_fr(const Fr& v, int d) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl_up(v.l[i], d);
    return r;
}
__device__ __forceinline__ Fr shfl_down_fr(const Fr& v, int d) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], d);
    return r;
}
__device__ __forceinline__ Fr shfl_fr(const Fr& v, int src) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl(v.l[i], src);
    return r;
}

__device__ __forceinline__ void scan_totals_body(const Fr* tot_a, const Fr* tot_b, uint32_t nb, Fr* bpre_a, Fr* bsuf_b, Fr* inv_b,
                                                 Fr* total_out, uint32_t p, Fr* sh);
struct ScanTotals {          // what the last workgroup of a product does with the block totals (scan_totals_body)
    Fr *bpre_a, *bsuf_b, *inv_b, *total_out;
    uint32_t* done;          // per product: workgroups that have stored their totals (zero between launches)
    // non-null: the factors of product p are omega^r - points[p] (the denominators of the SHPLONK quotients), formed
    // here instead of being read from a and b
    const Fr *omegas, *points;
    uint32_t* stamp;         // with total_out, pinned host memory (null: none): stamp[p] = seq once total_out[p] is stored
    uint32_t seq;
};
// grid (n / (bs E), cnt), bs = blockDim.x = min(256, n / E).  Thread `slot` of product p owns elements [slot E, slot E + E):
// pre_a[p][slot] = product of a over the workgroup's elements before them, suf_b[p][slot] = product of b over the
// workgroup's elements after them
template <int E>
__global__ __launch_bounds__(SCAN_THREADS) void scan_local_kernel(const Fr* a, const Fr* b, uint32_t n, Fr* pre_a, Fr* suf_b,
                                                                  Fr* tot_a, Fr* tot_b, ScanTotals T) {
    __shared__ Fr wa[SCAN_THREADS / 64], wb[SCAN_THREADS / 64];
    __shared__ Fr sh_tot[SCAN_THREADS];
    __shared__ uint32_t is_last;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, bs = blockDim.x, nw = (bs + 63) >> 6;
    const uint32_t wl = bs < 64 ? bs : 64;                 // lanes of a wave that hold a thread
    const uint32_t slot = blockIdx.x * bs + t;
    Fr ia, sb;                             // the thread's own products, then inclusive prefix / suffix inside the wave
    if (T.omegas) {
        const Fr pt = ld(T.points + blockIdx.y);
        const Fr* om = T.omegas + (size_t)slot * E;
        ia = ld(om) - pt;
#pragma unroll
        for (int j = 1; j < E; j++) ia = ia * (ld(om + j) - pt);
        sb = ia;
    } else {
        const size_t e0 = (size_t)blockIdx.y * n + (size_t)slot * E;
        ia = ld(a + e0);
        sb = ld(b + e0);
#pragma unroll
        for (int j = 1; j < E; j++) {
            ia = ia * ld(a + e0 + j);
            sb = sb * ld(b + e0 + j);
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Fr ua = shfl_up_fr(ia, off), db = shfl_down_fr(sb, off);
        if ((int)lane >= off) ia = ia * ua;
        if (lane + off < wl) sb = sb * db;
    }
    if (lane == wl - 1) wa[wave] = ia;
    if (lane == 0) wb[wave] = sb;
    __syncthreads();
    Fr ea = shfl_up_fr(ia, 1), sx = shfl_down_fr(sb, 1);   // exclusive inside the wave
    if (lane == 0) ea = Fr::one();
    if (lane == wl - 1) sx = Fr::one();
    for (uint32_t w = 0; w < wave; w++) ea = ea * wa[w];
    for (uint32_t w = wave + 1; w < nw; w++) sx = sx * wb[w];
    const size_t th = (size_t)blockIdx.y * (n / E) + slot;
    st(pre_a + th, ea);
    st(suf_b + th, sx);
    if (t == 0) {
        const uint32_t nb = gridDim.x;
        Fr ta = wa[0], tb = wb[0];
        for (uint32_t w = 1; w < nw; w++) { ta = ta * wa[w]; tb = tb * wb[w]; }
        st(tot_a + (size_t)blockIdx.y * nb + blockIdx.x, ta);
        st(tot_b + (size_t)blockIdx.y * nb + blockIdx.x, tb);
        // the workgroup that stores a product's last totals scans them (and inverts their product): no second dispatch.
        // Device-scope fences on both sides of the count: the other workgroups may have run on other XCDs.
        __threadfence();
        is_last = atomicAdd(T.done + blockIdx.y, 1u) + 1 == nb;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (t == 0) T.done[blockIdx.y] = 0;
    scan_totals_body(tot_a, tot_b, gridDim.x, T.bpre_a, T.bsuf_b, T.inv_b, T.total_out, blockIdx.y, sh_tot);
    // (thread 0 stored the total: its own fence orders the stamp behind it)
    if (t == 0 && T.total_out && T.stamp) publish_stamp(T.stamp + blockIdx.y, T.seq);
}

// The same scans for a lone proof (one element per thread, nb <= 64 blocks of >= 128 threads), ONE DIRECTION PER WORKGROUP: grid (nb, cnt, 2),
// z = 0 the prefix products of a, z = 1 the suffix products of b.  A lane of scan_local_kernel multiplies twice per step
// (its prefix and its suffix); here the two chains run side by side on different workgroups -- ten dependent products
// before the block totals are out instead of twenty-one -- and the workgroup that stores a product's last total scans the
// totals on two waves, one per direction.
__global__ __launch_bounds__(SCAN_THREADS) void scan_split_kernel(const Fr* a, const Fr* b, uint32_t n, Fr* pre_a, Fr* suf_b,
                                                                  Fr* tot_a, Fr* tot_b, ScanTotals T) {
    __shared__ Fr wv[SCAN_THREADS / 64];
    __shared__ uint32_t is_last;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, bs = blockDim.x, nw = (bs + 63) >> 6;
    const uint32_t wl = bs < 64 ? bs : 64;                 // lanes of a wave that hold a thread
    const uint32_t slot = blockIdx.x * bs + t, p = blockIdx.y, nb = gridDim.x;
    const bool fwd = blockIdx.z == 0;
    Fr v;
    if (T.omegas) v = ld(T.omegas + slot) - ld(T.points + p);
    else v = ld((fwd ? a : b) + (size_t)p * n + slot);
    // inclusive scan inside the wave, towards higher lanes (prefix) or lower ones (suffix)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Fr o = fwd ? shfl_up_fr(v, off) : shfl_down_fr(v, off);
        if (fwd ? (int)lane >= off : lane + off < wl) v = v * o;
    }
    if (lane == (fwd ? wl - 1 : 0)) wv[wave] = v;
    __syncthreads();
    // the waves before (after) this one, then the thread's inclusive product over the workgroup
    Fr carry = Fr::one();
    bool any = false;
    for (uint32_t w = 0; w < nw; w++) {
        if (fwd ? w < wave : w > wave) {
            carry = any ? carry * wv[w] : wv[w];
            any = true;
        }
    }
    if (any) v = v * carry;
    Fr ex = fwd ? shfl_up_fr(v, 1) : shfl_down_fr(v, 1);  // exclusive
    if (lane == (fwd ? 0 : wl - 1)) ex = carry;
    st((fwd ? pre_a : suf_b) + (size_t)p * n + slot, ex);
    if (t == (fwd ? bs - 1 : 0)) {                         // this thread's inclusive product is the workgroup's total
        st((fwd ? tot_a : tot_b) + (size_t)p * nb + blockIdx.x, v);
        __threadfence();
        is_last = atomicAdd(T.done + p, 1u) + 1 == 2 * nb;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (t == 0) T.done[p] = 0;
    if (t >= 128) return;
    // totals: wave 0 the exclusive prefixes of tot_a, wave 1 the exclusive suffixes of tot_b and the product of all b
    const bool tf = wave == 0;
    Fr x = lane < nb ? ld((tf ? tot_a : tot_b) + (size_t)p * nb + lane) : Fr::one();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Fr o = tf ? shfl_up_fr(x, off) : shfl_down_fr(x, off);
        if (tf ? (int)lane >= off : lane + off < 64) x = x * o;
    }
    Fr e = tf ? shfl_up_fr(x, 1) : shfl_down_fr(x, 1);
    if (lane == (tf ? 0u : 63u)) e = Fr::one();
    if (lane < nb) st((tf ? T.bpre_a : T.bsuf_b) + (size_t)p * nb + lane, e);
    if (!tf && lane == 0) {                                // lane 0's inclusive suffix = the product of all b
        if (T.total_out) {
            st(T.total_out + p, x);
            if (T.stamp) publish_stamp(T.stamp + p, T.seq);
        } else {
            st(T.inv_b + p, Fr::inv(x));
        }
    }
}

// The threads of one workgroup (256, or fewer when the product has one block) for product p, nb <= 256 block totals.
// bpre_a[k] = prod_{j<k} tot_a[j]; bsuf_b[k] = prod_{j>k} tot_b[j]; inv_b = 1 / prod_j tot_b[j]
// total_out (pinned host memory) non-null: the product of all b goes there instead of its inverse into inv_b --
// a lone proof inverts it on the host (3 us + a stream round trip) rather than on one GPU lane (55 us)
__device__ __forceinline__ void scan_totals_body(const Fr* tot_a, const Fr* tot_b, uint32_t nb, Fr* bpre_a, Fr* bsuf_b, Fr* inv_b,
                                                 Fr* total_out, uint32_t p, Fr* sh /* [SCAN_THREADS] */) {
    const uint32_t t = threadIdx.x;
    if (nb <= 64) {
        // the usual case (n <= 2^14): one wave, both scans by shuffles in one pass, no barriers
        if (t >= 64) return;
        const uint32_t wl = blockDim.x < 64 ? blockDim.x : 64;   // nb <= wl: a product with several blocks has full ones
        const Fr xa = t < nb ? ld(tot_a + (size_t)p * nb + t) : Fr::one();
        const Fr xb = t < nb ? ld(tot_b + (size_t)p * nb + t) : Fr::one();
        Fr ia = xa, sb = xb;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            Fr ua = shfl_up_fr(ia, off), db = shfl_down_fr(sb, off);
            if ((int)t >= off) ia = ia * ua;
            if (t + off < wl) sb = sb * db;
        }
        Fr ea = shfl_up_fr(ia, 1), eb = shfl_down_fr(sb, 1);
        if (t == 0) ea = Fr::one();
        if (t == wl - 1) eb = Fr::one();
        if (t < nb) {
            st(bpre_a + (size_t)p * nb + t, ea);
            st(bsuf_b + (size_t)p * nb + t, eb);
        }
        if (t == 0) {                                       // lane 0's inclusive suffix = the product of all b
            if (total_out) st(total_out + p, sb);
            else st(inv_b + p, Fr::inv(sb));
        }
        return;
    }
    const Fr va = t < nb ? ld(tot_a + (size_t)p * nb + t) : Fr::one();
    const Fr vb = t < nb ? ld(tot_b + (size_t)p * nb + t) : Fr::one();
    Fr ea = block_scan_fwd_excl(va, sh, t, SCAN_THREADS);
    Fr eb = block_scan_bwd_excl(vb, sh, t, SCAN_THREADS);
    if (t < nb) {
        st(bpre_a + (size_t)p * nb + t, ea);
        st(bsuf_b + (size_t)p * nb + t, eb);
    }
    if (t == 0) {
        if (total_out) st(total_out + p, vb * eb);
        else st(inv_b + p, Fr::inv(vb * eb));
    }
}

// inverses of the totals computed on the host travel as kernel arguments (n = 0: read inv_* from device memory)
constexpr uint32_t INV_LIST_MAX = 8;
struct InvList {
    uint32_t n;
    Fr v[INV_LIST_MAX];
};

// z[p][i] = c_p * prod_{r<i} num_r / prod_{r<i} den_r for i <= usable, blinding rows after it.
// Products [0, n_sets) are the chained permutation sets (c_p = prod_{j<p} z_j[usable]); the rest
// (lookups) start at 1.  Launched like scan_local_kernel<E>, whose outputs pre_num / suf_den / bpre / bsuf are.
// 1 / prod_{r<i} den_r is formed as (prod_{r>=i} den_r) / (prod of all den): no element is inverted.
template <int E>
__global__ __launch_bounds__(SCAN_THREADS) void grand_product_apply_kernel(const Fr* num, const Fr* den, const Fr* pre_num,
                                                                          const Fr* suf_den, const Fr* bpre, const Fr* bsuf,
                                                                          const Fr* inv_den, uint32_t n, uint32_t n_sets,
                                                                          uint32_t usable, uint32_t bf, Fr* zbase, size_t zstride,
                                                                          const Fr* blinds, InvList il) {
    __shared__ Fr chain;                   // c_p / (product of all den_p)
    const uint32_t p = blockIdx.y, bs = blockDim.x, tpp = n / E, nb = gridDim.x;
    auto inv_of = [&](uint32_t q) { return il.n ? il.v[q] : ld(inv_den + q); };
    if (threadIdx.x == 0) {
        Fr c = inv_of(p);
        if (p < n_sets) {
            const uint32_t su = usable / E, ju = usable % E, ku = su / bs;
            for (uint32_t q = 0; q < p; q++) {             // z_q[usable] from the chunk that holds row `usable`
                const size_t thq = (size_t)q * tpp + su, eq = (size_t)q * n + (size_t)su * E;
                Fr up = ld(pre_num + thq) * ld(bpre + (size_t)q * nb + ku);
                for (uint32_t j = 0; j < ju; j++) up = up * ld(num + eq + j);
                Fr dn = ld(suf_den + thq) * ld(bsuf + (size_t)q * nb + ku);
                for (uint32_t j = ju; j < (uint32_t)E; j++) dn = dn * ld(den + eq + j);
                c = c * (up * dn * inv_of(q));
            }
        }
        chain = c;
    }
    __syncthreads();
    const uint32_t slot = blockIdx.x * bs + threadIdx.x, i0 = slot * E;
    Fr* z = zbase + (size_t)p * zstride;
    const Fr* bl = blinds + (size_t)p * bf;
    if (i0 > usable) {
#pragma unroll
        for (int j = 0; j < E; j++) st(z + i0 + j, ld(bl + (i0 + j - usable - 1)));
        return;
    }
    const size_t th = (size_t)p * tpp + slot, k = (size_t)p * nb + blockIdx.x, e0 = (size_t)p * n + i0;
    Fr sf[E];                              // sf[j] = product of den from element j of the chunk to the end of the vector
    sf[E - 1] = ld(den + e0 + E - 1) * (ld(suf_den + th) * ld(bsuf + k));
#pragma unroll
    for (int j = E - 2; j >= 0; j--) sf[j] = ld(den + e0 + j) * sf[j + 1];
    Fr pf = ld(pre_num + th) * ld(bpre + k) * chain;
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t i = i0 + j;
        st(z + i, i > usable ? ld(bl + (i - usable - 1)) : pf * sf[j]);
        if (j + 1 < E) pf = pf * ld(num + e0 + j);
    }
}

// out[p][i] = 1 / v[p][i] from the scans of v against itself; v = in, or omega^r - points[p] when omegas is non-null
// (as in scan_local_kernel).  Launched like scan_local_kernel<E>.
template <int E>
__global__ __launch_bounds__(SCAN_THREADS) void batch_invert_apply_kernel(const Fr* in, const Fr* omegas, const Fr* points,
                                                                         const Fr* pre, const Fr* suf, const Fr* bpre,
                                                                         const Fr* bsuf, const Fr* inv_tot, uint32_t n, Fr* out,
                                                                         InvList il) {
    const uint32_t p = blockIdx.y, bs = blockDim.x, nb = gridDim.x;
    const uint32_t slot = blockIdx.x * bs + threadIdx.x;
    const size_t th = (size_t)p * (n / E) + slot, k = (size_t)p * nb + blockIdx.x, e0 = (size_t)p * n + (size_t)slot * E;
    Fr v[E];
    if (omegas) {
        const Fr pt = ld(points + p);
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = ld(omegas + (size_t)slot * E + j) - pt;
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = ld(in + e0 + j);
    }
    Fr sf[E];                              // product of every element after element j of the chunk
    sf[E - 1] = ld(suf + th) * ld(bsuf + k);
#pragma unroll
    for (int j = E - 2; j >= 0; j--) sf[j] = sf[j + 1] * v[j + 1];
    Fr pf = ld(pre + th) * ld(bpre + k) * (il.n ? il.v[p] : ld(inv_tot + p));   // (product of every element before) / total
#pragma unroll
    for (int j = 0; j < E; j++) {
        st(out + e0 + j, pf * sf[j]);
        if (j + 1 < E) pf = pf * v[j];
    }
}

// ------------------------------------------------------------------ lookup permutation
// (halo2 lookup::prover::permute_expression_pair) -- all on device:
// rank sort of the usable rows (n^2 / 2 comparisons are cheap at n = 2^13), then flags,
// prefix sums and binary searches in one workgroup.
// a < b as 256-bit integers: the borrow of a - b (one carry chain, no branches)
__device__ __forceinline__ bool raw_less(const Fr& a, const Fr& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fr d;
    return Fr::raw_sub_borrow(d, a, b) != 0;
#else
    return Fr::raw_lt(a, b);
#endif
}
__device__ __forceinline__ int raw_cmp(const Fr& a, const Fr& b) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] < b.l[i]) return -1;
        if (a.l[i] > b.l[i]) return 1;
    }
    return 0;
}

__global__ void to_raw_kernel(const Fr* in, Fr* out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) st(out + i, ld(in + i).to_raw());
}

// Stable sort of the usable rows by (value, row): tiles of 1024 elements are sorted in LDS (bitonic,
// [limb][element] layout), then every element finds its rank as the sum over the tiles of a binary
// search (10 steps per tile) -- 80 comparisons per element where the all-pairs rank sort needed 8192
// (that kernel was 10 % of the prover's vector instructions).
constexpr uint32_t SORT_TILE = 1024;

// (a, ia) < (b, ib) for raw 256-bit values with the row index as tie break
__device__ __forceinline__ bool key_less(const Fr& a, uint32_t ia, const Fr& b, uint32_t ib) {
    return raw_less(a, b) || (!raw_less(b, a) && ia < ib);
}

// blockIdx.x = tile, blockIdx.y = array; 512 threads, one compare-exchange each per step
__global__ __launch_bounds__(512) void sort_tiles_kernel(const Fr* raw, Fr* tile_sorted, uint32_t* tile_idx, uint32_t u, size_t stride) {
    __shared__ uint32_t val[8][SORT_TILE];
    __shared__ uint32_t idx[SORT_TILE];
    const uint32_t t = threadIdx.x, base = blockIdx.x * SORT_TILE;
    const Fr* src = raw + (size_t)blockIdx.y * stride;
    for (uint32_t e = t; e < SORT_TILE; e += 512) {
        const uint32_t i = base + e;
        Fr v;
        if (i < u) v = ld(src + i);
        else {
#pragma unroll
            for (int l = 0; l < 8; l++) v.l[l] = 0xffffffffu;          // padding sorts to the end
        }
#pragma unroll
        for (int l = 0; l < 8; l++) val[l][e] = v.l[l];
        idx[e] = i < u ? i : 0xffffffffu;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= SORT_TILE; k <<= 1) {
        for (uint32_t j = k >> 1; j >= 1; j >>= 1) {
            const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // element without bit j
            const uint32_t hi = lo | j;
            const bool asc = (lo & k) == 0;
            Fr a, b;
#pragma unroll
            for (int l = 0; l < 8; l++) { a.l[l] = val[l][lo]; b.l[l] = val[l][hi]; }
            const uint32_t ia = idx[lo], ib = idx[hi];
            const bool swap = asc ? key_less(b, ib, a, ia) : key_less(a, ia, b, ib);
            if (swap) {
#pragma unroll
                for (int l = 0; l < 8; l++) { val[l][lo] = b.l[l]; val[l][hi] = a.l[l]; }
                idx[lo] = ib;
                idx[hi] = ia;
            }
            __syncthreads();
        }
    }
    Fr* dst = tile_sorted + (size_t)blockIdx.y * stride;
    uint32_t* di = tile_idx + (size_t)blockIdx.y * stride;
    for (uint32_t e = t; e < SORT_TILE; e += 512) {
        const uint32_t i = base + e;
        if (i < u) {
            Fr v;
#pragma unroll
            for (int l = 0; l < 8; l++) v.l[l] = val[l][e];
            st(dst + i, v);
            di[i] = idx[e];
        }
    }
}

// sorted[rank(i)] = raw[i], rank(i) = number of (value, row) keys below (raw[i], i) over all sorted tiles.
// G = 2^g >= number of tiles consecutive lanes share one element, one tile each (ten dependent loads
// instead of eighty), and add their counts with a shuffle butterfly.
__global__ __launch_bounds__(256) void rank_from_tiles_kernel(const Fr* raw, const Fr* tile_sorted, const uint32_t* tile_idx,
                                                              Fr* sorted, uint32_t u, size_t stride, uint32_t G) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = tid / G, tile = tid % G;
    const size_t o = (size_t)blockIdx.y * stride;
    const uint32_t base = tile * SORT_TILE;
    uint32_t lo = 0;
    Fr x = Fr::zero();
    if (i < u) {
        x = ld(raw + o + i);
        if (base < u) {
            uint32_t hi = u - base < SORT_TILE ? u - base : SORT_TILE;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (key_less(ld(tile_sorted + o + base + mid), tile_idx[o + base + mid], x, i)) lo = mid + 1;
                else hi = mid;
            }
        }
    }
    for (uint32_t m = G >> 1; m >= 1; m >>= 1) lo += __shfl_xor(lo, (int)m);
    if (i < u && tile == 0) st(sorted + o + lo, x);
}

__device__ __forceinline__ bool bsearch_raw(const Fr* arr, uint32_t count, const Fr& key) {
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        int c = raw_cmp(ld(arr + mid), key);
        if (c == 0) return true;
        if (c < 0) lo = mid + 1;
        else hi = mid;
    }
    return false;
}

// exclusive prefix sums of TWO counters per thread over the workgroup: shuffles inside a wave, one LDS step across the
// waves -- three barriers where a Hillis-Steele scan in LDS takes twenty per counter (on a 1024-thread workgroup a
// barrier is what the time goes into)
__device__ __forceinline__ void block_scan2_u32_excl(uint32_t& a, uint32_t& b, uint32_t* sh /* [>= 2 * waves + 2] */, uint32_t t,
                                                     uint32_t nt, uint32_t* total_a, uint32_t* total_b) {
    const uint32_t lane = t & 63, wave = t >> 6, waves = (nt + 63) >> 6;
    uint32_t ia = a, ib = b;                                   // inclusive scans inside the wave
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t ua = __shfl_up(ia, off), ub = __shfl_up(ib, off);
        if (lane >= off) { ia += ua; ib += ub; }
    }
    if (lane == 63 || t == nt - 1) { sh[2 * wave] = ia; sh[2 * wave + 1] = ib; }
    __syncthreads();
    if (t == 0) {                                              // at most sixteen waves: a serial pass
        uint32_t ra = 0, rb = 0;
        for (uint32_t w = 0; w < waves; w++) {
            const uint32_t xa = sh[2 * w], xb = sh[2 * w + 1];
            sh[2 * w] = ra; sh[2 * w + 1] = rb;
            ra += xa; rb += xb;
        }
        sh[2 * waves] = ra; sh[2 * waves + 1] = rb;
    }
    __syncthreads();
    const uint32_t base_a = sh[2 * wave], base_b = sh[2 * wave + 1];
    *total_a = sh[2 * waves];
    *total_b = sh[2 * waves + 1];
    a = base_a + ia - a;
    b = base_b + ib - b;
    __syncthreads();
}

// a = sorted input (raw), tb = sorted table (raw); writes permuted input / table (Montgomery)
// rows [0,u) and the blinding rows [u,n) of both; err[l] != 0 when the input of lookup l is not in the table (one plain
// store per workgroup: `err` and `blinds` may live in pinned host memory, read and written over the host link).
// One workgroup per lookup (the last one of lookup_flags_kernel): only prefix sums over the flags and the scatter are left here.
constexpr uint32_t LOOKUP_REP_LDS = 8192;
struct LookupPermuteArgs {
    Fr *pin_all, *ptab_all;
    size_t pstride;
    uint32_t* scratch_all;
    const Fr* blinds;
    uint32_t* err;
    uint32_t* done;          // per lookup: flag workgroups finished (zero between launches)
    Fr* tmont_all;           // scratch, lookup l at l * tmont_stride: the sorted table in Montgomery form, written by the flag
    size_t tmont_stride;     // pass of every workgroup so that the one permuting workgroup only copies
};
__device__ __forceinline__ void lookup_permute_body(const Fr* tmont, uint32_t u, uint32_t n,
                                                    Fr* pin_all, Fr* ptab_all, size_t pstride, uint32_t* scratch_all,
                                                    const uint32_t* flags_all, const Fr* blinds, uint32_t* err, uint32_t l,
                                                    uint32_t* sh /* [1024] */, uint32_t* rep_lds /* [LOOKUP_REP_LDS] */) {
    const uint32_t t = threadIdx.x, nt = blockDim.x;
    Fr* pin = pin_all + (size_t)l * pstride;
    Fr* ptab = ptab_all + (size_t)l * pstride;
    // rows of repeated inputs, ascending: in LDS when they fit (a round trip through global memory otherwise)
    uint32_t* rep = u <= LOOKUP_REP_LDS ? rep_lds : scratch_all + (size_t)l * n;
    const uint32_t* fl = flags_all + (size_t)l * n;
    const uint32_t per = (u + nt - 1) / nt;
    const uint32_t r0 = t * per < u ? t * per : u, r1 = r0 + per < u ? r0 + per : u;
    // a thread's flags packed four bits apiece: read once (up to sixteen rows a thread; beyond that the loops reload)
    uint64_t packed = 0;
    const bool cached = per <= 16;
    uint32_t nrep = 0, nkeep = 0, bad = 0;
    for (uint32_t i = r0; i < r1; i++) {
        const uint32_t f = fl[i];
        if (cached) packed |= (uint64_t)(f & 15u) << (4 * (i - r0));
        nrep += (f & 1u) ? 0u : 1u;
        bad += (f >> 1) & 1u;
        nkeep += (f >> 2) & 1u;
    }
    auto flag_of = [&](uint32_t i) -> uint32_t { return cached ? (uint32_t)(packed >> (4 * (i - r0))) & 15u : fl[i]; };
    const int any_bad = __syncthreads_or(bad != 0);
    uint32_t tot_rep, tot_keep;
    uint32_t rep_off = nrep, keep_off = nkeep;
    block_scan2_u32_excl(rep_off, keep_off, sh, t, nt, &tot_rep, &tot_keep);
    if (t == 0) err[l] = (any_bad ? 1u : 0u) | (tot_rep != tot_keep ? 2u : 0u);
    // list the repeated rows (the permuted input, and the table entries of first occurrences, were written by the flag
    // pass of every workgroup: only what needs the prefix sums is left to this one)
    uint32_t q = rep_off;
    for (uint32_t i = r0; i < r1; i++)
        if (!(flag_of(i) & 1u)) rep[q++] = i;
    __threadfence_block();
    __syncthreads();
    // leftover table entries (ascending) go to the repeated rows taken from the END
    q = keep_off;
    for (uint32_t i = r0; i < r1; i++) {
        if (flag_of(i) & 4u) {
            if (q < tot_rep) st(ptab + rep[tot_rep - 1 - q], ld(tmont + i));
            q++;
        }
    }
    const uint32_t nb = n - u;
    if (t < nb) {
        st(pin + u + t, ld(blinds + (size_t)(2 * l) * nb + t));
        st(ptab + u + t, ld(blinds + (size_t)(2 * l + 1) * nb + t));
    }
}

// flags of the sorted input a / sorted table tb (raw), one lane per usable row, whole chip:
//   bit 0  a[i] is the first of its run          bit 1  that first occurrence is missing from the table
//   bit 2  tb[i] stays in the leftover list (it is not the first of a run that the input also holds)
// grid (ceil(u / blockDim), L), blockDim = min(1024, n).  The workgroup that finishes a lookup's flags last goes on with
// its permutation (lookup_permute_body: prefix sums over the flags and the scatter, one workgroup's work) -- no second
// dispatch; device-scope fences on both sides of the count, as the flags come from workgroups on other XCDs.
__global__ __launch_bounds__(1024) void lookup_flags_kernel(const Fr* a_all, const Fr* tb_all, size_t sstride, uint32_t u,
                                                            uint32_t n, uint32_t* flags_all, LookupPermuteArgs pa) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t rep_lds[LOOKUP_REP_LDS];
    __shared__ uint32_t is_last;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (i < u) {
        const Fr* a = a_all + (size_t)l * sstride;
        const Fr* tb = tb_all + (size_t)l * sstride;
        const Fr ai = ld(a + i), ti = ld(tb + i);
        uint32_t f = 0;
        const bool first = i == 0 || raw_cmp(ld(a + i - 1), ai) != 0;
        if (first) {
            f |= 1u;
            if (!bsearch_raw(tb, u, ai)) f |= 2u;
        }
        const bool tfirst = i == 0 || raw_cmp(ld(tb + i - 1), ti) != 0;
        if (!(tfirst && bsearch_raw(a, u, ti))) f |= 4u;
        flags_all[(size_t)l * n + i] = f;
        // the permuted input is the sorted input; where a value occurs first, the permuted table repeats it
        const Fr am = Fr::from_raw(ai);
        st(pa.pin_all + (size_t)l * pa.pstride + i, am);
        if (f & 1u) st(pa.ptab_all + (size_t)l * pa.pstride + i, am);
        if (f & 4u) st(pa.tmont_all + (size_t)l * pa.tmont_stride + i, Fr::from_raw(ti));
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(pa.done + l, 1u) + 1 == gridDim.x;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (threadIdx.x == 0) pa.done[l] = 0;
    lookup_permute_body(pa.tmont_all + (size_t)l * pa.tmont_stride, u, n, pa.pin_all, pa.ptab_all, pa.pstride, pa.scratch_all, flags_all, pa.blinds,
                        pa.err, l, sh, rep_lds);
}

// ------------------------------------------------------------------ quotient terms (extended domain)
struct PermQArgs {
    PermArgs p;                 // col = cosets, sigma = sigma cosets, xs = the points c_j * omega^t
    const Fr* pz;               // [n_sets][stride]
    const Fr* l0; const Fr* l_last; const Fr* l_active;
    uint32_t n_sets, mask;      // mask = n - 1: rotations move inside a coset (index = j * n + t)
    size_t stride;
    int32_t rot_last;
    Fr y;
    Fr* h;
};

struct LookupQArgs {
    const Fr* z; const Fr* pin; const Fr* ptab;       // cosets
    const Fr* ain; const Fr* stab;                     // theta-compressed input / table on the coset
    const Fr* l0; const Fr* l_last; const Fr* l_active;
    uint32_t mask;                                     // n - 1 (coset-major quotient domain)
    Fr y, beta, gamma;
    Fr* h;
};

// the five lookup terms folded into h at row i
__device__ __forceinline__ Fr lookup_quotient_terms(Fr h, const LookupQArgs& a, uint32_t i) {
    const Fr l0 = ld(a.l0 + i), ll = ld(a.l_last + i), la = ld(a.l_active + i);
    const Fr z = ld(a.z + i), ap = ld(a.pin + i), sp = ld(a.ptab + i);
    const uint32_t base = i & ~a.mask, inext = base | ((i + 1) & a.mask), iprev = base | ((i - 1) & a.mask);
    h = h * a.y + l0 * (Fr::one() - z);
    h = h * a.y + ll * (Fr::sqr(z) - z);
    Fr lhs = ld(a.z + inext) * (ap + a.beta) * (sp + a.gamma);
    Fr rhs = z * (ld(a.ain + i) + a.beta) * (ld(a.stab + i) + a.gamma);
    h = h * a.y + la * (lhs - rhs);
    Fr d = ap - sp;
    h = h * a.y + l0 * d;
    h = h * a.y + la * d * (ap - ld(a.pin + iprev));
    return h;
}

__global__ void lookup_quotient_kernel(LookupQArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    st(a.h + i, lookup_quotient_terms(ld(a.h + i), a, i));
}

// with_lookup: the terms of the first lookup follow in the same pass (h stays in registers, one dispatch fewer)
__global__ __launch_bounds__(256, 2) void perm_quotient_kernel(PermQArgs a, LookupQArgs lq, uint32_t with_lookup) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ne = a.stride;
    const uint32_t base = i & ~a.mask;
    Fr h = ld(a.h + i);
    const Fr l0 = ld(a.l0 + i), ll = ld(a.l_last + i), la = ld(a.l_active + i);
    const Fr one = Fr::one();
    h = h * a.y + l0 * (one - ld(a.pz + i));
    Fr zl = ld(a.pz + (size_t)(a.n_sets - 1) * ne + i);
    h = h * a.y + ll * (Fr::sqr(zl) - zl);
    const uint32_t ilast = base | ((i + (uint32_t)a.rot_last) & a.mask);
    for (uint32_t s = 1; s < a.n_sets; s++)
        h = h * a.y + l0 * (ld(a.pz + (size_t)s * ne + i) - ld(a.pz + (size_t)(s - 1) * ne + ilast));
    const uint32_t inext = base | ((i + 1) & a.mask);
    const Fr x = ld(a.p.xs + i);
    for (uint32_t s = 0; s < a.n_sets; s++) {
        Fr left = ld(a.pz + (size_t)s * ne + inext), right = ld(a.pz + (size_t)s * ne + i);
        uint32_t lo = s * a.p.chunk, hi = lo + a.p.chunk < a.p.n_cols ? lo + a.p.chunk : a.p.n_cols;
        for (uint32_t c = lo; c < hi; c++) {
            Fr v = ld(a.p.col[c] + i);
            left = left * (a.p.beta * ld(a.p.sigma + (size_t)c * a.p.sigma_stride + i) + v + a.p.gamma);
            right = right * (a.p.dbeta[c] * x + v + a.p.gamma);
        }
        h = h * a.y + la * (left - right);
    }
    if (with_lookup) h = lookup_quotient_terms(h, lq, i);
    st(a.h + i, h);
}

// Quotient pieces from the per-coset interpolants.  The inverse transform of h on coset c_j * H gives c_j^t * sum_m
// h_m[t] * (c_j^n)^m with h = sum_m X^(m n) h_m; its final store multiplies by c_j^-t (NttOpts::out_table), so q_j[t] here
// is sum_m h_m[t] * (c_j^n)^m and every coefficient index t is one nc x nc Vandermonde solve (matrix inverse precomputed
// at keygen): pieces[m][t] = sum_j vinv[m][j] * q_j[t]
constexpr uint32_t MAX_COSETS = 16;
// grid (n / block, nc): one thread per (piece m, coefficient t) -- nc products deep instead of nc^2
__global__ void coset_combine_kernel(const Fr* q, const Fr* vinv, uint32_t nc, uint32_t n, Fr* pieces) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (t >= n) return;
    Fr acc = Fr::zero();
    for (uint32_t j = 0; j < nc; j++) acc = acc + ld(vinv + m * nc + j) * ld(q + (size_t)j * n + t);
    st(pieces + (size_t)m * n + t, acc);
}

// ------------------------------------------------------------------ evaluations
// pw[p][i] = point_p^i
// the row of blockIdx.y == n_points forms hx = sum_i x^(n i) piece_i instead (Horner over the quotient pieces): the two
// are independent and share a dispatch; the row behind it does the same over the pieces' values on the domain
// (pieces_lag -> hx_lag: the combination is linear, so hx needs no transform of its own)
// sq (bits > 0): sq.v[p][b] = point_p^(2^b), b < bits, squared out by the host and passed by value: a thread's first
// power is then the product over the set bits of its index -- at most `bits` dependent products, six or seven on
// average, where square-and-multiply from the point itself chains `bits` squarings and as many products
constexpr uint32_t POW_TABLE_POINTS = 8, POW_TABLE_BITS = 14;
struct PowTable { Fr v[POW_TABLE_POINTS][POW_TABLE_BITS]; };
__global__ void powers_kernel(const Fr* points, uint32_t n_points, uint32_t n, uint32_t per, Fr* pw, const Fr* pieces,
                              uint32_t n_pieces, Fr xn, Fr* hx, const Fr* pieces_lag, Fr* hx_lag, PowTable sq, uint32_t bits) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p = blockIdx.y;
    if (p >= n_points) {
        const Fr* src = p == n_points ? pieces : pieces_lag;
        Fr* dst = p == n_points ? hx : hx_lag;
        for (uint32_t i = t * per; i < (t + 1) * per && i < n; i++) {
            Fr acc = Fr::zero();
            for (int q = (int)n_pieces - 1; q >= 0; q--) acc = acc * xn + ld(src + (size_t)q * n + i);
            st(dst + i, acc);
        }
        return;
    }
    uint32_t i0 = t * per;
    if (i0 >= n) return;
    Fr x = ld(points + p);
    Fr acc;
    if (bits) {
        acc = Fr::one();
        for (uint32_t b = 0; b < bits; b++)
            if ((i0 >> b) & 1u) acc = acc * sq.v[p][b];
    } else {
        acc = Fr::pow_u64(x, i0);
    }
    Fr* dst = pw + (size_t)p * n;
    for (uint32_t j = 0; j < per && i0 + j < n; j++) {
        st(dst + i0 + j, acc);
        acc = acc * x;
    }
}

struct EvalJob { const Fr* poly; uint32_t point; uint32_t pad; };

// out[job] = sum_i poly[i] * pw[point][i]
// 1024 threads per evaluation: at n = 8192 eight products per thread and a shuffle + LDS reduction (a lone proof's
// 59 evaluations are 59 workgroups -- the per-thread chain is what the kernel's duration is made of)
constexpr int EVAL_THREADS = 1024;
// stamp (pinned host memory, null: none): stamp[job] = seq once out[job] is stored -- what a lone proof's host waits for
__global__ __launch_bounds__(EVAL_THREADS) void eval_kernel(const EvalJob* jobs, const Fr* pw, uint32_t n, Fr* out, uint32_t* stamp,
                                                            uint32_t seq) {
    __shared__ Fr sh[EVAL_THREADS / 64];
    const EvalJob jb = jobs[blockIdx.x];
    const Fr* w = pw + (size_t)jb.point * n;
    // four coefficient / power pairs requested before the first product (a thread's eight pairs at n = 8192 would
    // otherwise be eight memory latencies in a row)
    Fr acc = Fr::zero();
    uint32_t i = threadIdx.x;
    for (; i + 3 * EVAL_THREADS < n; i += 4 * EVAL_THREADS) {
        const Fr c0 = ld(jb.poly + i), c1 = ld(jb.poly + i + EVAL_THREADS), c2 = ld(jb.poly + i + 2 * EVAL_THREADS),
                 c3 = ld(jb.poly + i + 3 * EVAL_THREADS);
        const Fr p0 = ld(w + i), p1 = ld(w + i + EVAL_THREADS), p2 = ld(w + i + 2 * EVAL_THREADS), p3 = ld(w + i + 3 * EVAL_THREADS);
        acc = acc + c0 * p0;
        acc = acc + c1 * p1;
        acc = acc + c2 * p2;
        acc = acc + c3 * p3;
    }
    for (; i < n; i += EVAL_THREADS) acc = acc + ld(jb.poly + i) * ld(w + i);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + shfl_down_fr(acc, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = sh[0];
        for (int k = 1; k < EVAL_THREADS / 64; k++) s = s + sh[k];
        st(out + blockIdx.x, s);
        if (stamp) publish_stamp(stamp + blockIdx.x, seq);
    }
}

// ------------------------------------------------------------------ SHPLONK in evaluation form
// Every opened polynomial has degree < n, so the quotients (f_i - R_i) / Z_i and
// L / (diff_0 (X - mu)) are fixed by their values on the size-n domain: they are formed
// point-wise and committed with the Lagrange-basis SRS -- the same group elements as the
// textbook coefficient-form division, with no polynomial division and no extra NTT.
constexpr int MAX_SETS = 8, MAX_SET_POLYS = 96, MAX_SET_ROTS = 4, MAX_POINTS = 8;

// the polynomials of a rotation set (their values on the domain): fixed per proving lane, uploaded once
struct LincombArgs {
    const Fr* poly[MAX_SET_POLYS];
    uint32_t count;
};
// zeta^j, j < the largest set: the j-th polynomial of every set is scaled by zeta^j.  Passed by value (3 KB of kernel
// arguments) -- what changes from proof to proof reaches the kernels without an upload of its own
struct ZetaPowers { Fr v[MAX_SET_POLYS]; };
// grid (n / 64, sets), 64 * parts threads: wave w of a workgroup sums the polynomials j = w, w + parts, ... of its 64
// elements and wave 0 adds the partial sums up.  A set holds up to fifty polynomials: one thread per element is a chain
// of fifty products, which is what a lone proof waits for (parts = 4); a pool lane keeps one wave per 64 elements.
__global__ __launch_bounds__(256) void lincomb_kernel(const LincombArgs* args, ZetaPowers zp, uint32_t n, Fr* out) {
    __shared__ Fr part_sum[3][64];
    const uint32_t lane = threadIdx.x & 63, part = threadIdx.x >> 6, parts = blockDim.x >> 6;
    const uint32_t i = blockIdx.x * 64 + lane;
    const LincombArgs& a = args[blockIdx.y];
    Fr acc = Fr::zero();
    if (i < n)
        for (uint32_t j = part; j < a.count; j += parts) acc = acc + zp.v[j] * ld(a.poly[j] + i);
    if (parts > 1) {
        if (part) part_sum[part - 1][lane] = acc;
        __syncthreads();
        if (part) return;
        for (uint32_t p = 1; p < parts; p++) acc = acc + part_sum[p - 1][lane];
    }
    if (i < n) st(out + (size_t)blockIdx.y * n + i, acc);
}

// d[p][i] = omega^i - point_p
struct ShplonkArgs {
    uint32_t n_sets;
    uint32_t n_rots[MAX_SETS];
    uint32_t rot_pt[MAX_SETS][MAX_SET_ROTS];     // index of the set's points in `dinv`
    Fr rcoef[MAX_SETS][MAX_SET_ROTS];            // R_i coefficients, low to high
    Fr coef[MAX_SETS];                           // nu^i
    const Fr* f;                                 // [n_sets][n] combined set polynomials (values)
    const Fr* dinv;                              // [points][n] 1 / (omega^r - point)
    const Fr* omegas;                            // omega^r
};

// W(omega^r) = sum_i nu^i (f_i - R_i)(omega^r) / Z_i(omega^r); the arguments (1.5 KB) by value
__global__ void shplonk_w_kernel(ShplonkArgs a, uint32_t n, Fr* out) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    Fr acc = Fr::zero();
    for (uint32_t i = 0; i < a.n_sets; i++) {
        Fr rv = a.rcoef[i][0];
        for (uint32_t t = 1; t < a.n_rots[i]; t++) rv = rv + a.rcoef[i][t] * ld(a.omegas + (((uint64_t)r * t) & (n - 1)));
        Fr v = ld(a.f + (size_t)i * n + r) - rv;
        for (uint32_t t = 0; t < a.n_rots[i]; t++) v = v * ld(a.dinv + (size_t)a.rot_pt[i][t] * n + r);
        acc = acc + a.coef[i] * v;
    }
    st(out + r, acc);
}

// W'(omega^r) = (sum_i coef_i f_i(omega^r) - c0 - c1 W(omega^r)) / (omega^r - mu)
struct ShplonkWpArgs {                           // by value: a few hundred bytes of kernel arguments, no upload
    uint32_t n_sets;
    Fr coef[MAX_SETS];                           // nu^i diff_i / diff_0
    Fr c0, c1;                                   // sum_i coef_i R_i(mu), Z_T(mu) / diff_0
    const Fr* f;                                 // [n_sets][n] combined set polynomials (values)
    const Fr* w;                                 // W values
};
// 1 / (omega^r - mu) is formed on the way from the product scans of omega^r - mu (what batch_invert_apply_kernel<E> would
// store for product 0): no dispatch and no round trip of n values for the inverses.  Launched like scan_local_kernel<E>.
template <int E>
__global__ __launch_bounds__(SCAN_THREADS) void shplonk_wp_inv_kernel(ShplonkWpArgs a, const Fr* omegas, const Fr* mu, const Fr* pre,
                                                                      const Fr* suf, const Fr* bpre, const Fr* bsuf, const Fr* inv_tot,
                                                                      InvList il, uint32_t n, Fr* out) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.x, e0 = slot * E;
    const Fr pt = ld(mu);
    Fr v[E];
#pragma unroll
    for (int j = 0; j < E; j++) v[j] = ld(omegas + e0 + j) - pt;
    Fr sf[E];                              // product of every element after element j of the chunk
    sf[E - 1] = ld(suf + slot) * ld(bsuf + k);
#pragma unroll
    for (int j = E - 2; j >= 0; j--) sf[j] = sf[j + 1] * v[j + 1];
    Fr pf = ld(pre + slot) * ld(bpre + k) * (il.n ? il.v[0] : ld(inv_tot));   // (product of every element before) / total
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t r = e0 + j;
        Fr acc = Fr::zero();
        for (uint32_t i = 0; i < a.n_sets; i++) acc = acc + a.coef[i] * ld(a.f + (size_t)i * n + r);
        acc = acc - a.c0 - a.c1 * ld(a.w + r);
        st(out + r, acc * (pf * sf[j]));
        if (j + 1 < E) pf = pf * v[j];
    }
}

}  // namespace shp
