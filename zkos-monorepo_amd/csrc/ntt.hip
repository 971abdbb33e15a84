// Radix-2 NTT / iNTT over BN254 Fr for gfx950.
//
// Replaces halo2_proofs v0.3.0 EvaluationDomain::{best_fft, lagrange_to_coeff,
// coeff_to_extended, extended_to_coeff} on the proving path (keygen and the quotient phase;
// conventions pinned by /root/reference/resources/ppot_0080_11_raw's g_lagrange and the delta
// literal at /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:475).
//
// Design (HBM-bound shape, integer-ALU-bound in practice):
//   * N = L1 * L2 [* L3].  One pass per digit; a pass moves every element HBM -> LDS -> HBM
//     exactly once, so a transform costs P x 64 B/element of traffic with P in {1,2,3}.
//   * A workgroup owns a 512- or 1024-element LDS tile (two 16-byte planes so that
//     ds_read_b128/ds_write_b128 are lane-contiguous and bank-conflict free) holding 2^c
//     independent length-2^s sub-transforms whose elements are ADJACENT in memory: every
//     global access is a run of 2^c * 32 B >= 128 B.
//   * Inside the tile: s decimation-in-frequency stages, twiddles from a 32 KiB table of
//     w_2048^j that stays in L2; the bit reversal is folded into the read-out index.
//   * Between passes: the element is scaled by w_N^(k_p * lo), read from a table of all N powers for domains up to
//     2^24 (the extra 32 B per element is HBM traffic the transform has room for; a second product is not), and as
//     hi[e >> h] * lo[e & mask] from two small tables above that.
//   * The last pass writes in digit-reversed = natural order, 2^c adjacent outputs per run,
//     so there is no separate bit-reversal sweep.  1/n, and the zeta-coset scalings of
//     coeff_to_extended / extended_to_coeff, are fused into the first load / last store.
#include <stdlib.h>

#include "ctx.hpp"
#include "lazy29.hip.hpp"
#include "ntt_lazy.hip.hpp"

namespace shp {

// Tile sizes: 512 elements (16 KiB of LDS, 256 threads) for domains up to 2^18 -- two passes of at most nine stages, and
// nine or ten tiles per CU at different stages hide each other's barriers (the prover's 2^13 transforms: pool 1915 ->
// 1955 proofs/s, a lone proof 2.0 -> 1.9 ms against 2048-element tiles) -- and 1024 elements (512 threads) above, where
// the longer runs of adjacent elements pay (2^22: 0.63 ms against 0.645 / 0.65 for 512 / 2048; 2^24: 2.51 / 2.60 / 2.47).
constexpr int NTT_TILE_SMALL = 9, NTT_TILE_LARGE = 10;

struct PassParams {
    const Fr* src;
    Fr* dst;
    size_t src_stride, dst_stride;
    uint32_t log_n, s, lo_bits, c, s1, s2, h;
    uint32_t tw_shift;        // the twiddle tables hold powers of w_(2^(log_n + tw_shift)): exponents are shifted up
    uint32_t in_len;
    int is_first, is_final, pre_pattern, post_pattern, do_scale;
    const Fr* small_tw;
    const Fr* tw_lo;
    const Fr* tw_hi;
    const Fr* tw_pass;        // this pass's inter-pass twiddles in the order the store reads them: [kp][lo] (null: tw_lo / tw_hi)
    Fr scale, zeta, zeta2;
    const Fr* twist;          // first pass only (null: none)
    uint32_t src_div;         // first pass: transforms per input polynomial (>= 1)
    const Fr* out_table;      // final pass only (null: none): per-output factors [transform][2^log_n], in place of `scale`
    // ntt_dit_pass_kernel only
    uint32_t j0;              // stages done by the passes before this one
    const Fr* dit_tw;         // twiddles by stage: [2^j - 1 + k]; one table per coset when dit_cosets > 1
    uint32_t dit_cosets;      // transforms per input polynomial (1: the plain transform, whose k = 0 twiddles are one)
    int dit_plain;            // the table's k = 0 entries are one: those butterflies skip the product
    int dit_scaled;           // the transform's very first stage also multiplies its first operand by dit_g (2: dit_g = 32, by shifts)
    Fr dit_g;
};

__device__ __forceinline__ Fr lds_load(const uint4* lo, const uint4* hi, uint32_t i) {
    Fr r;
    uint4 a = lo[i], b = hi[i];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_store(uint4* lo, uint4* hi, uint32_t i, const Fr& v) {
    lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr gload(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void gstore(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// one butterfly per thread and stage: THREADS = half the tile
template <int THREADS, int TILE_LOG>
__global__ __launch_bounds__(THREADS) void ntt_pass_kernel(PassParams p) {
    __shared__ uint4 sh_lo[1 << TILE_LOG];
    __shared__ uint4 sh_hi[1 << TILE_LOG];
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t poly = p.is_first && p.src_div > 1 ? blockIdx.y / p.src_div : blockIdx.y;
    const Fr* src = p.src + (size_t)poly * p.src_stride;
    const Fr* twist = p.is_first && p.twist ? p.twist + ((size_t)(blockIdx.y - poly * p.src_div) << p.log_n) : nullptr;
    Fr* dst = p.dst + (size_t)blockIdx.y * p.dst_stride;
    const uint32_t s = p.s, c = p.c;
    const uint32_t L = 1u << s, V = 1u << c, E = L << c;

    // ---- tile coordinates
    uint32_t hi = 0, lo_base = 0, k1_base = 0, k2 = 0;
    if (!p.is_final) {
        uint32_t lo_tiles_log = p.lo_bits - c;
        hi = tile >> lo_tiles_log;
        lo_base = (tile & ((1u << lo_tiles_log) - 1)) << c;
    } else {
        k2 = tile & ((1u << p.s2) - 1);
        k1_base = (tile >> p.s2) << c;
    }

    // ---- load HBM -> LDS
    for (uint32_t e = tid; e < E; e += THREADS) {
        size_t gi;
        uint32_t li;
        if (!p.is_final) {
            uint32_t v = e & (V - 1), j = e >> c;
            gi = ((size_t)hi << (s + p.lo_bits)) | ((size_t)j << p.lo_bits) | (lo_base | v);
            li = e;    // j * V + v
        } else {
            uint32_t j = e & (L - 1), r = e >> s;
            uint32_t k1 = k1_base | r;
            gi = ((((size_t)k1 << p.s2) | k2) << s) | j;
            li = j * V + r;
        }
        Fr val;
        if (p.is_first && gi >= p.in_len) {
            val = Fr::zero();
        } else {
            val = gload(src + gi);
            if (p.is_first && p.pre_pattern) {
                uint32_t m = (uint32_t)(gi % 3);
                if (m == 1) val = val * p.zeta;
                else if (m == 2) val = val * p.zeta2;
            }
            if (twist) val = val * gload(twist + gi);
        }
        lds_store(sh_lo, sh_hi, li, val);
    }
    __syncthreads();

    // ---- s decimation-in-frequency stages in LDS, one per sweep.  (Two stages per sweep on a quad of elements in
    // registers -- half the LDS round trips and barriers -- was measured and lost: 134 VGPRs allow one tile per CU
    // instead of two, 2^24 in 2.88 ms instead of 2.5; capped at 128 VGPRs the quad spills to scratch.)
    for (int t = (int)s - 1; t >= 0; t--) {
        const uint32_t half = 1u << t;
        for (uint32_t b = tid; b < (E >> 1); b += THREADS) {
            uint32_t v = b & (V - 1), jb = b >> c;
            uint32_t grp = jb >> t, pos = jb & (half - 1);
            uint32_t j0 = (grp << (t + 1)) | pos;
            uint32_t i0 = j0 * V + v, i1 = (j0 + half) * V + v;
            Fr a = lds_load(sh_lo, sh_hi, i0);
            Fr bb = lds_load(sh_lo, sh_hi, i1);
            Fr u = a + bb;
            Fr d = a - bb;
            if (pos) d = d * gload(p.small_tw + ((size_t)pos << (10 - t)));
            lds_store(sh_lo, sh_hi, i0, u);
            lds_store(sh_lo, sh_hi, i1, d);
        }
        __syncthreads();
    }

    // ---- store LDS -> HBM (bit reversal folded into the LDS read index)
    const uint32_t hi_bits = p.log_n - s - p.lo_bits;
    for (uint32_t e = tid; e < E; e += THREADS) {
        uint32_t v = e & (V - 1), kp = e >> c;
        uint32_t j = s ? (__brev(kp) >> (32 - s)) : 0;
        Fr val = lds_load(sh_lo, sh_hi, j * V + v);
        size_t go;
        if (!p.is_final) {
            uint32_t lo = lo_base | v;
            go = ((size_t)hi << (s + p.lo_bits)) | ((size_t)kp << p.lo_bits) | lo;
            uint64_t et = (((uint64_t)kp * lo) << hi_bits) << p.tw_shift;
            if (p.tw_pass) {
                // the pass's own table: lanes with adjacent outputs read adjacent entries (whole lines, like the data)
                if (et) val = val * gload(p.tw_pass + (((size_t)kp << p.lo_bits) | lo));
            } else if (et) {
                Fr tw = gload(p.tw_lo + (et & ((1u << p.h) - 1)));
                if (p.h < p.log_n + p.tw_shift) tw = gload(p.tw_hi + (et >> p.h)) * tw;
                val = val * tw;
            }
        } else {
            uint32_t k1 = k1_base | v;
            go = (size_t)k1 + ((size_t)k2 << p.s1) + ((size_t)kp << (p.s1 + p.s2));
            if (p.out_table) val = val * gload(p.out_table + ((size_t)blockIdx.y << p.log_n) + go);
            else if (p.do_scale) val = val * p.scale;
            if (p.post_pattern) {
                uint32_t m = (uint32_t)(go % 3);
                if (m == 1) val = val * p.zeta2;
                else if (m == 2) val = val * p.zeta;
            }
        }
        gstore(dst + go, val);
    }
}

// ------------------------------------------------------------------ decimation in time, every factor in the butterflies
// The transform as the textbook in-place decimation in time, read in the index space of the INPUT: stage j (0 <= j <
// log_n) pairs the elements whose indices differ in bit log_n - 1 - j, (a, b) -> (a + t b, a - t b), and the twiddle t =
// w_(2^(j+1))^k depends on the j bits ABOVE that bit alone (k = those bits reversed); after the last stage position i holds
// output bitrev(i).  A pass takes the next s bits from the top: its tile is every element that agrees in the bits above
// (already done) and below (still to do), 2^c neighbours wide as in ntt_pass_kernel, so the passes need no transposition
// and -- the point -- NO product between them: the factor w_N^(k_p lo) that the decomposition N = L1 L2 pays per element
// at the pass boundary is what makes the later stages' twiddles differ from tile to tile here (they come from a table of
// all 2^log_n - 1 of them, a tile of the final pass reads the 2^s - 1 per column it needs, neighbouring columns
// neighbouring entries).  A coset c H costs nothing either: f(c w^k) = f_e(c^2 w^2k) + c w^k f_o(c^2 w^2k), so stage j's
// twiddles are the plain ones times c^(2^(log_n-1-j)) -- a table per coset (ntt_build_coset_table) instead of a product
// per input coefficient.  The number of products is the classical one either way (ntt_pass_kernel skips its trivial
// twiddles: 2.5 + 2 in the butterflies + 1 between the passes + 1 twist per element of a 2^13 coset transform, 6.5 here,
// where no twiddle of a coset is trivial); what goes is the twist's and the inter-pass factor's table loads and index
// arithmetic, and -- below -- two of a butterfly's three conditional subtractions.  A factor on every output (dit_g) is
// a factor on both operands of the first stage: table entry 0 carries it for the second, the first takes it at the tile
// load, by shifts when it is 2^5.
template <int THREADS, int TILE_LOG>
__global__ __launch_bounds__(THREADS) void ntt_dit_pass_kernel(PassParams p) {
    __shared__ uint4 sh_lo[1 << TILE_LOG];
    __shared__ uint4 sh_hi[1 << TILE_LOG];
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t poly = p.is_first ? blockIdx.y / p.dit_cosets : blockIdx.y;
    const Fr* src = p.src + (size_t)poly * p.src_stride;
    Fr* dst = p.dst + (size_t)blockIdx.y * p.dst_stride;
    const Fr* tw = p.dit_tw + ((size_t)(blockIdx.y % p.dit_cosets) << p.log_n);
    const uint32_t s = p.s, c = p.c, j0 = p.j0;
    const uint32_t L = 1u << s, V = 1u << c, E = L << c;
    const uint32_t lo_bits = p.lo_bits;                    // index bits below this pass's (0 in the final pass)

    // ---- tile coordinates.  Not final: one value `a` of the bits above, V adjacent values of the bits below.  Final: V
    // values of `a` that differ in their TOP c bits -- their outputs are neighbours in the bit-reversed (natural) order.
    uint32_t a_fix = 0, lo_base = 0;
    if (!p.is_final) {
        const uint32_t lo_tiles_log = lo_bits - c;
        a_fix = tile >> lo_tiles_log;
        lo_base = (tile & ((1u << lo_tiles_log) - 1)) << c;
    } else {
        a_fix = tile;                                       // the low j0 - c bits of a
    }
    auto brev = [](uint32_t x, uint32_t bits) -> uint32_t { return bits ? __brev(x) >> (32 - bits) : 0; };

    // ---- this thread's butterflies: one per stage (a tile has at most twice as many elements as the workgroup threads), in
    // column v_t; the twiddle of stage jj is entry 2^(j0+jj) - 1 + tw_k(jj) and is requested a stage ahead -- the first one before
    // the tile is loaded -- so that a table entry that comes from the L2 or beyond arrives behind a barrier, not in front
    // of the product
    const bool active = tid < (E >> 1);
    const uint32_t v_t = tid & (V - 1), jbb = tid >> c;
    const uint32_t a_t = p.is_final ? ((v_t << (j0 - c)) | a_fix) : a_fix;
    const uint32_t ra_t = brev(a_t, j0);
    auto tw_k = [&](uint32_t jj) -> uint32_t { return (brev(jbb >> (s - 1 - jj), jj) << j0) | ra_t; };
    Fr w = Fr::zero();
    if (active && s) w = gload(tw + (((size_t)1 << j0) - 1 + tw_k(0)));

    // ---- load HBM -> LDS, [element of the pass][column]
    for (uint32_t e = tid; e < E; e += THREADS) {
        size_t gi;
        uint32_t li;
        if (!p.is_final) {
            const uint32_t v = e & (V - 1), jb = e >> c;
            gi = ((size_t)a_fix << (s + lo_bits)) | ((size_t)jb << lo_bits) | (lo_base | v);
            li = e;
        } else {
            const uint32_t jb = e & (L - 1), v = e >> s;
            const uint32_t a = (v << (j0 - c)) | a_fix;
            gi = ((size_t)a << s) | jb;
            li = jb * V + v;
        }
        Fr val = gload(src + gi);
        // a factor on every output is a factor on both operands of the very first stage: its second operand finds it in
        // the twiddle table's first entry, its first operand -- the half of the inputs whose top index bit is clear --
        // takes it here
        if (p.dit_scaled && p.is_first && (li >> c) < (L >> 1)) val = p.dit_scaled == 2 ? fr_times32(val) : val * p.dit_g;
        lds_store(sh_lo, sh_hi, li, val);
    }
    __syncthreads();

    // ---- s stages, one butterfly per thread and stage
    for (uint32_t jj = 0; jj < s; jj++) {
        const uint32_t t = s - 1 - jj, half = 1u << t, j = j0 + jj;
        if (active) {
            const uint32_t k = tw_k(jj);
            const uint32_t grp = jbb >> t, pos = jbb & (half - 1);
            const uint32_t jA = (grp << (t + 1)) | pos;
            const uint32_t i0 = jA * V + v_t, i1 = (jA + half) * V + v_t;
            // values are lazy, in [0, 4p): x comes down to [0, 2p), the product leaves y there without its last
            // subtraction, the sum and the difference (+ 2p) are below 4p again -- one conditional subtraction a
            // butterfly instead of three
            Fr y = lds_load(sh_lo, sh_hi, i1);
            if (k || !p.dit_plain) y = Fr::mul_lazy(y, w);
            else fr_reduce_2p(y);
            // the next stage's twiddle, into the registers the product has just finished with
            if (jj + 1 < s) w = gload(tw + (((size_t)1 << (j + 1)) - 1 + tw_k(jj + 1)));
            Fr x = lds_load(sh_lo, sh_hi, i0);
            fr_reduce_2p(x);
            lds_store(sh_lo, sh_hi, i0, fr_add_raw(x, y));
            lds_store(sh_lo, sh_hi, i1, fr_sub_2p(x, y));
        }
        __syncthreads();
    }

    // ---- store LDS -> HBM: in place between passes; bit-reversed = natural order, V adjacent outputs a run, at the end
    for (uint32_t e = tid; e < E; e += THREADS) {
        if (!p.is_final) {
            const uint32_t v = e & (V - 1), jb = e >> c;
            const size_t go = ((size_t)a_fix << (s + lo_bits)) | ((size_t)jb << lo_bits) | (lo_base | v);
            gstore(dst + go, lds_load(sh_lo, sh_hi, e));
        } else {
            const uint32_t vv = e & (V - 1), kp = e >> c;
            Fr val = lds_load(sh_lo, sh_hi, brev(kp, s) * V + brev(vv, c));
            const size_t go = ((size_t)kp << j0) | ((size_t)brev(a_fix, j0 - c) << c) | vv;
            // (a product takes the lazy value as it is and returns it reduced)
            if (p.out_table) val = val * gload(p.out_table + ((size_t)blockIdx.y << p.log_n) + go);
            else if (p.do_scale) val = val * p.scale;
            else { fr_reduce_2p(val); val.reduce_once(); }
            gstore(dst + go, val);
        }
    }
}

// one thread per table entry (see ntt_build_coset_table)
__global__ void ntt_dit_table_kernel(Fr c, Fr w, Fr g, uint32_t log_n, Fr* out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e + 1 >= ((size_t)1 << log_n)) {
        if (e + 1 == ((size_t)1 << log_n)) gstore(out + e, Fr::zero());   // the table's unused last element
        return;
    }
    const uint32_t j = 63 - __clzll((unsigned long long)(e + 1));
    const size_t k = e + 1 - ((size_t)1 << j);
    Fr acc = c, sq = w;
    for (size_t x = k; x; x >>= 1) {
        if (x & 1) acc = acc * sq;
        sq = Fr::sqr(sq);
    }
    for (uint32_t i = j + 1; i < log_n; i++) acc = Fr::sqr(acc);
    if (e == 0) acc = acc * g;
    gstore(out + e, acc);
}

// ------------------------------------------------------------------ the tile in 29-bit lazy arithmetic (SHP_NTT29=1)
// The same pass with the tile's arithmetic in nine unsaturated 29-bit limbs (lazy29.hip.hpp): a butterfly's product is
// 162 multiply-adds with no carry instruction, its sum nine additions and its difference eighteen, with no comparison
// against p anywhere inside the tile.  Elements stay in the Montgomery radix 2^256 of the rest of the library -- a value
// x 2^256 is split into limbs as the integer it is -- and every factor the kernel multiplies by (the twiddle tables,
// 1/n, the zeta pattern) is kept times 2^5, so that Lazy29's product a b 2^-261 leaves the radix where it was; a
// caller's twist table, which cannot be rescaled, is read shifted by five bits (an unreduced 32 t, allowed as the one
// lazy operand).
//
// Decimation in TIME inside the tile (bit-reversed placement at the load, natural read-out): u = a + w b, d = a - w b + 2p.
// The product reduces its input, so a value's bound grows by at most 2p per stage (22p after ten stages; decimation in
// frequency would double it on the sum side of every stage) and the pass's last product -- inter-pass twiddle, 1/n, or
// the Montgomery one -- brings it below 2p for the one conditional subtraction of the store.  Limbs: a product is
// normalised (limbs 0..7 below 2^29); a sum adds less than 2^29 to a limb, a difference less than 2^30 (2p is spread as
// kp_spread(2, 1)), so limbs stay below 2^32 for three stages: every third stage normalises what it stores, and where
// the product's lazy operand could exceed 2^31 (the third stage of such a group) it is normalised first.
using R29 = Lazy29<FrParams>;
using E29 = R29::El;

struct Pass29 {
    const uint4* small_a;      // w_2048^j 2^5: limbs 0..3
    const uint4* small_b;      //               limbs 4..7
    const uint32_t* small_c;   //               limb 8
    Fr zeta32, zeta2_32;       // first pass with pre_pattern: zeta 2^5, zeta^2 2^5
    Fr out32[3];               // final pass: the factor of output i, by i mod 3 when post_pattern, [0] otherwise, times 2^5
};

__device__ __forceinline__ E29 lds29_load(const uint4* lo, const uint4* hi, const uint32_t* top, uint32_t i) {
    const uint4 a = lo[i], b = hi[i];
    E29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = top[i];
    return r;
}
__device__ __forceinline__ void lds29_store(uint4* lo, uint4* hi, uint32_t* top, uint32_t i, const E29& v) {
    lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    top[i] = v.l[8];
}
// a normalised value below 2^256 as eight 32-bit words
__device__ __forceinline__ Fr pack29(const E29& t) {
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = (32 * j) / 29, sft = (32 * j) % 29;
        uint32_t v = t.l[i] >> sft;
        v |= t.l[i + 1] << (29 - sft);
        if (29 - sft + 29 < 32 && i + 2 < 9) v |= t.l[i + 2] << (58 - sft);
        r.l[j] = v;
    }
    return r;
}

template <int THREADS, int TILE_LOG>
__global__ __launch_bounds__(THREADS) void ntt29_pass_kernel(PassParams p, Pass29 q) {
    __shared__ uint4 sh_lo[1 << TILE_LOG];
    __shared__ uint4 sh_hi[1 << TILE_LOG];
    __shared__ uint32_t sh_top[1 << TILE_LOG];
    constexpr l29::L9 KP2 = R29::kp_spread(2, 1);
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t poly = p.is_first && p.src_div > 1 ? blockIdx.y / p.src_div : blockIdx.y;
    const Fr* src = p.src + (size_t)poly * p.src_stride;
    const Fr* twist = p.is_first && p.twist ? p.twist + ((size_t)(blockIdx.y - poly * p.src_div) << p.log_n) : nullptr;
    Fr* dst = p.dst + (size_t)blockIdx.y * p.dst_stride;
    const uint32_t s = p.s, c = p.c;
    const uint32_t L = 1u << s, V = 1u << c, E = L << c;

    uint32_t hi = 0, lo_base = 0, k1_base = 0, k2 = 0;
    if (!p.is_final) {
        uint32_t lo_tiles_log = p.lo_bits - c;
        hi = tile >> lo_tiles_log;
        lo_base = (tile & ((1u << lo_tiles_log) - 1)) << c;
    } else {
        k2 = tile & ((1u << p.s2) - 1);
        k1_base = (tile >> p.s2) << c;
    }

    // ---- load HBM -> limbs -> LDS, sub-transform index bit-reversed (value bound after this block: 1.2 p)
    for (uint32_t e = tid; e < E; e += THREADS) {
        size_t gi;
        uint32_t j, col;
        if (!p.is_final) {
            col = e & (V - 1); j = e >> c;
            gi = ((size_t)hi << (s + p.lo_bits)) | ((size_t)j << p.lo_bits) | (lo_base | col);
        } else {
            j = e & (L - 1); col = e >> s;
            uint32_t k1 = k1_base | col;
            gi = ((((size_t)k1 << p.s2) | k2) << s) | j;
        }
        const uint32_t li = (s ? (__brev(j) >> (32 - s)) : 0) * V + col;
        E29 val;
        if (p.is_first && gi >= p.in_len) {
#pragma unroll
            for (int i = 0; i < 9; i++) val.l[i] = 0;
        } else {
            val = R29::split<0>(gload(src + gi));
            if (p.is_first && p.pre_pattern) {
                uint32_t m = (uint32_t)(gi % 3);
                if (m == 1) val = R29::mul(val, R29::split<0>(q.zeta32));
                else if (m == 2) val = R29::mul(val, R29::split<0>(q.zeta2_32));
            }
            if (twist) val = R29::mul(val, R29::split<5>(gload(twist + gi)));
        }
        lds29_store(sh_lo, sh_hi, sh_top, li, val);
    }
    __syncthreads();

    // ---- s decimation-in-time stages.  (Requesting a thread's next twiddle one stage ahead, so that its latency passes
    // behind the sums, stores and barrier, was measured: more instructions, the same time.)
    for (uint32_t t = 0; t < s; t++) {
        const uint32_t half = 1u << t;
        const uint32_t phase = t % 3;                       // stages since the stored values were last normalised
        for (uint32_t b = tid; b < (E >> 1); b += THREADS) {
            uint32_t col = b & (V - 1), jb = b >> c;
            uint32_t grp = jb >> t, pos = jb & (half - 1);
            uint32_t j0 = (grp << (t + 1)) | pos;
            uint32_t i0 = j0 * V + col, i1 = (j0 + half) * V + col;
            E29 a = lds29_load(sh_lo, sh_hi, sh_top, i0);
            E29 bb = lds29_load(sh_lo, sh_hi, sh_top, i1);
            if (t) {
                // limbs below 2^29 + phase 2^30: at phase 2 beyond what the product's lazy operand may hold
                if (phase == 2) R29::normalise(bb);
                const uint32_t ti = pos << (10 - t);
                const uint4 wa = q.small_a[ti], wb = q.small_b[ti];
                E29 w;
                w.l[0] = wa.x; w.l[1] = wa.y; w.l[2] = wa.z; w.l[3] = wa.w;
                w.l[4] = wb.x; w.l[5] = wb.y; w.l[6] = wb.z; w.l[7] = wb.w;
                w.l[8] = q.small_c[ti];
                bb = R29::mul(bb, w);                       // pos = 0 multiplies by the Montgomery one: no divergent branch
            }
            E29 u, d;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                u.l[i] = a.l[i] + bb.l[i];
                d.l[i] = a.l[i] + KP2.l[i] - bb.l[i];
            }
            if (phase == 2) { R29::normalise(u); R29::normalise(d); }
            lds29_store(sh_lo, sh_hi, sh_top, i0, u);
            lds29_store(sh_lo, sh_hi, sh_top, i1, d);
        }
        __syncthreads();
    }

    // ---- LDS -> one reducing product -> eight words -> HBM (natural read-out)
    const uint32_t hi_bits = p.log_n - s - p.lo_bits;
    const bool lazy_limbs = s % 3 != 0;                     // the last stage did not normalise what it stored
    for (uint32_t e = tid; e < E; e += THREADS) {
        uint32_t col = e & (V - 1), kp = e >> c;
        E29 val = lds29_load(sh_lo, sh_hi, sh_top, kp * V + col);
        // limbs below 2^29 + 2 2^30: normalise unless they are known to be below 2^31
        if (lazy_limbs && s % 3 == 2) R29::normalise(val);
        size_t go;
        Fr f;
        if (!p.is_final) {
            uint32_t lo = lo_base | col;
            go = ((size_t)hi << (s + p.lo_bits)) | ((size_t)kp << p.lo_bits) | lo;
            uint64_t et = (((uint64_t)kp * lo) << hi_bits) << p.tw_shift;
            f = p.tw_pass ? gload(p.tw_pass + (((size_t)kp << p.lo_bits) | lo)) : gload(p.tw_lo + (et & ((1u << p.h) - 1)));
            if (!p.tw_pass && p.h < p.log_n + p.tw_shift) {
                const E29 twv = R29::mul(R29::split<0>(gload(p.tw_hi + (et >> p.h))), R29::split<0>(f));
                val = R29::mul(val, twv);
            } else {
                val = R29::mul(val, R29::split<0>(f));
            }
        } else {
            uint32_t k1 = k1_base | col;
            go = (size_t)k1 + ((size_t)k2 << p.s1) + ((size_t)kp << (p.s1 + p.s2));
            const uint32_t m = p.post_pattern ? (uint32_t)(go % 3) : 0;
            if (p.out_table) f = gload(p.out_table + ((size_t)blockIdx.y << p.log_n) + go);   // (times 2^5, like out32)
            else f = m == 0 ? q.out32[0] : m == 1 ? q.out32[1] : q.out32[2];
            val = R29::mul(val, R29::split<0>(f));
        }
        Fr out = pack29(val);
        out.reduce_once();
        gstore(dst + go, out);
    }
}

// ------------------------------------------------------------------ host side
static Fr pow2k(Fr x, uint32_t times) {
    for (uint32_t i = 0; i < times; i++) x = Fr::sqr(x);
    return x;
}

static Fr fr_root28() {
    // 7^((r-1) >> 28)
    Fr e = Fr::modulus();
    e.l[0] -= 1;
    Fr sh = Fr::zero();
    for (int i = 0; i < 8; i++) {
        uint64_t v = e.l[i] >> 28;
        if (i < 7) v |= (uint64_t)e.l[i + 1] << 4;
        sh.l[i] = (uint32_t)v;
    }
    return Fr::pow(Fr::from_u64(7), sh);
}

Fr fr_root_of_unity(uint32_t k) {
    static const Fr root = fr_root28();
    SHP_REQUIRE(k <= 28, SHP_ERR_INVALID_ARG, "log_n > 28");
    return pow2k(root, 28 - k);
}
Fr fr_delta() { return Fr::pow_u64(Fr::from_u64(7), 1ull << 28); }
Fr fr_zeta() {
    // (r-1)/3 by long division over the 32-bit limbs
    Fr e = Fr::modulus();
    e.l[0] -= 1;
    Fr q = Fr::zero();
    uint64_t rem = 0;
    for (int i = 7; i >= 0; i--) {
        uint64_t cur = (rem << 32) | e.l[i];
        q.l[i] = (uint32_t)(cur / 3);
        rem = cur % 3;
    }
    return Fr::pow(Fr::from_u64(7), q);
}

// SHP_NTT29=1 (read once per process): the tile runs in 29-bit lazy arithmetic and every table holds its powers times 2^5
static bool ntt29_mode() {
    static const bool on = getenv("SHP_NTT29") && atoi(getenv("SHP_NTT29")) == 1;
    return on;
}
static Fr table_first() { return ntt29_mode() ? Fr::from_u64(32) : Fr::one(); }
Fr ntt_out_table_factor() { return table_first(); }

// v[i] = first * base^i
static Fr* upload_powers(const Fr& base, size_t count, const Fr& first = table_first()) {
    std::vector<Fr> v(count);
    Fr acc = first;
    for (size_t i = 0; i < count; i++) {
        v[i] = acc;
        acc = acc * base;
    }
    Fr* d = nullptr;
    SHP_HIP(hipMalloc(&d, count * sizeof(Fr)));
    SHP_HIP(hipMemcpy(d, v.data(), count * sizeof(Fr), hipMemcpyHostToDevice));
    return d;
}

// out[i] = base^i for i < count, on the device: thread t raises base to t * 64 by square-and-multiply and walks 64
// entries from there (the tables of the large domains have up to 2^24 entries; the host loop above takes a second there)
__global__ void ntt_powers_kernel(Fr base, Fr first, size_t count, Fr* out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = t * 64;
    if (i0 >= count) return;
    Fr acc = first, sq = base;
    for (size_t e = i0; e; e >>= 1) {
        if (e & 1) acc = acc * sq;
        sq = Fr::sqr(sq);
    }
    const size_t i1 = i0 + 64 < count ? i0 + 64 : count;
    for (size_t i = i0; i < i1; i++) {
        gstore(out + i, acc);
        acc = acc * base;
    }
}

static Fr* device_powers(shp_ctx* ctx, const Fr& base, size_t count) {
    if (count <= 4096) return upload_powers(base, count);
    Fr* d = nullptr;
    SHP_HIP(hipMalloc(&d, count * sizeof(Fr)));
    const size_t threads = (count + 63) / 64;
    hipLaunchKernelGGL(ntt_powers_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, ctx->stream, base, table_first(), count, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d); SHP_HIP(e); }
    return d;
}

// table[(kp << lo_bits) | lo] = first * base^((kp * lo) << hi_bits): one thread per lo raises base to lo << hi_bits and walks
// the 2^s values of kp from there; threads with adjacent lo store adjacent entries
__global__ void ntt_pass_table_kernel(Fr base, Fr first, uint32_t s, uint32_t lo_bits, uint32_t hi_bits, Fr* out) {
    const size_t lo = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lo >> lo_bits) return;
    Fr step = Fr::one(), sq = base;
    for (size_t e = lo << hi_bits; e; e >>= 1) {
        if (e & 1) step = step * sq;
        sq = Fr::sqr(sq);
    }
    Fr acc = first;
    for (size_t kp = 0; kp < ((size_t)1 << s); kp++) {
        gstore(out + ((kp << lo_bits) | lo), acc);
        acc = acc * step;
    }
}

// SHP_NTT_DIT=0 (read once per process) restores the transposing passes of ntt_pass_kernel for every transform
static bool ntt_dit_mode() {
    static const bool on = !(getenv("SHP_NTT_DIT") && atoi(getenv("SHP_NTT_DIT")) == 0) && !ntt29_mode();
    return on;
}
bool ntt_dit_enabled() { return ntt_dit_mode(); }

void ntt_build_coset_table(shp_ctx* ctx, uint32_t log_n, const Fr& c, const Fr& g, bool inverse, Fr* out, hipStream_t st) {
    (void)ctx;
    SHP_REQUIRE(log_n >= 1 && log_n <= 27 && out, SHP_ERR_INVALID_ARG, "ntt: coset table arguments");
    const Fr w = inverse ? Fr::inv(fr_root_of_unity(log_n)) : fr_root_of_unity(log_n);
    const size_t count = (size_t)1 << log_n;
    hipLaunchKernelGGL(ntt_dit_table_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, c, w, g, log_n, out);
    SHP_HIP(hipGetLastError());
}

// the pass structure of a transform: digits (stages per pass) and the tile size they are cut for
static int ntt_digits(uint32_t log_n, uint32_t digits[3], uint32_t& tile_log) {
    tile_log = log_n <= 2 * NTT_TILE_SMALL ? NTT_TILE_SMALL : NTT_TILE_LARGE;
    digits[0] = digits[1] = digits[2] = 0;
    if (log_n <= tile_log) {
        digits[0] = log_n;
        return 1;
    }
    if (log_n <= 2 * tile_log) {
        digits[1] = log_n / 2;
        digits[0] = log_n - digits[1];
        return 2;
    }
    digits[2] = log_n / 3;
    digits[1] = (log_n - digits[2]) / 2;
    digits[0] = log_n - digits[1] - digits[2];
    return 3;
}

// tables shared by every size (caller holds ctx->mu)
static void ntt_small_tables(shp_ctx* ctx) {
    if (ctx->ntt_small[0]) return;
    Fr w = fr_root_of_unity(11);
    ctx->ntt_small[0] = upload_powers(w, 1024);
    ctx->ntt_small[1] = upload_powers(Fr::inv(w), 1024);
    ctx->zeta = fr_zeta();
    ctx->zeta2 = Fr::sqr(ctx->zeta);
    if (ntt29_mode()) {
        for (int d = 0; d < 2; d++) {
            std::vector<uint4> pa(1024), pb(1024);
            std::vector<uint32_t> pc(1024);
            Fr acc = Fr::from_u64(32);
            const Fr base = d ? Fr::inv(w) : w;
            for (int i = 0; i < 1024; i++) {
                const E29 e = R29::split<0>(acc);
                pa[i] = make_uint4(e.l[0], e.l[1], e.l[2], e.l[3]);
                pb[i] = make_uint4(e.l[4], e.l[5], e.l[6], e.l[7]);
                pc[i] = e.l[8];
                acc = acc * base;
            }
            SHP_HIP(hipMalloc(&ctx->ntt_small29_a[d], 1024 * sizeof(uint4)));
            SHP_HIP(hipMalloc(&ctx->ntt_small29_b[d], 1024 * sizeof(uint4)));
            SHP_HIP(hipMalloc(&ctx->ntt_small29_c[d], 1024 * sizeof(uint32_t)));
            SHP_HIP(hipMemcpy(ctx->ntt_small29_a[d], pa.data(), 1024 * sizeof(uint4), hipMemcpyHostToDevice));
            SHP_HIP(hipMemcpy(ctx->ntt_small29_b[d], pb.data(), 1024 * sizeof(uint4), hipMemcpyHostToDevice));
            SHP_HIP(hipMemcpy(ctx->ntt_small29_c[d], pc.data(), 1024 * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
}

const NttTables& ntt_tables_for(shp_ctx* ctx, uint32_t log_n) {
    std::lock_guard<std::mutex> g(ctx->mu);
    ntt_small_tables(ctx);
    auto it = ctx->ntt_tables.find(log_n);
    if (it != ctx->ntt_tables.end()) return it->second;
    NttTables t;
    t.log_n = log_n;
    // Up to 2^24 points every non-final pass has a table of its own with ONE entry per element it stores, laid out in the
    // order of the store: [kp][lo] (2^log_n entries for the first pass -- 512 MB per direction at 2^24 -- and 2^(log_n -
    // digits[0]) for the middle pass of three).  HBM has the room, and one product per element per pass boundary instead
    // of hi * lo is ALU time saved; the layout matters as much: with all powers in one table indexed by the exponent
    // (kp * lo) a wave gathered 32-byte entries kp lines apart -- four times the table's bytes in lines fetched, which
    // held the 2^20..2^24 transforms at the memory system's rate, not the ALU's.  Above 2^24 (SHP_NTT_FULL_TABLE_LOG
    // overrides the limit) the exponent is split into two small table lookups and a product.
    static const uint32_t full_log = getenv("SHP_NTT_FULL_TABLE_LOG") ? (uint32_t)atoi(getenv("SHP_NTT_FULL_TABLE_LOG")) : 24;
    t.h = (log_n + 1) / 2;
    Fr w = fr_root_of_unity(log_n);
    Fr wi = Fr::inv(w);
    t.lo[0] = device_powers(ctx, w, (size_t)1 << t.h);
    t.lo[1] = device_powers(ctx, wi, (size_t)1 << t.h);
    t.hi[0] = upload_powers(pow2k(w, t.h), (size_t)1 << (log_n - t.h));
    t.hi[1] = upload_powers(pow2k(wi, t.h), (size_t)1 << (log_n - t.h));
    if (log_n <= full_log) {
        uint32_t digits[3], tile_log;
        const int P = ntt_digits(log_n, digits, tile_log);
        uint32_t consumed = 0;
        for (int pass = 0; pass + 1 < P; pass++) {
            const uint32_t s = digits[pass], lo_bits = log_n - consumed - s, hi_bits = consumed;
            for (int d = 0; d < 2; d++) {
                Fr* tab = nullptr;
                SHP_HIP(hipMalloc(&tab, sizeof(Fr) << (s + lo_bits)));
                t.pass[d][pass] = tab;
                const uint32_t threads = 1u << lo_bits;
                hipLaunchKernelGGL(ntt_pass_table_kernel, dim3((threads + 255) / 256), dim3(256), 0, ctx->stream, d ? wi : w,
                                   table_first(), s, lo_bits, hi_bits, tab);
            }
            consumed += s;
        }
        SHP_HIP(hipGetLastError());
        SHP_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (ntt_dit_mode() && log_n <= full_log) {
        for (int d = 0; d < 2; d++) {
            SHP_HIP(hipMalloc(&t.dit[d], sizeof(Fr) << log_n));
            ntt_build_coset_table(ctx, log_n, Fr::one(), Fr::one(), d == 1, t.dit[d], ctx->stream);
        }
        SHP_HIP(hipStreamSynchronize(ctx->stream));
    }
    t.n_inv = Fr::inv(Fr::from_u64(1ull << log_n));
    return ctx->ntt_tables.emplace(log_n, t).first->second;
}

// `T`: the inter-pass twiddle tables, powers of w_(2^(log_n + tw_shift))
static void ntt_launch_with(shp_ctx* ctx, const NttTables& T, uint32_t tw_shift, const Fr* d_in, Fr* d_out, uint32_t log_n,
                            uint32_t batch, const NttOpts& o, hipStream_t st, Fr* tmp_in) {
    const size_t N = (size_t)1 << log_n;
    const int dir = o.inverse ? 1 : 0;
    SHP_REQUIRE(!(o.out_table && o.post_pattern), SHP_ERR_INVALID_ARG, "ntt: out_table with post_pattern");

    uint32_t digits[3], tile_log;
    const int P = ntt_digits(log_n, digits, tile_log);
    Fr* tmp = tmp_in;
    if (P > 1 && !tmp) {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->ntt_tmp.reserve((size_t)batch * N * sizeof(Fr));
        tmp = ctx->ntt_tmp.as<Fr>();
    }
    const uint32_t in_len = o.in_len ? o.in_len : (uint32_t)N;
    const size_t in_stride = o.in_stride ? o.in_stride : (o.in_len ? o.in_len : N);

    // decimation in time with every factor in the butterflies: whole-length inputs without the zeta patterns, tables of
    // this very domain (not the strided master tables of the allocation-free entry point)
    const bool coset = o.twist != nullptr;
    if (ntt_dit_mode() && tw_shift == 0 && T.dit[dir] && in_len == N && !o.pre_pattern && !o.post_pattern &&
        (!coset || (o.coset_tw && o.twist_count))) {
        uint32_t done = 0;
        for (int pass = 0; pass < P; pass++) {
            PassParams p = {};
            p.log_n = log_n;
            p.s = digits[pass];
            p.j0 = done;
            p.is_first = pass == 0;
            p.is_final = pass == P - 1;
            p.lo_bits = log_n - done - p.s;
            const uint32_t cmax = tile_log - p.s;
            p.c = p.is_final ? std::min(cmax, done) : std::min(cmax, p.lo_bits);
            p.do_scale = o.inverse ? 1 : 0;
            p.scale = T.n_inv;
            p.out_table = p.is_final ? o.out_table : nullptr;
            p.dit_tw = coset ? o.coset_tw : T.dit[dir];
            p.dit_cosets = coset ? o.twist_count : 1;
            p.dit_plain = coset ? 0 : 1;
            p.dit_scaled = coset && o.coset_scaled ? (o.coset_g == Fr::from_u64(32) ? 2 : 1) : 0;
            p.dit_g = o.coset_g;
            if (p.is_first) { p.src = d_in; p.src_stride = in_stride; }
            else { p.src = tmp; p.src_stride = N; }
            if (p.is_final) { p.dst = d_out; p.dst_stride = N; }
            else { p.dst = tmp; p.dst_stride = N; }
            const uint32_t tiles = (uint32_t)(N >> (p.s + p.c));
            if (tile_log == NTT_TILE_SMALL)
                hipLaunchKernelGGL((ntt_dit_pass_kernel<256, NTT_TILE_SMALL>), dim3(tiles, batch), dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL((ntt_dit_pass_kernel<512, NTT_TILE_LARGE>), dim3(tiles, batch), dim3(512), 0, st, p);
            SHP_HIP(hipGetLastError());
            done += p.s;
        }
        return;
    }

    uint32_t consumed = 0;
    for (int pass = 0; pass < P; pass++) {
        PassParams p;
        p.log_n = log_n;
        p.s = digits[pass];
        p.is_first = pass == 0;
        p.is_final = pass == P - 1;
        p.lo_bits = log_n - consumed - p.s;
        p.s1 = 0;
        p.s2 = 0;
        if (p.is_final) {
            if (P == 2) p.s1 = digits[0];
            if (P == 3) { p.s1 = digits[0]; p.s2 = digits[1]; }
            uint32_t cmax = tile_log - p.s;
            p.c = cmax < p.s1 ? cmax : p.s1;
        } else {
            uint32_t cmax = tile_log - p.s;
            p.c = cmax < p.lo_bits ? cmax : p.lo_bits;
        }
        p.h = T.h;
        p.tw_shift = tw_shift;
        p.in_len = in_len;
        p.pre_pattern = o.pre_pattern;
        p.post_pattern = o.post_pattern;
        p.do_scale = o.inverse ? 1 : 0;
        p.scale = T.n_inv;
        p.zeta = ctx->zeta;
        p.zeta2 = ctx->zeta2;
        p.twist = o.twist;
        p.src_div = o.twist && o.twist_count ? o.twist_count : 1;
        p.out_table = p.is_final ? o.out_table : nullptr;
        p.small_tw = ctx->ntt_small[dir];
        p.tw_lo = T.lo[dir];
        p.tw_hi = T.hi[dir];
        p.tw_pass = pass + 1 < P && tw_shift == 0 ? T.pass[dir][pass] : nullptr;
        if (p.is_first) { p.src = d_in; p.src_stride = in_stride; }
        else { p.src = tmp; p.src_stride = N; }
        if (p.is_final) { p.dst = d_out; p.dst_stride = N; }
        else { p.dst = tmp; p.dst_stride = N; }
        uint32_t tiles = (uint32_t)(N >> (p.s + p.c));
        if (ntt29_mode()) {
            Pass29 q;
            q.small_a = ctx->ntt_small29_a[dir];
            q.small_b = ctx->ntt_small29_b[dir];
            q.small_c = ctx->ntt_small29_c[dir];
            const Fr c32 = Fr::from_u64(32);
            q.zeta32 = ctx->zeta * c32;
            q.zeta2_32 = ctx->zeta2 * c32;
            const Fr f0 = o.inverse ? T.n_inv * c32 : c32;          // 1/n, or the Montgomery one: the pass's reducing product
            q.out32[0] = f0;
            q.out32[1] = f0 * ctx->zeta2;
            q.out32[2] = f0 * ctx->zeta;
            if (tile_log == NTT_TILE_SMALL)
                hipLaunchKernelGGL((ntt29_pass_kernel<256, NTT_TILE_SMALL>), dim3(tiles, batch), dim3(256), 0, st, p, q);
            else
                hipLaunchKernelGGL((ntt29_pass_kernel<512, NTT_TILE_LARGE>), dim3(tiles, batch), dim3(512), 0, st, p, q);
        } else if (tile_log == NTT_TILE_SMALL)
            hipLaunchKernelGGL((ntt_pass_kernel<256, NTT_TILE_SMALL>), dim3(tiles, batch), dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((ntt_pass_kernel<512, NTT_TILE_LARGE>), dim3(tiles, batch), dim3(512), 0, st, p);
        SHP_HIP(hipGetLastError());
        consumed += p.s;
    }
}

void ntt_launch(shp_ctx* ctx, const Fr* d_in, Fr* d_out, uint32_t log_n, uint32_t batch, const NttOpts& o,
                hipStream_t st, Fr* tmp_in) {
    SHP_REQUIRE(log_n >= 1 && log_n <= 27, SHP_ERR_INVALID_ARG, "ntt: log_n out of range [1,27]");
    if (batch == 0) return;
    ntt_launch_with(ctx, ntt_tables_for(ctx, log_n), 0, d_in, d_out, log_n, batch, o, st, tmp_in);
}

// ------------------------------------------------------------------ the entry point that allocates nothing
// Callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it)
// get shp_ntt_fr with every piece of device memory beyond their own buffer provided by the library up front:
//   * the ping-pong buffer of the multi-pass transform is this file-scope device array (each GPU the library runs
//     on holds its own copy until the process ends);
//   * the inter-pass twiddles come from two 1024-entry tables of powers of w_(2^20) built by shp_ctx_create
//     (a domain of 2^log_n points uses every 2^(20 - log_n)-th power).
constexpr uint32_t NTT_STATIC_LOG = 20;                                   // at most 2^20 elements (32 MiB) per call
static __device__ __align__(256) unsigned char scratch[((size_t)1 << NTT_STATIC_LOG) * sizeof(Fr)];

void ntt_static_setup(shp_ctx* ctx) {
    std::lock_guard<std::mutex> g(ctx->mu);
    ntt_small_tables(ctx);
    // `scratch` is a host-side stand-in for the symbol; this device's copy of the array lives at the address
    // the runtime reports here (the context's device is current)
    void* ptr = nullptr;
    SHP_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    SHP_REQUIRE(ptr, SHP_ERR_HIP, "ntt: the static ping-pong array has no device address");
    ctx->ntt_static_tmp = reinterpret_cast<Fr*>(ptr);
    const Fr w = fr_root_of_unity(NTT_STATIC_LOG), wi = Fr::inv(w);
    ctx->ntt_master_lo[0] = upload_powers(w, 1024);
    ctx->ntt_master_lo[1] = upload_powers(wi, 1024);
    ctx->ntt_master_hi[0] = upload_powers(pow2k(w, 10), 1024);
    ctx->ntt_master_hi[1] = upload_powers(pow2k(wi, 10), 1024);
}

void ntt_static_teardown(shp_ctx* ctx) {
    for (int d = 0; d < 2; d++) {
        if (ctx->ntt_master_lo[d]) (void)hipFree(ctx->ntt_master_lo[d]);
        if (ctx->ntt_master_hi[d]) (void)hipFree(ctx->ntt_master_hi[d]);
        ctx->ntt_master_lo[d] = ctx->ntt_master_hi[d] = nullptr;
    }
    ctx->ntt_static_tmp = nullptr;
}

}  // namespace shp

extern "C" int32_t shp_ntt_fr_noalloc(shp_ctx* ctx, void* d_data, uint32_t log_n, uint32_t batch, int32_t inverse,
                                      void* stream) {
    using namespace shp;
    return guarded([&] {
        SHP_REQUIRE(ctx && d_data, SHP_ERR_INVALID_ARG, "ntt_noalloc: null argument");
        SHP_REQUIRE(log_n >= 1 && log_n <= NTT_STATIC_LOG, SHP_ERR_INVALID_ARG,
                    "ntt_noalloc: log_n out of range [1,20] (the static ping-pong array holds 2^20 elements)");
        SHP_REQUIRE(((uint64_t)batch << log_n) <= (1ull << NTT_STATIC_LOG), SHP_ERR_INVALID_ARG,
                    "ntt_noalloc: batch * 2^log_n exceeds the 2^20 elements of the static ping-pong array");
        SHP_REQUIRE(ctx->ntt_static_tmp && ctx->ntt_master_lo[0], SHP_ERR_INTERNAL, "ntt_noalloc: context not set up");
        SHP_HIP(hipSetDevice(ctx->device));
        if (batch == 0) return;
        NttTables T;
        T.log_n = log_n;
        T.h = 10;
        for (int d = 0; d < 2; d++) { T.lo[d] = ctx->ntt_master_lo[d]; T.hi[d] = ctx->ntt_master_hi[d]; }
        T.n_inv = Fr::inv(Fr::from_u64(1ull << log_n));
        NttOpts o;
        o.inverse = inverse != 0;
        // every element of the ping-pong array that a pass reads was written by the pass before it in this same
        // call: nothing needs clearing between calls
        ntt_launch_with(ctx, T, NTT_STATIC_LOG - log_n, (const Fr*)d_data, (Fr*)d_data, log_n, batch, o, ctx->pick(stream),
                        ctx->ntt_static_tmp);
    });
}
