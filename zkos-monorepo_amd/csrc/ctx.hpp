// The per-(process, GPU) context behind shp_ctx.
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <mutex>

#include "common.hpp"
#include "field.hip.hpp"

namespace shp {

// Twiddle tables of one NTT size (device resident, built once):
//   small[j]   = w_2048^j            j < 1024   (butterflies inside an LDS tile; shared by all sizes)
//   lo[j]      = w_N^j               j < 2^h    } inter-pass twiddle w_N^e = hi[e >> h] * lo[e & (2^h-1)]
//   hi[j]      = w_N^(j * 2^h)       j < 2^(log_n - h)
//   pass[d][p][(kp << lo_bits) | lo] = w_N^((kp * lo) << hi_bits)   the inter-pass twiddles of the non-final pass p in the
//                                      order its store reads them (domains up to 2^24: adjacent outputs read adjacent entries)
struct NttTables {
    uint32_t log_n = 0, h = 0;
    Fr* lo[2] = {nullptr, nullptr};      // [0] forward, [1] inverse
    Fr* hi[2] = {nullptr, nullptr};
    Fr* pass[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // dit[d][2^j - 1 + k] = w_(2^(j+1))^k, k < 2^j: every butterfly twiddle of the decimation-in-time transform, stage by
    // stage (2^log_n - 1 entries; see ntt_dit_pass_kernel)
    Fr* dit[2] = {nullptr, nullptr};
    Fr n_inv;
};

}  // namespace shp

struct shp_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    std::mutex mu;
    shp::Fr* ntt_small[2] = {nullptr, nullptr};          // w_2048^j and its inverse powers
    // the same powers for the 29-bit lazy tile (SHP_NTT29=1): times 2^5 (Montgomery radix 2^261), split into nine 29-bit
    // limbs, stored as three planes (limbs 0-3, 4-7, 8) so that a twiddle is two 16-byte loads and one 4-byte load
    uint4* ntt_small29_a[2] = {nullptr, nullptr};
    uint4* ntt_small29_b[2] = {nullptr, nullptr};
    uint32_t* ntt_small29_c[2] = {nullptr, nullptr};
    std::map<uint32_t, shp::NttTables> ntt_tables;
    shp::DevBuf ntt_tmp;                                  // ping-pong buffer of the multi-pass NTT
    // the allocation-free NTT entry point: powers of w_(2^20) split as lo[j] = w^j, hi[j] = w^(1024 j)
    // (j < 1024, [0] forward, [1] inverse; every smaller domain's twiddles are a stride of these), and this
    // device's address of the library's static ping-pong array -- both set up once by shp_ctx_create
    shp::Fr* ntt_master_lo[2] = {nullptr, nullptr};
    shp::Fr* ntt_master_hi[2] = {nullptr, nullptr};
    shp::Fr* ntt_static_tmp = nullptr;
    shp::Fr zeta, zeta2;                                  // coset generator of the extended domain
    std::atomic<uint64_t> hbm_limit{0};                   // bytes the process may hold on the device (0: no cap)

    hipStream_t pick(void* s) const { return s ? reinterpret_cast<hipStream_t>(s) : stream; }
};

namespace shp {
const NttTables& ntt_tables_for(shp_ctx* ctx, uint32_t log_n);
Fr fr_root_of_unity(uint32_t k);     // primitive 2^k-th root (host)
Fr fr_zeta();                         // 7^((r-1)/3)
Fr fr_delta();                        // 7^(2^28)
// set-up / tear-down of what shp_ntt_fr_noalloc needs (called by shp_ctx_create / shp_ctx_destroy with the
// context's device current)
void ntt_static_setup(shp_ctx* ctx);
void ntt_static_teardown(shp_ctx* ctx);

// launchers used by the C ABI and the prover
struct NttOpts {
    bool inverse = false;     // w^-1 tables and 1/n scaling
    uint32_t in_len = 0;      // 0 = 2^log_n; else elements >= in_len of each input poly read as zero
    size_t in_stride = 0;     // elements between input polys (0 = in_len or 2^log_n)
    int pre_pattern = 0;      // 1: input i scaled by [1, zeta, zeta^2][i % 3]
    int post_pattern = 0;     // 1: output i scaled by [1, zeta^2, zeta][i % 3]
    // coset evaluation: `twist_count` transforms per input polynomial, transform b reads polynomial b / twist_count
    // with coefficient i scaled by twist[(b % twist_count) * 2^log_n + i] (= c_j^i: values on the coset c_j * H)
    const Fr* twist = nullptr;
    uint32_t twist_count = 0;
    // output i of transform b scaled by out_table[b * 2^log_n + i] INSTEAD of the inverse transform's 1/n (the table
    // carries that factor): the product the final store spends on 1/n anyway does a second job.  Every entry must also
    // carry ntt_out_table_factor(); not together with post_pattern.
    const Fr* out_table = nullptr;
    // The same coset evaluation with the twist folded into the butterflies' twiddles (ntt_dit_pass_kernel): per coset a
    // table built by ntt_build_coset_table, [coset_count][2^log_n]; output = coset_g * (values on the coset) when
    // coset_scaled.  A caller sets it NEXT TO `twist`: the launcher takes whichever path is enabled.
    const Fr* coset_tw = nullptr;
    Fr coset_g;
    bool coset_scaled = false;
};
// table[2^j - 1 + k] = (c w_N^k)^(2^(log_n - 1 - j)), k < 2^j, j < log_n, with entry 0 times g: the twiddles of a
// decimation-in-time transform onto the coset c H (c = 1: onto H itself); `out` holds 2^log_n elements
bool ntt_dit_enabled();                // false with SHP_NTT_DIT=0 (or SHP_NTT29=1): every transform takes ntt_pass_kernel
void ntt_build_coset_table(shp_ctx* ctx, uint32_t log_n, const Fr& c, const Fr& g, bool inverse, Fr* out, hipStream_t st);
// 1, or 2^5 when the 29-bit tile is selected (SHP_NTT29=1: its factors are kept times 2^5, see ntt.hip)
Fr ntt_out_table_factor();
// `tmp`: caller-owned ping-pong buffer of batch * 2^log_n elements (needed when log_n > 11);
// nullptr = the context's shared buffer (single-stream use only)
void ntt_launch(shp_ctx* ctx, const Fr* d_in, Fr* d_out, uint32_t log_n, uint32_t batch, const NttOpts& o,
                hipStream_t st, Fr* tmp = nullptr);
}  // namespace shp
