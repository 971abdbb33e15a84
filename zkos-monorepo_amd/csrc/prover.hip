// PLONK / KZG-SHPLONK prover driver for gfx950: keygen (device-resident proving key) and
// create_proof, behind shp_keygen / shp_prove.
//
// Replaces, for a circuit given as a circuit plan, the reference's
//   shielder_circuits::generate_keys_with_min_k  (/root/reference/crates/shielder_bindings/build.rs:19-32)
//   shielder_circuits::generate_proof            (/root/reference/crates/shielder_bindings/src/circuits/mod.rs:103-111)
// i.e. halo2_proofs v0.3.0 keygen_pk / create_proof with ProverSHPLONK and the EVM Keccak
// transcript.  Proof layout, challenge order and the identities proven are those the
// reference's verifier generator fixes (codegen/util.rs:133-245, evaluator.rs:45-223,
// pcs/bdfg21.rs, templates/Halo2Verifier.sol:247-307); phase order and blinding draws are
// those of halo2_proofs v0.3.0 create_proof (draw order: include/shielder_hip.h at shp_prove_with_randomness).  The host only sequences phases, hashes the transcript and handles
// a few scalars per phase; every vector operation runs in HIP kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "prover_kernels.hip.hpp"
#include "quotient29.hip.hpp"
#include "poseidon2.hip.hpp"
#include "verify.hpp"

using namespace shp;

namespace {
struct PolyKey {
    int kind;       // 0 adv, 1 pz, 2 lz, 3 lpi, 4 lpt, 5 fix, 6 sig, 7 h, 8 rand
    uint32_t idx;
    bool operator<(const PolyKey& o) const { return kind != o.kind ? kind < o.kind : idx < o.idx; }
    bool operator==(const PolyKey& o) const { return kind == o.kind && idx == o.idx; }
};
struct RotSet {
    std::vector<int32_t> rots;          // sorted
    std::vector<PolyKey> polys;
};
}  // namespace

// the verifier's query order (pcs.rs:60-104) grouped into rotation sets (bdfg21.rs:443-494)
static void build_rotation_sets(const PlanData& P, std::vector<int32_t>& superset, std::vector<RotSet>& sets) {
    std::vector<std::pair<PolyKey, int32_t>> qs;
    const uint32_t S = P.num_sets, L = (uint32_t)P.lookups.size();
    for (auto& q : P.adv_q) qs.push_back({{0, q.col}, q.rot});
    for (uint32_t s = 0; s < S; s++) { qs.push_back({{1, s}, 0}); qs.push_back({{1, s}, 1}); }
    for (int s = (int)S - 2; s >= 0; s--) qs.push_back({{1, (uint32_t)s}, P.rot_last});
    for (uint32_t l = 0; l < L; l++) {
        qs.push_back({{2, l}, 0}); qs.push_back({{3, l}, 0}); qs.push_back({{4, l}, 0});
        qs.push_back({{3, l}, -1}); qs.push_back({{2, l}, 1});
    }
    for (auto& q : P.fix_q) qs.push_back({{5, q.col}, q.rot});
    for (uint32_t i = 0; i < P.perm_cols.size(); i++) qs.push_back({{6, i}, 0});
    qs.push_back({{7, 0}, 0});
    qs.push_back({{8, 0}, 0});
    std::vector<std::pair<PolyKey, std::vector<int32_t>>> polys;
    for (auto& q : qs) {
        if (std::find(superset.begin(), superset.end(), q.second) == superset.end()) superset.push_back(q.second);
        auto it = std::find_if(polys.begin(), polys.end(), [&](auto& e) { return e.first == q.first; });
        if (it == polys.end()) polys.push_back({q.first, {q.second}});
        else it->second.push_back(q.second);
    }
    std::sort(superset.begin(), superset.end());
    for (auto& pe : polys) {
        std::vector<int32_t> rs = pe.second;
        std::sort(rs.begin(), rs.end());
        auto it = std::find_if(sets.begin(), sets.end(), [&](RotSet& s) { return s.rots == rs; });
        if (it == sets.end()) sets.push_back({rs, {pe.first}});
        else it->polys.push_back(pe.first);
    }
}

// ------------------------------------------------------------------------------ proving key
struct shp_pk {
    shp_ctx* ctx = nullptr;
    shp_srs* srs = nullptr;
    PlanData plan;
    uint32_t n = 0, ne = 0, NV = 0;               // NV = committed Lagrange polys per proof
    // The quotient domain: the quotient has degree < (d - 1) n, so nc = d - 1 cosets c_j * H of size n carry it,
    // c_j = zeta * omega_ext^j -- a subset of halo2's extended domain zeta * H_ext (2^e n points), laid out
    // coset-major: index j * n + t <-> point c_j * omega^t.  nq = nc * n (7 n instead of 8 n for the withdraw
    // circuit).  twist[j][i] = c_j^i turns coefficients into values on coset j with one size-n NTT; cinv_n[j][t] =
    // c_j^-t / n (the factors the inverse transform of the quotient's cosets stores with) and the inverse Vandermonde
    // matrix on the nodes c_j^n turn the per-coset interpolants back into the quotient pieces (coset_combine_kernel).
    uint32_t nc = 0, nq = 0;
    Fr *twist = nullptr, *vinv = nullptr, *cinv_n = nullptr;
    Fr* coset_tw = nullptr;                  // [nc][n]: the coset transforms' butterfly twiddles (ntt_build_coset_table), with the twist in them
    Fr coset_g;                              // the factor the twists carry (1, or 2^5 for a key with a generated quotient kernel)
    // device-resident constants
    Fr *fixed_vals = nullptr, *fixed_polys = nullptr, *fixed_cos = nullptr;
    Fr *sigma_vals = nullptr, *sigma_polys = nullptr, *sigma_cos = nullptr;
    Fr *l0 = nullptr, *l_last = nullptr, *l_active = nullptr;
    Fr *x_ext = nullptr, *omegas = nullptr, *consts = nullptr;
    Instr* gate_prog = nullptr;
    void (*gate_fn)(ProgArgs) = nullptr;           // the compiled gate program, when the plan is one the library ships
    // the compiled quotient kernel (gates, permutation and lookup terms in one pass, 29-bit lazy arithmetic), when the
    // plan is one the library ships.  Non-null: twist, x_ext, consts29 and with them fixed_cos, sigma_cos, l0, l_last,
    // l_active are in Montgomery radix 2^261 (quotient29.hip.hpp) and the quotient phase runs this kernel alone
    void (*quot_fn)(Quot29Args) = nullptr;
    Fr* consts29 = nullptr;
    Fr* dx_ext = nullptr;                          // [PC][nq] delta^c x over the quotient domain (radix 2^261)
    std::vector<Instr*> lk_in_prog, lk_tab_prog;
    Query *adv_q = nullptr, *fix_q = nullptr, *inst_q = nullptr;
    // host copies
    std::vector<G1Affine> fixed_comm, sigma_comm;
    uint8_t vk_digest[32];
    uint8_t own_digest[32];                        // the library's Keccak digest over plan and commitments, whatever vk_digest is
    bool digest_injected = false;                  // vk_digest = the caller's transcript_repr, not the library's own
    std::vector<uint8_t> plan_bytes;               // the circuit plan the key was generated from (shp_pk_save)
    Fr omega, omega_inv, delta, zeta;
    std::vector<Fr> delta_pow;
    std::vector<int32_t> rot_superset;             // SHPLONK rotation sets of the plan (fixed per key)
    std::vector<RotSet> rot_sets;
    std::vector<void*> owned;

    template <class T>
    T* dalloc(size_t count) {
        void* p = nullptr;
        SHP_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
        owned.push_back(p);
        return (T*)p;
    }
    template <class T>
    T* dupload(const std::vector<T>& v) {
        T* p = dalloc<T>(v.size());
        if (!v.empty()) SHP_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    ~shp_pk() {
        for (void* p : owned) (void)hipFree(p);
    }
};

// Gate programs compiled ahead of time (tools/gen_gate_kernels.py): the circuits are fixed when the library is built, as
// they are for the reference (/root/reference/crates/shielder_bindings/build.rs:19-48), so the plans the library ships
// get their gate expressions as straight-line code -- no operand moves through the slot file, no dispatch: 36.6 k
// vector instructions per 64 rows of the withdraw program instead of the interpreter's 53 k.  Any other plan is
// interpreted; both give the same values.
#include "gate_kernels.inc"

typedef void (*GateKernelFn)(ProgArgs);

// FNV-1a over what the generated code depends on: slot count, instructions, the three query lists
static uint64_t gate_program_hash(const PlanData& P) {
    uint64_t h = 0xCBF29CE484222325ull;
    auto eat = [&](uint32_t w) {
        for (int i = 0; i < 4; i++) {
            h = (h ^ ((w >> (8 * i)) & 0xffu)) * 0x100000001B3ull;
        }
    };
    eat(P.gate_slots);
    eat((uint32_t)P.gate_prog.size());
    for (const Instr& in : P.gate_prog) { eat(in.op); eat(in.dst); eat(in.a); eat(in.b); }
    for (const std::vector<Query>* qs : {&P.adv_q, &P.fix_q, &P.inst_q}) {
        eat((uint32_t)qs->size());
        for (const Query& q : *qs) { eat(q.col); eat((uint32_t)q.rot); }
    }
    return h;
}

static GateKernelFn find_gate_kernel(const PlanData& P) {
    if (P.phased()) return nullptr;                  // the generated kernels know nothing of per-proof challenges
    const char* e = getenv("SHP_GATE_AOT");
    if (e && e[0] == '0') return nullptr;                    // A/B and tests: always interpret
    const uint64_t h = gate_program_hash(P);
    for (const GateKernelEntry& g : GATE_KERNELS)
        if (g.hash == h && g.n_instr == P.gate_prog.size()) return g.fn;
    return nullptr;
}

// The whole numerator of the quotient as one generated kernel per shipped plan (tools/gen_quotient_kernels.py); the
// kernels are compiled in quotient.hip.
// FNV-1a over what the generated quotient kernel depends on: the gate program and query lists, the constraint degree
// (it sets the size of a permutation set), the permutation columns, every lookup's compression programs
static uint64_t quotient_hash(const PlanData& P) {
    uint64_t h = 0xCBF29CE484222325ull;
    auto eat = [&](uint32_t w) {
        for (int i = 0; i < 4; i++) {
            h = (h ^ ((w >> (8 * i)) & 0xffu)) * 0x100000001B3ull;
        }
    };
    auto eat_prog = [&](uint32_t slots, const std::vector<Instr>& prog) {
        eat(slots);
        eat((uint32_t)prog.size());
        for (const Instr& in : prog) { eat(in.op); eat(in.dst); eat(in.a); eat(in.b); }
    };
    eat_prog(P.gate_slots, P.gate_prog);
    for (const std::vector<Query>* qs : {&P.adv_q, &P.fix_q, &P.inst_q}) {
        eat((uint32_t)qs->size());
        for (const Query& q : *qs) { eat(q.col); eat((uint32_t)q.rot); }
    }
    eat(P.degree);
    eat((uint32_t)P.perm_cols.size());
    for (const auto& c : P.perm_cols) { eat(c.first); eat(c.second); }
    eat((uint32_t)P.lookups.size());
    for (const LookupPlan& lk : P.lookups) {
        eat(lk.m);
        eat_prog(lk.in_slots, lk.in_prog);
        eat_prog(lk.tab_slots, lk.tab_prog);
    }
    return h;
}

static QuotientKernelFn find_quotient_kernel(const PlanData& P) {
    if (P.phased()) return nullptr;
    for (const char* name : {"SHP_GATE_AOT", "SHP_QUOT29"}) {       // A/B and tests: the separate radix-2^256 kernels
        const char* e = getenv(name);
        if (e && e[0] == '0') return nullptr;
    }
    return quotient_kernel_for(quotient_hash(P));
}

// `b` non-null: a second program of the same size in the same launch
static void launch_prog(hipStream_t st, const ProgArgs& a, uint32_t size, bool throughput, const ProgArgs* b = nullptr) {
    // up to 16 slots: the register-file interpreter (no LDS) for throughput lanes -- it leaves the CU's LDS to the
    // other lanes' kernels (+3 % pool throughput); a lone proof is 17 us faster with the LDS one (231 vs 214 us).
    // SHP_PROG_REG=0 / 1 forces one of them (A/B, tests)
    static const char* const force = getenv("SHP_PROG_REG");
    const bool use_reg = force ? force[0] != '0' : throughput;
    const ProgArgs& second = b ? *b : a;
    const dim3 grid(size / PROG_THREADS, b ? 2 : 1);
    const uint32_t nslots = std::max(a.nslots, second.nslots);
    if (use_reg && nslots <= 16) {
        hipLaunchKernelGGL(prog_fold_reg_kernel, grid, dim3(PROG_THREADS), 0, st, a, second);
        return;
    }
    size_t lds = (size_t)nslots * 8 * PROG_THREADS * 4;
    static std::once_flag attr_once;       // lanes launch from several host threads
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)prog_fold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 8 * PROG_THREADS * 4);
    });
    hipLaunchKernelGGL(prog_fold_kernel, grid, dim3(PROG_THREADS), lds, st, a, second);
}

// stored_comm: commitments read from a saved key (fixed then sigma); nullptr = compute them (MSMs)
// transcript_repr_le32 (optional): the first word of the transcript, 32-byte little-endian canonical -- halo2's
// vk.transcript_repr() (absorbed at /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:247-248, emitted at
// src/lib/codegen.rs:162) for callers that hold the halo2 VerifyingKey; nullptr = the library's own digest.
static void find_witness_permutations(PlanData& P);
static void keygen_impl(shp_ctx* ctx, shp_srs* srs, const uint8_t* plan_bytes, size_t len, shp_pk** out,
                        const std::vector<G1Affine>* stored_comm = nullptr, const uint8_t* transcript_repr_le32 = nullptr) {
    SHP_REQUIRE(ctx && srs && plan_bytes && out, SHP_ERR_INVALID_ARG, "keygen: null argument");
    SHP_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<shp_pk> pk(new shp_pk);
    pk->ctx = ctx;
    pk->srs = srs;
    pk->plan_bytes.assign(plan_bytes, plan_bytes + len);
    PlanData& P = pk->plan;
    P.parse(plan_bytes, len);
    find_witness_permutations(P);
    SHP_REQUIRE(P.k == srs->k, SHP_ERR_INVALID_ARG, "keygen: plan k differs from the SRS k");
    SHP_REQUIRE(P.n >= 64 && P.k <= 16 && P.perm_cols.size() <= MAX_PERM_COLS, SHP_ERR_INVALID_ARG,
                "keygen: unsupported plan shape");
    keccak256(plan_bytes, len, P.plan_hash);
    const uint32_t n = P.n, k = P.k, ek = P.ext_k, ne = 1u << ek;
    const uint32_t F = P.num_fixed, PC = (uint32_t)P.perm_cols.size(), L = (uint32_t)P.lookups.size();
    pk->n = n;
    pk->ne = ne;
    pk->NV = P.num_advice + 3 * L + P.num_sets;
    hipStream_t st = ctx->stream;
    pk->omega = fr_root_of_unity(k);
    pk->omega_inv = Fr::inv(pk->omega);
    pk->delta = fr_delta();
    pk->zeta = fr_zeta();
    Fr om_ext = fr_root_of_unity(ek);
    const uint32_t nc = P.degree - 1, nq = nc * n;
    SHP_REQUIRE(nc >= 1 && nc <= MAX_COSETS && nc <= (1u << (ek - k)), SHP_ERR_INVALID_ARG, "keygen: unsupported constraint degree");
    pk->nc = nc;
    pk->nq = nq;
    // omega^i; the quotient domain's points c_j omega^t, twists c_j^i and untwists c_j^-t
    pk->quot_fn = find_quotient_kernel(P);
    // a key with a generated quotient kernel keeps what that kernel reads in Montgomery radix 2^261: a factor 32 on the
    // twists (every coset transform of this key passes it on to its output) and on the points of the quotient domain
    const Fr radix_shift = pk->quot_fn ? Fr::from_u64(32) : Fr::one();
    std::vector<Fr> om(n), xe(nq), tw(nq), ci(nq), cj(nc);
    {
        Fr a = Fr::one();
        for (uint32_t i = 0; i < n; i++) { om[i] = a; a = a * pk->omega; }
        a = pk->zeta;
        for (uint32_t j = 0; j < nc; j++) { cj[j] = a; a = a * om_ext; }
        for (uint32_t j = 0; j < nc; j++) {
            const Fr cinv1 = Fr::inv(cj[j]);
            Fr p = radix_shift, q = Fr::one();
            for (uint32_t t = 0; t < n; t++) {
                xe[(size_t)j * n + t] = cj[j] * om[t] * radix_shift;
                tw[(size_t)j * n + t] = p;
                ci[(size_t)j * n + t] = q;
                p = p * cj[j];
                q = q * cinv1;
            }
        }
    }
    pk->omegas = pk->dupload(om);
    pk->x_ext = pk->dupload(xe);
    pk->twist = pk->dupload(tw);
    pk->coset_g = radix_shift;
    if (ntt_dit_enabled()) {
        pk->coset_tw = pk->dalloc<Fr>((size_t)nq);
        for (uint32_t j = 0; j < nc; j++) ntt_build_coset_table(ctx, k, cj[j], radix_shift, false, pk->coset_tw + (size_t)j * n, st);
    }
    {
        const Fr n_inv = Fr::inv(Fr::from_u64(n)) * ntt_out_table_factor();
        for (Fr& v : ci) v = v * n_inv;
        pk->cinv_n = pk->dupload(ci);
    }
    pk->delta_pow.resize(PC);
    {
        Fr a = Fr::one();
        for (uint32_t c = 0; c < PC; c++) { pk->delta_pow[c] = a; a = a * pk->delta; }
    }
    // fixed columns and sigma columns (Lagrange values)
    std::vector<Fr> fv((size_t)F * n, Fr::zero());
    for (uint32_t c = 0; c < F; c++)
        for (auto& e : P.fixed[c]) fv[(size_t)c * n + e.first] = e.second;
    std::vector<Fr> sv((size_t)PC * n);
    for (uint32_t c = 0; c < PC; c++)
        for (uint32_t r = 0; r < n; r++) {
            uint32_t tgt = P.mapping[c][r];
            sv[(size_t)c * n + r] = pk->delta_pow[tgt / n] * om[tgt % n];
        }
    pk->fixed_vals = pk->dupload(fv);
    pk->sigma_vals = pk->dupload(sv);
    pk->fixed_polys = pk->dalloc<Fr>((size_t)F * n);
    pk->sigma_polys = pk->dalloc<Fr>((size_t)PC * n);
    pk->fixed_cos = pk->dalloc<Fr>((size_t)F * nq);
    pk->sigma_cos = pk->dalloc<Fr>((size_t)PC * nq);
    Fr* tmp = pk->dalloc<Fr>((size_t)std::max<uint32_t>({F, PC, 3}) * nq);
    NttOpts inv;
    inv.inverse = true;
    NttOpts ext;                                   // coefficients -> values on the nc cosets, [poly][coset][n]
    ext.twist = pk->twist;
    ext.twist_count = nc;
    ext.coset_tw = pk->coset_tw; ext.coset_g = pk->coset_g; ext.coset_scaled = pk->quot_fn != nullptr;
    ntt_launch(ctx, pk->fixed_vals, pk->fixed_polys, k, F, inv, st, tmp);
    ntt_launch(ctx, pk->fixed_polys, pk->fixed_cos, k, F * nc, ext, st, tmp);
    ntt_launch(ctx, pk->sigma_vals, pk->sigma_polys, k, PC, inv, st, tmp);
    ntt_launch(ctx, pk->sigma_polys, pk->sigma_cos, k, PC * nc, ext, st, tmp);
    // l_0, l_last, l_active on the quotient domain
    std::vector<Fr> lv((size_t)3 * n, Fr::zero());
    lv[0] = Fr::one();
    lv[(size_t)n + P.usable] = Fr::one();
    for (uint32_t r = 0; r < P.usable; r++) lv[(size_t)2 * n + r] = Fr::one();
    Fr* d_lv = pk->dupload(lv);
    Fr* d_lp = pk->dalloc<Fr>((size_t)3 * n);
    Fr* d_lc = pk->dalloc<Fr>((size_t)3 * nq);
    ntt_launch(ctx, d_lv, d_lp, k, 3, inv, st, tmp);
    ntt_launch(ctx, d_lp, d_lc, k, 3 * nc, ext, st, tmp);
    pk->l0 = d_lc;
    pk->l_last = d_lc + nq;
    pk->l_active = d_lc + 2 * (size_t)nq;
    // per coset: 1 / (c_j^n - 1); and the inverse of the Vandermonde matrix V[j][m] = (c_j^n)^m (Gauss-Jordan)
    {
        std::vector<Fr> ti(nc), node(nc);
        for (uint32_t j = 0; j < nc; j++) {
            node[j] = Fr::pow_u64(cj[j], n);
            ti[j] = Fr::inv(node[j] - Fr::one());
        }
        std::vector<Fr> A((size_t)nc * nc), I((size_t)nc * nc, Fr::zero());
        for (uint32_t j = 0; j < nc; j++) {
            Fr p = Fr::one();
            for (uint32_t m = 0; m < nc; m++) { A[(size_t)j * nc + m] = p; p = p * node[j]; }
            I[(size_t)j * nc + j] = Fr::one();
        }
        for (uint32_t col = 0; col < nc; col++) {
            uint32_t piv = col;
            while (piv < nc && A[(size_t)piv * nc + col].is_zero()) piv++;
            SHP_REQUIRE(piv < nc, SHP_ERR_INTERNAL, "keygen: singular coset matrix");
            for (uint32_t c = 0; c < nc; c++) {
                std::swap(A[(size_t)piv * nc + c], A[(size_t)col * nc + c]);
                std::swap(I[(size_t)piv * nc + c], I[(size_t)col * nc + c]);
            }
            const Fr pinv = Fr::inv(A[(size_t)col * nc + col]);
            for (uint32_t c = 0; c < nc; c++) {
                A[(size_t)col * nc + c] = A[(size_t)col * nc + c] * pinv;
                I[(size_t)col * nc + c] = I[(size_t)col * nc + c] * pinv;
            }
            for (uint32_t r = 0; r < nc; r++) {
                if (r == col) continue;
                const Fr f = A[(size_t)r * nc + col];
                if (f.is_zero()) continue;
                for (uint32_t c = 0; c < nc; c++) {
                    A[(size_t)r * nc + c] = A[(size_t)r * nc + c] - f * A[(size_t)col * nc + c];
                    I[(size_t)r * nc + c] = I[(size_t)r * nc + c] - f * I[(size_t)col * nc + c];
                }
            }
        }
        // the division by the vanishing polynomial is a constant per coset (1 / (c_j^n - 1)) and everything between the
        // numerator and the solve is linear: folded into column j of the inverse matrix, it costs nothing per proof
        for (uint32_t m = 0; m < nc; m++)
            for (uint32_t j = 0; j < nc; j++) I[(size_t)m * nc + j] = I[(size_t)m * nc + j] * ti[j];
        pk->vinv = pk->dupload(I);                 // h_m = sum_j vinv[m][j] * y_j / (c_j^n - 1)
    }
    {
        // a circuit with several phases reads its challenges as constants behind the plan's own: this table is the template
        // every proving lane copies and completes per proof
        std::vector<Fr> cst(P.consts);
        cst.resize(P.consts.size() + P.challenge_phase.size(), Fr::zero());
        pk->consts = pk->dupload(cst);
    }
    if (pk->quot_fn) {
        std::vector<Fr> c29(P.consts);
        for (Fr& c : c29) c = c * radix_shift;
        pk->consts29 = pk->dupload(c29);
        std::vector<Fr> dx((size_t)PC * nq);
        for (uint32_t c = 0; c < PC; c++)
            for (uint32_t i = 0; i < nq; i++) dx[(size_t)c * nq + i] = pk->delta_pow[c] * xe[i];     // xe carries the 32
        pk->dx_ext = pk->dupload(dx);
    }
    pk->gate_prog = pk->dupload(P.device_prog(P.gate_prog));
    pk->gate_fn = find_gate_kernel(P);
    for (auto& lk : P.lookups) {
        pk->lk_in_prog.push_back(pk->dupload(P.device_prog(lk.in_prog)));
        pk->lk_tab_prog.push_back(pk->dupload(P.device_prog(lk.tab_prog)));
    }
    pk->adv_q = pk->dupload(P.adv_q);
    pk->fix_q = pk->dupload(P.fix_q);
    pk->inst_q = pk->dupload(P.inst_q);
    // commitments to the fixed and sigma columns (Lagrange basis)
    pk->fixed_comm.resize(F);
    pk->sigma_comm.resize(PC);
    if (stored_comm) {
        SHP_REQUIRE(stored_comm->size() == (size_t)F + PC, SHP_ERR_FORMAT, "pk_load: commitment count differs from the plan");
        std::copy(stored_comm->begin(), stored_comm->begin() + F, pk->fixed_comm.begin());
        std::copy(stored_comm->begin() + F, stored_comm->end(), pk->sigma_comm.begin());
    } else {
        if (F) msm_run(srs->msm_gl, pk->fixed_vals, n, n, F, pk->fixed_comm.data(), st);
        if (PC) msm_run(srs->msm_gl, pk->sigma_vals, n, n, PC, pk->sigma_comm.data(), st);
    }
    SHP_HIP(hipStreamSynchronize(st));
    // vk digest: keccak("shielder-hip-vk-v1" | keccak(plan) | commitments) mod r
    std::vector<uint8_t> body;
    const char* tag = "shielder-hip-vk-v1";
    body.insert(body.end(), tag, tag + 18);
    body.insert(body.end(), P.plan_hash, P.plan_hash + 32);
    auto push_pt = [&](const G1Affine& p) {
        uint8_t b[64];
        if (p.is_inf()) memset(b, 0, 64);
        else { to_be32(p.x, b); to_be32(p.y, b + 32); }
        body.insert(body.end(), b, b + 64);
    };
    for (auto& p : pk->fixed_comm) push_pt(p);
    for (auto& p : pk->sigma_comm) push_pt(p);
    uint8_t h[32];
    keccak256(body.data(), body.size(), h);
    Fr raw;
    for (int i = 0; i < 8; i++)
        raw.l[i] = ((uint32_t)h[31 - 4 * i]) | ((uint32_t)h[30 - 4 * i] << 8) | ((uint32_t)h[29 - 4 * i] << 16) |
                   ((uint32_t)h[28 - 4 * i] << 24);
    to_be32(Fr::from_raw(raw), pk->vk_digest);
    memcpy(pk->own_digest, pk->vk_digest, 32);
    if (transcript_repr_le32) {
        Fr t;
        memcpy(t.l, transcript_repr_le32, 32);
        SHP_REQUIRE(!t.geq_p(), SHP_ERR_FORMAT, "keygen: transcript_repr is not a canonical field element");
        to_be32(Fr::from_raw(t), pk->vk_digest);
        pk->digest_injected = true;
    }
    build_rotation_sets(P, pk->rot_superset, pk->rot_sets);
    SHP_REQUIRE(pk->rot_superset.size() <= MAX_POINTS && pk->rot_sets.size() <= MAX_SETS, SHP_ERR_INVALID_ARG,
                "keygen: too many rotation sets");
    *out = pk.release();
}

// P1: the plan's witness program over Montgomery values (the reference's `synthesize`).
// Ops [lo, hi) with a caller-owned scratch row.
static void run_witness_range(const PlanData& P, const std::vector<Fr>& inputs, Fr* grid /* [A][n], pinned */, Fr* scratch,
                              uint32_t lo, uint32_t hi, const Fr* chal = nullptr /* the challenges squeezed so far */) {
    const uint32_t A = P.num_advice, n = P.n;
    auto cell = [&](uint32_t c, uint32_t r) -> Fr& { return c == A ? scratch[r] : grid[(size_t)c * n + r]; };
    // ranges that are whole Poseidon2 permutations run natively (find_witness_permutations): the first one at or after lo
    size_t wp = std::lower_bound(P.wperms.begin(), P.wperms.end(), lo, [](const WPermutation& a, uint32_t v) { return a.lo < v; }) -
                P.wperms.begin();
    for (uint32_t i = lo; i < hi; i++) {
        if (wp < P.wperms.size() && P.wperms[wp].lo == i) {
            const WPermutation& f = P.wperms[wp++];
            if (f.hi <= hi) {
                poseidon2_trace_host(grid, n, f.cols, f.r0);
                i = f.hi - 1;
                continue;
            }
        }
        const WOp& w = P.wops[i];
        Fr v;
        switch (w.op) {
            case W_INPUT: v = inputs[w.imm]; break;
            case W_CONST: v = P.consts[w.imm]; break;
            case W_CHAL:
                if (!chal) throw Error(SHP_ERR_FORMAT, "witness: challenge read outside a proof");
                v = chal[w.imm];
                break;
            case W_COPY: v = cell(w.ac, w.ar); break;
            case W_ADD: v = cell(w.ac, w.ar) + cell(w.bc, w.br); break;
            case W_SUB: v = cell(w.ac, w.ar) - cell(w.bc, w.br); break;
            case W_MUL: v = cell(w.ac, w.ar) * cell(w.bc, w.br); break;
            case W_ADDC: v = cell(w.ac, w.ar) + P.consts[w.imm]; break;
            case W_MULC: v = cell(w.ac, w.ar) * P.consts[w.imm]; break;
            case W_POW7: {
                Fr x = cell(w.ac, w.ar), x2 = Fr::sqr(x), x3 = x2 * x;
                v = Fr::sqr(x3) * x;
                break;
            }
            case W_BITS: case W_SHR: {
                Fr raw = cell(w.ac, w.ar).to_raw(), o = Fr::zero();
                if (w.op == W_BITS) {
                    for (uint32_t b = 0; b < w.imm; b++) o.l[b >> 5] |= raw.l[b >> 5] & (1u << (b & 31));
                } else {
                    for (uint32_t b = w.imm; b < 256; b++)
                        if ((raw.l[b >> 5] >> (b & 31)) & 1) o.l[(b - w.imm) >> 5] |= 1u << ((b - w.imm) & 31);
                }
                v = Fr::from_raw(o);
                break;
            }
            default: throw Error(SHP_ERR_FORMAT, "witness: bad op");
        }
        cell(w.dc, w.dr) = v;
    }
}

// The circuits' Poseidon2 chip writes a permutation out as ~1770 cell operations (an S-box is one, a linear layer two
// dozen additions), and the interpreter above spends 31 us on them where the permutation itself takes 12.  Ranges of the
// witness program that ARE such a permutation are found here, once per key, and run natively from then on
// (poseidon2_trace_host).  Nothing is assumed about how a plan was produced:
//   * shape: a range that starts at an op reading the grid, reads grid cells of at most eight columns in rows
//     r0 .. r0 + 55 only, reads no scratch cell it has not written, takes no witness input or challenge, and writes every
//     cell of those columns' rows r0 + 1 .. r0 + 56 exactly once and no other grid cell;
//   * what it leaves in scratch is not read afterwards (a later op writes a scratch cell before it reads it);
//   * meaning: on two random input states the interpreted range and the native permutation (state order = ascending
//     column index) fill the 456 cells with the same values -- the cells are polynomials of degree < 7^60 < 2^169 in the
//     input state, so two different fillings agree on a random state with probability < 2^-84.
// A plan without such ranges, or with another hash in the same shape, is interpreted as before.  SHP_WITNESS_NATIVE=0
// turns the search off.
static void find_witness_permutations(PlanData& P) {
    P.wperms.clear();
    static const bool off = getenv("SHP_WITNESS_NATIVE") && atoi(getenv("SHP_WITNESS_NATIVE")) == 0;
    const uint32_t A = P.num_advice, n = P.n, nops = (uint32_t)P.wops.size(), ns = std::max<uint32_t>(P.n_wscratch, 1);
    if (off || nops < 64 * P2_TRACE_ROWS / 8 || A < 8) return;
    auto n_reads = [](const WOp& w) {
        switch (w.op) {
            case W_INPUT: case W_CONST: case W_CHAL: return 0;
            case W_ADD: case W_SUB: case W_MUL: return 2;
            default: return 1;
        }
    };
    std::vector<Fr> grid, ref, scratch(ns);
    std::vector<uint8_t> written;                         // per candidate: scratch cells written, then grid cells written
    const std::vector<Fr> no_inputs;
    uint32_t i = 0;
    while (i < nops) {
        const WOp& first = P.wops[i];
        // a start: reads the grid, writes scratch
        if (!(n_reads(first) >= 1 && first.ac != A && first.dc == A) || first.ar + P2_TRACE_ROWS >= P.usable) { i++; continue; }
        const uint32_t r0 = first.ar;
        uint32_t cols[8], n_cols = 0, cells = 0, j = i;
        bool ok = true;
        written.assign(ns + 8 * (size_t)P2_TRACE_ROWS, 0);
        auto col_index = [&](uint32_t c) -> int {
            for (uint32_t q = 0; q < n_cols; q++)
                if (cols[q] == c) return (int)q;
            if (n_cols == 8) return -1;
            cols[n_cols] = c;
            return (int)n_cols++;
        };
        for (; j < nops && ok && cells < 8 * P2_TRACE_ROWS; j++) {
            const WOp& w = P.wops[j];
            if (w.op == W_INPUT || w.op == W_CHAL || w.op > W_SHR) { ok = false; break; }
            const uint32_t rc[2][2] = {{w.ac, w.ar}, {w.bc, w.br}};
            for (int q = 0; q < n_reads(w) && ok; q++) {
                if (rc[q][0] == A) ok = rc[q][1] < ns && written[rc[q][1]];
                else ok = rc[q][1] >= r0 && rc[q][1] < r0 + P2_TRACE_ROWS && col_index(rc[q][0]) >= 0;
            }
            if (!ok) break;
            if (w.dc == A) {
                ok = w.dr < ns;
                if (ok) written[w.dr] = 1;
            } else {
                const int q = col_index(w.dc);
                ok = q >= 0 && w.dr > r0 && w.dr <= r0 + P2_TRACE_ROWS;
                if (ok) {
                    uint8_t& seen = written[ns + (size_t)q * P2_TRACE_ROWS + (w.dr - r0 - 1)];
                    ok = !seen;
                    seen = 1;
                    cells++;
                }
            }
        }
        if (!ok || cells != 8 * P2_TRACE_ROWS || n_cols != 8) { i++; continue; }
        const uint32_t hi = j;
        // what the range leaves in scratch must be dead
        {
            std::vector<uint8_t> dirty(written.begin(), written.begin() + ns);
            for (uint32_t t = hi; t < nops && ok; t++) {
                const WOp& w = P.wops[t];
                const uint32_t rc[2][2] = {{w.ac, w.ar}, {w.bc, w.br}};
                for (int q = 0; q < n_reads(w); q++)
                    if (rc[q][0] == A && rc[q][1] < ns && dirty[rc[q][1]]) ok = false;
                if (w.dc == A && w.dr < ns) dirty[w.dr] = 0;
            }
        }
        if (!ok) { i++; continue; }
        std::sort(cols, cols + 8);
        // the same values from both, on two random states
        grid.assign((size_t)A * n, Fr::zero());
        ref.assign((size_t)A * n, Fr::zero());
        uint32_t key[8] = {0x50325452u, i, r0, 0, 0, 0, 0, 0};
        for (uint32_t trial = 0; trial < 2 && ok; trial++) {
            for (uint32_t q = 0; q < 8; q++) {
                const Fr v = fr_draw(key, (uint64_t)trial * 8 + q);
                grid[(size_t)cols[q] * n + r0] = v;
                ref[(size_t)cols[q] * n + r0] = v;
            }
            WPermutation probe{i, hi, r0, {cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], cols[6], cols[7]}};
            std::vector<WPermutation> none;
            none.swap(P.wperms);                          // interpret: no native ranges while probing
            run_witness_range(P, no_inputs, ref.data(), scratch.data(), i, hi);
            none.swap(P.wperms);
            poseidon2_trace_host(grid.data(), n, probe.cols, r0);
            for (uint32_t q = 0; q < 8 && ok; q++)
                ok = memcmp(&grid[(size_t)cols[q] * n + r0], &ref[(size_t)cols[q] * n + r0], (P2_TRACE_ROWS + 1) * sizeof(Fr)) == 0;
        }
        if (!ok) { i++; continue; }
        P.wperms.push_back(WPermutation{i, hi, r0, {cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], cols[6], cols[7]}});
        i = hi;
    }
    if (getenv("SHP_PHASE_TRACE"))
        fprintf(stderr, "[witness] %zu Poseidon2 permutations of the program run natively (%u of %u ops)\n", P.wperms.size(),
                (uint32_t)(P.wperms.size() ? P.wperms.size() * (P.wperms[0].hi - P.wperms[0].lo) : 0), nops);
}

// ---- witness schedule: the program cut into tasks that may run on several host threads ------
// The witness program is a straight-line list of cell operations; at k = 13 it is ~40 k operations
// (19 Poseidon2 permutations) = 0.44 ms of a 3.5 ms proof on one core, and most of it is independent
// work (the 13 Merkle levels hash given inputs).  Built once per key:
//   * a scratch cell that is written and read later makes the ops in between one RUN; ops outside
//     any run are glue (copies, inputs);
//   * adjacent scratch runs that depend directly on each other (the rounds of one permutation) merge
//     into one task; glue stays separate, so a copy between two permutations does not chain them;
//   * tasks get levels from their advice-cell dependencies (RAW, WAR, WAW); a level's tasks are spread
//     over the threads by size.  Every task runs its ops in program order with a private scratch row,
//     so the values are those of the sequential program.
struct WitnessSchedule {
    struct Range { uint32_t lo, hi; };
    uint32_t n_threads = 1;
    bool serial_tail = false;                               // the last level runs on the calling thread in program order
    bool main_spared = false;                               // the calling thread has that tail only and joins no level's barrier
    std::vector<std::vector<std::vector<Range>>> plan;      // [level][thread] -> ranges

    // spare_main: the calling thread gets no tasks of its own (it has kernels to launch while the helpers work) -- only
    // the serial tail
    void build(const PlanData& P, uint32_t threads, bool spare_main = false) {
        n_threads = threads;
        serial_tail = false;
        main_spared = false;
        plan.clear();
        const uint32_t A = P.num_advice, n = P.n, nops = (uint32_t)P.wops.size();
        if (threads <= 1 || nops == 0) return;
        auto reads = [&](const WOp& w, uint32_t out[2][2]) -> int {
            switch (w.op) {
                case W_INPUT: case W_CONST: return 0;
                case W_ADD: case W_SUB: case W_MUL:
                    out[0][0] = w.ac; out[0][1] = w.ar; out[1][0] = w.bc; out[1][1] = w.br; return 2;
                default: out[0][0] = w.ac; out[0][1] = w.ar; return 1;
            }
        };
        // 1. runs: cover[i] = true when a scratch value is live across the boundary i-1 | i
        std::vector<int32_t> sc_writer(std::max<uint32_t>(P.n_wscratch, 1), -1);
        std::vector<int32_t> diff(nops + 1, 0);
        for (uint32_t i = 0; i < nops; i++) {
            const WOp& w = P.wops[i];
            uint32_t r[2][2];
            int nr = reads(w, r);
            for (int q = 0; q < nr; q++)
                if (r[q][0] == A && sc_writer[r[q][1]] >= 0) { diff[sc_writer[r[q][1]] + 1]++; diff[i + 1]--; }
            if (w.dc == A) sc_writer[w.dr] = (int32_t)i;
        }
        std::vector<uint8_t> joined(nops, 0);                  // op i is glued to op i-1 by live scratch
        int32_t live = 0;
        for (uint32_t i = 0; i < nops; i++) { live += diff[i]; joined[i] = live > 0; }
        // groups of ops glued by live scratch are scratch runs; every other op is a glue op of its own
        // (NOT merged with its neighbours: the copies between two permutations would chain them)
        struct Run { uint32_t lo, hi; bool scratch; };
        std::vector<Run> runs;
        for (uint32_t i = 0; i < nops;) {
            uint32_t j = i + 1;
            while (j < nops && joined[j]) j++;
            runs.push_back({i, j, j - i >= 2});
            i = j;
        }
        // 2. tasks + advice-cell dependencies
        struct Task { uint32_t lo, hi; bool scratch; uint32_t level; };
        std::vector<Task> tasks;
        std::vector<int32_t> last_writer((size_t)A * n, -1), last_reader((size_t)A * n, -1);
        bool hazard = false;
        auto deps_of = [&](uint32_t lo, uint32_t hi, std::vector<int32_t>& out) {
            out.clear();
            for (uint32_t i = lo; i < hi; i++) {
                const WOp& w = P.wops[i];
                uint32_t r[2][2];
                int nr = reads(w, r);
                for (int q = 0; q < nr; q++)
                    if (r[q][0] < A) { int32_t t = last_writer[(size_t)r[q][0] * n + r[q][1]]; if (t >= 0) out.push_back(t); }
                if (w.dc < A) {
                    // the schedule relies on single assignment of advice cells: a second write, or a write
                    // after a read of the still-empty cell, keeps the program sequential
                    size_t c = (size_t)w.dc * n + w.dr;
                    if (last_writer[c] >= 0 || last_reader[c] >= 0) hazard = true;
                }
            }
        };
        auto commit = [&](uint32_t lo, uint32_t hi, int32_t task) {
            for (uint32_t i = lo; i < hi; i++) {
                const WOp& w = P.wops[i];
                uint32_t r[2][2];
                int nr = reads(w, r);
                for (int q = 0; q < nr; q++)
                    if (r[q][0] < A) last_reader[(size_t)r[q][0] * n + r[q][1]] = task;
                if (w.dc < A) last_writer[(size_t)w.dc * n + w.dr] = task;
            }
        };
        std::vector<int32_t> deps;
        std::vector<std::vector<int32_t>> task_deps;
        for (const Run& r : runs) {
            const int32_t prev = (int32_t)tasks.size() - 1;
            deps_of(r.lo, r.hi, deps);
            const bool on_prev = std::find(deps.begin(), deps.end(), prev) != deps.end();
            if (prev >= 0 && r.scratch && tasks[prev].scratch && on_prev) {
                tasks[prev].hi = r.hi;
                for (int32_t d : deps) if (d != prev) task_deps[prev].push_back(d);
                commit(r.lo, r.hi, prev);
            } else {
                tasks.push_back({r.lo, r.hi, r.scratch, 0});
                task_deps.push_back(deps);
                commit(r.lo, r.hi, (int32_t)tasks.size() - 1);
            }
        }
        if (hazard) { n_threads = 1; return; }
        uint32_t n_levels = 0;
        for (size_t t = 0; t < tasks.size(); t++) {
            uint32_t lv = 0;
            for (int32_t d : task_deps[t]) if ((size_t)d != t) lv = std::max(lv, tasks[d].level + 1);
            tasks[t].level = lv;
            n_levels = std::max(n_levels, lv + 1);
        }
        // A program ends in a chain of glue (copies of the last hashes into their output cells, the range checks' limbs): a
        // handful of ops per level, and a level costs every thread a barrier.  From the first level on after which all
        // that is left is small, the rest runs on the calling thread in program order (an op's inputs come from earlier
        // ops), behind one barrier instead of ten.
        {
            std::vector<uint64_t> ops_at(n_levels, 0);
            for (auto& t : tasks) ops_at[t.level] += t.hi - t.lo;
            uint32_t tail = n_levels;
            uint64_t rest = 0;
            while (tail > 1 && rest + ops_at[tail - 1] <= 64) rest += ops_at[--tail];
            if (tail + 1 < n_levels) {
                for (auto& t : tasks) if (t.level > tail) t.level = tail;
                n_levels = tail + 1;
                serial_tail = true;
            }
        }
        // 3. per level: longest tasks first onto the least loaded thread
        main_spared = spare_main && threads > 1;
        plan.assign(n_levels, std::vector<std::vector<Range>>(threads));
        for (uint32_t lv = 0; lv < n_levels; lv++) {
            std::vector<const Task*> at;
            for (auto& t : tasks) if (t.level == lv) at.push_back(&t);
            if (serial_tail && lv + 1 == n_levels) {           // program order on thread 0
                std::sort(at.begin(), at.end(), [](const Task* a, const Task* b) { return a->lo < b->lo; });
                for (const Task* t : at) plan[lv][0].push_back({t->lo, t->hi});
                continue;
            }
            std::sort(at.begin(), at.end(), [](const Task* a, const Task* b) { return a->hi - a->lo > b->hi - b->lo; });
            std::vector<uint64_t> load(threads, 0);
            const uint32_t first = spare_main && threads > 1 ? 1 : 0;
            for (const Task* t : at) {
                uint32_t best = first;
                for (uint32_t th = first + 1; th < threads; th++) if (load[th] < load[best]) best = th;
                plan[lv][best].push_back({t->lo, t->hi});
                load[best] += t->hi - t->lo;
            }
            // program order inside a thread (tasks of one level are independent, this is for locality)
            for (auto& v : plan[lv]) std::sort(v.begin(), v.end(), [](const Range& a, const Range& b) { return a.lo < b.lo; });
            if (getenv("SHP_PHASE_TRACE")) {
                uint64_t mx = 0, sum = 0;
                for (uint64_t l : load) { mx = std::max(mx, l); sum += l; }
                fprintf(stderr, "[witness] level %u: %zu tasks, %llu ops, busiest thread %llu\n", lv, at.size(),
                        (unsigned long long)sum, (unsigned long long)mx);
            }
        }
    }
};

// helper threads of one proving lane for the witness schedule (the caller is thread 0)
struct WitnessWorkers {
    uint32_t T = 1;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> gen{0};
    std::atomic<uint64_t> arrivals{0};        // monotonic: job base + (level + 1) * T is a level's barrier
    bool stop = false;
    // the job
    const PlanData* P = nullptr;
    const std::vector<Fr>* inputs = nullptr;
    Fr* grid = nullptr;
    const WitnessSchedule* sched = nullptr;
    uint64_t base = 0;
    std::vector<std::vector<Fr>> scratch;

    WitnessWorkers(uint32_t threads, uint32_t n_scratch) : T(threads), scratch(threads, std::vector<Fr>(std::max<uint32_t>(n_scratch, 1), Fr::zero())) {
        for (uint32_t t = 1; t < T; t++) th.emplace_back([this, t] { worker(t); });
    }
    ~WitnessWorkers() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            gen.fetch_add(1);
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    // a level's barrier: a short busy wait (the levels of a proof are microseconds apart), then yielding
    void await(uint64_t target) {
        for (uint32_t spins = 0; arrivals.load(std::memory_order_acquire) < target; spins++) {
#if defined(__x86_64__)
            if (spins < 4096) { __builtin_ia32_pause(); continue; }
#endif
            std::this_thread::yield();
        }
    }
    void levels(uint32_t tid) {
        const uint32_t nl = (uint32_t)sched->plan.size();
        if (sched->main_spared) {
            // The calling thread has kernels to launch meanwhile: it joins no barrier (the helpers would wait for it at
            // the first one), waits for their last level and runs the serial tail, if any, itself.
            const uint32_t par = sched->serial_tail ? nl - 1 : nl, parts = T - 1;
            if (tid == 0) {
                await(base + (uint64_t)par * parts);
                if (sched->serial_tail)
                    for (const auto& r : sched->plan[nl - 1][0]) run_witness_range(*P, *inputs, grid, scratch[0].data(), r.lo, r.hi);
                return;
            }
            for (uint32_t lv = 0; lv < par; lv++) {
                for (const auto& r : sched->plan[lv][tid]) run_witness_range(*P, *inputs, grid, scratch[tid].data(), r.lo, r.hi);
                arrivals.fetch_add(1, std::memory_order_acq_rel);
                await(base + (uint64_t)(lv + 1) * parts);
            }
            return;
        }
        for (uint32_t lv = 0; lv < nl; lv++) {
            for (const auto& r : sched->plan[lv][tid]) run_witness_range(*P, *inputs, grid, scratch[tid].data(), r.lo, r.hi);
            arrivals.fetch_add(1, std::memory_order_acq_rel);
            await(base + (uint64_t)(lv + 1) * T);
        }
    }
    void worker(uint32_t tid) {
        uint64_t seen = 0;
        for (;;) {
            // spin for a while (back-to-back proofs), then sleep
            auto t0 = std::chrono::steady_clock::now();
            while (gen.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {   // longer than one proof
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
                    break;
                }
                std::this_thread::yield();
            }
            seen = gen.load(std::memory_order_acquire);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (stop) return;
            }
            levels(tid);
        }
    }
    // start: the helpers begin; finish: the calling thread does its share and leaves with the last level complete
    void start(const PlanData& P_, const std::vector<Fr>& in, Fr* g, const WitnessSchedule& sc) {
        P = &P_; inputs = &in; grid = g; sched = &sc;
        base = arrivals.load(std::memory_order_acquire);
        {
            std::lock_guard<std::mutex> lk(mu);
            gen.fetch_add(1, std::memory_order_acq_rel);
        }
        cv.notify_all();
    }
    void finish() { levels(0); }
    void run(const PlanData& P_, const std::vector<Fr>& in, Fr* g, const WitnessSchedule& sc) {
        start(P_, in, g, sc);
        finish();
    }
};

// ------------------------------------------------------------------------------ prover instance
struct shp_prover {
    shp_pk* pk = nullptr;
    hipStream_t st = nullptr;
    bool own_stream = false;
    MsmWorkspace msm_ws;
    PowTable pow_table = {};               // the opening points squared out (powers_kernel)
    MsmWorkspace msm_ws_rand;              // a lone proof commits to its random polynomial while the host still runs the witness program
    MsmWorkspace msm_ws_perm;              // ... and to its lookups' permuted columns, when those go ahead of the witness
    KernelTimers timers;
    uint64_t proofs_done = 0;
    // device workspace
    Fr *vals = nullptr, *polys = nullptr, *cos = nullptr;     // [NV + 1][n | n | ne]  (+1 = instance column)
    Fr *cin = nullptr, *ctab = nullptr;                        // [L][n] compressed lookup input / table
    Fr *num = nullptr, *den = nullptr;                         // [S + L][n]
    Fr *sort_raw = nullptr, *sort_out = nullptr;               // [2L][n]
    uint32_t* rep_scratch = nullptr;
    Fr *rand_poly = nullptr, *h = nullptr, *aux_ext = nullptr; // h [ne], aux_ext [2][ne]
    Fr *hx = nullptr, *lag2 = nullptr;                         // hx coeffs; [2][n] Lagrange values of hx, rand
    Fr* pw = nullptr;
    EvalJob* jobs = nullptr;                                   // fixed per prover (buffer addresses, point indices): uploaded once
    bool jobs_uploaded = false;
    Fr *fsets = nullptr, *dinv = nullptr, *wvals = nullptr;
    LincombArgs* lin = nullptr;                                // the rotation sets' polynomials: fixed per prover, uploaded once
    bool lin_uploaded = false;
    Fr* ntt_tmp = nullptr;
    Fr* h_grid = nullptr;                                      // pinned host staging of the witness grid
    // The grid and the device columns are cleared once; the witness program rewrites the same cells every proof.  A proof
    // from caller-assigned columns (shp_prove_assigned) writes other cells: grid_foreign makes the next witness-program
    // proof clear the grid first, dev_rows (leading rows of the device columns that may be non-zero) lets every proof
    // clear what the previous one left beyond its own rows.
    bool grid_foreign = false;
    uint32_t dev_rows = 0;
    PinnedBuf h_rand;                                          // caller-supplied coefficients of the random polynomial
    WitnessSchedule wsched;                                    // witness program cut into parallel tasks
    std::unique_ptr<WitnessWorkers> wworkers;
    std::vector<Fr> wscratch;                                  // scratch row of the sequential path
    uint32_t* rank = nullptr;                                  // [2L][n] rank-sort scratch
    Fr *sc_pre = nullptr, *sc_suf = nullptr;                   // [SC][n] block-local scans
    Fr *sc_tot = nullptr;                                      // [6][SC][nb]: tot_a, tot_b, bpre_a, bsuf_b | [SC] inv
    uint32_t sc_cnt = 0;
    uint32_t* sc_done = nullptr;                               // [SC] workgroups of a product scan that have stored their totals
    uint32_t* lk_done = nullptr;                               // [L] flag workgroups of a lookup that have finished
    std::vector<void*> owned;
    template <class T>
    T* dalloc(size_t count) {
        void* p = nullptr;
        SHP_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
        owned.push_back(p);
        return (T*)p;
    }
    // a pool's cross-lane MSM combiner (null: this lane launches its own commitments) and the event that hands
    // the lane's stream position over to it
    Fr* consts_lane = nullptr;             // several phases: the key's constants + this proof's challenges
    // A lone proof builds the permuted columns of its single-expression lookups WHILE the host still runs the rest of the
    // witness program: the few witness ops those lookups' input columns depend on (early_ranges, program order) run first,
    // the columns (early_cols) go to the device and the lookup kernels run on a second stream, beside the random
    // polynomial's commitment.  Empty ranges: the plan does not allow it (see early_lookup_ops).
    hipStream_t st2 = nullptr;
    hipEvent_t ev2 = nullptr;
    // The coefficient forms and cosets of the columns that are final when the advice commitment is launched, on the second
    // stream while the first one commits and builds the grand products (a lone proof): ev_cols = those columns are on the
    // device, ev_tr = their cosets are.  ntt_tmp2 is that stream's transform scratch.
    hipEvent_t ev_cols = nullptr, ev_tr = nullptr;
    hipEvent_t ev_in = nullptr;                    // the second stream has read the advice columns it took ahead
    Fr* ntt_tmp2 = nullptr;
    void second_stream() {                         // created on first use: pool lanes never need it
        if (!st2) SHP_HIP(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));   // default priority
        if (!ev2) SHP_HIP(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
        if (!ev_cols) SHP_HIP(hipEventCreateWithFlags(&ev_cols, hipEventDisableTiming));
        if (!ev_tr) SHP_HIP(hipEventCreateWithFlags(&ev_tr, hipEventDisableTiming));
        if (!ev_in) SHP_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    }
    std::vector<WitnessSchedule::Range> early_ranges;
    std::vector<uint32_t> early_cols;
    bool throughput = false;               // shp_prover_mode(1): scheduling choices of a pool lane
    bool keep_phases = false;
    std::vector<std::pair<const char*, double>> last_phases;   // host phase clock of the last proof (profiling on)
    shp::MsmCombiner* comb = nullptr;
    bool comb_profile_owner = false;
    hipEvent_t comb_event = nullptr;
    // page-locked staging for every small host<->device copy of a proof (copies from pageable memory take the
    // runtime's slow path and serialise concurrent lanes); bump-allocated, rewound at the start of each proof
    PinnedBuf stage;
    size_t stage_off = 0;
    uint32_t stamp_seq = 0;          // sequence numbers of the results a lone proof waits for in pinned memory (wait_stamps)
    void* stage_alloc(size_t bytes) {
        const size_t at = (stage_off + 63) & ~(size_t)63;
        SHP_REQUIRE(at + bytes <= stage.cap, SHP_ERR_INTERNAL, "prover: staging arena exhausted");
        stage_off = at + bytes;
        return (uint8_t*)stage.p + at;
    }
    ~shp_prover() {
        if (comb_event) (void)hipEventDestroy(comb_event);
        if (ev2) (void)hipEventDestroy(ev2);
        if (ev_cols) (void)hipEventDestroy(ev_cols);
        if (ev_tr) (void)hipEventDestroy(ev_tr);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (st2) (void)hipStreamDestroy(st2);
        stage.release();
        h_rand.release();
        for (void* p : owned) (void)hipFree(p);
        msm_ws.release();
        msm_ws_rand.release();
        msm_ws_perm.release();
        if (h_grid) (void)hipHostFree(h_grid);
        if (own_stream && st) (void)hipStreamDestroy(st);
    }
};

// The witness ops that the input columns of the plan's lookups depend on, as ranges in program order, and those columns.
// Nothing (empty ranges) unless every lookup is over single expressions without public inputs, the plan has one phase,
// and the ops are few (range-check limbs come from the inputs through a handful of shifts and masks, not through the
// hashes): then running them ahead of the program costs microseconds and hides the lookups' sort and permutation.
static void early_lookup_ops(const PlanData& P, std::vector<WitnessSchedule::Range>& ranges, std::vector<uint32_t>& cols) {
    ranges.clear();
    cols.clear();
    if (P.phased() || P.lookups.empty() || P.wops.empty()) return;
    const uint32_t A = P.num_advice, n = P.n;
    std::vector<uint8_t> is_col(A, 0);
    for (const LookupPlan& lk : P.lookups) {
        if (lk.m != 1) return;
        for (const auto* prog : {&lk.in_prog, &lk.tab_prog})
            for (const Instr& in : *prog) {
                if (in.op == OP_INST || in.op == OP_CHAL) return;
                if (in.op == OP_ADV) is_col[P.adv_q[in.a].col] = 1;
            }
    }
    // backwards over the program: an op is kept when something kept later reads what it writes
    const size_t cells = (size_t)A * n + std::max<uint32_t>(P.n_wscratch, 1);
    std::vector<uint8_t> needed(cells, 0);
    for (uint32_t c = 0; c < A; c++)
        if (is_col[c]) std::fill(needed.begin() + (size_t)c * n, needed.begin() + (size_t)(c + 1) * n, 1);
    auto id = [&](uint32_t c, uint32_t r) { return (size_t)c * n + r; };   // the scratch row is "column" A
    std::vector<uint8_t> keep(P.wops.size(), 0);
    size_t kept = 0;
    for (size_t i = P.wops.size(); i-- > 0;) {
        const WOp& w = P.wops[i];
        const size_t d = id(w.dc, w.dr);
        if (!needed[d]) continue;
        keep[i] = 1;
        kept++;
        needed[d] = 0;                                   // what was written before this op is another value
        switch (w.op) {
            case W_INPUT: case W_CONST: case W_CHAL: break;
            case W_ADD: case W_SUB: case W_MUL: needed[id(w.ac, w.ar)] = 1; needed[id(w.bc, w.br)] = 1; break;
            default: needed[id(w.ac, w.ar)] = 1; break;
        }
    }
    if (kept == 0 || kept > 4096 || kept * 8 > P.wops.size()) return;     // not cheap: leave the program alone
    for (uint32_t i = 0; i < P.wops.size();) {
        if (!keep[i]) { i++; continue; }
        uint32_t j = i + 1;
        while (j < P.wops.size() && keep[j]) j++;
        ranges.push_back({i, j});
        i = j;
    }
    for (uint32_t c = 0; c < A; c++)
        if (is_col[c]) cols.push_back(c);
    if (getenv("SHP_PHASE_TRACE"))
        fprintf(stderr, "[witness] ahead of the program: %zu of %zu ops in %zu runs feed the lookups' %zu columns\n", kept,
                P.wops.size(), ranges.size(), cols.size());
}

static void set_witness_threads(shp_prover* pr, uint32_t threads) {
    threads = std::max<uint32_t>(1, std::min<uint32_t>(threads, 16));
    pr->wworkers.reset();
    if (pr->pk->plan.phased()) return;     // a phase's part of the program runs in order on the calling thread
    pr->wsched.build(pr->pk->plan, threads, !pr->early_ranges.empty() && threads >= 3);
    if (pr->wsched.n_threads > 1 && !pr->wsched.plan.empty())
        pr->wworkers.reset(new WitnessWorkers(pr->wsched.n_threads, pr->pk->plan.n_wscratch));
}

// SHP_POLL_RESULTS=0: latency lanes wait for the stream, as pool lanes do, instead of spinning on result stamps
static bool poll_results_enabled() {
    static const bool on = !(getenv("SHP_POLL_RESULTS") && atoi(getenv("SHP_POLL_RESULTS")) == 0);
    return on;
}

static void prover_create_impl(shp_pk* pk, shp_prover** out) {
    SHP_REQUIRE(pk && out, SHP_ERR_INVALID_ARG, "prover_create: null argument");
    SHP_HIP(hipSetDevice(pk->ctx->device));          // contexts of several GPUs may live in one process
    std::unique_ptr<shp_prover> pr(new shp_prover);
    pr->pk = pk;
    SHP_HIP(hipStreamCreateWithFlags(&pr->st, hipStreamNonBlocking));
    pr->own_stream = true;
    const PlanData& P = pk->plan;
    const uint32_t n = pk->n, ne = pk->nq, NV = pk->NV, L = (uint32_t)P.lookups.size(), S = P.num_sets;   // ne: quotient domain
    pr->vals = pr->dalloc<Fr>((size_t)(NV + 1) * n);
    // unassigned witness rows are never uploaded; on the prover's own stream (non-blocking: the null stream's
    // memset would not be ordered before the first proof's uploads)
    SHP_HIP(hipMemsetAsync(pr->vals, 0, (size_t)(NV + 1) * n * sizeof(Fr), pr->st));
    pr->polys = pr->dalloc<Fr>((size_t)(NV + 1) * n);
    pr->cos = pr->dalloc<Fr>((size_t)(NV + 1) * ne);
    pr->cin = pr->dalloc<Fr>((size_t)L * n);
    pr->ctab = pr->dalloc<Fr>((size_t)L * n);
    pr->num = pr->dalloc<Fr>((size_t)(S + L) * n);
    pr->den = pr->dalloc<Fr>((size_t)(S + L) * n);
    pr->sort_raw = pr->dalloc<Fr>((size_t)2 * L * n);
    pr->sort_out = pr->dalloc<Fr>((size_t)2 * L * n);
    pr->rep_scratch = pr->dalloc<uint32_t>((size_t)L * n);
    pr->h = pr->dalloc<Fr>(ne);
    pr->aux_ext = pr->dalloc<Fr>((size_t)2 * ne);
    pr->hx = pr->dalloc<Fr>((size_t)2 * n);          // h(X) folded at x, then the random polynomial: one NTT batch
    pr->rand_poly = pr->hx + n;
    pr->lag2 = pr->dalloc<Fr>((size_t)2 * n);
    pr->pw = pr->dalloc<Fr>((size_t)MAX_POINTS * n);
    pr->jobs = pr->dalloc<EvalJob>(512);
    pr->fsets = pr->dalloc<Fr>((size_t)MAX_SETS * n);
    pr->dinv = pr->dalloc<Fr>((size_t)MAX_POINTS * n);
    pr->wvals = pr->dalloc<Fr>((size_t)2 * n);
    pr->lin = pr->dalloc<LincombArgs>(MAX_SETS);
    pr->ntt_tmp = pr->dalloc<Fr>((size_t)(NV + 1) * ne);
    pr->rank = pr->dalloc<uint32_t>((size_t)2 * L * n);
    pr->stage.reserve((size_t)1 << 20);
    SHP_HIP(hipHostMalloc((void**)&pr->h_grid, (size_t)(P.num_advice + 1) * n * sizeof(Fr), hipHostMallocDefault));
    memset(pr->h_grid, 0, (size_t)(P.num_advice + 1) * n * sizeof(Fr));
    pr->sc_cnt = std::max<uint32_t>(S + L, MAX_POINTS);
    pr->sc_pre = pr->dalloc<Fr>((size_t)pr->sc_cnt * n);
    pr->sc_suf = pr->dalloc<Fr>((size_t)pr->sc_cnt * n);
    pr->sc_tot = pr->dalloc<Fr>((size_t)5 * pr->sc_cnt * SCAN_THREADS);
    pr->sc_done = pr->dalloc<uint32_t>(pr->sc_cnt);            // zero between launches: the last workgroup resets its counter
    SHP_HIP(hipMemset(pr->sc_done, 0, pr->sc_cnt * sizeof(uint32_t)));
    pr->lk_done = pr->dalloc<uint32_t>(std::max<size_t>(P.lookups.size(), 1));
    SHP_HIP(hipMemset(pr->lk_done, 0, std::max<size_t>(P.lookups.size(), 1) * sizeof(uint32_t)));
    pr->msm_ws.timers = &pr->timers;
    pr->msm_ws.poll_results = poll_results_enabled();       // a new lane is a latency lane until shp_prover_mode says otherwise
    pr->wscratch.assign(std::max<uint32_t>(P.n_wscratch, 1), Fr::zero());
    early_lookup_ops(P, pr->early_ranges, pr->early_cols);
    // (no second stream yet: a pool lane never uses one, and every stream a process creates shifts the mapping of the
    // lanes' streams onto the hardware queues -- one idle stream per lane cost the pool 5 % and the interpreted plans 20 %)
    if (P.phased()) {
        const size_t cnt = P.consts.size() + P.challenge_phase.size();
        pr->consts_lane = pr->dalloc<Fr>(cnt);
        SHP_HIP(hipMemcpyAsync(pr->consts_lane, pk->consts, cnt * sizeof(Fr), hipMemcpyDeviceToDevice, pr->st));
    }
    {
        const char* e = getenv("SHP_WITNESS_THREADS");
        set_witness_threads(pr.get(), e && *e ? (uint32_t)atoi(e) : 8);   // 14 independent permutations: 215 us at 4, 175 at 8
    }
    *out = pr.release();
}

// ------------------------------------------------------------------------------ host pieces
// coefficients (low to high) of the polynomial through (xs, ys)
static std::vector<Fr> interpolate(const std::vector<Fr>& xs, const std::vector<Fr>& ys) {
    size_t m = xs.size();
    std::vector<Fr> out(m, Fr::zero());
    std::vector<std::vector<Fr>> nums(m);
    std::vector<Fr> den(m), pre(m);
    Fr acc = Fr::one();
    for (size_t i = 0; i < m; i++) {
        std::vector<Fr> num(1, Fr::one());
        Fr d = Fr::one();
        for (size_t j = 0; j < m; j++) {
            if (i == j) continue;
            std::vector<Fr> nx(num.size() + 1, Fr::zero());
            for (size_t t = 0; t < num.size(); t++) {
                nx[t + 1] = nx[t + 1] + num[t];
                nx[t] = nx[t] - xs[j] * num[t];
            }
            num.swap(nx);
            d = d * (xs[i] - xs[j]);
        }
        nums[i].swap(num);
        den[i] = d;
        pre[i] = acc;
        acc = acc * d;
    }
    Fr inv = Fr::inv(acc);                    // one inversion for the m denominators (distinct points: acc != 0)
    for (size_t i = m; i-- > 0;) {
        Fr sc = ys[i] * (inv * pre[i]);
        inv = inv * den[i];
        for (size_t t = 0; t < m; t++) out[t] = out[t] + sc * nums[i][t];
    }
    return out;
}

// host-side phase clock (SHP_PHASE_TRACE=1): wall time between phase marks, printed per proof
namespace {
struct PhaseClock {
    bool on, print;
    std::chrono::steady_clock::time_point t0, last;
    std::vector<std::pair<const char*, double>> marks;
    std::vector<std::pair<const char*, double>>* keep;      // the prover's copy of the last proof's phases (profiling on)
    double wait0;
    explicit PhaseClock(std::vector<std::pair<const char*, double>>* keep_ = nullptr)
        : print(getenv("SHP_PHASE_TRACE") != nullptr), keep(keep_), wait0(shp::wait_us_accum()) {
        on = print || keep;
        t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char* name) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        marks.push_back({name, std::chrono::duration<double, std::micro>(now - last).count()});
        last = now;
    }
    ~PhaseClock() {
        if (keep) *keep = marks;
        if (!print) return;
        double tot = std::chrono::duration<double, std::micro>(last - t0).count();
        fprintf(stderr, "[phases] total %.0f us (%.0f waiting for the stream):", tot, shp::wait_us_accum() - wait0);
        for (auto& m : marks) fprintf(stderr, " %s=%.0f", m.first, m.second);
        fprintf(stderr, "\n");
    }
};
}  // namespace

// ------------------------------------------------------------------------------ create_proof
// What a proof is made from.  Witness: either the plan's witness-program inputs (32-byte LE canonical each), or the advice
// columns a caller synthesised itself (the reference runs the circuit's `synthesize`) plus the public inputs.
// Randomness (the reference's `rng`): either a 32-byte seed expanded with ChaCha20, or the draws themselves.
struct ProveInput {
    const uint8_t* inputs_le32 = nullptr;    // witness program: n_inputs x 32 bytes
    size_t n_inputs = 0;
    const Fr* advice = nullptr;              // assigned columns: [A][advice_stride] Montgomery, host memory
    size_t advice_stride = 0;                //   elements between columns (0: n)
    uint32_t assigned_rows = 0;              //   rows [assigned_rows, usable) of every column are zero and not read
    const Fr* instances = nullptr;           //   public inputs, Montgomery
    size_t n_instances = 0;
    // several phases: the caller synthesises phase by phase (halo2 calls `synthesize` once per phase with the challenges
    // squeezed so far); returns its [A][n] columns, of which those of `phase` are read, or null to give up
    const uint64_t* (*phase_fn)(void* user, uint32_t phase, const uint64_t* challenges, size_t n_challenges) = nullptr;
    void* phase_user = nullptr;
    const uint8_t* seed = nullptr;           // 32 bytes, or
    const Fr* draws = nullptr;               // n_draws field elements in draw order (Montgomery)
    size_t n_draws = 0;
};

// Fr::random draws one proof consumes, in this order: per phase (a circuit without phases has one), per advice column of the
// phase its bf+1 blinding rows (column-major), then one unused commitment blind per column of the phase; per lookup bf+1 permuted-input rows, bf+1 permuted-table rows, 2 unused blinds; per grand product
// (permutation sets, then lookups) bf rows + 1 unused blind; the n coefficients of the random polynomial + 1 unused blind;
// one unused blind per quotient piece.
static uint64_t randomness_len(const PlanData& P) {
    const uint64_t A = P.num_advice, L = P.lookups.size(), S = P.num_sets, nb = P.bf + 1;
    return A * nb + A + L * (2 * nb + 2) + (S + L) * (P.bf + 1) + P.n + 1 + (P.degree - 1);
}

static void prove_impl(shp_prover* pr, const ProveInput& in, std::vector<uint8_t>& proof_out, std::vector<Fr>& instances_out) {
    const uint8_t* const inputs_le32 = in.inputs_le32;
    const size_t n_inputs = in.n_inputs;
    shp_pk* pk = pr->pk;
    shp_ctx* ctx = pk->ctx;
    shp_srs* srs = pk->srs;
    const PlanData& P = pk->plan;
    hipStream_t st = pr->st;
    SHP_HIP(hipSetDevice(ctx->device));                 // the caller may be any host thread (pool workers, FFI callers)
    const uint32_t n = pk->n, ne = pk->nq, k = P.k, bf = P.bf, u = P.usable;   // ne = nc * n: the quotient domain
    const uint32_t A = P.num_advice, L = (uint32_t)P.lookups.size(), S = P.num_sets, NV = pk->NV;
    const uint32_t PC = (uint32_t)P.perm_cols.size(), F = P.num_fixed;
    const uint32_t nb = bf + 1;
    const bool assigned = in.advice != nullptr || in.phase_fn != nullptr;
    const bool phased = P.phased();
    SHP_REQUIRE(!(phased && in.advice && P.num_phases > 1), SHP_ERR_INVALID_ARG,
                "prove_assigned: a circuit with several phases is synthesised phase by phase (shp_prove_assigned_phased)");
    if (assigned) {
        SHP_REQUIRE(in.assigned_rows <= u, SHP_ERR_INVALID_ARG, "prove_assigned: more assigned rows than usable rows");
        SHP_REQUIRE(!in.advice_stride || (in.assigned_rows && in.advice_stride >= in.assigned_rows && !in.phase_fn), SHP_ERR_INVALID_ARG,
                    "prove_assigned: compact columns hold exactly their assigned rows");
        SHP_REQUIRE(in.n_instances == P.num_instance && (in.instances || !in.n_instances), SHP_ERR_INVALID_ARG,
                    "prove_assigned: wrong number of public inputs");
    } else {
        SHP_REQUIRE(inputs_le32 && n_inputs == P.n_winputs, SHP_ERR_INVALID_ARG, "prove: wrong number of witness inputs");
    }
    SHP_REQUIRE((in.seed != nullptr) != (in.draws != nullptr), SHP_ERR_INVALID_ARG, "prove: pass a seed or a randomness buffer");
    if (in.draws) {
        SHP_REQUIRE(in.n_draws == randomness_len(P), SHP_ERR_INVALID_ARG, "prove: randomness buffer of the wrong length (shp_pk_randomness_len)");
        for (size_t i = 0; i < in.n_draws; i++)
            SHP_REQUIRE(!in.draws[i].geq_p(), SHP_ERR_FORMAT, "prove: randomness buffer holds a non-canonical field element");
    }
    const uint32_t I_PIN = A, I_PZ = A + 2 * L, I_LZ = A + 2 * L + S, I_INST = NV;
    auto V = [&](uint32_t idx) { return pr->vals + (size_t)idx * n; };
    auto PL = [&](uint32_t idx) { return pr->polys + (size_t)idx * n; };
    auto CS = [&](uint32_t idx) { return pr->cos + (size_t)idx * ne; };
    // product scans over `cnt` vectors of n elements (prover_kernels.hip.hpp "product scans")
    // elements per thread: a lone proof keeps the chains short, pool lanes (the GPU is full anyway) save two thirds of the products
    const uint32_t sce = pr->throughput ? SCAN_E_POOL : 1;
    const uint32_t sbs = std::min<uint32_t>(SCAN_THREADS, n / sce), snb = n / (sbs * sce);
    Fr* const sc_ta = pr->sc_tot;
    Fr* const sc_tb = pr->sc_tot + (size_t)pr->sc_cnt * SCAN_THREADS;
    Fr* const sc_bpre = pr->sc_tot + (size_t)2 * pr->sc_cnt * SCAN_THREADS;
    Fr* const sc_bsuf = pr->sc_tot + (size_t)3 * pr->sc_cnt * SCAN_THREADS;
    Fr* const sc_inv = pr->sc_tot + (size_t)4 * pr->sc_cnt * SCAN_THREADS;
    // A lone proof (latency mode) inverts the totals on the host: the product goes to pinned memory, the stream is
    // drained, the inverses come back as kernel arguments.  Pool lanes keep the inversion on the device (one lane,
    // 55 us, but no extra round trip while fifteen other lanes fill the GPU).
    // diff_points non-null: product p runs over omega^r - diff_points[p] (a and b are not read)
    // host_points (with diff_points): the same points on the host.  Then nobody waits for the total: the product of
    // omega^r - p over the whole domain is p^n - 1 (n even), which the host has inverted before the kernel starts.
    auto product_scans = [&](const Fr* a, const Fr* b, uint32_t cnt, const Fr* diff_points = nullptr,
                             const Fr* host_points = nullptr, hipStream_t on = nullptr) -> InvList {
        hipStream_t st = on ? on : pr->st;                // (shadows the lane's stream: a caller may name another one)
        InvList il;
        il.n = 0;
        const bool closed_form = diff_points && host_points && cnt <= INV_LIST_MAX && n >= 2;
        const bool on_host = closed_form || (!pr->throughput && cnt <= INV_LIST_MAX);
        Fr* totals = on_host ? (Fr*)pr->stage_alloc(cnt * sizeof(Fr)) : nullptr;
        ScanTotals T;
        T.bpre_a = sc_bpre; T.bsuf_b = sc_bsuf; T.inv_b = sc_inv; T.total_out = totals; T.done = pr->sc_done;
        T.omegas = diff_points ? pk->omegas : nullptr; T.points = diff_points;
        const bool poll = on_host && !closed_form && !pr->timers.enabled && pr->msm_ws.poll_results;
        T.stamp = poll ? (uint32_t*)pr->stage_alloc(cnt * sizeof(uint32_t)) : nullptr;
        T.seq = poll ? (++pr->stamp_seq ? pr->stamp_seq : ++pr->stamp_seq) : 0;
        for (uint32_t i = 0; poll && i < cnt; i++) T.stamp[i] = T.seq - 1;
        if (sce == 1 && snb <= 64 && sbs >= 128)            // a lone proof: one direction per workgroup, half the chain (two waves scan the totals)
            hipLaunchKernelGGL(scan_split_kernel, dim3(snb, cnt, 2), dim3(sbs), 0, st, a, b, n, pr->sc_pre, pr->sc_suf, sc_ta, sc_tb, T);
        else
            hipLaunchKernelGGL(sce == 1 ? scan_local_kernel<1> : scan_local_kernel<SCAN_E_POOL>, dim3(snb, cnt), dim3(sbs), 0, st, a,
                               b, n, pr->sc_pre, pr->sc_suf, sc_ta, sc_tb, T);
        if (closed_form) {
            il.n = cnt;
            for (uint32_t i = 0; i < cnt; i++) {
                const Fr tot = Fr::pow_u64(host_points[i], n) - Fr::one();
                SHP_REQUIRE(!tot.is_zero(), SHP_ERR_INTERNAL, "prove: an opening point lies on the domain");   // probability n 2^-254
                il.v[i] = Fr::inv(tot);
            }
        } else if (on_host) {
            if (poll) wait_stamps(T.stamp, cnt, T.seq, st);
            else sync_stream(st);
            il.n = cnt;
            for (uint32_t i = 0; i < cnt; i++) il.v[i] = Fr::inv(totals[i]);
        }
        return il;
    };
    // out[p][r] = 1 / (omega^r - points[p])
    auto invert_point_diffs = [&](const Fr* points, const Fr* host_points, Fr* out, uint32_t cnt, hipStream_t on = nullptr) {
        hipStream_t st = on ? on : pr->st;
        pr->timers.begin(KID_BATCH_INVERT, st);
        const InvList il = product_scans(nullptr, nullptr, cnt, points, host_points, st);
        hipLaunchKernelGGL(sce == 1 ? batch_invert_apply_kernel<1> : batch_invert_apply_kernel<SCAN_E_POOL>, dim3(snb, cnt),
                           dim3(sbs), 0, st, (const Fr*)nullptr, pk->omegas, points, pr->sc_pre, pr->sc_suf, sc_bpre, sc_bsuf,
                           sc_inv, n, out, il);
        pr->timers.end(st);
    };

    uint32_t key[8] = {0};
    if (in.seed) memcpy(key, in.seed, 32);
    uint64_t ctr = 0;
    auto draw = [&]() { return in.draws ? in.draws[ctr++] : fr_draw(key, ctr++); };

    PhaseClock pc(pr->keep_phases ? &pr->last_phases : nullptr);
    pr->stage_off = 0;                                  // the previous proof's copies have all completed
    auto h2d = [&](void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        void* sbuf = pr->stage_alloc(bytes);
        memcpy(sbuf, src, bytes);
        SHP_HIP(hipMemcpyAsync(dst, sbuf, bytes, hipMemcpyHostToDevice, st));
    };
    // Small inputs that every consuming thread reads at most once (blinding values, evaluation points) are not uploaded
    // at all: the kernel reads them from the pinned arena over the host link.  An upload is a dispatch of its own, and
    // on a busy chip every dispatch costs its stream tens of microseconds whatever its size.
    auto staged = [&](const void* src, size_t bytes) -> const void* {
        void* sbuf = pr->stage_alloc(std::max<size_t>(bytes, 1));
        memcpy(sbuf, src, bytes);
        return sbuf;
    };
    // commitments: `batch` columns over one basis (optionally a second job over another), through the pool's
    // combiner when this lane has one; either way the lane's stream has drained when the points are back
    auto commit = [&](shp_msm* m, const Fr* scalars, uint32_t batch, G1Affine* out, const MsmRows* rows = nullptr,
                      shp_msm* m2 = nullptr, const Fr* scalars2 = nullptr, uint32_t batch2 = 0) {
        if (batch + batch2 == 0) return;
        if (pr->comb && m->direct && (!batch2 || m2->direct)) {
            MsmJob jobs[2];
            jobs[0] = MsmJob{m, scalars, n, batch, rows ? *rows : MsmRows{n, 0, 0}, out};
            if (batch2) jobs[1] = MsmJob{m2, scalars2, n, batch2, MsmRows{n, 0, 0}, out + batch};
            pr->comb->run(jobs, batch2 ? 2 : 1, st, pr->comb_event);
        } else if (batch2) {
            msm_run_ws2(m, scalars, n, batch, m2, scalars2, n, batch2, n, pr->msm_ws, out, st, rows);
        } else {
            msm_run_ws(m, pr->msm_ws, scalars, n, n, batch, out, st);
        }
    };
    // a proof that failed between a launch and its wait leaves the launch behind: drain it
    if (pr->msm_ws.pend_total || pr->msm_ws_rand.pend_total) {
        sync_stream(st);
        pr->msm_ws.pend_total = pr->msm_ws_rand.pend_total = 0;
    }
    if (pr->msm_ws_perm.pend_total) {
        sync_stream(pr->st2);
        pr->msm_ws_perm.pend_total = 0;
    }
    // Commitments in two halves (msm_launch_jobs / msm_finish_jobs) for a lone proof's lane: work that does not depend on the
    // next challenge is queued BEHIND the commitment kernel and runs while the host waits for the points (their stamps in
    // pinned memory, not the stream), normalises and hashes them.  Pool lanes, lanes behind a combiner and profiled lanes
    // commit in one piece.
    const bool split_commits = pr->msm_ws.poll_results && !pr->comb && !pr->timers.enabled && srs->msm_gl->direct;
    // P9's transforms: coefficient form, then the nc cosets of the quotient domain, of every column.  (Cut into column
    // ranges and queued behind the earlier commitments as well -- advice columns are final after P2 -- they filled those
    // waits, but a two-pass transform launch has a floor of 20 us whatever its size: what the waits gained the smaller
    // launches lost again, measured.)
    NttOpts ntt_fwd;
    NttOpts ntt_inv;
    ntt_inv.inverse = true;
    NttOpts ntt_ext;                                     // values on the nc cosets of the quotient domain
    ntt_ext.twist = pk->twist; ntt_ext.twist_count = pk->nc;
    ntt_ext.coset_tw = pk->coset_tw; ntt_ext.coset_g = pk->coset_g; ntt_ext.coset_scaled = pk->quot_fn != nullptr;
    uint32_t inv_done = 0, ext_done = 0;                 // columns [0, inv_done) have coefficients, [0, ext_done) cosets
    bool transforms_ahead = false;                       // ... the leading ones of them on the second stream (ev_tr)
    auto inv_upto = [&](uint32_t c) {
        if (c <= inv_done) return;
        ntt_launch(ctx, pr->vals + (size_t)inv_done * n, pr->polys + (size_t)inv_done * n, k, c - inv_done, ntt_inv, st, pr->ntt_tmp);
        inv_done = c;
    };
    auto ext_upto = [&](uint32_t c) {
        if (c <= ext_done) return;
        inv_upto(c);
        ntt_launch(ctx, pr->polys + (size_t)ext_done * n, pr->cos + (size_t)ext_done * ne, k, (c - ext_done) * pk->nc, ntt_ext, st,
                   pr->ntt_tmp);
        ext_done = c;
    };
    auto commit_launch = [&](MsmWorkspace& ws, shp_msm* m, const Fr* scalars, uint32_t batch, G1Affine* out, const MsmRows* rows = nullptr) {
        MsmJob job{m, scalars, n, batch, rows ? *rows : MsmRows{n, 0, 0}, out};
        ws.poll_results = true;
        msm_launch_jobs(&job, 1, ws, st);
    };
    // ---- P7 first: the random polynomial depends on no challenge and not on the witness -- its coefficients (blinding stream
    // position known up front) are drawn on the device and its commitment runs while the host executes the witness program
    const uint64_t ctr_rand = (uint64_t)A * nb + A + (uint64_t)L * (2 * nb + 2) + (uint64_t)(S + L) * (bf + 1);
    G1Affine rand_comm;
    auto ingest = [&](uint32_t y0, uint32_t rows, uint32_t rows_now_, uint32_t n_inst, hipStream_t on = nullptr) {
        IngestArgs ia;
        ia.grid = pr->h_grid; ia.vals = V(0); ia.inst = V(I_INST); ia.rand_poly = pr->rand_poly; ia.h_rand = nullptr;
        ia.n = n; ia.n_adv = A; ia.rows_now = rows_now_; ia.u = u; ia.n_inst = n_inst; ia.y0 = y0;
        memcpy(ia.key.w, key, 32);
        ia.counter0 = ctr_rand;
        if (in.draws && y0 + rows > A + 1) {
            pr->h_rand.reserve((size_t)n * sizeof(Fr));
            memcpy(pr->h_rand.p, in.draws + ctr_rand, (size_t)n * sizeof(Fr));
            ia.h_rand = pr->h_rand.as<Fr>();
        }
        hipLaunchKernelGGL(ingest_kernel, dim3(ceil_div(n, 256), rows), dim3(256), 0, on ? on : st, ia);
    };
    // Every commitment of a proof goes through the LAGRANGE basis and its one multiples table: the polynomials that exist
    // as coefficients (this one, the quotient pieces) are transformed to their values on the domain first -- the same group
    // element, 17 windows instead of the 19 a second table of affordable size would have, and no second table (82 GB).
    Fr* const rand_lag = pr->lag2 + n;                   // also what the evaluation-form opening reads later
    auto rand_values = [&] { ntt_launch(ctx, pr->rand_poly, rand_lag, k, 1, ntt_fwd, st, pr->ntt_tmp); };
    bool rand_launched = false;
    auto launch_rand = [&] {
        if (!split_commits || rand_launched) return;
        ingest(A + 1, 1, 0, 0);
        rand_values();
        commit_launch(pr->msm_ws_rand, srs->msm_gl, rand_lag, 1, &rand_comm);
        rand_launched = true;
    };
    // (a proof whose lookups go ahead of the witness queues their kernels first: they are what the advice commitment will
    // wait for, this commitment only has to be there by the grand products')
    bool lookups_may_go_ahead = !assigned && !phased && !pr->early_ranges.empty() && P.adv_rows >= pr->dev_rows;
    {
        static const bool off = getenv("SHP_EARLY_LOOKUPS") && atoi(getenv("SHP_EARLY_LOOKUPS")) == 0;
        if (off) lookups_may_go_ahead = false;
    }
    if (!lookups_may_go_ahead) launch_rand();
    // ---- P0/P1: witness, blinding rows, instance column
    std::vector<Fr> instances;
    Fr* const grid = pr->h_grid;                        // pinned: [A][n] advice, then the instance column
    Fr* const instcol = grid + (size_t)A * n;           // rows past the instances stay zero (cleared at creation)
    uint32_t rows_now;                                  // leading rows of the advice columns that this proof may fill
    Transcript tr(pk->vk_digest);
    // halo2's transcripts refuse the identity ("cannot write points at infinity to the transcript": create_proof fails with
    // Error::Transcript) and the reference's verifier refuses (0, 0) as a proof point
    // (/root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:89-99): a circuit one of whose commitments is the
    // identity -- a quotient piece that is identically zero, say -- has no proof, and proving it is refused here as well
    auto put_point = [&](const G1Affine& pt) {
        SHP_REQUIRE(!(pt.x.is_zero() && pt.y.is_zero()), SHP_ERR_INVALID_ARG,
                    "prove: a commitment is the point at infinity, which the transcript refuses (as halo2's does): this circuit and witness have no proof");
        tr.write_point(pt.x, pt.y);
    };
    std::vector<G1Affine> pts(std::max<uint32_t>({A + 2 * L + 1, S + L + 1, P.degree}));
    std::vector<Fr> chal(P.challenge_phase.size(), Fr::zero());
    ProgArgs pa;
    pa.out_raw = nullptr;
    pa.consts = pk->consts;
    pa.adv_q = pk->adv_q; pa.fix_q = pk->fix_q; pa.inst_q = pk->inst_q;
    auto lagrange_domain = [&](ProgArgs& a) {
        a.adv_base = V(0); a.adv_stride = n;
        a.fix_base = pk->fixed_vals; a.fix_stride = n;
        a.inst_base = V(I_INST);
        a.size_mask = n - 1; a.rot_step = 1; a.hi_mask = 0;
    };
    auto extended_domain = [&](ProgArgs& a) {
        a.adv_base = CS(0); a.adv_stride = ne;
        a.fix_base = pk->fixed_cos; a.fix_stride = ne;
        a.inst_base = CS(I_INST);
        a.size_mask = n - 1; a.rot_step = 1; a.hi_mask = ~(n - 1);
    };
    std::vector<Fr> hb;                                 // host staging of blinding values
    // P3's device work up to the permuted columns V(I_PIN ...): compress the lookups' expressions with theta, sort, permute,
    // blind.  Returns the per-lookup error words (pinned; valid once the stream has passed the flags kernel).
    // after_reads: recorded on ls once the advice columns have been read (behind the compression)
    auto lookup_columns = [&](const Fr& theta, hipStream_t ls, hipEvent_t after_reads = nullptr) -> uint32_t* {
        for (uint32_t l = 0; l < L; l++) {
            const LookupPlan& lk = P.lookups[l];
            lagrange_domain(pa);
            pa.mult = theta; pa.acc_in = nullptr;
            pa.prog = pk->lk_in_prog[l]; pa.n_instr = (uint32_t)lk.in_prog.size(); pa.nslots = lk.in_slots;
            // the compressed expressions in Montgomery form and, for the sort, as plain integers in the same pass;
            // input and table expression in one launch
            pa.out = pr->cin + (size_t)l * n;
            pa.out_raw = pr->sort_raw + (size_t)(2 * l) * n;
            ProgArgs pt = pa;
            pt.prog = pk->lk_tab_prog[l]; pt.n_instr = (uint32_t)lk.tab_prog.size(); pt.nslots = lk.tab_slots;
            pt.out = pr->ctab + (size_t)l * n;
            pt.out_raw = pr->sort_raw + (size_t)(2 * l + 1) * n;
            launch_prog(ls, pa, n, pr->throughput, &pt);
            pa.out_raw = nullptr;
        }
        if (after_reads) SHP_HIP(hipEventRecord(after_reads, ls));
        pr->timers.begin(KID_RANK_SORT, ls);
        SHP_REQUIRE(2 * L <= pr->sc_cnt, SHP_ERR_INVALID_ARG, "prove: too many lookups for the sort scratch");
        hipLaunchKernelGGL(sort_tiles_kernel, dim3(ceil_div(u, SORT_TILE), 2 * L), dim3(512), 0, ls, pr->sort_raw, pr->sc_pre,
                           pr->rank, u, (size_t)n);
        uint32_t G = 1;
        while (G < ceil_div(u, SORT_TILE)) G <<= 1;                       // <= 64 for k <= 16
        hipLaunchKernelGGL(rank_from_tiles_kernel, dim3(ceil_div((size_t)u * G, 256), 2 * L), dim3(256), 0, ls, pr->sort_raw,
                           pr->sc_pre, pr->rank, pr->sort_out, u, (size_t)n, G);
        pr->timers.end(ls);
        hb.clear();
        for (uint32_t l = 0; l < L; l++) {
            for (uint32_t j = 0; j < 2 * nb; j++) hb.push_back(draw());   // permuted input rows, then table rows
            ctr += 2;                                                     // two (unused) commitment blinds
        }
        const Fr* lk_blinds = (const Fr*)staged(hb.data(), hb.size() * sizeof(Fr));
        uint32_t* herr = (uint32_t*)pr->stage_alloc(L * sizeof(uint32_t));   // written by the kernel, read after the commit
        for (uint32_t l = 0; l < L; l++) herr[l] = 0;
        uint32_t nt = std::min<uint32_t>(1024, n);
        // sorted input of lookup l at sort_out[2l], sorted table at sort_out[2l+1]
        LookupPermuteArgs lp;
        lp.pin_all = V(I_PIN); lp.ptab_all = V(I_PIN + 1); lp.pstride = (size_t)2 * n; lp.scratch_all = pr->rep_scratch;
        lp.blinds = lk_blinds; lp.err = herr; lp.done = pr->lk_done;
        lp.tmont_all = pr->sort_raw + n; lp.tmont_stride = (size_t)2 * n;     // the unsorted table keys are dead by now
        hipLaunchKernelGGL(lookup_flags_kernel, dim3(ceil_div(u, nt), L), dim3(nt), 0, ls, pr->sort_out, pr->sort_out + n,
                           (size_t)2 * n, u, n, pr->rank, lp);
        return herr;
    };
    // A lookup over single expressions (one input, one table expression: a range check) is not compressed with theta at
    // all, so its permuted columns depend on the advice columns alone: they are built before the advice commitment and
    // committed in the SAME launch -- one commitment launch and one host round trip less per proof.  The transcript still
    // takes the advice points, squeezes theta, then takes the permuted columns' points.
    bool early_lookups = L > 0 && !phased;
    for (const LookupPlan& lk : P.lookups) early_lookups = early_lookups && lk.m == 1;
    {
        static const bool off = getenv("SHP_EARLY_LOOKUPS") && atoi(getenv("SHP_EARLY_LOOKUPS")) == 0;
        if (off) early_lookups = false;
    }
    uint32_t* early_err = nullptr;
    bool lookups_ahead = false;
    // positions in the blinding stream: advice column c's row r; the lookups' values (behind the advice columns' rows and
    // unused blinds)
    auto blind_of = [&](uint32_t c, uint32_t r) {
        const uint64_t at = (uint64_t)c * nb + (r - u);
        return in.draws ? in.draws[at] : fr_draw(key, at);
    };
    const uint64_t lookup_ctr0 = (uint64_t)A * nb + A, lookup_ctr1 = lookup_ctr0 + (uint64_t)L * (2 * nb + 2);
    // every job of one launch, through the pool's combiner when this lane has one
    auto commit_jobs = [&](MsmJob* jobs, uint32_t count) {
        bool direct = true;
        for (uint32_t j = 0; j < count; j++) direct = direct && jobs[j].m->direct;
        if (!direct) {
            for (uint32_t j = 0; j < count; j++) msm_run_ws(jobs[j].m, pr->msm_ws, jobs[j].scalars, n, n, jobs[j].batch, jobs[j].h_out, st);
        } else if (pr->comb) {
            pr->comb->run(jobs, count, st, pr->comb_event);
        } else {
            msm_run_jobs(jobs, count, pr->msm_ws, st);
        }
    };
    const Fr* advice_once = in.advice;                  // one phase, no challenge: the callback is asked once
    if (!phased && in.phase_fn) {
        advice_once = (const Fr*)in.phase_fn(in.phase_user, 0, nullptr, 0);
        SHP_REQUIRE(advice_once, SHP_ERR_INVALID_ARG, "prove_assigned: the phase callback returned no columns");
    }
    if (!phased) {
        if (assigned) {
            // the caller's columns: rows [0, assigned_rows) through the pinned grid (asynchronous copies from pageable memory
            // serialise the streams of concurrent lanes), values validated on the way
            rows_now = in.assigned_rows ? in.assigned_rows : u;
            for (uint32_t c = 0; c < A; c++) {
                const Fr* src = advice_once + (size_t)c * (in.advice_stride ? in.advice_stride : n);
                Fr* dst = grid + (size_t)c * n;
                for (uint32_t r = 0; r < rows_now; r++) {
                    SHP_REQUIRE(!src[r].geq_p(), SHP_ERR_FORMAT, "prove_assigned: advice value is not a canonical field element");
                    dst[r] = src[r];
                }
            }
            pr->grid_foreign = true;
            instances.assign(in.instances, in.instances + in.n_instances);
            for (auto& v : instances) SHP_REQUIRE(!v.geq_p(), SHP_ERR_FORMAT, "prove_assigned: public input is not a canonical field element");
        } else {
            std::vector<Fr> inputs(n_inputs);
            for (size_t i = 0; i < n_inputs; i++) {
                Fr raw;
                memcpy(raw.l, inputs_le32 + 32 * i, 32);
                SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "prove: witness input is not a canonical field element");
                inputs[i] = Fr::from_raw(raw);
            }
            if (pr->grid_foreign) {                         // an assigned-columns proof left other cells behind
                memset(grid, 0, (size_t)A * n * sizeof(Fr));
                pr->grid_foreign = false;
            }
            rows_now = P.adv_rows;
            // the lookups' input columns ahead of the program, and their sort and permutation on the second stream while
            // the program runs (a lone proof only; a proof that has rows of its predecessor to clear keeps the usual order)
            const bool ahead = split_commits && early_lookups && !pr->early_ranges.empty() && rows_now >= pr->dev_rows;
            // (the helpers read `inputs` and write the grid until finish() returns: on every path out of this block)
            struct Joiner {
                WitnessWorkers* w = nullptr;
                void join() { if (w) { WitnessWorkers* x = w; w = nullptr; x->finish(); } }
                ~Joiner() { join(); }
            } started;
            if (ahead) {
                pr->second_stream();
                for (const auto& r : pr->early_ranges) run_witness_range(P, inputs, grid, pr->wscratch.data(), r.lo, r.hi);
                for (uint32_t c : pr->early_cols)
                    for (uint32_t r = u; r < n; r++) grid[(size_t)c * n + r] = blind_of(c, r);
                // the helpers start on the program (it writes the same values into the cells above again: nothing reads
                // them on the host meanwhile) while this thread queues the lookups' kernels
                if (pr->wworkers) { pr->wworkers->start(P, inputs, grid, pr->wsched); started.w = pr->wworkers.get(); }
                for (size_t k0 = 0; k0 < pr->early_cols.size();) {           // runs of consecutive columns
                    size_t k1 = k0 + 1;
                    while (k1 < pr->early_cols.size() && pr->early_cols[k1] == pr->early_cols[k1 - 1] + 1) k1++;
                    ingest(pr->early_cols[k0], (uint32_t)(k1 - k0), rows_now, 0, pr->st2);
                    k0 = k1;
                }
                ctr = lookup_ctr0;                            // the lookups' blinding values sit behind the advice columns'
                early_err = lookup_columns(Fr::zero(), pr->st2, pr->ev_in);
                SHP_REQUIRE(ctr == lookup_ctr1, SHP_ERR_INTERNAL, "prove: blinding stream position of the lookups");
                ctr = 0;
                SHP_HIP(hipEventRecord(pr->ev2, pr->st2));
                // their commitment follows them on that stream, in a launch of its own: it waits neither for the witness
                // nor for the advice commitment, and the transcript takes its points behind the advice columns' all the same
                {
                    MsmJob pj{srs->msm_gl, V(I_PIN), n, 2 * L, MsmRows{n, 0, 0}, pts.data() + A};
                    pr->msm_ws_perm.poll_results = true;
                    msm_launch_jobs(&pj, 1, pr->msm_ws_perm, pr->st2);
                }
                lookups_ahead = true;
            }
            launch_rand();
            // no per-proof clearing: the program writes the same cells every time (fixed plan), the blinding rows are
            // redrawn below, everything else has been zero since prover_create
            if (started.w) started.join();
            else if (pr->wworkers) pr->wworkers->run(P, inputs, grid, pr->wsched);
            else run_witness_range(P, inputs, grid, pr->wscratch.data(), 0, (uint32_t)P.wops.size());
            instances.resize(P.instance_map.size());
            for (size_t i = 0; i < instances.size(); i++)
                instances[i] = grid[(size_t)P.instance_map[i].first * n + P.instance_map[i].second];
        }
        pc.mark("run_witness");
        for (uint32_t c = 0; c < A; c++)
            for (uint32_t r = u; r < n; r++) grid[(size_t)c * n + r] = draw();
        pc.mark("blind_draws");
        ctr += A;                                           // one (unused) commitment blind per advice column
        if (lookups_ahead) ctr = lookup_ctr1;               // drawn when the lookups were built
        for (size_t i = 0; i < instances.size(); i++) instcol[i] = instances[i];
        // only the rows this proof assigns and the blinding rows travel; the rest is zero on both sides
        if (A && rows_now < pr->dev_rows)                   // the previous proof filled more rows than this one
            hipLaunchKernelGGL(zero_rows_kernel, dim3(ceil_div(pr->dev_rows - rows_now, 256), A), dim3(256), 0, st, V(0), (size_t)n,
                               rows_now, pr->dev_rows);
        pr->dev_rows = rows_now;
        for (auto& v : instances) tr.common_scalar(v);
        pc.mark("witness+upload");
        // ---- P2: advice commitments.  The random polynomial of P7 depends on no challenge: its
        // coefficients (blinding stream position known up front) and its commitment ride in the same launch.
        const MsmRows adv_rows{rows_now, u, n - u};       // assigned rows + blinding rows; the rest is zero
        MsmJob jobs[3];
        uint32_t n_jobs = 0;
        jobs[n_jobs++] = MsmJob{srs->msm_gl, V(0), n, A, adv_rows, pts.data()};
        if (split_commits) {
            // witness rows and instances in one dispatch; the random polynomial is on its way already
            if (lookups_ahead) {
                // the columns that went ahead are simply uploaded again with the others, one dispatch for all, once the second
                // stream has read them (long ago); their permuted columns have a commitment launch of their own there
                SHP_HIP(hipStreamWaitEvent(st, pr->ev_in, 0));
                ingest(0, A + 1, rows_now, (uint32_t)instances.size());
            } else {
                ingest(0, A + 1, rows_now, (uint32_t)instances.size());
                if (early_lookups) {
                    early_err = lookup_columns(Fr::zero(), st);
                    jobs[n_jobs++] = MsmJob{srs->msm_gl, V(I_PIN), n, 2 * L, MsmRows{n, 0, 0}, pts.data() + A};
                }
            }
            pr->msm_ws.poll_results = true;
            // every column in front of the grand products is final now: its coefficient form and cosets (P9) are computed on
            // the second stream, next to this commitment and the grand products, instead of behind them
            static const bool tr_ahead_on = !(getenv("SHP_TRANSFORMS_AHEAD") && atoi(getenv("SHP_TRANSFORMS_AHEAD")) == 0);
            const uint32_t final_cols = (early_lookups || L == 0) ? I_PZ : A;
            const bool tr_ahead = tr_ahead_on && final_cols > 0;
            if (tr_ahead) {
                pr->second_stream();
                SHP_HIP(hipEventRecord(pr->ev_cols, st));
            }
            msm_launch_jobs(jobs, n_jobs, pr->msm_ws, st);
            if (tr_ahead) {
                if (!pr->ntt_tmp2) pr->ntt_tmp2 = pr->dalloc<Fr>((size_t)(NV + 1) * ne);
                SHP_HIP(hipStreamWaitEvent(pr->st2, pr->ev_cols, 0));
                ntt_launch(ctx, pr->vals, pr->polys, k, final_cols, ntt_inv, pr->st2, pr->ntt_tmp2);
                ntt_launch(ctx, pr->polys, pr->cos, k, final_cols * pk->nc, ntt_ext, pr->st2, pr->ntt_tmp2);
                SHP_HIP(hipEventRecord(pr->ev_tr, pr->st2));
                inv_done = ext_done = final_cols;
                transforms_ahead = true;
            }
            msm_finish_jobs(pr->msm_ws_rand, st);
            msm_finish_jobs(pr->msm_ws, st);
        } else {
            // witness rows, instances and the random polynomial reach the device in one dispatch (ingest_kernel), and the random
            // polynomial's commitment rides in the advice columns' launch
            ingest(0, A + 2, rows_now, (uint32_t)instances.size());
            rand_values();
            if (early_lookups) {
                early_err = lookup_columns(Fr::zero(), st);
                jobs[n_jobs++] = MsmJob{srs->msm_gl, V(I_PIN), n, 2 * L, MsmRows{n, 0, 0}, pts.data() + A};
            }
            jobs[n_jobs++] = MsmJob{srs->msm_gl, rand_lag, n, 1, MsmRows{n, 0, 0}, &rand_comm};
            commit_jobs(jobs, n_jobs);
        }
        for (uint32_t c = 0; c < A; c++) put_point(pts[c]);
    } else {
        // Several phases (halo2 create_proof's loop over the phases; the reference's verifier reads the proof this way,
        // /root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:267-282): per phase the witness with the challenges
        // known so far, blinding rows and commitments of this phase's columns, then the challenges usable after it.  No
        // overlap of the commitments with later work here: a phase's witness needs the challenges the commitments produce.
        std::vector<Fr> inputs;
        if (!assigned) {
            inputs.resize(n_inputs);
            for (size_t i = 0; i < n_inputs; i++) {
                Fr raw;
                memcpy(raw.l, inputs_le32 + 32 * i, 32);
                SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "prove: witness input is not a canonical field element");
                inputs[i] = Fr::from_raw(raw);
            }
            if (pr->grid_foreign) {
                memset(grid, 0, (size_t)A * n * sizeof(Fr));
                pr->grid_foreign = false;
            }
            rows_now = P.adv_rows;
        } else {
            rows_now = in.assigned_rows ? in.assigned_rows : u;
            pr->grid_foreign = true;
        }
        const MsmRows adv_rows{rows_now, u, n - u};
        for (uint32_t ph = 0; ph < P.num_phases; ph++) {
            const auto runs = P.phase_runs(ph);
            if (assigned) {
                const Fr* src_all = in.advice;
                if (in.phase_fn) {
                    src_all = (const Fr*)in.phase_fn(in.phase_user, ph, (const uint64_t*)chal.data(), chal.size());
                    SHP_REQUIRE(src_all, SHP_ERR_INVALID_ARG, "prove_assigned: the phase callback returned no columns");
                }
                for (const auto& run : runs)
                    for (uint32_t c = run.first; c < run.first + run.second; c++) {
                        const Fr* src = src_all + (size_t)c * n;
                        Fr* dst = grid + (size_t)c * n;
                        for (uint32_t r = 0; r < rows_now; r++) {
                            SHP_REQUIRE(!src[r].geq_p(), SHP_ERR_FORMAT, "prove_assigned: advice value is not a canonical field element");
                            dst[r] = src[r];
                        }
                    }
                if (ph == 0) {
                    instances.assign(in.instances, in.instances + in.n_instances);
                    for (auto& v : instances)
                        SHP_REQUIRE(!v.geq_p(), SHP_ERR_FORMAT, "prove_assigned: public input is not a canonical field element");
                }
            } else {
                run_witness_range(P, inputs, grid, pr->wscratch.data(), P.wops_begin(ph), P.wops_end[ph], chal.data());
                if (ph == 0) {
                    instances.resize(P.instance_map.size());
                    for (size_t i = 0; i < instances.size(); i++)
                        instances[i] = grid[(size_t)P.instance_map[i].first * n + P.instance_map[i].second];
                }
            }
            // blinding rows of this phase's columns, then one (unused) commitment blind per column
            uint32_t cols = 0;
            for (const auto& run : runs)
                for (uint32_t c = run.first; c < run.first + run.second; c++, cols++)
                    for (uint32_t r = u; r < n; r++) grid[(size_t)c * n + r] = draw();
            ctr += cols;
            if (ph == 0) {
                for (size_t i = 0; i < instances.size(); i++) instcol[i] = instances[i];
                if (A && rows_now < pr->dev_rows)
                    hipLaunchKernelGGL(zero_rows_kernel, dim3(ceil_div(pr->dev_rows - rows_now, 256), A), dim3(256), 0, st, V(0),
                                       (size_t)n, rows_now, pr->dev_rows);
                pr->dev_rows = rows_now;
                for (auto& v : instances) tr.common_scalar(v);
                // the instance column (and, unless it is on its way already, the random polynomial)
                ingest(A, split_commits ? 1 : 2, rows_now, (uint32_t)instances.size());
            }
            for (const auto& run : runs) ingest(run.first, run.second, rows_now, 0);
            for (const auto& run : runs) {
                MsmJob job{srs->msm_gl, V(run.first), n, run.second, adv_rows, pts.data() + run.first};
                if (!srs->msm_gl->direct) msm_run_ws(srs->msm_gl, pr->msm_ws, V(run.first), n, n, run.second, pts.data() + run.first, st);
                else if (pr->comb) pr->comb->run(&job, 1, st, pr->comb_event);
                else msm_run_jobs(&job, 1, pr->msm_ws, st);
            }
            if (ph == 0 && !split_commits) {
                rand_values();
                commit(srs->msm_gl, rand_lag, 1, &rand_comm);
            }
            for (const auto& run : runs)
                for (uint32_t c = run.first; c < run.first + run.second; c++) put_point(pts[c]);
            for (size_t i = 0; i < chal.size(); i++)
                if (P.challenge_phase[i] == ph) chal[i] = tr.squeeze();
        }
        if (split_commits) msm_finish_jobs(pr->msm_ws_rand, st);
        pc.mark("advice_phases");
    }
    const Fr theta = tr.squeeze();

    if (phased) {                                       // every challenge is known: they complete the lane's constant table
        h2d(pr->consts_lane + P.consts.size(), chal.data(), chal.size() * sizeof(Fr));
        pa.consts = pr->consts_lane;
    }
    pc.mark("advice_msm");
    // ---- P3: lookups -- compress, permute, commit (unless their columns went out with the advice columns' launch)
    if (L && !early_lookups) {
        uint32_t* const herr = lookup_columns(theta, st);
        if (split_commits) {
            commit_launch(pr->msm_ws, srs->msm_gl, V(I_PIN), 2 * L, pts.data());
            msm_finish_jobs(pr->msm_ws, st);
        } else {
            commit(srs->msm_gl, V(I_PIN), 2 * L, pts.data());             // the MSM run synchronises the stream
        }
        for (uint32_t l = 0; l < L; l++)
            SHP_REQUIRE(herr[l] == 0, SHP_ERR_UNSATISFIED, "prove: lookup input not contained in the table");
        for (uint32_t i = 0; i < 2 * L; i++) put_point(pts[i]);
    } else if (L) {
        if (lookups_ahead) msm_finish_jobs(pr->msm_ws_perm, pr->st2);
        for (uint32_t l = 0; l < L; l++)
            SHP_REQUIRE(early_err[l] == 0, SHP_ERR_UNSATISFIED, "prove: lookup input not contained in the table");
        for (uint32_t i = 0; i < 2 * L; i++) put_point(pts[A + i]);
    }
    pc.mark("lookup");
    // ---- P4
    const Fr beta = tr.squeeze();
    const Fr gamma = tr.squeeze();
    // ---- P5/P6: grand products (permutation sets chained, lookups), P7: random polynomial
    PermArgs pg;
    pg.n_cols = PC; pg.chunk = P.chunk; pg.beta = beta; pg.gamma = gamma;
    pg.sigma = pk->sigma_vals; pg.sigma_stride = n; pg.xs = pk->omegas;
    for (uint32_t c = 0; c < PC; c++) {
        uint32_t t = P.perm_cols[c].first, ix = P.perm_cols[c].second;
        pg.col[c] = t == COL_ADVICE ? V(ix) : t == COL_FIXED ? pk->fixed_vals + (size_t)ix * n : V(I_INST);
        pg.dbeta[c] = pk->delta_pow[c] * beta;
    }
    if (S + L) {
        LookupRatioArgs lr;
        lr.cin = pr->cin; lr.ctab = pr->ctab; lr.pin = V(I_PIN); lr.ptab = V(I_PIN + 1); lr.pstride = (size_t)2 * n;
        lr.beta = beta; lr.gamma = gamma;
        hipLaunchKernelGGL(perm_ratio_kernel, dim3(ceil_div(n, 256), S + L), dim3(256), 0, st, pg, lr, S, n, pr->num, pr->den);
    }
    hb.clear();
    for (uint32_t p = 0; p < S + L; p++) {
        for (uint32_t j = 0; j < bf; j++) hb.push_back(draw());
        ctr += 1;
    }
    const Fr* z_blinds = (const Fr*)staged(hb.data(), hb.size() * sizeof(Fr));
    pr->timers.begin(KID_GRAND_PRODUCT, st);
    if (S + L) {                                           // a plan may have neither a permutation nor a lookup
        const InvList il = product_scans(pr->num, pr->den, S + L);
        hipLaunchKernelGGL(sce == 1 ? grand_product_apply_kernel<1> : grand_product_apply_kernel<SCAN_E_POOL>, dim3(snb, S + L),
                           dim3(sbs), 0, st, pr->num, pr->den, pr->sc_pre, pr->sc_suf, sc_bpre, sc_bsuf, sc_inv, n, S, u, bf,
                           V(I_PZ), (size_t)n, z_blinds, il);
    }
    pr->timers.end(st);
    SHP_REQUIRE(ctr == ctr_rand, SHP_ERR_INTERNAL, "prove: blinding stream position of the random polynomial");
    ctr += n;
    ctr += 1;
    // ---- P9 (coefficient forms and cosets of everything committed so far + the instance column) needs no challenge: for
    // a lone proof it is queued behind the grand products' commitment and runs while the host waits for those points
    // (with the leading columns' transforms on the second stream, the rest -- the grand products and the instance column
    // -- follow them there, next to the grand products' commitment instead of behind it)
    auto transforms = [&] {
        if (transforms_ahead) {
            const uint32_t c0 = ext_done, cnt = NV + 1 - ext_done;
            SHP_HIP(hipStreamWaitEvent(pr->st2, pr->ev_cols, 0));
            ntt_launch(ctx, pr->vals + (size_t)c0 * n, pr->polys + (size_t)c0 * n, k, cnt, ntt_inv, pr->st2, pr->ntt_tmp2);
            ntt_launch(ctx, pr->polys + (size_t)c0 * n, pr->cos + (size_t)c0 * ne, k, cnt * pk->nc, ntt_ext, pr->st2, pr->ntt_tmp2);
            SHP_HIP(hipEventRecord(pr->ev_tr, pr->st2));
            inv_done = ext_done = NV + 1;
            SHP_HIP(hipStreamWaitEvent(st, pr->ev_tr, 0));
            return;
        }
        ext_upto(NV + 1);
    };
    if (split_commits && S + L) {
        if (transforms_ahead) SHP_HIP(hipEventRecord(pr->ev_cols, st));   // the grand products are on the device
        commit_launch(pr->msm_ws, srs->msm_gl, V(I_PZ), S + L, pts.data());
        transforms();
        msm_finish_jobs(pr->msm_ws, st);
    } else {
        commit(srs->msm_gl, V(I_PZ), S + L, pts.data());
    }
    for (uint32_t i = 0; i < S + L; i++) put_point(pts[i]);
    put_point(rand_comm);
    pc.mark("products_msm");
    // ---- P8
    const Fr y = tr.squeeze();
    if (!(split_commits && S + L)) transforms();
    // ---- P10: quotient
    const uint32_t ebs = std::min<uint32_t>(256, n);      // element-wise kernels over the quotient domain (n >= 64)
    if (pk->quot_fn) {
        // one generated kernel for the whole numerator; the challenges go in in its radix (2^261)
        const Fr c32 = Fr::from_u64(32);
        Quot29Args qa;
        qa.cos = pr->cos; qa.ne = ne;
        qa.fixed_cos = pk->fixed_cos; qa.sigma_cos = pk->sigma_cos;
        qa.l0 = pk->l0; qa.l_last = pk->l_last; qa.l_active = pk->l_active; qa.dx_ext = pk->dx_ext;
        qa.consts = pk->consts29;
        qa.i_pin = I_PIN; qa.i_pz = I_PZ; qa.i_lz = I_LZ; qa.i_inst = I_INST;
        qa.mask = n - 1; qa.rot_last = P.rot_last;
        qa.y = y * c32; qa.theta = theta * c32; qa.beta = beta * c32; qa.gamma = gamma * c32;
        const uint32_t last_cols = PC - (S ? (S - 1) * P.chunk : 0);
        SHP_REQUIRE(!beta.is_zero(), SHP_ERR_INTERNAL, "prove: the challenge beta is zero");     // probability 2^-254
        qa.beta_inv = Fr::inv(beta) * c32;
        qa.beta_pow_full = Fr::pow_u64(beta, P.chunk) * c32;
        qa.beta_pow_last = Fr::pow_u64(beta, last_cols) * c32;
        qa.h = pr->h;
        pr->timers.begin(KID_PROG, st);
        // a lone proof's 896 waves would leave every SIMD with at most one wave, issuing every other slot: half-filled
        // waves (32 points each) put two on a SIMD and finish sooner; pool lanes keep full waves (half the instructions)
        const uint32_t qbs = pr->throughput ? QUOT29_THREADS : QUOT29_THREADS / 2;
        hipLaunchKernelGGL(pk->quot_fn, dim3(ne / qbs), dim3(qbs), 0, st, qa);
        pr->timers.end(st);
    } else {
        extended_domain(pa);
        pa.mult = y; pa.acc_in = nullptr;
        pa.prog = pk->gate_prog; pa.n_instr = (uint32_t)P.gate_prog.size(); pa.nslots = P.gate_slots;
        pa.out = pr->h;
        pr->timers.begin(KID_PROG, st);
        if (pk->gate_fn) hipLaunchKernelGGL(pk->gate_fn, dim3(ne / PROG_THREADS), dim3(PROG_THREADS), 0, st, pa);
        else launch_prog(st, pa, ne, pr->throughput);
        pr->timers.end(st);
        // the theta-compressed input and table of lookup l on the quotient domain, then its terms -- for the first lookup in
        // the same pass as the permutation terms
        auto lookup_args = [&](uint32_t l) {
            const LookupPlan& lk = P.lookups[l];
            extended_domain(pa);
            pa.mult = theta; pa.acc_in = nullptr;
            pa.prog = pk->lk_in_prog[l]; pa.n_instr = (uint32_t)lk.in_prog.size(); pa.nslots = lk.in_slots;
            pa.out = pr->aux_ext;
            ProgArgs pt = pa;
            pt.prog = pk->lk_tab_prog[l]; pt.n_instr = (uint32_t)lk.tab_prog.size(); pt.nslots = lk.tab_slots;
            pt.out = pr->aux_ext + ne;
            launch_prog(st, pa, ne, pr->throughput, &pt);
            LookupQArgs q;
            q.z = CS(I_LZ + l); q.pin = CS(I_PIN + 2 * l); q.ptab = CS(I_PIN + 2 * l + 1);
            q.ain = pr->aux_ext; q.stab = pr->aux_ext + ne;
            q.l0 = pk->l0; q.l_last = pk->l_last; q.l_active = pk->l_active;
            q.mask = n - 1; q.y = y; q.beta = beta; q.gamma = gamma; q.h = pr->h;
            return q;
        };
        uint32_t l_first = 0;
        if (S) {
            PermQArgs q;
            q.p = pg;
            q.p.sigma = pk->sigma_cos; q.p.sigma_stride = ne; q.p.xs = pk->x_ext;
            for (uint32_t c = 0; c < PC; c++) {
                uint32_t t = P.perm_cols[c].first, ix = P.perm_cols[c].second;
                q.p.col[c] = t == COL_ADVICE ? CS(ix) : t == COL_FIXED ? pk->fixed_cos + (size_t)ix * ne : CS(I_INST);
            }
            q.pz = CS(I_PZ);
            q.l0 = pk->l0; q.l_last = pk->l_last; q.l_active = pk->l_active;
            q.n_sets = S; q.mask = n - 1; q.stride = ne; q.rot_last = P.rot_last; q.y = y; q.h = pr->h;
            LookupQArgs lq = {};
            if (L) { lq = lookup_args(0); l_first = 1; }
            hipLaunchKernelGGL(perm_quotient_kernel, dim3(ne / ebs), dim3(ebs), 0, st, q, lq, L ? 1u : 0u);
        }
        for (uint32_t l = l_first; l < L; l++) {
            const LookupQArgs q = lookup_args(l);
            hipLaunchKernelGGL(lookup_quotient_kernel, dim3(ne / ebs), dim3(ebs), 0, st, q);
        }
    }
    // per-coset interpolants, then the nc x nc solve per coefficient: the quotient pieces [nc][n] (in the buffer
    // the lookup compression no longer needs)
    // (the inverse transform's final store multiplies by c_j^-t / n instead of 1 / n: the combination is left with one
    // product per term)
    NttOpts ntt_inv_h = ntt_inv;
    ntt_inv_h.out_table = pk->cinv_n;
    ntt_launch(ctx, pr->h, pr->h, k, pk->nc, ntt_inv_h, st, pr->ntt_tmp);
    Fr* const hp = pr->aux_ext;
    hipLaunchKernelGGL(coset_combine_kernel, dim3(ceil_div(n, 256), pk->nc), dim3(256), 0, st, pr->h, pk->vinv, pk->nc, n, hp);
    const uint32_t npieces = P.degree - 1;
    ctr += npieces;                                      // one (unused) blind per quotient piece
    SHP_REQUIRE(ctr == randomness_len(P), SHP_ERR_INTERNAL, "prove: blinding stream length");
    Fr* const hp_lag = pr->h;                            // the pieces' values on the domain (the interpolants are consumed)
    ntt_launch(ctx, hp, hp_lag, k, npieces, ntt_fwd, st, pr->ntt_tmp);
    commit(srs->msm_gl, hp_lag, npieces, pts.data());
    for (uint32_t i = 0; i < npieces; i++) put_point(pts[i]);
    pc.mark("quotient_msm");
    // ---- P11: evaluations
    const Fr x = tr.squeeze();
    const std::vector<int32_t>& superset = pk->rot_superset;      // fixed per key (keygen)
    const std::vector<RotSet>& sets = pk->rot_sets;
    std::map<int32_t, uint32_t> pt_index;
    std::vector<Fr> points(superset.size());
    for (size_t i = 0; i < superset.size(); i++) {
        int32_t r = superset[i];
        pt_index[r] = (uint32_t)i;
        points[i] = x * (r >= 0 ? Fr::pow_u64(pk->omega, (uint64_t)r) : Fr::pow_u64(pk->omega_inv, (uint64_t)(-r)));
    }
    const Fr* d_points = (const Fr*)staged(points.data(), points.size() * sizeof(Fr));
    {
        // a lone proof: one power per thread (the chain is the exponentiation alone, ~20 dependent products); a pool
        // lane: 16 consecutive powers per thread (exponentiation + 16 products: 2.3 products per power instead of 20)
        const uint32_t per = pr->throughput ? 16 : 1;
        const Fr xn = Fr::pow_u64(x, n);                    // the extra grid row combines the quotient pieces into hx
        // a lone proof: the host squares every point out (k squarings each) so that a thread's power is a product over
        // the set bits of its index
        PowTable& sq = pr->pow_table;                       // 3.5 KB passed by value; its unused entries are never read
        uint32_t bits = 0;
        if (!pr->throughput && k <= POW_TABLE_BITS && points.size() <= POW_TABLE_POINTS) {
            bits = k;
            for (size_t p = 0; p < points.size(); p++) {
                Fr v = points[p];
                for (uint32_t b = 0; b < k; b++) { sq.v[p][b] = v; v = Fr::sqr(v); }
            }
        }
        hipLaunchKernelGGL(powers_kernel, dim3(ceil_div(ceil_div(n, per), 256), (uint32_t)points.size() + 2), dim3(256), 0, st,
                           d_points, (uint32_t)points.size(), n, per, pr->pw, (const Fr*)hp, npieces, xn, pr->hx,
                           (const Fr*)hp_lag, pr->lag2, sq, bits);
    }
    // the eval list in proof order, then h(x) (needed only for the opening argument)
    std::vector<EvalJob> jobs;
    std::vector<std::pair<PolyKey, int32_t>> job_key;
    auto coeff_of = [&](PolyKey kq) -> const Fr* {
        switch (kq.kind) {
            case 0: return PL(kq.idx);
            case 1: return PL(I_PZ + kq.idx);
            case 2: return PL(I_LZ + kq.idx);
            case 3: return PL(I_PIN + 2 * kq.idx);
            case 4: return PL(I_PIN + 2 * kq.idx + 1);
            case 5: return pk->fixed_polys + (size_t)kq.idx * n;
            case 6: return pk->sigma_polys + (size_t)kq.idx * n;
            case 7: return pr->hx;
            default: return pr->rand_poly;
        }
    };
    auto add_job = [&](PolyKey kq, int32_t rot) {
        jobs.push_back({coeff_of(kq), pt_index.at(rot), 0});
        job_key.push_back({kq, rot});
    };
    for (auto& q : P.adv_q) add_job({0, q.col}, q.rot);
    for (auto& q : P.fix_q) add_job({5, q.col}, q.rot);
    add_job({8, 0}, 0);
    for (uint32_t i = 0; i < PC; i++) add_job({6, i}, 0);
    for (uint32_t s = 0; s < S; s++) {
        add_job({1, s}, 0);
        add_job({1, s}, 1);
        if (s + 1 != S) add_job({1, s}, P.rot_last);
    }
    for (uint32_t l = 0; l < L; l++) {
        add_job({2, l}, 0); add_job({2, l}, 1); add_job({3, l}, 0); add_job({3, l}, -1); add_job({4, l}, 0);
    }
    const size_t n_proof_evals = jobs.size();
    add_job({7, 0}, 0);
    SHP_REQUIRE(jobs.size() <= 512, SHP_ERR_INVALID_ARG, "prove: too many evaluations");
    if (!pr->jobs_uploaded) {                               // coefficient buffers and point indices never change
        h2d(pr->jobs, jobs.data(), jobs.size() * sizeof(EvalJob));
        pr->jobs_uploaded = true;
    }
    // the evaluations land in the pinned arena directly (one store per job over PCIe, no copy dispatch)
    pc.mark("evals_host");
    Fr* const evals = (Fr*)pr->stage_alloc(jobs.size() * sizeof(Fr));
    const bool poll_evals = pr->msm_ws.poll_results && !pr->timers.enabled;
    uint32_t* const eval_stamp = poll_evals ? (uint32_t*)pr->stage_alloc(jobs.size() * sizeof(uint32_t)) : nullptr;
    const uint32_t eval_seq = poll_evals ? (++pr->stamp_seq ? pr->stamp_seq : ++pr->stamp_seq) : 0;
    for (size_t i = 0; poll_evals && i < jobs.size(); i++) eval_stamp[i] = eval_seq - 1;
    hipLaunchKernelGGL(eval_kernel, dim3((uint32_t)jobs.size()), dim3(EVAL_THREADS), 0, st, pr->jobs, pr->pw, n, evals, eval_stamp,
                       eval_seq);
    // (the values of hx and of the random polynomial on the domain, which the evaluation-form opening reads, exist already:
    // lag2 from the powers launch above, rand_lag from the random polynomial's commitment)
    // 1 / (omega^r - point) for every opening point: it depends on x alone, so it is queued here, behind the evaluations,
    // and runs while the host hashes them and prepares the opening argument.  (On the second stream, next to the
    // evaluations, it gained nothing: the extra event and launch calls cost the host what the overlap saved.)
    invert_point_diffs(d_points, points.data(), pr->dinv, (uint32_t)points.size());
    if (poll_evals) wait_stamps(eval_stamp, (uint32_t)jobs.size(), eval_seq, st);
    else sync_stream(st);
    std::map<std::pair<PolyKey, int32_t>, Fr> ev;
    for (size_t i = 0; i < jobs.size(); i++) ev[job_key[i]] = evals[i];
    for (size_t i = 0; i < n_proof_evals; i++) tr.write_scalar(evals[i]);
    pc.mark("evals");
    // ---- P12: SHPLONK, W
    const Fr zeta_c = tr.squeeze();
    const Fr nu = tr.squeeze();
    auto values_of = [&](PolyKey kq) -> const Fr* {
        switch (kq.kind) {
            case 0: return V(kq.idx);
            case 1: return V(I_PZ + kq.idx);
            case 2: return V(I_LZ + kq.idx);
            case 3: return V(I_PIN + 2 * kq.idx);
            case 4: return V(I_PIN + 2 * kq.idx + 1);
            case 5: return pk->fixed_vals + (size_t)kq.idx * n;
            case 6: return pk->sigma_vals + (size_t)kq.idx * n;
            case 7: return pr->lag2;
            default: return pr->lag2 + n;
        }
    };
    std::vector<LincombArgs> lin(sets.size());
    ZetaPowers zpow;
    {
        size_t longest = 0;
        for (const RotSet& rs : sets) longest = std::max(longest, rs.polys.size());
        Fr z = Fr::one();
        for (size_t j = 0; j < MAX_SET_POLYS; j++) {
            zpow.v[j] = j < longest ? z : Fr::zero();
            if (j < longest) z = z * zeta_c;
        }
    }
    ShplonkArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n_sets = (uint32_t)sets.size();
    std::vector<std::vector<Fr>> rcoefs(sets.size());
    Fr nu_pow = Fr::one();
    for (size_t i = 0; i < sets.size(); i++) {
        const RotSet& rs = sets[i];
        SHP_REQUIRE(rs.polys.size() <= MAX_SET_POLYS && rs.rots.size() <= MAX_SET_ROTS, SHP_ERR_INVALID_ARG,
                    "prove: rotation set too large");
        lin[i].count = (uint32_t)rs.polys.size();
        std::vector<Fr> comb(rs.rots.size(), Fr::zero()), xs(rs.rots.size());
        Fr zp = Fr::one();
        for (size_t j = 0; j < rs.polys.size(); j++) {
            lin[i].poly[j] = values_of(rs.polys[j]);
            for (size_t t = 0; t < rs.rots.size(); t++) comb[t] = comb[t] + zp * ev.at({rs.polys[j], rs.rots[t]});
            zp = zp * zeta_c;
        }
        for (size_t t = 0; t < rs.rots.size(); t++) xs[t] = points[pt_index[rs.rots[t]]];
        rcoefs[i] = interpolate(xs, comb);
        sa.n_rots[i] = (uint32_t)rs.rots.size();
        for (size_t t = 0; t < rs.rots.size(); t++) {
            sa.rot_pt[i][t] = pt_index[rs.rots[t]];
            sa.rcoef[i][t] = rcoefs[i][t];
        }
        sa.coef[i] = nu_pow;
        nu_pow = nu_pow * nu;
    }
    sa.f = pr->fsets; sa.dinv = pr->dinv; sa.omegas = pk->omegas;
    pc.mark("W_host");
    if (!pr->lin_uploaded) {                                // buffer addresses: the same for every proof of this lane
        h2d(pr->lin, lin.data(), lin.size() * sizeof(LincombArgs));
        pr->lin_uploaded = true;
    }
    hipLaunchKernelGGL(lincomb_kernel, dim3(ceil_div(n, 64), (uint32_t)sets.size()), dim3(pr->throughput ? 64 : 256), 0, st, pr->lin,
                       zpow, n, pr->fsets);
    hipLaunchKernelGGL(shplonk_w_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, sa, n, pr->wvals);
    commit(srs->msm_gl, pr->wvals, 1, pts.data());
    put_point(pts[0]);
    pc.mark("W");
    // ---- P13: W'
    const Fr mu = tr.squeeze();
    Fr zt = Fr::one();
    for (auto& p : points) zt = zt * (mu - p);
    std::vector<Fr> diffs(sets.size());
    for (size_t i = 0; i < sets.size(); i++) {
        Fr d = Fr::one();
        for (size_t j = 0; j < superset.size(); j++)
            if (std::find(sets[i].rots.begin(), sets[i].rots.end(), superset[j]) == sets[i].rots.end()) d = d * (mu - points[j]);
        diffs[i] = d;
    }
    const Fr d0inv = Fr::inv(diffs[0]);
    nu_pow = Fr::one();
    ShplonkWpArgs wp;
    wp.c0 = Fr::zero();
    for (size_t i = 0; i < sets.size(); i++) {
        Fr rmu = Fr::zero();
        for (size_t t = rcoefs[i].size(); t-- > 0;) rmu = rmu * mu + rcoefs[i][t];
        wp.coef[i] = nu_pow * diffs[i] * d0inv;
        wp.c0 = wp.c0 + wp.coef[i] * rmu;
        nu_pow = nu_pow * nu;
    }
    for (size_t i = sets.size(); i < (size_t)MAX_SETS; i++) wp.coef[i] = Fr::zero();
    wp.c1 = zt * d0inv;
    wp.n_sets = sa.n_sets; wp.f = pr->fsets; wp.w = pr->wvals;
    const Fr* d_mu = (const Fr*)staged(&mu, sizeof(Fr));
    {
        // 1 / (omega^r - mu) is formed inside the kernel that divides by it, from the product scans
        pr->timers.begin(KID_BATCH_INVERT, st);
        const InvList il = product_scans(nullptr, nullptr, 1, d_mu, &mu);
        hipLaunchKernelGGL(sce == 1 ? shplonk_wp_inv_kernel<1> : shplonk_wp_inv_kernel<SCAN_E_POOL>, dim3(snb), dim3(sbs), 0, st, wp,
                           pk->omegas, d_mu, pr->sc_pre, pr->sc_suf, sc_bpre, sc_bsuf, sc_inv, il, n, pr->wvals + n);
        pr->timers.end(st);
    }
    commit(srs->msm_gl, pr->wvals + n, 1, pts.data());
    put_point(pts[0]);
    pc.mark("Wprime");
    SHP_HIP(hipGetLastError());
    pr->proofs_done++;
    proof_out.swap(tr.proof);
    instances_out.swap(instances);
    (void)F;
}

int shp_pk_device(const shp_pk* pk) { return pk->ctx->device; }

void shp_pk_shape(const shp_pk* pk, uint32_t* num_advice, uint32_t* n, uint32_t* usable, bool* phased, uint64_t* n_draws) {
    const PlanData& P = pk->plan;
    *num_advice = P.num_advice; *n = P.n; *usable = P.usable; *phased = P.phased(); *n_draws = randomness_len(P);
}

void shp_pk_vk_view(const shp_pk* pk, VkView& out) {
    out.plan = &pk->plan;
    out.vk_digest = pk->vk_digest;
    out.fixed_comm = &pk->fixed_comm;
    out.sigma_comm = &pk->sigma_comm;
    out.omega = pk->omega;
    out.delta = pk->delta;
    out.superset = &pk->rot_superset;
    out.sets.clear();
    for (const RotSet& rs : pk->rot_sets) {
        std::vector<std::pair<int, uint32_t>> keys;
        for (const PolyKey& k : rs.polys) keys.push_back({k.kind, k.idx});
        out.sets.push_back({rs.rots, keys});
    }
    out.g2 = pk->srs->g2;
    out.s_g2 = pk->srs->s_g2;
    out.has_g2 = pk->srs->has_g2;
}

// ------------------------------------------------------------------------------ C ABI
extern "C" {

int32_t shp_keygen(shp_ctx* ctx, shp_srs* srs, const uint8_t* plan, size_t plan_len, shp_pk** out) {
    return guarded([&] { keygen_impl(ctx, srs, plan, plan_len, out); });
}

int32_t shp_keygen_with_transcript_repr(shp_ctx* ctx, shp_srs* srs, const uint8_t* plan, size_t plan_len,
                                        const uint8_t transcript_repr_le32[32], shp_pk** out) {
    return guarded([&] { keygen_impl(ctx, srs, plan, plan_len, out, nullptr, transcript_repr_le32); });
}

int32_t shp_plan_min_k(const uint8_t* plan, size_t plan_len, uint32_t* k_min) {
    return guarded([&] {
        SHP_REQUIRE(plan && k_min, SHP_ERR_INVALID_ARG, "plan_min_k: null argument");
        PlanData P;
        P.parse(plan, plan_len);
        *k_min = P.min_k();
    });
}

int32_t shp_plan_retarget(const uint8_t* plan, size_t plan_len, uint32_t k, uint8_t* out, size_t out_cap, size_t* out_len) {
    return guarded([&] {
        SHP_REQUIRE(plan && out_len, SHP_ERR_INVALID_ARG, "plan_retarget: null argument");
        PlanData P;
        P.parse(plan, plan_len);
        const std::vector<uint8_t> bytes = P.retarget(k).serialize();
        *out_len = bytes.size();
        if (!out) return;
        SHP_REQUIRE(out_cap >= bytes.size(), SHP_ERR_INVALID_ARG, "plan_retarget: buffer too small");
        memcpy(out, bytes.data(), bytes.size());
    });
}

int32_t shp_generate_keys_with_min_k(shp_ctx* ctx, const shp_srs* full_srs, const uint8_t* plan, size_t plan_len,
                                     const uint8_t* transcript_repr_le32, shp_srs** out_srs, uint32_t* out_k, shp_pk** out_pk) {
    return guarded([&] {
        SHP_REQUIRE(ctx && full_srs && plan && out_srs && out_pk, SHP_ERR_INVALID_ARG, "generate_keys_with_min_k: null argument");
        PlanData P;
        P.parse(plan, plan_len);
        const uint32_t kk = P.min_k();
        SHP_REQUIRE(kk <= full_srs->k, SHP_ERR_INVALID_ARG, "generate_keys_with_min_k: the circuit needs more rows than the SRS has powers");
        std::vector<uint8_t> bytes(plan, plan + plan_len);
        if (kk != P.k) bytes = P.retarget(kk).serialize();
        shp_srs* small = nullptr;
        int32_t rc = shp_srs_downsize(full_srs, kk, &small);
        SHP_REQUIRE(rc == SHP_OK, rc, std::string("generate_keys_with_min_k: ") + shp_last_error());
        try {
            keygen_impl(ctx, small, bytes.data(), bytes.size(), out_pk, nullptr, transcript_repr_le32);
        } catch (...) {
            (void)shp_srs_destroy(small);
            throw;
        }
        *out_srs = small;
        if (out_k) *out_k = kk;
    });
}

int32_t shp_pk_plan(const shp_pk* pk, uint8_t* out, size_t out_cap, size_t* out_len) {
    return guarded([&] {
        SHP_REQUIRE(pk && out_len, SHP_ERR_INVALID_ARG, "pk_plan: null argument");
        *out_len = pk->plan_bytes.size();
        if (!out) return;
        SHP_REQUIRE(out_cap >= pk->plan_bytes.size(), SHP_ERR_INVALID_ARG, "pk_plan: buffer too small");
        memcpy(out, pk->plan_bytes.data(), pk->plan_bytes.size());
    });
}

int32_t shp_pk_randomness_len(const shp_pk* pk, size_t* n_draws) {
    return guarded([&] {
        SHP_REQUIRE(pk && n_draws, SHP_ERR_INVALID_ARG, "pk_randomness_len: null argument");
        *n_draws = (size_t)randomness_len(pk->plan);
    });
}

int32_t shp_blinding_stream(const uint8_t seed[32], uint64_t first, size_t count, uint64_t* out_fr) {
    return guarded([&] {
        SHP_REQUIRE(seed && (out_fr || !count), SHP_ERR_INVALID_ARG, "blinding_stream: null argument");
        uint32_t key[8];
        memcpy(key, seed, 32);
        for (size_t i = 0; i < count; i++) {
            const Fr v = fr_draw(key, first + i);
            memcpy(out_fr + 4 * i, v.l, 32);
        }
    });
}

// ---- marshall_pk / unmarshall_pk (/root/reference/crates/shielder_bindings/build.rs:24-32,
// src/circuits/mod.rs:38-46).  Everything in the key except the commitments is a cheap, deterministic
// GPU function of (plan, SRS) -- keygen without the MSMs is a few NTTs -- so the saved key is the
// plan plus the verifying-key material, not tens of MB of polynomials; loading rebuilds the device
// resident tables in milliseconds (the TEE decodes its pk.bin on every request, server.rs:170-173).
//   "SHPK0002" | u32 k | u32 flags | 32 B SRS fingerprint | u32 plan_len | plan | u32 F | u32 PC |
//   F+PC x 64 B commitments | 32 B vk digest
// flags bit 0: the digest is a caller-supplied transcript_repr (shp_keygen_with_transcript_repr), not the library's own.
// The SRS fingerprint (Keccak over k, g[0], g[1], g_lagrange[0], g2, s_g2) binds the key to the SRS its commitments were
// computed over: loading it against another SRS of the same k fails instead of yielding proofs that never verify.
static const char PK_MAGIC[8] = {'S', 'H', 'P', 'K', '0', '0', '0', '3'};

int32_t shp_pk_save(const shp_pk* pk, uint8_t* out, size_t out_cap, size_t* out_len) {
    return guarded([&] {
        SHP_REQUIRE(pk && out_len, SHP_ERR_INVALID_ARG, "pk_save: null argument");
        const uint32_t F = (uint32_t)pk->fixed_comm.size(), PC = (uint32_t)pk->sigma_comm.size();
        const size_t total = 8 + 4 + 4 + 32 + 4 + pk->plan_bytes.size() + 4 + 4 + (size_t)(F + PC) * sizeof(G1Affine) + 64;
        *out_len = total;
        if (!out) return;
        SHP_REQUIRE(out_cap >= total, SHP_ERR_INVALID_ARG, "pk_save: buffer too small");
        uint8_t* p = out;
        auto put = [&](const void* src, size_t n) { memcpy(p, src, n); p += n; };
        const uint32_t k = pk->plan.k, pl = (uint32_t)pk->plan_bytes.size(), flags = pk->digest_injected ? 1u : 0u;
        put(PK_MAGIC, 8); put(&k, 4); put(&flags, 4); put(pk->srs->fingerprint, 32);
        put(&pl, 4); put(pk->plan_bytes.data(), pl); put(&F, 4); put(&PC, 4);
        put(pk->fixed_comm.data(), (size_t)F * sizeof(G1Affine));
        put(pk->sigma_comm.data(), (size_t)PC * sizeof(G1Affine));
        put(pk->vk_digest, 32);
        put(pk->own_digest, 32);        // the integrity check of a key whose vk_digest is the caller's transcript_repr
    });
}

int32_t shp_pk_load(shp_ctx* ctx, shp_srs* srs, const uint8_t* bytes, size_t len, shp_pk** out) {
    return guarded([&] {
        SHP_REQUIRE(ctx && srs && bytes && out, SHP_ERR_INVALID_ARG, "pk_load: null argument");
        const uint8_t* p = bytes;
        const uint8_t* end = bytes + len;
        auto need = [&](size_t n) { SHP_REQUIRE((size_t)(end - p) >= n, SHP_ERR_FORMAT, "pk_load: truncated key"); };
        need(52);
        SHP_REQUIRE(memcmp(p, PK_MAGIC, 8) == 0, SHP_ERR_FORMAT, "pk_load: bad magic / version");
        uint32_t k, flags, pl, F, PC;
        memcpy(&k, p + 8, 4); memcpy(&flags, p + 12, 4);
        SHP_REQUIRE(flags <= 1, SHP_ERR_FORMAT, "pk_load: unknown flags");
        SHP_REQUIRE(memcmp(p + 16, srs->fingerprint, 32) == 0, SHP_ERR_FORMAT,
                    "pk_load: the key was generated over a different SRS");
        memcpy(&pl, p + 48, 4);
        p += 52;
        need(pl);
        const uint8_t* plan = p;
        p += pl;
        need(8);
        memcpy(&F, p, 4); memcpy(&PC, p + 4, 4);
        p += 8;
        SHP_REQUIRE(F <= 4096 && PC <= MAX_PERM_COLS, SHP_ERR_FORMAT, "pk_load: implausible column counts");
        need((size_t)(F + PC) * sizeof(G1Affine) + 64);
        std::vector<G1Affine> comm(F + PC);
        memcpy(comm.data(), p, comm.size() * sizeof(G1Affine));
        p += comm.size() * sizeof(G1Affine);
        SHP_REQUIRE(p + 64 == end, SHP_ERR_FORMAT, "pk_load: trailing bytes");
        uint8_t repr_le[32];
        for (int i = 0; i < 32; i++) repr_le[i] = p[31 - i];       // the digest is stored big-endian, as absorbed
        shp_pk* pk = nullptr;
        keygen_impl(ctx, srs, plan, pl, &pk, &comm, (flags & 1) ? repr_le : nullptr);
        std::unique_ptr<shp_pk> guard(pk);
        SHP_REQUIRE(pk->plan.k == k, SHP_ERR_FORMAT, "pk_load: k differs from the plan");
        // The library's own digest is recomputed from the stored plan and commitments: accidental corruption of either
        // does not load.  (It does not prove that the commitments ARE the plan's: that would take the MSMs the saved
        // form exists to avoid; a key file is trusted input, like the reference's pk.bin.)
        // A key made with a caller-supplied transcript_repr takes its vk_digest from the file, so that comparison proves
        // nothing for it: the library's own digest is stored beside it and checked for every key.
        SHP_REQUIRE(memcmp(pk->vk_digest, p, 32) == 0 && memcmp(pk->own_digest, p + 32, 32) == 0, SHP_ERR_FORMAT,
                    "pk_load: verifying-key digest mismatch");
        *out = guard.release();
    });
}

int32_t shp_pk_destroy(shp_pk* pk) {
    return guarded([&] { delete pk; });
}

int32_t shp_pk_info(const shp_pk* pk, uint32_t* k, uint32_t* n_witness_inputs, uint32_t* n_instances, uint32_t* proof_len) {
    return guarded([&] {
        SHP_REQUIRE(pk, SHP_ERR_INVALID_ARG, "pk_info: null pk");
        const PlanData& P = pk->plan;
        uint32_t L = (uint32_t)P.lookups.size(), S = P.num_sets;
        uint32_t n_points = P.num_advice + 2 * L + S + L + 1 + (P.degree - 1);
        uint32_t n_evals = (uint32_t)(P.adv_q.size() + P.fix_q.size() + 1 + P.perm_cols.size() + (S ? 3 * S - 1 : 0) + 5 * L);
        if (k) *k = P.k;
        if (n_witness_inputs) *n_witness_inputs = P.n_winputs;
        if (n_instances) *n_instances = P.num_instance;
        if (proof_len) *proof_len = 64 * n_points + 32 * n_evals + 128;
    });
}

int32_t shp_pk_phases(const shp_pk* pk, uint32_t* num_phases, uint32_t* num_challenges, uint32_t* num_advice,
                      uint32_t* advice_phase, size_t advice_cap) {
    return guarded([&] {
        SHP_REQUIRE(pk, SHP_ERR_INVALID_ARG, "pk_phases: null pk");
        const PlanData& P = pk->plan;
        if (num_advice) *num_advice = P.num_advice;
        if (num_phases) *num_phases = P.num_phases;
        if (num_challenges) *num_challenges = (uint32_t)P.challenge_phase.size();
        if (advice_phase) {
            SHP_REQUIRE(advice_cap >= P.num_advice, SHP_ERR_INVALID_ARG, "pk_phases: one entry per advice column is written");
            for (uint32_t c = 0; c < P.num_advice; c++) advice_phase[c] = P.advice_phase[c];
        }
    });
}

int32_t shp_pk_gate_compiled(const shp_pk* pk, int32_t* compiled) {
    return guarded([&] {
        SHP_REQUIRE(pk && compiled, SHP_ERR_INVALID_ARG, "pk_gate_compiled: null argument");
        *compiled = pk->quot_fn ? 2 : pk->gate_fn ? 1 : 0;
    });
}

int32_t shp_pk_witness_native(const shp_pk* pk, uint32_t* ranges, uint32_t* ops, uint32_t* total_ops) {
    return guarded([&] {
        SHP_REQUIRE(pk && ranges && ops && total_ops, SHP_ERR_INVALID_ARG, "pk_witness_native: null argument");
        *ranges = (uint32_t)pk->plan.wperms.size();
        *ops = 0;
        for (const WPermutation& w : pk->plan.wperms) *ops += w.hi - w.lo;
        *total_ops = (uint32_t)pk->plan.wops.size();
    });
}

int32_t shp_pk_vk_digest(const shp_pk* pk, uint8_t out[32]) {
    return guarded([&] {
        SHP_REQUIRE(pk && out, SHP_ERR_INVALID_ARG, "pk_vk_digest: null argument");
        memcpy(out, pk->vk_digest, 32);
    });
}

int32_t shp_pk_commitments(const shp_pk* pk, void* h_fixed, size_t n_fixed, void* h_sigma, size_t n_sigma) {
    return guarded([&] {
        SHP_REQUIRE(pk, SHP_ERR_INVALID_ARG, "pk_commitments: null pk");
        SHP_REQUIRE(n_fixed == pk->fixed_comm.size() && n_sigma == pk->sigma_comm.size(), SHP_ERR_INVALID_ARG,
                    "pk_commitments: wrong counts");
        if (h_fixed) memcpy(h_fixed, pk->fixed_comm.data(), n_fixed * sizeof(G1Affine));
        if (h_sigma) memcpy(h_sigma, pk->sigma_comm.data(), n_sigma * sizeof(G1Affine));
    });
}

int32_t shp_prover_create(shp_pk* pk, shp_prover** out) {
    return guarded([&] { prover_create_impl(pk, out); });
}

int32_t shp_public_inputs(const shp_pk* pk, const uint8_t* witness_le32, size_t n_inputs, uint8_t* instances_le32,
                          size_t instances_cap) {
    return guarded([&] {
        SHP_REQUIRE(pk && witness_le32 && instances_le32, SHP_ERR_INVALID_ARG, "public_inputs: null argument");
        const PlanData& P = pk->plan;
        SHP_REQUIRE(n_inputs == P.n_winputs, SHP_ERR_INVALID_ARG, "public_inputs: wrong number of witness inputs");
        SHP_REQUIRE(instances_cap >= (size_t)P.num_instance * 32, SHP_ERR_INVALID_ARG, "public_inputs: instance buffer too small");
        std::vector<Fr> inputs(n_inputs);
        for (size_t i = 0; i < n_inputs; i++) {
            Fr raw;
            memcpy(raw.l, witness_le32 + 32 * i, 32);
            SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "public_inputs: witness input is not a canonical field element");
            inputs[i] = Fr::from_raw(raw);
        }
        // the witness program on the host, no GPU work: only the cells the public inputs are read from matter
        std::vector<Fr> grid((size_t)P.num_advice * P.n, Fr::zero()), scratch(std::max<uint32_t>(P.n_wscratch, 1), Fr::zero());
        run_witness_range(P, inputs, grid.data(), scratch.data(), 0, P.wops_end[0]);   // public inputs are first-phase cells
        for (size_t i = 0; i < P.instance_map.size(); i++) {
            Fr raw = grid[(size_t)P.instance_map[i].first * P.n + P.instance_map[i].second].to_raw();
            memcpy(instances_le32 + 32 * i, raw.l, 32);
        }
    });
}

int32_t shp_witness_columns(const shp_pk* pk, const uint8_t* witness_le32, size_t n_inputs, uint64_t* advice, size_t n_columns,
                            size_t n_rows, uint64_t* instances, size_t n_instances, size_t* assigned_rows) {
    return guarded([&] {
        SHP_REQUIRE(pk && witness_le32 && advice, SHP_ERR_INVALID_ARG, "witness_columns: null argument");
        const PlanData& P = pk->plan;
        SHP_REQUIRE(!P.phased(), SHP_ERR_INVALID_ARG, "witness_columns: the columns of a circuit with several phases depend on the proof's challenges");
        SHP_REQUIRE(n_inputs == P.n_winputs, SHP_ERR_INVALID_ARG, "witness_columns: wrong number of witness inputs");
        SHP_REQUIRE(n_columns == P.num_advice && n_rows == P.n, SHP_ERR_INVALID_ARG,
                    "witness_columns: the advice block must be num_advice columns of 2^k rows");
        SHP_REQUIRE(!instances || n_instances == P.num_instance, SHP_ERR_INVALID_ARG, "witness_columns: wrong number of public inputs");
        std::vector<Fr> inputs(n_inputs);
        for (size_t i = 0; i < n_inputs; i++) {
            Fr raw;
            memcpy(raw.l, witness_le32 + 32 * i, 32);
            SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "witness_columns: witness input is not a canonical field element");
            inputs[i] = Fr::from_raw(raw);
        }
        Fr* grid = reinterpret_cast<Fr*>(advice);
        memset(grid, 0, (size_t)P.num_advice * P.n * sizeof(Fr));
        std::vector<Fr> scratch(std::max<uint32_t>(P.n_wscratch, 1), Fr::zero());
        run_witness_range(P, inputs, grid, scratch.data(), 0, (uint32_t)P.wops.size());
        if (instances)
            for (size_t i = 0; i < P.instance_map.size(); i++)
                reinterpret_cast<Fr*>(instances)[i] = grid[(size_t)P.instance_map[i].first * P.n + P.instance_map[i].second];
        if (assigned_rows) *assigned_rows = P.adv_rows;
    });
}

int32_t shp_prover_witness_threads(shp_prover* p, uint32_t threads) {
    return guarded([&] {
        SHP_REQUIRE(p, SHP_ERR_INVALID_ARG, "prover_witness_threads: null prover");
        set_witness_threads(p, threads);
    });
}

int32_t shp_prover_mode(shp_prover* p, uint32_t mode) {
    return guarded([&] {
        SHP_REQUIRE(p && mode <= 1, SHP_ERR_INVALID_ARG, "prover_mode: null prover or unknown mode");
        // waves an MSM launch of a throughput lane should have before its lanes take more windows each: fewer waves are
        // fewer partial sums for the reduction trees (a fifth of the kernel's instructions at 512), too few leave the
        // launch to too few CUs.  Measured at 48 lanes, alternated on one box: 128 -> 2498, 256 -> 2520, 384 -> 2506,
        // 512 -> 2490 proofs/s (at 16 lanes and the trees of an earlier build the optimum was 512)
        static const uint64_t tp_waves = getenv("SHP_TP_MIN_WAVES") ? (uint64_t)atol(getenv("SHP_TP_MIN_WAVES")) : 256;
        p->msm_ws.min_waves = mode ? tp_waves : 0;
        p->throughput = mode != 0;
        p->msm_ws.poll_results = !mode && poll_results_enabled();   // pool lanes sleep in their stream waits instead
        if (mode) set_witness_threads(p, 1);
    });
}

int32_t shp_prover_profile(shp_prover* p, int32_t enable) {
    return guarded([&] {
        SHP_REQUIRE(p, SHP_ERR_INVALID_ARG, "prover_profile: null prover");
        p->timers.enabled = enable == 1;                     // 2: the phase clock alone (no events in the stream)
        p->keep_phases = enable != 0;
        p->timers.reset();
        if (p->comb && p->comb_profile_owner) {              // idle pool assumed (bench toggles between steps)
            p->comb->timers.enabled = enable != 0;
            p->comb->timers.reset();
        }
    });
}

int32_t shp_prover_profile_read(shp_prover* p, uint32_t index, const char** name, uint64_t* launches, double* total_ms) {
    return guarded([&] {
        SHP_REQUIRE(p && index < KID_COUNT, SHP_ERR_INVALID_ARG, "prover_profile_read: bad argument");
        if (name) *name = KERNEL_NAMES[index];
        const bool with_comb = p->comb && p->comb_profile_owner;
        if (launches) *launches = p->timers.launches[index] + (with_comb ? p->comb->timers.launches[index] : 0);
        if (total_ms) *total_ms = p->timers.total_ms[index] + (with_comb ? p->comb->timers.total_ms[index] : 0.0);
    });
}

}  // extern "C"

void shp_prover_attach_combiner(shp_prover* p, shp::MsmCombiner* c, bool profile_owner) {
    if (!p) return;
    p->comb = c;
    p->comb_profile_owner = c && profile_owner;
    if (c && !p->comb_event) SHP_HIP(hipEventCreateWithFlags(&p->comb_event, hipEventDisableTiming));
}

extern "C" {

int32_t shp_prover_phase_read(shp_prover* p, uint32_t index, const char** name, double* us) {
    return guarded([&] {
        SHP_REQUIRE(p && index < p->last_phases.size(), SHP_ERR_INVALID_ARG, "prover_phase_read: no such phase");
        if (name) *name = p->last_phases[index].first;
        if (us) *us = p->last_phases[index].second;
    });
}

int32_t shp_prover_destroy(shp_prover* p) {
    return guarded([&] { delete p; });
}

static void proof_out_copy(const std::vector<uint8_t>& proof, const std::vector<Fr>& inst, uint8_t* proof_out, size_t proof_cap,
                           size_t* proof_len, uint8_t* instances_le32, size_t instances_cap) {
    SHP_REQUIRE(proof.size() <= proof_cap, SHP_ERR_INVALID_ARG, "prove: proof buffer too small");
    memcpy(proof_out, proof.data(), proof.size());
    *proof_len = proof.size();
    if (instances_le32) {
        SHP_REQUIRE(inst.size() * 32 <= instances_cap, SHP_ERR_INVALID_ARG, "prove: instance buffer too small");
        for (size_t i = 0; i < inst.size(); i++) {
            Fr raw = inst[i].to_raw();
            memcpy(instances_le32 + 32 * i, raw.l, 32);
        }
    }
}

int32_t shp_prove(shp_prover* p, const uint8_t* witness_le32, size_t n_inputs, const uint8_t seed[32], uint8_t* proof_out,
                  size_t proof_cap, size_t* proof_len, uint8_t* instances_le32, size_t instances_cap) {
    return guarded([&] {
        SHP_REQUIRE(p && witness_le32 && seed && proof_out && proof_len, SHP_ERR_INVALID_ARG, "prove: null argument");
        std::vector<uint8_t> proof;
        std::vector<Fr> inst;
        ProveInput in;
        in.inputs_le32 = witness_le32; in.n_inputs = n_inputs; in.seed = seed;
        prove_impl(p, in, proof, inst);
        proof_out_copy(proof, inst, proof_out, proof_cap, proof_len, instances_le32, instances_cap);
    });
}

int32_t shp_prove_with_randomness(shp_prover* p, const uint8_t* witness_le32, size_t n_inputs, const uint64_t* randomness,
                                  size_t n_draws, uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                                  uint8_t* instances_le32, size_t instances_cap) {
    return guarded([&] {
        SHP_REQUIRE(p && witness_le32 && randomness && proof_out && proof_len, SHP_ERR_INVALID_ARG, "prove_with_randomness: null argument");
        std::vector<uint8_t> proof;
        std::vector<Fr> inst;
        ProveInput in;
        in.inputs_le32 = witness_le32; in.n_inputs = n_inputs;
        in.draws = reinterpret_cast<const Fr*>(randomness); in.n_draws = n_draws;
        prove_impl(p, in, proof, inst);
        proof_out_copy(proof, inst, proof_out, proof_cap, proof_len, instances_le32, instances_cap);
    });
}

int32_t shp_prove_assigned(shp_prover* p, const uint64_t* advice, size_t n_columns, size_t n_rows, size_t assigned_rows,
                           const uint64_t* instances, size_t n_instances, const uint64_t* randomness, size_t n_draws,
                           const uint8_t* seed, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
    return guarded([&] {
        SHP_REQUIRE(p && advice && proof_out && proof_len, SHP_ERR_INVALID_ARG, "prove_assigned: null argument");
        const PlanData& P = p->pk->plan;
        SHP_REQUIRE(n_columns == P.num_advice && n_rows == P.n, SHP_ERR_INVALID_ARG,
                    "prove_assigned: the advice block must be num_advice columns of 2^k rows");
        SHP_REQUIRE(assigned_rows <= P.usable, SHP_ERR_INVALID_ARG, "prove_assigned: more assigned rows than usable rows");
        std::vector<uint8_t> proof;
        std::vector<Fr> inst;
        ProveInput in;
        in.advice = reinterpret_cast<const Fr*>(advice);
        in.assigned_rows = (uint32_t)assigned_rows;
        in.instances = reinterpret_cast<const Fr*>(instances);
        in.n_instances = n_instances;
        if (randomness) { in.draws = reinterpret_cast<const Fr*>(randomness); in.n_draws = n_draws; }
        else in.seed = seed;
        prove_impl(p, in, proof, inst);
        proof_out_copy(proof, inst, proof_out, proof_cap, proof_len, nullptr, 0);
    });
}

}  // extern "C"

// a pool lane's form of shp_prove_assigned: columns that hold their first `rows` rows only, `rows` apart
int32_t shp_prove_assigned_rows(shp_prover* p, const uint64_t* advice, size_t n_columns, size_t rows, const uint64_t* instances,
                                size_t n_instances, const uint64_t* randomness, size_t n_draws, const uint8_t* seed,
                                uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
    return guarded([&] {
        SHP_REQUIRE(p && advice && proof_out && proof_len && rows, SHP_ERR_INVALID_ARG, "prove_assigned: null argument");
        const PlanData& P = p->pk->plan;
        SHP_REQUIRE(n_columns == P.num_advice && rows <= P.usable, SHP_ERR_INVALID_ARG, "prove_assigned: bad column block");
        std::vector<uint8_t> proof;
        std::vector<Fr> inst;
        ProveInput in;
        in.advice = reinterpret_cast<const Fr*>(advice);
        in.advice_stride = rows;
        in.assigned_rows = (uint32_t)rows;
        in.instances = reinterpret_cast<const Fr*>(instances);
        in.n_instances = n_instances;
        if (randomness) { in.draws = reinterpret_cast<const Fr*>(randomness); in.n_draws = n_draws; }
        else in.seed = seed;
        prove_impl(p, in, proof, inst);
        proof_out_copy(proof, inst, proof_out, proof_cap, proof_len, nullptr, 0);
    });
}

extern "C" {

int32_t shp_prove_assigned_phased(shp_prover* p, shp_phase_fn synthesize, void* user, size_t n_columns, size_t n_rows,
                                  size_t assigned_rows, const uint64_t* instances, size_t n_instances, const uint64_t* randomness,
                                  size_t n_draws, const uint8_t* seed, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
    return guarded([&] {
        SHP_REQUIRE(p && synthesize && proof_out && proof_len, SHP_ERR_INVALID_ARG, "prove_assigned_phased: null argument");
        const PlanData& P = p->pk->plan;
        SHP_REQUIRE(n_columns == P.num_advice && n_rows == P.n, SHP_ERR_INVALID_ARG,
                    "prove_assigned_phased: the advice block must be num_advice columns of 2^k rows");
        SHP_REQUIRE(assigned_rows <= P.usable, SHP_ERR_INVALID_ARG, "prove_assigned_phased: more assigned rows than usable rows");
        std::vector<uint8_t> proof;
        std::vector<Fr> inst;
        ProveInput in;
        in.phase_fn = synthesize;
        in.phase_user = user;
        in.assigned_rows = (uint32_t)assigned_rows;
        in.instances = reinterpret_cast<const Fr*>(instances);
        in.n_instances = n_instances;
        if (randomness) { in.draws = reinterpret_cast<const Fr*>(randomness); in.n_draws = n_draws; }
        else in.seed = seed;
        prove_impl(p, in, proof, inst);
        proof_out_copy(proof, inst, proof_out, proof_cap, proof_len, nullptr, 0);
    });
}

int32_t shp_poseidon2_hash_host(const uint8_t* inputs_le32, uint32_t arity, uint8_t out_le32[32]) {
    return guarded([&] {
        SHP_REQUIRE(inputs_le32 && out_le32 && arity >= 1 && arity <= 7, SHP_ERR_INVALID_ARG, "poseidon2_hash_host: bad argument");
        Fr in[7];
        for (uint32_t i = 0; i < arity; i++) {
            Fr raw;
            memcpy(raw.l, inputs_le32 + 32 * i, 32);
            SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "poseidon2_hash_host: non-canonical input");
            in[i] = Fr::from_raw(raw);
        }
        Fr o = poseidon2_hash_host(in, arity).to_raw();
        memcpy(out_le32, o.l, 32);
    });
}

}  // extern "C"
