// Circuit plan parser: the constraint system + fixed data + witness program the
// prover is driven by.  Layout documented in zkos-monorepo_amd/circuits/builder.py;
// the fields are the quantities the reference's verifier generator reads off a halo2
// ConstraintSystem (/root/reference/crates/halo2-verifier/src/lib/codegen/util.rs:42-131).
// Version 2 adds phases: advice columns assigned after challenges have been squeezed (util.rs:70-96, 133-165).
#pragma once
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "common.hpp"
#include "field.hip.hpp"

namespace shp {

// OP_CHAL (a = challenge index) exists in plans only: the device programs read the proof's challenges as constants
// placed behind the plan's own (PlanData::device_prog)
enum { OP_CONST = 0, OP_ADV, OP_FIX, OP_INST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_OUT, OP_CHAL };
enum { W_INPUT = 0, W_CONST, W_COPY, W_ADD, W_SUB, W_MUL, W_ADDC, W_MULC, W_POW7, W_BITS, W_SHR, W_CHAL };
enum { COL_ADVICE = 0, COL_FIXED = 1, COL_INSTANCE = 2 };

struct Instr { uint32_t op, dst, a, b; };
struct Query { uint32_t col; int32_t rot; };
struct WOp { uint32_t op, dc, dr, ac, ar, bc, br, imm; };
// a range of witness ops that is one Poseidon2 permutation written out row by row (found by the prover when it takes a
// plan, never part of the plan's bytes): the state columns in state order, the row of the input state
struct WPermutation { uint32_t lo, hi, r0, cols[8]; };
struct LookupPlan {
    uint32_t m = 0, in_slots = 0, tab_slots = 0;
    std::vector<Instr> in_prog, tab_prog;
};

struct PlanData {
    uint32_t k = 0, num_advice = 0, num_fixed = 0, num_instance = 0, bf = 0, degree = 0;
    uint32_t gate_slots = 0, num_gates = 0, n_winputs = 0, n_wscratch = 0;
    std::vector<Query> adv_q, fix_q, inst_q;
    std::vector<std::pair<uint32_t, uint32_t>> perm_cols;        // (type, index)
    std::vector<Fr> consts;                                      // Montgomery
    std::vector<Instr> gate_prog;
    std::vector<LookupPlan> lookups;
    std::vector<std::vector<std::pair<uint32_t, Fr>>> fixed;     // sparse (row, value)
    std::vector<std::vector<uint32_t>> mapping;                  // per permutation column: target col * n + row
    std::vector<WOp> wops;
    std::vector<std::pair<uint32_t, uint32_t>> instance_map;     // advice cell of public input i
    // phases (version 2; a version-1 plan has one phase, no challenges)
    uint32_t num_phases = 1;
    std::vector<uint32_t> advice_phase;                          // per advice column
    std::vector<uint32_t> challenge_phase;                       // challenge i is squeezed after the commitments of this phase
    std::vector<uint32_t> wops_end;                              // witness ops of phase p: [wops_end[p-1], wops_end[p])
    uint8_t plan_hash[32];
    std::vector<WPermutation> wperms;                            // sorted by lo, disjoint (see WPermutation)
    // derived
    uint32_t n = 0, usable = 0, ext_k = 0, chunk = 0, num_sets = 0;
    uint32_t adv_rows = 0;       // rows [adv_rows, usable) of every advice column are never assigned (stay zero)
    int32_t rot_last = 0;

    static Fr read_fr(const uint8_t* p) {
        Fr raw;
        memcpy(raw.l, p, 32);
        SHP_REQUIRE(!raw.geq_p(), SHP_ERR_FORMAT, "plan: non-canonical field element");
        return Fr::from_raw(raw);
    }

    void parse(const uint8_t* d, size_t len);
    // the plan's bytes again (parse(serialize()) is the identity on everything parse reads)
    std::vector<uint8_t> serialize() const;
    // generate_keys_with_min_k (/root/reference/crates/shielder_bindings/build.rs:22): the smallest k whose usable rows
    // (2^k - (bf + 1)) hold everything the plan places on rows -- fixed values, copy constraints, witness cells, public
    // inputs -- and the same plan laid out for another k (rows keep their index, the permutation mapping is re-encoded)
    uint32_t min_k() const;
    PlanData retarget(uint32_t new_k) const;
    bool phased() const { return num_phases > 1 || !challenge_phase.empty(); }
    uint32_t wops_begin(uint32_t phase) const { return phase ? wops_end[phase - 1] : 0; }
    // maximal runs [first, first + count) of consecutive advice columns that belong to `phase`
    std::vector<std::pair<uint32_t, uint32_t>> phase_runs(uint32_t phase) const {
        std::vector<std::pair<uint32_t, uint32_t>> runs;
        for (uint32_t c = 0; c < num_advice; c++) {
            if (advice_phase[c] != phase) continue;
            if (!runs.empty() && runs.back().first + runs.back().second == c) runs.back().second++;
            else runs.push_back({c, 1});
        }
        return runs;
    }
    // a program as the device kernels read it: challenge i is constant consts.size() + i of the lane's own table
    std::vector<Instr> device_prog(const std::vector<Instr>& prog) const {
        std::vector<Instr> out = prog;
        for (Instr& in : out)
            if (in.op == OP_CHAL) { in.op = OP_CONST; in.a += (uint32_t)consts.size(); }
        return out;
    }
};

inline void PlanData::parse(const uint8_t* d, size_t len) {
    SHP_REQUIRE(len >= 84 && memcmp(d, "SHPL", 4) == 0, SHP_ERR_FORMAT, "plan: bad magic");
    uint32_t h[20];
    memcpy(h, d + 4, 80);
    SHP_REQUIRE(h[0] == 1 || h[0] == 2, SHP_ERR_FORMAT, "plan: unsupported version");
    k = h[1]; num_advice = h[2]; num_fixed = h[3]; num_instance = h[4]; bf = h[5]; degree = h[6];
    const uint32_t naq = h[7], nfq = h[8], niq = h[9], npc = h[10], ncst = h[11], ngi = h[12];
    gate_slots = h[13]; num_gates = h[14];
    const uint32_t nlk = h[15];
    n_winputs = h[16];
    const uint32_t nwo = h[17];
    n_wscratch = h[18];
    const uint32_t nim = h[19];
    SHP_REQUIRE(k >= 4 && k <= 20 && degree >= 3 && degree <= 16 && bf >= 3 && bf < 64, SHP_ERR_FORMAT, "plan: bad header");
    // counts are bounded before anything is sized from them (a corrupted header must not allocate)
    SHP_REQUIRE(num_advice >= 1 && num_advice <= 1024 && num_fixed <= 1024 && num_instance <= 1024 && nlk <= 64 && npc <= 1024 &&
                    n_winputs <= (1u << 20) && n_wscratch <= (1u << 24),
                SHP_ERR_FORMAT, "plan: implausible column / lookup counts");
    n = 1u << k;
    size_t off = 84;
    auto need = [&](size_t bytes) { SHP_REQUIRE(off + bytes <= len, SHP_ERR_FORMAT, "plan: truncated"); };
    auto rd_queries = [&](std::vector<Query>& v, uint32_t cnt, uint32_t ncols) {
        need((size_t)cnt * 8);
        v.resize(cnt);
        for (uint32_t i = 0; i < cnt; i++) {
            memcpy(&v[i].col, d + off, 4);
            memcpy(&v[i].rot, d + off + 4, 4);
            off += 8;
            SHP_REQUIRE(v[i].col < ncols && v[i].rot > -64 && v[i].rot < 64, SHP_ERR_FORMAT, "plan: bad query");
        }
    };
    rd_queries(adv_q, naq, num_advice);
    rd_queries(fix_q, nfq, num_fixed);
    rd_queries(inst_q, niq, 1);
    need((size_t)npc * 8);
    perm_cols.resize(npc);
    for (uint32_t i = 0; i < npc; i++) {
        memcpy(&perm_cols[i].first, d + off, 4);
        memcpy(&perm_cols[i].second, d + off + 4, 4);
        off += 8;
        uint32_t t = perm_cols[i].first, ix = perm_cols[i].second;
        SHP_REQUIRE((t == COL_ADVICE && ix < num_advice) || (t == COL_FIXED && ix < num_fixed) || (t == COL_INSTANCE && ix == 0),
                    SHP_ERR_FORMAT, "plan: bad permutation column");
    }
    need((size_t)ncst * 32);
    consts.resize(ncst);
    for (uint32_t i = 0; i < ncst; i++) { consts[i] = read_fr(d + off); off += 32; }
    auto rd_prog = [&](std::vector<Instr>& v, uint32_t cnt, uint32_t slots) {
        need((size_t)cnt * 16);
        v.resize(cnt);
        memcpy(v.data(), d + off, (size_t)cnt * 16);
        off += (size_t)cnt * 16;
        for (auto& in : v) {
            SHP_REQUIRE(in.op <= OP_CHAL, SHP_ERR_FORMAT, "plan: bad opcode");
            bool ok = in.dst < slots || in.op == OP_OUT;
            if (in.op == OP_CONST) ok = ok && in.a < ncst;
            else if (in.op == OP_CHAL) ok = ok && h[0] == 2;     // the index is checked once the challenges are known
            else if (in.op == OP_ADV) ok = ok && in.a < naq;
            else if (in.op == OP_FIX) ok = ok && in.a < nfq;
            else if (in.op == OP_INST) ok = ok && in.a < niq;
            else if (in.op == OP_NEG || in.op == OP_OUT) ok = ok && in.a < slots;
            else ok = ok && in.a < slots && in.b < slots;
            SHP_REQUIRE(ok, SHP_ERR_FORMAT, "plan: bad program operand");
        }
    };
    SHP_REQUIRE(gate_slots >= 1 && gate_slots <= 48, SHP_ERR_FORMAT, "plan: too many expression slots");
    rd_prog(gate_prog, ngi, gate_slots);
    lookups.resize(nlk);
    for (auto& lk : lookups) {
        need(20);
        uint32_t hh[5];
        memcpy(hh, d + off, 20);
        off += 20;
        lk.m = hh[0]; lk.in_slots = hh[2]; lk.tab_slots = hh[4];
        SHP_REQUIRE(lk.in_slots >= 1 && lk.in_slots <= 48 && lk.tab_slots >= 1 && lk.tab_slots <= 48, SHP_ERR_FORMAT,
                    "plan: too many lookup slots");
        rd_prog(lk.in_prog, hh[1], lk.in_slots);
        rd_prog(lk.tab_prog, hh[3], lk.tab_slots);
    }
    fixed.resize(num_fixed);
    for (auto& col : fixed) {
        need(4);
        uint32_t nnz;
        memcpy(&nnz, d + off, 4);
        off += 4;
        need((size_t)nnz * 36);
        col.resize(nnz);
        for (uint32_t i = 0; i < nnz; i++) {
            memcpy(&col[i].first, d + off, 4);
            SHP_REQUIRE(col[i].first < n, SHP_ERR_FORMAT, "plan: fixed row out of range");
            col[i].second = read_fr(d + off + 4);
            off += 36;
        }
    }
    mapping.resize(npc);
    for (auto& m : mapping) {
        need((size_t)n * 4);
        m.resize(n);
        memcpy(m.data(), d + off, (size_t)n * 4);
        off += (size_t)n * 4;
        for (uint32_t v : m) SHP_REQUIRE(v < npc * n, SHP_ERR_FORMAT, "plan: permutation target out of range");
    }
    need((size_t)nwo * 32);
    wops.resize(nwo);
    memcpy(wops.data(), d + off, (size_t)nwo * 32);
    off += (size_t)nwo * 32;
    for (auto& w : wops) {
        auto cell_ok = [&](uint32_t c, uint32_t r) {
            return (c < num_advice && r < n) || (c == num_advice && r < (n_wscratch ? n_wscratch : 1));
        };
        bool ok = w.op <= W_CHAL && cell_ok(w.dc, w.dr);
        if (w.op == W_INPUT) ok = ok && w.imm < n_winputs;
        else if (w.op == W_CHAL) ok = ok && h[0] == 2;
        else if (w.op == W_CONST) ok = ok && w.imm < ncst;
        else {
            ok = ok && cell_ok(w.ac, w.ar);
            if (w.op == W_ADD || w.op == W_SUB || w.op == W_MUL) ok = ok && cell_ok(w.bc, w.br);
            if (w.op == W_ADDC || w.op == W_MULC) ok = ok && w.imm < ncst;
            if (w.op == W_BITS || w.op == W_SHR) ok = ok && w.imm < 254;
        }
        SHP_REQUIRE(ok, SHP_ERR_FORMAT, "plan: bad witness op");
    }
    need((size_t)nim * 8);
    instance_map.resize(nim);
    for (uint32_t i = 0; i < nim; i++) {
        memcpy(&instance_map[i].first, d + off, 4);
        memcpy(&instance_map[i].second, d + off + 4, 4);
        off += 8;
        SHP_REQUIRE(instance_map[i].first < num_advice && instance_map[i].second < n, SHP_ERR_FORMAT, "plan: bad instance cell");
    }
    num_phases = 1;
    advice_phase.assign(num_advice, 0);
    challenge_phase.clear();
    wops_end.assign(1, nwo);
    if (h[0] == 2) {
        need(8);
        uint32_t nch;
        memcpy(&num_phases, d + off, 4);
        memcpy(&nch, d + off + 4, 4);
        off += 8;
        SHP_REQUIRE(num_phases >= 1 && num_phases <= 8 && nch <= 64, SHP_ERR_FORMAT, "plan: implausible phase / challenge counts");
        auto rd_u32s = [&](std::vector<uint32_t>& v, uint32_t cnt) {
            need((size_t)cnt * 4);
            v.resize(cnt);
            if (cnt) memcpy(v.data(), d + off, (size_t)cnt * 4);
            off += (size_t)cnt * 4;
        };
        rd_u32s(advice_phase, num_advice);
        rd_u32s(challenge_phase, nch);
        rd_u32s(wops_end, num_phases);
        std::vector<uint32_t> cols(num_phases, 0), chals(num_phases, 0);
        for (uint32_t ph : advice_phase) {
            SHP_REQUIRE(ph < num_phases, SHP_ERR_FORMAT, "plan: advice column in a phase that does not exist");
            cols[ph]++;
        }
        for (uint32_t ph : challenge_phase) {
            SHP_REQUIRE(ph < num_phases, SHP_ERR_FORMAT, "plan: challenge after a phase that does not exist");
            chals[ph]++;
        }
        for (uint32_t ph = 0; ph < num_phases; ph++) {
            SHP_REQUIRE(cols[ph] >= 1, SHP_ERR_FORMAT, "plan: a phase without advice columns");
            // the verifier squeezes at least one challenge after every phase but the last (Halo2Verifier.sol:278-281)
            SHP_REQUIRE(ph + 1 == num_phases || chals[ph] >= 1, SHP_ERR_FORMAT, "plan: a phase before the last draws no challenge");
            SHP_REQUIRE(wops_end[ph] <= nwo && (!ph || wops_end[ph - 1] <= wops_end[ph]), SHP_ERR_FORMAT, "plan: bad phase bounds of the witness program");
        }
        SHP_REQUIRE(wops_end[num_phases - 1] == nwo, SHP_ERR_FORMAT, "plan: witness ops behind the last phase");
        // an op of phase p may read the challenges of earlier phases only, and writes advice cells of its own phase only
        // (halo2 records the advice of the current phase and hands out the challenges squeezed before it)
        for (uint32_t ph = 0; ph < num_phases; ph++)
            for (uint32_t i = wops_begin(ph); i < wops_end[ph]; i++) {
                const WOp& w = wops[i];
                if (w.op == W_CHAL) SHP_REQUIRE(w.imm < nch && challenge_phase[w.imm] < ph, SHP_ERR_FORMAT, "plan: witness op reads a challenge that is not known yet");
                if (w.dc < num_advice) SHP_REQUIRE(advice_phase[w.dc] == ph, SHP_ERR_FORMAT, "plan: witness op writes a column of another phase");
            }
        auto check_prog = [&](const std::vector<Instr>& v) {
            for (const Instr& in : v)
                if (in.op == OP_CHAL) SHP_REQUIRE(in.a < nch, SHP_ERR_FORMAT, "plan: bad challenge index");
        };
        check_prog(gate_prog);
        for (const auto& lk : lookups) { check_prog(lk.in_prog); check_prog(lk.tab_prog); }
        for (const auto& im : instance_map)
            SHP_REQUIRE(advice_phase[im.first] == 0, SHP_ERR_FORMAT, "plan: a public input is not a first-phase cell");
    }
    SHP_REQUIRE(off == len && nim == num_instance, SHP_ERR_FORMAT, "plan: trailing bytes or instance count mismatch");
    usable = n - (bf + 1);
    for (auto& w : wops)
        if (w.dc < num_advice) adv_rows = std::max(adv_rows, w.dr + 1);
    adv_rows = std::min(adv_rows, usable);
    ext_k = k;
    while ((1ull << ext_k) < (uint64_t)n * (degree - 1)) ext_k++;
    chunk = degree - 2;
    num_sets = npc ? (npc + chunk - 1) / chunk : 0;
    rot_last = -(int32_t)(bf + 1);
}

inline std::vector<uint8_t> PlanData::serialize() const {
    std::vector<uint8_t> out;
    auto put = [&](const void* p, size_t bytes) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        out.insert(out.end(), b, b + bytes);
    };
    auto put_fr = [&](const Fr& v) {
        const Fr raw = v.to_raw();
        put(raw.l, 32);
    };
    const uint32_t h[20] = {phased() ? 2u : 1u, k, num_advice, num_fixed, num_instance, bf, degree, (uint32_t)adv_q.size(), (uint32_t)fix_q.size(),
                            (uint32_t)inst_q.size(), (uint32_t)perm_cols.size(), (uint32_t)consts.size(), (uint32_t)gate_prog.size(),
                            gate_slots, num_gates, (uint32_t)lookups.size(), n_winputs, (uint32_t)wops.size(), n_wscratch,
                            (uint32_t)instance_map.size()};
    put("SHPL", 4);
    put(h, 80);
    for (const auto* qs : {&adv_q, &fix_q, &inst_q})
        for (const Query& q : *qs) { put(&q.col, 4); put(&q.rot, 4); }
    for (const auto& pc : perm_cols) { put(&pc.first, 4); put(&pc.second, 4); }
    for (const Fr& c : consts) put_fr(c);
    if (!gate_prog.empty()) put(gate_prog.data(), gate_prog.size() * sizeof(Instr));
    for (const LookupPlan& lk : lookups) {
        const uint32_t hh[5] = {lk.m, (uint32_t)lk.in_prog.size(), lk.in_slots, (uint32_t)lk.tab_prog.size(), lk.tab_slots};
        put(hh, 20);
        if (!lk.in_prog.empty()) put(lk.in_prog.data(), lk.in_prog.size() * sizeof(Instr));
        if (!lk.tab_prog.empty()) put(lk.tab_prog.data(), lk.tab_prog.size() * sizeof(Instr));
    }
    for (const auto& col : fixed) {
        const uint32_t nnz = (uint32_t)col.size();
        put(&nnz, 4);
        for (const auto& e : col) { put(&e.first, 4); put_fr(e.second); }
    }
    for (const auto& m : mapping) put(m.data(), m.size() * 4);
    if (!wops.empty()) put(wops.data(), wops.size() * sizeof(WOp));
    for (const auto& im : instance_map) { put(&im.first, 4); put(&im.second, 4); }
    if (phased()) {
        const uint32_t nch = (uint32_t)challenge_phase.size();
        put(&num_phases, 4);
        put(&nch, 4);
        put(advice_phase.data(), advice_phase.size() * 4);
        if (nch) put(challenge_phase.data(), (size_t)nch * 4);
        put(wops_end.data(), wops_end.size() * 4);
    }
    return out;
}

inline uint32_t PlanData::min_k() const {
    uint32_t rows = 1;                                           // rows in use: highest row index + 1
    for (const auto& col : fixed)
        for (const auto& e : col) rows = std::max(rows, e.first + 1);
    for (size_t c = 0; c < mapping.size(); c++)
        for (uint32_t r = 0; r < n; r++)
            if (mapping[c][r] != (uint32_t)(c * n + r)) rows = std::max(rows, r + 1);
    for (const WOp& w : wops)
        if (w.dc < num_advice) rows = std::max(rows, w.dr + 1);
    for (const auto& im : instance_map) rows = std::max(rows, im.second + 1);
    rows = std::max(rows, num_instance);
    uint32_t kk = 4;                                             // the smallest k a plan may have
    while ((1ull << kk) < (uint64_t)rows + bf + 1 || (1u << kk) < 64) kk++;
    return std::min(kk, k);
}

inline PlanData PlanData::retarget(uint32_t new_k) const {
    SHP_REQUIRE(new_k >= min_k() && new_k <= 20, SHP_ERR_INVALID_ARG, "plan: the circuit does not fit the requested k");
    PlanData q = *this;
    q.k = new_k;
    const uint32_t nn = 1u << new_k;
    for (size_t c = 0; c < mapping.size(); c++) {
        q.mapping[c].assign(nn, 0);
        for (uint32_t r = 0; r < nn; r++) {
            const uint32_t tgt = r < n ? mapping[c][r] : (uint32_t)(c * n + r);
            const uint32_t tc = r < n ? tgt / n : (uint32_t)c, tr = r < n ? tgt % n : r;
            SHP_REQUIRE(tr < nn, SHP_ERR_INTERNAL, "plan: permutation target beyond the new domain");
            q.mapping[c][r] = tc * nn + tr;
        }
    }
    const std::vector<uint8_t> bytes = q.serialize();            // derived fields (n, usable, ext_k ...) through parse
    PlanData out;
    out.parse(bytes.data(), bytes.size());
    return out;
}

}  // namespace shp
