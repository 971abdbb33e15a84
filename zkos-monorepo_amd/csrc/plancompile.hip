// Circuit description -> circuit plan: the compile step of the plan emitter, inside the library.
//
// A host that holds the halo2 ConstraintSystem (the reference's circuits: Circuit::configure + a keygen-time
// synthesize, /root/reference/crates/shielder_bindings/build.rs:19-32) can serialise what it sees with a trivial
// walker -- the quantities the reference's verifier generator reads in
// /root/reference/crates/halo2-verifier/src/lib/codegen/util.rs:42-131: queries in the constraint system's own order,
// permutation columns, gate polynomials and lookup expressions as expression nodes, fixed column values, the copy
// constraints (or the finished permutation mapping) -- and leaves everything that is specific to this prover to
// shp_plan_compile: blinding factors and degree (halo2's ConstraintSystem::{blinding_factors, degree}), common
// subexpression merging, the linear three-address programs of the gate / lookup interpreters with their slot
// allocation (leaves are re-materialised at every use instead of held in a slot: a slot costs registers or LDS in
// prog_fold_*_kernel, a reload is one L2-resident read), and the permutation mapping built from the copy constraints
// the way halo2's permutation::keygen::Assembly::copy joins cycles.
//
// Circuit description, little endian ("SHCD", version 1; version 2 for a circuit with several phases or challenges):
//   header  20 x u32: version, k, num_advice, num_fixed, num_instance, n_adv_q, n_fix_q, n_inst_q, n_perm_cols,
//                     n_consts, n_nodes, n_gates, n_lookups, n_copies, has_mapping, n_winputs, n_wops, n_wscratch,
//                     n_instance_map, n_challenges (version 1: 0)
//   advice / fixed / instance queries   (u32 column, i32 rotation) each, in the constraint system's order
//   permutation columns                 (u32 type: 0 advice 1 fixed 2 instance, u32 index) each
//   constants                           32 bytes little-endian canonical each
//   nodes                               (u32 kind, u32 a, u32 b) each, children before parents:
//                                       0 CONST(a = constant) 1 ADVICE(a = query) 2 FIXED(a = query) 3 INSTANCE(a = query)
//                                       4 SUM(a, b) 5 DIFFERENCE(a, b) 6 PRODUCT(a, b) 7 NEGATED(a) 8 SCALED(a, b = constant)
//                                       9 CHALLENGE(a = challenge index; version 2)
//   gates                               u32 root node each, in gate / polynomial order
//   lookups                             u32 m, m input roots, m table roots each
//   fixed columns                       per column: u32 nnz, then (u32 row, 32-byte value) each
//   copy constraints (has_mapping = 0)  (u32 type, index, row) x 2 each   |   mapping (has_mapping = 1) u32[2^k] per column
//   witness program                     8 x u32 per op (the plan's own encoding; may be empty: shp_prove_assigned)
//   instance map                        (u32 advice column, u32 row) per public input
//   phases (version 2)                  u32 num_phases, then the phase of every advice column (cs.advice_column_phase()), the
//                                       phase of every challenge (cs.challenge_phase()), the end of every phase in the witness
//                                       program (u32 each)
// Host code only; the .hip suffix is for the build.
#include <string.h>

#include <map>
#include <tuple>

#include "plan.hpp"

namespace {
using namespace shp;

enum { N_CONST = 0, N_ADV, N_FIX, N_INST, N_SUM, N_DIFF, N_PROD, N_NEG, N_SCALED, N_CHAL };

struct Node { uint32_t kind, a, b; };

struct Reader {
    const uint8_t* d;
    size_t len, off = 0;
    void need(size_t n) const { SHP_REQUIRE(off + n <= len && off + n >= off, SHP_ERR_FORMAT, "circuit description: truncated"); }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, d + off, 4); off += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
};

// hash-consed expression DAG of one description + the compile of a list of roots into a three-address program
struct Compiler {
    std::vector<Node> nodes;                 // canonical (merged) nodes
    std::vector<uint32_t> canon;             // description node -> canonical node
    std::vector<uint32_t> degree;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> intern;

    uint32_t add(Node nd) {
        if (nd.kind == N_SCALED) {           // SCALED(e, c) = PRODUCT(e, CONST c)
            const uint32_t c = add(Node{N_CONST, nd.b, 0});
            nd = Node{N_PROD, nd.a, c};
        }
        auto key = std::make_tuple(nd.kind, nd.a, nd.b);
        auto it = intern.find(key);
        if (it != intern.end()) return it->second;
        uint32_t deg = 0;
        switch (nd.kind) {
            case N_CONST: case N_CHAL: deg = 0; break;
            case N_ADV: case N_FIX: case N_INST: deg = 1; break;
            case N_SUM: case N_DIFF: deg = std::max(degree[nd.a], degree[nd.b]); break;
            case N_PROD: deg = degree[nd.a] + degree[nd.b]; break;
            default: deg = degree[nd.a]; break;
        }
        nodes.push_back(nd);
        degree.push_back(deg);
        intern[key] = (uint32_t)nodes.size() - 1;
        return (uint32_t)nodes.size() - 1;
    }
    static bool leaf(uint32_t kind) { return kind <= N_INST || kind == N_CHAL; }

    // roots -> program (OUT after each root, in order) and the number of slots it needs
    void compile(const std::vector<uint32_t>& roots, std::vector<Instr>& prog, uint32_t& nslots) const {
        // 1. occurrences in evaluation order: inner nodes once, leaves once per use
        struct Occ { uint32_t node, kid[2], nkids; };
        std::vector<Occ> order;
        std::map<uint32_t, uint32_t> placed;            // inner node -> occurrence
        std::vector<std::pair<uint32_t, bool>> seq;     // (occurrence, is the OUT marker)
        // iterative post-order (expression trees of real circuits are thousands of nodes deep in places)
        struct Frame { uint32_t node; uint32_t next; uint32_t kid[2]; };
        auto visit = [&](uint32_t root) -> uint32_t {
            if (!leaf(nodes[root].kind)) {
                auto it = placed.find(root);
                if (it != placed.end()) return it->second;
            }
            std::vector<Frame> st{Frame{root, 0, {0, 0}}};
            uint32_t result = 0;
            while (!st.empty()) {
                Frame& f = st.back();
                const Node& nd = nodes[f.node];
                const uint32_t nk = leaf(nd.kind) ? 0 : (nd.kind == N_NEG ? 1 : 2);
                if (f.next < nk) {
                    const uint32_t child = f.next == 0 ? nd.a : nd.b;
                    auto it = leaf(nodes[child].kind) ? placed.end() : placed.find(child);
                    if (it != placed.end()) {
                        f.kid[f.next++] = it->second;
                    } else {
                        st.push_back(Frame{child, 0, {0, 0}});
                    }
                    continue;
                }
                const uint32_t occ = (uint32_t)order.size();
                order.push_back(Occ{f.node, {f.kid[0], f.kid[1]}, nk});
                seq.push_back({occ, false});
                if (!leaf(nd.kind)) placed[f.node] = occ;
                st.pop_back();
                if (st.empty()) result = occ;
                else { Frame& p = st.back(); p.kid[p.next++] = occ; }
            }
            return result;
        };
        for (uint32_t r : roots) {
            const uint32_t occ = visit(r);
            seq.push_back({occ, true});
        }
        // 2. last use of every occurrence
        std::vector<uint32_t> last_use(order.size(), 0);
        for (uint32_t pos = 0; pos < seq.size(); pos++) {
            const Occ& o = order[seq[pos].first];
            if (seq[pos].second) last_use[seq[pos].first] = pos;
            else for (uint32_t k = 0; k < o.nkids; k++) last_use[o.kid[k]] = pos;
        }
        // 3. slots: an operand whose last use is this instruction frees its slot before the destination is chosen
        std::vector<uint32_t> slot_of(order.size(), 0), free_list;
        nslots = 0;
        prog.clear();
        for (uint32_t pos = 0; pos < seq.size(); pos++) {
            const uint32_t occ = seq[pos].first;
            const Occ& o = order[occ];
            if (seq[pos].second) {
                prog.push_back(Instr{OP_OUT, 0, slot_of[occ], 0});
                if (last_use[occ] == pos) free_list.push_back(slot_of[occ]);
                continue;
            }
            const uint32_t ka = o.nkids > 0 ? slot_of[o.kid[0]] : 0, kb = o.nkids > 1 ? slot_of[o.kid[1]] : 0;
            for (uint32_t k = 0; k < o.nkids; k++) {
                if (k == 1 && o.kid[1] == o.kid[0]) break;
                if (last_use[o.kid[k]] == pos) free_list.push_back(slot_of[o.kid[k]]);
            }
            uint32_t dst;
            if (!free_list.empty()) { dst = free_list.back(); free_list.pop_back(); }
            else dst = nslots++;
            slot_of[occ] = dst;
            const Node& nd = nodes[o.node];
            switch (nd.kind) {
                case N_CONST: prog.push_back(Instr{OP_CONST, dst, nd.a, 0}); break;
                case N_ADV: prog.push_back(Instr{OP_ADV, dst, nd.a, 0}); break;
                case N_FIX: prog.push_back(Instr{OP_FIX, dst, nd.a, 0}); break;
                case N_INST: prog.push_back(Instr{OP_INST, dst, nd.a, 0}); break;
                case N_CHAL: prog.push_back(Instr{OP_CHAL, dst, nd.a, 0}); break;
                case N_SUM: prog.push_back(Instr{OP_ADD, dst, ka, kb}); break;
                case N_DIFF: prog.push_back(Instr{OP_SUB, dst, ka, kb}); break;
                case N_PROD: prog.push_back(Instr{OP_MUL, dst, ka, kb}); break;
                default: prog.push_back(Instr{OP_NEG, dst, ka, 0}); break;
            }
        }
        nslots = std::max<uint32_t>(nslots, 1);
    }
};

void compile_description(const uint8_t* d, size_t len, std::vector<uint8_t>& out) {
    SHP_REQUIRE(d && len >= 84 && memcmp(d, "SHCD", 4) == 0, SHP_ERR_FORMAT, "circuit description: bad magic");
    Reader r{d, len, 4};
    uint32_t h[20];
    for (auto& v : h) v = r.u32();
    SHP_REQUIRE((h[0] == 1 && h[19] == 0) || (h[0] == 2 && h[19] <= 64), SHP_ERR_FORMAT, "circuit description: unsupported version");
    const uint32_t nch = h[19];
    PlanData P;
    P.k = h[1]; P.num_advice = h[2]; P.num_fixed = h[3]; P.num_instance = h[4];
    const uint32_t naq = h[5], nfq = h[6], niq = h[7], npc = h[8], ncst = h[9], nn = h[10], ng = h[11], nlk = h[12], ncp = h[13],
                   has_map = h[14], nwo = h[16], nim = h[18];
    P.n_winputs = h[15]; P.n_wscratch = h[17];
    SHP_REQUIRE(P.k >= 4 && P.k <= 20 && P.num_advice >= 1 && P.num_advice <= 1024 && P.num_fixed <= 1024 && P.num_instance <= 1024 &&
                    npc <= 1024 && nlk <= 64 && nn <= (1u << 24) && ng <= (1u << 20) && ncst <= (1u << 20) && has_map <= 1,
                SHP_ERR_FORMAT, "circuit description: implausible header");
    // every count is followed by at least that many fixed-size records: one that the remaining bytes cannot hold is
    // refused before anything is sized by it (queries 8 B, permutation columns 8 B, constants 32 B, nodes 12 B, roots 4 B,
    // copy constraints 24 B, witness operations 32 B, instance cells 8 B)
    SHP_REQUIRE((uint64_t)naq * 8 + (uint64_t)nfq * 8 + (uint64_t)niq * 8 + (uint64_t)npc * 8 + (uint64_t)ncst * 32 + (uint64_t)nn * 12 +
                        (uint64_t)ng * 4 + (has_map ? 0 : (uint64_t)ncp * 24) + (uint64_t)nwo * 32 + (uint64_t)nim * 8 <= len,
                SHP_ERR_FORMAT, "circuit description: header counts exceed the description's length");
    P.n = 1u << P.k;
    const uint32_t n = P.n;
    auto queries = [&](std::vector<Query>& v, uint32_t cnt, uint32_t ncols) {
        v.resize(cnt);
        for (auto& q : v) {
            q.col = r.u32(); q.rot = r.i32();
            SHP_REQUIRE(q.col < ncols && q.rot > -64 && q.rot < 64, SHP_ERR_FORMAT, "circuit description: bad query");
        }
    };
    queries(P.adv_q, naq, P.num_advice);
    queries(P.fix_q, nfq, P.num_fixed);
    queries(P.inst_q, niq, 1);
    P.perm_cols.resize(npc);
    for (auto& pc : P.perm_cols) {
        pc.first = r.u32(); pc.second = r.u32();
        SHP_REQUIRE((pc.first == COL_ADVICE && pc.second < P.num_advice) || (pc.first == COL_FIXED && pc.second < P.num_fixed) ||
                        (pc.first == COL_INSTANCE && pc.second == 0), SHP_ERR_FORMAT, "circuit description: bad permutation column");
    }
    P.consts.resize(ncst);
    for (auto& c : P.consts) { r.need(32); c = PlanData::read_fr(d + r.off); r.off += 32; }
    Compiler C;
    C.canon.resize(nn);
    for (uint32_t i = 0; i < nn; i++) {
        Node nd{r.u32(), r.u32(), r.u32()};
        SHP_REQUIRE(nd.kind <= N_CHAL, SHP_ERR_FORMAT, "circuit description: bad node kind");
        bool ok = true;
        switch (nd.kind) {
            case N_CONST: ok = nd.a < ncst; break;
            case N_ADV: ok = nd.a < naq; break;
            case N_FIX: ok = nd.a < nfq; break;
            case N_INST: ok = nd.a < niq; break;
            case N_CHAL: ok = nd.a < nch; break;
            case N_NEG: ok = nd.a < i; if (ok) nd.a = C.canon[nd.a]; nd.b = 0; break;
            case N_SCALED: ok = nd.a < i && nd.b < ncst; if (ok) nd.a = C.canon[nd.a]; break;
            default: ok = nd.a < i && nd.b < i; if (ok) { nd.a = C.canon[nd.a]; nd.b = C.canon[nd.b]; } break;
        }
        SHP_REQUIRE(ok, SHP_ERR_FORMAT, "circuit description: node refers to a later node, query or constant out of range");
        if (Compiler::leaf(nd.kind)) nd.b = 0;
        C.canon[i] = C.add(nd);
    }
    auto root = [&]() {
        const uint32_t v = r.u32();
        SHP_REQUIRE(v < nn, SHP_ERR_FORMAT, "circuit description: bad root");
        return C.canon[v];
    };
    std::vector<uint32_t> gates(ng);
    for (auto& g : gates) g = root();
    P.num_gates = ng;
    // halo2 ConstraintSystem::degree: the permutation argument's required degree is 3 whether or not any column is
    // equality-enabled (halo2 v0.3.0 permutation::Argument::required_degree), each lookup max(4, 2 + deg(input) +
    // deg(table)), then the gates
    uint32_t degree = 3;
    for (uint32_t g : gates) degree = std::max(degree, C.degree[g]);
    C.compile(gates, P.gate_prog, P.gate_slots);
    P.lookups.resize(nlk);
    for (auto& lk : P.lookups) {
        lk.m = r.u32();
        SHP_REQUIRE(lk.m >= 1 && lk.m <= 64, SHP_ERR_FORMAT, "circuit description: bad lookup width");
        std::vector<uint32_t> in(lk.m), tab(lk.m);
        for (auto& v : in) v = root();
        for (auto& v : tab) v = root();
        uint32_t di = 1, dt = 1;
        for (uint32_t v : in) di = std::max(di, C.degree[v]);
        for (uint32_t v : tab) dt = std::max(dt, C.degree[v]);
        degree = std::max(degree, std::max<uint32_t>(4, 2 + di + dt));
        C.compile(in, lk.in_prog, lk.in_slots);
        C.compile(tab, lk.tab_prog, lk.tab_slots);
    }
    P.degree = degree;
    // halo2 ConstraintSystem::blinding_factors: max(3, most queries to one advice column) + 2
    {
        std::vector<uint32_t> per_col(P.num_advice, 0);
        uint32_t mx = 1;
        for (auto& q : P.adv_q) mx = std::max(mx, ++per_col[q.col]);
        P.bf = std::max<uint32_t>(3, mx) + 2;
    }
    SHP_REQUIRE(n > P.bf + 1, SHP_ERR_FORMAT, "circuit description: k too small for the blinding rows");
    const uint32_t usable = n - (P.bf + 1);
    P.fixed.resize(P.num_fixed);
    for (auto& col : P.fixed) {
        const uint32_t nnz = r.u32();
        r.need((size_t)nnz * 36);
        col.resize(nnz);
        for (auto& e : col) {
            e.first = r.u32();
            SHP_REQUIRE(e.first < usable, SHP_ERR_FORMAT, "circuit description: fixed value outside the usable rows");
            e.second = PlanData::read_fr(d + r.off);
            r.off += 32;
        }
    }
    P.mapping.assign(npc, std::vector<uint32_t>(n));
    if (has_map) {
        for (auto& m : P.mapping) {
            r.need((size_t)n * 4);
            memcpy(m.data(), d + r.off, (size_t)n * 4);
            r.off += (size_t)n * 4;
            for (uint32_t v : m) SHP_REQUIRE(v < npc * n, SHP_ERR_FORMAT, "circuit description: permutation target out of range");
        }
    } else {
        // permutation::keygen::Assembly::copy: cycles joined in the order of the copy constraints, the smaller into the larger
        std::vector<uint32_t> aux((size_t)npc * n), sizes((size_t)npc * n, 1);
        for (uint32_t c = 0; c < npc; c++)
            for (uint32_t row = 0; row < n; row++) { P.mapping[c][row] = c * n + row; aux[(size_t)c * n + row] = c * n + row; }
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> pidx;
        for (uint32_t c = 0; c < npc; c++) pidx[P.perm_cols[c]] = c;
        auto at = [&](uint32_t cell) -> uint32_t& { return P.mapping[cell / n][cell % n]; };
        for (uint32_t i = 0; i < ncp; i++) {
            uint32_t v[6];
            for (auto& x : v) x = r.u32();
            auto ia = pidx.find({v[0], v[1]}), ib = pidx.find({v[3], v[4]});
            SHP_REQUIRE(ia != pidx.end() && ib != pidx.end() && v[2] < usable && v[5] < usable, SHP_ERR_FORMAT,
                        "circuit description: copy constraint on a column without equality or outside the usable rows");
            uint32_t ca = ia->second * n + v[2], cb = ib->second * n + v[5];
            uint32_t la = aux[ca], lb = aux[cb];
            if (la == lb) continue;
            if (sizes[la] < sizes[lb]) { std::swap(ca, cb); std::swap(la, lb); }
            sizes[la] += sizes[lb];
            uint32_t cur = cb;
            do { aux[cur] = la; cur = at(cur); } while (cur != cb);
            std::swap(at(ca), at(cb));
        }
    }
    P.wops.resize(nwo);
    r.need((size_t)nwo * 32);
    if (nwo) memcpy(P.wops.data(), d + r.off, (size_t)nwo * 32);
    r.off += (size_t)nwo * 32;
    P.instance_map.resize(nim);
    for (auto& im : P.instance_map) { im.first = r.u32(); im.second = r.u32(); }
    P.advice_phase.assign(P.num_advice, 0);
    P.wops_end.assign(1, nwo);
    if (h[0] == 2) {
        P.num_phases = r.u32();
        SHP_REQUIRE(P.num_phases >= 1 && P.num_phases <= 8, SHP_ERR_FORMAT, "circuit description: implausible number of phases");
        for (auto& v : P.advice_phase) v = r.u32();
        P.challenge_phase.resize(nch);
        for (auto& v : P.challenge_phase) v = r.u32();
        P.wops_end.resize(P.num_phases);
        for (auto& v : P.wops_end) v = r.u32();
    }
    SHP_REQUIRE(r.off == len, SHP_ERR_FORMAT, "circuit description: trailing bytes");
    out = P.serialize();
    PlanData check;                                   // everything parse() validates (operands, cells, counts) holds for the result
    check.parse(out.data(), out.size());
}

}  // namespace

extern "C" int32_t shp_plan_compile(const uint8_t* description, size_t len, uint8_t* out, size_t out_cap, size_t* out_len) {
    return shp::guarded([&] {
        SHP_REQUIRE(description && out_len, SHP_ERR_INVALID_ARG, "plan_compile: null argument");
        std::vector<uint8_t> plan;
        compile_description(description, len, plan);
        *out_len = plan.size();
        if (!out) return;
        SHP_REQUIRE(out_cap >= plan.size(), SHP_ERR_INVALID_ARG, "plan_compile: buffer too small");
        memcpy(out, plan.data(), plan.size());
    });
}
