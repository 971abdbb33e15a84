"""PLONKish circuit front-end that emits a *circuit plan*: the flat, versioned binary the
MI355X prover (libshielder_hip) and the CPU oracle both consume.

It plays the role of the reference's circuit crates at build time
(`Circuit::configure` / `synthesize` + `generate_keys_with_min_k`, called from
/root/reference/crates/shielder_bindings/build.rs:19-32): the quantities written here are
exactly those the reference's verifier generator reads off a halo2 ConstraintSystem
(/root/reference/crates/halo2-verifier/src/lib/codegen/util.rs:42-131): k, blinding factors,
advice/fixed/instance queries, permutation columns, gate polynomials, lookups, degree.

Plan layout (little endian; parsed by csrc/plan.hpp):
  header (20 x u32) | advice queries | fixed queries | instance queries | permutation columns |
  constants (32 B canonical LE) | gate program | lookup programs | fixed column values (sparse) |
  permutation mapping (u32[n] per permutation column) | witness program | instance map
Version 2 (circuits with several phases: advice columns assigned after a challenge has been drawn, halo2's
`advice_column_in(phase)` / `challenge_usable_after(phase)`, which the reference's verifier generator supports --
codegen/util.rs:70-96, 133-165, templates/Halo2Verifier.sol:267-282) appends
  num_phases | num_challenges | phase of every advice column | phase of every challenge | end of every phase in the
  witness program (u32 each)
and adds the expression leaf OP_CHAL (a = challenge index) and the witness op W_CHAL (imm = challenge index).  A circuit
with one phase and no challenge is written as version 1, byte for byte as before.
"""
import struct

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MAGIC = b"SHPL"
VERSION = 1
VERSION_PHASES = 2

# gate / lookup program opcodes
OP_CONST, OP_ADV, OP_FIX, OP_INST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_OUT, OP_CHAL = range(10)
# witness program opcodes
W_INPUT, W_CONST, W_COPY, W_ADD, W_SUB, W_MUL, W_ADDC, W_MULC, W_POW7, W_BITS, W_SHR, W_CHAL = range(12)
COL_ADVICE, COL_FIXED, COL_INSTANCE = 0, 1, 2


# ------------------------------------------------------------------------------ expressions
class Expr:
    """hash-consed expression node: structurally equal expressions are the same object"""
    __slots__ = ("kind", "args", "degree")
    _intern = {}

    def __new__(cls, kind, args, degree):
        key = (kind,) + tuple(id(a) if isinstance(a, Expr) else a for a in args)
        hit = cls._intern.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.kind, self.args, self.degree = kind, args, degree
        cls._intern[key] = self
        return self

    def __add__(self, o): return Expr("add", (self, _lift(o)), max(self.degree, _lift(o).degree))
    def __radd__(self, o): return _lift(o) + self
    def __sub__(self, o): return Expr("sub", (self, _lift(o)), max(self.degree, _lift(o).degree))
    def __rsub__(self, o): return _lift(o) - self
    def __mul__(self, o): return Expr("mul", (self, _lift(o)), self.degree + _lift(o).degree)
    def __rmul__(self, o): return _lift(o) * self
    def __neg__(self): return Expr("neg", (self,), self.degree)

    def pow(self, e):
        r, b = None, self
        while e:
            if e & 1:
                r = b if r is None else r * b
            e >>= 1
            if e:
                b = b * b
        return r


def _lift(v):
    return v if isinstance(v, Expr) else Const(v)


def Const(v): return Expr("const", (int(v) % R_MOD,), 0)
def Adv(col, rot=0): return Expr("adv", (col, rot), 1)
def Fix(col, rot=0): return Expr("fix", (col, rot), 1)
def Inst(col, rot=0): return Expr("inst", (col, rot), 1)
def Chal(index): return Expr("chal", (index,), 0)          # a challenge is a constant of the proof: degree 0


class _Compiler:
    """hash-consed DAG -> linear three-address program with slot re-use"""

    def __init__(self, plan):
        self.plan = plan

    def compile(self, outputs):
        order, ids = [], {}

        LEAVES = ("const", "adv", "fix", "inst", "chal")

        def visit(e):
            # Leaves are re-materialised at every use instead of being kept alive across gates: a load
            # costs one L2-resident read, a live slot costs 2 KB of LDS per wave -- with every shared
            # column value pinned the withdraw gates needed 30 slots = 60 KB, i.e. two waves per CU.
            k = id(e)
            if e.kind not in LEAVES and k in ids:
                return ids[k]
            kids = [visit(a) for a in e.args if isinstance(a, Expr)]
            nid = len(order)
            if e.kind not in LEAVES:
                ids[k] = nid
            order.append((e, kids))
            return nid

        out_ids = [visit(e) for e in outputs]
        # sequence: nodes in topological order with OUT markers right after the node is available
        # (an OUT consumes its node; keep program order = gate order)
        last_use = {}
        seq = []
        emitted = set()

        def emit(nid):
            if nid in emitted:
                return
            e, kids = order[nid]
            for kd in kids:
                emit(kd)
            emitted.add(nid)
            seq.append(("node", nid))

        for oid in out_ids:
            emit(oid)
            seq.append(("out", oid))
        for pos, (kind, nid) in enumerate(seq):
            if kind == "node":
                for kd in order[nid][1]:
                    last_use[kd] = pos
            else:
                last_use[nid] = pos
        slot_of, free, nslots, prog = {}, [], 0, []
        for pos, (kind, nid) in enumerate(seq):
            if kind == "out":
                prog.append((OP_OUT, 0, slot_of[nid], 0))
            else:
                e, kids = order[nid]
                ksl = [slot_of[kd] for kd in kids]
                # release operands whose last use is this instruction (dst may alias an operand)
                for kd in set(kids):
                    if last_use.get(kd) == pos:
                        free.append(slot_of[kd])
                if free:
                    dst = free.pop()
                else:
                    dst = nslots
                    nslots += 1
                slot_of[nid] = dst
                if e.kind == "const": prog.append((OP_CONST, dst, self.plan.const_index(e.args[0]), 0))
                elif e.kind == "adv": prog.append((OP_ADV, dst, self.plan.adv_query(*e.args), 0))
                elif e.kind == "fix": prog.append((OP_FIX, dst, self.plan.fix_query(*e.args), 0))
                elif e.kind == "inst": prog.append((OP_INST, dst, self.plan.inst_query(*e.args), 0))
                elif e.kind == "chal": prog.append((OP_CHAL, dst, e.args[0], 0))
                elif e.kind == "add": prog.append((OP_ADD, dst, ksl[0], ksl[1]))
                elif e.kind == "sub": prog.append((OP_SUB, dst, ksl[0], ksl[1]))
                elif e.kind == "mul": prog.append((OP_MUL, dst, ksl[0], ksl[1]))
                elif e.kind == "neg": prog.append((OP_NEG, dst, ksl[0], 0))
                else: raise ValueError(e.kind)
            if kind == "out" and last_use.get(nid) == pos:
                free.append(slot_of[nid])
        return prog, max(nslots, 1)


# ------------------------------------------------------------------------------ the plan builder
class PlanBuilder:
    def __init__(self, k):
        self.k, self.n = k, 1 << k
        self.num_advice = self.num_fixed = 0
        self.num_instance = 0                      # number of public inputs (one instance column)
        self.adv_q, self.fix_q, self.inst_q = [], [], []
        self.perm_cols = []                        # (type, index)
        self.consts, self._const_ix = [], {}
        self.gates = []                            # Expr
        self.lookups = []                          # ([input exprs], [table exprs])
        self.fixed_vals = {}                       # (col,row) -> int
        self.copies = []                           # ((type,idx,row),(type,idx,row))
        self.wops = []
        self.n_winputs = 0
        self.n_wscratch = 0
        self.instance_map = []                     # (advice col, row) whose value is public input i
        self.advice_phase = []                     # phase of every advice column
        self.challenge_phase = []                  # challenge i is drawn after the commitments of this phase
        self.wops_end = []                         # witness ops of phase p: [wops_end[p-1], wops_end[p])

    # ---- columns & queries
    def advice_column(self, phase=0):
        self.num_advice += 1
        self.advice_phase.append(phase)
        return self.num_advice - 1

    def challenge(self, phase=0):
        """halo2's challenge_usable_after(phase): drawn from the transcript after the advice commitments of `phase`,
        known to the witness program from phase + 1 on and to every expression"""
        self.challenge_phase.append(phase)
        return len(self.challenge_phase) - 1

    def next_phase(self):
        """the witness ops recorded from here on run after the challenges of the phase that ends here are known"""
        self.wops_end.append(len(self.wops))

    def num_phases(self):
        return max(self.advice_phase + [0]) + 1

    def _phase_tail(self, description=False):
        """what version 2 appends (None for a single-phase circuit without challenges); the description carries the number of
        challenges in its header instead"""
        P = self.num_phases()
        if P == 1 and not self.challenge_phase:
            return None
        assert sorted(set(self.advice_phase)) == list(range(P)), "every phase needs an advice column"
        assert all(ph < P for ph in self.challenge_phase), "challenge after a phase that does not exist"
        assert all(any(ph == p for ph in self.challenge_phase) for p in range(P - 1)), "a phase before the last draws no challenge"
        ends = list(self.wops_end) + [len(self.wops)] * (P - len(self.wops_end))
        assert len(ends) == P and ends == sorted(ends) and ends[-1] == len(self.wops), "next_phase() calls do not match the phases"
        assert all(self.advice_phase[c] == 0 for c, _ in self.instance_map), "public inputs are first-phase cells"
        return struct.pack("<I", P) + (b"" if description else struct.pack("<I", len(self.challenge_phase))) + struct.pack("<%dI" % self.num_advice, *self.advice_phase) + \
            struct.pack("<%dI" % len(self.challenge_phase), *self.challenge_phase) + struct.pack("<%dI" % P, *ends)

    def fixed_column(self):
        self.num_fixed += 1
        return self.num_fixed - 1

    def const_index(self, v):
        v %= R_MOD
        if v not in self._const_ix:
            self._const_ix[v] = len(self.consts)
            self.consts.append(v)
        return self._const_ix[v]

    def _query(self, lst, col, rot):
        if (col, rot) not in lst:
            lst.append((col, rot))
        return lst.index((col, rot))

    def adv_query(self, col, rot): return self._query(self.adv_q, col, rot)
    def fix_query(self, col, rot): return self._query(self.fix_q, col, rot)
    def inst_query(self, col, rot): return self._query(self.inst_q, col, rot)

    def enable_equality(self, ctype, idx):
        if (ctype, idx) not in self.perm_cols:
            self.perm_cols.append((ctype, idx))
            # halo2's enable_equality queries the column at Rotation::cur()
            {COL_ADVICE: self.adv_query, COL_FIXED: self.fix_query, COL_INSTANCE: self.inst_query}[ctype](idx, 0)

    # ---- constraints
    def gate(self, expr): self.gates.append(expr)
    def lookup(self, inputs, tables):
        assert len(inputs) == len(tables) and inputs
        self.lookups.append((list(inputs), list(tables)))

    def fix(self, col, row, value): self.fixed_vals[(col, row)] = int(value) % R_MOD
    def copy(self, a, b): self.copies.append((a, b))

    # ---- witness program (cells are (advice col, row); col == num_advice addresses scratch)
    def w(self, op, dst, a=(0, 0), b=(0, 0), imm=0):
        self.wops.append((op, dst[0], dst[1], a[0], a[1], b[0], b[1], imm))

    def scratch(self, i):
        self.n_wscratch = max(self.n_wscratch, i + 1)
        return ("S", i)

    # ---- derived quantities (halo2 ConstraintSystem::{degree, blinding_factors})
    def degree(self):
        d = 3                 # halo2 v0.3.0 permutation::Argument::required_degree: 3 with or without equality columns
        for ins, tabs in self.lookups:
            di = max([1] + [e.degree for e in ins])
            dt = max([1] + [e.degree for e in tabs])
            d = max(d, max(4, 2 + di + dt))
        for g in self.gates:
            d = max(d, g.degree)
        return d

    def blinding_factors(self):
        per_col = {}
        for c, _ in self.adv_q:
            per_col[c] = per_col.get(c, 0) + 1
        return max(3, max(per_col.values(), default=1)) + 2

    # ---- the circuit description the library compiles itself (shp_plan_compile, csrc/plancompile.hip)
    def build_description(self):
        """What a host holding a halo2 ConstraintSystem would emit: queries, permutation columns, expressions as nodes,
        fixed values, copy constraints, witness program -- no compiled programs, no degree, no mapping."""
        self.build()                                   # registers every query and constant (idempotent)
        nodes, ids = [], {}
        KIND = {"const": 0, "adv": 1, "fix": 2, "inst": 3, "add": 4, "sub": 5, "mul": 6, "neg": 7, "chal": 9}

        def node(e):
            k = id(e)
            if k in ids:
                return ids[k]
            stack = [(e, False)]
            while stack:                               # iterative post-order: children before parents
                cur, done = stack.pop()
                if id(cur) in ids:
                    continue
                kids = [a for a in cur.args if isinstance(a, Expr)]
                if not done and kids:
                    stack.append((cur, True))
                    stack.extend((a, False) for a in reversed(kids))
                    continue
                if cur.kind == "const": rec = (0, self.const_index(cur.args[0]), 0)
                elif cur.kind == "adv": rec = (1, self.adv_query(*cur.args), 0)
                elif cur.kind == "fix": rec = (2, self.fix_query(*cur.args), 0)
                elif cur.kind == "inst": rec = (3, self.inst_query(*cur.args), 0)
                elif cur.kind == "chal": rec = (9, cur.args[0], 0)
                elif cur.kind == "neg": rec = (7, ids[id(kids[0])], 0)
                else: rec = (KIND[cur.kind], ids[id(kids[0])], ids[id(kids[1])])
                ids[id(cur)] = len(nodes)
                nodes.append(rec)
            return ids[k]

        gates = [node(g) for g in self.gates]
        lookups = [([node(e) for e in ins], [node(e) for e in tabs]) for ins, tabs in self.lookups]
        scratch_col = self.num_advice
        cell = lambda c: (scratch_col, c[1]) if c[0] == "S" else c
        wops = []
        for op, dc, dr, ac, ar, bc, br, imm in self.wops:
            d, a, b_ = cell((dc, dr)), cell((ac, ar)), cell((bc, br))
            wops.append((op, d[0], d[1], a[0], a[1], b_[0], b_[1], imm))
        tail = self._phase_tail(description=True)
        hdr = [1 if tail is None else 2, self.k, self.num_advice, self.num_fixed, self.num_instance, len(self.adv_q), len(self.fix_q),
               len(self.inst_q), len(self.perm_cols), len(self.consts), len(nodes), len(gates), len(lookups), len(self.copies), 0,
               self.n_winputs, len(wops), self.n_wscratch, len(self.instance_map), len(self.challenge_phase)]
        out = bytearray(b"SHCD" + struct.pack("<20I", *hdr))
        for lst in (self.adv_q, self.fix_q, self.inst_q):
            for c, r in lst:
                out += struct.pack("<Ii", c, r)
        for t, i in self.perm_cols:
            out += struct.pack("<II", t, i)
        for v in self.consts:
            out += v.to_bytes(32, "little")
        for rec in nodes:
            out += struct.pack("<3I", *rec)
        for g in gates:
            out += struct.pack("<I", g)
        for ins, tabs in lookups:
            out += struct.pack("<I", len(ins))
            for v in ins + tabs:
                out += struct.pack("<I", v)
        per_col = [[] for _ in range(self.num_fixed)]
        for (c, r), v in sorted(self.fixed_vals.items()):
            if v:
                per_col[c].append((r, v))
        for lst in per_col:
            out += struct.pack("<I", len(lst))
            for r, v in lst:
                out += struct.pack("<I", r) + v.to_bytes(32, "little")
        for (ta, ia, ra), (tb, ib, rb) in self.copies:
            out += struct.pack("<6I", ta, ia, ra, tb, ib, rb)
        for ins in wops:
            out += struct.pack("<8I", *ins)
        for c, r in self.instance_map:
            out += struct.pack("<II", c, r)
        return bytes(out) + (tail or b"")

    # ---- serialisation
    def build(self):
        n = self.n
        # compile first: compilation registers the queries
        comp = _Compiler(self)
        gate_prog, gate_slots = comp.compile(self.gates)
        lk = []
        for ins, tabs in self.lookups:
            pi, si = comp.compile(ins)
            pt, st = comp.compile(tabs)
            lk.append((len(ins), pi, si, pt, st))
        for t, i in list(self.perm_cols):
            self.enable_equality(t, i)
        bf, deg = self.blinding_factors(), self.degree()
        usable = n - (bf + 1)
        # permutation mapping from the copy constraints: cycles joined in insertion order
        pidx = {c: i for i, c in enumerate(self.perm_cols)}
        mapping = [[ci * n + r for r in range(n)] for ci in range(len(self.perm_cols))]
        aux = [[ci * n + r for r in range(n)] for ci in range(len(self.perm_cols))]   # cycle representative
        sizes = [[1] * n for _ in self.perm_cols]
        for (ta, ia, ra), (tb, ib, rb) in self.copies:
            ca, cb = pidx[(ta, ia)], pidx[(tb, ib)]
            assert ra < usable and rb < usable, "copy constraint outside the usable rows"
            la, lb = aux[ca][ra], aux[cb][rb]
            if la == lb:
                continue
            # merge the smaller cycle into the larger (halo2 permutation::keygen::Assembly::copy)
            if sizes[la // n][la % n] < sizes[lb // n][lb % n]:
                (ca, ra, la), (cb, rb, lb) = (cb, rb, lb), (ca, ra, la)
            sizes[la // n][la % n] += sizes[lb // n][lb % n]
            cur = cb * n + rb
            start = cur
            while True:
                aux[cur // n][cur % n] = la
                cur = mapping[cur // n][cur % n]
                if cur == start:
                    break
            ma, mb = mapping[ca][ra], mapping[cb][rb]
            mapping[ca][ra], mapping[cb][rb] = mb, ma
        out = bytearray()
        scratch_col = self.num_advice

        def cell(c):
            if c[0] == "S":
                return (scratch_col, c[1])
            return c

        wops = []
        for op, dc, dr, ac, ar, bc, br, imm in self.wops:
            d, a, b = cell((dc, dr)), cell((ac, ar)), cell((bc, br))
            wops.append((op, d[0], d[1], a[0], a[1], b[0], b[1], imm))
        tail = self._phase_tail()
        hdr = [VERSION if tail is None else VERSION_PHASES, self.k, self.num_advice, self.num_fixed, self.num_instance, bf, deg,
               len(self.adv_q), len(self.fix_q), len(self.inst_q), len(self.perm_cols), len(self.consts),
               len(gate_prog), gate_slots, len(self.gates), len(lk),
               self.n_winputs, len(wops), self.n_wscratch, len(self.instance_map)]
        out += MAGIC + struct.pack("<20I", *hdr)
        for lst in (self.adv_q, self.fix_q, self.inst_q):
            for c, r in lst:
                out += struct.pack("<Ii", c, r)
        for t, i in self.perm_cols:
            out += struct.pack("<II", t, i)
        for v in self.consts:
            out += v.to_bytes(32, "little")
        for ins in gate_prog:
            out += struct.pack("<4I", *ins)
        for m, pi, si, pt, st in lk:
            out += struct.pack("<5I", m, len(pi), si, len(pt), st)
            for ins in pi + pt:
                out += struct.pack("<4I", *ins)
        per_col = [[] for _ in range(self.num_fixed)]
        for (c, r), v in sorted(self.fixed_vals.items()):
            assert r < usable, "fixed value outside the usable rows"
            if v:
                per_col[c].append((r, v))
        for lst in per_col:
            out += struct.pack("<I", len(lst))
            for r, v in lst:
                out += struct.pack("<I", r) + v.to_bytes(32, "little")
        import array
        for col in mapping:
            out += array.array("I", col).tobytes()
        for ins in wops:
            out += struct.pack("<8I", *ins)
        for c, r in self.instance_map:
            out += struct.pack("<II", c, r)
        return bytes(out) + (tail or b"")
