"""CPU oracle for the whole proving path (TEST INFRASTRUCTURE, never product code).

A restatement of halo2_proofs v0.3.0 `keygen_pk` / `create_proof` / `verify_proof` with the
KZG-SHPLONK (Bdfg21) multi-open and the EVM Keccak transcript, for a circuit given as a
*circuit plan* (zkos-monorepo_amd/circuits/builder.py).  halo2_proofs is an un-vendored git
dependency of the reference (Cargo.lock:2332-2334, tag v0.3.0); its algorithm is restated from
the published design and anchored on what the reference itself pins:
  proof layout / transcript rules / challenge order   crates/halo2-verifier/src/lib/codegen/util.rs:133-245,
                                                      templates/Halo2Verifier.sol:89-124, 247-307
  quotient identity (gates, permutation, lookups)     codegen/evaluator.rs:45-223, codegen.rs:237-254
  SHPLONK rotation sets and pairing inputs            codegen/pcs.rs:60-104, pcs/bdfg21.rs:21-494
PARITY STATUS: proof *bytes* are unpinned (the reference holds no golden proof and its prover
crates cannot be built here); every primitive underneath is pinned (oracle/oracle.h).

Python only sequences the work; all field/group arithmetic on vectors runs in liboracle.so
(OpenMP), so this is also the timed CPU baseline ("port").
"""
import ctypes as C
import struct

import numpy as np

from . import binding as B

R = B.R_MOD
Q = B.Q_MOD
OP_CONST, OP_ADV, OP_FIX, OP_INST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_OUT, OP_CHAL = range(10)
W_INPUT, W_CONST, W_COPY, W_ADD, W_SUB, W_MUL, W_ADDC, W_MULC, W_POW7, W_BITS, W_SHR, W_CHAL = range(12)
COL_ADVICE, COL_FIXED, COL_INSTANCE = 0, 1, 2


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fe(x):
    """python int -> (4,) Montgomery limbs"""
    return B.fr_array([x])[0]


def ints(a):
    return B.fr_ints(a)


def _v2(fn, a, b):
    out = np.empty_like(a)
    getattr(B.lib(), fn)(_p(out), _p(a), _p(b), C.c_size_t(a.shape[0]))
    return out


def vadd(a, b): return _v2("orc_vec_add", a, b)
def vsub(a, b): return _v2("orc_vec_sub", a, b)
def vmul(a, b): return _v2("orc_vec_mul", a, b)


def vneg(a):
    out = np.empty_like(a)
    B.lib().orc_vec_neg(_p(out), _p(a), C.c_size_t(a.shape[0]))
    return out


def vscale(a, c):
    out = np.empty_like(a)
    B.lib().orc_vec_scale(_p(out), _p(a), _p(np.ascontiguousarray(c)), C.c_size_t(a.shape[0]))
    return out


def vaddc(a, c):
    out = np.empty_like(a)
    B.lib().orc_vec_addc(_p(out), _p(a), _p(np.ascontiguousarray(c)), C.c_size_t(a.shape[0]))
    return out


def vfill(n, c):
    out = np.empty((n, 4), dtype=np.uint64)
    B.lib().orc_vec_fill(_p(out), _p(np.ascontiguousarray(c)), C.c_size_t(n))
    return out


def vfold(acc, y, e):
    out = np.empty_like(acc)
    B.lib().orc_vec_fold(_p(out), _p(acc), _p(np.ascontiguousarray(y)), _p(e), C.c_size_t(acc.shape[0]))
    return out


def powers(first, base, n):
    out = np.empty((n, 4), dtype=np.uint64)
    B.lib().orc_powers(_p(out), _p(fe(first)), _p(fe(base)), C.c_size_t(n))
    return out


def draws(seed, counter0, count):
    out = np.empty((count, 4), dtype=np.uint64)
    B.lib().orc_fr_draw_many(C.c_char_p(seed), C.c_uint64(counter0), C.c_size_t(count), _p(out))
    return out


def kate_div(coeff, a):
    n = coeff.shape[0]
    out = np.zeros((n, 4), dtype=np.uint64)
    B.lib().orc_kate_div(_p(np.ascontiguousarray(coeff)), C.c_size_t(n), _p(fe(a)), _p(out))
    return out          # degree n-2, top coefficient slot left zero


def g1_be64(pt):
    (x, y), = B.g1_affine_ints(np.asarray(pt).reshape(1, 8))
    return x.to_bytes(32, "big") + y.to_bytes(32, "big")


# ------------------------------------------------------------------------------ plan
class Plan:
    def __init__(self, data: bytes):
        assert data[:4] == b"SHPL", "not a circuit plan"
        h = struct.unpack_from("<20I", data, 4)
        assert h[0] in (1, 2)
        (_, self.k, self.num_advice, self.num_fixed, self.num_instance, self.bf, self.degree, naq, nfq, niq, npc,
         ncst, ngi, self.gate_slots, self.num_gates, nlk, self.n_winputs, nwo, self.n_wscratch, nim) = h
        self.n = 1 << self.k
        self.data = data
        off = 84

        def pairs(cnt, fmt):
            nonlocal off
            out = [struct.unpack_from(fmt, data, off + 8 * i) for i in range(cnt)]
            off += 8 * cnt
            return out

        self.adv_q, self.fix_q, self.inst_q = pairs(naq, "<Ii"), pairs(nfq, "<Ii"), pairs(niq, "<Ii")
        self.perm_cols = pairs(npc, "<II")
        self.consts = [int.from_bytes(data[off + 32 * i: off + 32 * i + 32], "little") for i in range(ncst)]
        off += 32 * ncst

        def prog(cnt):
            nonlocal off
            out = [struct.unpack_from("<4I", data, off + 16 * i) for i in range(cnt)]
            off += 16 * cnt
            return out

        self.gate_prog = prog(ngi)
        self.lookups = []
        for _ in range(nlk):
            m, ni, si, nt, st = struct.unpack_from("<5I", data, off)
            off += 20
            self.lookups.append((m, prog(ni), si, prog(nt), st))
        self.fixed = []
        for _ in range(self.num_fixed):
            (nnz,) = struct.unpack_from("<I", data, off)
            off += 4
            col = {}
            for i in range(nnz):
                (r,) = struct.unpack_from("<I", data, off)
                col[r] = int.from_bytes(data[off + 4: off + 36], "little")
                off += 36
            self.fixed.append(col)
        self.mapping = []
        for _ in range(npc):
            self.mapping.append(np.frombuffer(data, dtype=np.uint32, count=self.n, offset=off).copy())
            off += 4 * self.n
        self.wops = [struct.unpack_from("<8I", data, off + 32 * i) for i in range(nwo)]
        off += 32 * nwo
        self.instance_map = pairs(nim, "<II")
        # version 2: phases (halo2's advice_column_in / challenge_usable_after; codegen/util.rs:70-96)
        self.num_phases, self.advice_phase, self.challenge_phase, self.wops_end = 1, [0] * self.num_advice, [], [nwo]
        if h[0] == 2:
            self.num_phases, nch = struct.unpack_from("<2I", data, off)
            off += 8
            self.advice_phase = list(struct.unpack_from("<%dI" % self.num_advice, data, off))
            off += 4 * self.num_advice
            self.challenge_phase = list(struct.unpack_from("<%dI" % nch, data, off))
            off += 4 * nch
            self.wops_end = list(struct.unpack_from("<%dI" % self.num_phases, data, off))
            off += 4 * self.num_phases
        assert off == len(data), "trailing bytes in plan"
        self.usable = self.n - (self.bf + 1)
        self.ext_k = self.k
        while (1 << self.ext_k) < self.n * (self.degree - 1):
            self.ext_k += 1
        self.chunk = self.degree - 2
        self.num_sets = (len(self.perm_cols) + self.chunk - 1) // self.chunk if self.perm_cols else 0
        self.rot_last = -(self.bf + 1)


# ------------------------------------------------------------------------------ witness VM
def new_grid(plan: Plan):
    return [[0] * plan.n for _ in range(plan.num_advice)] + [[0] * max(plan.n_wscratch, 1)]


def synthesize(plan: Plan, inputs, challenges=(), grid=None, phase=None):
    """P1: run the plan's witness program over canonical ints -> advice[col][row], instances.
    phase: run only that phase's part of the program, on `grid` (the state the earlier phases left), with the challenges
    drawn so far; None = the whole program (a single-phase plan, or all challenges known up front)"""
    assert len(inputs) == plan.n_winputs
    A = plan.num_advice
    grid = new_grid(plan) if grid is None else grid
    cst = plan.consts
    lo = 0 if phase in (None, 0) else plan.wops_end[phase - 1]
    hi = len(plan.wops) if phase is None else plan.wops_end[phase]
    for op, dc, dr, ac, ar, bc, br, imm in plan.wops[lo:hi]:
        if op == W_INPUT: v = inputs[imm] % R
        elif op == W_CHAL: v = challenges[imm]
        elif op == W_CONST: v = cst[imm]
        elif op == W_COPY: v = grid[ac][ar]
        elif op == W_ADD: v = (grid[ac][ar] + grid[bc][br]) % R
        elif op == W_SUB: v = (grid[ac][ar] - grid[bc][br]) % R
        elif op == W_MUL: v = (grid[ac][ar] * grid[bc][br]) % R
        elif op == W_ADDC: v = (grid[ac][ar] + cst[imm]) % R
        elif op == W_MULC: v = (grid[ac][ar] * cst[imm]) % R
        elif op == W_POW7: v = pow(grid[ac][ar], 7, R)
        elif op == W_BITS: v = grid[ac][ar] & ((1 << imm) - 1)
        elif op == W_SHR: v = grid[ac][ar] >> imm
        else: raise ValueError(op)
        grid[dc][dr] = v
    advice = grid[:A]
    instances = [advice[c][r] for c, r in plan.instance_map]
    return advice, instances


# ------------------------------------------------------------------------------ programs
def run_program_vec(prog, nslots, consts, adv, fix, inst, size, chal=()):
    """vectorised interpreter: adv/fix/inst are callables query_index -> (size,4) array; yields OUT values"""
    slots = [None] * nslots
    for op, dst, a, b in prog:
        if op == OP_CONST: slots[dst] = vfill(size, fe(consts[a]))
        elif op == OP_CHAL: slots[dst] = vfill(size, fe(chal[a]))
        elif op == OP_ADV: slots[dst] = adv(a)
        elif op == OP_FIX: slots[dst] = fix(a)
        elif op == OP_INST: slots[dst] = inst(a)
        elif op == OP_ADD: slots[dst] = vadd(slots[a], slots[b])
        elif op == OP_SUB: slots[dst] = vsub(slots[a], slots[b])
        elif op == OP_MUL: slots[dst] = vmul(slots[a], slots[b])
        elif op == OP_NEG: slots[dst] = vneg(slots[a])
        elif op == OP_OUT: yield slots[a]
        else: raise ValueError(op)


def run_program_scalar(prog, nslots, consts, adv, fix, inst, chal=()):
    slots = [0] * nslots
    for op, dst, a, b in prog:
        if op == OP_CONST: slots[dst] = consts[a]
        elif op == OP_CHAL: slots[dst] = chal[a]
        elif op == OP_ADV: slots[dst] = adv(a)
        elif op == OP_FIX: slots[dst] = fix(a)
        elif op == OP_INST: slots[dst] = inst(a)
        elif op == OP_ADD: slots[dst] = (slots[a] + slots[b]) % R
        elif op == OP_SUB: slots[dst] = (slots[a] - slots[b]) % R
        elif op == OP_MUL: slots[dst] = (slots[a] * slots[b]) % R
        elif op == OP_NEG: slots[dst] = (-slots[a]) % R
        elif op == OP_OUT: yield slots[a]
        else: raise ValueError(op)


def mock_check(plan: Plan, advice, instances, challenges=()):
    """MockProver-style row check of gates on the usable rows (debug aid; python ints)"""
    n = plan.n
    fixed = [[c.get(r, 0) for r in range(n)] for c in plan.fixed]
    inst = instances + [0] * (n - len(instances))
    bad = []
    for row in range(plan.usable):
        outs = list(run_program_scalar(
            plan.gate_prog, plan.gate_slots, plan.consts,
            lambda q: advice[plan.adv_q[q][0]][(row + plan.adv_q[q][1]) % n],
            lambda q: fixed[plan.fix_q[q][0]][(row + plan.fix_q[q][1]) % n],
            lambda q: inst[(row + plan.inst_q[q][1]) % n], challenges))
        for gi, v in enumerate(outs):
            if v:
                bad.append((row, gi))
    # copy constraints
    cols = []
    for t, i in plan.perm_cols:
        cols.append(advice[i] if t == COL_ADVICE else fixed[i] if t == COL_FIXED else inst)
    for ci, m in enumerate(plan.mapping):
        for r in range(plan.usable):
            tgt = int(m[r])
            if cols[ci][r] != cols[tgt // n][tgt % n]:
                bad.append(("copy", ci, r))
    return bad


# ------------------------------------------------------------------------------ transcript
class PointAtInfinity(ValueError):
    """create_proof's failure when a commitment is the identity (see Transcript.write_point)"""


class Transcript:
    """EVM Keccak transcript: Halo2Verifier.sol:105-124, 247-307"""

    def __init__(self, vk_digest: bytes):
        self.buf = bytearray(vk_digest)
        self.proof = bytearray()

    def common_scalar(self, x):
        self.buf += (x % R).to_bytes(32, "big")

    def write_scalar(self, x):
        self.common_scalar(x)
        self.proof += (x % R).to_bytes(32, "big")

    def common_point(self, be64):
        self.buf += be64

    def write_point(self, pt):
        b = g1_be64(pt)
        # halo2's transcripts refuse the identity ("cannot write points at infinity to the transcript", an
        # Error::Transcript out of create_proof), and the reference's verifier refuses (0, 0) as a proof point
        # (/root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:89-99, read_ec_point: y^2 = x^3 + 3 fails): a circuit whose
        # commitment is the identity -- a quotient piece that is identically zero, say -- has no proof
        if b == bytes(64):
            raise PointAtInfinity("a commitment is the point at infinity: the transcript refuses it")
        self.buf += b
        self.proof += b

    def squeeze(self):
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b"")
        h = B.keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R


# ------------------------------------------------------------------------------ keygen
class ProvingKey:
    pass


def _lagrange_column(n, sparse):
    col = np.zeros((n, 4), dtype=np.uint64)
    if sparse:
        rows = sorted(sparse)
        col[rows] = B.fr_array([sparse[r] for r in rows])
    return col


def keygen(plan: Plan, srs, transcript_repr=None) -> ProvingKey:
    """keygen_vk + keygen_pk: fixed/sigma polys, cosets, commitments, l0/l_last/l_active.
    transcript_repr (int): halo2's vk.transcript_repr(), absorbed first by the verifier (Halo2Verifier.sol:247-248), for
    callers that hold the halo2 VerifyingKey; None = the build's own digest over plan and commitments"""
    assert srs.k == plan.k
    n, k, ek = plan.n, plan.k, plan.ext_k
    pk = ProvingKey()
    pk.plan = plan
    pk.omega = ints(B.root_of_unity(k))[0]
    pk.omega_ext = ints(B.root_of_unity(ek))[0]
    pk.delta = ints(B.fr_delta())[0]
    pk.zeta = ints(B.fr_zeta())[0]
    pk.fixed_values = [_lagrange_column(n, c) for c in plan.fixed]
    pk.fixed_polys = [B.ntt(v, k, inverse=True) for v in pk.fixed_values]
    pk.fixed_cosets = [B.coeff_to_extended(p, k, ek) for p in pk.fixed_polys]
    # sigma_c(omega^r) = delta^{c'} * omega^{r'} for mapping (c,r) -> (c',r')
    om = powers(1, pk.omega, n)
    dpow = [pow(pk.delta, c, R) for c in range(len(plan.perm_cols))]
    pk.sigma_values = []
    for m in plan.mapping:
        tc, tr = m // n, m % n
        vals = om[tr]
        out = np.empty_like(vals)
        for c in np.unique(tc):
            sel = np.nonzero(tc == c)[0]
            out[sel] = vscale(np.ascontiguousarray(vals[sel]), fe(dpow[int(c)]))
        pk.sigma_values.append(out)
    pk.sigma_polys = [B.ntt(v, k, inverse=True) for v in pk.sigma_values]
    pk.sigma_cosets = [B.coeff_to_extended(p, k, ek) for p in pk.sigma_polys]
    one = fe(1)
    l0 = np.zeros((n, 4), dtype=np.uint64); l0[0] = one
    llast = np.zeros((n, 4), dtype=np.uint64); llast[plan.usable] = one
    lblind = np.zeros((n, 4), dtype=np.uint64); lblind[plan.usable + 1:] = one
    lact = vsub(vsub(vfill(n, one), llast), lblind)
    ext = lambda v: B.coeff_to_extended(B.ntt(v, k, inverse=True), k, ek)
    pk.l0, pk.l_last, pk.l_active = ext(l0), ext(llast), ext(lact)
    pk.fixed_commitments = [B.msm(srs.g_lagrange, v) for v in pk.fixed_values]
    pk.sigma_commitments = [B.msm(srs.g_lagrange, v) for v in pk.sigma_values]
    body = b"shielder-hip-vk-v1" + B.keccak256(plan.data)
    for c in pk.fixed_commitments + pk.sigma_commitments:
        body += g1_be64(c) if c.any() else bytes(64)
    pk.vk_digest = (int.from_bytes(B.keccak256(body), "big") % R).to_bytes(32, "big")
    if transcript_repr is not None:
        assert 0 <= transcript_repr < R
        pk.vk_digest = int(transcript_repr).to_bytes(32, "big")
    # (zeta * omega_ext^i)^n - 1 has period 2^(ek-k); inverses
    per = 1 << (ek - k)
    zn = pow(pk.zeta, n, R)
    wn = pow(pk.omega_ext, n, R)
    pk.t_inv = B.fr_array([pow((zn * pow(wn, i, R) - 1) % R, -1, R) for i in range(per)])
    pk.x_ext = powers(pk.zeta, pk.omega_ext, 1 << ek)      # X on the extended coset
    return pk


# ------------------------------------------------------------------------------ lookup permutation
def permute_expression_pair(plan, inp, tab):
    """halo2 lookup::prover::permute_expression_pair on the usable rows (python ints)"""
    u = plan.usable
    a = sorted(inp[:u])
    left = {}
    for v in tab[:u]:
        left[v] = left.get(v, 0) + 1
    out_t = [None] * u
    repeated = []
    for row in range(u):
        if row == 0 or a[row] != a[row - 1]:
            out_t[row] = a[row]
            if left.get(a[row], 0) == 0:
                raise ValueError("lookup input not in table")
            left[a[row]] -= 1
        else:
            repeated.append(row)
    for v in sorted(left):
        for _ in range(left[v]):
            out_t[repeated.pop()] = v
    assert not repeated
    return a, out_t


# ------------------------------------------------------------------------------ SHPLONK helpers
def build_queries(plan, names):
    """the verifier's query order (pcs.rs:60-104) as (poly key, rotation) pairs"""
    qs = []
    for c, r in plan.adv_q:
        qs.append((("adv", c), r))
    for s in range(plan.num_sets):
        qs += [(("pz", s), 0), (("pz", s), 1)]
    for s in reversed(range(plan.num_sets - 1)):
        qs.append((("pz", s), plan.rot_last))
    for l in range(len(plan.lookups)):
        qs += [(("lz", l), 0), (("lpi", l), 0), (("lpt", l), 0), (("lpi", l), -1), (("lz", l), 1)]
    for c, r in plan.fix_q:
        qs.append((("fix", c), r))
    for i in range(len(plan.perm_cols)):
        qs.append((("sig", i), 0))
    qs += [(("h",), 0), (("rand",), 0)]
    del names
    return qs


def rotation_sets(queries):
    """bdfg21.rs:443-494: group polys by identical rotation set, first-seen order"""
    superset, polys = set(), []
    for key, rot in queries:
        superset.add(rot)
        for ent in polys:
            if ent[0] == key:
                assert rot not in ent[1]
                ent[1].append(rot)
                break
        else:
            polys.append((key, [rot]))
    sets = []
    for key, rots in polys:
        rs = tuple(sorted(rots))
        for st in sets:
            if st[0] == rs:
                if key not in st[1]:
                    st[1].append(key)
                break
        else:
            sets.append((rs, [key]))
    return sorted(superset), sets


def lagrange_interp(xs, ys):
    """coefficients (low to high) of the polynomial through (xs, ys), python ints"""
    m = len(xs)
    coeffs = [0] * m
    for i in range(m):
        num = [1]
        den = 1
        for j in range(m):
            if i == j:
                continue
            num = [(-xs[j] * num[0]) % R] + [((num[t - 1] if t > 0 else 0) - xs[j] * (num[t] if t < len(num) else 0)) % R
                                              for t in range(1, len(num) + 1)]
            den = den * (xs[i] - xs[j]) % R
        sc = ys[i] * pow(den, -1, R) % R
        for t in range(m):
            coeffs[t] = (coeffs[t] + sc * num[t]) % R
    return coeffs


# ------------------------------------------------------------------------------ prover
def prove(pk: ProvingKey, srs, inputs, seed: bytes, return_trace=False, randomness=None, assigned=None):
    """create_proof: phases P0..P13 (witness, advice, lookups, grand products, quotient, evaluations, SHPLONK). `seed` keys the
    ChaCha20 blinding stream; draws are consumed in the order include/shielder_hip.h gives at shp_prove_with_randomness.
    randomness: (count, 4) Montgomery array of pre-drawn Fr::random values used INSTEAD of the seed's stream (the
    reference's `rng` handed over as its draws); assigned: (advice_ints[col][row], instances) from the caller's own
    synthesis instead of the plan's witness program; for a plan with several phases a callable
    assigned(phase, challenges) -> (advice_ints[col][row], instances) asked once per phase (the columns of that phase are
    read from its answer)."""
    plan = pk.plan
    n, k, ek, bf, u = plan.n, plan.k, plan.ext_k, plan.bf, plan.usable
    A = plan.num_advice
    ne = 1 << ek
    estep = 1 << (ek - k)
    ctr = [0]

    def draw(count):
        if randomness is not None:
            out = np.ascontiguousarray(randomness[ctr[0]:ctr[0] + count])
            assert out.shape[0] == count, "randomness buffer too short"
        else:
            out = draws(seed, ctr[0], count)
        ctr[0] += count
        return out

    trace = {}
    # P0-P2, once per phase (halo2 create_proof: synthesize with the challenges known so far, blind and commit the columns
    # of this phase, then squeeze the challenges usable after it; proof layout Halo2Verifier.sol:267-282)
    advice = [None] * A
    chal = [None] * len(plan.challenge_phase)
    grid = new_grid(plan)
    tr = None
    for ph in range(plan.num_phases):
        if assigned is None:
            advice_ints, inst_now = synthesize(plan, inputs, chal, grid, ph)
        else:
            advice_ints, inst_now = assigned(ph, list(chal)) if callable(assigned) else assigned
        if ph == 0:
            instances = list(inst_now)
            tr = Transcript(pk.vk_digest)
            for v in instances:
                tr.common_scalar(v)
        cols = [c for c in range(A) if plan.advice_phase[c] == ph]
        # P1: blinding rows, then one (unused) blind per column
        for c in cols:
            col = B.fr_array(advice_ints[c])
            col[u:] = draw(bf + 1)
            advice[c] = col
        draw(len(cols))
        # P2
        for c in cols:
            tr.write_point(B.msm(srs.g_lagrange, advice[c]))
        for i, cp in enumerate(plan.challenge_phase):
            if cp == ph:
                chal[i] = tr.squeeze()
    inst_col = _lagrange_column(n, {i: v for i, v in enumerate(instances)})
    fixed = pk.fixed_values
    L = len(plan.lookups)
    theta = tr.squeeze() if L else None
    if not L:
        theta = tr.squeeze()

    def lag_adv(q):
        c, r = plan.adv_q[q]
        return np.roll(advice[c], -r, axis=0)

    def lag_fix(q):
        c, r = plan.fix_q[q]
        return np.roll(fixed[c], -r, axis=0)

    def lag_inst(q):
        return np.roll(inst_col, -plan.inst_q[q][1], axis=0)

    # P3 lookups: compress, permute, commit
    lk = []
    for (m, pi, si, pt, st) in plan.lookups:
        def compress(prog, ns):
            acc = None
            for e in run_program_vec(prog, ns, plan.consts, lag_adv, lag_fix, lag_inst, n, chal):
                acc = e if acc is None else vadd(vscale(acc, fe(theta)), e)
            return acc
        cin, ctab = compress(pi, si), compress(pt, st)
        pin, ptab = permute_expression_pair(plan, ints(cin), ints(ctab))
        pin_v = np.zeros((n, 4), dtype=np.uint64); pin_v[:u] = B.fr_array(pin)
        ptab_v = np.zeros((n, 4), dtype=np.uint64); ptab_v[:u] = B.fr_array(ptab)
        pin_v[u:] = draw(bf + 1)
        ptab_v[u:] = draw(bf + 1)
        draw(2)
        tr.write_point(B.msm(srs.g_lagrange, pin_v))
        tr.write_point(B.msm(srs.g_lagrange, ptab_v))
        lk.append({"cin": cin, "ctab": ctab, "pin": pin_v, "ptab": ptab_v})
    # P4
    beta = tr.squeeze()
    gamma = tr.squeeze()
    fb, fg = fe(beta), fe(gamma)
    # P5 permutation grand products
    pcols = []
    for t, i in plan.perm_cols:
        pcols.append(advice[i] if t == COL_ADVICE else fixed[i] if t == COL_FIXED else inst_col)
    om = powers(1, pk.omega, n)
    pz = []
    last_z = fe(1)
    for s in range(plan.num_sets):
        cols = list(range(s * plan.chunk, min((s + 1) * plan.chunk, len(pcols))))
        den = vfill(n, fe(1))
        num = vfill(n, fe(1))
        for ci in cols:
            den = vmul(den, vaddc(vadd(vscale(pk.sigma_values[ci], fb), pcols[ci]), fg))
            dco = pow(pk.delta, ci, R) * beta % R
            num = vmul(num, vaddc(vadd(vscale(om, fe(dco)), pcols[ci]), fg))
        mod = vmul(num, B.batch_invert(den))
        z = np.empty((n, 4), dtype=np.uint64)
        B.lib().orc_running_product(_p(z), _p(np.ascontiguousarray(last_z)), _p(mod), C.c_size_t(n))
        z[n - bf:] = draw(bf)
        last_z = z[u].copy()
        draw(1)
        pz.append(z)
        tr.write_point(B.msm(srs.g_lagrange, z))
    # P6 lookup grand products
    for d in lk:
        den = vmul(vaddc(d["pin"], fb), vaddc(d["ptab"], fg))
        num = vmul(vaddc(d["cin"], fb), vaddc(d["ctab"], fg))
        mod = vmul(num, B.batch_invert(den))
        z = np.empty((n, 4), dtype=np.uint64)
        B.lib().orc_running_product(_p(z), _p(fe(1)), _p(mod), C.c_size_t(n))
        z[n - bf:] = draw(bf)
        draw(1)
        d["z"] = z
        tr.write_point(B.msm(srs.g_lagrange, z))
    # P7 vanishing: random polynomial (coefficient basis) + blind
    rand_poly = draw(n)
    draw(1)
    tr.write_point(B.msm(srs.g, rand_poly))
    # P8
    y = tr.squeeze()
    fy = fe(y)
    # P9 coefficient forms, P10 quotient
    to_coeff = lambda v: B.ntt(v, k, inverse=True)
    to_ext = lambda p: B.coeff_to_extended(p, k, ek)
    adv_poly = [to_coeff(v) for v in advice]
    inst_poly = to_coeff(inst_col)
    adv_cos = [to_ext(p) for p in adv_poly]
    inst_cos = to_ext(inst_poly)
    pz_poly = [to_coeff(z) for z in pz]
    pz_cos = [to_ext(p) for p in pz_poly]
    for d in lk:
        for nm in ("pin", "ptab", "z"):
            d[nm + "_poly"] = to_coeff(d[nm])
            d[nm + "_cos"] = to_ext(d[nm + "_poly"])

    def rot(arr, r):
        return np.roll(arr, -r * estep, axis=0)

    def ext_adv(q):
        c, r = plan.adv_q[q]
        return rot(adv_cos[c], r)

    def ext_fix(q):
        c, r = plan.fix_q[q]
        return rot(pk.fixed_cosets[c], r)

    def ext_inst(q):
        return rot(inst_cos, plan.inst_q[q][1])

    h = np.zeros((ne, 4), dtype=np.uint64)
    fold = lambda acc, e: vfold(acc, fy, e)
    for e in run_program_vec(plan.gate_prog, plan.gate_slots, plan.consts, ext_adv, ext_fix, ext_inst, ne, chal):
        h = fold(h, e)
    one_e = vfill(ne, fe(1))
    if plan.num_sets:
        ccos = []
        for t, i in plan.perm_cols:
            ccos.append(adv_cos[i] if t == COL_ADVICE else pk.fixed_cosets[i] if t == COL_FIXED else inst_cos)
        h = fold(h, vmul(pk.l0, vsub(one_e, pz_cos[0])))
        zl = pz_cos[-1]
        h = fold(h, vmul(pk.l_last, vsub(vmul(zl, zl), zl)))
        for s in range(1, plan.num_sets):
            h = fold(h, vmul(pk.l0, vsub(pz_cos[s], rot(pz_cos[s - 1], plan.rot_last))))
        for s in range(plan.num_sets):
            cols = list(range(s * plan.chunk, min((s + 1) * plan.chunk, len(ccos))))
            left = rot(pz_cos[s], 1)
            right = pz_cos[s]
            for ci in cols:
                left = vmul(left, vaddc(vadd(vscale(pk.sigma_cosets[ci], fb), ccos[ci]), fg))
                dco = pow(pk.delta, ci, R) * beta % R
                right = vmul(right, vaddc(vadd(vscale(pk.x_ext, fe(dco)), ccos[ci]), fg))
            h = fold(h, vmul(pk.l_active, vsub(left, right)))
    for (m, pi, si, pt, st), d in zip(plan.lookups, lk):
        def compress_ext(prog, ns):
            acc = None
            for e in run_program_vec(prog, ns, plan.consts, ext_adv, ext_fix, ext_inst, ne, chal):
                acc = e if acc is None else vadd(vscale(acc, fe(theta)), e)
            return acc
        Ain, Stab = compress_ext(pi, si), compress_ext(pt, st)
        z, a_, s_ = d["z_cos"], d["pin_cos"], d["ptab_cos"]
        h = fold(h, vmul(pk.l0, vsub(one_e, z)))
        h = fold(h, vmul(pk.l_last, vsub(vmul(z, z), z)))
        lhs = vmul(rot(z, 1), vmul(vaddc(a_, fb), vaddc(s_, fg)))
        rhs = vmul(z, vmul(vaddc(Ain, fb), vaddc(Stab, fg)))
        h = fold(h, vmul(pk.l_active, vsub(lhs, rhs)))
        ams = vsub(a_, s_)
        h = fold(h, vmul(pk.l0, ams))
        h = fold(h, vmul(pk.l_active, vmul(ams, vsub(a_, rot(a_, -1)))))
    hq = np.empty_like(h)
    B.lib().orc_vec_mul_periodic(_p(hq), _p(h), _p(pk.t_inv), C.c_size_t(pk.t_inv.shape[0]), C.c_size_t(ne))
    hcoef = B.extended_to_coeff(hq, ek)
    npieces = plan.degree - 1
    assert not hcoef[n * npieces:].any(), "quotient has too high a degree: witness does not satisfy the circuit"
    pieces = [np.ascontiguousarray(hcoef[i * n:(i + 1) * n]) for i in range(npieces)]
    draw(npieces)
    for p_ in pieces:
        tr.write_point(B.msm(srs.g, p_))
    # P11
    x = tr.squeeze()
    xr = lambda r: x * pow(pk.omega, r % n if r >= 0 else (n + r) % n, R) % R
    ev = lambda poly, r: ints(B.eval_poly(poly, fe(xr(r))))[0]
    evals = {}
    for c, r in plan.adv_q:
        evals[(("adv", c), r)] = ev(adv_poly[c], r)
        tr.write_scalar(evals[(("adv", c), r)])
    for c, r in plan.fix_q:
        evals[(("fix", c), r)] = ev(pk.fixed_polys[c], r)
        tr.write_scalar(evals[(("fix", c), r)])
    evals[(("rand",), 0)] = ev(rand_poly, 0)
    tr.write_scalar(evals[(("rand",), 0)])
    for i in range(len(plan.perm_cols)):
        evals[(("sig", i), 0)] = ev(pk.sigma_polys[i], 0)
        tr.write_scalar(evals[(("sig", i), 0)])
    for s in range(plan.num_sets):
        for r in (0, 1) + ((plan.rot_last,) if s != plan.num_sets - 1 else ()):
            evals[(("pz", s), r)] = ev(pz_poly[s], r)
            tr.write_scalar(evals[(("pz", s), r)])
    for l, d in enumerate(lk):
        for key, r in ((("lz", l), 0), (("lz", l), 1), (("lpi", l), 0), (("lpi", l), -1), (("lpt", l), 0)):
            poly = {"lz": d["z_poly"], "lpi": d["pin_poly"], "lpt": d["ptab_poly"]}[key[0]]
            evals[(key, r)] = ev(poly, r)
            tr.write_scalar(evals[(key, r)])
    # h(X) = sum_i x^(n i) h_i(X): the single quotient polynomial opened at x
    xn = pow(x, n, R)
    hx = np.zeros((n, 4), dtype=np.uint64)
    for p_ in reversed(pieces):
        hx = vadd(vscale(hx, fe(xn)), p_)
    evals[(("h",), 0)] = ints(B.eval_poly(hx, fe(x)))[0]
    polys = {("h",): hx, ("rand",): rand_poly}
    for c in range(A): polys[("adv", c)] = adv_poly[c]
    for c in range(plan.num_fixed): polys[("fix", c)] = pk.fixed_polys[c]
    for i in range(len(plan.perm_cols)): polys[("sig", i)] = pk.sigma_polys[i]
    for s in range(plan.num_sets): polys[("pz", s)] = pz_poly[s]
    for l, d in enumerate(lk):
        polys[("lz", l)], polys[("lpi", l)], polys[("lpt", l)] = d["z_poly"], d["pin_poly"], d["ptab_poly"]
    # P12-P13 SHPLONK
    superset, sets = rotation_sets(build_queries(plan, None))
    zeta_c = tr.squeeze()
    nu = tr.squeeze()
    point = {r: xr(r) for r in superset}
    fsets = []
    hacc = np.zeros((n, 4), dtype=np.uint64)
    nu_pow = 1
    for rots, keys in sets:
        f = np.zeros((n, 4), dtype=np.uint64)
        comb = {r: 0 for r in rots}
        zp = 1
        for key in keys:
            f = vadd(f, vscale(polys[key], fe(zp)))
            for r in rots:
                comb[r] = (comb[r] + zp * evals[(key, r)]) % R
            zp = zp * zeta_c % R
        rcoef = lagrange_interp([point[r] for r in rots], [comb[r] for r in rots])
        num = f.copy()
        num[:len(rcoef)] = vsub(np.ascontiguousarray(f[:len(rcoef)]), B.fr_array(rcoef))
        for r in rots:
            num = kate_div(num, point[r])
        hacc = vadd(hacc, vscale(num, fe(nu_pow)))
        fsets.append((rots, f, rcoef))
        nu_pow = nu_pow * nu % R
    tr.write_point(B.msm(srs.g, hacc))
    mu = tr.squeeze()
    zt = 1
    for r in superset:
        zt = zt * (mu - point[r]) % R
    Lp = np.zeros((n, 4), dtype=np.uint64)
    nu_pow = 1
    diff0 = None
    for rots, f, rcoef in fsets:
        diff = 1
        for r in superset:
            if r not in rots:
                diff = diff * (mu - point[r]) % R
        if diff0 is None:
            diff0 = diff
        r_mu = sum(c * pow(mu, i, R) for i, c in enumerate(rcoef)) % R
        g = f.copy()
        g[0] = fe((ints(f[0:1])[0] - r_mu) % R)
        Lp = vadd(Lp, vscale(g, fe(nu_pow * diff % R)))
        nu_pow = nu_pow * nu % R
    Lp = vsub(Lp, vscale(hacc, fe(zt)))
    Lp = vscale(Lp, fe(pow(diff0, -1, R)))
    assert ints(B.eval_poly(Lp, fe(mu)))[0] == 0
    tr.write_point(B.msm(srs.g, kate_div(Lp, mu)))
    if return_trace:
        trace.update(advice=advice, instances=instances, theta=theta, beta=beta, gamma=gamma, y=y, x=x,
                     zeta=zeta_c, nu=nu, mu=mu, evals=evals, pz=pz, lk=lk, hcoef=hcoef, draws=ctr[0],
                     adv_poly=adv_poly, challenges=list(chal))
        return bytes(tr.proof), instances, trace
    return bytes(tr.proof), instances


# ------------------------------------------------------------------------------ verifier
def proof_layout(plan):
    L, S = len(plan.lookups), plan.num_sets
    n_points = plan.num_advice + 2 * L + S + L + 1 + (plan.degree - 1)
    n_evals = len(plan.adv_q) + len(plan.fix_q) + 1 + len(plan.perm_cols) + (3 * S - 1 if S else 0) + 5 * L
    return n_points, n_evals, 64 * n_points + 32 * n_evals + 128


def verify(pk: ProvingKey, proof: bytes, instances, tau=None, g2=None, s_g2=None, trace=None):
    """verify_proof restated from Halo2Verifier.sol + codegen.  The final
    check e(lhs, g2) = e(W', s g2) is done with the trapdoor when `tau` is known (synthetic SRS)
    or with the BN254 pairing (oracle/pairing.py) when g2 / s_g2 are given.
    trace (dict, optional): receives what the reference's contract keeps in its named memory words
    (Halo2Verifier.sol:39-73) -- challenges, x^n, Lagrange values, instance / quotient evaluations, quotient
    commitment, pairing inputs -- as ints, keyed by the contract's constant names; pinned against the contract's own
    run by tests/golden/verifier_trace.json."""
    plan = pk.plan
    n = plan.n
    L, S = len(plan.lookups), plan.num_sets
    n_points, n_evals, plen = proof_layout(plan)
    if len(proof) != plen or len(instances) != plan.num_instance:
        return False
    if any(not (0 <= v < R) for v in instances):
        return False
    pos = [0]

    def rd_point(tr):
        b = proof[pos[0]:pos[0] + 64]
        pos[0] += 64
        x_, y_ = int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big")
        if x_ >= Q or y_ >= Q or (y_ * y_ - x_ * x_ * x_ - 3) % Q:
            raise ValueError("point not on curve")
        tr.common_point(b)
        return B.g1_affine_array([(x_, y_)])[0]

    tr = Transcript(pk.vk_digest)
    for v in instances:
        tr.common_scalar(v)
    try:
        adv_c = [None] * plan.num_advice
        chal = [None] * len(plan.challenge_phase)
        for ph in range(plan.num_phases):
            for c in range(plan.num_advice):
                if plan.advice_phase[c] == ph:
                    adv_c[c] = rd_point(tr)
            for i, cp in enumerate(plan.challenge_phase):
                if cp == ph:
                    chal[i] = tr.squeeze()
        theta = tr.squeeze()
        lk_c = [(rd_point(tr), rd_point(tr)) for _ in range(L)]
        beta = tr.squeeze()
        gamma = tr.squeeze()
        pz_c = [rd_point(tr) for _ in range(S)]
        lz_c = [rd_point(tr) for _ in range(L)]
        rand_c = rd_point(tr)
        y = tr.squeeze()
        h_c = [rd_point(tr) for _ in range(plan.degree - 1)]
        x = tr.squeeze()
        evs = []
        for _ in range(n_evals):
            v = int.from_bytes(proof[pos[0]:pos[0] + 32], "big")
            pos[0] += 32
            if v >= R:
                return False
            tr.common_scalar(v)
            evs.append(v)
        zeta_c = tr.squeeze()
        nu = tr.squeeze()
        W = rd_point(tr)
        mu = tr.squeeze()
        Wp = rd_point(tr)
    except ValueError:
        return False
    it = iter(evs)
    evals = {}
    for c, r in plan.adv_q: evals[(("adv", c), r)] = next(it)
    for c, r in plan.fix_q: evals[(("fix", c), r)] = next(it)
    evals[(("rand",), 0)] = next(it)
    for i in range(len(plan.perm_cols)): evals[(("sig", i), 0)] = next(it)
    for s in range(S):
        for r in (0, 1) + ((plan.rot_last,) if s != S - 1 else ()):
            evals[(("pz", s), r)] = next(it)
    for l in range(L):
        for key, r in ((("lz", l), 0), (("lz", l), 1), (("lpi", l), 0), (("lpi", l), -1), (("lpt", l), 0)):
            evals[(key, r)] = next(it)
    # Lagrange data (Halo2Verifier.sol:392-470)
    om = pk.omega
    xn = pow(x, n, R)
    lw = plan.bf + 1
    def lag(i):
        wi = pow(om, i % n, R)
        return (xn - 1) * pow(n, -1, R) % R * wi % R * pow((x - wi) % R, -1, R) % R
    l_last = lag(-lw)
    l_blind = sum(lag(i) for i in range(-lw + 1, 0)) % R
    l_0 = lag(0)
    inst_eval = sum(lag(i) * v for i, v in enumerate(instances)) % R
    l_act = (1 - l_last - l_blind) % R
    adv = lambda q: evals[(("adv", plan.adv_q[q][0]), plan.adv_q[q][1])]
    fix = lambda q: evals[(("fix", plan.fix_q[q][0]), plan.fix_q[q][1])]
    inst = lambda q: inst_eval
    numer = 0
    for e in run_program_scalar(plan.gate_prog, plan.gate_slots, plan.consts, adv, fix, inst, chal):
        numer = (numer * y + e) % R
    def col_eval(t, i):
        return evals[(("adv", i), 0)] if t == COL_ADVICE else evals[(("fix", i), 0)] if t == COL_FIXED else inst_eval
    if S:
        z0 = evals[(("pz", 0), 0)]
        numer = (numer * y + l_0 * (1 - z0)) % R
        zl = evals[(("pz", S - 1), 0)]
        numer = (numer * y + l_last * (zl * zl - zl)) % R
        for s in range(1, S):
            numer = (numer * y + l_0 * (evals[(("pz", s), 0)] - evals[(("pz", s - 1), plan.rot_last)])) % R
        for s in range(S):
            cols = list(range(s * plan.chunk, min((s + 1) * plan.chunk, len(plan.perm_cols))))
            lhs, rhs = evals[(("pz", s), 1)], evals[(("pz", s), 0)]
            for ci in cols:
                v = col_eval(*plan.perm_cols[ci])
                lhs = lhs * (v + beta * evals[(("sig", ci), 0)] + gamma) % R
                rhs = rhs * (v + beta * pow(pk.delta, ci, R) * x + gamma) % R
            numer = (numer * y + l_act * (lhs - rhs)) % R
    for l, (m, pi, si, pt, st) in enumerate(plan.lookups):
        def comp(prog, ns):
            acc = None
            for e in run_program_scalar(prog, ns, plan.consts, adv, fix, inst, chal):
                acc = e if acc is None else (acc * theta + e) % R
            return acc
        Ain, Stab = comp(pi, si), comp(pt, st)
        z, zn_, a_, ap, s_ = (evals[(("lz", l), 0)], evals[(("lz", l), 1)], evals[(("lpi", l), 0)],
                              evals[(("lpi", l), -1)], evals[(("lpt", l), 0)])
        numer = (numer * y + l_0 * (1 - z)) % R
        numer = (numer * y + l_last * (z * z - z)) % R
        numer = (numer * y + l_act * (zn_ * (a_ + beta) * (s_ + gamma) - z * (Ain + beta) * (Stab + gamma))) % R
        numer = (numer * y + l_0 * (a_ - s_)) % R
        numer = (numer * y + l_act * (a_ - s_) * (a_ - ap)) % R
    h_eval = numer * pow((xn - 1) % R, -1, R) % R
    evals[(("h",), 0)] = h_eval
    if trace is not None:
        if chal:
            trace.update(USER_CHALLENGES=list(chal))
        trace.update(THETA_MPTR=theta, BETA_MPTR=beta, GAMMA_MPTR=gamma, Y_MPTR=y, X_MPTR=x, ZETA_MPTR=zeta_c, NU_MPTR=nu,
                     MU_MPTR=mu, X_N_MPTR=xn, X_N_MINUS_1_INV_MPTR=pow((xn - 1) % R, -1, R), L_LAST_MPTR=l_last,
                     L_BLIND_MPTR=l_blind, L_0_MPTR=l_0, INSTANCE_EVAL_MPTR=inst_eval, QUOTIENT_EVAL_MPTR=h_eval)
    # commitment to h: sum_i x^(n i) H_i
    h_comm = None
    for c in reversed(h_c):
        h_comm = c if h_comm is None else B.g1_add(B.g1_mul(h_comm, fe(xn)), c)
    if trace is not None:
        hx_, hy_ = B.g1_affine_ints(h_comm.reshape(1, 8))[0]
        trace.update(QUOTIENT_X_MPTR=hx_, QUOTIENT_Y_MPTR=hy_)
    comms = {("h",): h_comm, ("rand",): rand_c}
    for c in range(plan.num_advice): comms[("adv", c)] = adv_c[c]
    for c in range(plan.num_fixed): comms[("fix", c)] = pk.fixed_commitments[c]
    for i in range(len(plan.perm_cols)): comms[("sig", i)] = pk.sigma_commitments[i]
    for s in range(S): comms[("pz", s)] = pz_c[s]
    for l in range(L):
        comms[("lz", l)], comms[("lpi", l)], comms[("lpt", l)] = lz_c[l], lk_c[l][0], lk_c[l][1]
    superset, sets = rotation_sets(build_queries(plan, None))
    xr = lambda r: x * pow(om, r % n, R) % R
    point = {r: xr(r) for r in superset}
    z0v = 1
    for r in sets[0][0]:
        z0v = z0v * (mu - point[r]) % R
    diffs = []
    for rots, _ in sets:
        d = 1
        for r in superset:
            if r not in rots:
                d = d * (mu - point[r]) % R
        diffs.append(d)
    d0inv = pow(diffs[0], -1, R)
    acc_pt, r_total, nu_pow = None, 0, 1
    for si_, (rots, keys) in enumerate(sets):
        # barycentric coefficients at mu (bdfg21.rs:140-200)
        co = {}
        for ri in rots:
            c = (mu - point[ri]) % R
            for rj in rots:
                if rj != ri:
                    c = c * (point[ri] - point[rj]) % R
            co[ri] = pow(c, -1, R)
        csum = sum(co.values()) % R
        r_eval, zp, F = 0, 1, None
        for key in keys:
            r_eval = (r_eval + zp * sum(co[r] * evals[(key, r)] for r in rots)) % R
            zp = zp * zeta_c % R
        for key in reversed(keys):
            F = comms[key] if F is None else B.g1_add(B.g1_mul(F, fe(zeta_c)), comms[key])
        r_eval = r_eval * pow(csum, -1, R) % R
        coef = nu_pow * diffs[si_] % R * d0inv % R
        r_total = (r_total + coef * r_eval) % R
        term = B.g1_mul(F, fe(coef))
        acc_pt = term if acc_pt is None else B.g1_add(acc_pt, term)
        nu_pow = nu_pow * nu % R
    G = B.g1_affine_array([(1, 2)])[0]
    lhs = B.g1_add(acc_pt, B.g1_mul(G, fe((-r_total) % R)))
    lhs = B.g1_add(lhs, B.g1_mul(W, fe((-z0v) % R)))
    lhs = B.g1_add(lhs, B.g1_mul(Wp, fe(mu)))
    if trace is not None:
        lx_, ly_ = B.g1_affine_ints(lhs.reshape(1, 8))[0]
        rx_, ry_ = B.g1_affine_ints(Wp.reshape(1, 8))[0]
        trace.update(G1_SCALAR_MPTR=(-r_total) % R, PAIRING_LHS_X_MPTR=lx_, PAIRING_LHS_Y_MPTR=ly_, PAIRING_RHS_X_MPTR=rx_,
                     PAIRING_RHS_Y_MPTR=ry_)
    if tau is not None:
        return bool(np.array_equal(lhs, B.g1_mul(Wp, fe(tau))))
    from . import pairing
    return pairing.check_kzg(B.g1_affine_ints(lhs.reshape(1, 8))[0], B.g1_affine_ints(Wp.reshape(1, 8))[0], g2, s_g2)
