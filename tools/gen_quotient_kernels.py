#!/usr/bin/env python3
"""Ahead-of-time QUOTIENT kernels (gates + permutation + lookup terms in one pass) in 29-bit lazy arithmetic.

For every circuit plan the library ships, this tool writes one straight-line HIP kernel that forms the numerator of the
quotient polynomial at one point of the quotient domain: the plan's gate program, the permutation argument's terms and
every lookup's terms (its two compression programs inline), folded with y in the order the prover has always used --
so the values, and the proof bytes, are those of the interpreter and the separate kernels.  The arithmetic is
Lazy29<FrParams> (csrc/lazy29.hip.hpp): 9 limbs of 29 bits, Montgomery radix 2^261, values kept as small multiples of
p.  The inputs of such a kernel are stored in that radix (the key's coset tables carry the factor 32, see keygen in
csrc/prover.hip), so nothing is converted per element; only the result goes back to radix 2^256.

Lazy arithmetic needs a bound for every intermediate value.  They are tracked HERE, statically, while the code is
emitted: every value has V (value < V p) and L (limbs 0..7 < L 2^29) and the emitter inserts a carry pass or a reduction
exactly where a rule of lazy29.hip.hpp would otherwise be broken.  `--selfcheck` runs the very same emission on concrete
integers (random and extreme inputs), asserting at every operation that no 64-bit column sum or 32-bit limb overflows,
that no subtraction goes negative, that the value respects its static bound, and that the result is the one plain
modular arithmetic gives.

   python tools/gen_quotient_kernels.py               # rewrites csrc/quotient_kernels.inc
   python tools/gen_quotient_kernels.py --check       # exit 1 if the committed file is not what the plans generate
   python tools/gen_quotient_kernels.py --selfcheck   # the bound checker described above
"""
import importlib
import math
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_gate_kernels import selector_runs  # noqa: E402

OUT = os.path.join(ROOT, "zkos-monorepo_amd", "csrc", "quotient_kernels.inc")
OP_CONST, OP_ADV, OP_FIX, OP_INST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_OUT = range(9)
COL_ADVICE, COL_FIXED, COL_INSTANCE = 0, 1, 2

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617      # BN254 scalar field
M29 = (1 << 29) - 1
R261 = pow(2, 261, P)
P29 = [(P >> (29 * i)) & M29 for i in range(9)]
PINV29 = (-pow(P, -1, 1 << 29)) & M29
MUL_SHRINK = 164                      # p / 2^261 < 1 / 164 (lazy29.hip.hpp)
V_MUL_MAX = 160                       # operands of a product: value below 160 p keeps the top limb below 2^29
V_ADD_MAX = 1000                      # the top limb is a 32-bit word: value < 2^264 ~ 1290 p
# Leaves whose loads are in flight ahead of the one in use.  Left to itself the compiler puts every load next to its use (it
# is short of registers) and the wave then waits out the memory latency 144 times: half of a lone wave's time.  Measured in the
# pool: 2295 / 2337 / 2351 / 2345 proofs/s at 0 / 3 / 4 / 6 (each leaf in flight holds eight registers; the kernels are capped
# at 256 VGPRs -- two waves per SIMD -- and spill 28 bytes at 4, 100 at 6)
PREFETCH_AHEAD = int(os.environ.get("QUOT29_PREFETCH_AHEAD", "4"))
SECTION_MIN_LOADS = int(os.environ.get("QUOT29_SECTION_MIN_LOADS", "12"))    # gates are grouped into sections of at least this many loads


# ------------------------------------------------------------------------------------------------ plan parsing
def parse_plan(data):
    """the parts of a circuit plan (layout: zkos-monorepo_amd/circuits/builder.py) the quotient depends on"""
    assert data[:4] == b"SHPL"
    h = struct.unpack_from("<20I", data, 4)
    deg, naq, nfq, niq, npc, ncst, ngi, slots, nlk = h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[15]
    off = 84

    def recs(cnt, fmt, size):
        nonlocal off
        out = [struct.unpack_from(fmt, data, off + size * i) for i in range(cnt)]
        off += size * cnt
        return out

    adv_q, fix_q, inst_q = recs(naq, "<Ii", 8), recs(nfq, "<Ii", 8), recs(niq, "<Ii", 8)
    perm_cols = recs(npc, "<II", 8)
    off += 32 * ncst
    prog = recs(ngi, "<4I", 16)
    lookups = []
    for _ in range(nlk):
        m, n_in, s_in, n_tab, s_tab = struct.unpack_from("<5I", data, off)
        off += 20
        lookups.append((m, s_in, recs(n_in, "<4I", 16), s_tab, recs(n_tab, "<4I", 16)))
    return dict(degree=deg, slots=slots, prog=prog, adv_q=adv_q, fix_q=fix_q, inst_q=inst_q, perm_cols=perm_cols,
                lookups=lookups, n_consts=ncst)


def quotient_hash(pl):
    """FNV-1a (64 bit) over what the generated code depends on; csrc/prover.hip computes the same value at keygen"""
    h = 0xCBF29CE484222325

    def eat(word):
        nonlocal h
        for b in struct.pack("<I", word & 0xFFFFFFFF):
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF

    def eat_prog(slots, prog):
        eat(slots)
        eat(len(prog))
        for ins in prog:
            for w in ins:
                eat(w)

    eat_prog(pl["slots"], pl["prog"])
    for qs in (pl["adv_q"], pl["fix_q"], pl["inst_q"]):
        eat(len(qs))
        for col, rot in qs:
            eat(col)
            eat(rot)
    eat(pl["degree"])
    eat(len(pl["perm_cols"]))
    for t, ix in pl["perm_cols"]:
        eat(t)
        eat(ix)
    eat(len(pl["lookups"]))
    for m, s_in, p_in, s_tab, p_tab in pl["lookups"]:
        eat(m)
        eat_prog(s_in, p_in)
        eat_prog(s_tab, p_tab)
    return h


# ------------------------------------------------------------------------------------------------ concrete arithmetic
def c_value(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def c_split(v):
    assert 0 <= v < (1 << 256)
    return [(v >> (29 * i)) & M29 for i in range(9)]


def c_limbs_ok(l):
    assert all(0 <= x < (1 << 32) for x in l), "a limb left its 32-bit word"


def c_mul(a, b):
    acc, q, r = 0, [0] * 9, [0] * 9
    for k in range(9):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
        for i in range(k):
            acc += q[i] * P29[k - i]
        assert acc < (1 << 64), "column sum overflow"
        q[k] = ((acc & 0xFFFFFFFF) * PINV29) & M29
        acc += q[k] * P29[0]
        assert acc < (1 << 64), "column sum overflow"
        assert acc & M29 == 0
        acc >>= 29
    for k in range(9, 17):
        for i in range(k - 8, 9):
            acc += a[i] * b[k - i] + q[i] * P29[k - i]
        assert acc < (1 << 64), "column sum overflow"
        r[k - 9] = acc & M29
        acc >>= 29
    assert acc < (1 << 32), "top limb overflow"
    r[8] = acc
    return r


def c_sqr(a):
    d = [x << 1 for x in a]
    c_limbs_ok(d)
    acc, q, r = 0, [0] * 9, [0] * 9
    for k in range(17):
        lo = 0 if k < 9 else k - 8
        i = lo
        while 2 * i < k:
            acc += d[i] * a[k - i]
            i += 1
        if k % 2 == 0:
            acc += a[k // 2] * a[k // 2]
        if k < 9:
            for i in range(k):
                acc += q[i] * P29[k - i]
            assert acc < (1 << 64), "column sum overflow"
            q[k] = ((acc & 0xFFFFFFFF) * PINV29) & M29
            acc += q[k] * P29[0]
            assert acc < (1 << 64), "column sum overflow"
            acc >>= 29
        else:
            for i in range(k - 8, 9):
                acc += q[i] * P29[k - i]
            assert acc < (1 << 64), "column sum overflow"
            r[k - 9] = acc & M29
            acc >>= 29
    assert acc < (1 << 32)
    r[8] = acc
    return r


def c_normalise(a):
    a = list(a)
    for i in range(8):
        a[i + 1] += a[i] >> 29
        a[i] &= M29
    c_limbs_ok(a)
    return a


def kp_spread(K, S):
    t, carry = [0] * 9, 0
    for i in range(9):
        acc = K * P29[i] + carry
        t[i] = acc & M29 if i < 8 else acc
        carry = acc >> 29
    for i in range(8):
        t[i] += S << 29
        t[i + 1] -= S
    assert all(0 <= x < (1 << 32) for x in t)
    return t


# ------------------------------------------------------------------------------------------------ the emitter
class Val:
    """one SSA value of the generated code: name, static bounds (value < V p, limbs 0..7 < L 2^29), and -- in a
    self-check run -- its concrete limbs `c` and the field element `m` it must spell (c = m 2^261 mod p)"""
    __slots__ = ("name", "V", "L", "c", "m", "zero")

    def __init__(self, name, V, L, c=None, m=None, zero=False):
        self.name, self.V, self.L, self.c, self.m, self.zero = name, V, L, c, m, zero


class Emitter:
    def __init__(self, source=None):
        self.lines = []
        self.n = 0
        self.kps = {}                 # (K, S) -> constant name
        self.loads = {}               # pointer expression -> Val
        self.source = source          # self-check: pointer expression -> stored 256-bit value (canonical, radix 2^261)
        self.n_mul = self.n_add = self.n_norm = self.n_red = 0
        self.n_loads_since_barrier = 0
        self.used = set()             # names that appear as an operand (a leaf whose only consumer was folded away is dead)
        self.leaves = []
        self.leaf_ptrs = []           # pointer expression of every leaf load, in program order
        self.ahead = None             # second pass: the first pass's leaf_ptrs -- leaves are loaded PREFETCH_AHEAD loads early
        self.raw_emitted = 0

    # -- bookkeeping
    def fresh(self):
        self.n += 1
        return "t%d" % self.n

    def use(self, *vals):
        for v in vals:
            self.used.add(v.name)

    def check(self, v):
        if v.c is None:
            return v
        c_limbs_ok(v.c)
        assert all(x < v.L << 29 for x in v.c[:8]), "limb bound broken: %s" % v.name
        val = c_value(v.c)
        assert val < v.V * P + 1e-9 * P, "value bound broken: %s (%.3f p, bound %.3f)" % (v.name, val / P, v.V)
        assert val % P == v.m * R261 % P, "wrong value: %s" % v.name
        return v

    def kp(self, K, S):
        if (K, S) not in self.kps:
            self.kps[(K, S)] = "KP_%d_%d" % (K, S)
        return self.kps[(K, S)]

    def barrier(self, min_loads=0):
        """Forget the loaded values: a later use loads again instead of holding nine registers across hundreds of
        operations (the compiler barrier keeps the compiler from merging the two loads back into one).  Every barrier
        is also a stall -- the loads after it cannot start before it -- so short sections (fewer than `min_loads`
        loads since the last barrier) are merged with what follows."""
        if self.n_loads_since_barrier < min_loads:
            return
        self.n_loads_since_barrier = 0
        self.lines.append('    asm volatile("" ::: "memory");')
        self.loads = {k: v for k, v in self.loads.items() if k == "ONE" or k.startswith("a.") and " " not in k}

    # -- leaves
    def load(self, ptr):
        if ptr in self.loads:
            return self.loads[ptr]
        name = self.fresh()
        self.n_loads_since_barrier += 1
        self.leaves.append(name)
        k = len(self.leaf_ptrs)
        self.leaf_ptrs.append(ptr)
        if self.ahead is None:
            self.lines.append("    const E29 %s = ld29(%s);" % (name, ptr))
        else:
            # the raw words of this leaf were requested PREFETCH_AHEAD leaves ago; request those of the leaves ahead now (the
            # compiler barrier keeps each request where it is written instead of next to its use)
            assert self.ahead[k] == ptr
            for j in range(self.raw_emitted, min(k + PREFETCH_AHEAD + 1, len(self.ahead))):
                self.lines.append('    const Fr raw%d = ld(%s); asm volatile("" ::: "memory");' % (j, self.ahead[j]))
            self.raw_emitted = max(self.raw_emitted, min(k + PREFETCH_AHEAD + 1, len(self.ahead)))
            self.lines.append("    const E29 %s = R29::split<0>(raw%d);" % (name, k))
        c = m = None
        if self.source is not None:
            stored = self.source(ptr)
            c, m = c_split(stored), stored * pow(R261, -1, P) % P
        v = self.check(Val(name, 1.0, 1, c, m))
        self.loads[ptr] = v
        return v

    def arg(self, expr, key):
        """a field element passed by value (canonical, radix 2^261)"""
        return self.load_named("split29r(%s)" % expr, key)

    def load_named(self, expr, key):
        if key in self.loads:
            return self.loads[key]
        name = self.fresh()
        self.lines.append("    const E29 %s = %s;" % (name, expr))
        c = m = None
        if self.source is not None:
            stored = self.source(key)
            c, m = c_split(stored), stored * pow(R261, -1, P) % P
        v = self.check(Val(name, 1.0, 1, c, m))
        self.loads[key] = v
        return v

    def one(self):
        if "ONE" in self.loads:
            return self.loads["ONE"]
        name = self.fresh()
        self.lines.append("    const E29 %s = R29::from_l9(R29::ONE);" % name)
        v = self.check(Val(name, 1.0, 1, c_split(R261) if self.source is not None else None, 1))
        self.loads["ONE"] = v
        return v

    def zero(self):
        return Val("<zero>", 0.0, 1, [0] * 9 if self.source is not None else None, 0, zero=True)

    # -- shaping
    def normalised(self, a):
        if a.L == 1:
            return a
        self.n_norm += 1
        self.use(a)
        name = self.fresh()
        self.lines.append("    E29 %s = %s; R29::normalise(%s);" % (name, a.name, name))
        return self.check(Val(name, a.V, 1, c_normalise(a.c) if a.c is not None else None, a.m))

    def reduced(self, a):
        """a value below (V / 164 + 1) p: one product by the Montgomery one"""
        self.n_red += 1
        return self.mul(a, self.one(), force=True)

    # -- arithmetic
    def add(self, a, b):
        if a.zero:
            return b
        if b.zero:
            return a
        if a.V + b.V > V_ADD_MAX:
            a, b = (self.reduced(a), b) if a.V >= b.V else (a, self.reduced(b))
        if a.L + b.L > 4:
            a, b = (self.normalised(a), b) if a.L >= b.L else (a, self.normalised(b))
        self.n_add += 1
        self.use(a, b)
        name = self.fresh()
        self.lines.append("    const E29 %s = add29(%s, %s);" % (name, a.name, b.name))
        c = [x + y for x, y in zip(a.c, b.c)] if a.c is not None else None
        return self.check(Val(name, a.V + b.V, a.L + b.L, c, (a.m + b.m) % P if a.c is not None else None))

    def sub(self, a, b):
        if b.zero:
            return a
        if b.V > V_MUL_MAX:
            b = self.reduced(b)
        if b.L > 3:
            b = self.normalised(b)
        if a.L > 3:
            a = self.normalised(a)
        K, S = int(math.ceil(b.V + 0.5)), b.L
        if a.V + K > V_ADD_MAX:
            a = self.reduced(a)
        self.n_add += 1
        self.use(a, b)
        name = self.fresh()
        if a.zero:
            self.lines.append("    const E29 %s = neg29(%s, %s);" % (name, b.name, self.kp(K, S)))
        else:
            self.lines.append("    const E29 %s = R29::sub(%s, %s, %s);" % (name, a.name, b.name, self.kp(K, S)))
        c = None
        if b.c is not None:
            kp = kp_spread(K, S)
            ac = a.c if not a.zero else [0] * 9
            r = [x + k - y for x, k, y in zip(ac, kp, b.c)]
            assert all(0 <= x < (1 << 32) for x in r), "a limb of a subtraction left its word"
            c = c_normalise(r)
        return self.check(Val(name, a.V + K, 1, c, (a.m - b.m) % P if c is not None else None))

    def neg(self, a):
        return self.sub(self.zero(), a)

    def mul(self, a, b, force=False):
        if a.zero or b.zero:
            return self.zero()
        if not force:
            if a.V > V_MUL_MAX:
                a = self.reduced(a)
            if b.V > V_MUL_MAX:
                b = self.reduced(b)
        if a is b or a.name == b.name:
            a = self.normalised(a)
            self.n_mul += 1
            self.use(a)
            name = self.fresh()
            self.lines.append("    const E29 %s = R29::sqr(%s);" % (name, a.name))
            c = c_sqr(a.c) if a.c is not None else None
            return self.check(Val(name, a.V * a.V / MUL_SHRINK + 1, 1, c, a.m * a.m % P if c is not None else None))
        while a.L * b.L > 4:
            if a.L >= b.L:
                a = self.normalised(a)
            else:
                b = self.normalised(b)
        self.n_mul += 1
        self.use(a, b)
        name = self.fresh()
        self.lines.append("    const E29 %s = R29::mul(%s, %s);" % (name, a.name, b.name))
        c = c_mul(a.c, b.c) if a.c is not None else None
        return self.check(Val(name, a.V * b.V / MUL_SHRINK + 1, 1, c, a.m * b.m % P if c is not None else None))

    def fold(self, h, y, term):
        """h y + term"""
        return self.add(self.mul(h, y), term)


# ------------------------------------------------------------------------------------------------ code generation
def row_expr(rot):
    if rot == 0:
        return "i"
    return "(base | ((i + (uint32_t)(%d)) & a.mask))" % rot


def emit_program(em, pl, slots, prog, mult, what):
    """the three-address program as straight-line lazy arithmetic; returns the folded value (sum of OUT values times
    descending powers of `mult`)"""
    adv_q, fix_q = pl["adv_q"], pl["fix_q"]
    runs = selector_runs(prog, fix_q)
    elided = {v[3]: i for i, v in runs.items()}
    s = [em.zero() for _ in range(slots)]
    acc, inner = em.zero(), None
    powers = {}

    def power(k):                     # `mult` is a callable: the challenge is split into limbs only if a fold uses it
        if k == 1:
            return mult()
        if k not in powers:
            h = power(k // 2)
            sq = em.mul(h, h)
            powers[k - k % 2] = sq
            if k % 2:
                powers[k] = em.mul(sq, mult())
        return powers[k]

    def fold(h, term):
        return term if h.zero else em.fold(h, mult(), term)

    gate_val = {}
    em.lines.append("    // ---- %s: %d instructions" % (what, len(prog)))
    for i, (op, dst, x, y) in enumerate(prog):
        if op == OP_CONST:
            s[dst] = em.load("a.consts + %d" % x)
        elif op == OP_ADV:
            s[dst] = em.load("a.cos + (size_t)%d * a.ne + %s" % (adv_q[x][0], row_expr(adv_q[x][1])))
        elif op == OP_FIX:
            s[dst] = em.load("a.fixed_cos + (size_t)%d * a.ne + %s" % (fix_q[x][0], row_expr(fix_q[x][1])))
        elif op == OP_INST:
            s[dst] = em.load("a.cos + (size_t)a.i_inst * a.ne + %s" % row_expr(pl["inst_q"][x][1]))
        elif op == OP_ADD:
            s[dst] = em.add(s[x], s[y])
        elif op == OP_SUB:
            s[dst] = em.sub(s[x], s[y])
        elif op == OP_MUL:
            if i in elided:
                gate_val[elided[i]] = s[runs[elided[i]][4]]          # the selector is applied once per run
            else:
                s[dst] = em.mul(s[x], s[y])
        elif op == OP_NEG:
            s[dst] = em.neg(s[x])
        elif op == OP_OUT:
            em.barrier(SECTION_MIN_LOADS)
            if i in runs:
                rid, pos, length, w, expr_slot, (col, rot) = runs[i]
                inner = gate_val[i] if pos == 0 else fold(inner, gate_val[i])
                if pos == length - 1:
                    sel = em.load("a.fixed_cos + (size_t)%d * a.ne + %s" % (col, row_expr(rot)))
                    acc = em.mul(sel, inner) if acc.zero else em.add(em.mul(acc, power(length)), em.mul(sel, inner))
            else:
                acc = fold(acc, s[x])
        else:
            raise ValueError(op)
    return acc


def emit_quotient(em, pl):
    y = em.arg("a.y", "a.y")
    h = emit_program(em, pl, pl["slots"], pl["prog"], lambda: y, "gates")
    perm, chunk = pl["perm_cols"], pl["degree"] - 2
    n_sets = (len(perm) + chunk - 1) // chunk if perm else 0
    L = len(pl["lookups"])
    l0, ll, la = (lambda: em.load("a.l0 + i")), (lambda: em.load("a.l_last + i")), (lambda: em.load("a.l_active + i"))
    one = em.one()
    if n_sets:
        em.barrier()
        em.lines.append("    // ---- permutation argument: %d columns in %d sets" % (len(perm), n_sets))
        gamma = em.arg("a.gamma", "a.gamma")

        def pz(s, row):
            return em.load("a.cos + (size_t)(a.i_pz + %d) * a.ne + %s" % (s, row))

        h = em.fold(h, y, em.mul(l0(), em.sub(one, pz(0, "i"))))
        zl = pz(n_sets - 1, "i")
        h = em.fold(h, y, em.mul(ll(), em.sub(em.mul(zl, zl), zl)))
        for s in range(1, n_sets):
            h = em.fold(h, y, em.mul(l0(), em.sub(pz(s, "i"), pz(s - 1, "ilast"))))
        # beta sigma + v + gamma = beta (sigma + t) and beta delta^c x + v + gamma = beta (delta^c x + t) with
        # t = (v + gamma) / beta: three products per column instead of four (delta^c x is a table of the key), and
        # beta^(columns of the set) once per set
        binv = em.arg("a.beta_inv", "a.beta_inv")
        for s in range(n_sets):
            em.barrier()
            left, right = pz(s, "inext"), pz(s, "i")
            cols = range(s * chunk, min(len(perm), (s + 1) * chunk))
            for c in cols:
                t, ix = perm[c]
                if t == COL_ADVICE:
                    v = em.load("a.cos + (size_t)%d * a.ne + i" % ix)
                elif t == COL_FIXED:
                    v = em.load("a.fixed_cos + (size_t)%d * a.ne + i" % ix)
                else:
                    v = em.load("a.cos + (size_t)a.i_inst * a.ne + i")
                tq = em.mul(em.add(v, gamma), binv)
                left = em.mul(left, em.add(em.load("a.sigma_cos + (size_t)%d * a.ne + i" % c), tq))
                right = em.mul(right, em.add(em.load("a.dx_ext + (size_t)%d * a.ne + i" % c), tq))
            bpow = em.arg("a.beta_pow_full", "a.beta_pow_full") if len(cols) == chunk else em.arg("a.beta_pow_last", "a.beta_pow_last")
            h = em.fold(h, y, em.mul(em.mul(la(), em.sub(left, right)), bpow))
    for l, (m, s_in, p_in, s_tab, p_tab) in enumerate(pl["lookups"]):
        em.barrier()
        theta = lambda: em.arg("a.theta", "a.theta")
        beta, gamma = em.arg("a.beta", "a.beta"), em.arg("a.gamma", "a.gamma")
        ain = emit_program(em, pl, s_in, p_in, theta, "lookup %d: input expressions folded with theta" % l)
        stab = emit_program(em, pl, s_tab, p_tab, theta, "lookup %d: table expressions folded with theta" % l)
        em.lines.append("    // ---- lookup %d: the five terms" % l)

        def col(idx, row):
            return em.load("a.cos + (size_t)(%s) * a.ne + %s" % (idx, row))

        z, zn = col("a.i_lz + %d" % l, "i"), col("a.i_lz + %d" % l, "inext")
        ap, sp = col("a.i_pin + %d" % (2 * l), "i"), col("a.i_pin + %d" % (2 * l + 1), "i")
        aprev = col("a.i_pin + %d" % (2 * l), "iprev")
        h = em.fold(h, y, em.mul(l0(), em.sub(one, z)))
        h = em.fold(h, y, em.mul(ll(), em.sub(em.mul(z, z), z)))
        lhs = em.mul(em.mul(zn, em.add(ap, beta)), em.add(sp, gamma))
        rhs = em.mul(em.mul(z, em.add(ain, beta)), em.add(stab, gamma))
        h = em.fold(h, y, em.mul(la(), em.sub(lhs, rhs)))
        d = em.sub(ap, sp)
        h = em.fold(h, y, em.mul(l0(), d))
        h = em.fold(h, y, em.mul(em.mul(la(), d), em.sub(ap, aprev)))
    # back to radix 2^256, canonical: to_fp multiplies by 2^256 (result below (V / 164 + 1) p < 2 p) and subtracts p once
    if h.V > V_MUL_MAX:
        h = em.reduced(h)
    if h.L > 4:
        h = em.normalised(h)
    em.use(h)
    dead = {n for n in em.leaves if n not in em.used}       # e.g. a selector loaded by the program and applied once per run instead
    dead_raw = {"raw%d" % k for k, n in enumerate(em.leaves) if n in dead}
    em.lines = [l for l in em.lines if not any(l.startswith("    const E29 %s = ld29(" % n) or l.startswith("    const E29 %s = R29::split<0>(raw" % n) for n in dead)
                and not any(l.startswith("    const Fr %s = ld(" % r) for r in dead_raw)]
    em.lines.append("    st(a.h + i, R29::to_fp(%s));" % h.name)
    return h


def kernel_text(name, pl):
    em = Emitter()
    if PREFETCH_AHEAD:
        dry = Emitter()
        emit_quotient(dry, pl)
        em.ahead = dry.leaf_ptrs
    emit_quotient(em, pl)
    head = ["__global__ __launch_bounds__(QUOT29_THREADS, 2) void %s(Quot29Args a) {" % name,
            "    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;",
            "    const uint32_t base = i & ~a.mask;",
            "    const uint32_t inext = base | ((i + 1) & a.mask), iprev = base | ((i - 1) & a.mask);",
            "    const uint32_t ilast = base | ((i + (uint32_t)a.rot_last) & a.mask);",
            "    (void)inext; (void)iprev; (void)ilast;"]
    for (K, S), nm in sorted(em.kps.items()):
        head.append("    constexpr l29::L9 %s = R29::kp_spread(%d, %d);" % (nm, K, S))
    stats = "// %d products, %d additions / subtractions, %d carry passes, %d reductions" % (em.n_mul, em.n_add, em.n_norm, em.n_red)
    return stats + "\n" + "\n".join(head + em.lines + ["}"])


def shipped_plans():
    circuits = importlib.import_module("zkos-monorepo_amd.circuits")
    return [("withdraw", circuits.build_withdraw_plan()[0]),
            ("deposit", circuits.build_deposit_plan()[0]),
            ("new_account", circuits.build_new_account_plan()[0])]


def distinct_plans():
    seen = {}
    for label, plan in shipped_plans():
        pl = parse_plan(plan)
        h = quotient_hash(pl)
        if h in seen:
            seen[h] = (seen[h][0] + ", " + label, pl)
        else:
            seen[h] = (label, pl)
    return seen


def generate():
    out = ["// GENERATED by tools/gen_quotient_kernels.py from the shipped circuit plans -- do not edit.",
           "// One straight-line kernel per distinct plan structure: the gate program, the permutation terms and the lookup",
           "// terms of the quotient's numerator at one point of the quotient domain, in 29-bit lazy arithmetic (bounds",
           "// tracked by the generator, see its --selfcheck), and the registry keygen consults.  Compiled in quotient.hip.", ""]
    reg = []
    for h, (label, pl) in sorted(distinct_plans().items()):
        name = "quotient29_%016x" % h
        out.append("// %s: gate program of %d instructions, %d permutation columns, %d lookups" %
                   (label, len(pl["prog"]), len(pl["perm_cols"]), len(pl["lookups"])))
        out.append(kernel_text(name, pl))
        out.append("")
        reg.append("    {0x%016xull, %s}," % (h, name))
    out.append("struct QuotientKernelEntry { uint64_t hash; void (*fn)(Quot29Args); };")
    out.append("static const QuotientKernelEntry QUOTIENT_KERNELS[] = {")
    out += reg
    out.append("};")
    return "\n".join(out) + "\n"


# ------------------------------------------------------------------------------------------------ self-check
def c_to_fp(c):
    """lazy29.hip.hpp to_fp: product by 2^256, repack, one conditional subtraction"""
    t = c_mul(c, c_split(pow(2, 256, P)))
    v = c_value(t)
    assert v < 2 * P and v < (1 << 256)
    return v - P if v >= P else v


def selfcheck(rounds=40, seed=1):
    rng = random.Random(seed)
    for label, pl in distinct_plans().values():
        for r in range(rounds):
            mode = r % 5
            table = {}

            def source(key):
                if key not in table:
                    if mode == 0:
                        table[key] = P - 1
                    elif mode == 1:
                        table[key] = 0
                    elif mode == 2:
                        table[key] = rng.choice((0, 1, P - 1, P - 2, (1 << 253) - 1))
                    else:
                        table[key] = rng.randrange(P)
                return table[key]

            em = Emitter(source)
            h = emit_quotient(em, pl)
            got = c_to_fp(h.c if h.L == 1 else h.c)
            assert got == h.m * pow(2, 256, P) % P, "final value differs"
        print("%s: %d rounds, %d products / %d additions / %d carry passes / %d reductions per point -- bounds hold" %
              (label, rounds, em.n_mul, em.n_add, em.n_norm, em.n_red))


def main():
    if "--selfcheck" in sys.argv:
        selfcheck()
        return
    text = generate()
    if "--check" in sys.argv:
        same = os.path.exists(OUT) and open(OUT).read() == text
        print("quotient_kernels.inc is %s" % ("current" if same else "STALE: run tools/gen_quotient_kernels.py"))
        sys.exit(0 if same else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, "(%d bytes)" % len(text))


if __name__ == "__main__":
    main()
