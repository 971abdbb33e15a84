"""A small circuit with three phases -- advice columns assigned after challenges have been drawn from the transcript.

None of the reference's circuits has a second phase, but its verifier generator supports them
(/root/reference/crates/halo2-verifier/src/lib/codegen/util.rs:70-96 remaps advice columns and challenges by phase,
util.rs:133-165 and templates/Halo2Verifier.sol:267-282 read the commitments and squeeze the challenges phase by phase).
This circuit walks that path: a random linear combination and a running sum, the textbook uses of halo2's
`advice_column_in(SecondPhase)` / `challenge_usable_after(FirstPhase)`.

  phase 0   a, b            witness inputs; a is range-checked; a[0] is the public input
            r0, r1          challenges usable after phase 0 (two in one phase: the second is squeezed without new input)
  phase 1   c = a + r0 b    random linear combination of the first-phase columns
            s               running sum: s[0] = 0, s[i+1] = r1 s[i] + c[i]
            r2              challenge usable after phase 1
  phase 2   d = r2 c + s    a third-phase column over second-phase ones
            r3              challenge usable after the last phase: known to the gates only
  gates     q (c - a - r0 b),  q (s(next) - r1 s - c),  q r3 (d - r2 c - s),  q (b - a^2 - 3)
  lookup    q (c - r0 b)  in  [0, 16)      (an input expression that contains a challenge)
  copies    s[0] = a fixed zero,  d[0] = c[0] (d[0] = r2 c[0] + s[0] would break it, so row 0 of d is the copy and the
            gate starts at row 1),  a[0] = instance[0]

The advice columns are declared in the order a, c, b, d, s: the phases interleave, so commitment order (by phase, then by
column) differs from column order.
"""
from .builder import (PlanBuilder, Adv, Fix, Chal, Const, COL_ADVICE, COL_FIXED, COL_INSTANCE, W_INPUT, W_CONST, W_COPY, W_ADD,
                      W_MUL, W_ADDC, W_CHAL)

RANGE = 16


def build_phased_plan(k=6, rows=20, describe=False):
    """-> (plan bytes | description bytes, number of witness inputs).  Witness inputs: a[0..rows), each < 16."""
    b = PlanBuilder(k)
    a = b.advice_column(0)
    c = b.advice_column(1)
    bb = b.advice_column(0)
    d = b.advice_column(2)
    s = b.advice_column(1)
    q, q2, tab, zero = b.fixed_column(), b.fixed_column(), b.fixed_column(), b.fixed_column()
    r0, r1 = b.challenge(0), b.challenge(0)
    r2 = b.challenge(1)
    r3 = b.challenge(2)
    b.num_instance = 1
    b.gate(Fix(q) * (Adv(c) - Adv(a) - Chal(r0) * Adv(bb)))
    b.gate(Fix(q) * (Adv(s, 1) - Chal(r1) * Adv(s) - Adv(c)))
    b.gate(Fix(q2) * Chal(r3) * (Adv(d) - Chal(r2) * Adv(c) - Adv(s)))
    b.gate(Fix(q) * (Adv(bb) - Adv(a) * Adv(a) - Const(3)))
    b.lookup([Fix(q) * (Adv(c) - Chal(r0) * Adv(bb))], [Fix(tab)])
    for t, i in ((COL_ADVICE, a), (COL_ADVICE, s), (COL_ADVICE, d), (COL_ADVICE, c), (COL_FIXED, zero), (COL_INSTANCE, 0)):
        b.enable_equality(t, i)
    for r in range(RANGE):
        b.fix(tab, r, r)
    for r in range(rows):
        b.fix(q, r, 1)
        if r:
            b.fix(q2, r, 1)
    # ---- phase 0
    for r in range(rows):
        b.w(W_INPUT, (a, r), imm=r)
        t = b.scratch(0)
        b.w(W_MUL, t, (a, r), (a, r))
        b.w(W_ADDC, (bb, r), t, imm=b.const_index(3))
    b.n_winputs = rows
    b.next_phase()
    # ---- phase 1: c and the running sum
    ch0, ch1, ch2 = b.scratch(1), b.scratch(2), b.scratch(3)
    b.w(W_CHAL, ch0, imm=r0)
    b.w(W_CHAL, ch1, imm=r1)
    b.w(W_CONST, (s, 0), imm=b.const_index(0))
    for r in range(rows):
        t = b.scratch(0)
        b.w(W_MUL, t, ch0, (bb, r))
        b.w(W_ADD, (c, r), (a, r), t)
        b.w(W_MUL, t, ch1, (s, r))
        b.w(W_ADD, (s, r + 1), t, (c, r))
    b.next_phase()
    # ---- phase 2
    b.w(W_CHAL, ch2, imm=r2)
    b.w(W_COPY, (d, 0), (c, 0))
    for r in range(1, rows):
        t = b.scratch(0)
        b.w(W_MUL, t, ch2, (c, r))
        b.w(W_ADD, (d, r), t, (s, r))
    b.copy((COL_ADVICE, s, 0), (COL_FIXED, zero, 0))
    b.copy((COL_ADVICE, d, 0), (COL_ADVICE, c, 0))
    b.instance_map.append((a, 0))
    b.copy((COL_ADVICE, a, 0), (COL_INSTANCE, 0, 0))
    return (b.build_description() if describe else b.build()), rows


def phased_witness_inputs(rows, rnd):
    return [rnd.randrange(RANGE) for _ in range(rows)]

