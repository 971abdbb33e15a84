"""Random circuits with several phases (helper of the CPU and GPU phase tests; no test of its own).

halo2 lets a circuit place advice columns in later phases (`advice_column_in`) and draw challenges between them
(`challenge_usable_after`); the reference's verifier generator reads such proofs phase by phase
(/root/reference/crates/halo2-verifier/src/lib/codegen/util.rs:70-96, templates/Halo2Verifier.sol:267-282).  A generated
circuit here has 2-3 phases, 1-2 columns per phase declared in shuffled order (so the phases interleave), 1-2 challenges
after every phase but the last and 0-1 after the last, and per later-phase column a gate  q (col - E)  where E is a random
expression over columns of earlier phases (rotations 0 / 1), constants and the challenges known by then; the witness program
computes E with W_CHAL.  One lookup (optionally with a challenge in its input expression), copy constraints that tie cells of
different phases, 0-2 public inputs.  The circuit is satisfied by construction, whatever the challenges turn out to be."""
import importlib
import random

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
T = 12


def random_phased_plan(seed, describe=False):
    """-> (plan bytes | description bytes, k, witness input values)"""
    B = importlib.import_module("zkos-monorepo_amd.circuits.builder")
    rnd = random.Random(0x9E3779B9 ^ seed)
    k = rnd.choice([6, 6, 7])
    b = B.PlanBuilder(k)
    P = rnd.choice([2, 2, 3])
    decl = [ph for ph in range(P) for _ in range(rnd.randint(1, 2))]
    decl += [0] if decl.count(0) < 2 else []                # two first-phase columns: one for the lookup, one free
    rnd.shuffle(decl)
    cols = [b.advice_column(ph) for ph in decl]
    by_phase = {ph: [c for c, p in zip(cols, decl) if p == ph] for ph in range(P)}
    chals = {ph: [b.challenge(ph) for _ in range(rnd.randint(1, 2) if ph < P - 1 else rnd.randint(0, 1))] for ph in range(P)}
    q, tab, fz = b.fixed_column(), b.fixed_column(), b.fixed_column()
    lk_col = by_phase[0][0]
    rows = rnd.randint(8, 24)
    n_inst = rnd.randint(0, 2)
    b.num_instance = n_inst

    def earlier(ph):
        return [c for p in range(ph) for c in by_phase[p]]

    def known(ph):
        return [i for p in range(ph) for i in chals[p]]

    def tree(ph, deg):
        r = rnd.random()
        if deg <= 1 or r < 0.2:
            r = rnd.random()
            if r < 0.55:
                return ("adv", rnd.choice(earlier(ph)), rnd.choice([0, 0, 1]))
            if r < 0.85 and known(ph):
                return ("chal", rnd.choice(known(ph)))
            return ("const", rnd.randrange(1, 1000))
        if r < 0.6:
            d1 = rnd.randint(1, deg - 1)
            return ("mul", tree(ph, d1), tree(ph, deg - d1))
        return (rnd.choice(["add", "sub"]), tree(ph, deg), tree(ph, rnd.randint(1, deg)))

    def expr(t):
        if t[0] == "adv": return B.Adv(t[1], t[2])
        if t[0] == "chal": return B.Chal(t[1])
        if t[0] == "const": return B.Const(t[1])
        x, y = expr(t[1]), expr(t[2])
        return x * y if t[0] == "mul" else x + y if t[0] == "add" else x - y

    trees = {}
    for ph in range(1, P):
        for c in by_phase[ph]:
            trees[c] = tree(ph, rnd.randint(1, 3))
            e = expr(trees[c])
            gate = B.Fix(q) * (B.Adv(c) - e)
            late = [i for p in range(ph, P) for i in chals[p]]
            if late and rnd.random() < 0.5:                 # a challenge that only the gates see (drawn after this phase)
                gate = B.Chal(rnd.choice(late)) * gate
            b.gate(gate)
    lk_in = B.Adv(lk_col)
    c_last = by_phase[P - 1][0]
    all_ch = [i for p in range(P) for i in chals[p]]
    if rnd.random() < 0.6:
        lk_in = lk_in + B.Chal(rnd.choice(all_ch)) * (B.Adv(c_last) - expr(trees[c_last]))
    b.lookup([B.Fix(q) * lk_in], [B.Fix(tab)])
    for r in range(T):
        b.fix(tab, r, r)
    for r in range(rows):
        b.fix(q, r, 1)
    eq_cols = [c for c in cols if rnd.random() < 0.8] or [cols[0]]
    if n_inst and lk_col not in eq_cols:
        eq_cols.append(lk_col)
    for c in eq_cols:
        b.enable_equality(B.COL_ADVICE, c)
    b.enable_equality(B.COL_FIXED, fz)
    if n_inst:
        b.enable_equality(B.COL_INSTANCE, 0)
    # ---- witness program, phase by phase
    values, idx = [], 0
    for c in by_phase[0]:
        for r in range(rows + 1):                           # one row more: rotation 1 of the last selected row
            b.w(B.W_INPUT, (c, r), imm=idx)
            idx += 1
            values.append(rnd.randrange(T) if c == lk_col else rnd.randrange(R) if rnd.random() < 0.7 else rnd.randrange(3))
    b.n_winputs = idx
    spare = rows + 2                                        # an unselected row for the cells that copy constraints tie
    for ph in range(1, P):
        b.next_phase()
        slot = {}
        for i in known(ph):
            slot[i] = b.scratch(len(slot))
            b.w(B.W_CHAL, slot[i], imm=i)
        base = len(slot)
        for c in by_phase[ph]:
            for r in range(rows + 1):
                nxt = [base]

                def emit(t):
                    if t[0] == "adv":
                        return (t[1], r + t[2])
                    if t[0] == "chal":
                        return slot[t[1]]
                    sc = b.scratch(nxt[0]); nxt[0] += 1
                    if t[0] == "const":
                        b.w(B.W_CONST, sc, imm=b.const_index(t[1]))
                    else:
                        x, y = emit(t[1]), emit(t[2])
                        b.w({"mul": B.W_MUL, "add": B.W_ADD, "sub": B.W_SUB}[t[0]], sc, x, y)
                    return sc
                if r < rows:
                    b.w(B.W_COPY, (c, r), emit(trees[c]))
                else:                                       # the row behind the selected ones: read by rotation 1, so assigned
                    b.w(B.W_CONST, (c, r), imm=b.const_index(rnd.randrange(50)))
            if c in eq_cols:                                # a cell of this phase tied to a cell of an earlier phase
                src = rnd.choice([e for e in earlier(ph) if e in eq_cols] or [None])
                if src is not None:
                    sr = rnd.randrange(rows)
                    b.w(B.W_COPY, (c, spare), (src, sr))
                    b.copy((B.COL_ADVICE, c, spare), (B.COL_ADVICE, src, sr))
            if c in eq_cols and rnd.random() < 0.5:         # and one to a fixed zero
                b.w(B.W_CONST, (c, spare + 1), imm=b.const_index(0))
                b.copy((B.COL_ADVICE, c, spare + 1), (B.COL_FIXED, fz, 0))
    for i in range(n_inst):
        b.instance_map.append((lk_col, i))
        b.copy((B.COL_ADVICE, lk_col, i), (B.COL_INSTANCE, 0, i))
    return (b.build_description() if describe else b.build()), k, values
