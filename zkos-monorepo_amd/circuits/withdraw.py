"""Withdraw-shaped stand-in circuit for the Shielder `WithdrawCircuit`.

The real constraint system lives in the private, un-vendored zkOS-circuits crate, which is not part of the
reference tree; this is a documented stand-in with the same witness
(`WithdrawProverKnowledge`, /root/reference/crates/shielder_bindings/src/circuits/withdraw.rs:88-114),
the same 8 public inputs in the same order (withdraw.rs:18-27,
/root/reference/contracts/Shielder.sol:679-701), the same hash (Poseidon2 t=8, arity 7), the same
Merkle shape (arity 7, height 13: /root/reference/crates/shielder-setup/lib.rs:3-6) and the same
engine features (degree-8 S-box gates, multi-chunk permutation argument, one lookup range
check, one instance column):

  h_nullifier_old = H(nullifier_old)                                   -> instance 1
  note_old        = H(id, nullifier_old, balance_old, token)
  note_old in path[0];  node_l = H(path[l]);  node_l in path[l+1];  node_{H-1} -> instance 0
  balance_new     = balance_old - value;  balance_new, value range-checked (lookup)
  note_new        = H(id, nullifier_new, balance_new, token)           -> instance 2
  value, token, commitment, mac_salt                                   -> instances 3..6
  mac_commitment  = H(mac_salt, id)                                    -> instance 7
"""
from .builder import (PlanBuilder, Adv, Fix, Const, COL_ADVICE, COL_FIXED, COL_INSTANCE,
                      W_INPUT, W_CONST, W_COPY, W_ADD, W_SUB, W_ADDC, W_MULC, W_POW7, W_BITS, W_SHR, R_MOD)
from . import poseidon2_params as P2

POSEIDON_ROWS = 58     # 1 input row + 4 full + 48 partial + 4 full transitions


class WithdrawLayout:
    """names the public inputs in instance order (withdraw.rs:18-27)"""
    INSTANCES = ["MerkleRoot", "HashedOldNullifier", "HashedNewNote", "WithdrawalValue", "TokenAddress",
                 "Commitment", "MacSalt", "MacCommitment"]

    def __init__(self, height, kind="withdraw"):
        assert kind in ("withdraw", "deposit")
        self.height = height
        self.kind = kind
        self.n_inputs = 4 + 7 * height + 4
        if kind == "deposit":
            # DepositInstance order (crates/shielder_bindings/src/circuits/deposit.rs:17-27,
            # contracts/Shielder.sol:504-519)
            self.INSTANCES = ["MerkleRoot", "HashedOldNullifier", "HashedNewNote", "DepositValue", "Commitment",
                              "TokenAddress", "MacSalt", "MacCommitment"]

    # witness input vector layout: the order of WithdrawCircuit::prove's arguments (withdraw.rs:88-114);
    # DepositCircuit::prove swaps the tail: value, commitment, nullifier_new, mac_salt (deposit.rs:88-114)
    ID, NULLIFIER_OLD, BALANCE_OLD, TOKEN, PATH = 0, 1, 2, 3, 4
    def path(self, level, j): return 4 + 7 * level + j
    @property
    def VALUE(self): return 4 + 7 * self.height
    @property
    def NULLIFIER_NEW(self): return (5 if self.kind == "withdraw" else 6) + 7 * self.height
    @property
    def COMMITMENT(self): return (6 if self.kind == "withdraw" else 5) + 7 * self.height
    @property
    def MAC_SALT(self): return 7 + 7 * self.height


class NewAccountLayout:
    """NewAccount-shaped stand-in: the 9 arguments of NewAccountCircuit::prove in order
    (/root/reference/crates/shielder_bindings/src/circuits/new_account.rs:108-140) and the 13 public inputs in the
    order the contract assembles them (/root/reference/contracts/Shielder.sol:345-370)"""
    INSTANCES = ["HashedNote", "Prenullifier", "InitialDeposit", "Commitment", "TokenAddress",
                 "AnonymityRevokerPublicKeyX", "AnonymityRevokerPublicKeyY", "EncryptedKeyCiphertext1X",
                 "EncryptedKeyCiphertext1Y", "EncryptedKeyCiphertext2X", "EncryptedKeyCiphertext2Y", "MacSalt",
                 "MacCommitment"]
    ID, NULLIFIER, INITIAL_DEPOSIT, COMMITMENT, TOKEN, ENCRYPTION_SALT, MAC_SALT, AR_X, AR_Y = range(9)
    kind = "new_account"
    height = 0
    n_inputs = 9


def _m4(a, b, c, d):
    ab, cd = a + b, c + d
    t2 = (b + b) + cd
    t3 = (d + d) + ab
    cd2 = cd + cd
    ab2 = ab + ab
    t4 = (cd2 + cd2) + t3
    t5 = (ab2 + ab2) + t2
    return [t3 + t5, t5, t2 + t4, t4]


def _external(s):
    x = _m4(*s[0:4]) + _m4(*s[4:8])
    w = [x[i] + x[i + 4] for i in range(4)]
    return [x[i] + w[i] for i in range(4)] + [x[i + 4] + w[i] for i in range(4)]


def build_deposit_plan(k=13, height=13, range_bits=12, range_chunks=10, describe=False):
    """Deposit-shaped stand-in (DepositProverKnowledge, crates/shielder_bindings/src/circuits/deposit.rs:88-114):
    the withdraw relation with balance_new = balance_old + value and the Deposit instance / argument order."""
    return build_withdraw_plan(k, height, range_bits, range_chunks, kind="deposit", describe=describe)


def build_new_account_plan(k=12, range_bits=11, range_chunks=11, describe=False):
    """NewAccount-shaped stand-in on the same engine (same columns, gates and lookup as the withdraw plan):
      prenullifier  = H(id)                                              -> instance 1
      note          = H(id, nullifier, initial_deposit, token)           -> instance 0
      initial_deposit range-checked, -> instance 2;  commitment, token, AR key x / y -> instances 3..6
      mac_commitment = H(mac_salt, id)                                   -> instances 11, 12
    DEVIATION: the reference encrypts a symmetric key derived from id to the anonymity revoker with in-circuit Grumpkin
    ElGamal (instances 7..10 are the two ciphertext points); that gadget lives in the un-vendored zkOS-circuits crate.
    Here the four coordinates are Poseidon2 placeholders, c1 = (H(salt, ar_x), H(salt, ar_y)), c2 = (H(id, c1x),
    H(id, c1y)): the argument vector, the 13 public inputs and their order are the reference's, the relation is not."""
    return build_withdraw_plan(k, 0, range_bits, range_chunks, kind="new_account", describe=describe)


def build_withdraw_plan(k=13, height=13, range_bits=12, range_chunks=10, kind="withdraw", describe=False):
    """-> (circuit plan, layout); describe=True: (circuit description for shp_plan_compile, layout) instead"""
    b = PlanBuilder(k)
    lay = NewAccountLayout() if kind == "new_account" else WithdrawLayout(height, kind)
    b.n_winputs = lay.n_inputs
    b.num_instance = len(lay.INSTANCES)
    s = [b.advice_column() for _ in range(8)]
    wcol, zcol = b.advice_column(), b.advice_column()
    rc = [b.fixed_column() for _ in range(8)]
    q_init, q_full, q_part, q_add, q_member, q_range, q_rsum = [b.fixed_column() for _ in range(7)]
    tab, cst = b.fixed_column(), b.fixed_column()

    cur = [Adv(c, 0) for c in s]
    nxt = [Adv(c, 1) for c in s]
    # ---- gates (order fixes the y-powers of the quotient)
    ext0 = _external(cur)
    for i in range(8):
        b.gate(Fix(q_init) * (nxt[i] - ext0[i]))
    sb = [(cur[i] + Fix(rc[i])).pow(7) for i in range(8)]
    extf = _external(sb)
    for i in range(8):
        b.gate(Fix(q_full) * (nxt[i] - extf[i]))
    t = sb[0]
    tot = t
    for i in range(1, 8):
        tot = tot + cur[i]
    b.gate(Fix(q_part) * (nxt[0] - (Const(P2.DIAG[0]) * t + tot)))
    for i in range(1, 8):
        b.gate(Fix(q_part) * (nxt[i] - (Const(P2.DIAG[i]) * cur[i] + tot)))
    b.gate(Fix(q_add) * (cur[0] + cur[1] - cur[2]))
    mem = cur[0] - cur[1]
    for j in range(2, 8):
        mem = mem * (cur[0] - cur[j])
    b.gate(Fix(q_member) * mem)
    b.gate(Fix(q_rsum) * (Adv(zcol, 0) - (Adv(wcol, 0) + Const(1 << range_bits) * Adv(zcol, 1))))
    b.lookup([Fix(q_range) * Adv(wcol, 0)], [Fix(tab)])

    for c in s:
        b.enable_equality(COL_ADVICE, c)
    b.enable_equality(COL_ADVICE, zcol)
    b.enable_equality(COL_INSTANCE, 0)
    b.enable_equality(COL_FIXED, cst)

    for r in range(1 << range_bits):
        b.fix(tab, r, r)

    state = {"row": 0, "crow": 0, "consts": {}}

    def const_cell(v):
        v %= R_MOD
        if v not in state["consts"]:
            r = state["crow"]
            state["crow"] += 1
            b.fix(cst, r, v)
            state["consts"][v] = (COL_FIXED, cst, r)
        return state["consts"][v]

    def A(col, row): return (COL_ADVICE, col, row)

    def poseidon(inputs):
        """inputs: list of ('in', idx) | ('cell', (col,row)); returns the output cell (col,row)"""
        r0 = state["row"]
        state["row"] += POSEIDON_ROWS
        arity = len(inputs)
        for i in range(8):
            dst = (s[i], r0)
            if i < arity:
                kind, v = inputs[i]
                if kind == "in":
                    b.w(W_INPUT, dst, imm=v)
                else:
                    b.w(W_COPY, dst, v)
                    b.copy(A(*dst), A(*v))
            else:
                val = (arity << 64) if i == 7 else 0
                b.w(W_CONST, dst, imm=b.const_index(val))
                b.copy(A(*dst), const_cell(val))
        S = [b.scratch(i) for i in range(40)]

        def wext(src, dst_row):
            """dst row = external(src cells); src are 8 cells, scratch 8.. used for temporaries"""
            def m4(a, bb, c, d, o):
                ab, cd, t2, t3, t4, t5 = S[16], S[17], S[18], S[19], S[20], S[21]
                b.w(W_ADD, ab, a, bb); b.w(W_ADD, cd, c, d)
                b.w(W_ADD, t2, bb, bb); b.w(W_ADD, t2, t2, cd)
                b.w(W_ADD, t3, d, d); b.w(W_ADD, t3, t3, ab)
                b.w(W_ADD, t4, cd, cd); b.w(W_ADD, t4, t4, t4); b.w(W_ADD, t4, t4, t3)
                b.w(W_ADD, t5, ab, ab); b.w(W_ADD, t5, t5, t5); b.w(W_ADD, t5, t5, t2)
                b.w(W_ADD, o[0], t3, t5); b.w(W_COPY, o[1], t5); b.w(W_ADD, o[2], t2, t4); b.w(W_COPY, o[3], t4)
            X = S[24:32]
            m4(src[0], src[1], src[2], src[3], X[0:4])
            m4(src[4], src[5], src[6], src[7], X[4:8])
            for i in range(4):
                b.w(W_ADD, S[32 + i], X[i], X[i + 4])
            for i in range(4):
                b.w(W_ADD, (s[i], dst_row), X[i], S[32 + i])
                b.w(W_ADD, (s[i + 4], dst_row), X[i + 4], S[32 + i])

        row = r0
        b.fix(q_init, row, 1)
        wext([(s[i], row) for i in range(8)], row + 1)
        row += 1
        for rnd in range(8):
            if rnd == 4:
                for pr in range(48):
                    b.fix(q_part, row, 1)
                    b.fix(rc[0], row, P2.RC_PART[pr])
                    b.w(W_ADDC, S[0], (s[0], row), imm=b.const_index(P2.RC_PART[pr]))
                    b.w(W_POW7, S[0], S[0])
                    b.w(W_COPY, S[1], S[0])
                    for i in range(1, 8):
                        b.w(W_ADD, S[1], S[1], (s[i], row))
                    b.w(W_MULC, S[2], S[0], imm=b.const_index(P2.DIAG[0]))
                    b.w(W_ADD, (s[0], row + 1), S[2], S[1])
                    for i in range(1, 8):
                        b.w(W_MULC, S[2], (s[i], row), imm=b.const_index(P2.DIAG[i]))
                        b.w(W_ADD, (s[i], row + 1), S[2], S[1])
                    row += 1
            b.fix(q_full, row, 1)
            for i in range(8):
                b.fix(rc[i], row, P2.RC_FULL[rnd][i])
                b.w(W_ADDC, S[i], (s[i], row), imm=b.const_index(P2.RC_FULL[rnd][i]))
                b.w(W_POW7, S[i], S[i])
            wext(S[0:8], row + 1)
            row += 1
        assert row == r0 + POSEIDON_ROWS - 1
        return (s[0], row), r0

    def expose(cell, idx):
        b.instance_map.append(cell)
        b.copy(A(*cell), (COL_INSTANCE, 0, idx))

    def range_check(cell):
        r0 = state["row"]
        state["row"] += range_chunks + 1
        b.w(W_COPY, (zcol, r0), cell)
        b.copy(A(zcol, r0), A(*cell))
        for i in range(range_chunks):
            b.fix(q_rsum, r0 + i, 1)
            b.fix(q_range, r0 + i, 1)
            b.w(W_BITS, (wcol, r0 + i), (zcol, r0 + i), imm=range_bits)
            b.w(W_SHR, (zcol, r0 + i + 1), (zcol, r0 + i), imm=range_bits)
        b.copy(A(zcol, r0 + range_chunks), const_cell(0))

    def finish():
        usable = b.n - (b.blinding_factors() + 1)
        assert state["row"] <= usable and (1 << range_bits) <= usable, "circuit does not fit 2^k rows"
        return (b.build_description() if describe else b.build()), lay

    if kind == "new_account":
        pre, r_pre = poseidon([("in", lay.ID)])
        id_cell = (s[0], r_pre)
        note, r_note = poseidon([("cell", id_cell), ("in", lay.NULLIFIER), ("in", lay.INITIAL_DEPOSIT), ("in", lay.TOKEN)])
        deposit_cell, token_cell = (s[2], r_note), (s[3], r_note)
        range_check(deposit_cell)
        c1x, r1 = poseidon([("in", lay.ENCRYPTION_SALT), ("in", lay.AR_X)])
        salt_cell, arx_cell = (s[0], r1), (s[1], r1)
        c1y, r2 = poseidon([("cell", salt_cell), ("in", lay.AR_Y)])
        ary_cell = (s[1], r2)
        c2x, _ = poseidon([("cell", id_cell), ("cell", c1x)])
        c2y, _ = poseidon([("cell", id_cell), ("cell", c1y)])
        mac, r_mac = poseidon([("in", lay.MAC_SALT), ("cell", id_cell)])
        mac_salt_cell = (s[0], r_mac)
        fr = state["row"]                      # a row no selector touches: the commitment is a plain public input
        state["row"] += 1
        b.w(W_INPUT, (s[3], fr), imm=lay.COMMITMENT)
        for idx, cell in enumerate([note, pre, deposit_cell, (s[3], fr), token_cell, arx_cell, ary_cell, c1x, c1y, c2x, c2y,
                                    mac_salt_cell, mac]):
            expose(cell, idx)
        return finish()

    # ---- synthesis
    h_null, r_null = poseidon([("in", lay.NULLIFIER_OLD)])
    note_old, r_note = poseidon([("in", lay.ID), ("cell", (s[0], r_null)), ("in", lay.BALANCE_OLD), ("in", lay.TOKEN)])
    id_cell, null_old_cell = (s[0], r_note), (s[1], r_note)
    bal_old_cell, token_cell = (s[2], r_note), (s[3], r_note)
    del null_old_cell
    cur_node = note_old
    for lvl in range(height):
        node, r_lvl = poseidon([("in", lay.path(lvl, j)) for j in range(7)])
        # membership row: s0 = current node, s1..s7 = the level's 7 children
        mr = state["row"]
        state["row"] += 1
        b.fix(q_member, mr, 1)
        b.w(W_COPY, (s[0], mr), cur_node)
        b.copy(A(s[0], mr), A(*cur_node))
        for j in range(7):
            b.w(W_COPY, (s[j + 1], mr), (s[j], r_lvl))
            b.copy(A(s[j + 1], mr), A(s[j], r_lvl))
        cur_node = node
    root = cur_node
    # add row, gate s0 + s1 = s2.  withdraw: balance_new + value = balance_old; deposit: balance_old + value = balance_new
    ar = state["row"]
    state["row"] += 1
    b.fix(q_add, ar, 1)
    b.w(W_INPUT, (s[1], ar), imm=lay.VALUE)
    if kind == "withdraw":
        b.w(W_COPY, (s[2], ar), bal_old_cell)
        b.copy(A(s[2], ar), A(*bal_old_cell))
        b.w(W_SUB, (s[0], ar), (s[2], ar), (s[1], ar))
        bal_new_cell, value_cell = (s[0], ar), (s[1], ar)
    else:
        b.w(W_COPY, (s[0], ar), bal_old_cell)
        b.copy(A(s[0], ar), A(*bal_old_cell))
        b.w(W_ADD, (s[2], ar), (s[0], ar), (s[1], ar))
        bal_new_cell, value_cell = (s[2], ar), (s[1], ar)

    range_check(bal_new_cell)
    range_check(value_cell)
    note_new, r_nn = poseidon([("cell", id_cell), ("in", lay.NULLIFIER_NEW), ("cell", bal_new_cell), ("cell", token_cell)])
    del r_nn
    mac, r_mac = poseidon([("in", lay.MAC_SALT), ("cell", id_cell)])
    mac_salt_cell = (s[0], r_mac)
    # commitment is a plain public input: park it in a free cell of the add row
    b.w(W_INPUT, (s[3], ar), imm=lay.COMMITMENT)
    commitment_cell = (s[3], ar)

    expose(root, 0)
    expose(h_null, 1)
    expose(note_new, 2)
    expose(value_cell, 3)
    if kind == "withdraw":
        expose(token_cell, 4)
        expose(commitment_cell, 5)
    else:
        expose(commitment_cell, 4)
        expose(token_cell, 5)
    expose(mac_salt_cell, 6)
    expose(mac, 7)
    return finish()


def new_account_witness_inputs(lay, rnd, initial_deposit=10**18):
    """a witness input vector of the NewAccount-shaped plan (canonical ints); nothing to keep consistent: every
    public input is computed by the witness program"""
    from .builder import R_MOD as R
    return [rnd.randrange(R), rnd.randrange(R), initial_deposit, rnd.randrange(R), rnd.randrange(1 << 160), rnd.randrange(R),
            rnd.randrange(R), rnd.randrange(R), rnd.randrange(R)]


def withdraw_witness_inputs(lay, hash_fn, rnd, balance_old=10**18, value=10**17):
    """`random_correct_example`: a consistent witness input vector (canonical ints).
    hash_fn(list of ints) -> int is any Poseidon2-t8 implementation (tests pass the oracle's,
    bench.py the library's host hash)."""
    ident, n_old, n_new = (rnd.randrange(R_MOD) for _ in range(3))
    token, commitment, salt = rnd.randrange(1 << 160), rnd.randrange(R_MOD), rnd.randrange(R_MOD)
    note = hash_fn([ident, n_old, balance_old, token])
    path, node = [], note
    for _ in range(lay.height):
        sib = [rnd.randrange(R_MOD) for _ in range(7)]
        sib[rnd.randrange(7)] = node
        path.append(sib)
        node = hash_fn(sib)
    tail = [value, n_new, commitment, salt] if lay.kind == "withdraw" else [value, commitment, n_new, salt]
    flat = [ident, n_old, balance_old, token] + [x for lv in path for x in lv] + tail
    assert len(flat) == lay.n_inputs
    return flat
