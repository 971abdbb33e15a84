#!/usr/bin/env python3
"""Ahead-of-time gate kernels for the circuits the library ships plans for.

The reference fixes its circuits at build time (its build script generates parameters and keys for Deposit, Withdraw and
NewAccount: /root/reference/crates/shielder_bindings/build.rs:19-48).  The same holds for the gate expressions of a
plan: they are data, but they change only when the circuit does.  This tool turns the gate program of each shipped
plan into straight-line HIP -- one statement per program instruction, slots as local variables, queries and constants as
literals -- and writes zkos-monorepo_amd/csrc/gate_kernels.inc with a registry keyed by a hash of the program.  A key
whose gate program is in the registry launches the compiled kernel; every other plan runs the interpreter
(prog_fold_reg_kernel / prog_fold_kernel), whose results are identical.

   python tools/gen_gate_kernels.py            # rewrites the file
   python tools/gen_gate_kernels.py --check    # exit 1 if the committed file is not what the plans generate
"""
import importlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "zkos-monorepo_amd", "csrc", "gate_kernels.inc")

OP_CONST, OP_ADV, OP_FIX, OP_INST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_OUT = range(9)


def parse_plan(data):
    """the pieces of a circuit plan (layout: zkos-monorepo_amd/circuits/builder.py) the gate program depends on"""
    assert data[:4] == b"SHPL"
    h = struct.unpack_from("<20I", data, 4)
    naq, nfq, niq, npc, ncst, ngi, slots = h[7], h[8], h[9], h[10], h[11], h[12], h[13]
    off = 84

    def pairs(cnt, fmt):
        nonlocal off
        out = [struct.unpack_from(fmt, data, off + 8 * i) for i in range(cnt)]
        off += 8 * cnt
        return out

    adv_q, fix_q, inst_q = pairs(naq, "<Ii"), pairs(nfq, "<Ii"), pairs(niq, "<Ii")
    off += 8 * npc + 32 * ncst
    prog = [struct.unpack_from("<4I", data, off + 16 * i) for i in range(ngi)]
    return slots, prog, adv_q, fix_q, inst_q


def program_hash(slots, prog, adv_q, fix_q, inst_q):
    """FNV-1a (64 bit) over what the generated code depends on; csrc/prover.hip computes the same value at keygen"""
    h = 0xCBF29CE484222325

    def eat(word):
        nonlocal h
        for b in struct.pack("<I", word & 0xFFFFFFFF):
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF

    eat(slots)
    eat(len(prog))
    for ins in prog:
        for w in ins:
            eat(w)
    for qs in (adv_q, fix_q, inst_q):
        eat(len(qs))
        for col, rot in qs:
            eat(col)
            eat(rot)
    return h


def selector_runs(prog, fix_q):
    """Gates that are `selector * expression` with the SAME selector query in consecutive OUT instructions:
    acc = (..((acc y + s g1) y + s g2)..) y + s gk  ==  acc y^k + s (g1 y^(k-1) + .. + gk)
    -- k + 2 products instead of 2 k.  Returns {OUT index: (run id, position, run length, MUL index, expr slot, query)}
    for every OUT that takes part in a run of at least two; exact field arithmetic, so the values are the same."""
    last_write = {}
    info = {}                                   # OUT index -> (mul index, expr slot, selector query) when the pattern holds
    reads_after = {}
    for i, (op, dst, a, b) in enumerate(prog):
        if op == OP_OUT:
            w = last_write.get(a)
            if w is not None and prog[w][0] == OP_MUL:
                _, _, ma, mb = prog[w]
                for sel_slot, expr_slot in ((ma, mb), (mb, ma)):
                    lw = max((j for j in range(w) if prog[j][1] == sel_slot and prog[j][0] != OP_OUT), default=None)
                    if lw is not None and prog[lw][0] == OP_FIX and sel_slot != expr_slot:
                        # the product must feed this OUT alone: no other read of its slot until the slot is rewritten
                        other = False
                        for j in range(w + 1, len(prog)):
                            o, d, x, y = prog[j]
                            rd = (x,) if o in (OP_NEG, OP_OUT) else (x, y) if o in (OP_ADD, OP_SUB, OP_MUL) else ()
                            if j != i and a in rd:
                                other = True
                            if o != OP_OUT and d == a and j > i:
                                break
                        if not other:
                            info[i] = (w, expr_slot, fix_q[prog[lw][2]])
                        break
        else:
            last_write[dst] = i
    outs = [i for i, ins in enumerate(prog) if ins[0] == OP_OUT]
    runs, cur = [], []
    for i in outs:
        if i in info and cur and info[cur[-1]][2] == info[i][2]:
            cur.append(i)
        else:
            if len(cur) >= 2:
                runs.append(cur)
            cur = [i] if i in info else []
    if len(cur) >= 2:
        runs.append(cur)
    plan = {}
    for rid, run in enumerate(runs):
        for pos, i in enumerate(run):
            plan[i] = (rid, pos, len(run)) + info[i]
    return plan


def emit_kernel(name, slots, prog, adv_q, fix_q, inst_q):
    def row(rot):
        if rot == 0:
            return "row"
        return "((row & a.hi_mask) | ((row + (uint32_t)(%d * (int32_t)a.rot_step)) & a.size_mask))" % rot

    runs = selector_runs(prog, fix_q)
    elided = {v[3]: i for i, v in runs.items()}                 # MUL index -> its OUT
    lengths = sorted({v[2] for v in runs.values()})
    L = ["__global__ __launch_bounds__(PROG_THREADS) void %s(ProgArgs a) {" % name,
         "    const uint32_t row = blockIdx.x * PROG_THREADS + threadIdx.x;",
         "    Fr acc = a.acc_in ? ld(a.acc_in + row) : Fr::zero();",
         "    Fr " + ", ".join("s%d = Fr::zero()" % i for i in range(slots)) + ";"]
    if runs:
        L.append("    // runs of gates under one selector: acc y^k + selector * (g1 y^(k-1) + .. + gk), k + 2 products for 2 k")
        L.append("    Fr inner = Fr::zero();")
        # y^k by an addition chain written out here (Fr::pow_u64 walks a 256-bit exponent)
        have = {1: "a.mult"}

        def power(k):
            if k in have:
                return have[k]
            h = power(k // 2)
            L.append("    const Fr ysq%d = %s * %s;" % (k - k % 2, h, h))
            have[k - k % 2] = "ysq%d" % (k - k % 2)
            if k % 2:
                L.append("    const Fr yod%d = ysq%d * a.mult;" % (k, k - 1))
                have[k] = "yod%d" % k
            return have[k]

        for k in lengths:
            L.append("    const Fr ypow%d = %s;" % (k, power(k)))
    for i, (op, dst, a, b) in enumerate(prog):
        if op == OP_CONST:
            L.append("    s%d = ld(a.consts + %d);" % (dst, a))
        elif op == OP_ADV:
            L.append("    s%d = ld(a.adv_base + (size_t)%d * a.adv_stride + %s);" % (dst, adv_q[a][0], row(adv_q[a][1])))
        elif op == OP_FIX:
            L.append("    s%d = ld(a.fix_base + (size_t)%d * a.fix_stride + %s);" % (dst, fix_q[a][0], row(fix_q[a][1])))
        elif op == OP_INST:
            L.append("    s%d = ld(a.inst_base + %s);" % (dst, row(inst_q[a][1])))
        elif op == OP_ADD:
            L.append("    s%d = s%d + s%d;" % (dst, a, b))
        elif op == OP_SUB:
            L.append("    s%d = s%d - s%d;" % (dst, a, b))
        elif op == OP_MUL:
            if i in elided:
                L.append("    const Fr gate%d = s%d;" % (elided[i], runs[elided[i]][4]))   # the selector is applied once per run
            else:
                L.append("    s%d = s%d * s%d;" % (dst, a, b))
        elif op == OP_NEG:
            L.append("    s%d = Fr::neg(s%d);" % (dst, a))
        elif op == OP_OUT:
            if i in runs:
                rid, pos, length, w, expr_slot, (col, rot) = runs[i]
                L.append(("    inner = gate%d;" if pos == 0 else "    inner = inner * a.mult + gate%d;") % i)
                if pos == length - 1:
                    L.append("    acc = acc * ypow%d + ld(a.fix_base + (size_t)%d * a.fix_stride + %s) * inner;" % (length, col, row(rot)))
            else:
                L.append("    acc = acc * a.mult + s%d;" % a)
        else:
            raise ValueError(op)
    L += ["    st(a.out + row, acc);", "}"]
    return "\n".join(L)


def generate():
    circuits = importlib.import_module("zkos-monorepo_amd.circuits")
    plans = [("withdraw", circuits.build_withdraw_plan()[0]),
             ("deposit", circuits.build_deposit_plan()[0]),
             ("new_account", circuits.build_new_account_plan()[0])]
    seen = {}
    for label, plan in plans:
        parts = parse_plan(plan)
        h = program_hash(*parts)
        seen.setdefault(h, (label, parts))
        if seen[h][0] != label:
            seen[h] = (seen[h][0] + ", " + label, parts)
    out = ["// GENERATED by tools/gen_gate_kernels.py from the gate programs of the shipped circuit plans -- do not edit.",
           "// One straight-line kernel per distinct program (slots are local variables, queries and constant indices are",
           "// literals) and the registry prove_impl consults: a key whose gate program hashes to an entry launches the",
           "// compiled kernel instead of the interpreter.  Included by prover.hip inside namespace shp.", ""]
    reg = []
    for h, (label, parts) in sorted(seen.items()):
        name = "gate_kernel_%016x" % h
        out.append("// %s: %d instructions, %d slots" % (label, len(parts[1]), parts[0]))
        out.append(emit_kernel(name, *parts))
        out.append("")
        reg.append("    {0x%016xull, %du, %s}," % (h, len(parts[1]), name))
    out.append("struct GateKernelEntry { uint64_t hash; uint32_t n_instr; void (*fn)(ProgArgs); };")
    out.append("static const GateKernelEntry GATE_KERNELS[] = {")
    out += reg
    out.append("};")
    return "\n".join(out) + "\n"


def main():
    text = generate()
    if "--check" in sys.argv:
        same = os.path.exists(OUT) and open(OUT).read() == text
        print("gate_kernels.inc is %s" % ("current" if same else "STALE: run tools/gen_gate_kernels.py"))
        sys.exit(0 if same else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, "(%d bytes)" % len(text))


if __name__ == "__main__":
    main()
