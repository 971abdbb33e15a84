#!/usr/bin/env python3
"""Inputs for, and the check of, examples/shp_cli.cpp (the C ABI driven by a plain C++ host, no Python/torch
in the proving process).  CPU only -- this script never touches the GPU, so that the CLI can be started from the
same shell without an exec out of a GPU-initialised process:

  python tools/cli_check.py gen  /tmp/cli       # plan.bin, witnesses.bin (5 withdraw-shaped witnesses, k = 9)
  zkos-monorepo_amd/shp_cli /tmp/cli/plan.bin /tmp/cli/witnesses.bin 9 1234567890abcdef1234567890abcdef55 3 /tmp/cli/proofs.bin
  python tools/cli_check.py check /tmp/cli      # proofs.bin == the CPU oracle's proofs, byte for byte
(test infrastructure: imports the oracle)"""
import importlib
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "8")      # the oracle's tiny k = 9 vectors crawl on 256 OpenMP threads
TAU = 0x1234567890ABCDEF1234567890ABCDEF55
COUNT = 5


def material():
    import oracle as O
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    plan_bytes, lay = C.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
    H = lambda xs: O.fr_ints(O.poseidon2_hash(O.fr_array(xs)))[0]
    rnd = random.Random(99)
    wits = [C.withdraw_witness_inputs(lay, H, rnd, 2**31 + i, 2**20 + i) for i in range(COUNT)]
    return O, plan_bytes, wits


def main():
    cmd, d = sys.argv[1], sys.argv[2]
    O, plan_bytes, wits = material()
    if cmd == "gen":
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "plan.bin"), "wb").write(plan_bytes)
        open(os.path.join(d, "witnesses.bin"), "wb").write(b"".join(int(x).to_bytes(32, "little") for w in wits for x in w))
        print("wrote", d)
        return
    from oracle import plonk as PL
    plan = PL.Plan(plan_bytes)
    srs = O.srs_synthetic(9, TAU)
    opk = PL.keygen(plan, srs)
    got = open(os.path.join(d, "proofs.bin"), "rb").read()
    plen = PL.proof_layout(plan)[2]
    assert len(got) == plen * COUNT, (len(got), plen)
    for i, w in enumerate(wits):
        want, inst = PL.prove(opk, srs, w, bytes([i + 1]) * 32)
        assert got[i * plen:(i + 1) * plen] == want, f"proof {i} differs from the oracle"
        assert PL.verify(opk, want, inst, tau=TAU)
    print(f"cli_check ok: {COUNT} proofs from the C++ host are byte-identical to the oracle's")


if __name__ == "__main__":
    main()
