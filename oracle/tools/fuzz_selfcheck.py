#!/usr/bin/env python3
"""The oracle against itself on the fuzz generator's plans, CPU only: for every seed in [first, last) the oracle's prover
must either refuse the plan (a commitment at infinity: halo2's transcript refuses it) or produce a proof that the oracle's
verifier -- the restatement pinned to the reference's contract text by tests/golden/verifier_trace.json -- accepts.
   python oracle/tools/fuzz_selfcheck.py FIRST LAST [SECONDS]
Prints the seeds whose proof was rejected (none expected) and the refused ones."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                   # noqa: E402
from oracle import plonk as PL                  # noqa: E402
import test_gpu_fuzz_plans as T                 # noqa: E402  (the plan generator; nothing GPU-side runs at import)


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    limit = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    t0, bad, refused, seed = time.time(), [], [], first
    for seed in range(first, last):
        plan_bytes, k, values = T.random_plan(seed)
        plan = PL.Plan(plan_bytes)
        osrs = oracle.srs_synthetic(k, T.TAU)
        opk = PL.keygen(plan, osrs)
        try:
            proof, inst = PL.prove(opk, osrs, values, bytes([seed % 251 + 1]) * 32)
        except PL.PointAtInfinity:
            refused.append(seed)
            continue
        if not PL.verify(opk, proof, inst, tau=T.TAU):
            bad.append(seed)
            print("rejected:", seed, flush=True)
        if time.time() - t0 > limit:
            break
    print(f"seeds {first}..{seed}: rejected {bad}, refused (commitment at infinity) {refused}, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
