#!/usr/bin/env python3
"""Soak: N proofs through a 16-lane pool, EVERY proof checked by the native verifier (pairing included) and the public
inputs against shp_public_inputs.   python tools/soak.py [--proofs 20000] [--lanes 16]"""
import argparse, importlib, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shp = importlib.import_module("zkos-monorepo_amd")
circuits = importlib.import_module("zkos-monorepo_amd.circuits")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=20000)
    ap.add_argument("--lanes", type=int, default=16)
    a = ap.parse_args()
    ctx = shp.Context(0)
    plan, lay = circuits.build_withdraw_plan()
    params = shp.Params.synthetic(ctx, 13, 0xDEADBEEF12345678)
    pk = shp.ProvingKey(ctx, params, plan)
    pool = shp.Pool(pk, a.lanes, queue_cap=4 * a.lanes)
    rnd = random.Random(9)
    wits = [circuits.withdraw_witness_inputs(lay, shp.poseidon2_hash_host, rnd) for _ in range(16)]
    blobs = [b"".join(int(x).to_bytes(32, "little") for x in w) for w in wits]
    pubs = [pk.public_inputs(w) for w in wits]
    bad = [0]

    def one(i):
        t = None
        while t is None:
            try:
                t = pool.submit(blobs[i % 16], i.to_bytes(32, "little"))
            except shp.ShpError as e:
                if e.code != 7:
                    raise
                time.sleep(0.001)
        proof, inst = pool.wait(t)
        if inst != pubs[i % 16] or not pk.verify(proof, inst):
            bad[0] += 1

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=3 * a.lanes) as ex:
        list(ex.map(one, range(a.proofs)))
    el = time.time() - t0
    print(f"soak: {a.proofs} proofs in {el:.1f} s ({a.proofs / el:.0f}/s incl. verification), {bad[0]} bad")
    pool.close(); pk.close(); params.close()
    sys.exit(1 if bad[0] else 0)


if __name__ == "__main__":
    main()
