#!/usr/bin/env python3
"""Prove K withdraw-shaped proofs back to back on one GPU (profiling driver for rocprofv3).
   python tools/prove_loop.py [--k 13] [--proofs 10] [--lanes 1]"""
import argparse, importlib, os, random, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shp = importlib.import_module("zkos-monorepo_amd")
shp.set_hw_queues(int(os.environ.get("SHP_BENCH_HW_QUEUES", "22")))
circuits = importlib.import_module("zkos-monorepo_amd.circuits")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=10)
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--throughput", action="store_true", help="shp_prover_mode(1): the scheduling a pool lane uses")
    a = ap.parse_args()
    ctx = shp.Context(0)
    t0 = time.time()
    if a.small:
        plan, lay = circuits.build_withdraw_plan(k=9, height=1, range_bits=4, range_chunks=8)
        k = 9
    else:
        plan, lay = circuits.build_withdraw_plan()
        k = 13
    t1 = time.time()
    params = shp.Params.synthetic(ctx, k, 0xDEADBEEF12345678)
    t2 = time.time()
    pk = shp.ProvingKey(ctx, params, plan)
    t3 = time.time()
    print(f"plan {t1-t0:.3f}s srs {t2-t1:.3f}s keygen {t3-t2:.3f}s", flush=True)
    rnd = random.Random(1)
    bal, val = (10**18, 10**17) if not a.small else (2**31, 2**20)
    wits = [b"".join(int(x).to_bytes(32, "little") for x in
                     circuits.withdraw_witness_inputs(lay, shp.poseidon2_hash_host, rnd, bal, val)) for _ in range(4)]
    provers = [shp.Prover(pk) for _ in range(a.lanes)]
    if a.throughput:
        for pr in provers:
            pr.mode(True)
    provers[0].prove_bytes(wits[0], bytes(32))          # warm-up
    lat = []

    def lane(pr, count, out):
        for i in range(count):
            s = time.perf_counter()
            pr.prove_bytes(wits[i % 4], bytes([i % 251] * 32))
            out.append(time.perf_counter() - s)

    t0 = time.perf_counter()
    ths, outs = [], []
    for pr in provers:
        o = []
        outs.append(o)
        th = threading.Thread(target=lane, args=(pr, a.proofs, o))
        th.start()
        ths.append(th)
    for th in ths:
        th.join()
    el = time.perf_counter() - t0
    for o in outs:
        lat += o
    lat.sort()
    print(f"lanes {a.lanes} proofs {len(lat)} wall {el:.3f}s -> {len(lat)/el:.1f} proofs/s; "
          f"latency p50 {lat[len(lat)//2]*1e3:.2f} ms min {lat[0]*1e3:.2f} max {lat[-1]*1e3:.2f}", flush=True)


if __name__ == "__main__":
    main()
