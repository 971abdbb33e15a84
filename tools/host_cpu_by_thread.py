#!/usr/bin/env python3
"""Which host threads does a proving pool keep busy?  Runs batches through a 48-lane pool for a few seconds and prints the
CPU time of every thread of the process (from /proc/self/task), busiest first.
   python tools/host_cpu_by_thread.py [--lanes 48] [--seconds 4]"""
import argparse, importlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def thread_times():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        comm = f[f.index("(") + 1:f.rindex(")")]
        rest = f[f.rindex(")") + 2:].split()
        out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=48)
    ap.add_argument("--seconds", type=float, default=4.0)
    a = ap.parse_args()
    shp = importlib.import_module("zkos-monorepo_amd")
    shp.set_hw_queues(int(os.environ.get("SHP_BENCH_HW_QUEUES", "22")))
    circuits = importlib.import_module("zkos-monorepo_amd.circuits")
    shp.prefer_sleeping_waits(a.lanes)
    ctx = shp.Context(0)
    plan, lay = circuits.build_withdraw_plan()
    params = shp.Params.synthetic(ctx, 13, 0xDEADBEEF12345678)
    pk = shp.ProvingKey(ctx, params, plan)
    pool = shp.Pool(pk, a.lanes)
    rnd = random.Random(3)
    count = a.lanes * 8
    wits = [circuits.withdraw_witness_inputs(lay, shp.poseidon2_hash_host, rnd) for _ in range(8)]
    blob = b"".join(b"".join(int(x).to_bytes(32, "little") for x in wits[i % 8]) for i in range(count))
    seeds = b"".join(i.to_bytes(32, "little") for i in range(count))
    pool.prove_batch(blob, seeds, count)
    before, t0, proofs = thread_times(), time.perf_counter(), 0
    while time.perf_counter() - t0 < a.seconds:
        pool.prove_batch(blob, seeds, count)
        proofs += count
    el = time.perf_counter() - t0
    after = thread_times()
    rows = sorted(((after[t][1] - before.get(t, (None, 0.0))[1], after[t][0], t) for t in after), reverse=True)
    total = sum(r[0] for r in rows)
    print(f"{proofs / el:.0f} proofs/s, {total / el:.2f} cores busy, {1e3 * total / proofs:.2f} ms of host CPU per proof, {len(rows)} threads")
    for cpu, comm, tid in rows[:12]:
        print(f"  {comm:<20s} tid {tid:<8d} {cpu / el:5.2f} cores")
    by = {}
    for cpu, comm, tid in rows:
        by[comm] = by.get(comm, 0.0) + cpu
    print("  by name:", ", ".join(f"{k} {v / el:.2f}" for k, v in sorted(by.items(), key=lambda kv: -kv[1])))
    pool.close(); pk.close(); params.close()


if __name__ == "__main__":
    main()
