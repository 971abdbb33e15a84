#!/usr/bin/env python3
"""How full is the GPU while the proving pool runs?  Reads a rocprofv3 --kernel-trace CSV (one row per dispatch with
start / end timestamps, grid and workgroup sizes) and integrates, over the busiest stretch of the trace, the number of
kernels in flight and the waves they ask for against the chip's wave slots.
   python tools/occupancy_timeline.py <kernel_trace.csv> [out.json] [--slots 3072]"""
import csv, json, sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    slots = 3072
    if "--slots" in sys.argv:
        slots = int(sys.argv[sys.argv.index("--slots") + 1])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            threads = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("shp::", "")
            rows.append((s, e, (threads + 63) // 64, wg, name, r.get("Queue_Id", "")))
    rows.sort()
    t_min, t_max = rows[0][0], max(r[1] for r in rows)
    # steady state: the densest stretch -- the proofs of the timed steps -- is where dispatches start closest together;
    # take the window between the 35th and the 90th percentile of dispatch starts
    starts = [r[0] for r in rows]
    w0, w1 = starts[int(0.35 * len(starts))], starts[int(0.90 * len(starts))]
    ev = []
    for s, e, waves, wg, name, q in rows:
        s2, e2 = max(s, w0), min(e, w1)
        if e2 <= s2:
            continue
        ev.append((s2, 1, waves))
        ev.append((e2, -1, waves))
    ev.sort()
    hist_k = defaultdict(int)       # ns with k kernels in flight
    demand_ns = 0                   # integral of min(slots, waves asked for)
    under = defaultdict(int)        # ns with demand below 25 / 50 / 100 % of the slots
    k = w = 0
    last = w0
    for t, d, waves in ev:
        dt = t - last
        if dt > 0:
            hist_k[min(k, 20)] += dt
            demand_ns += min(slots, w) * dt
            for frac in (0.25, 0.5, 1.0):
                if w < frac * slots:
                    under[frac] += dt
        last = t
        k += d
        w += d * waves
    hist_k[min(k, 20)] += max(0, w1 - last)
    span = w1 - w0
    by_name = defaultdict(lambda: [0, 0, 0])
    queues = set()
    for s, e, waves, wg, name, q in rows:
        if s >= w0 and e <= w1:
            b = by_name[name]
            b[0] += 1
            b[1] += e - s
            b[2] += waves
            queues.add(q)
    res = {
        "trace_ms": (t_max - t_min) / 1e6, "window_ms": span / 1e6, "dispatches_in_window": sum(b[0] for b in by_name.values()),
        "queues_seen": len(queues), "wave_slots": slots,
        "kernels_in_flight_time_share": {str(kk): hist_k[kk] / span for kk in sorted(hist_k)},
        "mean_kernels_in_flight": sum(kk * v for kk, v in hist_k.items()) / span,
        "mean_wave_demand_over_slots": demand_ns / span / slots,
        "time_share_with_demand_below": {str(fr): under[fr] / span for fr in (0.25, 0.5, 1.0)},
        "per_kernel": {n: {"calls": b[0], "sum_ms": b[1] / 1e6, "mean_us": b[1] / b[0] / 1e3, "mean_waves": b[2] / b[0]}
                       for n, b in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:16]},
    }
    text = json.dumps(res, indent=1)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
