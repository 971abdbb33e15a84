#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV of tools/prove_loop.py --lanes 1:
GPU-busy time, idle gaps between consecutive dispatches (host round trips, launch latency) and
which kernel precedes the long gaps.   python tools/trace_gaps.py <kernel_trace.csv> [skip_ms]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
skip_first_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("shp::", "")) for r in rows))
# keep the steady-state tail: everything after the last msm_multiples / table kernel
last_setup = max((i for i, e in enumerate(ev) if "multiples" in e[2] or "table" in e[2] or "g1fft" in e[2]), default=-1)
ev = ev[last_setup + 1:]
t0 = ev[0][0]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - t0
print(f"dispatches {len(ev)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms")
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = s1 - e0
    if g > 0:
        gaps[(n0, n1)][0] += g
        gaps[(n0, n1)][1] += 1
print("largest idle gaps (after kernel -> before kernel): total_us count avg_us")
for (a, b), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {a[:34]:34s} -> {b[:34]:34s} {g/1e3:9.1f} {c:4d} {g/c/1e3:8.1f}")
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    per[n][0] += e - s
    per[n][1] += 1
print("busy by kernel: total_us count avg_us")
for n, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {n[:50]:50s} {d/1e3:9.1f} {c:4d} {d/c/1e3:8.1f}")

# concurrency view (multi-lane traces): union of busy intervals and time-weighted mean overlap
pts = []
for s_, e_, _ in ev:
    pts.append((s_, 1)); pts.append((e_, -1))
pts.sort()
cur = 0; last = pts[0][0]; union = 0; area = 0
for t_, d_ in pts:
    if cur > 0:
        union += t_ - last
        area += (t_ - last) * cur
    cur += d_; last = t_
print(f"union busy {union/1e6:.3f} ms of span {span/1e6:.3f} ms ({100.0*union/span:.1f}%), mean kernels in flight while busy {area/max(union,1):.2f}")
