#!/usr/bin/env python3
"""Ordered kernel timeline of the LAST proof in a rocprofv3 --kernel-trace CSV of tools/prove_loop.py --lanes 1:
start offset, duration, gap to the previous dispatch.   python tools/trace_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("shp::", "").replace("void ", "")) for r in rows)
# a proof starts at its chacha_fr_kernel (random polynomial, first launch of P2)
starts = [i for i, e in enumerate(ev) if e[2].startswith("chacha_fr_kernel")]
lo = starts[-1]
# include the uploads just before
while lo > 0 and "copyBuffer" in ev[lo - 1][2] and ev[lo][0] - ev[lo - 1][1] < 200000:
    lo -= 1
t0 = ev[lo][0]
prev_end = t0
for s, e, n in ev[lo:]:
    print(f"{(s - t0)/1e3:9.1f} us  +{(s - prev_end)/1e3:7.1f} gap  {(e - s)/1e3:8.1f} us  {n}")
    prev_end = e
print(f"total span {(ev[-1][1] - t0)/1e3:.1f} us")
