#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes of the same command
(`--pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`; no trace flags in those runs).
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section), hence the x2.   usage: pmc_traffic.py <fetch.csv> <write.csv> <out.json>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_id          # the build the counters belong to (bench.py flags a summary of another build as stale)


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("shp::", "").replace("void ", "")
        d[n][0] += 1
        d[n][1] += float(r["Counter_Value"])
    return d


f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(f, key=lambda k: -f[k][1]):
    fc, fv = f[k]
    wc, wv = w.get(k, (1, 0.0))
    out[k] = {"launches": fc, "fetch_bytes_per_launch_x2corr": fv / fc * 1024 * 2, "write_bytes_per_launch": wv / max(wc, 1) * 1024}
json.dump(dict(out, _kernel_source_id=kernel_source_id()), open(sys.argv[3], "w"), indent=1)
for k in list(out)[:12]:
    print(f"{k[:44]:44s} n={out[k]['launches']:6d} fetch {out[k]['fetch_bytes_per_launch_x2corr']/1e6:9.3f} MB  write {out[k]['write_bytes_per_launch']/1e6:8.3f} MB")
