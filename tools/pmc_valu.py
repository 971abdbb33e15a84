#!/usr/bin/env python3
"""Per-kernel VALU instruction counts per proof from a rocprofv3 `--pmc SQ_INSTS_VALU SQ_WAVES` pass of
tools/prove_loop.py (counter collection serialises dispatches, so the figures are per launch and exact).
Setup kernels (SRS powers, G1 FFT, multiples tables) are left out; keygen's few launches of the proving kernels
are not, so run enough proofs (>= 12) for their share to vanish.
usage: pmc_valu.py <counter_collection.csv> <proofs in the run, warm-up included> [out.json]"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_id          # the build the counters belong to (bench.py flags a summary of another build as stale)
SETUP = ("msm_multiples_kernel", "g1fft_", "msm_table_kernel", "srs_powers_kernel", "diag_", "lagrange_")
d = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("shp::", "").replace("void ", "")
    if n.startswith(SETUP):
        continue
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        d[n][0] += 1
        d[n][1] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "SQ_WAVES":
        d[n][2] += float(r["Counter_Value"])
proofs = float(sys.argv[2])
tot = sum(v[1] for v in d.values())
print(f"{'kernel':40s} {'launches/proof':>14s} {'VALU inst/proof':>16s} {'share':>6s} {'waves/proof':>12s}")
for k, (c, v, w) in sorted(d.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k[:40]:40s} {c/proofs:14.1f} {v/proofs:16.3e} {100*v/tot:5.1f}% {w/proofs:12.0f}")
# 4 cycles per wave64 VALU instruction, 1024 SIMDs, 2.4 GHz
print(f"total {tot/proofs:.3e} wave-instructions per proof -> {tot/proofs*4/1024/2.4e9*1e6:.0f} us per proof with every SIMD issuing every 4 cycles")
if len(sys.argv) > 3:
    json.dump({"_kernel_source_id": kernel_source_id(), "proofs": proofs, "valu_wave_instructions_per_proof": tot / proofs,
               "per_kernel": {k: v[1] / proofs for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])}},
              open(sys.argv[3], "w"), indent=1)
