#!/usr/bin/env python3
"""The dispatches of the LAST proof of a one-lane rocprofv3 --kernel-trace run, in order: name, waves, duration and the gap
to the previous dispatch's end.   python tools/dispatch_sequence.py <kernel_trace.csv> [last_kernel_of_a_proof]"""
import csv, sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            threads = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("shp::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (threads + 63) // 64, name))
    rows.sort()
    # the last proof: everything after the last SHPLONK witness kernel (the end of a proof) but one
    first = sys.argv[2] if len(sys.argv) > 2 else "shplonk_wp_inv_kernel"
    ends = [i for i, r in enumerate(rows) if r[3].startswith(first)]
    begin = ends[-2] + 1 if len(ends) > 1 else 0
    while rows[begin][3].startswith("msm_direct"):              # the commitment of W' still belongs to the proof before
        begin += 1
    prev_end = rows[begin][0]
    total = 0
    for s, e, waves, name in rows[begin:]:
        print(f"{name:36s} waves {waves:6d}  {(e - s) / 1e3:8.1f} us   gap {(s - prev_end) / 1e3:8.1f} us")
        prev_end = max(prev_end, e)
        total += e - s
    print(f"{len(rows) - begin} dispatches, kernel time {total / 1e3:.1f} us, span {(prev_end - rows[begin][0]) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
