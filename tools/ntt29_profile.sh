# Standalone NTT in both tile arithmetics (SHP_NTT29=0/1): kernel durations (rocprofv3 --kernel-trace --stats) and, in a
# run of its own, instruction and wait counters
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
for m in 0 1; do
  export SHP_NTT29=$m
  rm -rf $out/nttk_$m $out/nttq_$m
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/nttk_$m -- python3 tools/microbench.py --ntt 13 20 24 --msm --reps 5 > $out/nttk_$m.log 2>&1
  cp $(ls $out/nttk_$m/*/*kernel_stats.csv | head -1) $out/r03_ntt29_mode${m}_kernel_stats.csv
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/nttq_$m -- python3 tools/microbench.py --ntt 13 24 --msm --reps 1 > $out/nttq_$m.log 2>&1
  cp $(ls $out/nttq_$m/*/*counter_collection.csv | head -1) $out/r03_ntt29_mode${m}_pmc.csv
  rm -rf $out/nttk_$m $out/nttq_$m
done
python3 - <<'PY'
import csv, collections
for m in (0,1):
    print("SHP_NTT29 =", m)
    for r in csv.DictReader(open(f"gpurun_out/r03_ntt29_mode{m}_kernel_stats.csv")):
        if "pass_kernel" in r["Name"]: print("   %-50s calls %s  average %.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
    d=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f"gpurun_out/r03_ntt29_mode{m}_pmc.csv")):
        if "pass_kernel" not in r["Kernel_Name"]: continue
        key=(r["Kernel_Name"].split("(")[0][-32:], int(r["Grid_Size"]))
        d[key][r["Counter_Name"]]+=float(r["Counter_Value"]);
        if r["Counter_Name"]=="SQ_WAVES": cnt[key]+=1
    for k,v in sorted(d.items()): print("   per launch", k, {a: "%.3e"%(b/cnt[k]) for a,b in sorted(v.items())})
PY
grep ntt_log_n $out/nttk_0.log $out/nttk_1.log
