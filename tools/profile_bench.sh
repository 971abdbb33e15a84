#!/bin/bash
# Round profile on the MI355X box (run through gpurun from the repo root):
#   tools/profile_bench.sh <tag>      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# Separate rocprofv3 runs: kernel trace + stats of the bench command, of a ONE-lane proving loop (per-kernel times
# without other lanes in the way) and of the standalone MSM / NTT sizes; then FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU
# counter passes (no trace flags with --pmc).  Every step has its own time limit; the first failure ends the script.
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/${tag}_stats $out/${tag}_lane1 $out/${tag}_micro $out/${tag}_fetch $out/${tag}_write $out/${tag}_valu
echo "[profile] bench kernel trace"; 
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 5 --warmup 1 --cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --micro > $out/${tag}_stats.log 2>&1
cp $(ls $out/${tag}_stats/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
echo "[profile] one-lane proving loop kernel trace"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_lane1 -- python3 tools/prove_loop.py --proofs 40 --lanes 1 --throughput > $out/${tag}_lane1.log 2>&1
cp $(ls $out/${tag}_lane1/*/*kernel_stats.csv | head -1) $out/${tag}_prove_lane1_kernel_stats.csv
echo "[profile] standalone MSM / NTT kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_micro -- python3 tools/microbench.py --ntt 20 22 24 --msm 20 22 24 --reps 5 > $out/${tag}_microbench.log 2>&1
cp $(ls $out/${tag}_micro/*/*kernel_stats.csv | head -1) $out/${tag}_microbench_kernel_stats.csv
echo "[profile] FETCH_SIZE"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-proofs 0 --latency-proofs 2 --interpreted-steps 0 --assigned-steps 0 --micro > $out/${tag}_fetch.log 2>&1
echo "[profile] WRITE_SIZE"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --steps 2 --warmup 1 --cpu-proofs 0 --latency-proofs 2 --interpreted-steps 0 --assigned-steps 0 --micro > $out/${tag}_write.log 2>&1
python3 tools/pmc_traffic.py $(ls $out/${tag}_fetch/*/*counter_collection.csv | head -1) $(ls $out/${tag}_write/*/*counter_collection.csv | head -1) $out/${tag}_pmc_traffic_bench.json > $out/${tag}_pmc_traffic.txt
echo "[profile] SQ_INSTS_VALU"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/${tag}_valu -- python3 tools/prove_loop.py --proofs 15 --throughput > $out/${tag}_valu.log 2>&1
python3 tools/pmc_valu.py $(ls $out/${tag}_valu/*/*counter_collection.csv | head -1) 16 $out/${tag}_pmc_valu_prove.json > $out/${tag}_pmc_valu_prove.txt
# the raw dumps are large: keep the summaries only
rm -rf $out/${tag}_stats $out/${tag}_lane1 $out/${tag}_micro $out/${tag}_fetch $out/${tag}_write $out/${tag}_valu
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-120
head -8 $out/${tag}_prove_lane1_kernel_stats.csv | cut -c1-120
head -12 $out/${tag}_microbench_kernel_stats.csv | cut -c1-120
head -6 $out/${tag}_pmc_traffic.txt
tail -1 $out/${tag}_pmc_valu_prove.txt
