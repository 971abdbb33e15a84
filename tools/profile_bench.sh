#!/bin/bash
# Round profile of the bench command on the MI355X box (run through gpurun from the repo root):
#   tools/profile_bench.sh <tag>      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# Three separate rocprofv3 runs: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE (no trace flags with --pmc).
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write $out/${tag}_valu
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 5 --warmup 1 --cpu-proofs 0 > $out/${tag}_stats.log 2>&1
cp $(ls $out/${tag}_stats/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-proofs 0 --latency-proofs 2 > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --steps 2 --warmup 1 --cpu-proofs 0 --latency-proofs 2 > $out/${tag}_write.log 2>&1
python3 tools/pmc_traffic.py $(ls $out/${tag}_fetch/*/*counter_collection.csv | head -1) $(ls $out/${tag}_write/*/*counter_collection.csv | head -1) $out/${tag}_pmc_traffic_bench.json > $out/${tag}_pmc_traffic.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/${tag}_valu -- python3 tools/prove_loop.py --proofs 15 --throughput > $out/${tag}_valu.log 2>&1
python3 tools/pmc_valu.py $(ls $out/${tag}_valu/*/*counter_collection.csv | head -1) 16 $out/${tag}_pmc_valu_prove.json > $out/${tag}_pmc_valu_prove.txt
# the raw counter dumps are large: keep the summaries only
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write $out/${tag}_valu
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-120
head -6 $out/${tag}_pmc_traffic.txt
tail -1 $out/${tag}_pmc_valu_prove.txt
