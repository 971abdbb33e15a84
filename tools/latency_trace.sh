# Kernel-by-kernel trace of a LONE proof (latency mode, one lane): gpurun_out/<tag>_dispatch_sequence_latency.txt
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/lat_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/lat_trace -- python3 tools/prove_loop.py --proofs 30 --lanes 1 > $out/${tag}_lat_trace.log 2>&1
python3 tools/dispatch_sequence.py $(ls $out/lat_trace/*/*kernel_trace.csv | head -1) > $out/${tag}_dispatch_sequence_latency.txt
rm -rf $out/lat_trace
tail -2 $out/${tag}_lat_trace.log
cat $out/${tag}_dispatch_sequence_latency.txt
