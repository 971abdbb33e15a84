#!/bin/bash
# the bench's pool throughput for several values of one argument, alternated on one box:
#   tools/sweep_bench_arg.sh --lanes "40 48 56 64" [rounds]
set -e
arg=$1; vals=$2; rounds=${3:-2}
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for v in $vals; do
    timeout -k 10 300 python3 bench.py $arg $v --cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --latency-proofs 10 --micro > gpurun_out/sweep_arg.json 2> gpurun_out/sweep_arg.err
    python3 - "$arg" $v <<'PY'
import json, sys
d = json.loads(open('gpurun_out/sweep_arg.json').read().strip().splitlines()[-1])
print('%s %s' % (sys.argv[1], sys.argv[2]), round(d['value'], 1), 'proofs/s  host cpu', d['config'].get('host_cpu_util_rank0'), flush=True)
PY
  done
done
