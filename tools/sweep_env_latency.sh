#!/bin/bash
# the lone proof's p50 for several values of one environment variable, alternated on one box:
#   tools/sweep_env_latency.sh SHP_MSM_MIN_WAVES "512 768 1024" [rounds]
set -e
var=$1; vals=$2; rounds=${3:-2}
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for v in $vals; do
    env $var=$v timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --latency-proofs 400 --micro > gpurun_out/sweep_lat.json 2> gpurun_out/sweep_lat.err
    python3 - $var $v <<'PY'
import json, sys
d = json.loads(open('gpurun_out/sweep_lat.json').read().strip().splitlines()[-1])
print('%s=%s' % (sys.argv[1], sys.argv[2]), 'p50', round(d['p50_latency_ms'], 4), 'ms', {k: round(v) for k, v in d['phase_us_p50'].items()}, flush=True)
PY
  done
done
