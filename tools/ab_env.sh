#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh VAR A B [rounds] [bench.py arguments]; prints the pool
# throughput, the lone proof's p50 and the standalone NTT times of each run, alternating the two settings
set -e
var=$1; a=$2; b=$3; rounds=${4:-2}; shift 4 || true
args=${*:---cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --micro 20 24}
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for v in $a $b; do
    env $var=$v timeout -k 10 300 python3 bench.py $args > gpurun_out/abenv_$v.json 2> gpurun_out/abenv_$v.err
    python3 - $var $v <<'PY'
import json, sys
d = json.loads(open('gpurun_out/abenv_%s.json' % sys.argv[2]).read().strip().splitlines()[-1])
ntt = {m['log_n']: round(m['ms'], 4) for m in d.get('micro', {}).get('ntt', [])}
print('%s=%s' % (sys.argv[1], sys.argv[2]), round(d['value'], 1), 'proofs/s  p50', round(d['p50_latency_ms'], 4), 'ms  ntt ms', ntt, flush=True)
PY
  done
done
