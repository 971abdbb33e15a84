#!/bin/bash
# pool throughput over hardware queue counts and lane counts: tools/sweep_queues.sh "16 22 28 32" "40 48 56"
set -e
mkdir -p gpurun_out
for q in $1; do for l in $2; do
  SHP_BENCH_HW_QUEUES=$q timeout -k 10 300 python bench.py --lanes $l --interpreted-steps 0 --assigned-steps 0 --steps 3 > gpurun_out/sweep.json 2> gpurun_out/sweep.err
  python - $q $l <<'PY'
import json, sys
d = json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('queues', sys.argv[1], 'lanes', sys.argv[2], round(d['value'], 1), 'proofs/s')
PY
done; done
