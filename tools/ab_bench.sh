#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_bench.sh <other.so> [rounds] [bench.py arguments];
# prints the pool throughput (and the assigned / interpreted blocks when they run) of each run
set -e
other=$1; rounds=${2:-2}; shift; shift || true
args=${*:---interpreted-steps 0 --assigned-steps 0}
so=zkos-monorepo_amd/libshielder_hip.so
mkdir -p gpurun_out
cp $so gpurun_out/ab_this.so
for r in $(seq $rounds); do
  for v in this other; do
    if [ $v = this ]; then cp gpurun_out/ab_this.so $so; else cp $other $so; fi
    timeout -k 10 300 python bench.py $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    python - $v <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
extra = {k: round(d[k]['value'], 1) for k in ('assigned', 'interpreted') if isinstance(d.get(k), dict) and 'value' in d[k]}
print(sys.argv[1], round(d['value'], 1), 'proofs/s', extra)
PY
  done
done
cp gpurun_out/ab_this.so $so
rm -f gpurun_out/ab_this.so
