#!/bin/bash
# A/B of two builds of the library on one box, lone-proof latency: tools/ab_latency.sh <other.so> [rounds]
set -e
other=$1; rounds=${2:-3}
so=zkos-monorepo_amd/libshielder_hip.so
mkdir -p gpurun_out
cp $so gpurun_out/ab_this.so
for r in $(seq $rounds); do
  for v in this other; do
    if [ $v = this ]; then cp gpurun_out/ab_this.so $so; else cp $other $so; fi
    echo -n "$v: "; timeout -k 10 120 python tools/prove_loop.py --proofs 300 --lanes 1 2>/dev/null | tail -1
  done
done
cp gpurun_out/ab_this.so $so
rm -f gpurun_out/ab_this.so
