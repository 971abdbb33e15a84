#!/bin/bash
# the prover's byte-exactness tests under every switch that selects another code path (none may change a proof byte)
set -e
mkdir -p gpurun_out
for e in "SHP_MSM_LAZY29=0" "SHP_NTT29=1" "SHP_NTT_DIT=0" "SHP_TRANSFORMS_AHEAD=0" "SHP_EARLY_LOOKUPS=0" "SHP_POLL_RESULTS=0" "SHP_QUOT29=0" "SHP_GATE_AOT=0 SHP_QUOT29=0" "SHP_PROG_REG=0 SHP_GATE_AOT=0 SHP_QUOT29=0" "SHP_MSM_TABLE_GB=0" "SHP_WITNESS_THREADS=1" "SHP_WITNESS_NATIVE=0" "SHP_MSM_TRACE=1"; do
  echo "== $e"
  # (one test asserts that the shipped kernels ARE selected: not under the switches that deselect them)
  env $e timeout -k 10 400 python -u -m pytest tests/test_gpu_prover.py tests/test_gpu_phases.py -x -q --timeout 300 --timeout-method thread -m gpu -k "not new_account_shaped" > gpurun_out/env_matrix.log 2>&1 || { tail -20 gpurun_out/env_matrix.log; exit 1; }
  tail -1 gpurun_out/env_matrix.log
done
