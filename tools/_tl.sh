#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$1 -- python3 tools/prove_loop.py --proofs 3 > gpurun_out/tl_$1.log 2>&1
python tools/trace_timeline.py $(ls gpurun_out/tl_$1/*/*kernel_trace.csv | head -1) > gpurun_out/timeline_$1.txt
