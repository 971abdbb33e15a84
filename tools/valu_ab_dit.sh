set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for m in 0 1; do
  export SHP_NTT_DIT=$m
  rm -rf $out/valu_dit$m
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/valu_dit$m -- python3 tools/prove_loop.py --proofs 15 --throughput > $out/valu_dit$m.log 2>&1
  python3 tools/pmc_valu.py $(ls $out/valu_dit$m/*/*counter_collection.csv | head -1) 16 $out/valu_dit$m.json > $out/valu_dit$m.txt
  rm -rf $out/valu_dit$m
  echo "SHP_NTT_DIT=$m"; grep -E "ntt|total" $out/valu_dit$m.txt | cut -c1-150
done
