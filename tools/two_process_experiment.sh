# One proving process with 48 lanes against two processes with 24 lanes each on the same GPU (SHP_MSM_TABLE_GB=80: a c = 14
# table each): does a second HIP runtime (its own locks, its own hardware queues) raise what one GPU proves?
set -e
cd "$GRAFT_REPO_ROOT"
export SHP_MSM_TABLE_GB=80
echo "== one process, 48 lanes"
timeout -k 10 300 python bench.py --lanes 48 --steps 30 --cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --latency-proofs 5 > gpurun_out/tp_one.json 2> gpurun_out/tp_one.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/tp_one.json') if l.startswith('{')][0]); print('one:', d['value'], d['config']['msm_tables'])"
echo "== two processes, 24 lanes each"
timeout -k 10 300 python bench.py --lanes 24 --steps 30 --cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --latency-proofs 5 > gpurun_out/tp_a.json 2> gpurun_out/tp_a.err &
PA=$!
timeout -k 10 300 python bench.py --lanes 24 --steps 30 --cpu-proofs 0 --interpreted-steps 0 --assigned-steps 0 --latency-proofs 5 > gpurun_out/tp_b.json 2> gpurun_out/tp_b.err &
PB=$!
wait $PA; wait $PB
python -c "
import json
a=json.loads([l for l in open('gpurun_out/tp_a.json') if l.startswith('{')][0]); b=json.loads([l for l in open('gpurun_out/tp_b.json') if l.startswith('{')][0])
print('two:', a['value'], b['value'], a['value']+b['value'], a['ms_per_step'], b['ms_per_step'])"
