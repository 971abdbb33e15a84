"""CPU: the N>1 path of bench.py (sharding, barrier, max-over-ranks, one JSON line from rank 0) with
world_size 2 over gloo; and the sharding helper."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_partitions_cover_everything():
    sys.path.insert(0, ROOT)
    import bench
    for total in (0, 1, 7, 1024):
        for world in (1, 2, 3, 8):
            parts = [bench.shard(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_run_prints_one_json_line():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--lanes", "2", "--selftest-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    # rank 1 sleeps twice as long per step as rank 0: the max over ranks must be reported
    assert j["ms_per_step"] >= 2 * 0.002 * 2 * 1e3 * 0.9
