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


def test_gpus_flag_outside_torchrun_spawns_the_ranks_itself():
    """`bench.py --gpus 2` started as a plain process launches the two-rank run as a child (before touching a GPU) instead of
    silently reporting a one-GPU number; --batch shards a fixed batch (strong scaling)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "7",
           "--selftest-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["shard"] == [0, 4]
    assert "torch.distributed.run" in out.stderr


def test_gpus_flag_must_match_the_launch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--selftest-cpu"], env=env,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
