#!/usr/bin/env python3
"""Rate of the Merkle witness service: arity-7 Poseidon2 tree of height 13 over N notes built on
the GPU, then getMerklePath for batches of leaves.   python tools/merkle_bench.py [--leaves 1000000] [--queries 4096]"""
import argparse, ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
shp = importlib.import_module("zkos-monorepo_amd")
B = importlib.import_module("zkos-monorepo_amd.binding")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=int, default=1000000)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--height", type=int, default=13)
    a = ap.parse_args()
    ctx = shp.Context(0)
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 2**63, size=(a.leaves, 4), dtype=np.uint64)
    raw[:, 3] &= (1 << 60) - 1
    t = shp.to_device(raw)
    ctx.sync()
    t0 = time.perf_counter()
    tree = shp.MerkleTree(ctx, t, a.leaves, a.height)
    root = tree.root()
    t1 = time.perf_counter()
    hashes = sum(-(-a.leaves // 7 ** (l + 1)) for l in range(a.height))
    idx = (C.c_uint64 * a.queries)(*[int(x) for x in rng.integers(0, a.leaves, size=a.queries)])
    per = 7 * a.height + 1
    out = C.create_string_buffer(32 * per * a.queries)
    B._check(B.lib().shp_merkle_paths(tree.h, idx, C.c_size_t(a.queries), out))
    t2 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        B._check(B.lib().shp_merkle_paths(tree.h, idx, C.c_size_t(a.queries), out))
    t3 = time.perf_counter()
    print(json.dumps({"leaves": a.leaves, "height": a.height, "poseidon2_hashes": hashes, "build_ms": 1e3 * (t1 - t0),
                      "hashes_per_s": hashes / (t1 - t0), "path_queries": a.queries, "paths_per_s": a.queries * reps / (t3 - t2),
                      "path_bytes": 32 * per, "root_low64": root & (2**64 - 1)}))
    tree.close()


if __name__ == "__main__":
    main()
