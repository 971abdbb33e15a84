#!/usr/bin/env python3
"""Micro-benchmarks of the standalone primitives (BASELINE.json configs 4 and 5) on one GPU.

  python tools/microbench.py [--ntt 20 22 24] [--msm 16 20] [--reps 5]
Reports ms per launch group, algorithmic bytes / time vs the 8 TB/s HBM roofline (64*N for the
NTT, 96*N for the MSM: a 32-byte scalar and a 64-byte affine base per term) and the measured Fr-mul ceiling.
"""
import argparse, importlib, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shp = importlib.import_module("zkos-monorepo_amd")


def rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 60) - 1)
    return raw


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntt", type=int, nargs="*", default=[13, 16, 20, 22, 24])
    ap.add_argument("--msm", type=int, nargs="*", default=[13, 16, 20, 22, 24])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--window", type=int, default=0)
    a = ap.parse_args()
    ctx = shp.Context(0)
    out = {"fr_mul_per_s": ctx.fr_mul_rate(4096)}
    print(json.dumps({"fr_mul_per_s": out["fr_mul_per_s"]}), flush=True)
    for lg in a.ntt:
        n = 1 << lg
        batch = max(1, (1 << 20) // n)
        t = shp.to_device(rand_fr(n * batch, lg))
        ctx.ntt(t, lg, batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ctx.ntt(t, lg, batch)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        gbs = 64.0 * n * batch / (ms * 1e-3) / 1e9
        print(json.dumps({"ntt_log_n": lg, "batch": batch, "ms": ms, "alg_GBps": gbs, "frac_hbm": gbs / 8000.0}), flush=True)
        del t
    # bases: distinct multiples generated on device would need a kernel; use a host-side walk over
    # a small table tiled to size (MSM cost does not depend on the bases being distinct)
    for lg in a.msm:
        n = 1 << lg
        import ctypes
        base = rand_points(min(n, 4096))
        bases = np.tile(base, (n // base.shape[0], 1))
        t0 = time.time()
        m = shp.Msm(ctx, bases, a.window)
        t_tab = time.time() - t0
        s = shp.to_device(rand_fr(n, 99 + lg))
        m.run(s, n, 1)
        t0 = time.time()
        for _ in range(a.reps):
            m.run(s, n, 1)
        ms = (time.time() - t0) / a.reps * 1e3
        adds, nbytes = m.stats()
        print(json.dumps({"msm_log_n": lg, "ms": ms, "table_s": t_tab, "point_adds": adds, "adds_per_s": adds / (ms * 1e-3),
                          "alg_GBps": nbytes / (ms * 1e-3) / 1e9, "frac_hbm": nbytes / (ms * 1e-3) / 8e12}), flush=True)
        m.close()
        del ctypes


def rand_points(n):
    """n curve points as k*G for small k: built with python ints on the host (tooling only)"""
    Q = shp.Q_MOD
    def add(p, q):
        if p is None: return q
        if q is None: return p
        (x1, y1), (x2, y2) = p, q
        if x1 == x2:
            if (y1 + y2) % Q == 0: return None
            l = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
        else:
            l = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
        x3 = (l * l - x1 - x2) % Q
        return (x3, (l * (x1 - x3) - y1) % Q)
    pts, p = [], (1, 2)
    g = (1, 2)
    for _ in range(n):
        pts.append(p)
        p = add(p, g)
    RM = 1 << 256
    a = np.zeros((n, 8), dtype=np.uint64)
    for i, (x, y) in enumerate(pts):
        xm, ym = x * RM % Q, y * RM % Q
        a[i, :4] = [(xm >> (64 * j)) & (2**64 - 1) for j in range(4)]
        a[i, 4:] = [(ym >> (64 * j)) & (2**64 - 1) for j in range(4)]
    return a


if __name__ == "__main__":
    main()
