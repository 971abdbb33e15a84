import importlib, json, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
shp = importlib.import_module("zkos-monorepo_amd")
ctx = shp.Context(0)
what, lanes = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(5)
if what == "msm":
    batch = 7
    params = shp.Params.synthetic(ctx, 13, 0xDEADBEEF12345678)
    g = params.get_g()
    plans = [shp.Msm(ctx, g, 0) for _ in range(lanes)]      # SHP_MSM_TABLE_GB=2.5: c = 8 tables, 2 GB per lane
    raw = rng.integers(0, 2**63, size=(batch * 8192, 4), dtype=np.uint64); raw[:, 3] &= (1 << 60) - 1
    def work(i, reps):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            t = shp.to_device(raw)
            for _ in range(reps):
                plans[i].run(t, 8192, batch)
            s.synchronize()
    unit = batch * 8192 * 32
else:
    lg, batch = 16, 16
    raw = rng.integers(0, 2**63, size=(batch << lg, 4), dtype=np.uint64); raw[:, 3] &= (1 << 60) - 1
    def work(i, reps):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            t = shp.to_device(raw)
            for _ in range(reps):
                ctx.ntt(t, lg, batch)
            s.synchronize()
    unit = batch * (1 << lg) // 2 * lg
work(0, 2)
reps = 30
ths = [threading.Thread(target=work, args=(i, reps)) for i in range(lanes)]
t0 = time.time()
[t.start() for t in ths]; [t.join() for t in ths]
el = time.time() - t0
print(json.dumps({"what": what, "lanes": lanes, "G_units_per_s": unit * reps * lanes / el / 1e9, "ms_per_call": el / reps * 1e3}))
