import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
shp = importlib.import_module("zkos-monorepo_amd")
mb = importlib.import_module("tools.microbench") if False else None
ctx = shp.Context(0)
params = shp.Params.synthetic(ctx, 13, 0xDEADBEEF12345678)
g = params.get_g()
m = shp.Msm(ctx, g, 0)
rng = np.random.default_rng(5)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
raw = rng.integers(0, 2**63, size=(batch * 8192, 4), dtype=np.uint64)
raw[:, 3] &= (1 << 60) - 1
s = shp.to_device(raw)
m.run(s, 8192, batch)
t0 = time.time()
reps = 10
for _ in range(reps):
    m.run(s, 8192, batch)
ms = (time.time() - t0) / reps * 1e3
adds, _ = m.stats()
print(json.dumps({"budget": os.environ.get("SHP_MSM_TABLE_GB"), "batch": batch, "ms": ms, "adds": adds, "Gadds_per_s": adds / ms / 1e6}))
