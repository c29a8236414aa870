import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polars_amd as pl
import bench
pl.init(0)
n = 1 << 25
rng = np.random.default_rng(7)
v = rng.integers(-1000, 1000, n).astype(np.int64)
k = rng.integers(0, 1_000_000, n).astype(np.int64); k[rng.random(n) < 0.5] = 777_777
key = k * 1_000_003 - 5
df = pl.DataFrame({"key": key, "v": v})
q = df.lazy().group_by("key").agg(pl.col("v").sum().alias("s"), pl.col("v").count().alias("c"))
F = pl._ffi
for rep in range(3):
    F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
    t0 = time.perf_counter(); r = q.collect(); dt = time.perf_counter() - t0
    ks = {k: round(v[1]) for k, v in bench.kernel_stats(pl).items()}
    print(os.environ.get("PLX_PART_GEN"), os.environ.get("PLX_SAMPLE_CACHE"), rep, round(dt * 1e3, 2), "ms", r.height, ks, pl.last_plan()[:200], flush=True)
