"""Development probe: config 3 on zipf keys at a given size, plan + per-kernel times for a few steps (python tools/zipf_probe.py [log2_rows])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polars_amd as pl
import bench
from polars_amd import datagen, queries
pl.init(0)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
seed = 20
kz = datagen.zipf_native(pl, "key", n, seed, 0, 1_000_000)
v = datagen.uniform_native(pl, "v", pl.Int64, n, seed, 1, 0, 1000)
df = pl.DataFrame([kz, v])
F = pl._ffi
for step in range(3):
    F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
    t0 = time.perf_counter()
    out = queries.cfg3(df.lazy()).collect()
    F.check(F.lib().plx_synchronize())
    dt = (time.perf_counter() - t0) * 1e3
    st = bench.kernel_stats(pl)
    print(f"step {step}: {dt:.2f} ms  groups={out.height}  plan={pl.last_plan()}")
    for k, x in sorted(st.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"     {k}: {x[0]} x {x[1] / x[0]:.1f} us")
if n <= 1 << 27:
    k = datagen.zipf_native_host_mt(0, n, seed, 0, 1_000_000)
    vv = datagen.uniform_native_host_mt("Int64", 0, n, seed, 1, 0, 1000)
    o = np.argsort(out["key"].to_numpy())
    want = np.bincount(k, weights=vv, minlength=1_000_000).astype(np.int64)
    present = np.nonzero(np.bincount(k, minlength=1_000_000))[0]
    print("keys ok", np.array_equal(out["key"].to_numpy()[o], present), "sums ok", np.array_equal(out["v_sum"].to_numpy()[o], want[present]))
