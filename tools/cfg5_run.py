import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polars_amd as pl
from polars_amd import _ffi as F
import bench
pl.init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for name in ("cfg5", "cfg3"):
    wl = bench.make_workload(pl, name, rows, 1)
    wl.step(); print(pl.last_plan())
    F.lib().plx_profile_clear(); F.lib().plx_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r, _k = wl.step()
    F.lib().plx_synchronize(); dt = (time.perf_counter() - t0) / 3
    st = bench.kernel_stats(pl); F.lib().plx_profile_enable(0)
    print(f"{name} {rows} rows: {dt*1e3:.3f} ms/step")
    for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:4]: print(f"   {k:28s} avg {v[1]/v[0]:10.1f} us")
    del wl; F.lib().plx_memory_trim(); torch.cuda.empty_cache()
