import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polars_amd as pl
from polars_amd import _ffi as F
import bench
pl.init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = bench.make_workload(pl, "cfg3", rows, 1)
wl.step(); print(pl.last_plan())
F.lib().plx_profile_clear(); F.lib().plx_profile_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): r, _k = wl.step()
F.lib().plx_synchronize(); dt = (time.perf_counter() - t0) / steps
st = bench.kernel_stats(pl)
print(f"cfg3 {rows} rows: {dt*1e3:.3f} ms/step  result {r}")
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]): print(f"   {k:28s} x{v[0]/steps:4.1f}  avg {v[1]/v[0]:10.1f} us")
