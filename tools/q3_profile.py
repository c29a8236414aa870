import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polars_amd as pl
from polars_amd import _ffi as F
import bench
pl.init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wl = bench.make_workload(pl, "q3", rows, 1)
wl.step()
print(pl.last_plan())
F.lib().plx_profile_clear(); F.lib().plx_profile_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n): r, _k = wl.step()
F.lib().plx_synchronize(); dt = (time.perf_counter() - t0) / n
st = bench.kernel_stats(pl)
print(f"q3: {dt*1e3:.3f} ms/step, traced {sum(v[1] for v in st.values())/n/1e3:.3f} ms/step, result {r}")
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]): print(f"   {k:28s} x{v[0]/n:5.1f}  avg {v[1]/v[0]:10.1f} us   total/step {v[1]/n/1e3:8.3f} ms")
