"""Where does a query's wall time go beyond its dominant kernel?  Prints every traced launch."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import polars_amd as pl
from polars_amd import queries, _ffi as F
import bench

pl.init(0)
for name, rows in [("cfg2", 200_000_000), ("q1", 200_000_000), ("cfg3", 100_000_000)]:
    wl = bench.make_workload(pl, name, rows, 1)
    for _ in range(2): wl.step()
    F.lib().plx_profile_clear(); F.lib().plx_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): wl.step()
    F.lib().plx_synchronize(); dt = (time.perf_counter() - t0) / n
    st = bench.kernel_stats(pl)
    F.lib().plx_profile_enable(0)
    ksum = sum(v[1] for v in st.values()) / n
    print(f"{name}: {dt*1e3:.3f} ms/step, traced kernels {ksum/1e3:.3f} ms/step")
    for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]): print(f"   {k:32s} x{v[0]/n:.0f}  {v[1]/v[0]:.1f} us")
    del wl
    F.lib().plx_memory_trim(); torch.cuda.empty_cache()
