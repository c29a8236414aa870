"""PCIe-inclusive rates of the boundary's host-buffer path (plx_column_from_host): pageable vs page-locked-in-place."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["PLX_PIN_UPLOADS"] = mode
import polars_amd as pl
pl.init(0)
n = 250_000_000   # 2 GB per column
a = np.arange(n, dtype=np.int64)
for rep in range(3):
    t0 = time.perf_counter(); s = pl.Series("a", a); dt = time.perf_counter() - t0
    print(f"PLX_PIN_UPLOADS={mode}: upload {a.nbytes/1e9:.1f} GB in {dt*1e3:.1f} ms = {a.nbytes/dt/1e9:.1f} GB/s")
    del s
t0 = time.perf_counter(); b = pl.Series("a", a).to_numpy(); dt = time.perf_counter() - t0
assert (b[-5:] == a[-5:]).all()
print(f"upload + download round trip {dt*1e3:.1f} ms")
