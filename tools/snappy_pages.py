"""Is the Snappy read bound by its LONGEST stream?  The bench's 2e7-row table written with and without dictionary pages (pyarrow writes a
dictionary page of up to 1 MB per column chunk -- one Snappy stream, decoded by one workgroup round after round) and with smaller dictionary limits.
GPU box only:  python tools/snappy_pages.py"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import polars_amd as pl
import bench

pl.init(0)
F = pl._ffi
n = 20_000_000
rng = np.random.default_rng(3)
t = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(1, 4 * n, n))), "l_quantity": pa.array(rng.integers(1, 51, n)), "l_extendedprice": pa.array(rng.random(n) * 1e5),
              "l_returnflag": pa.array(np.array(["R", "A", "N"])[rng.integers(0, 3, n)]), "l_shipdate": pa.array(rng.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us")),
              "l_nullable": pa.array(rng.integers(0, 1 << 30, n), mask=rng.random(n) < 0.1)})
d = tempfile.mkdtemp(prefix="plx_snappy_pages_")
for label, kw in (("default", {}), ("no_dictionary", {"use_dictionary": False}), ("dict_limit_128k", {"dictionary_pagesize_limit": 128 << 10}),
                  ("pages_64k", {"data_page_size": 64 << 10, "dictionary_pagesize_limit": 64 << 10})):
    p = os.path.join(d, f"li_{label}.parquet")
    pq.write_table(t, p, compression="snappy", row_group_size=1 << 20, **kw)
    pl.read_parquet(p)
    F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pl.read_parquet(p); F.check(F.lib().plx_synchronize()); ts.append((time.perf_counter() - t0) * 1e3)
    ks = {k: (v[0], round(v[1] / 5)) for k, v in bench.kernel_stats(pl).items() if v[1] / 5 > 300}
    F.check(F.lib().plx_profile_enable(0))
    ts.sort()
    print(f"{label}: file {os.path.getsize(p) / 1e6:.0f} MB, read min {ts[0]:.1f} med {ts[2]:.1f} ms; kernels (launches, us per read): {ks}", flush=True)
    os.remove(p)
os.rmdir(d)
