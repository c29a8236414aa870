#!/usr/bin/env python
"""One zstd Parquet file (the lineitem-like table of tools/parquet_bench.py) read a few times: for rocprofv3 --kernel-trace over the zstd passes.
usage: python tools/zstd_read.py [rows] [reads] [level]"""
import os
import sys
import tempfile
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pl  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
level = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
rng = np.random.default_rng(3)
t = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(1, 4 * n, n))), "l_quantity": pa.array(rng.integers(1, 51, n)),
              "l_extendedprice": pa.array(rng.random(n) * 1e5), "l_discount": pa.array(rng.integers(0, 11, n) / 100.0),
              "l_returnflag": pa.array(np.array(["R", "A", "N"])[rng.integers(0, 3, n)]),
              "l_shipdate": pa.array(rng.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us")),
              "l_nullable": pa.array(rng.integers(0, 1 << 30, n), mask=rng.random(n) < 0.1)})
d = tempfile.mkdtemp()
path = os.path.join(d, "li_zstd.parquet")
pq.write_table(t, path, compression="zstd", compression_level=level, row_group_size=1 << 20)
pl.init(0)
F = pl._ffi
cols = sys.argv[4].split(",") if len(sys.argv) > 4 else None
pl.read_parquet(path, columns=cols)
for _ in range(reads):
    t0 = time.perf_counter(); df = pl.read_parquet(path, columns=cols); F.check(F.lib().plx_synchronize()); print("read_ms", round((time.perf_counter() - t0) * 1e3, 2), flush=True)
want = pq.read_table(path, columns=cols)
ok = all(np.array_equal(df[c].to_numpy(), want.column(c).to_numpy()) for c in want.column_names if c not in ("l_returnflag", "l_shipdate", "l_nullable"))
print("matches_pyarrow", ok)
