#!/usr/bin/env python
"""The lineitem-like table written WITHOUT dictionaries in 1 MB pages (eight zstd blocks a page: what a writer that keeps PLAIN values in large pages produces), zstd level 3:
device passes vs the host threads.  usage: python tools/zstd_read_plain.py [rows] [reads]"""
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
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(3)
t = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(1, 4 * n, n))), "l_quantity": pa.array(rng.integers(1, 51, n)),
              "l_extendedprice": pa.array(rng.random(n) * 1e5), "l_discount": pa.array(rng.integers(0, 11, n) / 100.0),
              "l_shipdate": pa.array(rng.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us")),
              "l_nullable": pa.array(rng.integers(0, 1 << 30, n), mask=rng.random(n) < 0.1)})
d = tempfile.mkdtemp()
path = os.path.join(d, "li_plain_zstd.parquet")
pq.write_table(t, path, compression="zstd", compression_level=3, use_dictionary=False, data_page_size=1 << 20, row_group_size=1 << 20, max_rows_per_page=int(os.environ.get("PAGE_ROWS", "20000")))
print("file_MB", round(os.path.getsize(path) / 1e6, 1), "pages", sum(1 for _ in range(1)))
pl.init(0)
F = pl._ffi
want = pq.read_table(path)
for mode in ("device", "host"):
    if mode == "host":
        os.environ["PLX_PARQUET_ZSTD"] = "host"
    pl.read_parquet(path)
    ts = []
    for _ in range(reads):
        t0 = time.perf_counter(); df = pl.read_parquet(path); F.check(F.lib().plx_synchronize()); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    ok = all(np.array_equal(df[c].to_numpy(), want.column(c).to_numpy()) for c in ("l_orderkey", "l_quantity", "l_extendedprice", "l_discount"))
    print(mode, "read_ms", ts, "matches_pyarrow", ok, flush=True)
