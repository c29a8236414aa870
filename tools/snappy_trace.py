"""Measurement: host timeline of one Snappy Parquet read (PLX_PARQUET_TRACE=1 of the traced read only) -- where the column threads are while the read lasts.
python tools/snappy_trace.py [rows]"""
import os, sys, tempfile, time
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(3)
t = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(1, 4 * n, n))), "l_quantity": pa.array(rng.integers(1, 51, n)),
              "l_extendedprice": pa.array(rng.random(n) * 1e5), "l_returnflag": pa.array(np.array(["R", "A", "N"])[rng.integers(0, 3, n)]),
              "l_shipdate": pa.array(rng.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us")),
              "l_nullable": pa.array(rng.integers(0, 1 << 30, n), mask=rng.random(n) < 0.1)})
d = tempfile.mkdtemp()
path = os.path.join(d, "li_snappy.parquet")
pq.write_table(t, path, compression="snappy", row_group_size=1 << 20)
os.environ["PLX_PARQUET_TRACE"] = "1"
import polars_amd as pl
pl.init(0)
F = pl._ffi
for i in range(8):
    print(f"---- read {i}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); df = pl.read_parquet(path); F.check(F.lib().plx_synchronize()); print(f"---- read {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr, flush=True)
