"""How many host threads should a Parquet read use?  Writes the bench's 2e7-row lineitem-like table once per codec, then times
pl.read_parquet in child processes under (PLX_PARQUET_THREADS, PLX_HOST_THREADS) combinations (both are read once per process).
GPU box only:  python tools/scan_threads.py > gpurun_out/scan_threads.txt"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(paths):
    import polars_amd as pl
    pl.init(0)
    F = pl._ffi
    out = []
    for p in paths:
        pl.read_parquet(p)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); pl.read_parquet(p); F.check(F.lib().plx_synchronize()); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out.append(f"{os.path.basename(p)}: min {ts[0]:.1f} med {ts[2]:.1f}")
    print("  " + " | ".join(out), flush=True)


def main():
    import tempfile
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq
    n = 20_000_000
    rng = np.random.default_rng(3)
    t = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(1, 4 * n, n))), "l_quantity": pa.array(rng.integers(1, 51, n)), "l_extendedprice": pa.array(rng.random(n) * 1e5),
                  "l_returnflag": pa.array(np.array(["R", "A", "N"])[rng.integers(0, 3, n)]), "l_shipdate": pa.array(rng.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us")),
                  "l_nullable": pa.array(rng.integers(0, 1 << 30, n), mask=rng.random(n) < 0.1)})
    d = tempfile.mkdtemp(prefix="plx_scan_threads_")
    paths = []
    for codec in ("zstd", "snappy", "none"):
        p = os.path.join(d, f"li_{codec}.parquet")
        pq.write_table(t, p, compression=codec, row_group_size=1 << 20)
        paths.append(p)
    del t
    combos = [(1, 64), (4, 64), (4, 16), (4, 32), (2, 32), (2, 64), (3, 24), (6, 16), (1, 128)]
    for pt, ht in combos:
        print(f"PLX_PARQUET_THREADS={pt} PLX_HOST_THREADS={ht}", flush=True)
        env = dict(os.environ, PLX_PARQUET_THREADS=str(pt), PLX_HOST_THREADS=str(ht))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + paths, env=env, timeout=300)
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    child(sys.argv[2:]) if len(sys.argv) > 1 and sys.argv[1] == "--child" else main()
