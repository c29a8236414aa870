#!/usr/bin/env python
"""Device Parquet decode rate: a lineitem-like file (PLAIN doubles / int64 keys, dictionary-encoded low-cardinality ints and
strings, a nullable column) written by pyarrow, read with polars_amd.read_parquet (decoder="device"), per-kernel times from the
library's HIP-event profile, pyarrow's own multi-threaded decode of the same file as the CPU yardstick.
usage (GPU box): python tools/parquet_bench.py [rows] -> one JSON line per codec"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pl  # noqa: E402


def only_snappy(path):
    """child process of main(): time one existing Snappy file (with whatever PLX_SNAPPY_KERNEL the parent set) and check it against pyarrow"""
    pl.init(0)
    F = pl._ffi
    import bench
    want = pq.read_table(path)
    df = pl.read_parquet(path)
    ok = all(np.array_equal(df[c].to_numpy(), want.column(c).to_numpy()) for c in ("l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_shipdate") if c in want.column_names and c != "l_shipdate") \
        and df["l_returnflag"].to_list()[:1000] == want.column("l_returnflag").to_pylist()[:1000]
    F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); df = pl.read_parquet(path); F.check(F.lib().plx_synchronize()); ts.append(time.perf_counter() - t0)
    ks = {k: round(v[1] / 3) for k, v in bench.kernel_stats(pl).items()}
    print(json.dumps({"codec": "snappy@kernel" + os.environ.get("PLX_SNAPPY_KERNEL", "1"), "read_s": round(min(ts), 4), "kernel_us_per_read": ks, "matches_pyarrow": bool(ok)}))


def main():
    if len(sys.argv) > 3 and sys.argv[2] == "--only-snappy":
        return only_snappy(sys.argv[3])
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    rng = np.random.default_rng(3)
    t = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(1, 4 * n, n))), "l_quantity": pa.array(rng.integers(1, 51, n)),
                  "l_extendedprice": pa.array(rng.random(n) * 1e5), "l_discount": pa.array(rng.integers(0, 11, n) / 100.0),
                  "l_returnflag": pa.array(np.array(["R", "A", "N"])[rng.integers(0, 3, n)]),
                  "l_shipdate": pa.array(rng.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us")),
                  "l_nullable": pa.array(rng.integers(0, 1 << 30, n), mask=rng.random(n) < 0.1)})
    decoded = sum(c.nbytes for c in t.columns)
    pl.init(0)
    F = pl._ffi
    import bench
    d = tempfile.mkdtemp()
    codecs = tuple(os.environ.get("PLX_PQBENCH_CODECS", "none,snappy,snappy@kernel2,snappy@host,zstd,zstd@host,lz4,gzip").split(","))
    for codec in codecs:          # lz4 (raw) / gzip: pages inflated by host threads, then the uncompressed device path; zstd: device passes (round 6), zstd@host: the host threads
        os.environ.pop("PLX_PARQUET_SNAPPY", None)
        os.environ.pop("PLX_PARQUET_ZSTD", None)
        if codec == "snappy@kernel2":        # the same file through pq_snappy_kernel_v2 (batched LDS loads); needs its own process: the switch is read once
            import subprocess
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "--only-snappy", os.path.join(d, "li_snappy.parquet")], capture_output=True, text=True,
                               env={**os.environ, "PLX_SNAPPY_KERNEL": "2"})
            print((r.stdout.strip().splitlines() or [json.dumps({"codec": codec, "error": r.stderr[-300:]})])[-1])
            sys.stderr.write(r.stderr[-2000:])
            continue
        if codec == "snappy@host":           # the same file, Snappy pages inflated by the host threads instead of pq_snappy (experiment switch)
            os.environ["PLX_PARQUET_SNAPPY"] = "host"
            path = os.path.join(d, "li_snappy.parquet")
        elif codec == "zstd@host":           # the same file, zstd pages inflated by the host threads instead of the device passes
            os.environ["PLX_PARQUET_ZSTD"] = "host"
            path = os.path.join(d, "li_zstd.parquet")
        else:
            path = os.path.join(d, f"li_{codec}.parquet")
            pq.write_table(t, path, compression=codec, row_group_size=1 << 20)
        fbytes = os.path.getsize(path)
        pl.read_parquet(path)        # warm: page cache, pool, pinned staging
        F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); df = pl.read_parquet(path); F.check(F.lib().plx_synchronize()); ts.append(time.perf_counter() - t0)
        ks = {k: round(v[1] / 3) for k, v in bench.kernel_stats(pl).items()}
        F.check(F.lib().plx_profile_enable(0))
        t0 = time.perf_counter(); pq.read_table(path); t_pa = time.perf_counter() - t0
        best = min(ts)
        print(json.dumps({"rows": n, "codec": codec, "file_bytes": fbytes, "decoded_bytes": decoded, "read_s": round(best, 4), "file_GBps": round(fbytes / best / 1e9, 2),
                          "decoded_GBps": round(decoded / best / 1e9, 2), "rows_per_s": round(n / best), "kernel_us_per_read": ks, "kernel_ms_total": round(sum(ks.values()) / 1e3, 2),
                          "pyarrow_read_table_s": round(t_pa, 4), "pyarrow_threads": pa.cpu_count()}))
        del df
    os.environ.pop("PLX_PARQUET_SNAPPY", None)
    os.environ.pop("PLX_PARQUET_ZSTD", None)
    if os.environ.get("PLX_PQBENCH_NO_IPC"):
        return
    # the same table as an uncompressed Arrow IPC file: no decode at all, buffers are DMA'd into place (strings: device dictionary encode)
    import pyarrow.ipc as ipc
    path = os.path.join(d, "li.arrow")
    with ipc.new_file(path, t.schema) as w:
        for b in t.to_batches(max_chunksize=1 << 20):
            w.write_batch(b)
    fbytes = os.path.getsize(path)
    pl.read_ipc(path)
    F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); df = pl.read_ipc(path); F.check(F.lib().plx_synchronize()); ts.append(time.perf_counter() - t0)
    ks = {k: round(v[1] / 3) for k, v in bench.kernel_stats(pl).items()}
    F.check(F.lib().plx_profile_enable(0))
    t0 = time.perf_counter(); ipc.open_file(path).read_all(); t_pa = time.perf_counter() - t0
    best = min(ts)
    print(json.dumps({"rows": n, "format": "arrow_ipc_uncompressed", "file_bytes": fbytes, "read_s": round(best, 4), "file_GBps": round(fbytes / best / 1e9, 2),
                      "rows_per_s": round(n / best), "kernel_us_per_read": ks, "pyarrow_read_all_s_mmap_zero_copy": round(t_pa, 4)}))
    # ... and with LZ4-frame bodies (pyarrow's feather default): buffers inflated by host threads in parallel
    path = os.path.join(d, "li_lz4.arrow")
    with ipc.new_file(path, t.schema, options=ipc.IpcWriteOptions(compression="lz4")) as w:
        for b in t.to_batches(max_chunksize=1 << 20):
            w.write_batch(b)
    fbytes = os.path.getsize(path)
    pl.read_ipc(path)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); df = pl.read_ipc(path); F.check(F.lib().plx_synchronize()); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ipc.open_file(path).read_all(); t_pa = time.perf_counter() - t0
    best = min(ts)
    print(json.dumps({"rows": n, "format": "arrow_ipc_lz4", "file_bytes": fbytes, "read_s": round(best, 4), "file_GBps": round(fbytes / best / 1e9, 2),
                      "decoded_GBps": round(decoded / best / 1e9, 2), "rows_per_s": round(n / best), "pyarrow_read_all_s": round(t_pa, 4)}))


if __name__ == "__main__":
    main()
