#!/bin/bash
# the file scans of the bench line alone (Parquet none / Snappy / zstd, Arrow IPC: 2e7 rows) + the Snappy kernel's phase clock
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/scan
PLX_SNAPPY_TIMING=${PLX_SNAPPY_TIMING:-0} timeout 600 python - > gpurun_out/scan/scan.json 2> gpurun_out/scan/scan.err <<'PY'
import json
import polars_amd as pl
import bench
pl.init(0)
r = bench.scan_extra(pl, 20_000_000)
for k, v in r["files"].items():
    print(k, v["read_ms"], v["pyarrow_read_ms"], v["verified"], {a: b for a, b in v["kernel_us"].items() if b > 500})
PY
cat gpurun_out/scan/scan.json; tail -5 gpurun_out/scan/scan.err
