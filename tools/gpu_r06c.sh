cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_zzz_scan_host_paths.py tests/test_gpu_scan_strings.py tests/test_gpu_io.py -x -q -m gpu 2>&1 | tail -4
for B in 0 20 40 64 1000; do
PLX_PARQUET_SNAPPY_BATCH=$((B * 1048576)) python - <<PY
import json, os, bench, polars_amd as pl
pl.init(0)
r = bench.scan_extra(pl, 20_000_000)
f = r["files"]
print("batch MB", os.environ.get("PLX_PARQUET_SNAPPY_BATCH"), {k: (v["read_ms"], v.get("kernel_us", {}).get("pq_snappy")) for k, v in f.items()})
PY
done
