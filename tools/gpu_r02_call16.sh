#!/bin/bash
# Snappy kernel phase clock on the decode-rate benchmark + the parquet GPU tests (no torch).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02p
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 100 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_io.py -m gpu -q --timeout 90 > $OUT/pytest_parquet.log 2>&1; el "parquet gpu tests exit $?"
tail -3 $OUT/pytest_parquet.log
PLX_SNAPPY_TIMING=1 timeout 100 python tools/parquet_bench.py 2e7 > $OUT/parquet_bench.jsonl 2> $OUT/parquet_bench.err; el "parquet bench exit $?"
cat $OUT/parquet_bench.jsonl; grep pq_snappy $OUT/parquet_bench.err | tail -7
el "end"
