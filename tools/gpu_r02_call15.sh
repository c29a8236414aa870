#!/bin/bash
# Round 2, re-entry session: first run of the device Parquet decoder on hardware (GPU tests + decode rate).  No torch import.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02o
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 150 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_io.py -m gpu -q --timeout 120 --durations=6 > $OUT/pytest_parquet.log 2>&1; el "parquet gpu tests exit $?"
tail -40 $OUT/pytest_parquet.log
timeout 100 python tools/parquet_bench.py 2e7 > $OUT/parquet_bench.jsonl 2> $OUT/parquet_bench.err; el "parquet bench exit $?"
cat $OUT/parquet_bench.jsonl; tail -3 $OUT/parquet_bench.err
el "end"
