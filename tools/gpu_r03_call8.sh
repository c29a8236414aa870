#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
cd $R
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 200 python -m pytest tests/test_gpu_ipc.py tests/test_gpu_io.py -m gpu -q --timeout 90 > $OUT/pytest_ipc.log 2>&1; echo "ipc tests exit $?"; tail -3 $OUT/pytest_ipc.log
unset PLX_SKIP_TORCH_PREIMPORT
PLX_IPC_TIMING=1 timeout 400 python tools/parquet_bench.py 2e7 > $OUT/parquet_bench.jsonl 2> $OUT/parquet_bench.err; echo "scan bench exit $?"
grep -E "arrow_ipc|\"codec\": \"(none|snappy|zstd)\"" $OUT/parquet_bench.jsonl | cut -c1-300
grep "plx ipc" $OUT/parquet_bench.err | tail -14
