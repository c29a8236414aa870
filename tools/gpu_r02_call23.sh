#!/bin/bash
# Last GPU call of the round: the IPC tests after the host view assembly went parallel, and the scan benchmark (Parquet + IPC legs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02w
mkdir -p $OUT
cd $R
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 25 python -m pytest tests/test_gpu_ipc.py -m gpu -q --timeout 20 > $OUT/pytest_ipc.log 2>&1; echo "ipc tests exit $?"; tail -2 $OUT/pytest_ipc.log
timeout 25 python tools/parquet_bench.py 2e7 > $OUT/parquet_bench.jsonl 2> $OUT/parquet_bench.err; echo "scan bench exit $?"
cut -c1-330 $OUT/parquet_bench.jsonl
