#!/bin/bash
# Arrow IPC scan on hardware (first run) + the parquet / io GPU tests again after the staging refactor.  No torch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02r
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 120 python -m pytest tests/test_gpu_ipc.py tests/test_gpu_parquet.py tests/test_gpu_io.py -m gpu -q --timeout 90 --durations=5 > $OUT/pytest_ipc.log 2>&1; el "ipc + parquet gpu tests exit $?"
tail -40 $OUT/pytest_ipc.log | cut -c1-300
el "end"
