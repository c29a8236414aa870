#!/bin/bash
# Round 2, re-entry session (9 GPU-minutes left): validates HEAD's last unvalidated pieces (per-node fill_null, sample pass,
# planner boundary) and fills the skew timings -- no torch import anywhere in this call.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02n
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 110 python -m pytest tests/test_gpu_null_exprs.py tests/test_gpu_queries.py -m gpu -q --timeout 100 --durations=8 \
  -k "null or boolean or skew or planned or cancelling or direct_mode or declared" > $OUT/pytest_sel.log 2>&1; el "selected gpu tests exit $?"
tail -15 $OUT/pytest_sel.log
timeout 70 python tools/skew_timing.py 26 2>$OUT/skew.err | tail -1 > $OUT/skew_timing_2p26.json; el "skew timing exit $?"
cat $OUT/skew_timing_2p26.json | cut -c1-1500
el "end"
