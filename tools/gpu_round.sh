#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [tag] [pytest-args...]
TAG=${1:-r1}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/device.txt 2>&1
nproc >> $OUT/device.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 "$@" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q1 -o q1 -- python $R/bench.py --steps 5 --warmup 2 --no-extras --no-cpu > $OUT/rocprof_q1.log 2>&1
echo "rocprof exit $?"
find $OUT/prof_q1 -name "*stats*" | head; 
f=$(find $OUT/prof_q1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f
# keep the merge-back small: drop the raw per-dispatch trace
find $OUT/prof_q1 -name "*kernel_trace.csv" -size +5M -delete
