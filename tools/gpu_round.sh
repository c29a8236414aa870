#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace + PMC traffic.  Everything lands in gpurun_out/<tag>/.
TAG=${1:-r1}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 120 --durations=12 "$@" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -18 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for WL in q1 q3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$WL -o $WL -- python $R/bench.py --workload $WL --steps 10 --warmup 2 --no-extras --no-cpu > $OUT/rocprof_$WL.log 2>&1
  echo "rocprof $WL exit $?"
  f=$(find $OUT/prof_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 $f | cut -c1-200
  find $OUT/prof_$WL -name "*kernel_trace.csv" -size +2M -delete
done
cd $R; bash tools/pmc_round.sh $TAG q3 2>&1 | grep -v columns | tail -14
