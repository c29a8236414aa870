#!/bin/bash
# Round 2, re-entry session: rocprofv3 kernel stats of the FINAL code for the workloads that changed most since profiles/r02's
# first collection (q1 headline, q3, cfg3, cfg5) + that session's own bench lines.  Stats only: the PMC traffic files stay.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=r02q
OUT=$R/gpurun_out/$TAG
P=$OUT/profiles
mkdir -p $P
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
cd /tmp && export TMPDIR=/tmp
for WL in q1 q3 cfg3 cfg5; do
  PLX_BENCH_VERIFY=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o $WL -- python $R/bench.py --workload $WL --steps 10 --warmup 3 --no-extras --no-cpu > $OUT/stats_$WL.json 2> $OUT/stats_$WL.err
  el "stats $WL exit $?"
  f=$(find $OUT/stats_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${WL}_kernel_stats.csv && head -4 $f | cut -c1-150
  grep -h '^{' $OUT/stats_$WL.json | tail -1 > $P/${WL}_bench_line_same_session.json
  python3 - $P/${WL}_bench_line_same_session.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], {k: round(v["avg_us"]) for k, v in d["kernels"].items()})
PY
  find $OUT -name "*.csv" -size +2M -delete
done
el "end"
