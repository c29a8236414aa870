#!/bin/bash
# rocprofv3 evidence for every benchmark workload: per-kernel stats (--kernel-trace --stats) and HBM traffic from the TCC counters
# (FETCH_SIZE and WRITE_SIZE in SEPARATE passes: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"; never
# combined with sys/hip/hsa tracing).  usage (GPU box): bash tools/pmc_all.sh <tag> [workloads...]  -> gpurun_out/<tag>/profiles/
TAG=${1:-r02p}; shift
WLS=${@:-q1 q1j q3 q3s q3h q3d q3dc joinm joinmh semim q3f filterm gather cfg2 cfg2n cfg3 cfg3z cfg3s cfg3w cfg5 cfg5s cfg5l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
P=$OUT/profiles
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for WL in $WLS; do
  BWL=$WL; unset PLX_Q3_SHUFFLED
  if [ "$WL" = "q3s" ]; then BWL=q3; export PLX_Q3_SHUFFLED=1; fi      # Q3 over both tables in a seeded random row order (the partitioned probe)
  PLX_BENCH_VERIFY=0 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o $WL -- python $R/bench.py --workload $BWL --steps 10 --warmup 2 --no-extras --no-cpu > $OUT/stats_$WL.json 2> $OUT/stats_$WL.err
  echo "stats $WL exit $?"
  f=$(find $OUT/stats_$WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${WL}_kernel_stats.csv && head -5 $f | cut -c1-160
  # (the FULL record of the same session: bench.py prints it on the line before the headline; it carries the per-kernel table)
  grep -h '^\[bench\] full record: ' $OUT/stats_$WL.json | tail -1 | sed 's/^\[bench\] full record: //' > $P/${WL}_bench_line_same_session.json
  for C in FETCH_SIZE WRITE_SIZE; do
    PLX_BENCH_VERIFY=0 timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${WL}_$C -o pmc -- python $R/bench.py --workload $BWL --steps 3 --warmup 1 --no-extras --no-cpu > $OUT/pmc_${WL}_$C.log 2>&1
    echo "pmc $WL $C exit $?"
  done
  ff=$(find $OUT/pmc_${WL}_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find $OUT/pmc_${WL}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  [ -n "$ff" ] && [ -n "$fw" ] && python3 $R/tools/pmc_summarise.py $WL $ff $fw $P/${WL}_pmc.json
  find $OUT -name "*.csv" -size +2M -delete
done
rocminfo 2>/dev/null | grep -m1 -A12 "gfx950" > $P/agent_info.txt
ls -la $P
