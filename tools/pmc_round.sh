#!/bin/bash
# HBM traffic of the dominant kernels from the TCC counters (separate passes: FETCH_SIZE costs 3 of the 4
# TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: tools/pmc_round.sh <tag> <workload>
TAG=${1:-r1}; WL=${2:-q1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${WL}_$C -o pmc -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu > $OUT/pmc_${WL}_$C.log 2>&1
  echo "$C exit $?"
  f=$(find $OUT/pmc_${WL}_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$C" <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(f)))
if rows: print("columns:", list(rows[0].keys()))
for r in rows:
    if r.get("Counter_Name") != c: continue
    k = r["Kernel_Name"][:90]
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
    print(f"{c}  {k:92s} dispatches={n} mean={v/n:.1f}")
PY
  find $OUT/pmc_${WL}_$C -name "*.csv" -size +3M -delete
done
