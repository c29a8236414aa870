#!/bin/bash
# generation-3 scatter with the one-partition-per-thread scan and the descriptor copy-out: whole GPU suite, then the workloads it carries
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03n
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu_all.log 2>&1; echo "whole gpu suite exit $?"
tail -4 $OUT/pytest_gpu_all.log | cut -c1-300
for WL in cfg3 cfg5 q3s cfg5s; do
  BWL=$WL; unset PLX_Q3_SHUFFLED
  if [ "$WL" = "q3s" ]; then BWL=q3; export PLX_Q3_SHUFFLED=1; fi
  timeout 300 python bench.py --workload $BWL --steps 8 --warmup 3 --no-extras --no-cpu > $OUT/$WL.json 2> $OUT/$WL.err; echo "$WL exit $?"
  python - $OUT/$WL.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); k = d.get("kernels") or {}
        print("   ms/step", d.get("ms_per_step"), "median", d.get("ms_per_step_median"), "verified", (d.get("verified") or {}).get("ok"), {n: round(v["avg_us"]) for n, v in list(k.items())[:5]})
PY
done
