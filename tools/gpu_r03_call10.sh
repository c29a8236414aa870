#!/bin/bash
# string-key group-by: where the time goes (experiment variants; results of variant runs are wrong by construction)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03j
mkdir -p $OUT
cd $R
for V in 3; do
  PLX_BENCH_VERIFY=0 PLX_STRGROUP_VARIANT=$V timeout 120 python bench.py --workload cfg5s --steps 3 --warmup 1 --no-extras --no-cpu > $OUT/v$V.json 2> $OUT/v$V.err
  python - $OUT/v$V.json $V <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); k = d.get("kernels") or {}
        print("variant", sys.argv[2], "ms/step", d.get("ms_per_step"), {n: round(v["avg_us"]) for n, v in k.items() if n.startswith("strgroup")})
PY
done
