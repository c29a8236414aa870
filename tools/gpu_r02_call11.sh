#!/bin/bash
# Round 2, eleventh GPU session: touched bitmap for the join output step, 256-partition default; join + partition tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02k
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
run_w() { local name=$1; local wl=$2; shift 2; ( export "$@" X=1; timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --no-cpu > $OUT/$name.json 2> $OUT/$name.err ); rc=$?
  python - "$OUT/$name.json" "$name" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    v = d.get("verified") or {}
    print(sys.argv[2], "ms/step", d["ms_per_step"], "cold", d.get("cold_first_step_ms"), "frac", d["roofline"]["frac"], {k: v2["avg_us"] for k, v2 in d["kernels"].items()}, "verified", v.get("ok"), v.get("error", v.get("note", "")))
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
  [ $rc -ne 0 ] && tail -5 $OUT/$name.err; }
run_w q3 q3 X2=1
run_w q3f q3f X2=1
run_w cfg3 cfg3 X2=1
run_w cfg5 cfg5 X2=1
el "bench done"
timeout 600 python -m pytest tests/test_gpu_queries.py -m gpu -q --timeout 300 -x > $OUT/pytest_queries.log 2>&1; el "query tests exit $?"
tail -5 $OUT/pytest_queries.log
el "end"
