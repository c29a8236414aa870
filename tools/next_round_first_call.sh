#!/bin/bash
# First GPU session of the next round (~9 GPU-minutes): the state round 3 left is fully validated on hardware (388 GPU tests, smoke, every bench workload
# oracle-verified: profiles/r03/final_session4_summary.txt), so this only re-establishes the starting measurements on the new round's boxes.
#   1. the whole GPU suite + smoke
#   2. the driver's bench line
#   3. kernel stats + FETCH / WRITE counters of every workload -> gpurun_out/<tag>/profiles/ (copy to profiles/r04/ and set bench.py PMC_ROUND = "r04")
#   4. the string-key group-by against group count and skew (tools/strgroup_sweep.py)
# usage: gpurun --timeout 1500 -- 'bash tools/next_round_first_call.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=r04a
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 700 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu_all.log 2>&1; el "whole gpu suite exit $?"
tail -4 $OUT/pytest_gpu_all.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; el "smoke exit $?"
timeout 400 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "bench exit $?"
python - $OUT/bench_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", d["config"]["workload"], "ms/step", d["ms_per_step"], "median", d.get("ms_per_step_median"), "frac", d["roofline"]["frac"], "verified", (d.get("verified") or {}).get("ok"))
for k, v in (d.get("extras") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        r = v.get("roofline") or {}
        print(" ", k, v["ms_per_step"], "frac", r.get("frac"), "verified", (v.get("verified") or {}).get("ok"), "cold", v.get("cold_first_step_ms"))
PY
bash tools/pmc_all.sh $TAG q1 q3 q3s q3f cfg2 cfg3 cfg5 cfg5s > $OUT/pmc_all.log 2>&1; el "pmc refresh exit $?"
timeout 300 python tools/strgroup_sweep.py 26 > $OUT/strgroup_sweep_2p26.json 2> $OUT/strgroup_sweep.err; el "string group-by sweep exit $?"
el "end"
