#!/bin/bash
# The default bench.py run (what the driver runs) after the harness stopped garbage-collecting inside timed regions.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02v
mkdir -p $OUT
cd $R
timeout 200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench exit $?"
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", d["config"]["workload"], "ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"), "verified", (d.get("verified") or {}).get("ok"))
print("step_ms", d["step_ms"])
for k, v in (d.get("extras") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(" ", k, v["ms_per_step"], "frac", (v.get("roofline") or {}).get("frac"), "cold", v.get("cold_first_step_ms"), "verified", (v.get("verified") or {}).get("ok"))
PY
