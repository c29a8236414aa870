#!/bin/bash
# smoke() with the parquet leg, the scan benchmark with the IPC leg, then the default bench.py run (what the driver runs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02s
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; el "smoke exit $?"
tail -2 $OUT/smoke.log | cut -c1-300
timeout 60 python tools/parquet_bench.py 2e7 > $OUT/parquet_bench.jsonl 2> $OUT/parquet_bench.err; el "scan bench exit $?"
cut -c1-420 $OUT/parquet_bench.jsonl
timeout 200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "bench exit $?"
python - $OUT/bench_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", d["config"]["workload"], "ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "cpu", d.get("cpu_baseline", {}).get("value"), "verified", (d.get("verified") or {}).get("ok"))
for k, v in (d.get("extras") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(" ", k, v["ms_per_step"], "frac", (v.get("roofline") or {}).get("frac"), "cold", v.get("cold_first_step_ms"), "verified", (v.get("verified") or {}).get("ok"))
PY
el "end"
