#!/bin/bash
# Round 2, closing GPU session: the whole GPU suite, smoke(), the default bench.py run (headline + extras + CPU baseline), then
# rocprofv3 kernel stats + FETCH/WRITE PMC passes for every workload (tools/pmc_all.sh) -> gpurun_out/r02z/ (copy to profiles/r02/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02z
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; el "gpu suite exit $?"
tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; el "smoke exit $?"
tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "bench exit $?"
python - $OUT/bench_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", d["config"]["workload"], "ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "cpu", d.get("cpu_baseline", {}).get("value"), "verified", (d.get("verified") or {}).get("ok"))
for k, v in (d.get("extras") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(" ", k, v["ms_per_step"], "frac", (v.get("roofline") or {}).get("frac"), "cold", v.get("cold_first_step_ms"), "verified", (v.get("verified") or {}).get("ok"))
PY
python tools/skew_timing.py 26 2>/dev/null | tail -1 > $OUT/skew_timing_2p26.json; python tools/skew_timing.py 28 2>/dev/null | tail -1 > $OUT/skew_timing_2p28.json; el "skew timing done"
bash tools/pmc_all.sh r02z q1 q3 q3f cfg2 cfg3 cfg5 cfg5s > $OUT/pmc_all.log 2>&1; el "pmc_all exit $?"
grep -E "hbm|exit" $OUT/pmc_all.log | head -80
el "end"
