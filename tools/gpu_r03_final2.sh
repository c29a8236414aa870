#!/bin/bash
# Round 3, second closing session (after the string-key group-by landed): counters + kernel stats of cfg5s on the final kernels, the whole GPU suite,
# smoke(), the full default bench line (what the driver runs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03w
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
bash tools/pmc_all.sh r03w cfg5s > $OUT/pmc_all.log 2>&1; el "pmc refresh exit $?"
grep -E "^(cfg3|cfg5|q3s|cfg5s) " $OUT/pmc_all.log | head -8
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu_all.log 2>&1; el "whole gpu suite exit $?"
tail -4 $OUT/pytest_gpu_all.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; el "smoke exit $?"; tail -2 $OUT/smoke.log | cut -c1-300
timeout 400 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "bench exit $?"
python - $OUT/bench_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", d["config"]["workload"], "ms/step", d["ms_per_step"], "median", d.get("ms_per_step_median"), "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "verified", (d.get("verified") or {}).get("ok"))
for k, v in (d.get("extras") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        r = v.get("roofline") or {}
        print(" ", k, v["ms_per_step"], "frac", r.get("frac"), "hbm_frac", r.get("hbm_frac"), "traffic", r.get("traffic"), "verified", (v.get("verified") or {}).get("ok"), "cold", v.get("cold_first_step_ms"))
    elif isinstance(v, dict) and "files" in v:
        for fk, fv in v["files"].items():
            print("  scan", fk, fv.get("read_ms"), "ms", fv.get("file_GBps"), "GB/s file; pyarrow", fv.get("pyarrow_read_ms"), "ms; verified", fv.get("verified"))
    elif isinstance(v, dict) and "pcie_inclusive_GBps" in v:
        print(" ", k, v)
    elif isinstance(v, dict) and "error" in v:
        print(" ", k, "ERROR", v["error"])
PY
el "end"
