#!/bin/bash
# Round 2, fourth GPU session: three-table Q3, RCCL exchange at world size 1, the full suite, the full bench, rocprofv3 evidence.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 300 python -m pytest tests/test_gpu_queries.py tests/test_gpu_datagen.py tests/test_gpu_kernels.py -m gpu -q --timeout 200 -x -k "three_tables or customer or rccl" > $OUT/pytest_new.log 2>&1; el "new tests exit $?"
tail -15 $OUT/pytest_new.log
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; el "gpu suite exit $?"
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "full bench exit $?"
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
o = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r02d/"
try:
    d = json.load(open(o + "bench_full.json"))
    print("Q1", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], "verified", (d.get("verified") or {}).get("ok"), "cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "kind", "cores", "seconds")})
    for k, v in d.get("extras", {}).items():
        print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), "cold", v.get("cold_first_step_ms"), "verified", (v.get("verified") or {}).get("ok"), v.get("error"), {a: b["avg_us"] for a, b in v.get("kernels", {}).items()})
    print(d.get("note"))
except Exception as e:
    print("bench_full unreadable", e)
PY
tail -3 $OUT/bench_full.err
bash tools/pmc_all.sh r02d q1 q3 q3f cfg2 cfg3 cfg5 > $OUT/pmc_all.log 2>&1; el "pmc exit $?"
grep -E "hbm|exit" $OUT/pmc_all.log | tail -60
el "end"
