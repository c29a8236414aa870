#!/bin/bash
# Round 2, third GPU session: compact register file (all fused kernels), scatter prefetch depth, Q3 with per-word u32 ranks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
run_w() { local name=$1; local wl=$2; shift 2; ( export "$@" X=1; timeout 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-extras --no-cpu > $OUT/$name.json 2> $OUT/$name.err ); rc=$?
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
run_w cfg3_depth2 cfg3 PLX_PART_PREFETCH=2
run_w cfg3_depth1 cfg3 PLX_PART_PREFETCH=1 PLX_BENCH_VERIFY=0
run_w cfg5_depth2 cfg5 PLX_PART_PREFETCH=2
run_w cfg5_depth1 cfg5 PLX_PART_PREFETCH=1 PLX_BENCH_VERIFY=0
run_w q3_rank32 q3 X2=1
run_w q1 q1 X2=1
run_w cfg2 cfg2 X2=1
el "bench variants done"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; el "gpu suite exit $?"
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "full bench exit $?"
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
o = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r02c/"
try:
    d = json.load(open(o + "bench_full.json"))
    print("Q1", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], "verified", (d.get("verified") or {}).get("ok"), "cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "kind", "cores", "seconds")})
    for k, v in d.get("extras", {}).items():
        print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), "cold", v.get("cold_first_step_ms"), "verified", (v.get("verified") or {}).get("ok"), v.get("error"), {a: b["avg_us"] for a, b in v.get("kernels", {}).items()})
except Exception as e:
    print("bench_full unreadable", e)
PY
el "end"
