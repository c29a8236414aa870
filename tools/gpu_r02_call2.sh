#!/bin/bash
# Round 2, second GPU session: write-granularity microbenchmark, prefetch depth of the scatter pass, the reworked Q3 pipeline,
# plugin ABI tests, then the whole GPU suite with durations.  -> gpurun_out/r02b/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 120 tools/micro_scatter_bw.bin > $OUT/micro_scatter_bw.txt 2>&1; el "micro exit $?"; cat $OUT/micro_scatter_bw.txt | tee -a $OUT/summary.txt
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
run_w q3_new q3 X2=1
el "bench variants done"
timeout 200 python -m pytest tests/test_gpu_plugin_abi.py -m gpu -q --timeout 150 > $OUT/pytest_plugin.log 2>&1; el "plugin abi tests exit $?"
tail -6 $OUT/pytest_plugin.log
timeout 900 python -m pytest tests -m gpu -q --timeout 300 --durations=15 > $OUT/pytest_gpu.log 2>&1; el "gpu suite exit $?"
tail -25 $OUT/pytest_gpu.log
el "end"
