#!/bin/bash
# Round 2, first GPU session: the second-generation partitioned group-by (correctness first, then timings and a knob sweep),
# the plugin-ABI tests, and one full bench line with the oracle verification of every workload.  -> gpurun_out/r02a/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
# 1. correctness of the new kernels (small tests first: a hang or a crash must not eat the session)
timeout 420 python -m pytest tests/test_gpu_queries.py -m gpu -q --timeout 200 -x -k "partitioned or declared or skew" --durations=8 > $OUT/pytest_part.log 2>&1; el "partition tests exit $?"
tail -12 $OUT/pytest_part.log
timeout 200 python -m pytest tests/test_gpu_plugin_abi.py -m gpu -q --timeout 150 > $OUT/pytest_plugin.log 2>&1; el "plugin abi tests exit $?"
tail -6 $OUT/pytest_plugin.log
# 2. cfg3 / cfg5 at 1e9 rows: v2 (default), v1 for comparison; verification on
run() {   # name, env..., -- workload
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu > $OUT/$name.json 2> $OUT/$name.err
  local rc=$?
  python - "$OUT/$name.json" "$name" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms/step", d["ms_per_step"], "cold", d.get("cold_first_step_ms"), "frac", d["roofline"]["frac"], {k: v["avg_us"] for k, v in d["kernels"].items()}, "verified", (d.get("verified") or {}).get("ok"), (d.get("verified") or {}).get("error", ""))
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
  [ $rc -ne 0 ] && tail -5 $OUT/$name.err
}
W3="--workload cfg3"; W5="--workload cfg5"
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
run_w cfg3_v2 cfg3 PLX_PART_V=2; el "cfg3 v2 done"
run_w cfg5_v2 cfg5 PLX_PART_V=2; el "cfg5 v2 done"
run_w cfg3_v1 cfg3 PLX_PART_V=1 PLX_BENCH_VERIFY=0; el "cfg3 v1 done"
run_w cfg5_v1 cfg5 PLX_PART_V=1 PLX_BENCH_VERIFY=0; el "cfg5 v1 done"
grep -h "partitioned" $OUT/cfg3_v2.err $OUT/cfg5_v2.err 2>/dev/null | head -3
# 3. knob sweep (v2, no verification)
for kn in "PLX_PART_BLOCK=512" "PLX_PART_BLOCK=256" "PLX_PART_LOG2_PARTS=8" "PLX_PART_HOT=0" ; do
  run_w "cfg3_$(echo $kn | tr '=' '_')" cfg3 PLX_BENCH_VERIFY=0 $kn
done
for kn in "PLX_PART_DIRECT=0" "PLX_PART_BLOCK=512" "PLX_PART_RING_LINES=2" "PLX_PART_HOT=0"; do
  run_w "cfg5_$(echo $kn | tr '=' '_')" cfg5 PLX_BENCH_VERIFY=0 $kn
done
el "sweep done"
# 4. the full default bench line (headline Q1 + CPU baseline on the same rows + extras, every workload verified)
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "full bench exit $?"
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
o = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r02a/"
try:
    d = json.load(open(o + "bench_full.json"))
    print("Q1", d["ms_per_step"], d["roofline"]["frac"], "verified", d.get("verified"), "cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "kind", "cores", "seconds")}, d["cpu_baseline"].get("polars"))
    for k, v in d.get("extras", {}).items():
        print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), "cold", v.get("cold_first_step_ms"), "verified", v.get("verified"), v.get("error"))
    print("note", d.get("note"))
except Exception as e:
    print("bench_full unreadable", e)
PY
tail -5 $OUT/bench_full.err
el "end"
