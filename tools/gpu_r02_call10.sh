#!/bin/bash
# Round 2, tenth GPU session: one flush per two-tile round (256 partitions, 4-line rings) vs per tile (512, 2-line); aggregation pass with three chunks in flight.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02j
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
run_w cfg5_np512 cfg5 X2=1
run_w cfg5_np256_t2 cfg5 PLX_PART_DIRECT_LOG2_PARTS=8
run_w cfg5_np256_t1 cfg5 PLX_PART_DIRECT_LOG2_PARTS=8 PLX_PART_TILES=1
run_w cfg3_np512 cfg3 X2=1
run_w cfg3_np256_t2 cfg3 PLX_PART_DIRECT_LOG2_PARTS=8
run_w cfg5s cfg5s X2=1
el "bench variants done"
timeout 400 python -m pytest tests/test_gpu_queries.py -m gpu -q --timeout 200 -x -k "partitioned or declared or skew or config" > $OUT/pytest_part.log 2>&1; el "partition tests exit $?"
tail -5 $OUT/pytest_part.log
timeout 300 python -m pytest tests/test_gpu_strview.py -m gpu -q --timeout 200 > $OUT/pytest_strview.log 2>&1; el "strview tests exit $?"
tail -3 $OUT/pytest_strview.log
el "end"
