#!/bin/bash
# Round 2, eighth GPU session: string table fix, build-scan OR merging on/off, scatter geometry + ablation runs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 300 python -m pytest tests/test_gpu_strview.py -m gpu -q --timeout 200 > $OUT/pytest_strview.log 2>&1; el "strview tests exit $?"
tail -5 $OUT/pytest_strview.log
timeout 300 python -m pytest tests/test_gpu_queries.py -m gpu -q --timeout 200 -x -k "q3 or join" > $OUT/pytest_join.log 2>&1; el "join tests exit $?"
tail -3 $OUT/pytest_join.log
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
run_w q3_merge q3 X2=1
run_w q3_nomerge q3 PLX_BUILD_MERGE=0
run_w q3f q3f X2=1
run_w cfg5s cfg5s X2=1
run_w cfg5_base cfg5 X2=1
run_w cfg5_np256_b512x2 cfg5 PLX_PART_DIRECT_LOG2_PARTS=8 PLX_PART_RING_LINES=2 PLX_PART_BLOCK=512 PLX_PART2_WGS_PER_CU=2
run_w cfg5_np256_b1024 cfg5 PLX_PART_DIRECT_LOG2_PARTS=8 PLX_PART_RING_LINES=2 PLX_PART_BLOCK=1024
run_w cfg5_np256_r4 cfg5 PLX_PART_DIRECT_LOG2_PARTS=8 PLX_PART_RING_LINES=4 PLX_PART_BLOCK=1024
run_w cfg5_np128_b256x4 cfg5 PLX_PART_DIRECT_LOG2_PARTS=7 PLX_PART_RING_LINES=2 PLX_PART_BLOCK=256 PLX_PART2_WGS_PER_CU=4
run_w cfg5_nostore cfg5 PLX_PART_ABLATE=1 PLX_BENCH_VERIFY=0
run_w cfg5_noring cfg5 PLX_PART_ABLATE=2 PLX_BENCH_VERIFY=0
run_w cfg5_neither cfg5 PLX_PART_ABLATE=3 PLX_BENCH_VERIFY=0
el "end"
