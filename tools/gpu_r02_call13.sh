#!/bin/bash
# Round 2, thirteenth GPU session: workgroups per CU, per scan kind.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02m
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
run_w() { local name=$1; local wl=$2; shift 2; ( export "$@" X=1; timeout 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-extras --no-cpu > $OUT/$name.json 2> $OUT/$name.err ); rc=$?
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
for b in 4 5 6 7 5 4; do run_w q1_bpc${b}_$RANDOM q1 PLX_BPC_LDSAGG=$b PLX_BENCH_VERIFY=0; done
for b in 4 5 6 7 5 4; do run_w cfg2_bpc${b}_$RANDOM cfg2 PLX_BPC_REGAGG=$b PLX_BENCH_VERIFY=0; done
for b in 4 5 6 8; do run_w q3_build$b q3 PLX_BPC_DIRECT_BUILD=$b PLX_BENCH_VERIFY=0; done
for b in 8 10 12 16 8; do run_w q3_probe${b}_$RANDOM q3 PLX_BPC_DIRECT_PROBE=$b PLX_BPC_DIRECT_BUILD=5 PLX_BENCH_VERIFY=0; done
el "end"
