#!/bin/bash
# Round 3, GPU call 4: the third-generation partition scatter (tile sort + carry lines, packed records) in the engine.
# correctness first (whole -m gpu suite: the 2^25-row skew tests and the 1e9-row oracle comparisons run gen 3 by default), then cfg3 / cfg5 variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
timeout 500 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu_all.log 2>&1; el "whole gpu suite exit $?"
tail -6 $OUT/pytest_gpu_all.log | cut -c1-300
run() {
  local wl=$1; shift
  echo "== $wl $*" | tee -a $OUT/variants.txt
  env "$@" PLX_BENCH_VERIFY_BUDGET_S=60 timeout 200 python $R/bench.py --workload $wl --steps 10 --warmup 3 --no-extras --no-cpu 2>$OUT/err_$wl.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'cold', d.get('cold_first_step_ms'), 'verified', (d.get('verified') or {}).get('ok'), 'frac', d['roofline']['frac'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $OUT/variants.txt
}
run cfg3 PLX_PART_GEN=2
run cfg3
run cfg3 PLX_PART_PACK=1
run cfg3 PLX_PART_PACK=0
run cfg3 PLX_PART_TILES=2
run cfg3 PLX_SAMPLE_CACHE=0
run cfg5 PLX_PART_GEN=2
run cfg5
run cfg5 PLX_PART_TILES=2
el "variants done"
python tools/cfg3_run.py 1000000000 5 > $OUT/cfg3_plan.txt 2>&1; head -3 $OUT/cfg3_plan.txt | cut -c1-400
el "end"
