#!/bin/bash
# Round 3, GPU call 6: partitioned join probe (unordered probe keys), shuffled Q3 from the native generator with oracle verification
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
export PLX_JIT_VERBOSE=1
timeout 300 python -m pytest tests -m gpu_unvalidated -q --timeout 200 -x > $OUT/pytest_unvalidated.log 2>&1; el "gpu_unvalidated exit $?"
tail -25 $OUT/pytest_unvalidated.log | cut -c1-400
unset PLX_JIT_VERBOSE
run() {
  local wl=$1; shift
  echo "== $wl $*" | tee -a $OUT/variants.txt
  env "$@" PLX_BENCH_VERIFY_BUDGET_S=60 timeout 300 python $R/bench.py --workload $wl --steps 10 --warmup 3 --no-extras --no-cpu 2>$OUT/err_$wl.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'cold', d.get('cold_first_step_ms'), 'verified', (d.get('verified') or {}).get('ok'), 'frac', d['roofline']['frac'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $OUT/variants.txt
  tail -3 $OUT/err_$wl.txt | cut -c1-300
}
run q3 PLX_Q3_SHUFFLED=1
run q3 PLX_Q3_SHUFFLED=1 PLX_PROBE_PARTITIONED=0
run q3
run cfg3
run cfg5
el "end"
