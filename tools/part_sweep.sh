#!/bin/bash
# Partitioned group-by tuning sweep (BASELINE config 3: 1e9 rows, 1e6 keys): one bench line per knob setting.
# usage (on the GPU box): bash tools/part_sweep.sh [workload=cfg3]   -> gpurun_out/part_sweep.txt
WL=${1:-cfg3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/part_sweep.txt
mkdir -p $R/gpurun_out; : > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout 120 python $R/bench.py --workload $WL --steps 5 --warmup 1 --no-extras --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})" >> $OUT
}
run PLX_PART_WGS_PER_CU=2 PLX_PART_BUF_ROWS=8
run PLX_PART_WGS_PER_CU=1 PLX_PART_BUF_ROWS=8
run PLX_PART_WGS_PER_CU=1 PLX_PART_BUF_ROWS=16
run PLX_PART_WGS_PER_CU=2 PLX_PART_BUF_ROWS=4
run PLX_PART_LOG2_PARTS=10 PLX_PART_BUF_ROWS=4
run PLX_PART_LOG2_PARTS=10 PLX_PART_BUF_ROWS=8
cat $OUT
