#!/bin/bash
# Per-step times of the headline after the profile events are pre-created; plus warm-up 8 as a control.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02u
mkdir -p $OUT
cd $R
PLX_BENCH_VERIFY=0 timeout 60 python bench.py --no-extras --no-cpu > $OUT/a.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/a.json')); print('stock', d['ms_per_step'], d['step_ms'])"
PLX_BENCH_VERIFY=0 timeout 60 python bench.py --no-extras --no-cpu --warmup 8 > $OUT/b.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/b.json')); print('warmup8', d['ms_per_step'], d['step_ms'])"
