#!/bin/bash
# Why was the headline step 6.1 ms in the full run and 4.5 ms with --no-extras?  Per-step times, three ways.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02t
mkdir -p $OUT
cd $R
for tag in a b; do
  PLX_BENCH_VERIFY=0 timeout 60 python bench.py --no-extras --no-cpu > $OUT/noextras_$tag.json 2>/dev/null
  python -c "import json,sys; d=json.load(open('$OUT/noextras_$tag.json')); print('noextras_$tag', d['ms_per_step'], d['step_ms'])"
done
timeout 150 python bench.py > $OUT/full.json 2> $OUT/full.err
python -c "
import json
d=json.loads([l for l in open('$OUT/full.json') if l.startswith('{')][-1]); print('full', d['ms_per_step'], d['step_ms'])"
