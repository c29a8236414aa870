#!/bin/bash
# first-run (one_shot_ms / cold) figures of the group-by workloads with and without the up-front range pass for dense-looking keys
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/oneshot
for lr in 1 0; do
  for wl in cfg3 cfg3z cfg3s cfg5; do
    PLX_LEARN_DENSE_RANGE=$lr timeout 300 python bench.py --workload $wl --no-extras --no-cpu --steps 3 --warmup 2 > gpurun_out/oneshot/${wl}_$lr.json 2> gpurun_out/oneshot/${wl}_$lr.err
    python - gpurun_out/oneshot/${wl}_$lr.json $wl $lr <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2:], 'steady', d['ms_per_step'], 'one_shot', d.get('one_shot_ms'), 'cold', d.get('cold_first_step_ms'), d.get('verified',{}).get('ok'))
except Exception as e: print(sys.argv[2:], 'failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
  done
done
python -m pytest tests/test_gpu_datagen.py tests/test_gpu_partition_geometry.py -x -q -p no:cacheprovider 2>&1 | tail -3
