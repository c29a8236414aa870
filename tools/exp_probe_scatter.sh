#!/bin/bash
# measurement: what the join probe's scatter spends its time on -- PLX_PART_ABLATE 1 = no HBM stores of the records, 2 = no LDS tile writes, 3 = neither
# (the records are garbage then: no partition keeps any, the join finds nothing; not verified)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/exps
for ab in 0 1 2 3; do
  v=1; [ "$ab" != "0" ] && v=0
  PLX_PART_ABLATE=$ab PLX_BENCH_VERIFY=$v timeout 300 python bench.py --workload q3h --no-extras --no-cpu --steps 3 --warmup 1 > gpurun_out/exps/q3h_ab$ab.json 2> gpurun_out/exps/q3h_ab$ab.err
  python - gpurun_out/exps/q3h_ab$ab.json $ab <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    ks=sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['launches'])[:3]
    print(sys.argv[2:], d['ms_per_step'], [(k,v['launches'],round(v['avg_us'])) for k,v in ks])
except Exception as e: print(sys.argv[2:], 'failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done
python -m pytest tests/test_gpu_strgroup.py -x -q -p no:cacheprovider -k "reference_vector or null_string" 2>&1 | tail -2
