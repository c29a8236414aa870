#!/bin/bash
# the two-column-key group-by (cfg3w) and the string-key group-by (cfg5s) after a change to the LDS slot protocols; PLX_PART_ABLATE 4 / 8 take parts of the wide-key
# aggregation pass out (results wrong, not verified)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/expw
for spec in ${@:-cfg3w:0 cfg5s:0}; do
  wl=${spec%%:*}; ab=${spec#*:}
  v=1; [ "$ab" != "0" ] && v=0
  PLX_PART_ABLATE=$ab PLX_BENCH_VERIFY=$v timeout 300 python bench.py --workload $wl --no-extras --no-cpu --steps 3 --warmup 1 > gpurun_out/expw/${wl}_ab$ab.json 2> gpurun_out/expw/${wl}_ab$ab.err
  python - gpurun_out/expw/${wl}_ab$ab.json $wl $ab <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    ks=sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['launches'])[:4]
    print(sys.argv[2:], d['ms_per_step'], d.get('verified',{}).get('ok'), [(k,v['launches'],round(v['avg_us'])) for k,v in ks])
except Exception as e: print(sys.argv[2:], 'failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done
