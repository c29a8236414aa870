#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/expw
python - <<'PY' 2>&1 | tail -12
import numpy as np, os, time
import polars_amd as pl
import bench
pl.init(0)
class A: pass
wl = bench.make_workload(pl, "cfg3w", int(os.environ.get("ROWS", 1000000000)), 1234)
out = wl.step()
print(pl.last_plan()[:900])
for lp in ("", "10"):
    if lp: os.environ["PLX_PART_LOG2_PARTS"] = lp
PY
for lp in 9 10; do
  PLX_PART_LOG2_PARTS=$lp PLX_BENCH_VERIFY=0 timeout 300 python bench.py --workload cfg3w --no-extras --no-cpu --steps 3 --warmup 1 > gpurun_out/expw/cfg3w_lp$lp.json 2> gpurun_out/expw/cfg3w_lp$lp.err
  python - gpurun_out/expw/cfg3w_lp$lp.json $lp <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    ks=sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['launches'])[:4]
    print(sys.argv[2:], d['ms_per_step'], [(k,v['launches'],round(v['avg_us'])) for k,v in ks])
except Exception as e: print(sys.argv[2:], 'failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done
