#!/bin/bash
# string-key group-by with the hot-key path: tests (old and new), the sweep (default policy and forced operator), cfg5s
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03v
mkdir -p $OUT
cd $R
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 400 python -m pytest tests/test_gpu_zzzz_round3_c.py -m "gpu or gpu_unvalidated" -q --timeout 200 2>&1 | tail -15 | cut -c1-400
unset PLX_SKIP_TORCH_PREIMPORT
timeout 400 python tools/strgroup_sweep.py 26 > $OUT/sweep_default.json 2> $OUT/sweep_default.err; echo "sweep exit $?"
PLX_STRGROUP_FORCE=1 timeout 400 python tools/strgroup_sweep.py 26 > $OUT/sweep_forced.json 2> $OUT/sweep_forced.err; echo "forced sweep exit $?"
python - $OUT/sweep_default.json $OUT/sweep_forced.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); print(f.split("/")[-1])
    for k, v in d.items():
        if isinstance(v, dict): print("  ", k, {r: (x["ms_median"], x["groups"], x.get("operator"), x.get("spot_check_ok")) for r, x in v.items()}, {n: u for n, u in v["views"]["kernel_us"].items() if "strgroup" in n})
PY
timeout 300 python bench.py --workload cfg5s --steps 6 --warmup 2 --no-extras --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernels') or {}
        print(d['config']['workload'], d.get('ms_per_step'), (d.get('verified') or {}).get('ok'), {n: round(v['avg_us']) for n,v in list(k.items())[:3]})
"
