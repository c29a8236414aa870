#!/bin/bash
# string-key group-by operator: first contact with hardware (tests, then config 5 from raw strings both ways)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
cd $R
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 300 python -m pytest tests/test_gpu_zzzz_round3_c.py -m gpu -q --timeout 120 > $OUT/pytest_strgroup.log 2>&1; echo "strgroup tests exit $?"; tail -25 $OUT/pytest_strgroup.log | cut -c1-400
unset PLX_SKIP_TORCH_PREIMPORT
PLX_STRGROUP_TIMING=1 timeout 200 python bench.py --workload cfg5s --steps 2 --warmup 1 --no-extras --no-cpu 2>&1 | grep "plx strgroup" | tail -2
timeout 300 python bench.py --workload cfg5s --steps 6 --warmup 2 --no-extras --no-cpu > $OUT/cfg5s_views.json 2> $OUT/cfg5s_views.err; echo "cfg5s (views) exit $?"
python - <<'PY' $OUT/cfg5s_views.json
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("metric", "value", "ms_per_step", "verified", "roofline")})
        print(json.dumps(d.get("kernels", d.get("kernel_breakdown")), indent=None)[:1500])
PY
tail -5 $OUT/cfg5s_views.err | cut -c1-300
