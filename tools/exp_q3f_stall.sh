#!/bin/bash
# measurement: the three-table Q3 as a secondary workload behind the hashed-key Q3 (the sequence of the full bench run), with every mapping of fresh device
# memory logged (PLX_POOL_TRACE=1) and a time stamp per step on stderr
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/stall
PLX_POOL_TRACE=1 PLX_BENCH_EXTRAS=q3f PLX_BENCH_Q3_SHUFFLED=0 PLX_BENCH_E2E=0 PLX_BENCH_SCAN=0 PLX_BENCH_STEP_TRACE=1 timeout 600 python bench.py --workload q3h --no-cpu --steps 5 --warmup 2 > gpurun_out/stall/out.json 2> gpurun_out/stall/err.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/stall/out.json') if l.startswith('{')][-1])
v=d['extras'].get('tpch_q3_three_tables_sf100', {})
print('q3h', d['ms_per_step'], 'q3f', v.get('ms_per_step'), v.get('step_ms'))
PY
grep -n "plx pool\|STEP" gpurun_out/stall/err.log | tail -60
