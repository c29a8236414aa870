cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_materialise.py tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -8
PLX_BENCH_EXTRAS=semim,filterm PLX_BENCH_Q3_SHUFFLED=0 PLX_BENCH_E2E=0 PLX_BENCH_SCAN=0 PLX_BENCH_DEADLINE_S=900 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_semim.log 2> gpurun_out/bench_semim.err
tail -c 800 gpurun_out/bench_semim.err
python - <<'PY'
import json
d=json.load(open('bench_extras.json'))
for k,v in d.get('extras',{}).items():
    if 'q1' in k: continue
    print(k, json.dumps({a:b for a,b in v.items() if a in ('ms_per_step','cold_first_step_ms','step_ms','result_rows','kernels','error','plan')})[:2200])
    print('  verified', json.dumps(v.get('verified'))[:600], 'frac', (v.get('roofline') or {}).get('frac'))
PY
