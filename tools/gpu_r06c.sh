cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 1500 python -m pytest tests/test_gpu_materialise.py -x -q -m gpu 2>&1 | tail -8
PLX_BENCH_EXTRAS=joinm PLX_BENCH_Q3_SHUFFLED=0 PLX_BENCH_E2E=0 PLX_BENCH_SCAN=0 PLX_BENCH_DEADLINE_S=900 timeout 1200 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/r06c/bench.log 2> gpurun_out/r06c/bench.err
tail -c 1000 gpurun_out/r06c/bench.err
python - <<'PY'
import json
d=json.load(open('bench_extras.json'))
for k,v in d.get('extras',{}).items():
    print(k, json.dumps({a:b for a,b in v.items() if a in ('ms_per_step','cold_first_step_ms','step_ms','result_rows','kernels','verified','error','plan')})[:2200])
    print('   frac', (v.get('roofline') or {}).get('frac'))
PY
