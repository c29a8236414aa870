cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
PLX_JOIN_PART_BUILD=2 timeout 900 python -m pytest tests/test_gpu_materialise.py tests/test_gpu_queries.py tests/test_gpu_join_partitioned.py tests/test_gpu_join_duplicate_keys.py -x -q -m gpu 2>&1 | tail -4
for cfg in "A"; do
  echo "=== $cfg"
  if [ $cfg = B ]; then export PLX_JIT_DEFINES=-DPLX_NO_TWO_TILES; export PLX_JIT_CACHE_DIR=/tmp/jitB; fi
  PLX_BENCH_EXTRAS=q3h,q3dc,q3d,joinmh PLX_BENCH_Q3_SHUFFLED=0 PLX_BENCH_E2E=0 PLX_BENCH_SCAN=0 PLX_BENCH_DEADLINE_S=900 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/r06d/bench_$cfg.log 2> gpurun_out/r06d/bench_$cfg.err
  tail -c 400 gpurun_out/r06d/bench_$cfg.err
  python - <<'PY'
import json
d=json.load(open('bench_extras.json'))
for k,v in d.get('extras',{}).items():
    if 'q1' in k: continue
    print(k, json.dumps({a:b for a,b in v.items() if a in ('ms_per_step','cold_first_step_ms','step_ms','result_rows','kernels','error','plan')})[:2200])
    print('  verified', (v.get('verified') or {}).get('ok'), 'frac', (v.get('roofline') or {}).get('frac'))
PY
done
