cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_partition_geometry.py tests/test_gpu_datagen.py tests/test_gpu_queries.py tests/test_gpu_zz_full_size.py -x -q -m gpu 2>&1 | tail -8
for cfg in; do
  echo "=== PLX_PART_PAIR=$cfg"
  PLX_PART_PAIR=$cfg PLX_BENCH_EXTRAS=cfg3s PLX_BENCH_Q3_SHUFFLED=0 PLX_BENCH_E2E=0 PLX_BENCH_SCAN=0 PLX_BENCH_DEADLINE_S=900 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_cfg3s_$cfg.log 2> gpurun_out/bench_cfg3s_$cfg.err
  tail -c 500 gpurun_out/bench_cfg3s_$cfg.err
  python - <<'PY'
import json
d=json.load(open('bench_extras.json'))
for k,v in d.get('extras',{}).items():
    if 'q1' in k: continue
    print(k, json.dumps({a:b for a,b in v.items() if a in ('ms_per_step','cold_first_step_ms','one_shot_ms','step_ms','result_rows','kernels','error','plan')})[:1800])
    print('  verified', (v.get('verified') or {}).get('ok'), 'frac', (v.get('roofline') or {}).get('frac'))
PY
done
