#!/bin/bash
# measurement: gen-3 scatter workgroups of 512 threads, two per CU, against the default (1024, one per CU)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/expb
for cfg in "0 1" "512 2" "512 3"; do
  set -- $cfg
  for wl in q3s cfg3 cfg5 cfg3s; do
    bwl=$wl; unset PLX_Q3_SHUFFLED; [ "$wl" = "q3s" ] && { bwl=q3; export PLX_Q3_SHUFFLED=1; }
    PLX_P3_BLOCK=$1 PLX_PART2_WGS_PER_CU=$2 PLX_BENCH_VERIFY=1 timeout 300 python bench.py --workload $bwl --no-extras --no-cpu --steps 5 --warmup 2 > gpurun_out/expb/${wl}_$1_$2.json 2> gpurun_out/expb/${wl}_$1_$2.err
    python - gpurun_out/expb/${wl}_$1_$2.json $wl $1 $2 <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    ks=sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['launches'])[:3]
    print(sys.argv[2:], d['ms_per_step'], d.get('verified',{}).get('ok'), [(k,round(v['avg_us'])) for k,v in ks])
except Exception as e: print(sys.argv[2:], 'failed', e)
PY
  done
done
