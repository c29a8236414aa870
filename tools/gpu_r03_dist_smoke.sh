#!/bin/bash
# Round 3: smoke runs of the N > 1 code paths of bench.py on a ONE-GPU box (nothing here is a scaling measurement):
#  (a) Q1 with two ranks sharing device 0, torch.distributed over gloo: shard generation, per-rank query, all-gather + combine of the partials, max-over-ranks timing
#  (b) the same with --scaling strong
#  (c) the sharded cfg3 / cfg5 operator at world size 1 through the library's RCCL communicator (self-exchange): pre-aggregation -> exchange -> merge at 1e9 rows,
#      rank-0 verification against the oracle, and --mode rows
#  (d) skew timing of the partitioned group-by (zipf / one hot key vs uniform) on the generation-3 scatter
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03x
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print({k: d.get(k) for k in ('value','n_gpus','ms_per_step','scaling','exchange_mode','shrink_estimate','partial_rows_per_rank','groups_total')}, 'cfg', d['config'].get('workload'), d['config'].get('rows_per_gpu'), 'verified', (d.get('verified') or {}).get('ok'), (d.get('verified') or {}).get('against','')[:60], 'shuffle', d.get('shuffle'))
" $1; }
PLX_DIST_BACKEND=gloo PLX_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 5 --warmup 2 --rows 100000000 > $OUT/q1_2ranks_weak.json 2> $OUT/q1_2ranks_weak.err; el "q1 two ranks (gloo, one device) exit $?"; show $OUT/q1_2ranks_weak.json
PLX_DIST_BACKEND=gloo PLX_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 5 --warmup 2 --scaling strong > $OUT/q1_2ranks_strong.json 2> $OUT/q1_2ranks_strong.err; el "q1 two ranks strong exit $?"; show $OUT/q1_2ranks_strong.json
for wl in cfg3 cfg5; do
  PLX_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --workload $wl > $OUT/${wl}_sharded_ws1.json 2> $OUT/${wl}_sharded_ws1.err; el "$wl sharded operator at world size 1 exit $?"; show $OUT/${wl}_sharded_ws1.json
done
PLX_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --workload cfg3 --mode rows --rows 200000000 > $OUT/cfg3_sharded_rows_ws1.json 2> $OUT/cfg3_sharded_rows_ws1.err; el "cfg3 sharded --mode rows exit $?"; show $OUT/cfg3_sharded_rows_ws1.json
timeout 300 python tools/skew_timing.py > $OUT/skew_timing_2p26.json 2> $OUT/skew.err; el "skew timing exit $?"; cut -c1-600 $OUT/skew_timing_2p26.json | head -5
tail -3 $OUT/*.err | cut -c1-300 | tail -30
el "end"
