#!/bin/bash
# repeat the verified full-size workloads and the new kernels' tests: sporadic wrong answers (races in the claim protocols / the scan) would show here
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03s
mkdir -p $OUT
cd $R
export PLX_SKIP_TORCH_PREIMPORT=1
for I in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_zzzz_round3_c.py tests/test_gpu_zzzz_round3_d.py -m gpu -q --timeout 200 2>&1 | tail -1
  for WL in cfg3 cfg5 cfg5s q3s; do
    BWL=$WL; unset PLX_Q3_SHUFFLED
    if [ "$WL" = "q3s" ]; then BWL=q3; export PLX_Q3_SHUFFLED=1; fi
    timeout 300 python bench.py --workload $BWL --steps 6 --warmup 2 --no-extras --no-cpu > $OUT/$WL.$I.json 2> $OUT/$WL.$I.err; echo "pass $I $WL exit $? $(grep -o '"ok": [a-z]*' $OUT/$WL.$I.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $OUT/$WL.$I.json | head -1)"
  done
done
