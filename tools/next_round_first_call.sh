#!/bin/bash
# First GPU session of the next round: validates on hardware everything that was written after round 1's GPU budget ran out,
# then collects the measurements the next kernel work starts from.  Everything lands in gpurun_out/first_call/.
# usage: /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/next_round_first_call.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/first_call
mkdir -p $OUT
cd $R
# 1. the tests marked non-strict xfail (generator kernels, parquet scan): run them for real
timeout 120 python -m pytest tests/test_gpu_datagen.py tests/test_gpu_io.py tests/test_gpu_null_exprs.py --runxfail -q --timeout 100 > $OUT/unverified_tests.log 2>&1; echo "unverified tests exit $?" | tee -a $OUT/summary.txt
# 2. the whole GPU suite
timeout 300 python -m pytest tests -m gpu -q --timeout 120 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "gpu suite exit $?" | tee -a $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log
# 3. bench through the guard with the library's generators, then with the torch generators (same kernels, different inputs)
timeout 240 python bench.py > $OUT/bench_native.json 2> $OUT/bench_native.err; echo "bench (native datagen, guarded) exit $?" | tee -a $OUT/summary.txt
grep -c "native data generator unavailable" $OUT/bench_native.err | sed 's/^/fallbacks to torch generators: /' | tee -a $OUT/summary.txt
PLX_BENCH_DATAGEN=torch PLX_BENCH_GUARD=0 timeout 240 python bench.py --no-extras > $OUT/bench_torch.json 2> $OUT/bench_torch.err; echo "bench (torch datagen) exit $?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json, os
o = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/first_call/"
for f in ("bench_native.json", "bench_torch.json"):
    try:
        d = json.load(open(o + f))
        print(f, d["ms_per_step"], d["roofline"]["frac"], d["config"]["description"][-60:], {k: v.get("ms_per_step") for k, v in d.get("extras", {}).items()}, d.get("note"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
# 4. optional full-size properties, partition sweep, sort
PLX_FULL_SIZE=1 timeout 200 python -m pytest tests/test_gpu_zz_full_size.py -q --timeout 180 > $OUT/full_size.log 2>&1; echo "full-size tests exit $?" | tee -a $OUT/summary.txt
timeout 300 bash tools/part_sweep.sh cfg3 > $OUT/part_sweep.log 2>&1; cp $R/gpurun_out/part_sweep.txt $OUT/ 2>/dev/null
timeout 100 python tools/sort_bench.py > $OUT/sort_bench.json 2>/dev/null
cat $OUT/summary.txt
