cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06full
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06full/gpu_tests.log
cat gpurun_out/r06full/gpu_tests.log
( time python bench.py ) > gpurun_out/r06full/bench_default.log 2> gpurun_out/r06full/bench_default.err
tail -c 600 gpurun_out/r06full/bench_default.err
tail -c 4200 gpurun_out/r06full/bench_default.log
cp bench_extras.json gpurun_out/r06full/bench_default_run.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
