cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 1500 python tools/exp_filter.py 2>&1 | tee gpurun_out/r06b/exp_filter.log | tail -30
timeout 1200 python -m pytest tests/test_gpu_datagen.py -x -q -m gpu 2>&1 | tail -15
