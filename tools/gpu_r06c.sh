cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/pmc_all.sh r06q q3d 2>&1 | tail -12
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_materialise.py -x -q -m gpu --durations=8 2>&1 | tail -16
