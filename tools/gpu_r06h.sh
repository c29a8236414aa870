cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -3
