cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_materialise.py tests/test_gpu_join_partitioned.py -x -q -m gpu -k "filled_from_lds" 2>&1 | tail -30
