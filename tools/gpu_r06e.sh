cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_datagen.py -x -q -m gpu -k "48_bit" 2>&1 | tail -15
