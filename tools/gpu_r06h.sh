cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_parquet.py tests/test_gpu_zzz_scan_host_paths.py -x -q -m gpu 2>&1 | tail -3
