cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_parquet.py -x -q -m gpu 2>&1 | tail -3
PAGE_ROWS=1000000 timeout 600 python tools/zstd_read_plain.py 2e7 6 2>&1 | grep "^device\|^host" | cut -c1-200
PLX_PARQUET_ZSTD_HOST_SEQS=0 PAGE_ROWS=1000000 timeout 600 python tools/zstd_read_plain.py 2e7 6 2>&1 | grep "^device" | cut -c1-200
timeout 600 python tools/zstd_read_plain.py 2e7 6 2>&1 | grep "^device\|^host" | cut -c1-200
timeout 300 python tools/zstd_read.py 2e7 8 0 2>&1 | grep "read_ms\|matches" | tr '\n' ' '
