cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PLX_PARQUET_TRACE=1 timeout 300 python tools/zstd_read.py 2e7 4 0 2>&1 | grep -v "^W\|^I" | tail -44
