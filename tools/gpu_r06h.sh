cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp polars_amd/libpolars_amd.so /tmp/final.so
for r in far notog far notog; do
cp build/zlibs/lib$r.so polars_amd/libpolars_amd.so
echo "== $r"
PLX_ZSTD_TIMING=1 timeout 600 python tools/zstd_read_plain.py 2e7 1 2>&1 | grep "pq_zstd" | head -6 | sed 's/.*matches=\([0-9]*\).*/\1/' | sort -n | tr '\n' ' '
echo
timeout 600 python tools/zstd_read_plain.py 2e7 6 2>&1 | grep "^device" | cut -c1-200
done
cp /tmp/final.so polars_amd/libpolars_amd.so
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_parquet.py -x -q -m gpu 2>&1 | tail -3
