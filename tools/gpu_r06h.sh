cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -3
for c in l_orderkey l_nullable; do
echo "== $c"
PLX_ZSTD_TIMING=1 timeout 300 python tools/zstd_read.py 2e7 2 0 $c 2>&1 | grep -v "^W\|^I" | tail -4
done
