cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_parquet.py -x -q -m gpu 2>&1 | tail -3
for c in l_orderkey l_nullable; do
echo "== $c"
PLX_ZSTD_TIMING=1 timeout 300 python tools/zstd_read.py 2e7 2 0 $c 2>&1 | grep -v "^W\|^I" | tail -4
done
PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,zstd,zstd@host,zstd timeout 900 python tools/parquet_bench.py 2e7 > gpurun_out/r06h/pqbench.log 2> gpurun_out/r06h/pqbench.err
cut -c1-420 gpurun_out/r06h/pqbench.log
timeout 300 python tools/zstd_read.py 2e7 10 0 2>&1 | grep "read_ms\|matches" | tr '\n' ' '
