cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_parquet.py -x -q -m gpu 2>&1 | tail -3
for b in 0 16777216 33554432 67108864; do
echo "== batch $b"
PLX_PARQUET_ZSTD_BATCH=$b timeout 300 python tools/zstd_read.py 2e7 6 0 2>&1 | grep -v "^W\|^I" | tail -7 | tr '\n' ' '
done
echo
PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,zstd,zstd@host,zstd timeout 900 python tools/parquet_bench.py 2e7 > gpurun_out/r06h/pqbench.log 2> gpurun_out/r06h/pqbench.err
cut -c1-420 gpurun_out/r06h/pqbench.log
