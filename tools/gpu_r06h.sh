cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_parquet.py tests/test_gpu_ipc.py -x -q -m gpu 2>&1 | tail -3
PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,snappy,zstd,none,snappy,zstd timeout 900 python tools/parquet_bench.py 2e7 > gpurun_out/r06h/pqbench.log 2> gpurun_out/r06h/pqbench.err
cut -c1-250 gpurun_out/r06h/pqbench.log
PLX_IO_THREADS=0 PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,zstd timeout 900 python tools/parquet_bench.py 2e7 2>/dev/null | cut -c1-250
PLX_PARQUET_TRACE=1 timeout 300 python tools/zstd_read.py 2e7 3 0 2>&1 | grep "of the walk" | tail -7
