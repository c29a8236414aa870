cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06i
bash tools/gpu_full.sh
PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,snappy,zstd,zstd@host,zstd,zstd@host timeout 900 python tools/parquet_bench.py 2e7 > gpurun_out/r06i/scan_codecs.jsonl 2> gpurun_out/r06i/pqbench.err
cut -c1-300 gpurun_out/r06i/scan_codecs.jsonl
( for c in l_orderkey l_extendedprice l_shipdate l_nullable; do PLX_ZSTD_TIMING=1 timeout 300 python tools/zstd_read.py 2e7 2 0 $c 2>&1 | grep "pq_zstd\|read_ms" | tail -2; done ) > gpurun_out/r06i/scan_zstd_phase_clock.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06i/prof -o scan_zstd -- python $GRAFT_REPO_ROOT/tools/zstd_read.py 2e7 10 > $GRAFT_REPO_ROOT/gpurun_out/r06i/zstd_read.log 2>&1
grep "read_ms\|matches" $GRAFT_REPO_ROOT/gpurun_out/r06i/zstd_read.log | tr '\n' ' '
ls $GRAFT_REPO_ROOT/gpurun_out/r06i/prof
