cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
for c in l_orderkey; do
echo "== $c"
PLX_ZSTD_TIMING=1 timeout 300 python tools/zstd_read.py 2e7 2 0 $c 2>&1 | grep -v "^W\|^I" | tail -4
done
PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,zstd,zstd@host timeout 900 python tools/parquet_bench.py 2e7 > gpurun_out/r06h/pqbench.log 2> gpurun_out/r06h/pqbench.err
cat gpurun_out/r06h/pqbench.log
tail -c 600 gpurun_out/r06h/pqbench.err
cd /tmp
PLX_PARQUET_TRACE=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06h/prof -o zstd -- python $GRAFT_REPO_ROOT/tools/zstd_read.py 2e7 3 > $GRAFT_REPO_ROOT/gpurun_out/r06h/zstd_read.log 2>&1
grep -v "^W\|^I\|^E" $GRAFT_REPO_ROOT/gpurun_out/r06h/zstd_read.log | tail -60
