cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_zzz_scan_host_paths.py -x -q -m gpu 2>&1 | tail -4
PLX_PQBENCH_NO_IPC=1 PLX_PQBENCH_CODECS=none,zstd,zstd@host timeout 900 python tools/parquet_bench.py 2e7 > gpurun_out/r06h/pqbench.log 2> gpurun_out/r06h/pqbench.err
cat gpurun_out/r06h/pqbench.log
tail -c 600 gpurun_out/r06h/pqbench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06h/prof -o zstd -- python $GRAFT_REPO_ROOT/tools/zstd_read.py 2e7 3 > $GRAFT_REPO_ROOT/gpurun_out/r06h/zstd_read.log 2>&1
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r06h/zstd_read.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r06h/prof/**/*kernel_trace.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    z = [r for r in rows if 'zstd' in r['Kernel_Name']]
    for r in z[-14:]:
        print(r['Kernel_Name'][:40], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms')
PY
