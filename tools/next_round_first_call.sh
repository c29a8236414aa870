#!/bin/bash
# First GPU session of the next round: validates what round 2 could only check on the CPU (no GPU minutes were left), then collects the
# starting measurements.  ~8 GPU-minutes.  usage: gpurun --timeout 1300 -- 'bash tools/next_round_first_call.sh'
#
# CPU-verified only at the end of round 2 (the device paths they reuse had run on hardware; the new code in them is host code):
#   * Parquet: ZSTD / GZIP / LZ4_RAW pages (host_codecs.hpp -> chunk image -> the uncompressed device path)
#   * Parquet: string columns with PLAIN pages (host views -> plx_strview_dict_encode)
#   * Parquet: DELTA_* / BYTE_STREAM_SPLIT / INT96 columns (host decode, one upload)
#   * Arrow IPC: LZ4-frame / ZSTD bodies
#   * scans over several files (plx_frame_concat + dictionary unification)
#   * Q1 / Q3 over the reference's own TPC-H sample files (tests/golden/pds_heads, through scan_ipc)
#   * slice pushdown into scans, Datetime ms / ns / time zones, INT96 -> ns, logical Arrow export, write_parquet / write_ipc, concat / IR::Union,
#     hive-partitioned directories, row-group shards, the reference's own Parquet / IPC fixture files (tests/golden/io_files)
#   all of the above: tests/test_gpu_zzz_scan_host_paths.py (sorted last on purpose)
#   * pq_snappy_kernel_v2 (PLX_SNAPPY_KERNEL=2: batched LDS loads in next / mark / rank) and PLX_PARQUET_SNAPPY=host, timed beside the default by tools/parquet_bench.py
#   * bench.py extras.parquet_ipc_scan_2e7_rows (scan_extra)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
t0=$(date +%s)
el() { echo "[+$(( $(date +%s) - t0 ))s] $*" | tee -a $OUT/summary.txt; }
export PLX_SKIP_TORCH_PREIMPORT=1
timeout 200 python -m pytest tests -m gpu_unvalidated -q --timeout 90 > $OUT/pytest_unvalidated.log 2>&1; el "gpu_unvalidated tests exit $?"
tail -15 $OUT/pytest_unvalidated.log | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_ipc.py tests/test_gpu_io.py tests/test_gpu_zzz_scan_host_paths.py -m gpu -q --timeout 90 --durations=5 > $OUT/pytest_scan.log 2>&1; el "scan gpu tests exit $?"
tail -15 $OUT/pytest_scan.log | cut -c1-250
PLX_SNAPPY_KERNEL=2 timeout 120 python -m pytest tests/test_gpu_parquet.py -m gpu -q --timeout 90 > $OUT/pytest_snappy_v2.log 2>&1; el "snappy kernel v2 gpu tests exit $?"
tail -3 $OUT/pytest_snappy_v2.log | cut -c1-250
unset PLX_SKIP_TORCH_PREIMPORT
PLX_SNAPPY_TIMING=1 timeout 360 python tools/parquet_bench.py 2e7 > $OUT/parquet_bench.jsonl 2> $OUT/parquet_bench.err; el "scan bench exit $?"
cut -c1-330 $OUT/parquet_bench.jsonl; grep pq_snappy $OUT/parquet_bench.err | tail -3 | cut -c1-400
timeout 240 python tools/q1_from_files.py 3e7 8 > $OUT/q1_from_files.jsonl 2> $OUT/q1_from_files.err; el "q1 from parquet files exit $?"
cut -c1-400 $OUT/q1_from_files.jsonl
timeout 200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; el "bench exit $?"
python - $OUT/bench_full.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline", d["config"]["workload"], "ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "verified", (d.get("verified") or {}).get("ok"))
print("step_ms", d.get("step_ms"))
for k, v in (d.get("extras") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(" ", k, v["ms_per_step"], "frac", (v.get("roofline") or {}).get("frac"), "verified", (v.get("verified") or {}).get("ok"))
    elif isinstance(v, dict) and "files" in v:
        for fk, fv in v["files"].items():
            print("  scan", fk, fv.get("read_ms"), "ms", fv.get("file_GBps"), "GB/s file; pyarrow", fv.get("pyarrow_read_ms"), "ms; verified", fv.get("verified"))
    elif isinstance(v, dict) and "error" in v:
        print(" ", k, "ERROR", v["error"])
PY
bash tools/pmc_all.sh r03a q3 q3f cfg3 cfg5 > $OUT/pmc_all.log 2>&1; el "pmc refresh exit $? (copy gpurun_out/r03a/profiles/* to profiles/)"
timeout 400 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu_all.log 2>&1; el "whole gpu suite exit $?"
tail -4 $OUT/pytest_gpu_all.log
el "end"
