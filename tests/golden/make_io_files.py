#!/usr/bin/env python
"""Copies the reference's own Parquet / Arrow IPC test fixtures into tests/golden/io_files/ (data files of a few hundred bytes to 2 KB,
written by Polars' writer ("Arrow2"), parquet-mr, parquet-cpp and Impala):

  py-polars/tests/unit/io/files/{small, foods1 (zstd), foods2 (LZ4_RAW), tz_aware (ns, UTC), empty_datapage_v2.snappy (an all-null
  v2 page without value bytes, test_parquet.py:848), nested_maps.snappy}.parquet, delta-table/*.parquet, iceberg-table/data/*/*.parquet (gzip),
  foods1.ipc, foods2.ipc; docs/assets/data/alltypes_plain.parquet (Impala: INT96 timestamps, un-annotated binary strings).

/root/reference does not exist on the GPU box, so tests read the committed copies: tests/test_parquet_emu_cpu.py runs the product's
reader over every column of them against pyarrow, tests/test_ipc_cpu.py the IPC host half."""
import glob
import hashlib
import json
import os
import shutil

REF = "/root/reference"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "io_files")
os.makedirs(dst, exist_ok=True)
files = sorted(glob.glob(REF + "/py-polars/tests/unit/io/files/**/*.parquet", recursive=True)) + sorted(glob.glob(REF + "/py-polars/tests/unit/io/files/*.ipc")) + \
    [REF + "/docs/assets/data/alltypes_plain.parquet"]
index = {}
for src in files:
    rel = os.path.relpath(src, REF)
    name = os.path.basename(src)
    if "delta-table" in rel:
        name = "delta_" + name[:19] + ".parquet"
    if "iceberg-table" in rel:
        name = "iceberg_" + rel.split("ts_day=")[1][:10] + ".parquet"
    shutil.copyfile(src, os.path.join(dst, name))
    os.chmod(os.path.join(dst, name), 0o644)
    index[name] = {"source": rel, "sha256": hashlib.sha256(open(src, "rb").read()).hexdigest()}
with open(os.path.join(dst, "INDEX.json"), "w") as f:
    json.dump(index, f, indent=1, sort_keys=True)
print("wrote", len(index), "files to", dst)
