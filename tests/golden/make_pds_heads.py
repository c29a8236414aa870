#!/usr/bin/env python
"""Copies the reference's own TPC-H sample tables (Arrow IPC files WRITTEN BY the reference's IPC writer, 10 rows each)
into tests/golden/pds_heads/ as read-only fixtures: /root/reference/examples/datasets/pds_heads/{lineitem,orders,customer}.feather.
They are data, ~13 KB together; /root/reference does not exist on the GPU box, so the tests read the committed copies.
tests/test_ipc_cpu.py checks the library's IPC reader against them buffer by buffer, tests/test_gpu_ipc.py runs Q1 / Q3 on them."""
import hashlib
import json
import os
import shutil

HEADS = "/root/reference/examples/datasets/pds_heads"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pds_heads")
os.makedirs(dst, exist_ok=True)
sums = {}
for t in ("lineitem", "orders", "customer"):
    shutil.copyfile(os.path.join(HEADS, t + ".feather"), os.path.join(dst, t + ".feather"))
    os.chmod(os.path.join(dst, t + ".feather"), 0o644)
    sums[t + ".feather"] = hashlib.sha256(open(os.path.join(dst, t + ".feather"), "rb").read()).hexdigest()
with open(os.path.join(dst, "SHA256.json"), "w") as f:
    json.dump({"source": "examples/datasets/pds_heads", "sha256": sums}, f, indent=1)
print("wrote", dst, sums)
