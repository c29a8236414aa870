#!/usr/bin/env python
"""Writes tests/golden/pds_heads_schema.json: the Arrow schemas of the reference's own TPC-H sample tables
(/root/reference/examples/datasets/pds_heads/{lineitem,orders}.feather, 10 rows each).  Only the schema
(column names and Arrow types) is recorded -- no rows."""
import json
import os

import pyarrow as pa

HEADS = "/root/reference/examples/datasets/pds_heads"
out = {}
for t in ("lineitem", "orders"):
    with pa.OSFile(os.path.join(HEADS, t + ".feather"), "rb") as fh:
        tb = pa.ipc.open_file(fh).read_all()
    out[t] = {"source": f"examples/datasets/pds_heads/{t}.feather", "rows": tb.num_rows, "columns": [[n, str(ty)] for n, ty in zip(tb.schema.names, tb.schema.types)]}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pds_heads_schema.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", dst)
