"""Body of tests/test_gpu_kernels.py::test_library_exchange_on_rccl_world_size_one (run as a script in its own process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pl  # noqa: E402
from polars_amd import dist as pdist, queries  # noqa: E402



def stage(msg):                      # progress on stderr: a timeout in the parent then shows where the worker stopped
    print("[rccl_worker]", msg, file=sys.stderr, flush=True)


pl.init(0)
stage("library initialised")
rng = np.random.default_rng(71)
n = 1_000_003
key = rng.integers(0, 50_000, n).astype(np.int64)
v = rng.integers(-100, 100, n).astype(np.int64)
x = rng.uniform(0, 1, n)
c8 = rng.integers(0, 200, n).astype(np.uint8)
df = pl.DataFrame({"key": key, "v": v, "x": x, "c8": c8})
stage("creating the communicator")
comm = pdist.LibComm(pl)
stage("communicator up")
assert (comm.rank, comm.world_size) == (0, 1)
out = comm.exchange_by_key(df, "key")
assert out.height == n and out.columns == df.columns and comm.rows_sent == 0 and comm.bytes_sent == 0     # nothing crosses the fabric at one rank
k2, v2, x2, c2 = (out[c].to_numpy() for c in ("key", "v", "x", "c8"))
o1, o2 = np.lexsort((x, v, key)), np.lexsort((x2, v2, k2))
assert np.array_equal(key[o1], k2[o2]) and np.array_equal(v[o1], v2[o2]) and np.array_equal(x[o1], x2[o2]) and np.array_equal(c8[o1], c2[o2])
stage("exchange checked")
res = pdist.sharded_groupby(comm, df, "key", lambda d: queries.cfg3(d.lazy()).collect(), always_exchange=True)
ref = queries.cfg3(df.lazy()).collect()
a, b = res.sort_host("key"), ref.sort_host("key")
assert a["key"] == b["key"] and a["v_sum"] == b["v_sum"] and a["v_count"] == b["v_count"]
same = comm.allgather(ref)
assert same.height == ref.height and np.array_equal(same["v_sum"].to_numpy(), ref["v_sum"].to_numpy())
try:
    comm.exchange_by_key(pl.DataFrame([pl.Series("key", key[:10]), pl.Series("b", np.arange(10) % 2 == 0)]), "key")     # bit-packed Boolean column
    raise SystemExit("a Boolean column was exchanged")
except pl.PlxError:
    pass
comm.close()
print("RCCL_WORKER_OK")
