import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polars_amd as pl
from polars_amd import dist as pdist
import bench
pl.init(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
wl = bench.make_workload(pl, "cfg3", n, seed=10)
df = wl.step()[1][0]
comm = pdist.LibComm(pl)
c = pl.col
def sums(d):
    r = d.lazy().select(c("key").sum().alias("sk"), c("v").sum().alias("sv"), pl.len().alias("n")).collect()
    return r["sk"].to_list()[0], r["sv"].to_list()[0], r["n"].to_list()[0]
print("before", sums(df), flush=True)
for rep in range(3):
    out = comm.exchange_by_key(df, "key")
    print("after ", sums(out), "sync" if not os.environ.get("PLX_COMM_NO_SYNC") else "nosync", flush=True)
    spec = pdist.GroupBySpec("key", [("v_sum", "v", "sum"), ("v_count", "v", "count")])
    g = pdist.LibFrameOps(pl).final(out, spec)
    print("  groups", g.height, "sum of sums", int(np.asarray(g["v_sum"].to_numpy()).astype(np.int64).sum()), "sum of counts", int(np.asarray(g["v_count"].to_numpy()).astype(np.int64).sum()), pl.last_plan()[:150], flush=True)
    del out, g
