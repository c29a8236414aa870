import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polars_amd as pl
from polars_amd import dist as pdist
import bench
pl.init(0)
comm = pdist.LibComm(pl)
c = pl.col
def sums(d):
    r = d.lazy().select(c("key").sum().alias("sk"), c("v").sum().alias("sv"), pl.len().alias("n")).collect()
    return r["sk"].to_list()[0], r["sv"].to_list()[0], r["n"].to_list()[0]
for n in (50_000_000, 100_000_000, 134_000_000, 135_000_000, 150_000_000, 200_000_000, 300_000_000):
    wl = bench.make_workload(pl, "cfg3", n, seed=10)
    df = wl.step()[1][0]
    b = sums(df)
    out = comm.exchange_by_key(df, "key")
    a = sums(out)
    # how many rows of the exchanged frame are all-zero?
    z = out.lazy().filter((c("key") == 0) & (c("v") == 0)).select(pl.len().alias("z")).collect()["z"].to_list()[0]
    print(n, "bytes/col", n * 8, "ratio", round(a[0] / b[0], 5), round(a[1] / b[1], 5), "rows", a[2], "zero rows", z, flush=True)
    del df, out, wl
