"""Generic (interpreted) fused programs vs their AOT twins: the same queries with one extra aggregate so no
pre-instantiated shape matches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polars_amd as pl
from polars_amd import _ffi as F, queries, datagen
import bench
pl.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000

def run(name, lf, rows, bytes_per_row, steps=4):
    lf.collect(); plan = pl.last_plan()
    F.lib().plx_profile_clear(); F.lib().plx_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): lf.collect()
    F.lib().plx_synchronize(); dt = (time.perf_counter() - t0) / steps
    st = bench.kernel_stats(pl); F.lib().plx_profile_enable(0)
    k, v = max(st.items(), key=lambda kv: kv[1][1])
    print(f"{name:26s} {dt*1e3:8.3f} ms/step   {k}: {v[1]/v[0]:.1f} us = {rows*bytes_per_row/(v[1]/v[0]*1e-6)/1e12:.2f} TB/s   [{'aot' if 'aot' in plan else 'generic'}]")

cols = datagen.lineitem_device(n, seed=3); torch.cuda.synchronize()
df = datagen.frame_from_torch(pl, cols, datagen.LINEITEM_Q1_COLS)
c = pl.col
run("q1 (aot)", queries.q1(df.lazy()), n, 42)
disc_price = c("l_extendedprice") * (1 - c("l_discount"))
q1g = (df.lazy().filter(c("l_shipdate") <= queries.Q1_CUTOFF).group_by("l_returnflag", "l_linestatus")
       .agg(c("l_quantity").sum().alias("sum_qty"), c("l_extendedprice").sum().alias("sum_base_price"), disc_price.sum().alias("sum_disc_price"),
            (disc_price * (1 + c("l_tax"))).sum().alias("sum_charge"), c("l_quantity").mean().alias("avg_qty"), c("l_extendedprice").mean().alias("avg_price"),
            c("l_discount").mean().alias("avg_disc"), pl.len().alias("count_order"), c("l_tax").max().alias("max_tax")))
run("q1 + max(tax) (generic)", q1g, n, 42)
del df, cols; torch.cuda.empty_cache()
g = torch.Generator(device="cuda"); g.manual_seed(1)
a = torch.randint(0, 2**31, (n,), generator=g, device="cuda", dtype=torch.int64)
x = torch.rand((n,), generator=g, device="cuda", dtype=torch.float64) * 100
y = torch.rand((n,), generator=g, device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
df = pl.DataFrame([pl.Series.from_torch("a", a), pl.Series.from_torch("x", x), pl.Series.from_torch("y", y)])
run("cfg2 (aot)", queries.cfg2(df.lazy()), n, 24)
run("cfg2 + min(y) (generic)", df.lazy().filter(c("a") > 2**30).select((c("x") * (1 - c("y"))).sum(), c("x").mean(), c("a").sum(), c("y").min()), n, 24)
