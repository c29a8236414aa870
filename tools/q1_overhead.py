"""Whole-query wall time of Q1 SF100 split into: plx_execute_plan (C), frame wrapping (Python), download + to_dict."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import polars_amd as pl  # noqa: E402
from polars_amd import _ffi as F, datagen, queries  # noqa: E402
from polars_amd.frame import DataFrame  # noqa: E402

pl.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000_000
cols = datagen.lineitem_device(n, seed=10)
df = datagen.frame_from_torch(pl, cols, datagen.LINEITEM_Q1_COLS)
torch.cuda.synchronize()
for name, lf in (("q1", queries.q1(df.lazy())), ("q1_sorted", queries.q1_sorted(df.lazy()))):
    for _ in range(3):
        lf.collect().to_dict()
    K = 20
    acc = [0.0] * 4
    F.lib().plx_profile_clear(); F.lib().plx_profile_enable(1)
    for _ in range(K):
        F.lib().plx_synchronize()
        t0 = time.perf_counter()
        (ir, n_ir, ae, n_ae, keep), root, schema, _low = lf._lowered_c()
        out = C.c_uint64()
        t1 = time.perf_counter()
        F.check(F.lib().plx_execute_plan(ir, n_ir, ae, n_ae, root, 0, C.byref(out)))
        t2 = time.perf_counter()
        res = DataFrame._from_frame_handle(out.value, schema)
        t3 = time.perf_counter()
        d = res.to_dict()
        t4 = time.perf_counter()
        for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4))):
            acc[i] += b - a
    import bench
    st = bench.kernel_stats(pl)
    F.lib().plx_profile_enable(0)
    print(f"{name}: lowered_c {acc[0]/K*1e6:.1f} us | plx_execute_plan {acc[1]/K*1e6:.1f} us | wrap frame {acc[2]/K*1e6:.1f} us | to_dict {acc[3]/K*1e6:.1f} us | total {sum(acc)/K*1e6:.1f} us")
    for k, v in sorted(st.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:32s} x{v[0]/K:.1f}  {v[1]/v[0]:.1f} us")
