"""Measurement: the filter -> frame kernels (fused_sinks.hpp BallotSink, kernels_filter.hip compact_by_ballots) under their grid sizes (PLX_BPC_BALLOTS, PLX_BPC_COMPACT).
One subprocess per configuration (the run-time compiled kernel is cached per process and shape).  python tools/exp_filter.py [rows]"""
import itertools, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, ctypes as C
sys.path.insert(0, %r)
import numpy as np
import polars_amd as pl
pl.init(0)
F = pl._ffi
n = int(sys.argv[1])
def col(name, dt, npn, stream, lo, hi, scale=1.0):
    h = C.c_uint64()
    F.check(F.lib().plx_datagen_uniform(pl.datatypes.physical_code(dt) if hasattr(pl.datatypes, "physical_code") else dt.physical, n, 20, stream, lo, hi, C.c_double(scale), C.byref(h)))
    return pl.Series._from_handle(name, h.value, dt)
sys.path.insert(0, %r)
import bench
df = pl.DataFrame([bench.native_uniform_column(pl, "a", pl.Int64, "Int64", n, 20, 0, 0, 2 ** 31), bench.native_uniform_column(pl, "x", pl.Float64, "Float64", n, 20, 1, 0, 10 ** 9, 1e-7),
                   bench.native_uniform_column(pl, "y", pl.Float64, "Float64", n, 20, 2, 0, 10 ** 9, 1e-9)])
lf = df.lazy().filter(pl.col("a") > (1 << 30))
for _ in range(2): out = lf.collect()
F.check(F.lib().plx_synchronize())
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = lf.collect(); F.check(F.lib().plx_synchronize()); ts.append((time.perf_counter() - t0) * 1e3)
sys.path.insert(0, %r)
st = None
F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
out = lf.collect(); F.check(F.lib().plx_synchronize())
st = bench.kernel_stats(pl)
F.check(F.lib().plx_profile_enable(0))
print("RESULT", round(min(ts), 2), round(sorted(ts)[len(ts) // 2], 2), out.height, {k: round(v[1] / v[0], 1) for k, v in st.items()})
''' % (ROOT, ROOT, ROOT)

def main():
    rows = sys.argv[1] if len(sys.argv) > 1 else "1000000000"
    res = []
    import itertools
    for variant, bpc_c in itertools.product((44, 42, 43, 82, 83, 24), (8, 12, 16)):
        env = dict(os.environ, PLX_BPC_COMPACT=str(bpc_c), PLX_COMPACT_VARIANT=str(variant))
        p = subprocess.run([sys.executable, "-c", CHILD, rows], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        r = {"variant": variant, "bpc_compact": bpc_c, "out": line[0] if line else (p.stderr[-400:])}
        print(json.dumps(r), flush=True)
        res.append(r)
    return 0

if __name__ == "__main__":
    sys.exit(main())
