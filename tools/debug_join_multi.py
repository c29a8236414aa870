"""debug driver (GPU): the duplicate-key fused join at growing sizes, every stage announced on stderr"""
import faulthandler, sys, time, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import polars_amd as pl
pl.init(0)
import test_gpu_join_duplicate_keys as T

def say(*a):
    print(f"[{time.strftime('%X')}]", *a, file=sys.stderr, flush=True)

for n, n_keys, hashed, by_attr in ((100_000, 5_000, True, True), (1_000_000, 50_000, True, True), ((1 << 22) + 4321, 150_000, True, True), ((1 << 22) + 4321, 150_000, False, True), ((1 << 22) + 4321, 150_000, True, False)):
    rng = np.random.default_rng(1)
    B, P, host = T._frames(pl, rng, n, n_keys, hashed)
    say("frames ready", n, n_keys, hashed, by_attr)
    faulthandler.dump_traceback_later(60, exit=True)
    t0 = time.time()
    out = T._query(pl, P, B, by_attr)
    say("fused done", round(time.time() - t0, 2), "s;", pl.last_plan()[:300])
    T._check(out, T._expected(*host, by_attr), by_attr)
    say("checked")
    ref = T._query(pl, P, B, by_attr, no_fusion=True)
    say("unfused done", pl.last_plan()[:200])
    T._check(ref, T._expected(*host, by_attr), by_attr)
    faulthandler.cancel_dump_traceback_later()
    say("ok")
