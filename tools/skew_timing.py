#!/usr/bin/env python
"""Skew through the partitioned group-by: wall time of group_by(key).agg(sum, count) over 2^26 rows / ~1e6 keys
for uniform keys, zipf s = 1.1 and one key holding half of the rows; the same query, the same (hash-mode) plan family.
usage (GPU box): python tools/skew_timing.py [log2_rows] -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pl  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    n = 1 << lg
    pl.init(0)
    rng = np.random.default_rng(7)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    cases = {}
    cases["uniform"] = rng.integers(0, 1_000_000, n).astype(np.int64)
    cases["zipf_1.1"] = ((rng.zipf(1.1, n) - 1) % 1_000_000).astype(np.int64)
    k = rng.integers(0, 1_000_000, n).astype(np.int64); k[rng.random(n) < 0.5] = 777_777
    cases["one_hot_key_50pct"] = k
    out = {"rows": n}
    for name, key in cases.items():
        key = key * 1_000_003 - 5          # not a dense range: raw 64-bit keys, hash mode
        df = pl.DataFrame({"key": key, "v": v})
        q = df.lazy().group_by("key").agg(pl.col("v").sum().alias("s"), pl.col("v").count().alias("c"))
        q.collect(); q.collect()
        pl.synchronize() if hasattr(pl, "synchronize") else None
        ts = []
        F = pl._ffi
        F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
        for _ in range(7):
            t0 = time.perf_counter(); r = q.collect(); ts.append(time.perf_counter() - t0)
        F.check(F.lib().plx_synchronize())
        import bench
        ks = {k: round(v[1] / 7) for k, v in bench.kernel_stats(pl).items()}      # us per query
        F.check(F.lib().plx_profile_enable(0))
        ts.sort()
        out[name] = {"ms_median": round(ts[len(ts) // 2] * 1e3, 3), "ms_min": round(ts[0] * 1e3, 3), "groups": r.height, "kernel_us": ks, "plan": pl.last_plan()[:160]}
        del df, q, r
    base = out["uniform"]["ms_median"]
    for name in ("zipf_1.1", "one_hot_key_50pct"):
        out[name]["vs_uniform"] = round(out[name]["ms_median"] / base, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
