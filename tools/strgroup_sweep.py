#!/usr/bin/env python
"""The string-key group-by operator (plx_strview_groupby) against group count and skew: 2^26 rows of inline 12-byte strings "id%010d", one f64 value
column, group_by(k).agg(sum, mean); uniform keys over 1e2 / 1e3 / 1e4 / 1e6 distinct strings, zipf s = 1.1 and one string holding half of the rows over 1e6;
beside it the encode-then-group route on the same views.  usage (GPU box): python tools/strgroup_sweep.py [log2_rows] -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pl  # noqa: E402
import bench  # noqa: E402


def views_of_ids(ids: np.ndarray) -> np.ndarray:
    """The inline Utf8View words of "id%010d" % id: {len = 12 | bytes 0-3, bytes 4-11}."""
    digits = np.zeros((len(ids), 10), np.uint8)
    x = ids.astype(np.int64).copy()
    for j in range(9, -1, -1):
        digits[:, j] = 48 + x % 10
        x //= 10
    raw = np.zeros((len(ids), 16), np.uint8)
    raw[:, 0] = 12
    raw[:, 4] = ord("i"); raw[:, 5] = ord("d")
    raw[:, 6:16] = digits
    return raw.reshape(-1).view(np.uint64)


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    n = 1 << lg
    pl.init(0)
    F = pl._ffi
    rng = np.random.default_rng(11)
    v = pl.Series("v", rng.uniform(0, 100, n))
    cases = {"uniform_1e2": rng.integers(0, 100, n), "uniform_1e3": rng.integers(0, 1000, n), "uniform_1e4": rng.integers(0, 10_000, n), "uniform_1e6": rng.integers(0, 1_000_000, n),
             "zipf_1.1_1e6": (rng.zipf(1.1, n) - 1) % 1_000_000}
    k = rng.integers(0, 1_000_000, n); k[rng.random(n) < 0.5] = 777_777
    cases["one_string_50pct_1e6"] = k
    out = {"rows": n}
    for name, ids in cases.items():
        views = pl.Series("views", views_of_ids(ids), pl.UInt64)
        res = {}
        for route in ("views", "encode"):
            def step():
                key = pl.Series.from_device_views("k", views, encode="deferred" if route == "views" else "eager")
                return pl.DataFrame([key, v]).lazy().group_by("k").agg(pl.col("v").sum().alias("s"), pl.col("v").mean().alias("m")).collect()
            r = step(); step()
            F.check(F.lib().plx_synchronize())
            F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); r = step(); F.check(F.lib().plx_synchronize()); ts.append(time.perf_counter() - t0)
            ks = {kn: round(kv[1] / 5) for kn, kv in bench.kernel_stats(pl).items() if kv[1] / 5 >= 20}
            F.check(F.lib().plx_profile_enable(0))
            ts.sort()
            res[route] = {"ms_median": round(ts[2] * 1e3, 3), "groups": r.height, "kernel_us": ks, "operator": "StringViewGroupBy" in pl.last_plan()}
            if route == "views":
                got = dict(zip(r["k"].to_list(), r["s"].to_list()))
                want = np.bincount(ids, v.to_numpy())
                bad = [g for g in list(got)[:2000] if abs(got[g] - want[int(g[2:])]) > 1e-9 * max(1.0, abs(want[int(g[2:])]))]
                res[route]["spot_check_ok"] = not bad and len(got) == int((np.bincount(ids) > 0).sum())
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
