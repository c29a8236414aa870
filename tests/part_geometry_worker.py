"""One partitioned group-by under the geometry the environment dictates (PLX_PART_LOG2_PARTS / PLX_PART_DIRECT_LOG2_PARTS / PLX_PART_TILES /
PLX_PART_PACK are read once per process: tests/test_gpu_partition_geometry.py starts this script once per geometry).  argv: mode (hash | direct), input
(hot: a stretch of rows from 48 keys, which the sample makes heavy hitters -- the scatter's hot-key build; flat: uniform keys; hot1 / flat1: the same with
ONE f64 value column -- what travels two rows a record, fused::kPackPair) and the substrings the plan description of the second run must contain.  Checks the result against numpy and prints the plan."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pl  # noqa: E402


def main():
    mode, shape, want = sys.argv[1], sys.argv[2], sys.argv[3:]
    pl.init(0)
    rng = np.random.default_rng(97)
    n, G = 17_300_001, 200_000
    ids = rng.integers(0, G, n)
    if shape.startswith("hot"):
        ids[n // 2: n // 2 + n // 50] = rng.integers(0, 48, n // 50)
    v = rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int64)
    x = rng.uniform(-1, 1, n)
    wide_v = shape.endswith("v")                                    # flatv / hotv: keys over all 64 bits, ONE Int64 value spanning 2^41 (fused::kPackPairV)
    if wide_v:
        v = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    if mode == "hash" and wide_v:
        key = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).astype(np.int64)
        df = pl.DataFrame({"k": key, "v": v, "x": x})
    elif mode == "hash":
        key = ids.astype(np.int64) * 1_000_003 - 10 ** 12           # sparse 64-bit keys: hash partitions
        df = pl.DataFrame({"k": key, "v": v, "x": x})
    else:
        key = ids.astype(np.uint32)                                 # dense ids: direct-address partitions
        df = pl.DataFrame([pl.Series("k", key, dtype=pl.Categorical([], pl.UInt32)), pl.Series("v", v), pl.Series("x", x)])
    one = shape.endswith("1")
    q = (df.lazy().group_by("k").agg(pl.col("v").sum().alias("s"), pl.col("v").count().alias("n")) if wide_v else
         df.lazy().group_by("k").agg(pl.col("x").sum().alias("xs"), pl.col("x").mean().alias("xm"), pl.len().alias("n")) if one else
         df.lazy().group_by("k").agg(pl.col("v").sum().alias("s"), pl.col("x").sum().alias("xs"), pl.len().alias("n")))
    for run in range(2):                                            # the second run knows the key range (packed / fused records)
        out = q.collect()
        plan = pl.last_plan()
        print(f"run {run}: {plan}")
        assert "partitioned(v3," in plan, plan
        if run == 1:                                                # (the first run of dense ids has no key range yet and takes hash partitions)
            for w in want:
                assert w in plan, (w, plan)
        k = out["k"].to_numpy()
        order = np.argsort(k)
        present = np.unique(ids)
        assert np.array_equal(k[order], np.unique(key)), "keys"
        if wide_v:                                                  # (sums up to 2^40 x rows per key: exact integer arithmetic on both sides)
            order = np.argsort(k)
            uk, inv = np.unique(key, return_inverse=True)
            wsum = np.zeros(len(uk), np.int64); np.add.at(wsum, inv, v)
            assert np.array_equal(k[order], uk) and np.array_equal(out["s"].to_numpy()[order], wsum) and np.array_equal(out["n"].to_numpy()[order], np.bincount(inv)), "wide values"
            continue
        assert np.array_equal(out["n"].to_numpy()[order], np.bincount(ids, minlength=G)[present]), "len"
        if one:
            cnt = np.bincount(ids, minlength=G)[present]
            assert np.allclose(out["xm"].to_numpy()[order], np.bincount(ids, x, minlength=G)[present] / cnt, rtol=1e-9, atol=1e-9), "mean"
        else:
            assert np.array_equal(out["s"].to_numpy()[order], np.bincount(ids, v, minlength=G)[present].astype(np.int64)), "int sum"       # |sums| < 2^53: exact in the float accumulator of bincount
        assert np.allclose(out["xs"].to_numpy()[order], np.bincount(ids, x, minlength=G)[present], rtol=1e-9, atol=1e-9), "float sum"
    print("OK")


if __name__ == "__main__":
    main()
