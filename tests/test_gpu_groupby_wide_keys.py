"""Multi-column group keys that do not bit-pack ("wide" keys: the reference row-encodes them -- crates/polars-row/src/encode.rs, crates/polars-expr/src/hash_keys.rs:334
RowEncodedKeys, groups/row_encoded.rs) on the PARTITIONED path: rows scattered by the hash of the key words + null mask, per-partition LDS tables that compare word
by word (partition2_device.hpp: process_wide).  Checked against numpy / pandas on the same rows and against the HBM-table path of the same library."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def close(a, b):
    return np.allclose(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64), rtol=1e-9, atol=1e-9)


def _frame_rows(df, cols):
    d = {c: df[c]._download() for c in cols}
    out = []
    for c in cols:
        vals, valid = d[c]
        valid = np.ones(len(vals), bool) if valid is None else np.asarray(valid, bool)
        out.append(np.where(valid, np.asarray(vals).astype(np.float64), np.nan))
    return np.stack(out, axis=1)


def test_two_int64_keys_take_the_partitioned_path(pl):
    """2 x Int64 keys over the full 64-bit range (nothing to pack), 1.7e7 rows, ~3e5 groups: sum / count / min / max / mean / len."""
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(201)
    n, G = 17_000_003, 300_000
    g = rng.integers(0, G, n)
    ka = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    kb = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    x = rng.uniform(-1, 1, n)
    df = pl.DataFrame({"a": ka[g], "b": kb[g], "v": v, "x": x})
    c = pl.col
    q = df.lazy().group_by("a", "b").agg(c("v").sum().alias("s"), c("v").count().alias("n"), c("v").min().alias("mn"), c("x").max().alias("mx"), c("x").mean().alias("m"), pl.len().alias("len"))
    out = q.collect()
    plan = pl.last_plan()
    assert "partitioned(v3,hash" in plan and "lds_wide_key_table(words=2" in plan, plan
    ref = q.collect(no_partition=True)
    assert "wide_hash_hbm_table" in pl.last_plan(), pl.last_plan()
    assert out.height == ref.height == len(np.unique(g))
    a, b = out.to_dict(), ref.to_dict()
    oa, ob = np.lexsort((a["b"], a["a"])), np.lexsort((b["b"], b["a"]))
    for col in ("a", "b", "s", "n", "mn", "len"):
        assert np.array_equal(np.asarray(a[col])[oa], np.asarray(b[col])[ob]), col
    for col in ("mx", "m"):
        assert close(np.asarray(a[col])[oa], np.asarray(b[col])[ob]), col
    want = pd.DataFrame({"a": ka[g], "b": kb[g], "v": v, "x": x}).groupby(["a", "b"]).agg(s=("v", "sum"), n=("v", "count"), mn=("v", "min"), mx=("x", "max"), m=("x", "mean"), len=("v", "size")).reset_index()
    want = want.sort_values(["a", "b"]).reset_index(drop=True)
    for col in ("a", "b", "s", "n", "mn", "len"):
        assert np.array_equal(np.asarray(a[col])[oa], want[col].to_numpy()), col
    assert close(np.asarray(a["mx"])[oa], want["mx"].to_numpy()) and close(np.asarray(a["m"])[oa], want["m"].to_numpy())


def test_wide_keys_with_null_key_columns_and_a_float_key(pl):
    """Three key columns -- Int64 with nulls, Float64 (-0.0 == +0.0, all NaNs one key: total_ord.rs:40-48), UInt64 beyond 2^63 -- and a nullable value: a null in a key
    column is a key value of its own, (null, x) and (y, null) are different groups; the partitioned path agrees with the HBM-table path row for row."""
    rng = np.random.default_rng(202)
    n, G = 17_000_003, 100_000
    g = rng.integers(0, G, n)
    ka = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    ka_valid = rng.random(G) > 0.03
    kf = rng.choice(np.array([0.0, -0.0, 1.5, np.nan, -np.inf, 2.0 ** 60]), G)
    ku = rng.integers(0, 1 << 63, G).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    v_valid = rng.random(n) > 0.1
    df = pl.DataFrame([pl.Series("a", ka[g], validity=ka_valid[g]), pl.Series("f", kf[g]), pl.Series("u", ku[g]), pl.Series("v", v, validity=v_valid)])
    c = pl.col
    q = df.lazy().group_by("a", "f", "u").agg(c("v").sum().alias("s"), c("v").count().alias("n"), pl.len().alias("len"))
    out = q.collect()
    plan = pl.last_plan()
    assert "partitioned(v3,hash" in plan and "lds_wide_key_table(words=4" in plan, plan          # three key words + the null mask
    ref = q.collect(no_partition=True)
    assert out.height == ref.height
    cols = ["a", "f", "u", "s", "n", "len"]
    ra, rb = _frame_rows(out, cols), _frame_rows(ref, cols)
    key = lambda r: np.lexsort(tuple(np.nan_to_num(r[:, i], nan=1e300, posinf=1e301, neginf=-1e301) for i in range(r.shape[1] - 1, -1, -1)))
    assert np.array_equal(ra[key(ra)], rb[key(rb)], equal_nan=True)
    # (the HBM-table path is pinned to the oracle by tests/test_gpu_kernels.py::test_groupby_multi_key_wide and the group_by KATs)
    assert int(np.isnan(ra[:, 0]).sum()) > 0 and int(out["len"].to_numpy().sum()) == n and int(out["n"].to_numpy().sum()) == int(v_valid.sum())


def test_wide_key_table_overflow_plans_more_partitions(pl, monkeypatch):
    """An estimate that is far too low (every block of the planner's strided sample -- 8 blocks of 2^17 rows -- sees the same few keys): the aggregation pass reports a
    full LDS table and the pass is repeated ONCE, at the largest plan."""
    rng = np.random.default_rng(203)
    n, G = 17_000_003, 900_000
    g = rng.integers(0, G, n)
    per, stride = ((1 << 20) // 8) & ~127, (n // 8) & ~127
    for b in range(8):
        g[b * stride: b * stride + per] = rng.integers(0, 5000, per)                           # what the planner samples: ~5000 groups
    ka = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    kb = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    v = rng.integers(0, 100, n).astype(np.int64)
    df = pl.DataFrame({"a": ka[g], "b": kb[g], "v": v})
    out = df.lazy().group_by("a", "b").agg(pl.col("v").sum().alias("s"), pl.len().alias("len")).collect()
    plan = pl.last_plan()
    assert plan.count("lds-overflow(P=") == 1 and "P=512" in plan and "lds_wide_key_table" in plan, plan
    assert out.height == len(np.unique(g)) and int(out["s"].to_numpy().sum()) == int(v.sum()) and int(out["len"].to_numpy().sum()) == n


def test_wide_keys_with_one_heavy_group(pl):
    """Half of the rows share ONE (a, b) pair (the wide-key path has no hot-key cells: they all meet in one partition, on one group's LDS cells), the rest spread over
    200 000 pairs: same groups and sums as numpy."""
    rng = np.random.default_rng(204)
    n, G = 17_000_000, 200_000
    g = rng.integers(0, G, n)
    g[rng.random(n) < 0.5] = 7
    ka = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    kb = rng.integers(-(1 << 62), 1 << 62, G).astype(np.int64)
    v = rng.integers(0, 1000, n).astype(np.int64)
    out = pl.DataFrame({"a": ka[g], "b": kb[g], "v": v}).lazy().group_by("a", "b").agg(pl.col("v").sum().alias("s"), pl.len().alias("len")).collect()
    assert "lds_wide_key_table" in pl.last_plan(), pl.last_plan()
    cnt = np.bincount(g, minlength=G); sums = np.bincount(g, weights=v, minlength=G).astype(np.int64)
    present = np.nonzero(cnt)[0]
    got = {(int(a), int(b)): (int(s), int(c)) for a, b, s, c in zip(out["a"].to_numpy(), out["b"].to_numpy(), out["s"].to_numpy(), out["len"].to_numpy())}
    assert len(got) == len(present)
    assert got == {(int(ka[i]), int(kb[i])): (int(sums[i]), int(cnt[i])) for i in present}
