"""Fused join -> group-by with DUPLICATE build keys (engine.cpp fused_join_groupby, multi-value mode of the build table: chains of build rows per key, a group = a build
row, every probe row contributes to each row of its key's chain).  Reference: the build tables map a key to a list of rows (crates/polars-ops/src/frame/join/hash_join/
single_keys.rs:16-167) and the probe emits one pair per entry (single_keys_inner.rs:11-38); the group-by above sees one joined row per pair.  Ground truth here: pandas
merge -> groupby on the host, integer results bit-exact, float sums 1e-6 relative; every query also runs unfused (PLX_PLAN_NO_FUSION: pair list -> gather -> group-by)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pd = pytest.importorskip("pandas")

HASH_MULT = np.uint64(0x9E3779B97F4A7C15)


def _expected(pkey, pvalid, x, w, bkey, battr, battr_valid, by_attr):
    """inner join on k, then group by (k[, a]) with sum(x), sum(w), len: pandas on the host"""
    P = pd.DataFrame({"k": pkey, "x": x, "w": w})[pvalid]
    B = pd.DataFrame({"k": bkey, "a": pd.array(battr, dtype="Int64")})
    if battr_valid is not None:
        B.loc[~battr_valid, "a"] = pd.NA
    J = P.merge(B, on="k", how="inner")
    keys = ["k", "a"] if by_attr else ["k"]
    g = J.groupby(keys, dropna=False, sort=True).agg(sx=("x", "sum"), sw=("w", "sum"), n=("x", "size")).reset_index()
    return g


def _check(out, exp, by_attr):
    cols = ["k", "a"] if by_attr else ["k"]
    got = pd.DataFrame({c: pd.array(out[c].to_list(), dtype="Int64") for c in cols} | {"sx": out["sx"].to_numpy(), "sw": out["sw"].to_numpy(), "n": out["n"].to_numpy().astype(np.int64)})
    got = got.sort_values(cols, na_position="last").reset_index(drop=True)
    exp = exp.sort_values(cols, na_position="last").reset_index(drop=True)
    assert len(got) == len(exp), (len(got), len(exp))
    for c in cols:
        assert got[c].isna().tolist() == exp[c].isna().tolist() and got[c].dropna().astype(np.int64).tolist() == exp[c].dropna().astype(np.int64).tolist(), c
    assert got["sx"].astype(np.int64).tolist() == exp["sx"].astype(np.int64).tolist()
    assert got["n"].tolist() == exp["n"].astype(np.int64).tolist()
    assert np.allclose(got["sw"].to_numpy(), exp["sw"].to_numpy(), rtol=1e-6, atol=0)


def _frames(pl, rng, n, n_keys, hashed, null_attr=False, same_attr=False, dense=False):
    """build side: every key 1..8 times (dbgen partsupp has 4 rows per part); probe side: keys uniform over [0, 2 * n_keys) (half of them match nothing), 3 % null"""
    reps = rng.integers(1, 9, n_keys)
    ids = np.repeat(np.arange(n_keys, dtype=np.int64), reps)
    rng.shuffle(ids)
    nb = len(ids)
    # the attribute: a per-row value (so (k, a) is unique per build row) or -- same_attr -- one of TWO values per key: duplicate build rows that are ONE group
    battr = rng.integers(0, 2, nb).astype(np.int64) if same_attr else np.arange(nb, dtype=np.int64) * 7 + 3
    battr_valid = (rng.random(nb) > 0.2) if null_attr else None
    enc = (lambda a: (a.astype(np.uint64) * HASH_MULT).astype(np.int64)) if hashed else ((lambda a: a * 3 + 17) if not dense else (lambda a: a + 5))
    pid = rng.integers(0, 2 * n_keys, n).astype(np.int64)
    pvalid = rng.random(n) > 0.03
    x = rng.integers(-100, 100, n).astype(np.int64)
    w = rng.normal(size=n)
    bkey, pkey = enc(ids), enc(pid)
    B = pl.DataFrame([pl.Series("k", bkey), pl.Series("a", battr, validity=battr_valid) if battr_valid is not None else pl.Series("a", battr)])
    P = pl.DataFrame([pl.Series("k", pkey, validity=pvalid), pl.Series("x", x), pl.Series("w", w)])
    return B, P, (pkey, pvalid, x, w, bkey, battr, battr_valid)


def _query(pl, P, B, by_attr, **kw):
    c = pl.col
    keys = ("k", "a") if by_attr else ("k",)
    return P.lazy().join(B.lazy(), on="k").group_by(*keys).agg(c("x").sum().alias("sx"), c("w").sum().alias("sw"), pl.len().alias("n")).collect(**kw)


@pytest.mark.parametrize("hashed,by_attr", [(True, True), (False, True), (True, False)])
def test_duplicate_build_keys_take_the_fused_multi_value_path(pl, hashed, by_attr):
    rng = np.random.default_rng(41 + 2 * hashed + by_attr)
    n, n_keys = (1 << 22) + 4321, 150_000
    B, P, host = _frames(pl, rng, n, n_keys, hashed)
    out = _query(pl, P, B, by_attr)
    plan = pl.last_plan()
    assert "FusedJoinGroupBy{" in plan and "multi-value (row chains" in plan, plan
    _check(out, _expected(*host, by_attr), by_attr)
    ref = _query(pl, P, B, by_attr, no_fusion=True)
    assert "FusedJoinGroupBy" not in pl.last_plan()
    _check(ref, _expected(*host, by_attr), by_attr)


def test_duplicate_keys_found_by_the_direct_address_build_fall_through_to_the_chains(pl):
    """A dense build key range takes the bitmap build first; two pairs on one bit = duplicate keys -> the hash-table pipeline in multi-value mode (no second detection pass)."""
    rng = np.random.default_rng(5)
    B, P, host = _frames(pl, rng, (1 << 22) + 99, 120_000, hashed=False, dense=True)
    out = _query(pl, P, B, True)
    plan = pl.last_plan()
    assert "multi-value (row chains" in plan and "direct-address" not in plan, plan
    _check(out, _expected(*host, True), True)


def test_duplicate_rows_that_are_one_group_share_their_cells(pl):
    """Build rows of one key that agree on the build-side group column are ONE group of the reference's group-by (each probe row counted once per build row): the chain's
    representative rows (canonicalise_chains) collect them -- with and without nulls in that column (null == null for grouping)."""
    for seed, null_attr in ((11, False), (12, True)):
        rng = np.random.default_rng(seed)
        B, P, host = _frames(pl, rng, (1 << 22) + 7, 90_000, hashed=True, null_attr=null_attr, same_attr=True)
        out = _query(pl, P, B, True)
        assert "multi-value (row chains" in pl.last_plan(), pl.last_plan()
        _check(out, _expected(*host, True), True)


def test_multi_value_path_through_the_partitioned_probe_and_edge_keys(pl, monkeypatch):
    """The partitioned hash probe in front of the chains (forced), build keys that include the EMPTY pattern (-1: all ones) and 0, each several times."""
    rng = np.random.default_rng(77)
    n, n_keys = (1 << 22) + 1001, 60_000
    B, P, host = _frames(pl, rng, n, n_keys, hashed=True)
    pkey, pvalid, x, w, bkey, battr, bvalid = host
    special = np.array([-1, -1, -1, 0, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).min], dtype=np.int64)
    bkey = np.concatenate([bkey, special]); battr = np.concatenate([battr, np.arange(len(special), dtype=np.int64) + 10 ** 12])
    pkey = pkey.copy(); pkey[:3000] = rng.choice(special, 3000)
    B = pl.DataFrame({"k": bkey, "a": battr})
    P = pl.DataFrame([pl.Series("k", pkey, validity=pvalid), pl.Series("x", x), pl.Series("w", w)])
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    out = _query(pl, P, B, True)
    plan = pl.last_plan()
    assert "multi-value (row chains" in plan and "partitioned_hash_probe(" in plan, plan
    _check(out, _expected(pkey, pvalid, x, w, bkey, battr, None, True), True)


def test_float_build_group_column_with_duplicate_keys_runs_unfused(pl):
    """A float build-side group column cannot be compared bitwise (-0.0 == 0.0, NaN == NaN for grouping): duplicate keys then leave the fused path, same answer."""
    rng = np.random.default_rng(3)
    nb, n = 50_000, 400_000
    bkey = rng.integers(0, 20_000, nb).astype(np.int64)
    a = rng.choice(np.array([0.0, 1.5, 2.5]), nb)
    pkey = rng.integers(0, 40_000, n).astype(np.int64)
    x = rng.integers(0, 10, n).astype(np.int64)
    B = pl.DataFrame({"k": bkey, "a": a}); P = pl.DataFrame({"k": pkey, "x": x})
    out = P.lazy().join(B.lazy(), on="k").group_by("k", "a").agg(pl.col("x").sum().alias("sx"), pl.len().alias("n")).collect()
    assert "FusedJoinGroupBy" not in pl.last_plan()
    J = pd.DataFrame({"k": pkey, "x": x}).merge(pd.DataFrame({"k": bkey, "a": a}), on="k")
    g = J.groupby(["k", "a"]).agg(sx=("x", "sum"), n=("x", "size")).reset_index().sort_values(["k", "a"])
    got = pd.DataFrame({"k": out["k"].to_numpy(), "a": out["a"].to_numpy(), "sx": out["sx"].to_numpy(), "n": out["n"].to_numpy().astype(np.int64)}).sort_values(["k", "a"])
    assert got["k"].tolist() == g["k"].tolist() and got["a"].tolist() == g["a"].tolist() and got["sx"].tolist() == g["sx"].tolist() and got["n"].tolist() == g["n"].tolist()


# ---- LEFT joins on the fused path (single_keys_left.rs:106-195): matched rows as the inner join, unmatched rows grouped by their own key with nulls in the build columns ----
def _expected_left(pkey, pvalid, x, w, bkey, battr, battr_valid, by_attr):
    inner = _expected(pkey, pvalid, x, w, bkey, battr, battr_valid, by_attr)
    P = pd.DataFrame({"k": pd.array(pkey, dtype="Int64"), "x": x, "w": w})
    P.loc[~pvalid, "k"] = pd.NA
    un = P[~(pvalid & np.isin(pkey, bkey))]                       # no build row with this key (a null key matches nothing)
    g = un.groupby("k", dropna=False, sort=True).agg(sx=("x", "sum"), sw=("w", "sum"), n=("x", "size")).reset_index()
    if by_attr:
        g.insert(1, "a", pd.array([pd.NA] * len(g), dtype="Int64"))
        inner["a"] = inner["a"].astype("Int64")
    inner["k"] = inner["k"].astype("Int64")
    if not by_attr:
        # grouped by the key alone, a key's matched and unmatched rows cannot both exist: the two parts never share a group
        pass
    return pd.concat([inner, g], ignore_index=True)


def _query_left(pl, P, B, by_attr, **kw):
    c = pl.col
    keys = ("k", "a") if by_attr else ("k",)
    return P.lazy().join(B.lazy(), on="k", how="left").group_by(*keys).agg(c("x").sum().alias("sx"), c("w").sum().alias("sw"), pl.len().alias("n")).collect(**kw)


@pytest.mark.parametrize("dups,by_attr", [(False, True), (True, True), (True, False)])
def test_left_join_group_by_takes_the_fused_path(pl, dups, by_attr):
    """LEFT JOIN -> group_by on the fused pipeline: unique and duplicate build keys, null probe keys (their own group, null key), probe keys beyond the build range."""
    rng = np.random.default_rng(301 + 2 * dups + by_attr)
    n, n_keys = (1 << 22) + 999, 120_000
    B, P, host = _frames(pl, rng, n, n_keys, hashed=False, dense=True)
    pkey, pvalid, x, w, bkey, battr, bvalid = host
    if not dups:      # one row per key
        bkey, first = np.unique(bkey, return_index=True)
        battr = battr[first]
        B = pl.DataFrame({"k": bkey, "a": battr})
    out = _query_left(pl, P, B, by_attr)
    plan = pl.last_plan()
    assert "FusedJoinGroupBy{" in plan and "LeftJoinUnmatched{" in plan and ("multi-value" in plan) == dups, plan
    exp = _expected_left(pkey, pvalid, x, w, bkey, battr, None, by_attr)
    _check(out, exp, by_attr)
    ref = _query_left(pl, P, B, by_attr, no_fusion=True)
    assert "FusedJoinGroupBy" not in pl.last_plan()
    _check(ref, exp, by_attr)


def test_left_join_with_predicates_on_both_sides_and_every_row_unmatched(pl):
    """Predicates below a left join: the left one filters rows, the right one only decides which rows MATCH (a left row whose partners all fail it is unmatched, not dropped);
    and a build side that keeps no row at all: every left row is unmatched."""
    rng = np.random.default_rng(77)
    n, nb = (1 << 22) + 5, 200_000
    bkey = rng.permutation(nb).astype(np.int64)
    battr = rng.integers(0, 1000, nb).astype(np.int64)
    pkey = rng.integers(0, nb + 50_000, n).astype(np.int64)
    x = rng.integers(-9, 9, n).astype(np.int64)
    w = rng.random(n)
    B = pl.DataFrame({"k": bkey, "a": battr}); P = pl.DataFrame({"k": pkey, "x": x, "w": w})
    c = pl.col
    for cut in (500, -1):
        q = (P.lazy().filter(c("x") > -5).join(B.lazy().filter(c("a") < cut), on="k", how="left").group_by("k", "a")
             .agg(c("x").sum().alias("sx"), c("w").sum().alias("sw"), pl.len().alias("n")))
        out = q.collect()
        assert "LeftJoinUnmatched{" in pl.last_plan(), pl.last_plan()
        keep = x > -5
        bk = battr < cut
        exp = _expected_left(pkey[keep], np.ones(int(keep.sum()), bool), x[keep], w[keep], bkey[bk], battr[bk], None, True)
        _check(out, exp, True)
