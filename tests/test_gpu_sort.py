"""Sort / top-k / slice / semi+anti joins on the HIP path against the oracle's restatement of
arg_sort_multiple (oracle/pyoracle.py sort_indices, pinned by tests/test_oracle_golden.py).  The sort is stable, so
the index vector itself is compared (bit-exact), not just the sorted values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NP2PL = {np.int8: "Int8", np.int16: "Int16", np.int32: "Int32", np.int64: "Int64", np.uint8: "UInt8", np.uint16: "UInt16", np.uint32: "UInt32",
         np.uint64: "UInt64", np.float32: "Float32", np.float64: "Float64", np.bool_: "Boolean"}


def _rand_col(rng, n, t, card):
    if t is np.bool_:
        return rng.integers(0, 2, n).astype(bool)
    if t in (np.float32, np.float64):
        v = rng.integers(-card, card, n).astype(t) / 4
        special = rng.random(n)
        v = np.where(special < 0.02, np.nan, v)
        v = np.where((special > 0.02) & (special < 0.03), np.inf, v)
        v = np.where((special > 0.03) & (special < 0.04), -np.inf, v)
        v = np.where((special > 0.04) & (special < 0.06), -0.0, v)
        return v.astype(t)
    ii = np.iinfo(t)
    if card is None:   # full range incl. the extremes
        v = rng.integers(ii.min, ii.max, n, dtype=t, endpoint=True)
        if n > 2:
            v[0], v[1] = ii.min, ii.max
        return v
    lo = max(ii.min, -card) if ii.min < 0 else 0
    return rng.integers(lo, min(ii.max, card), n).astype(t)


def _series(pl, name, v, m):
    return pl.Series(name, v, dtype=getattr(pl, NP2PL[v.dtype.type]), validity=m)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 257, 4096, 4097, 100_003])
def test_arg_sort_fuzz_matches_oracle(pl, orc, n):
    rng = np.random.default_rng(1000 + n)
    types = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64, np.bool_]
    for trial in range(14 if n < 5000 else 6):
        nk = int(rng.integers(1, 4))
        keys, sers = [], []
        for j in range(nk):
            t = types[int(rng.integers(0, len(types)))]
            card = [3, 50, 100_000, None][int(rng.integers(0, 4))]
            v = _rand_col(rng, n, t, card if t not in (np.float32, np.float64) or card else 1000)
            m = None if rng.random() < 0.5 else rng.random(n) < [0.9, 0.5, 0.0][int(rng.integers(0, 3))]
            d, nl = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            keys.append((v, m, d, nl))
            sers.append(_series(pl, f"k{j}", v, m))
        exp = orc.sort_indices(keys)
        got = pl.arg_sort_by(sers, [k[2] for k in keys], [k[3] for k in keys]).to_numpy()
        assert got.dtype == np.uint32 and np.array_equal(got, exp), (n, trial, [(k[0].dtype, k[2], k[3], k[1] is not None) for k in keys], pl.last_plan())


@pytest.mark.parametrize("first", ["f64_distinct", "i64_low_card", "i32_nulls_first", "f64_nulls_last", "constant"])
@pytest.mark.parametrize("limit", [1, 10, 5000])
def test_top_k_selection_equals_sort_head(pl, orc, first, limit):
    rng = np.random.default_rng(77)
    n = 400_000
    m0 = None
    if first == "f64_distinct":
        k0 = rng.uniform(-1e6, 1e6, n)
    elif first == "i64_low_card":
        k0 = rng.integers(-20, 20, n).astype(np.int64)
    elif first == "i32_nulls_first":
        k0 = rng.integers(-10_000, 10_000, n).astype(np.int32); m0 = rng.random(n) < 0.999
    elif first == "f64_nulls_last":
        k0 = rng.normal(size=n); m0 = rng.random(n) < 0.5
    else:
        k0 = np.full(n, 7, dtype=np.int64)
    k1 = rng.integers(0, 1000, n).astype(np.int32)
    for desc in (False, True):
        nl = first != "i32_nulls_first"
        keys = [(k0, m0, desc, nl), (k1, None, not desc, False)]
        exp = orc.sort_indices(keys, limit)
        got = pl.arg_sort_by([_series(pl, "a", k0, m0), _series(pl, "b", k1, None)], [desc, not desc], [nl, False], limit=limit).to_numpy()
        assert np.array_equal(got, exp), (first, limit, desc, pl.last_plan())
        if first in ("f64_distinct", "i32_nulls_first") or (first == "f64_nulls_last" and limit < 5000):
            assert "top_k_select" in pl.last_plan(), pl.last_plan()    # selection actually ran (not a full sort)


def test_lazy_sort_slice_head_tail(pl, orc):
    rng = np.random.default_rng(3)
    n = 50_000
    a = rng.integers(0, 100, n).astype(np.int64)
    b = rng.uniform(size=n)
    bm = rng.random(n) < 0.8
    df = pl.DataFrame([pl.Series("a", a), pl.Series("b", b, validity=bm), pl.Series("row", np.arange(n, dtype=np.uint32))])
    order = orc.sort_indices([(a, None, True, False), (b, bm, False, True)])
    out = df.lazy().sort("a", "b", descending=[True, False], nulls_last=[False, True]).collect()
    assert np.array_equal(out["row"].to_numpy(), order) and out.columns == ["a", "b", "row"]
    assert out["b"].null_count() == int((~bm).sum())
    for off, ln in [(0, 10), (5, 20), (n - 3, 10), (-7, 3), (-7, 100), (-(n + 5), 10), (n + 1, 4), (100, None), (0, 0)]:
        got = df.lazy().sort("a", "b", descending=[True, False], nulls_last=[False, True]).slice(off, ln).collect()["row"].to_numpy()
        start = max(off + n, 0) if off < 0 else min(off, n)
        stop = n if ln is None else min(max((off + n if off < 0 else off) + ln, 0), n)
        assert np.array_equal(got, order[start:max(stop, start)]), (off, ln)
    assert np.array_equal(df.lazy().sort("b", nulls_last=True).tail(4).collect()["row"].to_numpy(), orc.sort_indices([(b, bm, False, True)])[-4:])
    # slice without a sort below it
    assert np.array_equal(df.slice(10, 5)["row"].to_numpy(), np.arange(10, 15))
    assert np.array_equal(df.tail(3)["a"].to_numpy(), a[-3:])
    # expression keys
    e = df.lazy().sort((pl.col("a") % 7), "row", descending=[False, True]).head(5).collect()["row"].to_numpy()
    assert np.array_equal(e, orc.sort_indices([(a % 7, None, False, False), (np.arange(n), None, True, False)], 5))
    with pytest.raises(ValueError, match=r"the length of `descending` \(1\) does not match the length of `by` \(2\)"):
        df.lazy().sort("a", "b", descending=[True])


def test_series_sort_and_top_k(pl):
    s = pl.Series("a", [3, 8, None, 1, 5, 2], dtype=pl.Int64)
    assert s.sort().to_list() == [None, 1, 2, 3, 5, 8]
    assert s.sort(descending=True, nulls_last=True).to_list() == [8, 5, 3, 2, 1, None]
    assert sorted(s.top_k(3).to_list()) == [3, 5, 8] and sorted(s.bottom_k(2).to_list()) == [1, 2]
    assert pl.Series("e", [], dtype=pl.Float64).sort().to_list() == []


def test_large_sort_properties(pl):
    """Size-independent properties at 2e7 rows (no oracle): the output is a permutation, keys come out non-decreasing and
    ties keep input order.  Host side is numpy only (torch kernels load very slowly on a cold GPU box)."""
    n = 20_000_000
    rng = np.random.default_rng(5)
    k = rng.integers(-2 ** 62, 2 ** 62, n, dtype=np.int64)
    s = pl.Series("k", k)
    idxs = s.arg_sort()
    assert "passes=8" in pl.last_plan(), pl.last_plan()
    idx = idxs.to_numpy()
    sk = s.gather(idxs).to_numpy()
    assert np.array_equal(sk, k[idx])                         # gather agrees with the host
    assert bool((sk[1:] >= sk[:-1]).all())
    seen = np.zeros(n, dtype=bool); seen[idx] = True
    assert bool(seen.all())                                   # permutation
    # narrow key range: most digit passes are skipped; descending; many ties
    k2 = (k & 1023).astype(np.int32)
    s2 = pl.Series("k2", k2)
    idx2 = s2.arg_sort(descending=True).to_numpy().astype(np.int64)
    assert "passes=2" in pl.last_plan(), pl.last_plan()
    sk2 = k2[idx2]
    assert bool((sk2[1:] <= sk2[:-1]).all())
    assert bool(((idx2[1:] > idx2[:-1]) | (sk2[1:] != sk2[:-1])).all())    # stable: ties in input order
    seen[:] = False; seen[idx2] = True
    assert bool(seen.all())


@pytest.mark.parametrize("how", ["semi", "anti"])
def test_semi_anti_join_large(pl, orc, how):
    rng = np.random.default_rng(8)
    nl, nr = 300_000, 50_000
    lk = rng.integers(0, 200_000, nl).astype(np.int64); lm = rng.random(nl) < 0.95
    rk = rng.integers(0, 200_000, nr).astype(np.int64); rm = rng.random(nr) < 0.9
    L = pl.DataFrame([pl.Series("k", lk, validity=lm), pl.Series("row", np.arange(nl, dtype=np.uint32))])
    R = pl.DataFrame([pl.Series("k", rk, validity=rm), pl.Series("z", np.arange(nr, dtype=np.int64))])
    out = L.join(R, on="k", how=how)
    exp = orc.semi_anti_join(orc.JOIN_SEMI if how == "semi" else orc.JOIN_ANTI, lk, lm, rk, rm)
    assert out.columns == ["k", "row"] and np.array_equal(out["row"].to_numpy(), exp)
    # empty right side
    E = pl.DataFrame([pl.Series("k", np.zeros(0, dtype=np.int64)), pl.Series("z", np.zeros(0, dtype=np.int64))])
    assert L.join(E, on="k", how=how).height == (0 if how == "semi" else nl)
    # filter -> semi join -> group_by composes through the per-node path
    agg = L.lazy().join(R.lazy(), on="k", how=how).select(pl.col("row").cast(pl.Int64).sum().alias("s"), pl.len().alias("n")).collect().to_dict()
    assert agg["s"][0] == int(exp.astype(np.int64).sum()) and agg["n"][0] == len(exp)


@pytest.mark.parametrize("n_orders", [2000, 150_000])
def test_q1_sorted_and_q3_top10(pl, orc, n_orders):
    from polars_amd import datagen, queries
    cols = datagen.lineitem_host(n_orders * 4, seed=5)
    df = datagen.to_frame(pl, cols, datagen.LINEITEM_Q1_COLS)
    exp = orc.q1(cols, datagen.us(1998, 9, 2))
    out = queries.q1_sorted(df.lazy()).collect()
    order = np.lexsort((exp["l_linestatus"], exp["l_returnflag"]))
    assert out["l_returnflag"].to_numpy().tolist() == exp["l_returnflag"][order].tolist()
    assert out["l_linestatus"].to_numpy().tolist() == exp["l_linestatus"][order].tolist()
    assert np.allclose(out["sum_charge"].to_numpy(), exp["sum_charge"][order], rtol=1e-9)
    orders, li = datagen.orders_lineitem_host(n_orders, seed=22)
    L = datagen.to_frame(pl, li, datagen.LINEITEM_Q3_COLS)
    O = datagen.to_frame(pl, orders, datagen.ORDERS_Q3_COLS)
    e3 = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, datagen.us(1995, 3, 15))
    top = queries.q3_top10(L.lazy(), O.lazy()).collect()
    assert "FusedJoinGroupBy" in pl.last_plan() and "_sort[" in pl.last_plan(), pl.last_plan()
    # the group order out of the join is unspecified, so ties on (revenue, o_orderdate) may resolve differently: compare keys
    o3 = np.lexsort((e3["o_orderdate"], -e3["revenue"]))[:10]
    assert top.height == min(10, len(e3["revenue"]))
    assert np.allclose(top["revenue"].to_numpy(), e3["revenue"][o3], rtol=1e-9)
    assert top["o_orderdate"].to_numpy().tolist() == e3["o_orderdate"][o3].tolist()
    if len(np.unique(e3["revenue"][o3])) == len(o3):
        assert top["l_orderkey"].to_numpy().tolist() == e3["l_orderkey"][o3].tolist()


@pytest.mark.parametrize("how", ["inner", "left", "semi", "anti"])
def test_multi_key_join_matches_row_encoded_oracle(pl, orc, how):
    """Two / three key columns (ints of different widths, a boolean, nulls): the engine packs them into one Int64,
    the oracle row-encodes them (oracle/pyoracle.py encode_key_rows); same pairs."""
    rng = np.random.default_rng(21)
    nl, nr = 40_000, 9_000
    la, ra = rng.integers(-50, 50, nl).astype(np.int32), rng.integers(-60, 40, nr).astype(np.int32)
    lb, rb = rng.integers(0, 300, nl).astype(np.int64) * 1_000_003, rng.integers(0, 300, nr).astype(np.int64) * 1_000_003
    lc, rc = rng.integers(0, 2, nl).astype(bool), rng.integers(0, 2, nr).astype(bool)
    lam, rbm = rng.random(nl) < 0.97, rng.random(nr) < 0.9
    L = pl.DataFrame([pl.Series("a", la, validity=lam), pl.Series("b", lb), pl.Series("c", lc), pl.Series("lrow", np.arange(nl, dtype=np.int64))])
    R = pl.DataFrame([pl.Series("a", ra), pl.Series("b", rb, validity=rbm), pl.Series("c", rc), pl.Series("rrow", np.arange(nr, dtype=np.int64))])
    lk, lv, rk, rv = orc.encode_key_rows([(la, lam), (lb, None), (lc.astype(np.uint8), None)], [(ra, None), (rb, rbm), (rc.astype(np.uint8), None)])
    out = L.join(R, on=["a", "b", "c"], how=how)
    assert "packed 3 key columns" in pl.last_plan(), pl.last_plan()
    if how in ("semi", "anti"):
        exp = orc.semi_anti_join(orc.JOIN_SEMI if how == "semi" else orc.JOIN_ANTI, lk, lv, rk, rv)
        assert out.columns == ["a", "b", "c", "lrow"] and np.array_equal(out["lrow"].to_numpy(), exp)
        return
    li, ri, rvalid = orc.join(orc.JOIN_LEFT if how == "left" else orc.JOIN_INNER, lk, lv, rk, rv)
    assert out.columns == ["a", "b", "c", "lrow", "rrow"]          # all three right key columns are coalesced away
    d = out.to_dict()
    got = sorted(zip(d["lrow"], [(-1 if x is None else x) for x in d["rrow"]]))
    want = sorted(zip(li.tolist(), [int(r) if (rvalid is None or rvalid[i]) else -1 for i, r in enumerate(ri.tolist())]))
    assert got == want
    assert out.height > 1000
