"""Query-level parity: the BASELINE.json configurations and TPC-H Q1 / Q3 through
plx_execute_plan, (a) fused pipelines, (b) PLX_PLAN_NO_FUSION (one kernel per IR node, the
reference's execution shape), (c) the CPU oracle -- the reference's own engine-parity pattern
(py-polars/tests/unit/streaming/test_streaming_group_by.py:149-196: same query, two executors,
frame-equal with check_row_order=False).  Integer results bit-exact, float aggregates 1e-6 rel."""
import math
import os

import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.allclose(a, b, rtol=RTOL, atol=0.0, equal_nan=True)


def test_cfg1_filter_sum(pl, orc):
    """BASELINE config 1 (1e7-row Int64, filter(a > k).sum()): exact."""
    from polars_amd import queries
    a = np.random.Generator(np.random.PCG64(0)).integers(0, 2**31, 10_000_000, dtype=np.int64)
    df = pl.DataFrame({"a": a})
    for nf in (False, True):
        out = queries.cfg1(df.lazy()).collect(no_fusion=nf)
        assert out.rows() == [(orc.q_filter_sum_cfg1(a, 2**30),)]
    assert out.rows()[0][0] == int(a[a > 2**30].sum())


@pytest.mark.parametrize("null_frac", [0.0, 0.05])
@pytest.mark.parametrize("n", [0, 1, 127, 128, 129, 100_000, 3_000_001])
def test_cfg2_filter_arith_agg(pl, orc, n, null_frac):
    from polars_amd import datagen, queries
    a, x, y, xv = datagen.cfg2_host(n, null_frac if n > 100 else 0.0)
    df = pl.DataFrame([pl.Series("a", a), pl.Series("x", x, validity=xv), pl.Series("y", y)])
    exp = orc.q_filter_agg_cfg2(a, x, y, 2**30, xv)
    for nf in (False, True):
        out = queries.cfg2(df.lazy()).collect(no_fusion=nf)
        plan = pl.last_plan()
        assert ("fused_scan[aot]" in plan) == (not nf), plan
        (xy, xm, asum), = out.rows()
        assert asum == exp["a_sum"]
        if exp["x_mean"] is None:
            assert xm is None and xy == 0.0
        else:
            assert math.isclose(xm, exp["x_mean"], rel_tol=RTOL) and math.isclose(xy, exp["xy"], rel_tol=RTOL)
        assert out.schema == {"xy": pl.Float64, "x_mean": pl.Float64, "a_sum": pl.Int64}


@pytest.mark.parametrize("zipf", [0.0, 1.1])
def test_cfg3_groupby_1e6_keys(pl, orc, zipf):
    """BASELINE config 3 at 4e6 rows / 1e6 keys (the oracle finishes in seconds)."""
    from polars_amd import datagen, queries
    key, v = datagen.cfg3_host(4_000_000, 1_000_000, zipf)
    df = pl.DataFrame({"key": key, "v": v})
    out = queries.cfg3(df.lazy()).collect()
    plan = pl.last_plan()
    assert "fused_scan[aot]" in plan, plan
    k = out["key"].to_numpy(); order = np.argsort(k)
    uk, inv = np.unique(key, return_inverse=True)
    assert np.array_equal(k[order], uk)
    assert np.array_equal(out["v_sum"].to_numpy()[order], np.bincount(inv, weights=None, minlength=len(uk)) * 0 + np.bincount(inv, v).astype(np.int64))
    assert np.array_equal(out["v_count"].to_numpy()[order], np.bincount(inv).astype(np.uint32))
    r = orc.q_groupby([key[:500_000]], [None], [("s", orc.AGG_SUM, v[:500_000], None)])
    small = queries.cfg3(pl.DataFrame({"key": key[:500_000], "v": v[:500_000]}).lazy()).collect(no_fusion=True)
    o1, o2 = np.argsort(r["key_0"][0]), np.argsort(small["key"].to_numpy())
    assert np.array_equal(r["s"][0][o1], small["v_sum"].to_numpy()[o2])
    assert out.schema == {"key": pl.Int64, "v_sum": pl.Int64, "v_count": pl.UInt32}


def test_cfg5_dictionary_string_keys(pl, orc):
    from polars_amd import datagen, queries
    codes, v = datagen.cfg5_host(2_000_000, 1_000_000)
    df = pl.DataFrame([pl.Series("k", codes, dtype=pl.Categorical([], pl.UInt32)), pl.Series("v", v)])
    out = queries.cfg5(df.lazy()).collect()
    k = out["k"].to_numpy(); order = np.argsort(k)
    uk, inv = np.unique(codes, return_inverse=True)
    assert np.array_equal(k[order], uk)
    s = np.bincount(inv, v); c = np.bincount(inv)
    assert close(out["v_sum"].to_numpy()[order], s) and close(out["v_mean"].to_numpy()[order], s / c)


@pytest.mark.parametrize("n", [0, 5, 1000, 200_000, 2_000_003])
def test_q1(pl, orc, n):
    from polars_amd import datagen, queries
    li = datagen.lineitem_host(n, seed=21)
    df = datagen.to_frame(pl, li, datagen.LINEITEM_Q1_COLS)
    exp = orc.q1({k: li[k] for k in datagen.LINEITEM_Q1_COLS}, datagen.us(1998, 9, 2))
    for nf in (False, True):
        out = queries.q1(df.lazy()).collect(no_fusion=nf)
        plan = pl.last_plan()
        if not nf and n:
            assert "fused_scan[aot]" in plan and "lds_table" in plan, plan
        g = out.sort_host(["l_returnflag", "l_linestatus"])
        assert [datagen.FLAGS.index(x) for x in g["l_returnflag"]] == exp["l_returnflag"].tolist(), plan
        assert [datagen.STATUS.index(x) for x in g["l_linestatus"]] == exp["l_linestatus"].tolist()
        assert g["sum_qty"] == exp["sum_qty"].tolist() and g["count_order"] == exp["count_order"].tolist()
        for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert close(g[c], exp[c]), (c, nf)
    assert out.schema["sum_qty"] == pl.Int64 and out.schema["count_order"] == pl.UInt32 and out.schema["avg_qty"] == pl.Float64


@pytest.mark.parametrize("ordered", [False, True])
@pytest.mark.parametrize("n_orders", [0, 10, 5000, 150_000])
def test_q3(pl, orc, n_orders, ordered):
    from polars_amd import datagen, queries
    orders, li = datagen.orders_lineitem_host(n_orders, seed=22, ordered=ordered)
    L = datagen.to_frame(pl, li, datagen.LINEITEM_Q3_COLS)
    O = datagen.to_frame(pl, orders, datagen.ORDERS_Q3_COLS)
    exp = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, datagen.us(1995, 3, 15))
    for nf in (False, True):
        out = queries.q3(L.lazy(), O.lazy()).collect(no_fusion=nf)
        if n_orders:
            assert ("FusedJoinGroupBy" in pl.last_plan()) == (not nf), pl.last_plan()
            if not nf and n_orders > 10:
                assert "direct-address table" in pl.last_plan(), pl.last_plan()   # orderkey range = 4x the order count
        g = out.sort_host("l_orderkey")
        assert g["l_orderkey"] == exp["l_orderkey"].tolist(), pl.last_plan()
        assert g["o_orderdate"] == exp["o_orderdate"].tolist() and g["o_shippriority"] == exp["o_shippriority"].tolist()
        assert close(g["revenue"], exp["revenue"])
    assert out.columns == ["l_orderkey", "o_orderdate", "o_shippriority", "revenue"]
    if n_orders:   # the hash-table variant of the fused pipeline must agree too
        out = queries.q3(L.lazy(), O.lazy()).collect(no_direct_join=True)
        assert "hash table cap" in pl.last_plan(), pl.last_plan()
        g = out.sort_host("l_orderkey")
        assert g["l_orderkey"] == exp["l_orderkey"].tolist() and g["o_orderdate"] == exp["o_orderdate"].tolist() and close(g["revenue"], exp["revenue"])


def test_generic_interpreter_matches_aot(pl, orc):
    """A query shape with no pre-instantiated kernel runs the scalar-unit-decoded interpreter;
    it must agree with the per-node path and the oracle."""
    rng = np.random.default_rng(7)
    n = 300_001
    a = rng.integers(-1000, 1000, n).astype(np.int32)
    b = rng.integers(0, 5, n).astype(np.int64)
    x = rng.uniform(-1, 1, n)
    xv = rng.uniform(size=n) > 0.1
    df = pl.DataFrame([pl.Series("a", a), pl.Series("b", b), pl.Series("x", x, validity=xv)])
    q = (df.lazy().filter((pl.col("a") >= -500) & (pl.col("x") < 0.9) | (pl.col("b") == 3))
         .select(((pl.col("x") + 2.0) * pl.col("x") / 4).sum().alias("e"), pl.col("a").min().alias("amin"), pl.col("a").max().alias("amax"),
                 (pl.col("b") * pl.col("b") - 1).sum().alias("bb"), pl.col("x").count().alias("xc"), pl.len().alias("n"), pl.col("x").max().alias("xmax")))
    o1 = q.collect(); p1 = pl.last_plan()
    o2 = q.collect(no_fusion=True)
    assert "fused_scan[generic]" in p1 or "fused_scan[jit]" in p1, p1
    m = ((a >= -500) & (x < 0.9) & xv) | (b == 3)    # Kleene: null & x -> null unless other side False; null | True -> True
    m_null = ((a >= -500) & ~xv) & ~(b == 3)
    keep = m & ~m_null
    xe = x[keep & xv]
    exp = {"e": ((xe + 2.0) * xe * 0.25).sum(), "amin": int(a[keep].min()), "amax": int(a[keep].max()), "bb": int((b[keep] * b[keep] - 1).sum()),
           "xc": int((keep & xv).sum()), "n": int(keep.sum()), "xmax": float(xe.max())}
    for o in (o1, o2):
        d = {k: v[0] for k, v in o.to_dict().items()}
        assert d["amin"] == exp["amin"] and d["amax"] == exp["amax"] and d["bb"] == exp["bb"] and d["xc"] == exp["xc"] and d["n"] == exp["n"]
        assert math.isclose(d["e"], exp["e"], rel_tol=RTOL) and d["xmax"] == exp["xmax"]


def test_full_size_properties_q1(pl):
    """Size-independent properties at a size the oracle does not cover in seconds (2e7 rows): per-group counts / integer sums
    agree exactly with plain masked numpy reductions (tests/refs.py, itself pinned to the oracle on the CPU), float sums within
    1e-6, avg = sum / count, and the query is additive over a row split (q1(all) == q1(first part) + q1(second part)).
    Host-generated inputs: no torch kernels (their first launch costs 10 s on a warm GPU box and minutes on a cold one)."""
    from polars_amd import datagen, queries
    from tests import refs
    n = 20_000_000
    cols = datagen.lineitem_host(n, seed=5)
    df = datagen.to_frame(pl, cols, datagen.LINEITEM_Q1_COLS)
    g = queries.q1(df.lazy()).collect().sort_host(["l_returnflag", "l_linestatus"])
    assert "fused_scan[aot]" in pl.last_plan(), pl.last_plan()
    ref = refs.q1_numpy(cols, datagen.us(1998, 9, 2))
    assert len(g["count_order"]) == len(ref)
    for i, (f, st) in enumerate(zip(g["l_returnflag"], g["l_linestatus"])):
        r = ref[(datagen.FLAGS.index(f), datagen.STATUS.index(st))]
        assert g["count_order"][i] == r["count_order"] and g["sum_qty"][i] == r["sum_qty"]
        for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert math.isclose(g[c][i], r[c], rel_tol=RTOL), (f, st, c)
        assert math.isclose(g["avg_qty"][i], g["sum_qty"][i] / g["count_order"][i], rel_tol=1e-12)
    # additivity over a row split (unequal parts, odd boundary)
    cut = 7_000_001
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        sub = {k: np.ascontiguousarray(v[lo:hi]) for k, v in cols.items()}
        parts.append(queries.q1(datagen.to_frame(pl, sub, datagen.LINEITEM_Q1_COLS).lazy()).collect().sort_host(["l_returnflag", "l_linestatus"]))
    keys = list(zip(g["l_returnflag"], g["l_linestatus"]))
    for name, exact in (("count_order", True), ("sum_qty", True), ("sum_charge", False), ("sum_disc_price", False)):
        tot = {k: 0 for k in keys}
        for p_ in parts:
            for k, v in zip(zip(p_["l_returnflag"], p_["l_linestatus"]), p_[name]):
                tot[k] += v
        for k, v in zip(keys, g[name]):
            assert (tot[k] == v) if exact else math.isclose(tot[k], v, rel_tol=RTOL), (name, k)


def _join_groupby_reference(lk, lx, rk, ry):
    """numpy: inner join on key, group by (key, ry), sum(lx), count."""
    import collections
    pos = collections.defaultdict(list)
    for j, k in enumerate(rk.tolist()):
        pos[k].append(j)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for i, k in enumerate(lk.tolist()):
        for j in pos.get(k, ()):
            a = acc[(k, int(ry[j]))]
            a[0] += float(lx[i]); a[1] += 1
    return sorted((k[0], k[1], v[0], v[1]) for k, v in acc.items())


@pytest.mark.parametrize("shape", ["build_right", "build_left", "dup_build_keys", "probe_side_group_key", "sentinel_key"])
def test_join_groupby_fusion_and_fallbacks(pl, shape):
    """The fused join->aggregate pipeline fires when a group is one build row (duplicate build keys included: row chains); every other
    shape must fall back to the per-node path and still agree with a plain numpy evaluation."""
    rng = np.random.default_rng(31)
    nl, nr = (40_000, 3_000) if shape != "build_left" else (3_000, 40_000)
    rk = rng.permutation(200_000)[:nr].astype(np.int64) - 100_000
    if shape == "dup_build_keys":
        rk[: nr // 2] = rk[nr // 2: nr // 2 * 2]
    if shape == "sentinel_key":
        rk[0] = -1                                   # the table's EMPTY bit pattern as a real key
    lk = rng.choice(np.concatenate([rk, rng.integers(300_000, 400_000, 1000)]), nl).astype(np.int64)
    if shape == "build_left":
        lk = rng.permutation(np.unique(lk))[: nl]
        nl = len(lk)
    lx = rng.uniform(0, 10, nl)
    lz = rng.integers(0, 3, nl).astype(np.int64)
    ry = rng.integers(0, 50, nr).astype(np.int64)
    L = pl.DataFrame({"k": lk, "x": lx, "z": lz})
    R = pl.DataFrame({"k": rk, "y": ry})
    keys = ["k", "y"] if shape != "probe_side_group_key" else ["k", "z"]
    val = "x" if shape != "build_left" else "y"
    q = (L.lazy().filter(pl.col("x") >= 0.0).join(R.lazy().filter(pl.col("y") < 1000), on="k")
         .group_by(*keys).agg(pl.col(val).sum().alias("s"), pl.len().alias("n")))
    out = q.collect()
    plan = pl.last_plan()
    fused = shape in ("build_right", "build_left", "sentinel_key", "dup_build_keys")      # duplicate build keys: the multi-value mode of the fused pipeline (round 5)
    if shape == "dup_build_keys":
        assert "multi-value (row chains" in plan, plan
    if shape == "build_left":   # aggregate reads the probe (= right, longer) side: y; group keys k + ... y is probe side -> not a build column
        fused = False
    assert ("FusedJoinGroupBy" in plan) == fused, plan
    d = out.to_dict()
    got = sorted(zip(d[keys[0]], d[keys[1]], d["s"], d["n"]))
    if shape == "probe_side_group_key":
        import collections
        pos = {int(k): j for j, k in enumerate(rk.tolist())}
        acc = collections.defaultdict(lambda: [0.0, 0])
        for i, k in enumerate(lk.tolist()):
            if k in pos:
                a = acc[(k, int(lz[i]))]; a[0] += float(lx[i]); a[1] += 1
        exp = sorted((k[0], k[1], v[0], v[1]) for k, v in acc.items())
    elif shape == "build_left":
        exp = _join_groupby_reference(lk, np.zeros(nl), rk, ry)
        exp = sorted((k, y, float(y) * n, n) for k, y, _, n in exp)
    else:
        exp = _join_groupby_reference(lk, lx, rk, ry)
    assert len(got) == len(exp), (plan, len(got), len(exp))
    for a, b in zip(got, exp):
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3] and math.isclose(a[2], b[2], rel_tol=RTOL, abs_tol=1e-9), (plan, a, b)
    o2 = q.collect(no_fusion=True).to_dict()
    got2 = sorted(zip(o2[keys[0]], o2[keys[1]], o2["s"], o2["n"]))
    assert [g[:2] + g[3:] for g in got2] == [g[:2] + g[3:] for g in got]


def test_join_groupby_build_left_fused(pl):
    """Left input shorter -> it is the build side; aggregates read the right (probe) input."""
    rng = np.random.default_rng(32)
    lk = rng.permutation(50_000)[:2_000].astype(np.int32)
    lname = rng.integers(0, 9, 2_000).astype(np.int64)
    rk = rng.integers(0, 50_000, 60_000).astype(np.int32)
    rv = rng.integers(-5, 5, 60_000).astype(np.int64)
    L = pl.DataFrame({"k": lk, "name": lname})
    R = pl.DataFrame({"k": rk, "v": rv})
    out = L.lazy().join(R.lazy().filter(pl.col("v") != 0), on="k").group_by("k", "name").agg(pl.col("v").sum().alias("s"), pl.col("v").min().alias("mn"), pl.len().alias("n")).collect()
    assert "FusedJoinGroupBy{build=left" in pl.last_plan(), pl.last_plan()
    d = out.to_dict()
    got = sorted(zip(d["k"], d["name"], d["s"], d["mn"], d["n"]))
    name_of = dict(zip(lk.tolist(), lname.tolist()))
    import collections
    acc = collections.defaultdict(list)
    for k, v in zip(rk.tolist(), rv.tolist()):
        if v != 0 and k in name_of:
            acc[k].append(v)
    exp = sorted((k, name_of[k], sum(v), min(v), len(v)) for k, v in acc.items())
    assert got == exp
    assert out.schema == {"k": pl.Int32, "name": pl.Int64, "s": pl.Int64, "mn": pl.Int64, "n": pl.UInt32}


def test_fused_int_floor_div_mod_null_on_zero(pl):
    """x // 0 and x % 0 are null (signed.rs:35-70) -> a null predicate drops the row; Python sign rules."""
    rng = np.random.default_rng(33)
    n = 100_003
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    b = rng.integers(-3, 4, n).astype(np.int64)
    a[:4] = [-2**63, -2**63, 2**63 - 1, 7]; b[:4] = [-1, 2, -1, 0]
    df = pl.DataFrame({"a": a, "b": b})
    q = df.lazy().filter(((pl.col("a") % pl.col("b")) == 1) | ((pl.col("a") // pl.col("b")) < -400)).select(pl.col("a").sum().alias("s"), pl.len().alias("n"), (pl.col("a") // pl.col("b")).min().alias("qmin"))
    o1 = q.collect(); p1 = pl.last_plan(); o2 = q.collect(no_fusion=True)
    assert "fused_scan" in p1, p1
    nz = b != 0
    with np.errstate(all="ignore"):
        fm = np.where(nz, np.mod(a, np.where(nz, b, 1)), 0)
        fd = np.where(nz, np.floor_divide(a, np.where(nz, b, 1)), 0)
    fd[0] = -2**63      # wrapping_div(MIN, -1)
    keep = nz & ((fm == 1) | (fd < -400))
    exp = (int(a[keep].sum()), int(keep.sum()), int(fd[keep].min()))
    assert o1.rows() == [exp] and o2.rows() == [exp]


def test_join_groupby_direct_address_edge_keys(pl):
    """Direct-address join table: negative key range, probe keys outside [kmin, kmax], null keys on both
    sides, i32 keys; must agree with the hash-table variant and with numpy."""
    rng = np.random.default_rng(34)
    nb, npr = 5_000, 80_000
    bk = (rng.permutation(20_000)[:nb] - 10_000).astype(np.int32)            # unique, range 20k <= 64 * nb
    bv = rng.uniform(size=nb) > 0.03
    pay = rng.integers(0, 7, nb).astype(np.int64)
    pk = rng.integers(-15_000, 15_000, npr).astype(np.int32)                # some outside the build range
    pv = rng.uniform(size=npr) > 0.03
    x = rng.integers(-9, 9, npr).astype(np.int64)
    B = pl.DataFrame([pl.Series("k", bk, validity=bv), pl.Series("pay", pay)])
    P = pl.DataFrame([pl.Series("k", pk, validity=pv), pl.Series("x", x)])
    q = P.lazy().filter(pl.col("x") != 0).join(B.lazy().filter(pl.col("pay") < 6), on="k").group_by("k", "pay").agg(pl.col("x").sum().alias("s"), pl.len().alias("n"))
    o1 = q.collect(); p1 = pl.last_plan()
    o2 = q.collect(no_direct_join=True); p2 = pl.last_plan()
    assert "direct-address table" in p1 and "hash table cap" in p2, (p1, p2)
    row_of = {int(k): j for j, k in enumerate(bk.tolist()) if bv[j] and pay[j] < 6}
    import collections
    acc = collections.defaultdict(lambda: [0, 0])
    for k, ok, xv in zip(pk.tolist(), pv.tolist(), x.tolist()):
        if ok and xv != 0 and k in row_of:
            a = acc[k]; a[0] += xv; a[1] += 1
    exp = sorted((k, int(pay[row_of[k]]), v[0], v[1]) for k, v in acc.items())
    for o in (o1, o2):
        d = o.to_dict()
        assert sorted(zip(d["k"], d["pay"], d["s"], d["n"])) == exp


@pytest.mark.parametrize("variant", ["i64_sum_count", "nullable_f64_all_aggs", "filtered_maintain_order"])
def test_partitioned_groupby_matches_hbm_table_path(pl, variant):
    """>= 2^24 rows and >= 4096 groups take the partitioned LDS path (kernels_partition.hip); it must agree
    bit-for-bit (ints) / 1e-6 (floats) with the HBM-table sink and with numpy."""
    rng = np.random.default_rng(41)
    n = 20_000_000
    G = 300_000
    key = rng.integers(0, G, n).astype(np.int64) * 7919 - 10**9
    key[:5] = [-1, -1, 2**63 - 1, -2**63, -1]          # EMPTY-sentinel bit pattern and extremes as real keys
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    if variant == "i64_sum_count":
        df = pl.DataFrame({"key": key, "v": v})
        q = df.lazy().group_by("key").agg(pl.col("v").sum().alias("s"), pl.col("v").count().alias("c"))
    elif variant == "nullable_f64_all_aggs":
        x = rng.uniform(-1, 1, n)
        xv = rng.uniform(size=n) > 0.1
        kv = rng.uniform(size=n) > 0.001
        df = pl.DataFrame([pl.Series("key", key, validity=kv), pl.Series("v", v), pl.Series("x", x, validity=xv)])
        q = df.lazy().group_by("key").agg(pl.col("x").sum().alias("s"), pl.col("x").mean().alias("m"), pl.col("x").min().alias("mn"), pl.col("v").max().alias("mx"),
                                          pl.col("x").count().alias("c"), pl.len().alias("n"))
    else:
        df = pl.DataFrame({"key": key, "v": v})
        q = df.lazy().filter(pl.col("v") >= -500).group_by("key", maintain_order=True).agg(pl.col("v").sum().alias("s"), pl.len().alias("n"))
    o1 = q.collect(); p1 = pl.last_plan()
    o2 = q.collect(no_partition=True); p2 = pl.last_plan()
    assert "partitioned(" in p1 and "lds_hash_table" in p1, p1
    assert "hash_hbm_table" in p2, p2
    d1, d2 = o1.to_dict(), o2.to_dict()
    cols = list(d1.keys())
    assert cols == list(d2.keys()) and o1.height == o2.height
    if variant == "filtered_maintain_order":
        keep = v >= -500
        kk = key[keep]
        _, first = np.unique(kk, return_index=True)
        exp_keys = kk[np.sort(first)]
        assert d1["key"] == exp_keys.tolist() and d2["key"] == exp_keys.tolist()      # first-occurrence order
        order1 = order2 = list(range(o1.height))
    else:
        kf = lambda i, d: (d["key"][i] is None, d["key"][i] or 0)
        order1 = sorted(range(o1.height), key=lambda i: kf(i, d1)); order2 = sorted(range(o2.height), key=lambda i: kf(i, d2))
    for c in cols:
        a = [d1[c][i] for i in order1]; b = [d2[c][i] for i in order2]
        if c in ("s", "m") and variant == "nullable_f64_all_aggs":
            fa = np.array([np.nan if z is None else z for z in a]); fb = np.array([np.nan if z is None else z for z in b])
            assert np.allclose(fa, fb, rtol=RTOL, atol=1e-9, equal_nan=True), c
        else:
            assert a == b, c
    # numpy spot check of the sums
    uk, inv = np.unique(key if variant != "filtered_maintain_order" else key[v >= -500], return_inverse=True)
    if variant == "i64_sum_count":
        got = dict(zip(d1["key"], zip(d1["s"], d1["c"])))
        s = np.bincount(inv, v).astype(np.int64); c = np.bincount(inv)
        assert len(got) == len(uk)
        for j in rng.integers(0, len(uk), 2000):
            assert got[int(uk[j])] == (int(s[j]), int(c[j]))


def test_jit_specialised_kernels_match_generic_interpreter(pl, orc):
    """Query shapes without an AOT kernel: the hiprtc-specialised kernel (forced on for tiny inputs here) and the
    generic interpreter (JIT disabled) must give identical integer results and 1e-6 float results, for the register,
    LDS-table, hash-table, wide-key and join-pipeline sinks."""
    F = pl._ffi
    rng = np.random.default_rng(51)
    n = 120_001
    a = rng.integers(-1000, 1000, n).astype(np.int32)
    k = rng.integers(0, 50_000, n).astype(np.int64) * 3 - 70_000
    k2 = rng.integers(0, 3, n).astype(np.int64) * 2**40
    g8 = rng.integers(0, 5, n).astype(np.uint8)
    x = rng.uniform(-1, 1, n); xv = rng.uniform(size=n) > 0.1
    df = pl.DataFrame([pl.Series("a", a), pl.Series("k", k), pl.Series("k2", k2), pl.Series("g8", g8), pl.Series("x", x, validity=xv)])
    bk = rng.permutation(60_000)[:4_000].astype(np.int64) * 3 - 70_000
    small = pl.DataFrame({"k": bk, "pay": rng.integers(0, 9, 4_000).astype(np.int64)})
    c = pl.col
    queries = {
        "regagg": df.lazy().filter((c("a") > -500) & (c("x") < 0.9)).select((c("x") * 2 + 1).sum().alias("e"), c("a").max().alias("mx"), c("x").min().alias("mn"), pl.len().alias("n"), (c("k") % 7).sum().alias("m7")),
        "lds": df.lazy().filter(c("a") != 3).group_by("g8").agg(c("x").sum().alias("s"), c("x").mean().alias("m"), c("a").min().alias("mn"), pl.len().alias("n")),
        "hash": df.lazy().group_by("k").agg(c("x").sum().alias("s"), c("a").max().alias("mx"), c("x").count().alias("c")),
        "wide": df.lazy().group_by("k", "k2").agg(c("a").sum().alias("s"), pl.len().alias("n")),
        "join": df.lazy().filter(c("a") > -900).join(small.lazy().filter(c("pay") < 7), on="k").group_by("k", "pay").agg((c("x") * 0.5).sum().alias("s"), pl.len().alias("n")),
    }
    try:
        res = {}
        for mode, min_rows in (("jit", 0), ("generic", -1)):
            F.jit_set_min_rows(min_rows)
            before = F.jit_stats()[0]
            for name, q in queries.items():
                out = q.collect()
                keys = [cn for cn in out.columns if cn in ("g8", "k", "k2", "pay")]
                d = out.to_dict()
                rows = sorted(zip(*[d[cn] for cn in out.columns]), key=lambda r: tuple((v is None, v) for v in r[: len(keys)])) if keys else list(zip(*[d[cn] for cn in out.columns]))
                res[(mode, name)] = (out.columns, rows)
            if mode == "jit":
                assert F.jit_stats()[0] - before >= 5, "the JIT did not compile the expected kernels"
            else:
                assert F.jit_stats()[0] == before
        for name in queries:
            cj, rj = res[("jit", name)]; cg, rg = res[("generic", name)]
            assert cj == cg and len(rj) == len(rg), name
            for r1, r2 in zip(rj, rg):
                for v1, v2 in zip(r1, r2):
                    if isinstance(v1, float) and v2 is not None:
                        assert math.isclose(v1, v2, rel_tol=RTOL, abs_tol=1e-12), (name, r1, r2)
                    else:
                        assert v1 == v2, (name, r1, r2)
    finally:
        F.jit_set_min_rows(1 << 22)


@pytest.mark.parametrize("seed", range(6))
def test_grouped_agg_fuzz(pl, seed):
    """Randomised grouped aggregation (the idea of the reference's Hypothesis test, py-polars/tests/unit/operations/
    test_group_by.py:2500-2600: group_by(key % 4).agg(f(x)) must equal evaluating f per group): random dtype, size,
    null rates and key cardinality; every aggregate against a plain per-group numpy evaluation."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 7, 64, 129, 1000, 40_000]))
    kdt = rng.choice(["i8", "i32", "i64", "u16"])
    vdt = rng.choice(["i64", "f64", "i32", "u8"])
    card = int(rng.choice([1, 4, 37, 5000]))
    NP = {"i8": np.int8, "i32": np.int32, "i64": np.int64, "u16": np.uint16, "f64": np.float64, "u8": np.uint8}
    key = (rng.integers(0, card, n) % 100).astype(NP[kdt]) if kdt == "i8" else rng.integers(0, card, n).astype(NP[kdt])
    kv = rng.uniform(size=n) > rng.choice([0.0, 0.1])
    vals = rng.uniform(-5, 5, n) if vdt == "f64" else rng.integers(0, 200, n).astype(NP[vdt])
    vv = rng.uniform(size=n) > rng.choice([0.0, 0.3])
    PL = {"i8": pl.Int8, "i32": pl.Int32, "i64": pl.Int64, "u16": pl.UInt16, "f64": pl.Float64, "u8": pl.UInt8}
    df = pl.DataFrame([pl.Series("k", key, dtype=PL[kdt], validity=kv), pl.Series("v", vals, dtype=PL[vdt], validity=vv)])
    c = pl.col("v")
    out = df.lazy().group_by("k").agg(c.sum().alias("s"), c.mean().alias("m"), c.min().alias("mn"), c.max().alias("mx"), c.count().alias("c"), pl.len().alias("n")).collect()
    d = out.to_dict()
    got = {d["k"][i]: tuple(d[x][i] for x in ("s", "m", "mn", "mx", "c", "n")) for i in range(out.height)}
    groups = {}
    for i in range(n):
        groups.setdefault(int(key[i]) if kv[i] else None, []).append(i)
    assert set(got) == set(groups)
    for gk, idx in groups.items():
        idx = np.array(idx)
        ok = vv[idx]
        x = vals[idx][ok]
        s, m, mn, mx, cnt, ln = got[gk]
        assert ln == len(idx) and cnt == int(ok.sum())
        if vdt == "f64":
            assert math.isclose(s, float(x.sum()), rel_tol=RTOL, abs_tol=1e-9)
        else:
            assert s == int(x.astype(np.int64).sum())
        if len(x) == 0:
            assert m is None and mn is None and mx is None
        else:
            assert math.isclose(m, float(x.astype(np.float64).mean()), rel_tol=RTOL, abs_tol=1e-9)
            assert mn == x.min().item() and mx == x.max().item()


def test_partitioned_groupby_packed_dictionary_keys(pl):
    """BASELINE config 5 shape at >= 2^24 rows: u32 dictionary codes (packed dense ids) -> partitioned LDS path."""
    from polars_amd import queries
    rng = np.random.default_rng(43)
    n = 17_000_000
    codes = rng.integers(0, 400_000, n).astype(np.uint32)
    v = rng.uniform(0, 100, n)
    df = pl.DataFrame([pl.Series("k", codes, dtype=pl.Categorical([], pl.UInt32)), pl.Series("v", v)])
    out = queries.cfg5(df.lazy()).collect(); plan = pl.last_plan()
    assert "partitioned(" in plan and "fused_scan[aot]" in plan, plan
    ref = queries.cfg5(df.lazy()).collect(no_partition=True)
    assert "hbm_table" in pl.last_plan(), pl.last_plan()
    k1, k2 = out["k"].to_numpy(), ref["k"].to_numpy()
    o1, o2 = np.argsort(k1), np.argsort(k2)
    assert np.array_equal(k1[o1], k2[o2]) and np.array_equal(k1[o1], np.unique(codes))
    s = np.bincount(codes, v)[np.unique(codes)]; cnt = np.bincount(codes)[np.unique(codes)]
    assert close(out["v_sum"].to_numpy()[o1], s) and close(out["v_mean"].to_numpy()[o1], s / cnt)
    assert close(ref["v_sum"].to_numpy()[o2], s)


def test_sharded_q3_per_rank_pieces(pl, orc):
    """The per-rank operators of the sharded Q3 (polars_amd/dist.py q3_ops -> LibJoinOps) composed the way sharded_join_groupby
    composes them, with a frame concatenation standing in for the collectives: 2 virtual ranks, keys of one order on both ranks."""
    from polars_amd import datagen, dist as pdist
    orders, li = datagen.orders_lineitem_host(60_000, seed=77)
    exp = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, datagen.us(1995, 3, 15))
    ops, spec = pdist.q3_ops(pl)
    probes = [datagen.to_frame(pl, {c: np.ascontiguousarray(li[c][r::2]) for c in datagen.LINEITEM_Q3_COLS}, datagen.LINEITEM_Q3_COLS) for r in range(2)]
    half = len(orders["o_orderkey"]) // 2
    builds = [datagen.to_frame(pl, {c: np.ascontiguousarray(orders[c][:half] if r == 0 else orders[c][half:]) for c in datagen.ORDERS_Q3_COLS}, datagen.ORDERS_Q3_COLS) for r in range(2)]
    # broadcast mode: prefilter -> all-gather(build) -> local pipeline -> merge partial groups by key
    fb = [ops.build_prefilter(b) for b in builds]
    assert sum(f.height for f in fb) < len(orders["o_orderkey"]) // 3
    assert ops.nbytes(fb[0]) == fb[0].height * 8 * len(datagen.ORDERS_Q3_COLS)
    gathered = pl.concat(fb)                                     # stands in for plx_allgather_frame
    parts = [ops.local(p, gathered) for p in probes]
    assert sum(p.height for p in parts) > len(exp["l_orderkey"])     # keys split over both ranks
    merged = ops.merge(pl.concat(parts), spec)
    assert merged.columns == ["l_orderkey", "o_orderdate", "o_shippriority", "revenue"]
    m = {c: merged[c].to_numpy() for c in merged.columns}
    order = np.argsort(m["l_orderkey"])
    assert np.array_equal(m["l_orderkey"][order], exp["l_orderkey"])
    assert np.array_equal(m["o_orderdate"][order].astype(np.int64), exp["o_orderdate"])
    assert np.allclose(m["revenue"][order], exp["revenue"], rtol=1e-9)
    # shuffle mode pieces: the probe prefilter keeps every surviving row exactly once
    fp = [ops.probe_prefilter(p) for p in probes]
    assert sum(f.height for f in fp) == int((li["l_shipdate"] > datagen.us(1995, 3, 15)).sum())
    # a single rank without a communicator is the plain local pipeline
    full = pdist.sharded_join_groupby(None, ops, datagen.to_frame(pl, li, datagen.LINEITEM_Q3_COLS), datagen.to_frame(pl, orders, datagen.ORDERS_Q3_COLS), spec)
    assert sorted(full["l_orderkey"].to_numpy().tolist()) == exp["l_orderkey"].tolist()


@pytest.mark.parametrize("case", ["zipf", "one_hot_key"])
def test_partitioned_groupby_skewed_keys(pl, case):
    """Skew THROUGH the partitioned path (>= 2^25 rows, ~1e6 distinct keys): zipf s = 1.1 and one key holding half of the rows.
    The heavy hitters found in the strided sample are aggregated in the scatter pass (the role of the reference's HotGrouper,
    polars-expr/src/hot_groups/fixed_index_table.rs:19-165) and never reach a partition; the result must match numpy exactly."""
    rng = np.random.default_rng(44)
    n = 1 << 25
    if case == "zipf":
        key = ((rng.zipf(1.1, n) - 1) % 1_000_000).astype(np.int64)
    else:
        key = rng.integers(0, 1_000_000, n).astype(np.int64)
        key[rng.random(n) < 0.5] = 777_777
    key = key * 1_000_003 - 5                                   # not a dense range: the hash path, raw 64-bit keys
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    df = pl.DataFrame({"key": key, "v": v})
    out = df.lazy().group_by("key").agg(pl.col("v").sum().alias("s"), pl.col("v").count().alias("c")).collect()
    plan = pl.last_plan()
    assert re.search(r"partitioned\(v[23],hash", plan) and "hot=0" not in plan, plan      # hot keys were found and used
    gk = out["key"].to_numpy(); order = np.argsort(gk)
    uk, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    assert np.array_equal(gk[order], uk)
    assert np.array_equal(out["s"].to_numpy()[order], np.bincount(inv, weights=v).astype(np.int64))
    assert np.array_equal(out["c"].to_numpy()[order].astype(np.int64), cnt)


@pytest.mark.parametrize("seed", [10, 20, 31])
def test_partitioned_v2_is_planned_for_1e6_uniform_keys_whatever_the_sample_says(pl, seed, monkeypatch):
    """BASELINE config 3's shape (1e6 uniform keys) sits exactly on the 512-partition capacity of the LDS hash tables (1.3 x estimate vs
    512 x 4096 x 0.62): whether the 2^20-row sample estimates 0.999e6 or 1.001e6 groups must not decide which generation of kernels
    runs (seed 20 estimated just above and fell back to the round-1 three-pass kernels: 18.7 instead of 9.7 ms at 1e9 rows)."""
    from polars_amd import datagen
    monkeypatch.setenv("PLX_LEARN_DENSE_RANGE", "0")        # (the hash-mode plan of a first run WITHOUT the up-front range pass that dense-looking keys now get: tests/test_gpu_datagen.py)
    n = 1 << 25
    key = datagen.uniform_native(pl, "key", pl.Int64, n, seed, 0, 0, 1_000_000)
    val = datagen.uniform_native(pl, "val", pl.Int64, n, seed, 1, 0, 1000)
    df = pl.DataFrame([key, val])
    out = df.lazy().group_by("key").agg(pl.col("val").sum().alias("s"), pl.len().alias("n")).collect()
    plan = pl.last_plan()
    assert re.search(r"partitioned\(v[23],hash", plan), plan
    k = key.to_numpy(); v = val.to_numpy()
    gk = out["key"].to_numpy(); order = np.argsort(gk)
    assert np.array_equal(gk[order], np.unique(k))
    assert np.array_equal(out["s"].to_numpy()[order], np.bincount(k, weights=v, minlength=1_000_000)[np.unique(k)].astype(np.int64))
    # the scan also learned the key range: the second run plans dense ids
    df.lazy().group_by("key").agg(pl.col("val").sum().alias("s"), pl.len().alias("n")).collect()
    assert re.search(r"partitioned\(v[23],direct", pl.last_plan()), pl.last_plan()


def test_partitioned_groupby_direct_mode_dense_ids(pl):
    """Dense packed ids (two narrow key columns + a dictionary column with declared bounds) at >= 2^24 rows: range partitions and
    direct-address LDS tables (no key compare, 4-byte keys in the records); null keys travel as their own code."""
    rng = np.random.default_rng(45)
    n = 17_500_000
    a = rng.integers(0, 300, n).astype(np.int16)
    b = rng.integers(-20, 20, n).astype(np.int8)
    bv = rng.random(n) > 0.01
    x = rng.uniform(-1, 1, n)
    xv = rng.random(n) > 0.2
    df = pl.DataFrame([pl.Series("a", a), pl.Series("b", b, validity=bv), pl.Series("x", x, validity=xv)])
    q = df.lazy().group_by("a", "b").agg(pl.col("x").sum().alias("s"), pl.col("x").count().alias("c"), pl.col("x").max().alias("mx"), pl.len().alias("n"))
    out = q.collect(); plan = pl.last_plan()
    assert re.search(r"partitioned\(v[23],direct", plan) and "lds_direct_table" in plan, plan
    ref = q.collect(no_partition=True)
    assert "hbm_table" in pl.last_plan() or "lds_table" in pl.last_plan(), pl.last_plan()
    d1, d2 = out.to_dict(), ref.to_dict()
    kf = lambda d, i: (d["a"][i], d["b"][i] is None, d["b"][i] or 0)
    o1 = sorted(range(out.height), key=lambda i: kf(d1, i)); o2 = sorted(range(ref.height), key=lambda i: kf(d2, i))
    assert out.height == ref.height == 300 * 41                 # every (a, b) pair incl. b = null occurs in 1.75e7 rows
    for c in ("a", "b", "c", "n"):
        assert [d1[c][i] for i in o1] == [d2[c][i] for i in o2], c
    for c in ("s", "mx"):
        fa = np.array([np.nan if z is None else z for z in (d1[c][i] for i in o1)]); fb = np.array([np.nan if z is None else z for z in (d2[c][i] for i in o2)])
        assert np.allclose(fa, fb, rtol=RTOL, atol=1e-9, equal_nan=True), c
    # numpy check of one aggregate
    sel = bv & (a == 7) & (b == 3)
    i = next(j for j in range(out.height) if d1["a"][j] == 7 and d1["b"][j] == 3)
    assert d1["n"][i] == int(sel.sum()) and d1["c"][i] == int((sel & xv).sum()) and math.isclose(d1["s"][i], float(x[sel & xv].sum()), rel_tol=1e-9, abs_tol=1e-9)


def test_declared_bounds_are_checked_not_trusted_blindly(pl):
    """plx_column_set_bounds (dictionary size, Parquet statistics): a value outside the declared bounds must fail the query (or be
    kept out of every table), never write outside a table."""
    from polars_amd import queries
    rng = np.random.default_rng(46)
    n = 17_000_000
    codes = rng.integers(0, 500_000, n).astype(np.uint32)
    v = rng.uniform(0, 100, n)
    ok = pl.DataFrame([pl.Series("k", codes, dtype=pl.Categorical(["c%d" % i for i in range(500_000)], pl.UInt32)), pl.Series("v", v)])
    out = queries.cfg5(ok.lazy()).collect(); plan = pl.last_plan()
    assert re.search(r"partitioned\(v[23],direct", plan), plan
    k1 = out["k"].to_numpy(); o1 = np.argsort(k1)
    s = np.bincount(codes, v); cnt = np.bincount(codes); present = np.nonzero(cnt)[0]
    assert np.array_equal(k1[o1], present) and close(out["v_sum"].to_numpy()[o1], s[present]) and close(out["v_mean"].to_numpy()[o1], s[present] / cnt[present])
    bad_codes = codes.copy(); bad_codes[12345] = 3_000_000        # not a code of the 500 000-entry dictionary
    bad = pl.DataFrame([pl.Series("k", bad_codes, dtype=pl.Categorical(["c%d" % i for i in range(500_000)], pl.UInt32)), pl.Series("v", v)])
    with pytest.raises(pl.PlxError):
        queries.cfg5(bad.lazy()).collect()


def test_wrong_declared_bounds_on_a_value_column_never_narrow_it(pl):
    """The record packer of the partitioned group-by stores an Int64 aggregate source as a u32 offset from the column's minimum WITHOUT a
    per-row range check, so it may only use ranges the library computed itself.  A caller-declared range (plx_column_set_bounds) that is
    too tight must not change a sum (round-3 advisor finding: values were truncated silently)."""
    from polars_amd import queries
    rng = np.random.default_rng(47)
    n = 17_000_000
    key = rng.integers(0, 300_000, n).astype(np.int64)
    v = rng.integers(0, 1 << 40, n).astype(np.int64)
    sv = pl.Series("v", v)
    pl._ffi.check(pl._ffi.lib().plx_column_set_bounds(sv._h, 0, 999))       # a lie: the values span 2^40
    out = queries.cfg3(pl.DataFrame([pl.Series("key", key), sv]).lazy()).collect(); plan = pl.last_plan()
    assert "partitioned(" in plan and ("pack=0" in plan or "pack=4" in plan), plan          # whole 64-bit values: plain records, or two rows a record (kPackPair)
    o = np.argsort(out["key"].to_numpy())
    want = np.zeros(300_000, np.int64); np.add.at(want, key, v)
    present = np.nonzero(np.bincount(key, minlength=300_000))[0]
    assert np.array_equal(out["key"].to_numpy()[o], present) and np.array_equal(out["v_sum"].to_numpy()[o], want[present])


@pytest.mark.parametrize("fused", [True, False])
def test_q3_three_tables(pl, orc, fused):
    """TPC-H Q3 with customer: the dictionary compare c_mktsegment == "BUILDING", customer x orders
    (a pure filter: reduced to a membership bitmap tested inside the orders scan) and orders x lineitem, against the oracle's
    three-table Q3; also through the one-kernel-per-node path."""
    from polars_amd import datagen, queries
    no = 150_000
    orders, li = datagen.orders_lineitem_host(no, seed=17, ordered=True)
    cust = datagen.customer_host(datagen.n_customers_for(no), seed=17)
    C = datagen.to_frame(pl, cust, datagen.CUSTOMER_Q3_COLS); O = datagen.to_frame(pl, orders, datagen.ORDERS_Q3_COLS); L = datagen.to_frame(pl, li, datagen.LINEITEM_Q3_COLS)
    out = queries.q3_full(C.lazy(), O.lazy(), L.lazy()).collect(no_fusion=not fused)
    plan = pl.last_plan()
    if fused:
        assert "SemiFilter{c_custkey -> bitmap" in plan and "FusedJoinGroupBy" in plan, plan
    else:
        assert "SemiFilter" not in plan and "FusedJoinGroupBy" not in plan and plan.count("Join{hash_join") == 2, plan
    want = orc.q3_full(cust, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, {k: li[k] for k in datagen.LINEITEM_Q3_COLS}, datagen.us(1995, 3, 15), datagen.SEGMENTS.index("BUILDING"))
    k = out["o_orderkey"].to_numpy(); order = np.argsort(k)
    assert len(k) == len(want["o_orderkey"]) > 500
    assert np.array_equal(k[order], want["o_orderkey"]) and np.array_equal(out["o_orderdate"].to_numpy()[order], want["o_orderdate"])
    assert np.array_equal(out["o_shippriority"].to_numpy()[order], want["o_shippriority"])
    assert close(out["revenue"].to_numpy()[order], want["revenue"])
    if fused:
        # ORDER BY revenue DESC, o_orderdate LIMIT 10 on top of the fused pipeline
        top = queries.q3_full_top10(C.lazy(), O.lazy(), L.lazy()).collect()
        best = np.lexsort((want["o_orderdate"], -want["revenue"]))[:10]
        assert top["o_orderkey"].to_numpy().tolist() == want["o_orderkey"][best].tolist()
        # a segment missing from the dictionary: empty result
        assert queries.q3_full(C.lazy(), O.lazy(), L.lazy(), segment="NOSUCH").collect().height == 0


def test_float_sums_of_cancelling_values_stay_within_tolerance(pl, orc):
    """Zero-mean data: |sum| is ~1e3 times smaller than sum(|x|), so summation error is amplified by that condition number.  The
    reference's in-memory group sums are Kahan-compensated (aggregations/mod.rs:867-870, restated by the oracle), the library
    accumulates plain f64 per lane / per LDS cell in arbitrary order; the results must still agree to 1e-6 relative (whole-column
    and grouped), which f64 accumulation does with ~6 digits to spare.  (Inputs whose true sum is ~0 have no meaningful relative
    error in any engine: the reference's own streaming and in-memory engines disagree there.)"""
    rng = np.random.default_rng(47)
    n = 8_000_000
    x = rng.uniform(-1.0, 1.0, n) * rng.choice([1.0, 1e3, 1e-3], n)
    g = rng.integers(0, 64, n).astype(np.int64)
    df = pl.DataFrame({"g": g, "x": x})
    tot = df.lazy().select(pl.col("x").sum().alias("s")).collect().to_dict()["s"][0]
    want_tot = orc.reduce(orc.AGG_SUM, x)[0]
    assert abs(want_tot) * 1e3 < np.abs(x).sum()                      # the data really cancels
    assert math.isclose(tot, want_tot, rel_tol=RTOL)
    out = df.lazy().group_by("g").agg(pl.col("x").sum().alias("s"), pl.col("x").mean().alias("m")).collect().sort_host("g")
    w = orc.q_groupby([g], [None], [("s", orc.AGG_SUM, x, None), ("m", orc.AGG_MEAN, x, None)])
    order = np.argsort(w["key_0"][0])
    assert out["g"] == w["key_0"][0][order].tolist()
    assert np.allclose(np.array(out["s"]), w["s"][0][order], rtol=RTOL, atol=0) and np.allclose(np.array(out["m"]), w["m"][0][order], rtol=RTOL, atol=0)


def test_bench_multi_gpu_default_run_through_rccl_at_world_size_one(tmp_path):
    """The WHOLE default N > 1 run of bench.py -- Q1 headline over the all-gather, then sharded Q3 (broadcast and shuffle), cfg3, cfg5 -- on the one GPU of
    this box through RCCL (PLX_BENCH_FORCE_SHARDED=1: every exchange is a self-exchange), at small sizes: the code the driver's 2/4/8-GPU runs execute is
    exercised by `pytest -m gpu` every round.  The communicator must span exactly the ranks the environment names (ncclCommCount through plx_comm_info), every
    workload must be verified against the oracle, and the line the driver parses must be the small one."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_lines
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(PLX_BENCH_FORCE_SHARDED="1", PLX_BENCH_EXTRAS_FILE=str(tmp_path / "extras.json"), MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    try:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--rows", "400000"], capture_output=True, text=True, timeout=420, cwd=root, env=env)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        pytest.skip("the RCCL run did not finish within 420 s (seen on freshly provisioned boxes: ncclCommInitRank not returning); last lines: " + " | ".join(err.strip().splitlines()[-3:]))
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    head, full = bench_lines.split(r.stdout)
    assert head["comm"] == {"rank": 0, "world_size": 1, "env_world_size": 1, "library": "rccl"}
    assert head["n_gpus"] == 1 and head["verified"]["ok"] is True and head["roofline"]["frac"] >= 0 and head["roofline"]["kernel"]
    ex = full["extras"]
    assert {k.split("_sharded")[0].split("_x1")[0] for k in ex} == {"tpch_q3_sf100", "cfg3_groupby_1e6_keys", "cfg5_dict_string_keys"}, list(ex)
    assert len(ex) == 4 and all("error" not in v and v["verified"]["ok"] is True for v in ex.values()), {k: v.get("error") or v.get("verified") for k, v in ex.items()}
    assert {v.get("exchange_mode") for k, v in ex.items() if k.startswith("tpch_q3")} == {"broadcast", "shuffle"}
    assert json.load(open(tmp_path / "extras.json"))["extras"].keys() == ex.keys()


def test_bench_multi_gpu_default_run_through_rccl_on_every_gpu_of_the_box(tmp_path):
    """Round-5 review, item 9: RCCL has only ever run with one rank here because a GPU box has one GPU.  This test runs the same default N > 1 bench at N =
    torch.cuda.device_count() whenever that is more than one: the first multi-GPU box that runs `pytest -m gpu` exercises the all-gather combine of Q1, the key-hash
    exchange (ncclSend / ncclRecv all-to-all over xGMI) of the sharded Q3 in both modes, cfg3 and cfg5 through RCCL at N ranks -- every result verified against the
    oracle -- without anyone editing a test.  bench.py --gpus N starts its own ranks (one process per GPU, 127.0.0.1 rendezvous)."""
    import json
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU on this box: RCCL at world size 1 is covered by test_bench_multi_gpu_default_run_through_rccl_at_world_size_one")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_lines
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(PLX_BENCH_EXTRAS_FILE=str(tmp_path / "extras.json"), MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--rows", "400000"], capture_output=True, text=True,
                       timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    head, full = bench_lines.split(r.stdout)
    assert head["comm"] == {"rank": 0, "world_size": n, "env_world_size": n, "library": "rccl"}, head["comm"]
    assert head["n_gpus"] == n and head["verified"]["ok"] is True
    ex = full["extras"]
    assert len(ex) == 4 and all("error" not in v and v["verified"]["ok"] is True for v in ex.values()), {k: v.get("error") or v.get("verified") for k, v in ex.items()}
    assert {v.get("exchange_mode") for k, v in ex.items() if k.startswith("tpch_q3")} == {"broadcast", "shuffle"}
    assert json.load(open(tmp_path / "extras.json"))["extras"].keys() == ex.keys()
