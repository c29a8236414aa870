"""The radix-partitioned join probe (kernels_partition.hip: partitioned_probe_hits / partitioned_hash_probe_hits): probe rows partitioned by key range
(build keys with a dense range: LDS bitmap slices) or by the key's hash (64-bit keys without one: LDS Bloom filters of the build hash table's keys), the
ordinary probe over the surviving candidates.  Reference: crates/polars-ops/src/frame/join/hash_join/single_keys.rs:16-167, single_keys_inner.rs:11-149."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HASH_MULT = np.uint64(0x9E3779B97F4A7C15)      # odd: k -> k * HASH_MULT mod 2^64 is a bijection (bench.py's hashed-key Q3 uses the same)


def hashed(a):
    return (a.astype(np.uint64) * HASH_MULT).astype(np.int64)


def close(a, b):
    return np.allclose(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64), rtol=1e-6, atol=0)


@pytest.mark.parametrize("ordered", [False, True])
def test_q3_partitioned_probe_matches_the_oracle(pl, orc, monkeypatch, ordered):
    """TPC-H Q3 with the probe side radix-partitioned by key range and probed against LDS-resident bitmap slices (forced: the planner
    only picks it for unordered keys over bitmaps far larger than an L2).  Same groups, same sums as the oracle and as the direct probe.
    Reference: crates/polars-ops/src/frame/join/hash_join/single_keys_inner.rs:11-149 (partitioned probe_inner)."""
    from polars_amd import datagen, queries
    orders, li = datagen.orders_lineitem_host(1_200_000, seed=31, ordered=ordered)
    assert len(li["l_orderkey"]) >= 1 << 22                     # the scatter's program is JIT-compiled: inputs below 2^22 rows would not take the path
    L = datagen.to_frame(pl, li, datagen.LINEITEM_Q3_COLS)
    O = datagen.to_frame(pl, orders, datagen.ORDERS_Q3_COLS)
    exp = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, datagen.us(1995, 3, 15))
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    out = queries.q3(L.lazy(), O.lazy()).collect()
    plan = pl.last_plan()
    assert "partitioned_probe(" in plan and "direct-address table" in plan, plan
    g = out.sort_host("l_orderkey")
    assert g["l_orderkey"] == exp["l_orderkey"].tolist(), plan
    assert g["o_orderdate"] == exp["o_orderdate"].tolist() and g["o_shippriority"] == exp["o_shippriority"].tolist()
    assert close(g["revenue"], exp["revenue"])
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "0")
    out0 = queries.q3(L.lazy(), O.lazy()).collect()
    assert "partitioned_probe(" not in pl.last_plan()
    g0 = out0.sort_host("l_orderkey")
    assert g0["l_orderkey"] == g["l_orderkey"] and close(g0["revenue"], g["revenue"])


def test_partitioned_probe_null_and_out_of_range_probe_keys(pl, monkeypatch):
    """Probe keys that are null, below the build key range or above it match nothing (and must not fail the query); a probe key equal to
    the smallest / largest build key matches."""
    rng = np.random.default_rng(9)
    nb, n = 200_000, (1 << 22) + 12345
    bkey = (np.arange(nb, dtype=np.int64) * 7 + 1000)
    battr = rng.integers(0, 100, nb).astype(np.int64)
    pkey = rng.integers(0, nb * 7 + 3000, n).astype(np.int64)            # below 1000 and above the largest build key: out of range
    pkey[:4] = [bkey[0], bkey[-1], bkey[0] - 1, bkey[-1] + 1]
    valid = rng.random(n) > 0.05
    x = rng.integers(-50, 50, n).astype(np.int64)
    B = pl.DataFrame({"k": bkey, "a": battr})
    P = pl.DataFrame([pl.Series("k", pkey, validity=valid), pl.Series("x", x)])
    c = pl.col
    q = lambda: P.lazy().join(B.lazy(), on="k").group_by("k", "a").agg(c("x").sum().alias("sx"), pl.len().alias("n")).collect()
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    got = q()
    assert "partitioned_probe(" in pl.last_plan(), pl.last_plan()
    g = got.sort_host("k")
    inb = valid & (pkey >= 1000) & ((pkey - 1000) % 7 == 0) & (pkey <= bkey[-1])
    keys, inv = np.unique(pkey[inb], return_inverse=True)
    assert g["k"] == keys.tolist()
    assert g["sx"] == np.bincount(inv, weights=x[inb]).astype(np.int64).tolist() and g["n"] == np.bincount(inv).tolist()
    assert g["a"] == battr[(keys - 1000) // 7].tolist()


@pytest.mark.parametrize("stride,nb", [(1, 100_003), (37, 700_001), (251, 1_200_007), (131, 2_500_003)])
def test_partitioned_probe_cuts_the_key_range_into_equal_slices(pl, monkeypatch, stride, nb):
    """The key range is cut into P equal slices (range / P rounded up to whole bitmap words), whatever the range: ranges that are no power of two, slices
    that fit the LDS as they are (exact) and slices that only fit as Bloom filters; keys on both sides of every slice boundary and the last key of the range
    find their partition (id / slice through the reciprocal + one correction step, partition2_device.hpp make_record2)."""
    rng = np.random.default_rng(stride)
    n = (1 << 22) + 777
    kmin = -12345
    bkey = kmin + np.arange(nb, dtype=np.int64) * stride
    battr = rng.integers(0, 1000, nb).astype(np.int64)
    rng_keys = int(bkey[-1] - kmin) + 1
    pkey = kmin + rng.integers(0, rng_keys, n).astype(np.int64)
    hit = rng.random(n) < 0.4
    pkey[hit] = bkey[rng.integers(0, nb, int(hit.sum()))]
    per = -(-rng_keys // 256)
    slice_ = max((per + 63) // 64 * 64, 512)
    edges = kmin + np.arange(1, 256, dtype=np.int64) * slice_
    edges = edges[edges < bkey[-1]]
    m = len(edges)
    pkey[:m] = edges; pkey[m:2 * m] = edges - 1; pkey[2 * m:3 * m] = edges + 1       # both sides of every slice boundary
    pkey[3 * m:3 * m + 3] = [bkey[0], bkey[-1], bkey[-1] + 1]
    x = rng.integers(-50, 50, n).astype(np.int64)
    B = pl.DataFrame({"k": bkey, "a": battr})
    P = pl.DataFrame({"k": pkey, "x": x})
    c = pl.col
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    got = P.lazy().join(B.lazy(), on="k").group_by("k", "a").agg(c("x").sum().alias("sx"), pl.len().alias("n")).collect()
    plan = pl.last_plan()
    assert "partitioned_probe(" in plan, plan
    assert ("lds_bitmap=" in plan) == (slice_ <= 1 << 20), plan
    inb = (pkey >= kmin) & (pkey <= bkey[-1]) & ((pkey - kmin) % stride == 0)
    cand = int(plan.split("candidates=")[1].split(")")[0])
    assert int(inb.sum()) <= cand <= int(inb.sum()) + int(0.05 * n), (cand, int(inb.sum()))
    g = got.sort_host("k")
    keys, inv = np.unique(pkey[inb], return_inverse=True)
    assert g["k"] == keys.tolist()
    assert g["sx"] == np.bincount(inv, weights=x[inb]).astype(np.int64).tolist() and g["n"] == np.bincount(inv).tolist()
    assert g["a"] == battr[(keys - kmin) // stride].tolist()


@pytest.mark.parametrize("ordered", [False, True])
def test_q3_on_hashed_keys_takes_the_partitioned_hash_probe(pl, orc, monkeypatch, ordered):
    """TPC-H Q3 with orderkey * 0x9E3779B97F4A7C15 mod 2^64 on both sides: no dense key range, so the build side is an open-addressing hash table; the probe
    side is partitioned by the key's hash, filtered against per-partition LDS Bloom filters of the table's keys, and the hash probe runs over the candidates
    (forced here: the planner picks it for tables beyond the caches).  Same groups and sums as the oracle and as the plain hash probe."""
    from polars_amd import datagen, queries
    orders, li = datagen.orders_lineitem_host(1_200_000, seed=33, ordered=ordered)
    orders["o_orderkey"] = hashed(orders["o_orderkey"]); li["l_orderkey"] = hashed(li["l_orderkey"])
    assert len(li["l_orderkey"]) >= 1 << 22
    L = datagen.to_frame(pl, li, datagen.LINEITEM_Q3_COLS)
    O = datagen.to_frame(pl, orders, datagen.ORDERS_Q3_COLS)
    exp = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, datagen.us(1995, 3, 15))
    assert len(exp["l_orderkey"]) > 1000
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    out = queries.q3(L.lazy(), O.lazy()).collect()
    plan = pl.last_plan()
    assert "partitioned_hash_probe(" in plan and "hash table cap=" in plan, plan
    g = out.sort_host("l_orderkey")
    assert g["l_orderkey"] == exp["l_orderkey"].tolist(), plan
    assert g["o_orderdate"] == exp["o_orderdate"].tolist() and g["o_shippriority"] == exp["o_shippriority"].tolist()
    assert close(g["revenue"], exp["revenue"])
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "0")
    out0 = queries.q3(L.lazy(), O.lazy()).collect()
    assert "partitioned_hash_probe(" not in pl.last_plan()
    g0 = out0.sort_host("l_orderkey")
    assert g0["l_orderkey"] == g["l_orderkey"] and close(g0["revenue"], g["revenue"])


@pytest.mark.parametrize("dt", [np.int64, np.uint64])
def test_partitioned_hash_probe_edge_keys(pl, monkeypatch, dt):
    """Sparse 64-bit build keys including the key whose bits equal the table's EMPTY pattern (-1 / 2^64 - 1), 0, the smallest and the largest Int64 (UInt64: keys
    above 2^63); null probe keys and probe keys absent from the build side match nothing; the Bloom filters' false positives are removed by the final key compare."""
    rng = np.random.default_rng(10)
    nb, n = 300_000, (1 << 22) + 4321
    edge = np.array([-1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max], np.int64)
    bkey = np.unique(np.concatenate([rng.integers(-(1 << 62), 1 << 62, nb).astype(np.int64), edge])).view(dt)
    bkey = np.sort(bkey)
    battr = rng.integers(0, 100, len(bkey)).astype(np.int64)
    pkey = np.where(rng.random(n) < 0.3, bkey[rng.integers(0, len(bkey), n)], rng.integers(-(1 << 62), 1 << 62, n).astype(np.int64).view(dt))
    pkey[:4] = edge.view(dt)
    valid = rng.random(n) > 0.05
    valid[:4] = True
    x = rng.integers(-50, 50, n).astype(np.int64)
    B = pl.DataFrame({"k": bkey, "a": battr})
    P = pl.DataFrame([pl.Series("k", pkey, validity=valid), pl.Series("x", x)])
    c = pl.col
    q = lambda: P.lazy().join(B.lazy(), on="k").group_by("k", "a").agg(c("x").sum().alias("sx"), pl.len().alias("n")).collect()
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    got = q()
    plan = pl.last_plan()
    assert "partitioned_hash_probe(" in plan, plan
    cand = int(plan.split("candidates=")[1].split(")")[0])
    inb = valid & np.isin(pkey, bkey)
    assert int(inb.sum()) <= cand <= int(inb.sum()) + int(0.05 * n), (cand, int(inb.sum()))       # every true match is a candidate; few false positives
    g = got.sort_host("k")
    keys, inv = np.unique(pkey[inb], return_inverse=True)
    assert g["k"] == keys.tolist() and {int(v) for v in edge.view(dt)} <= set(g["k"])
    assert g["sx"] == np.bincount(inv, weights=x[inb]).astype(np.int64).tolist() and g["n"] == np.bincount(inv).tolist()
    assert g["a"] == battr[np.searchsorted(bkey, keys)].tolist()
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "0")                                             # the plain hash probe: the compaction without the candidates' filter
    g0 = q().sort_host("k")
    assert "partitioned_hash_probe(" not in pl.last_plan()
    assert g0["k"] == g["k"] and g0["sx"] == g["sx"] and g0["n"] == g["n"] and g0["a"] == g["a"]


@pytest.mark.parametrize("dt", [np.int64, np.uint64])
def test_join_group_by_on_a_table_filled_from_lds(pl, monkeypatch, dt):
    """PLX_JOIN_PART_BUILD=2 forces the windowed build (k::partitioned_join_build) at a size numpy checks: 1.5e6 sparse 64-bit build keys with the EMPTY pattern, 0 and the
    extreme values among them, nulls on both sides, a predicate on the build side; the aggregates live in cells numbered per KEY (not per slot), the output step walks the
    cells.  The plain and the partitioned probe agree with numpy, and so does the plain build."""
    rng = np.random.default_rng(12)
    nb, n = 1_500_000, (1 << 22) + 4321
    edge = np.array([-1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max], np.int64)
    bkey = np.unique(np.concatenate([rng.integers(-(1 << 62), 1 << 62, nb).astype(np.int64), edge])).view(dt)
    bkey = bkey[rng.permutation(len(bkey))]
    bvalid = rng.random(len(bkey)) > 0.02
    bvalid[np.isin(bkey, edge.view(dt))] = True
    battr = rng.integers(0, 100, len(bkey)).astype(np.int64)
    bsel = rng.integers(0, 10, len(bkey)).astype(np.int32)
    bsel[np.isin(bkey, edge.view(dt))] = 1
    pkey = np.where(rng.random(n) < 0.3, bkey[rng.integers(0, len(bkey), n)], rng.integers(-(1 << 62), 1 << 62, n).astype(np.int64).view(dt))
    pkey[:4] = edge.view(dt)
    valid = rng.random(n) > 0.05
    valid[:4] = True
    x = rng.integers(-50, 50, n).astype(np.int64)
    B = pl.DataFrame([pl.Series("k", bkey, validity=bvalid), pl.Series("a", battr), pl.Series("s", bsel)])
    P = pl.DataFrame([pl.Series("k", pkey, validity=valid), pl.Series("x", x)])
    c = pl.col
    q = lambda: P.lazy().join(B.lazy().filter(c("s") != 0), on="k").group_by("k", "a").agg(c("x").sum().alias("sx"), pl.len().alias("n")).collect()
    live = bkey[bvalid & (bsel != 0)]
    order = np.argsort(live, kind="stable")
    live_attr = battr[bvalid & (bsel != 0)][order]
    live = live[order]
    inb = valid & np.isin(pkey, live)
    keys, inv = np.unique(pkey[inb], return_inverse=True)
    want = (keys.tolist(), np.bincount(inv, weights=x[inb]).astype(np.int64).tolist(), np.bincount(inv).tolist(), live_attr[np.searchsorted(live, keys)].tolist())
    for build, probe in (("2", "2"), ("2", "0"), ("0", "2")):
        monkeypatch.setenv("PLX_JOIN_PART_BUILD", build)
        monkeypatch.setenv("PLX_PROBE_PARTITIONED", probe)
        g = q().sort_host("k")
        plan = pl.last_plan()
        assert ("partitioned build(" in plan) == (build == "2") and ("partitioned_hash_probe(" in plan) == (probe == "2"), plan
        assert (g["k"], g["sx"], g["n"], g["a"]) == want, plan
        assert {int(v) for v in edge.view(dt)} <= set(g["k"])
