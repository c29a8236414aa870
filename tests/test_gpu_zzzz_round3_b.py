"""GPU tests added in round 3, second batch (partitioned join probe).  `gpu_unvalidated` until a gpurun session has passed them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu          # validated on hardware: gpurun_out/r03g


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
