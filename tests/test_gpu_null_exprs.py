"""is_null / is_not_null / fill_null(literal) on the GPU: fused (compare-with-self + IFNULL, opcodes the kernels already ran) and,
for all three, the per-node path (the validity bitmap shared as a Boolean column; a select kernel for fill_null).  The lowering is pinned on the CPU
(tests/test_program_eval_cpu.py::test_is_null_is_not_null_fill_null)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_null_expressions_fused_and_per_node(pl):
    rng = np.random.default_rng(8)
    n = 300_000
    a, am = rng.integers(-20, 20, n).astype(np.int64), rng.random(n) < 0.7
    x, xm = rng.normal(size=n), rng.random(n) < 0.6
    k = rng.integers(0, 5, n).astype(np.int64)
    df = pl.DataFrame([pl.Series("a", a, validity=am), pl.Series("x", x, validity=xm), pl.Series("k", k)])
    c = pl.col
    q = (df.lazy().filter(c("a").is_not_null() & (c("x").is_null() | (c("x").fill_null(2.5) > 0.0)))
         .select(pl.len().alias("n"), c("a").fill_null(7).sum().alias("a7"), c("x").fill_null(-1.0).sum().alias("xf")))
    out = q.collect().to_dict()
    assert "FusedFilterAgg" in pl.last_plan(), pl.last_plan()
    keep = am & (~xm | (np.where(xm, x, 2.5) > 0.0))
    assert out["n"][0] == int(keep.sum()) and out["a7"][0] == int(np.where(am, a, 7)[keep].sum())
    assert np.isclose(out["xf"][0], np.where(xm, x, -1.0)[keep].sum(), rtol=1e-9)
    # per-node path for the predicates
    cnt = df.lazy().filter(c("a").is_null()).select(pl.len().alias("n")).collect(no_fusion=True).to_dict()["n"][0]
    assert cnt == int((~am).sum())
    cnt = df.lazy().filter(c("x").is_not_null() & c("k").is_not_null()).select(pl.len().alias("n")).collect(no_fusion=True).to_dict()["n"][0]
    assert cnt == int(xm.sum())
    # fill_null per node (ops::fill_null): unfused aggregate, with_columns (never fused), a Boolean column, and a float32 literal
    assert df.lazy().select(c("a").fill_null(0).sum().alias("s")).collect(no_fusion=True).to_dict()["s"][0] == int(np.where(am, a, 0).sum())
    w = df.lazy().with_columns(c("a").fill_null(-3).alias("a3"), c("x").fill_null(9.5).alias("x9")).collect()
    assert np.array_equal(w["a3"].to_numpy(), np.where(am, a, -3)) and np.array_equal(w["x9"].to_numpy(), np.where(xm, x, 9.5))
    assert w["a3"].null_count() == 0 and w["x9"].null_count() == 0
    b, bm = rng.random(n) < 0.5, rng.random(n) < 0.8
    f32 = rng.normal(size=n).astype(np.float32)
    d2 = pl.DataFrame([pl.Series("b", b, validity=bm), pl.Series("f", f32, validity=bm)])
    w2 = d2.lazy().with_columns(c("b").fill_null(True).alias("bt"), c("f").fill_null(1.5).alias("ff")).collect()
    assert np.array_equal(w2["bt"].to_numpy(), np.where(bm, b, True)) and np.array_equal(w2["ff"].to_numpy(), np.where(bm, f32, np.float32(1.5)))
    g = df.lazy().filter(c("a").is_null()).group_by("k").agg(pl.len().alias("nulls")).collect().sort_host("k")
    assert g["nulls"] == [int(((k == i) & ~am).sum()) for i in g["k"]]


def test_boolean_sum_and_mean(pl):
    rng = np.random.default_rng(10)
    n = 200_000
    a, am = rng.integers(-20, 20, n).astype(np.int64), rng.random(n) < 0.7
    k = rng.integers(0, 6, n).astype(np.int64)
    df = pl.DataFrame([pl.Series("a", a, validity=am), pl.Series("k", k)])
    c = pl.col
    g = df.lazy().group_by("k").agg(c("a").is_null().sum().alias("nulls"), (c("a") > 3).sum().alias("gt3"), (c("a") > 3).mean().alias("frac")).collect()
    assert "FusedFilterGroupBy" in pl.last_plan() and g.schema["nulls"] == pl.UInt32 and g.schema["frac"] == pl.Float64
    d = g.sort_host("k")
    for i, kv in enumerate(d["k"]):
        m = k == kv
        assert d["nulls"][i] == int((m & ~am).sum()) and d["gt3"][i] == int((m & am & (a > 3)).sum())
        assert np.isclose(d["frac"][i], (a > 3)[m & am].mean(), rtol=1e-12)
