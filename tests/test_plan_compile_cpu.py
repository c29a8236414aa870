"""Host logic without a GPU: the mirror API's type coercion (plan.py restates
polars-plan type_coercion/binary.rs), the arenas that cross the C ABI, and the engine's fused
program compiler (plx_describe_fusion on schema-only placeholder columns).  The benchmark
queries must hit their pre-instantiated (AOT) kernels."""
import ctypes as C

import pytest

import polars_amd as pl
from polars_amd import _ffi as F
from polars_amd import queries as Q


def ph(name, dtype, n=1 << 20, nullable=False, rng=None):
    h = C.c_uint64()
    F.check(F.lib().plx_column_placeholder(dtype.physical, n, int(nullable), 1 if rng else 0, rng[0] if rng else 0, rng[1] if rng else 0, C.byref(h)))
    return pl.Series._from_handle(name, h.value, dtype)


def lineitem():
    flag, status = pl.Categorical(["A", "N", "R"], pl.UInt8), pl.Categorical(["F", "O"], pl.UInt8)
    return pl.DataFrame([ph("l_shipdate", pl.Datetime), ph("l_returnflag", flag, rng=(0, 2)), ph("l_linestatus", status, rng=(0, 1)),
                         ph("l_quantity", pl.Int64), ph("l_extendedprice", pl.Float64), ph("l_discount", pl.Float64), ph("l_tax", pl.Float64)])


def test_benchmark_queries_hit_aot_kernels():
    cfg = pl.DataFrame([ph("a", pl.Int64), ph("x", pl.Float64), ph("y", pl.Float64)])
    cfgn = pl.DataFrame([ph("a", pl.Int64), ph("x", pl.Float64, nullable=True), ph("y", pl.Float64)])
    gb = pl.DataFrame([ph("key", pl.Int64), ph("v", pl.Int64)])
    gb5 = pl.DataFrame([ph("k", pl.Categorical([], pl.UInt32), rng=(0, 999_999)), ph("v", pl.Float64)])
    expect = [(Q.cfg2(cfg.lazy()), 0), (Q.cfg2(cfgn.lazy()), 1), (Q.cfg1(cfg.lazy()), 2), (Q.q1(lineitem().lazy()), 3), (Q.cfg3(gb.lazy()), 4),
              (Q.cfg5(gb5.lazy()), 5), (Q.cfg3w(pl.DataFrame([ph("k1", pl.Int64), ph("k2", pl.Int64), ph("v", pl.Int64)]).lazy()), 13)]
    for q, sid in expect:
        fusable, got, why, dump = q.describe_fusion()
        assert fusable and got == sid, (sid, got, why, dump)


def test_q1_program_shape():
    fusable, sid, why, dump = Q.q1(lineitem().lazy()).describe_fusion()
    assert fusable, why
    assert "inputs=7" in dump                      # every lineitem column is read exactly once
    # sum_qty, sum_base_price, sum_disc_price, sum_charge, sum(qty as f64), sum_disc, len  -> 7 cells (avg_* reuse sums + len)
    assert dump.count("(") - dump.split("aggs=[")[0].count("(") == 7, dump


def test_unfusable_shapes_report_a_reason():
    df = pl.DataFrame([ph("a", pl.Int64), ph("b", pl.Int64)])
    q = df.lazy().filter(pl.col("a") // pl.col("b") > 3).select(pl.col("a").sum())
    fusable, sid, why, _ = q.describe_fusion()
    assert fusable, why           # 64-bit integer floor-div / mod run inside the fused program (divisor 0 -> null)
    dfx = pl.DataFrame([ph("x", pl.Float64), ph("y", pl.Float64)])
    fusable, sid, why, _ = dfx.lazy().filter(pl.col("x") // pl.col("y") > 3.0).select(pl.col("x").sum()).describe_fusion()
    assert not fusable and "floor-div" in why
    q = df.lazy().select(pl.col("a").sum(), pl.col("b"))
    fusable, sid, why, _ = q.describe_fusion()
    assert not fusable and "mixes" in why


def test_generic_program_is_compiled_for_unknown_shapes():
    df = pl.DataFrame([ph("a", pl.Int32), ph("x", pl.Float64, nullable=True)])
    q = df.lazy().filter((pl.col("a") > 5) & (pl.col("x") <= 2.5)).select((pl.col("x") * 2 + 1).sum(), pl.col("a").max(), pl.len())
    fusable, sid, why, dump = q.describe_fusion()
    assert fusable and sid == -1, (why, dump)
    assert "in_dtype=[3,10?,]" in dump


def test_type_coercion_inserts_casts():
    low = pl.plan.Lowering()
    schema = {"i": pl.Int64, "f": pl.Float64, "s": pl.Int16, "d": pl.Datetime}
    idx, dt = low.lower_expr(pl.col("i") + pl.col("f"), schema)
    assert dt == pl.Float64
    node = low.aexprs[idx]
    assert low.aexprs[node["lhs"]]["kind"] == F.AE_CAST and low.aexprs[node["lhs"]]["dtype"] == F.F64
    # python int literal takes the column's dtype (int literal vs Int64 column => Int64)
    idx, dt = low.lower_expr(pl.col("s") > 3, schema)
    lit = low.aexprs[low.aexprs[idx]["rhs"]]
    assert dt == pl.Boolean and lit["kind"] == F.AE_LITERAL and lit["dtype"] == F.I16
    # float literal vs int column: the column is cast to Float64
    idx, dt = low.lower_expr(pl.col("i") * 0.5, schema)
    assert dt == pl.Float64 and low.aexprs[low.aexprs[idx]["lhs"]]["kind"] == F.AE_CAST
    # true division of ints yields Float64; sum dtypes follow sum_output_dtype
    assert low.lower_expr(pl.col("i") / pl.col("i"), schema)[1] == pl.Float64
    assert low.lower_expr(pl.col("s").sum(), schema)[1] == pl.Int64
    assert low.lower_expr(pl.col("i").mean(), schema)[1] == pl.Float64
    assert low.lower_expr(pl.col("i").count(), schema)[1] == pl.UInt32
    # a date literal arrives as the physical i64 of Datetime[us]
    import datetime as dtm
    idx, dt = low.lower_expr(pl.col("d") <= dtm.datetime(1998, 9, 2), schema)
    lit = low.aexprs[low.aexprs[idx]["rhs"]]
    assert lit["dtype"] == F.I64 and lit["lit"] == 904694400000000
    with pytest.raises(KeyError):
        low.lower_expr(pl.col("nope"), schema)
    with pytest.raises(TypeError):
        low.lower_expr(pl.col("i") & pl.col("f"), schema)


def test_join_schema_and_suffix():
    L = pl.DataFrame([ph("k", pl.Int64), ph("rain", pl.Float64)])
    R = pl.DataFrame([ph("k", pl.Int64), ph("rain", pl.Float64), ph("z", pl.Int32)])
    low = pl.plan.Lowering()
    root, schema = low.lower_node(L.lazy().join(R.lazy(), on="k")._node)
    assert list(schema) == ["k", "rain", "rain_right", "z"]      # general.rs:17-49 _finish_join
    assert low.irs[root]["kind"] == F.IR_JOIN and low.irs[root]["suffix"] == "_right"


def test_sort_slice_and_semi_anti_lowering():
    """IR::Sort / IR::Slice / semi+anti joins reach the C ABI as PLX_IR_SORT / PLX_IR_SLICE / how (no GPU needed)."""
    df = pl.DataFrame([ph("a", pl.Int64), ph("b", pl.Float64, nullable=True)])
    lf = df.lazy().sort("a", pl.col("b") * 2, descending=[True, False], nulls_last=[False, True]).slice(-5, 3)
    low, root, schema = lf._lower()
    kinds = [n["kind"] for n in low.irs]
    assert kinds == [F.IR_SCAN, F.IR_SORT, F.IR_SLICE] and root == 2 and list(schema) == ["a", "b"]
    s = low.irs[1]
    assert s["sort_descending"] == [1, 0] and s["sort_nulls_last"] == [0, 1] and len(s["keys"]) == 2
    assert (low.irs[2]["slice_offset"], low.irs[2]["slice_len"]) == (-5, 3)
    (ir, n_ir, ae, n_ae, keep), *_ = lf._lowered_c()
    assert n_ir == 3 and ir[1].sort_descending[0] == 1 and ir[1].sort_nulls_last[1] == 1 and ir[2].slice_offset == -5 and ir[2].slice_len == 3
    # top_k(k, by) == sort(descending, nulls last).head(k); bottom_k: ascending
    low, _, _ = df.lazy().top_k(7, by=["a", "b"], reverse=[False, True])._lower()
    assert low.irs[1]["sort_descending"] == [1, 0] and low.irs[1]["sort_nulls_last"] == [1, 1] and low.irs[2]["slice_len"] == 7
    low, _, _ = df.lazy().bottom_k(2, by="a")._lower()
    assert low.irs[1]["sort_descending"] == [0] and low.irs[2]["slice_len"] == 2
    with pytest.raises(ValueError, match=r"the length of `descending` \(1\) does not match the length of `by` \(2\)"):
        df.lazy().sort("a", "b", descending=[True])
    with pytest.raises(ValueError, match=r"the length of `reverse` \(1\) does not match the length of `by` \(2\)"):
        df.lazy().top_k(1, by=["a", "b"], reverse=[True])
    with pytest.raises(ValueError, match="negative slice lengths"):
        df.lazy().slice(0, -1)
    other = pl.DataFrame([ph("a", pl.Int64), ph("z", pl.Int64)])
    for how, code in (("semi", F.JOIN_SEMI), ("anti", F.JOIN_ANTI)):
        low, root, schema = df.lazy().join(other.lazy(), on="a", how=how)._lower()
        assert low.irs[root]["how"] == code and list(schema) == ["a", "b"]      # left columns only


def test_drop_rename_and_collect_schema_are_projections():
    df = pl.DataFrame([ph("a", pl.Int64), ph("b", pl.Float64), ph("c", pl.Int16)])
    lf = df.lazy()
    assert lf.collect_schema() == {"a": pl.Int64, "b": pl.Float64, "c": pl.Int16}
    assert list(lf.drop("b").collect_schema()) == ["a", "c"] and list(lf.drop(["a", "b"]).collect_schema()) == ["c"]
    assert lf.rename({"a": "k", "c": "z"}).collect_schema() == {"k": pl.Int64, "b": pl.Float64, "z": pl.Int16}
    assert lf.drop("b")._node.kind == "select" and lf.rename({"a": "k"})._node.kind == "select"
    for bad in (lambda: lf.drop("nope"), lambda: lf.rename({"nope": "x"})):
        with pytest.raises(KeyError):
            bad()


def test_drop_nulls_is_a_filter_on_is_not_null():
    df = pl.DataFrame([ph("a", pl.Int64, nullable=True), ph("b", pl.Float64, nullable=True), ph("c", pl.Int16)])
    lf = df.lazy().drop_nulls(["a", "b"])
    assert lf._node.kind == "filter"
    low, root, schema = lf._lower()
    pred = low.aexprs[low.irs[root]["predicate"]]
    assert pred["kind"] == F.AE_BINARY and pred["op"] == F.OP_AND and low.aexprs[pred["lhs"]]["kind"] == F.AE_IS_NOT_NULL and low.aexprs[pred["rhs"]]["kind"] == F.AE_IS_NOT_NULL
    assert df.lazy().drop_nulls("a")._node.kind == "filter" and list(schema) == ["a", "b", "c"]
    fusable, _, why, _ = df.lazy().drop_nulls().select(pl.col("c").sum()).describe_fusion()
    assert fusable, why
