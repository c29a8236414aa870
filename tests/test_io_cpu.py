"""polars_amd/io.py without a GPU: what a Parquet scan decides to read.  Projection pushdown (only the columns the plan touches),
row-group skipping from min / max statistics for the simple conjuncts above the scan, merging of several uses of one scan."""
import datetime as dt

import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import polars_amd as pl
from polars_amd import datagen, io
from polars_amd import queries as Q


@pytest.fixture(scope="module")
def lineitem_file(tmp_path_factory):
    """A 16-column lineitem (the PDS-H schema of examples/datasets/pds_heads/lineitem.feather), sorted by l_shipdate, 40 row groups."""
    n = 40_000
    li = datagen.lineitem_host(n, seed=6)
    order = np.argsort(li["l_shipdate"], kind="stable")
    rng = np.random.default_rng(1)
    t = pa.table({
        "l_orderkey": pa.array(rng.integers(1, 10_000, n)), "l_partkey": pa.array(rng.integers(1, 1000, n)), "l_suppkey": pa.array(rng.integers(1, 100, n)),
        "l_linenumber": pa.array(rng.integers(1, 8, n)), "l_quantity": pa.array(li["l_quantity"][order]), "l_extendedprice": pa.array(li["l_extendedprice"][order]),
        "l_discount": pa.array(li["l_discount"][order]), "l_tax": pa.array(li["l_tax"][order]),
        "l_returnflag": pa.array([datagen.FLAGS[c] for c in li["l_returnflag"][order]], pa.large_string()),
        "l_linestatus": pa.array([datagen.STATUS[c] for c in li["l_linestatus"][order]], pa.large_string()),
        "l_shipdate": pa.array(li["l_shipdate"][order], pa.timestamp("us")), "l_commitdate": pa.array(li["l_shipdate"][order], pa.timestamp("us")),
        "l_receiptdate": pa.array(li["l_shipdate"][order], pa.timestamp("us")), "l_shipinstruct": pa.array(["NONE"] * n, pa.large_string()),
        "l_shipmode": pa.array(["AIR"] * n, pa.large_string()), "l_comment": pa.array(["x"] * n, pa.large_string())})
    path = str(tmp_path_factory.mktemp("pq") / "lineitem.parquet")
    pq.write_table(t, path, row_group_size=1000)
    return path, li, order


def test_schema_and_laziness(lineitem_file):
    path, _, _ = lineitem_file
    lf = pl.scan_parquet(path)
    src = lf._node.frame
    assert isinstance(src, io.ParquetFrame) and src.num_rows == 40_000 and len(src.schema) == 16
    assert src.schema["l_shipdate"] == pl.Datetime and src.schema["l_quantity"] == pl.Int64 and src.schema["l_extendedprice"] == pl.Float64
    assert src.schema["l_returnflag"].physical == pl.UInt32.physical and src._df is None       # strings arrive as dictionary codes; nothing read yet


def test_q1_reads_seven_of_sixteen_columns(lineitem_file):
    path, _, _ = lineitem_file
    lf = Q.q1(pl.scan_parquet(path))
    io.reset_scans(lf._node); io.push_down(lf._node)
    src = lf._node.input.input.frame if lf._node.kind != "scan" else None
    node = lf._node
    while node.kind != "scan":
        node = node.input
    src = node.frame
    assert sorted(src.selected_columns()) == sorted(datagen.LINEITEM_Q1_COLS)
    # Q1's predicate keeps ~98 % of a table sorted by ship date: only the last row groups can be skipped
    rgs = src.selected_row_groups()
    assert 35 <= len(rgs) < 40 and rgs == list(range(len(rgs)))
    assert src._preds == [("l_shipdate", pl._ffi.OP_LE, Q.Q1_CUTOFF)]


def test_row_group_skipping_follows_the_statistics(lineitem_file):
    path, li, order = lineitem_file
    ship = li["l_shipdate"][order]
    lo, hi = dt.datetime(1994, 1, 1), dt.datetime(1994, 3, 1)
    c = pl.col
    lf = pl.scan_parquet(path).filter((c("l_shipdate") >= lo) & (c("l_shipdate") < hi) & (c("l_quantity") > 0)).select(c("l_extendedprice").sum())
    io.reset_scans(lf._node); io.push_down(lf._node)
    node = lf._node
    while node.kind != "scan":
        node = node.input
    src = node.frame
    assert sorted(src.selected_columns()) == ["l_extendedprice", "l_quantity", "l_shipdate"]
    rgs = src.selected_row_groups()
    us = lambda d: int((d - dt.datetime(1970, 1, 1)).total_seconds()) * 1_000_000
    rows = np.nonzero((ship >= us(lo)) & (ship < us(hi)))[0]
    want = sorted(set((rows // 1000).tolist()))
    assert set(want) <= set(rgs) and len(rgs) <= len(want) + 2 and len(rgs) < 6        # every matching group is read, at most the two boundary groups extra
    # literal on the left, != and ==, and a predicate that nothing can satisfy
    lf2 = pl.scan_parquet(path).filter((dt.datetime(2030, 1, 1) < c("l_shipdate"))).select(pl.len())
    io.reset_scans(lf2._node); io.push_down(lf2._node)
    n2 = lf2._node
    while n2.kind != "scan":
        n2 = n2.input
    assert n2.frame.selected_row_groups() == [] and n2.frame.selected_columns() == ["l_shipdate"]


def test_plans_without_pruning_opportunities(lineitem_file):
    path, _, _ = lineitem_file
    c = pl.col
    # no select: every column is needed; an OR predicate is not a conjunct of simple comparisons
    lf = pl.scan_parquet(path).filter((c("l_quantity") > 49) | (c("l_tax") > 0.07))
    io.reset_scans(lf._node); io.push_down(lf._node)
    src = lf._node.input.frame
    assert len(src.selected_columns()) == 16 and len(src.selected_row_groups()) == 40
    # the same file scanned twice in one plan with different needs: union of columns, no predicate pushdown
    a = pl.scan_parquet(path)
    shared = a._node.frame
    left = a.filter(c("l_quantity") > 10).select("l_orderkey", "l_quantity")
    right = pl.LazyFrame(a._node).filter(c("l_tax") > 0.01).select("l_orderkey", "l_tax")
    j = left.join(right, on="l_orderkey").group_by("l_orderkey").agg(c("l_quantity").sum(), c("l_tax").max())
    io.reset_scans(j._node); io.push_down(j._node)
    assert sorted(shared.selected_columns()) == ["l_orderkey", "l_quantity", "l_tax"] and len(shared.selected_row_groups()) == 40 and shared._preds == []
    # join: each side only reads its own columns
    orders = pa.table({"o_orderkey": pa.array(np.arange(100)), "o_custkey": pa.array(np.arange(100)), "o_orderdate": pa.array(np.arange(100), pa.timestamp("us")),
                       "o_shippriority": pa.array(np.zeros(100, np.int64)), "o_comment": pa.array(["c"] * 100, pa.large_string())})
    opath = path.replace("lineitem", "orders")
    pq.write_table(orders, opath)
    q3 = Q.q3(pl.scan_parquet(path), pl.scan_parquet(opath))
    io.reset_scans(q3._node); io.push_down(q3._node)
    jn = q3._node.input
    lsrc, rsrc = jn.left, jn.right
    while lsrc.kind != "scan":
        lsrc = lsrc.input
    while rsrc.kind != "scan":
        rsrc = rsrc.input
    assert sorted(lsrc.frame.selected_columns()) == sorted(datagen.LINEITEM_Q3_COLS)
    assert sorted(rsrc.frame.selected_columns()) == sorted(datagen.ORDERS_Q3_COLS)


def test_benchmark_tables_follow_the_reference_sample_schemas(tmp_path):
    """The synthetic TPC-H columns (polars_amd/datagen.py) and the scan's dtype mapping agree with the Arrow schemas of the
    reference's own sample tables (examples/datasets/pds_heads/*.feather, recorded by tests/golden/make_pds_schema.py)."""
    import json
    import os
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pds_heads_schema.json")))
    arrow = {"int64": pa.int64(), "double": pa.float64(), "large_string": pa.large_string(), "timestamp[us]": pa.timestamp("us")}
    lt = datagen.logical_dtypes(pl)
    for table, needed in (("lineitem", set(datagen.LINEITEM_Q1_COLS) | set(datagen.LINEITEM_Q3_COLS)), ("orders", set(datagen.ORDERS_Q3_COLS))):
        cols = dict(ref[table]["columns"])
        assert needed <= set(cols), (table, needed - set(cols))
        # a file with the reference's schema maps to the dtypes the generators use
        t = pa.table({n: pa.array([], arrow[ty]) for n, ty in cols.items()})
        path = str(tmp_path / f"{table}.parquet")
        pq.write_table(t, path)
        schema = pl.scan_parquet(path)._node.frame.schema
        for n in needed:
            if cols[n] == "large_string":
                assert schema[n].physical == pl.UInt32.physical and lt[n].name == "Categorical"      # strings = dictionary codes on the device
            else:
                want = {"int64": pl.Int64, "double": pl.Float64, "timestamp[us]": pl.Datetime}[cols[n]]
                assert schema[n] == want and lt.get(n, want) == want, (table, n)
    li = datagen.lineitem_host(10, seed=1)
    assert li["l_quantity"].dtype == np.int64 and li["l_extendedprice"].dtype == np.float64 and li["l_shipdate"].dtype == np.int64


def test_float_statistics_never_prune_a_group_that_may_hold_nan(tmp_path):
    """Parquet min / max ignore NaN, the engine's float comparisons are a total order with NaN greatest
    (comparisons/simd.rs:171-275): `x > 5` must keep a row group [1.0, NaN, 2.0] whose statistics say max = 2."""
    c = pl.col
    t = pa.table({"x": pa.array([1.0, float("nan"), 2.0, 10.0, 11.0, 12.0]), "k": pa.array([1, 2, 3, 4, 5, 6])})
    path = str(tmp_path / "nan.parquet")
    pq.write_table(t, path, row_group_size=3)

    def groups(pred):
        lf = pl.scan_parquet(path).filter(pred).select(c("k").sum())
        io.reset_scans(lf._node); io.push_down(lf._node)
        n = lf._node
        while n.kind != "scan":
            n = n.input
        return n.frame.selected_row_groups()
    assert groups(c("x") > 5.0) == [0, 1]            # the NaN row of group 0 is > 5 in total order
    assert groups(c("x") >= 5.0) == [0, 1]
    assert groups(c("x") != 1.5) == [0, 1]
    assert groups(c("x") == float("nan")) == [0, 1]
    assert groups(c("x") < 5.0) == [0]               # NaN is never below a number: pruning from min stays valid
    assert groups(c("x") <= 0.5) == []
    assert groups(c("k") > 3) == [1]                 # integer statistics prune as before


def test_a_plan_that_reads_no_column_keeps_one_for_the_row_count(lineitem_file):
    path, _, _ = lineitem_file
    lf = pl.scan_parquet(path).select(pl.len())
    io.reset_scans(lf._node); io.push_down(lf._node)
    src = lf._node.input.frame
    cols = src.selected_columns()
    assert len(cols) == 1 and src.schema[cols[0]].np_dtype.itemsize <= 8      # COUNT(*) must see num_rows rows, not an empty frame


def test_split_by_rows_and_dictionary_union():
    """The two pure pieces under a scan that several processes share: contiguous, balanced row-group runs and one dictionary for all."""
    from polars_amd import io
    rng = np.random.default_rng(3)
    for _ in range(200):
        rows = rng.integers(0, 1000, int(rng.integers(0, 40))).tolist()
        parts = int(rng.integers(1, 10))
        runs = io.split_by_rows(rows, parts)
        assert len(runs) == parts and [i for r in runs for i in r] == list(range(len(rows)))          # contiguous, in order, complete
        if rows and sum(rows):
            loads = [sum(rows[i] for i in r) for r in runs]
            assert max(loads) <= sum(rows) / parts + max(rows)                                         # never more than one row group over the fair share
    assert io.split_by_rows([10] * 8, 4) == [[0, 1], [2, 3], [4, 5], [6, 7]] and io.split_by_rows([1, 1, 1, 1, 1, 100], 3) == [[0, 1, 2, 3, 4], [5], []]
    union, remaps = io.dictionary_union([["a", "b"], ["b", "c", "a", "b"], [], ["d"]])
    assert union == ["a", "b", "c", "d"] and [r.tolist() for r in remaps] == [[0, 1], [1, 2, 0, 1], [], [3]] and all(r.dtype == np.uint32 for r in remaps)
    with pytest.raises(ValueError):
        io.ParquetFrame.__new__(io.ParquetFrame)._set_shard((2, 2))


def test_logical_arrow_types():
    """frame.logical_arrow: the host half of Series.to_arrow() -- physical export + mirror dtype -> Arrow logical type."""
    import pyarrow as pa
    from polars_amd.frame import logical_arrow
    from polars_amd.io import string_column_dtype
    assert logical_arrow(pa.array([1, None], pa.int32()), pl.Date).type == pa.date32()
    a = logical_arrow(pa.array([1_000, None], pa.int64()), pl.Datetime("ns", "UTC"))
    assert a.type == pa.timestamp("ns", "UTC") and a.null_count == 1
    assert logical_arrow(pa.array([5], pa.int64()), pl.Datetime).type == pa.timestamp("us")
    codes = pa.array([1, 0, None, 1], pa.uint32())
    d = logical_arrow(codes, pl.Categorical(["x", "y"]))
    assert pa.types.is_dictionary(d.type) and d.to_pylist() == ["y", "x", None, "y"]
    s = logical_arrow(codes, string_column_dtype(["x", "y"]))
    assert s.type == pa.large_string() and s.to_pylist() == ["y", "x", None, "y"]
    b = logical_arrow(codes, string_column_dtype([b"\x00", b"\xff\x01"]))
    assert b.type == pa.large_binary() and b.to_pylist() == [b"\xff\x01", b"\x00", None, b"\xff\x01"]
    assert logical_arrow(pa.array([1.5]), pl.Float64).type == pa.float64()
    assert pl.Datetime("ms") == pl.Datetime and pl.Datetime("ms").time_unit == "ms" and repr(pl.Datetime) == "Datetime"
    with pytest.raises(ValueError):
        pl.Datetime("s")


def test_slice_pushdown_reads_only_the_row_groups_it_overlaps(tmp_path):
    """head / slice directly above a scan (also through row-preserving projections): only the overlapping row groups are read and
    the Slice node's offset is rebased to the first of them (slice_pushdown_lp.rs); a filter in between, an aggregate in between or
    a second use of the scan with other rows switch it off.  Checked by replaying the plan's arithmetic on row indices."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from polars_amd import _ffi as F
    n = 10_000
    path = str(tmp_path / "t.parquet")
    pq.write_table(pa.table({"k": np.arange(n), "v": np.arange(n) * 2}), path, row_group_size=1000)
    c = pl.col

    def plan_of(lf):
        low, root, _ = lf._lower()
        node = lf._node
        while node.kind != "scan":
            node = node.input
        ir = next(d for d in low.irs if d["kind"] == F.IR_SLICE)
        return node.frame, ir["slice_offset"], ir["slice_len"]

    def rows_read(src):
        return np.concatenate([np.arange(g * 1000, (g + 1) * 1000) for g in src.selected_row_groups()] or [np.arange(0)])

    for off, ln in [(0, 5), (0, 1000), (999, 2), (1000, 1), (2500, 4000), (9990, 100), (10_000, 5), (20_000, 5), (3000, 0), (-5, 5), (-1500, 10), (-20_000, 3)]:
        for build in (lambda: pl.scan_parquet(path).slice(off, ln), lambda: pl.scan_parquet(path).select(c("k"), (c("v") + 1).alias("w")).slice(off, ln),
                      lambda: pl.scan_parquet(path).with_columns((c("v") * 2).alias("w")).slice(off, ln)):
            src, o2, l2 = plan_of(build())
            got = rows_read(src)
            m = len(got)
            start = max(0, m + o2) if o2 < 0 else min(o2, m)
            want = np.arange(n)[off:off + ln] if off >= 0 else np.arange(n)[max(0, n + off):][:ln]
            assert np.array_equal(got[start:start + l2], want), (off, ln)
            assert len(src.selected_row_groups()) <= (ln + 999) // 1000 + 1 or off < 0, (off, ln)
    assert len(plan_of(pl.scan_parquet(path).head(5))[0].selected_row_groups()) == 1
    # not pushed: a filter between slice and scan (the slice counts FILTERED rows), an aggregate, a sort
    for lf in (pl.scan_parquet(path).filter(c("k") >= 0).slice(2500, 10), pl.scan_parquet(path).sort("k").slice(2500, 10)):
        src, o2, _ = plan_of(lf)
        assert len(src.selected_row_groups()) == 10 and o2 == 2500
    # a filter ABOVE the slice does not matter
    src, o2, _ = plan_of(pl.scan_parquet(path).slice(2500, 10).filter(c("k") > 0))
    assert src.selected_row_groups() == [2] and o2 == 500
    # two uses of one scan with different windows: all rows, offsets untouched
    base = pl.scan_parquet(path)
    j = base.slice(2500, 10).join(base.slice(7000, 10), on="k")
    low, _, _ = j._lower()
    assert sorted(d["slice_offset"] for d in low.irs if d["kind"] == F.IR_SLICE) == [2500, 7000]
    assert len(base._node.frame.selected_row_groups()) == 10


def test_explain_reports_what_each_scan_reads(tmp_path):
    import pyarrow as pa
    import pyarrow.parquet as pq
    for i in range(3):
        pq.write_table(pa.table({"k": np.arange(5000) + 5000 * i, "v": np.arange(5000) * 1.0, "s": ["a"] * 5000}), str(tmp_path / f"p{i}.parquet"), row_group_size=1000)
    c = pl.col
    text = pl.scan_parquet(str(tmp_path), shard=(1, 2)).filter(c("k") >= 7000).select(c("v").sum()).explain()
    assert "Parquet SCAN [" in text and "PROJECT 2/3 COLUMNS: k, v" in text and "ROW GROUPS 4/15" in text and "[k >= 7000]" in text and "SHARD: 1 of 2" in text
    text = pl.scan_parquet(str(tmp_path / "p0.parquet")).select("k").slice(1500, 10).explain()
    assert "PROJECT 1/3 COLUMNS: k" in text and "ROW GROUPS 1/5" in text and "SLICE: offset 1500, length 10" in text


def test_hive_partitioned_directory(tmp_path):
    """key=value directories under a scanned directory become columns (after the file's own; integers when every value parses as
    one), predicates on them skip whole files -- crates/polars-io/src/hive.rs; the reference turns this on by default for a directory
    source and for nothing else (a list of the same files has no partition columns)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from polars_amd import io
    files = []
    for y in (1994, 1995, 1996):
        for seg in ("A", "B"):
            os.makedirs(tmp_path / f"year={y}" / f"seg={seg}")
            files.append(str(tmp_path / f"year={y}" / f"seg={seg}" / "part-0.parquet"))
            pq.write_table(pa.table({"k": np.arange(3000), "v": np.arange(3000) * 1.0}), files[-1], row_group_size=1000)
    src = io.ParquetFrame(str(tmp_path))
    assert list(src.schema) == ["k", "v", "year", "seg"] and src.schema["year"] == pl.Int64 and src.schema["seg"].from_strings
    c = pl.col
    lf = pl.scan_parquet(str(tmp_path)).filter((c("year") >= 1995) & (c("year") < 1996)).select(c("v").sum())
    text = lf.explain()
    assert "ROW GROUPS 6/18" in text and "PROJECT 2/4 COLUMNS: v, year" in text
    node = lf._node
    while node.kind != "scan":
        node = node.input
    assert node.frame.selected_row_groups() == list(range(6, 12))
    lf = pl.scan_parquet(str(tmp_path)).filter((c("seg") == "B") & (c("year") != 1994)).select(c("v").sum())
    assert "ROW GROUPS 6/18" in lf.explain()                                      # string partition values prune on == / != as well
    assert list(io.ParquetFrame(files).schema) == ["k", "v"]                       # an explicit list: no partition columns
    with pytest.raises(ValueError):                                               # a key that is also a column of the files
        os.makedirs(tmp_path / "bad" / "k=1")
        pq.write_table(pa.table({"k": np.arange(3)}), str(tmp_path / "bad" / "k=1" / "f.parquet"))
        io.ParquetFrame(str(tmp_path / "bad"))


def test_scan_parquet_keyword_arguments(tmp_path):
    """n_rows (a pushed-down slice), hive_partitioning on / off, use_statistics, include_file_paths -- polars.scan_parquet's keywords."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    for y in (1, 2):
        os.makedirs(tmp_path / f"y={y}")
        pq.write_table(pa.table({"k": np.arange(4000) + 4000 * (y - 1)}), str(tmp_path / f"y={y}" / "f.parquet"), row_group_size=1000)
    files = sorted(str(p) for p in tmp_path.glob("y=*/f.parquet"))
    c = pl.col
    lf = pl.scan_parquet(str(tmp_path), n_rows=1500)
    assert lf._node.kind == "slice" and "ROW GROUPS 2/8" in lf.explain()
    assert list(pl.scan_parquet(str(tmp_path), hive_partitioning=False).collect_schema()) == ["k"]
    assert list(pl.scan_parquet(files, hive_partitioning=True).collect_schema()) == ["k", "y"]
    sc = pl.scan_parquet(files, include_file_paths="path").collect_schema()
    assert list(sc) == ["k", "path"] and sc["path"].from_strings
    assert "ROW GROUPS 4/8" in pl.scan_parquet(files, include_file_paths="path").filter(c("path") == files[1]).explain()
    q = lambda **kw: pl.scan_parquet(str(tmp_path), **kw).filter(c("k") >= 7000).select(c("k").sum()).explain()
    assert "ROW GROUPS 1/8" in q() and "ROW GROUPS 8/8" in q(use_statistics=False)


def test_ns_datetime_statistics_do_not_prune_against_a_plain_integer(tmp_path):
    """Round-2 advisor (medium): stats() of a Datetime[ns] column are in microseconds while a plain integer literal is compared in ns ticks
    by the kernel -- such a predicate must not prune (499 rows match in the last group here); a datetime literal still does."""
    import datetime as dt
    import pyarrow as pa
    import pyarrow.parquet as pq
    n = 4000
    ts = np.arange(n, dtype=np.int64) * 1_000_000_007 + 1_600_000_000_000_000_000
    path = str(tmp_path / "ns.parquet")
    pq.write_table(pa.table({"ts": pa.array(ts, pa.timestamp("ns")), "v": np.arange(n)}), path, row_group_size=1000, version="2.6")
    lf = pl.scan_parquet(path).filter(pl.col("ts") > int(ts[3500])).select(pl.len())
    assert "ROW GROUPS 4/4" in lf.explain()
    when = dt.datetime(1970, 1, 1) + dt.timedelta(microseconds=int(ts[3500] // 1000))
    assert "ROW GROUPS 1/4" in pl.scan_parquet(path).filter(pl.col("ts") > when).select(pl.len()).explain()


def test_hive_values_are_url_decoded_and_the_default_partition_is_null(tmp_path):
    """crates/polars-io/src/hive.rs: percent-decoding, __HIVE_DEFAULT_PARTITION__ -> null; integers only for plain decimals."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from polars_amd import io
    for d in ("a=1/b=x%20y", "a=__HIVE_DEFAULT_PARTITION__/b=z", "a=3/b=__HIVE_DEFAULT_PARTITION__"):
        os.makedirs(tmp_path / "t" / d)
        pq.write_table(pa.table({"k": np.arange(10)}), str(tmp_path / "t" / d / "f.parquet"))
    src = io.ParquetFrame(str(tmp_path / "t"))
    assert src.schema["a"] == pl.Int64 and src.schema["b"].from_strings
    assert sorted(src._dec._hive["a"], key=str) == [1, 3, None] and sorted(src._dec._hive["b"], key=str) == [None, "x y", "z"]
    lf = pl.scan_parquet(str(tmp_path / "t")).filter(pl.col("b") == "x y").select(pl.len())
    assert "ROW GROUPS 2/3" in lf.explain()            # the null partition value is never pruned through statistics; the filter drops it on the device
    for d in ("c=1_000", "c=5"):
        os.makedirs(tmp_path / "u" / d)
        pq.write_table(pa.table({"k": np.arange(3)}), str(tmp_path / "u" / d / "f.parquet"))
    assert io.ParquetFrame(str(tmp_path / "u")).schema["c"].from_strings      # '1_000' is not an integer
