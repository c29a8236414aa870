"""polars_amd/polars_engine.py (the B3 attachment of SURVEY.md 8(b)) without a polars wheel: a stand-in NodeTraverser presents
this package's own lowered plans through the node classes of the reference (crates/polars-python/src/lazyframe/visitor/
nodes.rs, expr_nodes.rs: same class names, same attributes), the engine translates them back, and the re-lowered arenas must
equal the original ones.  Also: unsupported nodes leave the plan to the CPU engine (no set_udf), raise_on_fail raises."""
import ctypes as C
import enum

import pytest

import polars_amd as pl
from polars_amd import _ffi as F
from polars_amd import polars_engine as eng
from polars_amd import queries as Q


# ---- the reference's node classes (attribute names as exposed by #[pyo3(get)]) -------------------------------------------
class Operator(enum.Enum):      # visitor/expr_nodes.rs:56-80
    Eq = 0; EqValidity = 1; NotEq = 2; NotEqValidity = 3; Lt = 4; LtEq = 5; Gt = 6; GtEq = 7; Plus = 8; Minus = 9; Multiply = 10; Divide = 11
    TrueDivide = 12; FloorDivide = 13; Modulus = 14; And = 15; Or = 16; Xor = 17; LogicalAnd = 18; LogicalOr = 19


def _cls(name, *fields):
    def __init__(self, **kw):
        for f in fields:
            setattr(self, f, kw.get(f))
    return type(name, (), {"__init__": __init__})


Column, Literal, BinaryExpr, Cast, Agg, Len, Alias = (_cls("Column", "name"), _cls("Literal", "value", "dtype"), _cls("BinaryExpr", "left", "op", "right"),
                                                     _cls("Cast", "expr", "dtype", "options"), _cls("Agg", "name", "arguments", "options"), _cls("Len"), _cls("Alias", "expr", "name"))
PyExprIR = _cls("PyExprIR", "node", "output_name")
DataFrameScan, Filter, Select, HStack = _cls("DataFrameScan", "df", "projection", "selection"), _cls("Filter", "input", "predicate"), _cls("Select", "input", "expr", "should_broadcast"), _cls("HStack", "input", "exprs", "should_broadcast")
GroupBy, Join = _cls("GroupBy", "input", "keys", "aggs", "apply", "maintain_order", "options"), _cls("Join", "input_left", "input_right", "left_on", "right_on", "options")
Sort, Slice, Cache = _cls("Sort", "input", "by_column", "sort_options", "slice"), _cls("Slice", "input", "offset", "len"), _cls("Cache", "input", "id_")

OPS = {F.OP_EQ: Operator.Eq, F.OP_NE: Operator.NotEq, F.OP_LT: Operator.Lt, F.OP_LE: Operator.LtEq, F.OP_GT: Operator.Gt, F.OP_GE: Operator.GtEq, F.OP_PLUS: Operator.Plus,
       F.OP_MINUS: Operator.Minus, F.OP_MULTIPLY: Operator.Multiply, F.OP_TRUE_DIVIDE: Operator.TrueDivide, F.OP_FLOOR_DIVIDE: Operator.FloorDivide, F.OP_MODULUS: Operator.Modulus,
       F.OP_AND: Operator.And, F.OP_OR: Operator.Or, F.OP_XOR: Operator.Xor}
AGGS = {F.AGG_SUM: "sum", F.AGG_MEAN: "mean", F.AGG_MIN: "min", F.AGG_MAX: "max", F.AGG_COUNT: "count"}
PHYS = {F.BOOL: pl.Boolean, F.I8: pl.Int8, F.I16: pl.Int16, F.I32: pl.Int32, F.I64: pl.Int64, F.U8: pl.UInt8, F.U16: pl.UInt16, F.U32: pl.UInt32, F.U64: pl.UInt64,
        F.F32: pl.Float32, F.F64: pl.Float64}
HOW = {F.JOIN_INNER: "inner", F.JOIN_LEFT: "left", F.JOIN_SEMI: "semi", F.JOIN_ANTI: "anti"}


class FakeTraverser:
    """NodeTraverser (visit.rs:47-230) over the arenas of a polars_amd.plan.Lowering: what `collect(post_opt_callback=...)` would
    hand the callback for the same (already coerced, already optimized) plan."""

    def __init__(self, low, root):
        self.low, self.cur, self.udf = low, root, None

    def get_node(self): return self.cur
    def set_node(self, n): self.cur = n
    def set_udf(self, fn, is_pure): self.udf = (fn, is_pure)
    def get_schema(self): raise NotImplementedError

    def view_expression(self, i):
        d = self.low.aexprs[i]
        k = d["kind"]
        if k == F.AE_COLUMN: return Column(name=d["name"])
        if k == F.AE_LITERAL: return Literal(value=None if d["is_null"] else d["lit"], dtype=PHYS[d["dtype"]])
        if k == F.AE_BINARY: return BinaryExpr(left=d["lhs"], op=OPS[d["op"]], right=d["rhs"])
        if k == F.AE_CAST: return Cast(expr=d["lhs"], dtype=PHYS[d["dtype"]], options=0)
        if k == F.AE_AGG:
            if d["op"] == F.AGG_LEN: return Agg(name="count", arguments=[d["lhs"]], options=True)     # count(include_nulls = true)
            return Agg(name=AGGS[d["op"]], arguments=[d["lhs"]], options=False if d["op"] in (F.AGG_MIN, F.AGG_MAX, F.AGG_COUNT) else None)
        if k == F.AE_LEN: return Len()
        if k == F.AE_ALIAS: return Alias(expr=d["lhs"], name=d["name"])
        raise NotImplementedError(k)

    def _e(self, i):
        # ExprIR = node + output name: aliases live in the name, the node is the aliased expression (as in the reference's arenas)
        d, name = self.low.aexprs[i], None
        j = i
        while self.low.aexprs[j]["kind"] == F.AE_ALIAS:
            name = name or self.low.aexprs[j]["name"]; j = self.low.aexprs[j]["lhs"]
        if name is None:
            k = j
            while True:
                x = self.low.aexprs[k]
                if x["kind"] == F.AE_COLUMN: name = x["name"]; break
                if x["kind"] == F.AE_LEN: name = "len"; break
                if x["kind"] == F.AE_LITERAL: name = "literal"; break
                k = x["lhs"]
        _ = d
        return PyExprIR(node=j, output_name=name)

    def view_current_node(self):
        d = self.low.irs[self.cur]
        k = d["kind"]
        if k == F.IR_SCAN: return DataFrameScan(df=d["frame"], projection=None, selection=None)
        if k == F.IR_FILTER: return Filter(input=d["input"], predicate=self._e(d["predicate"]))
        if k == F.IR_SELECT: return Select(input=d["input"], expr=[self._e(e) for e in d["exprs"]], should_broadcast=True)
        if k == F.IR_HSTACK: return HStack(input=d["input"], exprs=[self._e(e) for e in d["exprs"]], should_broadcast=True)
        if k == F.IR_GROUPBY: return GroupBy(input=d["input"], keys=[self._e(e) for e in d["keys"]], aggs=[self._e(e) for e in d["exprs"]], apply=None, maintain_order=bool(d["maintain_order"]), options=None)
        if k == F.IR_JOIN:
            how = HOW[d["how"]]
            return Join(input_left=d["input"], input_right=d["input_right"], left_on=[self._e(e) for e in d["keys"]], right_on=[self._e(e) for e in d["keys_right"]],
                        options=(how, False, None, d["suffix"], True, "none"))
        if k == F.IR_SORT: return Sort(input=d["input"], by_column=[self._e(e) for e in d["keys"]], sort_options=(bool(d["maintain_order"]), [bool(x) for x in d["sort_nulls_last"]], [bool(x) for x in d["sort_descending"]]), slice=None)
        if k == F.IR_SLICE: return Slice(input=d["input"], offset=d["slice_offset"], len=d["slice_len"])
        raise NotImplementedError(k)


def ph(name, dtype, n=1 << 20, nullable=False, rng=None):
    h = C.c_uint64()
    F.check(F.lib().plx_column_placeholder(dtype.physical, n, int(nullable), 1 if rng else 0, rng[0] if rng else 0, rng[1] if rng else 0, C.byref(h)))
    return pl.Series._from_handle(name, h.value, dtype)


def frames():
    flag, status = pl.Categorical(["A", "N", "R"], pl.UInt8), pl.Categorical(["F", "O"], pl.UInt8)
    li = pl.DataFrame([ph("l_shipdate", pl.Datetime), ph("l_returnflag", flag, rng=(0, 2)), ph("l_linestatus", status, rng=(0, 1)), ph("l_quantity", pl.Int64),
                       ph("l_extendedprice", pl.Float64), ph("l_discount", pl.Float64), ph("l_tax", pl.Float64), ph("l_orderkey", pl.Int64)])
    orders = pl.DataFrame([ph("o_orderkey", pl.Int64, n=1 << 18), ph("o_custkey", pl.Int64, n=1 << 18), ph("o_orderdate", pl.Datetime, n=1 << 18), ph("o_shippriority", pl.Int64, n=1 << 18)])
    return li, orders


def canonical(low, root):
    """Arenas as nested tuples, independent of node numbering."""
    def ex(i):
        d = low.aexprs[i]
        if d["kind"] == F.AE_CAST and low.aexprs[d["lhs"]]["kind"] == F.AE_LITERAL and low.aexprs[d["lhs"]]["dtype"] == d["dtype"]:
            return ex(d["lhs"])     # a no-op cast: the stand-in traverser shows literals with their physical dtype (Int64 for a Datetime literal)
        return (d["kind"], d["op"], ex(d["lhs"]) if d["lhs"] >= 0 else None, ex(d["rhs"]) if d["rhs"] >= 0 else None, d["dtype"], d["is_null"],
                d["lit"] if d["kind"] == F.AE_LITERAL else None, d["name"])
    def ir(i):
        d = low.irs[i]
        return (d["kind"], ir(d["input"]) if d["input"] >= 0 else None, ir(d["input_right"]) if d["input_right"] >= 0 else None, ex(d["predicate"]) if d["predicate"] >= 0 else None,
                id(d["frame"]) if d["frame"] is not None else None, tuple(ex(e) for e in d["exprs"]), tuple(ex(e) for e in d["keys"]), tuple(ex(e) for e in d["keys_right"]), d["how"],
                d["maintain_order"], d["suffix"], tuple(d["sort_descending"]), tuple(d["sort_nulls_last"]), d["slice_offset"], d["slice_len"])
    return ir(root)


def roundtrip(lf):
    low, root, _ = lf._lower()
    nt = FakeTraverser(low, root)
    back = eng.Translator(nt, frame_of=lambda node: node.df).plan()
    low2, root2, _ = back._lower()
    return canonical(low, root), canonical(low2, root2)


@pytest.mark.parametrize("query", ["q1", "q1_sorted", "q3", "q3_top10", "cfg2", "semi", "with_columns_left_join"])
def test_translation_round_trips(query):
    li, orders = frames()
    c = pl.col
    lf = {"q1": lambda: Q.q1(li.lazy()), "q1_sorted": lambda: Q.q1_sorted(li.lazy()), "q3": lambda: Q.q3(li.lazy(), orders.lazy()),
          "q3_top10": lambda: Q.q3_top10(li.lazy(), orders.lazy()),
          "cfg2": lambda: li.lazy().filter(c("l_quantity") > 2 ** 20).select((c("l_extendedprice") * (1 - c("l_discount"))).sum().alias("xy"), c("l_tax").mean(), pl.len()),
          "semi": lambda: li.lazy().join(orders.lazy(), left_on="l_orderkey", right_on="o_orderkey", how="semi").group_by("l_returnflag").agg(c("l_quantity").count(), c("l_tax").min()),
          "with_columns_left_join": lambda: (li.lazy().with_columns((c("l_quantity") // 3).alias("q3"), (c("l_tax") / 4.0).alias("t4"))
                                             .join(orders.lazy(), left_on="l_orderkey", right_on="o_orderkey", how="left", suffix="_o").slice(-10, 5))}[query]()
    a, b = roundtrip(lf)
    assert a == b


def test_callback_commits_supported_plans_and_leaves_the_rest_to_the_cpu_engine():
    li, orders = frames()
    low, root, _ = Q.q1(li.lazy()).sort("l_returnflag")._lower()
    nt = FakeTraverser(low, root)
    eng.execute_with_amd(nt, None, frame_of=lambda node: node.df)
    assert nt.udf is not None and nt.udf[1] is True and nt.get_node() == root
    # an IR node outside the hot path: no udf, the traverser is back at the root
    class WithCache(FakeTraverser):
        def view_current_node(self):
            return Cache(input=0, id_=1) if self.cur == root else super().view_current_node()
    nt2 = WithCache(low, root)
    eng.execute_with_amd(nt2, None, frame_of=lambda node: node.df)
    assert nt2.udf is None and nt2.get_node() == root
    with pytest.raises(eng.NotSupported, match="Cache"):
        eng.execute_with_amd(WithCache(low, root), None, raise_on_fail=True, frame_of=lambda node: node.df)
    # unsupported expression / join options
    class Ternary: pass
    class WithTernary(FakeTraverser):
        def view_expression(self, i):
            return Ternary() if self.low.aexprs[i]["kind"] == F.AE_AGG else super().view_expression(i)
    nt3 = WithTernary(low, root)
    eng.execute_with_amd(nt3, None, frame_of=lambda node: node.df)
    assert nt3.udf is None
    jl, jr, _ = li.lazy().join(orders.lazy(), left_on="l_orderkey", right_on="o_orderkey")._lower()
    class FullJoin(FakeTraverser):
        def view_current_node(self):
            n = super().view_current_node()
            if type(n).__name__ == "Join":
                n.options = ("full", False, None, "_right", True, "none")
            return n
    nt4 = FullJoin(jl, jr)
    eng.execute_with_amd(nt4, None, frame_of=lambda node: node.df)
    assert nt4.udf is None


# ---- IR::Scan of a Parquet file -> this package's device scan --------------------------------------------------------------------------
Scan = _cls("Scan", "paths", "file_info", "hive_parts", "predicate", "file_options", "scan_type")
FileOptions = _cls("FileOptions", "n_rows", "with_columns", "cache", "row_index", "rechunk")


class MapTraverser:
    """A hand-built optimized plan: node id -> plan node, expression id -> expression node (same protocol as FakeTraverser)."""

    def __init__(self, nodes, exprs, root):
        self.nodes, self.exprs, self.cur, self.udf = nodes, exprs, root, None

    def get_node(self): return self.cur
    def set_node(self, n): self.cur = n
    def set_udf(self, fn, is_pure): self.udf = (fn, is_pure)
    def view_current_node(self): return self.nodes[self.cur]
    def view_expression(self, i): return self.exprs[i]


def test_file_scan_node_becomes_a_device_parquet_scan(tmp_path):
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq
    from polars_amd import io
    n = 10_000
    t = pa.table({"k": np.arange(n) % 5, "v": np.arange(n), "d": pa.array(np.arange(n).astype(np.int32), pa.date32()), "unused": np.zeros(n),
                  "nested": pa.array([[1]] * n)})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=1000, compression="zstd")
    # SELECT k, sum(v) FROM scan(path, columns=[k, v], predicate: v >= 7500 on an integer column) GROUP BY k   -- as the optimizer leaves it: projection and
    # predicate pushed into the scan node
    exprs = {0: Column(name="v"), 1: Literal(value=7500, dtype=pl.Int64), 2: BinaryExpr(left=0, op=Operator.GtEq, right=1), 3: Column(name="k"),
             4: Agg(name="sum", arguments=[0], options=None)}
    nodes = {0: Scan(paths=[path], file_info=None, hive_parts=None, predicate=PyExprIR(node=2, output_name="v"),
                     file_options=FileOptions(n_rows=None, with_columns=["k", "v"], cache=True, row_index=None, rechunk=False), scan_type=("parquet", "{}", "null")),
             1: GroupBy(input=0, keys=[PyExprIR(node=3, output_name="k")], aggs=[PyExprIR(node=4, output_name="v")], apply=None, maintain_order=False, options=None)}
    lf = eng.Translator(MapTraverser(nodes, exprs, 1)).plan()
    kinds = []
    node = lf._node
    while True:
        kinds.append(node.kind)
        if node.kind == "scan":
            break
        node = node.input
    assert kinds == ["group_by", "filter", "scan"]
    src = node.frame
    assert isinstance(src, io.ParquetFrame) and src.decoder == "device" and list(src.schema) == ["k", "v"]        # the nested column was projected away: no TypeError
    io.reset_scans(lf._node); io.push_down(lf._node)
    assert sorted(src.selected_columns()) == ["k", "v"] and src.selected_row_groups() == [7, 8, 9]              # statistics prune what the predicate excludes
    # what cannot be taken: files whose schemas differ, other formats, cloud options, a row index; a missing file is left to the CPU engine too
    other = str(tmp_path / "other.parquet")
    pq.write_table(pa.table({"k": np.arange(10).astype(np.float64)}), other)
    for change, word in (({"paths": [path, other]}, "columns differ"), ({"scan_type": ("csv", "{}", "null")}, "csv scan"), ({"paths": []}, "no files"), ({"scan_type": ("parquet", "{}", '{"aws": 1}')}, "cloud"),
                         ({"file_options": FileOptions(n_rows=None, with_columns=None, cache=True, row_index=("idx", 0), rechunk=False)}, "row index"),
                         ({"paths": [str(tmp_path / "absent.parquet")]}, "parquet file")):
        kw = dict(paths=[path], file_info=None, hive_parts=None, predicate=None, file_options=FileOptions(n_rows=None, with_columns=["k"], cache=True, row_index=None, rechunk=False),
                  scan_type=("parquet", "{}", "null"))
        kw.update(change)
        with pytest.raises(eng.NotSupported) as ei:
            eng.Translator(MapTraverser({0: Scan(**kw)}, {}, 0)).plan()
        assert word in str(ei.value)
    # all columns requested -> the nested one makes the schema unbuildable: TypeError, which execute_with_amd turns into "CPU engine runs it"
    nt = MapTraverser({0: Scan(paths=[path], file_info=None, hive_parts=None, predicate=None, file_options=FileOptions(n_rows=(0, 10), with_columns=None, cache=True, row_index=None, rechunk=False),
                               scan_type=("parquet", "{}", "null"))}, {}, 0)
    eng.execute_with_amd(nt)
    assert nt.udf is None
    # a slice pushed into the scan comes back as a Slice node
    kw["paths"] = [path]; kw["file_options"] = FileOptions(n_rows=(5, 10), with_columns=["k"], cache=True, row_index=None, rechunk=False)
    lf2 = eng.Translator(MapTraverser({0: Scan(**kw)}, {}, 0)).plan()
    assert lf2._node.kind == "slice" and lf2._node.input.kind == "scan"
    # pre_slice AND predicate on one scan: the reference slices first, then filters (multi_scan apply_extra_ops.rs :264 before :334)
    kw["predicate"] = PyExprIR(node=1, output_name="k")
    lf3 = eng.Translator(MapTraverser({0: Scan(**kw)}, {1: BinaryExpr(left=2, op=Operator.Gt, right=3), 2: Column(name="k"), 3: Literal(value=7, dtype=pl.Int64)}, 0)).plan()
    assert lf3._node.kind == "filter" and lf3._node.input.kind == "slice" and lf3._node.input.input.kind == "scan"


def test_file_scan_over_several_files_and_ipc(tmp_path):
    """The optimizer hands over the expanded path list: several local files with one schema are one device scan whose row groups are
    numbered across the files (statistics pruning picks row groups out of every file); scan_type "ipc" goes to the IPC reader."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.ipc as ipc
    import pyarrow.parquet as pq
    from polars_amd import io, ipc_io
    paths = []
    for f in range(3):
        n = 4000
        t = pa.table({"v": np.arange(n) + 10_000 * f, "s": pa.array(np.array(["a", "b", "c" + str(f)])[np.arange(n) % 3])})
        paths.append(str(tmp_path / f"part-{f}.parquet"))
        pq.write_table(t, paths[-1], row_group_size=1000)
    exprs = {0: Column(name="v"), 1: Literal(value=12_500, dtype=pl.Int64), 2: BinaryExpr(left=0, op=Operator.GtEq, right=1),
             3: Literal(value=21_000, dtype=pl.Int64), 4: BinaryExpr(left=0, op=Operator.Lt, right=3), 5: BinaryExpr(left=2, op=Operator.And, right=4)}
    fo = FileOptions(n_rows=None, with_columns=None, cache=True, row_index=None, rechunk=False)
    nodes = {0: Scan(paths=["file://" + paths[0]] + paths[1:], file_info=None, hive_parts=None, predicate=PyExprIR(node=5, output_name="v"), file_options=fo,
                     scan_type=("parquet", "{}", "null"))}
    lf = eng.Translator(MapTraverser(nodes, exprs, 0)).plan()
    src = lf._node.input.frame
    assert isinstance(src, io.ParquetFrame) and src.path == paths and src.num_row_groups == 12 and src.num_rows == 12_000
    io.reset_scans(lf._node); io.push_down(lf._node)
    assert src.selected_row_groups() == [6, 7, 8]                    # file 1: v in [12000, 14000); file 2: v in [20000, 21000)
    runs = []
    for i, part in enumerate(src._dec.parts):                        # the read goes file by file: record what each file is asked for (no GPU here)
        part.read = (lambda i: lambda rgs, cols: runs.append((i, list(rgs), list(cols))) or (f"frame{i}", 1000 * len(rgs), 8000 * len(rgs)))(i)
    joined = []
    real, io.concat_frames = io.concat_frames, lambda dfs: joined.append(list(dfs)) or "all"
    try:
        assert src._dec.read(src.selected_row_groups(), ["v"]) == ("all", 3000, 24000)
    finally:
        io.concat_frames = real
    assert runs == [(1, [2, 3], ["v"]), (2, [0], ["v"])] and joined == [["frame1", "frame2"]]
    # the same directory through the user-facing entry points: a directory, a glob, a list
    for source in (str(tmp_path), str(tmp_path / "part-*.parquet"), paths):
        assert io.ParquetFrame(source).path == paths
    with pytest.raises(FileNotFoundError):
        io.ParquetFrame(str(tmp_path / "nothing-*.parquet"))
    # Arrow IPC files
    ipaths = []
    for f in range(2):
        ipaths.append(str(tmp_path / f"b{f}.arrow"))
        with ipc.new_file(ipaths[-1], pa.schema([("v", pa.int64())])) as w:
            for lo in (0, 100):
                w.write_batch(pa.record_batch({"v": np.arange(lo, lo + 100)}))
    nodes = {0: Scan(paths=ipaths, file_info=None, hive_parts=None, predicate=None, file_options=fo, scan_type=("ipc", "{}", "null"))}
    lf = eng.Translator(MapTraverser(nodes, {}, 0)).plan()
    src = lf._node.frame
    assert isinstance(src, ipc_io.IpcFrame) and src.num_row_groups == 4 and src.num_rows == 400 and list(src.schema) == ["v"]


Union = _cls("Union", "inputs", "slice", "rows", "maintain_order")


def test_union_node_becomes_a_device_concat():
    """IR::Union (visitor/nodes.rs:361-370): the inputs are translated one by one, the node becomes a scan over a deferred source that
    collects them and concatenates on the device (io.ConcatFrame); a slice carried by the node stays a Slice; inputs whose schemas
    differ are left to the CPU engine (which has supertype rules for them)."""
    from polars_amd import io
    a = pl.DataFrame([ph("k", pl.Int64), ph("s", pl.Categorical(["x", "y"]), rng=(0, 1)), ph("t", pl.Datetime("ns"))])
    b = pl.DataFrame([ph("k", pl.Int64, n=1000), ph("s", pl.Categorical(["y", "z"]), rng=(0, 1), n=1000), ph("t", pl.Datetime("ns"), n=1000)])
    exprs = {0: Column(name="k"), 1: Literal(value=5, dtype=pl.Int64), 2: BinaryExpr(left=0, op=Operator.Gt, right=1)}
    nodes = {0: DataFrameScan(df=a, projection=None, selection=None), 1: DataFrameScan(df=b, projection=None, selection=None),
             2: Filter(input=1, predicate=PyExprIR(node=2, output_name="k")), 3: Union(inputs=[0, 2], slice=(3, 10), rows=(None, 0), maintain_order=True)}
    lf = eng.Translator(MapTraverser(nodes, exprs, 3), frame_of=lambda node: node.df).plan()
    assert lf._node.kind == "slice" and (lf._node.offset, lf._node.length) == (3, 10) and lf._node.input.kind == "scan"
    src = lf._node.input.frame
    assert isinstance(src, io.ConcatFrame) and list(src.schema) == ["k", "s", "t"] and src.schema["t"].time_unit == "ns" and len(src._lfs) == 2
    assert src._lfs[1]._node.kind == "filter"
    low, root, schema = lf._lower()                            # lowers without touching a GPU: the source is only a schema until collect()
    assert list(schema) == ["k", "s", "t"]
    # the user-facing spelling
    assert isinstance(pl.concat([a.lazy(), b.lazy().filter(pl.col("k") > 5)])._node.frame, io.ConcatFrame)
    # schemas that differ: names, dtypes, time units
    for other in (pl.DataFrame([ph("k", pl.Int64), ph("s2", pl.Categorical(["x"]))]), pl.DataFrame([ph("k", pl.Int32), ph("s", pl.Categorical(["x"])), ph("t", pl.Datetime("ns"))]),
                  pl.DataFrame([ph("k", pl.Int64), ph("s", pl.Categorical(["x"])), ph("t", pl.Datetime("us"))])):
        bad = {0: DataFrameScan(df=a, projection=None, selection=None), 1: DataFrameScan(df=other, projection=None, selection=None),
               2: Union(inputs=[0, 1], slice=None, rows=(None, 0), maintain_order=True)}
        with pytest.raises(eng.NotSupported) as ei:
            eng.Translator(MapTraverser(bad, {}, 2), frame_of=lambda node: node.df).plan()
        assert "union" in str(ei.value)
    with pytest.raises(NotImplementedError):
        pl.concat([a, b], how="horizontal")
