"""Attachment to an installed `polars` wheel through the optimized-IR seam -- the way the
cuDF engine plugs in, no Rust needed:

    import polars, polars_amd.polars_engine as amd
    df = lazy_frame.collect(post_opt_callback=amd.execute_with_amd)          # or amd.collect(lazy_frame)

`LazyFrame.collect(post_opt_callback=cb)` hands `cb(nt, duration_since_start)` a `NodeTraverser` over the OPTIMIZED
plan (crates/polars-python/src/lazyframe/visit.rs:47-230; py-polars lazyframe/engine.py:377-434 and :946-970 for the
GPU engine).  The callback walks the IR (`view_current_node` / `get_inputs` / `view_expression`: node classes in
crates/polars-python/src/lazyframe/visitor/nodes.rs and expr_nodes.rs), rebuilds the query on the mirror API of this
package -- which lowers to the plx_ir / plx_aexpr arenas of include/polars_amd.h -- and, if every node is on the GPU hot
path, swaps the plan for a Python scan with `nt.set_udf(fn, True)` (visit.rs:112-129).  PythonScanExec then calls
`fn(with_columns, predicate, n_rows, should_time)` and expects a polars DataFrame, or (DataFrame, timings) when
should_time is set (crates/polars-mem-engine/src/executors/scan/python_scan.rs:97-117).  Anything unsupported =>
the callback returns without `set_udf` and the CPU engine runs the plan (the contract of
docs/source/user-guide/gpu-support.md), unless raise_on_fail is set.

No `polars` wheel exists in the build image, so this module is exercised by tests/test_polars_engine_cpu.py against a
stand-in traverser that presents this package's own lowered plans through the reference's node classes.
"""
from __future__ import annotations

import datetime as _dt
import time
from functools import partial
from typing import Any, Callable, Dict, List, Optional

from . import _ffi as F
from . import datatypes as T
from . import expr as E
from . import plan as P
from .frame import DataFrame, LazyFrame, Series


class NotSupported(Exception):
    """The plan leaves the GPU hot path: the CPU engine must run it."""


# Operator enum of visitor/expr_nodes.rs:56-80 (by name) -> plx_operator
_OPS = {"Eq": F.OP_EQ, "NotEq": F.OP_NE, "Lt": F.OP_LT, "LtEq": F.OP_LE, "Gt": F.OP_GT, "GtEq": F.OP_GE, "Plus": F.OP_PLUS, "Minus": F.OP_MINUS,
        "Multiply": F.OP_MULTIPLY, "TrueDivide": F.OP_TRUE_DIVIDE, "FloorDivide": F.OP_FLOOR_DIVIDE, "Modulus": F.OP_MODULUS, "And": F.OP_AND, "Or": F.OP_OR,
        "Xor": F.OP_XOR, "LogicalAnd": F.OP_AND, "LogicalOr": F.OP_OR}
_AGGS = {"sum": F.AGG_SUM, "mean": F.AGG_MEAN, "min": F.AGG_MIN, "max": F.AGG_MAX, "count": F.AGG_COUNT}
_DTYPES = {"Boolean": T.Boolean, "Int8": T.Int8, "Int16": T.Int16, "Int32": T.Int32, "Int64": T.Int64, "UInt8": T.UInt8, "UInt16": T.UInt16, "UInt32": T.UInt32,
           "UInt64": T.UInt64, "Float32": T.Float32, "Float64": T.Float64, "Date": T.Date, "Datetime": T.Datetime}


def _dtype(dt: Any) -> Optional[T.DataType]:
    """polars DataType (class or instance) -> mirror dtype; None for the dyn-literal placeholder types."""
    if isinstance(dt, T.DataType):
        return dt
    name = getattr(dt, "__name__", None) or type(dt).__name__
    text = str(dt)
    if name.startswith("Unknown") or text.startswith("Unknown"):
        return None
    for key, val in _DTYPES.items():
        if name == key or text == key or text.startswith(key + "("):
            if key == "Datetime" and "time_unit" in text:
                import re
                unit = re.search(r"time_unit='(\w+)'", text)
                zone = re.search(r"time_zone='([^']+)'", text)
                if unit is None or unit.group(1) not in T.DatetimeType.UNITS:
                    raise NotSupported(f"Datetime time unit: {text}")
                if unit.group(1) != "us" or zone is not None:
                    return T.Datetime(unit.group(1), zone.group(1) if zone else None)      # casts between units are refused when the plan is lowered (plan.py)
            return val
    raise NotSupported(f"dtype {text}")


def _op_name(op: Any) -> str:
    return getattr(op, "name", None) or str(op).split(".")[-1]


class Translator:
    """NodeTraverser -> mirror LazyFrame.  `frame_of(node)` turns a DataFrameScan node into a device-resident
    polars_amd.DataFrame (default: import the polars frame through Arrow)."""

    def __init__(self, nt: Any, frame_of: Optional[Callable[[Any], DataFrame]] = None):
        self.nt = nt
        self.frame_of = frame_of or self._import_polars_frame

    @staticmethod
    def _import_polars_frame(node: Any) -> DataFrame:
        import polars
        df = polars.DataFrame._from_pydf(node.df)          # visitor/nodes.rs:219-226: the scan carries the PyDataFrame itself
        if node.projection is not None:
            df = df.select(list(node.projection))
        tbl = df.to_arrow()
        return DataFrame([Series.from_arrow(name, tbl.column(name)) for name in tbl.column_names])

    def _file_scan(self, node: Any) -> LazyFrame:
        """IR::Scan (visitor/nodes.rs:199-215, filled at :451-499): paths, predicate, file_options (UnifiedScanArgs: with_columns = the
        optimizer's projection pushdown, n_rows = slice pushdown, row_index), scan_type = ("parquet", options json, cloud options json).
        Local Parquet / Arrow IPC files (one or several with the same schema) become this package's device scan (io.scan_parquet /
        ipc_io.scan_ipc: metadata parsed by the library, pages decoded on the GPU, files concatenated on the device); the pushed-down predicate is re-applied as a filter, whose simple conjuncts prune row groups by statistics."""
        from .io import scan_parquet
        from .ipc_io import scan_ipc
        paths = [str(p) for p in (node.paths or [])]
        if not paths:
            raise NotSupported("scan over no files")
        if any("://" in p and not p.startswith("file://") for p in paths):
            raise NotSupported("scan of a remote object")
        paths = [p[7:] if p.startswith("file://") else p for p in paths]
        st = node.scan_type
        fmt = st[0] if isinstance(st, (tuple, list)) else str(st)
        if fmt not in ("parquet", "ipc"):
            raise NotSupported(f"{fmt} scan")
        if isinstance(st, (tuple, list)) and len(st) > 2 and st[2] not in (None, "null"):
            raise NotSupported("scan with cloud options")
        if getattr(node, "hive_parts", None) is not None:
            raise NotSupported("hive-partitioned scan")
        fo = node.file_options
        if getattr(fo, "row_index", None) is not None:
            raise NotSupported("scan with a row index")
        cols = getattr(fo, "with_columns", None)
        try:
            lf = (scan_parquet if fmt == "parquet" else scan_ipc)(paths if len(paths) > 1 else paths[0], columns=list(cols) if cols is not None else None)
        except F.PlxError as e:                      # unreadable / malformed file: the CPU engine reports it its own way
            raise NotSupported(f"{fmt} file: {e.msg}")
        except (TypeError, ValueError) as e:         # a type outside the hot path, or files whose schemas differ (the CPU engine has its own rules for those)
            raise NotSupported(f"{fmt} scan: {e}")
        # the reference applies the scan's pre_slice physically BEFORE its predicate (polars-stream multi_scan apply_extra_ops.rs: slice
        # at :264, predicate at :334): scan_parquet(p, n_rows=N).filter(pred) = the first N rows, then filtered
        nr = getattr(fo, "n_rows", None)
        if nr is not None:
            lf = lf.slice(int(nr[0]), int(nr[1]))
        if getattr(node, "predicate", None) is not None:
            lf = lf.filter(self.named(node.predicate))
        return lf

    # -- expressions -------------------------------------------------------------------------------------------
    def expr(self, node_id: int) -> E.Expr:
        x = self.nt.view_expression(node_id)
        kind = type(x).__name__
        if kind == "Column":
            return E.col(str(x.name))
        if kind == "Literal":
            dt = _dtype(x.dtype)
            v = x.value
            if v is not None and not isinstance(v, (bool, int, float, _dt.date, _dt.datetime)):
                raise NotSupported(f"literal {type(v).__name__}")
            return E.lit(v, dtype=dt) if dt is not None else E.lit(v)
        if kind == "BinaryExpr":
            name = _op_name(x.op)
            if name not in _OPS:
                raise NotSupported(f"operator {name}")
            return E.Expr("binary", _OPS[name], self.expr(x.left), self.expr(x.right))
        if kind == "Cast":
            dt = _dtype(x.dtype)
            if dt is None:
                raise NotSupported("cast to an unknown dtype")
            return self.expr(x.expr).cast(dt)
        if kind == "Agg":
            name = str(x.name)
            if name not in _AGGS or len(x.arguments) != 1:
                raise NotSupported(f"aggregation {name}")
            if name in ("min", "max") and x.options:
                raise NotSupported("min / max with propagate_nans")
            if name == "count" and x.options:                    # include_nulls: count of rows = len of the column
                return E.Expr("agg", F.AGG_LEN, self.expr(x.arguments[0]))
            return E.Expr("agg", _AGGS[name], self.expr(x.arguments[0]))
        if kind == "Len":
            return E.len()
        if kind == "Alias":
            return self.expr(x.expr).alias(str(x.name))
        raise NotSupported(f"expression node {kind}")

    def named(self, e: Any) -> E.Expr:
        """PyExprIR (visit.rs:19-24): expression node + output name."""
        out = self.expr(e.node)
        return out if P.expr_output_name(out) == e.output_name else out.alias(e.output_name)

    # -- plan nodes --------------------------------------------------------------------------------------------
    def plan(self, node_id: Optional[int] = None) -> LazyFrame:
        nt = self.nt
        if node_id is not None:
            nt.set_node(node_id)
        node = nt.view_current_node()
        kind = type(node).__name__
        if kind == "DataFrameScan":
            if getattr(node, "selection", None) is not None:
                raise NotSupported("scan with a pushed-down selection")
            return self.frame_of(node).lazy()
        if kind == "Scan":
            return self._file_scan(node)
        if kind == "Union":                                 # visitor/nodes.rs:361-370, filled at :783-790
            from .io import concat
            try:
                out = concat([self.plan(i) for i in node.inputs])
            except (TypeError, ValueError) as e:                # differing schemas: the CPU engine applies its own supertype rules
                raise NotSupported(f"union: {e}")
            if getattr(node, "slice", None) is not None:
                out = out.slice(int(node.slice[0]), int(node.slice[1]))
            return out
        if kind == "Filter":
            return self.plan(node.input).filter(self.named(node.predicate))
        if kind in ("Select", "Reduce"):
            return self.plan(node.input).select(*[self.named(e) for e in node.expr])
        if kind == "HStack":
            return self.plan(node.input).with_columns(*[self.named(e) for e in node.exprs])
        if kind == "SimpleProjection":
            inp = self.plan(node.input)
            nt.set_node(node_id if node_id is not None else nt.get_node())
            return inp.select(*[E.col(c) for c in nt.get_schema().keys()])
        if kind == "GroupBy":
            if getattr(node, "apply", None) is not None:
                raise NotSupported("group_by with a Python apply")
            opts = getattr(node, "options", None)
            if opts is not None and (getattr(opts, "dynamic", None) is not None or getattr(opts, "rolling", None) is not None or getattr(opts, "slice", None) is not None):
                raise NotSupported("dynamic / rolling / sliced group_by")
            keys = [self.named(e) for e in node.keys]
            return self.plan(node.input).group_by(*keys, maintain_order=bool(node.maintain_order)).agg(*[self.named(e) for e in node.aggs])
        if kind == "Join":
            how, nulls_equal, jslice, suffix, coalesce, _maintain = node.options      # visitor/nodes.rs:583-652
            how = how if isinstance(how, str) else how[0]
            if how not in ("inner", "left", "semi", "anti") or nulls_equal or jslice is not None or not coalesce and how in ("inner", "left"):
                raise NotSupported(f"join options how={how} nulls_equal={nulls_equal} slice={jslice} coalesce={coalesce}")
            left, right = self.plan(node.input_left), self.plan(node.input_right)
            return left.join(right, left_on=[self.named(e) for e in node.left_on], right_on=[self.named(e) for e in node.right_on], how=how, suffix=str(suffix))
        if kind == "Sort":
            maintain_order, nulls_last, descending = node.sort_options                 # visitor/nodes.rs:533-549
            out = self.plan(node.input).sort([self.named(e) for e in node.by_column], descending=list(descending), nulls_last=list(nulls_last),
                                             maintain_order=bool(maintain_order))
            if node.slice is not None:
                out = out.slice(int(node.slice[0]), int(node.slice[1]))
            return out
        if kind == "Slice":
            return self.plan(node.input).slice(int(node.offset), int(node.len))
        raise NotSupported(f"plan node {kind}")


def _run(lf: LazyFrame, with_columns: Optional[List[str]], predicate: Any, n_rows: Optional[int], should_time: bool):
    """The function PythonScanExec calls (python_scan.rs:97-117)."""
    import polars
    if predicate is not None:
        raise NotSupported("predicate pushed into the GPU scan")
    t0 = time.monotonic_ns()
    out = lf.collect()
    t1 = time.monotonic_ns()
    df = polars.from_arrow(out.to_arrow())
    if with_columns is not None:
        df = df.select(with_columns)
    if n_rows is not None:
        df = df.head(n_rows)
    return (df, [(t0, t1, "amd-gpu: " + F.last_plan()[:120])]) if should_time else df


def execute_with_amd(nt: Any, duration_since_start: Optional[int] = None, *, raise_on_fail: bool = False, frame_of: Optional[Callable[[Any], DataFrame]] = None) -> None:
    """post_opt_callback: translate the optimized plan; on success replace it by a GPU scan, else leave it to the CPU engine."""
    root = nt.get_node()
    try:
        lf = Translator(nt, frame_of).plan()
        lf._lower()                    # type coercion / dtype errors surface here, before the plan is committed to the GPU
    except (NotSupported, TypeError, KeyError, F.UnsupportedError) as e:
        nt.set_node(root)
        if raise_on_fail:
            raise
        _ = e
        return
    nt.set_node(root)
    nt.set_udf(partial(_run, lf), True)


def collect(lazy_frame: Any, *, raise_on_fail: bool = False):
    """`polars.LazyFrame` -> `polars.DataFrame` through the MI355X backend where the plan allows it."""
    return lazy_frame.collect(post_opt_callback=partial(execute_with_amd, raise_on_fail=raise_on_fail))
