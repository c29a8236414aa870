"""Logical plan of the mirror API and its lowering to the IR / AExpr arenas that cross
the C ABI (plx_ir / plx_aexpr).

In the reference this work is done by polars-plan: DSL -> IR conversion and the
optimizer's type-coercion pass, which inserts ``AExpr::Cast`` so that every
``BinaryExpr`` reaches the physical planner with same-typed operands
(crates/polars-plan/src/plans/conversion/type_coercion/binary.rs:172-...).  polars-plan
is out of scope (kept as the API surface), so this module restates the small part the
hot path needs; the GPU engine itself only ever sees coerced arenas.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Tuple

from . import _ffi as F
from . import datatypes as T
from .expr import Expr, literal_physical

Schema = Dict[str, T.DataType]


# ---- logical plan nodes ------------------------------------------------------------------
class Node:
    def __init__(self, kind: str, **kw):
        self.kind = kind
        self.__dict__.update(kw)


def sum_dtype(dt: T.DataType) -> T.DataType:
    # sum_output_dtype, crates/polars-core/src/chunked_array/ops/aggregate/mod.rs:55-64
    if dt == T.Boolean:
        return T.UInt32
    if dt in (T.Int8, T.Int16, T.UInt8, T.UInt16):
        return T.Int64
    return dt


class Lowering:
    """Builds the arenas. ``aexprs`` / ``irs`` are lists of plain dicts; ``to_c`` marshals
    them into the plx_aexpr / plx_ir structs of include/polars_amd.h."""

    def __init__(self):
        self.aexprs: List[dict] = []
        self.irs: List[dict] = []

    # -- expressions ---------------------------------------------------------------------
    def _push(self, **kw) -> int:
        node = dict(kind=0, op=0, lhs=-1, rhs=-1, dtype=0, is_null=0, lit=None, name=None)
        node.update(kw)
        self.aexprs.append(node)
        return len(self.aexprs) - 1

    def _cast(self, idx: int, frm, to: T.DataType) -> int:
        if frm == to or (isinstance(frm, T.DataType) and frm.physical == to.physical and frm.name == to.name):
            return idx
        return self._push(kind=F.AE_CAST, lhs=idx, dtype=to.physical)

    def _lit_node(self, value, dt: T.DataType) -> int:
        return self._push(kind=F.AE_LITERAL, dtype=dt.physical, lit=value, is_null=0)

    def _materialise_dyn(self, pending, target: Optional[T.DataType]) -> Tuple[int, T.DataType]:
        """A python int / float literal takes the dtype of what it meets."""
        value, tag = pending
        if tag == "dyn_int":
            if target is not None and target.is_integer():
                info_ok = True
                try:
                    import numpy as np
                    ii = np.iinfo(target.np_dtype)
                    info_ok = ii.min <= value <= ii.max
                except Exception:
                    info_ok = False
                if info_ok:
                    return self._lit_node(int(value), target), target
                raise OverflowError(f"literal {value} does not fit {target}")
            if target is not None and target.is_float():
                return self._lit_node(float(value), target), target
            if target is not None and target in (T.Date, T.Datetime):
                return self._lit_node(int(value), target), target
            dt = T.Int32 if -(2 ** 31) <= value < 2 ** 31 else T.Int64
            return self._lit_node(int(value), dt), dt
        # dyn_float
        if target is not None and target == T.Float32:
            return self._lit_node(float(value), T.Float32), T.Float32
        return self._lit_node(float(value), T.Float64), T.Float64

    def lower_expr(self, e: Expr, schema: Schema, want: Optional[T.DataType] = None):
        """Returns (arena index, logical dtype). Unresolved python literals return
        (('dyn', value, tag), None) through ``_lower_maybe_dyn``."""
        idx, dt = self._lower_maybe_dyn(e, schema)
        if isinstance(idx, tuple):
            return self._materialise_dyn(idx, want)
        return idx, dt

    def _lower_maybe_dyn(self, e: Expr, schema: Schema):
        k = e.kind
        if k == "col":
            if e.name not in schema:
                raise KeyError(f"column not found: {e.name}")
            return self._push(kind=F.AE_COLUMN, name=e.name), schema[e.name]
        if k == "lit":
            if e.value is None:
                dt = e.dtype or T.Int32
                return self._push(kind=F.AE_LITERAL, dtype=dt.physical, lit=0, is_null=1), dt
            value, tag = literal_physical(e.value)
            if e.dtype is not None:
                v = float(value) if e.dtype.is_float() else (bool(value) if e.dtype == T.Boolean else int(value))
                return self._lit_node(v, e.dtype), e.dtype
            if isinstance(tag, str):
                return (value, tag), None
            return self._lit_node(value, tag), tag
        if k == "alias":
            idx, dt = self.lower_expr(e.lhs, schema)
            return self._push(kind=F.AE_ALIAS, lhs=idx, name=e.name), dt
        if k == "cast":
            idx, dt = self.lower_expr(e.lhs, schema, e.dtype)
            if isinstance(dt, T.DatetimeType) and isinstance(e.dtype, T.DatetimeType) and dt.time_unit != e.dtype.time_unit:
                raise TypeError(f"cast Datetime[{dt.time_unit}] -> Datetime[{e.dtype.time_unit}] (a multiply / floor-divide of the ticks) is not on this path")
            if dt == e.dtype:
                return idx, dt
            return self._push(kind=F.AE_CAST, lhs=idx, dtype=e.dtype.physical), e.dtype
        if k == "not":
            idx, dt = self.lower_expr(e.lhs, schema)
            if dt != T.Boolean:
                raise TypeError("~ needs a boolean expression")
            return self._push(kind=F.AE_NOT, lhs=idx), T.Boolean
        if k in ("is_null", "is_not_null"):
            idx, _ = self.lower_expr(e.lhs, schema)
            return self._push(kind=F.AE_IS_NULL if k == "is_null" else F.AE_IS_NOT_NULL, lhs=idx), T.Boolean
        if k == "fill_null":
            idx, dt = self.lower_expr(e.lhs, schema)
            li, ldt = self.lower_expr(e.rhs, schema, dt)        # the literal takes the column's dtype (python ints / floats are dynamic)
            if ldt.physical != dt.physical:
                raise TypeError(f"fill_null value of type {ldt} for a {dt} column")
            return self._push(kind=F.AE_FILL_NULL, lhs=idx, rhs=li), dt
        if k == "len":
            return self._push(kind=F.AE_LEN), T.UInt32
        if k == "agg":
            idx, dt = self.lower_expr(e.lhs, schema)
            if e.op == F.AGG_SUM:
                out = sum_dtype(dt)
            elif e.op == F.AGG_MEAN:
                out = T.Float32 if dt == T.Float32 else T.Float64
            elif e.op in (F.AGG_COUNT, F.AGG_LEN):
                out = T.UInt32
            else:
                out = dt
            return self._push(kind=F.AE_AGG, op=e.op, lhs=idx), out
        if k == "binary":
            return self._lower_binary(e, schema)
        raise TypeError(f"unsupported expression {e!r}")

    def _lower_string_compare(self, e: Expr, schema: Schema):
        """Categorical column ==/!= string literal: the string is looked up in the column's dictionary and the physical codes
        are compared -- what the reference does for a Categorical against a string scalar
        (crates/polars-core/src/chunked_array/comparison/categorical.rs: the rev-map lookup, then `equal` on the physical).
        A string that is not in the dictionary equals no row.  Returns None when `e` is not such a comparison."""
        for col_side, lit_side in ((e.lhs, e.rhs), (e.rhs, e.lhs)):
            if lit_side.kind == "lit" and isinstance(lit_side.value, str):
                ci, cdt = self._lower_maybe_dyn(col_side, schema)
                if isinstance(ci, tuple) or not isinstance(cdt, T.Categorical):
                    raise TypeError("a string literal can only be compared with a dictionary-encoded (Categorical) column on this path")
                if e.op not in (F.OP_EQ, F.OP_NE):
                    raise TypeError("only == and != are supported between a Categorical column and a string")
                cats = cdt.categories
                code = cats.index(lit_side.value) if lit_side.value in cats else len(cats)
                phys = T.PHYSICAL_TO_DTYPE[cdt.physical]
                if code > int(__import__("numpy").iinfo(phys.np_dtype).max):      # absent string and a full code space: no row can match
                    return self._lit_node(e.op == F.OP_NE, T.Boolean), T.Boolean
                lit = self._lit_node(code, phys)
                return self._push(kind=F.AE_BINARY, op=e.op, lhs=ci, rhs=lit), T.Boolean
        return None

    def _lower_mixed_time_units(self, op: int, li: int, ldt, ri: int, rdt):
        """Datetime column <cmp> Datetime literal of another time unit (a python datetime is "us").  The reference coerces both sides to
        the COARSER unit (get_time_units, crates/polars-core/src/utils/mod.rs:804-811) and casts the finer side by floor division
        (chunked_array/logical/datetime.rs:59-66: `v.div_euclid(d)`).  No cast kernel runs here: the comparison is rewritten, exactly,
        into the column's own unit.  Column coarser than the literal: the literal is floor-divided.  Column finer by the factor f:
        floor(x / f) <= L  <=>  x <= f L + f - 1;  floor(x / f) >= L  <=>  x >= f L;  < and > shift L by one; == is both bounds, != neither."""
        cmp_ops = (F.OP_EQ, F.OP_NE, F.OP_LT, F.OP_LE, F.OP_GT, F.OP_GE)
        is_lit = lambda i: self.aexprs[i]["kind"] == F.AE_LITERAL and not self.aexprs[i]["is_null"]
        if op not in cmp_ops or is_lit(li) == is_lit(ri):
            raise TypeError(f"Datetime[{ldt.time_unit}] with Datetime[{rdt.time_unit}]: only comparisons of a column with a literal cross time units on this path")
        if is_lit(li):                                              # literal <op> column  ->  column <flipped op> literal
            li, ldt, ri, rdt = ri, rdt, li, ldt
            op = {F.OP_LT: F.OP_GT, F.OP_LE: F.OP_GE, F.OP_GT: F.OP_LT, F.OP_GE: F.OP_LE}.get(op, op)
        L = int(self.aexprs[ri]["lit"])
        cu, lu = ldt.ticks_per_second(), rdt.ticks_per_second()
        cmp = lambda o, v: self._push(kind=F.AE_BINARY, op=o, lhs=li, rhs=self._lit_node(int(v), ldt))
        if cu < lu:                                                 # the column's unit is the coarser one: only the literal is cast
            return cmp(op, L // (lu // cu)), T.Boolean
        f = cu // lu
        i64_min, i64_max = -(1 << 63), (1 << 63) - 1
        # x <= hi / x >= lo with bounds that may lie outside i64: then the comparison is constant for every non-null x (and stays null for nulls)
        le = lambda hi: cmp(F.OP_LE, i64_max) if hi >= i64_max else cmp(F.OP_LT, i64_min) if hi < i64_min else cmp(F.OP_LE, hi)
        ge = lambda lo: cmp(F.OP_GE, i64_min) if lo <= i64_min else cmp(F.OP_GT, i64_max) if lo > i64_max else cmp(F.OP_GE, lo)
        gt = lambda hi: cmp(F.OP_GT, i64_max) if hi >= i64_max else cmp(F.OP_GE, i64_min) if hi < i64_min else cmp(F.OP_GT, hi)
        lt = lambda lo: cmp(F.OP_LT, i64_min) if lo <= i64_min else cmp(F.OP_LE, i64_max) if lo > i64_max else cmp(F.OP_LT, lo)
        upper = lambda l: f * l + f - 1                             # floor(x / f) <= l  <=>  x <= upper(l)
        lower = lambda l: f * l                                     # floor(x / f) >= l  <=>  x >= lower(l)
        if op == F.OP_LE:
            return le(upper(L)), T.Boolean
        if op == F.OP_LT:
            return le(upper(L - 1)), T.Boolean
        if op == F.OP_GE:
            return ge(lower(L)), T.Boolean
        if op == F.OP_GT:
            return ge(lower(L + 1)), T.Boolean
        if op == F.OP_EQ:
            return self._push(kind=F.AE_BINARY, op=F.OP_AND, lhs=ge(lower(L)), rhs=le(upper(L))), T.Boolean
        return self._push(kind=F.AE_BINARY, op=F.OP_OR, lhs=lt(lower(L)), rhs=gt(upper(L))), T.Boolean

    def _lower_binary(self, e: Expr, schema: Schema):
        op = e.op
        if (e.lhs.kind == "lit" and isinstance(e.lhs.value, str)) or (e.rhs.kind == "lit" and isinstance(e.rhs.value, str)):
            return self._lower_string_compare(e, schema)
        li, ldt = self._lower_maybe_dyn(e.lhs, schema)
        ri, rdt = self._lower_maybe_dyn(e.rhs, schema)
        ldyn, rdyn = isinstance(li, tuple), isinstance(ri, tuple)
        if not ldyn and not rdyn and isinstance(ldt, T.DatetimeType) and isinstance(rdt, T.DatetimeType) and ldt.time_unit != rdt.time_unit:
            return self._lower_mixed_time_units(op, li, ldt, ri, rdt)
        if op in (F.OP_AND, F.OP_OR, F.OP_XOR):
            if ldyn or rdyn or ldt != T.Boolean or rdt != T.Boolean:
                raise TypeError("& | ^ need boolean operands on this path")
            return self._push(kind=F.AE_BINARY, op=op, lhs=li, rhs=ri), T.Boolean
        # --- type coercion (type_coercion/binary.rs) ---
        if ldyn and rdyn:
            li, ldt = self._materialise_dyn(li, None)
            ri, rdt = self._materialise_dyn(ri, None)
            ldyn = rdyn = False
        if ldyn or rdyn:
            other_dt = rdt if ldyn else ldt
            value, tag = li if ldyn else ri
            if tag == "dyn_float" and (other_dt.is_integer() or other_dt == T.Boolean):
                # int column vs float literal: the column is cast to Float64
                oi = ri if ldyn else li
                oi = self._cast(oi, other_dt, T.Float64)
                lit_i, _ = self._materialise_dyn((value, tag), T.Float64)
                li, ri = (lit_i, oi) if ldyn else (oi, lit_i)
                ldt = rdt = T.Float64
            else:
                try:
                    lit_i, lit_dt = self._materialise_dyn((value, tag), other_dt)
                except OverflowError:
                    lit_i, lit_dt = self._materialise_dyn((value, tag), None)
                if ldyn:
                    li, ldt = lit_i, lit_dt
                else:
                    ri, rdt = lit_i, lit_dt
        if ldt != rdt or (isinstance(ldt, T.DataType) and ldt.physical != rdt.physical):
            if {ldt.name, rdt.name} <= {"Date", "Datetime"} and ldt != rdt:
                raise TypeError("cannot compare Date with Datetime on this path")
            st = T.supertype(ldt, rdt)
            li = self._cast(li, ldt, st)
            ri = self._cast(ri, rdt, st)
            ldt = rdt = st
        if op in (F.OP_EQ, F.OP_NE, F.OP_LT, F.OP_LE, F.OP_GT, F.OP_GE):
            out = T.Boolean
        elif op == F.OP_TRUE_DIVIDE:
            out = ldt if ldt.is_float() else T.Float64
        else:
            out = ldt
        return self._push(kind=F.AE_BINARY, op=op, lhs=li, rhs=ri), out

    # -- IR ----------------------------------------------------------------------------------
    def lower_node(self, n: Node) -> Tuple[int, Schema]:
        def push(**kw) -> int:
            node = dict(kind=0, input=-1, input_right=-1, predicate=-1, frame=None, exprs=[], keys=[], keys_right=[], how=0,
                        maintain_order=0, suffix="_right", sort_descending=[], sort_nulls_last=[], slice_offset=0, slice_len=0)
            node.update(kw)
            self.irs.append(node)
            return len(self.irs) - 1

        k = n.kind
        if k == "scan":
            return push(kind=F.IR_SCAN, frame=n.frame), dict(n.frame.schema)
        if k == "filter":
            inp, schema = self.lower_node(n.input)
            p, dt = self.lower_expr(n.predicate, schema)
            if dt != T.Boolean:
                raise TypeError("filter predicate must be boolean")
            return push(kind=F.IR_FILTER, input=inp, predicate=p), schema
        if k in ("select", "with_columns"):
            inp, schema = self.lower_node(n.input)
            exprs, out_schema = [], ({} if k == "select" else dict(schema))
            for e in n.exprs:
                idx, dt = self.lower_expr(e, schema)
                exprs.append(idx)
                out_schema[expr_output_name(e)] = dt
            return push(kind=F.IR_SELECT if k == "select" else F.IR_HSTACK, input=inp, exprs=exprs), out_schema
        if k == "group_by":
            inp, schema = self.lower_node(n.input)
            keys, aggs, out_schema = [], [], {}
            for e in n.keys:
                idx, dt = self.lower_expr(e, schema)
                keys.append(idx)
                out_schema[expr_output_name(e)] = dt
            for e in n.aggs:
                idx, dt = self.lower_expr(e, schema)
                aggs.append(idx)
                out_schema[expr_output_name(e)] = dt
            return push(kind=F.IR_GROUPBY, input=inp, keys=keys, exprs=aggs, maintain_order=int(n.maintain_order)), out_schema
        if k == "join":
            li, ls = self.lower_node(n.left)
            ri, rs = self.lower_node(n.right)
            lk, rk = [], []
            for a, b in zip(n.left_on, n.right_on):
                ai, adt = self.lower_expr(a, ls)
                bi, bdt = self.lower_expr(b, rs)
                if isinstance(adt, T.DatetimeType) and isinstance(bdt, T.DatetimeType) and adt.time_unit != bdt.time_unit:
                    raise TypeError(f"join keys Datetime[{adt.time_unit}] and Datetime[{bdt.time_unit}]: casts between time units are not on this path")
                if adt.physical != bdt.physical:
                    st = T.supertype(adt, bdt)
                    ai = self._cast(ai, adt, st)
                    bi = self._cast(bi, bdt, st)
                lk.append(ai)
                rk.append(bi)
            how = {"inner": F.JOIN_INNER, "left": F.JOIN_LEFT, "semi": F.JOIN_SEMI, "anti": F.JOIN_ANTI}[n.how]
            if n.how in ("semi", "anti"):   # left columns only (single_keys_semi_anti.rs)
                return push(kind=F.IR_JOIN, input=li, input_right=ri, keys=lk, keys_right=rk, how=how, suffix=n.suffix), dict(ls)
            out_schema = dict(ls)
            right_key_names = {b.name for a, b in zip(n.left_on, n.right_on) if b.kind == "col" and a.kind == "col"}
            for name, dt in rs.items():
                if name in right_key_names:
                    continue
                out_schema[name + n.suffix if name in out_schema else name] = dt
            return push(kind=F.IR_JOIN, input=li, input_right=ri, keys=lk, keys_right=rk, how=how, suffix=n.suffix), out_schema
        if k == "sort":
            inp, schema = self.lower_node(n.input)
            keys = [self.lower_expr(e, schema)[0] for e in n.by]
            return push(kind=F.IR_SORT, input=inp, keys=keys, sort_descending=[int(bool(x)) for x in n.descending],
                        sort_nulls_last=[int(bool(x)) for x in n.nulls_last], maintain_order=int(n.maintain_order)), schema
        if k == "slice":
            inp, schema = self.lower_node(n.input)
            offset = int(n.offset)
            from . import io as _io
            src = _io.scan_under(n.input)
            if src is not None:                  # the scan below reads only the row groups this slice overlaps: the offset counts from its first row
                offset -= src.window_skip(int(n.offset), int(n.length))
            return push(kind=F.IR_SLICE, input=inp, slice_offset=offset, slice_len=int(n.length)), schema
        raise TypeError(f"unsupported plan node {k}")

    # -- marshalling to the C structs -----------------------------------------------------------
    def to_c(self):
        keep: List[Any] = []
        n_ae = len(self.aexprs)
        ae = (F.AExpr * max(n_ae, 1))()
        for i, d in enumerate(self.aexprs):
            a = ae[i]
            a.kind, a.op, a.lhs, a.rhs, a.dtype, a.is_null = d["kind"], d["op"], d["lhs"], d["rhs"], d["dtype"], d["is_null"]
            if d["kind"] == F.AE_LITERAL and not d["is_null"]:
                dt = d["dtype"]
                if dt == F.F64:
                    a.lit.f64 = float(d["lit"])
                elif dt == F.F32:
                    a.lit.f32 = float(d["lit"])
                elif dt in (F.U8, F.U16, F.U32, F.U64, F.BOOL):
                    a.lit.u = int(d["lit"]) & 0xFFFFFFFFFFFFFFFF
                else:
                    a.lit.i = int(d["lit"])
            if d["name"] is not None:
                b = d["name"].encode()
                keep.append(b)
                a.name = b
        n_ir = len(self.irs)
        ir = (F.IR * n_ir)()
        for i, d in enumerate(self.irs):
            r = ir[i]
            r.kind, r.input, r.input_right, r.predicate = d["kind"], d["input"], d["input_right"], d["predicate"]
            r.frame = d["frame"]._frame_handle() if d["frame"] is not None else 0
            for field, nfield in (("exprs", "n_exprs"), ("keys", "n_keys"), ("keys_right", "n_keys_right")):
                vals = d[field]
                arr = (C.c_int32 * max(len(vals), 1))(*vals)
                keep.append(arr)
                setattr(r, field, C.cast(arr, C.POINTER(C.c_int32)))
                setattr(r, nfield, len(vals))
            r.how, r.maintain_order = d["how"], d["maintain_order"]
            sb = d["suffix"].encode()
            keep.append(sb)
            r.suffix = sb
            if d["kind"] == F.IR_SORT:
                for field in ("sort_descending", "sort_nulls_last"):
                    arr = (C.c_uint8 * max(len(d[field]), 1))(*d[field])
                    keep.append(arr)
                    setattr(r, field, C.cast(arr, C.POINTER(C.c_uint8)))
            r.slice_offset, r.slice_len = d["slice_offset"], d["slice_len"]
        return ir, n_ir, ae, n_ae, keep


def expr_output_name(e: Expr) -> str:
    """Output column name: alias, else the leftmost leaf column (polars convention)."""
    if e.kind == "alias":
        return e.name
    if e.kind == "col":
        return e.name
    if e.kind == "len":
        return "len"
    if e.kind == "lit":
        return "literal"
    return expr_output_name(e.lhs)
