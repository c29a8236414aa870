"""Expression DSL mirroring the subset of ``polars.Expr`` that reaches the hot path
(AExpr::{Column, Literal, BinaryExpr, Cast, Agg, Len} and Operator::{Eq..Or},
crates/polars-plan/src/plans/aexpr/mod.rs:150-259, dsl/expr/mod.rs:683-707).
"""
from __future__ import annotations

import datetime as _dt
from typing import Any, Optional

import numpy as np

from . import _ffi as F
from . import datatypes as T


class Expr:
    __slots__ = ("kind", "op", "lhs", "rhs", "name", "value", "dtype")

    def __init__(self, kind: str, op: Optional[int] = None, lhs: "Optional[Expr]" = None, rhs: "Optional[Expr]" = None,
                 name: Optional[str] = None, value: Any = None, dtype: Optional[T.DataType] = None):
        self.kind, self.op, self.lhs, self.rhs, self.name, self.value, self.dtype = kind, op, lhs, rhs, name, value, dtype

    # -- operators ---------------------------------------------------------------------
    def _bin(self, op: int, other: Any, swap: bool = False) -> "Expr":
        o = other if isinstance(other, Expr) else lit(other)
        return Expr("binary", op, o, self) if swap else Expr("binary", op, self, o)

    def __add__(self, o): return self._bin(F.OP_PLUS, o)
    def __radd__(self, o): return self._bin(F.OP_PLUS, o, True)
    def __sub__(self, o): return self._bin(F.OP_MINUS, o)
    def __rsub__(self, o): return self._bin(F.OP_MINUS, o, True)
    def __mul__(self, o): return self._bin(F.OP_MULTIPLY, o)
    def __rmul__(self, o): return self._bin(F.OP_MULTIPLY, o, True)
    def __truediv__(self, o): return self._bin(F.OP_TRUE_DIVIDE, o)
    def __rtruediv__(self, o): return self._bin(F.OP_TRUE_DIVIDE, o, True)
    def __floordiv__(self, o): return self._bin(F.OP_FLOOR_DIVIDE, o)
    def __rfloordiv__(self, o): return self._bin(F.OP_FLOOR_DIVIDE, o, True)
    def __mod__(self, o): return self._bin(F.OP_MODULUS, o)
    def __rmod__(self, o): return self._bin(F.OP_MODULUS, o, True)
    def __eq__(self, o): return self._bin(F.OP_EQ, o)  # type: ignore[override]
    def __ne__(self, o): return self._bin(F.OP_NE, o)  # type: ignore[override]
    def __lt__(self, o): return self._bin(F.OP_LT, o)
    def __le__(self, o): return self._bin(F.OP_LE, o)
    def __gt__(self, o): return self._bin(F.OP_GT, o)
    def __ge__(self, o): return self._bin(F.OP_GE, o)
    def __and__(self, o): return self._bin(F.OP_AND, o)
    def __rand__(self, o): return self._bin(F.OP_AND, o, True)
    def __or__(self, o): return self._bin(F.OP_OR, o)
    def __ror__(self, o): return self._bin(F.OP_OR, o, True)
    def __xor__(self, o): return self._bin(F.OP_XOR, o)
    def __invert__(self): return Expr("not", lhs=self)
    def __hash__(self): return id(self)

    def __neg__(self): return lit(0)._bin(F.OP_MINUS, self)                      # 0 - x, as the reference lowers unary minus for integers / floats
    def is_between(self, lower, upper, closed: str = "both") -> "Expr":
        """lower <= x <= upper (py-polars expr.is_between); closed in {"both", "left", "right", "none"}."""
        if closed not in ("both", "left", "right", "none"):
            raise ValueError(f"closed must be one of 'both', 'left', 'right', 'none', got {closed!r}")
        lo = self.__ge__(lower) if closed in ("both", "left") else self.__gt__(lower)
        hi = self.__le__(upper) if closed in ("both", "right") else self.__lt__(upper)
        return lo & hi

    def is_in(self, values) -> "Expr":
        """x is one of a short list of literals (Expr.is_in with a literal list): OR of equalities; a null x gives null, as the
        reference's is_in does with nulls_equal=False.  Strings compare through the column's dictionary like ==."""
        vals = [v for v in values if v is not None]      # nulls_equal=False: a null in the list equals nothing
        if not vals:
            return self.ne(self)                         # all-false for every non-null x, null for null x
        out = self.eq(vals[0])
        for v in vals[1:]:
            out = out | self.eq(v)
        return out

    def is_null(self) -> "Expr": return Expr("is_null", lhs=self)
    def is_not_null(self) -> "Expr": return Expr("is_not_null", lhs=self)
    def fill_null(self, value: Any) -> "Expr":
        """Replace nulls by a literal (py-polars expr.fill_null(value)); strategies are outside the hot path."""
        if value is None or isinstance(value, Expr) and value.kind != "lit":
            raise TypeError("fill_null takes a non-null literal on this path")
        return Expr("fill_null", lhs=self, rhs=value if isinstance(value, Expr) else lit(value))

    def eq(self, o): return self.__eq__(o)
    def ne(self, o): return self.__ne__(o)
    def not_(self): return self.__invert__()

    # -- aggregations -------------------------------------------------------------------
    def sum(self): return Expr("agg", F.AGG_SUM, self)
    def mean(self): return Expr("agg", F.AGG_MEAN, self)
    def min(self): return Expr("agg", F.AGG_MIN, self)
    def max(self): return Expr("agg", F.AGG_MAX, self)
    def count(self): return Expr("agg", F.AGG_COUNT, self)
    def len(self): return Expr("agg", F.AGG_LEN, self)

    # -- misc -----------------------------------------------------------------------------
    def alias(self, name: str) -> "Expr": return Expr("alias", lhs=self, name=name)
    def cast(self, dtype: T.DataType) -> "Expr": return Expr("cast", lhs=self, dtype=dtype)

    def __repr__(self) -> str:
        if self.kind == "col": return f"col({self.name!r})"
        if self.kind == "lit": return f"lit({self.value!r})"
        if self.kind == "binary": return f"({self.lhs!r} <{self.op}> {self.rhs!r})"
        if self.kind == "agg": return f"{self.lhs!r}.agg{self.op}()"
        if self.kind == "alias": return f"{self.lhs!r}.alias({self.name!r})"
        if self.kind == "cast": return f"{self.lhs!r}.cast({self.dtype})"
        if self.kind == "not": return f"~{self.lhs!r}"
        if self.kind in ("is_null", "is_not_null"): return f"{self.lhs!r}.{self.kind}()"
        if self.kind == "fill_null": return f"{self.lhs!r}.fill_null({self.rhs!r})"
        return self.kind

    def __bool__(self):
        raise TypeError("the truth value of an Expr is ambiguous; use & / | / ~")


def col(name: str) -> Expr:
    return Expr("col", name=name)


def lit(value: Any, dtype: Optional[T.DataType] = None) -> Expr:
    return Expr("lit", value=value, dtype=dtype)


def len() -> Expr:  # noqa: A001 - mirrors pl.len()
    return Expr("len")


def sum(name: str) -> Expr:  # noqa: A001
    return col(name).sum()


def mean(name: str) -> Expr:
    return col(name).mean()


def min(name: str) -> Expr:  # noqa: A001
    return col(name).min()


def max(name: str) -> Expr:  # noqa: A001
    return col(name).max()


def count(name: str) -> Expr:
    return col(name).count()


_EPOCH = _dt.date(1970, 1, 1)


def literal_physical(value: Any):
    """(python value, logical dtype or None for dynamic int/float) of a literal."""
    if value is None:
        return None, None
    if isinstance(value, (bool, np.bool_)):
        return bool(value), T.Boolean
    if isinstance(value, np.generic):
        return value.item(), T.NP_TO_DTYPE[value.dtype]
    if isinstance(value, _dt.datetime):
        delta = value - _dt.datetime(1970, 1, 1)
        return (delta.days * 86400 + delta.seconds) * 1_000_000 + delta.microseconds, T.Datetime
    if isinstance(value, _dt.date):
        return (value - _EPOCH).days, T.Date
    if isinstance(value, int):
        return value, "dyn_int"
    if isinstance(value, float):
        return value, "dyn_float"
    raise TypeError(f"unsupported literal {value!r}")
