"""Logical dtypes of the Python mirror API and their physical (Arrow primitive) layout.

Logical types only matter on the host: the GPU path sees the physical representation,
exactly as the reference's kernels do (Date = i32 days, Datetime = i64 us, Categorical /
dictionary-encoded strings = u32 codes; SURVEY.md section 2 rows 18 and 8(d) cfg 5).
"""
from __future__ import annotations

import numpy as np

from . import _ffi as F


class DataType:
    def __init__(self, name: str, physical: int, np_dtype):
        self.name = name
        self.physical = physical
        self.np_dtype = np.dtype(np_dtype) if np_dtype is not None else None

    def __repr__(self) -> str:
        return self.name

    def __eq__(self, other) -> bool:
        return isinstance(other, DataType) and self.name == other.name

    def __hash__(self) -> int:
        return hash(self.name)

    def is_float(self) -> bool:
        return self.physical in (F.F32, F.F64) and self.name.startswith("Float")

    def is_integer(self) -> bool:
        return self.name.startswith(("Int", "UInt"))

    def is_numeric(self) -> bool:
        return self.is_float() or self.is_integer()


Boolean = DataType("Boolean", F.BOOL, None)
Int8 = DataType("Int8", F.I8, np.int8)
Int16 = DataType("Int16", F.I16, np.int16)
Int32 = DataType("Int32", F.I32, np.int32)
Int64 = DataType("Int64", F.I64, np.int64)
UInt8 = DataType("UInt8", F.U8, np.uint8)
UInt16 = DataType("UInt16", F.U16, np.uint16)
UInt32 = DataType("UInt32", F.U32, np.uint32)
UInt64 = DataType("UInt64", F.U64, np.uint64)
Float32 = DataType("Float32", F.F32, np.float32)
Float64 = DataType("Float64", F.F64, np.float64)
Date = DataType("Date", F.I32, np.int32)          # days since epoch


class DatetimeType(DataType):
    """Datetime(time_unit, time_zone): i64 ticks since the epoch in "ms" | "us" | "ns" (crates/polars-core/src/datatypes/time_unit.rs).
    `Datetime` is the "us" instance (what python datetimes and TPC-H dates become); `Datetime("ns")` makes another, as
    polars.Datetime("ns") does.  All compare equal as dtypes (`dt == Datetime` asks "is this a Datetime"); `time_unit` tells them apart.
    The time zone is carried, never interpreted: the physical values are UTC instants either way."""
    UNITS = ("ms", "us", "ns")

    def __init__(self, time_unit: str = "us", time_zone=None):
        if time_unit not in self.UNITS:
            raise ValueError(f"time unit {time_unit!r} (ms, us or ns)")
        super().__init__("Datetime", F.I64, np.int64)
        self.time_unit, self.time_zone = time_unit, time_zone

    def __call__(self, time_unit: str = "us", time_zone=None) -> "DatetimeType":
        return DatetimeType(time_unit, time_zone)

    def __repr__(self) -> str:
        return "Datetime" if self.time_unit == "us" and self.time_zone is None else f"Datetime(time_unit={self.time_unit!r}, time_zone={self.time_zone!r})"

    def ticks_per_second(self) -> int:
        return {"ms": 1_000, "us": 1_000_000, "ns": 1_000_000_000}[self.time_unit]


Datetime = DatetimeType("us")  # microseconds since epoch


class Categorical(DataType):
    """Dictionary-encoded strings: u32 codes on the device, the dictionary on the host."""

    def __init__(self, categories=(), index_dtype: "DataType | None" = None):
        # Polars' Categorical physical is u32; a narrower code width (u8 / u16) is allowed for
        # low-cardinality dictionaries (TPC-H flags), which is what SURVEY 8(d) sizes Q1 with.
        phys = index_dtype.physical if index_dtype is not None else F.U32
        npdt = index_dtype.np_dtype if index_dtype is not None else np.uint32
        super().__init__("Categorical", phys, npdt)
        self.categories = categories if hasattr(categories, "_load") else list(categories)   # a device-built dictionary stays lazy
        self.from_strings = False          # True: the column was plain strings where it came from (a file scan); to_arrow() gives strings back

    def __eq__(self, other) -> bool:
        return isinstance(other, Categorical)

    def __hash__(self) -> int:
        return hash("Categorical")


PHYSICAL_TO_DTYPE = {F.BOOL: Boolean, F.I8: Int8, F.I16: Int16, F.I32: Int32, F.I64: Int64, F.U8: UInt8, F.U16: UInt16,
                     F.U32: UInt32, F.U64: UInt64, F.F32: Float32, F.F64: Float64}
NP_TO_DTYPE = {np.dtype(np.int8): Int8, np.dtype(np.int16): Int16, np.dtype(np.int32): Int32, np.dtype(np.int64): Int64,
               np.dtype(np.uint8): UInt8, np.dtype(np.uint16): UInt16, np.dtype(np.uint32): UInt32, np.dtype(np.uint64): UInt64,
               np.dtype(np.float32): Float32, np.dtype(np.float64): Float64, np.dtype(np.bool_): Boolean}

_SIGNED = [Int8, Int16, Int32, Int64]
_UNSIGNED = [UInt8, UInt16, UInt32, UInt64]


def supertype(a: DataType, b: DataType) -> DataType:
    """Numeric supertype used by type coercion (restates the numeric part of
    polars-core/src/utils/supertype.rs get_supertype)."""
    if a == b:
        return a
    if a.is_float() or b.is_float():
        if a.is_float() and b.is_float():
            return Float64
        f, o = (a, b) if a.is_float() else (b, a)
        if f == Float32 and o in (Int8, Int16, UInt8, UInt16):
            return Float32
        return Float64
    if a in _SIGNED and b in _SIGNED:
        return _SIGNED[max(_SIGNED.index(a), _SIGNED.index(b))]
    if a in _UNSIGNED and b in _UNSIGNED:
        return _UNSIGNED[max(_UNSIGNED.index(a), _UNSIGNED.index(b))]
    if (a in _SIGNED and b in _UNSIGNED) or (a in _UNSIGNED and b in _SIGNED):
        s, u = (a, b) if a in _SIGNED else (b, a)
        need = _UNSIGNED.index(u) + 1
        if need >= 4:
            return Float64  # u64 vs signed
        return _SIGNED[max(_SIGNED.index(s), need)]
    if a == Boolean and b.is_numeric():
        return b
    if b == Boolean and a.is_numeric():
        return a
    if a.physical == b.physical:
        return a
    raise TypeError(f"no supertype for {a} and {b}")
