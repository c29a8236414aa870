"""polars_amd -- MI355X-native execution backend for the Polars hot path
(filter / gather, primitive compare + arithmetic, hash group-by-aggregate, hash join).

The compute lives in ``libpolars_amd.so`` (hand-written gfx950 HIP kernels + C++ host
engine) behind the C ABI of ``include/polars_amd.h``.  This package is the Python
stand-in for the Rust host shim: a small mirror of the Polars LazyFrame / Expr API that
lowers queries to the IR / AExpr arenas the ABI consumes.  There is no CPU fallback.
"""
from . import _ffi, plan
from ._ffi import PlxError, UnsupportedError, init, last_plan
from .datatypes import (Boolean, Categorical, DataType, Date, Datetime, Float32, Float64, Int8, Int16, Int32, Int64, UInt8,
                        UInt16, UInt32, UInt64)
from .expr import Expr, col, count, len, lit, max, mean, min, sum  # noqa: A004
from .frame import DataFrame, GroupBy, LazyFrame, Series, arg_sort_by
from .io import concat, read_parquet, scan_parquet
from .ipc_io import read_ipc, scan_ipc

__all__ = ["init", "last_plan", "PlxError", "UnsupportedError", "DataFrame", "LazyFrame", "GroupBy", "Series", "arg_sort_by", "scan_parquet", "read_parquet", "concat", "scan_ipc", "read_ipc", "Expr", "col", "lit",
           "len", "sum", "mean", "min", "max", "count", "DataType", "Boolean", "Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16",
           "UInt32", "UInt64", "Float32", "Float64", "Date", "Datetime", "Categorical"]
__version__ = "0.1.0"


def datagen_categories(column: str):
    """Dictionary of a dictionary-encoded TPC-H column (host side)."""
    from . import datagen
    return {"l_returnflag": datagen.FLAGS, "l_linestatus": datagen.STATUS}[column]
