"""Series / DataFrame / LazyFrame mirror of the Polars API for the hot path.

``LazyFrame.collect()`` lowers the plan (plan.py) and hands the arenas to
``plx_execute_plan`` -- the place where, inside Polars, ``create_physical_plan`` would
dispatch to the GPU executor (crates/polars-mem-engine/src/planner/lp.rs:326-878).
Columns live in HBM between operators; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Iterable, List, Optional, Sequence, Union

import numpy as np

from . import _ffi as F
from . import datatypes as T
from . import plan as P
from .expr import Expr, col as _col


def _pack_validity(valid: np.ndarray) -> np.ndarray:
    return np.packbits(np.asarray(valid, dtype=bool), bitorder="little")


class Series:
    """A named device-resident column (handle into libpolars_amd)."""

    def __init__(self, name: str = "", values: Any = None, dtype: Optional[T.DataType] = None, validity: Any = None, *, _handle: int = 0,
                 _dtype: Optional[T.DataType] = None, _keepalive: Any = None, _raw: Any = None):
        self.name = name
        self._keepalive = _keepalive
        self._raw = None                  # (views, data) of a Utf8View column whose dictionary encoding is deferred (from_device_views(encode="deferred"))
        self._hh, self._dt = 0, None
        if _raw is not None:
            self._raw = _raw
            return
        if _handle:
            self._h = _handle
            self.dtype = _dtype if _dtype is not None else self._query_dtype()
            self._declare_dictionary_bounds()
            return
        F.ensure_init()
        self._h, self.dtype = _upload(values, dtype, validity)
        self._declare_dictionary_bounds()

    # The column handle and the dtype are attributes of every Series; for a Utf8View column with deferred encoding they come into being
    # (plx_strview_dict_encode_device) the first time anything asks for them -- a group-by keyed on the column that the string-key operator
    # can serve (LazyFrame.collect -> _string_key_group_by) never does.
    @property
    def _h(self) -> int:
        if not self._hh and self._raw is not None:
            self._encode_raw()
        return self._hh

    @_h.setter
    def _h(self, v: int) -> None:
        self._hh = v

    @property
    def dtype(self) -> T.DataType:
        if self._dt is None and self._raw is not None:
            self._encode_raw()
        return self._dt

    @dtype.setter
    def dtype(self, v) -> None:
        self._dt = v

    def _encode_raw(self) -> None:
        views, data = self._raw
        codes, d = C.c_uint64(), C.c_uint64()
        F.check(F.lib().plx_strview_dict_encode_device(views._h, data._h if data is not None else 0, C.byref(codes), C.byref(d)))
        self._hh, self._dt = codes.value, T.Categorical(DeviceDictionary(d.value), T.UInt32)
        self._declare_dictionary_bounds()

    def _is_raw_views(self) -> bool:
        """A Utf8View column not (yet) dictionary-encoded."""
        return self._raw is not None and not self._hh

    def _declare_dictionary_bounds(self) -> None:
        """Dictionary codes lie in [0, len(categories)): tell the planner (plx_column_set_bounds), which then groups / joins
        on them with dense tables without a statistics pass over the column."""
        if isinstance(self.dtype, T.Categorical) and self.dtype.categories:
            F.check(F.lib().plx_column_set_bounds(self._h, 0, len(self.dtype.categories) - 1))

    # -- construction ----------------------------------------------------------------------
    @classmethod
    def _from_handle(cls, name: str, handle: int, dtype: Optional[T.DataType] = None) -> "Series":
        return cls(name, _handle=handle, _dtype=dtype)

    @classmethod
    def from_device(cls, name: str, dtype: T.DataType, values_ptr: int, n: int, validity_ptr: int = 0, keepalive: Any = None) -> "Series":
        """Wrap caller-owned HBM (e.g. a torch tensor's data_ptr()) without copying."""
        F.ensure_init()
        h = C.c_uint64()
        F.check(F.lib().plx_column_from_device(dtype.physical, C.c_void_p(values_ptr), C.c_void_p(validity_ptr or None), n, C.byref(h)))
        return cls(name, _handle=h.value, _dtype=dtype, _keepalive=keepalive)

    @classmethod
    def from_torch(cls, name: str, tensor, dtype: Optional[T.DataType] = None) -> "Series":
        import torch
        m = {torch.int8: T.Int8, torch.int16: T.Int16, torch.int32: T.Int32, torch.int64: T.Int64, torch.uint8: T.UInt8,
             torch.float32: T.Float32, torch.float64: T.Float64}
        t = tensor.contiguous()
        return cls.from_device(name, dtype or m[t.dtype], t.data_ptr(), t.numel(), keepalive=t)

    @classmethod
    def from_arrow(cls, name: str, arr) -> "Series":
        """Import through the Arrow C Data Interface (plx_column_import_arrow)."""
        import pyarrow as pa
        F.ensure_init()
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks()
        logical = None
        if hasattr(pa.types, "is_string_view") and (pa.types.is_string_view(arr.type) or pa.types.is_binary_view(arr.type)):
            return cls._from_string_views(name, arr)
        if pa.types.is_dictionary(arr.type) or pa.types.is_string(arr.type) or pa.types.is_large_string(arr.type):
            d = arr if pa.types.is_dictionary(arr.type) else arr.dictionary_encode()
            logical = T.Categorical(d.dictionary.to_pylist())
            logical.from_strings = not pa.types.is_dictionary(arr.type)
            arr = d.indices.cast(pa.uint32())
        elif pa.types.is_date32(arr.type):
            logical = T.Date
        elif pa.types.is_timestamp(arr.type):
            if arr.type.unit == "s":                      # the reference has no seconds unit: such arrays become milliseconds on import
                arr = arr.cast(pa.timestamp("ms", arr.type.tz))
            logical = T.Datetime if arr.type.unit == "us" and arr.type.tz is None else T.Datetime(arr.type.unit, arr.type.tz)
        a, s = F.ArrowArray(), F.ArrowSchema()
        arr._export_to_c(C.addressof(a), C.addressof(s))
        h = C.c_uint64()
        F.check(F.lib().plx_column_import_arrow(C.byref(a), C.byref(s), C.byref(h)))
        return cls(name, _handle=h.value, _dtype=logical)

    @classmethod
    def _from_string_views(cls, name: str, arr) -> "Series":
        """Utf8View / BinaryView array -> dictionary column encoded ON THE DEVICE (plx_strview_dict_encode: the 16-byte views and
        the data buffers are uploaded as they are, the library hashes / compares the views the way the reference's BinviewKeys
        do and hands back u32 codes + the dictionary).  Codes are in first-claim order, not sorted."""
        import pyarrow as pa
        F.ensure_init()
        a, s = F.ArrowArray(), F.ArrowSchema()
        arr._export_to_c(C.addressof(a), C.addressof(s))
        try:
            n, nb = a.length, a.n_buffers
            views = a.buffers[1]
            views = (views or 0) + 16 * a.offset
            n_data = max(nb - 3, 0)                       # [validity, views, data..., variadic sizes]
            sizes_ptr = C.cast(a.buffers[nb - 1], C.POINTER(C.c_int64)) if n_data else None
            data_ptrs = (C.c_void_p * max(n_data, 1))(*[a.buffers[2 + i] for i in range(n_data)])
            data_sizes = (C.c_int64 * max(n_data, 1))(*[sizes_ptr[i] for i in range(n_data)])
            validity = a.buffers[0] if a.null_count != 0 else None
            codes, d = C.c_uint64(), C.c_uint64()
            F.check(F.lib().plx_strview_dict_encode(C.c_void_p(views), C.c_void_p(validity), a.offset, n, data_ptrs, data_sizes, n_data, C.byref(codes), C.byref(d)))
        finally:
            for st, ty in ((a, F.ArrowArray), (s, F.ArrowSchema)):
                if st.release:
                    C.CFUNCTYPE(None, C.POINTER(ty))(st.release)(C.byref(st))
        return cls(name, _handle=codes.value, _dtype=T.Categorical(DeviceDictionary(d.value, binary=pa.types.is_binary_view(arr.type)), T.UInt32))

    @classmethod
    def from_device_views(cls, name: str, views: "Series", data: "Optional[Series]" = None, *, validity: "Optional[Series]" = None, encode: str = "eager") -> "Series":
        """Utf8View column whose views already sit in HBM (a UInt64 Series of 2 n words; `data`: a UInt8 Series with the long
        strings' bytes, or None when every string is <= 12 bytes) -> dictionary column, encoded on the device.
        validity: a Boolean Series of n rows (True = valid), the array's validity bitmap; its nulls are stamped into `views` IN PLACE
        (plx_strview_stamp_nulls: a null entry is a view with the length word 0xFFFFFFFF) -- the raw-view operators carry no bitmap.
        encode="deferred": the column stays a column of views until an operator needs dictionary codes; group_by(<this column>).agg(sum /
        mean / count / len of one Float64 / Int64 column) then runs on the views themselves (plx_strview_groupby) and never encodes."""
        if encode not in ("eager", "deferred"):
            raise ValueError(f"encode must be 'eager' or 'deferred', not {encode!r}")
        if validity is not None:
            if len(validity) * 2 != len(views):
                raise ValueError("validity must have one entry per view")
            F.ensure_init()
            F.check(F.lib().plx_strview_stamp_nulls(views._h, validity._h))
        if encode == "deferred":
            if len(views) % 2:
                raise ValueError("views must hold 2 n UInt64 words")
            return cls(name, _raw=(views, data))
        F.ensure_init()
        codes, d = C.c_uint64(), C.c_uint64()
        F.check(F.lib().plx_strview_dict_encode_device(views._h, data._h if data is not None else 0, C.byref(codes), C.byref(d)))
        return cls(name, _handle=codes.value, _dtype=T.Categorical(DeviceDictionary(d.value), T.UInt32))

    def _query_dtype(self) -> T.DataType:
        dt = C.c_int32()
        F.check(F.lib().plx_column_info(self._h, C.byref(dt), None, None))
        return T.PHYSICAL_TO_DTYPE[dt.value]

    def __del__(self):
        try:
            if getattr(self, "_hh", 0) and F._lib is not None:
                F._lib.plx_column_free(self._hh)
        except Exception:
            pass

    # -- metadata -----------------------------------------------------------------------------
    def __len__(self) -> int:
        if self._is_raw_views():
            return len(self._raw[0]) // 2
        n = C.c_int64()
        F.check(F.lib().plx_column_info(self._h, None, C.byref(n), None))
        return n.value

    def null_count(self) -> int:
        n = C.c_int64()
        F.check(F.lib().plx_column_info(self._h, None, None, C.byref(n)))
        return n.value

    def rename(self, name: str) -> "Series":
        if self._is_raw_views():
            return Series(name, _raw=self._raw)
        F.check(F.lib().plx_column_retain(self._h))
        return Series(name, _handle=self._h, _dtype=self.dtype, _keepalive=self._keepalive)

    alias = rename

    def device_ptrs(self):
        v, m = C.c_void_p(), C.c_void_p()
        F.check(F.lib().plx_column_device_ptrs(self._h, C.byref(v), C.byref(m)))
        return v.value or 0, m.value or 0

    # -- download -----------------------------------------------------------------------------
    def _download(self):
        n = len(self)
        phys = self.dtype.physical
        if phys == F.BOOL:
            vbuf = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
        else:
            vbuf = np.zeros(n, dtype=T.PHYSICAL_TO_DTYPE[phys].np_dtype)
        mbuf = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
        hv = C.c_int32()
        F.check(F.lib().plx_column_to_host(self._h, vbuf.ctypes.data_as(C.c_void_p), mbuf.ctypes.data_as(C.c_void_p), C.byref(hv)))
        values = np.unpackbits(vbuf, bitorder="little")[:n].astype(bool) if phys == F.BOOL else vbuf
        valid = np.unpackbits(mbuf, bitorder="little")[:n].astype(bool) if hv.value else None
        if valid is not None and valid.all():
            valid = None
        return values, valid

    def to_numpy(self) -> np.ndarray:
        """Physical values (rows that are null hold unspecified values; see ``validity``)."""
        return self._download()[0]

    def validity(self) -> Optional[np.ndarray]:
        return self._download()[1]

    def to_list(self) -> list:
        values, valid = self._download()
        out = values.tolist()
        if isinstance(self.dtype, T.Categorical) and self.dtype.categories:
            cats = list(self.dtype.categories)
            out = [cats[c] if c < len(cats) else None for c in out]
        if valid is not None:
            out = [v if ok else None for v, ok in zip(out, valid.tolist())]
        return out

    def to_torch(self):
        """Device-to-device copy into a torch tensor on the current CUDA device (values only;
        the column must be null-free).  u16/u32/u64 are reinterpreted as the signed torch dtype."""
        import torch
        if self.null_count():
            raise ValueError("to_torch: column has nulls")
        m = {F.I8: torch.int8, F.I16: torch.int16, F.I32: torch.int32, F.I64: torch.int64, F.U8: torch.uint8, F.U16: torch.int16,
             F.U32: torch.int32, F.U64: torch.int64, F.F32: torch.float32, F.F64: torch.float64}
        n = len(self)
        t = torch.empty(n, dtype=m[self.dtype.physical], device="cuda")
        if n:
            F.check(F.lib().plx_column_copy_to_device(self._h, C.c_void_p(t.data_ptr()), None))
        return t

    def to_arrow(self):
        """Arrow array with the LOGICAL type: the library exports the physical buffers (plx_column_export_arrow), the annotations are put
        back here -- Date -> date32, Datetime -> timestamp[unit, tz], dictionary codes -> a dictionary array (or, for columns that were
        strings in the file they were scanned from, large_string: what the reference's scan of that file yields)."""
        import pyarrow as pa
        a, s = F.ArrowArray(), F.ArrowSchema()
        F.check(F.lib().plx_column_export_arrow(self._h, C.byref(a), C.byref(s)))
        return logical_arrow(pa.Array._import_from_c(C.addressof(a), C.addressof(s)), self.dtype)

    # -- kernel-level operators (one reference kernel family each) ------------------------------
    def _binary_col(self, fn, op: int, other: "Series") -> "Series":
        h = C.c_uint64()
        F.check(fn(op, self._h, other._h, C.byref(h)))
        return Series._from_handle(self.name, h.value)

    def _scalar(self, value) -> F.Scalar:
        s = F.Scalar()
        phys = self.dtype.physical
        if phys == F.F64:
            s.f64 = float(value)
        elif phys == F.F32:
            s.f32 = float(value)
        elif phys in (F.U8, F.U16, F.U32, F.U64):
            s.u = int(value)
        else:
            s.i = int(value)
        return s

    def cmp(self, op: int, other) -> "Series":
        if isinstance(other, Series):
            return self._binary_col(F.lib().plx_cmp, op, other)
        h = C.c_uint64()
        F.check(F.lib().plx_cmp_scalar(op, self._h, self._scalar(other), C.byref(h)))
        return Series._from_handle(self.name, h.value, T.Boolean)

    def __gt__(self, o): return self.cmp(F.GT, o)
    def __ge__(self, o): return self.cmp(F.GE, o)
    def __lt__(self, o): return self.cmp(F.LT, o)
    def __le__(self, o): return self.cmp(F.LE, o)
    def eq(self, o): return self.cmp(F.EQ, o)
    def ne(self, o): return self.cmp(F.NE, o)

    def arith(self, op: int, other, scalar_on_left: bool = False) -> "Series":
        if isinstance(other, Series):
            return self._binary_col(F.lib().plx_arith, op, other)
        h = C.c_uint64()
        F.check(F.lib().plx_arith_scalar(op, self._h, self._scalar(other), int(scalar_on_left), C.byref(h)))
        return Series._from_handle(self.name, h.value)

    def __add__(self, o): return self.arith(F.ADD, o)
    def __sub__(self, o): return self.arith(F.SUB, o)
    def __mul__(self, o): return self.arith(F.MUL, o)
    def __truediv__(self, o): return self.arith(F.TRUE_DIV, o)
    def __floordiv__(self, o): return self.arith(F.FLOOR_DIV, o)
    def __mod__(self, o): return self.arith(F.MOD, o)
    def __radd__(self, o): return self.arith(F.ADD, o, True)
    def __rsub__(self, o): return self.arith(F.SUB, o, True)
    def __rmul__(self, o): return self.arith(F.MUL, o, True)
    def __rtruediv__(self, o): return self.arith(F.TRUE_DIV, o, True)
    def __rfloordiv__(self, o): return self.arith(F.FLOOR_DIV, o, True)
    def __rmod__(self, o): return self.arith(F.MOD, o, True)

    def __and__(self, o: "Series"): return self._binary_col(F.lib().plx_bitmap_binop, F.AND, o)
    def __or__(self, o: "Series"): return self._binary_col(F.lib().plx_bitmap_binop, F.OR, o)
    def __xor__(self, o: "Series"): return self._binary_col(F.lib().plx_bitmap_binop, F.XOR, o)

    def __invert__(self) -> "Series":
        h = C.c_uint64()
        F.check(F.lib().plx_bitmap_not(self._h, C.byref(h)))
        return Series._from_handle(self.name, h.value, T.Boolean)

    def cast(self, dtype: T.DataType) -> "Series":
        h = C.c_uint64()
        F.check(F.lib().plx_cast(self._h, dtype.physical, C.byref(h)))
        return Series._from_handle(self.name, h.value, dtype)

    def filter(self, mask: "Series") -> "Series":
        h = C.c_uint64()
        F.check(F.lib().plx_filter(self._h, mask._h, C.byref(h)))
        return Series._from_handle(self.name, h.value, self.dtype)

    def gather(self, idx: "Series") -> "Series":
        h = C.c_uint64()
        F.check(F.lib().plx_gather(self._h, idx._h, C.byref(h)))
        return Series._from_handle(self.name, h.value, self.dtype)

    def arg_sort(self, *, descending: bool = False, nulls_last: bool = False, limit: int = -1) -> "Series":
        """Stable arg-sort (Series.arg_sort); limit >= 0 keeps the first `limit` indices (top-k selection)."""
        F.ensure_init()
        h = C.c_uint64()
        by = (C.c_uint64 * 1)(self._h)
        d, nl = (C.c_uint8 * 1)(int(descending)), (C.c_uint8 * 1)(int(nulls_last))
        F.check(F.lib().plx_sort_indices(by, 1, d, nl, int(limit), C.byref(h)))
        return Series._from_handle(self.name, h.value, T.UInt32)

    def sort(self, *, descending: bool = False, nulls_last: bool = False) -> "Series":
        return self.gather(self.arg_sort(descending=descending, nulls_last=nulls_last))

    def top_k(self, k: int = 5) -> "Series":
        return self.gather(self.arg_sort(descending=True, nulls_last=True, limit=k))

    def bottom_k(self, k: int = 5) -> "Series":
        return self.gather(self.arg_sort(descending=False, nulls_last=True, limit=k))

    def _reduce(self, op: int):
        v, dt, ok = F.Scalar(), C.c_int32(), C.c_int32()
        F.check(F.lib().plx_reduce(op, self._h, C.byref(v), C.byref(dt), C.byref(ok)))
        if not ok.value:
            return None
        d = dt.value
        if d == F.F64:
            return v.f64
        if d == F.F32:
            return float(np.float32(v.f32))
        if d in (F.U8, F.U16, F.U32, F.U64):
            return int(v.u) & ((1 << (8 * F.DTYPE_WIDTH[d])) - 1)
        if d == F.BOOL:
            return bool(v.u & 1)
        w = 8 * F.DTYPE_WIDTH[d]
        x = int(v.u) & ((1 << w) - 1)
        return x - (1 << w) if x >= (1 << (w - 1)) else x

    def sum(self): return self._reduce(F.AGG_SUM)
    def mean(self): return self._reduce(F.AGG_MEAN)
    def min(self): return self._reduce(F.AGG_MIN)
    def max(self): return self._reduce(F.AGG_MAX)
    def count(self): return self._reduce(F.AGG_COUNT)
    def len(self): return len(self)

    def __repr__(self) -> str:
        return f"Series({self.name!r}, {self.dtype}, len={len(self)})"


def logical_arrow(arr, dtype: T.DataType):
    """physical Arrow array + mirror dtype -> Arrow array of the logical type (host side, no copy for Date / Datetime)"""
    import pyarrow as pa
    if isinstance(dtype, T.Categorical):
        cats = list(dtype.categories)
        if not cats and arr.null_count != len(arr):
            return arr                                    # codes without a known dictionary stay codes
        binary = any(isinstance(c, bytes) for c in cats)
        values = pa.array(cats, pa.large_binary() if binary else pa.large_string())
        d = pa.DictionaryArray.from_arrays(arr.cast(pa.uint32()), values)
        return d.cast(values.type) if getattr(dtype, "from_strings", False) else d
    if dtype.name == "Date":
        return arr.cast(pa.date32())
    if isinstance(dtype, T.DatetimeType):
        return arr.cast(pa.timestamp(dtype.time_unit, dtype.time_zone))
    return arr


def arg_sort_by(by: Sequence["Series"], descending=False, nulls_last=False, limit: int = -1) -> "Series":
    """pl.arg_sort_by over device columns: stable multi-key arg-sort through plx_sort_indices (limit >= 0: top-k)."""
    F.ensure_init()
    n = len(by)
    d, nl = _per_key(descending, n, "descending", "exprs"), _per_key(nulls_last, n, "nulls_last", "exprs")
    h = C.c_uint64()
    F.check(F.lib().plx_sort_indices((C.c_uint64 * n)(*[s._h for s in by]), n, (C.c_uint8 * n)(*map(int, d)), (C.c_uint8 * n)(*map(int, nl)),
                                     int(limit), C.byref(h)))
    return Series._from_handle(by[0].name, h.value, T.UInt32)


class DeviceDictionary:
    """Categories of a dictionary built on the device (plx_strview_dict_encode): a list-like whose strings are downloaded on
    first use only -- a group-by on the codes never needs them, so a 1e6-entry dictionary costs nothing until results are
    printed.  len() is known without a download."""

    def __init__(self, handle: int, binary: bool = False):
        n = C.c_int64()
        F.check(F.lib().plx_strdict_info(handle, C.byref(n), None))
        self._h, self._n, self._binary, self._items = handle, n.value, binary, None

    def _load(self) -> list:
        if self._items is None:
            self._items = _download_dictionary(self._h, self._binary)
            F.lib().plx_strdict_free(self._h)
            self._h = 0
        return self._items

    def __len__(self): return self._n
    def __bool__(self): return self._n > 0
    def __iter__(self): return iter(self._load())
    def __getitem__(self, i): return self._load()[i]
    def __contains__(self, x): return x in self._load()
    def index(self, x): return self._load().index(x)
    def __eq__(self, other): return list(self) == list(other)
    def __repr__(self): return f"DeviceDictionary({self._n} strings)"

    def __del__(self):
        try:
            if getattr(self, "_h", 0) and F._lib is not None:
                F._lib.plx_strdict_free(self._h)
        except Exception:
            pass


def _download_dictionary(dict_handle: int, binary: bool = False) -> list:
    """Strings of a device-built dictionary (plx_strdict_to_host), in code order."""
    n, total = C.c_int64(), C.c_int64()
    F.check(F.lib().plx_strdict_info(dict_handle, C.byref(n), C.byref(total)))
    offsets = np.zeros(n.value + 1, np.int64)
    raw = np.zeros(max(total.value, 1), np.uint8)
    F.check(F.lib().plx_strdict_to_host(dict_handle, offsets.ctypes.data_as(C.c_void_p), raw.ctypes.data_as(C.c_void_p)))
    blob = raw.tobytes()
    o = offsets.tolist()
    if binary:
        return [blob[o[i]:o[i + 1]] for i in range(n.value)]
    return [blob[o[i]:o[i + 1]].decode("utf-8", errors="surrogateescape") for i in range(n.value)]


def _upload(values: Any, dtype: Optional[T.DataType], validity: Any):
    """Host data -> HBM column. Accepts numpy arrays, python lists (None = null),
    numpy masked arrays and pyarrow arrays."""
    try:
        import pyarrow as pa
        if isinstance(values, (pa.Array, pa.ChunkedArray)):
            s = Series.from_arrow("", values)
            h = s._h
            F.check(F.lib().plx_column_retain(h))
            return h, s.dtype
    except ImportError:
        pass
    valid = None
    if isinstance(values, np.ma.MaskedArray):
        valid = ~np.ma.getmaskarray(values)
        values = values.data
    if isinstance(values, (list, tuple)):
        has_none = any(v is None for v in values)
        if any(isinstance(v, str) for v in values):
            cats = sorted({v for v in values if v is not None})
            lut = {c: i for i, c in enumerate(cats)}
            valid = np.array([v is not None for v in values], dtype=bool) if has_none else None
            values = np.array([lut[v] if v is not None else 0 for v in values], dtype=np.uint32)
            dtype = T.Categorical(cats)
        else:
            if has_none:
                valid = np.array([v is not None for v in values], dtype=bool)
                fill = False if all(isinstance(v, bool) for v in values if v is not None) else 0
                values = [fill if v is None else v for v in values]
            if dtype is not None and dtype.np_dtype is not None:
                values = np.array(values, dtype=dtype.np_dtype)
            elif dtype == T.Boolean:
                values = np.array(values, dtype=bool)
            else:
                values = np.array(values)
                if values.dtype == np.dtype("O") or values.size == 0:
                    values = values.astype(np.float64 if dtype is None else dtype.np_dtype)
    values = np.ascontiguousarray(values)
    if validity is not None:
        valid = np.asarray(validity, dtype=bool)
    if dtype is None:
        dtype = T.NP_TO_DTYPE[values.dtype]
    elif dtype.np_dtype is not None and values.dtype != dtype.np_dtype:
        values = values.astype(dtype.np_dtype)
    n = values.shape[0]
    if dtype == T.Boolean:
        vbuf = np.packbits(values.astype(bool), bitorder="little")
    else:
        vbuf = values
    vbuf = np.ascontiguousarray(vbuf)
    vptr = vbuf.ctypes.data_as(C.c_void_p) if vbuf.size else C.c_void_p(0)
    if n == 0:
        dummy = np.zeros(8, dtype=np.uint8)
        vptr = dummy.ctypes.data_as(C.c_void_p)
    mptr = C.c_void_p(0)
    mbuf = None
    if valid is not None and not valid.all():
        mbuf = _pack_validity(valid)
        mptr = mbuf.ctypes.data_as(C.c_void_p)
    h = C.c_uint64()
    F.check(F.lib().plx_column_from_host(dtype.physical, vptr, mptr, 0, n, C.byref(h)))
    return h.value, dtype


class DataFrame:
    def __init__(self, data: Union[Dict[str, Any], Sequence[Series], None] = None):
        self._cols: List[Series] = []
        self._fh = 0
        if data is None:
            return
        if isinstance(data, dict):
            for name, v in data.items():
                self._cols.append(v.rename(name) if isinstance(v, Series) else Series(name, v))
        else:
            self._cols = list(data)
        n = {len(c) for c in self._cols}
        if len(n) > 1:
            raise ValueError(f"columns have different lengths: {sorted(n)}")

    # -- schema -----------------------------------------------------------------------------------
    @property
    def columns(self) -> List[str]:
        return [c.name for c in self._cols]

    @property
    def schema(self) -> Dict[str, T.DataType]:
        return {c.name: c.dtype for c in self._cols}

    @property
    def height(self) -> int:
        return len(self._cols[0]) if self._cols else 0

    @property
    def shape(self):
        return (self.height, len(self._cols))

    def __getitem__(self, name: str) -> Series:
        for c in self._cols:
            if c.name == name:
                return c
        raise KeyError(name)

    def get_columns(self) -> List[Series]:
        return list(self._cols)

    def _frame_handle(self) -> int:
        if not self._fh:
            n = len(self._cols)
            names = (C.c_char_p * max(n, 1))(*[c.name.encode() for c in self._cols])
            hs = (C.c_uint64 * max(n, 1))(*[c._h for c in self._cols])
            h = C.c_uint64()
            F.check(F.lib().plx_frame_new(names, hs, n, C.byref(h)))
            self._fh = h.value
        return self._fh

    def __del__(self):
        try:
            if getattr(self, "_fh", 0) and F._lib is not None:
                F._lib.plx_frame_free(self._fh)
        except Exception:
            pass

    @classmethod
    def _from_frame_handle(cls, fh: int, schema_hint: Optional[Dict[str, T.DataType]] = None) -> "DataFrame":
        w = C.c_int32()
        F.check(F.lib().plx_frame_shape(fh, None, C.byref(w)))
        cols = []
        for i in range(w.value):
            name, h = C.c_char_p(), C.c_uint64()
            F.check(F.lib().plx_frame_column(fh, i, C.byref(name), C.byref(h)))
            nm = name.value.decode()
            s = Series._from_handle(nm, h.value)
            hint = (schema_hint or {}).get(nm)
            if hint is not None and hint.physical == s.dtype.physical:
                s.dtype = hint
            cols.append(s)
        df = cls(cols)
        df._fh = fh
        return df

    # -- eager conveniences (all lazy underneath) ---------------------------------------------
    def lazy(self) -> "LazyFrame":
        return LazyFrame(P.Node("scan", frame=self))

    def filter(self, predicate: Expr) -> "DataFrame":
        return self.lazy().filter(predicate).collect()

    def select(self, *exprs) -> "DataFrame":
        return self.lazy().select(*exprs).collect()

    def with_columns(self, *exprs) -> "DataFrame":
        return self.lazy().with_columns(*exprs).collect()

    def group_by(self, *keys, maintain_order: bool = False) -> "GroupBy":
        return GroupBy(self.lazy(), keys, maintain_order, eager=True)

    def join(self, other: "DataFrame", on=None, how: str = "inner", left_on=None, right_on=None, suffix: str = "_right") -> "DataFrame":
        return self.lazy().join(other.lazy(), on=on, how=how, left_on=left_on, right_on=right_on, suffix=suffix).collect()

    def sort(self, by, *more_by, descending=False, nulls_last=False, maintain_order: bool = False) -> "DataFrame":
        return self.lazy().sort(by, *more_by, descending=descending, nulls_last=nulls_last, maintain_order=maintain_order).collect()

    def slice(self, offset: int, length: Optional[int] = None) -> "DataFrame":
        return self.lazy().slice(offset, length).collect()

    def drop(self, *columns) -> "DataFrame":
        return self.lazy().drop(*columns).collect()

    def rename(self, mapping: Dict[str, str]) -> "DataFrame":
        return self.lazy().rename(mapping).collect()

    def drop_nulls(self, subset=None) -> "DataFrame":
        return self.lazy().drop_nulls(subset).collect()

    def head(self, n: int = 5) -> "DataFrame":
        return self.lazy().head(n).collect()

    def tail(self, n: int = 5) -> "DataFrame":
        return self.lazy().tail(n).collect()

    def top_k(self, k: int, *, by, reverse=False) -> "DataFrame":
        return self.lazy().top_k(k, by=by, reverse=reverse).collect()

    def bottom_k(self, k: int, *, by, reverse=False) -> "DataFrame":
        return self.lazy().bottom_k(k, by=by, reverse=reverse).collect()

    # -- host export --------------------------------------------------------------------------------
    def _download_all(self):
        """[(values, validity or None)] for every column with ONE device synchronisation
        (plx_frame_to_host)."""
        ncol = len(self._cols)
        if ncol == 0:
            return []
        fh = self._frame_handle()
        h = C.c_int64()
        F.check(F.lib().plx_frame_shape(fh, C.byref(h), None))
        n = h.value
        vbufs, mbufs = [], []
        for c in self._cols:
            phys = c.dtype.physical
            vbufs.append(np.zeros((n + 7) // 8 + 8, dtype=np.uint8) if phys == F.BOOL else np.zeros(n, dtype=T.PHYSICAL_TO_DTYPE[phys].np_dtype))
            mbufs.append(np.zeros((n + 7) // 8 + 8, dtype=np.uint8))
        vp = (C.c_void_p * ncol)(*[b.ctypes.data for b in vbufs])
        mp = (C.c_void_p * ncol)(*[b.ctypes.data for b in mbufs])
        hv = (C.c_int32 * ncol)()
        F.check(F.lib().plx_frame_to_host(fh, vp, mp, hv))
        out = []
        for i, c in enumerate(self._cols):
            values = np.unpackbits(vbufs[i], bitorder="little")[:n].astype(bool) if c.dtype.physical == F.BOOL else vbufs[i]
            valid = np.unpackbits(mbufs[i], bitorder="little")[:n].astype(bool) if hv[i] else None
            if valid is not None and valid.all():
                valid = None
            out.append((values, valid))
        return out

    def to_dict(self) -> Dict[str, list]:
        res = {}
        for c, (values, valid) in zip(self._cols, self._download_all()):
            out = values.tolist()
            if isinstance(c.dtype, T.Categorical) and c.dtype.categories:
                cats = list(c.dtype.categories)
                out = [cats[x] if x < len(cats) else None for x in out]
            if valid is not None:
                out = [v if ok else None for v, ok in zip(out, valid.tolist())]
            res[c.name] = out
        return res

    def rows(self) -> List[tuple]:
        cols = [c.to_list() for c in self._cols]
        return list(zip(*cols)) if cols else []

    def to_arrow(self):
        import pyarrow as pa
        return pa.table({c.name: c.to_arrow() for c in self._cols})

    def write_parquet(self, path: str, *, compression: str = "zstd", row_group_size: Optional[int] = None) -> None:
        """Results to a Parquet file (DataFrame.write_parquet; zstd like the reference's default).  The encoder is pyarrow's, on the host:
        writing is the step after the path (result frames are small), only the download is this library's."""
        import pyarrow.parquet as pq
        pq.write_table(self.to_arrow(), path, compression=compression, row_group_size=row_group_size)

    def write_ipc(self, path: str, *, compression: Optional[str] = None) -> None:
        """Results to an Arrow IPC (Feather V2) file (DataFrame.write_ipc); compression: None, "lz4" or "zstd"."""
        import pyarrow as pa
        t = self.to_arrow()
        with pa.ipc.new_file(path, t.schema, options=pa.ipc.IpcWriteOptions(compression=compression)) as w:
            w.write_table(t)

    def sort_host(self, by: Union[str, Sequence[str]]) -> Dict[str, list]:
        """Host-side ordering helper for comparing unordered results (sort / top-k are out
        of scope on the GPU: SURVEY.md section 8e)."""
        by = [by] if isinstance(by, str) else list(by)
        d = self.to_dict()
        n = self.height
        key = lambda i: tuple((d[b][i] is None, d[b][i]) for b in by)
        order = sorted(range(n), key=key)
        return {k: [v[i] for i in order] for k, v in d.items()}

    def __repr__(self) -> str:
        return f"DataFrame(shape={self.shape}, schema={self.schema})"


def _as_exprs(items: Iterable[Any]) -> List[Expr]:
    out: List[Expr] = []
    for it in items:
        if isinstance(it, (list, tuple)):
            out.extend(_as_exprs(it))
        elif isinstance(it, str):
            out.append(_col(it))
        else:
            out.append(it)
    return out


def _per_key(flag, n: int, what: str, keys_name: str) -> List[bool]:
    if isinstance(flag, (list, tuple)):
        if len(flag) != n:
            raise ValueError(f"the length of `{what}` ({len(flag)}) does not match the length of `{keys_name}` ({n})")
        return [bool(x) for x in flag]
    return [bool(flag)] * n


class LazyFrame:
    def __init__(self, node: P.Node):
        self._node = node

    def filter(self, predicate: Expr) -> "LazyFrame":
        return LazyFrame(P.Node("filter", input=self._node, predicate=predicate))

    def select(self, *exprs, **named) -> "LazyFrame":
        es = _as_exprs(exprs) + [e.alias(k) for k, e in named.items()]
        return LazyFrame(P.Node("select", input=self._node, exprs=es))

    def with_columns(self, *exprs, **named) -> "LazyFrame":
        es = _as_exprs(exprs) + [e.alias(k) for k, e in named.items()]
        return LazyFrame(P.Node("with_columns", input=self._node, exprs=es))

    def group_by(self, *keys, maintain_order: bool = False) -> "GroupBy":
        return GroupBy(self, keys, maintain_order)

    def join(self, other: "LazyFrame", on=None, how: str = "inner", left_on=None, right_on=None, suffix: str = "_right") -> "LazyFrame":
        if on is not None:
            left_on = right_on = on
        lo, ro = _as_exprs([left_on]), _as_exprs([right_on])
        return LazyFrame(P.Node("join", left=self._node, right=other._node, left_on=lo, right_on=ro, how=how, suffix=suffix))

    def sort(self, by, *more_by, descending=False, nulls_last=False, maintain_order: bool = False) -> "LazyFrame":
        """LazyFrame.sort (py-polars lazyframe/frame.py sort): `descending` / `nulls_last` are one flag or one per key.
        The GPU sort is always stable, which satisfies maintain_order either way."""
        keys = _as_exprs([by, *more_by])
        return LazyFrame(P.Node("sort", input=self._node, by=keys, descending=_per_key(descending, len(keys), "descending", "by"),
                                nulls_last=_per_key(nulls_last, len(keys), "nulls_last", "by"), maintain_order=maintain_order))

    def collect_schema(self) -> Dict[str, T.DataType]:
        """Output columns and dtypes of the plan (LazyFrame.collect_schema); nothing is executed."""
        return dict(self._lower()[2])

    def drop(self, *columns) -> "LazyFrame":
        """All columns but the named ones (LazyFrame.drop): a projection."""
        gone = {c for x in columns for c in ([x] if isinstance(x, str) else list(x))}
        have = list(self.collect_schema())
        missing = gone - set(have)
        if missing:
            raise KeyError(f"column not found: {sorted(missing)[0]}")
        return self.select(*[_col(n) for n in have if n not in gone])

    def drop_nulls(self, subset=None) -> "LazyFrame":
        """Rows without a null in the named columns (all columns by default) -- LazyFrame.drop_nulls: a filter on the AND of is_not_null."""
        names = list(self.collect_schema()) if subset is None else ([subset] if isinstance(subset, str) else list(subset))
        if not names:
            return self
        pred = _col(names[0]).is_not_null()
        for n in names[1:]:
            pred = pred & _col(n).is_not_null()
        return self.filter(pred)

    def rename(self, mapping: Dict[str, str]) -> "LazyFrame":
        """Columns renamed old -> new, order kept (LazyFrame.rename): a projection with aliases."""
        have = list(self.collect_schema())
        missing = set(mapping) - set(have)
        if missing:
            raise KeyError(f"column not found: {sorted(missing)[0]}")
        return self.select(*[_col(n).alias(mapping[n]) if n in mapping else _col(n) for n in have])

    def slice(self, offset: int, length: Optional[int] = None) -> "LazyFrame":
        if length is not None and length < 0:
            raise ValueError(f"negative slice lengths ({length!r}) are invalid for LazyFrame")
        return LazyFrame(P.Node("slice", input=self._node, offset=int(offset), length=(1 << 40) if length is None else int(length)))

    def head(self, n: int = 5) -> "LazyFrame":
        return self.slice(0, n)

    def limit(self, n: int = 5) -> "LazyFrame":
        return self.head(n)

    def tail(self, n: int = 5) -> "LazyFrame":
        return self.slice(-n, n)

    def top_k(self, k: int, *, by, reverse=False) -> "LazyFrame":
        """The k largest rows by `by` (nulls are the smallest), as sort(descending).head(k): the engine turns a Slice
        directly above a Sort into a radix select (polars-stream/src/nodes/top_k.rs)."""
        keys = _as_exprs([by])
        rev = _per_key(reverse, len(keys), "reverse", "by")
        return self.sort(keys, descending=[not r for r in rev], nulls_last=True).head(k)

    def bottom_k(self, k: int, *, by, reverse=False) -> "LazyFrame":
        keys = _as_exprs([by])
        rev = _per_key(reverse, len(keys), "reverse", "by")
        return self.sort(keys, descending=rev, nulls_last=True).head(k)

    # -- execution -----------------------------------------------------------------------------------
    def _lower(self, materialise: bool = False):
        """Lowers the plan.  `materialise`: first read every deferred source (file scans, concat inputs) -- only then are the dictionaries
        of their string columns known, and both the lowering of string literals (a literal becomes its dictionary code) and the result
        schema depend on them.  Without it (explain(), schema queries) nothing is read."""
        from . import io as _io
        if _io.has_file_scan(self._node):    # file scans: tell them which columns / row groups this plan reads (io.push_down)
            _io.reset_scans(self._node)
            _io.push_down(self._node)
        if materialise:
            _io.materialise_sources(self._node)
        low = P.Lowering()
        root, schema = low.lower_node(self._node)
        return low, root, schema

    def _lowered_c(self):
        """Lowered arenas marshalled to the C structs, cached: a LazyFrame is immutable, so repeated collect() calls
        (the benchmark loop, a served query) skip the ~125 us of Python lowering."""
        cached = getattr(self, "_c_cache", None)
        if cached is None:
            from . import io as _io
            deferred = _io.has_deferred_source(self._node)
            low, root, schema = self._lower(materialise=deferred)
            c_arenas = low.to_c()
            cached = (c_arenas, root, schema, low)
            # A scan source is shared by every LazyFrame derived from the same scan_parquet(): a sibling plan's collect() re-materialises it
            # (other columns / row groups) and frees the frame this plan's arenas point at -- plans over deferred sources are lowered afresh.
            if not deferred:
                self._c_cache = cached
        return cached

    def collect(self, *, no_fusion: bool = False, no_direct_join: bool = False, no_partition: bool = False) -> DataFrame:
        F.ensure_init()
        F.set_plan_note(None)
        if not (no_fusion or no_partition):
            fast = _string_key_group_by(self._node)
            if fast is not None:
                return fast
        (ir, n_ir, ae, n_ae, keep), root, schema, _low = self._lowered_c()
        out = C.c_uint64()
        flags = (F.PLAN_NO_FUSION if no_fusion else 0) | (F.PLAN_NO_DIRECT_JOIN if no_direct_join else 0) | (F.PLAN_NO_PARTITION if no_partition else 0)
        F.check(F.lib().plx_execute_plan(ir, n_ir, ae, n_ae, root, flags, C.byref(out)))
        return DataFrame._from_frame_handle(out.value, schema)

    def explain(self) -> str:
        """Physical plan chosen by the last collect() on this thread; for plans over file scans also what each scan reads (projection,
        row groups left by statistics / slice / shard -- known before anything is read, no GPU needed)."""
        from . import io as _io
        lines = [F.last_plan()] if F._lib is not None and F.last_plan() else []
        if _io.has_file_scan(self._node):
            self._lower()
            lines += _io.describe_scans(self._node)
        return "\n".join(lines)

    def jit_selftest(self) -> None:
        """Compile (not run) the run-time specialised kernels of this query with hiprtc; raises PlxError with the
        compiler log on failure.  Needs no GPU."""
        low, root, _ = self._lower()
        ir, n_ir, ae, n_ae, keep = low.to_c()
        F.check(F.lib().plx_jit_selftest(ir, n_ir, ae, n_ae, root))
        del keep

    def debug_program(self) -> dict:
        """The complete compiled form of this (fusable) pipeline -- register program, immediates, aggregate cells, key
        packing, finalisation -- as a dict (plx_debug_program_json).  Compile only: works on placeholder columns, no GPU."""
        import json
        low, root, _ = self._lower()
        ir, n_ir, ae, n_ae, keep = low.to_c()
        buf = C.create_string_buffer(1 << 16)
        F.check(F.lib().plx_debug_program_json(ir, n_ir, ae, n_ae, root, buf, len(buf)))
        del keep
        return json.loads(buf.value.decode())

    def describe_fusion(self):
        """(fusable, static_shape_id, reason, program dump) -- compile only, no kernel launch."""
        low, root, _ = self._lower()
        ir, n_ir, ae, n_ae, keep = low.to_c()
        fus, sid = C.c_int32(), C.c_int32()
        why = C.create_string_buffer(512)
        F.check(F.lib().plx_describe_fusion(ir, n_ir, ae, n_ae, root, C.byref(fus), C.byref(sid), why, 512))
        del keep
        return bool(fus.value), sid.value, why.value.decode(), F.last_plan()


def _string_key_group_by(node: P.Node) -> Optional[DataFrame]:
    """group_by(<Utf8View column still held as views>).agg(sum / mean / count / len of ONE Float64 / Int64 column) straight over a DataFrame:
    the string-key operator (plx_strview_groupby: rows partitioned by the view's hash, per-partition LDS tables keyed by the view) instead of
    encode-then-group.  None when the plan is anything else or the operator declines (PLX_ERR_UNSUPPORTED: a string over 12 bytes, too many
    distinct strings) -- collect() then goes the usual way, which encodes the column.  Semantics as the reference's group-by on a String
    key (crates/polars-expr/src/hash_keys.rs:413-452): one row per distinct string, sum of an all-null group 0, its mean null."""
    if node.kind != "group_by" or node.maintain_order or len(node.keys) != 1 or node.keys[0].kind != "col" or not node.aggs:
        return None
    src = node.input
    if src.kind != "scan":
        return None
    frame = src.frame
    if type(frame) is not DataFrame:
        # a single-file IPC scan that hands its string columns out as views (scan_ipc(string_keys="deferred")): read what THIS plan needs, then look at the columns
        if getattr(getattr(frame, "_dec", None), "string_keys", "encoded") != "deferred":
            return None
        from . import io as _io
        _io.reset_scans(node)
        _io.push_down(node)
        frame = frame.materialise()
    cols = {c.name: c for c in frame._cols}
    key = cols.get(node.keys[0].name)
    if key is None or not key._is_raw_views():       # (bytes behind the views -- strings over 12 bytes -- do not matter here: the operator declines such a column itself)
        return None
    plan, value = [], None                   # (output name, "sum" | "mean" | "count" | "len")
    for e in node.aggs:
        out = P.expr_output_name(e)
        while e.kind == "alias":
            e = e.lhs
        if e.kind == "len":
            plan.append((out, "len"))
            continue
        kinds = {F.AGG_SUM: "sum", F.AGG_MEAN: "mean", F.AGG_COUNT: "count", F.AGG_LEN: "len"}
        if e.kind != "agg" or e.op not in kinds or e.lhs.kind != "col" or e.lhs.name == key.name or e.lhs.name not in cols:
            return None
        if value is not None and value.name != e.lhs.name:
            return None
        value = cols[e.lhs.name]
        plan.append((out, kinds[e.op]))
    if value is None or value._is_raw_views() or value.dtype not in (T.Float64, T.Int64) or len({o for o, _ in plan} | {key.name}) != len(plan) + 1:
        return None
    hs = [C.c_uint64() for _ in range(5)]
    try:
        F.check(F.lib().plx_strview_groupby(key._raw[0]._h, value._h, *[C.byref(h) for h in hs]))
    except F.UnsupportedError:
        return None
    codes, d, s_sum, s_cnt, s_len = (h.value for h in hs)
    parts = {"sum": Series("__sum", _handle=s_sum, _dtype=value.dtype), "count": Series("__count", _handle=s_cnt, _dtype=T.UInt32), "len": Series("__len", _handle=s_len, _dtype=T.UInt32)}
    k = Series(key.name, _handle=codes, _dtype=T.Categorical(DeviceDictionary(d), T.UInt32))
    if not any(a == "mean" for _, a in plan):
        return DataFrame([k] + [parts[a].rename(o) for o, a in plan])
    # mean = sum / count, null for a group without a valid value (count % count is null exactly then), computed by the library over the G result rows
    n = _col("__count")
    outs = [_col(key.name)] + [((_col("__sum").cast(T.Float64) / (n + n % n).cast(T.Float64)) if a == "mean" else _col("__" + a)).alias(o) for o, a in plan]
    desc = F.last_plan()
    out = DataFrame([k, parts["sum"], parts["count"], parts["len"]]).lazy().select(*outs).collect()
    F.set_plan_note(desc + F.last_plan())
    return out


class GroupBy:
    def __init__(self, lf: LazyFrame, keys, maintain_order: bool, eager: bool = False):
        self._lf, self._keys, self._mo, self._eager = lf, _as_exprs(keys), maintain_order, eager

    def agg(self, *aggs, **named):
        es = _as_exprs(aggs) + [e.alias(k) for k, e in named.items()]
        out = LazyFrame(P.Node("group_by", input=self._lf._node, keys=self._keys, aggs=es, maintain_order=self._mo))
        return out.collect() if self._eager else out

    # -- shorthands of polars' GroupBy / LazyGroupBy (py-polars dataframe/group_by.py, lazyframe/group_by.py) --------------
    def len(self, name: Optional[str] = None):
        """Rows per group (column "len")."""
        from .expr import len as _len
        return self.agg(_len().alias(name or "len"))

    def _value_columns(self) -> List[str]:
        """Every input column that is not a (plain-column) group key."""
        from . import io as _io
        keys = {k.name for k in self._keys if k.kind == "col"}
        return [c for c in _io.output_names(self._lf._node) if c not in keys]

    def _all(self, method: str):
        return self.agg(*[getattr(_col(c), method)() for c in self._value_columns()])

    def sum(self): return self._all("sum")
    def mean(self): return self._all("mean")
    def min(self): return self._all("min")
    def max(self): return self._all("max")
    def count(self): return self._all("count")
