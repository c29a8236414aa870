"""Arrow IPC file (Feather V2) scan -> device columns (the IPC half).

The library parses the FlatBuffers footer and record-batch messages itself (`plx_ipc_*`: polars_amd/csrc/ipc_format.hpp + ipc.cpp; no
pyarrow on this path).  An uncompressed file's buffers are already in the device's layout, so they go file -> page-locked staging ->
HBM without any decode; dictionary indices are widened to u32 codes on the device, Utf8 / LargeUtf8 / Utf8View columns are
dictionary-encoded on the device.  "Row groups" of the scan planning = record batches (IPC carries no statistics: only the projection
is pushed down).  Reference: crates/polars-io/src/ipc/ipc_file.rs, crates/polars-arrow/src/io/ipc/read.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _ffi as F
from . import datatypes as T
from . import plan as P
from .io import DATETIME_UNITS, ParquetFrame, _MultiDecoder, expand_paths, string_column_dtype


class _IpcDecoder:
    name = "device"

    def __init__(self, path: str, string_keys: str = "encoded"):
        h = C.c_uint64()
        F.check(F.lib().plx_ipc_open(path.encode(), C.byref(h)))
        self._h = h.value
        self.string_keys = string_keys
        n, g, c = C.c_int64(), C.c_int32(), C.c_int32()
        F.check(F.lib().plx_ipc_shape(self._h, C.byref(n), C.byref(g), C.byref(c)))
        self.num_rows, self.num_row_groups = n.value, g.value
        self.names, self._info = [], {}
        for i in range(c.value):
            nm, dt, lg, nl = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int32()
            F.check(F.lib().plx_ipc_column_info(self._h, i, C.byref(nm), C.byref(dt), C.byref(lg), C.byref(nl)))
            name = nm.value.decode()
            self.names.append(name)
            self._info[name] = (i, dt.value, lg.value, bool(nl.value))

    def __del__(self):
        try:
            if getattr(self, "_h", 0) and F._lib is not None:
                F._lib.plx_ipc_close(self._h)
        except Exception:
            pass

    def dtype(self, name: str) -> T.DataType:
        _, dt, lg, _ = self._info[name]
        if dt < 0:
            raise TypeError(f"ipc column {name!r} has a type outside the hot path (nested, decimal, timestamp in seconds, ...)")
        if lg == 1:
            return T.Date
        if lg in DATETIME_UNITS:
            tz = C.c_char_p()
            F.check(F.lib().plx_ipc_column_timezone(self._h, self._info[name][0], C.byref(tz)))
            zone = tz.value.decode() if tz.value else None
            return T.Datetime if lg == 2 and not zone else T.Datetime(DATETIME_UNITS[lg], zone)
        if lg in (3, 4):
            return string_column_dtype() if self._plain_strings(name) else T.Categorical([])
        return T.PHYSICAL_TO_DTYPE[dt]

    def _plain_strings(self, name: str) -> bool:
        """True for Utf8 / LargeUtf8 / Utf8View (and binary) columns, False for columns that are dictionary-encoded in the file (those
        stay dictionaries, as the reference reads them: Categorical)."""
        n, tb = C.c_int64(), C.c_int64()
        return F.lib().plx_ipc_categories(self._h, self._info[name][0], C.byref(n), C.byref(tb)) != 0

    def stats(self, g: int, name: str):
        return None

    def rows_of(self, g: int) -> int:
        return self.batch_info(g)["rows"]

    def literal(self, name: str, value: Any, like: Any) -> Any:
        raise TypeError("ipc files carry no statistics")

    def batch_info(self, b: int) -> Dict[str, Any]:
        n, bb, comp = C.c_int64(), C.c_int64(), C.c_int32()
        F.check(F.lib().plx_ipc_batch_info(self._h, b, C.byref(n), C.byref(bb), C.byref(comp)))
        return {"rows": n.value, "body_bytes": bb.value, "compression": {0: None, 1: "lz4", 2: "zstd"}[comp.value]}

    def read(self, rgs: List[int], cols: List[str]):
        from .frame import DataFrame, DeviceDictionary
        F.ensure_init()
        from .frame import Series
        a_b = (C.c_int32 * max(len(rgs), 1))(*rgs)
        # string_keys="deferred": Utf8 / LargeUtf8 columns come as the views the device builds from offsets + bytes, not dictionary-encoded -- a group-by keyed on such
        # a column runs on the views (plx_strview_groupby), anything else encodes the column on first use.  Columns the library does not hand out as views
        # (nulls, Utf8View, dictionary-encoded in the file) are read the usual way.
        raw = {}
        if self.string_keys == "deferred":
            for n in cols:
                i, dt, lg, _ = self._info[n]
                if lg == 3 and self._plain_strings(n):
                    v, d = C.c_uint64(), C.c_uint64()
                    st = F.lib().plx_ipc_read_string_views(self._h, a_b, len(rgs), i, C.byref(v), C.byref(d))
                    if st == F.ERR_UNSUPPORTED:
                        continue
                    F.check(st)
                    raw[n] = Series(n, _raw=(Series._from_handle("views", v.value, T.UInt64), Series._from_handle("data", d.value, T.UInt8)))
        all_cols, cols = cols, [n for n in cols if n not in raw]
        idx = [self._info[n][0] for n in cols]
        a_col = (C.c_int32 * max(len(idx), 1))(*idx)
        fh = C.c_uint64()
        F.check(F.lib().plx_ipc_read(self._h, a_b, len(rgs), a_col, len(idx), C.byref(fh)))
        hint = {}
        for n in cols:
            i, dt, lg, _ = self._info[n]
            if lg in (3, 4):
                sd = C.c_uint64()
                if F.lib().plx_ipc_column_strdict(self._h, i, C.byref(sd)) == 0:       # encoded on the device: the dictionary stays there until asked for
                    hint[n] = string_column_dtype(DeviceDictionary(sd.value, binary=lg == 4))
                else:
                    hint[n] = T.Categorical(self.categories(n), T.UInt32)
            elif lg:
                hint[n] = self.dtype(n)
        df = DataFrame._from_frame_handle(fh.value, hint)
        for s in df.get_columns():
            s._declare_dictionary_bounds()
        if raw:
            have = {s.name: s for s in df.get_columns()}
            df = DataFrame([raw[n] if n in raw else have[n] for n in all_cols])
        nbytes = sum(self.batch_info(b)["body_bytes"] for b in rgs)
        return df, df.height, nbytes

    def categories(self, name: str) -> list:
        """Dictionary values of a column that is dictionary-encoded in the file (host side; no GPU needed)."""
        col, _, lg, _ = self._info[name]
        n, tb = C.c_int64(), C.c_int64()
        F.check(F.lib().plx_ipc_categories(self._h, col, C.byref(n), C.byref(tb)))
        off = np.zeros(n.value + 1, np.int64)
        raw = np.zeros(max(tb.value, 1), np.uint8)
        F.check(F.lib().plx_ipc_categories_to_host(self._h, col, off.ctypes.data_as(C.c_void_p), raw.ctypes.data_as(C.c_void_p)))
        b = raw.tobytes()
        items = [b[off[i]:off[i + 1]] for i in range(n.value)]
        return items if lg == 4 else [x.decode("utf-8", "replace") for x in items]


class IpcFrame(ParquetFrame):
    """Scan source over an Arrow IPC file: the same lazy materialisation and projection pushdown as ParquetFrame."""

    def __init__(self, path, columns: Optional[Sequence[str]] = None, shard=None, string_keys: str = "encoded"):
        if string_keys not in ("encoded", "deferred"):
            raise ValueError(f"string_keys must be 'encoded' or 'deferred', not {string_keys!r}")
        self._set_shard(shard)
        paths = expand_paths(path, suffixes=(".arrow", ".feather", ".ipc"))
        self.path = paths[0] if len(paths) == 1 else paths
        # (several files are concatenated on the device, which needs their dictionaries: deferral is a single-file matter)
        self._dec = _IpcDecoder(paths[0], string_keys) if len(paths) == 1 else _MultiDecoder(paths, _IpcDecoder)
        names = list(columns) if columns is not None else list(self._dec.names)
        self._schema = {n: self._dec.dtype(n) for n in names}
        self._need = set()
        self._preds = None
        self._window = None
        self._df = None
        self._loaded = None
        self.last_read = {}


def scan_ipc(path, columns: Optional[Sequence[str]] = None, shard=None, *, string_keys: str = "encoded"):
    """LazyFrame over an Arrow IPC (Feather V2) file (mirrors polars.scan_ipc for the path's dtypes); uncompressed, LZ4-frame and Zstandard bodies.
    string_keys="deferred" (one file): Utf8 / LargeUtf8 columns without nulls stay columns of views until an operator needs dictionary codes;
    scan_ipc(...).group_by(<such a column>).agg(sum / mean / count / len of one Float64 / Int64 column) then never encodes (plx_strview_groupby)."""
    from .frame import LazyFrame
    return LazyFrame(P.Node("scan", frame=IpcFrame(path, columns, shard, string_keys)))


def read_ipc(path, columns: Optional[Sequence[str]] = None):
    pf = IpcFrame(path, columns)
    pf.request(None, [])
    return pf.materialise()
