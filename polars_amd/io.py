"""Parquet scan -> device columns: the step BEFORE the hot path.

The reference reads Parquet with its own decoder and pushes projections and predicates into the scan
(crates/polars-io/src/parquet/read, crates/polars-plan/src/plans/optimizer/{projection_pushdown, predicate_pushdown},
row-group skipping by statistics: crates/polars-io/src/predicates.rs).  Here:

* the DECODER is on the device (decoder="device", the default): the library parses the footer itself, the selected column chunks
  cross PCIe exactly as they are stored -- compressed and encoded -- and are decompressed / decoded in HBM
  (`plx_parquet_*`, polars_amd/csrc/parquet*.{hpp,cpp} + kernels_parquet.hip; Snappy and Zstandard on the device, gzip / lz4-raw pages inflated by the
  library's own host threads first).  decoder="host" keeps the round-1 path (pyarrow decodes, Arrow buffers are uploaded) for files
  outside the device decoder's codecs / types (brotli, decimals, nested columns),
* projection pushdown -- only the columns the plan reads are fetched (TPC-H Q1 touches 7 of lineitem's 16),
* predicate pushdown to row groups -- conjuncts `column <cmp> literal` of the filters directly above the scan skip the row
  groups whose min / max statistics cannot match (the filter itself still runs on the GPU, exactly).
"""
from __future__ import annotations

import datetime as _dt
import re as _re
from urllib.parse import unquote as _url_unquote
from typing import Any, Dict, List, Optional, Sequence, Set, Tuple

from . import _ffi as F
from . import datatypes as T
from . import plan as P
from .expr import Expr

Pred = Tuple[str, int, Any]          # (column, plx comparison operator, python literal)

_EPOCH = _dt.datetime(1970, 1, 1)


def _mirror_dtype(t) -> T.DataType:
    import pyarrow as pa
    if pa.types.is_dictionary(t) or pa.types.is_string(t) or pa.types.is_large_string(t):
        return T.Categorical([])          # dictionary codes on the device; the dictionary is known after the read
    if pa.types.is_timestamp(t):
        if t.unit == "s":
            raise TypeError("timestamp in seconds (ms, us and ns are on the hot path)")
        return T.Datetime if t.unit == "us" and t.tz is None else T.Datetime(t.unit, t.tz)
    if pa.types.is_date32(t):
        return T.Date
    m = {pa.int8(): T.Int8, pa.int16(): T.Int16, pa.int32(): T.Int32, pa.int64(): T.Int64, pa.uint8(): T.UInt8, pa.uint16(): T.UInt16, pa.uint32(): T.UInt32,
         pa.uint64(): T.UInt64, pa.float32(): T.Float32, pa.float64(): T.Float64, pa.bool_(): T.Boolean}
    if t in m:
        return m[t]
    raise TypeError(f"parquet column type {t} is outside the hot path")


class _HostDecoder:
    """decoder="host": pyarrow reads and decodes on the CPU, the decoded Arrow buffers are uploaded (the round-1 path; kept for files
    the device decoder does not cover: brotli pages, decimals, nested columns)."""
    name = "host"

    def __init__(self, path: str):
        import pyarrow.parquet as pq
        self._pf = pq.ParquetFile(path)
        md = self._pf.metadata
        self.names = list(self._pf.schema_arrow.names)
        self.num_rows, self.num_row_groups = md.num_rows, md.num_row_groups
        self._col_index = {md.schema.column(i).name: i for i in range(md.num_columns)}

    def dtype(self, name: str) -> T.DataType:
        return _mirror_dtype(self._pf.schema_arrow.field(name).type)

    def stats(self, g: int, name: str):
        """(min, max) in the domain pyarrow reports statistics in, or None"""
        if name not in self._col_index:
            return None
        st = self._pf.metadata.row_group(g).column(self._col_index[name]).statistics
        if st is None or not st.has_min_max:
            return None
        return st.min, st.max

    def literal(self, name: str, value: Any, like: Any) -> Any:
        return _comparable(value, like)

    def rows_of(self, g: int) -> int:
        return self._pf.metadata.row_group(g).num_rows

    def read(self, rgs: List[int], cols: List[str]):
        from .frame import DataFrame, Series
        tbl = self._pf.read_row_groups(rgs, columns=cols) if rgs else self._pf.schema_arrow.empty_table().select(cols)
        return DataFrame([Series.from_arrow(n, tbl.column(n)) for n in cols]), tbl.num_rows, tbl.nbytes


def string_column_dtype(categories=()) -> T.Categorical:
    """dtype of a column that is plain strings in the file: u32 dictionary codes here, strings again in to_arrow()"""
    dt = T.Categorical(categories, T.UInt32)
    dt.from_strings = True
    return dt


DATETIME_UNITS = {2: "us", 5: "ms", 6: "ns"}      # the library's logical kinds of Datetime columns (include/polars_amd.h: plx_parquet_column_info)


class _DeviceDecoder:
    """decoder="device" (default): the library parses the footer itself (plx_parquet_open: no pyarrow anywhere on this path), the
    selected column chunks are copied to HBM as stored and decoded by kernels (plx_parquet_read; polars_amd/csrc/parquet*.{hpp,cpp},
    kernels_parquet.hip).  Metadata and statistics are available without a GPU."""
    name = "device"
    _LOGICAL = {1: "date", 2: "datetime", 3: "string", 4: "binary"}

    def __init__(self, path: str):
        import ctypes as C
        h = C.c_uint64()
        F.check(F.lib().plx_parquet_open(path.encode(), C.byref(h)))
        self._h = h.value
        n, g, c = C.c_int64(), C.c_int32(), C.c_int32()
        F.check(F.lib().plx_parquet_shape(self._h, C.byref(n), C.byref(g), C.byref(c)))
        self.num_rows, self.num_row_groups = n.value, g.value
        self.names, self._info = [], {}
        for i in range(c.value):
            nm, dt, lg, nl = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int32()
            F.check(F.lib().plx_parquet_column_info(self._h, i, C.byref(nm), C.byref(dt), C.byref(lg), C.byref(nl)))
            name = nm.value.decode()
            self.names.append(name)
            self._info[name] = (i, dt.value, lg.value, bool(nl.value))

    def __del__(self):
        try:
            if getattr(self, "_h", 0) and F._lib is not None:
                F._lib.plx_parquet_close(self._h)
        except Exception:
            pass

    def dtype(self, name: str) -> T.DataType:
        _, dt, lg, _ = self._info[name]
        if dt < 0:
            raise TypeError(f"parquet column {name!r} has a type outside the hot path (nested, decimal, ...)")
        if lg == 1:
            return T.Date
        if lg in DATETIME_UNITS:
            import ctypes as C
            tz = C.c_char_p()
            F.check(F.lib().plx_parquet_column_timezone(self._h, self._info[name][0], C.byref(tz)))
            zone = tz.value.decode() if tz.value else None
            return T.Datetime if lg == 2 and not zone else T.Datetime(DATETIME_UNITS[lg], zone)
        if lg in (3, 4):
            return string_column_dtype()
        return T.PHYSICAL_TO_DTYPE[dt]

    def rows_of(self, g: int) -> int:
        import ctypes as C
        n = C.c_int64()
        F.check(F.lib().plx_parquet_row_group_info(self._h, g, C.byref(n), None))
        return n.value

    def stats(self, g: int, name: str):
        import ctypes as C
        if name not in self._info:
            return None
        i, dt, lg, _ = self._info[name]
        has, mn, mx = C.c_int32(), F.Scalar(), F.Scalar()
        F.check(F.lib().plx_parquet_chunk_info(self._h, g, i, None, None, None, None, C.byref(has), C.byref(mn), C.byref(mx), None))
        if not has.value:
            return None
        pick = (lambda s: s.f64) if dt == F.F64 else (lambda s: float(s.f32)) if dt == F.F32 else (lambda s: s.u) if dt in (F.U8, F.U16, F.U32, F.U64) else \
               (lambda s: bool(s.u)) if dt == F.BOOL else (lambda s: s.i // 1000) if lg == 6 else (lambda s: s.i)       # Datetime[ns]: in microseconds, see literal()
        return pick(mn), pick(mx)

    def literal(self, name: str, value: Any, like: Any) -> Any:
        """The literal in the PHYSICAL domain the library reports statistics in (Date = days, Datetime = microseconds)."""
        lg = self._info[name][2]
        if lg in DATETIME_UNITS:
            us = None
            if isinstance(value, _dt.datetime):
                v = value.replace(tzinfo=None) if value.tzinfo is None else value.astimezone(_dt.timezone.utc).replace(tzinfo=None)
                d = v - _EPOCH
                us = (d.days * 86400 + d.seconds) * 1_000_000 + d.microseconds
            elif isinstance(value, _dt.date):
                us = (value - _EPOCH.date()).days * 86_400_000_000
            if us is not None:
                # the comparison happens in the coarser of the two units (plan.Lowering._lower_mixed_time_units): ms columns against the
                # floor-divided literal; ns columns are floor-divided into microseconds -- stats() does that to their min / max
                return us // 1000 if lg == 5 else us
        if lg == 1:
            if isinstance(value, _dt.datetime):
                return (value.date() - _EPOCH.date()).days
            if isinstance(value, _dt.date):
                return (value - _EPOCH.date()).days
        if lg == 6 and isinstance(value, (int, float)) and not isinstance(value, bool):
            # a plain number against a Datetime[ns] column is compared in ns ticks by the kernel, while stats() reports this column's
            # min / max floor-divided to microseconds: bring the literal into that domain (floor: every comparison below stays conservative
            # except the strict ones at the boundary, so those do not prune -- see selected_row_groups' TypeError path)
            raise TypeError("numeric literal against Datetime[ns] statistics: not pruned")
        if isinstance(value, (bool, int, float)):
            return value
        raise TypeError("statistics and literal are not comparable")

    def chunk_info(self, g: int, name: str) -> Dict[str, Any]:
        import ctypes as C
        i = self._info[name][0]
        codec, enc, cb, ub, nc = C.c_int32(), C.c_uint32(), C.c_int64(), C.c_int64(), C.c_int64()
        F.check(F.lib().plx_parquet_chunk_info(self._h, g, i, C.byref(codec), C.byref(enc), C.byref(cb), C.byref(ub), None, None, None, C.byref(nc)))
        return {"codec": codec.value, "encodings": enc.value, "compressed_bytes": cb.value, "uncompressed_bytes": ub.value, "null_count": nc.value}

    def read(self, rgs: List[int], cols: List[str]):
        import ctypes as C
        from .frame import DataFrame
        F.ensure_init()
        idx = [self._info[n][0] for n in cols]
        a_rg = (C.c_int32 * max(len(rgs), 1))(*rgs)
        a_col = (C.c_int32 * max(len(idx), 1))(*idx)
        fh = C.c_uint64()
        F.check(F.lib().plx_parquet_read(self._h, a_rg, len(rgs), a_col, len(idx), C.byref(fh)))
        hint = {}
        for n in cols:
            i, dt, lg, _ = self._info[n]
            if lg in (3, 4):
                sd = C.c_uint64()
                if F.lib().plx_parquet_column_strdict(self._h, i, C.byref(sd)) == 0:   # PLAIN string pages: dictionary built on the device, downloaded lazily
                    from .frame import DeviceDictionary
                    hint[n] = string_column_dtype(DeviceDictionary(sd.value, binary=lg == 4))
                else:
                    hint[n] = string_column_dtype(self._categories(i, binary=lg == 4))
            elif lg:
                hint[n] = self.dtype(n)
        df = DataFrame._from_frame_handle(fh.value, hint)
        for s in df.get_columns():
            s._declare_dictionary_bounds()
        nbytes = sum(self.chunk_info(g, n)["compressed_bytes"] for g in rgs for n in cols)
        return df, df.height, nbytes

    def _categories(self, col: int, binary: bool) -> list:
        import ctypes as C

        import numpy as np
        n, tb = C.c_int64(), C.c_int64()
        F.check(F.lib().plx_parquet_categories(self._h, col, C.byref(n), C.byref(tb)))
        off = np.zeros(n.value + 1, np.int64)
        raw = np.zeros(max(tb.value, 1), np.uint8)
        F.check(F.lib().plx_parquet_categories_to_host(self._h, col, off.ctypes.data_as(C.c_void_p), raw.ctypes.data_as(C.c_void_p)))
        b = raw.tobytes()
        items = [b[off[i]:off[i + 1]] for i in range(n.value)]
        return items if binary else [x.decode("utf-8", "replace") for x in items]


def dictionary_union(dictionaries):
    """[categories of part 0, categories of part 1, ...] -> (union in first-appearance order, [u32 remap table of part i: old code -> union code])."""
    import numpy as np
    union, index, remaps = [], {}, []
    for cats in dictionaries:
        cats = list(cats)
        remap = np.empty(len(cats), np.uint32)
        for i, c in enumerate(cats):
            j = index.get(c)
            if j is None:
                j = index[c] = len(union)
                union.append(c)
            remap[i] = j
        remaps.append(remap)
    return union, remaps


def remap_codes(s, remap, union):
    """Dictionary column `s` re-expressed in the dictionary `union`: its codes go through the u32 table `remap` (a gather on the device;
    null codes stay null, the gather honours the index validity)."""
    import numpy as np
    from .frame import Series
    if s.dtype.physical != F.U32:                                   # narrow codes (u8 / u16 dictionaries built by the caller): one width everywhere
        s = s.cast(T.UInt32)
    if len(remap) and not np.array_equal(remap, np.arange(len(remap), dtype=np.uint32)):
        name = s.name
        s = Series("remap", np.ascontiguousarray(remap, np.uint32), T.UInt32).gather(s)
        s.name = name
    else:
        s = s.rename(s.name)
    from_strings = getattr(s.dtype, "from_strings", False)
    s.dtype = T.Categorical(union, T.UInt32)
    s.dtype.from_strings = from_strings
    s._declare_dictionary_bounds()
    return s


def concat_frames(dfs):
    """Vertical concatenation of device frames with equal schemas (plx_frame_concat: device-to-device copies, bitmaps merged at bit
    granularity).  Dictionary columns are first brought onto one dictionary (dictionary_union / remap_codes)."""
    import ctypes as C

    from .frame import DataFrame
    dfs = [d for d in dfs]
    if len(dfs) == 1:
        return dfs[0]
    names = dfs[0].columns
    for d in dfs:
        if d.columns != names:
            raise ValueError(f"frames to concatenate have different columns: {d.columns} vs {names}")
    hint = {}
    for n in names:
        if not isinstance(dfs[0][n].dtype, T.Categorical):
            continue
        union, remaps = dictionary_union([d[n].dtype.categories for d in dfs])
        new = [remap_codes(d[n], remap, union) for d, remap in zip(dfs, remaps)]
        hint[n] = new[0].dtype
        dfs = [DataFrame([new[i] if c.name == n else c for c in d.get_columns()]) for i, d in enumerate(dfs)]
    handles = (C.c_uint64 * len(dfs))(*[d._frame_handle() for d in dfs])
    out = C.c_uint64()
    F.check(F.lib().plx_frame_concat(handles, len(dfs), C.byref(out)))
    for n in names:
        dt = dfs[0][n].dtype
        if n not in hint and dt.name in ("Date", "Datetime"):
            hint[n] = dt
    res = DataFrame._from_frame_handle(out.value, hint)
    for s in res.get_columns():
        s._declare_dictionary_bounds()
    return res


class ConcatFrame:
    """Scan source of concat([...]) (IR::Union; executors/union.rs): the inputs are collected -- each its own plan -- and concatenated on
    the device (concat_frames).  It looks like an in-memory frame to the lowering; projections above it are not pushed into the inputs
    (for several FILES with one schema, scan_parquet / scan_ipc over the list is the tool: it prunes and projects per file)."""

    def __init__(self, lfs):
        self._lfs = list(lfs)
        if not self._lfs:
            raise ValueError("concat of no frames")
        schemas = [lf._lower()[2] for lf in self._lfs]
        first = schemas[0]
        for sc in schemas[1:]:
            if list(sc) != list(first):
                raise ValueError(f"concat: frames have different columns: {list(sc)} vs {list(first)}")
            for n in first:
                a, b = first[n], sc[n]
                if a != b or a.physical != b.physical or getattr(a, "time_unit", None) != getattr(b, "time_unit", None):
                    raise TypeError(f"concat: column {n!r} is {a} in one frame and {b} in another (no supertype casting on this path)")
        self._schema = dict(first)
        self._df = None

    @property
    def schema(self) -> Dict[str, T.DataType]:
        return dict(self._df.schema) if self._df is not None else dict(self._schema)

    def materialise(self):
        if self._df is None:
            self._df = concat_frames([lf.collect() for lf in self._lfs])
        return self._df

    def _frame_handle(self) -> int:
        return self.materialise()._frame_handle()


def concat(items, how: str = "vertical"):
    """polars.concat(items, how="vertical"): DataFrames -> DataFrame (device-to-device copies, dictionaries unified); if any item is a
    LazyFrame -> LazyFrame over a deferred source."""
    from .frame import DataFrame, LazyFrame
    if how != "vertical":
        raise NotImplementedError(f"concat how={how!r} (vertical is on this path)")
    items = list(items)
    if items and all(isinstance(x, DataFrame) for x in items):
        return concat_frames(items)
    return LazyFrame(P.Node("scan", frame=ConcatFrame([x.lazy() if isinstance(x, DataFrame) else x for x in items])))


def split_by_rows(rows: Sequence[int], parts: int) -> List[List[int]]:
    """Indices 0..len(rows)-1 cut into `parts` CONTIGUOUS runs of about equal row totals (entry i goes to the run in which the middle
    of its row range falls): the row-group shards of a scan that several GPUs share (independent row ranges).  Runs
    may be empty when there are fewer row groups than parts."""
    total = sum(rows)
    out: List[List[int]] = [[] for _ in range(parts)]
    acc = 0
    for i, r in enumerate(rows):
        k = min(parts - 1, int((2 * acc + r) * parts // (2 * total))) if total > 0 else 0
        out[k].append(i)
        acc += r
    return out


class _MultiDecoder:
    """Several files behind one scan (a glob or a list of paths: TPC-H tables usually come as directories of files).  Row groups are
    numbered across the files in path order; statistics come from the file a row group lives in; a read fetches file by file (each
    through the device decoder) and concatenates on the device."""

    def __init__(self, paths: Sequence[str], make, hive: Optional[List[Dict[str, str]]] = None):
        self.paths = list(paths)
        self.parts = [make(p) for p in self.paths]
        first = self.parts[0]
        self.name = first.name
        self.names = list(first.names)
        for p, part in zip(self.paths[1:], self.parts[1:]):
            if list(part.names) != self.names:
                raise ValueError(f"{p}: columns differ from {self.paths[0]}")
        # hive partitions (key=value directories under the scanned directory; crates/polars-io/src/hive.rs, crates/polars-plan/src/plans/hive.rs):
        # one constant column per key, after the file's columns; integers when every value parses as one, strings otherwise.  Their
        # "statistics" are exact (min = max = the value), so a predicate on a partition column skips whole files.
        self._hive: Dict[str, list] = {}
        if hive and any(hive):
            keys = list(hive[0])
            if any(list(h) != keys for h in hive):
                raise ValueError("hive partition keys differ between files")
            for k in keys:
                if k in self.names:
                    raise ValueError(f"hive partition key {k!r} is also a column of the files")
                # values are URL-decoded and the default-partition marker is a null (crates/polars-io/src/hive.rs, plans/hive.rs); integers
                # only when every non-null value is a plain decimal (python's int() would also take '1_000' or ' 5')
                raw = [None if h[k] == "__HIVE_DEFAULT_PARTITION__" else _url_unquote(h[k]) for h in hive]
                if any(v is not None for v in raw) and all(v is None or _re.fullmatch(r"-?\d+", v) for v in raw):
                    self._hive[k] = [None if v is None else int(v) for v in raw]
                else:
                    self._hive[k] = raw
            self.names += keys
        self.num_rows = sum(p.num_rows for p in self.parts)
        self._map = [(i, g) for i, p in enumerate(self.parts) for g in range(p.num_row_groups)]
        self.num_row_groups = len(self._map)

    def dtype(self, name: str) -> T.DataType:
        if name in self._hive:
            return T.Int64 if any(isinstance(v, int) for v in self._hive[name]) else string_column_dtype()
        dts = [p.dtype(name) for p in self.parts]
        for p, d in zip(self.paths, dts):
            if d != dts[0] or d.physical != dts[0].physical:
                raise TypeError(f"{p}: column {name!r} is {d}, {self.paths[0]} has {dts[0]}")
        return dts[0]

    def stats(self, g: int, name: str):
        i, lg = self._map[g]
        if name in self._hive:
            v = self._hive[name][i]
            if v is None:
                return None                                     # a null partition value: never pruned through statistics
            return (v, v)                                       # exact: strings too (only == and != reach them, see literal())
        return self.parts[i].stats(lg, name)

    def rows_of(self, g: int) -> int:
        i, lg = self._map[g]
        return self.parts[i].rows_of(lg)

    def literal(self, name: str, value: Any, like: Any) -> Any:
        if name in self._hive:
            if isinstance(like, int) and isinstance(value, int) and not isinstance(value, bool):
                return value
            if isinstance(like, str) and isinstance(value, str):
                return value
            raise TypeError("statistics and literal are not comparable")
        return self.parts[0].literal(name, value, like)

    def _with_hive(self, df, i: int, rows: int, cols: List[str]):
        """the frame of file i with its partition columns appended (constant columns uploaded once per read), in `cols` order"""
        import numpy as np
        from .frame import DataFrame, Series
        have = {c.name: c for c in df.get_columns()} if df is not None else {}
        out = []
        for n in cols:
            if n in self._hive:
                v = self._hive[n][i]
                is_int = any(isinstance(x, int) for x in self._hive[n])
                null = np.zeros(rows, bool) if v is None else None          # __HIVE_DEFAULT_PARTITION__: an all-null column for this file
                if is_int:
                    out.append(Series(n, np.full(rows, v or 0, np.int64), T.Int64, null))
                else:
                    out.append(Series(n, np.zeros(rows, np.uint32), string_column_dtype([] if v is None else [v]), null))
            else:
                out.append(have[n])
        return DataFrame(out)

    def read(self, rgs: List[int], cols: List[str]):
        runs: List[Tuple[int, List[int]]] = []                 # consecutive row groups of one file are one read
        for g in rgs:
            i, lg = self._map[g]
            if runs and runs[-1][0] == i:
                runs[-1][1].append(lg)
            else:
                runs.append((i, [lg]))
        file_cols = [c for c in cols if c not in self._hive]
        if not runs:
            df, r, b = self.parts[0].read([], file_cols)
            return (self._with_hive(df, 0, 0, cols) if self._hive else df), r, b
        dfs, rows, nbytes = [], 0, 0
        for i, lgs in runs:
            if file_cols:
                df, r, b = self.parts[i].read(lgs, file_cols)
            else:                                               # only partition columns are wanted: the row count comes from the metadata
                df, r, b = None, sum(self.parts[i].rows_of(g) for g in lgs), 0
            dfs.append(self._with_hive(df, i, r, cols) if self._hive else df); rows += r; nbytes += b
        return concat_frames(dfs), rows, nbytes


def expand_paths(source, suffixes=(".parquet",)) -> List[str]:
    """str (a file, a directory, or a glob pattern) or a sequence of paths -> the sorted list of files, as polars.scan_parquet accepts."""
    import glob
    import os
    if isinstance(source, (list, tuple)):
        out: List[str] = []
        for s in source:
            out += expand_paths(s, suffixes)
        return out
    source = os.fspath(source)
    if os.path.isdir(source):
        found = sorted(f for sfx in suffixes for f in glob.glob(os.path.join(source, "**", "*" + sfx), recursive=True))
    elif any(ch in source for ch in "*?["):
        found = sorted(glob.glob(source, recursive=True))
    else:
        found = [source]
    if not found:
        raise FileNotFoundError(f"no files match {source!r}")
    return found


def hive_parts(source, files: List[str], force: Optional[bool] = None) -> Optional[List[Dict[str, str]]]:
    """key=value directory names between a scanned DIRECTORY and each of its files (the reference enables hive partitioning by
    default exactly then: a single directory as the source); None for any other source or when no such directory exists.
    force=False: never; force=True: for any source, every key=value directory of the absolute path counts."""
    import os
    if force is False:
        return None
    is_dir = not isinstance(source, (list, tuple)) and os.path.isdir(os.fspath(source))
    if not is_dir and not force:
        return None
    root = os.path.abspath(os.fspath(source)) if is_dir else os.sep
    out = []
    for f in files:
        rel = os.path.relpath(os.path.dirname(os.path.abspath(f)), root)
        parts = {}
        for seg in ([] if rel == "." else rel.split(os.sep)):
            if "=" in seg:
                k, v = seg.split("=", 1)
                parts[k] = v
        out.append(parts)
    return out if any(out) else None


class ParquetFrame:
    """A scan source: looks like a DataFrame to the plan lowering (`schema`, `_frame_handle()`), materialises lazily."""

    def __init__(self, path, columns: Optional[Sequence[str]] = None, decoder: str = "device", shard: Optional[Tuple[int, int]] = None, *,
                 hive_partitioning: Optional[bool] = None, use_statistics: bool = True, include_file_paths: Optional[str] = None):
        if decoder not in ("device", "host"):
            raise ValueError("decoder must be 'device' or 'host'")
        self._set_shard(shard)
        self._use_statistics = bool(use_statistics)
        paths = expand_paths(path)
        self.path = paths[0] if len(paths) == 1 else paths
        make = _DeviceDecoder if decoder == "device" else _HostDecoder
        hive = hive_parts(path, paths, hive_partitioning)
        if include_file_paths:                              # one more constant string column per file (scan_parquet(include_file_paths=...))
            hive = [dict(h, **{include_file_paths: p}) for h, p in zip(hive or [{} for _ in paths], paths)]
        self._dec = make(paths[0]) if len(paths) == 1 and not hive else _MultiDecoder(paths, make, hive)
        names = list(columns) if columns is not None else list(self._dec.names)
        self._schema: Dict[str, T.DataType] = {n: self._dec.dtype(n) for n in names}
        self._need: Optional[Set[str]] = set()          # None = every column of the schema
        self._preds: Optional[List[Pred]] = None        # None = not requested yet; [] = no pushdown
        self._window: Any = None                        # None = not requested yet; (offset, length) of a slice directly above; False = every row
        self._df = None
        self._loaded: Optional[Tuple[frozenset, Tuple[int, ...]]] = None
        self.last_read: Dict[str, Any] = {}

    def _set_shard(self, shard) -> None:
        """shard = (rank, world): this process reads the rank-th of `world` contiguous runs of the row groups that survive pruning
        (split_by_rows); None = all of them."""
        if shard is not None:
            rank, world = int(shard[0]), int(shard[1])
            if not (world >= 1 and 0 <= rank < world):
                raise ValueError(f"shard {shard!r}: need 0 <= rank < world")
            shard = (rank, world)
        self._shard = shard

    @property
    def decoder(self) -> str:
        return self._dec.name

    @property
    def schema(self) -> Dict[str, T.DataType]:
        out = dict(self._schema)
        if self._df is not None:          # loaded columns know their dictionaries
            out.update({n: d for n, d in self._df.schema.items() if n in out})
        return out

    @property
    def num_rows(self) -> int:
        return self._dec.num_rows

    @property
    def num_row_groups(self) -> int:
        return self._dec.num_row_groups

    # -- what the plan needs (called by LazyFrame._lower through plan.push_down) ------------------------------------
    def request(self, columns: Optional[Set[str]], predicates: List[Pred], window: Optional[Tuple[int, int]] = None) -> None:
        """One call per use of this scan in a plan; uses are merged: union of the columns, the predicates only if every use
        carries the same ones, the row window (a slice directly above the scan: slice pushdown, crates/polars-plan/src/plans/optimizer/
        slice_pushdown_lp.rs) only if every use carries the same one."""
        first = self._preds is None
        self._window = (window or False) if first else (self._window if self._window == (window or False) else False)
        if columns is None or self._need is None:
            self._need = None
        else:
            self._need |= set(columns)
        preds = sorted(predicates, key=repr)
        if self._preds is None:
            self._preds = preds
        elif self._preds != preds:
            self._preds = []

    def reset_requests(self) -> None:
        self._need, self._preds, self._window = set(), None, None

    def selected_columns(self) -> List[str]:
        cols = [n for n in self._schema if self._need is None or n in self._need]
        if not cols and self._schema:
            # a plan that reads no column (select(len())) still needs the row count: keep the narrowest column, so the
            # uploaded frame has the scan's height instead of 0
            width = lambda n: getattr(getattr(self._schema[n], "np_dtype", None), "itemsize", None) or 64
            cols = [min(self._schema, key=lambda n: (width(n), list(self._schema).index(n)))]
        return cols

    def selected_row_groups(self) -> List[int]:
        """Row groups that can contain a matching row according to their column statistics."""
        preds = (self._preds or []) if getattr(self, "_use_statistics", True) else []
        keep = []
        for g in range(self._dec.num_row_groups):
            ok = True
            for name, op, value in preds:
                st = self._dec.stats(g, name)
                if st is None:
                    continue
                lo, hi = st
                try:
                    v = self._dec.literal(name, value, lo)
                    if isinstance(lo, float) or isinstance(hi, float) or isinstance(v, float):
                        # Parquet min/max exclude NaN, but the engine compares floats in TOTAL order (NaN == NaN, NaN greatest:
                        # comparisons/simd.rs:171-275), so a group may hold a NaN row that satisfies >, >=, != or == NaN although
                        # the statistics say otherwise.  Only the "below" predicates can be pruned from [min, max] (NaN is never
                        # below a non-NaN literal); a NaN literal prunes nothing.
                        if v != v:
                            possible = True
                        else:
                            possible = {F.OP_LT: lo < v, F.OP_LE: lo <= v}.get(op, True)
                    else:
                        possible = {F.OP_GT: hi > v, F.OP_GE: hi >= v, F.OP_LT: lo < v, F.OP_LE: lo <= v, F.OP_EQ: lo <= v <= hi, F.OP_NE: not (lo == hi == v)}[op]
                except TypeError:
                    possible = True
                if not possible:
                    ok = False
                    break
            if ok:
                keep.append(g)
        if getattr(self, "_shard", None) is not None and self._shard[1] > 1:
            rank, world = self._shard
            keep = [keep[i] for i in split_by_rows([self._dec.rows_of(g) for g in keep], world)[rank]]
        return self._windowed(keep)[0]

    def _windowed(self, keep: List[int]) -> Tuple[List[int], int]:
        """(row groups that overlap the pushed-down slice, rows of `keep` in front of the first of them)"""
        w = getattr(self, "_window", None)
        if not w or self._preds:
            return keep, 0
        rows = [self._dec.rows_of(g) for g in keep]
        total = sum(rows)
        off, length = w
        lo = max(0, total + off) if off < 0 else min(off, total)
        hi = total if off < 0 else min(total, lo + max(length, 0))          # a slice counted from the end keeps the end: its offset is not rebased
        out, skipped, acc = [], 0, 0
        for g, r in zip(keep, rows):
            if acc + r > lo and acc < hi:
                out.append(g)
            elif not out and acc + r <= lo:
                skipped += r
            acc += r
        return out, (skipped if out else 0)

    def window_skip(self, offset: int, length: int) -> int:
        """Rows the scan leaves out in front of the slice (offset, length) it was asked to cover: the plan's Slice node subtracts them
        from its offset.  0 when that slice was not pushed down (another use of the scan needs other rows) or counts from the end."""
        if getattr(self, "_window", None) != (offset, length) or offset < 0:
            return 0
        saved, self._window = self._window, None
        try:
            keep = self.selected_row_groups()
        finally:
            self._window = saved
        return self._windowed(keep)[1]

    def describe(self) -> str:
        """What the current plan makes this scan read (after push_down): the shape of the reference's explain() line for a scan --
        `Parquet SCAN [paths] / PROJECT k/n COLUMNS / SELECTION` -- plus the row groups the statistics, the slice and the shard leave."""
        paths = self.path if isinstance(self.path, list) else [self.path]
        shown = ", ".join(paths[:2]) + (f", ... {len(paths) - 2} more" if len(paths) > 2 else "")
        cols, rgs = self.selected_columns(), self.selected_row_groups()
        ops = {F.OP_EQ: "==", F.OP_NE: "!=", F.OP_LT: "<", F.OP_LE: "<=", F.OP_GT: ">", F.OP_GE: ">="}
        kind = type(self).__name__.replace("Frame", "")
        lines = [f"{kind} SCAN [{shown}] decoder={self.decoder}", f"  PROJECT {len(cols)}/{len(self._dec.names)} COLUMNS: {', '.join(cols)}",
                 f"  ROW GROUPS {len(rgs)}/{self._dec.num_row_groups}"]
        if self._preds and getattr(self, "_use_statistics", True):
            lines.append("  STATISTICS PRUNING: " + " & ".join(f"[{n} {ops.get(o, o)} {v!r}]" for n, o, v in self._preds))
        if getattr(self, "_window", None):
            lines.append(f"  SLICE: offset {self._window[0]}, length {self._window[1]}")
        if getattr(self, "_shard", None) and self._shard[1] > 1:
            lines.append(f"  SHARD: {self._shard[0]} of {self._shard[1]}")
        return "\n".join(lines)

    # -- materialisation ---------------------------------------------------------------------------------------------------
    def materialise(self):
        cols, rgs = self.selected_columns(), self.selected_row_groups()
        key = (frozenset(cols), tuple(rgs))
        if self._df is not None and self._loaded == key:
            return self._df
        self._df, rows, nbytes = self._dec.read(rgs, cols)
        self._loaded = key
        self.last_read = {"columns": cols, "row_groups": len(rgs), "of_row_groups": self._dec.num_row_groups, "rows": rows, "of_rows": self._dec.num_rows,
                          "bytes": nbytes, "decoder": self._dec.name}
        return self._df

    def _frame_handle(self) -> int:
        return self.materialise()._frame_handle()


def _comparable(value: Any, like: Any) -> Any:
    """The literal in the domain of the statistics (pyarrow reports timestamps as datetime, dates as date)."""
    if isinstance(like, _dt.datetime):
        if isinstance(value, _dt.datetime):
            return value.replace(tzinfo=like.tzinfo) if value.tzinfo is None else value
        if isinstance(value, _dt.date):
            return _dt.datetime(value.year, value.month, value.day, tzinfo=like.tzinfo)
        if isinstance(value, int):
            return _dt.datetime(1970, 1, 1, tzinfo=like.tzinfo) + _dt.timedelta(microseconds=value)
    if isinstance(like, _dt.date) and isinstance(value, _dt.datetime):
        return value.date()
    if isinstance(value, (bool, int, float)) and isinstance(like, (bool, int, float)):
        return value
    if type(value) is type(like):
        return value
    raise TypeError("statistics and literal are not comparable")


def scan_parquet(path, columns: Optional[Sequence[str]] = None, decoder: str = "device", shard: Optional[Tuple[int, int]] = None, *, n_rows: Optional[int] = None,
                 hive_partitioning: Optional[bool] = None, use_statistics: bool = True, include_file_paths: Optional[str] = None):
    """LazyFrame over a Parquet file, a directory / glob of files or a list of them (mirrors polars.scan_parquet for the path's dtypes).
    Nothing is read until collect().
    decoder="device": column chunks are decoded on the GPU (UNCOMPRESSED / SNAPPY / ZSTD / GZIP / LZ4_RAW, PLAIN / dictionary pages); "host": pyarrow.
    n_rows / hive_partitioning / use_statistics / include_file_paths: as in polars.scan_parquet.
    shard=(rank, world): one process per GPU, each reading its own run of row groups (polars_amd.dist.scan_shard() gives the pair of the
    running process group; string columns then need dist.unify_dictionaries before their codes meet another rank's)."""
    from .frame import LazyFrame
    lf = LazyFrame(P.Node("scan", frame=ParquetFrame(path, columns, decoder, shard, hive_partitioning=hive_partitioning, use_statistics=use_statistics,
                                                      include_file_paths=include_file_paths)))
    return lf if n_rows is None else lf.head(int(n_rows))        # n_rows: a slice, pushed into the scan (only the row groups it overlaps are read)


def read_parquet(path, columns: Optional[Sequence[str]] = None, decoder: str = "device"):
    """Eager variant: decode (only `columns`) into device columns."""
    pf = ParquetFrame(path, columns, decoder)
    pf.request(None, [])
    return pf.materialise()


# ---- plan analysis: which columns / predicates reach which scan (projection_pushdown / predicate_pushdown restated) --------------
def expr_columns(e: Optional[Expr], out: Optional[Set[str]] = None) -> Set[str]:
    out = set() if out is None else out
    if e is None:
        return out
    if e.kind == "col":
        out.add(e.name)
    for child in (e.lhs, e.rhs):
        if isinstance(child, Expr):
            expr_columns(child, out)
    return out


def output_names(n: P.Node) -> List[str]:
    k = n.kind
    if k == "scan":
        return list(n.frame.schema)
    if k in ("filter", "sort", "slice"):
        return output_names(n.input)
    if k == "select":
        return [P.expr_output_name(e) for e in n.exprs]
    if k == "with_columns":
        names = output_names(n.input)
        return names + [x for x in (P.expr_output_name(e) for e in n.exprs) if x not in names]
    if k == "group_by":
        return [P.expr_output_name(e) for e in n.keys] + [P.expr_output_name(e) for e in n.aggs]
    if k == "join":
        left = output_names(n.left)
        if n.how in ("semi", "anti"):
            return left
        rkeys = {b.name for a, b in zip(n.left_on, n.right_on) if a.kind == "col" and b.kind == "col"}
        return left + [(r + n.suffix if r in left else r) for r in output_names(n.right) if r not in rkeys]
    raise TypeError(k)


def _conjuncts(e: Expr, out: List[Expr]) -> List[Expr]:
    if e.kind == "binary" and e.op == F.OP_AND:
        _conjuncts(e.lhs, out); _conjuncts(e.rhs, out)
    else:
        out.append(e)
    return out


_FLIP = {F.OP_LT: F.OP_GT, F.OP_GT: F.OP_LT, F.OP_LE: F.OP_GE, F.OP_GE: F.OP_LE, F.OP_EQ: F.OP_EQ, F.OP_NE: F.OP_NE}


def simple_predicates(e: Expr) -> List[Pred]:
    """Conjuncts of the form column <cmp> literal (either side)."""
    out: List[Pred] = []
    for c in _conjuncts(e, []):
        if c.kind != "binary" or c.op not in _FLIP:
            continue
        l, r = c.lhs, c.rhs
        if l.kind == "col" and r.kind == "lit" and r.value is not None:
            out.append((l.name, c.op, r.value))
        elif r.kind == "col" and l.kind == "lit" and l.value is not None:
            out.append((r.name, _FLIP[c.op], l.value))
    return out


def _row_preserving(exprs) -> bool:
    """no aggregate anywhere: the node maps row i to row i"""
    def walk(e) -> bool:
        if e is None:
            return True
        if e.kind in ("agg", "len"):
            return False
        return all(walk(c) for c in (e.lhs, e.rhs) if isinstance(c, Expr))
    return all(walk(e) for e in exprs) and any(expr_columns(e) for e in exprs)


def scan_under(node: P.Node):
    """The file scan a Slice node's window reaches: through row-preserving select / with_columns nodes only."""
    while True:
        if node.kind == "scan":
            return node.frame if isinstance(node.frame, ParquetFrame) else None
        if node.kind in ("select", "with_columns") and _row_preserving(node.exprs):
            node = node.input
            continue
        return None


def push_down(node: P.Node, needed: Optional[Set[str]] = None, preds: Optional[List[Pred]] = None, window: Optional[Tuple[int, int]] = None) -> None:
    """Tells every ParquetFrame under `node` which columns the plan reads, which simple predicates sit directly above it and which
    rows a slice directly above it keeps."""
    k = node.kind
    if k == "scan":
        if isinstance(node.frame, ParquetFrame):
            node.frame.request(needed, preds or [], window)
        return
    if k == "filter":
        cols = expr_columns(node.predicate)
        below = preds if preds is not None else []
        push_down(node.input, None if needed is None else needed | cols, below + simple_predicates(node.predicate))
        return
    if k == "select":
        cols: Set[str] = set()
        for e in node.exprs:
            expr_columns(e, cols)
        push_down(node.input, cols, None, window if _row_preserving(node.exprs) else None)
        return
    if k == "with_columns":
        new = {P.expr_output_name(e) for e in node.exprs}
        cols = set()
        for e in node.exprs:
            expr_columns(e, cols)
        push_down(node.input, None if needed is None else (needed - new) | cols, None, window if _row_preserving(node.exprs) else None)
        return
    if k == "group_by":
        cols = set()
        for e in list(node.keys) + list(node.aggs):
            expr_columns(e, cols)
        push_down(node.input, cols, None)
        return
    if k == "sort":
        cols = set()
        for e in node.by:
            expr_columns(e, cols)
        push_down(node.input, None if needed is None else needed | cols, None)
        return
    if k == "slice":
        push_down(node.input, needed, None, (int(node.offset), int(node.length)))
        return
    if k == "join":
        lnames, rnames = output_names(node.left), output_names(node.right)
        lk, rk = set(), set()
        for e in node.left_on:
            expr_columns(e, lk)
        for e in node.right_on:
            expr_columns(e, rk)
        if needed is None:
            push_down(node.left, None, None); push_down(node.right, None, None)
            return
        lneed = {n for n in needed if n in lnames} | lk
        rneed = rk | {n for n in rnames if n in needed or (n + node.suffix) in needed}
        push_down(node.left, lneed, None); push_down(node.right, rneed, None)
        return
    raise TypeError(f"unsupported plan node {k}")


def describe_scans(node: P.Node) -> List[str]:
    """describe() of every file scan under `node`, left to right (requests must have been pushed down: LazyFrame._lower does)."""
    if node.kind == "scan":
        return [node.frame.describe()] if isinstance(node.frame, ParquetFrame) else []
    out: List[str] = []
    for attr in ("input", "left", "right"):
        child = getattr(node, attr, None)
        if isinstance(child, P.Node):
            out += describe_scans(child)
    return out


def has_file_scan(node: P.Node) -> bool:
    if node.kind == "scan":
        return isinstance(node.frame, ParquetFrame)
    return any(isinstance(getattr(node, a, None), P.Node) and has_file_scan(getattr(node, a)) for a in ("input", "left", "right"))


def _children(node: P.Node):
    return [getattr(node, a) for a in ("input", "left", "right") if isinstance(getattr(node, a, None), P.Node)]


def has_deferred_source(node: P.Node) -> bool:
    """A source whose frame (and the dictionaries of its string columns) only exists after materialise(): file scans and concat inputs."""
    if node.kind == "scan":
        return isinstance(node.frame, (ParquetFrame, ConcatFrame))
    return any(has_deferred_source(c) for c in _children(node))


def materialise_sources(node: P.Node) -> None:
    """Reads every deferred source under `node` (after push_down told the file scans what the plan needs)."""
    if node.kind == "scan":
        if isinstance(node.frame, (ParquetFrame, ConcatFrame)):
            node.frame.materialise()
        return
    for c in _children(node):
        materialise_sources(c)


def reset_scans(node: P.Node) -> None:
    if node.kind == "scan":
        if isinstance(node.frame, ParquetFrame):
            node.frame.reset_requests()
        return
    for attr in ("input", "left", "right"):
        child = getattr(node, attr, None)
        if isinstance(child, P.Node):
            reset_scans(child)
