"""plx_parquet_* without a GPU: the library's own footer / schema / statistics reader (polars_amd/csrc/parquet_format.hpp: Thrift compact
protocol, FileMetaData, logical types) against pyarrow's view of the same files, and the loud failure of plx_parquet_read when no
device is bound (there is no host decode path in the product)."""
import ctypes as C
import datetime as dt

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import polars_amd as pl
from polars_amd import _ffi as F
from polars_amd import io

RNG = np.random.default_rng(5)


def sample(n):
    m = RNG.random(n) < 0.2
    return pa.table({
        "i8": pa.array(RNG.integers(-100, 100, n).astype(np.int8)), "u16": pa.array(RNG.integers(0, 60000, n).astype(np.uint16)),
        "i32": pa.array(RNG.integers(-10**9, 10**9, n).astype(np.int32), mask=m), "u32": pa.array(RNG.integers(0, 2**32, n).astype(np.uint32)),
        "i64": pa.array(np.sort(RNG.integers(-10**15, 10**15, n))), "u64": pa.array(RNG.integers(0, 2**63, n).astype(np.uint64) * 2),
        "f32": pa.array(RNG.normal(size=n).astype(np.float32)), "f64": pa.array(RNG.normal(size=n), mask=m), "b": pa.array(RNG.random(n) < 0.5),
        "date": pa.array(RNG.integers(0, 20000, n).astype(np.int32), pa.date32()), "ts": pa.array(np.sort(RNG.integers(0, 2**50, n)), pa.timestamp("us")),
        "ts_ms": pa.array(RNG.integers(0, 2**40, n), pa.timestamp("ms")), "s": pa.array(np.array(["a", "bb", "ccc"])[RNG.integers(0, 3, n)]),
        "bin": pa.array([b"x"] * n, pa.binary()), "dec": pa.array([None] * n, pa.decimal128(12, 2)), "lst": pa.array([[1, 2]] * n),
        "st": pa.array([{"p": 1, "q": "z"}] * n),
    })


def lib_meta(path):
    h = C.c_uint64()
    F.check(F.lib().plx_parquet_open(path.encode(), C.byref(h)))
    return h.value


def test_schema_shape_and_statistics_match_pyarrow(tmp_path):
    n = 10_000
    t = sample(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=3000, compression="snappy")
    md = pq.ParquetFile(path).metadata
    h = lib_meta(path)
    rows, groups, cols = C.c_int64(), C.c_int32(), C.c_int32()
    F.check(F.lib().plx_parquet_shape(h, C.byref(rows), C.byref(groups), C.byref(cols)))
    assert (rows.value, groups.value, cols.value) == (md.num_rows, md.num_row_groups, md.num_columns)      # leaf columns, nested ones included
    want = {"i8": (F.I8, 0), "u16": (F.U16, 0), "i32": (F.I32, 0), "u32": (F.U32, 0), "i64": (F.I64, 0), "u64": (F.U64, 0), "f32": (F.F32, 0), "f64": (F.F64, 0),
            "b": (F.BOOL, 0), "date": (F.I32, 1), "ts": (F.I64, 2), "ts_ms": (F.I64, 5), "s": (F.U32, 3), "bin": (F.U32, 4), "dec": (-1, 0)}
    names = []
    for i in range(cols.value):
        nm, dtp, lg, nl = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int32()
        F.check(F.lib().plx_parquet_column_info(h, i, C.byref(nm), C.byref(dtp), C.byref(lg), C.byref(nl)))
        name = nm.value.decode()
        names.append(name)
        assert name == md.schema.column(i).path
        if name in want:
            assert (dtp.value, lg.value) == want[name], name
            assert nl.value == 1
        else:
            assert dtp.value == -1 and "." in name          # leaves of the list / struct columns: outside the hot path
    assert names[:15] == t.column_names[:15]
    for g in range(groups.value):
        gr, gb = C.c_int64(), C.c_int64()
        F.check(F.lib().plx_parquet_row_group_info(h, g, C.byref(gr), C.byref(gb)))
        rg = md.row_group(g)
        assert gr.value == rg.num_rows and gb.value == sum(rg.column(i).total_compressed_size for i in range(md.num_columns))
        for i, name in enumerate(names):
            col = rg.column(i)
            codec, enc, cb, ub, has, mn, mx, nc = C.c_int32(), C.c_uint32(), C.c_int64(), C.c_int64(), C.c_int32(), F.Scalar(), F.Scalar(), C.c_int64()
            F.check(F.lib().plx_parquet_chunk_info(h, g, i, C.byref(codec), C.byref(enc), C.byref(cb), C.byref(ub), C.byref(has), C.byref(mn), C.byref(mx), C.byref(nc)))
            assert cb.value == col.total_compressed_size and ub.value == col.total_uncompressed_size
            assert {"UNCOMPRESSED": 0, "SNAPPY": 1}[col.compression] == codec.value
            enc_bits = {"PLAIN": 0, "PLAIN_DICTIONARY": 2, "RLE": 3, "BIT_PACKED": 4, "RLE_DICTIONARY": 8}
            assert enc.value == sum({1 << enc_bits[e] for e in col.encodings}), (name, col.encodings)
            st = col.statistics
            assert nc.value == (st.null_count if st is not None and st.has_null_count else -1)
            if name not in want or want[name][0] < 0 or name in ("s", "bin"):
                assert has.value == 0
                continue
            assert has.value == 1, name
            lo, hi = st.min, st.max
            if name == "date":
                lo, hi = (lo - dt.date(1970, 1, 1)).days, (hi - dt.date(1970, 1, 1)).days
            if name in ("ts", "ts_ms"):              # statistics come in the column's own unit
                us = lambda d: (d.replace(tzinfo=None) - dt.datetime(1970, 1, 1)) // dt.timedelta(microseconds=1 if name == "ts" else 1000)
                lo, hi = us(lo), us(hi)
            pick = {"f64": lambda s: s.f64, "f32": lambda s: s.f32, "b": lambda s: bool(s.u)}.get(name, (lambda s: s.u) if name.startswith("u") else (lambda s: s.i))
            assert pick(mn) == lo and pick(mx) == hi, (name, pick(mn), lo)
    F.check(F.lib().plx_parquet_close(h))
    assert F.lib().plx_parquet_shape(h, None, None, None) != 0 and "invalid parquet handle" in F.lib().plx_last_error().decode()


def test_v2_pages_old_converted_types_and_required_columns(tmp_path):
    """Files the way other writers shape them: format version 1.0 (ConvertedType annotations only), required (non-nullable) fields."""
    n = 500
    schema = pa.schema([pa.field("k", pa.int64(), nullable=False), pa.field("u8", pa.uint8(), nullable=False), pa.field("s", pa.string()),
                        pa.field("ts", pa.timestamp("us")), pa.field("d", pa.date32())])
    t = pa.table({"k": np.arange(n), "u8": np.arange(n).astype(np.uint8), "s": ["v"] * n, "ts": pa.array(np.arange(n), pa.timestamp("us")),
                  "d": pa.array(np.arange(n).astype(np.int32), pa.date32())}, schema=schema)
    path = str(tmp_path / "old.parquet")
    pq.write_table(t, path, version="1.0", data_page_version="2.0")
    src = io.ParquetFrame(path)
    assert src.decoder == "device"
    assert src.schema == {"k": pl.Int64, "u8": pl.UInt8, "s": pl.Categorical([]), "ts": pl.Datetime, "d": pl.Date}
    h = lib_meta(path)
    nl = C.c_int32()
    F.check(F.lib().plx_parquet_column_info(h, 0, None, None, None, C.byref(nl)))
    assert nl.value == 0
    F.check(F.lib().plx_parquet_column_info(h, 2, None, None, None, C.byref(nl)))
    assert nl.value == 1


def test_not_a_parquet_file_and_missing_file(tmp_path):
    p = tmp_path / "junk.parquet"
    p.write_bytes(b"PAR1" + b"\x00" * 100 + b"NOPE")
    h = C.c_uint64()
    assert F.lib().plx_parquet_open(str(p).encode(), C.byref(h)) == 1 and "PAR1" in F.lib().plx_last_error().decode()
    assert F.lib().plx_parquet_open(str(tmp_path / "absent.parquet").encode(), C.byref(h)) == 1 and "cannot open" in F.lib().plx_last_error().decode()
    p.write_bytes(b"PAR1" + b"\x15\x00" * 20 + (40).to_bytes(4, "little") + b"PAR1")       # a footer that is not a FileMetaData struct
    assert F.lib().plx_parquet_open(str(p).encode(), C.byref(h)) == 1
    with pytest.raises(pl.PlxError):
        io.ParquetFrame(str(p))


def test_read_needs_a_gpu_and_says_so(tmp_path):
    """No host decode path in the library: without a bound device plx_parquet_read fails with PLX_ERR_HIP (this container has no GPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_gpu_parquet.py")
    path = str(tmp_path / "t.parquet")
    pq.write_table(pa.table({"a": np.arange(10)}), path)
    h = lib_meta(path)
    rg, col, fh = (C.c_int32 * 1)(0), (C.c_int32 * 1)(0), C.c_uint64()
    assert F.lib().plx_parquet_read(h, rg, 1, col, 1, C.byref(fh)) == 2          # PLX_ERR_HIP
    assert "GPU" in F.lib().plx_last_error().decode()


def test_device_and_host_decoder_plan_the_same_scan(tmp_path):
    """Projection / row-group pruning decisions are the same whichever decoder supplies the statistics."""
    n = 20_000
    day = np.sort(RNG.integers(0, 3000, n)).astype(np.int32)
    t = pa.table({"d": pa.array(day, pa.date32()), "ts": pa.array(day.astype(np.int64) * 86_400_000_000, pa.timestamp("us")), "x": pa.array(RNG.normal(size=n)),
                  "k": pa.array(np.arange(n)), "u": pa.array(np.arange(n).astype(np.uint32)), "s": pa.array(["a"] * n)})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=1000)
    c = pl.col
    preds = [c("d") >= dt.date(1975, 1, 1), (c("ts") < dt.datetime(1972, 6, 1)) & (c("k") > 100), c("u") == 5000, c("k") != 7, c("x") < -10.0, c("x") > 0.0,
             (c("d") > dt.date(1990, 1, 1)), dt.datetime(1973, 1, 1) <= c("ts"), c("ts") >= dt.date(1976, 2, 3)]
    for p in preds:
        got = {}
        for dec in ("device", "host"):
            lf = pl.scan_parquet(path, decoder=dec).filter(p).select(c("k").sum())
            io.reset_scans(lf._node); io.push_down(lf._node)
            node = lf._node
            while node.kind != "scan":
                node = node.input
            got[dec] = (sorted(node.frame.selected_columns()), node.frame.selected_row_groups())
        assert got["device"] == got["host"], p
        assert len(got["device"][1]) <= 20
