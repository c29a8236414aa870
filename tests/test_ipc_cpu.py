"""plx_ipc_* without a GPU: the library's own Arrow IPC metadata reader (polars_amd/csrc/ipc_format.hpp: FlatBuffers footer, schema,
record-batch and dictionary messages) against pyarrow's view of the same files; the host half of the reader under AddressSanitizer
over corrupted files; the loud failure of plx_ipc_read when no device is bound."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

import polars_amd as pl
from polars_amd import _ffi as F
from polars_amd import ipc_io

RNG = np.random.default_rng(9)


def sample(n):
    m = RNG.random(n) < 0.2
    words = np.array(["", "a", "BUILDING", "a much longer string that does not fit in twelve bytes", "ünï"])
    return pa.table({
        "i8": pa.array(RNG.integers(-100, 100, n).astype(np.int8)), "u16": pa.array(RNG.integers(0, 60000, n).astype(np.uint16), mask=m),
        "i32": pa.array(RNG.integers(-10**9, 10**9, n).astype(np.int32)), "u32": pa.array(RNG.integers(0, 2**32, n).astype(np.uint32)),
        "i64": pa.array(RNG.integers(-10**15, 10**15, n), mask=m), "u64": pa.array(RNG.integers(0, 2**63, n).astype(np.uint64)),
        "f32": pa.array(RNG.normal(size=n).astype(np.float32)), "f64": pa.array(RNG.normal(size=n), mask=m), "b": pa.array(RNG.random(n) < 0.5, mask=m),
        "date": pa.array(RNG.integers(0, 20000, n).astype(np.int32), pa.date32()), "ts": pa.array(RNG.integers(0, 2**50, n), pa.timestamp("us")),
        "s": pa.array(words[RNG.integers(0, 5, n)], mask=m), "ls": pa.array(words[RNG.integers(0, 5, n)], pa.large_string()),
        "sv": pa.array(words[RNG.integers(0, 5, n)], pa.string_view()), "bin": pa.array([b"\x00\x01"] * n, pa.binary()),
        "d8": pa.array(words[RNG.integers(0, 5, n)]).dictionary_encode().cast(pa.dictionary(pa.int8(), pa.string())),
        "d32": pa.array(words[RNG.integers(0, 3, n)], mask=m).dictionary_encode(),
        "ts_ms": pa.array(RNG.integers(0, 2**40, n), pa.timestamp("ms", "Europe/Vienna")),
        # outside the hot path
        "dec": pa.array([None] * n, pa.decimal128(12, 2)), "lst": pa.array([[1, 2]] * n),
        "st": pa.array([{"p": 1, "q": "z"}] * n), "tail": pa.array(np.arange(n)),
    })


def write(path, t, chunk=None, **opts):
    with ipc.new_file(path, t.schema, options=ipc.IpcWriteOptions(**opts)) as w:
        for b in t.to_batches(max_chunksize=chunk):
            w.write_batch(b)


def test_schema_batches_and_dictionaries_match_pyarrow(tmp_path):
    n = 2500
    t = sample(n)
    path = str(tmp_path / "t.arrow")
    write(path, t, chunk=700)
    rd = ipc.open_file(path)
    src = ipc_io._IpcDecoder(path)
    assert src.num_rows == n and src.num_row_groups == rd.num_record_batches == 4 and src.names == t.column_names
    want = {"i8": pl.Int8, "u16": pl.UInt16, "i32": pl.Int32, "u32": pl.UInt32, "i64": pl.Int64, "u64": pl.UInt64, "f32": pl.Float32, "f64": pl.Float64,
            "b": pl.Boolean, "date": pl.Date, "ts": pl.Datetime, "tail": pl.Int64}
    for name, dt in want.items():
        assert src.dtype(name) == dt and src.dtype(name).physical == dt.physical, name
    for name in ("s", "ls", "sv", "bin", "d8", "d32"):
        assert isinstance(src.dtype(name), pl.Categorical)
    assert src.dtype("ts_ms").time_unit == "ms" and src.dtype("ts").time_unit == "us" and src.dtype("ts_ms") == pl.Datetime       # the file's unit is kept
    assert src.dtype("ts_ms").time_zone == "Europe/Vienna" and src.dtype("ts").time_zone is None                                     # ... and its zone
    assert src.dtype("s").from_strings and src.dtype("sv").from_strings and not src.dtype("d8").from_strings      # plain strings vs dictionaries in the file
    for name in ("dec", "lst", "st"):
        with pytest.raises(TypeError):
            src.dtype(name)
    assert src._info["bin"][2] == 4 and src._info["s"][2] == 3 and src._info["i8"][3] is True
    for b in range(4):
        info = src.batch_info(b)
        assert info["rows"] == rd.get_batch(b).num_rows and info["compression"] is None and info["body_bytes"] > 0
    # dictionaries of the dictionary-encoded columns: the values the file holds, in the file's order (pyarrow unifies across batches when writing)
    for name in ("d8", "d32"):
        col = rd.read_all().column(name).combine_chunks()
        assert src.categories(name) == col.dictionary.to_pylist()
    with pytest.raises(pl.PlxError):
        src.categories("s")             # not dictionary-encoded in the file: its dictionary is built on the device at read time


def test_compressed_files_are_recognised(tmp_path):
    t = pa.table({"a": np.arange(1000)})
    for codec, name in (("lz4", "lz4"), ("zstd", "zstd")):
        path = str(tmp_path / f"{codec}.arrow")
        write(path, t, compression=codec)
        src = ipc_io._IpcDecoder(path)
        assert src.batch_info(0)["compression"] == name and src.num_rows == 1000


def test_not_an_ipc_file(tmp_path):
    p = tmp_path / "junk.arrow"
    p.write_bytes(b"ARROW1\x00\x00" + b"\x00" * 64 + b"NOPE!!")
    h = C.c_uint64()
    assert F.lib().plx_ipc_open(str(p).encode(), C.byref(h)) == 1 and "ARROW1" in F.lib().plx_last_error().decode()
    assert F.lib().plx_ipc_open(str(tmp_path / "absent").encode(), C.byref(h)) == 1 and "cannot open" in F.lib().plx_last_error().decode()
    # a stream-format file (no footer) is not a file-format file
    sink = str(tmp_path / "stream.arrows")
    t = pa.table({"a": np.arange(10)})
    with ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    assert F.lib().plx_ipc_open(sink.encode(), C.byref(h)) == 1


def test_read_needs_a_gpu_and_says_so(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_gpu_ipc.py")
    path = str(tmp_path / "t.arrow")
    write(path, pa.table({"a": np.arange(10)}))
    h = C.c_uint64()
    F.check(F.lib().plx_ipc_open(path.encode(), C.byref(h)))
    b, col, fh = (C.c_int32 * 1)(0), (C.c_int32 * 1)(0), C.c_uint64()
    assert F.lib().plx_ipc_read(h.value, b, 1, col, 1, C.byref(fh)) == 2          # PLX_ERR_HIP
    assert "GPU" in F.lib().plx_last_error().decode()


def test_projection_reaches_the_ipc_scan(tmp_path):
    n = 1000
    t = pa.table({"k": np.arange(n) % 7, "v": RNG.normal(size=n), "w": RNG.normal(size=n), "s": pa.array(["x"] * n)})
    path = str(tmp_path / "t.arrow")
    write(path, t, chunk=300)
    from polars_amd import io
    c = pl.col
    lf = pl.scan_ipc(path).filter(c("v") > 0).group_by("k").agg(c("v").sum())
    io.reset_scans(lf._node); io.push_down(lf._node)
    node = lf._node
    while node.kind != "scan":
        node = node.input
    assert sorted(node.frame.selected_columns()) == ["k", "v"] and node.frame.selected_row_groups() == [0, 1, 2, 3]


def test_host_reader_under_address_sanitizer(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "ipc_asan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(here, "emu", "ipc_meta_main.cpp"), "-lpthread"], check=True)
    t = sample(800)
    files = []
    for i, (chunk, codec) in enumerate(((None, None), (150, None), (300, "lz4"), (300, "zstd"))):
        p = str(tmp_path / f"good{i}.arrow")
        write(p, t, chunk=chunk, **({"compression": codec} if codec else {}))
        files.append(p)
        raw = open(p, "rb").read()
        flen = struct.unpack("<i", raw[-10:-6])[0]
        for k in range(80):
            b = bytearray(raw)
            lo, hi = (len(b) - 10 - flen, len(b) - 6) if k % 2 else (8, len(b) - 10 - flen)        # footer bytes or message / body bytes
            for _ in range(1 + k % 4):
                b[int(RNG.integers(lo, hi))] ^= 1 << int(RNG.integers(0, 8))
            q = str(tmp_path / f"bad{i}_{k}.arrow")
            open(q, "wb").write(b if k % 9 else b[:int(RNG.integers(16, len(b)))])
            files.append(q)
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    counts = dict(kv.split("=") for kv in r.stdout.split())
    assert int(counts["ok"]) >= 4 and int(counts["invalid"]) > 20 and int(counts["strings"]) > 0, r.stdout


def _ipc_emu():
    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "emu", "libipc_emu.so"), os.path.join(here, "emu", "ipc_emu.cpp")
    csrc = os.path.join(os.path.dirname(here), "polars_amd", "csrc")
    deps = [src] + [os.path.join(csrc, h) for h in ("ipc_reader.hpp", "ipc_format.hpp", "host_codecs.hpp", "file_io.hpp")]
    import parquet_emu
    parquet_emu.build_if_stale(so, src, deps)
    l = C.CDLL(so)
    l.ipcemu_buffer.restype = C.c_int64
    l.ipcemu_buffer.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    l.ipcemu_last_error.restype = C.c_char_p
    return l


def _check_buffers(path, names):
    """every buffer of the named columns in every record batch, as the host half hands it to the upload, against pyarrow's; returns the count"""
    l = _ipc_emu()
    rd = ipc.open_file(path)
    checked = 0
    for b in range(rd.num_record_batches):
        batch = rd.get_batch(b)
        for ci, name in enumerate(names):
            arr = batch.column(ci)
            arr = arr.indices if pa.types.is_dictionary(arr.type) else arr
            for which, buf in enumerate(arr.buffers()):
                if buf is None or which > 2 or (pa.types.is_string_view(arr.type) and which >= 2 and buf.size == 0):
                    continue
                out = np.zeros(buf.size + 64, np.uint8)
                got = l.ipcemu_buffer(path.encode(), b, ci, which, out.ctypes.data_as(C.c_void_p), out.nbytes)
                assert got >= 0, l.ipcemu_last_error().decode()
                want = np.frombuffer(buf, np.uint8)
                if which == 0:          # validity: the file may pad differently; compare the bits that matter
                    nb = (len(arr) + 7) // 8
                    w = np.unpackbits(want[:nb], bitorder="little")[arr.offset:arr.offset + len(arr)]
                    g = np.unpackbits(out[:nb], bitorder="little")[:len(arr)]
                    assert np.array_equal(w, g), (name, b)
                else:
                    k = min(got, len(want))
                    assert k > 0 and np.array_equal(out[:k], want[:k]), (name, b, which)
                checked += 1
    return checked


@pytest.mark.parametrize("codec", [None, "lz4", "zstd"])
def test_buffers_as_uploaded_match_pyarrow(tmp_path, codec):
    """Every buffer of every hot-path column in every record batch, as the product's host half hands it to the upload (body compression
    undone by host_codecs.hpp: LZ4 frames are what pyarrow's feather writer produces by default), equals pyarrow's buffer."""
    n = 3000
    t = sample(n).select(["i8", "u16", "i32", "i64", "f32", "f64", "b", "date", "ts", "s", "ls", "sv", "d8", "d32", "tail"])
    path = str(tmp_path / "t.arrow")
    write(path, t, chunk=1100, **({"compression": codec} if codec else {}))
    assert _check_buffers(path, t.column_names) > 60


PDS_HEADS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pds_heads")


@pytest.mark.parametrize("name", ["lineitem", "orders", "customer"])
def test_files_written_by_the_reference(name):
    """tests/golden/pds_heads/*.feather are the reference's own TPC-H sample tables, written by ITS IPC writer (copied by
    tests/golden/make_pds_heads.py): the library's metadata reader agrees with pyarrow on them, every column maps to
    a hot-path dtype (int64, double, timestamp[us], large_string), and the host half delivers every buffer as pyarrow sees it."""
    import hashlib
    import json
    path = os.path.join(PDS_HEADS, name + ".feather")
    want_sum = json.load(open(os.path.join(PDS_HEADS, "SHA256.json")))["sha256"][name + ".feather"]
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want_sum
    t = ipc.open_file(path).read_all()
    src = ipc_io.IpcFrame(path)
    assert src.num_rows == t.num_rows == 10 and list(src.schema) == t.column_names
    for f in t.schema:
        got = src.schema[f.name]
        want = "Categorical" if pa.types.is_large_string(f.type) else "Datetime" if pa.types.is_timestamp(f.type) else {"int64": "Int64", "double": "Float64"}[str(f.type)]
        assert got.name == want, (f.name, got, f.type)
    assert _check_buffers(path, t.column_names) >= t.num_columns + sum(pa.types.is_large_string(f.type) for f in t.schema)      # no nulls in these files: values (+ data) buffers only


@pytest.mark.parametrize("name", ["foods1.ipc", "foods2.ipc"])
def test_the_reference_s_own_ipc_fixtures(name):
    """py-polars/tests/unit/io/files/foods{1,2}.ipc (copied by tests/golden/make_io_files.py): written by the reference's IPC writer."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_files", name)
    t = ipc.open_file(path).read_all()
    src = ipc_io.IpcFrame(path)
    assert src.num_rows == t.num_rows == 27 and list(src.schema) == t.column_names == ["category", "calories", "fats_g", "sugars_g"]
    assert src.schema["category"].from_strings and src.schema["fats_g"] == pl.Float64
    assert _check_buffers(path, t.column_names) >= 5
