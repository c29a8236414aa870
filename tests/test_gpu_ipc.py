"""Arrow IPC file scan on the GPU (plx_ipc_read through polars_amd.read_ipc / scan_ipc) against pyarrow's read of the same file: every
dtype of the hot path, record batches whose row counts are not multiples of 8 or 64 (bitmaps concatenated at arbitrary bit offsets),
nulls, dictionary-encoded strings with narrow indices, Utf8 / LargeUtf8 / Utf8View columns encoded on the device, batch subsets."""
import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(31)


def table(n):
    m = lambda: RNG.random(n) < 0.2
    words = np.array(["", "a", "BUILDING", "a much longer string that does not fit in twelve bytes", "ünï", "another long one, different from the first"])
    return pa.table({
        "i8": pa.array(RNG.integers(-100, 100, n).astype(np.int8), mask=m()), "u16": pa.array(RNG.integers(0, 60000, n).astype(np.uint16)),
        "i32": pa.array(RNG.integers(-10**9, 10**9, n).astype(np.int32), mask=m()), "u32": pa.array(RNG.integers(0, 2**32, n).astype(np.uint32)),
        "i64": pa.array(RNG.integers(-10**15, 10**15, n), mask=m()), "u64": pa.array(RNG.integers(0, 2**63, n).astype(np.uint64) * 2 + 1),
        "f32": pa.array(RNG.normal(size=n).astype(np.float32)), "f64": pa.array(np.where(RNG.random(n) < 0.05, np.nan, RNG.normal(size=n)), mask=m()),
        "b": pa.array(RNG.random(n) < 0.5, mask=m()), "b_req": pa.array(RNG.random(n) < 0.1),
        "date": pa.array(RNG.integers(0, 20000, n).astype(np.int32), pa.date32(), mask=m()), "ts": pa.array(RNG.integers(0, 2**50, n), pa.timestamp("us")),
        "s": pa.array(words[RNG.integers(0, 6, n)], mask=m()), "ls": pa.array(words[RNG.integers(0, 6, n)], pa.large_string()),
        "sv": pa.array(words[RNG.integers(0, 6, n)], pa.string_view(), mask=m()),
        "d8": pa.array(words[RNG.integers(0, 6, n)]).dictionary_encode().cast(pa.dictionary(pa.int8(), pa.string())),
        "d32": pa.array(words[RNG.integers(0, 3, n)], mask=m()).dictionary_encode(),
        "some_nulls_late": pa.array(np.arange(n), mask=np.arange(n) > n - 50),        # only the last batch has nulls
    })


def write(path, t, chunk=None, **opts):
    with ipc.new_file(path, t.schema, options=ipc.IpcWriteOptions(**opts)) as w:
        for b in t.to_batches(max_chunksize=chunk):
            w.write_batch(b)


def compare(df, want_table, names):
    for name in names:
        want = want_table.column(name).combine_chunks()
        s = df[name]
        n = len(want)
        assert len(s) == n, name
        values, valid = s._download()
        wl = want.to_pylist()
        wvalid = np.array([v is not None for v in wl], bool)
        assert s.null_count() == want.null_count, name
        if want.null_count:
            assert valid is not None and np.array_equal(valid, wvalid), name
        else:
            assert valid is None, name
        t = want.type
        if pa.types.is_dictionary(t) or pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_string_view(t):
            assert s.to_list() == wl, name
            continue
        if pa.types.is_timestamp(t) or pa.types.is_date32(t):
            want = want.cast(pa.int64() if pa.types.is_timestamp(t) else pa.int32())
            wl = want.to_pylist()
        if pa.types.is_floating(t):
            e = np.asarray(want.to_numpy(zero_copy_only=False), dtype=values.dtype)
            u = np.uint32 if values.dtype == np.float32 else np.uint64
            assert np.array_equal(values[wvalid].view(u), e[wvalid].view(u)), name
        else:
            e = np.fromiter((x if x is not None else 0 for x in wl), dtype=values.dtype, count=n)
            assert np.array_equal(values[wvalid], e[wvalid]), name


@pytest.mark.parametrize("chunk", [None, 1000, 333])
def test_ipc_read_matches_pyarrow(pl, tmp_path, chunk):
    n = 5003
    t = table(n)
    path = str(tmp_path / "t.arrow")
    write(path, t, chunk=chunk)
    df = pl.read_ipc(path)
    assert df.columns == t.column_names and df.height == n
    compare(df, t, t.column_names)


def test_batch_subsets_and_lazy_scan(pl, tmp_path):
    n = 4000
    t = table(n)
    path = str(tmp_path / "t.arrow")
    write(path, t, chunk=777)
    from polars_amd import ipc_io
    src = ipc_io.IpcFrame(path, columns=["i64", "s", "b", "d8"])
    df, rows, _ = src._dec.read([4, 1], ["i64", "s", "b", "d8"])
    want = pa.concat_tables([pa.Table.from_batches([ipc.open_file(path).get_batch(b)]) for b in (4, 1)])
    assert rows == want.num_rows
    compare(df, want, ["i64", "s", "b", "d8"])
    # a query straight from the file: only the projected columns are read
    c = pl.col
    lf = pl.scan_ipc(path).filter(c("u16") > 30000).group_by("d8").agg(c("u32").sum().alias("su"), pl.len().alias("n"))
    out = lf.collect().sort_host("d8")
    node = lf._node
    while node.kind != "scan":
        node = node.input
    assert sorted(node.frame.last_read["columns"]) == ["d8", "u16", "u32"]
    u16 = t.column("u16").to_numpy(); u32 = t.column("u32").to_numpy().astype(np.uint64); d8 = np.array(t.column("d8").to_pylist())
    for i, key in enumerate(out["d8"]):
        mk = (u16 > 30000) & (d8 == key)
        assert out["n"][i] == int(mk.sum()) and out["su"][i] == int(u32[mk].sum()) & 0xffffffff       # a UInt32 sum stays UInt32 (wrapping), as in Polars


def test_unsupported_ipc_files_are_status_codes(pl, tmp_path):
    t = pa.table({"a": np.arange(1000), "l": pa.array([[1]] * 1000), "ms": pa.array(np.arange(1000), pa.timestamp("s"))})
    path2 = str(tmp_path / "nested.arrow")
    write(path2, t)
    with pytest.raises(TypeError):
        pl.read_ipc(path2)                       # the mirror refuses the nested column when building the schema ...
    assert pl.read_ipc(path2, columns=["a"])["a"].sum() == 999 * 1000 // 2      # ... its neighbours are readable
    import ctypes as C
    F = pl._ffi
    h, fh = C.c_uint64(), C.c_uint64()
    F.check(F.lib().plx_ipc_open(path2.encode(), C.byref(h)))
    b, cols = (C.c_int32 * 1)(0), (C.c_int32 * 1)(2)
    assert F.lib().plx_ipc_read(h.value, b, 1, cols, 1, C.byref(fh)) == 3 and "timestamp in seconds" in F.lib().plx_last_error().decode()
