"""The device Parquet decoder on the GPU (plx_parquet_read through polars_amd.read_parquet / scan_parquet): every dtype of the hot path,
both codecs, both data-page versions, dictionary on / off, nulls, row-group subsets, against pyarrow's decode of the same file; the
CPU harness (tests/test_parquet_emu_cpu.py) pins the same kernel bodies thread by thread, here they run as wavefronts."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(77)


def table(n, null_frac=0.2):
    m = lambda: RNG.random(n) < null_frac
    words = np.array(["", "a", "bb", "BUILDING", "AUTOMOBILE", "a much longer string that does not fit in twelve bytes", "ünïcödé"])
    return pa.table({
        "i8": pa.array(RNG.integers(-128, 128, n).astype(np.int8)), "i16": pa.array(RNG.integers(-30000, 30000, n).astype(np.int16), mask=m()),
        "i32": pa.array(RNG.integers(-2**31, 2**31, n).astype(np.int32), mask=m()), "i64": pa.array(RNG.integers(-2**62, 2**62, n), mask=m()),
        "u8": pa.array(RNG.integers(0, 256, n).astype(np.uint8), mask=m()), "u16": pa.array(RNG.integers(0, 65536, n).astype(np.uint16)),
        "u32": pa.array(RNG.integers(0, 2**32, n).astype(np.uint32), mask=m()), "u64": pa.array(RNG.integers(0, 2**63, n).astype(np.uint64) * 2 + 1, mask=m()),
        "f32": pa.array(RNG.normal(size=n).astype(np.float32), mask=m()), "f64": pa.array(np.where(RNG.random(n) < 0.05, np.nan, RNG.normal(size=n)), mask=m()),
        "b": pa.array(RNG.random(n) < 0.5, mask=m()), "b_req": pa.array(RNG.random(n) < 0.1),
        "date": pa.array(RNG.integers(0, 20000, n).astype(np.int32), pa.date32(), mask=m()), "ts": pa.array(RNG.integers(0, 2**50, n), pa.timestamp("us"), mask=m()),
        "s": pa.array(words[RNG.integers(0, len(words), n)], mask=m()), "low_card": pa.array(RNG.integers(0, 3, n)),
        "const": pa.array(np.full(n, 42, np.int64)), "all_null": pa.array(np.zeros(n, np.int64), mask=np.ones(n, bool)),
        "runs": pa.array(np.repeat(RNG.integers(0, 5, (n + 99) // 100), 100)[:n], mask=np.repeat(RNG.random((n + 49) // 50) < 0.3, 50)[:n]),
        "period7": pa.array(np.tile(np.arange(7, dtype=np.int64) * 1_000_003, n // 7 + 1)[:n]),
    })


def compare(df, want_table, names):
    for name in names:
        want = want_table.column(name).combine_chunks()
        s = df[name]
        n = len(want)
        assert len(s) == n, name
        values, valid = s._download()
        wvalid = np.array([v is not None for v in want.to_pylist()], bool) if want.null_count else np.ones(n, bool)
        assert s.null_count() == want.null_count, name
        if want.null_count:
            assert valid is not None and np.array_equal(valid, wvalid), name
        else:
            assert valid is None, name
        t = want.type
        if pa.types.is_string(t) or pa.types.is_large_string(t):
            assert s.to_list() == want.to_pylist(), name
            continue
        if pa.types.is_timestamp(t) or pa.types.is_date32(t):
            want = want.cast(pa.int64() if pa.types.is_timestamp(t) else pa.int32())
        if pa.types.is_floating(t):
            e = np.asarray(want.to_numpy(zero_copy_only=False), dtype=values.dtype)
            u = np.uint32 if values.dtype == np.float32 else np.uint64
            assert np.array_equal(values[wvalid].view(u), e[wvalid].view(u)), name
        else:
            e = np.fromiter((x if x is not None else 0 for x in want.to_pylist()), dtype=values.dtype, count=n)
            assert np.array_equal(values[wvalid], e[wvalid]), name
        assert np.all(values[~wvalid] == 0), name


def decode_and_compare(pl, tmp_path, compression, version, dictionary):
    n = 20_000
    t = table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression=compression, data_page_version=version, use_dictionary=dictionary, row_group_size=6500, data_page_size=4096)
    names = t.column_names       # without a dictionary the string column has PLAIN pages: views from host threads, dictionary built on the device
    df = pl.read_parquet(path, columns=names)
    assert df.columns == names and df.height == n
    compare(df, pq.read_table(path), names)
    if not dictionary:
        assert pl.read_parquet(path, columns=["s"], decoder="host")["s"].to_list() == df["s"].to_list()


@pytest.mark.parametrize("compression", ["none", "snappy"])
@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_device_decode_matches_pyarrow(pl, tmp_path, compression, version):
    """Everything decoded by kernels: Snappy, levels, dictionary and PLAIN values.  (Pages of other codecs and string columns without a
    dictionary take host threads first: tests/test_gpu_zzz_scan_host_paths.py.)"""
    decode_and_compare(pl, tmp_path, compression, version, True)


def test_many_pages_many_row_groups_and_subsets(pl, tmp_path):
    """2e6 rows in 16 row groups of ~125 pages per column: thousands of Snappy wavefronts and run tables in one launch each."""
    n = 2_000_000
    v = RNG.integers(0, 1 << 40, n)
    valid = RNG.random(n) > 0.1
    t = pa.table({"k": pa.array(np.arange(n)), "v": pa.array(v, mask=~valid), "d": pa.array(v % 1000, mask=~valid), "f": pa.array(RNG.normal(size=n)),
                  "s": pa.array(np.array(["R", "A", "N"])[v % 3]), "z": pa.array(np.zeros(n, np.int64)), "p": pa.array(np.arange(n, dtype=np.int32) // 16)})
    path = str(tmp_path / "big.parquet")
    pq.write_table(t, path, compression="snappy", row_group_size=125_000, data_page_size=8192, use_dictionary=["d", "s"])
    src = pl.scan_parquet(path)._node.frame
    assert src.decoder == "device" and src.num_row_groups == 16
    df = pl.read_parquet(path)
    compare(df, t, t.column_names)
    assert df["k"].sum() == n * (n - 1) // 2 and df["v"].null_count() == int((~valid).sum())
    # pruned scan: only the row groups whose statistics can match, only the columns the plan touches
    c = pl.col
    lf = pl.scan_parquet(path).filter((c("k") >= 1_300_000) & (c("k") < 1_400_000)).group_by("s").agg(c("v").sum().alias("sv"), pl.len().alias("n"))
    out = lf.collect().sort_host("s")
    node = lf._node
    while node.kind != "scan":
        node = node.input
    read = node.frame.last_read
    assert read["decoder"] == "device" and sorted(read["columns"]) == ["k", "s", "v"] and read["row_groups"] <= 2
    sel = (np.arange(n) >= 1_300_000) & (np.arange(n) < 1_400_000)
    for i, key in enumerate(out["s"]):
        mk = sel & (np.array(["R", "A", "N"])[v % 3] == key)
        assert out["n"][i] == int(mk.sum()) and out["sv"][i] == int(v[mk & valid].sum())


def test_unsupported_and_corrupt_files_are_status_codes(pl, tmp_path):
    n = 5000
    t = pa.table({"a": pa.array(RNG.integers(0, 9, n), mask=RNG.random(n) < 0.3), "z": pa.array(np.arange(n))})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression={"a": "snappy", "z": "brotli"})
    with pytest.raises(pl.UnsupportedError) as ei:
        pl.read_parquet(path, columns=["z"])
    assert "BROTLI" in str(ei.value)
    compare(pl.read_parquet(path, columns=["a"]), t, ["a"])
    # flip bits inside the page bytes: an error (or a well-formed different value), never a hang or a crash; the library stays usable
    raw = bytearray(open(path, "rb").read())
    bad = str(tmp_path / "bad.parquet")
    seen = set()
    for trial in range(25):
        b = bytearray(raw)
        for _ in range(3):
            b[int(RNG.integers(4, 4000))] ^= 1 << int(RNG.integers(0, 8))
        open(bad, "wb").write(b)
        try:
            pl.read_parquet(bad, columns=["a"])
            seen.add("ok")
        except pl.PlxError as e:
            seen.add("error")
            assert e.code in (1, 3)
    assert "error" in seen
    compare(pl.read_parquet(path, columns=["a"]), t, ["a"])


def test_q1_from_parquet_both_decoders(pl, orc, tmp_path):
    from polars_amd import datagen, queries
    n = 300_000
    li = datagen.lineitem_host(n, seed=8)
    t = pa.table({"l_quantity": pa.array(li["l_quantity"]), "l_extendedprice": pa.array(li["l_extendedprice"]), "l_discount": pa.array(li["l_discount"]),
                  "l_tax": pa.array(li["l_tax"]), "l_returnflag": pa.array([datagen.FLAGS[c] for c in li["l_returnflag"]]),
                  "l_linestatus": pa.array([datagen.STATUS[c] for c in li["l_linestatus"]]), "l_shipdate": pa.array(li["l_shipdate"], pa.timestamp("us")),
                  "l_comment": pa.array(["x"] * n)})
    path = str(tmp_path / "lineitem.parquet")
    pq.write_table(t, path, row_group_size=50_000, compression="snappy")
    want = orc.q1(li, datagen.us(1998, 9, 2))
    for decoder in ("device", "host"):
        out = queries.q1(pl.scan_parquet(path, decoder=decoder)).collect().sort_host(["l_returnflag", "l_linestatus"])
        assert [datagen.FLAGS.index(x) for x in out["l_returnflag"]] == want["l_returnflag"].tolist(), decoder
        assert out["count_order"] == want["count_order"].tolist() and out["sum_qty"] == want["sum_qty"].tolist(), decoder
        for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert np.allclose(np.array(out[c]), want[c], rtol=1e-6, atol=0), (decoder, c)
