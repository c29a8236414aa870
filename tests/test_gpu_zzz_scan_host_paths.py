"""Scan paths whose newest code is HOST code feeding device paths that had already run on hardware: pages of host-inflated codecs
(zstd; host_codecs.hpp), string columns without a dictionary (host threads assemble Utf8Views, the device dictionary encoder takes
over), whole-column host decodes (DELTA_*, BYTE_STREAM_SPLIT, INT96), compressed Arrow IPC bodies, scans over several files
(plx_frame_concat + dictionary unification), and the reference's own TPC-H sample files (tests/golden/pds_heads).  Their host halves
are pinned on the CPU (tests/test_parquet_emu_cpu.py, tests/test_ipc_cpu.py, tests/test_polars_engine_cpu.py); they were written
after round 2's GPU minutes were spent, so this file sorts last: a surprise here cannot hide the rest of the suite behind `-x`."""
import datetime as dt
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pyarrow.parquet as pq
import pytest

import test_gpu_ipc as I
from test_gpu_parquet import compare, decode_and_compare

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(78)
PDS_HEADS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pds_heads")


@pytest.mark.parametrize("compression,version,dictionary", [("zstd", "1.0", True), ("zstd", "2.0", True), ("zstd", "1.0", False), ("none", "1.0", False), ("snappy", "1.0", False)])
def test_host_inflated_pages_and_plain_strings(pl, tmp_path, compression, version, dictionary):
    decode_and_compare(pl, tmp_path, compression, version, dictionary)


def test_host_decoded_encodings_arrive_on_the_device(pl, tmp_path):
    """DELTA_BINARY_PACKED / BYTE_STREAM_SPLIT / DELTA_*_BYTE_ARRAY / INT96 columns: decoded by the library's host threads
    (parquet_reader.hpp: read_fixed_column_host, read_string_column_host), uploaded as finished columns / views."""
    n = 30_000
    words = np.array(["", "a", "prefix-shared-0001", "prefix-shared-0002", "prefix-shared-and-longer-0003", "zebra"])
    m = lambda: RNG.random(n) < 0.2
    t = pa.table({"d64": pa.array(np.cumsum(RNG.integers(-5, 50, n)), mask=m()), "d32": pa.array(RNG.integers(-2**31, 2**31, n).astype(np.int32)),
                  "bss64": pa.array(RNG.normal(size=n), mask=m()), "bss32": pa.array(RNG.normal(size=n).astype(np.float32)),
                  "dl": pa.array(words[RNG.integers(0, len(words), n)], mask=m()), "dba": pa.array(np.sort(words[RNG.integers(0, len(words), n)])),
                  "plain_i64": pa.array(RNG.integers(0, 1 << 40, n))})
    enc = {"d64": "DELTA_BINARY_PACKED", "d32": "DELTA_BINARY_PACKED", "bss64": "BYTE_STREAM_SPLIT", "bss32": "BYTE_STREAM_SPLIT", "dl": "DELTA_LENGTH_BYTE_ARRAY",
           "dba": "DELTA_BYTE_ARRAY", "plain_i64": "PLAIN"}
    path = str(tmp_path / "v2.parquet")
    pq.write_table(t, path, compression="zstd", data_page_version="2.0", use_dictionary=False, column_encoding=enc, row_group_size=11_000, data_page_size=4096)
    df = pl.read_parquet(path)
    compare(df, t, t.column_names)
    us = RNG.integers(-10**15, 2 * 10**15, n)
    t96 = pa.table({"ts": pa.array(us, pa.timestamp("us"), mask=RNG.random(n) < 0.1)})
    path96 = str(tmp_path / "int96.parquet")
    pq.write_table(t96, path96, use_deprecated_int96_timestamps=True, compression="snappy")
    s = pl.read_parquet(path96)["ts"]
    values, valid = s._download()
    want_valid = np.array([x is not None for x in t96.column("ts").to_pylist()])
    assert s.dtype == pl.Datetime and s.dtype.time_unit == "ns"                # INT96 -> Datetime[ns], the reference's default
    assert np.array_equal(valid, want_valid) and np.array_equal(values[want_valid], us[want_valid] * 1000)


def test_scan_over_several_files_unifies_dictionaries(pl, tmp_path):
    """A directory of files with one schema is one scan: frames are read file by file and concatenated on the device
    (plx_frame_concat); string columns, whose dictionaries differ from file to file, are first brought onto one dictionary."""
    rng = np.random.default_rng(5)
    parts, paths = [], []
    for f, (n, words) in enumerate([(3001, ["a", "b", "c"]), (1999, ["c", "zz", "a", "only here"]), (2500, ["b"])]):
        t = pa.table({"i": pa.array(rng.integers(0, 10**9, n), mask=rng.random(n) < 0.1), "f": rng.normal(size=n), "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.3),
                      "s": pa.array(np.array(words)[rng.integers(0, len(words), n)], mask=rng.random(n) < 0.2),
                      "d": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32())})
        paths.append(str(tmp_path / f"part-{f}.parquet"))
        pq.write_table(t, paths[-1], row_group_size=777, compression=["none", "snappy", "zstd"][f], use_dictionary=f != 1)
        parts.append(t)
    want = pa.concat_tables(parts)
    df = pl.read_parquet(str(tmp_path))
    assert df.height == want.num_rows
    compare(df, want, want.column_names)
    cats = list(df["s"].dtype.categories)
    assert sorted(cats) == ["a", "b", "c", "only here", "zz"] and len(set(cats)) == 5
    # pruning across files, then a group-by on the unified string column
    c = pl.col
    out = pl.scan_parquet(paths).filter(c("d") >= 0).group_by("s").agg(pl.len().alias("n"), c("f").sum().alias("sf")).collect().sort_host("s")
    s = np.array([x if x is not None else "\0" for x in want.column("s").to_pylist()]); fv = want.column("f").to_numpy()
    for i, key in enumerate(out["s"]):
        mk = s == (key if key is not None else "\0")
        assert out["n"][i] == int(mk.sum()) and abs(out["sf"][i] - fv[mk].sum()) < 1e-9 * max(1.0, np.abs(fv[mk]).sum())
    assert len(out["s"]) == 6


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_compressed_bodies(pl, tmp_path, codec):
    """LZ4-frame (pyarrow's feather default) and Zstandard bodies: buffers are inflated by the library's own host decoders
    (host_codecs.hpp), then uploaded like any other."""
    n = 3001
    t = I.table(n)
    path = str(tmp_path / "t.arrow")
    I.write(path, t, chunk=1000, compression=codec)
    df = pl.read_ipc(path)
    I.compare(df, t, t.column_names)


def test_tpch_queries_over_the_reference_sample_files(pl):
    """Q1 and the two-table Q3 straight from the files the reference ships (written by its IPC writer: int64 / double / timestamp[us] /
    large_string columns), through scan_ipc; expected values computed with numpy from pyarrow's read of the same files."""
    from polars_amd import queries
    li = ipc.open_file(os.path.join(PDS_HEADS, "lineitem.feather")).read_all()
    od = ipc.open_file(os.path.join(PDS_HEADS, "orders.feather")).read_all()
    col = lambda t, n: np.array(t.column(n).to_pylist()) if pa.types.is_large_string(t.column(n).type) else t.column(n).to_numpy()
    cutoff = dt.datetime(1996, 4, 1)                                # inside the 10 rows' ship dates: the filter removes some of them
    out = queries.q1(pl.scan_ipc(os.path.join(PDS_HEADS, "lineitem.feather")), cutoff).collect().sort_host(["l_returnflag", "l_linestatus"])
    keep = col(li, "l_shipdate") <= np.datetime64(cutoff, "us")
    fl, st, q, p, d, x = (col(li, n)[keep] for n in ("l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_discount", "l_tax"))
    groups = sorted(set(zip(fl.tolist(), st.tolist())))
    assert list(zip(out["l_returnflag"], out["l_linestatus"])) == groups and len(groups) >= 2
    for i, (a, b) in enumerate(groups):
        m = (fl == a) & (st == b)
        assert out["count_order"][i] == int(m.sum()) and out["sum_qty"][i] == int(q[m].sum())
        for name, want in (("sum_base_price", p[m].sum()), ("sum_disc_price", (p[m] * (1 - d[m])).sum()), ("sum_charge", (p[m] * (1 - d[m]) * (1 + x[m])).sum()),
                           ("avg_qty", q[m].mean()), ("avg_price", p[m].mean()), ("avg_disc", d[m].mean())):
            assert abs(out[name][i] - want) <= 1e-6 * abs(want), (name, a, b)           # float sums: rel-tol 1e-6 (SURVEY.md 8(c) semantics 5)
    # Q3 on (lineitem, orders): every customer passes the segment stand-in (seg_mod = 1); the date splits the ten orders
    date = dt.datetime(1996, 2, 1)
    lf = queries.q3(pl.scan_ipc(os.path.join(PDS_HEADS, "lineitem.feather")), pl.scan_ipc(os.path.join(PDS_HEADS, "orders.feather")), date, seg_mod=1)
    got = lf.collect().sort_host("l_orderkey")
    d64 = np.datetime64(date, "us")
    odate, oprio = col(od, "o_orderdate"), col(od, "o_shippriority")
    ok = {int(k): (int(odate[i].astype("datetime64[us]").astype(np.int64)), int(oprio[i])) for i, k in enumerate(col(od, "o_orderkey")) if odate[i] < d64}      # Datetime columns download as microseconds
    want = {}
    for k, sd, p, d in zip(col(li, "l_orderkey"), col(li, "l_shipdate"), col(li, "l_extendedprice"), col(li, "l_discount")):
        if sd > d64 and int(k) in ok:
            want[int(k)] = want.get(int(k), 0.0) + p * (1 - d)
    assert got["l_orderkey"] == sorted(want) and len(want) >= 1
    for i, k in enumerate(got["l_orderkey"]):
        assert abs(got["revenue"][i] - want[k]) <= 1e-9 * want[k] and got["o_shippriority"][i] == ok[k][1] and got["o_orderdate"][i] == ok[k][0]
    # the three-table form on these heads has no customer for any of the ten orders: an empty result with the right columns
    empty = queries.q3_full(pl.scan_ipc(os.path.join(PDS_HEADS, "customer.feather")), pl.scan_ipc(os.path.join(PDS_HEADS, "orders.feather")),
                            pl.scan_ipc(os.path.join(PDS_HEADS, "lineitem.feather")), date).collect()
    assert empty.height == 0 and empty.columns == ["o_orderkey", "o_orderdate", "o_shippriority", "revenue"]


def test_row_group_shards_of_one_scan_add_up(pl, tmp_path):
    """scan_parquet(..., shard=(rank, world)): what each of `world` processes would read, here one after the other in one process --
    the shards' frames concatenated (dictionaries unified on the way) are the whole file, and per-shard partial aggregates add up."""
    rng = np.random.default_rng(6)
    n = 50_000
    t = pa.table({"k": np.arange(n), "v": pa.array(rng.integers(0, 1000, n), mask=rng.random(n) < 0.1), "s": pa.array(np.array(["x", "y", "z", "w"])[(np.arange(n) // 9000) % 4])})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=3000, compression="snappy")
    from polars_amd import io
    world = 3
    shards = []
    for rank in range(world):
        src = io.ParquetFrame(path, shard=(rank, world))
        src.request(None, [])
        shards.append(src.materialise())
    assert sum(d.height for d in shards) == n and min(d.height for d in shards) >= n // world - 3000
    compare(io.concat_frames(shards), t, t.column_names)
    c = pl.col
    total = sum(pl.scan_parquet(path, shard=(rank, world)).filter(c("k") >= 10_000).select(c("v").sum().alias("sv")).collect()["sv"].to_list()[0] for rank in range(world))
    v = t.column("v").to_numpy(zero_copy_only=False)
    assert total == int(np.nansum(v[10_000:]))


IO_FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_files")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(IO_FILES) if f.endswith(".parquet") and not f.startswith("nested")))
def test_the_reference_s_own_parquet_fixtures_on_the_device(pl, name):
    """The files the reference's I/O tests read (tests/golden/io_files, see tests/test_parquet_emu_cpu.py for what each one is about),
    decoded on the device."""
    path = os.path.join(IO_FILES, name)
    want = pq.read_table(path)
    df = pl.read_parquet(path)
    assert df.columns == want.column_names and df.height == want.num_rows
    for n in want.column_names:
        w = want.column(n).combine_chunks()
        s = df[n]
        assert s.null_count() == w.null_count, n
        if pa.types.is_binary(w.type) or pa.types.is_string(w.type) or pa.types.is_large_string(w.type):
            assert s.to_list() == w.to_pylist(), n
            continue
        if pa.types.is_timestamp(w.type):
            assert s.dtype.time_unit == w.type.unit, n                     # ns for INT96 and for tz_aware.parquet
            w = w.cast(pa.int64())
        values, valid = s._download()
        wl = w.to_pylist()
        ok = np.array([x is not None for x in wl], bool)
        assert (valid is None and ok.all()) or np.array_equal(valid, ok), n
        got = values[ok].tolist()
        exp = [x for x in wl if x is not None]
        assert got == exp or np.allclose(got, exp, rtol=0, atol=0, equal_nan=True), n


def test_logical_types_survive_the_arrow_export(pl, tmp_path):
    """Series.to_arrow(): Date / Datetime[unit] / string columns of a scan come back with their Arrow types (what the Polars attachment
    hands to polars.from_arrow), dictionaries stay dictionaries."""
    t = pa.table({"d": pa.array([1, None, 3], pa.date32()), "ns": pa.array([1, 2, None], pa.timestamp("ns")), "ms": pa.array([5, 6, 7], pa.timestamp("ms")),
                  "s": pa.array(["x", None, "y"]), "i": pa.array([1, 2, 3])})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path)
    back = pl.read_parquet(path).to_arrow()
    assert back.schema.types == [pa.date32(), pa.timestamp("ns"), pa.timestamp("ms"), pa.large_string(), pa.int64()]
    for n in t.column_names:
        assert back.column(n).to_pylist() == t.column(n).to_pylist(), n
    cat = pl.Series.from_arrow("c", pa.array(["a", "b", "a"]).dictionary_encode()).to_arrow()
    assert pa.types.is_dictionary(cat.type) and cat.to_pylist() == ["a", "b", "a"]


def test_slice_pushed_into_the_scan(pl, tmp_path):
    """head / slice directly above a scan read only the overlapping row groups (tests/test_io_cpu.py pins the planning); the rows that
    come back are the same as slicing the whole table."""
    n = 20_000
    t = pa.table({"k": np.arange(n), "s": pa.array(np.array(["a", "b", "c"])[np.arange(n) % 3]), "v": pa.array(np.arange(n) * 0.5, mask=np.arange(n) % 7 == 0)})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=1500, compression="snappy")
    for off, ln in [(0, 5), (1499, 3), (4000, 6000), (19_990, 100), (-10, 4), (25_000, 3)]:
        lf = pl.scan_parquet(path).slice(off, ln)
        df = lf.collect()
        want = t.slice(off, ln) if off >= 0 else t.slice(max(0, n + off), ln)
        if off >= n:
            want = t.slice(0, 0)
        assert df.height == want.num_rows, (off, ln)
        compare(df, want, t.column_names)
        read = lf._node.input.frame.last_read
        assert read["row_groups"] <= (ln + 1499) // 1500 + 1 or off < 0, (off, ln, read)
    out = pl.scan_parquet(path).select(pl.col("k"), (pl.col("v") * 2).alias("w")).head(7).collect()
    assert out["k"].to_list() == list(range(7)) and out["w"].to_list() == [None, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]


def test_results_written_and_scanned_again(pl, tmp_path):
    """write_parquet / write_ipc of a result frame (host encoder) and a scan of what was written: logical types make the round trip."""
    t = pa.table({"d": pa.array([1, None, 3, 4], pa.date32()), "ts": pa.array([10, 20, None, 40], pa.timestamp("us")), "s": pa.array(["x", None, "y", "x"]),
                  "f": pa.array([0.5, 1.5, None, 2.5]), "b": pa.array([True, None, False, True])})
    src = str(tmp_path / "in.parquet")
    pq.write_table(t, src)
    df = pl.read_parquet(src)
    p2, p3 = str(tmp_path / "out.parquet"), str(tmp_path / "out.arrow")
    df.write_parquet(p2)
    df.write_ipc(p3, compression="zstd")
    assert pq.read_table(p2).to_pydict() == t.to_pydict()
    for again in (pl.read_parquet(p2), pl.read_ipc(p3)):
        compare(again, t, t.column_names)


def test_concat_of_frames_and_plans(pl):
    """polars.concat(how="vertical"): eager frames and lazy plans; string dictionaries of the parts are unified."""
    a = pl.DataFrame({"k": np.arange(5), "s": pl.Series.from_arrow("s", pa.array(["x", "y", None, "x", "y"])), "f": np.arange(5) * 0.5})
    b = pl.DataFrame({"k": np.arange(5, 9), "s": pl.Series.from_arrow("s", pa.array(["z", "y", "z", None])), "f": np.arange(4) * 2.0})
    both = pl.concat([a, b])
    assert both.height == 9 and both["k"].to_list() == list(range(9)) and both["s"].to_list() == ["x", "y", None, "x", "y", "z", "y", "z", None]
    assert both["f"].to_list() == [0.0, 0.5, 1.0, 1.5, 2.0, 0.0, 2.0, 4.0, 6.0]
    c = pl.col
    out = pl.concat([a.lazy().filter(c("k") >= 3), b.lazy()]).group_by("s").agg(c("f").sum().alias("sf"), pl.len().alias("n")).collect().sort_host("s")
    assert out["s"] == ["x", "y", "z", None] and out["n"] == [1, 2, 2, 1] and out["sf"] == [1.5, 4.0, 4.0, 6.0]


def test_hive_partition_columns(pl, tmp_path):
    """Partition columns of a hive-style directory arrive as constant columns per file; a predicate on one skips files."""
    want_rows = []
    for y in (1994, 1995):
        for seg in ("A", "B"):
            os.makedirs(tmp_path / f"year={y}" / f"seg={seg}")
            k = np.arange(2500) + (y - 1994) * 10_000 + (seg == "B") * 5000
            pq.write_table(pa.table({"k": k, "v": k * 0.5}), str(tmp_path / f"year={y}" / f"seg={seg}" / "part-0.parquet"), row_group_size=1000)
            want_rows += [(int(x), y, seg) for x in k]
    df = pl.read_parquet(str(tmp_path))
    assert df.columns == ["k", "v", "year", "seg"] and df.height == 10_000
    assert list(zip(df["k"].to_list(), df["year"].to_list(), df["seg"].to_list())) == want_rows
    c = pl.col
    lf = pl.scan_parquet(str(tmp_path)).filter(c("year") == 1995).group_by("seg").agg(c("v").sum().alias("sv"), pl.len().alias("n"))
    out = lf.collect().sort_host("seg")
    assert out["seg"] == ["A", "B"] and out["n"] == [2500, 2500]
    assert out["sv"] == [sum(x * 0.5 for x, y, s in want_rows if y == 1995 and s == seg) for seg in ("A", "B")]
    only = pl.scan_parquet(str(tmp_path)).select(c("year").sum().alias("sy")).collect()          # partition columns only: no file column is read
    assert only["sy"].to_list() == [(1994 + 1995) * 5000]


def test_scan_keywords_n_rows_and_file_paths(pl, tmp_path):
    paths = []
    for i in range(3):
        paths.append(str(tmp_path / f"f{i}.parquet"))
        pq.write_table(pa.table({"k": np.arange(2000) + 2000 * i}), paths[-1], row_group_size=500)
    df = pl.scan_parquet(paths, include_file_paths="path", n_rows=2500).collect()
    assert df.height == 2500 and df["k"].to_list() == list(range(2500))
    assert df["path"].to_list() == [paths[0]] * 2000 + [paths[1]] * 500
    out = pl.scan_parquet(paths, include_file_paths="path").group_by("path").agg(pl.len().alias("n")).collect().sort_host("path")
    assert out["path"] == sorted(paths) and out["n"] == [2000, 2000, 2000]


def test_tpch_q6_on_the_device(pl):
    """TPC-H Q6 (queries.q6: is_between on dates and discounts, a quantity bound, one product sum) against numpy on the generator's host twin."""
    from polars_amd import datagen, queries
    li = datagen.lineitem_host(500_000, seed=6)
    names = ["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"]
    out = queries.q6(datagen.to_frame(pl, li, names).lazy()).collect()
    m = (li["l_shipdate"] >= datagen.us(1994, 1, 1)) & (li["l_shipdate"] < datagen.us(1995, 1, 1)) & (li["l_discount"] >= 0.05) & (li["l_discount"] <= 0.07) & (li["l_quantity"] < 24)
    want = float((li["l_extendedprice"][m] * li["l_discount"][m]).sum())
    assert abs(out["revenue"].to_list()[0] - want) <= 1e-6 * want          # float sum: 1e-6 relative (BASELINE.json north_star)
