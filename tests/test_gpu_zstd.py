"""Zstandard pages on the device (polars_amd/csrc/parquet_zstd.hpp: pq_zstd_entropy + pq_zstd_execute behind the host index pass).  The reference writes zstd by
default (crates/polars-parquet/src/parquet/compression.rs:103-120) and inflates pages with the zstd crate (compression.rs:137-138, 221-236); ground truth here is
pyarrow's decode of the same file.  The same bodies run on the CPU harness against the real codec and against the library's host decoder on corrupt streams
(tests/test_parquet_emu_cpu.py); what only the hardware can show -- the wavefront's LDS ordering, the ring flushes, four blocks sharing a wavefront -- is below."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from test_gpu_parquet import compare, table

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(606)


def kernels_of(pl, fn):
    """names of the kernels the library's HIP-event tracer saw while fn() ran -> {name: launches}"""
    F = pl._ffi
    F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
    try:
        out = fn()
        F.check(F.lib().plx_synchronize())
        cap = 65536
        recs = (F.ProfileRecord * cap)()
        n = C.c_int32()
        F.check(F.lib().plx_profile_fetch(recs, cap, C.byref(n)))
        seen = {}
        for i in range(n.value):
            nm = recs[i].name.decode()
            seen[nm] = seen.get(nm, 0) + 1
    finally:
        F.check(F.lib().plx_profile_enable(0))
    return out, seen


@pytest.mark.parametrize("version,dictionary", [("1.0", True), ("2.0", True), ("2.0", False)])
@pytest.mark.parametrize("level", [1, 9, 19])
def test_every_dtype_through_the_zstd_passes(pl, tmp_path, version, dictionary, level):
    n = 20_000
    t = table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", compression_level=level, data_page_version=version, use_dictionary=dictionary, row_group_size=6500, data_page_size=4096)
    df, seen = kernels_of(pl, lambda: pl.read_parquet(path, columns=t.column_names))
    assert seen.get("pq_zstd_entropy", 0) > 0 and seen.get("pq_zstd_execute", 0) > 0, seen
    compare(df, pq.read_table(path), t.column_names)


def test_host_switch_keeps_the_kernels_out(pl, tmp_path, monkeypatch):
    t = table(5000)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd")
    monkeypatch.setenv("PLX_PARQUET_ZSTD", "host")
    df, seen = kernels_of(pl, lambda: pl.read_parquet(path))
    assert "pq_zstd_entropy" not in seen and "pq_zstd_execute" not in seen, seen
    compare(df, t, t.column_names)


def payload_table(n):
    """columns whose pages stress one part of the decoder each"""
    key = np.sort(RNG.integers(0, 1 << 40, n))                                      # a match per value that reads what the previous match wrote (LDS ring, sequence order)
    far = np.tile(RNG.integers(0, 1 << 62, 12_000), n // 12_000 + 1)[:n]             # matches ~96 KB back: beyond the 32 KB ring, read from HBM behind a flush; across blocks
    price = np.round(RNG.uniform(900, 105_000, n), 2)                               # Huffman literals in four streams, few sequences
    rnd = RNG.integers(-2**63, 2**63 - 1, n)                                        # incompressible: raw blocks
    zeros = np.zeros(n, np.int64)                                                   # RLE blocks / one long match at offset 1 or 8
    codes = RNG.integers(0, 7, n).astype(np.int8)                                   # short Huffman codes
    runs = np.repeat(RNG.integers(0, 1 << 30, (n + 96) // 97), 97)[:n]              # long matches (cooperative copies), repeat offsets
    dense = np.sort(RNG.integers(1, 4 * n, n))                                      # one sequence a value: a literal byte or two, the rest from the value above (the row path)
    ticks = np.cumsum(RNG.integers(0, 3, n)).astype(np.int32)                       # ... four bytes wide, many equal neighbours (matches of several rows)
    disc = RNG.integers(0, 11, n) / 100.0                                           # eleven distinct doubles: every value a match a few rows (near) or a few thousand rows (beyond the ring) up
    return pa.table({"key": key, "dense": dense, "ticks": ticks, "disc": disc, "far": far, "price": price, "rnd": rnd, "zeros": zeros, "codes": codes, "runs": runs,
                     "nullable": pa.array(RNG.integers(0, 1 << 30, n), mask=RNG.random(n) < 0.1)})


@pytest.mark.parametrize("level", [1, 3, 12])
@pytest.mark.parametrize("page_size", [1 << 20, 64 << 10])
def test_pages_of_many_blocks(pl, tmp_path, level, page_size):
    """1 MB pages = eight 128 KB blocks each: four blocks share a wavefront in the entropy pass, a page's wavefront walks its blocks in order in the execute pass
    (repeat offsets and matches cross block borders).  No dictionaries: the values themselves are what zstd sees."""
    n = 1_500_000
    t = payload_table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", compression_level=level, use_dictionary=False, data_page_size=page_size, row_group_size=600_000)
    df, seen = kernels_of(pl, lambda: pl.read_parquet(path))
    assert seen.get("pq_zstd_entropy", 0) > 0 and seen.get("pq_zstd_execute", 0) > 0, seen
    compare(df, t, t.column_names)
    # the same pages through the host threads: bit-identical columns
    import os
    os.environ["PLX_PARQUET_ZSTD"] = "host"
    try:
        dh = pl.read_parquet(path)
    finally:
        del os.environ["PLX_PARQUET_ZSTD"]
    for name in t.column_names:
        a, va = df[name]._download()
        b, vb = dh[name]._download()
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)) and ((va is None and vb is None) or np.array_equal(va, vb)), name


def test_long_dictionary_pages_and_dictionary_indices(pl, tmp_path):
    """pyarrow's default layout: a dictionary page of up to 1 MB per chunk (inflated by a host thread while the walk goes on, like a long Snappy dictionary page), data pages
    of bit-packed indices, a fall-back to PLAIN pages when the dictionary is full -- all three in one column chunk."""
    n = 1_200_000
    t = pa.table({"k": pa.array(np.sort(RNG.integers(0, 1 << 40, n))), "few": pa.array(RNG.integers(0, 50, n)), "s": pa.array(np.array(["R", "A", "N"])[RNG.integers(0, 3, n)]),
                  "ship": pa.array(RNG.integers(694224000, 912470400, n) * 1_000_000, pa.timestamp("us"))})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", row_group_size=1 << 20)
    df, seen = kernels_of(pl, lambda: pl.read_parquet(path))
    assert seen.get("pq_zstd_entropy", 0) > 0, seen
    compare(df, t, t.column_names)


def test_corrupt_zstd_pages_are_status_codes(pl, tmp_path):
    """Bit flips inside zstd page bytes: the index pass rejects a header, the kernels flag the stream (PE_ZSTD), or a well-formed different value comes out -- never a hang
    or a crash, and the library stays usable."""
    n = 60_000
    t = pa.table({"k": pa.array(np.sort(RNG.integers(0, 1 << 40, n))), "v": pa.array(np.round(RNG.uniform(0, 1e5, n), 2))})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", use_dictionary=False, data_page_size=1 << 20)
    raw = bytearray(open(path, "rb").read())
    md_len = int.from_bytes(raw[-8:-4], "little")
    bad = str(tmp_path / "bad.parquet")
    seen = set()
    for trial in range(40):
        b = bytearray(raw)
        for _ in range(1 + trial % 3):
            b[int(RNG.integers(64, len(b) - 8 - md_len))] ^= 1 << int(RNG.integers(0, 8))
        open(bad, "wb").write(b)
        try:
            pl.read_parquet(bad)
            seen.add("ok")
        except pl.PlxError as e:
            seen.add("error")
            assert e.code in (1, 3), e
    assert "error" in seen
    compare(pl.read_parquet(path), t, t.column_names)


def test_q1_from_a_zstd_file(pl, orc, tmp_path):
    """the file the reference would write (zstd is its default) -> scan -> TPC-H Q1, against the oracle on the same rows"""
    from polars_amd import datagen, queries
    n = 300_000
    li = datagen.lineitem_host(n, seed=8)
    t = pa.table({"l_quantity": pa.array(li["l_quantity"]), "l_extendedprice": pa.array(li["l_extendedprice"]), "l_discount": pa.array(li["l_discount"]),
                  "l_tax": pa.array(li["l_tax"]), "l_returnflag": pa.array([datagen.FLAGS[c] for c in li["l_returnflag"]]),
                  "l_linestatus": pa.array([datagen.STATUS[c] for c in li["l_linestatus"]]), "l_shipdate": pa.array(li["l_shipdate"], pa.timestamp("us"))})
    path = str(tmp_path / "lineitem.parquet")
    pq.write_table(t, path, row_group_size=50_000, compression="zstd")
    want = orc.q1(li, datagen.us(1998, 9, 2))
    out, seen = kernels_of(pl, lambda: queries.q1(pl.scan_parquet(path)).collect().sort_host(["l_returnflag", "l_linestatus"]))
    assert seen.get("pq_zstd_execute", 0) > 0, seen
    assert [datagen.FLAGS.index(x) for x in out["l_returnflag"]] == want["l_returnflag"].tolist()
    assert out["count_order"] == want["count_order"].tolist() and out["sum_qty"] == want["sum_qty"].tolist()
    for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
        assert np.allclose(np.array(out[c]), want[c], rtol=1e-6, atol=0), c


def test_pages_of_many_sequences_take_the_host_threads(pl, tmp_path, monkeypatch):
    """1 MB pages of sorted keys are 1.3e5 sequences each: beyond PLX_PARQUET_ZSTD_HOST_SEQS (50 000) a page is inflated by the host pool while the column's other pages stay
    device streams; the price column of the same file (Huffman literals, no sequences) stays on the device whatever its pages' size.  Same columns with the limit off."""
    n = 1_200_000
    t = pa.table({"k": pa.array(np.sort(RNG.integers(1, 4 * n, n))), "price": pa.array(np.round(RNG.uniform(900, 105_000, n), 2)),
                  "mixed": pa.array(np.where(np.arange(n) < n // 2, np.sort(RNG.integers(1, 4 * n, n)), RNG.integers(-2**62, 2**62, n)))})        # chunks of both kinds of page
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", compression_level=3, use_dictionary=False, data_page_size=1 << 20, max_rows_per_page=1 << 20, row_group_size=400_000)
    df, seen = kernels_of(pl, lambda: pl.read_parquet(path))
    assert seen.get("pq_zstd_entropy", 0) > 0, seen
    compare(df, t, t.column_names)
    monkeypatch.setenv("PLX_PARQUET_ZSTD_HOST_SEQS", "0")
    d0 = pl.read_parquet(path)
    for name in t.column_names:
        a, _ = df[name]._download()
        b, _ = d0[name]._download()
        assert np.array_equal(a, b), name
