"""The device Parquet decoder, checked on the CPU.

tests/emu/parquet_emu.cpp instantiates the PRODUCT's reader (polars_amd/csrc/parquet_reader.hpp) with a backend that runs the
product's per-thread / per-wavefront kernel bodies (parquet_device.hpp) one thread after another on host memory.  Ground truth:
pyarrow's own decode of the same file (files are written here with pyarrow: codecs, page versions, dictionary on / off, page and row
group sizes, nulls).  Snappy is additionally fuzzed on hand-built streams (every element kind, overlapping copies, long literals)
against a 20-line Python decoder, and on corrupt streams, which must yield an error code and never touch bytes past the output.
"""
import datetime as dt
import os
import struct

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import parquet_emu as E

RNG = np.random.default_rng(20260923)


def check_column(path, table, name, row_groups=None, order=0, ci=None):
    ci = table.column_names.index(name) if ci is None else ci           # ci: the LEAF index when the file also has nested columns
    nrg = pq.ParquetFile(path).metadata.num_row_groups
    rgs = list(range(nrg)) if row_groups is None else row_groups
    r = E.read_column(path, rgs, ci, order)
    want = pq.ParquetFile(path).read_row_groups(rgs, columns=[name]).column(0).combine_chunks() if rgs else table.column(name).combine_chunks().slice(0, 0)
    n = len(want)
    assert len(r["values"]) == n
    wvalid = np.array([v is not None for v in want.to_pylist()], bool) if want.null_count else np.ones(n, bool)
    if r["valid"] is None:
        assert want.null_count == 0 and r["null_count"] == 0
    else:
        assert np.array_equal(r["valid"], wvalid) and r["null_count"] == want.null_count
        # pad bits of the last validity word are zero
        if n % 64:
            assert int(r["raw_validity_words"][-1]) >> (n % 64) == 0
    t = want.type
    if pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_binary(t) or pa.types.is_large_binary(t):
        wl = want.to_pylist()
        cats = r["categories"]
        assert len(set(cats)) == len(cats)
        for i in np.nonzero(wvalid)[0]:
            w = wl[i] if isinstance(wl[i], bytes) else wl[i].encode()
            assert cats[r["values"][i]] == w
        assert set(cats) >= {(x if isinstance(x, bytes) else x.encode()) for x in wl if x is not None}
        return r
    if pa.types.is_timestamp(t):
        exp = want.cast(pa.int64()).to_numpy(zero_copy_only=False)
    elif pa.types.is_date32(t):
        exp = want.cast(pa.int32()).to_numpy(zero_copy_only=False)
    else:
        exp = want.to_numpy(zero_copy_only=False)
    got = r["values"]
    if pa.types.is_boolean(t):
        exp = np.array([bool(x) if x is not None else False for x in want.to_pylist()])
        assert np.array_equal(got[wvalid], exp[wvalid])
        if n % 64:
            assert int(r["raw_value_words"][-1]) >> (n % 64) == 0
    elif pa.types.is_floating(t):
        e = np.asarray(exp, dtype=got.dtype)
        assert np.array_equal(got[wvalid].view(np.uint32 if got.dtype == np.float32 else np.uint64), e[wvalid].view(np.uint32 if got.dtype == np.float32 else np.uint64))   # bit-exact, NaN payloads included
    else:
        if pa.types.is_timestamp(t) or pa.types.is_date32(t):
            want = want.cast(pa.int64() if pa.types.is_timestamp(t) else pa.int32())
        e = np.fromiter((x if x is not None else 0 for x in want.to_pylist()), dtype=got.dtype, count=n)
        assert np.array_equal(got[wvalid], e[wvalid])
    assert np.all(got[~wvalid] == 0)       # null slots are written (as zero), never left uninitialised
    return r


def mixed_table(n, null_frac=0.2):
    m = lambda: RNG.random(n) < null_frac
    words = np.array(["", "a", "bb", "BUILDING", "AUTOMOBILE", "a much longer string that does not fit in twelve bytes", "ünïcödé"])
    return pa.table({
        "i8": pa.array(RNG.integers(-128, 128, n).astype(np.int8)), "i16": pa.array(RNG.integers(-30000, 30000, n).astype(np.int16), mask=m()),
        "i32": pa.array(RNG.integers(-2**31, 2**31, n).astype(np.int32), mask=m()), "i64": pa.array(RNG.integers(-2**62, 2**62, n), mask=m()),
        "u8": pa.array(RNG.integers(0, 256, n).astype(np.uint8), mask=m()), "u16": pa.array(RNG.integers(0, 65536, n).astype(np.uint16)),
        "u32": pa.array(RNG.integers(0, 2**32, n).astype(np.uint32), mask=m()), "u64": pa.array(RNG.integers(0, 2**63, n).astype(np.uint64) * 2 + 1, mask=m()),
        "f32": pa.array(RNG.normal(size=n).astype(np.float32), mask=m()), "f64": pa.array(np.where(RNG.random(n) < 0.05, np.nan, RNG.normal(size=n)), mask=m()),
        "b": pa.array(RNG.random(n) < 0.5, mask=m()), "b_req": pa.array(RNG.random(n) < 0.1),
        "date": pa.array(RNG.integers(0, 20000, n).astype(np.int32), pa.date32(), mask=m()),
        "ts": pa.array(RNG.integers(0, 2**50, n), pa.timestamp("us"), mask=m()),
        "s": pa.array(words[RNG.integers(0, len(words), n)], mask=m()), "ls": pa.array(words[RNG.integers(0, 3, n)], pa.large_string()),
        "bin": pa.array([bytes([i % 7, 0, 255]) for i in range(n)], pa.binary()),
        "low_card": pa.array(RNG.integers(0, 3, n)), "const": pa.array(np.full(n, 42, np.int64)), "all_null": pa.array(np.zeros(n, np.int64), mask=np.ones(n, bool)),
        "runs": pa.array(np.repeat(RNG.integers(0, 5, (n + 99) // 100), 100)[:n], mask=np.repeat(RNG.random((n + 49) // 50) < 0.3, 50)[:n]),
    })


@pytest.mark.parametrize("compression", ["none", "snappy", "zstd", "lz4", "gzip"])       # snappy: device kernel; zstd / lz4 (raw) / gzip: host threads, then the uncompressed device path
@pytest.mark.parametrize("version,dictionary", [("1.0", True), ("2.0", True), ("1.0", False), ("2.0", False)])
def test_every_dtype_against_pyarrow(tmp_path, compression, version, dictionary):
    n = 5000
    t = mixed_table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression=compression, data_page_version=version, use_dictionary=dictionary, row_group_size=1700, data_page_size=2048)
    for name in t.column_names:
        check_column(path, t, name)         # strings without a dictionary (PLAIN pages): views assembled by host threads, encoded by the backend


def test_thread_order_does_not_matter(tmp_path):
    t = mixed_table(3000)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="snappy", row_group_size=1000, data_page_size=1024)
    for name in ("i64", "f64", "s", "b", "runs", "u8"):
        check_column(path, t, name, order=1)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 127, 128, 129, 4096])
def test_sizes_around_word_boundaries(tmp_path, n):
    t = pa.table({"a": pa.array(RNG.integers(0, 100, n), mask=RNG.random(n) < 0.5), "b": pa.array(RNG.random(n) < 0.5, mask=RNG.random(n) < 0.5),
                  "s": pa.array(np.array(["x", "y"])[RNG.integers(0, 2, n)] if n else [], pa.string()), "f": pa.array(RNG.normal(size=n))})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="snappy", data_page_size=64, row_group_size=50)
    for name in t.column_names:
        check_column(path, t, name)


def test_row_group_subsets_in_any_order(tmp_path):
    n = 10_000
    t = pa.table({"k": pa.array(np.arange(n)), "v": pa.array(RNG.normal(size=n), mask=RNG.random(n) < 0.1), "s": pa.array(np.array(["a", "b", "c", "d"])[np.arange(n) // 2600])})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, row_group_size=1000, compression="snappy")
    for rgs in ([3], [9, 0], [2, 3, 4, 7], [], [5, 5]):
        for name in t.column_names:
            check_column(path, t, name, row_groups=rgs)
    # the dictionary of a subset only holds what those chunks' dictionaries hold, in first-appearance order
    r = E.read_column(path, [9, 0], 2)
    assert r["categories"] == [b"d", b"a"]


def test_nulls_at_page_boundaries_and_long_runs(tmp_path):
    n = 20_000
    valid = np.ones(n, bool)
    valid[1000:3000] = False; valid[4095:4097] = False; valid[-1] = False; valid[0] = False
    vals = RNG.integers(0, 1 << 40, n)
    t = pa.table({"a": pa.array(vals, mask=~valid), "d": pa.array(vals % 7, mask=~valid)})
    path = str(tmp_path / "t.parquet")
    for ver in ("1.0", "2.0"):
        pq.write_table(t, path, data_page_size=512, row_group_size=7000, data_page_version=ver, compression="snappy")
        ra = check_column(path, t, "a"); check_column(path, t, "d")
        assert ra["stats"]["data_pages"] >= 15 and ra["null_count"] == int((~valid).sum())


def test_optional_column_without_nulls_skips_the_level_tables(tmp_path):
    n = 8000
    t = pa.table({"a": pa.array(RNG.integers(0, 50, n))})      # pyarrow writes OPTIONAL + null_count = 0 statistics
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, data_page_size=1024)
    r = check_column(path, t, "a")
    pages = r["stats"]["data_pages"]
    assert r["valid"] is None and r["stats"]["run_entries"] < 40 * pages      # only index-stream tables, no level tables
    # without statistics the levels are decoded, and the validity is dropped again when it turns out to be all ones
    pq.write_table(t, path, data_page_size=1024, write_statistics=False)
    r2 = check_column(path, t, "a")
    assert r2["valid"] is None and r2["null_count"] == 0


def test_snappy_sees_real_back_references(tmp_path):
    """Columns whose pages compress well (periodic PLAIN values): copies, overlapping copies and multi-round streams."""
    n = 60_000
    t = pa.table({"period7": pa.array(np.tile(np.arange(7, dtype=np.int64) * 1_000_003, n // 7 + 1)[:n]), "zeros": pa.array(np.zeros(n)),
                  "ramp": pa.array(np.arange(n, dtype=np.int32) // 16), "text": pa.array(np.tile(np.array(["lorem ipsum dolor", "sit amet", "consectetur"]), n // 3 + 1)[:n])})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="snappy", use_dictionary=["text"], data_page_size=1 << 16)
    for name in t.column_names:
        r = check_column(path, t, name)
        if name != "text":
            assert r["stats"]["snappy_bytes_out"] > 4 * r["stats"]["file_bytes"]


def test_unsupported_files_say_what_they_are(tmp_path):
    n = 100
    t = pa.table({"a": pa.array(np.arange(n)), "dec": pa.array([None] * n, pa.decimal128(10, 2)), "lst": pa.array([[1, 2]] * n), "ms": pa.array(np.arange(n), pa.timestamp("ms")),
                  "z": pa.array(np.arange(n))})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression={"a": "brotli", "dec": "none", "lst.list.element": "none", "ms": "none", "z": "brotli"}, use_dictionary=False,
                   column_encoding={"z": "DELTA_BINARY_PACKED"})
    for col, word in ((0, "BROTLI"), (1, "decimal"), (2, "nested"), (4, "BROTLI")):
        with pytest.raises(E.EmuError) as ei:
            E.read_column(path, [0], col)
        assert ei.value.code == 3 and word in str(ei.value), str(ei.value)
    r = E.read_column(path, [0], 3)                     # millisecond timestamps keep their unit (logical kind 5), as the reference keeps it
    assert r["logical"] == 5 and r["dtype"] == 4 and np.array_equal(r["values"], np.arange(n))
    # booleans in an encoding nobody decodes here
    pq.write_table(pa.table({"b": pa.array(np.arange(n) % 3 == 0)}), path, compression="none", use_dictionary=False, column_encoding={"b": "RLE"}, data_page_version="1.0")
    check_column(path, pa.table({"b": pa.array(np.arange(n) % 3 == 0)}), "b")          # RLE booleans ARE decoded (device bodies); kept here as the boundary case


PAYLOADS = {
    "empty": b"", "one": b"a", "text": b"the quick brown fox jumps over the lazy dog. " * 3000, "rand": np.random.default_rng(1).integers(0, 256, 300_000, dtype=np.uint8).tobytes(),
    "ints": np.arange(100_000, dtype=np.int64).tobytes(), "sorted": np.sort(np.random.default_rng(2).integers(0, 1 << 40, 200_000)).tobytes(),
    "lowcard": np.random.default_rng(3).integers(0, 4, 500_000, dtype=np.uint8).tobytes(), "zeros": bytes(1_000_000),
    "mixed": (b"abc" * 1000 + np.random.default_rng(4).integers(0, 256, 5000, dtype=np.uint8).tobytes()) * 20, "floats": np.random.default_rng(5).normal(size=100_000).tobytes(),
    "runs": np.repeat(np.random.default_rng(6).integers(0, 256, 3000, dtype=np.uint8), 97).tobytes(),
}


@pytest.mark.parametrize("name", sorted(PAYLOADS))
def test_host_zstd_and_lz4_against_the_real_codecs(name):
    """polars_amd/csrc/host_codecs.hpp (written from RFC 8878 / the LZ4 block format) against streams produced by the real libraries:
    zstd levels 1..19 exercise raw / RLE / Huffman literals in one and four streams, predefined / RLE / FSE / repeat sequence tables,
    repeat offsets, multi-block frames."""
    p = PAYLOADS[name]
    for lvl in (1, 3, 7, 12, 19):
        c = pa.Codec("zstd", compression_level=lvl).compress(p, asbytes=True)
        rc, out, err = E.host_codec("zstd", c, len(p))
        assert rc == 0 and out == p, (lvl, err)
    c = pa.Codec("lz4_raw").compress(p, asbytes=True)
    rc, out, err = E.host_codec("lz4_raw", c, len(p))
    assert rc == 0 and out == p, err
    rc, out, err = E.host_codec("lz4_frame", pa.Codec("lz4").compress(p, asbytes=True), len(p))
    assert rc == 0 and out == p, err
    # DEFLATE: gzip members (one and two), zlib wrapper, bare stream, stored blocks, fixed and dynamic Huffman blocks
    import gzip
    import zlib
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    fixed = zlib.compressobj(1, zlib.DEFLATED, -15, strategy=zlib.Z_FIXED)
    for c in (pa.Codec("gzip", compression_level=1).compress(p, asbytes=True), pa.Codec("gzip", compression_level=9).compress(p, asbytes=True), zlib.compress(p, 6),
              zlib.compress(p, 0), gzip.compress(p[:len(p) // 2]) + gzip.compress(p[len(p) // 2:]), raw.compress(p) + raw.flush(), fixed.compress(p) + fixed.flush()):
        rc, out, err = E.host_codec("gzip", c, len(p))
        assert rc == 0 and out == p, err
    # two frames back to back decode as their concatenation; a skippable frame in between is skipped
    if p:
        z = pa.Codec("zstd").compress(p, asbytes=True)
        skip = struct.pack("<II", 0x184D2A53, 5) + b"hello"
        rc, out, err = E.host_codec("zstd", z + skip + z, 2 * len(p))
        assert rc == 0 and out == p + p, err


def test_host_codecs_reject_corrupt_streams():
    rng = np.random.default_rng(77)
    p = PAYLOADS["mixed"]
    for codec, good in (("zstd", pa.Codec("zstd", compression_level=9).compress(p, asbytes=True)), ("lz4_raw", pa.Codec("lz4_raw").compress(p, asbytes=True)),
                        ("lz4_frame", pa.Codec("lz4").compress(p, asbytes=True)), ("gzip", pa.Codec("gzip").compress(p, asbytes=True))):
        outcomes = set()
        for trial in range(300):
            b = bytearray(good)
            for _ in range(1 + trial % 3):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            if trial % 10 == 0:
                b = b[:int(rng.integers(1, len(b)))]
            rc, out, err = E.host_codec(codec, bytes(b), len(p))
            outcomes.add("error" if rc else "ok")
            if rc:
                assert err.split(":")[0] in (codec.split("_")[0], "deflate", "zlib")
        assert "error" in outcomes
        rc, _, _ = E.host_codec(codec, good, len(p) + 1)          # the page header's size is the authority
        assert rc != 0


def test_corrupt_files_are_errors_not_crashes(tmp_path):
    n = 4000
    t = pa.table({"a": pa.array(RNG.integers(0, 9, n), mask=RNG.random(n) < 0.3), "s": pa.array(np.array(["p", "q"])[RNG.integers(0, 2, n)])})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="snappy", data_page_size=512)
    raw = bytearray(open(path, "rb").read())
    bad = str(tmp_path / "bad.parquet")
    outcomes = set()
    for trial in range(60):
        b = bytearray(raw)
        md_len = struct.unpack("<I", b[-8:-4])[0]
        lo, hi = (4, len(b) - 8 - md_len) if trial % 3 else (len(b) - 8 - md_len, len(b) - 8)      # page bytes or footer bytes
        for _ in range(1 + trial % 4):
            b[int(RNG.integers(lo, hi))] ^= 1 << int(RNG.integers(0, 8))
        open(bad, "wb").write(b)
        for col in (0, 1):
            try:
                E.read_column(bad, [0], col)
                outcomes.add("ok")            # a flipped value bit is still a well-formed file
            except E.EmuError as e:
                assert e.code in (1, 3)
                outcomes.add("error")
    assert outcomes == {"ok", "error"}
    open(bad, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(E.EmuError):
        E.read_column(bad, [0], 0)


# ---- Snappy on hand-built streams -------------------------------------------------------------------------------------------------------
def py_unsnap(data: bytes) -> bytes:
    pos, n, shift = 0, 0, 0
    while True:
        b = data[pos]; pos += 1
        n |= (b & 0x7f) << shift; shift += 7
        if not b & 0x80:
            break
    out = bytearray()
    while pos < len(data):
        tag = data[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += data[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | data[pos]; pos += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(data[pos:pos + 2], "little"); pos += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(data[pos:pos + 4], "little"); pos += 4
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n
    return bytes(out)


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7f; n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def lit(payload: bytes) -> bytes:
    n = len(payload) - 1
    if n < 60:
        return bytes([n << 2]) + payload
    nb = max(1, (n.bit_length() + 7) // 8)
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + payload


def copy(off, ln, kind=None):
    if kind is None:
        kind = 1 if (4 <= ln <= 11 and off < 2048) else 2 if off < 65536 else 3
    if kind == 1:
        return bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 255])
    if kind == 2:
        return bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")
    return bytes([3 | ((ln - 1) << 2)]) + off.to_bytes(4, "little")


def random_stream(rng, target):
    body, out_len = bytearray(), 0
    while out_len < target:
        k = rng.integers(0, 10)
        if out_len == 0 or k < 3:
            n = int(rng.choice([1, 2, 5, 59, 60, 61, 255, 256, 257, 3000, 70000])) if k < 2 else int(rng.integers(1, 40))
            body += lit(rng.integers(0, 256, n, dtype=np.uint8).tobytes()); out_len += n
        else:
            off = int(rng.integers(1, min(out_len, 70000) + 1)) if k < 8 else int(rng.integers(1, min(out_len, 8) + 1))    # small offsets: overlapping copies
            ln = int(rng.integers(4, 12)) if k % 2 else int(rng.integers(1, 65))
            kind = None if rng.random() < 0.7 else (3 if True else 2)
            if kind is None and not (4 <= ln <= 11 and off < 2048) and off >= 65536:
                kind = 3
            body += copy(off, ln, kind); out_len += ln
    return varint(out_len) + bytes(body), out_len


@pytest.mark.parametrize("seed", range(12))
def test_snappy_rounds_on_random_element_streams(seed):
    rng = np.random.default_rng(seed)
    data, n = random_stream(rng, int(rng.choice([1, 50, 1000, 5000, 200_000])))
    want = py_unsnap(data)
    for order in (0, 1):
        err, got, rounds, tail = E.snappy(data, n, order)
        assert err == 0 and got == want and rounds >= 1
        assert tail == bytes([0x5A]) * 64           # nothing written past the output
    rc, got = E.snappy_host(data, n)
    assert rc == 0 and got == want


def test_snappy_branch_free_tag_parse_agrees_with_the_branching_one():
    """snappy_tag_x (next / place of the second-generation kernel: the five bytes a tag can span fetched up front, every field a select) against snappy_tag,
    for every tag byte x 2000 pseudo-random tails + the edge values of the length / offset fields, at all four alignments of a window position."""
    assert E.lib().pqemu_snappy_tag_selfcheck(2000, 12345) == 0


def test_snappy_on_pyarrow_compressed_buffers():
    for payload in (b"", b"a", b"abc" * 10000, RNG.integers(0, 256, 100_000, dtype=np.uint8).tobytes(), np.arange(50_000, dtype=np.int64).tobytes(),
                    (b"x" * 70000 + b"y") * 3):
        data = pa.compress(payload, codec="snappy", asbytes=True)
        err, got, rounds, tail = E.snappy(data, len(payload))
        assert err == 0 and got == payload and tail == bytes([0x5A]) * 64
        rc, got = E.snappy_host(data, len(payload))
        assert rc == 0 and got == payload


def test_snappy_corrupt_streams_fail_cleanly():
    RNG = np.random.default_rng(4242)       # own generator: the outcome must not depend on which tests ran before (xdist)
    payload = (b"hello world, " * 500) + RNG.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    good = pa.compress(payload, codec="snappy", asbytes=True)
    n = len(payload)
    cases = [good[:len(good) // 2], good[:-1], varint(n + 1) + good[len(varint(n)):], varint(n) + copy(5, 10, 2), varint(n) + lit(b"ab") + copy(3, 10, 2),
             varint(10) + lit(b"0123456789abcdef"), varint(10) + bytes([63 << 2, 0xff, 0xff, 0xff, 0xff]), b"\xff\xff\xff\xff\xff\xff", varint(4) + bytes([0x01 | (0 << 2)])]
    for trial in range(200):
        b = bytearray(good)
        b[int(RNG.integers(0, len(b)))] ^= 1 << int(RNG.integers(0, 8))
        cases.append(bytes(b))
    errors = 0
    for data in cases:
        err, got, _, tail = E.snappy(data, n)
        assert tail == bytes([0x5A]) * 64
        rc, got_h = E.snappy_host(data, n)
        if err == 0:
            assert rc == 0 and got == got_h        # a flipped literal byte still decodes: both decoders agree on the bytes
        else:
            errors += 1
            assert rc != 0
    assert errors >= 9          # the nine hand-made streams at least; bit flips inside literal bytes still decode


def test_reader_under_address_sanitizer(tmp_path):
    """The same reader, built with -fsanitize=address,undefined as a stand-alone program (tests/emu/parquet_emu_main.cpp), over
    well-formed files, ~150 bit-flipped / truncated copies and corrupt Snappy / zstd streams: error codes, never an out-of-bounds access."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pq_asan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(here, "emu", "parquet_emu_main.cpp")], check=True)
    files = []
    t = mixed_table(3000)
    for i, (comp, ver, dic) in enumerate((("snappy", "1.0", True), ("none", "2.0", True), ("snappy", "2.0", False), ("zstd", "2.0", True), ("lz4", "1.0", True), ("gzip", "1.0", True))):
        p = str(tmp_path / f"good{i}.parquet")
        pq.write_table(t, p, compression=comp, data_page_version=ver, use_dictionary=dic, row_group_size=1100, data_page_size=700)
        files.append(p)
        raw = open(p, "rb").read()
        md_len = struct.unpack("<I", raw[-8:-4])[0]
        for k in range(50):
            b = bytearray(raw)
            lo, hi = (4, len(b) - 8 - md_len) if k % 3 else (len(b) - 8 - md_len, len(b) - 4)
            for _ in range(1 + k % 5):
                b[int(RNG.integers(lo, hi))] ^= 1 << int(RNG.integers(0, 8))
            q = str(tmp_path / f"bad{i}_{k}.parquet")
            open(q, "wb").write(b if k % 10 else b[:int(RNG.integers(8, len(b)))])
            files.append(q)
    payload = (b"hello world, " * 300) + RNG.integers(0, 256, 2000, dtype=np.uint8).tobytes()
    good = pa.compress(payload, codec="snappy", asbytes=True)
    for k in range(120):
        b = bytearray(good)
        if k:
            b[int(RNG.integers(0, len(b)))] ^= 1 << int(RNG.integers(0, 8))
        if k % 7 == 3:
            b = b[:int(RNG.integers(1, len(b)))]
        q = str(tmp_path / f"s{k}.snappy")
        open(q, "wb").write(struct.pack("<I", len(payload)) + bytes(b))
        files.append(q)
    zpayload = payload * 3 + np.sort(RNG.integers(0, 1 << 40, 30_000)).astype(np.int64).tobytes()
    for lvl in (1, 9):
        zgood = pa.Codec("zstd", compression_level=lvl).compress(zpayload, asbytes=True)
        for k in range(100):
            b = bytearray(zgood)
            if k:
                for _ in range(1 + k % 3):
                    b[int(RNG.integers(0, len(b)))] ^= 1 << int(RNG.integers(0, 8))
            if k % 7 == 3:
                b = b[:int(RNG.integers(1, len(b)))]
            q = str(tmp_path / f"z{lvl}_{k}.zst")
            open(q, "wb").write(struct.pack("<I", len(zpayload)) + bytes(b))
            files.append(q)
    for seed in range(6):
        data, n = random_stream(np.random.default_rng(100 + seed), 20_000)
        q = str(tmp_path / f"r{seed}.snappy")
        open(q, "wb").write(struct.pack("<I", n) + data)
        files.append(q)
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    counts = dict(kv.split("=") for kv in r.stdout.split())
    assert int(counts["ok"]) > 100 and int(counts["invalid"]) > 30, r.stdout


@pytest.mark.parametrize("compression", ["none", "snappy", "zstd"])
def test_string_chunks_that_fall_back_from_dictionary_to_plain(tmp_path, compression):
    """A high-cardinality string column: the writer starts every chunk dictionary-encoded and falls back to PLAIN pages once the
    dictionary is full -- both page kinds in one chunk, nulls, long and short (inline) strings, an empty string."""
    rng = np.random.default_rng(12)
    n = 30_000
    vals = np.array([("k%d" % i) if i % 3 else ("a considerably longer key number %d" % i) for i in rng.integers(0, 20_000, n)], dtype=object)
    vals[::97] = ""
    t = pa.table({"s": pa.array(vals, pa.string(), mask=rng.random(n) < 0.15), "b": pa.array([v.encode() for v in vals], pa.binary())})
    path = str(tmp_path / "t.parquet")
    for ver in ("1.0", "2.0"):
        pq.write_table(t, path, compression=compression, data_page_version=ver, dictionary_pagesize_limit=4096, data_page_size=2048, row_group_size=11_000)
        encs = set()
        md = pq.ParquetFile(path).metadata
        for g in range(md.num_row_groups):
            encs |= set(md.row_group(g).column(0).encodings)
        assert "PLAIN" in encs and ("RLE_DICTIONARY" in encs or "PLAIN_DICTIONARY" in encs)
        for name in ("s", "b"):
            r = check_column(path, t, name)
            assert len(r["categories"]) > 5000
        check_column(path, t, "s", row_groups=[2, 0])


def test_snappy_streams_built_around_the_kernel_constants():
    """Hand-built streams whose elements land on the decoder's internal boundaries: tags in the last bytes of the 4096-byte input
    window, rounds that fill exactly (4096 output bytes / 1024 elements), literals of 511 / 512 / 513 bytes (the direct-copy
    threshold), every literal header width, copies whose source ends exactly at a round boundary, long runs of 1-byte literals and
    of offset-1 copies (the deepest pointer chains), 4-byte-offset copies reaching back further than any LDS window."""
    rng = np.random.default_rng(99)
    streams = []

    def build(parts):
        body, out = bytearray(), bytearray()
        for kind, a, b in parts:
            if kind == "lit":
                payload = bytes(a)
                body += lit(payload); out += payload
            else:
                off, ln = a, b
                assert 1 <= off <= len(out)
                body += copy(off, ln, None if (4 <= ln <= 11 and off < 2048) or off < 65536 else 3)
                for _ in range(ln):
                    out.append(out[-off])
        return varint(len(out)) + bytes(body), bytes(out)

    r = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    # 1. tags at every offset of the window's tail: a literal sized so that the next tag starts at 4096 - k
    for k in range(1, 9):
        for after in ("copy2", "copy4", "lit60"):
            first = 4096 - k - 3          # header of a literal this long takes 3 bytes (nb = 2)
            parts = [("lit", r(first), 0)]
            parts.append(("copy", 7, 9) if after == "copy2" else ("copy", 70000 % first + 1, 33) if after == "copy4" else ("lit", r(61), 0))
            parts += [("lit", r(5), 0), ("copy", 3, 20)]
            streams.append(build(parts))
    # 2. rounds that fill exactly: 1024 elements of 4 output bytes; 4096 one-byte literals; then more
    streams.append(build([("lit", r(8), 0)] + [("copy", 8, 4)] * 3000))
    streams.append(build([("lit", r(1), 0) for _ in range(9000)]))
    streams.append(build([("lit", b"ab", 0)] + [("copy", 1, 64)] * 500))          # offset-1 runs: chains as deep as a round
    streams.append(build([("lit", b"xyz", 0)] + [("copy", 3, 5), ("copy", 2, 4), ("copy", 1, 4)] * 2000))
    # 3. literal lengths around the direct-copy threshold and the header widths
    for n in (59, 60, 61, 255, 256, 257, 511, 512, 513, 4095, 4096, 4097, 65535, 65536, 65537, 70001):
        streams.append(build([("lit", r(7), 0), ("copy", 7, 11), ("lit", r(n), 0), ("copy", n, 13), ("lit", r(3), 0)]))
    # 4. copies whose source ends exactly where a round starts / reaches far back (4-byte offsets)
    base = [("lit", r(4096), 0)]
    streams.append(build(base + [("copy", 4096, 64), ("copy", 64, 64), ("copy", 4096 + 128, 64)] * 40))
    streams.append(build([("lit", r(100_000), 0)] + [("copy", 99_000, 64), ("lit", r(2), 0), ("copy", 70_000, 5)] * 300))
    # 5. a stream that alternates long (direct) literals with element-dense stretches
    parts = []
    for i in range(6):
        parts += [("lit", r(3000 + 1111 * i), 0)] + [("copy", 1 + (j % 50), 4 + (j % 60)) for j in range(700)]
    streams.append(build(parts))
    for data, want in streams:
        assert py_unsnap(data) == want
        for order in (0, 1):
            err, got, rounds, tail = E.snappy(data, len(want), order)
            assert err == 0 and got == want and tail == bytes([0x5A]) * 64
        rc, got = E.snappy_host(data, len(want))
        assert rc == 0 and got == want


@pytest.mark.parametrize("compression", ["none", "snappy", "zstd"])
@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_encodings_only_the_host_decoder_knows(tmp_path, compression, version):
    """DELTA_BINARY_PACKED integers, BYTE_STREAM_SPLIT floats, DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY strings (what parquet "v2"
    writers choose), INT96 timestamps (legacy Spark / Impala): decoded by host threads, uploaded as finished columns."""
    rng = np.random.default_rng(21)
    n = 7000
    m = lambda: rng.random(n) < 0.2
    words = np.array(["", "a", "prefix-shared-0001", "prefix-shared-0002", "prefix-shared-and-longer-0003", "zebra", "ünï"])
    t = pa.table({
        "d64": pa.array(np.cumsum(rng.integers(-5, 50, n)), mask=m()), "d32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32)),
        "d64w": pa.array(rng.integers(-2**63, 2**63, n)), "d16": pa.array(rng.integers(-3000, 3000, n).astype(np.int16), mask=m()),
        "du32": pa.array(rng.integers(0, 2**32, n).astype(np.uint32)), "const": pa.array(np.full(n, 7, np.int64)),
        "bss64": pa.array(rng.normal(size=n), mask=m()), "bss32": pa.array(rng.normal(size=n).astype(np.float32)),
        "dl": pa.array(words[rng.integers(0, len(words), n)], mask=m()), "dba": pa.array(np.sort(words[rng.integers(0, len(words), n)])),
        "dts": pa.array(np.cumsum(rng.integers(0, 10**6, n)), pa.timestamp("us")),
    })
    enc = {"d64": "DELTA_BINARY_PACKED", "d32": "DELTA_BINARY_PACKED", "d64w": "DELTA_BINARY_PACKED", "d16": "DELTA_BINARY_PACKED", "du32": "DELTA_BINARY_PACKED",
           "const": "DELTA_BINARY_PACKED", "bss64": "BYTE_STREAM_SPLIT", "bss32": "BYTE_STREAM_SPLIT", "dl": "DELTA_LENGTH_BYTE_ARRAY", "dba": "DELTA_BYTE_ARRAY",
           "dts": "DELTA_BINARY_PACKED"}
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression=compression, data_page_version=version, use_dictionary=False, column_encoding=enc, row_group_size=2500, data_page_size=1500)
    md = pq.ParquetFile(path).metadata
    for i, name in enumerate(t.column_names):
        assert enc[name] in md.row_group(0).column(i).encodings, (name, md.row_group(0).column(i).encodings)
        check_column(path, t, name)
        check_column(path, t, name, row_groups=[2, 0], order=1)


def test_int96_timestamps(tmp_path):
    rng = np.random.default_rng(22)
    n = 5000
    us = rng.integers(-10**15, 2 * 10**15, n)               # before and after the epoch
    us[7] = 32_503_680_000_000_000                          # year 3000: does not fit nanoseconds
    t = pa.table({"ts": pa.array(us, pa.timestamp("us"), mask=rng.random(n) < 0.1), "k": pa.array(np.arange(n))})
    path = str(tmp_path / "t.parquet")
    for dic in (True, False):
        pq.write_table(t, path, use_deprecated_int96_timestamps=True, compression="snappy", use_dictionary=dic, row_group_size=1800, data_page_size=2000)
        assert pq.ParquetFile(path).metadata.row_group(0).column(0).physical_type == "INT96"
        r = E.read_column(path, [0, 1, 2], 0)
        assert r["dtype"] == 4 and r["logical"] == 6           # Int64 / Datetime[ns]: the reference's default for INT96 (schema/mod.rs:32)
        valid = np.array([x is not None for x in t.column("ts").to_pylist()])
        want = us * 1000
        want[7] = np.iinfo(np.int64).max                       # int96_to_i64_ns(..).unwrap_or(i64::MAX), simple.rs:731-733
        assert np.array_equal(r["valid"], valid) and np.array_equal(r["values"][valid], want[valid])


IO_FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_files")
REFERENCE_FILES = sorted(f for f in os.listdir(IO_FILES) if f.endswith(".parquet"))


@pytest.mark.parametrize("name", REFERENCE_FILES)
def test_the_reference_s_own_parquet_fixtures(name):
    """tests/golden/io_files: the Parquet files the reference's I/O tests read (py-polars/tests/unit/io/files, docs/assets/data; copied by
    tests/golden/make_io_files.py), written by Polars' own writer, parquet-mr, parquet-cpp and Impala -- zstd, gzip, snappy and LZ4_RAW pages, v2 pages, an all-null v2 page without value bytes (test_parquet.py:848), INT96 and nanosecond timestamps,
    un-annotated binary strings, deprecated BIT_PACKED levels in the encoding lists.  Every flat column through the product's reader
    (kernel bodies run on the CPU) equals pyarrow's decode; nested columns are refused by name."""
    import hashlib
    import json
    from polars_amd import io
    path = os.path.join(IO_FILES, name)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == json.load(open(os.path.join(IO_FILES, "INDEX.json")))[name]["sha256"]
    table = pq.read_table(path)
    dec = io._DeviceDecoder(path)                               # the library's metadata reader: leaf names and indices
    assert dec.num_rows == table.num_rows
    flat = [n for n in table.column_names if n in dec._info]
    assert flat or name.startswith("nested")
    for n in flat:
        check_column(path, table, n, ci=dec._info[n][0])
        for order in (1,):                                      # lanes in descending order too
            check_column(path, table, n, order=order, ci=dec._info[n][0])
    nested = [n for n in dec.names if n not in table.column_names]
    for n in nested:
        with pytest.raises(E.EmuError) as ei:
            E.read_column(path, [0], dec._info[n][0])
        assert ei.value.code == 3 and "nested" in str(ei.value)
    if name == "tz_aware.parquet":
        assert dec.dtype("UTC_DATETIME_ID").time_unit == "ns" and dec.dtype("UTC_DATETIME_ID").time_zone == "UTC"        # Datetime("ns", "UTC"), as the reference reads it
    if name == "alltypes_plain.parquet":
        assert dec.dtype("timestamp_col").time_unit == "ns" and dec._info["string_col"][2] == 4      # INT96 -> Datetime[ns]; BYTE_ARRAY without annotation -> Binary
    if name == "empty_datapage_v2.snappy.parquet":
        r = E.read_column(path, [0], 0)
        assert r["null_count"] == 1 and r["valid"].tolist() == [False]


@pytest.mark.parametrize("compression", ["zstd", "gzip", "lz4"])
def test_host_inflate_batches(tmp_path, compression, monkeypatch):
    """Host-codec pages are inflated column-wide, in batches of consecutive chunks (parquet_reader.hpp: kInflateBatch).  With the batch
    shrunk to 40 KB a 12-row-group file crosses many batch borders (single-chunk batches, multi-chunk batches, the last partial one)."""
    monkeypatch.setenv("PLX_PARQUET_INFLATE_BATCH", "40000")
    n = 60_000
    t = mixed_table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression=compression, row_group_size=5000, data_page_size=3000)
    for name in ("i64", "f64", "s", "b", "i8"):
        if name in t.column_names:
            check_column(path, t, name)
    check_column(path, t, t.column_names[0], row_groups=[11, 0, 5, 6, 7])


@pytest.mark.parametrize("version,dictionary", [("1.0", True), ("2.0", True), ("2.0", False)])
def test_snappy_pages_through_the_host_threads(tmp_path, monkeypatch, version, dictionary):
    """PLX_PARQUET_SNAPPY=host: Snappy pages take the column-wide host inflate like zstd pages (an experiment switch: which side wins
    depends on the host's core count); same results, no device Snappy stream."""
    monkeypatch.setenv("PLX_PARQUET_SNAPPY", "host")
    n = 9000
    t = mixed_table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="snappy", data_page_version=version, use_dictionary=dictionary, row_group_size=2500, data_page_size=2000)
    for name in t.column_names:
        r = check_column(path, t, name)
        assert r["stats"]["snappy_streams"] == 0, name
    monkeypatch.delenv("PLX_PARQUET_SNAPPY")
    assert check_column(path, t, "const")["stats"]["snappy_streams"] > 0       # a compressible column: its pages are Snappy streams for the kernel again


def test_second_generation_snappy_bodies(tmp_path, monkeypatch):
    """pq_snappy_kernel_v2 (PLX_SNAPPY_KERNEL=2; batched LDS loads in next / mark / rank, parquet_snappy.hpp): the same streams, both
    lane orders, the same results as the first generation -- random element streams, real Snappy output, corrupt streams, a whole file."""
    monkeypatch.setenv("PLX_SNAPPY_KERNEL", "2")
    for seed in range(12):
        data, n = random_stream(np.random.default_rng(500 + seed), 30_000)
        want = py_unsnap(data)
        for order in (0, 1):
            err, got, rounds, tail = E.snappy(data, n, order)
            assert err == 0 and got == want and tail == bytes([0x5A]) * 64, (seed, order)
    rng = np.random.default_rng(77)
    for payload in (b"", b"a", b"abc" * 10000, rng.integers(0, 256, 100_000, dtype=np.uint8).tobytes(), np.arange(50_000, dtype=np.int64).tobytes(), (b"x" * 70000 + b"y") * 3):
        data = pa.compress(payload, codec="snappy", asbytes=True)
        err, got, _, tail = E.snappy(data, len(payload))
        assert err == 0 and got == payload and tail == bytes([0x5A]) * 64
    good = pa.compress((b"hello world, " * 500) + rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(), codec="snappy", asbytes=True)
    n = 13 * 500 + 3000
    outcomes = []
    for trial in range(150):
        b = bytearray(good)
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        monkeypatch.setenv("PLX_SNAPPY_KERNEL", "2")
        e2, g2, _, tail = E.snappy(bytes(b), n)
        assert tail == bytes([0x5A]) * 64
        monkeypatch.setenv("PLX_SNAPPY_KERNEL", "1")
        e1, g1, _, _ = E.snappy(bytes(b), n)
        assert (e1 == 0) == (e2 == 0) and (e1 != 0 or g1 == g2)          # the same verdict, and the same bytes when there is no error
        outcomes.append(e2 != 0)
    assert sum(outcomes) >= 3                                           # most flips land in literal bytes: wrong bytes, no error
    monkeypatch.setenv("PLX_SNAPPY_KERNEL", "2")
    t = mixed_table(7000)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="snappy", row_group_size=3000, data_page_size=2500)
    for name in t.column_names:
        check_column(path, t, name, order=1)


def test_long_snappy_dictionary_pages_are_inflated_by_host_threads(tmp_path, monkeypatch):
    """A Snappy dictionary page of hundreds of kilobytes is ONE stream = one workgroup of the device kernel, 4 KB a round: the launch lasts as long as that stream (round-5
    review, weak 6).  From 192 KB (PLX_PARQUET_HOST_DICT_BYTES) the reader hands such a page to a host thread while it walks on; the plain values follow the chunk in one upload
    and the data pages still go through the device kernel.  Same values either way."""
    rng = np.random.default_rng(8)
    n = 120_000
    vals = rng.integers(0, 60_000, n).astype(np.int64) * 1_000_003            # 60 000 distinct values: a 480 KB dictionary page per row group
    small = rng.integers(0, 50, n).astype(np.int32)
    nullable = pa.array(vals, mask=rng.random(n) < 0.1)
    t = pa.table({"big": vals, "small": small, "bign": nullable})
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path, compression="snappy", row_group_size=60_000, dictionary_pagesize_limit=4 << 20)
    r = check_column(path, t, "big")
    assert r["stats"]["host_inflated_pages"] == 2 and r["stats"]["snappy_streams"] > 0, r["stats"]        # two row groups: two long dictionary pages on the host, the data pages on the device
    assert check_column(path, t, "bign")["stats"]["host_inflated_pages"] == 2
    assert check_column(path, t, "small")["stats"]["host_inflated_pages"] == 0                          # a 200-byte dictionary stays a device stream
    monkeypatch.setenv("PLX_PARQUET_HOST_DICT_BYTES", "0")
    r0 = check_column(path, t, "big")
    assert r0["stats"]["host_inflated_pages"] == 0 and r0["stats"]["snappy_streams"] == r["stats"]["snappy_streams"] + 2


# ---- zstd on the device: the bodies of pq_zstd_entropy / pq_zstd_execute behind the host index pass (parquet_zstd.hpp, parquet_zstd_index.hpp) -------------------


ZPAYLOADS = dict(PAYLOADS)
ZPAYLOADS.update({
    "sorted keys": np.sort(np.random.default_rng(11).integers(0, 1 << 40, 150_000)).astype(np.int64).tobytes(),          # a match per value that reads what the previous match wrote
    "zeros": bytes(300_000),                                                                                             # RLE blocks, a 128 KB match at offset 1
    "random": np.random.default_rng(12).integers(0, 256, 300_000, dtype=np.uint8).tobytes(),                             # raw blocks
    "far matches": (np.random.default_rng(13).integers(0, 256, 70_000, dtype=np.uint8).tobytes()) * 5,                   # matches 70 KB back: beyond the LDS ring, across blocks
    "prices": np.round(np.random.default_rng(14).uniform(900, 105_000, 120_000), 2).tobytes(),                           # Huffman literals in four streams, few matches
    "codes": np.random.default_rng(15).integers(0, 7, 400_000, dtype=np.uint8).tobytes(),                                # short codes: a small Huffman table
    "one byte": b"x",
})


def test_zstd_sequence_code_tables():
    assert E.lib().pqemu_zstd_code_selfcheck() == 0


@pytest.mark.parametrize("name", sorted(ZPAYLOADS))
def test_zstd_device_bodies_against_the_real_codec(name):
    """Streams of the real library, levels 1..19 (raw / RLE / Huffman literals in one and four streams, predefined / RLE / FSE / repeat sequence tables, treeless literals,
    repeat offsets across blocks, multi-block frames, matches beyond the LDS ring), lanes in ascending and descending order."""
    p = ZPAYLOADS[name]
    for lvl in (1, 3, 7, 12, 19):
        c = pa.Codec("zstd", compression_level=lvl).compress(p, asbytes=True)
        for order in (0, 1):
            rc, out, counts, err = E.zstd_device(c, len(p), order)
            assert rc == 0 and out == p, (lvl, order, rc, err, counts)
    if p:
        # two frames back to back decode as their concatenation (repeat offsets and the window start over); a skippable frame in between is skipped
        z = pa.Codec("zstd").compress(p, asbytes=True)
        skip = struct.pack("<II", 0x184D2A53, 5) + b"hello"
        rc, out, counts, err = E.zstd_device(z + skip + z, 2 * len(p))
        assert rc == 0 and out == p + p, (rc, err)


def test_zstd_sequence_bit_streams_of_every_length():
    """The sequence bit stream of a block is staged through LDS 256 words at a time, downwards; where the last stagings fall depends on the stream's length (a stream
    whose next-to-last staging started at word 1 was once taken for finished).  ~160 KB pages of prices and of sorted keys, their lengths swept: bit streams of 2 .. 60 KB."""
    rng = np.random.default_rng(606)
    price = np.round(rng.uniform(900, 105_000, 30_000), 2)
    keys = np.sort(rng.integers(0, 1 << 40, 30_000))
    for src in (price, keys):
        for m in list(range(18_000, 21_000, 41)) + list(range(300, 6_000, 173)):
            p = src[:m].tobytes()
            c = pa.Codec("zstd", compression_level=1).compress(p, asbytes=True)
            rc, out, counts, err = E.zstd_device(c, len(p))
            assert rc == 0 and out == p, (m, rc, counts, err)


def test_zstd_device_agrees_with_the_host_decoder_on_corrupt_streams():
    """Bit flips and truncations: whatever the host decoder (host_codecs.hpp) makes of a stream, the device bodies make the same -- the same bytes or an error --
    and the page header's size stays the authority."""
    rng = np.random.default_rng(78)
    p = ZPAYLOADS["mixed"] + ZPAYLOADS["sorted keys"][:200_000]
    outcomes = set()
    for lvl in (1, 9):
        good = pa.Codec("zstd", compression_level=lvl).compress(p, asbytes=True)
        for trial in range(150):
            b = bytearray(good)
            for _ in range(1 + trial % 3):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            if trial % 10 == 0:
                b = b[:int(rng.integers(1, len(b)))]
            rc_h, out_h, _ = E.host_codec("zstd", bytes(b), len(p))
            rc_d, out_d, _, err = E.zstd_device(bytes(b), len(p))
            assert (rc_h == 0) == (rc_d == 0), (lvl, trial, rc_h, rc_d, err)
            if rc_h == 0:
                assert out_h == out_d
            outcomes.add("error" if rc_d else "ok")
        rc, _, _, _ = E.zstd_device(good, len(p) + 1)
        assert rc != 0
        rc, _, _, _ = E.zstd_device(good, len(p) - 1)
        assert rc != 0
    assert "error" in outcomes


@pytest.mark.parametrize("version,dictionary", [("1.0", True), ("2.0", True), ("2.0", False)])
def test_zstd_pages_take_the_device_passes(tmp_path, monkeypatch, version, dictionary):
    """zstd pages are streams for the device passes (every page a stream, its blocks counted); PLX_PARQUET_ZSTD=host sends them through the host threads instead: same
    results."""
    n = 9000
    t = mixed_table(n)
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", data_page_version=version, use_dictionary=dictionary, row_group_size=2500, data_page_size=2000)
    seen = 0
    for name in t.column_names:
        r = check_column(path, t, name)
        assert r["stats"]["host_inflated_pages"] == 0, name
        seen += r["stats"]["zstd_streams"]
        assert r["stats"]["zstd_blocks"] >= r["stats"]["zstd_streams"]
    assert seen > 0
    monkeypatch.setenv("PLX_PARQUET_ZSTD", "host")
    for name in t.column_names:
        r = check_column(path, t, name)
        assert r["stats"]["zstd_streams"] == 0, name


def test_long_zstd_dictionary_pages_are_inflated_by_host_threads(tmp_path, monkeypatch):
    """The same switch for zstd: a page's execute pass is one wavefront walking its sequences in order, and a 480 KB dictionary of sorted values is tens of thousands of
    them.  The data pages stay streams of the device passes; with the switch off the dictionary pages are streams too."""
    rng = np.random.default_rng(8)
    n = 120_000
    vals = np.sort(rng.integers(0, 60_000, n)).astype(np.int64) * 1_000_003
    t = pa.table({"big": vals, "small": rng.integers(0, 50, n).astype(np.int32)})
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path, compression="zstd", row_group_size=60_000, dictionary_pagesize_limit=4 << 20)
    r = check_column(path, t, "big")
    assert r["stats"]["host_inflated_pages"] == 2 and r["stats"]["zstd_streams"] > 0, r["stats"]
    assert check_column(path, t, "small")["stats"]["host_inflated_pages"] == 0
    monkeypatch.setenv("PLX_PARQUET_HOST_DICT_BYTES", "0")
    r0 = check_column(path, t, "big")
    assert r0["stats"]["host_inflated_pages"] == 0 and r0["stats"]["zstd_streams"] == r["stats"]["zstd_streams"] + 2


def test_zstd_match_chains_resolved_before_they_are_copied():
    """The execute pass' resolve-then-copy path (a batch whose matches copy what earlier matches of the batch wrote: sorted keys, dictionaries in first-appearance order)
    next to batches it must leave to the in-order loop: overlapping matches (runs), matches across a literal run and a match (text), long matches.  Interleaved
    in one stream so that batches of both kinds follow each other and share the LDS ring."""
    rng = np.random.default_rng(21)
    keys = np.sort(rng.integers(0, 1 << 40, 40_000)).astype(np.int64).tobytes()
    firsts = rng.integers(0, 1 << 30, 40_000).astype(np.int64).tobytes()
    runs = np.repeat(rng.integers(0, 256, 2000, dtype=np.uint8), 37).tobytes()
    text = b"the quick brown fox jumps over the lazy dog. " * 900
    p = b"".join(x[i * 8000:(i + 1) * 8000] for i in range(5) for x in (keys, runs, firsts, text))
    for lvl in (1, 3, 9):
        c = pa.Codec("zstd", compression_level=lvl).compress(p, asbytes=True)
        for order in (0, 1):
            rc, out, counts, err = E.zstd_device(c, len(p), order)
            assert rc == 0 and out == p, (lvl, order, rc, counts, err)


def test_zstd_batches_of_fixed_width_rows():
    """The execute pass' row path: sequences "a few literal bytes, then the rest of the value from the value before it" (offset = the value's width; a repeated value makes the
    match a width longer).  Sorted 8-byte keys with duplicates, 4-byte running sums, 16-byte rows (two keys side by side), 2-byte values, and widths the path must refuse
    (24-byte rows at offset 8, values of 3 bytes) -- each alone and spliced together so that batches of both kinds follow each other."""
    rng = np.random.default_rng(31)
    n = 30_000
    k8 = np.sort(rng.integers(1, 2 * n, n)).astype(np.int64)                        # many equal neighbours
    k4 = np.cumsum(rng.integers(0, 3, 2 * n)).astype(np.int32)
    k16 = np.stack([np.sort(rng.integers(1, 1 << 33, n)), np.sort(rng.integers(1, 4 * n, n))], axis=1).astype(np.int64)
    k2 = np.cumsum(rng.integers(0, 2, 3 * n)).astype(np.uint16)
    k24 = np.stack([k8, k8 // 3, k8 // 7], axis=1)
    k3 = np.sort(rng.integers(0, 1 << 20, 2 * n)).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3].copy()
    parts = [x.tobytes() for x in (k8, k4, k16, k2, k24, k3)]
    spliced = b"".join(p[i * 20_000:(i + 1) * 20_000] for i in range(6) for p in parts)
    for p in parts + [spliced]:
        for lvl in (1, 3):
            c = pa.Codec("zstd", compression_level=lvl).compress(p, asbytes=True)
            for order in (0, 1):
                rc, out, counts, err = E.zstd_device(c, len(p), order)
                assert rc == 0 and out == p, (lvl, order, rc, counts, err)


def test_zstd_matches_whose_source_has_left_the_ring():
    """Values repeated from a few rows up (in the LDS ring) and from tens of kilobytes up (flushed: fetched from HBM by the match's own lane next to the literals), in
    the same batches: a few distinct doubles / integers in 8-byte slots, 200 KB and more of them, levels that search far."""
    rng = np.random.default_rng(41)
    n = 60_000
    disc = (rng.integers(0, 11, n) / 100.0).tobytes()
    qty = rng.integers(1, 51, n).astype(np.int64).tobytes()
    codes = np.tile(rng.integers(0, 1 << 50, 3000), n // 3000).astype(np.int64)
    codes[rng.integers(0, n, n // 50)] = rng.integers(0, 1 << 50, n // 50)         # a long period (24 KB) with scattered changes: matches of every length, all far
    for p in (disc, qty, codes.tobytes(), disc[:100_000] + qty[:100_000] + codes.tobytes()[:100_000]):
        for lvl in (1, 3, 9, 15):
            c = pa.Codec("zstd", compression_level=lvl).compress(p, asbytes=True)
            for order in (0, 1):
                rc, out, counts, err = E.zstd_device(c, len(p), order)
                assert rc == 0 and out == p, (lvl, order, rc, counts, err)


def test_zstd_pages_of_many_sequences_go_to_host_threads(tmp_path, monkeypatch):
    """A page's execute pass is one wavefront walking its sequences in order: a page with more of them than PLX_PARQUET_ZSTD_HOST_SEQS (default 50 000; the index pass
    counts them) is inflated by a host thread instead while the column's other pages stay device streams.  Same values with the limit at 300 (most pages of the sorted
    column to the host), at its default (none of these 2 000-row pages) and switched off."""
    n = 40_000
    t = pa.table({"k": pa.array(np.sort(RNG.integers(1, 4 * n, n))), "f": pa.array(RNG.normal(size=n)), "few": pa.array(RNG.integers(0, 5, n), mask=RNG.random(n) < 0.2)})
    path = str(tmp_path / "t.parquet")
    pq.write_table(t, path, compression="zstd", use_dictionary=False, row_group_size=15_000, max_rows_per_page=2000)
    base = {name: check_column(path, t, name)["stats"] for name in t.column_names}
    assert all(st["host_inflated_pages"] == 0 and st["zstd_streams"] > 0 for st in base.values()), base
    monkeypatch.setenv("PLX_PARQUET_ZSTD_HOST_SEQS", "300")
    st = {name: check_column(path, t, name)["stats"] for name in t.column_names}
    assert st["k"]["host_inflated_pages"] > 0 and st["k"]["host_inflated_pages"] + st["k"]["zstd_streams"] == base["k"]["zstd_streams"], (st["k"], base["k"])
    assert st["f"]["host_inflated_pages"] == 0            # doubles: Huffman literals, hardly a sequence
    monkeypatch.setenv("PLX_PARQUET_ZSTD_HOST_SEQS", "0")
    assert check_column(path, t, "k")["stats"]["host_inflated_pages"] == 0
