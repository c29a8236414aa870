"""String columns of file scans in predicates and group keys: a literal compared with a scanned column on the first collect (the dictionary exists only after
the scan was materialised), hive string partition keys, Datetime[ns] statistics against plain numbers, concat of lazy scans grouped by a string key."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

pytestmark = pytest.mark.gpu


def test_string_literal_filter_on_a_scanned_column_first_collect(pl, tmp_path):
    """Round-2 advisor finding (high): a string literal compared with a string column of a FILE SCAN was lowered to a dictionary code
    before the file had been read (empty dictionary -> code 0 = the first category).  The first collect() must already be right.
    Reference behaviour: categorical.rs rev-map lookup happens on the materialised column."""
    rng = np.random.default_rng(5)
    n = 20_000
    words = np.array(["a", "b", "c", "d"])
    s = words[rng.integers(0, 4, n)]
    v = rng.integers(0, 100, n)
    path = str(tmp_path / "s.parquet")
    pq.write_table(pa.table({"s": pa.array(s), "v": pa.array(v)}), path, row_group_size=5000)
    c = pl.col
    for lit in ("b", "a", "zzz"):
        out = pl.scan_parquet(path).filter(c("s") == lit).select(c("v").sum().alias("sv"), pl.len().alias("n")).collect()   # a FRESH scan each time: first collect
        assert out["n"].to_list() == [int((s == lit).sum())], lit
        assert out["sv"].to_list() == [int(v[s == lit].sum())], lit
    out = pl.scan_parquet(path).filter(c("s") != "b").select(pl.len().alias("n")).collect()
    assert out["n"].to_list() == [int((s != "b").sum())]
    out = pl.scan_parquet(path).filter(c("s").is_in(["b", "d", "nope"])).select(pl.len().alias("n")).collect()
    assert out["n"].to_list() == [int(np.isin(s, ["b", "d"]).sum())]
    lf = pl.scan_parquet(path).filter(c("s") == "c").select(pl.len().alias("n"))
    assert lf.collect()["n"].to_list() == lf.collect()["n"].to_list() == [int((s == "c").sum())]     # second collect of the same plan agrees


def test_hive_string_key_filter_and_null_partition(pl, tmp_path):
    want = []
    for seg, name in (("A", "seg=A"), ("B%20x", "seg=B%20x"), (None, "seg=__HIVE_DEFAULT_PARTITION__")):
        os.makedirs(tmp_path / name)
        k = np.arange(1000) + len(want)
        pq.write_table(pa.table({"k": k}), str(tmp_path / name / "p.parquet"))
        want += [(int(x), None if seg is None else seg.replace("%20", " ")) for x in k]
    c = pl.col
    df = pl.read_parquet(str(tmp_path))
    assert sorted(zip(df["k"].to_list(), [x or "" for x in df["seg"].to_list()])) == sorted((k, s or "") for k, s in want)
    out = pl.scan_parquet(str(tmp_path)).filter(c("seg") == "B x").select(c("k").sum().alias("sk"), pl.len().alias("n")).collect()
    assert out["n"].to_list() == [1000] and out["sk"].to_list() == [sum(k for k, s in want if s == "B x")]
    out = pl.scan_parquet(str(tmp_path)).filter(c("seg").is_null()).select(pl.len().alias("n")).collect()
    assert out["n"].to_list() == [1000]


def test_ns_datetime_column_against_int_literal_is_not_pruned_wrongly(pl, tmp_path):
    """Advisor (medium): row-group statistics of a Datetime[ns] column are reported in microseconds; a plain integer literal is in ns."""
    n = 4000
    ts = (np.arange(n, dtype=np.int64) * 1_000_000_007 + 1_600_000_000_000_000_000)
    path = str(tmp_path / "ns.parquet")
    pq.write_table(pa.table({"ts": pa.array(ts, pa.timestamp("ns")), "v": np.arange(n)}), path, row_group_size=1000, version="2.6")
    lit = int(ts[3500])
    out = pl.scan_parquet(path).filter(pl.col("ts") > lit).select(pl.len().alias("n")).collect()
    assert out["n"].to_list() == [n - 3501]


def test_concat_lazy_group_by_string_key(pl):
    """Round-2 GPUTEST failure: the union dictionary of a concat's inputs is only known after they are collected."""
    a = pl.DataFrame({"k": np.arange(5), "s": pl.Series.from_arrow("s", pa.array(["x", "y", None, "x", "y"])), "f": np.arange(5) * 0.5})
    b = pl.DataFrame({"k": np.arange(5, 9), "s": pl.Series.from_arrow("s", pa.array(["z", "y", "z", None])), "f": np.arange(4) * 2.0})
    c = pl.col
    out = pl.concat([a.lazy().filter(c("k") >= 3), b.lazy()]).group_by("s").agg(c("f").sum().alias("sf"), pl.len().alias("n")).collect().sort_host("s")
    assert out["s"] == ["x", "y", "z", None] and out["n"] == [1, 2, 2, 1] and out["sf"] == [1.5, 4.0, 4.0, 6.0]
    out = pl.concat([a.lazy(), b.lazy()]).filter(c("s") == "z").select(pl.len().alias("n")).collect()
    assert out["n"].to_list() == [2]
