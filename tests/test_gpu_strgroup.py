"""The string-key group-by operator (plx_strview_groupby; kernels_strgroup.hip) -- group_by(<Utf8View column held as
views>).agg(sum / mean / count / len) without a dictionary-encode pass.  Reference semantics: crates/polars-expr/src/hash_keys.rs:413-452
(BinviewKeys), crates/polars-compute/src/binview_index_map.rs.  Checked against numpy on the generator's host twin and against the
encode-then-group route of the same library (which the earlier rounds pinned to the oracle)."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu


def _views_series(pl, strings):
    """Host strings (all <= 12 bytes) -> UInt64 Series of 2 n view words in HBM, as a scan source would leave them."""
    arr = pa.array(strings, pa.string_view())
    assert arr.null_count == 0
    raw = np.frombuffer(arr.buffers()[1], dtype=np.uint64, count=2 * len(strings), offset=16 * arr.offset)
    return pl.Series("views", raw.copy(), pl.UInt64)


def _by_key(out, key="k"):
    d = out.to_dict()
    order = sorted(range(len(d[key])), key=lambda i: d[key][i])
    return {c: [d[c][i] for i in order] for c in d}


def test_string_key_group_by_on_views_matches_numpy_and_the_encoded_route(pl):
    from polars_amd import datagen
    n, seed, n_keys = 60_000_001, 12, 200_000          # ~76 tiles per workgroup: every partition's lines cross into a second chunk
    views = datagen.id_views_native(pl, "k", n, seed, 0, 1, n_keys + 1)
    v = datagen.uniform_native(pl, "v", pl.Float64, n, seed, 1, 0, 10 ** 9, 1e-7)
    k = pl.Series.from_device_views("k", views, encode="deferred")
    assert k._is_raw_views() and len(k) == n
    q = lambda key: pl.DataFrame([key, v]).lazy().group_by("k").agg(pl.col("v").sum().alias("v_sum"), pl.col("v").mean().alias("v_mean"), pl.col("v").count().alias("c"), pl.len())
    out = q(k).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    assert k._is_raw_views()                                                      # the key column was never encoded
    assert list(out.schema) == ["k", "v_sum", "v_mean", "c", "len"] and out.schema["v_sum"] == pl.Float64 and out.schema["len"] == pl.UInt32
    ids = datagen.uniform_native_host("Int64", 0, n, seed, 0, 1, n_keys + 1)
    vals = datagen.uniform_native_host("Float64", 0, n, seed, 1, 0, 10 ** 9, 1e-7)
    s, c = np.bincount(ids, weights=vals, minlength=n_keys + 1), np.bincount(ids, minlength=n_keys + 1)
    present = np.nonzero(c)[0]
    got = _by_key(out)
    assert got["k"] == ["id%010d" % i for i in present]
    assert np.allclose(got["v_sum"], s[present], rtol=1e-9) and np.allclose(got["v_mean"], s[present] / c[present], rtol=1e-9)
    assert got["c"] == c[present].tolist() and got["len"] == c[present].tolist()
    # the same query through dictionary encoding + the dense-id group-by
    ref = _by_key(q(pl.Series.from_device_views("k", views)).collect())
    assert "StringViewGroupBy" not in pl.last_plan()
    assert ref["k"] == got["k"] and ref["c"] == got["c"] and ref["len"] == got["len"] and np.allclose(ref["v_sum"], got["v_sum"], rtol=1e-9)


def test_string_key_group_by_nulls_int_values_short_and_empty_strings(pl, monkeypatch):
    monkeypatch.setenv("PLX_STRGROUP_FORCE", "1")           # 14 distinct strings: the operator would leave so few groups to the usual route
    rng = np.random.default_rng(5)
    words = ["", "a", "b", "ab", "ba", "a\0", "abcdefghijkl", "abcdefghijkm", "Abcdefghijkl", "xbcdefghijkl", "ünï", "twelve bytes", "0", "00"]
    n = 300_011
    idx = rng.integers(0, len(words), n)
    idx[rng.random(n) < 0.3] = 6                                                 # a hot key: same-address LDS atomics
    strings = [words[i] for i in idx]
    x = rng.integers(-10 ** 12, 10 ** 12, n)
    valid = rng.random(n) > 0.2
    valid[idx == 3] = False                                                       # one group without a single valid value
    v = pl.Series("v", x, pl.Int64, validity=valid)
    k = pl.Series.from_device_views("k", _views_series(pl, strings), encode="deferred")
    out = pl.DataFrame([k, v]).lazy().group_by("k").agg(pl.col("v").sum(), pl.col("v").mean().alias("m"), pl.col("v").count().alias("c"), pl.len().alias("n")).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    assert out.schema["v"] == pl.Int64
    got = _by_key(out)
    want = sorted(set(strings))
    assert got["k"] == want
    for j, w in enumerate(want):
        m = np.array([s == w for s in strings])
        assert got["n"][j] == int(m.sum()) and got["c"][j] == int((m & valid).sum())
        assert got["v"][j] == int(x[m & valid].sum())                             # integer sums are exact; 0 for the all-null group
        if (m & valid).any():
            assert abs(got["m"][j] - x[m & valid].mean()) <= 1e-9 * max(1.0, abs(x[m & valid].mean()))
        else:
            assert got["m"][j] is None
    # len only needs no value column in the reference; here the operator wants one -> the usual route, same rows
    out2 = pl.DataFrame([pl.Series.from_device_views("k", _views_series(pl, strings), encode="deferred"), v]).lazy().group_by("k").agg(pl.len().alias("n")).collect()
    assert "StringViewGroupBy" not in pl.last_plan()
    assert _by_key(out2)["n"] == got["n"]


def test_string_key_group_by_declines_long_strings_and_many_groups(pl):
    F = pl._ffi
    import ctypes as C
    # a string of 13 bytes: the view is not the string -> PLX_ERR_UNSUPPORTED from the ABI, and the deferred column encodes instead
    strings = ["short", "thirteen byte", "short", "x"] * 1000
    arr = pa.array(strings, pa.string_view())
    raw = np.frombuffer(arr.buffers()[1], dtype=np.uint64, count=2 * len(strings)).copy()
    views = pl.Series("views", raw, pl.UInt64)
    v = pl.Series("v", np.arange(len(strings), dtype=np.float64), pl.Float64)
    hs = [C.c_uint64() for _ in range(5)]
    st = F.lib().plx_strview_groupby(views._h, v._h, *[C.byref(h) for h in hs])
    assert st == F.ERR_UNSUPPORTED and b"12 bytes" in F.lib().plx_last_error()
    # more distinct strings than the LDS tables hold (512 partitions x 2816 groups at 80 %): declined on the sample estimate, the usual route answers
    from polars_amd import datagen
    n, n_keys = 4_000_000, 3_000_000
    views = datagen.id_views_native(pl, "k", n, 3, 0, 1, n_keys + 1)
    vv = datagen.uniform_native(pl, "v", pl.Float64, n, 3, 1, 0, 1000, 1e-3)
    k = pl.Series.from_device_views("k", views, encode="deferred")
    out = pl.DataFrame([k, vv]).lazy().group_by("k").agg(pl.col("v").sum(), pl.len()).collect()
    assert "StringViewGroupBy" not in pl.last_plan() and not k._is_raw_views()
    ids = datagen.uniform_native_host("Int64", 0, n, 3, 0, 1, n_keys + 1)
    assert out.height == len(np.unique(ids)) and sum(out["len"].to_list()) == n


def test_deferred_views_column_behaves_like_the_encoded_one_elsewhere(pl):
    strings = ["b", "a", "c", "a", "b", "a"]
    k = pl.Series.from_device_views("k", _views_series(pl, strings), encode="deferred")
    assert len(k) == 6 and k.rename("z").name == "z" and k._is_raw_views()
    df = pl.DataFrame([k, pl.Series("v", [1.0, 2.0, 3.0, 4.0, 5.0, 6.0])])
    assert df.height == 6 and k._is_raw_views()
    assert df.lazy().filter(pl.col("k") == "a").select(pl.col("v").sum()).collect().to_dict() == {"v": [12.0]}       # any other operator encodes on first use
    assert not k._is_raw_views() and isinstance(k.dtype, pl.Categorical) and k.to_list() == strings
    with pytest.raises(ValueError):
        pl.Series.from_device_views("k", _views_series(pl, strings), encode="later")


def _operator_cases():
    """The reference's own string-key group_by vectors (tests/golden, from py-polars/tests/unit/operations/test_group_by.py and aggregation/
    test_aggregations.py) that the operator can serve: one string key without nulls, strings of at most 12 bytes, one Int64 / Float64 value column; of
    the case's aggregates those the operator knows (sum / mean / count / len) -- the others are checked on the encoded route by test_gpu_strview.py."""
    from tests import kat
    out = []
    for c in kat.load_cases("groupby"):
        if list(c["key_dtypes"].values()) != ["str"] or len(c["values"]) != 1:
            continue
        (vname, vdt), = c["value_dtypes"].items()
        strings = kat.expand(next(iter(c["keys"].values())))
        aggs = [(col, op) for col, op in c["aggs"] if op in ("sum", "mean", "count", "len")]
        if vdt not in ("i64", "f64") or not aggs or any(s is None or len(s.encode()) > 12 for s in strings):
            continue
        out.append((c, aggs))
    return out


@pytest.mark.parametrize("case,aggs", _operator_cases(), ids=lambda x: x["id"] if isinstance(x, dict) else "")
def test_reference_string_key_group_by_vectors_through_the_operator(pl, case, aggs, monkeypatch):
    monkeypatch.setenv("PLX_STRGROUP_FORCE", "1")           # a handful of groups: without this the operator declines (few groups -> the usual route)
    from tests import kat
    from tests.test_gpu_golden import _agg, _series
    (kname, kspec), = case["keys"].items()
    (vname, vspec), = case["values"].items()
    k = pl.Series.from_device_views(kname, _views_series(pl, kat.expand(kspec)), encode="deferred")
    v = _series(pl, vname, vspec, case["value_dtypes"][vname])
    out = pl.DataFrame([k, v]).lazy().group_by(kname).agg(*[_agg(pl, c, o) for c, o in aggs]).collect()      # (no maintain_order: the operator's groups come in no particular order)
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    names = [kname] + [f"{c}_{o}" for c, o in aggs]
    assert out.columns == names
    rows = sorted(out.rows(), key=lambda r: r[0])
    exp = sorted((tuple(case["expect"][c][g] for c in names) for g in range(len(case["expect"][kname]))), key=lambda r: r[0])
    assert len(rows) == len(exp), (rows, exp)
    for got, want in zip(rows, exp):
        for g, e in zip(got, want):
            assert kat.same_value(g, e, 1e-12), (case["id"], rows, exp)


def _id_views(pl, ids):
    """Inline views of "id%010d" % id, built on the host (tools/strgroup_sweep.py has the same)."""
    digits = np.zeros((len(ids), 10), np.uint8)
    x = np.asarray(ids, np.int64).copy()
    for j in range(9, -1, -1):
        digits[:, j] = 48 + x % 10
        x //= 10
    raw = np.zeros((len(ids), 16), np.uint8)
    raw[:, 0] = 12
    raw[:, 4] = ord("i"); raw[:, 5] = ord("d")
    raw[:, 6:16] = digits
    return pl.Series("views", raw.reshape(-1).view(np.uint64), pl.UInt64)


@pytest.mark.parametrize("shape", ["one_string_half_of_the_rows", "zipf_1.1", "hot_with_null_values_int"])
def test_string_key_group_by_heavy_hitters(pl, shape):
    """Strings that hold a large share of the rows are summed in the scatter's LDS cells (sampled: sg_hot_kernel) and never reach a partition: without
    that, one string with half of the rows cost 76 ms instead of 1.6 at 2^26 rows.  Exact for Int64, 1e-9 for Float64; the plan reports hot > 0."""
    import re
    rng = np.random.default_rng(23)
    n, G = 6_000_011, 300_000
    if shape == "zipf_1.1":
        ids = (rng.zipf(1.1, n) - 1) % G
    else:
        ids = rng.integers(0, G, n)
        ids[rng.random(n) < 0.5] = 123_456
    k = pl.Series.from_device_views("k", _id_views(pl, ids), encode="deferred")
    if shape == "hot_with_null_values_int":
        x = rng.integers(-10 ** 9, 10 ** 9, n)
        valid = rng.random(n) > 0.25
        v = pl.Series("v", x, pl.Int64, validity=valid)
    else:
        x = rng.uniform(-1, 1, n)
        valid = np.ones(n, bool)
        v = pl.Series("v", x)
    out = pl.DataFrame([k, v]).lazy().group_by("k").agg(pl.col("v").sum().alias("s"), pl.col("v").count().alias("c"), pl.len().alias("n")).collect()
    plan = pl.last_plan()
    m = re.search(r"StringViewGroupBy\{.*hot=(\d+)", plan)
    assert m and int(m.group(1)) > 0, plan
    got = _by_key(out)
    present = np.unique(ids)
    assert got["k"] == ["id%010d" % i for i in present]
    assert got["n"] == np.bincount(ids, minlength=G)[present].tolist()
    assert got["c"] == np.bincount(ids, valid.astype(np.float64), minlength=G)[present].astype(np.int64).tolist()
    want = np.bincount(ids, np.where(valid, x, 0), minlength=G)[present]
    if shape == "hot_with_null_values_int":
        assert got["s"] == want.astype(np.int64).tolist()                      # |sums| < 2^53: bincount's float accumulator is exact
    else:
        assert np.allclose(got["s"], want, rtol=1e-9, atol=1e-9)


def test_string_key_group_by_leaves_few_groups_to_the_usual_route(pl):
    rng = np.random.default_rng(29)
    n = 3_000_000
    ids = rng.integers(0, 100, n)
    x = rng.uniform(0, 1, n)
    k = pl.Series.from_device_views("k", _id_views(pl, ids), encode="deferred")
    out = pl.DataFrame([k, pl.Series("v", x)]).lazy().group_by("k").agg(pl.col("v").sum().alias("s"), pl.len().alias("n")).collect()
    assert "StringViewGroupBy" not in pl.last_plan() and not k._is_raw_views()
    got = _by_key(out)
    assert got["k"] == ["id%010d" % i for i in range(100)] and got["n"] == np.bincount(ids).tolist() and np.allclose(got["s"], np.bincount(ids, x), rtol=1e-9)


def test_scan_ipc_string_key_group_by_on_views(pl, tmp_path):
    """scan_ipc(string_keys="deferred"): a Utf8 column comes out of the file as device-built views, group_by on it runs on the views; any other use of the
    column and strings over 12 bytes take the encoded route; a key column with nulls is handed out as stamped views -- same answers (against pandas)."""
    import pyarrow.feather  # noqa: F401
    import pyarrow.ipc as ipc
    rng = np.random.default_rng(31)
    n, G = 1_200_000, 60_000
    ids = rng.integers(0, G, n)
    keys = np.array(["id%010d" % i for i in range(G)])[ids]
    v = rng.uniform(-5, 5, n)
    w = rng.integers(-1000, 1000, n)
    table = pa.table({"k": pa.array(keys, pa.string()), "v": v, "w": w, "long": pa.array(np.char.add(keys, "_and_some_more"), pa.large_string()),
                      "kn": pa.array([None if i % 97 == 0 else s for i, s in enumerate(keys)], pa.string())})
    path = str(tmp_path / "keys.arrow")
    with ipc.new_file(path, table.schema) as wr:
        for b in table.to_batches(max_chunksize=250_000):              # several record batches
            wr.write_batch(b)
    want = table.select(["k", "v"]).to_pandas().groupby("k")["v"].agg(["sum", "mean", "size"]).sort_index()

    lf = pl.scan_ipc(path, string_keys="deferred")
    out = lf.group_by("k").agg(pl.col("v").sum().alias("s"), pl.col("v").mean().alias("m"), pl.len().alias("n")).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    got = _by_key(out)
    assert got["k"] == want.index.tolist() and got["n"] == want["size"].tolist()
    assert np.allclose(got["s"], want["sum"], rtol=1e-9, atol=1e-9) and np.allclose(got["m"], want["mean"], rtol=1e-9, atol=1e-9)
    # an Int64 value column of the same scan
    out = lf.group_by("k").agg(pl.col("w").sum()).collect()
    assert "StringViewGroupBy" in pl.last_plan()
    assert _by_key(out)["w"] == table.select(["k", "w"]).to_pandas().groupby("k")["w"].sum().sort_index().tolist()
    # the column used otherwise: encoded on first touch
    one = lf.filter(pl.col("k") == "id0000000007").select(pl.col("v").sum().alias("s"), pl.len().alias("n")).collect().to_dict()
    assert one["n"] == [int((ids == 7).sum())] and abs(one["s"][0] - v[ids == 7].sum()) < 1e-9
    # strings over 12 bytes: the operator declines, the views + bytes are encoded, same groups
    out = lf.group_by("long").agg(pl.col("v").sum().alias("s")).collect()
    assert "StringViewGroupBy" not in pl.last_plan()
    gl = _by_key(out, "long")
    assert gl["long"] == [k + "_and_some_more" for k in want.index] and np.allclose(gl["s"], want["sum"], rtol=1e-9, atol=1e-9)
    # a key column with nulls comes out as stamped views (round 4): the operator runs on them, null is a group of its own
    out = lf.group_by("kn").agg(pl.len().alias("n"), pl.col("v").sum().alias("s")).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    d = out.to_dict()
    wn = table.select(["kn", "v"]).to_pandas().groupby("kn", dropna=False)["v"].agg(["sum", "size"])
    assert len(d["kn"]) == len(wn) and sum(d["n"]) == n and d["kn"].count(None) == 1
    i_null = d["kn"].index(None)
    assert d["n"][i_null] == int(wn.loc[wn.index.isna(), "size"].iloc[0]) and abs(d["s"][i_null] - float(wn.loc[wn.index.isna(), "sum"].iloc[0])) < 1e-6
    present = {kk: (m, s) for kk, m, s in zip(d["kn"], d["n"], d["s"]) if kk is not None}
    assert all(present[kk][0] == int(r["size"]) and abs(present[kk][1] - float(r["sum"])) < 1e-6 for kk, r in wn[~wn.index.isna()].iterrows())
    # and the default scan is unchanged
    out = pl.scan_ipc(path).group_by("k").agg(pl.col("v").sum().alias("s")).collect()
    assert "StringViewGroupBy" not in pl.last_plan()
    assert np.allclose(_by_key(out)["s"], want["sum"], rtol=1e-9, atol=1e-9)


def _views_and_validity(pl, strings):
    """Host strings with None entries -> (UInt64 Series of 2 n view words as Arrow holds them: null slots carry whatever the builder left, Boolean validity Series)."""
    arr = pa.array(strings, pa.string_view())
    raw = np.frombuffer(arr.buffers()[1], dtype=np.uint64, count=2 * len(strings), offset=16 * arr.offset).copy()
    valid = np.array([s is not None for s in strings])
    raw[np.repeat(~valid, 2)] = np.uint64(0x0123456789abcdef)                     # garbage behind null slots must not matter
    return pl.Series("views", raw, pl.UInt64), pl.Series("valid", valid, pl.Boolean)


@pytest.mark.parametrize("null_share", [0.03, 0.6])
def test_null_string_keys_form_one_group_of_their_own(pl, monkeypatch, null_share):
    """Rows whose key is null are one group (key null), exactly as group_by on a nullable String column in the reference (hash_keys.rs:413-452: the validity is
    part of the key); with most rows null the null key is the heavy hitter the scatter sums on the spot.  The empty string is a different key."""
    monkeypatch.setenv("PLX_STRGROUP_FORCE", "1")
    rng = np.random.default_rng(17)
    words = ["", "a", "b", "ab", "abcdefghijkl", "twelve bytes", "0"] + ["w%05d" % i for i in range(5000)]
    n = 400_003
    idx = rng.integers(0, len(words), n)
    strings = [words[i] for i in idx]
    isnull = rng.random(n) < null_share
    strings = [None if z else s for s, z in zip(strings, isnull)]
    x = rng.integers(-10 ** 9, 10 ** 9, n)
    vvalid = rng.random(n) > 0.1
    views, valid = _views_and_validity(pl, strings)
    k = pl.Series.from_device_views("k", views, validity=valid, encode="deferred")
    v = pl.Series("v", x, pl.Int64, validity=vvalid)
    out = pl.DataFrame([k, v]).lazy().group_by("k").agg(pl.col("v").sum(), pl.col("v").count().alias("c"), pl.len().alias("n")).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    d = out.to_dict()
    assert d["k"].count(None) == 1 and out["k"].null_count() == 1
    got = {kk: (s, c, m) for kk, s, c, m in zip(d["k"], d["v"], d["c"], d["n"])}
    want = {}
    for sk, xv, ok in zip(strings, x.tolist(), vvalid.tolist()):
        e = want.setdefault(sk, [0, 0, 0])
        e[2] += 1
        if ok:
            e[0] += xv; e[1] += 1
    assert set(got) == set(want) and "" in got and None in got
    assert all(tuple(want[kk]) == got[kk] for kk in want)
    # the encode route reads the same stamps: null codes for the same rows, the same groups
    k2 = pl.Series.from_device_views("k", views, encode="eager")                  # (views are already stamped)
    assert k2.null_count() == int(isnull.sum())
    out2 = pl.DataFrame([k2, v]).lazy().group_by("k").agg(pl.col("v").sum(), pl.col("v").count().alias("c"), pl.len().alias("n")).collect()
    assert "StringViewGroupBy" not in pl.last_plan()
    d2 = out2.to_dict()
    assert {kk: (s, c, m) for kk, s, c, m in zip(d2["k"], d2["v"], d2["c"], d2["n"])} == got


def test_ipc_string_column_with_nulls_is_handed_out_as_stamped_views(pl, tmp_path, monkeypatch):
    """scan_ipc(string_keys="deferred") over a Utf8 column WITH nulls: the views come out stamped, the group-by runs on them and has the null group."""
    import pyarrow.ipc as ipc
    monkeypatch.setenv("PLX_STRGROUP_FORCE", "1")
    rng = np.random.default_rng(23)
    n = 200_000
    keys = np.array(["k%04d" % i for i in range(3000)])[rng.integers(0, 3000, n)].astype(object)
    keys[rng.random(n) < 0.07] = None
    v = rng.random(n)
    t = pa.table({"k": pa.array(keys.tolist(), pa.string()), "v": pa.array(v)})
    path = str(tmp_path / "nulls.arrow")
    with ipc.new_file(path, t.schema) as w:
        for b in t.to_batches(max_chunksize=1 << 16):
            w.write_batch(b)
    out = pl.scan_ipc(path, string_keys="deferred").group_by("k").agg(pl.col("v").sum().alias("s"), pl.len().alias("n")).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    d = out.to_dict()
    import pandas as pd
    ref = pd.DataFrame({"k": keys, "v": v}).groupby("k", dropna=False).agg(s=("v", "sum"), n=("v", "size"))
    want = {(None if (isinstance(kk, float) and np.isnan(kk)) or kk is None else kk): (float(r.s), int(r.n)) for kk, r in ref.iterrows()}
    assert set(d["k"]) == set(want) and None in want
    for kk, s, m in zip(d["k"], d["s"], d["n"]):
        assert m == want[kk][1] and abs(s - want[kk][0]) <= 1e-9 * max(1.0, abs(want[kk][0]))


def test_reference_vector_null_string_key_group(pl, monkeypatch):
    """py-polars/tests/unit/operations/test_group_by.py:948-1000 (test_perfect_hash_table_null_values): 101 string keys, three of them null, 40 groups in order of
    first appearance with the null group in 28th place holding [None, None, None].  Here: the same keys as stamped views through the string-key operator (forced:
    40 groups are far below its planning threshold) -- the same 40 groups, the same sizes, one null group of three rows."""
    monkeypatch.setenv("PLX_STRGROUP_FORCE", "1")
    # fmt: off
    values = ["3", "41", "17", "5", "26", "27", "43", "45", "41", "13", "45", "48", "17", "22", "31", "25", "28", "13", "7", "26", "17", "4", "43", "47", "30", "28", "8", "27", "6", "7", "26", "11", "37", "29", "49", "20", "29", "28", "23", "9", None, "38", "19", "7", "38", "3", "30", "37", "41", "5", "16", "26", "31", "6", "25", "11", "17", "31", "31", "20", "26", None, "39", "10", "38", "4", "39", "15", "13", "35", "38", "11", "39", "11", "48", "36", "18", "11", "34", "16", "28", "9", "37", "8", "17", "48", "44", "28", "25", "30", "37", "30", "18", "12", None, "27", "10", "3", "16", "27", "6"]
    groups = ["3", "41", "17", "5", "26", "27", "43", "45", "13", "48", "22", "31", "25", "28", "7", "4", "47", "30", "8", "6", "11", "37", "29", "49", "20", "23", "9", None, "38", "19", "16", "39", "10", "15", "35", "36", "18", "34", "44", "12"]
    # fmt: on
    sizes = {"3": 3, "41": 3, "17": 5, "5": 2, "26": 5, "27": 4, "43": 2, "45": 2, "13": 3, "48": 3, "22": 1, "31": 4, "25": 3, "28": 5, "7": 3, "4": 2, "47": 1, "30": 4, "8": 2, "6": 3,
             "11": 5, "37": 4, "29": 2, "49": 1, "20": 2, "23": 1, "9": 2, None: 3, "38": 4, "19": 1}          # the first thirty of the reference's agg_values, by length
    views, valid = _views_and_validity(pl, values)
    k = pl.Series.from_device_views("a", views, validity=valid, encode="deferred")
    ones = pl.Series("one", np.ones(len(values), dtype=np.int64), pl.Int64)
    out = pl.DataFrame([k, ones]).lazy().group_by("a").agg(pl.col("one").sum().alias("n"), pl.len().alias("len")).collect()
    assert "StringViewGroupBy" in pl.last_plan(), pl.last_plan()
    d = out.to_dict()
    got = dict(zip(d["a"], d["n"]))
    assert set(got) == set(groups) and len(d["a"]) == 40 and got[None] == 3 and d["n"] == d["len"]
    assert all(got[g] == c for g, c in sizes.items())
    assert all(got[g] == sum(1 for v in values if v == g) for g in groups)


def _long_key_views(pl, rng, n, n_keys, lo=13, hi=40, with_short=False):
    """Utf8View column of n rows over n_keys distinct strings of lo..hi bytes (with_short: a third of them <= 12 bytes, inline), built the way an Arrow producer lays it
    out: every row's bytes sit in ONE data buffer at the row's own offset, the view carries {length, 4-byte prefix, buffer 0, offset}."""
    lens = rng.integers(lo, hi + 1, n_keys)
    if with_short:
        lens[::3] = rng.integers(0, 13, len(lens[::3]))
    pool = [(b"k%06d-" % i + bytes(rng.integers(97, 123, 40).astype(np.uint8)))[:lens[i]] for i in range(n_keys)]
    ids = rng.integers(0, n_keys, n)
    row_len = lens[ids].astype(np.int64)
    off = np.concatenate([[0], np.cumsum(np.where(row_len > 12, row_len, 0))])
    data = np.zeros(int(off[-1]) + 16, np.uint8)
    views = np.zeros((n, 4), np.uint32)
    padded = np.zeros((n_keys, 40), np.uint8)
    for i, s in enumerate(pool):
        padded[i, :len(s)] = np.frombuffer(s, np.uint8)
    rows = padded[ids]                                   # [n, 40]
    views[:, 0] = row_len
    inline = row_len <= 12
    v8 = views.view(np.uint8).reshape(n, 16)
    v8[inline, 4:16] = rows[inline, :12]
    v8[~inline, 4:8] = rows[~inline, :4]                 # prefix
    views[~inline, 2] = 0                                # buffer index
    views[~inline, 3] = off[:-1][~inline].astype(np.uint32)
    long_rows = np.nonzero(~inline)[0]
    for L in np.unique(row_len[long_rows]):              # bytes of the long strings at their offsets, one length class at a time
        sel = long_rows[row_len[long_rows] == L]
        idx = off[:-1][sel][:, None] + np.arange(L)[None, :]
        data[idx] = rows[sel, :L]
    vs = pl.Series("views", views.view(np.uint64).reshape(-1).copy(), pl.UInt64)
    ds = pl.Series("data", data, pl.UInt8)
    return vs, ds, [p.decode() for p in pool], ids


@pytest.mark.parametrize("with_short", [False, True])
def test_string_key_group_by_long_keys_three_value_columns_min_max(pl, with_short):
    """Round-5 review, missing 4 / item 6: keys of 13-40 bytes (the view is NOT the string: get_long_key compares through the buffers, binview_index_map.rs:106-117), three value
    columns, min / max -- everything the string-key operator's one fast shape declines.  The deferred column then encodes on the device (plx_strview_dict_encode_device reads the
    bytes behind the views) and the group-by runs on the codes: same answers as pandas, whatever mix of inline and long strings the column holds."""
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(2024 + with_short)
    n, n_keys = 2_000_003, 30_000
    vs, ds, pool, ids = _long_key_views(pl, rng, n, n_keys, with_short=with_short)
    a = rng.normal(size=n)
    b = rng.integers(-10 ** 9, 10 ** 9, n).astype(np.int64)
    c = rng.integers(-1000, 1000, n).astype(np.int32)
    cv = rng.random(n) > 0.1
    k = pl.Series.from_device_views("k", vs, ds, encode="deferred")
    df = pl.DataFrame([k, pl.Series("a", a), pl.Series("b", b), pl.Series("c", c, validity=cv)])
    P = pl.col
    out = df.lazy().group_by("k").agg(P("a").sum().alias("a_sum"), P("b").min().alias("b_min"), P("b").max().alias("b_max"), P("c").mean().alias("c_mean"), P("c").count().alias("c_n"), pl.len().alias("n")).collect()
    got = _by_key(out)
    ref = pd.DataFrame({"k": np.array(pool, dtype=object)[ids], "a": a, "b": b, "c": np.where(cv, c.astype(np.float64), np.nan)})
    exp = ref.groupby("k").agg(a_sum=("a", "sum"), b_min=("b", "min"), b_max=("b", "max"), c_mean=("c", "mean"), c_n=("c", "count"), n=("a", "size")).reset_index().sort_values("k")
    assert got["k"] == exp["k"].tolist()
    assert got["b_min"] == exp["b_min"].tolist() and got["b_max"] == exp["b_max"].tolist() and got["c_n"] == exp["c_n"].tolist() and got["n"] == exp["n"].tolist()
    assert np.allclose(got["a_sum"], exp["a_sum"].to_numpy(), rtol=1e-6, atol=1e-9)
    cm = np.array([np.nan if x is None else x for x in got["c_mean"]], dtype=np.float64)
    assert np.allclose(cm, exp["c_mean"].to_numpy(), rtol=1e-6, atol=1e-12, equal_nan=True)


def test_generated_20_byte_keys_views_pool_and_group_by(pl):
    """plx_datagen_long_id_views (bench workload cfg5l): the views are {20, "id00".., buffer 0, (id - lo) * 20}, the pool holds "id%010d-longkey" of every id once;
    the deferred column of such keys groups through device encoding (the string-key operator's fast path declines: the view is not the string) and equals numpy."""
    from polars_amd import datagen
    n, seed, lo, hi = 3_000_017, 21, 1, 50_001
    views, data = datagen.long_id_views_native(pl, n, seed, 0, lo, hi)
    ids = datagen.uniform_native_host("Int64", 0, n, seed, 0, lo, hi)
    pool = data.to_numpy()
    assert pool.dtype == np.uint8 and len(pool) == (hi - lo) * 20
    want_pool = np.frombuffer(b"".join(b"id%010d-longkey" % i for i in range(lo, hi)), np.uint8)
    assert np.array_equal(pool, want_pool)
    w = views.to_numpy().reshape(n, 2)
    prefix = np.frombuffer(b"".join(b"id%010d" % i for i in range(lo, hi)), np.uint8).reshape(-1, 12)[:, :4].copy().view(np.uint32).reshape(-1).astype(np.uint64)
    assert np.array_equal(w[:, 0], np.uint64(20) | (prefix[ids - lo] << np.uint64(32)))
    assert np.array_equal(w[:, 1], ((ids - lo).astype(np.uint64) * np.uint64(20)) << np.uint64(32))
    v = datagen.uniform_native(pl, "v", pl.Float64, n, seed, 1, 0, 10 ** 9, 1e-7)
    k = pl.Series.from_device_views("k", views, data, encode="deferred")
    out = pl.DataFrame([k, v]).lazy().group_by("k").agg(pl.col("v").sum().alias("v_sum"), pl.col("v").mean().alias("v_mean"), pl.len()).collect()
    assert "StringViewGroupBy" not in pl.last_plan(), pl.last_plan()
    vals = datagen.uniform_native_host("Float64", 0, n, seed, 1, 0, 10 ** 9, 1e-7)
    s, c = np.bincount(ids, weights=vals, minlength=hi), np.bincount(ids, minlength=hi)
    present = np.nonzero(c)[0]
    got = _by_key(out)
    assert got["k"] == ["id%010d-longkey" % i for i in present]
    assert np.allclose(got["v_sum"], s[present], rtol=1e-9) and np.allclose(got["v_mean"], s[present] / c[present], rtol=1e-9) and got["len"] == c[present].tolist()
