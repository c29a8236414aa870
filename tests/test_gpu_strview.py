"""Raw Utf8View / BinaryView keys (reference: crates/polars-expr/src/hash_keys.rs:413-452 BinviewKeys,
crates/polars-compute/src/binview_index_map.rs): the library encodes the 16-byte views into dictionary codes ON THE DEVICE.
Checked against the oracle's restatement of the view index map (codes up to renaming: first-claim order on the device is not
first-appearance order), on inline strings, long strings (prefix + data buffer), shared prefixes / lengths, empty strings, nulls,
chunk offsets; the reference's string-key group_by vectors run through it; group-bys on the encoded column use dense tables."""
import re

import numpy as np
import pyarrow as pa
import pytest

from tests import kat
from tests.test_gpu_golden import _agg, _series

pytestmark = pytest.mark.gpu


def check_encoding(pl, orc, strings, arr=None):
    arr = arr if arr is not None else pa.array(strings, pa.string_view())
    s = pl.Series.from_arrow("k", arr)
    assert isinstance(s.dtype, pl.Categorical) and len(s) == len(strings)
    codes, valid, cats = orc.binview_dict_encode(strings)
    got = s.to_list()                                      # codes mapped through the device-built dictionary
    assert got == list(strings)
    assert sorted(s.dtype.categories) == sorted(cats) and len(set(s.dtype.categories)) == len(cats)       # one code per distinct string
    raw = s.to_numpy()
    v = np.ones(len(strings), bool) if valid is None else valid
    # same partition of the rows as the oracle's index map (a bijection between the two code spaces)
    pairs = set(zip(raw[v].tolist(), codes[v].tolist()))
    assert len(pairs) == len(cats) == len({a for a, _ in pairs}) == len({b for _, b in pairs})
    assert s.null_count() == int((~v).sum())
    return s


def test_inline_long_and_tricky_strings(pl, orc):
    base = ["", "a", "ab", "abc", "abcd", "abcde", "twelve bytes", "thirteen byte", "same prefix and length A", "same prefix and length B",
            "same prefix and length A", "x" * 12, "x" * 13, "x" * 14, "x" * 100, None, "a\0b", "a", "", None, "Ünïcödé strîng", "id0000000042"]
    check_encoding(pl, orc, base)
    rng = np.random.default_rng(81)
    words = ["id%010d" % i for i in rng.integers(0, 3000, 200_000)]                                  # config 5's keys: always inline
    check_encoding(pl, orc, words)
    long_words = ["customer#%09d/segment=%s" % (i, "AB"[i % 2] * (i % 7)) for i in rng.integers(0, 5000, 100_000)]     # 21..27 bytes: data buffer
    mixed = [None if rng.random() < 0.05 else w for w in long_words[:50_000]] + words[:50_000]
    check_encoding(pl, orc, mixed)
    sliced = pa.array(mixed, pa.string_view()).slice(1234, 60_001)                                     # Arrow offset honoured (views and validity)
    check_encoding(pl, orc, mixed[1234:1234 + 60_001], sliced)
    check_encoding(pl, orc, [])
    check_encoding(pl, orc, [None, None])


@pytest.mark.parametrize("case", [c for c in kat.load_cases("groupby") if "str" in c["key_dtypes"].values()], ids=lambda c: c["id"])
def test_reference_string_key_groupby_vectors_through_device_encoding(pl, case):
    cols = []
    for n, spec in case["keys"].items():
        if case["key_dtypes"][n] == "str":
            cols.append(pl.Series.from_arrow(n, pa.array(kat.expand(spec), pa.string_view())))
        else:
            cols.append(_series(pl, n, spec, case["key_dtypes"][n]))
    cols += [_series(pl, n, s, case["value_dtypes"][n]) for n, s in case["values"].items()]
    out = pl.DataFrame(cols).lazy().group_by(*case["keys"].keys(), maintain_order=case["maintain_order"]).agg(*[_agg(pl, c, o) for c, o in case["aggs"]]).collect()
    names = list(case["expect"].keys())
    rows = out.rows()
    exp_rows = [tuple(case["expect"][c][g] for c in names) for g in range(len(case["expect"][names[0]]))]
    if not case["maintain_order"]:
        nk = len(case["keys"])
        keyf = lambda r: tuple((x is None, x) for x in r[:nk])
        rows = sorted(rows, key=keyf); exp_rows = sorted(exp_rows, key=keyf)
    assert len(rows) == len(exp_rows), (rows, exp_rows)
    for got, exp in zip(rows, exp_rows):
        for g, e in zip(got, exp):
            assert kat.same_value(g, e, 1e-12), (case["id"], rows, exp_rows)


def test_config5_from_raw_strings(pl):
    """BASELINE config 5 starting from Utf8View keys generated in HBM (inline 12-byte "id%010d" strings): device-side dictionary
    encoding, then the dense-id partitioned group-by; against numpy on the generator's host twin."""
    from polars_amd import datagen, queries
    n, seed, n_keys = 17_000_000, 9, 300_000
    views = datagen.id_views_native(pl, "k", n, seed, 0, 1, n_keys + 1)
    k = pl.Series.from_device_views("k", views)
    assert len(k) == n and len(k.dtype.categories) <= n_keys
    v = datagen.uniform_native(pl, "v", pl.Float64, n, seed, 1, 0, 10 ** 9, 1e-7)
    out = queries.cfg5(pl.DataFrame([k, v]).lazy()).collect()
    assert re.search(r"partitioned\(v[23],direct", pl.last_plan()), pl.last_plan()
    ids = datagen.uniform_native_host("Int64", 0, n, seed, 0, 1, n_keys + 1)
    vals = datagen.uniform_native_host("Float64", 0, n, seed, 1, 0, 10 ** 9, 1e-7)
    s, c = np.bincount(ids, weights=vals, minlength=n_keys + 1), np.bincount(ids, minlength=n_keys + 1)
    present = np.nonzero(c)[0]
    got = out.to_dict()
    order = np.argsort(np.array(got["k"]))
    assert [got["k"][i] for i in order] == ["id%010d" % i for i in present]
    assert np.allclose(np.array(got["v_sum"])[order], s[present], rtol=1e-9) and np.allclose(np.array(got["v_mean"])[order], s[present] / c[present], rtol=1e-9)
