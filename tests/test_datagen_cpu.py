"""The library's counter-based lineitem generator (polars_amd/csrc/datagen_device.hpp) without a GPU: the host twin of the
kernel (plx_datagen_lineitem_q1_host) is pinned against an independent numpy restatement of the same arithmetic, its
distributions are the TPC-H-like ones of polars_amd/datagen.py, and the oracle's Q1 over it has the expected group structure."""
import numpy as np

from polars_amd import datagen

M64 = (1 << 64) - 1
DAY = 86_400_000_000


def _mix(z):
    z = (z + 0x9e3779b97f4a7c15) & M64
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
    return z ^ (z >> 31)


def _rand(key, i, s, lo, hi):
    return lo + ((_mix((key + i * 8 + s) & M64) * (hi - lo)) >> 64)


def _row(seed, i):
    key = _mix(seed)
    ship = datagen.START + _rand(key, i, 0, 1, 2647) * DAY
    qty = _rand(key, i, 1, 1, 51)
    price = (qty * _rand(key, i, 2, 90000, 210000)) / 100.0
    disc = _rand(key, i, 3, 0, 11) / 100.0
    tax = _rand(key, i, 4, 0, 9) / 100.0
    receipt = ship + _rand(key, i, 5, 1, 31) * DAY
    coin = _mix((key + i * 8 + 6) & M64) >> 63
    flag = (0 if coin else 2) if receipt <= datagen.CURRENT else 1
    return ship, flag, int(ship > datagen.CURRENT), qty, price, disc, tax


def test_host_generator_matches_python_restatement():
    for seed, row0 in ((10, 0), (11, 123_456_789), (2 ** 40 + 7, 599_999_000)):
        got = datagen.lineitem_native_host(row0, 300, seed)
        for j in range(300):
            exp = _row(seed, row0 + j)
            assert tuple(got[c][j].item() for c in datagen.LINEITEM_Q1_COLS) == exp, (seed, row0 + j)


def test_generator_is_a_pure_function_of_seed_and_row():
    a = datagen.lineitem_native_host(1000, 500, 3)
    b = datagen.lineitem_native_host(1200, 300, 3)
    for c in datagen.LINEITEM_Q1_COLS:
        assert np.array_equal(a[c][200:], b[c])
    assert not np.array_equal(a["l_quantity"], datagen.lineitem_native_host(1000, 500, 4)["l_quantity"])
    assert len(datagen.lineitem_native_host(0, 0, 1)["l_quantity"]) == 0


def test_distributions_and_q1_group_structure(orc):
    n = 400_000
    cols = datagen.lineitem_native_host(0, n, 10)
    assert cols["l_quantity"].min() == 1 and cols["l_quantity"].max() == 50
    assert cols["l_discount"].min() == 0.0 and cols["l_discount"].max() == 0.10 and cols["l_tax"].max() == 0.08
    assert np.all(np.round(cols["l_extendedprice"] * 100) / 100 == cols["l_extendedprice"])       # two decimals
    assert 900.0 <= cols["l_extendedprice"].min() and cols["l_extendedprice"].max() <= 50 * 2100.0
    days = (cols["l_shipdate"] - datagen.START) // DAY
    assert days.min() >= 1 and days.max() <= 2646 and np.all((cols["l_shipdate"] - datagen.START) % DAY == 0)
    cutoff = datagen.us(1998, 9, 2)
    sel = float((cols["l_shipdate"] <= cutoff).mean())
    assert 0.90 < sel < 0.999                                  # Q1 keeps most of the table (TPC-H: ~98 %)
    r = orc.q1(cols, cutoff)
    pairs = sorted(zip(r["l_returnflag"].tolist(), r["l_linestatus"].tolist()))
    assert pairs == [(0, 0), (1, 0), (1, 1), (2, 0)]           # (A,F) (N,F) (N,O) (R,F)
    assert abs(r["avg_qty"].mean() - 25.5) < 0.5 and abs(r["avg_disc"].mean() - 0.05) < 0.002
    # the host numpy generator of the tests has the same shape (same ranges, same groups)
    ref = orc.q1(datagen.lineitem_host(n, seed=10), cutoff)
    assert sorted(zip(ref["l_returnflag"].tolist(), ref["l_linestatus"].tolist())) == pairs
    assert np.allclose(np.sort(ref["count_order"]) / n, np.sort(r["count_order"]) / n, atol=0.01)


def test_q3_generator_host_twin(orc):
    n = 20_000
    orders, li, cnt = datagen.orders_lineitem_native_host(0, n, n, seed=5)
    i = np.arange(n)
    assert np.array_equal(orders["o_orderkey"], (i // 8) * 32 + (i % 8) + 1)            # sparse dbgen keys, ascending
    assert cnt.min() == 1 and cnt.max() == 7 and len(li["l_orderkey"]) == int(cnt.sum())
    assert np.array_equal(li["l_orderkey"], np.repeat(orders["o_orderkey"], cnt))       # dbgen row order
    odate = np.repeat(orders["o_orderdate"], cnt)
    lag = (li["l_shipdate"] - odate) // DAY
    assert lag.min() >= 1 and lag.max() <= 121 and np.all((li["l_shipdate"] - odate) % DAY == 0)
    d = (orders["o_orderdate"] - datagen.START) // DAY
    assert d.min() >= 0 and d.max() <= 2405 and orders["o_orderdate"].max() <= datagen.END_ORDERS
    assert orders["o_custkey"].min() >= 1 and orders["o_custkey"].max() <= max(2, n // 10)
    assert np.all(orders["o_shippriority"] == 0)
    # a sub-range is the same table (pure function of seed / order / line)
    o2, l2, c2 = datagen.orders_lineitem_native_host(5000, 300, n, seed=5)
    lo = int(cnt[:5000].sum())
    assert np.array_equal(o2["o_orderdate"], orders["o_orderdate"][5000:5300]) and np.array_equal(c2, cnt[5000:5300])
    for c in datagen.LINEITEM_Q3_COLS:
        assert np.array_equal(l2[c], li[c][lo:lo + len(l2[c])])
    # Q3 over it has the shape the benchmark relies on: ~10 % of the orders pass the build filter, a few per cent of them end up as groups
    r = orc.q3(li, orders, datagen.us(1995, 3, 15))
    assert 0.003 * n < len(r["l_orderkey"]) < 0.03 * n and np.all(r["revenue"] > 0)
    ref_o, ref_l = datagen.orders_lineitem_host(n, seed=5, ordered=True)
    ref = orc.q3(ref_l, ref_o, datagen.us(1995, 3, 15))
    assert 0.7 < len(r["l_orderkey"]) / len(ref["l_orderkey"]) < 1.4                    # same selectivities as the numpy generator


def test_uniform_generator_host_twin():
    a = datagen.uniform_native_host("Int64", 0, 100_000, 7, 0, 0, 2 ** 31)
    assert a.min() >= 0 and a.max() < 2 ** 31 and abs(a.mean() / 2 ** 30 - 1) < 0.02
    k = datagen.uniform_native_host("UInt32", 0, 100_000, 7, 1, 0, 1_000_000)
    assert k.dtype == np.uint32 and k.max() < 1_000_000 and len(np.unique(k)) > 90_000
    x = datagen.uniform_native_host("Float64", 0, 100_000, 7, 2, 0, 10 ** 9, 1e-7)
    assert 0.0 <= x.min() and x.max() < 100.0 and abs(x.mean() - 50.0) < 0.5
    assert np.array_equal(datagen.uniform_native_host("Int64", 500, 100, 7, 0, 0, 2 ** 31), a[500:600])
    assert not np.array_equal(datagen.uniform_native_host("Int64", 0, 1000, 7, 3, 0, 2 ** 31), a[:1000])       # streams are independent
    # restated: lo + ((mix(mix(seed) + i * 8 + stream) * span) >> 64)
    key = _mix(7)
    for i in (0, 1, 99_999):
        assert a[i] == (_mix((key + i * 8 + 0) & M64) * 2 ** 31) >> 64


def test_zipf_generator_host_twin():
    """plx_datagen_zipf_host: key = floor(1 / x^10) - 1 with x uniform in [n_keys^-0.1, 1) in 62-bit fixed point -- pinned against a Python big-integer
    restatement; the density falls like k^-1.1 (P(key >= k) = ((k + 1)^-0.1 - x0) / (1 - x0))."""
    nk, seed, strm = 1_000_000, 11, 0
    k = datagen.zipf_native_host_mt(0, 400_000, seed, strm, nk, threads=2)
    assert k.dtype == np.int64 and k.min() == 0 and k.max() < nk
    x0 = datagen.zipf_x0_q62(nk)
    one = 1 << 62
    key = _mix(seed)
    for i in (0, 1, 7, 399_999):
        x = x0 + ((_mix((key + i * 8 + strm) & M64) * (one - x0)) >> 64)
        x2 = (x * x) >> 62; x4 = (x2 * x2) >> 62; x8 = (x4 * x4) >> 62; x10 = (x8 * x2) >> 62
        assert k[i] == min(one // x10 - 1, nk - 1), i
    assert np.array_equal(datagen.zipf_native_host_mt(1234, 999, seed, strm, nk, threads=1), k[1234:2233])          # a pure function of (seed, row)
    x0f = nk ** -0.1
    for kk in (1, 2, 10, 1000):
        want = ((kk) ** -0.1 - x0f) / (1 - x0f)                      # P(key >= kk) = P(K >= kk + 1 in units of floor(1 / x^10)) = P(x <= (kk + 1 - 1 ...))
        got = float((k >= kk - 0).mean()) if kk == 0 else float((k + 1 >= kk).mean())
        assert abs(got - want) < 0.01, (kk, got, want)
    assert 0.08 < float((k == 0).mean()) < 0.10 and 0.04 < float((k == 1).mean()) < 0.06


def test_customer_host_twin_properties():
    """plx_datagen_customer_host: dense keys in order, five market segments roughly uniform, a pure function of (seed, key)."""
    a = datagen.customer_native_host(0, 50_000, seed=9)
    assert np.array_equal(a["c_custkey"], np.arange(1, 50_001)) and a["c_mktsegment"].dtype == np.uint8 and a["c_mktsegment"].max() == 4
    frac = np.bincount(a["c_mktsegment"], minlength=5) / 50_000
    assert np.all(np.abs(frac - 0.2) < 0.01)
    b = datagen.customer_native_host(12_345, 1000, seed=9)
    assert np.array_equal(b["c_custkey"], a["c_custkey"][12_345:13_345]) and np.array_equal(b["c_mktsegment"], a["c_mktsegment"][12_345:13_345])
    assert not np.array_equal(datagen.customer_native_host(0, 1000, seed=10)["c_mktsegment"], a["c_mktsegment"][:1000])
