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
