"""The reference's expression-plugin ABI (_polars_plugin_*: plugin.rs:70-137, SeriesExport version_0.rs:7-16) on the GPU:
results against the CPU oracle, ownership and error conventions, and concurrent calls from two host threads with the
CallerContext parallel bit set (each thread then runs on its own HIP stream; the pool hands buffers between streams)."""
import threading

import numpy as np
import pyarrow as pa
import pytest

from tests import plugin_abi as P

pytestmark = pytest.mark.gpu


def _arr(values, valid=None, type=None):
    return pa.array(values, type=type, mask=None if valid is None else ~valid)


def test_cmp_arith_filter_reduce_through_the_plugin_abi(pl, orc):
    rng = np.random.default_rng(7)
    n = 100_003
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    b = rng.integers(-1000, 1000, n).astype(np.int64)
    av = rng.random(n) > 0.1
    # column vs column with nulls, chunked input
    res, inp = P.call("plx_cmp", [pa.chunked_array([_arr(a[:40_000], av[:40_000]), _arr(a[40_000:], av[40_000:])]), _arr(b)], {"op": "gt"}, names=["a", "b"])
    assert inp.released == 2 and inp.out_name == b"a"
    want = orc.cmp(orc.GT, a, b)
    got = res.to_numpy(zero_copy_only=False)
    assert res.null_count == int((~av).sum())
    assert np.array_equal(np.asarray(got[av], bool), want[av])
    # literal on either side (length-1 series = broadcast)
    res, _ = P.call("plx_lt", [_arr(a), _arr(np.array([17], np.int64))])
    assert np.array_equal(res.to_numpy(zero_copy_only=False), orc.cmp(orc.LT, a, 17))
    res, _ = P.call("plx_lt", [_arr(np.array([17], np.int64)), _arr(a)])
    assert np.array_equal(res.to_numpy(zero_copy_only=False), orc.cmp(orc.GT, a, 17))
    # arithmetic: wrapping ints, true division of ints -> f64 via x * (1 / lit), floor-div by zero -> null
    res, _ = P.call("plx_arith", [_arr(a), _arr(b)], {"op": "mul"})
    assert np.array_equal(res.to_numpy(), orc.arith(orc.MUL, a, b)[0])
    res, _ = P.call("plx_truediv", [_arr(a), _arr(np.array([7], np.int64))])
    assert res.type == pa.float64() and np.array_equal(res.to_numpy(), orc.arith(orc.TRUE_DIV, a, 7, mode=1)[0])
    z = b.copy(); z[::5] = 0
    res, _ = P.call("plx_floordiv", [_arr(a), _arr(z)])
    vals, extra = orc.arith(orc.FLOOR_DIV, a, z)
    assert res.null_count == int((~extra).sum()) and np.array_equal(res.to_numpy(zero_copy_only=False)[extra].astype(np.int64), vals[extra])
    x = rng.uniform(-5, 5, n)
    res, _ = P.call("plx_sub", [_arr(np.array([1.0])), _arr(x)])
    assert np.array_equal(res.to_numpy(), orc.arith(orc.SUB, 1.0, x, mode=2)[0])
    # filter (null mask rows drop) and reductions
    m = rng.random(n) > 0.5
    mv = rng.random(n) > 0.05
    res, _ = P.call("plx_filter", [_arr(x), _arr(m, mv)])
    assert np.array_equal(res.to_numpy(), x[m & mv])
    res, _ = P.call("plx_sum", [_arr(a, av)])
    assert res.to_pylist() == [orc.reduce(orc.AGG_SUM, a, av)[0]]
    res, _ = P.call("plx_mean", [_arr(x)])
    assert res.to_pylist()[0] == pytest.approx(orc.reduce(orc.AGG_MEAN, x)[0], rel=1e-12)
    res, _ = P.call("plx_sum", [_arr(a.astype(np.int16))])
    assert res.type == pa.int64() and res.to_pylist() == [int(a.sum())]


def test_plugin_errors_follow_the_reference_convention(pl):
    res, inp = P.call("plx_gt", [_arr(np.arange(4, dtype=np.int64)), _arr(np.arange(4, dtype=np.int32))])
    assert res is None and inp.released == 2 and "dtype" in P.last_error()
    res, inp = P.call("plx_cmp", [_arr(np.arange(4, dtype=np.int64)), _arr(np.arange(4, dtype=np.int64))], {"op": "spaceship"})
    assert res is None and inp.released == 2 and "kwargs" in P.last_error()
    res, inp = P.call("plx_sum", [pa.array(["a", "b"])])
    assert res is None and inp.released == 1 and "unsupported Arrow format" in P.last_error()
    ok, _ = P.call("plx_sum", [_arr(np.arange(4, dtype=np.int64))])      # the library keeps working after failures
    assert ok.to_pylist() == [6]


def test_two_threads_call_concurrently_on_their_own_streams(pl):
    """What rayon workers do to a plugin (CallerContext bit 0, version_0.rs:136-162): both threads hammer the library at the
    same time; every result must be that thread's own (no buffer handed to the other stream while still in use)."""
    n, rounds = 1_000_003, 25
    errs = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for r in range(rounds):
                a = rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int64)
                b = rng.integers(1, 1000, n).astype(np.int64)
                res, inp = P.call("plx_arith", [_arr(a), _arr(b)], {"op": "add"}, parallel=True)
                assert inp.released == 2
                got = res.to_numpy()
                if not np.array_equal(got, a + b):
                    errs.append((seed, r, "add"))
                res, _ = P.call("plx_filter", [_arr(a), _arr(b > 500)], parallel=True)
                if not np.array_equal(res.to_numpy(), a[b > 500]):
                    errs.append((seed, r, "filter"))
                res, _ = P.call("plx_sum", [_arr(a)], parallel=True)
                if res.to_pylist() != [int(a.sum())]:
                    errs.append((seed, r, "sum"))
        except Exception as e:   # noqa: BLE001
            errs.append((seed, repr(e)))
    ts = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:5]
