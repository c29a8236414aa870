"""Host logic of the deferred Utf8View column and of the string-key group-by's plan matcher (frame._string_key_group_by), without a GPU: schema-only
placeholder columns stand in for device columns, and the one call that would reach a kernel (plx_strview_groupby) is replaced by a recorder."""
import ctypes as C

import pytest

import polars_amd as pl
from polars_amd import _ffi as F
from polars_amd import frame as FR


def ph(name, dtype, n=1000):
    h = C.c_uint64()
    F.check(F.lib().plx_column_placeholder(dtype.physical, n, 0, 0, 0, 0, C.byref(h)))
    return pl.Series._from_handle(name, h.value, dtype)


def raw_key(n=1000, name="k"):
    return pl.Series.from_device_views(name, ph("views", pl.UInt64, 2 * n), encode="deferred")


def test_deferred_views_column_is_a_column_without_touching_the_library():
    k = raw_key()
    assert k._is_raw_views() and len(k) == 1000 and k.name == "k"
    r = k.rename("z")
    assert r._is_raw_views() and r.name == "z" and len(r) == 1000
    df = pl.DataFrame([k, ph("v", pl.Float64)])
    assert df.height == 1000 and df.columns == ["k", "v"] and k._is_raw_views()          # neither the length nor the names need the dictionary
    with pytest.raises(ValueError):
        pl.Series.from_device_views("k", ph("views", pl.UInt64, 7), encode="deferred")     # an odd number of view words
    with pytest.raises(ValueError):
        pl.Series.from_device_views("k", ph("views", pl.UInt64, 8), encode="sometime")
    with pytest.raises(ValueError):
        pl.DataFrame([raw_key(10), ph("v", pl.Float64, 11)])                               # lengths are checked on the views, too


class Recorder:
    """Stands in for plx_strview_groupby: notes the call, then answers PLX_ERR_UNSUPPORTED (the caller must fall back to its usual route)."""

    def __init__(self):
        self.calls = []

    def __call__(self, views, value, *outs):
        self.calls.append((views, value))
        return F.ERR_UNSUPPORTED


@pytest.fixture
def recorder(monkeypatch):
    rec = Recorder()

    class Lib:
        def __getattr__(self, name):
            return rec if name == "plx_strview_groupby" else getattr(F._lib, name)
    real = F.lib
    monkeypatch.setattr(F, "lib", lambda: Lib() if F._lib is not None else real())
    real()                                                                                  # make sure the library is loaded before the proxy is used
    return rec


def plan(df, keys, aggs, **kw):
    return df.lazy().group_by(*keys, **kw).agg(*aggs)._node


def test_plan_matcher_takes_exactly_the_plans_the_operator_serves(recorder):
    k, v, w = raw_key(), ph("v", pl.Float64), ph("w", pl.Int64)
    df = pl.DataFrame([k, v, w])
    col = pl.col
    # served: one raw string key, sum / mean / count / len of ONE Float64 or Int64 column (the recorder declines, so the answer is None -- but it was asked)
    for aggs in ([col("v").sum()], [col("v").sum().alias("s"), col("v").mean().alias("m"), col("v").count().alias("c"), pl.len()], [col("w").sum(), pl.len().alias("n")]):
        n0 = len(recorder.calls)
        assert FR._string_key_group_by(plan(df, ["k"], aggs)) is None
        assert len(recorder.calls) == n0 + 1, aggs
        assert recorder.calls[-1][0] == k._raw[0]._h
    assert k._is_raw_views()                                                                # asking never encoded the column
    # not served, and the library is not even asked
    n0 = len(recorder.calls)
    not_served = [
        plan(df, ["k"], [col("v").sum()], maintain_order=True),                             # groups in first-appearance order
        plan(df, ["k"], [col("v").min()]),                                                  # an aggregate the operator does not know
        plan(df, ["k"], [col("v").sum(), col("w").sum()]),                                  # two value columns
        plan(df, ["k"], [pl.len()]),                                                        # no value column at all
        plan(df, ["k"], [(col("v") * 2).sum()]),                                            # an expression, not a column
        plan(df, ["k"], [col("v").sum().alias("k")]),                                       # output name collides with the key
        plan(df, ["k", "w"], [col("v").sum()]),                                             # two keys
        plan(df, ["w"], [col("v").sum()]),                                                  # the key is not a views column
        df.lazy().filter(col("v") > 0).group_by("k").agg(col("v").sum())._node,             # not straight over a DataFrame
        df.lazy().select(col("k"))._node,                                                   # not a group-by
    ]
    for node in not_served:
        assert FR._string_key_group_by(node) is None
    assert len(recorder.calls) == n0
    v32 = pl.DataFrame([raw_key(), ph("v", pl.Float32)])
    assert FR._string_key_group_by(plan(v32, ["k"], [col("v").sum()])) is None and len(recorder.calls) == n0     # Float32 sums stay Float32 in the reference: not this operator
