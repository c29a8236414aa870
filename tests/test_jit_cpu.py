"""Run-time kernel specialisation (polars_amd/csrc/jit.cpp): hiprtc must be able to compile, for gfx950 and without
a GPU, the specialised kernel of every sink for query shapes that have no pre-instantiated (AOT) kernel."""
import pytest

import polars_amd as pl
from tests.test_plan_compile_cpu import ph


def frames():
    t = pl.DataFrame([ph("a", pl.Int32), ph("k", pl.Int64), ph("k2", pl.Int64), ph("x", pl.Float64, nullable=True), ph("b", pl.Boolean)])
    small = pl.DataFrame([ph("k", pl.Int64, n=1 << 10), ph("pay", pl.Int16, n=1 << 10), ph("w", pl.Float64, n=1 << 10)])
    return t, small


def test_select_regagg():
    t, _ = frames()
    q = t.lazy().filter((pl.col("a") > 5) & (pl.col("x") <= 2.5) | pl.col("b")).select((pl.col("x") * 2 + 1).sum(), pl.col("a").max(), pl.col("x").min(), pl.len(),
                                                                                         (pl.col("k") // 7).sum(), (pl.col("k") % pl.col("k2")).count())
    assert q.describe_fusion()[1] == -1      # no AOT shape: this is what the JIT is for
    q.jit_selftest()


def test_groupby_table_sinks():
    t, _ = frames()
    q = t.lazy().filter(pl.col("a") != 0).group_by("k").agg(pl.col("x").sum(), pl.col("x").mean().alias("m"), pl.col("a").min().alias("mn"), pl.len())
    assert q.describe_fusion()[1] == -1
    q.jit_selftest()                           # LDS table, dense HBM table and hash HBM table sinks


def test_groupby_wide_key():
    t, _ = frames()
    q = t.lazy().group_by("k", "k2", "x").agg(pl.col("a").sum(), pl.len())
    fusable, sid, why, _ = q.describe_fusion()
    assert fusable and sid == -1, why
    q.jit_selftest()


def test_join_groupby_pipeline():
    t, small = frames()
    q = (t.lazy().filter(pl.col("a") > 0).join(small.lazy().filter(pl.col("pay") < 3), on="k").group_by("k", "pay")
         .agg((pl.col("x") * 0.5).sum().alias("s"), pl.len()))
    fusable, sid, why, dump = q.describe_fusion()
    assert fusable and sid == -1 and dump.count("\n") == 3, (why, dump)     # count / build / probe programs + the scatter program of the partitioned probe
    q.jit_selftest()


def test_selftest_reports_unfusable():
    t, _ = frames()
    with pytest.raises(pl.UnsupportedError):
        t.lazy().select(pl.col("a").sum(), pl.col("k")).jit_selftest()


def test_filter_to_frame_kernel():
    # Filter -> frame in one pass (fused_sinks.hpp fused_filter_body): predicate program + ordered compaction, specialised at run time
    t, _ = frames()
    q = t.lazy().filter((pl.col("a") > 5) & ((pl.col("k") % 3) == 1) | pl.col("x").is_null())
    q.jit_selftest()


def test_groupby_pair_packed_records():
    # one f64 value without nulls over integer keys: the partitioned path may send two rows a record (fused::kPackPair) -- its scatter (1-4 tiles, with and without the
    # hot-key path) and aggregation kernels compile at run time
    t = pl.DataFrame([ph("k", pl.Int64), ph("y", pl.Float64), ph("a", pl.Int32)])
    q = t.lazy().filter(pl.col("a") > 3).group_by("k").agg(pl.col("y").sum().alias("s"), pl.col("y").mean().alias("m"), pl.len())
    fusable, sid, why, _ = q.describe_fusion()
    assert fusable and sid == -1, why
    q.jit_selftest()
