"""The library's lineitem generator kernel against its host twin, and TPC-H Q1 over a generated table against the oracle.
bench.py performs the same spot check before every headline measurement (and falls back to the torch generators if it
fails), so a defect here cannot corrupt a measurement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 4097, 1_000_003])
def test_device_generator_equals_host_twin(pl, n):
    from polars_amd import datagen
    df = datagen.lineitem_native(pl, n, seed=12)
    assert df.height == n and df.columns == datagen.LINEITEM_Q1_COLS
    want = datagen.lineitem_native_host(0, n, 12)
    for c in datagen.LINEITEM_Q1_COLS:
        assert np.array_equal(df[c].to_numpy(), want[c]), c


def test_q1_over_generated_table_matches_oracle(pl, orc):
    from polars_amd import datagen, queries
    n = 500_000
    df = datagen.lineitem_native(pl, n, seed=3)
    g = queries.q1(df.lazy()).collect().sort_host(["l_returnflag", "l_linestatus"])
    assert "fused_scan[aot]" in pl.last_plan(), pl.last_plan()
    want = orc.q1(datagen.lineitem_native_host(0, n, 3), datagen.us(1998, 9, 2))
    assert [datagen.FLAGS.index(x) for x in g["l_returnflag"]] == want["l_returnflag"].tolist()
    assert g["count_order"] == want["count_order"].tolist() and g["sum_qty"] == want["sum_qty"].tolist()
    for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
        assert np.allclose(np.array(g[c]), want[c], rtol=1e-6, atol=0), c


@pytest.mark.parametrize("n_orders", [0, 1, 5000, 300_001])
def test_q3_device_generator_equals_host_twin(pl, n_orders):
    from polars_amd import datagen
    O, L = datagen.orders_lineitem_native(pl, n_orders, seed=8)
    wo, wl, cnt = datagen.orders_lineitem_native_host(0, n_orders, n_orders, 8)
    assert O.height == n_orders and L.height == int(cnt.sum())
    for c in datagen.ORDERS_Q3_COLS:
        assert np.array_equal(O[c].to_numpy(), wo[c]), c
    for c in datagen.LINEITEM_Q3_COLS:
        assert np.array_equal(L[c].to_numpy(), wl[c]), c


def test_uniform_device_generator_equals_host_twin(pl):
    from polars_amd import datagen
    n = 1_000_003
    for name, dt, args in (("Int64", pl.Int64, (3, 0, 0, 2 ** 31, 1.0)), ("UInt32", pl.UInt32, (3, 1, 0, 1_000_000, 1.0)), ("Float64", pl.Float64, (3, 2, 0, 10 ** 9, 1e-7))):
        s = datagen.uniform_native(pl, "c", dt, n, *args)
        assert np.array_equal(s.to_numpy(), datagen.uniform_native_host(name, 0, n, *args)), name


def test_zipf_device_generator_equals_host_twin(pl):
    from polars_amd import datagen
    n = 1_000_003
    s = datagen.zipf_native(pl, "key", n, 17, 0, 1_000_000)
    assert np.array_equal(s.to_numpy(), datagen.zipf_native_host_mt(0, n, 17, 0, 1_000_000))


def test_dropped_statistics_change_the_plan_not_the_result(pl, monkeypatch):
    """plx_column_drop_statistics (bench.py one_shot_ms): a group-by on a raw Int64 key learns the key range on its first run and plans dense ids on the
    second; after the drop it plans like the first run again -- and gives the same groups every time.  (PLX_LEARN_DENSE_RANGE=0: without the range pass that
    the planner now runs up front for keys whose sample looks dense -- the next test.)"""
    import re
    monkeypatch.setenv("PLX_LEARN_DENSE_RANGE", "0")
    from polars_amd import queries
    rng = np.random.default_rng(5)
    n = 17_000_000
    key = rng.integers(1000, 301_000, n).astype(np.int64)
    v = rng.integers(0, 1000, n).astype(np.int64)
    df = pl.DataFrame({"key": key, "v": v})
    plans, outs = [], []
    for step in range(3):
        if step == 2:
            for c in df.get_columns():
                pl._ffi.check(pl._ffi.lib().plx_column_drop_statistics(c._h))
        outs.append(queries.cfg3(df.lazy()).collect().sort_host("key")); plans.append(pl.last_plan())
    assert "hash" in plans[0] and "key_range_learned" in plans[0], plans[0]
    assert re.search(r"partitioned\(v3,direct", plans[1]), plans[1]
    assert "hash" in plans[2] and "direct" not in plans[2], plans[2]
    assert outs[0] == outs[1] == outs[2]


def test_dense_looking_keys_get_their_range_before_the_first_run(pl):
    """A single Int64 key nobody has statistics for, >= 2^24 rows: a 65536-row sample says whether the keys LOOK dense; if so the exact range pass (8 B / row)
    runs before planning and the FIRST run already takes direct-address partitions; sparse 64-bit keys get no pass and take hash partitions as before."""
    import re
    from polars_amd import queries
    rng = np.random.default_rng(6)
    n = 17_000_000
    ids = rng.integers(0, 300_000, n)
    v = rng.integers(0, 1000, n).astype(np.int64)
    want_n = np.bincount(ids, minlength=300_000)
    want_s = np.bincount(ids, v, minlength=300_000).astype(np.int64)
    for name, key, unmap in (("dense", (ids + 1000).astype(np.int64), lambda k: k - 1000), ("sparse", ids.astype(np.int64) * 1_000_003 - 10 ** 12, lambda k: (k + 10 ** 12) // 1_000_003)):
        df = pl.DataFrame({"key": key, "v": v})
        out = queries.cfg3(df.lazy()).collect().sort_host("key")
        plan = pl.last_plan()
        if name == "dense":
            assert "KeyRange{key: sample looks dense" in plan and re.search(r"partitioned\(v3,direct", plan), plan
        else:
            assert "KeyRange{" not in plan and re.search(r"partitioned\(v3,hash", plan), plan
        k = unmap(np.array(out["key"], dtype=np.int64))                       # (sort_host: a dict of lists)
        agg_cols = [c for c in out if c != "key"]
        present = np.nonzero(want_n)[0]
        assert np.array_equal(k, present), name
        got = {c: np.array(out[c], dtype=np.int64) for c in agg_cols}
        assert any(np.array_equal(g, want_s[present]) for g in got.values()) and any(np.array_equal(g, want_n[present]) for g in got.values()), (name, agg_cols)


def test_customer_generator_matches_host_twin(pl):
    from polars_amd import datagen
    n = 300_001
    df = datagen.customer_native(pl, n, seed=5)
    want = datagen.customer_native_host(0, n, seed=5)
    assert np.array_equal(df["c_custkey"].to_numpy(), want["c_custkey"]) and np.array_equal(df["c_mktsegment"].to_numpy(), want["c_mktsegment"])


def _outside_the_sample(n, runs=1024, run=1024):
    """row indices no run of the planner's strided range sample (k::sample_minmax: `runs` evenly spaced runs of `run` rows) looks at"""
    stride = (n - run) // (runs - 1)
    return stride // 2 + stride * np.arange(3, 9)            # the middle of six gaps between runs


def test_a_wrong_guess_about_a_small_key_range_is_caught_by_the_lds_table(pl):
    """Keys whose sampled range fits an LDS table (<= 12 bits): ids below 2^bits decode to the right key whatever the guess was; a key beyond them is reported by the table
    sink (LdsAggSink `oob`) and the query runs again from exact statistics.  Nullable keys and multi-column keys are never guessed about (exact passes)."""
    from polars_amd import queries
    rng = np.random.default_rng(13)
    n = 17_000_000
    ids = rng.integers(0, 3000, n).astype(np.int64)
    v = rng.integers(0, 1000, n).astype(np.int64)
    ids[_outside_the_sample(n)] = 5000 + np.arange(6)
    df = pl.DataFrame({"key": ids, "v": v})
    out = queries.cfg3(df.lazy()).collect().sort_host("key")
    plan = pl.last_plan()
    assert "AssumedBoundsViolated{" in plan, plan
    keys, inv = np.unique(ids, return_inverse=True)
    assert np.array_equal(np.array(out["key"], dtype=np.int64), keys)
    sums = np.zeros(len(keys), np.int64); np.add.at(sums, inv, v)
    assert any(np.array_equal(np.array(out[c], dtype=np.int64), sums) for c in out if c != "key")
    # a nullable key: no guess (the null code sits right above the assumed maximum)
    valid = rng.random(n) > 0.01
    dfn = pl.DataFrame([pl.Series("key", ids, validity=valid), pl.Series("v", v)])
    outn = queries.cfg3(dfn.lazy()).collect()
    plan = pl.last_plan()
    assert "KeyRange{key: sample looks dense -> range pass}" in plan and "AssumedBoundsViolated{" not in plan, plan
    assert outn.height == len(np.unique(ids[valid])) + 1


@pytest.mark.parametrize("violate", ["nothing", "key", "value"])
def test_bounds_assumed_from_a_sample_are_checked_per_row_and_a_wrong_guess_runs_again(pl, violate):
    """First group-by over columns nobody has statistics for: the planner GUESSES their bounds from a strided sample (engine.cpp assume_range) instead of two exact
    min / max passes, plans direct-address tables and narrowed values with them, and every row is checked against them in the scatter.  A key or a value outside the guess
    (placed where the sample does not look) must not produce a wrong answer: the query is planned again from exact statistics, and these columns are never guessed about again."""
    import re
    from polars_amd import queries
    rng = np.random.default_rng(12)
    n = 17_000_000
    ids = rng.integers(0, 300_000, n).astype(np.int64)
    v = rng.integers(0, 1000, n).astype(np.int64)
    hole = _outside_the_sample(n)
    if violate == "key":
        ids[hole] = 5_000_000 + np.arange(len(hole))          # needs more key bits than the sampled span: beyond any free slack
    if violate == "value":
        v[hole] = (1 << 40) + np.arange(len(hole))            # does not fit a 32-bit offset from the sampled minimum
    df = pl.DataFrame({"key": ids + 1000, "v": v})
    out = queries.cfg3(df.lazy()).collect().sort_host("key")
    plan = pl.last_plan()
    assert ("bounds assumed from the sample" in plan) == (violate == "nothing"), plan          # (the plan string describes the run that produced the result)
    assert ("AssumedBoundsViolated{" in plan) == (violate != "nothing"), plan
    assert re.search(r"partitioned\(v3,direct" if violate != "key" else r"partitioned\(v3,hash", plan), plan      # (the stray keys make the exact range a 23-bit one: hash partitions)
    keys, inv = np.unique(ids + 1000, return_inverse=True)
    assert np.array_equal(np.array(out["key"], dtype=np.int64), keys)
    cols = [c for c in out if c != "key"]
    sums = np.zeros(len(keys), np.int64); np.add.at(sums, inv, v)
    cnts = np.bincount(inv)
    got = [np.array(out[c], dtype=np.int64) for c in cols]
    assert any(np.array_equal(g, sums) for g in got) and any(np.array_equal(g, cnts) for g in got)
    # the next run: the same plan without a guess when it was right (the bounds stay, checked per row), exact statistics when it was wrong
    out2 = queries.cfg3(df.lazy()).collect().sort_host("key")
    plan2 = pl.last_plan()
    assert "AssumedBoundsViolated{" not in plan2 and "bounds assumed" not in plan2, plan2
    assert out2 == out


def test_guessed_bounds_never_pack_several_keys_or_sit_under_a_null_code(pl):
    """Round-5 advisor finding: query 1 (single key under a FILTER) leaves guessed, unverified bounds on its key column -- the filter hid the outliers from the per-row
    check.  Query 2 groups by (k1, k2): packed with the guess, an outlier of k1 would spill into k2's bits and merge groups silently.  lower_keys drops a guess that
    nobody verified whenever the key is nullable or packed next to other columns (exact statistics instead)."""
    rng = np.random.default_rng(31)
    n = 17_000_000
    k1 = rng.integers(0, 200_000, n).astype(np.int64)
    k2 = rng.integers(0, 50, n).astype(np.int64)
    v = rng.integers(0, 1000, n).astype(np.int64)
    flag = np.ones(n, np.int64)
    hole = _outside_the_sample(n)
    k1[hole] = 3_000_000 + np.arange(len(hole))       # outliers where the sample does not look ...
    flag[hole] = 0                                    # ... and that query 1's predicate filters out
    df = pl.DataFrame({"k1": k1, "k2": k2, "v": v, "flag": flag})
    c = pl.col
    out1 = df.lazy().filter(c("flag") == 1).group_by("k1").agg(c("v").sum().alias("s")).collect()
    plan1 = pl.last_plan()
    assert "bounds assumed from the sample" in plan1 and "AssumedBoundsViolated{" not in plan1, plan1
    assert out1.height == len(np.unique(k1[flag == 1]))
    out2 = df.lazy().group_by("k1", "k2").agg(c("v").sum().alias("s"), pl.len().alias("n")).collect()
    keys, inv = np.unique(k1 * 64 + k2, return_inverse=True)
    got_k = np.array(out2["k1"].to_numpy(), np.int64) * 64 + np.array(out2["k2"].to_numpy(), np.int64)
    order = np.argsort(got_k)
    assert np.array_equal(got_k[order], keys), pl.last_plan()
    sums = np.zeros(len(keys), np.int64); np.add.at(sums, inv, v)
    assert np.array_equal(out2["s"].to_numpy()[order], sums) and np.array_equal(out2["n"].to_numpy()[order].astype(np.int64), np.bincount(inv))
    # the same guess under a NULL code: a nullable view of the key column
    valid = rng.random(n) > 0.01
    dfn = pl.DataFrame([pl.Series("k1", k1, validity=valid), pl.Series("v", v)])
    outn = dfn.lazy().group_by("k1").agg(c("v").sum().alias("s")).collect()
    assert outn.height == len(np.unique(k1[valid])) + 1


@pytest.mark.parametrize("dtype", ["Int64", "Int32", "UInt32"])
@pytest.mark.parametrize("violate", ["key", "value"])
def test_a_key_or_value_BELOW_the_assumed_minimum_is_caught(pl, dtype, violate):
    """Round-5 review, weak 1(i): the guessed bounds were only tested from above.  A key below the assumed minimum wraps (key - kmin) around as an unsigned id -- far
    outside the table, reported by the partition check; a value below the assumed minimum does not fit its narrowed field -- reported by the per-row source check."""
    from polars_amd import queries
    rng = np.random.default_rng(17)
    n = 17_000_000
    np_t = {"Int64": np.int64, "Int32": np.int32, "UInt32": np.uint32}[dtype]
    base = 2_000_000                                    # the sampled minimum sits well above zero: a guess that starts at its own minimum, not at 0
    ids = (base + rng.integers(0, 300_000, n)).astype(np_t)
    v = (5_000_000 + rng.integers(0, 1000, n)).astype(np.int64)
    hole = _outside_the_sample(n)
    if violate == "key":
        ids[hole] = np.arange(len(hole)).astype(np_t) + (0 if dtype == "UInt32" else 0)          # far below the sampled minimum
        if dtype != "UInt32":
            ids[hole[:2]] = np.array([-5, -70_000], dtype=np_t)                                    # and below zero for the signed dtypes
    else:
        v[hole] = -(1 << 35) - np.arange(len(hole))                                              # far below the sampled minimum of the narrowed value column
    df = pl.DataFrame([pl.Series("key", ids), pl.Series("v", v)])
    out = queries.cfg3(df.lazy()).collect().sort_host("key")
    plan = pl.last_plan()
    keys, inv = np.unique(ids.astype(np.int64), return_inverse=True)
    assert np.array_equal(np.array(out["key"], dtype=np.int64), keys), plan
    cols = [c for c in out if c != "key"]
    sums = np.zeros(len(keys), np.int64); np.add.at(sums, inv, v)
    got = [np.array(out[c], dtype=np.int64) for c in cols]
    assert any(np.array_equal(g, sums) for g in got) and any(np.array_equal(g, np.bincount(inv)) for g in got), plan
    # (whether the planner guessed at all depends on the dtype -- narrow keys get exact statistics cheaply; when it did, the violation must have been noticed)
    if "bounds assumed from the sample" in plan or "AssumedBoundsViolated{" in plan:
        assert "AssumedBoundsViolated{" in plan, plan


@pytest.mark.parametrize("violate", [False, True])
def test_a_value_sent_as_a_48_bit_offset_is_checked_against_its_assumed_bounds(pl, violate):
    """Sparse 64-bit keys (hash partitions) and an Int64 value spanning 2^41: the value travels as a 48-bit offset from its minimum, two rows a record (fused::kPackPairV).  On the
    first run that minimum is the planner's GUESS from a strided sample: a value the offset cannot hold, placed where the sample does not look, must be reported by the
    scatter's per-row check and the query planned again from exact statistics -- never a truncated sum."""
    from polars_amd import queries
    rng = np.random.default_rng(14)
    n = 17_000_000
    key = (rng.integers(0, 300_000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).astype(np.int64)
    v = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    if violate:
        v[_outside_the_sample(n)] = (1 << 60) + np.arange(6)
    df = pl.DataFrame({"key": key, "v": v})
    out = queries.cfg3(df.lazy()).collect().sort_host("key")
    plan = pl.last_plan()
    assert ("AssumedBoundsViolated{" in plan) == violate, plan
    assert "partitioned(v3,hash" in plan and ("pack=5" in plan) == (not violate), plan      # (exact bounds of 2^60: the values travel whole)
    uk, inv = np.unique(key, return_inverse=True)
    sums = np.zeros(len(uk), np.int64); np.add.at(sums, inv, v)
    assert np.array_equal(np.array(out["key"], dtype=np.int64), uk)
    got = [np.array(out[c], dtype=np.int64) for c in out if c != "key"]
    assert any(np.array_equal(g, sums) for g in got) and any(np.array_equal(g, np.bincount(inv)) for g in got)
    out2 = queries.cfg3(df.lazy()).collect().sort_host("key")
    assert "AssumedBoundsViolated{" not in pl.last_plan() and out2 == out
