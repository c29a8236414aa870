"""bench.py's checks of the timed results (`verified`), exercised without a GPU: a stand-in "library result" computed with
plain numpy from the generators' host twins must pass, the same result with one perturbed cell must fail.  The checks
themselves compare against the CPU oracle (oracle/pyoracle.py) evaluated block by block over the host twin."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from polars_amd import datagen  # noqa: E402


class Col:
    def __init__(self, a): self.a = np.asarray(a)
    def to_numpy(self): return self.a


class Frame(dict):
    def __getitem__(self, k): return Col(dict.__getitem__(self, k))
    def raw(self, k): return dict.__getitem__(self, k)


def test_q1_blocks_match_single_pass_and_detect_corruption():
    from oracle import pyoracle as orc
    orc.set_threads(4)
    n, seed = 300_000, 10
    want, done, t, first = bench.q1_oracle_blocks(n, seed, budget_s=60, block=70_000)     # 5 blocks, ragged tail
    assert done == n and t > 0 and len(first["l_shipdate"]) == 70_000
    cols = datagen.lineitem_native_host(0, n, seed)
    one = orc.q1({k: cols[k] for k in datagen.LINEITEM_Q1_COLS}, datagen.us(1998, 9, 2))
    got = {k: one[k].tolist() for k in one}
    got["l_returnflag"] = [datagen.FLAGS[c] for c in got["l_returnflag"]]               # the library returns categories
    got["l_linestatus"] = [datagen.STATUS[c] for c in got["l_linestatus"]]
    v = bench.compare_q1(got, want)
    assert v["ok"] and v["max_rel_err"] < 1e-9, v
    bad = dict(got); bad["sum_qty"] = list(got["sum_qty"]); bad["sum_qty"][0] += 1
    assert not bench.compare_q1(bad, want)["ok"]
    bad = dict(got); bad["sum_charge"] = [x * (1 + 3e-6) for x in got["sum_charge"]]
    assert not bench.compare_q1(bad, want)["ok"]
    # the baseline leg on the same rows: returns the verification with it
    base, ver = bench.cpu_baseline_q1(1.0, rows=n, seed=seed, gpu_result=got)
    assert ver["ok"] and ver["rows"] == n and base["kind"] in ("port", "reference") and "SAME rows" in base["sample"]


def test_cfg2_check():
    n, seed = 250_000, 20
    a = datagen.uniform_native_host("Int64", 0, n, seed, 0, 0, 2 ** 31)
    x = datagen.uniform_native_host("Float64", 0, n, seed, 1, 0, 10 ** 9, 1e-7)
    y = datagen.uniform_native_host("Float64", 0, n, seed, 2, 0, 10 ** 9, 1e-9)
    m = a > 2 ** 30
    got = {"xy": [float((x[m] * (1 - y[m])).sum())], "x_mean": [float(x[m].mean())], "a_sum": [int(a[m].sum())]}
    v = bench.verify_cfg2(got, n, seed, 60, block=100_000)
    assert v["ok"] and v["rows"] == n, v
    got["a_sum"][0] += 1
    assert bench.verify_cfg2(got, n, seed, 60, block=100_000)["ok"] is False
    assert bench.verify_cfg2(got, n, seed, 0.0)["ok"] is None          # no budget: reported as not covered, never as passed


def test_cfg2_check_with_nulls():
    """the nullable variant (cfg2_nulls5pct_1e9): x is null where the generator's stream-3 value is below 5; a result that ignored the bitmap fails"""
    n, seed = 250_000, 20
    a = datagen.uniform_native_host("Int64", 0, n, seed, 0, 0, 2 ** 31)
    x = datagen.uniform_native_host("Float64", 0, n, seed, 1, 0, 10 ** 9, 1e-7)
    y = datagen.uniform_native_host("Float64", 0, n, seed, 2, 0, 10 ** 9, 1e-9)
    valid = datagen.uniform_native_host("UInt32", 0, n, seed, 3, 0, 100) >= bench.CFG2_NULL_PCT
    assert 0.04 < 1 - valid.mean() < 0.06
    m = a > 2 ** 30
    got = {"xy": [float((x[m & valid] * (1 - y[m & valid])).sum())], "x_mean": [float(x[m & valid].mean())], "a_sum": [int(a[m].sum())]}
    assert bench.verify_cfg2(got, n, seed, 60, block=100_000, null_pct=bench.CFG2_NULL_PCT)["ok"] is True
    blind = {"xy": [float((x[m] * (1 - y[m])).sum())], "x_mean": [float(x[m].mean())], "a_sum": [int(a[m].sum())]}
    assert bench.verify_cfg2(blind, n, seed, 60, block=100_000, null_pct=bench.CFG2_NULL_PCT)["ok"] is False


def test_groupby_checks_zipf_and_sparse_keys():
    """cfg3_zipf_1e9 (keys from the zipf generator's host twin) and cfg3_sparse_keys_1e9 (ids times an odd 64-bit constant: the result keys map back
    through the inverse multiplier)"""
    n, seed, nk = 300_000, 20, 5000
    kz = datagen.zipf_native_host_mt(0, n, seed, 0, nk, threads=2)
    v = datagen.uniform_native_host("Int64", 0, n, seed, 1, 0, 1000)
    present = np.nonzero(np.bincount(kz, minlength=nk))[0]
    f = Frame(key=present[::-1], v_sum=np.bincount(kz, weights=v, minlength=nk).astype(np.int64)[present][::-1], v_count=np.bincount(kz, minlength=nk).astype(np.uint32)[present][::-1])
    gen = lambda r0, m: datagen.zipf_native_host_mt(r0, m, seed, 0, nk, threads=2)
    r = bench.verify_groupby_dense(f, "key", "v_sum", n, seed, nk, "Int64", "Int64", (0, 1000), ("count", "v_count"), 60, block=110_000, key_gen=gen)
    assert r["ok"] and r["groups"] == len(present) < nk, r
    f.raw("v_count")[0] += 1
    assert bench.verify_groupby_dense(f, "key", "v_sum", n, seed, nk, "Int64", "Int64", (0, 1000), ("count", "v_count"), 60, block=110_000, key_gen=gen)["ok"] is False
    ids = datagen.uniform_native_host("Int64", 0, n, seed, 0, 0, nk)
    w = datagen.uniform_native_host("Int64", 0, n, seed, 1, -(1 << 40), 1 << 40)
    assert w.min() < -(1 << 39) and w.max() > (1 << 39)
    sparse = (np.arange(nk).astype(np.uint64) * np.uint64(bench.HASHED_KEY_MULT % (1 << 64))).astype(np.int64)          # what the device multiply produces
    assert len(np.unique(sparse)) == nk and np.array_equal((sparse.astype(np.uint64) * np.uint64(bench.HASHED_KEY_INV)).astype(np.int64), np.arange(nk))
    sums = np.zeros(nk, np.int64); np.add.at(sums, ids, w)
    perm = np.random.default_rng(1).permutation(nk)
    f = Frame(key=sparse[perm], v_sum=sums[perm], v_count=np.bincount(ids, minlength=nk).astype(np.uint32)[perm])
    unmap = lambda k: (k.astype(np.uint64) * np.uint64(bench.HASHED_KEY_INV)).astype(np.int64)
    r = bench.verify_groupby_dense(f, "key", "v_sum", n, seed, nk, "Int64", "Int64", (-(1 << 40), 1 << 40), ("count", "v_count"), 60, block=110_000, key_unmap=unmap)
    assert r["ok"], r
    f.raw("v_sum")[7] -= 1
    assert bench.verify_groupby_dense(f, "key", "v_sum", n, seed, nk, "Int64", "Int64", (-(1 << 40), 1 << 40), ("count", "v_count"), 60, block=110_000, key_unmap=unmap)["ok"] is False


def test_groupby_checks_cfg3_cfg5():
    n, seed, nk = 400_000, 20, 1000
    key = datagen.uniform_native_host("Int64", 0, n, seed, 0, 0, nk)
    v = datagen.uniform_native_host("Int64", 0, n, seed, 1, 0, 1000)
    perm = np.random.default_rng(0).permutation(nk)                    # the library's group order is arbitrary
    f = Frame(key=np.arange(nk)[perm], v_sum=np.bincount(key, weights=v, minlength=nk).astype(np.int64)[perm], v_count=np.bincount(key, minlength=nk).astype(np.uint32)[perm])
    r = bench.verify_groupby_dense(f, "key", "v_sum", n, seed, nk, "Int64", "Int64", (0, 1000), ("count", "v_count"), 60, block=150_000)
    assert r["ok"] and r["groups"] == nk, r
    f.raw("v_sum")[3] += 1
    assert bench.verify_groupby_dense(f, "key", "v_sum", n, seed, nk, "Int64", "Int64", (0, 1000), ("count", "v_count"), 60, block=150_000)["ok"] is False
    codes = datagen.uniform_native_host("UInt32", 0, n, seed, 0, 0, nk)
    w = datagen.uniform_native_host("Float64", 0, n, seed, 1, 0, 10 ** 9, 1e-7)
    s, c = np.bincount(codes, weights=w, minlength=nk), np.bincount(codes, minlength=nk)
    f = Frame(k=np.arange(nk, dtype=np.uint32)[perm], v_sum=s[perm], v_mean=(s / c)[perm])
    r = bench.verify_groupby_dense(f, "k", "v_sum", n, seed, nk, "UInt32", "Float64", (0, 10 ** 9, 1e-7), ("mean", "v_mean"), 60, block=150_000)
    assert r["ok"], r
    f.raw("v_mean")[5] *= 1.00001
    assert bench.verify_groupby_dense(f, "k", "v_sum", n, seed, nk, "UInt32", "Float64", (0, 10 ** 9, 1e-7), ("mean", "v_mean"), 60, block=150_000)["ok"] is False


def test_q3_check():
    from oracle import pyoracle as orc
    orc.set_threads(4)
    no, seed = 60_000, 20
    o, li, cnt = datagen.orders_lineitem_native_host(0, no, no, seed)
    o["o_shippriority"] = np.zeros(no, np.int64)
    w = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: o[k] for k in datagen.ORDERS_Q3_COLS}, datagen.us(1995, 3, 15))
    perm = np.random.default_rng(1).permutation(len(w["l_orderkey"]))
    f = Frame({k: w[k][perm] for k in w})
    r = bench.verify_q3(f, no, seed, 60, block=25_000, oracle_orders=20_000)
    assert r["ok"] and r["covers_whole_input"] and r["groups_checked"] == len(perm) and r["rows"] == no + len(li["l_orderkey"]), r
    f.raw("revenue")[0] *= 1.001
    assert bench.verify_q3(f, no, seed, 60, block=25_000, oracle_orders=20_000)["ok"] is False


def test_q3_three_tables_check():
    from oracle import pyoracle as orc
    orc.set_threads(4)
    no, seed = 60_000, 21
    o, li, cnt = datagen.orders_lineitem_native_host(0, no, no, seed)
    o["o_shippriority"] = np.zeros(no, np.int64)
    cust = datagen.customer_native_host(0, datagen.n_customers_for(no), seed)
    w = orc.q3_full(cust, {k: o[k] for k in datagen.ORDERS_Q3_COLS}, {k: li[k] for k in datagen.LINEITEM_Q3_COLS}, datagen.us(1995, 3, 15), datagen.SEGMENTS.index("BUILDING"))
    assert len(w["o_orderkey"]) > 50
    perm = np.random.default_rng(2).permutation(len(w["o_orderkey"]))
    f = Frame({k: w[k][perm] for k in w})
    r = bench.verify_q3(f, no, seed, 60, block=25_000, oracle_orders=20_000, customer_seed=seed)
    assert r["ok"] and r["covers_whole_input"] and r["groups_checked"] == len(perm), r
    # the two-table stand-in (o_custkey % 5 == 0) is a different query: its check must reject this result
    f2 = Frame({("l_orderkey" if k == "o_orderkey" else k): w[k][perm] for k in w})
    assert bench.verify_q3(f2, no, seed, 60, block=25_000, oracle_orders=20_000)["ok"] is False


def test_q1_results_compare_across_key_representations():
    """bench.py at N > 1: the combined Q1 result went through pack_q1 / unpack_q1 (the two keys as dictionary codes), the per-rank results come straight from
    to_dict() (strings) -- compare_q1_dicts must see the same groups in both (it compared 0 with 'A' and reported a failed verification for every real
    multi-GPU Q1 run; the gloo dry run's stand-in results carry codes on both sides and never showed it)."""
    import bench
    from polars_amd import datagen
    a = {"l_returnflag": ["A", "N", "R"], "l_linestatus": ["F", "O", "F"], "sum_qty": [10, 20, 30], "count_order": [1, 2, 3], "sum_base_price": [1.5, 2.5, 3.5],
         "sum_disc_price": [1.0, 2.0, 3.0], "sum_charge": [1.1, 2.2, 3.3], "avg_qty": [10.0, 10.0, 10.0], "avg_price": [1.5, 1.25, 3.5 / 3], "avg_disc": [0.1, 0.2, 0.3]}
    packed = bench.unpack_q1(bench.pack_q1(a))
    assert len(packed) == 1 and packed[0]["l_returnflag"] == [datagen.FLAGS.index(v) for v in a["l_returnflag"]]
    combined = bench.combine_q1_results(packed)                                  # what every rank holds after the all-gather (codes)
    assert bench.compare_q1_dicts(combined, bench.combine_q1_results([a]))       # ... against the merge of the ranks' own results (strings)
    b = dict(a, sum_qty=[10, 21, 30])
    assert not bench.compare_q1_dicts(combined, bench.combine_q1_results([b]))
    c = dict(a, l_returnflag=["A", "N", "N"])
    assert not bench.compare_q1_dicts(combined, bench.combine_q1_results([c]))
