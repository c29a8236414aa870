"""Pins the CPU oracle (oracle/plx_oracle.cpp) against the known-answer vectors transcribed
from the reference's own tests (tests/golden/reference_kats.json, each case cites its
source) and against independent implementations (numpy, pyarrow, pandas), the way the
reference cross-checks against pandas (py-polars/tests/unit/streaming/test_streaming_join.py:61-112).
CPU only."""
import math

import numpy as np
import pytest

from tests import kat

AGG = {"sum": 0, "mean": 1, "min": 2, "max": 3, "count": 4, "len": 5}


def _to_py(v, ok):
    if not ok:
        return None
    return v.item() if hasattr(v, "item") else v


@pytest.mark.parametrize("case", kat.load_cases("groupby"), ids=lambda c: c["id"])
@pytest.mark.parametrize("threads", [1, 3])
def test_groupby_kats(orc, case, threads):
    orc.set_threads(threads)
    keys, kvalid, cats = [], [], {}
    for name, spec in case["keys"].items():
        a, v, c = kat.column(spec, case["key_dtypes"][name])
        keys.append(a); kvalid.append(v); cats[name] = c
    aggs = []
    for col, op in case["aggs"]:
        a, v, _ = kat.column(case["values"][col], case["value_dtypes"][col])
        aggs.append((f"{col}_{op}", AGG[op], a, v))
    r = orc.q_groupby(keys, kvalid, aggs, maintain_order=case["maintain_order"])
    G = len(r["key_0"][0])
    rows = []
    for g in range(G):
        row = []
        for i, name in enumerate(case["keys"]):
            v, ok = r[f"key_{i}"]
            x = _to_py(v[g], ok[g])
            if x is not None and cats[name] is not None:
                x = cats[name][x]
            row.append(x)
        for name, _, _, _ in aggs:
            v, ok = r[name]
            row.append(_to_py(v[g], ok[g]))
        rows.append(row)
    exp_cols = list(case["keys"].keys()) + [n for n, _, _, _ in aggs]
    exp_rows = [[case["expect"][c][g] for c in exp_cols] for g in range(len(case["expect"][exp_cols[0]]))]
    if not case["maintain_order"]:
        keyf = lambda row: tuple((x is None, x) for x in row[: len(case["keys"])])
        rows.sort(key=keyf); exp_rows.sort(key=keyf)
    assert len(rows) == len(exp_rows)
    for got, exp in zip(rows, exp_rows):
        for g, e in zip(got, exp):
            assert kat.same_value(g, e), (case["id"], got, exp)
    for name, dt in case.get("expect_dtypes", {}).items():
        assert r[name][0].dtype == kat.NP[dt], (case["id"], name, r[name][0].dtype)
    orc.set_threads(1)


@pytest.mark.parametrize("case", kat.load_cases("reduce"), ids=lambda c: c["id"])
def test_reduce_kats(orc, case):
    a, v, _ = kat.column(case["values"], case["dtype"])
    got, odt = orc.reduce(AGG[case["op"]], a, v)
    assert kat.same_value(got, case["expect"], case.get("rtol", 1e-12))
    if "expect_dtype" in case:
        assert odt == orc.DT_OF[np.dtype(kat.NP[case["expect_dtype"]])], (case["id"], odt)


ARITH_OPS = {"add": "ADD", "sub": "SUB", "mul": "MUL", "true_div": "TRUE_DIV", "floor_div": "FLOOR_DIV", "mod": "MOD"}


def _operand(spec, dtype):
    """KAT operand -> (array or python scalar, validity or None, is_scalar)"""
    if isinstance(spec, dict) and "scalar" in spec:
        return kat.NP[dtype](kat.scalar(spec["scalar"])).item(), None, True
    a, v, _ = kat.column(spec, dtype)
    return a, v, False


@pytest.mark.parametrize("case", kat.load_cases("binary"), ids=lambda c: c["id"])
def test_binary_arithmetic_kats(orc, case):
    """One arithmetic operator on primitive columns, column / scalar on either side, nulls in -> nulls out (the validity of a binary kernel's result is the AND
    of its inputs' validities, crates/polars-core/src/chunked_array/ops/arity.rs:203-214, plus the divisor != 0 mask of integer floor-div / mod)."""
    l, lv, ls = _operand(case["lhs"], case["dtype"])
    r, rv, rs = _operand(case["rhs"], case["dtype"])
    op = getattr(orc, ARITH_OPS[case["op"]])
    vals, extra = orc.arith(op, l, r, mode=2 if ls else 1 if rs else 0)
    assert vals.dtype == kat.NP[case["expect_dtype"]], (vals.dtype, case["expect_dtype"])
    valid = np.ones(len(vals), bool)
    for v in (lv, rv, extra):
        if v is not None:
            valid &= v
    got = [vals[i].item() if valid[i] else None for i in range(len(vals))]
    assert len(got) == len(case["expect"])
    for g, e in zip(got, case["expect"]):
        assert kat.same_value(g, e, 1e-15), (case["id"], got, case["expect"])


CMP_OPS = {"eq": "EQ", "ne": "NE", "lt": "LT", "le": "LE", "gt": "GT", "ge": "GE"}
FLIP = {"eq": "eq", "ne": "ne", "lt": "gt", "le": "ge", "gt": "lt", "ge": "le"}


@pytest.mark.parametrize("case", kat.load_cases("compare"), ids=lambda c: c["id"])
def test_compare_kats(orc, case):
    l, lv, ls = _operand(case["lhs"], case["dtype"])
    r, rv, rs = _operand(case["rhs"], case["dtype"])
    for name, exp in case["expect"].items():
        if ls:          # scalar on the left: the comparison is evaluated with the operands swapped (the broadcast kernels take the scalar on the right)
            vals = orc.cmp(getattr(orc, CMP_OPS[FLIP[name]]), r, l)
            valid = rv
        else:
            vals = orc.cmp(getattr(orc, CMP_OPS[name]), l, r)
            valid = lv if rs else (None if lv is None and rv is None else (np.ones(len(vals), bool) if lv is None else lv) & (np.ones(len(vals), bool) if rv is None else rv))
        got = [bool(vals[i]) if (valid is None or valid[i]) else None for i in range(len(vals))]
        assert got == exp, (case["id"], name, got, exp)


@pytest.mark.parametrize("case", kat.load_cases("bool_logic"), ids=lambda c: c["id"])
def test_bool_logic_kats(case):
    """Kleene and / or / not on nullable Booleans (crates/polars-core/src/chunked_array/comparison/mod.rs test_bitwise_ops, test_kleene) against the numpy
    restatement the compiled-program interpreter uses (tests/program_eval.py)."""
    from tests import program_eval as pe
    l, lv, _ = kat.column(case["lhs"], "bool")
    r, rv, _ = kat.column(case["rhs"], "bool")
    lv = np.ones(len(l), bool) if lv is None else lv
    rv = np.ones(len(r), bool) if rv is None else rv
    for name, exp in case["expect"].items():
        if name == "not_rhs":
            vals, valid = ~r, rv
        else:
            vals, valid = pe.kleene(name, l, lv, r, rv)
        got = [bool(vals[i]) if valid[i] else None for i in range(len(vals))]
        assert got == exp, (case["id"], name, got, exp)


def _join_frames(orc, case):
    L, R = {}, {}
    cats = {}
    # strings on both sides share one dictionary
    for side, dst in (("left", L), ("right", R)):
        for name, spec in case[side].items():
            dt = case[side + "_dtypes"][name]
            if dt == "str":
                allv = sorted({v for s in ("left", "right") for v in case[s].get(name, []) if v is not None})
                lut = {c: i for i, c in enumerate(allv)}
                vals = list(spec)
                valid = np.array([x is not None for x in vals])
                dst[name] = (np.array([lut[x] if x is not None else 0 for x in vals], dtype=np.uint32), None if valid.all() else valid)
                cats[name] = allv
            else:
                a, v, _ = kat.column(spec, dt)
                dst[name] = (a, v)
    return L, R, cats


@pytest.mark.parametrize("case", kat.load_cases("join"), ids=lambda c: c["id"])
@pytest.mark.parametrize("threads", [1, 2, 7])
def test_join_kats(orc, case, threads):
    """crates/polars/tests/it/core/joins.rs:53-78 runs the same vectors for 1..7 threads."""
    orc.set_threads(threads)
    L, R, cats = _join_frames(orc, case)
    on = case["on"]
    how = orc.JOIN_LEFT if case["how"] == "left" else orc.JOIN_INNER
    if isinstance(on, list):
        # several key columns: the rows' key tuples numbered over both sides (what the engine's key packing amounts to); a null in any column = a null key
        def tuples(F):
            valid = np.ones(len(F[on[0]][0]), bool)
            for c in on:
                if F[c][1] is not None:
                    valid &= F[c][1]
            return [tuple(int(F[c][0][i]) for c in on) if valid[i] else None for i in range(len(valid))], valid
        lt, lvalid = tuples(L); rt, rvalid_k = tuples(R)
        ids = {t: i for i, t in enumerate(sorted({t for t in lt + rt if t is not None}))}
        lk = np.array([ids[t] if t is not None else 0 for t in lt], dtype=np.int64); rk = np.array([ids[t] if t is not None else 0 for t in rt], dtype=np.int64)
        li, ri, rvalid = orc.join(how, lk, None if lvalid.all() else lvalid, rk, None if rvalid_k.all() else rvalid_k)
    else:
        li, ri, rvalid = orc.join(how, L[on][0], L[on][1], R[on][0], R[on][1])
    if "expect_rows" in case:
        assert len(li) == case["expect_rows"], (case["id"], len(li))
    out = {}
    for name, (a, v) in L.items():
        vals, ok = orc.gather(a, v, li)
        out[name] = [(_to_py(x, o)) for x, o in zip(vals, ok)]
    for name, (a, v) in R.items():
        if name == on or (isinstance(on, list) and name in on):
            continue
        vals, ok = orc.gather(a, v, ri, rvalid)
        nm = name + "_right" if name in out else name
        out[nm] = [(_to_py(x, o)) for x, o in zip(vals, ok)]
    if "expect" in case:
        exp = case["expect"]
        cols = list(exp.keys())
        got_rows = sorted([tuple(out[c][i] for c in cols) for i in range(len(li))], key=lambda r: tuple((x is None, x) for x in r))
        exp_rows = sorted([tuple(exp[c][i] for c in cols) for i in range(len(exp[cols[0]]))], key=lambda r: tuple((x is None, x) for x in r))
        assert len(got_rows) == len(exp_rows), (got_rows, exp_rows)
        for g, e in zip(got_rows, exp_rows):
            for a, b in zip(g, e):
                assert kat.same_value(a, b), (case["id"], got_rows, exp_rows)
    elif "expect_column_sorted_by_key" in case:
        for c, expv in case["expect_column_sorted_by_key"].items():
            order = sorted(range(len(li)), key=lambda i: (out[on][i], i))
            # probe order (left rows in order, build duplicates in insertion order) == maintain_order="left_right"
            assert [out[c][i] for i in order] == expv
    for c, n_null in case.get("expect_null_count", {}).items():
        assert sum(x is None for x in out[c]) == n_null, (case["id"], c, out[c])
    orc.set_threads(1)


def test_total_ordering_floats(orc):
    """py-polars/tests/unit/operations/test_comparison.py:209-226: normal < nan, nan == nan."""
    case = kat.load_cases("cmp_total_order")[0]
    vals = [kat.scalar(v) for v in case["values"] if v is not None]
    for dt in case["dtypes"]:
        a = np.array([l for l in vals for _ in vals], dtype=kat.NP[dt])
        b = np.array([r for _ in vals for r in vals], dtype=kat.NP[dt])

        def ref(l, r):
            if math.isnan(l) and math.isnan(r): return "="
            if math.isnan(l) or l > r: return ">"
            if math.isnan(r) or l < r: return "<"
            return "="
        order = [ref(float(l), float(r)) for l, r in zip(a, b)]
        exp = {orc.EQ: [o == "=" for o in order], orc.NE: [o != "=" for o in order], orc.LT: [o == "<" for o in order],
               orc.LE: [o in "<=" for o in order], orc.GT: [o == ">" for o in order], orc.GE: [o in ">=" for o in order]}
        for op, e in exp.items():
            assert orc.cmp(op, a, b).tolist() == e, (dt, op)
            # broadcast form (verify_total_ordering_broadcast)
            for j, r in enumerate(vals):
                sel = slice(j, None, len(vals))
                assert orc.cmp(op, a[sel].copy(), kat.NP[dt](r)).tolist() == e[sel], (dt, op, r)


def test_filter_sweep(orc):
    case = kat.load_cases("filter_sweep")[0]
    for dt in case["dtypes"]:
        for size in case["sizes"]:
            for sel in case["selectivities"]:
                p, m, exp = kat.filter_sweep_inputs(dt, size, sel)
                got, _ = orc.filter(p, None, m)
                assert got.dtype == exp.dtype and np.array_equal(got, exp), (dt, size, sel)


@pytest.mark.parametrize("case", kat.load_cases("arith"), ids=lambda c: c["id"])
def test_arith_kats(orc, case):
    a, _, _ = kat.column(case["lhs"], case["dtype"])
    b, _, _ = kat.column(case["rhs"], case["dtype"])
    OPS = {"add": orc.ADD, "sub": orc.SUB, "mul": orc.MUL, "floor_div": orc.FLOOR_DIV, "mod": orc.MOD}
    for name, exp in case["expect"].items():
        vals, extra = orc.arith(OPS[name], a, b)
        got = [None if (extra is not None and not extra[i]) else int(vals[i]) for i in range(len(vals))]
        assert got == exp, (name, got, exp)
        # scalar forms must agree with the column form element by element
        for i in range(len(a)):
            v1, e1 = orc.arith(OPS[name], a[i:i + 1].copy(), b[i].item(), mode=1)
            g1 = None if (e1 is not None and not e1[0]) else int(v1[0])
            assert g1 == exp[i], (name, "col-scalar", i)


# ---- independent cross-checks --------------------------------------------------------------------
def test_float_sum_matches_numpy_pairwise_tolerance(orc):
    rng = np.random.default_rng(0)
    for n in [0, 1, 127, 128, 129, 1000, 4096, 100_003]:
        x = rng.uniform(-1, 1, n) * 1e6
        got, _ = orc.reduce(orc.AGG_SUM, x)
        assert math.isclose(got, math.fsum(x), rel_tol=1e-9, abs_tol=1e-6)
        v = rng.uniform(size=n) < 0.9
        if n:
            got, _ = orc.reduce(orc.AGG_SUM, x, v)
            assert math.isclose(got, math.fsum(x[v]), rel_tol=1e-9, abs_tol=1e-6)


def test_int_sums_wrap_and_upcast(orc):
    rng = np.random.default_rng(1)
    for dt in ["i8", "i16", "u8", "u16"]:
        x = rng.integers(np.iinfo(kat.NP[dt]).min, np.iinfo(kat.NP[dt]).max, 10_000, dtype=kat.NP[dt])
        got, odt = orc.reduce(orc.AGG_SUM, x)
        assert odt == orc.I64 and got == int(x.astype(np.int64).sum())
    x = np.array([2**62, 2**62, 2**62], dtype=np.int64)
    got, odt = orc.reduce(orc.AGG_SUM, x)
    assert odt == orc.I64 and got == ((3 * 2**62 + 2**63) % 2**64) - 2**63
    xi = np.array([2**31 - 1, 1], dtype=np.int32)
    got, odt = orc.reduce(orc.AGG_SUM, xi)
    assert odt == orc.I32 and got == -2**31


def test_groupby_matches_pandas(orc):
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(2)
    n = 20_000
    key = rng.integers(0, 100, n).astype(np.int64)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    x = rng.uniform(0, 100, n)
    xv = rng.uniform(size=n) > 0.05   # 5% nulls as in benchmark/conftest.py:7-9
    for threads in (1, 4):
        orc.set_threads(threads)
        r = orc.q_groupby([key], [None], [("s", orc.AGG_SUM, v, None), ("c", orc.AGG_COUNT, x, xv), ("m", orc.AGG_MEAN, x, xv),
                                          ("mn", orc.AGG_MIN, x, xv), ("mx", orc.AGG_MAX, v, None), ("n", orc.AGG_LEN, None, None)])
        order = np.argsort(r["key_0"][0])
        df = pd.DataFrame({"k": key, "v": v, "x": np.where(xv, x, np.nan)})
        g = df.groupby("k")
        assert np.array_equal(r["key_0"][0][order], np.array(sorted(g.groups.keys())))
        assert np.array_equal(r["s"][0][order], g["v"].sum().to_numpy())
        assert np.array_equal(r["c"][0][order], g["x"].count().to_numpy())
        assert np.allclose(r["m"][0][order], g["x"].mean().to_numpy(), rtol=1e-12)
        assert np.array_equal(r["mn"][0][order], g["x"].min().to_numpy())
        assert np.array_equal(r["mx"][0][order], g["v"].max().to_numpy())
        assert np.array_equal(r["n"][0][order], g.size().to_numpy())
    orc.set_threads(1)


def test_join_matches_pandas(orc):
    """inner / left on one key vs pandas.merge after sorting (test_streaming_join.py:61-112)."""
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(3)
    lk = rng.integers(0, 500, 5000).astype(np.int64)
    rk = rng.integers(0, 700, 3000).astype(np.int64)
    for how, name in ((orc.JOIN_INNER, "inner"), (orc.JOIN_LEFT, "left")):
        for threads in (1, 5):
            orc.set_threads(threads)
            li, ri, rv = orc.join(how, lk, None, rk, None)
            got = sorted(zip(li.tolist(), [(-1 if (rv is not None and not ok) else int(r)) for r, ok in zip(ri, rv if rv is not None else np.ones(len(ri), bool))]))
            m = pd.merge(pd.DataFrame({"k": lk, "li": np.arange(len(lk))}), pd.DataFrame({"k": rk, "ri": np.arange(len(rk))}), on="k", how=name)
            exp = sorted(zip(m["li"].tolist(), m["ri"].fillna(-1).astype(int).tolist()))
            assert got == exp
    orc.set_threads(1)


def test_hash_partition_is_stable_and_balanced(orc):
    keys = np.arange(100_000, dtype=np.int64)
    p = orc.hash_partition(keys, None, 8, seed=0)
    assert p.max() == 7 and p.min() == 0
    counts = np.bincount(p, minlength=8)
    assert counts.min() > 100_000 / 8 * 0.9
    valid = np.ones(len(keys), bool); valid[::7] = False
    pn = orc.hash_partition(keys, valid, 8, seed=0)
    assert (pn[::7] == 0).all() and np.array_equal(pn[valid], p[valid])   # nulls -> partition 0


def test_q1_native_drivers_agree(orc):
    """The C++ Q1 drivers used as bench.py's cpu_baseline (in-memory GroupByExec shape and the streaming /
    partitioned group-by shape) agree with the step-by-step Python-driven oracle."""
    from polars_amd import datagen
    li = datagen.lineitem_host(300_000, seed=5)
    cols = {k: li[k] for k in datagen.LINEITEM_Q1_COLS}
    cut = datagen.us(1998, 9, 2)
    ref = orc.q1(cols, cut)
    for threads in (1, 4):
        orc.set_threads(threads)
        for streaming in (False, True):
            got = orc.q1_native(cols, cut, streaming=streaming, morsel=7_000)
            for k in ref:
                if ref[k].dtype.kind == "f":
                    assert np.allclose(got[k], ref[k], rtol=1e-9, atol=0), (k, streaming, threads)
                else:
                    assert np.array_equal(got[k], ref[k]), (k, streaming, threads)
    orc.set_threads(1)


def _kat_keys(case, by, desc, nl):
    cols = {n: kat.column(spec, case["dtypes"][n]) for n, spec in case["frame"].items()}
    keys = [(cols[b][0], cols[b][1], bool(d), bool(x)) for b, d, x in zip(by, desc, nl)]
    return cols, keys


def _check_sorted_frame(case, cols, idx):
    exp = case["expect"]
    names = list(exp.keys())
    got = []
    for i in idx:
        got.append(tuple(_to_py(cols[c][0][i], cols[c][1] is None or cols[c][1][i]) for c in names))
    want = [tuple(exp[c][i] for c in names) for i in range(len(exp[names[0]]))]
    if case.get("unordered"):
        srt = lambda rows: sorted(rows, key=lambda r: tuple((x is None, x) for x in r))
        got, want = srt(got), srt(want)
    assert len(got) == len(want), (got, want)
    for g, e in zip(got, want):
        for a, b in zip(g, e):
            assert kat.same_value(a, b), (case["id"], got, want)


@pytest.mark.parametrize("case", kat.load_cases("sort"), ids=lambda c: c["id"])
@pytest.mark.parametrize("impl", ["cmp", "lexsort"])
def test_sort_kats(orc, case, impl):
    """Both restatements of arg_sort_multiple (comparator and vectorised) against the reference's sort tests."""
    cols, keys = _kat_keys(case, case["by"], case["descending"], case["nulls_last"])
    fn = orc.sort_indices_cmp if impl == "cmp" else orc.sort_indices
    _check_sorted_frame(case, cols, fn(keys, case.get("limit")))


@pytest.mark.parametrize("case", kat.load_cases("top_k"), ids=lambda c: c["id"])
def test_top_k_kats(orc, case):
    # top_k(k, by, reverse) == sort(by, descending = not reverse, nulls_last).head(k); bottom_k: descending = reverse
    desc = [bool(r) if case["bottom"] else (not r) for r in case["reverse"]]
    cols, keys = _kat_keys(case, case["by"], desc, [True] * len(desc))
    _check_sorted_frame(case, cols, orc.sort_indices(keys, case["k"]))
    _check_sorted_frame(case, cols, orc.sort_indices_cmp(keys, case["k"]))


@pytest.mark.parametrize("case", kat.load_cases("semi_anti"), ids=lambda c: c["id"])
def test_semi_anti_kats(orc, case):
    L, R, cats = _join_frames(orc, case)
    on = case["on"]
    how = orc.JOIN_SEMI if case["how"] == "semi" else orc.JOIN_ANTI
    if isinstance(on, list):
        lk, lv, rk, rv = orc.encode_key_rows([L[c] for c in on], [R[c] for c in on])
        idx = orc.semi_anti_join(how, lk, lv, rk, rv)
    else:
        idx = orc.semi_anti_join(how, L[on][0], L[on][1], R[on][0], R[on][1])
    for name, expv in case["expect"].items():
        a, v = L[name]
        got = [_to_py(a[i], v is None or v[i]) for i in idx]
        if name in cats:
            got = [None if g is None else cats[name][g] for g in got]
        assert len(got) == len(expv) and all(kat.same_value(g, e) for g, e in zip(got, expv)), (case["id"], name, got, expv)


def test_sort_restatements_agree_on_random_inputs(orc):
    rng = np.random.default_rng(5)
    for t in range(120):
        n = int(rng.integers(0, 80))
        keys = []
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 3))
            v = (rng.integers(-3, 4, n).astype(np.int64) if kind == 0 else
                 rng.choice([0.0, -0.0, 1.5, -2.0, np.nan, np.inf, -np.inf], n) if kind == 1 else rng.integers(0, 2, n).astype(bool))
            m = None if rng.random() < 0.4 else rng.random(n) < 0.7
            keys.append((v, m, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
        lim = None if rng.random() < 0.5 else int(rng.integers(0, n + 2))
        assert np.array_equal(orc.sort_indices(keys, lim), orc.sort_indices_cmp(keys, lim)), t


def test_numpy_q1_reference_matches_oracle(orc):
    """tests/refs.py q1_numpy (the GPU tests' large-size reference) against the oracle's operator-by-operator Q1."""
    from polars_amd import datagen
    from tests import refs
    cols = datagen.lineitem_host(300_000, seed=9)
    cutoff = datagen.us(1998, 9, 2)
    want = orc.q1({k: cols[k] for k in datagen.LINEITEM_Q1_COLS}, cutoff)
    got = refs.q1_numpy(cols, cutoff)
    assert len(got) == len(want["l_returnflag"]) >= 4
    for i, key in enumerate(zip(want["l_returnflag"].tolist(), want["l_linestatus"].tolist())):
        g = got[key]
        assert g["count_order"] == want["count_order"][i] and g["sum_qty"] == want["sum_qty"][i]
        for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert math.isclose(g[c], want[c][i], rel_tol=1e-9), (key, c)


def test_sort_oracle_against_pyarrow():
    """Independent implementation check (the reference cross-checks against pandas the same way): pyarrow's stable
    sort_indices with one null placement for all keys, integer / NaN-free float / boolean keys, per-key direction."""
    import pyarrow as pa
    import pyarrow.compute as pc
    from oracle import pyoracle as orc
    rng = np.random.default_rng(12)
    for trial in range(60):
        n = int(rng.integers(1, 400))
        nk = int(rng.integers(1, 4))
        nulls_last = bool(rng.integers(0, 2))
        keys, arrays, sort_keys = [], {}, []
        for j in range(nk):
            kind = int(rng.integers(0, 3))
            v = (rng.integers(-5, 6, n).astype(np.int64) if kind == 0 else rng.integers(-8, 9, n).astype(np.float64) / 4 if kind == 1 else rng.integers(0, 2, n).astype(bool))
            m = None if rng.random() < 0.4 else rng.random(n) < 0.75
            desc = bool(rng.integers(0, 2))
            keys.append((v, m, desc, nulls_last))
            arrays[f"k{j}"] = pa.array(v, mask=None if m is None else ~m)
            sort_keys.append((f"k{j}", "descending" if desc else "ascending"))
        want = pc.sort_indices(pa.table(arrays), sort_keys=sort_keys, null_placement="at_end" if nulls_last else "at_start").to_numpy()
        assert np.array_equal(orc.sort_indices(keys), want), (trial, sort_keys, nulls_last)


def test_semi_anti_oracle_against_pandas():
    import pandas as pd
    from oracle import pyoracle as orc
    rng = np.random.default_rng(13)
    for trial in range(20):
        nl, nr = int(rng.integers(0, 300)), int(rng.integers(0, 200))
        lk, rk = rng.integers(0, 60, nl).astype(np.int64), rng.integers(0, 60, nr).astype(np.int64)
        lm, rm = rng.random(nl) < 0.9, rng.random(nr) < 0.8
        left = pd.DataFrame({"k": pd.array(np.where(lm, lk, 0), dtype="Int64")})
        left.loc[~lm, "k"] = pd.NA
        right_keys = set(rk[rm].tolist())
        matched = left["k"].map(lambda x: (x is not pd.NA) and (x in right_keys)).to_numpy(dtype=bool) if nl else np.zeros(0, bool)
        assert np.array_equal(orc.semi_anti_join(orc.JOIN_SEMI, lk, lm, rk, rm), np.nonzero(matched)[0])
        assert np.array_equal(orc.semi_anti_join(orc.JOIN_ANTI, lk, lm, rk, rm), np.nonzero(~matched)[0])


def test_binview_index_map_restatement(orc):
    """The oracle's view index map (binview_index_map.rs restated): dense indices in first-appearance order, equality by length +
    bytes -- strings that share the 4-byte prefix or the length stay distinct, the empty string is a value, nulls get no index."""
    s = ["abcd_x", "abcd_y", "abcd_x", "", None, "abcd", "a much longer string than twelve bytes A", "a much longer string than twelve bytes B", "", "abcd_y"]
    codes, valid, cats = orc.binview_dict_encode(s)
    assert cats == ["abcd_x", "abcd_y", "", "abcd", "a much longer string than twelve bytes A", "a much longer string than twelve bytes B"]
    assert codes.tolist() == [0, 1, 0, 2, 0, 3, 4, 5, 2, 1] and valid.tolist() == [True, True, True, True, False, True, True, True, True, True]
    assert orc.binview_parts(b"twelve bytes") == (12, int.from_bytes(b"twel", "little"), True) and orc.binview_parts(b"thirteen byte")[2] is False
    # the reference's own string-key vector (test_group_by.py:32-52): a, b, a, b, b, c -> three groups in first-appearance order
    c2, v2, k2 = orc.binview_dict_encode(["a", "b", "a", "b", "b", "c"])
    assert k2 == ["a", "b", "c"] and v2 is None and np.bincount(c2, weights=[1, 2, 3, 4, 5, 6]).tolist() == [4.0, 11.0, 6.0]
