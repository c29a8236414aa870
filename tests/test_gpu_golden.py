"""The reference's own known-answer vectors (tests/golden/reference_kats.json) run through
the HIP path (C ABI -> gfx950 kernels).  Same file, same expectations as
tests/test_oracle_golden.py."""
import math

import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu

PLDT = {"i8": "Int8", "i16": "Int16", "i32": "Int32", "i64": "Int64", "u8": "UInt8", "u16": "UInt16", "u32": "UInt32", "u64": "UInt64",
        "f32": "Float32", "f64": "Float64", "bool": "Boolean"}


def _series(pl, name, spec, dtype):
    vals = [kat.scalar(v) for v in kat.expand(spec)]
    if dtype == "str":
        return pl.Series(name, vals)
    return pl.Series(name, vals, dtype=getattr(pl, PLDT[dtype]))


def _agg(pl, col, op):
    e = pl.col(col)
    return {"sum": e.sum, "mean": e.mean, "min": e.min, "max": e.max, "count": e.count, "len": e.len}[op]().alias(f"{col}_{op}")


@pytest.mark.parametrize("case", kat.load_cases("groupby"), ids=lambda c: c["id"])
@pytest.mark.parametrize("no_fusion", [False, True])
def test_groupby_kats(pl, case, no_fusion):
    cols = [_series(pl, n, s, case["key_dtypes"][n]) for n, s in case["keys"].items()]
    cols += [_series(pl, n, s, case["value_dtypes"][n]) for n, s in case["values"].items()]
    df = pl.DataFrame(cols)
    q = df.lazy().group_by(*case["keys"].keys(), maintain_order=case["maintain_order"]).agg(*[_agg(pl, c, o) for c, o in case["aggs"]])
    out = q.collect(no_fusion=no_fusion)
    names = list(case["expect"].keys())
    assert out.columns == names
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
    for name, dt in case.get("expect_dtypes", {}).items():
        assert out.schema[name] == getattr(pl, PLDT[dt]), (name, out.schema)


@pytest.mark.parametrize("case", kat.load_cases("reduce"), ids=lambda c: c["id"])
def test_reduce_kats(pl, case):
    s = _series(pl, "v", case["values"], case["dtype"])
    got = getattr(s, case["op"])()
    assert kat.same_value(got, case["expect"], case.get("rtol", 1e-12))
    # the same through a plan (fused register sink)
    out = pl.DataFrame([s]).lazy().select(getattr(pl.col("v"), case["op"])()).collect()
    assert kat.same_value(out.rows()[0][0], case["expect"], case.get("rtol", 1e-12))
    if "expect_dtype" in case:
        assert out.schema["v"] == getattr(pl, PLDT[case["expect_dtype"]]), out.schema


def _operand(pl, name, spec, dtype):
    if isinstance(spec, dict) and "scalar" in spec:
        return kat.NP[dtype](kat.scalar(spec["scalar"])).item(), True
    return _series(pl, name, spec, dtype), False


@pytest.mark.parametrize("case", kat.load_cases("binary"), ids=lambda c: c["id"])
def test_binary_arithmetic_kats(pl, case):
    """One arithmetic operator through the per-kernel entry points (plx_arith / plx_arith_scalar) AND through a plan (the fused register program where the
    operator fuses, the per-node kernels otherwise)."""
    F = pl._ffi
    OPS = {"add": F.ADD, "sub": F.SUB, "mul": F.MUL, "true_div": F.TRUE_DIV, "floor_div": F.FLOOR_DIV, "mod": F.MOD}
    l, ls = _operand(pl, "l", case["lhs"], case["dtype"])
    r, rs = _operand(pl, "r", case["rhs"], case["dtype"])
    out = r.arith(OPS[case["op"]], l, True) if ls else l.arith(OPS[case["op"]], r)
    assert out.dtype == getattr(pl, PLDT[case["expect_dtype"]]), (out.dtype, case["expect_dtype"])
    got = out.to_list()
    assert len(got) == len(case["expect"])
    for g, e in zip(got, case["expect"]):
        assert kat.same_value(g, e, 1e-15), (case["id"], got, case["expect"])
    # through expressions
    import operator
    pyop = {"add": operator.add, "sub": operator.sub, "mul": operator.mul, "true_div": operator.truediv, "floor_div": operator.floordiv, "mod": operator.mod}[case["op"]]
    cols = [x for x, is_s in ((l, ls), (r, rs)) if not is_s]
    le = pl.lit(l, dtype=getattr(pl, PLDT[case["dtype"]])) if ls else pl.col("l")
    re_ = pl.lit(r, dtype=getattr(pl, PLDT[case["dtype"]])) if rs else pl.col("r")
    res = pl.DataFrame(cols).lazy().select(pyop(le, re_).alias("o")).collect()
    got2 = res["o"].to_list()
    assert res.schema["o"] == getattr(pl, PLDT[case["expect_dtype"]]), res.schema
    for g, e in zip(got2, case["expect"]):
        assert kat.same_value(g, e, 1e-15), (case["id"], "plan", got2, case["expect"])


@pytest.mark.parametrize("case", kat.load_cases("compare"), ids=lambda c: c["id"])
def test_compare_kats(pl, case):
    F = pl._ffi
    OPS = {"eq": F.EQ, "ne": F.NE, "lt": F.LT, "le": F.LE, "gt": F.GT, "ge": F.GE}
    FLIP = {"eq": "eq", "ne": "ne", "lt": "gt", "le": "ge", "gt": "lt", "ge": "le"}
    l, ls = _operand(pl, "l", case["lhs"], case["dtype"])
    r, rs = _operand(pl, "r", case["rhs"], case["dtype"])
    for name, exp in case["expect"].items():
        got = r.cmp(OPS[FLIP[name]], l).to_list() if ls else l.cmp(OPS[name], r).to_list()
        assert got == exp, (case["id"], name, got, exp)


@pytest.mark.parametrize("case", kat.load_cases("bool_logic"), ids=lambda c: c["id"])
def test_bool_logic_kats(pl, case):
    l, r = _series(pl, "l", case["lhs"], "bool"), _series(pl, "r", case["rhs"], "bool")
    for name, exp in case["expect"].items():
        got = (~r).to_list() if name == "not_rhs" else (l & r).to_list() if name == "and" else (l | r).to_list()
        assert got == exp, (case["id"], name, got, exp)
        if name != "not_rhs":      # the same through a fused predicate program (Kleene OP_AND / OP_OR)
            e = (pl.col("l") & pl.col("r")) if name == "and" else (pl.col("l") | pl.col("r"))
            assert pl.DataFrame([l, r]).lazy().select(e.alias("o")).collect()["o"].to_list() == exp, (case["id"], name, "plan")


@pytest.mark.parametrize("case", kat.load_cases("join"), ids=lambda c: c["id"])
def test_join_kats(pl, case):
    # strings on both sides must share one dictionary: build it over both frames
    def frame(side):
        cols = []
        for n, spec in case[side].items():
            dt = case[side + "_dtypes"][n]
            if dt == "str":
                allv = sorted({v for s in ("left", "right") for v in case[s].get(n, []) if v is not None})
                lut = {c: i for i, c in enumerate(allv)}
                codes = [lut[v] if v is not None else None for v in spec]
                cols.append(pl.Series(n, codes, dtype=pl.UInt32))
            else:
                cols.append(_series(pl, n, spec, dt))
        return pl.DataFrame(cols)
    L, R = frame("left"), frame("right")
    out = L.join(R, on=case["on"], how=case["how"])
    if "expect_rows" in case:
        assert out.height == case["expect_rows"], (case["id"], out.height)
    for c, n_null in case.get("expect_null_count", {}).items():
        assert out[c].null_count() == n_null, (case["id"], c)
    if "expect" in case:
        exp = case["expect"]
        names = list(exp.keys())
        d = out.to_dict()
        srt = lambda rows: sorted(rows, key=lambda r: tuple((x is None, x) for x in r))
        got_rows = srt([tuple(d[c][i] for c in names) for i in range(out.height)])
        exp_rows = srt([tuple(exp[c][i] for c in names) for i in range(len(exp[names[0]]))])
        assert len(got_rows) == len(exp_rows), (got_rows, exp_rows)
        for g, e in zip(got_rows, exp_rows):
            for a, b in zip(g, e):
                assert kat.same_value(a, b), (case["id"], got_rows, exp_rows)
        assert [c for c in out.columns if c in names] == names   # column order / `_right` suffix rule
    elif "expect_column_sorted_by_key" in case:
        d = out.to_dict()
        on = case["on"]
        for c, expv in case["expect_column_sorted_by_key"].items():
            assert sorted(zip(d[on], d[c])) == sorted(zip(sorted(d[on]), expv))


def _single_key_join_cases():
    out = []
    for c in kat.load_cases("join"):
        if isinstance(c["on"], str) and "expect" in c and c["how"] in ("inner", "left"):
            out.append(c)
    return out


@pytest.mark.parametrize("case", _single_key_join_cases(), ids=lambda c: c["id"])
def test_join_kats_through_the_fused_join_group_by(pl, case):
    """The reference's single-key inner / left join vectors once more, through the FUSED join -> group-by pipeline (duplicate build keys: row chains; left joins: the
    unmatched rows as groups of their own): group the reference's expected joined frame by (key, one column of the build side) and count -- the same query on the
    device must give exactly those groups, with the fused path where its preconditions hold (integer key; integer build-side group column when build keys repeat)."""
    key = case["on"]
    def frame(side):
        return pl.DataFrame([_series(pl, n, spec, case[side + "_dtypes"][n]) for n, spec in case[side].items()])
    if any(dt == "str" for side in ("left", "right") for dt in case[side + "_dtypes"].values()):
        pytest.skip("string columns travel as dictionary codes in the other join KAT test")
    L, R = frame("left"), frame("right")
    nl, nr = len(next(iter(case["left"].values()))), len(next(iter(case["right"].values())))
    build = "right" if (case["how"] == "left" or nl > nr) else "left"          # the engine's rule: a left join builds on the right table, an inner join on the shorter one
    exp = case["expect"]
    names = list(exp.keys())
    # a non-key column of the build side as it is named in the joined frame (`_right` suffix when both sides have it)
    gcol = None
    for n in case[build]:
        if n == key:
            continue
        joined = n + "_right" if (build == "right" and n in case["left"]) else n
        if joined in names:
            gcol = joined
            break
    gkeys = [key] + ([gcol] if gcol is not None else [])          # (no other build column in the expected frame: the join key alone)
    n_exp = len(exp[names[0]])
    want = {}
    for i in range(n_exp):
        k = tuple(kat.scalar(exp[c][i]) for c in gkeys)
        k = tuple("nan" if isinstance(x, float) and x != x else x for x in k)
        want[k] = want.get(k, 0) + 1
    out = L.lazy().join(R.lazy(), on=key, how=case["how"]).group_by(*gkeys).agg(pl.len().alias("n")).collect()
    plan = pl.last_plan()
    d = out.to_dict()
    got = {}
    for i in range(out.height):
        k = tuple("nan" if isinstance(x, float) and x != x else x for x in (d[c][i] for c in gkeys))
        assert k not in got, (case["id"], k)
        got[k] = d["n"][i]
    assert got == want, (case["id"], plan, got, want)
    bdt = "i64" if gcol is None else case[build + "_dtypes"][[n for n in case[build] if (n + "_right" if (build == "right" and n in case["left"]) else n) == gcol][0]]
    bkeys = [v for v in case[build][key] if v is not None]
    dup = len(set(bkeys)) != len(bkeys)
    if case[build + "_dtypes"][key] not in ("f32", "f64", "str") and not (dup and bdt in ("f32", "f64")):
        assert "FusedJoinGroupBy{" in plan, (case["id"], plan)
        assert ("multi-value" in plan) == dup and ("LeftJoinUnmatched{" in plan) == (case["how"] == "left"), (case["id"], plan)


def test_total_ordering_floats(pl):
    case = kat.load_cases("cmp_total_order")[0]
    vals = [kat.scalar(v) for v in case["values"]]
    for dt in case["dtypes"]:
        npdt = kat.NP[dt]
        lhs = [l for l in vals for _ in vals]
        rhs = [r for _ in vals for r in vals]
        a = pl.Series("l", lhs, dtype=getattr(pl, PLDT[dt]))
        b = pl.Series("r", rhs, dtype=getattr(pl, PLDT[dt]))

        def ref(l, r):
            if l is None or r is None: return None
            l, r = float(npdt(l)), float(npdt(r))
            if math.isnan(l) and math.isnan(r): return "="
            if math.isnan(l) or l > r: return ">"
            if math.isnan(r) or l < r: return "<"
            return "="
        order = [ref(l, r) for l, r in zip(lhs, rhs)]
        F = pl._ffi
        table = {F.EQ: "=", F.NE: "<>", F.LT: "<", F.LE: "<=", F.GT: ">", F.GE: ">="}
        for op, accept in table.items():
            exp = [None if o is None else (o in accept) for o in order]
            assert a.cmp(op, b).to_list() == exp, (dt, op)
            for j, r in enumerate(vals):   # broadcast form
                if r is None:
                    continue
                sub = pl.Series("l", lhs[j::len(vals)], dtype=getattr(pl, PLDT[dt]))
                assert sub.cmp(op, r).to_list() == exp[j::len(vals)], (dt, op, r)


def test_filter_sweep(pl):
    case = kat.load_cases("filter_sweep")[0]
    for di, dt in enumerate(case["dtypes"]):
        for size in case["sizes"]:
            # every size for i64 / bool; a strided subset for the other widths (same kernel template family)
            if dt not in ("i64", "bool") and size % 8 != (di % 8) and size < 64:
                continue
            for sel in case["selectivities"]:
                p, m, exp = kat.filter_sweep_inputs(dt, size, sel)
                s = pl.Series("p", p, dtype=pl.Boolean if dt == "bool" else getattr(pl, PLDT[dt]))
                got = s.filter(pl.Series("m", m, dtype=pl.Boolean)).to_numpy()
                assert np.array_equal(got, exp), (dt, size, sel)


@pytest.mark.parametrize("case", kat.load_cases("arith"), ids=lambda c: c["id"])
def test_arith_kats(pl, case):
    dt = getattr(pl, PLDT[case["dtype"]])
    a, b = pl.Series("a", case["lhs"], dtype=dt), pl.Series("b", case["rhs"], dtype=dt)
    F = pl._ffi
    OPS = {"add": F.ADD, "sub": F.SUB, "mul": F.MUL, "floor_div": F.FLOOR_DIV, "mod": F.MOD}
    for name, exp in case["expect"].items():
        assert a.arith(OPS[name], b).to_list() == exp, name
        for i in range(len(exp)):
            one = pl.Series("a", case["lhs"][i:i + 1], dtype=dt)
            assert one.arith(OPS[name], case["rhs"][i]).to_list() == [exp[i]], (name, i)
    # the same through expressions (fused path cannot do floor-div: falls to per-node kernels)
    df = pl.DataFrame([a, b])
    out = df.lazy().select((pl.col("a") + pl.col("b")).alias("add"), (pl.col("a") // pl.col("b")).alias("floor_div")).collect()
    assert out["add"].to_list() == case["expect"]["add"]
    if "floor_div" in case["expect"]:
        assert out["floor_div"].to_list() == case["expect"]["floor_div"]


def _check_frame(case, d, n_rows):
    exp = case["expect"]
    names = list(exp.keys())
    got = [tuple(d[c][i] for c in names) for i in range(n_rows)]
    want = [tuple(exp[c][i] for c in names) for i in range(len(exp[names[0]]))]
    if case.get("unordered"):
        srt = lambda rows: sorted(rows, key=lambda r: tuple((x is None, x) for x in r))
        got, want = srt(got), srt(want)
    assert len(got) == len(want), (case["id"], got, want)
    for g, e in zip(got, want):
        for a, b in zip(g, e):
            assert kat.same_value(a, b), (case["id"], got, want)


@pytest.mark.parametrize("case", kat.load_cases("sort"), ids=lambda c: c["id"])
def test_sort_kats(pl, case):
    df = pl.DataFrame([_series(pl, n, spec, case["dtypes"][n]) for n, spec in case["frame"].items()])
    lf = df.lazy().sort(case["by"], descending=case["descending"], nulls_last=case["nulls_last"], maintain_order=True)
    if "limit" in case:
        lf = lf.head(case["limit"])
    out = lf.collect()
    assert "_sort[" in pl.last_plan(), pl.last_plan()      # rank_sort (small) or radix_sort
    _check_frame(case, out.to_dict(), out.height)
    # the kernel-level entry gives the same order
    if len(case["by"]) == 1:
        s = df[case["by"][0]]
        idx = s.arg_sort(descending=case["descending"][0], nulls_last=case["nulls_last"][0], limit=case.get("limit", -1))
        a, b = df[case["by"][0]].gather(idx).to_list(), out[case["by"][0]].to_list()
        assert len(a) == len(b) and all(kat.same_value(x, y) for x, y in zip(a, b)), (a, b)


@pytest.mark.parametrize("case", kat.load_cases("top_k"), ids=lambda c: c["id"])
def test_top_k_kats(pl, case):
    df = pl.DataFrame([_series(pl, n, spec, case["dtypes"][n]) for n, spec in case["frame"].items()])
    fn = df.bottom_k if case["bottom"] else df.top_k
    out = fn(case["k"], by=case["by"], reverse=case["reverse"])
    _check_frame(case, out.to_dict(), out.height)


@pytest.mark.parametrize("case", kat.load_cases("semi_anti"), ids=lambda c: c["id"])
def test_semi_anti_join_kats(pl, case):
    luts = {}

    def frame(side):
        cols = []
        for n, spec in case[side].items():
            dt = case[side + "_dtypes"][n]
            if dt == "str":
                allv = sorted({v for s in ("left", "right") for v in case[s].get(n, []) if v is not None})
                luts[n] = allv
                lut = {c: i for i, c in enumerate(allv)}
                cols.append(pl.Series(n, [lut[v] if v is not None else None for v in spec], dtype=pl.UInt32))
            else:
                cols.append(_series(pl, n, spec, dt))
        return pl.DataFrame(cols)
    L, R = frame("left"), frame("right")
    out = L.join(R, on=case["on"], how=case["how"])
    assert out.columns == L.columns                      # left columns only
    # the per-node hash join, or (round 6) the right side as a membership bitmap tested inside the left side's filter scan
    plan = pl.last_plan()
    assert ("hash_semi_join" if case["how"] == "semi" else "hash_anti_join") in plan or ("FusedSemiAntiJoin{" + case["how"]) in plan, plan
    d = out.to_dict()
    for n, cats in luts.items():
        d[n] = [None if c is None else cats[c] for c in d[n]]
    _check_frame(case, d, out.height)                    # left order is part of the contract
