"""The C++ expression compiler without a GPU: the pipelines it emits (plx_debug_program_json: register program, aggregate
cells, key packing, finalisation) are interpreted row by row in numpy (tests/program_eval.py restates the device
semantics of fused_device.hpp) and compared with the oracle / plain numpy on real data.  This pins predicate lowering
(Kleene logic over nulls), arithmetic / casts / literal folding, floor-div and mod by zero, the aggregate decompositions
(mean = f64 sum / count, min / max with NaN and null handling), key packing with null codes, wide keys and float key
canonicalisation -- everything between the IR arenas and the kernels."""
import ctypes as C
import math

import numpy as np
import pytest

import polars_amd as pl
from polars_amd import _ffi as F
from polars_amd import datagen
from polars_amd import queries as Q
from tests import program_eval as pe

NPDT = {np.int8: "Int8", np.int16: "Int16", np.int32: "Int32", np.int64: "Int64", np.uint8: "UInt8", np.uint16: "UInt16", np.uint32: "UInt32", np.uint64: "UInt64",
        np.float64: "Float64", np.bool_: "Boolean"}


def ph_like(name, values, valid=None, dtype=None, stats=True):
    """Placeholder column (schema + min/max statistics, no device memory) mirroring a host array."""
    dt = dtype or getattr(pl, NPDT[values.dtype.type])
    rng = None
    if stats and values.dtype.kind in "iu" and len(values):
        sel = values if valid is None else values[valid]
        if len(sel):
            rng = (int(sel.min()), int(sel.max()))
    h = C.c_uint64()
    F.check(F.lib().plx_column_placeholder(dt.physical, len(values), int(valid is not None), 1 if rng else 0, rng[0] if rng else 0, rng[1] if rng else 0, C.byref(h)))
    return pl.Series._from_handle(name, h.value, dt)


def frame_like(cols, dtypes=None, stats=True):
    return pl.DataFrame([ph_like(n, v, m, (dtypes or {}).get(n), stats) for n, (v, m) in cols.items()])


def by_key(res, key_names):
    """Rows of a group_by result as {key tuple: {column: value}} (None = null, NaN keys canonical)."""
    n = len(next(iter(res.values()))[0])
    def cell(c, i):
        v, m = res[c]
        if m is not None and not m[i]:
            return None
        x = v[i].item()
        return "nan" if isinstance(x, float) and math.isnan(x) else x
    return {tuple(cell(k, i) for k in key_names): {c: cell(c, i) for c in res if c not in key_names} for i in range(n)}


def close(a, b, rtol=1e-9):
    if a is None or b is None or isinstance(a, str) or isinstance(b, str):
        return a == b
    # (abs_tol: a sum of standard normals that nearly cancels -- -8.9e-7 from ~50 terms -- differs in the 15th digit of its TERMS, not of itself)
    return a == b or math.isclose(a, b, rel_tol=rtol, abs_tol=1e-12)


def test_cfg2_program_matches_oracle(orc):
    for null_frac in (0.0, 0.3):
        a, x, y, xv = datagen.cfg2_host(200_000, null_frac)
        cols = {"a": (a, None), "x": (x, xv), "y": (y, None)}
        prog = Q.cfg2(frame_like(cols).lazy()).debug_program()
        got = pe.evaluate(prog, cols)
        want = orc.q_filter_agg_cfg2(a, x, y, 2 ** 30, xv)
        assert got["a_sum"][0][0] == want["a_sum"]
        assert math.isclose(got["xy"][0][0], want["xy"], rel_tol=1e-9) and math.isclose(got["x_mean"][0][0], want["x_mean"], rel_tol=1e-9)


def test_q1_program_matches_oracle(orc):
    li = datagen.lineitem_host(150_000, seed=2)
    cols = {k: (li[k], None) for k in datagen.LINEITEM_Q1_COLS}
    lt = datagen.logical_dtypes(pl)
    prog = Q.q1(frame_like(cols, lt).lazy()).debug_program()
    assert prog["key_plan"]["packed"] and len(prog["inputs"]) == 7
    got = by_key(pe.evaluate(prog, cols), ["l_returnflag", "l_linestatus"])
    want = orc.q1(li, datagen.us(1998, 9, 2))
    assert len(got) == len(want["l_returnflag"])
    for i, key in enumerate(zip(want["l_returnflag"].tolist(), want["l_linestatus"].tolist())):
        g = got[key]
        assert g["count_order"] == want["count_order"][i] and g["sum_qty"] == want["sum_qty"][i]
        for c in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            assert close(g[c], float(want[c][i])), (key, c)


@pytest.mark.parametrize("shape", ["packed_nullable_keys", "wide_keys", "raw_float_key", "raw_i64_key"])
def test_group_by_programs_match_oracle(orc, shape):
    import zlib
    rng = np.random.default_rng(zlib.crc32(shape.encode()) % 1000)       # (hash() of a str changes from process to process: the data must not)
    n = 60_000
    v = rng.integers(-1000, 1000, n).astype(np.int64); vm = rng.random(n) < 0.9
    x = rng.normal(size=n); x[rng.random(n) < 0.02] = np.nan; xm = rng.random(n) < 0.85
    u = rng.integers(0, 2 ** 40, n).astype(np.uint64)
    if shape == "packed_nullable_keys":
        keys = {"k0": (rng.integers(-5, 20, n).astype(np.int32), rng.random(n) < 0.9), "k1": (rng.integers(0, 3, n).astype(np.uint8), None),
                "k2": (rng.integers(0, 2, n).astype(bool), rng.random(n) < 0.95)}
    elif shape == "wide_keys":
        keys = {"k0": (rng.integers(-2 ** 62, 2 ** 62, 40).astype(np.int64)[rng.integers(0, 40, n)], rng.random(n) < 0.9),
                "k1": (rng.integers(-2 ** 62, 2 ** 62, 30).astype(np.int64)[rng.integers(0, 30, n)], None)}
    elif shape == "raw_float_key":
        kf = rng.choice([0.0, -0.0, 1.5, -2.25, np.nan, np.inf, 3.0, 7.5], n)
        keys = {"k0": (kf, rng.random(n) < 0.9)}
    else:
        keys = {"k0": (rng.integers(-2 ** 62, 2 ** 62, 500).astype(np.int64)[rng.integers(0, 500, n)], rng.random(n) < 0.97)}
    cols = dict(keys); cols.update({"v": (v, vm), "x": (x, xm), "u": (u, None)})
    lf = (frame_like(cols, stats=shape != "wide_keys").lazy().filter(pl.col("u") > 2 ** 20)
          .group_by(*keys.keys()).agg(pl.col("v").sum().alias("v_sum"), pl.col("v").mean().alias("v_mean"), pl.col("v").min().alias("v_min"), pl.col("v").max().alias("v_max"),
                                      pl.col("v").count().alias("v_count"), pl.len().alias("n"), pl.col("x").sum().alias("x_sum"), pl.col("x").min().alias("x_min"),
                                      pl.col("x").max().alias("x_max"), pl.col("x").mean().alias("x_mean"), pl.col("u").max().alias("u_max")))
    prog = lf.debug_program()
    kp = prog["key_plan"]
    assert (kp["packed"], kp["wide"]) == {"packed_nullable_keys": (1, 0), "wide_keys": (0, 1), "raw_float_key": (0, 0), "raw_i64_key": (0, 0)}[shape]
    got = by_key(pe.evaluate(prog, cols), list(keys))
    sel = u > 2 ** 20
    fk = [np.ascontiguousarray(kv[sel]) for kv, _ in keys.values()]
    fm = [None if km is None else km[sel] for _, km in keys.values()]
    A = orc
    spec = [("v_sum", A.AGG_SUM, v[sel], vm[sel]), ("v_mean", A.AGG_MEAN, v[sel], vm[sel]), ("v_min", A.AGG_MIN, v[sel], vm[sel]), ("v_max", A.AGG_MAX, v[sel], vm[sel]),
            ("v_count", A.AGG_COUNT, v[sel], vm[sel]), ("n", A.AGG_LEN, None, None), ("x_sum", A.AGG_SUM, x[sel], xm[sel]), ("x_min", A.AGG_MIN, x[sel], xm[sel]),
            ("x_max", A.AGG_MAX, x[sel], xm[sel]), ("x_mean", A.AGG_MEAN, x[sel], xm[sel]), ("u_max", A.AGG_MAX, u[sel], None)]
    r = orc.q_groupby(fk, fm, spec)
    want = {}
    names = list(keys)
    res_named = {names[i]: r[f"key_{i}"] for i in range(len(names))}
    res_named.update({s[0]: r[s[0]] for s in spec})
    want = by_key(res_named, names)
    assert set(got) == set(want), (len(got), len(want))
    for k, row in want.items():
        for c, wv in row.items():
            assert close(got[k][c], wv, 1e-9), (shape, k, c, got[k][c], wv)


def test_predicate_logic_and_integer_division(orc):
    """Kleene and / or / not over nullable booleans, comparisons of nullable columns, int floor-div / mod (divisor 0 -> null),
    literal folding (x / 4.0 == x * 0.25) -- against three-valued logic written out in numpy."""
    rng = np.random.default_rng(77)
    n = 80_000
    a = rng.integers(-50, 50, n).astype(np.int64); am = rng.random(n) < 0.8
    b = rng.integers(-5, 6, n).astype(np.int64); bm = rng.random(n) < 0.9
    x = rng.normal(size=n); xm = rng.random(n) < 0.85
    cols = {"a": (a, am), "b": (b, bm), "x": (x, xm)}
    c = pl.col
    pred = ((c("a") // c("b") > 2) | ~(c("x") / 4.0 < 0.1)) & ((c("a") % c("b") != 1) | (c("b") >= 0))
    lf = frame_like(cols).lazy().filter(pred).select(pl.len().alias("n"), c("a").sum().alias("a_sum"), (c("x") / 4.0).sum().alias("x4"), c("x").count().alias("xc"))
    got = pe.evaluate(lf.debug_program(), cols)

    def kleene_or(p, pv, q, qv):    # value, valid
        return (p & pv) | (q & qv), (p & pv) | (q & qv) | (pv & qv)

    def kleene_and(p, pv, q, qv):
        return p & q & pv & qv | False, (~p & pv) | (~q & qv) | (pv & qv)
    nz = b != 0
    bs = np.where(nz, b, 1)
    dv, dvalid = np.floor_divide(a, bs), am & bm & nz
    mv = np.mod(a, bs)
    t1, t1v = dv > 2, dvalid
    t2, t2v = ~(x * 0.25 < 0.1), xm
    l, lv = kleene_or(t1, t1v, t2, t2v)
    t3, t3v = mv != 1, dvalid
    t4, t4v = b >= 0, bm
    r_, rv = kleene_or(t3, t3v, t4, t4v)
    p, pv = kleene_and(l & lv, lv, r_ & rv, rv)
    keep = p & pv
    assert got["n"][0][0] == int(keep.sum()) and 0 < keep.sum() < n
    assert got["a_sum"][0][0] == int(a[keep & am].sum())
    assert got["xc"][0][0] == int((keep & xm).sum())
    assert math.isclose(got["x4"][0][0], float((x * 0.25)[keep & xm].sum()), rel_tol=1e-9)


# ---- random expression trees: compiler output (interpreted) vs a direct three-valued numpy evaluation of the same tree ----
class _Gen:
    """Random typed expression trees over four columns; every node carries its mirror-API expression and a numpy
    evaluation (values, valid) that follows the reference's rules: nulls propagate through arithmetic and comparisons,
    and / or are Kleene, integer + - * wrap, / is float division (col / literal = col * (1 / literal)), // and % follow
    Python sign rules with divisor 0 -> null."""

    def __init__(self, rng, cols):
        self.rng, self.cols = rng, cols

    def leaf(self, ty):
        r = self.rng
        if ty == "i":
            if r.random() < 0.7:
                name = ["a", "b"][int(r.integers(0, 2))]
                v, m = self.cols[name]
                return pl.col(name), (v, np.ones(len(v), bool) if m is None else m)
            k = int(r.integers(-7, 8))
            return pl.lit(k, dtype=pl.Int64), (np.full(self.n, k, np.int64), np.ones(self.n, bool))
        if r.random() < 0.7:
            name = ["x", "y"][int(r.integers(0, 2))]
            v, m = self.cols[name]
            return pl.col(name), (v, np.ones(len(v), bool) if m is None else m)
        k = float(r.integers(-6, 7)) / 2 + 0.25
        return pl.lit(k, dtype=pl.Float64), (np.full(self.n, k, np.float64), np.ones(self.n, bool))

    @property
    def n(self):
        return len(self.cols["a"][0])

    def num(self, ty, depth):
        r = self.rng
        if depth == 0 or r.random() < 0.25:
            return self.leaf(ty)
        with np.errstate(all="ignore"):
            if ty == "f" and r.random() < 0.2:          # int -> float through true division
                (ea, (va, ma)), (eb, (vb, mb)) = self.num("i", depth - 1), self.num("i", depth - 1)
                return ea / eb, (va.astype(np.float64) / vb.astype(np.float64), ma & mb)
            (ea, (va, ma)), (eb, (vb, mb)) = self.num(ty, depth - 1), self.num(ty, depth - 1)
            op = int(r.integers(0, 5 if ty == "i" else 4))
            if op == 0: return ea + eb, (va + vb, ma & mb)
            if op == 1: return ea - eb, (va - vb, ma & mb)
            if op == 2: return ea * eb, (va * vb, ma & mb)
            if ty == "f":
                return ea / eb, (va / vb, ma & mb)
            nz = vb != 0
            vs = np.where(nz, vb, 1)
            if op == 3: return ea // eb, (np.floor_divide(va, vs), ma & mb & nz)
            return ea % eb, (np.mod(va, vs), ma & mb & nz)

    def boolean(self, depth):
        r = self.rng
        if depth == 0 or r.random() < 0.35:
            ty = "i" if r.random() < 0.5 else "f"
            (ea, (va, ma)), (eb, (vb, mb)) = self.num(ty, 1), self.num(ty, 1)
            op = int(r.integers(0, 6))
            if ty == "f":      # total order: NaN == NaN, NaN greatest
                an, bn = np.isnan(va), np.isnan(vb)
                eq = (an & bn) | (va == vb)
                lt = ~an & (bn | (va < vb)) & ~eq
            else:
                eq, lt = va == vb, va < vb
            val = [eq, ~eq, lt, lt | eq, ~(lt | eq), ~lt][op]
            e = [ea == eb, ea != eb, ea < eb, ea <= eb, ea > eb, ea >= eb][op]
            return e, (val, ma & mb)
        if r.random() < 0.2:
            e, (v, m) = self.boolean(depth - 1)
            return ~e, (~v, m)
        (ea, (va, ma)), (eb, (vb, mb)) = self.boolean(depth - 1), self.boolean(depth - 1)
        if r.random() < 0.5:
            return ea & eb, (va & vb, (~va & ma) | (~vb & mb) | (ma & mb))
        return ea | eb, (va | vb, (va & ma) | (vb & mb) | (ma & mb))


@pytest.mark.parametrize("seed", range(40))
def test_random_expression_trees(seed):
    rng = np.random.default_rng(9000 + seed)
    n = 4000
    cols = {"a": (rng.integers(-40, 40, n).astype(np.int64), rng.random(n) < 0.85), "b": (rng.integers(-4, 5, n).astype(np.int64), None),
            "x": (np.round(rng.normal(size=n) * 8) / 4, rng.random(n) < 0.9), "y": (rng.integers(-3, 4, n).astype(np.float64) / 2, None)}
    g = _Gen(rng, cols)
    pe_, (pv, pm) = g.boolean(3)
    fe, (fv, fm) = g.num("f", 3)
    ie, (iv, im) = g.num("i", 3)
    lf = frame_like(cols).lazy().filter(pe_).select(pl.len().alias("n"), fe.sum().alias("fs"), fe.count().alias("fc"), fe.min().alias("fmin"),
                                                    ie.sum().alias("isum"), ie.max().alias("imax"), ie.mean().alias("imean"))
    try:
        prog = lf.debug_program()
    except pl.UnsupportedError as e:      # e.g. more than 16 live values / 32 ops: the engine runs such trees on the per-node path
        assert "fusable" in str(e) or "program" in str(e) or "live values" in str(e), str(e)
        return
    got = pe.evaluate(prog, cols)
    assert pe.split_matches(prog, cols)          # late-materialisation order is safe whenever the engine says so (slots are reused)
    keep = pv & pm
    assert got["n"][0][0] == int(keep.sum())
    fsel = keep & fm
    assert got["fc"][0][0] == int(fsel.sum())
    with np.errstate(all="ignore"):
        want_fs = float(fv[fsel].sum())
    assert (math.isnan(want_fs) and math.isnan(got["fs"][0][0])) or close(float(got["fs"][0][0]), want_fs, 1e-9)
    ford = fsel & ~np.isnan(fv)
    fmin_v, fmin_m = got["fmin"]
    if not fsel.any():
        assert not fmin_m[0]
    elif not ford.any():
        assert fmin_m[0] and math.isnan(fmin_v[0])
    else:
        assert fmin_m[0] and fmin_v[0] == fv[ford].min()
    isel = keep & im
    assert got["isum"][0][0] == int(iv[isel].sum())            # wrapping sums coincide with exact ones at these magnitudes
    imax_v, imax_m = got["imax"]
    assert bool(imax_m[0]) == bool(isel.any()) and (not isel.any() or imax_v[0] == iv[isel].max())
    imean_v, imean_m = got["imean"]
    assert bool(imean_m[0]) == bool(isel.any()) and (not isel.any() or close(float(imean_v[0]), float(iv[isel].mean()), 1e-9))


# ---- fused join -> group-by (TPC-H Q3 shape): three compiled scans interpreted on the CPU -------------------------------
def test_q3_pipeline_matches_oracle(orc):
    orders, li = datagen.orders_lineitem_host(30_000, seed=14, ordered=True)
    lt = datagen.logical_dtypes(pl)
    lcols = {k: (li[k], None) for k in datagen.LINEITEM_Q3_COLS}
    ocols = {k: (orders[k], None) for k in datagen.ORDERS_Q3_COLS}
    prog = Q.q3(frame_like(lcols, lt).lazy(), frame_like(ocols, lt).lazy()).debug_program()
    assert prog["kind"] == "join_group_by" and prog["build_side"] == "right" and prog["build_key"] == "o_orderkey" and prog["probe_key"] == "l_orderkey"
    assert [g["name"] for g in prog["group_keys"]] == ["l_orderkey", "o_orderdate", "o_shippriority"]
    # late materialisation: the probe scan loads l_extendedprice / l_discount only for rows that find an order
    pp = prog["probe"]
    late = [pp["ops"][i] for i in range(len(pp["ops"])) if not (pp["early_mask"] >> i) & 1]
    assert pp["any_late"] == 1 and sorted(pp["inputs"][op[2]]["name"] for op in late if op[0] == pe.OP_LOAD) == ["l_discount", "l_extendedprice"]
    assert pe.split_matches(pp, lcols)
    got = pe.evaluate_join(prog, ocols, lcols)
    want = orc.q3(li, orders, datagen.us(1995, 3, 15))
    order = np.argsort(got["l_orderkey"][0])
    assert len(order) == len(want["l_orderkey"]) > 100
    assert np.array_equal(got["l_orderkey"][0][order], want["l_orderkey"])
    assert np.array_equal(got["o_orderdate"][0][order], want["o_orderdate"]) and np.array_equal(got["o_shippriority"][0][order], want["o_shippriority"])
    assert np.allclose(got["revenue"][0][order], want["revenue"], rtol=1e-9)


def test_q3_three_tables_pipeline_matches_oracle(orc):
    """TPC-H Q3 with customer: the nested inner join customer x orders only FILTERS orders (no customer column is used above
    it, c_custkey is unique), so the compiler reduces it to a membership bitmap tested inside the build scan (OP_BITLOOKUP);
    the dictionary compare c_mktsegment == "BUILDING" is a compare of codes."""
    no = 30_000
    orders, li = datagen.orders_lineitem_host(no, seed=15, ordered=True)
    cust = datagen.customer_host(datagen.n_customers_for(no), seed=15)
    lt = datagen.logical_dtypes(pl)
    lcols = {k: (li[k], None) for k in datagen.LINEITEM_Q3_COLS}
    ocols = {k: (orders[k], None) for k in datagen.ORDERS_Q3_COLS}
    ccols = {k: (cust[k], None) for k in datagen.CUSTOMER_Q3_COLS}
    lf = Q.q3_full(frame_like(ccols, lt).lazy(), frame_like(ocols, lt).lazy(), frame_like(lcols, lt).lazy())
    ok, sid, why, dump = lf.describe_fusion()
    assert ok and dump.count("\n") == 4, (why, dump)                       # count / build / probe / semi filter / the partitioned probe's scatter
    prog = lf.debug_program()
    assert prog["kind"] == "join_group_by" and prog["build_key"] == "o_orderkey" and prog["probe_key"] == "l_orderkey"
    assert [g["name"] for g in prog["group_keys"]] == ["o_orderkey", "o_orderdate", "o_shippriority"]
    (sm,) = prog["semis"]
    assert sm["side"] == "build" and sm["filter_key"] == "c_custkey" and sm["payload_key"] == "o_custkey" and (int(sm["kmin"]), int(sm["kmax"])) == (1, len(cust["c_custkey"]))
    assert any(op[0] == pe.OP_BITLOOKUP for op in prog["build"]["ops"]) and any(op[0] == pe.OP_BITLOOKUP for op in prog["count"]["ops"])
    got = pe.evaluate_join(prog, ocols, lcols, [ccols])
    want = orc.q3_full(cust, {k: orders[k] for k in datagen.ORDERS_Q3_COLS}, {k: li[k] for k in datagen.LINEITEM_Q3_COLS}, datagen.us(1995, 3, 15), datagen.SEGMENTS.index("BUILDING"))
    order = np.argsort(got["o_orderkey"][0])
    assert len(order) == len(want["o_orderkey"]) > 100
    assert np.array_equal(got["o_orderkey"][0][order], want["o_orderkey"]) and np.array_equal(got["o_orderdate"][0][order], want["o_orderdate"])
    assert np.allclose(got["revenue"][0][order], want["revenue"], rtol=1e-9)
    # a segment that is not in the dictionary matches no customer: empty result, not an error
    none = Q.q3_full(frame_like(ccols, lt).lazy(), frame_like(ocols, lt).lazy(), frame_like(lcols, lt).lazy(), segment="NOSUCH").debug_program()
    assert len(pe.evaluate_join(none, ocols, lcols, [ccols])["o_orderkey"][0]) == 0
    # duplicate customer keys: the join would multiply rows -> the rewrite must refuse (engine: per-node path)
    dup = {"c_custkey": (np.concatenate([cust["c_custkey"], cust["c_custkey"][:50]]), None), "c_mktsegment": (np.concatenate([cust["c_mktsegment"], np.ones(50, np.uint8)]), None)}
    progd = Q.q3_full(frame_like(dup, lt).lazy(), frame_like(ocols, lt).lazy(), frame_like(lcols, lt).lazy()).debug_program()
    seg1 = cust["c_mktsegment"][:50] == 1
    assert (pe.evaluate_join(progd, ocols, lcols, [dup]) is None) == bool(seg1.any())
    # a customer column used above the join: not a pure filter -> not fused
    bad = (frame_like(ccols, lt).lazy().join(frame_like(ocols, lt).lazy(), left_on="c_custkey", right_on="o_custkey")
           .join(frame_like(lcols, lt).lazy(), left_on="o_orderkey", right_on="l_orderkey").group_by("o_orderkey", "c_mktsegment").agg(pl.len()))
    assert bad.describe_fusion()[0] is False


@pytest.mark.parametrize("build", ["right", "left"])
def test_join_group_by_pipeline_matches_numpy(build):
    """Both build sides, nullable keys on both sides, predicates on both inputs, several aggregates incl. mean / min / count of a
    nullable probe column; group keys = join key + a build-side column."""
    rng = np.random.default_rng(31 if build == "right" else 32)
    nb, npr = 3_000, 20_000
    bkey = rng.permutation(np.arange(10_000, dtype=np.int64))[:nb]; bkm = rng.random(nb) < 0.95
    battr = rng.integers(0, 5, nb).astype(np.int64); bflag = rng.integers(0, 100, nb).astype(np.int64)
    pkey = rng.integers(0, 10_000, npr).astype(np.int64); pkm = rng.random(npr) < 0.9
    pv = rng.integers(-100, 100, npr).astype(np.int64); pvm = rng.random(npr) < 0.8
    px = rng.normal(size=npr)
    bcols = {"k": (bkey, bkm), "attr": (battr, None), "flag": (bflag, None)}
    pcols = {"k": (pkey, pkm), "v": (pv, pvm), "x": (px, None)}
    Bf, Pf = frame_like(bcols).lazy().filter(pl.col("flag") < 70), frame_like(pcols).lazy().filter(pl.col("x") > -1.0)
    aggs = [pl.col("v").sum().alias("v_sum"), pl.col("v").mean().alias("v_mean"), pl.col("v").min().alias("v_min"), pl.col("v").count().alias("v_cnt"),
            (pl.col("x") * 2.0).sum().alias("x2"), pl.len().alias("n")]
    if build == "right":     # probe (longer) on the left
        lf = Pf.join(Bf, on="k").group_by("k", "attr").agg(*aggs)
    else:
        lf = Bf.join(Pf, on="k").group_by("k", "attr").agg(*aggs)
    prog = lf.debug_program()
    assert prog["build_side"] == build
    got = pe.evaluate_join(prog, bcols, pcols)
    # reference: straightforward numpy / python
    bsel = (bflag < 70) & bkm
    attr_of = {int(k): int(a) for k, a in zip(bkey[bsel], battr[bsel])}
    psel = (px > -1.0) & pkm
    acc = {}
    for i in np.nonzero(psel)[0]:
        k = int(pkey[i])
        if k in attr_of:
            acc.setdefault(k, []).append(i)
    want = {}
    for k, rows in acc.items():
        rows = np.array(rows)
        vv = pv[rows][pvm[rows]]
        want[(k, attr_of[k])] = {"v_sum": int(vv.sum()), "v_mean": float(vv.mean()) if len(vv) else None, "v_min": int(vv.min()) if len(vv) else None, "v_cnt": len(vv),
                                 "x2": float((px[rows] * 2.0).sum()), "n": len(rows)}
    g = by_key(got, ["k", "attr"])
    assert set(g) == set(want) and len(want) > 500
    for key, row in want.items():
        for c, wv in row.items():
            assert close(g[key][c], wv), (key, c, g[key][c], wv)


def _pandas_join_groupby(bcols, pcols, how, keys, bpred=None, ppred=None):
    """pandas: [filter] both frames -> merge on k -> group by `keys` (nulls are groups) -> sum(v), len"""
    pd = pytest.importorskip("pandas")
    def frame(cols, pred):
        d = {}
        for n, (v, m) in cols.items():
            a = pd.array(v, dtype="Int64")
            if m is not None:
                a[~m] = pd.NA
            d[n] = a
        f = pd.DataFrame(d)
        return f if pred is None else f[pred(f)]
    B, P = frame(bcols, bpred), frame(pcols, ppred)
    # a null key matches nothing (pandas would match NA with NA): join the valid keys, keep the null-key probe rows of a left join as unmatched
    J = P[P["k"].notna()].merge(B[B["k"].notna()], on="k", how=how)
    if how == "left":
        J = pd.concat([J, P[P["k"].isna()]], ignore_index=True)
    g = J.groupby(keys, dropna=False).agg(s=("v", "sum"), n=("v", "size")).reset_index()
    return {tuple(None if pd.isna(x) else int(x) for x in row[:len(keys)]): {"s": int(row[len(keys)]), "n": int(row[len(keys) + 1])} for row in g.itertuples(index=False)}


def test_join_pipeline_with_duplicate_build_keys_groups_by_build_row():
    """Duplicate build keys (round 5: the multi-value mode of the fused pipeline): every probe row contributes once per build row of its key; build rows of a key that
    agree on the build-side group column are ONE group.  The compiled programs through the numpy interpreter against pandas merge -> groupby."""
    bcols = {"k": (np.array([1, 2, 2, 3, 2, 7, 7], dtype=np.int64), None), "attr": (np.array([10, 20, 21, 30, 20, 70, 70], dtype=np.int64), None)}
    pcols = {"k": (np.array([2, 3, 3, 9, 1, 2, 7, 8], dtype=np.int64), None), "v": (np.arange(8, dtype=np.int64), None)}      # (the longer side probes)
    lf = frame_like(pcols).lazy().join(frame_like(bcols).lazy(), on="k").group_by("k", "attr").agg(pl.col("v").sum().alias("s"), pl.len().alias("n"))
    got = by_key(pe.evaluate_join(lf.debug_program(), bcols, pcols), ["k", "attr"])
    assert got == _pandas_join_groupby(bcols, pcols, "inner", ["k", "attr"])
    assert got[(2, 20)] == {"s": 2 * (0 + 5), "n": 4} and got[(7, 70)] == {"s": 12, "n": 2}          # two build rows (2, 20): every probe row of key 2 counts twice
    rng = np.random.default_rng(8)
    nb, npr = 4_000, 30_000
    bcols = {"k": (rng.integers(0, 1500, nb).astype(np.int64), rng.random(nb) < 0.97), "attr": (rng.integers(0, 3, nb).astype(np.int64), rng.random(nb) < 0.9),
             "flag": (rng.integers(0, 100, nb).astype(np.int64), None)}
    pcols = {"k": (rng.integers(0, 3000, npr).astype(np.int64), rng.random(npr) < 0.95), "v": (rng.integers(-50, 50, npr).astype(np.int64), None)}
    for keys in (["k", "attr"], ["k"]):
        lf = (frame_like(pcols).lazy().filter(pl.col("v") > -40).join(frame_like(bcols).lazy().filter(pl.col("flag") < 80), on="k").group_by(*keys)
              .agg(pl.col("v").sum().alias("s"), pl.len().alias("n")))
        got = by_key(pe.evaluate_join(lf.debug_program(), bcols, pcols), keys)
        want = _pandas_join_groupby(bcols, pcols, "inner", keys, bpred=lambda f: f["flag"] < 80, ppred=lambda f: f["v"] > -40)
        assert got == want and len(want) > 1000, keys


def test_left_join_pipeline_keeps_the_unmatched_rows_as_groups_of_their_own():
    """LEFT JOIN -> group_by: the matched rows as the inner join (duplicate build keys included), the rows without a partner grouped by their own key with nulls in the
    build-side group column; a null probe key matches nothing and is a group of its own; a predicate on the right table decides who MATCHES, not who survives."""
    rng = np.random.default_rng(9)
    nb, npr = 3_000, 25_000
    bcols = {"k": (rng.integers(100, 1600, nb).astype(np.int64), None), "attr": (rng.integers(0, 4, nb).astype(np.int64), None), "flag": (rng.integers(0, 100, nb).astype(np.int64), None)}
    pcols = {"k": (rng.integers(0, 2500, npr).astype(np.int64), rng.random(npr) < 0.96), "v": (rng.integers(-50, 50, npr).astype(np.int64), None)}
    for keys in (["k", "attr"], ["k"]):
        lf = (frame_like(pcols).lazy().filter(pl.col("v") > -45).join(frame_like(bcols).lazy().filter(pl.col("flag") < 60), on="k", how="left").group_by(*keys)
              .agg(pl.col("v").sum().alias("s"), pl.len().alias("n")))
        prog = lf.debug_program()
        assert prog["how"] == "left" and prog["build_side"] == "right" and "unmatched" in prog
        got = by_key(pe.evaluate_join(prog, bcols, pcols), keys)
        want = _pandas_join_groupby(bcols, pcols, "left", keys, bpred=lambda f: f["flag"] < 60, ppred=lambda f: f["v"] > -45)
        assert got == want and len(want) > 1500, keys
        assert any(k[0] is None for k in want) and (keys == ["k"] or any(k[1] is None and k[0] is not None for k in want))


@pytest.mark.parametrize("case", [c for c in __import__("tests.kat", fromlist=["kat"]).load_cases("groupby")], ids=lambda c: c["id"])
def test_reference_group_by_kats_through_the_compiler(case):
    """The reference's group_by known-answer vectors, run through the C++ compiler + the numpy interpreter (no GPU): the same
    cases tests/test_gpu_golden.py runs on the hardware."""
    from tests import kat
    cols, key_names = {}, list(case["keys"])
    cats = {}
    for name, spec in case["keys"].items():
        a, v, c = kat.column(spec, case["key_dtypes"][name])
        cols[name] = (a, v); cats[name] = c
    aggs = []
    for col, op in case["aggs"]:
        if col not in cols:
            a, v, _ = kat.column(case["values"][col], case["value_dtypes"][col])
            if a.dtype == np.float32:
                pytest.skip("f32 values aggregate on the per-node f32 path, not in the f64 fused program")
            cols[col] = (a, v)
        e = pl.col(col)
        aggs.append({"sum": e.sum, "mean": e.mean, "min": e.min, "max": e.max, "count": e.count, "len": e.len}[op]().alias(f"{col}_{op}"))
    try:
        prog = frame_like(cols).lazy().group_by(*key_names, maintain_order=case["maintain_order"]).agg(*aggs).debug_program()
    except pl.UnsupportedError:
        pytest.skip("not a fused shape (runs on the per-node path on the GPU)")
    got = by_key(pe.evaluate(prog, cols), key_names)
    exp = case["expect"]
    n = len(exp[key_names[0]])
    assert len(got) == n, (len(got), n)
    for i in range(n):
        key = []
        for k in key_names:
            v = exp[k][i]
            key.append(None if v is None else (cats[k].index(v) if cats[k] is not None else kat.scalar(v)))
        row = got[tuple(key)]
        for c, vals in exp.items():
            if c in key_names:
                continue
            g = row[c]
            assert kat.same_value(float("nan") if g == "nan" else g, vals[i], case.get("rtol", 1e-12)), (case["id"], key, c, g, vals[i])


def test_api_shorthands_lower_to_the_same_programs():
    """is_between / unary minus / GroupBy.len / GroupBy.sum are sugar over the existing IR: they must compile and evaluate like the spelled-out forms."""
    rng = np.random.default_rng(4)
    n = 5000
    cols = {"k": (rng.integers(0, 7, n).astype(np.int64), None), "a": (rng.integers(-50, 50, n).astype(np.int64), rng.random(n) < 0.9), "x": (rng.normal(size=n), None)}
    c = pl.col
    f = frame_like(cols)
    short = f.lazy().filter(c("a").is_between(-10, 20, closed="left")).select((-c("x")).sum().alias("s"), pl.len().alias("n"))
    long_ = f.lazy().filter((c("a") >= -10) & (c("a") < 20)).select((0.0 - c("x")).sum().alias("s"), pl.len().alias("n"))
    a, b = pe.evaluate(short.debug_program(), cols), pe.evaluate(long_.debug_program(), cols)
    keep = (cols["a"][0] >= -10) & (cols["a"][0] < 20) & cols["a"][1]
    assert a["n"][0][0] == b["n"][0][0] == int(keep.sum()) and close(float(a["s"][0][0]), float(-cols["x"][0][keep].sum()), 1e-9) and a["s"][0][0] == b["s"][0][0]
    with pytest.raises(ValueError, match="closed must be"):
        c("a").is_between(0, 1, closed="sideways")
    g1 = by_key(pe.evaluate(f.lazy().group_by("k").len().debug_program(), cols), ["k"])
    assert {k[0]: v["len"] for k, v in g1.items()} == {int(k): int((cols["k"][0] == k).sum()) for k in np.unique(cols["k"][0])}
    g2 = by_key(pe.evaluate(f.lazy().group_by("k").sum().debug_program(), cols), ["k"])
    for k, row in g2.items():
        m = cols["k"][0] == k[0]
        assert row["a"] == int(cols["a"][0][m & cols["a"][1]].sum()) and close(row["x"], float(cols["x"][0][m].sum()), 1e-9)
    assert list(f.lazy().group_by("k").mean()._lower()[2]) == ["k", "a", "x"]


def test_is_null_is_not_null_fill_null():
    """Null handling expressions compile to opcodes the kernels already run (compare-with-self + IFNULL) and mean what
    py-polars' is_null / is_not_null / fill_null(literal) mean."""
    rng = np.random.default_rng(8)
    n = 20_000
    a, am = rng.integers(-20, 20, n).astype(np.int64), rng.random(n) < 0.7
    x, xm = rng.normal(size=n), rng.random(n) < 0.6
    x[rng.random(n) < 0.05] = np.nan
    b, bm = rng.integers(0, 2, n).astype(bool), rng.random(n) < 0.8
    k = rng.integers(0, 5, n).astype(np.int64)
    cols = {"a": (a, am), "x": (x, xm), "b": (b, bm), "k": (k, None)}
    c = pl.col
    f = frame_like(cols)
    lf = (f.lazy().filter(c("a").is_not_null() & (c("x").is_null() | (c("x").fill_null(2.5) > 0.0)) & c("b").fill_null(True))
          .select(pl.len().alias("n"), c("a").fill_null(7).sum().alias("a7"), c("x").fill_null(-1.0).sum().alias("xf"), c("x").count().alias("xc"),
                  (c("a").fill_null(0) + 1).max().alias("amax")))
    prog = lf.debug_program()
    got = pe.evaluate(prog, cols)
    xf = np.where(xm, x, 2.5)
    with np.errstate(invalid="ignore"):
        keep = am & (~xm | (np.isnan(xf) | (xf > 0.0))) & np.where(bm, b, True)        # NaN > 0.0 in the total order
    assert got["n"][0][0] == int(keep.sum()) and 0 < keep.sum() < n
    assert got["a7"][0][0] == int(np.where(am, a, 7)[keep].sum())
    want_xf = float(np.where(xm, x, -1.0)[keep].sum())
    assert (math.isnan(want_xf) and math.isnan(got["xf"][0][0])) or close(float(got["xf"][0][0]), want_xf)
    assert got["xc"][0][0] == int((keep & xm).sum()) and got["amax"][0][0] == int((np.where(am, a, 0) + 1)[keep].max())
    # a filled column is no longer nullable: its count is the group length; is_null of a non-nullable column folds to a constant
    g = f.lazy().group_by("k").agg(c("a").fill_null(0).count().alias("cnt"), c("a").count().alias("valid"))
    by = by_key(pe.evaluate(g.debug_program(), cols), ["k"])
    gn = f.lazy().filter(c("a").is_null() & c("k").is_not_null()).group_by("k").agg(pl.len().alias("nulls"))
    byn = by_key(pe.evaluate(gn.debug_program(), cols), ["k"])
    assert pe.evaluate(f.lazy().filter(c("k").is_null()).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0] == 0
    for kv, row in by.items():
        m = k == kv[0]
        assert row["cnt"] == int(m.sum()) and row["valid"] == int((m & am).sum()) and byn[kv]["nulls"] == int((m & ~am).sum())
    with pytest.raises(TypeError):
        c("a").fill_null(None)
    with pytest.raises((TypeError, OverflowError)):
        f.lazy().select(c("a").fill_null("zero").sum())._lower()


def test_boolean_sum_and_mean_in_fused_programs():
    """sum / mean of a boolean expression (e.g. null counts per group: col.is_null().sum()) reuse the integer / float cells."""
    rng = np.random.default_rng(10)
    n = 30_000
    a, am = rng.integers(-20, 20, n).astype(np.int64), rng.random(n) < 0.7
    b, bm = rng.integers(0, 2, n).astype(bool), rng.random(n) < 0.85
    k = rng.integers(0, 6, n).astype(np.int64)
    cols = {"a": (a, am), "b": (b, bm), "k": (k, None)}
    c = pl.col
    lf = frame_like(cols).lazy().group_by("k").agg(c("a").is_null().sum().alias("nulls"), (c("a") > 3).sum().alias("gt3"), c("b").sum().alias("b_true"),
                                                    c("b").mean().alias("b_frac"), (c("a") > 3).mean().alias("gt3_frac"))
    prog = lf.debug_program()
    assert [f["out_dtype"] for f in prog["finals"]][:3] == [pe.U32, pe.U32, pe.U32]
    got = by_key(pe.evaluate(prog, cols), ["k"])
    for kv, row in got.items():
        m = k == kv[0]
        assert row["nulls"] == int((m & ~am).sum()) and row["gt3"] == int((m & am & (a > 3)).sum()) and row["b_true"] == int((m & bm & b).sum())
        assert close(row["b_frac"], float(b[m & bm].mean())) and close(row["gt3_frac"], float((a > 3)[m & am].mean()))
    with pytest.raises(pl.UnsupportedError, match="min / max of a boolean"):
        frame_like(cols).lazy().select(c("b").max()).debug_program()


@pytest.mark.parametrize("seed", range(30))
def test_random_group_by_shapes(orc, seed):
    """Random key sets (dtypes, ranges up to the packing limits, nullability, 1-3 keys) and aggregates through lower_keys / lower_agg:
    packed, wide and raw keys must all decode back to the oracle's groups."""
    rng = np.random.default_rng(7000 + seed)
    n = 8000
    nk = int(rng.integers(1, 4))
    keys = {}
    for j in range(nk):
        kind = int(rng.integers(0, 7))
        card = int(rng.integers(2, 40))
        if kind == 0: v = rng.integers(0, 2, n).astype(bool)
        elif kind == 1: v = rng.integers(-card, card, n).astype(np.int8)
        elif kind == 2: v = (rng.integers(0, card, n) * 1000 - 2 ** 31 + 5).astype(np.int32)                       # large negative offset
        elif kind == 3: v = rng.integers(2 ** 62 - card, 2 ** 62, n).astype(np.int64)                             # near the top of the range
        elif kind == 4: v = (rng.integers(0, card, n).astype(np.uint64) * np.uint64(2 ** 58)).astype(np.uint64)   # UInt64: never packed
        elif kind == 5: v = rng.choice([0.0, -0.0, 1.25, -3.5, np.nan, np.inf], n)
        else: v = rng.integers(-2 ** 40, 2 ** 40, card).astype(np.int64)[rng.integers(0, card, n)]                # sparse wide range
        m = None if rng.random() < 0.5 else rng.random(n) < 0.85
        keys[f"k{j}"] = (v, m)
    val = rng.integers(-500, 500, n).astype(np.int64); vm = rng.random(n) < 0.9
    x = rng.normal(size=n); xm = None
    cols = dict(keys); cols.update({"v": (val, vm), "x": (x, xm)})
    c = pl.col
    lf = frame_like(cols, stats=bool(rng.integers(0, 2)) or True).lazy().group_by(*keys).agg(c("v").sum().alias("s"), c("v").min().alias("mn"), c("v").count().alias("cnt"),
                                                                                                pl.len().alias("n"), c("x").mean().alias("xm"), c("x").max().alias("xx"))
    try:
        prog = lf.debug_program()
    except pl.UnsupportedError as e:
        assert "group keys" in str(e) or "fusable" in str(e), str(e)
        return
    got = by_key(pe.evaluate(prog, cols), list(keys))
    A = orc
    spec = [("s", A.AGG_SUM, val, vm), ("mn", A.AGG_MIN, val, vm), ("cnt", A.AGG_COUNT, val, vm), ("n", A.AGG_LEN, None, None), ("xm", A.AGG_MEAN, x, xm), ("xx", A.AGG_MAX, x, xm)]
    kv = [np.ascontiguousarray(a.astype(np.uint8) if a.dtype == np.bool_ else a) for a, _ in keys.values()]
    r = orc.q_groupby(kv, [m for _, m in keys.values()], spec)
    named = {name: r[f"key_{i}"] for i, name in enumerate(keys)}
    for name, (a, _) in keys.items():
        if a.dtype == np.bool_:
            named[name] = (named[name][0].astype(bool), named[name][1])
    named.update({s[0]: r[s[0]] for s in spec})
    want = by_key(named, list(keys))
    assert set(got) == set(want), (seed, prog["key_plan"]["packed"], prog["key_plan"]["wide"], len(got), len(want))
    for key, row in want.items():
        for col, wv in row.items():
            assert close(got[key][col], wv), (seed, key, col, got[key][col], wv)


@pytest.mark.parametrize("unit", ["ms", "ns"])
def test_datetime_literal_against_columns_of_other_time_units(unit):
    """A python datetime (Datetime[us]) compared with a Datetime[ms] / Datetime[ns] column.  The reference coerces to the coarser unit
    and casts the finer side by floor division (utils/mod.rs:804-811, logical/datetime.rs:59-66); the lowering rewrites the comparison
    into the column's unit instead of casting.  Checked for all six operators, on both sides of the epoch, with nulls, through the
    compiled program."""
    import datetime as dtm
    rng = np.random.default_rng(7)
    n = 4000
    per_us = {"ms": None, "ns": 1000}[unit]
    lits = [dtm.datetime(1970, 1, 1), dtm.datetime(1970, 1, 1, 0, 0, 0, 1), dtm.datetime(1969, 12, 31, 23, 59, 59, 999_999), dtm.datetime(1995, 3, 15, 12, 0, 0, 123_456),
            dtm.datetime(1960, 6, 1, 1, 2, 3, 999)]
    us_of = lambda d: ((d - dtm.datetime(1970, 1, 1)) // dtm.timedelta(microseconds=1))
    centres = np.array([us_of(d) for d in lits], np.int64)
    if unit == "ns":
        t = (centres[rng.integers(0, len(lits), n)] * 1000 + rng.integers(-2500, 2500, n)).astype(np.int64)          # a few microseconds around every literal
    else:
        t = (centres[rng.integers(0, len(lits), n)] // 1000 + rng.integers(-3, 4, n)).astype(np.int64)
    valid = rng.random(n) > 0.1
    cols = {"t": (t, valid)}
    df = frame_like(cols, {"t": pl.Datetime(unit)})
    assert df.schema["t"].time_unit == unit
    npop = {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal, "==": np.equal, "!=": np.not_equal}
    for d in lits:
        L = us_of(d)
        for name, fn in npop.items():
            c = pl.col("t")
            e = {"<": c < d, "<=": c <= d, ">": c > d, ">=": c >= d, "==": c.eq(d), "!=": c.ne(d)}[name]
            if unit == "ns":
                want = fn(t // per_us, L)                     # column floor-divided into microseconds
            else:
                want = fn(t, L // 1000)                       # literal floor-divided into milliseconds
            got = pe.evaluate(df.lazy().filter(e).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0]
            assert got == int((want & valid).sum()), (unit, name, d)
            flipped = {"<": d > c, "<=": d >= c, ">": d < c, ">=": d <= c, "==": c.eq(d), "!=": c.ne(d)}[name]      # literal on the left
            assert pe.evaluate(df.lazy().filter(flipped).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0] == got
    with pytest.raises(TypeError):                            # two columns of different units would need the cast kernel
        frame_like({"a": (t, None), "b": (t, None)}, {"a": pl.Datetime("ns"), "b": pl.Datetime("us")}).lazy().filter(pl.col("a") < pl.col("b")).debug_program()
    # bounds beyond i64 in the column's unit: constant comparisons, nulls stay out
    if unit == "ns":
        far = dtm.datetime(9999, 1, 1)
        assert pe.evaluate(df.lazy().filter(pl.col("t") < far).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0] == int(valid.sum())
        assert pe.evaluate(df.lazy().filter(pl.col("t") >= far).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0] == 0


def test_is_between_all_closures():
    rng = np.random.default_rng(11)
    n = 5000
    x = rng.integers(-20, 20, n).astype(np.int64)
    valid = rng.random(n) > 0.15
    cols = {"x": (x, valid)}
    df = frame_like(cols)
    for closed, fn in (("both", lambda v: (v >= -3) & (v <= 7)), ("left", lambda v: (v >= -3) & (v < 7)), ("right", lambda v: (v > -3) & (v <= 7)), ("none", lambda v: (v > -3) & (v < 7))):
        got = pe.evaluate(df.lazy().filter(pl.col("x").is_between(-3, 7, closed=closed)).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0]
        assert got == int((fn(x) & valid).sum()), closed
    import datetime as dtm
    t = (rng.integers(0, 3000, n) * 86_400_000_000).astype(np.int64)
    tdf = frame_like({"t": (t, None)}, {"t": pl.Datetime})
    lo, hi = dtm.datetime(1972, 1, 1), dtm.datetime(1975, 6, 1)
    us = lambda d: (d - dtm.datetime(1970, 1, 1)) // dtm.timedelta(microseconds=1)
    got = pe.evaluate(tdf.lazy().filter(pl.col("t").is_between(lo, hi)).select(pl.len().alias("n")).debug_program(), {"t": (t, None)})["n"][0][0]
    assert got == int(((t >= us(lo)) & (t <= us(hi))).sum())
    with pytest.raises(ValueError):
        pl.col("x").is_between(0, 1, closed="sideways")


def test_is_in_a_literal_list():
    rng = np.random.default_rng(12)
    n = 4000
    x = rng.integers(0, 12, n).astype(np.int64)
    valid = rng.random(n) > 0.2
    codes = rng.integers(0, 4, n).astype(np.uint32)
    cols = {"x": (x, valid), "s": (codes, None)}
    df = frame_like(cols, {"s": pl.Categorical(["AIR", "MAIL", "RAIL", "SHIP"], pl.UInt32)})
    got = pe.evaluate(df.lazy().filter(pl.col("x").is_in([3, 5, 11])).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0]
    assert got == int((np.isin(x, [3, 5, 11]) & valid).sum())
    got = pe.evaluate(df.lazy().filter(pl.col("s").is_in(["MAIL", "SHIP", "TRUCK"])).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0]
    assert got == int(np.isin(codes, [1, 3]).sum())                      # a string absent from the dictionary matches nothing
    # the empty list matches nothing (null stays null -> dropped by the filter); a null in the list equals nothing (nulls_equal=False)
    got = pe.evaluate(df.lazy().filter(pl.col("x").is_in([])).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0]
    assert got == 0
    got = pe.evaluate(df.lazy().filter(pl.col("x").is_in([3, None])).select(pl.len().alias("n")).debug_program(), cols)["n"][0][0]
    assert got == int(((x == 3) & valid).sum())


def test_q6_program_matches_numpy():
    li = datagen.lineitem_host(200_000, seed=5)
    names = ["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"]
    cols = {k: (li[k], None) for k in names}
    lf = Q.q6(frame_like(cols, datagen.logical_dtypes(pl)).lazy())
    fusable, _, why, _ = lf.describe_fusion()
    assert fusable, why
    got = pe.evaluate(lf.debug_program(), cols)["revenue"][0][0]
    m = (li["l_shipdate"] >= datagen.us(1994, 1, 1)) & (li["l_shipdate"] < datagen.us(1995, 1, 1)) & (li["l_discount"] >= 0.05) & (li["l_discount"] <= 0.07) & (li["l_quantity"] < 24)
    want = float((li["l_extendedprice"][m] * li["l_discount"][m]).sum())
    assert m.sum() > 1000 and math.isclose(got, want, rel_tol=1e-9)
