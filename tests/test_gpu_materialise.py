"""Operators that RETURN FRAMES on the fused machinery (round 6): filter -> frame (engine.cpp fused_filter_frame: fused_sinks.hpp BallotSink -> kernels_filter.hip compact_by_ballots) and the
materialising join (engine.cpp fused_join_frame: fused build scan -> candidates (partitioned LDS probe, or the one-pass row-id filter) -> join::join_pairs -> multi-column
gathers).  Reference: crates/polars-compute/src/filter/mod.rs:18-110 (order-preserving, null predicate = false), crates/polars-ops/src/frame/join/mod.rs:564-652
(_inner_join_from_series / _left_join_from_series: pairs, then gathers; maintain_order = none: the row order is unspecified, so joins are compared as sorted row sets),
hash_join/single_keys_inner.rs:11-149, single_keys_left.rs:106-195, gather/primitive.rs:9-78.  Ground truth: numpy for the filter (bit-exact, in order), the CPU oracle's
join (orc.join) + numpy gathers for the joins."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HASH_MULT = np.uint64(0x9E3779B97F4A7C15)


def _frame(pl, rng, n):
    a = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    xv = rng.random(n) > 0.1
    x = rng.normal(size=n)
    i32 = rng.integers(-1000, 1000, n).astype(np.int32)
    u8 = rng.integers(0, 255, n).astype(np.uint8)
    i16 = rng.integers(-3000, 3000, n).astype(np.int16)
    b = rng.random(n) > 0.4
    bv = rng.random(n) > 0.2
    df = pl.DataFrame([pl.Series("a", a), pl.Series("x", x, validity=xv), pl.Series("i", i32), pl.Series("u", u8), pl.Series("h", i16), pl.Series("b", b, validity=bv)])
    return df, dict(a=a, x=x, xv=xv, i=i32, u=u8, h=i16, b=b, bv=bv)


def _check_filter(out, h, keep):
    assert out.height == int(keep.sum())
    assert np.array_equal(out["a"].to_numpy(), h["a"][keep])
    assert np.array_equal(out["i"].to_numpy(), h["i"][keep])
    assert np.array_equal(out["u"].to_numpy(), h["u"][keep])
    assert np.array_equal(out["h"].to_numpy(), h["h"][keep])
    x, xv = out["x"]._download()
    want_v = h["xv"][keep]
    assert np.array_equal(xv if xv is not None else np.ones(len(x), bool), want_v)
    assert np.array_equal(x[want_v], h["x"][keep][want_v])
    b, bv = out["b"]._download()
    want_bv = h["bv"][keep]
    assert np.array_equal(bv if bv is not None else np.ones(len(b), bool), want_bv)
    assert np.array_equal(np.asarray(b, bool)[want_bv], h["b"][keep][want_bv])


@pytest.mark.parametrize("n", [0, 1, 63, 127, 128, 2047, 2048, 2049, 100_003, (1 << 22) + 77_777])
def test_filter_to_frame_one_pass_matches_numpy(pl, n):
    """every tile boundary (wave tile 128, quad 512 / 1024, bitmap tile 2048), the generic interpreter below 2^22 rows and the run-time compiled kernel above"""
    rng = np.random.default_rng(600 + n % 1000)
    df, h = _frame(pl, rng, n)
    c = pl.col
    out = df.lazy().filter((c("a") > 0) & (c("i") < 500)).collect()
    if n:
        assert "FusedFilter{" in pl.last_plan(), pl.last_plan()
    _check_filter(out, h, (h["a"] > 0) & (h["i"] < 500))
    ref = df.lazy().filter((c("a") > 0) & (c("i") < 500)).collect(no_fusion=True)
    assert "FusedFilter{" not in pl.last_plan()
    _check_filter(ref, h, (h["a"] > 0) & (h["i"] < 500))


@pytest.mark.parametrize("sel", ["none", "all", "sparse", "dense", "null_pred", "stacked"])
def test_filter_to_frame_selectivities_and_null_predicates(pl, sel):
    rng = np.random.default_rng(77)
    n = 300_011
    df, h = _frame(pl, rng, n)
    c = pl.col
    if sel == "none":
        q, keep = df.lazy().filter(c("a") > (1 << 41)), np.zeros(n, bool)
    elif sel == "all":
        q, keep = df.lazy().filter(c("a") > -(1 << 41)), np.ones(n, bool)
    elif sel == "sparse":
        q, keep = df.lazy().filter((c("i") == 7) & (c("u") < 100)), (h["i"] == 7) & (h["u"] < 100)
    elif sel == "dense":
        q, keep = df.lazy().filter(c("i") != 7), h["i"] != 7
    elif sel == "null_pred":        # a null predicate value drops the row (filter/mod.rs:21-27)
        q, keep = df.lazy().filter(c("x") > 0.0), h["xv"] & (h["x"] > 0.0)
    else:                           # Filter over Filter: one conjunction, one pass
        q, keep = df.lazy().filter(c("a") > 0).filter(c("x").is_not_null()).filter(c("h") < 0), (h["a"] > 0) & h["xv"] & (h["h"] < 0)
    out = q.collect()
    assert "FusedFilter{" in pl.last_plan(), pl.last_plan()
    _check_filter(out, h, keep)


# ---------------------------------------------------------------------------------------------------------------- joins
def _join_inputs(rng, n_probe, n_build, dup, hashed, null_keys=True):
    n_keys = max(n_build // (4 if dup else 1), 1)
    if dup:
        bid = rng.integers(0, n_keys, n_build).astype(np.int64)            # ~4 rows per key, some keys absent
    else:
        bid = rng.permutation(n_keys * 2)[:n_build].astype(np.int64)         # unique keys, half of the id range
    pid = rng.integers(0, n_keys * 2, n_probe).astype(np.int64)
    enc = (lambda v: (v.astype(np.uint64) * HASH_MULT).astype(np.int64)) if hashed else (lambda v: v * 3 + 11)
    bk, pk = enc(bid), enc(pid)
    bv = (rng.random(n_build) > 0.02) if null_keys else None
    pv = (rng.random(n_probe) > 0.03) if null_keys else None
    return dict(pk=pk, pv=pv, px=rng.integers(0, 1000, n_probe).astype(np.int64), pw=rng.normal(size=n_probe), pd=rng.integers(0, 100, n_probe).astype(np.int32),
                bk=bk, bv=bv, by=rng.integers(0, 1 << 30, n_build).astype(np.int64), bz=rng.integers(0, 50, n_build).astype(np.int32))


def _frames(pl, h):
    P = pl.DataFrame([pl.Series("k", h["pk"], validity=h["pv"]) if h["pv"] is not None else pl.Series("k", h["pk"]), pl.Series("x", h["px"]), pl.Series("w", h["pw"]), pl.Series("d", h["pd"])])
    B = pl.DataFrame([pl.Series("k", h["bk"], validity=h["bv"]) if h["bv"] is not None else pl.Series("k", h["bk"]), pl.Series("y", h["by"]), pl.Series("z", h["bz"])])
    return P, B


def _expected_join(orc, h, how, pmask=None, bmask=None):
    """(x, w, d, y, z, y_valid) rows of P[pmask] JOIN B[bmask] through the oracle's pairs + numpy gathers"""
    psel = np.nonzero(pmask)[0] if pmask is not None else np.arange(len(h["pk"]))
    bsel = np.nonzero(bmask)[0] if bmask is not None else np.arange(len(h["bk"]))
    pv = h["pv"][psel] if h["pv"] is not None else None
    bv = h["bv"][bsel] if h["bv"] is not None else None
    li, ri, rvalid = orc.join(1 if how == "left" else 0, h["pk"][psel], pv, h["bk"][bsel], bv)
    pi = psel[li]
    if rvalid is None:
        rvalid = np.ones(len(li), bool)
    bi = bsel[np.where(rvalid, ri, 0)]
    kvalid = h["pv"][pi] if h["pv"] is not None else np.ones(len(pi), bool)
    return dict(k=np.where(kvalid, h["pk"][pi], 0), kvalid=kvalid, x=h["px"][pi], w=h["pw"][pi], d=h["pd"][pi], y=np.where(rvalid, h["by"][bi], 0), z=np.where(rvalid, h["bz"][bi], 0), rvalid=rvalid)


def _got_join(out):
    k, kv = out["k"]._download()
    y, yv = out["y"]._download()
    z, zv = out["z"]._download()
    n = out.height
    kv = kv if kv is not None else np.ones(n, bool)
    yv = yv if yv is not None else np.ones(n, bool)
    zv = zv if zv is not None else np.ones(n, bool)
    assert np.array_equal(yv, zv)
    return dict(k=np.where(kv, k, 0), kvalid=kv, x=out["x"].to_numpy(), w=out["w"].to_numpy(), d=out["d"].to_numpy(), y=np.where(yv, y, 0), z=np.where(zv, z, 0), rvalid=yv)


def _same_rows(got, want):
    assert len(got["x"]) == len(want["x"]), (len(got["x"]), len(want["x"]))
    def order(t):
        return np.lexsort((t["w"].view(np.int64), t["z"], t["y"], t["rvalid"], t["d"], t["x"], t["k"], t["kvalid"]))
    og, ow = order(got), order(want)
    for c in ("k", "kvalid", "x", "d", "y", "z", "rvalid"):
        assert np.array_equal(got[c][og], want[c][ow]), c
    assert np.array_equal(got["w"][og].view(np.int64), want["w"][ow].view(np.int64))


@pytest.mark.parametrize("dup,hashed,how", [(True, True, "inner"), (False, False, "left")])
def test_join_to_frame_matches_the_oracle_at_2_pow_24_rows(pl, orc, monkeypatch, dup, hashed, how):
    """>= 2^24 probe rows: the default plan takes the fused join -> frame path; unique and duplicate build keys, inner and left, null keys on both sides"""
    rng = np.random.default_rng(900 + 4 * dup + 2 * hashed + (how == "left"))
    n_probe, n_build = (1 << 24) + 12_345, 1_500_000
    h = _join_inputs(rng, n_probe, n_build, dup, hashed)
    P, B = _frames(pl, h)
    out = P.lazy().join(B.lazy(), on="k", how=how).collect()
    plan = pl.last_plan()
    assert "FusedJoinFrame{" in plan and ("multi-value" in plan) == dup, plan
    _same_rows(_got_join(out), _expected_join(orc, h, how))


@pytest.mark.parametrize("dup", [False, True])
@pytest.mark.parametrize("how", ["inner", "left"])
@pytest.mark.parametrize("partitioned", [False, True])
def test_join_to_frame_with_predicates_on_both_sides(pl, orc, monkeypatch, dup, how, partitioned):
    """Filters below the join are fused into the build scan and the candidate selection; forced at a size the oracle finishes quickly.  partitioned: the probe side
    goes through the radix-partitioned LDS probe (inner joins), otherwise through the one-pass row-id filter."""
    monkeypatch.setenv("PLX_JOIN_MATERIALISE", "2")
    if partitioned:
        monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2")
    rng = np.random.default_rng(950 + 4 * dup + 2 * (how == "left") + partitioned)
    n_probe, n_build = (1 << 22) + 999, 300_000          # (from 2^22 rows the probe side's scatter is compiled at run time: below, the partitioned probe declines)
    h = _join_inputs(rng, n_probe, n_build, dup, hashed=True)
    P, B = _frames(pl, h)
    c = pl.col
    q = P.lazy().filter((c("d") < 60) & (c("x") > 100)).join(B.lazy().filter(c("z") != 3), on="k", how=how)
    out = q.collect()
    plan = pl.last_plan()
    assert "FusedJoinFrame{" in plan, plan
    if partitioned and how == "inner":
        assert "partitioned_hash_probe(" in plan, plan
    want = _expected_join(orc, h, how, pmask=(h["pd"] < 60) & (h["px"] > 100), bmask=h["bz"] != 3)
    _same_rows(_got_join(out), want)
    # the per-node path (first-generation join kernels) agrees
    ref = q.collect(no_fusion=True)
    assert "FusedJoinFrame{" not in pl.last_plan()
    _same_rows(_got_join(ref), want)


@pytest.mark.parametrize("how,partitioned,dup", [("inner", True, False), ("inner", False, False), ("left", False, False), ("inner", True, True)])
def test_join_to_frame_on_a_table_filled_from_lds(pl, orc, monkeypatch, how, partitioned, dup):
    """PLX_JOIN_PART_BUILD=2: the build side's pairs are binned into the table's windows and every window is filled from an LDS image (k::partitioned_join_build); probe
    sequences wrap inside the windows.  Unique hashed keys with the key equal to the table's EMPTY pattern (-1) and the extreme Int64 values among them, null keys on
    both sides, a predicate on the build side.  Duplicate build keys are noticed by the fill and the table is built again the plain way in multi-value mode."""
    monkeypatch.setenv("PLX_JOIN_MATERIALISE", "2")
    monkeypatch.setenv("PLX_JOIN_PART_BUILD", "2")
    monkeypatch.setenv("PLX_PROBE_PARTITIONED", "2" if partitioned else "0")
    rng = np.random.default_rng(970 + 2 * (how == "left") + partitioned + 4 * dup)
    n_probe, n_build = (1 << 22) + 999, 1_300_000
    h = _join_inputs(rng, n_probe, n_build, dup, hashed=True)
    if not dup:
        edge = np.array([-1, np.iinfo(np.int64).min, np.iinfo(np.int64).max], np.int64)
        assert not np.isin(edge, h["bk"]).any()
        h["bk"][:3] = edge; h["bv"][:3] = True; h["bz"][:3] = 1
        h["pk"][:6] = np.tile(edge, 2); h["pv"][:6] = True
    P, B = _frames(pl, h)
    c = pl.col
    q = P.lazy().join(B.lazy().filter(c("z") != 3), on="k", how=how)
    out = q.collect()
    plan = pl.last_plan()
    assert "FusedJoinFrame{" in plan and ("partitioned build(" in plan) == (not dup) and ("multi-value" in plan) == dup, plan
    if partitioned:
        assert "partitioned_hash_probe(" in plan, plan
    want = _expected_join(orc, h, how, bmask=h["bz"] != 3)
    got = _got_join(out)
    _same_rows(got, want)
    if not dup:
        assert {-1, int(np.iinfo(np.int64).min), int(np.iinfo(np.int64).max)} <= set(got["k"][got["rvalid"]].tolist())


def test_join_to_frame_select_gathers_only_the_named_columns(pl, orc, monkeypatch):
    monkeypatch.setenv("PLX_JOIN_MATERIALISE", "2")
    rng = np.random.default_rng(5)
    h = _join_inputs(rng, 500_000, 50_000, dup=False, hashed=False, null_keys=False)
    P, B = _frames(pl, h)
    c = pl.col
    out = P.lazy().filter(c("d") < 50).join(B.lazy(), on="k").select("k", "y", (c("x") * 2).alias("x2")).collect()
    plan = pl.last_plan()
    assert "FusedJoinFrame{" in plan and "gather x3" in plan, plan
    assert out.columns == ["k", "y", "x2"]
    want = _expected_join(orc, h, "inner", pmask=h["pd"] < 50)
    got = np.stack([out["k"].to_numpy(), out["y"].to_numpy(), out["x2"].to_numpy()])
    exp = np.stack([want["k"], want["y"], want["x"] * 2])
    assert np.array_equal(got[:, np.lexsort(got)], exp[:, np.lexsort(exp)])


def test_join_to_frame_edge_cases(pl, orc, monkeypatch):
    """empty sides, a build side the predicate empties, name clashes (suffix), the left table as the build side"""
    monkeypatch.setenv("PLX_JOIN_MATERIALISE", "2")
    c = pl.col
    P = pl.DataFrame([pl.Series("k", np.array([1, 2, 3, 4, 5, 2], np.int64)), pl.Series("v", np.array([10, 20, 30, 40, 50, 60], np.int64))])
    B = pl.DataFrame([pl.Series("k", np.array([2, 4, 6], np.int64)), pl.Series("v", np.array([200, 400, 600], np.int64))])
    out = P.lazy().join(B.lazy(), on="k").collect()
    assert "FusedJoinFrame{" in pl.last_plan()
    assert out.columns == ["k", "v", "v_right"]
    assert sorted(zip(out["k"].to_list(), out["v"].to_list(), out["v_right"].to_list())) == [(2, 20, 200), (2, 60, 200), (4, 40, 400)]
    # the shorter LEFT table becomes the build side: left columns are gathered at the build index
    out = B.lazy().join(P.lazy(), on="k").collect()
    assert "build=left" in pl.last_plan(), pl.last_plan()
    assert sorted(zip(out["k"].to_list(), out["v"].to_list(), out["v_right"].to_list())) == [(2, 200, 20), (2, 200, 60), (4, 400, 40)]
    out = P.lazy().join(B.lazy().filter(c("v") > 1000), on="k").collect()
    assert out.height == 0 and out.columns == ["k", "v", "v_right"]
    out = P.lazy().join(B.lazy().filter(c("v") > 1000), on="k", how="left").collect()
    assert out.height == 6 and out["v_right"].to_list() == [None] * 6 and sorted(out["v"].to_list()) == [10, 20, 30, 40, 50, 60]
    empty = pl.DataFrame([pl.Series("k", np.zeros(0, np.int64)), pl.Series("v", np.zeros(0, np.int64))])
    assert P.lazy().join(empty.lazy(), on="k").collect().height == 0
    assert empty.lazy().join(P.lazy(), on="k").collect().height == 0
    assert P.lazy().join(empty.lazy(), on="k", how="left").collect().height == 6


@pytest.mark.parametrize("how", ["semi", "anti"])
@pytest.mark.parametrize("hashed", [False, True])
def test_semi_anti_join_is_a_filter_against_a_membership_bitmap(pl, orc, how, hashed):
    """Join(semi | anti) over filtered inputs: the right side becomes a membership bitmap over its key range (its predicate fused into the build scan), the left side is
    FILTERED by its own predicates AND the bitmap test in one predicate program -- left columns, left order, no pairs (single_keys_semi_anti.rs).  Null keys on both
    sides (a null key is nobody's member: semi drops it, anti keeps it), duplicate right keys, a sparse and a dense result; hashed keys have no range a bitmap could
    cover and take the per-node join.  Checked against the oracle's semi / anti join and the per-node path."""
    rng = np.random.default_rng(990 + 2 * hashed + (how == "anti"))
    n_probe, n_build = (1 << 22) + 999, 300_000
    h = _join_inputs(rng, n_probe, n_build, dup=True, hashed=hashed)
    P, B = _frames(pl, h)
    c = pl.col
    for ppred, pmask, bpred, bmask in ((c("d") < 60, h["pd"] < 60, c("z") != 3, h["bz"] != 3), (None, np.ones(n_probe, bool), c("z") == 7, h["bz"] == 7)):
        lf = P.lazy().filter(ppred) if ppred is not None else P.lazy()
        q = lf.join(B.lazy().filter(bpred), on="k", how=how)
        out = q.collect()
        plan = pl.last_plan()
        assert ("FusedSemiAntiJoin{" in plan) == (not hashed), plan
        if hashed:
            assert "semi / anti join not fused: right key range too wide" in plan, plan
        psel, bsel = np.nonzero(pmask)[0], np.nonzero(bmask)[0]
        idx = orc.semi_anti_join(orc.JOIN_SEMI if how == "semi" else orc.JOIN_ANTI, h["pk"][psel], h["pv"][psel], h["bk"][bsel], h["bv"][bsel])
        rows = psel[idx]
        assert out.columns == ["k", "x", "w", "d"] and out.height == len(rows)
        k, kv = out["k"]._download()
        kv = kv if kv is not None else np.ones(len(rows), bool)
        assert np.array_equal(kv, h["pv"][rows]) and np.array_equal(np.where(kv, k, 0), np.where(h["pv"][rows], h["pk"][rows], 0))
        assert np.array_equal(out["x"].to_numpy(), h["px"][rows]) and np.array_equal(out["d"].to_numpy(), h["pd"][rows])
        assert np.array_equal(out["w"].to_numpy().view(np.int64), h["pw"][rows].view(np.int64))
        ref = q.collect(no_fusion=True)
        assert "FusedSemiAntiJoin{" not in pl.last_plan() and ref.height == out.height and np.array_equal(ref["x"].to_numpy(), out["x"].to_numpy())


# ------------------------------------------------------------------------------------------- join -> group-by, pair form
@pytest.mark.parametrize("dup", [False, True])
@pytest.mark.parametrize("shape", ["agg_reads_build", "key_is_build_column", "key_is_probe_column"])
def test_join_group_by_beyond_the_in_place_form(pl, monkeypatch, dup, shape):
    """Round-5 review, missing 2: group-bys over a join whose aggregates read BUILD-side columns, or whose keys do not contain the join key, used to drop to the
    per-node join.  They take the pair form now (fused join -> frame restricted to the referenced columns, then the fused group-by).  Ground truth: pandas."""
    pd = pytest.importorskip("pandas")
    monkeypatch.setenv("PLX_JOIN_MATERIALISE", "2")
    rng = np.random.default_rng(1200 + dup)
    n_probe, n_build = (1 << 22) + 4321, 300_000
    h = _join_inputs(rng, n_probe, n_build, dup, hashed=True, null_keys=False)
    h["bc"] = rng.random(n_build) * 10.0                         # a build-side Float64 payload
    P, B = _frames(pl, h)
    B = pl.DataFrame([B["k"], B["y"], B["z"], pl.Series("c", h["bc"])])
    c = pl.col
    j = P.lazy().filter(c("d") < 70).join(B.lazy().filter(c("z") != 3), on="k")
    Pd = pd.DataFrame({"k": h["pk"], "x": h["px"], "w": h["pw"], "d": h["pd"]})
    Bd = pd.DataFrame({"k": h["bk"], "y": h["by"], "z": h["bz"], "c": h["bc"]})
    J = Pd[Pd.d < 70].merge(Bd[Bd.z != 3], on="k")
    if shape == "agg_reads_build":
        out = j.group_by("k").agg((c("w") * c("c")).sum().alias("s"), c("y").max().alias("m"), pl.len().alias("n")).collect()
        keys = ["k"]
        exp = J.assign(s=J.w * J.c).groupby("k").agg(s=("s", "sum"), m=("y", "max"), n=("x", "size")).reset_index()
    elif shape == "key_is_build_column":
        out = j.group_by("z").agg(c("x").sum().alias("s"), c("c").sum().alias("m"), pl.len().alias("n")).collect()
        keys = ["z"]
        exp = J.groupby("z").agg(s=("x", "sum"), m=("c", "sum"), n=("x", "size")).reset_index()
    else:
        out = j.group_by("d", "z").agg(c("x").sum().alias("s"), c("c").sum().alias("m"), pl.len().alias("n")).collect()
        keys = ["d", "z"]
        exp = J.groupby(["d", "z"]).agg(s=("x", "sum"), m=("c", "sum"), n=("x", "size")).reset_index()
    plan = pl.last_plan()
    assert "FusedJoinGroupBy{pair form" in plan and "FusedJoinFrame{" in plan, plan
    got = pd.DataFrame({k: out[k].to_numpy() for k in keys} | {"s": out["s"].to_numpy(), "m": out["m"].to_numpy(), "n": out["n"].to_numpy().astype(np.int64)})
    got = got.sort_values(keys).reset_index(drop=True)
    exp = exp.sort_values(keys).reset_index(drop=True)
    assert len(got) == len(exp)
    for k in keys:
        assert np.array_equal(got[k].to_numpy().astype(np.int64), exp[k].to_numpy().astype(np.int64))
    assert np.array_equal(got["n"].to_numpy(), exp["n"].to_numpy().astype(np.int64))
    for col in ("s", "m"):
        if np.issubdtype(exp[col].dtype, np.integer):
            assert np.array_equal(got[col].to_numpy().astype(np.int64), exp[col].to_numpy())
        else:
            assert np.allclose(got[col].to_numpy(), exp[col].to_numpy(), rtol=1e-6, atol=1e-9)       # float aggregates: 1e-6 relative (north_star)
