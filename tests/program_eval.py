"""Row-level interpreter of the engine's compiled pipelines (plx_debug_program_json), in numpy.

Test infrastructure: the C++ expression compiler (polars_amd/csrc/engine.cpp, class Compiler + lower_keys + lower_agg)
turns `[Filter]* -> Select | GroupBy` plans into a register program, aggregate cells, a key packing and a finalisation
recipe; the GPU kernels execute exactly that (fused_device.hpp exec_op / agg_row_value, kernels_fused.hip finalize).
This module restates those device semantics over whole numpy columns, so the CPU tests can run what the compiler
emitted against the oracle without a GPU.  Opcodes / kinds mirror polars_amd/csrc/fused.hpp.
"""
import numpy as np

(OP_NOP, OP_LOAD, OP_CONST, OP_ADD_F, OP_SUB_F, OP_MUL_F, OP_DIV_F, OP_ADD_I, OP_SUB_I, OP_MUL_I, OP_I2F, OP_U2F, OP_CMP_I, OP_CMP_U, OP_CMP_F,
 OP_AND, OP_OR, OP_XOR, OP_NOT, OP_IFNULL, OP_MOV, OP_CANON_F, OP_FDIV_I, OP_MOD_I, OP_FDIV_U, OP_MOD_U, OP_BITLOOKUP, OP_MASKV) = range(28)
(AGG_NONE, AGG_SUM_F, AGG_SUM_I, AGG_COUNT, AGG_COUNT_ORD, AGG_LEN, AGG_MIN_F, AGG_MAX_F, AGG_MIN_I, AGG_MAX_I, AGG_MIN_U, AGG_MAX_U, AGG_FIRST_ROW) = range(13)
FIN_COPY64, FIN_TRUNC32, FIN_MEAN, FIN_MINMAX_I, FIN_MINMAX_F, FIN_NARROW = range(6)
# plx_dtype (include/polars_amd.h)
BOOL, I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 = range(11)
NP = {I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, U8: np.uint8, U16: np.uint16, U32: np.uint32, U64: np.uint64, F32: np.float32, F64: np.float64}
NONE = 255
U = np.uint64


def _f(a):
    return a.view(np.float64)


def _u(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def kleene(op, ta, va, tb, vb):
    """Kleene and / or on nullable Booleans (values, validity) -> (values, validity): false wins over null in `and`, true wins over null in `or`
    (crates/polars-core/src/chunked_array/comparison/mod.rs test_kleene; polars-compute bitwise kernels)."""
    if op == "and":
        return ta & tb, (~tb & vb) | (~ta & va) | (ta & va & tb & vb)
    return ta | tb, (ta & va) | (tb & vb) | (~ta & va & ~tb & vb)


def _cmp(op, a, b, floats):
    if floats:   # total order: NaN == NaN, NaN greatest (dev.hpp tot_*)
        an, bn = np.isnan(a), np.isnan(b)
        eq = (an & bn) | (a == b)
        lt = ~an & (bn | (a < b)) & ~eq
    else:
        eq, lt = a == b, a < b
    return [eq, ~eq, lt, lt | eq, ~(lt | eq), ~lt][op]


def _floor_div_mod(x, y, want_div, signed):
    nz = y != 0
    ys = np.where(nz, y, 1)
    with np.errstate(over="ignore"):
        if signed:
            xs, yy = x.view(np.int64), ys.view(np.int64)
            q, m = np.floor_divide(xs, yy), np.mod(xs, yy)          # numpy = Python sign rules = the reference's floor semantics
            neg1 = yy == -1                                          # wrapping_div(MIN, -1) = MIN, remainder 0
            q = np.where(neg1, (U(0) - x).view(np.int64), q); m = np.where(neg1, 0, m)
            r = (q if want_div else m).astype(np.int64).view(np.uint64)
        else:
            r = (x // ys) if want_div else (x % ys)
    return np.where(nz, r, U(0)), nz


def split_order(prog):
    """Op order of the late-materialisation split (fused.hpp split_program / fused_device.hpp run_split): every op the predicate
    and the key depend on (bit pc of early_mask), then the rest, each group in program order."""
    m = int(prog.get("early_mask", (1 << len(prog["ops"])) - 1))
    idx = range(len(prog["ops"]))
    return [i for i in idx if (m >> i) & 1] + [i for i in idx if not (m >> i) & 1]


def live_out_slots(prog):
    out = {a[1] for a in prog["aggs"]} | set(prog.get("keys", []))
    for k in ("pred", "key"):
        if prog[k] != NONE:
            out.add(prog[k])
    return out


def split_matches(prog, cols, luts=None):
    """True when running the program in split order leaves the predicate, key and aggregate sources exactly as program order does
    (what the engine's compile-time check promises whenever it reports any_late)."""
    a, pa = run_rows(prog, cols, luts)
    b, pb = run_rows(prog, cols, luts, split=True)
    if not np.array_equal(pa, pb):
        return False
    for s in live_out_slots(prog):
        (va, ma), (vb, mb) = a[s], b[s]
        if not np.array_equal(ma, mb) or not np.array_equal(va[ma], vb[mb]):
            return False
    return True


def run_rows(prog, cols, luts=None, split=False):
    """Executes the register program over all rows.  cols: {name: (values ndarray, valid bool ndarray or None)}.
    split: run the ops in split_order (the probe kernel's order) instead of program order.
    luts: {index: bool array over the lookup bitmap's key range} for OP_BITLOOKUP (fused_device.hpp: bit (a - imm) of lut c, 0
    outside the range, validity of a).  Returns (slots {slot: (u64 values, bool valid)}, pass mask)."""
    n = len(next(iter(cols.values()))[0]) if cols else 0
    slots = {}
    ones = np.ones(n, dtype=bool)
    for op in ([prog["ops"][i] for i in split_order(prog)] if split else prog["ops"]):
        code, dst, a, b, c, imm = op[0], op[1], op[2], op[3], op[4], U(int(op[5]))
        if code == OP_LOAD:
            inp = prog["inputs"][a]
            v, m = cols[inp["name"]]
            dt = inp["dtype"]
            if dt == BOOL:
                d = v.astype(np.uint64)
            elif dt == F64:
                d = _u(v)
            elif dt in (I8, I16, I32, I64):
                d = v.astype(np.int64).view(np.uint64)
            elif dt in (U8, U16, U32, U64):
                d = v.astype(np.uint64)
            else:
                raise NotImplementedError(f"input dtype {dt}")
            vd = ones.copy() if m is None else m.copy()
        elif code == OP_CONST:
            d, vd = np.full(n, imm, dtype=np.uint64), ones.copy()
        else:
            (x, vx), (y, vy) = slots[a], slots[b]
            vd = vx & vy
            with np.errstate(all="ignore"):
                if code == OP_ADD_F: d = _u(_f(x) + _f(y))
                elif code == OP_SUB_F: d = _u(_f(x) - _f(y))
                elif code == OP_MUL_F: d = _u(_f(x) * _f(y))
                elif code == OP_DIV_F: d = _u(_f(x) / _f(y))
                elif code == OP_ADD_I: d = x + y
                elif code == OP_SUB_I: d = x - y
                elif code == OP_MUL_I: d = x * y
                elif code == OP_I2F: d, vd = _u(x.view(np.int64).astype(np.float64)), vx
                elif code == OP_U2F: d, vd = _u(x.astype(np.float64)), vx
                elif code == OP_CMP_I: d = _cmp(c, x.view(np.int64), y.view(np.int64), False).astype(np.uint64)
                elif code == OP_CMP_U: d = _cmp(c, x, y, False).astype(np.uint64)
                elif code == OP_CMP_F: d = _cmp(c, _f(x), _f(y), True).astype(np.uint64)
                elif code in (OP_AND, OP_OR):
                    ta, tb = (x & U(1)).astype(bool), (y & U(1)).astype(bool)
                    t, vd = kleene("and" if code == OP_AND else "or", ta, vx, tb, vy)
                    d = t.astype(np.uint64)
                elif code == OP_XOR: d = (x ^ y) & U(1)
                elif code == OP_NOT: d, vd = (~x) & U(1), vx
                elif code == OP_CANON_F:
                    f = _f(x)
                    d, vd = np.where(np.isnan(f), U(0x7ff8000000000000), _u(f + 0.0)), vx
                elif code in (OP_FDIV_I, OP_MOD_I):
                    d, nz = _floor_div_mod(x, y, code == OP_FDIV_I, True); vd = vd & nz
                elif code in (OP_FDIV_U, OP_MOD_U):
                    d, nz = _floor_div_mod(x, y, code == OP_FDIV_U, False); vd = vd & nz
                elif code == OP_IFNULL: d, vd = np.where(vx, x, imm), ones.copy()
                elif code == OP_BITLOOKUP:
                    bits = luts[c]
                    idx = x - imm                                     # wrapping u64: below the range -> huge -> outside
                    inside = vx & (idx < U(len(bits)))
                    d = np.zeros(n, np.uint64)
                    d[inside] = bits[idx[inside].astype(np.int64)].astype(np.uint64)
                    vd = vx
                elif code == OP_MASKV: d, vd = x, vx & vy & (y & U(1)).astype(bool)
                elif code in (OP_MOV, OP_NOP): d, vd = x, vx
                else:
                    raise NotImplementedError(f"opcode {code}")
        slots[dst] = (np.ascontiguousarray(d, dtype=np.uint64), vd)
        # snapshot semantics: a later op may overwrite a slot; consumers read the value current at their turn
    if prog["pred"] == NONE:
        passed = ones
    else:
        pv, pm = slots[prog["pred"]]
        passed = (pv & U(1)).astype(bool) & pm
    return slots, passed


def _snapshot_sources(prog, cols):
    """Aggregate / key sources must be read when the program ends; a slot can be reused, so re-run and capture the
    final contents (the compiler guarantees sources stay live until the end)."""
    return run_rows(prog, cols)


def _cells(prog, slots, passed, rows, group_of, n_groups):
    """Aggregate cells [n_groups][n_aggs] as uint64 bit patterns (agg_row_value + agg_combine)."""
    out = np.zeros((n_groups, len(prog["aggs"])), dtype=np.uint64)
    for k, (kind, src) in enumerate(prog["aggs"]):
        if kind in (AGG_LEN, AGG_FIRST_ROW) or src == NONE:     # these kinds read no source slot (the compiler stores src = 0 for them)
            v, valid = np.zeros(len(passed), np.uint64), np.ones(len(passed), bool)
        else:
            v, valid = slots[src]
        sel = passed & valid if kind not in (AGG_LEN, AGG_FIRST_ROW) else passed
        g = group_of[sel]
        if kind in (AGG_LEN, AGG_COUNT):
            out[:, k] = np.bincount(g, minlength=n_groups).astype(np.uint64)
        elif kind == AGG_COUNT_ORD:
            ok = sel & ~np.isnan(_f(v))
            out[:, k] = np.bincount(group_of[ok], minlength=n_groups).astype(np.uint64)
        elif kind == AGG_SUM_I:
            acc = np.zeros(n_groups, np.uint64); np.add.at(acc, g, v[sel]); out[:, k] = acc
        elif kind == AGG_SUM_F:
            acc = np.zeros(n_groups, np.float64); np.add.at(acc, g, _f(v)[sel]); out[:, k] = _u(acc)
        elif kind in (AGG_MIN_F, AGG_MAX_F):
            f = _f(v); ok = sel & ~np.isnan(f)
            acc = np.full(n_groups, np.inf if kind == AGG_MIN_F else -np.inf)
            (np.minimum if kind == AGG_MIN_F else np.maximum).at(acc, group_of[ok], f[ok]); out[:, k] = _u(acc)
        elif kind in (AGG_MIN_I, AGG_MAX_I):
            acc = np.full(n_groups, np.iinfo(np.int64).max if kind == AGG_MIN_I else np.iinfo(np.int64).min, dtype=np.int64)
            (np.minimum if kind == AGG_MIN_I else np.maximum).at(acc, g, v.view(np.int64)[sel]); out[:, k] = acc.view(np.uint64)
        elif kind in (AGG_MIN_U, AGG_MAX_U):
            acc = np.full(n_groups, np.iinfo(np.uint64).max if kind == AGG_MIN_U else 0, dtype=np.uint64)
            (np.minimum if kind == AGG_MIN_U else np.maximum).at(acc, g, v[sel]); out[:, k] = acc
        elif kind == AGG_FIRST_ROW:
            acc = np.full(n_groups, np.iinfo(np.uint64).max, dtype=np.uint64)
            np.minimum.at(acc, g, rows[sel].astype(np.uint64)); out[:, k] = acc
        else:
            raise NotImplementedError(f"aggregate kind {kind}")
    return out


def _finalise(fs, cells):
    """One output column from the cells (finalize_batch_kernel): -> (values, valid or None)."""
    kind, a, b, c, dt = fs["kind"], fs["a"], fs["b"], fs["c"], fs["out_dtype"]
    ca = cells[:, a]
    if kind in (FIN_COPY64, FIN_NARROW, FIN_TRUNC32):
        if dt == F64:
            return _f(np.ascontiguousarray(ca)).copy(), None
        return ca.astype(NP[dt]) if dt in (U8, U16, U32, U64) else ca.view(np.int64).astype(NP[dt]), None   # wrapping narrow
    cnt = cells[:, b]
    if kind == FIN_MEAN:
        with np.errstate(all="ignore"):
            return _f(np.ascontiguousarray(ca)) / cnt.astype(np.float64), cnt != 0
    if kind == FIN_MINMAX_I:
        vals = ca.astype(NP[dt]) if dt in (U8, U16, U32, U64) else ca.view(np.int64).astype(NP[dt])
        return vals, cnt != 0
    if kind == FIN_MINMAX_F:
        vals = _f(np.ascontiguousarray(ca)).copy()
        vals[(cells[:, c] == 0) & (cnt != 0)] = np.nan              # only NaNs among the valid rows
        return vals, cnt != 0
    raise NotImplementedError(f"final kind {kind}")


def evaluate(prog, cols, luts=None):
    """-> {output name: (values, valid or None)}; group_by results carry one row per group, in unspecified order."""
    slots, passed = run_rows(prog, cols, luts)
    n = len(passed)
    rows = np.arange(n)
    res = {}
    if prog["kind"] == "select":
        cells = _cells(prog, slots, passed, rows, np.zeros(n, dtype=np.int64), 1)
    else:
        kp = prog["key_plan"]
        if kp["wide"]:
            words = [np.where(slots[s][1], slots[s][0], U(0)) for s in prog["keys"]]
            valids = [slots[s][1] for s in prog["keys"]]
            ident = np.stack(words + [v.astype(np.uint64) for v in valids], axis=1)
        else:
            kv, km = slots[prog["key"]]
            ident = np.stack([np.where(km, kv, U(0)), km.astype(np.uint64)], axis=1)
        sel_rows = np.nonzero(passed)[0]
        uniq, inv = np.unique(ident[sel_rows], axis=0, return_inverse=True)
        group_of = np.zeros(n, dtype=np.int64); group_of[sel_rows] = np.asarray(inv).reshape(-1)
        G = len(uniq)
        cells = _cells(prog, slots, passed, rows, group_of, G)
        nk = len(kp["parts"])
        for i, part in enumerate(kp["parts"]):
            dt = part["dtype"]
            if kp["wide"]:
                word, valid = uniq[:, i], uniq[:, nk + i].astype(bool)
            elif kp["packed"]:
                code = (uniq[:, 0] >> U(part["shift"])) & U(int(part["mask"]))
                null_code = int(part["null_code"])
                valid = code != U(null_code) if null_code != (1 << 64) - 1 else np.ones(G, bool)
                word = code + np.int64(int(part["min"])).astype(np.uint64) if int(part["min"]) >= 0 else code - U(-int(part["min"]))
            else:
                word, valid = uniq[:, 0], uniq[:, 1].astype(bool)
            if dt == F64:
                vals = _f(np.ascontiguousarray(word)).copy()
            elif dt == BOOL:
                vals = (word & U(1)).astype(bool)
            elif dt in (U8, U16, U32, U64):
                vals = word.astype(NP[dt])
            else:
                vals = word.view(np.int64).astype(NP[dt])
            res[part["name"]] = (vals, None if valid.all() else valid)
    for o in prog["outputs"]:
        if o["final"] < 0:
            raise NotImplementedError(f"output {o['name']} is a row expression over aggregates (evaluated by the per-node kernels)")
        res[o["name"]] = _finalise(prog["finals"][o["final"]], cells)
    return res


def build_luts(prog, filter_cols):
    """Membership bitmaps of the pipeline's semi filters (engine.cpp: nested inner joins that only filter): for each entry of
    prog["semis"] the filter program runs over its frame (filter_cols[i]) and sets bit (key - kmin) of every passing row with a
    valid key.  Returns {lut index: bool array}, or None when a filter side's keys are not unique (the rewrite does not apply)."""
    luts = {}
    for sm, cols in zip(prog.get("semis", []), filter_cols):
        slots, passed = run_rows(sm["filter"], cols)
        kv, km = slots[sm["filter"]["key"]]
        kmin, kmax = int(sm["kmin"]), int(sm["kmax"])
        bits = np.zeros(max(kmax - kmin + 1, 1), bool)
        idx = (kv[passed & km].view(np.int64) - kmin)
        idx = idx[(idx >= 0) & (idx < len(bits))]
        if len(np.unique(idx)) != len(idx):
            return None
        bits[idx] = True
        luts[sm["lut"]] = bits
    return luts


def evaluate_join(prog, build_cols, probe_cols, filter_cols=()):
    """The fused join -> group-by pipeline (engine.cpp fused_join_groupby): build scan (predicate + key) -> probe scan
    (predicate + key + aggregates landing in the cells of the matching build rows) -> groups with at least one probe row.
    Build keys may repeat (round 5, the multi-value mode of the table): every probe row then contributes once per build row of
    its key, a group is a build row -- or the rows of a key that agree on all build-side group columns (one group of the
    reference's group-by over the joined frame).  prog["how"] == "left": the probe rows without a partner are grouped by
    their own key (prog["unmatched"]: a group_by program behind NOT member(key)), nulls in the build-side group columns.
    build_cols / probe_cols: {original column name of that frame: (values, valid or None)}; filter_cols: the frames of
    prog["semis"] in order.  Returns {output name: (values, valid or None)}, or None when a semi filter's keys are not unique."""
    luts = build_luts(prog, filter_cols)
    if luts is None:
        return None
    b_slots, b_pass = run_rows(prog["build"], build_cols, luts)
    bk, bkm = b_slots[prog["build"]["key"]]
    ins = b_pass & bkm                                   # null build keys are never inserted
    rows_b = np.nonzero(ins)[0]
    keys_b = bk[rows_b]
    # the count scan sizes the tables: it must count exactly the inserted rows
    c_slots, c_pass = run_rows(prog["count"], build_cols, luts)
    ccells = _cells(prog["count"], c_slots, c_pass, np.arange(len(c_pass)), np.zeros(len(c_pass), dtype=np.int64), 1)
    assert int(ccells[0, 0]) == len(rows_b), (int(ccells[0, 0]), len(rows_b))
    order = np.argsort(keys_b, kind="stable")
    skeys, srows = keys_b[order], rows_b[order]
    nb = len(skeys)
    # representatives: the first build row (in this order) among the rows of one key that agree on every build-side group column
    ident = [skeys.astype(np.uint64)]
    for gk in prog["group_keys"]:
        if not gk["is_join_key"]:
            v, m = build_cols[gk["build_col"]]
            mm = np.ones(len(v), bool) if m is None else m
            ident += [np.where(mm[srows], v[srows].astype(np.int64).view(np.uint64), U(0)), mm[srows].astype(np.uint64)]
    if nb:
        _, first, inv = np.unique(np.stack(ident, axis=1), axis=0, return_index=True, return_inverse=True)
        rep = first[np.asarray(inv).reshape(-1)]
    else:
        rep = np.zeros(0, np.int64)
    p_slots, p_pass = run_rows(prog["probe"], probe_cols, luts, split=True)     # the probe kernel's op order
    pk, pkm = p_slots[prog["probe"]["key"]]
    lo, hi = np.searchsorted(skeys, pk, "left"), np.searchsorted(skeys, pk, "right")
    sel = p_pass & pkm & (hi > lo)
    prow = np.nonzero(sel)[0]
    cnt = (hi - lo)[prow]
    xrow = np.repeat(prow, cnt)                                                  # one entry per (probe row, build row of its key)
    xpos = np.repeat(lo[prow], cnt) + (np.arange(len(xrow)) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    x_slots = {s_: (v[xrow], m[xrow]) for s_, (v, m) in p_slots.items()}
    cells = _cells(prog["probe"], x_slots, np.ones(len(xrow), bool), xrow, rep[xpos].astype(np.int64), nb)
    live = cells[:, prog["len_idx"]] != 0 if nb else np.zeros(0, bool)
    cells = cells[live]
    res = {}
    for gk in prog["group_keys"]:
        if gk["is_join_key"]:
            word = skeys[live]
            dt = gk["dtype"]
            vals = word.astype(NP[dt]) if dt in (U8, U16, U32, U64) else word.view(np.int64).astype(NP[dt])
            res[gk["name"]] = (vals, None)
        else:
            v, m = build_cols[gk["build_col"]]
            rows = srows[live]
            res[gk["name"]] = (v[rows], None if m is None or m[rows].all() else m[rows])
    for o in prog["outputs"]:
        if o["final"] < 0:
            raise NotImplementedError(f"output {o['name']} is a row expression over aggregates")
        res[o["name"]] = _finalise(prog["finals"][o["final"]], cells)
    if prog.get("how") == "left":
        un = prog["unmatched"]
        kmin, rng = int(un["kmin"]), int(un["range"])
        bits = np.zeros(max(rng, 1), bool)
        idx = skeys.view(np.int64) - kmin
        bits[idx[(idx >= 0) & (idx < len(bits))]] = True
        # (the unmatched program numbers its lookups on its own: the probe side's semi filters first, then the membership bitmap)
        n_bsemi = sum(1 for sm in prog.get("semis", []) if sm["side"] == "build")
        un_luts = {i - n_bsemi: b for i, b in luts.items() if i >= n_bsemi}
        un_luts[un["lut"]] = bits
        tail = evaluate(un["program"], probe_cols, un_luts)
        n_t = len(next(iter(tail.values()))[0])
        for name, (v, m) in list(res.items()):
            if name in tail:
                tv, tm = tail[name]
            else:                                                                # a build-side group column: null for a row without a partner
                tv, tm = np.zeros(n_t, v.dtype), np.zeros(n_t, bool)
            mm = np.concatenate([np.ones(len(v), bool) if m is None else m, np.ones(n_t, bool) if tm is None else tm])
            res[name] = (np.concatenate([v, tv.astype(v.dtype)]), None if mm.all() else mm)
    return res
