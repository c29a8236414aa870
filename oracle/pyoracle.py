"""numpy-facing binding of the CPU oracle (oracle/plx_oracle.cpp -> libplx_oracle.so).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under polars_amd/ does.  The oracle is a
restatement of the reference (pola-rs/polars 0.55.1) algorithms -- see the header of
plx_oracle.cpp for the file:line each function follows -- pinned by the golden vectors
transcribed from the reference's own tests (tests/golden/, tests/test_oracle_golden.py).

The query helpers at the bottom (q_cfg2, q_groupby, q1, q3) execute the benchmark queries
in the *reference's* operator order: materialise the predicate bitmap, filter every
column, evaluate each arithmetic node into a full column, build per-group index lists,
aggregate per group (SURVEY.md sections 3.2-3.4).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libplx_oracle.so")

BOOL, I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 = range(11)
EQ, NE, LT, LE, GT, GE = range(6)
ADD, SUB, MUL, TRUE_DIV, FLOOR_DIV, MOD = range(6)
AGG_SUM, AGG_MEAN, AGG_MIN, AGG_MAX, AGG_COUNT, AGG_LEN, AGG_FIRST = range(7)
JOIN_INNER, JOIN_LEFT = range(2)

NP_OF = {I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, U8: np.uint8, U16: np.uint16, U32: np.uint32, U64: np.uint64,
         F32: np.float32, F64: np.float64}
DT_OF = {np.dtype(v): k for k, v in NP_OF.items()}

_lib: Optional[C.CDLL] = None


def build() -> None:
    """Compile the oracle with the committed recipe (oracle/Makefile)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.orc_groupby_build.restype = C.c_void_p
        l.orc_join.restype = C.c_void_p
        l.orc_groups_count.restype = C.c_int64
        l.orc_pairs_count.restype = C.c_int64
        l.orc_partitioner_seed.restype = C.c_uint64
        l.orc_partitioner_seed.argtypes = [C.c_uint64]
        _lib = l
    return _lib


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def hardware_threads() -> int:
    return int(lib().orc_hardware_threads())


def _p(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def pack(bits: Optional[np.ndarray]) -> Optional[np.ndarray]:
    """bool array -> LSB-first bitmap padded to whole u64 words (+8 bytes)."""
    if bits is None:
        return None
    b = np.packbits(np.asarray(bits, dtype=bool), bitorder="little")
    out = np.zeros(((len(bits) + 63) // 64) * 8 + 8, dtype=np.uint8)
    out[: len(b)] = b
    return out


def unpack(bitmap: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(bitmap, bitorder="little")[:n].astype(bool)


def _dt(a: np.ndarray) -> int:
    return DT_OF[a.dtype]


# ------------------------------------------------------------------- kernels ----
def cmp(op: int, a: np.ndarray, b) -> np.ndarray:
    """Values only (validity = AND of inputs is the caller's business, arity.rs:203-214)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.bool_:
        # Boolean arrays compare as bitmaps, false < true (crates/polars-compute/src/comparisons/boolean.rs:9-70): eq !(l ^ r), ne l ^ r, lt !l & r, le !l | r
        l, r = a, (np.full(len(a), bool(b)) if not isinstance(b, np.ndarray) else b.astype(bool))
        return {EQ: ~(l ^ r), NE: l ^ r, LT: ~l & r, LE: ~l | r, GT: l & ~r, GE: l | ~r}[op]
    n = len(a)
    scalar = not isinstance(b, np.ndarray)
    bb = np.array([b], dtype=a.dtype) if scalar else np.ascontiguousarray(b.astype(a.dtype, copy=False))
    out = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
    rc = lib().orc_cmp(_dt(a), op, _p(a), _p(bb), int(scalar), C.c_int64(n), _p(out))
    assert rc == 0
    return unpack(out, n)


def arith(op: int, a, b, mode: int = 0) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """mode 0: col OP col, 1: col OP scalar(b), 2: scalar(a) OP col(b).
    Returns (values, extra_valid) where extra_valid is the rhs != 0 mask of integer floor-div / mod."""
    if mode == 2:
        col = np.ascontiguousarray(b)
        l = np.array([a], dtype=col.dtype)
        r = col
    elif mode == 1:
        col = np.ascontiguousarray(a)
        l = col
        r = np.array([b], dtype=col.dtype)
    else:
        col = np.ascontiguousarray(a)
        l = col
        r = np.ascontiguousarray(b)
    n = len(col)
    dt = _dt(col)
    odt = F64 if (op == TRUE_DIV and dt not in (F32, F64)) else dt
    out = np.zeros(n, dtype=NP_OF[odt])
    extra = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
    has_extra, out_dt = C.c_int(0), C.c_int(0)
    rc = lib().orc_arith(dt, op, _p(l), _p(r), mode, C.c_int64(n), _p(out), _p(extra), C.byref(has_extra), C.byref(out_dt))
    assert rc == 0 and out_dt.value == odt
    return out, (unpack(extra, n) if has_extra.value else None)


def filter(values: np.ndarray, validity: Optional[np.ndarray], mask: np.ndarray, mask_validity: Optional[np.ndarray] = None):
    """(filtered values, filtered validity or None). `values` bool => bitmap column."""
    n = len(values)
    isb = values.dtype == np.bool_
    vin = pack(values) if isb else np.ascontiguousarray(values)
    width = 0 if isb else values.dtype.itemsize
    vout = np.zeros(((n + 63) // 64) * 8 + 8, dtype=np.uint8) if isb else np.zeros(n, dtype=values.dtype)
    nout = C.c_int64(0)
    vb = pack(validity) if validity is not None else None
    vob = np.zeros((n + 7) // 8 + 8, dtype=np.uint8) if validity is not None else None
    mb, mvb = pack(mask), (pack(mask_validity) if mask_validity is not None else None)
    rc = lib().orc_filter(width, _p(vin), _p(vb), _p(mb), _p(mvb), C.c_int64(n), _p(vout), _p(vob), C.byref(nout))
    assert rc == 0
    k = nout.value
    ov = unpack(vout, k) if isb else vout[:k].copy()
    return ov, (unpack(vob, k) if validity is not None else None)


def gather(values: np.ndarray, validity: Optional[np.ndarray], idx: np.ndarray, idx_validity: Optional[np.ndarray] = None):
    n = len(idx)
    isb = values.dtype == np.bool_
    vin = pack(values) if isb else np.ascontiguousarray(values)
    width = 0 if isb else values.dtype.itemsize
    out = np.zeros((n + 7) // 8 + 8, dtype=np.uint8) if isb else np.zeros(n, dtype=values.dtype)
    ov = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    rc = lib().orc_gather(width, _p(vin), _p(pack(validity)), _p(idx), _p(pack(idx_validity)), C.c_int64(n), _p(out), _p(ov))
    assert rc == 0
    return (unpack(out, n) if isb else out), unpack(ov, n)


def _scalar_from_bits(bits: int, dt: int):
    raw = np.array([bits], dtype=np.uint64)
    if dt == F64:
        return float(raw.view(np.float64)[0])
    if dt == F32:
        return float(raw.view(np.float32)[0])
    if dt in (U8, U16, U32, U64):
        return int(bits) & ((1 << (8 * np.dtype(NP_OF[dt]).itemsize)) - 1)
    w = 8 * np.dtype(NP_OF[dt]).itemsize
    x = int(bits) & ((1 << w) - 1)
    return x - (1 << w) if x >= (1 << (w - 1)) else x


def reduce(op: int, values: np.ndarray, validity: Optional[np.ndarray] = None):
    """-> (python value or None, output dtype code)."""
    n = len(values)
    isb = values.dtype == np.bool_
    vin = pack(values) if isb else np.ascontiguousarray(values)
    dt = BOOL if isb else _dt(values)
    bits, odt, ok = C.c_uint64(0), C.c_int(0), C.c_int(0)
    rc = lib().orc_reduce(dt, op, _p(vin), _p(pack(validity)), C.c_int64(n), C.byref(bits), C.byref(odt), C.byref(ok))
    assert rc == 0, f"orc_reduce rc={rc}"
    if not ok.value:
        return None, odt.value
    return _scalar_from_bits(bits.value, odt.value), odt.value


def key_bits(k: np.ndarray) -> np.ndarray:
    """to_bit_repr (into_groups.rs:156-186) + float canonicalisation (total_ord.rs:40-48)."""
    if k.dtype == np.bool_:
        return k.astype(np.uint64)
    if k.dtype.kind == "f":
        d = k.astype(np.float64) + 0.0
        bits = d.view(np.uint64).copy()
        bits[np.isnan(d)] = np.uint64(0x7FF8000000000000)
        return bits
    if k.dtype.kind == "i":
        return k.astype(np.int64).view(np.uint64).copy()
    return k.astype(np.uint64)


class Groups:
    def __init__(self, keys: Sequence[np.ndarray], valids: Sequence[Optional[np.ndarray]], maintain_order: bool = False):
        self.n = len(keys[0])
        self._kb = [np.ascontiguousarray(key_bits(k)) for k in keys]
        self._vb = [pack(v) for v in valids]
        nk = len(keys)
        kp = (C.c_void_p * nk)(*[b.ctypes.data for b in self._kb])
        vp = (C.c_void_p * nk)(*[(b.ctypes.data if b is not None else None) for b in self._vb])
        self._g = C.c_void_p(lib().orc_groupby_build(nk, kp, vp, C.c_int64(self.n), int(maintain_order)))
        assert self._g.value, "orc_groupby_build failed"
        self.count = int(lib().orc_groups_count(self._g))
        self.first = np.zeros(self.count, dtype=np.uint32)
        lib().orc_groups_first(self._g, _p(self.first))

    def agg(self, op: int, values: Optional[np.ndarray], validity: Optional[np.ndarray] = None):
        """-> (out values, out validity bool array)."""
        G = self.count
        if op == AGG_LEN or values is None:
            out = np.zeros(G, dtype=np.uint32)
            ov = np.zeros((G + 7) // 8 + 8, dtype=np.uint8)
            odt = C.c_int(0)
            rc = lib().orc_groups_agg(self._g, I64, AGG_LEN, None, None, _p(out), _p(ov), C.byref(odt))
            assert rc == 0
            return out, unpack(ov, G)
        isb = values.dtype == np.bool_
        v = values.astype(np.uint8) if isb else np.ascontiguousarray(values)
        dt = _dt(v)
        out = np.zeros(max(G, 1) * 8, dtype=np.uint8)
        ov = np.zeros((G + 7) // 8 + 8, dtype=np.uint8)
        odt = C.c_int(0)
        rc = lib().orc_groups_agg(self._g, dt, op, _p(v), _p(pack(validity)), _p(out), _p(ov), C.byref(odt))
        assert rc == 0
        o = out.view(NP_OF[odt.value])[:G].copy()
        return o, unpack(ov, G)

    def __del__(self):
        try:
            if self._g:
                lib().orc_groups_free(self._g)
        except Exception:
            pass


def join(how: int, lk: np.ndarray, lv: Optional[np.ndarray], rk: np.ndarray, rv: Optional[np.ndarray]):
    """-> (left_idx u32, right_idx u32, right_valid bool or None)"""
    a, b = np.ascontiguousarray(key_bits(lk)), np.ascontiguousarray(key_bits(rk))
    p = C.c_void_p(lib().orc_join(how, _p(a), _p(pack(lv)), C.c_int64(len(a)), _p(b), _p(pack(rv)), C.c_int64(len(b))))
    m = int(lib().orc_pairs_count(p))
    li, ri = np.zeros(m, dtype=np.uint32), np.zeros(m, dtype=np.uint32)
    rvb = np.zeros((m + 7) // 8 + 8, dtype=np.uint8)
    lib().orc_pairs_get(p, _p(li), _p(ri), _p(rvb))
    lib().orc_pairs_free(p)
    return li, ri, (unpack(rvb, m) if how == JOIN_LEFT else None)


JOIN_SEMI, JOIN_ANTI = 2, 3


def semi_anti_join(how: int, lk: np.ndarray, lv: Optional[np.ndarray], rk: np.ndarray, rv: Optional[np.ndarray]) -> np.ndarray:
    """Left row indices kept by a SEMI (how=2) / ANTI (how=3) join, in left order.
    Restates polars-ops/src/frame/join/hash_join/single_keys_semi_anti.rs: a hash SET of the valid right keys is
    built, every left row is probed in order; a null left key never matches (kept by ANTI, dropped by SEMI)."""
    a, b = key_bits(lk), key_bits(rk)
    if rv is not None:
        b = b[rv]
    matched = np.isin(a, b)
    if lv is not None:
        matched &= lv
    keep = matched if how == JOIN_SEMI else ~matched
    return np.nonzero(keep)[0].astype(np.uint32)


def encode_key_rows(lkeys, rkeys):
    """Multi-column join keys -> one id per distinct key tuple, shared by both sides (the role polars-row's row
    encoding plays at polars-ops/src/frame/join/mod.rs:367-370): returns (left ids i64, left valid, right ids, right valid).
    lkeys / rkeys = [(values, valid or None)] per key column; a null in any column makes the row's key null."""
    nl = len(lkeys[0][0])
    cols = [np.concatenate([key_bits(a), key_bits(b)]) for (a, _), (b, _) in zip(lkeys, rkeys)]
    _, inv = np.unique(np.stack(cols, axis=1), axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1).astype(np.int64)

    def all_valid(keys, n):
        v = np.ones(n, dtype=bool)
        for _, m in keys:
            if m is not None:
                v &= m
        return None if v.all() else v
    return inv[:nl], all_valid(lkeys, nl), inv[nl:], all_valid(rkeys, len(rkeys[0][0]))


def _tot_cmp(a, b) -> int:
    """TotalOrd for one non-null value pair (polars-utils/src/total_ord.rs): NaN == NaN, NaN greatest, -0.0 == 0.0."""
    an, bn = isinstance(a, float) and a != a, isinstance(b, float) and b != b
    if an or bn:
        return 0 if (an and bn) else (1 if an else -1)
    return (a > b) - (a < b)


def sort_indices_cmp(keys, limit: Optional[int] = None) -> np.ndarray:
    """arg_sort_multiple restated with the reference's comparator, for SMALL inputs (pure Python).
    keys = [(values, valid or None, descending, nulls_last)].  Per key reorder_cmp (polars-utils/src/sort.rs:113-130):
    equal -> next key (ordering_other_columns, polars-core/src/chunked_array/ops/sort/mod.rs:350-367); a null is
    Greater when nulls_last else Less (NOT flipped by descending); otherwise the total order, reversed if descending.
    Ties keep input order (maintain_order / stable, arg_sort_multiple.rs:64-75)."""
    import functools
    n = len(keys[0][0])
    cols = [(v.tolist(), None if m is None else m.tolist(), bool(d), bool(nl)) for v, m, d, nl in keys]

    def cmp_rows(i, j):
        for v, m, d, nl in cols:
            a_null, b_null = m is not None and not m[i], m is not None and not m[j]
            if a_null and b_null:
                continue
            if a_null:
                return 1 if nl else -1
            if b_null:
                return -1 if nl else 1
            c = _tot_cmp(v[i], v[j])
            if c:
                return -c if d else c
        return 0
    order = sorted(range(n), key=functools.cmp_to_key(cmp_rows))
    return np.array(order[:limit] if limit is not None else order, dtype=np.uint32)


def sort_indices(keys, limit: Optional[int] = None) -> np.ndarray:
    """Same order as sort_indices_cmp, vectorised: every key becomes (null rank, dense value rank) and the stable
    np.lexsort applies them last key first.  Dense ranks come from np.unique, which orders NaN last (greatest) and
    treats -0.0 == 0.0, i.e. the reference's TotalOrd."""
    lex = []   # np.lexsort: LAST entry is the primary key
    for v, m, d, nl in reversed(list(keys)):
        v = np.asarray(v)
        if v.dtype == np.bool_:
            v = v.astype(np.uint8)
        _, rank = np.unique(v, return_inverse=True)
        rank = rank.astype(np.int64)
        if d:
            rank = -rank
        if m is not None:
            rank = np.where(m, rank, 0)                    # all nulls tie within a key
            null_rank = np.where(m, 0, 1) if nl else np.where(m, 1, 0)
            lex.append(rank)
            lex.append(null_rank)
        else:
            lex.append(rank)
    order = np.lexsort(lex).astype(np.uint32)
    return order[:limit] if limit is not None else order


def hash_partition(keys: np.ndarray, valid: Optional[np.ndarray], n_parts: int, seed: int = 0) -> np.ndarray:
    kb = np.ascontiguousarray(key_bits(keys))
    out = np.zeros(len(kb), dtype=np.uint32)
    lib().orc_hash_partition(_p(kb), _p(pack(valid)), C.c_int64(len(kb)), int(n_parts), C.c_uint64(seed), _p(out))
    return out


# --------------------------------------------------- reference-shaped query plans ----
def q_filter_agg_cfg2(a: np.ndarray, x: np.ndarray, y: np.ndarray, k: int, x_valid: Optional[np.ndarray] = None) -> Dict[str, object]:
    """filter(a > k).select((x*(1-y)).sum(), x.mean(), a.sum())  (BASELINE config 2).
    FilterExec materialises the mask and filters all three columns (filter.rs:94-114),
    then ProjectionExec evaluates each expression node into a column."""
    m = cmp(GT, a, k)
    af, _ = filter(a, None, m)
    xf, xv = filter(x, x_valid, m)
    yf, _ = filter(y, None, m)
    one_minus, _ = arith(SUB, 1.0, yf, mode=2)
    prod, _ = arith(MUL, xf, one_minus, mode=0)
    return {"xy": reduce(AGG_SUM, prod, xv)[0], "x_mean": reduce(AGG_MEAN, xf, xv)[0], "a_sum": reduce(AGG_SUM, af)[0], "rows": int(m.sum())}


def q_filter_sum_cfg1(a: np.ndarray, k: int) -> int:
    m = cmp(GT, a, k)
    af, _ = filter(a, None, m)
    return reduce(AGG_SUM, af)[0]


def q_groupby(keys: Sequence[np.ndarray], key_valids: Sequence[Optional[np.ndarray]], aggs: List[Tuple[str, int, Optional[np.ndarray], Optional[np.ndarray]]],
              maintain_order: bool = False) -> Dict[str, Tuple[np.ndarray, Optional[np.ndarray]]]:
    """group_by(keys).agg(...): aggs = [(name, op, values, validity)].  Returns
    {key_i: (values, validity), name: (values, validity)} in the oracle's group order."""
    g = Groups(keys, key_valids, maintain_order)
    out: Dict[str, Tuple[np.ndarray, Optional[np.ndarray]]] = {}
    for i, (k, kv) in enumerate(zip(keys, key_valids)):
        vals, val = gather(k, kv, g.first)
        out[f"key_{i}"] = (vals, val)
    for name, op, v, vv in aggs:
        out[name] = g.agg(op, v, vv)
    return out


def q1(cols: Dict[str, np.ndarray], cutoff: int) -> Dict[str, np.ndarray]:
    """TPC-H Q1 in the reference's operator order; returns rows
    sorted by (l_returnflag, l_linestatus)."""
    m = cmp(LE, cols["l_shipdate"], cutoff)
    f = {k: filter(v, None, m)[0] for k, v in cols.items()}
    one_minus, _ = arith(SUB, 1.0, f["l_discount"], mode=2)
    disc_price, _ = arith(MUL, f["l_extendedprice"], one_minus)
    one_plus, _ = arith(ADD, 1.0, f["l_tax"], mode=2)
    charge, _ = arith(MUL, disc_price, one_plus)
    r = q_groupby([f["l_returnflag"], f["l_linestatus"]], [None, None], [
        ("sum_qty", AGG_SUM, f["l_quantity"], None), ("sum_base_price", AGG_SUM, f["l_extendedprice"], None),
        ("sum_disc_price", AGG_SUM, disc_price, None), ("sum_charge", AGG_SUM, charge, None),
        ("avg_qty", AGG_MEAN, f["l_quantity"], None), ("avg_price", AGG_MEAN, f["l_extendedprice"], None),
        ("avg_disc", AGG_MEAN, f["l_discount"], None), ("count_order", AGG_LEN, None, None)])
    order = np.lexsort((r["key_1"][0], r["key_0"][0]))
    out = {"l_returnflag": r["key_0"][0][order], "l_linestatus": r["key_1"][0][order]}
    for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"):
        out[k] = r[k][0][order]
    return out


def q3(li: Dict[str, np.ndarray], orders: Dict[str, np.ndarray], date: int, seg_mod: int = 5) -> Dict[str, np.ndarray]:
    """Q3 restated on the two big tables (SURVEY.md 8(d) cfg 4): filter both sides, inner join
    on orderkey, gather payload, revenue = price*(1-disc), group by (orderkey, orderdate, shippriority).
    Returns rows sorted by l_orderkey."""
    mo1 = cmp(LT, orders["o_orderdate"], date)
    modv, _ = arith(MOD, orders["o_custkey"], seg_mod, mode=1)
    mo = mo1 & cmp(EQ, modv, 0)
    o = {k: filter(v, None, mo)[0] for k, v in orders.items()}
    ml = cmp(GT, li["l_shipdate"], date)
    l = {k: filter(v, None, ml)[0] for k, v in li.items()}
    lidx, ridx, _ = join(JOIN_INNER, l["l_orderkey"], None, o["o_orderkey"], None)
    j = {k: gather(v, None, lidx)[0] for k, v in l.items()}
    j.update({k: gather(v, None, ridx)[0] for k, v in o.items() if k != "o_orderkey"})
    one_minus, _ = arith(SUB, 1.0, j["l_discount"], mode=2)
    rev, _ = arith(MUL, j["l_extendedprice"], one_minus)
    r = q_groupby([j["l_orderkey"], j["o_orderdate"], j["o_shippriority"]], [None, None, None], [("revenue", AGG_SUM, rev, None)])
    order = np.argsort(r["key_0"][0], kind="stable")
    return {"l_orderkey": r["key_0"][0][order], "o_orderdate": r["key_1"][0][order], "o_shippriority": r["key_2"][0][order],
            "revenue": r["revenue"][0][order]}


def q3_full(customer: Dict[str, np.ndarray], orders: Dict[str, np.ndarray], li: Dict[str, np.ndarray], date: int, segment_code: int) -> Dict[str, np.ndarray]:
    """TPC-H Q3 with all three tables in the reference's operator order (predicates already pushed below the joins, as the
    optimizer does): FilterExec on each scan, JoinExec customer x orders on custkey (inner; the joined frame is materialised
    by gathers, hash_join/single_keys_inner.rs + frame/join/mod.rs:564-652), JoinExec with lineitem on orderkey, the revenue
    column, GroupByExec on (o_orderkey, o_orderdate, o_shippriority).  Returns rows sorted by o_orderkey."""
    mc = cmp(EQ, customer["c_mktsegment"], np.array(segment_code, dtype=customer["c_mktsegment"].dtype).item())
    c = {k: filter(v, None, mc)[0] for k, v in customer.items()}
    mo = cmp(LT, orders["o_orderdate"], date)
    o = {k: filter(v, None, mo)[0] for k, v in orders.items()}
    cidx, oidx, _ = join(JOIN_INNER, c["c_custkey"], None, o["o_custkey"], None)
    co = {k: gather(v, None, oidx)[0] for k, v in o.items() if k != "o_custkey"}     # right key coalesced into c_custkey
    co.update({k: gather(v, None, cidx)[0] for k, v in c.items()})
    ml = cmp(GT, li["l_shipdate"], date)
    l = {k: filter(v, None, ml)[0] for k, v in li.items()}
    a_idx, l_idx, _ = join(JOIN_INNER, co["o_orderkey"], None, l["l_orderkey"], None)
    j = {k: gather(v, None, a_idx)[0] for k, v in co.items()}
    j.update({k: gather(v, None, l_idx)[0] for k, v in l.items() if k != "l_orderkey"})
    one_minus, _ = arith(SUB, 1.0, j["l_discount"], mode=2)
    rev, _ = arith(MUL, j["l_extendedprice"], one_minus)
    r = q_groupby([j["o_orderkey"], j["o_orderdate"], j["o_shippriority"]], [None, None, None], [("revenue", AGG_SUM, rev, None)])
    order = np.argsort(r["key_0"][0], kind="stable")
    return {"o_orderkey": r["key_0"][0][order], "o_orderdate": r["key_1"][0][order], "o_shippriority": r["key_2"][0][order],
            "revenue": r["revenue"][0][order]}


def q1_native(cols: Dict[str, np.ndarray], cutoff: int, streaming: bool = False, morsel: int = 100_000) -> Dict[str, np.ndarray]:
    """TPC-H Q1 end to end in C++ (multi-threaded per orc_set_threads), no Python between the steps.
    streaming=False: orc_q1, the in-memory FilterExec -> GroupByExec sequence (index lists per group);
    streaming=True : orc_q1_streaming, the morsel-driven partitioned group-by the reference picks for this
    shape (thread-local hot tables).  Same result layout as q1()."""
    n = len(cols["l_shipdate"])
    cap = 64
    o = {"l_returnflag": np.zeros(cap, np.uint8), "l_linestatus": np.zeros(cap, np.uint8), "sum_qty": np.zeros(cap, np.int64),
         "sum_base_price": np.zeros(cap), "sum_disc_price": np.zeros(cap), "sum_charge": np.zeros(cap), "avg_qty": np.zeros(cap),
         "avg_price": np.zeros(cap), "avg_disc": np.zeros(cap), "count_order": np.zeros(cap, np.uint32)}
    c = {k: np.ascontiguousarray(v) for k, v in cols.items()}
    f = lib().orc_q1_streaming if streaming else lib().orc_q1
    f.restype = C.c_int64
    head = [_p(c["l_shipdate"]), _p(c["l_returnflag"]), _p(c["l_linestatus"]), _p(c["l_quantity"]), _p(c["l_extendedprice"]), _p(c["l_discount"]),
            _p(c["l_tax"]), C.c_int64(n), C.c_int64(cutoff)]
    if streaming:
        head.append(C.c_int64(morsel))
    G = f(*head, cap, *[_p(o[k]) for k in o])
    assert G >= 0
    order = np.lexsort((o["l_linestatus"][:G], o["l_returnflag"][:G]))
    return {k: v[:G][order] for k, v in o.items()}


# ------------------------------------------ block-wise (partial-state) checkers for the full-size runs ----
def cfg2_partial(a: np.ndarray, x: np.ndarray, y: np.ndarray, k: int, x_valid: Optional[np.ndarray] = None, morsel: int = 100_000) -> Dict[str, object]:
    """Partial states of BASELINE config 2 over one row block (orc_cfg2_partial, multi-threaded): they ADD across blocks.
    -> {"sum_xy", "sum_x", "count_x", "sum_a" (wrapping int64), "rows"}."""
    out = np.zeros(2, np.float64)
    iout = np.zeros(3, np.int64)
    rc = lib().orc_cfg2_partial(_p(np.ascontiguousarray(a, np.int64)), _p(np.ascontiguousarray(x, np.float64)), _p(np.ascontiguousarray(y, np.float64)),
                                _p(pack(x_valid)), C.c_int64(len(a)), C.c_int64(k), C.c_int64(morsel), _p(out), _p(iout))
    assert rc == 0
    return {"sum_xy": float(out[0]), "sum_x": float(out[1]), "count_x": int(iout[0]), "sum_a": int(iout[1]), "rows": int(iout[2])}


def cfg2_combine(parts) -> Dict[str, object]:
    """Block partials -> the query's result row (x_mean = sum / count, null when count == 0; a_sum wraps like Int64)."""
    sxy = sum(p["sum_xy"] for p in parts); sx = sum(p["sum_x"] for p in parts); cx = sum(p["count_x"] for p in parts)
    sa = sum(p["sum_a"] for p in parts)
    sa = (sa + 2 ** 63) % 2 ** 64 - 2 ** 63
    return {"xy": sxy, "x_mean": (sx / cx) if cx else None, "a_sum": sa, "rows": sum(p["rows"] for p in parts)}


def groupby_dense_partial(keys: np.ndarray, vals: np.ndarray, sums: np.ndarray, counts: np.ndarray, morsel: int = 100_000) -> None:
    """group_by(key).agg(v.sum(), v.count()) partial states of one row block, ADDED into sums[n_slots] (int64 or float64)
    and counts[n_slots] (int64); keys (Int64 or UInt32) must lie in [0, n_slots) (orc_groupby_dense_partial)."""
    keys, vals = np.ascontiguousarray(keys), np.ascontiguousarray(vals)
    assert keys.dtype in (np.int64, np.uint32) and vals.dtype in (np.int64, np.float64) and sums.dtype == vals.dtype and counts.dtype == np.int64
    rc = lib().orc_groupby_dense_partial(_dt(keys), _p(keys), _dt(vals), _p(vals), C.c_int64(len(keys)), C.c_int64(len(sums)), C.c_int64(morsel), _p(sums), _p(counts))
    assert rc == 0, f"orc_groupby_dense_partial rc={rc}"


Q1_SUMS = ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "count_order")


def q1_combine(parts) -> Dict[str, np.ndarray]:
    """Q1 results of disjoint row blocks (q1 / q1_native layout) -> the result over their union: sums and counts add,
    the averages are recombined from (avg x count) -- the partial / final split of reduce/mean.rs:82-132."""
    acc: Dict[Tuple[int, int], Dict[str, float]] = {}
    for p in parts:
        for i in range(len(p["l_returnflag"])):
            m = acc.setdefault((int(p["l_returnflag"][i]), int(p["l_linestatus"][i])), {k: 0 for k in Q1_SUMS} | {"_disc": 0.0})
            c = int(p["count_order"][i])
            m["sum_qty"] += int(p["sum_qty"][i]); m["count_order"] += c
            for k in ("sum_base_price", "sum_disc_price", "sum_charge"):
                m[k] += float(p[k][i])
            m["_disc"] += float(p["avg_disc"][i]) * c
    ks = sorted(acc)
    out = {"l_returnflag": np.array([k[0] for k in ks], np.uint8), "l_linestatus": np.array([k[1] for k in ks], np.uint8)}
    for k in ("sum_qty", "count_order"):
        out[k] = np.array([acc[g][k] for g in ks], np.int64)
    for k in ("sum_base_price", "sum_disc_price", "sum_charge"):
        out[k] = np.array([acc[g][k] for g in ks], np.float64)
    cnt = out["count_order"].astype(np.float64)
    out["avg_qty"] = out["sum_qty"].astype(np.float64) / cnt
    out["avg_price"] = out["sum_base_price"] / cnt
    out["avg_disc"] = np.array([acc[g]["_disc"] for g in ks], np.float64) / cnt
    return out


# ------------------------------------------------------------ raw string keys (Utf8View) ----
def binview_parts(s: bytes):
    """(len, prefix u32, inline?) of a string's 16-byte view (crates/polars-arrow/src/array/binview/view.rs:20-29,55: strings of
    <= 12 bytes live inside the view, zero padded; longer ones keep a 4-byte prefix + buffer index + offset)."""
    n = len(s)
    return n, int.from_bytes(s[:4].ljust(4, b"\0"), "little"), n <= 12


def binview_dict_encode(strings):
    """The reference's view index map restated (crates/polars-compute/src/binview_index_map.rs; used by BinviewKeys,
    crates/polars-expr/src/hash_keys.rs:413-452): every distinct string gets a dense index in first-appearance order, equality
    is by length + bytes (inline views compare as the 16-byte value, long ones prefix first, then the bytes); nulls get no index.
    strings: list of str | bytes | None -> (codes u32 (0 for null), valid bool array or None, categories in index order)."""
    index, cats = {}, []
    codes = np.zeros(len(strings), np.uint32)
    valid = np.ones(len(strings), bool)
    for i, s in enumerate(strings):
        if s is None:
            valid[i] = False
            continue
        b = s.encode() if isinstance(s, str) else bytes(s)
        key = (binview_parts(b)[:2], b)            # (len, prefix) first, then the bytes: what the view compare does
        j = index.get(key)
        if j is None:
            j = index[key] = len(cats)
            cats.append(s)
        codes[i] = j
    return codes, (None if valid.all() else valid), cats
