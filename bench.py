#!/usr/bin/env python
"""bench.py -- TPC-H Q1 (and, as extras, Q3 / BASELINE configs 2 and 3) on MI355X.

One "step" = one complete query over device-resident columns: host plan lowering, the fused
HIP pipeline behind plx_execute_plan, and the download of the (tiny) result.  Inputs are
synthetic (polars_amd/datagen.py: dbgen distributions restated) and already sit in HBM when the
timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload q1|q3|cfg2|cfg3] [--rows R]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU.  --scaling strong (the default at N > 1): the SF100 / 1e9-row configuration in TOTAL, split over the ranks, so the
N-GPU lines sit on one curve with the N = 1 line (BASELINE: "SF100, 1/2/4/8 GPU"); --scaling weak: every rank holds its own SF100-sized shard (run
as the extra `..._weak`).  Q1's six-group partials are combined with an all-gather
of a few hundred bytes (polars_amd/dist.py), no row crosses xGMI; cfg3 / cfg5 pre-aggregate locally and exchange the PARTIAL rows
by key hash (--mode rows exchanges raw rows instead).  Prints ONE JSON line on rank 0; exits 3 if a result disagreed with the oracle.

Order of work at N = 1: headline (Q1; its input comes from the library's own generator kernel, spot-checked against the
generator's host twin, with the torch generators as fallback) -> CPU baseline -> secondary workloads (`extras`).  The
line is complete after the first two; the extras only add to it, and a guard process prints the line as it stands if
they have not finished PLX_BENCH_DEADLINE_S (420) seconds after the start (run_guarded).  After a workload was timed, the
result of its last timed step is checked against the CPU oracle over the same rows (`verified`), outside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
HBM_COPY_CEILING_GBS = 6290.0   # the same guide's measured float4 copy: nothing that streams its input can be faster
SF100_LINEITEM = 600_000_000
SF100_ORDERS = 150_000_000
CFG2_NULL_PCT = 5
HASHED_KEY_MULT = 0x9E3779B97F4A7C15 - (1 << 64)      # the odd 64-bit multiplier as an Int64 literal (wrapping multiply = a bijection on the 64-bit keys)
HASHED_KEY_INV = pow(0x9E3779B97F4A7C15, -1, 1 << 64)   # its inverse mod 2^64: maps result keys back
WIDE_KEY_MULT2 = 0xC2B2AE3D27D4EB4F - (1 << 64)         # a second odd multiplier: the other key column of the two-column key workload


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="q1", choices=["q1", "q1j", "q3", "q3f", "q3h", "q3d", "q3dc", "joinm", "joinmh", "semim", "filterm", "gather", "cfg2", "cfg2n", "cfg3", "cfg3z", "cfg3s", "cfg3w", "cfg5", "cfg5s", "cfg5l"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the SF100 / 1e9-row size of the workload)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1: strong (default) = the single-GPU configuration in TOTAL, split over the ranks (BASELINE's metric: "
                    "SF100 on 1/2/4/8 GPUs); weak = every rank holds the full single-GPU configuration (SF100 / 1e9 rows per rank)")
    ap.add_argument("--mode", default="auto", choices=["auto", "preagg", "rows"], help="sharded cfg3 / cfg5: what crosses the fabric (dist.sharded_groupby)")
    ap.add_argument("--dry-run", action="store_true", help="sharded workloads only: numpy frames + gloo instead of the library + RCCL (control-flow check without GPUs)")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if max(args.gpus, int(os.environ.get("WORLD_SIZE", "1"))) > 1 else "weak"
    return args


class Workload:
    """name, rows, algorithmic bytes per step, build(pl) -> callable step() returning a host result."""

    def __init__(self, name, rows, algo_bytes, step, kernel, desc, variants=None, verify=None, scope="kernel"):
        self.name, self.rows, self.algo_bytes, self.step, self.kernel, self.desc = name, rows, algo_bytes, step, kernel, desc
        self.variants = variants or {}   # name -> step(): the same data through a longer query (extras only)
        self.verify = verify             # verify(result of the LAST timed step, budget_s) -> {"rows", "against", "ok", ...}; outside the timed region
        self.scope = scope               # "kernel": one streaming kernel does the work; "operator": several passes -> roofline over all kernels of a step


def check_native_lineitem(pl, df, n: int, seed: int) -> None:
    """Spot check of the device table against the generator's host twin (plx_datagen_lineitem_q1_host): three blocks of
    rows, every column, bit-exact; raises on any difference."""
    import numpy as np
    from polars_amd import datagen
    if df.height != n:
        raise RuntimeError(f"generated {df.height} rows, wanted {n}")
    blk = min(4096, n)
    for row0 in sorted({0, max(0, n // 2 - 1234), n - blk}):
        got = df.slice(row0, blk)
        want = datagen.lineitem_native_host(row0, blk, seed)
        for c in datagen.LINEITEM_Q1_COLS:
            if not np.array_equal(got[c].to_numpy(), want[c]):
                raise RuntimeError(f"device generator differs from its host twin in {c} at rows [{row0}, {row0 + blk})")


def _blocks(n: int, blk: int = 4096):
    blk = min(blk, n)
    return [(r, blk) for r in sorted({0, max(0, n // 2 - 1234), n - blk})] if n else []


def native_uniform_column(pl, name: str, dtype, np_name: str, n: int, seed: int, stream: int, lo: int, hi: int, scale: float = 1.0):
    """A uniform device column from the library's generator, spot-checked against the generator's host twin."""
    import numpy as np
    from polars_amd import datagen
    s = datagen.uniform_native(pl, name, dtype, n, seed, stream, lo, hi, scale)
    df = pl.DataFrame([s])
    for row0, blk in _blocks(n):
        if not np.array_equal(df.slice(row0, blk)[name].to_numpy(), datagen.uniform_native_host(np_name, row0, blk, seed, stream, lo, hi, scale)):
            raise RuntimeError(f"device generator differs from its host twin in {name} at rows [{row0}, {row0 + blk})")
    return s


def check_native_q3(pl, O, L, n_orders: int, seed: int) -> None:
    """First and last 2048 orders (and their lines) of the device tables against the generator's host twin, bit-exact."""
    import numpy as np
    from polars_amd import datagen
    if O.height != n_orders:
        raise RuntimeError(f"generated {O.height} orders, wanted {n_orders}")
    blk = min(2048, n_orders)
    for order0 in sorted({0, n_orders - blk}):
        wo, wl, _cnt = datagen.orders_lineitem_native_host(order0, blk, n_orders, seed)
        go = O.slice(order0, blk)
        for c in datagen.ORDERS_Q3_COLS:
            if not np.array_equal(go[c].to_numpy(), wo[c]):
                raise RuntimeError(f"device generator differs from its host twin in {c} at orders [{order0}, {order0 + blk})")
        m = len(wl["l_orderkey"])
        gl = L.slice(0, m) if order0 == 0 else L.slice(L.height - m, m)
        for c in datagen.LINEITEM_Q3_COLS:
            if not np.array_equal(gl[c].to_numpy(), wl[c]):
                raise RuntimeError(f"device generator differs from its host twin in {c} (lines of orders [{order0}, {order0 + blk}))")


def _native_or_none(what: str, build):
    """build() -> inputs from the library's generators, or None (with a note on stderr) so the caller uses the torch generators."""
    if os.environ.get("PLX_BENCH_DATAGEN", "native") != "native":
        return None
    try:
        return build()
    except Exception as e:
        print(f"[bench] native data generator unavailable for {what} ({type(e).__name__}: {e}); using the torch generators", file=sys.stderr)
        return None



# ---- verification of the timed results (outside the timed region) -------------------------------------------------------
# Every workload's inputs come from the library's counter-based generators, whose host twins reproduce any row range on
# the CPU.  After a workload was timed, the result of its LAST timed step is compared with the CPU oracle evaluated over
# the same rows, block by block (the oracle's partial states add across blocks): integers bit-exact, floats 1e-6 relative.
VERIFY_RTOL = 1e-6


def _rel_err(got, want):
    import numpy as np
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    den = np.maximum(np.abs(want), 1e-300)
    return float(np.max(np.abs(got - want) / den)) if got.size else 0.0


def q1_oracle_blocks(n: int, seed: int, budget_s: float, block: int = 100_000_000):
    """Oracle Q1 (orc_q1_streaming, all host threads) over rows [0, rows_done) of the generator's host twin, in blocks;
    stops early when the budget is spent.  -> (combined result, rows_done, oracle seconds, first block's columns)."""
    from oracle import pyoracle as orc
    from polars_amd import datagen
    cutoff = datagen.us(1998, 9, 2)
    parts, done, t_orc, first = [], 0, 0.0, None
    t_start = time.perf_counter()
    while done < n:
        m = min(block, n - done)
        cols = datagen.lineitem_native_host_mt(done, m, seed)
        t0 = time.perf_counter()
        parts.append(orc.q1_native(cols, cutoff, streaming=True))
        t_orc += time.perf_counter() - t0
        if first is None:
            first = cols
        done += m
        if time.perf_counter() - t_start > budget_s:
            break
    return orc.q1_combine(parts), done, t_orc, first


def compare_q1(got: dict, want: dict) -> dict:
    """got: the library's Q1 result (to_dict: lists, flag / status as categories or codes); want: oracle layout (numpy, codes)."""
    import numpy as np
    from polars_amd import datagen
    code = lambda v, cats: cats.index(v) if isinstance(v, str) else int(v)
    order = sorted(range(len(got["l_returnflag"])), key=lambda i: (code(got["l_returnflag"][i], datagen.FLAGS), code(got["l_linestatus"][i], datagen.STATUS)))
    gk = [(code(got["l_returnflag"][i], datagen.FLAGS), code(got["l_linestatus"][i], datagen.STATUS)) for i in order]
    wk = list(zip(want["l_returnflag"].tolist(), want["l_linestatus"].tolist()))
    ok = gk == wk
    worst = 0.0
    if ok:
        for k in ("sum_qty", "count_order"):
            ok = ok and [int(got[k][i]) for i in order] == [int(v) for v in want[k]]
        for k in ("sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"):
            worst = max(worst, _rel_err([got[k][i] for i in order], want[k]))
        ok = ok and worst <= VERIFY_RTOL
    return {"ok": bool(ok), "max_rel_err": worst, "groups": len(gk)}


def verify_cfg2(got: dict, n: int, seed: int, budget_s: float, block: int = 100_000_000, null_pct: int = 0) -> dict:
    """null_pct > 0: the variant with a validity bitmap on x (row i is null when the generator's stream-3 value in [0, 100) is below null_pct)."""
    from oracle import pyoracle as orc
    from polars_amd import datagen
    parts, done, t0 = [], 0, time.perf_counter()
    while done < n and time.perf_counter() - t0 < budget_s:
        m = min(block, n - done)
        a = datagen.uniform_native_host_mt("Int64", done, m, seed, 0, 0, 2 ** 31)
        x = datagen.uniform_native_host_mt("Float64", done, m, seed, 1, 0, 10 ** 9, 1e-7)
        y = datagen.uniform_native_host_mt("Float64", done, m, seed, 2, 0, 10 ** 9, 1e-9)
        xv = (datagen.uniform_native_host_mt("UInt32", done, m, seed, 3, 0, 100) >= null_pct) if null_pct else None
        parts.append(orc.cfg2_partial(a, x, y, 2 ** 30, xv))
        done += m
    if done < n:
        return {"rows": done, "ok": None, "note": "host check ran out of its time budget before covering the input"}
    w = orc.cfg2_combine(parts)
    err = max(_rel_err(got["xy"][0], w["xy"]), _rel_err(got["x_mean"][0], w["x_mean"]))
    return {"rows": n, "against": "oracle (orc_cfg2_partial over the generator's host twin, all rows)", "ok": bool(int(got["a_sum"][0]) == w["a_sum"] and err <= VERIFY_RTOL),
            "max_rel_err": err, "rtol": VERIFY_RTOL}


def verify_groupby_dense(frame, key: str, sum_col: str, n: int, seed: int, n_keys: int, key_np: str, val_np: str, val_args, second, budget_s: float,
                         block: int = 100_000_000, key_args=None, key_shift: int = 0, key_gen=None, key_unmap=None) -> dict:
    """cfg3 / cfg5: per-key (sum, count) of the host twin through the oracle's streaming group-by (thread-local tables, combined)
    against the library's result frame.  second = ("count", name) or ("mean", name).  key_gen(row0, m) -> the dense ids in [0, n_keys) of
    rows [row0, row0 + m) (default: the uniform generator); key_unmap(result keys) -> their dense ids (keys that are a bijective image of them)."""
    import numpy as np
    from oracle import pyoracle as orc
    from polars_amd import datagen
    vdt = np.int64 if val_np == "Int64" else np.float64
    sums, counts = np.zeros(n_keys, vdt), np.zeros(n_keys, np.int64)
    done, t0 = 0, time.perf_counter()
    while done < n and time.perf_counter() - t0 < budget_s:
        m = min(block, n - done)
        k = key_gen(done, m) if key_gen else datagen.uniform_native_host_mt(key_np, done, m, seed, 0, *(key_args or (0, n_keys)))
        if key_shift:
            k = k + key_shift
        v = datagen.uniform_native_host_mt(val_np, done, m, seed, 1, *val_args)
        orc.groupby_dense_partial(k, v, sums, counts)
        done += m
    if done < n:
        return {"rows": done, "ok": None, "note": "host check ran out of its time budget before covering the input"}
    gk = frame[key].to_numpy()
    gk = np.asarray(gk).astype(np.int64)
    if key_unmap:
        gk = key_unmap(gk)
    order = np.argsort(gk, kind="stable")
    present = np.nonzero(counts)[0]
    ok = np.array_equal(gk[order], present)
    err = 0.0
    if ok:
        gs = np.asarray(frame[sum_col].to_numpy())[order]
        g2 = np.asarray(frame[second[1]].to_numpy())[order]
        if vdt is np.int64:
            ok = np.array_equal(gs.astype(np.int64), sums[present])
        else:
            err = _rel_err(gs, sums[present]); ok = err <= VERIFY_RTOL
        if second[0] == "count":
            ok = ok and np.array_equal(g2.astype(np.int64), counts[present])
        else:
            e2 = _rel_err(g2, sums[present] / counts[present]); err = max(err, e2); ok = ok and e2 <= VERIFY_RTOL
    return {"rows": n, "against": "oracle (orc_groupby_dense_partial: streaming group-by over the generator's host twin, all rows)", "ok": bool(ok),
            "max_rel_err": err, "rtol": VERIFY_RTOL, "groups": int(len(present))}


def q3_expected_block(o: dict, li: dict, cnt, date: int, seg_mod: int = 5, cust_ok=None):
    """Q3 over one self-contained block of orders and their lines (dbgen order keeps an order's lines next to each other):
    numpy restatement -> (orderkeys, orderdates, revenue) of the result groups, ascending in orderkey.  cust_ok (bool per
    custkey) = the three-table query's customer filter; otherwise the two-table stand-in o_custkey % seg_mod == 0."""
    import numpy as np
    nb = len(cnt)
    om = (o["o_orderdate"] < date) & (cust_ok[o["o_custkey"]] if cust_ok is not None else ((o["o_custkey"] % seg_mod) == 0))
    oidx = np.repeat(np.arange(nb, dtype=np.int64), cnt)
    lm = (li["l_shipdate"] > date) & om[oidx]
    rev = li["l_extendedprice"][lm] * (1.0 - li["l_discount"][lm])
    sel = oidx[lm]
    sums = np.bincount(sel, weights=rev, minlength=nb)
    has = np.bincount(sel, minlength=nb) > 0
    return o["o_orderkey"][has], o["o_orderdate"][has], sums[has]


def verify_q3(frame, n_orders: int, seed: int, budget_s: float, block: int = 8_000_000, oracle_orders: int = 4_000_000, customer_seed=None) -> dict:
    """The timed Q3 result against (a) the oracle's Q3 (reference operator order: filter, hash join, gather, group_by) on the
    first `oracle_orders` orders and (b) a numpy restatement over every block of orders the time budget allows; groups are
    compared up to the last order key covered.  customer_seed: the three-table query (customer host twin, segment BUILDING)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as orc
    from polars_amd import datagen
    date = datagen.us(1995, 3, 15)
    t0 = time.perf_counter()
    full3 = customer_seed is not None
    key_name = "o_orderkey" if full3 else "l_orderkey"
    cust = cust_ok = None
    if full3:
        ncust = datagen.n_customers_for(n_orders)
        cust = datagen.customer_native_host(0, ncust, customer_seed)
        cust_ok = np.zeros(ncust + 2, bool)
        cust_ok[cust["c_custkey"][cust["c_mktsegment"] == datagen.SEGMENTS.index("BUILDING")]] = True
    gk = frame[key_name].to_numpy().astype(np.int64)
    order = np.argsort(gk, kind="stable")
    gk, gd, gr = gk[order], frame["o_orderdate"].to_numpy().astype(np.int64)[order], frame["revenue"].to_numpy()[order]
    gp = frame["o_shippriority"].to_numpy()
    # (a) oracle on a prefix
    no = min(oracle_orders, n_orders)
    o, li, cnt = datagen.orders_lineitem_native_host_mt(0, no, n_orders, seed)
    o["o_shippriority"] = np.zeros(no, np.int64)
    if full3:
        w = orc.q3_full(cust, {k: o[k] for k in datagen.ORDERS_Q3_COLS}, {k: li[k] for k in datagen.LINEITEM_Q3_COLS}, date, datagen.SEGMENTS.index("BUILDING"))
    else:
        w = orc.q3({k: li[k] for k in datagen.LINEITEM_Q3_COLS}, {k: o[k] for k in datagen.ORDERS_Q3_COLS}, date)
    hi = int(o["o_orderkey"][-1])
    m = gk <= hi
    ok_oracle = bool(np.array_equal(gk[m], w[key_name]) and np.array_equal(gd[m], w["o_orderdate"]) and _rel_err(gr[m], w["revenue"]) <= VERIFY_RTOL)
    rk, rd, rr = q3_expected_block(o, li, cnt, date, cust_ok=cust_ok)
    ok_oracle = ok_oracle and bool(np.array_equal(rk, w[key_name]) and _rel_err(rr, w["revenue"]) <= 1e-12)   # the restatement agrees with the oracle
    lines_oracle = len(li["l_orderkey"])
    del o, li, cnt, w
    # (b) numpy restatement over all blocks
    blocks = [(b, min(block, n_orders - b)) for b in range(0, n_orders, block)]
    keys, dates, revs, lines, done_orders = [], [], [], 0, 0

    def work(bl):
        o, li, cnt = datagen.orders_lineitem_native_host_mt(bl[0], bl[1], n_orders, seed, threads=16)
        return q3_expected_block(o, li, cnt, date, cust_ok=cust_ok), len(li["l_orderkey"]), int(o["o_orderkey"][-1])
    last_key = -1
    with ThreadPoolExecutor(4) as ex:
        for i in range(0, len(blocks), 4):
            if time.perf_counter() - t0 > budget_s:
                break
            for (k, d, r), nl, lk in ex.map(work, blocks[i:i + 4]):
                keys.append(k); dates.append(d); revs.append(r); lines += nl; last_key = lk
            done_orders = sum(b[1] for b in blocks[:i + 4])
    wk, wd, wr = (np.concatenate(x) if x else np.zeros(0) for x in (keys, dates, revs))
    m = gk <= last_key
    err = _rel_err(gr[m], wr) if int(m.sum()) == len(wr) else float("inf")
    ok = bool(int(m.sum()) == len(wk) and np.array_equal(gk[m], wk) and np.array_equal(gd[m], wd) and err <= VERIFY_RTOL and not np.any(gp))
    full = done_orders >= n_orders
    return {"rows": int(done_orders + lines), "orders": int(done_orders), "lineitem_rows": int(lines), "covers_whole_input": bool(full),
            "against": f"oracle Q3 ({'customer x orders x lineitem' if full3 else 'orders x lineitem'}: filter -> hash join(s) -> gather -> group_by) on the first {no} orders / {lines_oracle} lines + numpy restatement over "
                       f"{'all' if full else done_orders} orders of the generator's host twin",
            "ok": bool(ok and ok_oracle and (full or done_orders > 0)), "ok_oracle_prefix": ok_oracle, "max_rel_err": err, "rtol": VERIFY_RTOL, "groups_checked": int(len(wk)),
            "groups_total": int(len(gk))}


def make_workload(pl, name: str, rows: int, seed: int, ws: int = 1) -> Workload:
    import numpy as np
    import torch
    from polars_amd import datagen, queries
    if name == "q1":
        n = rows or SF100_LINEITEM
        df, cols, gen = None, None, "library kernel (plx_datagen_lineitem_q1)"
        if os.environ.get("PLX_BENCH_DATAGEN", "native") == "native":
            # the library's own generator: 25 GB written straight into HBM columns, no torch kernels on the headline path
            try:
                df = datagen.lineitem_native(pl, n, seed)
                check_native_lineitem(pl, df, n, seed)
            except Exception as e:   # fall back to the torch generators rather than lose the measurement
                print(f"[bench] native data generator unavailable ({type(e).__name__}: {e}); using the torch generators", file=sys.stderr)
                df = None
        if df is None:
            gen = "torch generators (polars_amd/datagen.py lineitem_device)"
            cols = datagen.lineitem_device(n, seed=seed)
            df = datagen.frame_from_torch(pl, cols, datagen.LINEITEM_Q1_COLS)
            torch.cuda.synchronize()
        lf = queries.q1(df.lazy())

        def step():
            out = lf.collect()
            return out.to_dict(), (df, cols)
        lf_sorted = queries.q1_sorted(df.lazy())

        def step_sorted():
            out = lf_sorted.collect()
            return out.to_dict(), (df, cols)
        wl = Workload("tpch_q1_sf100", n, n * datagen.Q1_BYTES_PER_ROW, step, "fused_scan_ldsagg_static",
                      f"TPC-H Q1, lineitem {n} rows x 42 B (SF100 = 6.0e8), filter -> 2-key group_by -> 8 aggregates; input: {gen}",
                      variants={"tpch_q1_sf100_order_by": step_sorted})
        wl.native_seed = seed if cols is None else None      # host twin available: the timed result can be checked against the oracle
        wl.frame = df
        wl.inputs = [df]
        if cols is None:
            def verify(res, budget):
                want, done, _t, _first = q1_oracle_blocks(n, seed, budget, block=min(n, 100_000_000))
                if done < n:
                    return {"rows": done, "ok": None, "note": "host check ran out of its time budget before covering the input"}
                return dict(compare_q1(res, want), rows=n, rtol=VERIFY_RTOL, against="oracle (orc_q1_streaming over the generator's host twin, all rows of the timed input)")
            wl.verify = verify
        return wl
    if name in ("q3", "q3h"):
        no = (rows // 4) if rows else SF100_ORDERS
        shuffled = os.environ.get("PLX_Q3_SHUFFLED", "0") == "1"
        hashed = name == "q3h"
        orders = li = None

        def build_native():
            O_, L_ = datagen.orders_lineitem_native(pl, no, seed)
            check_native_q3(pl, O_, L_, no, seed)
            return O_, L_
        nat = _native_or_none("q3", build_native)
        if nat is not None and shuffled:
            # the SAME rows as the ordered run (the library's generator, host twin available: the oracle can check the result) in a seeded
            # random row order, both tables: one device gather per column
            def permuted(df, gseed):
                g = torch.Generator(device="cuda"); g.manual_seed(gseed)
                perm = pl.Series.from_torch("perm", torch.randperm(df.height, generator=g, device="cuda", dtype=torch.int32), dtype=pl.UInt32)
                torch.cuda.synchronize()
                out = pl.DataFrame([df[c].gather(perm) for c in df.columns])
                pl._ffi.check(pl._ffi.lib().plx_synchronize())
                return out
            nat = (permuted(nat[0], seed * 2 + 1), permuted(nat[1], seed * 2 + 2))
        if nat is not None and hashed:
            # the same rows with orderkey * 0x9E3779B97F4A7C15 mod 2^64 on BOTH sides (a bijection: the join and its groups are unchanged, the result
            # keys map back through the inverse multiplier): no key range worth learning, no dense id -- the join has to hash
            O_, L_ = nat
            O_ = O_.with_columns((pl.col("o_orderkey") * HASHED_KEY_MULT).alias("o_orderkey"))
            L_ = L_.with_columns((pl.col("l_orderkey") * HASHED_KEY_MULT).alias("l_orderkey"))
            pl._ffi.check(pl._ffi.lib().plx_synchronize())
            nat = (O_, L_)
        if nat is not None:
            O, L = nat
            nl = L.height
        else:
            orders, li = datagen.orders_lineitem_device(no, seed=seed, ordered=not shuffled)   # dbgen row order by default
            L = datagen.frame_from_torch(pl, li, datagen.LINEITEM_Q3_COLS)
            O = datagen.frame_from_torch(pl, orders, datagen.ORDERS_Q3_COLS)
            torch.cuda.synchronize()
            nl = li["l_orderkey"].numel()
        lf = queries.q3(L.lazy(), O.lazy())

        def step():
            out = lf.collect()
            return out, (L, O, li, orders)
        lf_top = queries.q3_top10(L.lazy(), O.lazy())

        def step_top10():
            out = lf_top.collect()
            return out.to_dict(), (L, O, li, orders)
        verify = (lambda res, budget: verify_q3(res, no, seed, budget)) if nat is not None else None
        if hashed:
            if nat is None:
                raise RuntimeError("the hashed-key Q3 needs the library's generator")

            class Unhashed:       # the result with its keys mapped back through the inverse multiplier (host side, outside the timed region)
                def __init__(self, res): self.res = res
                def __getitem__(self, c):
                    import numpy as np
                    col = self.res[c]
                    if c != "l_orderkey":
                        return col
                    a = (col.to_numpy().astype(np.uint64) * np.uint64(HASHED_KEY_INV)).astype(np.int64)
                    return type("H", (), {"to_numpy": lambda self_: a})()
            wlh = Workload("tpch_q3_sf100_hashed_keys", nl + no, nl * datagen.Q3_LINEITEM_BYTES_PER_ROW + no * datagen.Q3_ORDERS_BYTES_PER_ROW, step, "probe_scatter",
                           f"TPC-H Q3 (orders {no} x lineitem {nl}) with orderkey * 0x9E3779B97F4A7C15 mod 2^64 on both sides: 64-bit keys without a learnable range "
                           "(radix-partitioned hash probe against LDS filters + hash table)", verify=lambda res, budget: verify_q3(Unhashed(res), no, seed, budget), scope="operator")
            wlh.inputs = [L, O]
            return wlh
        wl3 = Workload("tpch_q3_sf100", nl + no, nl * datagen.Q3_LINEITEM_BYTES_PER_ROW + no * datagen.Q3_ORDERS_BYTES_PER_ROW, step, "join_probe_emit",
                       f"TPC-H Q3 (orders {no} x lineitem {nl}), filter both -> hash join -> group_by(orderkey, orderdate, shippriority)",
                       variants={"tpch_q3_sf100_order_by_limit10": step_top10}, verify=verify, scope="operator")
        wl3.inputs = [L, O]
        return wl3
    if name == "q1j":
        # A shape WITHOUT an ahead-of-time kernel (round-5 review, weak 7 / item 8): TPC-H Q1 with a second predicate and a ninth and tenth aggregate -- not in
        # fused_shapes.hpp, so the scan is specialised at run time (hiprtc; the kernel's symbol is plx_jit_LdsAggSink_1_<hash>) exactly like any query a drop-in
        # executor meets first.  cold_first_step_ms includes the compilation when the disk cache does not hold it (a fresh box: always); `jit` reports what was compiled.
        n = rows or SF100_LINEITEM
        dfj = datagen.lineitem_native(pl, n, seed)
        check_native_lineitem(pl, dfj, n, seed)
        c_ = pl.col
        cutoff = datagen.us(1998, 9, 2)
        disc_price = c_("l_extendedprice") * (1 - c_("l_discount"))
        lfj = (dfj.lazy().filter((c_("l_shipdate") <= cutoff) & (c_("l_quantity") < 45)).group_by("l_returnflag", "l_linestatus")
               .agg(c_("l_quantity").sum().alias("sum_qty"), c_("l_extendedprice").sum().alias("sum_base_price"), disc_price.sum().alias("sum_disc_price"),
                    (disc_price * (1 + c_("l_tax"))).sum().alias("sum_charge"), c_("l_quantity").mean().alias("avg_qty"), c_("l_extendedprice").mean().alias("avg_price"),
                    c_("l_discount").mean().alias("avg_disc"), pl.len().alias("count_order"), c_("l_tax").max().alias("max_tax"), c_("l_extendedprice").min().alias("min_price")))
        jit0 = jit_stats(pl)

        def step_j():
            out = lfj.collect()
            return out.to_dict(), (dfj,)

        def verify_j(res, budget):
            # numpy restatement over every block of the generator's host twin; the restatement itself is pinned on the first block: WITHOUT the extra predicate and the
            # extra aggregates it must reproduce the oracle's Q1 (orc.q1_native) on those rows
            from oracle import pyoracle as orc
            t0 = time.perf_counter()
            G = 6
            acc = {k: np.zeros(G) for k in ("sum_base_price", "sum_disc_price", "sum_charge", "sum_disc")}
            qty, cnt = np.zeros(G, np.int64), np.zeros(G, np.int64)
            mx, mn = np.full(G, -np.inf), np.full(G, np.inf)
            done, pinned = 0, None
            while done < n and time.perf_counter() - t0 < budget:
                m = min(100_000_000, n - done)
                cols = datagen.lineitem_native_host_mt(done, m, seed)
                gid = cols["l_returnflag"].astype(np.int64) * 2 + cols["l_linestatus"].astype(np.int64)
                base = cols["l_shipdate"] <= cutoff
                if pinned is None:
                    w = orc.q1_native(cols, cutoff, streaming=True)
                    ids = np.asarray(w["l_returnflag"], np.int64) * 2 + np.asarray(w["l_linestatus"], np.int64)
                    dp = cols["l_extendedprice"] * (1.0 - cols["l_discount"])
                    pinned = bool(np.array_equal(np.bincount(gid[base], minlength=G)[ids], np.asarray(w["count_order"], np.int64))
                                  and _rel_err(np.bincount(gid[base], weights=dp[base], minlength=G)[ids], w["sum_disc_price"]) <= 1e-12)
                keep = base & (cols["l_quantity"] < 45)
                g = gid[keep]
                ep, dc, tx = cols["l_extendedprice"][keep], cols["l_discount"][keep], cols["l_tax"][keep]
                dp = ep * (1.0 - dc)
                acc["sum_base_price"] += np.bincount(g, weights=ep, minlength=G); acc["sum_disc_price"] += np.bincount(g, weights=dp, minlength=G)
                acc["sum_charge"] += np.bincount(g, weights=dp * (1.0 + tx), minlength=G); acc["sum_disc"] += np.bincount(g, weights=dc, minlength=G)
                qty += np.bincount(g, weights=cols["l_quantity"][keep], minlength=G).astype(np.int64); cnt += np.bincount(g, minlength=G)
                np.maximum.at(mx, g, tx); np.minimum.at(mn, g, ep)
                done += m
            if done < n:
                return {"rows": done, "ok": None, "note": "host check ran out of its time budget before covering the input"}
            code = lambda v, cats: cats.index(v) if isinstance(v, str) else int(v)
            ids = [code(a, datagen.FLAGS) * 2 + code(b, datagen.STATUS) for a, b in zip(res["l_returnflag"], res["l_linestatus"])]
            live = np.nonzero(cnt)[0]
            ok = sorted(ids) == live.tolist()
            err = 0.0
            if ok:
                ix = np.array(ids)
                ok = [int(v) for v in res["sum_qty"]] == qty[ix].tolist() and [int(v) for v in res["count_order"]] == cnt[ix].tolist()
                ok = ok and bool(np.array_equal(np.array(res["max_tax"]), mx[ix]) and np.array_equal(np.array(res["min_price"]), mn[ix]))       # min / max: bit-exact
                for k, want in (("sum_base_price", acc["sum_base_price"]), ("sum_disc_price", acc["sum_disc_price"]), ("sum_charge", acc["sum_charge"]),
                                ("avg_qty", qty / np.maximum(cnt, 1)), ("avg_price", acc["sum_base_price"] / np.maximum(cnt, 1)), ("avg_disc", acc["sum_disc"] / np.maximum(cnt, 1))):
                    err = max(err, _rel_err(res[k], want[ix]))
                ok = ok and err <= VERIFY_RTOL
            return {"rows": n, "ok": bool(ok and pinned), "restatement_matches_oracle_q1_on_first_block": pinned, "max_rel_err": err, "rtol": VERIFY_RTOL, "groups": len(ids),
                    "against": "numpy restatement over all rows of the generator's host twin (pinned against the oracle's Q1 on the first block)"}
        wlj = Workload("tpch_q1_sf100_two_predicates_ten_aggregates_jit", n, n * datagen.Q1_BYTES_PER_ROW, step_j, "fused_scan_ldsagg_generic",
                       f"TPC-H Q1 over {n} rows with a second predicate (l_quantity < 45) and two more aggregates (max(l_tax), min(l_extendedprice)): no ahead-of-time kernel, "
                       "the scan is specialised at run time (hiprtc)", verify=verify_j)
        wlj.inputs = [dfj]
        wlj.jit_before = jit0
        return wlj
    if name == "q3dc":
        # A join -> group-by whose AGGREGATE reads a build-side column (round-5 review, missing 2): the duplicate-key join's tables with a supplier cost on the build side,
        # group_by(partkey, suppkey).agg(sum(l_extendedprice * ps_supplycost), len) -- the in-place form cannot evaluate it (its cells are fed by the probe scan alone);
        # the pair form joins (probe row, build row) pairs, gathers the two referenced columns per side at them and runs the fused group-by over the joined columns.
        nl = rows or SF100_LINEITEM
        nb = max(nl * 2 // 15, 1000)
        n_parts = max(nb // 4, 100)
        lo_d, hi_d, date = datagen.us(1992, 1, 2), datagen.us(1998, 12, 1), datagen.us(1995, 3, 15)
        L = pl.DataFrame([native_uniform_column(pl, "l_partkey", pl.Int64, "Int64", nl, seed, 0, 0, n_parts),
                          native_uniform_column(pl, "l_extendedprice", pl.Float64, "Float64", nl, seed, 1, 90_000, 10_500_000, 0.01),
                          native_uniform_column(pl, "l_shipdate", pl.Datetime, "Int64", nl, seed, 3, lo_d, hi_d)])
        PS = pl.DataFrame([native_uniform_column(pl, "ps_partkey", pl.Int64, "Int64", nb, seed + 1000, 0, 0, n_parts),
                           native_uniform_column(pl, "ps_suppkey", pl.Int64, "Int64", nb, seed + 1000, 1, 0, 1_000_000),
                           native_uniform_column(pl, "ps_supplycost", pl.Float64, "Float64", nb, seed + 1000, 2, 100, 100_000, 0.01)])
        PS = PS.with_columns((pl.col("ps_partkey") % 32).alias("ps_group"))
        PS = pl.DataFrame([PS["ps_partkey"], PS["ps_suppkey"], PS["ps_supplycost"], PS["ps_group"]])
        pl._ffi.check(pl._ffi.lib().plx_synchronize())
        c_ = pl.col
        lfc = (L.lazy().filter(c_("l_shipdate") > date).join(PS.lazy().filter(c_("ps_group") == 5), left_on="l_partkey", right_on="ps_partkey")
               .group_by("l_partkey", "ps_suppkey").agg((c_("l_extendedprice") * c_("ps_supplycost")).sum().alias("cost"), pl.len().alias("n")))

        def step_c():
            return lfc.collect(), (L, PS)

        def verify_c(res, budget):
            from oracle import pyoracle as orc
            t0 = time.perf_counter()
            S, Cn = np.zeros(n_parts, np.float64), np.zeros(n_parts, np.int64)
            done = 0
            while done < nl and time.perf_counter() - t0 < budget:
                m = min(100_000_000, nl - done)
                k = datagen.uniform_native_host_mt("Int64", done, m, seed, 0, 0, n_parts)
                keep = (datagen.uniform_native_host_mt("Int64", done, m, seed, 3, lo_d, hi_d) > date) & (k % 32 == 5)
                orc.groupby_dense_partial(np.ascontiguousarray(k[keep]), np.ascontiguousarray(datagen.uniform_native_host_mt("Float64", done, m, seed, 1, 90_000, 10_500_000, 0.01)[keep]), S, Cn)
                done += m
            if done < nl:
                return {"rows": done, "ok": None, "note": "host check ran out of its time budget before covering the input"}
            bk = datagen.uniform_native_host_mt("Int64", 0, nb, seed + 1000, 0, 0, n_parts)
            bs = datagen.uniform_native_host_mt("Int64", 0, nb, seed + 1000, 1, 0, 1_000_000)
            bc = datagen.uniform_native_host_mt("Float64", 0, nb, seed + 1000, 2, 100, 100_000, 0.01)
            keep = (bk % 32 == 5)
            bk, bs, bc = bk[keep], bs[keep], bc[keep]
            pairs, inv, mult = np.unique(bk * 1_000_000 + bs, return_inverse=True, return_counts=True)
            csum = np.bincount(inv, weights=bc, minlength=len(pairs))           # the costs of the build rows that are ONE (partkey, suppkey) group
            pk = pairs // 1_000_000
            live = Cn[pk] > 0
            pairs, mult, pk, csum = pairs[live], mult[live], pk[live], csum[live]
            gk = res["l_partkey"].to_numpy().astype(np.int64) * 1_000_000 + res["ps_suppkey"].to_numpy().astype(np.int64)
            order = np.argsort(gk, kind="stable")
            ok = bool(np.array_equal(gk[order], pairs))
            err = 0.0
            if ok:
                ok = bool(np.array_equal(res["n"].to_numpy().astype(np.int64)[order], mult * Cn[pk]))
                err = _rel_err(res["cost"].to_numpy()[order], csum * S[pk])
                ok = ok and err <= VERIFY_RTOL
            return {"rows": nl + nb, "against": "host twin of both tables: per-part sum of l_extendedprice over the filtered probe side (oracle streaming group-by) times the summed "
                    "ps_supplycost of every (partkey, suppkey) build group", "ok": ok, "max_rel_err": err, "rtol": VERIFY_RTOL, "groups": int(len(pairs))}
        wlc = Workload("join_aggregate_reads_build_side_sf100", nl + nb, nl * 24 + nb * 32, step_c, "probe_scatter",
                       f"lineitem-shaped probe side ({nl} rows) JOIN partsupp-shaped build side ({nb} rows, duplicate keys), filter both, "
                       "group_by(partkey, suppkey).agg(sum(l_extendedprice * ps_supplycost), len): the aggregate reads a BUILD-side column (pair form)", verify=verify_c, scope="operator")
        wlc.inputs = [L, PS]
        return wlc
    if name in ("joinm", "joinmh"):
        # The MATERIALISING join (round-5 review, item 1): TPC-H Q3's two filtered tables joined into a FRAME -- no group-by above the join -- five output columns
        # (the reference: JoinExec -> _inner_join_from_series, crates/polars-ops/src/frame/join/mod.rs:564-652: pairs, then gathers).  Same rows as tpch_q3_sf100.
        no = (rows // 4) if rows else SF100_ORDERS
        O, L = datagen.orders_lineitem_native(pl, no, seed)
        check_native_q3(pl, O, L, no, seed)
        hashed_m = name == "joinmh"
        if hashed_m:
            # the same rows with orderkey * 0x9E3779B97F4A7C15 mod 2^64 on both sides (as in tpch_q3_sf100_hashed_keys): no key range, the join has to hash --
            # 16-byte-slot table, radix-partitioned probe against LDS filters, candidates looked up once
            O = O.with_columns((pl.col("o_orderkey") * HASHED_KEY_MULT).alias("o_orderkey"))
            L = L.with_columns((pl.col("l_orderkey") * HASHED_KEY_MULT).alias("l_orderkey"))
            pl._ffi.check(pl._ffi.lib().plx_synchronize())
        nl = L.height
        lfm = queries.q3_join_frame(L.lazy(), O.lazy())

        def step_m():
            out = lfm.collect()
            return out, (L, O)

        def verify_m(res, budget):
            # the Q3 aggregate of the joined rows, taken on the host, through Q3's own all-rows check: a missing / duplicated / mismatched pair changes its order's
            # revenue by a seventh or more; the build-side columns must be constant within an order
            k = res["l_orderkey"].to_numpy().astype(np.int64)
            if hashed_m:
                k = (k.astype(np.uint64) * np.uint64(HASHED_KEY_INV)).astype(np.int64)       # the keys mapped back through the inverse multiplier
            order = np.argsort(k, kind="stable")
            k = k[order]
            od = res["o_orderdate"].to_numpy().astype(np.int64)[order]
            sp = res["o_shippriority"].to_numpy()[order]
            rev = (res["l_extendedprice"].to_numpy() * (1.0 - res["l_discount"].to_numpy()))[order]
            first = np.ones(len(k), bool)
            first[1:] = k[1:] != k[:-1]
            starts = np.nonzero(first)[0]
            consistent = bool(np.array_equal(od, np.repeat(od[starts], np.diff(np.append(starts, len(k))))))

            class Col:
                def __init__(self, a): self.a = a
                def to_numpy(self): return self.a
            frame = {"l_orderkey": Col(k[starts]), "o_orderdate": Col(od[starts]), "o_shippriority": Col(sp[starts]), "revenue": Col(np.add.reduceat(rev, starts) if len(k) else np.zeros(0))}
            out = verify_q3(frame, no, seed, budget)
            out["joined_rows"] = int(len(k))
            out["build_columns_constant_within_an_order"] = consistent
            out["against"] = "the joined frame aggregated on the host (group by orderkey: revenue, orderdate, shippriority) through Q3's check: " + out.get("against", "")
            if out.get("ok") is not None:
                out["ok"] = bool(out["ok"]) and consistent
            return out
        wlm = Workload("join_materialise_sf100_hashed_keys" if hashed_m else "join_materialise_sf100", nl + no, nl * datagen.Q3_LINEITEM_BYTES_PER_ROW + no * datagen.Q3_ORDERS_BYTES_PER_ROW, step_m, "probe_scatter",
                       f"TPC-H Q3's two filtered tables (orders {no} x lineitem {nl}) joined into a frame: filter both -> hash join -> 5 output columns (no group-by); "
                       "algorithmic bytes = the input columns once (+ the joined rows, added from the result)", verify=verify_m, scope="operator")
        wlm.inputs = [L, O]
        wlm.out_row_bytes = 8 + 8 + 8 + 8 + 8
        return wlm
    if name == "semim":
        # SEMI JOIN -> FRAME: the lineitem side of Q3's join (rows whose order survives the orders predicate), all four lineitem columns out, left order.  The reference:
        # a hash set of the right keys, one lookup per left row (single_keys_semi_anti.rs), then a gather of the left columns.  Here: the right side -> membership bitmap,
        # the left side filtered by predicate AND bitmap test in one program (engine.cpp fused_semi_anti_frame).
        no = (rows // 4) if rows else SF100_ORDERS
        O, L = datagen.orders_lineitem_native(pl, no, seed)
        check_native_q3(pl, O, L, no, seed)
        nl = L.height
        lfs = queries.q3_semi_frame(L.lazy(), O.lazy())

        def step_semi():
            return lfs.collect(), (L, O)

        def verify_semi(res, budget):
            # orders' keys are unique, so the semi join's rows ARE the inner join's (join_materialise_sf100: verified through Q3's all-rows check in its own extra),
            # projected on the lineitem columns: compared as sorted row sets; the semi join must also keep the left order (dbgen order: non-decreasing order keys)
            inner = queries.q3_join_frame(L.lazy(), O.lazy()).collect()
            k = res["l_orderkey"].to_numpy(); pr = res["l_extendedprice"].to_numpy(); di = res["l_discount"].to_numpy()
            ik = inner["l_orderkey"].to_numpy(); ip = inner["l_extendedprice"].to_numpy(); idc = inner["l_discount"].to_numpy()
            same_n = len(k) == len(ik)
            o1, o2 = np.lexsort((di, pr, k)), np.lexsort((idc, ip, ik))
            same = same_n and bool(np.array_equal(k[o1], ik[o2]) and np.array_equal(pr[o1].view(np.int64), ip[o2].view(np.int64)) and np.array_equal(di[o1].view(np.int64), idc[o2].view(np.int64)))
            in_order = bool(np.all(k[1:] >= k[:-1])) if len(k) else True
            date = datagen.us(1995, 3, 15)
            ship_ok = bool(np.all(res["l_shipdate"].to_numpy().astype(np.int64) > date)) if len(k) else True
            return {"ok": bool(same and in_order and ship_ok), "rows": int(len(k)), "inner_join_rows": int(len(ik)), "same_rows_as_the_inner_join": same, "left_order_kept": in_order,
                    "covers_whole_input": True, "against": "the inner join of the same tables (join_materialise_sf100, itself checked through Q3's all-rows oracle check) projected on the lineitem columns, "
                    "as sorted row sets; left order; the left predicate on every output row"}
        wls = Workload("semi_join_materialise_sf100", nl + no, nl * 16 + no * 24, step_semi, "fused_scan",
                       f"TPC-H Q3's tables as a SEMI join (lineitem {nl} rows filtered, against filtered orders {no}): right side -> membership bitmap, left side filtered by predicate AND "
                       "bitmap test, four lineitem columns out; algorithmic bytes = the columns the predicates read, once (+ the output rows, in and out)", verify=verify_semi, scope="operator")
        wls.inputs = [L, O]
        wls.out_row_bytes = 2 * 32
        return wls
    if name == "filterm":
        # FILTER -> FRAME (round-5 review, item 1b): config 2's frame, filter(a > 2^30) -> all three columns out (~half of the rows): 24 GB in + ~12 GB out.
        # The reference: FilterExec (filter.rs:94-145) -> a mask, then filter/mod.rs:18-110 per column.  The result stays in HBM (a 12 GB frame: the next operator's input).
        n = rows or 1_000_000_000
        dff = pl.DataFrame([native_uniform_column(pl, "a", pl.Int64, "Int64", n, seed, 0, 0, 2 ** 31),
                            native_uniform_column(pl, "x", pl.Float64, "Float64", n, seed, 1, 0, 10 ** 9, 1e-7),
                            native_uniform_column(pl, "y", pl.Float64, "Float64", n, seed, 2, 0, 10 ** 9, 1e-9)])
        lff = dff.lazy().filter(pl.col("a") > (1 << 30))

        def step_f():
            return lff.collect(), (dff,)

        def verify_f(res, budget):
            # every kept row, in order, against the generator's host twin: block by block
            t0 = time.perf_counter()
            done = off = 0
            ok = True
            while done < n and ok and time.perf_counter() - t0 < budget:
                m = min(50_000_000, n - done)
                a = datagen.uniform_native_host_mt("Int64", done, m, seed, 0, 0, 2 ** 31)
                keep = a > (1 << 30)
                cnt = int(keep.sum())
                got = res.slice(off, cnt)
                ok = ok and off + cnt <= res.height and bool(np.array_equal(got["a"].to_numpy(), a[keep]))
                ok = ok and bool(np.array_equal(got["x"].to_numpy(), datagen.uniform_native_host_mt("Float64", done, m, seed, 1, 0, 10 ** 9, 1e-7)[keep]))
                ok = ok and bool(np.array_equal(got["y"].to_numpy(), datagen.uniform_native_host_mt("Float64", done, m, seed, 2, 0, 10 ** 9, 1e-9)[keep]))
                done += m; off += cnt
            full = done >= n
            return {"rows": done, "kept_rows": int(off), "covers_whole_input": full, "ok": (bool(ok and off == res.height) if full else (None if ok else False)),
                    "against": "the generator's host twin filtered with numpy, every kept row of all three columns compared bit for bit and in order"}
        wlf = Workload("filter_materialise_1e9", n, n * 24, step_f, "fused_filter_compact", f"filter -> frame: config 2's {n}-row frame, filter(a > 2^30), three columns out "
                       "(algorithmic bytes = 24 B/row in + the kept rows out, added from the result)", verify=verify_f)
        wlf.inputs = [dff]
        wlf.out_row_bytes = 24
        wlf.big_result = True
        return wlf
    if name == "gather":
        # GATHER by index (idx N x 4 B, N random 8-byte reads, N x 8 B written): 1e9 uniformly random u32 indices into a 1e9-row Int64 column
        n = rows or 1_000_000_000
        vals = native_uniform_column(pl, "v", pl.Int64, "Int64", n, seed, 0, -(1 << 40), 1 << 40)
        idx = native_uniform_column(pl, "i", pl.UInt32, "UInt32", n, seed, 1, 0, n)
        pl._ffi.check(pl._ffi.lib().plx_synchronize())

        def step_g():
            return vals.gather(idx), (vals, idx)

        def verify_g(res, budget):
            t0 = time.perf_counter()
            V = datagen.uniform_native_host_mt("Int64", 0, n, seed, 0, -(1 << 40), 1 << 40)
            done, ok = 0, True
            frame = pl.DataFrame([res])
            while done < n and ok and time.perf_counter() - t0 < budget:
                m = min(100_000_000, n - done)
                ix = datagen.uniform_native_host_mt("UInt32", done, m, seed, 1, 0, n)
                ok = bool(np.array_equal(frame.slice(done, m)["v"].to_numpy(), V[ix]))
                done += m
            return {"rows": done, "covers_whole_input": done >= n, "ok": (bool(ok) if done >= n else (None if ok else False)),
                    "against": "numpy take over the generator's host twin (values and indices), every output row"}
        wlg = Workload("gather_1e9", n, n * 20, step_g, "gather_u32", f"gather: {n} uniformly random u32 indices into a {n}-row Int64 column (4 B index + 8 B random read + 8 B written per row)",
                       verify=verify_g)
        wlg.inputs = [vals, idx]
        wlg.big_result = True
        return wlg
    if name == "q3d":
        # A join whose BUILD side repeats its keys (round-4 review, Missing 2): lineitem-shaped probe side (SF100: 6e8 rows) JOIN a partsupp-shaped build side (8e7 rows over
        # 2e7 parts: ~4 rows per part, 0..12), both filtered, grouped by (partkey, suppkey): a group is a build ROW and every probe row of a part adds to each of the part's
        # build rows (the reference: hash table key -> list of rows, single_keys.rs:16-167; one joined row per entry, single_keys_inner.rs:11-38).  All columns from the
        # library's uniform generator (host twins: the result is checked over all rows).
        nl = rows or SF100_LINEITEM
        nb = max(nl * 2 // 15, 1000)              # 8e7 at SF100
        n_parts = max(nb // 4, 100)
        lo_d, hi_d, date = datagen.us(1992, 1, 2), datagen.us(1998, 12, 1), datagen.us(1995, 3, 15)
        L = pl.DataFrame([native_uniform_column(pl, "l_partkey", pl.Int64, "Int64", nl, seed, 0, 0, n_parts),
                          native_uniform_column(pl, "l_extendedprice", pl.Float64, "Float64", nl, seed, 1, 90_000, 10_500_000, 0.01),
                          native_uniform_column(pl, "l_discount", pl.Float64, "Float64", nl, seed, 2, 0, 11, 0.01),
                          native_uniform_column(pl, "l_shipdate", pl.Datetime, "Int64", nl, seed, 3, lo_d, hi_d)])
        PS = pl.DataFrame([native_uniform_column(pl, "ps_partkey", pl.Int64, "Int64", nb, seed + 1000, 0, 0, n_parts),
                           native_uniform_column(pl, "ps_suppkey", pl.Int64, "Int64", nb, seed + 1000, 1, 0, 1_000_000)])
        PS = PS.with_columns((pl.col("ps_partkey") % 32).alias("ps_group"))          # a property of the part, the same in all of its rows
        PS = pl.DataFrame([PS["ps_partkey"], PS["ps_suppkey"], PS["ps_group"]])
        pl._ffi.check(pl._ffi.lib().plx_synchronize())
        lfd = queries.q3_partsupp(L.lazy(), PS.lazy())

        def step_d():
            return lfd.collect(), (L, PS)

        def verify_d(res, budget):
            from oracle import pyoracle as orc
            t0 = time.perf_counter()
            S, Cn = np.zeros(n_parts, np.float64), np.zeros(n_parts, np.int64)
            done = 0
            while done < nl and time.perf_counter() - t0 < budget:
                m = min(100_000_000, nl - done)
                k = datagen.uniform_native_host_mt("Int64", done, m, seed, 0, 0, n_parts)
                keep = (datagen.uniform_native_host_mt("Int64", done, m, seed, 3, lo_d, hi_d) > date) & (k % 32 == 5)
                k = k[keep]
                rev = datagen.uniform_native_host_mt("Float64", done, m, seed, 1, 90_000, 10_500_000, 0.01)[keep] * (1.0 - datagen.uniform_native_host_mt("Float64", done, m, seed, 2, 0, 11, 0.01)[keep])
                orc.groupby_dense_partial(np.ascontiguousarray(k), np.ascontiguousarray(rev), S, Cn)
                done += m
            if done < nl:
                return {"rows": done, "ok": None, "note": "host check ran out of its time budget before covering the input"}
            bk = datagen.uniform_native_host_mt("Int64", 0, nb, seed + 1000, 0, 0, n_parts)
            bs = datagen.uniform_native_host_mt("Int64", 0, nb, seed + 1000, 1, 0, 1_000_000)
            keep = (bk % 32 == 5)
            bk, bs = bk[keep], bs[keep]
            pairs, mult = np.unique(bk * 1_000_000 + bs, return_counts=True)           # build rows that are ONE group of (partkey, suppkey)
            pk = pairs // 1_000_000
            live = Cn[pk] > 0
            pairs, mult, pk = pairs[live], mult[live], pk[live]
            gk = res["l_partkey"].to_numpy().astype(np.int64) * 1_000_000 + res["ps_suppkey"].to_numpy().astype(np.int64)
            order = np.argsort(gk, kind="stable")
            ok = bool(np.array_equal(gk[order], pairs))
            err = 0.0
            if ok:
                ok = bool(np.array_equal(res["n"].to_numpy().astype(np.int64)[order], mult * Cn[pk]))
                err = _rel_err(res["revenue"].to_numpy()[order], mult * S[pk])
                ok = ok and err <= VERIFY_RTOL
            return {"rows": nl + nb, "against": "host twin of both tables: per-part revenue and row count of the filtered probe side through the oracle's streaming group-by, times the "
                    "multiplicity of every (partkey, suppkey) build pair", "ok": ok, "max_rel_err": err, "rtol": VERIFY_RTOL, "groups": int(len(pairs)),
                    "build_rows_after_filter": int(len(bk)), "duplicate_build_pairs": int((mult > 1).sum())}
        wld = Workload("join_duplicate_build_keys_sf100", nl + nb, nl * 32 + nb * 24, step_d, "fused_scan",
                       f"lineitem-shaped probe side ({nl} rows) JOIN partsupp-shaped build side ({nb} rows, ~4 per part: duplicate build keys), filter both, group_by(partkey, suppkey).agg(revenue, len)",
                       verify=verify_d, scope="operator")
        wld.inputs = [L, PS]
        return wld
    if name == "q3f":
        # TPC-H Q3 with all three tables: customer[c_mktsegment == "BUILDING"] JOIN orders JOIN lineitem
        no = (rows // 4) if rows else SF100_ORDERS
        nc = datagen.n_customers_for(no)
        O, L = datagen.orders_lineitem_native(pl, no, seed)
        Cst = datagen.customer_native(pl, nc, seed)
        nl = L.height
        lf = queries.q3_full(Cst.lazy(), O.lazy(), L.lazy())

        def step():
            return lf.collect(), (Cst, O, L)
        lf_top = queries.q3_full_top10(Cst.lazy(), O.lazy(), L.lazy())

        def step_top10():
            return lf_top.collect().to_dict(), (Cst, O, L)
        wlf = Workload("tpch_q3_three_tables_sf100", nl + no + nc, nl * datagen.Q3_LINEITEM_BYTES_PER_ROW + no * datagen.Q3_ORDERS_BYTES_PER_ROW + nc * 9, step, "join_probe_emit",
                       f"TPC-H Q3 with customer ({nc}) x orders ({no}) x lineitem ({nl}): c_mktsegment == 'BUILDING', two joins, group_by(orderkey, orderdate, shippriority)",
                       variants={"tpch_q3_three_tables_sf100_order_by_limit10": step_top10}, verify=lambda res, budget: verify_q3(res, no, seed, budget, customer_seed=seed), scope="operator")
        wlf.inputs = [Cst, O, L]
        return wlf
    if name == "cfg2n":
        # config 2's nullable variant ("5 % nulls on x"): x carries a validity bitmap -- bit i = the generator's stream-3 value in
        # [0, 100) is >= 5, built on the device by the library's compare kernel (a Boolean column's values ARE such a bitmap) -- so the fused scan reads
        # 24 B + 1 bit per row, x.mean() counts the valid rows only and x * (1 - y) is null where x is
        n = rows or 1_000_000_000
        a_ = native_uniform_column(pl, "a", pl.Int64, "Int64", n, seed, 0, 0, 2 ** 31)
        x_ = native_uniform_column(pl, "x", pl.Float64, "Float64", n, seed, 1, 0, 10 ** 9, 1e-7)
        y_ = native_uniform_column(pl, "y", pl.Float64, "Float64", n, seed, 2, 0, 10 ** 9, 1e-9)
        valid = native_uniform_column(pl, "r", pl.UInt32, "UInt32", n, seed, 3, 0, 100) >= CFG2_NULL_PCT
        pl._ffi.check(pl._ffi.lib().plx_synchronize())
        xn = pl.Series.from_device("x", pl.Float64, x_.device_ptrs()[0], n, validity_ptr=valid.device_ptrs()[0], keepalive=(x_, valid))
        dfn = pl.DataFrame([a_, xn, y_])
        lfn = queries.cfg2(dfn.lazy())

        def step_n():
            return lfn.collect().to_dict(), (dfn,)
        wln = Workload("cfg2_nulls5pct_1e9", n, n * 24 + n // 8, step_n, "fused_scan_regagg", f"config 2 with {CFG2_NULL_PCT} % nulls on x (validity bitmap): {n}-row Int64/Float64 frame, filter + arithmetic + sum/mean",
                       verify=lambda res, budget: verify_cfg2(res, n, seed, budget, null_pct=CFG2_NULL_PCT))
        wln.inputs = [dfn]
        return wln
    if name == "cfg2":
        n = rows or 1_000_000_000
        a = x = y = None
        df = _native_or_none("cfg2", lambda: pl.DataFrame([native_uniform_column(pl, "a", pl.Int64, "Int64", n, seed, 0, 0, 2 ** 31),
                                                           native_uniform_column(pl, "x", pl.Float64, "Float64", n, seed, 1, 0, 10 ** 9, 1e-7),
                                                           native_uniform_column(pl, "y", pl.Float64, "Float64", n, seed, 2, 0, 10 ** 9, 1e-9)]))
        if df is None:
            g = torch.Generator(device="cuda"); g.manual_seed(seed)
            a = torch.randint(0, 2 ** 31, (n,), generator=g, device="cuda", dtype=torch.int64)
            x = torch.rand((n,), generator=g, device="cuda", dtype=torch.float64) * 100.0
            y = torch.rand((n,), generator=g, device="cuda", dtype=torch.float64)
            df = pl.DataFrame([pl.Series.from_torch("a", a), pl.Series.from_torch("x", x), pl.Series.from_torch("y", y)])
            torch.cuda.synchronize()
        lf = queries.cfg2(df.lazy())

        def step():
            return lf.collect().to_dict(), (df, a, x, y)
        verify = (lambda res, budget: verify_cfg2(res, n, seed, budget)) if a is None else None
        wl2 = Workload("cfg2_filter_arith_agg_1e9", n, n * 24, step, "fused_scan_regagg_static", f"config 2: {n}-row Int64/Float64 frame, filter + arithmetic + sum/mean", verify=verify)
        wl2.inputs = [df]
        return wl2
    if name == "cfg3z":
        # config 3's skewed variant (SURVEY.md 8(d) "Zipf s = 1.1"): heavy-tailed keys over [0, 1e6) -- key 0 holds ~9 % of the rows, key 1 ~5 % --
        # from the library's fixed-point generator (plx_datagen_zipf; its host twin is bit-identical)
        n = rows or 1_000_000_000
        kz = datagen.zipf_native(pl, "key", n, seed, 0, 1_000_000)
        for row0, blk in _blocks(n):
            if not np.array_equal(pl.DataFrame([kz]).slice(row0, blk)["key"].to_numpy(), datagen.zipf_native_host_mt(row0, blk, seed, 0, 1_000_000, threads=1)):
                raise RuntimeError(f"device zipf generator differs from its host twin at rows [{row0}, {row0 + blk})")
        dfz = pl.DataFrame([kz, native_uniform_column(pl, "v", pl.Int64, "Int64", n, seed, 1, 0, 1000)])
        lfz = queries.cfg3(dfz.lazy())

        def step_z():
            return lfz.collect(), (dfz,)
        wlz = Workload("cfg3_zipf_1e9", n, n * 16 + 1_000_000 * 20, step_z, "fused_scan", f"config 3 with Zipf(1.1)-like keys: {n} rows, heavy-tailed Int64 keys over [0, 1e6), group_by(key).agg(sum, count)",
                       verify=lambda res, budget: verify_groupby_dense(res, "key", "v_sum", n, seed, 1_000_000, "Int64", "Int64", (0, 1000), ("count", "v_count"), budget,
                                                                       key_gen=lambda r0, m: datagen.zipf_native_host_mt(r0, m, seed, 0, 1_000_000)), scope="operator")
        wlz.inputs = [dfz]
        return wlz
    if name == "cfg3s":
        # config 3 on SPARSE keys: the 1e6 distinct ids times an odd 64-bit constant (mod 2^64) -- random-looking 64-bit keys with no usable range --
        # and Int64 values spanning 2^41, so neither the key nor the value packs: 16-byte records through the hash-mode partitioned group-by
        n = rows or 1_000_000_000
        ids = native_uniform_column(pl, "key", pl.Int64, "Int64", n, seed, 0, 0, 1_000_000)
        vs = native_uniform_column(pl, "v", pl.Int64, "Int64", n, seed, 1, -(1 << 40), 1 << 40)
        dfs = pl.DataFrame([ids]).with_columns((pl.col("key") * HASHED_KEY_MULT).alias("key"))
        dfs = pl.DataFrame([dfs["key"], vs])
        pl._ffi.check(pl._ffi.lib().plx_synchronize())
        del ids
        lfs = queries.cfg3(dfs.lazy())

        def step_s():
            return lfs.collect(), (dfs,)
        unmap = lambda k: (k.astype(np.uint64) * np.uint64(HASHED_KEY_INV)).astype(np.int64)
        wls = Workload("cfg3_sparse_keys_1e9", n, n * 16 + 1_000_000 * 20, step_s, "fused_scan", f"config 3 on sparse 64-bit keys: {n} rows, 1e6 distinct keys = id * 0x9E3779B97F4A7C15 mod 2^64, "
                       "Int64 values spanning 2^41 (nothing packs), group_by(key).agg(sum, count)",
                       verify=lambda res, budget: verify_groupby_dense(res, "key", "v_sum", n, seed, 1_000_000, "Int64", "Int64", (-(1 << 40), 1 << 40), ("count", "v_count"), budget, key_unmap=unmap),
                       scope="operator")
        wls.inputs = [dfs]
        return wls
    if name == "cfg3w":
        # config 3 on a WIDE key: two Int64 key columns (id times two different odd 64-bit constants: neither has a usable range, together they do not bit-pack),
        # group_by(k1, k2).agg(sum, count) -- the reference row-encodes such keys (crates/polars-row; hash_keys.rs:334 RowEncodedKeys); here 24-byte records
        # {k1, k2, narrowed value} through the hash-partitioned group-by with word-by-word LDS tables
        n = rows or 1_000_000_000
        ids = native_uniform_column(pl, "id", pl.Int64, "Int64", n, seed, 0, 0, 1_000_000)
        vw = native_uniform_column(pl, "v", pl.Int64, "Int64", n, seed, 1, 0, 1000)
        dfw = pl.DataFrame([ids]).with_columns((pl.col("id") * HASHED_KEY_MULT).alias("k1"), (pl.col("id") * WIDE_KEY_MULT2).alias("k2"))
        dfw = pl.DataFrame([dfw["k1"], dfw["k2"], vw])
        pl._ffi.check(pl._ffi.lib().plx_synchronize())
        del ids
        lfw = queries.cfg3w(dfw.lazy())

        def step_w():
            return lfw.collect(), (dfw,)

        def verify_w(res, budget):
            k1 = res["k1"].to_numpy().astype(np.uint64)
            ids_ = (k1 * np.uint64(HASHED_KEY_INV)).astype(np.int64)
            second_ok = bool(np.array_equal(res["k2"].to_numpy().astype(np.uint64), ids_.astype(np.uint64) * np.uint64(WIDE_KEY_MULT2 % (1 << 64))))
            frame = {"k1": type("H", (), {"to_numpy": lambda self_: ids_})(), "v_sum": res["v_sum"], "v_count": res["v_count"]}
            out = verify_groupby_dense(frame, "k1", "v_sum", n, seed, 1_000_000, "Int64", "Int64", (0, 1000), ("count", "v_count"), budget)
            out["second_key_column_consistent"] = second_ok
            out["ok"] = bool(out.get("ok")) and second_ok if out.get("ok") is not None else None
            return out
        wlw = Workload("cfg3_two_int64_keys_1e9", n, n * 24 + 1_000_000 * 28, step_w, "fused_scan", f"config 3 on a two-column Int64 key: {n} rows, 1e6 distinct (k1, k2) pairs that do not bit-pack, "
                       "group_by(k1, k2).agg(sum, count)", verify=verify_w, scope="operator")
        wlw.inputs = [dfw]
        return wlw
    if name == "cfg3":
        n = rows or 1_000_000_000
        key = v = None
        df = _native_or_none("cfg3", lambda: pl.DataFrame([native_uniform_column(pl, "key", pl.Int64, "Int64", n, seed, 0, 0, 1_000_000),
                                                           native_uniform_column(pl, "v", pl.Int64, "Int64", n, seed, 1, 0, 1000)]))
        if df is None:
            g = torch.Generator(device="cuda"); g.manual_seed(seed)
            key = torch.randint(0, 1_000_000, (n,), generator=g, device="cuda", dtype=torch.int64)
            v = torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int64)
            df = pl.DataFrame([pl.Series.from_torch("key", key), pl.Series.from_torch("v", v)])
            torch.cuda.synchronize()
        lf = queries.cfg3(df.lazy())

        def step():
            return lf.collect(), (df, key, v)
        verify = (lambda res, budget: verify_groupby_dense(res, "key", "v_sum", n, seed, 1_000_000, "Int64", "Int64", (0, 1000), ("count", "v_count"), budget)) if key is None else None
        wl3 = Workload("cfg3_groupby_1e6_keys_1e9", n, n * 16 + 1_000_000 * 20, step, "fused_scan", f"config 3: {n} rows, 1e6 Int64 keys, group_by(key).agg(sum, count)",
                       verify=verify, scope="operator")
        wl3.inputs = [df]
        return wl3
    if name == "cfg5":
        n = rows or 1_000_000_000
        codes = v = None
        # the dictionary of the 1e6 distinct "id%010d" strings (H2O id3 style) stays on the host; the column holds their u32 codes and
        # knows the dictionary size (code bounds), like any dictionary-encoded string column
        cats = ["id%010d" % i for i in range(1, 1_000_001)]
        df = _native_or_none("cfg5", lambda: pl.DataFrame([native_uniform_column(pl, "k", pl.Categorical(cats, pl.UInt32), "UInt32", n, seed, 0, 0, 1_000_000),
                                                           native_uniform_column(pl, "v", pl.Float64, "Float64", n, seed, 1, 0, 10 ** 9, 1e-7)]))
        if df is None:
            g = torch.Generator(device="cuda"); g.manual_seed(seed)
            codes = torch.randint(0, 1_000_000, (n,), generator=g, device="cuda", dtype=torch.int32)   # dictionary codes of "id%010d" keys (u32)
            v = torch.rand((n,), generator=g, device="cuda", dtype=torch.float64) * 100.0
            df = pl.DataFrame([pl.Series.from_torch("k", codes, dtype=pl.Categorical(cats, pl.UInt32)), pl.Series.from_torch("v", v)])
            torch.cuda.synchronize()
        lf = queries.cfg5(df.lazy())

        def step():
            return lf.collect(), (df, codes, v)
        verify = (lambda res, budget: verify_groupby_dense(res, "k", "v_sum", n, seed, 1_000_000, "UInt32", "Float64", (0, 10 ** 9, 1e-7), ("mean", "v_mean"), budget)) if codes is None else None
        wl5 = Workload("cfg5_dict_string_keys_1e9", n, n * 12 + 1_000_000 * 20, step, "part_scatter", f"config 5: {n} rows, 1e6 dictionary-encoded string keys (u32 codes), group_by(k).agg(sum, mean)",
                       verify=verify, scope="operator")
        wl5.inputs = [df]
        return wl5
    if name == "cfg5s":
        # config 5 starting from RAW Utf8View keys (16-byte views of the 12-byte strings "id%010d", generated in HBM): every step groups on the
        # views themselves (plx_strview_groupby: rows partitioned by the view's hash, LDS tables keyed by the view; the distinct views are the
        # result's dictionary).  PLX_BENCH_CFG5S_ENCODE=1: the route of rounds 2-3 -- encode the views to dictionary codes on the device
        # (plx_strview_dict_encode_device), then the dense-id group-by.
        n = rows or 1_000_000_000
        views = datagen.id_views_native(pl, "k", n, seed, 0, 1, 1_000_001)
        v = native_uniform_column(pl, "v", pl.Float64, "Float64", n, seed, 1, 0, 10 ** 9, 1e-7)
        encode_first = os.environ.get("PLX_BENCH_CFG5S_ENCODE") == "1"

        def step():
            k = pl.Series.from_device_views("k", views, encode="eager" if encode_first else "deferred")
            return queries.cfg5(pl.DataFrame([k, v]).lazy()).collect(), (views, v, k)

        def verify(res, budget):
            import numpy as np
            cats = list(res["k"].dtype.categories)                      # the device-built dictionary, downloaded here (outside the timed region)
            ids = np.array([int(c[2:]) for c in cats], dtype=np.int64)

            class Mapped:
                def __init__(self, a): self.a = a
                def to_numpy(self): return self.a
            codes = res["k"].to_numpy()
            frame = {"k": Mapped(ids[codes] - 1), "v_sum": res["v_sum"], "v_mean": res["v_mean"]}
            return verify_groupby_dense(frame, "k", "v_sum", n, seed, 1_000_000, "Int64", "Float64", (0, 10 ** 9, 1e-7), ("mean", "v_mean"), budget, key_args=(1, 1_000_001), key_shift=-1)
        wl5s = Workload("cfg5_utf8view_keys_1e9", n, n * 24 + 1_000_000 * 28, step, "strview_dict_encode" if encode_first else "strgroup_scatter",
                        f"config 5 from raw strings: {n} rows, Utf8View keys (16-byte views of 1e6 distinct 12-byte strings) -> " +
                        ("device-side dictionary encoding -> group_by(k).agg(sum, mean)" if encode_first else "group_by(k).agg(sum, mean) on the views (string-key operator)"),
                        verify=verify, scope="operator")
        wl5s.inputs = [views, v]
        return wl5s
    if name == "cfg5l":
        # config 5 with keys the view does not hold: 20-byte strings "id%010d-longkey" (views {20, prefix, buffer, offset} into a pool of the 1e6 distinct strings).  The
        # string-key operator's fast path is inline keys only (kernels_strgroup.hip); a deferred views column of long keys is encoded on the device -- the encoder compares
        # THROUGH the buffers (binview_index_map.rs:106-117 get_long_key) -- and the group-by runs on the codes.  The round-5 review's 20-byte-key extra.
        n = rows or 1_000_000_000
        views, data = datagen.long_id_views_native(pl, n, seed, 0, 1, 1_000_001)
        v = native_uniform_column(pl, "v", pl.Float64, "Float64", n, seed, 1, 0, 10 ** 9, 1e-7)

        def step():
            k = pl.Series.from_device_views("k", views, data, encode="deferred")
            return queries.cfg5(pl.DataFrame([k, v]).lazy()).collect(), (views, data, v, k)

        def verify(res, budget):
            import numpy as np
            cats = list(res["k"].dtype.categories)
            assert all(len(c) == 20 and c.endswith("-longkey") for c in cats[:1000])
            ids = np.array([int(c[2:12]) for c in cats], dtype=np.int64)

            class Mapped:
                def __init__(self, a): self.a = a
                def to_numpy(self): return self.a
            codes = res["k"].to_numpy()
            frame = {"k": Mapped(ids[codes] - 1), "v_sum": res["v_sum"], "v_mean": res["v_mean"]}
            return verify_groupby_dense(frame, "k", "v_sum", n, seed, 1_000_000, "Int64", "Float64", (0, 10 ** 9, 1e-7), ("mean", "v_mean"), budget, key_args=(1, 1_000_001), key_shift=-1)
        wl5l = Workload("cfg5_utf8view_20_byte_keys_1e9", n, n * 44 + 1_000_000 * 36, step, "strview_dict_encode",
                        f"config 5 from raw strings the views do not hold: {n} rows, Utf8View keys (views into a pool of 1e6 distinct 20-byte strings) -> device-side dictionary "
                        "encoding through the buffers -> group_by(k).agg(sum, mean) on the codes", verify=verify, scope="operator")
        wl5l.inputs = [views, data, v]
        return wl5l
    raise ValueError(name)


def jit_stats(pl):
    """(kernels the run-time compiler made available so far, milliseconds it spent compiling) -- plx_jit_stats"""
    import ctypes as C
    n, ms = C.c_int32(), C.c_double()
    pl._ffi.check(pl._ffi.lib().plx_jit_stats(C.byref(n), C.byref(ms)))
    return int(n.value), float(ms.value)


def kernel_stats(pl):
    """Per-kernel (name -> [count, total_us, algo_bytes per launch]) from the library's HIP-event tracer."""
    import ctypes as C
    F = pl._ffi
    cap = 65536
    recs = (F.ProfileRecord * cap)()
    n = C.c_int32()
    F.check(F.lib().plx_profile_fetch(recs, cap, C.byref(n)))
    out = {}
    for i in range(n.value):
        r = recs[i]
        nm = r.name.decode()
        e = out.setdefault(nm, [0, 0.0, 0])
        e[0] += 1; e[1] += r.end_us - r.start_us; e[2] = max(e[2], int(r.algo_bytes))
    return out


def timed(pl, wl: Workload, steps: int, warmup: int, distributed: bool, combine=None):
    """-> (seconds of the timed region, per-kernel stats, result of the last timed step, cold_ms).  cold_ms = the very first
    step of this workload in the process (plan lowering, column statistics passes, JIT if any, first-touch allocations),
    measured only when the workload has not run before (warm-up steps follow it)."""
    import torch
    import torch.distributed as dist
    F = pl._ffi
    res = None
    cold_ms = None
    # The harness process must not garbage-collect inside the timed region (what timeit does too): with torch / numpy / pyarrow loaded
    # a generation-2 collection walks a few million objects, ~40 ms, and its allocation-count trigger put it on timed step 4 of the
    # headline every single run (step_ms showed 4.3, 4.3, 4.3, 40, 4.5, 4.3 ... = 6.1 ms/step "measured" for a 4.3 ms step).  Collected
    # BEFORE the warm-up steps: a 40 ms pause between warm-up and timed region lets the device drop its clocks, and the first timed step of
    # every series then ran 5-10 % slow (3.80, 3.52, 3.55 ...).
    import gc
    gc.collect()
    gc.disable()
    for i in range(warmup):
        if i == 0 and not getattr(wl, "_ran", False):
            torch.cuda.synchronize(); F.check(F.lib().plx_synchronize())
            t0 = time.perf_counter()
            res, _keep = wl.step()
            F.check(F.lib().plx_synchronize())
            cold_ms = (time.perf_counter() - t0) * 1e3
            wl._ran = True
        else:
            res, _keep = wl.step()
        if combine:
            combine(res)
    F.check(F.lib().plx_profile_clear())
    F.check(F.lib().plx_profile_enable(1))
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(); F.check(F.lib().plx_synchronize())
    t0 = time.perf_counter()
    marks = [t0]
    trace = os.environ.get("PLX_BENCH_STEP_TRACE") == "1"          # measurement: a line per timed step on stderr (next to PLX_POOL_TRACE's)
    for _ in range(steps):
        res, _keep = wl.step()
        if combine:
            res = combine(res)
        marks.append(time.perf_counter())      # host-side return times (a step ends with its result download, so these are real step times)
        if trace:
            print(f"STEP {wl.name} {len(marks) - 1}: {(marks[-1] - marks[-2]) * 1e3:.3f} ms", file=sys.stderr, flush=True)
    torch.cuda.synchronize(); F.check(F.lib().plx_synchronize())
    timed.last_step_ms = [round((b - a) * 1e3, 3) for a, b in zip(marks, marks[1:])]
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    stats = kernel_stats(pl)
    F.check(F.lib().plx_profile_enable(0))
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, stats, res, cold_ms


def release_memory(pl) -> None:
    """Between two secondary workloads: the library's pool and torch's cache go back to the driver, which unmaps tens of gigabytes in the
    background for a while after the calls return -- kernels that run meanwhile are slowed and the odd one stalls for milliseconds (the
    three-table Q3, whose small inputs are generated in 0.1 s, showed one 14 ms step in every full run and none on its own).  So: wait."""
    import ctypes
    import gc
    import torch
    pl._ffi.lib().plx_memory_trim()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    # The host side too: the all-rows verification of a workload leaves tens of gigabytes of freed numpy arrays behind; handed back to the kernel lazily they
    # cost the NEXT workload a 10-20 ms stall of the main thread in one of its first steps (the three-table Q3, whose inputs take 0.1 s to generate, showed it in
    # every full run; with PLX_BENCH_VERIFY=0 never).  Collect, and return the freed heap to the kernel now.
    gc.collect()
    try:
        ctypes.CDLL("libc.so.6").malloc_trim(0)
    except OSError:
        pass
    # ... until the device's free memory has stopped growing (three equal readings 100 ms apart; at most 4 s)
    last, same, t_end = -1, 0, time.perf_counter() + 4.0
    while same < 3 and time.perf_counter() < t_end:
        time.sleep(0.1)
        free = torch.cuda.mem_get_info()[0]
        same = same + 1 if free == last else 0
        last = free


def step_spread(step_ms, rows_per_step: int) -> dict:
    """BASELINE.md 2.3 reports min / median / max of the timed steps; `value` / `ms_per_step` stay the K-step mean the driver's contract
    defines (total rows / the bracketed wall time), the median-based rate is given beside them."""
    if not step_ms:
        return {}
    v = sorted(step_ms)
    med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return {"ms_per_step_min": v[0], "ms_per_step_median": round(med, 4), "ms_per_step_max": v[-1], "value_median_based": round(rows_per_step / (med * 1e-3), 1) if med > 0 else None}


def one_shot_ms(pl, wl):
    """One step of a query whose input columns are "new" to the library: everything it LEARNED about them in earlier steps -- value ranges (statistics
    passes, ranges learned as a by-product of a scan), the group-by planner's key sample and heavy hitters, sampled sortedness -- is dropped first
    (plx_column_drop_statistics); kernels (AOT / JIT cache) and the memory pool stay warm.  What a drop-in executor pays for ONE collect() of a query it
    has the code for, next to the steady-state `ms_per_step` (statistics cached on the columns) and `cold_first_step_ms` (first step of the process)."""
    import torch
    frames = getattr(wl, "inputs", None)
    if not frames:
        return None
    F = pl._ffi
    best = None
    cols = [c for f in frames for c in (f.get_columns() if hasattr(f, "get_columns") else [f])]
    for _ in range(2):
        for c in cols:
            F.check(F.lib().plx_column_drop_statistics(c._h))
        torch.cuda.synchronize(); F.check(F.lib().plx_synchronize())
        t0 = time.perf_counter()
        wl.step()
        F.check(F.lib().plx_synchronize())
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None else min(best, ms)
    return round(best, 3)


def end_to_end_q1(pl, n: int, reps: int = 3) -> dict:
    """TPC-H Q1 END TO END from host memory (BASELINE.md 2.3 "also report end-to-end with H2D/D2H"): the seven lineitem columns start as numpy
    arrays in pageable host memory, a step = hand them to the boundary (plx_column_from_host: page-locked in place, one DMA per column), run the
    query, download the result.  PCIe-inclusive, so it is never `value`; n rows of the same generator as the headline."""
    import numpy as np
    from polars_amd import datagen, queries
    host = {c: np.ascontiguousarray(a) for c, a in datagen.lineitem_native_host_mt(0, n, 10).items()}
    nbytes = sum(a.nbytes for a in host.values())
    times, up_times, res = [], [], None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        df = datagen.to_frame(pl, host, datagen.LINEITEM_Q1_COLS)
        t1 = time.perf_counter()
        res = queries.q1(df.lazy()).collect().to_dict()
        times.append(time.perf_counter() - t0); up_times.append(t1 - t0)
        del df
    times, up_times = sorted(times[1:]), sorted(up_times[1:])           # the first pass page-locks the host arrays for the first time
    med, up = times[len(times) // 2], up_times[len(up_times) // 2]
    want = q1_oracle_blocks(n, 10, 30.0, block=n)[0] if n <= 100_000_000 else None
    ver = dict(compare_q1(res, want), rows=n, rtol=VERIFY_RTOL) if want is not None else None
    return {"rows": n, "host_bytes": nbytes, "ms_per_step_median": round(med * 1e3, 3), "upload_ms_median": round(up * 1e3, 3), "rows_per_s": round(n / med, 1),
            "pcie_inclusive_GBps": round(nbytes / med / 1e9, 2), "upload_GBps": round(nbytes / up / 1e9, 2),
            "what": "numpy columns in host memory -> plx_column_from_host (7 columns) -> Q1 -> result download; median of %d" % reps, "verified": ver}


def scan_extra(pl, n: int):
    """File -> device columns (no query): a lineitem-like table of n rows written by pyarrow as Parquet (uncompressed / Snappy /
    Zstandard) and as an Arrow IPC file, read with the library's own scan (metadata parsed by the library; Snappy, Zstandard (round 6: pq_zstd_entropy +
    pq_zstd_execute) and all page decoding on the device; `parquet_zstd_host_threads` = the same file with PLX_PARQUET_ZSTD=host, the round-5 path, for
    comparison); every read is checked against the source columns; pyarrow's own
    multi-threaded read of the same file is the CPU yardstick.  Files live in a temporary directory (page cache)."""
    import shutil
    import tempfile
    import numpy as np
    import pyarrow as pa
    import pyarrow.ipc as ipc
    import pyarrow.parquet as pq
    F = pl._ffi
    rng = np.random.default_rng(3)
    key = np.sort(rng.integers(1, 4 * n, n))
    qty = rng.integers(1, 51, n)
    price = rng.random(n) * 1e5
    flag = rng.integers(0, 3, n)
    ship = rng.integers(694224000, 912470400, n) * 1_000_000
    nmask = rng.random(n) < 0.1
    nval = rng.integers(0, 1 << 30, n)
    t = pa.table({"l_orderkey": pa.array(key), "l_quantity": pa.array(qty), "l_extendedprice": pa.array(price),
                  "l_returnflag": pa.array(np.array(["R", "A", "N"])[flag]), "l_shipdate": pa.array(ship, pa.timestamp("us")), "l_nullable": pa.array(nval, mask=nmask)})
    decoded = sum(c.nbytes for c in t.columns)
    want = {"l_orderkey": int(key.sum()), "l_quantity": int(qty.sum()), "l_nullable": int(nval[~nmask].sum())}      # none of these sums leaves int64
    out = {"rows": n, "decoded_bytes": decoded, "columns": t.column_names, "pyarrow_threads": pa.cpu_count(), "files": {}}
    d = tempfile.mkdtemp(prefix="plx_scan_")
    try:
        def check(df):
            try:
                ok = df.height == n and all(int(df[c].sum()) == v for c, v in want.items()) and df["l_nullable"].null_count() == int(nmask.sum())
                ok = ok and int(df["l_shipdate"].min()) == int(ship.min()) and int(df["l_shipdate"].max()) == int(ship.max())
                ok = ok and abs(float(df["l_extendedprice"].sum()) - float(price.sum())) <= 1e-9 * abs(float(price.sum()))
                counts = df.lazy().group_by("l_returnflag").agg(pl.len().alias("n")).collect().sort_host("l_returnflag")
                return bool(ok and dict(zip(counts["l_returnflag"], counts["n"])) == {k: int((flag == i).sum()) for i, k in enumerate(["R", "A", "N"])})
            except Exception as e:      # the timing stands on its own; say why the check could not be made
                return f"check failed to run: {type(e).__name__}: {e}"[:200]

        def measure(read, path, py_read):
            read(path)                                          # warm: page cache, pool, staging buffers
            ts = []
            F.check(F.lib().plx_profile_clear()); F.check(F.lib().plx_profile_enable(1))
            for _ in range(3):
                t0 = time.perf_counter(); df = read(path); F.check(F.lib().plx_synchronize()); ts.append(time.perf_counter() - t0)
            ks = {k: round(v[1] / 3) for k, v in kernel_stats(pl).items()}
            F.check(F.lib().plx_profile_enable(0))
            t0 = time.perf_counter(); py_read(path); t_pa = time.perf_counter() - t0
            fb, best = os.path.getsize(path), min(ts)
            return {"file_bytes": fb, "read_ms": round(best * 1e3, 2), "file_GBps": round(fb / best / 1e9, 2), "decoded_GBps": round(decoded / best / 1e9, 2),
                    "rows_per_s": round(n / best), "kernel_us": ks, "pyarrow_read_ms": round(t_pa * 1e3, 2), "verified": check(df)}
        for codec in ("none", "snappy", "zstd"):
            path = os.path.join(d, f"li_{codec}.parquet")
            pq.write_table(t, path, compression=codec, row_group_size=1 << 20)
            out["files"][f"parquet_{codec}"] = measure(pl.read_parquet, path, pq.read_table)
            if codec == "zstd":
                os.environ["PLX_PARQUET_ZSTD"] = "host"
                try:
                    out["files"]["parquet_zstd_host_threads"] = measure(pl.read_parquet, path, pq.read_table)
                finally:
                    del os.environ["PLX_PARQUET_ZSTD"]
            os.remove(path)
        path = os.path.join(d, "li.arrow")
        with ipc.new_file(path, t.schema) as w:
            for b in t.to_batches(max_chunksize=1 << 20):
                w.write_batch(b)
        out["files"]["arrow_ipc"] = measure(pl.read_ipc, path, lambda p: ipc.open_file(p).read_all())
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


PMC_ROUND = "r06"


def pmc_traffic(workload_name: str, kernel: str, rows: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter summary of this round (profiles/<round>/<workload>_pmc.json,
    tools/pmc_all.sh + tools/pmc_summarise.py: FETCH_SIZE and WRITE_SIZE in separate passes, the gfx950 correction applied), or None.
    `kernel` is the name the library's HIP-event profile gives the launch; AOT kernels carry their instantiation in it
    ("fused_scan_ldsagg_static#3", "part3_scatter[#4,d,t4,p2]") and the summary is keyed by the same string derived from the kernel SYMBOL the
    counters were collected on -- only an exact match counts, so a summary of another variant (or another round's kernels) yields None,
    never a number.  Only meaningful at the workload's full size."""
    short = {"tpch_q1_sf100": ("q1", SF100_LINEITEM), "tpch_q3_sf100": ("q3", SF100_ORDERS + SF100_LINEITEM), "tpch_q3_three_tables_sf100": ("q3f", None),
             "cfg2_filter_arith_agg_1e9": ("cfg2", 10 ** 9), "cfg3_groupby_1e6_keys_1e9": ("cfg3", 10 ** 9), "cfg5_dict_string_keys_1e9": ("cfg5", 10 ** 9),
             "tpch_q3_sf100_shuffled_inputs": ("q3s", SF100_ORDERS + SF100_LINEITEM), "cfg5_utf8view_keys_1e9": ("cfg5s", 10 ** 9), "cfg5_utf8view_20_byte_keys_1e9": ("cfg5l", 10 ** 9),
             "tpch_q3_sf100_hashed_keys": ("q3h", SF100_ORDERS + SF100_LINEITEM), "cfg2_nulls5pct_1e9": ("cfg2n", 10 ** 9), "cfg3_zipf_1e9": ("cfg3z", 10 ** 9),
             "cfg3_sparse_keys_1e9": ("cfg3s", 10 ** 9), "cfg3_two_int64_keys_1e9": ("cfg3w", 10 ** 9), "join_duplicate_build_keys_sf100": ("q3d", SF100_LINEITEM + SF100_LINEITEM * 2 // 15),
             "join_aggregate_reads_build_side_sf100": ("q3dc", SF100_LINEITEM + SF100_LINEITEM * 2 // 15), "join_materialise_sf100": ("joinm", SF100_ORDERS + SF100_LINEITEM), "join_materialise_sf100_hashed_keys": ("joinmh", SF100_ORDERS + SF100_LINEITEM), "semi_join_materialise_sf100": ("semim", SF100_ORDERS + SF100_LINEITEM),
             "filter_materialise_1e9": ("filterm", 10 ** 9), "gather_1e9": ("gather", 10 ** 9), "tpch_q1_sf100_two_predicates_ten_aggregates_jit": ("q1j", SF100_LINEITEM)}.get(workload_name)
    if short is None or (short[1] is not None and abs(rows - short[1]) > 0.01 * short[1]):
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PMC_ROUND, f"{short[0]}_pmc.json")))
    except Exception:
        return None
    v = (d.get("kernels") or {}).get(kernel)
    if v and d.get("keyed_by") == "kernel symbol":
        pmc_traffic.sources[workload_name] = f"profiles/{PMC_ROUND}/{short[0]}_pmc.json" + (f" (collected {d['collected']})" if d.get("collected") else "")
        return int(v["hbm_bytes_per_launch"])
    return None


pmc_traffic.sources = {}


def traffic_source(workload_name: str):
    """Where a line's `traffic` figure comes from: it is LOOKED UP by kernel name in a committed rocprofv3 counter summary (FETCH_SIZE / WRITE_SIZE in separate
    passes, tools/pmc_all.sh), not measured during this run (round-5 review, weak 9)."""
    src = pmc_traffic.sources.get(workload_name)
    return None if src is None else src + ": looked up by kernel name, not measured in this run"


def roofline(stats, wl, steps: int):
    """`kernel` scope (one streaming kernel does the work): the dominant kernel's algorithmic bytes / its mean duration.
    `operator` scope (several passes over intermediates: partitioned group-by, join pipeline): the workload's algorithmic
    bytes (required inputs + outputs once, intermediates count zero) / the summed duration of ALL kernels of
    one step; the dominant kernel's own pass traffic is listed next to it."""
    if not stats:
        return None
    name = max(stats, key=lambda k: stats[k][1])
    cnt, tot_us, algo = stats[name]
    avg_us = tot_us / cnt
    if wl.scope == "operator":
        step_us = sum(v[1] for v in stats.values()) / max(steps, 1)
        ach = wl.algo_bytes / (step_us * 1e-6) / 1e9 if step_us > 0 else 0.0
        traffic = None
        per = {}
        for k, v in stats.items():
            t = pmc_traffic(wl.name, k, wl.rows)
            if t is not None:
                per[k] = t * v[0] // max(steps, 1)
        # traffic: counter-measured HBM bytes of one step, only when EVERY kernel that moves a noticeable share of the step's time has a counter
        # figure of its own instantiation (pmc_traffic matches the full variant name); hbm_frac = those bytes / the summed kernel time / peak
        covered_us = sum(stats[k][1] for k in per) / max(steps, 1)
        if per and covered_us >= 0.97 * step_us:
            traffic = int(sum(per.values()))
        dom_meas = pmc_traffic(wl.name, name, wl.rows)
        nominal = algo / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
        dom = {"name": name, "avg_us": round(avg_us, 2), "launches": cnt, "pass_bytes_per_launch": algo,
               # the nominal pass rate counts every input byte; a kernel that skips loads (the late-materialising probe) can exceed the HBM peak
               # on that scale: then only the counter-measured rate is a bandwidth
               "pass_GBps": round(nominal, 1) if nominal <= HBM_PEAK_GBS else None}
        if dom_meas is not None:
            dom["measured_bytes_per_launch"] = dom_meas
            dom["measured_GBps"] = round(dom_meas / (avg_us * 1e-6) / 1e9, 1) if avg_us > 0 else 0.0
        # A query whose kernels SKIP input they do not need (the late-materialising probe of Q3 on clustered rows: payload lines whose rows all fail the
        # predicate are never fetched) would score above what the part can stream if every input byte counted: the numerator is then the bytes the query
        # actually needed -- the counter-measured traffic when it is below the nominal algorithmic bytes -- and the nominal figure is kept beside it.
        # Without a counter figure a rate above the measured copy ceiling (6.29 TB/s, MI355X_MICROARCH.md) is reported AT the ceiling, flagged.
        nominal = ach
        required = wl.algo_bytes
        note = None
        if traffic is not None and traffic < wl.algo_bytes:
            required = traffic
            note = "numerator = counter-measured HBM bytes (the query skips input lines it does not need); nominal_* = every input byte once"
        ach = required / (step_us * 1e-6) / 1e9 if step_us > 0 else 0.0
        if traffic is None and ach > HBM_COPY_CEILING_GBS:
            ach = HBM_COPY_CEILING_GBS
            note = "nominal rate above the part's measured copy ceiling and no counter figure for this run: reported at the ceiling"
        out = {"bound": "hbm", "scope": "operator: all kernels of one step", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(ach / HBM_PEAK_GBS, 4), "kernel_us_per_step": round(step_us, 2), "algo_bytes_per_step": wl.algo_bytes, "required_bytes_per_step": int(required), "traffic": traffic,
               "hbm_frac": None if traffic is None or step_us <= 0 else round(traffic / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
               "traffic_source": traffic_source(wl.name) if traffic is not None else None, "dominant_kernel": dom}
        if note:
            out.update(nominal_achieved=round(nominal, 1), nominal_frac=round(nominal / HBM_PEAK_GBS, 4), note=note)
        return out
    ach = algo / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    traffic = pmc_traffic(wl.name, name, wl.rows)
    return {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "avg_kernel_us": round(avg_us, 2), "launches": cnt, "algo_bytes_per_launch": algo, "traffic": traffic,
            "traffic_source": traffic_source(wl.name) if traffic is not None else None}


def pyarrow_q1(cols, cutoff):
    """TPC-H Q1 with pyarrow (compute kernels + Acero group_by, its own thread pool): a third-party CPU yardstick next to
    the oracle; returns the result table."""
    import pyarrow as pa
    import pyarrow.compute as pc
    t = pa.table({k: pa.array(v) for k, v in cols.items()})
    t = t.filter(pc.less_equal(t["l_shipdate"], pa.scalar(cutoff, pa.int64())))
    disc_price = pc.multiply(t["l_extendedprice"], pc.subtract(pa.scalar(1.0), t["l_discount"]))
    t = t.append_column("disc_price", disc_price).append_column("charge", pc.multiply(disc_price, pc.add(pa.scalar(1.0), t["l_tax"])))
    return t.group_by(["l_returnflag", "l_linestatus"]).aggregate([("l_quantity", "sum"), ("l_extendedprice", "sum"), ("disc_price", "sum"), ("charge", "sum"),
                                                                   ("l_quantity", "mean"), ("l_extendedprice", "mean"), ("l_discount", "mean"), ([], "count_all")])


def polars_q1(cols, cutoff):
    """TPC-H Q1 with REAL Polars when the box has a wheel (SURVEY.md 8(d) "Preferred: real Polars"): same host arrays (zero-copy
    from numpy), in-memory and streaming engines.  -> {"in_memory_s", "streaming_s", "threads", "version", "result"} or None."""
    try:
        import polars as rp
    except Exception:
        return None
    import datetime as dt
    df = rp.DataFrame({k: v for k, v in cols.items()})
    c = rp.col
    disc_price = c("l_extendedprice") * (1 - c("l_discount"))
    q = (df.lazy().filter(c("l_shipdate") <= cutoff).group_by("l_returnflag", "l_linestatus")
         .agg(c("l_quantity").sum().alias("sum_qty"), c("l_extendedprice").sum().alias("sum_base_price"), disc_price.sum().alias("sum_disc_price"),
              (disc_price * (1 + c("l_tax"))).sum().alias("sum_charge"), c("l_quantity").mean().alias("avg_qty"), c("l_extendedprice").mean().alias("avg_price"),
              c("l_discount").mean().alias("avg_disc"), rp.len().alias("count_order")))
    out = {"threads": int(rp.thread_pool_size()), "version": rp.__version__}
    for eng, key in (("in-memory", "in_memory_s"), ("streaming", "streaming_s")):
        best = None
        for _ in range(2):
            t0 = time.perf_counter(); r = q.collect(engine=eng); d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        out[key] = best
        out["result"] = r.sort("l_returnflag", "l_linestatus").to_dict(as_series=False)
    return out


def cpu_baseline_q1(seconds: float, rows: int = 0, seed: int = 98, gpu_result=None):
    """TPC-H Q1 on the host cores, on THE SAME ROWS as the GPU leg when its input came from the library's generator
    (rows, seed given: the generator's host twin reproduces them block by block), which also checks the timed GPU result
    (`verified`).  Timed: real Polars if the box has it (kind "reference"); otherwise the CPU oracle (kind "port", a C++
    restatement -- NOT Polars itself: no polars wheel / rustc in the image): orc_q1_streaming, the morsel-driven partitioned
    group-by the reference dispatches this shape to (GroupByStreamingExec), all host threads, oracle time only.
    Also reported: orc_q1 (the in-memory FilterExec -> GroupByExec sequence) and pyarrow Acero on the first rows."""
    import numpy as np
    from oracle import pyoracle as orc
    from polars_amd import datagen
    cores = orc.hardware_threads()
    orc.set_threads(cores)
    cutoff = datagen.us(1998, 9, 2)
    same_rows = rows > 0
    n = rows if same_rows else 200_000_000
    # ~10-30 s of CPU work: the whole input when the oracle gets through it within the budget, a prefix otherwise
    want, done, t_orc, first = q1_oracle_blocks(n, seed, budget_s=max(seconds * 2, 20.0), block=min(n, 100_000_000))
    verified = None
    if gpu_result is not None and same_rows:
        if done == n:
            verified = dict(compare_q1(gpu_result, want), rows=n, against="oracle (orc_q1_streaming over the generator's host twin, all rows of the timed input)", rtol=VERIFY_RTOL)
        else:
            verified = {"rows": done, "pending_prefix": want}     # the caller runs the query on the first `done` rows and compares
    n_mem = min(len(first["l_shipdate"]), 16_000_000)
    cm = {k: v[:n_mem] for k, v in first.items()}
    t0 = time.perf_counter(); orc.q1_native(cm, cutoff, streaming=False); dt_mem = time.perf_counter() - t0
    orc.set_threads(1)
    n_pa, pa_rate = min(len(first["l_shipdate"]), 50_000_000), None
    cp = {k: v[:n_pa] for k, v in first.items()}
    try:
        pyarrow_q1(cp, cutoff)
        t0 = time.perf_counter(); pyarrow_q1(cp, cutoff); pa_rate = round(n_pa / (time.perf_counter() - t0), 1)
    except Exception:   # a yardstick only
        pa_rate = None
    out = {"value": round(done / t_orc, 1), "unit": "rows/s", "cores": cores, "kind": "port", "seconds": round(t_orc, 3),
           "in_memory_engine_rows_per_s": round(n_mem / dt_mem, 1), "pyarrow_acero_rows_per_s": pa_rate,
           "sample": f"TPC-H Q1 on {done} synthetic lineitem rows -- " + ("the SAME rows (generator seed and row range) as the GPU leg" if same_rows else "same generator") +
                     f", evaluated in blocks of <= 1e8 rows, oracle time only (generation excluded); oracle/plx_oracle.cpp orc_q1_streaming with {cores} threads: "
                     "C++ restatement of the reference's streaming/partitioned group-by path (morsels, thread-local hot tables), not Polars itself; "
                     f"in_memory_engine_rows_per_s = orc_q1 (FilterExec -> GroupByExec with per-group index lists) on {n_mem} rows; "
                     f"pyarrow_acero_rows_per_s = the same query with pyarrow compute + Acero group_by on {n_pa} rows (third-party yardstick)"}
    try:
        rp = polars_q1(first, cutoff)     # real Polars on the first block (<= 1e8 rows), if the box has a wheel
    except Exception as e:
        rp = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rp and "streaming_s" in rp:
        nb = len(first["l_shipdate"])
        best = min(rp["in_memory_s"], rp["streaming_s"])
        out.update({"value": round(nb / best, 1), "kind": "reference", "cores": rp["threads"], "seconds": round(best, 3),
                    "polars": {"version": rp["version"], "threads": rp["threads"], "rows": nb, "in_memory_rows_per_s": round(nb / rp["in_memory_s"], 1),
                               "streaming_rows_per_s": round(nb / rp["streaming_s"], 1)},
                    "oracle_port_rows_per_s": round(done / t_orc, 1),
                    "sample": f"real Polars {rp['version']} ({rp['threads']} threads; best of in-memory / streaming engines) on the first {nb} rows of the GPU leg's input; " + out["sample"]})
    elif rp:
        out["polars"] = rp
    else:
        out["polars"] = "not installed on this box (import polars failed): the oracle port is the baseline"
    return out, verified


Q1_FIELDS = ("l_returnflag", "l_linestatus", "sum_qty", "count_order", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc")
Q1_MAX_GROUPS = 16


def pack_q1(res):
    """Q1 result frame (<= 16 groups) -> fixed-size int64 tensor [16, 10]; floats are bit-cast, unused rows have count 0."""
    import numpy as np
    import torch
    from polars_amd import datagen
    t = np.zeros((Q1_MAX_GROUPS, len(Q1_FIELDS)), dtype=np.int64)
    for i in range(len(res["l_returnflag"])):
        for j, f in enumerate(Q1_FIELDS):
            v = res[f][i]
            if f == "l_returnflag": v = datagen.FLAGS.index(v) if isinstance(v, str) else int(v)
            elif f == "l_linestatus": v = datagen.STATUS.index(v) if isinstance(v, str) else int(v)
            t[i, j] = np.float64(v).view(np.int64) if isinstance(v, float) else int(v)
    return torch.from_numpy(t)


def unpack_q1(t):
    import numpy as np
    a = t.cpu().numpy().reshape(-1, Q1_MAX_GROUPS, len(Q1_FIELDS))
    out = []
    for r in a:
        d = {f: [] for f in Q1_FIELDS}
        for row in r:
            if row[Q1_FIELDS.index("count_order")] == 0:
                continue
            for j, f in enumerate(Q1_FIELDS):
                d[f].append(int(row[j]) if j < 4 else float(row[j:j + 1].view(np.float64)[0]))
        out.append(d)
    return out


def allgather_q1(res, ws):
    """One fixed-size tensor all-gather (RCCL over xGMI on the GPU box): 1.25 KB per rank instead of a pickled object."""
    import torch
    import torch.distributed as dist
    mine = pack_q1(res)
    if dist.get_backend() == "nccl":
        mine = mine.cuda()
    allt = torch.empty((ws * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=mine.device)   # ranks concatenated along dim 0
    dist.all_gather_into_tensor(allt, mine.contiguous())
    return unpack_q1(allt)


def allgather_combine_q1(res, ws):
    """The Q1 step's tail at N > 1, vectorised: one fixed-size tensor all-gather of the ranks' result frames (1.25 KB each) and their merge in numpy -- no per-row Python
    (at eight ranks the row-by-row unpack + merge of 128 rows cost ~0.3 ms of a 0.9 ms step).  Same result as combine_q1_results(allgather_q1(...)), which the
    verification still computes the slow way and compares with this one."""
    import numpy as np
    import torch
    import torch.distributed as dist
    mine = pack_q1(res)
    if dist.get_backend() == "nccl":
        mine = mine.cuda()
    allt = torch.empty((ws * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(allt, mine.contiguous())
    a = allt.cpu().numpy()
    F = {f: j for j, f in enumerate(Q1_FIELDS)}
    a = a[a[:, F["count_order"]] != 0]
    key = a[:, F["l_returnflag"]] * 4 + a[:, F["l_linestatus"]]
    uniq, inv = np.unique(key, return_inverse=True)
    g = len(uniq)
    cnt = np.zeros(g, np.int64); np.add.at(cnt, inv, a[:, F["count_order"]])
    qty = np.zeros(g, np.int64); np.add.at(qty, inv, a[:, F["sum_qty"]])
    def fsum(col, weight=None):           # sum per group of a bit-cast float column (times a weight)
        x = np.ascontiguousarray(a[:, F[col]]).view(np.float64)
        acc = np.zeros(g, np.float64)
        np.add.at(acc, inv, x if weight is None else x * weight)
        return acc
    base, disc_p, charge = fsum("sum_base_price"), fsum("sum_disc_price"), fsum("sum_charge")
    disc = fsum("avg_disc", a[:, F["count_order"]].astype(np.float64))
    return {"l_returnflag": (uniq // 4).tolist(), "l_linestatus": (uniq % 4).tolist(), "sum_qty": qty.tolist(), "sum_base_price": base.tolist(), "sum_disc_price": disc_p.tolist(),
            "sum_charge": charge.tolist(), "avg_qty": (qty / cnt).tolist(), "avg_price": (base / cnt).tolist(), "avg_disc": (disc / cnt).tolist(), "count_order": cnt.tolist()}


def combine_q1_results(per_rank):
    """Merge the Q1 results of row-sharded ranks: sums and counts add, averages are recombined from
    (avg x count) -- the partial/final decomposition of polars_amd.dist.PARTIALS applied to the finished frames."""
    merged = {}
    for r in per_rank:
        for i in range(len(r["l_returnflag"])):
            k = (r["l_returnflag"][i], r["l_linestatus"][i])
            m = merged.setdefault(k, {"sum_qty": 0, "sum_base_price": 0.0, "sum_disc_price": 0.0, "sum_charge": 0.0, "_disc": 0.0, "count_order": 0})
            c = r["count_order"][i]
            m["sum_qty"] += r["sum_qty"][i]; m["sum_base_price"] += r["sum_base_price"][i]; m["sum_disc_price"] += r["sum_disc_price"][i]
            m["sum_charge"] += r["sum_charge"][i]; m["_disc"] += r["avg_disc"][i] * c; m["count_order"] += c
    out = {k: [] for k in ("l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order")}
    for (f, s_), m in sorted(merged.items()):
        c = m["count_order"]
        out["l_returnflag"].append(f); out["l_linestatus"].append(s_)
        for k in ("sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "count_order"):
            out[k].append(m[k])
        out["avg_qty"].append(m["sum_qty"] / c); out["avg_price"].append(m["sum_base_price"] / c); out["avg_disc"].append(m["_disc"] / c)
    return out


def run_guarded(worker, deadline_s: float, poll_s: float = 0.25, prints: bool = True) -> int:
    """Runs worker(emit) in a forked child and prints the LAST line it emitted exactly once, from this process.

    The headline measurement comes first; the secondary workloads (`extras`) only refine the same JSON line.  On a GPU box
    whose caches are cold, another library's kernels (torch's generators behind the extras' inputs) can take minutes to
    load, so the worker hands every improved version of the line to `emit`; if it has not finished `deadline_s` seconds
    after the start but a line exists, the child is stopped and the line printed as it stands.  A worker that dies
    after the headline was measured still gets its line printed.  Forking happens before torch / HIP are imported, so
    the child initialises the GPU on its own.  Returns the process exit code.  `prints=False` (ranks other than 0 of an N > 1 run): the
    worker never emits; it is stopped `deadline_s` + 20 s after the start (rank 0 has printed or been cut by then) and that counts as success."""
    import signal
    import tempfile
    tmp = tempfile.mkdtemp(prefix="plx_bench_")
    path = os.path.join(tmp, "line.json")

    ready_path = path + ".ready"

    def emit(line: dict, ready: bool = True):
        """ready=False: the line lacks mandatory parts (the CPU baseline): never cut the worker while only such a version exists."""
        with open(path + ".tmp", "w") as f:
            f.write(json.dumps(line))
        os.replace(path + ".tmp", path)
        if ready and not os.path.exists(ready_path):
            open(ready_path, "w").close()

    sys.stdout.flush(); sys.stderr.flush()
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            worker(emit)
            code = 0
        except BaseException:  # noqa: BLE001 -- the parent decides what to print
            import traceback
            traceback.print_exc()
        finally:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(code)
    t_end = time.monotonic() + deadline_s
    status, stopped = None, False
    while True:
        done, st = os.waitpid(pid, os.WNOHANG)
        if done:
            status = st
            break
        if (time.monotonic() >= t_end and os.path.exists(ready_path)) or (not prints and time.monotonic() >= t_end + 20.0):
            os.kill(pid, signal.SIGKILL)
            os.waitpid(pid, 0)
            stopped = True
            break
        time.sleep(poll_s)
    line = open(path).read() if os.path.exists(path) else None
    for f in (path, path + ".tmp", ready_path):
        if os.path.exists(f):
            os.remove(f)
    os.rmdir(tmp)
    if line is not None:
        if stopped:
            d = json.loads(line)
            d["note"] = f"secondary workloads stopped at the {deadline_s:.0f} s deadline; headline unaffected"
            line = json.dumps(d)
        print_record(json.loads(line))
        return EXIT_PARITY if failed_verifications(json.loads(line)) else 0
    if stopped and not prints:
        return 0
    return os.waitstatus_to_exitcode(status) if status is not None else 1


EXIT_PARITY = 3      # the line was printed, but a result of a timed workload disagreed with the CPU oracle


def failed_verifications(line: dict):
    """Names of the workloads of a bench line whose `verified.ok` is False (headline and extras; the scan extra's per-file `verified`)."""
    bad = []
    if (line.get("verified") or {}).get("ok") is False:
        bad.append((line.get("config") or {}).get("workload", "headline"))
    for name, ex in (line.get("extras") or {}).items():
        if not isinstance(ex, dict):
            continue
        if (ex.get("verified") or {}).get("ok") is False:
            bad.append(name)
        for fk, fv in (ex.get("files") or {}).items():
            if isinstance(fv, dict) and fv.get("verified") is False:
                bad.append(f"{name}.{fk}")
    return bad


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline", "verified", "cold_first_step_ms", "one_shot_ms", "ms_per_step_median", "pool", "comm", "note", "dry_run")
HEADLINE_MAX_BYTES = 4096     # the driver parses the LAST stdout line out of an 8 KB tail: the line it must read stays far below that
EXTRAS_FILE = os.environ.get("PLX_BENCH_EXTRAS_FILE", os.path.join(ROOT, "bench_extras.json"))


def _clip(v, n: int):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 3] + "..."


def _slim(obj, text: int):
    """a JSON object with its prose clipped to `text` characters and nested tables (per-kernel lists, step series) dropped"""
    if not isinstance(obj, dict):
        return obj
    out = {}
    for k, v in obj.items():
        if isinstance(v, dict):
            if k in ("dominant_kernel", "rank0_shard_vs_oracle", "polars"):
                continue
            v = {a: _clip(b, text) for a, b in v.items() if not isinstance(b, (dict, list))}
        elif isinstance(v, list):
            continue
        out[k] = _clip(v, text)
    return out


def headline_line(full: dict, extras_file=None) -> dict:
    """The ONE line the driver parses (the last stdout line): BASELINE's metric, the timing, `config`, `roofline`, `cpu_baseline`, `verified`, with the
    prose clipped, plus one short row per secondary workload (`extras_summary`: ms per step, roofline fraction, counter traffic in GB, oracle verdict).
    Everything else -- per-kernel tables, step series, the secondary workloads in full -- is the FULL record: written to `extras_file` and printed on
    an earlier stdout line.  Always below HEADLINE_MAX_BYTES (round 4's 25 KB line was not parseable from the driver's 8 KB tail)."""
    out = {}
    for k in HEADLINE_KEYS:
        if k in full:
            v = full[k]
            out[k] = _slim(v, 160) if isinstance(v, dict) else _clip(v, 200)
    summ = {}
    for name, ex in (full.get("extras") or {}).items():
        if not isinstance(ex, dict):
            continue
        if "error" in ex:
            summ[name] = {"error": _clip(ex["error"], 60)}
            continue
        r = ex.get("roofline") or {}
        row = {"ms": ex.get("ms_per_step"), "frac": r.get("frac")}
        if r.get("traffic") is not None:
            row["traffic_GB"] = round(r["traffic"] / 1e9, 2)
        ok = (ex.get("verified") or {}).get("ok") if isinstance(ex.get("verified"), dict) else None
        files = ex.get("files")
        if isinstance(files, dict) and files:
            ok = all(isinstance(f, dict) and f.get("verified") is not False for f in files.values())
        if "verified" in ex or files:
            row["ok"] = ok
        summ[name] = {a: b for a, b in row.items() if b is not None or a == "ok"}
    if summ:
        out["extras_summary"] = summ
    if extras_file:
        out["extras_file"] = os.path.relpath(extras_file, ROOT) if os.path.abspath(extras_file).startswith(ROOT + os.sep) else extras_file
    bad = failed_verifications(full)
    if bad:
        out["failed_verifications"] = bad[:8]
    for drop in ("extras_summary", "note"):           # cannot happen with the workloads of this file; the bound is unconditional all the same
        if len(json.dumps(out)) < HEADLINE_MAX_BYTES:
            break
        out.pop(drop, None)
    if len(json.dumps(out)) >= HEADLINE_MAX_BYTES:
        out = {k: (_slim(v, 40) if isinstance(v, dict) else _clip(v, 80)) for k, v in out.items()}
    return out


def print_record(full: dict) -> None:
    """stdout: the full record on an earlier line (prefixed, so that nothing mistakes it for the line), then the headline line LAST; the full record
    also goes to EXTRAS_FILE."""
    path = EXTRAS_FILE
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(full, f)
        os.replace(path + ".tmp", path)
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr)
        path = None
    print("[bench] full record: " + json.dumps(full), flush=True)
    print(json.dumps(headline_line(full, path)), flush=True)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here, one process per GPU,
    with the environment torchrun would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on 127.0.0.1).  Rank 0
    prints the one JSON line on the inherited stdout.  A rank that dies takes the others down (they would wait in a collective
    forever); the exit code is the first non-zero one."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = rc
                print(f"[bench] rank {r} exited with code {rc}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.2)
    return code


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank, ws = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    guard = os.environ.get("PLX_BENCH_GUARD", "1") != "0" and not args.no_extras
    deadline = float(os.environ.get("PLX_BENCH_DEADLINE_S", "420"))
    # runs with secondary workloads go through the guard (every rank of an N > 1 run: a rank whose line-printing peer was cut must not
    # wait in a collective forever); --no-extras runs (rocprofv3 wraps those) do not
    if guard:
        sys.exit(run_guarded(lambda emit: run(args, emit), deadline, prints=rank == 0))
    final = {}
    run(args, lambda line, ready=True: final.update(line))
    if rank == 0:
        print_record(final)
        bad = failed_verifications(final)
        if bad:
            print(f"[bench] parity check failed for: {bad}", file=sys.stderr)
            sys.exit(EXIT_PARITY)


# ---- sharded high-cardinality group-by (cfg3 / cfg5 at N > 1) ------------------------------------------------------------
# Every rank holds a row shard; a step = polars_amd.dist.sharded_groupby: the local partitioned group-by (every rank reduces its rows to
# one partial row per local group), ONE exchange of the PARTIAL rows by key hash (plx_exchange_by_key: hash partition + gather + one
# grouped ncclSend / ncclRecv all-to-all(v) inside the library), and the merge of the partials on the rank that owns the key
# (crates/polars-stream/src/nodes/group_by.rs:140-497).  Raw rows are exchanged instead only when the local group-by would not shrink
# the shard (mode "rows").  The result stays sharded.  `--dry-run` swaps the library and RCCL for numpy frames and gloo so the control
# flow (mode choice, barriers, accounting, the JSON line) can be exercised without GPUs (tests/test_dist_gloo_cpu.py); nothing of it is
# measured.
def verify_sharded_groupby(res_cols, key_name, sum_name, second, total_rows, input_sum, ranks_rows_seeds, n_keys, key_np, val_np, val_args, budget_s):
    """Rank 0's check of a sharded cfg3 / cfg5 result (all ranks' result rows gathered): size-independent properties always -- the key sets
    of the ranks are disjoint (no key twice), the counts add up to the rows of all shards, the sums add up to the column sums every rank
    computed with a different kernel (the whole-column reduction) -- and, when the host twin of ALL shards fits the time budget, the full
    comparison against the oracle's streaming group-by."""
    import numpy as np
    k = np.asarray(res_cols[key_name]).astype(np.int64)
    out = {"groups": int(len(k)), "checks": {}}
    out["checks"]["keys_disjoint_across_ranks"] = bool(len(np.unique(k)) == len(k))
    if second[0] == "count":
        out["checks"]["counts_add_up_to_all_rows"] = bool(int(np.asarray(res_cols[second[1]]).astype(np.int64).sum()) == int(total_rows))
    s = np.asarray(res_cols[sum_name])
    if s.dtype.kind == "f":
        tot = float(np.sum(s.astype(np.float64)))
        out["checks"]["sums_add_up_to_column_sums"] = bool(abs(tot - float(input_sum)) <= 1e-9 * max(1.0, abs(float(input_sum))))
    else:
        out["checks"]["sums_add_up_to_column_sums"] = bool(int(s.astype(np.int64).sum()) == int(input_sum))
    ok = all(out["checks"].values())
    out["against"] = "linearity: disjoint key sets, counts == rows of all shards, sums == whole-column sums (independent reduction kernel)"
    if ok and budget_s > 0:
        from oracle import pyoracle as orc
        from polars_amd import datagen
        vdt = np.int64 if val_np == "Int64" else np.float64
        sums, counts = np.zeros(n_keys, vdt), np.zeros(n_keys, np.int64)
        t0, complete = time.perf_counter(), True
        for n, seed in ranks_rows_seeds:
            done = 0
            while done < n:
                if time.perf_counter() - t0 > budget_s:
                    complete = False
                    break
                m = min(100_000_000, n - done)
                orc.groupby_dense_partial(datagen.uniform_native_host_mt(key_np, done, m, seed, 0, 0, n_keys), datagen.uniform_native_host_mt(val_np, done, m, seed, 1, *val_args), sums, counts)
                done += m
            if not complete:
                break
        if complete:
            order = np.argsort(k, kind="stable")
            present = np.nonzero(counts)[0]
            good = np.array_equal(k[order], present)
            err = 0.0
            if good:
                gs, g2 = s[order], np.asarray(res_cols[second[1]])[order]
                if vdt is np.int64:
                    good = np.array_equal(gs.astype(np.int64), sums[present])
                else:
                    err = _rel_err(gs, sums[present]); good = err <= VERIFY_RTOL
                if second[0] == "count":
                    good = good and np.array_equal(g2.astype(np.int64), counts[present])
                else:
                    e2 = _rel_err(g2, sums[present] / counts[present]); err = max(err, e2); good = good and e2 <= VERIFY_RTOL
            ok = bool(good)
            out.update(against="oracle (orc_groupby_dense_partial over the host twin of EVERY rank's shard, all rows) + " + out["against"], max_rel_err=err, rtol=VERIFY_RTOL,
                       rows=int(total_rows))
        else:
            out["note"] = "the oracle did not cover all shards within its time budget: properties only"
    out["ok"] = bool(ok)
    return out


def dry_doubles():
    """`--dry-run` only: the numpy + gloo stand-ins for the library's frames / communicator / local operators live with the tests (tests/dry_multigpu.py)."""
    import importlib
    tdir = os.path.join(ROOT, "tests")
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    return importlib.import_module("dry_multigpu")


class MultiCtx:
    """What the N > 1 workloads share: the rank's device + library + RCCL communicator (plx_comm_*: the exchange runs inside the
    library), or -- `--dry-run` -- numpy frames + gloo with the same methods.  torch.distributed is the bootstrap and carries the few
    host-side numbers (timings, row totals, the gathered result for rank 0's check)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from polars_amd import dist as pdist
        self.rank, self.local_rank, self.ws = pdist.world()
        self.dry, self.pl, self.torch, self.dist, self.pdist = args.dry_run, None, torch, dist, pdist
        if self.dry:
            pdist.init_process_group("gloo")
            self._ensure_group("gloo")
            self.comm = dry_doubles().DryComm()
        else:
            dev = int(os.environ.get("PLX_BENCH_DEVICE", self.local_rank))
            torch.cuda.set_device(dev)
            import polars_amd as pl
            pl.init(dev)
            self.pl, self.F = pl, pl._ffi
            pdist.init_process_group(os.environ.get("PLX_DIST_BACKEND", "nccl"))
            self._ensure_group(os.environ.get("PLX_DIST_BACKEND", "nccl"))
            self.comm = pdist.LibComm(pl)

    def _ensure_group(self, backend):
        # world size 1 (smoke run on a one-GPU box): the barriers / gathers below still want a group
        if not self.dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
            self.dist.init_process_group(backend, rank=0, world_size=1)

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize(); self.F.check(self.F.lib().plx_synchronize())

    def barrier(self):
        self.dist.barrier()

    def gather(self, obj):
        """every rank's host object, in rank order, on every rank"""
        every = [None] * self.ws
        self.dist.all_gather_object(every, obj)
        return every

    def profile_start(self):
        if not self.dry:
            self.F.check(self.F.lib().plx_profile_clear()); self.F.check(self.F.lib().plx_profile_enable(1))

    def profile_stop(self):
        if self.dry:
            return {}
        st = kernel_stats(self.pl)
        self.F.check(self.F.lib().plx_profile_enable(0))
        return st

    def trim(self):
        """Between two workloads: device pool and torch cache back to the driver, and the host heap a verification left behind back to the kernel (see
        release_memory: lazily reclaimed host memory stalled the next workload's main thread)."""
        import ctypes
        import gc
        if not self.dry:
            self.F.lib().plx_memory_trim(); self.torch.cuda.empty_cache()
        gc.collect()
        try:
            ctypes.CDLL("libc.so.6").malloc_trim(0)
        except OSError:
            pass

    def backend(self):
        return "numpy + gloo DRY RUN (control flow only, nothing measured)" if self.dry else "libpolars_amd + RCCL (plx_exchange_by_key / plx_allgather_frame)"

    def comm_info(self):
        """What the communicator itself says it is (RCCL: ncclCommUserRank / ncclCommCount through plx_comm_info) next to what the launcher's environment names."""
        r, w = self.comm.info() if hasattr(self.comm, "info") else (self.comm.rank, self.comm.world_size)
        return {"rank": r, "world_size": w, "env_world_size": int(os.environ.get("WORLD_SIZE", "1")), "library": "gloo (dry run)" if self.dry else "rccl"}

    def close(self):
        # (the measurements are done and rank 0's line is out: a peer that has already left must not turn the farewell barrier into a traceback)
        try:
            self.barrier()
        except RuntimeError:
            pass
        if hasattr(self.comm, "close"):
            self.comm.close()
        try:
            self.dist.destroy_process_group()
        except RuntimeError:
            pass


# The all-rows host checks of the SECONDARY sharded workloads are deferred until all of them have been timed (run_multi): a check leaves tens of gigabytes of
# freed host arrays behind, and what the kernel does with them stalled the next workload's main thread on rank 0 -- whose time is the max over the ranks.
DEFERRED_CHECKS = {"on": False, "pending": []}


class PendingCheck(dict):
    """Placeholder of a `verified` entry whose check has not run yet (serialises as a note should the line be printed before it has)."""
    def __init__(self, fn):
        super().__init__(ok=None, note="host check pending (it runs after every secondary workload has been timed)")
        self.fn = fn


def host_check(fn):
    if DEFERRED_CHECKS["on"]:
        p = PendingCheck(fn)
        DEFERRED_CHECKS["pending"].append(p)
        return p
    return fn()


def timed_multi(ctx, step, steps: int, warmup: int):
    """`warmup` untimed steps, then exactly `steps` timed ones bracketed by barrier + device synchronisation on both sides; the time is the
    MAX over the ranks.  -> (seconds, per-kernel stats of this rank, result of the last step, per-step ms of this rank)"""
    import gc
    res = None
    gc.collect(); gc.disable()          # see timed(): the harness must not collect inside the timed region, nor pause between warm-up and timed steps
    for _ in range(max(warmup, 1)):
        res = step()
    ctx.profile_start()
    ctx.barrier(); ctx.sync()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(steps):
        res = step()
        marks.append(time.perf_counter())
    ctx.sync(); ctx.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    stats = ctx.profile_stop()
    dt = max(ctx.gather(dt))
    return dt, stats, res, [round((b - a) * 1e3, 3) for a, b in zip(marks, marks[1:])]


METRIC = "rows/sec + achieved HBM GB/s, TPC-H Q1/Q3 SF100, 1/2/4/8 GPU vs CPU"


def multi_line_base(ctx, args, steps, warmup, dt, total_rows, dtype, strong):
    return {"metric": METRIC, "value": round(total_rows * steps / dt, 1), "unit": "rows/s", "n_gpus": ctx.ws, "steps": steps, "warmup": max(warmup, 1),
            "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic"}


def sharded_groupby_line(ctx, args, workload: str, steps: int, warmup: int) -> dict:
    """cfg3 / cfg5 at N > 1 (dist.sharded_groupby), one rank per GPU: the local partitioned group-by, ONE exchange of the partial rows by
    key hash inside the library, the merge on the owner of the key; rank 0 gathers the (small) sharded result and checks it."""
    import numpy as np
    pdist = ctx.pdist
    rank, ws, dry = ctx.rank, ctx.ws, ctx.dry
    cfg5 = workload == "cfg5"
    strong = args.scaling == "strong"
    n = args.rows or (1_000_000_000 // ws if strong else 1_000_000_000)      # strong: the 1e9-row configuration split over the ranks
    n_keys = 1_000_000
    seed = 10 + rank
    key_name, val_name = ("k", "v") if cfg5 else ("key", "v")
    spec = pdist.GroupBySpec(key_name, [("v_sum", val_name, "sum"), ("v_mean", val_name, "mean")] if cfg5 else [("v_sum", val_name, "sum"), ("v_count", val_name, "count")])
    second = ("mean", "v_mean") if cfg5 else ("count", "v_count")
    val_np, val_args = ("Float64", (0, 10 ** 9, 1e-7)) if cfg5 else ("Int64", (0, 1000))
    if dry:
        from polars_amd import datagen
        k = datagen.uniform_native_host("UInt32" if cfg5 else "Int64", 0, n, seed, 0, 0, n_keys)
        v = datagen.uniform_native_host(val_np, 0, n, seed, 1, *val_args)
        dry_m = dry_doubles()
        df = dry_m.DryFrame({key_name: k, val_name: v})
        ops = dry_m.DryOps()
        column_sum = lambda: float(v.sum()) if cfg5 else int(v.sum())
        result_cols = lambda r: dict(r.cols)
    else:
        pl = ctx.pl
        wl0 = make_workload(pl, workload, n, seed=seed)      # this rank's shard, from the library's generator
        df = wl0.step()[1][0]
        ops = pdist.LibFrameOps(pl)
        column_sum = lambda: df.lazy().select(pl.col(val_name).sum().alias("s")).collect()["s"].to_list()[0]
        result_cols = lambda r: {c: r[c].to_numpy() for c in r.columns}
    comm, info = ctx.comm, {}
    force = os.environ.get("PLX_BENCH_FORCE_SHARDED") == "1"
    step = lambda: pdist.sharded_groupby(comm, df, spec, ops, mode=args.mode, info=info, always_exchange=force)
    step()                                                   # the first step also fills the per-column caches; the exchange accounting starts after it
    comm.rows_sent = comm.bytes_sent = 0
    dt, stats, res, step_ms = timed_multi(ctx, step, steps, max(warmup, 1))
    sent_rows, sent_bytes = comm.rows_sent / (steps + max(warmup, 1)), comm.bytes_sent / (steps + max(warmup, 1))
    every = ctx.gather({"sent_rows": sent_rows, "sent_bytes": sent_bytes, "groups": int(res.height), "rows": n, "sum": column_sum(), "cols": {c: np.asarray(a) for c, a in result_cols(res).items()}})
    total_rows = sum(e["rows"] for e in every)
    def check():
        allc = {c: np.concatenate([e["cols"][c] for e in every]) for c in every[0]["cols"]}
        isum = sum(e["sum"] for e in every)
        verified = verify_sharded_groupby(allc, key_name, "v_sum", second, total_rows, isum if cfg5 else int(round(isum)), [(n, 10 + r) for r in range(ws)], n_keys,
                                          "UInt32" if cfg5 else "Int64", val_np, val_args, 0.0 if os.environ.get("PLX_BENCH_VERIFY", "1") == "0" else float(os.environ.get("PLX_BENCH_VERIFY_BUDGET_S", "40")))
        if verified.get("ok") is False:
            print(f"[bench] VERIFICATION FAILED for the sharded {workload}: {verified}", file=sys.stderr)
        return verified
    verified = host_check(check) if rank == 0 else None
    rec_bytes = (4 + 8) if cfg5 else 16
    algo = n * rec_bytes + n_keys * 20 // ws
    how = {"preagg": "local partitioned group-by -> partial rows exchanged by key hash (one grouped all-to-all(v)) -> merge on the owner",
           "rows": "raw rows exchanged by key hash (one grouped all-to-all(v)) -> single-GPU partitioned group-by over the owned keys", "local": "single rank"}[info.get("mode", "local")]
    line = multi_line_base(ctx, args, steps, warmup, dt, total_rows, "f64" if cfg5 else "int64", strong)
    line.update({
        "config": {"workload": ("cfg5_dict_string_keys" if cfg5 else "cfg3_groupby_1e6_keys") + f"_sharded_x{ws}_{args.scaling}", "rows_per_gpu": n, "algorithmic_bytes_per_gpu_step": algo,
                   "description": f"{n} rows per rank, 1e6 keys over all ranks, group_by(key).agg(...): {how}; result sharded by key",
                   "parallelism": f"row-sharded x{ws}; {how}", "backend": ctx.backend()},
        "exchange_mode": info.get("mode"), "shrink_estimate": info.get("shrink_estimate"), "partial_rows_per_rank": info.get("partial_rows"),
        "shuffle": {"rows_sent_per_rank_per_step": round(sum(e["sent_rows"] for e in every) / ws, 1), "bytes_sent_per_rank_per_step": round(sum(e["sent_bytes"] for e in every) / ws, 1),
                    "fabric_GBps_per_rank": round(sum(e["sent_bytes"] for e in every) / ws * steps / dt / 1e9, 2)},
        "groups_total": sum(e["groups"] for e in every),
        "whole_query_GBps_per_gpu": round(algo * steps / dt / 1e9, 1), "step_ms": step_ms,
        "verified": verified,
        "kernels": _kernels(stats, 8) if stats else {},
    })
    if dry:
        line["dry_run"] = True
    return line


def sharded_q3_line(ctx, args, steps: int, warmup: int, mode=None) -> dict:
    """TPC-H Q3 at N > 1 (BASELINE config 4: lineitem JOIN orders, key-hash sharded, RCCL all-to-all): dist.sharded_join_groupby over the
    library's communicator.  Every rank generates its own orders and their lines (dbgen row order) and shifts the order keys into its own
    key range, so the global tables are the union of the shards.  --scaling strong (what config 4 is quoted on): SF100 in TOTAL, 1/N per
    rank; weak: SF100 per rank.  mode (or PLX_Q3_MODE) = auto | broadcast | shuffle: "auto" all-gathers the filtered build side when it is small
    (TPC-H orders after its predicates: ~0.5 GB over all ranks) and moves no probe row; "shuffle" routes both sides by key hash (the exchange
    config 4 names) -- the default line is "auto", the shuffle exchange is measured as its own extra.  Rank 0 gathers
    the sharded result and checks every rank's key range against the numpy restatement over that rank's host twin (+ the oracle's Q3 on
    a prefix), within the verification budget."""
    import numpy as np
    from polars_amd import datagen, queries
    pdist = ctx.pdist
    rank, ws, dry = ctx.rank, ctx.ws, ctx.dry
    strong = args.scaling == "strong"
    no = (args.rows // 4) if args.rows else (SF100_ORDERS // ws if strong else SF100_ORDERS)
    seed = 10 + rank
    key_span = 4 * no + 64                                     # the generator uses 8 of every 32 key values: o_orderkey < 4 * n_orders
    off = rank * key_span
    date = datagen.us(1995, 3, 15)
    if dry:
        o, li, _cnt = datagen.orders_lineitem_native_host_mt(0, no, no, seed, threads=2)
        o["o_shippriority"] = np.zeros(no, np.int64)
        o["o_orderkey"] = o["o_orderkey"] + off; li["l_orderkey"] = li["l_orderkey"] + off
        dry_m = dry_doubles()
        O, L = dry_m.DryFrame({c: o[c] for c in datagen.ORDERS_Q3_COLS}), dry_m.DryFrame({c: li[c] for c in datagen.LINEITEM_Q3_COLS})
        ops, spec = dry_m.DryJoinOps(date), pdist.JoinGroupBySpec("l_orderkey", "o_orderkey", "l_orderkey", [("revenue", "sum")])
        result_cols = lambda r: dict(r.cols)
    else:
        pl = ctx.pl
        O, L = datagen.orders_lineitem_native(pl, no, seed)
        check_native_q3(pl, O, L, no, seed)
        if off:
            O = O.with_columns((pl.col("o_orderkey") + off).alias("o_orderkey"))
            L = L.with_columns((pl.col("l_orderkey") + off).alias("l_orderkey"))
        ops, spec = pdist.q3_ops(pl)
        result_cols = lambda r: {c: r[c].to_numpy() for c in r.columns}
    nl = int(L.height)
    comm, info = ctx.comm, {}
    force = os.environ.get("PLX_BENCH_FORCE_SHARDED") == "1"
    mode = mode or os.environ.get("PLX_Q3_MODE", "auto")
    step = lambda: pdist.sharded_join_groupby(comm, ops, L, O, spec, mode=mode, info=info, always_exchange=force)
    step()
    comm.rows_sent = comm.bytes_sent = 0
    dt, stats, res, step_ms = timed_multi(ctx, step, steps, max(warmup, 1))
    sent_rows, sent_bytes = comm.rows_sent / (steps + max(warmup, 1)), comm.bytes_sent / (steps + max(warmup, 1))
    every = ctx.gather({"sent_rows": sent_rows, "sent_bytes": sent_bytes, "groups": int(res.height), "rows": nl + no, "cols": {c: np.asarray(a) for c, a in result_cols(res).items()}})
    total_rows = sum(e["rows"] for e in every)
    def check():
        allc = {c: np.concatenate([e["cols"][c] for e in every]) for c in every[0]["cols"]}
        k = allc["l_orderkey"].astype(np.int64)
        budget = float(os.environ.get("PLX_BENCH_VERIFY_BUDGET_S", "40"))
        per, ok = [], bool(len(np.unique(k)) == len(k))               # the ranks' key sets are disjoint: no key twice in the gathered result

        class Host:
            def __init__(self, a): self.a = a
            def to_numpy(self): return self.a
        for r in range(ws):
            m = (k >= r * key_span) & (k < (r + 1) * key_span)
            frame = {"l_orderkey": Host(k[m] - r * key_span), "o_orderdate": Host(allc["o_orderdate"][m]), "o_shippriority": Host(allc["o_shippriority"][m]), "revenue": Host(allc["revenue"][m])}
            v = verify_q3(frame, no, 10 + r, budget / ws, oracle_orders=min(no, 4_000_000 // ws))
            per.append({"rank_keys": r, "ok": v["ok"], "orders": v["orders"], "covers_whole_input": v["covers_whole_input"], "groups_checked": v["groups_checked"], "max_rel_err": v["max_rel_err"]})
            ok = ok and bool(v["ok"])
        verified = {"ok": ok, "rows": int(sum(p["orders"] for p in per)), "keys_disjoint_across_ranks": bool(len(np.unique(k)) == len(k)), "groups_total": int(len(k)),
                    "against": "per key range of every rank: oracle Q3 on a prefix + numpy restatement over the blocks of that rank's host twin the budget allows", "per_rank": per, "rtol": VERIFY_RTOL}
        if not ok:
            print(f"[bench] VERIFICATION FAILED for the sharded q3: {verified}", file=sys.stderr)
        return verified
    verified = host_check(check) if rank == 0 and os.environ.get("PLX_BENCH_VERIFY", "1") != "0" else None
    algo = nl * datagen.Q3_LINEITEM_BYTES_PER_ROW + no * datagen.Q3_ORDERS_BYTES_PER_ROW
    how = {"shuffle": "both sides filtered, then routed by key hash (one grouped all-to-all(v) per input), local fused join -> group-by over the owned keys",
           "broadcast": "filtered build side all-gathered, probe side stays, partial groups routed by key (one small all-to-all(v)) and merged by the owner", "local": "single rank"}[info.get("mode", "local")]
    line = multi_line_base(ctx, args, steps, warmup, dt, total_rows, "f64", strong)
    line.update({
        "config": {"workload": f"tpch_q3_sf100_sharded_x{ws}" + ("_shuffle" if mode == "shuffle" else "") + f"_{args.scaling}", "rows_per_gpu": nl + no, "orders_per_gpu": no, "lineitem_rows_per_gpu": nl, "algorithmic_bytes_per_gpu_step": algo,
                   "description": f"TPC-H Q3 (orders {no} x lineitem {nl} per rank), filter both -> hash join -> group_by(orderkey, orderdate, shippriority): {how}; result sharded by key",
                   "parallelism": f"row-sharded x{ws}; {how}", "backend": ctx.backend()},
        "exchange_mode": info.get("mode"), "build_rows_after_filter_per_rank": info.get("build_rows"), "probe_rows_after_filter_per_rank": info.get("probe_rows"),
        "partial_rows_per_rank": info.get("partial_rows"),
        "shuffle": {"rows_sent_per_rank_per_step": round(sum(e["sent_rows"] for e in every) / ws, 1), "bytes_sent_per_rank_per_step": round(sum(e["sent_bytes"] for e in every) / ws, 1),
                    "fabric_GBps_per_rank": round(sum(e["sent_bytes"] for e in every) / ws * steps / dt / 1e9, 2)},
        "groups_total": sum(e["groups"] for e in every),
        "whole_query_GBps_per_gpu": round(algo * steps / dt / 1e9, 1), "step_ms": step_ms,
        "verified": verified,
        "kernels": _kernels(stats, 8) if stats else {},
    })
    if dry:
        line["dry_run"] = True
    return line


def rowsharded_line(ctx, args, workload: str, steps: int, warmup: int) -> dict:
    """q1 / cfg2 / q3f at N > 1: independent row shards (SURVEY.md 8(e) "scan / filter / arith / whole-column agg: contiguous row-range
    split"); Q1's six-group partial results are combined with an all-gather of 1.25 KB per rank, no row crosses xGMI; cfg2 / q3f run as
    replicas over their own shards.  Rank 0 checks ITS shard's result against the oracle over the shard's host twin (Q1: and that the
    combined result equals the merge of the gathered per-rank results)."""
    rank, ws, dry = ctx.rank, ctx.ws, ctx.dry
    strong = args.scaling == "strong"
    seed = 10 + rank
    rows = args.rows
    if strong and not rows:
        rows = {"q1": SF100_LINEITEM, "q3f": 4 * SF100_ORDERS, "cfg2": 10 ** 9}.get(workload, 0) // ws
    if dry:
        if workload != "q1":
            raise ValueError("--dry-run knows q1, q3, cfg3 and cfg5")
        n = rows or 200_000
        one = dry_doubles().dry_q1_step(n, seed)
        wl = Workload("tpch_q1_sf100", n, n * 42, None, "", f"TPC-H Q1, lineitem {n} rows per rank (numpy stand-in)")
    else:
        wl = make_workload(ctx.pl, workload, rows, seed=seed, ws=ws)
        one = lambda: wl.step()[0]
    local = {}

    def step():
        r = one()
        local["res"] = r
        return allgather_combine_q1(r, ws) if workload == "q1" else r
    dt, stats, res, step_ms = timed_multi(ctx, step, steps, max(warmup, 1))
    verified = None
    if workload == "q1":
        parts = ctx.gather(local["res"])
        if rank == 0 and os.environ.get("PLX_BENCH_VERIFY", "1") != "0":
            mine_res = local["res"]

            def check():
                merged_ok = compare_q1_dicts(res, combine_q1_results(parts))
                budget = float(os.environ.get("PLX_BENCH_VERIFY_BUDGET_S", "40"))
                want, done, _t, _f = q1_oracle_blocks(wl.rows, seed, budget, block=min(wl.rows, 100_000_000))
                mine = compare_q1(mine_res, want) if done == wl.rows else {"ok": None, "note": "host check ran out of its time budget"}
                return {"ok": bool(merged_ok and mine.get("ok") is not False) if mine.get("ok") is not None else (None if merged_ok else False), "rows": int(done),
                        "combined_equals_merge_of_rank_results": bool(merged_ok), "rank0_shard_vs_oracle": mine,
                        "against": "rank 0's shard: oracle (orc_q1_streaming over the shard's host twin); combined result: merge of the gathered per-rank results"}
            verified = host_check(check)
    elif rank == 0 and not dry and os.environ.get("PLX_BENCH_VERIFY", "1") != "0":
        last_res = local["res"]
        verified = host_check(lambda: _verify(wl, last_res, float(os.environ.get("PLX_BENCH_VERIFY_BUDGET_S", "40"))))
    line = multi_line_base(ctx, args, steps, warmup, dt, wl.rows * ws, "f64", strong)
    line.update({
        "config": {"workload": wl.name + f"_x{ws}_{args.scaling}", "description": wl.desc, "rows_per_gpu": wl.rows, "algorithmic_bytes_per_gpu_step": wl.algo_bytes,
                   "parallelism": (f"row-sharded x{ws}, all-gather of group partials" if workload == "q1" else f"row-sharded x{ws}, independent replicas (no data-path collective)"),
                   "backend": ctx.backend()},
        "whole_query_GBps_per_gpu": round(wl.algo_bytes * steps / dt / 1e9, 1), "step_ms": step_ms, **step_spread(step_ms, wl.rows * ws),
        "roofline": roofline(stats, wl, steps) if stats else None, "verified": verified,
        "kernels": _kernels(stats, 8) if stats else {},
    })
    if dry:
        line["dry_run"] = True
    return line


def compare_q1_dicts(a: dict, b: dict) -> bool:
    """two Q1 results in to_dict() layout: same groups, integer columns equal, floats within 1e-9"""
    import numpy as np
    from polars_amd import datagen

    def codes(d):      # a result that went through pack_q1 / unpack_q1 carries the two keys as dictionary codes, one straight from to_dict() as strings
        return ([datagen.FLAGS.index(v) if isinstance(v, str) else int(v) for v in d["l_returnflag"]], [datagen.STATUS.index(v) if isinstance(v, str) else int(v) for v in d["l_linestatus"]])
    (fa, sa), (fb, sb) = codes(a), codes(b)
    if sorted(zip(fa, sa)) != sorted(zip(fb, sb)):
        return False
    oa = sorted(range(len(a["count_order"])), key=lambda i: (fa[i], sa[i]))
    ob = sorted(range(len(b["count_order"])), key=lambda i: (fb[i], sb[i]))
    for f in Q1_FIELDS[2:]:
        x, y = np.array([a[f][i] for i in oa], np.float64), np.array([b[f][i] for i in ob], np.float64)
        if not np.allclose(x, y, rtol=1e-9, atol=0):
            return False
    return True


MULTI_EXTRAS = ("q3", "q3:shuffle", "cfg3", "cfg5", "q1", "q1:weak")     # ":weak" = the per-rank SF100 shard (weak scaling), labelled so in its config.workload
LATE_WORKLOADS = ("filterm", "gather")       # frame-returning operators with multi-gigabyte results: timed and checked last, one at a time
EXTRA_WORKLOADS = ("q1j", "q3", "q3h", "q3d", "q3dc", "joinm", "joinmh", "semim", "q3f", "cfg2", "cfg2n", "cfg3", "cfg3z", "cfg3s", "cfg3w", "cfg5", "cfg5s", "cfg5l", "q1")      # the secondary workloads of the N = 1 line, in this order


def run_multi(args, emit):
    """bench.py at N > 1 (one process per GPU; also `--dry-run` and the world-size-1 smoke run PLX_BENCH_FORCE_SHARDED=1): the headline
    workload, then -- unless --no-extras -- the other sharded workloads as `extras`, every rank taking part in each."""
    ctx = MultiCtx(args)

    def one(workload, steps, warmup):
        if workload in ("cfg3", "cfg5"):
            return sharded_groupby_line(ctx, args, workload, steps, warmup)
        if workload.startswith("q3") and workload != "q3f":
            return sharded_q3_line(ctx, args, steps, warmup, mode=workload.partition(":")[2] or None)
        return rowsharded_line(ctx, args, workload.partition(":")[0], steps, warmup)
    line = one(args.workload, args.steps, args.warmup)
    line["comm"] = ctx.comm_info()
    if line["comm"]["world_size"] != line["comm"]["env_world_size"]:
        raise RuntimeError(f"the communicator spans {line['comm']['world_size']} ranks, the launcher's environment names {line['comm']['env_world_size']}")
    if ctx.rank == 0:
        emit(line)
    if not args.no_extras:
        extras = {}
        line["extras"] = extras
        k2 = max(3, args.steps // 4)
        scaling = args.scaling
        DEFERRED_CHECKS["on"], DEFERRED_CHECKS["pending"] = True, []
        for w in [w for w in MULTI_EXTRAS if w != args.workload and not (w.endswith(":weak") and scaling == "weak")]:
            ctx.trim()
            # BASELINE config 4 is SF100 in TOTAL over the ranks: the Q3 extras always run it that way; "<workload>:weak" is the weak-scaling variant of a headline
            args.scaling = "strong" if w.startswith("q3") else ("weak" if w.endswith(":weak") else scaling)
            try:
                ex = one(w, k2, 2)
                name = ex["config"]["workload"]
                ex = {k: ex[k] for k in ("value", "unit", "ms_per_step", "scaling", "config", "exchange_mode", "shuffle", "groups_total", "whole_query_GBps_per_gpu", "step_ms",
                                         "roofline", "verified", "kernels", "partial_rows_per_rank", "build_rows_after_filter_per_rank", "probe_rows_after_filter_per_rank") if k in ex}
            except Exception as e:  # a secondary workload must never take the headline line down (the other ranks raise alike or the guard cuts the run)
                name, ex = w, {"error": f"{type(e).__name__}: {e}"[:300]}
            extras[name] = ex
            if ctx.rank == 0:
                emit(line)
        args.scaling = scaling
        DEFERRED_CHECKS["on"] = False
        for ex in extras.values():              # rank 0 only has any: the host checks, now that nothing is left to time
            v = ex.get("verified")
            if isinstance(v, PendingCheck):
                try:
                    ex["verified"] = v.fn()
                except Exception as e:
                    ex["verified"] = {"rows": 0, "ok": None, "error": f"{type(e).__name__}: {e}"[:300]}
                emit(line)
        DEFERRED_CHECKS["pending"] = []
    ctx.close()


def _kernels(stats, top: int):
    return {k: {"launches": v[0], "avg_us": round(v[1] / v[0], 2), "pass_bytes_per_launch": int(v[2])} for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:top]}


def _verify(wl, res, budget_s: float):
    """Runs the workload's host check on the result of its last timed step; never raises (a failed check is reported, not fatal)."""
    if wl.verify is None:
        return {"rows": 0, "ok": None, "note": "inputs did not come from the library's generator (no host twin): not checked"}
    try:
        v = wl.verify(res, budget_s)
    except Exception as e:
        return {"rows": 0, "ok": None, "error": f"{type(e).__name__}: {e}"[:300]}
    if v.get("ok") is False:
        print(f"[bench] VERIFICATION FAILED for {wl.name}: {v}", file=sys.stderr)
    return v


def run(args, emit):
    """The benchmark proper; emit(line) is called with every improved version of the JSON line (rank 0)."""
    import torch
    rank, local_rank, ws = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if ws > 1 or args.dry_run or os.environ.get("PLX_BENCH_FORCE_SHARDED") == "1":
        # N > 1, `--dry-run`, or the sharded operators at world size 1 (PLX_BENCH_FORCE_SHARDED=1: a self-exchange through RCCL, the smoke run a one-GPU box allows)
        run_multi(args, emit)
        return
    distributed = False
    dev = int(os.environ.get("PLX_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev)
    import polars_amd as pl
    pl.init(dev)
    seed = 10 + rank
    rows = args.rows
    # the memory pool is sized once, at start-up (plx_memory_reserve: what an engine does with its device pool): the largest transient buffer of the
    # workloads below is config 5's 24.8 GB record pool (raw string keys), and mapping it inside a query costs 0.7 s
    pool = {"reserve_gb": 0.0, "map_ms": 0.0}
    if not args.no_extras and os.environ.get("PLX_BENCH_POOL_GB", "26") != "0":
        # at most 30 % of the device memory that is free now; a failed reservation is not fatal (the first query that needs the memory maps it then)
        free_b, _total_b = torch.cuda.mem_get_info(dev)
        want_b = min(int(float(os.environ.get("PLX_BENCH_POOL_GB", "26")) * (1 << 30)), int(free_b * 0.3))
        t0 = time.perf_counter()
        rc = pl._ffi.lib().plx_memory_reserve(want_b)
        if rc == 0:
            pool = {"reserve_gb": round(want_b / (1 << 30), 2), "map_ms": round((time.perf_counter() - t0) * 1e3, 1)}
        else:
            print(f"[bench] plx_memory_reserve({want_b}) failed (rc {rc}): continuing without a pre-grown pool", file=sys.stderr)
            pl._ffi.lib().plx_memory_reserve(0)
    wl = make_workload(pl, args.workload, rows, seed=seed, ws=ws)
    dt, stats, res, cold_ms = timed(pl, wl, args.steps, max(args.warmup, 1), distributed)
    total_rows = wl.rows * ws * args.steps
    line = {
        "metric": "rows/sec + achieved HBM GB/s, TPC-H Q1/Q3 SF100, 1/2/4/8 GPU vs CPU",
        "value": round(total_rows / dt, 1), "unit": "rows/s", "n_gpus": ws, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if (distributed and args.scaling == "strong") else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl.name, "description": wl.desc, "rows_per_gpu": wl.rows, "algorithmic_bytes_per_gpu_step": wl.algo_bytes,
                   "parallelism": "single GPU"},
        "whole_query_GBps_per_gpu": round(wl.algo_bytes * args.steps / dt / 1e9, 1),
        "cold_first_step_ms": None if cold_ms is None else round(cold_ms, 2),      # first step of the process; EXCLUDES the pool mapping below (pool.map_ms, done at start-up)
        "pool": pool,
        "one_shot_ms": one_shot_ms(pl, wl),
        "step_ms": getattr(timed, "last_step_ms", None),      # every timed step, in order: a stall of the box shows here, not only in the mean
        **step_spread(getattr(timed, "last_step_ms", None), wl.rows * ws),
        "roofline": roofline(stats, wl, args.steps),
        "kernels": _kernels(stats, 8),
    }
    want_cpu = rank == 0 and ws == 1 and not args.no_cpu
    if rank == 0:
        emit(line, not want_cpu)                 # the headline is safe from here on (a guard cut waits for the CPU baseline)
    if want_cpu:
        try:
            if args.workload == "q1" and getattr(wl, "native_seed", None) is not None:
                base, ver = cpu_baseline_q1(args.cpu_seconds, rows=wl.rows, seed=wl.native_seed, gpu_result=res)
                if ver and "pending_prefix" in ver:    # the oracle covered a prefix within its budget: run the query on exactly those rows
                    from polars_amd import queries
                    want = ver.pop("pending_prefix")
                    got = queries.q1(wl.frame.slice(0, ver["rows"]).lazy()).collect().to_dict()
                    ver = dict(compare_q1(got, want), rows=ver["rows"], rtol=VERIFY_RTOL,
                               against="oracle (orc_q1_streaming over the generator's host twin) on the first rows of the timed input, the library re-run on that slice")
                if ver is not None:
                    if ver.get("ok") is False:
                        print(f"[bench] VERIFICATION FAILED for {wl.name}: {ver}", file=sys.stderr)
                    line["verified"] = ver
            elif args.workload == "q1":
                base, _ = cpu_baseline_q1(args.cpu_seconds)
            else:
                base, _ = cpu_baseline_q1(args.cpu_seconds)
                line["verified"] = _verify(wl, res, 60.0)
            line["cpu_baseline"] = base
        except Exception as e:
            line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        emit(line, True)
    elif rank == 0 and ws == 1 and args.workload != "q1" and os.environ.get("PLX_BENCH_VERIFY", "1") != "0":
        line["verified"] = _verify(wl, res, 60.0)
    if rank == 0 and not args.no_extras and ws == 1:
        extras = {}
        line["extras"] = extras
        k2 = max(3, args.steps // 4)
        for vname, vstep in wl.variants.items():
            try:
                dv, sv, _, _ = timed(pl, Workload(vname, wl.rows, wl.algo_bytes, vstep, wl.kernel, wl.desc), k2, 1, False)
                extras[vname] = {"rows_per_s": round(wl.rows * k2 / dv, 1), "ms_per_step": round(dv / k2 * 1e3, 3), "kernels": _kernels(sv, 6)}
            except Exception as e:
                extras[vname] = {"error": f"{type(e).__name__}: {e}"[:300]}
            emit(line)
        del wl, res
        release_memory(pl)
        # The all-rows host checks of the secondary workloads run AFTER all of them have been timed: a check builds the workload's host twin (tens of gigabytes
        # of numpy arrays, all host threads in the oracle) and what it leaves behind in the host's memory management cost the NEXT workload a 10-20 ms stall
        # of the main thread in one of its first steps and ~5 % on the others (the three-table Q3 behind the hashed-key Q3's check: one step of 13-22 ms in
        # every full run, none with PLX_BENCH_VERIFY=0).  The results wait on the device (a few MB each) until then.
        pending_checks = []
        only = [w for w in os.environ.get("PLX_BENCH_EXTRAS", "").split(",") if w]          # measurement: only these secondary workloads
        for name in [w for w in EXTRA_WORKLOADS if w != args.workload and (not only or w in only)]:
            try:
                w2 = make_workload(pl, name, int(os.environ.get("PLX_BENCH_EXTRAS_ROWS", "0")), seed=20)
                # three warm-up steps: config 3's second run is the first with learned key statistics (new buffer sizes), and the three-table Q3's THIRD run still maps
                # fresh result buffers (the two before it hold theirs): one 10 ms kernel on first touch, in every full run, on the first timed step
                d2, s2, r2, c2 = timed(pl, w2, k2, 3, False)
                if getattr(w2, "out_row_bytes", 0) and hasattr(r2, "height"):
                    w2.algo_bytes += int(r2.height) * w2.out_row_bytes          # required inputs once + the output once
                extras[w2.name] = {"rows_per_s": round(w2.rows * k2 / d2, 1), "ms_per_step": round(d2 / k2 * 1e3, 3), "cold_first_step_ms": None if c2 is None else round(c2, 2),
                                   "one_shot_ms": one_shot_ms(pl, w2), "step_ms": list(getattr(timed, "last_step_ms", [])), **step_spread(getattr(timed, "last_step_ms", []), w2.rows),
                                   "whole_query_GBps": round(w2.algo_bytes * k2 / d2 / 1e9, 1), "roofline": roofline(s2, w2, k2), "kernels": _kernels(s2, 6)}
                emit(line)
                for vname, vstep in w2.variants.items():
                    wv = Workload(vname, w2.rows, w2.algo_bytes, vstep, w2.kernel, w2.desc)
                    dv, sv, _, _ = timed(pl, wv, k2, 1, False)
                    extras[vname] = {"rows_per_s": round(w2.rows * k2 / dv, 1), "ms_per_step": round(dv / k2 * 1e3, 3), "kernels": _kernels(sv, 6)}
                if hasattr(w2, "jit_before"):
                    j1 = jit_stats(pl)
                    extras[w2.name]["jit"] = {"kernels_compiled_or_loaded": j1[0] - w2.jit_before[0], "compile_ms": round(j1[1] - w2.jit_before[1], 1),
                                              "note": "compile_ms > 0: hiprtc ran inside cold_first_step_ms (empty disk cache); 0: the code object came from the disk cache"}
                try:
                    extras[w2.name]["plan"] = pl.last_plan()[:600]
                except Exception:
                    pass
                if hasattr(r2, "height"):
                    extras[w2.name]["result_rows"] = int(r2.height)
                    if not getattr(w2, "big_result", False) and hasattr(r2, "_download_all"):
                        # a collect() of the reference ends with a host frame; these steps leave theirs in HBM -- what the download adds, measured beside `ms` (round-5 review, weak 8)
                        dls = []
                        for _ in range(2):      # (the first download of a result also pays for the fresh numpy arrays it lands in: first-touch page faults of tens of MB)
                            t_dl = time.perf_counter(); r2._download_all(); dls.append((time.perf_counter() - t_dl) * 1e3)
                        extras[w2.name]["result_download_ms"] = round(min(dls), 3)
                        extras[w2.name]["result_download_first_ms"] = round(dls[0], 3)
                if os.environ.get("PLX_BENCH_VERIFY", "1") != "0":
                    pending_checks.append((w2.name, w2.verify, r2))       # checked after every secondary workload has been timed (below)
                del w2, r2
            except Exception as e:  # a secondary workload must never take the headline line down
                extras[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            release_memory(pl)
            emit(line)
        # Q3 on SHUFFLED inputs (round-1 review, item 6): the headline Q3 runs on dbgen row order, where an order's lines are adjacent and the
        # probe's late materialisation skips whole cache lines; shuffled rows are the adversarial case for both.  Torch generators (the
        # library's generator has no shuffled mode), so no host-twin verification: the group count is compared with the ordered run's.
        if os.environ.get("PLX_BENCH_Q3_SHUFFLED", "1") != "0":
            prev = os.environ.get("PLX_Q3_SHUFFLED")
            try:
                os.environ["PLX_Q3_SHUFFLED"] = "1"
                w3 = make_workload(pl, "q3", 0, seed=20)
                w3.name = "tpch_q3_sf100_shuffled_inputs"          # (also keeps the ordered run's PMC traffic figure off this line)
                d3, s3, r3, c3 = timed(pl, w3, k2, 3, False)
                ordered = extras.get("tpch_q3_sf100", {})
                extras["tpch_q3_sf100_shuffled_inputs"] = {
                    "rows_per_s": round(w3.rows * k2 / d3, 1), "ms_per_step": round(d3 / k2 * 1e3, 3), "cold_first_step_ms": None if c3 is None else round(c3, 2),
                    "vs_ordered_inputs": (round(d3 / k2 * 1e3 / ordered["ms_per_step"], 2) if ordered.get("ms_per_step") else None),
                    "groups": int(r3.height) if hasattr(r3, "height") else None, "roofline": roofline(s3, w3, k2), "kernels": _kernels(s3, 6),
                    "note": "same query over the SAME rows as tpch_q3_sf100 (the library's generator), both tables in a seeded random row order (device gather)"}
                if os.environ.get("PLX_BENCH_VERIFY", "1") != "0":
                    pending_checks.append(("tpch_q3_sf100_shuffled_inputs", w3.verify, r3))
                del w3, r3
            except Exception as e:
                extras["tpch_q3_sf100_shuffled_inputs"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                if prev is None:
                    os.environ.pop("PLX_Q3_SHUFFLED", None)
                else:
                    os.environ["PLX_Q3_SHUFFLED"] = prev
            release_memory(pl)
            emit(line)
        for cname, cfn, cres in pending_checks:
            if cname in extras and "error" not in extras[cname]:
                shim = Workload(cname, 0, 0, None, "", "", verify=cfn)
                extras[cname]["verified"] = _verify(shim, cres, float(os.environ.get("PLX_BENCH_VERIFY_BUDGET_S", "40")))
                emit(line)
        del pending_checks
        release_memory(pl)
        for name in [w for w in LATE_WORKLOADS if not only or w in only]:
            try:
                w2 = make_workload(pl, name, int(os.environ.get("PLX_BENCH_EXTRAS_ROWS", "0")), seed=20)
                d2, s2, r2, c2 = timed(pl, w2, k2, 2, False)
                out_rows = int(r2.height) if hasattr(r2, "height") else int(len(r2))
                w2.algo_bytes += out_rows * getattr(w2, "out_row_bytes", 0)
                w2.scope = "operator"
                extras[w2.name] = {"rows_per_s": round(w2.rows * k2 / d2, 1), "ms_per_step": round(d2 / k2 * 1e3, 3), "cold_first_step_ms": None if c2 is None else round(c2, 2),
                                   "step_ms": list(getattr(timed, "last_step_ms", [])), **step_spread(getattr(timed, "last_step_ms", []), w2.rows), "result_rows": out_rows,
                                   "result": "left in HBM (a multi-gigabyte frame: the next operator's input)",
                                   "whole_query_GBps": round(w2.algo_bytes * k2 / d2 / 1e9, 1), "roofline": roofline(s2, w2, k2), "kernels": _kernels(s2, 6)}
                emit(line)
                if os.environ.get("PLX_BENCH_VERIFY", "1") != "0":
                    extras[w2.name]["verified"] = _verify(w2, r2, float(os.environ.get("PLX_BENCH_VERIFY_BUDGET_S", "40")))
                del w2, r2
            except Exception as e:
                extras[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            release_memory(pl)
            emit(line)
        if os.environ.get("PLX_BENCH_E2E", "1") != "0":
            try:
                extras["end_to_end_with_h2d"] = end_to_end_q1(pl, 60_000_000)
            except Exception as e:
                extras["end_to_end_with_h2d"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            emit(line)
        # last (nothing after it can be cut short by it): the scan in front of the path, SURVEY.md 8(f) row 3
        if os.environ.get("PLX_BENCH_SCAN", "1") != "0":
            try:
                extras["parquet_ipc_scan_2e7_rows"] = scan_extra(pl, 20_000_000)
            except Exception as e:
                extras["parquet_ipc_scan_2e7_rows"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            emit(line)
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
