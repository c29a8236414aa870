"""Synthetic inputs of the benchmark configurations:
TPC-H-style lineitem / orders columns (dbgen distributions restated, column types from the
reference's examples/datasets/pds_heads/*.feather) and the config 2 / 3 / 5 frames.

Two generators with the same distributions: numpy (host; parity tests and the CPU-baseline
sample) and torch (device-resident; bench.py at SF100, where the columns must already sit in
HBM when the timed region starts).  Values differ between the two (different RNGs); parity is
always checked on the same arrays.
"""
from __future__ import annotations

import datetime as _dt
from typing import Dict, Tuple

import numpy as np

DAY_US = 86_400_000_000
FLAGS = ["A", "N", "R"]      # l_returnflag dictionary (codes 0,1,2)
STATUS = ["F", "O"]          # l_linestatus dictionary (codes 0,1)
SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]   # c_mktsegment dictionary (codes 0..4)
CUSTOMER_Q3_COLS = ["c_custkey", "c_mktsegment"]
LINEITEM_Q1_COLS = ["l_shipdate", "l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_discount", "l_tax"]
LINEITEM_Q3_COLS = ["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"]
ORDERS_Q3_COLS = ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"]
Q1_BYTES_PER_ROW = 8 + 1 + 1 + 8 * 4   # 42 B/row
Q3_LINEITEM_BYTES_PER_ROW = 32
Q3_ORDERS_BYTES_PER_ROW = 32


def us(y: int, m: int, d: int) -> int:
    """Datetime[us] physical value (microseconds since epoch) of a date."""
    return (_dt.date(y, m, d) - _dt.date(1970, 1, 1)).days * DAY_US


START = us(1992, 1, 1)
END_ORDERS = us(1998, 8, 2)
CURRENT = us(1995, 6, 17)


def logical_dtypes(pl) -> Dict[str, object]:
    return {"l_shipdate": pl.Datetime, "o_orderdate": pl.Datetime, "l_returnflag": pl.Categorical(FLAGS, pl.UInt8),
            "l_linestatus": pl.Categorical(STATUS, pl.UInt8), "c_mktsegment": pl.Categorical(SEGMENTS, pl.UInt8)}


def n_customers_for(n_orders: int) -> int:
    """Customers referenced by `n_orders` orders (o_custkey is uniform in [1, n_customers]; TPC-H: 10 orders per customer)."""
    return max(2, n_orders // 10)


def customer_host(n_customers: int, seed: int = 10) -> Dict[str, np.ndarray]:
    """numpy customer table (parity tests): dense keys 1..n, uniform market segment codes."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    return {"c_custkey": np.arange(1, n_customers + 1, dtype=np.int64), "c_mktsegment": rng.integers(0, len(SEGMENTS), n_customers).astype(np.uint8)}


def to_frame(pl, cols: Dict[str, np.ndarray], names):
    """Upload host columns as a DataFrame with the logical dtypes of the TPC-H schema."""
    lt = logical_dtypes(pl)
    return pl.DataFrame([pl.Series(n, cols[n], dtype=lt.get(n)) for n in names])


# ------------------------------------------------------------------------ numpy ----
def _line_columns_host(rng: np.random.Generator, shipdate: np.ndarray) -> Dict[str, np.ndarray]:
    n = len(shipdate)
    qty = rng.integers(1, 51, n, dtype=np.int64)
    price = np.round(qty.astype(np.float64) * rng.integers(90_000, 210_000, n).astype(np.float64) / 100.0, 2)
    disc = rng.integers(0, 11, n).astype(np.float64) / 100.0
    tax = rng.integers(0, 9, n).astype(np.float64) / 100.0
    receipt = shipdate + rng.integers(1, 31, n, dtype=np.int64) * DAY_US
    flag = np.where(receipt <= CURRENT, np.where(rng.random(n) < 0.5, 0, 2), 1).astype(np.uint8)
    status = (shipdate > CURRENT).astype(np.uint8)
    return {"l_shipdate": shipdate, "l_returnflag": flag, "l_linestatus": status, "l_quantity": qty, "l_extendedprice": price,
            "l_discount": disc, "l_tax": tax}


def lineitem_host(n_rows: int, seed: int = 10) -> Dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    ship = START + rng.integers(1, 2526 + 121, n_rows, dtype=np.int64) * DAY_US
    return _line_columns_host(rng, ship)


def orders_lineitem_host(n_orders: int, seed: int = 10, ordered: bool = False) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """orders (sparse keys: 8 of every 32 used) and its lineitem (1-7 lines per order,
    l_shipdate = o_orderdate + U[1,121] days).  ordered=True keeps dbgen's row order (both tables
    ascending in orderkey); the default shuffles both tables (adversarial for hash probing)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    i = np.arange(n_orders, dtype=np.int64)
    okey = (i // 8) * 32 + (i % 8) + 1
    if not ordered:
        okey = okey[rng.permutation(n_orders)]
    odate = START + rng.integers(0, (END_ORDERS - START) // DAY_US + 1, n_orders, dtype=np.int64) * DAY_US
    orders = {"o_orderkey": okey, "o_custkey": rng.integers(1, max(2, n_orders // 10) + 1, n_orders, dtype=np.int64),
              "o_orderdate": odate, "o_shippriority": np.zeros(n_orders, dtype=np.int64)}
    cnt = rng.integers(1, 8, n_orders)
    lkey = np.repeat(okey, cnt)
    ship = np.repeat(odate, cnt) + rng.integers(1, 122, len(lkey), dtype=np.int64) * DAY_US
    if not ordered:
        shuffle = rng.permutation(len(lkey))
        lkey, ship = lkey[shuffle], ship[shuffle]
    li = _line_columns_host(rng, ship)
    li["l_orderkey"] = lkey
    return orders, li


def cfg2_host(n: int, null_frac: float = 0.0):
    a = np.random.Generator(np.random.PCG64(1)).integers(0, 2 ** 31, n, dtype=np.int64)
    x = np.random.Generator(np.random.PCG64(2)).uniform(0, 100, n)
    y = np.random.Generator(np.random.PCG64(3)).uniform(0, 1, n)
    xv = None
    if null_frac > 0:
        xv = np.ones(n, dtype=bool)
        xv[np.random.Generator(np.random.PCG64(4)).choice(n, int(n * null_frac), replace=False)] = False
    return a, x, y, xv


def cfg3_host(n: int, n_keys: int = 1_000_000, zipf: float = 0.0):
    if zipf > 0:
        key = (np.random.Generator(np.random.PCG64(6)).zipf(zipf, n) - 1) % n_keys
        key = key.astype(np.int64)
    else:
        key = np.random.Generator(np.random.PCG64(5)).integers(0, n_keys, n, dtype=np.int64)
    v = np.random.Generator(np.random.PCG64(7)).integers(0, 1000, n, dtype=np.int64)
    return key, v


def cfg5_host(n: int, n_keys: int = 1_000_000):
    """Dictionary codes of keys "id%010d" (H2O id3 style); the dictionary itself is
    ["id%010d" % i for i in 1..n_keys] and stays on the host."""
    codes = np.random.Generator(np.random.PCG64(8)).integers(0, n_keys, n, dtype=np.uint32)
    v = np.random.Generator(np.random.PCG64(2)).uniform(0, 100, n)
    return codes, v


# ----------------------------------------------------------------------- native ----
def lineitem_native_host(row0: int, n: int, seed: int = 10) -> Dict[str, np.ndarray]:
    """Rows [row0, row0 + n) of the library's counter-based lineitem generator, evaluated on the host (no GPU):
    plx_datagen_lineitem_q1_host, the CPU twin of the kernel behind lineitem_native."""
    import ctypes as C

    from . import _ffi as F
    out = {"l_shipdate": np.zeros(n, np.int64), "l_returnflag": np.zeros(n, np.uint8), "l_linestatus": np.zeros(n, np.uint8),
           "l_quantity": np.zeros(n, np.int64), "l_extendedprice": np.zeros(n, np.float64), "l_discount": np.zeros(n, np.float64),
           "l_tax": np.zeros(n, np.float64)}
    ptrs = [C.c_void_p(out[c].ctypes.data) if n else C.c_void_p(0) for c in LINEITEM_Q1_COLS]
    F.check(F.lib().plx_datagen_lineitem_q1_host(row0, n, seed, *ptrs))
    return out


def lineitem_native(pl, n_rows: int, seed: int = 10):
    """TPC-H Q1 lineitem columns generated straight into HBM by the library's own kernel (kernels_datagen.hip): no
    torch kernels, no host staging.  Returns a DataFrame with the logical dtypes of the TPC-H schema."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    hs = (C.c_uint64 * 7)()
    F.check(F.lib().plx_datagen_lineitem_q1(n_rows, seed, hs))
    lt = logical_dtypes(pl)
    phys = {"l_shipdate": pl.Int64, "l_returnflag": pl.UInt8, "l_linestatus": pl.UInt8, "l_quantity": pl.Int64, "l_extendedprice": pl.Float64,
            "l_discount": pl.Float64, "l_tax": pl.Float64}
    return pl.DataFrame([pl.Series._from_handle(c, hs[i], lt.get(c, phys[c])) for i, c in enumerate(LINEITEM_Q1_COLS)])


def orders_lineitem_native_host(order0: int, n: int, n_orders_total: int, seed: int = 10):
    """Orders [order0, order0 + n) of the library's Q3 generator and their lines, on the host (no GPU)."""
    import ctypes as C

    from . import _ffi as F
    o = {"o_orderkey": np.zeros(n, np.int64), "o_custkey": np.zeros(n, np.int64), "o_orderdate": np.zeros(n, np.int64), "o_shippriority": np.zeros(n, np.int64)}
    cnt = np.zeros(n, np.uint32)
    cap = 7 * n
    li = {"l_orderkey": np.zeros(cap, np.int64), "l_extendedprice": np.zeros(cap, np.float64), "l_discount": np.zeros(cap, np.float64), "l_shipdate": np.zeros(cap, np.int64)}
    nl = C.c_int64()
    p = lambda a: C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)
    F.check(F.lib().plx_datagen_orders_lineitem_host(order0, n, n_orders_total, seed, p(o["o_orderkey"]), p(o["o_custkey"]), p(o["o_orderdate"]), p(cnt), cap,
                                                     p(li["l_orderkey"]), p(li["l_extendedprice"]), p(li["l_discount"]), p(li["l_shipdate"]), C.byref(nl)))
    return o, {k: v[:nl.value] for k, v in li.items()}, cnt


def orders_lineitem_native(pl, n_orders: int, seed: int = 10):
    """TPC-H Q3 orders / lineitem generated in HBM by the library (dbgen row order) -> (orders frame, lineitem frame)."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    ho, hl = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
    F.check(F.lib().plx_datagen_orders_lineitem(n_orders, seed, ho, hl))
    lt = logical_dtypes(pl)
    O = pl.DataFrame([pl.Series._from_handle(c, ho[i], lt.get(c, pl.Int64)) for i, c in enumerate(ORDERS_Q3_COLS)])
    ldt = {"l_orderkey": pl.Int64, "l_extendedprice": pl.Float64, "l_discount": pl.Float64, "l_shipdate": pl.Datetime}
    L = pl.DataFrame([pl.Series._from_handle(c, hl[i], lt.get(c, ldt[c])) for i, c in enumerate(LINEITEM_Q3_COLS)])
    return O, L


def customer_native_host(row0: int, n: int, seed: int = 10) -> Dict[str, np.ndarray]:
    """Rows [row0, row0 + n) of the library's customer generator on the host (plx_datagen_customer_host)."""
    import ctypes as C

    from . import _ffi as F
    out = {"c_custkey": np.zeros(n, np.int64), "c_mktsegment": np.zeros(n, np.uint8)}
    F.check(F.lib().plx_datagen_customer_host(row0, n, seed, C.c_void_p(out["c_custkey"].ctypes.data if n else 0), C.c_void_p(out["c_mktsegment"].ctypes.data if n else 0)))
    return out


def customer_native(pl, n_customers: int, seed: int = 10):
    """customer (c_custkey, c_mktsegment) generated in HBM by the library -> DataFrame."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    hs = (C.c_uint64 * 2)()
    F.check(F.lib().plx_datagen_customer(n_customers, seed, hs))
    lt = logical_dtypes(pl)
    return pl.DataFrame([pl.Series._from_handle("c_custkey", hs[0], pl.Int64), pl.Series._from_handle("c_mktsegment", hs[1], lt["c_mktsegment"])])


_NP_OF = {"Int64": np.int64, "UInt32": np.uint32, "Float64": np.float64}


def uniform_native_host(dtype_name: str, row0: int, n: int, seed: int, stream: int, lo: int, hi: int, scale: float = 1.0) -> np.ndarray:
    import ctypes as C

    from . import _ffi as F
    phys = {"Int64": F.I64, "UInt32": F.U32, "Float64": F.F64}[dtype_name]
    out = np.zeros(n, _NP_OF[dtype_name])
    F.check(F.lib().plx_datagen_uniform_host(phys, row0, n, seed, stream, lo, hi, scale, C.c_void_p(out.ctypes.data) if n else C.c_void_p(0)))
    return out


def uniform_native(pl, name: str, dtype, n: int, seed: int, stream: int, lo: int, hi: int, scale: float = 1.0):
    """One uniform device column from the library's generator: lo + floor(U * (hi - lo)) (Float64: times `scale`)."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    h = C.c_uint64()
    F.check(F.lib().plx_datagen_uniform(dtype.physical, n, seed, stream, lo, hi, scale, C.byref(h)))
    return pl.Series._from_handle(name, h.value, dtype)


def zipf_x0_q62(n_keys: int) -> int:
    """x0 = n_keys ** -0.1 as the 62-bit fixed-point integer the zipf generator takes (plx_datagen_zipf: key = floor(1 / x^10) - 1, x uniform in [x0, 1))."""
    return int(round(float(n_keys) ** -0.1 * (1 << 62)))


def zipf_native(pl, name: str, n: int, seed: int, stream: int, n_keys: int):
    """A heavy-tailed Int64 key column in [0, n_keys) from the library's generator (density ~ k^-1.1: config 3's Zipf variant)."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    h = C.c_uint64()
    F.check(F.lib().plx_datagen_zipf(n, seed, stream, zipf_x0_q62(n_keys), n_keys, C.byref(h)))
    return pl.Series._from_handle(name, h.value, pl.Int64)


def zipf_native_host_mt(row0: int, n: int, seed: int, stream: int, n_keys: int, threads=None) -> np.ndarray:
    """Host twin of zipf_native: rows [row0, row0 + n), bit-identical (integer fixed point), over the host threads."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from . import _ffi as F
    out = np.empty(n, np.int64)
    lib, x0 = F.lib(), zipf_x0_q62(n_keys)

    def work(bl):
        b, m = bl
        F.check(lib.plx_datagen_zipf_host(row0 + b, m, seed, stream, x0, n_keys, C.c_void_p(out.ctypes.data + b * 8)))
    nt = _host_threads(threads)
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(work, _split(n, nt * 4, 4096)))
    return out


def id_views_native(pl, name: str, n: int, seed: int, stream: int, lo: int, hi: int):
    """A Utf8View key column generated in HBM: row i holds the (inline, 12-byte) view of "id%010d" % uniform_value(seed, stream, i, lo, hi)
    -> UInt64 Series of 2 n words (feed it to pl.Series.from_device_views).  Host twin: uniform_native_host("Int64", ...) gives the
    values, the strings are "id%010d" % value."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    h = C.c_uint64()
    F.check(F.lib().plx_datagen_id_views(n, seed, stream, lo, hi, C.byref(h)))
    return pl.Series._from_handle(name, h.value, pl.UInt64)


def long_id_views_native(pl, n: int, seed: int, stream: int, lo: int, hi: int):
    """A Utf8View key column of 20-byte strings generated in HBM: row i = "id%010d-longkey" % uniform_value(seed, stream, i, lo, hi).  The views are NOT inline: {20, prefix,
    buffer 0, offset} into a pool holding each distinct string once -> (views: UInt64 Series of 2 n words, data: UInt8 Series) for pl.Series.from_device_views."""
    import ctypes as C

    from . import _ffi as F
    F.ensure_init()
    hv, hd = C.c_uint64(), C.c_uint64()
    F.check(F.lib().plx_datagen_long_id_views(n, seed, stream, lo, hi, C.byref(hv), C.byref(hd)))
    return pl.Series._from_handle("views", hv.value, pl.UInt64), pl.Series._from_handle("data", hd.value, pl.UInt8)


# ---- multi-threaded host twins (ctypes releases the GIL): full-size checks in bench.py / tests -------------------------
def _host_threads(threads=None) -> int:
    import os
    return max(1, int(threads or min(64, os.cpu_count() or 1)))


def _split(n: int, parts: int, align: int = 1):
    per = max(align, -(-n // max(1, parts)))
    per = -(-per // align) * align
    return [(b, min(per, n - b)) for b in range(0, n, per)]


def lineitem_native_host_mt(row0: int, n: int, seed: int = 10, threads=None) -> Dict[str, np.ndarray]:
    """lineitem_native_host evaluated by `threads` host threads writing disjoint slices of the same arrays."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from . import _ffi as F
    out = {"l_shipdate": np.empty(n, np.int64), "l_returnflag": np.empty(n, np.uint8), "l_linestatus": np.empty(n, np.uint8),
           "l_quantity": np.empty(n, np.int64), "l_extendedprice": np.empty(n, np.float64), "l_discount": np.empty(n, np.float64),
           "l_tax": np.empty(n, np.float64)}
    lib = F.lib()

    def work(bl):
        b, m = bl
        ptrs = [C.c_void_p(out[c].ctypes.data + b * out[c].itemsize) for c in LINEITEM_Q1_COLS]
        F.check(lib.plx_datagen_lineitem_q1_host(row0 + b, m, seed, *ptrs))
    nt = _host_threads(threads)
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(work, _split(n, nt * 4, 4096)))
    return out


def uniform_native_host_mt(dtype_name: str, row0: int, n: int, seed: int, stream: int, lo: int, hi: int, scale: float = 1.0, threads=None) -> np.ndarray:
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from . import _ffi as F
    phys = {"Int64": F.I64, "UInt32": F.U32, "Float64": F.F64}[dtype_name]
    out = np.empty(n, _NP_OF[dtype_name])
    lib = F.lib()

    def work(bl):
        b, m = bl
        F.check(lib.plx_datagen_uniform_host(phys, row0 + b, m, seed, stream, lo, hi, scale, C.c_void_p(out.ctypes.data + b * out.itemsize)))
    nt = _host_threads(threads)
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(work, _split(n, nt * 4, 4096)))
    return out


def orders_lineitem_native_host_mt(order0: int, n: int, n_orders_total: int, seed: int = 10, threads=None):
    """orders_lineitem_native_host over sub-blocks of orders in parallel, concatenated in order: (orders, lineitem, n_lines)."""
    from concurrent.futures import ThreadPoolExecutor
    nt = _host_threads(threads)
    blocks = _split(n, nt * 2, 1024)
    with ThreadPoolExecutor(nt) as ex:
        parts = list(ex.map(lambda bl: orders_lineitem_native_host(order0 + bl[0], bl[1], n_orders_total, seed), blocks))
    if not parts:
        return orders_lineitem_native_host(order0, 0, n_orders_total, seed)
    o = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]}
    li = {k: np.concatenate([p[1][k] for p in parts]) for k in parts[0][1]}
    return o, li, np.concatenate([p[2] for p in parts])


# ------------------------------------------------------------------------ torch ----
def _line_columns_device(torch, g, shipdate):
    n = shipdate.numel()
    dev = shipdate.device
    qty = torch.randint(1, 51, (n,), generator=g, device=dev, dtype=torch.int64)
    price = (qty.to(torch.float64) * torch.randint(90_000, 210_000, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.float64) / 100.0)
    price = torch.round(price * 100.0) / 100.0
    disc = torch.randint(0, 11, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.float64) / 100.0
    tax = torch.randint(0, 9, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.float64) / 100.0
    receipt = shipdate + torch.randint(1, 31, (n,), generator=g, device=dev, dtype=torch.int64) * DAY_US
    coin = torch.rand((n,), generator=g, device=dev) < 0.5
    flag = torch.where(receipt <= CURRENT, torch.where(coin, 0, 2), 1).to(torch.uint8)
    del receipt, coin
    status = (shipdate > CURRENT).to(torch.uint8)
    return {"l_shipdate": shipdate, "l_returnflag": flag, "l_linestatus": status, "l_quantity": qty, "l_extendedprice": price,
            "l_discount": disc, "l_tax": tax}


def lineitem_device(n_rows: int, seed: int = 10, device: str = "cuda"):
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    ship = START + torch.randint(1, 2526 + 121, (n_rows,), generator=g, device=device, dtype=torch.int64) * DAY_US
    return _line_columns_device(torch, g, ship)


def orders_lineitem_device(n_orders: int, seed: int = 10, device: str = "cuda", ordered: bool = True):
    """Device-resident orders / lineitem.  ordered=True (default) = dbgen row order: orders ascending in
    o_orderkey, lineitem ascending in l_orderkey -- what TPC-H / PDS-H data looks like; ordered=False
    shuffles both tables."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    i = torch.arange(n_orders, device=device, dtype=torch.int64)
    okey = (i // 8) * 32 + (i % 8) + 1
    if not ordered:
        okey = okey[torch.randperm(n_orders, generator=g, device=device)]
    odate = START + torch.randint(0, (END_ORDERS - START) // DAY_US + 1, (n_orders,), generator=g, device=device, dtype=torch.int64) * DAY_US
    orders = {"o_orderkey": okey, "o_custkey": torch.randint(1, max(2, n_orders // 10) + 1, (n_orders,), generator=g, device=device, dtype=torch.int64),
              "o_orderdate": odate, "o_shippriority": torch.zeros(n_orders, device=device, dtype=torch.int64)}
    cnt = torch.randint(1, 8, (n_orders,), generator=g, device=device, dtype=torch.int64)
    lkey = torch.repeat_interleave(okey, cnt)
    ship = torch.repeat_interleave(odate, cnt) + torch.randint(1, 122, (lkey.numel(),), generator=g, device=device, dtype=torch.int64) * DAY_US
    if not ordered:
        shuffle = torch.randperm(lkey.numel(), generator=g, device=device)
        lkey, ship = lkey[shuffle], ship[shuffle]
        del shuffle
    del cnt
    n = lkey.numel()
    qty = torch.randint(1, 51, (n,), generator=g, device=device, dtype=torch.int64)
    price = torch.round(qty.to(torch.float64) * torch.randint(90_000, 210_000, (n,), generator=g, device=device, dtype=torch.int64).to(torch.float64)) / 100.0
    disc = torch.randint(0, 11, (n,), generator=g, device=device, dtype=torch.int64).to(torch.float64) / 100.0
    li = {"l_orderkey": lkey, "l_extendedprice": price, "l_discount": disc, "l_shipdate": ship}
    return orders, li


def frame_from_torch(pl, cols, names):
    """Wrap device tensors as a DataFrame without copying (plx_column_from_device)."""
    lt = logical_dtypes(pl)
    return pl.DataFrame([pl.Series.from_torch(n, cols[n], dtype=lt.get(n)) for n in names])
