"""Independent numpy evaluations of the benchmark queries (no oracle, no GPU): used by the GPU tests at sizes the oracle
does not cover in seconds, and themselves checked against the oracle on the CPU (tests/test_oracle_golden.py)."""
import numpy as np


def q1_numpy(cols, cutoff):
    """TPC-H Q1 -> {(flag_code, status_code): {aggregate: value}} with plain masked numpy reductions."""
    sel = cols["l_shipdate"] <= cutoff
    gid = cols["l_returnflag"].astype(np.int64) * 4 + cols["l_linestatus"].astype(np.int64)
    out = {}
    for g in np.unique(gid[sel]).tolist():
        m = sel & (gid == g)
        qty, price, disc, tax = cols["l_quantity"][m], cols["l_extendedprice"][m], cols["l_discount"][m], cols["l_tax"][m]
        dp = price * (1 - disc)
        n = int(m.sum())
        out[(g // 4, g % 4)] = {"count_order": n, "sum_qty": int(qty.sum()), "sum_base_price": float(price.sum()), "sum_disc_price": float(dp.sum()),
                                "sum_charge": float((dp * (1 + tax)).sum()), "avg_qty": float(qty.sum()) / n, "avg_price": float(price.sum()) / n,
                                "avg_disc": float(disc.sum()) / n}
    return out
