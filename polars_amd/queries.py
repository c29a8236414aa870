"""The benchmark queries of BASELINE.json / SURVEY.md section 8(d) and Appendix A, written
in the mirror API.  Shared by bench.py, the parity tests and tools/gen_fused_shapes.py."""
from __future__ import annotations

import datetime as dt

from . import expr as E

Q1_CUTOFF = dt.datetime(1998, 9, 2)
Q3_DATE = dt.datetime(1995, 3, 15)


def cfg1(lf, k=2 ** 30):
    """filter(a > k).select(a.sum())  -- BASELINE config 1"""
    return lf.filter(E.col("a") > k).select(E.col("a").sum())


def cfg2(lf, k=2 ** 30):
    """filter(a > k).select((x*(1-y)).sum(), x.mean(), a.sum())  -- BASELINE config 2"""
    return lf.filter(E.col("a") > k).select((E.col("x") * (1 - E.col("y"))).sum().alias("xy"), E.col("x").mean().alias("x_mean"),
                                            E.col("a").sum().alias("a_sum"))


def cfg3(lf):
    """group_by(key).agg(v.sum(), v.count())  -- BASELINE config 3"""
    return lf.group_by("key").agg(E.col("v").sum().alias("v_sum"), E.col("v").count().alias("v_count"))


def cfg3w(lf):
    """group_by(k1, k2).agg(v.sum(), v.count())  -- config 3 on a two-column (wide) key: the reference row-encodes it (hash_keys.rs:334 RowEncodedKeys)"""
    return lf.group_by("k1", "k2").agg(E.col("v").sum().alias("v_sum"), E.col("v").count().alias("v_count"))


def cfg5(lf):
    """group_by(k).agg(v.sum(), v.mean()) on dictionary-encoded string keys -- BASELINE config 5"""
    return lf.group_by("k").agg(E.col("v").sum().alias("v_sum"), E.col("v").mean().alias("v_mean"))


def q1(lineitem, cutoff=Q1_CUTOFF):
    """TPC-H Q1; the final sort by (flag, status) happens on the host."""
    c = E.col
    disc_price = c("l_extendedprice") * (1 - c("l_discount"))
    return (lineitem.filter(c("l_shipdate") <= cutoff)
            .group_by("l_returnflag", "l_linestatus")
            .agg(c("l_quantity").sum().alias("sum_qty"),
                 c("l_extendedprice").sum().alias("sum_base_price"),
                 disc_price.sum().alias("sum_disc_price"),
                 (disc_price * (1 + c("l_tax"))).sum().alias("sum_charge"),
                 c("l_quantity").mean().alias("avg_qty"),
                 c("l_extendedprice").mean().alias("avg_price"),
                 c("l_discount").mean().alias("avg_disc"),
                 E.len().alias("count_order")))


def q6(lineitem, year: int = 1994, discount: float = 0.06, quantity: int = 24):
    """TPC-H Q6 (forecasting revenue change): one filter + one sum, the shape of BASELINE config 2 on lineitem's columns --
    sum(l_extendedprice * l_discount) over shipdate in [year, year + 1), discount within +-0.01, quantity below the bound."""
    c = E.col
    lo, hi = dt.datetime(year, 1, 1), dt.datetime(year + 1, 1, 1)
    return (lineitem.filter(c("l_shipdate").is_between(lo, hi, closed="left") & c("l_discount").is_between(round(discount - 0.01, 2), round(discount + 0.01, 2))
                            & (c("l_quantity") < quantity))
            .select((c("l_extendedprice") * c("l_discount")).sum().alias("revenue")))


def q3(lineitem, orders, date=Q3_DATE, seg_mod=5):
    """TPC-H Q3 restated on the two big tables (SURVEY.md 8(d) cfg 4): the customer
    market-segment filter is approximated by o_custkey % seg_mod == 0."""
    c = E.col
    o = orders.filter((c("o_orderdate") < date) & ((c("o_custkey") % seg_mod) == 0))
    li = lineitem.filter(c("l_shipdate") > date)
    return (li.join(o, left_on="l_orderkey", right_on="o_orderkey")
            .group_by("l_orderkey", "o_orderdate", "o_shippriority")
            .agg((c("l_extendedprice") * (1 - c("l_discount"))).sum().alias("revenue")))


def q3_join_frame(lineitem, orders, date=Q3_DATE, seg_mod=5):
    """Q3's two filtered tables joined into a FRAME (no group-by above the join): the materialising hash join, five output columns."""
    c = E.col
    o = orders.filter((c("o_orderdate") < date) & ((c("o_custkey") % seg_mod) == 0))
    li = lineitem.filter(c("l_shipdate") > date)
    return li.join(o, left_on="l_orderkey", right_on="o_orderkey").select("l_orderkey", "o_orderdate", "o_shippriority", "l_extendedprice", "l_discount")


def q3_semi_frame(lineitem, orders, date=Q3_DATE, seg_mod=5):
    """The lineitem rows of Q3's join as a SEMI join: lineitem[l_shipdate > date] whose order is among orders[o_orderdate < date, o_custkey % seg_mod == 0] --
    left columns only, left order (single_keys_semi_anti.rs)."""
    c = E.col
    o = orders.filter((c("o_orderdate") < date) & ((c("o_custkey") % seg_mod) == 0))
    li = lineitem.filter(c("l_shipdate") > date)
    return li.join(o, left_on="l_orderkey", right_on="o_orderkey", how="semi")


def q3_partsupp(lineitem, partsupp, date=Q3_DATE, group=5):
    """A join whose BUILD side repeats its keys: lineitem[l_shipdate > date] JOIN partsupp[ps_group == group] ON partkey (dbgen's partsupp holds four rows per part;
    the shape of TPC-H Q9 / Q20's partsupp joins), grouped by (l_partkey, ps_suppkey) -- a group is a build row, every lineitem row of a part contributes to each of the
    part's suppliers.  ps_group is a property of the part (the same for all of its rows): the predicate keeps parts with ALL their partsupp rows."""
    c = E.col
    ps = partsupp.filter(c("ps_group") == group)
    li = lineitem.filter(c("l_shipdate") > date)
    return (li.join(ps, left_on="l_partkey", right_on="ps_partkey")
            .group_by("l_partkey", "ps_suppkey")
            .agg((c("l_extendedprice") * (1 - c("l_discount"))).sum().alias("revenue"), E.len().alias("n")))


def q3_full(customer, orders, lineitem, date=Q3_DATE, segment="BUILDING"):
    """TPC-H Q3 with all three tables, in the form the optimizer hands the physical planner (single-table
    predicates pushed below the joins): customer[c_mktsegment == segment] JOIN orders[o_orderdate < date] ON custkey, then
    JOIN lineitem[l_shipdate > date] ON orderkey, grouped by (o_orderkey, o_orderdate, o_shippriority)."""
    c = E.col
    cust = customer.filter(c("c_mktsegment") == segment)
    o = cust.join(orders.filter(c("o_orderdate") < date), left_on="c_custkey", right_on="o_custkey")
    j = o.join(lineitem.filter(c("l_shipdate") > date), left_on="o_orderkey", right_on="l_orderkey")
    return (j.group_by("o_orderkey", "o_orderdate", "o_shippriority")
            .agg((c("l_extendedprice") * (1 - c("l_discount"))).sum().alias("revenue")))


def q3_full_top10(customer, orders, lineitem, date=Q3_DATE, segment="BUILDING"):
    """... ORDER BY revenue DESC, o_orderdate LIMIT 10"""
    return q3_full(customer, orders, lineitem, date, segment).sort("revenue", "o_orderdate", descending=[True, False]).head(10)


def q1_sorted(lineitem, cutoff=Q1_CUTOFF):
    """TPC-H Q1 including its ORDER BY l_returnflag, l_linestatus (device radix sort of the <= 6 result rows)."""
    return q1(lineitem, cutoff).sort("l_returnflag", "l_linestatus")


def q3_top10(lineitem, orders, date=Q3_DATE, seg_mod=5):
    """TPC-H Q3 including its ORDER BY revenue DESC, o_orderdate LIMIT 10: the Slice directly above the Sort becomes a
    radix select over the ~1e6 groups."""
    return q3(lineitem, orders, date, seg_mod).sort("revenue", "o_orderdate", descending=[True, False]).head(10)
