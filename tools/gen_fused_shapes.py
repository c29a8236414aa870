#!/usr/bin/env python
"""Regenerates polars_amd/csrc/fused_shapes.hpp from the host expression compiler.

The AOT-specialised fused kernels are keyed on the byte-exact program the compiler emits
for a query shape.  Rather than hand-maintaining those programs, this script builds the
benchmark queries over placeholder (schema-only) columns, asks the library to compile
them (plx_describe_fusion -- no GPU needed) and writes the programs out as constexpr
tables.  Workflow: make -C polars_amd/csrc && python tools/gen_fused_shapes.py && make ...
"""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polars_amd as pl  # noqa: E402
from polars_amd import _ffi as F  # noqa: E402
from polars_amd import queries as Q  # noqa: E402


def ph(name, dtype, n=1 << 20, nullable=False, rng=None):
    h = C.c_uint64()
    F.check(F.lib().plx_column_placeholder(dtype.physical, n, int(nullable), 1 if rng else 0, rng[0] if rng else 0, rng[1] if rng else 0, C.byref(h)))
    return pl.Series._from_handle(name, h.value, dtype)


def frames():
    cfg = pl.DataFrame([ph("a", pl.Int64), ph("x", pl.Float64), ph("y", pl.Float64)])
    cfgn = pl.DataFrame([ph("a", pl.Int64), ph("x", pl.Float64, nullable=True), ph("y", pl.Float64)])
    gb = pl.DataFrame([ph("key", pl.Int64), ph("v", pl.Int64)])
    gb5 = pl.DataFrame([ph("k", pl.Categorical([], pl.UInt32), rng=(0, 999_999)), ph("v", pl.Float64)])
    flag = pl.Categorical(["A", "N", "R"], pl.UInt8)
    status = pl.Categorical(["F", "O"], pl.UInt8)
    li = pl.DataFrame([ph("l_shipdate", pl.Datetime), ph("l_returnflag", flag, rng=(0, 2)), ph("l_linestatus", status, rng=(0, 1)),
                       ph("l_quantity", pl.Int64), ph("l_extendedprice", pl.Float64), ph("l_discount", pl.Float64), ph("l_tax", pl.Float64)])
    li3 = pl.DataFrame([ph("l_orderkey", pl.Int64, n=1 << 22), ph("l_extendedprice", pl.Float64, n=1 << 22), ph("l_discount", pl.Float64, n=1 << 22),
                        ph("l_shipdate", pl.Datetime, n=1 << 22)])
    orders = pl.DataFrame([ph("o_orderkey", pl.Int64), ph("o_custkey", pl.Int64), ph("o_orderdate", pl.Datetime), ph("o_shippriority", pl.Int64)])
    from polars_amd import datagen
    cust = pl.DataFrame([ph("c_custkey", pl.Int64, n=1 << 18, rng=(1, 1 << 18)), ph("c_mktsegment", pl.Categorical(datagen.SEGMENTS, pl.UInt8), n=1 << 18, rng=(0, 4))])
    return cfg, cfgn, gb, gb5, li, li3, orders, cust


def shapes():
    """(name, doc, query, program index): join pipelines yield three programs (count, build, probe)."""
    cfg, cfgn, gb, gb5, li, li3, orders, cust = frames()
    q3 = Q.q3(li3.lazy(), orders.lazy())
    q3f = Q.q3_full(cust.lazy(), orders.lazy(), li3.lazy())
    return [(n, d, q, 0) for n, d, q in _single(cfg, cfgn, gb, gb5, li)] + [
        ("SHAPE_Q3_COUNT", "TPC-H Q3 build side: rows passing the orders predicate (sizes the join table)", q3, 0),
        ("SHAPE_Q3_BUILD", "TPC-H Q3 build scan: orders predicate fused, o_orderkey -> row", q3, 1),
        ("SHAPE_Q3_PROBE", "TPC-H Q3 probe scan: l_shipdate predicate + revenue expression fused, aggregates per build row", q3, 2),
        ("SHAPE_Q3F_COUNT", "TPC-H Q3 (three tables) build side count: o_orderdate predicate AND membership of o_custkey in the BUILDING customers' bitmap", q3f, 0),
        ("SHAPE_Q3F_BUILD", "TPC-H Q3 (three tables) build scan: same predicate, o_orderkey -> row", q3f, 1),
        ("SHAPE_Q3F_SEMI", "TPC-H Q3 (three tables) customer scan: c_mktsegment == code -> membership bitmap over c_custkey", q3f, 3),
        ("SHAPE_Q3_PROBE_SCATTER", "TPC-H Q3 probe side, predicate + key only, row id as payload: the scatter of the partitioned probe (unordered probe keys)", q3, 3),
        ("SHAPE_GB2_SUM_CNT_I64", "group_by(k1:i64, k2:i64).agg(v.sum(), v.count())  [config 3 on a two-column key: word-by-word LDS tables]",
         Q.cfg3w(pl.DataFrame([ph("k1", pl.Int64), ph("k2", pl.Int64), ph("v", pl.Int64)]).lazy()), 0),
        ("SHAPE_Q3D_BUILD", "lineitem JOIN partsupp (duplicate build keys) build scan: ps_group predicate fused, ps_partkey -> row chains [bench workload q3d]",
         Q.q3_partsupp(pl.DataFrame([ph("l_partkey", pl.Int64, n=1 << 22), ph("l_extendedprice", pl.Float64, n=1 << 22), ph("l_discount", pl.Float64, n=1 << 22), ph("l_shipdate", pl.Datetime, n=1 << 22)]).lazy(),
                       pl.DataFrame([ph("ps_partkey", pl.Int64), ph("ps_suppkey", pl.Int64), ph("ps_group", pl.Int64)]).lazy()), 1),
    ]


def _single(cfg, cfgn, gb, gb5, li):
    return [
        ("SHAPE_CFG2", "filter(a > k).select((x*(1-y)).sum(), x.mean(), a.sum())  [BASELINE config 2]", Q.cfg2(cfg.lazy())),
        ("SHAPE_CFG2_NULLX", "config 2 with a nullable x", Q.cfg2(cfgn.lazy())),
        ("SHAPE_CFG1", "filter(a > k).select(a.sum())  [BASELINE config 1]", Q.cfg1(cfg.lazy())),
        ("SHAPE_Q1", "TPC-H Q1: 2 dictionary keys packed into a dense group id, 7 aggregate cells", Q.q1(li.lazy())),
        ("SHAPE_GB_SUM_CNT_I64", "group_by(key:i64).agg(v.sum(), v.count())  [BASELINE config 3]", Q.cfg3(gb.lazy())),
        ("SHAPE_GB_SUM_MEAN_U32_F64", "group_by(k:u32 codes).agg(v.sum(), v.mean())  [BASELINE config 5]", Q.cfg5(gb5.lazy())),
    ]


def parse(dump):
    m = re.match(r"inputs=(\d+) pred=(\d+) key=(\d+) ops=\[(.*?)\] aggs=\[(.*?)\] in_dtype=\[(.*?)\]", dump)
    n_in, pred, key = int(m.group(1)), int(m.group(2)), int(m.group(3))
    ops = [tuple(int(v) for v in t.split(",")) for t in re.findall(r"\(([^)]*)\)", m.group(4))]
    aggs = [tuple(int(v) for v in t.split(",")) for t in re.findall(r"\(([^)]*)\)", m.group(5))]
    ind = [t for t in m.group(6).split(",") if t]
    mk = re.search(r" keys=\[(.*?)\]", dump)
    keys = [int(t) for t in mk.group(1).split(",") if t] if mk else []
    return n_in, pred, key, ops, aggs, ind, keys


HEADER = '''// fused_shapes.hpp -- GENERATED by tools/gen_fused_shapes.py; do not edit by hand.
// Programs that are pre-instantiated (AOT) so the interpreter in kernels_fused.hip folds
// to straight-line code.  Each entry is byte-identical to what the host expression
// compiler (engine.cpp, class Compiler) emits for the query in its comment;
// tests/test_plan_compile.py asserts that these queries hit their specialisation.
// Anything else runs the generic (scalar-unit-decoded) interpreter.
#pragma once
#include "../../include/polars_amd.h"
#include "fused.hpp"

#ifndef PLX_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define PLX_HD __host__ __device__
#else
#define PLX_HD
#endif
#endif

namespace plx {
namespace fused {

PLX_HD constexpr Op mkop(uint8_t code, uint8_t dst, uint8_t a = 0, uint8_t b = 0, uint8_t c = 0) {
  return Op{code, dst, a, b, c, {0, 0, 0}};
}

'''


def main():
    out = [HEADER, "enum StaticShapeId {\n"]
    items = shapes()
    for i, (name, doc, _, _) in enumerate(items):
        out.append(f"  {name} = {i},  // {doc}\n")
    out.append("  kNumStaticShapes\n};\n#define PLX_HAVE_Q3_SHAPES 1\n#define PLX_HAVE_Q3FULL_SHAPES 1\n#define PLX_HAVE_Q3_PROBE_SCATTER 1\n#define PLX_HAVE_GB2_SHAPE 1\n#define PLX_HAVE_Q3D_SHAPE 1\n\nPLX_HD constexpr Shape static_shape(int id) {\n  Shape s{};\n  s.pred = kNone;\n  s.key = kNone;\n  switch (id) {\n")
    for name, doc, q, which in items:
        ok, sid, why, dump = q.describe_fusion()
        if not ok:
            raise SystemExit(f"{name}: not fusable: {why}")
        n_in, pred, key, ops, aggs, ind, keys = parse(dump.split("\n")[which])
        out.append(f"    case {name}: {{  // {doc}\n")
        out.append(f"      s.n_inputs = {n_in}; s.n_ops = {len(ops)}; s.n_aggs = {len(aggs)}; s.pred = {pred}; s.key = {key};\n")
        if keys:
            out.append(f"      s.n_keys = {len(keys)};" + "".join(f" s.keys[{j}] = {k};" for j, k in enumerate(keys)) + "\n")
        for j, t in enumerate(ind):
            nullable = t.endswith("?")
            out.append(f"      s.in_dtype[{j}] = {int(t.rstrip('?'))}; s.in_nullable[{j}] = {1 if nullable else 0};\n")
        for j, (code, dst, a, b, c) in enumerate(ops):
            out.append(f"      s.ops[{j}] = mkop({code}, {dst}, {a}, {b}, {c});\n")
        for j, (kind, src) in enumerate(aggs):
            out.append(f"      s.aggs[{j}] = Agg{{{kind}, {src}}};\n")
        out.append("    } break;\n")
    out.append("    default: break;\n  }\n  return s;\n}\n\n}  // namespace fused\n}  // namespace plx\n")
    path = os.path.join(ROOT, "polars_amd", "csrc", "fused_shapes.hpp")
    new = "".join(out)
    old = open(path).read() if os.path.exists(path) else ""
    if new != old:
        open(path, "w").write(new)
        print("updated", path)
    else:
        print("unchanged", path)


if __name__ == "__main__":
    main()
