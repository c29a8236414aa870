"""Multi-GPU execution: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box; "gloo" in the CPU tests).

The reference has no distributed backend; its intra-process exchange is the HashPartitioner
(crates/polars-utils/src/hashing.rs:72-121) writing per-partition index lists that other
threads read (crates/polars-expr/src/hash_keys.rs:263-314), and its partitioned group-by
rewrites every aggregate into a per-partition partial plus a final combine
(crates/polars-stream/src/nodes/group_by.rs:252-497 `combine_locals`).  This module keeps those
two ideas with GPUs as the partitions:

* low-cardinality group-by / whole-frame aggregates (TPC-H Q1): every rank aggregates its row
  shard locally, the G x state partials are all-gathered (a few hundred bytes) and combined --
  no row ever crosses xGMI;
* high-cardinality group-by and joins: rows are routed by key hash with ONE all-to-all per
  operator input (`exchange_by_key`), after which every rank owns a disjoint key set and runs
  the single-GPU operator unchanged; results stay sharded (concatenation of disjoint parts).

Everything here moves FRAMES (device DataFrames of libpolars_amd): the exchange runs inside the
library on RCCL (`LibComm`: plx_exchange_by_key / plx_allgather_frame), the per-rank compute is the
single-GPU operator behind a small ops object (`LibFrameOps`, `LibJoinOps`).  torch.distributed is
only the bootstrap (rank 0's RCCL id, a few host-side agreements).  The CPU tests and
`bench.py --dry-run` inject numpy frames + a gloo communicator with the same methods.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# partial/final decomposition of the aggregates on the hot path: op -> (partial ops, combine ops)
#   sum -> sum | sum ; count/len -> count/len | sum ; min/max -> min/max | min/max ;
#   mean -> (sum as f64, count) | sum, sum then divide   (reduce/mean.rs:82-132 keeps (f64, usize))
PARTIALS = {"sum": [("sum", "sum")], "count": [("count", "sum")], "len": [("len", "sum")], "min": [("min", "min")], "max": [("max", "max")],
            "mean": [("sum_f64", "sum"), ("count", "sum")]}


def world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def single_node_rccl_env():
    """RCCL bootstrap defaults for ONE node (what bench.py and the tests run): the rendezvous sockets go over loopback and the
    InfiniBand transport is not probed -- a GPU box without a usable NIC otherwise spends minutes (or forever) looking for one.
    Data never uses these sockets: ranks of one node talk over xGMI.  Only defaults: an explicit environment wins, and nothing is
    set when MASTER_ADDR names another host."""
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")


def init_process_group(backend: Optional[str] = None):
    """Rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, ws = world()
    if ws == 1 or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        single_node_rccl_env()
    dist.init_process_group(backend=backend, rank=rank, world_size=ws)


def scan_shard(group=None) -> Tuple[int, int]:
    """(rank, world) for scan_parquet / scan_ipc(..., shard=...): the initialised process group's, else the torchrun environment's."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        pass
    rank, _, ws = world()
    return rank, ws


def unify_dictionaries(df, group=None, remap=None):
    """Every rank of a sharded scan decodes ITS row groups, so a string column's codes index a per-rank dictionary.  Before codes
    of different ranks meet (an all-gather of partial aggregates keyed by the column, an exchange by it), all ranks agree on one
    dictionary: the category lists are all-gathered (host objects, a few KB for the TPC-H flags), the union is taken in rank order,
    and the local codes go through a u32 remap table on the device (io.remap_codes).  Returns the frame with the remapped columns;
    a no-op without a process group.  `remap(series, table, union)` is injectable for the CPU tests."""
    from . import datatypes as T
    from . import io
    try:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    except ImportError:
        on = False
    names = [c.name for c in df.get_columns() if isinstance(c.dtype, T.Categorical)]
    if not on or not names:
        return df
    rank, ws = dist.get_rank(group), dist.get_world_size(group)
    mine = {n: list(df[n].dtype.categories) for n in names}
    every: List[Optional[dict]] = [None] * ws
    dist.all_gather_object(every, mine, group=group)
    remap = remap or io.remap_codes
    cols = []
    for c in df.get_columns():
        if c.name in mine:
            union, tables = io.dictionary_union([r[c.name] for r in every])
            c = remap(c, tables[rank], union)
        cols.append(c)
    return type(df)(cols)


class LibComm:
    """RCCL communicator owned by libpolars_amd (include/polars_amd.h plx_comm_*): the exchange runs inside the library --
    hash partition, gather and ONE grouped ncclSend / ncclRecv all-to-all(v) on the library's stream, no torch kernels.
    torch.distributed is only the bootstrap: rank 0's 128-byte RCCL id travels through it once.  world size 1 needs no
    process group at all (a self-exchange: what the single-GPU test exercises)."""

    def __init__(self, pl, group=None):
        import ctypes as C
        F = pl._ffi
        F.ensure_init()
        self.pl, self._F = pl, F
        rank, ws = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                rank, ws = dist.get_rank(group), dist.get_world_size(group)
        except ImportError:
            dist = None
        single_node_rccl_env()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            F.check(F.lib().plx_comm_unique_id(ident))
        if ws > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_uint64()
        F.check(F.lib().plx_comm_init(ident, rank, ws, C.byref(h)))
        self._h, self.rank, self.world_size, self._group = h.value, rank, ws, group
        self.rows_sent = self.bytes_sent = 0          # over the fabric, accumulated

    def info(self):
        """(rank, world size) as RCCL reports them for the live communicator (plx_comm_info: ncclCommUserRank / ncclCommCount)"""
        import ctypes as C
        r, w = C.c_int32(), C.c_int32()
        self._F.check(self._F.lib().plx_comm_info(self._h, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def exchange_by_key(self, df, key: str, seed: int = 0):
        """Every row of `df` goes to rank hash_partition(key); returns the rows this rank now owns (a DataFrame with the same
        columns / logical dtypes)."""
        import ctypes as C
        F = self._F
        out, rows, nbytes = C.c_uint64(), C.c_uint64(), C.c_uint64()
        F.check(F.lib().plx_exchange_by_key(self._h, df._frame_handle(), key.encode(), seed, C.byref(out), C.byref(rows), C.byref(nbytes)))
        self.rows_sent += rows.value; self.bytes_sent += nbytes.value
        return self.pl.DataFrame._from_frame_handle(out.value, df.schema)

    def agree(self, value: float) -> float:
        """Rank 0's value on every rank (plan decisions every rank must take alike); host-side, through the bootstrap process group."""
        if self.world_size == 1:
            return float(value)
        import torch.distributed as dist
        box = [float(value)]
        dist.broadcast_object_list(box, src=0, group=self._group)
        return float(box[0])

    def total(self, value: float) -> float:
        """Sum of one host number per rank, on every rank (sizes that decide a plan every rank must take alike)."""
        if self.world_size == 1:
            return float(value)
        import torch.distributed as dist
        every = [None] * self.world_size
        dist.all_gather_object(every, float(value), group=self._group)
        return float(sum(every))

    def allgather(self, df):
        """Concatenation of every rank's frame (rank order) on every rank."""
        import ctypes as C
        F = self._F
        out = C.c_uint64()
        F.check(F.lib().plx_allgather_frame(self._h, df._frame_handle(), C.byref(out)))
        return self.pl.DataFrame._from_frame_handle(out.value, df.schema)

    def close(self):
        if getattr(self, "_h", 0):
            self._F.lib().plx_comm_free(self._h)
            self._h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupBySpec:
    """group_by(key).agg(...) in the form the sharded operator splits it: aggs = [(out_name, value column or "" for len, op)], op in
    PARTIALS.  The split is the reference's own (crates/polars-stream/src/nodes/group_by.rs:140-250 local pre-aggregation, :252-497
    combine_locals; reduce/mean.rs:82-132 keeps (f64 sum, count)): every rank reduces its rows to one partial row per local group,
    partial rows are routed by key hash, the owner of a key combines them."""

    def __init__(self, key: str, aggs: Sequence[Tuple[str, str, str]]):
        self.key, self.aggs = key, [(o, c or "", op) for o, c, op in aggs]
        for _, _, op in self.aggs:
            if op not in PARTIALS:
                raise ValueError(f"aggregate {op!r} has no partial / combine decomposition on this path")

    def partial_aggs(self) -> List[Tuple[str, str, str]]:
        return [(f"{o}__p{i}", c, pop) for o, c, op in self.aggs for i, (pop, _) in enumerate(PARTIALS[op])]

    def merge_aggs(self) -> List[Tuple[str, str, str]]:
        return [(f"{o}__p{i}", f"{o}__p{i}", cop) for o, c, op in self.aggs for i, (_, cop) in enumerate(PARTIALS[op])]


class LibFrameOps:
    """The local queries of a sharded group-by as libpolars_amd plans (each one is the single-GPU operator: the fused scan into LDS /
    partitioned tables); tests/ and bench.py --dry-run inject a numpy double with the same four methods."""

    def __init__(self, pl):
        self.pl = pl

    def _expr(self, col: str, op: str):
        pl = self.pl
        if op == "len":
            return pl.len()
        if op == "sum_f64":
            return pl.col(col).cast(pl.Float64).sum()
        return getattr(pl.col(col), op)()

    def final(self, df, spec: GroupBySpec):
        return df.lazy().group_by(spec.key).agg(*[self._expr(c, op).alias(o) for o, c, op in spec.aggs]).collect()

    def partial(self, df, spec: GroupBySpec):
        return df.lazy().group_by(spec.key).agg(*[self._expr(c, op).alias(o) for o, c, op in spec.partial_aggs()]).collect()

    def merge(self, part, spec: GroupBySpec, source_schema=None):
        """partial rows of the keys this rank owns -> the final rows.  mean = sum of the f64 partial sums / sum of the counts, null when
        no value was counted (count % count is null exactly then: integer mod by zero, arithmetic/signed.rs:35-70)."""
        pl = self.pl
        lf = part.lazy().group_by(spec.key).agg(*[self._expr(c, op).alias(o) for o, c, op in spec.merge_aggs()])
        outs = [pl.col(spec.key)]
        for o, c, op in spec.aggs:
            if op == "mean":
                n = pl.col(f"{o}__p1")
                m = pl.col(f"{o}__p0") / (n + n % n).cast(pl.Float64)
                if source_schema is not None and source_schema.get(c) == pl.Float32:      # Float32.mean() stays Float32 (reduce/mean.rs:29-80)
                    m = m.cast(pl.Float32)
                outs.append(m.alias(o))
            else:
                outs.append(pl.col(f"{o}__p0").alias(o))
        return lf.select(*outs).collect()

    def distinct_in_prefix(self, df, key: str, n: int) -> int:
        return self.pl.DataFrame([df[key]]).slice(0, n).lazy().group_by(key).agg(self.pl.len().alias("n")).collect().height


def estimate_shrink(rows: int, sample_rows: int, sample_distinct: int) -> float:
    """How much a local group-by shrinks `rows` rows, from the number of distinct keys in a sample: with G equally likely keys a sample of
    n rows holds u = G (1 - exp(-n / G)) distinct ones; G is solved by bisection and pushed through the same formula for the whole shard
    (skew only makes the true shrink larger).  Returns rows / expected local groups; 1.0 when every sampled key was distinct."""
    import math
    n, u = float(sample_rows), float(sample_distinct)
    if rows <= 0 or n <= 0 or u <= 0:
        return 1.0
    if u >= 0.995 * n:                      # (nearly) all distinct: G is not identifiable from this sample, assume no shrink
        return 1.0
    lo, hi = u, 1e15
    for _ in range(200):
        g = math.sqrt(lo * hi)
        if g * -math.expm1(-n / g) < u:
            lo = g
        else:
            hi = g
    g = math.sqrt(lo * hi)
    return rows / max(1.0, g * -math.expm1(-rows / g))


PREAGG_MIN_SHRINK = 8.0          # pre-aggregate before the exchange when the local group-by shrinks the shard at least this much
PREAGG_SAMPLE_ROWS = 1 << 20


def sharded_groupby(comm, df, spec, ops=None, *, mode: str = "auto", always_exchange: bool = False, info: Optional[dict] = None):
    """group_by(key).agg(...) over row shards; the result stays sharded by key (the concatenation over the ranks is
    the global result; null keys live on rank 0: hashing.rs:111-115).

    mode "preagg": local group-by -> ONE exchange of the partial rows by key hash (G x state bytes cross xGMI, not the rows) ->
                   the owner combines the partials (group_by.rs:140-497).  BASELINE configs 3 / 5 (1e9 rows, 1e6 keys per rank): ~20 MB
                   per rank instead of ~14 GB.
    mode "rows"  : ONE exchange of the raw rows by key hash -> the single-GPU operator over the disjoint key set; right when almost
                   every row is its own group (the local aggregate would not shrink anything).
    mode "auto"  : "preagg" when the distinct keys of a 2^20-row sample predict a local shrink >= 8x (estimate_shrink); the choice is
                   agreed across ranks (rank 0's estimate is broadcast through the communicator's agree()).
    `spec`: a GroupBySpec.
    `comm`: LibComm (RCCL inside the library) or any object with world_size, exchange_by_key(df, key) and agree(value);
    `ops`: LibFrameOps or a double.  `info`, if given, receives {"mode", "shrink_estimate", "partial_rows"}."""
    alone = comm is None or (comm.world_size == 1 and not always_exchange)
    if alone:
        if info is not None:
            info.update(mode="local", shrink_estimate=None, partial_rows=None)
        return ops.final(df, spec)
    if mode == "auto":
        n = min(int(df.height), PREAGG_SAMPLE_ROWS)
        shrink = estimate_shrink(int(df.height), n, ops.distinct_in_prefix(df, spec.key, n)) if n else 1.0
        shrink = comm.agree(shrink)                    # every rank must take the same branch (the exchanges must pair up)
        mode = "preagg" if shrink >= PREAGG_MIN_SHRINK else "rows"
    else:
        shrink = None
    if mode == "rows":
        out = ops.final(comm.exchange_by_key(df, spec.key), spec)
        if info is not None:
            info.update(mode="rows", shrink_estimate=shrink, partial_rows=None)
        return out
    if mode != "preagg":
        raise ValueError(f"sharded_groupby mode {mode!r}")
    part = ops.partial(df, spec)
    owned = comm.exchange_by_key(part, spec.key)
    if info is not None:
        info.update(mode="preagg", shrink_estimate=shrink, partial_rows=int(part.height))
    return ops.merge(owned, spec, getattr(df, "schema", None))


class JoinGroupBySpec:
    """`probe JOIN build ON probe_key == build_key -> GROUP BY (result_key, attributes) -> aggregates` (TPC-H Q3's shape) in the form the
    sharded operator needs: `merge` = [(aggregate column of the local result, "sum" | "min" | "max")], every other column of the local
    result is a group attribute (result_key first; the attributes are functionally determined by it)."""

    def __init__(self, probe_key: str, build_key: str, result_key: str, merge: Sequence[Tuple[str, str]]):
        self.probe_key, self.build_key, self.result_key, self.merge = probe_key, build_key, result_key, list(merge)
        for _, op in self.merge:
            if op not in ("sum", "min", "max"):
                raise ValueError(f"partial aggregate {op!r} cannot be merged across ranks")


class LibJoinOps:
    """The per-rank queries of a sharded join -> group-by as libpolars_amd plans over device frames.  `local(probe, build)` is the
    single-GPU fused filter -> join -> group-by (a LazyFrame builder such as queries.q3); `build_filter` / `probe_filter` are the
    single-input predicates pushed below the exchange (the reference pushes them below the join the same way,
    polars-plan predicate_pushdown), so only surviving rows cross xGMI.  bench.py --dry-run and the gloo tests inject a numpy double
    with the same methods."""

    def __init__(self, pl, local, build_filter=None, probe_filter=None):
        self.pl, self._local, self._bf, self._pf = pl, local, build_filter, probe_filter

    def build_prefilter(self, df):
        return df if self._bf is None else df.lazy().filter(self._bf).collect()

    def probe_prefilter(self, df):
        return df if self._pf is None else df.lazy().filter(self._pf).collect()

    def local(self, probe, build):
        return self._local(probe.lazy(), build.lazy()).collect()

    def merge(self, part, spec: JoinGroupBySpec):
        """partial groups of the keys this rank owns -> one row per key (group_by(key, attributes).agg(merge ops))"""
        pl = self.pl
        ops = dict(spec.merge)
        keys = [c for c in part.columns if c not in ops]
        return part.lazy().group_by(*keys).agg(*[getattr(pl.col(c), op)().alias(c) for c, op in spec.merge]).collect()

    def nbytes(self, df) -> int:
        return int(sum(df.height * (t.np_dtype.itemsize if getattr(t, "np_dtype", None) is not None else 1) for t in df.schema.values()))


BROADCAST_BUILD_BYTES = 2 << 30       # "auto": all-gather the filtered build side when it is at most this large over ALL ranks


def sharded_join_groupby(comm, ops, probe, build, spec: JoinGroupBySpec, *, mode: str = "auto", always_exchange: bool = False, info: Optional[dict] = None):
    """Sharded `probe JOIN build -> GROUP BY -> aggregates` over row shards of both inputs (BASELINE config 4: SF100 lineitem JOIN orders
    over 8 GPUs); the partitioned build / probe of crates/polars-stream/src/nodes/joins/equi_join.rs:446-760 with GPUs as the
    partitions (HashPartitioner, null keys -> partition 0: crates/polars-utils/src/hashing.rs:72-121).  Frames in, frame out; every
    exchange is the library's (comm.exchange_by_key / comm.allgather: one grouped RCCL all-to-all(v) / all-gather(v) per call).

    mode "shuffle"  : both sides are filtered, then routed by key hash (one exchange per input: a grace hash join); every key is then
                      owned by one rank and the local result is final.
    mode "broadcast": the filtered build side is all-gathered (the small relation: filtered TPC-H orders at SF100 is ~0.5 GB in all),
                      the probe side never moves; rows of one key may sit on several ranks, so the partial groups are routed by
                      result key (one small exchange) and merged by their owner.
    mode "auto"     : broadcast when the filtered build side of ALL ranks is at most BROADCAST_BUILD_BYTES (agreed across ranks).
    The result stays sharded by key.  `info` receives {"mode", "build_rows", "probe_rows", "partial_rows"}."""
    alone = comm is None or (comm.world_size == 1 and not always_exchange)
    if alone:
        if info is not None:
            info.update(mode="local", build_rows=None, probe_rows=None, partial_rows=None)
        return ops.local(probe, build)
    build_f = ops.build_prefilter(build)
    if mode == "auto":
        mode = "broadcast" if comm.total(ops.nbytes(build_f)) <= BROADCAST_BUILD_BYTES else "shuffle"
    if mode == "shuffle":
        probe_f = ops.probe_prefilter(probe)
        p2 = comm.exchange_by_key(probe_f, spec.probe_key)
        b2 = comm.exchange_by_key(build_f, spec.build_key)
        if info is not None:
            info.update(mode="shuffle", build_rows=int(build_f.height), probe_rows=int(probe_f.height), partial_rows=None)
        return ops.local(p2, b2)
    if mode != "broadcast":
        raise ValueError(f"sharded_join_groupby mode {mode!r}")
    part = ops.local(probe, comm.allgather(build_f))
    owned = comm.exchange_by_key(part, spec.result_key)
    if info is not None:
        info.update(mode="broadcast", build_rows=int(build_f.height), probe_rows=None, partial_rows=int(part.height))
    return ops.merge(owned, spec)


def q3_ops(pl):
    """(LibJoinOps, JoinGroupBySpec) of TPC-H Q3 on the two big tables (queries.q3): orders predicate and lineitem predicate pushed below
    the exchange, the fused filter -> join -> group-by as the local operator."""
    from . import queries
    c = pl.col
    ops = LibJoinOps(pl, queries.q3, build_filter=(c("o_orderdate") < queries.Q3_DATE) & ((c("o_custkey") % 5) == 0), probe_filter=c("l_shipdate") > queries.Q3_DATE)
    return ops, JoinGroupBySpec("l_orderkey", "o_orderkey", "l_orderkey", [("revenue", "sum")])
