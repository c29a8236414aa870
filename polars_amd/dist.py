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

Everything here moves `torch.Tensor`s; the per-rank compute is delegated to a `LocalOps`
object.  The product `HipLocalOps` calls libpolars_amd through the C ABI on device memory; the
CPU tests inject an oracle-backed LocalOps to exercise the exchange logic under gloo.
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


def torch_sync():
    """Work queued on torch's current stream must be visible to the library's stream before it reads the tensors."""
    import torch
    torch.cuda.current_stream().synchronize()


class LocalOps:
    """Per-rank compute used by the exchange layer (tensors in, tensors out)."""

    def hash_partition(self, key, n_parts: int, seed: int = 0):
        """-> (perm int64 tensor grouping rows by partition, counts list[int])"""
        raise NotImplementedError

    def take(self, col, perm):
        return col[perm]

    def groupby_partial(self, keys: Dict[str, object], values: Dict[str, object], aggs: Sequence[Tuple[str, str, str]]):
        """aggs = [(out_name, value column, partial op)] -> dict of tensors, one row per local group (keys + outs)."""
        raise NotImplementedError

    def mean_from_partials(self, total, count):
        """f64 sum / count of the mean decomposition (reduce/mean.rs:82-132)."""
        import torch
        return total.to(torch.float64) / count.to(torch.float64)


class HipLocalOps(LocalOps):
    """LocalOps on libpolars_amd (device tensors are wrapped zero-copy, results copied D2D)."""

    def __init__(self, pl):
        self.pl = pl

    def _series(self, name, t):
        return self.pl.Series.from_torch(name, t)

    def hash_partition(self, key, n_parts: int, seed: int = 0):
        import ctypes as C

        import torch
        F = self.pl._ffi
        torch.cuda.current_stream().synchronize()
        s = self._series("k", key)
        h = C.c_uint64()
        counts = (C.c_int64 * n_parts)()
        F.check(F.lib().plx_hash_partition(s._h, n_parts, seed, C.byref(h), counts))
        perm = self.pl.Series._from_handle("perm", h.value, self.pl.UInt32)
        return perm.cast(self.pl.Int64).to_torch(), list(counts)     # widened by the library's cast kernel (index tensors are int64 in torch)

    def mean_from_partials(self, total, count):
        pl = self.pl
        torch_sync()
        return (self._series("s", total).cast(pl.Float64) / self._series("c", count).cast(pl.Float64)).to_torch()

    def groupby_partial(self, keys, values, aggs):
        import torch
        pl = self.pl
        torch.cuda.current_stream().synchronize()
        cols = [self._series(n, t) for n, t in keys.items()] + [self._series(n, t) for n, t in values.items()]
        exprs = []
        for out, col, op in aggs:
            e = pl.col(col) if col else None
            if op == "sum_f64":
                e = e.cast(pl.Float64).sum()
            elif op == "len":
                e = pl.len()
            else:
                e = getattr(e, op)()
            exprs.append(e.alias(out))
        df = pl.DataFrame(cols).lazy().group_by(*keys.keys()).agg(*exprs).collect()
        return {c: df[c].to_torch() for c in df.columns}


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


P2P_CHUNK_BYTES = 1 << 29        # largest per-peer segment of one all-to-all round on the torch path (exchange_by_key); the library path: comm.cpp kP2PChunk
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


def allgather_concat(t, group=None):
    """Variable-length all-gather of a 1-D tensor (sizes first, then padded payload)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    ws = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes + [1])
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    outs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)])


def exchange_by_key(ops: LocalOps, key, cols: Dict[str, object], seed: int = 0, group=None) -> Dict[str, object]:
    """Route every row to rank hash_partition(key) with one all-to-all per column.
    Nulls (none on this path yet) would go to partition 0 like the reference's null_partition()."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(cols)
    ws = dist.get_world_size(group)
    perm, counts = ops.hash_partition(key, ws, seed)
    send = torch.tensor(counts, dtype=torch.int64, device=key.device)
    recv = torch.zeros_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_counts = [int(x) for x in recv.tolist()]
    # RCCL 2.26 delivers only the first half of a point-to-point transfer above 2^30 bytes (measured on MI355X, see comm.cpp kP2PChunk): a column whose
    # per-peer segment is larger than P2P_CHUNK_BYTES goes in several rounds of row slabs; the round count is agreed across ranks
    biggest = torch.tensor([max(counts + recv_counts + [0])], dtype=torch.int64, device=key.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    biggest = int(biggest.item())
    send_off = [0] + list(np.cumsum(counts)); recv_off = [0] + list(np.cumsum(recv_counts))
    out = {}
    for name, t in cols.items():
        src = ops.take(t, perm).contiguous()
        dst = torch.empty(sum(recv_counts), dtype=t.dtype, device=t.device)
        lim = max(1, P2P_CHUNK_BYTES // max(1, t.element_size()))
        rounds = max(1, -(-biggest // lim))
        if rounds == 1:
            dist.all_to_all_single(dst, src, output_split_sizes=recv_counts, input_split_sizes=counts, group=group)
        else:
            for r in range(rounds):
                ins = [min(max(c - r * lim, 0), lim) for c in counts]
                outs = [min(max(c - r * lim, 0), lim) for c in recv_counts]
                src_r = torch.cat([src[int(send_off[p]) + r * lim: int(send_off[p]) + r * lim + ins[p]] for p in range(ws)]) if sum(ins) else src[:0]
                dst_r = torch.empty(sum(outs), dtype=t.dtype, device=t.device)
                dist.all_to_all_single(dst_r, src_r.contiguous(), output_split_sizes=outs, input_split_sizes=ins, group=group)
                at = 0
                for p in range(ws):
                    if outs[p]:
                        dst[int(recv_off[p]) + r * lim: int(recv_off[p]) + r * lim + outs[p]] = dst_r[at: at + outs[p]]
                    at += outs[p]
        out[name] = dst
    return out


def _combine(op: str, t, inverse, n_groups: int):
    import torch
    if t.dtype in (torch.int32, torch.int16, torch.int8, torch.uint8):
        t = t.to(torch.int64)   # counts / narrow partials combine in 64 bit
    if op == "sum":
        return torch.zeros(n_groups, dtype=t.dtype, device=t.device).index_add_(0, inverse, t)
    red = "amin" if op == "min" else "amax"
    init = torch.full((n_groups,), float("inf") if op == "min" else float("-inf"), dtype=torch.float64, device=t.device) if t.dtype.is_floating_point else \
        torch.full((n_groups,), torch.iinfo(t.dtype).max if op == "min" else torch.iinfo(t.dtype).min, dtype=t.dtype, device=t.device)
    return init.to(t.dtype).scatter_reduce_(0, inverse, t, reduce=red)


def groupby_agg(ops: LocalOps, keys: Dict[str, object], values: Dict[str, object], aggs: Sequence[Tuple[str, str, str]], *, mode: str = "auto",
                group=None) -> Dict[str, object]:
    """Sharded group_by(keys).agg(...): aggs = [(out_name, value column, op)], op in PARTIALS.

    mode "gather" : local aggregate -> all-gather partials -> every rank combines (replicated, tiny result)
    mode "shuffle": all-to-all rows by key hash -> local aggregate -> result stays sharded by key
    mode "auto"   : "gather" unless the local aggregate shrinks the data by less than 8x.
    Result columns: keys + out_names.  mean is null-free here (count == 0 groups cannot exist without nulls).
    """
    import torch
    import torch.distributed as dist
    distributed = dist.is_initialized() and dist.get_world_size(group) > 1
    partial_aggs: List[Tuple[str, str, str]] = []
    for out, col, op in aggs:
        for i, (pop, _) in enumerate(PARTIALS[op]):
            partial_aggs.append((f"{out}__p{i}", col, pop))
    if mode == "shuffle" and distributed:
        if len(keys) != 1:
            raise NotImplementedError("shuffle mode routes on a single key column")
        kname = next(iter(keys))
        moved = exchange_by_key(ops, keys[kname], {**keys, **values}, group=group)
        keys = {k: moved[k] for k in keys}
        values = {k: moved[k] for k in values}
        distributed = False   # key sets are disjoint now: the local result is final
    part = ops.groupby_partial(keys, values, partial_aggs)
    if distributed:
        part = {k: allgather_concat(v, group) for k, v in part.items()}
        # combine partials of equal keys: pack the key tuple into rows and unique them
        kt = torch.stack([part[k].to(torch.int64) for k in keys], dim=1)
        uniq, inverse = torch.unique(kt, dim=0, return_inverse=True)
        n = uniq.shape[0]
        comb = {k: uniq[:, i].to(part[k].dtype) for i, k in enumerate(keys)}
        for out, col, op in aggs:
            for i, (_, cop) in enumerate(PARTIALS[op]):
                comb[f"{out}__p{i}"] = _combine(cop, part[f"{out}__p{i}"], inverse, n)
        part = comb
    res = {k: part[k] for k in keys}
    for out, col, op in aggs:
        if op == "mean":
            res[out] = ops.mean_from_partials(part[f"{out}__p0"], part[f"{out}__p1"])
        else:
            res[out] = part[f"{out}__p0"]
    return res


def allgather_columns(cols: Dict[str, object], group=None) -> Dict[str, object]:
    """Replicate a (small) frame on every rank: one variable-length all-gather per column."""
    return {k: allgather_concat(v, group) for k, v in cols.items()}


def join_groupby(ops: LocalOps, probe: Dict[str, object], build: Dict[str, object], probe_key: str, build_key: str, local_fn,
                 merge: Sequence[Tuple[str, str]], result_key: str, *, mode: str = "auto", build_bytes_limit: int = 2 << 30,
                 build_prefilter=None, probe_prefilter=None, group=None) -> Dict[str, object]:
    """Sharded `probe JOIN build ON key -> GROUP BY (key, build columns) -> aggregates` (TPC-H Q3 shape).

    Every rank holds a row shard of both inputs.  `local_fn(probe_cols, build_cols) -> {column: tensor}` runs the
    single-GPU fused pipeline on what the rank holds after the exchange and returns per-group partial rows; `merge`
    lists (column, "sum" | "min" | "max") for the partial aggregates, the remaining columns are group attributes
    (functionally determined by `result_key`).

    mode "broadcast": the build side is all-gathered (it is the small relation: filtered TPC-H orders at SF100 is
                      ~0.35 GB), the probe side never moves; rows of one key may sit on several ranks, so the partial
                      groups are merged by key with one small all-to-all.
    mode "shuffle"  : both sides are routed by key hash with one all-to-all per input (grace hash join); every key is
                      then owned by one rank and the local result is final.
    mode "auto"     : broadcast when the global build side is below `build_bytes_limit`.
    `build_prefilter` / `probe_prefilter` (cols -> cols) are the single-input predicates pushed below the exchange, so
    only surviving rows cross xGMI (the reference pushes them below the join the same way, predicate_pushdown/mod.rs).
    The result stays sharded by key (disjoint key sets across ranks)."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_initialized() and dist.get_world_size(group) > 1
    if not distributed:
        return local_fn(probe, build)
    if build_prefilter is not None:
        build = build_prefilter(build)
    if mode == "auto":
        local_bytes = sum(int(t.numel()) * t.element_size() for t in build.values())
        tot = torch.tensor([local_bytes], dtype=torch.int64, device=next(iter(build.values())).device)
        dist.all_reduce(tot, group=group)
        mode = "broadcast" if int(tot.item()) <= build_bytes_limit else "shuffle"
    if mode == "shuffle":
        if probe_prefilter is not None:
            probe = probe_prefilter(probe)
        probe2 = exchange_by_key(ops, probe[probe_key], probe, group=group)
        build2 = exchange_by_key(ops, build[build_key], build, group=group)
        return local_fn(probe2, build2)
    # broadcast
    part = local_fn(probe, allgather_columns(build, group))
    moved = exchange_by_key(ops, part[result_key], part, group=group)
    # merge partial groups of equal key with the local group-by operator; the attribute columns are functionally
    # determined by the key, so grouping by (key, attributes) yields one row per key
    merge_ops = dict(merge)
    keys = {n: t for n, t in moved.items() if n not in merge_ops}
    vals = {n: t for n, t in moved.items() if n in merge_ops}
    if result_key not in keys:
        raise ValueError("result_key must not be one of the merged aggregates")
    return ops.groupby_partial(keys, vals, [(n, n, op) for n, op in merge])


class Q3Local:
    """The per-rank pieces of sharded TPC-H Q3 on libpolars_amd (device tensors in/out): the orders predicate pushed
    below the exchange and the fused filter -> join -> group-by pipeline (polars_amd/queries.py q3)."""

    def __init__(self, pl):
        from . import datagen, queries
        self.pl, self.datagen, self.queries = pl, datagen, queries
        self.ops = HipLocalOps(pl)

    @staticmethod
    def _cols(df):
        return {n: df[n].to_torch() for n in df.columns}

    def build_prefilter(self, bc):
        pl, c = self.pl, self.pl.col
        f = self.datagen.frame_from_torch(pl, bc, self.datagen.ORDERS_Q3_COLS)
        return self._cols(f.lazy().filter((c("o_orderdate") < self.queries.Q3_DATE) & ((c("o_custkey") % 5) == 0)).collect())

    def probe_prefilter(self, pc):
        pl, c = self.pl, self.pl.col
        f = self.datagen.frame_from_torch(pl, pc, self.datagen.LINEITEM_Q3_COLS)
        return self._cols(f.lazy().filter(c("l_shipdate") > self.queries.Q3_DATE).collect())

    def local(self, pc, bc):
        L = self.datagen.frame_from_torch(self.pl, pc, self.datagen.LINEITEM_Q3_COLS)
        O = self.datagen.frame_from_torch(self.pl, bc, self.datagen.ORDERS_Q3_COLS)
        return self._cols(self.queries.q3(L.lazy(), O.lazy()).collect())

    def run(self, lineitem, orders, mode: str = "broadcast", group=None):
        return join_groupby(self.ops, lineitem, orders, "l_orderkey", "o_orderkey", self.local, [("revenue", "sum")], "l_orderkey", mode=mode,
                            build_prefilter=self.build_prefilter, probe_prefilter=self.probe_prefilter, group=group)
