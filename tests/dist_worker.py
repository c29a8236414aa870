"""Worker of tests/test_dist_gloo_cpu.py: one process per rank, gloo backend, CPU tensors.
The per-rank compute is an oracle-backed LocalOps (test infrastructure) -- what is under test
is polars_amd.dist: key-hash routing with one all-to-all, partial/final aggregate
decomposition, variable-length all-gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402
from polars_amd import dist as pdist  # noqa: E402

AGG = {"sum": orc.AGG_SUM, "count": orc.AGG_COUNT, "len": orc.AGG_LEN, "min": orc.AGG_MIN, "max": orc.AGG_MAX}


class OracleLocalOps(pdist.LocalOps):
    def hash_partition(self, key, n_parts, seed=0):
        p = orc.hash_partition(key.numpy(), None, n_parts, seed)
        perm = np.argsort(p, kind="stable")
        return torch.from_numpy(perm.astype(np.int64)), np.bincount(p, minlength=n_parts).tolist()

    def groupby_partial(self, keys, values, aggs):
        ks = [k.numpy() for k in keys.values()]
        spec = []
        for out, col, op in aggs:
            v = values[col].numpy() if col else None
            if op == "sum_f64":
                spec.append((out, orc.AGG_SUM, v.astype(np.float64), None))
            else:
                spec.append((out, AGG[op], v, None))
        r = orc.q_groupby(ks, [None] * len(ks), spec)
        res = {name: torch.from_numpy(np.ascontiguousarray(r[f"key_{i}"][0])) for i, name in enumerate(keys)}
        for out, _, _ in aggs:
            a = r[out][0]
            res[out] = torch.from_numpy(np.ascontiguousarray(a.astype(np.int64) if a.dtype == np.uint32 else a))
        return res


def main():
    pdist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    ops = OracleLocalOps()
    out_dir = sys.argv[1]
    # every rank owns a different row shard of the same logical table
    rng = np.random.default_rng(1000 + rank)
    n = 20_000 + 1000 * rank
    key = torch.from_numpy(rng.integers(0, 3000, n).astype(np.int64))
    flag = torch.from_numpy(rng.integers(0, 3, n).astype(np.int64))
    v = torch.from_numpy(rng.integers(-50, 50, n).astype(np.int64))
    x = torch.from_numpy(rng.uniform(0, 1, n))
    aggs = [("s", "v", "sum"), ("m", "x", "mean"), ("mn", "v", "min"), ("mx", "x", "max"), ("n", "", "len")]
    # (a) exchange_by_key: every row lands on the rank its key hashes to, nothing lost
    moved = pdist.exchange_by_key(ops, key, {"key": key, "v": v})
    owner = orc.hash_partition(moved["key"].numpy(), None, ws, 0)
    assert (owner == rank).all(), "row routed to the wrong rank"
    tot = torch.tensor([moved["key"].numel(), int(moved["v"].sum())], dtype=torch.int64)
    mine = torch.tensor([n, int(v.sum())], dtype=torch.int64)
    dist.all_reduce(tot); dist.all_reduce(mine)
    assert torch.equal(tot, mine), "rows or values lost in the all-to-all"
    # (a') the same exchange in several rounds of row slabs (what a per-peer segment above 2^29 bytes triggers on RCCL): identical result
    saved = pdist.P2P_CHUNK_BYTES
    pdist.P2P_CHUNK_BYTES = 8 * 1000                              # 1000 rows of an int64 column per peer and round
    moved2 = pdist.exchange_by_key(ops, key, {"key": key, "v": v})
    pdist.P2P_CHUNK_BYTES = saved
    assert torch.equal(moved2["key"], moved["key"]) and torch.equal(moved2["v"], moved["v"]), "chunked exchange differs from the single-round exchange"
    # (b) low-cardinality group-by: local partials + all-gather + combine (replicated result)
    g = pdist.groupby_agg(ops, {"flag": flag}, {"v": v, "x": x}, aggs, mode="gather")
    # (c) high-cardinality group-by: shuffle by key hash, result sharded by key
    s = pdist.groupby_agg(ops, {"key": key}, {"v": v, "x": x}, aggs, mode="shuffle")
    # (d) bench.py --gpus N: per-rank Q1 frames are all-gathered as one fixed-size tensor and merged on every rank
    import bench
    from polars_amd import datagen
    li = datagen.lineitem_host(30_000 + 500 * rank, seed=200 + rank)
    qcols = {c: li[c] for c in datagen.LINEITEM_Q1_COLS}
    mine = {c: a.tolist() for c, a in orc.q1_native(qcols, datagen.us(1998, 9, 2), streaming=True).items()}
    merged = bench.combine_q1_results(bench.allgather_q1(mine, ws))
    np.savez(os.path.join(out_dir, f"q1_rank{rank}.npz"), **{c: np.asarray(a) for c, a in qcols.items()}, **{"m_" + c: np.asarray(a) for c, a in merged.items()})
    # (e) sharded join -> group-by (Q3 shape), broadcast and shuffle modes; the local pipeline is the oracle's q3
    orders, li = datagen.orders_lineitem_host(30000 + 300 * rank, seed=300 + rank)
    # make order keys globally unique across ranks (each rank generated its own key space)
    off = rank * 10_000_000
    orders["o_orderkey"] = orders["o_orderkey"] + off; li["l_orderkey"] = li["l_orderkey"] + off
    # scatter this rank's lineitem rows so that keys of one order also live on OTHER ranks' probe shards
    allkeys = [torch.from_numpy(li[c]) for c in datagen.LINEITEM_Q3_COLS]
    probe = {c: pdist.allgather_concat(t)[rank::ws].contiguous() for c, t in zip(datagen.LINEITEM_Q3_COLS, allkeys)}
    build = {c: torch.from_numpy(orders[c]) for c in datagen.ORDERS_Q3_COLS}
    date = datagen.us(1995, 3, 15)

    def local_q3(pc, bc):
        r = orc.q3({c: t.numpy() for c, t in pc.items()}, {c: t.numpy() for c, t in bc.items()}, date)
        return {c: torch.from_numpy(np.ascontiguousarray(a)) for c, a in r.items()}
    def build_pre(bc):
        m = torch.from_numpy((bc["o_orderdate"].numpy() < date) & (bc["o_custkey"].numpy() % 5 == 0))
        return {c: t[m] for c, t in bc.items()}

    def probe_pre(pc):
        m = pc["l_shipdate"] > date
        return {c: t[m] for c, t in pc.items()}
    for mode in ("broadcast", "shuffle", "auto"):
        r = pdist.join_groupby(ops, probe, build, "l_orderkey", "o_orderkey", local_q3, [("revenue", "sum")], "l_orderkey", mode=mode,
                               build_prefilter=build_pre if mode != "broadcast" else None, probe_prefilter=probe_pre if mode == "shuffle" else None)
        np.savez(os.path.join(out_dir, f"q3_{mode}_rank{rank}.npz"), **{c: t.numpy() for c, t in r.items()})
    np.savez(os.path.join(out_dir, f"q3_in_rank{rank}.npz"), **{"p_" + c: t.numpy() for c, t in probe.items()}, **{"b_" + c: t.numpy() for c, t in build.items()})
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), key=key.numpy(), flag=flag.numpy(), v=v.numpy(), x=x.numpy(),
             **{f"g_{k}": t.numpy() for k, t in g.items()}, **{f"s_{k}": t.numpy() for k, t in s.items()})
    # (g) the frame-level sharded group-by (dist.sharded_groupby: what bench.py --gpus N --workload cfg3 runs): pre-aggregation before the
    # exchange vs raw-row exchange vs the sample-driven choice, with NULL keys (one group, owned by rank 0) and null values; numpy doubles
    # stand in for the library frames and the RCCL communicator (bench.DryFrame / DryComm / DryOps)
    r3 = np.random.default_rng(4000 + rank)
    m = 30_000 + 700 * rank
    gk = r3.integers(0, 2500, m).astype(np.int64); gk_valid = r3.random(m) > 0.02
    gv = r3.integers(-1000, 1000, m).astype(np.int64); gv_valid = r3.random(m) > 0.1
    gv_valid[gk == 7] = False                                     # a group whose values are ALL null: sum 0, count 0, mean / min null
    gx = r3.uniform(-1, 1, m)
    shard = bench.DryFrame({"key": gk, "v": gv, "x": gx}, {"key": gk_valid, "v": gv_valid})
    spec = pdist.GroupBySpec("key", [("v_sum", "v", "sum"), ("v_count", "v", "count"), ("v_mean", "v", "mean"), ("v_min", "v", "min"), ("x_max", "x", "max"), ("n", "", "len")])
    comm, fops = bench.DryComm(), bench.DryOps()
    save = {"in_key": gk, "in_key_valid": gk_valid, "in_v": gv, "in_v_valid": gv_valid, "in_x": gx}
    for mode in ("preagg", "rows", "auto"):
        comm.rows_sent = comm.bytes_sent = 0
        info = {}
        out = pdist.sharded_groupby(comm, shard, spec, fops, mode=mode, info=info)
        for c, a in out.cols.items():
            save[f"{mode}_{c}"] = a
            save[f"{mode}_{c}__valid"] = out.validity(c)
        save[f"{mode}_rows_sent"] = np.array([comm.rows_sent]); save[f"{mode}_mode"] = np.array([info["mode"]])
    # a high-cardinality shard (every key distinct): the sample says the local aggregate shrinks nothing -> "auto" must exchange rows
    uniq = bench.DryFrame({"key": (np.arange(5000, dtype=np.int64) * ws + rank), "v": np.ones(5000, np.int64), "x": np.zeros(5000)})
    info = {}
    pdist.sharded_groupby(comm, uniq, spec, fops, mode="auto", info=info)
    save["auto_unique_mode"] = np.array([info["mode"]])
    np.savez(os.path.join(out_dir, f"sgb_rank{rank}.npz"), **save)
    # (f) a scan shared by the ranks: every rank takes its run of row groups (io.split_by_rows through scan_shard()), decodes them --
    # here with pyarrow, the planning and the dictionary agreement are what is under test -- and the per-rank dictionaries of the
    # string column are unified (dist.unify_dictionaries; the device remap is replaced by its numpy twin)
    import pyarrow as pa
    import pyarrow.parquet as pq
    from polars_amd import datatypes as T
    from polars_amd import io
    ds = os.path.join(out_dir, "dataset")
    if rank == 0:
        os.makedirs(ds)
        r2 = np.random.default_rng(77)
        for f, n_f in enumerate((5000, 12000, 3000)):
            words = np.array([f"w{f}", "shared", f"only{f}", "zz"])
            pq.write_table(pa.table({"k": np.arange(n_f) + 100_000 * f, "s": pa.array(words[r2.integers(0, 4, n_f)], mask=r2.random(n_f) < 0.1)}),
                           os.path.join(ds, f"part-{f}.parquet"), row_group_size=1000 + 500 * f)
    dist.barrier()
    assert pdist.scan_shard() == (rank, ws)
    src = io.ParquetFrame(ds, shard=pdist.scan_shard())
    src.request(None, [("k", io.F.OP_GE, 2000)])                    # prunes the first two row groups of part-0 on every rank alike
    rgs = src.selected_row_groups()
    tables = []
    for g in rgs:
        i, lg = src._dec._map[g]
        tables.append(pq.ParquetFile(src._dec.paths[i]).read_row_group(lg))
    tbl = pa.concat_tables(tables) if tables else pa.table({"k": pa.array([], pa.int64()), "s": pa.array([], pa.string())})
    enc = tbl.column("s").combine_chunks().dictionary_encode()

    class Col:
        def __init__(self, name, codes, valid, dtype):
            self.name, self.codes, self.valid, self.dtype = name, codes, valid, dtype

    class Frame:
        def __init__(self, cols): self.cols = list(cols)
        def get_columns(self): return list(self.cols)
        def __getitem__(self, n): return next(c for c in self.cols if c.name == n)

    valid = np.array([x is not None for x in enc.indices.to_pylist()], bool)
    codes = np.array([x or 0 for x in enc.indices.to_pylist()], np.uint32)
    local = Frame([Col("k", tbl.column("k").to_numpy(), None, T.Int64), Col("s", codes, valid, T.Categorical(enc.dictionary.to_pylist(), T.UInt32))])
    one = pdist.unify_dictionaries(local, remap=lambda c, table, union: Col(c.name, table[c.codes] if len(table) else c.codes, c.valid, T.Categorical(union, T.UInt32)))
    np.savez(os.path.join(out_dir, f"scan_rank{rank}.npz"), rgs=np.array(rgs, np.int64), k=one["k"].codes, codes=one["s"].codes, valid=one["s"].valid,
             union=np.array(list(one["s"].dtype.categories), dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
