"""Worker of tests/test_dist_gloo_cpu.py: one process per rank, gloo backend, numpy frames.
What is under test is polars_amd.dist -- the control flow of the sharded operators (mode choice agreed across ranks, partial / final
aggregate decomposition, which frames cross the fabric, the merge on the owner) -- with the numpy doubles of tests/dry_multigpu.py standing in for the
device frames, the library's RCCL communicator and the per-rank operators."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc  # noqa: E402
from polars_amd import dist as pdist  # noqa: E402


def main():
    pdist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    out_dir = sys.argv[1]
    import bench  # noqa: F401 -- (the verification helpers)
    import dry_multigpu as dry
    # (a) the exchange double itself: every row lands on ONE rank per key, nothing lost, all-gather = concatenation in rank order
    rng = np.random.default_rng(1000 + rank)
    n = 20_000 + 1000 * rank
    key = rng.integers(0, 3000, n).astype(np.int64)
    v = rng.integers(-50, 50, n).astype(np.int64)
    comm0 = dry.DryComm()
    moved = comm0.exchange_by_key(dry.DryFrame({"key": key, "v": v}), "key")
    owners = comm0.allgather(dry.DryFrame({"key": np.unique(moved.cols["key"]), "r": np.full(len(np.unique(moved.cols["key"])), rank, np.int64)}))
    assert len(np.unique(owners.cols["key"])) == owners.height, "a key landed on two ranks"
    tot = torch.tensor([moved.height, int(moved.cols["v"].sum())], dtype=torch.int64)
    mine = torch.tensor([n, int(v.sum())], dtype=torch.int64)
    dist.all_reduce(tot); dist.all_reduce(mine)
    assert torch.equal(tot, mine), "rows or values lost in the all-to-all"
    assert comm0.total(rank + 1) == ws * (ws + 1) / 2
    # (d) bench.py --gpus N: per-rank Q1 frames are all-gathered as one fixed-size tensor and merged on every rank
    from polars_amd import datagen
    li = datagen.lineitem_host(30_000 + 500 * rank, seed=200 + rank)
    qcols = {c: li[c] for c in datagen.LINEITEM_Q1_COLS}
    mine = {c: a.tolist() for c, a in orc.q1_native(qcols, datagen.us(1998, 9, 2), streaming=True).items()}
    merged = bench.combine_q1_results(bench.allgather_q1(mine, ws))
    np.savez(os.path.join(out_dir, f"q1_rank{rank}.npz"), **{c: np.asarray(a) for c, a in qcols.items()}, **{"m_" + c: np.asarray(a) for c, a in merged.items()})
    # (e) sharded join -> group-by (Q3 shape; dist.sharded_join_groupby: what bench.py --gpus N --workload q3 runs), all three modes
    orders, li = datagen.orders_lineitem_host(30000 + 300 * rank, seed=300 + rank)
    # make order keys globally unique across ranks (each rank generated its own key space)
    off = rank * 10_000_000
    orders["o_orderkey"] = orders["o_orderkey"] + off; li["l_orderkey"] = li["l_orderkey"] + off
    # scatter this rank's lineitem rows so that keys of one order also live on OTHER ranks' probe shards
    comm = dry.DryComm()
    allrows = comm.allgather(dry.DryFrame({c: li[c] for c in datagen.LINEITEM_Q3_COLS}))
    probe = dry.DryFrame({c: np.ascontiguousarray(allrows.cols[c][rank::ws]) for c in datagen.LINEITEM_Q3_COLS})
    build = dry.DryFrame({c: orders[c] for c in datagen.ORDERS_Q3_COLS})
    date = datagen.us(1995, 3, 15)
    jops, jspec = dry.DryJoinOps(date), pdist.JoinGroupBySpec("l_orderkey", "o_orderkey", "l_orderkey", [("revenue", "sum")])
    # the numpy double of the local operator against the oracle's q3 on this rank's inputs
    w = orc.q3(probe.cols, build.cols, date)
    g = jops.local(probe, build)
    o = np.argsort(g.cols["l_orderkey"])
    assert np.array_equal(g.cols["l_orderkey"][o], w["l_orderkey"]) and np.allclose(g.cols["revenue"][o], w["revenue"], rtol=1e-12)
    saved = pdist.BROADCAST_BUILD_BYTES
    for mode in ("broadcast", "shuffle", "auto", "auto_small_limit"):
        pdist.BROADCAST_BUILD_BYTES = 1000 if mode == "auto_small_limit" else saved      # a build side over the limit: "auto" must shuffle
        comm.rows_sent = comm.bytes_sent = 0
        info = {}
        r = pdist.sharded_join_groupby(comm, jops, probe, build, jspec, mode=mode.split("_")[0], info=info)
        np.savez(os.path.join(out_dir, f"q3_{mode}_rank{rank}.npz"), mode=np.array([info["mode"]]), rows_sent=np.array([comm.rows_sent]), **r.cols)
    pdist.BROADCAST_BUILD_BYTES = saved
    np.savez(os.path.join(out_dir, f"q3_in_rank{rank}.npz"), **{"p_" + c: a for c, a in probe.cols.items()}, **{"b_" + c: a for c, a in build.cols.items()})
    # (g) the frame-level sharded group-by (dist.sharded_groupby: what bench.py --gpus N --workload cfg3 runs): pre-aggregation before the
    # exchange vs raw-row exchange vs the sample-driven choice, with NULL keys (one group, owned by rank 0) and null values; numpy doubles
    # stand in for the library frames and the RCCL communicator (tests/dry_multigpu.py: DryFrame / DryComm / DryOps)
    r3 = np.random.default_rng(4000 + rank)
    m = 30_000 + 700 * rank
    gk = r3.integers(0, 2500, m).astype(np.int64); gk_valid = r3.random(m) > 0.02
    gv = r3.integers(-1000, 1000, m).astype(np.int64); gv_valid = r3.random(m) > 0.1
    gv_valid[gk == 7] = False                                     # a group whose values are ALL null: sum 0, count 0, mean / min null
    gx = r3.uniform(-1, 1, m)
    shard = dry.DryFrame({"key": gk, "v": gv, "x": gx}, {"key": gk_valid, "v": gv_valid})
    spec = pdist.GroupBySpec("key", [("v_sum", "v", "sum"), ("v_count", "v", "count"), ("v_mean", "v", "mean"), ("v_min", "v", "min"), ("x_max", "x", "max"), ("n", "", "len")])
    comm, fops = dry.DryComm(), dry.DryOps()
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
    uniq = dry.DryFrame({"key": (np.arange(5000, dtype=np.int64) * ws + rank), "v": np.ones(5000, np.int64), "x": np.zeros(5000)})
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
