"""N > 1 path on CPU: world_size 2 and 3, gloo backend, one process per rank
(python -m torch.distributed.run, rendezvous on 127.0.0.1).  Checks polars_amd.dist against a
single-process pandas evaluation of the concatenated shards."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_lines  # noqa: E402


@pytest.mark.parametrize("ws", [2, 3])
def test_sharded_groupby_and_exchange(tmp_path, ws):
    pd = pytest.importorskip("pandas")
    port = 29511 + ws
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ws}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # Q1 merged across ranks == oracle on the concatenated shards, identical on every rank
    from oracle import pyoracle as orc
    from polars_amd import datagen
    qfiles = [np.load(f) for f in sorted(glob.glob(str(tmp_path / "q1_rank*.npz")))]
    assert len(qfiles) == ws
    whole = orc.q1_native({c: np.concatenate([q[c] for q in qfiles]) for c in datagen.LINEITEM_Q1_COLS}, datagen.us(1998, 9, 2), streaming=True)
    for q in qfiles:
        for c, v in whole.items():
            got = q["m_" + c]
            assert np.allclose(got.astype(np.float64), v.astype(np.float64), rtol=1e-9), c
    # sharded join -> group-by: both modes equal the oracle's q3 on the concatenated inputs; results disjoint by key
    ins = [np.load(f) for f in sorted(glob.glob(str(tmp_path / "q3_in_rank*.npz")))]
    li_all = {c: np.concatenate([i["p_" + c] for i in ins]) for c in datagen.LINEITEM_Q3_COLS}
    or_all = {c: np.concatenate([i["b_" + c] for i in ins]) for c in datagen.ORDERS_Q3_COLS}
    exp = orc.q3(li_all, or_all, datagen.us(1995, 3, 15))
    assert len(exp["l_orderkey"]) > 100
    sent = {}
    for mode, took in (("broadcast", "broadcast"), ("shuffle", "shuffle"), ("auto", "broadcast"), ("auto_small_limit", "shuffle")):
        outs = [np.load(f) for f in sorted(glob.glob(str(tmp_path / f"q3_{mode}_rank*.npz")))]
        assert len(outs) == ws and all(str(o["mode"][0]) == took for o in outs), mode      # the size-driven choice, agreed across ranks
        sent[mode] = sum(int(o["rows_sent"][0]) for o in outs)
        keys = [set(o["l_orderkey"].tolist()) for o in outs]
        for i in range(ws):
            for j in range(i + 1, ws):
                assert not (keys[i] & keys[j]), mode
        got = {c: np.concatenate([o[c] for o in outs]) for c in exp}
        order = np.argsort(got["l_orderkey"], kind="stable")
        assert np.array_equal(got["l_orderkey"][order], exp["l_orderkey"]), mode
        assert np.array_equal(got["o_orderdate"][order], exp["o_orderdate"]) and np.array_equal(got["o_shippriority"][order], exp["o_shippriority"]), mode
        assert np.allclose(got["revenue"][order], exp["revenue"], rtol=1e-9), mode
    # broadcast moves only partial groups (the build side is all-gathered, not counted as routed rows); shuffle moves the filtered rows of both sides
    assert 0 < sent["broadcast"] < sent["shuffle"] and sent["auto"] == sent["broadcast"] and sent["auto_small_limit"] == sent["shuffle"]
    # the shared scan: shards are disjoint contiguous runs covering every row group the predicate keeps, every rank ends up with the
    # SAME dictionary, and codes decoded through it give back the file's strings
    import pyarrow.parquet as pq
    scans = [np.load(f, allow_pickle=True) for f in sorted(glob.glob(str(tmp_path / "scan_rank*.npz")))]
    assert len(scans) == ws
    got_rgs = [s["rgs"].tolist() for s in scans]
    flat = [g for r in got_rgs for g in r]
    n_groups = 5 + 8 + 2                                              # 5000 / 1000, 12000 / 1500, 3000 / 2000 (rounded up)
    assert flat == list(range(2, n_groups)), got_rgs                  # in rank order: contiguous, disjoint, complete; k >= 2000 dropped two
    rows = [len(s["k"]) for s in scans]
    assert max(rows) - min(rows) <= 2000 + 1000, rows                 # balanced to within about a row group
    union = scans[0]["union"].tolist()
    assert all(s["union"].tolist() == union for s in scans) and len(set(union)) == len(union) == 8
    whole = pq.read_table(str(tmp_path / "dataset")).to_pandas().sort_values("k")
    whole = whole[whole["k"] >= 2000 - 0]                              # row groups 0 and 1 of part-0 hold k < 2000 exactly
    k = np.concatenate([s["k"] for s in scans]); order = np.argsort(k, kind="stable")
    words = np.array([union[c] if ok else None for s in scans for c, ok in zip(s["codes"].tolist(), s["valid"].tolist())], dtype=object)[order]
    assert np.array_equal(k[order], whole["k"].to_numpy())
    want = whole["s"].to_numpy()
    assert all((a is None and (b is None or b != b)) or a == b for a, b in zip(words.tolist(), want.tolist()))
    # the frame-level sharded group-by: every mode gives the pandas answer on the concatenated shards (null key = one group, on rank 0;
    # all-null groups: sum 0, count 0, mean / min null); pre-aggregation moves far fewer rows than the raw-row exchange
    sg = [np.load(f) for f in sorted(glob.glob(str(tmp_path / "sgb_rank*.npz")))]
    assert len(sg) == ws
    kk = np.concatenate([s["in_key"] for s in sg]).astype(np.float64); kk[~np.concatenate([s["in_key_valid"] for s in sg])] = np.nan
    vv = np.concatenate([s["in_v"] for s in sg]).astype(np.float64); vv[~np.concatenate([s["in_v_valid"] for s in sg])] = np.nan
    whole_df = pd.DataFrame({"key": kk, "v": vv, "x": np.concatenate([s["in_x"] for s in sg])})
    gb = whole_df.groupby("key", dropna=False)
    want = pd.DataFrame({"v_sum": gb["v"].sum(), "v_count": gb["v"].count(), "v_mean": gb["v"].mean(), "v_min": gb["v"].min(), "x_max": gb["x"].max(), "n": gb.size()}).reset_index()
    want = want.sort_values("key", na_position="last").reset_index(drop=True)
    for mode in ("preagg", "rows", "auto"):
        got_k = np.concatenate([s[f"{mode}_key"] for s in sg]).astype(np.float64)
        got_kv = np.concatenate([s[f"{mode}_key__valid"] for s in sg])
        assert sum(int((~s[f"{mode}_key__valid"]).sum()) for s in sg[1:]) == 0 and int((~sg[0][f"{mode}_key__valid"]).sum()) == 1, mode     # the null group lives on rank 0 only
        got_k[~got_kv] = np.inf
        order = np.argsort(got_k, kind="stable")
        assert len(got_k) == len(want) and len(np.unique(got_k)) == len(got_k), mode                       # disjoint key sets
        wk = want["key"].to_numpy().copy(); wk[np.isnan(wk)] = np.inf
        assert np.array_equal(got_k[order], wk), mode
        col = lambda c: np.concatenate([s[f"{mode}_{c}"] for s in sg])[order]
        colv = lambda c: np.concatenate([s[f"{mode}_{c}__valid"] for s in sg])[order]
        assert np.array_equal(col("v_sum"), want["v_sum"].to_numpy().astype(np.int64)) and np.array_equal(col("v_count"), want["v_count"].to_numpy()), mode
        assert np.array_equal(col("n"), want["n"].to_numpy()) and np.array_equal(col("x_max"), want["x_max"].to_numpy()), mode
        has = want["v_count"].to_numpy() > 0
        assert np.array_equal(colv("v_mean"), has) and np.array_equal(colv("v_min"), has) and not has.all(), mode
        assert np.allclose(col("v_mean")[has], want["v_mean"].to_numpy()[has], rtol=1e-12) and np.array_equal(col("v_min")[has], want["v_min"].to_numpy()[has].astype(np.int64)), mode
    pre, raw = sum(int(s["preagg_rows_sent"][0]) for s in sg), sum(int(s["rows_rows_sent"][0]) for s in sg)
    assert pre < raw / 5 and raw > 0.4 * len(whole_df) * (ws - 1) / ws                                     # ~2500 groups per rank instead of ~30000 rows
    assert all(str(s["auto_mode"][0]) == "preagg" and str(s["auto_unique_mode"][0]) == "rows" for s in sg)   # the sample-driven choice, agreed across ranks


def test_partial_final_decomposition_table():
    from polars_amd import dist as pdist
    assert pdist.PARTIALS["mean"] == [("sum_f64", "sum"), ("count", "sum")]          # reduce/mean.rs keeps (f64 sum, count)
    assert pdist.PARTIALS["count"] == [("count", "sum")] and pdist.PARTIALS["min"] == [("min", "min")]


def test_bench_q1_rank_combine_matches_single_shard():
    """bench.py --gpus N: the per-rank Q1 frames are merged with combine_q1_results; merging the oracle's results on
    two row shards must equal the oracle on the whole table."""
    import numpy as np
    import bench
    from oracle import pyoracle as orc
    from polars_amd import datagen
    li = datagen.lineitem_host(60_000, seed=3)
    cols = {k: li[k] for k in datagen.LINEITEM_Q1_COLS}
    cut = datagen.us(1998, 9, 2)
    whole = orc.q1_native(cols, cut, streaming=True)
    parts = []
    for lo, hi in ((0, 25_001), (25_001, 60_000)):
        r = orc.q1_native({k: v[lo:hi] for k, v in cols.items()}, cut, streaming=True)
        parts.append({k: v.tolist() for k, v in r.items()})
    # through the fixed-size tensor packing used by the all-gather
    import torch
    packed = torch.stack([bench.pack_q1(p) for p in parts])
    parts2 = bench.unpack_q1(packed)
    merged = bench.combine_q1_results(parts2)
    for k, v in whole.items():
        if v.dtype.kind == "f":
            assert np.allclose(np.array(merged[k]), v, rtol=1e-9), k
        else:
            assert merged[k] == v.tolist(), k


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2 --dry-run` WITHOUT torchrun (the shape of the driver's N = 1 command with another N): bench.py starts the
    two ranks itself and rank 0 prints ONE line with n_gpus = 2 -- the Q1 headline plus the sharded Q3 (BASELINE config 4, strong scaling,
    both sides exchanged by key hash), cfg3 and cfg5 as extras, every one checked by rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--rows", "120000"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    head, d = bench_lines.split(r.stdout)
    # the line the driver parses: SF100 in TOTAL (strong scaling) by default at N > 1, small, and it carries the verdicts of the extras
    assert head["n_gpus"] == 2 and head["scaling"] == "strong" and head["config"]["workload"] == "tpch_q1_sf100_x2_strong" and head["verified"]["ok"] is True
    assert "roofline" in head and "extras" not in head and head["extras_file"]
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["config"]["workload"] == "tpch_q1_sf100_x2_strong" and d["verified"]["ok"] is True
    ex = d["extras"]
    assert set(ex) == {"tpch_q3_sf100_sharded_x2_strong", "tpch_q3_sf100_sharded_x2_shuffle_strong", "cfg3_groupby_1e6_keys_sharded_x2_strong", "cfg5_dict_string_keys_sharded_x2_strong",
                       "tpch_q1_sf100_x2_weak"}, list(ex)
    assert set(head["extras_summary"]) == set(ex) and all(v["ok"] is True for v in head["extras_summary"].values())
    assert ex["tpch_q1_sf100_x2_weak"]["scaling"] == "weak" and ex["tpch_q1_sf100_x2_weak"]["verified"]["ok"] is True
    qa = ex["tpch_q3_sf100_sharded_x2_strong"]
    assert qa["scaling"] == "strong" and qa["exchange_mode"] == "broadcast" and qa["verified"]["ok"] is True and qa["partial_rows_per_rank"] > 0
    q3 = ex["tpch_q3_sf100_sharded_x2_shuffle_strong"]
    assert q3["scaling"] == "strong" and q3["exchange_mode"] == "shuffle" and q3["verified"]["ok"] is True and q3["verified"]["keys_disjoint_across_ranks"] is True
    assert all(p["covers_whole_input"] and p["ok"] for p in q3["verified"]["per_rank"]) and q3["shuffle"]["rows_sent_per_rank_per_step"] > 0
    assert q3["shuffle"]["bytes_sent_per_rank_per_step"] == q3["shuffle"]["rows_sent_per_rank_per_step"] * 32        # four 8-byte columns on either side
    for w in ("cfg3_groupby_1e6_keys_sharded_x2_strong", "cfg5_dict_string_keys_sharded_x2_strong"):
        assert ex[w]["verified"]["ok"] is True and ex[w]["scaling"] == "strong", w


def test_sharded_groupby_bench_dry_run_prints_a_complete_line():
    """bench.py --gpus 2 --workload cfg3 --dry-run under gloo: the sharded operator's control flow (exchange by key hash, per-rank
    group-by over disjoint key sets, barriers, max-over-ranks timing, shuffle accounting) with numpy frames standing in for the
    library -- so the first multi-GPU run is a measurement, not a debug session.  The key sets of the ranks must be disjoint and
    cover the union: groups_total == distinct keys of both shards together."""
    import json
    import subprocess
    import sys

    import numpy as np

    from polars_amd import datagen
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = 200_000
    for wl, kdt in (("cfg3", "Int64"), ("cfg5", "UInt32")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
               os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", wl, "--rows", str(rows), "--dry-run", "--scaling", "weak"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
        assert r.returncode == 0, r.stderr[-2000:]
        head, d = bench_lines.split(r.stdout)
        assert all(k in head for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "scaling", "config", "verified")) and head["verified"]["ok"] is True
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "shuffle"):
            assert k in d, k
        assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dry_run"] is True and d["config"]["rows_per_gpu"] == rows
        keys = np.concatenate([datagen.uniform_native_host(kdt, 0, rows, 10 + rank, 0, 0, 1_000_000) for rank in range(2)])
        assert d["groups_total"] == len(np.unique(keys))
        # about half of every shard's rows leave the rank; every row carries key + value
        assert 0.4 * rows < d["shuffle"]["rows_sent_per_rank_per_step"] < 0.6 * rows
        assert d["shuffle"]["bytes_sent_per_rank_per_step"] == d["shuffle"]["rows_sent_per_rank_per_step"] * (12 if wl == "cfg5" else 16)
        assert d["exchange_mode"] == "rows" and d["shrink_estimate"] < 2          # 2e5 rows over 1e6 keys: the local group-by would shrink nothing
        assert d["verified"]["ok"] is True and all(d["verified"]["checks"].values()) and "oracle" in d["verified"]["against"]
    # few keys per rank (keys 0..1e6 but 2e6 rows would be slow here: force the mode instead): partial rows cross the fabric, not rows; strong scaling label
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29542",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg3", "--rows", str(rows), "--dry-run", "--mode", "preagg", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-2000:]
    _, d = bench_lines.split(r.stdout)
    assert d["exchange_mode"] == "preagg" and d["scaling"] == "strong" and d["verified"]["ok"] is True
    local_groups = [len(np.unique(datagen.uniform_native_host("Int64", 0, rows, 10 + rank, 0, 0, 1_000_000))) for rank in range(2)]
    assert d["partial_rows_per_rank"] in local_groups
    assert 0.4 * min(local_groups) < d["shuffle"]["rows_sent_per_rank_per_step"] < 0.6 * max(local_groups)
    assert d["shuffle"]["bytes_sent_per_rank_per_step"] == d["shuffle"]["rows_sent_per_rank_per_step"] * (8 + 8 + 4)     # key + i64 partial sum + u32 partial count


def test_bench_eight_ranks_dry_run_is_quiet_and_complete():
    """`python bench.py --gpus 8 --dry-run` (the largest N the driver launches): eight ranks under gloo, the headline plus all four sharded workloads verified,
    and nothing on stderr that looks like a failure -- the farewell barrier tolerates a peer that has already left."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--rows", "120000"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Traceback" not in r.stderr, r.stderr[-2000:]
    head, d = bench_lines.split(r.stdout)
    assert head["n_gpus"] == 8 and head["scaling"] == "strong" and head["config"]["workload"] == "tpch_q1_sf100_x8_strong"
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["verified"]["ok"] is True
    ex = d["extras"]
    assert set(ex) == {"tpch_q3_sf100_sharded_x8_strong", "tpch_q3_sf100_sharded_x8_shuffle_strong", "cfg3_groupby_1e6_keys_sharded_x8_strong", "cfg5_dict_string_keys_sharded_x8_strong",
                       "tpch_q1_sf100_x8_weak"}
    assert all(v["verified"]["ok"] is True for v in ex.values())
    assert ex["tpch_q3_sf100_sharded_x8_strong"]["scaling"] == "strong" and ex["tpch_q3_sf100_sharded_x8_shuffle_strong"]["exchange_mode"] == "shuffle"
