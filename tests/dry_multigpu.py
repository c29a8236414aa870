"""numpy + gloo doubles of the library's frames, communicator and local operators: what `bench.py --dry-run` and tests/test_dist_gloo_cpu.py run the
N > 1 control flow on without GPUs (mode choice, barriers, accounting, the JSON lines).  Test infrastructure: nothing here is measured or shipped, and
bench.py imports it only under --dry-run."""
from __future__ import annotations

Q1_FIELDS = ("l_returnflag", "l_linestatus", "sum_qty", "count_order", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc")


class DryFrame:
    """numpy stand-in for a device DataFrame (dry run only): name -> values, plus name -> validity (bool array) for nullable columns."""

    def __init__(self, cols, valid=None, schema=None):
        self.cols = dict(cols)
        self.valid = {k: v for k, v in (valid or {}).items() if v is not None}
        self.schema = schema

    @property
    def height(self):
        return len(next(iter(self.cols.values()))) if self.cols else 0

    def validity(self, name):
        import numpy as np
        v = self.valid.get(name)
        return np.ones(self.height, bool) if v is None else v


class DryComm:
    """gloo stand-in for dist.LibComm (dry run only): same routing rule shape (a hash of the key modulo world size, null keys to rank 0:
    hashing.rs:111-115), one all_to_all per column, validity as one byte per row."""

    def __init__(self):
        import torch.distributed as dist
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self.rows_sent = self.bytes_sent = 0

    def agree(self, value):
        import torch.distributed as dist
        box = [float(value)]
        dist.broadcast_object_list(box, src=0)
        return float(box[0])

    def total(self, value):
        import torch.distributed as dist
        every = [None] * self.world_size
        dist.all_gather_object(every, float(value))
        return float(sum(every))

    def allgather(self, df):
        """concatenation of every rank's frame in rank order (pickled numpy over gloo: dry run only)"""
        import numpy as np
        import torch.distributed as dist
        every = [None] * self.world_size
        dist.all_gather_object(every, (df.cols, {n: df.validity(n) for n in df.valid}))
        nullable = {n for _, v in every for n in v}
        cols = {n: np.concatenate([c[n] for c, _ in every]) for n in df.cols}
        valid = {n: np.concatenate([v[n] if n in v else np.ones(len(c[n]), bool) for c, v in every]) for n in nullable}
        return DryFrame(cols, valid, df.schema)

    def exchange_by_key(self, df, key, seed=0):
        import numpy as np
        import torch
        import torch.distributed as dist
        ws = self.world_size
        kv = df.cols[key]
        k = (kv.view(np.uint64) if kv.dtype.itemsize == 8 else kv.astype(np.uint64))
        part = ((k * np.uint64(0x55fbfd6bfc5458e9)) >> np.uint64(40)) % np.uint64(ws)
        part = np.where(df.validity(key), part, np.uint64(0)).astype(np.int64)
        order = np.argsort(part, kind="stable")
        counts = np.bincount(part, minlength=ws).astype(np.int64)
        send = torch.from_numpy(counts.copy()); recv = torch.zeros_like(send)
        dist.all_to_all_single(recv, send)
        rc = [int(x) for x in recv.tolist()]
        # a column travels with a validity byte per row when ANY rank holds nulls in it (the sends and receives must pair up)
        has = torch.tensor([1 if n in df.valid else 0 for n in df.cols], dtype=torch.int64)
        dist.all_reduce(has, op=dist.ReduceOp.MAX)
        out, out_valid = {}, {}
        away = int(sum(int(c) for i, c in enumerate(counts) if i != self.rank))

        def a2a(v):
            src = torch.from_numpy(np.ascontiguousarray(v[order]).view(np.uint8).reshape(-1))
            w = v.dtype.itemsize
            dst = torch.empty(sum(rc) * w, dtype=torch.uint8)
            dist.all_to_all_single(dst, src, output_split_sizes=[c * w for c in rc], input_split_sizes=[int(c) * w for c in counts])
            self.bytes_sent += away * w
            return dst.numpy().view(v.dtype)
        for (name, v), nullable in zip(df.cols.items(), has.tolist()):
            out[name] = a2a(v)
            if nullable:
                out_valid[name] = a2a(df.validity(name).astype(np.uint8)).astype(bool)
        self.rows_sent += away
        return DryFrame(out, out_valid, df.schema)


class DryOps:
    """numpy stand-in for dist.LibFrameOps (dry run only): the same four local queries over DryFrames, null-aware (null key = its own
    group; sum / count / min / max / mean skip null values; min / max / mean of no value = null)."""

    @staticmethod
    def _groups(df, key):
        import numpy as np
        kv, valid = df.cols[key], df.validity(key)
        uniq, inv = np.unique(kv[valid], return_inverse=True)
        gid = np.full(df.height, len(uniq), np.int64)
        gid[valid] = inv
        has_null = bool((~valid).any())
        keys = np.concatenate([uniq, np.zeros(1, kv.dtype)]) if has_null else uniq
        kvalid = np.concatenate([np.ones(len(uniq), bool), np.zeros(1, bool)]) if has_null else None
        return gid, len(keys), keys, kvalid

    def _aggregate(self, df, key, aggs):
        import numpy as np
        gid, ng, keys, kvalid = self._groups(df, key)
        cols, valid = {key: keys}, {key: kvalid}
        for out, col, op in aggs:
            if op == "len":
                cols[out] = np.bincount(gid, minlength=ng).astype(np.uint32)
                continue
            v, ok = df.cols[col], df.validity(col)
            g = gid[ok]
            if op == "count":
                cols[out] = np.bincount(g, minlength=ng).astype(np.uint32)
            elif op in ("sum", "sum_f64"):
                x = v[ok].astype(np.float64) if op == "sum_f64" or v.dtype.kind == "f" else v[ok].astype(np.uint32 if v.dtype == np.uint32 else np.int64)
                acc = np.zeros(ng, x.dtype)
                np.add.at(acc, g, x)
                cols[out] = acc
            elif op in ("min", "max"):
                fn, init = (np.minimum, np.inf) if op == "min" else (np.maximum, -np.inf)
                if v.dtype.kind == "f":
                    acc = np.full(ng, init, v.dtype)
                else:
                    info = np.iinfo(v.dtype)
                    acc = np.full(ng, info.max if op == "min" else info.min, v.dtype)
                fn.at(acc, g, v[ok])
                seen = np.bincount(g, minlength=ng) > 0
                cols[out] = np.where(seen, acc, np.zeros(1, v.dtype))
                valid[out] = None if seen.all() else seen
            else:
                raise ValueError(op)
        return DryFrame(cols, valid, df.schema)

    def final(self, df, spec):
        import numpy as np
        from polars_amd.dist import PARTIALS
        part = self._aggregate(df, spec.key, [(f"{o}__p{i}", c, pop) for o, c, op in spec.aggs for i, (pop, _) in enumerate(PARTIALS[op])])
        return self._finish(part, spec)

    def partial(self, df, spec):
        return self._aggregate(df, spec.key, spec.partial_aggs())

    def merge(self, part, spec, source_schema=None):
        return self._finish(self._aggregate(part, spec.key, spec.merge_aggs()), spec)

    @staticmethod
    def _finish(part, spec):
        import numpy as np
        cols, valid = {spec.key: part.cols[spec.key]}, {spec.key: part.valid.get(spec.key)}
        for o, c, op in spec.aggs:
            if op == "mean":
                n = part.cols[f"{o}__p1"].astype(np.float64)
                with np.errstate(invalid="ignore", divide="ignore"):
                    cols[o] = np.where(n > 0, part.cols[f"{o}__p0"] / np.where(n > 0, n, 1.0), 0.0)
                valid[o] = None if (n > 0).all() else n > 0
            else:
                cols[o] = part.cols[f"{o}__p0"]
                valid[o] = part.valid.get(f"{o}__p0")
        return DryFrame(cols, valid, part.schema)

    def distinct_in_prefix(self, df, key, n):
        import numpy as np
        return len(np.unique(df.cols[key][:n][df.validity(key)[:n]])) + int((~df.validity(key)[:n]).any())


class DryJoinOps:
    """numpy stand-in for dist.LibJoinOps on TPC-H Q3's shape (dry run only): the two single-table predicates, the local
    filter -> join -> group-by, the merge of partial groups."""

    def __init__(self, date, seg_mod=5):
        self.date, self.seg_mod = date, seg_mod

    @staticmethod
    def _take(df, m):
        return DryFrame({c: v[m] for c, v in df.cols.items()}, None, df.schema)

    def build_prefilter(self, df):
        return self._take(df, (df.cols["o_orderdate"] < self.date) & (df.cols["o_custkey"] % self.seg_mod == 0))

    def probe_prefilter(self, df):
        return self._take(df, df.cols["l_shipdate"] > self.date)

    def local(self, probe, build):
        import numpy as np
        b, p = self.build_prefilter(build), self.probe_prefilter(probe)
        order = np.argsort(b.cols["o_orderkey"], kind="stable")
        bk = b.cols["o_orderkey"][order]
        pos = np.searchsorted(bk, p.cols["l_orderkey"])
        hit = (pos < len(bk)) & (bk[np.minimum(pos, max(len(bk) - 1, 0))] == p.cols["l_orderkey"]) if len(bk) else np.zeros(p.height, bool)
        slot = pos[hit]
        rev = p.cols["l_extendedprice"][hit] * (1.0 - p.cols["l_discount"][hit])
        sums = np.bincount(slot, weights=rev, minlength=len(bk))
        has = np.bincount(slot, minlength=len(bk)) > 0
        return DryFrame({"l_orderkey": bk[has], "o_orderdate": b.cols["o_orderdate"][order][has], "o_shippriority": b.cols["o_shippriority"][order][has], "revenue": sums[has]})

    def merge(self, part, spec):
        import numpy as np
        uniq, first, inv = np.unique(part.cols[spec.result_key], return_index=True, return_inverse=True)
        out = {c: v[first] for c, v in part.cols.items()}
        for c, op in spec.merge:
            assert op == "sum"
            out[c] = np.bincount(inv, weights=part.cols[c], minlength=len(uniq))
        return DryFrame(out)

    def nbytes(self, df):
        return int(sum(v.nbytes for v in df.cols.values()))


def dry_q1_step(n: int, seed: int):
    """numpy stand-in for the per-rank Q1 (dry run only): the result in the layout of DataFrame.to_dict()."""
    import numpy as np
    from polars_amd import datagen
    li = datagen.lineitem_native_host_mt(0, n, seed, threads=2)
    cutoff = datagen.us(1998, 9, 2)

    def step():
        m = li["l_shipdate"] <= cutoff
        g = (li["l_returnflag"][m].astype(np.int64) * 2 + li["l_linestatus"][m].astype(np.int64))
        cnt = np.bincount(g, minlength=6)
        qty, price, disc, tax = (li[c][m] for c in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"))
        s = lambda w: np.bincount(g, weights=w, minlength=6)
        dp = price * (1 - disc)
        out = {f: [] for f in Q1_FIELDS}
        for i in np.nonzero(cnt)[0]:
            c = int(cnt[i])
            out["l_returnflag"].append(int(i) // 2); out["l_linestatus"].append(int(i) % 2)
            out["sum_qty"].append(int(qty[g == i].sum())); out["count_order"].append(c)
            out["sum_base_price"].append(float(s(price)[i])); out["sum_disc_price"].append(float(s(dp)[i])); out["sum_charge"].append(float(s(dp * (1 + tax))[i]))
            out["avg_qty"].append(float(s(qty.astype(np.float64))[i]) / c); out["avg_price"].append(float(s(price)[i]) / c); out["avg_disc"].append(float(s(disc)[i]) / c)
        return out
    return step
