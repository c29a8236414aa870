#!/usr/bin/env python
"""Runs bench.py's secondary workloads in its order inside ONE process and prints each one's plan string and step time:
which kernels a workload gets must not depend on what ran before it.  usage (GPU box): python tools/debug_extras_order.py [names...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import polars_amd as pl  # noqa: E402


def main():
    import torch
    names = sys.argv[1:] or ["q3", "q3f", "cfg2", "cfg3", "cfg5"]
    pl.init(0)
    for name in names:
        w = bench.make_workload(pl, name, 0, seed=20)
        dt, stats, res, cold = bench.timed(pl, w, 3, 1, False)
        print(name, "ms/step", round(dt / 3 * 1e3, 3), "cold", None if cold is None else round(cold, 2), {k: round(v[1] / v[0]) for k, v in stats.items()}, flush=True)
        print("   plan:", pl.last_plan()[:700], flush=True)
        del w, res
        pl._ffi.lib().plx_memory_trim()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
