"""Sort / top-k timings on the GPU (not the headline metric): rows/s of plx_sort_indices for a few key shapes."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import polars_amd as pl  # noqa: E402

pl.init(0)
F = pl._ffi
out = {}
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
cases = {
    "i64_full_range": torch.randint(-2 ** 62, 2 ** 62, (n,), device="cuda", dtype=torch.int64, generator=g),
    "f64_uniform": torch.rand((n,), device="cuda", dtype=torch.float64, generator=g) * 1e5,
    "i32_1e6_values": torch.randint(0, 1_000_000, (n,), device="cuda", dtype=torch.int32, generator=g),
}
for name, t in cases.items():
    s = pl.Series.from_torch(name, t)
    for limit in (-1, 10):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); F.check(F.lib().plx_synchronize())
            t0 = time.perf_counter()
            idx = s.arg_sort(descending=(limit > 0), nulls_last=True, limit=limit)
            F.check(F.lib().plx_synchronize())
            best = min(best, time.perf_counter() - t0)
        out[f"{name}_{'full' if limit < 0 else 'top10'}"] = {"rows": n, "ms": round(best * 1e3, 3), "rows_per_s": round(n / best, 1), "plan": pl.last_plan()}
    del s
print(json.dumps(out, indent=1))
