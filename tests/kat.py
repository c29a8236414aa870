"""Decoding of tests/golden/reference_kats.json into numpy inputs (shared by the oracle and
the GPU golden tests)."""
from __future__ import annotations

import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NP = {"i8": np.int8, "i16": np.int16, "i32": np.int32, "i64": np.int64, "u8": np.uint8, "u16": np.uint16, "u32": np.uint32,
      "u64": np.uint64, "f32": np.float32, "f64": np.float64, "bool": np.bool_}


def load_cases(kind=None):
    with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
        cases = json.load(f)
    return [c for c in cases if kind is None or c["kind"] == kind]


def scalar(v):
    if isinstance(v, str):
        return {"nan": math.nan, "-nan": -math.nan, "inf": math.inf, "-inf": -math.inf}.get(v, v)
    return v


def expand(spec):
    if isinstance(spec, dict) and "repeat" in spec:
        return list(spec["repeat"]) * int(spec["times"])
    return list(spec)


def column(spec, dtype):
    """-> (values ndarray, validity bool ndarray or None, categories or None).
    Strings become dictionary codes over the sorted distinct values (what the backend's
    upload does)."""
    vals = expand(spec)
    valid = np.array([v is not None for v in vals], dtype=bool)
    if dtype == "str":
        cats = sorted({v for v in vals if v is not None})
        lut = {c: i for i, c in enumerate(cats)}
        arr = np.array([lut[v] if v is not None else 0 for v in vals], dtype=np.uint32)
        return arr, (None if valid.all() else valid), cats
    arr = np.array([scalar(v) if v is not None else 0 for v in vals], dtype=NP[dtype])
    return arr, (None if valid.all() else valid), None


def same_value(got, exp, rtol=1e-12):
    """got: python scalar or None; exp: JSON scalar."""
    exp = scalar(exp)
    if exp is None or got is None:
        return exp is None and got is None
    if isinstance(exp, str) or isinstance(got, str):
        return got == exp
    if isinstance(exp, float) and math.isnan(exp):
        return isinstance(got, float) and math.isnan(got)
    if isinstance(exp, float) or isinstance(got, float):
        if math.isinf(exp):
            return got == exp
        return math.isclose(float(got), float(exp), rel_tol=rtol, abs_tol=0.0) or float(got) == float(exp)
    return got == exp


def filter_sweep_inputs(dtype, size, selectivity):
    """py-polars/tests/unit/operations/test_filter.py:271-286 -- same generator, same seeds."""
    rng = np.random.Generator(np.random.PCG64(size * 100 + int(100 * selectivity)))
    payload = rng.uniform(size=size) * 100.0
    mask = rng.uniform(size=size) < selectivity
    if dtype == "bool":
        p = payload != 0.0   # Series.cast(Boolean): non-zero -> true
    else:
        p = payload.astype(NP[dtype])  # Series.cast(int): truncation toward zero (values in [0, 100))
    return p, mask, p[mask]
