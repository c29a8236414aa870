#!/usr/bin/env python
"""Writes tests/golden/reference_kats.json: known-answer vectors TRANSCRIBED from the
reference's own tests (pola-rs/polars 0.55.1).  The reference cannot be imported or built in
this image (no rustc, no polars wheel), so these are hand transcriptions; every case cites the
test it comes from.  Both the CPU oracle (tests/test_oracle_golden.py) and the HIP path
(tests/test_gpu_golden.py) are checked against this one file.

null is JSON null; NaN / inf are the strings "nan" / "inf" / "-inf".
Result rows of unordered operators are compared after sorting (as the reference's tests do).
"""
import json
import os

C = []

# ---- group_by ---------------------------------------------------------------------
C.append(dict(id="group_by_sum", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:32-52",
              keys={"a": ["a", "b", "a", "b", "b", "c"]}, key_dtypes={"a": "str"},
              values={"b": [1, 2, 3, 4, 5, 6]}, value_dtypes={"b": "i64"},
              aggs=[["b", "sum"]], maintain_order=True,
              expect={"a": ["a", "b", "c"], "b_sum": [4, 11, 6]}))
C.append(dict(id="group_by_count", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:54-70",
              keys={"b": ["a", "a", "b", "b", "b"]}, key_dtypes={"b": "str"},
              values={"a": [1, 2, 3, 4, 5]}, value_dtypes={"a": "i64"},
              aggs=[["a", "count"]], maintain_order=True,
              expect={"b": ["a", "b"], "a_count": [2, 3]}))
for dt, odt in [("u8", "f64"), ("i8", "f64"), ("u16", "f64"), ("i16", "f64"), ("u32", "f64"), ("i32", "f64"), ("u64", "f64"), ("f32", "f32"), ("f64", "f64")]:
    C.append(dict(id=f"group_by_mean_{dt}", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:82-178",
                  keys={"key": ["a", "a", "a", "b"]}, key_dtypes={"key": "str"},
                  values={"v": [1, 2, 3, 4]}, value_dtypes={"v": dt},
                  aggs=[["v", "mean"]], maintain_order=True,
                  expect={"key": ["a", "b"], "v_mean": [2, 4]}, expect_dtypes={"v_mean": odt}))
for dt in ("i32", "u32"):
    C.append(dict(id=f"group_by_mean_overflow_{dt}", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:886-897",
                  keys={"group": {"repeat": [1, 2], "times": 50000}}, key_dtypes={"group": dt},
                  values={"data": {"repeat": [10000000, 10000000], "times": 50000}}, value_dtypes={"data": dt},
                  aggs=[["data", "mean"]], maintain_order=False,
                  expect={"group": [1, 2], "data_mean": [10000000.0, 10000000.0]}))
C.append(dict(id="group_by_null_keys", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:1124-1132",
              keys={"a": [None, None, None, None], "b": [1, 1, 2, 2]}, key_dtypes={"a": "i64", "b": "i64"},
              values={"c": [10, 20, 30, 40]}, value_dtypes={"c": "i64"},
              aggs=[["c", "len"]], maintain_order=True,
              expect={"a": [None, None], "b": [1, 2], "c_len": [2, 2]},
              note="reference aggregates a string list column; the group structure (null key is a group) is what is pinned"))
C.append(dict(id="group_by_nulls_mean_21838", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:1153-1161",
              keys={"a": [1] * 10 + [2] * 10 + [3] * 10}, key_dtypes={"a": "i64"},
              values={"b": [1] * 10 + [None] * 20}, value_dtypes={"b": "i64"},
              aggs=[["b", "mean"]], maintain_order=False,
              expect={"a": [1, 2, 3], "b_mean": [1.0, None, None]}))
C.append(dict(id="group_by_sum_all_null_f32", kind="groupby", source="py-polars/tests/unit/operations/aggregation/test_aggregations.py:476-488",
              keys={"b": [1, 1, 1]}, key_dtypes={"b": "i64"},
              values={"a": [None, None, None]}, value_dtypes={"a": "f32"},
              aggs=[["a", "sum"]], maintain_order=False,
              expect={"b": [1], "a_sum": [0.0]}))
C.append(dict(id="nan_inf_aggregation", kind="groupby", source="py-polars/tests/unit/operations/aggregation/test_aggregations.py:565-603",
              keys={"group": ["both nan", "both nan", "nan and 5", "nan and 5", "nan and null", "nan and null", "both none", "both none",
                              "both inf", "both inf", "inf and null", "inf and null"]}, key_dtypes={"group": "str"},
              values={"value": ["nan", "nan", "nan", 5, "nan", None, None, None, "inf", "inf", "inf", None]}, value_dtypes={"value": "f64"},
              aggs=[["value", "min"], ["value", "max"], ["value", "mean"]], maintain_order=True,
              expect={"group": ["both nan", "nan and 5", "nan and null", "both none", "both inf", "inf and null"],
                      "value_min": ["nan", 5, "nan", None, "inf", "inf"], "value_max": ["nan", 5, "nan", None, "inf", "inf"],
                      "value_mean": ["nan", "nan", "nan", None, "inf", "inf"]}))
C.append(dict(id="sum_inf_not_nan_25849", kind="groupby", source="py-polars/tests/unit/operations/aggregation/test_aggregations.py:1333-1336",
              keys={"g": ["X"] * 9}, key_dtypes={"g": "str"},
              values={"x": [10.0, None, 10.0, 10.0, 10.0, 10.0, "inf", 10.0, 10.0]}, value_dtypes={"x": "f64"},
              aggs=[["x", "sum"]], maintain_order=False, expect={"g": ["X"], "x_sum": ["inf"]}))

# ---- whole-column reductions ------------------------------------------------------------
C.append(dict(id="mean_overflow", kind="reduce", source="py-polars/tests/unit/operations/aggregation/test_aggregations.py:341-344",
              values=[9223372036854775800, 100], dtype="i64", op="mean", expect=4.611686018427388e18, rtol=1e-9))
C.append(dict(id="sum_empty_f32", kind="reduce", source="py-polars/tests/unit/operations/aggregation/test_aggregations.py:476-478",
              values=[], dtype="f32", op="sum", expect=0.0))
C.append(dict(id="sum_all_null_f32", kind="reduce", source="py-polars/tests/unit/operations/aggregation/test_aggregations.py:480-481",
              values=[None], dtype="f32", op="sum", expect=0.0))

# ---- joins -------------------------------------------------------------------------------
C.append(dict(id="inner_join_days", kind="join", how="inner", source="crates/polars/tests/it/core/joins.rs:40-78",
              left={"days": [0, 1, 2], "temp": [22.1, 19.9, 7.0], "rain": [0.2, 0.1, 0.3]}, left_dtypes={"days": "i32", "temp": "f64", "rain": "f64"},
              right={"days": [1, 2, 3, 1], "rain": [0.1, 0.2, 0.3, 0.4]}, right_dtypes={"days": "i32", "rain": "f64"},
              on="days",
              expect={"days": [1, 2, 1], "temp": [19.9, 7.0, 19.9], "rain": [0.1, 0.3, 0.1], "rain_right": [0.1, 0.2, 0.4]}))
C.append(dict(id="left_join_days", kind="join", how="left", source="crates/polars/tests/it/core/joins.rs:80-102",
              left={"days": [0, 1, 2, 3, 4], "temp": [22.1, 19.9, 7.0, 2.0, 3.0]}, left_dtypes={"days": "i32", "temp": "f64"},
              right={"days": [1, 2], "rain": [0.1, 0.2]}, right_dtypes={"days": "i32", "rain": "f64"},
              on="days",
              expect={"days": [0, 1, 2, 3, 4], "temp": [22.1, 19.9, 7.0, 2.0, 3.0], "rain": [None, 0.1, 0.2, None, None]}))
for dt in ("i8", "i16", "i32", "i64"):
    C.append(dict(id=f"join_negative_integers_{dt}", kind="join", how="inner", source="py-polars/tests/unit/operations/test_join.py:131-153",
                  left={"a": [-1, -6, -3, 0]}, left_dtypes={"a": dt},
                  right={"a": [-6, -1, -4, -2, 0], "b": [-6, -1, -4, -2, 0]}, right_dtypes={"a": dt, "b": dt},
                  on="a", expect={"a": [-6, -1, 0], "b": [-6, -1, 0]}))
C.append(dict(id="join_dup_keys_strings_as_codes", kind="join", how="inner", source="py-polars/tests/unit/operations/test_join.py:230-250",
              left={"a": ["a", "b", "a", "z"], "b": [1, 2, 3, 4], "c": [6, 5, 4, 3]}, left_dtypes={"a": "str", "b": "i64", "c": "i64"},
              right={"a": ["b", "c", "b", "a"], "k": [0, 3, 9, 6], "c": [1, 0, 2, 1]}, right_dtypes={"a": "str", "k": "i64", "c": "i64"},
              on="a", expect_column_sorted_by_key={"b": [1, 3, 2, 2]}))
C.append(dict(id="join_null_keys_never_match", kind="join", how="inner", source="py-polars/tests/unit/operations/test_join.py:1289-1310",
              left={"a": [None, 2, 1, 1, 5]}, left_dtypes={"a": "i64"},
              right={"a": [1, 1, None, 2], "b": [6, 7, 8, 9]}, right_dtypes={"a": "i64", "b": "i64"},
              on="a", expect={"a": [2, 1, 1, 1, 1], "b": [9, 6, 7, 6, 7]}))

# ---- comparison total order -------------------------------------------------------------------
C.append(dict(id="total_ordering_float", kind="cmp_total_order", source="py-polars/tests/unit/operations/test_comparison.py:209-226,343-371",
              values=[0.0, -0.0, -1.0, 1.0, "-nan", "nan", "-inf", "inf", None], dtypes=["f32", "f64"],
              rule="normal < nan, nan == nan, nulls propagate"))

# ---- filter sweep (generated from seeds exactly as the reference test does) ---------------------
C.append(dict(id="filter_sweep", kind="filter_sweep", source="py-polars/tests/unit/operations/test_filter.py:271-286",
              dtypes=["bool", "i8", "i16", "i32", "i64"], sizes=list(range(64)) + [100, 1000, 10000],
              selectivities=[0.0, 0.01, 0.1, 0.5, 0.9, 0.99, 1.000001],
              seed_rule="PCG64(size*100 + int(100*selectivity)); payload = uniform(size)*100 cast to dtype; mask = uniform(size) < selectivity; expect = payload[mask]"))

# ---- arithmetic -----------------------------------------------------------------------------------
C.append(dict(id="int_floor_div_mod_by_zero_is_null", kind="arith", source="crates/polars-compute/src/arithmetic/signed.rs:35-70; py-polars/tests/unit/operations/arithmetic/test_arithmetic.py:840",
              lhs=[7, -7, 0, 5, -9], rhs=[2, 2, 0, 0, -4], dtype="i64",
              expect={"floor_div": [3, -4, None, None, 2], "mod": [1, 1, None, None, -1], "add": [9, -5, 0, 5, -13], "mul": [14, -14, 0, 0, 36]}))
C.append(dict(id="int_wrapping", kind="arith", source="crates/polars-compute/src/arithmetic/signed.rs:12-33",
              lhs=[9223372036854775807, -9223372036854775808, 4611686018427387904], rhs=[1, -1, 2], dtype="i64",
              expect={"add": [-9223372036854775808, 9223372036854775807, 4611686018427387906], "mul": [9223372036854775807, -9223372036854775808, -9223372036854775808],
                      "sub": [9223372036854775806, -9223372036854775807, 4611686018427387902]}))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(out, "w") as f:
    json.dump(C, f, indent=1)
print(f"wrote {len(C)} cases -> {out}")
