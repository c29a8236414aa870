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

# perfect-hash (categorical) keys with nulls, groups in first-appearance order: the reference collects the group members, pinned here by their count
_PH_VALUES = ["3", "41", "17", "5", "26", "27", "43", "45", "41", "13", "45", "48", "17", "22", "31", "25", "28", "13", "7", "26", "17", "4", "43", "47", "30", "28", "8", "27", "6", "7", "26", "11", "37", "29", "49", "20", "29", "28", "23", "9", None, "38", "19", "7", "38", "3", "30", "37", "41", "5", "16", "26", "31", "6", "25", "11", "17", "31", "31", "20", "26", None, "39", "10", "38", "4", "39", "15", "13", "35", "38", "11", "39", "11", "48", "36", "18", "11", "34", "16", "28", "9", "37", "8", "17", "48", "44", "28", "25", "30", "37", "30", "18", "12", None, "27", "10", "3", "16", "27", "6"]
_PH_GROUPS = ["3", "41", "17", "5", "26", "27", "43", "45", "13", "48", "22", "31", "25", "28", "7", "4", "47", "30", "8", "6", "11", "37", "29", "49", "20", "23", "9", None, "38", "19", "16", "39", "10", "15", "35", "36", "18", "34", "44", "12"]
_PH_COUNTS = [3, 3, 5, 2, 5, 4, 2, 2, 3, 3, 1, 4, 3, 5, 3, 2, 1, 4, 2, 3, 5, 4, 2, 1, 2, 1, 2, 3, 4, 1, 3, 3, 2, 1, 1, 1, 2, 1, 1, 1]
assert sum(_PH_COUNTS) == len(_PH_VALUES) and len(_PH_GROUPS) == len(_PH_COUNTS)
C.append(dict(id="perfect_hash_table_null_values", kind="groupby", source="py-polars/tests/unit/operations/test_group_by.py:948-1010",
              keys={"a": _PH_VALUES}, key_dtypes={"a": "str"}, values={"c": [1] * len(_PH_VALUES)}, value_dtypes={"c": "i64"},
              aggs=[["c", "len"]], maintain_order=True, expect={"a": _PH_GROUPS, "c_len": _PH_COUNTS},
              note="the reference aggregates the key column itself into lists; the list lengths (and the group order, null group included) are what is pinned"))

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

C.append(dict(id="left_join_dup_keys_strings_as_codes", kind="join", how="left", source="py-polars/tests/unit/operations/test_join.py:251-256",
              left={"a": ["a", "b", "a", "z"], "b": [1, 2, 3, 4], "c": [6, 5, 4, 3]}, left_dtypes={"a": "str", "b": "i64", "c": "i64"},
              right={"a": ["b", "c", "b", "a"], "k": [0, 3, 9, 6], "c": [1, 0, 2, 1]}, right_dtypes={"a": "str", "k": "i64", "c": "i64"},
              on="a", expect_column_sorted_by_key={"b": [1, 3, 2, 2, 4]}, expect_null_count={"c_right": 1}))
C.append(dict(id="left_join_keeps_every_left_row_nulls_included", kind="join", how="left", source="py-polars/tests/unit/operations/test_join.py:1316-1345",
              left={"a": [None, 2, 1, 1, 5]}, left_dtypes={"a": "i64"},
              right={"a": [1, None, 2, 6], "b": [6, 7, 8, 9]}, right_dtypes={"a": "i64", "b": "i64"},
              on="a", expect={"a": [None, 2, 1, 1, 5]}))
C.append(dict(id="left_join_three_keys_chunks_alignment_4720", kind="join", how="left", source="py-polars/tests/unit/operations/test_join.py:346-400",
              left={"index1": [0, 0, 1, 1], "index2": [10, 10, 11, 11], "index3": [100, 101, 100, 101]}, left_dtypes={"index1": "i64", "index2": "i64", "index3": "i64"},
              right={"index1": [0, 1], "index2": [10, 11], "index3": [100, 101]}, right_dtypes={"index1": "i64", "index2": "i64", "index3": "i64"},
              on=["index1", "index2", "index3"], expect={"index1": [0, 0, 1, 1], "index2": [10, 10, 11, 11], "index3": [100, 101, 100, 101]},
              note="the left side is the reference's df1 x df2 cross join, written out"))
_J4 = [None if a % 6 == 0 else a for a in range(138)]
C.append(dict(id="inner_join_four_keys_with_validity", kind="join", how="inner", source="py-polars/tests/unit/operations/test_join.py:980-1001",
              left={c: _J4 for c in "abcd"}, left_dtypes={c: "i64" for c in "abcd"}, right={c: _J4 for c in "abcd"}, right_dtypes={c: "i64" for c in "abcd"},
              on=["a", "b", "c", "d"], expect_rows=115, note="nulls_equal=False (the default): the 23 rows whose keys are null match nothing; 138 rows = two validity words and a remainder"))

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

# ---- sort / top-k -----------------------------------------------------------------
# frame columns + by / descending / nulls_last (+ limit = head after the sort); `expect` is the full output in order
# unless "unordered": true (the reference test uses check_row_order=False / check_order=False there).
TS = "py-polars/tests/unit/operations/test_sort.py"
TK = "py-polars/tests/unit/operations/test_top_k.py"
C.append(dict(id="sort_dates_multiples", kind="sort", source=TS + ":50-76", note="datetimes as their physical i64 (day index)",
              frame={"date": [0, 0, 1, 1, 2], "values": [5, 4, 3, 2, 1]}, dtypes={"date": "i64", "values": "i64"},
              by=["date", "values"], descending=[False, False], nulls_last=[False, False],
              expect={"values": [4, 5, 2, 3, 1]}))
for i, (nl, desc, ex, ey) in enumerate([
        ([False, True], [False, False], [None, None, 1, 3], [3, None, 2, 1]),
        ([True, False], [False, False], [1, 3, None, None], [2, 1, None, 3]),
        ([True, False], [True, True], [3, 1, None, None], [1, 2, None, 3]),
        ([False, True], [True, True], [None, None, 3, 1], [3, None, 1, 2]),
        ([False, True], [True, False], [None, None, 3, 1], [3, None, 1, 2])]):
    C.append(dict(id=f"sort_multi_nulls_last_{i}", kind="sort", source=TS + ":159-191",
                  frame={"x": [None, 1, None, 3], "y": [3, 2, None, 1]}, dtypes={"x": "i64", "y": "i64"},
                  by=["x", "y"], descending=desc, nulls_last=nl, expect={"x": ex, "y": ey}))
C.append(dict(id="sort_nans_3740", kind="sort", source=TS + ":301-310",
              frame={"key": [1, 2, 3, 4, 5], "val": [0.0, None, "nan", "-inf", "inf"]}, dtypes={"key": "i64", "val": "f64"},
              by=["val"], descending=[False], nulls_last=[False], expect={"key": [2, 4, 1, 5, 3]}))
C.append(dict(id="sort_args_nulls_first", kind="sort", source=TS + ":686-709",
              frame={"a": [1, 2, None], "b": [6.0, 5.0, 4.0]}, dtypes={"a": "i64", "b": "f64"},
              by=["a", "b"], descending=[False, False], nulls_last=[False, False], expect={"a": [None, 1, 2], "b": [4.0, 6.0, 5.0]}))
C.append(dict(id="sort_args_nulls_last", kind="sort", source=TS + ":714-716",
              frame={"a": [1, 2, None], "b": [6.0, 5.0, 4.0]}, dtypes={"a": "i64", "b": "f64"},
              by=["a"], descending=[False], nulls_last=[True], expect={"a": [1, 2, None], "b": [6.0, 5.0, 4.0]}))
C.append(dict(id="sort_descending", kind="sort", source=TS + ":803-808",
              frame={"a": [1, 2, 3], "b": [4, 5, 6]}, dtypes={"a": "i64", "b": "i64"},
              by=["a", "b"], descending=[True, True], nulls_last=[False, False], expect={"a": [3, 2, 1], "b": [6, 5, 4]}))
for desc, nl, eb, ef in [(False, False, [None, False, False, True, True], [3.0, 2.0, 5.0, 1.0, 4.0]),
                         (False, True, [False, False, True, True, None], [2.0, 5.0, 1.0, 4.0, 3.0]),
                         (True, True, [True, True, False, False, None], [1.0, 4.0, 2.0, 5.0, 3.0]),
                         (True, False, [None, True, True, False, False], [3.0, 1.0, 4.0, 2.0, 5.0])]:
    C.append(dict(id=f"sort_bool_with_null_12139_desc{int(desc)}_nl{int(nl)}", kind="sort", source=TS + ":925-961",
                  frame={"bool": [True, False, None, True, False], "float": [1.0, 2.0, 3.0, 4.0, 5.0]}, dtypes={"bool": "bool", "float": "f64"},
                  by=["bool"], descending=[desc], nulls_last=[nl], expect={"bool": eb, "float": ef}))
for desc in (True, False):
    for nl in (True, False):
        # the reference test builds its expectation with this rule (test_sort.py:1013-1018)
        sentinel = 100 if desc ^ nl else -100
        rx = sorted([1, 3, None, 2, None], key=lambda k: sentinel if k is None else k, reverse=desc)
        ry = sorted([1, 3, 0, 2, 0], key=lambda k: sentinel if k == 0 else k, reverse=desc)
        for by in (["x"], ["x", "y"]):
            C.append(dict(id=f"sort_descending_nulls_last_desc{int(desc)}_nl{int(nl)}_{len(by)}key", kind="sort", source=TS + ":1005-1027",
                          frame={"x": [1, 3, None, 2, None], "y": [1, 3, 0, 2, 0]}, dtypes={"x": "i64", "y": "i64"},
                          by=by, descending=[desc] * len(by), nulls_last=[nl] * len(by), expect={"x": rx, "y": ry}))
C.append(dict(id="sort_top_k_fast_path", kind="sort", source=TS + ":858-871",
              frame={"a": [1, 2, None], "b": [6.0, 5.0, 4.0]}, dtypes={"a": "i64", "b": "f64"},
              by=["b"], descending=[False], nulls_last=[False], limit=3, expect={"a": [None, 2, 1], "b": [4.0, 5.0, 6.0]}))
C.append(dict(id="sort_head_maintain_order", kind="sort", source=TK + ":609-616",
              frame={"x": [2, 0, 8, 0, 0, 0, 7, 0, 9, 0], "y": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]}, dtypes={"x": "i64", "y": "i64"},
              by=["x"], descending=[False], nulls_last=[False], limit=4, expect={"x": [0, 0, 0, 0], "y": [1, 3, 4, 5]}))
C.append(dict(id="top_k_9385_bool_sort_slice", kind="sort", source=TK + ":393-396",
              frame={"b": [True, False]}, dtypes={"b": "bool"}, by=["b"], descending=[False], nulls_last=[False], limit=1, expect={"b": [False]}))
C.append(dict(id="top_k_series", kind="top_k", source=TK + ":35-39", k=3, reverse=[False], bottom=False, unordered=True,
              frame={"a": [3, 8, 1, 5, 2]}, dtypes={"a": "i64"}, by=["a"], expect={"a": [8, 5, 3]}))
C.append(dict(id="bottom_k_series", kind="top_k", source=TK + ":35-40", k=4, reverse=[False], bottom=True, unordered=True,
              frame={"a": [3, 8, 1, 5, 2]}, dtypes={"a": "i64"}, by=["a"], expect={"a": [3, 2, 1, 5]}))
C.append(dict(id="top_k_more_than_rows", kind="top_k", source=TK + ":51-55", k=10, reverse=[False], bottom=False, unordered=True,
              frame={"test": [2, 4, 1, 3]}, dtypes={"test": "i64"}, by=["test"], expect={"test": [4, 3, 2, 1]}))
TKDF = dict(frame={"a": [1, 2, 3, 4, 2, 2, None], "b": [None, 2, 1, 4, 3, 2, None]}, dtypes={"a": "i64", "b": "i64"}, by=["a", "b"], unordered=True)
C.append(dict(id="top_k_df_two_keys", kind="top_k", source=TK + ":94-106", k=3, reverse=[False, False], bottom=False, expect={"a": [4, 3, 2], "b": [4, 1, 3]}, **TKDF))
C.append(dict(id="top_k_df_two_keys_reverse", kind="top_k", source=TK + ":108-112", k=3, reverse=[True, True], bottom=False, expect={"a": [1, 2, 2], "b": [None, 2, 2]}, **TKDF))
C.append(dict(id="bottom_k_df_two_keys_reverse", kind="top_k", source=TK + ":113-117", k=4, reverse=[True, True], bottom=True, expect={"a": [4, 3, 2, 2], "b": [4, 1, 3, 2]}, **TKDF))
C.append(dict(id="top_k_reverse", kind="top_k", source=TK + ":379-383", k=1, reverse=[True, True], bottom=False, unordered=True,
              frame={"a": [1, 2, 3], "b": [4, 5, 6]}, dtypes={"a": "i64", "b": "i64"}, by=["a", "b"], expect={"a": [1], "b": [4]}))

# ---- semi / anti joins: left rows kept, left order --------------------------------------------------------------
TJ = "py-polars/tests/unit/operations/test_join.py"
for how, ek, ep in (("anti", [1, 2], ["f", "i"]), ("semi", [3], [None])):
    C.append(dict(id=f"{how}_join_null_in_right", kind="semi_anti", how=how, source=TJ + ":29-41",
                  left={"key": [1, 2, 3], "payload": ["f", "i", None]}, left_dtypes={"key": "i64", "payload": "str"},
                  right={"key": [3, 4, 5, None]}, right_dtypes={"key": "i64"}, on="key", expect={"key": ek, "payload": ep}))
for how, ex in (("anti", [1]), ("semi", [0, 0])):
    C.append(dict(id=f"{how}_join_sorted_null", kind="semi_anti", how=how, source=TJ + ":676-688",
                  left={"x": [0, 0, 1]}, left_dtypes={"x": "i64"}, right={"x": [0, None], "y": [0, 1]}, right_dtypes={"x": "i64", "y": "i64"},
                  on="x", expect={"x": ex}))
for how, ea, ex in (("semi", [1, 9], [10, 90]), ("anti", [], [])):
    C.append(dict(id=f"{how}_join_28264", kind="semi_anti", how=how, source=TJ + ":4325-4350",
                  left={"a": [1, 9], "x": [10, 90]}, left_dtypes={"a": "i64", "x": "i64"}, right={"a": [1, 9], "y": [100, 900]}, right_dtypes={"a": "i64", "y": "i64"},
                  on="a", expect={"a": ea, "x": ex}))

for how, ea, eb, ep in (("anti", [1, 2, 1], ["a", "b", "a"], [10, 20, 40]), ("semi", [3], ["c"], [30])):
    C.append(dict(id=f"{how}_join_two_keys", kind="semi_anti", how=how, source=TJ + ":50-67",
                  left={"a": [1, 2, 3, 1], "b": ["a", "b", "c", "a"], "payload": [10, 20, 30, 40]}, left_dtypes={"a": "i64", "b": "str", "payload": "i64"},
                  right={"a": [3, 3, 4, 5], "b": ["c", "c", "d", "e"]}, right_dtypes={"a": "i64", "b": "str"},
                  on=["a", "b"], expect={"a": ea, "b": eb, "payload": ep}))

# ---- a3 arithmetic on primitive columns (round 4): kind "binary" = one operator, lhs / rhs = a column (list) or {"scalar": v}, nulls in / nulls out ----
TA = "py-polars/tests/unit/operations/arithmetic/test_arithmetic.py"
TL = "py-polars/tests/unit/lazyframe/test_lazyframe.py"
TD = "py-polars/tests/unit/dataframe/test_df.py"


def B(id, source, op, lhs, rhs, dtype, expect, expect_dtype=None, **kw):
    C.append(dict(id=id, kind="binary", source=source, op=op, lhs=lhs, rhs=rhs, dtype=dtype, expect=expect, expect_dtype=expect_dtype or dtype, **kw))


S = lambda v: {"scalar": v}
# test_arithmetic (lazyframe): a = [1, 2, 3] against integer literals on either side
for i, (op, lhs, rhs, exp) in enumerate([("mod", [1, 2, 3], S(2), [1, 0, 1]), ("mod", S(2), [1, 2, 3], [0, 0, 2]), ("floor_div", S(1), [1, 2, 3], [1, 0, 0]),
                                         ("mul", S(1), [1, 2, 3], [1, 2, 3]), ("add", S(1), [1, 2, 3], [2, 3, 4]), ("sub", S(1), [1, 2, 3], [0, -1, -2]),
                                         ("floor_div", [1, 2, 3], S(2), [0, 1, 1]), ("mul", [1, 2, 3], S(2), [2, 4, 6]), ("add", [1, 2, 3], S(2), [3, 4, 5]),
                                         ("sub", [1, 2, 3], S(2), [-1, 0, 1])]):
    B(f"lazy_arithmetic_{i + 1}", TL + ":910-941", op, lhs, rhs, "i64", exp)
# test_arithmetic_series: s = [1, 2] as Int64 and as Float64
for dt, one in (("i64", 1), ("f64", 1.0)):
    fl = dt == "f64"
    B(f"series_mul_{dt}", TA + ":446-456", "mul", [1, 2], [1, 2], dt, [1, 4])
    B(f"series_truediv_{dt}", TA + ":446-456", "true_div", [1, 2], [1, 2], dt, [1.0, 1.0], "f64")
    B(f"series_truediv_scalar_left_{dt}", TA + ":462", "true_div", S(one), [1, 2], dt, [1.0, 0.5], "f64")
    B(f"series_floordiv_scalar_left_{dt}", TA + ":463-464", "floor_div", S(one), [1, 2], dt, [1.0, 0.0] if fl else [1, 0])
    B(f"series_mod_scalar_left_{dt}", TA + ":466", "mod", S(one), [1, 2], dt, [0, 1])
    B(f"series_mod_scalar_right_{dt}", TA + ":467", "mod", [1, 2], S(one), dt, [0, 0])
    B(f"series_floordiv_scalar_right_{dt}", TA + ":457", "floor_div", [1, 2], S(2 * one), dt, [0, 1])
# test_df_series_division: true division of ints is Float64, floor division stays Int64
B("df_series_truediv_int", TA + ":425-443", "true_div", [2, 2, 10, 5, 6, 6], [2, 2, 2, 2, 2, 2], "i64", [1.0, 1.0, 5.0, 2.5, 3.0, 3.0], "f64")
B("df_series_floordiv_int", TA + ":425-443", "floor_div", [2, 2, 10, 5, 6, 6], [2, 2, 2, 2, 2, 2], "i64", [1, 1, 5, 2, 3, 3])
# test_arithmetic_on_df: Float64 columns against scalars
B("df_mul_scalar_f64", TA + ":365-371", "mul", [1.0, 2.0], S(2.0), "f64", [2.0, 4.0])
B("df_add_scalar_left_f64", TA + ":373-375", "add", S(2.0), [3.0, 4.0], "f64", [5.0, 6.0])
B("df_div_scalar_f64", TA + ":377-379", "true_div", [1.0, 2.0], S(2.0), "f64", [0.5, 1.0])
B("df_mod_scalar_f64", TA + ":385-387", "mod", [3.0, 4.0], S(2.0), "f64", [1.0, 0.0])
# test_arithmetic_null_count: a null on either side is a null result
B("null_propagation_col_col", TA + ":273-285", "add", [1, None, 2], [None, 2, 1], "i64", [None, None, 3])
B("null_propagation_scalar_left", TA + ":273-285", "add", S(1), [None, 2, 1], "i64", [None, 3, 2])
B("null_propagation_scalar_right", TA + ":273-285", "add", [1, None, 2], S(1), "i64", [2, None, 3])
# test_integer_divide_scalar_zero_lhs_19142: 0 // [1, 0] and 0 % [1, 0]
B("int_floordiv_scalar_zero_lhs", TA + ":840-842", "floor_div", S(0), [1, 0], "i64", [0, None])
B("int_mod_scalar_zero_lhs", TA + ":840-842", "mod", S(0), [1, 0], "i64", [0, None])
# test_floordiv_truediv (test_df.py): Python semantics for negative operands; x = [0, -1, -2, -3], y = [-0.0, -3.0, 5.0, -7.0], z = [10, 3, -5, 7]
for n in (3, -3):
    B(f"floordiv_python_semantics_int_by_{n}", TD + ":2972-2988", "floor_div", [0, -1, -2, -3, 10, 3, -5, 7], S(n), "i64", [v // n for v in (0, -1, -2, -3, 10, 3, -5, 7)])
    B(f"truediv_python_semantics_int_by_{n}", TD + ":2972-2988", "true_div", [0, -1, -2, -3, 10, 3, -5, 7], S(n), "i64", [v / n for v in (0, -1, -2, -3, 10, 3, -5, 7)], "f64")
    B(f"floordiv_python_semantics_f64_by_{n}", TD + ":2972-2988", "floor_div", [-0.0, -3.0, 5.0, -7.0], S(float(n)), "f64", [v // n for v in (-0.0, -3.0, 5.0, -7.0)])
B("floordiv_int_frame_frame", TD + ":2990-3000", "floor_div", [0, -1, -2, -3], [2, -2, 2, 3], "i64", [0, 0, -1, -1])
# test_int_operator_stability: small integer dtypes keep their dtype under + - * // (wrapping), / gives Float64
for dt in ("i8", "u8", "i16", "u16", "i32", "u32", "u64"):
    B(f"int_operator_stability_{dt}_add", TA + ":662-668", "add", [10], S(2), dt, [12])
    B(f"int_operator_stability_{dt}_truediv", TA + ":662-668", "true_div", [10], S(2), dt, [5.0], "f64")
B("float_floor_divide", TL + ":944-949", "floor_div", [10.4], S(0.5), "f64", [10.4 // 0.5])

# ---- a1 comparisons: kind "compare" -----------------------------------------------------------------------------------------------
TC = "py-polars/tests/unit/operations/test_comparison.py"
RC = "crates/polars-core/src/chunked_array/comparison/mod.rs"


def CMP(id, source, lhs, rhs, dtype, expect):
    C.append(dict(id=id, kind="compare", source=source, lhs=lhs, rhs=rhs, dtype=dtype, expect=expect))


CMP("comparison_expr_expr", TC + ":90-112", [1, 2, 3], [2, 1, 3], "i64",
    {"eq": [False, False, True], "ne": [True, True, False], "lt": [True, False, False], "le": [True, False, True], "gt": [False, True, False], "ge": [False, True, True]})
CMP("comparison_order_null_column", TC + ":22-43", [42, 42], [None, None], "i64", {op: [None, None] for op in ("lt", "le", "gt", "ge", "eq", "ne")})
CMP("comparison_nulls_single", TC + ":46-62", [None], [None], "i64", {"eq": [None], "ne": [None]})
CMP("null_handling_i32", RC + ":1214-1278", [1, None, 3], [1, 2, 3], "i32",
    {"eq": [True, None, True], "ne": [False, None, False], "gt": [False, None, False], "ge": [True, None, True], "lt": [False, None, False], "le": [True, None, True]})
# test_broadcasting_numeric: a = [1, 2, 3] against the unit-length columns [1] and [3], on either side
for name, rhs, exp in (("one", S(1), {"eq": [True, False, False], "ne": [False, True, True], "gt": [False, True, True], "lt": [False, False, False], "ge": [True, True, True], "le": [True, False, False]}),
                       ("three", S(3), {"eq": [False, False, True], "ne": [True, True, False], "gt": [False, False, False], "lt": [True, True, False], "ge": [False, False, True], "le": [True, True, True]})):
    CMP(f"broadcasting_numeric_{name}_right", RC + ":1412-1468", [1, 2, 3], rhs, "i32", exp)
CMP("broadcasting_numeric_one_left", RC + ":1412-1468", S(1), [1, 2, 3], "i32",
    {"eq": [True, False, False], "ne": [False, True, True], "gt": [False, False, False], "lt": [False, True, True], "ge": [True, False, False], "le": [True, True, True]})
CMP("broadcasting_numeric_three_left", RC + ":1412-1468", S(3), [1, 2, 3], "i32",
    {"eq": [False, False, True], "ne": [True, True, False], "gt": [True, True, False], "lt": [False, False, False], "ge": [True, True, True], "le": [False, False, True]})
# test_total_ordering_bool_series: false < true, null propagates
CMP("total_ordering_bool", TC + ":455-468", [None, None, None, False, False, False, True, True, True], [None, False, True, None, False, True, None, False, True], "bool",
    {"eq": [None, None, None, None, True, False, None, False, True], "ne": [None, None, None, None, False, True, None, True, False],
     "lt": [None, None, None, None, False, True, None, False, False], "le": [None, None, None, None, True, True, None, False, True],
     "gt": [None, None, None, None, False, False, None, True, False], "ge": [None, None, None, None, True, False, None, True, True]})

# ---- a2 Boolean logic (Kleene): kind "bool_logic" ---------------------------------------------------------------------------------
C.append(dict(id="bitwise_ops", kind="bool_logic", source=RC + ":1093-1103", lhs=[True, False, False], rhs=[True, True, None],
              expect={"or": [True, True, None], "and": [True, False, False], "not_rhs": [False, False, None]}))
C.append(dict(id="kleene_or_true", kind="bool_logic", source=RC + ":1305-1316", lhs=[True, False, None], rhs=[True, True, True], expect={"or": [True, True, True]}))
C.append(dict(id="kleene_or_false", kind="bool_logic", source=RC + ":1305-1316", lhs=[True, False, None], rhs=[False, False, False], expect={"or": [True, False, None]}))

# ---- a6 whole-column aggregates: kind "reduce" (more of them) -----------------------------------------------------------------------
TG = "py-polars/tests/unit/operations/aggregation/test_aggregations.py"
RA = "crates/polars-core/src/chunked_array/ops/aggregate/mod.rs"
C.append(dict(id="boolean_mean", kind="reduce", source=TG + ":42-54", values=[True, False, None, True], dtype="bool", op="mean", expect=0.6666666666666666))
C.append(dict(id="boolean_sum_is_index_type", kind="reduce", source=TA + ":244-254; " + TG + ":497-501", values=[True, False, True], dtype="bool", op="sum", expect=2, expect_dtype="u32"))
for dt in ("i16", "u16", "i8", "u8", "i32", "u32", "i64", "u64"):
    # the reference parametrises Int16 / UInt16 (test_int16_max_12904); the other widths are the same case carried over by this file, and say so
    src = TG + ":605-611" if dt in ("i16", "u16") else TG + ":605-611 (the reference parametrises Int16 / UInt16; this width is the same case carried over)"
    C.append(dict(id=f"int_min_max_skip_null_{dt}", kind="reduce", source=src, values=[None, 1], dtype=dt, op="min", expect=1))
    C.append(dict(id=f"int_max_skip_null_{dt}", kind="reduce", source=src, values=[None, 1], dtype=dt, op="max", expect=1))
IDS = [130352432, 130352277, 130352611, 130352833, 130352305, 130352258, 130352764, 130352475, 130352368, 130352346]
C.append(dict(id="min_2850", kind="reduce", source=TG + ":746-774", values=IDS, dtype="i64", op="min", expect=130352258))
C.append(dict(id="max_2850", kind="reduce", source=TG + ":746-774", values=IDS, dtype="i64", op="max", expect=130352833))
C.append(dict(id="sum_inf_not_nan", kind="reduce", source=TG + ":1333-1336", values=[10.0, None, 10.0, 10.0, 10.0, 10.0, "inf", 10.0, 10.0], dtype="f64", op="sum", expect="inf"))
C.append(dict(id="min_full_nan", kind="reduce", source="py-polars/tests/unit/series/test_series.py:2105-2107", values=["nan", "nan"], dtype="f64", op="min", expect="nan"))
C.append(dict(id="max_full_nan", kind="reduce", source="py-polars/tests/unit/series/test_series.py:2105-2107", values=["nan", "nan"], dtype="f64", op="max", expect="nan"))
for dt in ("f32", "f64"):
    C.append(dict(id=f"agg_float_min_ignores_nan_{dt}", kind="reduce", source=RA + ":755-764", values=[1.0, "nan"], dtype=dt, op="min", expect=1.0))
    C.append(dict(id=f"agg_float_min_ignores_nan_reversed_{dt}", kind="reduce", source=RA + ":755-764", values=["nan", 1.0], dtype=dt, op="min", expect=1.0))
C.append(dict(id="mean_f32_with_null", kind="reduce", source=RA + ":802-814", values=[1.0, 2.0, None], dtype="f32", op="mean", expect=1.5))
C.append(dict(id="mean_all_null_f32", kind="reduce", source=RA + ":815-825", values=[None, None, None], dtype="f32", op="mean", expect=None))
C.append(dict(id="count_all_null_column", kind="reduce", source=TG + ":737-743", values=[None, None, None, None, None, None], dtype="i64", op="count", expect=0))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(out, "w") as f:
    json.dump(C, f, indent=1)
print(f"wrote {len(C)} cases -> {out}")
