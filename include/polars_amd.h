/*
 * polars_amd.h -- C ABI of the MI355X-native columnar execution backend for the
 * Polars hot path (filter / gather, primitive compare + arithmetic, whole-column
 * and grouped aggregation, hash join), built for gfx950.
 *
 * This header is the drop-in boundary (seam B1).  Every entry
 * point is `extern "C"`, takes plain pointers / sizes / opaque integer handles and
 * returns an `int` status (0 = ok).  No torch / C++ types cross it.  A Rust
 * `extern "C"` block in polars-mem-engine binds these 1:1.
 *
 * Reference interfaces each group of entry points replaces (paths relative to
 * the pola-rs/polars tree):
 *
 *   plx_version / plx_last_error
 *       pyo3-polars/pyo3-polars/src/derive.rs:26-66
 *       (_polars_plugin_get_version, _polars_plugin_get_last_error_message)
 *       crates/polars-ffi/src/lib.rs:12-13 (MAJOR=0, MINOR=1)
 *   plx_series_export, plx_column_import_series / plx_column_export_series
 *       crates/polars-ffi/src/version_0.rs:7-16 (SeriesExport), :61-107
 *       (export_series / import_series)
 *   ArrowSchema / ArrowArray, plx_column_import_arrow / plx_column_export_arrow
 *       crates/polars-arrow/src/ffi/generated.rs:6-34 (Arrow C Data Interface)
 *   plx_cmp / plx_cmp_scalar
 *       crates/polars-compute/src/comparisons/mod.rs:4-75 (TotalEqKernel/TotalOrdKernel)
 *       null rule crates/polars-core/src/chunked_array/ops/arity.rs:203-214,430-454
 *   plx_bitmap_binop / plx_bitmap_not
 *       crates/polars-expr/src/expressions/binary.rs:110-118
 *   plx_arith / plx_arith_scalar
 *       crates/polars-compute/src/arithmetic/mod.rs:8-150 (ArithmeticKernel)
 *   plx_cast
 *       crates/polars-expr/src/expressions/cast.rs (numeric casts only)
 *   plx_filter
 *       crates/polars-compute/src/filter/mod.rs:18-28 (filter)
 *   plx_gather
 *       crates/polars-compute/src/gather/primitive.rs:9-78 (take_primitive_unchecked)
 *   plx_reduce
 *       crates/polars-core/src/chunked_array/ops/aggregate/mod.rs:86-137,240-246,307-316
 *   plx_groupby_agg
 *       crates/polars-core/src/frame/group_by/mod.rs:29-98 (group_by_with_series)
 *       + crates/polars-core/src/frame/group_by/aggregations/mod.rs:854-1018
 *   plx_join_indices
 *       crates/polars-ops/src/frame/join/hash_join/single_keys_dispatch.rs:234-357
 *       (hash_join_inner / hash_join_left -> (left_idx, right_idx))
 *   plx_hash_partition
 *       crates/polars-utils/src/hashing.rs:72-121 (HashPartitioner) and
 *       crates/polars-expr/src/hash_keys.rs:263-314 (gen_idxs_per_partition)
 *   plx_ir / plx_aexpr / plx_execute_plan
 *       crates/polars-plan/src/plans/ir/mod.rs:53-187 (IR arena),
 *       crates/polars-plan/src/plans/aexpr/mod.rs:150-259 (AExpr arena),
 *       crates/polars-mem-engine/src/planner/lp.rs:75-100,326-878 (create_physical_plan)
 *       crates/polars-mem-engine/src/executors/executor.rs:10-16 (Executor::execute)
 *   plx_profile_*
 *       crates/polars-expr/src/state/node_timer.rs:14-70 (NodeTimer)
 *
 * Threading: every call is thread-safe (the reference calls plugins from rayon workers,
 * crates/polars-ffi/src/version_0.rs:136-162).  A thread that wants its own HIP stream installs it
 * with plx_set_stream; device buffers recycled by the library's pool between threads are ordered
 * behind the releasing stream (event recorded at release, waited for on re-use).  A column handed
 * to another thread must only be used there after the producing thread synchronised
 * (plx_synchronize) -- the same rule a SeriesExport hand-over obeys.  Errors never unwind across
 * the ABI; a non-zero status means `plx_last_error()` (thread-local) holds the message.
 */
#ifndef POLARS_AMD_H
#define POLARS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLX_ABI_MAJOR 0
#define PLX_ABI_MINOR 1

/* ---- status codes ------------------------------------------------------ */
#define PLX_OK 0
#define PLX_ERR_INVALID 1      /* bad argument / dtype mismatch (PolarsError::InvalidOperation) */
#define PLX_ERR_HIP 2          /* HIP runtime error; message has hipGetErrorString */
#define PLX_ERR_UNSUPPORTED 3  /* node/dtype outside the hot path -> caller falls back to CPU */
#define PLX_ERR_OOM 4
#define PLX_ERR_SHAPE 5        /* length mismatch (PolarsError::ShapeMismatch) */
#define PLX_ERR_NOT_FOUND 6    /* unknown column (PolarsError::ColumnNotFound) */
#define PLX_ERR_CANCELLED 7    /* host cancel flag raised (ExecutionState::should_stop) */

/* ---- physical dtypes (Arrow primitive layouts) --------------------------- */
typedef enum plx_dtype {
  PLX_BOOL = 0, /* bit-packed LSB-first values bitmap */
  PLX_I8 = 1,
  PLX_I16 = 2,
  PLX_I32 = 3, /* also Date */
  PLX_I64 = 4, /* also Datetime / Duration */
  PLX_U8 = 5,
  PLX_U16 = 6,
  PLX_U32 = 7, /* also IdxSize, Categorical physical */
  PLX_U64 = 8,
  PLX_F32 = 9,
  PLX_F64 = 10
} plx_dtype;

/* comparison operators: polars_plan::dsl::Operator (dsl/expr/mod.rs:683-707) */
typedef enum plx_cmp_op { PLX_EQ = 0, PLX_NE = 1, PLX_LT = 2, PLX_LE = 3, PLX_GT = 4, PLX_GE = 5 } plx_cmp_op;

/* arithmetic operators */
typedef enum plx_arith_op {
  PLX_ADD = 0,
  PLX_SUB = 1,
  PLX_MUL = 2,
  PLX_TRUE_DIV = 3,  /* ints -> f64, floats stay */
  PLX_FLOOR_DIV = 4, /* ints: by zero -> null */
  PLX_MOD = 5        /* ints: by zero -> null */
} plx_arith_op;

typedef enum plx_bitmap_op { PLX_AND = 0, PLX_OR = 1, PLX_XOR = 2 } plx_bitmap_op;

/* aggregations: IRAggExpr (plans/aexpr/mod.rs) subset on the hot path */
typedef enum plx_agg_op {
  PLX_AGG_SUM = 0,
  PLX_AGG_MEAN = 1,
  PLX_AGG_MIN = 2,
  PLX_AGG_MAX = 3,
  PLX_AGG_COUNT = 4, /* non-null count, u32 (IdxSize) */
  PLX_AGG_LEN = 5,   /* row count incl. nulls, u32 */
  PLX_AGG_FIRST = 6  /* first row of each group (group_by only; used for key columns) */
} plx_agg_op;

typedef enum plx_join_how { PLX_JOIN_INNER = 0, PLX_JOIN_LEFT = 1, PLX_JOIN_SEMI = 2, PLX_JOIN_ANTI = 3 } plx_join_how;

/* A 64-bit scalar passed by bit pattern; interpreted according to a plx_dtype. */
typedef union plx_scalar {
  int64_t i;
  uint64_t u;
  double f64;
  float f32;
} plx_scalar;

/* Opaque handles (0 is never valid). */
typedef uint64_t plx_column; /* device-resident Arrow-layout column */
typedef uint64_t plx_frame;  /* ordered set of named columns of equal length */

/* ---- Arrow C Data Interface (standard layout) ------------------------- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

/* Mirror of polars_ffi::version_0::SeriesExport (a chunked Arrow array). */
typedef struct plx_series_export {
  struct ArrowSchema* field;
  struct ArrowArray** arrays;
  size_t len;
  void (*release)(struct plx_series_export*);
  void* private_data;
} plx_series_export;

/* ---- library / device -------------------------------------------------- */
/* (major << 16) | minor, same packing as _polars_plugin_get_version. */
uint32_t plx_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* plx_last_error(void);
/* Bind the calling process to one GPU (one process per GPU). Idempotent. */
int plx_init(int device_ordinal);
int plx_shutdown(void);
/* Launch all subsequent work of this thread on `hip_stream` (hipStream_t as void*;
 * NULL = the library's own stream). */
int plx_set_stream(void* hip_stream);
int plx_synchronize(void);
/* Cooperative cancel flag checked between kernel launches. */
int plx_set_cancel(int flag);
/* name (<=255 chars), CU count, HBM bytes of the bound device. */
int plx_device_info(char* name_out, size_t name_cap, int32_t* cu_count, uint64_t* hbm_bytes);
/* bytes currently held by the library's device pool / high-water mark */
int plx_memory_stats(uint64_t* in_use, uint64_t* high_water);
/* return cached free blocks to the driver */
int plx_memory_trim(void);
/* Pre-grows the device memory pool: after the call its cache holds a free block of at least `bytes`.  Mapping device memory costs ~30 ms per GB
 * (hipMalloc); an executor that expects queries with large intermediates (the record pool of a partitioned group-by over 1e9 rows: 12-25 GB) sizes the
 * pool once at start-up, the way the reference's GPU engine sizes its RMM pool, instead of paying inside its first query. */
int plx_memory_reserve(uint64_t bytes);

/* ---- columns ----------------------------------------------------------- */
/* Copy a host Arrow-layout buffer pair to HBM. `validity` may be NULL (no nulls);
 * `bit_offset` applies to both a PLX_BOOL values bitmap and the validity bitmap
 * (Arrow `offset`); for non-bool values the caller passes the already-offset
 * values pointer. */
int plx_column_from_host(plx_dtype dtype, const void* values, const uint8_t* validity,
                         int64_t bit_offset, int64_t len, plx_column* out);
/* Wrap device buffers owned by the caller (e.g. a torch tensor): no copy; the
 * caller keeps them alive until plx_column_free. validity may be NULL. */
int plx_column_from_device(plx_dtype dtype, void* dev_values, void* dev_validity, int64_t len,
                           plx_column* out);
/* Planning-only column: dtype / length / nullability (/ integer range statistics) but NO
 * device memory.  Lets plx_describe_fusion run without a GPU; every compute entry point
 * rejects it. */
int plx_column_placeholder(plx_dtype dtype, int64_t len, int nullable, int has_range, int64_t range_min, int64_t range_max,
                           plx_column* out);
/* Caller-provided value bounds of an integer column (dictionary size of a Categorical, Parquet column-chunk min / max
 * statistics): every valid value lies in [lo, hi].  The planner uses them the way the reference uses sortedness flags and
 * metadata statistics -- to choose dense / direct-address group tables without a pass over the data.  Kernels that rely on
 * the bounds check them per row and fail the query (PLX_ERR_INVALID) if a value lies outside. */
int plx_column_set_bounds(plx_column col, int64_t lo, int64_t hi);
/* Forgets what the library has LEARNED about the column (value range computed by a statistics pass or as a by-product of a scan, the group-by
 * planner's key sample and heavy hitters, the sampled sortedness): the next query pays for them again, as the first query on a fresh column
 * does.  Bounds declared with plx_column_set_bounds stay.  Measurement support (bench.py `one_shot_ms`); never changes a result. */
int plx_column_drop_statistics(plx_column col);
/* Arrow C Data Interface import: copies to HBM, then calls array->release and
 * schema->release (callee takes ownership: plugin.rs:122-125 convention). */
int plx_column_import_arrow(struct ArrowArray* array, struct ArrowSchema* schema, plx_column* out);
/* Chunked import (SeriesExport): chunks are concatenated on device. Takes ownership. */
int plx_column_import_series(plx_series_export* series, plx_column* out);
/* Export to host memory owned by the returned structs (released via ->release). */
int plx_column_export_arrow(plx_column col, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
int plx_column_export_series(plx_column col, const char* name, plx_series_export* out);
/* Plain download. values_out must hold len*width bytes (BOOL: (len+7)/8);
 * validity_out (may be NULL) must hold (len+7)/8 bytes; *has_validity_out tells
 * whether the column carries a validity bitmap (if not, validity_out is all ones). */
int plx_column_to_host(plx_column col, void* values_out, uint8_t* validity_out, int32_t* has_validity_out);
/* Device-to-device copy into caller-owned HBM (e.g. a torch tensor): values (len*width bytes;
 * BOOL: (len+7)/8) and, if dev_validity_out != NULL, (len+7)/8 validity bytes (all ones when the
 * column has no bitmap).  Synchronises the library stream before returning. */
int plx_column_copy_to_device(plx_column col, void* dev_values_out, void* dev_validity_out);
int plx_column_info(plx_column col, plx_dtype* dtype, int64_t* len, int64_t* null_count);
int plx_column_device_ptrs(plx_column col, void** values, void** validity);
int plx_column_retain(plx_column col);
int plx_column_free(plx_column col);

/* ---- kernel-level entry points (one reference kernel family each) -------- */
/* out: PLX_BOOL, validity = lhs.validity AND rhs.validity. Total-order float
 * semantics (NaN == NaN, NaN greatest). */
int plx_cmp(plx_cmp_op op, plx_column lhs, plx_column rhs, plx_column* out);
int plx_cmp_scalar(plx_cmp_op op, plx_column lhs, plx_scalar rhs, plx_column* out);
/* Boolean columns: values op, validity AND (binary.rs:110-118). */
int plx_bitmap_binop(plx_bitmap_op op, plx_column lhs, plx_column rhs, plx_column* out);
int plx_bitmap_not(plx_column col, plx_column* out);
/* Same-dtype operands (type coercion happens in the optimizer). */
int plx_arith(plx_arith_op op, plx_column lhs, plx_column rhs, plx_column* out);
/* scalar_on_left: computes scalar OP col instead of col OP scalar. */
int plx_arith_scalar(plx_arith_op op, plx_column col, plx_scalar scalar, int scalar_on_left, plx_column* out);
/* numeric -> numeric cast (non-strict: out-of-range -> null like polars cast(strict=False)). */
int plx_cast(plx_column col, plx_dtype to, plx_column* out);
/* Stream compaction by a PLX_BOOL mask; null mask values are false. */
int plx_filter(plx_column col, plx_column mask, plx_column* out);
/* out[i] = col[idx[i]]; idx is PLX_U32 (IdxSize); null idx -> null. */
int plx_gather(plx_column col, plx_column idx, plx_column* out);
/* Whole-column reduction. *out_value receives the scalar in the output dtype
 * *out_dtype (sum: SumCast rule; mean: f64, f32 stays f32; count/len: u32);
 * *out_valid = 0 when the result is null (mean/min/max of empty or all-null). */
int plx_reduce(plx_agg_op op, plx_column col, plx_scalar* out_value, plx_dtype* out_dtype, int32_t* out_valid);

/* Hash group-by with in-place aggregation.  n_keys >= 1 key columns (integer,
 * bool or float; a null key forms its own group; floats canonicalised).
 * aggs[i] applies to values[i] (ignored for PLX_AGG_LEN, may be 0).
 * Outputs: out_keys[n_keys] (one row per group) and out_aggs[n_aggs].
 * Row order of groups is unspecified unless maintain_order != 0 (then groups are
 * ordered by first occurrence). */
int plx_groupby_agg(const plx_column* keys, int32_t n_keys, const plx_column* values,
                    const plx_agg_op* aggs, int32_t n_aggs, int32_t maintain_order,
                    plx_column* out_keys, plx_column* out_aggs);

/* Equi-join on one key column pair -> row-index pairs (PLX_U32). For LEFT joins the
 * right index column carries nulls for unmatched left rows. Null keys never match.
 * Pair order is unspecified (sort before comparing), as in the reference.
 * SEMI / ANTI (single_keys_semi_anti.rs): *out_left_idx = the left rows with (without) a match on the right, in
 * left order; *out_right_idx = 0 (no column).  A null left key never matches (kept by ANTI, dropped by SEMI). */
int plx_join_indices(plx_join_how how, plx_column left_key, plx_column right_key,
                     plx_column* out_left_idx, plx_column* out_right_idx);

/* Stable multi-key arg-sort (arg_sort_multiple; polars-core/src/chunked_array/ops/sort/arg_sort_multiple.rs,
 * SortExec polars-mem-engine/src/executors/sort.rs).  descending / nulls_last: n_by flags each, NULL = all 0.
 * nulls_last is an absolute position (not flipped by descending); floats in total order (NaN greatest,
 * -0.0 == +0.0); ties keep input order.  limit >= 0 returns only the first `limit` indices of that order
 * (sort + slice -> top-k, slice_pushdown_lp.rs / polars-stream top_k.rs) without sorting the whole input.
 * *out_idx: PLX_U32 row indices. */
int plx_sort_indices(const plx_column* by, int32_t n_by, const uint8_t* descending, const uint8_t* nulls_last, int64_t limit,
                     plx_column* out_idx);

/* Key-hash partitioning for the multi-GPU exchange: partition p of row i is
 * mulhi(dirty_hash(key[i]) * seed', n_partitions) (HashPartitioner; nulls -> 0).
 * out_perm (PLX_U32, len rows) lists row indices grouped by partition;
 * counts_out[n_partitions] (host) receives rows per partition. */
int plx_hash_partition(plx_column key, int32_t n_partitions, uint64_t seed, plx_column* out_perm,
                       int64_t* counts_out);

/* ---- frames ------------------------------------------------------------- */
int plx_frame_new(const char* const* names, const plx_column* cols, int32_t n_cols, plx_frame* out);
int plx_frame_free(plx_frame f);
/* Vertical concatenation (polars.concat(how="vertical"), crates/polars-core/src/frame/mod.rs:609 vstack_mut): same column names and dtypes in
 * every frame; columns are copied device-to-device, validity bitmaps merged at bit granularity.  Used by multi-file scans. */
int plx_frame_concat(const plx_frame* frames, int32_t n_frames, plx_frame* out);
int plx_frame_shape(plx_frame f, int64_t* height, int32_t* width);
/* name_out points into library-owned storage valid until the frame is freed. */
int plx_frame_column(plx_frame f, int32_t i, const char** name_out, plx_column* col_out);
/* dtypes_out[width] (plx_dtype) of the frame's columns. */
int plx_frame_dtypes(plx_frame f, int32_t* dtypes_out);
/* Download every column with one stream synchronisation: values_out[i] / validity_out[i] as in
 * plx_column_to_host (validity_out[i] may be NULL), has_validity_out[width]. */
int plx_frame_to_host(plx_frame f, void* const* values_out, uint8_t* const* validity_out, int32_t* has_validity_out);

/* ---- plan execution (Executor seam) ------------------------------------ */
typedef enum plx_aexpr_kind {
  PLX_AE_COLUMN = 0,  /* name */
  PLX_AE_LITERAL = 1, /* dtype, lit, is_null */
  PLX_AE_BINARY = 2,  /* op = plx_operator, lhs, rhs */
  PLX_AE_CAST = 3,    /* lhs, dtype */
  PLX_AE_AGG = 4,     /* op = plx_agg_op, lhs */
  PLX_AE_LEN = 5,     /* pl.len() */
  PLX_AE_ALIAS = 6,   /* lhs, name */
  PLX_AE_NOT = 7,     /* lhs (boolean) */
  PLX_AE_IS_NULL = 8,     /* lhs -> Boolean, never null (FunctionExpr::Boolean(IsNull)) */
  PLX_AE_IS_NOT_NULL = 9, /* lhs -> Boolean, never null */
  PLX_AE_FILL_NULL = 10   /* lhs, rhs = non-null literal of the same dtype (fill_null(literal)); inside fused pipelines only */
} plx_aexpr_kind;

/* polars_plan::dsl::Operator subset */
typedef enum plx_operator {
  PLX_OP_EQ = 0, PLX_OP_NE = 1, PLX_OP_LT = 2, PLX_OP_LE = 3, PLX_OP_GT = 4, PLX_OP_GE = 5,
  PLX_OP_PLUS = 6, PLX_OP_MINUS = 7, PLX_OP_MULTIPLY = 8, PLX_OP_TRUE_DIVIDE = 9,
  PLX_OP_FLOOR_DIVIDE = 10, PLX_OP_MODULUS = 11, PLX_OP_AND = 12, PLX_OP_OR = 13, PLX_OP_XOR = 14
} plx_operator;

typedef struct plx_aexpr {
  int32_t kind; /* plx_aexpr_kind */
  int32_t op;   /* plx_operator or plx_agg_op */
  int32_t lhs;  /* arena index of the (left) input, -1 if none */
  int32_t rhs;  /* arena index of the right input, -1 if none */
  int32_t dtype; /* literal dtype / cast target (plx_dtype) */
  int32_t is_null; /* literal is NULL */
  plx_scalar lit;
  const char* name; /* column name / alias */
} plx_aexpr;

typedef enum plx_ir_kind {
  PLX_IR_SCAN = 0,    /* frame */
  PLX_IR_FILTER = 1,  /* input, predicate */
  PLX_IR_SELECT = 2,  /* input, exprs */
  PLX_IR_HSTACK = 3,  /* input, exprs (with_columns) */
  PLX_IR_GROUPBY = 4, /* input, keys, exprs (aggs), maintain_order */
  PLX_IR_JOIN = 5,    /* input, input_right, keys (left_on), keys_right (right_on), how, suffix */
  PLX_IR_SORT = 6,    /* input, keys (by), sort_descending[n_keys], sort_nulls_last[n_keys] (IR::Sort; always stable) */
  PLX_IR_SLICE = 7    /* input, slice_offset (negative: from the end), slice_len (IR::Slice); directly above a Sort it becomes top-k */
} plx_ir_kind;

typedef struct plx_ir {
  int32_t kind; /* plx_ir_kind */
  int32_t input;
  int32_t input_right;
  int32_t predicate;
  plx_frame frame;
  const int32_t* exprs;
  int32_t n_exprs;
  const int32_t* keys;
  int32_t n_keys;
  const int32_t* keys_right;
  int32_t n_keys_right;
  int32_t how; /* plx_join_how */
  int32_t maintain_order;
  const char* suffix; /* join suffix, NULL = "_right" */
  const uint8_t* sort_descending; /* PLX_IR_SORT: n_keys flags, NULL = ascending */
  const uint8_t* sort_nulls_last; /* PLX_IR_SORT: n_keys flags, NULL = nulls first */
  int64_t slice_offset;           /* PLX_IR_SLICE */
  int64_t slice_len;              /* PLX_IR_SLICE: rows kept (clamped to the input; slice_offsets, polars-core/src/utils/mod.rs:340-358) */
} plx_ir;

/* plan flags */
#define PLX_PLAN_NO_FUSION 1u /* force one kernel per node (reference-shaped execution) */
#define PLX_PLAN_NO_PARTITION 4u /* high-cardinality group-by: always the HBM-table sink, never the partitioned LDS path */
#define PLX_PLAN_NO_DIRECT_JOIN 2u /* fused join->aggregate: always use the hash table, never the direct-address table */

/* Build the physical plan for IR node `root` and execute it. The output frame is
 * owned by the caller (plx_frame_free). PLX_ERR_UNSUPPORTED means: run this
 * subtree on the CPU engine instead (same contract as docs/user-guide/gpu-support.md). */
int plx_execute_plan(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root,
                     uint32_t flags, plx_frame* out);
/* Compile-only: lowers the `[Filter]* -> Select | GroupBy` pipeline rooted at `root`
 * into its fused register program without launching anything (works on placeholder
 * columns, no GPU needed).  *fusable = 0 with a reason in why_not if the plan needs the
 * one-kernel-per-node path; *static_shape_id >= 0 if a pre-instantiated (AOT) kernel
 * matches.  plx_last_plan_description() then returns a dump of the program. */
int plx_describe_fusion(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root, int32_t* fusable,
                        int32_t* static_shape_id, char* why_not, size_t why_cap);
/* Compile-only, no GPU needed: the complete compiled form of the fusable pipeline rooted at `root` (same arguments as
 * plx_describe_fusion) as a JSON document in buf (NUL-terminated, truncated to cap): register program with immediates,
 * input columns, aggregate cells, key packing / decoding, finalisation of every output.  The CPU tests interpret it row by
 * row (tests/program_eval.py).  PLX_ERR_UNSUPPORTED with the reason in plx_last_error() if not fusable. */
int plx_debug_program_json(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root, char* buf, size_t cap);
/* Run-time kernel specialisation (hiprtc): query shapes without a pre-instantiated kernel get one compiled on
 * first use (inputs of >= PLX_JIT_MIN_ROWS rows, default 2^22; PLX_JIT=0 disables; failures fall back to the
 * generic interpreter).  plx_jit_selftest compiles (does not run: no GPU needed) the kernels of the fused pipeline
 * rooted at `root` -- same arguments as plx_describe_fusion -- and returns PLX_OK, or PLX_ERR_INVALID with the
 * compiler log in plx_last_error().  plx_jit_stats: kernels compiled so far / total compile milliseconds. */
int plx_jit_selftest(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root);
int plx_jit_stats(int32_t* compiled, double* compile_ms);
/* Inputs of at least min_rows rows use the JIT (default 2^22); min_rows < 0 disables it. */
int plx_jit_set_min_rows(int64_t min_rows);
/* Human-readable physical plan (which fused pipeline / kernels were chosen) of the
 * last plx_execute_plan on this thread. */
const char* plx_last_plan_description(void);

/* ---- synthetic benchmark data (no reference counterpart: dbgen lives outside the reference tree) -------
 * Fills n_rows of the TPC-H Q1 lineitem columns directly in HBM with a counter-based generator (row i is a pure
 * function of (seed, i): polars_amd/csrc/datagen_device.hpp), so bench.py gets its 25 GB SF100 input without
 * depending on another library's kernels.  out_cols[7], in this order: l_shipdate (PLX_I64, Datetime[us]),
 * l_returnflag (PLX_U8, codes 0..2 = A,N,R), l_linestatus (PLX_U8, 0..1 = F,O), l_quantity (PLX_I64, 1..50),
 * l_extendedprice (PLX_F64), l_discount (PLX_F64, 0..0.10), l_tax (PLX_F64, 0..0.08). */
int plx_datagen_lineitem_q1(int64_t n_rows, uint64_t seed, plx_column* out_cols);
/* The same generator evaluated on the host for rows [row0, row0 + n) (no GPU needed): pins the kernel's arithmetic
 * in the CPU tests and lets callers spot-check a device table.  Plain host arrays, any may be NULL. */
int plx_datagen_lineitem_q1_host(int64_t row0, int64_t n, uint64_t seed, int64_t* shipdate, uint8_t* returnflag, uint8_t* linestatus,
                                 int64_t* quantity, double* extendedprice, double* discount, double* tax);

/* TPC-H Q3 inputs in dbgen row order (both tables ascending in orderkey; sparse keys: 8 of every 32 used; 1-7 lines per
 * order; l_shipdate = o_orderdate + 1..121 days).  out_orders[4]: o_orderkey, o_custkey, o_orderdate (Datetime[us]),
 * o_shippriority (all PLX_I64); out_lineitem[4]: l_orderkey (I64), l_extendedprice (F64), l_discount (F64), l_shipdate (I64). */
int plx_datagen_orders_lineitem(int64_t n_orders, uint64_t seed, plx_column* out_orders, plx_column* out_lineitem);
/* Host twin for orders [order0, order0 + n) of a table of n_orders_total orders: per-order arrays (n entries, n_lines = lines
 * of each order) and the lines of those orders in order (line_cap = capacity of the line arrays; *n_lines_out = lines written,
 * PLX_ERR_INVALID if they do not fit).  Any output may be NULL. */
int plx_datagen_orders_lineitem_host(int64_t order0, int64_t n, int64_t n_orders_total, uint64_t seed, int64_t* orderkey, int64_t* custkey,
                                     int64_t* orderdate, uint32_t* n_lines, int64_t line_cap, int64_t* l_orderkey, double* l_extendedprice,
                                     double* l_discount, int64_t* l_shipdate, int64_t* n_lines_out);
/* One uniform column of n rows: value i = lo + floor(U_i * (hi - lo)), U_i from stream `stream` (0..7) of row i;
 * dtype PLX_I64 / PLX_U32: the integer; PLX_F64: the integer times `scale`.  plx_datagen_uniform_host: rows
 * [row0, row0 + n) into a host array of that dtype. */
/* customer (TPC-H Q3 columns): out_cols[2] = c_custkey (PLX_I64, 1..n in order), c_mktsegment (PLX_U8 codes 0..4 =
 * AUTOMOBILE, BUILDING, FURNITURE, HOUSEHOLD, MACHINERY); plx_datagen_customer_host: rows [row0, row0 + n) on the CPU. */
int plx_datagen_customer(int64_t n_customers, uint64_t seed, plx_column* out_cols);
int plx_datagen_customer_host(int64_t row0, int64_t n, uint64_t seed, int64_t* custkey, uint8_t* segment);
int plx_datagen_uniform(int32_t dtype, int64_t n_rows, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, double scale, plx_column* out);
int plx_datagen_uniform_host(int32_t dtype, int64_t row0, int64_t n, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, double scale, void* out);
/* One heavy-tailed PLX_I64 key column in [0, n_keys) (BASELINE config 3's "Zipf s = 1.1" variant): key i = floor(1 / x_i^10) - 1, x_i uniform in
 * [x0, 1) from stream `stream` of row i, x0 = n_keys^(-1/10) handed over as x0_q62 = round(x0 * 2^62); integer fixed-point arithmetic only, so
 * plx_datagen_zipf_host (rows [row0, row0 + n) on the CPU) is bit-identical. */
int plx_datagen_zipf(int64_t n_rows, uint64_t seed, uint32_t stream, uint64_t x0_q62, int64_t n_keys, plx_column* out);
int plx_datagen_zipf_host(int64_t row0, int64_t n, uint64_t seed, uint32_t stream, uint64_t x0_q62, int64_t n_keys, int64_t* out);

/* ---- raw Utf8View / BinaryView keys: device-side dictionary encoding -----------------------------
 * The reference hashes and compares the 16-byte views directly (crates/polars-expr/src/hash_keys.rs:413-452 BinviewKeys,
 * crates/polars-compute/src/binview_index_map.rs; view layout crates/polars-arrow/src/array/binview/view.rs:20-29,55:
 * {len u32, 12 inline bytes} or {len u32, prefix u32, buffer index u32, offset u32}).  plx_strview_dict_encode builds that
 * index map on the device once, when the column enters: `views` = n 16-byte views (host), `data_buffers` = the array's variadic
 * data buffers (host; may be empty when every string is <= 12 bytes).  out_codes = a PLX_U32 column of dictionary codes (validity
 * = the input's; bounds [0, n_distinct) declared), out_dict = the dictionary (code -> string), kept on the device.
 * plx_strview_dict_encode_device: the same for views already in HBM (a PLX_U64 column of 2 n words; `data` a PLX_U8 column or 0).
 * Nulls of a view column in HBM: the raw-view entry points (plx_strview_dict_encode_device, plx_strview_groupby, plx_ipc_read_string_views) carry no bitmap --
 * a null entry is a view whose length word is 0xFFFFFFFF (no Arrow view has it: lengths are non-negative int32; the other 12 bytes zero).
 * plx_strview_stamp_nulls(views, valid) writes those stamps into the column from the array's validity (valid: a PLX_BOOL column of n rows, true = valid -- the
 * BinaryViewArray's validity bitmap, crates/polars-arrow/src/array/binview/mod.rs, imported as Boolean values; a NULL entry of `valid` itself counts as not valid);
 * encode gives such rows null codes.  A view buffer the caller lent (plx_column_from_device) or that another column shares is copied first: the COLUMN behind the
 * handle is stamped, never memory somebody else can see.
 * plx_strdict_to_host: offsets[n_strings + 1] + the concatenated bytes, in code order. */
typedef uint64_t plx_strdict;
int plx_strview_dict_encode(const void* views, const uint8_t* validity, int64_t bit_offset, int64_t n, const void* const* data_buffers, const int64_t* data_sizes,
                            int32_t n_data_buffers, plx_column* out_codes, plx_strdict* out_dict);
int plx_strview_dict_encode_device(plx_column views_u64_pairs, plx_column data_u8, plx_column* out_codes, plx_strdict* out_dict);
int plx_strview_stamp_nulls(plx_column views_u64_pairs, plx_column valid_bool);
/* group_by(<raw Utf8View key>).agg(sum, count, len of ONE numeric column), the key never dictionary-encoded first (kernels_strgroup.hip; the reference's
 * BinviewKeys group-by: crates/polars-expr/src/hash_keys.rs:413-452, crates/polars-compute/src/binview_index_map.rs).  views: a PLX_U64 column of 2 n words in
 * HBM (inline strings: <= 12 bytes each); value: a PLX_F64 / PLX_I64 column of n rows, nulls allowed.  Outputs, one row per distinct string, in no particular
 * order: out_codes = 0 .. G-1 (PLX_U32) with *out_dict holding the G strings in that order; out_sum (the value's dtype; 0 for a group without a valid value),
 * out_count (valid values, PLX_U32), out_len (rows, PLX_U32) -- mean = sum / count.  Rows with a null key (stamped views, above) form one group of their own,
 * as in the reference (a null is a key: hash_keys.rs:413-452 keeps the validity in the key): that group's code is null, its dictionary entry empty.  Returns PLX_ERR_UNSUPPORTED when the input is outside the fast path (a
 * string longer than 12 bytes, more distinct strings than the LDS tables hold, fewer than ~4096 of them): the caller then encodes (plx_strview_dict_encode_device) and groups on the codes. */
int plx_strview_groupby(plx_column views_u64_pairs, plx_column value, plx_column* out_codes, plx_strdict* out_dict, plx_column* out_sum, plx_column* out_count, plx_column* out_len);
int plx_strdict_info(plx_strdict dict, int64_t* n_strings, int64_t* total_bytes);
int plx_strdict_to_host(plx_strdict dict, int64_t* offsets, uint8_t* bytes);
int plx_strdict_free(plx_strdict dict);
/* synthetic Utf8View column (benchmark support, BASELINE config 5 from raw strings): 2 n PLX_U64 words = the inline views of
 * "id%010d" % (lo + floor(U * (hi - lo))) with the same counter-based U as plx_datagen_uniform(stream) */
int plx_datagen_id_views(int64_t n_rows, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, plx_column* out_views);
/* the same for keys the view does NOT hold: 20-byte strings "id%010d-longkey"; out_views = 2 n PLX_U64 words {length 20 | 4-byte prefix, buffer 0 | offset}, out_data = the
 * (hi - lo) * 20 bytes of the distinct strings, each once, at (value - lo) * 20 -- rows with equal keys share their bytes, as after a gather of a string column
 * (the reference compares such keys through the buffers: crates/polars-compute/src/binview_index_map.rs:106-117 get_long_key) */
int plx_datagen_long_id_views(int64_t n_rows, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, plx_column* out_views, plx_column* out_data);

/* ---- Parquet scan -> device columns ------------------------------------
 * The scan in front of the hot path: the reference decodes Parquet on the CPU (crates/polars-parquet/src/parquet/read/page/reader.rs:183-300
 * page walk, parquet/read/compression.rs:70-135 decompress, arrow/read/deserialize/{primitive,boolean,dictionary_encoded,binview} decode;
 * driven by crates/polars-io/src/parquet/read/read_impl.rs and crates/polars-stream/src/nodes/io_sources/parquet).  Here the host reads
 * METADATA only (footer, page headers, string dictionary pages); the column-chunk bytes travel to HBM as stored -- compressed and
 * encoded, one DMA per chunk -- and are decoded by kernels: Snappy, RLE / bit-packed hybrid levels and dictionary indices, PLAIN
 * values, null expansion, integer narrowing.
 *   plx_parquet_open            footer + schema; works without a GPU (planning / row-group pruning on any machine)
 *   plx_parquet_column_info     leaf column -> name, plx_dtype (-1: outside the hot path's dtypes: nested, decimal, ...),
 *                               logical kind (0 none, 1 Date = days in PLX_I32, 2 / 5 / 6 Datetime[us / ms / ns] in PLX_I64 -- the stored
 *                               unit is kept; INT96 columns come out as ns like the reference's default --, 3 String / 4 Binary =
 *                               PLX_U32 dictionary codes + plx_parquet_categories), nullable.  `name` stays valid until the next
 *                               call on the same thread.
 *   plx_parquet_chunk_info      codec, bit mask of the page encodings, sizes, and the chunk statistics as scalars of the column's
 *                               dtype (has_min_max = 0 when absent or not usable, e.g. strings); null_count -1 = unknown
 *   plx_parquet_read            row_groups x columns -> frame (rows in row-group order as given).  Codecs: UNCOMPRESSED, SNAPPY and ZSTD (device
 *                               kernels), GZIP and LZ4_RAW (pages inflated by host threads with the library's own decoders, then the same kernels);
 *                               encodings: PLAIN (strings: see plx_parquet_column_strdict), PLAIN_DICTIONARY / RLE_DICTIONARY, RLE levels;
 *                               DELTA_BINARY_PACKED / BYTE_STREAM_SPLIT / DELTA_*_BYTE_ARRAY / INT96 (decoded by host threads);
 *                               data pages v1 and v2; anything else
 *                               is PLX_ERR_UNSUPPORTED naming what it met (the caller decodes that file on the host), a malformed
 *                               file is PLX_ERR_INVALID.  Needs plx_init: there is no host decode path in the library.
 *   plx_parquet_categories*     dictionary of a String / Binary column as of its last read: offsets[n + 1] + concatenated bytes,
 *                               index = code (first-appearance order over the chunk dictionaries read) */
typedef uint64_t plx_parquet;
int plx_parquet_open(const char* path, plx_parquet* out);
int plx_parquet_close(plx_parquet file);
int plx_parquet_shape(plx_parquet file, int64_t* num_rows, int32_t* num_row_groups, int32_t* num_columns);
int plx_parquet_column_info(plx_parquet file, int32_t column, const char** name, int32_t* dtype, int32_t* logical, int32_t* nullable);
/* time zone of a Datetime column: "UTC" when the file marks the timestamps as instants (isAdjustedToUTC; the reference then reads
 * Datetime(unit, "UTC"), crates/polars-parquet/src/arrow/read/schema/convert.rs), "" otherwise.  Valid until the next call on the thread. */
int plx_parquet_column_timezone(plx_parquet file, int32_t column, const char** timezone);
int plx_parquet_row_group_info(plx_parquet file, int32_t row_group, int64_t* num_rows, int64_t* compressed_bytes);
int plx_parquet_chunk_info(plx_parquet file, int32_t row_group, int32_t column, int32_t* codec, uint32_t* encodings, int64_t* compressed_bytes,
                           int64_t* uncompressed_bytes, int32_t* has_min_max, plx_scalar* min, plx_scalar* max, int64_t* null_count);
int plx_parquet_read(plx_parquet file, const int32_t* row_groups, int32_t n_row_groups, const int32_t* columns, int32_t n_columns, plx_frame* out);
int plx_parquet_categories(plx_parquet file, int32_t column, int64_t* n_strings, int64_t* total_bytes);
int plx_parquet_categories_to_host(plx_parquet file, int32_t column, int64_t* offsets, uint8_t* bytes);
/* String columns holding PLAIN (not dictionary-encoded) pages: host threads assemble 16-byte views over the page payloads, the dictionary is
 * built on the device (plx_strview_dict_encode); this hands its handle over (caller frees it with plx_strdict_free).  PLX_ERR_NOT_FOUND
 * when the column's last read went through the chunk dictionaries (plx_parquet_categories*). */
int plx_parquet_column_strdict(plx_parquet file, int32_t column, plx_strdict* out);

/* ---- Arrow IPC file (Feather V2) scan -> device columns -------------------------
 * The reference reads IPC files with crates/polars-arrow/src/io/ipc/read/{file.rs,common.rs,read_basic.rs,schema.rs} (driven by
 * crates/polars-io/src/ipc/ipc_file.rs and crates/polars-stream/src/nodes/io_sources/ipc.rs): FlatBuffers footer + one message per
 * record batch whose body holds the column buffers in Arrow layout.  An uncompressed file needs no decoding at all: the host parses
 * the metadata, every selected buffer goes file -> page-locked staging -> HBM in one DMA, record batches are concatenated in place.
 *   plx_ipc_open / _shape / _column_info / _batch_info   metadata; work without a GPU.  dtype / logical as for plx_parquet_column_info.
 *   plx_ipc_read                batches x columns -> frame.  LZ4_FRAME / ZSTD compressed bodies are inflated per buffer by host threads
 *                               (the library's own decoders); nested columns, non-us timestamps -> PLX_ERR_UNSUPPORTED naming what it met.
 *   strings                     dictionary-encoded in the file: indices become PLX_U32 codes (widened on the device), values through
 *                               plx_ipc_categories*.  Utf8 / LargeUtf8 / Utf8View columns: views are assembled on the host, the
 *                               dictionary is built on the device (plx_strview_dict_encode); plx_ipc_column_strdict hands its handle
 *                               over (caller frees it with plx_strdict_free). */
typedef uint64_t plx_ipc;
int plx_ipc_open(const char* path, plx_ipc* out);
int plx_ipc_close(plx_ipc file);
int plx_ipc_shape(plx_ipc file, int64_t* num_rows, int32_t* num_batches, int32_t* num_columns);
int plx_ipc_column_info(plx_ipc file, int32_t column, const char** name, int32_t* dtype, int32_t* logical, int32_t* nullable);
int plx_ipc_column_timezone(plx_ipc file, int32_t column, const char** timezone);      /* the Timestamp type's timezone string, "" = none */
int plx_ipc_batch_info(plx_ipc file, int32_t batch, int64_t* num_rows, int64_t* body_bytes, int32_t* compressed);
int plx_ipc_read(plx_ipc file, const int32_t* batches, int32_t n_batches, const int32_t* columns, int32_t n_columns, plx_frame* out);
/* One Utf8 / LargeUtf8 / Binary column of the selected record batches as it is needed for a group-by ON THE VIEWS (plx_strview_groupby): *out_views = a PLX_U64 column of
 * 2 n words (16-byte views built on the device from the file's offsets + bytes), *out_data = the bytes behind the views of strings over 12 bytes (PLX_U8; the pair is what
 * plx_strview_dict_encode_device takes when the column has to be encoded after all); null entries are stamped views.  PLX_ERR_UNSUPPORTED: a Utf8View column, a column that is
 * dictionary-encoded in the file -- read those through plx_ipc_read.  (The reference reads string columns as views and hashes them per operator:
 * crates/polars-io/src/ipc/ipc_file.rs, crates/polars-expr/src/hash_keys.rs:413-452.) */
int plx_ipc_read_string_views(plx_ipc file, const int32_t* batches, int32_t n_batches, int32_t column, plx_column* out_views, plx_column* out_data);
int plx_ipc_categories(plx_ipc file, int32_t column, int64_t* n_strings, int64_t* total_bytes);
int plx_ipc_categories_to_host(plx_ipc file, int32_t column, int64_t* offsets, uint8_t* bytes);
int plx_ipc_column_strdict(plx_ipc file, int32_t column, plx_strdict* out);

/* ---- multi-GPU exchange (one process per GPU, RCCL over xGMI) -----------------------------
 * The exchange step of the sharded operators; shape of the reference's in-process exchange:
 * crates/polars-utils/src/hashing.rs:72-121 (HashPartitioner), crates/polars-stream/src/nodes/group_by.rs:252-497
 * (combine_locals).  A communicator wraps an RCCL communicator created from a 128-byte unique id that rank 0 obtains
 * (plx_comm_unique_id) and hands to the other ranks by any means (torch.distributed / MPI / a file).  librccl is
 * loaded with dlopen on first use.
 *   plx_exchange_by_key  every row of `frame` goes to rank plx_hash_partition(key) -- rows with a NULL key to rank 0
 *                        (null_partition(), hashing.rs:111-115); all columns travel in ONE grouped ncclSend / ncclRecv
 *                        all-to-all(v) on the library's stream (nullable and Boolean columns included: validity bitmaps and
 *                        bit-packed values cross as one byte per row and are re-packed on receipt); ONE host round trip
 *                        (the [world x world] row counts, for the receive allocations) and ONE stream synchronisation at the end
 *                        (the gathered per-destination buffers return to the pool only after RCCL is done with them);
 *                        *out = the rows this rank received.  rows_sent / bytes_sent: what left this rank over the fabric.
 *                        The sharded group-by calls it on PARTIAL aggregate rows (polars_amd/dist.py sharded_groupby:
 *                        local pre-aggregation first, group_by.rs:140-497), the sharded join on rows.
 *   plx_allgather_frame  concatenation of every rank's (small) frame in rank order, on every rank (same column kinds).  */
typedef uint64_t plx_comm;
int plx_comm_unique_id(uint8_t* out_128_bytes);
int plx_comm_init(const uint8_t* unique_id_128_bytes, int32_t rank, int32_t world_size, plx_comm* out);
int plx_comm_info(plx_comm comm, int32_t* rank, int32_t* world_size);
int plx_comm_free(plx_comm comm);
int plx_exchange_by_key(plx_comm comm, plx_frame frame, const char* key, uint64_t seed, plx_frame* out, uint64_t* rows_sent, uint64_t* bytes_sent);
int plx_allgather_frame(plx_comm comm, plx_frame frame, plx_frame* out);

/* ---- tracing (NodeTimer equivalent) --------------------------------------- */
typedef struct plx_profile_record {
  char name[48];      /* kernel / node name */
  double start_us;    /* hipEvent time relative to plx_profile_enable */
  double end_us;
  uint64_t algo_bytes; /* algorithmic bytes of this launch (0 if n/a) */
  uint64_t rows;
} plx_profile_record;
int plx_profile_enable(int on);
/* Resolves pending events (synchronises), copies up to cap records, returns count in *n. */
int plx_profile_fetch(plx_profile_record* out, int32_t cap, int32_t* n);
int plx_profile_clear(void);

/* ---- the reference's expression-plugin entry points ------------------------------------
 * What an UNMODIFIED Polars dlopens (crates/polars-plan/src/plans/aexpr/function_expr/plugin.rs:23-137
 * call_plugin, :139-227 plugin_field; normally generated by pyo3-polars-derive/src/lib.rs:140-163):
 *   _polars_plugin_get_version()                 (major << 16) | minor, here (0, 1)   [pyo3-polars derive.rs:56-66]
 *   _polars_plugin_get_last_error_message()      thread-local C string of the last failure   [derive.rs:26-45]
 *   _polars_plugin_<name>(inputs, n, kwargs, kwargs_len, out, ctx)
 *       inputs: n SeriesExport values (crates/polars-ffi/src/version_0.rs:7-16) the CALLEE takes ownership of and
 *       releases (plugin.rs:122-125); out: written on success, out->private_data == NULL on failure; kwargs: pickled
 *       dict (only the key "op" is read); ctx: CallerContext (version_0.rs:136-162), bit 0 = the caller is already
 *       parallel -> the call runs on a HIP stream private to the calling thread.  A length-1 input is a broadcast literal.
 *   _polars_plugin_field_<name>(fields, n, out, kwargs, kwargs_len)   output field (minor 1 signature, plugin.rs:186-207)
 * Functions: plx_cmp / plx_arith (kwargs {"op": "gt" | ... | "add" | ...}) and one symbol per operator for callers
 * without kwargs: plx_eq ne lt le gt ge, plx_add sub mul truediv floordiv mod, plx_filter(values, mask),
 * plx_sum mean min max (length-1 result).  Python side: polars.plugins.register_plugin_function(
 *     plugin_path=".../libpolars_amd.so", function_name="plx_gt", args=[pl.col("a"), pl.lit(3)]).           */
typedef struct plx_caller_context { uint64_t bitflags; } plx_caller_context;
uint32_t _polars_plugin_get_version(void);
char* _polars_plugin_get_last_error_message(void);
#define PLX_DECLARE_PLUGIN(name)                                                                                            \
  void _polars_plugin_##name(const plx_series_export* inputs, size_t n_inputs, const uint8_t* kwargs, size_t kwargs_len,    \
                             plx_series_export* out, const void* ctx);                                                      \
  void _polars_plugin_field_##name(const struct ArrowSchema* fields, size_t n_fields, struct ArrowSchema* out,              \
                                   const uint8_t* kwargs, size_t kwargs_len);
PLX_DECLARE_PLUGIN(plx_cmp) PLX_DECLARE_PLUGIN(plx_arith) PLX_DECLARE_PLUGIN(plx_filter)
PLX_DECLARE_PLUGIN(plx_eq) PLX_DECLARE_PLUGIN(plx_ne) PLX_DECLARE_PLUGIN(plx_lt) PLX_DECLARE_PLUGIN(plx_le) PLX_DECLARE_PLUGIN(plx_gt) PLX_DECLARE_PLUGIN(plx_ge)
PLX_DECLARE_PLUGIN(plx_add) PLX_DECLARE_PLUGIN(plx_sub) PLX_DECLARE_PLUGIN(plx_mul) PLX_DECLARE_PLUGIN(plx_truediv) PLX_DECLARE_PLUGIN(plx_floordiv) PLX_DECLARE_PLUGIN(plx_mod)
PLX_DECLARE_PLUGIN(plx_sum) PLX_DECLARE_PLUGIN(plx_mean) PLX_DECLARE_PLUGIN(plx_min) PLX_DECLARE_PLUGIN(plx_max)

#ifdef __cplusplus
}
#endif
#endif /* POLARS_AMD_H */
