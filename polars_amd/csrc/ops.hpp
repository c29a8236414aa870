// ops.hpp -- column-level operators (one reference kernel family each): the
// validity/dtype logic around the raw kernels.  Used by the C ABI kernel-level entry
// points and by the materialising expression evaluator.
#pragma once
#include "core.hpp"

namespace plx {
namespace ops {

// comparisons/mod.rs:4-75 + arity.rs:203-214,430-454: out validity = AND of input validities
ColumnPtr cmp(int op, const ColumnPtr& lhs, const ColumnPtr& rhs);
ColumnPtr cmp_scalar(int op, const ColumnPtr& lhs, plx_scalar rhs, bool scalar_null = false);
// boolean.rs and/or (Kleene), xor; not keeps validity
ColumnPtr bool_binop(int op, const ColumnPtr& lhs, const ColumnPtr& rhs);
ColumnPtr bool_not(const ColumnPtr& c);
// arithmetic/mod.rs:8-150; integer floor-div / mod by zero -> null
ColumnPtr arith(int op, const ColumnPtr& lhs, const ColumnPtr& rhs);
ColumnPtr arith_scalar(int op, const ColumnPtr& col, plx_scalar s, bool scalar_on_left);
ColumnPtr cast(const ColumnPtr& c, int to);
// filter/mod.rs:18-28
ColumnPtr filter(const ColumnPtr& c, const ColumnPtr& mask);
struct PreparedMask;  // mask AND mask-validity + tile offsets, shared by all columns of a frame
std::shared_ptr<PreparedMask> prepare_mask(const ColumnPtr& mask);
int64_t prepared_rows(const PreparedMask& m);
ColumnPtr filter_prepared(const ColumnPtr& c, const PreparedMask& m);
// gather/primitive.rs:9-78
ColumnPtr gather(const ColumnPtr& c, const ColumnPtr& idx);
std::vector<ColumnPtr> gather_columns(const std::vector<ColumnPtr>& cols, const ColumnPtr& idx);   // one launch for plain 4- / 8-byte columns
// whole-column aggregate -> scalar (aggregate/mod.rs)
struct ScalarValue { plx_scalar v; int dtype; bool valid; };
ScalarValue reduce(int agg_op, const ColumnPtr& c);
ColumnPtr scalar_column(const ScalarValue& s);                 // length-1 column
ColumnPtr full_column(int dtype, plx_scalar v, bool valid, int64_t len);  // broadcast literal
ColumnPtr fill_null(const ColumnPtr& c, plx_scalar v);                     // valid ? value : literal (same dtype)
ColumnPtr concat(const std::vector<ColumnPtr>& chunks);
ColumnPtr slice_copy(const ColumnPtr& c, int64_t offset, int64_t len);

// min / max of an integer column over valid rows (cached on the column); false if no valid rows
// (allow_assumed: the caller checks every row against the bounds and can re-run -- the group-by planner; see Column::range_assumed)
bool int_range(const ColumnPtr& c, int64_t* mn, int64_t* mx, bool allow_assumed = false);

}  // namespace ops
void strview_encode_device(const uint64_t* views, const ColumnPtr& validity_holder, Buf data, int64_t n, plx_column* out_codes, uint64_t* out_dict);   // abi.cpp
void strview_encode_device_bases(const uint64_t* views, const ColumnPtr& validity_holder, Buf data, Buf buf_base, int64_t n, plx_column* out_codes, uint64_t* out_dict);   // abi.cpp
}  // namespace plx
