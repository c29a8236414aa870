// sort.hpp -- stable multi-key arg-sort and top-k selection (kernels_sort.hip).
// Replaces, for primitive key columns,
//   arg_sort / arg_sort_multiple   polars-core/src/chunked_array/ops/sort/{mod.rs, arg_sort_multiple.rs, arg_sort.rs}
//   SortExec                       polars-mem-engine/src/executors/sort.rs
//   sort + slice -> top-k          polars-plan/src/plans/optimizer/slice_pushdown_lp.rs, polars-stream/src/nodes/top_k.rs
// Order (arg_sort.rs / total_ord.rs): nulls first unless nulls_last (an absolute position, not flipped by
// `descending`); floats in total order with NaN greatest and -0.0 == +0.0; ties keep input order.
#pragma once
#include <string>
#include <vector>

#include "core.hpp"

namespace plx {
namespace sort {

struct SortKey {
  ColumnPtr col;
  bool descending = false;
  bool nulls_last = false;
};

// PLX_U32 row indices of the first `limit` rows (limit < 0: all rows) of the stable sort by `keys`.
ColumnPtr sort_indices(const std::vector<SortKey>& keys, int64_t limit, std::string* desc);

}  // namespace sort
}  // namespace plx
