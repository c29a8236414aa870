// engine.hpp -- physical planner + executors of the MI355X backend.
//
// Mirrors the shape of the reference's in-memory engine
//   create_physical_plan      polars-mem-engine/src/planner/lp.rs:75-100,326-878
//   create_physical_expr      polars-expr/src/planner.rs:129-694
//   Executor::execute         polars-mem-engine/src/executors/executor.rs:10-16
// but plans FUSED pipelines wherever the IR allows it:
//   [Filter]* -> Select(aggregations)          => one fused scan kernel (register sink)
//   [Filter]* -> GroupBy(keys, aggregations)   => one fused scan kernel (LDS / dense / hash sink)
//   Join -> ...                                => hash join kernels (build / probe / emit)
// and otherwise falls back to one kernel per node (reference-shaped execution), which is
// also what PLX_PLAN_NO_FUSION forces.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "core.hpp"
#include "fused.hpp"

namespace plx {
namespace engine {

struct AE {
  int kind = 0, op = 0, lhs = -1, rhs = -1, dtype = 0, is_null = 0;
  plx_scalar lit{};
  std::string name;
};
struct IRN {
  int kind = 0, input = -1, input_right = -1, predicate = -1;
  plx_frame frame = 0;
  std::vector<int> exprs, keys, keys_right;
  int how = 0, maintain_order = 0;
  std::string suffix = "_right";
  std::vector<uint8_t> sort_descending, sort_nulls_last;
  int64_t slice_offset = 0, slice_len = 0;
};
struct Plan {
  std::vector<IRN> ir;
  std::vector<AE> ae;
  uint32_t flags = 0;
  std::string desc;  // physical plan description (which kernels / pipelines ran)
  std::map<int, FramePtr> memo;  // subtrees already executed for a fusion attempt that then fell back: the per-node path reuses them
  // An upper bound on the number of groups of the group-by about to run that the PLAN knows (0: none): the pair form of a join -> group-by whose keys are functions of the
  // build row has at most as many groups as build rows survive the build side's predicate.  It replaces the planner's sampled estimate -- the joined rows arrive clustered by
  // key hash, and a strided sample of clustered rows undercounts by orders of magnitude (2.5e6 groups estimated as 3e4: LDS tables overflow, the fall-back is the per-row
  // HBM table at a fiftieth of the speed).
  double group_hint = 0;
};

Plan import_plan(const plx_ir* ir, int n_ir, const plx_aexpr* ae, int n_ae, uint32_t flags);
FramePtr execute(Plan& plan, int root);

// dtype an expression evaluates to over `schema` (AExpr::to_field equivalent)
int infer_dtype(const Plan& plan, int e, const Frame& schema);
std::string output_name(const Plan& plan, int e);

// generic group-by used by plx_groupby_agg and the GroupBy executor (keys / values already columns)
void groupby_columns(const std::vector<ColumnPtr>& keys, const std::vector<ColumnPtr>& values, const std::vector<int>& aggs, bool maintain_order,
                     std::vector<ColumnPtr>& out_keys, std::vector<ColumnPtr>& out_aggs, std::string* desc);

// compile-only entry used by tests: lowers `Filter*(pred) -> Select/GroupBy` rooted at
// `root` and reports the fused shape (and whether an AOT specialisation matches)
// without touching the GPU.  Returns false if the plan is not fusable.
bool describe_fusion(Plan& plan, int root, fused::Shape* shape, int* static_id, std::string* why_not);
// The complete compiled form of a fusable `[Filter]* -> Select | GroupBy` pipeline as JSON (register program with its
// immediates, input columns, aggregate cells, key packing / decoding, finalisation of every output): what the GPU will
// execute, in a form the CPU tests interpret row by row (tests/program_eval.py).  Compile only.
bool dump_program_json(Plan& plan, int root, std::string* json, std::string* why_not);
// GroupBy directly over an inner Join: the three programs (count, build, probe) of the fused
// join->aggregate pipeline, or false + reason when the per-node path would run.
bool describe_join_fusion(Plan& plan, int root, std::vector<fused::Shape>* shapes, std::string* why_not);
// root = a Filter node: the predicate program of the one-pass filter -> frame kernel (k::fused_filter); false + why_not when it does not compile
bool describe_filter_fusion(Plan& plan, int root, fused::Shape* shape, std::string* why_not);

}  // namespace engine
}  // namespace plx
