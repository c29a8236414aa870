// join.hpp -- hash join on one key column pair (kernels_join.hip).
// Replaces polars-ops/src/frame/join/hash_join/{single_keys.rs:16-167 (build_tables),
// single_keys_inner.rs:11-149 (probe_inner / hash_join_tuples_inner),
// single_keys_left.rs:106-195, single_keys_dispatch.rs:234-357,476-553}.
#pragma once
#include <string>

#include "core.hpp"
#include "fused.hpp"

namespace plx {
namespace join {

// (left_idx, right_idx) as PLX_U32 columns; LEFT join: right_idx nullable.
void join_indices(int how, const ColumnPtr& left_key, const ColumnPtr& right_key, ColumnPtr& left_idx, ColumnPtr& right_idx, std::string* desc);

// Pairs of an inner / left join from a build table the fused build scan filled (fused::JoinAggTable: unique keys, or chains of rows per key) over a candidate
// list of probe rows (`cand`: PLX_U32, null = every row of probe_key).  Pair order = candidate order; a left join keeps every candidate (build_idx nullable).
void join_pairs(int how, const ColumnPtr& probe_key, const ColumnPtr& cand, const fused::JoinAggTable& t, ColumnPtr& probe_idx, ColumnPtr& build_idx, std::string* desc);

// the same against a direct-address build table (unique build keys over a dense range: bitmap + rank, fused::DirectJoinTable) and its slot -> build row map
void join_pairs_direct(int how, const ColumnPtr& probe_key, const ColumnPtr& cand, const fused::DirectJoinTable& dt, const uint32_t* slot_row, ColumnPtr& probe_idx, ColumnPtr& build_idx,
                       std::string* desc);

// HashPartitioner (polars-utils/src/hashing.rs:72-121): rows grouped by partition.
void hash_partition(const ColumnPtr& key, int n_partitions, uint64_t seed, ColumnPtr& perm, int64_t* counts_out);
// same, the per-partition row counts stay on the device ([n_partitions] u64): no host round trip (the exchange all-gathers them)
void hash_partition_dev(const ColumnPtr& key, int n_partitions, uint64_t seed, ColumnPtr& perm, Buf& counts_dev);

}  // namespace join
}  // namespace plx
