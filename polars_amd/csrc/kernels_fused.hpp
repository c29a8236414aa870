// kernels_fused.hpp -- launchers of the fused scan kernels (kernels_fused.hip) and
// the join kernels (kernels_join.hip).
#pragma once
#include <string>
#include <vector>

#include "core.hpp"
#include "fused.hpp"

namespace plx {
namespace k {

// filter -> exprs -> whole-frame aggregates held in registers.
// out_host[k] receives the 64-bit pattern of aggregate k
// (synchronises).  static_id = fused::find_static_shape(sh) or -1.
void fused_regagg(const fused::Shape& sh, const fused::Args& args, int static_id, uint64_t* out_host);
// filter -> exprs -> LDS-resident table for dense group ids in [0, n_groups); out_dev
// [n_groups][n_aggs] 64-bit patterns in HBM.  lds_agg_copies() == 0 => does not fit.
int lds_agg_copies(int n_groups, int n_aggs);
void fused_lds_agg(const fused::Shape& sh, const fused::Args& args, int n_groups, int static_id, uint64_t* out_dev, unsigned int* oob = nullptr /* device flag: a group id outside the table */);

// filter -> exprs -> atomics into an HBM table. Tables must be initialised with
// fill_u64(keys, cap + 2, kEmptyKey) / init_agg_cells(acc, slots, sh).
void fused_dense_agg(const fused::Shape& sh, const fused::Args& args, const fused::DenseTable& t, int static_id);
void fused_hash_agg(const fused::Shape& sh, const fused::Args& args, const fused::HashTable& t, int static_id);
void fused_wide_agg(const fused::Shape& sh, const fused::Args& args, const fused::WideTable& t);
// occupied slots of a wide table -> out_words[n_keys][G] (u64 each), out_kvalid[n_keys][G] (u8), out_acc[G][n_aggs];
// nullptr outputs = count only.  Returns the group count (synchronises).
int64_t wide_compact(const fused::WideTable& t, int n_keys, int n_aggs, int64_t out_stride, uint64_t* out_words, uint8_t* out_kvalid, uint64_t* out_acc);
// fused join->aggregate: build scan (key = sh.key, rows passing sh.pred) / probe scan (aggregates into t.acc)
void fused_join_build(const fused::Shape& sh, const fused::Args& args, const fused::JoinAggTable& t, int static_id);
void fused_probe_agg(const fused::Shape& sh, const fused::Args& args, const fused::JoinAggTable& t, int static_id);
// slots whose AGG_LEN cell (len_idx) is non-zero -> out_keys (u64), out_rows (u32 build row), out_acc; nullptr outputs = count only
int64_t join_agg_compact(const fused::JoinAggTable& t, int n_aggs, int len_idx, uint64_t* out_keys, uint32_t* out_rows, uint64_t* out_acc);
// multi-value join table (duplicate build keys; fused::JoinAggTable::links): representatives of the rows of every key's chain, and the compaction of row-indexed cells
void canonicalise_chains(const fused::JoinAggTable& t, const fused::RepCols& rc, unsigned int max_chain, unsigned int* flags);
int64_t rows_agg_compact(const uint64_t* acc, int64_t n_rows, int n_aggs, int len_idx, uint32_t* out_rows, uint64_t* out_acc);
// multi-value table whose cells are per KEY (slot): the groups are the representatives of every matched key's chain, each with the key's aggregate taken as many times as the
// group has build rows (sums / counts scaled, min / max / first as they are).  Allocates *out_rows (u32 build rows) and *out_acc ([G][n_aggs]); returns G.  Synchronises.
int64_t chains_agg_compact(const fused::JoinAggTable& t, const fused::Shape& sh, const uint64_t* acc, int len_idx, Buf* out_rows, Buf* out_acc);
// number of waves a fused scan over n_rows launches (sizes per-wave reservations)
int64_t scan_waves(int64_t n_rows);
// filter -> frame, first half: `sh` / `args` = the predicate program (pred only, no aggregates) -> ballots + kept-row count per 128-row wave tile
// (out.ballots [n_wave_tiles][2] u64, out.counts [n_wave_tiles] u32); kernels.hpp selection_finish / compact_by_ballots do the rest
void fused_ballots(const fused::Shape& sh, const fused::Args& args, const fused::BallotOut& out, int static_id);
// semi-join filter side -> membership bitmap (BitmapBuild)
void fused_bitmap_build(const fused::Shape& sh, const fused::Args& args, const fused::BitmapBuild& t, int static_id);
// direct-address variants (DirectJoinTable)
void fused_direct_build(const fused::Shape& sh, const fused::Args& args, const fused::DirectJoinTable& t, int static_id);
void fused_direct_probe_agg(const fused::Shape& sh, const fused::Args& args, const fused::DirectJoinTable& t, int static_id);
// probe rows that pass `sh`'s predicate and whose key's bit is set in `t`: ballots + counts per 128-row wave tile (kernels.hpp selection_finish / compact_by_ballots)
void fused_direct_hits(const fused::Shape& sh, const fused::Args& args, const fused::DirectJoinTable& t, const fused::BallotOut& out, int static_id);
// slot_row[rank of a build key in key order] = its build row (unique build keys; after direct_rank)
void direct_slot_rows(const fused::DirectJoinTable& t, int64_t n_used, uint32_t* slot_row);
// rank step (popcount per 512-bit block, exclusive scan over the blocks, u32 rank per word into rank_out[(range/512 + 1) * 8]);
// returns the number of set bits and, in *pairs_out, the pairs the build scan appended (n_pairs_dev: a zeroed device word); more
// pairs than bits = duplicate build keys (synchronises)
uint64_t direct_rank(const fused::DirectJoinTable& t, uint32_t* rank_out, int64_t n_used, uint64_t* n_pairs_dev, uint64_t* pairs_out);
// output step: one pass over the first n_used ordinals of the pair list: pairs whose slot has a non-zero AGG_LEN cell -> out_keys / out_rows / out_acc
int64_t direct_agg_compact(const fused::DirectJoinTable& t, int64_t n_used, int n_aggs, int len_idx, uint64_t* out_keys, uint32_t* out_rows, uint64_t* out_acc);
void fill_u64(uint64_t* p, int64_t n, uint64_t v);
void init_agg_cells(uint64_t* acc, int64_t n_slots, const fused::Shape& sh);
// Gather occupied table slots into dense arrays; returns the group count (synchronises).
// cap >= 0: hash table with cap regular slots + 2 special; cap < 0: dense table
// (out_keys = slot index, last slot = null group).  occ_agg: index of an AGG_LEN cell
// (group exists iff non-zero) or -1 to use the key array.
int64_t table_compact(const uint64_t* keys, const uint64_t* acc, int64_t n_slots, int64_t cap, int n_aggs, int occ_agg, uint64_t* out_keys,
                      uint8_t* out_key_valid, uint64_t* out_acc);

// aggregate cells [G][n_aggs] -> typed output column (+ validity bitmap, may be null)
void finalize_aggs(const uint64_t* acc, int n_aggs, int64_t G, const fused::FinalSpec& sp, void* out, uint64_t* out_valid);
// partitioned high-cardinality group-by (kernels_partition.hip): plan (false = does not apply) and run
// (returns the group count, -1 = an LDS table overflowed: use the HBM-table sink instead)
bool partition_plan(const fused::Shape& sh, double est_groups, bool any_nullable, fused::PartitionPlan* out);
int64_t partitioned_agg(const fused::Shape& sh, const fused::Args& args, const fused::PartitionPlan& pp, int static_id, Buf* out_keys, Buf* out_kvalid,
                        Buf* out_acc, std::string* desc);
// second generation (partition2_device.hpp): packed_bits > 0 = the key is a dense packed id of that many bits (direct-address
// LDS tables when they fit); hot_keys = heavy hitters pre-aggregated in the scatter pass (select_hot_keys on a sample table)
// value range of a record source (rec_layout2(...).src_slot[j]) when it is a plain integer column with cached statistics: lets the third
// generation scatter pack records (fused::kPackNarrow / kPackFused)
struct SrcRange { bool known = false; int64_t mn = 0, mx = 0; bool check = false; /* bounds nobody verified (the planner's sample): narrow with a per-row check */ };
bool partition_plan2(const fused::Shape& sh, double est_groups, int packed_bits, int len_idx, int64_t n_rows, int n_hot, fused::PartPlan2* out,
                     const SrcRange* src_ranges = nullptr /* [fused::kMaxSrc] */, const SrcRange* key_range = nullptr /* exact range of a single 64-bit key, if known */);
// hot_rows: sample rows the returned keys account for
void select_hot_keys(const fused::HashTable& t, int n_aggs, int len_idx, uint64_t threshold, std::vector<uint64_t>* out, uint64_t* hot_rows = nullptr);
// key_range_out (may be null): [2] receives the exact signed min / max of the valid keys the scatter pass streamed (hash mode
// only; min > max when it saw none) -- statistics gathered as a by-product
int64_t partitioned_agg2(const fused::Shape& sh, const fused::Args& args, const fused::PartPlan2& pp, int static_id, const std::vector<uint64_t>& hot_keys, Buf* out_keys,
                         Buf* out_kvalid, Buf* out_acc, std::string* desc, int64_t* key_range_out = nullptr, int64_t* wide_stride_out = nullptr);
// (wide key -- Shape::n_keys != 0: *out_keys = [n_keys][stride] key words, *out_kvalid = [n_keys][stride] valid flags, *wide_stride_out = stride)
// ---- partitioned join probe (probe keys in no particular order) ----------------------------------------------------------------
// A direct-address join table is a bitmap over the build key range: 75 MB for TPC-H SF100 orders.  Probe keys that arrive in key
// order walk it line by line out of the L2; probe keys in RANDOM order fetch one 128-B line from the Infinity Cache per row -- the
// fabric, not HBM, then bounds the probe (SF100 Q3 on shuffled inputs: 7.0 ms against 1.9 ms).  So unordered probe rows are first
// radix-partitioned by the high bits of (key - kmin) with the scatter of the partitioned group-by (records = key low bits + row id,
// predicate fused), and every partition is then probed by ONE workgroup against LDS only: its slice of the bitmap when that fits, else a
// Bloom filter built from the slice's set bits in the kernel's prologue; the CANDIDATE row ids (hits + < 1-2 % false positives) come back as
// one u32 column and the caller runs the ordinary probe kernel -- which tests the exact bitmap -- over just those rows (the reference's partitioned build/probe:
// crates/polars-ops/src/frame/join/hash_join/single_keys.rs:16-167, single_keys_inner.rs:11-149).
// `sh` / `args`: the probe side's predicate + key program with ONE aggregate AGG_FIRST_ROW (its row id is the record payload).
// Returns false when the geometry or the JIT is not available (the caller probes directly); *hits: PLX_U32 column of matching probe rows.
void touch_filter_set(const int64_t* keys, const uint64_t* validity, int64_t n, unsigned int log2_bits, uint64_t* filter);
bool partitioned_probe_hits(const fused::Shape& sh, const fused::Args& args, const fused::DirectJoinTable& dt, uint64_t n_build, int static_id, ColumnPtr* hits, std::string* desc);
bool partitioned_hash_probe_hits(const fused::Shape& sh, const fused::Args& args, const fused::JoinAggTable& ht, uint64_t n_build, int static_id, ColumnPtr* hits, std::string* desc);
// Build of a join table WITHOUT a device atomic per row: the rows that pass `sh`'s predicate (sh: the plain build's predicate + key program) become {key, row} pairs, the
// pairs are binned by the WINDOW of the table their hash falls into and every window is filled from an LDS image (t.log2_window set by the caller; t.flags / t.count
// zeroed; the table needs no memset).  The keys are numbered densely (fused::jt_cell): `cells`, when given, receives key and build row of every cell for
// cells_agg_compact.  false: not available for this shape / size -- nothing was touched, build the plain way.  Duplicate keys: t.flags[0]; a full window: t.flags[1].
struct JoinCells { Buf key, row; };
bool partitioned_join_build(const fused::Shape& sh, const fused::Args& args, int static_id, const fused::JoinAggTable& t, JoinCells* cells, std::string* desc);
// output step of the fused join -> aggregate over a windowed table: cells whose LEN aggregate is not 0 -> dense (key, build row, cells); n_cells = keys + 1
int64_t cells_agg_compact(const JoinCells& cells, const uint64_t* acc, int64_t n_cells, int n_aggs, int len_idx, uint64_t* out_keys, uint32_t* out_rows, uint64_t* out_acc);
// fraction of adjacent pairs (strided sample) of an integer column that are non-decreasing: 1.0 = sorted ascending
double sample_sortedness(const ColumnPtr& c);
// smallest / largest valid value among 65536 rows (64 evenly spaced runs); false: no valid value in the sample / not an integer column
bool sample_minmax(const ColumnPtr& c, int64_t* mn, int64_t* mx, int blocks = 64);
// all jobs of a batch (key decodes + aggregate finalisations) in one launch
void finalize_batch(const uint64_t* acc, int n_aggs, int64_t G, const fused::FinBatch& b);
// packed group keys -> one key column
void decode_key(const uint64_t* packed, const uint8_t* kvalid, int64_t G, const fused::KeyDecode& kd, void* out, uint64_t* out_valid);

}  // namespace k
}  // namespace plx
