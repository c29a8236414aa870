// kernels.hpp -- host-callable launchers of the hand-written gfx950 kernels.
// Every launcher enqueues on plx::stream() and returns immediately unless noted.
#pragma once
#include <string>
#include "core.hpp"
#include "kconfig.hpp"

namespace plx {
namespace k {

// launch geometry helpers -------------------------------------------------------
int grid_for(int64_t work_items, int items_per_block, int blocks_per_cu = 8);

// ---- elementwise (kernels_elementwise.hip) -------------------------------------
// compare -> LSB-first bitmap words (pad bits cleared); b == nullptr => scalar rhs
void cmp(int dtype, int op, const void* a, const void* b, plx_scalar s, int64_t n, uint64_t* out_bits);
// mode 0 col-col, 1 col-scalar, 2 scalar-col; out dtype = dtype, f64 for int TRUE_DIV
void arith(int dtype, int op, int mode, const void* a, const void* b, plx_scalar s, int64_t n, void* out);
// numeric cast; ok_bits (may be null) receives 1 where the value was representable
void cast(int from, int to, const void* in, int64_t n, void* out, uint64_t* ok_bits);
void cast_from_bool(const uint64_t* bits, int to, int64_t n, void* out);
// bitmap word ops; n_bits used to clear pad bits. b may be nullptr for op NOT (op = 3)
void bitmap_op(int op, const uint64_t* a, const uint64_t* b, int64_t n_bits, uint64_t* out);
// out = a & b & c with nullptr meaning all-ones; at least one non-null
void bitmap_and3(const uint64_t* a, const uint64_t* b, const uint64_t* c, int64_t n_bits, uint64_t* out);
int64_t bitmap_popcount(const uint64_t* a, int64_t n_bits);  // synchronises
// Kleene and (op 0) / or (op 1) of boolean columns (polars-compute/src/boolean.rs);
// validity pointers may be null (= all valid); out_valid may be null when both are null
void bool_kleene(int op, const uint64_t* lv, const uint64_t* lvalid, const uint64_t* rv, const uint64_t* rvalid, int64_t n_bits,
                 uint64_t* out_v, uint64_t* out_valid);
// OR n_bits of src (bit 0 aligned) into dst starting at bit dst_off; dst must be pre-zeroed there
void bitmap_blit(uint64_t* dst, int64_t dst_off, const uint64_t* src, int64_t n_bits);
// out[i] = pattern (width 1/2/4/8 bytes)
void fill(int width, void* out, uint64_t pattern, int64_t n);
void fill_null(int width /* 0: bitmap */, const void* in, const uint64_t* validity, uint64_t pattern, int64_t n, void* out);
void fill_iota_u32(uint32_t* out, int64_t n);
// result download of small frames: up to kPackMax device buffers are copied into one staging buffer by ONE launch
constexpr int kPackMax = 32;
struct PackBatch { const void* src[kPackMax]; uint32_t bytes[kPackMax]; uint32_t off[kPackMax]; int n; };
void pack_buffers(const PackBatch& b, void* staging);

// ---- synthetic data (kernels_datagen.hip; benchmark / test support) ------------------------------
void datagen_lineitem_q1(int64_t n, uint64_t seed, int64_t* shipdate, uint8_t* flag, uint8_t* status, int64_t* qty, double* price, double* disc, double* tax);
void datagen_orders(int64_t n, uint64_t seed, int64_t cust_hi, int64_t* okey, int64_t* cust, int64_t* odate, int64_t* prio, uint32_t* n_lines);
void datagen_lines(int64_t n_orders, uint64_t seed, const uint64_t* offsets, const int64_t* okey, const int64_t* odate, int64_t* lkey, double* price, double* disc, int64_t* ship);
void datagen_uniform(int dtype, int64_t n, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, double scale, void* out);
void datagen_zipf(int64_t n, uint64_t seed, uint32_t stream, uint64_t x0_q62, int64_t n_keys, int64_t* out);
void datagen_customer(int64_t n, uint64_t seed, int64_t* custkey, uint8_t* segment);

// ---- raw Utf8View / BinaryView keys (kernels_strview.hip) ---------------------------------------
// device-side dictionary encoding of 16-byte views (+ concatenated data buffers): u32 code per row (first-claim order), the number
// of distinct strings and their views ([n_distinct][2] u64; long strings carry their absolute offset into `data`).  Synchronises.
// stamps_valid (may be null; used when validity == null): the nulls of the column are STAMPED views (kStrviewNullLen, below) -- they become null keys, and the bitmap read off
// the stamps comes back in *stamps_valid with their number in *stamps_nulls (the encode's own pass over the views: no separate one).
void strview_dict_encode(const uint64_t* views, const uint64_t* validity, const uint8_t* data, const uint64_t* buf_base, int64_t n, Buf* out_codes, Buf* out_dict_views,
                         int64_t* n_distinct, Buf* stamps_valid = nullptr, int64_t* stamps_nulls = nullptr);
// dictionary -> offsets[n + 1] (u64) + contiguous bytes on the device
// Utf8 / LargeUtf8 arrays (offsets + bytes, already in HBM) -> 16-byte views: {len, 12 inline bytes} or {len, 4-byte prefix, buffer 0, offset
// data_base + start}; null rows (validity bit row0 + i clear) become all-zero views, or null stamps (below) with stamp_nulls.  *err (device u32) is set when an offset pair is not
// monotonic, leaves the data buffer or a long string starts beyond 4 GiB.
void strviews_from_offsets(const void* offsets, bool large, const uint8_t* data, uint64_t data_base, int64_t data_len, int64_t n, uint64_t* views_out, const uint64_t* validity, int64_t row0, bool stamp_nulls,
                           unsigned int* err);
// A null entry of a view column that travels WITHOUT a bitmap (plx_strview_groupby, plx_strview_dict_encode_device, plx_ipc_read_string_views) is a view whose length
// word is kStrviewNullLen -- no Arrow view has it (lengths are non-negative int32).  strview_stamp_nulls writes the stamps from a bitmap (in place);
// strview_dict_encode reads the bitmap back off the stamps in its own pass (stamps_valid).
constexpr uint32_t kStrviewNullLen = 0xffffffffu;
void strview_stamp_nulls(uint64_t* views, const uint64_t* validity, int64_t n);
// group_by(raw Utf8View key).agg(sum / count / len of one 8-byte numeric column) without a dictionary-encode pass (kernels_strgroup.hip); -1 = not on the fast path.
// A stamped (null) key is a key of its own; strview_null_group finds its group among the G result views (-> 0 or 1), clears its bit of `valid` and empties its view.
int64_t strview_groupby(const uint64_t* views, const uint64_t* values, const uint64_t* val_validity, int64_t n, bool is_f64, Buf* out_views, Buf* out_sum, Buf* out_cnt, Buf* out_len,
                        std::string* desc);
int64_t strview_null_group(uint64_t* gviews, int64_t G, uint64_t* valid);
void strdict_materialise(const uint64_t* dict_views, const uint8_t* data, int64_t n, Buf* out_offsets, Buf* out_bytes, uint64_t* total_bytes);
// synthetic Utf8View column (benchmark support): the inline view of "id%010d" % value for value = lo + floor(U * (hi - lo)) of row i
void datagen_id_views(int64_t n, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, uint64_t* out_views);
// 20-byte strings "id%010d-longkey": views {20, prefix, buffer 0, offset} into out_pool[(hi - lo) * 20], which holds every distinct string once
void datagen_long_id_views(int64_t n, uint64_t seed, uint32_t stream, int64_t lo, int64_t hi, uint64_t* out_views, uint8_t* out_pool);

// ---- reductions (kernels_reduce.hip) ------------------------------------------
struct ReduceResult {
  uint64_t isum;      // wrapping 64-bit sum of sign/zero-extended values (ints)
  double fsum;        // f64 sum (all types)
  uint64_t minmax_lo; // min bits (T widened to 64 bit / f64)
  uint64_t minmax_hi; // max bits
  uint64_t n_valid;   // valid (and mask-selected) rows
  uint64_t n_ordered; // valid rows that are not NaN (== n_valid for ints)
};
// one pass computing every whole-column aggregate; synchronises to return the result
ReduceResult reduce_all(int dtype, const void* values, const uint64_t* validity, int64_t n);

// ---- filter / gather (kernels_filter.hip) --------------------------------------
struct FilterPlan {
  int64_t n = 0;        // input rows
  int64_t n_out = 0;    // kept rows
  Buf tile_offsets;     // exclusive prefix of kept rows per 2048-row tile
  const uint64_t* mask = nullptr;
};
FilterPlan filter_prepare(const uint64_t* mask, int64_t n);  // synchronises (needs n_out)
// width 1/2/4/8 bytes; width 0 compacts a bitmap (`values` = bits). out_validity may be null.
void filter_apply(const FilterPlan& p, int width, const void* values, const uint64_t* validity, void* out_values,
                  uint64_t* out_validity);
// ---- filter -> frame (fused_sinks.hpp BallotSink -> compact_by_ballots) ----
// The selection a predicate scan leaves behind: per 128-row wave tile two ballots (rows of even / odd parity: lane l of the scan holds rows 2l, 2l + 1) and, after
// the scan over the tiles' counts, the number of kept rows before each tile.
struct Selection {
  int64_t n = 0, n_out = 0;
  Buf ballots;     // [n_wave_tiles][2] u64
  Buf offsets;     // [n_wave_tiles + 1] u64
};
constexpr int kCompactMaxCols = 12;
struct CompactCols {
  int n_cols;
  const void* in[kCompactMaxCols];   // plain fixed-width values
  void* out[kCompactMaxCols];        // each sized for Selection::n_out rows
  uint8_t width[kCompactMaxCols];    // 1, 2, 4, 8
  int n_w[4];                        // (filled by compact_by_ballots: columns per width class, widest first)
};
Selection selection_finish(Buf ballots, Buf counts, int64_t n);    // device scan of the counts; synchronises (n_out)
// the kept rows of up to kCompactMaxCols columns, in row order, in ONE pass over the inputs (+ their row indices when row_ids is given)
void compact_by_ballots(const Selection& sel, const CompactCols& cols, uint32_t* row_ids);
// the same selection as an LSB-first bitmap + 2048-row tile offsets: what filter_apply / filter_rowids work from (*mask_keep owns the bitmap)
FilterPlan selection_to_plan(const Selection& sel, Buf* mask_keep);
// kept row indices of a selection, ascending
void filter_rowids(const FilterPlan& p, uint32_t* out);
void gather(int width, const void* values, const uint64_t* validity, const uint32_t* idx, const uint64_t* idx_validity,
            int64_t n_idx, void* out, uint64_t* out_validity);
constexpr int kGatherMultiMax = 8;
// n_cols (<= kGatherMultiMax) columns of 4- or 8-byte values, no validity anywhere, gathered at the same indices in one launch
void gather_multi(int n_cols, const int* widths, const void* const* values, const uint32_t* idx, int64_t n_idx, void* const* out);

}  // namespace k
}  // namespace plx
