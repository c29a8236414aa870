// ops.cpp -- column-level operators: dtype checks, validity combination and output
// allocation around the raw gfx950 kernels.
#include "ops.hpp"

#include <cmath>

#include "kernels.hpp"

namespace plx {

int64_t column_null_count(const ColumnPtr& c) {
  if (!c->validity) return 0;
  if (c->null_count < 0) c->null_count = c->len - k::bitmap_popcount(c->valid_words(), c->len);
  return c->null_count;
}

namespace ops {

static void require_same_len(const ColumnPtr& a, const ColumnPtr& b, const char* what) {
  PLX_REQUIRE(a->len == b->len, PLX_ERR_SHAPE, std::string(what) + ": length mismatch " + std::to_string(a->len) + " vs " + std::to_string(b->len));
}
static void require_same_dtype(const ColumnPtr& a, const ColumnPtr& b, const char* what) {
  PLX_REQUIRE(a->dtype == b->dtype, PLX_ERR_INVALID,
              std::string(what) + ": dtype mismatch " + dtype_name(a->dtype) + " vs " + dtype_name(b->dtype) + " (type coercion inserts casts upstream)");
}
// out validity = a AND b (either may be absent)
static Buf and_validity(const Buf& a, const Buf& b, int64_t n) {
  if (!a) return b;
  if (!b) return a;
  Buf out = dev_alloc_zero(bitmap_bytes(n));
  k::bitmap_op(0, a->as<uint64_t>(), b->as<uint64_t>(), n, out->as<uint64_t>());
  return out;
}
static Buf all_null_validity(int64_t n) { return dev_alloc_zero(bitmap_bytes(n)); }

ColumnPtr cmp(int op, const ColumnPtr& lhs, const ColumnPtr& rhs) {
  require_same_len(lhs, rhs, "cmp");
  require_same_dtype(lhs, rhs, "cmp");
  PLX_REQUIRE(op >= PLX_EQ && op <= PLX_GE, PLX_ERR_INVALID, "cmp: bad operator");
  auto out = std::make_shared<Column>();
  out->dtype = PLX_BOOL; out->len = lhs->len;
  out->values = dev_alloc_zero(bitmap_bytes(lhs->len));
  if (lhs->dtype == PLX_BOOL) {
    // Boolean columns compare as bitmaps, false < true (crates/polars-compute/src/comparisons/boolean.rs:9-70): == is !(l ^ r), != is l ^ r,
    // < is !l & r, <= is !l | r, > and >= the same with the operands swapped
    const uint64_t* a = lhs->values->as<uint64_t>();
    const uint64_t* b = rhs->values->as<uint64_t>();
    uint64_t* o = out->values->as<uint64_t>();
    if (op == PLX_EQ || op == PLX_NE) {
      k::bitmap_op(2, a, b, lhs->len, o);
      if (op == PLX_EQ) k::bitmap_op(3, o, nullptr, lhs->len, o);
    } else {
      const bool swap = op == PLX_GT || op == PLX_GE;
      k::bitmap_op(3, swap ? b : a, nullptr, lhs->len, o);                                  // !l
      k::bitmap_op((op == PLX_LT || op == PLX_GT) ? 0 : 1, o, swap ? a : b, lhs->len, o);   // & r  /  | r
    }
  } else {
    plx_scalar z; z.u = 0;
    k::cmp(lhs->dtype, op, lhs->data(), rhs->data(), z, lhs->len, out->values->as<uint64_t>());
  }
  out->validity = and_validity(lhs->validity, rhs->validity, lhs->len);
  if (!out->validity) out->null_count = 0;
  return out;
}

ColumnPtr cmp_scalar(int op, const ColumnPtr& lhs, plx_scalar rhs, bool scalar_null) {
  PLX_REQUIRE(op >= PLX_EQ && op <= PLX_GE, PLX_ERR_INVALID, "cmp: bad operator");
  PLX_REQUIRE(lhs->dtype != PLX_BOOL, PLX_ERR_UNSUPPORTED, "cmp_scalar on boolean column");
  auto out = std::make_shared<Column>();
  out->dtype = PLX_BOOL; out->len = lhs->len;
  out->values = dev_alloc_zero(bitmap_bytes(lhs->len));
  if (scalar_null) { out->validity = all_null_validity(lhs->len); out->null_count = lhs->len; return out; }
  k::cmp(lhs->dtype, op, lhs->data(), nullptr, rhs, lhs->len, out->values->as<uint64_t>());
  out->validity = lhs->validity;
  out->null_count = lhs->null_count;
  return out;
}

ColumnPtr bool_binop(int op, const ColumnPtr& lhs, const ColumnPtr& rhs) {
  require_same_len(lhs, rhs, "bool_binop");
  PLX_REQUIRE(lhs->dtype == PLX_BOOL && rhs->dtype == PLX_BOOL, PLX_ERR_INVALID, "bitand/bitor/xor need boolean operands");
  auto out = std::make_shared<Column>();
  out->dtype = PLX_BOOL; out->len = lhs->len;
  out->values = dev_alloc_zero(bitmap_bytes(lhs->len));
  const bool any_valid = lhs->validity || rhs->validity;
  if (op == PLX_XOR) {
    k::bitmap_op(2, lhs->values->as<uint64_t>(), rhs->values->as<uint64_t>(), lhs->len, out->values->as<uint64_t>());
    out->validity = and_validity(lhs->validity, rhs->validity, lhs->len);
  } else {
    PLX_REQUIRE(op == PLX_AND || op == PLX_OR, PLX_ERR_INVALID, "bool_binop: bad operator");
    if (any_valid) out->validity = dev_alloc_zero(bitmap_bytes(lhs->len));
    k::bool_kleene(op == PLX_AND ? 0 : 1, lhs->values->as<uint64_t>(), lhs->valid_words(), rhs->values->as<uint64_t>(), rhs->valid_words(), lhs->len,
                   out->values->as<uint64_t>(), any_valid ? out->validity->as<uint64_t>() : nullptr);
  }
  if (!out->validity) out->null_count = 0;
  return out;
}

ColumnPtr bool_not(const ColumnPtr& c) {
  PLX_REQUIRE(c->dtype == PLX_BOOL, PLX_ERR_INVALID, "not: boolean operand required");
  auto out = std::make_shared<Column>();
  out->dtype = PLX_BOOL; out->len = c->len;
  out->values = dev_alloc_zero(bitmap_bytes(c->len));
  k::bitmap_op(3, c->values->as<uint64_t>(), nullptr, c->len, out->values->as<uint64_t>());
  out->validity = c->validity; out->null_count = c->null_count;
  return out;
}

static int arith_out_dtype(int op, int dt) { return (op == PLX_TRUE_DIV && !dtype_is_float(dt)) ? PLX_F64 : dt; }

ColumnPtr arith(int op, const ColumnPtr& lhs, const ColumnPtr& rhs) {
  require_same_len(lhs, rhs, "arith");
  require_same_dtype(lhs, rhs, "arith");
  PLX_REQUIRE(lhs->dtype != PLX_BOOL, PLX_ERR_UNSUPPORTED, "arithmetic on boolean columns");
  PLX_REQUIRE(op >= PLX_ADD && op <= PLX_MOD, PLX_ERR_INVALID, "arith: bad operator");
  auto out = std::make_shared<Column>();
  out->dtype = arith_out_dtype(op, lhs->dtype); out->len = lhs->len;
  out->values = dev_alloc(values_bytes(out->dtype, lhs->len));
  plx_scalar z; z.u = 0;
  k::arith(lhs->dtype, op, 0, lhs->data(), rhs->data(), z, lhs->len, out->values->ptr);
  out->validity = and_validity(lhs->validity, rhs->validity, lhs->len);
  if ((op == PLX_FLOOR_DIV || op == PLX_MOD) && dtype_is_int(lhs->dtype)) {
    // signed.rs:35-70: rhs == 0 -> null
    Buf nz = dev_alloc_zero(bitmap_bytes(lhs->len));
    k::cmp(rhs->dtype, PLX_NE, rhs->data(), nullptr, z, rhs->len, nz->as<uint64_t>());
    out->validity = and_validity(out->validity, nz, lhs->len);
  }
  if (!out->validity) out->null_count = 0;
  return out;
}

static bool scalar_is_zero(int dt, plx_scalar s) {
  switch (dtype_width(dt)) {
    case 1: return (s.u & 0xff) == 0;
    case 2: return (s.u & 0xffff) == 0;
    case 4: return dt == PLX_F32 ? s.f32 == 0.0f : (s.u & 0xffffffffu) == 0;
    default: return dt == PLX_F64 ? s.f64 == 0.0 : s.u == 0;
  }
}

ColumnPtr arith_scalar(int op, const ColumnPtr& col, plx_scalar s, bool scalar_on_left) {
  PLX_REQUIRE(col->dtype != PLX_BOOL, PLX_ERR_UNSUPPORTED, "arithmetic on boolean columns");
  PLX_REQUIRE(op >= PLX_ADD && op <= PLX_MOD, PLX_ERR_INVALID, "arith: bad operator");
  auto out = std::make_shared<Column>();
  out->dtype = arith_out_dtype(op, col->dtype); out->len = col->len;
  out->values = dev_alloc(values_bytes(out->dtype, col->len));
  out->validity = col->validity; out->null_count = col->null_count;
  const bool int_divmod = (op == PLX_FLOOR_DIV || op == PLX_MOD) && dtype_is_int(col->dtype);
  if (int_divmod && !scalar_on_left && scalar_is_zero(col->dtype, s)) {
    // signed.rs:104-106: x // 0 -> full null
    PLX_HIP(hipMemsetAsync(out->values->ptr, 0, values_bytes(out->dtype, col->len), stream()));
    out->validity = all_null_validity(col->len); out->null_count = col->len;
    return out;
  }
  k::arith(col->dtype, op, scalar_on_left ? 2 : 1, col->data(), nullptr, s, col->len, out->values->ptr);
  if (int_divmod && scalar_on_left) {
    // signed.rs:141-151: s // x with x == 0 -> null
    plx_scalar z; z.u = 0;
    Buf nz = dev_alloc_zero(bitmap_bytes(col->len));
    k::cmp(col->dtype, PLX_NE, col->data(), nullptr, z, col->len, nz->as<uint64_t>());
    out->validity = and_validity(col->validity, nz, col->len);
    out->null_count = -1;
  }
  return out;
}

ColumnPtr cast(const ColumnPtr& c, int to) {
  if (c->dtype == to) return c;
  PLX_REQUIRE(to >= PLX_I8 && to <= PLX_F64, PLX_ERR_UNSUPPORTED, "cast: unsupported target dtype");
  auto out = std::make_shared<Column>();
  out->dtype = to; out->len = c->len;
  out->values = dev_alloc(values_bytes(to, c->len));
  if (c->dtype == PLX_BOOL) {
    k::cast_from_bool(c->values->as<uint64_t>(), to, c->len, out->values->ptr);
    out->validity = c->validity; out->null_count = c->null_count;
    return out;
  }
  // can the cast fail (value not representable)?  float->int and narrowing / sign-changing int casts
  const bool lossy = !dtype_is_float(to) && (dtype_is_float(c->dtype) || dtype_width(to) < dtype_width(c->dtype) ||
                                              (dtype_is_signed(c->dtype) != dtype_is_signed(to) && !(dtype_is_unsigned(c->dtype) && dtype_width(to) > dtype_width(c->dtype))));
  Buf ok;
  if (lossy) ok = dev_alloc_zero(bitmap_bytes(c->len));
  k::cast(c->dtype, to, c->data(), c->len, out->values->ptr, ok ? ok->as<uint64_t>() : nullptr);
  out->validity = and_validity(c->validity, ok, c->len);
  out->null_count = lossy ? -1 : c->null_count;
  if (!out->validity) out->null_count = 0;
  return out;
}

// ------------------------------------------------------------------- filter ---
struct PreparedMask {
  Buf bits;  // mask values AND mask validity
  k::FilterPlan plan;
};
int64_t prepared_rows(const PreparedMask& m) { return m.plan.n_out; }

std::shared_ptr<PreparedMask> prepare_mask(const ColumnPtr& mask) {
  PLX_REQUIRE(mask->dtype == PLX_BOOL, PLX_ERR_INVALID, "filter: predicate must be boolean");
  auto pm = std::make_shared<PreparedMask>();
  if (mask->validity) {  // filter/mod.rs:21-27: null -> false
    pm->bits = dev_alloc_zero(bitmap_bytes(mask->len));
    k::bitmap_op(0, mask->values->as<uint64_t>(), mask->validity->as<uint64_t>(), mask->len, pm->bits->as<uint64_t>());
  } else pm->bits = mask->values;
  pm->plan = k::filter_prepare(pm->bits->as<uint64_t>(), mask->len);
  return pm;
}

ColumnPtr filter_prepared(const ColumnPtr& c, const PreparedMask& m) {
  PLX_REQUIRE(c->len == m.plan.n, PLX_ERR_SHAPE, "filter: mask length " + std::to_string(m.plan.n) + " != column length " + std::to_string(c->len));
  const int64_t n_out = m.plan.n_out;
  if (n_out == c->len) return c;  // filter/mod.rs:47-49 all-true fast path
  auto out = std::make_shared<Column>();
  out->dtype = c->dtype; out->len = n_out;
  out->values = c->dtype == PLX_BOOL ? dev_alloc_zero(bitmap_bytes(n_out)) : dev_alloc(values_bytes(c->dtype, n_out));
  if (c->validity) out->validity = dev_alloc_zero(bitmap_bytes(n_out)); else out->null_count = 0;
  k::filter_apply(m.plan, dtype_width(c->dtype), c->data(), c->valid_words(), out->values->ptr, out->validity ? out->validity->as<uint64_t>() : nullptr);
  return out;
}

ColumnPtr filter(const ColumnPtr& c, const ColumnPtr& mask) {
  require_same_len(c, mask, "filter");
  auto pm = prepare_mask(mask);
  return filter_prepared(c, *pm);
}

ColumnPtr gather(const ColumnPtr& c, const ColumnPtr& idx) {
  PLX_REQUIRE(idx->dtype == PLX_U32, PLX_ERR_INVALID, "gather: indices must be u32 (IdxSize)");
  auto out = std::make_shared<Column>();
  out->dtype = c->dtype; out->len = idx->len;
  out->values = c->dtype == PLX_BOOL ? dev_alloc_zero(bitmap_bytes(idx->len)) : dev_alloc(values_bytes(c->dtype, idx->len));
  const bool need_valid = c->validity || idx->validity;
  if (need_valid) out->validity = dev_alloc_zero(bitmap_bytes(idx->len)); else out->null_count = 0;
  k::gather(dtype_width(c->dtype), c->data(), c->valid_words(), idx->values->as<uint32_t>(), idx->valid_words(), idx->len, out->values->ptr,
            need_valid ? out->validity->as<uint64_t>() : nullptr);
  return out;
}

// several columns at the same (valid) indices: one launch when every column is a plain 4- or 8-byte one, column by column otherwise
std::vector<ColumnPtr> gather_columns(const std::vector<ColumnPtr>& cols, const ColumnPtr& idx) {
  PLX_REQUIRE(idx->dtype == PLX_U32, PLX_ERR_INVALID, "gather: indices must be u32 (IdxSize)");
  bool plain = !idx->validity && cols.size() >= 2 && cols.size() <= (size_t)k::kGatherMultiMax;
  for (const ColumnPtr& c : cols) plain = plain && !c->validity && c->dtype != PLX_BOOL && (dtype_width(c->dtype) == 4 || dtype_width(c->dtype) == 8);
  std::vector<ColumnPtr> out;
  if (!plain) { for (const ColumnPtr& c : cols) out.push_back(gather(c, idx)); return out; }
  std::vector<int> widths; std::vector<const void*> src; std::vector<void*> dst;
  for (const ColumnPtr& c : cols) {
    auto o = std::make_shared<Column>();
    o->dtype = c->dtype; o->len = idx->len; o->null_count = 0;
    o->values = dev_alloc(values_bytes(c->dtype, idx->len));
    widths.push_back(dtype_width(c->dtype)); src.push_back(c->data()); dst.push_back(o->values->ptr);
    out.push_back(o);
  }
  k::gather_multi((int)cols.size(), widths.data(), src.data(), idx->values->as<uint32_t>(), idx->len, dst.data());
  return out;
}

// ------------------------------------------------------------------- reduce ---
static int sum_out_dtype(int dt) {
  switch (dt) {
    case PLX_BOOL: return PLX_U32;
    case PLX_I8: case PLX_I16: case PLX_U8: case PLX_U16: return PLX_I64;
    default: return dt;
  }
}

ScalarValue reduce(int op, const ColumnPtr& c) {
  ScalarValue r; r.v.u = 0; r.valid = true; r.dtype = PLX_U32;
  const int64_t nulls = column_null_count(c);
  const int64_t n_valid = c->len - nulls;
  if (op == PLX_AGG_LEN) { r.v.u = (uint32_t)c->len; return r; }
  if (op == PLX_AGG_COUNT) { r.v.u = (uint32_t)n_valid; return r; }
  if (c->dtype == PLX_BOOL) {
    // BooleanChunked::sum / mean (aggregate/mod.rs:253-300)
    int64_t trues = 0;
    if (c->len) {
      if (c->validity) {
        Buf t = dev_alloc_zero(bitmap_bytes(c->len));
        k::bitmap_op(0, c->values->as<uint64_t>(), c->validity->as<uint64_t>(), c->len, t->as<uint64_t>());
        trues = k::bitmap_popcount(t->as<uint64_t>(), c->len);
      } else trues = k::bitmap_popcount(c->values->as<uint64_t>(), c->len);
    }
    if (op == PLX_AGG_SUM) { r.dtype = PLX_U32; r.v.u = (uint32_t)trues; return r; }
    if (op == PLX_AGG_MEAN) { r.dtype = PLX_F64; if (n_valid == 0) r.valid = false; else r.v.f64 = (double)trues / (double)n_valid; return r; }
    fail(PLX_ERR_UNSUPPORTED, "reduce: min/max on boolean not on the hot path");
  }
  k::ReduceResult rr = k::reduce_all(c->dtype, c->data(), c->valid_words(), c->len);
  const int dt = c->dtype;
  switch (op) {
    case PLX_AGG_SUM: {
      r.dtype = sum_out_dtype(dt);
      if (dt == PLX_F64) r.v.f64 = rr.fsum;
      else if (dt == PLX_F32) r.v.f32 = (float)rr.fsum;
      else if (r.dtype == PLX_I32) r.v.i = (int64_t)(int32_t)(uint32_t)rr.isum;   // wrapping at the column's width
      else if (r.dtype == PLX_U32) r.v.u = (uint64_t)(uint32_t)rr.isum;
      else r.v.u = rr.isum;
      return r;
    }
    case PLX_AGG_MEAN: {
      r.dtype = dt == PLX_F32 ? PLX_F32 : PLX_F64;
      if (n_valid == 0) { r.valid = false; return r; }
      double m = rr.fsum / (double)n_valid;
      if (dt == PLX_F32) r.v.f32 = (float)m; else r.v.f64 = m;
      return r;
    }
    case PLX_AGG_MIN: case PLX_AGG_MAX: {
      r.dtype = dt;
      if (n_valid == 0) { r.valid = false; return r; }
      uint64_t bits = op == PLX_AGG_MIN ? rr.minmax_lo : rr.minmax_hi;
      if (dtype_is_float(dt)) {
        double d; memcpy(&d, &bits, 8);
        if (rr.n_ordered == 0) d = std::nan("");  // every valid value is NaN
        if (dt == PLX_F32) r.v.f32 = (float)d; else r.v.f64 = d;
      } else r.v.u = bits;
      return r;
    }
    default: fail(PLX_ERR_INVALID, "reduce: bad aggregation");
  }
}

ColumnPtr full_column(int dtype, plx_scalar v, bool valid, int64_t len) {
  auto out = std::make_shared<Column>();
  out->dtype = dtype; out->len = len;
  if (dtype == PLX_BOOL) {
    out->values = dev_alloc_zero(bitmap_bytes(len));
    if (valid && (v.u & 1) && len) {
      PLX_HIP(hipMemsetAsync(out->values->ptr, 0xff, (size_t)((len + 7) / 8), stream()));
      k::bitmap_op(0, out->values->as<uint64_t>(), out->values->as<uint64_t>(), len, out->values->as<uint64_t>());  // clears pad bits
    }
  } else {
    out->values = dev_alloc(values_bytes(dtype, len));
    k::fill(dtype_width(dtype), out->values->ptr, valid ? v.u : 0, len);
  }
  if (!valid) { out->validity = all_null_validity(len); out->null_count = len; } else out->null_count = 0;
  return out;
}
ColumnPtr scalar_column(const ScalarValue& s) { return full_column(s.dtype, s.v, s.valid, 1); }

// fill_null(literal) per node (reference: ChunkFillNullValue, polars-core/src/chunked_array/ops/fill_null.rs): same dtype, no nulls left
ColumnPtr fill_null(const ColumnPtr& c, plx_scalar v) {
  if (!c->validity || column_null_count(c) == 0) {
    if (!c->validity) return c;
    auto same = std::make_shared<Column>(*c);
    same->validity = nullptr; same->null_count = 0;
    return same;
  }
  auto out = std::make_shared<Column>();
  out->dtype = c->dtype; out->len = c->len; out->null_count = 0;
  const int w = dtype_width(c->dtype);
  out->values = w ? dev_alloc(values_bytes(c->dtype, c->len)) : dev_alloc_zero(bitmap_bytes(c->len));
  k::fill_null(w, c->values->ptr, c->validity->as<uint64_t>(), v.u, c->len, out->values->ptr);
  if (c->range_state == 1) out->range_state = 0;      // the literal may lie outside the cached range: recompute on demand
  return out;
}

ColumnPtr concat(const std::vector<ColumnPtr>& chunks) {
  PLX_REQUIRE(!chunks.empty(), PLX_ERR_INVALID, "concat: no chunks");
  if (chunks.size() == 1) return chunks[0];
  int64_t total = 0; bool any_valid = false;
  for (auto& c : chunks) { require_same_dtype(chunks[0], c, "concat"); total += c->len; any_valid |= (bool)c->validity; }
  auto out = std::make_shared<Column>();
  out->dtype = chunks[0]->dtype; out->len = total;
  const int w = dtype_width(out->dtype);
  out->values = w ? dev_alloc(values_bytes(out->dtype, total)) : dev_alloc_zero(bitmap_bytes(total));
  if (any_valid) out->validity = dev_alloc_zero(bitmap_bytes(total)); else out->null_count = 0;
  int64_t off = 0;
  for (auto& c : chunks) {
    if (c->len == 0) continue;
    if (w) PLX_HIP(hipMemcpyAsync((char*)out->values->ptr + off * w, c->data(), (size_t)c->len * w, hipMemcpyDeviceToDevice, stream()));
    else k::bitmap_blit(out->values->as<uint64_t>(), off, c->values->as<uint64_t>(), c->len);
    if (any_valid) {
      if (c->validity) k::bitmap_blit(out->validity->as<uint64_t>(), off, c->validity->as<uint64_t>(), c->len);
      else { ColumnPtr ones = full_column(PLX_BOOL, plx_scalar{1}, true, c->len); k::bitmap_blit(out->validity->as<uint64_t>(), off, ones->values->as<uint64_t>(), c->len); }
    }
    off += c->len;
  }
  return out;
}

ColumnPtr slice_copy(const ColumnPtr& c, int64_t offset, int64_t len) {
  PLX_REQUIRE(offset >= 0 && len >= 0 && offset + len <= c->len, PLX_ERR_INVALID, "slice out of bounds");
  auto idx = std::make_shared<Column>();
  idx->dtype = PLX_U32; idx->len = len; idx->values = dev_alloc(values_bytes(PLX_U32, len)); idx->null_count = 0;
  k::fill_iota_u32(idx->values->as<uint32_t>(), len);
  if (offset) { plx_scalar s; s.u = (uint32_t)offset; ColumnPtr shifted = arith_scalar(PLX_ADD, idx, s, false); return gather(c, shifted); }
  return gather(c, idx);
}

bool int_range(const ColumnPtr& c, int64_t* mn, int64_t* mx, bool allow_assumed) {
  PLX_REQUIRE(dtype_is_int(c->dtype), PLX_ERR_INVALID, "int_range: integer column required");
  // bounds the group-by planner only guessed (Column::range_assumed) are good for callers that check every row against them and can run again; everybody else gets
  // exact statistics (computed now, cached as such)
  if (c->range_assumed && !allow_assumed) { c->range_state = 0; c->range_trusted = true; c->range_assumed = false; c->range_verified = false; }
  if (c->range_state == 0) {
    if (c->len == 0) c->range_state = 2;
    else {
      k::ReduceResult rr = k::reduce_all(c->dtype, c->data(), c->valid_words(), c->len);
      if (rr.n_valid == 0) c->range_state = 2;
      else { c->range_state = 1; c->range_min = (int64_t)rr.minmax_lo; c->range_max = (int64_t)rr.minmax_hi; }
    }
  }
  if (c->range_state != 1) return false;
  *mn = c->range_min; *mx = c->range_max;
  return true;
}

}  // namespace ops
}  // namespace plx
