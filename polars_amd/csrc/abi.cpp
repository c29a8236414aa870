// abi.cpp -- the extern "C" boundary declared in include/polars_amd.h.
// Every entry point converts C++ exceptions into a status code + thread-local message
// (same convention as _polars_plugin_get_last_error_message, pyo3-polars derive.rs:26-45);
// nothing unwinds across the ABI.
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>

#include "core.hpp"
#include "datagen_device.hpp"
#include "engine.hpp"
#include "fused_shapes.hpp"
#include "jit.hpp"
#include "join.hpp"
#include "kernels.hpp"
#include "ops.hpp"
#include "scan.hpp"
#include "sort.hpp"

namespace plx {
void init_device(int ordinal);
void clear_handles();
const std::string& last_error_ref();
namespace comm {
void unique_id(uint8_t* out128);
uint64_t init(const uint8_t* id128, int rank, int ws);
void destroy(uint64_t h);
void info(uint64_t h, int* rank, int* ws);
FramePtr exchange_by_key(uint64_t h, const FramePtr& in, const std::string& key, uint64_t seed, uint64_t* rows_sent, uint64_t* bytes_sent);
FramePtr allgather_frame(uint64_t h, const FramePtr& in);
}  // namespace comm
}  // namespace plx

using namespace plx;

#define PLX_TRY try {
#define PLX_CATCH                                                                   \
  }                                                                                 \
  catch (const plx::Error& e) { plx::set_last_error(e.msg); return e.code; }        \
  catch (const std::bad_alloc&) { plx::set_last_error("host out of memory"); return PLX_ERR_OOM; } \
  catch (const std::exception& e) { plx::set_last_error(std::string("PANIC: ") + e.what()); return PLX_ERR_INVALID; } \
  catch (...) { plx::set_last_error("PANIC"); return PLX_ERR_INVALID; }              \
  return PLX_OK;

static thread_local std::string t_plan_desc;

extern "C" {

uint32_t plx_version(void) { return ((uint32_t)PLX_ABI_MAJOR << 16) | (uint32_t)PLX_ABI_MINOR; }
const char* plx_last_error(void) { return plx::last_error_ref().c_str(); }

int plx_init(int device_ordinal) { PLX_TRY init_device(device_ordinal); PLX_CATCH }
int plx_shutdown(void) {
  PLX_TRY
  clear_handles();
  pool_trim();
  PLX_CATCH
}
int plx_set_stream(void* hip_stream) { PLX_TRY device(); set_thread_stream((hipStream_t)hip_stream); PLX_CATCH }
int plx_synchronize(void) { PLX_TRY PLX_HIP(hipStreamSynchronize(stream())); PLX_CATCH }
int plx_set_cancel(int flag) { PLX_TRY device().cancel.store(flag); PLX_CATCH }
int plx_device_info(char* name_out, size_t name_cap, int32_t* cu_count, uint64_t* hbm_bytes) {
  PLX_TRY
  Device& d = device();
  if (name_out && name_cap) snprintf(name_out, name_cap, "%s", d.name.c_str());
  if (cu_count) *cu_count = d.cu_count;
  if (hbm_bytes) *hbm_bytes = d.hbm_bytes;
  PLX_CATCH
}
int plx_memory_stats(uint64_t* in_use, uint64_t* high_water) { PLX_TRY pool_stats(in_use, high_water); PLX_CATCH }
int plx_memory_trim(void) { PLX_TRY pool_trim(); PLX_CATCH }
int plx_memory_reserve(uint64_t bytes) { PLX_TRY device(); pool_reserve((size_t)bytes); PLX_CATCH }

// ---- columns ---------------------------------------------------------------------
int plx_column_from_host(plx_dtype dtype, const void* values, const uint8_t* validity, int64_t bit_offset, int64_t len, plx_column* out) {
  PLX_TRY
  PLX_REQUIRE(out, PLX_ERR_INVALID, "null out pointer");
  PLX_REQUIRE(values || len == 0, PLX_ERR_INVALID, "null values pointer");
  *out = register_column(column_from_host(dtype, values, validity, bit_offset, len));
  PLX_CATCH
}
int plx_column_from_device(plx_dtype dtype, void* dev_values, void* dev_validity, int64_t len, plx_column* out) {
  PLX_TRY
  device();
  PLX_REQUIRE(out && (dev_values || len == 0), PLX_ERR_INVALID, "null pointer");
  auto c = std::make_shared<Column>();
  c->dtype = dtype; c->len = len;
  c->values = dev_borrow(dev_values, values_bytes(dtype, len));
  if (dev_validity) c->validity = dev_borrow(dev_validity, bitmap_bytes(len)); else c->null_count = 0;
  *out = register_column(c);
  PLX_CATCH
}
int plx_column_placeholder(plx_dtype dtype, int64_t len, int nullable, int has_range, int64_t range_min, int64_t range_max, plx_column* out) {
  PLX_TRY
  auto c = std::make_shared<Column>();
  c->dtype = dtype; c->len = len;
  c->null_count = nullable ? 1 : 0;
  if (has_range) { c->range_state = 1; c->range_min = range_min; c->range_max = range_max; }
  *out = register_column(c);
  PLX_CATCH
}

int plx_column_set_bounds(plx_column col, int64_t lo, int64_t hi) {
  PLX_TRY
  ColumnPtr c = get_column(col);
  PLX_REQUIRE(dtype_is_int(c->dtype) && lo <= hi, PLX_ERR_INVALID, "set_bounds: integer columns only, lo <= hi");
  c->range_state = 1; c->range_min = lo; c->range_max = hi; c->range_trusted = false;
  PLX_CATCH
}

int plx_column_drop_statistics(plx_column col) {
  PLX_TRY
  ColumnPtr c = get_column(col);
  if (c->range_trusted || c->range_assumed) { c->range_state = 0; c->range_min = c->range_max = 0; c->range_trusted = true; c->range_assumed = false; c->range_verified = false; }       // bounds the caller declared are part of the column, not a cache
  c->no_assume = false;
  std::atomic_store(&c->key_sample, std::shared_ptr<void>());
  c->order_state = 0;
  c->repeats_as_build_key = false;
  PLX_CATCH
}

static int dtype_from_format(const char* f) {
  if (!f) return -1;
  if (!strcmp(f, "b")) return PLX_BOOL;
  if (!strcmp(f, "c")) return PLX_I8;
  if (!strcmp(f, "C")) return PLX_U8;
  if (!strcmp(f, "s")) return PLX_I16;
  if (!strcmp(f, "S")) return PLX_U16;
  if (!strcmp(f, "i")) return PLX_I32;
  if (!strcmp(f, "I")) return PLX_U32;
  if (!strcmp(f, "l")) return PLX_I64;
  if (!strcmp(f, "L")) return PLX_U64;
  if (!strcmp(f, "f")) return PLX_F32;
  if (!strcmp(f, "g")) return PLX_F64;
  if (!strncmp(f, "tdD", 3)) return PLX_I32;                      // date32
  if (!strncmp(f, "ts", 2) || !strncmp(f, "tD", 2)) return PLX_I64;  // timestamp / duration
  if (!strncmp(f, "tdm", 3)) return PLX_I64;
  return -1;
}
static const char* format_of_dtype(int dt) {
  static const char* f[] = {"b", "c", "s", "i", "l", "C", "S", "I", "L", "f", "g"};
  return f[dt];
}

static ColumnPtr import_one(struct ArrowArray* a, int dt) {
  PLX_REQUIRE(a->n_buffers >= 2, PLX_ERR_INVALID, "arrow import: primitive arrays carry 2 buffers");
  const uint8_t* validity = (const uint8_t*)a->buffers[0];
  const uint8_t* values = (const uint8_t*)a->buffers[1];
  if (a->null_count == 0) validity = nullptr;
  const void* vptr = values;
  if (dt != PLX_BOOL && values) vptr = values + (size_t)a->offset * dtype_width(dt);
  return column_from_host(dt, vptr, validity, a->offset, a->length);
}

int plx_column_import_arrow(struct ArrowArray* array, struct ArrowSchema* schema, plx_column* out) {
  PLX_TRY
  PLX_REQUIRE(array && schema && out, PLX_ERR_INVALID, "null pointer");
  int dt = dtype_from_format(schema->format);
  if (dt < 0) fail(PLX_ERR_UNSUPPORTED, std::string("arrow import: unsupported format '") + (schema->format ? schema->format : "") + "' (strings enter as dictionary codes)");
  ColumnPtr c = import_one(array, dt);
  if (array->release) array->release(array);
  if (schema->release) schema->release(schema);
  *out = register_column(c);
  PLX_CATCH
}

int plx_column_import_series(plx_series_export* s, plx_column* out) {
  PLX_TRY
  PLX_REQUIRE(s && s->field && out, PLX_ERR_INVALID, "null pointer");
  int dt = dtype_from_format(s->field->format);
  if (dt < 0) fail(PLX_ERR_UNSUPPORTED, "series import: unsupported dtype");
  std::vector<ColumnPtr> chunks;
  for (size_t i = 0; i < s->len; i++) chunks.push_back(import_one(s->arrays[i], dt));
  ColumnPtr c = chunks.empty() ? column_from_host(dt, nullptr, nullptr, 0, 0) : ops::concat(chunks);
  if (s->release) s->release(s);  // callee owns the inputs (plugin.rs:122-125)
  *out = register_column(c);
  PLX_CATCH
}

namespace {
struct ExportHolder {
  std::vector<uint8_t> values, validity;
  const void* bufs[2];
  std::string name;
};
void release_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  delete (ExportHolder*)a->private_data;
  a->release = nullptr;
}
void release_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  delete (std::string*)s->private_data;
  s->release = nullptr;
}
void fill_export(const ColumnPtr& c, const char* name, struct ArrowArray* oa, struct ArrowSchema* os) {
  auto* h = new ExportHolder();
  const size_t vb = c->dtype == PLX_BOOL ? (size_t)((c->len + 7) / 8) : (size_t)c->len * dtype_width(c->dtype);
  h->values.resize(vb + 8);
  h->validity.resize((size_t)((c->len + 7) / 8) + 8);
  int32_t hv = 0;
  column_to_host(c, h->values.data(), h->validity.data(), &hv);
  int64_t nulls = hv ? column_null_count(c) : 0;
  h->bufs[0] = nulls ? h->validity.data() : nullptr;
  h->bufs[1] = h->values.data();
  memset(oa, 0, sizeof(*oa));
  oa->length = c->len; oa->null_count = nulls; oa->offset = 0; oa->n_buffers = 2; oa->buffers = h->bufs; oa->release = release_array; oa->private_data = h;
  memset(os, 0, sizeof(*os));
  auto* nm = new std::string(name ? name : "");
  os->format = format_of_dtype(c->dtype); os->name = nm->c_str(); os->flags = 2 /* ARROW_FLAG_NULLABLE */; os->release = release_schema; os->private_data = nm;
}
void release_series(plx_series_export* s) {
  if (!s || !s->release) return;
  if (s->arrays) { for (size_t i = 0; i < s->len; i++) { if (s->arrays[i]) { if (s->arrays[i]->release) s->arrays[i]->release(s->arrays[i]); delete s->arrays[i]; } } delete[] s->arrays; }
  if (s->field) { if (s->field->release) s->field->release(s->field); delete s->field; }
  s->release = nullptr;
}
}  // namespace

int plx_column_export_arrow(plx_column col, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
  PLX_TRY
  PLX_REQUIRE(out_array && out_schema, PLX_ERR_INVALID, "null pointer");
  fill_export(get_column(col), "", out_array, out_schema);
  PLX_CATCH
}
int plx_column_export_series(plx_column col, const char* name, plx_series_export* out) {
  PLX_TRY
  PLX_REQUIRE(out, PLX_ERR_INVALID, "null pointer");
  out->field = new ArrowSchema();
  out->arrays = new ArrowArray*[1];
  out->arrays[0] = new ArrowArray();
  out->len = 1;
  fill_export(get_column(col), name, out->arrays[0], out->field);
  out->release = release_series;
  out->private_data = out->arrays;  // non-null == success (plugin.rs:127-133)
  PLX_CATCH
}
int plx_column_to_host(plx_column col, void* values_out, uint8_t* validity_out, int32_t* has_validity_out) {
  PLX_TRY
  ColumnPtr c = get_column(col);
  PLX_REQUIRE(c->values || c->len == 0, PLX_ERR_INVALID, "placeholder column has no data");
  column_to_host(c, values_out, validity_out, has_validity_out);
  PLX_CATCH
}
int plx_column_copy_to_device(plx_column col, void* dev_values_out, void* dev_validity_out) {
  PLX_TRY
  ColumnPtr c = get_column(col);
  PLX_REQUIRE(c->values || c->len == 0, PLX_ERR_INVALID, "placeholder column has no data");
  const size_t vb = c->dtype == PLX_BOOL ? (size_t)((c->len + 7) / 8) : (size_t)c->len * dtype_width(c->dtype);
  const size_t nb = (size_t)((c->len + 7) / 8);
  if (dev_values_out && vb) PLX_HIP(hipMemcpyAsync(dev_values_out, c->values->ptr, vb, hipMemcpyDeviceToDevice, stream()));
  if (dev_validity_out && nb) {
    if (c->validity) PLX_HIP(hipMemcpyAsync(dev_validity_out, c->validity->ptr, nb, hipMemcpyDeviceToDevice, stream()));
    else PLX_HIP(hipMemsetAsync(dev_validity_out, 0xff, nb, stream()));
  }
  PLX_HIP(hipStreamSynchronize(stream()));
  PLX_CATCH
}
int plx_column_info(plx_column col, plx_dtype* dtype, int64_t* len, int64_t* null_count) {
  PLX_TRY
  ColumnPtr c = get_column(col);
  if (dtype) *dtype = (plx_dtype)c->dtype;
  if (len) *len = c->len;
  if (null_count) *null_count = c->values ? column_null_count(c) : c->null_count;
  PLX_CATCH
}
int plx_column_device_ptrs(plx_column col, void** values, void** validity) {
  PLX_TRY
  ColumnPtr c = get_column(col);
  if (values) *values = c->values ? c->values->ptr : nullptr;
  if (validity) *validity = c->validity ? c->validity->ptr : nullptr;
  PLX_CATCH
}
int plx_column_retain(plx_column col) { PLX_TRY retain_column(col); PLX_CATCH }
int plx_column_free(plx_column col) { PLX_TRY free_column(col); PLX_CATCH }

// ---- kernel-level entry points ------------------------------------------------------
int plx_cmp(plx_cmp_op op, plx_column lhs, plx_column rhs, plx_column* out) { PLX_TRY *out = register_column(ops::cmp(op, get_column(lhs), get_column(rhs))); PLX_CATCH }
int plx_cmp_scalar(plx_cmp_op op, plx_column lhs, plx_scalar rhs, plx_column* out) { PLX_TRY *out = register_column(ops::cmp_scalar(op, get_column(lhs), rhs)); PLX_CATCH }
int plx_bitmap_binop(plx_bitmap_op op, plx_column lhs, plx_column rhs, plx_column* out) { PLX_TRY *out = register_column(ops::bool_binop(op, get_column(lhs), get_column(rhs))); PLX_CATCH }
int plx_bitmap_not(plx_column col, plx_column* out) { PLX_TRY *out = register_column(ops::bool_not(get_column(col))); PLX_CATCH }
int plx_arith(plx_arith_op op, plx_column lhs, plx_column rhs, plx_column* out) { PLX_TRY *out = register_column(ops::arith(op, get_column(lhs), get_column(rhs))); PLX_CATCH }
int plx_arith_scalar(plx_arith_op op, plx_column col, plx_scalar scalar, int scalar_on_left, plx_column* out) {
  PLX_TRY *out = register_column(ops::arith_scalar(op, get_column(col), scalar, scalar_on_left != 0)); PLX_CATCH
}
int plx_cast(plx_column col, plx_dtype to, plx_column* out) { PLX_TRY *out = register_column(ops::cast(get_column(col), to)); PLX_CATCH }
int plx_filter(plx_column col, plx_column mask, plx_column* out) { PLX_TRY *out = register_column(ops::filter(get_column(col), get_column(mask))); PLX_CATCH }
int plx_gather(plx_column col, plx_column idx, plx_column* out) { PLX_TRY *out = register_column(ops::gather(get_column(col), get_column(idx))); PLX_CATCH }
int plx_reduce(plx_agg_op op, plx_column col, plx_scalar* out_value, plx_dtype* out_dtype, int32_t* out_valid) {
  PLX_TRY
  ops::ScalarValue r = ops::reduce(op, get_column(col));
  if (out_value) *out_value = r.v;
  if (out_dtype) *out_dtype = (plx_dtype)r.dtype;
  if (out_valid) *out_valid = r.valid ? 1 : 0;
  PLX_CATCH
}

int plx_groupby_agg(const plx_column* keys, int32_t n_keys, const plx_column* values, const plx_agg_op* aggs, int32_t n_aggs, int32_t maintain_order,
                    plx_column* out_keys, plx_column* out_aggs) {
  PLX_TRY
  PLX_REQUIRE(keys && n_keys >= 1 && (n_aggs == 0 || (values && aggs && out_aggs)) && out_keys, PLX_ERR_INVALID, "bad arguments");
  std::vector<ColumnPtr> k, v, ok, oa;
  std::vector<int> a;
  for (int i = 0; i < n_keys; i++) k.push_back(get_column(keys[i]));
  for (int i = 0; i < n_aggs; i++) { a.push_back(aggs[i]); v.push_back((aggs[i] == PLX_AGG_LEN || values[i] == 0) ? nullptr : get_column(values[i])); }
  std::string d;
  engine::groupby_columns(k, v, a, maintain_order != 0, ok, oa, &d);
  t_plan_desc = d;
  for (int i = 0; i < n_keys; i++) out_keys[i] = register_column(ok[i]);
  for (int i = 0; i < n_aggs; i++) out_aggs[i] = register_column(oa[i]);
  PLX_CATCH
}

int plx_join_indices(plx_join_how how, plx_column left_key, plx_column right_key, plx_column* out_left_idx, plx_column* out_right_idx) {
  PLX_TRY
  ColumnPtr li, ri;
  std::string d;
  join::join_indices(how, get_column(left_key), get_column(right_key), li, ri, &d);
  t_plan_desc = d;
  *out_left_idx = register_column(li);
  *out_right_idx = ri ? register_column(ri) : 0;
  PLX_CATCH
}

int plx_sort_indices(const plx_column* by, int32_t n_by, const uint8_t* descending, const uint8_t* nulls_last, int64_t limit, plx_column* out_idx) {
  PLX_TRY
  PLX_REQUIRE(by && n_by > 0 && out_idx, PLX_ERR_INVALID, "sort_indices: need at least one key column");
  std::vector<sort::SortKey> keys;
  for (int i = 0; i < n_by; i++) {
    sort::SortKey k;
    k.col = get_column(by[i]); k.descending = descending && descending[i]; k.nulls_last = nulls_last && nulls_last[i];
    keys.push_back(k);
  }
  std::string d;
  ColumnPtr idx = sort::sort_indices(keys, limit, &d);
  t_plan_desc = d;
  *out_idx = register_column(idx);
  PLX_CATCH
}

int plx_hash_partition(plx_column key, int32_t n_partitions, uint64_t seed, plx_column* out_perm, int64_t* counts_out) {
  PLX_TRY
  PLX_REQUIRE(out_perm && counts_out, PLX_ERR_INVALID, "null pointer");
  ColumnPtr perm;
  join::hash_partition(get_column(key), n_partitions, seed, perm, counts_out);
  *out_perm = register_column(perm);
  PLX_CATCH
}

// ---- frames -----------------------------------------------------------------------
int plx_frame_new(const char* const* names, const plx_column* cols, int32_t n_cols, plx_frame* out) {
  PLX_TRY
  PLX_REQUIRE(out && (n_cols == 0 || (names && cols)), PLX_ERR_INVALID, "null pointer");
  auto f = std::make_shared<Frame>();
  for (int i = 0; i < n_cols; i++) {
    ColumnPtr c = get_column(cols[i]);
    if (i == 0) f->height = c->len;
    PLX_REQUIRE(c->len == f->height, PLX_ERR_SHAPE, std::string("frame columns have different lengths (") + names[i] + ")");
    PLX_REQUIRE(f->find(names[i]) < 0, PLX_ERR_INVALID, std::string("duplicate column name ") + names[i]);
    f->names.push_back(names[i]); f->cols.push_back(c);
  }
  *out = register_frame(f);
  PLX_CATCH
}
int plx_frame_free(plx_frame f) { PLX_TRY free_frame(f); PLX_CATCH }
int plx_frame_concat(const plx_frame* frames, int32_t n_frames, plx_frame* out) {
  PLX_TRY
  PLX_REQUIRE(frames && out && n_frames >= 1, PLX_ERR_INVALID, "frame_concat: bad arguments");
  std::vector<FramePtr> fs;
  for (int32_t i = 0; i < n_frames; i++) fs.push_back(get_frame(frames[i]));
  auto o = std::make_shared<Frame>();
  o->names = fs[0]->names;
  for (const FramePtr& f : fs) {
    PLX_REQUIRE(f->names == o->names, PLX_ERR_SHAPE, "frame_concat: frames have different columns");
    o->height += f->height;
  }
  for (size_t c = 0; c < o->names.size(); c++) {
    std::vector<ColumnPtr> chunks;
    for (const FramePtr& f : fs) chunks.push_back(f->cols[c]);
    o->cols.push_back(ops::concat(chunks));        // dtype mismatches are reported by the op
  }
  *out = register_frame(o);
  PLX_CATCH
}
int plx_frame_shape(plx_frame f, int64_t* height, int32_t* width) {
  PLX_TRY
  FramePtr fr = get_frame(f);
  if (height) *height = fr->height;
  if (width) *width = (int32_t)fr->cols.size();
  PLX_CATCH
}
int plx_frame_column(plx_frame f, int32_t i, const char** name_out, plx_column* col_out) {
  PLX_TRY
  FramePtr fr = get_frame(f);
  PLX_REQUIRE(i >= 0 && i < (int)fr->cols.size(), PLX_ERR_INVALID, "column index out of range");
  if (name_out) *name_out = fr->names[i].c_str();
  if (col_out) *col_out = register_column(fr->cols[i]);
  PLX_CATCH
}

int plx_frame_dtypes(plx_frame f, int32_t* dtypes_out) {
  PLX_TRY
  FramePtr fr = get_frame(f);
  PLX_REQUIRE(dtypes_out || fr->cols.empty(), PLX_ERR_INVALID, "null pointer");
  for (size_t i = 0; i < fr->cols.size(); i++) dtypes_out[i] = fr->cols[i]->dtype;
  PLX_CATCH
}
int plx_frame_to_host(plx_frame f, void* const* values_out, uint8_t* const* validity_out, int32_t* has_validity_out) {
  PLX_TRY
  FramePtr fr = get_frame(f);
  // small results (a group-by / top-k output): one pack launch + ONE page-locked D2H copy instead of a staged pageable copy per buffer
  {
    struct Piece { const void* src; void* dst; size_t bytes; };
    std::vector<Piece> pieces;
    size_t total = 0;
    bool ok = pinned_bounce() != nullptr;
    for (size_t i = 0; ok && i < fr->cols.size(); i++) {
      const ColumnPtr& c = fr->cols[i];
      if (!(c->values || c->len == 0)) { ok = false; break; }
      const size_t vb = c->dtype == PLX_BOOL ? (size_t)((c->len + 7) / 8) : (size_t)c->len * dtype_width(c->dtype);
      const size_t nb = (size_t)((c->len + 7) / 8);
      if (values_out && values_out[i] && vb) { pieces.push_back({c->values->ptr, values_out[i], vb}); total += (vb + 15) & ~size_t(15); }
      if (validity_out && validity_out[i] && nb && c->validity) { pieces.push_back({c->validity->ptr, validity_out[i], nb}); total += (nb + 15) & ~size_t(15); }
      if (total > kBounceBytes) ok = false;
    }
    if (ok && !pieces.empty()) {
      Buf staging = dev_alloc(total);
      std::vector<size_t> offs(pieces.size());
      size_t at = 0;
      for (size_t j = 0; j < pieces.size(); j++) { offs[j] = at; at += (pieces[j].bytes + 15) & ~size_t(15); }
      for (size_t base = 0; base < pieces.size(); base += k::kPackMax) {
        k::PackBatch b{}; b.n = (int)std::min<size_t>(k::kPackMax, pieces.size() - base);
        for (int j = 0; j < b.n; j++) { b.src[j] = pieces[base + j].src; b.bytes[j] = (uint32_t)pieces[base + j].bytes; b.off[j] = (uint32_t)offs[base + j]; }
        k::pack_buffers(b, staging->ptr);
      }
      std::lock_guard<std::mutex> lk(bounce_mutex());
      uint8_t* host = reinterpret_cast<uint8_t*>(pinned_bounce());
      PLX_HIP(hipMemcpyAsync(host, staging->ptr, total, hipMemcpyDeviceToHost, stream()));
      PLX_HIP(hipStreamSynchronize(stream()));
      for (size_t j = 0; j < pieces.size(); j++) memcpy(pieces[j].dst, host + offs[j], pieces[j].bytes);
      for (size_t i = 0; i < fr->cols.size(); i++) {
        const ColumnPtr& c = fr->cols[i];
        const size_t nb = (size_t)((c->len + 7) / 8);
        if (validity_out && validity_out[i] && nb && !c->validity) memset(validity_out[i], 0xff, nb);
        if (has_validity_out) has_validity_out[i] = c->validity ? 1 : 0;
      }
      return PLX_OK;
    }
    if (ok && pieces.empty() && !fr->cols.empty()) {   // zero-row frame: nothing to copy
      for (size_t i = 0; i < fr->cols.size(); i++) if (has_validity_out) has_validity_out[i] = fr->cols[i]->validity ? 1 : 0;
      return PLX_OK;
    }
  }
  for (size_t i = 0; i < fr->cols.size(); i++) {
    const ColumnPtr& c = fr->cols[i];
    PLX_REQUIRE(c->values || c->len == 0, PLX_ERR_INVALID, "placeholder column has no data");
    const size_t vb = c->dtype == PLX_BOOL ? (size_t)((c->len + 7) / 8) : (size_t)c->len * dtype_width(c->dtype);
    const size_t nb = (size_t)((c->len + 7) / 8);
    if (values_out && values_out[i] && vb) PLX_HIP(hipMemcpyAsync(values_out[i], c->values->ptr, vb, hipMemcpyDeviceToHost, stream()));
    if (validity_out && validity_out[i] && nb) {
      if (c->validity) PLX_HIP(hipMemcpyAsync(validity_out[i], c->validity->ptr, nb, hipMemcpyDeviceToHost, stream()));
      else memset(validity_out[i], 0xff, nb);
    }
    if (has_validity_out) has_validity_out[i] = c->validity ? 1 : 0;
  }
  PLX_HIP(hipStreamSynchronize(stream()));
  PLX_CATCH
}

// ---- the reference's expression-plugin ABI -----------------------------------------------------
// An UNMODIFIED Polars can dlopen this library as an expression plugin: its loader resolves
//   _polars_plugin_get_version, _polars_plugin_get_last_error_message, _polars_plugin_<name>, _polars_plugin_field_<name>
// (crates/polars-plan/src/plans/aexpr/function_expr/plugin.rs:23-137,139-227; the plugin side is normally generated by
// pyo3-polars-derive/src/lib.rs:140-163).  Convention restated here: the callee takes ownership of the input SeriesExports
// and calls their release (plugin.rs:122-125); the output is written to *out, a null out->private_data means failure and the
// caller then fetches the thread-local message; kwargs are a pickled dict (serde-pickle) -- the only key read here is "op".
// CallerContext bit 0 = the caller is already parallel (version_0.rs:136-162): such calls come from rayon workers at the same
// time, so each of those threads gets its own HIP stream (no host threads are ever spawned here).
extern "C++" {
namespace {
thread_local std::string t_plugin_error;
struct PluginCtx { uint64_t bitflags; };

hipStream_t plugin_thread_stream() {
  static thread_local hipStream_t s = nullptr;
  if (!s) PLX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}
// value of a string-valued key in a pickled {str: str} dict: scans SHORT_BINUNICODE (0x8c), BINUNICODE ('X') and
// SHORT_BINSTRING ('U') tokens, which is how every pickle protocol >= 2 writes short str items
std::string pickle_str_value(const uint8_t* kw, size_t n, const char* key) {
  std::vector<std::string> toks;
  for (size_t i = 0; i < n;) {
    const uint8_t c = kw[i];
    size_t len = 0, hdr = 0;
    if ((c == 0x8c || c == 'U') && i + 1 < n) { len = kw[i + 1]; hdr = 2; }
    else if (c == 'X' && i + 4 < n) { len = (size_t)kw[i + 1] | ((size_t)kw[i + 2] << 8) | ((size_t)kw[i + 3] << 16) | ((size_t)kw[i + 4] << 24); hdr = 5; }
    if (hdr && i + hdr + len <= n) { toks.emplace_back((const char*)kw + i + hdr, len); i += hdr + len; }
    else i++;
  }
  for (size_t i = 0; i + 1 < toks.size(); i++) if (toks[i] == key) return toks[i + 1];
  return "";
}
// a length-1 input is a literal (Polars broadcasts it): read it on the host instead of uploading it
bool series_scalar(const plx_series_export& s, int dt, plx_scalar* out, bool* valid) {
  int64_t total = 0; const ArrowArray* one = nullptr;
  for (size_t i = 0; i < s.len; i++) { total += s.arrays[i]->length; if (s.arrays[i]->length) one = s.arrays[i]; }
  if (total != 1 || !one || one->n_buffers < 2) return false;
  const uint8_t* vb = (const uint8_t*)one->buffers[0];
  *valid = one->null_count == 0 || !vb || ((vb[one->offset >> 3] >> (one->offset & 7)) & 1);
  out->u = 0;
  const uint8_t* p = (const uint8_t*)one->buffers[1];
  if (!*valid || !p) return true;
  if (dt == PLX_BOOL) { out->u = (p[one->offset >> 3] >> (one->offset & 7)) & 1; return true; }
  const int w = dtype_width(dt);
  p += (size_t)one->offset * w;
  switch (dt) {
    case PLX_I8: out->i = *(const int8_t*)p; break;
    case PLX_I16: { int16_t v; memcpy(&v, p, 2); out->i = v; } break;
    case PLX_I32: { int32_t v; memcpy(&v, p, 4); out->i = v; } break;
    case PLX_F32: memcpy(&out->f32, p, 4); break;
    default: memcpy(&out->u, p, (size_t)w); break;
  }
  return true;
}
struct PluginInput { ColumnPtr col; bool is_scalar = false; plx_scalar scalar{}; bool scalar_valid = true; int dtype = 0; std::string name; };

// takes ownership of every input (also on failure), runs body(inputs) -> column, exports it under the first input's name
template <class Body>
void plugin_call(const plx_series_export* inputs, size_t n, plx_series_export* out, const void* ctx, size_t want_inputs, Body body) {
  std::vector<plx_series_export> own;
  for (size_t i = 0; i < n; i++) own.push_back(inputs[i]);        // moved out of the caller's slice (the caller forgets it)
  auto release_all = [&] { for (auto& s : own) if (s.release) s.release(&s); };
  hipStream_t prev = nullptr; bool switched = false;
  try {
    PLX_REQUIRE(out, PLX_ERR_INVALID, "null output SeriesExport");
    PLX_REQUIRE(n == want_inputs, PLX_ERR_INVALID, "expected " + std::to_string(want_inputs) + " input series, got " + std::to_string(n));
    device();
    if (ctx && (reinterpret_cast<const PluginCtx*>(ctx)->bitflags & 1)) { prev = stream(); set_thread_stream(plugin_thread_stream()); switched = true; }
    std::vector<PluginInput> in(n);
    for (size_t i = 0; i < n; i++) {
      plx_series_export& s = own[i];
      PLX_REQUIRE(s.field, PLX_ERR_INVALID, "input series without a field");
      in[i].dtype = dtype_from_format(s.field->format);
      in[i].name = s.field->name ? s.field->name : "";
      if (in[i].dtype < 0) fail(PLX_ERR_UNSUPPORTED, std::string("plugin input: unsupported Arrow format '") + (s.field->format ? s.field->format : "") + "'");
      if (n > 1 && series_scalar(s, in[i].dtype, &in[i].scalar, &in[i].scalar_valid)) in[i].is_scalar = true;
      if (i + 1 == n && n > 1) { bool all = true; for (size_t j = 0; j < n; j++) all = all && in[j].is_scalar; if (all) { in[0].is_scalar = false; if (!in[0].col) {
        std::vector<ColumnPtr> ch0; for (size_t c = 0; c < own[0].len; c++) ch0.push_back(import_one(own[0].arrays[c], in[0].dtype)); in[0].col = ops::concat(ch0); } } }
      if (!in[i].is_scalar) {
        std::vector<ColumnPtr> chunks;
        for (size_t c = 0; c < s.len; c++) chunks.push_back(import_one(s.arrays[c], in[i].dtype));
        in[i].col = chunks.empty() ? column_from_host(in[i].dtype, nullptr, nullptr, 0, 0) : ops::concat(chunks);
        in[i].is_scalar = false;
      }
    }
    release_all();                                                // host buffers are no longer referenced (uploads synchronise)
    ColumnPtr res = body(in);
    plx_series_export tmp{};
    tmp.field = new ArrowSchema();
    tmp.arrays = new ArrowArray*[1];
    tmp.arrays[0] = new ArrowArray();
    tmp.len = 1;
    fill_export(res, in[0].name.c_str(), tmp.arrays[0], tmp.field);
    tmp.release = release_series;
    tmp.private_data = tmp.arrays;
    *out = tmp;
    if (switched) set_thread_stream(prev == device().own_stream ? nullptr : prev);
    return;
  } catch (const plx::Error& e) { t_plugin_error = e.msg; }
  catch (const std::exception& e) { t_plugin_error = std::string("PANIC: ") + e.what(); }
  catch (...) { t_plugin_error = "PANIC"; }
  release_all();
  if (switched) { try { set_thread_stream(prev == device().own_stream ? nullptr : prev); } catch (...) {} }
  if (out) out->private_data = nullptr;
}
int cmp_op_of(const std::string& s) {
  static const char* n[] = {"eq", "ne", "lt", "le", "gt", "ge"};
  for (int i = 0; i < 6; i++) if (s == n[i]) return i;
  return -1;
}
int arith_op_of(const std::string& s) {
  static const char* n[] = {"add", "sub", "mul", "truediv", "floordiv", "mod"};
  for (int i = 0; i < 6; i++) if (s == n[i]) return i;
  return -1;
}
ColumnPtr plugin_cmp(int op, std::vector<PluginInput>& in) {
  PLX_REQUIRE(op >= 0, PLX_ERR_INVALID, "plx_cmp: kwargs must carry op in {eq, ne, lt, le, gt, ge}");
  PLX_REQUIRE(in[0].dtype == in[1].dtype, PLX_ERR_INVALID, "plx_cmp: operands differ in dtype (type coercion happens in the optimizer)");
  static const int flip[] = {PLX_EQ, PLX_NE, PLX_GT, PLX_GE, PLX_LT, PLX_LE};
  if (in[1].is_scalar) return ops::cmp_scalar(op, in[0].col, in[1].scalar, !in[1].scalar_valid);
  if (in[0].is_scalar) return ops::cmp_scalar(flip[op], in[1].col, in[0].scalar, !in[0].scalar_valid);
  return ops::cmp(op, in[0].col, in[1].col);
}
ColumnPtr plugin_arith(int op, std::vector<PluginInput>& in) {
  PLX_REQUIRE(op >= 0, PLX_ERR_INVALID, "plx_arith: kwargs must carry op in {add, sub, mul, truediv, floordiv, mod}");
  PLX_REQUIRE(in[0].dtype == in[1].dtype, PLX_ERR_INVALID, "plx_arith: operands differ in dtype (type coercion happens in the optimizer)");
  auto null_like = [&](const ColumnPtr& c) { const int odt = (op == PLX_TRUE_DIV && dtype_is_int(c->dtype)) ? PLX_F64 : c->dtype; plx_scalar z; z.u = 0; return ops::full_column(odt, z, false, c->len); };
  if (in[1].is_scalar) return in[1].scalar_valid ? ops::arith_scalar(op, in[0].col, in[1].scalar, false) : null_like(in[0].col);
  if (in[0].is_scalar) return in[0].scalar_valid ? ops::arith_scalar(op, in[1].col, in[0].scalar, true) : null_like(in[1].col);
  return ops::arith(op, in[0].col, in[1].col);
}
ColumnPtr plugin_reduce(int agg, std::vector<PluginInput>& in) { return ops::scalar_column(ops::reduce(agg, in[0].col)); }

// output field of a plugin function: name of the first input, dtype by the rule of the operator
enum FieldRule { FIELD_BOOL, FIELD_SAME, FIELD_ARITH, FIELD_SUM, FIELD_MEAN };
void plugin_field(const ArrowSchema* fields, size_t n, ArrowSchema* out, FieldRule rule, int arith_op) {
  try {
    PLX_REQUIRE(fields && n >= 1 && out, PLX_ERR_INVALID, "plugin field: bad arguments");
    const int dt = dtype_from_format(fields[0].format);
    if (dt < 0) fail(PLX_ERR_UNSUPPORTED, std::string("plugin field: unsupported Arrow format '") + (fields[0].format ? fields[0].format : "") + "'");
    int odt = dt;
    switch (rule) {
      case FIELD_BOOL: odt = PLX_BOOL; break;
      case FIELD_SAME: break;
      case FIELD_ARITH: odt = (arith_op == PLX_TRUE_DIV && dtype_is_int(dt)) ? PLX_F64 : dt; break;
      case FIELD_SUM: odt = (dt == PLX_I8 || dt == PLX_I16 || dt == PLX_U8 || dt == PLX_U16) ? PLX_I64 : (dt == PLX_BOOL ? PLX_U32 : dt); break;   // sum_output_dtype, aggregate/mod.rs:55-64
      case FIELD_MEAN: odt = dt == PLX_F32 ? PLX_F32 : PLX_F64; break;
    }
    memset(out, 0, sizeof(*out));
    auto* nm = new std::string(fields[0].name ? fields[0].name : "");
    // temporal formats keep their logical type when the physical type is unchanged
    const bool keep = (rule == FIELD_SAME) && fields[0].format && fields[0].format[0] == 't';
    auto* fmt = new std::string(keep ? fields[0].format : format_of_dtype(odt));
    struct Holder { std::string* name; std::string* fmt; };
    out->format = fmt->c_str(); out->name = nm->c_str(); out->flags = 2;
    out->private_data = new Holder{nm, fmt};
    out->release = [](ArrowSchema* s) { if (!s || !s->release) return; auto* h = (Holder*)s->private_data; delete h->name; delete h->fmt; delete h; s->release = nullptr; };
    return;
  } catch (const plx::Error& e) { t_plugin_error = e.msg; }
  catch (...) { t_plugin_error = "PANIC"; }
  if (out) out->release = nullptr;     // ArrowSchema::is_null(): no release callback = failure
}
}  // namespace
}  // extern "C++"

uint32_t _polars_plugin_get_version(void) { return ((uint32_t)PLX_ABI_MAJOR << 16) | (uint32_t)PLX_ABI_MINOR; }
char* _polars_plugin_get_last_error_message(void) { return const_cast<char*>(t_plugin_error.c_str()); }

#define PLX_PLUGIN_SIG const plx_series_export* inputs, size_t n_inputs, const uint8_t* kwargs, size_t kwargs_len, plx_series_export* out, const void* ctx
#define PLX_FIELD_SIG const struct ArrowSchema* fields, size_t n_fields, struct ArrowSchema* out, const uint8_t* kwargs, size_t kwargs_len
void _polars_plugin_plx_cmp(PLX_PLUGIN_SIG) {
  const int op = cmp_op_of(pickle_str_value(kwargs, kwargs_len, "op"));
  plugin_call(inputs, n_inputs, out, ctx, 2, [&](std::vector<PluginInput>& in) { return plugin_cmp(op, in); });
}
void _polars_plugin_field_plx_cmp(PLX_FIELD_SIG) { (void)kwargs; (void)kwargs_len; plugin_field(fields, n_fields, out, FIELD_BOOL, 0); }
void _polars_plugin_plx_arith(PLX_PLUGIN_SIG) {
  const int op = arith_op_of(pickle_str_value(kwargs, kwargs_len, "op"));
  plugin_call(inputs, n_inputs, out, ctx, 2, [&](std::vector<PluginInput>& in) { return plugin_arith(op, in); });
}
void _polars_plugin_field_plx_arith(PLX_FIELD_SIG) { plugin_field(fields, n_fields, out, FIELD_ARITH, arith_op_of(pickle_str_value(kwargs, kwargs_len, "op"))); }
// one symbol per operator as well, for callers that pass no kwargs
#define PLX_PLUGIN_CMP(NAME, OP)                                                                                                                  \
  void _polars_plugin_plx_##NAME(PLX_PLUGIN_SIG) { (void)kwargs; (void)kwargs_len; plugin_call(inputs, n_inputs, out, ctx, 2, [&](std::vector<PluginInput>& in) { return plugin_cmp(OP, in); }); } \
  void _polars_plugin_field_plx_##NAME(PLX_FIELD_SIG) { (void)kwargs; (void)kwargs_len; plugin_field(fields, n_fields, out, FIELD_BOOL, 0); }
#define PLX_PLUGIN_ARITH(NAME, OP)                                                                                                                \
  void _polars_plugin_plx_##NAME(PLX_PLUGIN_SIG) { (void)kwargs; (void)kwargs_len; plugin_call(inputs, n_inputs, out, ctx, 2, [&](std::vector<PluginInput>& in) { return plugin_arith(OP, in); }); } \
  void _polars_plugin_field_plx_##NAME(PLX_FIELD_SIG) { (void)kwargs; (void)kwargs_len; plugin_field(fields, n_fields, out, FIELD_ARITH, OP); }
#define PLX_PLUGIN_REDUCE(NAME, AGG, RULE)                                                                                                        \
  void _polars_plugin_plx_##NAME(PLX_PLUGIN_SIG) { (void)kwargs; (void)kwargs_len; plugin_call(inputs, n_inputs, out, ctx, 1, [&](std::vector<PluginInput>& in) { return plugin_reduce(AGG, in); }); } \
  void _polars_plugin_field_plx_##NAME(PLX_FIELD_SIG) { (void)kwargs; (void)kwargs_len; plugin_field(fields, n_fields, out, RULE, 0); }
PLX_PLUGIN_CMP(eq, PLX_EQ) PLX_PLUGIN_CMP(ne, PLX_NE) PLX_PLUGIN_CMP(lt, PLX_LT) PLX_PLUGIN_CMP(le, PLX_LE) PLX_PLUGIN_CMP(gt, PLX_GT) PLX_PLUGIN_CMP(ge, PLX_GE)
PLX_PLUGIN_ARITH(add, PLX_ADD) PLX_PLUGIN_ARITH(sub, PLX_SUB) PLX_PLUGIN_ARITH(mul, PLX_MUL) PLX_PLUGIN_ARITH(truediv, PLX_TRUE_DIV)
PLX_PLUGIN_ARITH(floordiv, PLX_FLOOR_DIV) PLX_PLUGIN_ARITH(mod, PLX_MOD)
PLX_PLUGIN_REDUCE(sum, PLX_AGG_SUM, FIELD_SUM) PLX_PLUGIN_REDUCE(mean, PLX_AGG_MEAN, FIELD_MEAN) PLX_PLUGIN_REDUCE(min, PLX_AGG_MIN, FIELD_SAME) PLX_PLUGIN_REDUCE(max, PLX_AGG_MAX, FIELD_SAME)
void _polars_plugin_plx_filter(PLX_PLUGIN_SIG) {
  (void)kwargs; (void)kwargs_len;
  plugin_call(inputs, n_inputs, out, ctx, 2, [&](std::vector<PluginInput>& in) {
    PLX_REQUIRE(in[1].dtype == PLX_BOOL, PLX_ERR_INVALID, "plx_filter: the mask must be Boolean");
    ColumnPtr mask = in[1].col;
    if (in[1].is_scalar) { plx_scalar v = in[1].scalar; mask = ops::full_column(PLX_BOOL, v, in[1].scalar_valid, in[0].col->len); }
    return ops::filter(in[0].col, mask);
  });
}
void _polars_plugin_field_plx_filter(PLX_FIELD_SIG) { (void)kwargs; (void)kwargs_len; plugin_field(fields, n_fields, out, FIELD_SAME, 0); }

// ---- synthetic benchmark data --------------------------------------------------------
int plx_datagen_lineitem_q1(int64_t n_rows, uint64_t seed, plx_column* out_cols) {
  PLX_TRY
  PLX_REQUIRE(n_rows >= 0 && out_cols, PLX_ERR_INVALID, "datagen: bad arguments");
  static const int dts[7] = {PLX_I64, PLX_U8, PLX_U8, PLX_I64, PLX_F64, PLX_F64, PLX_F64};
  ColumnPtr c[7];
  for (int i = 0; i < 7; i++) { c[i] = make_column(dts[i], n_rows, false); c[i]->null_count = 0; }
  k::datagen_lineitem_q1(n_rows, seed, c[0]->values->as<int64_t>(), c[1]->values->as<uint8_t>(), c[2]->values->as<uint8_t>(), c[3]->values->as<int64_t>(),
                         c[4]->values->as<double>(), c[5]->values->as<double>(), c[6]->values->as<double>());
  for (int i = 0; i < 7; i++) out_cols[i] = register_column(c[i]);
  PLX_CATCH
}
int plx_datagen_lineitem_q1_host(int64_t row0, int64_t n, uint64_t seed, int64_t* shipdate, uint8_t* returnflag, uint8_t* linestatus, int64_t* quantity,
                                 double* extendedprice, double* discount, double* tax) {
  PLX_TRY
  PLX_REQUIRE(row0 >= 0 && n >= 0, PLX_ERR_INVALID, "datagen: bad arguments");
  for (int64_t j = 0; j < n; j++) {
    const datagen::LineitemRow r = datagen::lineitem_row(seed, (uint64_t)(row0 + j));
    if (shipdate) shipdate[j] = r.shipdate;
    if (returnflag) returnflag[j] = r.returnflag;
    if (linestatus) linestatus[j] = r.linestatus;
    if (quantity) quantity[j] = r.quantity;
    if (extendedprice) extendedprice[j] = r.extendedprice;
    if (discount) discount[j] = r.discount;
    if (tax) tax[j] = r.tax;
  }
  PLX_CATCH
}

int plx_datagen_orders_lineitem(int64_t n_orders, uint64_t seed, plx_column* out_orders, plx_column* out_lineitem) {
  PLX_TRY
  PLX_REQUIRE(n_orders >= 0 && out_orders && out_lineitem, PLX_ERR_INVALID, "datagen: bad arguments");
  const int64_t cust_hi = std::max<int64_t>(2, n_orders / 10) + 1;
  ColumnPtr o[4];
  for (auto& c : o) { c = make_column(PLX_I64, n_orders, false); c->null_count = 0; }
  Buf cnt = dev_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(n_orders, 1));
  Buf offsets = dev_alloc(sizeof(uint64_t) * (size_t)(n_orders + 1));
  k::datagen_orders(n_orders, seed, cust_hi, o[0]->values->as<int64_t>(), o[1]->values->as<int64_t>(), o[2]->values->as<int64_t>(), o[3]->values->as<int64_t>(), cnt->as<uint32_t>());
  k::exclusive_scan_u32(cnt->as<uint32_t>(), offsets->as<uint64_t>(), n_orders);
  uint64_t n_lines = 0;
  d2h_sync(&n_lines, offsets->as<uint64_t>() + n_orders, 8);
  static const int ldt[4] = {PLX_I64, PLX_F64, PLX_F64, PLX_I64};
  ColumnPtr l[4];
  for (int i = 0; i < 4; i++) { l[i] = make_column(ldt[i], (int64_t)n_lines, false); l[i]->null_count = 0; }
  k::datagen_lines(n_orders, seed, offsets->as<uint64_t>(), o[0]->values->as<int64_t>(), o[2]->values->as<int64_t>(), l[0]->values->as<int64_t>(), l[1]->values->as<double>(),
                   l[2]->values->as<double>(), l[3]->values->as<int64_t>());
  for (int i = 0; i < 4; i++) { out_orders[i] = register_column(o[i]); out_lineitem[i] = register_column(l[i]); }
  PLX_CATCH
}
int plx_datagen_orders_lineitem_host(int64_t order0, int64_t n, int64_t n_orders_total, uint64_t seed, int64_t* orderkey, int64_t* custkey, int64_t* orderdate,
                                     uint32_t* n_lines, int64_t line_cap, int64_t* l_orderkey, double* l_extendedprice, double* l_discount, int64_t* l_shipdate,
                                     int64_t* n_lines_out) {
  PLX_TRY
  PLX_REQUIRE(order0 >= 0 && n >= 0 && n_orders_total >= order0 + n, PLX_ERR_INVALID, "datagen: bad arguments");
  const int64_t cust_hi = std::max<int64_t>(2, n_orders_total / 10) + 1;
  int64_t at = 0;
  for (int64_t j = 0; j < n; j++) {
    const datagen::OrderRow r = datagen::order_row(seed, (uint64_t)(order0 + j), cust_hi);
    if (orderkey) orderkey[j] = r.orderkey;
    if (custkey) custkey[j] = r.custkey;
    if (orderdate) orderdate[j] = r.orderdate;
    if (n_lines) n_lines[j] = r.n_lines;
    const bool want_lines = l_orderkey || l_extendedprice || l_discount || l_shipdate;
    for (uint32_t q = 0; q < r.n_lines; q++, at++) {
      if (!want_lines) continue;
      PLX_REQUIRE(at < line_cap, PLX_ERR_INVALID, "datagen: line arrays too small");
      const datagen::Q3LineRow lr = datagen::q3_line_row(seed, (uint64_t)(order0 + j), q, r.orderdate);
      if (l_orderkey) l_orderkey[at] = r.orderkey;
      if (l_extendedprice) l_extendedprice[at] = lr.extendedprice;
      if (l_discount) l_discount[at] = lr.discount;
      if (l_shipdate) l_shipdate[at] = lr.shipdate;
    }
  }
  if (n_lines_out) *n_lines_out = at;
  PLX_CATCH
}
int plx_datagen_customer(int64_t n_customers, uint64_t seed, plx_column* out_cols) {
  PLX_TRY
  PLX_REQUIRE(n_customers >= 0 && out_cols, PLX_ERR_INVALID, "datagen: bad arguments");
  ColumnPtr k = make_column(PLX_I64, n_customers, false), sgm = make_column(PLX_U8, n_customers, false);
  k->null_count = 0; sgm->null_count = 0;
  k::datagen_customer(n_customers, seed, k->values->as<int64_t>(), sgm->values->as<uint8_t>());
  out_cols[0] = register_column(k); out_cols[1] = register_column(sgm);
  PLX_CATCH
}
int plx_datagen_customer_host(int64_t row0, int64_t n, uint64_t seed, int64_t* custkey, uint8_t* segment) {
  PLX_TRY
  PLX_REQUIRE(row0 >= 0 && n >= 0, PLX_ERR_INVALID, "datagen: bad arguments");
  for (int64_t j = 0; j < n; j++) {
    if (custkey) custkey[j] = row0 + j + 1;
    if (segment) segment[j] = datagen::customer_segment(seed, (uint64_t)(row0 + j + 1));
  }
  PLX_CATCH
}
int plx_datagen_uniform(int32_t dtype, int64_t n_rows, uint64_t seed, uint32_t stream_id, int64_t lo, int64_t hi, double scale, plx_column* out) {
  PLX_TRY
  PLX_REQUIRE(n_rows >= 0 && out && hi > lo && stream_id < 8, PLX_ERR_INVALID, "datagen: bad arguments");
  PLX_REQUIRE(dtype == PLX_I64 || dtype == PLX_U32 || dtype == PLX_F64, PLX_ERR_UNSUPPORTED, "datagen_uniform: dtype must be Int64, UInt32 or Float64");
  ColumnPtr c = make_column(dtype, n_rows, false); c->null_count = 0;
  k::datagen_uniform(dtype, n_rows, seed, stream_id, lo, hi, scale, c->values->ptr);
  *out = register_column(c);
  PLX_CATCH
}
int plx_datagen_uniform_host(int32_t dtype, int64_t row0, int64_t n, uint64_t seed, uint32_t stream_id, int64_t lo, int64_t hi, double scale, void* out) {
  PLX_TRY
  PLX_REQUIRE(row0 >= 0 && n >= 0 && (out || n == 0) && hi > lo && stream_id < 8, PLX_ERR_INVALID, "datagen: bad arguments");
  PLX_REQUIRE(dtype == PLX_I64 || dtype == PLX_U32 || dtype == PLX_F64, PLX_ERR_UNSUPPORTED, "datagen_uniform: dtype must be Int64, UInt32 or Float64");
  for (int64_t j = 0; j < n; j++) {
    const int64_t v = datagen::uniform_value(seed, stream_id, (uint64_t)(row0 + j), lo, hi);
    if (dtype == PLX_I64) reinterpret_cast<int64_t*>(out)[j] = v;
    else if (dtype == PLX_U32) reinterpret_cast<uint32_t*>(out)[j] = (uint32_t)v;
    else reinterpret_cast<double*>(out)[j] = (double)v * scale;
  }
  PLX_CATCH
}

int plx_datagen_zipf(int64_t n_rows, uint64_t seed, uint32_t stream_id, uint64_t x0_q62, int64_t n_keys, plx_column* out) {
  PLX_TRY
  PLX_REQUIRE(n_rows >= 0 && out && n_keys > 0 && stream_id < 8 && x0_q62 > 0 && x0_q62 < (1ull << 62), PLX_ERR_INVALID, "datagen_zipf: bad arguments");
  ColumnPtr c = make_column(PLX_I64, n_rows, false); c->null_count = 0;
  k::datagen_zipf(n_rows, seed, stream_id, x0_q62, n_keys, c->values->as<int64_t>());
  *out = register_column(c);
  PLX_CATCH
}
int plx_datagen_zipf_host(int64_t row0, int64_t n, uint64_t seed, uint32_t stream_id, uint64_t x0_q62, int64_t n_keys, int64_t* out) {
  PLX_TRY
  PLX_REQUIRE(row0 >= 0 && n >= 0 && (out || n == 0) && n_keys > 0 && stream_id < 8 && x0_q62 > 0 && x0_q62 < (1ull << 62), PLX_ERR_INVALID, "datagen_zipf: bad arguments");
  for (int64_t j = 0; j < n; j++) out[j] = datagen::zipf_value(seed, stream_id, (uint64_t)(row0 + j), x0_q62, n_keys);
  PLX_CATCH
}

// ---- plans -------------------------------------------------------------------------
int plx_execute_plan(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root, uint32_t flags, plx_frame* out) {
  PLX_TRY
  PLX_REQUIRE(out, PLX_ERR_INVALID, "null out pointer");
  engine::Plan p = engine::import_plan(ir, n_ir, exprs, n_exprs, flags);
  FramePtr f = engine::execute(p, root);
  t_plan_desc = p.desc;
  *out = register_frame(f);
  PLX_CATCH
}
const char* plx_last_plan_description(void) { return t_plan_desc.c_str(); }

int plx_debug_program_json(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root, char* buf, size_t cap) {
  PLX_TRY
  PLX_REQUIRE(buf && cap > 0, PLX_ERR_INVALID, "null buffer");
  engine::Plan p = engine::import_plan(ir, n_ir, exprs, n_exprs, 0);
  std::string json, why;
  if (!engine::dump_program_json(p, root, &json, &why)) fail(PLX_ERR_UNSUPPORTED, "not a fusable pipeline: " + why);
  PLX_REQUIRE(json.size() + 1 <= cap, PLX_ERR_INVALID, "buffer too small for the program dump (" + std::to_string(json.size() + 1) + " bytes)");
  memcpy(buf, json.c_str(), json.size() + 1);
  PLX_CATCH
}

int plx_describe_fusion(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root, int32_t* fusable, int32_t* static_shape_id,
                        char* why_not, size_t why_cap) {
  PLX_TRY
  engine::Plan p = engine::import_plan(ir, n_ir, exprs, n_exprs, 0);
  // dump of one program, so a mismatch with fused_shapes.hpp is easy to repair
  auto dump = [](const fused::Shape& sh) {
    std::string d = "inputs=" + std::to_string(sh.n_inputs) + " pred=" + std::to_string(sh.pred) + " key=" + std::to_string(sh.key) + " ops=[";
    for (int i = 0; i < sh.n_ops; i++) d += "(" + std::to_string(sh.ops[i].code) + "," + std::to_string(sh.ops[i].dst) + "," + std::to_string(sh.ops[i].a) + "," + std::to_string(sh.ops[i].b) + "," + std::to_string(sh.ops[i].c) + ")";
    d += "] aggs=[";
    for (int i = 0; i < sh.n_aggs; i++) d += "(" + std::to_string(sh.aggs[i].kind) + "," + std::to_string(sh.aggs[i].src) + ")";
    d += "] in_dtype=[";
    for (int i = 0; i < sh.n_inputs; i++) d += std::to_string(sh.in_dtype[i]) + (sh.in_nullable[i] ? "?" : "") + ",";
    d += "]";
    if (sh.n_keys) { d += " keys=["; for (int i = 0; i < sh.n_keys; i++) d += std::to_string(sh.keys[i]) + ","; d += "]"; }     // a wide (multi-column) key: one slot per key column
    return d;
  };
  std::string why;
  int sid = -1;
  bool ok;
  PLX_REQUIRE(root >= 0 && root < (int)p.ir.size(), PLX_ERR_INVALID, "bad root");
  const engine::IRN& rn = p.ir[root];
  if (rn.kind == PLX_IR_GROUPBY && rn.input >= 0 && p.ir[rn.input].kind == PLX_IR_JOIN) {
    // fused join -> aggregate: three programs (count, build, probe), one per line; the id is the probe program's
    std::vector<fused::Shape> shapes;
    ok = engine::describe_join_fusion(p, root, &shapes, &why);
    if (ok) {   // count, build, probe, then one program per nested filter join (semi filter)
      sid = fused::find_static_shape(shapes[2]);
      t_plan_desc = dump(shapes[0]);
      for (size_t i = 1; i < shapes.size(); i++) t_plan_desc += "\n" + dump(shapes[i]);
    }
  } else {
    fused::Shape sh{};
    ok = engine::describe_fusion(p, root, &sh, &sid, &why);
    if (ok) t_plan_desc = dump(sh);
  }
  if (fusable) *fusable = ok ? 1 : 0;
  if (static_shape_id) *static_shape_id = sid;
  if (why_not && why_cap) snprintf(why_not, why_cap, "%s", why.c_str());
  PLX_CATCH
}

int plx_jit_selftest(const plx_ir* ir, int32_t n_ir, const plx_aexpr* exprs, int32_t n_exprs, int32_t root) {
  PLX_TRY
  engine::Plan p = engine::import_plan(ir, n_ir, exprs, n_exprs, 0);
  PLX_REQUIRE(root >= 0 && root < (int)p.ir.size(), PLX_ERR_INVALID, "bad root");
  const engine::IRN& rn = p.ir[root];
  std::string why;
  std::vector<std::pair<fused::Shape, jit::Sink>> jobs;
  if (rn.kind == PLX_IR_GROUPBY && rn.input >= 0 && p.ir[rn.input].kind == PLX_IR_JOIN) {
    std::vector<fused::Shape> shapes;
    PLX_REQUIRE(engine::describe_join_fusion(p, root, &shapes, &why), PLX_ERR_UNSUPPORTED, "not fusable: " + why);
    jobs = {{shapes[0], jit::REGAGG}, {shapes[1], jit::JOIN_BUILD}, {shapes[1], jit::DIRECT_BUILD}, {shapes[2], jit::PROBE_AGG}, {shapes[2], jit::DIRECT_PROBE}};
    for (size_t i = 3; i + 1 < shapes.size(); i++) jobs.push_back({shapes[i], jit::BITMAP_BUILD});
    for (uint32_t tiles : {2u, 4u}) jobs.push_back({shapes.back(), jit::part3_scatter_sink(fused::kP2Direct, tiles, fused::kPackRowid, false)});   // the partitioned probe's scatter
  } else if (rn.kind == PLX_IR_FILTER) {
    fused::Shape sh{};
    PLX_REQUIRE(engine::describe_filter_fusion(p, root, &sh, &why), PLX_ERR_UNSUPPORTED, "not fusable: " + why);
    jobs = {{sh, jit::BALLOT}};
  } else {
    fused::Shape sh{}; int sid = -1;
    PLX_REQUIRE(engine::describe_fusion(p, root, &sh, &sid, &why), PLX_ERR_UNSUPPORTED, "not fusable: " + why);
    if (rn.kind == PLX_IR_SELECT) jobs = {{sh, jit::REGAGG}};
    else jobs = {{sh, jit::LDSAGG}, {sh, jit::DENSE}, {sh, jit::HASH}, {sh, jit::PART_COUNT}, {sh, jit::PART_SCATTER}, {sh, jit::PART_AGG},
                  {sh, jit::PART2_SCATTER_HASH}, {sh, jit::PART2_SCATTER_DIRECT}, {sh, jit::PART2_AGG_HASH}, {sh, jit::PART2_AGG_DIRECT},
                  {sh, jit::PART2_SCATTER_HASH_T2}, {sh, jit::PART2_SCATTER_DIRECT_T2}};
    if (rn.kind != PLX_IR_SELECT && sh.n_keys < 2) {
      // third generation scatter: hash / direct x 1 / 2 / 4 tiles, and every packing the shape admits, with the matching aggregation kernels
      for (uint32_t mode = 0; mode < 2; mode++) {
        const uint32_t best = fused::best_static_pack(sh, mode);
        for (uint32_t pack = 0; pack <= best; pack++) {
          if (pack == fused::kPackNarrow && fused::rec_layout2(sh, mode, fused::kPackNarrow).rec_words == fused::rec_layout2(sh, mode, fused::kPackNone).rec_words) continue;
          for (uint32_t tiles : {1u, 2u, 3u, 4u}) { jobs.push_back({sh, jit::part3_scatter_sink(mode, tiles, pack, false)}); jobs.push_back({sh, jit::part3_scatter_sink(mode, tiles, pack, true)}); }
          jobs.push_back({sh, jit::part3_agg_sink(mode, pack)});
        }
      }
      for (uint32_t mode = 0; mode < 2; mode++) {            // two rows a record (one 64-bit value that does not narrow; direct-address slots or 48-bit key offsets)
        if (!fused::pair_pack_ok(sh, mode)) continue;
        for (uint32_t tiles : {1u, 2u, 3u, 4u}) { if (mode == fused::kP2Hash && tiles > 3) continue; jobs.push_back({sh, jit::part3_scatter_sink(mode, tiles, fused::kPackPair, false)}); jobs.push_back({sh, jit::part3_scatter_sink(mode, tiles, fused::kPackPair, true)}); }
        jobs.push_back({sh, jit::part3_agg_sink(mode, fused::kPackPair)});
      }
      if (fused::pairv_pack_ok(sh, fused::kP2Hash)) {        // ... with the value as a 48-bit offset
        for (uint32_t tiles : {1u, 2u, 3u}) { jobs.push_back({sh, jit::part3_scatter_sink(fused::kP2Hash, tiles, fused::kPackPairV, false)}); jobs.push_back({sh, jit::part3_scatter_sink(fused::kP2Hash, tiles, fused::kPackPairV, true)}); }
        jobs.push_back({sh, jit::part3_agg_sink(fused::kP2Hash, fused::kPackPairV)});
      }
    }
    if (sh.n_keys >= 2) {
      // wide key: the HBM table sink, and the partitioned path -- hash partitions, plain or narrowed records, no hot keys
      jobs = {{sh, jit::WIDE}};
      for (uint32_t pack = 0; pack <= std::min<uint32_t>(fused::best_static_pack(sh, fused::kP2Hash), fused::kPackNarrow); pack++) {
        for (uint32_t tiles : {1u, 2u, 3u}) jobs.push_back({sh, jit::part3_scatter_sink(fused::kP2Hash, tiles, pack, false)});
        jobs.push_back({sh, jit::part3_agg_sink(fused::kP2Hash, pack)});
      }
    }
  }
  for (auto& j : jobs) {
    const std::string log = jit::selftest(j.first, j.second);
    if (!log.empty()) fail(PLX_ERR_INVALID, "jit selftest failed: " + log.substr(0, 4000));
  }
  PLX_CATCH
}
int plx_jit_set_min_rows(int64_t min_rows) { PLX_TRY jit::set_min_rows(min_rows); PLX_CATCH }
int plx_jit_stats(int32_t* compiled, double* compile_ms) {
  PLX_TRY
  int c = 0; double ms = 0;
  jit::stats(&c, &ms);
  if (compiled) *compiled = c;
  if (compile_ms) *compile_ms = ms;
  PLX_CATCH
}

// ---- raw string keys: device-side dictionary (kernels_strview.hip) ------------------------------------
extern "C++" {
namespace {
struct StrDict { Buf views, data, offsets, bytes; int64_t n = 0; uint64_t total = 0; bool materialised = false; };
std::mutex g_strdict_mu;
std::vector<std::unique_ptr<StrDict>> g_strdicts;     // handle = index + 1
StrDict& get_strdict(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_strdict_mu);
  if (h == 0 || h > g_strdicts.size() || !g_strdicts[h - 1]) fail(PLX_ERR_INVALID, "invalid string dictionary handle");
  return *g_strdicts[h - 1];
}
uint64_t register_strdict(std::unique_ptr<StrDict> d) { std::lock_guard<std::mutex> lk(g_strdict_mu); g_strdicts.push_back(std::move(d)); return (uint64_t)g_strdicts.size(); }
void materialise(StrDict& d) {
  if (d.materialised) return;
  k::strdict_materialise(d.views->as<uint64_t>(), d.data ? d.data->as<uint8_t>() : nullptr, d.n, &d.offsets, &d.bytes, &d.total);
  d.materialised = true;
}
// shared tail of both entry points: views / validity / data are on the device
// from_stamps: the column has no bitmap, its nulls are stamped views (the raw-view interfaces) -- the encoder reads the bitmap off the stamps in its own pass
void encode_on_device(const uint64_t* views, const ColumnPtr& validity_holder, Buf data, Buf buf_base, int64_t n, plx_column* out_codes, plx_strdict* out_dict, bool from_stamps = false) {
  Buf codes, dviews, svalid;
  int64_t nd = 0, snulls = 0;
  k::strview_dict_encode(views, validity_holder ? validity_holder->valid_words() : nullptr, data ? data->as<uint8_t>() : nullptr, buf_base ? buf_base->as<uint64_t>() : nullptr, n, &codes, &dviews, &nd,
                         from_stamps && !validity_holder ? &svalid : nullptr, &snulls);
  auto c = std::make_shared<Column>();
  c->dtype = PLX_U32; c->len = n; c->values = codes;
  if (validity_holder && validity_holder->validity) c->validity = validity_holder->validity;
  else if (snulls > 0) { c->validity = svalid; c->null_count = snulls; }
  else c->null_count = 0;
  if (nd > 0) { c->range_state = 1; c->range_min = 0; c->range_max = nd - 1; c->range_trusted = true; }   // codes are dense by construction
  auto d = std::make_unique<StrDict>();
  d->views = dviews; d->data = data; d->n = nd;
  *out_codes = register_column(c);
  *out_dict = register_strdict(std::move(d));
}
}  // namespace
namespace plx {
// views (and the long strings' bytes, one buffer, absolute offsets) already in HBM -> dictionary codes + dictionary handle; the scan sources
// that build their views on the device (ipc.cpp) end here
void strview_encode_device(const uint64_t* views, const ColumnPtr& validity_holder, Buf data, int64_t n, plx_column* out_codes, plx_strdict* out_dict) {
  Buf bb = dev_alloc_zero(8);
  encode_on_device(views, validity_holder, data, bb, n, out_codes, out_dict);
}
// ... with the long strings' bytes in several concatenated buffers: `buf_base` = the device array of their start offsets inside `data` (what plx_strview_dict_encode
// builds from its host buffers; the Parquet reader's host-string path brings everything to the device itself: parquet.cpp)
void strview_encode_device_bases(const uint64_t* views, const ColumnPtr& validity_holder, Buf data, Buf buf_base, int64_t n, plx_column* out_codes, plx_strdict* out_dict) {
  encode_on_device(views, validity_holder, data, buf_base, n, out_codes, out_dict);
}
}  // namespace plx
}  // extern "C++"

int plx_strview_dict_encode(const void* views, const uint8_t* validity, int64_t bit_offset, int64_t n, const void* const* data_buffers, const int64_t* data_sizes,
                            int32_t n_data_buffers, plx_column* out_codes, plx_strdict* out_dict) {
  PLX_TRY
  PLX_REQUIRE(out_codes && out_dict && n >= 0 && (views || n == 0) && n_data_buffers >= 0 && (n_data_buffers == 0 || (data_buffers && data_sizes)), PLX_ERR_INVALID, "strview_dict_encode: bad arguments");
  device();
  Buf dv = dev_alloc((size_t)std::max<int64_t>(n, 1) * 16);
  if (n) h2d_sync_pinned(dv->ptr, views, (size_t)n * 16);
  // validity rides on a throw-away byte column so that bit offsets are normalised by the usual import path
  ColumnPtr vh;
  if (validity) {
    std::vector<uint8_t> zeros((size_t)n);
    vh = column_from_host(PLX_U8, zeros.data(), validity, bit_offset, n);
  }
  uint64_t total = 0;
  std::vector<uint64_t> base((size_t)std::max(n_data_buffers, 1), 0);
  for (int i = 0; i < n_data_buffers; i++) { base[i] = total; total += (uint64_t)data_sizes[i]; }
  Buf data = dev_alloc((size_t)total + 64), bb = dev_alloc(sizeof(uint64_t) * base.size());
  for (int i = 0; i < n_data_buffers; i++) {
    if (!data_sizes[i]) continue;
    // large buffers are page-locked for their copy (h2d_sync_pinned); the many small ones of a page-per-buffer source are queued and waited for once, below
    if ((uint64_t)data_sizes[i] >= ((uint64_t)32 << 20)) h2d_sync_pinned((uint8_t*)data->ptr + base[i], data_buffers[i], (size_t)data_sizes[i]);
    else h2d_async((uint8_t*)data->ptr + base[i], data_buffers[i], (size_t)data_sizes[i]);
  }
  h2d_async(bb->ptr, base.data(), sizeof(uint64_t) * base.size());
  PLX_HIP(hipStreamSynchronize(stream()));
  encode_on_device(dv->as<uint64_t>(), vh, data, bb, n, out_codes, out_dict);
  PLX_CATCH
}
int plx_strview_dict_encode_device(plx_column views_u64_pairs, plx_column data_u8, plx_column* out_codes, plx_strdict* out_dict) {
  PLX_TRY
  PLX_REQUIRE(out_codes && out_dict, PLX_ERR_INVALID, "null pointer");
  ColumnPtr v = get_column(views_u64_pairs);
  PLX_REQUIRE(v->dtype == PLX_U64 && v->len % 2 == 0 && (v->values || v->len == 0), PLX_ERR_INVALID, "views must be a UInt64 column of 2 n words");
  Buf data, bb = dev_alloc_zero(8);
  if (data_u8) { ColumnPtr d = get_column(data_u8); PLX_REQUIRE(d->dtype == PLX_U8, PLX_ERR_INVALID, "data must be a UInt8 column"); data = d->values; }
  // nulls arrive as stamped views (plx_strview_stamp_nulls / plx_ipc_read_string_views): the bitmap the encoder and the code column want is read off the stamps
  const int64_t n = v->len / 2;
  encode_on_device(v->values ? v->values->as<uint64_t>() : nullptr, nullptr, data, bb, n, out_codes, out_dict, /*from_stamps=*/true);
  // the dictionary's views point into nothing else than `data`; inline strings need no buffer at all
  PLX_CATCH
}
int plx_strview_stamp_nulls(plx_column views_u64_pairs, plx_column valid_bool) {
  PLX_TRY
  ColumnPtr v = get_column(views_u64_pairs), m = get_column(valid_bool);
  PLX_REQUIRE(v->dtype == PLX_U64 && v->len % 2 == 0, PLX_ERR_INVALID, "views must be a UInt64 column of 2 n words");
  PLX_REQUIRE(m->dtype == PLX_BOOL, PLX_ERR_INVALID, "the validity of the views is a Boolean column (true = valid: the array's validity bitmap as its values)");
  PLX_REQUIRE(m->len == v->len / 2, PLX_ERR_SHAPE, "validity column and views differ in length");
  PLX_REQUIRE((v->values && m->values) || v->len == 0, PLX_ERR_INVALID, "placeholder column has no data");
  if (!v->len) return PLX_OK;
  // a null entry of the Boolean column itself means "not valid" (its value bit is whatever was stored): valid = value AND validity
  ColumnPtr eff = m;
  if (m->validity) {
    auto a = std::make_shared<Column>(); a->dtype = PLX_BOOL; a->len = m->len; a->values = m->values; a->null_count = 0;
    auto b = std::make_shared<Column>(); b->dtype = PLX_BOOL; b->len = m->len; b->values = m->validity; b->null_count = 0;
    eff = ops::bool_binop(PLX_AND, a, b);
  }
  // the stamps are written INTO the view buffer: a buffer the caller lent (plx_column_from_device) or one that other columns share is copied first, so that nobody
  // else's views change under them
  if (!v->values->owned || v->values.use_count() > 1) {
    Buf copy = dev_alloc(v->values->bytes);
    PLX_HIP(hipMemcpyAsync(copy->ptr, v->values->ptr, v->values->bytes, hipMemcpyDeviceToDevice, stream()));
    v->values = copy;
  }
  k::strview_stamp_nulls(v->values->as<uint64_t>(), eff->values->as<uint64_t>(), m->len);
  PLX_HIP(hipStreamSynchronize(stream()));      // (`eff` may be a temporary)
  PLX_CATCH
}
int plx_strview_groupby(plx_column views_u64_pairs, plx_column value, plx_column* out_codes, plx_strdict* out_dict, plx_column* out_sum, plx_column* out_count, plx_column* out_len) {
  PLX_TRY
  PLX_REQUIRE(out_codes && out_dict && out_sum && out_count && out_len, PLX_ERR_INVALID, "null pointer");
  ColumnPtr v = get_column(views_u64_pairs), x = get_column(value);
  PLX_REQUIRE(v->dtype == PLX_U64 && v->len % 2 == 0, PLX_ERR_INVALID, "views must be a UInt64 column of 2 n words");
  const int64_t n = v->len / 2;
  PLX_REQUIRE(x->len == n, PLX_ERR_SHAPE, "value column and views differ in length");
  if (n == 0 || (x->dtype != PLX_F64 && x->dtype != PLX_I64)) fail(PLX_ERR_UNSUPPORTED, "string group-by fast path: one Float64 / Int64 value column over a non-empty key");
  PLX_REQUIRE(v->values && x->values, PLX_ERR_INVALID, "placeholder column has no data");
  Buf gviews, gsum, gcnt, glen;
  std::string desc;
  const int64_t G = k::strview_groupby(v->values->as<uint64_t>(), x->values->as<uint64_t>(), x->validity ? x->valid_words() : nullptr, n, x->dtype == PLX_F64, &gviews, &gsum, &gcnt, &glen, &desc);
  if (G < 0) fail(PLX_ERR_UNSUPPORTED, "string group-by fast path not applicable (a string longer than 12 bytes, or more distinct strings than its LDS tables hold)");
  auto col = [&](int dtype, const Buf& b) { auto c = std::make_shared<Column>(); c->dtype = dtype; c->len = G; c->values = b; c->null_count = 0; return c; };
  ColumnPtr codes = make_column(PLX_U32, G, false);
  codes->null_count = 0;
  if (G) { k::fill_iota_u32(codes->values->as<uint32_t>(), G); codes->range_state = 1; codes->range_min = 0; codes->range_max = G - 1; codes->range_trusted = true; }
  if (G) {      // the null key's group (stamped views): its code is null, its dictionary entry the empty string
    Buf bits = dev_alloc(bitmap_bytes(G));
    const int64_t nulls = k::strview_null_group(gviews->as<uint64_t>(), G, bits->as<uint64_t>());
    if (nulls > 0) { codes->validity = bits; codes->null_count = nulls; }
  }
  auto d = std::make_unique<StrDict>();
  d->views = gviews; d->data = nullptr; d->n = G;
  *out_codes = register_column(codes);
  *out_dict = register_strdict(std::move(d));
  *out_sum = register_column(col(x->dtype, gsum));
  *out_count = register_column(col(PLX_U32, gcnt));
  *out_len = register_column(col(PLX_U32, glen));
  t_plan_desc = "StringViewGroupBy{" + desc + ", rows=" + std::to_string(n) + ", groups=" + std::to_string(G) + "}; ";
  PLX_CATCH
}
int plx_strdict_info(plx_strdict dict, int64_t* n_strings, int64_t* total_bytes) {
  PLX_TRY
  StrDict& d = get_strdict(dict);
  if (total_bytes) materialise(d);
  if (n_strings) *n_strings = d.n;
  if (total_bytes) *total_bytes = (int64_t)d.total;
  PLX_CATCH
}
int plx_strdict_to_host(plx_strdict dict, int64_t* offsets, uint8_t* bytes) {
  PLX_TRY
  StrDict& d = get_strdict(dict);
  materialise(d);
  if (offsets) d2h_sync(offsets, d.offsets->ptr, sizeof(int64_t) * (size_t)(d.n + 1));
  if (bytes && d.total) d2h_sync(bytes, d.bytes->ptr, (size_t)d.total);
  PLX_CATCH
}
int plx_strdict_free(plx_strdict dict) {
  PLX_TRY
  std::lock_guard<std::mutex> lk(g_strdict_mu);
  if (dict && dict <= g_strdicts.size()) g_strdicts[dict - 1].reset();
  PLX_CATCH
}
int plx_datagen_id_views(int64_t n_rows, uint64_t seed, uint32_t stream_id, int64_t lo, int64_t hi, plx_column* out_views) {
  PLX_TRY
  PLX_REQUIRE(n_rows >= 0 && out_views && hi > lo && lo >= 0 && hi <= 10000000000ll && stream_id < 8, PLX_ERR_INVALID, "datagen_id_views: bad arguments");
  ColumnPtr c = make_column(PLX_U64, n_rows * 2, false); c->null_count = 0;
  k::datagen_id_views(n_rows, seed, stream_id, lo, hi, c->values->as<uint64_t>());
  *out_views = register_column(c);
  PLX_CATCH
}

int plx_datagen_long_id_views(int64_t n_rows, uint64_t seed, uint32_t stream_id, int64_t lo, int64_t hi, plx_column* out_views, plx_column* out_data) {
  PLX_TRY
  PLX_REQUIRE(n_rows >= 0 && out_views && out_data && hi > lo && lo >= 0 && hi <= 10000000000ll && (hi - lo) * 20 < ((int64_t)1 << 32) && stream_id < 8, PLX_ERR_INVALID, "datagen_long_id_views: bad arguments");
  ColumnPtr c = make_column(PLX_U64, n_rows * 2, false); c->null_count = 0;
  ColumnPtr d = make_column(PLX_U8, (hi - lo) * 20, false); d->null_count = 0;
  k::datagen_long_id_views(n_rows, seed, stream_id, lo, hi, c->values->as<uint64_t>(), d->values->as<uint8_t>());
  *out_views = register_column(c);
  *out_data = register_column(d);
  PLX_CATCH
}

// ---- multi-GPU exchange (comm.cpp) -----------------------------------------------------
int plx_comm_unique_id(uint8_t* out) { PLX_TRY PLX_REQUIRE(out, PLX_ERR_INVALID, "null pointer"); comm::unique_id(out); PLX_CATCH }
int plx_comm_init(const uint8_t* id, int32_t rank, int32_t world_size, plx_comm* out) { PLX_TRY PLX_REQUIRE(out, PLX_ERR_INVALID, "null pointer"); *out = comm::init(id, rank, world_size); PLX_CATCH }
int plx_comm_info(plx_comm c, int32_t* rank, int32_t* world_size) { PLX_TRY int r = 0, w = 0; comm::info(c, &r, &w); if (rank) *rank = r; if (world_size) *world_size = w; PLX_CATCH }
int plx_comm_free(plx_comm c) { PLX_TRY comm::destroy(c); PLX_CATCH }
int plx_exchange_by_key(plx_comm c, plx_frame frame, const char* key, uint64_t seed, plx_frame* out, uint64_t* rows_sent, uint64_t* bytes_sent) {
  PLX_TRY
  PLX_REQUIRE(key && out, PLX_ERR_INVALID, "null pointer");
  *out = register_frame(comm::exchange_by_key(c, get_frame(frame), key, seed, rows_sent, bytes_sent));
  PLX_CATCH
}
int plx_allgather_frame(plx_comm c, plx_frame frame, plx_frame* out) {
  PLX_TRY
  PLX_REQUIRE(out, PLX_ERR_INVALID, "null pointer");
  *out = register_frame(comm::allgather_frame(c, get_frame(frame)));
  PLX_CATCH
}

// ---- tracing -------------------------------------------------------------------------
int plx_profile_enable(int on) { PLX_TRY device(); profile_enable(on != 0); PLX_CATCH }
int plx_profile_fetch(plx_profile_record* out, int32_t cap, int32_t* n) {
  PLX_TRY
  int c = profile_fetch(out, cap);
  if (n) *n = c;
  PLX_CATCH
}
int plx_profile_clear(void) { PLX_TRY profile_clear(); PLX_CATCH }

}  // extern "C"
