// core.hpp -- host-side runtime of the MI355X backend: errors, device context,
// stream, HBM pool allocator, device columns / frames and the handle tables the
// C ABI hands out.  Reference counterparts: polars-buffer (Buffer<T>/SharedStorage),
// polars-arrow Bitmap (bitmap/immutable.rs:56-68), polars-core Column/DataFrame,
// polars-expr ExecutionState (state/execution_state.rs:184-207).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/polars_amd.h"

namespace plx {

// ---------------------------------------------------------------- errors ----
struct Error : std::exception {
  int code;
  std::string msg;
  Error(int c, std::string m) : code(c), msg(std::move(m)) {}
  const char* what() const noexcept override { return msg.c_str(); }
};
[[noreturn]] inline void fail(int code, const std::string& m) { throw Error(code, m); }
void set_last_error(const std::string& m);

#define PLX_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      ::plx::fail(PLX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)
#define PLX_REQUIRE(cond, code, msg) \
  do { if (!(cond)) ::plx::fail(code, msg); } while (0)

// ---------------------------------------------------------------- dtypes ----
inline int dtype_width(int dt) {
  switch (dt) {
    case PLX_I8: case PLX_U8: return 1;
    case PLX_I16: case PLX_U16: return 2;
    case PLX_I32: case PLX_U32: case PLX_F32: return 4;
    case PLX_I64: case PLX_U64: case PLX_F64: return 8;
    default: return 0;  // PLX_BOOL: bit-packed
  }
}
inline bool dtype_is_float(int dt) { return dt == PLX_F32 || dt == PLX_F64; }
inline bool dtype_is_signed(int dt) { return dt >= PLX_I8 && dt <= PLX_I64; }
inline bool dtype_is_unsigned(int dt) { return dt >= PLX_U8 && dt <= PLX_U64; }
inline bool dtype_is_int(int dt) { return dt >= PLX_I8 && dt <= PLX_U64; }
const char* dtype_name(int dt);
// bytes of a bit-packed buffer for n rows, padded so kernels may read whole u64 words
inline size_t bitmap_bytes(int64_t n) { return (size_t)(((n + 63) / 64) * 8 + 8); }
inline size_t values_bytes(int dt, int64_t n) { return dt == PLX_BOOL ? bitmap_bytes(n) : (size_t)n * dtype_width(dt); }

// ---------------------------------------------------------------- device ----
struct Device {
  int ordinal = -1;
  int cu_count = 256;
  uint64_t hbm_bytes = 0;
  std::string name;
  hipStream_t own_stream = nullptr;
  std::atomic<int> cancel{0};
};
Device& device();          // throws if plx_init has not bound a GPU
bool device_ready();
hipStream_t stream();      // stream of the calling thread (plx_set_stream) or the library stream
void set_thread_stream(hipStream_t s);
void check_cancel();       // ExecutionState::should_stop equivalent

// ------------------------------------------------------------ HBM pool ------
// Size-class caching allocator on top of hipMalloc.  All work of a process is issued
// in stream order on one stream at a time, so a block returned to the pool may be
// handed out again immediately (the next kernel touching it is ordered after the
// last one that used it).  Sized for 288 GB: blocks >= 1 GiB are not cached.
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;     // usable bytes requested
  size_t cap = 0;       // size class
  bool owned = true;    // false: borrowed from the caller (plx_column_from_device)
  ~DevBuf();
  template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
};
using Buf = std::shared_ptr<DevBuf>;
Buf dev_alloc(size_t bytes);               // uninitialised, 256-B aligned, padded by >= 64 B
Buf dev_alloc_zero(size_t bytes);          // hipMemsetAsync 0 on the current stream
Buf dev_alloc_transient(size_t bytes);     // a query's big scratch buffer (record pools): any cached block that holds it is taken, not only one within +25 %
void pool_reserve(size_t bytes);           // make sure the cache holds a free block of at least `bytes`
Buf dev_borrow(void* p, size_t bytes);
void pool_stats(uint64_t* in_use, uint64_t* high_water);
void pool_trim(bool force = false);   // force: the reserved block (pool_reserve) is released as well

// --------------------------------------------------------------- columns ----
struct Column {
  int dtype = PLX_I64;
  int64_t len = 0;
  Buf values;     // PLX_BOOL: bitmap
  Buf validity;   // null => no nulls
  int64_t null_count = -1;  // -1 unknown (computed lazily)
  // cached statistics of integer columns (zone-map style), filled lazily by ops::int_range
  int range_state = 0;      // 0 unknown, 1 known, 2 no valid rows
  int64_t range_min = 0, range_max = 0;
  bool range_trusted = true;   // computed by the library (exact); false: caller-provided bounds (plx_column_set_bounds) or bounds the planner ASSUMED from a sample
  bool range_assumed = false;  // range_min / range_max are the planner's guess (a strided sample + slack; engine.cpp assume_range): used like declared bounds -- every kernel
                               // that addresses a table or narrows a value with them checks each row -- and when a row falls outside them the query is planned again from an exact pass
  bool range_verified = false; // the assumed bounds have since been checked against EVERY row (a scan without a predicate that narrowed this column): valid, if not tight
  bool no_assume = false;      // a guess about this column was wrong once: exact statistics only
  // what the group-by planner learned from its strided sample of this column AS A KEY (engine.cpp KeySample: heavy hitters, distinct count,
  // group estimate): a column is immutable, so the next group-by on it with no predicate skips the 8 sample launches (0.3 ms per query)
  std::shared_ptr<void> key_sample;
  int order_state = 0;         // 0 unknown, 1 (roughly) ascending, 2 unordered: sampled once when the column is the probe key of a large join (k::sample_sortedness)
  bool repeats_as_build_key = false;   // learned by a join that built on this column: some key occurs more than once (the next join skips the unique-key attempt; a fact about the column, whatever the predicate was)
  const void* data() const { return values ? values->ptr : nullptr; }
  const uint64_t* valid_words() const { return validity ? validity->as<uint64_t>() : nullptr; }
};
using ColumnPtr = std::shared_ptr<Column>;

struct Frame {
  std::vector<std::string> names;
  std::vector<ColumnPtr> cols;
  int64_t height = 0;
  int find(const std::string& n) const {
    for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
    return -1;
  }
};
using FramePtr = std::shared_ptr<Frame>;

ColumnPtr make_column(int dtype, int64_t len, bool with_validity);
ColumnPtr column_from_host(int dtype, const void* values, const uint8_t* validity, int64_t bit_offset, int64_t len);
int64_t column_null_count(const ColumnPtr& c);
void column_to_host(const ColumnPtr& c, void* values_out, uint8_t* validity_out, int32_t* has_validity);

// handle tables
plx_column register_column(ColumnPtr c);
ColumnPtr get_column(plx_column h);
void retain_column(plx_column h);
void free_column(plx_column h);
plx_frame register_frame(FramePtr f);
FramePtr get_frame(plx_frame h);
void free_frame(plx_frame h);

// -------------------------------------------------------------- profiling ---
struct ProfileScope {
  int idx = -1;
  ProfileScope(const char* name, uint64_t algo_bytes, uint64_t rows);
  ~ProfileScope();
};
void profile_enable(bool on);
int profile_fetch(plx_profile_record* out, int cap);
void profile_clear();

// small host<->device helpers (async on the current stream + sync where noted)
void d2h_sync(void* dst, const void* src, size_t bytes);   // small copies go through a page-locked bounce buffer (one DMA, no staged pageable path)
// library-owned page-locked host buffer of kBounceBytes (nullptr if it could not be allocated); one user at a time (the library stream)
constexpr size_t kBounceBytes = size_t(1) << 20;
void* pinned_bounce();
std::mutex& bounce_mutex();
void h2d_async(void* dst, const void* src, size_t bytes);  // src must stay alive until sync
void h2d_sync_pinned(void* dst, const void* src, size_t bytes);  // large buffers: page-locked in place, one DMA; synchronises

}  // namespace plx
