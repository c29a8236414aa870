// core.cpp -- device context, HBM pool, columns, handle tables, tracing.
#include "core.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace plx {

// ---------------------------------------------------------------- errors ----
static thread_local std::string t_last_error;
void set_last_error(const std::string& m) { t_last_error = m; }
const std::string& last_error_ref() { return t_last_error; }

const char* dtype_name(int dt) {
  static const char* n[] = {"bool", "i8", "i16", "i32", "i64", "u8", "u16", "u32", "u64", "f32", "f64"};
  return (dt >= 0 && dt <= PLX_F64) ? n[dt] : "?";
}

// ---------------------------------------------------------------- device ----
static Device g_dev;
static std::mutex g_dev_mu;
static thread_local hipStream_t t_stream = nullptr;
static thread_local bool t_stream_set = false;

bool device_ready() { return g_dev.ordinal >= 0; }
Device& device() {
  if (g_dev.ordinal < 0) fail(PLX_ERR_HIP, "plx_init() has not bound a GPU (no HIP device available?)");
  return g_dev;
}
void init_device(int ordinal) {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (g_dev.ordinal == ordinal) return;
  if (g_dev.ordinal >= 0) fail(PLX_ERR_INVALID, "plx_init: process already bound to device " + std::to_string(g_dev.ordinal) + " (one process per GPU)");
  // The Parquet reader decodes the columns of a read on up to six HIP streams (parquet.cpp ColumnWorkers).  The runtime multiplexes a process' streams onto
  // GPU_MAX_HW_QUEUES hardware queues -- four by default -- and kernels of streams that share a queue run one after the other: with eight, the Snappy read of the
  // 2e7-row file takes 27.6 instead of 35-36 ms.  Only effective when this is the process' first HIP call; a value the caller has set is left alone.
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) fail(PLX_ERR_HIP, std::string("plx_init: no HIP device: ") + hipGetErrorString(e));
  PLX_REQUIRE(ordinal >= 0 && ordinal < n, PLX_ERR_INVALID, "plx_init: bad device ordinal");
  PLX_HIP(hipSetDevice(ordinal));
  hipDeviceProp_t p;
  PLX_HIP(hipGetDeviceProperties(&p, ordinal));
  g_dev.cu_count = p.multiProcessorCount;
  g_dev.hbm_bytes = p.totalGlobalMem;
  g_dev.name = p.name;
  PLX_HIP(hipStreamCreateWithFlags(&g_dev.own_stream, hipStreamNonBlocking));
  g_dev.ordinal = ordinal;
}
hipStream_t stream() { return t_stream_set ? t_stream : device().own_stream; }
static std::atomic<bool> g_multi_stream{false};   // a caller stream was installed at least once: freed blocks carry an event
void set_thread_stream(hipStream_t s) {
  // drain the calling thread's old stream before switching (its pending frees are then safe on any stream)
  if (device_ready()) (void)hipStreamSynchronize(stream());
  if (s == nullptr) { t_stream_set = false; t_stream = nullptr; }
  else { t_stream_set = true; t_stream = s; g_multi_stream.store(true, std::memory_order_release); }
}
void check_cancel() {
  if (g_dev.cancel.load(std::memory_order_relaxed)) fail(PLX_ERR_CANCELLED, "query cancelled");
}

// ------------------------------------------------------------ HBM pool ------
namespace {
// A cached block remembers the stream its last owner released it on and (once more than one stream is in use: one stream
// per calling thread, include/polars_amd.h) an event recorded on that stream at release time.  Re-use on the SAME stream
// is ordered by the stream itself; re-use on another stream first waits for the event, so a block is never handed to
// stream B while kernels the releasing thread submitted on stream A can still touch it.
struct FreeBlock { void* ptr; hipStream_t stream; hipEvent_t ev; };
struct Pool {
  std::mutex mu;
  std::multimap<size_t, FreeBlock> free_blocks;  // cap -> block
  uint64_t in_use = 0, high = 0, cached = 0;
  size_t reserve = 0;     // plx_memory_reserve: trims keep one cached block of at least this size
} g_pool;

size_t size_class(size_t bytes) {
  bytes += 64;  // tail pad so vector loads of the last partial vector stay in-bounds
  if (bytes <= 4096) return 4096;
  if (bytes <= (size_t(1) << 21)) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
  const size_t g = size_t(1) << 21;  // 2 MiB granules above
  return (bytes + g - 1) / g * g;
}
}  // namespace

DevBuf::~DevBuf() {
  if (!owned || !ptr) return;
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.in_use -= cap;
  // Every block is cached, multi-GB ones included: hipMalloc / hipFree of a filtered SF100 column
  // costs tens of milliseconds (page-table work + an implicit device sync), far more than the
  // kernels that fill it, and 288 GB of HBM leaves room.  dev_alloc trims the cache on OOM and
  // when it grows past half of the device memory.
  FreeBlock fb{ptr, nullptr, nullptr};
  if (device_ready()) {
    fb.stream = stream();
    if (g_multi_stream.load(std::memory_order_acquire)) {
      if (hipEventCreateWithFlags(&fb.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(fb.ev, fb.stream) != hipSuccess) {
        (void)hipGetLastError();
        if (fb.ev) { (void)hipEventDestroy(fb.ev); fb.ev = nullptr; }
        (void)hipStreamSynchronize(fb.stream);   // no event: make the release point a hard one
      }
    }
  }
  g_pool.free_blocks.emplace(cap, fb);
  g_pool.cached += cap;
}

static Buf dev_alloc_impl(size_t bytes, bool transient) {
  Device& dev = device();
  size_t cap = size_class(bytes);
  void* p = nullptr;
  bool over = false;
  FreeBlock reused{nullptr, nullptr, nullptr};
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.free_blocks.lower_bound(cap);
    // exact class below 2 MiB; best fit within +25% above (large blocks rarely repeat their exact size).  A TRANSIENT request (a query's record pool: gone when
    // the query returns) takes the smallest cached block that holds it up to FOUR times its size: mapping fresh memory costs ~30 ms per GB (hipMalloc), and the
    // block goes back to the cache in a moment.  Not beyond: a 64 MB record pool that checked out the 26 GB reserved block would leave the next large transient of
    // the same query to map fresh memory after all (and count 26 GB as in use).
    if (it != g_pool.free_blocks.end() && (it->first == cap || (cap > (size_t(1) << 21) && (it->first <= cap + cap / 4 || (transient && cap >= (size_t(64) << 20) && it->first / 4 <= cap))))) {
      reused = it->second; p = reused.ptr; cap = it->first; g_pool.free_blocks.erase(it); g_pool.cached -= cap;
    }
    over = g_pool.cached > dev.hbm_bytes / 2;
  }
  if (p && reused.stream != stream()) {
    // released on another stream: order this stream behind the release point (event), or drain the other stream when the
    // block was released before a second stream existed
    if (reused.ev) (void)hipStreamWaitEvent(stream(), reused.ev, 0);
    else if (reused.stream) (void)hipStreamSynchronize(reused.stream);
  }
  if (reused.ev) (void)hipEventDestroy(reused.ev);
  if (over) pool_trim();
  if (!p) {
    static const bool trace = getenv("PLX_POOL_TRACE") && getenv("PLX_POOL_TRACE")[0] == '1';      // measurement: every mapping of fresh memory, with its cost
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, cap);
    if (trace) fprintf(stderr, "[plx pool] hipMalloc %.1f MB%s: %.2f ms\n", cap / 1048576.0, transient ? " (transient)" : "", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess) {
      (void)hipGetLastError();
      pool_trim(true);       // out of memory: every cached block goes back to the driver, the reserved one (plx_memory_reserve) included
      e = hipMalloc(&p, cap);
      if (e != hipSuccess) fail(PLX_ERR_OOM, "hipMalloc(" + std::to_string(cap) + ") failed: " + hipGetErrorString(e));
    }
  }
  auto b = std::make_shared<DevBuf>();
  b->ptr = p; b->bytes = bytes; b->cap = cap; b->owned = true;
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.in_use += cap; g_pool.high = std::max(g_pool.high, g_pool.in_use);
  return b;
}
Buf dev_alloc(size_t bytes) { return dev_alloc_impl(bytes, false); }
Buf dev_alloc_transient(size_t bytes) { return dev_alloc_impl(bytes, true); }
// A free block of at least `bytes` in the pool's cache (mapped now, so that no query pays for it): what an engine does when it sizes its memory pool at start-up.
void pool_reserve(size_t bytes) {
  { std::lock_guard<std::mutex> lk(g_pool.mu); g_pool.reserve = bytes ? size_class(bytes) : 0; }
  if (bytes) { Buf b = dev_alloc_impl(bytes, true); }      // allocated (or found), released into the cache at once
}
Buf dev_alloc_zero(size_t bytes) {
  Buf b = dev_alloc(bytes);
  PLX_HIP(hipMemsetAsync(b->ptr, 0, b->cap, stream()));
  return b;
}
Buf dev_borrow(void* p, size_t bytes) {
  auto b = std::make_shared<DevBuf>();
  b->ptr = p; b->bytes = bytes; b->cap = bytes; b->owned = false;
  return b;
}
void pool_stats(uint64_t* in_use, uint64_t* high_water) {
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (in_use) *in_use = g_pool.in_use;
  if (high_water) *high_water = g_pool.high;
}
void pool_trim(bool force) {
  if (!device_ready()) return;
  (void)hipStreamSynchronize(stream());
  std::lock_guard<std::mutex> lk(g_pool.mu);
  // the reserved block (plx_memory_reserve) survives an ordinary trim: the smallest cached block that is at least that large.  `force` (the allocator ran out of
  // device memory) releases it too: idle cached memory must never be the reason a request fails.
  auto keep = (g_pool.reserve && !force) ? g_pool.free_blocks.lower_bound(g_pool.reserve) : g_pool.free_blocks.end();
  std::pair<size_t, FreeBlock> kept{0, FreeBlock{nullptr, nullptr, nullptr}};
  for (auto it = g_pool.free_blocks.begin(); it != g_pool.free_blocks.end(); ++it) {
    auto& kv = *it;
    if (kv.second.ev) { (void)hipEventSynchronize(kv.second.ev); (void)hipEventDestroy(kv.second.ev); kv.second.ev = nullptr; }
    else if (kv.second.stream && kv.second.stream != stream()) (void)hipStreamSynchronize(kv.second.stream);
    if (it == keep) { kept = {kv.first, kv.second}; continue; }
    (void)hipFree(kv.second.ptr);
  }
  g_pool.free_blocks.clear(); g_pool.cached = 0;
  if (kept.second.ptr) { g_pool.free_blocks.emplace(kept.first, kept.second); g_pool.cached = kept.first; }
}

// --------------------------------------------------------------- helpers ----
void* pinned_bounce() {
  static void* buf = [] { void* p = nullptr; if (hipHostMalloc(&p, kBounceBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } return p; }();
  return buf;
}
static std::mutex g_bounce_mu;
std::mutex& bounce_mutex() { return g_bounce_mu; }
void d2h_sync(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return;
  void* b = bytes <= 4096 ? pinned_bounce() : nullptr;
  if (b) {
    std::lock_guard<std::mutex> lk(g_bounce_mu);
    PLX_HIP(hipMemcpyAsync(b, src, bytes, hipMemcpyDeviceToHost, stream()));
    PLX_HIP(hipStreamSynchronize(stream()));
    memcpy(dst, b, bytes);
    return;
  }
  PLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream()));
  PLX_HIP(hipStreamSynchronize(stream()));
}
void h2d_async(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return;
  PLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream()));
}

// Host -> HBM copy of a caller-owned Arrow buffer.  Large buffers are page-locked in place (hipHostRegister) so the copy
// is one DMA at PCIe rate instead of the runtime's staged pageable path; the registration is dropped once the copy is
// done.  Synchronises (the caller may free the buffer on return).  PLX_PIN_UPLOADS=0 forces the pageable path.
static const size_t kPinThreshold = size_t(32) << 20;
void h2d_sync_pinned(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return;
  static const bool pin = [] { const char* e = getenv("PLX_PIN_UPLOADS"); return !(e && e[0] == '0'); }();
  bool registered = false;
  if (pin && bytes >= kPinThreshold) registered = hipHostRegister(const_cast<void*>(src), bytes, hipHostRegisterDefault) == hipSuccess;
  if (!registered) (void)hipGetLastError();   // clear a failed registration (already-registered or unsupported memory)
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream());
  const hipError_t e2 = hipStreamSynchronize(stream());
  if (registered) (void)hipHostUnregister(const_cast<void*>(src));
  PLX_HIP(e);
  PLX_HIP(e2);
}

// --------------------------------------------------------------- columns ----
ColumnPtr make_column(int dtype, int64_t len, bool with_validity) {
  auto c = std::make_shared<Column>();
  c->dtype = dtype; c->len = len;
  c->values = (dtype == PLX_BOOL) ? dev_alloc_zero(bitmap_bytes(len)) : dev_alloc(values_bytes(dtype, len));
  if (with_validity) c->validity = dev_alloc_zero(bitmap_bytes(len)); else c->null_count = 0;
  return c;
}

// repack a host bitmap with an arbitrary bit offset into an offset-0 bitmap
static std::vector<uint8_t> repack_bits(const uint8_t* bits, int64_t bit_offset, int64_t len) {
  std::vector<uint8_t> out(bitmap_bytes(len), 0);
  if (bit_offset % 8 == 0) { memcpy(out.data(), bits + bit_offset / 8, (size_t)((len + 7) / 8)); }
  else {
    for (int64_t i = 0; i < len; i++) {
      int64_t j = i + bit_offset;
      if ((bits[j >> 3] >> (j & 7)) & 1) out[i >> 3] |= uint8_t(1u << (i & 7));
    }
  }
  if (len & 7) out[(len - 1) >> 3] &= uint8_t((1u << (len & 7)) - 1);  // clear pad bits
  return out;
}

ColumnPtr column_from_host(int dtype, const void* values, const uint8_t* validity, int64_t bit_offset, int64_t len) {
  PLX_REQUIRE(dtype >= PLX_BOOL && dtype <= PLX_F64, PLX_ERR_INVALID, "column_from_host: bad dtype");
  PLX_REQUIRE(len >= 0, PLX_ERR_INVALID, "column_from_host: negative length");
  auto c = std::make_shared<Column>();
  c->dtype = dtype; c->len = len;
  std::vector<uint8_t> vbits, nbits;  // staging must outlive the async copies -> sync below
  if (dtype == PLX_BOOL) {
    c->values = dev_alloc_zero(bitmap_bytes(len));
    if (len) { vbits = repack_bits((const uint8_t*)values, bit_offset, len); h2d_async(c->values->ptr, vbits.data(), vbits.size()); }
  } else {
    c->values = dev_alloc(values_bytes(dtype, len));
    h2d_sync_pinned(c->values->ptr, values, (size_t)len * dtype_width(dtype));
  }
  if (validity) {
    nbits = repack_bits(validity, bit_offset, len);
    int64_t nulls = 0;
    for (int64_t i = 0; i < len; i++) nulls += !((nbits[i >> 3] >> (i & 7)) & 1);
    c->null_count = nulls;
    if (nulls > 0) { c->validity = dev_alloc_zero(bitmap_bytes(len)); h2d_async(c->validity->ptr, nbits.data(), nbits.size()); }
  } else c->null_count = 0;
  PLX_HIP(hipStreamSynchronize(stream()));
  return c;
}

void column_to_host(const ColumnPtr& c, void* values_out, uint8_t* validity_out, int32_t* has_validity) {
  size_t vb = c->dtype == PLX_BOOL ? (size_t)((c->len + 7) / 8) : (size_t)c->len * dtype_width(c->dtype);
  if (values_out && vb) PLX_HIP(hipMemcpyAsync(values_out, c->values->ptr, vb, hipMemcpyDeviceToHost, stream()));
  size_t nb = (size_t)((c->len + 7) / 8);
  if (validity_out && nb) {
    if (c->validity) PLX_HIP(hipMemcpyAsync(validity_out, c->validity->ptr, nb, hipMemcpyDeviceToHost, stream()));
    else memset(validity_out, 0xff, nb);
  }
  if (has_validity) *has_validity = c->validity ? 1 : 0;
  PLX_HIP(hipStreamSynchronize(stream()));
}

// ----------------------------------------------------------- handle tables --
namespace {
std::mutex g_h_mu;
std::unordered_map<uint64_t, std::pair<ColumnPtr, int>> g_cols;  // handle -> (col, refcount)
std::unordered_map<uint64_t, FramePtr> g_frames;
uint64_t g_next_handle = 1;
}  // namespace

plx_column register_column(ColumnPtr c) {
  std::lock_guard<std::mutex> lk(g_h_mu);
  uint64_t h = g_next_handle++;
  g_cols[h] = {std::move(c), 1};
  return h;
}
ColumnPtr get_column(plx_column h) {
  std::lock_guard<std::mutex> lk(g_h_mu);
  auto it = g_cols.find(h);
  if (it == g_cols.end()) fail(PLX_ERR_INVALID, "invalid column handle " + std::to_string(h));
  return it->second.first;
}
void retain_column(plx_column h) {
  std::lock_guard<std::mutex> lk(g_h_mu);
  auto it = g_cols.find(h);
  if (it == g_cols.end()) fail(PLX_ERR_INVALID, "invalid column handle " + std::to_string(h));
  it->second.second++;
}
void free_column(plx_column h) {
  ColumnPtr keep;  // destroy outside the lock
  std::lock_guard<std::mutex> lk(g_h_mu);
  auto it = g_cols.find(h);
  if (it == g_cols.end()) fail(PLX_ERR_INVALID, "invalid column handle " + std::to_string(h));
  if (--it->second.second == 0) { keep = std::move(it->second.first); g_cols.erase(it); }
}
plx_frame register_frame(FramePtr f) {
  std::lock_guard<std::mutex> lk(g_h_mu);
  uint64_t h = g_next_handle++;
  g_frames[h] = std::move(f);
  return h;
}
FramePtr get_frame(plx_frame h) {
  std::lock_guard<std::mutex> lk(g_h_mu);
  auto it = g_frames.find(h);
  if (it == g_frames.end()) fail(PLX_ERR_INVALID, "invalid frame handle " + std::to_string(h));
  return it->second;
}
void free_frame(plx_frame h) {
  FramePtr keep;
  std::lock_guard<std::mutex> lk(g_h_mu);
  auto it = g_frames.find(h);
  if (it == g_frames.end()) fail(PLX_ERR_INVALID, "invalid frame handle " + std::to_string(h));
  keep = std::move(it->second);
  g_frames.erase(it);
}
void clear_handles() {
  std::lock_guard<std::mutex> lk(g_h_mu);
  g_cols.clear(); g_frames.clear();
}

// -------------------------------------------------------------- profiling ---
namespace {
struct Rec { std::string name; hipEvent_t a, b; uint64_t bytes, rows; };
std::mutex g_p_mu;
bool g_prof_on = false;
hipEvent_t g_prof_origin = nullptr;
std::vector<Rec> g_recs;
}  // namespace

void profile_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_p_mu);
  if (on && !g_prof_origin) { PLX_HIP(hipEventCreate(&g_prof_origin)); }
  if (on) PLX_HIP(hipEventRecord(g_prof_origin, stream()));
  g_prof_on = on;
}
ProfileScope::ProfileScope(const char* name, uint64_t algo_bytes, uint64_t rows) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_p_mu);
  Rec r; r.name = name; r.bytes = algo_bytes; r.rows = rows;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, stream());
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
}
ProfileScope::~ProfileScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_p_mu);
  if (idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].b, stream());
}
int profile_fetch(plx_profile_record* out, int cap) {
  if (device_ready()) (void)hipStreamSynchronize(stream());
  std::lock_guard<std::mutex> lk(g_p_mu);
  int n = 0;
  for (auto& r : g_recs) {
    if (n >= cap) break;
    float s = 0, e = 0;
    if (hipEventElapsedTime(&s, g_prof_origin, r.a) != hipSuccess) continue;
    if (hipEventElapsedTime(&e, g_prof_origin, r.b) != hipSuccess) continue;
    plx_profile_record& o = out[n++];
    memset(&o, 0, sizeof(o));
    snprintf(o.name, sizeof(o.name), "%s", r.name.c_str());
    o.start_us = s * 1000.0; o.end_us = e * 1000.0; o.algo_bytes = r.bytes; o.rows = r.rows;
  }
  return n;
}
void profile_clear() {
  std::lock_guard<std::mutex> lk(g_p_mu);
  for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_recs.clear();
}

}  // namespace plx
