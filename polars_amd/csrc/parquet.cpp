// parquet.cpp -- device-side Parquet scan behind the C ABI ("Parquet/IPC scan -> device").
//
// plx_parquet_open parses the footer on the host (no GPU needed: metadata, schema and row-group statistics are available for
// planning / pruning on any machine).  plx_parquet_read moves the selected column chunks to HBM AS STORED (one read + one DMA per
// chunk through a page-locked double buffer) and decodes them there (parquet_reader.hpp orchestrates, kernels_parquet.hip runs
// the bodies of parquet_device.hpp).  Reference: crates/polars-io/src/parquet/read/read_impl.rs (row groups x projection),
// crates/polars-parquet/src/{parquet/read,arrow/read/deserialize}; crates/polars-stream/src/nodes/io_sources/parquet.
#include <algorithm>
#include <memory>
#include <mutex>
#include <exception>
#include <thread>
#include <functional>
#include <deque>
#include <condition_variable>

#include "core.hpp"
#include "host_stage.hpp"
#include "kernels.hpp"
#include "ops.hpp"
#include "parquet_kernels.hpp"
#include "parquet_reader.hpp"
#include "scan.hpp"

using namespace plx;

namespace {

// HBM + kernel launches on the calling thread's stream
struct HipBackend {
  using Mem = Buf;
  PinnedStage& stage_ = PinnedStage::for_this_thread();
  uint64_t encoded_bytes = 0;

  Mem alloc(size_t bytes) { return dev_alloc(bytes ? bytes : 8); }
  uint64_t addr(const Mem& m) { return m ? (uint64_t)m->ptr : 0; }
  // page-locked staging, two buffers: chunk k + 1 is read from the file while chunk k is on its way to HBM (host_stage.hpp)
  uint8_t* host_stage(size_t bytes) { return stage_.get(bytes); }
  void upload(uint64_t dst, const void* src, size_t bytes) {
    encoded_bytes += bytes;
    stage_.upload((void*)dst, src, bytes);
  }
  void discard_pending() { (void)hipStreamSynchronize(stream()); }
  // string columns with PLAIN pages: the views assembled by the reader's host threads go through the device dictionary encoder
  void encode_string_views(pq::File& f, int leaf, const uint8_t* views, const uint8_t* validity, int64_t n, const std::vector<const void*>& ptrs,
                           const std::vector<int64_t>& sizes, pq::ColumnResult<HipBackend>* res) {
    // Everything goes to the device from here (round 5; through plx_strview_dict_encode before: it registered the 16 n bytes of views with the driver for one copy,
    // uploaded n zero bytes to carry the validity bitmap and synchronised after every page's buffer -- most of the 115 ms a 2e7-row PLAIN string column took to read).
    // The views were assembled in this thread's page-locked staging buffer (read_string_column_host): one DMA.
    Buf dv = dev_alloc((size_t)std::max<int64_t>(n, 1) * 16);
    if (n) stage_.upload(dv->ptr, views, (size_t)n * 16);
    ColumnPtr vh;
    if (validity && res->null_count > 0) {
      vh = std::make_shared<Column>();
      vh->dtype = PLX_U8; vh->len = n; vh->null_count = res->null_count;
      vh->validity = dev_alloc_zero(bitmap_bytes(n));
      h2d_async(vh->validity->ptr, validity, (size_t)((n + 7) / 8));
    }
    uint64_t total = 0;
    std::vector<uint64_t> base(std::max<size_t>(ptrs.size(), 1), 0);
    for (size_t i = 0; i < ptrs.size(); i++) { base[i] = total; total += (uint64_t)sizes[i]; }
    Buf data = dev_alloc((size_t)total + 64), bb = dev_alloc(sizeof(uint64_t) * base.size());
    for (size_t i = 0; i < ptrs.size(); i++) if (sizes[i]) h2d_async((uint8_t*)data->ptr + base[i], ptrs[i], (size_t)sizes[i]);      // page payloads (pageable): queued, one wait below
    h2d_async(bb->ptr, base.data(), sizeof(uint64_t) * base.size());
    PLX_HIP(hipStreamSynchronize(stream()));       // the host buffers may go away when this returns
    plx_column codes = 0;
    plx_strdict dict = 0;
    strview_encode_device_bases(dv->as<uint64_t>(), vh, data, bb, n, &codes, &dict);
    ColumnPtr c = get_column(codes);
    free_column(codes);
    res->values = c->values;
    res->validity = c->validity;
    res->has_validity = (bool)c->validity;
    std::lock_guard<std::mutex> lk(f.meta_mu);
    auto it = f.strdicts.find(leaf);
    if (it != f.strdicts.end() && it->second) plx_strdict_free(it->second);
    f.strdicts[leaf] = dict;
    f.categories.erase(leaf);
  }
  // descriptor arrays from pageable memory: done when this returns
  void upload_small(uint64_t dst, const void* src, size_t bytes) {
    if (!bytes) return;
    PLX_HIP(hipMemcpyAsync((void*)dst, src, bytes, hipMemcpyHostToDevice, stream()));
    PLX_HIP(hipStreamSynchronize(stream()));
  }
  void zero(uint64_t dst, size_t bytes) { PLX_HIP(hipMemsetAsync((void*)dst, 0, bytes, stream())); }
  uint64_t read_u64(uint64_t a) { uint64_t v = 0; d2h_sync(&v, (const void*)a, 8); return v; }
  uint32_t read_u32(uint64_t a) { uint32_t v = 0; d2h_sync(&v, (const void*)a, 4); return v; }
  void scan_u32(const uint32_t* in, uint64_t* out, int64_t n) { k::exclusive_scan_u32(in, out, n); }

  void run_snappy(const pq::DecompJob* jobs, uint32_t n, uint64_t bytes_out, uint32_t* err) { k::pq_snappy(jobs, n, bytes_out, err); }
  void run_zstd(pq::ZstdBlock* blocks, const uint32_t* order, uint32_t n_compressed, uint32_t n_huf_only, const pq::ZstdHufDesc* hufs, const pq::ZstdFseDesc* fses, const pq::ZstdStream* streams,
                uint32_t n_streams, uint64_t bytes_in, uint64_t bytes_out, uint32_t* err) {
    k::pq_zstd(blocks, order, n_compressed, n_huf_only, hufs, fses, streams, n_streams, bytes_in, bytes_out, err);
  }
  void run_page_prepare(pq::PageDesc* pages, uint32_t n, uint32_t* err) { k::pq_page_prepare(pages, n, err); }
  void run_count_runs(const pq::PageDesc* pages, uint32_t n, bool with_levels, uint32_t* counts, uint32_t* err) { k::pq_count_runs(pages, n, with_levels, counts, err); }
  void run_fill_runs(const pq::PageDesc* pages, uint32_t n, const uint64_t* offs, pq::RunEntry* runs) { k::pq_fill_runs(pages, n, offs, runs); }
  void run_validity(const pq::PageDesc* pages, uint32_t n, const pq::RunEntry* runs, const uint64_t* offs, uint64_t n_rows, uint64_t* validity, uint32_t* popc,
                    uint32_t* err) {
    k::pq_validity(pages, n, runs, offs, n_rows, validity, popc, err);
  }
  void run_page_valid0(pq::PageDesc* pages, uint32_t n, const uint64_t* validity, const uint64_t* prefix) { k::pq_page_valid0(pages, n, validity, prefix); }
  void run_decode(const pq::ColumnDecode& c, void* out, uint32_t out_width, uint32_t* err) { k::pq_decode(c, out, out_width, encoded_bytes, err); }
};

// The columns of one plx_parquet_read run on a few persistent host threads, each with a HIP stream of its own (core.cpp: one stream per calling thread; freed blocks carry
// events across streams).  A column is a chain upload -> decompress -> decode with host round trips in between (descriptor uploads, error words, run counts), and the
// Snappy kernel runs one workgroup per page: alone, a column's ~1000 pages fill the chip twice and its tail runs on a few CUs -- with several columns in flight the chip stays
// full and one column's uploads overlap another's kernels (2e7-row lineitem-like file: Snappy read 68 -> see DESIGN.md 4.5).  The threads persist, so their page-locked staging
// buffers (host_stage.hpp: per thread) are allocated once.  PLX_PARQUET_THREADS=1 reads the columns one after another on the caller's stream.
class ColumnWorkers {
 public:
  static ColumnWorkers& get() { static ColumnWorkers* w = new ColumnWorkers(); return *w; }      // (never destroyed: the threads outlive static destruction)
  int size() const { return (int)queues_.size(); }
  // Runs every task; rethrows the first exception (after all tasks have finished).  Task -> worker is a function of the weights alone
  // (heaviest first onto the least loaded worker): reading the same columns again puts each column on the thread whose page-locked
  // staging buffers already have its size -- a worker that meets a larger column than it has seen re-allocates them (tens of ms).
  void run(std::vector<std::function<void()>>& tasks, const std::vector<uint64_t>& weights) {
    std::vector<std::exception_ptr> errs(tasks.size());
    std::vector<size_t> order(tasks.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weights[a] > weights[b]; });
    std::vector<uint64_t> load(queues_.size(), 0);
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i : order) {
        const size_t w = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        load[w] += std::max<uint64_t>(weights[i], 1);
        queues_[w].push_back([&tasks, &errs, i] { try { tasks[i](); } catch (...) { errs[i] = std::current_exception(); } });
      }
      pending_ += tasks.size();
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    lk.unlock();
    for (auto& e : errs) if (e) std::rethrow_exception(e);
  }

 private:
  ColumnWorkers() {
    int n = 6;       // (2e7-row file, none / Snappy / zstd: 4 workers 17.8 / 38.2 / 40-45 ms, 6 workers 16.9 / 36.4 / 36.3, 8 workers 22.9 / 38.4 / 45.1)
    if (const char* e = getenv("PLX_PARQUET_THREADS")) n = std::max(1, std::min(16, atoi(e)));
    const int ordinal = device().ordinal;
    queues_.resize((size_t)n);
    for (int i = 0; i < n; i++) {
      std::thread([this, ordinal, i] {
        (void)hipSetDevice(ordinal);
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) set_thread_stream(s);
        pq::host_thread_share = std::max<size_t>(8, 128 / queues_.size());
        std::deque<std::function<void()>>& q = queues_[(size_t)i];
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&q] { return !q.empty(); });
            job = std::move(q.front());
            q.pop_front();
          }
          job();
          (void)hipStreamSynchronize(stream());          // what the task produced is complete before anybody on another stream looks at it
          std::lock_guard<std::mutex> lk(mu_);
          if (--pending_ == 0) done_cv_.notify_all();
        }
      }).detach();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::deque<std::function<void()>>> queues_;      // one per worker; sized once, before the threads start
  size_t pending_ = 0;
};

std::mutex g_mu;
std::vector<std::unique_ptr<pq::File>> g_files;   // handle = index + 1
std::vector<std::unique_ptr<std::mutex>> g_file_mu;   // one reader at a time per handle (a read rewrites the handle's string dictionaries)
std::mutex& file_mutex(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (h == 0 || h > g_file_mu.size() || !g_file_mu[h - 1]) fail(PLX_ERR_INVALID, "invalid parquet handle");
  return *g_file_mu[h - 1];
}

pq::File& get_file(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (h == 0 || h > g_files.size() || !g_files[h - 1]) fail(PLX_ERR_INVALID, "invalid parquet handle");
  return *g_files[h - 1];
}

// per-thread copy of the last column name handed out (the pointer stays valid until the next call on this thread)
thread_local std::string t_name;

}  // namespace

#define PQ_TRY try {
#define PQ_CATCH                                                                                   \
  }                                                                                                \
  catch (const plx::Error& e) { plx::set_last_error(e.msg); return e.code; }                        \
  catch (const pq::Unsupported& e) { plx::set_last_error(std::string("parquet: ") + e.what()); return PLX_ERR_UNSUPPORTED; } \
  catch (const pq::FormatError& e) { plx::set_last_error(std::string("parquet: ") + e.what()); return PLX_ERR_INVALID; }     \
  catch (const plx::IoError& e) { plx::set_last_error(std::string("parquet: ") + e.what()); return PLX_ERR_INVALID; }        \
  catch (const std::bad_alloc&) { plx::set_last_error("host out of memory"); return PLX_ERR_OOM; } \
  catch (const std::exception& e) { plx::set_last_error(std::string("PANIC: ") + e.what()); return PLX_ERR_INVALID; }        \
  catch (...) { plx::set_last_error("PANIC"); return PLX_ERR_INVALID; }                             \
  return PLX_OK;

extern "C" {

int plx_parquet_open(const char* path, plx_parquet* out) {
  PQ_TRY
  PLX_REQUIRE(path && out, PLX_ERR_INVALID, "null argument");
  std::unique_ptr<pq::File> f = pq::open_file(path);
  std::lock_guard<std::mutex> lk(g_mu);
  g_files.push_back(std::move(f));
  g_file_mu.push_back(std::make_unique<std::mutex>());
  *out = (plx_parquet)g_files.size();
  PQ_CATCH
}

int plx_parquet_close(plx_parquet file) {
  PQ_TRY
  std::lock_guard<std::mutex> lk(g_mu);
  if (file && file <= g_files.size()) g_files[file - 1].reset();
  PQ_CATCH
}

int plx_parquet_shape(plx_parquet file, int64_t* num_rows, int32_t* num_row_groups, int32_t* num_columns) {
  PQ_TRY
  pq::File& f = get_file(file);
  if (num_rows) *num_rows = f.md.num_rows;
  if (num_row_groups) *num_row_groups = (int32_t)f.md.row_groups.size();
  if (num_columns) *num_columns = (int32_t)f.md.leaves.size();
  PQ_CATCH
}

int plx_parquet_column_info(plx_parquet file, int32_t column, const char** name, int32_t* dtype, int32_t* logical, int32_t* nullable) {
  PQ_TRY
  pq::File& f = get_file(file);
  PLX_REQUIRE(column >= 0 && (size_t)column < f.md.leaves.size(), PLX_ERR_INVALID, "parquet column index out of range");
  const pq::Leaf& l = f.md.leaves[column];
  const pq::LeafType t = pq::leaf_type(l);
  if (name) { t_name = l.name; *name = t_name.c_str(); }
  if (dtype) *dtype = t.dtype;
  if (logical) *logical = t.logical;
  if (nullable) *nullable = l.repetition == pq::REP_OPTIONAL ? 1 : 0;
  PQ_CATCH
}

int plx_parquet_column_timezone(plx_parquet file, int32_t column, const char** timezone) {
  PQ_TRY
  pq::File& f = get_file(file);
  PLX_REQUIRE(column >= 0 && (size_t)column < f.md.leaves.size() && timezone, PLX_ERR_INVALID, "parquet column index out of range");
  const pq::Leaf& l = f.md.leaves[column];
  // Parquet has no zone names: isAdjustedToUTC = instants, which the reference reads as Datetime(unit, "UTC") (schema/convert.rs)
  const bool ts = l.logical == pq::LG_TIMESTAMP_MILLIS || l.logical == pq::LG_TIMESTAMP_MICROS || l.logical == pq::LG_TIMESTAMP_NANOS;
  t_name = ts && l.utc ? "UTC" : "";
  *timezone = t_name.c_str();
  PQ_CATCH
}

int plx_parquet_row_group_info(plx_parquet file, int32_t row_group, int64_t* num_rows, int64_t* compressed_bytes) {
  PQ_TRY
  pq::File& f = get_file(file);
  PLX_REQUIRE(row_group >= 0 && (size_t)row_group < f.md.row_groups.size(), PLX_ERR_INVALID, "parquet row group index out of range");
  const pq::RowGroup& g = f.md.row_groups[row_group];
  if (num_rows) *num_rows = g.num_rows;
  if (compressed_bytes) {
    int64_t b = 0;
    for (const pq::ColumnChunk& c : g.columns) b += c.total_compressed_size;
    *compressed_bytes = b;
  }
  PQ_CATCH
}

int plx_parquet_chunk_info(plx_parquet file, int32_t row_group, int32_t column, int32_t* codec, uint32_t* encodings, int64_t* compressed_bytes,
                           int64_t* uncompressed_bytes, int32_t* has_min_max, plx_scalar* min, plx_scalar* max, int64_t* null_count) {
  PQ_TRY
  pq::File& f = get_file(file);
  PLX_REQUIRE(row_group >= 0 && (size_t)row_group < f.md.row_groups.size(), PLX_ERR_INVALID, "parquet row group index out of range");
  PLX_REQUIRE(column >= 0 && (size_t)column < f.md.leaves.size(), PLX_ERR_INVALID, "parquet column index out of range");
  const pq::ColumnChunk& c = f.md.row_groups[row_group].columns[column];
  if (codec) *codec = c.codec;
  if (encodings) *encodings = c.encodings;
  if (compressed_bytes) *compressed_bytes = c.total_compressed_size;
  if (uncompressed_bytes) *uncompressed_bytes = c.total_uncompressed_size;
  plx_scalar mn, mx;
  mn.u = mx.u = 0;
  const bool has = pq::chunk_min_max(f.md.leaves[column], c, &mn, &mx);
  if (has_min_max) *has_min_max = has ? 1 : 0;
  if (min) *min = mn;
  if (max) *max = mx;
  if (null_count) *null_count = c.stats.has_null_count ? c.stats.null_count : -1;
  PQ_CATCH
}

int plx_parquet_read(plx_parquet file, const int32_t* row_groups, int32_t n_row_groups, const int32_t* columns, int32_t n_columns, plx_frame* out) {
  PQ_TRY
  PLX_REQUIRE(out && (columns || n_columns == 0) && (row_groups || n_row_groups == 0), PLX_ERR_INVALID, "null argument");
  pq::File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  device();   // fails loudly without a GPU: there is no host decode path in the library
  std::vector<int> rgs(row_groups, row_groups + n_row_groups);
  auto frame = std::make_shared<Frame>();
  try {
    std::vector<ColumnPtr> cols((size_t)n_columns);
    auto read_one = [&](int32_t i) {
      check_cancel();
      HipBackend be;       // the page-locked staging buffers behind it belong to the host thread and are kept (host_stage.hpp)
      pq::ReadStats st;
      pq::ColumnResult<HipBackend> r = pq::read_column(be, f, rgs, columns[i], &st);
      auto col = std::make_shared<Column>();
      col->dtype = r.dtype; col->len = r.len; col->values = r.values;
      if (r.has_validity) col->validity = r.validity;
      col->null_count = r.null_count;
      cols[(size_t)i] = col;
    };
    static const bool serial = [] { const char* e = getenv("PLX_PARQUET_THREADS"); return e && atoi(e) == 1; }();
    if (n_columns > 1 && !serial) {
      std::vector<std::function<void()>> tasks;
      std::vector<uint64_t> weights((size_t)n_columns, 0);      // stored bytes of the column's selected chunks
      for (int32_t i = 0; i < n_columns; i++) {
        tasks.push_back([&read_one, i] { read_one(i); });
        if (columns[i] >= 0 && (size_t)columns[i] < f.md.leaves.size())
          for (int g : rgs)
            if (g >= 0 && (size_t)g < f.md.row_groups.size()) weights[(size_t)i] += (uint64_t)std::max<int64_t>(f.md.row_groups[g].columns[columns[i]].total_uncompressed_size, 0);
      }
      ColumnWorkers::get().run(tasks, weights);
    } else {
      for (int32_t i = 0; i < n_columns; i++) read_one(i);
    }
    for (int32_t i = 0; i < n_columns; i++) {
      frame->names.push_back(f.md.leaves[columns[i]].name);
      frame->cols.push_back(cols[(size_t)i]);
      frame->height = cols[(size_t)i]->len;
    }
  } catch (...) {
    (void)hipStreamSynchronize(stream());   // descriptor vectors / staging of the failed column may still be in flight
    throw;
  }
  if (n_columns == 0) {
    int64_t n = 0;
    for (int g : rgs) { PLX_REQUIRE(g >= 0 && (size_t)g < f.md.row_groups.size(), PLX_ERR_INVALID, "parquet row group index out of range"); n += f.md.row_groups[g].num_rows; }
    frame->height = n;
  }
  *out = register_frame(frame);
  PQ_CATCH
}

int plx_parquet_column_strdict(plx_parquet file, int32_t column, plx_strdict* out) {
  PQ_TRY
  pq::File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  PLX_REQUIRE(out, PLX_ERR_INVALID, "null out pointer");
  auto it = f.strdicts.find(column);
  PLX_REQUIRE(it != f.strdicts.end() && it->second, PLX_ERR_NOT_FOUND, "no device dictionary: the column's pages were dictionary-encoded (plx_parquet_categories) or it has not been read");
  *out = it->second;
  it->second = 0;       // ownership moves to the caller (plx_strdict_free)
  PQ_CATCH
}

int plx_parquet_categories(plx_parquet file, int32_t column, int64_t* n_strings, int64_t* total_bytes) {
  PQ_TRY
  pq::File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  auto it = f.categories.find(column);
  PLX_REQUIRE(it != f.categories.end(), PLX_ERR_NOT_FOUND, "no string dictionary: the column has not been read (or is not a string column)");
  int64_t b = 0;
  for (const std::string& s : it->second) b += (int64_t)s.size();
  if (n_strings) *n_strings = (int64_t)it->second.size();
  if (total_bytes) *total_bytes = b;
  PQ_CATCH
}

int plx_parquet_categories_to_host(plx_parquet file, int32_t column, int64_t* offsets, uint8_t* bytes) {
  PQ_TRY
  pq::File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  auto it = f.categories.find(column);
  PLX_REQUIRE(it != f.categories.end(), PLX_ERR_NOT_FOUND, "no string dictionary: the column has not been read (or is not a string column)");
  PLX_REQUIRE(offsets, PLX_ERR_INVALID, "null offsets pointer");
  int64_t off = 0, i = 0;
  for (const std::string& s : it->second) {
    offsets[i++] = off;
    if (!s.empty()) { PLX_REQUIRE(bytes, PLX_ERR_INVALID, "null bytes pointer"); memcpy(bytes + off, s.data(), s.size()); }
    off += (int64_t)s.size();
  }
  offsets[i] = off;
  PQ_CATCH
}

}  // extern "C"
