// ipc.cpp -- Arrow IPC file (Feather V2) scan -> device columns behind the C ABI ("Parquet/IPC scan -> device").
//
// An IPC file already holds the hot path's layout (LZ4_FRAME / ZSTD compressed bodies are inflated buffer by buffer on the host first,
// host_codecs.hpp): a primitive column of a record batch is a values buffer + a validity
// bitmap, exactly what a device column is.  So there is nothing to decode: the host parses the FlatBuffers metadata (ipc_format.hpp),
// every selected buffer travels file -> page-locked staging -> HBM in one DMA, batches are concatenated in place (bitmaps that do not
// start on a word boundary are merged by the bitmap blit kernel), dictionary indices are widened to u32 codes by the cast kernel, and
// string columns that are not dictionary-encoded (Utf8 / LargeUtf8 / Utf8View) go through the device-side dictionary encoder of
// kernels_strview.hip.  Reference: crates/polars-arrow/src/io/ipc/read/{file.rs,common.rs,read_basic.rs,schema.rs},
// crates/polars-io/src/ipc/ipc_file.rs, crates/polars-stream/src/nodes/io_sources/ipc.rs.
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <atomic>
#include <thread>

#include <chrono>
#include "core.hpp"
#include "host_stage.hpp"
#include "ipc_reader.hpp"
#include "kernels.hpp"
#include "ops.hpp"

using namespace plx;

namespace {

using ipc::ColType; using ipc::File; using ipc::Slot; using ipc::Unsupported; using ipc::col_type; using ipc::decode_strings; using ipc::load_dictionaries;
using ipc::open_file; using ipc::read_block_meta; using ipc::read_buffer; using ipc::slot_of;
using ipc::LO_NONE; using ipc::LO_DATE; using ipc::LO_DATETIME_US; using ipc::LO_STRING; using ipc::LO_BINARY;

int idx_dtype(int bits, bool is_signed) {
  switch (bits) {
    case 8: return is_signed ? PLX_I8 : PLX_U8;
    case 16: return is_signed ? PLX_I16 : PLX_U16;
    case 32: return is_signed ? PLX_I32 : PLX_U32;
    default: return is_signed ? PLX_I64 : PLX_U64;
  }
}

// one bitmap of a batch (n bits at body + r) OR-ed into dst at bit dst_off; dst is zeroed
void blit_bitmap(const File& f, PinnedStage& st, const ipc::BatchMeta& bm, int64_t body, const ipc::BufferRef& r, int64_t n, uint64_t* dst, int64_t dst_off) {
  const size_t bytes = (size_t)((n + 7) / 8);
  Buf tmp = dev_alloc(bitmap_bytes(n));
  uint8_t* h = st.get(bytes);
  ipc::load_buffer(f, bm, body, r, h, bytes);       // stored, or LZ4-frame / zstd decompressed by this host thread (host_codecs.hpp)
  st.upload(tmp->ptr, h, bytes);
  k::bitmap_blit(dst, dst_off, tmp->as<uint64_t>(), n);
}
void blit_ones(int64_t n, uint64_t* dst, int64_t dst_off) {
  Buf ones = dev_alloc(bitmap_bytes(n));
  k::fill(8, ones->ptr, ~0ull, (n + 63) / 64);
  k::bitmap_blit(dst, dst_off, ones->as<uint64_t>(), n);
}

ColumnPtr read_fixed_column(File& f, const std::vector<int>& bsel, int col, const ColType& ct, int64_t total) {
  PinnedStage& st = PinnedStage::for_this_thread();
  const ipc::Field& fl = f.footer.fields[col];
  bool any_nulls = false;
  for (int b : bsel) {
    const ipc::BatchMeta& bm = f.batches[b];
    Slot s = slot_of(f, bm, col);
    if (s.node >= bm.nodes.size() || s.buf + 2 > bm.buffers.size()) throw ipc::FormatError("record batch without the column's buffers");
    if (bm.nodes[s.node].length != bm.length) throw ipc::FormatError("column length differs from its record batch");
    if (bm.nodes[s.node].null_count > 0) any_nulls = true;
  }
  const int file_dtype = fl.has_dictionary ? idx_dtype(fl.index_bits, fl.index_signed) : ct.dtype;
  auto c = std::make_shared<Column>();
  c->dtype = file_dtype; c->len = total;
  c->values = dev_alloc(values_bytes(file_dtype, total) + 8);
  if (file_dtype == PLX_BOOL) PLX_HIP(hipMemsetAsync(c->values->ptr, 0, bitmap_bytes(total), stream()));
  if (any_nulls) { c->validity = dev_alloc_zero(bitmap_bytes(total)); } else c->null_count = 0;
  int64_t row = 0, nulls = 0;
  // Compressed bodies: a values buffer is one LZ4 frame / zstd stream of up to a whole record batch -- inflating them one after the
  // other on this thread would leave the host idle.  The buffers of all selected batches are inflated in parallel straight into a
  // page-locked image of the column (groups of ~256 MB: group k is on its way over PCIe while group k + 1 is inflated).
  bool any_compressed = false;
  for (int b : bsel) any_compressed = any_compressed || f.batches[b].compressed;
  // Uncompressed bodies of MANY batches take the same route: one positional read per (batch, column) buffer on the calling thread -- each
  // spawning its own few slice threads -- left a 1 GB file at 8 GB/s (pyarrow's mmap "read": 2x faster); a pool over all the column's buffers
  // fills the page-locked image at memory speed and the column crosses PCIe in one DMA while the next column is read.
  const bool values_in_parallel = (any_compressed || bsel.size() > 1) && file_dtype != PLX_BOOL;
  if (values_in_parallel) {
    struct Task { const ipc::BatchMeta* bm; int64_t body; const ipc::BufferRef* vb; size_t off, bytes; };
    const size_t kGroup = (size_t)256 << 20;
    size_t i = 0;
    int64_t row0 = 0;
    while (i < bsel.size()) {
      std::vector<Task> tasks;
      size_t bytes = 0, j = i;
      int64_t rows = 0;
      while (j < bsel.size() && (j == i || bytes + (size_t)f.batches[bsel[j]].length * (size_t)ct.width <= kGroup)) {
        const ipc::BatchMeta& bm = f.batches[bsel[j]];
        const Slot s = slot_of(f, bm, col);
        const size_t nb = (size_t)bm.length * (size_t)ct.width;
        if (nb) tasks.push_back({&bm, f.body_off[bsel[j]], &bm.buffers[s.buf + 1], bytes, nb});
        bytes += nb; rows += bm.length; j++;
      }
      if (bytes) {
        uint8_t* img = st.get(bytes);
        const size_t threads = std::min<size_t>(std::min<size_t>(64, std::max(2u, std::thread::hardware_concurrency() / 2)), tasks.size());
        std::vector<std::exception_ptr> errs(std::max<size_t>(threads, 1));
        std::atomic<size_t> next{0};
        auto work = [&](size_t t) {
          try {
            for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
              const Task& tk = tasks[k];
              if (!tk.bm->compressed) {               // stored: one positional read by this pool thread (no nested slice threads)
                if ((int64_t)tk.bytes > tk.vb->length) throw ipc::FormatError("buffer shorter than the array needs");
                f.pread_exact(img + tk.off, tk.bytes, tk.body + tk.vb->offset);
              } else ipc::load_buffer(f, *tk.bm, tk.body, *tk.vb, img + tk.off, tk.bytes);
            }
          } catch (...) { errs[t] = std::current_exception(); }
        };
        if (threads <= 1) work(0);
        else {
          std::vector<std::thread> pool;
          for (size_t t = 0; t < threads; t++) pool.emplace_back(work, t);
          for (std::thread& th : pool) th.join();
        }
        for (std::exception_ptr& ep : errs) if (ep) std::rethrow_exception(ep);
        st.upload((uint8_t*)c->values->ptr + (size_t)row0 * (size_t)ct.width, img, bytes);
      }
      row0 += rows;
      i = j;
    }
  }
  for (int b : bsel) {
    const ipc::BatchMeta& bm = f.batches[b];
    const Slot s = slot_of(f, bm, col);
    const int64_t n = bm.length, body = f.body_off[b];
    const ipc::BufferRef& vb = bm.buffers[s.buf + 1];
    if (n == 0) continue;
    if (file_dtype == PLX_BOOL) {
      blit_bitmap(f, st, bm, body, vb, n, c->values->as<uint64_t>(), row);
    } else if (values_in_parallel) {
      // done above
    } else {
      const size_t bytes = (size_t)n * (size_t)ct.width;
      uint8_t* h = st.get(bytes);
      ipc::load_buffer(f, bm, body, vb, h, bytes);
      st.upload((uint8_t*)c->values->ptr + (size_t)row * (size_t)ct.width, h, bytes);
    }
    if (any_nulls) {
      const int64_t nc = bm.nodes[s.node].null_count;
      nulls += nc;
      if (nc > 0) blit_bitmap(f, st, bm, body, bm.buffers[s.buf], n, c->validity->as<uint64_t>(), row);
      else blit_ones(n, c->validity->as<uint64_t>(), row);
    }
    row += n;
  }
  if (any_nulls) c->null_count = nulls;
  if (fl.has_dictionary && file_dtype != PLX_U32) {
    // dictionary indices -> u32 codes on the device
    auto out = std::make_shared<Column>();
    out->dtype = PLX_U32; out->len = total; out->validity = c->validity; out->null_count = c->null_count;
    out->values = dev_alloc(values_bytes(PLX_U32, total) + 8);
    k::cast(file_dtype, PLX_U32, c->values->ptr, total, out->values->ptr, nullptr);
    return out;
  }
  c->dtype = ct.dtype;
  return c;
}

// rows [0, n) in parallel slices (the 16-byte views of a string column are independent of each other; one host thread assembles
// ~2e8 of them per second, a 2e7-row column would otherwise cost more than its upload and its device encode together)
template <class Fn> void parallel_rows(int64_t n, Fn&& fn) {
  const int64_t kMinRows = int64_t(1) << 16;
  const int threads = (int)std::min<int64_t>(16, n / kMinRows);
  if (threads < 2) { fn((int64_t)0, n); return; }
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> errs((size_t)threads);
  const int64_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    const int64_t b = std::min(n, t * per), e = std::min(n, b + per);
    pool.emplace_back([&fn, &errs, t, b, e] { try { if (e > b) fn(b, e); } catch (...) { errs[(size_t)t] = std::current_exception(); } });
  }
  for (std::thread& th : pool) th.join();
  for (std::exception_ptr& ep : errs) if (ep) std::rethrow_exception(ep);
}
// bits [off, off + n) of a little-endian bitmap := 1
void set_bits(uint8_t* bits, int64_t off, int64_t n) {
  int64_t i = off, end = off + n;
  for (; i < end && (i & 7); i++) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
  if (end - i >= 8) { memset(bits + (i >> 3), 0xff, (size_t)((end - i) >> 3)); i += (end - i) & ~int64_t(7); }
  for (; i < end; i++) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
}

// Utf8 / LargeUtf8 (+ binary twins), not dictionary-encoded: offsets and bytes of every selected batch cross PCIe as they are stored (a pool
// of host threads fills two page-locked images: 4-8 B of offsets per row instead of 16 B of host-assembled views out of pageable memory -- the
// 1-character l_returnflag column of a 2e7-row file took 115 of the file's 150 ms that way), the views are built by a kernel
// (k::strviews_from_offsets) and encoded on the device like every other string column.
// raw != nullptr: the column is NOT dictionary-encoded -- its views (2 x total UInt64 words) and the bytes behind them are handed back as they sit in HBM
// (plx_ipc_read_string_views: a group-by keyed on the column runs on the views, kernels_strgroup.hip); nulls are stamped into the views
struct RawViews { ColumnPtr views, data; };
ColumnPtr read_offset_string_column(File& f, const std::vector<int>& bsel, int col, int64_t total, bool large, plx_strdict* dict_out, RawViews* raw = nullptr) {
  PinnedStage& st = PinnedStage::for_this_thread();
  const size_t ow = large ? 8 : 4;
  struct Part { const ipc::BatchMeta* bm; int64_t body, n, row0; const ipc::BufferRef *vbits, *offs, *data; size_t off_at, data_at, data_len; };
  std::vector<Part> parts;
  std::vector<uint8_t> validity((size_t)(total + 7) / 8 + 8, 0);
  bool any_nulls = false;
  size_t off_bytes = 0, data_bytes = 0;
  int64_t row = 0;
  for (int b : bsel) {
    const ipc::BatchMeta& bm = f.batches[b];
    const Slot s = slot_of(f, bm, col);
    const int64_t n = bm.length, body = f.body_off[b];
    if (s.node >= bm.nodes.size() || s.buf + 3 > bm.buffers.size()) throw ipc::FormatError("record batch without the column's buffers");
    if (bm.nodes[s.node].length != n) throw ipc::FormatError("column length differs from its record batch");
    const int64_t nc = bm.nodes[s.node].null_count;
    if (nc > 0) {
      any_nulls = true;
      const std::vector<uint8_t> vbits = read_buffer(f, bm, body, bm.buffers[s.buf]);
      if ((int64_t)vbits.size() - 16 < (n + 7) / 8) throw ipc::FormatError("bitmap buffer shorter than the array");
      for (int64_t i = 0; i < n; i++)
        if ((vbits[(size_t)i >> 3] >> (i & 7)) & 1) validity[(size_t)(row + i) >> 3] |= (uint8_t)(1u << ((row + i) & 7));
    } else set_bits(validity.data(), row, n);
    const int64_t dlen = bm.compressed ? ipc::compressed_buffer_length(f, body, bm.buffers[s.buf + 2]) : bm.buffers[s.buf + 2].length;
    if (dlen < 0) throw ipc::FormatError("string data buffer with a negative length");
    if (n) parts.push_back({&bm, body, n, row, &bm.buffers[s.buf], &bm.buffers[s.buf + 1], &bm.buffers[s.buf + 2], off_bytes, data_bytes, (size_t)dlen});
    if (n) { off_bytes += (size_t)(n + 1) * ow; data_bytes += ((size_t)dlen + 15) & ~size_t(15); }
    row += n;
  }
  if (data_bytes >= ((size_t)1 << 32)) throw Unsupported("string column with 4 GiB or more of bytes in one read");
  Buf d_offs = dev_alloc(std::max<size_t>(off_bytes, 8) + 64), d_data = dev_alloc(std::max<size_t>(data_bytes, 8) + 64), d_views = dev_alloc((size_t)std::max<int64_t>(total, 1) * 16);
  Buf err = dev_alloc_zero(8);
  // both images through the pool: tasks = (part, which buffer)
  for (int which = 0; which < 2; which++) {
    const size_t bytes = which == 0 ? off_bytes : data_bytes;
    if (!bytes) continue;
    uint8_t* img = st.get(bytes);
    const size_t threads = std::min<size_t>(std::min<size_t>(64, std::max(2u, std::thread::hardware_concurrency() / 2)), parts.size());
    std::vector<std::exception_ptr> errs(std::max<size_t>(threads, 1));
    std::atomic<size_t> next{0};
    auto work = [&](size_t t) {
      try {
        for (size_t k = next.fetch_add(1); k < parts.size(); k = next.fetch_add(1)) {
          const Part& p = parts[k];
          const ipc::BufferRef& r = which == 0 ? *p.offs : *p.data;
          const size_t need = which == 0 ? (size_t)(p.n + 1) * ow : p.data_len;
          uint8_t* dst = img + (which == 0 ? p.off_at : p.data_at);
          if (!need) continue;
          if (!p.bm->compressed) {
            if ((int64_t)need > r.length) throw ipc::FormatError("buffer shorter than the array needs");
            f.pread_exact(dst, need, p.body + r.offset);
          } else ipc::load_buffer(f, *p.bm, p.body, r, dst, need);
        }
      } catch (...) { errs[t] = std::current_exception(); }
    };
    if (threads <= 1) work(0);
    else {
      std::vector<std::thread> pool;
      for (size_t t = 0; t < threads; t++) pool.emplace_back(work, t);
      for (std::thread& th : pool) th.join();
    }
    for (std::exception_ptr& ep : errs) if (ep) std::rethrow_exception(ep);
    st.upload(which == 0 ? d_offs->ptr : d_data->ptr, img, bytes);
  }
  ColumnPtr vh;
  if (any_nulls) {
    std::vector<uint8_t> zeros((size_t)total);
    vh = column_from_host(PLX_U8, zeros.data(), validity.data(), 0, total);
  }
  for (const Part& p : parts)
    k::strviews_from_offsets((const uint8_t*)d_offs->ptr + p.off_at, large, d_data->as<uint8_t>(), (uint64_t)p.data_at, (int64_t)p.data_len, p.n, d_views->as<uint64_t>() + (size_t)p.row0 * 2,
                             vh ? vh->valid_words() : nullptr, p.row0, raw != nullptr, err->as<unsigned int>());
  uint32_t bad = 0;
  d2h_sync(&bad, err->ptr, 4);
  if (bad) throw ipc::FormatError("string offsets outside the data buffer");
  if (raw) {
    // (nulls travel as stamped views: kernels.hpp kStrviewNullLen)
    auto col_of = [](int dtype, int64_t len, const Buf& b) { auto c = std::make_shared<Column>(); c->dtype = dtype; c->len = len; c->values = b; c->null_count = 0; return c; };
    raw->views = col_of(PLX_U64, total * 2, d_views);
    raw->data = col_of(PLX_U8, (int64_t)data_bytes, d_data);
    return nullptr;
  }
  plx_column codes = 0;
  uint64_t dict = 0;
  strview_encode_device(d_views->as<uint64_t>(), vh, d_data, total, &codes, &dict);
  ColumnPtr c = get_column(codes);
  free_column(codes);       // the frame keeps the column alive
  *dict_out = (plx_strdict)dict;
  return c;
}

// Utf8View (+ binary twin; also the fallback shape of the function above), not dictionary-encoded: 16-byte views are assembled on the host (offsets ->
// {len, inline bytes | prefix, buffer, offset}; view buffer indices rebased across batches), hashing / comparing / encoding happens
// on the device (plx_strview_dict_encode: kernels_strview.hip)
ColumnPtr read_string_column(File& f, const std::vector<int>& bsel, int col, int64_t total, plx_strdict* dict_out) {
  const ipc::Field& fl = f.footer.fields[col];
  const bool is_view = fl.type == ipc::TY_UTF8_VIEW || fl.type == ipc::TY_BINARY_VIEW;
  const bool large = fl.type == ipc::TY_LARGE_UTF8 || fl.type == ipc::TY_LARGE_BINARY;
  if (!is_view && !getenv("PLX_IPC_HOST_VIEWS")) return read_offset_string_column(f, bsel, col, total, large, dict_out);
  // views are written row by row below (every row, in parallel slices): no 16 B/row zero fill by one thread first
  std::unique_ptr<uint8_t[]> views_mem(new uint8_t[(size_t)total * 16 + 16]);
  uint8_t* const views = views_mem.get();
  std::vector<uint8_t> validity((size_t)(total + 7) / 8 + 8, 0);
  std::vector<std::vector<uint8_t>> data;
  bool any_nulls = false;
  int64_t row = 0;
  for (int b : bsel) {
    const ipc::BatchMeta& bm = f.batches[b];
    const Slot s = slot_of(f, bm, col);
    const int64_t n = bm.length, body = f.body_off[b];
    if (s.node >= bm.nodes.size() || s.buf + (is_view ? 2 : 3) > bm.buffers.size()) throw ipc::FormatError("record batch without the column's buffers");
    if (bm.nodes[s.node].length != n) throw ipc::FormatError("column length differs from its record batch");
    const int64_t nc = bm.nodes[s.node].null_count;
    std::vector<uint8_t> vbits;
    if (nc > 0) {
      any_nulls = true;
      vbits = read_buffer(f, bm, body, bm.buffers[s.buf]);
      if ((int64_t)vbits.size() - 16 < (n + 7) / 8) throw ipc::FormatError("bitmap buffer shorter than the array");
    }
    if (nc == 0) set_bits(validity.data(), row, n);
    else
      for (int64_t i = 0; i < n; i++)
        if ((vbits[(size_t)i >> 3] >> (i & 7)) & 1) validity[(size_t)(row + i) >> 3] |= (uint8_t)(1u << ((row + i) & 7));
    const uint32_t base = (uint32_t)data.size();
    if (is_view) {
      const int64_t nvar = s.variadic < bm.variadic_counts.size() ? bm.variadic_counts[s.variadic] : 0;
      if (s.buf + 2 + (size_t)nvar > bm.buffers.size()) throw ipc::FormatError("view array without its data buffers");
      std::vector<uint8_t> v = read_buffer(f, bm, body, bm.buffers[s.buf + 1]);
      if ((int64_t)v.size() - 16 < n * 16) throw ipc::FormatError("views buffer shorter than the array");
      for (int64_t k = 0; k < nvar; k++) data.push_back(read_buffer(f, bm, body, bm.buffers[s.buf + 2 + (size_t)k]));
      parallel_rows(n, [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; i++) {
        uint8_t* dst = views + 16 * (size_t)(row + i);
        memcpy(dst, v.data() + 16 * i, 16);
        uint32_t len, bi, off;
        memcpy(&len, dst, 4);
        const bool ok = nc == 0 || ((vbits[(size_t)i >> 3] >> (i & 7)) & 1);
        if (!ok) { memset(dst, 0, 16); continue; }
        if (len > 12) {
          memcpy(&bi, dst + 8, 4); memcpy(&off, dst + 12, 4);
          if ((int64_t)bi >= nvar || (uint64_t)off + len > data[base + bi].size() - 16) throw ipc::FormatError("view points outside its data buffer");
          bi += base;
          memcpy(dst + 8, &bi, 4);
        }
      }
      });
    } else {
      std::vector<uint8_t> offs = read_buffer(f, bm, body, bm.buffers[s.buf + 1]);
      data.push_back(read_buffer(f, bm, body, bm.buffers[s.buf + 2]));
      const std::vector<uint8_t>& d = data.back();
      const size_t ow = large ? 8 : 4;
      if (n && (int64_t)offs.size() - 16 < (n + 1) * (int64_t)ow) throw ipc::FormatError("offsets buffer shorter than the array");
      if ((int64_t)d.size() - 16 >= ((int64_t)1 << 32)) throw Unsupported("string data buffer of 4 GiB or more in one record batch");
      parallel_rows(n, [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; i++) {
        int64_t a, e;
        if (large) { memcpy(&a, offs.data() + 8 * i, 8); memcpy(&e, offs.data() + 8 * (i + 1), 8); }
        else { int32_t a32, e32; memcpy(&a32, offs.data() + 4 * i, 4); memcpy(&e32, offs.data() + 4 * (i + 1), 4); a = a32; e = e32; }
        if (a < 0 || e < a || e > (int64_t)d.size() - 16 || e - a > 0x7fffffff) throw ipc::FormatError("string offsets outside the data buffer");
        const bool ok = nc == 0 || ((vbits[(size_t)i >> 3] >> (i & 7)) & 1);
        uint8_t* dst = views + 16 * (size_t)(row + i);
        memset(dst, 0, 16);                     // null rows and the unused inline bytes of short strings are zero (views are compared whole)
        if (!ok) continue;
        const uint32_t len = (uint32_t)(e - a), off = (uint32_t)a;
        memcpy(dst, &len, 4);
        if (len <= 12) memcpy(dst + 4, d.data() + a, len);
        else { memcpy(dst + 4, d.data() + a, 4); memcpy(dst + 8, &base, 4); memcpy(dst + 12, &off, 4); }
      }
      });
    }
    row += n;
  }
  std::vector<const void*> ptrs;
  std::vector<int64_t> sizes;
  for (const std::vector<uint8_t>& d : data) { ptrs.push_back(d.data()); sizes.push_back((int64_t)d.size() - 16); }
  plx_column codes = 0;
  plx_strdict dict = 0;
  int rc = plx_strview_dict_encode(views, any_nulls ? validity.data() : nullptr, 0, total, ptrs.empty() ? nullptr : ptrs.data(), sizes.empty() ? nullptr : sizes.data(),
                                   (int32_t)ptrs.size(), &codes, &dict);
  if (rc != PLX_OK) fail(rc, plx_last_error());
  ColumnPtr c = get_column(codes);
  free_column(codes);       // the frame keeps the column alive
  *dict_out = dict;
  return c;
}

std::mutex g_mu;
std::vector<std::unique_ptr<File>> g_files;   // handle = index + 1
std::vector<std::unique_ptr<std::mutex>> g_file_mu;   // one reader at a time per handle (reads fill the handle's dictionary caches)
std::mutex& file_mutex(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (h == 0 || h > g_file_mu.size() || !g_file_mu[h - 1]) fail(PLX_ERR_INVALID, "invalid ipc handle");
  return *g_file_mu[h - 1];
}
File& get_file(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (h == 0 || h > g_files.size() || !g_files[h - 1]) fail(PLX_ERR_INVALID, "invalid ipc handle");
  return *g_files[h - 1];
}
thread_local std::string t_name;

}  // namespace

#define IPC_TRY try {
#define IPC_CATCH                                                                                   \
  }                                                                                                 \
  catch (const plx::Error& e) { plx::set_last_error(e.msg); return e.code; }                         \
  catch (const Unsupported& e) { plx::set_last_error(std::string("ipc: ") + e.what()); return PLX_ERR_UNSUPPORTED; }   \
  catch (const ipc::FormatError& e) { plx::set_last_error(std::string("ipc: ") + e.what()); return PLX_ERR_INVALID; } \
  catch (const plx::IoError& e) { plx::set_last_error(std::string("ipc: ") + e.what()); return PLX_ERR_INVALID; }     \
  catch (const std::bad_alloc&) { plx::set_last_error("host out of memory"); return PLX_ERR_OOM; }  \
  catch (const std::exception& e) { plx::set_last_error(std::string("PANIC: ") + e.what()); return PLX_ERR_INVALID; } \
  catch (...) { plx::set_last_error("PANIC"); return PLX_ERR_INVALID; }                              \
  return PLX_OK;

extern "C" {

int plx_ipc_open(const char* path, plx_ipc* out) {
  IPC_TRY
  PLX_REQUIRE(path && out, PLX_ERR_INVALID, "null argument");
  std::unique_ptr<File> f = open_file(path);
  std::lock_guard<std::mutex> lk(g_mu);
  g_files.push_back(std::move(f));
  g_file_mu.push_back(std::make_unique<std::mutex>());
  *out = (plx_ipc)g_files.size();
  IPC_CATCH
}

int plx_ipc_close(plx_ipc file) {
  IPC_TRY
  std::lock_guard<std::mutex> lk(g_mu);
  if (file && file <= g_files.size()) g_files[file - 1].reset();
  IPC_CATCH
}

int plx_ipc_shape(plx_ipc file, int64_t* num_rows, int32_t* num_batches, int32_t* num_columns) {
  IPC_TRY
  File& f = get_file(file);
  if (num_rows) *num_rows = f.num_rows;
  if (num_batches) *num_batches = (int32_t)f.batches.size();
  if (num_columns) *num_columns = (int32_t)f.footer.fields.size();
  IPC_CATCH
}

int plx_ipc_column_info(plx_ipc file, int32_t column, const char** name, int32_t* dtype, int32_t* logical, int32_t* nullable) {
  IPC_TRY
  File& f = get_file(file);
  PLX_REQUIRE(column >= 0 && (size_t)column < f.footer.fields.size(), PLX_ERR_INVALID, "ipc column index out of range");
  const ipc::Field& fl = f.footer.fields[column];
  const ColType t = col_type(fl);
  if (name) { t_name = fl.name; *name = t_name.c_str(); }
  if (dtype) *dtype = t.dtype;
  if (logical) *logical = t.logical;
  if (nullable) *nullable = fl.nullable ? 1 : 0;
  IPC_CATCH
}

int plx_ipc_column_timezone(plx_ipc file, int32_t column, const char** timezone) {
  IPC_TRY
  File& f = get_file(file);
  PLX_REQUIRE(column >= 0 && (size_t)column < f.footer.fields.size() && timezone, PLX_ERR_INVALID, "ipc column index out of range");
  t_name = f.footer.fields[column].timezone;
  *timezone = t_name.c_str();
  IPC_CATCH
}

int plx_ipc_batch_info(plx_ipc file, int32_t batch, int64_t* num_rows, int64_t* body_bytes, int32_t* compressed) {
  IPC_TRY
  File& f = get_file(file);
  PLX_REQUIRE(batch >= 0 && (size_t)batch < f.batches.size(), PLX_ERR_INVALID, "ipc record batch index out of range");
  if (num_rows) *num_rows = f.batches[batch].length;
  if (body_bytes) *body_bytes = f.footer.batches[batch].body_len;
  if (compressed) *compressed = f.batches[batch].compressed ? 1 + f.batches[batch].codec : 0;
  IPC_CATCH
}

int plx_ipc_read(plx_ipc file, const int32_t* batches, int32_t n_batches, const int32_t* columns, int32_t n_columns, plx_frame* out) {
  IPC_TRY
  PLX_REQUIRE(out && (columns || n_columns == 0) && (batches || n_batches == 0), PLX_ERR_INVALID, "null argument");
  File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  device();   // fails loudly without a GPU
  std::vector<int> bsel(batches, batches + n_batches);
  int64_t total = 0;
  for (int b : bsel) {
    PLX_REQUIRE(b >= 0 && (size_t)b < f.batches.size(), PLX_ERR_INVALID, "ipc record batch index out of range");
    if (f.batches[b].compressed && f.batches[b].codec != 0 && f.batches[b].codec != 1) throw Unsupported("record batch body compressed with an unknown codec");
    total += f.batches[b].length;
  }
  auto frame = std::make_shared<Frame>();
  frame->height = total;
  try {
    for (int32_t i = 0; i < n_columns; i++) {
      check_cancel();
      const int col = columns[i];
      PLX_REQUIRE(col >= 0 && (size_t)col < f.footer.fields.size(), PLX_ERR_INVALID, "ipc column index out of range");
      const ipc::Field& fl = f.footer.fields[col];
      const ColType ct = col_type(fl);
      if (ct.dtype < 0) throw Unsupported("column '" + fl.name + "': " + ct.why + " is outside the hot path's dtypes");
      static const bool timing = getenv("PLX_IPC_TIMING") != nullptr;      // per-column host time on stderr (measurement only)
      const auto t_col = std::chrono::steady_clock::now();
      struct Report { bool on; const std::string& name; std::chrono::steady_clock::time_point t0; int64_t rows;
                      ~Report() { if (on) fprintf(stderr, "[plx ipc] column %-20s %8.2f ms host (%lld rows)\n", name.c_str(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (long long)rows); } } report{timing, fl.name, t_col, total};
      ColumnPtr c;
      if (ct.strings && !fl.has_dictionary) {
        plx_strdict d = 0;
        c = read_string_column(f, bsel, col, total, &d);
        auto it = f.strdicts.find(col);
        if (it != f.strdicts.end() && it->second) plx_strdict_free(it->second);
        f.strdicts[col] = d;
      } else {
        if (fl.has_dictionary) load_dictionaries(f);
        c = read_fixed_column(f, bsel, col, ct, total);
        if (fl.has_dictionary) {
          auto it = f.dicts.find(fl.dict_id);
          const int64_t nd = it == f.dicts.end() ? 0 : (int64_t)it->second.size();
          if (nd > 0) { c->range_state = 1; c->range_min = 0; c->range_max = nd - 1; c->range_trusted = false; }   // declared, checked by the kernels
        }
      }
      frame->names.push_back(fl.name);
      frame->cols.push_back(c);
    }
    PLX_HIP(hipStreamSynchronize(stream()));     // staging buffers and temporaries of the last column are done
  } catch (...) {
    (void)hipStreamSynchronize(stream());
    throw;
  }
  *out = register_frame(frame);
  IPC_CATCH
}

int plx_ipc_read_string_views(plx_ipc file, const int32_t* batches, int32_t n_batches, int32_t column, plx_column* out_views, plx_column* out_data) {
  IPC_TRY
  PLX_REQUIRE(out_views && out_data && (batches || n_batches == 0), PLX_ERR_INVALID, "null argument");
  File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  PLX_REQUIRE(column >= 0 && (size_t)column < f.footer.fields.size(), PLX_ERR_INVALID, "ipc column index out of range");
  const ipc::Field& fl = f.footer.fields[column];
  const ColType ct = col_type(fl);
  const bool is_view = fl.type == ipc::TY_UTF8_VIEW || fl.type == ipc::TY_BINARY_VIEW;
  if (!ct.strings || fl.has_dictionary || is_view) throw Unsupported("column '" + fl.name + "': views are handed out for Utf8 / LargeUtf8 / Binary columns that are not dictionary-encoded in the file");
  std::vector<int> bsel(batches, batches + n_batches);
  int64_t total = 0;
  for (int b : bsel) {
    PLX_REQUIRE(b >= 0 && (size_t)b < f.batches.size(), PLX_ERR_INVALID, "ipc record batch index out of range");
    if (f.batches[b].compressed && f.batches[b].codec != 0 && f.batches[b].codec != 1) throw Unsupported("record batch body compressed with an unknown codec");
    total += f.batches[b].length;
  }
  device();   // (after the checks that need no device) fails loudly without a GPU
  RawViews raw;
  try {
    read_offset_string_column(f, bsel, column, total, fl.type == ipc::TY_LARGE_UTF8 || fl.type == ipc::TY_LARGE_BINARY, nullptr, &raw);
    PLX_HIP(hipStreamSynchronize(stream()));
  } catch (...) {
    (void)hipStreamSynchronize(stream());
    throw;
  }
  *out_views = register_column(raw.views);
  *out_data = register_column(raw.data);
  IPC_CATCH
}

int plx_ipc_categories(plx_ipc file, int32_t column, int64_t* n_strings, int64_t* total_bytes) {
  IPC_TRY
  File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  PLX_REQUIRE(column >= 0 && (size_t)column < f.footer.fields.size(), PLX_ERR_INVALID, "ipc column index out of range");
  const ipc::Field& fl = f.footer.fields[column];
  PLX_REQUIRE(fl.has_dictionary && col_type(fl).dtype >= 0, PLX_ERR_NOT_FOUND, "not a dictionary-encoded string column (strings encoded on the device: plx_ipc_column_strdict)");
  load_dictionaries(f);
  const std::vector<std::string>& d = f.dicts[fl.dict_id];
  int64_t b = 0;
  for (const std::string& s : d) b += (int64_t)s.size();
  if (n_strings) *n_strings = (int64_t)d.size();
  if (total_bytes) *total_bytes = b;
  IPC_CATCH
}

int plx_ipc_categories_to_host(plx_ipc file, int32_t column, int64_t* offsets, uint8_t* bytes) {
  IPC_TRY
  File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  PLX_REQUIRE(column >= 0 && (size_t)column < f.footer.fields.size(), PLX_ERR_INVALID, "ipc column index out of range");
  const ipc::Field& fl = f.footer.fields[column];
  PLX_REQUIRE(fl.has_dictionary && col_type(fl).dtype >= 0 && offsets, PLX_ERR_NOT_FOUND, "not a dictionary-encoded string column");
  load_dictionaries(f);
  int64_t off = 0, i = 0;
  for (const std::string& s : f.dicts[fl.dict_id]) {
    offsets[i++] = off;
    if (!s.empty()) { PLX_REQUIRE(bytes, PLX_ERR_INVALID, "null bytes pointer"); memcpy(bytes + off, s.data(), s.size()); }
    off += (int64_t)s.size();
  }
  offsets[i] = off;
  IPC_CATCH
}

int plx_ipc_column_strdict(plx_ipc file, int32_t column, plx_strdict* out) {
  IPC_TRY
  File& f = get_file(file);
  std::lock_guard<std::mutex> reading(file_mutex(file));
  PLX_REQUIRE(out, PLX_ERR_INVALID, "null out pointer");
  auto it = f.strdicts.find(column);
  PLX_REQUIRE(it != f.strdicts.end() && it->second, PLX_ERR_NOT_FOUND, "no device dictionary: the column has not been read (or is dictionary-encoded in the file: plx_ipc_categories)");
  *out = it->second;
  it->second = 0;       // ownership moves to the caller (plx_strdict_free)
  IPC_CATCH
}

}  // extern "C"
