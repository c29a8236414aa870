// parquet_reader.hpp -- Parquet column chunks -> device columns, host orchestration.
//
// The host only touches METADATA (and, for the codecs without a device kernel -- GZIP, LZ4_RAW -- decompresses pages on a pool of
// threads: host_codecs.hpp): the footer (parquet_format.hpp), the Thrift page headers inside each column chunk, the block headers and table
// descriptions inside zstd pages (parquet_zstd_index.hpp), and the
// dictionary pages of string columns (a few KB each, unified into one column-wide dictionary).  The chunk bytes themselves go to HBM
// exactly as they are in the file -- compressed, encoded -- in one copy per chunk, and everything else happens there
// (parquet_device.hpp): Snappy, Zstandard, level / index run tables, validity, dense -> row expansion, dictionary lookup, integer narrowing.
//
// `read_column<B>` is a template over the execution backend B (HBM + kernel launches in parquet.cpp; host memory + the same bodies
// run thread by thread in tests/emu/parquet_emu.cpp), so the page walk, the stream planning and the dictionary handling below are
// what the CPU tests execute.
//
// Reference shape: crates/polars-parquet/src/parquet/read/page/reader.rs:183-300 (page iteration of a chunk),
// parquet/read/compression.rs:70-135 (decompress per page), arrow/read/deserialize/{primitive,boolean,dictionary_encoded,binview}
// (decode per page into an Arrow array), crates/polars-io/src/parquet/read/read_impl.rs (row groups x projected columns).
#pragma once
#include <mutex>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>
#include <exception>
#include <string>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/polars_amd.h"
#include "file_io.hpp"
#include <future>

#include "host_codecs.hpp"
#include "parquet_device.hpp"
#include "parquet_format.hpp"
#include "parquet_zstd_index.hpp"

namespace plx {
namespace pq {

struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- file ---------------------------------------------------------------------------------------------------------------------------
struct File : FileReader {
  FileMetaData md;
  // column-wide dictionaries of the string columns read last (leaf index -> categories)
  std::unordered_map<int, std::vector<std::string>> categories;
  // ... or, for string columns with PLAIN pages (dictionary built on the device), the handle of that dictionary (plx_strdict; 0: none)
  std::unordered_map<int, uint64_t> strdicts;
  // the columns of one read run on several host threads (parquet.cpp: one HIP stream each): the two maps above are only touched under this lock
  std::mutex meta_mu;
};

inline std::unique_ptr<File> open_file(const std::string& path) {
  auto f = std::make_unique<File>();
  f->open(path);
  if (f->size < 12) throw FormatError(path + " is too small to be a Parquet file");
  uint8_t tail[8];
  f->pread_exact(tail, 8, f->size - 8);
  if (memcmp(tail + 4, "PAR1", 4) != 0) {
    if (memcmp(tail + 4, "PARE", 4) == 0) throw Unsupported("encrypted Parquet footer");
    throw FormatError(path + " does not end with the PAR1 magic");
  }
  uint32_t flen = load_u32(tail);
  if ((int64_t)flen + 12 > f->size) throw FormatError("footer length past the start of the file");
  std::vector<uint8_t> footer(flen);
  f->pread_exact(footer.data(), flen, f->size - 8 - flen);
  f->md = parse_file_metadata(footer.data(), flen);
  return f;
}

// ---- leaf -> device dtype -------------------------------------------------------------------------------------------------------------
enum LogicalOut { LO_NONE = 0, LO_DATE = 1, LO_DATETIME_US = 2, LO_STRING = 3, LO_BINARY = 4, LO_DATETIME_MS = 5, LO_DATETIME_NS = 6 };

struct LeafType {
  int dtype = -1;          // plx_dtype, -1: outside the hot path (why says what it is)
  int logical = LO_NONE;
  uint32_t src_width = 0;  // bytes per PLAIN value in the file (0: bit-packed booleans / byte arrays)
  std::string why;
};

inline LeafType leaf_type(const Leaf& l) {
  LeafType t;
  if (l.nested) { t.why = "nested / repeated column"; return t; }
  switch (l.type) {
    case PT_BOOLEAN: t.dtype = PLX_BOOL; t.src_width = 0; return t;
    case PT_INT32:
      t.src_width = 4;
      if (l.logical == LG_NONE) { t.dtype = PLX_I32; return t; }
      if (l.logical == LG_DATE) { t.dtype = PLX_I32; t.logical = LO_DATE; return t; }
      if (l.logical == LG_INT && l.int_bits <= 32) {
        t.dtype = l.int_bits == 8 ? (l.int_signed ? PLX_I8 : PLX_U8) : l.int_bits == 16 ? (l.int_signed ? PLX_I16 : PLX_U16) : (l.int_signed ? PLX_I32 : PLX_U32);
        return t;
      }
      t.why = l.logical == LG_DECIMAL ? "decimal" : "annotated INT32";
      return t;
    case PT_INT64:
      t.src_width = 8;
      if (l.logical == LG_NONE) { t.dtype = PLX_I64; return t; }
      if (l.logical == LG_INT && l.int_bits == 64) { t.dtype = l.int_signed ? PLX_I64 : PLX_U64; return t; }
      if (l.logical == LG_TIMESTAMP_MICROS) { t.dtype = PLX_I64; t.logical = LO_DATETIME_US; return t; }
      if (l.logical == LG_TIMESTAMP_MILLIS) { t.dtype = PLX_I64; t.logical = LO_DATETIME_MS; return t; }       // the stored unit is kept, as the reference keeps it
      if (l.logical == LG_TIMESTAMP_NANOS) { t.dtype = PLX_I64; t.logical = LO_DATETIME_NS; return t; }        // (Datetime("ms" | "us" | "ns"): schema/convert.rs)
      t.why = l.logical == LG_DECIMAL ? "decimal" : "annotated INT64";
      return t;
    case PT_FLOAT: t.dtype = PLX_F32; t.src_width = 4; return t;
    case PT_DOUBLE: t.dtype = PLX_F64; t.src_width = 8; return t;
    case PT_BYTE_ARRAY:
      if (l.logical == LG_STRING || l.logical == LG_NONE) { t.dtype = PLX_U32; t.logical = l.logical == LG_STRING ? LO_STRING : LO_BINARY; return t; }
      t.why = "annotated BYTE_ARRAY";
      return t;
    // legacy timestamps, converted by host threads (read_fixed_column_host) to NANOSECONDS like the reference's default
    // (int96_coerce_to_timeunit = Nanosecond, crates/polars-parquet/src/arrow/read/schema/mod.rs:32)
    case PT_INT96: t.dtype = PLX_I64; t.logical = LO_DATETIME_NS; t.src_width = 12; return t;
    default: t.why = l.logical == LG_DECIMAL ? "decimal (FIXED_LEN_BYTE_ARRAY)" : "FIXED_LEN_BYTE_ARRAY"; return t;
  }
}

// host threads for page-sized tasks: half the hardware threads (the other half is the caller's: other columns, the GPU driver), 2 .. 64 (PLX_HOST_THREADS overrides the cap; 128 measured
// no faster on a 256-thread host).  A thread that reads one column of several at a time (parquet.cpp ColumnWorkers) lowers its own cap to its share: 4 columns x 32 threads inflate a zstd
// file in 34-36 ms where 4 x 64 take 36-53 and 1 x 64 takes 142 (tools/scan_threads.py, 2e7 rows).
inline thread_local size_t host_thread_share = 0;       // 0 = this thread has the host to itself
inline size_t host_threads(size_t tasks) {
  static const size_t cap = [] { const char* e = getenv("PLX_HOST_THREADS"); const long v = e ? atol(e) : 0; return v >= 1 && v <= 1024 ? (size_t)v : (size_t)0; }();
  const size_t mine = cap ? cap : host_thread_share ? host_thread_share : (size_t)64;
  return std::min<size_t>(std::min<size_t>(mine, std::max(2u, std::thread::hardware_concurrency() / 2)), tasks);
}

// PLX_PARQUET_TRACE=1: where the host thread of a column is, in ms since the process' first traced event (stderr; measurement only)
inline bool trace_on() { static const bool on = getenv("PLX_PARQUET_TRACE") != nullptr; return on; }
inline void trace_point(const std::string& column, const char* what) {
  const bool on = trace_on();
  if (!on) return;
  static const auto t0 = std::chrono::steady_clock::now();
  fprintf(stderr, "[plx parquet] %9.2f ms  %-18s %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), column.c_str(), what);
}

inline uint32_t out_width_of(int dtype) {
  switch (dtype) {
    case PLX_BOOL: return 0;
    case PLX_I8: case PLX_U8: return 1;
    case PLX_I16: case PLX_U16: return 2;
    case PLX_I32: case PLX_U32: case PLX_F32: return 4;
    default: return 8;
  }
}

// ---- host Snappy (string dictionary pages; pages of the columns that host threads decode) ---------------------------------------------------------------------------
inline void snappy_decompress_into(const uint8_t* in, size_t n, uint8_t* out, size_t expect);
inline std::vector<uint8_t> snappy_decompress_host(const uint8_t* in, size_t n, size_t expect) {
  std::vector<uint8_t> out(expect);
  snappy_decompress_into(in, n, out.data(), expect);
  return out;
}
inline void snappy_decompress_into(const uint8_t* in, size_t n, uint8_t* out, size_t expect) {
  size_t pos = 0, out_len = 0;
  for (int shift = 0;; shift += 7) {
    if (pos >= n || shift > 28) throw FormatError("snappy: bad preamble");
    uint8_t b = in[pos++];
    out_len |= (size_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) break;
  }
  if (out_len != expect) throw FormatError("snappy: uncompressed length differs from the page header");
  size_t o = 0;
  while (pos < n) {
    uint8_t tag = in[pos++];
    size_t len, off = 0;
    if ((tag & 3) == 0) {
      len = (tag >> 2) + 1;
      if (len > 60) {
        size_t nb = len - 60;
        if (pos + nb > n) throw FormatError("snappy: truncated literal length");
        len = 0;
        for (size_t b = 0; b < nb; b++) len |= (size_t)in[pos + b] << (8 * b);
        len += 1; pos += nb;
      }
      if (len > n - pos || len > out_len - o) throw FormatError("snappy: literal past the end");
      if (len <= 16 && n - pos >= 16 && out_len - o >= 16) codec::copy16(out + o, in + pos);
      else memcpy(out + o, in + pos, len);
      pos += len; o += len;
      continue;
    }
    if ((tag & 3) == 1) {
      if (pos + 1 > n) throw FormatError("snappy: truncated copy");
      len = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[pos]; pos += 1;
    } else if ((tag & 3) == 2) {
      if (pos + 2 > n) throw FormatError("snappy: truncated copy");
      len = (tag >> 2) + 1; off = in[pos] | ((size_t)in[pos + 1] << 8); pos += 2;
    } else {
      if (pos + 4 > n) throw FormatError("snappy: truncated copy");
      len = (tag >> 2) + 1; off = load_u32(in + pos); pos += 4;
    }
    if (off == 0 || off > o || len > out_len - o) throw FormatError("snappy: bad back-reference");
    codec::match_copy(out + o, off, len, out_len - o - len);
    o += len;
  }
  if (o != out_len) throw FormatError("snappy: stream ends early");
}

// ---- read one column --------------------------------------------------------------------------------------------------------------------
struct ReadStats {
  uint64_t file_bytes = 0, data_pages = 0, dict_pages = 0, snappy_streams = 0, snappy_bytes_out = 0, run_entries = 0, host_inflated_pages = 0, host_inflated_bytes = 0;
  uint64_t zstd_streams = 0, zstd_blocks = 0, zstd_bytes_out = 0;
};

template <class B> struct ColumnResult {
  typename B::Mem values;
  typename B::Mem validity;      // empty: no nulls
  bool has_validity = false;
  int dtype = PLX_I64;
  int logical = LO_NONE;
  int64_t len = 0;
  int64_t null_count = 0;
};

inline std::string error_bits_text(uint32_t e) {
  std::string s;
  auto add = [&](uint32_t bit, const char* t) { if (e & bit) { if (!s.empty()) s += ", "; s += t; } };
  add(PE_LEVELS, "definition levels do not cover the page");
  add(PE_RUNS, "malformed RLE / bit-packed run header");
  add(PE_VALUES, "value bytes missing");
  add(PE_DICT_INDEX, "dictionary index out of range");
  add(PE_SNAPPY, "malformed Snappy stream");
  add(PE_DEF_LEVEL, "definition level > 1 in a flat column");
  add(PE_ZSTD, "malformed zstd stream");
  return s;
}

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// pages of the codecs without a device kernel are inflated by host threads (host_codecs.hpp)
// Which codecs are inflated by host threads.  GZIP / LZ4_RAW always (no device kernel).  ZSTD has one since round 6 (pq_zstd_entropy + pq_zstd_execute, parquet_zstd.hpp), the
// default; PLX_PARQUET_ZSTD=host keeps its pages on the host threads.  SNAPPY has one (pq_snappy), the default;
// PLX_PARQUET_SNAPPY=host sends Snappy pages through the host threads too -- on a many-core host the column-wide parallel inflate may
// outrun the device kernel, whose time is set by the longest single stream; an experiment switch until both are timed.
inline bool snappy_on_host() {
  const char* e = getenv("PLX_PARQUET_SNAPPY");        // read per column chunk: cheap, and a test can flip it
  return e && !strcmp(e, "host");
}
// Snappy dictionary pages from this many uncompressed bytes are inflated by host threads instead of the device kernel (PLX_PARQUET_HOST_DICT_BYTES; 0 = never)
inline size_t host_dict_min_bytes() {
  const char* e = getenv("PLX_PARQUET_HOST_DICT_BYTES");        // read per page: cheap, and a test can flip it
  if (e) { const long long v = atoll(e); return v <= 0 ? (size_t)-1 : (size_t)v; }
  return (size_t)192 << 10;
}
inline bool zstd_on_host() {
  const char* e = getenv("PLX_PARQUET_ZSTD");          // read per column chunk: cheap, and a test can flip it
  return e && !strcmp(e, "host");
}
// A zstd page's execute pass is ONE wavefront walking the page's sequences in order (parquet_zstd.hpp): a 1 MB page of sorted keys -- what a writer with 1 MB pages, the
// reference's own, produces -- is 1.3e5 of them, several milliseconds that no other wavefront can shorten.  A page with more sequences than this (counted by the index
// pass; PLX_PARQUET_ZSTD_HOST_SEQS, 0 = no limit) is inflated by a host thread instead, while the column's device passes run.
inline size_t zstd_host_sequences() {
  const char* e = getenv("PLX_PARQUET_ZSTD_HOST_SEQS");        // read per page: cheap, and a test can flip it
  if (e) { const long long v = atoll(e); return v <= 0 ? (size_t)-1 : (size_t)v; }
  return 50000;
}
inline bool is_host_codec(int codec_id) {
  return (codec_id == CODEC_ZSTD && zstd_on_host()) || codec_id == CODEC_LZ4_RAW || codec_id == CODEC_GZIP || (codec_id == CODEC_SNAPPY && snappy_on_host());
}

inline void host_inflate(int codec_id, const uint8_t* src, size_t n, uint8_t* dst, size_t out) {
  if (codec_id == CODEC_SNAPPY) { if (out || n) snappy_decompress_into(src, n, dst, out); }
  else if (codec_id == CODEC_ZSTD) codec::zstd_decompress(src, n, dst, out);
  else if (codec_id == CODEC_GZIP) codec::gzip_decompress(src, n, dst, out);
  else codec::lz4_raw_decompress(src, n, dst, out);
}

// thrown by the device reader when a string column turns out to hold PLAIN (not dictionary-encoded) data pages: read_column then
// takes the host-views path below
struct NeedsHostStrings {};
// ... and when a fixed-width column holds pages in an encoding only the host decoder knows (DELTA_BINARY_PACKED, BYTE_STREAM_SPLIT; INT96)
struct NeedsHostValues {};

template <class B> ColumnResult<B> read_column_device(B& be, File& f, const std::vector<int>& row_groups, int leaf_idx, ReadStats* stats) {
  const FileMetaData& md = f.md;
  if (leaf_idx < 0 || (size_t)leaf_idx >= md.leaves.size()) throw FormatError("column index out of range");
  const Leaf& leaf = md.leaves[leaf_idx];
  const LeafType lt = leaf_type(leaf);
  if (lt.dtype < 0) throw Unsupported("column '" + leaf.name + "': " + lt.why + " is outside the hot path's dtypes");
  const bool is_bytes = leaf.type == PT_BYTE_ARRAY;
  const bool optional = leaf.repetition == REP_OPTIONAL;
  trace_point(leaf.name, "start");

  ColumnResult<B> res;
  res.dtype = lt.dtype; res.logical = lt.logical;

  // -- plan: chunk placement in the blob ------------------------------------------------------------------------------------------------
  struct ChunkRef { const ColumnChunk* c; int64_t rows; size_t blob_off, blob_cap; };
  std::vector<ChunkRef> chunks;
  size_t blob_bytes = 0;
  int64_t n_rows = 0;
  bool nulls_possible = false;
  for (int g : row_groups) {
    if (g < 0 || (size_t)g >= md.row_groups.size()) throw FormatError("row group index out of range");
    const RowGroup& rg = md.row_groups[g];
    const ColumnChunk& c = rg.columns[leaf_idx];
    if (!c.has_meta) throw FormatError("column chunk without metadata");
    if (c.external_file) throw Unsupported("column chunk stored in another file");
    if (c.type != leaf.type) throw FormatError("column chunk type differs from the schema");
    const bool host_codec = is_host_codec(c.codec);       // decompressed by host threads (host_codecs.hpp)
    if (c.codec != CODEC_UNCOMPRESSED && c.codec != CODEC_SNAPPY && c.codec != CODEC_ZSTD && !host_codec)
      throw Unsupported(std::string("column '") + leaf.name + "': codec " + codec_name(c.codec) + " has no decompressor here (UNCOMPRESSED, SNAPPY, ZSTD, GZIP and LZ4_RAW do)");
    if (c.num_values != rg.num_rows) throw FormatError("flat column chunk whose value count differs from the row group's rows");
    if (rg.num_rows == 0) continue;            // an empty row group has nothing to fetch (writers leave its data page offset at 0)
    if (c.start() < 4 || c.total_compressed_size < 0 || c.start() + c.total_compressed_size > f.size - 8) throw FormatError("column chunk outside the file");
    // what goes to HBM: the chunk as stored, or -- host codecs -- its image with every page payload decompressed (the metadata's
    // uncompressed total, page headers included, bounds it)
    // (no codec here expands more than ~32768 : 1; a corrupt total must not become an allocation)
    if (host_codec && (c.total_uncompressed_size < 0 || c.total_uncompressed_size > ((int64_t)1 << 36) || c.total_uncompressed_size > 65536 * c.total_compressed_size + 4096))
      throw FormatError("column chunk with an absurd uncompressed size");
    const size_t cap = align16((size_t)(host_codec ? c.total_uncompressed_size : c.total_compressed_size));
    chunks.push_back({&c, rg.num_rows, blob_bytes, cap});
    blob_bytes += cap;
    n_rows += rg.num_rows;
    if (optional && !(c.stats.has_null_count && c.stats.null_count == 0)) nulls_possible = true;
  }
  res.len = n_rows;
  if (n_rows >= (int64_t)1 << 40) throw Unsupported("more than 2^40 rows in one read");
  const uint32_t out_width = out_width_of(lt.dtype);
  auto out_bytes = [&](int64_t n) { return lt.dtype == PLX_BOOL ? (size_t)(((n + 63) / 64) * 8 + 8) : (size_t)n * out_width; };
  if (n_rows == 0) {
    res.values = be.alloc(out_bytes(0));
    if (is_bytes) { std::lock_guard<std::mutex> lk(f.meta_mu); f.categories[leaf_idx].clear(); }
    return res;
  }

  typename B::Mem blob = be.alloc(blob_bytes + 64);
  const uint64_t blob_addr = be.addr(blob);

  // -- page walk (host, headers only) + upload of the chunk bytes as they are -------------------------------------------------------------
  std::vector<PageDesc> pages;
  std::vector<DictDesc> dicts;
  std::vector<DecompJob> jobs;
  std::vector<size_t> job_of_page;            // per page: index into jobs or npos
  std::vector<size_t> job_of_dict;            // per dict
  std::vector<uint32_t> remap;                // string columns: concatenated chunk-dictionary -> column code tables
  std::vector<size_t> remap_base_of_dict;     // per dict (strings)
  std::vector<std::string> categories;
  std::unordered_map<std::string, uint32_t> cat_index;
  const size_t npos = (size_t)-1;
  uint64_t row0 = 0;
  // Host codecs: pages are inflated by host threads.  When every chunk of the read uses one (the normal case: a file has one codec), the
  // pages of ALL chunks are inflated in one parallel pass after the page walk -- a chunk alone has too few pages to keep the cores busy --
  // into one page-locked image of the column, laid out like the device blob.  Mixed columns fall back to chunk-by-chunk.
  struct Inflate { const uint8_t* src; size_t n; uint8_t* dst; size_t out; bool copy; int codec; };
  auto run_inflate = [&](const std::vector<Inflate>& tasks) {
    const size_t threads = host_threads(tasks.size());
    std::vector<std::exception_ptr> errs(std::max<size_t>(threads, 1));
    std::atomic<size_t> next{0};
    auto work = [&](size_t t) {
      try {
        for (size_t i = next.fetch_add(1); i < tasks.size(); i = next.fetch_add(1)) {
          const Inflate& j = tasks[i];
          if (j.copy) { if (j.n != j.out) throw FormatError("uncompressed part of a page whose two sizes differ"); if (j.n) memcpy(j.dst, j.src, j.n); }
          else host_inflate(j.codec, j.src, j.n, j.dst, j.out);
        }
      } catch (...) { errs[t] = std::current_exception(); }
    };
    if (threads <= 1) { if (!tasks.empty()) work(0); }
    else {
      std::vector<std::thread> pool;
      for (size_t t = 0; t < threads; t++) pool.emplace_back(work, t);
      for (std::thread& th : pool) th.join();
    }
    for (std::exception_ptr& ep : errs)
      if (ep) {
        try { std::rethrow_exception(ep); }
        catch (const codec::CodecError& e) { throw FormatError(std::string("column '") + leaf.name + "': " + e.what()); }
      }
  };
  bool all_host = !chunks.empty();
  for (const ChunkRef& ch : chunks) all_host = all_host && is_host_codec(ch.c->codec);
  // ... in batches of consecutive chunks of about kInflateBatch image bytes: the page-locked image stays bounded, and batch k is on its
  // way over PCIe (the other staging buffer) while batch k + 1 is inflated
  const char* batch_env = getenv("PLX_PARQUET_INFLATE_BATCH");       // bytes; the tests shrink it to cross batch borders with small files
  const size_t kInflateBatch = batch_env && atoll(batch_env) > 0 ? (size_t)atoll(batch_env) : (size_t)256 << 20;
  std::vector<size_t> batch_last(chunks.size(), 0);       // per chunk: index of the last chunk of its batch
  for (size_t i = 0; i < chunks.size();) {
    size_t j = i, bytes = 0;
    while (j < chunks.size() && (j == i || bytes + chunks[j].blob_cap <= kInflateBatch)) bytes += chunks[j++].blob_cap;
    for (size_t k = i; k < j; k++) batch_last[k] = j - 1;
    i = j;
  }
  // Device Snappy, optionally in BATCHES of chunks (PLX_PARQUET_SNAPPY_BATCH = stored bytes per launch): the streams of the chunks whose bytes are already on their way
  // start while the host walks (and uploads) the next ones.  Measured on the 2e7-row file (steady state: walk + uploads 14 ms, then 8-12 ms of kernels): batches of 20 / 40 /
  // 64 MB read in 25.9 / 28.3 / 30.4 ms against 26.4 with ONE launch at the end -- inside the box-to-box noise -- while the summed kernel time grows (33.6 / 24.5 / 22.3 vs
  // 21.5 ms: every launch lasts as long as its longest stream).  So the default stays one launch; the knob stays for slower hosts.
  typename B::Mem err_mem = be.alloc(64);
  be.zero(be.addr(err_mem), 64);
  uint32_t* err = (uint32_t*)be.addr(err_mem);
  std::vector<typename B::Mem> snappy_mem;             // scratch + job arrays of the launches: alive until the column is decoded
  // (a kernel launched early writes into `snappy_mem` while the walk goes on: if the walk throws -- a page this path does not decode -- the stream is drained BEFORE
  //  the buffers above go back to the pool; declared after them, destroyed before them)
  struct SyncOnUnwind { B& b; int n = std::uncaught_exceptions(); ~SyncOnUnwind() { if (std::uncaught_exceptions() > n) b.discard_pending(); } } sync_on_unwind{be};
  size_t jobs_launched = 0, stored_since_launch = 0;
  static const size_t kSnappyBatch = [] { const char* e = getenv("PLX_PARQUET_SNAPPY_BATCH"); const long long v = e ? atoll(e) : 0; return v > 0 ? (size_t)v : (size_t)-1; }();
  // zstd: the same knob (PLX_PARQUET_ZSTD_BATCH, 1 GB by default: no effect on a column chunk set below that).  Measured on the 2e7-row file: one launch 31 ms; 64 MB batches 40-44, 32 MB 39-48, 16 MB 56-65 -- also with the
  // uploads on a stream of their own and the descriptor arrays queued without a wait (tried, removed).  The passes are bound by their longest CHAIN (a block's sequences,
  // a page's matches: milliseconds whatever the launch holds), so k launches one after the other on the column's stream cost k chains where one launch costs one.
  // The default of 1 GB of stored bytes only bounds the scratch of a very large read (16 bytes a sequence, up to one a value: a 1e9-row column would ask for 16 GB at once).
  static const size_t kZstdBatch = [] { const char* e = getenv("PLX_PARQUET_ZSTD_BATCH"); const long long v = e ? atoll(e) : (1ll << 30); return v > 0 ? (size_t)v : (size_t)-1; }();
  ZstdPlan zplan;                                      // the zstd pages since the last launch, indexed while their stored bytes were at hand (parquet_zstd_index.hpp)
  std::vector<size_t> zjobs;                           // ... and their job indices, in the plan's stream order
  double index_ms = 0, pread_ms = 0, stage_ms = 0, upload_ms = 0;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto launch_snappy = [&]() {
    if (jobs_launched == jobs.size()) return;
    size_t total = 0;
    std::vector<size_t> off(jobs.size() - jobs_launched);
    for (size_t j = jobs_launched; j < jobs.size(); j++) { off[j - jobs_launched] = total; total += align16((size_t)jobs[j].uncomp_size + 16); }
    typename B::Mem scratch = be.alloc(total + 64);
    const uint64_t base = be.addr(scratch);
    for (size_t j = jobs_launched; j < jobs.size(); j++) jobs[j].dst = base + off[j - jobs_launched];
    snappy_mem.push_back(scratch);
    // one workgroup per stream, started in array order: the longest streams first.  Pages and dictionaries refer to the streams' output addresses, not to job indices.
    std::vector<DecompJob> ordered;
    uint64_t bytes_out = 0;
    {
      size_t z = 0;
      for (size_t j = jobs_launched; j < jobs.size(); j++) {
        if (z < zjobs.size() && zjobs[z] == j) { z++; continue; }
        ordered.push_back(jobs[j]); bytes_out += jobs[j].uncomp_size;
      }
    }
    if (!ordered.empty()) {
      std::stable_sort(ordered.begin(), ordered.end(), [](const DecompJob& a, const DecompJob& b) { return a.uncomp_size > b.uncomp_size; });
      typename B::Mem jm = be.alloc(ordered.size() * sizeof(DecompJob) + 64);
      be.upload_small(be.addr(jm), ordered.data(), ordered.size() * sizeof(DecompJob));
      be.run_snappy((const DecompJob*)be.addr(jm), (uint32_t)ordered.size(), bytes_out, err);
      if (stats) { stats->snappy_streams += ordered.size(); stats->snappy_bytes_out += bytes_out; }
      snappy_mem.push_back(jm);
    }
    if (!zjobs.empty()) {
      // zstd: literal buffers + sequence records of all blocks, the plan's arrays, then the two passes (entropy: a wavefront per block; execute: a wavefront per page)
      uint64_t z_in = 0, z_out = 0;
      for (size_t i = 0; i < zjobs.size(); i++) { zplan.streams[i].dst = jobs[zjobs[i]].dst; z_in += jobs[zjobs[i]].comp_size; z_out += jobs[zjobs[i]].uncomp_size; }
      typename B::Mem lit = be.alloc((size_t)zplan.lit_bytes + 64), seq = be.alloc((size_t)zplan.n_seq * 16 + 64);
      zstd_plan_place(zplan, be.addr(lit), be.addr(seq));
      uint32_t n_huf_only = 0;
      const std::vector<uint32_t> order_idx = zstd_plan_order(zplan, &n_huf_only);
      const size_t nb = zplan.blocks.size() * sizeof(ZstdBlock), no = order_idx.size() * 4, nh = zplan.hufs.size() * sizeof(ZstdHufDesc), nf = zplan.fses.size() * sizeof(ZstdFseDesc),
                   ns = zplan.streams.size() * sizeof(ZstdStream);
      const size_t ob = 0, oo = align16(ob + nb), oh = align16(oo + no), of = align16(oh + nh), os = align16(of + nf), all = align16(os + ns);
      std::vector<uint8_t> image(all, 0);            // one upload for the five arrays
      if (nb) memcpy(image.data() + ob, zplan.blocks.data(), nb);
      if (no) memcpy(image.data() + oo, order_idx.data(), no);
      if (nh) memcpy(image.data() + oh, zplan.hufs.data(), nh);
      memcpy(image.data() + of, zplan.fses.data(), nf);
      memcpy(image.data() + os, zplan.streams.data(), ns);
      typename B::Mem zm = be.alloc(all + 64);
      const uint64_t za = be.addr(zm);
      be.upload_small(za, image.data(), all);
      be.run_zstd((ZstdBlock*)(za + ob), (const uint32_t*)(za + oo), (uint32_t)order_idx.size(), n_huf_only, (const ZstdHufDesc*)(za + oh), (const ZstdFseDesc*)(za + of), (const ZstdStream*)(za + os),
                  (uint32_t)zplan.streams.size(), z_in, z_out, err);
      if (stats) { stats->zstd_streams += zplan.streams.size(); stats->zstd_blocks += zplan.blocks.size(); stats->zstd_bytes_out += z_out; }
      snappy_mem.push_back(lit); snappy_mem.push_back(seq); snappy_mem.push_back(zm);
      zplan.reset(); zjobs.clear();
    }
    jobs_launched = jobs.size();
    stored_since_launch = 0;
  };
  uint8_t* batch_image = nullptr;
  size_t batch_base = 0;
  std::vector<Inflate> inflate_all;
  struct HostDict { std::vector<uint8_t> comp, plain; size_t out = 0, dict_index = 0; std::future<void> done; };
  struct HostPage { std::vector<uint8_t> comp; size_t out = 0, page_index = 0; };
  std::vector<std::unique_ptr<HostPage>> host_pages;     // zstd data pages of too many sequences for one wavefront: inflated by host threads behind the launch (below)
  std::vector<std::unique_ptr<HostDict>> host_dicts;     // long Snappy dictionary pages inflated by host threads while the walk goes on (below)
  std::vector<std::vector<uint8_t>> stored_all;          // the stored bytes of a batch's chunks stay alive until its pass has run
  struct ImageUpload { size_t blob_off, bytes; };
  std::vector<ImageUpload> image_uploads;
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const ChunkRef& ch = chunks[ci];
    const ColumnChunk& c = *ch.c;
    if (all_host && (ci == 0 || batch_last[ci - 1] != batch_last[ci])) {      // first chunk of a batch
      const ChunkRef& last = chunks[batch_last[ci]];
      batch_base = ch.blob_off;
      batch_image = be.host_stage(last.blob_off + last.blob_cap - batch_base + 64);
    }
    const size_t sz = (size_t)c.total_compressed_size;
    const bool host_codec = is_host_codec(c.codec);
    const bool codec_on = (c.codec == CODEC_SNAPPY || c.codec == CODEC_ZSTD) && !host_codec;          // pages decompressed on the device
    const bool zstd_on = codec_on && c.codec == CODEC_ZSTD;
    // a device stream: Snappy as it is; zstd with its headers indexed now, while the stored bytes are in the staging buffer
    // (page_index: the data page the stream belongs to, npos for a dictionary page; false = the page went to the host threads instead of becoming a job)
    auto push_job = [&](const uint8_t* stored, uint64_t dev, uint32_t comp, uint32_t uncomp, size_t page_index) -> bool {
      if (zstd_on) {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t n_streams = zplan.streams.size(), n_blocks = zplan.blocks.size(), n_hufs = zplan.hufs.size(), n_fses = zplan.fses.size();
        const uint64_t lit0 = zplan.lit_bytes, seq0 = zplan.n_seq, comp0 = zplan.n_compressed;
        try { zstd_index_stream(zplan, stored, comp, dev, uncomp); }
        catch (const codec::CodecError& e) { throw FormatError(std::string("column '") + leaf.name + "': " + e.what()); }
        index_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (page_index != npos && zplan.n_seq - seq0 > zstd_host_sequences()) {
          zplan.streams.resize(n_streams); zplan.blocks.resize(n_blocks); zplan.hufs.resize(n_hufs); zplan.fses.resize(n_fses);
          zplan.lit_bytes = lit0; zplan.n_seq = seq0; zplan.n_compressed = comp0;
          auto hp = std::make_unique<HostPage>();
          hp->comp.assign(stored, stored + comp); hp->out = uncomp; hp->page_index = page_index;
          host_pages.push_back(std::move(hp));
          if (stats) { stats->host_inflated_pages++; stats->host_inflated_bytes += uncomp; }
          return false;
        }
        zjobs.push_back(jobs.size());
      }
      jobs.push_back(DecompJob{dev, 0, comp, uncomp});
      return true;
    };
    // Device codec / none: the stored bytes are staged and uploaded as they are.  Host codec: the stored bytes stay in pageable memory;
    // what is staged and uploaded is the chunk's IMAGE -- the page payloads decompressed, back to back -- and the pages then look
    // like pages of an uncompressed file to every kernel.
    std::vector<uint8_t> stored_here;
    uint8_t* host = nullptr;
    uint8_t* image = nullptr;
    size_t ipos = 0;
    if (host_codec) {
      std::vector<uint8_t>& stored = all_host ? (stored_all.emplace_back(), stored_all.back()) : stored_here;
      stored.resize(sz + 16); host = stored.data();
      image = all_host ? batch_image + (ch.blob_off - batch_base) : be.host_stage(ch.blob_cap + 16);
    } else { const double t0 = now_ms(); host = be.host_stage(sz + 16); stage_ms += now_ms() - t0; }
    { const double t0 = now_ms(); f.pread_sliced(host, sz, c.start()); pread_ms += now_ms() - t0; }
    if (stats) stats->file_bytes += sz;
    std::vector<Inflate> inflate;
    size_t pos = 0;
    int64_t seen = 0;
    size_t chunk_dict = npos;
    while (seen < c.num_values) {
      if (pos >= sz) throw FormatError("column chunk ends before all its values were found");
      PageHeader h = parse_page_header(host + pos, sz - pos);
      pos += h.header_bytes;
      if ((size_t)h.compressed_size > sz - pos) throw FormatError("page runs past its column chunk");
      const bool data_page = h.type == PAGE_DATA || h.type == PAGE_DATA_V2;
      size_t place = pos;                                     // offset of the payload inside what is uploaded
      if (host_codec && (data_page || (h.type == PAGE_DICTIONARY && !is_bytes))) {
        const size_t out = (size_t)h.uncompressed_size;
        if (out > ch.blob_cap - ipos) throw FormatError("pages decompress to more than the chunk metadata says");
        place = ipos;
        if (h.type == PAGE_DATA_V2) {
          // v2: level bytes are stored uncompressed in front of the (optionally compressed) values
          if (h.def_len < 0 || h.rep_len < 0 || (int64_t)h.def_len + h.rep_len > h.compressed_size || (int64_t)h.def_len + h.rep_len > h.uncompressed_size)
            throw FormatError("v2 level bytes exceed the page");
          const size_t lv = (size_t)h.def_len + (size_t)h.rep_len;
          inflate.push_back({host + pos, lv, image + ipos, lv, true, c.codec});
          const bool no_values = (size_t)h.compressed_size == lv;     // all-null page without value bytes: nothing to inflate, whatever is_compressed says
          if (no_values && out != lv) throw FormatError("v2 page without value bytes whose sizes differ");
          inflate.push_back({host + pos + lv, (size_t)h.compressed_size - lv, image + ipos + lv, out - lv, !h.is_compressed || no_values, c.codec});
        } else {
          inflate.push_back({host + pos, (size_t)h.compressed_size, image + ipos, out, false, c.codec});
        }
        ipos += out;
      }
      const uint64_t payload = blob_addr + ch.blob_off + place;
      // sizes of the payload as the kernels will see it
      const uint32_t seen_comp = host_codec ? (uint32_t)h.uncompressed_size : (uint32_t)h.compressed_size;
      if (h.type == PAGE_DICTIONARY) {
        if (chunk_dict != npos) throw FormatError("two dictionary pages in one column chunk");
        if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICTIONARY) throw Unsupported(std::string("dictionary page encoding ") + encoding_name(h.encoding));
        DictDesc d{};
        d.n = (uint32_t)h.num_values;
        chunk_dict = dicts.size();
        if (is_bytes) {
          // strings: the host reads the (small) dictionary, assigns column-wide codes in first-appearance order
          std::vector<uint8_t> plain;
          const uint8_t* p = host + pos;
          size_t n = (size_t)h.compressed_size;
          if (codec_on && !zstd_on) { plain = snappy_decompress_host(p, n, (size_t)h.uncompressed_size); p = plain.data(); n = plain.size(); }
          else if (host_codec || zstd_on) {
            plain.resize((size_t)h.uncompressed_size);
            try {
              host_inflate(c.codec, p, n, plain.data(), plain.size());
            } catch (const codec::CodecError& e) { throw FormatError(std::string("dictionary page: ") + e.what()); }
            p = plain.data(); n = plain.size();
          }
          remap_base_of_dict.push_back(remap.size());
          size_t q = 0;
          for (uint32_t i = 0; i < d.n; i++) {
            if (q + 4 > n) throw FormatError("string dictionary page ends early");
            uint32_t len = load_u32(p + q);
            q += 4;
            if (len > n - q) throw FormatError("string dictionary entry runs past the page");
            std::string s((const char*)p + q, len);
            q += len;
            auto it = cat_index.find(s);
            uint32_t code;
            if (it == cat_index.end()) { code = (uint32_t)categories.size(); cat_index.emplace(s, code); categories.push_back(std::move(s)); }
            else code = it->second;
            remap.push_back(code);
          }
          job_of_dict.push_back(npos);
        } else {
          if ((uint64_t)d.n * lt.src_width > (uint64_t)h.uncompressed_size && lt.src_width) throw FormatError("dictionary page smaller than its entry count");
          if (lt.src_width == 0) throw Unsupported("dictionary-encoded booleans");
          d.values = payload;
          if (codec_on && (size_t)h.uncompressed_size >= host_dict_min_bytes()) {
            // A LONG Snappy stream is the device kernel's critical path -- one workgroup, 4 KB a round: pyarrow's dictionary pages of up to 1 MB ran for 13 ms next to
            // a thousand data pages of 2 ms each, and the launch lasts as long as its longest stream.  A host thread inflates such a page in about a millisecond while the
            // walk goes on (the compressed bytes are copied out of the staging buffer first: the next chunk reuses it); the plain values travel in one upload behind it.
            auto hd = std::make_unique<HostDict>();
            hd->comp.assign(host + pos, host + pos + (size_t)h.compressed_size);
            hd->plain.resize((size_t)h.uncompressed_size + 16);
            hd->out = (size_t)h.uncompressed_size;
            hd->dict_index = dicts.size();
            HostDict* hp = hd.get();
            const int dict_codec = c.codec;      // (zstd: the page's execute pass is one wavefront walking its sequences in order -- 1.3e5 of them in a 1 MB dictionary of sorted keys)
            hd->done = std::async(std::launch::async, [hp, dict_codec] {
              if (dict_codec == CODEC_SNAPPY) snappy_decompress_into(hp->comp.data(), hp->comp.size(), hp->plain.data(), hp->out);
              else host_inflate(dict_codec, hp->comp.data(), hp->comp.size(), hp->plain.data(), hp->out);
            });
            host_dicts.push_back(std::move(hd));
            job_of_dict.push_back(npos);
            if (stats) { stats->host_inflated_pages++; stats->host_inflated_bytes += (uint64_t)h.uncompressed_size; }
          } else if (codec_on) {
            job_of_dict.push_back(jobs.size());
            push_job(host + pos, payload, (uint32_t)h.compressed_size, (uint32_t)h.uncompressed_size, npos);
          } else job_of_dict.push_back(npos);
          remap_base_of_dict.push_back(npos);
        }
        dicts.push_back(d);
        if (stats) stats->dict_pages++;
      } else if (data_page) {
        PageDesc p{};
        const bool v2 = h.type == PAGE_DATA_V2;
        p.src = payload; p.comp_size = seen_comp; p.uncomp_size = (uint32_t)h.uncompressed_size;
        p.num_values = (uint32_t)h.num_values; p.row0 = row0 + (uint64_t)seen;
        p.flags = (v2 ? PF_V2 : 0u) | (optional ? PF_HAS_DEF : 0u);
        if (h.encoding == ENC_PLAIN_DICTIONARY || h.encoding == ENC_RLE_DICTIONARY) {
          if (chunk_dict == npos) throw FormatError("dictionary-encoded page without a dictionary page");
          p.flags |= PF_DICT; p.dict = (uint32_t)chunk_dict;
        } else if (h.encoding == ENC_PLAIN) {
          if (is_bytes) throw NeedsHostStrings{};
        } else if (h.encoding == ENC_RLE && leaf.type == PT_BOOLEAN) {
          p.flags |= PF_RLE_VALUES;
        } else {
          if (is_bytes && (h.encoding == ENC_DELTA_LENGTH_BYTE_ARRAY || h.encoding == ENC_DELTA_BYTE_ARRAY)) throw NeedsHostStrings{};
          if (!is_bytes && (h.encoding == ENC_DELTA_BINARY_PACKED || h.encoding == ENC_BYTE_STREAM_SPLIT)) throw NeedsHostValues{};
          throw Unsupported("column '" + leaf.name + "': page encoding " + encoding_name(h.encoding));
        }
        size_t job = npos;
        if (v2) {
          if (h.rep_len != 0) throw Unsupported("repetition levels in a flat column");
          if (h.def_len < 0 || h.def_len > h.compressed_size) throw FormatError("v2 level bytes exceed the page");
          if (!optional && h.def_len != 0) throw FormatError("definition levels in a required column");
          p.v2_def_len = (uint32_t)h.def_len;
          // an all-null v2 page may carry NO value bytes at all although is_compressed is set (parquet-mr writes them; the reference's
          // fixture empty_datapage_v2.snappy.parquet, py-polars/tests/unit/io/test_parquet.py:848): nothing to inflate
          if (h.is_compressed && h.compressed_size == h.def_len && h.uncompressed_size != h.def_len) throw FormatError("v2 page without value bytes whose sizes differ");
          if (codec_on && h.is_compressed && h.compressed_size > h.def_len) {
            p.flags |= PF_COMPRESSED;
            if (h.uncompressed_size < h.def_len) throw FormatError("v2 page smaller than its level bytes");
            job = jobs.size();
            if (!push_job(host + pos + (size_t)h.def_len, payload + (uint64_t)h.def_len, (uint32_t)(h.compressed_size - h.def_len), (uint32_t)(h.uncompressed_size - h.def_len), pages.size())) job = npos;
          }
        } else {
          if (optional && h.def_encoding != ENC_RLE) throw Unsupported(std::string("definition levels encoded as ") + encoding_name(h.def_encoding));
          if (codec_on) {
            p.flags |= PF_COMPRESSED;
            job = jobs.size();
            if (!push_job(host + pos, payload, (uint32_t)h.compressed_size, (uint32_t)h.uncompressed_size, pages.size())) job = npos;
          }
        }
        if (c.codec == CODEC_UNCOMPRESSED && h.compressed_size != h.uncompressed_size) throw FormatError("uncompressed page whose two sizes differ");
        job_of_page.push_back(job);
        pages.push_back(p);
        seen += h.num_values;
        if (stats) stats->data_pages++;
      }  // index pages and unknown page types are skipped
      pos += (size_t)h.compressed_size;
    }
    if (seen != c.num_values) throw FormatError("pages of a column chunk hold more values than its metadata says");
    if (host_codec) {
      if (stats) { stats->host_inflated_pages += inflate.size(); stats->host_inflated_bytes += ipos; }
      if (all_host) {
        inflate_all.insert(inflate_all.end(), inflate.begin(), inflate.end());
        image_uploads.push_back({ch.blob_off, ipos});
        if (ci == batch_last[ci]) {
          run_inflate(inflate_all);
          for (const ImageUpload& u : image_uploads) be.upload(blob_addr + u.blob_off, batch_image + (u.blob_off - batch_base), u.bytes);
          inflate_all.clear(); image_uploads.clear(); stored_all.clear();
        }
      } else {
        run_inflate(inflate);
        be.upload(blob_addr + ch.blob_off, image, ipos);
      }
    } else {
      { const double t0 = now_ms(); be.upload(blob_addr + ch.blob_off, host, sz); upload_ms += now_ms() - t0; }
      stored_since_launch += sz;
      if (codec_on && stored_since_launch >= (zstd_on ? kZstdBatch : kSnappyBatch) && ci + 1 < chunks.size()) launch_snappy();
    }
    row0 += (uint64_t)ch.rows;
  }
  trace_point(leaf.name, "chunks read, uploads queued");
  if (trace_on()) fprintf(stderr, "[plx parquet]    %s: of the walk -- file reads %.2f ms, waits for a staging buffer %.2f, upload calls %.2f, zstd index pass %.2f\n", leaf.name.c_str(), pread_ms, stage_ms, upload_ms, index_ms);
  // -- the streams of the last chunks; pages and dictionaries learn where their bytes will be ------------------------------------------------------
  launch_snappy();
  for (size_t i = 0; i < pages.size(); i++) if (job_of_page[i] != npos) pages[i].dst = jobs[job_of_page[i]].dst;
  for (size_t i = 0; i < dicts.size(); i++) if (job_of_dict[i] != npos) dicts[i].values = jobs[job_of_dict[i]].dst;
  typename B::Mem host_page_mem{};
  if (!host_pages.empty()) {
    // (the column's device passes are running: these pages are inflated next to them, by the bounded pool of the host codecs)
    size_t total = 0;
    std::vector<size_t> off(host_pages.size());
    for (size_t i = 0; i < host_pages.size(); i++) { off[i] = total; total += align16(host_pages[i]->out + 16); }
    host_page_mem = be.alloc(total + 64);
    uint8_t* st = be.host_stage(total + 64);
    std::vector<Inflate> tasks;
    for (size_t i = 0; i < host_pages.size(); i++) tasks.push_back({host_pages[i]->comp.data(), host_pages[i]->comp.size(), st + off[i], host_pages[i]->out, false, CODEC_ZSTD});
    run_inflate(tasks);
    for (size_t i = 0; i < host_pages.size(); i++) pages[host_pages[i]->page_index].dst = be.addr(host_page_mem) + off[i];
    be.upload(be.addr(host_page_mem), st, total);
    host_pages.clear();
    trace_point(leaf.name, "zstd pages of many sequences inflated on the host, uploaded");
  }
  typename B::Mem host_dict_mem{};
  if (!host_dicts.empty()) {
    size_t total = 0;
    std::vector<size_t> off(host_dicts.size());
    for (size_t i = 0; i < host_dicts.size(); i++) { off[i] = total; total += align16(host_dicts[i]->out + 16); }
    host_dict_mem = be.alloc(total + 64);
    uint8_t* st = be.host_stage(total + 64);
    for (size_t i = 0; i < host_dicts.size(); i++) {
      try { host_dicts[i]->done.get(); }
      catch (const FormatError& e) { for (size_t j = i + 1; j < host_dicts.size(); j++) host_dicts[j]->done.wait(); throw FormatError(std::string("column '") + leaf.name + "': dictionary page: " + e.what()); }
      catch (const codec::CodecError& e) { for (size_t j = i + 1; j < host_dicts.size(); j++) host_dicts[j]->done.wait(); throw FormatError(std::string("column '") + leaf.name + "': dictionary page: " + e.what()); }
      memcpy(st + off[i], host_dicts[i]->plain.data(), host_dicts[i]->out);
      dicts[host_dicts[i]->dict_index].values = be.addr(host_dict_mem) + off[i];
    }
    be.upload(be.addr(host_dict_mem), st, total);
    host_dicts.clear();
    trace_point(leaf.name, "long dictionary pages inflated on the host, uploaded");
  }
  typename B::Mem remap_mem{};
  if (is_bytes) {
    remap_mem = be.alloc(remap.size() * 4 + 64);
    if (!remap.empty()) be.upload_small(be.addr(remap_mem), remap.data(), remap.size() * 4);
    for (size_t i = 0; i < dicts.size(); i++) dicts[i].values = be.addr(remap_mem) + remap_base_of_dict[i] * 4;
    { std::lock_guard<std::mutex> lk(f.meta_mu); f.categories[leaf_idx] = std::move(categories); }
  }

  const uint32_t n_pages = (uint32_t)pages.size();
  typename B::Mem pages_mem = be.alloc(pages.size() * sizeof(PageDesc) + 64);
  typename B::Mem dicts_mem = be.alloc(dicts.size() * sizeof(DictDesc) + 64);
  be.upload_small(be.addr(pages_mem), pages.data(), pages.size() * sizeof(PageDesc));
  if (!dicts.empty()) be.upload_small(be.addr(dicts_mem), dicts.data(), dicts.size() * sizeof(DictDesc));

  // -- device passes (the Snappy streams are already running: launch_snappy) ---------------------------------------------------------------------
  PageDesc* d_pages = (PageDesc*)be.addr(pages_mem);
  be.run_page_prepare(d_pages, n_pages, err);
  // run tables: entries per (page, stream) -> exclusive prefix -> fill
  typename B::Mem counts = be.alloc((size_t)n_pages * 2 * 4 + 64);
  typename B::Mem offs = be.alloc(((size_t)n_pages * 2 + 1) * 8 + 64);
  be.run_count_runs(d_pages, n_pages, nulls_possible, (uint32_t*)be.addr(counts), err);   // level tables only when nulls are possible
  be.scan_u32((const uint32_t*)be.addr(counts), (uint64_t*)be.addr(offs), (int64_t)n_pages * 2);
  const uint64_t n_entries = be.read_u64(be.addr(offs) + (uint64_t)n_pages * 2 * 8);
  trace_point(leaf.name, "streams inflated, runs counted (first host read)");
  if (stats) stats->run_entries += n_entries;
  typename B::Mem runs = be.alloc((size_t)n_entries * sizeof(RunEntry) + 64);
  be.run_fill_runs(d_pages, n_pages, (const uint64_t*)be.addr(offs), (RunEntry*)be.addr(runs));

  const uint64_t n_words = ((uint64_t)n_rows + 63) / 64;
  typename B::Mem word_prefix{};
  const uint64_t* d_validity = nullptr;
  const uint64_t* d_prefix = nullptr;
  if (nulls_possible) {
    res.validity = be.alloc((size_t)n_words * 8 + 8);
    res.has_validity = true;
    typename B::Mem popc = be.alloc((size_t)n_words * 4 + 64);
    word_prefix = be.alloc(((size_t)n_words + 1) * 8 + 64);
    be.zero(be.addr(res.validity) + n_words * 8, 8);
    be.run_validity(d_pages, n_pages, (const RunEntry*)be.addr(runs), (const uint64_t*)be.addr(offs), (uint64_t)n_rows, (uint64_t*)be.addr(res.validity),
                    (uint32_t*)be.addr(popc), err);
    be.scan_u32((const uint32_t*)be.addr(popc), (uint64_t*)be.addr(word_prefix), (int64_t)n_words);
    d_validity = (const uint64_t*)be.addr(res.validity);
    d_prefix = (const uint64_t*)be.addr(word_prefix);
    be.run_page_valid0(d_pages, n_pages, d_validity, d_prefix);
  }

  res.values = be.alloc(out_bytes(n_rows));
  ColumnDecode cd{};
  cd.pages = d_pages; cd.n_pages = n_pages; cd.dicts = (const DictDesc*)be.addr(dicts_mem); cd.runs = (const RunEntry*)be.addr(runs);
  cd.run_off = (const uint64_t*)be.addr(offs); cd.validity = d_validity; cd.word_prefix = d_prefix; cd.n_rows = (uint64_t)n_rows;
  cd.src_width = lt.src_width; cd.dict_width = is_bytes ? 4u : lt.src_width;
  if (lt.dtype == PLX_BOOL) be.zero(be.addr(res.values) + n_words * 8, 8);
  be.run_decode(cd, (void*)be.addr(res.values), out_width, err);

  // -- one synchronisation: error word (+ the valid-row total) ---------------------------------------------------------------------------
  uint32_t e = be.read_u32(be.addr(err_mem));
  trace_point(leaf.name, "decoded");
  if (e) throw FormatError("column '" + leaf.name + "' of " + f.path + ": " + error_bits_text(e));
  if (nulls_possible) {
    uint64_t valid = be.read_u64(be.addr(word_prefix) + n_words * 8);
    res.null_count = n_rows - (int64_t)valid;
    if (res.null_count == 0) { res.validity = typename B::Mem{}; res.has_validity = false; }
  }
  return res;
}

// ---- string columns with PLAIN pages: views assembled on the host, dictionary built on the device ------------------------------------------
// Variable-length values are a chain (each length prefix says where the next one starts), and a column that is not dictionary-encoded
// has as many distinct values as it likes -- so there is no per-chunk dictionary to remap.  Host threads (one page at a time each)
// decompress the page, decode its definition levels and -- for the dictionary-encoded pages such a chunk may start with -- its
// indices, and write one 16-byte view per row ({len, inline bytes} or {len, prefix, buffer, offset}: the reference's own layout,
// crates/polars-arrow/src/array/binview/view.rs:20-29) pointing into the page payloads, which become the column's data buffers.  The
// views then go through the device-side dictionary encoder (backend: plx_strview_dict_encode, kernels_strview.hip) exactly like a
// Utf8View column handed over by the caller.  Reference: arrow/read/deserialize/binview/{required,optional}.rs, parquet/encoding/plain_byte_array.rs.
inline std::vector<uint32_t> decode_hybrid_host(const uint8_t* s, size_t len, uint32_t bits, size_t count) {
  std::vector<uint32_t> out;
  out.reserve(count);
  if (bits == 0) { out.assign(count, 0); return out; }
  if (bits > 32) throw FormatError("hybrid stream with more than 32 bits per value");
  const size_t vbytes = (bits + 7) / 8;
  const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1);
  size_t pos = 0;
  while (out.size() < count) {
    if (pos >= len) throw FormatError("hybrid stream ends before the page's values");
    uint32_t h = 0;
    for (uint32_t shift = 0;; shift += 7) {
      if (pos >= len || shift > 28) throw FormatError("malformed run header");
      uint32_t b = s[pos++];
      h |= (b & 0x7f) << shift;
      if (!(b & 0x80)) break;
    }
    if (h & 1) {
      size_t bytes = (size_t)(h >> 1) * bits;
      if (bytes > len - pos) bytes = len - pos;
      size_t n = bytes * 8 / bits;
      if (n > count - out.size()) n = count - out.size();
      if (n == 0) throw FormatError("empty bit-packed run");
      for (size_t i = 0; i < n; i++) {
        const size_t bit = i * bits;
        uint64_t w = 0;
        const size_t at = pos + (bit >> 3), avail = len - at < 8 ? len - at : 8;
        memcpy(&w, s + at, avail);
        out.push_back((uint32_t)(w >> (bit & 7)) & mask);
      }
      pos += bytes;
    } else {
      size_t n = h >> 1;
      if (vbytes > len - pos) throw FormatError("run value past the stream");
      uint32_t v = 0;
      for (size_t b = 0; b < vbytes; b++) v |= (uint32_t)s[pos + b] << (8 * b);
      pos += vbytes;
      if (n > count - out.size()) n = count - out.size();
      if (n == 0 && pos >= len) throw FormatError("hybrid stream ends before the page's values");
      out.insert(out.end(), n, v & mask);
    }
  }
  return out;
}

// DELTA_BINARY_PACKED (parquet/encoding/delta_bitpacked/decoder.rs): header {block size, miniblocks per block, total count, first value},
// then per block {min delta, one bit width per miniblock, bit-packed deltas}; value[i] = value[i - 1] + min delta + delta (wrapping).
// Appends the stream's values to out; returns the bytes consumed.
inline size_t delta_binary_packed_host(const uint8_t* p, size_t n, std::vector<int64_t>& out) {
  size_t pos = 0;
  auto uleb = [&]() {
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      if (pos >= n) throw FormatError("delta stream ends inside its header");
      const uint8_t b = p[pos++];
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    throw FormatError("delta stream: varint too long");
  };
  auto zigzag = [&]() { const uint64_t v = uleb(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); };
  const uint64_t block = uleb(), minis = uleb(), total = uleb();
  if (block == 0 || block % 128 || minis == 0 || block % minis || (block / minis) % 32 || block > (1u << 20)) throw FormatError("delta stream with an invalid block shape");
  if (total > ((uint64_t)1 << 31)) throw FormatError("delta stream with an absurd value count");
  int64_t value = zigzag();
  const uint64_t per_mini = block / minis;
  uint64_t left = total;
  if (left) { out.push_back(value); left--; }
  std::vector<uint8_t> widths((size_t)minis);
  while (left) {
    const int64_t min_delta = zigzag();
    if (minis > n - pos) throw FormatError("delta stream ends inside a block header");
    memcpy(widths.data(), p + pos, (size_t)minis);
    pos += (size_t)minis;
    for (uint64_t m = 0; m < minis && left; m++) {
      const uint32_t bw = widths[(size_t)m];
      if (bw > 64) throw FormatError("delta stream with a bit width above 64");
      const size_t bytes = (size_t)(per_mini * bw / 8);
      if (bytes > n - pos) throw FormatError("delta stream ends inside a miniblock");
      const uint64_t take = left < per_mini ? left : per_mini;
      for (uint64_t i = 0; i < take; i++) {
        uint64_t d = 0;
        if (bw) {
          const uint64_t bit = i * bw;
          const size_t at = pos + (size_t)(bit >> 3);
          unsigned __int128 w = 0;
          const size_t avail = n - at < 16 ? n - at : 16;
          memcpy(&w, p + at, avail);
          w >>= (bit & 7);
          d = bw == 64 ? (uint64_t)w : (uint64_t)w & (((uint64_t)1 << bw) - 1);
        }
        value = (int64_t)((uint64_t)value + (uint64_t)min_delta + d);
        out.push_back(value);
      }
      pos += bytes;
      left -= take;
    }
  }
  return pos;
}

inline void make_view(uint8_t* dst, const uint8_t* bytes, uint32_t len, uint32_t buffer, uint32_t offset) {
  memset(dst, 0, 16);
  memcpy(dst, &len, 4);
  if (len <= 12) { if (len) memcpy(dst + 4, bytes, len); }
  else { memcpy(dst + 4, bytes, 4); memcpy(dst + 8, &buffer, 4); memcpy(dst + 12, &offset, 4); }
}

inline void page_inflate(int codec_id, const uint8_t* src, size_t n, uint8_t* dst, size_t out) {
  if (n == 0 && out == 0) return;        // the value section of an all-null v2 page
  try {
    if (codec_id == CODEC_UNCOMPRESSED) { if (n != out) throw FormatError("uncompressed page whose two sizes differ"); if (n) memcpy(dst, src, n); }
    else if (codec_id == CODEC_SNAPPY) { std::vector<uint8_t> v = snappy_decompress_host(src, n, out); if (out) memcpy(dst, v.data(), out); }
    else host_inflate(codec_id, src, n, dst, out);
  } catch (const codec::CodecError& e) { throw FormatError(e.what()); }
}

template <class B> ColumnResult<B> read_string_column_host(B& be, File& f, const std::vector<int>& row_groups, int leaf_idx, ReadStats* stats) {
  const FileMetaData& md = f.md;
  const Leaf& leaf = md.leaves[leaf_idx];
  const LeafType lt = leaf_type(leaf);
  const bool optional = leaf.repetition == REP_OPTIONAL;
  trace_point(leaf.name, "start (host string path)");
  ColumnResult<B> res;
  res.dtype = lt.dtype; res.logical = lt.logical;
  struct ChunkRef { const ColumnChunk* c; int64_t rows; };
  std::vector<ChunkRef> chunks;
  int64_t n_rows = 0;
  for (int g : row_groups) {
    const RowGroup& rg = md.row_groups[g];          // indices, metadata and codec were validated by the device reader before it gave up
    const ColumnChunk& c = rg.columns[leaf_idx];
    if (rg.num_rows == 0) continue;
    chunks.push_back({&c, rg.num_rows});
    n_rows += rg.num_rows;
  }
  res.len = n_rows;
  uint8_t* const views = be.host_stage((size_t)n_rows * 16 + 16);      // page-locked on the GPU: the views cross PCIe in one DMA (nothing else asks for a staging buffer before they are handed over)
  std::vector<uint8_t> validity((size_t)(n_rows + 7) / 8 + 8, 0);
  std::vector<std::vector<uint8_t>> data;           // one per page payload / dictionary page: the column's data buffers
  int64_t nulls = 0;
  uint64_t row0 = 0;
  // All pages of all chunks are decoded in ONE parallel pass (round 5; chunk after chunk before: a chunk has a handful of pages, so a handful of threads were busy, and the
  // validity bits of its rows were merged by the calling thread -- 80 ms of a 115 ms read of 2e7 one-character strings).
  struct Task { PageHeader h; const uint8_t* stored; uint64_t row0; size_t buffer; size_t chunk; };
  struct ChunkState { std::vector<uint8_t> stored; std::vector<uint8_t> dict_views; bool have_dict = false; int codec = 0; };       // dict_views: 16 bytes per dictionary entry
  std::vector<ChunkState> state(chunks.size());
  std::vector<Task> tasks;
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const ChunkRef& ch = chunks[ci];
    const ColumnChunk& c = *ch.c;
    const size_t sz = (size_t)c.total_compressed_size;
    std::vector<uint8_t>& stored = state[ci].stored;
    stored.resize(sz + 16);
    f.pread_sliced(stored.data(), sz, c.start());
    if (stats) stats->file_bytes += sz;
    std::vector<uint8_t>& dict_views = state[ci].dict_views;
    bool& have_dict = state[ci].have_dict;
    state[ci].codec = c.codec;
    size_t pos = 0;
    int64_t seen = 0;
    while (seen < c.num_values) {
      if (pos >= sz) throw FormatError("column chunk ends before all its values were found");
      PageHeader h = parse_page_header(stored.data() + pos, sz - pos);
      pos += h.header_bytes;
      if ((size_t)h.compressed_size > sz - pos) throw FormatError("page runs past its column chunk");
      if (h.type == PAGE_DICTIONARY) {
        if (have_dict) throw FormatError("two dictionary pages in one column chunk");
        if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICTIONARY) throw Unsupported(std::string("dictionary page encoding ") + encoding_name(h.encoding));
        const size_t buf = data.size();
        data.emplace_back((size_t)h.uncompressed_size + 16);
        page_inflate(c.codec, stored.data() + pos, (size_t)h.compressed_size, data[buf].data(), (size_t)h.uncompressed_size);
        const uint8_t* p = data[buf].data();
        const size_t n = (size_t)h.uncompressed_size;
        dict_views.resize((size_t)h.num_values * 16);
        size_t q = 0;
        for (int32_t i = 0; i < h.num_values; i++) {
          if (q + 4 > n) throw FormatError("string dictionary page ends early");
          const uint32_t len = load_u32(p + q);
          q += 4;
          if (len > n - q) throw FormatError("string dictionary entry runs past the page");
          make_view(dict_views.data() + 16 * (size_t)i, p + q, len, (uint32_t)buf, (uint32_t)q);
          q += len;
        }
        have_dict = true;
        if (stats) stats->dict_pages++;
      } else if (h.type == PAGE_DATA || h.type == PAGE_DATA_V2) {
        tasks.push_back({h, stored.data() + pos, row0 + (uint64_t)seen, data.size(), ci});
        data.emplace_back();
        seen += h.num_values;
        if (stats) stats->data_pages++;
      }
      pos += (size_t)h.compressed_size;
    }
    if (seen != c.num_values) throw FormatError("pages of a column chunk hold more values than its metadata says");
    row0 += (uint64_t)ch.rows;
  }
  {
    // pages in parallel: each writes the views and the validity bits of its own rows (the bytes two pages share are OR-ed atomically)
    const size_t threads = host_threads(tasks.size());
    std::vector<std::exception_ptr> errs(std::max<size_t>(threads, 1));
    std::vector<int64_t> nulls_of(std::max<size_t>(threads, 1), 0);
    std::atomic<size_t> next_task{0};
    auto work = [&](size_t t) {
      try {
        std::vector<uint8_t> valid;
        for (size_t k = next_task.fetch_add(1); k < tasks.size(); k = next_task.fetch_add(1)) {
          const Task& tk = tasks[k];
          const PageHeader& h = tk.h;
          const ChunkState& cs = state[tk.chunk];
          const std::vector<uint8_t>& dict_views = cs.dict_views;
          const bool have_dict = cs.have_dict;
          struct { int codec; } c{cs.codec};
          const bool v2 = h.type == PAGE_DATA_V2;
          const size_t nvals = (size_t)h.num_values;
          std::vector<uint8_t>& payload = data[tk.buffer];
          payload.resize((size_t)h.uncompressed_size + 16);
          const uint8_t* levels = nullptr; size_t levels_len = 0, values_off = 0;
          if (v2) {
            if (h.rep_len != 0) throw Unsupported("repetition levels in a flat column");
            if (h.def_len < 0 || h.def_len > h.compressed_size || h.def_len > h.uncompressed_size) throw FormatError("v2 level bytes exceed the page");
            const size_t lv = (size_t)h.def_len;
            if (lv) memcpy(payload.data(), tk.stored, lv);
            page_inflate(h.is_compressed ? c.codec : (int)CODEC_UNCOMPRESSED, tk.stored + lv, (size_t)h.compressed_size - lv, payload.data() + lv, (size_t)h.uncompressed_size - lv);
            levels = payload.data(); levels_len = lv; values_off = lv;
          } else {
            page_inflate(c.codec, tk.stored, (size_t)h.compressed_size, payload.data(), (size_t)h.uncompressed_size);
            if (optional) {
              if (h.def_encoding != ENC_RLE) throw Unsupported(std::string("definition levels encoded as ") + encoding_name(h.def_encoding));
              if (h.uncompressed_size < 4) throw FormatError("page too small for its level length");
              const uint32_t ll = load_u32(payload.data());
              if (ll > (uint32_t)h.uncompressed_size - 4) throw FormatError("level bytes exceed the page");
              levels = payload.data() + 4; levels_len = ll; values_off = 4 + (size_t)ll;
            }
          }
          valid.assign(nvals, 1);
          size_t present = nvals;
          if (optional) {
            std::vector<uint32_t> lv = decode_hybrid_host(levels, levels_len, 1, nvals);
            present = 0;
            for (size_t i = 0; i < nvals; i++) { if (lv[i] > 1) throw FormatError("definition level > 1 in a flat column"); valid[i] = (uint8_t)lv[i]; present += lv[i]; }
          }
          const uint8_t* vals = payload.data() + values_off;
          const size_t vlen = (size_t)h.uncompressed_size - values_off;
          uint8_t* out = views + 16 * (size_t)tk.row0;
          if (h.encoding == ENC_PLAIN) {
            size_t q = 0;
            for (size_t i = 0; i < nvals; i++) {
              if (!valid[i]) { memset(out + 16 * i, 0, 16); continue; }
              if (q + 4 > vlen) throw FormatError("string page ends before its values");
              const uint32_t len = load_u32(vals + q);
              q += 4;
              if (len > vlen - q) throw FormatError("string value runs past the page");
              make_view(out + 16 * i, vals + q, len, (uint32_t)tk.buffer, (uint32_t)(values_off + q));
              q += len;
            }
          } else if (h.encoding == ENC_PLAIN_DICTIONARY || h.encoding == ENC_RLE_DICTIONARY) {
            if (!have_dict) throw FormatError("dictionary-encoded page without a dictionary page");
            if (vlen < 1 && present) throw FormatError("dictionary-encoded page without its bit width");
            const uint32_t bw = vlen ? vals[0] : 0;
            std::vector<uint32_t> idx = decode_hybrid_host(vals + (vlen ? 1 : 0), vlen ? vlen - 1 : 0, bw, present);
            const size_t nd = dict_views.size() / 16;
            size_t d = 0;
            for (size_t i = 0; i < nvals; i++) {
              if (!valid[i]) { memset(out + 16 * i, 0, 16); continue; }
              const uint32_t ix = idx[d++];
              if (ix >= nd) throw FormatError("dictionary index out of range");
              memcpy(out + 16 * i, dict_views.data() + 16 * (size_t)ix, 16);
            }
          } else if (h.encoding == ENC_DELTA_LENGTH_BYTE_ARRAY) {
            // all lengths first (delta-packed), then the bytes back to back (parquet/encoding/delta_length_byte_array)
            std::vector<int64_t> lens;
            const size_t used = delta_binary_packed_host(vals, vlen, lens);
            if (lens.size() != present) throw FormatError("delta-length page with a different value count than its levels");
            size_t q = used, d = 0;
            for (size_t i = 0; i < nvals; i++) {
              if (!valid[i]) { memset(out + 16 * i, 0, 16); continue; }
              const int64_t len = lens[d++];
              if (len < 0 || (uint64_t)len > vlen - q || len > 0x7fffffff) throw FormatError("string value runs past the page");
              make_view(out + 16 * i, vals + q, (uint32_t)len, (uint32_t)tk.buffer, (uint32_t)(values_off + q));
              q += (size_t)len;
            }
          } else if (h.encoding == ENC_DELTA_BYTE_ARRAY) {
            // incremental encoding: value = first `prefix` bytes of the previous value + suffix; the values are rebuilt into a buffer of
            // their own, appended behind the page payload (parquet/encoding/delta_byte_array)
            std::vector<int64_t> prefixes, suffix_lens;
            size_t used = delta_binary_packed_host(vals, vlen, prefixes);
            used += delta_binary_packed_host(vals + used, vlen - used, suffix_lens);
            if (prefixes.size() != present || suffix_lens.size() != present) throw FormatError("delta-byte-array page with a different value count than its levels");
            size_t total = 0;
            { int64_t prev = 0; for (size_t k2 = 0; k2 < present; k2++) { if (prefixes[k2] < 0 || prefixes[k2] > prev || suffix_lens[k2] < 0) throw FormatError("delta-byte-array page with an invalid prefix"); prev = prefixes[k2] + suffix_lens[k2]; total += (size_t)prev; if (total > ((size_t)1 << 31)) throw FormatError("delta-byte-array page expands beyond 2 GiB"); } }
            const size_t base = payload.size();                 // rebuilt values live behind the payload (+ its 16 pad bytes)
            payload.resize(base + total + 16);
            vals = payload.data() + values_off;                 // the resize may have moved the payload
            size_t q = used, w = base, prev_at = 0, prev_len = 0, d = 0;
            for (size_t i = 0; i < nvals; i++) {
              if (!valid[i]) { memset(out + 16 * i, 0, 16); continue; }
              const size_t pre = (size_t)prefixes[d], suf = (size_t)suffix_lens[d];
              d++;
              if (suf > vlen - q) throw FormatError("string suffix runs past the page");
              if (pre) memmove(payload.data() + w, payload.data() + prev_at, pre);
              if (suf) memcpy(payload.data() + w + pre, vals + q, suf);
              q += suf;
              make_view(out + 16 * i, payload.data() + w, (uint32_t)(pre + suf), (uint32_t)tk.buffer, (uint32_t)w);
              prev_at = w; prev_len = pre + suf; w += prev_len;
            }
          } else {
            throw Unsupported("column '" + leaf.name + "': page encoding " + encoding_name(h.encoding));
          }
          // validity bits of rows [row0, row0 + nvals): whole bytes are this page's alone, the first / last byte may be shared with a neighbour
          {
            uint64_t r = tk.row0;
            size_t i = 0;
            auto or_bit_range = [&](size_t upto) {        // rows i .. upto - 1 share one byte
              uint8_t bits = 0;
              for (; i < upto; i++, r++) bits |= (uint8_t)((valid[i] & 1u) << (r & 7));
              if (bits) __atomic_fetch_or(&validity[(size_t)((r - 1) >> 3)], bits, __ATOMIC_RELAXED);
            };
            if (r & 7) or_bit_range(std::min<size_t>(nvals, (size_t)(8 - (r & 7))));
            for (; i + 8 <= nvals; i += 8, r += 8) {
              uint8_t bits = 0;
              for (size_t j = 0; j < 8; j++) bits |= (uint8_t)((valid[i + j] & 1u) << j);
              validity[(size_t)(r >> 3)] = bits;
            }
            if (i < nvals) or_bit_range(nvals);
            nulls_of[t] += (int64_t)(nvals - present);
          }
        }
      } catch (...) { errs[t] = std::current_exception(); }
    };
    if (threads <= 1) { if (!tasks.empty()) work(0); }
    else {
      std::vector<std::thread> pool;
      for (size_t t = 0; t < threads; t++) pool.emplace_back(work, t);
      for (std::thread& th : pool) th.join();
    }
    for (std::exception_ptr& ep : errs) if (ep) std::rethrow_exception(ep);
    for (int64_t n : nulls_of) nulls += n;
  }
  res.null_count = nulls;
  std::vector<const void*> ptrs;
  std::vector<int64_t> sizes;
  for (const std::vector<uint8_t>& d : data) { ptrs.push_back(d.data()); sizes.push_back(d.size() >= 16 ? (int64_t)d.size() - 16 : 0); }
  trace_point(leaf.name, "string views built by the host threads");
  be.encode_string_views(f, leaf_idx, views, nulls ? validity.data() : nullptr, n_rows, ptrs, sizes, &res);
  trace_point(leaf.name, "strings encoded");
  return res;
}

// ---- fixed-width columns in encodings without a kernel: decoded by host threads ------------------------------------------------------------
// DELTA_BINARY_PACKED integers and BYTE_STREAM_SPLIT floats (what "v2" writers choose), INT96 timestamps (legacy Spark / Impala; -> us):
// host threads (a page each) decompress, decode levels and values, and write the finished column -- values + validity bitmap -- into
// the staging buffer; one upload, no kernel.  The chunk's PLAIN / dictionary pages (writers mix them) take the same route here.
template <class B> ColumnResult<B> read_fixed_column_host(B& be, File& f, const std::vector<int>& row_groups, int leaf_idx, ReadStats* stats) {
  const FileMetaData& md = f.md;
  const Leaf& leaf = md.leaves[leaf_idx];
  const LeafType lt = leaf_type(leaf);
  if (lt.dtype < 0) throw Unsupported("column '" + leaf.name + "': " + lt.why + " is outside the hot path's dtypes");
  if (lt.dtype == PLX_BOOL || leaf.type == PT_BYTE_ARRAY) throw Unsupported("column '" + leaf.name + "': no host value decoder for this type");
  const bool optional = leaf.repetition == REP_OPTIONAL;
  const bool int96 = leaf.type == PT_INT96;
  const uint32_t sw = lt.src_width, ow = out_width_of(lt.dtype);
  ColumnResult<B> res;
  res.dtype = lt.dtype; res.logical = lt.logical;
  struct ChunkRef { const ColumnChunk* c; int64_t rows; };
  std::vector<ChunkRef> chunks;
  int64_t n_rows = 0;
  for (int g : row_groups) {
    if (g < 0 || (size_t)g >= md.row_groups.size()) throw FormatError("row group index out of range");
    const RowGroup& rg = md.row_groups[g];
    const ColumnChunk& c = rg.columns[leaf_idx];
    if (!c.has_meta) throw FormatError("column chunk without metadata");
    if (c.external_file) throw Unsupported("column chunk stored in another file");
    if (c.type != leaf.type) throw FormatError("column chunk type differs from the schema");
    if (c.codec != CODEC_UNCOMPRESSED && c.codec != CODEC_SNAPPY && c.codec != CODEC_ZSTD && c.codec != CODEC_LZ4_RAW && c.codec != CODEC_GZIP)
      throw Unsupported(std::string("column '") + leaf.name + "': codec " + codec_name(c.codec) + " has no decompressor here");
    if (c.num_values != rg.num_rows) throw FormatError("flat column chunk whose value count differs from the row group's rows");
    if (rg.num_rows == 0) continue;
    if (c.start() < 4 || c.total_compressed_size < 0 || c.start() + c.total_compressed_size > f.size - 8) throw FormatError("column chunk outside the file");
    chunks.push_back({&c, rg.num_rows});
    n_rows += rg.num_rows;
  }
  res.len = n_rows;
  if (n_rows >= (int64_t)1 << 40) throw Unsupported("more than 2^40 rows in one read");
  const size_t out_bytes = (size_t)n_rows * ow;
  res.values = be.alloc(out_bytes + 8);
  if (n_rows == 0) return res;
  uint8_t* out = be.host_stage(out_bytes + 16);
  std::vector<uint8_t> validity((size_t)(n_rows + 7) / 8 + 8, 0);
  int64_t nulls = 0;
  uint64_t row0 = 0;
  auto put = [&](uint64_t row, uint64_t bits) {
    switch (ow) {
      case 1: out[row] = (uint8_t)bits; break;
      case 2: { uint16_t v = (uint16_t)bits; memcpy(out + 2 * row, &v, 2); break; }
      case 4: { uint32_t v = (uint32_t)bits; memcpy(out + 4 * row, &v, 4); break; }
      default: memcpy(out + 8 * row, &bits, 8); break;
    }
  };
  auto plain_at = [&](const uint8_t* v, size_t i) -> uint64_t {
    if (int96) {
      // int96_to_i64_ns (crates/polars-parquet/src/parquet/types.rs:222-234): Julian day (unsigned) and nanoseconds of the day ->
      // nanoseconds since the epoch, i64::MAX where that does not fit (simple.rs:731-733)
      int64_t nanos; uint32_t jd;
      memcpy(&nanos, v + 12 * i, 8); memcpy(&jd, v + 12 * i + 8, 4);
      const int64_t seconds = ((int64_t)jd - 2440588) * 86400;
      int64_t ns, sum;
      if (__builtin_mul_overflow(seconds, (int64_t)1000000000, &ns) || __builtin_add_overflow(ns, nanos, &sum)) return (uint64_t)INT64_MAX;
      return (uint64_t)sum;
    }
    if (sw == 8) return load_u64(v + 8 * i);
    return (uint64_t)load_u32(v + 4 * i);
  };
  struct Task { PageHeader h; const uint8_t* stored; uint64_t row0; };
  for (const ChunkRef& ch : chunks) {
    const ColumnChunk& c = *ch.c;
    const size_t sz = (size_t)c.total_compressed_size;
    std::vector<uint8_t> stored(sz + 16);
    f.pread_sliced(stored.data(), sz, c.start());
    if (stats) stats->file_bytes += sz;
    std::vector<uint64_t> dict;
    bool have_dict = false;
    std::vector<Task> tasks;
    size_t pos = 0;
    int64_t seen = 0;
    while (seen < c.num_values) {
      if (pos >= sz) throw FormatError("column chunk ends before all its values were found");
      PageHeader h = parse_page_header(stored.data() + pos, sz - pos);
      pos += h.header_bytes;
      if ((size_t)h.compressed_size > sz - pos) throw FormatError("page runs past its column chunk");
      if (h.type == PAGE_DICTIONARY) {
        if (have_dict) throw FormatError("two dictionary pages in one column chunk");
        if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICTIONARY) throw Unsupported(std::string("dictionary page encoding ") + encoding_name(h.encoding));
        std::vector<uint8_t> plain((size_t)h.uncompressed_size + 16);
        page_inflate(c.codec, stored.data() + pos, (size_t)h.compressed_size, plain.data(), (size_t)h.uncompressed_size);
        if ((uint64_t)h.num_values * sw > (uint64_t)h.uncompressed_size) throw FormatError("dictionary page smaller than its entry count");
        dict.resize((size_t)h.num_values);
        for (size_t i = 0; i < dict.size(); i++) dict[i] = plain_at(plain.data(), i);
        have_dict = true;
        if (stats) stats->dict_pages++;
      } else if (h.type == PAGE_DATA || h.type == PAGE_DATA_V2) {
        tasks.push_back({h, stored.data() + pos, row0 + (uint64_t)seen});
        seen += h.num_values;
        if (stats) stats->data_pages++;
      }
      pos += (size_t)h.compressed_size;
    }
    if (seen != c.num_values) throw FormatError("pages of a column chunk hold more values than its metadata says");
    std::vector<std::vector<uint8_t>> page_valid(tasks.size());
    const size_t threads = host_threads(tasks.size());
    std::vector<std::exception_ptr> errs(std::max<size_t>(threads, 1));
    auto work = [&](size_t t) {
      try {
        std::vector<uint8_t> payload;
        std::vector<int64_t> deltas;
        for (size_t k = t; k < tasks.size(); k += std::max<size_t>(threads, 1)) {
          const Task& tk = tasks[k];
          const PageHeader& h = tk.h;
          const bool v2 = h.type == PAGE_DATA_V2;
          const size_t nvals = (size_t)h.num_values;
          payload.assign((size_t)h.uncompressed_size + 16, 0);
          const uint8_t* levels = nullptr; size_t levels_len = 0, values_off = 0;
          if (v2) {
            if (h.rep_len != 0) throw Unsupported("repetition levels in a flat column");
            if (h.def_len < 0 || h.def_len > h.compressed_size || h.def_len > h.uncompressed_size) throw FormatError("v2 level bytes exceed the page");
            const size_t lv = (size_t)h.def_len;
            if (lv) memcpy(payload.data(), tk.stored, lv);
            page_inflate(h.is_compressed ? c.codec : (int)CODEC_UNCOMPRESSED, tk.stored + lv, (size_t)h.compressed_size - lv, payload.data() + lv, (size_t)h.uncompressed_size - lv);
            levels = payload.data(); levels_len = lv; values_off = lv;
          } else {
            page_inflate(c.codec, tk.stored, (size_t)h.compressed_size, payload.data(), (size_t)h.uncompressed_size);
            if (optional) {
              if (h.def_encoding != ENC_RLE) throw Unsupported(std::string("definition levels encoded as ") + encoding_name(h.def_encoding));
              if (h.uncompressed_size < 4) throw FormatError("page too small for its level length");
              const uint32_t ll = load_u32(payload.data());
              if (ll > (uint32_t)h.uncompressed_size - 4) throw FormatError("level bytes exceed the page");
              levels = payload.data() + 4; levels_len = ll; values_off = 4 + (size_t)ll;
            }
          }
          std::vector<uint8_t>& valid = page_valid[k];
          valid.assign(nvals, 1);
          size_t present = nvals;
          if (optional) {
            std::vector<uint32_t> lv = decode_hybrid_host(levels, levels_len, 1, nvals);
            present = 0;
            for (size_t i = 0; i < nvals; i++) { if (lv[i] > 1) throw FormatError("definition level > 1 in a flat column"); valid[i] = (uint8_t)lv[i]; present += lv[i]; }
          }
          const uint8_t* vals = payload.data() + values_off;
          const size_t vlen = (size_t)h.uncompressed_size - values_off;
          // dense value d of the page, as the bits of the source type
          std::vector<uint32_t> idx;
          deltas.clear();
          enum { K_PLAIN, K_DICT, K_DELTA, K_SPLIT } kind;
          if (h.encoding == ENC_PLAIN) {
            kind = K_PLAIN;
            if ((uint64_t)present * sw > vlen) throw FormatError("value bytes missing");
          } else if (h.encoding == ENC_PLAIN_DICTIONARY || h.encoding == ENC_RLE_DICTIONARY) {
            kind = K_DICT;
            if (!have_dict) throw FormatError("dictionary-encoded page without a dictionary page");
            if (vlen < 1 && present) throw FormatError("dictionary-encoded page without its bit width");
            idx = decode_hybrid_host(vals + (vlen ? 1 : 0), vlen ? vlen - 1 : 0, vlen ? vals[0] : 0, present);
          } else if (h.encoding == ENC_DELTA_BINARY_PACKED && !int96 && (leaf.type == PT_INT32 || leaf.type == PT_INT64)) {
            kind = K_DELTA;
            delta_binary_packed_host(vals, vlen, deltas);
            if (deltas.size() != present) throw FormatError("delta page with a different value count than its levels");
          } else if (h.encoding == ENC_BYTE_STREAM_SPLIT && !int96) {
            kind = K_SPLIT;
            if ((uint64_t)present * sw > vlen) throw FormatError("value bytes missing");
          } else {
            throw Unsupported("column '" + leaf.name + "': page encoding " + encoding_name(h.encoding));
          }
          size_t d = 0;
          for (size_t i = 0; i < nvals; i++) {
            if (!valid[i]) { put(tk.row0 + i, 0); continue; }
            uint64_t bits;
            switch (kind) {
              case K_PLAIN: bits = plain_at(vals, d); break;
              case K_DICT: { const uint32_t ix = idx[d]; if (ix >= dict.size()) throw FormatError("dictionary index out of range"); bits = dict[ix]; break; }
              case K_DELTA: bits = (uint64_t)deltas[d]; break;
              default: {
                bits = 0;
                for (uint32_t b = 0; b < sw; b++) bits |= (uint64_t)vals[(size_t)b * present + d] << (8 * b);
                break;
              }
            }
            d++;
            put(tk.row0 + i, bits);
          }
        }
      } catch (...) { errs[t] = std::current_exception(); }
    };
    if (threads <= 1) { if (!tasks.empty()) work(0); }
    else {
      std::vector<std::thread> pool;
      for (size_t t = 0; t < threads; t++) pool.emplace_back(work, t);
      for (std::thread& th : pool) th.join();
    }
    for (std::exception_ptr& ep : errs) if (ep) std::rethrow_exception(ep);
    for (size_t k = 0; k < tasks.size(); k++) {
      const std::vector<uint8_t>& valid = page_valid[k];
      uint64_t r = tasks[k].row0;
      for (size_t i = 0; i < valid.size(); i++, r++) {
        if (valid[i]) validity[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
        else nulls++;
      }
    }
    row0 += (uint64_t)ch.rows;
  }
  be.upload(be.addr(res.values), out, out_bytes);
  res.null_count = nulls;
  if (nulls) {
    const size_t vb = (size_t)((n_rows + 63) / 64) * 8;
    res.validity = be.alloc(vb + 8);
    be.zero(be.addr(res.validity), vb + 8);
    be.upload_small(be.addr(res.validity), validity.data(), (size_t)(n_rows + 7) / 8);
    res.has_validity = true;
  }
  be.discard_pending();            // the staging buffer of this column is about to be reused
  return res;
}

template <class B> ColumnResult<B> read_column(B& be, File& f, const std::vector<int>& row_groups, int leaf_idx, ReadStats* stats) {
  if (leaf_idx >= 0 && (size_t)leaf_idx < f.md.leaves.size() && f.md.leaves[leaf_idx].type == PT_INT96 && !f.md.leaves[leaf_idx].nested)
    return read_fixed_column_host(be, f, row_groups, leaf_idx, stats);
  try {
    return read_column_device(be, f, row_groups, leaf_idx, stats);
  } catch (const NeedsHostStrings&) {
    be.discard_pending();          // uploads of the abandoned attempt
    return read_string_column_host(be, f, row_groups, leaf_idx, stats);
  } catch (const NeedsHostValues&) {
    be.discard_pending();
    return read_fixed_column_host(be, f, row_groups, leaf_idx, stats);
  }
}

// ---- statistics of a chunk as typed scalars (row-group pruning) ---------------------------------------------------------------------------
// PLAIN-encoded single values: little-endian of the physical type.  Returns false when the chunk has none usable for the dtype
// (strings: byte-wise order says nothing about dictionary codes; deprecated fields are only right for signed types).
inline bool chunk_min_max(const Leaf& leaf, const ColumnChunk& c, plx_scalar* mn, plx_scalar* mx) {
  const LeafType lt = leaf_type(leaf);
  if (lt.dtype < 0 || leaf.type == PT_BYTE_ARRAY || !c.stats.has_min || !c.stats.has_max) return false;
  const bool is_unsigned = lt.dtype == PLX_U8 || lt.dtype == PLX_U16 || lt.dtype == PLX_U32 || lt.dtype == PLX_U64;
  if (c.stats.from_deprecated && is_unsigned) return false;
  auto one = [&](const std::string& raw, plx_scalar* out) {
    out->u = 0;
    switch (leaf.type) {
      case PT_BOOLEAN: if (raw.size() < 1) return false; out->u = raw[0] & 1; return true;
      case PT_INT32: {
        if (raw.size() < 4) return false;
        int32_t v; memcpy(&v, raw.data(), 4);
        if (is_unsigned) out->u = (uint32_t)v; else out->i = v;
        return true;
      }
      case PT_INT64: if (raw.size() < 8) return false; memcpy(&out->i, raw.data(), 8); return true;
      case PT_FLOAT: if (raw.size() < 4) return false; memcpy(&out->f32, raw.data(), 4); return true;
      case PT_DOUBLE: if (raw.size() < 8) return false; memcpy(&out->f64, raw.data(), 8); return true;
      default: return false;
    }
  };
  return one(c.stats.min, mn) && one(c.stats.max, mx);
}

}  // namespace pq
}  // namespace plx
