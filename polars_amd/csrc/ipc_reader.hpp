// ipc_reader.hpp -- the host-only half of the Arrow IPC scan: file open (magic, footer, record-batch metadata), column typing,
// buffer / node slots of a column inside a record batch, dictionary batches -> string lists.  No HIP: compiled into libpolars_amd.so
// (ipc.cpp adds the device half) and, unchanged, into the sanitizer driver of the tests (tests/emu/ipc_meta_main.cpp).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/polars_amd.h"
#include "file_io.hpp"
#include "host_codecs.hpp"
#include "ipc_format.hpp"

namespace plx {
namespace ipc {

struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

enum LogicalOut { LO_NONE = 0, LO_DATE = 1, LO_DATETIME_US = 2, LO_STRING = 3, LO_BINARY = 4, LO_DATETIME_MS = 5, LO_DATETIME_NS = 6 };

struct ColType {
  int dtype = -1;           // plx_dtype; -1: outside the hot path
  int logical = LO_NONE;
  int width = 0;            // bytes per value in the file (0: bit-packed)
  bool strings = false;     // needs a dictionary (host categories or device encode)
  std::string why;
};

inline ColType col_type(const Field& f) {
  ColType t;
  if (f.n_children > 0 && !f.has_dictionary) { t.why = std::string("nested type ") + type_name(f.type); return t; }
  const bool stringish = f.type == TY_UTF8 || f.type == TY_LARGE_UTF8 || f.type == TY_UTF8_VIEW || f.type == TY_BINARY ||
                         f.type == TY_LARGE_BINARY || f.type == TY_BINARY_VIEW;
  if (f.has_dictionary) {
    if (!stringish) { t.why = std::string("dictionary of ") + type_name(f.type); return t; }
    if (f.index_bits != 8 && f.index_bits != 16 && f.index_bits != 32 && f.index_bits != 64) { t.why = "dictionary index width"; return t; }
    t.dtype = PLX_U32; t.strings = true; t.width = f.index_bits / 8;
    t.logical = (f.type == TY_UTF8 || f.type == TY_LARGE_UTF8 || f.type == TY_UTF8_VIEW) ? LO_STRING : LO_BINARY;
    return t;
  }
  switch (f.type) {
    case TY_INT:
      t.width = f.bit_width / 8;
      switch (f.bit_width) {
        case 8: t.dtype = f.is_signed ? PLX_I8 : PLX_U8; return t;
        case 16: t.dtype = f.is_signed ? PLX_I16 : PLX_U16; return t;
        case 32: t.dtype = f.is_signed ? PLX_I32 : PLX_U32; return t;
        case 64: t.dtype = f.is_signed ? PLX_I64 : PLX_U64; return t;
        default: t.why = "integer width"; return t;
      }
    case TY_FLOAT:
      if (f.precision == 1) { t.dtype = PLX_F32; t.width = 4; return t; }
      if (f.precision == 2) { t.dtype = PLX_F64; t.width = 8; return t; }
      t.why = "half-precision float";
      return t;
    case TY_BOOL: t.dtype = PLX_BOOL; t.width = 0; return t;
    case TY_DATE:
      if (f.unit == 0) { t.dtype = PLX_I32; t.logical = LO_DATE; t.width = 4; return t; }
      t.why = "date in milliseconds";
      return t;
    case TY_TIMESTAMP:
      if (f.unit == 2) { t.dtype = PLX_I64; t.logical = LO_DATETIME_US; t.width = 8; return t; }
      if (f.unit == 1) { t.dtype = PLX_I64; t.logical = LO_DATETIME_MS; t.width = 8; return t; }
      if (f.unit == 3) { t.dtype = PLX_I64; t.logical = LO_DATETIME_NS; t.width = 8; return t; }
      t.why = "timestamp in seconds";                    // the reference multiplies these into milliseconds on import: not on this path
      return t;
    case TY_UTF8: case TY_LARGE_UTF8: case TY_UTF8_VIEW:
      t.dtype = PLX_U32; t.logical = LO_STRING; t.strings = true; return t;
    case TY_BINARY: case TY_LARGE_BINARY: case TY_BINARY_VIEW:
      t.dtype = PLX_U32; t.logical = LO_BINARY; t.strings = true; return t;
    default: t.why = type_name(f.type); return t;
  }
}

struct File : FileReader {
  Footer footer;
  std::vector<BatchMeta> batches;              // metadata of every record batch block (read at open: lengths give the row count)
  std::vector<int64_t> body_off;                    // file offset of every batch body
  int64_t num_rows = 0;
  std::map<int64_t, std::vector<std::string>> dicts;   // dictionary id -> values (read on first use)
  bool dicts_loaded = false;
  std::map<int, plx_strdict> strdicts;              // column -> device dictionary of its last read (strings encoded on the device)
};

// the Message flatbuffer of a block: [0xFFFFFFFF] [i32 size] bytes ...   (files written before 0.15 lack the continuation word)
inline BatchMeta read_block_meta(const File& f, const Block& b, int64_t* body) {
  if (b.offset < 8 || b.offset + b.meta_len + b.body_len > f.size) throw FormatError("block outside the file");
  if (b.meta_len < 8) throw FormatError("block metadata too short");
  std::vector<uint8_t> m((size_t)b.meta_len);
  f.pread_exact(m.data(), m.size(), b.offset);
  uint32_t w0, w1;
  memcpy(&w0, m.data(), 4); memcpy(&w1, m.data() + 4, 4);
  size_t at = 4, len = w0;
  if (w0 == 0xffffffffu) { at = 8; len = w1; }
  if (len > m.size() - at) throw FormatError("message longer than its block");
  *body = b.offset + b.meta_len;
  BatchMeta bm = parse_message(m.data() + at, len);
  for (const BufferRef& r : bm.buffers)
    if (r.offset + r.length > b.body_len) throw FormatError("buffer outside its message body");
  return bm;
}

inline std::unique_ptr<File> open_file(const std::string& path) {
  auto f = std::make_unique<File>();
  f->open(path);
  if (f->size < 24) throw FormatError(path + " is too small to be an Arrow IPC file");
  uint8_t head[8], tail[10];
  f->pread_exact(head, 8, 0);
  f->pread_exact(tail, 10, f->size - 10);
  if (memcmp(head, "ARROW1", 6) != 0 || memcmp(tail + 4, "ARROW1", 6) != 0) throw FormatError(path + " is not an Arrow IPC file (ARROW1 magic)");
  int32_t flen;
  memcpy(&flen, tail, 4);
  if (flen <= 0 || (int64_t)flen + 18 > f->size) throw FormatError("footer length past the start of the file");
  std::vector<uint8_t> fb((size_t)flen);
  f->pread_exact(fb.data(), fb.size(), f->size - 10 - flen);
  f->footer = parse_footer(fb.data(), fb.size());
  for (const Block& b : f->footer.batches) {
    int64_t body = 0;
    BatchMeta bm = read_block_meta(*f, b, &body);
    if (bm.is_dictionary) throw FormatError("dictionary batch listed as a record batch");
    f->num_rows += bm.length;
    f->batches.push_back(std::move(bm));
    f->body_off.push_back(body);
  }
  return f;
}

// where field `col`'s buffers and nodes start in a batch (fields before it take n_buffers (+ their variadic data buffers) / n_nodes)
struct Slot { size_t buf = 0, node = 0, variadic = 0; };
inline Slot slot_of(const File& f, const BatchMeta& bm, int col) {
  Slot s;
  for (int i = 0; i < col; i++) {
    const Field& fl = f.footer.fields[i];
    s.buf += (size_t)fl.n_buffers; s.node += (size_t)fl.n_nodes;
    if (fl.variadic) {
      if (s.variadic >= bm.variadic_counts.size()) throw FormatError("variadicBufferCounts shorter than the view columns");
      s.buf += (size_t)bm.variadic_counts[s.variadic++];
    }
  }
  return s;
}

// Body compression (Message.fbs BodyCompression): every buffer of a compressed record batch starts with its uncompressed length as an
// int64 (-1: the bytes that follow are stored as they are), then one LZ4 frame / Zstandard frame.
inline int64_t compressed_buffer_length(const File& f, int64_t body, const BufferRef& r) {
  if (r.length == 0) return 0;
  if (r.length < 8) throw FormatError("compressed buffer shorter than its length prefix");
  int64_t ulen;
  f.pread_exact(&ulen, 8, body + r.offset);
  // no LZ4 / zstd stream expands more than ~32768 : 1 (a zstd RLE block: 4 bytes -> 128 KB): anything beyond is a corrupt length, and
  // must not become an allocation
  if (ulen < -1 || ulen > ((int64_t)1 << 31) || ulen > 65536 * (r.length - 8) + 1024) throw FormatError("compressed buffer with an absurd uncompressed length");
  return ulen == -1 ? r.length - 8 : ulen;
}
// the bytes of one buffer, decompressed when the batch is compressed, into dst[0, need) (need <= its uncompressed length)
inline void load_buffer(const File& f, const BatchMeta& bm, int64_t body, const BufferRef& r, uint8_t* dst, size_t need) {
  if (!need) return;
  if (!bm.compressed) {
    if ((int64_t)need > r.length) throw FormatError("buffer shorter than the array needs");
    f.pread_sliced(dst, need, body + r.offset);
    return;
  }
  if (r.length < 8) throw FormatError("compressed buffer shorter than its length prefix");
  std::vector<uint8_t> raw((size_t)r.length);
  f.pread_sliced(raw.data(), raw.size(), body + r.offset);
  int64_t ulen;
  memcpy(&ulen, raw.data(), 8);
  if (ulen == -1) {
    if ((int64_t)need > r.length - 8) throw FormatError("buffer shorter than the array needs");
    memcpy(dst, raw.data() + 8, need);
    return;
  }
  if (ulen < 0 || (int64_t)need > ulen) throw FormatError("buffer shorter than the array needs");
  if (ulen > ((int64_t)1 << 31) || ulen > 65536 * (r.length - 8) + 1024) throw FormatError("compressed buffer with an absurd uncompressed length");
  try {
    if ((size_t)ulen == need) {
      if (bm.codec == 0) codec::lz4_frame_decompress(raw.data() + 8, raw.size() - 8, dst, need);
      else codec::zstd_decompress(raw.data() + 8, raw.size() - 8, dst, need);
    } else {                                       // padded buffers: decompress all of it, hand out the front
      std::vector<uint8_t> full((size_t)ulen);
      if (bm.codec == 0) codec::lz4_frame_decompress(raw.data() + 8, raw.size() - 8, full.data(), full.size());
      else codec::zstd_decompress(raw.data() + 8, raw.size() - 8, full.data(), full.size());
      memcpy(dst, full.data(), need);
    }
  } catch (const codec::CodecError& e) { throw FormatError(std::string("compressed buffer: ") + e.what()); }
}
// a whole buffer (+ 16 readable pad bytes behind it)
inline std::vector<uint8_t> read_buffer(const File& f, const BatchMeta& bm, int64_t body, const BufferRef& r) {
  const int64_t n = bm.compressed ? compressed_buffer_length(f, body, r) : r.length;
  std::vector<uint8_t> v((size_t)n + 16, 0);
  load_buffer(f, bm, body, r, v.data(), (size_t)n);
  return v;
}

// strings of one array laid out as [validity, offsets, data] / [validity, views, data...] -> appended to out
inline void decode_strings(const File& f, const Field& fl, const BatchMeta& bm, int64_t body, size_t buf0, size_t node0, size_t var0, std::vector<std::string>* out) {
  if (node0 >= bm.nodes.size() || buf0 + 2 > bm.buffers.size()) throw FormatError("string array without its buffers");
  const int64_t n = bm.nodes[node0].length;
  if (fl.type == TY_UTF8_VIEW || fl.type == TY_BINARY_VIEW) {
    const int64_t nvar = var0 < bm.variadic_counts.size() ? bm.variadic_counts[var0] : 0;
    if (buf0 + 2 + (size_t)nvar > bm.buffers.size()) throw FormatError("view array without its data buffers");
    std::vector<uint8_t> views = read_buffer(f, bm, body, bm.buffers[buf0 + 1]);
    if ((int64_t)views.size() - 16 < n * 16) throw FormatError("views buffer shorter than the array");
    std::vector<std::vector<uint8_t>> data;
    for (int64_t k = 0; k < nvar; k++) data.push_back(read_buffer(f, bm, body, bm.buffers[buf0 + 2 + (size_t)k]));
    for (int64_t i = 0; i < n; i++) {
      const uint8_t* v = views.data() + 16 * i;
      uint32_t len, bi, off;
      memcpy(&len, v, 4);
      if (len <= 12) { out->emplace_back((const char*)v + 4, len); continue; }
      memcpy(&bi, v + 8, 4); memcpy(&off, v + 12, 4);
      if (bi >= data.size() || (uint64_t)off + len > data[bi].size() - 16) throw FormatError("view points outside its data buffer");
      out->emplace_back((const char*)data[bi].data() + off, len);
    }
    return;
  }
  if (buf0 + 3 > bm.buffers.size()) throw FormatError("string array without its buffers");
  const bool large = fl.type == TY_LARGE_UTF8 || fl.type == TY_LARGE_BINARY;
  std::vector<uint8_t> offs = read_buffer(f, bm, body, bm.buffers[buf0 + 1]), data = read_buffer(f, bm, body, bm.buffers[buf0 + 2]);
  const size_t ow = large ? 8 : 4;
  if (n && (int64_t)offs.size() - 16 < (n + 1) * (int64_t)ow) throw FormatError("offsets buffer shorter than the array");
  auto off_at = [&](int64_t i) -> int64_t {
    if (large) { int64_t v; memcpy(&v, offs.data() + 8 * i, 8); return v; }
    int32_t v; memcpy(&v, offs.data() + 4 * i, 4); return v;
  };
  for (int64_t i = 0; i < n; i++) {
    int64_t a = off_at(i), b = off_at(i + 1);
    if (a < 0 || b < a || b > (int64_t)data.size() - 16) throw FormatError("string offsets outside the data buffer");
    out->emplace_back((const char*)data.data() + a, (size_t)(b - a));
  }
}

inline void load_dictionaries(File& f) {
  if (f.dicts_loaded) return;
  for (const Block& b : f.footer.dictionaries) {
    int64_t body = 0;
    BatchMeta bm = read_block_meta(f, b, &body);
    if (!bm.is_dictionary) throw FormatError("record batch listed as a dictionary");
    if (bm.compressed && bm.codec != 0 && bm.codec != 1) throw Unsupported("dictionary batch compressed with an unknown codec");
    const Field* fl = nullptr;
    for (const Field& x : f.footer.fields) if (x.has_dictionary && x.dict_id == bm.dict_id) { fl = &x; break; }
    if (!fl || col_type(*fl).dtype < 0) continue;       // dictionary of a column outside the hot path
    std::vector<std::string>& d = f.dicts[bm.dict_id];
    if (!bm.is_delta) d.clear();
    decode_strings(f, *fl, bm, body, 0, 0, 0, &d);
  }
  f.dicts_loaded = true;
}

}  // namespace ipc
}  // namespace plx
