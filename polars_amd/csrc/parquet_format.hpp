// parquet_format.hpp -- the metadata side of a Parquet file, host only: Thrift compact-protocol reader, FileMetaData,
// PageHeader, schema -> leaf columns.  No HIP, no third-party code: it is compiled into libpolars_amd.so and, unchanged, into the
// CPU harness of the tests (tests/emu/parquet_emu.cpp).
//
// Reference counterparts: crates/polars-parquet/src/parquet/read/metadata.rs (footer: 4-byte length + "PAR1"),
// parquet/metadata/{file_metadata.rs,column_chunk_metadata.rs,schema_descriptor.rs}, parquet/read/page/reader.rs:183-300 (page
// header walk), parquet/handwritten_thrift/*.  Field ids are the ones of parquet-format's parquet.thrift.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace plx {
namespace pq {

struct FormatError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

enum PhysicalType { PT_BOOLEAN = 0, PT_INT32 = 1, PT_INT64 = 2, PT_INT96 = 3, PT_FLOAT = 4, PT_DOUBLE = 5, PT_BYTE_ARRAY = 6, PT_FIXED_LEN_BYTE_ARRAY = 7 };
enum Encoding {
  ENC_PLAIN = 0, ENC_PLAIN_DICTIONARY = 2, ENC_RLE = 3, ENC_BIT_PACKED = 4, ENC_DELTA_BINARY_PACKED = 5, ENC_DELTA_LENGTH_BYTE_ARRAY = 6,
  ENC_DELTA_BYTE_ARRAY = 7, ENC_RLE_DICTIONARY = 8, ENC_BYTE_STREAM_SPLIT = 9
};
enum Codec { CODEC_UNCOMPRESSED = 0, CODEC_SNAPPY = 1, CODEC_GZIP = 2, CODEC_LZO = 3, CODEC_BROTLI = 4, CODEC_LZ4 = 5, CODEC_ZSTD = 6, CODEC_LZ4_RAW = 7 };
enum PageType { PAGE_DATA = 0, PAGE_INDEX = 1, PAGE_DICTIONARY = 2, PAGE_DATA_V2 = 3 };
enum Repetition { REP_REQUIRED = 0, REP_OPTIONAL = 1, REP_REPEATED = 2 };
// what the logical / converted type annotations boil down to for the hot path's dtypes
enum Logical {
  LG_NONE = 0, LG_STRING = 1, LG_DATE = 2, LG_TIMESTAMP_MILLIS = 3, LG_TIMESTAMP_MICROS = 4, LG_TIMESTAMP_NANOS = 5, LG_INT = 6, LG_DECIMAL = 7,
  LG_OTHER = 8   // anything else (TIME, JSON, UUID, LIST, MAP, ...): outside the hot path
};

inline const char* codec_name(int c) {
  static const char* n[] = {"UNCOMPRESSED", "SNAPPY", "GZIP", "LZO", "BROTLI", "LZ4", "ZSTD", "LZ4_RAW"};
  return c >= 0 && c < 8 ? n[c] : "?";
}
inline const char* encoding_name(int e) {
  static const char* n[] = {"PLAIN", "?", "PLAIN_DICTIONARY", "RLE", "BIT_PACKED", "DELTA_BINARY_PACKED", "DELTA_LENGTH_BYTE_ARRAY", "DELTA_BYTE_ARRAY",
                            "RLE_DICTIONARY", "BYTE_STREAM_SPLIT"};
  return e >= 0 && e < 10 ? n[e] : "?";
}

// ---- Thrift compact protocol ------------------------------------------------------------------------------------------------
enum ThriftType { T_STOP = 0, T_TRUE = 1, T_FALSE = 2, T_BYTE = 3, T_I16 = 4, T_I32 = 5, T_I64 = 6, T_DOUBLE = 7, T_BINARY = 8, T_LIST = 9, T_SET = 10, T_MAP = 11, T_STRUCT = 12 };

class ThriftReader {
 public:
  ThriftReader(const uint8_t* p, size_t n) : p_(p), begin_(p), end_(p + n) {}
  size_t consumed() const { return (size_t)(p_ - begin_); }

  uint8_t u8() {
    if (p_ >= end_) throw FormatError("thrift: unexpected end of data");
    return *p_++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      uint8_t b = u8();
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    throw FormatError("thrift: varint too long");
  }
  int64_t zigzag() {
    uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  std::string binary() {
    uint64_t n = varint();
    if (n > (uint64_t)(end_ - p_)) throw FormatError("thrift: binary longer than the buffer");
    std::string s((const char*)p_, (size_t)n);
    p_ += n;
    return s;
  }
  void skip_binary() {
    uint64_t n = varint();
    if (n > (uint64_t)(end_ - p_)) throw FormatError("thrift: binary longer than the buffer");
    p_ += n;
  }
  void list_header(int* elem_type, uint32_t* size) {
    uint8_t b = u8();
    *elem_type = b & 15;
    uint32_t n = b >> 4;
    if (n == 15) n = (uint32_t)varint();
    *size = n;
  }
  // on_field(id, type) -> true if it consumed the value; booleans carry their value in `type`
  template <class F> void read_struct(F&& on_field) {
    if (++depth_ > 32) throw FormatError("thrift: nesting too deep");
    int16_t last = 0;
    for (;;) {
      uint8_t b = u8();
      if (b == T_STOP) break;
      int type = b & 15, delta = b >> 4;
      int16_t id = delta ? (int16_t)(last + delta) : (int16_t)zigzag();
      last = id;
      if (!on_field(id, type)) skip(type);
    }
    --depth_;
  }
  void skip(int type) {
    switch (type) {
      case T_TRUE: case T_FALSE: return;
      case T_BYTE: u8(); return;
      case T_I16: case T_I32: case T_I64: varint(); return;
      case T_DOUBLE:
        if (end_ - p_ < 8) throw FormatError("thrift: unexpected end of data");
        p_ += 8;
        return;
      case T_BINARY: skip_binary(); return;
      case T_LIST: case T_SET: {
        int et; uint32_t n;
        list_header(&et, &n);
        for (uint32_t i = 0; i < n; i++) {
          if (et == T_TRUE || et == T_FALSE) u8();   // booleans inside a list take one byte each
          else skip(et);
        }
        return;
      }
      case T_MAP: {
        uint64_t n = varint();
        if (!n) return;
        uint8_t kv = u8();
        for (uint64_t i = 0; i < n; i++) { skip_elem(kv >> 4); skip_elem(kv & 15); }
        return;
      }
      case T_STRUCT: read_struct([](int16_t, int) { return false; }); return;
      default: throw FormatError("thrift: unknown field type " + std::to_string(type));
    }
  }

 private:
  void skip_elem(int t) { if (t == T_TRUE || t == T_FALSE) u8(); else skip(t); }
  const uint8_t* p_;
  const uint8_t* begin_;
  const uint8_t* end_;
  int depth_ = 0;
};

// ---- file metadata ----------------------------------------------------------------------------------------------------------------
struct Statistics {
  bool has_min = false, has_max = false, has_null_count = false;
  bool from_deprecated = false;      // min / max came from the deprecated fields 1 / 2 (signed byte-wise order)
  std::string min, max;              // PLAIN-encoded single values
  int64_t null_count = 0;
};

struct ColumnChunk {
  int type = -1;
  int codec = 0;
  uint32_t encodings = 0;            // bit e set: Encoding e appears in the chunk
  int64_t num_values = 0;
  int64_t total_uncompressed_size = 0, total_compressed_size = 0;
  int64_t data_page_offset = 0, dictionary_page_offset = -1;
  bool has_meta = false;
  bool external_file = false;        // file_path set: the chunk lives in another file
  Statistics stats;
  // byte range of the chunk in the file: [start, start + total_compressed_size)
  int64_t start() const {
    return dictionary_page_offset > 0 && (data_page_offset <= 0 || dictionary_page_offset < data_page_offset) ? dictionary_page_offset : data_page_offset;
  }
};

struct RowGroup {
  std::vector<ColumnChunk> columns;  // one per leaf, in schema (depth-first) order
  int64_t num_rows = 0;
  int64_t total_byte_size = 0;
};

struct Leaf {
  std::string name;                  // top-level field name (nested leaves: dotted path)
  int type = -1;                     // PhysicalType
  int type_length = 0;
  int repetition = REP_REQUIRED;
  int logical = LG_NONE;
  int int_bits = 0;                  // LG_INT: 8 / 16 / 32 / 64
  bool int_signed = true;
  bool utc = false;
  bool nested = false;               // below a group or repeated: outside the hot path's flat columns
};

struct SchemaElement {
  int type = -1, type_length = 0, repetition = REP_REQUIRED, num_children = 0, converted = -1;
  int logical = LG_NONE, int_bits = 0;
  bool int_signed = true, utc = false, has_logical = false;
  std::string name;
};

struct FileMetaData {
  int32_t version = 0;
  int64_t num_rows = 0;
  std::string created_by;
  std::vector<Leaf> leaves;
  std::vector<RowGroup> row_groups;
};

namespace detail {

inline void parse_logical(ThriftReader& r, SchemaElement& e) {
  e.has_logical = true;
  e.logical = LG_OTHER;
  r.read_struct([&](int16_t id, int type) {
    if (type != T_STRUCT) return false;
    switch (id) {
      case 1: e.logical = LG_STRING; return false;     // the empty member struct is skipped by the caller
      case 5: e.logical = LG_DECIMAL; return false;
      case 6: e.logical = LG_DATE; return false;
      case 8: {                                        // TIMESTAMP { 1: isAdjustedToUTC, 2: unit { 1 MILLIS | 2 MICROS | 3 NANOS } }
        r.read_struct([&](int16_t fid, int ft) {
          if (fid == 1 && (ft == T_TRUE || ft == T_FALSE)) { e.utc = ft == T_TRUE; return true; }
          if (fid == 2 && ft == T_STRUCT) {
            r.read_struct([&](int16_t uid, int ut) {
              if (ut == T_STRUCT) e.logical = uid == 1 ? LG_TIMESTAMP_MILLIS : uid == 2 ? LG_TIMESTAMP_MICROS : uid == 3 ? LG_TIMESTAMP_NANOS : LG_OTHER;
              return false;
            });
            return true;
          }
          return false;
        });
        return true;
      }
      case 10: {                                       // INTEGER { 1: i8 bitWidth, 2: bool isSigned }
        e.logical = LG_INT;
        r.read_struct([&](int16_t fid, int ft) {
          if (fid == 1 && ft == T_BYTE) { e.int_bits = (int8_t)r.u8(); return true; }
          if (fid == 2 && (ft == T_TRUE || ft == T_FALSE)) { e.int_signed = ft == T_TRUE; return true; }
          return false;
        });
        return true;
      }
      default: return false;
    }
  });
}

inline SchemaElement parse_schema_element(ThriftReader& r) {
  SchemaElement e;
  r.read_struct([&](int16_t id, int type) {
    switch (id) {
      case 1: if (type == T_I32) { e.type = (int)r.zigzag(); return true; } return false;
      case 2: if (type == T_I32) { e.type_length = (int)r.zigzag(); return true; } return false;
      case 3: if (type == T_I32) { e.repetition = (int)r.zigzag(); return true; } return false;
      case 4: if (type == T_BINARY) { e.name = r.binary(); return true; } return false;
      case 5: if (type == T_I32) { e.num_children = (int)r.zigzag(); return true; } return false;
      case 6: if (type == T_I32) { e.converted = (int)r.zigzag(); return true; } return false;
      case 10: if (type == T_STRUCT) { parse_logical(r, e); return true; } return false;
      default: return false;
    }
  });
  if (!e.has_logical && e.converted >= 0) {            // ConvertedType (deprecated annotation) -> the same summary
    switch (e.converted) {
      case 0: e.logical = LG_STRING; break;
      case 5: e.logical = LG_DECIMAL; break;
      case 6: e.logical = LG_DATE; break;
      case 9: e.logical = LG_TIMESTAMP_MILLIS; e.utc = true; break;
      case 10: e.logical = LG_TIMESTAMP_MICROS; e.utc = true; break;
      case 11: case 12: case 13: case 14: e.logical = LG_INT; e.int_signed = false; e.int_bits = 8 << (e.converted - 11); break;
      case 15: case 16: case 17: case 18: e.logical = LG_INT; e.int_signed = true; e.int_bits = 8 << (e.converted - 15); break;
      default: e.logical = LG_OTHER; break;
    }
  }
  return e;
}

inline Statistics parse_statistics(ThriftReader& r) {
  Statistics s;
  std::string dmin, dmax;
  bool has_dmin = false, has_dmax = false;
  r.read_struct([&](int16_t id, int type) {
    switch (id) {
      case 1: if (type == T_BINARY) { dmax = r.binary(); has_dmax = true; return true; } return false;
      case 2: if (type == T_BINARY) { dmin = r.binary(); has_dmin = true; return true; } return false;
      case 3: if (type == T_I64) { s.null_count = r.zigzag(); s.has_null_count = true; return true; } return false;
      case 5: if (type == T_BINARY) { s.max = r.binary(); s.has_max = true; return true; } return false;
      case 6: if (type == T_BINARY) { s.min = r.binary(); s.has_min = true; return true; } return false;
      default: return false;
    }
  });
  if (!s.has_min && !s.has_max && has_dmin && has_dmax) {
    s.min = dmin; s.max = dmax; s.has_min = s.has_max = true; s.from_deprecated = true;
  }
  return s;
}

inline void parse_column_meta(ThriftReader& r, ColumnChunk& c) {
  c.has_meta = true;
  r.read_struct([&](int16_t id, int type) {
    switch (id) {
      case 1: if (type == T_I32) { c.type = (int)r.zigzag(); return true; } return false;
      case 2:
        if (type == T_LIST) {
          int et; uint32_t n;
          r.list_header(&et, &n);
          for (uint32_t i = 0; i < n; i++) { int64_t e = r.zigzag(); if (e >= 0 && e < 32) c.encodings |= 1u << e; }
          return true;
        }
        return false;
      case 4: if (type == T_I32) { c.codec = (int)r.zigzag(); return true; } return false;
      case 5: if (type == T_I64) { c.num_values = r.zigzag(); return true; } return false;
      case 6: if (type == T_I64) { c.total_uncompressed_size = r.zigzag(); return true; } return false;
      case 7: if (type == T_I64) { c.total_compressed_size = r.zigzag(); return true; } return false;
      case 9: if (type == T_I64) { c.data_page_offset = r.zigzag(); return true; } return false;
      case 11: if (type == T_I64) { c.dictionary_page_offset = r.zigzag(); return true; } return false;
      case 12: if (type == T_STRUCT) { c.stats = parse_statistics(r); return true; } return false;
      default: return false;
    }
  });
}

inline ColumnChunk parse_column_chunk(ThriftReader& r) {
  ColumnChunk c;
  r.read_struct([&](int16_t id, int type) {
    if (id == 1 && type == T_BINARY) { c.external_file = !r.binary().empty(); return true; }
    if (id == 3 && type == T_STRUCT) { parse_column_meta(r, c); return true; }
    return false;
  });
  return c;
}

inline RowGroup parse_row_group(ThriftReader& r) {
  RowGroup g;
  r.read_struct([&](int16_t id, int type) {
    if (id == 1 && type == T_LIST) {
      int et; uint32_t n;
      r.list_header(&et, &n);
      if (et != T_STRUCT) throw FormatError("row group: columns is not a list of structs");
      g.columns.reserve(n);
      for (uint32_t i = 0; i < n; i++) g.columns.push_back(parse_column_chunk(r));
      return true;
    }
    if (id == 2 && type == T_I64) { g.total_byte_size = r.zigzag(); return true; }
    if (id == 3 && type == T_I64) { g.num_rows = r.zigzag(); return true; }
    return false;
  });
  return g;
}

// depth-first walk of the flattened schema list (parquet/metadata/schema_descriptor.rs): one Leaf per primitive column
inline size_t collect_leaves(const std::vector<SchemaElement>& els, size_t i, const std::string& prefix, bool nested, int depth, std::vector<Leaf>& out) {
  if (i >= els.size()) throw FormatError("schema: child count runs past the element list");
  if (depth > 64) throw FormatError("schema: nesting too deep");
  const SchemaElement& e = els[i];
  std::string path = prefix.empty() ? e.name : prefix + "." + e.name;
  if (e.num_children > 0) {
    size_t j = i + 1;
    for (int c = 0; c < e.num_children; c++) j = collect_leaves(els, j, path, true, depth + 1, out);
    return j;
  }
  Leaf l;
  l.name = path; l.type = e.type; l.type_length = e.type_length; l.repetition = e.repetition; l.logical = e.logical; l.int_bits = e.int_bits;
  l.int_signed = e.int_signed; l.utc = e.utc; l.nested = nested || e.repetition == REP_REPEATED;
  out.push_back(std::move(l));
  return i + 1;
}

}  // namespace detail

// footer bytes (the Thrift-encoded FileMetaData, without the trailing length + magic) -> FileMetaData
inline FileMetaData parse_file_metadata(const uint8_t* p, size_t n) {
  ThriftReader r(p, n);
  FileMetaData md;
  std::vector<SchemaElement> els;
  r.read_struct([&](int16_t id, int type) {
    switch (id) {
      case 1: if (type == T_I32) { md.version = (int32_t)r.zigzag(); return true; } return false;
      case 2:
        if (type == T_LIST) {
          int et; uint32_t cnt;
          r.list_header(&et, &cnt);
          if (et != T_STRUCT) throw FormatError("schema is not a list of structs");
          els.reserve(cnt);
          for (uint32_t i = 0; i < cnt; i++) els.push_back(detail::parse_schema_element(r));
          return true;
        }
        return false;
      case 3: if (type == T_I64) { md.num_rows = r.zigzag(); return true; } return false;
      case 4:
        if (type == T_LIST) {
          int et; uint32_t cnt;
          r.list_header(&et, &cnt);
          if (et != T_STRUCT) throw FormatError("row_groups is not a list of structs");
          md.row_groups.reserve(cnt);
          for (uint32_t i = 0; i < cnt; i++) md.row_groups.push_back(detail::parse_row_group(r));
          return true;
        }
        return false;
      case 6: if (type == T_BINARY) { md.created_by = r.binary(); return true; } return false;
      default: return false;
    }
  });
  if (els.empty()) throw FormatError("file metadata has no schema");
  size_t j = 1;
  for (int c = 0; c < els[0].num_children; c++) j = detail::collect_leaves(els, j, "", false, 0, md.leaves);
  for (const RowGroup& g : md.row_groups)
    if (g.columns.size() != md.leaves.size()) throw FormatError("row group has " + std::to_string(g.columns.size()) + " column chunks for " + std::to_string(md.leaves.size()) + " leaf columns");
  return md;
}

// ---- page headers ---------------------------------------------------------------------------------------------------------------
struct PageHeader {
  int type = -1;
  int32_t uncompressed_size = 0, compressed_size = 0;
  int32_t num_values = 0;
  int encoding = -1;
  int def_encoding = ENC_RLE, rep_encoding = ENC_RLE;
  // DATA_PAGE_V2
  int32_t num_nulls = -1, num_rows = -1, def_len = 0, rep_len = 0;
  bool is_compressed = true;
  size_t header_bytes = 0;           // bytes of the Thrift header itself; the payload follows
};

inline PageHeader parse_page_header(const uint8_t* p, size_t n) {
  ThriftReader r(p, n);
  PageHeader h;
  auto data_v1 = [&]() {
    r.read_struct([&](int16_t id, int type) {
      if (type != T_I32) return false;
      int32_t v = (int32_t)r.zigzag();
      if (id == 1) h.num_values = v; else if (id == 2) h.encoding = v; else if (id == 3) h.def_encoding = v; else if (id == 4) h.rep_encoding = v;
      return true;
    });
  };
  auto dict = [&]() {
    r.read_struct([&](int16_t id, int type) {
      if (type != T_I32) return false;
      int32_t v = (int32_t)r.zigzag();
      if (id == 1) h.num_values = v; else if (id == 2) h.encoding = v;
      return true;
    });
  };
  auto data_v2 = [&]() {
    r.read_struct([&](int16_t id, int type) {
      if (id == 7 && (type == T_TRUE || type == T_FALSE)) { h.is_compressed = type == T_TRUE; return true; }
      if (type != T_I32) return false;
      int32_t v = (int32_t)r.zigzag();
      switch (id) {
        case 1: h.num_values = v; break;
        case 2: h.num_nulls = v; break;
        case 3: h.num_rows = v; break;
        case 4: h.encoding = v; break;
        case 5: h.def_len = v; break;
        case 6: h.rep_len = v; break;
        default: break;
      }
      return true;
    });
  };
  r.read_struct([&](int16_t id, int type) {
    switch (id) {
      case 1: if (type == T_I32) { h.type = (int)r.zigzag(); return true; } return false;
      case 2: if (type == T_I32) { h.uncompressed_size = (int32_t)r.zigzag(); return true; } return false;
      case 3: if (type == T_I32) { h.compressed_size = (int32_t)r.zigzag(); return true; } return false;
      case 5: if (type == T_STRUCT) { data_v1(); return true; } return false;
      case 7: if (type == T_STRUCT) { dict(); return true; } return false;
      case 8: if (type == T_STRUCT) { data_v2(); return true; } return false;
      default: return false;
    }
  });
  h.header_bytes = r.consumed();
  if (h.compressed_size < 0 || h.uncompressed_size < 0 || h.num_values < 0) throw FormatError("page header with negative sizes");
  // no codec expands more than ~32768 : 1 (a zstd RLE block); a corrupt size must not become an allocation
  if ((int64_t)h.uncompressed_size > 65536 * (int64_t)h.compressed_size + 4096) throw FormatError("page header with an absurd uncompressed size");
  return h;
}

}  // namespace pq
}  // namespace plx
