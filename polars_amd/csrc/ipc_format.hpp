// ipc_format.hpp -- the metadata side of an Arrow IPC file (Feather V2), host only: a minimal FlatBuffers reader and the Footer /
// Schema / Message / RecordBatch tables of Arrow's format (Schema.fbs, Message.fbs, File.fbs).  No generated code, no third-party
// library; compiled into libpolars_amd.so and, unchanged, into the CPU harness of the tests.
//
// Reference counterparts: crates/polars-arrow/src/io/ipc/read/{file.rs (footer: "ARROW1" + i32 size at the end, blocks),
// schema.rs (fields, dictionaries), common.rs (record batch -> arrays), read_basic.rs (buffers)}, used by crates/polars-io/src/ipc.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace plx {
namespace ipc {

struct FormatError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- FlatBuffers, read-only, bounds-checked ---------------------------------------------------------------------------------------
class Flat {
 public:
  Flat(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  size_t size() const { return n_; }
  template <class T> T rd(size_t at) const {
    if (at > n_ || sizeof(T) > n_ - at) throw FormatError("flatbuffer: read past the end");
    T v;
    memcpy(&v, p_ + at, sizeof(T));
    return v;
  }
  size_t root() const { return follow(0); }
  // position an offset field at `at` points to
  size_t follow(size_t at) const {
    size_t t = at + rd<uint32_t>(at);
    if (t >= n_) throw FormatError("flatbuffer: offset past the end");
    return t;
  }
  // position of field `i` of the table at `t`, or 0 when absent
  size_t field(size_t t, int i) const {
    int32_t so = rd<int32_t>(t);
    int64_t vt = (int64_t)t - so;
    if (vt < 0 || (size_t)vt + 4 > n_) throw FormatError("flatbuffer: vtable outside the buffer");
    uint16_t vsize = rd<uint16_t>((size_t)vt);
    size_t slot = 4 + 2 * (size_t)i;
    if (slot + 2 > vsize) return 0;
    uint16_t off = rd<uint16_t>((size_t)vt + slot);
    return off ? t + off : 0;
  }
  template <class T> T scalar(size_t t, int i, T dflt) const {
    size_t f = field(t, i);
    return f ? rd<T>(f) : dflt;
  }
  size_t table(size_t t, int i) const {      // 0 when absent
    size_t f = field(t, i);
    return f ? follow(f) : 0;
  }
  std::string str(size_t t, int i) const {
    size_t f = field(t, i);
    if (!f) return std::string();
    size_t s = follow(f);
    uint32_t len = rd<uint32_t>(s);
    if (len > n_ - s - 4) throw FormatError("flatbuffer: string past the end");
    return std::string((const char*)p_ + s + 4, len);
  }
  // vector field: position of element 0 and the element count (0, 0 when absent)
  void vec(size_t t, int i, size_t elem_size, size_t* first, uint32_t* count) const {
    *first = 0; *count = 0;
    size_t f = field(t, i);
    if (!f) return;
    size_t v = follow(f);
    uint32_t len = rd<uint32_t>(v);
    if ((uint64_t)len * elem_size > n_ - v - 4) throw FormatError("flatbuffer: vector past the end");
    *first = v + 4; *count = len;
  }

 private:
  const uint8_t* p_;
  size_t n_;
};

// ---- Arrow schema -----------------------------------------------------------------------------------------------------------------
enum TypeId {
  TY_NONE = 0, TY_NULL = 1, TY_INT = 2, TY_FLOAT = 3, TY_BINARY = 4, TY_UTF8 = 5, TY_BOOL = 6, TY_DECIMAL = 7, TY_DATE = 8, TY_TIME = 9, TY_TIMESTAMP = 10,
  TY_INTERVAL = 11, TY_LIST = 12, TY_STRUCT = 13, TY_UNION = 14, TY_FIXED_BINARY = 15, TY_FIXED_LIST = 16, TY_MAP = 17, TY_DURATION = 18, TY_LARGE_BINARY = 19,
  TY_LARGE_UTF8 = 20, TY_LARGE_LIST = 21, TY_RUN_END = 22, TY_BINARY_VIEW = 23, TY_UTF8_VIEW = 24, TY_LIST_VIEW = 25, TY_LARGE_LIST_VIEW = 26
};

struct Field {
  std::string name;
  bool nullable = true;
  int type = TY_NONE;
  int bit_width = 0;         // TY_INT (also the index type of a dictionary-encoded field)
  bool is_signed = true;
  int precision = 0;         // TY_FLOAT: 0 half, 1 single, 2 double
  int unit = 0;              // TY_DATE: 0 day, 1 millisecond; TY_TIMESTAMP / TY_DURATION: 0 s, 1 ms, 2 us, 3 ns
  std::string timezone;      // TY_TIMESTAMP: "" = none
  bool has_dictionary = false;
  int64_t dict_id = -1;
  int index_bits = 32;
  bool index_signed = true;
  int n_children = 0;        // > 0: nested (outside the hot path)
  int n_buffers = 0;         // buffers this field (with its children) takes in a record batch, variadic data buffers excluded
  int n_nodes = 1;           // field nodes this field (with its children) takes
  bool variadic = false;     // Utf8View / BinaryView: + variadicBufferCounts[k] data buffers
};

inline const char* type_name(int t) {
  static const char* n[] = {"none", "null", "int", "float", "binary", "utf8", "bool", "decimal", "date", "time", "timestamp", "interval", "list", "struct", "union",
                            "fixed_size_binary", "fixed_size_list", "map", "duration", "large_binary", "large_utf8", "large_list", "run_end_encoded", "binary_view",
                            "utf8_view", "list_view", "large_list_view"};
  return t >= 0 && t <= 26 ? n[t] : "?";
}

namespace detail {
// buffers of ONE array of this type, children excluded (Arrow columnar format, "buffer listing for each layout")
inline int own_buffers(int type) {
  switch (type) {
    case TY_NULL: return 0;
    case TY_BINARY: case TY_UTF8: case TY_LARGE_BINARY: case TY_LARGE_UTF8: return 3;        // validity, offsets, data
    case TY_LIST: case TY_LARGE_LIST: case TY_MAP: return 2;                                  // validity, offsets
    case TY_LIST_VIEW: case TY_LARGE_LIST_VIEW: return 3;                                     // validity, offsets, sizes
    case TY_STRUCT: case TY_FIXED_LIST: return 1;                                             // validity
    case TY_UNION: return 1;                                                                  // type ids (+ offsets when dense: refused anyway)
    case TY_RUN_END: return 0;
    case TY_BINARY_VIEW: case TY_UTF8_VIEW: return 2;                                         // validity, views (+ variadic data buffers)
    default: return 2;                                                                        // validity, values
  }
}

inline Field parse_field(const Flat& fb, size_t t, int depth) {
  if (depth > 32) throw FormatError("schema: nesting too deep");
  Field f;
  f.name = fb.str(t, 0);
  f.nullable = fb.scalar<uint8_t>(t, 1, 0) != 0;
  f.type = fb.scalar<uint8_t>(t, 2, 0);
  size_t ty = fb.table(t, 3);
  if (ty) {
    switch (f.type) {
      case TY_INT: f.bit_width = fb.scalar<int32_t>(ty, 0, 0); f.is_signed = fb.scalar<uint8_t>(ty, 1, 0) != 0; break;
      case TY_FLOAT: f.precision = fb.scalar<int16_t>(ty, 0, 0); break;
      case TY_DATE: f.unit = fb.scalar<int16_t>(ty, 0, 1); break;
      case TY_TIMESTAMP: f.unit = fb.scalar<int16_t>(ty, 0, 0); f.timezone = fb.str(ty, 1); break;
      case TY_DURATION: f.unit = fb.scalar<int16_t>(ty, 0, 1); break;
      default: break;
    }
  }
  size_t de = fb.table(t, 4);
  if (de) {
    f.has_dictionary = true;
    f.dict_id = fb.scalar<int64_t>(de, 0, 0);
    size_t it = fb.table(de, 1);
    if (it) { f.index_bits = fb.scalar<int32_t>(it, 0, 32); f.index_signed = fb.scalar<uint8_t>(it, 1, 0) != 0; }
  }
  // a dictionary-encoded field is laid out as its INDEX type in record batches: validity + values
  f.n_buffers = f.has_dictionary ? 2 : own_buffers(f.type);
  f.variadic = !f.has_dictionary && (f.type == TY_BINARY_VIEW || f.type == TY_UTF8_VIEW);
  size_t first; uint32_t n;
  fb.vec(t, 5, 4, &first, &n);
  f.n_children = (int)n;
  if (!f.has_dictionary)
    for (uint32_t c = 0; c < n; c++) {
      Field ch = parse_field(fb, fb.follow(first + 4 * (size_t)c), depth + 1);
      f.n_buffers += ch.n_buffers; f.n_nodes += ch.n_nodes;
      if (ch.variadic) f.variadic = true;       // nested views: the column is refused anyway
    }
  return f;
}
}  // namespace detail

struct Block {
  int64_t offset = 0;
  int32_t meta_len = 0;
  int64_t body_len = 0;
};

struct Footer {
  int version = 0;
  std::vector<Field> fields;
  std::vector<Block> dictionaries, batches;
};

inline Footer parse_footer(const uint8_t* p, size_t n) {
  Flat fb(p, n);
  Footer ft;
  size_t root = fb.root();
  ft.version = fb.scalar<int16_t>(root, 0, 0);
  size_t schema = fb.table(root, 1);
  if (!schema) throw FormatError("footer without a schema");
  if (fb.scalar<int16_t>(schema, 0, 0) != 0) throw FormatError("big-endian Arrow file");
  size_t first; uint32_t cnt;
  fb.vec(schema, 1, 4, &first, &cnt);
  for (uint32_t i = 0; i < cnt; i++) ft.fields.push_back(detail::parse_field(fb, fb.follow(first + 4 * (size_t)i), 0));
  auto blocks = [&](int field, std::vector<Block>& out) {
    fb.vec(root, field, 24, &first, &cnt);
    for (uint32_t i = 0; i < cnt; i++) {
      Block b;
      b.offset = fb.rd<int64_t>(first + 24 * (size_t)i);
      b.meta_len = fb.rd<int32_t>(first + 24 * (size_t)i + 8);
      b.body_len = fb.rd<int64_t>(first + 24 * (size_t)i + 16);
      if (b.offset < 0 || b.meta_len < 0 || b.body_len < 0) throw FormatError("footer block with negative extents");
      out.push_back(b);
    }
  };
  blocks(2, ft.dictionaries);
  blocks(3, ft.batches);
  return ft;
}

// ---- messages ---------------------------------------------------------------------------------------------------------------------
struct BufferRef { int64_t offset = 0, length = 0; };      // relative to the message body
struct NodeRef { int64_t length = 0, null_count = 0; };

struct BatchMeta {
  int64_t length = 0;
  std::vector<NodeRef> nodes;
  std::vector<BufferRef> buffers;
  std::vector<int64_t> variadic_counts;
  bool compressed = false;
  int codec = 0;              // 0 LZ4_FRAME, 1 ZSTD
  // dictionary batches
  bool is_dictionary = false;
  int64_t dict_id = 0;
  bool is_delta = false;
};

inline void parse_record_batch(const Flat& fb, size_t rb, BatchMeta& m) {
  m.length = fb.scalar<int64_t>(rb, 0, 0);
  if (m.length < 0) throw FormatError("record batch with a negative length");
  size_t first; uint32_t cnt;
  fb.vec(rb, 1, 16, &first, &cnt);
  for (uint32_t i = 0; i < cnt; i++) m.nodes.push_back({fb.rd<int64_t>(first + 16 * (size_t)i), fb.rd<int64_t>(first + 16 * (size_t)i + 8)});
  fb.vec(rb, 2, 16, &first, &cnt);
  for (uint32_t i = 0; i < cnt; i++) m.buffers.push_back({fb.rd<int64_t>(first + 16 * (size_t)i), fb.rd<int64_t>(first + 16 * (size_t)i + 8)});
  size_t comp = fb.table(rb, 3);
  if (comp) { m.compressed = true; m.codec = fb.scalar<int8_t>(comp, 0, 0); }
  fb.vec(rb, 4, 8, &first, &cnt);
  for (uint32_t i = 0; i < cnt; i++) m.variadic_counts.push_back(fb.rd<int64_t>(first + 8 * (size_t)i));
  for (const NodeRef& nd : m.nodes) if (nd.length < 0 || nd.null_count < 0) throw FormatError("field node with negative counts");
  for (const BufferRef& b : m.buffers) if (b.offset < 0 || b.length < 0) throw FormatError("buffer with negative extents");
}

// the Message flatbuffer of a block (without the continuation / size prefix) -> BatchMeta
inline BatchMeta parse_message(const uint8_t* p, size_t n) {
  Flat fb(p, n);
  size_t root = fb.root();
  int header_type = fb.scalar<uint8_t>(root, 1, 0);
  size_t header = fb.table(root, 2);
  if (!header) throw FormatError("message without a header");
  BatchMeta m;
  if (header_type == 3) {
    parse_record_batch(fb, header, m);
  } else if (header_type == 2) {
    m.is_dictionary = true;
    m.dict_id = fb.scalar<int64_t>(header, 0, 0);
    m.is_delta = fb.scalar<uint8_t>(header, 2, 0) != 0;
    size_t rb = fb.table(header, 1);
    if (!rb) throw FormatError("dictionary batch without data");
    parse_record_batch(fb, rb, m);
  } else {
    throw FormatError("unexpected message type " + std::to_string(header_type) + " in a file block");
  }
  return m;
}

}  // namespace ipc
}  // namespace plx
