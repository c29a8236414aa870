// parquet_device.hpp -- the decode side of the device Parquet reader: what each GPU thread / wavefront does to raw column-chunk
// bytes sitting in HBM.  Every function is `PLX_HD` (host + device) and written against plain pointers, so the kernels in
// kernels_parquet.hip are thin launch shells and the very same bodies are run, thread by thread, by the CPU harness of the tests
// (tests/emu/parquet_emu.cpp) against pyarrow-written files.
//
// What is restated (algorithms of the reference's decoder, crates/polars-parquet/src):
//   * RLE / bit-packed hybrid: parquet/encoding/hybrid_rle/mod.rs:80-135 (run header = ULEB128; odd -> (h >> 1) groups of 8 bit-packed
//     values, clamped to the bytes that are there; even -> run of h >> 1 copies of one ceil(bits / 8)-byte value), uleb128.rs:27-62
//   * page layout: parquet/page/mod.rs:372-450 (v1: [u32 length + def levels] + values, all inside the compressed payload; v2: level
//     bytes uncompressed in front of the (optionally compressed) values)
//   * dictionary-encoded values: 1 byte bit width + hybrid runs of indices (arrow/read/deserialize/dictionary_encoded/*.rs), PLAIN fixed
//     width values (arrow/read/deserialize/primitive/plain/required.rs), PLAIN booleans bit-packed LSB first (deserialize/boolean.rs)
//   * optional columns: definition level 1 = value present; values are stored densely (only the present ones), so row r of a page
//     reads dense slot popcount(validity[page_row0 .. r))  (arrow/read/deserialize/utils/mod.rs `decode_page_validity`)
//   * Snappy raw format (snap crate, via parquet/compression.rs:144-230): varint uncompressed length, then literal / copy elements.
//
// GPU shape: nothing here is sequential over a column.  The only serial walks are per page: one thread reads the run headers of one
// level / index stream into a run table (a few hundred to a few thousand headers per 1 MB page), after which every row is decoded
// independently by binary search in that table; Snappy is one wavefront per page (serial tag parse by lane 0 from an LDS window,
// bytes produced by all 64 lanes through pointer jumping: parquet_snappy.hpp).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PLX_HD __host__ __device__ __forceinline__
#else
#define PLX_HD inline
#endif

namespace plx {
namespace pq {

// ---- descriptors (plain data, uploaded as arrays) ----------------------------------------------------------------------------------
enum PageFlags : uint32_t {
  PF_V2 = 1u,           // DATA_PAGE_V2 layout
  PF_COMPRESSED = 2u,   // the values part went through the decompressor: it sits at `dst`
  PF_DICT = 4u,         // values are dictionary indices
  PF_HAS_DEF = 8u,      // the page carries definition levels (optional column)
  PF_RLE_VALUES = 16u,  // boolean values as u32 length + 1-bit hybrid runs (Encoding RLE; what v2 writers emit for booleans)
};

struct PageDesc {
  uint64_t src;         // page payload as stored in the file (after the page header), in HBM
  uint64_t dst;         // decompressed bytes (v1: the whole payload; v2: the values part), 0 when the page is stored uncompressed
  uint64_t row0;        // first output row of the page
  uint32_t comp_size;   // bytes at src
  uint32_t uncomp_size; // bytes of the payload once decompressed
  uint32_t num_values;  // rows of the page (flat columns: values incl. nulls)
  uint32_t flags;
  uint32_t v2_def_len;  // v2: bytes of definition levels at the head of src
  uint32_t dict;        // PF_DICT: index into the DictDesc array
  // ---- filled on the device by page_prepare ----
  uint64_t def_ptr;     // definition-level runs (no length prefix)
  uint64_t val_ptr;     // values: PLAIN bytes, or hybrid runs of dictionary indices (after the bit-width byte)
  uint32_t def_len;
  uint32_t val_len;
  uint32_t bit_width;   // PF_DICT: bits per index
  uint32_t pad0;
  uint64_t valid0;      // filled by page_valid0: valid rows of the column before row0
};

struct DictDesc {
  uint64_t values;      // numeric: PLAIN values of the source width; strings: u32 remap table (chunk dictionary index -> column code)
  uint32_t n;           // entries
  uint32_t pad;
};

// Addresses travel as 64-bit integers inside descriptors that the kernels load from memory; a pointer made from such an integer is a GENERIC pointer to the
// compiler, and every access through it a FLAT instruction (counted by vmcnt AND lgkmcnt: each wait for an LDS operation then also waits for the global loads in
// flight -- the Snappy kernel had 126 of them).  On the device the integer is first made a global-address-space pointer; on the host (the CPU twin the tests run)
// it is a plain cast.
#if defined(__HIP_DEVICE_COMPILE__)
#define PQ_GPTR(T, x) ((T*)(__attribute__((address_space(1))) T*)(x))
#else
#define PQ_GPTR(T, x) ((T*)(x))
#endif

struct DecompJob {      // one Snappy stream
  uint64_t src, dst;
  uint32_t comp_size, uncomp_size;
};

struct RunEntry {       // one run of a hybrid stream
  uint32_t start;       // index of the run's first value within the stream
  uint32_t info;        // bit 31: bit-packed; bits 0..30: byte offset of the run's payload within the stream
};

enum ErrorBits : uint32_t {
  PE_LEVELS = 1u,       // level stream shorter than the page's row count / length prefix past the payload
  PE_RUNS = 2u,         // malformed run header
  PE_VALUES = 4u,       // value bytes missing for a row
  PE_DICT_INDEX = 8u,   // dictionary index out of range
  PE_SNAPPY = 16u,      // malformed Snappy stream
  PE_DEF_LEVEL = 32u,   // definition level > 1 in a flat column
  PE_ZSTD = 64u,        // malformed zstd stream
};

PLX_HD uint32_t load_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
PLX_HD uint64_t load_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
PLX_HD int popc64_hd(uint64_t x) { return __builtin_popcountll(x); }

// ---- page_prepare: one thread per page ----------------------------------------------------------------------------------------------
// Splits the (decompressed) payload into level and value streams (page/mod.rs:372-450).  Returns error bits.
PLX_HD uint32_t page_prepare(PageDesc& p) {
  const uint8_t* body = PQ_GPTR(const uint8_t, p.flags & PF_COMPRESSED ? p.dst : p.src);
  uint32_t err = 0;
  if (p.flags & PF_V2) {
    // levels are never compressed and stay at src; the values part is what was (or was not) decompressed
    uint32_t lv = p.v2_def_len;
    p.def_ptr = p.src; p.def_len = lv;
    if (p.flags & PF_COMPRESSED) { p.val_ptr = p.dst; p.val_len = p.uncomp_size >= lv ? p.uncomp_size - lv : 0; }
    else { p.val_ptr = p.src + lv; p.val_len = p.comp_size >= lv ? p.comp_size - lv : 0; }
    if (lv > p.comp_size) err |= PE_LEVELS;
  } else {
    uint32_t size = p.uncomp_size, off = 0;
    p.def_ptr = 0; p.def_len = 0;
    if (p.flags & PF_HAS_DEF) {
      if (size < 4) { err |= PE_LEVELS; }
      else {
        uint32_t n = load_u32(body);
        if (n > size - 4) { err |= PE_LEVELS; n = size - 4; }
        p.def_ptr = (uint64_t)(body + 4); p.def_len = n; off = 4 + n;
      }
    }
    p.val_ptr = (uint64_t)(body + off); p.val_len = size - off;
  }
  p.bit_width = 0;
  if ((p.flags & PF_DICT) && !err) {
    if (p.val_len < 1) {
      if (p.num_values) err |= PE_VALUES;
    } else {
      p.bit_width = *PQ_GPTR(const uint8_t, p.val_ptr);
      p.val_ptr += 1; p.val_len -= 1;
      if (p.bit_width > 32) { err |= PE_RUNS; p.bit_width = 0; }
    }
  }
  if ((p.flags & PF_RLE_VALUES) && !err) {
    if (p.val_len < 4) {
      if (p.num_values) err |= PE_VALUES;
      p.val_len = 0;
    } else {
      uint32_t n = load_u32(PQ_GPTR(const uint8_t, p.val_ptr));
      if (n > p.val_len - 4) { err |= PE_VALUES; n = p.val_len - 4; }
      p.val_ptr += 4; p.val_len = n; p.bit_width = 1;
    }
  }
  p.valid0 = 0;
  return err;
}

// ---- run tables: one thread per stream ----------------------------------------------------------------------------------------------
// Walks the run headers of a hybrid stream of at most `max_values` values.  emit(k, start, payload_offset, bitpacked) is called per
// run; *total receives the number of values the runs cover (clipped to max_values).  Returns the number of runs, or 0xffffffff when a
// header is malformed.  A stream of bit width 0 has no bytes: every value is 0.
template <class Emit> PLX_HD uint32_t walk_runs(const uint8_t* s, uint32_t len, uint32_t bits, uint32_t max_values, uint32_t* total, Emit&& emit) {
  *total = 0;
  if (bits == 0 || max_values == 0) { *total = max_values; return 0; }
  const uint32_t vbytes = (bits + 7) >> 3;
  uint32_t pos = 0, out = 0, k = 0;
  while (out < max_values && pos < len) {
    uint32_t h = 0;
    for (uint32_t shift = 0;; shift += 7) {
      if (pos >= len || shift > 28) return 0xffffffffu;
      uint32_t b = s[pos++];
      h |= (b & 0x7f) << shift;
      if (!(b & 0x80)) break;
    }
    if (h & 1) {
      uint64_t bytes = (uint64_t)(h >> 1) * bits;
      if (bytes > len - pos) bytes = len - pos;                 // a writer may stop the last group early (hybrid_rle/mod.rs:101-102)
      uint64_t count = bytes * 8 / bits;
      if (count > max_values - out) count = max_values - out;
      if (count) emit(k++, out, pos, true);
      pos += (uint32_t)bytes; out += (uint32_t)count;
    } else {
      uint32_t count = h >> 1;
      if (count > max_values - out) count = max_values - out;
      if (vbytes > len - pos) return 0xffffffffu;
      if (count) emit(k++, out, pos, false);
      pos += vbytes; out += count;
    }
  }
  *total = out;
  return k;
}

// stream s of page p: 0 = definition levels (1 bit, exactly num_values of them), 1 = dictionary indices / RLE booleans (bit_width
// bits; one per non-null row, which only the levels know: the walk stops at the end of the stream)
PLX_HD bool stream_params(const PageDesc& p, int s, const uint8_t** ptr, uint32_t* len, uint32_t* bits) {
  if (s == 0) {
    if (!(p.flags & PF_HAS_DEF)) return false;
    *ptr = PQ_GPTR(const uint8_t, p.def_ptr); *len = p.def_len; *bits = 1;
    return true;
  }
  if (!(p.flags & (PF_DICT | PF_RLE_VALUES))) return false;
  *ptr = PQ_GPTR(const uint8_t, p.val_ptr); *len = p.val_len; *bits = p.bit_width;
  return true;
}

// table entries stream s of page p needs: its runs + one sentinel {start = values covered}; 0 when the page has no such stream
PLX_HD uint32_t stream_entries(const PageDesc& p, int s, uint32_t* err) {
  const uint8_t* ptr; uint32_t len, bits, total;
  if (!stream_params(p, s, &ptr, &len, &bits)) return 0;
  uint32_t k = walk_runs(ptr, len, bits, p.num_values, &total, [](uint32_t, uint32_t, uint32_t, bool) {});
  if (k == 0xffffffffu) { *err |= PE_RUNS; return 1; }
  if (s == 0 && total != p.num_values) { *err |= PE_LEVELS; return 1; }
  return k + 1;
}
PLX_HD void stream_fill(const PageDesc& p, int s, RunEntry* out, uint32_t n_entries) {
  const uint8_t* ptr; uint32_t len, bits, total = 0;
  if (!n_entries || !stream_params(p, s, &ptr, &len, &bits)) return;
  uint32_t k = 0;
  if (n_entries > 1)
    k = walk_runs(ptr, len, bits, p.num_values, &total, [&](uint32_t i, uint32_t start, uint32_t off, bool packed) {
      if (i + 1 < n_entries) { out[i].start = start; out[i].info = off | (packed ? 0x80000000u : 0u); }
    });
  (void)k;
  out[n_entries - 1].start = n_entries > 1 ? total : 0;
  out[n_entries - 1].info = 0;
}

// ---- hybrid stream random access -----------------------------------------------------------------------------------------------------
// value i of a stream described by runs[0 .. n_runs) (+ bits, base pointer).  Streams of bit width 0 have no runs: every value is 0.
PLX_HD uint32_t find_run(const RunEntry* runs, uint32_t n_runs, uint32_t i) {
  uint32_t lo = 0, hi = n_runs;                 // last run with start <= i
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (runs[mid].start <= i) lo = mid; else hi = mid;
  }
  return lo;
}
PLX_HD uint32_t run_value(const uint8_t* s, RunEntry r, uint32_t bits, uint32_t i) {
  const uint8_t* payload = s + (r.info & 0x7fffffffu);
  const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
  if (r.info >> 31) {
    uint64_t bit = (uint64_t)(i - r.start) * bits;
    // <= 32 bits starting at bit & 7 of a byte: 5 bytes at most.  Buffers are padded by >= 8 bytes, so an 8-byte load is in bounds.
    uint64_t w = load_u64(payload + (bit >> 3));
    return (uint32_t)(w >> (bit & 7)) & mask;
  }
  uint32_t v = 0;
  for (uint32_t b = 0; b < ((bits + 7) >> 3); b++) v |= (uint32_t)payload[b] << (8 * b);
  return v & mask;
}

// ---- validity: one thread per 64 output rows --------------------------------------------------------------------------------------
// page containing row r: last page with row0 <= r and num_values > 0 ... pages are in row order; empty pages share a row0 with the
// next page, so "last page with row0 <= r" skips them.
PLX_HD uint32_t find_page(const PageDesc* pages, uint32_t n_pages, uint64_t r) {
  uint32_t lo = 0, hi = n_pages;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (pages[mid].row0 <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// bits [bit, bit + n) (n <= 64) of a little-endian bit stream; the 9 bytes touched must be readable (buffers are padded)
PLX_HD uint64_t get_bits(const uint8_t* s, uint64_t bit, uint32_t n) {
  const uint8_t* p = s + (bit >> 3);
  uint32_t sh = (uint32_t)(bit & 7);
  uint64_t lo = load_u64(p);
  uint64_t v = lo >> sh;
  if (sh && sh + n > 64) v |= (uint64_t)p[8] << (64 - sh);
  return n >= 64 ? v : v & ((1ull << n) - 1ull);
}

// Validity word w of the column (rows [64 w, 64 w + 64) clipped to n_rows).  run_off[2 * page] .. run_off[2 * page + 1] index the page's
// definition-level table (runs + sentinel), run_off[2 * page + 1] .. run_off[2 * page + 2] its dictionary-index table.  Pages without
// levels are all valid.  *err receives error bits.
PLX_HD uint64_t validity_word(const PageDesc* pages, uint32_t n_pages, const RunEntry* runs, const uint64_t* run_off, uint64_t n_rows, uint64_t w, uint32_t* err) {
  uint64_t r = w * 64, end = r + 64 < n_rows ? r + 64 : n_rows;
  uint64_t word = 0;
  if (r >= end) return 0;
  uint32_t pg = find_page(pages, n_pages, r);
  while (r < end) {
    const PageDesc& p = pages[pg];
    uint64_t page_end = p.row0 + p.num_values;
    if (r >= page_end) {                       // empty page or end of this page
      if (++pg >= n_pages) { *err |= PE_LEVELS; break; }
      continue;
    }
    uint32_t take_page = (uint32_t)((page_end < end ? page_end : end) - r);
    uint32_t sh = (uint32_t)(r - w * 64);
    if (!(p.flags & PF_HAS_DEF)) {
      word |= (take_page >= 64 ? ~0ull : ((1ull << take_page) - 1ull)) << sh;
      r += take_page;
      continue;
    }
    const RunEntry* pr = runs + run_off[2 * pg];
    uint32_t n_ent = (uint32_t)(run_off[2 * pg + 1] - run_off[2 * pg]);       // runs + sentinel
    if (n_ent < 2 || pr[n_ent - 1].start != p.num_values) { *err |= PE_LEVELS; r += take_page; continue; }
    uint32_t n_runs = n_ent - 1;
    const uint8_t* s = PQ_GPTR(const uint8_t, p.def_ptr);
    uint32_t i = (uint32_t)(r - p.row0), left = take_page;
    uint32_t k = find_run(pr, n_runs, i);
    while (left) {
      uint32_t run_end = pr[k + 1].start;                                    // the sentinel closes the last run
      uint32_t take = run_end - i < left ? run_end - i : left;
      if (run_end <= i) { *err |= PE_LEVELS; break; }                       // corrupt table: never spin
      uint64_t bits;
      if (pr[k].info >> 31) bits = get_bits(s + (pr[k].info & 0x7fffffffu), i - pr[k].start, take);
      else {
        uint32_t level = s[pr[k].info & 0x7fffffffu];
        if (level > 1) *err |= PE_DEF_LEVEL;
        bits = (level & 1) ? (take >= 64 ? ~0ull : ((1ull << take) - 1ull)) : 0ull;
      }
      word |= bits << (uint32_t)(p.row0 + i - w * 64);
      i += take; left -= take;
      if (left) k++;
    }
    r += take_page;
  }
  return word;
}

// valid rows of the column before row r, from the per-word exclusive prefix
PLX_HD uint64_t valid_before(const uint64_t* validity, const uint64_t* word_prefix, uint64_t r) {
  uint64_t w = r >> 6;
  uint32_t b = (uint32_t)(r & 63);
  return word_prefix[w] + (b ? (uint64_t)popc64_hd(validity[w] & ((1ull << b) - 1ull)) : 0);
}

// ---- values: one thread per row ----------------------------------------------------------------------------------------------------
// The source value of row r as raw little-endian bits (<= 8 bytes), already mapped through the dictionary.  `validity` == nullptr:
// the column has no nulls (level streams are skipped, dense slot = row within page).  src_width: bytes per PLAIN value (4 / 8), or 0
// for bit-packed booleans.  Returns false for a null row.
struct ColumnDecode {
  const PageDesc* pages;
  uint32_t n_pages;
  const DictDesc* dicts;
  const RunEntry* runs;        // index-stream runs
  const uint64_t* run_off;     // [2 * page + 1] .. [2 * page + 2]: the page's index runs (entry 2 * page: level runs)
  const uint64_t* validity;
  const uint64_t* word_prefix;
  uint64_t n_rows;
  uint32_t src_width;          // 0 (boolean), 4, 8
  uint32_t dict_width;         // bytes per dictionary entry: the source width, or 4 for string remap tables
};

PLX_HD bool decode_row(const ColumnDecode& c, uint32_t pg, uint64_t r, uint64_t* out_bits, uint32_t* err) {
  const PageDesc& p = c.pages[pg];
  uint64_t dense;
  if (c.validity) {
    if (!((c.validity[r >> 6] >> (r & 63)) & 1)) return false;
    dense = valid_before(c.validity, c.word_prefix, r) - p.valid0;
  } else {
    dense = r - p.row0;
  }
  if (p.flags & (PF_DICT | PF_RLE_VALUES)) {
    uint32_t idx = 0;
    if (p.bit_width) {
      const RunEntry* pr = c.runs + c.run_off[2 * pg + 1];
      uint32_t n_ent = (uint32_t)(c.run_off[2 * pg + 2] - c.run_off[2 * pg + 1]);   // runs + sentinel {start = indices in the stream}
      if (n_ent < 2 || dense >= pr[n_ent - 1].start) { *err |= PE_VALUES; *out_bits = 0; return true; }
      uint32_t k = find_run(pr, n_ent - 1, (uint32_t)dense);
      idx = run_value(PQ_GPTR(const uint8_t, p.val_ptr), pr[k], p.bit_width, (uint32_t)dense);
    }
    if (p.flags & PF_RLE_VALUES) { *out_bits = idx; return true; }
    const DictDesc& d = c.dicts[p.dict];
    if (idx >= d.n) { *err |= PE_DICT_INDEX; *out_bits = 0; return true; }
    const uint8_t* e = PQ_GPTR(const uint8_t, d.values) + (uint64_t)idx * c.dict_width;
    *out_bits = c.dict_width == 8 ? load_u64(e) : c.dict_width == 4 ? (uint64_t)load_u32(e) : (uint64_t)*e;
    return true;
  }
  const uint8_t* v = PQ_GPTR(const uint8_t, p.val_ptr);
  if (c.src_width == 0) {
    if ((dense >> 3) >= p.val_len) { *err |= PE_VALUES; *out_bits = 0; return true; }
    *out_bits = (v[dense >> 3] >> (dense & 7)) & 1;
    return true;
  }
  if ((dense + 1) * c.src_width > p.val_len) { *err |= PE_VALUES; *out_bits = 0; return true; }
  *out_bits = c.src_width == 8 ? load_u64(v + dense * 8) : (uint64_t)load_u32(v + dense * 4);
  return true;
}

// store `bits` (the source value) as the output dtype of width out_width: integer narrowing keeps the low bytes (two's complement),
// floats keep their width
PLX_HD void store_value(void* out, uint64_t r, uint32_t out_width, uint64_t bits) {
  switch (out_width) {
    case 1: ((uint8_t*)out)[r] = (uint8_t)bits; break;
    case 2: ((uint16_t*)out)[r] = (uint16_t)bits; break;
    case 4: ((uint32_t*)out)[r] = (uint32_t)bits; break;
    default: ((uint64_t*)out)[r] = bits; break;
  }
}

// rows [r0, r1) of a column into `out` (thread-contiguous block; the page is looked up once and advanced)
PLX_HD void decode_rows(const ColumnDecode& c, void* out, uint32_t out_width, uint64_t r0, uint64_t r1, uint32_t* err) {
  if (r0 >= r1) return;
  uint32_t pg = find_page(c.pages, c.n_pages, r0);
  for (uint64_t r = r0; r < r1; r++) {
    while (r >= c.pages[pg].row0 + c.pages[pg].num_values) {
      if (pg + 1 >= c.n_pages) { *err |= PE_VALUES; return; }
      pg++;
    }
    uint64_t bits = 0;
    if (!decode_row(c, pg, r, &bits, err)) bits = 0;
    store_value(out, r, out_width, bits);
  }
}

// Boolean output: bitmap word w of the column
PLX_HD uint64_t decode_bool_word(const ColumnDecode& c, uint64_t w, uint32_t* err) {
  uint64_t r0 = w * 64, r1 = r0 + 64 < c.n_rows ? r0 + 64 : c.n_rows;
  if (r0 >= r1) return 0;
  uint64_t word = 0;
  uint32_t pg = find_page(c.pages, c.n_pages, r0);
  for (uint64_t r = r0; r < r1; r++) {
    while (r >= c.pages[pg].row0 + c.pages[pg].num_values) {
      if (pg + 1 >= c.n_pages) { *err |= PE_VALUES; return word; }
      pg++;
    }
    uint64_t bits = 0;
    if (decode_row(c, pg, r, &bits, err)) word |= (bits & 1) << (r - r0);
  }
  return word;
}

}  // namespace pq
}  // namespace plx

#include "parquet_snappy.hpp"   // Snappy: one wavefront per stream (stage / parse / point / jump / gather rounds)
