// host_codecs.hpp -- page / buffer decompressors that run on HOST threads: Zstandard (RFC 8878), LZ4 raw blocks and LZ4 frames, DEFLATE in
// gzip / zlib wrappers (RFC 1950-1952), written from the format specifications, bounds-checked, no third-party code.  No HIP.
//
// Why they exist: Snappy pages are decompressed on the device (parquet_snappy.hpp).  Zstandard is what Polars itself writes by default
// (crates/polars-parquet/src/parquet/compression.rs:144-230 dispatches to the zstd crate), and its entropy stages (FSE-coded sequences
// on a backward bit stream, 4-way Huffman literals) are a different kernel family that is not written yet.  Until it is, such pages are
// decompressed by a pool of host threads, one page each, into the page-locked staging buffer, and from there everything is the
// uncompressed device path (levels, dictionary indices, values, nulls: kernels_parquet.hip) -- so a file written by Polars is still
// read without pyarrow and decoded on the device, only its decompression is not.
//
// zstd_decompress follows RFC 8878 section by section: frames (3.1.1), blocks (3.1.1.2), literals section with raw / RLE / Huffman
// literals in 1 or 4 streams (3.1.1.3.1, 4.2), Huffman weights direct or FSE-compressed (4.2.1), sequences section with predefined /
// RLE / FSE / repeat tables (3.1.1.3.2, 4.1), sequence execution with the three repeat offsets (3.1.1.5).  Dictionaries are refused.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace plx {
namespace codec {

struct CodecError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void copy16(uint8_t* d, const uint8_t* s) { uint64_t a, b; memcpy(&a, s, 8); memcpy(&b, s + 8, 8); memcpy(d, &a, 8); memcpy(d + 8, &b, 8); }

// match copy of an LZ77 decoder: n bytes from off bytes back, byte-sequential semantics (a match may overlap itself).  With
// `slack` >= 16 writable bytes behind the match it goes word by word (what is written past the match is overwritten by what follows).
inline void match_copy(uint8_t* d, size_t off, size_t n, size_t slack) {
  const uint8_t* m = d - off;
  if (slack >= 16 && off >= 8) {
    if (off >= 16) for (size_t k = 0; k < n; k += 16) copy16(d + k, m + k);
    else for (size_t k = 0; k < n; k += 8) { uint64_t w; memcpy(&w, m + k, 8); memcpy(d + k, &w, 8); }
  } else if (off >= n) memcpy(d, m, n);
  else if (off == 1) memset(d, m[0], n);
  else for (size_t k = 0; k < n; k++) d[k] = m[k];
}

// ---- LZ4 ------------------------------------------------------------------------------------------------------------------------------
// One LZ4 block: sequences of token (hi nibble literal length, lo nibble match length - 4; 15 = more length bytes of 255 follow),
// literals, 2-byte little-endian offset, [match length bytes]; the last sequence ends after its literals.  Appends to out at *op;
// matches may reach back into what out already holds (dependent blocks of a frame).
inline void lz4_block(const uint8_t* in, size_t n, uint8_t* out, size_t* op_io, size_t out_cap) {
  size_t ip = 0, op = *op_io;
  while (ip < n) {
    const uint8_t token = in[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      uint8_t b;
      do {
        if (ip >= n) throw CodecError("lz4: truncated literal length");
        b = in[ip++];
        lit += b;
      } while (b == 255);
    }
    if (lit > n - ip || lit > out_cap - op) throw CodecError("lz4: literals past the end");
    if (lit <= 16 && n - ip >= 16 && out_cap - op >= 16) copy16(out + op, in + ip);
    else memcpy(out + op, in + ip, lit);
    ip += lit; op += lit;
    if (ip >= n) break;                       // last sequence: literals only
    if (n - ip < 2) throw CodecError("lz4: truncated offset");
    const size_t off = in[ip] | ((size_t)in[ip + 1] << 8);
    ip += 2;
    size_t ml = (token & 15);
    if (ml == 15) {
      uint8_t b;
      do {
        if (ip >= n) throw CodecError("lz4: truncated match length");
        b = in[ip++];
        ml += b;
      } while (b == 255);
    }
    ml += 4;
    if (off == 0 || off > op || ml > out_cap - op) throw CodecError("lz4: bad match");
    match_copy(out + op, off, ml, out_cap - op - ml);
    op += ml;
  }
  *op_io = op;
}

// Parquet codec LZ4_RAW: the page is one block
inline void lz4_raw_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len) {
  size_t op = 0;
  lz4_block(in, n, out, &op, out_len);
  if (op != out_len) throw CodecError("lz4: block decodes to a different length than the page header says");
}

// LZ4 frame format (Arrow IPC body compression LZ4_FRAME): magic, FLG / BD [content size] [dictionary id] HC, blocks {u32 size, high bit =
// stored; 0 = end mark} [block checksum], [content checksum]; skippable frames are skipped.  Checksums are not verified.
inline void lz4_frame_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len) {
  size_t ip = 0, op = 0;
  auto rd32 = [&](size_t at) { uint32_t v; if (n - at < 4 || at > n) throw CodecError("lz4: truncated frame"); memcpy(&v, in + at, 4); return v; };
  while (ip < n) {
    const uint32_t magic = rd32(ip);
    if ((magic & 0xfffffff0u) == 0x184d2a50u) {
      const uint32_t len = rd32(ip + 4);
      if (len > n - ip - 8) throw CodecError("lz4: skippable frame past the end");
      ip += 8 + len;
      continue;
    }
    if (magic != 0x184d2204u) throw CodecError("lz4: not an LZ4 frame");
    ip += 4;
    if (n - ip < 3) throw CodecError("lz4: truncated frame descriptor");
    const uint8_t flg = in[ip];
    if ((flg >> 6) != 1) throw CodecError("lz4: unknown frame version");
    const bool block_checksum = flg & 0x10, content_size = flg & 0x08, content_checksum = flg & 0x04, dict_id = flg & 0x01;
    if (dict_id) throw CodecError("lz4: frame needs a dictionary");
    ip += 2 + (content_size ? 8 : 0) + 1;                 // FLG, BD, [content size], HC
    if (ip > n) throw CodecError("lz4: truncated frame descriptor");
    const size_t frame_start = op;
    for (;;) {
      const uint32_t bs = rd32(ip);
      ip += 4;
      if (bs == 0) break;
      const size_t len = bs & 0x7fffffffu;
      if (len > n - ip) throw CodecError("lz4: block past the end of the frame");
      if (bs & 0x80000000u) {
        if (len > out_len - op) throw CodecError("lz4: stored block past the output");
        memcpy(out + op, in + ip, len);
        op += len;
      } else {
        size_t rel = op - frame_start;                    // matches reach back to the start of the frame at most
        lz4_block(in + ip, len, out + frame_start, &rel, out_len - frame_start);
        op = frame_start + rel;
      }
      ip += len + (block_checksum ? 4 : 0);
      if (ip > n) throw CodecError("lz4: truncated block checksum");
    }
    if (content_checksum) { if (n - ip < 4) throw CodecError("lz4: truncated content checksum"); ip += 4; }
  }
  if (op != out_len) throw CodecError("lz4: frame decodes to a different length than its buffer header says");
}

// ---- DEFLATE / gzip (Parquet codec GZIP) ---------------------------------------------------------------------------------------------------
// RFC 1951 blocks (stored, fixed Huffman, dynamic Huffman) inside RFC 1952 members (header with optional extra / name / comment / header
// CRC, trailer CRC32 + ISIZE; the CRC is not verified) or a zlib wrapper (RFC 1950).  Canonical Huffman codes are decoded length by length
// from the per-length symbol counts -- slow next to table-driven decoders, plenty for pages that then cross PCIe.
namespace inflate_detail {

struct Bits {
  const uint8_t* p; size_t n; size_t pos = 0; uint64_t hold = 0; int cnt = 0;
  // at least 56 valid bits in one go while 8 input bytes remain (cnt stays a multiple of 8 short of 64: whole bytes only)
  void refill() {
    if (cnt < 48 && n - pos >= 8) {
      uint64_t w;
      memcpy(&w, p + pos, 8);
      hold |= w << cnt;
      const int take = (63 - cnt) >> 3;      // whole bytes that fit
      pos += (size_t)take; cnt += take * 8;
    }
  }
  uint32_t get(int need) {
    refill();
    while (cnt < need) {
      if (pos >= n) throw CodecError("deflate: stream ends inside a block");
      hold |= (uint64_t)p[pos++] << cnt;
      cnt += 8;
    }
    const uint32_t v = need ? (uint32_t)(hold & (((uint64_t)1 << need) - 1)) : 0;
    hold >>= need; cnt -= need;
    return v;
  }
  // the next `need` (<= 16) bits without consuming them; past the end of the stream they read as zero (a code that needs them fails in drop)
  uint32_t peek(int need) {
    refill();
    while (cnt < need && pos < n) { hold |= (uint64_t)p[pos++] << cnt; cnt += 8; }
    return (uint32_t)(hold & (((uint64_t)1 << need) - 1));
  }
  void drop(int used) {
    if (used > cnt) throw CodecError("deflate: stream ends inside a block");
    hold >>= used; cnt -= used;
  }
  void align() { const int whole = cnt / 8; pos -= (size_t)whole; hold = 0; cnt = 0; }      // back to the byte boundary: bytes pulled in ahead are handed back
};

constexpr int kFastBits = 10;
struct Huff {
  uint16_t count[16]; uint16_t symbol[288];
  uint16_t fast[1 << kFastBits];      // code (bit-reversed: DEFLATE codes arrive most significant bit first) -> length << 9 | symbol; 0: longer than kFastBits
};

inline void build(Huff& h, const uint8_t* lens, int n) {
  memset(h.count, 0, sizeof h.count);
  for (int i = 0; i < n; i++) h.count[lens[i]]++;
  // over-subscribed sets are malformed; incomplete ones are allowed only in the degenerate single-code case (checked by the caller's decode)
  int left = 1;
  for (int len = 1; len <= 15; len++) { left = (left << 1) - h.count[len]; if (left < 0) throw CodecError("deflate: over-subscribed Huffman code"); }
  uint16_t offs[16];
  offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + h.count[len]);
  for (int i = 0; i < n; i++) if (lens[i]) h.symbol[offs[lens[i]]++] = (uint16_t)i;
  // first-level table: every code of at most kFastBits bits, replicated over the bits that follow it
  memset(h.fast, 0, sizeof h.fast);
  int code = 0, index = 0;
  for (int len = 1; len <= kFastBits; len++) {
    for (int k = 0; k < h.count[len]; k++, code++, index++) {
      uint32_t rev = 0;
      for (int b = 0; b < len; b++) rev |= ((uint32_t)(code >> b) & 1) << (len - 1 - b);
      for (uint32_t fill = rev; fill < (1u << kFastBits); fill += 1u << len) h.fast[fill] = (uint16_t)((len << 9) | h.symbol[index]);
    }
    code <<= 1;
  }
}

inline int decode(Bits& b, const Huff& h) {
  const uint16_t f = h.fast[b.peek(kFastBits)];
  if (f) { b.drop(f >> 9); return f & 511; }
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)b.get(1);
    const int count = h.count[len];
    if (code - count < first) return h.symbol[index + (code - first)];
    index += count; first += count;
    first <<= 1; code <<= 1;
  }
  throw CodecError("deflate: invalid Huffman code");
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenBits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistBits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline void codes(Bits& b, const Huff& lit, const Huff& dist, uint8_t* out, size_t* op_io, size_t cap) {
  size_t op = *op_io;
  for (;;) {
    const int sym = decode(b, lit);
    if (sym < 256) {
      if (op >= cap) throw CodecError("deflate: output larger than the page header says");
      out[op++] = (uint8_t)sym;
    } else if (sym == 256) {
      break;
    } else {
      if (sym > 285) throw CodecError("deflate: invalid length code");
      const size_t len = kLenBase[sym - 257] + b.get(kLenBits[sym - 257]);
      const int ds = decode(b, dist);
      if (ds > 29) throw CodecError("deflate: invalid distance code");
      const size_t d = kDistBase[ds] + b.get(kDistBits[ds]);
      if (d > op || len > cap - op) throw CodecError("deflate: match outside the output");
      match_copy(out + op, d, len, cap - op - len);
      op += len;
    }
  }
  *op_io = op;
}

// one DEFLATE stream starting at in[*ip_io]; appends to out
inline void inflate(const uint8_t* in, size_t n, size_t* ip_io, uint8_t* out, size_t* op_io, size_t cap) {
  Bits b{in + *ip_io, n - *ip_io};
  size_t op = *op_io;
  const size_t start = op;          // matches may not reach before this member's own output
  for (;;) {
    const uint32_t last = b.get(1), type = b.get(2);
    if (type == 0) {
      b.align();
      if (b.n - b.pos < 4) throw CodecError("deflate: truncated stored block");
      const uint32_t len = b.p[b.pos] | ((uint32_t)b.p[b.pos + 1] << 8), nlen = b.p[b.pos + 2] | ((uint32_t)b.p[b.pos + 3] << 8);
      b.pos += 4;
      if ((len ^ 0xffff) != nlen) throw CodecError("deflate: stored block length check failed");
      if (len > b.n - b.pos || len > cap - op) throw CodecError("deflate: stored block past the end");
      memcpy(out + op, b.p + b.pos, len);
      b.pos += len; op += len;
    } else if (type == 1) {
      uint8_t lens[288];
      for (int i = 0; i < 144; i++) lens[i] = 8;
      for (int i = 144; i < 256; i++) lens[i] = 9;
      for (int i = 256; i < 280; i++) lens[i] = 7;
      for (int i = 280; i < 288; i++) lens[i] = 8;
      Huff lit, dist;
      build(lit, lens, 288);
      uint8_t dl[30];
      memset(dl, 5, 30);
      build(dist, dl, 30);
      size_t rel = op - start;
      codes(b, lit, dist, out + start, &rel, cap - start);
      op = start + rel;
    } else if (type == 2) {
      const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
      if (nlen > 286 || ndist > 30) throw CodecError("deflate: too many codes in a dynamic block");
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t lens[320];
      memset(lens, 0, sizeof lens);
      for (int i = 0; i < ncode; i++) lens[order[i]] = (uint8_t)b.get(3);
      Huff lencode;
      build(lencode, lens, 19);
      uint8_t ll[320];
      int idx = 0;
      while (idx < nlen + ndist) {
        const int sym = decode(b, lencode);
        if (sym < 16) ll[idx++] = (uint8_t)sym;
        else {
          int rep; uint8_t val = 0;
          if (sym == 16) { if (idx == 0) throw CodecError("deflate: repeat without a previous length"); val = ll[idx - 1]; rep = 3 + (int)b.get(2); }
          else if (sym == 17) rep = 3 + (int)b.get(3);
          else rep = 11 + (int)b.get(7);
          if (idx + rep > nlen + ndist) throw CodecError("deflate: code lengths overrun");
          while (rep--) ll[idx++] = val;
        }
      }
      if (ll[256] == 0) throw CodecError("deflate: no end-of-block code");
      Huff lit, dist;
      build(lit, ll, nlen);
      build(dist, ll + nlen, ndist);
      size_t rel = op - start;
      codes(b, lit, dist, out + start, &rel, cap - start);
      op = start + rel;
    } else {
      throw CodecError("deflate: reserved block type");
    }
    if (last) break;
  }
  *ip_io += b.pos - (size_t)(b.cnt / 8);      // whole bytes pulled in ahead but not used belong to what follows
  *op_io = op;
}

}  // namespace inflate_detail

// every gzip member (or one zlib stream, or a bare DEFLATE stream when neither header is there) of [in, in + n) into out[0, out_len)
inline void gzip_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len) {
  size_t ip = 0, op = 0;
  if (n >= 2 && in[0] == 0x1f && in[1] == 0x8b) {
    while (ip < n) {
      if (n - ip < 10 || in[ip] != 0x1f || in[ip + 1] != 0x8b) throw CodecError("gzip: bad member header");
      if (in[ip + 2] != 8) throw CodecError("gzip: compression method is not deflate");
      const uint8_t flg = in[ip + 3];
      ip += 10;
      if (flg & 4) { if (n - ip < 2) throw CodecError("gzip: truncated extra field"); const size_t xl = in[ip] | ((size_t)in[ip + 1] << 8); ip += 2; if (xl > n - ip) throw CodecError("gzip: extra field past the end"); ip += xl; }
      for (int f = 3; f <= 4; f++)                        // FNAME (bit 3), FCOMMENT (bit 4): zero-terminated
        if (flg & (1 << f)) { while (ip < n && in[ip]) ip++; if (ip >= n) throw CodecError("gzip: unterminated header string"); ip++; }
      if (flg & 2) { if (n - ip < 2) throw CodecError("gzip: truncated header crc"); ip += 2; }
      inflate_detail::inflate(in, n, &ip, out, &op, out_len);
      if (n - ip < 8) throw CodecError("gzip: truncated trailer");
      ip += 8;                                            // CRC32, ISIZE: the page header is the authority on the size
    }
  } else if (n >= 2 && (in[0] & 0x0f) == 8 && ((in[0] << 8 | in[1]) % 31) == 0) {
    if (in[1] & 0x20) throw CodecError("zlib: stream needs a preset dictionary");
    ip = 2;
    inflate_detail::inflate(in, n, &ip, out, &op, out_len);
  } else {
    inflate_detail::inflate(in, n, &ip, out, &op, out_len);
  }
  if (op != out_len) throw CodecError("gzip: stream decodes to a different length than the page header says");
}

// ---- Zstandard ------------------------------------------------------------------------------------------------------------------------
namespace zstd_detail {

inline int highest_bit(uint64_t v) { return v ? 63 - __builtin_clzll(v) : -1; }

// forward little-endian bit reader (FSE table descriptions)
struct FwdBits {
  const uint8_t* p; size_t n; size_t bit = 0;
  uint32_t read(int nb) {
    const size_t first = bit >> 3;
    if (nb <= 24 && first + 4 <= n) {              // the common case: one unaligned 4-byte load covers shift (<= 7) + nb bits
      uint32_t w;
      memcpy(&w, p + first, 4);
      const uint32_t v = (w >> (bit & 7)) & (((uint32_t)1 << nb) - 1);
      bit += nb;
      return v;
    }
    uint32_t v = 0;
    for (int i = 0; i < nb; i++) {
      size_t byte = (bit + i) >> 3;
      if (byte >= n) throw CodecError("zstd: table description runs past its section");
      v |= (uint32_t)((p[byte] >> ((bit + i) & 7)) & 1) << i;
    }
    bit += nb;
    return v;
  }
  void rewind(int nb) { bit -= nb; }
  size_t bytes_used() const { return (bit + 7) >> 3; }
};

// backward bit stream (4.1 / 4.2.2): starts at the highest set bit of the last byte; bits before the start read as zero
struct BackBits {
  const uint8_t* p; size_t n; int64_t off;        // bit offset of the next bit to hand out (counting down)
  BackBits(const uint8_t* src, size_t len) : p(src), n(len) {
    if (len == 0 || src[len - 1] == 0) throw CodecError("zstd: backward bit stream without its end mark");
    off = (int64_t)len * 8 - (8 - highest_bit(src[len - 1]));      // drop the padding and the mark itself
  }
  uint64_t read(int nb) {
    if (nb == 0) return 0;
    off -= nb;
    if (off >= 0) {
      const size_t byte = (size_t)(off >> 3);
      if (byte + 8 <= n && nb <= 56) {                  // the common case: one unaligned 8-byte load covers shift (<= 7) + nb bits
        uint64_t w;
        memcpy(&w, p + byte, 8);
        return (w >> (off & 7)) & (((uint64_t)1 << nb) - 1);
      }
    }
    // near either end of the stream: byte by byte, bits before the start are zero
    int64_t at = off; int take = nb;
    if (at < 0) { take += (int)at; at = 0; }
    uint64_t v = 0;
    if (take > 0) {
      size_t byte = (size_t)(at >> 3);
      int sh = (int)(at & 7), got = 0;
      while (got < take) {
        uint64_t bb = p[byte++] >> sh;
        int can = 8 - sh;
        v |= bb << got;
        got += can; sh = 0;
      }
      if (take < 64) v &= ((uint64_t)1 << take) - 1;
    }
    if (off < 0) v = (-off >= 64) ? 0 : v << (-off);
    return v;
  }
};

// one state of a sequence table with everything the decoder needs in 8 bytes: the symbol's base value and extra-bit count
// (literal length / match length / offset code) and the state transition
struct SeqEntry { uint32_t base_value; uint16_t next_base; uint8_t extra_bits; uint8_t nbits; };

struct FseTable {
  int log = 0;
  std::vector<uint8_t> symbol, nbits;
  std::vector<uint16_t> base;
  std::vector<SeqEntry> seq;            // sequence tables only (seq_entries)
  bool set = false;
};

inline void fse_build(const int16_t* freq, int nsym, int log, FseTable& t) {
  const int size = 1 << log;
  t.log = log; t.symbol.assign(size, 0); t.nbits.assign(size, 0); t.base.assign(size, 0); t.set = true;
  std::vector<uint16_t> desc(nsym, 0);
  int high = size;
  for (int s = 0; s < nsym; s++)
    if (freq[s] == -1) { t.symbol[--high] = (uint8_t)s; desc[s] = 1; }
  const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  int pos = 0;
  for (int s = 0; s < nsym; s++) {
    if (freq[s] <= 0) continue;
    desc[s] = (uint16_t)freq[s];
    for (int i = 0; i < freq[s]; i++) {
      t.symbol[pos] = (uint8_t)s;
      do { pos = (pos + step) & mask; } while (pos >= high);
    }
  }
  if (pos != 0) throw CodecError("zstd: FSE distribution does not fill its table");
  for (int i = 0; i < size; i++) {
    const int s = t.symbol[i];
    const uint16_t next = desc[s]++;
    t.nbits[i] = (uint8_t)(log - highest_bit(next));
    t.base[i] = (uint16_t)(((uint32_t)next << t.nbits[i]) - size);
  }
}

// 4.1.1: FSE table description -> normalised counts (-1 = "less than one"); returns the bytes it took
inline size_t fse_read_norm(const uint8_t* p, size_t n, int max_log, int max_sym, int16_t* freq, int* nsym, int* log_out) {
  FwdBits br{p, n};
  const int log = 5 + (int)br.read(4);
  if (log > max_log) throw CodecError("zstd: FSE accuracy log too large");
  int remaining = 1 << log, sym = 0;
  while (remaining > 0 && sym <= max_sym) {
    int bits = highest_bit((uint64_t)remaining + 1) + 1;
    uint32_t val = br.read(bits);
    const uint32_t lower = ((uint32_t)1 << (bits - 1)) - 1;
    const uint32_t threshold = ((uint32_t)1 << bits) - 1 - (uint32_t)(remaining + 1);
    if ((val & lower) < threshold) { br.rewind(1); val &= lower; }
    else if (val > lower) val -= threshold;
    const int proba = (int)val - 1;
    remaining -= proba < 0 ? -proba : proba;
    freq[sym++] = (int16_t)proba;
    if (proba == 0) {
      uint32_t rep = br.read(2);
      for (;;) {
        for (uint32_t i = 0; i < rep && sym <= max_sym; i++) freq[sym++] = 0;
        if (rep == 3) rep = br.read(2); else break;
      }
    }
  }
  if (remaining != 0 || sym > max_sym + 1) throw CodecError("zstd: malformed FSE distribution");
  *nsym = sym; *log_out = log;
  return br.bytes_used();
}

// ... -> table
inline size_t fse_read(const uint8_t* p, size_t n, int max_log, int max_sym, FseTable& t) {
  int16_t freq[256];
  int sym = 0, log = 0;
  const size_t used = fse_read_norm(p, n, max_log, max_sym, freq, &sym, &log);
  fse_build(freq, sym, log, t);
  return used;
}

struct HufTable {
  int max_bits = 0;
  std::vector<uint8_t> symbol, nbits;
  std::vector<uint16_t> entry;          // (code length << 8) | symbol: one load per decoded byte in the fast loop
  bool set = false;
};

inline void huf_build(const uint8_t* bits, int nsym, HufTable& t) {
  int max_bits = 0;
  uint32_t rank_count[17] = {0};
  for (int i = 0; i < nsym; i++) {
    if (bits[i] > 11) throw CodecError("zstd: Huffman code longer than 11 bits");
    max_bits = bits[i] > max_bits ? bits[i] : max_bits;
    rank_count[bits[i]]++;
  }
  if (max_bits == 0) throw CodecError("zstd: empty Huffman tree");
  const uint32_t size = 1u << max_bits;
  t.max_bits = max_bits; t.symbol.assign(size, 0); t.nbits.assign(size, 0); t.set = true;
  uint32_t rank_idx[18];
  rank_idx[max_bits] = 0;
  for (int i = max_bits; i >= 1; i--) {
    rank_idx[i - 1] = rank_idx[i] + rank_count[i] * (1u << (max_bits - i));
    if (rank_idx[i - 1] > size) throw CodecError("zstd: Huffman weights overflow the table");
    for (uint32_t k = rank_idx[i]; k < rank_idx[i - 1]; k++) t.nbits[k] = (uint8_t)i;
  }
  if (rank_idx[0] != size) throw CodecError("zstd: Huffman weights do not fill the table");
  for (int s = 0; s < nsym; s++) {
    if (!bits[s]) continue;
    const uint32_t code = rank_idx[bits[s]], len = 1u << (max_bits - bits[s]);
    for (uint32_t k = 0; k < len; k++) t.symbol[code + k] = (uint8_t)s;
    rank_idx[bits[s]] += len;
  }
  t.entry.resize(size);
  for (uint32_t k = 0; k < size; k++) t.entry[k] = (uint16_t)((uint16_t)t.nbits[k] << 8 | t.symbol[k]);
}

// 4.2.1: Huffman tree description -> code length per symbol (bits[0 .. *nsym)); returns the bytes it took
inline size_t huf_read_bits(const uint8_t* p, size_t n, uint8_t* bits, int* nsym) {
  if (n < 1) throw CodecError("zstd: missing Huffman tree description");
  const int hb = p[0];
  uint8_t weights[260];
  int nw = 0;
  size_t used;
  if (hb >= 128) {
    nw = hb - 127;
    const size_t bytes = (size_t)(nw + 1) / 2;
    if (bytes > n - 1) throw CodecError("zstd: Huffman weights run past the literals section");
    for (int i = 0; i < nw; i++) weights[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
    used = 1 + bytes;
  } else {
    if ((size_t)hb > n - 1 || hb == 0) throw CodecError("zstd: FSE-compressed Huffman weights run past the literals section");
    const uint8_t* q = p + 1;
    FseTable ft;
    const size_t th = fse_read(q, (size_t)hb, 6, 255, ft);
    if (th >= (size_t)hb) throw CodecError("zstd: no room for the Huffman weight stream");
    BackBits bs(q + th, (size_t)hb - th);
    uint32_t s1 = (uint32_t)bs.read(ft.log), s2 = (uint32_t)bs.read(ft.log);
    for (;;) {
      if (nw >= 254) throw CodecError("zstd: too many Huffman weights");     // two more may follow below: 255 weights + the implied one = 256 symbols
      weights[nw++] = ft.symbol[s1];
      s1 = ft.base[s1] + (uint32_t)bs.read(ft.nbits[s1]);
      if (bs.off < 0) { weights[nw++] = ft.symbol[s2]; break; }
      if (nw >= 254) throw CodecError("zstd: too many Huffman weights");
      weights[nw++] = ft.symbol[s2];
      s2 = ft.base[s2] + (uint32_t)bs.read(ft.nbits[s2]);
      if (bs.off < 0) { weights[nw++] = ft.symbol[s1]; break; }
    }
    used = 1 + (size_t)hb;
  }
  // weights -> code lengths; the last symbol's weight is implied (the sum of 2^(w-1) is a power of two)
  uint64_t sum = 0;
  for (int i = 0; i < nw; i++) {
    if (weights[i] > 11) throw CodecError("zstd: Huffman weight above 11");
    sum += weights[i] ? (uint64_t)1 << (weights[i] - 1) : 0;
  }
  if (sum == 0) throw CodecError("zstd: all Huffman weights are zero");
  const int max_bits = highest_bit(sum) + 1;
  const uint64_t left = ((uint64_t)1 << max_bits) - sum;
  if (left & (left - 1)) throw CodecError("zstd: Huffman weights do not leave a power of two");
  const int last_weight = highest_bit(left) + 1;
  for (int i = 0; i < nw; i++) bits[i] = weights[i] ? (uint8_t)(max_bits + 1 - weights[i]) : 0;
  bits[nw] = (uint8_t)(max_bits + 1 - last_weight);
  *nsym = nw + 1;
  return used;
}

// ... -> table
inline size_t huf_read(const uint8_t* p, size_t n, HufTable& t) {
  uint8_t bits[260];
  int nsym = 0;
  const size_t used = huf_read_bits(p, n, bits, &nsym);
  huf_build(bits, nsym, t);
  return used;
}

// one Huffman-coded stream: the next max_bits bits (the first-read bit most significant) index the table; a symbol consumes its code length
struct HufCursor {
  const uint8_t* p; size_t n; int64_t off; uint8_t* out; size_t o, out_len;
  HufCursor(const uint8_t* src, size_t len, uint8_t* dst, size_t dst_len) : p(src), n(len), out(dst), o(0), out_len(dst_len) {
    if (len == 0 || src[len - 1] == 0) throw CodecError("zstd: backward bit stream without its end mark");
    off = (int64_t)len * 8 - (8 - highest_bit(src[len - 1]));
  }
  // bits [off - mb, off) with zero fill below the start of the stream
  uint32_t peek(int mb) const {
    const int64_t lo = off - mb;
    if (lo >= 0) {
      const size_t byte = (size_t)(lo >> 3);
      if (byte + 8 <= n) { uint64_t w; memcpy(&w, p + byte, 8); return (uint32_t)(w >> (lo & 7)) & ((1u << mb) - 1); }
    }
    uint32_t v = 0;
    for (int i = 0; i < mb; i++) {
      const int64_t bit = lo + i;
      if (bit >= 0 && (size_t)(bit >> 3) < n) v |= (uint32_t)((p[bit >> 3] >> (bit & 7)) & 1) << i;
    }
    return v;
  }
};

inline void huf_finish(const HufTable& t, HufCursor& c) {
  const int mb = t.max_bits;
  // the stream is exhausted exactly when every bit has been consumed: the last symbol's code ends at bit 0
  while (c.off > 0) {
    if (c.o >= c.out_len) throw CodecError("zstd: Huffman stream longer than its regenerated size");
    const uint32_t idx = c.peek(mb);
    c.out[c.o++] = t.symbol[idx];
    c.off -= t.nbits[idx];
  }
  if (c.off != 0 || c.o != c.out_len) throw CodecError("zstd: Huffman stream does not end where it should");
}

inline void huf_stream(const HufTable& t, const uint8_t* p, size_t n, uint8_t* out, size_t out_len) {
  HufCursor c(p, n, out, out_len);
  huf_finish(t, c);
}

// the four streams of a literals section side by side: four independent dependency chains per loop iteration
inline void huf_streams4(const HufTable& t, const uint8_t* const src[4], const size_t len[4], uint8_t* const dst[4], const size_t dst_len[4]) {
  HufCursor c0(src[0], len[0], dst[0], dst_len[0]), c1(src[1], len[1], dst[1], dst_len[1]), c2(src[2], len[2], dst[2], dst_len[2]), c3(src[3], len[3], dst[3], dst_len[3]);
  const int mb = t.max_bits, top = 64 - mb;
  const uint16_t* tbl = t.entry.data();
  // Four independent dependency chains per iteration.  Each stream refills a 64-bit container once (an unaligned load whose top
  // >= 56 bits are the next bits of the stream, first-read bit most significant) and takes five symbols from it: 5 x 11 bits fit.
  // While every stream has >= 64 bits and 5 output slots left, no bound can be crossed.
  auto refill = [](const HufCursor& c) -> uint64_t {
    const int64_t lo = c.off - 56;
    const size_t byte = (size_t)(lo >> 3);
    uint64_t w;
    memcpy(&w, c.p + byte, 8);
    return w << (64 - (c.off - (int64_t)byte * 8));
  };
  while (c0.off >= 64 && c1.off >= 64 && c2.off >= 64 && c3.off >= 64 && c0.o + 5 <= c0.out_len && c1.o + 5 <= c1.out_len && c2.o + 5 <= c2.out_len && c3.o + 5 <= c3.out_len) {
    uint64_t b0 = refill(c0), b1 = refill(c1), b2 = refill(c2), b3 = refill(c3);
    int u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    for (int k = 0; k < 5; k++) {
      const uint16_t e0 = tbl[b0 >> top], e1 = tbl[b1 >> top], e2 = tbl[b2 >> top], e3 = tbl[b3 >> top];
      c0.out[c0.o + k] = (uint8_t)e0; c1.out[c1.o + k] = (uint8_t)e1; c2.out[c2.o + k] = (uint8_t)e2; c3.out[c3.o + k] = (uint8_t)e3;
      b0 <<= e0 >> 8; b1 <<= e1 >> 8; b2 <<= e2 >> 8; b3 <<= e3 >> 8;
      u0 += e0 >> 8; u1 += e1 >> 8; u2 += e2 >> 8; u3 += e3 >> 8;
    }
    c0.o += 5; c1.o += 5; c2.o += 5; c3.o += 5;
    c0.off -= u0; c1.off -= u1; c2.off -= u2; c3.off -= u3;
  }
  huf_finish(t, c0); huf_finish(t, c1); huf_finish(t, c2); huf_finish(t, c3);
}

const uint32_t kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
const uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const uint32_t kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59,
                              67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
const uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const int16_t kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
const int16_t kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
const int16_t kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};

struct FrameState {
  HufTable huf;
  FseTable ll, of, ml;
  uint64_t rep[3] = {1, 4, 8};
};

// one of the three sequence tables (3.1.1.3.2.1): mode 0 predefined, 1 RLE, 2 FSE description, 3 repeat
inline size_t seq_table(int mode, const uint8_t* p, size_t n, int max_log, int max_sym, const int16_t* dflt, int ndflt, int dflt_log, FseTable& t) {
  switch (mode) {
    case 0: fse_build(dflt, ndflt, dflt_log, t); return 0;
    case 1: {
      if (n < 1) throw CodecError("zstd: missing RLE symbol");
      if (p[0] > max_sym) throw CodecError("zstd: RLE symbol out of range");
      t.log = 0; t.symbol.assign(1, p[0]); t.nbits.assign(1, 0); t.base.assign(1, 0); t.set = true;
      return 1;
    }
    case 2: return fse_read(p, n, max_log, max_sym, t);
    default:
      if (!t.set) throw CodecError("zstd: repeat mode without a previous table");
      return 0;
  }
}

// kind 0: literal lengths, 1: offsets (base 2^code, code extra bits), 2: match lengths
inline void seq_entries(FseTable& t, int kind) {
  const size_t size = t.symbol.size();
  t.seq.resize(size);
  for (size_t i = 0; i < size; i++) {
    const int c = t.symbol[i];
    SeqEntry e;
    if (kind == 0) { if (c > 35) throw CodecError("zstd: sequence code out of range"); e.base_value = kLLBase[c]; e.extra_bits = kLLBits[c]; }
    else if (kind == 2) { if (c > 52) throw CodecError("zstd: sequence code out of range"); e.base_value = kMLBase[c]; e.extra_bits = kMLBits[c]; }
    else { if (c > 31) throw CodecError("zstd: sequence code out of range"); e.base_value = (uint32_t)1 << c; e.extra_bits = (uint8_t)c; }
    e.next_base = t.base[i]; e.nbits = t.nbits[i];
    t.seq[i] = e;
  }
}

inline void block_compressed(FrameState& fs, const uint8_t* p, size_t n, std::vector<uint8_t>& lit_buf, uint8_t* out, size_t out_cap, size_t* op_io) {
  // ---- literals section (3.1.1.3.1) ----
  if (n < 1) throw CodecError("zstd: empty compressed block");
  const int ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
  size_t regen = 0, comp = 0, hdr = 0;
  int streams = 1;
  if (ltype < 2) {
    if (sf == 0 || sf == 2) { regen = p[0] >> 3; hdr = 1; }
    else if (sf == 1) { if (n < 2) throw CodecError("zstd: truncated literals header"); regen = (p[0] >> 4) | ((size_t)p[1] << 4); hdr = 2; }
    else { if (n < 3) throw CodecError("zstd: truncated literals header"); regen = (p[0] >> 4) | ((size_t)p[1] << 4) | ((size_t)p[2] << 12); hdr = 3; }
  } else {
    if (sf == 0 || sf == 1) {
      if (n < 3) throw CodecError("zstd: truncated literals header");
      const uint32_t v = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
      regen = (v >> 4) & 0x3ff; comp = (v >> 14) & 0x3ff; hdr = 3; streams = sf == 0 ? 1 : 4;
    } else if (sf == 2) {
      if (n < 4) throw CodecError("zstd: truncated literals header");
      const uint32_t v = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
      regen = (v >> 4) & 0x3fff; comp = v >> 18; hdr = 4; streams = 4;
    } else {
      if (n < 5) throw CodecError("zstd: truncated literals header");
      const uint64_t v = p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24) | ((uint64_t)p[4] << 32);
      regen = (size_t)((v >> 4) & 0x3ffff); comp = (size_t)(v >> 22); hdr = 5; streams = 4;
    }
  }
  if (regen > (size_t)128 * 1024) throw CodecError("zstd: literals larger than a block");
  lit_buf.resize(regen + 32);            // slack: literals are copied 16 bytes at a time
  uint8_t* lits = lit_buf.data();
  size_t pos = hdr;
  if (ltype == 0) {
    if (regen > n - pos) throw CodecError("zstd: raw literals past the block");
    memcpy(lits, p + pos, regen); pos += regen;
  } else if (ltype == 1) {
    if (pos >= n) throw CodecError("zstd: RLE literal missing");
    memset(lits, p[pos], regen); pos += 1;
  } else {
    if (comp > n - pos) throw CodecError("zstd: compressed literals past the block");
    const uint8_t* q = p + pos;
    size_t left = comp;
    if (ltype == 2) { const size_t th = huf_read(q, left, fs.huf); q += th; left -= th; }
    else if (!fs.huf.set) throw CodecError("zstd: treeless literals without a previous Huffman table");
    if (streams == 1) huf_stream(fs.huf, q, left, lits, regen);
    else {
      if (left < 6) throw CodecError("zstd: missing Huffman jump table");
      const size_t s1 = q[0] | ((size_t)q[1] << 8), s2 = q[2] | ((size_t)q[3] << 8), s3 = q[4] | ((size_t)q[5] << 8);
      if (s1 + s2 + s3 > left - 6) throw CodecError("zstd: Huffman streams past the literals section");
      const size_t s4 = left - 6 - s1 - s2 - s3, each = (regen + 3) / 4;
      if (each * 3 > regen) throw CodecError("zstd: regenerated size too small for four streams");
      const uint8_t* d = q + 6;
      const uint8_t* const src4[4] = {d, d + s1, d + s1 + s2, d + s1 + s2 + s3};
      const size_t len4[4] = {s1, s2, s3, s4};
      uint8_t* const dst4[4] = {lits, lits + each, lits + 2 * each, lits + 3 * each};
      const size_t dlen4[4] = {each, each, each, regen - 3 * each};
      huf_streams4(fs.huf, src4, len4, dst4, dlen4);
    }
    pos += comp;
  }
  // ---- sequences section (3.1.1.3.2) ----
  if (pos >= n) throw CodecError("zstd: missing sequences section");
  size_t nseq = p[pos++];
  if (nseq >= 128) {
    if (nseq < 255) { if (pos >= n) throw CodecError("zstd: truncated sequence count"); nseq = ((nseq - 128) << 8) + p[pos++]; }
    else { if (n - pos < 2) throw CodecError("zstd: truncated sequence count"); nseq = p[pos] + ((size_t)p[pos + 1] << 8) + 0x7f00; pos += 2; }
  }
  size_t op = *op_io, lp = 0;
  if (nseq == 0) {
    if (pos != n) throw CodecError("zstd: bytes after an empty sequences section");
    if (regen > out_cap - op) throw CodecError("zstd: output larger than the page header says");
    memcpy(out + op, lits, regen);
    *op_io = op + regen;
    return;
  }
  if (pos >= n) throw CodecError("zstd: missing compression modes");
  const int modes = p[pos++];
  if (modes & 3) throw CodecError("zstd: reserved bits set in the compression modes");
  { const int m = (modes >> 6) & 3; pos += seq_table(m, p + pos, n - pos, 9, 35, kLLDefault, 36, 6, fs.ll); if (m != 3 || fs.ll.seq.empty()) seq_entries(fs.ll, 0); }
  { const int m = (modes >> 4) & 3; pos += seq_table(m, p + pos, n - pos, 8, 31, kOFDefault, 29, 5, fs.of); if (m != 3 || fs.of.seq.empty()) seq_entries(fs.of, 1); }
  { const int m = (modes >> 2) & 3; pos += seq_table(m, p + pos, n - pos, 9, 52, kMLDefault, 53, 6, fs.ml); if (m != 3 || fs.ml.seq.empty()) seq_entries(fs.ml, 2); }
  if (pos >= n) throw CodecError("zstd: missing sequence bit stream");
  BackBits bs(p + pos, n - pos);
  uint32_t sl = (uint32_t)bs.read(fs.ll.log), so = (uint32_t)bs.read(fs.of.log), sm = (uint32_t)bs.read(fs.ml.log);
  const SeqEntry* const tl = fs.ll.seq.data();
  const SeqEntry* const to = fs.of.seq.data();
  const SeqEntry* const tm = fs.ml.seq.data();
  const uint8_t* const sp = bs.p;
  const int64_t fast_hi = (int64_t)bs.n * 8 - 64;        // below this bit offset an 8-byte load never passes the end of the stream
  uint64_t rep0 = fs.rep[0], rep1 = fs.rep[1], rep2 = fs.rep[2];
  for (size_t i = 0; i < nseq; i++) {
    const SeqEntry el = tl[sl], eo = to[so], em = tm[sm];
    uint64_t ov, ml, ll;
    if (bs.off <= fast_hi && bs.off >= 160 && i + 1 < nseq) {
      // <= 89 bits per sequence, far from both ends of the stream: unchecked 8-byte loads (a zero-bit read masks to 0)
      // three loads: the offset's extra bits (<= 31), both lengths' (<= 32 together), the three state updates (<= 26 together)
      int64_t off = bs.off;
      auto window = [&](int nb) -> uint64_t { off -= nb; uint64_t w; memcpy(&w, sp + (off >> 3), 8); return w >> (off & 7); };
      auto low = [](uint64_t v, int nb) -> uint64_t { return v & (((uint64_t)1 << nb) - 1); };
      ov = eo.base_value + low(window(eo.extra_bits), eo.extra_bits);
      const uint64_t w2 = window(em.extra_bits + el.extra_bits);            // [ll extra | ml extra] from the low end up: ll was written last
      ll = el.base_value + low(w2, el.extra_bits);
      ml = em.base_value + low(w2 >> el.extra_bits, em.extra_bits);
      const uint64_t w3 = window(el.nbits + em.nbits + eo.nbits);
      so = eo.next_base + (uint32_t)low(w3, eo.nbits);
      sm = em.next_base + (uint32_t)low(w3 >> eo.nbits, em.nbits);
      sl = el.next_base + (uint32_t)low(w3 >> (eo.nbits + em.nbits), el.nbits);
      bs.off = off;
    } else {
      ov = eo.base_value + bs.read(eo.extra_bits);
      ml = em.base_value + bs.read(em.extra_bits);
      ll = el.base_value + bs.read(el.extra_bits);
      if (i + 1 < nseq) {
        sl = el.next_base + (uint32_t)bs.read(el.nbits);
        sm = em.next_base + (uint32_t)bs.read(em.nbits);
        so = eo.next_base + (uint32_t)bs.read(eo.nbits);
      }
      if (bs.off < 0) throw CodecError("zstd: sequence bit stream ends early");
    }
    uint64_t offset;
    if (ov > 3) {
      offset = ov - 3;
      rep2 = rep1; rep1 = rep0; rep0 = offset;
    } else {
      uint64_t idx = ov - 1;
      if (ll == 0) idx++;
      if (idx == 0) offset = rep0;
      else {
        offset = idx == 1 ? rep1 : idx == 2 ? rep2 : rep0 - 1;
        if (idx > 1) rep2 = rep1;
        rep1 = rep0; rep0 = offset;
      }
    }
    if (ll > regen - lp || ll > out_cap - op) throw CodecError("zstd: sequence literals past their section / the output");
    if (ll <= 16 && out_cap - op >= 16) copy16(out + op, lits + lp);       // lit_buf has 32 bytes of slack
    else memcpy(out + op, lits + lp, ll);
    op += ll; lp += ll;
    if (offset == 0 || offset > op || ml > out_cap - op) throw CodecError("zstd: match outside the output (dictionaries are not supported)");
    match_copy(out + op, offset, ml, out_cap - op - ml);
    op += ml;
  }
  fs.rep[0] = rep0; fs.rep[1] = rep1; fs.rep[2] = rep2;
  if (bs.off != 0) throw CodecError("zstd: sequence bit stream not consumed exactly");
  const size_t rest = regen - lp;
  if (rest > out_cap - op) throw CodecError("zstd: output larger than the page header says");
  memcpy(out + op, lits + lp, rest);
  *op_io = op + rest;
}

}  // namespace zstd_detail

// every frame of [in, in + n) into out[0, out_len); throws unless exactly out_len bytes result
inline void zstd_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len) {
  using namespace zstd_detail;
  size_t ip = 0, op = 0;
  std::vector<uint8_t> lit_buf;
  while (ip < n) {
    if (n - ip < 4) throw CodecError("zstd: truncated frame magic");
    uint32_t magic;
    memcpy(&magic, in + ip, 4);
    if ((magic & 0xfffffff0u) == 0x184d2a50u) {            // skippable frame
      if (n - ip < 8) throw CodecError("zstd: truncated skippable frame");
      uint32_t len;
      memcpy(&len, in + ip + 4, 4);
      if (len > n - ip - 8) throw CodecError("zstd: skippable frame past the end");
      ip += 8 + len;
      continue;
    }
    if (magic != 0xfd2fb528u) throw CodecError("zstd: not a Zstandard frame");
    ip += 4;
    if (ip >= n) throw CodecError("zstd: truncated frame header");
    const uint8_t fhd = in[ip++];
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did_flag = fhd & 3;
    if (fhd & 8) throw CodecError("zstd: reserved bit set in the frame header");
    if (!single) { if (ip >= n) throw CodecError("zstd: truncated frame header"); ip++; }       // window descriptor: the whole output is the window here
    const int did_bytes = did_flag == 3 ? 4 : did_flag;
    if ((size_t)did_bytes > n - ip) throw CodecError("zstd: truncated frame header");
    uint32_t did = 0;
    for (int i = 0; i < did_bytes; i++) did |= (uint32_t)in[ip + i] << (8 * i);
    if (did) throw CodecError("zstd: frame needs a dictionary");
    ip += did_bytes;
    const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
    if ((size_t)fcs_bytes > n - ip) throw CodecError("zstd: truncated frame header");
    ip += fcs_bytes;                                        // the page header is the authority on the size
    FrameState fs;
    const size_t frame_start = op;
    for (;;) {
      if (n - ip < 3) throw CodecError("zstd: truncated block header");
      const uint32_t bh = in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
      ip += 3;
      const int last = bh & 1, type = (bh >> 1) & 3;
      const size_t bsize = bh >> 3;
      if (type == 0) {
        if (bsize > n - ip || bsize > out_len - op) throw CodecError("zstd: raw block past the end");
        memcpy(out + op, in + ip, bsize);
        ip += bsize; op += bsize;
      } else if (type == 1) {
        if (ip >= n || bsize > out_len - op) throw CodecError("zstd: RLE block past the end");
        memset(out + op, in[ip], bsize);
        ip += 1; op += bsize;
      } else if (type == 2) {
        if (bsize > n - ip) throw CodecError("zstd: compressed block past the end");
        // matches may reach back to the start of the frame only: hand the block a view that starts there
        size_t rel = op - frame_start;
        block_compressed(fs, in + ip, bsize, lit_buf, out + frame_start, out_len - frame_start, &rel);
        op = frame_start + rel;
        ip += bsize;
      } else {
        throw CodecError("zstd: reserved block type");
      }
      if (last) break;
    }
    if (checksum) { if (n - ip < 4) throw CodecError("zstd: truncated checksum"); ip += 4; }
  }
  if (op != out_len) throw CodecError("zstd: frames decode to a different length than the page header says");
}

}  // namespace codec
}  // namespace plx
