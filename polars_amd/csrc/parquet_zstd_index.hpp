// parquet_zstd_index.hpp -- the host half of the device zstd decoder (parquet_zstd.hpp): frame / block / section HEADERS and table
// DESCRIPTIONS of a compressed page -> ZstdBlock records, normalised counts and code lengths.  Metadata only: a few dozen bytes per block are
// read; literals, sequences and matches are decoded by the kernels.  RFC 8878 3.1.1 (frames), 3.1.1.2 (blocks), 3.1.1.3.1 (literals section
// header), 3.1.1.3.2 (sequences section header), 4.1.1 (FSE table description), 4.2.1 (Huffman tree description).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "host_codecs.hpp"
#include "parquet_zstd.hpp"

namespace plx {
namespace pq {

struct ZstdPlan {          // the streams of one launch
  std::vector<ZstdStream> streams;
  std::vector<ZstdBlock> blocks;
  std::vector<ZstdHufDesc> hufs;
  std::vector<ZstdFseDesc> fses;
  uint64_t lit_bytes = 0;  // literal buffers of the Huffman-coded blocks, back to back (each padded to 16 bytes)
  uint64_t n_seq = 0;      // sequence records
  uint64_t n_compressed = 0;

  ZstdPlan() { reset(); }
  void reset() {
    streams.clear(); blocks.clear(); hufs.clear(); fses.clear();
    lit_bytes = 0; n_seq = 0; n_compressed = 0;
    // entries 0..2: the predefined distributions (3.1.1.3.2.2)
    using namespace codec::zstd_detail;
    fses.push_back(make(kLLDefault, 36, 6));
    fses.push_back(make(kOFDefault, 29, 5));
    fses.push_back(make(kMLDefault, 53, 6));
  }
  static ZstdFseDesc make(const int16_t* norm, int nsym, int log) {
    ZstdFseDesc d;
    memset(&d, 0, sizeof d);
    for (int i = 0; i < nsym; i++) d.norm[i] = norm[i];
    d.log = (uint8_t)log; d.nsym = (uint8_t)nsym;
    return d;
  }
  bool empty() const { return streams.empty(); }
};

namespace zstd_index_detail {
using codec::CodecError;

// the tables in force while a frame is walked: indices into the plan, -1 = none yet
struct Tables { int64_t huf = -1, tab[3] = {-1, -1, -1}; };

inline void index_compressed(ZstdPlan& plan, Tables& tb, ZstdBlock& blk, const uint8_t* p, size_t n, uint64_t dev) {
  using namespace codec::zstd_detail;
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
  if (regen > (size_t)kZBlockMax) throw CodecError("zstd: literals larger than a block");
  size_t pos = hdr;
  blk.regen = (uint32_t)regen;
  blk.lit_streams = 1;
  if (ltype == 0) {
    if (regen > n - pos) throw CodecError("zstd: raw literals past the block");
    blk.lit_type = ZL_RAW; blk.lit = dev + pos;
    pos += regen;
  } else if (ltype == 1) {
    if (pos >= n) throw CodecError("zstd: RLE literal missing");
    blk.lit_type = ZL_RLE; blk.lit = p[pos];
    pos += 1;
  } else {
    if (comp > n - pos) throw CodecError("zstd: compressed literals past the block");
    const uint8_t* q = p + pos;
    size_t left = comp;
    if (ltype == 2) {
      ZstdHufDesc d;
      memset(&d, 0, sizeof d);
      uint8_t bits[260];
      int nsym = 0;
      const size_t th = huf_read_bits(q, left, bits, &nsym);
      // huf_read_bits made the weights complete (the implied last one fills the code space to a power of two), so the codes fill a table of 2^max_bits entries exactly;
      // what is left to check is the longest code (4.2.1: 11 bits)
      int max_bits = 0;
      for (int i = 0; i < nsym; i++) max_bits = bits[i] > max_bits ? bits[i] : max_bits;
      if (max_bits > 11) throw CodecError("zstd: Huffman code longer than 11 bits");
      if (max_bits == 0) throw CodecError("zstd: empty Huffman tree");
      memcpy(d.bits, bits, (size_t)nsym);
      d.nsym = (uint32_t)nsym; d.max_bits = (uint32_t)max_bits;
      tb.huf = (int64_t)plan.hufs.size();
      plan.hufs.push_back(d);
      q += th; left -= th;
    } else if (tb.huf < 0) throw CodecError("zstd: treeless literals without a previous Huffman table");
    if (streams == 4) {
      if (left < 6) throw CodecError("zstd: missing Huffman jump table");
      const size_t s1 = q[0] | ((size_t)q[1] << 8), s2 = q[2] | ((size_t)q[3] << 8), s3 = q[4] | ((size_t)q[5] << 8);
      if (s1 + s2 + s3 > left - 6) throw CodecError("zstd: Huffman streams past the literals section");
      if (((regen + 3) / 4) * 3 > regen) throw CodecError("zstd: regenerated size too small for four streams");
    }
    blk.lit_type = ZL_HUFFMAN; blk.lit_streams = (uint8_t)streams;
    blk.huf = (uint32_t)tb.huf;
    blk.huf_off = (uint32_t)(q - p); blk.huf_len = (uint32_t)left;
    blk.lit = plan.lit_bytes;                      // relative to the launch's literal scratch: zstd_plan_place adds its address
    plan.lit_bytes += (regen + 15) & ~(uint64_t)15;
    pos += comp;
  }
  // ---- sequences section header ----
  if (pos >= n) throw CodecError("zstd: missing sequences section");
  size_t nseq = p[pos++];
  if (nseq >= 128) {
    if (nseq < 255) { if (pos >= n) throw CodecError("zstd: truncated sequence count"); nseq = ((nseq - 128) << 8) + p[pos++]; }
    else { if (n - pos < 2) throw CodecError("zstd: truncated sequence count"); nseq = p[pos] + ((size_t)p[pos + 1] << 8) + 0x7f00; pos += 2; }
  }
  blk.nseq = (uint32_t)nseq;
  if (nseq == 0) {
    if (pos != n) throw CodecError("zstd: bytes after an empty sequences section");
    return;
  }
  if (pos >= n) throw CodecError("zstd: missing compression modes");
  const int modes = p[pos++];
  if (modes & 3) throw CodecError("zstd: reserved bits set in the compression modes");
  const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
  for (int t = 0; t < 3; t++) {
    const int m = (modes >> (6 - 2 * t)) & 3;
    if (m == 0) tb.tab[t] = t;
    else if (m == 1) {
      if (pos >= n) throw CodecError("zstd: missing RLE symbol");
      if (p[pos] > max_sym[t]) throw CodecError("zstd: RLE symbol out of range");
      ZstdFseDesc d;
      memset(&d, 0, sizeof d);
      d.rle = 1; d.rle_sym = p[pos];
      tb.tab[t] = (int64_t)plan.fses.size();
      plan.fses.push_back(d);
      pos += 1;
    } else if (m == 2) {
      int16_t freq[256];
      int nsym = 0, log = 0;
      const size_t used = fse_read_norm(p + pos, n - pos, max_log[t], max_sym[t], freq, &nsym, &log);
      // (fse_read_norm checked that the counts sum to the table size; the spread step is odd, so every state is visited once: nothing else can go wrong in the build)
      ZstdFseDesc d = ZstdPlan::make(freq, nsym, log);
      tb.tab[t] = (int64_t)plan.fses.size();
      plan.fses.push_back(d);
      pos += used;
    } else if (tb.tab[t] < 0) throw CodecError("zstd: repeat mode without a previous table");
    blk.tab[t] = (uint32_t)tb.tab[t];
  }
  if (pos >= n) throw CodecError("zstd: missing sequence bit stream");
  if (p[n - 1] == 0) throw CodecError("zstd: backward bit stream without its end mark");
  blk.bits_off = (uint32_t)pos; blk.bits_len = (uint32_t)(n - pos);
  blk.seq = plan.n_seq;                            // record index: zstd_plan_place turns it into an address
  plan.n_seq += nseq;
}
}  // namespace zstd_index_detail

// One compressed page: `host` = its stored bytes, `dev` = where the same bytes will lie in HBM.  Appends a stream (dst left 0: the launch
// places it) and its blocks to the plan; throws codec::CodecError on a malformed header.
inline void zstd_index_stream(ZstdPlan& plan, const uint8_t* in, size_t n, uint64_t dev, uint32_t uncomp_size) {
  using codec::CodecError;
  using namespace zstd_index_detail;
  ZstdStream s;
  memset(&s, 0, sizeof s);
  s.uncomp_size = uncomp_size;
  s.first_block = (uint32_t)plan.blocks.size();
  size_t ip = 0;
  uint64_t known_out = 0;          // raw + RLE bytes: may not exceed the page on their own
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
    if (!single) { if (ip >= n) throw CodecError("zstd: truncated frame header"); ip++; }
    const int did_bytes = did_flag == 3 ? 4 : did_flag;
    if ((size_t)did_bytes > n - ip) throw CodecError("zstd: truncated frame header");
    uint32_t did = 0;
    for (int i = 0; i < did_bytes; i++) did |= (uint32_t)in[ip + i] << (8 * i);
    if (did) throw CodecError("zstd: frame needs a dictionary");
    ip += did_bytes;
    const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
    if ((size_t)fcs_bytes > n - ip) throw CodecError("zstd: truncated frame header");
    ip += fcs_bytes;                                        // the page header is the authority on the size
    Tables tb;
    bool first = true;
    for (;;) {
      if (n - ip < 3) throw CodecError("zstd: truncated block header");
      const uint32_t bh = in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
      ip += 3;
      const int last = bh & 1, type = (bh >> 1) & 3;
      const size_t bsize = bh >> 3;
      ZstdBlock blk;
      memset(&blk, 0, sizeof blk);
      blk.src = dev + ip;
      blk.first_in_frame = first ? 1 : 0;
      first = false;
      if (type == 0) {
        if (bsize > n - ip) throw CodecError("zstd: raw block past the end");
        blk.type = ZB_RAW; blk.src_len = (uint32_t)bsize; blk.out_len = (uint32_t)bsize;
        known_out += bsize;
        ip += bsize;
      } else if (type == 1) {
        if (ip >= n) throw CodecError("zstd: RLE block past the end");
        blk.type = ZB_RLE; blk.src_len = 1; blk.out_len = (uint32_t)bsize; blk.lit = in[ip];
        known_out += bsize;
        ip += 1;
      } else if (type == 2) {
        if (bsize > n - ip) throw CodecError("zstd: compressed block past the end");
        if (bsize > (size_t)kZBlockMax) throw CodecError("zstd: block larger than 128 KB");
        blk.type = ZB_COMPRESSED; blk.src_len = (uint32_t)bsize;
        index_compressed(plan, tb, blk, in + ip, bsize, dev + ip);
        plan.n_compressed++;
        ip += bsize;
      } else {
        throw CodecError("zstd: reserved block type");
      }
      if (known_out > uncomp_size) throw CodecError("zstd: output larger than the page header says");
      plan.blocks.push_back(blk);
      if (last) break;
    }
    if (checksum) { if (n - ip < 4) throw CodecError("zstd: truncated checksum"); ip += 4; }
  }
  s.n_blocks = (uint32_t)plan.blocks.size() - s.first_block;
  // a page of nothing but sequence-free blocks of Huffman literals (prices, timestamps: zstd finds no matches in them) is decoded IN PLACE: the sizes are all in the headers
  {
    bool direct = s.n_blocks > 0;
    uint64_t total = 0;
    for (uint32_t b = 0; b < s.n_blocks; b++) {
      const ZstdBlock& k = plan.blocks[s.first_block + b];
      direct = direct && k.type == ZB_COMPRESSED && k.nseq == 0 && k.lit_type == ZL_HUFFMAN;
      total += k.regen;
    }
    if (direct) {
      if (total != uncomp_size) throw CodecError("zstd: frames decode to a different length than the page header says");
      uint32_t off = 0;
      for (uint32_t b = 0; b < s.n_blocks; b++) {
        ZstdBlock& k = plan.blocks[s.first_block + b];
        // (its slice of the literal scratch goes unused: a few KB per page)
        k.direct = 1; k.page_off = off;
        off += k.regen;
      }
      s.direct = 1;
    }
  }
  plan.streams.push_back(s);
}

// scratch addresses into the records (once the launch has allocated the literal buffers, the sequence records and the pages' outputs: streams[i].dst)
inline void zstd_plan_place(ZstdPlan& plan, uint64_t lit_base, uint64_t seq_base) {
  for (const ZstdStream& st : plan.streams)
    for (uint32_t b = 0; b < st.n_blocks; b++) {
      ZstdBlock& k = plan.blocks[st.first_block + b];
      if (k.type != ZB_COMPRESSED) continue;
      if (k.direct) k.lit = st.dst + k.page_off;
      else if (k.lit_type == ZL_HUFFMAN) k.lit += lit_base;
      k.seq = seq_base + k.seq * 16;
    }
}
// indices of the compressed blocks: first the blocks without sequences (Huffman literals only: sixteen a wavefront), then the others (four a wavefront); each part longest
// first (a launch lasts as long as its longest block started last, and a wavefront's blocks should be of a size)
inline std::vector<uint32_t> zstd_plan_order(const ZstdPlan& plan, uint32_t* n_huf_only) {
  std::vector<uint32_t> h, o;
  for (size_t i = 0; i < plan.blocks.size(); i++) {
    const ZstdBlock& k = plan.blocks[i];
    if (k.type != ZB_COMPRESSED) continue;
    (k.nseq == 0 && k.lit_type == ZL_HUFFMAN ? h : o).push_back((uint32_t)i);
  }
  auto longer = [&](uint32_t a, uint32_t b) { return plan.blocks[a].src_len > plan.blocks[b].src_len; };
  std::stable_sort(h.begin(), h.end(), longer);
  std::stable_sort(o.begin(), o.end(), longer);
  *n_huf_only = (uint32_t)h.size();
  h.insert(h.end(), o.begin(), o.end());
  return h;
}

}  // namespace pq
}  // namespace plx
