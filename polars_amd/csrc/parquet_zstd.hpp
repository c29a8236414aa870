// parquet_zstd.hpp -- Zstandard on the device (part of the Parquet scan: host + device bodies, like parquet_snappy.hpp).
//
// The reference inflates zstd pages with the zstd crate, one page after another (crates/polars-parquet/src/parquet/compression.rs:137-138,
// 221-236); Polars WRITES zstd by default (compression.rs:103-120).  Format: RFC 8878.
//
// A zstd frame is a chain four times over -- Huffman literals, FSE-coded sequences (one bit stream three state machines share), repeat
// offsets, LZ77 execution -- but the chains of the ENTROPY stages end at every block (<= 128 KB of output), and a 1 MB page is eight of them:
//
//   index    (host, while the page walk goes on: metadata only)  frame / block / section HEADERS are parsed into ZstdBlock records and the
//            table DESCRIPTIONS into normalised counts / code lengths (a few dozen bytes per block; repeat / treeless modes resolved to the
//            block that defined the table).  Nothing of the payload is decoded on the host.
//   entropy  all compressed blocks of all pages of the column at once, FOUR blocks a wavefront (sixteen lanes each; sixteen blocks a wavefront when they hold no
//            sequences): a block's chains run on single lanes -- sub-lanes build the Huffman table and the three FSE tables in LDS, sub-lanes 0..3 decode the (up to four)
//            Huffman streams into the block's literal buffer (or, for a page of nothing but such blocks, straight into the page's output), sub-lane 0 decodes the
//            sequences into {literal length, match length, offset VALUE} records -- and an instruction costs the wavefront the same whether one lane or four are live.
//            Repeat offsets cross block borders and are a chain of their own: they are left to the execute pass, so no block waits for its predecessor.
//   execute  one wavefront per page: blocks in order; 64 sequences at a time -- every lane places its own sequence's literals at their final position (prefix sums);
//            the repeat-offset history is a prefix "sum" too (every sequence is a small map on the three offsets, maps compose: a wavefront scan); then the matches:
//            batches of fixed-width ROWS ("a few literal bytes, the rest from the row above": sorted keys, timestamps) by one ballot per column, batches whose matches
//            copy earlier matches of the batch by pointer doubling over the matches, anything else in sequence order with all lanes copying bytes.  The last 16 KB of
//            output live in an LDS ring (a match that reads what the previous match wrote costs an LDS round trip, not an HBM one); the ring is flushed to HBM in
//            16-byte stores, matches that reach further back read the flushed bytes.
//
// Every phase is a function of (lane) between barriers, so the CPU harness (tests/emu/parquet_emu.cpp) runs the very same bodies lane after
// lane; W is the wavefront: W::lanes(f) runs f for every lane, W::sync() is the barrier.
#pragma once
#include <stdint.h>
#include <string.h>

#include "parquet_device.hpp"
#if !defined(__HIP_DEVICE_COMPILE__) && defined(PLX_ZSTD_DEBUG)
#include <cstdio>
#define ZDBG(...) fprintf(stderr, __VA_ARGS__)
#else
#define ZDBG(...)
#endif

namespace plx {
namespace pq {

constexpr uint32_t kZLanes = 64;                 // one wavefront per block (entropy) / per page (execute)
constexpr uint32_t kZRing = 16384;               // bytes of recent output held in LDS by the execute pass (16 KB: seven page wavefronts a CU; 32 KB: four -- a file of PLAIN values in 20 000-row pages read in 33-39 vs 40-48 ms)
constexpr uint32_t kZRingMask = kZRing - 1;
constexpr uint32_t kZPiece = 4096;               // bytes one cooperative copy step moves (long literal runs / long matches are cut into pieces)
constexpr uint32_t kZBatchSpan = kZRing / 2;     // output bytes of one batch of sequences
constexpr uint32_t kZLitLane = 64;               // literal bytes a lane places for its own sequence; longer runs go through the cooperative copy
constexpr uint32_t kZOfMask = (1u << 30) - 1;    // repeat-offset maps (execute pass): a slot's bits 31:30 = 0 -> an offset; j + 1 -> (incoming rep[j]) - low bits
constexpr uint32_t kZBlockMax = 128 * 1024;
constexpr uint32_t kZGroups = 4;                 // entropy pass: blocks per wavefront (a block's chains run on single lanes: four blocks share the instruction stream)
constexpr uint32_t kZGroupLanes = kZLanes / kZGroups;
constexpr uint32_t kZHufGroups = 16;             // ... blocks without sequences (Huffman literals only: 4.6 KB of LDS each): sixteen a wavefront, four lanes each
constexpr uint32_t kZHufGroupLanes = kZLanes / kZHufGroups;
constexpr uint32_t kZRowWidth = 32;              // execute pass: widest fixed-width value whose batches take the row path
constexpr uint32_t kZStageWords = 256;           // 8-byte words of a sequence bit stream staged in LDS at a time (+ 2 below them)

#if defined(__clang__)
#define PLX_UNROLL_Z _Pragma("unroll")
#else
#define PLX_UNROLL_Z
#endif

// a per-lane variable that lives across phases: a register on the device, one slot per lane on the CPU harness
template <class T> struct ZLaneVar {
#if defined(__HIP_DEVICE_COMPILE__)
  T x;
  PLX_HD T& operator[](uint32_t) { return x; }
#else
  T x[kZLanes];
  PLX_HD T& operator[](uint32_t lane) { return x[lane]; }
#endif
};

enum ZstdBlockType : uint8_t { ZB_RAW = 0, ZB_RLE = 1, ZB_COMPRESSED = 2 };
enum ZstdLitType : uint8_t { ZL_RAW = 0, ZL_RLE = 1, ZL_HUFFMAN = 2 };

struct ZstdSeqEntry { uint32_t base_value; uint16_t next_base; uint8_t extra_bits, nbits; };   // one state of a sequence table (8 bytes)

struct ZstdHufDesc {       // code length per symbol (the implied last weight resolved by the index pass)
  uint8_t bits[256];
  uint32_t nsym, max_bits;
};
struct ZstdFseDesc {       // normalised counts of one sequence table (-1 = "less than one"), or its RLE symbol
  int16_t norm[64];
  uint8_t log, nsym, rle, rle_sym;
  uint32_t pad;
};

struct ZstdBlock {         // one block of a frame
  uint64_t src;            // block content (behind the 3-byte block header), in HBM
  uint64_t lit;            // literals: raw -> their address inside src; Huffman -> the block's literal buffer; RLE -> the byte
  uint64_t seq;            // the block's sequence records (uint4 each)
  uint32_t src_len;        // content bytes (RLE: 1)
  uint32_t out_len;        // raw / RLE: bytes the block regenerates; compressed: written by the entropy pass
  uint32_t regen;          // literal bytes
  uint32_t nseq;
  uint32_t huf_off, huf_len;     // Huffman streams (jump table first when there are four) relative to src
  uint32_t bits_off, bits_len;   // sequence bit stream relative to src
  uint32_t huf;            // index of the ZstdHufDesc in force
  uint32_t tab[3];         // index of the ZstdFseDesc in force: literal lengths, offsets, match lengths
  uint8_t type, lit_type, lit_streams, first_in_frame;
  // ---- written by the entropy pass ----
  uint32_t direct;         // the page holds nothing but such blocks: `lit` is the block's place in the page's output, the execute pass has nothing to do
  uint32_t page_off;       // ... and this is its offset there
  uint32_t pad1;
  uint32_t lit_used;       // literal bytes the sequences consume
  uint32_t bad;
  uint32_t pad;
};

struct ZstdStream {        // one compressed page (or dictionary page): frames back to back
  uint64_t dst;
  uint32_t uncomp_size;
  uint32_t first_block, n_blocks;
  uint32_t direct;         // every block is a compressed block of Huffman literals without sequences: decoded in place by the entropy pass
};

// ---- bit streams ---------------------------------------------------------------------------------------------------------------------------
PLX_HD int z_hb(uint32_t v) { return 31 - __builtin_clz(v); }      // v != 0

// eight bytes at p + byte; bytes outside [0, n) read as zero
PLX_HD uint64_t z_ld64(const uint8_t* p, uint32_t n, int64_t byte) {
  if (byte >= 0 && (uint64_t)byte + 8 <= (uint64_t)n) return load_u64(p + byte);
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) {
    const int64_t b = byte + i;
    if (b >= 0 && b < (int64_t)n) v |= (uint64_t)p[b] << (8 * i);
  }
  return v;
}
// the 57 bits below bit `off` of a backward stream (bit i of the result = stream bit off - 57 + i; bits outside the stream read as zero)
PLX_HD uint64_t z_window(const uint8_t* p, uint32_t n, int64_t off) {
  const int64_t lo = off - 57;
  return (z_ld64(p, n, lo >> 3) >> (int)(lo & 7)) & (((uint64_t)1 << 57) - 1);
}
// `nb` bits of a window, `used` bits below its top already handed out
PLX_HD uint32_t z_field(uint64_t w, uint32_t used, uint32_t nb) { return (uint32_t)(w >> (57 - used - nb)) & (uint32_t)(((uint64_t)1 << nb) - 1); }

// first bit position of a backward stream (the bits are [0, off)); -1: no end mark
PLX_HD int64_t z_back_start(const uint8_t* p, uint32_t n) {
  if (n == 0 || p[n - 1] == 0) return -1;
  return (int64_t)n * 8 - (8 - z_hb(p[n - 1]));
}

// ---- entropy pass --------------------------------------------------------------------------------------------------------------------------
struct ZstdEntropyShared {
  uint16_t huf[2048];              // (code length << 8) | symbol, indexed by the next max_bits bits
  uint16_t huf_start[256];         // first table index of a symbol's codes
  ZstdSeqEntry ll[512], of[256], ml[512];
  union {
    struct {
      uint8_t sym[3][512];         // FSE build: symbol of a state
      uint16_t cnt[3][64];         // FSE build: occurrences of a symbol so far
    };
    uint64_t stage[kZStageWords + 3];   // (after the build) sequence bit stream: words stage_base .. of the stream (words below the stream's start are zero; one word of slack above)
  };
  int32_t stage_base, stage_top;   // word index of stage[0]; word that holds the next bit to read
  uint32_t seq_more;
  uint32_t bad;
};

// what a block WITHOUT sequences needs of it
struct ZstdHufShared {
  uint16_t huf[2048];
  uint16_t huf_start[256];
  uint32_t bad;
};

// sequence code -> base value / extra bits (RFC 8878 3.1.1.3.2.1.1)
PLX_HD uint32_t z_ll_base(uint32_t c) { return c < 16 ? c : c < 20 ? 16 + 2 * (c - 16) : c < 22 ? 24 + 4 * (c - 20) : c < 24 ? 32 + 8 * (c - 22) : c == 24 ? 48 : (uint32_t)64 << (c - 25); }
PLX_HD uint32_t z_ll_bits(uint32_t c) { return c < 16 ? 0 : c < 20 ? 1 : c < 22 ? 2 : c < 24 ? 3 : c == 24 ? 4 : c - 19; }
PLX_HD uint32_t z_ml_bits(uint32_t c) { return c < 32 ? 0 : c < 36 ? 1 : c < 38 ? 2 : c < 40 ? 3 : c < 42 ? 4 : c == 42 ? 5 : c - 36; }
PLX_HD uint32_t z_ml_base(uint32_t c) {
  if (c < 32) return c + 3;
  if (c < 36) return 35 + 2 * (c - 32);
  if (c < 38) return 43 + 4 * (c - 36);
  if (c < 40) return 51 + 8 * (c - 38);
  if (c < 42) return 67 + 16 * (c - 40);
  if (c == 42) return 99;
  return ((uint32_t)1 << (c - 36)) + 3;
}

// lane 0: where every symbol's codes start (codes of one length are consecutive, symbols ascending, longest codes first: 4.2.1)
template <class SH> PLX_HD void zstd_huf_starts(SH& sh, const ZstdHufDesc& d) {
  uint32_t rank_count[13], rank_idx[13];
  const uint32_t mb = d.max_bits < 1 ? 1 : d.max_bits > 11 ? 11 : d.max_bits, nsym = d.nsym > 256 ? 256 : d.nsym;
  for (uint32_t i = 0; i <= 12; i++) rank_count[i] = 0;
  for (uint32_t s = 0; s < nsym; s++) rank_count[d.bits[s] > 12 ? 12 : d.bits[s]]++;
  rank_idx[mb] = 0;
  for (uint32_t i = mb; i >= 1; i--) rank_idx[i - 1] = rank_idx[i] + rank_count[i] * (1u << (mb - i));
  for (uint32_t s = 0; s < nsym; s++) {
    const uint32_t b = d.bits[s];
    if (b == 0 || b > mb) { sh.huf_start[s] = 0xffff; continue; }
    sh.huf_start[s] = (uint16_t)rank_idx[b];
    rank_idx[b] += 1u << (mb - b);
  }
}
// every lane: the table entries of its symbols
template <class SH> PLX_HD void zstd_huf_fill(SH& sh, const ZstdHufDesc& d, uint32_t lane, uint32_t lanes) {
  const uint32_t mb = d.max_bits < 1 ? 1 : d.max_bits > 11 ? 11 : d.max_bits, nsym = d.nsym > 256 ? 256 : d.nsym;
  for (uint32_t s = lane; s < nsym; s += lanes) {
    const uint32_t st = sh.huf_start[s];
    if (st == 0xffff) continue;
    const uint32_t b = d.bits[s], len = 1u << (mb - b);
    const uint16_t e = (uint16_t)(b << 8 | s);
    for (uint32_t k = 0; k < len && st + k < 2048; k++) sh.huf[st + k] = e;
  }
}

// one Huffman stream: p[0, n) -> out[0, out_len); false = malformed
PLX_HD bool zstd_huf_stream(const uint16_t* tbl, uint32_t mb, const uint8_t* p, uint32_t n, uint8_t* out, uint32_t out_len) {
  int64_t off = z_back_start(p, n);
  if (off < 0) return false;
  uint32_t o = 0;
  const uint32_t top = 64 - mb;
  // Five symbols (<= 55 bits) per round from a 64-bit container whose top bit is the next bit of the stream.  The sixteen bytes the NEXT round's container will be cut
  // from are loaded a round ahead -- wherever this round ends (5 .. 55 bits further down), its 64 bits lie inside [off - 119, off - 5) -- so the load's latency hides behind
  // the five dependent table reads.
  if (off >= 128 && o + 5 <= out_len) {
    int64_t a = (off - 64) >> 3;
    uint64_t lo = z_ld64(p, n, a), hi = z_ld64(p, n, a + 8);
    while (off >= 128 && o + 5 <= out_len) {
      const uint32_t s = (uint32_t)(off - 8 * a - 64) & 63;       // 0 .. 57
      uint64_t b = s ? (lo >> s) | (hi << (64 - s)) : lo;
      a = (off - 119) >> 3;
      lo = load_u64(p + a); hi = z_ld64(p, n, a + 8);
      uint32_t used = 0;
      for (int k = 0; k < 5; k++) {
        const uint32_t e = tbl[b >> top];
        out[o + k] = (uint8_t)e;
        b <<= e >> 8;
        used += e >> 8;
      }
      o += 5;
      off -= used;
    }
  }
  while (off > 0) {
    if (o >= out_len) return false;
    const uint32_t idx = z_field(z_window(p, n, off), 0, mb);
    const uint32_t e = tbl[idx];
    out[o++] = (uint8_t)e;
    off -= e >> 8;
  }
  return off == 0 && o == out_len;
}

// lanes 0 .. streams - 1: the block's Huffman-coded literals
template <class SH> PLX_HD void zstd_huf_decode(SH& sh, const ZstdBlock& blk, const ZstdHufDesc& d, uint32_t lane) {
  if (lane >= blk.lit_streams) return;
  const uint8_t* p = PQ_GPTR(const uint8_t, blk.src) + blk.huf_off;
  uint8_t* out = PQ_GPTR(uint8_t, blk.lit);
  const uint32_t mb = d.max_bits < 1 ? 1 : d.max_bits > 11 ? 11 : d.max_bits;
  bool ok;
  if (blk.lit_streams == 1) ok = zstd_huf_stream(sh.huf, mb, p, blk.huf_len, out, blk.regen);
  else {
    // jump table: three 16-bit sizes; the fourth stream takes the rest (validated by the index pass, clamped here all the same)
    if (blk.huf_len < 6) { sh.bad = 1; return; }
    const uint32_t s1 = p[0] | (uint32_t)p[1] << 8, s2 = p[2] | (uint32_t)p[3] << 8, s3 = p[4] | (uint32_t)p[5] << 8, body = blk.huf_len - 6;
    if ((uint64_t)s1 + s2 + s3 > body) { sh.bad = 1; return; }
    const uint32_t each = (blk.regen + 3) / 4;
    if ((uint64_t)each * 3 > blk.regen) { sh.bad = 1; return; }
    const uint32_t start = lane == 0 ? 0 : lane == 1 ? s1 : lane == 2 ? s1 + s2 : s1 + s2 + s3;
    const uint32_t len = lane == 0 ? s1 : lane == 1 ? s2 : lane == 2 ? s3 : body - s1 - s2 - s3;
    const uint32_t olen = lane < 3 ? each : blk.regen - 3 * each;
    ok = zstd_huf_stream(sh.huf, mb, p + 6 + start, len, out + (size_t)lane * each, olen);
  }
  if (!ok) { sh.bad = 1; ZDBG("huf stream %u failed\n", lane); }
}

// lane t < 3: sequence table t (0 literal lengths, 1 offsets, 2 match lengths) from its normalised counts (4.1.1)
PLX_HD void zstd_fse_build(ZstdEntropyShared& sh, const ZstdFseDesc& d, uint32_t t) {
  ZstdSeqEntry* tab = t == 0 ? sh.ll : t == 1 ? sh.of : sh.ml;
  const uint32_t max_log = t == 1 ? 8 : 9, max_code = t == 0 ? 35 : t == 1 ? 31 : 52;
  auto entry = [&](uint32_t c, uint32_t next_base, uint32_t nbits) {
    ZstdSeqEntry e;
    if (c > max_code) { sh.bad = 1; ZDBG("fse code %u > max t=%u\n", c, t); c = 0; }
    if (t == 0) { e.base_value = z_ll_base(c); e.extra_bits = (uint8_t)z_ll_bits(c); }
    else if (t == 2) { e.base_value = z_ml_base(c); e.extra_bits = (uint8_t)z_ml_bits(c); }
    else { e.base_value = (uint32_t)1 << c; e.extra_bits = (uint8_t)c; }
    e.next_base = (uint16_t)next_base; e.nbits = (uint8_t)nbits;
    return e;
  };
  if (d.rle) { tab[0] = entry(d.rle_sym, 0, 0); return; }
  const uint32_t log = d.log > max_log ? max_log : d.log, size = 1u << log, nsym = d.nsym > 64 ? 64 : d.nsym;
  uint8_t* sym = sh.sym[t];
  uint16_t* cnt = sh.cnt[t];
  uint32_t high = size;
  for (uint32_t s = 0; s < nsym; s++) {
    cnt[s] = (uint16_t)(d.norm[s] == -1 ? 1 : d.norm[s] > 0 ? d.norm[s] : 0);
    if (d.norm[s] == -1 && high > 0) sym[--high] = (uint8_t)s;
  }
  const uint32_t step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  uint32_t pos = 0;
  for (uint32_t s = 0; s < nsym; s++) {
    if (d.norm[s] <= 0) continue;
    for (int i = 0; i < d.norm[s]; i++) {
      sym[pos] = (uint8_t)s;
      uint32_t guard = 0;
      do { pos = (pos + step) & mask; } while (pos >= high && ++guard < 1024);
    }
  }
  if (pos != 0) { sh.bad = 1; ZDBG("fse spread pos=%u t=%u\n", pos, t); }
  for (uint32_t i = 0; i < size; i++) {
    const uint32_t s = sym[i] < nsym ? sym[i] : 0;
    const uint32_t next = cnt[s]++;
    const uint32_t nb = next ? log - (uint32_t)z_hb(next) : log;
    tab[i] = entry(s, ((next << nb) - size) & 0xffff, nb);
  }
}

// ---- repeat offsets (3.1.1.5) as maps on the history {rep0, rep1, rep2} --------------------------------------------------------------------------
// A slot of a map is an offset (bits 31:30 = 0; 0 = invalid) or "incoming rep[j] - k" ((j + 1) << 30 | k).  A sequence's map depends on its offset value and on whether
// its literal length is zero; the offset the sequence uses is slot 0 of the history AFTER its map.
struct ZstdRepMap { uint32_t s[3]; };
PLX_HD ZstdRepMap zstd_rep_identity() { ZstdRepMap m; m.s[0] = 1u << 30; m.s[1] = 2u << 30; m.s[2] = 3u << 30; return m; }
PLX_HD ZstdRepMap zstd_rep_of_sequence(uint32_t ov, uint32_t ll) {
  const uint32_t in0 = 1u << 30, in1 = 2u << 30, in2 = 3u << 30;
  ZstdRepMap m;
  if (ov > 3) { const uint32_t v = ov - 3; m.s[0] = v > kZOfMask ? 0 : v; m.s[1] = in0; m.s[2] = in1; return m; }
  const uint32_t idx = ov - 1 + (ll == 0 ? 1u : 0u);       // ov = 0 cannot come out of the offset table (base values are >= 1): idx wraps to a large value -> invalid below
  if (idx == 0) { m.s[0] = in0; m.s[1] = in1; m.s[2] = in2; }
  else if (idx == 1) { m.s[0] = in1; m.s[1] = in0; m.s[2] = in2; }
  else if (idx == 2) { m.s[0] = in2; m.s[1] = in0; m.s[2] = in1; }
  else if (idx == 3) { m.s[0] = in0 | 1; m.s[1] = in0; m.s[2] = in1; }
  else { m.s[0] = 0; m.s[1] = in0; m.s[2] = in1; }
  return m;
}
// slot x of a later map, seen through the earlier map (or through the history itself: three offsets)
PLX_HD uint32_t zstd_rep_through(uint32_t x, const uint32_t* earlier) {
  const uint32_t tag = x >> 30, k = x & kZOfMask;
  const uint32_t y = tag == 1 ? earlier[0] : tag == 2 ? earlier[1] : earlier[2];
  // still relative: the decrements add up (a batch has 64 sequences: they stay far below 2^30); an offset: it shrinks, to "invalid" at worst
  const uint32_t moved = (y >> 30) ? y + k : (y > k ? y - k : 0);
  return tag ? moved : x;
}
// first `a`, then `b`
PLX_HD ZstdRepMap zstd_rep_compose(const ZstdRepMap& a, const ZstdRepMap& b) {
  ZstdRepMap r;
  r.s[0] = zstd_rep_through(b.s[0], a.s); r.s[1] = zstd_rep_through(b.s[1], a.s); r.s[2] = zstd_rep_through(b.s[2], a.s);
  return r;
}

// lane 0's state between two stagings of the sequence bit stream
struct ZstdSeqState {
  int32_t off;                 // next bit to read (the bits are [0, off)); a block's stream has < 2^20 bits
  uint32_t sl, so, sm, i, started, bad;
  uint32_t log_ll, log_of, log_ml;
  uint32_t lit_sum;
  uint64_t match_sum;
};
// the 32 bits below bit e of the staged stream (bit 31 of the result = stream bit e - 1); e - 32 >= 64 * stage_base
PLX_HD uint32_t z_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {       // ({hi, lo} >> sh)[31:0], sh in 0 .. 31
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
PLX_HD uint32_t zstd_staged32(const uint32_t* s32, int32_t dbase, int32_t e) {
  const int32_t start = e - 32;
  const uint32_t d0 = s32[(start >> 5) - dbase], d1 = s32[(start >> 5) - dbase + 1];
  return z_alignbit(d1, d0, (uint32_t)start & 31);
}
// the 96 bits below bit `off`: v2 = bits [off - 32, off), v1 the 32 below, v0 the 32 below those; off - 96 >= 64 * stage_base
PLX_HD void zstd_staged96(const uint32_t* s32, int32_t dbase, int32_t off, uint32_t* v2, uint32_t* v1, uint32_t* v0) {
  const int32_t start = off - 96, d = (start >> 5) - dbase;
  const uint32_t sh = (uint32_t)start & 31;
  const uint32_t r0 = s32[d], r1 = s32[d + 1], r2 = s32[d + 2], r3 = s32[d + 3];
  *v0 = z_alignbit(r1, r0, sh); *v1 = z_alignbit(r2, r1, sh); *v2 = z_alignbit(r3, r2, sh);
}
// the 32 bits that end `used` (0 .. 31) bits below the top of {hi, lo}
PLX_HD uint32_t z_below(uint32_t hi, uint32_t lo, uint32_t used) { return (uint32_t)(((((uint64_t)hi << 32) | lo) << used) >> 32); }
// the top nb (0 .. 31) bits of x
PLX_HD uint32_t z_top32(uint32_t x, uint32_t nb) { return (x >> 1) >> (31 - nb); }
PLX_HD void zstd_unpack_entry(uint64_t e, uint32_t* base_value, uint32_t* next_base, uint32_t* extra_bits, uint32_t* nbits) {
  *base_value = (uint32_t)e; *next_base = (uint32_t)(e >> 32) & 0xffff; *extra_bits = (uint32_t)(e >> 48) & 0xff; *nbits = (uint32_t)(e >> 56);
}

// lane 0: sequences -> records while their bits are staged (3.1.1.3.2, 3.1.1.5); sets seq_more / stage_top for the next staging.
// A lane is a poor serial machine (an instruction every four cycles), so the loop is branch-free and 32 bits wide: three 32-bit windows per sequence (offset extra bits;
// match + literal length extra bits; the three state updates); the repeat-offset history is the execute pass' business.
PLX_HD void zstd_seq_run(ZstdEntropyShared& sh, ZstdBlock& blk, ZstdSeqState& st) {
  uint32_t* rec = PQ_GPTR(uint32_t, blk.seq);
  const int32_t base = sh.stage_base, dbase = 2 * base;
  const int32_t lo_bits = 64 * (base + 2);     // the 96 bits below a position at or above this bit are staged (the last staging: down to bit 0, zeros below)
  const uint32_t* s32 = (const uint32_t*)sh.stage;
  const uint64_t* tll = (const uint64_t*)sh.ll;
  const uint64_t* tof = (const uint64_t*)sh.of;
  const uint64_t* tml = (const uint64_t*)sh.ml;
  int32_t off = st.off;
  uint32_t sl = st.sl, so = st.so, sm = st.sm, i = st.i;
  uint32_t lit_sum = st.lit_sum, bad = st.bad;
  uint64_t match_sum = st.match_sum;
  if (!st.started && !bad && off >= lo_bits) {
    const uint32_t log_ll = st.log_ll, log_of = st.log_of, log_ml = st.log_ml;
    const uint32_t a = zstd_staged32(s32, dbase, off);
    sl = z_top32(a, log_ll); so = z_top32(a << log_ll, log_of); sm = z_top32(a << (log_ll + log_of), log_ml);
    off -= (int32_t)(log_ll + log_of + log_ml);
    st.started = 1;
  }
  const uint32_t nseq = blk.nseq;
  if (st.started) {
    while (i < nseq && off >= lo_bits) {
      uint32_t bl, nl, xl, kl, bo, no, xo, ko, bm, nm, xm, km;
      zstd_unpack_entry(tll[sl & 511], &bl, &nl, &xl, &kl);
      zstd_unpack_entry(tof[so & 255], &bo, &no, &xo, &ko);
      zstd_unpack_entry(tml[sm & 511], &bm, &nm, &xm, &km);
      xo = xo > 31 ? 31 : xo; xm &= 31; xl &= 31;       // (tables built from validated descriptions hold nothing larger)
      // ONE read of the bit stream per sequence, at an address that depends on nothing but `off` (it goes out together with the three table reads): the 96 bits below
      // `off` hold the extra bits (offset <= 31, match length + literal length <= 32) and the three state updates (<= 26) wherever they fall
      uint32_t v2, v1, v0;
      zstd_staged96(s32, dbase, off, &v2, &v1, &v0);
      const uint32_t ov = bo + z_top32(v2, xo);
      const uint32_t b = z_below(v2, v1, xo);
      const uint32_t ml = bm + z_top32(b, xm);
      const uint32_t ll = bl + z_top32(b << xm, xl);
      const uint32_t u = xo + xm + xl;                      // 0 .. 63
      const uint32_t c = z_below(u < 32 ? v2 : v1, u < 32 ? v1 : v0, u & 31);
      const bool more = i + 1 < nseq;
      sl = nl + z_top32(c, kl & 15);
      sm = nm + z_top32(c << (kl & 15), km & 15);
      so = no + z_top32(c << ((kl & 15) + (km & 15)), ko & 15);
      off -= (int32_t)u + (more ? (int32_t)((kl & 15) + (km & 15) + (ko & 15)) : 0);
#if defined(__HIP_DEVICE_COMPILE__)
      typedef uint32_t v4 __attribute__((ext_vector_type(4)));
      v4 r4 = {ll, ml, ov, 0u};
      *(v4*)(rec + 4 * (size_t)i) = r4;
#else
      rec[4 * (size_t)i + 0] = ll; rec[4 * (size_t)i + 1] = ml; rec[4 * (size_t)i + 2] = ov; rec[4 * (size_t)i + 3] = 0;
#endif
      lit_sum += ll; match_sum += ml;
      i++;
    }
  }
  if (off < 0) bad = 1;
  st.off = off; st.sl = sl; st.so = so; st.sm = sm; st.i = i;
  st.lit_sum = lit_sum; st.match_sum = match_sum; st.bad = bad;
  if (bad || i >= nseq || lo_bits <= 0) {          // (lo_bits == 0: the stream's first word was staged -- nothing is left to stage)
    // finished (or stuck: a stream that ends early)
    if (i < nseq || off != 0 || lit_sum > blk.regen || match_sum > ((uint64_t)1 << 31)) { bad = 1; ZDBG("seq end: i=%u nseq=%u off=%d lit_sum=%u regen=%u\n", i, nseq, off, lit_sum, blk.regen); }
    blk.lit_used = lit_sum;
    blk.out_len = bad ? 0 : (uint32_t)(blk.regen + match_sum);
    if (bad) sh.bad = 1;
    sh.seq_more = 0;
  } else {
    sh.seq_more = 1;
    sh.stage_top = (off - 1) >> 6;
  }
}

// kZGroups compressed blocks, one wavefront: group g (lanes 16 g .. 16 g + 15) takes block order[first + g].  The chains of a block run on single lanes (Huffman
// streams: sub-lanes 0..3, sequences: sub-lane 0) and an instruction costs the wavefront the same whether one lane or four are live, so four blocks share the stream.
template <class W> PLX_HD void zstd_entropy_group(W& w, ZstdEntropyShared* shs, ZstdBlock* blocks, const uint32_t* order, uint32_t first, uint32_t n, const ZstdHufDesc* hufs,
                                                  const ZstdFseDesc* fses) {
  auto block_of = [&](uint32_t lane) -> ZstdBlock* { const uint32_t g = lane / kZGroupLanes; return first + g < n ? &blocks[order[first + g]] : nullptr; };
  w.lanes([&](uint32_t lane) { if (lane % kZGroupLanes == 0) { ZstdEntropyShared& sh = shs[lane / kZGroupLanes]; sh.bad = 0; sh.seq_more = 0; } });
  w.sync();
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk && blk->lit_type == ZL_HUFFMAN && lane % kZGroupLanes == 0) zstd_huf_starts(shs[lane / kZGroupLanes], hufs[blk->huf]);
  });
  w.sync();
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk && blk->lit_type == ZL_HUFFMAN) zstd_huf_fill(shs[lane / kZGroupLanes], hufs[blk->huf], lane % kZGroupLanes, kZGroupLanes);
  });
  w.sync();
  w.tick(0);
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk && blk->lit_type == ZL_HUFFMAN) zstd_huf_decode(shs[lane / kZGroupLanes], *blk, hufs[blk->huf], lane % kZGroupLanes);
  });
  w.sync();
  w.tick(1);
  ZLaneVar<ZstdSeqState> st;
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (!blk) return;
    ZstdEntropyShared& sh = shs[lane / kZGroupLanes];
    const uint32_t sub = lane % kZGroupLanes;
    if (blk->nseq) {
      if (sub < 3) zstd_fse_build(sh, fses[blk->tab[sub]], sub);
      if (sub == 0) {
        const uint8_t* bits = PQ_GPTR(const uint8_t, blk->src) + blk->bits_off;
        ZstdSeqState& s0 = st[lane];
        const int64_t start = blk->bits_len <= kZBlockMax ? z_back_start(bits, blk->bits_len) : -1;
        s0.off = (int32_t)start;
        s0.sl = s0.so = s0.sm = 0; s0.i = 0; s0.started = 0; s0.bad = start < 0 ? 1 : 0;
        s0.log_ll = fses[blk->tab[0]].rle ? 0 : fses[blk->tab[0]].log; s0.log_of = fses[blk->tab[1]].rle ? 0 : fses[blk->tab[1]].log;
        s0.log_ml = fses[blk->tab[2]].rle ? 0 : fses[blk->tab[2]].log;
        s0.lit_sum = 0; s0.match_sum = 0;
        sh.seq_more = 1;
        sh.stage_top = start >= 1 ? (int32_t)((start - 1) >> 6) : 0;
      }
    } else if (sub == 0) { blk->lit_used = 0; blk->out_len = blk->regen; }
  });
  w.tick(2);
  for (;;) {
    w.sync();
    uint32_t any = 0;
    for (uint32_t g = 0; g < kZGroups; g++) any |= shs[g].seq_more;
    if (!any) break;
    w.sync();
    // per group with sequences left: the next kZStageWords words of its bit stream (downwards from the word of the next bit), two more below them; words below the
    // stream's start are zero
    w.lanes([&](uint32_t lane) {
      ZstdBlock* blk = block_of(lane);
      ZstdEntropyShared& sh = shs[lane / kZGroupLanes];
      if (!blk || !sh.seq_more) return;
      const uint8_t* bits = PQ_GPTR(const uint8_t, blk->src) + blk->bits_off;
      const int32_t top = sh.stage_top, lo_word = top >= (int32_t)kZStageWords ? top - (int32_t)kZStageWords + 1 : 0, base = lo_word - 2;
      for (uint32_t k = lane % kZGroupLanes; k < kZStageWords + 2; k += kZGroupLanes) {
        const int64_t word = (int64_t)base + k;
        if (word <= top) sh.stage[k] = z_ld64(bits, blk->bits_len, word * 8);
      }
    });
    w.sync();
    w.lanes([&](uint32_t lane) {
      ZstdEntropyShared& sh = shs[lane / kZGroupLanes];
      if (lane % kZGroupLanes == 0 && sh.seq_more) {
        const int32_t top = sh.stage_top;
        sh.stage_base = (top >= (int32_t)kZStageWords ? top - (int32_t)kZStageWords + 1 : 0) - 2;
        sh.stage[kZStageWords + 2] = 0;
      }
    });
    w.sync();
    w.tick(3);
    w.lanes([&](uint32_t lane) {
      ZstdBlock* blk = block_of(lane);
      ZstdEntropyShared& sh = shs[lane / kZGroupLanes];
      if (blk && lane % kZGroupLanes == 0 && sh.seq_more) zstd_seq_run(sh, *blk, st[lane]);
    });
    w.tick(4);
  }
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk && lane % kZGroupLanes == 0) { blk->bad = shs[lane / kZGroupLanes].bad; w.count(5, blk->nseq); }
  });
}

// kZHufGroups blocks WITHOUT sequences, one wavefront: group g (lanes 4 g .. 4 g + 3) takes block order[first + g] -- its Huffman table, then its (up to four) streams
template <class W> PLX_HD void zstd_huf_group(W& w, ZstdHufShared* shs, ZstdBlock* blocks, const uint32_t* order, uint32_t first, uint32_t n, const ZstdHufDesc* hufs) {
  auto block_of = [&](uint32_t lane) -> ZstdBlock* { const uint32_t g = lane / kZHufGroupLanes; return first + g < n ? &blocks[order[first + g]] : nullptr; };
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk && lane % kZHufGroupLanes == 0) { ZstdHufShared& sh = shs[lane / kZHufGroupLanes]; sh.bad = 0; zstd_huf_starts(sh, hufs[blk->huf]); }
  });
  w.sync();
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk) zstd_huf_fill(shs[lane / kZHufGroupLanes], hufs[blk->huf], lane % kZHufGroupLanes, kZHufGroupLanes);
  });
  w.sync();
  w.tick(0);
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk) zstd_huf_decode(shs[lane / kZHufGroupLanes], *blk, hufs[blk->huf], lane % kZHufGroupLanes);
  });
  w.sync();
  w.tick(1);
  w.lanes([&](uint32_t lane) {
    ZstdBlock* blk = block_of(lane);
    if (blk && lane % kZHufGroupLanes == 0) { blk->lit_used = 0; blk->out_len = blk->regen; blk->bad = shs[lane / kZHufGroupLanes].bad; }
  });
}

// ---- execute pass --------------------------------------------------------------------------------------------------------------------------
struct ZstdExecShared {
  alignas(16) uint8_t ring[kZRing];
  alignas(16) uint32_t b_m[kZLanes + 4][4];                  // per sequence of the batch: {match start relative to the batch, offset, match length, one-step flag} (+ a group of slack)
  uint32_t b_ll[kZLanes], b_ml[kZLanes], b_of[kZLanes];      // the batch's records (offsets resolved)
  uint32_t b_r0[kZLanes], b_r1[kZLanes], b_r2[kZLanes];      // repeat-offset maps: the sequence's own, then (scan) of the batch up to and including it
  uint32_t b_lit[kZLanes], b_out[kZLanes];                   // exclusive prefixes: literal bytes / output bytes before the sequence
  uint32_t b_flag[kZLanes];                                  // the sequence does not fit the fast path (or the batch)
  uint32_t b_rowlit[4 * kZLanes];                            // row path: literal bytes at the head of each row of the batch
  uint32_t b_src[kZLanes];                                   // resolve-then-copy: where the match's bytes really come from (an output position)
  int32_t b_par[kZLanes];                                    // ... the earlier match of the batch whose bytes it copies, or -1
  uint32_t bad;
};

// the state of a page's wavefront (uniform: every lane holds the same values)
struct ZstdExecState {
  uint8_t* dst;
  uint32_t cap;            // bytes the page must decode to
  uint32_t cur;            // output position
  uint32_t flushed;        // ring bytes below this position are in HBM (a multiple of 16)
  uint32_t frame_start;
  uint32_t rep[3];
};

// sixteen bytes ring -> HBM (both addresses are multiples of 16: the page's output starts on one)
PLX_HD void z_copy16(uint8_t* dst, const uint8_t* src) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  *(v4*)dst = *(const v4*)src;
#else
  memcpy(dst, src, 16);
#endif
}
// ring -> HBM: whole 16-byte units below `to`; with `tail` (the page's end) the last bytes too
template <class W> PLX_HD void zstd_flush(W& w, ZstdExecShared& sh, ZstdExecState& st, uint32_t to, bool tail) {
  const uint32_t from = st.flushed, end16 = to & ~15u;
  if (end16 <= from && !tail) return;
  uint8_t* dst = st.dst;
  w.lanes([&](uint32_t lane) {
    for (uint32_t a = from + lane * 16; a < end16; a += kZLanes * 16) z_copy16(dst + a, &sh.ring[a & kZRingMask]);
    if (tail)
      for (uint32_t a = (end16 > from ? end16 : from) + lane; a < to; a += kZLanes) dst[a] = sh.ring[a & kZRingMask];
  });
  st.flushed = tail ? to : (end16 > from ? end16 : from);
}
// before `span` more bytes enter the ring: what they overwrite must be in HBM
template <class W> PLX_HD void zstd_room(W& w, ZstdExecShared& sh, ZstdExecState& st, uint32_t span) {
  if (st.cur + span - st.flushed > kZRing) { zstd_flush(w, sh, st, st.cur, false); w.sync(); }
}

// n literal bytes (from src, or n copies of `fill`) at the output position, in pieces
template <class W> PLX_HD bool zstd_emit_literals(W& w, ZstdExecShared& sh, ZstdExecState& st, const uint8_t* src, uint32_t n, bool rle, uint8_t fill) {
  if (n > st.cap - st.cur) return false;
  while (n) {
    const uint32_t piece = n < kZPiece ? n : kZPiece;
    zstd_room(w, sh, st, piece);
    const uint32_t cur = st.cur;
    w.lanes([&](uint32_t lane) {
      for (uint32_t t = lane; t < piece; t += kZLanes) sh.ring[(cur + t) & kZRingMask] = rle ? fill : src[t];
    });
    w.sync();
    st.cur += piece; n -= piece;
    if (!rle) src += piece;
  }
  return true;
}

// the bytes of one match piece: byte t of [d, d + n) = byte d - off + (t mod off): sources below `floor` have left the ring and are read from HBM (dst_addr: the page's
// output as an integer, made a global pointer where it is used: a pointer chosen between LDS and HBM would make every read a FLAT one)
PLX_HD void zstd_copy_match(ZstdExecShared& sh, uint64_t dst_addr, uint32_t d, uint32_t off, uint32_t n, uint32_t floor, uint32_t lane) {
  const bool overlap = off < n;
  for (uint32_t t = lane; t < n; t += kZLanes) {
    const uint32_t q = d - off + (overlap ? t % off : t);
    if (q >= floor) sh.ring[(d + t) & kZRingMask] = sh.ring[q & kZRingMask];
  }
  if (d - off < floor)
    for (uint32_t t = lane; t < n; t += kZLanes) {
      const uint32_t q = d - off + (overlap ? t % off : t);
      if (q < floor) sh.ring[(d + t) & kZRingMask] = PQ_GPTR(const uint8_t, dst_addr)[q];
    }
}
template <class W> PLX_HD bool zstd_emit_match(W& w, ZstdExecShared& sh, ZstdExecState& st, uint32_t off, uint32_t n) {
  if (off == 0 || off > st.cur - st.frame_start || n > st.cap - st.cur) return false;
  while (n) {
    const uint32_t piece = n < kZPiece ? n : kZPiece;
    zstd_room(w, sh, st, piece);
    const uint32_t cur = st.cur, floor = cur + piece > kZRing ? cur + piece - kZRing : 0;
    const uint64_t dst = (uint64_t)st.dst;
    w.lanes([&](uint32_t lane) { zstd_copy_match(sh, dst, cur, off, piece, floor, lane); });
    w.sync();
    st.cur += piece; n -= piece;
  }
  return true;
}

// one compressed block
template <class W> PLX_HD bool zstd_exec_block(W& w, ZstdExecShared& sh, ZstdExecState& st, const ZstdBlock& blk) {
  const bool lit_rle = blk.lit_type == ZL_RLE;
  const uint8_t fill = (uint8_t)blk.lit;
  const uint8_t* lits = lit_rle ? nullptr : PQ_GPTR(const uint8_t, blk.lit);
  const uint32_t* rec = PQ_GPTR(const uint32_t, blk.seq);
  if (blk.bad || blk.lit_used > blk.regen) { ZDBG("exec: block flagged bad=%u lit_used=%u regen=%u\n", blk.bad, blk.lit_used, blk.regen); return false; }
  uint32_t lp = 0;
  const uint32_t nseq = blk.nseq;
  struct Rec { uint32_t ll, ml, of; };
  ZLaneVar<Rec> nxt;             // the next batch's records: loaded while the current batch's matches run
  auto fetch = [&](uint32_t first, uint32_t lane) {
    Rec r = {0, 0, 0};
    if (first + lane < nseq) {
      const uint32_t* q = rec + 4 * (size_t)(first + lane);
#if defined(__HIP_DEVICE_COMPILE__)
      typedef uint32_t v4 __attribute__((ext_vector_type(4)));
      const v4 v = *(const v4*)q;
      r.ll = v.x; r.ml = v.y; r.of = v.z;
#else
      r.ll = q[0]; r.ml = q[1]; r.of = q[2];
#endif
    }
    return r;
  };
  w.lanes([&](uint32_t lane) { nxt[lane] = fetch(0, lane); if (lane == 0) sh.bad = 0; });
  for (uint32_t base = 0; base < nseq;) {
    const uint32_t n_in = nseq - base < kZLanes ? nseq - base : kZLanes;
    w.lanes([&](uint32_t lane) {
      const Rec r = nxt[lane];
      sh.b_ll[lane] = r.ll; sh.b_ml[lane] = r.ml;
      sh.b_lit[lane] = lane < n_in ? r.ll : 0; sh.b_out[lane] = lane < n_in ? r.ll + r.ml : 0;
      const ZstdRepMap m = lane < n_in ? zstd_rep_of_sequence(r.of, r.ll) : zstd_rep_identity();
      sh.b_r0[lane] = m.s[0]; sh.b_r1[lane] = m.s[1]; sh.b_r2[lane] = m.s[2];
    });
    w.sync();
    w.batch_scan(sh.b_lit, sh.b_out, sh.b_r0, sh.b_r1, sh.b_r2);       // exclusive sums of the two lengths, inclusive composition of the repeat-offset maps
    w.sync();
    const uint32_t rep_in[3] = {st.rep[0], st.rep[1], st.rep[2]};
    // the batch ends in front of the first sequence that does not fit: a long literal run / match (cooperative copies below), the batch's span, the literal buffer, the page
    const uint32_t lit_left = blk.regen - lp, room_left = st.cap - st.cur;
    w.lanes([&](uint32_t lane) {
      const uint32_t ll = sh.b_ll[lane], ml = sh.b_ml[lane], le = sh.b_lit[lane], oe = sh.b_out[lane];
      sh.b_of[lane] = zstd_rep_through(sh.b_r0[lane], rep_in);         // the history behind the sequence, slot 0
      sh.b_flag[lane] = lane >= n_in || ll > kZLitLane || ml > kZPiece || oe + ll + ml > kZBatchSpan || le + ll > lit_left || oe + ll + ml > room_left;
    });
    w.sync();
    const uint32_t cnt = w.first_flag(sh.b_flag);
    w.tick(0);
    if (cnt == 0) {
      // a sequence with a long literal run or a long match: cooperative copies, one piece at a time
      const uint32_t ll = sh.b_ll[0], ml = sh.b_ml[0], off = sh.b_of[0];
      w.sync();
      w.lanes([&](uint32_t lane) { nxt[lane] = fetch(base + 1, lane); });
      if (ll > blk.regen - lp) return false;
      if (!zstd_emit_literals(w, sh, st, lit_rle ? nullptr : lits + lp, ll, lit_rle, fill)) return false;
      lp += ll;
      if (!zstd_emit_match(w, sh, st, off, ml)) return false;
      { const uint32_t r1 = zstd_rep_through(sh.b_r1[0], rep_in), r2 = zstd_rep_through(sh.b_r2[0], rep_in); st.rep[0] = off; st.rep[1] = r1; st.rep[2] = r2; }
      base += 1;
      w.tick(4);
      continue;
    }
    const uint32_t span = sh.b_out[cnt - 1] + sh.b_ll[cnt - 1] + sh.b_ml[cnt - 1], lit_span = sh.b_lit[cnt - 1] + sh.b_ll[cnt - 1];
    zstd_room(w, sh, st, span);
    w.tick(1);
    const uint32_t cur = st.cur, frame_start = st.frame_start, floor = cur + span > kZRing ? cur + span - kZRing : 0;
    // every lane: the next batch's records on their way; its sequence's literals at their final place; its match must stay inside the frame
    w.lanes([&](uint32_t lane) {
      nxt[lane] = fetch(base + cnt, lane);
      if (lane < cnt) {
        const uint32_t ll = sh.b_ll[lane], rel = sh.b_out[lane], pos = cur + rel, lo = lp + sh.b_lit[lane];
        if (lit_rle) { for (uint32_t t = 0; t < ll; t++) sh.ring[(pos + t) & kZRingMask] = fill; }
        else if (ll && ll <= 8 && blk.regen - lo >= 8) {
          uint64_t v = load_u64(lits + lo);                    // one load for the common short run
          for (uint32_t t = 0; t < ll; t++) { sh.ring[(pos + t) & kZRingMask] = (uint8_t)v; v >>= 8; }
        } else { for (uint32_t t = 0; t < ll; t++) sh.ring[(pos + t) & kZRingMask] = lits[lo + t]; }
        const uint32_t off = sh.b_of[lane];
        if (off == 0 || off > pos + ll - frame_start) { sh.bad = 1; ZDBG("exec: offset %u at %u (lane %u)\n", off, pos + ll, lane); }
        // the common match: one step of the wavefront, no overlap, its source still in the ring
        const uint32_t ml = sh.b_ml[lane];
        uint32_t kind = (ml <= kZLanes && off >= ml && pos + ll >= off && pos + ll - off >= floor) ? 1u : 0u;
        // ... and the match whose source has LEFT the ring (all of it: flushed, final): its lane fetches it from HBM now, next to the literals -- in the in-order loop every
        // such match is a trip to HBM of its own (a 16 KB ring sees them: values repeated from a few thousand rows up)
        if (!kind && ml <= kZLanes && off >= ml && pos + ll >= off && pos + ll - off + ml <= floor && sh.bad == 0) {
          const uint8_t* far = PQ_GPTR(const uint8_t, (uint64_t)st.dst) + (pos + ll - off);
          for (uint32_t t = 0; t < ml; t += 8) {
            uint64_t v = load_u64(far + t);
            for (uint32_t e = 0; e < 8 && t + e < ml; e++) { sh.ring[(pos + ll + t + e) & kZRingMask] = (uint8_t)v; v >>= 8; }
          }
          kind = 2;
        }
        sh.b_m[lane][0] = rel + ll; sh.b_m[lane][1] = off; sh.b_m[lane][2] = ml; sh.b_m[lane][3] = kind;
      }
    });
    w.sync();
    w.tick(2);
    if (sh.bad) return false;
    const uint64_t dst = (uint64_t)st.dst;
    // RESOLVE, THEN COPY.  On sorted keys every match copies what the match before it wrote (the key's high bytes, offset 8): in sequence order that is 64 LDS round trips a
    // batch.  But a match that lies inside an earlier match of the batch only repeats that match's SOURCE: follow the chain in the index domain -- the earlier match by binary
    // search over the sequence starts, then pointer doubling (source += the parent's shift, parent = the parent's parent: 6 rounds for 64) -- and when every source lies in
    // literals or in earlier output, all matches copy at once.  Taken when every match of the batch is a one-step match whose bytes come from ONE place (not across the border
    // of a literal run and a match); anything else runs in sequence order below.
    // ROWS.  PLAIN fixed-width values whose high bytes repeat (sorted keys, timestamps, small integers in eight bytes) compress to one sequence a value: a few literal
    // bytes, then a match of the rest at offset = the value's width S -- "the other columns of this row are the previous row's" (a value equal to its predecessor just makes
    // the match S bytes longer).  When every sequence of the batch is of that kind (offset = S, literal run <= S, length a multiple of S), byte (row r, column c) is the
    // literal of the LAST row r' <= r whose literal run reaches column c, or the row in front of the batch: one ballot per column and 64 rows says which rows hold a literal
    // there, a count-leading-zeros per lane finds r'.  No chain is walked: S ballots and S byte moves a lane, where the in-order loop pays 64 LDS round trips.
    bool parallel = false;
    if (cnt >= 8) {
      const uint32_t S = sh.b_m[0][1];
      const uint32_t rows = S ? span / S : 0;
      if (S >= 1 && S <= kZRowWidth && rows * S == span && rows <= 4 * kZLanes) {
        w.lanes([&](uint32_t lane) {
          const uint32_t ll = sh.b_ll[lane], len = ll + sh.b_m[lane][2];
          sh.b_flag[lane] = lane < cnt && (sh.b_m[lane][1] != S || ll > S || len % S != 0);
          for (uint32_t r = lane; r < rows; r += kZLanes) sh.b_rowlit[r] = 0;
        });
        w.sync();
        if (w.first_flag(sh.b_flag) == kZLanes) {
          w.lanes([&](uint32_t lane) { if (lane < cnt) sh.b_rowlit[sh.b_out[lane] / S] = sh.b_ll[lane]; });        // literal bytes of the row a sequence starts; the rows it continues into have none
          w.sync();
          ZLaneVar<uint32_t[4]> my;          // the literal runs of this lane's rows (row = 64 q + lane); behind the batch: "all literal", nothing to move
          w.lanes([&](uint32_t lane) { PLX_UNROLL_Z for (uint32_t q = 0; q < 4; q++) my[lane][q] = q * kZLanes + lane < rows ? sh.b_rowlit[q * kZLanes + lane] : 0xffffffffu; });
          for (uint32_t c = 0; c < S; c++) {
            int32_t last = -1;               // the last row so far with a literal in column c (-1: the row in front of the batch)
            PLX_UNROLL_Z for (uint32_t q = 0; q < 4; q++) {              // (unrolled: the lane's four row slots stay in registers)
              if (q * kZLanes >= rows) break;
              const uint64_t lit = w.ballot([&](uint32_t lane) { return q * kZLanes + lane < rows && my[lane][q] > c; });
              const int32_t carry = last;
              w.lanes([&](uint32_t lane) {
                if (my[lane][q] <= c) {
                  const uint64_t below = lit & (((uint64_t)2 << lane) - 1);
                  const int32_t from_row = below ? (int32_t)(q * kZLanes + 63u - (uint32_t)__builtin_clzll(below)) : carry;
                  const uint32_t from = cur + (uint32_t)(from_row * (int32_t)S) + c;
                  sh.ring[(cur + (q * kZLanes + lane) * S + c) & kZRingMask] = sh.ring[from & kZRingMask];
                }
              });
              if (lit) last = (int32_t)(q * kZLanes + 63u - (uint32_t)__builtin_clzll(lit));
            }
          }
          parallel = true;
          w.count(6, 1);
          ZDBG("rows batch of %u, width %u, %u rows\n", cnt, S, rows);
        }
        w.sync();
      }
    }
    if (cnt >= 8 && !parallel) {
      w.lanes([&](uint32_t lane) {
        uint32_t flag = 0, src = 0;
        int32_t par = -1;
        if (lane < cnt) {
          const uint32_t dm = cur + sh.b_m[lane][0], off = sh.b_m[lane][1], n = sh.b_m[lane][2];
          if (!sh.b_m[lane][3]) flag = 1;
          else if (sh.b_m[lane][3] == 2) src = dm;            // fetched from HBM above: nothing to do (a copy onto itself)
          else {
            src = dm - off;
            if (src + n <= cur) par = -1;                  // earlier output
            else if (src < cur) flag = 1;                  // across the batch's start
            else {
              uint32_t lo = 0, hi = lane;                  // the last sequence that starts at or below src (starts ascend; sequence 0 starts at cur <= src)
              while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (cur + sh.b_out[mid] <= src) lo = mid; else hi = mid - 1; }
              const uint32_t dmj = cur + sh.b_m[lo][0], nj = sh.b_m[lo][2];
              if (src >= dmj) { if (lo < lane && src + n <= dmj + nj) par = sh.b_m[lo][3] == 2 ? -1 : (int32_t)lo; else flag = 1; }       // inside that sequence's match (one fetched from HBM above is final already)
              else if (src + n > dmj) flag = 1;            // across its literal run and its match
            }
          }
        }
        sh.b_src[lane] = src; sh.b_par[lane] = par; sh.b_flag[lane] = flag;
      });
      w.sync();
      parallel = w.first_flag(sh.b_flag) == kZLanes;
      for (uint32_t round = 0; parallel && round < 6; round++) {
        w.lanes([&](uint32_t lane) { sh.b_flag[lane] = lane < cnt && sh.b_par[lane] >= 0; });
        w.sync();
        if (w.first_flag(sh.b_flag) == kZLanes) break;
        struct Hop { uint32_t src; int32_t par; };
        ZLaneVar<Hop> hop;
        w.lanes([&](uint32_t lane) {
          Hop h = {sh.b_src[lane], sh.b_par[lane]};
          if (lane < cnt && h.par >= 0) { const uint32_t p = (uint32_t)h.par; h.src = sh.b_src[p] + (h.src - (cur + sh.b_m[p][0])); h.par = sh.b_par[p]; }
          hop[lane] = h;
        });
        w.sync();
        w.lanes([&](uint32_t lane) { sh.b_src[lane] = hop[lane].src; sh.b_par[lane] = hop[lane].par; });
        w.sync();
      }
      if (parallel) {
        w.lanes([&](uint32_t lane) { sh.b_flag[lane] = lane < cnt && (sh.b_par[lane] >= 0 || sh.b_src[lane] < floor); });       // (a source that has left the ring: rare, in order below)
        w.sync();
        parallel = w.first_flag(sh.b_flag) == kZLanes;
      }
      if (parallel) {
        w.lanes([&](uint32_t lane) {
          if (lane < cnt) {
            const uint32_t dm = cur + sh.b_m[lane][0], n = sh.b_m[lane][2], src = sh.b_src[lane];
            for (uint32_t t = 0; t < n; t++) sh.ring[(dm + t) & kZRingMask] = sh.ring[(src + t) & kZRingMask];
          }
        });
        w.count(6, 1);
        ZDBG("resolved batch of %u\n", cnt);
      }
    }
    // matches in sequence order, four at a time: the parameters of the next four are read from LDS while these four copy (a match waits for ONE thing: the bytes it reads)
    uint32_t m[4][4];
    PLX_UNROLL_Z for (int j = 0; j < 4; j++) { m[j][0] = sh.b_m[j][0]; m[j][1] = sh.b_m[j][1]; m[j][2] = sh.b_m[j][2]; m[j][3] = sh.b_m[j][3]; }
    for (uint32_t k = parallel ? cnt : 0; k < cnt; k += 4) {
      uint32_t q[4][4];
      PLX_UNROLL_Z for (int j = 0; j < 4; j++) { q[j][0] = w.uniform(m[j][0]); q[j][1] = w.uniform(m[j][1]); q[j][2] = w.uniform(m[j][2]); q[j][3] = w.uniform(m[j][3]); }
      if (k + 4 < cnt) {
        PLX_UNROLL_Z for (int j = 0; j < 4; j++) { m[j][0] = sh.b_m[k + 4 + j][0]; m[j][1] = sh.b_m[k + 4 + j][1]; m[j][2] = sh.b_m[k + 4 + j][2]; m[j][3] = sh.b_m[k + 4 + j][3]; }
      }
      PLX_UNROLL_Z for (int j = 0; j < 4; j++) {
        if (k + j < cnt) {
          const uint32_t d = cur + q[j][0], off = q[j][1], n = q[j][2];
          if (q[j][3] == 2) continue;        // fetched from HBM by its lane
          if (q[j][3]) w.lanes([&](uint32_t lane) { if (lane < n) sh.ring[(d + lane) & kZRingMask] = sh.ring[(d - off + lane) & kZRingMask]; });
          else w.lanes([&](uint32_t lane) { zstd_copy_match(sh, dst, d, off, n, floor, lane); });
          w.wave_fence();
        }
      }
    }
    w.sync();
    w.tick(3); w.count(5, cnt); w.count(7, 1);
    { const uint32_t r0 = zstd_rep_through(sh.b_r0[cnt - 1], rep_in), r1 = zstd_rep_through(sh.b_r1[cnt - 1], rep_in), r2 = zstd_rep_through(sh.b_r2[cnt - 1], rep_in);
      st.rep[0] = r0; st.rep[1] = r1; st.rep[2] = r2; }
    st.cur += span; lp += lit_span; base += cnt;
  }
  // the literals behind the last sequence
  if (!zstd_emit_literals(w, sh, st, lit_rle ? nullptr : lits + lp, blk.regen - lp, lit_rle, fill)) return false;
  w.tick(4);
  return true;
}

// one page: its blocks in order; false = malformed
template <class W> PLX_HD bool zstd_exec_stream(W& w, ZstdExecShared& sh, const ZstdStream& s, const ZstdBlock* blocks) {
  if (s.direct) {            // decoded in place by the entropy pass: only its verdict is left to collect
    for (uint32_t b = 0; b < s.n_blocks; b++) if (blocks[s.first_block + b].bad) return false;
    return true;
  }
  ZstdExecState st;
  st.dst = PQ_GPTR(uint8_t, s.dst); st.cap = s.uncomp_size; st.cur = 0; st.flushed = 0; st.frame_start = 0;
  st.rep[0] = 1; st.rep[1] = 4; st.rep[2] = 8;
  for (uint32_t b = 0; b < s.n_blocks; b++) {
    const ZstdBlock& blk = blocks[s.first_block + b];
    if (blk.first_in_frame) { st.frame_start = st.cur; st.rep[0] = 1; st.rep[1] = 4; st.rep[2] = 8; }
    bool ok;
    if (blk.type == ZB_RAW) ok = zstd_emit_literals(w, sh, st, PQ_GPTR(const uint8_t, blk.src), blk.src_len, false, 0);
    else if (blk.type == ZB_RLE) ok = zstd_emit_literals(w, sh, st, nullptr, blk.out_len, true, (uint8_t)blk.lit);
    else ok = zstd_exec_block(w, sh, st, blk);
    if (!ok) return false;
  }
  if (st.cur != st.cap) { ZDBG("exec: page ends at %u of %u\n", st.cur, st.cap); return false; }
  zstd_flush(w, sh, st, st.cur, true);
  return true;
}

}  // namespace pq
}  // namespace plx
