// parquet_snappy.hpp -- raw Snappy on the device, one 256-thread workgroup per stream (part of parquet_device.hpp: host + device bodies).
//
// A Snappy stream is a chain twice over: every element's position depends on all earlier tags, and a copy may read bytes the
// previous element produced (sorted keys: each 8-byte value copies 5-6 bytes of its predecessor at offset 8, so an "element-parallel"
// LZ decoder degenerates to one element at a time).  Measured on MI355X (2e7-row lineitem-like file, tools/parquet_bench.py): decoding
// element after element through HBM 1060 ms, through LDS 475 ms, a full tag parse on one lane + pointer jumping 257 ms, a one-LDS-read
// chain walk on one lane 139 ms, of which the walk was 85 % (in-kernel phase clock, PLX_SNAPPY_TIMING=1: 166 ns per element).  A GPU
// lane is a poor serial machine, so NOTHING here follows the chain element by element -- both chains are resolved by pointer doubling:
//
//   stage   the next kSnapWindow input bytes -> LDS
//   next    every window position b is decoded AS IF a tag started there: nl[b] = {output length, position of the following tag};
//           positions a chain cannot continue from (tag not complete in the window, end of input, a long literal) are STOP nodes
//   mark    which positions are real tags?  Those reachable from the round's first tag.  Jacobi pointer doubling: sweep k holds
//           jmp[b] = the 2^k-th successor of b; a marked node marks its 2^k-th successor; log2(chain length) sweeps
//   rank    marked positions in ascending order ARE the elements in stream order: element index and output offset are prefix sums
//           over the marked positions (each thread scans its 19 consecutive positions, thread 0 scans the 256 partial sums)
//   place   every thread decodes and validates the elements of its positions -> el[rank] = {source, dst, len}; the round ends
//           (`cut`) at the first STOP node on the chain or the first element that no longer fits (kSnapRound bytes, kSnapElems)
//   point   every output byte i of the round gets a POINTER: literal byte -> its input position; copy byte -> the output byte
//           `offset` earlier (in this round: an LDS index; before it: an absolute output position).  "byte i = byte i - offset" is
//           exactly Snappy's copy semantics, overlapping copies included.
//   jump    pointer jumping in LDS: ptr[i] = ptr[ptr[i]] while ptr[i] is a byte of this round -- a dependency chain of depth d
//           resolves in log2(d) sweeps instead of d steps
//   gather  byte i is loaded from the input (literal) or from the output of EARLIER rounds in HBM (made visible by the fence that ends
//           every round) and stored; threads handle bytes i = t, t + 256, ..., so a wavefront's loads and stores are 64 consecutive bytes
//
// A literal longer than kSnapDirect is a round of its own, copied input -> output 16 bytes per thread (incompressible pages are one
// 64 KB literal per Snappy block).  Every phase is a pure function of the previous one (double-buffered jmp, monotone marks, atomic
// min), so the result does not depend on the order threads run in -- which is what lets the CPU harness execute the same bodies.
// Format: snap crate / google snappy format_description.txt, used by the reference through
// crates/polars-parquet/src/parquet/compression.rs:144-230.
#pragma once
#include <stdint.h>
#include <string.h>

// full unrolling keeps the per-position register arrays of the second-generation bodies in registers (a loop left rolled indexes them
// dynamically: scratch); the host compiler of the CPU harness does not need it
#if defined(__clang__)
#define PLX_UNROLL _Pragma("unroll")
#else
#define PLX_UNROLL
#endif

namespace plx {
namespace pq {

constexpr uint32_t kSnapLanes = 256;         // threads per stream: one workgroup of 4 wavefronts ("lane" below = thread of the workgroup)
constexpr uint32_t kSnapWindow = 4096;       // input bytes staged per round: 256 lanes x 16 B
constexpr uint32_t kSnapRound = 4096;        // output bytes per round
constexpr uint32_t kSnapElems = 1024;        // elements per round
constexpr uint32_t kSnapDirect = 512;        // literals longer than this bypass the pointer machinery
constexpr uint32_t kSnapChunk = 19;          // positions per thread in rank / place (odd: conflict-free LDS strides)
constexpr uint32_t kSnapNodes = kSnapLanes * kSnapChunk;   // 4864 >= kSnapWindow + 5 + kSnapDirect + 1: every position a chain can reach
// Sweep k marks the chain nodes at distance < 2^(k+1) from the round's first tag.  A round takes at most kSnapElems elements, and the node
// right behind them must be seen (it cuts the round): distance kSnapElems < 2^11, so 11 sweeps always suffice.
constexpr uint32_t kSnapSweeps = 11;
constexpr uint32_t kSnapKindFar = 1u << 30;  // pointer kinds (bits 31:30): 0 = byte of this round (LDS index), 1 = earlier output, 2 = input
constexpr uint32_t kSnapKindLit = 2u << 30;
constexpr uint32_t kSnapPosMask = (1u << 30) - 1;
constexpr uint32_t kSnapStop = 0xffffffffu;  // nl[] of a STOP node
constexpr uint32_t kSnapLongNode = 0xfffffffeu;   // nl[] of a literal > kSnapDirect (or with a malformed length)
constexpr uint16_t kSnapEnd = (uint16_t)kSnapNodes;   // jmp[] of a node without successor: the index of a sentinel entry behind the nodes whose successor is itself (second-generation mark)
static_assert(kSnapNodes >= kSnapWindow + 5 + kSnapDirect + 1, "chain positions must fit the node arrays");
static_assert((1u << kSnapSweeps) > kSnapElems, "the node behind a full round must get its mark");

struct SnapElem {
  uint32_t src;     // bit 31: literal, bits 0..30 = input position of its bytes; else copy offset (>= 1)
  uint16_t dst;     // first output byte, relative to the round
  uint16_t len;
};
struct SnapShared {
  alignas(16) uint8_t win[kSnapWindow + 16];
  uint32_t nl[kSnapNodes];       // element nodes: output length << 16 | position of the next tag; else kSnapStop / kSnapLongNode
  union {
    uint16_t jmp[2][kSnapNodes + 2]; // mark phase: 2^k-th successor, double-buffered; [kSnapEnd] = the sentinel
    uint32_t ptr[kSnapRound];    // point / jump / gather phases
  };
  uint8_t mark[kSnapNodes + 4];      // (+ the sentinel's byte, written and never read)
  uint32_t part_cnt[kSnapLanes + 1], part_len[kSnapLanes + 1];   // per thread chunk: marked elements, their output bytes (then exclusive prefixes)
  SnapElem el[kSnapElems];
  uint32_t cut;         // first position of the chain that is not part of this round
  uint32_t n_el;
  uint32_t direct;      // 1: the round is one long literal {direct_src, direct_len} copied input -> output
  uint32_t direct_src, direct_len;
  uint32_t in_pos;      // next unparsed input byte
  uint32_t win_pos;     // input position of win[0]
  uint32_t out_pos;     // bytes produced after this round
  uint32_t round_out0;  // bytes produced before this round
  uint32_t done;        // 1: finished, 2: error
  uint32_t bad;         // set by any lane that meets a malformed element: the round is abandoned, the stream is in error
};

PLX_HD void snap_min(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // compiler builtin: no runtime header needed
#else
  if (v < *p) *p = v;
#endif
}

// the tag at t (4 more bytes readable): literal -> kind 0, *len = its length (0xffffffff: malformed); copy -> kind 1..3, *val = offset,
// *len = 4..64; *hdr = bytes of the tag itself (literal: incl. its length bytes)
PLX_HD uint32_t snappy_tag(const uint8_t* t, uint32_t* len, uint32_t* val, uint32_t* hdr) {
  const uint32_t tag = t[0], kind = tag & 3;
  if (kind == 0) {
    const uint32_t l = tag >> 2;
    if (l < 60) { *len = l + 1; *hdr = 1; }
    else {
      const uint32_t nb = l - 59;                         // 1..4 length bytes
      uint32_t v = 0;
      for (uint32_t b = 0; b < nb; b++) v |= (uint32_t)t[1 + b] << (8 * b);
      *len = v >= kSnapPosMask ? 0xffffffffu : v + 1;
      *hdr = 1 + nb;
    }
    *val = 0;
  } else if (kind == 1) { *len = ((tag >> 2) & 7) + 4; *val = ((tag >> 5) << 8) | t[1]; *hdr = 2; }
  else if (kind == 2) { *len = (tag >> 2) + 1; *val = t[1] | ((uint32_t)t[2] << 8); *hdr = 3; }
  else { *len = (tag >> 2) + 1; *val = t[1] | ((uint32_t)t[2] << 8) | ((uint32_t)t[3] << 16) | ((uint32_t)t[4] << 24); *hdr = 5; }
  return kind;
}

// The same parse without a branch (second-generation bodies).  snappy_tag reads the bytes behind the tag inside the branch of its kind: on the device every kind
// present in a wavefront is a serial pass with its own LDS round trip (measured: 11 us of a 56 us round in `next`, as much again in `place`).  Here the five bytes a
// tag can span are fetched up front -- snappy_peek: two aligned dwords around position b, shifted into place -- and every field is a select.
PLX_HD uint64_t snappy_peek(const uint8_t* win, uint32_t b) {          // bytes b .. b + 4 in bits 0 .. 39 (b <= kSnapWindow + 4: the pad behind the window is readable)
  uint32_t w[2];
  memcpy(w, __builtin_assume_aligned(win + (b & ~3u), 4), 8);      // (win is 16-byte aligned: two dword reads)
  return (((uint64_t)w[1] << 32) | w[0]) >> (8 * (b & 3u));
}
PLX_HD uint32_t snappy_tag_x(uint64_t x, uint32_t* len, uint32_t* val, uint32_t* hdr) {
  const uint32_t tag = (uint32_t)x & 0xffu, kind = tag & 3u, l = tag >> 2, rest = (uint32_t)(x >> 8);     // rest = the four bytes behind the tag
  // literal: l < 60 -> l + 1 bytes; else l - 59 = 1..4 length bytes
  const uint32_t nb = l >= 60 ? l - 59 : 0;
  const uint32_t lv = nb == 0 ? l : (nb == 4 ? rest : rest & ((1u << (8 * (nb & 3u))) - 1u));
  const uint32_t lit_len = (nb != 0 && lv >= kSnapPosMask) ? 0xffffffffu : lv + 1;
  const uint32_t c1_len = (l & 7u) + 4, c1_val = ((tag >> 5) << 8) | (rest & 0xffu);
  *len = kind == 0 ? lit_len : (kind == 1 ? c1_len : l + 1);
  *val = kind == 0 ? 0u : (kind == 1 ? c1_val : (kind == 2 ? (rest & 0xffffu) : rest));
  *hdr = kind == 0 ? 1 + nb : (kind == 3 ? 5u : kind + 1);
  return kind;
}

PLX_HD void snappy_begin(SnapShared& sh, const DecompJob& job) {
  // preamble: uncompressed length as a varint
  const uint8_t* in = PQ_GPTR(const uint8_t, job.src);
  uint32_t pos = 0, n = 0;
  bool ok = false;
  for (uint32_t shift = 0; shift <= 28 && pos < job.comp_size; shift += 7) {
    uint32_t b = in[pos++];
    n |= (b & 0x7f) << shift;
    if (!(b & 0x80)) { ok = true; break; }
  }
  sh.n_el = 0; sh.direct = 0; sh.direct_src = 0; sh.direct_len = 0; sh.in_pos = pos; sh.win_pos = pos; sh.out_pos = 0; sh.round_out0 = 0; sh.bad = 0;
  sh.cut = kSnapStop;
  // positions carry two kind bits: streams of 1 GiB and more are refused (Parquet pages are a few MB)
  if (job.comp_size > kSnapPosMask || job.uncomp_size > kSnapPosMask) ok = false;
  sh.done = (!ok || n != job.uncomp_size) ? 2u : (n == 0 ? 1u : 0u);
}

// stage: the window, 16 input bytes per lane (the 16 pad bytes behind it are cleared by lane 0)
PLX_HD void snappy_stage(SnapShared& sh, const DecompJob& job, uint32_t lane) {
  const uint8_t* in = PQ_GPTR(const uint8_t, job.src);
  constexpr uint32_t kPieces = kSnapWindow / 16 / kSnapLanes;     // 16-byte pieces per lane (1 with 256 lanes)
  uint8_t tmp[kPieces][16];
  for (uint32_t c = 0; c < kPieces; c++) {                // all loads are issued before the first LDS write waits for one
    const uint32_t base = sh.in_pos + (c * kSnapLanes + lane) * 16;   // consecutive lanes -> consecutive pieces
    if (base + 16 <= job.comp_size) memcpy(tmp[c], in + base, 16);
    else for (uint32_t b = 0; b < 16; b++) tmp[c][b] = base + b < job.comp_size ? in[base + b] : 0;
  }
  for (uint32_t c = 0; c < kPieces; c++) memcpy(sh.win + (c * kSnapLanes + lane) * 16, tmp[c], 16);
  if (lane == 0) {
    memset(sh.win + kSnapWindow, 0, 16);
    sh.win_pos = sh.in_pos; sh.round_out0 = sh.out_pos; sh.direct = 0; sh.cut = kSnapStop;
  }
}

// next: classify the nodes lane, lane + kSnapLanes, ...: element {length, successor} or STOP; start of the mark phase
PLX_HD void snappy_next(SnapShared& sh, const DecompJob& job, uint32_t lane) {
  const uint32_t avail = job.comp_size - sh.in_pos;       // input bytes left (in_pos <= comp_size)
  for (uint32_t b = lane; b < kSnapNodes; b += kSnapLanes) {
    uint32_t v = kSnapStop;
    uint16_t j = kSnapEnd;
    if (b + 5 <= kSnapWindow && b < avail) {              // tag + up to 4 length / offset bytes inside the window, inside the input
      uint32_t len, val, hdr;
      const uint32_t kind = snappy_tag(sh.win + b, &len, &val, &hdr);
      if (kind == 0 && len > kSnapDirect) v = kSnapLongNode;
      else {
        const uint32_t nxt = b + hdr + (kind == 0 ? len : 0);
        v = (len << 16) | nxt;
        j = (uint16_t)nxt;                                // < kSnapNodes; may be a STOP node, which then ends the chain
      }
    }
    sh.nl[b] = v;
    sh.jmp[0][b] = j;
    sh.mark[b] = b == 0;
  }
}

// mark: sweep `it` of the pointer doubling; returns whether this lane still saw a node with a successor
PLX_HD bool snappy_mark(SnapShared& sh, uint32_t it, uint32_t lane) {
  const uint16_t* src = sh.jmp[it & 1];
  uint16_t* dst = sh.jmp[(it & 1) ^ 1];
  bool changed = false;
  for (uint32_t b = lane; b < kSnapNodes; b += kSnapLanes) {
    const uint16_t j = src[b];
    if (j == kSnapEnd) { dst[b] = kSnapEnd; continue; }
    if (sh.mark[b]) sh.mark[j] = 1;                       // marks only ever land on nodes of the real chain
    dst[b] = src[j];
    changed = true;
  }
  return changed;
}

// ---- second-generation bodies of next / mark / rank (pq_snappy_kernel_v2; PLX_SNAPPY_KERNEL=2) ------------------------------------------
// The first-generation loops above touch LDS in program order: a store (which the compiler must assume may alias the next
// iteration's load -- the two halves of jmp[][], or the byte window against anything) sits between the loads of consecutive nodes, so every
// one of a lane's kSnapChunk = 19 nodes pays the full LDS latency, twice in mark (src[b], then src[src[b]]): ~290 cycles per node measured
// (profiles/r02/parquet_snappy_phase_clock.txt: mark is 38 % of the kernel).  Here all loads of a lane's nodes are issued before the first
// store -- 19 loads in flight, one wait -- which is also the weakest ordering the mark phase allows (a lane sees the marks as of the
// start of the sweep, what the any-order argument of the first generation already assumes).  Same results, checked by the CPU harness.
PLX_HD void snappy_next_v2(SnapShared& sh, const DecompJob& job, uint32_t lane) {
  const uint32_t avail = job.comp_size - sh.in_pos;
  uint64_t x[kSnapChunk];
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) { const uint32_t b = lane + k * kSnapLanes; x[k] = snappy_peek(sh.win, b < kSnapWindow ? b : kSnapWindow); }
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) {       // (the stores cannot alias the window words already in registers)
    const uint32_t b = lane + k * kSnapLanes;
    uint32_t len, val, hdr;
    const uint32_t kind = snappy_tag_x(x[k], &len, &val, &hdr);
    const bool node = b + 5 <= kSnapWindow && b < avail;      // tag + up to 4 length / offset bytes inside the window, inside the input
    const bool is_long = kind == 0 && len > kSnapDirect;
    const uint32_t nxt = b + hdr + (kind == 0 ? len : 0);
    sh.nl[b] = !node ? kSnapStop : (is_long ? kSnapLongNode : ((len << 16) | nxt));
    sh.jmp[0][b] = (!node || is_long) ? kSnapEnd : (uint16_t)nxt;      // < kSnapNodes; may be a STOP node, which then ends the chain
    sh.mark[b] = b == 0;
  }
  if (lane == 0) { sh.jmp[0][kSnapEnd] = kSnapEnd; sh.jmp[1][kSnapEnd] = kSnapEnd; }      // (the pointer phases of the last round wrote over the first)
}

PLX_HD bool snappy_mark_v2(SnapShared& sh, uint32_t it, uint32_t lane) {
  // 11 sweeps x 19 nodes a lane: this loop IS the mark phase, and it was 24 instructions a node (466 a sweep: issue-bound, 16 us of a round).  The sentinel entry
  // (its successor is itself, in both buffers) takes the "no successor" compare out of the dependent load and out of the mark store.
  const uint16_t* src = sh.jmp[it & 1];
  uint16_t* dst = sh.jmp[(it & 1) ^ 1];
  uint16_t j[kSnapChunk], jj[kSnapChunk];
  uint8_t m[kSnapChunk];
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) { const uint32_t b = lane + k * kSnapLanes; j[k] = src[b]; m[k] = sh.mark[b]; }
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) jj[k] = src[j[k]];
  uint32_t nearest = kSnapEnd;
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) {
    dst[lane + k * kSnapLanes] = jj[k];
    if (m[k]) sh.mark[j[k]] = 1;                          // (a marked node without successor marks the sentinel: nobody reads that byte)
    nearest = j[k] < nearest ? j[k] : nearest;
  }
  return nearest != kSnapEnd;
}

PLX_HD void snappy_rank_v2(SnapShared& sh, uint32_t lane) {
  uint32_t marked = 0, ends = 0;      // bit k: position k of the chunk is on the chain / is a STOP node or a long literal
  uint32_t v[kSnapChunk];
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) {
    const uint32_t b = lane * kSnapChunk + k;
    marked |= (uint32_t)(sh.mark[b] != 0) << k;
    v[k] = sh.nl[b];
    ends |= (uint32_t)(v[k] >= kSnapLongNode) << k;
  }
  const uint32_t stop = marked & ends;                                 // the chain ends at the first of these (nothing behind a STOP is marked)
  const uint32_t live = stop ? marked & ((stop & (0u - stop)) - 1u) : marked;
  uint32_t len = 0;
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) len += ((live >> k) & 1u) ? v[k] >> 16 : 0u;
  if (stop) snap_min(&sh.cut, lane * kSnapChunk + (uint32_t)__builtin_ctz(stop));
  sh.part_cnt[lane] = (uint32_t)__builtin_popcount(live); sh.part_len[lane] = len;
}

// rank, step 1: per thread chunk of kSnapChunk consecutive positions: marked elements, their output bytes; the first marked STOP
PLX_HD void snappy_rank(SnapShared& sh, uint32_t lane) {
  uint32_t cnt = 0, len = 0;
  for (uint32_t b = lane * kSnapChunk; b < (lane + 1) * kSnapChunk; b++) {
    if (!sh.mark[b]) continue;
    const uint32_t v = sh.nl[b];
    if (v >= kSnapLongNode) { snap_min(&sh.cut, b); break; }      // the chain ends here (nothing behind a STOP is marked)
    cnt++; len += v >> 16;
  }
  sh.part_cnt[lane] = cnt; sh.part_len[lane] = len;
}

// rank, step 2: lane 0 turns the partial sums into exclusive prefixes (256 independent LDS reads, a chain of adds)
PLX_HD void snappy_scan(SnapShared& sh) {
  uint32_t c = 0, l = 0;
  for (uint32_t t = 0; t < kSnapLanes; t++) {
    const uint32_t pc = sh.part_cnt[t], pl = sh.part_len[t];
    sh.part_cnt[t] = c; sh.part_len[t] = l;
    c += pc; l += pl;
  }
  sh.part_cnt[kSnapLanes] = c; sh.part_len[kSnapLanes] = l;
}

// second generation (pq_snappy_kernel_v2) of the scan: lane 0 alone walks 256 pairs in the first generation, each read waiting for the
// write before it (~100 cycles a step: ~12 us a round, two thirds of "rank_place" on the phase clock).  Three short steps instead:
// a) lanes 0..15 scan 16 pairs each (loads first, then stores) and leave their block totals, b) lane 0 scans the 16 totals, c) every
// lane adds its block's offset.  Scratch for the totals: the element array, which is dead between two rounds' place phases.
constexpr uint32_t kSnapScanBlocks = 16, kSnapScanPer = kSnapLanes / kSnapScanBlocks;
PLX_HD void snappy_scan_v2_blocks(SnapShared& sh, uint32_t lane) {
  if (lane >= kSnapScanBlocks) return;
  uint32_t* tmp = (uint32_t*)sh.el;
  uint32_t pc[kSnapScanPer], pl[kSnapScanPer];
  PLX_UNROLL
  for (uint32_t t = 0; t < kSnapScanPer; t++) { pc[t] = sh.part_cnt[lane * kSnapScanPer + t]; pl[t] = sh.part_len[lane * kSnapScanPer + t]; }
  uint32_t c = 0, l = 0;
  PLX_UNROLL
  for (uint32_t t = 0; t < kSnapScanPer; t++) {
    sh.part_cnt[lane * kSnapScanPer + t] = c; sh.part_len[lane * kSnapScanPer + t] = l;
    c += pc[t]; l += pl[t];
  }
  tmp[lane] = c; tmp[kSnapScanBlocks + lane] = l;
}
PLX_HD void snappy_scan_v2_totals(SnapShared& sh) {
  uint32_t* tmp = (uint32_t*)sh.el;
  uint32_t bc[kSnapScanBlocks], bl[kSnapScanBlocks];
  PLX_UNROLL
  for (uint32_t b = 0; b < kSnapScanBlocks; b++) { bc[b] = tmp[b]; bl[b] = tmp[kSnapScanBlocks + b]; }
  uint32_t c = 0, l = 0;
  PLX_UNROLL
  for (uint32_t b = 0; b < kSnapScanBlocks; b++) {
    tmp[2 * kSnapScanBlocks + b] = c; tmp[3 * kSnapScanBlocks + b] = l;
    c += bc[b]; l += bl[b];
  }
  sh.part_cnt[kSnapLanes] = c; sh.part_len[kSnapLanes] = l;
}
PLX_HD void snappy_scan_v2_offsets(SnapShared& sh, uint32_t lane) {
  const uint32_t* tmp = (const uint32_t*)sh.el;
  const uint32_t b = lane / kSnapScanPer;
  const uint32_t oc = tmp[2 * kSnapScanBlocks + b], ol = tmp[3 * kSnapScanBlocks + b];
  sh.part_cnt[lane] += oc; sh.part_len[lane] += ol;
}

// place: decode and validate the marked elements of this thread's positions into el[rank]; the first one that does not fit cuts the round
PLX_HD void snappy_place(SnapShared& sh, const DecompJob& job, uint32_t lane) {
  uint32_t r = sh.part_cnt[lane], d = sh.part_len[lane];
  const uint32_t win0 = sh.win_pos, round0 = sh.round_out0;
  for (uint32_t b = lane * kSnapChunk; b < (lane + 1) * kSnapChunk; b++) {
    if (!sh.mark[b]) continue;
    const uint32_t v = sh.nl[b];
    if (v >= kSnapLongNode) break;
    const uint32_t olen = v >> 16;
    if (r >= kSnapElems || d + olen > kSnapRound) { snap_min(&sh.cut, b); break; }   // monotone: everything behind it fails too
    uint32_t len, val, hdr;
    const uint32_t kind = snappy_tag(sh.win + b, &len, &val, &hdr);
    SnapElem el;
    el.dst = (uint16_t)d; el.len = (uint16_t)len;
    bool ok = len <= job.uncomp_size - (round0 + d);
    if (kind == 0) {
      const uint32_t src = win0 + b + hdr;
      ok = ok && src <= job.comp_size && len <= job.comp_size - src;
      el.src = 0x80000000u | src;
    } else {
      ok = ok && win0 + b + hdr <= job.comp_size && val >= 1 && val <= round0 + d;
      el.src = val;
    }
    if (!ok) sh.bad = 1;                                  // the round is abandoned before any pointer is formed
    sh.el[r] = el;
    r++; d += olen;
  }
}

// second generation (pq_snappy_kernel_v2).  Round 5: the loop runs over the MARKED positions of the chunk only (a wavefront makes as many trips as its busiest lane
// has elements: 8-10 of 19 positions on integer columns; the fully unrolled body over all 19 positions was 4500 instructions and 10 us of a 56 us round), the window
// bytes of the next marked position are fetched while the current one is decoded, and the tag is parsed without a branch (snappy_tag_x).  What `next` derived from
// the same bytes (element / STOP / long literal) is derived once more instead of being read back.
PLX_HD void snappy_place_v2(SnapShared& sh, const DecompJob& job, uint32_t lane) {
  uint32_t todo = 0;            // bit k: position k of the chunk is on the chain
  PLX_UNROLL
  for (uint32_t k = 0; k < kSnapChunk; k++) todo |= (uint32_t)(sh.mark[lane * kSnapChunk + k] != 0) << k;
  uint32_t r = sh.part_cnt[lane], d = sh.part_len[lane];
  const uint32_t win0 = sh.win_pos, round0 = sh.round_out0;
  const uint32_t avail = job.comp_size - win0;      // (win_pos == in_pos until finish)
  bool bad = false;
  uint32_t b_next = todo ? lane * kSnapChunk + (uint32_t)__builtin_ctz(todo) : 0;
  uint64_t x_next = snappy_peek(sh.win, b_next < kSnapWindow ? b_next : kSnapWindow);
  while (todo) {
    const uint32_t b = b_next;
    const uint64_t x = x_next;
    todo &= todo - 1;
    if (todo) {
      b_next = lane * kSnapChunk + (uint32_t)__builtin_ctz(todo);
      x_next = snappy_peek(sh.win, b_next < kSnapWindow ? b_next : kSnapWindow);
    }
    uint32_t len, val, hdr;
    const uint32_t kind = snappy_tag_x(x, &len, &val, &hdr);
    const bool node = b + 5 <= kSnapWindow && b < avail;
    if (!node || (kind == 0 && len > kSnapDirect)) break;       // a STOP node or a long literal (nl[b] >= kSnapLongNode): the chain ends here
    if (r >= kSnapElems || d + len > kSnapRound) { snap_min(&sh.cut, b); break; }   // monotone: everything behind it fails too
    const uint32_t src = win0 + b + hdr;
    const bool lit_ok = src <= job.comp_size && len <= job.comp_size - src;
    const bool copy_ok = src <= job.comp_size && val >= 1 && val <= round0 + d;
    if (!(len <= job.uncomp_size - (round0 + d) && (kind == 0 ? lit_ok : copy_ok))) bad = true;      // the round is abandoned before any pointer is formed
    SnapElem el;
    el.src = kind == 0 ? (0x80000000u | src) : val;
    el.dst = (uint16_t)d; el.len = (uint16_t)len;
    sh.el[r] = el;
    r++; d += len;
  }
  if (bad) sh.bad = 1;
}

// finish: lane 0 closes the round at `cut` (element count, bytes, where the next round starts, end of stream, a direct literal)
PLX_HD void snappy_finish(SnapShared& sh, const DecompJob& job) {
  const uint32_t c = sh.cut, win0 = sh.win_pos;
  const uint32_t avail = job.comp_size - win0;
  if (c >= kSnapNodes) { sh.done = 2; return; }           // cannot happen (every chain ends in a STOP node); never spin
  // elements and bytes in front of the cut: prefix of its chunk + the marked elements of the chunk before it
  const uint32_t t = c / kSnapChunk;
  uint32_t n = sh.part_cnt[t], bytes = sh.part_len[t];
  for (uint32_t b = t * kSnapChunk; b < c; b++)
    if (sh.mark[b]) { n++; bytes += sh.nl[b] >> 16; }
  uint32_t done = 0, next = c;
  if (c >= avail) {
    done = c == avail ? 1u : 2u;                          // a literal running past the end of the input
  } else if (sh.nl[c] == kSnapLongNode && n == 0) {
    uint32_t l, val, hdr;
    snappy_tag(sh.win + c, &l, &val, &hdr);
    const uint32_t src = win0 + c + hdr;
    if (l == 0xffffffffu || src > job.comp_size || l > job.comp_size - src || l > job.uncomp_size - sh.out_pos) { sh.done = 2; return; }
    sh.direct = 1; sh.direct_src = src; sh.direct_len = l;
    next = c + hdr + l; bytes = l;
    if (next >= avail) done = next == avail ? 1u : 2u;
  }
  if (done == 1 && sh.out_pos + bytes != job.uncomp_size) done = 2;     // the stream ends, the promised length is not reached
  if (!done && next == 0) done = 2;                       // no progress is impossible for a well-formed stream; never spin
  sh.n_el = n; sh.in_pos = win0 + next; sh.out_pos += bytes; sh.done = done;
}

// direct: the round's single long literal, 16 bytes per lane and step
PLX_HD void snappy_direct(const SnapShared& sh, const DecompJob& job, uint32_t lane) {
  const uint8_t* s = PQ_GPTR(const uint8_t, job.src) + sh.direct_src;
  uint8_t* d = PQ_GPTR(uint8_t, job.dst) + sh.round_out0;
  const uint32_t n = sh.direct_len;
  // four 16-byte pieces per lane and step, loaded before any is stored (input and output never overlap, which the compiler cannot know)
  for (uint32_t o0 = 0; o0 < n; o0 += 4 * kSnapLanes * 16) {
    uint8_t tmp[4][16];
    for (uint32_t c = 0; c < 4; c++) {
      const uint32_t o = o0 + (c * kSnapLanes + lane) * 16;
      if (o + 16 <= n) memcpy(tmp[c], s + o, 16);
      else for (uint32_t b = 0; b < 16; b++) tmp[c][b] = o + b < n ? s[o + b] : 0;
    }
    for (uint32_t c = 0; c < 4; c++) {
      const uint32_t o = o0 + (c * kSnapLanes + lane) * 16;
      if (o + 16 <= n) memcpy(d + o, tmp[c], 16);
      else for (uint32_t b = 0; o + b < n && b < 16; b++) d[o + b] = tmp[c][b];
    }
  }
}

// point: pointer of every output byte of the round
PLX_HD void snappy_point(SnapShared& sh, uint32_t lane) {
  const uint32_t n = sh.n_el, round0 = sh.round_out0;
  for (uint32_t k = lane; k < n; k += kSnapLanes) {       // element-major: no search, the element knows its bytes
    const SnapElem e = sh.el[k];
    const uint32_t end = (uint32_t)e.dst + e.len;
    if (e.src >> 31) {
      const uint32_t base = kSnapKindLit | (e.src & 0x7fffffffu);
      for (uint32_t i = e.dst; i < end; i++) sh.ptr[i] = base + (i - e.dst);
    } else {
      for (uint32_t i = e.dst; i < end; i++)
        sh.ptr[i] = e.src <= i ? i - e.src                // a byte of this round (strictly earlier: offset >= 1)
                               : kSnapKindFar | (round0 + i - e.src);   // place checked offset <= absolute position
    }
  }
}

// jump: one sweep of pointer jumping; returns whether this lane still followed a pointer into the round
PLX_HD bool snappy_jump(SnapShared& sh, uint32_t lane) {
  const uint32_t n_bytes = sh.out_pos - sh.round_out0;
  bool changed = false;
  for (uint32_t i = lane; i < n_bytes; i += kSnapLanes) {
    const uint32_t p = sh.ptr[i];
    if (p >> 30) continue;                                // resolved: input byte or earlier output
    sh.ptr[i] = sh.ptr[p];                                // p < i: any value found there is an ancestor of byte i
    changed = true;
  }
  return changed;
}

// second generation (pq_snappy_kernel_v2): the loads of all of a lane's bytes, then the dependent loads, then the stores -- ptr[] is
// updated in place, so in program order every load would wait for the store before it.  Reading ptr[p] before or after another
// update of it makes no difference to the result: whatever is found there is an ancestor of byte i (see snappy_jump).
PLX_HD bool snappy_jump_v2(SnapShared& sh, uint32_t lane) {
  const uint32_t n_bytes = sh.out_pos - sh.round_out0;
  constexpr uint32_t kPer = kSnapRound / kSnapLanes;
  uint32_t p[kPer], q[kPer];
  PLX_UNROLL
  for (uint32_t k = 0; k < kPer; k++) { const uint32_t i = lane + k * kSnapLanes; p[k] = i < n_bytes ? sh.ptr[i] : (1u << 30); }
  PLX_UNROLL
  for (uint32_t k = 0; k < kPer; k++) q[k] = sh.ptr[p[k] & (kSnapRound - 1)];      // (read for resolved pointers too -- some entry of the array -- and not used then)
  uint32_t least = 1u << 30;
  PLX_UNROLL
  for (uint32_t k = 0; k < kPer; k++) {
    if (!(p[k] >> 30)) sh.ptr[lane + k * kSnapLanes] = q[k];
    least = p[k] < least ? p[k] : least;
  }
  return !(least >> 30);
}

// gather: load every byte through its resolved pointer and store it
PLX_HD void snappy_gather(const SnapShared& sh, const DecompJob& job, uint32_t lane) {
  const uint8_t* in = PQ_GPTR(const uint8_t, job.src);
  uint8_t* gout = PQ_GPTR(uint8_t, job.dst);
  const uint32_t n_bytes = sh.out_pos - sh.round_out0, round0 = sh.round_out0;
  // Every resolved pointer leads outside the round (input, or output of earlier rounds), so the loads never alias the stores --
  // which the compiler cannot know: taken one byte at a time each load would wait for the previous store.  16 loads in flight,
  // then 16 stores.
  constexpr uint32_t kBatch = 16;
  for (uint32_t i0 = lane; i0 < n_bytes; i0 += kSnapLanes * kBatch) {
    uint8_t v[kBatch] = {};
    for (uint32_t k = 0; k < kBatch; k++) {
      const uint32_t i = i0 + kSnapLanes * k;
      if (i < n_bytes) {
        const uint32_t p = sh.ptr[i];
        const uint32_t at = p & kSnapPosMask;
        v[k] = (p >> 30) == 2 ? in[at] : gout[at];
      }
    }
    for (uint32_t k = 0; k < kBatch; k++) {
      const uint32_t i = i0 + kSnapLanes * k;
      if (i < n_bytes) gout[round0 + i] = v[k];
    }
  }
}

}  // namespace pq
}  // namespace plx
