// kernels_strgroup.hip -- group_by(<raw Utf8View key>).agg(sum / mean / count / len of ONE numeric column) without a dictionary-encode pass.
//
// The reference groups on string keys by hashing the 16-byte views (crates/polars-expr/src/hash_keys.rs:413-452 BinviewKeys;
// crates/polars-compute/src/binview_index_map.rs: an index map view -> group index that compares inline views by value).  The path of
// rounds 2-3 encoded the views to dictionary codes first (kernels_strview.hip): one random probe of a 128 MB table per row, bound by
// the ~60 G random line fetches per second the fabric delivers -- 22 ms per 1e9 rows before the group-by proper has started.  Nothing in a
// group-by needs the codes in ROW order, so this operator never builds them:
//   scatter   rows are radix-partitioned by the top 9 bits of the view's hash (the scheme of partition3_device.hpp -- rank by one LDS atomic,
//             tile sort, whole 128-B lines out, carry lines -- with 24-byte records {view, value}; the record's length dword also carries the value's
//             null flag and the next 27 hash bits); no random access to HBM at all
//   aggregate one workgroup per partition: an LDS tag table (16384 words {13-bit tag | 12-bit group index}) in front of dense per-group storage
//             {view, sum, count of valid values, rows}; a tag match is confirmed against the 16-byte view; the partition's groups -- each with
//             its view -- go straight to the dense output
//   skew      up to 64 strings that a sample finds in >= 1/512 of the rows are summed in LDS cells of the scatter kernel and never become records (one string
//             with half of the rows would otherwise put half of the rows into one aggregation workgroup, all on one LDS address: 76 ms instead of 3 at 2^26 rows)
// The distinct views ARE the dictionary of the result's key column.  Fast path: inline strings (<= 12 bytes: the view is the string), no null
// keys, aggregates sum / mean / count / len, at least ~4096 distinct strings (fewer: same-address LDS updates, the encoded route is faster); anything else -> -1,
// and the caller takes the encode-then-group route.
// Measured (MI355X, 1e9 rows, 1e6 distinct 12-byte strings, f64 values): scatter 9.7 ms + aggregate 5.95 ms = 16.1 ms a step, against 31.4 ms for
// encode-then-group; neither kernel is HBM-bound (48 GB and 24 GB of traffic: 5.0 and 4.0 TB/s) -- see the notes at each kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "core.hpp"
#include "dev.hpp"
#include "kernels.hpp"
#include "scan.hpp"

namespace plx {
namespace k {

using namespace dev;

namespace {
constexpr uint32_t kSgBlock = 1024;
constexpr uint32_t kSgRows = 3;                        // rows per thread and round: tiles of 3072 rows
constexpr uint32_t kSgTile = kSgBlock * kSgRows;
constexpr uint32_t kSgRW = 6;                          // record dwords: view (4) + value (2); dword 0 = length (4 bits) | value is null << 4 | 27 hash bits << 5
constexpr uint32_t kSgChunkRecs = 256;
constexpr uint32_t kSgChunkDw = kSgChunkRecs * kSgRW, kSgCapLines = kSgChunkDw / 32;
constexpr uint32_t kSgNoChunk = 0xffffffffu;
constexpr unsigned long long kSgEmpty = ~0ull;

__device__ __forceinline__ uint64_t sg_mix(uint64_t h, uint64_t w) { h ^= w; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; return h; }
__device__ __forceinline__ uint64_t sg_hash(uint64_t w0, uint64_t w1) { return sg_mix(sg_mix(0x9e3779b97f4a7c15ull, w0), w1) * 0x55fbfd6bfc5458e9ull; }

struct SgScatter {
  const unsigned long long* views;   // [n][2]
  const unsigned long long* values;  // [n] 8-byte values (f64 or i64 bits)
  const uint64_t* val_validity;      // may be null
  int64_t n;
  unsigned int* recs;
  unsigned int* chunk_part;
  unsigned int* chunk_fill;
  unsigned int* flags;               // [0] ran out of chunks, [1] a string longer than 12 bytes, [2] unused
  unsigned long long* timing;        // PLX_STRGROUP_TIMING=1: [9] 100 MHz ticks of thread 0 summed over workgroups, by phase; [8] rounds
  uint32_t chunks_per_wg, log2_parts;
  // heavy hitters of the sample (sg_hot_kernel): their rows never become records -- one string with half of the rows would put half of the rows into ONE
  // aggregation workgroup, every update on the same LDS address (measured: 76 ms instead of 1.6 for 2^26 rows) -- but are summed in LDS cells right here
  uint32_t n_hot;                    // <= kSgMaxHot
  const unsigned long long* hot_views;   // [n_hot][2]
  unsigned long long* hot_acc;       // [kSgMaxHot][3] sum bits, valid values, rows (global, zeroed): every workgroup adds its LDS cells at its end
  uint32_t is_f64;
};
constexpr uint32_t kSgMaxHot = 64, kSgHotSlots = 128, kSgHotCopies = 4;
constexpr size_t kSgHotLds = (size_t)kSgMaxHot * 16 + (size_t)kSgHotSlots * 4 + (size_t)kSgMaxHot * kSgHotCopies * 16;     // views | slot -> index | cells {sum, valid | rows << 32}

// the scatter's hash: three 32x32->64 multiply-adds over the string's bytes and its length (multilinear, so the high half is well mixed), folded
// and multiplied once more -- five quarter-rate instructions where a 64-bit finaliser costs a dozen; 36 bits are used (partition 9 + tag table 27)
__device__ __forceinline__ uint64_t sg_hash36(uint64_t w0, uint64_t w1) {
  uint64_t h = (uint64_t)(uint32_t)(w0 >> 32) * 0x9e3779b1u + (uint64_t)((uint32_t)w0 & 15u) * 0x85ebca6bc2b2ae35ull;
  h += (uint64_t)(uint32_t)w1 * 0xc2b2ae3du;
  h += (uint64_t)(uint32_t)(w1 >> 32) * 0x27d4eb2fu;
  const uint32_t g = (uint32_t)(h >> 32) ^ (uint32_t)h * 0x165667b1u;
  return (uint64_t)g * 0x9e3779b97f4a7c15ull;
}

// LDS: sorted [tile * 6] u32 | carry [NP][32] u32 | desc4 [NP] uint4 | cnt, off [NP + 1], lines_left, dstB, state [NP] u32 | wtot [8] | misc [4] | hot views [64][2] u64, hot slots [128] u32, hot cells [64][4][2] u64
__host__ __device__ inline size_t sg_scatter_lds(uint32_t NP) { return (size_t)kSgTile * kSgRW * 4 + (size_t)NP * 128 + (size_t)NP * 16 + ((size_t)NP * 5 + 1) * 4 + 32 + 16 + 12 + kSgHotLds; }

// One round = one tile of 3072 rows.  What shapes the schedule (measured phase by phase with PLX_STRGROUP_TIMING; the first version took 12.6 us a round):
//  * loads and stores share one counter per wave (vmcnt): rows are consumed (hashed, ranked) BEFORE the round's lines go out, one full round
//    after their loads were issued -- they have always arrived, and the stores have a round to drain;
//  * with no memory traffic at all the first version still took 8.7 us a round: the kernel is bound by VALU issue (two cycles per wave-instruction,
//    four waves per SIMD) and LDS round trips, not by HBM.  Hence: the scan is one partition per thread over eight waves (not eight per lane of
//    one wave while fifteen waves wait); it leaves the flush a ready-made descriptor per partition (one 16-byte read, no decoding arithmetic to
//    speak of); the flush fetches the words of four partitions before it acts on any; the hash is five multiply-adds;
//  * the copy-out: a 16-lane group per partition, four LDS instructions plus one per line beyond the first (a partition receives 36 dwords a round).
template <bool TIMING>
__global__ __launch_bounds__(kSgBlock) void strgroup_scatter_kernel(SgScatter p) {
  extern __shared__ unsigned long long sg_lds[];
  constexpr uint32_t NP = 512;
  unsigned int* sorted = reinterpret_cast<unsigned int*>(sg_lds);
  unsigned int* carry = sorted + (size_t)kSgTile * kSgRW;
  // per partition, for the flush: x = (first dword of its rows in the tile) - (carried dwords) [may be negative], y = carried dwords | new carry << 5 |
  // lines << 10 | has rows << 20 | lines continue in a newly opened chunk << 21, z = line index of its first line
  uint4* desc4 = reinterpret_cast<uint4*>(carry + (size_t)NP * 32);
  unsigned int* cnt = reinterpret_cast<unsigned int*>(desc4 + NP);
  unsigned int* off = cnt + NP;
  unsigned int* lines_left = off + NP + 1; // lines that still fit the current chunk / first line of the newly opened chunk(s): read only when bit 21 is set
  unsigned int* dstB = lines_left + NP;
  unsigned int* state = dstB + NP;         // carried dwords | lines used in the current chunk << 5 | (workgroup-local index of the current chunk + 1) << 11
  unsigned int* wtot = state + NP + 3;     // (16-byte aligned) per scan wave: rows | chunks needed << 16
  unsigned int* misc = wtot + 8;           // [0] chunks this workgroup has opened
  unsigned long long* hot_v = reinterpret_cast<unsigned long long*>(misc + 4);       // (misc is 16-byte aligned like wtot)
  unsigned int* hot_slot = reinterpret_cast<unsigned int*>(hot_v + kSgMaxHot * 2);   // hash slot -> hot index + 1 (0: empty), linear probing
  unsigned long long* hot_cell = reinterpret_cast<unsigned long long*>(hot_slot + kSgHotSlots);   // [hot][copy]{sum bits, valid values | rows << 32}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < NP; i += kSgBlock) { cnt[i] = 0; state[i] = kSgCapLines << 5; }
  if (tid < 4) misc[tid] = 0;
  if (p.n_hot) {
    for (uint32_t i = tid; i < kSgHotSlots; i += kSgBlock) hot_slot[i] = 0;
    for (uint32_t i = tid; i < kSgMaxHot * kSgHotCopies * 2; i += kSgBlock) hot_cell[i] = 0ull;
    for (uint32_t i = tid; i < p.n_hot * 2; i += kSgBlock) hot_v[i] = p.hot_views[i];
    __syncthreads();
    if (tid == 0)
      for (uint32_t i = 0; i < p.n_hot; i++) {
        uint32_t sl = (uint32_t)(sg_hash36(hot_v[i * 2], hot_v[i * 2 + 1]) >> 20) & (kSgHotSlots - 1u);
        while (hot_slot[sl]) sl = (sl + 1) & (kSgHotSlots - 1u);
        hot_slot[sl] = i + 1;
      }
  }
  __syncthreads();
  const uint32_t chunk0 = blockIdx.x * p.chunks_per_wg;
  const int64_t nrounds = (p.n + kSgTile - 1) / kSgTile;
  ulonglong2 v[kSgRows], vn[kSgRows];
  unsigned long long x[kSgRows], xn[kSgRows];
  uint32_t part[kSgRows];
  auto load = [&](int64_t rd, ulonglong2* vv, unsigned long long* xx) __attribute__((always_inline)) {
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) {
      const int64_t row = rd * kSgTile + (int64_t)j * kSgBlock + tid;
      if (row < p.n) { vv[j] = reinterpret_cast<const ulonglong2*>(p.views)[row]; xx[j] = p.values[row]; }
      else { vv[j] = make_ulonglong2(kSgEmpty, kSgEmpty); xx[j] = 0; }
    }
  };
  // rows of round rd (in v, x) -> partition and rank within (tile, partition): part = partition | rank << 10, all ones for a row that is not there
  auto rank = [&](int64_t rd) __attribute__((always_inline)) {
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) {
      const int64_t row = rd * kSgTile + (int64_t)j * kSgBlock + tid;
      bool live = row < p.n;
      const bool vnull = live && p.val_validity && !((p.val_validity[row >> 6] >> (row & 63)) & 1);
      // a null key (the view stamped with the impossible length kStrviewNullLen: plx_strview_stamp_nulls) is a key of its own -- from here on the view {15, 0}:
      // length nibble 15, which no string of the fast path has (<= 12 bytes); the result's group with that view becomes the null group
      if (live && (uint32_t)v[j].x == kStrviewNullLen) v[j] = make_ulonglong2(15ull, 0ull);
      else if (live && (uint32_t)v[j].x > 12u) { p.flags[1] = 1u; live = false; }     // a long string: the view is not the string -> the caller falls back
      part[j] = 0xffffffffu;
      uint64_t h = 0;
      if (live) {
        h = sg_hash36(v[j].x, v[j].y);
        if (p.n_hot) {
          for (uint32_t sl = (uint32_t)(h >> 20) & (kSgHotSlots - 1u);; sl = (sl + 1) & (kSgHotSlots - 1u)) {
            const uint32_t e = hot_slot[sl];
            if (!e) break;
            if (hot_v[(e - 1) * 2] == v[j].x && hot_v[(e - 1) * 2 + 1] == v[j].y) {
              unsigned long long* cell = hot_cell + ((size_t)(e - 1) * kSgHotCopies + ((uint32_t)wave & (kSgHotCopies - 1u))) * 2;
              atomicAdd(cell + 1, (1ull << 32) | (vnull ? 0ull : 1ull));
              if (!vnull) { if (p.is_f64) atomicAdd(reinterpret_cast<double*>(cell), __longlong_as_double((long long)x[j])); else atomicAdd(cell, x[j]); }
              live = false;
              break;
            }
          }
        }
      }
      if (live) {
        const uint32_t q = (uint32_t)(h >> (64 - 9));
        part[j] = q | atomicAdd(&cnt[q], 1u) << 10;
        // the record's first dword: length (4 bits) | value is null | the 27 hash bits below the partition's, which the aggregation kernel probes with
        v[j].x = (v[j].x & 0xffffffff0000000full) | (vnull ? 16u : 0u) | ((uint64_t)((uint32_t)(h >> (64 - 9 - 27)) & 0x7ffffffu) << 5);
      }
    }
  };
  unsigned long long t_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = TIMING ? wall_clock64() : 0;
#define SG_TICK(i) do { if (TIMING && tid == 0) { const unsigned long long now = wall_clock64(); t_acc[i] += now - t_last; t_last = now; } } while (0)
  int64_t rd = blockIdx.x;
  if (rd < nrounds) { load(rd, v, x); rank(rd); }
  if (rd + gridDim.x < nrounds) load(rd + gridDim.x, vn, xn);
  for (; rd < nrounds; rd += gridDim.x) {
    __syncthreads();                                                                          // A: the tile's ranks are complete, the previous flush is done with off / desc4
    SG_TICK(0);
    // scan, one partition per thread of waves 0-7
    uint32_t sc_c = 0, sc_st = 0, sc_v = 0, sc_incl = 0, sc_opened = 0;
    if (tid < (int)NP) {
      sc_c = cnt[tid]; sc_st = state[tid]; sc_opened = misc[0];
      const uint32_t nl = ((sc_st & 31u) + sc_c * kSgRW) >> 5, left = kSgCapLines - ((sc_st >> 5) & 63u);
      sc_v = sc_c | (nl > left ? (nl - left + kSgCapLines - 1) / kSgCapLines : 0u) << 16;   // rows (<= 3072 in all) and chunks to open, scanned together
      sc_incl = sc_v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)sc_incl, d, 64); if (lane >= d) sc_incl += o; }
      if (lane == 63) wtot[wave] = sc_incl;
    }
    __syncthreads();                                                                          // A2
    if (tid < (int)NP) {
      const uint4 t0 = reinterpret_cast<const uint4*>(wtot)[0], t1 = reinterpret_cast<const uint4*>(wtot)[1];
      const uint32_t wt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      uint32_t pre = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) { if (w < wave) pre += wt[w]; tot += wt[w]; }
      const uint32_t excl = pre + sc_incl - sc_v;
      const uint32_t o = excl & 0xffffu, c = sc_c, pp = (uint32_t)tid;
      uint32_t base = sc_opened + (excl >> 16);
      if (tid == (int)NP - 1) { misc[0] = sc_opened + (tot >> 16); off[NP] = tot & 0xffffu; }
      if (sc_opened + (tot >> 16) > p.chunks_per_wg) { if (tid == 0) p.flags[0] = 1u; base = 0; }      // the host raises; stay inside the workgroup's chunks
      const uint32_t cd = sc_st & 31u;
      uint32_t ln = (sc_st >> 5) & 63u, lc = sc_st >> 11;
      const uint32_t total = cd + c * kSgRW, nl = total >> 5, rem = total & 31u, left = kSgCapLines - ln;
      uint32_t y = cd | rem << 5 | nl << 10 | (c ? 1u << 20 : 0u), first_line = 0;
      off[pp] = o;
      cnt[pp] = 0;
      if (nl) {
        first_line = (chunk0 + lc - 1) * kSgCapLines + ln;
        if (nl > left) {
          const uint32_t extra = nl - left, need = (extra + kSgCapLines - 1) / kSgCapLines, first = base;
          if (lc) p.chunk_fill[chunk0 + lc - 1] = kSgChunkRecs;
          for (uint32_t e = 0; e < need; e++) { p.chunk_part[chunk0 + first + e] = pp; if (e + 1 < need) p.chunk_fill[chunk0 + first + e] = kSgChunkRecs; }
          if (left == 0) first_line = (chunk0 + first) * kSgCapLines;                         // everything goes to the new chunk(s), which are consecutive
          else { y |= 1u << 21; lines_left[pp] = left; dstB[pp] = (chunk0 + first) * kSgCapLines; }
          lc = first + need; ln = extra - (need - 1) * kSgCapLines;
        } else ln += nl;
      }
      desc4[pp] = make_uint4(o * kSgRW - cd, y, first_line, 0u);
      state[pp] = rem | ln << 5 | lc << 11;
    }
    SG_TICK(1);
    __syncthreads();                                                                          // B
    SG_TICK(2);
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) {
      if (part[j] == 0xffffffffu) continue;
      uint2* dst = reinterpret_cast<uint2*>(sorted + (size_t)(off[part[j] & 1023u] + (part[j] >> 10)) * kSgRW);
      dst[0] = make_uint2((uint32_t)v[j].x, (uint32_t)(v[j].x >> 32));
      dst[1] = make_uint2((uint32_t)v[j].y, (uint32_t)(v[j].y >> 32));
      dst[2] = make_uint2((uint32_t)x[j], (uint32_t)(x[j] >> 32));
    }
    // the next round's rows arrived a round ago: rank them now (cnt is zero again since the scan), then send for the rows after them
    SG_TICK(3);
    if (rd + gridDim.x < nrounds) {
      if (TIMING) { __builtin_amdgcn_s_waitcnt(0x0F70); SG_TICK(4); }                         // vmcnt(0): the rows' arrival, apart from the work on them
#pragma unroll
      for (uint32_t j = 0; j < kSgRows; j++) { v[j] = vn[j]; x[j] = xn[j]; }
      rank(rd + gridDim.x);
      SG_TICK(5);
    }
    __syncthreads();                                                                          // C: the tile is complete
    SG_TICK(6);
    // send for the rows after the next: the memory pipeline takes ~2 us to accept a workgroup's 96 wave-loads, and a wave whose loads are in can
    // start on its share of the flush while the others are still queueing -- ahead of barrier C everybody would wait for the last one
    if (rd + 2 * (int64_t)gridDim.x < nrounds) load(rd + 2 * (int64_t)gridDim.x, vn, xn);
    {
      // copy-out: a 16-lane group per partition (64 groups, eight partitions each, in two batches of four whose LDS reads are issued together): one
      // 16-byte descriptor read, one 8-byte read per line and lane (every offset here is even), one for the leftover, one 8-byte write into the carry
      const uint32_t g = (uint32_t)tid >> 4, l16 = (uint32_t)tid & 15u, d = l16 * 2;
#pragma unroll 1
      for (uint32_t b4 = 0; b4 < 2; b4++) {
        uint4 D[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) D[q] = desc4[g + (b4 * 4 + q) * 64u];
        uint2 w[4], r[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const uint32_t pp = g + (b4 * 4 + q) * 64u;
          const uint32_t c_dw = D[q].y & 31u, nl32 = (D[q].y >> 5) & (1023u << 5);            // lines * 32
          // dword d of the partition's stream (carry first, then its rows of the tile)
          w[q] = *reinterpret_cast<const uint2*>(d < c_dw ? carry + (size_t)pp * 32 + d : sorted + ((int)D[q].x + (int)d));
          // the new carry's dwords d, d + 1 = dwords lines * 32 + d of the stream (without a whole line the old carry stays and the rows are appended)
          const int ri = (int)D[q].x + (int)(nl32 + d);                                        // negative only where the value is not used (a dword that stays in the carry)
          r[q] = *reinterpret_cast<const uint2*>(sorted + (ri < 0 ? 0 : ri));
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const uint32_t pp = g + (b4 * 4 + q) * 64u;
          const uint32_t y = D[q].y, c_dw = y & 31u, rem = (y >> 5) & 31u, nl = (y >> 10) & 1023u;
          if (nl) {
            uint32_t left = 0xffffffffu, b = 0;
            if ((y >> 21) & 1u) { left = lines_left[pp]; b = dstB[pp]; }
            *reinterpret_cast<uint2*>(p.recs + (uint64_t)D[q].z * 32 + d) = w[q];
            for (uint32_t i = 1; i < nl; i++) {
              const uint2 ww = *reinterpret_cast<const uint2*>(sorted + ((int)D[q].x + (int)(i * 32 + d)));
              const uint64_t line = i < left ? (uint64_t)D[q].z + i : (uint64_t)b + (i - left);
              *reinterpret_cast<uint2*>(p.recs + line * 32 + d) = ww;
            }
          }
          if (((y >> 20) & 1u) && d >= (nl ? 0u : c_dw) && d < rem) *reinterpret_cast<uint2*>(carry + (size_t)pp * 32 + d) = r[q];
        }
      }
    }
    SG_TICK(7);
    if (TIMING && tid == 0) t_acc[8]++;
  }
  if (TIMING && tid == 0) { for (int i = 0; i < 9; i++) atomicAdd(&p.timing[i], t_acc[i]); }
  __syncthreads();
  for (uint32_t i = tid; i < p.n_hot * kSgHotCopies; i += kSgBlock) {
    const unsigned long long sum = hot_cell[(size_t)i * 2], cl = hot_cell[(size_t)i * 2 + 1];
    unsigned long long* g = p.hot_acc + (size_t)(i / kSgHotCopies) * 3;
    if (!cl) continue;
    if (p.is_f64) atomicAdd(reinterpret_cast<double*>(g), __longlong_as_double((long long)sum)); else atomicAdd(g, sum);
    atomicAdd(g + 1, cl & 0xffffffffull);
    atomicAdd(g + 2, cl >> 32);
  }
  for (uint32_t pp = tid; pp < NP; pp += kSgBlock) {
    const uint32_t st = state[pp], rem = st & 31u;
    uint32_t ln = (st >> 5) & 63u, lc = st >> 11;
    if (rem) {
      if (ln == kSgCapLines) {
        if (lc) p.chunk_fill[chunk0 + lc - 1] = kSgChunkRecs;
        const uint32_t local = atomicAdd(&misc[0], 1u);
        if (local >= p.chunks_per_wg) { p.flags[0] = 1u; continue; }
        p.chunk_part[chunk0 + local] = pp;
        lc = local + 1; ln = 0;
      }
      for (uint32_t i = 0; i < rem; i++) p.recs[((uint64_t)(chunk0 + lc - 1) * kSgCapLines + ln) * 32 + i] = carry[(size_t)pp * 32 + i];
    }
    if (lc) p.chunk_fill[chunk0 + lc - 1] = (ln * 32 + rem) / kSgRW;
  }
}
#undef SG_TICK

// chunk -> partition map -> per-partition chunk lists (counting sort; the same three steps as kernels_partition.hip)
__global__ __launch_bounds__(kBlock) void sg_chunk_hist_kernel(const unsigned int* __restrict__ chunk_part, int64_t n_chunks, uint32_t NP, unsigned int* __restrict__ counts) {
  extern __shared__ unsigned long long sg_lds[];
  unsigned int* h = reinterpret_cast<unsigned int*>(sg_lds);
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) { const unsigned int q = chunk_part[c]; if (q != kSgNoChunk) atomicAdd(&h[q], 1u); }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) if (h[i]) atomicAdd(&counts[i], h[i]);
}
__global__ __launch_bounds__(kBlock) void sg_chunk_place_kernel(const unsigned int* __restrict__ chunk_part, int64_t n_chunks, uint32_t NP, const unsigned long long* __restrict__ cl_off,
                                                                unsigned int* __restrict__ cursor, unsigned int* __restrict__ cl_ids) {
  extern __shared__ unsigned long long sg_lds[];
  unsigned int* h = reinterpret_cast<unsigned int*>(sg_lds);
  unsigned int* base = h + NP;
  const int64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
  const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < n_chunks ? beg + per : n_chunks;
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (int64_t c = beg + threadIdx.x; c < end; c += blockDim.x) { const unsigned int q = chunk_part[c]; if (q != kSgNoChunk) atomicAdd(&h[q], 1u); }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) { base[i] = h[i] ? atomicAdd(&cursor[i], h[i]) : 0u; h[i] = 0; }
  __syncthreads();
  for (int64_t c = beg + threadIdx.x; c < end; c += blockDim.x) {
    const unsigned int q = chunk_part[c];
    if (q == kSgNoChunk) continue;
    cl_ids[cl_off[q] + base[q] + atomicAdd(&h[q], 1u)] = (unsigned int)c;
  }
}

struct SgAgg {
  const unsigned int* recs;
  const unsigned int* chunk_fill;
  const unsigned long long* cl_off;
  const unsigned int* cl_ids;
  unsigned long long* counter;       // [0] groups written so far
  unsigned int* overflow;            // [0] 1: a partition has more groups than its LDS storage, 2: more groups than the output holds
  unsigned long long* out_views;     // [max_groups][2]
  unsigned long long* out_sum;       // [max_groups] f64 or i64 bits
  unsigned int* out_cnt;             // valid values per group
  unsigned int* out_len;             // rows per group
  uint32_t max_groups, is_f64;
  uint32_t has_nulls;                // 0: the value column has no validity bitmap -- valid values = rows, one LDS atomic less per record
};

constexpr uint32_t kSgTagSlots = 16384, kSgGroupCap = 2816;     // per partition: 64 KB of tag words + 2816 groups x 32 B = 152 KB of LDS
constexpr uint32_t kSgPending = 0xfffu;

// One workgroup per partition.  The probe loop is VALU- and LDS-issue bound, not latency bound (measured: with the probe loop 13 ms, without it
// 4.3 ms = the time of the record loads alone), so the loop body is as small as it can be:
//  * the record carries 27 bits of its view's hash (written by the scatter): no hashing here;
//  * tag table: 16384 words {13-bit tag | 12-bit group index}, at most 14 % full -- a probe is ONE 4-byte LDS read and two compares; a matching tag is
//    confirmed once against the group's 16-byte view (a false match, ~2e-5 of the lookups, just probes on);
//  * groups get dense indices in claim order (keys and cells live in [group] arrays): the output needs no compaction.
// Claim: CAS empty -> {tag | pending}; the winner takes the next group index, writes the view, then publishes {tag | index} (LDS operations of one
// wave complete in order); whoever sees {tag | pending} looks again.
// LDS: tags [16384] u32 | w0 [cap] u64 | w1 [cap] u64 | sum [cap] u64 | cnt [cap] u32 | len [cap] u32
__global__ __launch_bounds__(kSgBlock) void strgroup_agg_kernel(SgAgg a) {
  extern __shared__ unsigned long long sg_lds[];
  unsigned int* tags = reinterpret_cast<unsigned int*>(sg_lds);
  unsigned long long* w0s = sg_lds + kSgTagSlots / 2;
  unsigned long long* w1s = w0s + kSgGroupCap;
  unsigned long long* sums = w1s + kSgGroupCap;
  unsigned int* cnts = reinterpret_cast<unsigned int*>(sums + kSgGroupCap);
  unsigned int* lens = cnts + kSgGroupCap;
  __shared__ unsigned int n_groups, full;
  __shared__ unsigned long long gbase;
  const uint32_t p = blockIdx.x;
  for (uint32_t i = threadIdx.x; i < kSgTagSlots; i += blockDim.x) tags[i] = 0xffffffffu;
  for (uint32_t i = threadIdx.x; i < kSgGroupCap; i += blockDim.x) { sums[i] = 0ull; cnts[i] = 0; lens[i] = 0; }
  if (threadIdx.x == 0) { n_groups = 0; full = 0; }
  __syncthreads();
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint64_t c_beg = a.cl_off[p], c_end = a.cl_off[p + 1];
  constexpr uint32_t kPerLane = kSgChunkRecs / 64;
  static_assert(kPerLane == 4, "load_chunk unpacks four records per lane");
  uint2 ra[kPerLane][3], rb[kPerLane][3];
  uint32_t fill_a = 0, fill_b = 0;
  auto load_id = [&](uint64_t j) -> uint32_t { return a.cl_ids[j < c_end ? j : c_end - 1]; };
  // a lane takes FOUR CONSECUTIVE records of the chunk (96 bytes: six 16-byte loads) -- the order of records within a chunk means nothing to an
  // aggregation, and the memory pipeline accepts a wave-load every ~40 cycles whatever its width: twelve 8-byte loads per chunk cost twice the issue slots
  auto load_chunk = [&](uint64_t j, uint32_t id, uint2 (*r)[3], uint32_t& fill) __attribute__((always_inline)) {
    fill = j < c_end ? a.chunk_fill[id] : 0u;
    const uint4* base = reinterpret_cast<const uint4*>(a.recs + (uint64_t)id * kSgChunkDw) + (size_t)lane * 6;
    const uint4 q0 = base[0], q1 = base[1], q2 = base[2], q3 = base[3], q4 = base[4], q5 = base[5];
    r[0][0] = make_uint2(q0.x, q0.y); r[0][1] = make_uint2(q0.z, q0.w); r[0][2] = make_uint2(q1.x, q1.y);
    r[1][0] = make_uint2(q1.z, q1.w); r[1][1] = make_uint2(q2.x, q2.y); r[1][2] = make_uint2(q2.z, q2.w);
    r[2][0] = make_uint2(q3.x, q3.y); r[2][1] = make_uint2(q3.z, q3.w); r[2][2] = make_uint2(q4.x, q4.y);
    r[3][0] = make_uint2(q4.z, q4.w); r[3][1] = make_uint2(q5.x, q5.y); r[3][2] = make_uint2(q5.z, q5.w);
  };
  auto process = [&](uint2 (*r)[3], uint32_t fill) __attribute__((always_inline)) {
    uint32_t ts[kPerLane], g[kPerLane];          // tag slot; group index once found (kPending: not yet)
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      ts[u] = (r[u][0].x >> 5) & (kSgTagSlots - 1u);
      g[u] = (uint32_t)lane * 4u + u < fill ? kSgPending : 0u;
    }
    for (uint32_t it = 0; it < 4 * kSgTagSlots; it++) {
      bool all = true;
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) all = all && g[u] != kSgPending;
      if (__all(all)) break;
      uint32_t e[kPerLane];
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) if (g[u] == kSgPending) e[u] = lds_ld(&tags[ts[u]]);
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) {
        if (g[u] != kSgPending) continue;
        const uint32_t tag = r[u][0].x >> 19;                                 // the 13 hash bits above the slot's 14
        if (e[u] == 0xffffffffu) {
          if (atomicCAS(&tags[ts[u]], 0xffffffffu, tag << 12 | kSgPending) == 0xffffffffu) {
            const uint32_t idx = atomicAdd(&n_groups, 1u);
            if (idx >= kSgGroupCap) { full = 1; g[u] = 0; lds_st(&tags[ts[u]], tag << 12); continue; }
            lds_st(&w0s[idx], ((unsigned long long)r[u][0].y << 32) | (r[u][0].x & 15u));
            lds_st(&w1s[idx], ((unsigned long long)r[u][1].y << 32) | r[u][1].x);
            lds_order();                                                      // the view first, then the tag word that announces it
            lds_st(&tags[ts[u]], tag << 12 | idx);
            g[u] = idx;
          }
          // lost the race: the winner's word next round, same slot
        } else if ((e[u] >> 12) == tag) {
          const uint32_t idx = e[u] & kSgPending;
          if (idx != kSgPending) {
            lds_order();                                                      // (after the tag word that announced it)
            const unsigned long long k0 = lds_ld(&w0s[idx]), k1 = lds_ld(&w1s[idx]);
            if (k0 == (((unsigned long long)r[u][0].y << 32) | (r[u][0].x & 15u)) && k1 == (((unsigned long long)r[u][1].y << 32) | r[u][1].x)) g[u] = idx;
            else ts[u] = (ts[u] + 1) & (kSgTagSlots - 1u);                    // same tag, another string
          }
          // pending: the group's view is being written -- the same slot again next round
        } else ts[u] = (ts[u] + 1) & (kSgTagSlots - 1u);
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      if ((uint32_t)lane * 4u + u >= fill) continue;
      if (g[u] == kSgPending) { full = 1; continue; }       // the probe loop gave up on a record (cannot happen below the planned load): never drop a row silently -- the caller falls back
      atomicAdd(&lens[g[u]], 1u);
      if (!((r[u][0].x >> 4) & 1u)) {
        if (a.has_nulls) atomicAdd(&cnts[g[u]], 1u);
        const unsigned long long x = ((unsigned long long)r[u][2].y << 32) | r[u][2].x;
        if (a.is_f64) atomicAdd(reinterpret_cast<double*>(&sums[g[u]]), __longlong_as_double((long long)x));
        else atomicAdd(&sums[g[u]], x);
      }
    }
  };
  if (c_beg < c_end) {
    const uint64_t step = (uint64_t)nwaves;
    uint64_t j = c_beg + (uint64_t)wave;
    uint32_t id_next;
    load_chunk(j, load_id(j), ra, fill_a);
    id_next = load_id(j + step);
    for (;;) {
      if (j >= c_end) break;
      { const uint32_t id = id_next; id_next = load_id(j + 2 * step); load_chunk(j + step, id, rb, fill_b); } process(ra, fill_a); j += step;
      if (j >= c_end) break;
      { const uint32_t id = id_next; id_next = load_id(j + 2 * step); load_chunk(j + step, id, ra, fill_a); } process(rb, fill_b); j += step;
    }
  }
  __syncthreads();
  if (full) { if (threadIdx.x == 0) atomicExch(a.overflow, 1u); return; }
  const uint32_t ng = n_groups;
  if (threadIdx.x == 0) gbase = ng ? atomicAdd(a.counter, (unsigned long long)ng) : 0ull;
  __syncthreads();
  if (gbase + ng > a.max_groups) { if (threadIdx.x == 0) atomicExch(a.overflow, 2u); return; }
  for (uint32_t i = threadIdx.x; i < ng; i += blockDim.x) {
    const uint64_t o = gbase + i;
    a.out_views[o * 2] = w0s[i]; a.out_views[o * 2 + 1] = w1s[i];
    a.out_sum[o] = sums[i]; a.out_cnt[o] = a.has_nulls ? cnts[i] : lens[i]; a.out_len[o] = lens[i];
  }
}

// distinct views among S = 2^18 sampled rows: one 64-bit hash per row into a table of 4 S slots (a sample; hash collisions undercount by ~S / 2^64)
__global__ __launch_bounds__(kBlock) void sg_sample_kernel(const unsigned long long* __restrict__ views, int64_t n, int64_t S, unsigned long long* __restrict__ slots, uint32_t log2_cap,
                                                           unsigned int* __restrict__ res) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  // 64 evenly spaced runs of S / 64 rows (a prefix alone says nothing about clustered or sorted input; single strided rows would fetch a line each)
  const int64_t run = S / 64 > 0 ? S / 64 : 1, r = i / run, within = i % run;
  const int64_t n_runs = (S + run - 1) / run;
  const int64_t row = S >= n ? i : (int64_t)((__int128)r * (n - run) / (n_runs > 1 ? n_runs - 1 : 1)) + within;
  ulonglong2 v = reinterpret_cast<const ulonglong2*>(views)[row < n ? row : n - 1];
  if ((uint32_t)v.x == kStrviewNullLen) v = make_ulonglong2(15ull, 0ull);                 // a null key: one more distinct key (as the scatter sees it)
  else if ((uint32_t)v.x > 12u) { res[1] = 1u; return; }
  unsigned long long h = sg_hash(v.x, v.y);
  if (h == kSgEmpty) h = 0;
  const uint64_t mask = (1ull << log2_cap) - 1;
  // look before the CAS: with one string in half of the rows, 1e5 compare-and-swaps on ONE global address took 2 ms -- a plain load of a slot that already
  // holds the hash is served from the cache
  for (uint64_t sl = (h >> 20) & mask;; sl = (sl + 1) & mask) {
    unsigned long long cur = __hip_atomic_load(&slots[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kSgEmpty) {
      cur = atomicCAS(&slots[sl], kSgEmpty, h);
      if (cur == kSgEmpty) { atomicAdd(&res[0], 1u); return; }
    }
    if (cur == h) return;
  }
}
// d distinct in a sample of S rows out of n -> G = the solution of d = G (1 - exp(-S / G)) (uniform draws); -1: a long string in the sample
double sg_estimate_groups(const uint64_t* views, int64_t n) {
  const int64_t S = std::min<int64_t>(n, (int64_t)1 << 18);      // enough to tell 1e6 distinct strings (d / S = 0.88) from 2e6 (0.94) from "all distinct"
  const uint32_t log2_cap = 20;
  Buf slots = dev_alloc(8ull << log2_cap), res = dev_alloc_zero(8);
  PLX_HIP(hipMemsetAsync(slots->ptr, 0xff, 8ull << log2_cap, stream()));
  hipLaunchKernelGGL(sg_sample_kernel, dim3((unsigned)((S + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream(), (const unsigned long long*)views, n, S, slots->as<unsigned long long>(), log2_cap,
                     res->as<unsigned int>());
  PLX_HIP(hipGetLastError());
  uint32_t r[2] = {0, 0};
  d2h_sync(r, res->ptr, 8);
  if (r[1]) return -1.0;
  const double d = (double)r[0];
  if (S >= n) return d;
  if (d >= 0.999 * (double)S) return (double)n;
  double lo = d, hi = 1e15;
  for (int it = 0; it < 200; it++) { const double mid = std::sqrt(lo * hi); (mid * (1.0 - std::exp(-(double)S / mid)) < d ? lo : hi) = mid; }
  return std::min(hi, (double)n);
}
// heavy hitters: 4096 rows spread evenly over the input, counted by their 64-bit hash in an LDS table; a string with >= 1/512 of the sample (8 rows) is
// hot; the 64 most frequent of those go to the scatter.  One workgroup; res = {n_hot, hottest count, distinct hashes in the sample, sample size}
constexpr uint32_t kSgHotSample = 4096, kSgHotTable = 8192, kSgHotMinCount = kSgHotSample / 512;     // (a table as large as the sample is FULL when every string is different: 256-step probe chains)
__global__ __launch_bounds__(kSgBlock) void sg_hot_kernel(const unsigned long long* __restrict__ views, int64_t n, unsigned long long* __restrict__ hot_views, unsigned int* __restrict__ res) {
  extern __shared__ unsigned long long sg_lds[];                   // hashes [8192] u64 | counts [8192] u32 | first sample index [8192] u32 | candidates [256] u32
  unsigned long long* hs = sg_lds;
  unsigned int* hc = reinterpret_cast<unsigned int*>(hs + kSgHotTable);
  unsigned int* hrow = hc + kSgHotTable;
  unsigned int* cand = hrow + kSgHotTable;
  __shared__ unsigned int n_cand, n_distinct;
  const int64_t S = n < (int64_t)kSgHotSample ? n : (int64_t)kSgHotSample;
  for (uint32_t i = threadIdx.x; i < kSgHotTable; i += blockDim.x) { hs[i] = kSgEmpty; hc[i] = 0; }
  if (threadIdx.x == 0) { n_cand = 0; n_distinct = 0; }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < S; i += blockDim.x) {
    const int64_t row = (int64_t)((__int128)i * n / S);
    ulonglong2 v = reinterpret_cast<const ulonglong2*>(views)[row];
    if ((uint32_t)v.x == kStrviewNullLen) v = make_ulonglong2(15ull, 0ull);               // (a column that is mostly null: the null key is the heavy hitter)
    unsigned long long h = sg_hash(v.x, v.y);
    if (h == kSgEmpty) h = 0;
    uint32_t sl = (uint32_t)(h >> 20) & (kSgHotTable - 1u);
    for (uint32_t probe = 0; probe < 256; probe++, sl = (sl + 1) & (kSgHotTable - 1u)) {
      const unsigned long long old = atomicCAS(&hs[sl], kSgEmpty, h);
      if (old == kSgEmpty) { hrow[sl] = (unsigned int)i; atomicAdd(&n_distinct, 1u); }
      if (old == kSgEmpty || old == h) { atomicAdd(&hc[sl], 1u); break; }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kSgHotTable; i += blockDim.x)
    if (hc[i] >= kSgHotMinCount && S >= (int64_t)kSgHotSample / 4) { const uint32_t o = atomicAdd(&n_cand, 1u); if (o < 256) cand[o] = i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t nc = n_cand < 256u ? n_cand : 256u, top = 0;
    for (uint32_t a = 0; a < nc; a++)          // selection sort by count, descending: at most 256 entries
      for (uint32_t b = a + 1; b < nc; b++) if (hc[cand[b]] > hc[cand[a]]) { const unsigned int t = cand[a]; cand[a] = cand[b]; cand[b] = t; }
    if (nc) top = hc[cand[0]];
    if (nc > kSgMaxHot) nc = kSgMaxHot;
    for (uint32_t a = 0; a < nc; a++) {
      const int64_t row = (int64_t)((__int128)hrow[cand[a]] * n / S);
      const bool knull = (uint32_t)views[row * 2] == kStrviewNullLen;
      hot_views[a * 2] = knull ? 15ull : views[row * 2]; hot_views[a * 2 + 1] = knull ? 0ull : views[row * 2 + 1];
    }
    res[0] = nc; res[1] = top; res[2] = n_distinct; res[3] = (unsigned int)S;
  }
}
// the hot strings' groups behind the partitions' groups
__global__ void sg_hot_emit_kernel(uint32_t n_hot, const unsigned long long* __restrict__ hot_views, const unsigned long long* __restrict__ hot_acc, unsigned long long* counter, uint32_t max_groups,
                                   unsigned int* overflow, unsigned long long* out_views, unsigned long long* out_sum, unsigned int* out_cnt, unsigned int* out_len) {
  __shared__ unsigned long long base;
  if (threadIdx.x == 0) base = atomicAdd(counter, (unsigned long long)n_hot);
  __syncthreads();
  if (base + n_hot > max_groups) { if (threadIdx.x == 0) atomicExch(overflow, 2u); return; }
  const uint32_t i = threadIdx.x;
  if (i >= n_hot) return;
  out_views[(base + i) * 2] = hot_views[i * 2]; out_views[(base + i) * 2 + 1] = hot_views[i * 2 + 1];
  out_sum[base + i] = hot_acc[i * 3]; out_cnt[base + i] = (unsigned int)hot_acc[i * 3 + 1]; out_len[base + i] = (unsigned int)hot_acc[i * 3 + 2];
}
// the result's group views: the null key's group carries the view {15, 0} (length nibble 15) -> its bit of `valid` stays clear and its view becomes the empty string
__global__ __launch_bounds__(kBlock) void sg_null_group_kernel(unsigned long long* __restrict__ gviews, int64_t G, unsigned long long* __restrict__ valid, unsigned int* __restrict__ n_null) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;                        // one 64-bit validity word per thread
  if (w * 64 >= G) return;
  unsigned long long bits = 0;
  for (int b = 0; b < 64 && w * 64 + b < G; b++) {
    const int64_t i = w * 64 + b;
    if (gviews[i * 2] == 15ull && gviews[i * 2 + 1] == 0ull) { gviews[i * 2] = 0ull; atomicAdd(n_null, 1u); }
    else bits |= 1ull << b;
  }
  valid[w] = bits;
}
}  // namespace

// -> number of null-key groups (0 or 1) among the G group views strview_groupby returned; `valid` ([ceil(G / 64)] words) gets the groups' validity
int64_t strview_null_group(uint64_t* gviews, int64_t G, uint64_t* valid) {
  if (G <= 0) return 0;
  Buf cnt = dev_alloc_zero(8);
  hipLaunchKernelGGL(sg_null_group_kernel, dim3((unsigned)(((G + 63) / 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream(), (unsigned long long*)gviews, G, (unsigned long long*)valid, cnt->as<unsigned int>());
  PLX_HIP(hipGetLastError());
  uint32_t n = 0;
  d2h_sync(&n, cnt->ptr, 4);
  return (int64_t)n;
}

// views [n][2] / values [n] (8-byte, f64 when is_f64 else i64) / value validity (may be null) on the device.
// Returns the number of groups and fills *out_views ([G][2]), *out_sum ([G] u64 bits), *out_cnt / *out_len ([G] u32); -1: not on the fast path (a string longer
// than 12 bytes, more groups than the LDS storage of 512 partitions holds) -- the caller encodes and groups the usual way.
int64_t strview_groupby(const uint64_t* views, const uint64_t* values, const uint64_t* val_validity, int64_t n, bool is_f64, Buf* out_views, Buf* out_sum, Buf* out_cnt, Buf* out_len,
                        std::string* desc) {
  if (n <= 0) return -1;
  const uint32_t log2_parts = 9, NP = 1u << log2_parts;
  static_assert(kSgBlock == 1024, "the scatter kernel's scan wave and flush step take eight partitions per lane / sixteen per 32-lane group: 512 partitions, 1024 threads");
  const double est_groups = sg_estimate_groups(views, n);
  if (est_groups < 0 || est_groups > (double)NP * (double)kSgGroupCap * 0.8) return -1;      // a partition's groups must fit its LDS storage (2816) with room for the spread
  // few groups: a partition holds a handful of strings and every update of its workgroup lands on the same few LDS addresses (measured, 2^26 rows:
  // 100 strings 2.9 ms against 0.95 ms for encode-then-group, 1e4 strings 1.5 against 2.0) -- the usual route is the better one there
  const char* force = std::getenv("PLX_STRGROUP_FORCE");                                   // tests of the operator's semantics on small inputs
  if (est_groups < 4096.0 && !(force && force[0] == '1')) return -1;
  Buf hot_views = dev_alloc(16 * kSgMaxHot), hot_acc = dev_alloc_zero(24 * kSgMaxHot), hot_res = dev_alloc_zero(16);
  {
    const size_t lds = (size_t)kSgHotTable * 16 + 256 * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)sg_hot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); attr = true; }
    hipLaunchKernelGGL(sg_hot_kernel, dim3(1), dim3(kSgBlock), lds, stream(), (const unsigned long long*)views, n, hot_views->as<unsigned long long>(), hot_res->as<unsigned int>());
  }
  PLX_HIP(hipGetLastError());
  uint32_t hot[4] = {0, 0, 0, 0};
  d2h_sync(hot, hot_res->ptr, 16);
  const uint32_t n_hot = hot[0];
  const int64_t nrounds = (n + kSgTile - 1) / kSgTile;
  const uint32_t grid = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(nrounds, device().cu_count));
  const int64_t rounds_per_wg = (nrounds + grid - 1) / grid;
  const uint32_t chunks_per_wg = (uint32_t)(rounds_per_wg * kSgTile / kSgChunkRecs + NP + 2);
  if (chunks_per_wg >= (1u << 21)) return -1;                                              // the scatter packs workgroup-local chunk indices into 21 bits
  const int64_t n_chunks = (int64_t)grid * chunks_per_wg;
  Buf recs = dev_alloc_transient((size_t)n_chunks * kSgChunkDw * 4 + 256);
  Buf chunk_part = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks), chunk_fill = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  PLX_HIP(hipMemsetAsync(chunk_part->ptr, 0xff, sizeof(uint32_t) * (size_t)n_chunks, stream()));
  Buf meta = dev_alloc_zero(64);              // [0..1] group counter, [2] overflow, [3..5] scatter flags
  SgScatter sp{};
  sp.views = (const unsigned long long*)views; sp.values = (const unsigned long long*)values; sp.val_validity = val_validity; sp.n = n;
  sp.recs = recs->as<unsigned int>(); sp.chunk_part = chunk_part->as<unsigned int>(); sp.chunk_fill = chunk_fill->as<unsigned int>(); sp.flags = meta->as<unsigned int>() + 3;
  sp.chunks_per_wg = chunks_per_wg; sp.log2_parts = log2_parts;
  sp.n_hot = n_hot; sp.hot_views = hot_views->as<unsigned long long>(); sp.hot_acc = hot_acc->as<unsigned long long>(); sp.is_f64 = is_f64 ? 1u : 0u;
  static const bool timing = std::getenv("PLX_STRGROUP_TIMING") && std::getenv("PLX_STRGROUP_TIMING")[0] == '1';
  Buf tbuf;
  if (timing) { tbuf = dev_alloc_zero(128); sp.timing = tbuf->as<unsigned long long>(); }
  {
    ProfileScope ps("strgroup_scatter", (uint64_t)n * (24 + 24), (uint64_t)n);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)strgroup_scatter_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)strgroup_scatter_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError(); attr = true;
    }
    if (timing) hipLaunchKernelGGL(strgroup_scatter_kernel<true>, dim3(grid), dim3(kSgBlock), sg_scatter_lds(NP), stream(), sp);
    else hipLaunchKernelGGL(strgroup_scatter_kernel<false>, dim3(grid), dim3(kSgBlock), sg_scatter_lds(NP), stream(), sp);
    PLX_HIP(hipGetLastError());
  }
  Buf counts = dev_alloc_zero(sizeof(uint32_t) * (NP + 1)), cursor = dev_alloc_zero(sizeof(uint32_t) * (NP + 1));
  Buf cl_off = dev_alloc(sizeof(uint64_t) * (NP + 2)), cl_ids = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  {
    ProfileScope ps("part2_chunk_sort", (uint64_t)n_chunks * 12, (uint64_t)n_chunks);
    const int g = grid_for(n_chunks, kBlock * 16, 2);
    hipLaunchKernelGGL(sg_chunk_hist_kernel, dim3(g), dim3(kBlock), sizeof(unsigned int) * NP, stream(), chunk_part->as<unsigned int>(), n_chunks, NP, counts->as<unsigned int>());
    PLX_HIP(hipGetLastError());
    exclusive_scan_u32(counts->as<uint32_t>(), cl_off->as<uint64_t>(), NP);
    hipLaunchKernelGGL(sg_chunk_place_kernel, dim3(g), dim3(kBlock), sizeof(unsigned int) * NP * 2, stream(), chunk_part->as<unsigned int>(), n_chunks, NP,
                       cl_off->as<unsigned long long>(), cursor->as<unsigned int>(), cl_ids->as<unsigned int>());
    PLX_HIP(hipGetLastError());
  }
  const uint64_t max_groups = std::min<uint64_t>((uint64_t)NP * kSgGroupCap + kSgMaxHot, (uint64_t)n);
  *out_views = dev_alloc(16 * (size_t)max_groups + 16);
  *out_sum = dev_alloc(8 * (size_t)max_groups + 8);
  *out_cnt = dev_alloc(4 * (size_t)max_groups + 8);
  *out_len = dev_alloc(4 * (size_t)max_groups + 8);
  SgAgg ap{};
  ap.recs = recs->as<unsigned int>(); ap.chunk_fill = chunk_fill->as<unsigned int>(); ap.cl_off = cl_off->as<unsigned long long>(); ap.cl_ids = cl_ids->as<unsigned int>();
  ap.counter = meta->as<unsigned long long>(); ap.overflow = meta->as<unsigned int>() + 2;
  ap.out_views = (*out_views)->as<unsigned long long>(); ap.out_sum = (*out_sum)->as<unsigned long long>(); ap.out_cnt = (*out_cnt)->as<unsigned int>(); ap.out_len = (*out_len)->as<unsigned int>();
  ap.has_nulls = val_validity ? 1u : 0u;
  ap.max_groups = (uint32_t)std::min<uint64_t>(max_groups, 0xffffffffull); ap.is_f64 = is_f64 ? 1u : 0u;
  {
    ProfileScope ps("strgroup_agg_lds", (uint64_t)n * 24, (uint64_t)n);
    const size_t lds = (size_t)kSgTagSlots * 4 + (size_t)kSgGroupCap * 32;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)strgroup_agg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); attr = true; }
    hipLaunchKernelGGL(strgroup_agg_kernel, dim3(NP), dim3(kSgBlock), lds, stream(), ap);
    PLX_HIP(hipGetLastError());
  }
  if (n_hot) {
    hipLaunchKernelGGL(sg_hot_emit_kernel, dim3(1), dim3(kSgMaxHot), 0, stream(), n_hot, hot_views->as<unsigned long long>(), hot_acc->as<unsigned long long>(), ap.counter, ap.max_groups, ap.overflow,
                       ap.out_views, ap.out_sum, ap.out_cnt, ap.out_len);
    PLX_HIP(hipGetLastError());
  }
  uint32_t res[6] = {0, 0, 0, 0, 0, 0};
  d2h_sync(res, meta->ptr, 24);
  if (timing) {
    unsigned long long t[9];
    d2h_sync(t, tbuf->ptr, 72);
    const double g = (double)grid * 100.0;              // 100 MHz ticks -> microseconds per workgroup
    fprintf(stderr, "[plx strgroup] scatter thread 0, us per workgroup: barrier A %.0f | scan %.0f | barrier B %.0f | tile %.0f | rows arrive %.0f | rank %.0f | load issue + barrier C %.0f | flush %.0f  (%.0f rounds)\n",
            t[0] / g, t[1] / g, t[2] / g, t[3] / g, t[4] / g, t[5] / g, t[6] / g, t[7] / g, (double)t[8] / grid);
  }
  PLX_REQUIRE(!res[3], PLX_ERR_INVALID, "string group-by: a scatter workgroup ran out of chunks");
  if (res[2] || res[4] || res[5]) return -1;               // a partition with more groups than its LDS storage / long strings: the usual route
  if (desc) *desc = "strview_groupby(partitioned by view hash, P=512, rec=24B, tile=3072)+lds_tag_table(slots=16384, groups<=2816), est_groups=" + std::to_string((long long)est_groups) + ", hot=" + std::to_string(n_hot);
  return (int64_t)(((uint64_t)res[1] << 32) | res[0]);
}

}  // namespace k
}  // namespace plx
