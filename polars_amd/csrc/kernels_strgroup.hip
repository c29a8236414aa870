// kernels_strgroup.hip -- group_by(<raw Utf8View key>).agg(sum / mean / count / len of ONE numeric column) without a dictionary-encode pass.
//
// The reference groups on string keys by hashing the 16-byte views (crates/polars-expr/src/hash_keys.rs:413-452 BinviewKeys;
// crates/polars-compute/src/binview_index_map.rs: an index map view -> group index that compares inline views by value).  The path of
// rounds 2-3 encoded the views to dictionary codes first (kernels_strview.hip): one random probe of a 128 MB table per row, bound by
// the ~60 G random line fetches per second the fabric delivers -- 22 ms per 1e9 rows before the group-by proper has started.  Nothing in a
// group-by needs the codes in ROW order, so this operator never builds them:
//   scatter   rows are radix-partitioned by the top bits of the view's hash (the scatter of partition3_device.hpp -- rank by one LDS atomic,
//             tile sort, whole 128-B lines out, carry lines -- with 24-byte records {view, value}); no random access at all
//   aggregate one workgroup per partition: an LDS open-addressing table keyed by the 16-byte view (claim by CAS on the view's second word,
//             first word published right behind it; a reader that finds the second word but not yet the first looks again), cells
//             {sum, count of valid values, rows}; the partition's groups -- each with its view -- go straight to the dense output
// The distinct views ARE the dictionary of the result's key column.  Fast path: inline strings (<= 12 bytes: the view is the string), no null
// keys, aggregates sum / mean / count / len; anything else -> false, and the caller takes the encode-then-group route.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
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
constexpr uint32_t kSgRW = 6;                          // record dwords: view (4) + value (2); bit 31 of dword 0 (above the 4-bit length): value is null
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
  unsigned int* flags;               // [0] ran out of chunks, [1] a string longer than 12 bytes, [2] a view whose second word is the EMPTY pattern
  uint32_t chunks_per_wg, log2_parts;
};

// LDS: sorted [tile * 6] u32 | carry [NP][32] u32 | cnt, off[NP + 1], carry_dw, dstA, lines_left, dstB, cur_chunk, cur_lines [NP] u32 | misc [4]
__host__ __device__ inline size_t sg_scatter_lds(uint32_t NP) { return (size_t)kSgTile * kSgRW * 4 + (size_t)NP * 128 + ((size_t)NP * 8 + 1) * 4 + 16; }

__global__ __launch_bounds__(kSgBlock) void strgroup_scatter_kernel(SgScatter p) {
  extern __shared__ unsigned long long sg_lds[];
  const uint32_t NP = 1u << p.log2_parts;
  unsigned int* sorted = reinterpret_cast<unsigned int*>(sg_lds);
  unsigned int* carry = sorted + (size_t)kSgTile * kSgRW;
  unsigned int* cnt = carry + (size_t)NP * 32;
  unsigned int* off = cnt + NP;
  unsigned int* carry_dw = off + NP + 1;
  unsigned int* dstA = carry_dw + NP;
  unsigned int* lines_left = dstA + NP;
  unsigned int* dstB = lines_left + NP;
  unsigned int* cur_chunk = dstB + NP;
  unsigned int* cur_lines = cur_chunk + NP;
  unsigned int* misc = cur_lines + NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < NP; i += kSgBlock) { cnt[i] = 0; carry_dw[i] = 0; cur_chunk[i] = kSgNoChunk; cur_lines[i] = kSgCapLines; }
  if (tid < 4) misc[tid] = 0;
  __syncthreads();
  const uint32_t chunk0 = blockIdx.x * p.chunks_per_wg;
  auto open_chunks = [&](uint32_t part, uint32_t need) -> uint32_t {
    const uint32_t local = atomicAdd(&misc[0], need);
    if (local + need > p.chunks_per_wg) { p.flags[0] = 1u; return chunk0; }
    for (uint32_t e = 0; e < need; e++) p.chunk_part[chunk0 + local + e] = part;
    return chunk0 + local;
  };
  const int64_t nrounds = (p.n + kSgTile - 1) / kSgTile;
  const uint32_t per_lane = NP >> 6;
  ulonglong2 vn[kSgRows];
  unsigned long long xn[kSgRows];
  auto load = [&](int64_t rd, ulonglong2* v, unsigned long long* x) __attribute__((always_inline)) {
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) {
      const int64_t row = rd * kSgTile + (int64_t)j * kSgBlock + tid;
      if (row < p.n) { v[j] = reinterpret_cast<const ulonglong2*>(p.views)[row]; x[j] = p.values[row]; }
      else { v[j] = make_ulonglong2(kSgEmpty, kSgEmpty); x[j] = 0; }
    }
  };
  int64_t rd = blockIdx.x;
  if (rd < nrounds) load(rd, vn, xn);
  for (; rd < nrounds; rd += gridDim.x) {
    ulonglong2 v[kSgRows];
    unsigned long long x[kSgRows];
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) { v[j] = vn[j]; x[j] = xn[j]; }
    if (rd + gridDim.x < nrounds) load(rd + gridDim.x, vn, xn);
    uint32_t part[kSgRows];
    bool live[kSgRows], vnull[kSgRows];
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) {
      const int64_t row = rd * kSgTile + (int64_t)j * kSgBlock + tid;
      live[j] = row < p.n;
      vnull[j] = live[j] && p.val_validity && !((p.val_validity[row >> 6] >> (row & 63)) & 1);
      if (live[j] && (uint32_t)v[j].x > 12u) { p.flags[1] = 1u; live[j] = false; }          // a long string: the view is not the string -> the caller falls back
      if (live[j] && v[j].y == kSgEmpty) { p.flags[2] = 1u; live[j] = false; }
      part[j] = live[j] ? (uint32_t)(sg_hash(v[j].x, v[j].y) >> (64 - p.log2_parts)) : 0xffffffffu;
      if (live[j]) part[j] |= atomicAdd(&cnt[part[j]], 1u) << 10;                            // rank within (tile, partition) above the partition's 10 bits
    }
    __syncthreads();                                                                          // A
    if (wave == 0) {
      uint32_t s = 0;
      for (uint32_t q = 0; q < per_lane; q++) s += cnt[(uint32_t)lane * per_lane + q];
      uint32_t incl = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
      uint32_t o = incl - s;
      for (uint32_t q = 0; q < per_lane; q++) {
        const uint32_t pp = (uint32_t)lane * per_lane + q;
        const uint32_t c = cnt[pp];
        off[pp] = o; o += c;
        cnt[pp] = 0;
        const uint32_t nl = (carry_dw[pp] + c * kSgRW) >> 5;
        if (nl) {
          uint32_t ch = cur_chunk[pp], ln = cur_lines[pp];
          const uint32_t left = kSgCapLines - ln;
          dstA[pp] = ch * kSgCapLines + ln; lines_left[pp] = left;
          if (nl > left) {
            const uint32_t extra = nl - left, need = (extra + kSgCapLines - 1) / kSgCapLines;
            if (ch != kSgNoChunk) p.chunk_fill[ch] = kSgChunkRecs;
            const uint32_t first = open_chunks(pp, need);
            for (uint32_t e = 0; e + 1 < need; e++) p.chunk_fill[first + e] = kSgChunkRecs;
            dstB[pp] = first * kSgCapLines;
            ch = first + need - 1; ln = extra - (need - 1) * kSgCapLines;
          } else ln += nl;
          cur_chunk[pp] = ch; cur_lines[pp] = ln;
        }
      }
      if (lane == 63) off[NP] = o;
    }
    __syncthreads();                                                                          // B
#pragma unroll
    for (uint32_t j = 0; j < kSgRows; j++) {
      if (!live[j]) continue;
      unsigned int* dst = sorted + (size_t)(off[part[j] & 1023u] + (part[j] >> 10)) * kSgRW;
      dst[0] = (uint32_t)v[j].x | (vnull[j] ? 0x80000000u : 0u); dst[1] = (uint32_t)(v[j].x >> 32);
      dst[2] = (uint32_t)v[j].y; dst[3] = (uint32_t)(v[j].y >> 32);
      dst[4] = (uint32_t)x[j]; dst[5] = (uint32_t)(x[j] >> 32);
    }
    __syncthreads();                                                                          // C
    {
      const uint32_t g = (uint32_t)tid >> 4, l16 = (uint32_t)tid & 15u;
      for (uint32_t pp = g; pp < NP; pp += kSgBlock >> 4) {
        const uint32_t o_dw = off[pp] * kSgRW, r_dw = (off[pp + 1] - off[pp]) * kSgRW, c_dw = carry_dw[pp];
        const uint32_t total = c_dw + r_dw, nl = total >> 5, rem = total & 31u;
        const unsigned int* cy = carry + (size_t)pp * 32;
        const uint32_t a = dstA[pp], left = lines_left[pp], b = dstB[pp];
        for (uint32_t i = 0; i < nl; i++) {
          const uint32_t d = i * 32 + l16 * 2;
          uint2 w;
          w.x = d < c_dw ? cy[d] : sorted[o_dw + d - c_dw];
          w.y = d + 1 < c_dw ? cy[d + 1] : sorted[o_dw + d + 1 - c_dw];
          const uint64_t line = i < left ? (uint64_t)a + i : (uint64_t)b + (i - left);
          *reinterpret_cast<uint2*>(p.recs + line * 32 + l16 * 2) = w;
        }
        if (nl == 0) { for (uint32_t i = l16; i < r_dw; i += 16) carry[(size_t)pp * 32 + c_dw + i] = sorted[o_dw + i]; }
        else { for (uint32_t i = l16; i < rem; i += 16) carry[(size_t)pp * 32 + i] = sorted[o_dw + nl * 32 + i - c_dw]; }
        if (l16 == 0) carry_dw[pp] = rem;
      }
    }
  }
  __syncthreads();
  for (uint32_t pp = tid; pp < NP; pp += kSgBlock) {
    uint32_t ch = cur_chunk[pp], ln = cur_lines[pp];
    const uint32_t rem = carry_dw[pp];
    if (rem) {
      if (ln == kSgCapLines) { if (ch != kSgNoChunk) p.chunk_fill[ch] = kSgChunkRecs; ch = open_chunks(pp, 1); ln = 0; }
      for (uint32_t i = 0; i < rem; i++) p.recs[((uint64_t)ch * kSgCapLines + ln) * 32 + i] = carry[(size_t)pp * 32 + i];
    }
    if (ch != kSgNoChunk) p.chunk_fill[ch] = (ln * 32 + rem) / kSgRW;
  }
}

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
  unsigned int* overflow;            // [0] 1: an LDS table filled up, 2: more groups than the output holds
  unsigned long long* out_views;     // [max_groups][2]
  unsigned long long* out_sum;       // [max_groups] f64 or i64 bits
  unsigned int* out_cnt;             // valid values per group
  unsigned int* out_len;             // rows per group
  uint32_t log2_slots, log2_parts, max_groups, is_f64;
};

// LDS: w1 [NS] u64 (the claim word) | w0 [NS] u64 | sum [NS] u64 | cnt [NS] u32 | len [NS] u32
__global__ __launch_bounds__(kSgBlock) void strgroup_agg_kernel(SgAgg a) {
  extern __shared__ unsigned long long sg_lds[];
  const uint32_t NS = 1u << a.log2_slots, mask = NS - 1u;
  unsigned long long* w1s = sg_lds;
  unsigned long long* w0s = w1s + NS;
  unsigned long long* sums = w0s + NS;
  unsigned int* cnts = reinterpret_cast<unsigned int*>(sums + NS);
  unsigned int* lens = cnts + NS;
  __shared__ unsigned int n_occ, cursor_l, full;
  __shared__ unsigned long long gbase;
  const uint32_t p = blockIdx.x;
  for (uint32_t i = threadIdx.x; i < NS; i += blockDim.x) { w1s[i] = kSgEmpty; w0s[i] = kSgEmpty; sums[i] = 0ull; cnts[i] = 0; lens[i] = 0; }
  if (threadIdx.x == 0) { n_occ = 0; cursor_l = 0; full = 0; }
  __syncthreads();
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint64_t c_beg = a.cl_off[p], c_end = a.cl_off[p + 1];
  constexpr uint32_t kPerLane = kSgChunkRecs / 64;
  uint2 ra[kPerLane][3], rb[kPerLane][3];
  uint32_t fill_a = 0, fill_b = 0;
  auto load_id = [&](uint64_t j) -> uint32_t { return a.cl_ids[j < c_end ? j : c_end - 1]; };
  auto load_chunk = [&](uint64_t j, uint32_t id, uint2 (*r)[3], uint32_t& fill) __attribute__((always_inline)) {
    fill = j < c_end ? a.chunk_fill[id] : 0u;
    const uint2* base = reinterpret_cast<const uint2*>(a.recs + (uint64_t)id * kSgChunkDw);
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint2* q = base + (size_t)((uint32_t)lane + u * 64u) * 3;
      r[u][0] = q[0]; r[u][1] = q[1]; r[u][2] = q[2];
    }
  };
  auto process = [&](uint2 (*r)[3], uint32_t fill) __attribute__((always_inline)) {
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t i = (uint32_t)lane + u * 64u;
      const bool live = i < fill;
      const bool vnull = (r[u][0].x >> 31) & 1u;
      const unsigned long long w0 = ((unsigned long long)r[u][0].y << 32) | (r[u][0].x & 0x7fffffffu);
      const unsigned long long w1 = ((unsigned long long)r[u][1].y << 32) | r[u][1].x;
      const unsigned long long x = ((unsigned long long)r[u][2].y << 32) | r[u][2].x;
      uint32_t sl = (uint32_t)(sg_hash(w0, w1) >> (64 - a.log2_parts - a.log2_slots)) & mask;      // the bits below the partition's
      bool found = !live;
      // every lane runs the same number of rounds of {look, claim, publish} | wave barrier | {compare}: a lane never WAITS inside a round for a word
      // that a lane of its own wave is about to publish -- it looks again in the next round
      for (uint32_t it = 0; it < 4 * NS + 64; it++) {
        if (__all(found)) break;
        bool again = false;
        unsigned long long cur = kSgEmpty;
        if (!found) {
          cur = w1s[sl];
          if (cur == kSgEmpty) {
            const unsigned long long old = atomicCAS(&w1s[sl], kSgEmpty, w1);
            if (old == kSgEmpty) { w0s[sl] = w0; found = true; }
            else cur = old;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (!found) {
          if (cur == w1) {
            const unsigned long long k0 = *reinterpret_cast<volatile unsigned long long*>(&w0s[sl]);
            if (k0 == kSgEmpty) again = true;             // claimed, first word not published yet: same slot, next round
            else if (k0 == w0) found = true;
          }
          if (!found && !again) sl = (sl + 1) & mask;
        }
      }
      if (live && !found) { full = 1; continue; }
      if (!live) continue;
      atomicAdd(&lens[sl], 1u);
      if (!vnull) {
        atomicAdd(&cnts[sl], 1u);
        if (a.is_f64) atomicAdd(reinterpret_cast<double*>(&sums[sl]), __longlong_as_double((long long)x));
        else atomicAdd(&sums[sl], x);
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
  uint32_t mine = 0;
  for (uint32_t s = threadIdx.x; s < NS; s += blockDim.x) mine += w1s[s] != kSgEmpty;
  if (mine) atomicAdd(&n_occ, mine);
  __syncthreads();
  if (threadIdx.x == 0) gbase = n_occ ? atomicAdd(a.counter, (unsigned long long)n_occ) : 0ull;
  __syncthreads();
  if (gbase + n_occ > a.max_groups) { if (threadIdx.x == 0) atomicExch(a.overflow, 2u); return; }
  for (uint32_t s = threadIdx.x; s < NS; s += blockDim.x) {
    if (w1s[s] == kSgEmpty) continue;
    const uint64_t o = gbase + atomicAdd(&cursor_l, 1u);
    a.out_views[o * 2] = w0s[s]; a.out_views[o * 2 + 1] = w1s[s];
    a.out_sum[o] = sums[s]; a.out_cnt[o] = cnts[s]; a.out_len[o] = lens[s];
  }
}
// distinct views among the first S rows: one 64-bit hash per row into a table of 4 S slots (a sample; hash collisions undercount by ~S / 2^64)
__global__ __launch_bounds__(kBlock) void sg_sample_kernel(const unsigned long long* __restrict__ views, int64_t S, unsigned long long* __restrict__ slots, uint32_t log2_cap,
                                                           unsigned int* __restrict__ res) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const ulonglong2 v = reinterpret_cast<const ulonglong2*>(views)[i];
  if ((uint32_t)v.x > 12u) { res[1] = 1u; return; }
  unsigned long long h = sg_hash(v.x, v.y);
  if (h == kSgEmpty) h = 0;
  const uint64_t mask = (1ull << log2_cap) - 1;
  for (uint64_t sl = (h >> 20) & mask;; sl = (sl + 1) & mask) {
    const unsigned long long old = atomicCAS(&slots[sl], kSgEmpty, h);
    if (old == kSgEmpty) { atomicAdd(&res[0], 1u); return; }
    if (old == h) return;
  }
}
// d distinct in a sample of S rows out of n -> G = the solution of d = G (1 - exp(-S / G)) (uniform draws); -1: a long string in the sample
double sg_estimate_groups(const uint64_t* views, int64_t n) {
  const int64_t S = std::min<int64_t>(n, (int64_t)1 << 20);
  const uint32_t log2_cap = 22;
  Buf slots = dev_alloc(8ull << log2_cap), res = dev_alloc_zero(8);
  PLX_HIP(hipMemsetAsync(slots->ptr, 0xff, 8ull << log2_cap, stream()));
  hipLaunchKernelGGL(sg_sample_kernel, dim3((unsigned)((S + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream(), (const unsigned long long*)views, S, slots->as<unsigned long long>(), log2_cap,
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
}  // namespace

// views [n][2] / values [n] (8-byte, f64 when is_f64 else i64) / value validity (may be null) on the device.
// Returns the number of groups and fills *out_views ([G][2]), *out_sum ([G] u64 bits), *out_cnt / *out_len ([G] u32); -1: not on the fast path (a string longer
// than 12 bytes, more groups than the LDS tables of 512 partitions hold, a view with the EMPTY bit pattern) -- the caller encodes and groups the usual way.
int64_t strview_groupby(const uint64_t* views, const uint64_t* values, const uint64_t* val_validity, int64_t n, bool is_f64, Buf* out_views, Buf* out_sum, Buf* out_cnt, Buf* out_len,
                        std::string* desc) {
  if (n <= 0) return -1;
  const uint32_t log2_parts = 9, NP = 1u << log2_parts, log2_slots = 12;                    // 512 x 4096 slots of 32 B = 128 KB of LDS per partition
  const double est_groups = sg_estimate_groups(views, n);
  if (est_groups < 0 || est_groups > (double)NP * (double)(1u << log2_slots) * 0.6) return -1;
  const int64_t nrounds = (n + kSgTile - 1) / kSgTile;
  const uint32_t grid = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(nrounds, device().cu_count));
  const int64_t rounds_per_wg = (nrounds + grid - 1) / grid;
  const uint32_t chunks_per_wg = (uint32_t)(rounds_per_wg * kSgTile / kSgChunkRecs + NP + 2);
  const int64_t n_chunks = (int64_t)grid * chunks_per_wg;
  Buf recs = dev_alloc((size_t)n_chunks * kSgChunkDw * 4 + 256);
  Buf chunk_part = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks), chunk_fill = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  PLX_HIP(hipMemsetAsync(chunk_part->ptr, 0xff, sizeof(uint32_t) * (size_t)n_chunks, stream()));
  Buf meta = dev_alloc_zero(64);              // [0..1] group counter, [2] overflow, [3..5] scatter flags
  SgScatter sp{};
  sp.views = (const unsigned long long*)views; sp.values = (const unsigned long long*)values; sp.val_validity = val_validity; sp.n = n;
  sp.recs = recs->as<unsigned int>(); sp.chunk_part = chunk_part->as<unsigned int>(); sp.chunk_fill = chunk_fill->as<unsigned int>(); sp.flags = meta->as<unsigned int>() + 3;
  sp.chunks_per_wg = chunks_per_wg; sp.log2_parts = log2_parts;
  {
    ProfileScope ps("strgroup_scatter", (uint64_t)n * (24 + 24), (uint64_t)n);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)strgroup_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); attr = true; }
    hipLaunchKernelGGL(strgroup_scatter_kernel, dim3(grid), dim3(kSgBlock), sg_scatter_lds(NP), stream(), sp);
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
  const uint64_t max_groups = std::min<uint64_t>((uint64_t)NP << log2_slots, (uint64_t)n);
  *out_views = dev_alloc(16 * (size_t)max_groups + 16);
  *out_sum = dev_alloc(8 * (size_t)max_groups + 8);
  *out_cnt = dev_alloc(4 * (size_t)max_groups + 8);
  *out_len = dev_alloc(4 * (size_t)max_groups + 8);
  SgAgg ap{};
  ap.recs = recs->as<unsigned int>(); ap.chunk_fill = chunk_fill->as<unsigned int>(); ap.cl_off = cl_off->as<unsigned long long>(); ap.cl_ids = cl_ids->as<unsigned int>();
  ap.counter = meta->as<unsigned long long>(); ap.overflow = meta->as<unsigned int>() + 2;
  ap.out_views = (*out_views)->as<unsigned long long>(); ap.out_sum = (*out_sum)->as<unsigned long long>(); ap.out_cnt = (*out_cnt)->as<unsigned int>(); ap.out_len = (*out_len)->as<unsigned int>();
  ap.log2_slots = log2_slots; ap.log2_parts = log2_parts; ap.max_groups = (uint32_t)std::min<uint64_t>(max_groups, 0xffffffffull); ap.is_f64 = is_f64 ? 1u : 0u;
  {
    ProfileScope ps("strgroup_agg_lds", (uint64_t)n * 24, (uint64_t)n);
    const size_t lds = ((size_t)1 << log2_slots) * 32;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)strgroup_agg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); attr = true; }
    hipLaunchKernelGGL(strgroup_agg_kernel, dim3(NP), dim3(kSgBlock), lds, stream(), ap);
    PLX_HIP(hipGetLastError());
  }
  uint32_t res[6] = {0, 0, 0, 0, 0, 0};
  d2h_sync(res, meta->ptr, 24);
  PLX_REQUIRE(!res[3], PLX_ERR_INVALID, "string group-by: a scatter workgroup ran out of chunks");
  if (res[2] || res[4] || res[5]) return -1;               // table overflow / long strings / the EMPTY pattern: the usual route
  if (desc) *desc = "strview_groupby(partitioned by view hash, P=512, rec=24B, tile=3072)+lds_view_table(slots=4096), est_groups=" + std::to_string((long long)est_groups);
  return (int64_t)(((uint64_t)res[1] << 32) | res[0]);
}

}  // namespace k
}  // namespace plx
