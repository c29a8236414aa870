// kernels_sort.hip -- stable multi-key arg-sort (LSD radix) and top-k selection (MSD radix select).
//
// Reference behaviour (restated, not ported): arg_sort_multiple compares row-encoded keys with a stable
// comparison sort on the CPU thread pool (polars-core/src/chunked_array/ops/sort/arg_sort_multiple.rs,
// arg_sort.rs); sort followed by a slice keeps only the first rows (slice_pushdown_lp.rs -> SortExec slice /
// streaming top_k.rs).  GPU shape:
//   encode : every key column is turned into an order-preserving u64 (sign flip for ints, the IEEE total-order
//            flip for floats with NaN canonicalised to the greatest value and -0 -> +0, bitwise NOT for
//            descending) read THROUGH the current permutation, so narrow dtypes are never widened in HBM twice.
//   rounds : keys are processed last to first (LSD over keys); within a key, 8-bit digits least significant
//            first.  One pass over the encoded keys builds all 8 digit histograms; digits on which every row
//            agrees are skipped (an i32-range key costs 4 passes, a dictionary code 1-3).
//   pass   : per-workgroup digit histogram -> device exclusive scan (digit-major) -> stable scatter.  The scatter
//            ranks the 256 keys of a sub-tile with wave64 ballots (8 ballots = match-any on the digit), per-wave
//            digit counts meet in LDS, so ties keep input order without any atomics.
//   nulls  : one extra 1-bit pass per nullable key (null rank is an absolute position: nulls_last is not flipped
//            by `descending`, arg_sort.rs).
//   top-k  : `limit` << n: MSD radix select on the first key finds the smallest prefix bucket that contains the
//            limit-th row; rows at or below it (ties included) are compacted in row order and only those are
//            sorted -- same result as the full stable sort followed by head(limit).
// Everything is HBM-streaming integer work: 12 B read + 12 B written per row and pass.
#include <algorithm>

#include "dev.hpp"
#include "kernels.hpp"
#include "ops.hpp"
#include "scan.hpp"
#include "sort.hpp"

namespace plx {
namespace sort {

using namespace dev;
using k::kBlock;

constexpr int kItems = 16;                 // sub-tiles of kBlock keys per workgroup
constexpr int kTile = kBlock * kItems;     // 4096 keys per workgroup and pass
constexpr int kWaves = kBlock / kWave;

struct KeyCol {
  const void* values;
  const uint64_t* validity;
  int dtype;
};

enum EncodeMode { ENC_VALUE = 0, ENC_NULL_RANK = 1, ENC_SELECT = 2 };

__device__ __forceinline__ uint64_t flip_f64(double d) {
  const uint64_t b = (d != d) ? 0x7ff8000000000000ull : (uint64_t)__double_as_longlong(d + 0.0);   // one NaN, -0 -> +0
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ uint64_t encode_value(const KeyCol& kc, int64_t r) {
  constexpr uint64_t kSign = 0x8000000000000000ull;
  switch (kc.dtype) {
    case PLX_I8: return (uint64_t)(long long)reinterpret_cast<const int8_t*>(kc.values)[r] ^ kSign;
    case PLX_I16: return (uint64_t)(long long)reinterpret_cast<const int16_t*>(kc.values)[r] ^ kSign;
    case PLX_I32: return (uint64_t)(long long)reinterpret_cast<const int32_t*>(kc.values)[r] ^ kSign;
    case PLX_I64: return reinterpret_cast<const uint64_t*>(kc.values)[r] ^ kSign;
    case PLX_U8: return reinterpret_cast<const uint8_t*>(kc.values)[r];
    case PLX_U16: return reinterpret_cast<const uint16_t*>(kc.values)[r];
    case PLX_U32: return reinterpret_cast<const uint32_t*>(kc.values)[r];
    case PLX_F32: return flip_f64((double)reinterpret_cast<const float*>(kc.values)[r]);
    case PLX_F64: return flip_f64(reinterpret_cast<const double*>(kc.values)[r]);
    case PLX_BOOL: return (reinterpret_cast<const uint64_t*>(kc.values)[r >> 6] >> (r & 63)) & 1;
    default: return reinterpret_cast<const uint64_t*>(kc.values)[r];
  }
}

__global__ __launch_bounds__(kBlock) void encode_kernel(KeyCol kc, const uint32_t* __restrict__ perm, int64_t n, int descending, int nulls_last, int mode,
                                                        uint64_t* __restrict__ enc) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = perm ? (int64_t)perm[i] : i;
    const bool valid = !kc.validity || ((kc.validity[r >> 6] >> (r & 63)) & 1);
    uint64_t e;
    if (mode == ENC_NULL_RANK) e = (valid == (nulls_last != 0)) ? 0ull : 1ull;   // nulls_last: valid rows rank 0; nulls first: nulls rank 0
    else if (!valid) e = (mode == ENC_SELECT && nulls_last) ? ~0ull : 0ull;       // all nulls tie within a key
    else { e = encode_value(kc, r); if (descending) e = ~e; }
    enc[i] = e;
  }
}

// hist[d * 256 + v] = rows whose digit d (bits 8d..8d+7) equals v
__global__ __launch_bounds__(kBlock) void digit_hist_kernel(const uint64_t* __restrict__ enc, int64_t n, unsigned int* __restrict__ hist) {
  __shared__ unsigned int h[8 * 256];
  for (int j = threadIdx.x; j < 8 * 256; j += kBlock) h[j] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t e = enc[i];
#pragma unroll
    for (int d = 0; d < 8; d++) atomicAdd(&h[d * 256 + (int)((e >> (8 * d)) & 255)], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 8 * 256; j += kBlock) if (h[j]) atomicAdd(&hist[j], h[j]);
}

// block_hist[v * nblocks + b] = keys of tile b with digit v
__global__ __launch_bounds__(kBlock) void radix_count_kernel(const uint64_t* __restrict__ enc, int64_t n, int shift, unsigned int* __restrict__ block_hist, uint32_t nblocks) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kTile;
#pragma unroll 4
  for (int r = 0; r < kItems; r++) {
    const int64_t i = base + (int64_t)r * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((enc[i] >> shift) & 255)], 1u);
  }
  __syncthreads();
  block_hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one tile, staged through LDS so that HBM sees whole runs instead of 12-byte fragments:
//   0. the tile's keys are loaded once into registers and histogrammed (LDS atomics) -> local start of every digit value;
//   1. ranking rounds of kIPT * kBlock keys (key j of lane t = tile_base + round * kIPT * kBlock + j * kBlock + t): every
//      (item j, wave w) pair ranks its 64 keys with ballots (match-any on the digit) and publishes one count per digit;
//      after ONE barrier thread d turns the kIPT * kWaves counts of digit d into exclusive prefixes (item-major, then
//      wave = input order) and advances the digit's running position; after a second barrier every key knows its slot
//      in the LDS copy of the tile, which ends up sorted by digit with ties in input order;
//   2. the LDS copy is written out linearly: neighbouring lanes hold neighbouring keys of the same digit run, so the
//      stores of a run (4096 / 256 = 16 keys = 128 B of codes on average) coalesce.
constexpr int kIPT = 4;                          // keys per thread and ranking round
constexpr int kGroups = kIPT * kWaves;           // (item, wave) pairs of a round, in input order
static_assert(kItems % kIPT == 0, "a tile is a whole number of rounds");
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(const uint64_t* __restrict__ enc_in, const uint32_t* __restrict__ idx_in, int64_t n, int shift,
                                                               const uint64_t* __restrict__ offsets, uint32_t nblocks, uint64_t* __restrict__ enc_out,
                                                               uint32_t* __restrict__ idx_out) {
  __shared__ uint64_t s_enc[kTile];              // the tile, sorted by digit (32 KB)
  __shared__ uint32_t s_idx[kTile];              // 16 KB
  __shared__ uint64_t gbase[256];                // global slot of the digit's first key of this tile minus its local start
  __shared__ unsigned int hist[256];             // tile histogram, then the running local position of every digit value
  __shared__ unsigned int round_base[256];       // hist[] as it was at the start of the round
  __shared__ uint8_t cnt[kGroups][256];          // keys of (item, wave) with the digit (<= 64; zero between rounds)
  __shared__ uint16_t pre[kGroups][256];         // exclusive prefix of cnt over the (item, wave) pairs (<= kIPT * kBlock)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t base = (int64_t)blockIdx.x * kTile;
  uint64_t e[kItems]; uint32_t idx[kItems];
  hist[tid] = 0;
#pragma unroll
  for (int q = 0; q < kGroups; q++) cnt[q][tid] = 0;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const int64_t i = base + (int64_t)k * kBlock + tid;
    e[k] = i < n ? enc_in[i] : 0ull;
    idx[k] = i < n ? (idx_in ? idx_in[i] : (uint32_t)i) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; k++) if (base + (int64_t)k * kBlock + tid < n) atomicAdd(&hist[(int)((e[k] >> shift) & 255)], 1u);
  __syncthreads();
  if (wave == 0) {   // exclusive scan of the 256 counts: 4 per lane + a wave scan of the lane totals
    unsigned int v[4], tot = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { v[c] = hist[lane * 4 + c]; tot += v[c]; }
    unsigned int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    unsigned int run = incl - tot;
#pragma unroll
    for (int c = 0; c < 4; c++) { hist[lane * 4 + c] = run; run += v[c]; }
  }
  __syncthreads();
  gbase[tid] = offsets[(uint64_t)tid * nblocks + blockIdx.x] - hist[tid];
#pragma unroll
  for (int r = 0; r < kItems / kIPT; r++) {
    uint32_t d[kIPT], rank[kIPT]; bool valid[kIPT];
#pragma unroll
    for (int j = 0; j < kIPT; j++) {
      const int k = r * kIPT + j;
      valid[j] = base + (int64_t)k * kBlock + tid < n;
      d[j] = (uint32_t)((e[k] >> shift) & 255);
      uint64_t same = ballot(valid[j]);            // lanes of this wave holding the same digit
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const bool bit = (d[j] >> b) & 1;
        const uint64_t bal = ballot(bit);
        same &= bit ? bal : ~bal;
      }
      rank[j] = (uint32_t)prefix_rank(same);
      if (valid[j] && rank[j] == 0) cnt[j * kWaves + wave][d[j]] = (uint8_t)popc64(same);
    }
    __syncthreads();
    {
      uint32_t run = 0;
#pragma unroll
      for (int q = 0; q < kGroups; q++) { const uint32_t c = cnt[q][tid]; cnt[q][tid] = 0; pre[q][tid] = (uint16_t)run; run += c; }
      const unsigned int rb = hist[tid];
      round_base[tid] = rb;
      hist[tid] = rb + run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIPT; j++) {
      if (!valid[j]) continue;
      const uint32_t pos = round_base[d[j]] + pre[j * kWaves + wave][d[j]] + rank[j];
      s_enc[pos] = e[r * kIPT + j];
      s_idx[pos] = idx[r * kIPT + j];
    }
    // no third barrier: the next round's leaders write cnt (already zeroed); pre / round_base are rewritten only after the next
    // round's first barrier, which every lane reaches after the reads above
  }
  __syncthreads();
  const int64_t tile_n = n - base < kTile ? n - base : kTile;
#pragma unroll 4
  for (int k = 0; k < kItems; k++) {
    const int p = k * kBlock + tid;
    if (p >= tile_n) break;
    const uint64_t ev = s_enc[p];
    const uint64_t out = gbase[(int)((ev >> shift) & 255)] + (uint64_t)p;
    enc_out[out] = ev;
    idx_out[out] = s_idx[p];
  }
}

// top-k: histogram of the digit at `shift` among rows whose higher bits equal `prefix`
__global__ __launch_bounds__(kBlock) void select_hist_kernel(const uint64_t* __restrict__ enc, int64_t n, int shift, uint64_t prefix, int have_prefix,
                                                             unsigned int* __restrict__ hist) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t e = enc[i];
    if (!have_prefix || (e >> (shift + 8)) == prefix) atomicAdd(&h[(int)((e >> shift) & 255)], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// mask bit i = (enc[i] >> shift) <= bound
__global__ __launch_bounds__(kBlock) void select_mask_kernel(const uint64_t* __restrict__ enc, int64_t n, int shift, uint64_t bound, uint64_t* __restrict__ mask) {
  const int64_t n_round = (n + 63) & ~63ll;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * blockDim.x) {
    const bool keep = i < n && (enc[i] >> shift) <= bound;
    const uint64_t m = ballot(keep);
    if ((threadIdx.x & 63) == 0) mask[i >> 6] = m;
  }
}

// ---- small inputs (a group-by result, top-k candidates): ONE workgroup, comparison rank sort, no host round trips ----
// enc[j * m + i] = order-preserving code of key j for row i, with the null rank folded into nr (bit j of nr[i] set =
// the row sorts AFTER every non-flagged row on key j ... see rank_before).  out[rank(i)] = perm ? perm[i] : i.
constexpr int kSmallBlock = 1024;
constexpr int kSmallMax = 2048;
constexpr int kSmallMaxKeys = 8;

// per key two codes: hi = null rank (0 / 1), lo = value code (0 for nulls); row a sorts before row b iff the
// (hi, lo) sequence over the keys is lexicographically smaller, ties broken by the row position (stable)
__global__ __launch_bounds__(kSmallBlock) void small_rank_sort_kernel(const uint64_t* __restrict__ enc, const uint8_t* __restrict__ nr, int n_keys, int m,
                                                                      const uint32_t* __restrict__ perm, uint32_t* __restrict__ out) {
  for (int i = threadIdx.x; i < m; i += kSmallBlock) {
    uint64_t mine[kSmallMaxKeys]; uint8_t mnr[kSmallMaxKeys];
#pragma unroll
    for (int j = 0; j < kSmallMaxKeys; j++) if (j < n_keys) { mine[j] = enc[(size_t)j * m + i]; mnr[j] = nr[(size_t)j * m + i]; }
    uint32_t rank = 0;
    for (int o = 0; o < m; o++) {          // o is wave-uniform: the loads below are scalar / broadcast
      int c = 0;                           // <0: row o sorts before row i
#pragma unroll
      for (int j = 0; j < kSmallMaxKeys; j++) {
        if (j < n_keys && c == 0) {
          const uint8_t onr = nr[(size_t)j * m + o];
          const uint64_t oe = enc[(size_t)j * m + o];
          c = onr != mnr[j] ? (onr < mnr[j] ? -1 : 1) : (oe != mine[j] ? (oe < mine[j] ? -1 : 1) : 0);
        }
      }
      rank += (c < 0 || (c == 0 && o < i)) ? 1u : 0u;
    }
    out[rank] = perm ? perm[i] : (uint32_t)i;
  }
}

__global__ __launch_bounds__(kBlock) void encode_small_kernel(KeyCol kc, const uint32_t* __restrict__ perm, int m, int descending, int nulls_last,
                                                              uint64_t* __restrict__ enc, uint8_t* __restrict__ nr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int64_t r = perm ? (int64_t)perm[i] : i;
  const bool valid = !kc.validity || ((kc.validity[r >> 6] >> (r & 63)) & 1);
  uint64_t e = 0;
  if (valid) { e = encode_value(kc, r); if (descending) e = ~e; }
  enc[i] = e;
  nr[i] = (valid == (nulls_last != 0)) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------- host side ---
static KeyCol key_col(const ColumnPtr& c) { KeyCol kc; kc.values = c->data(); kc.validity = c->valid_words(); kc.dtype = c->dtype; return kc; }

struct Work {
  int64_t m = 0;                 // rows being sorted
  Buf enc[2], idx[2];            // ping-pong buffers
  int cur = 0;                   // idx[cur] holds the current permutation (when have_perm)
  bool have_perm = false;        // false: identity
  Buf hist, block_hist, offsets;
  uint32_t nblocks = 0;
  int passes = 0, skipped = 0;
};

static void encode(const SortKey& key, const uint32_t* perm, int64_t m, int mode, uint64_t* out) {
  if (!m) return;
  ProfileScope ps("sort_encode", (uint64_t)m * ((dtype_width(key.col->dtype) ? dtype_width(key.col->dtype) : 1) + 8 + (perm ? 4 : 0)), (uint64_t)m);
  hipLaunchKernelGGL(encode_kernel, dim3(k::grid_for(m, kBlock * 4)), dim3(kBlock), 0, stream(), key_col(key.col), perm, m, key.descending ? 1 : 0,
                     key.nulls_last ? 1 : 0, mode, out);
  PLX_HIP(hipGetLastError());
}

static void radix_pass(Work& w, int shift) {
  const int64_t m = w.m;
  {
    ProfileScope ps("sort_radix_count", (uint64_t)m * 8, (uint64_t)m);
    hipLaunchKernelGGL(radix_count_kernel, dim3(w.nblocks), dim3(kBlock), 0, stream(), w.enc[w.cur]->as<uint64_t>(), m, shift, w.block_hist->as<unsigned int>(), w.nblocks);
    PLX_HIP(hipGetLastError());
  }
  k::exclusive_scan_u32(w.block_hist->as<uint32_t>(), w.offsets->as<uint64_t>(), (int64_t)w.nblocks * 256);
  {
    ProfileScope ps("sort_radix_scatter", (uint64_t)m * 24, (uint64_t)m);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(w.nblocks), dim3(kBlock), 0, stream(), w.enc[w.cur]->as<uint64_t>(),
                       w.have_perm ? w.idx[w.cur]->as<uint32_t>() : (const uint32_t*)nullptr, m, shift, w.offsets->as<uint64_t>(), w.nblocks,
                       w.enc[w.cur ^ 1]->as<uint64_t>(), w.idx[w.cur ^ 1]->as<uint32_t>());
    PLX_HIP(hipGetLastError());
  }
  w.cur ^= 1;
  w.have_perm = true;
  w.passes++;
}

// stable sort of the current permutation by the digits of enc[cur] on which rows differ (max_digits least significant digits)
static void sort_encoded(Work& w, int max_digits) {
  PLX_HIP(hipMemsetAsync(w.hist->ptr, 0, 8 * 256 * sizeof(unsigned int), stream()));
  hipLaunchKernelGGL(digit_hist_kernel, dim3(k::grid_for(w.m, kBlock * 8, 4)), dim3(kBlock), 0, stream(), w.enc[w.cur]->as<uint64_t>(), w.m, w.hist->as<unsigned int>());
  PLX_HIP(hipGetLastError());
  std::vector<unsigned int> h(8 * 256);
  d2h_sync(h.data(), w.hist->ptr, h.size() * sizeof(unsigned int));
  for (int d = 0; d < max_digits; d++) {
    bool uniform = false;
    for (int v = 0; v < 256; v++) if ((int64_t)h[d * 256 + v] == w.m) { uniform = true; break; }
    if (uniform) { w.skipped++; continue; }
    // the next pass reads enc[cur]; enc must travel with the permutation, which radix_pass does
    radix_pass(w, 8 * d);
  }
}

static void sort_by_key(Work& w, const SortKey& key) {
  const uint32_t* perm = w.have_perm ? w.idx[w.cur]->as<uint32_t>() : nullptr;
  encode(key, perm, w.m, ENC_VALUE, w.enc[w.cur]->as<uint64_t>());
  const int wd = dtype_width(key.col->dtype);
  // unsigned / bool keys only populate their own width; signed and float keys use all 8 bytes (the histogram skips the uniform ones)
  const int digits = key.descending ? 8 : (key.col->dtype == PLX_BOOL ? 1 : (dtype_is_unsigned(key.col->dtype) ? wd : 8));
  sort_encoded(w, digits);
  if (key.col->validity && column_null_count(key.col) > 0) {
    perm = w.have_perm ? w.idx[w.cur]->as<uint32_t>() : nullptr;
    encode(key, perm, w.m, ENC_NULL_RANK, w.enc[w.cur]->as<uint64_t>());
    sort_encoded(w, 1);
  }
}

static void init_work(Work& w, int64_t m, Buf initial_perm) {
  w.m = m;
  const size_t m1 = (size_t)std::max<int64_t>(m, 1);
  for (int i = 0; i < 2; i++) { w.enc[i] = dev_alloc(m1 * 8); w.idx[i] = dev_alloc(m1 * 4); }
  if (initial_perm) { w.idx[0] = initial_perm; w.have_perm = true; }
  w.nblocks = (uint32_t)((m + kTile - 1) / kTile);
  if (!w.nblocks) w.nblocks = 1;
  w.hist = dev_alloc(8 * 256 * sizeof(unsigned int));
  w.block_hist = dev_alloc((size_t)w.nblocks * 256 * sizeof(unsigned int));
  w.offsets = dev_alloc(((size_t)w.nblocks * 256 + 1) * sizeof(uint64_t));
}

static ColumnPtr make_idx(int64_t n) {
  auto c = std::make_shared<Column>();
  c->dtype = PLX_U32; c->len = n; c->values = dev_alloc(values_bytes(PLX_U32, std::max<int64_t>(n, 1))); c->null_count = 0;
  return c;
}

ColumnPtr sort_indices(const std::vector<SortKey>& keys, int64_t limit, std::string* desc) {
  PLX_REQUIRE(!keys.empty(), PLX_ERR_INVALID, "sort needs at least one key");
  const int64_t n = keys[0].col->len;
  for (auto& kx : keys) PLX_REQUIRE(kx.col->len == n, PLX_ERR_SHAPE, "sort: key columns have different lengths");
  PLX_REQUIRE(n < 0xffffffffll, PLX_ERR_UNSUPPORTED, "sort: more rows than u32 IdxSize");
  const int64_t n_out = limit < 0 ? n : std::min(limit, n);
  ColumnPtr out = make_idx(n_out);
  if (n_out == 0) { if (desc) *desc = "sort[empty]"; return out; }
  std::string d;
  Buf cand;            // candidate rows of the top-k selection (row order)
  int64_t m = n;
  if (limit >= 0 && limit < n / 4 && n > 65536) {
    // ---- MSD radix select on the first key: smallest prefix bucket holding the limit-th row
    Buf enc0 = dev_alloc((size_t)n * 8);
    encode(keys[0], nullptr, n, ENC_SELECT, enc0->as<uint64_t>());
    Buf hist = dev_alloc(256 * sizeof(unsigned int));
    std::vector<unsigned int> h(256);
    uint64_t prefix = 0, need = (uint64_t)limit, below = 0, cand_rows = (uint64_t)n;
    int shift = 56, rounds = 0;
    bool have = false;
    const uint64_t good_enough = std::max<uint64_t>(2 * (uint64_t)limit, (uint64_t)kSmallMax);   // small enough for the one-workgroup sort
    for (;; shift -= 8) {
      PLX_HIP(hipMemsetAsync(hist->ptr, 0, 256 * sizeof(unsigned int), stream()));
      {
        ProfileScope ps("topk_select_hist", (uint64_t)n * 8, (uint64_t)n);
        hipLaunchKernelGGL(select_hist_kernel, dim3(k::grid_for(n, kBlock * 8, 4)), dim3(kBlock), 0, stream(), enc0->as<uint64_t>(), n, shift, prefix, have ? 1 : 0, hist->as<unsigned int>());
        PLX_HIP(hipGetLastError());
      }
      d2h_sync(h.data(), hist->ptr, 256 * sizeof(unsigned int));
      rounds++;
      uint64_t c = 0; int b = 0;
      for (; b < 255; b++) { if (c + h[b] >= need) break; c += h[b]; }
      below += c; need -= c;
      prefix = (prefix << 8) | (uint64_t)b; have = true;
      cand_rows = below + h[b];
      if (cand_rows <= good_enough || shift == 0) break;
    }
    if (cand_rows < (uint64_t)n / 2) {
      Buf mask = dev_alloc(bitmap_bytes(n));
      {
        ProfileScope ps("topk_select_mask", (uint64_t)n * 8 + (uint64_t)n / 8, (uint64_t)n);
        hipLaunchKernelGGL(select_mask_kernel, dim3(k::grid_for(n, kBlock * 8, 4)), dim3(kBlock), 0, stream(), enc0->as<uint64_t>(), n, shift, prefix, mask->as<uint64_t>());
        PLX_HIP(hipGetLastError());
      }
      k::FilterPlan fp = k::filter_prepare(mask->as<uint64_t>(), n);
      PLX_REQUIRE((uint64_t)fp.n_out == cand_rows, PLX_ERR_INVALID, "top-k: candidate count mismatch");
      Buf iota = dev_alloc((size_t)n * 4);
      k::fill_iota_u32(iota->as<uint32_t>(), n);
      cand = dev_alloc((size_t)std::max<int64_t>(fp.n_out, 1) * 4);
      k::filter_apply(fp, 4, iota->ptr, nullptr, cand->ptr, nullptr);
      m = fp.n_out;
      d += "top_k_select[" + std::to_string(rounds) + " digit rounds, candidates=" + std::to_string(m) + "] -> ";
    }
  }
  if (m <= kSmallMax && (int)keys.size() <= kSmallMaxKeys) {
    const int nk = (int)keys.size();
    Buf enc = dev_alloc((size_t)nk * m * 8), nr = dev_alloc((size_t)nk * m);
    const uint32_t* perm = cand ? cand->as<uint32_t>() : nullptr;
    ProfileScope ps("sort_small", (uint64_t)m * nk * 9, (uint64_t)m);
    for (int j = 0; j < nk; j++) {
      hipLaunchKernelGGL(encode_small_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream(), key_col(keys[j].col), perm, (int)m,
                         keys[j].descending ? 1 : 0, keys[j].nulls_last ? 1 : 0, enc->as<uint64_t>() + (size_t)j * m, nr->as<uint8_t>() + (size_t)j * m);
      PLX_HIP(hipGetLastError());
    }
    Buf sorted = dev_alloc((size_t)m * 4);
    hipLaunchKernelGGL(small_rank_sort_kernel, dim3(1), dim3(kSmallBlock), 0, stream(), enc->as<uint64_t>(), nr->as<uint8_t>(), nk, (int)m, perm, sorted->as<uint32_t>());
    PLX_HIP(hipGetLastError());
    PLX_HIP(hipMemcpyAsync(out->values->ptr, sorted->ptr, (size_t)n_out * 4, hipMemcpyDeviceToDevice, stream()));
    d += "rank_sort[rows=" + std::to_string(m) + ", keys=" + std::to_string(nk) + ", one workgroup]";
    if (desc) *desc = d;
    return out;
  }
  Work w;
  init_work(w, m, cand);
  for (size_t j = keys.size(); j-- > 0;) sort_by_key(w, keys[j]);
  if (w.have_perm) PLX_HIP(hipMemcpyAsync(out->values->ptr, w.idx[w.cur]->ptr, (size_t)n_out * 4, hipMemcpyDeviceToDevice, stream()));
  else k::fill_iota_u32(out->values->as<uint32_t>(), n_out);   // every digit of every key was uniform
  d += "radix_sort[rows=" + std::to_string(m) + ", keys=" + std::to_string(keys.size()) + ", passes=" + std::to_string(w.passes) + ", uniform digits skipped=" + std::to_string(w.skipped) + "]";
  if (desc) *desc = d;
  return out;
}

}  // namespace sort
}  // namespace plx
