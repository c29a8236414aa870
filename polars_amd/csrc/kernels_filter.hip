// kernels_filter.hip -- stream compaction by a selection bitmap, gather by u32 index,
// and the device-wide exclusive scan both build on.
//
// filter: replaces polars-compute/src/filter/{mod.rs:18-110,scalar.rs:85-138,
// boolean.rs:54-226,avx512.rs:48-115}.  GPU shape: one wave64 owns a 2048-row tile
// (32 mask words).  Pass 1 popcounts tiles, a device scan turns the counts into tile
// offsets, pass 2 re-reads the mask word (wave-uniform), ranks lanes with
// v_mbcnt (prefix popcount of the ballot-shaped mask) and writes kept rows densely
// -> coalesced 8-B stores.  All-zero words are skipped without touching the values
// (the reference's `m == 0` fast path); validity bits are compacted with the same
// ranks and merged into the output bitmap with two 64-bit atomic ORs per word.
//
// gather: replaces polars-compute/src/gather/primitive.rs:9-78.
#include "dev.hpp"
#include "kernels.hpp"
#include "scan.hpp"

namespace plx {
namespace k {

using namespace dev;

constexpr int kTileWords = 32;             // 2048 rows per tile
constexpr int kTileRows = kTileWords * 64;

__device__ __forceinline__ uint64_t mask_word(const uint64_t* mask, int64_t w, int64_t nwords, int64_t n) {
  if (w >= nwords) return 0;
  uint64_t m = mask[w];
  if (w == nwords - 1 && (n & 63)) m &= (~0ull) >> (64 - (n & 63));
  return m;
}

__global__ __launch_bounds__(kBlock) void tile_count_kernel(const uint64_t* __restrict__ mask, int64_t n, int64_t ntiles,
                                                            uint32_t* __restrict__ counts) {
  const int lane = lane_id();
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // two tiles per wave-iteration: lanes 0-31 -> tile t, lanes 32-63 -> tile t+1
  for (int64_t t = wave * 2; t < ntiles; t += nwaves * 2) {
    int64_t tile = t + (lane >> 5);
    uint64_t m = (tile < ntiles) ? mask_word(mask, tile * kTileWords + (lane & 31), nwords, n) : 0;
    uint32_t c = (uint32_t)popc64(m);
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) c += __shfl_xor(c, s, 64);
    if ((lane & 31) == 0 && tile < ntiles) counts[tile] = c;
  }
}

FilterPlan filter_prepare(const uint64_t* mask, int64_t n) {
  FilterPlan p;
  p.n = n; p.mask = mask;
  if (n == 0) return p;
  int64_t ntiles = (n + kTileRows - 1) / kTileRows;
  Buf counts = dev_alloc(sizeof(uint32_t) * (size_t)ntiles);
  p.tile_offsets = dev_alloc(sizeof(uint64_t) * (size_t)(ntiles + 1));
  {
    ProfileScope ps("filter_tile_count", (uint64_t)n / 8, (uint64_t)n);
    hipLaunchKernelGGL(tile_count_kernel, dim3(grid_for(ntiles, 8)), dim3(kBlock), 0, stream(), mask, n, ntiles, counts->as<uint32_t>());
    PLX_HIP(hipGetLastError());
  }
  exclusive_scan_u32(counts->as<uint32_t>(), p.tile_offsets->as<uint64_t>(), ntiles);  // writes ntiles+1 entries
  uint64_t total = 0;
  d2h_sync(&total, p.tile_offsets->as<uint64_t>() + ntiles, 8);
  p.n_out = (int64_t)total;
  return p;
}

// W = element type by width (uint8/16/32/64).  BITS: `values` is a bitmap to compact.
template <class W, bool BITS>
__global__ __launch_bounds__(kBlock) void filter_kernel(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ tile_off,
                                                        int64_t n, int64_t ntiles, const W* __restrict__ values,
                                                        const uint64_t* __restrict__ bits_in, W* __restrict__ out,
                                                        unsigned long long* __restrict__ bits_out) {
  const int lane = lane_id();
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    // lane j (<32) holds mask word j of the tile and its exclusive prefix of popcounts
    uint64_t mw = (lane < kTileWords) ? mask_word(mask, t * kTileWords + lane, nwords, n) : 0;
    uint32_t pc = (uint32_t)popc64(mw), incl = pc;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) { uint32_t o = __shfl_up(incl, s, 64); if (lane >= s) incl += o; }
    uint32_t excl = incl - pc;
    const uint64_t base_out = tile_off[t];
    if (__shfl(incl, 31, 64) == 0) continue;  // wave-uniform: nothing kept in this tile
#pragma unroll 4
    for (int j = 0; j < kTileWords; j++) {
      const uint64_t m = shfl_u64(mw, j);
      if (m == 0) continue;  // uniform
      const uint64_t o = base_out + (uint64_t)__shfl(excl, j, 64);
      const int64_t row = (t * kTileWords + j) * 64 + lane;
      const bool keep = (m >> lane) & 1;
      const int rank = prefix_rank(m);
      if constexpr (!BITS) {
        if (m == ~0ull) out[o + lane] = values[row];  // dense word: straight copy
        else if (keep) out[o + rank] = values[row];
      }
      if (bits_in) {
        const uint64_t vw = bits_in[(t * kTileWords + j)];
        uint64_t contrib = (keep && ((vw >> lane) & 1)) ? (1ull << rank) : 0ull;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) contrib |= shfl_xor_u64(contrib, s);
        if (lane == 0 && contrib) {
          const int sh = (int)(o & 63);
          atomicOr(&bits_out[o >> 6], (unsigned long long)(contrib << sh));
          if (sh && (contrib >> (64 - sh))) atomicOr(&bits_out[(o >> 6) + 1], (unsigned long long)(contrib >> (64 - sh)));
        }
      }
    }
  }
}

void filter_apply(const FilterPlan& p, int width, const void* values, const uint64_t* validity, void* out_values, uint64_t* out_validity) {
  if (p.n == 0 || p.n_out == 0) return;
  int64_t ntiles = (p.n + kTileRows - 1) / kTileRows;
  int grid = grid_for(ntiles, 4);
  ProfileScope ps("filter_compact", (uint64_t)p.n / 8 + (uint64_t)(p.n + p.n_out) * (uint64_t)width, (uint64_t)p.n);
  const uint64_t* toff = p.tile_offsets->as<uint64_t>();
  auto bo = reinterpret_cast<unsigned long long*>(out_validity);
  switch (width) {
    case 0:
      hipLaunchKernelGGL((filter_kernel<uint8_t, true>), dim3(grid), dim3(kBlock), 0, stream(), p.mask, toff, p.n, ntiles, (const uint8_t*)nullptr,
                         (const uint64_t*)values, (uint8_t*)nullptr, reinterpret_cast<unsigned long long*>(out_values));
      if (validity && out_validity)
        hipLaunchKernelGGL((filter_kernel<uint8_t, true>), dim3(grid), dim3(kBlock), 0, stream(), p.mask, toff, p.n, ntiles, (const uint8_t*)nullptr,
                           validity, (uint8_t*)nullptr, bo);
      break;
    case 1: hipLaunchKernelGGL((filter_kernel<uint8_t, false>), dim3(grid), dim3(kBlock), 0, stream(), p.mask, toff, p.n, ntiles, (const uint8_t*)values, out_validity ? validity : nullptr, (uint8_t*)out_values, bo); break;
    case 2: hipLaunchKernelGGL((filter_kernel<uint16_t, false>), dim3(grid), dim3(kBlock), 0, stream(), p.mask, toff, p.n, ntiles, (const uint16_t*)values, out_validity ? validity : nullptr, (uint16_t*)out_values, bo); break;
    case 4: hipLaunchKernelGGL((filter_kernel<uint32_t, false>), dim3(grid), dim3(kBlock), 0, stream(), p.mask, toff, p.n, ntiles, (const uint32_t*)values, out_validity ? validity : nullptr, (uint32_t*)out_values, bo); break;
    case 8: hipLaunchKernelGGL((filter_kernel<uint64_t, false>), dim3(grid), dim3(kBlock), 0, stream(), p.mask, toff, p.n, ntiles, (const uint64_t*)values, out_validity ? validity : nullptr, (uint64_t*)out_values, bo); break;
    default: fail(PLX_ERR_INVALID, "filter: bad element width");
  }
  PLX_HIP(hipGetLastError());
}

__device__ __forceinline__ unsigned long long spread_bits32(unsigned int x) {     // bit i of x -> bit 2i
  unsigned long long v = x;
  v = (v | (v << 16)) & 0x0000ffff0000ffffull;
  v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
  v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v;
}
// ---------------------------------------------------------------- filter -> frame: all payload columns in one pass ---
// Second half of the fused filter (first half: fused_sinks.hpp BallotSink).  A wave takes four consecutive 128-row wave tiles: their ballots and output offsets
// are wave-uniform (scalar loads), lane l owns rows 2l, 2l + 1 of each tile exactly as in the predicate scan, so a lane's two rows of an 8-byte column are ONE
// 16-byte load.  The two rows stay the raw bits that load brought -- taken apart only when stored: a conversion between load and store (or a width switch around
// each load) is a use, and a use is an s_waitcnt; the first version had every one of its 48 loads per tile behind a wait of its own.  Columns arrive sorted by
// width, one straight-line section per width, kCompactGroup columns in flight together.  No atomics, no LDS, no barriers.
template <int W> struct RowPair;
template <> struct RowPair<8> { using T = unsigned long long; using V = ulonglong2; static __device__ __forceinline__ T lo(const V& v) { return v.x; } static __device__ __forceinline__ T hi(const V& v) { return v.y; }
                                static __device__ __forceinline__ V make(T a, T b) { V v; v.x = a; v.y = b; return v; } };
template <> struct RowPair<4> { using T = unsigned int; using V = uint2; static __device__ __forceinline__ T lo(const V& v) { return v.x; } static __device__ __forceinline__ T hi(const V& v) { return v.y; }
                                static __device__ __forceinline__ V make(T a, T b) { V v; v.x = a; v.y = b; return v; } };
template <> struct RowPair<2> { using T = unsigned short; using V = unsigned int; static __device__ __forceinline__ T lo(const V& v) { return (T)(v & 0xffffu); } static __device__ __forceinline__ T hi(const V& v) { return (T)(v >> 16); }
                                static __device__ __forceinline__ V make(T a, T b) { return (V)a | ((V)b << 16); } };
template <> struct RowPair<1> { using T = unsigned char; using V = unsigned short; static __device__ __forceinline__ T lo(const V& v) { return (T)(v & 0xffu); } static __device__ __forceinline__ T hi(const V& v) { return (T)(v >> 8); }
                                static __device__ __forceinline__ V make(T a, T b) { return (V)((V)a | ((V)b << 8)); } };
constexpr int kWaveTile = 128;
template <int W, bool FULL, int kCompactU>
__device__ __forceinline__ void compact_load_col(const void* p, int64_t wrow, int64_t n_rows, typename RowPair<W>::V (&v)[kCompactU]) {
  using RP = RowPair<W>;
  const typename RP::T* q = static_cast<const typename RP::T*>(p) + wrow;
  if constexpr (FULL) {
#pragma unroll
    for (int u = 0; u < kCompactU; u++) v[u] = *reinterpret_cast<const typename RP::V*>(q + (int64_t)u * kWaveTile);
  } else {
#pragma unroll
    for (int u = 0; u < kCompactU; u++) {
      const int64_t row = wrow + (int64_t)u * kWaveTile;
      const typename RP::T a = row < n_rows ? q[(int64_t)u * kWaveTile] : (typename RP::T)0, b = row + 1 < n_rows ? q[(int64_t)u * kWaveTile + 1] : (typename RP::T)0;
      v[u] = RP::make(a, b);
    }
  }
}
template <int W, int kCompactU>
__device__ __forceinline__ void compact_store_col(void* out, const unsigned long long (&pos)[kCompactU], const unsigned long long (&b0)[kCompactU], const unsigned long long (&b1)[kCompactU],
                                                  const typename RowPair<W>::V (&v)[kCompactU]) {
  using RP = RowPair<W>;
  typename RP::T* o = static_cast<typename RP::T*>(out);
  const int lane = lane_id();
#pragma unroll
  for (int u = 0; u < kCompactU; u++) {
    const unsigned int p0 = (unsigned int)(b0[u] >> lane) & 1u, p1 = (unsigned int)(b1[u] >> lane) & 1u;
    if (p0) o[pos[u]] = RP::lo(v[u]);
    if (p1) o[pos[u] + p0] = RP::hi(v[u]);
  }
}
template <int W, bool FULL, int kCompactU, int kCompactGroup>
__device__ __forceinline__ void compact_move_cols(const CompactCols& cc, int first, int count, int64_t wrow, int64_t n_rows, const unsigned long long (&pos)[kCompactU],
                                                  const unsigned long long (&b0)[kCompactU], const unsigned long long (&b1)[kCompactU]) {
  for (int c0 = 0; c0 < count; c0 += kCompactGroup) {
    typename RowPair<W>::V v[kCompactGroup][kCompactU];
#pragma unroll
    for (int c = 0; c < kCompactGroup; c++) if (c0 + c < count) compact_load_col<W, FULL, kCompactU>(cc.in[first + c0 + c], wrow, n_rows, v[c]);
#pragma unroll
    for (int c = 0; c < kCompactGroup; c++) if (c0 + c < count) compact_store_col<W, kCompactU>(cc.out[first + c0 + c], pos, b0, b1, v[c]);
  }
}
template <bool FULL, int kCompactU, int kCompactGroup>
__device__ __forceinline__ void compact_quad(const CompactCols& cc, const unsigned long long* __restrict__ ballots, const unsigned long long* __restrict__ offsets, int64_t q, int64_t n_rows,
                                             int64_t n_wt, uint32_t* __restrict__ row_ids) {
  const int lane = lane_id();
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long b0[kCompactU], b1[kCompactU], pos[kCompactU];
  unsigned int any = 0;
#pragma unroll
  for (int u = 0; u < kCompactU; u++) {
    const int64_t t = q * kCompactU + u;
    const bool have = FULL || t < n_wt;                                     // wave-uniform
    b0[u] = have ? uniform_ld(ballots, (uint64_t)t * 2) : 0ull;
    b1[u] = have ? uniform_ld(ballots, (uint64_t)t * 2 + 1) : 0ull;
    pos[u] = (have ? uniform_ld(offsets, (uint64_t)t) : 0ull) + (unsigned long long)(popc64(b0[u] & lt) + popc64(b1[u] & lt));
    any |= (b0[u] | b1[u]) != 0ull;
  }
  if (!any) return;                                                         // wave-uniform: nothing kept in these 512 rows, nothing loaded
  const int64_t wrow = q * (int64_t)(kCompactU * kWaveTile) + (int64_t)lane * 2;
  if (row_ids) {
#pragma unroll
    for (int u = 0; u < kCompactU; u++) {
      const int64_t row = wrow + (int64_t)u * kWaveTile;
      const unsigned int p0 = (unsigned int)(b0[u] >> lane) & 1u, p1 = (unsigned int)(b1[u] >> lane) & 1u;
      if (p0) row_ids[pos[u]] = (uint32_t)row;
      if (p1) row_ids[pos[u] + p0] = (uint32_t)(row + 1);
    }
  }
  const int n8 = cc.n_w[0], n4 = cc.n_w[1], n2 = cc.n_w[2], n1 = cc.n_w[3];
  if (n8) compact_move_cols<8, FULL, kCompactU, kCompactGroup>(cc, 0, n8, wrow, n_rows, pos, b0, b1);
  if (n4) compact_move_cols<4, FULL, kCompactU, kCompactGroup>(cc, n8, n4, wrow, n_rows, pos, b0, b1);
  if (n2) compact_move_cols<2, FULL, kCompactU, kCompactGroup>(cc, n8 + n4, n2, wrow, n_rows, pos, b0, b1);
  if (n1) compact_move_cols<1, FULL, kCompactU, kCompactGroup>(cc, n8 + n4 + n2, n1, wrow, n_rows, pos, b0, b1);
}
template <int kCompactU, int kCompactGroup>
__global__ __launch_bounds__(kBlock) void compact_by_ballots_kernel(CompactCols cc, const unsigned long long* __restrict__ ballots, const unsigned long long* __restrict__ offsets, int64_t n_rows,
                                                                    uint32_t* __restrict__ row_ids) {
  // (the wave index through readfirstlane: the compiler then KNOWS it is wave-uniform and the ballots / offsets become scalar loads)
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (int64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n_wt = (n_rows + kWaveTile - 1) / kWaveTile, n_quads = (n_wt + kCompactU - 1) / kCompactU;
  for (int64_t q = wave; q < n_quads; q += nwaves) {
    if ((q + 1) * (int64_t)(kCompactU * kWaveTile) <= n_rows) compact_quad<true, kCompactU, kCompactGroup>(cc, ballots, offsets, q, n_rows, n_wt, row_ids);
    else compact_quad<false, kCompactU, kCompactGroup>(cc, ballots, offsets, q, n_rows, n_wt, row_ids);
  }
}
// ballots -> the LSB-first selection bitmap + per-2048-row tile offsets filter_apply works from (Boolean columns and validity bitmaps are compacted by it)
__global__ __launch_bounds__(kBlock) void ballots_to_mask_kernel(const unsigned long long* __restrict__ ballots, const unsigned long long* __restrict__ offsets, int64_t n_wt, int64_t n_tiles,
                                                                 unsigned long long* __restrict__ mask /* [n_tiles * 32] */, unsigned long long* __restrict__ tile_off /* [n_tiles + 1] */) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles * (kTileRows / kWaveTile); t += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long b0 = t < n_wt ? ballots[t * 2] : 0ull, b1 = t < n_wt ? ballots[t * 2 + 1] : 0ull;
    mask[t * 2] = spread_bits32((unsigned int)b0) | (spread_bits32((unsigned int)b1) << 1);
    mask[t * 2 + 1] = spread_bits32((unsigned int)(b0 >> 32)) | (spread_bits32((unsigned int)(b1 >> 32)) << 1);
    if (t % (kTileRows / kWaveTile) == 0) tile_off[t / (kTileRows / kWaveTile)] = offsets[t < n_wt ? t : n_wt];
    if (t == 0) tile_off[n_tiles] = offsets[n_wt];
  }
}

// row ids only (the candidate list of a join): one THREAD per wave tile -- 16 bytes of ballots in, the set bits walked in row order.  (The all-columns kernel spends its time on
// scalar-load latency when it has nothing to move: 0.56 ms for 6e8 rows with 3e6 hits; this one 0.05.)
__global__ __launch_bounds__(kBlock) void ballots_to_rowids_kernel(const unsigned long long* __restrict__ ballots, const unsigned long long* __restrict__ offsets, int64_t n_wt,
                                                                   uint32_t* __restrict__ row_ids) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_wt; t += (int64_t)gridDim.x * blockDim.x) {
    const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(ballots + t * 2);
    if (!(b.x | b.y)) continue;
    unsigned long long o = offsets[t];
    // lane l holds rows 2l, 2l + 1: the ballots interleave into the two 64-row masks of the tile
    unsigned long long m[2] = {spread_bits32((unsigned int)b.x) | (spread_bits32((unsigned int)b.y) << 1), spread_bits32((unsigned int)(b.x >> 32)) | (spread_bits32((unsigned int)(b.y >> 32)) << 1)};
#pragma unroll
    for (int h = 0; h < 2; h++)
      for (unsigned long long w = m[h]; w; w &= w - 1) row_ids[o++] = (uint32_t)(t * kWaveTile + h * 64 + __builtin_ctzll(w));
  }
}

Selection selection_finish(Buf ballots, Buf counts, int64_t n) {
  Selection s;
  s.n = n; s.ballots = ballots;
  if (n == 0) return s;
  const int64_t n_wt = (n + kWaveTile - 1) / kWaveTile;
  s.offsets = dev_alloc(sizeof(uint64_t) * (size_t)(n_wt + 1));
  exclusive_scan_u32(counts->as<uint32_t>(), s.offsets->as<uint64_t>(), n_wt);      // writes n_wt + 1 entries
  uint64_t total = 0;
  d2h_sync(&total, s.offsets->as<uint64_t>() + n_wt, 8);
  s.n_out = (int64_t)total;
  return s;
}
void compact_by_ballots(const Selection& sel, const CompactCols& cols_in, uint32_t* row_ids) {
  if (sel.n == 0 || sel.n_out == 0 || (cols_in.n_cols == 0 && !row_ids)) return;
  if (cols_in.n_cols == 0) {
    const int64_t n_wt = (sel.n + kWaveTile - 1) / kWaveTile;
    ProfileScope ps("filter_rowids", (uint64_t)n_wt * 24 + (uint64_t)sel.n_out * 4, (uint64_t)sel.n);
    hipLaunchKernelGGL(ballots_to_rowids_kernel, dim3(grid_for(n_wt, kBlock * 2)), dim3(kBlock), 0, stream(), sel.ballots->as<unsigned long long>(), sel.offsets->as<unsigned long long>(), n_wt, row_ids);
    PLX_HIP(hipGetLastError());
    return;
  }
  // one straight-line section per width in the kernel: columns sorted widest first
  CompactCols cc{};
  uint64_t bytes = (uint64_t)sel.n / 8 + (row_ids ? (uint64_t)sel.n_out * 4 : 0);
  for (int wi = 0; wi < 4; wi++) {
    const int w = 8 >> wi;
    for (int c = 0; c < cols_in.n_cols; c++) if (cols_in.width[c] == w) { cc.in[cc.n_cols] = cols_in.in[c]; cc.out[cc.n_cols] = cols_in.out[c]; cc.width[cc.n_cols] = (uint8_t)w; cc.n_cols++; cc.n_w[wi]++; }
  }
  PLX_REQUIRE(cc.n_cols == cols_in.n_cols, PLX_ERR_INVALID, "compact_by_ballots: payload widths must be 1, 2, 4 or 8 bytes");
  for (int c = 0; c < cc.n_cols; c++) bytes += (uint64_t)(sel.n + sel.n_out) * cc.width[c];
  // variant: wave tiles a wave moves per step x payload columns in flight together (PLX_COMPACT_VARIANT = "<U><G>": measurement)
  // measured on the 1e9-row frame (three 8-byte columns, half of the rows kept; tools/exp_filter.py): 8 wave tiles x 3 columns in flight 6.0-6.1 ms, 4 x 4 6.4-7.1,
  // 4 x 3 6.9, 8 x 2 6.0-6.6, 4 x 2 6.6-7.8, 2 x 4 6.7-8.1
  static const int variant = [] { const char* e = getenv("PLX_COMPACT_VARIANT"); return e ? atoi(e) : 83; }();
  static const int bpc = [] { const char* e = getenv("PLX_BPC_COMPACT"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? v : 12; }();
  const int U = variant / 10 == 8 ? 8 : variant / 10 == 2 ? 2 : 4;
  const int64_t n_quads = (sel.n + U * kWaveTile - 1) / (U * kWaveTile);
  ProfileScope ps("filter_compact_cols", bytes, (uint64_t)sel.n);
  const dim3 grid(grid_for(n_quads, kBlock / 64, bpc)), block(kBlock);
#define PLX_COMPACT_LAUNCH(UU, GG) hipLaunchKernelGGL((compact_by_ballots_kernel<UU, GG>), grid, block, 0, stream(), cc, sel.ballots->as<unsigned long long>(), sel.offsets->as<unsigned long long>(), sel.n, row_ids)
  switch (variant) {
    case 42: PLX_COMPACT_LAUNCH(4, 2); break;
    case 43: PLX_COMPACT_LAUNCH(4, 3); break;
    case 82: PLX_COMPACT_LAUNCH(8, 2); break;
    case 44: PLX_COMPACT_LAUNCH(4, 4); break;
    case 24: PLX_COMPACT_LAUNCH(2, 4); break;
    default: PLX_COMPACT_LAUNCH(8, 3); break;
  }
#undef PLX_COMPACT_LAUNCH
  PLX_HIP(hipGetLastError());
}
FilterPlan selection_to_plan(const Selection& sel, Buf* mask_keep) {
  FilterPlan p;
  p.n = sel.n; p.n_out = sel.n_out;
  if (sel.n == 0) return p;
  const int64_t n_wt = (sel.n + kWaveTile - 1) / kWaveTile, n_tiles = (sel.n + kTileRows - 1) / kTileRows;
  Buf mask = dev_alloc(sizeof(uint64_t) * (size_t)n_tiles * kTileWords);
  p.tile_offsets = dev_alloc(sizeof(uint64_t) * (size_t)(n_tiles + 1));
  hipLaunchKernelGGL(ballots_to_mask_kernel, dim3(grid_for(n_tiles * (kTileRows / kWaveTile), kBlock)), dim3(kBlock), 0, stream(), sel.ballots->as<unsigned long long>(),
                     sel.offsets->as<unsigned long long>(), n_wt, n_tiles, mask->as<unsigned long long>(), p.tile_offsets->as<unsigned long long>());
  PLX_HIP(hipGetLastError());
  p.mask = mask->as<uint64_t>();
  *mask_keep = mask;
  return p;
}

// the kept ROW INDICES of a selection (the probe-side index of a join whose candidates are all rows)
__global__ __launch_bounds__(kBlock) void filter_rowids_kernel(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ tile_off, int64_t n, int64_t ntiles, uint32_t* __restrict__ out) {
  const int lane = lane_id();
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    uint64_t mw = (lane < kTileWords) ? mask_word(mask, t * kTileWords + lane, nwords, n) : 0;
    uint32_t pc = (uint32_t)popc64(mw), incl = pc;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) { uint32_t o = __shfl_up(incl, s, 64); if (lane >= s) incl += o; }
    uint32_t excl = incl - pc;
    const uint64_t base_out = tile_off[t];
    if (__shfl(incl, 31, 64) == 0) continue;
#pragma unroll 4
    for (int j = 0; j < kTileWords; j++) {
      const uint64_t m = shfl_u64(mw, j);
      if (m == 0) continue;
      const uint64_t o = base_out + (uint64_t)__shfl(excl, j, 64);
      if ((m >> lane) & 1) out[o + prefix_rank(m)] = (uint32_t)((t * kTileWords + j) * 64 + lane);
    }
  }
}
void filter_rowids(const FilterPlan& p, uint32_t* out) {
  if (p.n == 0 || p.n_out == 0) return;
  const int64_t ntiles = (p.n + kTileRows - 1) / kTileRows;
  ProfileScope ps("filter_rowids", (uint64_t)p.n / 8 + (uint64_t)p.n_out * 4, (uint64_t)p.n);
  hipLaunchKernelGGL(filter_rowids_kernel, dim3(grid_for(ntiles, 4)), dim3(kBlock), 0, stream(), p.mask, p.tile_offsets->as<uint64_t>(), p.n, ntiles, out);
  PLX_HIP(hipGetLastError());
}

// ------------------------------------------------------------------- gather ---
template <class W, bool BITS>
__global__ __launch_bounds__(kBlock) void gather_kernel(const W* __restrict__ values, const uint64_t* __restrict__ bits_values,
                                                        const uint64_t* __restrict__ validity, const uint32_t* __restrict__ idx,
                                                        const uint64_t* __restrict__ idx_validity, int64_t n_idx, W* __restrict__ out,
                                                        uint64_t* __restrict__ out_bits, uint64_t* __restrict__ out_validity) {
  const int lane = lane_id();
  const int64_t nwords = (n_idx + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    const int64_t i = w * 64 + lane;
    bool ok = false, bitv = false;
    if (i < n_idx) {
      ok = !idx_validity || ((idx_validity[w] >> lane) & 1);
      const uint32_t j = ok ? idx[i] : 0u;
      if constexpr (BITS) bitv = ok && ((bits_values[j >> 6] >> (j & 63)) & 1);
      else out[i] = ok ? values[j] : (W)0;
      if (ok && validity) ok = (validity[j >> 6] >> (j & 63)) & 1;
    }
    if constexpr (BITS) { uint64_t b = ballot(bitv); if (lane == 0) out_bits[w] = b; }
    if (out_validity) { uint64_t m = ballot(ok); if (lane == 0) out_validity[w] = m; }
  }
}

void gather(int width, const void* values, const uint64_t* validity, const uint32_t* idx, const uint64_t* idx_validity, int64_t n_idx,
            void* out, uint64_t* out_validity) {
  if (n_idx == 0) return;
  ProfileScope ps("gather_u32", (uint64_t)n_idx * (4 + 2 * (uint64_t)(width ? width : 1)), (uint64_t)n_idx);
  int grid = grid_for(n_idx, kBlock * 2);
#define G(W, B) hipLaunchKernelGGL((gather_kernel<W, B>), dim3(grid), dim3(kBlock), 0, stream(), (const W*)values, (const uint64_t*)values, validity, idx, idx_validity, n_idx, (W*)out, (uint64_t*)out, out_validity)
  switch (width) {
    case 0: G(uint8_t, true); break;
    case 1: G(uint8_t, false); break;
    case 2: G(uint16_t, false); break;
    case 4: G(uint32_t, false); break;
    case 8: G(uint64_t, false); break;
    default: fail(PLX_ERR_INVALID, "gather: bad element width");
  }
#undef G
  PLX_HIP(hipGetLastError());
}

// several columns at the same indices in ONE launch (the partitioned join probe gathers every probe-side input at its candidates: four launches of a latency-bound
// kernel, each waiting for the one before; here a lane's loads of all columns are in flight together).  Columns of 4- or 8-byte values without validity.
struct GatherMulti {
  const void* values[kGatherMultiMax];
  void* out[kGatherMultiMax];
  uint8_t width[kGatherMultiMax];
  int n_cols;
};
__global__ __launch_bounds__(kBlock) void gather_multi_kernel(GatherMulti g, const uint32_t* __restrict__ idx, int64_t n_idx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_idx; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t j = idx[i];
    uint64_t v[kGatherMultiMax];
#pragma unroll
    for (int c = 0; c < kGatherMultiMax; c++)
      if (c < g.n_cols) v[c] = g.width[c] == 8 ? static_cast<const uint64_t*>(g.values[c])[j] : (uint64_t)static_cast<const uint32_t*>(g.values[c])[j];
#pragma unroll
    for (int c = 0; c < kGatherMultiMax; c++)
      if (c < g.n_cols) { if (g.width[c] == 8) static_cast<uint64_t*>(g.out[c])[i] = v[c]; else static_cast<uint32_t*>(g.out[c])[i] = (uint32_t)v[c]; }
  }
}
void gather_multi(int n_cols, const int* widths, const void* const* values, const uint32_t* idx, int64_t n_idx, void* const* out) {
  if (n_idx == 0 || n_cols == 0) return;
  PLX_REQUIRE(n_cols <= kGatherMultiMax, PLX_ERR_INVALID, "gather_multi: too many columns");
  GatherMulti g{};
  uint64_t bytes = 0;
  g.n_cols = n_cols;
  for (int c = 0; c < n_cols; c++) {
    PLX_REQUIRE(widths[c] == 4 || widths[c] == 8, PLX_ERR_INVALID, "gather_multi: 4- and 8-byte values only");
    g.values[c] = values[c]; g.out[c] = out[c]; g.width[c] = (uint8_t)widths[c]; bytes += 2 * (uint64_t)widths[c];
  }
  ProfileScope ps("gather_multi", (uint64_t)n_idx * (4 + bytes), (uint64_t)n_idx);
  hipLaunchKernelGGL(gather_multi_kernel, dim3(grid_for(n_idx, kBlock)), dim3(kBlock), 0, stream(), g, idx, n_idx);
  PLX_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace plx
