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
