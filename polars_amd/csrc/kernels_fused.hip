// kernels_fused.hip -- fused scan kernels: one streaming pass over the input columns,
// the expression program evaluated per row in VGPRs, rows handed to a sink.
//
// Replaces, for `Filter -> Select(agg)` and `Filter -> GroupBy` plans, the chain
// FilterExec -> ProjectionExec / GroupByExec of the reference
// (polars-mem-engine/src/executors/{filter.rs:94-145,projection.rs:20-115,
// group_by.rs:60-98}) which materialises a mask, the filtered frame, every
// intermediate expression column and per-group index lists.
//
// gfx950 mapping
//   * a wave64 owns 128-row tiles; lane l holds rows 2l, 2l+1 of the tile, so an
//     8-byte column is ONE global_load_dwordx4 per lane per tile (1 KiB per wave
//     instruction, fully coalesced); narrower columns use proportionally narrower
//     loads of the same two rows.  Tiles are handed out grid-stride; the grid is
//     sized to fill all 256 CUs with >= 2 workgroups each.
//   * the 16 program slots are two ext_vector registers-of-16 (one per row); slot
//     numbers are wave-uniform so dynamic slot access is VGPR index mode, and for the
//     pre-instantiated shapes everything is a compile-time constant.
//   * register sink (<= 8 dense groups): per-lane accumulators acc[G][n_aggs] in
//     VGPRs; a group whose ballot is empty in this wave-tile is skipped on the scalar
//     unit.  End of kernel: wave64 shuffle tree -> LDS across waves -> one partial
//     per workgroup -> tiny finish kernel.  No atomics, deterministic.
//   * dense / hash sinks: device-scope atomics straight into an HBM-resident table
//     (hardware f64 atomic add on gfx950); see HashAggSink.
#include "fused_sinks.hpp"
#include "kernels.hpp"
#include "kernels_fused.hpp"
#include "scan.hpp"
#include "jit.hpp"
#include <cstring>
#include <vector>

// Launch cases of the join-pipeline AOT shapes; empty when fused_shapes.hpp was generated without them.
#ifdef PLX_HAVE_Q3_SHAPES
#define PLX_STATIC_JOIN_BUILD_CASES \
  case SHAPE_Q3_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3_BUILD>, JoinBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break; \
  case SHAPE_Q3D_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3D_BUILD>, JoinBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#define PLX_STATIC_PROBE_AGG_CASES \
  case SHAPE_Q3_PROBE: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3_PROBE>, ProbeAggSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#define PLX_STATIC_REGAGG_EXTRA_CASES \
  case SHAPE_Q3_COUNT: PLX_LAUNCH_SCAN(StatProg<SHAPE_Q3_COUNT>, RegAggSink, grid, 0, sh, args, sp); break;
#define PLX_STATIC_DIRECT_BUILD_CASES \
  case SHAPE_Q3_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3_BUILD>, DirectBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break; \
  case SHAPE_Q3D_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3D_BUILD>, DirectBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#define PLX_STATIC_DIRECT_PROBE_CASES \
  case SHAPE_Q3_PROBE: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3_PROBE>, DirectProbeAggSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#ifdef PLX_HAVE_Q3FULL_SHAPES
#define PLX_STATIC_BITMAP_BUILD_CASES \
  case SHAPE_Q3F_SEMI: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3F_SEMI>, BitmapBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#define PLX_STATIC_Q3F_JOIN_BUILD_CASES \
  case SHAPE_Q3F_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3F_BUILD>, JoinBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#define PLX_STATIC_Q3F_DIRECT_BUILD_CASES \
  case SHAPE_Q3F_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3F_BUILD>, DirectBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
#define PLX_STATIC_Q3F_REGAGG_CASES \
  case SHAPE_Q3F_COUNT: PLX_LAUNCH_SCAN(StatProg<SHAPE_Q3F_COUNT>, RegAggSink, grid, 0, sh, args, sp); break;
#else
#define PLX_STATIC_BITMAP_BUILD_CASES
#define PLX_STATIC_Q3F_JOIN_BUILD_CASES
#define PLX_STATIC_Q3F_DIRECT_BUILD_CASES
#define PLX_STATIC_Q3F_REGAGG_CASES
#endif
#else
#define PLX_STATIC_BITMAP_BUILD_CASES
#define PLX_STATIC_Q3F_JOIN_BUILD_CASES
#define PLX_STATIC_Q3F_DIRECT_BUILD_CASES
#define PLX_STATIC_Q3F_REGAGG_CASES
#define PLX_STATIC_DIRECT_BUILD_CASES
#define PLX_STATIC_DIRECT_PROBE_CASES
#define PLX_STATIC_JOIN_BUILD_CASES
#define PLX_STATIC_PROBE_AGG_CASES
#define PLX_STATIC_REGAGG_EXTRA_CASES
#endif

#include <cstdlib>

namespace plx {
namespace k {

// name of a fused scan in the HIP-event profile: AOT kernels carry their shape id ("fused_scan_ldsagg_static#3" = fused_scan_kernel<StatProg<3>, LdsAggSink>), so
// a counter file collected on another instantiation is never read as this one's (bench.py pmc_traffic matches the full name)
static std::string scope_name(const char* aot, const char* other, int static_id) { return static_id >= 0 ? std::string(aot) + "#" + std::to_string(static_id) : std::string(other); }


using namespace dev;
using namespace fused;

// finish: combine per-workgroup partials [np][cells] -> out[cells]; cell = g * n_aggs + k
__global__ __launch_bounds__(kBlock) void partials_finish_kernel(Shape sh, const unsigned long long* __restrict__ partials, int np, int cells,
                                                                 int cell_stride, unsigned long long* __restrict__ out) {
  __shared__ uint64_t red[kBlock / 64];
  const int cell = blockIdx.x;
  const uint8_t kind = sh.aggs[cell % sh.n_aggs].kind;
  uint64_t x = agg_identity_dev(kind);
  for (int i = threadIdx.x; i < np; i += blockDim.x) x = agg_combine(kind, x, partials[(size_t)i * cell_stride + cell]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = agg_combine(kind, x, shfl_xor_u64(x, m));
  if (lane_id() == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) x = agg_combine(kind, x, red[w]);
    out[cell] = x;
  }
}

// ---- host launchers ----------------------------------------------------------------------
// workgroups per CU of a scan; `knob` names an environment variable that overrides the default (measurement runs only).
// The defaults are measured, and not monotonic in occupancy: on SF100 inputs the streaming scans run 8-10 % faster with 5 or 7
// workgroups per CU than with 4, 6 or 8 (Q1 4.15 vs 4.61 / 4.40 / 4.46 ms, config 2 3.82 vs 4.12 / 4.05 ms, Q3 build 0.85 vs
// 0.91 / 0.90 / 0.98 ms; gpurun_out/r02m, same box, repeated), while the probe scan wants 8 or 16 (1.83 ms; 2.34 with 10, 2.13
// with 12): the grid-stride distance between a wave's consecutive tiles decides which HBM channels its columns land on together.
static int scan_grid(int64_t n_rows, int blocks_per_cu, const char* knob = nullptr) {
  if (knob) { const char* e = getenv(knob); const int v = e ? atoi(e) : 0; if (v >= 1 && v <= 32) blocks_per_cu = v; }
  int64_t ntiles = (n_rows + kTileRows - 1) / kTileRows;
  return grid_for(ntiles, kBlock / 64, blocks_per_cu);
}
int64_t scan_waves(int64_t n_rows) { return (int64_t)scan_grid(n_rows, 8) * (kBlock / 64); }
static uint64_t algo_bytes(const Shape& sh, const Args& args) {
  uint64_t algo = 0;
  for (int i = 0; i < sh.n_inputs; i++) algo += (uint64_t)args.n_rows * dtype_width(sh.in_dtype[i]) + (args.in[i].validity ? (uint64_t)args.n_rows / 8 : 0);
  return algo;
}

#define PLX_LAUNCH_SCAN(PROG, SINK, grid, lds, sh, args, sp) \
  hipLaunchKernelGGL((fused_scan_kernel<PROG, SINK>), dim3(grid), dim3(kBlock), (lds), stream(), sh, args, sp)

void fused_regagg(const Shape& sh, const Args& args, int static_id, uint64_t* out_host) {
  const int grid = scan_grid(args.n_rows, 5, "PLX_BPC_REGAGG");
  Buf partials = dev_alloc(sizeof(uint64_t) * (size_t)(grid + 1) * kMaxAggs);
  unsigned long long* pp = partials->as<unsigned long long>();
  RegAggSink::Params sp{pp};
  {
    ProfileScope ps(scope_name("fused_scan_regagg_static", "fused_scan_regagg_generic", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
    switch (static_id) {
      case SHAPE_CFG2: PLX_LAUNCH_SCAN(StatProg<SHAPE_CFG2>, RegAggSink, grid, 0, sh, args, sp); break;
      case SHAPE_CFG2_NULLX: PLX_LAUNCH_SCAN(StatProg<SHAPE_CFG2_NULLX>, RegAggSink, grid, 0, sh, args, sp); break;
      case SHAPE_CFG1: PLX_LAUNCH_SCAN(StatProg<SHAPE_CFG1>, RegAggSink, grid, 0, sh, args, sp); break;
      PLX_STATIC_REGAGG_EXTRA_CASES
      PLX_STATIC_Q3F_REGAGG_CASES
      default: if (!jit::launch(sh, args, jit::REGAGG, &sp, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); PLX_LAUNCH_SCAN(DynProg, RegAggSink, grid, d.lds, sh, d.args, sp); } break;
    }
    PLX_HIP(hipGetLastError());
  }
  unsigned long long* fin = pp + (size_t)grid * kMaxAggs;
  hipLaunchKernelGGL(partials_finish_kernel, dim3(sh.n_aggs), dim3(kBlock), 0, stream(), sh, pp, grid, (int)sh.n_aggs, (int)kMaxAggs, fin);
  PLX_HIP(hipGetLastError());
  d2h_sync(out_host, fin, (size_t)sh.n_aggs * 8);
}

// the predicate program of a filter -> frame: per 128-row wave tile the ballots of its rows and their count (fused_sinks.hpp BallotSink); `out` sized by the caller
void fused_ballots(const Shape& sh, const Args& args, const BallotOut& out, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_ballots_static", jit::program_mode(-1, args.n_rows)[0] == 'j' ? "fused_scan_ballots[jit]" : "fused_scan_ballots[generic]", static_id).c_str(),
                  algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 5, "PLX_BPC_BALLOTS");
  if (!jit::launch(sh, args, jit::BALLOT, &out, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, BallotSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, out); }
  PLX_HIP(hipGetLastError());
}

// probe rows that pass the probe program's predicate and hit the direct-address table `t` -> ballots + counts per wave tile (fused_sinks.hpp DirectHitsSink)
void fused_direct_hits(const Shape& sh, const Args& args, const DirectJoinTable& t, const BallotOut& out, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_direct_hits_static", "fused_scan_direct_hits", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8, "PLX_BPC_DIRECT_HITS");
  DirectHits p{t, out};
  switch (static_id) {
#ifdef PLX_HAVE_Q3_PROBE_SCATTER
    case SHAPE_Q3_PROBE_SCATTER: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3_PROBE_SCATTER>, DirectHitsSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, p); break;
#endif
    default: if (!jit::launch(sh, args, jit::DIRECT_HITS, &p, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, DirectHitsSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, p); } break;
  }
  PLX_HIP(hipGetLastError());
}
// slot (rank of the key in key order) -> build row, from the pair list of a direct-address build with unique keys
__global__ __launch_bounds__(kBlock) void direct_slot_rows_kernel(DirectJoinTable t, int64_t n_used, unsigned int* __restrict__ slot_row) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_used; o += (int64_t)gridDim.x * blockDim.x) {
    if ((unsigned int)(o % kOrdChunk) >= t.chunk_used[o / kOrdChunk]) continue;      // unused tail of a reserved chunk
    const unsigned long long idx = t.ord_key[o] - (unsigned long long)t.kmin;
    slot_row[direct_slot(t, idx, t.bits[idx >> 6])] = t.ord_row[o];
  }
}
void direct_slot_rows(const DirectJoinTable& t, int64_t n_used, uint32_t* slot_row) {
  if (n_used == 0) return;
  ProfileScope ps("direct_slot_rows", (uint64_t)n_used * 28, (uint64_t)n_used);
  hipLaunchKernelGGL(direct_slot_rows_kernel, dim3(grid_for(n_used, kBlock * 4)), dim3(kBlock), 0, stream(), t, n_used, (unsigned int*)slot_row);
  PLX_HIP(hipGetLastError());
}

int lds_agg_copies(int n_groups, int n_aggs) {
  const size_t budget = 60 * 1024;  // leaves room for 2 workgroups per CU of the 160 KiB LDS
  int c = 16;
  while (c > 1 && (size_t)n_groups * n_aggs * c * 8 > budget) c >>= 1;
  if ((size_t)n_groups * n_aggs * c * 8 > budget) return 0;
  return c;
}

void fused_lds_agg(const Shape& sh, const Args& args, int n_groups, int static_id, uint64_t* out_dev /* [G][n_aggs] */, unsigned int* oob) {
  const int copies = lds_agg_copies(n_groups, sh.n_aggs);
  PLX_REQUIRE(copies > 0, PLX_ERR_INVALID, "fused_lds_agg: group table does not fit LDS");
  const int cells = n_groups * sh.n_aggs;
  const size_t lds = (size_t)cells * copies * 8;
  const int grid = scan_grid(args.n_rows, 5, "PLX_BPC_LDSAGG");
  const bool use_partials = n_groups <= 64;
  Buf partials;
  LdsAggSink::Params sp{};
  sp.n_groups = n_groups; sp.copies = copies; sp.oob = oob;
  if (use_partials) { partials = dev_alloc(sizeof(uint64_t) * (size_t)grid * cells); sp.partials = partials->as<unsigned long long>(); }
  else { init_agg_cells(out_dev, n_groups, sh); sp.global_acc = (unsigned long long*)out_dev; }
  {
    ProfileScope ps(scope_name("fused_scan_ldsagg_static", "fused_scan_ldsagg_generic", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
    switch (static_id) {
      case SHAPE_Q1: PLX_LAUNCH_SCAN(StatProg<SHAPE_Q1>, LdsAggSink, grid, lds, sh, args, sp); break;
      default: if (!jit::launch(sh, args, jit::LDSAGG, &sp, grid, lds)) { const DynLaunch d = dyn_launch(sh, args, lds); PLX_LAUNCH_SCAN(DynProg, LdsAggSink, grid, d.lds, sh, d.args, sp); } break;
    }
    PLX_HIP(hipGetLastError());
  }
  if (use_partials) {
    hipLaunchKernelGGL(partials_finish_kernel, dim3(cells), dim3(kBlock), 0, stream(), sh, sp.partials, grid, cells, cells, (unsigned long long*)out_dev);
    PLX_HIP(hipGetLastError());
  }
}

__global__ __launch_bounds__(kBlock) void fill_u64_kernel(unsigned long long* p, int64_t n, unsigned long long v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(kBlock) void init_acc_kernel(unsigned long long* acc, int64_t n_slots, Shape sh) {
  const int64_t total = n_slots * sh.n_aggs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    acc[i] = agg_identity_dev(sh.aggs[i % sh.n_aggs].kind);
}
void fill_u64(uint64_t* p, int64_t n, uint64_t v) {
  if (n == 0) return;
  hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), (unsigned long long*)p, n, (unsigned long long)v);
  PLX_HIP(hipGetLastError());
}
void init_agg_cells(uint64_t* acc, int64_t n_slots, const Shape& sh) {
  if (n_slots == 0) return;
  hipLaunchKernelGGL(init_acc_kernel, dim3(grid_for(n_slots * sh.n_aggs, kBlock * 4)), dim3(kBlock), 0, stream(), (unsigned long long*)acc, n_slots, sh);
  PLX_HIP(hipGetLastError());
}

void fused_dense_agg(const Shape& sh, const Args& args, const DenseTable& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps("fused_scan_denseagg", algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8);
  switch (static_id) {
    case SHAPE_GB_SUM_CNT_I64: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_GB_SUM_CNT_I64>, DenseAggSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
    case SHAPE_GB_SUM_MEAN_U32_F64: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_GB_SUM_MEAN_U32_F64>, DenseAggSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
    default: if (!jit::launch(sh, args, jit::DENSE, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, DenseAggSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}

void fused_hash_agg(const Shape& sh, const Args& args, const HashTable& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps("fused_scan_hashagg", algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8);
  switch (static_id) {
    case SHAPE_GB_SUM_CNT_I64: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_GB_SUM_CNT_I64>, HashAggSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
    case SHAPE_GB_SUM_MEAN_U32_F64: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_GB_SUM_MEAN_U32_F64>, HashAggSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;
    default: if (!jit::launch(sh, args, jit::HASH, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, HashAggSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}



// ---- slot compaction skeleton -----------------------------------------------------------------
// Every workgroup takes chunks of kBlock * kCompactItems slots; a thread tests kCompactItems slots,
// the workgroup scans the per-thread counts (wave shuffle scan + LDS) and reserves its output range
// with ONE device atomic per chunk.  (One atomic per wave-with-a-hit on a single counter word
// saturates at ~88 atomics/us: a 2^25-slot table took 5.6 ms that way, 0.3 ms this way.)
constexpr int kCompactItems = 8;
template <class Occ, class Emit>
__device__ __forceinline__ void compact_slots(int64_t n_slots, unsigned long long* counter, Occ occ, Emit emit) {
  __shared__ uint32_t wave_tot[kBlock / 64];
  __shared__ unsigned long long chunk_base;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t chunk = (int64_t)kBlock * kCompactItems;
  const int64_t nchunks = (n_slots + chunk - 1) / chunk;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {   // trip count is uniform across the workgroup
    const int64_t s0 = c * chunk + threadIdx.x;
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < kCompactItems; j++) { const int64_t s = s0 + (int64_t)j * kBlock; if (s < n_slots && occ(s)) bits |= 1u << j; }
    const uint32_t cnt = (uint32_t)__popc(bits);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) { if (w < wave) wave_off += wave_tot[w]; total += wave_tot[w]; }
    if (threadIdx.x == 0 && total) chunk_base = atomicAdd(counter, (unsigned long long)total);
    __syncthreads();
    if (total) {
      uint64_t o = chunk_base + wave_off + (incl - cnt);
#pragma unroll
      for (int j = 0; j < kCompactItems; j++) if ((bits >> j) & 1) { emit(s0 + (int64_t)j * kBlock, o); o++; }
    }
    __syncthreads();   // chunk_base / wave_tot are reused by the next chunk
  }
}
static int compact_grid(int64_t n_slots) { return grid_for(n_slots, kBlock * kCompactItems, 8); }

void fused_wide_agg(const Shape& sh, const Args& args, const WideTable& t) {
  if (args.n_rows == 0) return;
  ProfileScope ps("fused_scan_wideagg", algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8);
  if (!jit::launch(sh, args, jit::WIDE, &t, grid, 0)) {
    const DynLaunch d = dyn_launch(sh, args, 0);
    hipLaunchKernelGGL((fused_scan_kernel<DynProg, WideAggSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t);
  }
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void wide_compact_kernel(WideTable t, int n_keys, int n_aggs, int64_t out_stride, unsigned long long* __restrict__ counter,
                                                              unsigned long long* __restrict__ out_words, unsigned char* __restrict__ out_kvalid,
                                                              unsigned long long* __restrict__ out_acc) {
  const int64_t cap = (int64_t)1 << t.log2_cap;
  compact_slots(cap, counter, [&](int64_t s) { return t.tags[s] != kEmptyKey; },
                [&](int64_t s, uint64_t o) {
                  if (!out_words) return;
                  const uint64_t nullmask = t.has_null_word ? t.words[(size_t)n_keys * cap + s] : 0ull;
                  for (int j = 0; j < n_keys; j++) {
                    out_words[(size_t)j * out_stride + o] = t.words[(size_t)j * cap + s];
                    out_kvalid[(size_t)j * out_stride + o] = (unsigned char)(((nullmask >> j) & 1) ^ 1);
                  }
                  for (int k = 0; k < n_aggs; k++) out_acc[o * n_aggs + k] = t.acc[(size_t)s * n_aggs + k];
                });
}
int64_t wide_compact(const WideTable& t, int n_keys, int n_aggs, int64_t out_stride, uint64_t* out_words, uint8_t* out_kvalid, uint64_t* out_acc) {
  Buf counter = dev_alloc_zero(8);
  const int64_t cap = (int64_t)1 << t.log2_cap;
  ProfileScope ps("table_compact", (uint64_t)cap * 8 * (uint64_t)(1 + n_keys + n_aggs), (uint64_t)cap);
  hipLaunchKernelGGL(wide_compact_kernel, dim3(compact_grid(cap)), dim3(kBlock), 0, stream(), t, n_keys, n_aggs, out_stride,
                     counter->as<unsigned long long>(), (unsigned long long*)out_words, (unsigned char*)out_kvalid, (unsigned long long*)out_acc);
  PLX_HIP(hipGetLastError());
  uint64_t n = 0;
  d2h_sync(&n, counter->ptr, 8);
  return (int64_t)n;
}


void fused_join_build(const Shape& sh, const Args& args, const JoinAggTable& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_join_build_static", "fused_scan_join_build", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8);
  switch (static_id) {
    PLX_STATIC_JOIN_BUILD_CASES
    PLX_STATIC_Q3F_JOIN_BUILD_CASES
    default: if (!jit::launch(sh, args, jit::JOIN_BUILD, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, JoinBuildSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}
void fused_probe_agg(const Shape& sh, const Args& args, const JoinAggTable& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_probe_agg_static", "fused_scan_probe_agg", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8);
  switch (static_id) {
    PLX_STATIC_PROBE_AGG_CASES
    default: if (!jit::launch(sh, args, jit::PROBE_AGG, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, ProbeAggSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}

// (only some of the ahead-of-time shapes have a BitmapBuildSink instantiation: the tracer's name must say what actually ran)
static bool bitmap_build_has_static(int id) {
#ifdef PLX_HAVE_Q3_SHAPES
  if (id == SHAPE_Q3_BUILD) return true;
#ifdef PLX_HAVE_Q3FULL_SHAPES
  if (id == SHAPE_Q3F_SEMI) return true;
#endif
#endif
  return false;
}
void fused_bitmap_build(const Shape& sh, const Args& args, const BitmapBuild& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_bitmap_build_static", "fused_scan_bitmap_build", bitmap_build_has_static(static_id) ? static_id : -1).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8);
  switch (static_id) {
#ifdef PLX_HAVE_Q3_SHAPES
    case SHAPE_Q3_BUILD: hipLaunchKernelGGL((fused_scan_kernel<StatProg<SHAPE_Q3_BUILD>, BitmapBuildSink>), dim3(grid), dim3(kBlock), 0, stream(), sh, args, t); break;      // the semi join of Q3's tables
#endif
    PLX_STATIC_BITMAP_BUILD_CASES
    default: if (!jit::launch(sh, args, jit::BITMAP_BUILD, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, BitmapBuildSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}
void fused_direct_build(const Shape& sh, const Args& args, const DirectJoinTable& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_direct_build_static", "fused_scan_direct_build", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 5, "PLX_BPC_DIRECT_BUILD");
  switch (static_id) {
    PLX_STATIC_DIRECT_BUILD_CASES
    PLX_STATIC_Q3F_DIRECT_BUILD_CASES
    default: if (!jit::launch(sh, args, jit::DIRECT_BUILD, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, DirectBuildSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}
void fused_direct_probe_agg(const Shape& sh, const Args& args, const DirectJoinTable& t, int static_id) {
  if (args.n_rows == 0) return;
  ProfileScope ps(scope_name("fused_scan_direct_probe_agg_static", "fused_scan_direct_probe_agg", static_id).c_str(), algo_bytes(sh, args), (uint64_t)args.n_rows);
  const int grid = scan_grid(args.n_rows, 8, "PLX_BPC_DIRECT_PROBE");
  switch (static_id) {
    PLX_STATIC_DIRECT_PROBE_CASES
    default: if (!jit::launch(sh, args, jit::DIRECT_PROBE, &t, grid, 0)) { const DynLaunch d = dyn_launch(sh, args, 0); hipLaunchKernelGGL((fused_scan_kernel<DynProg, DirectProbeAggSink>), dim3(grid), dim3(kBlock), d.lds, stream(), sh, d.args, t); } break;
  }
  PLX_HIP(hipGetLastError());
}
// rank step of the direct-address join table: popcount per 512-bit block (one 64-B line per thread), scanned by
// exclusive_scan_u32; the same launch adds up the ordinals handed out of the pair-list chunks (pairs appended by the build scan)
__global__ __launch_bounds__(kBlock) void direct_popc_kernel(const unsigned long long* __restrict__ bits, int64_t n_blocks, uint32_t* __restrict__ counts,
                                                             const unsigned int* __restrict__ chunk_used, int64_t n_chunks, unsigned long long* __restrict__ n_pairs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_blocks; i += (int64_t)gridDim.x * blockDim.x) {
    const ulonglong2* line = reinterpret_cast<const ulonglong2*>(bits + (i << 3));
    const ulonglong2 a = line[0], b = line[1], c = line[2], d = line[3];
    counts[i] = (uint32_t)(__popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y) + __popcll(c.x) + __popcll(c.y) + __popcll(d.x) + __popcll(d.y));
  }
  unsigned long long mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_chunks; i += (int64_t)gridDim.x * blockDim.x) mine += chunk_used[i];
  const uint64_t w = wave_sum_u64(mine);
  if (lane_id() == 0 && w) atomicAdd(n_pairs, (unsigned long long)w);
}
// second half of the rank step: block prefix + popcounts of the block's earlier words -> one u32 rank per bitmap word
__global__ __launch_bounds__(kBlock) void direct_word_rank_kernel(const unsigned long long* __restrict__ bits, const unsigned long long* __restrict__ block_rank, int64_t n_blocks,
                                                                  unsigned int* __restrict__ rank) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_blocks; i += (int64_t)gridDim.x * blockDim.x) {
    const ulonglong2* line = reinterpret_cast<const ulonglong2*>(bits + (i << 3));
    const ulonglong2 a = line[0], b = line[1], c = line[2], d = line[3];
    unsigned int r = (unsigned int)block_rank[i];
    uint4 lo, hi;
    lo.x = r; r += (unsigned int)__popcll(a.x);
    lo.y = r; r += (unsigned int)__popcll(a.y);
    lo.z = r; r += (unsigned int)__popcll(b.x);
    lo.w = r; r += (unsigned int)__popcll(b.y);
    hi.x = r; r += (unsigned int)__popcll(c.x);
    hi.y = r; r += (unsigned int)__popcll(c.y);
    hi.z = r; r += (unsigned int)__popcll(d.x);
    hi.w = r;
    reinterpret_cast<uint4*>(rank + (i << 3))[0] = lo;
    reinterpret_cast<uint4*>(rank + (i << 3))[1] = hi;
  }
}
// -> set bits (= distinct build keys that passed); *pairs_out = pairs appended by the build scan (synchronises)
uint64_t direct_rank(const DirectJoinTable& t, uint32_t* rank_out, int64_t n_used, uint64_t* n_pairs_dev, uint64_t* pairs_out) {
  const int64_t n_blocks = (int64_t)(t.range / 512 + 1);
  const int64_t n_chunks = (n_used + kOrdChunk - 1) / kOrdChunk;
  Buf counts = dev_alloc(sizeof(uint32_t) * (size_t)n_blocks), block_rank = dev_alloc(sizeof(uint64_t) * (size_t)(n_blocks + 1));
  ProfileScope ps("direct_rank", (uint64_t)n_blocks * (64 + 4 + 8 + 64 + 8 + 32), (uint64_t)n_blocks);
  hipLaunchKernelGGL(direct_popc_kernel, dim3(grid_for(n_blocks, kBlock * 2)), dim3(kBlock), 0, stream(), t.bits, n_blocks, counts->as<uint32_t>(), t.chunk_used, n_chunks,
                     (unsigned long long*)n_pairs_dev);
  PLX_HIP(hipGetLastError());
  exclusive_scan_u32(counts->as<uint32_t>(), block_rank->as<uint64_t>(), n_blocks);
  hipLaunchKernelGGL(direct_word_rank_kernel, dim3(grid_for(n_blocks, kBlock * 2)), dim3(kBlock), 0, stream(), t.bits, block_rank->as<unsigned long long>(), n_blocks, (unsigned int*)rank_out);
  PLX_HIP(hipGetLastError());
  uint64_t total = 0;
  d2h_sync(&total, block_rank->as<uint64_t>() + n_blocks, 8);
  if (pairs_out) d2h_sync(pairs_out, n_pairs_dev, 8);     // the stream is idle: no extra wait
  return total;
}

// output step: pairs whose slot received at least one probe row (LEN cell != 0) -> dense (key, build row, cells)
// (a separate `touched` bitmap set by the probe was measured: the probe got 0.4 ms slower, this pass no faster)
__global__ __launch_bounds__(kBlock) void direct_pairs_compact_kernel(DirectJoinTable t, int64_t n_used, int n_aggs, int len_idx, unsigned long long* __restrict__ counter,
                                                                      unsigned long long* __restrict__ out_keys, unsigned int* __restrict__ out_rows,
                                                                      unsigned long long* __restrict__ out_acc) {
  auto slot_of = [&](int64_t o) -> unsigned long long {
    const unsigned long long idx = t.ord_key[o] - (unsigned long long)t.kmin;
    return direct_slot(t, idx, t.bits[idx >> 6]);
  };
  compact_slots(n_used, counter,
                [&](int64_t o) {
                  if ((unsigned int)(o % kOrdChunk) >= t.chunk_used[o / kOrdChunk]) return false;   // unused tail of a reserved chunk
                  if (t.touch_filter) {     // (a few MB: stays in the caches) no candidate of the partitioned probe had this key's bit -> no probe row matched it
                    const unsigned long long h = (t.ord_key[o] * kP2HashMult) >> (64u - t.log2_touch_bits);
                    if (!((t.touch_filter[h >> 6] >> (h & 63)) & 1ull)) return false;
                  }
                  return t.acc[(size_t)slot_of(o) * n_aggs + len_idx] != 0;
                },
                [&](int64_t o, uint64_t out) {
                  if (!out_keys) return;
                  const unsigned long long s = slot_of(o);
                  out_keys[out] = t.ord_key[o];
                  out_rows[out] = t.ord_row[o];
                  for (int k = 0; k < n_aggs; k++) out_acc[out * n_aggs + k] = t.acc[(size_t)s * n_aggs + k];
                });
}
// sets the touch-filter bit of every valid key of `keys` (the join key column gathered at the partitioned probe's candidate rows, widened to 64 bits)
__global__ __launch_bounds__(kBlock) void touch_filter_kernel(const long long* __restrict__ keys, const uint64_t* __restrict__ validity, int64_t n, unsigned int log2_bits,
                                                              unsigned long long* __restrict__ filter) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (validity && !((validity[i >> 6] >> (i & 63)) & 1ull)) continue;
    const unsigned long long h = ((unsigned long long)keys[i] * kP2HashMult) >> (64u - log2_bits);
    __hip_atomic_fetch_or(&filter[h >> 6], 1ull << (h & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
void touch_filter_set(const int64_t* keys, const uint64_t* validity, int64_t n, unsigned int log2_bits, uint64_t* filter) {
  if (n <= 0) return;
  ProfileScope ps("touch_filter", (uint64_t)n * 16, (uint64_t)n);
  hipLaunchKernelGGL(touch_filter_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), (const long long*)keys, validity, n, log2_bits, (unsigned long long*)filter);
  PLX_HIP(hipGetLastError());
}

int64_t direct_agg_compact(const DirectJoinTable& t, int64_t n_used, int n_aggs, int len_idx, uint64_t* out_keys, uint32_t* out_rows, uint64_t* out_acc) {
  if (n_used == 0) return 0;
  Buf counter = dev_alloc_zero(8);
  ProfileScope ps("table_compact", (uint64_t)n_used * 8 * (uint64_t)(2 + n_aggs), (uint64_t)n_used);
  hipLaunchKernelGGL(direct_pairs_compact_kernel, dim3(compact_grid(n_used)), dim3(kBlock), 0, stream(), t, n_used, n_aggs, len_idx, counter->as<unsigned long long>(),
                     (unsigned long long*)out_keys, (unsigned int*)out_rows, (unsigned long long*)out_acc);
  PLX_HIP(hipGetLastError());
  uint64_t n = 0;
  d2h_sync(&n, counter->ptr, 8);
  return (int64_t)n;
}

__global__ __launch_bounds__(kBlock) void join_agg_compact_kernel(JoinAggTable t, int n_aggs, int len_idx, unsigned long long* __restrict__ counter,
                                                                  unsigned long long* __restrict__ out_keys, unsigned int* __restrict__ out_rows,
                                                                  unsigned long long* __restrict__ out_acc) {
  const int64_t cap = (int64_t)1 << t.log2_cap;
  compact_slots(cap + 1, counter, [&](int64_t s) { return t.acc[(size_t)s * n_aggs + len_idx] != 0; },
                [&](int64_t s, uint64_t o) {
                  if (!out_keys) return;
                  out_keys[o] = s < cap ? *jt_key(t, (uint64_t)s) : kEmptyKey;
                  out_rows[o] = *jt_row(t, (uint64_t)s);
                  for (int k = 0; k < n_aggs; k++) out_acc[o * n_aggs + k] = t.acc[(size_t)s * n_aggs + k];
                });
}
int64_t join_agg_compact(const JoinAggTable& t, int n_aggs, int len_idx, uint64_t* out_keys, uint32_t* out_rows, uint64_t* out_acc) {
  Buf counter = dev_alloc_zero(8);
  const int64_t n_slots = ((int64_t)1 << t.log2_cap) + 1;
  ProfileScope ps("table_compact", (uint64_t)n_slots * 8 * (uint64_t)(2 + n_aggs), (uint64_t)n_slots);
  hipLaunchKernelGGL(join_agg_compact_kernel, dim3(compact_grid(n_slots)), dim3(kBlock), 0, stream(), t, n_aggs, len_idx, counter->as<unsigned long long>(),
                     (unsigned long long*)out_keys, (unsigned int*)out_rows, (unsigned long long*)out_acc);
  PLX_HIP(hipGetLastError());
  uint64_t n = 0;
  d2h_sync(&n, counter->ptr, 8);
  return (int64_t)n;
}

__global__ __launch_bounds__(kBlock) void cells_agg_compact_kernel(const unsigned long long* __restrict__ cell_key, const unsigned int* __restrict__ cell_row, const unsigned long long* __restrict__ acc,
                                                                   int64_t n_cells, int n_aggs, int len_idx, unsigned long long* __restrict__ counter,
                                                                   unsigned long long* __restrict__ out_keys, unsigned int* __restrict__ out_rows, unsigned long long* __restrict__ out_acc) {
  compact_slots(n_cells, counter, [&](int64_t c) { return acc[(size_t)c * n_aggs + len_idx] != 0; },
                [&](int64_t c, uint64_t o) {
                  if (!out_keys) return;
                  out_keys[o] = cell_key[c];
                  out_rows[o] = cell_row[c];
                  for (int k = 0; k < n_aggs; k++) out_acc[o * n_aggs + k] = acc[(size_t)c * n_aggs + k];
                });
}
int64_t cells_agg_compact(const JoinCells& cells, const uint64_t* acc, int64_t n_cells, int n_aggs, int len_idx, uint64_t* out_keys, uint32_t* out_rows, uint64_t* out_acc) {
  if (n_cells <= 0) return 0;
  Buf counter = dev_alloc_zero(8);
  ProfileScope ps("table_compact", (uint64_t)n_cells * (12 + 8 * (uint64_t)n_aggs), (uint64_t)n_cells);
  hipLaunchKernelGGL(cells_agg_compact_kernel, dim3(compact_grid(n_cells)), dim3(kBlock), 0, stream(), cells.key->as<unsigned long long>(), cells.row->as<unsigned int>(),
                     (const unsigned long long*)acc, n_cells, n_aggs, len_idx, counter->as<unsigned long long>(), (unsigned long long*)out_keys, (unsigned int*)out_rows, (unsigned long long*)out_acc);
  PLX_HIP(hipGetLastError());
  uint64_t n = 0;
  d2h_sync(&n, counter->ptr, 8);
  return (int64_t)n;
}

// ---- multi-value join table (duplicate build keys): representatives and row-indexed compaction --------------------------------
// One thread per slot of the build table.  Slots whose key was inserted once are skipped by the duplicate counter in the slot itself (no access to the links).  For a
// key with several build rows the thread walks the chain and gives every row its representative: the first row of the chain (in chain order) that agrees with it on all
// `rc` columns -- the build-side group columns -- null == null, values bitwise (integer-typed columns only; the planner keeps float columns out of this path).  With no
// such columns (the group key is the join key alone) every row of a key is one group: the chain's head represents them all.  A chain longer than `max_chain` raises
// flags[0]: the pairwise search is quadratic in the classes of a chain and runs in one thread.
__device__ __forceinline__ bool rep_cols_equal(const RepCols& rc, unsigned int a, unsigned int b) {
  for (int j = 0; j < rc.n; j++) {
    const bool va = !rc.valid[j] || ((rc.valid[j][a >> 6] >> (a & 63)) & 1ull), vb = !rc.valid[j] || ((rc.valid[j][b >> 6] >> (b & 63)) & 1ull);
    if (va != vb) return false;
    if (!va) continue;
    switch (rc.width[j]) {
      case 1: if (((const unsigned char*)rc.vals[j])[a] != ((const unsigned char*)rc.vals[j])[b]) return false; break;
      case 2: if (((const unsigned short*)rc.vals[j])[a] != ((const unsigned short*)rc.vals[j])[b]) return false; break;
      case 4: if (((const unsigned int*)rc.vals[j])[a] != ((const unsigned int*)rc.vals[j])[b]) return false; break;
      default: if (((const unsigned long long*)rc.vals[j])[a] != ((const unsigned long long*)rc.vals[j])[b]) return false; break;
    }
  }
  return true;
}
__global__ __launch_bounds__(kBlock) void canonicalise_chains_kernel(JoinAggTable t, RepCols rc, unsigned int max_chain, unsigned int* __restrict__ flags) {
  const int64_t cap = (int64_t)1 << t.log2_cap;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= cap; s += (int64_t)gridDim.x * blockDim.x) {
    const unsigned int* rw = jt_row(t, (uint64_t)s);
    const unsigned int head = rw[0];
    if (head == kNoRow32 || rw[1] == 0xffffffffu) continue;          // empty, or a key with ONE build row (its link already names itself)
    unsigned int len = 0, merged = 0;
    for (unsigned int o = head; o != kNoRow32; len++) {
      if (len >= max_chain) { flags[0] = 1u; break; }
      const unsigned long long lo = t.links[o];
      unsigned int rep = o;
      unsigned int q = head;
      for (unsigned int n = 0; n < len; n++) {                                                   // o is the len-th row of the chain: the len rows before it, in chain order
        const unsigned long long lq = t.links[q];
        if (rep == o && (unsigned int)(lq >> 32) == q && rep_cols_equal(rc, q, o)) rep = q;      // the FIRST representative that agrees (only representatives are candidates)
        q = (unsigned int)lq;
      }
      if (rep != o) merged++;
      t.links[o] = (lo & 0xffffffffull) | ((unsigned long long)rep << 32);
      o = (unsigned int)lo;
    }
    if (merged) atomicAdd(&flags[1], merged);        // (statistics: build rows that joined another row's group)
  }
}
void canonicalise_chains(const JoinAggTable& t, const RepCols& rc, unsigned int max_chain, unsigned int* flags) {
  const int64_t n_slots = ((int64_t)1 << t.log2_cap) + 1;
  ProfileScope ps("join_chain_representatives", (uint64_t)n_slots * 16, (uint64_t)n_slots);
  hipLaunchKernelGGL(canonicalise_chains_kernel, dim3(grid_for(n_slots, kBlock * 4)), dim3(kBlock), 0, stream(), t, rc, max_chain, flags);
  PLX_HIP(hipGetLastError());
}
// ---- multi-value table -> groups.  Cells are per KEY (slot); a group is a REPRESENTATIVE build row of the key's chain (canonicalise_chains) and stands for the m rows of
// the chain it represents: its aggregate is m copies of the key's (every probe row of the key joins each of those m build rows) -- sums and counts times m, min / max /
// first unchanged.  Pass 1 counts the representatives of every slot that matched (LEN != 0), a device scan lays them out, pass 2 writes row + scaled cells.
__device__ __forceinline__ unsigned int chain_reps(const JoinAggTable& t, int64_t s, unsigned int* reps /* may be null */, unsigned int* mult /* may be null */, unsigned int cap_out) {
  const unsigned int* rw = jt_row(t, (uint64_t)s);
  const unsigned int head = rw[0];
  if (head == kNoRow32) return 0;
  if (rw[1] == 0xffffffffu) { if (reps) { reps[0] = head; mult[0] = 1; } return 1; }      // a key with ONE build row
  unsigned int n = 0;
  for (unsigned int o = head, g = 0; o != kNoRow32 && g < (1u << 20); g++) {
    const unsigned long long l = t.links[o];
    if ((unsigned int)(l >> 32) == o) {                    // o represents itself: a group
      if (reps && n < cap_out) {
        unsigned int m = 0;
        for (unsigned int q = head, h = 0; q != kNoRow32 && h < (1u << 20); h++) { const unsigned long long lq = t.links[q]; m += (unsigned int)(lq >> 32) == o; q = (unsigned int)lq; }
        reps[n] = o; mult[n] = m;
      }
      n++;
    }
    o = (unsigned int)l;
  }
  return n;
}
__global__ __launch_bounds__(kBlock) void chains_count_kernel(JoinAggTable t, const unsigned long long* __restrict__ acc, int n_aggs, int len_idx, uint32_t* __restrict__ counts) {
  const int64_t cap = (int64_t)1 << t.log2_cap;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= cap; s += (int64_t)gridDim.x * blockDim.x)
    counts[s] = acc[(size_t)s * n_aggs + len_idx] != 0 ? chain_reps(t, s, nullptr, nullptr, 0) : 0u;
}
__global__ __launch_bounds__(kBlock) void chains_emit_kernel(JoinAggTable t, Shape sh, const unsigned long long* __restrict__ acc, int len_idx, const uint64_t* __restrict__ off,
                                                             unsigned int* __restrict__ out_rows, unsigned long long* __restrict__ out_acc) {
  const int64_t cap = (int64_t)1 << t.log2_cap;
  const int n_aggs = sh.n_aggs;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= cap; s += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t o0 = off[s], n = off[s + 1] - o0;
    if (!n) continue;
    // (chains are short -- dbgen's partsupp: 4 rows a key; a key repeated more than kChunk times is emitted in rounds)
    constexpr unsigned int kChunk = 16;
    unsigned int reps[kChunk], mult[kChunk];
    if (n <= kChunk) chain_reps(t, s, reps, mult, kChunk);
    for (uint64_t g = 0; g < n; g++) {
      unsigned int row, m;
      if (n <= kChunk) { row = reps[g]; m = mult[g]; }
      else {      // the g-th representative of a long chain: walk for it
        const unsigned int head = jt_row(t, (uint64_t)s)[0];
        unsigned int seen = 0; row = head; m = 0;
        for (unsigned int o = head, h = 0; o != kNoRow32 && h < (1u << 20); h++) { const unsigned long long l = t.links[o]; if ((unsigned int)(l >> 32) == o) { if (seen == g) { row = o; break; } seen++; } o = (unsigned int)l; }
        for (unsigned int q = head, h = 0; q != kNoRow32 && h < (1u << 20); h++) { const unsigned long long lq = t.links[q]; m += (unsigned int)(lq >> 32) == row; q = (unsigned int)lq; }
      }
      out_rows[o0 + g] = row;
      for (int k = 0; k < n_aggs; k++) {
        unsigned long long c = acc[(size_t)s * n_aggs + k];
        switch (sh.aggs[k].kind) {
          case AGG_SUM_F: c = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)c) * (double)m); break;
          case AGG_SUM_I: case AGG_COUNT: case AGG_COUNT_ORD: case AGG_LEN: c *= (unsigned long long)m; break;
          default: break;       // min / max / first row: the same for every copy
        }
        out_acc[(o0 + g) * n_aggs + k] = c;
      }
    }
  }
}
int64_t chains_agg_compact(const JoinAggTable& t, const Shape& sh, const uint64_t* acc, int len_idx, Buf* out_rows, Buf* out_acc) {
  const int64_t n_slots = ((int64_t)1 << t.log2_cap) + 1;
  Buf counts = dev_alloc(sizeof(uint32_t) * (size_t)n_slots), off = dev_alloc(sizeof(uint64_t) * (size_t)(n_slots + 1));
  ProfileScope ps("table_compact", (uint64_t)n_slots * (8 * (uint64_t)sh.n_aggs + 16), (uint64_t)n_slots);
  hipLaunchKernelGGL(chains_count_kernel, dim3(grid_for(n_slots, kBlock * 4)), dim3(kBlock), 0, stream(), t, (const unsigned long long*)acc, (int)sh.n_aggs, len_idx, counts->as<uint32_t>());
  PLX_HIP(hipGetLastError());
  exclusive_scan_u32(counts->as<uint32_t>(), off->as<uint64_t>(), n_slots);
  uint64_t total = 0;
  d2h_sync(&total, off->as<uint64_t>() + n_slots, 8);
  const int64_t g1 = std::max<int64_t>((int64_t)total, 1);
  *out_rows = dev_alloc(sizeof(uint32_t) * (size_t)g1);
  *out_acc = dev_alloc(sizeof(uint64_t) * (size_t)g1 * sh.n_aggs);
  if (total) {
    hipLaunchKernelGGL(chains_emit_kernel, dim3(grid_for(n_slots, kBlock * 4)), dim3(kBlock), 0, stream(), t, sh, (const unsigned long long*)acc, len_idx, off->as<uint64_t>(),
                       (*out_rows)->as<unsigned int>(), (*out_acc)->as<unsigned long long>());
    PLX_HIP(hipGetLastError());
    PLX_HIP(hipStreamSynchronize(stream()));
  }
  return (int64_t)total;
}

// groups of a table whose cells are per build ROW = build rows whose LEN cell is non-zero
__global__ __launch_bounds__(kBlock) void rows_agg_compact_kernel(const unsigned long long* __restrict__ acc, int64_t n_rows, int n_aggs, int len_idx, unsigned long long* __restrict__ counter,
                                                                  unsigned int* __restrict__ out_rows, unsigned long long* __restrict__ out_acc) {
  compact_slots(n_rows, counter, [&](int64_t s) { return acc[(size_t)s * n_aggs + len_idx] != 0; },
                [&](int64_t s, uint64_t o) {
                  if (!out_rows) return;
                  out_rows[o] = (unsigned int)s;
                  for (int k = 0; k < n_aggs; k++) out_acc[o * n_aggs + k] = acc[(size_t)s * n_aggs + k];
                });
}
int64_t rows_agg_compact(const uint64_t* acc, int64_t n_rows, int n_aggs, int len_idx, uint32_t* out_rows, uint64_t* out_acc) {
  if (n_rows == 0) return 0;
  Buf counter = dev_alloc_zero(8);
  ProfileScope ps("table_compact", (uint64_t)n_rows * 8 * (uint64_t)n_aggs, (uint64_t)n_rows);
  hipLaunchKernelGGL(rows_agg_compact_kernel, dim3(compact_grid(n_rows)), dim3(kBlock), 0, stream(), (const unsigned long long*)acc, n_rows, n_aggs, len_idx, counter->as<unsigned long long>(),
                     (unsigned int*)out_rows, (unsigned long long*)out_acc);
  PLX_HIP(hipGetLastError());
  uint64_t n = 0;
  d2h_sync(&n, counter->ptr, 8);
  return (int64_t)n;
}

// ---- table compaction: occupied slots -> dense output ----------------------------------
// Wave-aggregated output allocation: ballot the occupied lanes, one atomicAdd per wave
// reserves popcount slots, lanes write at base + prefix rank.
__global__ __launch_bounds__(kBlock) void hash_compact_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ acc,
                                                              int64_t n_slots, int64_t cap, int n_aggs, int occ_agg /* LEN/COUNT cell or -1 */,
                                                              unsigned long long* __restrict__ counter, unsigned long long* __restrict__ out_keys,
                                                              unsigned char* __restrict__ out_key_valid, unsigned long long* __restrict__ out_acc) {
  compact_slots(n_slots, counter,
                [&](int64_t s) {
                  if (occ_agg >= 0) return acc[(size_t)s * n_aggs + occ_agg] != 0;   // a group exists iff its row count is non-zero
                  return s < cap ? keys[s] != kEmptyKey : keys[s] == 0;
                },
                [&](int64_t s, uint64_t o) {
                  if (out_keys) {
                    uint64_t kv = 0; unsigned char valid = 1;
                    if (cap < 0) kv = (uint64_t)s;                   // dense table: the slot index is the packed key
                    else if (s < cap) kv = keys[s];
                    else if (s == cap) { kv = 0; valid = 0; }        // null-key group
                    else kv = kEmptyKey;                             // the key equal to the sentinel
                    out_keys[o] = kv; out_key_valid[o] = valid;
                  }
                  if (out_acc) for (int k = 0; k < n_aggs; k++) out_acc[o * n_aggs + k] = acc[(size_t)s * n_aggs + k];
                });
}

// ---- aggregate cells -> typed output column ------------------------------------------------
// one aggregate cell row -> one output element; returns validity
__device__ __forceinline__ bool finalize_one(const unsigned long long* cell, const FinalSpec& sp, void* out, int64_t g) {
  bool valid = true;
  switch (sp.kind) {
    case FIN_COPY64: reinterpret_cast<uint64_t*>(out)[g] = cell[sp.a]; break;
    case FIN_TRUNC32: reinterpret_cast<uint32_t*>(out)[g] = (uint32_t)cell[sp.a]; break;
    case FIN_NARROW:
      if (dtype_width_dev(sp.out_dtype) == 1) reinterpret_cast<uint8_t*>(out)[g] = (uint8_t)cell[sp.a];
      else reinterpret_cast<uint16_t*>(out)[g] = (uint16_t)cell[sp.a];
      break;
    case FIN_MEAN: {
      const uint64_t cnt = cell[sp.b];
      valid = cnt != 0;
      double m = valid ? as_f(cell[sp.a]) / (double)cnt : 0.0;
      if (sp.out_dtype == PLX_F32) reinterpret_cast<float*>(out)[g] = (float)m; else reinterpret_cast<double*>(out)[g] = m;
    } break;
    case FIN_MINMAX_I: {
      valid = cell[sp.b] != 0;
      const uint64_t v = valid ? cell[sp.a] : 0;
      switch (dtype_width_dev(sp.out_dtype)) {
        case 1: reinterpret_cast<uint8_t*>(out)[g] = (uint8_t)v; break;
        case 2: reinterpret_cast<uint16_t*>(out)[g] = (uint16_t)v; break;
        case 4: reinterpret_cast<uint32_t*>(out)[g] = (uint32_t)v; break;
        default: reinterpret_cast<uint64_t*>(out)[g] = v; break;
      }
    } break;
    default: {  // FIN_MINMAX_F
      valid = cell[sp.b] != 0;
      double v = valid ? as_f(cell[sp.a]) : 0.0;
      if (valid && cell[sp.c] == 0) v = __longlong_as_double(0x7ff8000000000000ll);
      reinterpret_cast<double*>(out)[g] = v;
    } break;
  }
  return valid;
}
// packed key (+ per-group valid flag) -> one key element; `bit` receives the value of a boolean key
__device__ __forceinline__ bool decode_one(const unsigned long long* packed, const unsigned char* kvalid, const KeyDecode& kd, void* out, int64_t g, bool& bit) {
  const uint64_t code = (packed[g] >> kd.shift) & kd.mask;
  const bool valid = (!kvalid || kvalid[g]) && (kd.mask == ~0ull || code != kd.null_code);
  const uint64_t v = valid ? code + (uint64_t)kd.min : 0;
  bit = valid && (v & 1);
  switch (kd.dtype) {
    case PLX_BOOL: break;  // written through a ballot by the caller
    case PLX_I8: case PLX_U8: reinterpret_cast<uint8_t*>(out)[g] = (uint8_t)v; break;
    case PLX_I16: case PLX_U16: reinterpret_cast<uint16_t*>(out)[g] = (uint16_t)v; break;
    case PLX_I32: case PLX_U32: reinterpret_cast<uint32_t*>(out)[g] = (uint32_t)v; break;
    case PLX_F32: { double d = as_f(v); reinterpret_cast<float*>(out)[g] = (float)d; } break;
    default: reinterpret_cast<uint64_t*>(out)[g] = v; break;
  }
  return valid;
}

__global__ __launch_bounds__(kBlock) void finalize_kernel(const unsigned long long* __restrict__ acc, int n_aggs, int64_t G, FinalSpec sp,
                                                          void* __restrict__ out, uint64_t* __restrict__ out_valid) {
  const int lane = lane_id();
  const int64_t nwords = (G + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    const int64_t g = w * 64 + lane;
    bool valid = false;
    if (g < G) valid = finalize_one(acc + (size_t)g * n_aggs, sp, out, g);
    if (out_valid) { uint64_t m = ballot(valid); if (lane == 0) out_valid[w] = m; }
  }
}

__global__ __launch_bounds__(kBlock) void finalize_batch_kernel(const unsigned long long* __restrict__ acc, int n_aggs, int64_t G, FinBatch b) {
  const int lane = lane_id();
  const int64_t nwords = (G + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    const int64_t g = w * 64 + lane;
    for (int j = 0; j < b.n; j++) {   // wave-uniform job list (kernel arguments)
      const FinJob& job = b.jobs[j];
      bool valid = false, bit = false;
      if (g < G) valid = job.is_key ? decode_one(job.packed, job.kvalid, job.kd, job.out, g, bit) : finalize_one(acc + (size_t)g * n_aggs, job.fs, job.out, g);
      if (job.is_key && job.kd.dtype == PLX_BOOL) { uint64_t bb = ballot(bit); if (lane == 0) reinterpret_cast<uint64_t*>(job.out)[w] = bb; }
      if (job.out_valid) { uint64_t m = ballot(valid); if (lane == 0) job.out_valid[w] = m; }
    }
  }
}
void finalize_batch(const uint64_t* acc, int n_aggs, int64_t G, const FinBatch& b) {
  if (G == 0 || b.n == 0) return;
  hipLaunchKernelGGL(finalize_batch_kernel, dim3(grid_for(G, kBlock)), dim3(kBlock), 0, stream(), (const unsigned long long*)acc, n_aggs, G, b);
  PLX_HIP(hipGetLastError());
}
void finalize_aggs(const uint64_t* acc, int n_aggs, int64_t G, const FinalSpec& sp, void* out, uint64_t* out_valid) {
  if (G == 0) return;
  hipLaunchKernelGGL(finalize_kernel, dim3(grid_for(G, kBlock)), dim3(kBlock), 0, stream(), (const unsigned long long*)acc, n_aggs, G, sp, out, out_valid);
  PLX_HIP(hipGetLastError());
}

// packed group key (+ per-group valid flag) -> one key column
__global__ __launch_bounds__(kBlock) void decode_key_kernel(const unsigned long long* __restrict__ packed, const unsigned char* __restrict__ kvalid, int64_t G,
                                                            KeyDecode kd, void* __restrict__ out, uint64_t* __restrict__ out_valid) {
  const int lane = lane_id();
  const int64_t nwords = (G + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    const int64_t g = w * 64 + lane;
    bool valid = false, bit = false;
    if (g < G) valid = decode_one(packed, kvalid, kd, out, g, bit);
    if (kd.dtype == PLX_BOOL) { uint64_t b = ballot(bit); if (lane == 0) reinterpret_cast<uint64_t*>(out)[w] = b; }
    if (out_valid) { uint64_t m = ballot(valid); if (lane == 0) out_valid[w] = m; }
  }
}
void decode_key(const uint64_t* packed, const uint8_t* kvalid, int64_t G, const KeyDecode& kd, void* out, uint64_t* out_valid) {
  if (G == 0) return;
  hipLaunchKernelGGL(decode_key_kernel, dim3(grid_for(G, kBlock)), dim3(kBlock), 0, stream(), (const unsigned long long*)packed, (const unsigned char*)kvalid, G, kd, out, out_valid);
  PLX_HIP(hipGetLastError());
}

int64_t table_compact(const uint64_t* keys, const uint64_t* acc, int64_t n_slots, int64_t cap, int n_aggs, int occ_agg, uint64_t* out_keys,
                      uint8_t* out_key_valid, uint64_t* out_acc) {
  Buf counter = dev_alloc_zero(8);
  ProfileScope ps("table_compact", (uint64_t)n_slots * 8 * (uint64_t)(1 + n_aggs), (uint64_t)n_slots);
  hipLaunchKernelGGL(hash_compact_kernel, dim3(compact_grid(n_slots)), dim3(kBlock), 0, stream(), (const unsigned long long*)keys,
                     (const unsigned long long*)acc, n_slots, cap, n_aggs, occ_agg, counter->as<unsigned long long>(), (unsigned long long*)out_keys,
                     (unsigned char*)out_key_valid, (unsigned long long*)out_acc);
  PLX_HIP(hipGetLastError());
  uint64_t n = 0;
  d2h_sync(&n, counter->ptr, 8);
  return (int64_t)n;
}

}  // namespace k

namespace fused {
int find_static_shape(const Shape& s) {
  for (int id = 0; id < kNumStaticShapes; id++) {
    Shape t = static_shape(id);
    if (memcmp(&t, &s, sizeof(Shape)) == 0) return id;
  }
  return -1;
}
}  // namespace fused
}  // namespace plx
