// kernels_reduce.hip -- whole-column reductions (sum / mean / min / max / count).
// One streaming pass computes every aggregate of a column at once (the kernel is
// HBM-bound, the extra VALU work is free): 16-B loads per lane, per-thread
// accumulators, wave64 shuffle tree, LDS across the 4 waves of a workgroup, one
// partial per workgroup, then a single-workgroup finish kernel.  No float atomics:
// results are deterministic for a given grid.
//
// Reference semantics: polars-compute/src/sum.rs:173-203 (wrapping integer sums),
// float_sum.rs:193-313 (f64 accumulation; the reference's pairwise order is not
// reproduced -- float sums are compared at 1e-6 relative), min_max/scalar.rs:23-73
// (NaN ignored unless all values are NaN), aggregate/mod.rs:240-246 (mean).
#include "dev.hpp"
#include "kernels.hpp"

namespace plx {
namespace k {

using namespace dev;

struct Partial {
  uint64_t isum;
  double fsum;
  uint64_t mn, mx;  // bit patterns of T widened (ints: sign/zero extended; floats: f64)
  uint64_t n_valid, n_ordered;
};

template <class T> struct Wide { using type = long long; };
template <> struct Wide<uint8_t> { using type = unsigned long long; };
template <> struct Wide<uint16_t> { using type = unsigned long long; };
template <> struct Wide<uint32_t> { using type = unsigned long long; };
template <> struct Wide<uint64_t> { using type = unsigned long long; };
template <> struct Wide<float> { using type = double; };
template <> struct Wide<double> { using type = double; };

template <class W> __device__ __forceinline__ uint64_t to_bits(W w) {
  if constexpr (is_fp<W>::value) return (uint64_t)__double_as_longlong((double)w); else return (uint64_t)w;
}
template <class W> __device__ __forceinline__ W from_bits(uint64_t b) {
  if constexpr (is_fp<W>::value) return (W)__longlong_as_double((long long)b); else return (W)b;
}

template <class W>
struct Acc {
  uint64_t isum = 0; double fsum = 0.0; W mn, mx; uint64_t nv = 0, no = 0; bool have = false;
  __device__ __forceinline__ void add(W x) {
    nv++;
    if constexpr (is_fp<W>::value) { fsum += x; if (x == x) { no++; if (!have) { mn = mx = x; have = true; } else { mn = x < mn ? x : mn; mx = x > mx ? x : mx; } } }
    else { isum += (uint64_t)x; fsum += (double)x; no++; if (!have) { mn = mx = x; have = true; } else { mn = x < mn ? x : mn; mx = x > mx ? x : mx; } }
  }
  __device__ __forceinline__ void merge(uint64_t oisum, double ofsum, W omn, W omx, uint64_t onv, uint64_t ono) {
    isum += oisum; fsum += ofsum; nv += onv;
    if (ono) { if (!no) { mn = omn; mx = omx; } else { mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx; } }
    no += ono; have = no > 0;
  }
};

template <class W>
__device__ __forceinline__ void block_reduce_store(Acc<W>& acc, Partial* out) {
  // wave tree
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint64_t oi = shfl_xor_u64(acc.isum, m);
    double of = shfl_xor_f64(acc.fsum, m);
    uint64_t omn = shfl_xor_u64(to_bits<W>(acc.no ? acc.mn : (W)0), m);
    uint64_t omx = shfl_xor_u64(to_bits<W>(acc.no ? acc.mx : (W)0), m);
    uint64_t onv = shfl_xor_u64(acc.nv, m);
    uint64_t ono = shfl_xor_u64(acc.no, m);
    acc.merge(oi, of, from_bits<W>(omn), from_bits<W>(omx), onv, ono);
  }
  __shared__ Partial sh[kBlock / 64];
  const int wave = threadIdx.x >> 6;
  if (lane_id() == 0) {
    sh[wave].isum = acc.isum; sh[wave].fsum = acc.fsum; sh[wave].mn = to_bits<W>(acc.no ? acc.mn : (W)0); sh[wave].mx = to_bits<W>(acc.no ? acc.mx : (W)0);
    sh[wave].n_valid = acc.nv; sh[wave].n_ordered = acc.no;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Acc<W> t;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t.merge(sh[w].isum, sh[w].fsum, from_bits<W>(sh[w].mn), from_bits<W>(sh[w].mx), sh[w].n_valid, sh[w].n_ordered);
    out->isum = t.isum; out->fsum = t.fsum; out->mn = to_bits<W>(t.no ? t.mn : (W)0); out->mx = to_bits<W>(t.no ? t.mx : (W)0);
    out->n_valid = t.nv; out->n_ordered = t.no;
  }
}

// Each wave owns runs of 64*V rows so the validity word(s) of a run are wave-uniform.
template <class T>
__global__ __launch_bounds__(kBlock) void reduce_kernel(const T* __restrict__ v, const uint64_t* __restrict__ valid, int64_t n,
                                                        Partial* __restrict__ partials) {
  using W = typename Wide<T>::type;
  constexpr int V = 16 / sizeof(T);
  Acc<W> acc;
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t run = 64 * V;              // rows per wave-iteration
  const int64_t nruns = n / run;
  for (int64_t r = wave; r < nruns; r += nwaves) {
    const int64_t base = r * run + (int64_t)lane * V;
    Pack<T, V> p = load_pack<T, V>(v + base);
    if (valid) {
      // rows base..base+V-1 live in one 64-bit word (V divides 64)
      uint64_t word = valid[base >> 6] >> (base & 63);
#pragma unroll
      for (int j = 0; j < V; j++) if ((word >> j) & 1) acc.add((W)p.v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) acc.add((W)p.v[j]);
    }
  }
  // tail rows
  for (int64_t i = nruns * run + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!valid || ((valid[i >> 6] >> (i & 63)) & 1)) acc.add((W)v[i]);
  }
  block_reduce_store<W>(acc, partials + blockIdx.x);
}

template <class W>
__global__ __launch_bounds__(kBlock) void reduce_finish_kernel(const Partial* __restrict__ partials, int np, Partial* __restrict__ out) {
  Acc<W> acc;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    const Partial& p = partials[i];
    acc.merge(p.isum, p.fsum, from_bits<W>(p.mn), from_bits<W>(p.mx), p.n_valid, p.n_ordered);
  }
  block_reduce_store<W>(acc, out);
}

template <class T>
static ReduceResult reduce_typed(const void* values, const uint64_t* validity, int64_t n) {
  using W = typename Wide<T>::type;
  constexpr int V = 16 / sizeof(T);
  int grid = grid_for(n, kBlock * V * 4);
  Buf partials = dev_alloc(sizeof(Partial) * (size_t)(grid + 1));
  hipLaunchKernelGGL((reduce_kernel<T>), dim3(grid), dim3(kBlock), 0, stream(), (const T*)values, validity, n, partials->as<Partial>());
  hipLaunchKernelGGL((reduce_finish_kernel<W>), dim3(1), dim3(kBlock), 0, stream(), partials->as<Partial>(), grid, partials->as<Partial>() + grid);
  PLX_HIP(hipGetLastError());
  Partial h;
  d2h_sync(&h, partials->as<Partial>() + grid, sizeof(Partial));
  ReduceResult r;
  r.isum = h.isum; r.fsum = h.fsum; r.minmax_lo = h.mn; r.minmax_hi = h.mx; r.n_valid = h.n_valid; r.n_ordered = h.n_ordered;
  return r;
}

ReduceResult reduce_all(int dtype, const void* values, const uint64_t* validity, int64_t n) {
  ProfileScope ps("reduce_all", (uint64_t)n * dtype_width(dtype) + (validity ? (uint64_t)n / 8 : 0), (uint64_t)n);
  if (n == 0) { ReduceResult r; memset(&r, 0, sizeof(r)); return r; }
  switch (dtype) {
    case PLX_I8: return reduce_typed<int8_t>(values, validity, n);
    case PLX_I16: return reduce_typed<int16_t>(values, validity, n);
    case PLX_I32: return reduce_typed<int32_t>(values, validity, n);
    case PLX_I64: return reduce_typed<int64_t>(values, validity, n);
    case PLX_U8: return reduce_typed<uint8_t>(values, validity, n);
    case PLX_U16: return reduce_typed<uint16_t>(values, validity, n);
    case PLX_U32: return reduce_typed<uint32_t>(values, validity, n);
    case PLX_U64: return reduce_typed<uint64_t>(values, validity, n);
    case PLX_F32: return reduce_typed<float>(values, validity, n);
    case PLX_F64: return reduce_typed<double>(values, validity, n);
    default: fail(PLX_ERR_UNSUPPORTED, "reduce: unsupported dtype");
  }
}

}  // namespace k
}  // namespace plx
