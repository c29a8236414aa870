// kernels_elementwise.hip -- compare -> bitmap, arithmetic, casts, bitmap logic.
// HBM-bound streaming kernels for gfx950: wave64 ballots produce one 64-bit bitmap word
// per wave-iteration (8 output bytes per `__ballot`), value kernels move 16 B per lane.
//
// Reference semantics restated (never the code):
//   compare   polars-compute/src/comparisons/simd.rs:93-286 (total order for floats)
//   arithmetic polars-compute/src/arithmetic/{signed,unsigned,float}.rs
//   bitmap ops polars-expr/src/expressions/binary.rs:110-118
#include "dev.hpp"
#include "kernels.hpp"

namespace plx {
namespace k {

using namespace dev;

int grid_for(int64_t work_items, int items_per_block, int blocks_per_cu) {
  int64_t need = (work_items + items_per_block - 1) / items_per_block;
  int64_t cap = (int64_t)device().cu_count * blocks_per_cu;
  if (need < 1) need = 1;
  return (int)(need < cap ? need : cap);
}

// ------------------------------------------------------------------ compare ---
// One wave = 64 rows per ballot; 4 ballots (256 rows) in flight per wave-iteration.
template <class T, bool SCALAR>
__global__ __launch_bounds__(kBlock) void cmp_kernel(const T* __restrict__ a, const T* __restrict__ b, T s, int op,
                                                     int64_t n, uint64_t* __restrict__ out) {
  const int lane = lane_id();
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = wave * 4; w < nwords; w += nwaves * 4) {
    if ((w + 4) * 64 <= n) {
      T x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { x[u] = a[(w + u) * 64 + lane]; y[u] = SCALAR ? s : b[(w + u) * 64 + lane]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        uint64_t m = ballot(cmp_apply<T>(op, x[u], y[u]));
        if (lane == u) out[w + u] = m;
      }
    } else {
      for (int u = 0; u < 4 && w + u < nwords; u++) {
        int64_t i = (w + u) * 64 + lane;
        bool r = false;
        if (i < n) { T x = a[i]; T y = SCALAR ? s : b[i]; r = cmp_apply<T>(op, x, y); }
        uint64_t m = ballot(r);
        if (lane == 0) out[w + u] = m;
      }
    }
  }
}

template <class T>
static void cmp_launch(int op, const void* a, const void* b, plx_scalar s, int64_t n, uint64_t* out) {
  if (n == 0) return;
  T sv; memcpy(&sv, &s, sizeof(T));
  int grid = grid_for(n, kBlock * 4);
  if (b) hipLaunchKernelGGL((cmp_kernel<T, false>), dim3(grid), dim3(kBlock), 0, stream(), (const T*)a, (const T*)b, sv, op, n, out);
  else hipLaunchKernelGGL((cmp_kernel<T, true>), dim3(grid), dim3(kBlock), 0, stream(), (const T*)a, (const T*)nullptr, sv, op, n, out);
}

#define PLX_DISPATCH(dt, M)                 \
  switch (dt) {                             \
    case PLX_I8: M(int8_t); break;          \
    case PLX_I16: M(int16_t); break;        \
    case PLX_I32: M(int32_t); break;        \
    case PLX_I64: M(int64_t); break;        \
    case PLX_U8: M(uint8_t); break;         \
    case PLX_U16: M(uint16_t); break;       \
    case PLX_U32: M(uint32_t); break;       \
    case PLX_U64: M(uint64_t); break;       \
    case PLX_F32: M(float); break;          \
    case PLX_F64: M(double); break;         \
    default: fail(PLX_ERR_UNSUPPORTED, std::string("unsupported dtype ") + dtype_name(dt)); \
  }

void cmp(int dtype, int op, const void* a, const void* b, plx_scalar s, int64_t n, uint64_t* out_bits) {
  ProfileScope ps("cmp_bitmap", (uint64_t)n * dtype_width(dtype) * (b ? 2 : 1) + (uint64_t)n / 8, (uint64_t)n);
#define M(T) cmp_launch<T>(op, a, b, s, n, out_bits)
  PLX_DISPATCH(dtype, M)
#undef M
  PLX_HIP(hipGetLastError());
}

// --------------------------------------------------------------- arithmetic ---
template <class T> struct uns { using type = T; };
template <> struct uns<int8_t> { using type = uint8_t; };
template <> struct uns<int16_t> { using type = uint16_t; };
template <> struct uns<int32_t> { using type = uint32_t; };
template <> struct uns<int64_t> { using type = uint64_t; };

template <class T> __device__ __forceinline__ T w_add(T a, T b) {
  if constexpr (is_fp<T>::value) return a + b;
  else { using U = typename uns<T>::type; return (T)(U)((U)a + (U)b); }
}
template <class T> __device__ __forceinline__ T w_sub(T a, T b) {
  if constexpr (is_fp<T>::value) return a - b;
  else { using U = typename uns<T>::type; return (T)(U)((U)a - (U)b); }
}
template <class T> __device__ __forceinline__ T w_mul(T a, T b) {
  if constexpr (is_fp<T>::value) return a * b;
  else if constexpr (sizeof(T) < 4) return (T)(uint32_t)((uint32_t)a * (uint32_t)b);
  else { using U = typename uns<T>::type; return (T)(U)((U)a * (U)b); }
}
// Python-style floor div / mod; (0, 0) when b == 0 (polars-utils/src/floor_divmod.rs)
template <class T> __device__ __forceinline__ void floor_divmod(T a, T b, T& d, T& m) {
  if constexpr (is_fp<T>::value) {
    d = floor(a / b); m = a - b * d;
  } else if constexpr (((T)-1) > (T)0) {  // unsigned
    if (b == 0) { d = 0; m = 0; return; }
    d = a / b; m = a % b;
  } else {
    if (b == 0) { d = 0; m = 0; return; }
    if (b == (T)-1) { d = w_sub<T>((T)0, a); m = 0; return; }
    d = a / b; m = a % b;
    if (m != 0 && ((a < 0) != (b < 0))) { d -= 1; m += b; }
  }
}

template <class T, class O>
__device__ __forceinline__ O arith_apply(int op, T x, T y) {
  switch (op) {
    case PLX_ADD: return (O)w_add<T>(x, y);
    case PLX_SUB: return (O)w_sub<T>(x, y);
    case PLX_MUL: return (O)w_mul<T>(x, y);
    case PLX_TRUE_DIV:
      if constexpr (is_fp<T>::value) return (O)(x / y);
      else return (O)((double)x / (double)y);
    case PLX_FLOOR_DIV: { T d, m; floor_divmod<T>(x, y, d, m); return (O)d; }
    default: { T d, m; floor_divmod<T>(x, y, d, m); return (O)m; }
  }
}

// MODE 0: a[i] op b[i]; 1: a[i] op s; 2: s op a[i].  `inv_mode` (col / scalar):
// floats and int true-div multiply by the precomputed reciprocal (float.rs:113-115,
// signed.rs:218-221); float floor-div/mod by scalar use it too (float.rs:75-96).
template <class T, class O, int MODE>
__global__ __launch_bounds__(kBlock) void arith_kernel(const T* __restrict__ a, const T* __restrict__ b, T s, O sinv, int op,
                                                       int64_t n, O* __restrict__ out) {
  constexpr int V = (16 / sizeof(T)) > 0 ? (16 / sizeof(T)) : 1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n / V;
  auto one = [&](T x, T y) -> O {
    if constexpr (MODE == 1) {
      if (op == PLX_TRUE_DIV) {
        if constexpr (is_fp<T>::value) return (O)(x * (T)sinv); else return (O)((double)x * (double)sinv);
      }
      if constexpr (is_fp<T>::value) {
        if (op == PLX_FLOOR_DIV) return (O)floor(x * (T)sinv);
        if (op == PLX_MOD) return (O)(x - y * floor(x * (T)sinv));
      }
    }
    return arith_apply<T, O>(op, x, y);
  };
  for (int64_t v = tid; v < nvec; v += nthreads) {
    Pack<T, V> xa = load_pack<T, V>(a + v * V), xb;
    if constexpr (MODE == 0) xb = load_pack<T, V>(b + v * V);
    Pack<O, V> r;
#pragma unroll
    for (int j = 0; j < V; j++) {
      T x = MODE == 2 ? s : xa.v[j];
      T y = MODE == 0 ? xb.v[j] : (MODE == 1 ? s : xa.v[j]);
      r.v[j] = one(x, y);
    }
    store_pack<O, V>(out + v * V, r);
  }
  for (int64_t i = nvec * V + tid; i < n; i += nthreads) {
    T x = MODE == 2 ? s : a[i];
    T y = MODE == 0 ? b[i] : (MODE == 1 ? s : a[i]);
    out[i] = one(x, y);
  }
}

template <class T, class O>
static void arith_launch_o(int op, int mode, const void* a, const void* b, plx_scalar s, int64_t n, void* out) {
  if (n == 0) return;
  T sv; memcpy(&sv, &s, sizeof(T));
  O sinv = (O)0;
  if (mode == 1) {
    if constexpr (is_fp<T>::value) sinv = (O)((T)1 / sv); else sinv = (O)(1.0 / (double)sv);
  }
  constexpr int V = (16 / sizeof(T)) > 0 ? (16 / sizeof(T)) : 1;
  int grid = grid_for(n, kBlock * V * 2);
  switch (mode) {
    case 0: hipLaunchKernelGGL((arith_kernel<T, O, 0>), dim3(grid), dim3(kBlock), 0, stream(), (const T*)a, (const T*)b, sv, sinv, op, n, (O*)out); break;
    case 1: hipLaunchKernelGGL((arith_kernel<T, O, 1>), dim3(grid), dim3(kBlock), 0, stream(), (const T*)a, (const T*)nullptr, sv, sinv, op, n, (O*)out); break;
    default: hipLaunchKernelGGL((arith_kernel<T, O, 2>), dim3(grid), dim3(kBlock), 0, stream(), (const T*)a, (const T*)nullptr, sv, sinv, op, n, (O*)out); break;
  }
}
template <class T>
static void arith_launch(int op, int mode, const void* a, const void* b, plx_scalar s, int64_t n, void* out) {
  if constexpr (is_fp<T>::value) arith_launch_o<T, T>(op, mode, a, b, s, n, out);
  else {
    if (op == PLX_TRUE_DIV) arith_launch_o<T, double>(op, mode, a, b, s, n, out);
    else arith_launch_o<T, T>(op, mode, a, b, s, n, out);
  }
}

void arith(int dtype, int op, int mode, const void* a, const void* b, plx_scalar s, int64_t n, void* out) {
  int ow = (op == PLX_TRUE_DIV && !dtype_is_float(dtype)) ? 8 : dtype_width(dtype);
  ProfileScope ps("arith", (uint64_t)n * (dtype_width(dtype) * (mode == 0 ? 2 : 1) + ow), (uint64_t)n);
#define M(T) arith_launch<T>(op, mode, a, b, s, n, out)
  PLX_DISPATCH(dtype, M)
#undef M
  PLX_HIP(hipGetLastError());
}

// --------------------------------------------------------------------- cast ---
template <class F, class T> __device__ __forceinline__ bool cast_one(F x, T& out) {
  if constexpr (is_fp<T>::value) { out = (T)x; return true; }
  else if constexpr (is_fp<F>::value) {
    // float -> int: truncate toward zero; NaN / out of range -> null
    double t = trunc((double)x);
    constexpr bool tsigned = ((T)-1) < (T)0;
    double lo = tsigned ? -ldexp(1.0, 8 * (int)sizeof(T) - 1) : 0.0;
    double hi = tsigned ? ldexp(1.0, 8 * (int)sizeof(T) - 1) : ldexp(1.0, 8 * (int)sizeof(T));
    bool ok = (t >= lo) && (t < hi);  // false for NaN
    out = ok ? (T)t : (T)0;
    return ok;
  } else {
    // int -> int: value must be representable
    constexpr bool fsigned = ((F)-1) < (F)0, tsigned = ((T)-1) < (T)0;
    bool ok;
    if constexpr (fsigned && tsigned) { long long v = (long long)x; ok = v == (long long)(T)v; }
    else if constexpr (fsigned && !tsigned) { long long v = (long long)x; ok = v >= 0 && (unsigned long long)v == (unsigned long long)(T)(unsigned long long)v; }
    else if constexpr (!fsigned && tsigned) { unsigned long long v = (unsigned long long)x; ok = v <= (unsigned long long)((((unsigned long long)1) << (8 * sizeof(T) - 1)) - 1); }
    else { unsigned long long v = (unsigned long long)x; ok = v == (unsigned long long)(T)v; }
    out = ok ? (T)x : (T)0;
    return ok;
  }
}

template <class F, class T>
__global__ __launch_bounds__(kBlock) void cast_kernel(const F* __restrict__ in, int64_t n, T* __restrict__ out, uint64_t* __restrict__ ok_bits) {
  const int lane = lane_id();
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    int64_t i = w * 64 + lane;
    bool ok = false;
    if (i < n) { T o; ok = cast_one<F, T>(in[i], o); out[i] = o; }
    uint64_t m = ballot(ok);
    if (ok_bits && lane == 0) ok_bits[w] = m;
  }
}

template <class F>
static void cast_from(int to, const void* in, int64_t n, void* out, uint64_t* ok_bits) {
  if (n == 0) return;
  int grid = grid_for(n, kBlock);
#define M(T) hipLaunchKernelGGL((cast_kernel<F, T>), dim3(grid), dim3(kBlock), 0, stream(), (const F*)in, n, (T*)out, ok_bits)
  PLX_DISPATCH(to, M)
#undef M
}
void cast(int from, int to, const void* in, int64_t n, void* out, uint64_t* ok_bits) {
  ProfileScope ps("cast", (uint64_t)n * (dtype_width(from) + dtype_width(to)), (uint64_t)n);
#define M(F) cast_from<F>(to, in, n, out, ok_bits)
  PLX_DISPATCH(from, M)
#undef M
  PLX_HIP(hipGetLastError());
}

template <class T>
__global__ __launch_bounds__(kBlock) void cast_bool_kernel(const uint64_t* __restrict__ bits, int64_t n, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (T)((bits[i >> 6] >> (i & 63)) & 1);
}
void cast_from_bool(const uint64_t* bits, int to, int64_t n, void* out) {
  if (n == 0) return;
  int grid = grid_for(n, kBlock * 4);
#define M(T) hipLaunchKernelGGL((cast_bool_kernel<T>), dim3(grid), dim3(kBlock), 0, stream(), bits, n, (T*)out)
  PLX_DISPATCH(to, M)
#undef M
  PLX_HIP(hipGetLastError());
}

// ------------------------------------------------------------- bitmap logic ---
// op: 0 and, 1 or, 2 xor, 3 not(a). Pad bits beyond n_bits are cleared.
__global__ __launch_bounds__(kBlock) void bitmap_op_kernel(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                           int64_t n_bits, uint64_t* __restrict__ out) {
  const int64_t nwords = (n_bits + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = a[w], r;
    switch (op) {
      case 0: r = x & b[w]; break;
      case 1: r = x | b[w]; break;
      case 2: r = x ^ b[w]; break;
      default: r = ~x; break;
    }
    if (w == nwords - 1 && (n_bits & 63)) r &= (~0ull) >> (64 - (n_bits & 63));
    out[w] = r;
  }
}
void bitmap_op(int op, const uint64_t* a, const uint64_t* b, int64_t n_bits, uint64_t* out) {
  if (n_bits == 0) return;
  int grid = grid_for((n_bits + 63) / 64, kBlock);
  hipLaunchKernelGGL(bitmap_op_kernel, dim3(grid), dim3(kBlock), 0, stream(), op, a, b, n_bits, out);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void bitmap_and3_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                             const uint64_t* __restrict__ c, int64_t n_bits, uint64_t* __restrict__ out) {
  const int64_t nwords = (n_bits + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = ~0ull;
    if (a) r &= a[w];
    if (b) r &= b[w];
    if (c) r &= c[w];
    if (w == nwords - 1 && (n_bits & 63)) r &= (~0ull) >> (64 - (n_bits & 63));
    out[w] = r;
  }
}
void bitmap_and3(const uint64_t* a, const uint64_t* b, const uint64_t* c, int64_t n_bits, uint64_t* out) {
  if (n_bits == 0) return;
  int grid = grid_for((n_bits + 63) / 64, kBlock);
  hipLaunchKernelGGL(bitmap_and3_kernel, dim3(grid), dim3(kBlock), 0, stream(), a, b, c, n_bits, out);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void bool_kleene_kernel(int op, const uint64_t* __restrict__ lv, const uint64_t* __restrict__ lval,
                                                             const uint64_t* __restrict__ rv, const uint64_t* __restrict__ rval, int64_t n_bits,
                                                             uint64_t* __restrict__ out_v, uint64_t* __restrict__ out_valid) {
  const int64_t nwords = (n_bits + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t a = lv[w], b = rv[w], av = lval ? lval[w] : ~0ull, bv = rval ? rval[w] : ~0ull, v, ok;
    if (op == 0) { v = a & b; ok = (~b & bv) | (~a & av) | (a & av & b & bv); }
    else { v = a | b; ok = (a & av) | (b & bv) | (~a & av & ~b & bv); }
    if (w == nwords - 1 && (n_bits & 63)) { uint64_t m = (~0ull) >> (64 - (n_bits & 63)); v &= m; ok &= m; }
    out_v[w] = v;
    if (out_valid) out_valid[w] = ok;
  }
}
void bool_kleene(int op, const uint64_t* lv, const uint64_t* lvalid, const uint64_t* rv, const uint64_t* rvalid, int64_t n_bits,
                 uint64_t* out_v, uint64_t* out_valid) {
  if (n_bits == 0) return;
  hipLaunchKernelGGL(bool_kleene_kernel, dim3(grid_for((n_bits + 63) / 64, kBlock)), dim3(kBlock), 0, stream(), op, lv, lvalid, rv, rvalid, n_bits, out_v, out_valid);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void bitmap_blit_kernel(unsigned long long* __restrict__ dst, int64_t dst_off, const uint64_t* __restrict__ src, int64_t n_bits) {
  const int64_t nwords = (n_bits + 63) >> 6;
  const int sh = (int)(dst_off & 63);
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = src[w];
    if (w == nwords - 1 && (n_bits & 63)) x &= (~0ull) >> (64 - (n_bits & 63));
    if (!x) continue;
    const int64_t dw = (dst_off >> 6) + w;
    atomicOr(&dst[dw], (unsigned long long)(x << sh));
    if (sh && (x >> (64 - sh))) atomicOr(&dst[dw + 1], (unsigned long long)(x >> (64 - sh)));
  }
}
void bitmap_blit(uint64_t* dst, int64_t dst_off, const uint64_t* src, int64_t n_bits) {
  if (n_bits == 0) return;
  hipLaunchKernelGGL(bitmap_blit_kernel, dim3(grid_for((n_bits + 63) / 64, kBlock)), dim3(kBlock), 0, stream(), (unsigned long long*)dst, dst_off, src, n_bits);
  PLX_HIP(hipGetLastError());
}

template <class W>
__global__ __launch_bounds__(kBlock) void fill_kernel(W* out, W v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = v;
}
void fill(int width, void* out, uint64_t pattern, int64_t n) {
  if (n == 0) return;
  int grid = grid_for(n, kBlock * 4);
  switch (width) {
    case 1: hipLaunchKernelGGL((fill_kernel<uint8_t>), dim3(grid), dim3(kBlock), 0, stream(), (uint8_t*)out, (uint8_t)pattern, n); break;
    case 2: hipLaunchKernelGGL((fill_kernel<uint16_t>), dim3(grid), dim3(kBlock), 0, stream(), (uint16_t*)out, (uint16_t)pattern, n); break;
    case 4: hipLaunchKernelGGL((fill_kernel<uint32_t>), dim3(grid), dim3(kBlock), 0, stream(), (uint32_t*)out, (uint32_t)pattern, n); break;
    case 8: hipLaunchKernelGGL((fill_kernel<uint64_t>), dim3(grid), dim3(kBlock), 0, stream(), (uint64_t*)out, (uint64_t)pattern, n); break;
    default: fail(PLX_ERR_INVALID, "fill: bad width");
  }
  PLX_HIP(hipGetLastError());
}

// fill_null(literal): out[i] = valid(i) ? in[i] : v   (per-node path; fused programs use OP_IFNULL)
template <class W>
__global__ __launch_bounds__(kBlock) void fill_null_kernel(const W* __restrict__ in, const uint64_t* __restrict__ validity, W v, int64_t n, W* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = ((validity[i >> 6] >> (i & 63)) & 1) ? in[i] : v;
}
__global__ __launch_bounds__(kBlock) void fill_null_bool_kernel(const uint64_t* __restrict__ in, const uint64_t* __restrict__ validity, int v, int64_t n_words, uint64_t* __restrict__ out) {
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x)
    out[w] = (in[w] & validity[w]) | (v ? ~validity[w] : 0ull);
}
void fill_null(int width, const void* in, const uint64_t* validity, uint64_t pattern, int64_t n, void* out) {
  if (n == 0) return;
  const int grid = grid_for(n, kBlock * 4);
  switch (width) {
    case 0: hipLaunchKernelGGL(fill_null_bool_kernel, dim3(grid_for((n + 63) / 64, kBlock * 4)), dim3(kBlock), 0, stream(), (const uint64_t*)in, validity, (int)(pattern & 1), (n + 63) / 64, (uint64_t*)out); break;
    case 1: hipLaunchKernelGGL((fill_null_kernel<uint8_t>), dim3(grid), dim3(kBlock), 0, stream(), (const uint8_t*)in, validity, (uint8_t)pattern, n, (uint8_t*)out); break;
    case 2: hipLaunchKernelGGL((fill_null_kernel<uint16_t>), dim3(grid), dim3(kBlock), 0, stream(), (const uint16_t*)in, validity, (uint16_t)pattern, n, (uint16_t*)out); break;
    case 4: hipLaunchKernelGGL((fill_null_kernel<uint32_t>), dim3(grid), dim3(kBlock), 0, stream(), (const uint32_t*)in, validity, (uint32_t)pattern, n, (uint32_t*)out); break;
    case 8: hipLaunchKernelGGL((fill_null_kernel<uint64_t>), dim3(grid), dim3(kBlock), 0, stream(), (const uint64_t*)in, validity, (uint64_t)pattern, n, (uint64_t*)out); break;
    default: fail(PLX_ERR_INVALID, "fill_null: bad width");
  }
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void popcount_kernel(const uint64_t* __restrict__ a, int64_t n_bits, unsigned long long* __restrict__ out) {
  const int64_t nwords = (n_bits + 63) >> 6;
  uint64_t acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = a[w];
    if (w == nwords - 1 && (n_bits & 63)) x &= (~0ull) >> (64 - (n_bits & 63));
    acc += (uint64_t)popc64(x);
  }
  acc = wave_sum_u64(acc);
  if (lane_id() == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}
int64_t bitmap_popcount(const uint64_t* a, int64_t n_bits) {
  if (n_bits == 0) return 0;
  Buf d = dev_alloc_zero(8);
  int grid = grid_for((n_bits + 63) / 64, kBlock * 4);
  hipLaunchKernelGGL(popcount_kernel, dim3(grid), dim3(kBlock), 0, stream(), a, n_bits, d->as<unsigned long long>());
  PLX_HIP(hipGetLastError());
  uint64_t h = 0;
  d2h_sync(&h, d->ptr, 8);
  return (int64_t)h;
}

__global__ __launch_bounds__(kBlock) void iota_kernel(uint32_t* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}
void fill_iota_u32(uint32_t* out, int64_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), out, n);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void pack_kernel(PackBatch b, uint8_t* __restrict__ out) {
  const int j = blockIdx.x;
  const uint8_t* src = reinterpret_cast<const uint8_t*>(b.src[j]);
  uint8_t* dst = out + b.off[j];
  for (uint32_t i = threadIdx.x; i < b.bytes[j]; i += kBlock) dst[i] = src[i];
}
void pack_buffers(const PackBatch& b, void* staging) {
  if (b.n == 0) return;
  hipLaunchKernelGGL(pack_kernel, dim3(b.n), dim3(kBlock), 0, stream(), b, reinterpret_cast<uint8_t*>(staging));
  PLX_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace plx
