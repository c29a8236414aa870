// dev.hpp -- device-side helpers for gfx950 (wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plx {
namespace dev {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
// number of set bits of m strictly below this lane
__device__ __forceinline__ int prefix_rank(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Reads and writes of LDS words that other lanes update concurrently (slot protocols of the LDS tables).  NOT `volatile`: the compiler's address-space inference
// skips volatile accesses, so a volatile access through a pointer derived from the dynamic LDS base became a FLAT instruction -- and every flat load is followed by
// s_waitcnt vmcnt(0), which also waits for the chunk loads in flight (measured: the slot search of the wide-key aggregation was 15 of 18.5 ms this way).  A relaxed
// workgroup-scope atomic access is a plain ds_read / ds_write; LDS operations of a wave are processed in order, lds_order() keeps the COMPILER from reordering them.
// A wave-uniform read of memory this kernel never writes (chunk lists, fills: written by the kernel before), through the CONSTANT address space: the load is
// invariant to the compiler whatever "memory" clobbers and atomics surround it, so with a uniform index it becomes an s_load (scalar cache) instead of a vector
// load + s_waitcnt vmcnt(0) -- which would also wait for every record load in flight.
template <class T> __device__ __forceinline__ T uniform_ld(const T* p, uint64_t i) {
  return reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p))[i];
}
template <class T> __device__ __forceinline__ T lds_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class T> __device__ __forceinline__ void lds_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  return __longlong_as_double((long long)shfl_xor_u64((uint64_t)__double_as_longlong(v), mask));
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl(lo, src, 64);
  hi = __shfl(hi, src, 64);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return v;
}

// 16-byte (or narrower) aligned pack of V elements -> one global_load_dwordx4 per lane
template <class T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };
template <class T, int V> __device__ __forceinline__ Pack<T, V> load_pack(const T* p) { return *reinterpret_cast<const Pack<T, V>*>(p); }
template <class T, int V> __device__ __forceinline__ void store_pack(T* p, const Pack<T, V>& x) { *reinterpret_cast<Pack<T, V>*>(p) = x; }

// Polars total order for floats (comparisons/simd.rs:171-275): NaN == NaN, NaN greatest.
template <class T> struct is_fp { static constexpr bool value = false; };
template <> struct is_fp<float> { static constexpr bool value = true; };
template <> struct is_fp<double> { static constexpr bool value = true; };

template <class T> __device__ __forceinline__ bool tot_eq(T a, T b) {
  if constexpr (is_fp<T>::value) return (a != a && b != b) || a == b;
  else return a == b;
}
template <class T> __device__ __forceinline__ bool tot_lt(T a, T b) {
  if constexpr (is_fp<T>::value) return !((a != a) || a >= b);
  else return a < b;
}
template <class T> __device__ __forceinline__ bool tot_le(T a, T b) {
  if constexpr (is_fp<T>::value) return (b != b) || a <= b;
  else return a <= b;
}
template <class T> __device__ __forceinline__ bool cmp_apply(int op, T a, T b) {
  switch (op) {
    case 0: return tot_eq(a, b);
    case 1: return !tot_eq(a, b);
    case 2: return tot_lt(a, b);
    case 3: return tot_le(a, b);
    case 4: return tot_lt(b, a);
    default: return tot_le(b, a);
  }
}

// min/max ignoring NaN (polars-utils/src/min_max.rs:31-48)
template <class T> __device__ __forceinline__ T min_ign(T a, T b) {
  if constexpr (is_fp<T>::value) { if (a != a) return b; if (b != b) return a; }
  return a < b ? a : b;
}
template <class T> __device__ __forceinline__ T max_ign(T a, T b) {
  if constexpr (is_fp<T>::value) { if (a != a) return b; if (b != b) return a; }
  return a > b ? a : b;
}

}  // namespace dev
}  // namespace plx
