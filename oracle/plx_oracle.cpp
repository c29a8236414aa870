// plx_oracle.cpp -- CPU restatement of the Polars hot path (TEST INFRASTRUCTURE ONLY).
//
// This file is the parity oracle and the timed "port" CPU baseline.  It is never
// linked into, imported by, or called from the product (polars_amd/); only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// The reference (pola-rs/polars 0.55.1, Rust nightly) cannot be compiled or
// imported in the build container (no rustc/cargo, no polars wheel), so this is a
// restatement of the reference algorithms from source, pinned against the golden
// vectors of the reference's own tests (tests/test_oracle_golden.py transcribes them,
// SURVEY.md section 4 / 8c).  Every function cites the reference file:line it follows
// (paths relative to the polars tree, crates/ prefix omitted where obvious).
//
// Layout contract: Arrow. values = contiguous little-endian; bitmaps LSB-first,
// bit i of byte i/8.  All bitmaps here start at bit offset 0.
//
// Build: see oracle/Makefile (g++ -O3 -march=native -shared -fPIC -pthread).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

enum DT { BOOL = 0, I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 };
enum CmpOp { EQ = 0, NE, LT, LE, GT, GE };
enum ArithOp { ADD = 0, SUB, MUL, TRUE_DIV, FLOOR_DIV, MOD };
enum AggOp { AGG_SUM = 0, AGG_MEAN, AGG_MIN, AGG_MAX, AGG_COUNT, AGG_LEN, AGG_FIRST };

int g_threads = 1;

inline bool getbit(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
inline void setbit(uint8_t* b, int64_t i, bool v) {
  if (v) b[i >> 3] |= uint8_t(1u << (i & 7));
  else b[i >> 3] &= uint8_t(~(1u << (i & 7)));
}

int dt_width(int dt) {
  switch (dt) {
    case I8: case U8: return 1;
    case I16: case U16: return 2;
    case I32: case U32: case F32: return 4;
    case I64: case U64: case F64: return 8;
    default: return 0;
  }
}

// Rayon-pool stand-in (polars-core/src/runtime.rs:8): run f(begin, end, tid) over
// `align`-aligned contiguous ranges on g_threads std::threads.
template <class F>
void parallel_ranges(int64_t n, int64_t align, F f) {
  int nt = g_threads;
  if (nt <= 1 || n < 4096) { f(int64_t(0), n, 0); return; }
  int64_t per = (n + nt - 1) / nt;
  per = (per + align - 1) / align * align;
  std::vector<std::thread> ts;
  for (int t = 0; t < nt; t++) {
    int64_t b = std::min<int64_t>(n, per * t), e = std::min<int64_t>(n, per * (t + 1));
    if (b >= e) break;
    ts.emplace_back([=] { f(b, e, t); });
  }
  for (auto& t : ts) t.join();
}

template <class F>
void parallel_tasks(int n_tasks, F f) {
  int nt = std::min(g_threads, n_tasks);
  if (nt <= 1) { for (int i = 0; i < n_tasks; i++) f(i); return; }
  std::atomic<int> next{0};
  std::vector<std::thread> ts;
  for (int t = 0; t < nt; t++)
    ts.emplace_back([&] { for (;;) { int i = next.fetch_add(1); if (i >= n_tasks) break; f(i); } });
  for (auto& t : ts) t.join();
}

#define DISPATCH_NUMERIC(dt, MACRO)            \
  switch (dt) {                                \
    case I8: { MACRO(int8_t) } break;          \
    case I16: { MACRO(int16_t) } break;        \
    case I32: { MACRO(int32_t) } break;        \
    case I64: { MACRO(int64_t) } break;        \
    case U8: { MACRO(uint8_t) } break;         \
    case U16: { MACRO(uint16_t) } break;       \
    case U32: { MACRO(uint32_t) } break;       \
    case U64: { MACRO(uint64_t) } break;       \
    case F32: { MACRO(float) } break;          \
    case F64: { MACRO(double) } break;         \
    default: return 1;                         \
  }

// ---------------------------------------------------------------------------
// Comparisons.  polars-compute/src/comparisons/simd.rs:93-169 (ints: plain
// ==, !=, <, <=), :171-275 (floats: total order, NaN == NaN, NaN is the maximum),
// gt/ge via swapped lt/le (comparisons/mod.rs:61-66).  Kernels ignore validity.
// ---------------------------------------------------------------------------
template <class T> inline bool tot_eq(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return (a != a && b != b) || a == b;
  else return a == b;
}
template <class T> inline bool tot_lt(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return !((a != a) || a >= b);  // simd.rs:222-229
  else return a < b;
}
template <class T> inline bool tot_le(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return (b != b) || a <= b;  // simd.rs:231-238
  else return a <= b;
}
template <class T> inline bool cmp_apply(int op, T a, T b) {
  switch (op) {
    case EQ: return tot_eq(a, b);
    case NE: return !tot_eq(a, b);
    case LT: return tot_lt(a, b);
    case LE: return tot_le(a, b);
    case GT: return tot_lt(b, a);
    default: return tot_le(b, a);
  }
}

template <class T>
void cmp_impl(int op, const T* l, const T* r, bool rs, int64_t n, uint8_t* out) {
  parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) setbit(out, i, cmp_apply<T>(op, l[i], rs ? r[0] : r[i]));
  });
}

// ---------------------------------------------------------------------------
// Arithmetic.  polars-compute/src/arithmetic/signed.rs:12-234, unsigned.rs,
// float.rs:8-126; FloorDivMod polars-utils/src/floor_divmod.rs.
// mode: 0 = col OP col, 1 = col OP scalar, 2 = scalar OP col.
// `extra_valid` (may be null) receives the "rhs != 0" mask of integer floor-div/mod
// (signed.rs:35-70); has_extra tells whether it was written.
// ---------------------------------------------------------------------------
template <class T> inline T wrap_add(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return a + b;
  else { using U = typename std::make_unsigned<T>::type; return T(U(U(a) + U(b))); }
}
template <class T> inline T wrap_sub(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return a - b;
  else { using U = typename std::make_unsigned<T>::type; return T(U(U(a) - U(b))); }
}
template <class T> inline T wrap_mul(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return a * b;
  else if constexpr (sizeof(T) < 4) { return T(uint32_t(uint32_t(a) * uint32_t(b))); }
  else { using U = typename std::make_unsigned<T>::type; return T(U(U(a) * U(b))); }
}
// floor_divmod.rs: signed -> Python-style floor; (0,0) when other == 0.
template <class T> inline void floor_divmod(T a, T b, T& d, T& m) {
  if constexpr (std::is_floating_point<T>::value) {
    d = std::floor(a / b); m = a - b * d;
  } else if constexpr (std::is_unsigned<T>::value) {
    if (b == 0) { d = 0; m = 0; return; }
    d = a / b; m = a % b;
  } else {
    if (b == 0) { d = 0; m = 0; return; }
    if (b == T(-1)) { d = wrap_sub<T>(T(0), a); m = 0; return; }  // wrapping_div(MIN,-1) = MIN
    d = a / b; m = a % b;
    if (m != 0 && ((a < 0) != (b < 0))) { d -= 1; m += b; }
  }
}

template <class T>
int arith_impl(int op, const T* l, const T* r, int mode, int64_t n, void* outv, uint8_t* extra_valid, int* has_extra) {
  constexpr bool is_f = std::is_floating_point<T>::value;
  *has_extra = 0;
  auto L = [&](int64_t i) { return mode == 2 ? l[0] : l[i]; };
  auto R = [&](int64_t i) { return mode == 1 ? r[0] : r[i]; };
  if (op == TRUE_DIV) {
    if constexpr (is_f) {
      T* out = (T*)outv;
      if (mode == 1) {
        // float.rs:113-115: col / s == prim_wrapping_mul_scalar(col, 1/s) (:64-73)
        T inv = T(1) / r[0];
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
          for (int64_t i = b; i < e; i++) out[i] = (inv == T(1)) ? l[i] : (inv == T(-1)) ? -l[i] : l[i] * inv;
        });
      } else {
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) out[i] = L(i) / R(i); });
      }
    } else {
      double* out = (double*)outv;
      if (mode == 1) {
        double inv = 1.0 / (double)r[0];  // signed.rs:218-221
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) out[i] = (double)l[i] * inv; });
      } else {
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
          for (int64_t i = b; i < e; i++) out[i] = (double)L(i) / (double)R(i);
        });
      }
    }
    return 0;
  }
  T* out = (T*)outv;
  switch (op) {
    case ADD:
      parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) out[i] = wrap_add<T>(L(i), R(i)); });
      break;
    case SUB:
      if (is_f && mode == 1) {  // float.rs:50-55: x - s == x + (-s); s == 0 -> unchanged
        T s = r[0];
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
          for (int64_t i = b; i < e; i++) out[i] = (s == T(0)) ? l[i] : wrap_add<T>(l[i], T(-s));
        });
      } else {
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) out[i] = wrap_sub<T>(L(i), R(i)); });
      }
      break;
    case MUL:
      // signed.rs:84-101 (x * 2^k as shift) is value-identical to the wrapping multiply.
      parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) out[i] = wrap_mul<T>(L(i), R(i)); });
      break;
    case FLOOR_DIV:
    case MOD: {
      if constexpr (is_f) {
        if (mode == 1) {  // float.rs:75-78, 93-96: uses inv = 1/s
          T s = r[0], inv = T(1) / s;
          parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; i++) {
              T fl = std::floor(l[i] * inv);
              out[i] = (op == FLOOR_DIV) ? fl : l[i] - s * fl;
            }
          });
        } else {
          parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; i++) {
              T d, m; floor_divmod<T>(L(i), R(i), d, m);
              out[i] = (op == FLOOR_DIV) ? d : m;
            }
          });
        }
      } else {
        // signed.rs:35-70 / unsigned.rs: value (0 where rhs==0) + validity mask rhs != 0.
        // Scalar forms (signed.rs:103-139,176-204: strength-reduced) are value-identical.
        *has_extra = 1;
        parallel_ranges(n, 64, [&](int64_t b, int64_t e, int) {
          for (int64_t i = b; i < e; i++) {
            T d, m; floor_divmod<T>(L(i), R(i), d, m);
            out[i] = (op == FLOOR_DIV) ? d : m;
            if (extra_valid) setbit(extra_valid, i, R(i) != 0);
          }
        });
      }
    } break;
    default: return 1;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Filter.  polars-compute/src/filter/mod.rs:18-28 (null mask -> false),
// scalar.rs:85-138 (64-row blocks: m==0 skip, m==!0 memcpy, popcnt<=16 sparse
// :9-27, else dense :29-46), validity via boolean.rs:54-101.
// ---------------------------------------------------------------------------
inline uint64_t load_mask64(const uint8_t* bytes, int64_t nbytes_avail) {
  uint64_t m = 0;
  memcpy(&m, bytes, (size_t)std::min<int64_t>(8, nbytes_avail));
  return m;
}

template <class T>
int64_t scalar_filter(const T* values, const uint8_t* mask, int64_t n, T* out) {
  int64_t w = 0, i = 0;
  int64_t nbytes = (n + 7) / 8;
  for (; i + 64 <= n; i += 64) {
    uint64_t m = load_mask64(mask + i / 8, nbytes - i / 8);
    if (m == 0) continue;
    if (m == ~0ull) { memcpy(out + w, values + i, 64 * sizeof(T)); w += 64; continue; }
    int pc = __builtin_popcountll(m);
    if (pc <= 16) {  // scalar_sparse_filter64
      uint64_t mm = m; int64_t ww = w;
      while (mm) { out[ww++] = values[i + __builtin_ctzll(mm)]; mm &= mm - 1; }
    } else {  // scalar_dense_filter64
      int64_t ww = w;
      for (int j = 0; j < 64; j++) { out[ww] = values[i + j]; ww += (m >> j) & 1; }
    }
    w += pc;
  }
  if (i < n) {
    uint64_t m = load_mask64(mask + i / 8, nbytes - i / 8) & ((1ull << (n - i)) - 1);
    while (m) { out[w++] = values[i + __builtin_ctzll(m)]; m &= m - 1; }
  }
  return w;
}

int64_t filter_bits(const uint8_t* bits, const uint8_t* mask, int64_t n, uint8_t* out) {
  int64_t w = 0;
  for (int64_t i = 0; i < n; i++)
    if (getbit(mask, i)) { setbit(out, w, getbit(bits, i)); w++; }
  return w;
}

// ---------------------------------------------------------------------------
// Float sum.  polars-compute/src/float_sum.rs: STRIPE=16, block=128 (:13-14),
// block sum :159-172 (non-simd form; the simd form :77-88 adds in the same order),
// horizontal sum :44-63, pairwise recursion :193-215, remainder = FIRST len%128
// elements summed naively and added last (:253-264); masked variant :217-289.
// ---------------------------------------------------------------------------
template <class F> inline F horizontal16(F* v) {
  int width = 16;
  while (width > 4) { for (int j = 0; j < width / 2; j++) v[j] = v[j] + v[width / 2 + j]; width /= 2; }
  return (v[0] + v[2]) + (v[1] + v[3]);
}
template <class T, class F>
F sum_block128(const T* f, const uint8_t* valid, int64_t bit0) {
  F vs[16]; for (int j = 0; j < 16; j++) vs[j] = F(0);
  for (int c = 0; c < 8; c++)
    for (int j = 0; j < 16; j++) {
      int k = c * 16 + j;
      F add = (!valid || getbit(valid, bit0 + k)) ? F(f[k]) : F(0);
      vs[j] = vs[j] + add;
    }
  return horizontal16(vs);
}
template <class T, class F>
F pairwise_sum(const T* f, int64_t n, const uint8_t* valid, int64_t bit0) {
  if (n == 128) return sum_block128<T, F>(f, valid, bit0);
  int64_t blocks = n / 128, left = (blocks / 2) * 128;
  return pairwise_sum<T, F>(f, left, valid, bit0) + pairwise_sum<T, F>(f + left, n - left, valid, bit0 + left);
}
template <class T, class F>
F float_sum(const T* f, int64_t n, const uint8_t* valid, int64_t bit0 = 0) {
  int64_t rem = n % 128;
  F mainsum = (n > rem) ? pairwise_sum<T, F>(f + rem, n - rem, valid, bit0 + rem) : F(0);
  F rest = F(0);
  for (int64_t i = 0; i < rem; i++) rest = rest + ((!valid || getbit(valid, bit0 + i)) ? F(f[i]) : F(0));
  return mainsum + rest;
}

// _split_offsets (polars-core/src/utils/mod.rs): n parts of len/n, the last takes the rest.
std::vector<std::pair<int64_t, int64_t>> split_offsets(int64_t len, int n) {
  std::vector<std::pair<int64_t, int64_t>> v;
  if (n <= 1) { v.push_back({0, len}); return v; }
  int64_t cs = len / n;
  for (int p = 0; p < n; p++) { int64_t off = p * cs; v.push_back({off, p == n - 1 ? len - off : cs}); }
  return v;
}

// polars-utils/src/kahan_sum.rs:35-46
struct Kahan {
  double sum = 0, err = 0;
  inline void add(double x) {
    double y = x - err, ns = sum + y, ne = (ns - sum) - y;
    sum = ns;
    if (std::isfinite(ne)) err = ne;
  }
};
struct KahanF {
  float sum = 0, err = 0;
  inline void add(float x) {
    float y = x - err, ns = sum + y, ne = (ns - sum) - y;
    sum = ns;
    if (std::isfinite(ne)) err = ne;
  }
};

// polars-utils/src/min_max.rs:31-48 (ignore-NaN min/max); ints plain.
template <class T> inline T min_ign(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
  else return a < b ? a : b;
}
template <class T> inline T max_ign(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }
  else return a > b ? a : b;
}

union Scalar { int64_t i; uint64_t u; double f64; float f32; };

template <class T> int sum_out_dtype_of(int dt) {
  // aggregate/mod.rs:27-64 SumCast / sum_output_dtype
  switch (dt) { case I8: case I16: case U8: case U16: return I64; default: return dt; }
}

// Whole-column reductions.  aggregate/mod.rs:86-137 (sum), :240-246 (mean),
// :307-316 (sum_reduce), thread split polars-expr/src/expressions/aggregation.rs:649-691.
template <class T>
int reduce_impl(int dt, int op, const T* v, const uint8_t* valid, int64_t n, Scalar* out, int* out_dt, int* out_valid) {
  constexpr bool is_f = std::is_floating_point<T>::value;
  int64_t nvalid = n;
  if (valid) { nvalid = 0; for (int64_t i = 0; i < n; i++) nvalid += getbit(valid, i); }
  *out_valid = 1; out->u = 0;
  const uint8_t* vmask = (valid && nvalid < n) ? valid : nullptr;  // "filter(|_| null_count > 0)"
  auto splits = (n >= 100000 && g_threads > 1) ? split_offsets(n, g_threads) : split_offsets(n, 1);
  switch (op) {
    case AGG_LEN: *out_dt = U32; out->u = (uint32_t)n; return 0;
    case AGG_COUNT: *out_dt = U32; out->u = (uint32_t)nvalid; return 0;
    case AGG_SUM: {
      *out_dt = sum_out_dtype_of<T>(dt);
      if constexpr (is_f) {
        std::vector<T> parts(splits.size());
        parallel_tasks((int)splits.size(), [&](int p) {
          auto [off, len] = splits[p];
          // sum(): all-null slice -> 0 (aggregate/mod.rs:90-92)
          parts[p] = float_sum<T, T>(v + off, len, vmask, off);
        });
        T total = parts.size() == 1 ? parts[0] : float_sum<T, T>(parts.data(), (int64_t)parts.size(), nullptr);
        if (sizeof(T) == 4) out->f32 = (float)total; else out->f64 = (double)total;
      } else {
        // sum.rs:190-203 wrapping_sum_arr_upcast; u8/u16/i8/i16 -> i64
        using S = typename std::conditional<(sizeof(T) < 4), int64_t, T>::type;
        using US = typename std::make_unsigned<S>::type;
        std::vector<US> parts(splits.size());
        parallel_tasks((int)splits.size(), [&](int p) {
          auto [off, len] = splits[p]; US acc = 0;
          for (int64_t i = off; i < off + len; i++) if (!vmask || getbit(vmask, i)) acc += US(S(v[i]));
          parts[p] = acc;
        });
        US tot = 0; for (auto x : parts) tot += x;
        if (sizeof(S) == 8) out->u = (uint64_t)tot;
        else if (std::is_signed<S>::value) out->i = (int64_t)(S)tot; else out->u = (uint64_t)tot;
      }
      return 0;
    }
    case AGG_MEAN: {
      // mean = _sum_as_f64 / (len - null_count); f32 input keeps f32 output (reduce/mean.rs:29-80)
      *out_dt = (dt == F32) ? F32 : F64;
      if (nvalid == 0) { *out_valid = 0; return 0; }
      std::vector<double> parts(splits.size());
      parallel_tasks((int)splits.size(), [&](int p) {
        auto [off, len] = splits[p];
        parts[p] = float_sum<T, double>(v + off, len, vmask, off);
      });
      double total = parts.size() == 1 ? parts[0] : float_sum<double, double>(parts.data(), (int64_t)parts.size(), nullptr);
      double m = total / (double)nvalid;
      if (dt == F32) out->f32 = (float)m; else out->f64 = m;
      return 0;
    }
    case AGG_MIN:
    case AGG_MAX: {
      // aggregate/mod.rs:139-192 + polars-compute/src/min_max/scalar.rs:23-73: skip nulls,
      // ignore NaN unless every valid value is NaN.
      *out_dt = dt;
      if (nvalid == 0) { *out_valid = 0; return 0; }
      bool have = false; T acc = T(0);
      for (int64_t i = 0; i < n; i++) {
        if (vmask && !getbit(vmask, i)) continue;
        if (!have) { acc = v[i]; have = true; }
        else acc = (op == AGG_MIN) ? min_ign<T>(acc, v[i]) : max_ign<T>(acc, v[i]);
      }
      memcpy(out, &acc, sizeof(T));
      if (!is_f && sizeof(T) < 8) { if (std::is_signed<T>::value) out->i = (int64_t)acc; else out->u = (uint64_t)acc; }
      return 0;
    }
  }
  return 1;
}

// ---------------------------------------------------------------------------
// Group-by.  Keys are handed over as u64 bit representations plus a validity
// bitmap per key column (into_groups.rs:156-186 to_bit_repr; floats are
// canonicalised by the caller exactly as total_ord.rs:40-48: -0 -> +0, one NaN).
// Single key: group_by_threaded_slice (hashing.rs:116-167): every thread scans
// all rows and keeps those with hash_to_partition(dirty_hash(k), n) == tid
// (polars-utils/src/hashing.rs:62-69,124-151); per-group (first, [idx...]).
// Multiple keys: the reference row-encodes then runs the same scheme
// (group_by/mod.rs:88-94); here the tuple of u64 words is the key.
// finish_group_order (hashing.rs:26-73): sorted => groups ordered by first idx.
// ---------------------------------------------------------------------------
const uint64_t RANDOM_ODD = 0x55fbfd6bfc5458e9ull;
inline uint64_t hash_to_partition(uint64_t h, uint64_t n) { return (uint64_t)(((unsigned __int128)h * n) >> 64); }

struct KeyTuple {
  uint64_t w[4];
  uint32_t nullmask;
};

struct Groups {
  int64_t n_rows = 0;
  std::vector<uint32_t> first;
  std::vector<std::vector<uint32_t>> all;
};

struct GroupTable {  // open addressing, linear probing; value = group slot
  std::vector<int64_t> slot;  // -1 empty, else index into keys/firsts
  std::vector<KeyTuple> keys;
  std::vector<uint32_t> first;
  std::vector<std::vector<uint32_t>> all;
  uint64_t mask = 0;
  int nk = 1;
  void init(int nkeys, size_t cap) { nk = nkeys; size_t c = 64; while (c < cap) c <<= 1; slot.assign(c, -1); mask = c - 1; }
  static inline uint64_t mix(const KeyTuple& k, int nk) {
    uint64_t h = k.nullmask * 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < nk; i++) { h ^= k.w[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
    return h;
  }
  inline bool eq(const KeyTuple& a, const KeyTuple& b) const {
    if (a.nullmask != b.nullmask) return false;
    for (int i = 0; i < nk; i++) if (a.w[i] != b.w[i]) return false;
    return true;
  }
  void grow() {
    std::vector<int64_t> ns(slot.size() * 2, -1); uint64_t nm = ns.size() - 1;
    for (size_t g = 0; g < keys.size(); g++) { uint64_t p = mix(keys[g], nk) & nm; while (ns[p] >= 0) p = (p + 1) & nm; ns[p] = (int64_t)g; }
    slot.swap(ns); mask = nm;
  }
  inline void insert(const KeyTuple& k, uint32_t idx) {
    uint64_t p = mix(k, nk) & mask;
    for (;;) {
      int64_t g = slot[p];
      if (g < 0) {
        slot[p] = (int64_t)keys.size(); keys.push_back(k); first.push_back(idx); all.emplace_back(1, idx);
        if (keys.size() * 2 > slot.size()) grow();
        return;
      }
      if (eq(keys[g], k)) { all[g].push_back(idx); return; }
      p = (p + 1) & mask;
    }
  }
};

Groups* build_groups(int nk, const uint64_t* const* keys, const uint8_t* const* valids, int64_t n, int sorted) {
  int np = (n > 1000 && g_threads > 1) ? g_threads : 1;  // into_groups.rs:25-28
  std::vector<GroupTable> tbls(np);
  auto make_key = [&](int64_t i, KeyTuple& k) {
    k.nullmask = 0;
    for (int c = 0; c < nk; c++) {
      bool ok = !valids[c] || getbit(valids[c], i);
      k.w[c] = ok ? keys[c][i] : 0;
      if (!ok) k.nullmask |= 1u << c;
    }
    for (int c = nk; c < 4; c++) k.w[c] = 0;
  };
  parallel_tasks(np, [&](int tid) {
    GroupTable& t = tbls[tid]; t.init(nk, 512);
    KeyTuple k;
    for (int64_t i = 0; i < n; i++) {
      make_key(i, k);
      uint64_t h = (nk == 1 ? k.w[0] : GroupTable::mix(k, nk)) * RANDOM_ODD;
      if (np > 1 && hash_to_partition(h, (uint64_t)np) != (uint64_t)tid) continue;
      t.insert(k, (uint32_t)i);
    }
  });
  Groups* g = new Groups(); g->n_rows = n;
  for (auto& t : tbls) {
    for (size_t j = 0; j < t.first.size(); j++) { g->first.push_back(t.first[j]); g->all.push_back(std::move(t.all[j])); }
  }
  if (sorted) {
    std::vector<size_t> ord(g->first.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return g->first[a] < g->first[b]; });
    std::vector<uint32_t> f2(ord.size()); std::vector<std::vector<uint32_t>> a2(ord.size());
    for (size_t i = 0; i < ord.size(); i++) { f2[i] = g->first[ord[i]]; a2[i] = std::move(g->all[ord[i]]); }
    g->first.swap(f2); g->all.swap(a2);
  }
  return g;
}

// Grouped aggregations.  frame/group_by/aggregations/mod.rs:854-926 (agg_sum:
// ints fold(+) wrapping, floats KahanSum in row order :867-870), :939-1018
// (agg_mean: Kahan f64 / (len - nulls), None when no valid value),
// series/implementations/mod.rs:145-154 (Int8/16,UInt8/16 upcast to Int64 first),
// count: polars-expr/src/expressions/count.rs:60-128 (u32), min/max :184-246.
template <class T>
int group_agg_impl(const Groups* g, int dt, int op, const T* v, const uint8_t* valid, void* outv, uint8_t* out_valid, int* out_dt) {
  constexpr bool is_f = std::is_floating_point<T>::value;
  int64_t G = (int64_t)g->first.size();
  auto isv = [&](uint32_t i) { return !valid || getbit(valid, i); };
  switch (op) {
    case AGG_LEN: { *out_dt = U32; uint32_t* o = (uint32_t*)outv;
      for (int64_t j = 0; j < G; j++) { o[j] = (uint32_t)g->all[j].size(); if (out_valid) setbit(out_valid, j, true); } return 0; }
    case AGG_COUNT: { *out_dt = U32; uint32_t* o = (uint32_t*)outv;
      parallel_ranges(G, 64, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) { uint32_t c = 0; for (uint32_t i : g->all[j]) c += isv(i); o[j] = c; if (out_valid) setbit(out_valid, j, true); } });
      return 0; }
    case AGG_FIRST: { *out_dt = dt; T* o = (T*)outv;
      for (int64_t j = 0; j < G; j++) { o[j] = v[g->first[j]]; if (out_valid) setbit(out_valid, j, isv(g->first[j])); } return 0; }
    case AGG_SUM: {
      *out_dt = sum_out_dtype_of<T>(dt);
      if constexpr (is_f) {
        T* o = (T*)outv;
        parallel_ranges(G, 64, [&](int64_t b, int64_t e, int) {
          for (int64_t j = b; j < e; j++) {
            if constexpr (sizeof(T) == 8) { Kahan k; for (uint32_t i : g->all[j]) if (isv(i)) k.add(v[i]); o[j] = k.sum; }
            else { KahanF k; for (uint32_t i : g->all[j]) if (isv(i)) k.add(v[i]); o[j] = k.sum; }
            if (out_valid) setbit(out_valid, j, true);
          } });
      } else {
        using S = typename std::conditional<(sizeof(T) < 4), int64_t, T>::type;
        using US = typename std::make_unsigned<S>::type;
        S* o = (S*)outv;
        parallel_ranges(G, 64, [&](int64_t b, int64_t e, int) {
          for (int64_t j = b; j < e; j++) { US acc = 0; for (uint32_t i : g->all[j]) if (isv(i)) acc += US(S(v[i])); o[j] = (S)acc; if (out_valid) setbit(out_valid, j, true); } });
      }
      return 0; }
    case AGG_MEAN: {
      *out_dt = (dt == F32) ? F32 : F64;
      parallel_ranges(G, 64, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
          Kahan k; int64_t c = 0;
          for (uint32_t i : g->all[j]) if (isv(i)) { k.add((double)v[i]); c++; }
          double m = c ? k.sum / (double)c : 0.0;
          if (dt == F32) ((float*)outv)[j] = (float)m; else ((double*)outv)[j] = m;
          if (out_valid) setbit(out_valid, j, c > 0);
        } });
      return 0; }
    case AGG_MIN: case AGG_MAX: {
      *out_dt = dt; T* o = (T*)outv;
      parallel_ranges(G, 64, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
          bool have = false; T acc = T(0);
          for (uint32_t i : g->all[j]) { if (!isv(i)) continue; if (!have) { acc = v[i]; have = true; } else acc = (op == AGG_MIN) ? min_ign<T>(acc, v[i]) : max_ign<T>(acc, v[i]); }
          o[j] = acc; if (out_valid) setbit(out_valid, j, have);
        } });
      return 0; }
  }
  return 1;
}

// ---------------------------------------------------------------------------
// Hash join on one key (u64 bit repr).  polars-ops/src/frame/join/hash_join/
// single_keys.rs:16-167 build_tables: pass 1 per-thread partition histogram
// (:52-66), prefix sums (:69-93), pass 2 scatter (key,idx) (:96-121), pass 3
// per-partition hash map key -> [idx...] in build order (:124-165).
// single_keys_inner.rs:11-38 probe_inner, :40-149 hash_join_tuples_inner:
// build on the SHORTER relation (hash_join/mod.rs:41-50 det_hash_prone_order),
// probe in probe-row order, one pair per build duplicate in insertion order.
// Left join: single_keys_left.rs:106-195 (probe = left, unmatched -> (idx, NULL)).
// Null keys never match (nulls_equal = false).
// ---------------------------------------------------------------------------
struct JoinTable {
  std::vector<int64_t> slot; std::vector<uint64_t> keys; std::vector<std::vector<uint32_t>> idx; uint64_t mask = 0;
  void init(size_t cap) { size_t c = 16; while (c < cap * 2) c <<= 1; slot.assign(c, -1); mask = c - 1; }
  static inline uint64_t mix(uint64_t k) { k *= 0x9e3779b97f4a7c15ull; return k ^ (k >> 29); }
  void insert(uint64_t k, uint32_t i) {
    uint64_t p = mix(k) & mask;
    for (;;) { int64_t g = slot[p]; if (g < 0) { slot[p] = (int64_t)keys.size(); keys.push_back(k); idx.emplace_back(1, i); return; }
      if (keys[g] == k) { idx[g].push_back(i); return; } p = (p + 1) & mask; }
  }
  const std::vector<uint32_t>* get(uint64_t k) const {
    uint64_t p = mix(k) & mask;
    for (;;) { int64_t g = slot[p]; if (g < 0) return nullptr; if (keys[g] == k) return &idx[g]; p = (p + 1) & mask; }
  }
};

struct Pairs { std::vector<uint32_t> left, right; std::vector<uint8_t> right_valid; bool has_right_valid = false; };

Pairs* join_impl(int how, const uint64_t* lk, const uint8_t* lv, int64_t nl, const uint64_t* rk, const uint8_t* rv, int64_t nr) {
  bool left_join = how == 1;
  // build side: shorter relation for inner; always right for left join
  bool swapped = !left_join && !(nl > nr);  // det_hash_prone_order: a = longer = probe
  const uint64_t* pk = left_join ? lk : (swapped ? rk : lk); const uint8_t* pv = left_join ? lv : (swapped ? rv : lv); int64_t np_ = left_join ? nl : (swapped ? nr : nl);
  const uint64_t* bk = left_join ? rk : (swapped ? lk : rk); const uint8_t* bv = left_join ? rv : (swapped ? lv : rv); int64_t nb = left_join ? nr : (swapped ? nl : nr);
  int P = (g_threads > 1 && nb >= 128 * g_threads) ? g_threads : 1;  // single_keys.rs:14 MIN_ELEMS_PER_THREAD
  // pass 1: histogram, pass 2: scatter, pass 3: per-partition table
  std::vector<std::vector<std::pair<uint64_t, uint32_t>>> parts(P);
  {
    std::vector<int64_t> cnt(P, 0);
    for (int64_t i = 0; i < nb; i++) { if (bv && !getbit(bv, i)) continue; cnt[hash_to_partition(bk[i] * RANDOM_ODD, P)]++; }
    for (int p = 0; p < P; p++) parts[p].reserve(cnt[p]);
    for (int64_t i = 0; i < nb; i++) { if (bv && !getbit(bv, i)) continue; parts[hash_to_partition(bk[i] * RANDOM_ODD, P)].push_back({bk[i], (uint32_t)i}); }
  }
  std::vector<JoinTable> tbls(P);
  parallel_tasks(P, [&](int p) { tbls[p].init(parts[p].size()); for (auto& kv : parts[p]) tbls[p].insert(kv.first, kv.second); });
  // probe in slices, flatten in order
  int S = std::max(1, g_threads);
  auto splits = split_offsets(np_, S);
  std::vector<std::vector<std::pair<uint32_t, int64_t>>> res(splits.size());
  parallel_tasks((int)splits.size(), [&](int s) {
    auto [off, len] = splits[s]; auto& out = res[s];
    for (int64_t i = off; i < off + len; i++) {
      const std::vector<uint32_t>* m = nullptr;
      if (!pv || getbit(pv, i)) m = tbls[hash_to_partition(pk[i] * RANDOM_ODD, P)].get(pk[i]);
      if (m) for (uint32_t b : *m) out.push_back({(uint32_t)i, (int64_t)b});
      else if (left_join) out.push_back({(uint32_t)i, -1});
    }
  });
  Pairs* pr = new Pairs(); pr->has_right_valid = left_join;
  int64_t w = 0;
  for (auto& r : res) {
    for (auto& t : r) {
      uint32_t a = t.first; int64_t b = t.second;
      if (left_join) {
        pr->left.push_back(a); pr->right.push_back(b < 0 ? 0u : (uint32_t)b);
        if ((w & 7) == 0) pr->right_valid.push_back(0);
        if (b >= 0) pr->right_valid[w >> 3] |= uint8_t(1u << (w & 7));
      } else if (swapped) { pr->left.push_back((uint32_t)b); pr->right.push_back(a); }
      else { pr->left.push_back(a); pr->right.push_back((uint32_t)b); }
      w++;
    }
  }
  return pr;
}

}  // namespace

// ===========================================================================
// C entry points (ctypes-friendly)
// ===========================================================================
extern "C" {

int orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; return 0; }
int orc_get_threads() { return g_threads; }
int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

int orc_cmp(int dt, int op, const void* lhs, const void* rhs, int rhs_scalar, int64_t n, uint8_t* out_bits) {
  memset(out_bits, 0, (size_t)((n + 7) / 8));
#define M(T) cmp_impl<T>(op, (const T*)lhs, (const T*)rhs, rhs_scalar != 0, n, out_bits);
  DISPATCH_NUMERIC(dt, M)
#undef M
  return 0;
}

// out must hold n elements of the output dtype (*out_dt). extra_valid: n bits or NULL.
int orc_arith(int dt, int op, const void* lhs, const void* rhs, int mode, int64_t n, void* out, uint8_t* extra_valid, int* has_extra, int* out_dt) {
  *out_dt = (op == TRUE_DIV && dt != F32 && dt != F64) ? F64 : dt;
  if (extra_valid) memset(extra_valid, 0, (size_t)((n + 7) / 8));
#define M(T) return arith_impl<T>(op, (const T*)lhs, (const T*)rhs, mode, n, out, extra_valid, has_extra);
  DISPATCH_NUMERIC(dt, M)
#undef M
  return 1;
}

// bitmap logic: polars-expr/src/expressions/binary.rs:110-118 (values op; validity AND elsewhere)
int orc_bitmap_binop(int op, const uint8_t* a, const uint8_t* b, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < (n + 7) / 8; i++) out[i] = op == 0 ? (a[i] & b[i]) : op == 1 ? (a[i] | b[i]) : (a[i] ^ b[i]);
  return 0;
}

// numeric cast, non-strict: unrepresentable -> null (ok_bits cleared)
int orc_cast(int from, int to, const void* in, int64_t n, void* out, uint8_t* ok_bits) {
  memset(ok_bits, 0xff, (size_t)((n + 7) / 8));
  auto loadd = [&](int64_t i, long double& v, bool& isnan_) {
    isnan_ = false;
    switch (from) {
      case I8: v = ((const int8_t*)in)[i]; break; case I16: v = ((const int16_t*)in)[i]; break;
      case I32: v = ((const int32_t*)in)[i]; break; case I64: v = ((const int64_t*)in)[i]; break;
      case U8: v = ((const uint8_t*)in)[i]; break; case U16: v = ((const uint16_t*)in)[i]; break;
      case U32: v = ((const uint32_t*)in)[i]; break; case U64: v = ((const uint64_t*)in)[i]; break;
      case F32: { float f = ((const float*)in)[i]; isnan_ = f != f; v = f; } break;
      case F64: { double f = ((const double*)in)[i]; isnan_ = f != f; v = f; } break;
    }
  };
  for (int64_t i = 0; i < n; i++) {
    long double v = 0; bool nan_; loadd(i, v, nan_);
    if (to == F64) { if (from == F32) ((double*)out)[i] = (double)((const float*)in)[i]; else if (from == F64) ((double*)out)[i] = ((const double*)in)[i]; else if (from == U64) ((double*)out)[i] = (double)((const uint64_t*)in)[i]; else if (from == I64) ((double*)out)[i] = (double)((const int64_t*)in)[i]; else ((double*)out)[i] = (double)v; continue; }
    if (to == F32) { if (from == F64) ((float*)out)[i] = (float)((const double*)in)[i]; else if (from == U64) ((float*)out)[i] = (float)((const uint64_t*)in)[i]; else if (from == I64) ((float*)out)[i] = (float)((const int64_t*)in)[i]; else ((float*)out)[i] = (float)v; continue; }
    long double t = (from == F32 || from == F64) ? truncl(v) : v;
    long double lo, hi;
    switch (to) {
      case I8: lo = -128; hi = 127; break; case I16: lo = -32768; hi = 32767; break;
      case I32: lo = -2147483648.0L; hi = 2147483647.0L; break; case I64: lo = -9223372036854775808.0L; hi = 9223372036854775807.0L; break;
      case U8: lo = 0; hi = 255; break; case U16: lo = 0; hi = 65535; break;
      case U32: lo = 0; hi = 4294967295.0L; break; case U64: lo = 0; hi = 18446744073709551615.0L; break;
      default: return 1;
    }
    bool ok = !nan_ && t >= lo && t <= hi && std::isfinite((double)v);
    if (!ok) { setbit(ok_bits, i, false); t = 0; }
    switch (to) {
      case I8: ((int8_t*)out)[i] = (int8_t)t; break; case I16: ((int16_t*)out)[i] = (int16_t)t; break;
      case I32: ((int32_t*)out)[i] = (int32_t)t; break; case I64: ((int64_t*)out)[i] = (int64_t)t; break;
      case U8: ((uint8_t*)out)[i] = (uint8_t)t; break; case U16: ((uint16_t*)out)[i] = (uint16_t)t; break;
      case U32: ((uint32_t*)out)[i] = (uint32_t)t; break; case U64: ((uint64_t*)out)[i] = (uint64_t)t; break;
    }
  }
  return 0;
}

// width in {1,2,4,8}; 0 = boolean (bit-packed values). mask_validity may be NULL.
// out buffers sized for n elements. Returns the number of kept rows in *out_n.
int orc_filter(int width, const void* values, const uint8_t* validity, const uint8_t* mask_bits, const uint8_t* mask_validity,
               int64_t n, void* out_values, uint8_t* out_validity, int64_t* out_n) {
  std::vector<uint8_t> m((size_t)((n + 7) / 8) + 8, 0);
  for (int64_t i = 0; i < (n + 7) / 8; i++) m[i] = mask_validity ? (mask_bits[i] & mask_validity[i]) : mask_bits[i];  // mod.rs:21-27
  if (n & 7) m[(n - 1) / 8] &= uint8_t((1u << (n & 7)) - 1);
  int64_t w = 0;
  switch (width) {
    case 0: w = filter_bits((const uint8_t*)values, m.data(), n, (uint8_t*)out_values); break;
    case 1: w = scalar_filter<uint8_t>((const uint8_t*)values, m.data(), n, (uint8_t*)out_values); break;
    case 2: w = scalar_filter<uint16_t>((const uint16_t*)values, m.data(), n, (uint16_t*)out_values); break;
    case 4: w = scalar_filter<uint32_t>((const uint32_t*)values, m.data(), n, (uint32_t*)out_values); break;
    case 8: w = scalar_filter<uint64_t>((const uint64_t*)values, m.data(), n, (uint64_t*)out_values); break;
    default: return 1;
  }
  if (validity && out_validity) filter_bits(validity, m.data(), n, out_validity);
  *out_n = w;
  return 0;
}

// gather/primitive.rs:9-78: out[i] = values[idx[i]]; null idx -> default + null;
// validity gathered bit by bit. out_validity always written (n_idx bits).
int orc_gather(int width, const void* values, const uint8_t* validity, const uint32_t* idx, const uint8_t* idx_validity,
               int64_t n_idx, void* out, uint8_t* out_validity) {
  for (int64_t i = 0; i < n_idx; i++) {
    bool ok = !idx_validity || getbit(idx_validity, i);
    uint32_t j = ok ? idx[i] : 0;
    if (width == 0) setbit((uint8_t*)out, i, ok ? getbit((const uint8_t*)values, j) : false);
    else if (ok) memcpy((char*)out + i * width, (const char*)values + (int64_t)j * width, (size_t)width);
    else memset((char*)out + i * width, 0, (size_t)width);
    setbit(out_validity, i, ok && (!validity || getbit(validity, j)));
  }
  return 0;
}

int orc_reduce(int dt, int op, const void* values, const uint8_t* validity, int64_t n, uint64_t* out_bits, int* out_dt, int* out_valid) {
  Scalar s; s.u = 0;
  if (dt == BOOL) {
    // BooleanChunked::sum (aggregate/mod.rs:253-267): count of set & valid bits as IdxSize
    const uint8_t* b = (const uint8_t*)values; int64_t c = 0, nv = 0;
    for (int64_t i = 0; i < n; i++) { bool ok = !validity || getbit(validity, i); nv += ok; c += ok && getbit(b, i); }
    *out_valid = 1;
    if (op == AGG_SUM) { *out_dt = U32; s.u = (uint32_t)c; }
    else if (op == AGG_COUNT) { *out_dt = U32; s.u = (uint32_t)nv; }
    else if (op == AGG_LEN) { *out_dt = U32; s.u = (uint32_t)n; }
    else if (op == AGG_MEAN) { *out_dt = F64; if (nv == 0) *out_valid = 0; else s.f64 = (double)c / (double)nv; }
    else return 1;
    *out_bits = s.u; return 0;
  }
#define M(T) { int rc = reduce_impl<T>(dt, op, (const T*)values, validity, n, &s, out_dt, out_valid); *out_bits = s.u; return rc; }
  DISPATCH_NUMERIC(dt, M)
#undef M
  return 1;
}

// keys: n_keys pointers to u64 bit-repr arrays (+ validity pointers, entries may be NULL)
void* orc_groupby_build(int n_keys, const uint64_t* const* keys, const uint8_t* const* valids, int64_t n, int maintain_order) {
  if (n_keys < 1 || n_keys > 4) return nullptr;
  return build_groups(n_keys, keys, valids, n, maintain_order);
}
int64_t orc_groups_count(void* g) { return (int64_t)((Groups*)g)->first.size(); }
int orc_groups_first(void* g, uint32_t* out) { auto* G = (Groups*)g; memcpy(out, G->first.data(), G->first.size() * 4); return 0; }
int orc_groups_free(void* g) { delete (Groups*)g; return 0; }
// out sized for n_groups of the output dtype; out_valid n_groups bits.
int orc_groups_agg(void* g, int dt, int op, const void* values, const uint8_t* validity, void* out, uint8_t* out_valid, int* out_dt) {
  const Groups* G = (const Groups*)g;
  if (op == AGG_LEN) return group_agg_impl<int64_t>(G, I64, op, nullptr, nullptr, out, out_valid, out_dt);
#define M(T) return group_agg_impl<T>(G, dt, op, (const T*)values, validity, out, out_valid, out_dt);
  DISPATCH_NUMERIC(dt, M)
#undef M
  return 1;
}

void* orc_join(int how, const uint64_t* lk, const uint8_t* lv, int64_t nl, const uint64_t* rk, const uint8_t* rv, int64_t nr) {
  return join_impl(how, lk, lv, nl, rk, rv, nr);
}
int64_t orc_pairs_count(void* p) { return (int64_t)((Pairs*)p)->left.size(); }
int orc_pairs_get(void* p, uint32_t* left, uint32_t* right, uint8_t* right_valid) {
  auto* P = (Pairs*)p;
  memcpy(left, P->left.data(), P->left.size() * 4); memcpy(right, P->right.data(), P->right.size() * 4);
  if (right_valid) { if (P->has_right_valid) memcpy(right_valid, P->right_valid.data(), P->right_valid.size()); else memset(right_valid, 0xff, (P->left.size() + 7) / 8); }
  return 0;
}
int orc_pairs_free(void* p) { delete (Pairs*)p; return 0; }

// HashPartitioner (polars-utils/src/hashing.rs:72-121): seed mixing + mulhi; nulls -> partition 0.
static inline uint64_t folded_multiply(uint64_t a, uint64_t b) { unsigned __int128 r = (unsigned __int128)a * b; return (uint64_t)r ^ (uint64_t)(r >> 64); }
uint64_t orc_partitioner_seed(uint64_t seed) {
  seed = folded_multiply(seed ^ 0x85921e81c41226a0ull, 0x3bc1d0faba166294ull);
  seed = folded_multiply(seed, 0xfbde893e21a73756ull);
  return seed | 1;
}
// part_out[i] = partition of row i, with hash = dirty_hash(key) = key * RANDOM_ODD
int orc_hash_partition(const uint64_t* keys, const uint8_t* valid, int64_t n, int n_parts, uint64_t seed, uint32_t* part_out) {
  uint64_t s = orc_partitioner_seed(seed);
  for (int64_t i = 0; i < n; i++) {
    if (valid && !getbit(valid, i)) { part_out[i] = 0; continue; }
    uint64_t h = keys[i] * RANDOM_ODD;
    part_out[i] = (uint32_t)hash_to_partition(h * s, (uint64_t)n_parts);
  }
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// TPC-H Q1 end to end in the reference's operator order, without any Python between the steps
// (this is what bench.py's cpu_baseline times):
//   FilterExec      predicate bitmap (comparisons/simd.rs), then every column filtered; the frame is
//                   split into one chunk per thread and the chunks are filtered in parallel
//                   (polars-mem-engine/src/executors/filter.rs:60-114), then re-assembled
//   expressions     1 - disc, price * (.), 1 + tax, (.) * (.)  -- one materialised column per
//                   BinaryExpr node (polars-expr/src/expressions/binary.rs:48-123)
//   GroupByExec     groups = (first, [idx...]) per distinct (flag, status) via partition-by-hash over
//                   threads (group_by/hashing.rs:116-167), then one aggregation per expression over
//                   the index lists (aggregations/mod.rs:854-1018), rayon over groups
// Outputs: up to `cap` groups, unsorted.  Returns the group count.
// ---------------------------------------------------------------------------------------------
template <class T>
static std::vector<T> filter_parallel(const T* v, const uint8_t* mask, int64_t n) {
  int nt = std::max(1, g_threads);
  int64_t per = ((n + nt - 1) / nt + 63) / 64 * 64;
  if (per == 0) per = 64;
  nt = (int)((n + per - 1) / per);
  std::vector<std::vector<T>> parts((size_t)std::max(nt, 1));
  parallel_tasks(nt, [&](int t) {
    int64_t b = per * t, e = std::min<int64_t>(n, b + per);
    parts[t].resize((size_t)(e - b));
    int64_t w = scalar_filter<T>(v + b, mask + b / 8, e - b, parts[t].data());
    parts[t].resize((size_t)w);
  });
  std::vector<int64_t> off((size_t)nt + 1, 0);
  for (int t = 0; t < nt; t++) off[t + 1] = off[t] + (int64_t)parts[t].size();
  std::vector<T> out((size_t)off[nt]);
  parallel_tasks(nt, [&](int t) { if (!parts[t].empty()) memcpy(out.data() + off[t], parts[t].data(), parts[t].size() * sizeof(T)); });
  return out;
}

extern "C" int64_t orc_q1(const int64_t* shipdate, const uint8_t* flag, const uint8_t* status, const int64_t* qty, const double* price,
                          const double* disc, const double* tax, int64_t n, int64_t cutoff, int cap, uint8_t* o_flag, uint8_t* o_status,
                          int64_t* o_sum_qty, double* o_sum_base, double* o_sum_disc_price, double* o_sum_charge, double* o_avg_qty,
                          double* o_avg_price, double* o_avg_disc, uint32_t* o_count) {
  // FilterExec
  std::vector<uint8_t> mask((size_t)((n + 63) / 64 * 8 + 8), 0);
  int64_t c = cutoff;
  cmp_impl<int64_t>(LE, shipdate, &c, true, n, mask.data());
  auto f_flag = filter_parallel<uint8_t>(flag, mask.data(), n);
  auto f_status = filter_parallel<uint8_t>(status, mask.data(), n);
  auto f_qty = filter_parallel<int64_t>(qty, mask.data(), n);
  auto f_price = filter_parallel<double>(price, mask.data(), n);
  auto f_disc = filter_parallel<double>(disc, mask.data(), n);
  auto f_tax = filter_parallel<double>(tax, mask.data(), n);
  const int64_t m = (int64_t)f_qty.size();
  // expression nodes, one materialised column each
  std::vector<double> one_minus((size_t)m), disc_price((size_t)m), one_plus((size_t)m), charge((size_t)m);
  int he = 0; double one = 1.0;
  arith_impl<double>(SUB, &one, f_disc.data(), 2, m, one_minus.data(), nullptr, &he);
  arith_impl<double>(MUL, f_price.data(), one_minus.data(), 0, m, disc_price.data(), nullptr, &he);
  arith_impl<double>(ADD, &one, f_tax.data(), 2, m, one_plus.data(), nullptr, &he);
  arith_impl<double>(MUL, disc_price.data(), one_plus.data(), 0, m, charge.data(), nullptr, &he);
  // GroupByExec on (flag, status)
  std::vector<uint64_t> k0((size_t)m), k1((size_t)m);
  parallel_ranges(m, 64, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; i++) { k0[i] = f_flag[i]; k1[i] = f_status[i]; } });
  const uint64_t* keys[2] = {k0.data(), k1.data()};
  const uint8_t* valids[2] = {nullptr, nullptr};
  std::unique_ptr<Groups> g(build_groups(2, keys, valids, m, 0));
  const int64_t G = (int64_t)g->first.size();
  if (G > cap) return -1;
  std::vector<uint8_t> ov((size_t)(G + 7) / 8 + 8);
  int odt = 0;
  for (int64_t j = 0; j < G; j++) { o_flag[j] = f_flag[g->first[j]]; o_status[j] = f_status[g->first[j]]; }
  group_agg_impl<int64_t>(g.get(), I64, AGG_SUM, f_qty.data(), nullptr, o_sum_qty, ov.data(), &odt);
  group_agg_impl<double>(g.get(), F64, AGG_SUM, f_price.data(), nullptr, o_sum_base, ov.data(), &odt);
  group_agg_impl<double>(g.get(), F64, AGG_SUM, disc_price.data(), nullptr, o_sum_disc_price, ov.data(), &odt);
  group_agg_impl<double>(g.get(), F64, AGG_SUM, charge.data(), nullptr, o_sum_charge, ov.data(), &odt);
  group_agg_impl<int64_t>(g.get(), I64, AGG_MEAN, f_qty.data(), nullptr, o_avg_qty, ov.data(), &odt);
  group_agg_impl<double>(g.get(), F64, AGG_MEAN, f_price.data(), nullptr, o_avg_price, ov.data(), &odt);
  group_agg_impl<double>(g.get(), F64, AGG_MEAN, f_disc.data(), nullptr, o_avg_disc, ov.data(), &odt);
  group_agg_impl<int64_t>(g.get(), I64, AGG_LEN, nullptr, nullptr, o_count, ov.data(), &odt);
  return G;
}

// ---------------------------------------------------------------------------------------------
// TPC-H Q1 the way the reference actually runs this shape: two plain-column keys, every aggregate
// pre-aggregatable and a sampled key cardinality <= 1000 => GroupByStreamingExec
// (polars-mem-engine/src/planner/lp.rs:19-50,660-693, executors/group_by_streaming.rs:117-257), i.e. the
// morsel-driven pipeline of polars-stream: each worker pulls morsels (POLARS_IDEAL_MORSEL_SIZE rows),
// evaluates the predicate and filters the morsel's columns, evaluates the expression nodes column-at-a-time
// on the (cache-resident) morsel, hashes the key pair into a thread-local fixed-size hot table
// (polars-expr/src/hot_groups/fixed_index_table.rs:19-165, 4096 slots) and updates the grouped reductions
// with plain `+=` (reduce/sum.rs:15-112, mean.rs:82-132 keeps (f64 sum, count), count.rs); the per-thread
// tables are combined at the end (nodes/group_by.rs:252-497).  Keys here are two u8 dictionary codes, so the
// hot table is indexed by (flag << 8 | status) and never evicts.
// ---------------------------------------------------------------------------------------------
extern "C" int64_t orc_q1_streaming(const int64_t* shipdate, const uint8_t* flag, const uint8_t* status, const int64_t* qty, const double* price,
                                    const double* disc, const double* tax, int64_t n, int64_t cutoff, int64_t morsel, int cap, uint8_t* o_flag,
                                    uint8_t* o_status, int64_t* o_sum_qty, double* o_sum_base, double* o_sum_disc_price, double* o_sum_charge,
                                    double* o_avg_qty, double* o_avg_price, double* o_avg_disc, uint32_t* o_count) {
  struct State { int64_t sum_qty = 0; double sum_base = 0, sum_dp = 0, sum_ch = 0, sum_qty_f = 0, sum_disc = 0; uint64_t cnt = 0; bool used = false; };
  const int nt = std::max(1, g_threads);
  if (morsel <= 0) morsel = 100000;
  // hot table: 65536-entry slot index (keys are two u8 codes) -> compact state vector
  std::vector<std::vector<int32_t>> slot_of((size_t)nt);
  std::vector<std::vector<State>> local((size_t)nt);
  std::atomic<int64_t> next{0};
  auto worker = [&](int tid) {
    std::vector<int32_t>& slot = slot_of[tid];
    slot.assign(65536, -1);
    std::vector<State>& tbl = local[tid];
    tbl.reserve(64);
    std::vector<uint8_t> mask((size_t)(morsel + 63) / 64 * 8 + 8), ff((size_t)morsel), fs((size_t)morsel);
    std::vector<int64_t> fq((size_t)morsel);
    std::vector<double> fp((size_t)morsel), fd((size_t)morsel), ft((size_t)morsel), t1((size_t)morsel), dp((size_t)morsel), t2((size_t)morsel), ch((size_t)morsel);
    for (;;) {
      const int64_t b = next.fetch_add(morsel);
      if (b >= n) break;
      const int64_t len = std::min<int64_t>(morsel, n - b);
      // filter node: predicate -> bitmap, then every column of the morsel is compacted
      memset(mask.data(), 0, mask.size());
      for (int64_t i = 0; i < len; i++) if (shipdate[b + i] <= cutoff) mask[i >> 3] |= uint8_t(1u << (i & 7));
      const int64_t m = scalar_filter<uint8_t>(flag + b, mask.data(), len, ff.data());
      scalar_filter<uint8_t>(status + b, mask.data(), len, fs.data());
      scalar_filter<int64_t>(qty + b, mask.data(), len, fq.data());
      scalar_filter<double>(price + b, mask.data(), len, fp.data());
      scalar_filter<double>(disc + b, mask.data(), len, fd.data());
      scalar_filter<double>(tax + b, mask.data(), len, ft.data());
      // select node: one materialised (morsel-sized) column per BinaryExpr
      for (int64_t i = 0; i < m; i++) t1[i] = 1.0 - fd[i];
      for (int64_t i = 0; i < m; i++) dp[i] = fp[i] * t1[i];
      for (int64_t i = 0; i < m; i++) t2[i] = 1.0 + ft[i];
      for (int64_t i = 0; i < m; i++) ch[i] = dp[i] * t2[i];
      // group-by node: hot-table slot per row, then one pass per reduction
      for (int64_t i = 0; i < m; i++) {
        int32_t& sl = slot[(size_t)ff[i] << 8 | fs[i]];
        if (sl < 0) { sl = (int32_t)tbl.size(); tbl.emplace_back(); }
        State& st = tbl[(size_t)sl];
        st.used = true;
        st.sum_qty += fq[i]; st.sum_base += fp[i]; st.sum_dp += dp[i]; st.sum_ch += ch[i]; st.sum_qty_f += (double)fq[i]; st.sum_disc += fd[i]; st.cnt++;
      }
    }
  };
  if (nt == 1) worker(0);
  else { std::vector<std::thread> ts; for (int t = 0; t < nt; t++) ts.emplace_back(worker, t); for (auto& t : ts) t.join(); }
  int64_t G = 0;
  for (int k = 0; k < 65536; k++) {
    State tot; bool used = false;
    for (int t = 0; t < nt; t++) { if (slot_of[t].empty() || slot_of[t][k] < 0) continue; const State& s = local[t][(size_t)slot_of[t][k]]; used = true;
      tot.sum_qty += s.sum_qty; tot.sum_base += s.sum_base; tot.sum_dp += s.sum_dp; tot.sum_ch += s.sum_ch; tot.sum_qty_f += s.sum_qty_f; tot.sum_disc += s.sum_disc; tot.cnt += s.cnt; }
    if (!used) continue;
    if (G >= cap) return -1;
    o_flag[G] = (uint8_t)(k >> 8); o_status[G] = (uint8_t)(k & 255);
    o_sum_qty[G] = tot.sum_qty; o_sum_base[G] = tot.sum_base; o_sum_disc_price[G] = tot.sum_dp; o_sum_charge[G] = tot.sum_ch;
    o_avg_qty[G] = tot.sum_qty_f / (double)tot.cnt; o_avg_price[G] = tot.sum_base / (double)tot.cnt; o_avg_disc[G] = tot.sum_disc / (double)tot.cnt;
    o_count[G] = (uint32_t)tot.cnt;
    G++;
  }
  return G;
}

// ---------------------------------------------------------------------------------------------
// Full-size checkers for bench.py / tests (BASELINE configs 2, 3, 5 at 1e9 rows): the same queries as
// pyoracle.q_filter_agg_cfg2 / q_groupby, executed the way the reference's streaming engine does -- morsels pulled by
// worker threads, thread-local reduction state combined at the end (polars-stream/src/nodes/group_by.rs:140-250 local
// phase, :252-497 combine_locals; reductions `+=` per group: polars-expr/src/reduce/sum.rs:15-112, mean.rs:82-132
// keeps (f64 sum, count), count.rs:6-128) -- so a 1e9-row input is checked in seconds.  They return PARTIAL STATES
// (sums and counts), which add across row blocks: the caller streams blocks of the host-twin generator through them.
// ---------------------------------------------------------------------------------------------

// filter(a > k).select((x*(1-y)).sum(), x.mean(), a.sum()): out = {sum_xy, sum_x} (f64), iout = {count_x_valid, sum_a (wrapping), rows_selected}
extern "C" int orc_cfg2_partial(const int64_t* a, const double* x, const double* y, const uint8_t* x_valid, int64_t n, int64_t k, int64_t morsel,
                                double* out, int64_t* iout) {
  if (morsel <= 0) morsel = 100000;
  const int nt = std::max(1, g_threads);
  struct St { double sxy = 0, sx = 0; int64_t cx = 0, rows = 0; uint64_t sa = 0; };
  std::vector<St> local((size_t)nt);
  std::atomic<int64_t> next{0};
  auto worker = [&](int tid) {
    St& s = local[(size_t)tid];
    std::vector<double> fx((size_t)morsel), fy((size_t)morsel), t1((size_t)morsel), pr((size_t)morsel);
    std::vector<uint8_t> fv((size_t)morsel);
    for (;;) {
      const int64_t b = next.fetch_add(morsel);
      if (b >= n) break;
      const int64_t len = std::min<int64_t>(morsel, n - b);
      int64_t m = 0;
      for (int64_t i = 0; i < len; i++) {            // filter node: mask, then every column compacted
        if (a[b + i] > k) { fx[m] = x[b + i]; fy[m] = y[b + i]; fv[m] = x_valid ? (uint8_t)getbit(x_valid, b + i) : 1; s.sa += (uint64_t)a[b + i]; m++; }
      }
      for (int64_t i = 0; i < m; i++) t1[i] = 1.0 - fy[i];        // one column per BinaryExpr node
      for (int64_t i = 0; i < m; i++) pr[i] = fx[i] * t1[i];
      for (int64_t i = 0; i < m; i++) if (fv[i]) { s.sxy += pr[i]; s.sx += fx[i]; s.cx++; }
      s.rows += m;
    }
  };
  if (nt == 1) worker(0);
  else { std::vector<std::thread> ts; for (int t = 0; t < nt; t++) ts.emplace_back(worker, t); for (auto& t : ts) t.join(); }
  St tot;
  for (auto& s : local) { tot.sxy += s.sxy; tot.sx += s.sx; tot.cx += s.cx; tot.rows += s.rows; tot.sa += s.sa; }
  out[0] = tot.sxy; out[1] = tot.sx;
  iout[0] = tot.cx; iout[1] = (int64_t)tot.sa; iout[2] = tot.rows;
  return 0;
}

// group_by(key).agg(v.sum(), v.count()) partial states for keys known to lie in [0, n_slots): thread-local tables indexed by
// the key (the role of the per-thread Grouper + VecGroupedReduction), combined at the end.  key_dt: I64 | U32; val_dt: I64 | F64.
// sums: int64 (wrapping) or double per slot, counts: rows per slot (all values valid); both are ADDED to (the caller zeroes them).
// Returns 0, or 2 if a key falls outside [0, n_slots).
extern "C" int orc_groupby_dense_partial(int key_dt, const void* keys, int val_dt, const void* vals, int64_t n, int64_t n_slots, int64_t morsel,
                                         void* sums, int64_t* counts) {
  if (morsel <= 0) morsel = 100000;
  if ((key_dt != I64 && key_dt != U32) || (val_dt != I64 && val_dt != F64)) return 1;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(g_threads, (n + morsel - 1) / morsel));
  std::vector<std::vector<int64_t>> lcnt((size_t)nt);
  std::vector<std::vector<uint64_t>> lsum((size_t)nt);   // int64 sums as wrapping u64, f64 sums as bit patterns of doubles
  std::atomic<int64_t> next{0};
  std::atomic<int> bad{0};
  auto worker = [&](int tid) {
    lcnt[(size_t)tid].assign((size_t)n_slots, 0);
    lsum[(size_t)tid].assign((size_t)n_slots, 0);
    int64_t* c = lcnt[(size_t)tid].data();
    uint64_t* su = lsum[(size_t)tid].data();
    double* sd = reinterpret_cast<double*>(su);
    for (;;) {
      const int64_t b = next.fetch_add(morsel);
      if (b >= n) break;
      const int64_t e = std::min<int64_t>(n, b + morsel);
      for (int64_t i = b; i < e; i++) {
        const int64_t key = key_dt == I64 ? ((const int64_t*)keys)[i] : (int64_t)((const uint32_t*)keys)[i];
        if (key < 0 || key >= n_slots) { bad.store(1); continue; }
        c[key]++;
        if (val_dt == I64) su[key] += (uint64_t)((const int64_t*)vals)[i];
        else sd[key] += ((const double*)vals)[i];
      }
    }
  };
  if (nt == 1) worker(0);
  else { std::vector<std::thread> ts; for (int t = 0; t < nt; t++) ts.emplace_back(worker, t); for (auto& t : ts) t.join(); }
  if (bad.load()) return 2;
  parallel_ranges(n_slots, 64, [&](int64_t b, int64_t e, int) {
    for (int64_t s = b; s < e; s++) {
      for (int t = 0; t < nt; t++) {
        counts[s] += lcnt[(size_t)t][(size_t)s];
        if (val_dt == I64) ((uint64_t*)sums)[s] += lsum[(size_t)t][(size_t)s];
        else ((double*)sums)[s] += reinterpret_cast<const double*>(lsum[(size_t)t].data())[s];
      }
    }
  });
  return 0;
}
