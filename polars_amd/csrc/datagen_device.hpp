// datagen_device.hpp -- counter-based synthetic TPC-H lineitem rows (benchmark / test support, not part of the hot path).
// Row i is a pure function of (seed, i): the device kernel and the host reference evaluate the same code, so the table is
// identical for any launch geometry and the CPU tests can pin the kernel's arithmetic without a GPU.
// Distributions follow polars_amd/datagen.py (_line_columns_host): TPC-H-like value ranges, Q1's filter keeps ~98 % of the rows
// and the (returnflag, linestatus) pairs are (A,F), (N,F), (N,O), (R,F).
#pragma once
#include <stdint.h>

#ifndef PLX_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define PLX_HD __host__ __device__
#else
#define PLX_HD
#endif
#endif

namespace plx {
namespace datagen {

constexpr int64_t kDayUs = 86400000000ll;
constexpr int64_t kStart = 8035ll * kDayUs;      // 1992-01-01 as Datetime[us]
constexpr int64_t kCurrent = 9298ll * kDayUs;    // 1995-06-17

struct LineitemRow {
  int64_t shipdate, quantity;
  uint8_t returnflag, linestatus;
  double extendedprice, discount, tax;
};

PLX_HD inline uint64_t mix64(uint64_t z) {       // splitmix64 output function
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
PLX_HD inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }   // v_mul_hi_u32 chain on the device
// uniform integer in [lo, hi) from stream s of row i
PLX_HD inline int64_t rand_range(uint64_t key, uint64_t i, uint32_t s, int64_t lo, int64_t hi) {
  return lo + (int64_t)mulhi64(mix64(key + i * 8 + s), (uint64_t)(hi - lo));
}
PLX_HD inline LineitemRow lineitem_row(uint64_t seed, uint64_t i) {
  const uint64_t key = mix64(seed);
  LineitemRow r;
  r.shipdate = kStart + rand_range(key, i, 0, 1, 2526 + 121) * kDayUs;
  r.quantity = rand_range(key, i, 1, 1, 51);
  r.extendedprice = (double)(r.quantity * rand_range(key, i, 2, 90000, 210000)) / 100.0;   // two decimals, correctly rounded
  r.discount = (double)rand_range(key, i, 3, 0, 11) / 100.0;
  r.tax = (double)rand_range(key, i, 4, 0, 9) / 100.0;
  const int64_t receipt = r.shipdate + rand_range(key, i, 5, 1, 31) * kDayUs;
  const bool coin = (mix64(key + i * 8 + 6) >> 63) != 0;
  r.returnflag = receipt <= kCurrent ? (coin ? 0 : 2) : 1;     // A / R before the current date, N after
  r.linestatus = r.shipdate > kCurrent ? 1 : 0;                // F / O
  return r;
}

// ---- orders + their lineitem rows (TPC-H Q3 columns), dbgen row order: both tables ascending in orderkey ------------
// Sparse order keys (8 of every 32 used), 1-7 lines per order, l_shipdate = o_orderdate + U[1, 121] days
// (polars_amd/datagen.py orders_lineitem_host / _device restated as a pure function of (seed, order, line)).
constexpr int64_t kOrderDays = 2406;             // days in [1992-01-01, 1998-08-02]
struct OrderRow {
  int64_t orderkey, custkey, orderdate;
  uint32_t n_lines;
};
struct Q3LineRow {
  int64_t shipdate;
  double extendedprice, discount;
};
PLX_HD inline OrderRow order_row(uint64_t seed, uint64_t i, int64_t cust_hi) {
  const uint64_t key = mix64(seed ^ 0x6f72646572730000ull);
  OrderRow r;
  r.orderkey = (int64_t)((i / 8) * 32 + (i % 8) + 1);
  r.orderdate = kStart + rand_range(key, i, 0, 0, kOrderDays) * kDayUs;
  r.custkey = rand_range(key, i, 1, 1, cust_hi);
  r.n_lines = (uint32_t)rand_range(key, i, 2, 1, 8);
  return r;
}
PLX_HD inline Q3LineRow q3_line_row(uint64_t seed, uint64_t order, uint32_t line, int64_t orderdate) {
  const uint64_t key = mix64(mix64(seed ^ 0x6c696e6573000000ull) + order);   // one stream family per order
  Q3LineRow r;
  r.shipdate = orderdate + rand_range(key, line, 0, 1, 122) * kDayUs;
  const int64_t qty = rand_range(key, line, 1, 1, 51);
  r.extendedprice = (double)(qty * rand_range(key, line, 2, 90000, 210000)) / 100.0;
  r.discount = (double)rand_range(key, line, 3, 0, 11) / 100.0;
  return r;
}

// ---- customer (TPC-H Q3 columns): dense keys 1..n in dbgen order, one of five market segments per customer ------------
constexpr int kSegments = 5;   // AUTOMOBILE, BUILDING, FURNITURE, HOUSEHOLD, MACHINERY (dictionary codes 0..4)
PLX_HD inline uint8_t customer_segment(uint64_t seed, uint64_t custkey) {
  return (uint8_t)rand_range(mix64(seed ^ 0x637573746f6d6572ull), custkey, 0, 0, kSegments);
}

// ---- one uniform column: lo + floor(U * (hi - lo)), optionally scaled to a double (BASELINE configs 2 / 3 / 5) -------
PLX_HD inline int64_t uniform_value(uint64_t seed, uint32_t stream, uint64_t i, int64_t lo, int64_t hi) {
  return rand_range(mix64(seed), i, stream & 7u, lo, hi);
}

// ---- one heavy-tailed key column (BASELINE config 3's "Zipf s = 1.1" variant): key = floor(1 / x^10) - 1 with x uniform in
// [x0, 1), x0 = n_keys^(-1/10) -- a Pareto variable of tail index 0.1, whose density falls like k^(-1.1): key 0 holds ~9 % of the rows, key 1 ~5 %,
// the tail thins out over [0, n_keys).  Fixed point (62 fractional bits, integer multiplies and ONE integer division): the device kernel and the
// host twin agree bit for bit, which floating-point pow() would not guarantee.  x0_q62 = round(x0 * 2^62), computed once by the caller.
PLX_HD inline uint64_t mul_q62(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 62); }
PLX_HD inline int64_t zipf_value(uint64_t seed, uint32_t stream, uint64_t i, uint64_t x0_q62, int64_t n_keys) {
  const uint64_t one = 1ull << 62;
  const uint64_t x = x0_q62 + mulhi64(mix64(mix64(seed) + i * 8 + (stream & 7u)), one - x0_q62);
  const uint64_t x2 = mul_q62(x, x), x4 = mul_q62(x2, x2), x8 = mul_q62(x4, x4), x10 = mul_q62(x8, x2);
  const uint64_t k = x10 ? one / x10 : (uint64_t)n_keys;     // floor(1 / x^10)
  const int64_t id = (int64_t)(k ? k - 1 : 0);
  return id < n_keys ? id : n_keys - 1;
}

// ---- Utf8View of the string "id%010d" % v (12 bytes: always inline; view layout polars-arrow/src/array/binview/view.rs:20-29) --------
// w0 = length (12) | bytes 0..3 << 32, w1 = bytes 4..11, little-endian
PLX_HD inline void id_view(uint64_t v, uint64_t* w0, uint64_t* w1) {
  unsigned char s[12];
  s[0] = 'i'; s[1] = 'd';
  for (int d = 11; d >= 2; d--) { s[d] = (unsigned char)('0' + v % 10); v /= 10; }
  uint64_t a = 12, b = 0;
  for (int i = 0; i < 4; i++) a |= (uint64_t)s[i] << (32 + 8 * i);
  for (int i = 0; i < 8; i++) b |= (uint64_t)s[4 + i] << (8 * i);
  *w0 = a; *w1 = b;
}

// ---- Utf8View of the 20-byte string "id%010d-longkey" % v: NOT inline -- {length 20, prefix "id00".., buffer 0, offset} into a pool that holds every distinct string
// once at (v - lo) * 20 (views of several rows may share bytes: what a gather of a string column leaves behind) ----------------------------------------------------
constexpr int kLongIdLen = 20;
PLX_HD inline void long_id_bytes(uint64_t v, unsigned char* s) {
  s[0] = 'i'; s[1] = 'd';
  for (int d = 11; d >= 2; d--) { s[d] = (unsigned char)('0' + v % 10); v /= 10; }
  const char* tail = "-longkey";
  for (int i = 0; i < 8; i++) s[12 + i] = (unsigned char)tail[i];
}
PLX_HD inline void long_id_view(uint64_t v, uint64_t lo, uint64_t* w0, uint64_t* w1) {
  unsigned char s[kLongIdLen];
  long_id_bytes(v, s);
  uint64_t a = (uint64_t)kLongIdLen;
  for (int i = 0; i < 4; i++) a |= (uint64_t)s[i] << (32 + 8 * i);
  *w0 = a; *w1 = ((v - lo) * (uint64_t)kLongIdLen) << 32;      // buffer index 0 | offset << 32
}

}  // namespace datagen
}  // namespace plx
