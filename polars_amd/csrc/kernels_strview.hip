// kernels_strview.hip -- raw Utf8View / BinaryView keys: device-side dictionary encoding.
//
// The reference groups / joins on string keys by hashing the 16-byte views (crates/polars-expr/src/hash_keys.rs:413-452
// BinviewKeys; crates/polars-compute/src/binview_index_map.rs: an index map view -> dense index that compares inline views
// by value and long strings by bytes; view layout crates/polars-arrow/src/array/binview/view.rs:20-29,55: {len u32, then 12 inline
// bytes, or prefix u32 + buffer index u32 + offset u32}).  Here the same index map is built ONCE, on the device, when a string
// column enters: every row's view is looked up / inserted in an open-addressing table of 32-byte slots in HBM (63-bit tag CAS:
// EMPTY -> tag|BUSY -> tag, the claimer publishes the view and its code before the tag, as the wide-key aggregation sink does) and the row gets the
// u32 code of its string; from then on the column is a dictionary column (codes in [0, n_distinct): dense ids, so group-bys on it
// plan direct-address tables).  Strings of <= 12 bytes never touch the data buffers: the view IS the string (Arrow pads inline
// views with zeros), which is the whole of BASELINE config 5's "id%010d" keys.
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
constexpr unsigned long long kEmptyTag = ~0ull, kBusy = 1ull << 63;

// one slot = 32 bytes, so a probe touches one 128-B line: tag, the view words the tag stands for, and the code
struct alignas(32) StrSlot {
  unsigned long long tag;      // kEmptyTag | tag|BUSY (claimed, being published) | tag (published: the slot never changes again)
  unsigned long long w0;       // len | prefix << 32 (bytes 0..7 of the view)
  unsigned long long w1;       // inline: bytes 8..15; long: absolute byte offset of the first-seen string in `data`
  unsigned int code, pad;
};
struct StrTable {
  StrSlot* slots;               // [cap]
  unsigned int* counter;        // [0] next code, [1] overflow flag
  uint32_t log2_cap, max_probe;
};
struct StrEncode {
  const unsigned long long* views;    // [n][2]
  const uint64_t* validity;           // may be null
  const unsigned char* data;          // all data buffers, concatenated
  const unsigned long long* buf_base; // [n_buffers] offset of each original data buffer inside `data`
  int64_t n;
  unsigned int* out_codes;            // [n] (null for a counting-only sample pass)
  unsigned long long* dict_views;     // [max_codes][2] view of every code's string (long strings rebased: absolute offset in w1's high half)
  uint32_t max_codes;
  // a view column that travels without a bitmap (validity == null) marks its nulls by STAMPED views (length kStrviewNullLen: plx_strview_stamp_nulls): stamps = 1 makes
  // such rows null keys; stamp_valid (may be null) receives the validity bitmap read off the stamps, stamp_nulls[0] their number -- no pass of its own over the views
  uint32_t stamps;
  unsigned long long* stamp_valid;
  unsigned long long* stamp_nulls;
};

__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int ld32(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st32(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t mix(uint64_t h, uint64_t w) { h ^= w; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; return h; }
// up to 8 bytes at p (any alignment) as a little-endian word.  One 8-byte, or 4 + 2 + 1-byte loads: global loads need no alignment on gfx950 (the code object runs in
// unaligned access mode; the copies below compile to single global_load instructions) -- byte by byte, a 20-byte key cost 20 loads to hash (1e9 rows: 47 ms, now measured again)
__device__ __forceinline__ uint64_t load_bytes(const unsigned char* p, uint32_t n) {
  uint64_t w = 0;
  if (n >= 8) { __builtin_memcpy(&w, p, 8); return w; }
  uint32_t i = 0;
  if (n & 4) { uint32_t x; __builtin_memcpy(&x, p, 4); w = x; i = 4; }
  if (n & 2) { unsigned short x; __builtin_memcpy(&x, p + i, 2); w |= (uint64_t)x << (8 * i); i += 2; }
  if (n & 1) w |= (uint64_t)p[i] << (8 * i);
  return w;
}
__device__ __forceinline__ uint64_t hash_long(const unsigned char* s, uint32_t len) {
  uint64_t h = 0x9e3779b97f4a7c15ull ^ len;
  uint32_t i = 0;
  for (; i + 8 <= len; i += 8) h = mix(h, load_bytes(s + i, 8));
  if (i < len) h = mix(h, load_bytes(s + i, len - i));
  return h;
}
__device__ __forceinline__ bool same_bytes(const unsigned char* a, const unsigned char* b, uint32_t len) {
  uint32_t i = 0;
  for (; i + 8 <= len; i += 8) if (load_bytes(a + i, 8) != load_bytes(b + i, 8)) return false;
  return i >= len || load_bytes(a + i, len - i) == load_bytes(b + i, len - i);
}

// Rows per thread in flight: the table probe is a dependent random access (one 128-B line per row out of a table that does not
// fit L2), so the kernel is bound by how many probes are outstanding, not by bytes.
constexpr int kStrRows = 4;

struct StrKey { uint64_t w0, w1, tag, slot; const unsigned char* bytes; uint32_t len; bool is_long, valid; };

__device__ __forceinline__ StrKey str_key(const StrEncode& e, const StrTable& t, int64_t row, ulonglong2 v) {
  StrKey k;
  k.valid = e.validity ? ((e.validity[row >> 6] >> (row & 63)) & 1) : (!e.stamps || (uint32_t)v.x != kStrviewNullLen);
  k.len = (uint32_t)v.x;
  k.is_long = k.len > 12;
  k.w0 = v.x; k.w1 = v.y; k.bytes = nullptr;
  uint64_t h;
  if (k.is_long) {
    const uint32_t buf = (uint32_t)v.y, off = (uint32_t)(v.y >> 32);
    const uint64_t abs_off = e.buf_base[buf] + off;
    k.bytes = e.data + abs_off;
    k.w1 = abs_off;
    h = k.valid ? hash_long(k.bytes, k.len) : 0;
  } else h = mix(mix(0x9e3779b97f4a7c15ull, k.w0), k.w1);
  h *= 0x55fbfd6bfc5458e9ull;
  k.tag = h & ~kBusy;
  if (k.tag == (kEmptyTag & ~kBusy)) k.tag ^= 1;
  k.slot = h >> (64 - t.log2_cap);
  return k;
}

__device__ __forceinline__ bool same_string(const StrEncode& e, const StrKey& k, unsigned long long s0, unsigned long long s1) {
  if (s0 != k.w0) return false;
  return k.is_long ? (s1 == k.w1 || same_bytes(e.data + s1, k.bytes, k.len)) : (s1 == k.w1);
}

// Two ways to read a slot.  The PLAIN read goes through this CU's L1 and this XCD's L2 and may be stale, but a published slot
// never changes, so a plain read that shows a published tag of ANOTHER string is final ("go on"), and one that shows this
// string's tag with matching words and a written code is a hit.  Anything else (EMPTY, BUSY, or halves that disagree) is
// repeated coherently (device scope), and that is where claiming happens.  Once the dictionary is warm (n_distinct
// << n, the case worth encoding) almost every row ends on plain reads.
__global__ __launch_bounds__(kBlock) void strview_encode_kernel(StrEncode e, StrTable t) {
  const uint64_t mask = (1ull << t.log2_cap) - 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < e.n; base += stride * kStrRows) {
    StrKey key[kStrRows];
    int64_t code[kStrRows];
    uint32_t probes[kStrRows];
    ulonglong2 a[kStrRows], b[kStrRows];
#pragma unroll
    for (int r = 0; r < kStrRows; r++) {
      const int64_t row = base + (int64_t)r * stride;
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      ulonglong2 v = make_ulonglong2(0, 0);
      if (row < e.n) { const u64x2 t2 = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(e.views) + row); v.x = t2.x; v.y = t2.y; }
      key[r] = str_key(e, t, row < e.n ? row : 0, v);
      if (row >= e.n) key[r].valid = false;
      code[r] = key[r].valid ? -1 : 0;
      probes[r] = 0;
      if (e.stamp_valid) {        // (wave-uniform) the 64 rows of this wave and r are one word of the bitmap: lane 0's row is a multiple of 64
        const uint64_t m = ballot(key[r].valid);
        const int64_t row_l0 = __shfl(row, 0, 64);
        if (lane_id() == 0 && row_l0 < e.n) {
          e.stamp_valid[row_l0 >> 6] = m;
          const int64_t rows = e.n - row_l0 < 64 ? e.n - row_l0 : 64;
          if (rows != popc64(m)) atomicAdd(e.stamp_nulls, (unsigned long long)(rows - popc64(m)));
        }
      }
    }
    // plain probes, kStrRows independent lines in flight
    bool any = true;
    while (any) {
#pragma unroll
      for (int r = 0; r < kStrRows; r++) {
        if (code[r] != -1) continue;          // resolved rows fetch nothing: the loop runs for the slowest row of the wave
        const ulonglong2* s = reinterpret_cast<const ulonglong2*>(t.slots + key[r].slot);
        a[r] = s[0]; b[r] = s[1];
      }
      any = false;
#pragma unroll
      for (int r = 0; r < kStrRows; r++) {
        if (code[r] != -1) continue;
        const unsigned long long tg = a[r].x;
        if (tg == kEmptyTag || (tg & kBusy)) { code[r] = -2; continue; }     // needs the coherent path, at this slot
        if (tg == key[r].tag) {
          // the slot's two 16-byte halves are two loads, and under cache thrash they can come from different moments (the half
          // with w1 / code OLDER than the half that shows the published tag): a hit needs every word to agree and a written
          // code; a disagreement is not "another string" (63-bit tags do not collide in practice) but a reason to look again
          const unsigned int c = (unsigned int)b[r].y;
          if (c != 0xffffffffu && same_string(e, key[r], a[r].y, b[r].x)) code[r] = (int64_t)c; else code[r] = -2;
          continue;
        }
        key[r].slot = (key[r].slot + 1) & mask;
        if (++probes[r] >= t.max_probe) { code[r] = -3; continue; }
        any = true;
      }
    }
    // coherent path: claim / wait / compare
#pragma unroll
    for (int r = 0; r < kStrRows; r++) {
      const StrKey k = key[r];
      uint64_t slot = k.slot;
      int64_t c = code[r];
      uint32_t probe = probes[r];
      bool failed = c == -3;
      while (c == -2 && !failed) {
        if (probe++ >= t.max_probe) { failed = true; break; }
        StrSlot* s = t.slots + slot;
        unsigned long long cur = ld(&s->tag);
        bool claimed = false;
        if (cur == kEmptyTag) {
          const unsigned long long old = atomicCAS(&s->tag, kEmptyTag, k.tag | kBusy);
          if (old == kEmptyTag) claimed = true; else cur = old;
        }
        if (claimed) {   // publish: view words + code (write-through) -> drain -> tag
          const unsigned int nc = atomicAdd(t.counter, 1u);
          st(&s->w0, k.w0); st(&s->w1, k.w1); st32(&s->code, nc);
          if (e.dict_views && nc < e.max_codes) { e.dict_views[(size_t)nc * 2] = k.w0; e.dict_views[(size_t)nc * 2 + 1] = k.w1; }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          st(&s->tag, k.tag);
          c = (int64_t)nc;
        }
        // every lane of the wave is past its publish before any lane starts to wait (a lane must never wait for a slot that a
        // lane of its own wave has claimed but not yet published)
        __builtin_amdgcn_wave_barrier();
        if (!claimed && (cur & ~kBusy) == k.tag) {
          while (cur & kBusy) { __builtin_amdgcn_s_sleep(1); cur = ld(&s->tag); }
          asm volatile("" ::: "memory");
          if (same_string(e, k, ld(&s->w0), ld(&s->w1))) c = (int64_t)ld32(&s->code);
        }
        slot = (slot + 1) & mask;
      }
      if (failed) { atomicExch(t.counter + 1, 1u); c = 0; }
      code[r] = c;
    }
    if (e.out_codes) {
#pragma unroll
      for (int r = 0; r < kStrRows; r++) {
        const int64_t row = base + (int64_t)r * stride;
        if (row < e.n) __builtin_nontemporal_store((unsigned int)code[r], e.out_codes + row);
      }
    }
  }
}

// dictionary materialisation: byte length of every code's string, then the bytes, contiguously in code order
__global__ __launch_bounds__(kBlock) void strdict_len_kernel(const unsigned long long* __restrict__ dict_views, int64_t n, uint32_t* __restrict__ lens) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) lens[i] = (uint32_t)dict_views[(size_t)i * 2];
}
__global__ __launch_bounds__(kBlock) void strdict_copy_kernel(const unsigned long long* __restrict__ dict_views, const unsigned char* __restrict__ data, int64_t n,
                                                              const unsigned long long* __restrict__ offsets, unsigned char* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long w0 = dict_views[(size_t)i * 2], w1 = dict_views[(size_t)i * 2 + 1];
    const uint32_t len = (uint32_t)w0;
    unsigned char* dst = out + offsets[i];
    if (len <= 12) {
      for (uint32_t b = 0; b < len; b++) dst[b] = (unsigned char)(b < 4 ? (w0 >> (32 + 8 * b)) : (w1 >> (8 * (b - 4))));
    } else {
      const unsigned char* src = data + w1;
      for (uint32_t b = 0; b < len; b++) dst[b] = src[b];
    }
  }
}
}  // namespace

// views / validity / data: device buffers.  Returns the codes (device, u32 per row), the number of distinct strings and the
// dictionary's views (device, [n_distinct][2], long strings carrying their absolute data offset).
void strview_dict_encode(const uint64_t* views, const uint64_t* validity, const uint8_t* data, const uint64_t* buf_base, int64_t n, Buf* out_codes, Buf* out_dict_views,
                         int64_t* n_distinct, Buf* stamps_valid, int64_t* stamps_nulls) {
  *out_codes = dev_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(n, 1));
  if (stamps_nulls) *stamps_nulls = 0;
  if (n == 0) { *out_dict_views = dev_alloc(16); *n_distinct = 0; return; }
  const bool stamps = stamps_valid && !validity;
  if (stamps) *stamps_valid = dev_alloc(bitmap_bytes(n));
  auto run = [&](int64_t rows, uint32_t log2_cap, uint32_t max_codes, unsigned int* codes, Buf* dict, uint32_t* distinct) -> bool {
    const uint64_t cap = 1ull << log2_cap;
    Buf slots = dev_alloc(sizeof(StrSlot) * cap), ctr = dev_alloc_zero(16);      // [0] codes handed out, [1] overflow, [2..3] nulls read off the stamps (u64)
    PLX_HIP(hipMemsetAsync(slots->ptr, 0xff, sizeof(StrSlot) * cap, stream()));
    if (dict) *dict = dev_alloc(16 * (size_t)std::max<uint32_t>(max_codes, 1));
    StrTable t{slots->as<StrSlot>(), ctr->as<unsigned int>(), log2_cap,
               (uint32_t)std::min<uint64_t>(cap, 1u << 12)};
    StrEncode e{(const unsigned long long*)views, validity, data, (const unsigned long long*)buf_base, rows, codes, dict ? (*dict)->as<unsigned long long>() : nullptr, max_codes,
                stamps ? 1u : 0u, stamps && codes ? (*stamps_valid)->as<unsigned long long>() : nullptr, ctr->as<unsigned long long>() + 1};
    {
      ProfileScope ps("strview_dict_encode", (uint64_t)rows * (16 + (codes ? 4 : 0)), (uint64_t)rows);
      hipLaunchKernelGGL(strview_encode_kernel, dim3(grid_for(rows, kBlock * kStrRows, 8)), dim3(kBlock), 0, stream(), e, t);
      PLX_HIP(hipGetLastError());
    }
    uint32_t res[4] = {0, 0, 0, 0};
    d2h_sync(res, ctr->ptr, 16);
    *distinct = res[0];
    if (stamps && codes && stamps_nulls) *stamps_nulls = (int64_t)((uint64_t)res[2] | ((uint64_t)res[3] << 32));
    return res[1] == 0;
  };
  // table size from a sample of the first rows: d distinct in S rows -> G ~ solution of d = G (1 - exp(-S / G))
  uint32_t log2_cap = 16;
  {
    const int64_t S = std::min<int64_t>(n, (int64_t)1 << 20);
    uint32_t d = 0;
    if (run(S, 22, 0, nullptr, nullptr, &d)) {
      double G = (double)d;
      if (S < n && d > 0) {
        if ((double)d >= 0.999 * (double)S) G = (double)n;
        else { double lo = d, hi = 1e15; for (int it = 0; it < 200; it++) { const double mid = std::sqrt(lo * hi); (mid * (1.0 - std::exp(-(double)S / mid)) < (double)d ? lo : hi) = mid; } G = std::min(hi, (double)n); }
      }
      static const double slack = [] { const char* e = getenv("PLX_STRVIEW_SLACK"); const double v = e ? atof(e) : 0.0; return (v >= 1.2 && v <= 8.0) ? v : 2.5; }();   // slots per expected string
      while (log2_cap < 34 && (double)(1ull << log2_cap) < slack * G) log2_cap++;
    } else log2_cap = 24;
  }
  for (int attempt = 0; attempt < 8; attempt++) {
    uint32_t d = 0;
    const uint32_t max_codes = (uint32_t)std::min<uint64_t>((uint64_t)n, 1ull << log2_cap);
    if (run(n, log2_cap, max_codes, (*out_codes)->as<unsigned int>(), out_dict_views, &d)) { *n_distinct = d; return; }
    log2_cap += 2;
    PLX_REQUIRE(log2_cap <= 34, PLX_ERR_OOM, "string dictionary table would exceed 2^34 slots");
  }
  fail(PLX_ERR_OOM, "string dictionary table kept overflowing");
}

namespace {
template <class OFF>
__global__ __launch_bounds__(kBlock) void strviews_from_offsets_kernel(const OFF* __restrict__ offs, const unsigned char* __restrict__ data, unsigned long long data_base, long long data_len,
                                                                       int64_t n, unsigned long long* __restrict__ views, const uint64_t* __restrict__ validity, int64_t row0, bool stamp_nulls, unsigned int* __restrict__ err) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long a = (long long)offs[i], e = (long long)offs[i + 1];
    const bool ok = !validity || ((validity[(row0 + i) >> 6] >> ((row0 + i) & 63)) & 1);
    unsigned long long w0 = !ok && stamp_nulls ? (unsigned long long)kStrviewNullLen : 0ull, w1 = 0;      // views handed out as they are carry their nulls as stamps
    if (a < 0 || e < a || e > data_len || e - a > 0x7fffffffll) { *err = 1u; }
    else if (ok) {
      const uint32_t len = (uint32_t)(e - a);
      const unsigned char* p = data + data_base + (unsigned long long)a;
      if (len <= 12) {
        w0 = (unsigned long long)len | (load_bytes(p, len < 4 ? len : 4) << 32);
        if (len > 4) w1 = load_bytes(p + 4, len - 4);
      } else {
        const unsigned long long off = data_base + (unsigned long long)a;
        if (off > 0xffffffffull) *err = 1u;
        w0 = (unsigned long long)len | (load_bytes(p, 4) << 32);
        w1 = (off & 0xffffffffull) << 32;                 // buffer index 0, offset in the high half
      }
    }
    views[(size_t)i * 2] = w0; views[(size_t)i * 2 + 1] = w1;
  }
}
}  // namespace
void strviews_from_offsets(const void* offsets, bool large, const uint8_t* data, uint64_t data_base, int64_t data_len, int64_t n, uint64_t* views_out, const uint64_t* validity, int64_t row0,
                           bool stamp_nulls, unsigned int* err) {
  if (n == 0) return;
  ProfileScope ps("strviews_from_offsets", (uint64_t)n * ((large ? 8 : 4) + 16), (uint64_t)n);
  const int grid = grid_for(n, kBlock * 4);
  if (large) hipLaunchKernelGGL((strviews_from_offsets_kernel<long long>), dim3(grid), dim3(kBlock), 0, stream(), (const long long*)offsets, (const unsigned char*)data, (unsigned long long)data_base,
                                (long long)data_len, n, (unsigned long long*)views_out, validity, row0, stamp_nulls, err);
  else hipLaunchKernelGGL((strviews_from_offsets_kernel<int>), dim3(grid), dim3(kBlock), 0, stream(), (const int*)offsets, (const unsigned char*)data, (unsigned long long)data_base,
                          (long long)data_len, n, (unsigned long long*)views_out, validity, row0, stamp_nulls, err);
  PLX_HIP(hipGetLastError());
}

// ---- nulls of a view column that travels WITHOUT a bitmap (the raw-view interfaces: plx_strview_groupby, plx_strview_dict_encode_device, plx_ipc_read_string_views) ----
namespace {
__global__ __launch_bounds__(kBlock) void strview_stamp_kernel(unsigned long long* __restrict__ views, const uint64_t* __restrict__ validity, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (!((validity[i >> 6] >> (i & 63)) & 1)) { views[i * 2] = (unsigned long long)kStrviewNullLen; views[i * 2 + 1] = 0ull; }
}
}  // namespace
void strview_stamp_nulls(uint64_t* views, const uint64_t* validity, int64_t n) {
  if (n <= 0 || !validity) return;
  hipLaunchKernelGGL(strview_stamp_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), (unsigned long long*)views, validity, n);
  PLX_HIP(hipGetLastError());
}
// dictionary -> contiguous bytes + offsets[n + 1] on the device
void strdict_materialise(const uint64_t* dict_views, const uint8_t* data, int64_t n, Buf* out_offsets, Buf* out_bytes, uint64_t* total_bytes) {
  *out_offsets = dev_alloc(sizeof(uint64_t) * (size_t)(n + 2));
  if (n == 0) { PLX_HIP(hipMemsetAsync((*out_offsets)->ptr, 0, 16, stream())); *out_bytes = dev_alloc(16); *total_bytes = 0; return; }
  Buf lens = dev_alloc(sizeof(uint32_t) * (size_t)n);
  hipLaunchKernelGGL(strdict_len_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), (const unsigned long long*)dict_views, n, lens->as<uint32_t>());
  PLX_HIP(hipGetLastError());
  exclusive_scan_u32(lens->as<uint32_t>(), (*out_offsets)->as<uint64_t>(), n);
  d2h_sync(total_bytes, (*out_offsets)->as<uint64_t>() + n, 8);
  *out_bytes = dev_alloc((size_t)*total_bytes + 16);
  hipLaunchKernelGGL(strdict_copy_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, stream(), (const unsigned long long*)dict_views, (const unsigned char*)data, n,
                     (*out_offsets)->as<unsigned long long>(), (*out_bytes)->as<unsigned char>());
  PLX_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace plx
