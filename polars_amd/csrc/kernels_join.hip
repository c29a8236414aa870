// kernels_join.hip -- hash join build / probe / emit and key-hash partitioning.
//
// Reference algorithm (restated, not ported): build_tables makes 3 passes and one
// hashbrown map per partition (polars-ops/src/frame/join/hash_join/single_keys.rs:16-167);
// probe_inner looks every probe key up and emits (idx_a, idx_b) per build duplicate
// (single_keys_inner.rs:11-38).  GPU shape:
//   build  : one open-addressing table in HBM (keys[cap] claimed with 64-bit CAS, slot =
//            top bits of key * RANDOM_ODD like DirtyHash); duplicates form a chain:
//            head[slot] = newest row (atomicExch), next[row] = previous head.
//   probe 1: per probe row, find the slot and count the chain -> counts[i]
//   scan   : device exclusive scan -> output offsets (kernels_scan.hip)
//   probe 2: walk the chain again and write the pairs at offsets[i] (coalesced per row run)
// Keys are read in their physical dtype and widened in registers (no materialised
// 64-bit key copy); floats are canonicalised (-0 -> +0, one NaN: total_ord.rs:40-48).
// Null keys never match (nulls_equal = false).
#include "dev.hpp"
#include "fused.hpp"
#include "join.hpp"
#include "kernels.hpp"
#include "ops.hpp"
#include "scan.hpp"

namespace plx {
namespace join {

using namespace dev;
using k::kBlock;

constexpr uint64_t kEmpty = ~0ull;
constexpr uint32_t kNoRow = 0xffffffffu;
constexpr uint64_t kRandomOdd = 0x55fbfd6bfc5458e9ull;

struct KeyCol {
  const void* values;
  const uint64_t* validity;
  int dtype;
  int64_t n;
};

__device__ __forceinline__ uint64_t load_key(const KeyCol& kc, int64_t i) {
  switch (kc.dtype) {
    case PLX_I8: return (uint64_t)(long long)reinterpret_cast<const int8_t*>(kc.values)[i];
    case PLX_I16: return (uint64_t)(long long)reinterpret_cast<const int16_t*>(kc.values)[i];
    case PLX_I32: return (uint64_t)(long long)reinterpret_cast<const int32_t*>(kc.values)[i];
    case PLX_U8: return reinterpret_cast<const uint8_t*>(kc.values)[i];
    case PLX_U16: return reinterpret_cast<const uint16_t*>(kc.values)[i];
    case PLX_U32: return reinterpret_cast<const uint32_t*>(kc.values)[i];
    case PLX_F32: { float f = reinterpret_cast<const float*>(kc.values)[i]; double d = (double)f; return (d != d) ? 0x7ff8000000000000ull : (uint64_t)__double_as_longlong(d + 0.0); }
    case PLX_F64: { double d = reinterpret_cast<const double*>(kc.values)[i]; return (d != d) ? 0x7ff8000000000000ull : (uint64_t)__double_as_longlong(d + 0.0); }
    case PLX_BOOL: return (reinterpret_cast<const uint64_t*>(kc.values)[i >> 6] >> (i & 63)) & 1;
    default: return reinterpret_cast<const uint64_t*>(kc.values)[i];
  }
}
__device__ __forceinline__ bool key_valid(const KeyCol& kc, int64_t i) { return !kc.validity || ((kc.validity[i >> 6] >> (i & 63)) & 1); }

struct Table {
  unsigned long long* keys;  // [cap + 1]; slot cap = the key whose bits equal kEmpty
  unsigned int* head;        // [cap + 1]
  unsigned int* next;        // [n_build]
  unsigned int* flags;       // [0] = a chain longer than 1 exists (build keys not unique)
  uint32_t log2_cap;
};

__device__ __forceinline__ int64_t find_or_claim(const Table& t, uint64_t key) {
  const uint64_t cap = 1ull << t.log2_cap;
  if (key == kEmpty) return (int64_t)cap;
  uint64_t slot = (key * kRandomOdd) >> (64 - t.log2_cap);
  for (;;) {
    unsigned long long cur = t.keys[slot];
    if (cur == key) return (int64_t)slot;
    if (cur == kEmpty) {
      unsigned long long old = atomicCAS(&t.keys[slot], (unsigned long long)kEmpty, (unsigned long long)key);
      if (old == kEmpty || old == key) return (int64_t)slot;
    }
    slot = (slot + 1) & (cap - 1);
  }
}
__device__ __forceinline__ int64_t find_slot(const Table& t, uint64_t key) {
  const uint64_t cap = 1ull << t.log2_cap;
  if (key == kEmpty) return (int64_t)cap;
  uint64_t slot = (key * kRandomOdd) >> (64 - t.log2_cap);
  for (;;) {
    unsigned long long cur = t.keys[slot];
    if (cur == key) return (int64_t)slot;
    if (cur == kEmpty) return -1;
    slot = (slot + 1) & (cap - 1);
  }
}

__global__ __launch_bounds__(kBlock) void join_build_kernel(KeyCol build, Table t) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < build.n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!key_valid(build, i)) { t.next[i] = kNoRow; continue; }
    const int64_t slot = find_or_claim(t, load_key(build, i));
    const unsigned int old = atomicExch(&t.head[slot], (unsigned int)i);
    t.next[i] = old;
    if (old != kNoRow) t.flags[0] = 1u;
  }
}

// counts[i] = number of build matches of probe row i (left join: at least 1)
__global__ __launch_bounds__(kBlock) void join_count_kernel(KeyCol probe, Table t, int how, uint32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < probe.n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c = 0;
    if (key_valid(probe, i)) {
      const int64_t slot = find_slot(t, load_key(probe, i));
      if (slot >= 0) { for (unsigned int r = t.head[slot]; r != kNoRow; r = t.next[r]) c++; }
    }
    // how: 0 inner, 1 left (unmatched rows emit one pair), 2 semi (row kept once if matched), 3 anti (kept if unmatched; null keys never match)
    counts[i] = how == 2 ? (c ? 1u : 0u) : how == 3 ? (c ? 0u : 1u) : (how == 1 && c == 0) ? 1u : c;
  }
}

// semi / anti: the kept probe rows, in probe order (single_keys_semi_anti.rs keeps left order the same way)
__global__ __launch_bounds__(kBlock) void join_emit_kept_kernel(const uint32_t* __restrict__ counts, const uint64_t* __restrict__ offsets, int64_t n, uint32_t* __restrict__ out_probe) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (counts[i]) out_probe[offsets[i]] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void join_emit_kernel(KeyCol probe, Table t, int left_join, const uint64_t* __restrict__ offsets,
                                                           uint32_t* __restrict__ out_probe, uint32_t* __restrict__ out_build) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < probe.n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t o = offsets[i];
    bool any = false;
    if (key_valid(probe, i)) {
      const int64_t slot = find_slot(t, load_key(probe, i));
      if (slot >= 0) {
        for (unsigned int r = t.head[slot]; r != kNoRow; r = t.next[r]) { out_probe[o] = (uint32_t)i; out_build[o] = r; o++; any = true; }
      }
    }
    if (left_join && !any) { out_probe[o] = (uint32_t)i; out_build[o] = kNoRow; }
  }
}

static KeyCol key_col(const ColumnPtr& c) { KeyCol kc; kc.values = c->data(); kc.validity = c->valid_words(); kc.dtype = c->dtype; kc.n = c->len; return kc; }
static int ceil_log2(uint64_t x) { int b = 0; while ((1ull << b) < x) b++; return b; }

void join_indices(int how, const ColumnPtr& left_key, const ColumnPtr& right_key, ColumnPtr& left_idx, ColumnPtr& right_idx, std::string* desc) {
  PLX_REQUIRE(left_key->dtype == right_key->dtype, PLX_ERR_INVALID,
              std::string("join keys have different dtypes (") + dtype_name(left_key->dtype) + ", " + dtype_name(right_key->dtype) + ")");
  PLX_REQUIRE(how == PLX_JOIN_INNER || how == PLX_JOIN_LEFT || how == PLX_JOIN_SEMI || how == PLX_JOIN_ANTI, PLX_ERR_UNSUPPORTED, "join type outside the hot path");
  PLX_REQUIRE(left_key->len < 0xffffffffll && right_key->len < 0xffffffffll, PLX_ERR_UNSUPPORTED, "join side exceeds u32 IdxSize");
  const bool left_join = how == PLX_JOIN_LEFT;
  const bool semi_anti = how == PLX_JOIN_SEMI || how == PLX_JOIN_ANTI;
  // det_hash_prone_order (hash_join/mod.rs:41-50): build on the shorter relation; left / semi / anti joins build on the right
  const bool swapped = !left_join && !semi_anti && !(left_key->len > right_key->len);
  const ColumnPtr& probe = (left_join || semi_anti) ? left_key : (swapped ? right_key : left_key);
  const ColumnPtr& build = (left_join || semi_anti) ? right_key : (swapped ? left_key : right_key);
  const int log2_cap = std::max(4, ceil_log2((uint64_t)std::max<int64_t>(build->len, 1) * 2));
  const uint64_t cap = 1ull << log2_cap;
  Buf keys = dev_alloc(sizeof(uint64_t) * (cap + 1));
  Buf head = dev_alloc(sizeof(uint32_t) * (cap + 1));
  Buf next = dev_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(build->len, 1));
  Buf flags = dev_alloc_zero(16);
  PLX_HIP(hipMemsetAsync(keys->ptr, 0xff, sizeof(uint64_t) * (cap + 1), stream()));
  PLX_HIP(hipMemsetAsync(head->ptr, 0xff, sizeof(uint32_t) * (cap + 1), stream()));
  Table t; t.keys = keys->as<unsigned long long>(); t.head = head->as<unsigned int>(); t.next = next->as<unsigned int>(); t.flags = flags->as<unsigned int>(); t.log2_cap = (uint32_t)log2_cap;
  const int kw = dtype_width(build->dtype) ? dtype_width(build->dtype) : 1;
  if (build->len) {
    ProfileScope ps("join_build", (uint64_t)build->len * kw, (uint64_t)build->len);
    hipLaunchKernelGGL(join_build_kernel, dim3(k::grid_for(build->len, kBlock * 2)), dim3(kBlock), 0, stream(), key_col(build), t);
    PLX_HIP(hipGetLastError());
  }
  const int64_t np = probe->len;
  Buf counts = dev_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(np, 1));
  Buf offsets = dev_alloc(sizeof(uint64_t) * (size_t)(np + 1));
  if (np) {
    ProfileScope ps("join_probe_count", (uint64_t)np * (kw + 4), (uint64_t)np);
    hipLaunchKernelGGL(join_count_kernel, dim3(k::grid_for(np, kBlock * 2)), dim3(kBlock), 0, stream(), key_col(probe), t, how, counts->as<uint32_t>());
    PLX_HIP(hipGetLastError());
  }
  k::exclusive_scan_u32(counts->as<uint32_t>(), offsets->as<uint64_t>(), np);
  uint64_t total = 0;
  d2h_sync(&total, offsets->as<uint64_t>() + np, 8);
  auto mk_idx = [&](int64_t n) { auto c = std::make_shared<Column>(); c->dtype = PLX_U32; c->len = n; c->values = dev_alloc(values_bytes(PLX_U32, n)); c->null_count = 0; return c; };
  if (semi_anti) {
    ColumnPtr kept = mk_idx((int64_t)total);
    if (total) {
      ProfileScope ps("join_emit_kept", (uint64_t)np * 12 + total * 4, (uint64_t)np);
      hipLaunchKernelGGL(join_emit_kept_kernel, dim3(k::grid_for(np, kBlock * 2)), dim3(kBlock), 0, stream(), counts->as<uint32_t>(), offsets->as<uint64_t>(), np, kept->values->as<uint32_t>());
      PLX_HIP(hipGetLastError());
    }
    if (desc) *desc = std::string(how == PLX_JOIN_SEMI ? "hash_semi_join" : "hash_anti_join") + "[build=right rows=" + std::to_string(build->len) + " cap=2^" + std::to_string(log2_cap) +
                      ", probe rows=" + std::to_string(np) + ", kept=" + std::to_string(total) + "]";
    left_idx = kept; right_idx = nullptr;
    return;
  }
  ColumnPtr pidx = mk_idx((int64_t)total), bidx = mk_idx((int64_t)total);
  if (total) {
    ProfileScope ps("join_probe_emit", (uint64_t)np * (kw + 8) + total * 8, (uint64_t)np);
    hipLaunchKernelGGL(join_emit_kernel, dim3(k::grid_for(np, kBlock * 2)), dim3(kBlock), 0, stream(), key_col(probe), t, left_join ? 1 : 0, offsets->as<uint64_t>(),
                       pidx->values->as<uint32_t>(), bidx->values->as<uint32_t>());
    PLX_HIP(hipGetLastError());
  }
  if (left_join && total) {
    // unmatched rows carry the kNoRow sentinel -> validity bitmap
    plx_scalar s; s.u = kNoRow;
    ColumnPtr ok = ops::cmp_scalar(PLX_NE, bidx, s);
    bidx->validity = ok->values; bidx->null_count = -1;
    if (column_null_count(bidx) == 0) { bidx->validity = nullptr; bidx->null_count = 0; }
  }
  if (desc) {
    uint32_t f = 0; d2h_sync(&f, flags->ptr, 4);
    *desc = std::string("hash_join[build=") + (left_join ? "right" : (swapped ? "left" : "right")) + " rows=" + std::to_string(build->len) + " cap=2^" + std::to_string(log2_cap) +
            (f ? " dup-keys" : " unique-keys") + ", probe rows=" + std::to_string(np) + ", pairs=" + std::to_string(total) + "]";
  }
  if (left_join || !swapped) { left_idx = pidx; right_idx = bidx; }
  else { left_idx = bidx; right_idx = pidx; }
}

// ------------------------------------------------ materialising join: pairs over a candidate list ---
// The frame-returning join of the reference (polars-ops/src/frame/join/mod.rs:564-652 _inner_join_from_series: pairs from hash_join/single_keys_inner.rs:40-149,
// then one gather per side) on the machinery of the fused join -> group-by: the build side is the fused::JoinAggTable its build scan fills (16-byte {key, row}
// slots; duplicate build keys: chains through links[]), the probe side arrives as a CANDIDATE list -- the rows that passed the probe side's predicate and the
// partitioned LDS filters of kernels_partition.hip (probe_hits_impl), a few per cent of the probe side for a selective build -- so the only random walk through
// HBM is one slot lookup per candidate, once: pass 1 leaves the build row (or chain head) and the pair count of every candidate, the pairs are then laid out by a
// device scan (duplicate keys / left joins) or by the selection-bitmap compaction of kernels_filter.hip (unique keys: 0 / 1 pairs per candidate) -- no second probe.
struct PairTable {
  const unsigned long long* slots; const unsigned long long* links; uint32_t log2_cap, log2_window;
  // direct-address variant (bits != null; unique build keys): bitmap over the key range + rank per word + slot -> build row (fused::DirectJoinTable, k::direct_slot_rows)
  const unsigned long long* bits; const unsigned int* rank; const unsigned int* slot_row; long long kmin; unsigned long long range;
};
__device__ __forceinline__ unsigned int pair_lookup(const PairTable& t, uint64_t key) {
  if (t.bits) {                                   // wave-uniform
    const uint64_t idx = key - (uint64_t)t.kmin;
    if (idx >= t.range) return kNoRow;
    const unsigned long long w = t.bits[idx >> 6];
    if (!((w >> (idx & 63)) & 1ull)) return kNoRow;
    return t.slot_row[(unsigned long long)t.rank[idx >> 6] + (unsigned long long)__popcll(w & ((1ull << (idx & 63)) - 1ull))];
  }
  const uint64_t cap = 1ull << t.log2_cap;
  if (key == fused::kEmptyKey) return (unsigned int)t.slots[cap * 2 + 1];                   // the key equal to the EMPTY pattern lives in slot `cap` (row kNoRow when absent)
  uint64_t slot = (key * fused::kP2HashMult) >> (64 - t.log2_cap);
  for (uint64_t n = 0; n <= cap; n++) {
    const unsigned long long cur = t.slots[slot * 2];
    if (cur == key) return (unsigned int)t.slots[slot * 2 + 1];
    if (cur == fused::kEmptyKey) return kNoRow;
    const uint64_t wmask = (1ull << (t.log2_window ? t.log2_window : t.log2_cap)) - 1ull;      // probe sequences wrap inside the table's windows (fused::jt_next)
    slot = (slot & ~wmask) | ((slot + 1) & wmask);
  }
  return kNoRow;
}
// head[i] = build row (multi-value: head of the chain) of candidate i or kNoRow; cnt[i] (may be null) = pairs it emits; mask (may be null) bit i = it has a match
__global__ __launch_bounds__(kBlock) void join_match_kernel(KeyCol probe, const uint32_t* __restrict__ cand, int64_t n, PairTable t, int left_join, uint32_t* __restrict__ head,
                                                            uint32_t* __restrict__ cnt, unsigned long long* __restrict__ mask) {
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t i = base + lane;
    unsigned int h = kNoRow, c = 0;
    if (i < n) {
      const int64_t row = cand ? (int64_t)cand[i] : i;
      if (key_valid(probe, row)) h = pair_lookup(t, load_key(probe, row));
      if (h != kNoRow) {
        c = 1;
        if (t.links) { unsigned int r = (unsigned int)t.links[h]; for (uint32_t g = 0; r != kNoRow && g < (1u << 24); g++) { c++; r = (unsigned int)t.links[r]; } }
      } else if (left_join) c = 1;
      head[i] = h;
      if (cnt) cnt[i] = c;
    }
    if (mask) { const uint64_t m = ballot(h != kNoRow); if (lane == 0) mask[base >> 6] = m; }
  }
}
__global__ __launch_bounds__(kBlock) void join_pairs_emit_kernel(const uint32_t* __restrict__ cand, int64_t n, const unsigned long long* __restrict__ links, const uint32_t* __restrict__ head,
                                                                 const uint64_t* __restrict__ off, uint32_t* __restrict__ out_probe, uint32_t* __restrict__ out_build) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t o = off[i];
    const uint64_t end = off[i + 1];
    if (o == end) continue;
    const uint32_t row = cand ? cand[i] : (uint32_t)i;
    unsigned int r = head[i];
    if (r == kNoRow) { out_probe[o] = row; out_build[o] = kNoRow; continue; }               // left join: no match
    for (; o < end && r != kNoRow; o++) { out_probe[o] = row; out_build[o] = r; r = links ? (unsigned int)links[r] : kNoRow; }
  }
}
__global__ __launch_bounds__(kBlock) void iota_u32_kernel(uint32_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}

static void join_pairs_impl(int how, const ColumnPtr& probe_key, const ColumnPtr& cand, const PairTable& t, ColumnPtr& probe_idx, ColumnPtr& build_idx, std::string* desc);
void join_pairs(int how, const ColumnPtr& probe_key, const ColumnPtr& cand, const fused::JoinAggTable& jt, ColumnPtr& probe_idx, ColumnPtr& build_idx, std::string* desc) {
  PairTable t{}; t.slots = jt.slots; t.links = jt.links; t.log2_cap = jt.log2_cap; t.log2_window = jt.log2_window;
  join_pairs_impl(how, probe_key, cand, t, probe_idx, build_idx, desc);
}
void join_pairs_direct(int how, const ColumnPtr& probe_key, const ColumnPtr& cand, const fused::DirectJoinTable& dt, const uint32_t* slot_row, ColumnPtr& probe_idx, ColumnPtr& build_idx,
                       std::string* desc) {
  PairTable t{}; t.bits = dt.bits; t.rank = dt.rank; t.slot_row = slot_row; t.kmin = dt.kmin; t.range = dt.range;
  join_pairs_impl(how, probe_key, cand, t, probe_idx, build_idx, desc);
}
static void join_pairs_impl(int how, const ColumnPtr& probe_key, const ColumnPtr& cand, const PairTable& t, ColumnPtr& probe_idx, ColumnPtr& build_idx, std::string* desc) {
  PLX_REQUIRE(how == PLX_JOIN_INNER || how == PLX_JOIN_LEFT, PLX_ERR_UNSUPPORTED, "join_pairs: inner and left joins");
  PLX_REQUIRE(probe_key->len < 0xffffffffll, PLX_ERR_UNSUPPORTED, "join side exceeds u32 IdxSize");
  const bool left = how == PLX_JOIN_LEFT, multi = t.links != nullptr;
  const int64_t n = cand ? cand->len : probe_key->len;
  auto mk_idx = [&](int64_t m) { auto c = std::make_shared<Column>(); c->dtype = PLX_U32; c->len = m; c->values = dev_alloc(values_bytes(PLX_U32, std::max<int64_t>(m, 1))); c->null_count = 0; return c; };
  auto null_out_no_row = [&](ColumnPtr& bidx) {          // unmatched rows of a left join carry the kNoRow sentinel -> validity bitmap
    if (!bidx->len) return;
    plx_scalar s; s.u = kNoRow;
    ColumnPtr ok = ops::cmp_scalar(PLX_NE, bidx, s);
    bidx->validity = ok->values; bidx->null_count = -1;
    if (column_null_count(bidx) == 0) { bidx->validity = nullptr; bidx->null_count = 0; }
  };
  if (n == 0) { probe_idx = mk_idx(0); build_idx = mk_idx(0); if (desc) *desc = "join_pairs[no candidates]"; return; }
  const uint32_t* cp = cand ? cand->values->as<uint32_t>() : nullptr;
  const int kw = dtype_width(probe_key->dtype) ? dtype_width(probe_key->dtype) : 1;
  ColumnPtr head = mk_idx(n);
  const bool counted = multi;                           // unique build keys: 0 / 1 pairs per candidate (left join: exactly 1), no scan
  Buf cnt = counted ? dev_alloc(sizeof(uint32_t) * (size_t)n) : nullptr;
  const int64_t nwords = (n + 63) >> 6;
  Buf mask = (!counted && !left) ? dev_alloc(sizeof(uint64_t) * (size_t)nwords) : nullptr;
  {
    ProfileScope ps("join_match", (uint64_t)n * (kw + 16 + (cand ? 4 : 0) + 4), (uint64_t)n);
    hipLaunchKernelGGL(join_match_kernel, dim3(k::grid_for(n, kBlock * 2)), dim3(kBlock), 0, stream(), key_col(probe_key), cp, n, t, left ? 1 : 0, head->values->as<uint32_t>(),
                       cnt ? cnt->as<uint32_t>() : nullptr, mask ? mask->as<unsigned long long>() : nullptr);
    PLX_HIP(hipGetLastError());
  }
  uint64_t total = 0;
  if (!counted && left) {
    // one pair per candidate: the candidate list IS the probe index, the heads are the build index
    if (cand) probe_idx = cand;
    else { probe_idx = mk_idx(n); hipLaunchKernelGGL(iota_u32_kernel, dim3(k::grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), probe_idx->values->as<uint32_t>(), n); PLX_HIP(hipGetLastError()); }
    build_idx = head;
    null_out_no_row(build_idx);
    total = (uint64_t)n;
  } else if (!counted) {
    const k::FilterPlan fp = k::filter_prepare(mask->as<uint64_t>(), n);
    total = (uint64_t)fp.n_out;
    probe_idx = mk_idx(fp.n_out); build_idx = mk_idx(fp.n_out);
    if (cand) k::filter_apply(fp, 4, cp, nullptr, probe_idx->values->ptr, nullptr);
    else k::filter_rowids(fp, probe_idx->values->as<uint32_t>());
    k::filter_apply(fp, 4, head->values->ptr, nullptr, build_idx->values->ptr, nullptr);
    PLX_HIP(hipStreamSynchronize(stream()));             // `mask` / the plan's offsets are read by the compactions
  } else {
    Buf off = dev_alloc(sizeof(uint64_t) * (size_t)(n + 1));
    k::exclusive_scan_u32(cnt->as<uint32_t>(), off->as<uint64_t>(), n);
    d2h_sync(&total, off->as<uint64_t>() + n, 8);
    PLX_REQUIRE(total < 0xffffffffull, PLX_ERR_UNSUPPORTED, "join output exceeds u32 IdxSize");
    probe_idx = mk_idx((int64_t)total); build_idx = mk_idx((int64_t)total);
    if (total) {
      ProfileScope ps("join_pairs_emit", (uint64_t)n * 24 + total * 8, (uint64_t)n);
      hipLaunchKernelGGL(join_pairs_emit_kernel, dim3(k::grid_for(n, kBlock * 2)), dim3(kBlock), 0, stream(), cp, n, t.links, head->values->as<uint32_t>(), off->as<uint64_t>(),
                         probe_idx->values->as<uint32_t>(), build_idx->values->as<uint32_t>());
      PLX_HIP(hipGetLastError());
    }
    if (left) null_out_no_row(build_idx);
    PLX_HIP(hipStreamSynchronize(stream()));
  }
  if (desc) *desc = std::string("join_pairs[") + (cand ? "candidates=" : "rows=") + std::to_string(n) + (multi ? ", multi-value chains" : t.bits ? ", direct-address table" : ", unique build keys") + " -> match" +
                    (counted ? "+scan+emit" : left ? "" : "+bitmap compaction") + ", pairs=" + std::to_string(total) + "]";
}

// -------------------------------------------------------------- partitioning ---
__device__ __forceinline__ uint32_t partition_of(const KeyCol& kc, int64_t i, uint64_t seed, uint32_t n_parts) {
  if (!key_valid(kc, i)) return 0;  // null_partition() == 0
  const uint64_t h = load_key(kc, i) * kRandomOdd;  // dirty_hash
  return (uint32_t)__umul64hi(h * seed, (uint64_t)n_parts);
}
// Two passes, no device atomics on the row path (a wave-aggregated bump of 8 global cursors took 32 ms for 3.2e8 rows: a few hot addresses serialise):
// every workgroup owns a contiguous slab of rows; pass 1 leaves its per-partition counts, a scan over (partition, workgroup) turns them into the start of every
// (workgroup, partition) run, pass 2 ranks its rows with LDS cursors and writes the permutation.  Rows of a partition keep their workgroup order (slabs ascend).
__global__ __launch_bounds__(kBlock) void partition_count_kernel(KeyCol kc, uint64_t seed, uint32_t n_parts, int64_t per_block, unsigned long long* __restrict__ block_counts /* [n_parts][grid] */,
                                                                 unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int lcount[];
  for (uint32_t p = threadIdx.x; p < n_parts; p += blockDim.x) lcount[p] = 0;
  __syncthreads();
  const int64_t beg = (int64_t)blockIdx.x * per_block, end = beg + per_block < kc.n ? beg + per_block : kc.n;
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&lcount[partition_of(kc, i, seed, n_parts)], 1u);
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < n_parts; p += blockDim.x) {
    block_counts[(size_t)p * gridDim.x + blockIdx.x] = lcount[p];
    if (lcount[p]) atomicAdd(&counts[p], (unsigned long long)lcount[p]);
  }
}
__global__ __launch_bounds__(kBlock) void partition_scatter_kernel(KeyCol kc, uint64_t seed, uint32_t n_parts, int64_t per_block, const unsigned long long* __restrict__ block_starts /* [n_parts][grid] */,
                                                                   uint32_t* __restrict__ perm) {
  extern __shared__ unsigned long long lcur[];       // [n_parts] next output position of this workgroup's run in each partition
  for (uint32_t p = threadIdx.x; p < n_parts; p += blockDim.x) lcur[p] = block_starts[(size_t)p * gridDim.x + blockIdx.x];
  __syncthreads();
  const int lane = lane_id();
  const int64_t beg = (int64_t)blockIdx.x * per_block, end = beg + per_block < kc.n ? beg + per_block : kc.n;
  for (int64_t base = beg + (threadIdx.x - lane); base < end; base += blockDim.x) {
    const int64_t i = base + lane;
    const bool active = i < end;
    const uint32_t p = active ? partition_of(kc, i, seed, n_parts) : 0xffffffffu;
    if (n_parts <= 64) {
      // few partitions: one LDS atomic per (wave, partition present), lanes take consecutive positions
      uint64_t todo = ballot(active);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lp = __shfl(p, leader, 64);
        const uint64_t same = ballot(active && p == lp);
        unsigned long long o = 0;
        if (lane == leader) o = atomicAdd(&lcur[lp], (unsigned long long)popc64(same));
        o = shfl_u64(o, leader);
        if (active && p == lp) perm[o + (uint64_t)prefix_rank(same)] = (uint32_t)i;
        todo &= ~same;
      }
    } else if (active) {
      perm[atomicAdd(&lcur[p], 1ull)] = (uint32_t)i;
    }
  }
}

void hash_partition_dev(const ColumnPtr& key, int n_partitions, uint64_t seed, ColumnPtr& perm, Buf& counts) {
  PLX_REQUIRE(n_partitions >= 1 && n_partitions <= 4096, PLX_ERR_INVALID, "hash_partition: 1..4096 partitions");
  // HashPartitioner::new seed mixing (hashing.rs:81-96)
  auto fold = [](uint64_t a, uint64_t b) { unsigned __int128 r = (unsigned __int128)a * b; return (uint64_t)r ^ (uint64_t)(r >> 64); };
  uint64_t s = fold(seed ^ 0x85921e81c41226a0ull, 0x3bc1d0faba166294ull);
  s = fold(s, 0xfbde893e21a73756ull) | 1;
  const int64_t n = key->len;
  perm = std::make_shared<Column>();
  perm->dtype = PLX_U32; perm->len = n; perm->values = dev_alloc(values_bytes(PLX_U32, n)); perm->null_count = 0;
  counts = dev_alloc_zero(sizeof(uint64_t) * (size_t)n_partitions);
  if (n) {
    ProfileScope ps("hash_partition", (uint64_t)n * (dtype_width(key->dtype) * 2 + 4), (uint64_t)n);
    // slabs of whole waves; enough workgroups to fill the chip, few enough that the (partition x workgroup) table stays small
    const int64_t max_blocks = std::max<int64_t>(1, std::min<int64_t>(2048, ((int64_t)1 << 22) / n_partitions));
    int64_t per_block = (n + max_blocks - 1) / max_blocks;
    per_block = std::max<int64_t>(kBlock, (per_block + kBlock - 1) / kBlock * kBlock);
    const int grid = (int)((n + per_block - 1) / per_block);
    const size_t cells = (size_t)n_partitions * (size_t)grid;
    Buf block_counts = dev_alloc(sizeof(uint64_t) * (cells + 1)), block_starts = dev_alloc(sizeof(uint64_t) * (cells + 1));
    hipLaunchKernelGGL(partition_count_kernel, dim3(grid), dim3(kBlock), sizeof(unsigned int) * (size_t)n_partitions, stream(), key_col(key), s, (uint32_t)n_partitions, per_block,
                       block_counts->as<unsigned long long>(), counts->as<unsigned long long>());
    PLX_HIP(hipGetLastError());
    k::exclusive_scan_u64(block_counts->as<uint64_t>(), block_starts->as<uint64_t>(), (int64_t)cells);
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(grid), dim3(kBlock), sizeof(unsigned long long) * (size_t)n_partitions, stream(), key_col(key), s, (uint32_t)n_partitions, per_block,
                       block_starts->as<unsigned long long>(), perm->values->as<uint32_t>());
    PLX_HIP(hipGetLastError());
  }
}

void hash_partition(const ColumnPtr& key, int n_partitions, uint64_t seed, ColumnPtr& perm, int64_t* counts_out) {
  Buf counts;
  hash_partition_dev(key, n_partitions, seed, perm, counts);
  std::vector<uint64_t> h((size_t)n_partitions, 0);
  if (key->len) d2h_sync(h.data(), counts->ptr, sizeof(uint64_t) * (size_t)n_partitions);
  for (int p = 0; p < n_partitions; p++) counts_out[p] = (int64_t)h[p];
}

}  // namespace join
}  // namespace plx
