// kernels_partition.hip -- high-cardinality hash group-by without global atomics on the row path.
//
// Why: the fused HBM-table sink (HashAggSink) issues one CAS + one atomic per aggregate per ROW.
// MI355X sustains ~24 G device-scope atomics/s chip-wide (tools/micro_atomics.hip; the same at
// workgroup scope on an XCD-private slice), so 1e9 rows x 2 aggregates cost 87 ms -- 4 % of the HBM
// roofline -- while LDS atomics keep up with HBM streaming (4e8 rows x 2 atomics in 1.7 ms).
//
// Shape of the reference (restated, not ported): the streaming group-by pre-aggregates per thread, then
// hash-partitions keys and finalises each partition in its own table
// (polars-stream/src/nodes/group_by.rs:140-250,252-497; HashPartitioner polars-utils/src/hashing.rs:72-121).
// Here, three passes:
//   1. part_count    fused scan (predicate + expressions as usual); partition = top bits of the key hash;
//                    per-workgroup LDS histogram -> global histogram -> exclusive scan = partition offsets
//   2. part_scatter  same fused scan; each row becomes a record [key, source values..., (validity), (row id)];
//                    records are staged in workgroup-shared LDS write-combining buffers (P x B records) and
//                    flushed B at a time (a whole 128-B line for 16-B records) at an offset reserved with ONE
//                    global atomic per flush
//   3. part_agg      one workgroup per partition: LDS open-addressing table (CAS on the key word, LDS atomics
//                    on the cells); when the partition is done its groups are written straight to the dense
//                    output arrays (one global atomic per partition) -- no table in HBM, no compaction pass
// HBM traffic: (1) reads the key/predicate columns, (2) reads all inputs + writes the records, (3) reads the
// records: ~3x the algorithmic bytes instead of a fraction of the atomic rate.
#include "fused_device.hpp"
#include "kernels_fused.hpp"
#include "scan.hpp"

namespace plx {
namespace k {

using namespace dev;
using namespace fused;

__device__ __forceinline__ uint32_t part_of(uint64_t key, bool kvalid, uint32_t log2_parts) {
  if (!kvalid) return 0;   // null_partition() == 0 (hashing.rs:111-115)
  return (uint32_t)((key * 0x55fbfd6bfc5458e9ull) >> (64 - log2_parts));
}

// ---- pass 1: histogram (runs inside fused_scan_kernel) -------------------------------------------
struct PartCountSink {
  struct Params { unsigned long long* hist; uint32_t log2_parts; };
  template <class S> __device__ __forceinline__ void init(const S&, const Params& p) {
    extern __shared__ unsigned long long lds_raw[];
    unsigned int* cnt = reinterpret_cast<unsigned int*>(lds_raw);
    for (uint32_t i = threadIdx.x; i < (1u << p.log2_parts); i += blockDim.x) cnt[i] = 0;
    __syncthreads();
  }
  template <class S> __device__ __forceinline__ void consume(const S& sh, const RegFile& rf, const bool pass[kRows], int64_t, const Params& p) {
    extern __shared__ unsigned long long lds_raw[];
    unsigned int* cnt = reinterpret_cast<unsigned int*>(lds_raw);
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r]) continue;
      atomicAdd(&cnt[part_of(rf.v[r][sh.key], (rf.valid[sh.key] >> r) & 1, p.log2_parts)], 1u);
    }
  }
  template <class S> __device__ __forceinline__ void finish(const S&, const Params& p) {
    extern __shared__ unsigned long long lds_raw[];
    unsigned int* cnt = reinterpret_cast<unsigned int*>(lds_raw);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (1u << p.log2_parts); i += blockDim.x) if (cnt[i]) atomicAdd(&p.hist[i], (unsigned long long)cnt[i]);
  }
};

template <class P>
__global__ __launch_bounds__(kBlock) void part_count_kernel(Shape dsh, Args args, PartCountSink::Params sp) {
  PartCountSink sink;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ntiles = (args.n_rows + kTileRows - 1) / kTileRows;
  if constexpr (P::kStatic) {
    constexpr Shape sh = static_shape(P::kId);
    sink.init(sh, sp);
    for (int64_t t = wave; t < ntiles; t += nwaves) { RegFile rf; bool pass[kRows]; int64_t row0; tile_rows<P>(dsh, args, t, rf, pass, row0); sink.consume(sh, rf, pass, row0, sp); }
    sink.finish(sh, sp);
  } else {
    sink.init(dsh, sp);
    for (int64_t t = wave; t < ntiles; t += nwaves) { RegFile rf; bool pass[kRows]; int64_t row0; tile_rows<P>(dsh, args, t, rf, pass, row0); sink.consume(dsh, rf, pass, row0, sp); }
    sink.finish(dsh, sp);
  }
}

// ---- pass 2: scatter through LDS write-combining buffers ------------------------------------------
// Workgroup-synchronous tile loop (every wave of the workgroup runs the same number of iterations, so
// __syncthreads inside the loop is legal -- unlike fused_scan_kernel, whose tiles are handed out per wave).
constexpr int kMaxSrc = 8;   // distinct aggregate sources carried by a record

struct Rec {
  uint64_t key, vbits, rowid;
  uint64_t src[kMaxSrc];
};
// Register-resident record (all indices compile-time: a dynamically indexed array would live in scratch).
template <class S>
__device__ __forceinline__ void make_record(const S& sh, const PartitionPlan& pp, const RegFile& rf, int r, int64_t row, Rec& rec) {
  const bool kvalid = (rf.valid[sh.key] >> r) & 1;
  rec.key = kvalid ? rf.v[r][sh.key] : 0ull;
  rec.vbits = kvalid ? (1ull << 63) : 0ull;
  rec.rowid = (uint64_t)row;
#pragma unroll
  for (int j = 0; j < kMaxSrc; j++) {
    rec.src[j] = 0;
    if (j < (int)pp.n_src) {
      rec.src[j] = rf.v[r][pp.src_slot[j]];
      if ((rf.valid[pp.src_slot[j]] >> r) & 1) rec.vbits |= 1ull << j;
    }
  }
}
__device__ __forceinline__ void store_record(unsigned long long* dst, const PartitionPlan& pp, const Rec& rec) {
  dst[0] = rec.key;
#pragma unroll
  for (int j = 0; j < kMaxSrc; j++) if (j < (int)pp.n_src) dst[1 + j] = rec.src[j];
  uint32_t w = 1 + pp.n_src;
  if (pp.has_valid) dst[w++] = rec.vbits;
  if (pp.has_rowid) dst[w] = rec.rowid;
}

template <class P>
__global__ __launch_bounds__(kBlock) void part_scatter_kernel(Shape dsh, Args args, PartitionPlan pp, unsigned long long* __restrict__ cursor,
                                                              unsigned long long* __restrict__ out) {
  extern __shared__ unsigned long long lds_raw[];
  const uint32_t NP = 1u << pp.log2_parts, B = pp.buf_rows, R = pp.rec_words;
  unsigned long long* buf = lds_raw;                                        // [NP][B][R]
  unsigned long long* fbase = buf + (size_t)NP * B * R;                     // [NP] global record index of a flush
  unsigned int* cnt = reinterpret_cast<unsigned int*>(fbase + NP);          // [NP]
  unsigned int* flist = cnt + NP;                                           // [NP]
  unsigned int& nflush = flist[NP];                                         // kept in the dynamic region: a static __shared__ in front of it
                                                                            // would break the 16-byte alignment the ulonglong2 copies need
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  const int64_t rows_per_block_tile = (int64_t)kBlock * kRows;
  const int64_t nbt = (args.n_rows + rows_per_block_tile - 1) / rows_per_block_tile;
  const int wave_in_block = threadIdx.x >> 6;
  for (int64_t bt = blockIdx.x; bt < nbt; bt += gridDim.x) {
    RegFile rf; bool pass[kRows]; int64_t row0;
    const int64_t tile = bt * (kBlock / 64) + wave_in_block;
    tile_rows<P>(dsh, args, tile, rf, pass, row0);
    Rec rec[kRows];
    uint32_t part[kRows];
    bool pending[kRows];
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      pending[r] = pass[r];
      if constexpr (P::kStatic) { constexpr Shape sh = static_shape(P::kId); make_record(sh, pp, rf, r, row0 + r, rec[r]); }
      else make_record(dsh, pp, rf, r, row0 + r, rec[r]);
      part[r] = part_of(rec[r].key, (rec[r].vbits >> 63) & 1, pp.log2_parts);
    }
    int any;
    do {
      if (threadIdx.x == 0) nflush = 0;
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        if (!pending[r]) continue;
        const unsigned int pos = atomicAdd(&cnt[part[r]], 1u);
        if (pos < B) {
          store_record(buf + ((size_t)part[r] * B + pos) * R, pp, rec[r]);
          pending[r] = false;
        }
      }
      __syncthreads();
      for (uint32_t p = threadIdx.x; p < NP; p += blockDim.x) {
        if (cnt[p] >= B) {
          const unsigned int slot = atomicAdd(&nflush, 1u);
          flist[slot] = p;
          fbase[slot] = atomicAdd(&cursor[p], (unsigned long long)B);
          cnt[p] = 0;
        }
      }
      __syncthreads();
      const uint32_t units_per_buf = B * R / 2;   // 16-byte units (B is even)
      const uint32_t total_units = nflush * units_per_buf;
      for (uint32_t u = threadIdx.x; u < total_units; u += blockDim.x) {
        const uint32_t slot = u / units_per_buf, off = u - slot * units_per_buf;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(buf + (size_t)flist[slot] * B * R + (size_t)off * 2);
        *reinterpret_cast<ulonglong2*>(out + (size_t)fbase[slot] * R + (size_t)off * 2) = v;
      }
      bool mine = false;
#pragma unroll
      for (int r = 0; r < kRows; r++) mine = mine || pending[r];
      any = __syncthreads_or(mine ? 1 : 0);
    } while (any);
  }
  // partial buffers
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < NP; p += blockDim.x) {
    const unsigned int c = cnt[p] < B ? cnt[p] : B;
    if (!c) continue;
    const unsigned long long base = atomicAdd(&cursor[p], (unsigned long long)c);
    for (uint32_t j = 0; j < c * R; j++) out[(size_t)base * R + j] = buf[(size_t)p * B * R + j];
  }
}

// ---- pass 3: per-partition LDS aggregation ---------------------------------------------------------
struct PartAggParams {
  const unsigned long long* recs;
  const unsigned long long* part_off;   // [NP + 1] record offsets
  unsigned long long* counter;          // [0] groups written so far
  unsigned int* overflow;               // [0] an LDS table filled up
  unsigned long long* out_keys;
  unsigned char* out_kvalid;
  unsigned long long* out_acc;
  uint32_t log2_slots;
  uint32_t max_groups;                  // capacity of the output arrays
};
constexpr int kAggBlock = 512;

__global__ __launch_bounds__(kAggBlock) void part_agg_kernel(Shape sh, PartitionPlan pp, PartAggParams ap) {
  extern __shared__ unsigned long long lds_raw[];
  const uint32_t S = 1u << ap.log2_slots, n_aggs = sh.n_aggs, R = pp.rec_words, NP = 1u << pp.log2_parts;
  unsigned long long* keys = lds_raw;                 // [S + 2]: slot S = null-key group, S + 1 = the key equal to EMPTY
  unsigned long long* cells = keys + S + 2;           // [(S + 2) * n_aggs]
  __shared__ unsigned int n_occ, cursor_l, full;
  __shared__ unsigned long long gbase;
  for (uint32_t p = blockIdx.x; p < NP; p += gridDim.x) {
    for (uint32_t i = threadIdx.x; i < S + 2; i += blockDim.x) keys[i] = kEmptyKey;
    for (uint32_t i = threadIdx.x; i < (S + 2) * n_aggs; i += blockDim.x) cells[i] = agg_identity_dev(sh.aggs[i % n_aggs].kind);
    if (threadIdx.x == 0) { n_occ = 0; cursor_l = 0; full = 0; }
    __syncthreads();
    const uint64_t beg = ap.part_off[p], end = ap.part_off[p + 1];
    for (uint64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
      const unsigned long long* rec = ap.recs + i * R;
      const uint64_t key = rec[0];
      const uint64_t vbits = pp.has_valid ? rec[1 + pp.n_src] : ~0ull;
      const uint64_t rowid = pp.has_rowid ? rec[1 + pp.n_src + (pp.has_valid ? 1 : 0)] : 0ull;
      uint32_t slot;
      if (!(vbits >> 63)) { slot = S; keys[S] = 0; }
      else if (key == kEmptyKey) { slot = S + 1; keys[S + 1] = 0; }
      else {
        slot = (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> (64 - ap.log2_slots));   // a second hash: the partition consumed the top bits of the first
        uint32_t probe = 0;
        for (;; probe++) {
          const unsigned long long cur = keys[slot];
          if (cur == key) break;
          if (cur == kEmptyKey) {
            const unsigned long long old = atomicCAS(&keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey || old == key) break;
          }
          slot = (slot + 1) & (S - 1);
          if (probe >= S) { full = 1; break; }
        }
        if (probe >= S) continue;
      }
      unsigned long long* cell = cells + (size_t)slot * n_aggs;
      for (uint32_t k = 0; k < n_aggs; k++) {
        const uint8_t kind = sh.aggs[k].kind;
        const uint8_t sj = pp.agg_src[k];
        const uint64_t v = sj != kNone ? rec[1 + sj] : 0ull;
        const bool valid = sj != kNone ? ((vbits >> sj) & 1) : true;
        const uint64_t x = agg_row_value(kind, v, true, valid, rowid);
        if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) {
          if (kind == AGG_SUM_F && !valid) continue;
          lds_atomic_agg(kind, cell + k, x);
        }
      }
    }
    __syncthreads();
    if (full) { if (threadIdx.x == 0) atomicExch(ap.overflow, 1u); __syncthreads(); continue; }
    // emit the partition's groups: count, reserve once, write
    uint32_t mine = 0;
    for (uint32_t s = threadIdx.x; s < S + 2; s += blockDim.x) mine += keys[s] != kEmptyKey;
    if (mine) atomicAdd(&n_occ, mine);
    __syncthreads();
    if (threadIdx.x == 0) gbase = n_occ ? atomicAdd(ap.counter, (unsigned long long)n_occ) : 0ull;
    __syncthreads();
    if (gbase + n_occ > ap.max_groups) { if (threadIdx.x == 0) atomicExch(ap.overflow, 2u); __syncthreads(); continue; }
    for (uint32_t s = threadIdx.x; s < S + 2; s += blockDim.x) {
      if (keys[s] == kEmptyKey) continue;
      const uint64_t o = gbase + atomicAdd(&cursor_l, 1u);
      ap.out_keys[o] = s < S ? keys[s] : (s == S ? 0ull : kEmptyKey);
      ap.out_kvalid[o] = s == S ? 0 : 1;
      for (uint32_t k = 0; k < n_aggs; k++) ap.out_acc[o * n_aggs + k] = cells[(size_t)s * n_aggs + k];
    }
    __syncthreads();
  }
}

// ---- host side -----------------------------------------------------------------------------------------
static uint64_t scan_bytes(const Shape& sh, const Args& args) {
  uint64_t b = 0;
  for (int i = 0; i < sh.n_inputs; i++) b += (uint64_t)args.n_rows * dtype_width(sh.in_dtype[i]) + (args.in[i].validity ? (uint64_t)args.n_rows / 8 : 0);
  return b;
}

bool partition_plan(const Shape& sh, double est_groups, bool any_nullable, PartitionPlan* out) {
  PartitionPlan pp{};
  // distinct aggregate sources
  for (int k = 0; k < kMaxAggs; k++) pp.agg_src[k] = kNone;
  for (int k = 0; k < sh.n_aggs; k++) {
    const uint8_t kind = sh.aggs[k].kind;
    if (kind == AGG_LEN) continue;
    if (kind == AGG_FIRST_ROW) { pp.has_rowid = 1; continue; }
    int j = -1;
    for (uint32_t t = 0; t < pp.n_src; t++) if (pp.src_slot[t] == sh.aggs[k].src) j = (int)t;
    if (j < 0) { j = (int)pp.n_src; pp.src_slot[pp.n_src++] = sh.aggs[k].src; }
    pp.agg_src[k] = (uint8_t)j;
  }
  if (pp.n_src > 8) return false;   // kMaxSrc
  pp.has_valid = any_nullable ? 1 : 0;
  pp.rec_words = 1 + pp.n_src + pp.has_valid + pp.has_rowid;
  // LDS table of pass 3: as many slots as fit ~144 KB; partitions so that a partition holds <= slots / 2 groups
  const size_t lds_budget = 144 * 1024;
  uint32_t log2_slots = 14;
  while (log2_slots > 8 && ((size_t)(1u << log2_slots) + 2) * 8 * (1 + sh.n_aggs) > lds_budget) log2_slots--;
  if (((size_t)(1u << log2_slots) + 2) * 8 * (1 + sh.n_aggs) > lds_budget) return false;
  const double per_part = (double)(1u << log2_slots) * 0.45;
  uint32_t log2_parts = 6;
  while (log2_parts < 10 && (double)(1u << log2_parts) * per_part < est_groups) log2_parts++;
  if ((double)(1u << log2_parts) * per_part < est_groups) return false;   // would need > 1024 partitions
  pp.log2_parts = log2_parts;
  pp.log2_slots = log2_slots;
  // write-combining buffers of pass 2
  uint32_t B = 8;
  auto scatter_lds = [&](uint32_t b) { return ((size_t)(1u << log2_parts) * b * pp.rec_words + (1u << log2_parts)) * 8 + (size_t)(1u << log2_parts) * 8 + 16; };
  while (B > 2 && scatter_lds(B) > lds_budget) B -= 2;
  if (scatter_lds(B) > lds_budget) return false;
  pp.buf_rows = B;
  *out = pp;
  return true;
}

#ifdef PLX_HAVE_Q3_SHAPES
#define PLX_PART_STATIC_CASES(KERNEL, ...)                                                                                     \
  case SHAPE_GB_SUM_CNT_I64: hipLaunchKernelGGL((KERNEL<StatProg<SHAPE_GB_SUM_CNT_I64>>), __VA_ARGS__); break;                 \
  case SHAPE_GB_SUM_MEAN_U32_F64: hipLaunchKernelGGL((KERNEL<StatProg<SHAPE_GB_SUM_MEAN_U32_F64>>), __VA_ARGS__); break;
#else
#define PLX_PART_STATIC_CASES(KERNEL, ...)
#endif

// Runs the three passes.  Outputs (allocated here): dense packed keys / valid flags / cells.  Returns the number of
// groups, or -1 if an LDS table overflowed (the caller falls back to the HBM-table sink).
int64_t partitioned_agg(const Shape& sh, const Args& args, const PartitionPlan& pp, int static_id, Buf* out_keys, Buf* out_kvalid, Buf* out_acc, std::string* desc) {
  const uint32_t NP = 1u << pp.log2_parts;
  const int grid = grid_for((args.n_rows + kTileRows - 1) / kTileRows, kBlock / 64, 8);
  Buf hist = dev_alloc_zero(sizeof(uint64_t) * (NP + 1));
  Buf part_off = dev_alloc(sizeof(uint64_t) * (NP + 2));
  {
    ProfileScope ps("part_count", scan_bytes(sh, args), (uint64_t)args.n_rows);
    PartCountSink::Params cp{hist->as<unsigned long long>(), pp.log2_parts};
    const size_t lds = sizeof(unsigned int) * NP;
    switch (static_id) {
      PLX_PART_STATIC_CASES(part_count_kernel, dim3(grid), dim3(kBlock), lds, stream(), sh, args, cp)
      default: hipLaunchKernelGGL((part_count_kernel<DynProg>), dim3(grid), dim3(kBlock), lds, stream(), sh, args, cp); break;
    }
    PLX_HIP(hipGetLastError());
  }
  exclusive_scan_u64(hist->as<uint64_t>(), part_off->as<uint64_t>(), NP);   // writes NP + 1 entries
  uint64_t total = 0;
  d2h_sync(&total, part_off->as<uint64_t>() + NP, 8);
  if (total == 0) { *out_keys = dev_alloc(8); *out_kvalid = dev_alloc(8); *out_acc = dev_alloc(8); return 0; }
  Buf recs = dev_alloc(sizeof(uint64_t) * (size_t)total * pp.rec_words + 64);
  Buf cursor = dev_alloc(sizeof(uint64_t) * NP);
  PLX_HIP(hipMemcpyAsync(cursor->ptr, part_off->ptr, sizeof(uint64_t) * NP, hipMemcpyDeviceToDevice, stream()));
  {
    ProfileScope ps("part_scatter", scan_bytes(sh, args) + total * pp.rec_words * 8, (uint64_t)args.n_rows);
    const size_t lds = ((size_t)NP * pp.buf_rows * pp.rec_words + NP) * 8 + (size_t)NP * 8 + 16;
    const int64_t nbt = (args.n_rows + (int64_t)kBlock * kRows - 1) / ((int64_t)kBlock * kRows);
    const int sgrid = (int)std::min<int64_t>(nbt, (int64_t)device().cu_count * (lds > 72 * 1024 ? 1 : 2));
    switch (static_id) {
      PLX_PART_STATIC_CASES(part_scatter_kernel, dim3(sgrid), dim3(kBlock), lds, stream(), sh, args, pp, cursor->as<unsigned long long>(), recs->as<unsigned long long>())
      default: hipLaunchKernelGGL((part_scatter_kernel<DynProg>), dim3(sgrid), dim3(kBlock), lds, stream(), sh, args, pp, cursor->as<unsigned long long>(), recs->as<unsigned long long>()); break;
    }
    PLX_HIP(hipGetLastError());
  }
  const uint64_t max_groups = std::min<uint64_t>(total, (uint64_t)NP * ((1ull << pp.log2_slots) + 2));
  *out_keys = dev_alloc(sizeof(uint64_t) * max_groups);
  *out_kvalid = dev_alloc(max_groups);
  *out_acc = dev_alloc(sizeof(uint64_t) * max_groups * sh.n_aggs);
  Buf ctr = dev_alloc_zero(16);
  PartAggParams ap{};
  ap.recs = recs->as<unsigned long long>(); ap.part_off = part_off->as<unsigned long long>(); ap.counter = ctr->as<unsigned long long>();
  ap.overflow = reinterpret_cast<unsigned int*>(ctr->as<unsigned long long>() + 1);
  ap.out_keys = (*out_keys)->as<unsigned long long>(); ap.out_kvalid = (*out_kvalid)->as<unsigned char>(); ap.out_acc = (*out_acc)->as<unsigned long long>();
  ap.log2_slots = pp.log2_slots; ap.max_groups = (uint32_t)std::min<uint64_t>(max_groups, 0xffffffffull);
  {
    ProfileScope ps("part_agg_lds", total * pp.rec_words * 8, total);
    const size_t lds = (((size_t)1 << pp.log2_slots) + 2) * 8 * (1 + sh.n_aggs);
    const int agrid = (int)std::min<uint32_t>(NP, (uint32_t)device().cu_count);
    hipLaunchKernelGGL(part_agg_kernel, dim3(agrid), dim3(kAggBlock), lds, stream(), sh, pp, ap);
    PLX_HIP(hipGetLastError());
  }
  uint64_t res[2] = {0, 0};
  d2h_sync(res, ctr->ptr, 16);
  if ((uint32_t)res[1]) return -1;
  if (desc) *desc = "partitioned(P=" + std::to_string(NP) + ",rec=" + std::to_string(pp.rec_words * 8) + "B,buf=" + std::to_string(pp.buf_rows) + ")+lds_hash_table(slots=" + std::to_string(1u << pp.log2_slots) + ")";
  return (int64_t)res[0];
}

}  // namespace k
}  // namespace plx
