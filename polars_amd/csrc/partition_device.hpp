// partition_device.hpp -- device side of the partitioned high-cardinality group-by (see kernels_partition.hip for the
// design).  Device-only header: compiled ahead of time by kernels_partition.hip (AOT shapes + generic interpreter) and
// at run time by hiprtc (jit.cpp) for any other program shape.
#pragma once
#include "fused_device.hpp"

namespace plx {
namespace k {

__device__ __forceinline__ uint32_t part_of(uint64_t key, bool kvalid, uint32_t log2_parts) {
  if (!kvalid) return 0;   // null_partition() == 0 (hashing.rs:111-115)
  return (uint32_t)((key * 0x55fbfd6bfc5458e9ull) >> (64 - log2_parts));
}

// ---- round structure shared by pass 1 and pass 2 ---------------------------------------------------
// Workgroup-synchronous: round rd covers kRoundTiles tiles per wave (kBlock * kRows * kRoundTiles rows); workgroup b
// handles rounds b, b + grid, ...  Both passes use the SAME grid, so pass 1's per-workgroup histogram tells pass 2
// exactly where each workgroup writes each partition: no atomics on the scatter path, deterministic output.
// AOT programs run 4 tiles per round (their register files are small); the generic interpreter's dynamically
// indexed register file is large, so it runs 1 tile per round.
constexpr int kStaticRoundTiles = 4;
template <class P> struct Round { static constexpr int kTiles = P::kStatic ? kStaticRoundTiles : 1; static constexpr int kRowsPerLane = kRows * kTiles; };

// Evaluates the program for the kRoundTiles tiles of round `rd` owned by this wave.  When the whole round lies inside
// the input (wave-uniform test) the tiles run back to back in straight-line code, so the compiler can issue the column
// loads of all tiles before the first use; the tail round takes the bounds-checked path.
template <class P, class RF>
__device__ __forceinline__ void round_rows(const Shape& dsh, const Args& args, int64_t rd, int wave_in_block, RF rf[Round<P>::kTiles], bool pass[Round<P>::kTiles][kRows]) {
  constexpr int kRoundTiles = Round<P>::kTiles;
  const int lane = lane_id();
  const int64_t first_tile = rd * kRoundTiles * (kBlock / 64);
  const bool all_full = (first_tile + (int64_t)kRoundTiles * (kBlock / 64)) * kTileRows <= args.n_rows;
  uint8_t pred;
  if constexpr (P::kStatic) { constexpr Shape sh = P::shape(); pred = sh.pred; } else pred = dsh.pred;
  if (all_full) {
#pragma unroll
    for (int t = 0; t < kRoundTiles; t++) {
      const int64_t row0 = (first_tile + (int64_t)t * (kBlock / 64) + wave_in_block) * kTileRows + (int64_t)lane * kRows;
      run_program<P, true>(dsh, args, row0, rf[t]);
    }
#pragma unroll
    for (int t = 0; t < kRoundTiles; t++) {
#pragma unroll
      for (int r = 0; r < kRows; r++) pass[t][r] = pred == kNone || ((rf[t].get(r, pred) & 1) && ((rf[t].getv(pred) >> r) & 1));
    }
  } else {
#pragma unroll
    for (int t = 0; t < kRoundTiles; t++) { int64_t row0; tile_rows<P>(dsh, args, first_tile + (int64_t)t * (kBlock / 64) + wave_in_block, rf[t], pass[t], row0); }
  }
}
template <class P>
__device__ __forceinline__ bool round_is_full(const Args& args, int64_t rd) {
  return (rd + 1) * Round<P>::kTiles * (kBlock / 64) * (int64_t)kTileRows <= args.n_rows;
}
template <class P>
__device__ __forceinline__ int64_t round_row0(int64_t rd, int t, int wave_in_block) {
  return ((rd * Round<P>::kTiles + t) * (kBlock / 64) + wave_in_block) * (int64_t)kTileRows + (int64_t)lane_id() * kRows;
}

// ---- pass 1: per-workgroup partition histogram -----------------------------------------------------
template <class P>
__device__ __forceinline__ void part_count_body(const Shape dsh, const Args args, uint32_t log2_parts, unsigned int* __restrict__ hist /* [grid][NP] */) {
  extern __shared__ unsigned long long lds_raw[];
  unsigned int* cnt = reinterpret_cast<unsigned int*>(lds_raw);
  const uint32_t NP = 1u << log2_parts;
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  constexpr int kRoundTiles = Round<P>::kTiles;
  const int64_t rows_per_round = (int64_t)kBlock * Round<P>::kRowsPerLane;
  const int64_t nrounds = (args.n_rows + rows_per_round - 1) / rows_per_round;
  const int wave_in_block = threadIdx.x >> 6;
  uint8_t key_slot;
  if constexpr (P::kStatic) { constexpr Shape sh = P::shape(); key_slot = sh.key; } else key_slot = dsh.key;
  for (int64_t rd = blockIdx.x; rd < nrounds; rd += gridDim.x) {
    typename RegFileOf<P>::type rf[kRoundTiles] = {make_regfile<P>(args)}; bool pass[kRoundTiles][kRows];
    round_rows<P>(dsh, args, rd, wave_in_block, rf, pass);
#pragma unroll
    for (int t = 0; t < kRoundTiles; t++) {
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        if (!pass[t][r]) continue;
        atomicAdd(&cnt[part_of(rf[t].get(r, key_slot), (rf[t].getv(key_slot) >> r) & 1, log2_parts)], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) hist[(size_t)blockIdx.x * NP + i] = cnt[i];
}

// hist[grid][NP] -> wg_prefix[grid][NP] (records of partition p written by workgroups < b) and total[NP]
__global__ __launch_bounds__(kBlock) void part_prefix_kernel(const unsigned int* __restrict__ hist, int grid, uint32_t NP, unsigned long long* __restrict__ wg_prefix,
                                                             unsigned long long* __restrict__ total) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NP) return;
  unsigned long long run = 0;
  for (int b = 0; b < grid; b++) { wg_prefix[(size_t)b * NP + p] = run; run += hist[(size_t)b * NP + p]; }
  total[p] = run;
}

// ---- pass 2: scatter through LDS write-combining buffers ------------------------------------------
// Workgroup-synchronous tile loop (every wave of the workgroup runs the same number of iterations, so
// __syncthreads inside the loop is legal -- unlike fused_scan_kernel, whose tiles are handed out per wave).
// Layout provider: compile-time for AOT / JIT programs, the plan's copy for the generic interpreter.
template <class P> struct LayoutOf {
  static constexpr bool kConst = true;
  static constexpr RecLayout kL = rec_layout(P::shape());
  __device__ __forceinline__ static constexpr RecLayout get(const PartitionPlan&) { return kL; }
};
template <> struct LayoutOf<DynProg> {
  static constexpr bool kConst = false;
  __device__ __forceinline__ static const RecLayout& get(const PartitionPlan& pp) { return pp.rec; }
};

struct Rec {
  uint64_t key, vbits, rowid;
  uint64_t src[kMaxSrc];
};
// Register-resident record (all indices compile-time: a dynamically indexed array would live in scratch).
template <class S, class RF>
__device__ __forceinline__ void make_record(const S& sh, const RecLayout& L, const RF& rf, int r, int64_t row, Rec& rec) {
  const bool kvalid = (rf.getv(sh.key) >> r) & 1;
  rec.key = kvalid ? rf.get(r, sh.key) : 0ull;
  rec.vbits = kvalid ? (1ull << 63) : 0ull;
  rec.rowid = (uint64_t)row;
#pragma unroll
  for (int j = 0; j < kMaxSrc; j++) {
    rec.src[j] = 0;
    if (j < (int)L.n_src) {
      rec.src[j] = rf.get(r, L.src_slot[j]);
      if ((rf.getv(L.src_slot[j]) >> r) & 1) rec.vbits |= 1ull << j;
    }
  }
}
__device__ __forceinline__ void store_record(unsigned long long* dst, const RecLayout& L, const Rec& rec) {
  dst[0] = rec.key;
#pragma unroll
  for (int j = 0; j < kMaxSrc; j++) if (j < (int)L.n_src) dst[1 + j] = rec.src[j];
  uint32_t w = 1 + L.n_src;
  if (L.has_valid) dst[w++] = rec.vbits;
  if (L.has_rowid) dst[w] = rec.rowid;
}

template <class P>
__device__ __forceinline__ void part_scatter_body(const Shape dsh, const Args args, const PartitionPlan pp, const unsigned long long* __restrict__ part_off,
                                                              const unsigned long long* __restrict__ wg_prefix, unsigned long long* __restrict__ out) {
  extern __shared__ unsigned long long lds_raw[];
  const RecLayout L = LayoutOf<P>::get(pp);   // compile-time constant for AOT programs
  const uint32_t NP = 1u << pp.log2_parts, B = pp.buf_rows, R = L.rec_words;
  unsigned long long* buf = lds_raw;                                        // [NP][B][R]
  unsigned long long* fbase = buf + (size_t)NP * B * R;                     // [NP] global record index of a flush
  unsigned long long* cur = fbase + NP;                                     // [NP] this workgroup's next record index per partition
  unsigned int* cnt = reinterpret_cast<unsigned int*>(cur + NP);            // [NP]
  unsigned int* flist = cnt + NP;                                           // [NP]
  unsigned int& nflush = flist[NP];                                         // kept in the dynamic region: a static __shared__ in front of it
                                                                            // would break the 16-byte alignment the ulonglong2 copies need
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) { cnt[i] = 0; cur[i] = part_off[i] + wg_prefix[(size_t)blockIdx.x * NP + i]; }
  __syncthreads();
  // One barrier round covers kRoundTiles tiles per wave: the loads of all tiles are independent, so several are in
  // flight per lane while the round's appends/flushes run once.
  constexpr int kRoundTiles = Round<P>::kTiles;
  constexpr int kRoundRows = Round<P>::kRowsPerLane;
  const int64_t rows_per_round = (int64_t)kBlock * kRoundRows;
  const int64_t nrounds = (args.n_rows + rows_per_round - 1) / rows_per_round;
  const int wave_in_block = threadIdx.x >> 6;
  // Software pipeline (AOT programs): the column loads of round r+1 are ISSUED before round r's records are appended
  // and flushed (a few microseconds of LDS work and barriers -- barriers do not drain VMEM), and only consumed
  // afterwards.  One register file and one record set: no copies.
  typename RegFileOf<P>::type rf[kRoundTiles] = {make_regfile<P>(args)};
  Rec rec[kRoundRows];
  uint32_t part[kRoundRows];
  bool pending[kRoundRows];
  auto finish_round = [&](int64_t rd, bool preloaded) {
    bool pass[kRoundTiles][kRows];
    bool done = false;
    if constexpr (P::kStatic) {
      if (preloaded) {
        constexpr Shape psh = P::shape();
#pragma unroll
        for (int t = 0; t < kRoundTiles; t++) {
          run_rest_full<P>(args, round_row0<P>(rd, t, wave_in_block), rf[t]);
#pragma unroll
          for (int r = 0; r < kRows; r++) pass[t][r] = psh.pred == kNone || ((rf[t].get(r, psh.pred) & 1) && ((rf[t].getv(psh.pred) >> r) & 1));
        }
        done = true;
      }
    }
    if (!done) round_rows<P>(dsh, args, rd, wave_in_block, rf, pass);
#pragma unroll
    for (int t = 0; t < kRoundTiles; t++) {
      const int64_t row0 = round_row0<P>(rd, t, wave_in_block);
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        const int q = t * kRows + r;
        pending[q] = pass[t][r];
        if constexpr (P::kStatic) { constexpr Shape sh = P::shape(); make_record(sh, L, rf[t], r, row0 + r, rec[q]); }
        else make_record(dsh, L, rf[t], r, row0 + r, rec[q]);
        part[q] = part_of(rec[q].key, (rec[q].vbits >> 63) & 1, pp.log2_parts);
      }
    }
  };
  auto issue_loads = [&](int64_t rd) -> bool {   // true if the loads of round rd are now in flight
    if constexpr (P::kStatic) {
      if (rd < nrounds && round_is_full<P>(args, rd)) {
#pragma unroll
        for (int t = 0; t < kRoundTiles; t++) run_loads_full<P>(args, round_row0<P>(rd, t, wave_in_block), rf[t]);
        return true;
      }
    }
    return false;
  };
  if ((int64_t)blockIdx.x < nrounds) finish_round(blockIdx.x, issue_loads(blockIdx.x));
  for (int64_t rd = blockIdx.x; rd < nrounds; rd += gridDim.x) {
    // records of round rd are in rec[]: they are copied out of the register file, which is free for the next round
    const int64_t rd_next = rd + gridDim.x;
    const bool preloaded = issue_loads(rd_next);
    int any;
    do {
      if (threadIdx.x == 0) nflush = 0;
#pragma unroll
      for (int q = 0; q < kRoundRows; q++) {
        if (!pending[q]) continue;
        const unsigned int pos = atomicAdd(&cnt[part[q]], 1u);
        if (pos < B) {
          store_record(buf + ((size_t)part[q] * B + pos) * R, L, rec[q]);
          pending[q] = false;
        }
      }
      __syncthreads();
      for (uint32_t p = threadIdx.x; p < NP; p += blockDim.x) {
        if (cnt[p] >= B) {
          const unsigned int slot = atomicAdd(&nflush, 1u);
          flist[slot] = p;
          fbase[slot] = cur[p];      // only this thread touches cur[p] in this phase
          cur[p] += B;
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
      for (int q = 0; q < kRoundRows; q++) mine = mine || pending[q];
      any = __syncthreads_or(mine ? 1 : 0);
    } while (any);
    if (rd_next < nrounds) finish_round(rd_next, preloaded);
  }
  // partial buffers
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < NP; p += blockDim.x) {
    const unsigned int c = cnt[p] < B ? cnt[p] : B;
    if (!c) continue;
    const unsigned long long base = cur[p];
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
constexpr int kAggBlock = 1024;

// Aggregate kinds and record layout are compile-time constants for AOT programs (the per-record agg loop is then
// straight-line code; with run-time kinds it is a chain of scalar switches and the kernel is instruction-bound).
template <class S>
__device__ __forceinline__ void part_agg_body(const S& sh, const RecLayout& L, const PartitionPlan& pp, const PartAggParams& ap) {
  extern __shared__ unsigned long long lds_raw[];
  const uint32_t NS = 1u << ap.log2_slots, n_aggs = sh.n_aggs, R = L.rec_words, NP = 1u << pp.log2_parts;
  unsigned long long* keys = lds_raw;                 // [NS + 2]: slot NS = null-key group, NS + 1 = the key equal to EMPTY
  unsigned long long* cells = keys + NS + 2;           // [(NS + 2) * n_aggs]
  __shared__ unsigned int n_occ, cursor_l, full;
  __shared__ unsigned long long gbase;
  for (uint32_t p = blockIdx.x; p < NP; p += gridDim.x) {
    for (uint32_t i = threadIdx.x; i < NS + 2; i += blockDim.x) keys[i] = kEmptyKey;
    for (uint32_t i = threadIdx.x; i < (NS + 2) * n_aggs; i += blockDim.x) cells[i] = agg_identity_dev(sh.aggs[i % n_aggs].kind);
    if (threadIdx.x == 0) { n_occ = 0; cursor_l = 0; full = 0; }
    __syncthreads();
    const uint64_t beg = ap.part_off[p], end = ap.part_off[p + 1];
    constexpr int kInFlight = 8;   // records per thread per batch; the next batch is loaded while this one is consumed
    unsigned long long n0[kInFlight], n1[kInFlight];
    auto load_batch = [&](uint64_t i0, unsigned long long* a0, unsigned long long* a1) {
#pragma unroll
      for (int u = 0; u < kInFlight; u++) {
        const uint64_t i = i0 + (uint64_t)u * blockDim.x;
        a0[u] = 0; a1[u] = 0;
        if (i < end) {
          if (R == 2) { const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(ap.recs + i * 2); a0[u] = v.x; a1[u] = v.y; }
          else { a0[u] = ap.recs[i * R]; a1[u] = R > 1 ? ap.recs[i * R + 1] : 0ull; }
        }
      }
    };
    const uint64_t step = (uint64_t)blockDim.x * kInFlight;
    if (beg + threadIdx.x < end) load_batch(beg + threadIdx.x, n0, n1);
    for (uint64_t i0 = beg + threadIdx.x; i0 < end; i0 += step) {
     unsigned long long w0[kInFlight], w1[kInFlight];
#pragma unroll
     for (int u = 0; u < kInFlight; u++) { w0[u] = n0[u]; w1[u] = n1[u]; }
     if (i0 + step < end) load_batch(i0 + step, n0, n1);
#pragma unroll
     for (int u = 0; u < kInFlight; u++) {
      const uint64_t i = i0 + (uint64_t)u * blockDim.x;
      if (i >= end) continue;
      const unsigned long long* rec = ap.recs + i * R;
      const uint64_t key = w0[u];
      const uint64_t vbits = L.has_valid ? rec[1 + L.n_src] : ~0ull;
      const uint64_t rowid = L.has_rowid ? rec[1 + L.n_src + (L.has_valid ? 1 : 0)] : 0ull;
      uint32_t slot;
      if (!(vbits >> 63)) { slot = NS; keys[NS] = 0; }
      else if (key == kEmptyKey) { slot = NS + 1; keys[NS + 1] = 0; }
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
          slot = (slot + 1) & (NS - 1);
          if (probe >= NS) { full = 1; break; }
        }
        if (probe >= NS) continue;
      }
      unsigned long long* cell = cells + (size_t)slot * n_aggs;
#pragma unroll
      for (uint32_t k = 0; k < (uint32_t)kMaxAggs; k++) {
        if (k >= n_aggs) break;
        const uint8_t kind = sh.aggs[k].kind;
        const uint8_t sj = L.agg_src[k];
        const uint64_t v = sj != kNone ? (sj == 0 ? w1[u] : rec[1 + sj]) : 0ull;
        const bool valid = sj != kNone ? ((vbits >> sj) & 1) : true;
        const uint64_t x = agg_row_value(kind, v, true, valid, rowid);
        if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) {
          if (kind == AGG_SUM_F && !valid) continue;
          lds_atomic_agg(kind, cell + k, x);
        }
      }
     }
    }
    __syncthreads();
    if (full) { if (threadIdx.x == 0) atomicExch(ap.overflow, 1u); __syncthreads(); continue; }
    // emit the partition's groups: count, reserve once, write
    uint32_t mine = 0;
    for (uint32_t s = threadIdx.x; s < NS + 2; s += blockDim.x) mine += keys[s] != kEmptyKey;
    if (mine) atomicAdd(&n_occ, mine);
    __syncthreads();
    if (threadIdx.x == 0) gbase = n_occ ? atomicAdd(ap.counter, (unsigned long long)n_occ) : 0ull;
    __syncthreads();
    if (gbase + n_occ > ap.max_groups) { if (threadIdx.x == 0) atomicExch(ap.overflow, 2u); __syncthreads(); continue; }
    for (uint32_t s = threadIdx.x; s < NS + 2; s += blockDim.x) {
      if (keys[s] == kEmptyKey) continue;
      const uint64_t o = gbase + atomicAdd(&cursor_l, 1u);
      ap.out_keys[o] = s < NS ? keys[s] : (s == NS ? 0ull : kEmptyKey);
      ap.out_kvalid[o] = s == NS ? 0 : 1;
      for (uint32_t k = 0; k < n_aggs; k++) ap.out_acc[o * n_aggs + k] = cells[(size_t)s * n_aggs + k];
    }
    __syncthreads();
  }
}

template <class P>
__global__ __launch_bounds__(kAggBlock) void part_agg_kernel(Shape dsh, PartitionPlan pp, PartAggParams ap) {
  if constexpr (P::kStatic) { constexpr Shape csh = P::shape(); constexpr RecLayout cl = rec_layout(P::shape()); part_agg_body(csh, cl, pp, ap); }
  else part_agg_body(dsh, pp.rec, pp, ap);
}

template <class P>
__global__ __launch_bounds__(kBlock) void part_count_kernel(Shape dsh, Args args, uint32_t log2_parts, unsigned int* __restrict__ hist) {
  part_count_body<P>(dsh, args, log2_parts, hist);
}
template <class P>
__global__ __launch_bounds__(kBlock) void part_scatter_kernel(Shape dsh, Args args, PartitionPlan pp, const unsigned long long* __restrict__ part_off,
                                                              const unsigned long long* __restrict__ wg_prefix, unsigned long long* __restrict__ out) {
  part_scatter_body<P>(dsh, args, pp, part_off, wg_prefix, out);
}

}  // namespace k
}  // namespace plx
