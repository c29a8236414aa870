// partition2_device.hpp -- device side of the partitioned high-cardinality group-by, second generation (see
// kernels_partition.hip for the pipeline).  Device-only header: compiled ahead of time by kernels_partition.hip for the
// benchmark shapes and at run time by hiprtc (jit.cpp) for any other program shape.
//
//   scatter  one 1024-thread workgroup per CU streams its rounds of the input (the fused program: predicate, key,
//            aggregate sources), turns every surviving row into a packed dword record and appends it to its partition's
//            LDS ring (one returning ds_add_u64 on {limit:fill} per row); after a barrier the waves write every COMPLETE
//            128-B line of every ring to the partition's current chunk (8 lanes x 16 B per line, 8 lines per store
//            instruction: full aligned lines only).  Chunks (256 records) come from the workgroup's PRIVATE region, handed
//            out by an LDS counter: no counting pass, no global atomics, no cross-workgroup coordination.  Rows of hot keys
//            (heavy hitters found in a sample) never reach the rings: they are aggregated in LDS accumulators on the spot.
//   sort     the chunk -> partition map is counting-sorted into per-partition chunk lists (three tiny kernels).
//   agg      one workgroup per partition (launched as a grid of P: the hardware balances them) walks the partition's
//            chunk list, double-buffered, into an LDS open-addressing table (hash mode) or an LDS direct-address table
//            (direct mode: dense packed ids, no key compare at all) and writes its groups straight to the dense output.
#pragma once
#include <type_traits>

#include "fused_device.hpp"

namespace plx {
namespace k {

constexpr int kP2MaxBlock = 1024;
constexpr int kP2AggBlock = 1024;

struct ScatterParams2 {
  unsigned int* recs;                       // record pool: chunk c = dwords [c * chunk_dw, (c + 1) * chunk_dw)
  unsigned int* chunk_part;                 // [scatter_grid * chunks_per_wg] partition of a chunk (kNoChunk: never handed out)
  unsigned int* chunk_fill;                 // records in the chunk
  unsigned int* flags;                      // [0] a workgroup ran out of chunks (cannot happen by construction; checked), [1] a dense id outside its declared range
  const unsigned long long* hot_tbl_keys;   // [1 << log2_hot_slots] open-addressing lookup of the hot keys (kEmptyKey = free)
  const unsigned int* hot_tbl_idx;          // hot-key ordinal of the slot
  unsigned long long* hot_out;              // [n_hot][n_aggs] cells, device-scope atomics at the end of the kernel
  long long* key_minmax;                    // [2] signed min / max of the valid keys the scan saw (by-product statistics; null: not wanted)
};

struct AggParams2 {
  const unsigned int* recs;
  const unsigned int* chunk_fill;
  const unsigned long long* cl_off;         // [P + 1] offsets into cl_ids
  const unsigned int* cl_ids;               // chunk ids grouped by partition
  unsigned long long* counter;              // [0] groups written so far
  unsigned int* overflow;                   // [0] 1: an LDS table filled up, 2: output capacity exceeded
  unsigned long long* out_keys;
  unsigned char* out_kvalid;
  unsigned long long* out_acc;
  uint32_t max_groups;
};

__device__ __forceinline__ uint32_t part2_of(uint64_t key, bool kvalid, uint32_t log2_parts) {
  if (!kvalid) return 0;   // null_partition() == 0 (hashing.rs:111-115)
  return (uint32_t)((key * 0x55fbfd6bfc5458e9ull) >> (64 - log2_parts));
}

// LDS layout of the scatter kernel (dynamic shared memory), in this order:
//   ring    [P][ring_dw] u32        staging rings (ring_dw = ring_lines * 32)
//   fl      [P] u64                 {limit : fill}: records appended to the current chunk so far (low), most that fit (high)
//   hot_k   [hot_slots] u64, hot_acc [n_hot * n_aggs * copies] u64
//   fdw     [P] u32                 dwords of the current chunk already written to HBM
//   chunk   [P] u32                 current chunk (kNoChunk: none yet)
//   hot_i   [hot_slots] u32
//   misc    [4] u32                 [0] next chunk of this workgroup's region, [1], [2] "some row is still pending" flags (alternating)
__host__ __device__ inline size_t part2_scatter_lds(uint32_t P, uint32_t ring_lines, uint32_t hot_slots, uint32_t n_hot, uint32_t n_aggs, uint32_t copies) {
  return (size_t)P * ring_lines * 128 + (size_t)P * 8 + (size_t)hot_slots * 8 + (size_t)n_hot * n_aggs * copies * 8 + (size_t)P * 4 * 2 + (size_t)hot_slots * 4 + 16;
}

// hash of a wide (multi-column) key: its 64-bit words and the null mask of its columns (the reference row-encodes such keys and hashes the bytes,
// crates/polars-row/src/encode.rs, crates/polars-expr/src/hash_keys.rs:334 RowEncodedKeys); the top bits pick the partition, lower bits the LDS slot
__device__ __forceinline__ uint64_t wide_key_mix(uint64_t h, uint64_t w) { h ^= w; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; return h; }
__device__ __forceinline__ uint64_t wide_key_hash(const uint64_t* w, uint32_t n_words, uint32_t nullmask) {
  uint64_t h = 0x9e3779b97f4a7c15ull ^ nullmask;
  for (uint32_t j = 0; j < n_words; j++) h = wide_key_mix(h, w[j]);
  return h * 0x55fbfd6bfc5458e9ull;
}

// ---- one row -> record dwords ---------------------------------------------------------------------------------------
template <int MODE, bool CHECK = true, class S, class RF>
__device__ __forceinline__ void make_record2(const S& sh, const RecLayout2& L, const PartPlan2& pp, const RF& rf, int r, int64_t row, unsigned int* rec /* [L.rec_words] */,
                                             uint32_t& part, bool& kvalid, uint64_t& key64, bool row_ok /* the row passed the predicate (and exists) */, uint64_t& narrow_viol /* wave-uniform: |= ballot(a valid value did not fit its narrowed field); only under pp.check_src */) {
  if (L.n_key_cols) {
    // wide key (hash partitions): one 64-bit word per key column (0 for a null), the columns' null mask folded into the hash and kept in the validity dword
    uint64_t w[kMaxKeys];
    uint32_t nullmask = 0;
#pragma unroll
    for (int j = 0; j < kMaxKeys; j++) {
      w[j] = 0;
      if (j < (int)L.n_key_cols) {
        const bool kv = (rf.getv(sh.keys[j]) >> r) & 1;
        w[j] = kv ? rf.get(r, sh.keys[j]) : 0ull;
        if (!kv) nullmask |= 1u << j;
        rec[2 * j] = (uint32_t)w[j]; rec[2 * j + 1] = (uint32_t)(w[j] >> 32);
      }
    }
    const uint64_t h = wide_key_hash(w, L.n_key_cols, nullmask);
    part = (uint32_t)(h >> (64 - pp.log2_parts));
    kvalid = true; key64 = 0;                        // (a null key column makes a group of its own; no hot keys / key statistics on this path)
    uint32_t vb = (nullmask ^ ((1u << L.n_key_cols) - 1u)) << 24;
#pragma unroll
    for (int j = 0; j < kMaxSrc; j++) {
      if (j < (int)L.n_src) {
        const uint64_t v = rf.get(r, L.src_slot[j]);
        if (L.src_kind[j] == 3) { rec[L.src_off[j]] = (uint32_t)(v - (uint64_t)pp.src_base[j]); if (CHECK && pp.check_src) narrow_viol |= __ballot(((v - (uint64_t)pp.src_base[j]) >> 32) != 0 && row_ok && ((rf.getv(L.src_slot[j]) >> r) & 1)); }
        else {
          rec[L.src_off[j]] = (uint32_t)v;
          if (!L.src_kind[j]) rec[L.src_off[j] + 1] = (uint32_t)(v >> 32);
        }
        if ((rf.getv(L.src_slot[j]) >> r) & 1) vb |= 1u << j;
      }
    }
    if (L.has_valid) rec[L.valid_off] = vb;
    if (L.has_rowid) { rec[L.rowid_off] = (uint32_t)(uint64_t)row; rec[L.rowid_off + 1] = (uint32_t)((uint64_t)row >> 32); }
    return;
  }
  kvalid = (rf.getv(sh.key) >> r) & 1;
  key64 = kvalid ? rf.get(r, sh.key) : 0ull;
  if constexpr (MODE == (int)kP2Direct) {
    // (wave-uniform branch) keys without a usable range -- the hashed partitioned probe: partition and record carry bits of the key's hash instead
    const uint64_t id = pp.hash_bits ? (key64 * kP2HashMult) >> (64u - pp.hash_bits) : key64 - (uint64_t)pp.key_base;
    uint64_t hi = id >> pp.key_shift, low = id;
    if (pp.interleave) {                                               // (wave-uniform) partition = low bits, slot = the bits above them
      hi = (id >> (pp.key_shift + pp.log2_parts)) ? 0xfffffffeull : (id & ((1ull << pp.log2_parts) - 1ull));
      low = id >> pp.log2_parts;
    } else if (pp.slice) {                                             // (wave-uniform) equal slices of the id range: id / slice by the reciprocal, one correction step
      hi = __umul64hi(id, pp.slice_magic);
      low = id - hi * (uint64_t)pp.slice;
      if (low >= (uint64_t)pp.slice) { hi++; low -= (uint64_t)pp.slice; }
    }
    part = hi > 0xfffffffeull ? 0xfffffffeu : (uint32_t)hi;           // far outside the id range: still "outside" after the narrowing
    rec[0] = (uint32_t)low & ((1u << pp.key_shift) - 1u);
  } else {
    part = part2_of(key64, kvalid, pp.log2_parts);
    rec[0] = (uint32_t)key64;
    if (L.key_words == 2) rec[1] = (uint32_t)(key64 >> 32);
  }
  uint32_t vbits = kvalid ? (1u << 31) : 0u;
#pragma unroll
  for (int j = 0; j < kMaxSrc; j++) {
    if (j < (int)L.n_src) {
      const uint64_t v = rf.get(r, L.src_slot[j]);
      // (pp.check_src: the bases come from bounds nobody has verified -- the planner's sample, engine.cpp assume_range: a value outside them is reported, never truncated silently)
      if (L.pack == kPackFused) { rec[0] |= (uint32_t)(v - (uint64_t)pp.src_base[0]) << pp.key_shift; if (CHECK && pp.check_src) narrow_viol |= __ballot(((v - (uint64_t)pp.src_base[0]) >> (32u - pp.key_shift)) != 0 && row_ok && ((rf.getv(L.src_slot[j]) >> r) & 1)); }      // one dword: key_low | (v - base) << key_shift
      else if (L.src_kind[j] == 3) { rec[L.src_off[j]] = (uint32_t)(v - (uint64_t)pp.src_base[j]); if (CHECK && pp.check_src) narrow_viol |= __ballot(((v - (uint64_t)pp.src_base[j]) >> 32) != 0 && row_ok && ((rf.getv(L.src_slot[j]) >> r) & 1)); }
      else {
        rec[L.src_off[j]] = (uint32_t)v;
        if (!L.src_kind[j]) rec[L.src_off[j] + 1] = (uint32_t)(v >> 32);
      }
      if ((rf.getv(L.src_slot[j]) >> r) & 1) vbits |= 1u << j;
    }
  }
  if (L.pack == kPackRowid) {              // {key_low | row << key_shift} as one 64-bit field
    const uint64_t f = (uint64_t)rec[0] | ((uint64_t)row << pp.key_shift);
    rec[0] = (uint32_t)f; rec[1] = (uint32_t)(f >> 32);
    return;
  }
  if (L.has_valid) rec[L.valid_off] = vbits;
  if (L.has_rowid) { rec[L.rowid_off] = (uint32_t)(uint64_t)row; rec[L.rowid_off + 1] = (uint32_t)((uint64_t)row >> 32); }
}

// ---- the scatter kernel ------------------------------------------------------------------------------------------------
// compile-time loop over the tiles of a round (their register files and record stashes must be indexed by constants)
template <int N, class F>
__device__ __forceinline__ void p2_static_for(F&& f) {
  if constexpr (N > 0) { p2_static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// TILES: tiles (kTileRows rows) each wave handles per round.  A round costs two barriers and one scan of the rings whatever it
// appends (measured: the pass is bound by instruction issue and barrier latency, not by bytes -- without ring writes AND without
// line stores it still takes 80 % of its time), so a round is made as wide as the rings allow: all TILES tiles are evaluated into
// records at once (one wait for their loads), the loads of the next round are issued at once, then everything is appended and
// the complete lines are flushed.  The host picks TILES / ring size / partition count together (partition_plan2).
template <class P, int MODE, int TILES = 1>
__device__ __forceinline__ void part2_scatter_body(const Shape dsh, const Args args, const PartPlan2 pp, const ScatterParams2 sp) {
  static_assert(P::kStatic, "the partitioned group-by runs specialised programs only (AOT or JIT)");
  extern __shared__ __attribute__((aligned(16))) unsigned long long p2_lds[];
  constexpr Shape sh = P::shape();
  constexpr RecLayout2 L = rec_layout2(P::shape(), (uint32_t)MODE);
  constexpr uint32_t RW = L.rec_words;
  constexpr uint32_t chunk_dw = kP2ChunkRecs * RW;
  const uint32_t NP = 1u << pp.log2_parts, ring_dw = pp.ring_lines * 32u, ring_mask = ring_dw - 1u;
  const uint32_t hot_slots = pp.n_hot ? (1u << pp.log2_hot_slots) : 0u;
  unsigned int* ring = reinterpret_cast<unsigned int*>(p2_lds);
  unsigned long long* fl = p2_lds + (size_t)NP * ring_dw / 2;
  unsigned long long* hot_k = fl + NP;
  unsigned long long* hot_acc = hot_k + hot_slots;
  unsigned int* fdw = reinterpret_cast<unsigned int*>(hot_acc + (size_t)pp.n_hot * sh.n_aggs * pp.hot_copies);
  unsigned int* chunk = fdw + NP;
  unsigned int* hot_i = chunk + NP;
  unsigned int* misc = hot_i + hot_slots;
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint32_t first_limit = ring_dw / RW < kP2ChunkRecs ? ring_dw / RW : kP2ChunkRecs;
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) { fl[i] = (unsigned long long)first_limit << 32; fdw[i] = 0; chunk[i] = kNoChunk; }
  for (uint32_t i = threadIdx.x; i < hot_slots; i += blockDim.x) { hot_k[i] = sp.hot_tbl_keys[i]; hot_i[i] = sp.hot_tbl_idx[i]; }
  for (uint32_t i = threadIdx.x; i < pp.n_hot * sh.n_aggs * pp.hot_copies; i += blockDim.x) hot_acc[i] = agg_identity_dev(sh.aggs[(i / pp.hot_copies) % sh.n_aggs].kind);
  if (threadIdx.x < 4) misc[threadIdx.x] = 0;
  if (threadIdx.x == 0 && pp.tiles != (uint32_t)TILES) sp.flags[0] = 1u;     // host and kernel disagree about the round geometry: fail the query
  __syncthreads();
  const uint32_t chunk0 = blockIdx.x * pp.chunks_per_wg;     // this workgroup's private chunk region
  // partitions a lane owns in the flush phase: wave w, lane l < lanes_per_wave owns partition w * lanes_per_wave + l
  const uint32_t lanes_per_wave = NP >= (uint32_t)nwaves ? NP / (uint32_t)nwaves : 1u;
  const bool owner = NP >= (uint32_t)nwaves ? ((uint32_t)lane < lanes_per_wave) : ((uint32_t)wave < NP && lane == 0);
  const uint32_t own_p = NP >= (uint32_t)nwaves ? (uint32_t)wave * lanes_per_wave + (uint32_t)lane : (uint32_t)wave;

  const int64_t rows_per_round = (int64_t)blockDim.x * kRows * TILES;
  const int64_t nrounds = (args.n_rows + rows_per_round - 1) / rows_per_round;
  auto tile_of = [&](int64_t rd, int t) { return (rd * TILES + t) * (int64_t)nwaves + wave; };
  auto round_full = [&](int64_t rd) { return (rd + 1) * rows_per_round <= args.n_rows; };
  RegFile rf[TILES];          // indexed by compile-time constants only (p2_static_for): a dynamically indexed register file is spilled to scratch
  long long kmin_seen = 0x7fffffffffffffffll, kmax_seen = (long long)0x8000000000000000ull;   // by-product statistics of the key
  uint64_t narrow_viol = 0;     // wave-uniform (scalar registers): lanes whose value did not fit its narrowed field (make_record2, pp.check_src)
  unsigned int rec[TILES][kRows][RW];
  uint32_t part[TILES][kRows];
  bool pending[TILES][kRows];
  // evaluates the TILES tiles of round rd (their column loads may already be in flight) and leaves the rows in rec / part /
  // pending; rows of hot keys are aggregated here and never become pending
  auto finish_round = [&](int64_t rd, bool preloaded) __attribute__((always_inline)) {
    p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      bool pass[kRows];
      const int64_t row0 = tile_of(rd, t) * (int64_t)kTileRows + (int64_t)lane * kRows;
      if (preloaded) {
        run_rest_full<P>(args, row0, rf[t]);
#pragma unroll
        for (int r = 0; r < kRows; r++) pass[r] = sh.pred == kNone || ((rf[t].get(r, sh.pred) & 1) && ((rf[t].getv(sh.pred) >> r) & 1));
      } else {
        int64_t r0;
        tile_rows<P>(dsh, args, tile_of(rd, t), rf[t], pass, r0);
      }
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        bool kvalid; uint64_t key64;
        make_record2<MODE>(sh, L, pp, rf[t], r, row0 + r, rec[t][r], part[t][r], kvalid, key64, pass[r], narrow_viol);
        pending[t][r] = pass[r];
        if (MODE == (int)kP2Hash && sp.key_minmax && pass[r] && kvalid) {
          kmin_seen = (long long)key64 < kmin_seen ? (long long)key64 : kmin_seen;
          kmax_seen = (long long)key64 > kmax_seen ? (long long)key64 : kmax_seen;
        }
        if (MODE == (int)kP2Direct && part[t][r] >= NP) { if (pass[r]) sp.flags[1] = 1u; pending[t][r] = false; }   // id outside the declared range: the query fails
        if (pp.n_hot && pass[r] && kvalid && key64 != kEmptyKey) {
          uint32_t s = (uint32_t)((key64 * 0x9e3779b97f4a7c15ull) >> (64 - pp.log2_hot_slots));
          int hot = -1;
          for (;;) {
            const unsigned long long hk = hot_k[s];
            if (hk == key64) { hot = (int)hot_i[s]; break; }
            if (hk == kEmptyKey) break;
            s = (s + 1) & (hot_slots - 1);
          }
          if (hot >= 0) {
            pending[t][r] = false;
            unsigned long long* cell = hot_acc + (size_t)hot * sh.n_aggs * pp.hot_copies + ((uint32_t)lane & (pp.hot_copies - 1));
#pragma unroll
            for (int k = 0; k < kMaxAggs; k++) {
              if (k < sh.n_aggs) {
                const Agg ag = sh.aggs[k];
                const uint64_t v = ag.src != kNone ? rf[t].get(r, ag.src) : 0ull;
                const bool valid = ag.src != kNone ? ((rf[t].getv(ag.src) >> r) & 1) : true;
                const uint64_t x = agg_row_value(ag.kind, v, true, valid, (uint64_t)(row0 + r));
                if ((x != agg_identity_dev(ag.kind) || ag.kind == AGG_SUM_F) && !(ag.kind == AGG_SUM_F && !valid)) lds_atomic_agg(ag.kind, cell + (size_t)k * pp.hot_copies, x);
              }
            }
          }
        }
      }
    });
  };
  auto issue_loads = [&](int64_t rd) __attribute__((always_inline)) -> bool {
    if (rd < nrounds && round_full(rd)) {
      p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        run_loads_full<P>(args, tile_of(rd, t) * (int64_t)kTileRows + (int64_t)lane * kRows, rf[t]);
      });
      return true;
    }
    return false;
  };
  // writes every complete line of the rings this lane owns to HBM and opens / closes chunks; `final_pass` also writes the
  // partial tail and records the fill of the last chunk
  auto flush_phase = [&](bool final_pass) __attribute__((always_inline)) {
    uint32_t fill = 0, lim = 0, f_dw = 0, ch = kNoChunk, nl = 0;
    if (owner) {
      const unsigned long long f = fl[own_p];
      lim = (uint32_t)(f >> 32);
      fill = (uint32_t)f < lim ? (uint32_t)f : lim;          // appends past the limit failed: they stay pending and come back
      f_dw = fdw[own_p]; ch = chunk[own_p];
      const uint32_t avail = fill * RW;
      const uint32_t target = fill >= kP2ChunkRecs ? chunk_dw : (avail & ~31u);
      nl = (target - f_dw) >> 5;
      if ((nl || (final_pass && avail > f_dw)) && ch == kNoChunk) {
        const uint32_t local = atomicAdd(&misc[0], 1u);
        if (local >= pp.chunks_per_wg) { sp.flags[0] = 1u; nl = 0; fill = 0; }
        else { ch = chunk0 + local; sp.chunk_part[ch] = own_p; }
      }
    }
    uint32_t done = 0;
    for (;;) {
      const uint64_t m = ballot(done < nl);
      if (!m) break;
      const int rank = prefix_rank(m);
      const int g = lane >> 3, sub = lane & 7;
      int src_lane = -1;
      {
        uint64_t mm = m;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (mm) { const int o = (int)__builtin_ctzll(mm); mm &= mm - 1; if (g == i) src_lane = o; }
        }
      }
      const uint32_t my_src = own_p * ring_dw + ((((f_dw >> 5) + done) & (pp.ring_lines - 1u)) << 5);
      const uint64_t my_dst = (uint64_t)ch * chunk_dw + f_dw + done * 32u;
      const int from = src_lane < 0 ? 0 : src_lane;
      const uint32_t src = (uint32_t)__shfl((int)my_src, from, 64);
      const uint64_t dst = shfl_u64(my_dst, from);
      if (src_lane >= 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(ring + src + sub * 4);
        if (!(pp.ablate & 1u)) *reinterpret_cast<uint4*>(sp.recs + dst + sub * 4) = v;
      }
      if (done < nl && rank < 8) done++;
    }
    if (owner) {
      f_dw += nl * 32u;
      if (final_pass && ch != kNoChunk) {
        const uint32_t avail = fill * RW;
        for (uint32_t w = f_dw; w < avail; w++) sp.recs[(uint64_t)ch * chunk_dw + w] = ring[own_p * ring_dw + (w & ring_mask)];
        sp.chunk_fill[ch] = fill;
      } else if (fill >= kP2ChunkRecs && f_dw == chunk_dw) {       // chunk complete: the next flush opens a new one
        sp.chunk_fill[ch] = kP2ChunkRecs;
        ch = kNoChunk; f_dw = 0; fill = 0;
      }
      uint32_t nlim = (f_dw + ring_dw) / RW;
      if (nlim > kP2ChunkRecs) nlim = kP2ChunkRecs;
      fl[own_p] = ((unsigned long long)nlim << 32) | fill;
      fdw[own_p] = f_dw; chunk[own_p] = ch;
    }
  };
  // appends this lane's pending rows of tile t; true = some are still pending (their partition's ring / chunk was full)
  auto append_pending = [&](auto tc) __attribute__((always_inline)) -> bool {
    constexpr int t = decltype(tc)::value;
    bool mine = false;
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pending[t][r]) continue;
      const unsigned long long old = atomicAdd(&fl[part[t][r]], 1ull);
      const uint32_t pos = (uint32_t)old, lim = (uint32_t)(old >> 32);
      if (pos < lim) {
        unsigned int* base = ring + (size_t)part[t][r] * ring_dw;
        const uint32_t d0 = pos * RW;
#pragma unroll
        for (uint32_t w = 0; w < RW; w++) if (!(pp.ablate & 2u)) base[(d0 + w) & ring_mask] = rec[t][r][w];
        pending[t][r] = false;
      } else mine = true;
    }
    return mine;
  };
  // "does any lane still hold a pending row?" with ONE barrier (__syncthreads_or costs two): lanes raise one of two alternating LDS
  // flags before the barrier, everybody reads it after; lane 0 clears the other flag, which nobody touches until the next use
  uint32_t sync_points = 0;
  auto any_pending = [&](bool mine) __attribute__((always_inline)) -> bool {
    const uint32_t par = sync_points & 1u;
    sync_points++;
    if (mine) misc[1 + par] = 1u;
    __syncthreads();
    const bool any = misc[1 + par] != 0;
    if (threadIdx.x == 0) misc[2 - par] = 0u;
    return any;
  };
  // Order of a round (the vector-memory counter of gfx9 counts loads AND stores, and the number of line stores of a flush is
  // not a compile-time constant, so a wait for loaded data is a wait for every store issued before it):
  //   append(rd) | barrier | evaluate round rd+1 from its loads (the stores still in flight are a whole round old by now) |
  //   issue the loads of round rd+2 | flush(rd): line stores | barrier
  // Rows that did not fit (rare: the rings are sized for the arrival rate) take the slow path first: flush, barrier, append again.
  const int64_t stride = (int64_t)gridDim.x;
  const int64_t rd_first = (int64_t)blockIdx.x;
  bool pre = false;
  if (rd_first < nrounds) {
    finish_round(rd_first, issue_loads(rd_first));
    pre = issue_loads(rd_first + stride);
  }
  auto append_round = [&]() __attribute__((always_inline)) -> bool {
    bool mine = false;
    p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) { mine = append_pending(tc) || mine; });
    return mine;
  };
  for (int64_t rd = rd_first; rd < nrounds; rd += stride) {
    bool any = any_pending(append_round());
    while (any) {
      flush_phase(false);
      __syncthreads();
      any = any_pending(append_round());
    }
    const int64_t rd_next = rd + stride;
    if (rd_next < nrounds) {                          // uniform across the workgroup
      finish_round(rd_next, pre);
      pre = issue_loads(rd_next + stride);
    }
    flush_phase(false);
    __syncthreads();
  }
  flush_phase(true);
  if (narrow_viol && lane == 0) sp.flags[2] = 1u;     // a value outside the bounds its narrowing assumed: the query is planned again (engine.cpp)
  if (MODE == (int)kP2Hash && sp.key_minmax) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const long long a = (long long)shfl_xor_u64((uint64_t)kmin_seen, m), b = (long long)shfl_xor_u64((uint64_t)kmax_seen, m);
      kmin_seen = a < kmin_seen ? a : kmin_seen; kmax_seen = b > kmax_seen ? b : kmax_seen;
    }
    if (lane == 0 && kmin_seen <= kmax_seen) { atomicMin(sp.key_minmax, kmin_seen); atomicMax(sp.key_minmax + 1, kmax_seen); }
  }
  if (pp.n_hot) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < pp.n_hot * sh.n_aggs; i += blockDim.x) {
      const uint8_t kind = sh.aggs[i % sh.n_aggs].kind;
      uint64_t x = hot_acc[(size_t)i * pp.hot_copies];
      for (uint32_t c = 1; c < pp.hot_copies; c++) x = agg_combine(kind, x, hot_acc[(size_t)i * pp.hot_copies + c]);
      if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) atomic_agg(kind, sp.hot_out + i, x);
    }
  }
}

template <class P, int MODE, int TILES = 1>
__global__ __launch_bounds__(kP2MaxBlock) void part2_scatter_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {
  part2_scatter_body<P, MODE, TILES>(dsh, args, pp, sp);
}

// ---- the aggregation kernel --------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) RecWords3 { unsigned int a, b, c; };
struct __attribute__((packed, aligned(4))) RecWords4 { unsigned int a, b, c, d; };
template <uint32_t RW>
__device__ __forceinline__ void load_rec2(const unsigned int* p, unsigned int* r) {
  if constexpr (RW == 4) { const uint4 v = *reinterpret_cast<const uint4*>(p); r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
  else if constexpr (RW == 3) { const RecWords3 v = *reinterpret_cast<const RecWords3*>(p); r[0] = v.a; r[1] = v.b; r[2] = v.c; }
  else if constexpr (RW == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); r[0] = v.x; r[1] = v.y; }
  else if constexpr (RW % 4 == 0) {      // (records of 4k dwords in a 16-byte aligned chunk)
#pragma unroll
    for (uint32_t w = 0; w < RW; w += 4) { const uint4 v = *reinterpret_cast<const uint4*>(p + w); r[w] = v.x; r[w + 1] = v.y; r[w + 2] = v.z; r[w + 3] = v.w; }
  } else if constexpr (RW % 2 == 0) {    // 8-byte aligned: two-dword loads (24-byte wide-key records: three instead of six)
#pragma unroll
    for (uint32_t w = 0; w < RW; w += 2) { const uint2 v = *reinterpret_cast<const uint2*>(p + w); r[w] = v.x; r[w + 1] = v.y; }
  } else {
    // an odd number of dwords (20-byte wide-key records: five): the record is only 4-byte aligned, which a global_load_dwordx4 / x3 does not mind -- two load
    // instructions per record instead of five (the memory pipeline takes a wave-load every few dozen cycles whatever its width)
    uint32_t w = 0;
#pragma unroll
    for (; w + 4 <= RW; w += 4) { const RecWords4 v = *reinterpret_cast<const RecWords4*>(p + w); r[w] = v.a; r[w + 1] = v.b; r[w + 2] = v.c; r[w + 3] = v.d; }
    if constexpr (RW % 4 == 3) { const RecWords3 v = *reinterpret_cast<const RecWords3*>(p + w); r[w] = v.a; r[w + 1] = v.b; r[w + 2] = v.c; }
    else {
#pragma unroll
      for (uint32_t x = RW / 4 * 4; x < RW; x++) r[x] = p[x];
    }
  }
}

// chunks a wave keeps in flight: one-dword records (1 KB a chunk) want six; up to four dwords three; wider records two -- their buffers are kPerLane x rec_words
// registers EACH, and with three the wide-key aggregation spilled (7 registers; a scratch reload waits for every chunk load in flight)
__host__ __device__ constexpr int p2_agg_chunks_in_flight(uint32_t rec_words) { return rec_words <= 1 ? 6 : rec_words <= 4 ? 3 : 2; }

template <class S, int MODE, int NB = 3>
__device__ __forceinline__ void part2_agg_body(const S& sh, const RecLayout2& L, const PartPlan2& pp, const AggParams2& ap) {
  static_assert(NB >= 2 && NB <= 8, "chunks in flight per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned long long p2_lds[];
  constexpr uint32_t kPerLane = kP2ChunkRecs / 64;    // records of a chunk per lane
  const bool direct = MODE == (int)kP2Direct;
  const uint32_t NS = direct ? 1u << pp.log2_slots : pp.n_slots, n_aggs = sh.n_aggs, RW = L.rec_words, chunk_dw = kP2ChunkRecs * RW;
  // wide key (L.n_key_cols != 0, hash mode): tag entries [pp.n_tags] u32 | group keys [NS][KW] u64 | cells [NS][n_aggs] (NS = groups the partition's storage holds);
  // no special slots (a null key column is part of the key: its bit of the null mask is hashed and compared) -- see process_wide
  const bool wide = !direct && L.n_key_cols != 0;
  const uint32_t KW = wide ? (uint32_t)L.n_key_cols + (pp.wide_null_word ? 1u : 0u) : 0u;
  const uint32_t MAXG = NS;
  unsigned long long* keys = p2_lds;                                    // hash mode: [NS + 2] (NS = null key, NS + 1 = the key equal to EMPTY)
  unsigned long long* gkeys = p2_lds + pp.n_tags / 2;                   // wide: [NS][KW], behind the 32-bit tag entries
  unsigned long long* cells = direct ? p2_lds : wide ? gkeys + (size_t)KW * NS : keys + NS + 2;          // [(NS (+2)) * n_aggs]
  const uint32_t n_slots = (direct || wide) ? NS : NS + 2;
  __shared__ unsigned int n_groups;
  __shared__ unsigned int n_occ, cursor_l, full;
  __shared__ unsigned long long gbase;
  const uint32_t p = blockIdx.x;
  if (wide) { for (uint32_t i = threadIdx.x; i < pp.n_tags; i += blockDim.x) reinterpret_cast<unsigned int*>(p2_lds)[i] = 0u; }
  else if (!direct) for (uint32_t i = threadIdx.x; i < n_slots; i += blockDim.x) keys[i] = kEmptyKey;
  for (uint32_t i = threadIdx.x; i < n_slots * n_aggs; i += blockDim.x) cells[i] = agg_identity_dev(sh.aggs[i % n_aggs].kind);
  if (threadIdx.x == 0) { n_occ = 0; cursor_l = 0; full = 0; n_groups = 0; }
  __syncthreads();
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint64_t c_beg = ap.cl_off[p], c_end = ap.cl_off[p + 1];
  // NB chunks per wave are in flight: the chunk being aggregated and the next NB - 1 (a partition count that gives every CU only one workgroup
  // leaves 16 waves to cover the HBM latency; three 3-KB chunks per wave do for 12-byte records, one-dword records -- 1 KB a chunk -- want six:
  // config 3's aggregation pass ran at 4.1 TB/s with three).  Every lane ALWAYS loads its kPerLane records of a chunk -- past the fill, or past
  // the end of the list (clamped to the last chunk), the words are simply ignored -- so the number of loads per chunk is a constant and the wait
  // for the oldest chunk leaves the younger ones in flight; the chunk id and fill come through the scalar cache (wave-uniform address).
  unsigned int bufs[NB][kPerLane][16];   // [..][RW] (RW <= 13); unused words are never touched
  uint32_t nn[NB];
#pragma unroll
  for (int s = 0; s < NB; s++) nn[s] = 0;
  auto load_chunk = [&](uint64_t j, unsigned int (*dst)[16], uint32_t& cnt) __attribute__((always_inline)) {
    const uint64_t jc = j < c_end ? j : c_end - 1;
    const uint32_t jlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)jc), jhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(jc >> 32));
    const uint64_t ju = ((uint64_t)jhi << 32) | jlo;
    const uint32_t id = uniform_ld(ap.cl_ids, ju);
    const uint32_t fill = uniform_ld(ap.chunk_fill, (uint64_t)id);
    cnt = j < c_end ? fill : 0u;
    const unsigned int* base = ap.recs + (uint64_t)id * chunk_dw;
    if (RW == 1) {     // one-dword records: the lane takes four consecutive ones with ONE 16-byte load (the memory pipeline accepts a wave-load every ~40 cycles whatever its width)
      static_assert(kPerLane == 4, "four records per lane");
      const uint4 q = reinterpret_cast<const uint4*>(base)[lane];
      dst[0][0] = q.x; dst[1][0] = q.y; dst[2][0] = q.z; dst[3][0] = q.w;
      return;
    }
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t i = (uint32_t)lane + u * 64u;
      switch (RW) {
        case 1: dst[u][0] = base[i]; break;
        case 2: load_rec2<2>(base + (size_t)i * 2, dst[u]); break;
        case 3: load_rec2<3>(base + (size_t)i * 3, dst[u]); break;
        case 4: load_rec2<4>(base + (size_t)i * 4, dst[u]); break;
        case 5: load_rec2<5>(base + (size_t)i * 5, dst[u]); break;
        case 6: load_rec2<6>(base + (size_t)i * 6, dst[u]); break;
        case 7: load_rec2<7>(base + (size_t)i * 7, dst[u]); break;
        case 8: load_rec2<8>(base + (size_t)i * 8, dst[u]); break;
        case 10: load_rec2<10>(base + (size_t)i * 10, dst[u]); break;
        case 12: load_rec2<12>(base + (size_t)i * 12, dst[u]); break;
        default:
#pragma unroll
          for (uint32_t w = 0; w < 16; w++) if (w < RW) dst[u][w] = base[(size_t)i * RW + w];
          break;
      }
    }
  };
  // Wide keys.  LDS: tag entries [NT] u32 in buckets of EIGHT (two 16-byte reads look at a whole bucket; NT = pp.n_tags, a multiple of 8, about four entries per
  // group of capacity: a load below 0.25, so a key sits in its home bucket but for ~1e-5 of them) | group keys [MAXG][KW] u64 | cells [MAXG][n_aggs]; MAXG = pp.n_slots
  // groups per partition, numbered by an LDS counter in order of first appearance.  Entry = busy << 31 | tag19 << 12 | group ordinal (0 = empty; tag19 != 0).
  //   fast path  straight-line, no divergent control flow: hash -> the bucket (2 x ds_read_b128) -> first entry with the record's tag -> the group's key words (one
  //              ds_read_b128 for two key columns) -> compare -> the cell atomics under that predicate.  Every record of a key that is already in the table ends here.
  //   slow path  only if some lane's record did not: a new key (claimed by CAS: empty -> busy | tag; ordinal from the counter; key words written; entry published --
  //              LDS operations of a wave complete in order, so whoever sees the published entry sees the words), a key pushed out of a full home bucket (the search
  //              goes on bucket by bucket), a tag that matched another key, an entry that is still busy (looked at again in the next round -- never a spin inside a
  //              round: the claimer may be a lane of the same wave).
  // Round 4 walked a table of 64-bit hash words slot by slot, every step behind a divergent branch: ~2500 instructions per 256-record chunk, most of them scalar
  // mask bookkeeping, on a pass that is bound by the instruction stream of its sixteen waves (8 of the pass's 11.5 ms at 1e9 records of a two-column key).
  unsigned int* tags = reinterpret_cast<unsigned int*>(p2_lds);
  const uint32_t NT = pp.n_tags, n_buckets = NT >> 3;
  auto wide_update = [&](const unsigned int* rec, uint32_t ord) __attribute__((always_inline)) {
    const uint32_t vbits = L.has_valid ? rec[L.valid_off] : 0xffffffffu;
    const uint64_t rowid = L.has_rowid ? ((uint64_t)rec[L.rowid_off] | ((uint64_t)rec[L.rowid_off + 1] << 32)) : 0ull;
    unsigned long long* cell = cells + (size_t)ord * n_aggs;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)kMaxAggs; k++) {
      if (k >= n_aggs) break;
      const uint8_t kind = sh.aggs[k].kind;
      const uint8_t sj = L.agg_src[k];
      uint64_t v = 0ull;
      bool valid = true;
      if (sj != kNone) {
        const uint32_t lo = rec[L.src_off[sj]];
        if (L.src_kind[sj] == 3) v = (uint64_t)pp.src_base[sj] + (uint64_t)lo;
        else v = L.src_kind[sj] == 0 ? ((uint64_t)lo | ((uint64_t)rec[L.src_off[sj] + 1] << 32)) : (L.src_kind[sj] == 1 ? (uint64_t)(long long)(int)lo : (uint64_t)lo);
        valid = (vbits >> sj) & 1;
      }
      const uint64_t x = agg_row_value(kind, v, true, valid, rowid);
      if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) {
        if (kind == AGG_SUM_F && !valid) continue;
        lds_atomic_agg(kind, cell + k, x);
      }
    }
  };
  auto wide_word = [&](const unsigned int* rec, uint32_t j) __attribute__((always_inline)) -> unsigned long long {      // key word j of a record (j == n_key_cols: the null mask)
    if (j < (uint32_t)L.n_key_cols) return (unsigned long long)rec[2 * j] | ((unsigned long long)rec[2 * j + 1] << 32);
    const uint32_t vbits = L.has_valid ? rec[L.valid_off] : 0xffffffffu;
    return (unsigned long long)(((vbits >> 24) & ((1u << L.n_key_cols) - 1u)) ^ ((1u << L.n_key_cols) - 1u));
  };
  auto process_wide = [&](unsigned int (*cur)[16], uint32_t cnt_cur) __attribute__((always_inline)) {
    uint32_t bucket[kPerLane], tagf[kPerLane];
    bool pending[kPerLane];
    // ---- fast path
    uint4 qa[kPerLane], qb[kPerLane];
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const unsigned int* rec = cur[u];
      // a hash of the key words for THIS table (the partition consumed the scatter's hash): 32-bit multiplies only
      uint32_t x = 0x9e3779b9u;
#pragma unroll
      for (uint32_t j = 0; j < (uint32_t)kMaxKeys + 1u; j++) {
        if (j >= KW) break;
        const unsigned long long w = wide_word(rec, j);
        x = (x ^ (uint32_t)w) * 0x85ebca77u;
        x = (x ^ (uint32_t)(w >> 32)) * 0xc2b2ae3du;
      }
      x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 13;
      bucket[u] = __umulhi(x, n_buckets);
      uint32_t t19 = (x * 0x297a2d39u) >> 13;
      if (!t19) t19 = 1u;
      tagf[u] = t19 << 12;
      qa[u] = *reinterpret_cast<const uint4*>(&tags[bucket[u] * 8u]);
      qb[u] = *reinterpret_cast<const uint4*>(&tags[bucket[u] * 8u + 4u]);
    }
    lds_order();
    uint32_t ord[kPerLane];
    bool hit[kPerLane];
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t t = tagf[u];
      const uint32_t e0 = qa[u].x, e1 = qa[u].y, e2 = qa[u].z, e3 = qa[u].w, e4 = qb[u].x, e5 = qb[u].y, e6 = qb[u].z, e7 = qb[u].w;
      // the first entry of the bucket that carries the tag, published (busy bit clear): compare the upper twenty bits
      uint32_t e = 0u;
      e = ((e7 & 0xfffff000u) == t) ? e7 : e; e = ((e6 & 0xfffff000u) == t) ? e6 : e; e = ((e5 & 0xfffff000u) == t) ? e5 : e; e = ((e4 & 0xfffff000u) == t) ? e4 : e;
      e = ((e3 & 0xfffff000u) == t) ? e3 : e; e = ((e2 & 0xfffff000u) == t) ? e2 : e; e = ((e1 & 0xfffff000u) == t) ? e1 : e; e = ((e0 & 0xfffff000u) == t) ? e0 : e;
      hit[u] = e != 0u;
      ord[u] = e & 0xfffu;
    }
    // the groups' key words (ordinal 0 for a miss: any valid address)
    unsigned long long kw[kPerLane][kMaxKeys + 1];
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      if (KW == 2u) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&gkeys[(size_t)ord[u] * 2u]);
        kw[u][0] = v.x; kw[u][1] = v.y;
      } else {
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)kMaxKeys + 1u; j++) if (j < KW) kw[u][j] = lds_ld(&gkeys[(size_t)ord[u] * KW + j]);
      }
    }
    lds_order();
    bool any_pending = false;
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t i = (uint32_t)lane + u * 64u;
      const unsigned int* rec = cur[u];
      bool same = hit[u];
#pragma unroll
      for (uint32_t j = 0; j < (uint32_t)kMaxKeys + 1u; j++) if (j < KW) same = same && kw[u][j] == wide_word(rec, j);
      const bool live = i < cnt_cur && !(pp.ablate & 8u);
      pending[u] = live && !same;
      any_pending = any_pending || pending[u];
      if (live && same && !(pp.ablate & 4u)) wide_update(rec, ord[u]);
    }
    if (!__any(any_pending)) return;
    // ---- slow path (see above): record by record, position by position
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {      // (unrolled: the record buffers are registers, a run-time index would put them into scratch)
      if (!__any(pending[u])) continue;
      const unsigned int* rec = cur[u];
      uint32_t pos = bucket[u] * 8u, steps = 0;
      bool todo = pending[u];
      for (uint32_t round = 0; __any(todo); round++) {
        if (round > 4u * NT + 64u) { full = 1; break; }
        if (lds_ld(&full)) break;                                                                // (a claimed entry may never be published once the storage is full)
        if (todo) {
          for (;;) {
            if (steps > NT) { full = 1; todo = false; break; }                                 // every entry belongs to another key: a full table
            const uint32_t e = lds_ld(&tags[pos]);
            if (e == 0u) {
              if (atomicCAS(&tags[pos], 0u, 0x80000000u | tagf[u]) != 0u) continue;           // somebody else took it: look at what is there now
              const uint32_t o = atomicAdd(&n_groups, 1u);
              if (o >= MAXG) { full = 1; todo = false; break; }                                // more groups than the partition's storage holds
#pragma unroll
              for (uint32_t j = 0; j < (uint32_t)kMaxKeys + 1u; j++) if (j < KW) lds_st(&gkeys[(size_t)o * KW + j], wide_word(rec, j));
              lds_order();                                                                       // the words first, then the entry that announces them
              lds_st(&tags[pos], tagf[u] | o);
              if (!(pp.ablate & 4u)) wide_update(rec, o);
              todo = false;
              break;
            }
            if ((e & 0x7ffff000u) == tagf[u]) {
              if (e & 0x80000000u) break;                                                        // its key words are being written: this entry again in the next round
              const uint32_t o = e & 0xfffu;
              bool same = true;
#pragma unroll
              for (uint32_t j = 0; j < (uint32_t)kMaxKeys + 1u; j++) if (j < KW) same = same && lds_ld(&gkeys[(size_t)o * KW + j]) == wide_word(rec, j);
              if (same) { if (!(pp.ablate & 4u)) wide_update(rec, o); todo = false; break; }
            }
            pos = pos + 1u == NT ? 0u : pos + 1u;
            steps++;
          }
        }
      }
    }
  };
  auto process = [&](unsigned int (*cur)[16], uint32_t cnt_cur) __attribute__((always_inline)) {
    if (wide) { process_wide(cur, cnt_cur); return; }
    if ((L.pack == kPackPair || L.pack == kPackPairV) && !direct) {
      // a record is a PAIR of rows {off0 lo, off1 lo, off0 hi16 | off1 hi16 << 16, value0, value1}, key = key_base + 48-bit offset (fused.hpp kPackPair, hash mode); the
      // offset 2^48 - 1 = the half is absent.  The same three passes as below (decode + first table word, resolve, update) over 2 x kPerLane rows.
      constexpr uint32_t NH = 2 * kPerLane;
      uint32_t slot[NH]; uint64_t key[NH], val[NH]; unsigned long long first[NH]; bool live[NH];
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) {
        const unsigned int* rec = cur[u];
        const bool in = (uint32_t)lane + u * 64u < cnt_cur;
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
          const uint32_t i = 2 * u + h;
          if (L.pack == kPackPairV) {      // {key0, key1, voff0 lo, voff1 lo, voff0 hi16 | voff1 hi16 << 16}: whole keys, values as 48-bit offsets
            const uint32_t hi = (rec[6] >> (16 * h)) & 0xffffu;
            live[i] = in && hi != 0xffffu;
            key[i] = (uint64_t)rec[2 * h] | ((uint64_t)rec[2 * h + 1] << 32);
            val[i] = (uint64_t)pp.src_base[0] + ((uint64_t)rec[4 + h] | ((uint64_t)hi << 32));
          } else {
            const uint64_t koff = (uint64_t)rec[h] | ((uint64_t)((rec[2] >> (16 * h)) & 0xffffu) << 32);
            live[i] = in && koff != kPairAbsent48;
            key[i] = (uint64_t)pp.key_base + koff; val[i] = (uint64_t)rec[3 + 2 * h] | ((uint64_t)rec[4 + 2 * h] << 32);
          }
          slot[i] = 0; first[i] = 0;
          if (!live[i]) continue;
          if (key[i] == kEmptyKey) { slot[i] = NS + 1; first[i] = kEmptyKey - 1; }
          else { slot[i] = (uint32_t)((((key[i] * 0x9e3779b97f4a7c15ull) >> 32) * (uint64_t)NS) >> 32); first[i] = keys[slot[i]]; }
        }
      }
#pragma unroll
      for (uint32_t i = 0; i < NH; i++) {
        if (!live[i]) continue;
        if (slot[i] >= NS) { keys[slot[i]] = 0; continue; }
        unsigned long long c = first[i];
        uint32_t sl = slot[i], probe = 0;
        for (;; probe++) {
          if (c == key[i]) break;
          if (c == kEmptyKey) {
            const unsigned long long old = atomicCAS(&keys[sl], (unsigned long long)kEmptyKey, (unsigned long long)key[i]);
            if (old == kEmptyKey || old == key[i]) break;
          }
          sl = sl + 1 == NS ? 0u : sl + 1;
          if (probe >= NS) { full = 1; live[i] = false; break; }
          c = keys[sl];
        }
        slot[i] = sl;
      }
#pragma unroll
      for (uint32_t i = 0; i < NH; i++) {
        if (!live[i]) continue;
        unsigned long long* cell = cells + (size_t)slot[i] * n_aggs;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)kMaxAggs; k++) {
          if (k >= n_aggs) break;
          const uint8_t kind = sh.aggs[k].kind;
          const uint64_t x = agg_row_value(kind, L.agg_src[k] != kNone ? val[i] : 0ull, true, true, 0ull);
          if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) lds_atomic_agg(kind, cell + k, x);
        }
      }
      return;
    }
    if (L.pack == kPackPair) {
      // a record is a PAIR of rows {slot0 | slot1 << 16, value0, value1} of a direct-address partition (fused.hpp kPackPair); slot 0xffff = the half is absent
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) {
        if ((uint32_t)lane + u * 64u >= cnt_cur) continue;
        const unsigned int* rec = cur[u];
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
          const uint32_t sl = h ? rec[0] >> 16 : rec[0] & 0xffffu;
          if (sl == kPairAbsent) continue;
          const uint64_t v = (uint64_t)rec[1 + 2 * h] | ((uint64_t)rec[2 + 2 * h] << 32);
          unsigned long long* cell = cells + (size_t)sl * n_aggs;
#pragma unroll
          for (uint32_t k = 0; k < (uint32_t)kMaxAggs; k++) {
            if (k >= n_aggs) break;
            const uint8_t kind = sh.aggs[k].kind;
            const uint64_t x = agg_row_value(kind, L.agg_src[k] != kNone ? v : 0ull, true, true, 0ull);
            if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) lds_atomic_agg(kind, cell + k, x);
          }
        }
      }
      return;
    }
    // the lane's records of the chunk are handled in three passes so that their LDS round trips overlap: (1) decode the key and
    // read the table word of its home slot for every record, (2) resolve the slot (hit on the first probe in the common case; the
    // CAS / linear-probe loop otherwise), (3) update the cells
    uint32_t slot[kPerLane];
    uint64_t key[kPerLane];
    unsigned long long first[kPerLane];
    bool live[kPerLane];
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t i = RW == 1 ? (uint32_t)lane * 4u + u : (uint32_t)lane + u * 64u;     // as load_chunk dealt the chunk's records out
      live[u] = i < cnt_cur;
      slot[u] = 0; key[u] = 0; first[u] = 0;
      if (!live[u]) continue;
      const unsigned int* rec = cur[u];
      if (direct) { slot[u] = L.pack == kPackFused ? (rec[0] & ((1u << pp.key_shift) - 1u)) : rec[0]; continue; }
      const uint32_t vbits = L.has_valid ? rec[L.valid_off] : 0xffffffffu;
      if (L.key_words == 2) key[u] = (uint64_t)rec[0] | ((uint64_t)rec[1] << 32);
      else key[u] = L.key_kind == 1 ? (uint64_t)(long long)(int)rec[0] : (uint64_t)rec[0];
      if (!(vbits >> 31)) { slot[u] = NS; first[u] = kEmptyKey - 1; }                 // resolved below without probing
      else if (key[u] == kEmptyKey) { slot[u] = NS + 1; first[u] = kEmptyKey - 1; }
      else {
        // a second hash (the partition consumed the top bits of the first); the table has ANY number of slots: slot = floor(hash32 * NS / 2^32)
        slot[u] = (uint32_t)((((key[u] * 0x9e3779b97f4a7c15ull) >> 32) * (uint64_t)NS) >> 32);
        first[u] = keys[slot[u]];
      }
    }
    if (!direct) {
#pragma unroll
      for (uint32_t u = 0; u < kPerLane; u++) {
        if (!live[u]) continue;
        if (slot[u] >= NS) { keys[slot[u]] = 0; continue; }          // null-key / EMPTY-pattern groups: mark the slot occupied
        unsigned long long c = first[u];
        uint32_t sl = slot[u], probe = 0;
        for (;; probe++) {
          if (c == key[u]) break;
          if (c == kEmptyKey) {
            const unsigned long long old = atomicCAS(&keys[sl], (unsigned long long)kEmptyKey, (unsigned long long)key[u]);
            if (old == kEmptyKey || old == key[u]) break;
          }
          sl = sl + 1 == NS ? 0u : sl + 1;
          if (probe >= NS) { full = 1; live[u] = false; break; }
          c = keys[sl];
        }
        slot[u] = sl;
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      if (!live[u]) continue;
      const unsigned int* rec = cur[u];
      const uint32_t vbits = L.has_valid ? rec[L.valid_off] : 0xffffffffu;
      const uint64_t rowid = L.has_rowid ? ((uint64_t)rec[L.rowid_off] | ((uint64_t)rec[L.rowid_off + 1] << 32)) : 0ull;
      unsigned long long* cell = cells + (size_t)slot[u] * n_aggs;
#pragma unroll
      for (uint32_t k = 0; k < (uint32_t)kMaxAggs; k++) {
        if (k >= n_aggs) break;
        const uint8_t kind = sh.aggs[k].kind;
        const uint8_t sj = L.agg_src[k];
        uint64_t v = 0ull;
        bool valid = true;
        if (sj != kNone) {
          const uint32_t lo = rec[L.src_off[sj]];
          if (L.pack == kPackFused) v = (uint64_t)pp.src_base[0] + (uint64_t)(lo >> pp.key_shift);
          else if (L.src_kind[sj] == 3) v = (uint64_t)pp.src_base[sj] + (uint64_t)lo;
          else v = L.src_kind[sj] == 0 ? ((uint64_t)lo | ((uint64_t)rec[L.src_off[sj] + 1] << 32)) : (L.src_kind[sj] == 1 ? (uint64_t)(long long)(int)lo : (uint64_t)lo);
          valid = (vbits >> sj) & 1;
        }
        const uint64_t x = agg_row_value(kind, v, true, valid, rowid);
        if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) {
          if (kind == AGG_SUM_F && !valid) continue;
          lds_atomic_agg(kind, cell + k, x);
        }
      }
    }
  };
  if (c_beg < c_end) {
    const uint64_t step = (uint64_t)nwaves;
    uint64_t j = c_beg + (uint64_t)wave;
#pragma unroll
    for (int s = 0; s + 1 < NB; s++) load_chunk(j + (uint64_t)s * step, bufs[s], nn[s]);
    for (bool done = false; !done;) {     // the buffers rotate by (compile-time) index: copying a buffer would wait for its loads
#pragma unroll
      for (int s = 0; s < NB; s++) {
        if (j >= c_end || lds_ld(&full)) { done = true; break; }      // a full table: the result is discarded anyway; probing it record by record would take O(slots) each
        load_chunk(j + (uint64_t)(NB - 1) * step, bufs[(s + NB - 1) % NB], nn[(s + NB - 1) % NB]);
        process(bufs[s], nn[s]);
        j += step;
      }
    }
  }
  __syncthreads();
  if (full) { if (threadIdx.x == 0) atomicExch(ap.overflow, 1u); return; }
  // emit the partition's groups: count, reserve once, write
  uint32_t mine = 0;
  for (uint32_t s = threadIdx.x; s < n_slots; s += blockDim.x) mine += direct ? (cells[(size_t)s * n_aggs + pp.len_idx] != 0) : wide ? (s < n_groups) : (keys[s] != kEmptyKey);      // wide: groups are numbered densely
  if (mine) atomicAdd(&n_occ, mine);
  __syncthreads();
  if (threadIdx.x == 0) gbase = n_occ ? atomicAdd(ap.counter, (unsigned long long)n_occ) : 0ull;
  __syncthreads();
  if (gbase + n_occ > ap.max_groups) { if (threadIdx.x == 0) atomicExch(ap.overflow, 2u); return; }
  for (uint32_t s = threadIdx.x; s < n_slots; s += blockDim.x) {
    if (direct ? (cells[(size_t)s * n_aggs + pp.len_idx] == 0) : wide ? (s >= n_groups) : (keys[s] == kEmptyKey)) continue;
    const uint64_t o = gbase + atomicAdd(&cursor_l, 1u);
    if (direct) { ap.out_keys[o] = pp.interleave ? (((uint64_t)s << pp.log2_parts) | p) : (((uint64_t)p << pp.key_shift) | s); ap.out_kvalid[o] = 1; }
    else if (wide) {       // key words and per-column valid flags, column-major with stride max_groups (the layout of the HBM-table path's result: FusedAggResult::wide_words / wide_valid)
      const uint32_t nm = pp.wide_null_word ? (uint32_t)gkeys[(size_t)s * KW + L.n_key_cols] : 0u;
      for (uint32_t j = 0; j < (uint32_t)L.n_key_cols; j++) {
        ap.out_keys[(size_t)j * ap.max_groups + o] = gkeys[(size_t)s * KW + j];
        ap.out_kvalid[(size_t)j * ap.max_groups + o] = (unsigned char)(((nm >> j) & 1u) ^ 1u);
      }
    }
    else { ap.out_keys[o] = s < NS ? keys[s] : (s == NS ? 0ull : kEmptyKey); ap.out_kvalid[o] = s == NS ? 0 : 1; }
    for (uint32_t k = 0; k < n_aggs; k++) ap.out_acc[o * n_aggs + k] = cells[(size_t)s * n_aggs + k];
  }
}

template <class P, int MODE, int PACK = 0>
__global__ __launch_bounds__(kP2AggBlock) void part2_agg_kernel(PartPlan2 pp, AggParams2 ap) {
  static_assert(P::kStatic, "the partitioned group-by runs specialised programs only (AOT or JIT)");
  constexpr Shape csh = P::shape();
  constexpr RecLayout2 cl = rec_layout2(P::shape(), (uint32_t)MODE, (uint32_t)PACK);
  part2_agg_body<Shape, MODE, p2_agg_chunks_in_flight(cl.rec_words)>(csh, cl, pp, ap);
}

}  // namespace k
}  // namespace plx
