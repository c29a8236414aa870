// partition3_device.hpp -- scatter pass of the partitioned high-cardinality group-by, third generation (kernels_partition.hip has the
// pipeline; the aggregation pass and the chunk bookkeeping are those of partition2_device.hpp).  Device-only header: compiled ahead of
// time for the benchmark shapes and at run time by hiprtc (jit.cpp) for any other program shape.
//
// The second-generation scatter appended every row to its partition's LDS ring and flushed complete lines after a barrier; measured
// (tools/micro_part3.hip, profiles/r03/micro_part3*.txt) its skeleton -- ring scan, two barriers per 2-4 K rows -- cost more than its
// bytes.  This one sorts a whole TILE of rows by partition inside the workgroup and writes whole lines only:
//
//   rank      one returning ds_add_u32 per row on the partition's counter: the row's rank within (tile, partition)
//   scan      one wave turns the counts into offsets of the sorted tile and, per partition, (carry + run) dwords into a number of whole
//             128-B lines and their destination: the rest of the partition's current chunk, then fresh chunks (consecutive, from the
//             workgroup's PRIVATE region: no counting pass, no global atomics)
//   sort      every row's packed record goes to its slot of the LDS tile
//   copy-out  a 16-lane group per partition streams carry + run out as aligned 128-B lines (8 B per lane) and leaves the dwords that do
//             not fill a line in the partition's CARRY line for the next round: partial lines never reach HBM (a partial line is a
//             read-modify-write at the DRAM: 6.1 ms per 1e9 rows with them, 5.0 ms without)
// Three barriers per round of blockDim * 2 * TILES rows (8192 for the benchmark shapes).  Rows of hot keys (heavy hitters of the sample)
// never become records: they are aggregated in LDS accumulators on the spot, as before.
#pragma once
#include "partition2_device.hpp"

namespace plx {
namespace k {

// LDS of the scatter kernel (dynamic shared memory), in this order (u64 arrays first):
//   hot_k [hot_slots] u64, hot_acc [n_hot * n_aggs * copies] u64
//   sorted [T * RW] u32           the tile, sorted by partition (T = block * kRows * TILES rows)
//   carry  [NP][32] u32           dwords of a partition that do not fill a line yet
//   cnt, off[NP + 1], carry_dw, dstA, lines_left, dstB, cur_chunk, cur_lines [NP] u32 each; hot_i [hot_slots] u32; misc [4] u32
__host__ __device__ inline size_t part3_scatter_lds(uint32_t T, uint32_t RW, uint32_t NP, uint32_t hot_slots, uint32_t n_hot, uint32_t n_aggs, uint32_t copies) {
  return (size_t)hot_slots * 8 + (size_t)n_hot * n_aggs * copies * 8 + (size_t)T * RW * 4 + (size_t)NP * 128 + ((size_t)NP * 8 + 1) * 4 + (size_t)hot_slots * 4 + 16;
}

template <class P, int MODE, int TILES, int PACK, bool HOT = true>
__device__ __forceinline__ void part3_scatter_body(const Shape dsh, const Args args, const PartPlan2 pp, const ScatterParams2 sp) {
  static_assert(P::kStatic, "the partitioned group-by runs specialised programs only (AOT or JIT)");
  extern __shared__ unsigned long long p2_lds[];
  constexpr Shape sh = P::shape();
  constexpr RecLayout2 L = rec_layout2(P::shape(), (uint32_t)MODE, (uint32_t)PACK);
  constexpr uint32_t RW = L.rec_words;
  constexpr uint32_t chunk_dw = kP2ChunkRecs * RW, cap_lines = chunk_dw / 32;      // a chunk holds whole records AND whole lines
  const uint32_t NP = 1u << pp.log2_parts;
  const uint32_t hot_slots = pp.n_hot ? (1u << pp.log2_hot_slots) : 0u;
  const uint32_t T = blockDim.x * kRows * TILES;
  unsigned long long* hot_k = p2_lds;
  unsigned long long* hot_acc = hot_k + hot_slots;
  unsigned int* sorted = reinterpret_cast<unsigned int*>(hot_acc + (size_t)pp.n_hot * sh.n_aggs * pp.hot_copies);
  unsigned int* carry = sorted + (size_t)T * RW;
  unsigned int* cnt = carry + (size_t)NP * 32;
  unsigned int* off = cnt + NP;
  unsigned int* carry_dw = off + NP + 1;
  unsigned int* dstA = carry_dw + NP;         // first destination of a partition's lines this round, in lines (chunk * cap_lines + line)
  unsigned int* lines_left = dstA + NP;       // lines that still fit there
  unsigned int* dstB = lines_left + NP;       // the rest goes here (fresh consecutive chunks), in lines
  unsigned int* cur_chunk = dstB + NP;
  unsigned int* cur_lines = cur_chunk + NP;   // lines of the current chunk already written (cap_lines: none open / full)
  unsigned int* hot_i = cur_lines + NP;
  unsigned int* misc = hot_i + hot_slots;     // [0] next chunk of this workgroup's region
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) { cnt[i] = 0; carry_dw[i] = 0; cur_chunk[i] = kNoChunk; cur_lines[i] = cap_lines; }
  for (uint32_t i = threadIdx.x; i < hot_slots; i += blockDim.x) { hot_k[i] = sp.hot_tbl_keys[i]; hot_i[i] = sp.hot_tbl_idx[i]; }
  for (uint32_t i = threadIdx.x; i < pp.n_hot * sh.n_aggs * pp.hot_copies; i += blockDim.x) hot_acc[i] = agg_identity_dev(sh.aggs[(i / pp.hot_copies) % sh.n_aggs].kind);
  if (threadIdx.x < 4) misc[threadIdx.x] = 0;
  if (threadIdx.x == 0 && (pp.tiles != (uint32_t)TILES || pp.pack != (uint32_t)PACK || pp.rec_words != RW)) sp.flags[0] = 1u;     // host and kernel disagree about the geometry: fail the query
  __syncthreads();
  const uint32_t chunk0 = blockIdx.x * pp.chunks_per_wg;     // this workgroup's private chunk region
  // `need` fresh consecutive chunks for partition p -> the first one
  auto open_chunks = [&](uint32_t p, uint32_t need) -> uint32_t {
    const uint32_t local = atomicAdd(&misc[0], need);
    if (local + need > pp.chunks_per_wg) { sp.flags[0] = 1u; return chunk0; }       // cannot happen by construction; the query fails if it does
    for (uint32_t e = 0; e < need; e++) sp.chunk_part[chunk0 + local + e] = p;
    return chunk0 + local;
  };

  const int64_t rows_per_round = (int64_t)blockDim.x * kRows * TILES;
  const int64_t nrounds = (args.n_rows + rows_per_round - 1) / rows_per_round;
  auto tile_of = [&](int64_t rd, int t) { return (rd * TILES + t) * (int64_t)nwaves + wave; };
  auto round_full = [&](int64_t rd) { return (rd + 1) * rows_per_round <= args.n_rows; };
  RegFile rf[TILES];          // indexed by compile-time constants only (p2_static_for)
  long long kmin_seen = 0x7fffffffffffffffll, kmax_seen = (long long)0x8000000000000000ull;   // by-product statistics of the key
  unsigned int rec[TILES][kRows][RW];
  uint32_t part[TILES][kRows];     // partition of a row that becomes a record (later: | rank << 10); kNotPending otherwise
  constexpr uint32_t kNotPending = 0xffffffffu;
  // evaluates the TILES tiles of round rd (their column loads may already be in flight) into rec / part / pending; rows of hot keys
  // are aggregated here and never become pending
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
        make_record2<MODE>(sh, L, pp, rf[t], r, row0 + r, rec[t][r], part[t][r], kvalid, key64);
        bool pend = pass[r];
        if (MODE == (int)kP2Hash && sp.key_minmax && pass[r] && kvalid) {
          kmin_seen = (long long)key64 < kmin_seen ? (long long)key64 : kmin_seen;
          kmax_seen = (long long)key64 > kmax_seen ? (long long)key64 : kmax_seen;
        }
        if (MODE == (int)kP2Direct && pp.oob_drop && !kvalid) pend = false;                                      // join probe: a null key matches nothing
        if (MODE == (int)kP2Direct && part[t][r] >= NP) { if (pass[r] && !pp.oob_drop) sp.flags[1] = 1u; pend = false; }   // id outside the declared range: the query fails (group-by) / the row matches nothing (join probe)
        if (HOT && pp.n_hot && pass[r] && kvalid && key64 != kEmptyKey) {
          uint32_t s = (uint32_t)((key64 * 0x9e3779b97f4a7c15ull) >> (64 - pp.log2_hot_slots));
          int hot = -1;
          for (;;) {
            const unsigned long long hk = hot_k[s];
            if (hk == key64) { hot = (int)hot_i[s]; break; }
            if (hk == kEmptyKey) break;
            s = (s + 1) & (hot_slots - 1);
          }
          if (hot >= 0) {
            pend = false;
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
        if (!pend) part[t][r] = kNotPending;
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

  const int64_t stride = (int64_t)gridDim.x;
  const int64_t rd_first = (int64_t)blockIdx.x;
  // When do the next round's column loads go out?  One-dword records leave room in the register file for the loaded columns of a whole round
  // next to the records of the current one: the loads are issued a full round ahead (kEarly).  Wider records do not: with both live the kernel
  // spilled, and every scratch reload is a vmcnt(0) that also waits for these very loads (measured, 1e9 rows: 12-B records 6.96 -> 6.09 ms,
  // 8-B 6.07 -> 5.62 with the late issue; 4-B 4.81 early vs 5.07 late) -- there they go out after the sort step, when the records have left
  // the registers, and land during the copy-out.
  constexpr bool kEarly = RW == 1 || TILES <= 2;
  bool pre_early = false;
  if (rd_first < nrounds) {
    finish_round(rd_first, issue_loads(rd_first));
    if (kEarly) pre_early = issue_loads(rd_first + stride);
  }
  const uint32_t per_lane = NP >> 6;          // partitions per lane of the scan wave (NP is a power of two >= 64)
  for (int64_t rd = rd_first; rd < nrounds; rd += stride) {
    // ---- rank: one LDS atomic per surviving row (kept in the row's `part` word: partition in the low 10 bits, rank above them)
    p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < kRows; r++) if (part[t][r] != kNotPending) part[t][r] |= atomicAdd(&cnt[part[t][r]], 1u) << 10;
    });
    __syncthreads();                                                                  // A: the counts are complete
    // ---- scan (one wave): tile offsets, lines and destinations of every partition
    if (wave == 0) {
      uint32_t s = 0;
      for (uint32_t q = 0; q < per_lane; q++) s += cnt[(uint32_t)lane * per_lane + q];
      uint32_t incl = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
      uint32_t o = incl - s;
      for (uint32_t q = 0; q < per_lane; q++) {
        const uint32_t p = (uint32_t)lane * per_lane + q;
        const uint32_t c = cnt[p];
        off[p] = o; o += c;
        cnt[p] = 0;
        const uint32_t nl = (carry_dw[p] + c * RW) >> 5;
        if (nl) {
          uint32_t ch = cur_chunk[p], ln = cur_lines[p];
          const uint32_t left = cap_lines - ln;                                        // 0 when no chunk is open (ln == cap_lines)
          dstA[p] = ch * cap_lines + ln; lines_left[p] = left;
          if (nl > left) {
            const uint32_t extra = nl - left, need = (extra + cap_lines - 1) / cap_lines;
            if (ch != kNoChunk) sp.chunk_fill[ch] = kP2ChunkRecs;
            const uint32_t first = open_chunks(p, need);
            for (uint32_t e = 0; e + 1 < need; e++) sp.chunk_fill[first + e] = kP2ChunkRecs;
            dstB[p] = first * cap_lines;
            ch = first + need - 1; ln = extra - (need - 1) * cap_lines;
          } else ln += nl;
          cur_chunk[p] = ch; cur_lines[p] = ln;
        }
      }
      if (lane == 63) off[NP] = o;
    }
    __syncthreads();                                                                  // B: offsets and destinations are known
    // ---- sort: every record to its slot of the tile
    p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        if (part[t][r] == kNotPending) continue;
        unsigned int* dst = sorted + (size_t)(off[part[t][r] & 1023u] + (part[t][r] >> 10)) * RW;
#pragma unroll
        for (uint32_t w = 0; w < RW; w++) if (!(pp.ablate & 2u)) dst[w] = rec[t][r][w];
      }
    });
    const int64_t rd_next = rd + stride;
    bool pre = pre_early;
    if (!kEarly) pre = issue_loads(rd_next);
    __syncthreads();                                                                  // C: the tile is sorted
    // ---- copy-out: whole lines to HBM, the rest into the carry lines
    {
      const uint32_t g = threadIdx.x >> 4, l16 = threadIdx.x & 15u;
      for (uint32_t p = g; p < NP; p += blockDim.x >> 4) {
        const uint32_t o_dw = off[p] * RW, r_dw = (off[p + 1] - off[p]) * RW, c_dw = carry_dw[p];
        const uint32_t total = c_dw + r_dw, nl = total >> 5, rem = total & 31u;
        const unsigned int* cy = carry + (size_t)p * 32;
        const uint32_t a = dstA[p], left = lines_left[p], b = dstB[p];
        for (uint32_t i = 0; i < nl; i++) {
          const uint32_t d = i * 32 + l16 * 2;
          uint2 w;
          w.x = d < c_dw ? cy[d] : sorted[o_dw + d - c_dw];
          w.y = d + 1 < c_dw ? cy[d + 1] : sorted[o_dw + d + 1 - c_dw];
          const uint64_t line = i < left ? (uint64_t)a + i : (uint64_t)b + (i - left);
          if (!(pp.ablate & 1u)) *reinterpret_cast<uint2*>(sp.recs + line * 32 + l16 * 2) = w;
        }
        // the new carry = dwords [nl * 32, total) of the stream (16 lanes of one wave: the reads above happen before these writes)
        if (nl == 0) { for (uint32_t i = l16; i < r_dw; i += 16) carry[(size_t)p * 32 + c_dw + i] = sorted[o_dw + i]; }
        else { for (uint32_t i = l16; i < rem; i += 16) carry[(size_t)p * 32 + i] = sorted[o_dw + nl * 32 + i - c_dw]; }
        if (l16 == 0) carry_dw[p] = rem;
      }
    }
    // ---- the next round's rows, evaluated from the loads issued before the copy-out
    // (no barrier here: the next round touches cnt -- reset before B -- and, only after its own barriers A and B, off / sorted / carry)
    if (rd_next < nrounds) {                                 // uniform across the workgroup
      finish_round(rd_next, pre);
      if (kEarly) pre_early = issue_loads(rd_next + stride);
    }
  }
  __syncthreads();
  // tails: the carry dwords go behind the lines written so far; the fill of every partition's last chunk, in records
  for (uint32_t p = threadIdx.x; p < NP; p += blockDim.x) {
    uint32_t ch = cur_chunk[p], ln = cur_lines[p];
    const uint32_t rem = carry_dw[p];
    if (rem) {
      if (ln == cap_lines) { if (ch != kNoChunk) sp.chunk_fill[ch] = kP2ChunkRecs; ch = open_chunks(p, 1); ln = 0; }
      for (uint32_t i = 0; i < rem; i++) sp.recs[((uint64_t)ch * cap_lines + ln) * 32 + i] = carry[(size_t)p * 32 + i];
    }
    if (ch != kNoChunk) sp.chunk_fill[ch] = (ln * 32 + rem) / RW;
  }
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

// HOT = false: a build without the hot-key path (the planner found no heavy hitter: the common case); the lookup loop and the LDS accumulator
// updates of eight unrolled rows otherwise cost registers the tile needs
template <class P, int MODE, int TILES, int PACK, bool HOT = true>
__global__ __launch_bounds__(kP2MaxBlock) void part3_scatter_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {
  part3_scatter_body<P, MODE, TILES, PACK, HOT>(dsh, args, pp, sp);
}

}  // namespace k
}  // namespace plx
