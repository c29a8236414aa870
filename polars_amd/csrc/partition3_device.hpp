// partition3_device.hpp -- scatter pass of the partitioned high-cardinality group-by, third generation (kernels_partition.hip has the
// pipeline; the aggregation pass and the chunk bookkeeping are those of partition2_device.hpp).  Device-only header: compiled ahead of
// time for the benchmark shapes and at run time by hiprtc (jit.cpp) for any other program shape.
//
// The second-generation scatter appended every row to its partition's LDS ring and flushed complete lines after a barrier; measured
// (tools/micro_part3.hip, profiles/r03/micro_part3*.txt) its skeleton -- ring scan, two barriers per 2-4 K rows -- cost more than its
// bytes.  This one sorts a whole TILE of rows by partition inside the workgroup and writes whole lines only:
//
//   rank      one returning ds_add_u32 per row on the partition's counter: the row's rank within (tile, partition)
//   scan      one partition per thread (NP / 64 waves, one prefix sum for the tile offsets AND the chunk allocation): per partition, (carry + run)
//             dwords become a number of whole 128-B lines and their destination -- the rest of the partition's current chunk, then fresh chunks
//             (consecutive, from the workgroup's PRIVATE region: no counting pass, no global atomics) -- left behind as ONE 16-byte descriptor
//   sort      every row's packed record goes to its slot of the LDS tile
//   copy-out  a 16-lane group per partition reads the descriptor (one ds_read_b128), streams carry + run out as aligned 128-B lines (8 B per
//             lane, one two-dword LDS read per line) and leaves the dwords that do not fill a line in the partition's CARRY line for the next
//             round: partial lines never reach HBM (a partial line is a read-modify-write at the DRAM: 6.1 ms per 1e9 rows with them, 5.0 without)
// The per-round cost is the LDS instruction stream, not HBM (kernels_strgroup.hip has the measurements): the scan used to be four to sixteen
// partitions per lane of ONE wave with an LDS atomic per chunk opening, the copy-out seven dword reads of per-partition words and two reads per
// line -- tools/micro_part3.hip scatter3c vs scatter3d, 1e9 rows: 4-B records 4.45 -> 3.93 ms, 12-B 6.45 -> 5.53 (profiles/r03/micro_part3_descriptor_flush.txt).
// Four barriers per round of blockDim * 2 * TILES rows (8192 for the benchmark shapes).  Rows of hot keys (heavy hitters of the sample)
// never become records: they are aggregated in LDS accumulators on the spot, as before.
#pragma once
#include "partition2_device.hpp"

namespace plx {
namespace k {

// LDS of the scatter kernel (dynamic shared memory, 16-byte aligned), in this order:
//   desc [NP] uint4               per partition and round, written by the scan for the copy-out: x = (first dword of its rows in the tile) - (carried dwords),
//                                 y = carried dwords | new carry << 5 | whole lines << 10 | has rows << 24 | lines continue in newly opened chunks << 25, z = its first line
//   hot_k [hot_slots] u64, hot_acc [n_hot * n_aggs * copies] u64
//   sorted [T * RW] u32           the tile, sorted by partition (T = block * kRows * TILES rows)
//   carry  [NP][32] u32           dwords of a partition that do not fill a line yet
//   cnt, off[NP + 1], carry_dw, lines_left, dstB, cur_chunk, cur_lines [NP] u32 each; hot_i [hot_slots] u32; wtot [16] u32; misc [4] u32
__host__ __device__ inline size_t part3_scatter_lds(uint32_t T, uint32_t RW, uint32_t NP, uint32_t hot_slots, uint32_t n_hot, uint32_t n_aggs, uint32_t copies) {
  return (size_t)NP * 16 + (size_t)hot_slots * 8 + (size_t)n_hot * n_aggs * copies * 8 + (size_t)T * RW * 4 + (size_t)NP * 128 + ((size_t)NP * 7 + 1) * 4 + (size_t)hot_slots * 4 + 64 + 16;
}

// CHECK: the per-row test of narrowed values against their field (pp.check_src: bases from bounds the planner only assumed) is compiled in.  The ahead-of-time kernels
// exist in both forms -- some instantiations sit exactly at their 128-register budget, and the plain form is what runs once a scan has verified the bounds.
template <class P, int MODE, int TILES, int PACK, bool HOT = true, bool CHECK = true>
__device__ __forceinline__ void part3_scatter_body(const Shape dsh, const Args args, const PartPlan2 pp, const ScatterParams2 sp) {
  static_assert(P::kStatic, "the partitioned group-by runs specialised programs only (AOT or JIT)");
  extern __shared__ __attribute__((aligned(16))) unsigned long long p2_lds[];
  constexpr Shape sh = P::shape();
  constexpr RecLayout2 L = rec_layout2(P::shape(), (uint32_t)MODE, (uint32_t)PACK);
  constexpr uint32_t RW = L.rec_words;                  // dwords of a record of the stream (kPackPair: a PAIR of rows)
  constexpr bool PAIRV = PACK == (int)kPackPairV;      // pairs with the VALUE as a 48-bit offset (hash mode)
  constexpr bool PAIR = PACK == (int)kPackPair || PAIRV;
  constexpr uint32_t SW = scatter_row_words(L);         // dwords of a row in registers / of the tile's budget per row
  static_assert(!PAIR || (MODE == (int)kP2Direct ? (RW == 5 && SW == 3) : (RW == 7 && SW == 4)), "kPackPair: {slot | 64-bit key, 64-bit value} rows");
  static_assert(!PAIRV || MODE == (int)kP2Hash, "kPackPairV: hash partitions");
  constexpr uint32_t chunk_dw = kP2ChunkRecs * RW, cap_lines = chunk_dw / 32;      // a chunk holds whole records AND whole lines
  const uint32_t NP = 1u << pp.log2_parts;
  const uint32_t hot_slots = pp.n_hot ? (1u << pp.log2_hot_slots) : 0u;
  const uint32_t T = blockDim.x * kRows * TILES;
  uint4* desc = reinterpret_cast<uint4*>(p2_lds);
  unsigned long long* hot_k = p2_lds + (size_t)NP * 2;
  unsigned long long* hot_acc = hot_k + hot_slots;
  unsigned int* sorted = reinterpret_cast<unsigned int*>(hot_acc + (size_t)pp.n_hot * sh.n_aggs * pp.hot_copies);
  unsigned int* carry = sorted + (size_t)T * SW;       // (pairs: at most T / 2 + NP / 2 records of RW = 2 SW - 1 dwords <= SW T: the planner checks NP * RW <= T)
  unsigned int* cnt = carry + (size_t)NP * 32;
  unsigned int* off = cnt + NP;
  unsigned int* carry_dw = off + NP + 1;
  unsigned int* lines_left = carry_dw + NP;   // read only for a partition whose lines continue in newly opened chunks: lines that still fit the current chunk ...
  unsigned int* dstB = lines_left + NP;       // ... and where the rest goes (fresh consecutive chunks), in lines
  unsigned int* cur_chunk = dstB + NP;
  unsigned int* cur_lines = cur_chunk + NP;   // lines of the current chunk already written (cap_lines: none open / full)
  unsigned int* hot_i = cur_lines + NP;
  unsigned int* wtot = hot_i + hot_slots;     // [16] per scan wave: rows | chunks to open << 16
  unsigned int* misc = wtot + 16;             // [0] next chunk of this workgroup's region
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) { cnt[i] = 0; carry_dw[i] = 0; cur_chunk[i] = kNoChunk; cur_lines[i] = cap_lines; }
  for (uint32_t i = threadIdx.x; i < hot_slots; i += blockDim.x) { hot_k[i] = sp.hot_tbl_keys[i]; hot_i[i] = sp.hot_tbl_idx[i]; }
  for (uint32_t i = threadIdx.x; i < pp.n_hot * sh.n_aggs * pp.hot_copies; i += blockDim.x) hot_acc[i] = agg_identity_dev(sh.aggs[(i / pp.hot_copies) % sh.n_aggs].kind);
  if (threadIdx.x < 4) misc[threadIdx.x] = 0;
  if (threadIdx.x == 0 && (pp.tiles != (uint32_t)TILES || pp.pack != (uint32_t)PACK || pp.rec_words != RW || (PAIR && MODE == (int)kP2Direct && pp.key_shift > 15))) sp.flags[0] = 1u;     // host and kernel disagree about the geometry: fail the query
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
  uint64_t narrow_viol = 0;     // wave-uniform (scalar registers): lanes whose value did not fit its narrowed field (make_record2, pp.check_src)
  unsigned int rec[TILES][kRows][SW];
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
        make_record2<MODE, CHECK>(sh, L, pp, rf[t], r, row0 + r, rec[t][r], part[t][r], kvalid, key64, pass[r], narrow_viol);
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
  // The join probe's scatter (row-id records: key + row, a predicate column or two) has the room as well: 121 registers, no scratch, with the loads a round ahead.
  constexpr bool kEarly = RW == 1 || TILES <= 2 || (PACK == (int)kPackRowid && sh.n_inputs <= 2);
  bool pre_early = false;
  if (rd_first < nrounds) {
    finish_round(rd_first, issue_loads(rd_first));
    if (kEarly) pre_early = issue_loads(rd_first + stride);
  }
  for (int64_t rd = rd_first; rd < nrounds; rd += stride) {
    // ---- rank: one LDS atomic per surviving row (kept in the row's `part` word: partition in the low 10 bits, rank above them)
    p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < kRows; r++) if (part[t][r] != kNotPending) part[t][r] |= atomicAdd(&cnt[part[t][r]], 1u) << 10;
    });
    __syncthreads();                                                                  // A: the counts are complete
    // ---- scan, one partition per thread: tile offsets, lines and destinations of every partition
    uint32_t sc_c = 0, sc_cd = 0, sc_ln = 0, sc_ch = 0, sc_v = 0, sc_incl = 0, sc_opened = 0, sc_odd = 0;
    if (threadIdx.x < NP) {
      const uint32_t p = threadIdx.x;
      sc_c = cnt[p]; sc_odd = sc_c & 1u; sc_cd = carry_dw[p]; sc_ln = cur_lines[p]; sc_ch = cur_chunk[p]; sc_opened = misc[0];
      if (PAIR) sc_c = (sc_c + 1u) >> 1;                                               // records = pairs: an odd row closes its pair alone
      const uint32_t nl = (sc_cd + sc_c * RW) >> 5, left = cap_lines - sc_ln;          // left: 0 when no chunk is open (cur_lines == cap_lines)
      sc_v = sc_c | (nl > left ? (nl - left + cap_lines - 1) / cap_lines : 0u) << 16;   // records (<= T < 2^16 in all) and chunks to open, scanned together
      sc_incl = sc_v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)sc_incl, d, 64); if (lane >= d) sc_incl += o; }
      if (lane == 63) wtot[wave] = sc_incl;
    }
    __syncthreads();                                                                  // A2: the scan waves' totals
    if (threadIdx.x < NP) {
      const uint32_t p = threadIdx.x;
      uint32_t pre = 0, tot = 0;
      for (uint32_t w = 0; w < (NP >> 6); w++) { const uint32_t x = wtot[w]; if (w < (uint32_t)wave) pre += x; tot += x; }
      const uint32_t excl = pre + sc_incl - sc_v, o = excl & 0xffffu, c = sc_c;
      uint32_t local = sc_opened + (excl >> 16);                                       // this partition's first fresh chunk, within the workgroup's region
      if (p == NP - 1) { misc[0] = sc_opened + (tot >> 16); off[NP] = tot & 0xffffu; }
      if (sc_opened + (tot >> 16) > pp.chunks_per_wg) { if (p == 0) sp.flags[0] = 1u; local = 0; }   // cannot happen by construction; the query fails if it does
      const uint32_t cd = sc_cd, total = cd + c * RW, nl = total >> 5, rem = total & 31u, left = cap_lines - sc_ln;
      uint32_t ln = sc_ln, ch = sc_ch, y = cd | rem << 5 | nl << 10 | (c ? 1u << 24 : 0u), first_line = 0;
      off[p] = o; cnt[p] = 0; carry_dw[p] = rem;
      // (the previous round's copy-out is behind barrier A: the tile may be written) the second half of an odd partition's last pair is absent
      if (PAIR && sc_odd) {
        unsigned int* last = sorted + (size_t)(o + c - 1u) * RW;
        if (MODE == (int)kP2Direct) reinterpret_cast<unsigned short*>(last)[1] = (unsigned short)kPairAbsent;
        else if (PAIRV) reinterpret_cast<unsigned short*>(last + 6)[1] = (unsigned short)0xffffu;
        else { last[1] = 0xffffffffu; reinterpret_cast<unsigned short*>(last + 2)[1] = (unsigned short)0xffffu; }      // offset 2^48 - 1
      }
      if (nl) {
        first_line = ch * cap_lines + ln;
        if (nl > left) {
          const uint32_t extra = nl - left, need = (extra + cap_lines - 1) / cap_lines, first = chunk0 + local;
          if (ch != kNoChunk) sp.chunk_fill[ch] = kP2ChunkRecs;
          for (uint32_t e = 0; e < need; e++) { sp.chunk_part[first + e] = p; if (e + 1 < need) sp.chunk_fill[first + e] = kP2ChunkRecs; }
          if (left == 0) first_line = first * cap_lines;                                // everything goes to the new chunk(s), which are consecutive
          else { y |= 1u << 25; lines_left[p] = left; dstB[p] = first * cap_lines; }
          ch = first + need - 1; ln = extra - (need - 1) * cap_lines;
        } else ln += nl;
        cur_chunk[p] = ch; cur_lines[p] = ln;
      }
      desc[p] = make_uint4(o * RW - cd, y, first_line, 0u);
    }
    __syncthreads();                                                                  // B: offsets and destinations are known
    // ---- sort: every record to its slot of the tile
    p2_static_for<TILES>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        if constexpr (PAIRV) {      // (every lane takes part: narrow_viol is wave-uniform) a value outside the bounds its 48-bit offset assumed
          const uint64_t voff = ((uint64_t)rec[t][r][2] | ((uint64_t)rec[t][r][3] << 32)) - (uint64_t)pp.src_base[0];
          if (CHECK && pp.check_src) narrow_viol |= __ballot(part[t][r] != kNotPending && voff >= kPairVLimit);
        }
        if (part[t][r] == kNotPending) continue;
        if constexpr (PAIRV) {
          const uint32_t rk = part[t][r] >> 10, h = rk & 1u;
          unsigned int* dst = sorted + (size_t)(off[part[t][r] & 1023u] + (rk >> 1)) * RW;
          const uint64_t voff = ((uint64_t)rec[t][r][2] | ((uint64_t)rec[t][r][3] << 32)) - (uint64_t)pp.src_base[0];
          dst[2 * h] = rec[t][r][0]; dst[2 * h + 1] = rec[t][r][1];
          dst[4 + h] = (uint32_t)voff; reinterpret_cast<unsigned short*>(dst + 6)[h] = (unsigned short)(voff >> 32);      // (a violating value is reported, the query runs again: never the marker)
          continue;
        }
        if constexpr (PAIR) {
          const uint32_t rk = part[t][r] >> 10, h = rk & 1u;                           // rank r of the partition's rows in the tile -> pair r / 2, half r % 2
          unsigned int* dst = sorted + (size_t)(off[part[t][r] & 1023u] + (rk >> 1)) * RW;
          if constexpr (MODE == (int)kP2Direct) {
            reinterpret_cast<unsigned short*>(dst)[h] = (unsigned short)rec[t][r][0];
            dst[1 + 2 * h] = rec[t][r][1]; dst[2 + 2 * h] = rec[t][r][2];
          } else {
            const uint64_t koff = ((uint64_t)rec[t][r][0] | ((uint64_t)rec[t][r][1] << 32)) - (uint64_t)pp.key_base;      // < 2^48 - 1: the key range is known (plan3)
            dst[h] = (uint32_t)koff; reinterpret_cast<unsigned short*>(dst + 2)[h] = (unsigned short)(koff >> 32);
            dst[3 + 2 * h] = rec[t][r][2]; dst[4 + 2 * h] = rec[t][r][3];
          }
          continue;
        }
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
      const uint32_t g = threadIdx.x >> 4, l16 = threadIdx.x & 15u, d = l16 * 2;
      for (uint32_t p = g; p < NP; p += blockDim.x >> 4) {
        const uint4 D = desc[p];
        const int s0 = (int)D.x;                                                       // dword i of the partition's stream (carry first) = sorted[s0 + i] for i >= c_dw
        const uint32_t y = D.y, c_dw = y & 31u, rem = (y >> 5) & 31u, nl = (y >> 10) & 0x3fffu;
        uint32_t left = 0xffffffffu, b = 0;
        if ((y >> 25) & 1u) { left = lines_left[p]; b = dstB[p]; }
        if (nl) {
          // the first line may start in the carry; a lane's two dwords come from ONE base (a two-dword read), and the one lane whose pair straddles the
          // end of the carry takes its second dword from the tile
          const unsigned int* src = d < c_dw ? carry + (size_t)p * 32 + d : sorted + (s0 + (int)d);
          uint2 w = make_uint2(src[0], src[1]);
          if (d < c_dw && d + 1 >= c_dw) w.y = sorted[s0 + (int)d + 1];
          if (!(pp.ablate & 1u)) *reinterpret_cast<uint2*>(sp.recs + (uint64_t)D.z * 32 + d) = w;
          for (uint32_t i = 1; i < nl; i++) {
            const unsigned int* s2 = sorted + (s0 + (int)(i * 32 + d));
            const uint2 w2 = make_uint2(s2[0], s2[1]);
            const uint64_t line = i < left ? (uint64_t)D.z + i : (uint64_t)b + (i - left);
            if (!(pp.ablate & 1u)) *reinterpret_cast<uint2*>(sp.recs + line * 32 + d) = w2;
          }
        }
        // the new carry = dwords [nl * 32, nl * 32 + rem) of the stream; without a whole line the old carry stays and the run is appended
        // (16 lanes of one wave: the reads above happen before these writes)
        if ((y >> 24) & 1u) {
          const unsigned int* s3 = sorted + (s0 + (int)(nl * 32 + d));
          const uint32_t r0 = s3[0], r1 = s3[1], lo = nl ? 0u : c_dw;
          if (d >= lo && d < rem) carry[(size_t)p * 32 + d] = r0;
          if (d + 1 >= lo && d + 1 < rem) carry[(size_t)p * 32 + d + 1] = r1;
        }
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
  if (CHECK && narrow_viol && lane == 0) sp.flags[2] = 1u;     // a value outside the bounds its narrowing assumed: the query is planned again (engine.cpp)
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
template <class P, int MODE, int TILES, int PACK, bool HOT = true, bool CHECK = false>
__global__ __launch_bounds__(kP2MaxBlock) void part3_scatter_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {
  part3_scatter_body<P, MODE, TILES, PACK, HOT, CHECK>(dsh, args, pp, sp);
}

}  // namespace k
}  // namespace plx
