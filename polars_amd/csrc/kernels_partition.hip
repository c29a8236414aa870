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
//                    flushed B at a time (a whole 128-B line for 16-B records); pass 1's per-workgroup histogram
//                    fixes every workgroup's write offsets in advance: no global atomics, deterministic output
//   3. part_agg      one workgroup per partition: LDS open-addressing table (CAS on the key word, LDS atomics
//                    on the cells); when the partition is done its groups are written straight to the dense
//                    output arrays (one global atomic per partition) -- no table in HBM, no compaction pass
// HBM traffic: (1) reads the key/predicate columns, (2) reads all inputs + writes the records, (3) reads the
// records: ~3x the algorithmic bytes instead of a fraction of the atomic rate.
#include "partition_device.hpp"
#include "partition3_device.hpp"
#include <algorithm>
#include <vector>
#include "kernels.hpp"
#include "kernels_fused.hpp"
#include "jit.hpp"
#include "scan.hpp"

namespace plx {
namespace k {

using namespace dev;
using namespace fused;

// ---- host side -----------------------------------------------------------------------------------------
static uint64_t scan_bytes(const Shape& sh, const Args& args) {
  uint64_t b = 0;
  for (int i = 0; i < sh.n_inputs; i++) b += (uint64_t)args.n_rows * dtype_width(sh.in_dtype[i]) + (args.in[i].validity ? (uint64_t)args.n_rows / 8 : 0);
  return b;
}

// Tuning knobs for experiments (unset = the defaults below): PLX_PART_LOG2_PARTS (6..10), PLX_PART_BUF_ROWS (even, 2..16),
// PLX_PART_WGS_PER_CU (1 | 2).  Read once per process.
static int env_int(const char* name, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !*e) return -1;
  const int v = atoi(e);
  return (v >= lo && v <= hi) ? v : -1;
}
static const int kEnvLog2Parts = env_int("PLX_PART_LOG2_PARTS", 6, 10);
static const int kEnvBufRows = env_int("PLX_PART_BUF_ROWS", 2, 16);
static const int kEnvWgsPerCu = env_int("PLX_PART_WGS_PER_CU", 1, 2);

bool partition_plan(const Shape& sh, double est_groups, bool any_nullable, PartitionPlan* out) {
  PartitionPlan pp{};
  pp.rec = rec_layout(sh);     // the layout AOT / JIT kernels derive at compile time from the same shape
  (void)any_nullable;          // rec_layout decides from the shape (nullable inputs, ops that can yield null)
  if (pp.rec.n_src > (uint32_t)kMaxSrc) return false;
  // LDS table of pass 3: as many slots as fit ~144 KB; partitions so that a partition holds <= slots / 2 groups
  const size_t lds_budget = 144 * 1024;
  uint32_t log2_slots = 14;
  while (log2_slots > 8 && ((size_t)(1u << log2_slots) + 2) * 8 * (1 + sh.n_aggs) > lds_budget) log2_slots--;
  if (((size_t)(1u << log2_slots) + 2) * 8 * (1 + sh.n_aggs) > lds_budget) return false;
  const double per_part = (double)(1u << log2_slots) * 0.62;   // expected groups per partition (the caller passes 1.3 x its estimate): LDS table load <= ~0.62
  uint32_t log2_parts = 6;
  while (log2_parts < 10 && (double)(1u << log2_parts) * per_part < est_groups) log2_parts++;
  if ((double)(1u << log2_parts) * per_part < est_groups) return false;   // would need > 1024 partitions
  if (kEnvLog2Parts > (int)log2_parts) log2_parts = (uint32_t)kEnvLog2Parts;   // only more partitions than needed: fewer would overflow the LDS tables
  pp.log2_parts = log2_parts;
  pp.log2_slots = log2_slots;
  // write-combining buffers of pass 2
  uint32_t B = 8;
  auto scatter_lds = [&](uint32_t b) { return ((size_t)(1u << log2_parts) * b * pp.rec.rec_words + 2 * (1u << log2_parts)) * 8 + (size_t)(1u << log2_parts) * 8 + 16; };
  if (kEnvBufRows > 0) B = (uint32_t)(kEnvBufRows & ~1);
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
  const size_t slds = ((size_t)NP * pp.buf_rows * pp.rec.rec_words + 2 * NP) * 8 + (size_t)NP * 8 + 16;
  const bool is_static = static_id == SHAPE_GB_SUM_CNT_I64 || static_id == SHAPE_GB_SUM_MEAN_U32_F64;   // the cases of PLX_PART_STATIC_CASES
  // no AOT kernels for this shape: specialise all three passes at run time (all or nothing -- passes 1 and 2 must share
  // the round geometry, which differs between specialised programs and the generic interpreter)
  const bool use_jit = !is_static && jit::ensure(sh, jit::PART_COUNT, args.n_rows) && jit::ensure(sh, jit::PART_SCATTER, args.n_rows) && jit::ensure(sh, jit::PART_AGG, args.n_rows);
  const int64_t rows_per_round = (int64_t)kBlock * kRows * ((is_static || use_jit) ? kStaticRoundTiles : 1);
  const int64_t nrounds = (args.n_rows + rows_per_round - 1) / rows_per_round;
  const int wgs_per_cu = slds > 80 * 1024 ? 1 : (kEnvWgsPerCu > 0 ? kEnvWgsPerCu : 2);
  const int sgrid = (int)std::min<int64_t>(nrounds, (int64_t)device().cu_count * wgs_per_cu);   // 2 workgroups per CU when two buffers sets fit the 160 KB LDS; the SAME grid for pass 1 and pass 2
  Buf hist = dev_alloc(sizeof(uint32_t) * (size_t)sgrid * NP);
  Buf wg_prefix = dev_alloc(sizeof(uint64_t) * (size_t)sgrid * NP);
  Buf totals = dev_alloc(sizeof(uint64_t) * (NP + 1));
  Buf part_off = dev_alloc(sizeof(uint64_t) * (NP + 2));
  {
    ProfileScope ps("part_count", scan_bytes(sh, args), (uint64_t)args.n_rows);
    const size_t lds = sizeof(unsigned int) * NP;
    switch (static_id) {
      PLX_PART_STATIC_CASES(part_count_kernel, dim3(sgrid), dim3(kBlock), lds, stream(), sh, args, pp.log2_parts, hist->as<unsigned int>())
      default:
        if (use_jit) {
          Shape shc = sh; Args ac = args; uint32_t lp = pp.log2_parts; unsigned int* hp = hist->as<unsigned int>();
          void* ka[] = {&shc, &ac, &lp, &hp};
          PLX_REQUIRE(jit::launch_raw(sh, jit::PART_COUNT, ka, sgrid, kBlock, lds), PLX_ERR_HIP, "jit launch failed (part_count)");
        } else { const DynLaunch d = dyn_launch(sh, args, lds); hipLaunchKernelGGL((part_count_kernel<DynProg>), dim3(sgrid), dim3(kBlock), d.lds, stream(), sh, d.args, pp.log2_parts, hist->as<unsigned int>()); }
        break;
    }
    PLX_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(part_prefix_kernel, dim3((NP + kBlock - 1) / kBlock), dim3(kBlock), 0, stream(), hist->as<unsigned int>(), sgrid, NP, wg_prefix->as<unsigned long long>(),
                     totals->as<unsigned long long>());
  PLX_HIP(hipGetLastError());
  exclusive_scan_u64(totals->as<uint64_t>(), part_off->as<uint64_t>(), NP);   // writes NP + 1 entries
  uint64_t total = 0;
  d2h_sync(&total, part_off->as<uint64_t>() + NP, 8);
  if (total == 0) { *out_keys = dev_alloc(8); *out_kvalid = dev_alloc(8); *out_acc = dev_alloc(8); return 0; }
  Buf recs = dev_alloc(sizeof(uint64_t) * (size_t)total * pp.rec.rec_words + 64);
  {
    ProfileScope ps("part_scatter", scan_bytes(sh, args) + total * pp.rec.rec_words * 8, (uint64_t)args.n_rows);
    switch (static_id) {
      PLX_PART_STATIC_CASES(part_scatter_kernel, dim3(sgrid), dim3(kBlock), slds, stream(), sh, args, pp, part_off->as<unsigned long long>(), wg_prefix->as<unsigned long long>(), recs->as<unsigned long long>())
      default:
        if (use_jit) {
          Shape shc = sh; Args ac = args; PartitionPlan ppc = pp;
          const unsigned long long* po = part_off->as<unsigned long long>(); const unsigned long long* wp = wg_prefix->as<unsigned long long>(); unsigned long long* ro = recs->as<unsigned long long>();
          void* ka[] = {&shc, &ac, &ppc, &po, &wp, &ro};
          PLX_REQUIRE(jit::launch_raw(sh, jit::PART_SCATTER, ka, sgrid, kBlock, slds), PLX_ERR_HIP, "jit launch failed (part_scatter)");
        } else { const DynLaunch d = dyn_launch(sh, args, slds); hipLaunchKernelGGL((part_scatter_kernel<DynProg>), dim3(sgrid), dim3(kBlock), d.lds, stream(), sh, d.args, pp, part_off->as<unsigned long long>(), wg_prefix->as<unsigned long long>(), recs->as<unsigned long long>()); }
        break;
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
    ProfileScope ps("part_agg_lds", total * pp.rec.rec_words * 8, total);
    const size_t lds = (((size_t)1 << pp.log2_slots) + 2) * 8 * (1 + sh.n_aggs);
    const int agrid = (int)std::min<uint32_t>(NP, (uint32_t)device().cu_count);
    switch (static_id) {
      PLX_PART_STATIC_CASES(part_agg_kernel, dim3(agrid), dim3(kAggBlock), lds, stream(), sh, pp, ap)
      default:
        if (use_jit) {
          Shape shc = sh; PartitionPlan ppc = pp; PartAggParams apc = ap;
          void* ka[] = {&shc, &ppc, &apc};
          PLX_REQUIRE(jit::launch_raw(sh, jit::PART_AGG, ka, agrid, kAggBlock, lds), PLX_ERR_HIP, "jit launch failed (part_agg)");
        } else hipLaunchKernelGGL((part_agg_kernel<DynProg>), dim3(agrid), dim3(kAggBlock), lds, stream(), sh, pp, ap);
        break;
    }
    PLX_HIP(hipGetLastError());
  }
  uint64_t res[2] = {0, 0};
  d2h_sync(res, ctr->ptr, 16);
  if ((uint32_t)res[1]) return -1;
  if (desc) *desc = "partitioned(P=" + std::to_string(NP) + ",rec=" + std::to_string(pp.rec.rec_words * 8) + "B,buf=" + std::to_string(pp.buf_rows) + ")+lds_hash_table(slots=" + std::to_string(1u << pp.log2_slots) + ")";
  return (int64_t)res[0];
}


// ======================================================================================================================
// Second generation (partition2_device.hpp): no counting pass, packed records, chunked private output regions, hot-key
// pre-aggregation, hash or direct-address LDS tables.  HBM traffic: inputs read ONCE + records written once + records
// read once (e.g. 16 + 12 + 12 B/row for an i64 key column of dense ids and an i64 value, against 8 + 32 + 16 before).
// ======================================================================================================================
static const int kEnvP2Block = env_int("PLX_PART_BLOCK", 256, 1024);
static const int kEnvP2Ring = env_int("PLX_PART_RING_LINES", 1, 16);
static const int kEnvP2Direct = env_int("PLX_PART_DIRECT", 0, 1);        // 0: never use the direct-address mode
static const int kEnvP2DirectLp = env_int("PLX_PART_DIRECT_LOG2_PARTS", 4, 9);   // direct mode: partitions when the id range allows (default 2^9)
static const int kEnvP3Block = env_int("PLX_P3_BLOCK", 0, 1024);         // measurement: threads of a gen-3 scatter workgroup (0 = 1024; a multiple of 64, >= the partition count)
static const int kEnvP2Wgs = env_int("PLX_PART2_WGS_PER_CU", 1, 4);       // scatter workgroups per CU (they must fit the LDS together)
static const int kEnvP2Ablate = env_int("PLX_PART_ABLATE", 0, 15);     // measurement only, results WRONG: 1 / 2 scatter (fused.hpp PartPlan2::ablate), 4 = wide-key aggregation without the cell updates, 8 = without the slot search
static const int kEnvP2Tiles = env_int("PLX_PART_TILES", 1, 4);           // tiles per wave and round (gen 2: default 2 when the rings absorb them; gen 3: 4 / 2 / 1 by LDS)
static const int kEnvP2Gen = env_int("PLX_PART_GEN", 2, 3);               // scatter generation (default 3: tile sort + carry lines; 2: rings + line flush)

static uint32_t floor_pow2(uint32_t x) { uint32_t p = 1; while (p * 2 <= x) p *= 2; return p; }
static uint32_t ceil_log2(uint64_t x) { uint32_t b = 0; while ((1ull << b) < x) b++; return b; }

// grid and chunk regions of the scatter pass for `tiles` tiles per wave and round
static void plan2_geometry(PartPlan2& pp, int64_t n_rows, uint32_t tiles) {
  pp.tiles = tiles;
  const int64_t rows_per_round = (int64_t)pp.block * kRows * tiles;
  const int64_t nrounds = (n_rows + rows_per_round - 1) / rows_per_round;
  pp.scatter_grid = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(nrounds, (int64_t)device().cu_count * std::max(1, kEnvP2Wgs)));
  const int64_t rounds_per_wg = (nrounds + pp.scatter_grid - 1) / pp.scatter_grid;
  // records a round leaves at most (kPackPair: two rows a record, every partition may close one pair alone)
  const int64_t recs_per_round = (pp.pack == kPackPair || pp.pack == kPackPairV) ? rows_per_round / 2 + (int64_t)(1u << pp.log2_parts) / 2 + 1 : rows_per_round;
  pp.chunks_per_wg = (uint32_t)(rounds_per_wg * recs_per_round / kP2ChunkRecs + (1u << pp.log2_parts) + 2);
}

static uint32_t bits_for(uint64_t span) { uint32_t b = 0; while (b < 64 && (span >> b)) b++; return b; }     // bits that hold 0..span

// third generation: packing from the value ranges, tile size from the LDS budget.  false: this shape / geometry stays on generation 2
static bool plan3(const Shape& sh, PartPlan2& pp, int64_t n_rows, const SrcRange* ranges, size_t lds_total, const SrcRange* key_range) {
  const uint32_t NP = 1u << pp.log2_parts;
  if (NP < 64 || NP > (uint32_t)kP2MaxBlock) return false;                    // the scan takes one partition per thread, in whole waves
  uint32_t pack = kPackNone;
  const char* pe = getenv("PLX_PART_PACK");
  const uint32_t pack_cap = pe ? (uint32_t)std::max(0, std::min(2, atoi(pe))) : 2u;
  if (ranges && pack_cap > 0) {
    const RecLayout2 Ln = rec_layout2(sh, pp.mode, kPackNarrow);
    bool narrow_ok = false, all_ok = true;
    for (uint32_t j = 0; j < Ln.n_src && j < (uint32_t)kMaxSrc; j++) {
      if (Ln.src_kind[j] != 3) continue;
      if (ranges[j].known && ranges[j].mx >= ranges[j].mn && (uint64_t)ranges[j].mx - (uint64_t)ranges[j].mn < 0xffffffffull) narrow_ok = true; else all_ok = false;
    }
    if (narrow_ok && all_ok) pack = kPackNarrow;
    if (pack_cap >= 2 && best_static_pack(sh, pp.mode) == kPackFused && ranges[0].known && ranges[0].mx >= ranges[0].mn &&
        pp.key_shift + bits_for((uint64_t)ranges[0].mx - (uint64_t)ranges[0].mn) <= 32) pack = kPackFused;
  }
  // a 64-bit value that does not narrow (f64) over direct-address slots: two rows a record, 10 bytes a row (fused.hpp kPackPair; PLX_PART_PAIR=0: measurement)
  static const bool pair_off = getenv("PLX_PART_PAIR") && getenv("PLX_PART_PAIR")[0] == '0';
  // (the tile must hold the lone halves too: checked below)
  if (pack == kPackNone && !pair_off && pair_pack_ok(sh, pp.mode)) {
    if (pp.mode == kP2Direct) { if (pp.key_shift <= 15) pack = kPackPair; }
    else if (key_range && key_range->known && !key_range->check && key_range->mx >= key_range->mn && (uint64_t)key_range->mx - (uint64_t)key_range->mn < kPairAbsent48) {
      pack = kPackPair; pp.key_base = key_range->mn;      // hash partitions of 64-bit keys whose exact range spans < 2^48 - 1: 48-bit offsets, 14 bytes a row
    }
  }
  // ... or, with a key of any range, an Int64 value column whose range spans < 2^48 - 2^32 as a 48-bit offset (kPackPairV; assumed bounds are checked per row)
  bool pairv = false;
  if (pack == kPackNone && !pair_off && pairv_pack_ok(sh, pp.mode) && ranges && ranges[0].known && ranges[0].mx >= ranges[0].mn && (uint64_t)ranges[0].mx - (uint64_t)ranges[0].mn < kPairVLimit) { pack = kPackPairV; pairv = true; }
  RecLayout2 L = rec_layout2(sh, pp.mode, pack);
  if (L.n_src > (uint32_t)kMaxSrc || L.rec_words > 13) return false;
  for (int j = 0; j < kMaxSrc; j++) pp.src_base[j] = (pack != kPackNone && ranges && ranges[j].known && (L.src_kind[j] == 3 || pack == kPackFused)) ? ranges[j].mn : 0;
  pp.check_src = 0;
  for (uint32_t j = 0; j < L.n_src && j < (uint32_t)kMaxSrc; j++) if (pack != kPackNone && ranges && ranges[j].known && ranges[j].check && (L.src_kind[j] == 3 || pack == kPackFused)) pp.check_src = 1;
  if (pairv) { pp.src_base[0] = ranges[0].mn; pp.check_src = ranges[0].check ? 1u : 0u; }
  const uint32_t hot_slots = pp.n_hot ? (1u << pp.log2_hot_slots) : 0;
  uint32_t tiles = 0;
  uint32_t block = kEnvP3Block >= 64 && (uint32_t)kEnvP3Block >= NP && kEnvP3Block % 64 == 0 ? (uint32_t)kEnvP3Block : (uint32_t)kP2MaxBlock;
  for (uint32_t t : {4u, 3u, 2u, 1u}) {
    if (kEnvP2Tiles > 0 && t > (uint32_t)kEnvP2Tiles) continue;
    if ((pp.mode == kP2Hash || pp.n_hot) && t > 3) continue;      // hash partitions keep the 64-bit key and its hash live, the hot-key path its lookups: four tiles spill (12-24 B / lane; a scratch reload waits for every load in flight)
    if (part3_scatter_lds(block * kRows * t, scatter_row_words(L), NP, hot_slots, pp.n_hot, sh.n_aggs, pp.hot_copies) <= lds_total) { tiles = t; break; }
  }
  if (!tiles) return false;
  // The per-round cost of the scatter is per PARTITION, so what counts is rows per round.  When the LDS holds only one or two tiles of a full 1024-thread workgroup
  // (wide records at 512 partitions: 88 KB of carry lines and bookkeeping before the tile), a SMALLER workgroup with one tile more stages more rows: 896 threads x 2 tiles
  // of 20-byte records = 3584 rows a round instead of 2048 (measured at 1e9 rows of a two-column key, round 5: 12.9 -> 10.6 ms at 832 threads; 704: 12.1, 768: 11.2;
  // 512 threads x 3 tiles: 19.0 -- the tile sort wants the threads).  At least 768 threads (and one per partition).
  if (!(kEnvP3Block >= 64) && !(kEnvP2Tiles > 0) && tiles < ((pp.mode == kP2Hash || pp.n_hot) ? 3u : 4u)) {
    for (uint32_t b = (uint32_t)kP2MaxBlock - 64; b >= 768 && b >= NP; b -= 64) {
      if (b * (tiles + 1) <= block * tiles) break;                       // no more rows a round than what we have
      if (part3_scatter_lds(b * kRows * (tiles + 1), scatter_row_words(L), NP, hot_slots, pp.n_hot, sh.n_aggs, pp.hot_copies) <= lds_total) { block = b; tiles = tiles + 1; break; }
    }
  }
  // (pairs need room for the partitions' lone halves in the tile: at most T / 2 + NP / 2 five-dword records in 3 T dwords)
  if ((pack == kPackPair || pack == kPackPairV) && NP * L.rec_words > block * kRows * tiles) { pack = kPackNone; L = rec_layout2(sh, pp.mode, pack); if (pp.mode == kP2Hash) pp.key_base = 0; pp.src_base[0] = 0; pp.check_src = 0; }
  pp.gen = 3; pp.pack = pack; pp.rec_words = L.rec_words; pp.block = block; pp.ring_lines = 0;
  plan2_geometry(pp, n_rows, tiles);
  // chunks are filled completely (the carry line keeps the remainder): whole chunks of the rows + one partial chunk per partition + slack
  return true;
}

bool partition_plan2(const Shape& sh, double est_groups, int packed_bits, int len_idx, int64_t n_rows, int n_hot, PartPlan2* out, const SrcRange* src_ranges, const SrcRange* key_range) {
  PartPlan2 pp{};
  pp.gen = 2;
  if (sh.key == kNone && !sh.n_keys) return false;
  const bool wide = sh.n_keys != 0;                // a multi-column key compared word by word: hash partitions only, no hot keys
  if (wide && n_hot) return false;
  const size_t lds_total = 160 * 1024 - 2048;      // leave room for the kernels' static shared variables
  pp.len_idx = (uint32_t)(len_idx < 0 ? 0 : len_idx);
  // ---- table geometry
  bool direct = false;
  if (!wide && packed_bits > 0 && packed_bits <= 25 && kEnvP2Direct != 0 && len_idx >= 0) {
    // direct-address LDS table: slots = low bits of the id; as many partitions as give every CU work, tables as large as fit
    uint32_t max_shift = 0;
    while (((size_t)1 << (max_shift + 1)) * sh.n_aggs * 8 <= 128 * 1024) max_shift++;
    // 256 partitions when the id range allows: one aggregation workgroup per CU (three chunks in flight per wave keep it fed) and
    // rings of four lines, which absorb two tiles per scatter round
    int shift = packed_bits - (kEnvP2DirectLp > 0 ? kEnvP2DirectLp : 8);
    if (shift > (int)max_shift) shift = (int)max_shift;
    if (shift < 6) shift = 6;
    const int lp = packed_bits - shift;
    if (lp >= 4 && lp <= 9) {
      direct = true; pp.mode = kP2Direct; pp.key_shift = (uint32_t)shift; pp.log2_slots = (uint32_t)shift; pp.log2_parts = (uint32_t)lp;
      static const bool no_interleave = getenv("PLX_PART_INTERLEAVE") && getenv("PLX_PART_INTERLEAVE")[0] == '0';
      pp.interleave = no_interleave ? 0u : 1u;
    }
  }
  if (!direct) {
    pp.mode = kP2Hash;
    // LDS open-addressing table of a partition: keys + cells, ANY number of slots (slot = mulhi(hash32, n_slots)), as many as 144 KB hold.  Fewer partitions
    // make the scatter faster (its per-round cost is per partition: 256 partitions of 16-byte records move two lines per partition and round, 512 one), so the
    // smallest partition count whose tables stay at or below a load of 0.85 x the caller's padded estimate (it passes 1.3 x its own: a real load of ~0.65).
    pp.wide_null_word = wide && shape_may_have_nulls(sh) ? 1u : 0u;
    // wide: group storage (key words (+ null mask) + cells) for n_slots groups, numbered in order of appearance, behind a tag table of FOUR 32-bit entries per group
    // of capacity (buckets of eight; ordinals are 12 bits: at most 4096 groups a partition)
    const size_t slot_bytes = wide ? 16 + 8 * ((size_t)sh.n_keys + pp.wide_null_word + (size_t)sh.n_aggs) : 8 * (1 + (size_t)sh.n_aggs);
    const uint32_t n_slots = wide ? (uint32_t)std::min<size_t>(((144 * 1024) / slot_bytes) & ~(size_t)1, (size_t)4096) : (uint32_t)std::min<size_t>((144 * 1024) / slot_bytes - 2, (size_t)1 << 14);
    pp.n_tags = wide ? n_slots * 4 : 0;
    if (n_slots < 256) return false;
    const double per_part = (double)n_slots * 0.85;
    uint32_t lp = 6;
    while (lp < 9 && (double)(1u << lp) * per_part < est_groups) lp++;
    // more than 512 partitions: the carry lines would not fit the LDS.  A table that does fill up is reported by the aggregation pass and the caller falls back
    if ((double)(1u << lp) * per_part < est_groups && (double)(1u << lp) * (double)n_slots * 0.95 < est_groups) return false;
    if (kEnvLog2Parts > (int)lp && kEnvLog2Parts <= 9) lp = (uint32_t)kEnvLog2Parts;
    pp.log2_parts = lp; pp.n_slots = n_slots; pp.log2_slots = 0; pp.key_shift = 0;
  }
  const RecLayout2 L = rec_layout2(sh, pp.mode);
  if (L.n_src > (uint32_t)kMaxSrc || L.rec_words > 13) return false;
  pp.rec_words = L.rec_words;
  const uint32_t NP = 1u << pp.log2_parts;
  // ---- hot keys
  pp.n_hot = (uint32_t)std::min<int>(n_hot, (int)kP2MaxHot);
  pp.log2_hot_slots = pp.n_hot ? std::max<uint32_t>(3, ceil_log2((uint64_t)pp.n_hot * 2)) : 0;
  pp.hot_copies = 1;
  if (pp.n_hot) { uint32_t c = 16; while (c > 1 && (size_t)pp.n_hot * sh.n_aggs * 8 * c > 8 * 1024) c >>= 1; pp.hot_copies = c; }
  pp.ablate = kEnvP2Ablate > 0 ? (uint32_t)kEnvP2Ablate : 0u;
  if (kEnvP2Gen != 2 && plan3(sh, pp, n_rows, src_ranges, lds_total, key_range)) { *out = pp; return true; }
  // ---- generation 2.  rings: as many 128-B lines per partition as the LDS holds (power of two)
  const uint32_t hot_slots = pp.n_hot ? (1u << pp.log2_hot_slots) : 0;
  const size_t fixed = part2_scatter_lds(NP, 0, hot_slots, pp.n_hot, sh.n_aggs, pp.hot_copies);
  if (fixed + (size_t)NP * 256 > lds_total) return false;
  uint32_t lines = floor_pow2((uint32_t)((lds_total - fixed) / ((size_t)NP * 128)));
  if (lines > 16) lines = 16;
  if (kEnvP2Ring > 0 && (uint32_t)kEnvP2Ring <= lines) lines = floor_pow2((uint32_t)kEnvP2Ring);
  if (lines < 2) return false;
  pp.ring_lines = lines;
  // ---- workgroup size: rows per round so that a partition receives well under its ring's free space per round
  const double cap_recs = (double)(lines * 32) / L.rec_words, line_recs = 32.0 / L.rec_words;
  const double lambda_max = std::max(1.0, (cap_recs - line_recs) * 0.55);
  uint32_t block = kP2MaxBlock;
  while (block > 256 && (double)block * kRows / NP > lambda_max) block >>= 1;
  if (kEnvP2Block > 0) block = floor_pow2((uint32_t)kEnvP2Block);
  pp.block = block;
  pp.ablate = kEnvP2Ablate > 0 ? (uint32_t)kEnvP2Ablate : 0u;
  // two tiles per round when the rings absorb them (same arrival-rate rule): half the barriers and ring scans per row
  uint32_t tiles = ((double)block * kRows * 2 / NP <= lambda_max) ? 2u : 1u;
  if (kEnvP2Tiles > 0) tiles = kEnvP2Tiles >= 2 ? 2u : 1u;
  plan2_geometry(pp, n_rows, tiles);
  *out = pp;
  return true;
}

// ---- hot keys: heavy hitters of a sample aggregation table --------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void hot_candidates_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ acc, int64_t cap, int n_aggs,
                                                                int len_idx, unsigned long long threshold, unsigned int* __restrict__ counter, unsigned long long* __restrict__ out /* [cap_out][2] */,
                                                                unsigned int cap_out) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = keys[s];
    if (k == kEmptyKey) continue;
    const unsigned long long c = acc[(size_t)s * n_aggs + len_idx];
    if (c < threshold) continue;
    const unsigned int i = atomicAdd(counter, 1u);
    if (i < cap_out) { out[(size_t)i * 2] = k; out[(size_t)i * 2 + 1] = c; }
  }
}
// keys of the sample table `t` (regular slots only: never the null key or the EMPTY-pattern key) whose row count is at least
// `threshold`, heaviest first, at most kP2MaxHot.  Synchronises.
void select_hot_keys(const HashTable& t, int n_aggs, int len_idx, uint64_t threshold, std::vector<uint64_t>* out, uint64_t* hot_rows) {
  out->clear();
  if (hot_rows) *hot_rows = 0;
  if (len_idx < 0) return;
  const unsigned int cap_out = 2048;
  Buf cand = dev_alloc(sizeof(uint64_t) * 2 * cap_out);
  Buf ctr = dev_alloc_zero(8);
  const int64_t cap = (int64_t)1 << t.log2_cap;
  hipLaunchKernelGGL(hot_candidates_kernel, dim3(grid_for(cap, kBlock * 4)), dim3(kBlock), 0, stream(), t.keys, t.acc, cap, n_aggs, len_idx, (unsigned long long)threshold,
                     ctr->as<unsigned int>(), cand->as<unsigned long long>(), cap_out);
  PLX_HIP(hipGetLastError());
  uint32_t n = 0;
  d2h_sync(&n, ctr->ptr, 4);
  n = std::min<uint32_t>(n, cap_out);
  if (!n) return;
  std::vector<uint64_t> host((size_t)n * 2);
  d2h_sync(host.data(), cand->ptr, host.size() * 8);
  std::vector<std::pair<uint64_t, uint64_t>> v;   // (count, key)
  for (uint32_t i = 0; i < n; i++) v.emplace_back(host[(size_t)i * 2 + 1], host[(size_t)i * 2]);
  std::sort(v.begin(), v.end(), [](const std::pair<uint64_t, uint64_t>& a, const std::pair<uint64_t, uint64_t>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
  for (size_t i = 0; i < v.size() && i < kP2MaxHot; i++) { out->push_back(v[i].second); if (hot_rows) *hot_rows += v[i].first; }
}

// ---- chunk -> partition map -> per-partition chunk lists (counting sort) ----------------------------------------------
__global__ __launch_bounds__(kBlock) void chunk_hist_kernel(const unsigned int* __restrict__ chunk_part, int64_t n_chunks, uint32_t NP, unsigned int* __restrict__ counts) {
  extern __shared__ unsigned long long lds_raw[];
  unsigned int* h = reinterpret_cast<unsigned int*>(lds_raw);
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
    const unsigned int p = chunk_part[c];
    if (p != kNoChunk) atomicAdd(&h[p], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) if (h[i]) atomicAdd(&counts[i], h[i]);
}
// every workgroup takes a contiguous block of chunk slots: local histogram -> one reservation per (workgroup, partition) -> placement
__global__ __launch_bounds__(kBlock) void chunk_place_kernel(const unsigned int* __restrict__ chunk_part, int64_t n_chunks, uint32_t NP, const unsigned long long* __restrict__ cl_off,
                                                             unsigned int* __restrict__ cursor, unsigned int* __restrict__ cl_ids) {
  extern __shared__ unsigned long long lds_raw[];
  unsigned int* h = reinterpret_cast<unsigned int*>(lds_raw);     // [NP] counts, then running positions
  unsigned int* base = h + NP;                                     // [NP] reserved start within the partition's list
  const int64_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
  const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < n_chunks ? beg + per : n_chunks;
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (int64_t c = beg + threadIdx.x; c < end; c += blockDim.x) { const unsigned int p = chunk_part[c]; if (p != kNoChunk) atomicAdd(&h[p], 1u); }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NP; i += blockDim.x) { base[i] = h[i] ? atomicAdd(&cursor[i], h[i]) : 0u; h[i] = 0; }
  __syncthreads();
  for (int64_t c = beg + threadIdx.x; c < end; c += blockDim.x) {
    const unsigned int p = chunk_part[c];
    if (p == kNoChunk) continue;
    cl_ids[cl_off[p] + base[p] + atomicAdd(&h[p], 1u)] = (unsigned int)c;
  }
}
// groups of the hot keys -> appended to the dense output (a hot key whose rows were all filtered out has LEN == 0: no group)
__global__ __launch_bounds__(kBlock) void hot_emit_kernel(const unsigned long long* __restrict__ hot_keys, const unsigned long long* __restrict__ hot_out, uint32_t n_hot, int n_aggs,
                                                          int len_idx, unsigned long long* __restrict__ counter, unsigned int* __restrict__ overflow, uint32_t max_groups,
                                                          unsigned long long* __restrict__ out_keys, unsigned char* __restrict__ out_kvalid, unsigned long long* __restrict__ out_acc) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_hot || hot_out[(size_t)h * n_aggs + len_idx] == 0) return;
  const unsigned long long o = atomicAdd(counter, 1ull);
  if (o >= max_groups) { atomicExch(overflow, 2u); return; }
  out_keys[o] = hot_keys[h]; out_kvalid[o] = 1;
  for (int k = 0; k < n_aggs; k++) out_acc[o * n_aggs + k] = hot_out[(size_t)h * n_aggs + k];
}

#ifdef PLX_HAVE_Q3_SHAPES
#define PLX_PART2_STATIC_CASES(KERNEL, MODE, ...)                                                                                        \
  case SHAPE_GB_SUM_CNT_I64: hipLaunchKernelGGL((KERNEL<StatProg<SHAPE_GB_SUM_CNT_I64>, MODE>), __VA_ARGS__); break;                      \
  case SHAPE_GB_SUM_MEAN_U32_F64: hipLaunchKernelGGL((KERNEL<StatProg<SHAPE_GB_SUM_MEAN_U32_F64>, MODE>), __VA_ARGS__); break;
#define PLX_PART2_STATIC_SCATTER_CASES(MODE, TILES, ...)                                                                                 \
  case SHAPE_GB_SUM_CNT_I64: hipLaunchKernelGGL((part2_scatter_kernel<StatProg<SHAPE_GB_SUM_CNT_I64>, MODE, TILES>), __VA_ARGS__); break; \
  case SHAPE_GB_SUM_MEAN_U32_F64: hipLaunchKernelGGL((part2_scatter_kernel<StatProg<SHAPE_GB_SUM_MEAN_U32_F64>, MODE, TILES>), __VA_ARGS__); break;
#else
#define PLX_PART2_STATIC_CASES(KERNEL, MODE, ...)
#define PLX_PART2_STATIC_SCATTER_CASES(MODE, TILES, ...)
#endif

// ---- third generation: the AOT instantiations of the benchmark shapes (anything else goes through the JIT) -------------------
// config 3 (Int64 key, Int64 value): first run hash mode (key range unknown), then direct mode; records 16 / 12 B unpacked, 12 / 8 B with
// the value narrowed to a u32 offset, 4 B with key_low and value fused into one dword.  config 5 (u32 dictionary codes, Float64 value): 12 B.
#ifdef PLX_HAVE_Q3_SHAPES
#define PLX_P3_COMBOS(X)                                                                                                              \
  X(SHAPE_GB_SUM_CNT_I64, kP2Hash, 3, kPackNone) X(SHAPE_GB_SUM_CNT_I64, kP2Hash, 3, kPackNarrow) X(SHAPE_GB_SUM_CNT_I64, kP2Hash, 3, kPackPair) X(SHAPE_GB_SUM_CNT_I64, kP2Hash, 3, kPackPairV) \
  X(SHAPE_GB_SUM_CNT_I64, kP2Direct, 4, kPackNone) X(SHAPE_GB_SUM_CNT_I64, kP2Direct, 4, kPackNarrow) X(SHAPE_GB_SUM_CNT_I64, kP2Direct, 4, kPackFused) \
  X(SHAPE_GB_SUM_MEAN_U32_F64, kP2Hash, 3, kPackNone) X(SHAPE_GB_SUM_MEAN_U32_F64, kP2Direct, 4, kPackNone) X(SHAPE_GB_SUM_MEAN_U32_F64, kP2Direct, 4, kPackPair) \
  X(SHAPE_GB2_SUM_CNT_I64, kP2Hash, 2, kPackNarrow)       /* two-column key at 512 partitions: 20-byte records, two tiles of an 896-thread workgroup (values that do not narrow: 24-byte records, run-time compiled) */
// ... and with the hot-key path compiled in (skewed keys: heavy hitters are summed in the scatter), for config 3's two runs -- key range unknown / known
#define PLX_P3_HOT_COMBOS(X) X(SHAPE_GB_SUM_CNT_I64, kP2Hash, 3, kPackNarrow) X(SHAPE_GB_SUM_CNT_I64, kP2Direct, 3, kPackFused)
#else
#define PLX_P3_COMBOS(X)
#define PLX_P3_HOT_COMBOS(X)
#endif
static bool part3_static_available(int static_id, uint32_t mode, uint32_t tiles, uint32_t pack, bool hot) {
#define X(ID, MODE, TILES, PACK) if (static_id == ID && mode == (uint32_t)MODE && tiles == (uint32_t)TILES && pack == (uint32_t)PACK) return true;
  if (hot) { PLX_P3_HOT_COMBOS(X) return false; }
  PLX_P3_COMBOS(X)
#undef X
  return false;
}
static void part3_static_scatter(int static_id, const Shape& sh, const Args& args, const PartPlan2& pp, const ScatterParams2& sp, size_t lds) {
#define X(ID, MODE, TILES, PACK)                                                                                                       \
  if (static_id == ID && pp.mode == (uint32_t)MODE && pp.tiles == (uint32_t)TILES && pp.pack == (uint32_t)PACK) {                        \
    auto kern = pp.check_src ? part3_scatter_kernel<StatProg<ID>, (int)MODE, TILES, (int)PACK, false, true> : part3_scatter_kernel<StatProg<ID>, (int)MODE, TILES, (int)PACK, false, false>;   \
    static bool attr_set[2] = {false, false};                                                                                          \
    if (!attr_set[pp.check_src ? 1 : 0]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); attr_set[pp.check_src ? 1 : 0] = true; } \
    hipLaunchKernelGGL(kern, dim3(pp.scatter_grid), dim3(pp.block), lds, stream(), sh, args, pp, sp);                                   \
    return;                                                                                                                            \
  }
  if (!pp.n_hot) { PLX_P3_COMBOS(X) }
#undef X
#define X(ID, MODE, TILES, PACK)                                                                                                       \
  if (static_id == ID && pp.mode == (uint32_t)MODE && pp.tiles == (uint32_t)TILES && pp.pack == (uint32_t)PACK) {                        \
    auto kern = pp.check_src ? part3_scatter_kernel<StatProg<ID>, (int)MODE, TILES, (int)PACK, true, true> : part3_scatter_kernel<StatProg<ID>, (int)MODE, TILES, (int)PACK, true, false>;     \
    static bool attr_set[2] = {false, false};                                                                                          \
    if (!attr_set[pp.check_src ? 1 : 0]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); attr_set[pp.check_src ? 1 : 0] = true; } \
    hipLaunchKernelGGL(kern, dim3(pp.scatter_grid), dim3(pp.block), lds, stream(), sh, args, pp, sp);                                   \
    return;                                                                                                                            \
  }
  if (pp.n_hot) { PLX_P3_HOT_COMBOS(X) }
#undef X
}
static void part3_static_agg(int static_id, const PartPlan2& pp, const AggParams2& ap, uint32_t NP, size_t lds) {
#define X(ID, MODE, TILES, PACK)                                                                                                       \
  if (static_id == ID && pp.mode == (uint32_t)MODE && pp.pack == (uint32_t)PACK) {                                                       \
    hipLaunchKernelGGL((part2_agg_kernel<StatProg<ID>, (int)MODE, (int)PACK>), dim3(NP), dim3(kP2AggBlock), lds, stream(), pp, ap);     \
    return;                                                                                                                            \
  }
  PLX_P3_COMBOS(X)
  PLX_P3_HOT_COMBOS(X)       // (the aggregation pass is the same kernel with or without hot keys; its instantiations are among the plain combos: a no-op here)
#undef X
}

// Runs scatter -> chunk sort -> aggregate (+ hot groups).  Outputs (allocated here): dense keys / valid flags / cells.
// Returns the number of groups, -1 if an LDS table overflowed (the caller plans more partitions or falls back), -2 if no specialised kernel is available.
int64_t partitioned_agg2(const Shape& sh, const Args& args, const PartPlan2& plan, int static_id, const std::vector<uint64_t>& hot_keys, Buf* out_keys, Buf* out_kvalid,
                         Buf* out_acc, std::string* desc, int64_t* key_range_out, int64_t* wide_stride_out) {
  PartPlan2 pp = plan;
  const uint32_t NP = 1u << pp.log2_parts;
  const bool direct = pp.mode == kP2Direct;
  const bool gen3 = pp.gen == 3;
  const bool is_static = gen3 ? part3_static_available(static_id, pp.mode, pp.tiles, pp.pack, pp.n_hot != 0)      // most AOT builds leave the hot-key path out
                              : (static_id == SHAPE_GB_SUM_CNT_I64 || static_id == SHAPE_GB_SUM_MEAN_U32_F64);   // the cases of PLX_PART2_STATIC_CASES
  const jit::Sink jk_scatter = gen3 ? jit::part3_scatter_sink(pp.mode, pp.tiles, pp.pack, pp.n_hot != 0)
                                    : pp.tiles == 2 ? (direct ? jit::PART2_SCATTER_DIRECT_T2 : jit::PART2_SCATTER_HASH_T2) : (direct ? jit::PART2_SCATTER_DIRECT : jit::PART2_SCATTER_HASH);
  const jit::Sink jk_agg = gen3 ? jit::part3_agg_sink(pp.mode, pp.pack) : (direct ? jit::PART2_AGG_DIRECT : jit::PART2_AGG_HASH);
  const bool use_jit = !is_static && jit::ensure(sh, jk_scatter, args.n_rows) && jit::ensure(sh, jk_agg, args.n_rows);
  if (!is_static && !use_jit) return -2;
  PLX_REQUIRE(pp.tiles == 1 || pp.tiles == 2 || (gen3 && (pp.tiles == 3 || pp.tiles == 4)), PLX_ERR_INVALID, "partitioned_agg2: tiles per round");
  // the kernels' names in the HIP-event profile carry the variant (mode, tiles, packing, record dwords): a counter file of another variant must never
  // be read as this one's (bench.py pmc_traffic matches the full name)
  const std::string sid = use_jit ? "jit" : "#" + std::to_string(static_id);
  const std::string scatter_name = std::string(gen3 ? "part3_scatter[" : "part2_scatter[") + sid + (direct ? ",d,t" : ",h,t") + std::to_string(pp.tiles) + (gen3 ? ",p" + std::to_string(pp.pack) : "") +
                                   (gen3 && pp.n_hot ? ",hot]" : "]");
  const std::string agg_name = "part_agg_lds[" + sid + (direct ? ",d,p" : ",h,p") + std::to_string(pp.pack) + "]";
  PLX_REQUIRE(pp.n_hot == hot_keys.size(), PLX_ERR_INVALID, "partitioned_agg2: plan / hot key list mismatch");
  const uint32_t row_bytes = (pp.pack == kPackPair || pp.pack == kPackPairV) ? pp.rec_words * 2u : pp.rec_words * 4u;      // bytes a row travels as (a pair of rows shares a 20- / 28-byte record)
  const uint32_t chunk_dw = kP2ChunkRecs * pp.rec_words;
  const int64_t n_chunks = (int64_t)pp.scatter_grid * pp.chunks_per_wg;
  Buf recs = dev_alloc_transient((size_t)n_chunks * chunk_dw * 4 + 256);
  Buf chunk_part = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks), chunk_fill = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  PLX_HIP(hipMemsetAsync(chunk_part->ptr, 0xff, sizeof(uint32_t) * (size_t)n_chunks, stream()));
  Buf meta = dev_alloc_zero(64);             // [0..1] group counter (u64), [2] aggregation overflow, [3..4] scatter flags
  // hot-key lookup table (host-built open addressing, same hash as the kernel) and their accumulators
  const uint32_t hot_slots = pp.n_hot ? (1u << pp.log2_hot_slots) : 0;
  std::vector<uint64_t> h_keys(std::max<uint32_t>(hot_slots, 1), kEmptyKey), h_list(hot_keys);
  std::vector<uint32_t> h_idx(std::max<uint32_t>(hot_slots, 1), 0);
  Buf hot_tbl_keys, hot_tbl_idx, hot_out, hot_list;
  if (pp.n_hot) {
    for (uint32_t i = 0; i < pp.n_hot; i++) {
      uint32_t s = (uint32_t)((hot_keys[i] * 0x9e3779b97f4a7c15ull) >> (64 - pp.log2_hot_slots));
      while (h_keys[s] != kEmptyKey) s = (s + 1) & (hot_slots - 1);
      h_keys[s] = hot_keys[i]; h_idx[s] = i;
    }
    hot_tbl_keys = dev_alloc(sizeof(uint64_t) * hot_slots); hot_tbl_idx = dev_alloc(sizeof(uint32_t) * hot_slots);
    hot_list = dev_alloc(sizeof(uint64_t) * pp.n_hot);
    hot_out = dev_alloc(sizeof(uint64_t) * (size_t)pp.n_hot * sh.n_aggs);
    h2d_async(hot_tbl_keys->ptr, h_keys.data(), sizeof(uint64_t) * hot_slots);
    h2d_async(hot_tbl_idx->ptr, h_idx.data(), sizeof(uint32_t) * hot_slots);
    h2d_async(hot_list->ptr, h_list.data(), sizeof(uint64_t) * pp.n_hot);
    init_agg_cells(hot_out->as<uint64_t>(), pp.n_hot, sh);
  }
  ScatterParams2 sp{};
  sp.recs = recs->as<unsigned int>(); sp.chunk_part = chunk_part->as<unsigned int>(); sp.chunk_fill = chunk_fill->as<unsigned int>();
  sp.flags = meta->as<unsigned int>() + 3;
  sp.hot_tbl_keys = pp.n_hot ? hot_tbl_keys->as<unsigned long long>() : nullptr; sp.hot_tbl_idx = pp.n_hot ? hot_tbl_idx->as<unsigned int>() : nullptr;
  sp.hot_out = pp.n_hot ? hot_out->as<unsigned long long>() : nullptr;
  // by-product statistics (hash mode, signed keys): exact min / max of the keys this scan streams anyway
  Buf minmax;
  if (key_range_out && !direct) {
    minmax = dev_alloc(16);
    const long long init[2] = {0x7fffffffffffffffll, (long long)0x8000000000000000ull};
    h2d_async(minmax->ptr, init, 16);
    PLX_HIP(hipStreamSynchronize(stream()));     // `init` is a stack object
    sp.key_minmax = minmax->as<long long>();
  }
  const size_t slds = gen3 ? part3_scatter_lds(pp.block * kRows * pp.tiles, (pp.pack == kPackPair || pp.pack == kPackPairV) ? (direct ? 3u : 4u) : pp.rec_words, NP, hot_slots, pp.n_hot, sh.n_aggs, pp.hot_copies)
                           : part2_scatter_lds(NP, pp.ring_lines, hot_slots, pp.n_hot, sh.n_aggs, pp.hot_copies);
  {
    // pass traffic: inputs read once + every surviving row written as one record (upper bound: all rows)
    ProfileScope ps(scatter_name.c_str(), scan_bytes(sh, args) + (uint64_t)args.n_rows * row_bytes, (uint64_t)args.n_rows);
    if (!use_jit && gen3) part3_static_scatter(static_id, sh, args, pp, sp, slds);
    else if (use_jit) {
      Shape shc = sh; Args ac = args; PartPlan2 ppc = pp; ScatterParams2 spc = sp;
      void* ka[] = {&shc, &ac, &ppc, &spc};
      PLX_REQUIRE(jit::launch_raw(sh, jk_scatter, ka, (int)pp.scatter_grid, (int)pp.block, slds), PLX_ERR_HIP, "jit launch failed (part2_scatter)");
    } else {
#define PLX_P2_SCATTER(MODE, T) \
  switch (static_id) { PLX_PART2_STATIC_SCATTER_CASES((int)MODE, T, dim3(pp.scatter_grid), dim3(pp.block), slds, stream(), sh, args, pp, sp) default: break; }
      if (direct) { if (pp.tiles == 1) { PLX_P2_SCATTER(kP2Direct, 1) } else { PLX_P2_SCATTER(kP2Direct, 2) } }
      else { if (pp.tiles == 1) { PLX_P2_SCATTER(kP2Hash, 1) } else { PLX_P2_SCATTER(kP2Hash, 2) } }
#undef PLX_P2_SCATTER
    }
    PLX_HIP(hipGetLastError());
  }
  if (pp.ablate & 3u) PLX_HIP(hipMemsetAsync(chunk_fill->ptr, 0, sizeof(uint32_t) * (size_t)n_chunks, stream()));   // measurement mode: the records are garbage, aggregate none of them
  // chunk lists
  Buf counts = dev_alloc_zero(sizeof(uint32_t) * (NP + 1)), cursor = dev_alloc_zero(sizeof(uint32_t) * (NP + 1));
  Buf cl_off = dev_alloc(sizeof(uint64_t) * (NP + 2)), cl_ids = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  {
    ProfileScope ps("part2_chunk_sort", (uint64_t)n_chunks * 12, (uint64_t)n_chunks);
    const int g = grid_for(n_chunks, kBlock * 16, 2);
    hipLaunchKernelGGL(chunk_hist_kernel, dim3(g), dim3(kBlock), sizeof(unsigned int) * NP, stream(), chunk_part->as<unsigned int>(), n_chunks, NP, counts->as<unsigned int>());
    PLX_HIP(hipGetLastError());
    exclusive_scan_u32(counts->as<uint32_t>(), cl_off->as<uint64_t>(), NP);
    hipLaunchKernelGGL(chunk_place_kernel, dim3(g), dim3(kBlock), sizeof(unsigned int) * NP * 2, stream(), chunk_part->as<unsigned int>(), n_chunks, NP,
                       cl_off->as<unsigned long long>(), cursor->as<unsigned int>(), cl_ids->as<unsigned int>());
    PLX_HIP(hipGetLastError());
  }
  const bool wide = sh.n_keys != 0;
  const uint64_t n_slots = direct ? ((uint64_t)1 << pp.log2_slots) : wide ? (uint64_t)pp.n_slots : ((uint64_t)pp.n_slots + 2);
  const uint64_t max_groups = std::min<uint64_t>((uint64_t)NP * n_slots + pp.n_hot, (uint64_t)args.n_rows + 1);
  *out_keys = dev_alloc(sizeof(uint64_t) * max_groups * (wide ? sh.n_keys : 1));        // wide: [n_keys][max_groups] words, [n_keys][max_groups] valid flags
  *out_kvalid = dev_alloc(max_groups * (wide ? sh.n_keys : 1));
  *out_acc = dev_alloc(sizeof(uint64_t) * max_groups * sh.n_aggs);
  AggParams2 ap{};
  ap.recs = recs->as<unsigned int>(); ap.chunk_fill = chunk_fill->as<unsigned int>(); ap.cl_off = cl_off->as<unsigned long long>(); ap.cl_ids = cl_ids->as<unsigned int>();
  ap.counter = meta->as<unsigned long long>(); ap.overflow = meta->as<unsigned int>() + 2;
  ap.out_keys = (*out_keys)->as<unsigned long long>(); ap.out_kvalid = (*out_kvalid)->as<unsigned char>(); ap.out_acc = (*out_acc)->as<unsigned long long>();
  ap.max_groups = (uint32_t)std::min<uint64_t>(max_groups, 0xffffffffull);
  if (wide_stride_out) *wide_stride_out = (int64_t)ap.max_groups;
  {
    ProfileScope ps(agg_name.c_str(), (uint64_t)args.n_rows * row_bytes, (uint64_t)args.n_rows);
    const size_t lds = wide ? (size_t)pp.n_tags * 4 + n_slots * 8 * ((size_t)sh.n_keys + pp.wide_null_word + sh.n_aggs) : n_slots * 8 * ((direct ? 0 : 1) + sh.n_aggs);
    if (!use_jit && gen3) part3_static_agg(static_id, pp, ap, NP, lds);
    else if (use_jit) {
      PartPlan2 ppc = pp; AggParams2 apc = ap;
      void* ka[] = {&ppc, &apc};
      PLX_REQUIRE(jit::launch_raw(sh, jk_agg, ka, (int)NP, kP2AggBlock, lds), PLX_ERR_HIP, "jit launch failed (part2_agg)");
    } else if (direct) {
      switch (static_id) { PLX_PART2_STATIC_CASES(part2_agg_kernel, (int)kP2Direct, dim3(NP), dim3(kP2AggBlock), lds, stream(), pp, ap) default: break; }
    } else {
      switch (static_id) { PLX_PART2_STATIC_CASES(part2_agg_kernel, (int)kP2Hash, dim3(NP), dim3(kP2AggBlock), lds, stream(), pp, ap) default: break; }
    }
    PLX_HIP(hipGetLastError());
  }
  if (pp.n_hot) {
    hipLaunchKernelGGL(hot_emit_kernel, dim3((pp.n_hot + kBlock - 1) / kBlock), dim3(kBlock), 0, stream(), hot_list->as<unsigned long long>(), hot_out->as<unsigned long long>(), pp.n_hot,
                       (int)sh.n_aggs, (int)pp.len_idx, ap.counter, ap.overflow, ap.max_groups, ap.out_keys, ap.out_kvalid, ap.out_acc);
    PLX_HIP(hipGetLastError());
  }
  uint32_t res[6] = {0, 0, 0, 0, 0, 0};
  d2h_sync(res, meta->ptr, 24);
  PLX_REQUIRE(!res[3], PLX_ERR_INVALID, "partitioned group-by: a scatter workgroup ran out of chunks");
  PLX_REQUIRE(!res[4], PLX_ERR_INVALID, "group key outside the bounds declared for its column (plx_column_set_bounds)");
  PLX_REQUIRE(!res[5], PLX_ERR_INVALID, "aggregated value outside the bounds assumed for its column");
  if (res[2]) return -1;
  if (minmax) { long long mm[2]; d2h_sync(mm, minmax->ptr, 16); key_range_out[0] = mm[0]; key_range_out[1] = mm[1]; }
  if (desc) *desc = std::string(gen3 ? "partitioned(v3," : "partitioned(v2,") + (direct ? "direct" : "hash") + ",P=" + std::to_string(NP) + ",rec=" + std::to_string(row_bytes) +
                    (gen3 ? "B,pack=" + std::to_string(pp.pack) + ",tile=" + std::to_string(pp.block * kRows * pp.tiles) : "B,ring=" + std::to_string(pp.ring_lines * 128) + "B") +
                    ",block=" + std::to_string(pp.block) + ",hot=" + std::to_string(pp.n_hot) + (use_jit ? ",kernels=jit" : ",kernels=aot") + ")+" + (direct ? "lds_direct_table(slots=" : wide ? "lds_wide_key_table(words=" + std::to_string(sh.n_keys + pp.wide_null_word) + ",slots=" : "lds_hash_table(slots=") +
                    std::to_string(direct ? 1u << pp.log2_slots : pp.n_slots) + ")";
  return (int64_t)(((uint64_t)res[1] << 32) | res[0]);
}


// ======================================================================================================================
// Partitioned join probe (kernels_fused.hpp: partitioned_probe_hits)
// ======================================================================================================================
// One workgroup per partition.  Its slice of the build bitmap (2^key_shift bits: up to 512 KB) does not fit the LDS, but the build keys in
// it are few (the planner requires a sparse bitmap): the prologue turns the slice's set bits into a Bloom filter in LDS (kBloomK probes over
// `bloom_bits` bits: ~10 bits per build key at TPC-H SF100 -> under 1 % false positives), and the partition's records are then streamed against
// LDS alone -- no access outside the chip per record, no dependent load chain (the first version re-checked positives of a FOLDED slice in
// HBM: every chunk waited ~3 us for that load, 1.7 ms per 3.2e8 records).  What comes out are CANDIDATE row ids: the ordinary probe kernel the
// caller runs over them tests the exact bitmap anyway.  Output slots are reserved with an LDS counter in the partition's private region (a
// million returning atomics on one global counter serialise: 9 ms).
constexpr uint32_t kBloomK = 4;
__device__ __forceinline__ uint32_t bloom_pos(uint32_t low, uint32_t i, uint32_t log2_bits) {
  constexpr uint32_t mult[kBloomK] = {0x9e3779b1u, 0x85ebca77u, 0xc2b2ae3du, 0x27d4eb2fu};
  return ((low + 1u) * mult[i]) >> (32u - log2_bits);
}
// Keys WITHOUT a usable range (hash_bits != 0; `bits` is null): the build side is an open-addressing hash table in HBM (JoinAggTable) whose slot hash is
// key * kP2HashMult -- the hash the scatter took partition and tag from -- so the keys of partition p sit in region p of the table: slots
// [p * cap / NP, (p + 1) * cap / NP) plus, linear probing, the run of occupied slots behind the region's end (wrapping at the table's end).  The prologue
// streams that region (coalesced) and puts the tag bits of every key that belongs to partition p into the Bloom filter; keys of partition p - 1 that spilled
// into this region are skipped (they are not p's).  Everything after the prologue is the same.
struct ProbeHashedBuild {
  const unsigned long long* table_slots;   // [cap + 1][2] {key, build row} slots of the build table (JoinAggTable); slot `cap` = the key whose bits equal kEmptyKey (present iff its row is set)
  uint32_t log2_cap;
  uint32_t hash_bits;                      // 0: the bitmap source
  uint32_t windowed;                       // the table's probe sequences wrap inside windows (JoinAggTable::log2_window): no key of a region sits behind its end
};
__global__ __launch_bounds__(kP2AggBlock) void probe_pass_kernel(const unsigned int* __restrict__ recs, const unsigned int* __restrict__ chunk_fill, const unsigned long long* __restrict__ cl_off,
                                                                 const unsigned int* __restrict__ cl_ids, const unsigned long long* __restrict__ bits, unsigned long long range,
                                                                 unsigned long long n_words, uint32_t key_shift, uint32_t slice /* keys per partition (a multiple of 64) */, uint32_t log2_bloom_bits, uint32_t exact, ProbeHashedBuild hb,
                                                                 unsigned int* __restrict__ hits /* a region of (chunks x 256) slots per partition */, unsigned int* __restrict__ part_hits) {
  extern __shared__ unsigned long long lds_raw[];
  unsigned int* bloom = reinterpret_cast<unsigned int*>(lds_raw);          // exact != 0: the slice itself (it fits), bit = key low bits
  __shared__ unsigned int wg_hits;
  const uint32_t p = blockIdx.x;
  const uint32_t W = hb.hash_bits ? 0u : slice >> 6;                      // 64-bit words of the slice
  const uint32_t bloom_words = 1u << (log2_bloom_bits - 5);
  if (threadIdx.x == 0) wg_hits = 0;
  for (uint32_t i = threadIdx.x; i < bloom_words; i += blockDim.x) bloom[i] = 0;
  __syncthreads();
  if (hb.hash_bits) {
    const uint32_t tag_mask = key_shift >= 32 ? 0xffffffffu : (1u << key_shift) - 1u;
    const uint32_t log2_parts = hb.hash_bits - key_shift;
    const unsigned long long cap = 1ull << hb.log2_cap, region = cap >> log2_parts;       // the host guarantees cap >= NP
    auto add = [&](unsigned long long key) {
      const unsigned long long id = (key * kP2HashMult) >> (64u - hb.hash_bits);
      if ((uint32_t)(id >> key_shift) != p) return;
      const uint32_t low = (uint32_t)id & tag_mask;
#pragma unroll
      for (uint32_t q = 0; q < kBloomK; q++) { const uint32_t pos = bloom_pos(low, q, log2_bloom_bits); atomicOr(&bloom[pos >> 5], 1u << (pos & 31u)); }
    };
    for (unsigned long long j = threadIdx.x; j < region; j += blockDim.x) { const unsigned long long key = hb.table_slots[((unsigned long long)p * region + j) * 2]; if (key != kEmptyKey) add(key); }
    if (threadIdx.x == 0) {
      // the run of occupied slots behind the region: keys of this partition that linear probing pushed past its end (short: the table is at most half full)
      for (unsigned long long s = (((unsigned long long)p + 1) * region) & (cap - 1), n = 0; n < cap && !hb.windowed; s = (s + 1) & (cap - 1), n++) { const unsigned long long key = hb.table_slots[s * 2]; if (key == kEmptyKey) break; add(key); }
      if ((unsigned int)hb.table_slots[cap * 2 + 1] != kNoRow32) add(kEmptyKey);                                  // the key equal to the EMPTY pattern lives in slot `cap`
    }
  }
  for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) {
    const unsigned long long wi = (unsigned long long)p * W + i;
    unsigned long long w = wi < n_words ? bits[wi] : 0ull;
    if (exact) { if (w) { bloom[i * 2] = (unsigned int)w; bloom[i * 2 + 1] = (unsigned int)(w >> 32); } continue; }
    while (w) {
      const uint32_t b = (uint32_t)__builtin_ctzll(w); w &= w - 1;
      const uint32_t low = i * 64u + b;
#pragma unroll
      for (uint32_t q = 0; q < kBloomK; q++) { const uint32_t pos = bloom_pos(low, q, log2_bloom_bits); atomicOr(&bloom[pos >> 5], 1u << (pos & 31u)); }
    }
  }
  __syncthreads();
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr uint32_t kPerLane = kP2ChunkRecs / 64;
  const uint64_t c_beg = cl_off[p], c_end = cl_off[p + 1];
  const unsigned long long region = c_beg * kP2ChunkRecs;                  // this partition's output region: one slot per record it may hold
  // Three chunks per wave are in flight and the chunk IDS run one more step ahead: a chunk's records can only be addressed once its id has
  // arrived, so loading id and records in the same step made every step wait for two dependent memory round trips (3 us per chunk).
  unsigned long long ra[kPerLane], rb[kPerLane], rc[kPerLane];             // raw 8-B records {key_low | row << key_shift}
  uint32_t fill_a = 0, fill_b = 0, fill_c = 0;
  const uint32_t low_mask = key_shift >= 32 ? 0xffffffffu : (1u << key_shift) - 1u;
  auto load_id = [&](uint64_t j) -> uint32_t { return cl_ids[j < c_end ? j : c_end - 1]; };   // past the end: the last chunk again, ignored (fill 0)
  auto load_chunk = [&](uint64_t j, uint32_t id, unsigned long long* r, uint32_t& fill) __attribute__((always_inline)) {
    fill = j < c_end ? chunk_fill[id] : 0u;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(recs) + (uint64_t)id * kP2ChunkRecs;
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) r[u] = base[(uint32_t)lane + u * 64u];              // the chunk is allocated in full
  };
  auto process = [&](const unsigned long long* r, uint32_t fill) __attribute__((always_inline)) {
    uint64_t m[kPerLane];
    unsigned int lo[kPerLane], rid[kPerLane];
    uint32_t total = 0;
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) { lo[u] = (unsigned int)r[u] & low_mask; rid[u] = (unsigned int)(r[u] >> key_shift); }
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      const uint32_t i = (uint32_t)lane + u * 64u;
      bool pos = i < fill && (hb.hash_bits || (unsigned long long)p * slice + lo[u] < range);
      if (exact) pos = pos && ((bloom[lo[u] >> 5] >> (lo[u] & 31u)) & 1u);
      else {
#pragma unroll
        for (uint32_t q = 0; q < kBloomK; q++) { const uint32_t b = bloom_pos(lo[u], q, log2_bloom_bits); pos = pos && ((bloom[b >> 5] >> (b & 31u)) & 1u); }
      }
      m[u] = ballot(pos); total += (uint32_t)popc64(m[u]);
    }
    if (!total) return;                                                                      // wave-uniform
    unsigned int o32 = 0;
    if (lane == 0) o32 = atomicAdd(&wg_hits, total);
    unsigned long long o = region + (unsigned long long)(unsigned int)__shfl((int)o32, 0, 64);
#pragma unroll
    for (uint32_t u = 0; u < kPerLane; u++) {
      if ((m[u] >> lane) & 1ull) hits[o + (uint64_t)prefix_rank(m[u])] = rid[u];             // row id (the caller guarantees < 2^32 rows)
      o += (uint64_t)popc64(m[u]);
    }
  };
  if (c_beg < c_end) {
    const uint64_t step = (uint64_t)nwaves;
    uint64_t j = c_beg + (uint64_t)wave;
    uint32_t id_next;
    load_chunk(j, load_id(j), ra, fill_a);
    load_chunk(j + step, load_id(j + step), rb, fill_b);
    id_next = load_id(j + 2 * step);
    for (;;) {                                                                               // the three buffers rotate by name
      if (j >= c_end) break;
      { const uint32_t id = id_next; id_next = load_id(j + 3 * step); load_chunk(j + 2 * step, id, rc, fill_c); } process(ra, fill_a); j += step;
      if (j >= c_end) break;
      { const uint32_t id = id_next; id_next = load_id(j + 3 * step); load_chunk(j + 2 * step, id, ra, fill_a); } process(rb, fill_b); j += step;
      if (j >= c_end) break;
      { const uint32_t id = id_next; id_next = load_id(j + 3 * step); load_chunk(j + 2 * step, id, rb, fill_b); } process(rc, fill_c); j += step;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) part_hits[p] = wg_hits;
}
// the partitions' hit regions -> one dense list
__global__ __launch_bounds__(kBlock) void probe_hits_compact_kernel(const unsigned int* __restrict__ regions, const unsigned long long* __restrict__ cl_off, const unsigned int* __restrict__ part_hits,
                                                                    const unsigned long long* __restrict__ out_off, unsigned int* __restrict__ out) {
  const uint32_t p = blockIdx.x;
  const unsigned int* src = regions + cl_off[p] * kP2ChunkRecs;
  unsigned int* dst = out + out_off[p];
  for (uint32_t i = threadIdx.x; i < part_hits[p]; i += blockDim.x) dst[i] = src[i];
}

// dt: the direct-address bitmap of the build side (keys with a dense range); ht: the build side's open-addressing hash table (keys without one: partition and
// record tag come from the key's hash, kP2HashMult).  Exactly one of them is given.
static bool probe_hits_impl(const Shape& sh, const Args& args, const DirectJoinTable* dtp, const JoinAggTable* ht, uint64_t n_build, int static_id, ColumnPtr* hits_out, std::string* desc) {
  if (args.n_rows >= (int64_t)0xfffffff0ll || sh.n_aggs != 1 || sh.aggs[0].kind != AGG_FIRST_ROW || sh.key == kNone || sh.n_keys) return false;
  const bool hashed = ht != nullptr;
  constexpr uint32_t kHashBits = 39;                                           // 8 partition bits + a 31-bit tag: the record is tag | row << 31
  const DirectJoinTable dt = dtp ? *dtp : DirectJoinTable{};
  const uint32_t bits_total = hashed ? kHashBits : std::max<uint32_t>(ceil_log2(dt.range), 15);
  PartPlan2 pp{};
  pp.mode = kP2Direct; pp.gen = 3; pp.pack = kPackRowid; pp.n_hot = 0; pp.hot_copies = 1; pp.len_idx = 0; pp.oob_drop = 1; pp.key_base = hashed ? 0 : dt.kmin;
  pp.hash_bits = hashed ? kHashBits : 0u;
  pp.ablate = kEnvP2Ablate > 0 ? ((uint32_t)kEnvP2Ablate & 3u) : 0u;      // measurement only (the records are garbage: no partition keeps any of them, the join finds nothing)
  pp.log2_parts = std::min<uint32_t>(8, bits_total - 9);                       // 256 partitions (the scatter's best geometry: 8192-row tiles), slices of >= 2^9 keys
  if (pp.log2_parts < 6) return false;
  pp.key_shift = bits_total - pp.log2_parts;
  if (!hashed) {
    // a key range is cut into P EQUAL slices (multiples of 64 keys: whole bitmap words), not at a power of two: SF100's 6.0e8-key range cut at 2^22 populated 143
    // of 256 partitions -- the probe pass, one workgroup per partition, ran on 56 % of the chip
    const uint64_t per = (dt.range + (((uint64_t)1 << pp.log2_parts) - 1)) >> pp.log2_parts;
    pp.slice = (uint32_t)std::max<uint64_t>((per + 63) & ~(uint64_t)63, 512);
    pp.slice_magic = ~0ull / pp.slice;                                          // floor((2^64 - 1) / slice): the quotient it gives is exact or one short (make_record2 corrects)
    pp.key_shift = std::max<uint32_t>(ceil_log2(pp.slice), 9);
  }
  pp.log2_slots = pp.key_shift;
  const RecLayout2 L = rec_layout2(sh, kP2Direct, kPackRowid);
  if (!L.has_rowid || L.n_src != 0 || L.rec_words != 2 || pp.key_shift > 32) return false;     // one 64-bit field: key low bits | row id << key_shift
  pp.rec_words = L.rec_words; pp.block = kP2MaxBlock;
  const uint32_t NP = 1u << pp.log2_parts;
  if (kEnvP3Block >= 64 && (uint32_t)kEnvP3Block >= NP && kEnvP3Block % 64 == 0) pp.block = (uint32_t)kEnvP3Block;
  const size_t lds_total = 160 * 1024 - 2048;
  uint32_t tiles = 0;
  for (uint32_t t : {4u, 2u, 1u}) if (part3_scatter_lds(kP2MaxBlock * kRows * t, L.rec_words, NP, 0, 0, sh.n_aggs, 1) <= lds_total) { tiles = t; break; }
  if (!tiles) return false;
  plan2_geometry(pp, args.n_rows, tiles);
  // LDS of the probe pass: the partition's bitmap slice itself when it fits (exact), else a Bloom filter of 2^20 bits (128 KB)
  const bool exact = !hashed && pp.key_shift <= 20;
  const uint32_t log2_bloom = exact ? std::max<uint32_t>(pp.key_shift, 6) : 20;
  // a Bloom filter of 2^20 bits with 4 probes stays under ~2 % false positives up to 2^17 keys: denser slices would flood the caller with candidates
  if (!exact && (hashed ? (double)n_build / (double)NP : (double)n_build * (double)pp.slice / (double)dt.range) > (double)(1u << 17)) return false;
  const jit::Sink jk = jit::part3_scatter_sink(pp.mode, pp.tiles, pp.pack, false);
#ifdef PLX_HAVE_Q3_PROBE_SCATTER
  const bool aot = static_id == SHAPE_Q3_PROBE_SCATTER && pp.tiles == 4;       // TPC-H Q3's probe side (two- and three-table variants share it)
#else
  const bool aot = false;
#endif
  if (!aot && !jit::ensure(sh, jk, args.n_rows)) return false;
  const uint32_t chunk_dw = kP2ChunkRecs * pp.rec_words;
  const int64_t n_chunks = (int64_t)pp.scatter_grid * pp.chunks_per_wg;
  Buf recs = dev_alloc_transient((size_t)n_chunks * chunk_dw * 4 + 256);
  Buf chunk_part = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks), chunk_fill = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  PLX_HIP(hipMemsetAsync(chunk_part->ptr, 0xff, sizeof(uint32_t) * (size_t)n_chunks, stream()));
  Buf meta = dev_alloc_zero(64);             // [0..1] hit counter (u64), [3..4] scatter flags
  ScatterParams2 sp{};
  sp.recs = recs->as<unsigned int>(); sp.chunk_part = chunk_part->as<unsigned int>(); sp.chunk_fill = chunk_fill->as<unsigned int>(); sp.flags = meta->as<unsigned int>() + 3;
  {
    const std::string name = aot ? "probe_scatter[#" + std::to_string(static_id) + ",d,t4,p3]" : std::string("probe_scatter[jit]");     // (one kernel for range and hash partitions: PartPlan2::hash_bits)
    ProfileScope ps(name.c_str(), scan_bytes(sh, args) + (uint64_t)args.n_rows * pp.rec_words * 4, (uint64_t)args.n_rows);
    const size_t slds = part3_scatter_lds(pp.block * kRows * pp.tiles, pp.rec_words, NP, 0, 0, sh.n_aggs, 1);
    if (aot) {
#ifdef PLX_HAVE_Q3_PROBE_SCATTER
      hipLaunchKernelGGL((part3_scatter_kernel<StatProg<SHAPE_Q3_PROBE_SCATTER>, (int)kP2Direct, 4, (int)kPackRowid, false>), dim3(pp.scatter_grid), dim3(pp.block), slds, stream(), sh, args, pp, sp);
      PLX_HIP(hipGetLastError());
#endif
    } else {
      Shape shc = sh; Args ac = args; PartPlan2 ppc = pp; ScatterParams2 spc = sp;
      void* ka[] = {&shc, &ac, &ppc, &spc};
      PLX_REQUIRE(jit::launch_raw(sh, jk, ka, (int)pp.scatter_grid, (int)pp.block, slds), PLX_ERR_HIP, "jit launch failed (probe scatter)");
    }
  }
  if (pp.ablate) PLX_HIP(hipMemsetAsync(chunk_fill->ptr, 0, sizeof(uint32_t) * (size_t)n_chunks, stream()));
  Buf counts = dev_alloc_zero(sizeof(uint32_t) * (NP + 1)), cursor = dev_alloc_zero(sizeof(uint32_t) * (NP + 1));
  Buf cl_off = dev_alloc(sizeof(uint64_t) * (NP + 2)), cl_ids = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks);
  {
    ProfileScope ps("part2_chunk_sort", (uint64_t)n_chunks * 12, (uint64_t)n_chunks);
    const int g = grid_for(n_chunks, kBlock * 16, 2);
    hipLaunchKernelGGL(chunk_hist_kernel, dim3(g), dim3(kBlock), sizeof(unsigned int) * NP, stream(), chunk_part->as<unsigned int>(), n_chunks, NP, counts->as<unsigned int>());
    PLX_HIP(hipGetLastError());
    exclusive_scan_u32(counts->as<uint32_t>(), cl_off->as<uint64_t>(), NP);
    hipLaunchKernelGGL(chunk_place_kernel, dim3(g), dim3(kBlock), sizeof(unsigned int) * NP * 2, stream(), chunk_part->as<unsigned int>(), n_chunks, NP,
                       cl_off->as<unsigned long long>(), cursor->as<unsigned int>(), cl_ids->as<unsigned int>());
    PLX_HIP(hipGetLastError());
  }
  auto hits = std::make_shared<Column>();
  hits->dtype = PLX_U32; hits->null_count = 0;
  Buf regions = dev_alloc(sizeof(uint32_t) * (size_t)n_chunks * kP2ChunkRecs + 16);          // a slot per record a partition may hold
  Buf part_hits = dev_alloc_zero(sizeof(uint32_t) * (NP + 1)), hit_off = dev_alloc(sizeof(uint64_t) * (NP + 2));
  ProbeHashedBuild hb{};
  if (hashed) {
    if (ht->log2_cap < pp.log2_parts) return false;                           // a table smaller than the partition count needs no partitioned probe
    hb.table_slots = ht->slots; hb.log2_cap = ht->log2_cap; hb.hash_bits = pp.hash_bits;
    hb.windowed = ht->log2_window && ht->log2_window + pp.log2_parts <= ht->log2_cap ? 1u : 0u;      // (windows no larger than a partition's region)
  }
  {
    ProfileScope ps("probe_pass_lds",     // one kernel symbol for the bitmap-slice, Bloom-from-bitmap and Bloom-from-hash-table sources: one tracer name (the plan description says which)
                    (uint64_t)args.n_rows * pp.rec_words * 4 + (hashed ? ((uint64_t)1 << ht->log2_cap) * 16 : dt.range / 8), (uint64_t)args.n_rows);
    const unsigned long long n_words = hashed ? 0ull : (dt.range / 512 + 1) * 8;
    hipLaunchKernelGGL(probe_pass_kernel, dim3(NP), dim3(kP2AggBlock), ((size_t)1 << (log2_bloom - 3)), stream(), recs->as<unsigned int>(), chunk_fill->as<unsigned int>(),
                       cl_off->as<unsigned long long>(), cl_ids->as<unsigned int>(), hashed ? nullptr : dt.bits, hashed ? 0ull : dt.range, n_words, pp.key_shift, pp.slice, log2_bloom, exact ? 1u : 0u, hb,
                       regions->as<unsigned int>(), part_hits->as<unsigned int>());
    PLX_HIP(hipGetLastError());
    exclusive_scan_u32(part_hits->as<uint32_t>(), hit_off->as<uint64_t>(), NP);              // hit_off[NP] = total
  }
  uint32_t res[5] = {0, 0, 0, 0, 0};
  d2h_sync(res, meta->ptr, 20);
  PLX_REQUIRE(!res[3], PLX_ERR_INVALID, "partitioned probe: a scatter workgroup ran out of chunks");
  uint64_t total_hits = 0;
  d2h_sync(&total_hits, hit_off->as<uint64_t>() + NP, 8);
  hits->len = (int64_t)total_hits;
  hits->values = dev_alloc(values_bytes(PLX_U32, std::max<int64_t>(hits->len, 1)));
  if (hits->len) {
    hipLaunchKernelGGL(probe_hits_compact_kernel, dim3(NP), dim3(kBlock), 0, stream(), regions->as<unsigned int>(), cl_off->as<unsigned long long>(), part_hits->as<unsigned int>(),
                       hit_off->as<unsigned long long>(), hits->values->as<unsigned int>());
    PLX_HIP(hipGetLastError());
  }
  *hits_out = hits;
  if (desc) *desc = std::string(hashed ? "partitioned_hash_probe(P=" : "partitioned_probe(P=") + std::to_string(NP) + ",rec=" + std::to_string(pp.rec_words * 4) + "B,tile=" + std::to_string(pp.block * kRows * pp.tiles) + (exact ? ",lds_bitmap=" : ",lds_bloom=") + std::to_string(((size_t)1 << (log2_bloom - 3)) >> 10) +
                    "KB,candidates=" + std::to_string(hits->len) + ")";
  return true;
}

bool partitioned_probe_hits(const Shape& sh, const Args& args, const DirectJoinTable& dt, uint64_t n_build, int static_id, ColumnPtr* hits_out, std::string* desc) {
  return probe_hits_impl(sh, args, &dt, nullptr, n_build, static_id, hits_out, desc);
}
// The same for a build side WITHOUT a dense key range (its open-addressing hash table `ht`, already built): rows -> partitions and 31-bit tags by the key's hash,
// per-partition Bloom filters in LDS from the table's keys; the candidates then go through the ordinary hash probe (fused_probe_agg), which compares whole keys.
bool partitioned_hash_probe_hits(const Shape& sh, const Args& args, const JoinAggTable& ht, uint64_t n_build, int static_id, ColumnPtr* hits_out, std::string* desc) {
  return probe_hits_impl(sh, args, nullptr, &ht, n_build, static_id, hits_out, desc);
}

// ======================================================================================================================
// Partitioned join BUILD (kernels_fused.hpp: partitioned_join_build)
// ======================================================================================================================
// The plain build (JoinBuildSink) claims a slot per build row with a compare-and-swap on a random line of a table far beyond the caches: every insert is a line across
// the fabric and back (SF100 orders: 1.46e7 inserts into 512 MB, 1.4 of the scan's 2.0 ms).  Here the table is made of WINDOWS of 2^log2_window slots that one workgroup
// each fills from an LDS image and writes out in whole lines (empty slots included: the table needs no memset); probe sequences wrap inside the window
// (JoinAggTable::log2_window, jt_next), so nothing of a window's keys lives outside it.  Three kernels:
//   pairs  the build side's predicate + key scan appends {key, row} pairs in per-wave ordinal chunks (the direct-address build's sink without its bitmap: one device atomic
//          per 1024 pairs);
//   bin    a workgroup per 32768 ordinals: counts its pairs per window in LDS, reserves room in each window's record region with ONE device atomic per (workgroup, window)
//          and copies the pairs there (a window's region holds as many records as the window has slots: more cannot be inserted anyway -> flags[1]);
//   fill   a workgroup per window: LDS compare-and-swap inserts, then the image goes out.
// A duplicate key raises flags[0] (the caller builds again in multi-value mode, the plain way), a window that fills up flags[1].  (Measured and dropped first: records
// radix-partitioned 256 / 512 ways by the scatter of §4.2 and windows that re-read their partition's chunk lists -- scatter 0.95 / 1.6 ms + fill 3.7 ms.)
constexpr uint32_t kFillBlock = 1024, kBinBlock = 1024, kBinTile = 32768;
__device__ __forceinline__ bool ord_live(const DirectJoinTable& dt, uint32_t ord) { return (ord & (kOrdChunk - 1)) < dt.chunk_used[ord / kOrdChunk]; }
__global__ __launch_bounds__(kBinBlock) void join_bin_kernel(DirectJoinTable dt, JoinAggTable t, ulonglong2* __restrict__ recs, unsigned int* __restrict__ win_fill) {
  extern __shared__ unsigned int bin_lds[];
  const uint32_t log2_nw = t.log2_cap - t.log2_window, NW = 1u << log2_nw, W = 1u << t.log2_window;
  unsigned int* cnt = bin_lds;            // [NW] pairs of this tile per window, then the cursor inside the reserved run
  unsigned int* base = bin_lds + NW;      // [NW] first record of the run reserved in the window's region
  const uint32_t n_res = min(*dt.counter, dt.n_ord);
  const uint64_t o0 = (uint64_t)blockIdx.x * kBinTile;
  if (o0 >= n_res) return;
  const uint32_t n = (uint32_t)min((uint64_t)kBinTile, n_res - o0);
  for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t ord = (uint32_t)o0 + i;
    if (!ord_live(dt, ord)) continue;
    const unsigned long long key = dt.ord_key[ord];
    if (key == kEmptyKey) continue;
    atomicAdd(&cnt[(uint32_t)((key * kP2HashMult) >> (64 - log2_nw))], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < NW; i += blockDim.x) { const unsigned int c = cnt[i]; base[i] = c ? atomicAdd(&win_fill[i], c) : 0u; cnt[i] = 0; }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t ord = (uint32_t)o0 + i;
    if (!ord_live(dt, ord)) continue;
    const unsigned long long key = dt.ord_key[ord];
    const unsigned int row = dt.ord_row[ord];
    if (key == kEmptyKey) {                               // the key equal to the EMPTY pattern lives in slot `cap`, outside every window
      const unsigned int old = atomicExch(jt_row(t, 1ull << t.log2_cap), row);
      if (old != kNoRow32) t.flags[0] = 1u;
      if (t.count) atomicAdd(t.count, 1ull);
      continue;
    }
    const uint32_t w = (uint32_t)((key * kP2HashMult) >> (64 - log2_nw));
    const unsigned int pos = base[w] + atomicAdd(&cnt[w], 1u);
    if (pos < W) recs[((uint64_t)w << t.log2_window) + pos] = make_ulonglong2(key, (unsigned long long)row);
    else t.flags[1] = 1u;
  }
}
__global__ __launch_bounds__(kFillBlock) void join_fill_kernel(const ulonglong2* __restrict__ recs, const unsigned int* __restrict__ win_fill, const unsigned long long* __restrict__ win_base,
                                                               JoinAggTable t, unsigned long long* __restrict__ cell_key, unsigned int* __restrict__ cell_row) {
  extern __shared__ unsigned long long fill_lds[];
  const uint32_t W = 1u << t.log2_window;
  unsigned long long* lkeys = fill_lds;            // [W]
  unsigned long long* lvals = fill_lds + W;        // [W] {row, cell index << 32}
  __shared__ unsigned int s_count;
  const uint64_t w = blockIdx.x;
  const uint32_t n = min(uniform_ld(win_fill, w), W);
  const uint64_t cell0 = uniform_ld(win_base, w);  // cells are numbered in record order: window by window
  for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) { lkeys[i] = kEmptyKey; lvals[i] = ~0ull; }
  if (threadIdx.x == 0) s_count = 0;
  if (w == 0 && threadIdx.x == 0) {                // the key equal to the EMPTY pattern (slot `cap`): the cell behind every window's
    const uint64_t total = win_base[1ull << (t.log2_cap - t.log2_window)];
    jt_row(t, 1ull << t.log2_cap)[1] = (unsigned int)total;
    if (cell_key) { cell_key[total] = kEmptyKey; cell_row[total] = *jt_row(t, 1ull << t.log2_cap); }
  }
  __syncthreads();
  unsigned int mine = 0;
  const ulonglong2* in = recs + (w << t.log2_window);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const ulonglong2 rec = in[i];
    const unsigned long long key = rec.x;
    if (cell_key) { cell_key[cell0 + i] = key; cell_row[cell0 + i] = (unsigned int)rec.y; }
    uint32_t ls = (uint32_t)((key * kP2HashMult) >> (64 - t.log2_cap)) & (W - 1);
    for (uint32_t probe = 0;; probe++) {
      const unsigned long long old = atomicCAS(&lkeys[ls], (unsigned long long)kEmptyKey, key);
      if (old == kEmptyKey) { lvals[ls] = (rec.y & 0xffffffffull) | ((cell0 + i) << 32); mine++; break; }
      if (old == key) { t.flags[0] = 1u; break; }                             // duplicate build key
      ls = (ls + 1) & (W - 1);
      if (probe >= W) { t.flags[1] = 1u; break; }                             // window full
    }
  }
  if (mine) atomicAdd(&s_count, mine);
  __syncthreads();
  ulonglong2* out = reinterpret_cast<ulonglong2*>(t.slots) + (w << t.log2_window);
  for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) out[i] = make_ulonglong2(lkeys[i], lvals[i]);
  if (threadIdx.x == 0 && t.count && s_count) atomicAdd(t.count, (unsigned long long)s_count);
}

// sh / args: the build side's predicate + key program (no aggregates: what the plain build runs).  t: slots allocated ([cap + 1][2] words, NOT initialised), flags and count
// zeroed, log2_window set by the caller.  cells (may be null): the keys and build rows in cell order ([keys + 1], allocated here) for the output step of a join ->
// aggregate.  false: the geometry does not fit (nothing was touched: memset the table and build the plain way).
bool partitioned_join_build(const Shape& sh, const Args& args, int static_id, const JoinAggTable& t, JoinCells* cells, std::string* desc) {
  if (sh.key == kNone || sh.n_keys || t.links || !t.log2_window || t.log2_cap < t.log2_window + 3) return false;
  const uint32_t log2_nw = t.log2_cap - t.log2_window;
  if (log2_nw > 14) return false;                                  // window counters in LDS (bin kernel: 8 bytes a window)
  const uint64_t ord_cap = (uint64_t)args.n_rows + (uint64_t)scan_waves(args.n_rows) * kOrdChunk + kOrdChunk;
  if (ord_cap >= 0xfffffff0ull) return false;
  const uint64_t NW = 1ull << log2_nw;
  Buf okey = dev_alloc_transient(sizeof(uint64_t) * ord_cap), orow = dev_alloc_transient(sizeof(uint32_t) * ord_cap), used = dev_alloc_zero(sizeof(uint32_t) * (ord_cap / kOrdChunk + 2));
  Buf meta = dev_alloc_zero(32);                                   // [0] ordinal counter, [2..3] the pair sink's flags
  Buf win_fill = dev_alloc_zero(sizeof(uint32_t) * (NW + 1)), win_base = dev_alloc(sizeof(uint64_t) * (NW + 2));
  Buf recs = dev_alloc_transient((size_t)16 << t.log2_cap);
  DirectJoinTable dt{};
  dt.bits = nullptr; dt.ord_key = okey->as<unsigned long long>(); dt.ord_row = orow->as<unsigned int>(); dt.chunk_used = used->as<unsigned int>();
  dt.counter = meta->as<unsigned int>(); dt.flags = meta->as<unsigned int>() + 2; dt.n_ord = (unsigned int)ord_cap;
  fused_direct_build(sh, args, dt, static_id);                     // (a null bitmap: the sink appends pairs only)
  PLX_HIP(hipMemsetAsync(t.slots + ((size_t)2 << t.log2_cap), 0xff, 16, stream()));       // slot `cap`
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)join_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)join_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError(); attr_set = true;
  }
  {
    // every reserved ordinal is visited (live or not: the capacity bounds the grid, workgroups past the counter leave at once)
    const uint64_t n_tiles = (ord_cap + kBinTile - 1) / kBinTile;
    ProfileScope ps("join_bin_windows", (uint64_t)args.n_rows / 4, (uint64_t)args.n_rows);
    hipLaunchKernelGGL(join_bin_kernel, dim3((unsigned int)n_tiles), dim3(kBinBlock), (size_t)NW * 8, stream(), dt, t, reinterpret_cast<ulonglong2*>(recs->ptr), win_fill->as<unsigned int>());
    PLX_HIP(hipGetLastError());
  }
  exclusive_scan_u32(win_fill->as<uint32_t>(), win_base->as<uint64_t>(), (int64_t)NW);       // win_base[NW] = records in all windows
  if (cells) {
    // (at most one record per row that passed; the table was sized for them: cap slots bound the cells whatever the count turns out to be)
    cells->key = dev_alloc(sizeof(uint64_t) * (((size_t)1 << t.log2_cap) + 1));
    cells->row = dev_alloc(sizeof(uint32_t) * (((size_t)1 << t.log2_cap) + 1));
  }
  {
    const size_t lds = ((size_t)16 << t.log2_window);
    ProfileScope ps("join_fill_lds", ((uint64_t)16 << t.log2_cap) * 3 / 2, NW);
    hipLaunchKernelGGL(join_fill_kernel, dim3((unsigned int)NW), dim3(kFillBlock), lds, stream(), reinterpret_cast<const ulonglong2*>(recs->ptr), win_fill->as<unsigned int>(),
                       win_base->as<unsigned long long>(), t, cells ? cells->key->as<unsigned long long>() : nullptr, cells ? cells->row->as<unsigned int>() : nullptr);
    PLX_HIP(hipGetLastError());
  }
  if (desc) *desc = "partitioned build(pairs -> " + std::to_string(NW) + " windows of " + std::to_string(1u << t.log2_window) + " slots filled from LDS)";
  return true;
}

// ---- is a key column (roughly) sorted?  fraction of non-decreasing adjacent pairs over 64 evenly spaced runs of 1024 rows ----------------
__global__ __launch_bounds__(kBlock) void sortedness_kernel(const void* __restrict__ values, int width, int is_signed, int64_t n, int64_t run, int64_t stride, unsigned int* __restrict__ out /* [2]: pairs, ordered */) {
  unsigned int pairs = 0, ordered = 0;
  const int64_t base = (int64_t)blockIdx.x * stride;
  for (int64_t i = threadIdx.x; i + 1 < run && base + i + 1 < n; i += blockDim.x) {
    long long a, b;
    const int64_t j = base + i;
    switch (width) {
      case 1: a = is_signed ? (long long)((const signed char*)values)[j] : (long long)((const unsigned char*)values)[j]; b = is_signed ? (long long)((const signed char*)values)[j + 1] : (long long)((const unsigned char*)values)[j + 1]; break;
      case 2: a = is_signed ? (long long)((const short*)values)[j] : (long long)((const unsigned short*)values)[j]; b = is_signed ? (long long)((const short*)values)[j + 1] : (long long)((const unsigned short*)values)[j + 1]; break;
      case 4: a = is_signed ? (long long)((const int*)values)[j] : (long long)((const unsigned int*)values)[j]; b = is_signed ? (long long)((const int*)values)[j + 1] : (long long)((const unsigned int*)values)[j + 1]; break;
      default: a = ((const long long*)values)[j]; b = ((const long long*)values)[j + 1]; break;
    }
    pairs++; ordered += a <= b;
  }
  if (pairs) { atomicAdd(&out[0], pairs); atomicAdd(&out[1], ordered); }
}
double sample_sortedness(const ColumnPtr& c) {
  if (!c || !dtype_is_int(c->dtype) || c->dtype == PLX_U64 || c->len < 2) return 1.0;
  const int64_t n = c->len, run = 1024;
  const int blocks = (int)std::min<int64_t>(64, (n + run - 1) / run);
  const int64_t stride = blocks > 1 ? (n - run) / (blocks - 1) : 0;
  Buf out = dev_alloc_zero(8);
  hipLaunchKernelGGL(sortedness_kernel, dim3(blocks), dim3(kBlock), 0, stream(), c->data(), dtype_width(c->dtype), dtype_is_signed(c->dtype) ? 1 : 0, n, run, stride, out->as<unsigned int>());
  PLX_HIP(hipGetLastError());
  uint32_t h[2] = {0, 0};
  d2h_sync(h, out->ptr, 8);
  return h[0] ? (double)h[1] / (double)h[0] : 1.0;
}

// ---- smallest / largest valid value over 64 evenly spaced runs of 1024 rows: does an integer key column LOOK dense? --------------------------------------
__global__ __launch_bounds__(kBlock) void sample_minmax_kernel(const void* __restrict__ values, const uint64_t* __restrict__ validity, int width, int is_signed, int64_t n, int64_t run, int64_t stride,
                                                               long long* __restrict__ out /* [2]: min, max */) {
  long long lo = 0x7fffffffffffffffll, hi = (long long)0x8000000000000000ull;
  const int64_t base = (int64_t)blockIdx.x * stride;
  for (int64_t i = threadIdx.x; i < run && base + i < n; i += blockDim.x) {
    const int64_t j = base + i;
    if (validity && !((validity[j >> 6] >> (j & 63)) & 1)) continue;
    long long a;
    switch (width) {
      case 1: a = is_signed ? (long long)((const signed char*)values)[j] : (long long)((const unsigned char*)values)[j]; break;
      case 2: a = is_signed ? (long long)((const short*)values)[j] : (long long)((const unsigned short*)values)[j]; break;
      case 4: a = is_signed ? (long long)((const int*)values)[j] : (long long)((const unsigned int*)values)[j]; break;
      default: a = ((const long long*)values)[j]; break;
    }
    lo = a < lo ? a : lo; hi = a > hi ? a : hi;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const long long a = (long long)shfl_xor_u64((uint64_t)lo, m), b = (long long)shfl_xor_u64((uint64_t)hi, m);
    lo = a < lo ? a : lo; hi = b > hi ? b : hi;
  }
  if (lane_id() == 0 && lo <= hi) { atomicMin(out, lo); atomicMax(out + 1, hi); }
}
bool sample_minmax(const ColumnPtr& c, int64_t* mn, int64_t* mx, int max_blocks) {
  if (!c || !dtype_is_int(c->dtype) || c->dtype == PLX_U64 || c->len < 1 || !c->values) return false;
  const int64_t n = c->len, run = 1024;
  const int blocks = (int)std::min<int64_t>(max_blocks, (n + run - 1) / run);
  const int64_t stride = blocks > 1 ? (n - run) / (blocks - 1) : 0;
  Buf out = dev_alloc(16);
  const long long init[2] = {0x7fffffffffffffffll, (long long)0x8000000000000000ull};
  h2d_async(out->ptr, init, 16);
  hipLaunchKernelGGL(sample_minmax_kernel, dim3(blocks), dim3(kBlock), 0, stream(), c->data(), c->valid_words(), dtype_width(c->dtype), dtype_is_signed(c->dtype) ? 1 : 0, n, run, stride, out->as<long long>());
  PLX_HIP(hipGetLastError());
  long long h[2] = {0, 0};
  d2h_sync(h, out->ptr, 16);
  if (h[0] > h[1]) return false;
  *mn = h[0]; *mx = h[1];
  return true;
}

}  // namespace k
}  // namespace plx
