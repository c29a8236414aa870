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

}  // namespace k
}  // namespace plx
