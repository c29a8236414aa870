// jit.hpp -- run-time specialisation of the fused scan kernel (hiprtc).
//
// The benchmark shapes are instantiated ahead of time (fused_shapes.hpp); any other program shape used to run
// the generic interpreter, which is instruction-fetch bound (~30 k instructions of switch bodies; 0.8 TB/s against
// 5.8 TB/s for an AOT shape).  jit::launch instantiates the SAME template (fused_sinks.hpp: fused_scan_body<P, Sink>)
// with the program as a compile-time constant for the shape at hand, once per (shape, sink) and process
// (~1-2 s of hiprtc, amortised over large inputs: only used from PLX_JIT_MIN_ROWS rows, default 2^22), and launches it.
// Returns false when JIT is disabled (PLX_JIT=0), the input is small, or compilation failed -- the caller then runs
// the generic interpreter, so results never depend on the JIT.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "fused.hpp"

namespace plx {
namespace jit {

enum Sink { REGAGG = 0, LDSAGG, DENSE, HASH, WIDE, JOIN_BUILD, PROBE_AGG, DIRECT_BUILD, DIRECT_PROBE, BITMAP_BUILD,
            // kernels of the partitioned group-by (partition_device.hpp); launched with launch_raw
            PART_COUNT, PART_SCATTER, PART_AGG,
            // second generation (partition2_device.hpp)
            PART2_SCATTER_HASH, PART2_SCATTER_DIRECT, PART2_AGG_HASH, PART2_AGG_DIRECT,
            PART2_SCATTER_HASH_T2, PART2_SCATTER_DIRECT_T2,      // two tiles per wave and round
            // third generation scatter (partition3_device.hpp): PART3_SCATTER + mode + 2 * (tiles - 1) + 8 * pack (tiles 1..4, pack 0..3: 32 kinds), and the
            // aggregation pass over packed records: PART3_AGG + mode + 2 * pack (8 kinds)
            PART3_SCATTER, PART3_AGG = PART3_SCATTER + 64,     // scatter: ... + 32 * (hot-key path compiled in)
            // the selection of a filter -> frame in ballot form (fused_sinks.hpp BallotSink; a plain scan sink, launched with launch()).  New kinds go HERE, at the
            // end: the number is part of the kernel symbols the tracers key their rows by
            BALLOT = PART3_AGG + 8,
            DIRECT_HITS,        // fused_sinks.hpp DirectHitsSink (params = fused::DirectHits)
            // the third-generation scatter / aggregation with two rows a record (fused::kPackPair): PART3_SCATTER_PAIR + (tiles - 1) + 4 * hot + 8 * mode, PART3_AGG_PAIR + mode
            PART3_SCATTER_PAIR, PART3_AGG_PAIR = PART3_SCATTER_PAIR + 16, PART3_AGG_PAIR_DIRECT,
            // ... and with the value as a 48-bit offset (fused::kPackPairV, hash mode): PART3_SCATTER_PAIRV + (tiles - 1) + 4 * hot
            PART3_SCATTER_PAIRV, PART3_AGG_PAIRV = PART3_SCATTER_PAIRV + 8,
            kNumSinks };
inline Sink part3_scatter_sink(uint32_t mode, uint32_t tiles, uint32_t pack, bool hot) {
  if (pack == fused::kPackPairV) return (Sink)(PART3_SCATTER_PAIRV + (tiles - 1) + (hot ? 4 : 0));
  if (pack == fused::kPackPair) return (Sink)(PART3_SCATTER_PAIR + (tiles - 1) + (hot ? 4 : 0) + 8 * mode);
  return (Sink)(PART3_SCATTER + mode + 2 * (tiles - 1) + 8 * pack + (hot ? 32 : 0));
}
inline Sink part3_agg_sink(uint32_t mode, uint32_t pack) { return pack == fused::kPackPairV ? PART3_AGG_PAIRV : pack == fused::kPackPair ? (Sink)(PART3_AGG_PAIR + mode) : (Sink)(PART3_AGG + mode + 2 * pack); }

bool launch(const fused::Shape& sh, const fused::Args& args, Sink sink, const void* params, int grid, size_t lds_bytes);
// Compile (or fetch) the specialised kernel of (shape, kind) without launching: lets a multi-kernel pipeline decide up
// front whether every stage is available.  n_rows gates on PLX_JIT / PLX_JIT_MIN_ROWS like launch().
bool ensure(const fused::Shape& sh, Sink kind, int64_t n_rows);
// Launch with an explicit argument list (pointers to each kernel argument, in the wrapper's order).
bool launch_raw(const fused::Shape& sh, Sink kind, void** kargs, int grid, int block, size_t lds_bytes);
// compile-only check of the JIT toolchain for one (shape, sink): "" on success, else the compiler log (no GPU needed)
std::string selftest(const fused::Shape& sh, Sink sink);
// "aot" (pre-instantiated), "jit" (run-time specialised) or "generic" (interpreter): how a program of this size runs
const char* program_mode(int static_id, int64_t n_rows);
// inputs of at least min_rows rows use the JIT; < 0 disables it (overrides PLX_JIT / PLX_JIT_MIN_ROWS)
void set_min_rows(int64_t min_rows);
// statistics for tests / explain: kernels compiled so far, total compile milliseconds
void stats(int* compiled, double* compile_ms);

}  // namespace jit
}  // namespace plx
