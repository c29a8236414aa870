// scan.hpp -- device-wide exclusive prefix sums (hand-written; used by filter
// compaction, join emit offsets and partition scatter).
#pragma once
#include "core.hpp"
namespace plx {
namespace k {
// out[0..n] (n+1 entries) = exclusive prefix sums of in[0..n); out[n] = total.
void exclusive_scan_u32(const uint32_t* in, uint64_t* out, int64_t n);
void exclusive_scan_u64(const uint64_t* in, uint64_t* out, int64_t n);
}  // namespace k
}  // namespace plx
