// kernels_scan.hip -- hierarchical exclusive scan: 2048 elements per workgroup
// (8 per lane), wave64 shuffle scan + LDS across the 4 waves, block totals scanned
// recursively, then one add pass.  3 passes over the data; the inputs here are tile /
// row counts (<= 4 B per 8 B..2 KB of column data), so this is never the dominant kernel.
#include "dev.hpp"
#include "kernels.hpp"
#include "scan.hpp"

namespace plx {
namespace k {
using namespace dev;

constexpr int kScanPerThread = 8;
constexpr int kScanPerBlock = kBlock * kScanPerThread;

template <class IN>
__global__ __launch_bounds__(kBlock) void scan_block_kernel(const IN* __restrict__ in, uint64_t* __restrict__ out, int64_t n,
                                                            uint64_t* __restrict__ block_sums) {
  __shared__ uint64_t wave_tot[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanPerBlock + (int64_t)threadIdx.x * kScanPerThread;
  uint64_t v[kScanPerThread], tsum = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; j++) { v[j] = (base + j < n) ? (uint64_t)in[base + j] : 0; tsum += v[j]; }
  // inclusive wave scan of thread sums
  uint64_t incl = tsum;
  const int lane = lane_id();
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    uint32_t lo = __shfl_up((uint32_t)incl, s, 64), hi = __shfl_up((uint32_t)(incl >> 32), s, 64);
    if (lane >= s) incl += ((uint64_t)hi << 32) | lo;
  }
  const int wave = threadIdx.x >> 6;
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint64_t wave_off = 0, total = 0;
  for (int w = 0; w < kBlock / 64; w++) { if (w < wave) wave_off += wave_tot[w]; total += wave_tot[w]; }
  uint64_t run = wave_off + incl - tsum;
#pragma unroll
  for (int j = 0; j < kScanPerThread; j++) { if (base + j < n) out[base + j] = run; run += v[j]; }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(uint64_t* __restrict__ out, int64_t n, const uint64_t* __restrict__ block_prefix) {
  const int64_t base = (int64_t)blockIdx.x * kScanPerBlock;
  const uint64_t add = block_prefix[blockIdx.x];
  for (int j = threadIdx.x; j < kScanPerBlock; j += kBlock) if (base + j < n) out[base + j] += add;
}

template <class IN>
static void scan_impl(const IN* in, uint64_t* out, int64_t n) {
  if (n == 0) { PLX_HIP(hipMemsetAsync(out, 0, 8, stream())); return; }
  int64_t nblocks = (n + kScanPerBlock - 1) / kScanPerBlock;
  Buf sums = dev_alloc(sizeof(uint64_t) * (size_t)nblocks);
  Buf prefix = dev_alloc(sizeof(uint64_t) * (size_t)(nblocks + 1));
  hipLaunchKernelGGL((scan_block_kernel<IN>), dim3((unsigned)nblocks), dim3(kBlock), 0, stream(), in, out, n, sums->as<uint64_t>());
  PLX_HIP(hipGetLastError());
  if (nblocks == 1) {
    PLX_HIP(hipMemcpyAsync(out + n, sums->ptr, 8, hipMemcpyDeviceToDevice, stream()));
    return;
  }
  scan_impl<uint64_t>(sums->as<uint64_t>(), prefix->as<uint64_t>(), nblocks);
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblocks), dim3(kBlock), 0, stream(), out, n, prefix->as<uint64_t>());
  PLX_HIP(hipGetLastError());
  PLX_HIP(hipMemcpyAsync(out + n, prefix->as<uint64_t>() + nblocks, 8, hipMemcpyDeviceToDevice, stream()));
}

void exclusive_scan_u32(const uint32_t* in, uint64_t* out, int64_t n) { ProfileScope ps("exclusive_scan", (uint64_t)n * 12, (uint64_t)n); scan_impl<uint32_t>(in, out, n); }
void exclusive_scan_u64(const uint64_t* in, uint64_t* out, int64_t n) { ProfileScope ps("exclusive_scan", (uint64_t)n * 16, (uint64_t)n); scan_impl<uint64_t>(in, out, n); }

}  // namespace k
}  // namespace plx
