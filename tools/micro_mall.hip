// micro_mall.hip -- does a buffer written by one kernel and read by the next stay on the chip (256 MB Infinity Cache)?
// For sizes S: (a) write S then read S, alternating kernels, 20 rounds; (b) read S twice (read-after-read); (c) a 128-B block scatter of S (the partition
// scatter's store pattern) followed by a read.  Reports effective GB/s per kernel.  If (a) is well above the ~5 TB/s mixed HBM rate for S <= 128 MB, a
// partitioned operator can be run slab by slab with its records never leaving the Infinity Cache.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro_mall.hip -o tools/micro_mall.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_write(uint4* p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v + 1, (unsigned)i, v);
}
__global__ __launch_bounds__(256) void k_read(const uint4* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 q = p[i]; acc += q.x + q.y + q.z + q.w; }
  if (acc == 0x123456789ull) out[0] = acc;
}
// 128-B lines written in a pseudo-random line order (8 lanes x 16 B per line)
__global__ __launch_bounds__(256) void k_scatter(uint4* p, size_t n_lines, unsigned v) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, sub = t & 7;
  for (size_t l = t >> 3; l < n_lines; l += ((size_t)gridDim.x * blockDim.x) >> 3) {
    const size_t dst = (l * 0x9E3779B97F4A7C15ull >> 20) % n_lines;
    p[dst * 8 + sub] = make_uint4(v, (unsigned)l, v, v);
  }
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned long long* out; CK(hipMalloc(&out, 64));
  const size_t MB = 1 << 20;
  const size_t big = 2048 * MB;
  uint4* buf; CK(hipMalloc(&buf, big));
  uint4* src; CK(hipMalloc(&src, big));
  CK(hipMemset(buf, 1, big)); CK(hipMemset(src, 2, big));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  for (size_t S : {16 * MB, 32 * MB, 64 * MB, 96 * MB, 128 * MB, 192 * MB, 256 * MB, 512 * MB, 1024 * MB, 2048 * MB}) {
    const size_t n = S / 16;
    const int R = 20;
    float ms;
    // (a) write then read
    for (int w = 0; w < 2; w++) { k_write<<<grid, 256, 0, st>>>(buf, n, w); k_read<<<grid, 256, 0, st>>>(buf, n, out); }
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < R; r++) { k_write<<<grid, 256, 0, st>>>(buf, n, r); k_read<<<grid, 256, 0, st>>>(buf, n, out); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double wr = 2.0 * S * R / (ms * 1e-3) / 1e9;
    // (b) read twice
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 2 * R; r++) k_read<<<grid, 256, 0, st>>>(buf, n, out);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double rr = 2.0 * S * R / (ms * 1e-3) / 1e9;
    // (c) block scatter then read
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < R; r++) { k_scatter<<<grid, 256, 0, st>>>(buf, S / 128, r); k_read<<<grid, 256, 0, st>>>(buf, n, out); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double sr = 2.0 * S * R / (ms * 1e-3) / 1e9;
    // (d) the slab pipeline's shape: read 4 S of input from a LARGE source (streams through), write S (records), read S (records): do the records stay resident while input streams by?
    const size_t n_in = (4 * S <= big ? 4 * S : big) / 16;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < R; r++) { k_read<<<grid, 256, 0, st>>>(src, n_in, out); k_write<<<grid, 256, 0, st>>>(buf, n, r); k_read<<<grid, 256, 0, st>>>(buf, n, out); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double pipe = ((double)n_in * 16 + 2.0 * S) * R / (ms * 1e-3) / 1e9;
    printf("S=%5zu MB  write+read %7.0f GB/s   read+read %7.0f GB/s   scatter+read %7.0f GB/s   stream-in(4S)+write+read %7.0f GB/s  (%.1f us per write+read pair)\n",
           S / MB, wr, rr, sr, pipe, 1e3 * (2.0 * S / (wr * 1e9)) * 1e3);
  }
  return 0;
}
