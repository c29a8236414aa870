// What bandwidth can a partition scatter reach on MI355X as a function of its WRITE GRANULARITY?
// Every workgroup streams its share of an 8-GB input sequentially (16 B / lane) and writes the same bytes out in blocks of G
// bytes whose destinations are a pseudo-random permutation of the output blocks (what 512 partitions x 256 workgroups
// look like to the memory system).  G = 128 B is one flushed line of the partitioned group-by; larger G = longer contiguous
// runs per (workgroup, partition).  Also: the pure copy (sequential writes) and a read-only pass as references.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro_scatter_bw.bin tools/micro_scatter_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// block b (G bytes) of the input goes to block perm(b) of the output; perm = multiply by an odd constant mod 2^k (a bijection)
template <int G, int WINDOW>
__global__ __launch_bounds__(1024) void k_scatter(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n_blocks_log2, uint64_t n16) {
  constexpr uint64_t per_block16 = G / 16;                 // 16-B units per block
  const uint64_t mask = (1ull << n_blocks_log2) - 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
    const uint64_t b = i / per_block16, off = i % per_block16;
    uint64_t pb;
    if (WINDOW == 0) pb = (b * 0x9e3779b97f4a7c15ull) & mask;            // anywhere in the output
    else {                                                               // random within a window of WINDOW blocks (working set that fits the Infinity Cache)
      const uint64_t w = b / WINDOW, j = b % WINDOW;
      pb = w * WINDOW + ((j * 0x9e3779b1ull + 12345) % WINDOW);
    }
    out[pb * per_block16 + off] = v;
  }
}
__global__ __launch_bounds__(1024) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ __launch_bounds__(1024) void k_read(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n16) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(1024) void k_write(uint4* __restrict__ out, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = make_uint4((unsigned)i, 1, 2, 3);
}

template <class F> static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const uint64_t bytes = 8ull << 30, n16 = bytes / 16;
  uint4 *in = nullptr, *out = nullptr;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
  CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
  const int grid = 256 * 2, blk = 1024;
  auto report = [&](const char* name, float ms, double moved) { printf("%-44s %8.3f ms  %7.1f GB/s (bytes moved %.1f GB)\n", name, ms, moved / ms / 1e6, moved / 1e9); };
  report("read only", time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(blk), 0, 0, in, out, n16); }), (double)bytes);
  report("write only (sequential)", time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(blk), 0, 0, out, n16); }), (double)bytes);
  report("copy (sequential read + sequential write)", time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(blk), 0, 0, in, out, n16); }), 2.0 * bytes);
#define RUN(G, W, label) { const uint64_t nb = bytes / G; uint64_t lg = 0; while ((1ull << lg) < nb) lg++; \
    report(label, time_ms([&] { hipLaunchKernelGGL((k_scatter<G, W>), dim3(grid), dim3(blk), 0, 0, in, out, lg, n16); }), 2.0 * bytes); }
  RUN(64, 0, "scatter, 64-B blocks, anywhere")
  RUN(128, 0, "scatter, 128-B blocks, anywhere")
  RUN(256, 0, "scatter, 256-B blocks, anywhere")
  RUN(512, 0, "scatter, 512-B blocks, anywhere")
  RUN(1024, 0, "scatter, 1-KB blocks, anywhere")
  RUN(4096, 0, "scatter, 4-KB blocks, anywhere")
  RUN(128, 1048576, "scatter, 128-B blocks, within 128-MB windows")
  RUN(128, 262144, "scatter, 128-B blocks, within 32-MB windows")
  RUN(256, 524288, "scatter, 256-B blocks, within 128-MB windows")
  return 0;
}
