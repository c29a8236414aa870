// Microbenchmarks that decide the high-cardinality group-by design on MI355X:
//   A  device-scope atomics on a random 1e6-slot table in HBM (what fused_scan_hashagg does)
//   B  workgroup-scope (XCD-L2-resident) atomics on an XCD-private 1/8 slice of the table
//   C  LDS atomics on a workgroup-private table
//   D  8-way partition scatter of (key, value) rows (read 16 B/row, write 16 B/row)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro_atomics.bin tools/micro_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// A: keys[i] in [0, G): two device-scope atomics per row into acc[2*G]
__global__ __launch_bounds__(256) void k_dev_atomics(const int64_t* __restrict__ key, const int64_t* __restrict__ val, int64_t n, unsigned long long* acc) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = key[i];
    atomicAdd(&acc[2 * k], (unsigned long long)val[i]);
    atomicAdd(&acc[2 * k + 1], 1ull);
  }
}
// A2: one 16-byte cell update as a single packed 64-bit atomic (sum in low 40 bits, count in high 24) -- half the atomics
__global__ __launch_bounds__(256) void k_dev_atomics_packed(const int64_t* __restrict__ key, const int64_t* __restrict__ val, int64_t n, unsigned long long* acc) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = key[i];
    atomicAdd(&acc[k], (unsigned long long)val[i] + (1ull << 40));
  }
}
// B: rows pre-partitioned by key range: bucket b holds rows with key in [b*G/8, (b+1)*G/8).  A workgroup works on the
// bucket of the XCD it runs on and uses workgroup-scope atomics (execute in that XCD's L2).
__global__ __launch_bounds__(256) void k_l2_atomics(const int64_t* __restrict__ key, const int64_t* __restrict__ val, const int64_t* __restrict__ bucket_off,
                                                    unsigned long long* acc, unsigned int* cursors, int chunk) {
  const uint32_t x = xcc_id() & 7;
  const int64_t b0 = bucket_off[x], b1 = bucket_off[x + 1];
  __shared__ int64_t s_base;
  for (;;) {
    if (threadIdx.x == 0) s_base = b0 + (int64_t)atomicAdd(&cursors[x * 32], 1u) * chunk;
    __syncthreads();
    const int64_t base = s_base;
    __syncthreads();
    if (base >= b1) break;
    const int64_t end = base + chunk < b1 ? base + chunk : b1;
    for (int64_t i = base + threadIdx.x; i < end; i += blockDim.x) {
      const int64_t k = key[i];
      __hip_atomic_fetch_add(&acc[2 * k], (unsigned long long)val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&acc[2 * k + 1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
// C: LDS atomics: each workgroup owns G_lds slots, keys folded into them
__global__ __launch_bounds__(256) void k_lds_atomics(const int64_t* __restrict__ key, const int64_t* __restrict__ val, int64_t n, unsigned long long* out, int g_lds) {
  extern __shared__ unsigned long long tbl[];
  for (int i = threadIdx.x; i < 2 * g_lds; i += blockDim.x) tbl[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(key[i] % g_lds);
    atomicAdd(&tbl[2 * k], (unsigned long long)val[i]);
    atomicAdd(&tbl[2 * k + 1], 1ull);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tbl[0] + tbl[1];
}
// D: 8-way partition by key range.  Pass 1 histogram per block, pass 2 scatter with per-wave aggregated offsets.
__global__ __launch_bounds__(256) void k_hist(const int64_t* __restrict__ key, int64_t n, int64_t G, unsigned long long* hist) {
  __shared__ unsigned int h[8];
  if (threadIdx.x < 8) h[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { int b = (int)(key[i] * 8 / G); c[b]++; }
  for (int b = 0; b < 8; b++) if (c[b]) atomicAdd(&h[b], c[b]);
  __syncthreads();
  if (threadIdx.x < 8) hist[(size_t)blockIdx.x * 8 + threadIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_scatter(const int64_t* __restrict__ key, const int64_t* __restrict__ val, int64_t n, int64_t G, const unsigned long long* __restrict__ block_off /* [grid][8] */,
                                                 int64_t* __restrict__ okey, int64_t* __restrict__ oval) {
  // each block processes a contiguous row range (same assignment as k_hist would need: here grid-stride, so use atomics on block cursors)
  __shared__ unsigned long long cur[8];
  if (threadIdx.x < 8) cur[threadIdx.x] = block_off[(size_t)blockIdx.x * 8 + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; base < n; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = base + lane;
    const bool act = i < n;
    const int64_t k = act ? key[i] : 0, v = act ? val[i] : 0;
    const int b = act ? (int)(k * 8 / G) : -1;
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const unsigned long long m = __ballot(b == p);
      if (m == 0) continue;
      unsigned long long o = 0;
      const int leader = __ffsll((long long)m) - 1;
      if (lane == leader) o = atomicAdd(&cur[p], (unsigned long long)__popcll(m));
      o = __shfl(o, leader, 64);   // only low 32 bits matter below 2^32 rows per block cursor
      unsigned long long o_hi = __shfl((unsigned int)(o >> 32), leader, 64);
      (void)o_hi;
      if (b == p) { const unsigned long long r = o + __popcll(m & ((1ull << lane) - 1)); okey[r] = k; oval[r] = v; }
    }
  }
}

int main() {
  const int64_t n = 400000000ll, G = 1000000;
  int64_t *key, *val, *okey, *oval; unsigned long long* acc;
  CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&okey, n * 8)); CK(hipMalloc(&oval, n * 8));
  CK(hipMalloc(&acc, (size_t)G * 16 + 4096));
  // fill keys on host-side generator via a tiny kernel
  auto fill = [] __device__(int64_t) {};
  (void)fill;
  {
    // generate with a kernel
    struct L { static __global__ void gen(int64_t* key, int64_t* val, int64_t n, int64_t G) {
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { uint64_t r = splitmix((uint64_t)i); key[i] = (int64_t)(r % (uint64_t)G); val[i] = (int64_t)((r >> 40) % 1000); } } };
    hipLaunchKernelGGL(L::gen, dim3(4096), dim3(256), 0, 0, key, val, n, G);
    CK(hipDeviceSynchronize());
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  auto report = [&](const char* name, double bytes) { printf("%-34s %8.3f ms  %7.1f GB/s  %6.2f Grows/s\n", name, ms, bytes / ms / 1e6, n / ms / 1e6); };
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(acc, 0, (size_t)G * 16));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dev_atomics, dim3(2048), dim3(256), 0, 0, key, val, n, acc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("A device atomics x2 (1e6 keys)", n * 16.0);
    CK(hipMemset(acc, 0, (size_t)G * 16));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dev_atomics_packed, dim3(2048), dim3(256), 0, 0, key, val, n, acc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("A2 device atomics x1 packed", n * 16.0);
  }
  // C
  for (int g_lds : {1024, 4096}) {
    unsigned long long* out; CK(hipMalloc(&out, 8 * 4096));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_atomics, dim3(1024), dim3(256), (size_t)g_lds * 16, 0, key, val, n, out, g_lds); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    char nm[64]; snprintf(nm, 64, "C LDS atomics x2 (%d slots/WG)", g_lds); report(nm, n * 16.0);
    CK(hipFree(out));
  }
  // D: partition
  const int grid = 2048;
  unsigned long long *hist, *boff; CK(hipMalloc(&hist, grid * 8 * 8)); CK(hipMalloc(&boff, grid * 8 * 8));
  std::vector<unsigned long long> hh(grid * 8), bo(grid * 8);
  int64_t bucket_off_h[9];
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, 0, key, n, G, hist); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  report("D1 histogram (keys only)", n * 8.0);
  CK(hipMemcpy(hh.data(), hist, grid * 64, hipMemcpyDeviceToHost));
  { unsigned long long run = 0; for (int b = 0; b < 8; b++) { bucket_off_h[b] = (int64_t)run; for (int g = 0; g < grid; g++) { bo[g * 8 + b] = run; run += hh[g * 8 + b]; } } bucket_off_h[8] = (int64_t)run; }
  CK(hipMemcpy(boff, bo.data(), grid * 64, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemcpy(boff, bo.data(), grid * 64, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, 0, key, val, n, G, boff, okey, oval); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  report("D2 scatter 8-way (r16+w16 B/row)", n * 32.0);
  // B on the partitioned rows
  int64_t* d_boff; CK(hipMalloc(&d_boff, 9 * 8)); CK(hipMemcpy(d_boff, bucket_off_h, 72, hipMemcpyHostToDevice));
  unsigned int* cursors; CK(hipMalloc(&cursors, 8 * 32 * 4));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(acc, 0, (size_t)G * 16)); CK(hipMemset(cursors, 0, 8 * 32 * 4));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_l2_atomics, dim3(2048), dim3(256), 0, 0, okey, oval, d_boff, acc, cursors, 16384); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  report("B XCD-L2 atomics x2 (partitioned)", n * 16.0);
  // verify B: total count == n and sum matches A
  std::vector<unsigned long long> a1((size_t)G * 2), a2((size_t)G * 2);
  CK(hipMemcpy(a1.data(), acc, (size_t)G * 16, hipMemcpyDeviceToHost));
  CK(hipMemset(acc, 0, (size_t)G * 16));
  hipLaunchKernelGGL(k_dev_atomics, dim3(2048), dim3(256), 0, 0, key, val, n, acc); CK(hipDeviceSynchronize());
  CK(hipMemcpy(a2.data(), acc, (size_t)G * 16, hipMemcpyDeviceToHost));
  size_t bad = 0; unsigned long long tot = 0; for (size_t i = 0; i < a1.size(); i++) { bad += a1[i] != a2[i]; if (i & 1) tot += a1[i]; }
  printf("B verification: %zu mismatching cells, total count %llu (expect %lld)\n", bad, tot, (long long)n);
  return 0;
}
