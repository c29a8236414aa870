// How fast are random hash-table probes on MI355X as a function of the table's footprint?
// 6e8 probe keys streamed (8 B each), ~54% probe a table of 2^25 slots; 43% of slots occupied, 1% of probes hit.
//   K8  : u64 key array (268 MB)        -- what ProbeAggSink does today
//   T1  : u8 tag array (32 MB) first, key array only on a tag match
//   T2  : u16 tag array (64 MB)
//   BM  : 1 bit per possible key value (75 MB bitmap over [0, 6e8))
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro_probe.bin tools/micro_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
constexpr uint64_t ODD = 0x55fbfd6bfc5458e9ull;
constexpr int LOG2 = 25;
constexpr uint64_t EMPTY = ~0ull;

__global__ void k_gen_probe(int64_t* key, int64_t n, int64_t domain) { for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) key[i] = (int64_t)(splitmix(i) % (uint64_t)domain) + 1; }
__global__ void k_build_bloom(unsigned int* bloom, int log2_bits, int64_t nb, int64_t domain) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = splitmix(0x1234567ull + i) % (uint64_t)domain + 1;
    const uint64_t b = (key * 0x9e3779b97f4a7c15ull) >> (64 - log2_bits);
    atomicOr(&bloom[b >> 5], 1u << (b & 31));
  }
}
__global__ void k_build(unsigned long long* keys, unsigned char* t1, unsigned short* t2, unsigned int* bm, int64_t nb, int64_t domain) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = splitmix(0x1234567ull + i) % (uint64_t)domain + 1;
    uint64_t s = (key * ODD) >> (64 - LOG2);
    for (;;) { unsigned long long old = atomicCAS(&keys[s], (unsigned long long)EMPTY, (unsigned long long)key); if (old == EMPTY || old == key) break; s = (s + 1) & ((1ull << LOG2) - 1); }
    const uint64_t h = key * ODD;
    t1[s] = (unsigned char)((h & 0xff) | 1); t2[s] = (unsigned short)((h & 0xffff) | 1);
    atomicOr(&bm[key >> 5], 1u << (key & 31));
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const int64_t* __restrict__ key, int64_t n, const unsigned long long* __restrict__ keys, const unsigned char* __restrict__ t1,
                                               const unsigned short* __restrict__ t2, const unsigned int* __restrict__ bm, unsigned long long* hits, int log2_bits = 0) {
  unsigned long long h = 0;
  const uint64_t mask = (1ull << LOG2) - 1;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * blockDim.x * 2) {
    longlong2 kk;
    if (MODE == 6) { kk.x = __builtin_nontemporal_load(key + i); kk.y = __builtin_nontemporal_load(key + i + 1); }
    else kk = *reinterpret_cast<const longlong2*>(key + i);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint64_t k = (uint64_t)(r ? kk.y : kk.x);
      if ((splitmix(k) & 127) >= 69) continue;    // ~54% of rows pass the predicate
      const uint64_t hh = k * ODD;
      uint64_t s = hh >> (64 - LOG2);
      if (MODE == 3) { if (!((bm[k >> 5] >> (k & 31)) & 1)) continue; }
      if (MODE >= 4) { const uint64_t b = (k * 0x9e3779b97f4a7c15ull) >> (64 - log2_bits); if (!((bm[b >> 5] >> (b & 31)) & 1)) continue; }
      for (;;) {
        if (MODE == 1) { const unsigned char t = t1[s]; if (t == 0) break; if (t != (unsigned char)((hh & 0xff) | 1)) { s = (s + 1) & mask; continue; } }
        if (MODE == 2) { const unsigned short t = t2[s]; if (t == 0) break; if (t != (unsigned short)((hh & 0xffff) | 1)) { s = (s + 1) & mask; continue; } }
        const unsigned long long cur = keys[s];
        if (cur == k) { h++; break; }
        if (cur == EMPTY) break;
        s = (s + 1) & mask;
      }
    }
  }
  if (h) atomicAdd(hits, h);
}
// bucketized table: 8 keys per 64-byte bucket; a probe reads the whole bucket (one line access), no chains
constexpr int BLOG2 = LOG2 - 3;
__global__ void k_build_bucket(unsigned long long* bk, int64_t nb, int64_t domain, unsigned int* ovf) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = splitmix(0x1234567ull + i) % (uint64_t)domain + 1;
    uint64_t b = (key * ODD) >> (64 - BLOG2);
    bool done = false;
    for (int t = 0; t < 64 && !done; t++) {
      for (int j = 0; j < 8 && !done; j++) { unsigned long long old = atomicCAS(&bk[b * 8 + j], (unsigned long long)EMPTY, (unsigned long long)key); if (old == EMPTY || old == key) done = true; }
      if (!done) { b = (b + 1) & ((1ull << BLOG2) - 1); atomicAdd(ovf, 1u); }
    }
  }
}
__global__ __launch_bounds__(256) void k_probe_bucket(const int64_t* __restrict__ key, int64_t n, const unsigned long long* __restrict__ bk, unsigned long long* hits) {
  unsigned long long h = 0;
  const uint64_t mask = (1ull << BLOG2) - 1;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * blockDim.x * 2) {
    const longlong2 kk = *reinterpret_cast<const longlong2*>(key + i);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint64_t k = (uint64_t)(r ? kk.y : kk.x);
      if ((splitmix(k) & 127) >= 69) continue;
      uint64_t b = (k * ODD) >> (64 - BLOG2);
      for (;;) {
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(bk + b * 8);
        const ulonglong2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
        const bool hit = a0.x == k || a0.y == k || a1.x == k || a1.y == k || a2.x == k || a2.y == k || a3.x == k || a3.y == k;
        if (hit) { h++; break; }
        if (a3.y == EMPTY) break;      // slots fill left to right: last slot empty => key absent
        b = (b + 1) & mask;
      }
    }
  }
  if (h) atomicAdd(hits, h);
}
int main() {
  const int64_t n = 600000000ll, nb = 14500000, domain = 600000000ll;
  int64_t* key; unsigned long long *keys, *hits; unsigned char* t1; unsigned short* t2; unsigned int* bm;
  CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&keys, 8ull << LOG2)); CK(hipMalloc(&t1, 1ull << LOG2)); CK(hipMalloc(&t2, 2ull << LOG2)); CK(hipMalloc(&bm, domain / 8 + 64)); CK(hipMalloc(&hits, 8));
  CK(hipMemset(keys, 0xff, 8ull << LOG2)); CK(hipMemset(t1, 0, 1ull << LOG2)); CK(hipMemset(t2, 0, 2ull << LOG2)); CK(hipMemset(bm, 0, domain / 8 + 64));
  hipLaunchKernelGGL(k_gen_probe, dim3(4096), dim3(256), 0, 0, key, n, domain);
  hipLaunchKernelGGL(k_build, dim3(4096), dim3(256), 0, 0, keys, t1, t2, bm, nb, domain);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms; unsigned long long h;
  const char* names[4] = {"K8  u64 keys (268 MB)", "T1  u8 tags (32 MB) + keys", "T2  u16 tags (64 MB) + keys", "BM  domain bitmap (75 MB) + keys"};
  for (int mode = 0; mode < 4; mode++) for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(hits, 0, 8)); CK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(2048), dim3(256), 0, 0, key, n, keys, t1, t2, bm, hits);
    if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(2048), dim3(256), 0, 0, key, n, keys, t1, t2, bm, hits);
    if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(2048), dim3(256), 0, 0, key, n, keys, t1, t2, bm, hits);
    if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(2048), dim3(256), 0, 0, key, n, keys, t1, t2, bm, hits);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, hits, 8, hipMemcpyDeviceToHost));
    if (rep) printf("%-36s %8.3f ms   hits %llu\n", names[mode], ms, h);
  }
  {
    unsigned long long* bk; unsigned int* ovf; CK(hipMalloc(&bk, 8ull << LOG2)); CK(hipMalloc(&ovf, 4)); CK(hipMemset(bk, 0xff, 8ull << LOG2)); CK(hipMemset(ovf, 0, 4));
    hipLaunchKernelGGL(k_build_bucket, dim3(4096), dim3(256), 0, 0, bk, nb, domain, ovf); CK(hipDeviceSynchronize());
    unsigned int o; CK(hipMemcpy(&o, ovf, 4, hipMemcpyDeviceToHost));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(hits, 0, 8)); CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_probe_bucket, dim3(2048), dim3(256), 0, 0, key, n, bk, hits);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, hits, 8, hipMemcpyDeviceToHost));
      if (rep) printf("B8  8-key buckets (268 MB), bucket overflows %u   %8.3f ms   hits %llu\n", o, ms, h);
    }
  }
  // Bloom filters of 1..16 MB in front of the u64 key table
  unsigned int* bloom; CK(hipMalloc(&bloom, 16 << 20));
  for (int lb : {23, 24, 25, 26, 27}) for (int nt = 0; nt < 2; nt++) {
    CK(hipMemset(bloom, 0, 16 << 20));
    hipLaunchKernelGGL(k_build_bloom, dim3(4096), dim3(256), 0, 0, bloom, lb, nb, domain); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(hits, 0, 8)); CK(hipEventRecord(e0));
      if (nt) hipLaunchKernelGGL(k_probe<6>, dim3(2048), dim3(256), 0, 0, key, n, keys, t1, t2, bloom, hits, lb);
      else hipLaunchKernelGGL(k_probe<4>, dim3(2048), dim3(256), 0, 0, key, n, keys, t1, t2, bloom, hits, lb);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, hits, 8, hipMemcpyDeviceToHost));
      if (rep) printf("Bloom %3d MB %s + keys               %8.3f ms   hits %llu\n", ((1 << lb) / 8) >> 20, nt ? "nt-stream" : "         ", ms, h);
    }
  }
  return 0;
}
