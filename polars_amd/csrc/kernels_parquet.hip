// kernels_parquet.hip -- launch shells of the device Parquet decoder.  The per-thread / per-wavefront bodies live in
// parquet_device.hpp (host + device, also executed by the CPU harness of the tests); this file only maps them onto the grid.
//
//   pq_snappy          one 256-thread workgroup per compressed stream: parquet_snappy.hpp
//   pq_zstd_entropy    one wavefront per four (sixteen: no sequences) compressed zstd blocks; pq_zstd_execute one wavefront per zstd page: parquet_zstd.hpp
//   pq_page_prepare    one thread per page: split the payload into level / value streams
//   pq_count_runs      one thread per (page, stream): entries its run table needs       } the only serial walks: run HEADERS of one
//   pq_fill_runs       one thread per (page, stream): the run table                      } stream of one page
//   pq_validity        one thread per 64 rows: validity word + its popcount
//   pq_page_valid0     one thread per page: valid rows before the page (dense-slot base)
//   pq_decode          one thread per row (coalesced stores), pq_decode_bool one thread per 64 rows
//
// Bound: HBM / PCIe -- the decoded column is written once, the encoded bytes are read once or twice (Snappy output is re-read by the
// decode pass); nothing here is arithmetic.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "dev.hpp"
#include "kernels.hpp"
#include "parquet_kernels.hpp"

namespace plx {
namespace k {
using namespace pq;

// phase clock (PLX_SNAPPY_TIMING=1): thread 0 of every workgroup adds the cycles it spent per phase to dbg[0..7]
// (stage, next, mark, rank + place, point, jump, gather/direct, fence), dbg[8] = rounds, dbg[9] = elements, dbg[10] = jump sweeps
#define PQ_TICK(slot)                                      \
  if (dbg && lane == 0) {                                  \
    const uint64_t now = wall_clock64();                   \
    t_acc[slot] += now - t_last;                           \
    t_last = now;                                          \
  }

__global__ __launch_bounds__(kSnapLanes) void pq_snappy_kernel(const DecompJob* __restrict__ jobs, uint32_t n_jobs, uint32_t* __restrict__ err,
                                                               unsigned long long* __restrict__ dbg) {
  __shared__ SnapShared sh;
  if (blockIdx.x >= n_jobs) return;
  const DecompJob job = jobs[blockIdx.x];
  const uint32_t lane = threadIdx.x;
  uint64_t t_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = dbg ? wall_clock64() : 0;
  if (lane == 0) snappy_begin(sh, job);
  __syncthreads();
  while (sh.done == 0) {
    snappy_stage(sh, job, lane);
    __syncthreads();
    PQ_TICK(0)
    snappy_next(sh, job, lane);
    __syncthreads();
    PQ_TICK(1)
    for (uint32_t it = 0; it < kSnapSweeps && __syncthreads_or(snappy_mark(sh, it, lane) ? 1 : 0); it++) {}   // barrier + "does any node still have a successor"
    __syncthreads();
    PQ_TICK(2)
    snappy_rank(sh, lane);
    __syncthreads();
    if (lane == 0) snappy_scan(sh);
    __syncthreads();
    snappy_place(sh, job, lane);
    __syncthreads();
    if (lane == 0) snappy_finish(sh, job);
    __syncthreads();
    PQ_TICK(3)
    if (sh.done == 2 || sh.bad) break;      // uniform: every lane reads the flags after the barrier
    t_acc[8] += 1; t_acc[9] += sh.n_el;
    if (sh.direct) {
      snappy_direct(sh, job, lane);
    } else {
      snappy_point(sh, lane);
      __syncthreads();
      PQ_TICK(4)
      while (__syncthreads_or(snappy_jump(sh, lane) ? 1 : 0)) { t_acc[10] += 1; }   // barrier + "did any lane still follow a pointer"
      PQ_TICK(5)
      snappy_gather(sh, job, lane);
    }
    PQ_TICK(6)
    __threadfence();        // later rounds read this output from HBM: stores complete, L1 dropped
    __syncthreads();
    PQ_TICK(7)
  }
  if (lane == 0 && (sh.done == 2 || sh.bad)) atomicOr(err, (uint32_t)PE_SNAPPY);
  if (dbg && lane == 0)
    for (int i = 0; i < 11; i++) atomicAdd(dbg + i, (unsigned long long)t_acc[i]);
}

// second generation (the default; PLX_SNAPPY_KERNEL=1 selects the first): the same rounds with the batched-load bodies of next / mark / rank / jump
// (parquet_snappy.hpp); timed against the first on hardware in round 3: 42.6 vs 51.4 ms.  The phase clock is an instantiation of its own (round 5: its
// accumulators are 24 registers the decoder needs for its per-position arrays).
#define PQ_TICK2(slot)                                     \
  if constexpr (TIMING) {                                  \
    if (lane == 0) {                                       \
      const uint64_t now = wall_clock64();                 \
      t_acc[slot] += now - t_last;                         \
      t_last = now;                                        \
    }                                                      \
  }
template <bool TIMING>
__global__ __launch_bounds__(kSnapLanes, 2) void pq_snappy_kernel_v2(const DecompJob* __restrict__ jobs, uint32_t n_jobs, uint32_t* __restrict__ err,
                                                                  unsigned long long* __restrict__ dbg) {
  __shared__ SnapShared sh;
  if (blockIdx.x >= n_jobs) return;
  const DecompJob job = jobs[blockIdx.x];
  const uint32_t lane0 = threadIdx.x;
  uint64_t t_acc[TIMING ? 15 : 1] = {}, t_last = TIMING ? wall_clock64() : 0;
  (void)t_acc; (void)t_last;
  if (lane0 == 0) snappy_begin(sh, job);
  __syncthreads();
  while (sh.done == 0) {
    // the lane index is made opaque once per round: everything the phases derive from it (19 LDS addresses per array and phase) would otherwise be
    // hoisted out of this loop and held in registers for the whole kernel -- 255 registers and a spill, against 2 workgroups per CU
    uint32_t lane = lane0;
    asm volatile("" : "+v"(lane));
    snappy_stage(sh, job, lane);
    __syncthreads();
    PQ_TICK2(0)
    snappy_next_v2(sh, job, lane);
    __syncthreads();
    PQ_TICK2(1)
    for (uint32_t it = 0; it < kSnapSweeps && __syncthreads_or(snappy_mark_v2(sh, it, lane) ? 1 : 0); it++) {}   // barrier + "does any node still have a successor"
    __syncthreads();
    PQ_TICK2(2)
    snappy_rank_v2(sh, lane);
    __syncthreads();
    PQ_TICK2(11)
    snappy_scan_v2_blocks(sh, lane);
    __syncthreads();
    if (lane == 0) snappy_scan_v2_totals(sh);
    __syncthreads();
    snappy_scan_v2_offsets(sh, lane);
    __syncthreads();
    PQ_TICK2(12)
    snappy_place_v2(sh, job, lane);
    __syncthreads();
    PQ_TICK2(13)
    if (lane == 0) snappy_finish(sh, job);
    __syncthreads();
    PQ_TICK2(3)
    if (sh.done == 2 || sh.bad) break;      // uniform: every lane reads the flags after the barrier
    if constexpr (TIMING) { t_acc[8] += 1; t_acc[9] += sh.n_el; }
    if (sh.direct) {
      snappy_direct(sh, job, lane);
    } else {
      snappy_point(sh, lane);
      __syncthreads();
      PQ_TICK2(4)
      while (__syncthreads_or(snappy_jump_v2(sh, lane) ? 1 : 0)) { if constexpr (TIMING) t_acc[10] += 1; }   // barrier + "did any lane still follow a pointer"
      PQ_TICK2(5)
      snappy_gather(sh, job, lane);
    }
    PQ_TICK2(6)
    __threadfence();        // later rounds read this output from HBM: stores complete, L1 dropped
    __syncthreads();
    PQ_TICK2(7)
  }
  if (lane0 == 0 && (sh.done == 2 || sh.bad)) atomicOr(err, (uint32_t)PE_SNAPPY);
  if constexpr (TIMING) {
    if (lane0 == 0)
      for (int i = 0; i < 15; i++) atomicAdd(dbg + i, (unsigned long long)t_acc[i]);
  }
}

__global__ __launch_bounds__(kBlock) void pq_page_prepare_kernel(PageDesc* __restrict__ pages, uint32_t n_pages, uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pages) return;
  PageDesc p = pages[i];
  const uint32_t e = page_prepare(p);
  pages[i] = p;
  if (e) atomicOr(err, e);
}

__global__ __launch_bounds__(kBlock) void pq_count_runs_kernel(const PageDesc* __restrict__ pages, uint32_t n_pages, int levels, uint32_t* __restrict__ counts,
                                                               uint32_t* __restrict__ err) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2 * n_pages) return;
  uint32_t e = 0;
  const int s = (int)(t & 1);
  counts[t] = (s == 0 && !levels) ? 0u : stream_entries(pages[t >> 1], s, &e);
  if (e) atomicOr(err, e);
}

__global__ __launch_bounds__(kBlock) void pq_fill_runs_kernel(const PageDesc* __restrict__ pages, uint32_t n_pages, const uint64_t* __restrict__ offs,
                                                              RunEntry* __restrict__ runs) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2 * n_pages) return;
  const uint32_t n = (uint32_t)(offs[t + 1] - offs[t]);
  if (n) stream_fill(pages[t >> 1], (int)(t & 1), runs + offs[t], n);
}

__global__ __launch_bounds__(kBlock) void pq_validity_kernel(const PageDesc* __restrict__ pages, uint32_t n_pages, const RunEntry* __restrict__ runs,
                                                             const uint64_t* __restrict__ offs, uint64_t n_rows, uint64_t* __restrict__ validity,
                                                             uint32_t* __restrict__ popc, uint32_t* __restrict__ err) {
  const uint64_t n_words = (n_rows + 63) >> 6;
  for (uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock) {
    uint32_t e = 0;
    const uint64_t word = validity_word(pages, n_pages, runs, offs, n_rows, w, &e);
    validity[w] = word;
    popc[w] = (uint32_t)__popcll(word);
    if (e) atomicOr(err, e);
  }
}

__global__ __launch_bounds__(kBlock) void pq_page_valid0_kernel(PageDesc* __restrict__ pages, uint32_t n_pages, const uint64_t* __restrict__ validity,
                                                                const uint64_t* __restrict__ word_prefix) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pages) return;
  pages[i].valid0 = valid_before(validity, word_prefix, pages[i].row0);
}

__global__ __launch_bounds__(kBlock) void pq_decode_kernel(ColumnDecode c, void* __restrict__ out, uint32_t out_width, uint32_t* __restrict__ err) {
  uint32_t e = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < c.n_rows; r += (uint64_t)gridDim.x * kBlock) decode_rows(c, out, out_width, r, r + 1, &e);
  if (e) atomicOr(err, e);
}

__global__ __launch_bounds__(kBlock) void pq_decode_bool_kernel(ColumnDecode c, uint64_t* __restrict__ out, uint32_t* __restrict__ err) {
  const uint64_t n_words = (c.n_rows + 63) >> 6;
  uint32_t e = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock) out[w] = decode_bool_word(c, w, &e);
  if (e) atomicOr(err, e);
}

static unsigned blocks_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

void pq_snappy(const DecompJob* jobs, uint32_t n_jobs, uint64_t bytes_out, uint32_t* err) {
  if (!n_jobs) return;
  static const bool timing = [] { const char* e = getenv("PLX_SNAPPY_TIMING"); return e && e[0] == '1'; }();
  static const bool v2 = [] { const char* e = getenv("PLX_SNAPPY_KERNEL"); return !(e && e[0] == '1'); }();
  Buf dbg;
  if (timing) dbg = dev_alloc_zero(15 * 8);
  {
    ProfileScope ps("pq_snappy", bytes_out * 2, n_jobs);
    // generation 2 (batched LDS loads) is the default since round 3: 42.6 vs 51.4 ms of pq_snappy on the 2e7-row file (gpurun_out/r03a), bit-identical
    // output on every stream of the GPU and CPU suites; PLX_SNAPPY_KERNEL=1 selects the first generation
    if (v2 && timing) hipLaunchKernelGGL(pq_snappy_kernel_v2<true>, dim3(n_jobs), dim3(kSnapLanes), 0, stream(), jobs, n_jobs, err, dbg->as<unsigned long long>());
    else if (v2) hipLaunchKernelGGL(pq_snappy_kernel_v2<false>, dim3(n_jobs), dim3(kSnapLanes), 0, stream(), jobs, n_jobs, err, (unsigned long long*)nullptr);
    else hipLaunchKernelGGL(pq_snappy_kernel, dim3(n_jobs), dim3(kSnapLanes), 0, stream(), jobs, n_jobs, err, timing ? dbg->as<unsigned long long>() : nullptr);
    PLX_HIP(hipGetLastError());
  }
  if (timing) {
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pq_snappy_kernel_v2<false>, (int)kSnapLanes, 0);
    fprintf(stderr, "[pq_snappy] workgroups per CU: %d; ", per_cu);
    unsigned long long h[15];
    d2h_sync(h, dbg->ptr, sizeof h);
    static const char* names[8] = {"stage", "next", "mark", "rank_place", "point", "jump", "gather", "fence"};
    fprintf(stderr, "[pq_snappy] streams=%u out=%.1f MB rounds=%llu elements=%llu jump_sweeps=%llu; 100 MHz ticks summed over streams:", n_jobs, bytes_out / 1e6, h[8], h[9], h[10]);
    const unsigned long long finish = h[3];
    if (v2) h[3] += h[11] + h[12] + h[13];       // the second generation clocks the four steps of rank_place separately
    for (int i = 0; i < 8; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
    if (v2) fprintf(stderr, " (rank_place = rank %llu + scan %llu + place %llu + finish %llu)", h[11], h[12], h[13], finish);
    fprintf(stderr, "\n");
  }
}
// ---- zstd (parquet_zstd.hpp): the wavefront of its bodies, and the two passes -----------------------------------------------------------------------------------
// TIMING (PLX_ZSTD_TIMING=1): lane 0 adds the 100 MHz ticks between two tick() calls to slot t_acc[slot]; count() adds to a counter slot
template <bool TIMING> struct ZstdDevWave {
  uint64_t t_acc[TIMING ? 8 : 1] = {};
  uint64_t t_last = TIMING ? wall_clock64() : 0;
  template <class F> __device__ __forceinline__ void lanes(F&& f) { f((uint32_t)threadIdx.x); }
  __device__ __forceinline__ void sync() { __syncthreads(); }
  __device__ __forceinline__ void wave_fence() { __builtin_amdgcn_wave_barrier(); }     // orders the wavefront's LDS accesses for the compiler; the hardware runs them in order
  __device__ __forceinline__ uint32_t uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }      // a value every lane holds -> a scalar register
  __device__ __forceinline__ void tick(int slot) {
    if constexpr (TIMING) { const uint64_t now = wall_clock64(); t_acc[slot] += now - t_last; t_last = now; }
  }
  __device__ __forceinline__ void count(int slot, uint32_t n) {          // called by the whole wavefront with one value, or by the one lane that owns the value
    if constexpr (TIMING) t_acc[slot] += (__builtin_popcountll(__ballot(1)) == 64 && threadIdx.x != 0) ? 0 : n;
  }
  __device__ __forceinline__ void report(unsigned long long* dbg) {
    if constexpr (TIMING) {
      for (int i = 5; i < 8; i++) {          // counters: summed over the lanes that counted (clock slots are lane 0's)
        uint32_t v = (uint32_t)t_acc[i];
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        t_acc[i] = v;
      }
      if (threadIdx.x == 0)
        for (int i = 0; i < 8; i++) atomicAdd(dbg + i, (unsigned long long)t_acc[i]);
    }
  }
  // a[lane] -> the sum of a[0 .. lane) (callers put barriers around it)
  __device__ __forceinline__ void exclusive_scan(uint32_t* a) {
    const uint32_t lane = threadIdx.x, v = a[lane];
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d, 64);
      if ((int)lane >= d) x += y;
    }
    a[lane] = x - v;
  }
  // repeat-offset maps (parquet_zstd.hpp): lane's map -> the composition of the maps of lanes 0 .. lane (inclusive)
  __device__ __forceinline__ void rep_scan(uint32_t* r0, uint32_t* r1, uint32_t* r2) {
    const uint32_t lane = threadIdx.x;
    ZstdRepMap m;
    m.s[0] = r0[lane]; m.s[1] = r1[lane]; m.s[2] = r2[lane];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      ZstdRepMap e;
      e.s[0] = __shfl_up(m.s[0], d, 64); e.s[1] = __shfl_up(m.s[1], d, 64); e.s[2] = __shfl_up(m.s[2], d, 64);
      if ((int)lane >= d) m = zstd_rep_compose(e, m);
    }
    r0[lane] = m.s[0]; r1[lane] = m.s[1]; r2[lane] = m.s[2];
  }
  // the three scans of a batch in one go: five shuffles a round (one LDS-crossbar latency), everything else in registers
  __device__ __forceinline__ void batch_scan(uint32_t* lit, uint32_t* out, uint32_t* r0, uint32_t* r1, uint32_t* r2) {
    const uint32_t lane = threadIdx.x, l0 = lit[lane], o0 = out[lane];
    uint32_t l = l0, o = o0;
    ZstdRepMap m;
    m.s[0] = r0[lane]; m.s[1] = r1[lane]; m.s[2] = r2[lane];
    // a batch of nothing but "the last offset again" (sorted values: every sequence) composes to the identity: no map needs to travel
    const bool maps = __ballot(m.s[0] != (1u << 30) || m.s[1] != (2u << 30) || m.s[2] != (3u << 30)) != 0;
    if (maps) {
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t yl = __shfl_up(l, d, 64), yo = __shfl_up(o, d, 64);
        ZstdRepMap e;
        e.s[0] = __shfl_up(m.s[0], d, 64); e.s[1] = __shfl_up(m.s[1], d, 64); e.s[2] = __shfl_up(m.s[2], d, 64);
        if ((int)lane >= d) { l += yl; o += yo; m = zstd_rep_compose(e, m); }
      }
      r0[lane] = m.s[0]; r1[lane] = m.s[1]; r2[lane] = m.s[2];
    } else {
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t yl = __shfl_up(l, d, 64), yo = __shfl_up(o, d, 64);
        if ((int)lane >= d) { l += yl; o += yo; }
      }
    }
    lit[lane] = l - l0; out[lane] = o - o0;
  }
  template <class F> __device__ __forceinline__ uint64_t ballot(F&& pred) { return (uint64_t)__ballot(pred((uint32_t)threadIdx.x) ? 1 : 0); }
  // index of the first nonzero flag[lane]; 64 if there is none
  __device__ __forceinline__ uint32_t first_flag(const uint32_t* flag) {
    const unsigned long long m = __ballot(flag[threadIdx.x] != 0);
    return m ? (uint32_t)__ffsll((long long)m) - 1 : 64u;
  }
};
// one wavefront per kZHufGroups compressed blocks without sequences (the first n_huf of the index list), then one per kZGroups of the others (each part longest blocks
// first, so a wavefront's blocks are of a size)
template <bool TIMING>
__global__ __launch_bounds__(kZLanes) void pq_zstd_entropy_kernel(ZstdBlock* __restrict__ blocks, const uint32_t* __restrict__ order, uint32_t n, uint32_t n_huf, uint32_t waves_huf,
                                                                  const ZstdHufDesc* __restrict__ hufs, const ZstdFseDesc* __restrict__ fses, unsigned long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char zstd_lds[];
  ZstdDevWave<TIMING> w;
  if (blockIdx.x < waves_huf) {
    zstd_huf_group(w, (ZstdHufShared*)zstd_lds, blocks, order, blockIdx.x * kZHufGroups, n_huf, hufs);
  } else {
    const uint32_t first = (blockIdx.x - waves_huf) * kZGroups;
    if (first >= n - n_huf) return;
    zstd_entropy_group(w, (ZstdEntropyShared*)zstd_lds, blocks, order + n_huf, first, n - n_huf, hufs, fses);
  }
  w.report(dbg);
}
// one wavefront per page
template <bool TIMING>
__global__ __launch_bounds__(kZLanes) void pq_zstd_execute_kernel(const ZstdStream* __restrict__ streams, uint32_t n, const ZstdBlock* __restrict__ blocks, uint32_t* __restrict__ err,
                                                                  unsigned long long* __restrict__ dbg) {
  __shared__ ZstdExecShared sh;
  if (blockIdx.x >= n) return;
  ZstdDevWave<TIMING> w;
  const ZstdStream s = streams[blockIdx.x];
  const bool ok = zstd_exec_stream(w, sh, s, blocks);
  if (!ok && threadIdx.x == 0) atomicOr(err, (uint32_t)PE_ZSTD);
  w.report(dbg + 8);
}
void pq_zstd(ZstdBlock* blocks, const uint32_t* order, uint32_t n_compressed, uint32_t n_huf_only, const ZstdHufDesc* hufs, const ZstdFseDesc* fses, const ZstdStream* streams, uint32_t n_streams,
             uint64_t bytes_in, uint64_t bytes_out, uint32_t* err) {
  static const bool timing = [] { const char* e = getenv("PLX_ZSTD_TIMING"); return e && e[0] == '1'; }();
  Buf dbg;
  if (timing) dbg = dev_alloc_zero(16 * 8);
  unsigned long long* d = timing ? dbg->as<unsigned long long>() : nullptr;
  if (n_compressed) {
    ProfileScope ps("pq_zstd_entropy", bytes_in, n_compressed);
    const uint32_t waves_huf = (n_huf_only + kZHufGroups - 1) / kZHufGroups, grid = waves_huf + (n_compressed - n_huf_only + kZGroups - 1) / kZGroups;
    const size_t lds = std::max(kZGroups * sizeof(ZstdEntropyShared), kZHufGroups * sizeof(ZstdHufShared));        // 4 x 17 KB / 16 x 4.6 KB: beyond the 64 KB a kernel gets without asking
    static bool attr_set[2] = {false, false};
    if (!attr_set[timing ? 1 : 0]) {
      if (timing) (void)hipFuncSetAttribute((const void*)pq_zstd_entropy_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      else (void)hipFuncSetAttribute((const void*)pq_zstd_entropy_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipGetLastError();
      attr_set[timing ? 1 : 0] = true;
    }
    if (timing) hipLaunchKernelGGL(pq_zstd_entropy_kernel<true>, dim3(grid), dim3(kZLanes), lds, stream(), blocks, order, n_compressed, n_huf_only, waves_huf, hufs, fses, d);
    else hipLaunchKernelGGL(pq_zstd_entropy_kernel<false>, dim3(grid), dim3(kZLanes), lds, stream(), blocks, order, n_compressed, n_huf_only, waves_huf, hufs, fses, d);
    PLX_HIP(hipGetLastError());
  }
  if (n_streams) {
    ProfileScope ps("pq_zstd_execute", bytes_out * 2, n_streams);
    if (timing) hipLaunchKernelGGL(pq_zstd_execute_kernel<true>, dim3(n_streams), dim3(kZLanes), 0, stream(), streams, n_streams, (const ZstdBlock*)blocks, err, d);
    else hipLaunchKernelGGL(pq_zstd_execute_kernel<false>, dim3(n_streams), dim3(kZLanes), 0, stream(), streams, n_streams, (const ZstdBlock*)blocks, err, d);
    PLX_HIP(hipGetLastError());
  }
  if (timing) {
    unsigned long long h[16];
    d2h_sync(h, dbg->ptr, sizeof h);
    fprintf(stderr, "[pq_zstd] blocks=%u pages=%u in=%.1f MB out=%.1f MB; 100 MHz ticks summed over wavefronts: entropy huf_build=%llu huf_decode=%llu fse_build=%llu stage=%llu seq=%llu (sequences=%llu) | "
            "execute plan=%llu room=%llu literals=%llu matches=%llu long/raw=%llu (sequences in batches=%llu, batches resolved then copied at once=%llu, batches=%llu)\n",
            n_compressed, n_streams, bytes_in / 1e6, bytes_out / 1e6, h[0], h[1], h[2], h[3], h[4], h[5], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
  }
}
void pq_page_prepare(PageDesc* pages, uint32_t n_pages, uint32_t* err) {
  if (!n_pages) return;
  hipLaunchKernelGGL(pq_page_prepare_kernel, dim3(blocks_for(n_pages)), dim3(kBlock), 0, stream(), pages, n_pages, err);
  PLX_HIP(hipGetLastError());
}
void pq_count_runs(const PageDesc* pages, uint32_t n_pages, bool levels, uint32_t* counts, uint32_t* err) {
  if (!n_pages) return;
  ProfileScope ps("pq_count_runs", 0, n_pages);
  hipLaunchKernelGGL(pq_count_runs_kernel, dim3(blocks_for(2ull * n_pages)), dim3(kBlock), 0, stream(), pages, n_pages, levels ? 1 : 0, counts, err);
  PLX_HIP(hipGetLastError());
}
void pq_fill_runs(const PageDesc* pages, uint32_t n_pages, const uint64_t* offs, RunEntry* runs) {
  if (!n_pages) return;
  ProfileScope ps("pq_fill_runs", 0, n_pages);
  hipLaunchKernelGGL(pq_fill_runs_kernel, dim3(blocks_for(2ull * n_pages)), dim3(kBlock), 0, stream(), pages, n_pages, offs, runs);
  PLX_HIP(hipGetLastError());
}
void pq_validity(const PageDesc* pages, uint32_t n_pages, const RunEntry* runs, const uint64_t* offs, uint64_t n_rows, uint64_t* validity, uint32_t* popc, uint32_t* err) {
  const uint64_t n_words = (n_rows + 63) >> 6;
  if (!n_words) return;
  ProfileScope ps("pq_validity", n_words * 12, n_rows);
  hipLaunchKernelGGL(pq_validity_kernel, dim3(grid_for((int64_t)n_words, kBlock)), dim3(kBlock), 0, stream(), pages, n_pages, runs, offs, n_rows, validity, popc, err);
  PLX_HIP(hipGetLastError());
}
void pq_page_valid0(PageDesc* pages, uint32_t n_pages, const uint64_t* validity, const uint64_t* word_prefix) {
  if (!n_pages) return;
  hipLaunchKernelGGL(pq_page_valid0_kernel, dim3(blocks_for(n_pages)), dim3(kBlock), 0, stream(), pages, n_pages, validity, word_prefix);
  PLX_HIP(hipGetLastError());
}
void pq_decode(const ColumnDecode& c, void* out, uint32_t out_width, uint64_t encoded_bytes, uint32_t* err) {
  if (!c.n_rows) return;
  if (out_width == 0) {
    ProfileScope ps("pq_decode_bool", encoded_bytes + c.n_rows / 8, c.n_rows);
    hipLaunchKernelGGL(pq_decode_bool_kernel, dim3(grid_for((int64_t)((c.n_rows + 63) >> 6), kBlock)), dim3(kBlock), 0, stream(), c, (uint64_t*)out, err);
  } else {
    ProfileScope ps("pq_decode", encoded_bytes + c.n_rows * out_width, c.n_rows);
    hipLaunchKernelGGL(pq_decode_kernel, dim3(grid_for((int64_t)c.n_rows, kBlock, 16)), dim3(kBlock), 0, stream(), c, out, out_width, err);
  }
  PLX_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace plx
