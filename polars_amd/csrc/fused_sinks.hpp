// fused_sinks.hpp -- the sinks of the fused scan kernel and the kernel body itself (device-only header: compiled
// ahead of time by kernels_fused.hip for the benchmark shapes and the generic interpreter, and at run time by
// hiprtc (jit.cpp) for any other program shape).  See kernels_fused.hip for the design notes.
#pragma once
#include <type_traits>

#include "fused_device.hpp"

namespace plx {
namespace k {

// ---- sink: per-lane register accumulators (no group-by) ---------------------------------
struct RegAggSink {
  struct Params { unsigned long long* partials; };  // [grid][kMaxAggs]
  uint64_t acc[kMaxAggs];
  template <class S> __device__ __forceinline__ void init(const S& sh, const Params&) {
#pragma unroll
    for (int k = 0; k < kMaxAggs; k++) acc[k] = (k < sh.n_aggs) ? agg_identity_dev(sh.aggs[k].kind) : 0ull;
  }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params&) {
#pragma unroll
    for (int r = 0; r < kRows; r++) {
#pragma unroll
      for (int k = 0; k < kMaxAggs; k++) {
        if (k < sh.n_aggs) {
          const Agg ag = sh.aggs[k];
          uint64_t v = rf.get(r, ag.src);
          bool valid = (rf.getv(ag.src) >> r) & 1;
          acc[k] = agg_combine(ag.kind, acc[k], agg_row_value(ag.kind, v, pass[r], valid, (uint64_t)(row0 + r)));
        }
      }
    }
  }
  template <class S> __device__ __forceinline__ void finish(const S& sh, const Params& p) {
    __shared__ uint64_t sh_acc[kBlock / 64][kMaxAggs];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kMaxAggs; k++) {
      if (k < sh.n_aggs) {
        uint64_t x = acc[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x = agg_combine(sh.aggs[k].kind, x, shfl_xor_u64(x, m));
        if (lane == 0) sh_acc[wave][k] = x;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < sh.n_aggs) {
      const int k = threadIdx.x;
      uint64_t x = sh_acc[0][k];
      for (int w = 1; w < (int)(blockDim.x >> 6); w++) x = agg_combine(sh.aggs[k].kind, x, sh_acc[w][k]);
      p.partials[(size_t)blockIdx.x * kMaxAggs + k] = x;
    }
  }
};

// ---- sink: workgroup-shared LDS table for dense group ids (1 < G <= ~1024) -----------------
// Layout lds[(g * n_aggs + k) * C + copy], copy = lane & (C-1).  With C = 16 every
// 16-lane group of a 64-bit DS instruction touches 16 distinct cells = all 32 banks once,
// so the LDS atomics run conflict-free at full rate no matter how skewed the groups are
// (TPC-H Q1: ~50% of rows fall in one group).  At the end the copies are folded and one
// partial per workgroup is written (G <= 64) or added to the HBM table with atomics.
struct LdsAggSink {
  struct Params {
    unsigned long long* partials;    // [grid][G][n_aggs]  (when global_acc == nullptr)
    unsigned long long* global_acc;  // [G][n_aggs] device-scope atomics (large G)
    int n_groups;
    int copies;                      // power of two
    unsigned int* oob;               // may be null: [0] = 1 when a group id fell outside the table (key bounds that were declared or assumed, not measured, and wrong)
  };
  template <class S> __device__ __forceinline__ void init(const S& sh, const Params& p) {
    extern __shared__ unsigned long long lds_tbl[];
    const int cells = p.n_groups * sh.n_aggs;
    for (int i = threadIdx.x; i < cells * p.copies; i += blockDim.x) lds_tbl[i] = agg_identity_dev(sh.aggs[(i / p.copies) % sh.n_aggs].kind);
    __syncthreads();
  }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    extern __shared__ unsigned long long lds_tbl[];
    const int copy = lane_id() & (p.copies - 1);
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r]) continue;
      const uint64_t gid64 = rf.get(r, sh.key);
      if (gid64 >= (uint64_t)p.n_groups) { if (p.oob) *p.oob = 1u; continue; }   // only possible when declared / assumed column bounds were wrong: stay inside the table, tell the host (the query fails or is planned again)
      const uint32_t gid = (uint32_t)gid64;
      unsigned long long* cells = lds_tbl + (size_t)gid * sh.n_aggs * p.copies + copy;
#pragma unroll
      for (int k = 0; k < kMaxAggs; k++) {
        if (k < sh.n_aggs) {
          const Agg ag = sh.aggs[k];
          uint64_t v = rf.get(r, ag.src);
          bool valid = (rf.getv(ag.src) >> r) & 1;
          if ((ag.kind == AGG_SUM_F || ag.kind == AGG_SUM_I || ag.kind == AGG_COUNT) && !valid) continue;
          uint64_t x = agg_row_value(ag.kind, v, true, valid, (uint64_t)(row0 + r));
          lds_atomic_agg(ag.kind, cells + k * p.copies, x);
        }
      }
    }
  }
  template <class S> __device__ __forceinline__ void finish(const S& sh, const Params& p) {
    extern __shared__ unsigned long long lds_tbl[];
    __syncthreads();
    const int cells = p.n_groups * sh.n_aggs;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
      const uint8_t kind = sh.aggs[i % sh.n_aggs].kind;
      uint64_t x = lds_tbl[(size_t)i * p.copies];
      for (int c = 1; c < p.copies; c++) x = agg_combine(kind, x, lds_tbl[(size_t)i * p.copies + c]);
      if (p.global_acc) { if (x != agg_identity_dev(kind) || kind == AGG_SUM_F) atomic_agg(kind, p.global_acc + i, x); }
      else p.partials[(size_t)blockIdx.x * cells + i] = x;
    }
  }
};

// ---- sink: direct-address table (dense integer keys) ---------------------------------
struct DenseAggSink {
  using Params = DenseTable;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r]) continue;
      bool kvalid = (rf.getv(sh.key) >> r) & 1;
      int64_t g = kvalid ? ((int64_t)rf.get(r, sh.key) - p.key_min) : p.n_groups;
      // see LdsAggSink: wrong declared / assumed bounds must not leave the table -- nor may a VALID key land in the null key's cell (id == n_groups)
      if ((uint64_t)g > (uint64_t)p.n_groups || (kvalid && g == p.n_groups)) { if (p.oob) *p.oob = 1u; continue; }
      atomic_row(sh, rf, r, row0 + r, p.acc + (size_t)g * sh.n_aggs);
    }
  }
};

// ---- sink: open-addressing hash table in HBM -----------------------------------------
// slot = top bits of key * RANDOM_ODD (the reference's DirtyHash,
// polars-utils/src/hashing.rs:124-151: "only the top bits are decent"), linear probing,
// 64-bit CAS claims a slot, payload updated with device-scope atomics.
struct HashAggSink {
  using Params = HashTable;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  __device__ __forceinline__ static int64_t find_slot(const Params& p, uint64_t key) {
    const uint64_t cap = 1ull << p.log2_cap;
    uint64_t slot = (key * 0x55fbfd6bfc5458e9ull) >> (64 - p.log2_cap);
    for (uint32_t probe = 0; probe < p.max_probe; probe++) {
      unsigned long long cur = p.keys[slot];
      if (cur == key) return (int64_t)slot;
      if (cur == kEmptyKey) {
        unsigned long long old = atomicCAS(&p.keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (old == kEmptyKey || old == key) return (int64_t)slot;
      }
      slot = (slot + 1) & (cap - 1);
    }
    atomicExch(p.overflow, 1u);
    return -1;
  }
  // wave_combine: up to kCombineRounds (8) times the first unprocessed lane's key is broadcast, the lanes holding the same key reduce
  // every aggregate with a shuffle tree and ONE of them keeps the total; then all such leaders update the table together, and
  // the rows no round reached go the ordinary way.  A key that holds a large share of the rows is almost surely picked in the
  // first rounds (share f: missed with probability (1 - f)^rounds), which is all this is for: the planner's 2^20-row sample must
  // not issue half a million device atomics on one address.  No memory operation inside the rounds (a table probe there would
  // serialise its latency round after round).
  static constexpr int kCombineRounds = 8;
  template <class S, class RF> __device__ __forceinline__ void consume_combined(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    const uint64_t cap = 1ull << p.log2_cap;
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      const bool kvalid = (rf.getv(sh.key) >> r) & 1;
      const uint64_t key = kvalid ? rf.get(r, sh.key) : 0ull;
      uint64_t xv[kMaxAggs];      // this row's contribution to every aggregate (the program is a compile-time constant out here)
#pragma unroll
      for (int k = 0; k < kMaxAggs; k++) {
        xv[k] = 0;
        if (k < sh.n_aggs) { const Agg ag = sh.aggs[k]; xv[k] = agg_row_value(ag.kind, rf.get(r, ag.src), true, (rf.getv(ag.src) >> r) & 1, (uint64_t)(row0 + r)); }
      }
      bool reached = false, keeps_total = false;
      uint64_t todo = ballot(pass[r]);
#pragma unroll 1
      for (int round = 0; round < kCombineRounds && todo; round++) {
        const int leader = (int)__builtin_ctzll(todo);
        const uint64_t lkey = shfl_u64(key, leader);
        const bool lvalid = __shfl((int)kvalid, leader, 64) != 0;
        const bool member = pass[r] && !reached && kvalid == lvalid && key == lkey;
        todo &= ~ballot(member);
#pragma unroll
        for (int k = 0; k < kMaxAggs; k++) {
          if (k < sh.n_aggs) {
            uint64_t x = member ? xv[k] : agg_identity_dev(sh.aggs[k].kind);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) x = agg_combine(sh.aggs[k].kind, x, shfl_xor_u64(x, m));
            if (lane == leader) xv[k] = x;          // the leader's own contribution becomes the group's
          }
        }
        if (member) reached = true;
        if (lane == leader) keeps_total = true;
      }
      if (!pass[r] || (reached && !keeps_total)) continue;      // another lane carries this row
      int64_t slot;
      if (!kvalid) { slot = (int64_t)cap; p.keys[cap] = 0; }
      else if (key == kEmptyKey) { slot = (int64_t)cap + 1; p.keys[cap + 1] = 0; }
      else slot = find_slot(p, key);
      if (slot < 0) continue;
#pragma unroll
      for (int k = 0; k < kMaxAggs; k++) {
        if (k < sh.n_aggs) {
          const uint8_t kind = sh.aggs[k].kind;
          if (xv[k] != agg_identity_dev(kind) || kind == AGG_SUM_F) atomic_agg(kind, p.acc + (size_t)slot * sh.n_aggs + k, xv[k]);
        }
      }
    }
  }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    if (p.wave_combine) { consume_combined(sh, rf, pass, row0, p); return; }      // uniform branch (kernel argument); also in the interpreter: a sample block is below the JIT's row threshold
    const uint64_t cap = 1ull << p.log2_cap;
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r]) continue;
      bool kvalid = (rf.getv(sh.key) >> r) & 1;
      uint64_t key = rf.get(r, sh.key);
      int64_t slot;
      if (!kvalid) { slot = (int64_t)cap; p.keys[cap] = 0; }
      else if (key == kEmptyKey) { slot = (int64_t)cap + 1; p.keys[cap + 1] = 0; }
      else slot = find_slot(p, key);
      if (slot < 0) continue;
      atomic_row(sh, rf, r, row0 + r, p.acc + (size_t)slot * sh.n_aggs);
    }
  }
};


// ---- sink: wide-key open-addressing table (keys of 2..4 columns that do not bit-pack) ----------
// Slot protocol: tags[s] goes EMPTY -> tag|BUSY (64-bit CAS) -> tag.  The claimer writes the key
// words with write-through (agent-scope) stores, drains them, then publishes the tag; readers
// load tag and words at agent scope (L2-served, never a stale L1 line).  Within one loop
// iteration the claim+publish code precedes the wait, so a lane never waits on a lane of its
// own wave that has not published yet; owners in other waves make progress independently.
struct WideAggSink {
  using Params = WideTable;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  __device__ __forceinline__ static uint64_t ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ static void st(unsigned long long* p, uint64_t v) { __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ static uint64_t mix(uint64_t h, uint64_t w) {
    h ^= w; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; return h;
  }
  __device__ __forceinline__ static int64_t find_or_insert(const Params& p, const uint64_t w[kMaxKeys + 1]) {
    const uint64_t cap = 1ull << p.log2_cap;
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (uint32_t j = 0; j < p.n_words; j++) h = mix(h, w[j]);
    h *= 0x55fbfd6bfc5458e9ull;
    uint64_t tag = h & ~kBusyBit;
    if (tag == (kEmptyKey & ~kBusyBit)) tag ^= 1;
    uint64_t slot = h >> (64 - p.log2_cap);
    int64_t found = -1;
    for (uint32_t probe = 0; probe < p.max_probe && found < 0; probe++) {
      uint64_t cur = ld(&p.tags[slot]);
      bool claimed = false;
      if (cur == kEmptyKey) {
        const unsigned long long old = atomicCAS(&p.tags[slot], (unsigned long long)kEmptyKey, (unsigned long long)(tag | kBusyBit));
        if (old == kEmptyKey) claimed = true; else cur = old;
      }
      if (claimed) {  // publish: words (write-through) -> drain -> tag
        for (uint32_t j = 0; j < p.n_words; j++) st(&p.words[(size_t)j * cap + slot], w[j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st(&p.tags[slot], tag);
        found = (int64_t)slot;
      }
      // every lane of the wave is past the publish before any lane starts to wait (convergent
      // marker: the two branches cannot be merged into an if/else whose else side runs first)
      __builtin_amdgcn_wave_barrier();
      if (!claimed && (cur & ~kBusyBit) == tag) {
        while (cur & kBusyBit) { __builtin_amdgcn_s_sleep(1); cur = ld(&p.tags[slot]); }
        asm volatile("" ::: "memory");
        bool same = true;
        for (uint32_t j = 0; j < p.n_words; j++) same = same && (ld(&p.words[(size_t)j * cap + slot]) == w[j]);
        if (same) found = (int64_t)slot;
      }
      slot = (slot + 1) & (cap - 1);
    }
    if (found < 0) atomicExch(p.overflow, 1u);
    return found;
  }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r]) continue;
      uint64_t w[kMaxKeys + 1];
      uint64_t nullmask = 0;
#pragma unroll
      for (int j = 0; j < kMaxKeys; j++) {
        w[j] = 0;
        if (j < sh.n_keys) {
          const bool kvalid = (rf.getv(sh.keys[j]) >> r) & 1;
          w[j] = kvalid ? rf.get(r, sh.keys[j]) : 0ull;
          if (!kvalid) nullmask |= 1ull << j;
        }
      }
      w[kMaxKeys] = 0;
      if (p.has_null_word) w[sh.n_keys] = nullmask;
      const int64_t slot = find_or_insert(p, w);
      if (slot < 0) continue;
      atomic_row(sh, rf, r, row0 + r, p.acc + (size_t)slot * sh.n_aggs);
    }
  }
};


// ---- sinks: fused join build / probe->aggregate ---------------------------------------------
// Replaces, for `GroupBy(join key + build-side columns) over Join(inner)`, the chain
// build_tables -> probe_inner -> gather of every payload column -> group_by of the reference
// (polars-ops/src/frame/join/hash_join/single_keys.rs:16-167, single_keys_inner.rs:11-149,
// polars-mem-engine/src/executors/{join.rs:41-121, group_by.rs:60-98}): no filtered frames,
// no (left_idx, right_idx) pairs and no joined frame are materialised.
struct JoinBuildSink {
  using Params = JoinAggTable;
  unsigned long long n = 0;      // rows this lane inserted
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) { n = 0; }
  template <class S> __device__ __forceinline__ void finish(const S&, const Params& p) {
    if (!p.count) return;
    const uint64_t w = wave_sum_u64(n);
    if (lane_id() == 0 && w) atomicAdd(p.count, (unsigned long long)w);
  }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    const uint64_t cap = 1ull << p.log2_cap;
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r] || !((rf.getv(sh.key) >> r) & 1)) continue;  // null keys never match
      n++;
      const uint64_t key = rf.get(r, sh.key);
      // One atomic per build row: the CAS winner owns the slot and stores its row with a plain store;
      // meeting the same key again means the build keys are not unique -> flag, the caller falls back.
      if (key == kEmptyKey) {
        const unsigned int old = atomicExch(jt_row(p, cap), (unsigned int)(row0 + r));
        if (p.links) { p.links[row0 + r] = (unsigned long long)old | ((unsigned long long)(row0 + r) << 32); if (old != kNoRow32) atomicAdd(jt_row(p, cap) + 1, 1u); }
        else if (old != kNoRow32) p.flags[0] = 1u;
        continue;
      }
      uint64_t slot = (key * kP2HashMult) >> (64 - p.log2_cap);
      if (p.links) {
        // multi-value mode (wave-uniform branch): find or claim the key's slot, then push this row onto the front of the key's chain
        for (uint32_t probe = 0;; probe++) {
          const unsigned long long old = atomicCAS(jt_key(p, slot), (unsigned long long)kEmptyKey, (unsigned long long)key);
          if (old == kEmptyKey || old == key) {
            const unsigned int prev = atomicExch(jt_row(p, slot), (unsigned int)(row0 + r));
            p.links[row0 + r] = (unsigned long long)prev | ((unsigned long long)(row0 + r) << 32);      // its own representative until canonicalise_chains says otherwise
            if (old == key) atomicAdd(jt_row(p, slot) + 1, 1u);                                           // marks the slot as one with duplicates (the word starts at 0xffffffff)
            break;
          }
          slot = jt_next(p, slot);
          if (probe > (1u << 16)) { p.flags[1] = 1u; break; }
        }
        continue;
      }
      // CAS first: the table is at most half full and build keys are (expected to be) unique, so the home slot is usually free -- a read before the CAS
      // would be a second trip across the fabric for nothing (SF100 Q3 on hashed keys: 1.5e7 inserts into a 400 MB table)
      for (uint32_t probe = 0;; probe++) {
        const unsigned long long old = atomicCAS(jt_key(p, slot), (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (old == kEmptyKey) { *jt_row(p, slot) = (unsigned int)(row0 + r); break; }
        if (old == key) { p.flags[0] = 1u; break; }
        slot = jt_next(p, slot);
        if (probe > (1u << 16)) { p.flags[1] = 1u; break; }
      }
    }
  }
};

struct ProbeAggSink {
  using Params = JoinAggTable;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    const uint64_t cap = 1ull << p.log2_cap;
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      int64_t slot = -1;
      if (pass[r] && ((rf.getv(sh.key) >> r) & 1)) {
        const uint64_t key = rf.get(r, sh.key);
        if (key == kEmptyKey) { if (*jt_row(p, cap) != kNoRow32) slot = (int64_t)cap; }
        else {
          uint64_t s = (key * kP2HashMult) >> (64 - p.log2_cap);
          for (;;) {
            const unsigned long long cur = *jt_key(p, s);
            if (cur == key) { slot = (int64_t)s; break; }
            if (cur == kEmptyKey) break;
            s = jt_next(p, s);
          }
        }
        // One contribution per probe row, into the cells of its KEY -- also when build keys repeat (multi-value mode).  The aggregates read the probe side only, so every
        // build row of the key would receive the very same contributions: the groups (build rows, or the rows of a key that agree on the build-side group columns) are
        // expanded from the key's cells when the table is compacted (k::chains_agg_compact: a group of m build rows = m copies of the key's aggregate).  Round 5 walked the
        // key's chain here and added into the cells of every row's representative: 8e7 x 2 device atomics for 2e7 candidates (3.6 ms of the duplicate-key join's 10.9).
        if (slot >= 0) atomic_row(sh, rf, r, row0 + r, p.acc + (size_t)jt_cell(p, (uint64_t)slot) * sh.n_aggs);
      }
    }
  }
};


// Ordinals are handed out in per-wave chunks: a wave reserves kOrdChunk ordinals with ONE device atomic
// and sub-allocates from them (one atomic per wave-row on a single counter word took 26 ms for the 1.5e8-row
// TPC-H orders scan; ~12 k atomics this way).  Unused tails of chunks stay empty (LEN cell 0).
// slot of key index idx (its bit is set in `word` = bits[idx >> 6]): build rows numbered in key order
__device__ __forceinline__ unsigned long long direct_slot(const DirectJoinTable& t, unsigned long long idx, unsigned long long word) {
  return (unsigned long long)t.rank[idx >> 6] + (unsigned long long)__popcll(word & ((1ull << (idx & 63)) - 1ull));
}
struct DirectBuildSink {
  using Params = DirectJoinTable;
  unsigned int next = 0, end = 0;   // wave-uniform
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) { next = 0; end = 0; }
  __device__ __forceinline__ void close_chunk(const Params& p) {
    if (end != 0 && lane_id() == 0) p.chunk_used[(end - kOrdChunk) / kOrdChunk] = next - (end - kOrdChunk);
  }
  template <class S> __device__ __forceinline__ void finish(const S&, const Params& p) { close_chunk(p); }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    bool part[kRows];
    unsigned int word[kRows];
    unsigned long long bit[kRows];
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      part[r] = false; word[r] = 0; bit[r] = 0;
      const bool ins = pass[r] && ((rf.getv(sh.key) >> r) & 1);
      const uint64_t m = ballot(ins);
      if (m == 0) continue;
      const unsigned int need = (unsigned int)popc64(m);
      if (next + need > end) {   // wave-uniform: reserve a fresh chunk
        close_chunk(p);
        unsigned int base = 0;
        if (lane_id() == 0) base = atomicAdd(p.counter, kOrdChunk);
        next = __shfl(base, 0, 64);
        end = next + kOrdChunk;
      }
      const unsigned int ord = next + (unsigned int)prefix_rank(m);
      next += need;
      if (!ins) continue;
      if (ord >= p.n_ord) { p.flags[1] = 1u; continue; }
      const uint64_t key = rf.get(r, sh.key);
      const uint64_t idx = key - (uint64_t)p.kmin;       // < range by construction (kmin/kmax cover the whole build column)
      p.ord_key[ord] = key;
      p.ord_row[ord] = (unsigned int)(row0 + r);
      part[r] = true; word[r] = (unsigned int)(idx >> 6); bit[r] = 1ull << (idx & 63);     // range <= 2^34: the word index fits 28 bits
    }
    if (!p.bits) return;                                   // (wave-uniform) pairs only: k::partitioned_join_build
    // fire-and-forget (no-return) atomics: a duplicate build key shows up as popcount(bits) < number of pairs, which the rank step
    // counts (the caller then falls back).  The two rows of a lane usually share a bitmap word when the build table is scanned in
    // key order: one atomic then.  (Merging across lanes with a segmented wave scan was measured: no faster, the scan is not
    // bound by the atomic rate.)
    static_assert(kRows == 2, "pairwise merge below");
    if (part[0] && part[1] && word[0] == word[1]) { bit[0] |= bit[1]; part[1] = false; }
#pragma unroll
    for (int r = 0; r < kRows; r++)
      if (part[r]) __hip_atomic_fetch_or(&p.bits[word[r]], bit[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

struct DirectProbeAggSink {
  using Params = DirectJoinTable;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r] || !((rf.getv(sh.key) >> r) & 1)) continue;
      const uint64_t idx = rf.get(r, sh.key) - (uint64_t)p.kmin;
      if (idx >= p.range) continue;
      const unsigned long long w = p.bits[idx >> 6];
      if (!((w >> (idx & 63)) & 1ull)) continue;
      atomic_row(sh, rf, r, row0 + r, p.acc + (size_t)direct_slot(p, idx, w) * sh.n_aggs);
    }
  }
};

// ---- sink: membership bitmap of a semi-join's filter side -------------------------------------------
struct BitmapBuildSink {
  using Params = BitmapBuild;
  unsigned long long n = 0;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) { n = 0; }
  template <class S> __device__ __forceinline__ void finish(const S&, const Params& p) {
    const uint64_t w = wave_sum_u64(n);
    if (lane_id() == 0 && w) atomicAdd(p.count, (unsigned long long)w);
  }
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t, const Params& p) {
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if (!pass[r] || !((rf.getv(sh.key) >> r) & 1)) continue;   // null keys never match
      const uint64_t idx = rf.get(r, sh.key) - (uint64_t)p.kmin;
      if (idx >= p.range) continue;
      n++;
      __hip_atomic_fetch_or(&p.bits[idx >> 6], 1ull << (idx & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
};

// ---- sink: the rows of a probe side that hit a direct-address join table, in ballot form (DirectHits) -----------------------------------------
struct DirectHitsSink {
  using Params = DirectHits;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  template <class S, class RF> __device__ __forceinline__ void consume(const S& sh, const RF& rf, const bool pass[kRows], int64_t row0, const Params& p) {
    bool hit[kRows];
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      bool ok = pass[r] && ((rf.getv(sh.key) >> r) & 1);            // null keys never match
      const uint64_t idx = rf.get(r, sh.key) - (uint64_t)p.t.kmin;
      ok = ok && idx < p.t.range;
      if (ok) ok = (p.t.bits[idx >> 6] >> (idx & 63)) & 1ull;
      hit[r] = ok;
    }
    const unsigned long long b0 = ballot(hit[0]), b1 = ballot(hit[1]);
    if (lane_id() == 0) {
      const int64_t t = row0 / kTileRows;
      *reinterpret_cast<ulonglong2*>(p.out.ballots + t * 2) = make_ulonglong2(b0, b1);
      p.out.counts[t] = (unsigned int)(popc64(b0) + popc64(b1));
    }
  }
};

// AOT / JIT probe scan with late materialisation: predicate + key for the tile, bitmap test, then the rest of the program for
// the lanes that hit (DirectJoinTable::opts & kDirectLateLoads; without it runs the whole program up front like every other sink)
template <class P, bool FULL>
__device__ __forceinline__ void direct_probe_tile(const Args& args, const DirectJoinTable& p, int64_t base, RegFile& rf) {
  constexpr Shape sh = P::shape();
  const int64_t row0 = base + (int64_t)lane_id() * kRows;
  run_split<P, FULL, true>(args, row0, rf);
  if (!(p.opts & kDirectLateLoads)) run_split<P, FULL, false>(args, row0, rf);
  bool hit[kRows], any = false;
  unsigned long long slot[kRows];
#pragma unroll
  for (int r = 0; r < kRows; r++) {
    bool ok = FULL || (row0 + r < args.n_rows);
    if (sh.pred != kNone) ok = ok && (rf.get(r, sh.pred) & 1) && ((rf.getv(sh.pred) >> r) & 1);
    ok = ok && ((rf.getv(sh.key) >> r) & 1);
    const uint64_t idx = rf.get(r, sh.key) - (uint64_t)p.kmin;
    ok = ok && idx < p.range;
    slot[r] = 0;
    if (ok) {
      const unsigned long long w = p.bits[idx >> 6];
      ok = (w >> (idx & 63)) & 1ull;
      if (ok) slot[r] = direct_slot(p, idx, w);
    }
    hit[r] = ok; any = any || ok;
  }
  if (any) {
    if (p.opts & kDirectLateLoads) run_split<P, FULL, false>(args, row0, rf);
#pragma unroll
    for (int r = 0; r < kRows; r++)
      if (hit[r]) atomic_row(sh, rf, r, row0 + r, p.acc + (size_t)slot[r] * sh.n_aggs);
  }
}

template <class P, class Sink>
__host__ __device__ constexpr bool late_probe() {
  if constexpr (P::kStatic && std::is_same<Sink, DirectProbeAggSink>::value) return split_program(P::shape()).any_late;
  else return false;
}
template <class P, class Sink>
__device__ __forceinline__ void fused_scan_body(const Shape dsh, const Args args, const typename Sink::Params sp) {   // by value: kernel arguments passed by
                                                                                                              // reference become addressable stack copies (spills)
  Sink sink;
  typename RegFileOf<P>::type rf = make_regfile<P>(args);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ntiles = (args.n_rows + kTileRows - 1) / kTileRows;
  if constexpr (late_probe<P, Sink>()) {
    for (int64_t t = wave; t < ntiles; t += nwaves) {
      const int64_t base = t * kTileRows;
      if (base + kTileRows <= args.n_rows) direct_probe_tile<P, true>(args, sp, base, rf);   // wave-uniform
      else direct_probe_tile<P, false>(args, sp, base, rf);
    }
  } else if constexpr (P::kStatic) {
    constexpr Shape sh = P::shape();
    sink.init(sh, sp);
    // (two tiles of a wave evaluated together, so that the second tile's bitmap lookups fly while the first waits, were measured on TPC-H Q3's orders scan with the
    // customer filter: no change -- that scan was bound by the LINES its lookups pull from the L2, which OP_MASKV halves)
    for (int64_t t = wave; t < ntiles; t += nwaves) {
      bool pass[kRows]; int64_t row0;
      tile_rows<P>(dsh, args, t, rf, pass, row0);
      sink.consume(sh, rf, pass, row0, sp);
    }
    sink.finish(sh, sp);
  } else {
    sink.init(dsh, sp);
    for (int64_t t = wave; t < ntiles; t += nwaves) {
      bool pass[kRows]; int64_t row0;
      tile_rows<P>(dsh, args, t, rf, pass, row0);
      sink.consume(dsh, rf, pass, row0, sp);
    }
    sink.finish(dsh, sp);
  }
}

template <class P, class Sink>
__global__ __launch_bounds__(kBlock) void fused_scan_kernel(Shape dsh, Args args, typename Sink::Params sp) {
  fused_scan_body<P, Sink>(dsh, args, sp);
}

// ---- sink: the selection of a filter -> frame, in ballot form (BallotOut, fused.hpp) ---------------------------------------------------
// First half of FilterExec without its intermediates (polars-mem-engine/src/executors/filter.rs:94-145: predicate column -> mask -> per-column filter): the
// predicate program runs inside the scan and what leaves it per 128-row wave tile is 16 bytes of ballots (lane l holds rows 2l, 2l + 1: one ballot per row
// parity) and the tile's kept-row count.  A device scan over the counts gives every wave tile its output offset; k::compact_by_ballots (kernels_filter.hip) then
// moves ALL payload columns in one pass -- loads coalesced and batched per column, ranks from v_mbcnt over the ballots, no atomics, no workgroup barriers.
// (Round 6 first built this as ONE pass with a chained scan across 2048-row tiles -- tickets + decoupled look-back, agent-scope atomics.  Measured on the 1e9-row
// frame: 6.5 ms with the look-back ablated, 19-22 ms with it -- half a million tiles polling a handful of hot lines across eight XCDs whose L2s are not coherent
// with each other -- and a floor of 5.7 ms from the ticket counter alone.  Two passes move 8 GB more and need neither.)
struct BallotSink {
  using Params = BallotOut;
  template <class S> __device__ __forceinline__ void init(const S&, const Params&) {}
  template <class S> __device__ __forceinline__ void finish(const S&, const Params&) {}
  template <class S, class RF> __device__ __forceinline__ void consume(const S&, const RF&, const bool pass[kRows], int64_t row0, const Params& p) {
    const unsigned long long b0 = ballot(pass[0]), b1 = ballot(pass[1]);
    if (lane_id() == 0) {
      const int64_t t = row0 / kTileRows;                          // lane 0's first row is the wave tile's first row
      *reinterpret_cast<ulonglong2*>(p.ballots + t * 2) = make_ulonglong2(b0, b1);
      p.counts[t] = (unsigned int)(popc64(b0) + popc64(b1));
    }
  }
};

}  // namespace k
}  // namespace plx
