// parquet_emu.cpp -- CPU harness of the device Parquet decoder (TEST INFRASTRUCTURE: never linked into libpolars_amd.so).
//
// It instantiates the product's own orchestration (polars_amd/csrc/parquet_reader.hpp: read_column<B>) with a backend whose "device
// memory" is host memory and whose "kernel launches" run the product's own per-thread / per-wavefront bodies
// (polars_amd/csrc/parquet_device.hpp) one thread after another -- the grid mapping of kernels_parquet.hip restated as loops.  What the
// CPU tests therefore cover: footer / page-header parsing, page planning, dictionary unification, Snappy (stage / parse / copy rounds
// of one wavefront), run tables, validity words, dense-slot ranks, value decode.  What they cannot cover: the launch shells and the
// memory-ordering of the real wavefront (the GPU tests do).
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../polars_amd/csrc/host_codecs.hpp"
#include "../../polars_amd/csrc/parquet_reader.hpp"

using namespace plx::pq;

namespace {

// the part of a round after the parse, lanes of a phase one after another (ascending or descending: the result must not depend on it)
template <class F> void for_lanes(int order, F&& f) {
  if (order == 0) for (uint32_t lane = 0; lane < kSnapLanes; lane++) f(lane);
  else for (uint32_t lane = kSnapLanes; lane-- > 0;) f(lane);
}
// one wavefront decompressing one stream: the loop of pq_snappy_kernel, every barrier-separated phase run lane after lane
uint32_t snappy_stream(const DecompJob& job, int order, uint32_t* rounds) {
  SnapShared sh;
  memset(&sh, 0xA5, sizeof sh);     // LDS is not zeroed
  snappy_begin(sh, job);
  uint32_t nr = 0;
  const char* gen = getenv("PLX_SNAPPY_KERNEL");      // default (as in the product): the bodies of pq_snappy_kernel_v2; "1": the first generation
  const bool v2 = !(gen && gen[0] == '1');
  while (sh.done == 0) {
    nr++;
    for_lanes(order, [&](uint32_t lane) { snappy_stage(sh, job, lane); });
    for_lanes(order, [&](uint32_t lane) { if (v2) snappy_next_v2(sh, job, lane); else snappy_next(sh, job, lane); });
    for (uint32_t it = 0; it < kSnapSweeps; it++) {   // __syncthreads_or(snappy_mark(...)) of the kernel
      bool any = false;
      for_lanes(order, [&](uint32_t lane) { any |= v2 ? snappy_mark_v2(sh, it, lane) : snappy_mark(sh, it, lane); });
      if (!any) break;
    }
    for_lanes(order, [&](uint32_t lane) { if (v2) snappy_rank_v2(sh, lane); else snappy_rank(sh, lane); });
    if (v2) {
      for_lanes(order, [&](uint32_t lane) { snappy_scan_v2_blocks(sh, lane); });
      snappy_scan_v2_totals(sh);
      for_lanes(order, [&](uint32_t lane) { snappy_scan_v2_offsets(sh, lane); });
    } else snappy_scan(sh);
    for_lanes(order, [&](uint32_t lane) { if (v2) snappy_place_v2(sh, job, lane); else snappy_place(sh, job, lane); });
    snappy_finish(sh, job);
    if (sh.done == 2 || sh.bad) break;
    if (sh.direct) { for_lanes(order, [&](uint32_t lane) { snappy_direct(sh, job, lane); }); continue; }
    for_lanes(order, [&](uint32_t lane) { snappy_point(sh, lane); });
    for (;;) {                                   // __syncthreads_or(snappy_jump(...)) of the kernel
      bool any = false;
      for_lanes(order, [&](uint32_t lane) { any |= v2 ? snappy_jump_v2(sh, lane) : snappy_jump(sh, lane); });
      if (!any) break;
    }
    for_lanes(order, [&](uint32_t lane) { snappy_gather(sh, job, lane); });
  }
  if (rounds) *rounds = nr;
  return (sh.done == 2 || sh.bad) ? (uint32_t)PE_SNAPPY : 0u;
}

// the wavefront of the zstd bodies (parquet_zstd.hpp): the lanes of a phase one after another
struct EmuWave {
  int order = 0;
  template <class F> void lanes(F&& f) {
    if (order == 0) for (uint32_t lane = 0; lane < kZLanes; lane++) f(lane);
    else for (uint32_t lane = kZLanes; lane-- > 0;) f(lane);
  }
  void sync() {}
  void wave_fence() {}
  uint32_t uniform(uint32_t x) { return x; }
  void tick(int) {}
  void count(int, uint32_t) {}
  void exclusive_scan(uint32_t* a) { uint32_t run = 0; for (uint32_t l = 0; l < kZLanes; l++) { const uint32_t v = a[l]; a[l] = run; run += v; } }
  void rep_scan(uint32_t* r0, uint32_t* r1, uint32_t* r2) {
    ZstdRepMap run = zstd_rep_identity();
    for (uint32_t l = 0; l < kZLanes; l++) {
      ZstdRepMap m; m.s[0] = r0[l]; m.s[1] = r1[l]; m.s[2] = r2[l];
      run = zstd_rep_compose(run, m);
      r0[l] = run.s[0]; r1[l] = run.s[1]; r2[l] = run.s[2];
    }
  }
  void batch_scan(uint32_t* lit, uint32_t* out, uint32_t* r0, uint32_t* r1, uint32_t* r2) { exclusive_scan(lit); exclusive_scan(out); rep_scan(r0, r1, r2); }
  template <class F> uint64_t ballot(F&& pred) { uint64_t m = 0; for (uint32_t l = 0; l < kZLanes; l++) if (pred(l)) m |= (uint64_t)1 << l; return m; }
  uint32_t first_flag(const uint32_t* flag) { for (uint32_t l = 0; l < kZLanes; l++) if (flag[l]) return l; return kZLanes; }
};
// pq_zstd_entropy + pq_zstd_execute (kernels_parquet.hip) as loops: one wavefront per compressed block, then one per page; returns PE_ZSTD or 0
uint32_t zstd_run(ZstdBlock* blocks, const uint32_t* order_idx, uint32_t n_compressed, uint32_t n_huf_only, const ZstdHufDesc* hufs, const ZstdFseDesc* fses, const ZstdStream* streams, uint32_t n_streams,
                  int order) {
  EmuWave w;
  w.order = order;
  uint32_t err = 0;
  auto each = [&](size_t n, auto&& f) {
    if (order == 0) for (size_t i = 0; i < n; i++) f(i);
    else for (size_t i = n; i-- > 0;) f(i);
  };
  // the grid of pq_zstd_entropy_kernel: wavefronts of sixteen sequence-free blocks first, then wavefronts of four blocks
  const uint32_t waves_h = (n_huf_only + kZHufGroups - 1) / kZHufGroups, n_other = n_compressed - n_huf_only, waves_o = (n_other + kZGroups - 1) / kZGroups;
  each(waves_h + waves_o, [&](size_t i) {
    if (i < waves_h) {
      auto sh = std::make_unique<ZstdHufShared[]>(kZHufGroups);
      memset(sh.get(), 0xA5, sizeof(ZstdHufShared) * kZHufGroups);     // LDS is not zeroed
      zstd_huf_group(w, sh.get(), blocks, order_idx, (uint32_t)(i * kZHufGroups), n_huf_only, hufs);
    } else {
      auto sh = std::make_unique<ZstdEntropyShared[]>(kZGroups);
      memset(sh.get(), 0xA5, sizeof(ZstdEntropyShared) * kZGroups);
      zstd_entropy_group(w, sh.get(), blocks, order_idx + n_huf_only, (uint32_t)((i - waves_h) * kZGroups), n_other, hufs, fses);
    }
  });
  each(n_streams, [&](size_t i) {
    auto sh = std::make_unique<ZstdExecShared>();
    memset(sh.get(), 0xA5, sizeof *sh);
    if (!zstd_exec_stream(w, *sh, streams[i], blocks)) err |= (uint32_t)PE_ZSTD;
  });
  return err;
}

struct HostBackend {
  using Mem = std::shared_ptr<std::vector<uint8_t>>;
  std::vector<uint8_t> stage_[2];
  int next_ = 0;
  int order = 0;   // 0: threads ascending, 1: descending (results must not depend on the order threads run in)

  Mem alloc(size_t bytes) { return std::make_shared<std::vector<uint8_t>>(bytes + 64, (uint8_t)0xA5); }   // device memory is not zeroed
  uint64_t addr(const Mem& m) { return m ? (uint64_t)m->data() : 0; }
  uint8_t* host_stage(size_t bytes) {
    std::vector<uint8_t>& s = stage_[next_];
    next_ ^= 1;
    s.assign(bytes, 0xEE);
    return s.data();
  }
  void upload(uint64_t dst, const void* src, size_t bytes) { memcpy((void*)dst, src, bytes); }
  void upload_small(uint64_t dst, const void* src, size_t bytes) { memcpy((void*)dst, src, bytes); }
  void discard_pending() {}
  // stand-in for the device dictionary encoder (plx_strview_dict_encode): codes in first-appearance order, categories into the file
  void encode_string_views(File& f, int leaf, const uint8_t* views, const uint8_t* validity, int64_t n, const std::vector<const void*>& ptrs,
                           const std::vector<int64_t>& sizes, ColumnResult<HostBackend>* res) {
    std::unordered_map<std::string, uint32_t> index;
    std::vector<std::string> cats;
    res->values = alloc((size_t)n * 4 + 8);
    uint32_t* codes = (uint32_t*)res->values->data();
    for (int64_t i = 0; i < n; i++) {
      const bool ok = !validity || ((validity[(size_t)i >> 3] >> (i & 7)) & 1);
      if (!ok) { codes[i] = 0; continue; }
      const uint8_t* v = views + 16 * (size_t)i;
      uint32_t len, bi, off;
      memcpy(&len, v, 4);
      std::string s;
      if (len <= 12) s.assign((const char*)v + 4, len);
      else {
        memcpy(&bi, v + 8, 4); memcpy(&off, v + 12, 4);
        if (bi >= ptrs.size() || (int64_t)off + len > sizes[bi]) throw FormatError("view points outside its data buffer");
        if (memcmp((const uint8_t*)ptrs[bi] + off, v + 4, 4) != 0) throw FormatError("view prefix differs from its data");
        s.assign((const char*)ptrs[bi] + off, len);
      }
      auto it = index.find(s);
      if (it == index.end()) { it = index.emplace(s, (uint32_t)cats.size()).first; cats.push_back(s); }
      codes[i] = it->second;
    }
    if (validity) {
      res->validity = alloc((size_t)((n + 63) / 64) * 8 + 8);
      memset(res->validity->data(), 0, res->validity->size());
      memcpy(res->validity->data(), validity, (size_t)(n + 7) / 8);
      res->has_validity = true;
    }
    f.categories[leaf] = std::move(cats);
  }
  void zero(uint64_t dst, size_t bytes) { memset((void*)dst, 0, bytes); }
  uint64_t read_u64(uint64_t a) { uint64_t v; memcpy(&v, (const void*)a, 8); return v; }
  uint32_t read_u32(uint64_t a) { uint32_t v; memcpy(&v, (const void*)a, 4); return v; }
  void scan_u32(const uint32_t* in, uint64_t* out, int64_t n) {
    uint64_t run = 0;
    for (int64_t i = 0; i < n; i++) { out[i] = run; run += in[i]; }
    out[n] = run;
  }
  template <class F> void for_threads(uint64_t n, F&& f) {
    if (order == 0) for (uint64_t t = 0; t < n; t++) f(t);
    else for (uint64_t t = n; t-- > 0;) f(t);
  }

  // one wavefront per stream: the loop of pq_snappy_kernel with the lanes of a phase run one after another
  void run_snappy(const DecompJob* jobs, uint32_t n, uint64_t, uint32_t* err) {
    for_threads(n, [&](uint64_t j) { *err |= snappy_stream(jobs[j], order, nullptr); });
  }
  void run_zstd(ZstdBlock* blocks, const uint32_t* order_idx, uint32_t n_compressed, uint32_t n_huf_only, const ZstdHufDesc* hufs, const ZstdFseDesc* fses, const ZstdStream* streams, uint32_t n_streams,
                uint64_t, uint64_t, uint32_t* err) {
    *err |= zstd_run(blocks, order_idx, n_compressed, n_huf_only, hufs, fses, streams, n_streams, order);
  }
  void run_page_prepare(PageDesc* pages, uint32_t n, uint32_t* err) { for_threads(n, [&](uint64_t i) { *err |= page_prepare(pages[i]); }); }
  void run_count_runs(const PageDesc* pages, uint32_t n, bool levels, uint32_t* counts, uint32_t* err) {
    for_threads(2ull * n, [&](uint64_t t) {
      int s = (int)(t & 1);
      counts[t] = (s == 0 && !levels) ? 0u : stream_entries(pages[t >> 1], s, err);
    });
  }
  void run_fill_runs(const PageDesc* pages, uint32_t n, const uint64_t* offs, RunEntry* runs) {
    for_threads(2ull * n, [&](uint64_t t) {
      uint32_t cnt = (uint32_t)(offs[t + 1] - offs[t]);
      if (cnt) stream_fill(pages[t >> 1], (int)(t & 1), runs + offs[t], cnt);
    });
  }
  void run_validity(const PageDesc* pages, uint32_t n, const RunEntry* runs, const uint64_t* offs, uint64_t n_rows, uint64_t* validity, uint32_t* popc, uint32_t* err) {
    for_threads((n_rows + 63) >> 6, [&](uint64_t w) {
      uint64_t word = validity_word(pages, n, runs, offs, n_rows, w, err);
      validity[w] = word;
      popc[w] = (uint32_t)__builtin_popcountll(word);
    });
  }
  void run_page_valid0(PageDesc* pages, uint32_t n, const uint64_t* validity, const uint64_t* prefix) {
    for_threads(n, [&](uint64_t i) { pages[i].valid0 = valid_before(validity, prefix, pages[i].row0); });
  }
  void run_decode(const ColumnDecode& c, void* out, uint32_t out_width, uint32_t* err) {
    if (out_width == 0) for_threads((c.n_rows + 63) >> 6, [&](uint64_t w) { ((uint64_t*)out)[w] = decode_bool_word(c, w, err); });
    else for_threads(c.n_rows, [&](uint64_t r) { decode_rows(c, out, out_width, r, r + 1, err); });
  }
};

struct EmuResult {
  ColumnResult<HostBackend> col;
  std::vector<std::string> categories;
  ReadStats stats;
};

thread_local std::string t_err;

}  // namespace

extern "C" {

const char* pqemu_last_error() { return t_err.c_str(); }

// returns 0 ok, 3 unsupported, 1 invalid
int pqemu_read_column(const char* path, const int* row_groups, int n_row_groups, int column, int thread_order, void** out) {
  try {
    std::unique_ptr<File> f = open_file(path);
    HostBackend be;
    be.order = thread_order;
    auto r = std::make_unique<EmuResult>();
    std::vector<int> rgs(row_groups, row_groups + n_row_groups);
    r->col = read_column(be, *f, rgs, column, &r->stats);
    auto it = f->categories.find(column);
    if (it != f->categories.end()) r->categories = it->second;
    *out = r.release();
    return 0;
  } catch (const Unsupported& e) { t_err = e.what(); return 3;
  } catch (const std::exception& e) { t_err = e.what(); return 1; }
}
void pqemu_free(void* h) { delete (EmuResult*)h; }
void pqemu_info(void* h, int* dtype, int* logical, int64_t* len, int64_t* null_count, int* has_validity, int64_t* n_categories, uint64_t* stats6) {
  EmuResult* r = (EmuResult*)h;
  *dtype = r->col.dtype; *logical = r->col.logical; *len = r->col.len; *null_count = r->col.null_count; *has_validity = r->col.has_validity ? 1 : 0;
  *n_categories = (int64_t)r->categories.size();
  stats6[0] = r->stats.file_bytes; stats6[1] = r->stats.data_pages; stats6[2] = r->stats.dict_pages; stats6[3] = r->stats.snappy_streams;
  stats6[4] = r->stats.snappy_bytes_out; stats6[5] = r->stats.run_entries; stats6[6] = r->stats.host_inflated_pages; stats6[7] = r->stats.host_inflated_bytes;
  stats6[8] = r->stats.zstd_streams; stats6[9] = r->stats.zstd_blocks;      // (the caller passes 10 words)
}
void pqemu_copy(void* h, void* values, size_t values_bytes, void* validity, size_t validity_bytes) {
  EmuResult* r = (EmuResult*)h;
  if (values_bytes) memcpy(values, r->col.values->data(), values_bytes);
  if (validity_bytes && r->col.has_validity) memcpy(validity, r->col.validity->data(), validity_bytes);
}
int64_t pqemu_category(void* h, int64_t i, char* buf, int64_t cap) {
  EmuResult* r = (EmuResult*)h;
  const std::string& s = r->categories[(size_t)i];
  if ((int64_t)s.size() <= cap) memcpy(buf, s.data(), s.size());
  return (int64_t)s.size();
}

// Snappy alone: the wavefront rounds against an arbitrary compressed buffer (fuzzed by the tests against a reference decoder)
int pqemu_snappy(const uint8_t* in, uint32_t n_in, uint8_t* out, uint32_t n_out, int thread_order, uint32_t* rounds) {
  HostBackend be;
  be.order = thread_order;
  std::vector<uint8_t> src(in, in + n_in);
  src.resize(n_in + 64, 0xCC);
  DecompJob job{(uint64_t)src.data(), (uint64_t)out, n_in, n_out};
  uint32_t nr = 0;
  uint32_t err = snappy_stream(job, thread_order, &nr);
  if (rounds) *rounds = nr;
  return (int)err;
}

// device zstd alone: index pass (host) + entropy / execute bodies over an arbitrary compressed buffer.  Returns 0, PE_ZSTD (the kernels flagged the stream) or -1 (the
// index pass rejected a header: t_err says why)
int pqemu_zstd(const uint8_t* in, uint32_t n_in, uint8_t* out, uint32_t n_out, int thread_order, uint32_t* counts) {
  try {
    std::vector<uint8_t> src(in, in + n_in);           // exact-size heap blocks: a sanitizer build sees any access past the stream or past the output
    std::vector<uint8_t> dst((size_t)n_out);
    ZstdPlan plan;
    zstd_index_stream(plan, src.data(), n_in, (uint64_t)src.data(), n_out);
    std::vector<uint8_t> lits(plan.lit_bytes, 0xA5), seqs(plan.n_seq * 16, 0xA5);
    plan.streams[0].dst = (uint64_t)dst.data();
    zstd_plan_place(plan, (uint64_t)lits.data(), (uint64_t)seqs.data());
    if (counts) { counts[0] = (uint32_t)plan.blocks.size(); counts[1] = (uint32_t)plan.n_compressed; counts[2] = (uint32_t)plan.n_seq; counts[3] = (uint32_t)plan.hufs.size(); counts[4] = (uint32_t)plan.fses.size(); }
    uint32_t n_huf_only = 0;
    const std::vector<uint32_t> order_idx = zstd_plan_order(plan, &n_huf_only);
    const uint32_t err = zstd_run(plan.blocks.data(), order_idx.data(), (uint32_t)order_idx.size(), n_huf_only, plan.hufs.data(), plan.fses.data(), plan.streams.data(), 1, thread_order);
    if (n_out) memcpy(out, dst.data(), n_out);
    return (int)err;
  } catch (const std::exception& e) { t_err = e.what(); return -1; }
}
// the closed forms of the sequence code tables against the RFC's tables (host_codecs.hpp); returns the number of disagreements
int pqemu_zstd_code_selfcheck() {
  using namespace plx::codec::zstd_detail;
  int bad = 0;
  for (uint32_t c = 0; c < 36; c++) bad += (z_ll_base(c) != kLLBase[c]) + (z_ll_bits(c) != kLLBits[c]);
  for (uint32_t c = 0; c < 53; c++) bad += (z_ml_base(c) != kMLBase[c]) + (z_ml_bits(c) != kMLBits[c]);
  return bad;
}

// the host decompressors of the product (polars_amd/csrc/host_codecs.hpp): codec 0 = zstd, 1 = lz4 raw block, 2 = lz4 frame, 3 = gzip / zlib / raw deflate
int pqemu_host_codec(int codec, const uint8_t* in, uint32_t n_in, uint8_t* out, uint32_t n_out) {
  try {
    std::vector<uint8_t> src(in, in + n_in);          // exact-size heap blocks: a sanitizer build sees any access past either end
    std::vector<uint8_t> dst(n_out);
    if (codec == 0) plx::codec::zstd_decompress(src.data(), src.size(), dst.data(), dst.size());
    else if (codec == 1) plx::codec::lz4_raw_decompress(src.data(), src.size(), dst.data(), dst.size());
    else if (codec == 2) plx::codec::lz4_frame_decompress(src.data(), src.size(), dst.data(), dst.size());
    else plx::codec::gzip_decompress(src.data(), src.size(), dst.data(), dst.size());
    if (n_out) memcpy(out, dst.data(), n_out);
    return 0;
  } catch (const std::exception& e) { t_err = e.what(); return 1; }
}

// the host Snappy of the product (string dictionary pages)
int pqemu_snappy_host(const uint8_t* in, uint32_t n_in, uint8_t* out, uint32_t n_out) {
  try {
    std::vector<uint8_t> v = snappy_decompress_host(in, n_in, n_out);
    memcpy(out, v.data(), v.size());
    return 0;
  } catch (const std::exception& e) { t_err = e.what(); return 1; }
}

// the branch-free tag parse of the second-generation bodies against the branching one (parquet_snappy.hpp): every tag byte x `n_rest` pseudo-random
// and edge-valued tails, at every byte alignment of the window; returns the number of disagreements (0 expected)
int64_t pqemu_snappy_tag_selfcheck(uint32_t n_rest, uint64_t seed) {
  alignas(16) uint8_t win[64];
  int64_t bad = 0;
  const uint32_t edges[] = {0u, 1u, 0xffu, 0x100u, 0xffffu, 0x10000u, 0xffffffu, 0x1000000u, 0x3ffffffeu, 0x3fffffffu, 0x40000000u, 0x7fffffffu, 0x80000000u, 0xffffffffu};
  const uint32_t n_edges = (uint32_t)(sizeof edges / sizeof edges[0]);
  uint64_t x = seed | 1;
  for (uint32_t tag = 0; tag < 256; tag++)
    for (uint32_t r = 0; r < n_rest + n_edges; r++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const uint32_t rest = r < n_edges ? edges[r] : (uint32_t)(x >> 17);
      for (uint32_t b = 8; b < 12; b++) {           // the four alignments of a window position
        for (uint32_t i = 0; i < 64; i++) win[i] = (uint8_t)(x >> (i & 31));
        win[b] = (uint8_t)tag;
        memcpy(win + b + 1, &rest, 4);
        uint32_t l0, v0, h0, l1, v1, h1;
        const uint32_t k0 = snappy_tag(win + b, &l0, &v0, &h0);
        const uint32_t k1 = snappy_tag_x(snappy_peek(win, b), &l1, &v1, &h1);
        if (k0 != k1 || l0 != l1 || v0 != v1 || h0 != h1) bad++;
      }
    }
  return bad;
}

}  // extern "C"
