// jit.cpp -- hiprtc instantiation of fused_scan_body<JitProg, Sink> for run-time program shapes (see jit.hpp).
#include "jit.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "core.hpp"
#include "kconfig.hpp"

namespace plx {
namespace jit {

using namespace fused;

namespace {
struct Entry { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; bool failed = false; };
std::mutex g_mu;
std::map<std::string, Entry> g_cache;
int g_compiled = 0;
double g_ms = 0;

const char* sink_type(Sink s) {
  static const char* n[] = {"RegAggSink", "LdsAggSink", "DenseAggSink", "HashAggSink", "WideAggSink", "JoinBuildSink", "ProbeAggSink", "DirectBuildSink", "DirectProbeAggSink", "BitmapBuildSink",
                            "part_count", "part_scatter", "part_agg", "part2_scatter_hash", "part2_scatter_direct", "part2_agg_hash", "part2_agg_direct", "part2_scatter_hash_t2", "part2_scatter_direct_t2"};
  if (s == PART3_AGG_PAIRV) return "part3_agg";
  if (s >= PART3_SCATTER_PAIRV && s < PART3_AGG_PAIRV) return "part3_scatter";
  if (s == PART3_AGG_PAIR || s == PART3_AGG_PAIR_DIRECT) return "part3_agg";
  if (s >= PART3_SCATTER_PAIR && s < PART3_AGG_PAIR) return "part3_scatter";
  if (s == BALLOT) return "BallotSink";
  if (s == DIRECT_HITS) return "DirectHitsSink";
  if (s >= PART3_AGG) return "part3_agg";
  if (s >= PART3_SCATTER) return "part3_scatter";
  return n[s];
}

std::string include_dir() {
  if (const char* e = getenv("PLX_JIT_INCLUDE")) return e;
  Dl_info info;
  if (dladdr((const void*)&stats, &info) && info.dli_fname) {
    std::string p = info.dli_fname;          // .../polars_amd/libpolars_amd.so
    size_t k = p.rfind('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
  }
  return "polars_amd/csrc";
}

// clang's own <stddef.h>/<stdint.h>: hiprtc has no system include paths
std::string resource_include() {
  if (const char* e = getenv("PLX_JIT_CLANG_INCLUDE")) return e;
  const char* roots[] = {"/opt/rocm/lib/llvm/lib/clang", "/opt/rocm/llvm/lib/clang"};
  for (const char* r : roots) {
    for (int v = 30; v >= 14; v--) {
      const std::string p = std::string(r) + "/" + std::to_string(v) + "/include";
      if (FILE* f = fopen((p + "/stddef.h").c_str(), "r")) { fclose(f); return p; }
    }
  }
  return "/opt/rocm/lib/llvm/lib/clang/22/include";
}
std::vector<std::string> compile_options() {
  std::vector<std::string> o = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-pass-failed", "-ffreestanding",
                                "-I" + include_dir(), "-I/opt/rocm/include", "-I" + resource_include()};
  // measurement runs: extra -D options for the run-time compiled kernels (PLX_JIT_DEFINES="-DPLX_FILTER_GROUP=2 -DPLX_FILTER_EARLY=0"); part of the cache key
  if (const char* e = getenv("PLX_JIT_DEFINES")) { std::istringstream is(e); std::string t; while (is >> t) if (t.compare(0, 2, "-D") == 0) o.push_back(t); }
  return o;
}

uint64_t fnv1a(const std::string& s, uint64_t h = 0xcbf29ce484222325ull) { for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; } return h; }

// Every run-time-compiled kernel gets a symbol of its own: plx_jit_<kind>_<sink number>_<8 hex digits of the program shape>.  Tracers (rocprofv3 --kernel-trace /
// --pmc) key their rows by symbol: under one shared name the scatter and the aggregation pass of a partitioned group-by were one row of the statistics.
std::string kernel_symbol(const Shape& sh, Sink sink) {
  std::string key((const char*)&sh, sizeof(Shape));
  key.push_back((char)sink);
  char out[96];
  snprintf(out, sizeof out, "plx_jit_%s_%d_%08x", sink_type(sink), (int)sink, (unsigned)(fnv1a(key) & 0xffffffffu));
  return out;
}

std::string source_for(const Shape& sh, Sink sink) {
  std::ostringstream o;
  const std::string sym = kernel_symbol(sh, sink);
  o << "#define plx_jit_kernel " << sym << "\n";
  o << ((sink == BALLOT || sink == DIRECT_HITS) ? "#include \"fused_sinks.hpp\"\n" : sink >= PART3_SCATTER ? "#include \"partition3_device.hpp\"\n" : sink >= PART2_SCATTER_HASH ? "#include \"partition2_device.hpp\"\n" : sink >= PART_COUNT ? "#include \"partition_device.hpp\"\n" : "#include \"fused_sinks.hpp\"\n") << "namespace plx { namespace k {\n"
       "struct JitProg {\n  static constexpr bool kStatic = true; static constexpr int kId = -2;\n  static constexpr Shape shape() {\n    Shape s{};\n";
  o << "    s.n_inputs = " << (int)sh.n_inputs << "; s.n_ops = " << (int)sh.n_ops << "; s.n_aggs = " << (int)sh.n_aggs << "; s.pred = " << (int)sh.pred
    << "; s.key = " << (int)sh.key << "; s.n_keys = " << (int)sh.n_keys << ";\n";
  for (int i = 0; i < kMaxKeys; i++) o << "    s.keys[" << i << "] = " << (int)sh.keys[i] << ";\n";
  for (int i = 0; i < sh.n_inputs; i++) o << "    s.in_dtype[" << i << "] = " << (int)sh.in_dtype[i] << "; s.in_nullable[" << i << "] = " << (int)sh.in_nullable[i] << ";\n";
  for (int i = 0; i < sh.n_ops; i++)
    o << "    s.ops[" << i << "] = mkop(" << (int)sh.ops[i].code << ", " << (int)sh.ops[i].dst << ", " << (int)sh.ops[i].a << ", " << (int)sh.ops[i].b << ", " << (int)sh.ops[i].c << ");\n";
  for (int i = 0; i < sh.n_aggs; i++) o << "    s.aggs[" << i << "] = Agg{" << (int)sh.aggs[i].kind << ", " << (int)sh.aggs[i].src << "};\n";
  o << "    return s;\n  }\n};\n";
  switch (sink) {
    case PART_COUNT:
      o << "extern \"C\" __global__ __launch_bounds__(kBlock) void plx_jit_kernel(Shape dsh, Args args, uint32_t log2_parts, unsigned int* hist) {\n"
           "  part_count_body<JitProg>(dsh, args, log2_parts, hist);\n}\n}}\n";
      break;
    case PART_SCATTER:
      o << "extern \"C\" __global__ __launch_bounds__(kBlock) void plx_jit_kernel(Shape dsh, Args args, PartitionPlan pp, const unsigned long long* part_off, "
           "const unsigned long long* wg_prefix, unsigned long long* out) {\n  part_scatter_body<JitProg>(dsh, args, pp, part_off, wg_prefix, out);\n}\n}}\n";
      break;
    case PART_AGG:
      o << "extern \"C\" __global__ __launch_bounds__(kAggBlock) void plx_jit_kernel(Shape dsh, PartitionPlan pp, PartAggParams ap) {\n"
           "  constexpr Shape csh = JitProg::shape(); constexpr RecLayout cl = rec_layout(JitProg::shape());\n  part_agg_body(csh, cl, pp, ap);\n}\n}}\n";
      break;
    case PART2_SCATTER_HASH: case PART2_SCATTER_DIRECT: case PART2_SCATTER_HASH_T2: case PART2_SCATTER_DIRECT_T2:
      o << "extern \"C\" __global__ __launch_bounds__(kP2MaxBlock) void plx_jit_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {\n"
           "  part2_scatter_body<JitProg, " << ((sink == PART2_SCATTER_DIRECT || sink == PART2_SCATTER_DIRECT_T2) ? 1 : 0) << ", "
        << ((sink == PART2_SCATTER_HASH_T2 || sink == PART2_SCATTER_DIRECT_T2) ? 2 : 1) << ">(dsh, args, pp, sp);\n}\n}}\n";
      break;
    case PART2_AGG_HASH: case PART2_AGG_DIRECT:
      o << "extern \"C\" __global__ __launch_bounds__(kP2AggBlock) void plx_jit_kernel(PartPlan2 pp, AggParams2 ap) {\n"
           "  constexpr Shape csh = JitProg::shape(); constexpr RecLayout2 cl = rec_layout2(JitProg::shape(), " << (sink == PART2_AGG_DIRECT ? 1 : 0) << "u);\n"
           "  part2_agg_body<Shape, " << (sink == PART2_AGG_DIRECT ? 1 : 0) << ", p2_agg_chunks_in_flight(cl.rec_words)>(csh, cl, pp, ap);\n}\n}}\n";
      break;
    default:
      if (sink == PART3_AGG_PAIRV) {
        o << "extern \"C\" __global__ __launch_bounds__(kP2AggBlock) void plx_jit_kernel(PartPlan2 pp, AggParams2 ap) {\n"
             "  constexpr Shape csh = JitProg::shape(); constexpr RecLayout2 cl = rec_layout2(JitProg::shape(), 0u, " << fused::kPackPairV << "u);\n"
             "  part2_agg_body<Shape, 0, p2_agg_chunks_in_flight(cl.rec_words)>(csh, cl, pp, ap);\n}\n}}\n";
        break;
      }
      if (sink >= PART3_SCATTER_PAIRV && sink < PART3_AGG_PAIRV) {
        const int v = (int)sink - (int)PART3_SCATTER_PAIRV, tiles = 1 + (v & 3), hot = (v >> 2) & 1;
        o << "extern \"C\" __global__ __launch_bounds__(kP2MaxBlock) void plx_jit_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {\n"
             "  part3_scatter_body<JitProg, 0, " << tiles << ", " << fused::kPackPairV << ", " << (hot ? "true" : "false") << ">(dsh, args, pp, sp);\n}\n}}\n";
        break;
      }
      if (sink == PART3_AGG_PAIR || sink == PART3_AGG_PAIR_DIRECT) {
        const int mode = sink == PART3_AGG_PAIR_DIRECT ? 1 : 0;
        o << "extern \"C\" __global__ __launch_bounds__(kP2AggBlock) void plx_jit_kernel(PartPlan2 pp, AggParams2 ap) {\n"
             "  constexpr Shape csh = JitProg::shape(); constexpr RecLayout2 cl = rec_layout2(JitProg::shape(), " << mode << "u, " << fused::kPackPair << "u);\n"
             "  part2_agg_body<Shape, " << mode << ", p2_agg_chunks_in_flight(cl.rec_words)>(csh, cl, pp, ap);\n}\n}}\n";
        break;
      }
      if (sink >= PART3_SCATTER_PAIR && sink < PART3_AGG_PAIR) {
        const int v = (int)sink - (int)PART3_SCATTER_PAIR, tiles = 1 + (v & 3), hot = (v >> 2) & 1, mode = v >> 3;
        o << "extern \"C\" __global__ __launch_bounds__(kP2MaxBlock) void plx_jit_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {\n"
             "  part3_scatter_body<JitProg, " << mode << ", " << tiles << ", " << fused::kPackPair << ", " << (hot ? "true" : "false") << ">(dsh, args, pp, sp);\n}\n}}\n";
        break;
      }
      if (sink >= PART3_AGG && sink != BALLOT && sink != DIRECT_HITS) {
        const int v = (int)sink - (int)PART3_AGG, mode = v & 1, pack = v >> 1;
        o << "extern \"C\" __global__ __launch_bounds__(kP2AggBlock) void plx_jit_kernel(PartPlan2 pp, AggParams2 ap) {\n"
             "  constexpr Shape csh = JitProg::shape(); constexpr RecLayout2 cl = rec_layout2(JitProg::shape(), " << mode << "u, " << pack << "u);\n"
             "  part2_agg_body<Shape, " << mode << ", p2_agg_chunks_in_flight(cl.rec_words)>(csh, cl, pp, ap);\n}\n}}\n";
        break;
      }
      if (sink >= PART3_SCATTER && sink != BALLOT && sink != DIRECT_HITS) {
        const int v = (int)sink - (int)PART3_SCATTER, mode = v & 1, tiles = 1 + ((v >> 1) & 3), pack = (v >> 3) & 3, hot = v >> 5;
        o << "extern \"C\" __global__ __launch_bounds__(kP2MaxBlock) void plx_jit_kernel(Shape dsh, Args args, PartPlan2 pp, ScatterParams2 sp) {\n"
             "  part3_scatter_body<JitProg, " << mode << ", " << tiles << ", " << pack << ", " << (hot ? "true" : "false") << ">(dsh, args, pp, sp);\n}\n}}\n";
        break;
      }
      o << "extern \"C\" __global__ __launch_bounds__(kBlock) void plx_jit_kernel(Shape dsh, Args args, " << sink_type(sink)
        << "::Params sp) {\n  fused_scan_body<JitProg, " << sink_type(sink) << ">(dsh, args, sp);\n}\n}}\n";
      break;
  }
  return o.str();
}

std::atomic<int64_t> g_min_rows{-2};   // -2: not initialised, -1: disabled
bool enabled(int64_t n_rows) {
  int64_t m = g_min_rows.load(std::memory_order_relaxed);
  if (m == -2) {
    const char* off = getenv("PLX_JIT");
    const char* e = getenv("PLX_JIT_MIN_ROWS");
    m = (off && off[0] == '0') ? -1 : (e ? (int64_t)atoll(e) : (int64_t)1 << 22);
    g_min_rows.store(m, std::memory_order_relaxed);
  }
  return m >= 0 && n_rows >= m;
}

// ---- disk cache of compiled code objects ------------------------------------------------------------------------------
// A shape without an AOT kernel costs 260-290 ms of hiprtc per process; the code object only depends on the generated source, the
// compile options and the device headers, so it is kept in $PLX_JIT_CACHE_DIR (default ~/.cache/polars_amd/jit; PLX_JIT_CACHE=0
// disables it) under a hash of exactly those inputs.  A cold process then loads the kernel in ~1 ms.
std::string cache_dir() {
  const char* off = getenv("PLX_JIT_CACHE");
  if (off && off[0] == '0') return "";
  if (const char* d = getenv("PLX_JIT_CACHE_DIR")) return d;
  const char* home = getenv("HOME");
  return home && *home ? std::string(home) + "/.cache/polars_amd/jit" : std::string("/tmp/polars_amd_jit");
}
std::string headers_fingerprint() {   // the device headers the generated source includes: a stale cache entry must never survive an upgrade
  static const std::string fp = [] {
    uint64_t h = 0xcbf29ce484222325ull;
    for (const char* f : {"fused.hpp", "fused_device.hpp", "fused_sinks.hpp", "fused_shapes.hpp", "partition_device.hpp", "partition2_device.hpp", "partition3_device.hpp", "dev.hpp", "kconfig.hpp"}) {
      if (FILE* fp2 = fopen((include_dir() + "/" + f).c_str(), "rb")) {
        char buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp2)) > 0) h = fnv1a(std::string(buf, n), h);
        fclose(fp2);
      }
    }
    char out[32]; snprintf(out, sizeof out, "%016llx", (unsigned long long)h);
    return std::string(out);
  }();
  return fp;
}
std::string cache_path(const std::string& src) {
  const std::string dir = cache_dir();
  if (dir.empty()) return "";
  std::string key = src + "|" + headers_fingerprint();
  for (auto& o : compile_options()) if (o.compare(0, 2, "-I") != 0) key += "|" + o;
  char name[64]; snprintf(name, sizeof name, "/%016llx%016llx.hsaco", (unsigned long long)fnv1a(key), (unsigned long long)fnv1a(key, 0x9e3779b97f4a7c15ull));
  return dir + name;
}
bool load_cached(const std::string& path, std::vector<char>* code) {
  if (path.empty()) return false;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  bool ok = n > 0;
  if (ok) { code->resize((size_t)n); ok = fread(code->data(), 1, (size_t)n, f) == (size_t)n; }
  fclose(f);
  return ok;
}
void store_cached(const std::string& path, const std::vector<char>& code) {
  if (path.empty()) return;
  const std::string dir = path.substr(0, path.rfind('/'));
  std::string acc;
  for (size_t i = 1; i <= dir.size(); i++) if (i == dir.size() || dir[i] == '/') { acc = dir.substr(0, i); (void)mkdir(acc.c_str(), 0755); }
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  if (FILE* f = fopen(tmp.c_str(), "wb")) {
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());     // atomic publish: readers never see a partial file
  }
}
int g_cache_hits = 0;

Entry compile(const Shape& sh, Sink sink) {
  Entry e;
  const auto t0 = std::chrono::steady_clock::now();
  const std::string src = source_for(sh, sink);
  const std::string cpath = cache_path(src);
  {
    std::vector<char> cached;
    if (load_cached(cpath, &cached) && hipModuleLoadData(&e.mod, cached.data()) == hipSuccess && hipModuleGetFunction(&e.fn, e.mod, kernel_symbol(sh, sink).c_str()) == hipSuccess) {
      g_cache_hits++; g_compiled++;      // stats(): specialised kernels made available, compiled or loaded
      if (getenv("PLX_JIT_VERBOSE")) fprintf(stderr, "[plx jit] %s: loaded from %s\n", sink_type(sink), cpath.c_str());
      return e;
    }
    (void)hipGetLastError();
    e = Entry{};
  }
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "plx_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { e.failed = true; return e; }
  const std::vector<std::string> optv = compile_options();
  std::vector<const char*> opts;
  for (auto& x : optv) opts.push_back(x.c_str());
  const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, 0); if (n) hiprtcGetProgramLog(prog, &log[0]);
    set_last_error("jit: hiprtc compile failed: " + log.substr(0, 2000));
    if (getenv("PLX_JIT_VERBOSE")) fprintf(stderr, "[plx jit] compile failed:\n%s\n", log.c_str());
    hiprtcDestroyProgram(&prog);
    e.failed = true;
    return e;
  }
  size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (hipModuleLoadData(&e.mod, code.data()) != hipSuccess || hipModuleGetFunction(&e.fn, e.mod, kernel_symbol(sh, sink).c_str()) != hipSuccess) { e.failed = true; return e; }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  g_compiled++; g_ms += ms;
  if (getenv("PLX_JIT_VERBOSE")) fprintf(stderr, "[plx jit] %s: %.0f ms, %zu bytes\n", sink_type(sink), ms, cs);
  return e;
}
}  // namespace

// compile-only check (no GPU needed: hiprtc cross-compiles): returns "" on success, else the compiler log
std::string selftest(const Shape& sh, Sink sink) {
  const std::string src = source_for(sh, sink);
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "plx_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return "hiprtcCreateProgram failed";
  const std::vector<std::string> optv = compile_options();
  std::vector<const char*> opts;
  for (auto& x : optv) opts.push_back(x.c_str());
  const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  std::string log;
  if (rc != HIPRTC_SUCCESS) { size_t n = 0; hiprtcGetProgramLogSize(prog, &n); log.assign(n, 0); if (n) hiprtcGetProgramLog(prog, &log[0]); if (log.empty()) log = "hiprtc error"; }
  else if (const char* dir = getenv("PLX_JIT_DUMP_DIR")) {
    // the code object next to its source, one pair per sink: `llvm-readelf --notes` on it gives registers / scratch of a run-time-compiled kernel without a GPU
    // (tests/test_kernel_resources_cpu.py)
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    const std::string base = std::string(dir) + "/" + sink_type(sink) + "_" + std::to_string((int)sink);
    if (FILE* f = fopen((base + ".hsaco").c_str(), "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
    if (FILE* f = fopen((base + ".hip").c_str(), "wb")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
  }
  hiprtcDestroyProgram(&prog);
  return log;
}

const char* program_mode(int static_id, int64_t n_rows) { return static_id >= 0 ? "aot" : (enabled(n_rows) ? "jit" : "generic"); }

void set_min_rows(int64_t min_rows) { g_min_rows.store(min_rows < 0 ? -1 : min_rows, std::memory_order_relaxed); }

void stats(int* compiled, double* compile_ms) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (compiled) *compiled = g_compiled;
  if (compile_ms) *compile_ms = g_ms;
}

static hipFunction_t get_fn(const Shape& sh, Sink sink) {
  std::string key((const char*)&sh, sizeof(Shape));
  key.push_back((char)sink);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_cache.find(key);
  if (it == g_cache.end()) it = g_cache.emplace(key, compile(sh, sink)).first;
  return it->second.failed ? nullptr : it->second.fn;
}

bool ensure(const Shape& sh, Sink kind, int64_t n_rows) { return enabled(n_rows) && get_fn(sh, kind) != nullptr; }

bool launch_raw(const Shape& sh, Sink kind, void** kargs, int grid, int block, size_t lds_bytes) {
  hipFunction_t fn = get_fn(sh, kind);
  if (!fn) return false;
  const hipError_t rc = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds_bytes, stream(), kargs, nullptr);
  if (rc != hipSuccess) { set_last_error(std::string("jit: launch failed: ") + hipGetErrorString(rc)); return false; }
  return true;
}

bool launch(const Shape& sh, const Args& args, Sink sink, const void* params, int grid, size_t lds_bytes) {
  if (!enabled(args.n_rows)) return false;
  Shape shc = sh; Args ac = args;
  void* kargs[] = {(void*)&shc, (void*)&ac, const_cast<void*>(params)};
  return launch_raw(sh, sink, kargs, grid, k::kBlock, lds_bytes);
}

}  // namespace jit
}  // namespace plx
