// comm.cpp -- the exchange step of the sharded operators, on RCCL directly (one process per GPU, xGMI between them).
//
// The reference has no distributed backend; the shape restated here is its in-process exchange: HashPartitioner
// (crates/polars-utils/src/hashing.rs:72-121) routes every row to the partition of its key hash, partitions are then
// finalised independently (crates/polars-stream/src/nodes/group_by.rs:252-497 combine_locals;
// nodes/joins/equi_join.rs:446-760).  With GPUs as the partitions:
//   plx_exchange_by_key   rows -> destination rank = plx_hash_partition(key) (the kernel the single-GPU parity tests pin against a CPU restatement of HashPartitioner);
//                         every column is gathered into destination order with
//                         the library's gather kernel; the row counts are exchanged with one tiny ncclAllGather; then ONE
//                         grouped ncclSend / ncclRecv all-to-all(v) moves every column (world_size x n_columns transfers in
//                         one ncclGroup) on the library's stream.  No torch kernels, no host staging of rows.
//   plx_allgather_frame   small frames (group partials, a filtered build side) replicated on every rank.
// librccl is loaded with dlopen on first use: a single-GPU process never touches it, and the library has no link-time
// dependency on it.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): an all-to-all keeps all seven links busy at once,
// which is why rows are exchanged in one grouped operation rather than rank by rank.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "core.hpp"
#include "join.hpp"
#include "ops.hpp"

namespace plx {
namespace {

// the handful of RCCL entry points used (signatures from /opt/rocm/include/rccl/rccl.h)
typedef struct { char internal[128]; } NcclId;
typedef void* NcclComm;
enum { kNcclUint8 = 1, kNcclInt64 = 4 };   // ncclDataType_t: ncclUint8 = 1, ncclInt64 = 4
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.so) break; }
    if (!r.so) return;
    auto sym = [&](const char* n) { return dlsym(r.so, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  if (!r.so || !r.GetUniqueId || !r.CommInitRank || !r.Send || !r.Recv || !r.AllGather || !r.GroupStart || !r.GroupEnd)
    fail(PLX_ERR_HIP, "librccl could not be loaded (multi-GPU exchange needs RCCL)");
  return r;
}
void nccl_check(int rc, const char* what) {
  if (rc != 0) fail(PLX_ERR_HIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "rccl error") + " (" + std::to_string(rc) + ")");
}

struct Comm { NcclComm nccl = nullptr; int rank = 0, ws = 1; };
std::mutex g_comm_mu;
std::vector<std::unique_ptr<Comm>> g_comms;   // handle = index + 1

Comm& get_comm(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (h == 0 || h > g_comms.size() || !g_comms[h - 1]) fail(PLX_ERR_INVALID, "invalid communicator handle");
  return *g_comms[h - 1];
}

// per-destination row counts of every rank -> [ws][ws] on the host (one ncclAllGather of ws int64 each + one D2H)
std::vector<int64_t> exchange_counts(Comm& c, const std::vector<int64_t>& mine) {
  const int ws = c.ws;
  Buf send = dev_alloc(sizeof(int64_t) * ws), all = dev_alloc(sizeof(int64_t) * ws * ws);
  h2d_async(send->ptr, mine.data(), sizeof(int64_t) * ws);
  nccl_check(rccl().AllGather(send->ptr, all->ptr, (size_t)ws, kNcclInt64, c.nccl, stream()), "ncclAllGather(counts)");
  std::vector<int64_t> host((size_t)ws * ws);
  d2h_sync(host.data(), all->ptr, host.size() * 8);
  return host;
}

}  // namespace

namespace comm {

void unique_id(uint8_t* out128) {
  NcclId id;
  nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out128, id.internal, 128);
}

uint64_t init(const uint8_t* id128, int rank, int ws) {
  PLX_REQUIRE(ws >= 1 && rank >= 0 && rank < ws && id128, PLX_ERR_INVALID, "comm_init: bad arguments");
  device();
  NcclId id;
  memcpy(id.internal, id128, 128);
  auto c = std::make_unique<Comm>();
  c->rank = rank; c->ws = ws;
  nccl_check(rccl().CommInitRank(&c->nccl, ws, id, rank), "ncclCommInitRank");
  std::lock_guard<std::mutex> lk(g_comm_mu);
  g_comms.push_back(std::move(c));
  return (uint64_t)g_comms.size();
}

void destroy(uint64_t h) {
  std::unique_ptr<Comm> c;
  {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    if (h == 0 || h > g_comms.size() || !g_comms[h - 1]) return;
    c = std::move(g_comms[h - 1]);
  }
  (void)hipStreamSynchronize(stream());
  if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
}

void info(uint64_t h, int* rank, int* ws) { Comm& c = get_comm(h); if (rank) *rank = c.rank; if (ws) *ws = c.ws; }

// Routes every row of `in` to rank hash_partition(key); returns the rows this rank received (all columns, same names) and,
// in *rows_sent / *bytes_sent, what left this rank over the fabric (rows kept locally do not count).
FramePtr exchange_by_key(uint64_t h, const FramePtr& in, const std::string& key, uint64_t seed, uint64_t* rows_sent, uint64_t* bytes_sent) {
  Comm& c = get_comm(h);
  const int ws = c.ws;
  const int ki = in->find(key);
  PLX_REQUIRE(ki >= 0, PLX_ERR_NOT_FOUND, "exchange_by_key: key column not found: " + key);
  for (size_t i = 0; i < in->cols.size(); i++) {
    const ColumnPtr& col = in->cols[i];
    PLX_REQUIRE(col->dtype != PLX_BOOL, PLX_ERR_UNSUPPORTED, "exchange_by_key: bit-packed Boolean columns cannot be sliced per destination (cast to UInt8 first)");
    PLX_REQUIRE(!col->validity || column_null_count(col) == 0, PLX_ERR_UNSUPPORTED, "exchange_by_key: nullable columns are not exchanged yet (" + in->names[i] + ")");
  }
  // destination of every row and the permutation that groups rows by destination (plx_hash_partition)
  ColumnPtr perm;
  std::vector<int64_t> send_cnt((size_t)ws, 0);
  join::hash_partition(in->cols[ki], ws, seed, perm, send_cnt.data());
  const std::vector<int64_t> all = exchange_counts(c, send_cnt);     // all[r * ws + d] = rows rank r sends to rank d
  std::vector<int64_t> recv_cnt((size_t)ws), send_off((size_t)ws + 1, 0), recv_off((size_t)ws + 1, 0);
  for (int r = 0; r < ws; r++) recv_cnt[r] = all[(size_t)r * ws + c.rank];
  for (int r = 0; r < ws; r++) { send_off[r + 1] = send_off[r] + send_cnt[r]; recv_off[r + 1] = recv_off[r] + recv_cnt[r]; }
  const int64_t n_out = recv_off[ws];
  auto out = std::make_shared<Frame>();
  out->height = n_out; out->names = in->names;
  std::vector<ColumnPtr> staged;
  for (const ColumnPtr& col : in->cols) {
    ColumnPtr g = ops::gather(col, perm);                             // destination order, contiguous per rank
    staged.push_back(g);
    ColumnPtr o = make_column(col->dtype, n_out, false);
    o->null_count = 0;
    out->cols.push_back(o);
  }
  uint64_t moved_rows = 0, moved_bytes = 0;
  {
    ProfileScope ps("rccl_all_to_all_v", 0, (uint64_t)in->height);
    nccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (size_t ci = 0; ci < staged.size(); ci++) {
      const size_t w = (size_t)dtype_width(staged[ci]->dtype);
      const uint8_t* src = (const uint8_t*)staged[ci]->values->ptr;
      uint8_t* dst = (uint8_t*)out->cols[ci]->values->ptr;
      for (int p = 0; p < ws; p++) {
        if (send_cnt[p]) nccl_check(rccl().Send(src + (size_t)send_off[p] * w, (size_t)send_cnt[p] * w, kNcclUint8, p, c.nccl, stream()), "ncclSend");
        if (recv_cnt[p]) nccl_check(rccl().Recv(dst + (size_t)recv_off[p] * w, (size_t)recv_cnt[p] * w, kNcclUint8, p, c.nccl, stream()), "ncclRecv");
        if (p != c.rank) { moved_bytes += (uint64_t)send_cnt[p] * w; }
      }
    }
    nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  }
  for (int p = 0; p < ws; p++) if (p != c.rank) moved_rows += (uint64_t)send_cnt[p];
  PLX_HIP(hipStreamSynchronize(stream()));   // the staged buffers go back to the pool when this returns
  if (rows_sent) *rows_sent = moved_rows;
  if (bytes_sent) *bytes_sent = moved_bytes;
  return out;
}

// Concatenation of every rank's frame, in rank order, on every rank (variable lengths: counts first, then one grouped
// send / recv per column -- an all-gather(v)).
FramePtr allgather_frame(uint64_t h, const FramePtr& in) {
  Comm& c = get_comm(h);
  const int ws = c.ws;
  for (size_t i = 0; i < in->cols.size(); i++) {
    PLX_REQUIRE(in->cols[i]->dtype != PLX_BOOL && (!in->cols[i]->validity || column_null_count(in->cols[i]) == 0), PLX_ERR_UNSUPPORTED,
                "allgather_frame: Boolean / nullable columns are not supported yet (" + in->names[i] + ")");
  }
  std::vector<int64_t> mine((size_t)ws, in->height);
  const std::vector<int64_t> all = exchange_counts(c, mine);          // all[r * ws + *] = height of rank r
  std::vector<int64_t> off((size_t)ws + 1, 0);
  for (int r = 0; r < ws; r++) off[r + 1] = off[r] + all[(size_t)r * ws];
  auto out = std::make_shared<Frame>();
  out->height = off[ws]; out->names = in->names;
  for (const ColumnPtr& col : in->cols) { ColumnPtr o = make_column(col->dtype, off[ws], false); o->null_count = 0; out->cols.push_back(o); }
  nccl_check(rccl().GroupStart(), "ncclGroupStart");
  for (size_t ci = 0; ci < in->cols.size(); ci++) {
    const size_t w = (size_t)dtype_width(in->cols[ci]->dtype);
    uint8_t* dst = (uint8_t*)out->cols[ci]->values->ptr;
    for (int p = 0; p < ws; p++) {
      if (in->height) nccl_check(rccl().Send(in->cols[ci]->values->ptr, (size_t)in->height * w, kNcclUint8, p, c.nccl, stream()), "ncclSend");
      const int64_t n = all[(size_t)p * ws];
      if (n) nccl_check(rccl().Recv(dst + (size_t)off[p] * w, (size_t)n * w, kNcclUint8, p, c.nccl, stream()), "ncclRecv");
    }
  }
  nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  PLX_HIP(hipStreamSynchronize(stream()));
  return out;
}

}  // namespace comm
}  // namespace plx
