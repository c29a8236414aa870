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

#include <algorithm>
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
  int (*CommCount)(const NcclComm, int*) = nullptr;      // optional: what RCCL itself says the communicator is (plx_comm_info)
  int (*CommUserRank)(const NcclComm, int*) = nullptr;
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
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
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

// One ncclSend / ncclRecv carries at most kP2PChunk bytes.  Measured on RCCL 2.26.6 (MI355X, gpurun_out/r03x, tools/debug_exchange2.py): a transfer of more than
// 2^30 bytes delivers only its first half -- 134e6 rows x 8 B arrive whole, 135e6 rows leave 67.5e6 zero rows behind, no error code.  Larger transfers
// are cut into pieces of 2^29 bytes inside the same group; both sides derive the piece count from the same byte count, so sends and receives pair up.
constexpr size_t kP2PChunk = size_t(1) << 29;
void send_chunked(const uint8_t* src, size_t bytes, int peer, Comm& c) {
  for (size_t o = 0; o < bytes; o += kP2PChunk) nccl_check(rccl().Send(src + o, std::min(kP2PChunk, bytes - o), kNcclUint8, peer, c.nccl, stream()), "ncclSend");
}
void recv_chunked(uint8_t* dst, size_t bytes, int peer, Comm& c) {
  for (size_t o = 0; o < bytes; o += kP2PChunk) nccl_check(rccl().Recv(dst + o, std::min(kP2PChunk, bytes - o), kNcclUint8, peer, c.nccl, stream()), "ncclRecv");
}
std::mutex g_comm_mu;
std::vector<std::unique_ptr<Comm>> g_comms;   // handle = index + 1

Comm& get_comm(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (h == 0 || h > g_comms.size() || !g_comms[h - 1]) fail(PLX_ERR_INVALID, "invalid communicator handle");
  return *g_comms[h - 1];
}

// One ncclAllGather of `n` int64 per rank + ONE D2H: all[r * n + i] = value i of rank r.  `send` is a device buffer of n int64.
std::vector<int64_t> allgather_i64(Comm& c, const Buf& send, size_t n) {
  const int ws = c.ws;
  Buf all = dev_alloc(sizeof(int64_t) * n * (size_t)ws);
  nccl_check(rccl().AllGather(send->ptr, all->ptr, n, kNcclInt64, c.nccl, stream()), "ncclAllGather(counts)");
  std::vector<int64_t> host(n * (size_t)ws);
  d2h_sync(host.data(), all->ptr, host.size() * 8);
  return host;
}

// What crosses the fabric for one column: fixed-width, null-free byte arrays ("wires").  Bit-packed buffers cannot be sliced per
// destination at arbitrary row offsets, so Boolean values and validity bitmaps travel as one byte per row and are re-packed on receipt
// (the reference's partitioner hands out row index lists and gathers, crates/polars-expr/src/hash_keys.rs:263-314: there is no bit
// slicing there either).
struct Wire { ColumnPtr data; size_t width; };
ColumnPtr bitmap_as_bytes(const Buf& bits, int64_t n) {
  auto b = std::make_shared<Column>();
  b->dtype = PLX_BOOL; b->len = n; b->values = bits; b->null_count = 0;
  return ops::cast(b, PLX_U8);
}
Buf bytes_as_bitmap(const ColumnPtr& bytes) {
  plx_scalar zero; zero.u = 0;
  return ops::cmp_scalar(PLX_NE, bytes, zero)->values;
}
// the wires of column `col` (already in its final row order): values, then validity when `with_validity`
void column_wires(const ColumnPtr& col, bool with_validity, std::vector<Wire>& out) {
  if (col->dtype == PLX_BOOL) out.push_back({bitmap_as_bytes(col->values, col->len), 1});
  else {
    auto v = std::make_shared<Column>(*col);
    v->validity = nullptr;
    out.push_back({v, (size_t)dtype_width(col->dtype)});
  }
  if (with_validity) {
    if (col->validity) out.push_back({bitmap_as_bytes(col->validity, col->len), 1});
    else { plx_scalar one; one.u = 1; out.push_back({ops::full_column(PLX_U8, one, true, col->len), 1}); }
  }
}
// received wires -> column
ColumnPtr column_from_wires(int dtype, int64_t n, bool with_validity, const std::vector<ColumnPtr>& recv, size_t& wi) {
  ColumnPtr o;
  if (dtype == PLX_BOOL) {
    o = std::make_shared<Column>();
    o->dtype = PLX_BOOL; o->len = n; o->values = bytes_as_bitmap(recv[wi++]);
  } else o = recv[wi++];
  o->dtype = dtype;
  if (with_validity) { o->validity = bytes_as_bitmap(recv[wi++]); o->null_count = -1; }
  else { o->validity = nullptr; o->null_count = 0; }
  return o;
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

// rank / world size as RCCL reports them for the live communicator (ncclCommUserRank / ncclCommCount), checked against what init() was given
void info(uint64_t h, int* rank, int* ws) {
  Comm& c = get_comm(h);
  int r = c.rank, w = c.ws;
  if (c.nccl && rccl().CommCount && rccl().CommUserRank) {
    nccl_check(rccl().CommCount(c.nccl, &w), "ncclCommCount");
    nccl_check(rccl().CommUserRank(c.nccl, &r), "ncclCommUserRank");
    PLX_REQUIRE(r == c.rank && w == c.ws, PLX_ERR_HIP, "communicator disagrees with its initialisation: rank " + std::to_string(r) + " of " + std::to_string(w) +
                " (initialised as " + std::to_string(c.rank) + " of " + std::to_string(c.ws) + ")");
  }
  if (rank) *rank = r;
  if (ws) *ws = w;
}

// Routes every row of `in` to rank hash_partition(key) -- null keys to rank 0 (null_partition(), hashing.rs:111-115) -- and returns
// the rows this rank received (all columns, same names; nullable and Boolean columns included); in *rows_sent / *bytes_sent what
// left this rank over the fabric (rows kept locally do not count).  Host round trips: ONE (the [ws x ws] counts + the per-column
// "some rank has nulls here" flags, needed for the receive allocations); the transfers are one ncclGroup = one fused RCCL kernel over
// all seven xGMI links, so columns are not packed into a per-peer staging buffer (that would add two D2D passes over the payload and
// save nothing on the wire).  One stream synchronisation at the end (the staged buffers are recycled by the pool after it).
FramePtr exchange_by_key(uint64_t h, const FramePtr& in, const std::string& key, uint64_t seed, uint64_t* rows_sent, uint64_t* bytes_sent) {
  Comm& c = get_comm(h);
  const int ws = c.ws;
  const int ki = in->find(key);
  PLX_REQUIRE(ki >= 0, PLX_ERR_NOT_FOUND, "exchange_by_key: key column not found: " + key);
  const size_t nc = in->cols.size();
  // destination of every row and the permutation that groups rows by destination (plx_hash_partition); counts stay on the device
  ColumnPtr perm;
  Buf counts_dev;
  join::hash_partition_dev(in->cols[ki], ws, seed, perm, counts_dev);
  Buf send = dev_alloc(sizeof(int64_t) * ((size_t)ws + nc));
  PLX_HIP(hipMemcpyAsync(send->ptr, counts_dev->ptr, sizeof(int64_t) * (size_t)ws, hipMemcpyDeviceToDevice, stream()));
  std::vector<int64_t> flags(nc, 0);
  for (size_t i = 0; i < nc; i++) flags[i] = (in->cols[i]->validity && in->cols[i]->null_count != 0) ? 1 : 0;   // an unknown null count (-1) counts as "may have nulls": no popcount + host sync here
  if (nc) h2d_async((uint8_t*)send->ptr + sizeof(int64_t) * (size_t)ws, flags.data(), sizeof(int64_t) * nc);
  const size_t stride = (size_t)ws + nc;
  const std::vector<int64_t> all = allgather_i64(c, send, stride);    // all[r * stride + d] = rows rank r sends to rank d; [.. + ws + i] = rank r has nulls in column i
  std::vector<int64_t> send_cnt((size_t)ws), recv_cnt((size_t)ws), send_off((size_t)ws + 1, 0), recv_off((size_t)ws + 1, 0);
  for (int r = 0; r < ws; r++) { send_cnt[r] = all[(size_t)c.rank * stride + r]; recv_cnt[r] = all[(size_t)r * stride + c.rank]; }
  for (int r = 0; r < ws; r++) { send_off[r + 1] = send_off[r] + send_cnt[r]; recv_off[r + 1] = recv_off[r] + recv_cnt[r]; }
  PLX_REQUIRE(send_off[ws] == in->height, PLX_ERR_INVALID, "exchange_by_key: partition counts do not add up to the frame height");
  const int64_t n_out = recv_off[ws];
  std::vector<char> nullable(nc, 0);
  for (size_t i = 0; i < nc; i++) for (int r = 0; r < ws; r++) if (all[(size_t)r * stride + ws + i]) nullable[i] = 1;
  std::vector<Wire> wires;
  for (size_t i = 0; i < nc; i++) column_wires(ops::gather(in->cols[i], perm), nullable[i], wires);     // destination order, contiguous per rank
  std::vector<ColumnPtr> recv;
  for (const Wire& w : wires) { ColumnPtr o = make_column(w.width == 1 ? PLX_U8 : w.width == 2 ? PLX_U16 : w.width == 4 ? PLX_U32 : PLX_U64, n_out, false); o->null_count = 0; recv.push_back(o); }
  uint64_t moved_rows = 0, moved_bytes = 0;
  {
    ProfileScope ps("rccl_all_to_all_v", 0, (uint64_t)in->height);
    nccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (size_t wi = 0; wi < wires.size(); wi++) {
      const size_t w = wires[wi].width;
      const uint8_t* src = (const uint8_t*)wires[wi].data->values->ptr;
      uint8_t* dst = (uint8_t*)recv[wi]->values->ptr;
      for (int p = 0; p < ws; p++) {
        if (send_cnt[p]) send_chunked(src + (size_t)send_off[p] * w, (size_t)send_cnt[p] * w, p, c);
        if (recv_cnt[p]) recv_chunked(dst + (size_t)recv_off[p] * w, (size_t)recv_cnt[p] * w, p, c);
        if (p != c.rank) moved_bytes += (uint64_t)send_cnt[p] * w;
      }
    }
    nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  }
  for (int p = 0; p < ws; p++) if (p != c.rank) moved_rows += (uint64_t)send_cnt[p];
  // The staged (gathered) wires go back to the pool when this returns.  The pool recycles in stream order, but whether every transfer of an RCCL group
  // is ordered on the caller's stream alone is RCCL's business (proxy threads, internal streams for local copies): wait for the exchange before the
  // buffers can be handed out again.  The consumer of the exchanged frame needs it finished anyway.
  PLX_HIP(hipStreamSynchronize(stream()));
  auto out = std::make_shared<Frame>();
  out->height = n_out; out->names = in->names;
  size_t wi = 0;
  for (size_t i = 0; i < nc; i++) out->cols.push_back(column_from_wires(in->cols[i]->dtype, n_out, nullable[i], recv, wi));
  if (rows_sent) *rows_sent = moved_rows;
  if (bytes_sent) *bytes_sent = moved_bytes;
  return out;
}

// Concatenation of every rank's frame, in rank order, on every rank (variable lengths: counts first, then one grouped
// send / recv per wire -- an all-gather(v)); nullable and Boolean columns travel like in exchange_by_key.
FramePtr allgather_frame(uint64_t h, const FramePtr& in) {
  Comm& c = get_comm(h);
  const int ws = c.ws;
  const size_t nc = in->cols.size();
  std::vector<int64_t> mine(1 + nc, 0);
  mine[0] = in->height;
  for (size_t i = 0; i < nc; i++) mine[1 + i] = (in->cols[i]->validity && in->cols[i]->null_count != 0) ? 1 : 0;
  Buf send = dev_alloc(sizeof(int64_t) * mine.size());
  h2d_async(send->ptr, mine.data(), sizeof(int64_t) * mine.size());
  const size_t stride = mine.size();
  const std::vector<int64_t> all = allgather_i64(c, send, stride);
  std::vector<int64_t> off((size_t)ws + 1, 0);
  for (int r = 0; r < ws; r++) off[r + 1] = off[r] + all[(size_t)r * stride];
  std::vector<char> nullable(nc, 0);
  for (size_t i = 0; i < nc; i++) for (int r = 0; r < ws; r++) if (all[(size_t)r * stride + 1 + i]) nullable[i] = 1;
  std::vector<Wire> wires;
  for (size_t i = 0; i < nc; i++) column_wires(in->cols[i], nullable[i], wires);
  std::vector<ColumnPtr> recv;
  for (const Wire& w : wires) { ColumnPtr o = make_column(w.width == 1 ? PLX_U8 : w.width == 2 ? PLX_U16 : w.width == 4 ? PLX_U32 : PLX_U64, off[ws], false); o->null_count = 0; recv.push_back(o); }
  nccl_check(rccl().GroupStart(), "ncclGroupStart");
  for (size_t wi = 0; wi < wires.size(); wi++) {
    const size_t w = wires[wi].width;
    uint8_t* dst = (uint8_t*)recv[wi]->values->ptr;
    for (int p = 0; p < ws; p++) {
      if (in->height) send_chunked((const uint8_t*)wires[wi].data->values->ptr, (size_t)in->height * w, p, c);
      const int64_t n = all[(size_t)p * stride];
      if (n) recv_chunked(dst + (size_t)off[p] * w, (size_t)n * w, p, c);
    }
  }
  nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  PLX_HIP(hipStreamSynchronize(stream()));       // as in exchange_by_key: the wires are recycled after this
  auto out = std::make_shared<Frame>();
  out->height = off[ws]; out->names = in->names;
  size_t wi = 0;
  for (size_t i = 0; i < nc; i++) out->cols.push_back(column_from_wires(in->cols[i]->dtype, off[ws], nullable[i], recv, wi));
  return out;
}

}  // namespace comm
}  // namespace plx
