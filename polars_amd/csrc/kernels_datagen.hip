// kernels_datagen.hip -- synthetic TPC-H lineitem (Q1 columns) generated straight into HBM columns.
// Benchmark / test support: bench.py needs 25 GB of resident input at SF100 and must not depend on another library's kernels
// to produce it.  One pass, 42 B written per row; every row is lineitem_row(seed, i) of datagen_device.hpp, which
// plx_datagen_lineitem_q1_host evaluates on the CPU (tests/test_datagen_cpu.py, bench.py's spot check).
#include "datagen_device.hpp"
#include "dev.hpp"
#include "kernels.hpp"

namespace plx {
namespace k {

__global__ __launch_bounds__(kBlock) void datagen_lineitem_q1_kernel(int64_t n, uint64_t seed, int64_t* __restrict__ shipdate, uint8_t* __restrict__ flag,
                                                                     uint8_t* __restrict__ status, int64_t* __restrict__ qty, double* __restrict__ price,
                                                                     double* __restrict__ disc, double* __restrict__ tax) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const datagen::LineitemRow r = datagen::lineitem_row(seed, (uint64_t)i);
    shipdate[i] = r.shipdate; flag[i] = r.returnflag; status[i] = r.linestatus; qty[i] = r.quantity;
    price[i] = r.extendedprice; disc[i] = r.discount; tax[i] = r.tax;
  }
}

void datagen_lineitem_q1(int64_t n, uint64_t seed, int64_t* shipdate, uint8_t* flag, uint8_t* status, int64_t* qty, double* price, double* disc, double* tax) {
  if (n <= 0) return;
  ProfileScope ps("datagen_lineitem_q1", (uint64_t)n * 42, (uint64_t)n);
  hipLaunchKernelGGL(datagen_lineitem_q1_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), n, seed, shipdate, flag, status, qty, price, disc, tax);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void datagen_orders_kernel(int64_t n, uint64_t seed, int64_t cust_hi, int64_t* __restrict__ okey, int64_t* __restrict__ cust,
                                                                int64_t* __restrict__ odate, int64_t* __restrict__ prio, uint32_t* __restrict__ n_lines) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const datagen::OrderRow r = datagen::order_row(seed, (uint64_t)i, cust_hi);
    okey[i] = r.orderkey; cust[i] = r.custkey; odate[i] = r.orderdate; prio[i] = 0; n_lines[i] = r.n_lines;
  }
}
// one thread per order writes its 1-7 lines at offsets[i] .. offsets[i + 1]: neighbouring threads write neighbouring runs
__global__ __launch_bounds__(kBlock) void datagen_lines_kernel(int64_t n_orders, uint64_t seed, const uint64_t* __restrict__ offsets, const int64_t* __restrict__ okey,
                                                               const int64_t* __restrict__ odate, int64_t* __restrict__ lkey, double* __restrict__ price,
                                                               double* __restrict__ disc, int64_t* __restrict__ ship) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_orders; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t beg = offsets[i], end = offsets[i + 1];
    const int64_t key = okey[i], date = odate[i];
    for (uint64_t o = beg; o < end; o++) {
      const datagen::Q3LineRow r = datagen::q3_line_row(seed, (uint64_t)i, (uint32_t)(o - beg), date);
      lkey[o] = key; price[o] = r.extendedprice; disc[o] = r.discount; ship[o] = r.shipdate;
    }
  }
}
void datagen_orders(int64_t n, uint64_t seed, int64_t cust_hi, int64_t* okey, int64_t* cust, int64_t* odate, int64_t* prio, uint32_t* n_lines) {
  if (n <= 0) return;
  ProfileScope ps("datagen_orders", (uint64_t)n * 36, (uint64_t)n);
  hipLaunchKernelGGL(datagen_orders_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), n, seed, cust_hi, okey, cust, odate, prio, n_lines);
  PLX_HIP(hipGetLastError());
}
void datagen_lines(int64_t n_orders, uint64_t seed, const uint64_t* offsets, const int64_t* okey, const int64_t* odate, int64_t* lkey, double* price, double* disc, int64_t* ship) {
  if (n_orders <= 0) return;
  ProfileScope ps("datagen_lines", (uint64_t)n_orders * 4 * 32, (uint64_t)n_orders);
  hipLaunchKernelGGL(datagen_lines_kernel, dim3(grid_for(n_orders, kBlock * 2)), dim3(kBlock), 0, stream(), n_orders, seed, offsets, okey, odate, lkey, price, disc, ship);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void datagen_customer_kernel(int64_t n, uint64_t seed, int64_t* __restrict__ custkey, uint8_t* __restrict__ segment) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    custkey[i] = i + 1;
    segment[i] = datagen::customer_segment(seed, (uint64_t)(i + 1));
  }
}
void datagen_customer(int64_t n, uint64_t seed, int64_t* custkey, uint8_t* segment) {
  if (n <= 0) return;
  ProfileScope ps("datagen_customer", (uint64_t)n * 9, (uint64_t)n);
  hipLaunchKernelGGL(datagen_customer_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), n, seed, custkey, segment);
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void datagen_id_views_kernel(int64_t n, uint64_t seed, uint32_t strm, int64_t lo, int64_t hi, ulonglong2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t w0, w1;
    datagen::id_view((uint64_t)datagen::uniform_value(seed, strm, (uint64_t)i, lo, hi), &w0, &w1);
    out[i] = make_ulonglong2(w0, w1);
  }
}
void datagen_id_views(int64_t n, uint64_t seed, uint32_t strm, int64_t lo, int64_t hi, uint64_t* out_views) {
  if (n <= 0) return;
  ProfileScope ps("datagen_id_views", (uint64_t)n * 16, (uint64_t)n);
  hipLaunchKernelGGL(datagen_id_views_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), n, seed, strm, lo, hi, reinterpret_cast<ulonglong2*>(out_views));
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void datagen_long_id_views_kernel(int64_t n, uint64_t seed, uint32_t strm, int64_t lo, int64_t hi, ulonglong2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t w0, w1;
    datagen::long_id_view((uint64_t)datagen::uniform_value(seed, strm, (uint64_t)i, lo, hi), (uint64_t)lo, &w0, &w1);
    out[i] = make_ulonglong2(w0, w1);
  }
}
__global__ __launch_bounds__(kBlock) void datagen_long_id_pool_kernel(int64_t lo, int64_t hi, unsigned char* __restrict__ pool) {
  for (int64_t v = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < hi; v += (int64_t)gridDim.x * blockDim.x) datagen::long_id_bytes((uint64_t)v, pool + (v - lo) * datagen::kLongIdLen);
}
void datagen_long_id_views(int64_t n, uint64_t seed, uint32_t strm, int64_t lo, int64_t hi, uint64_t* out_views, uint8_t* out_pool) {
  ProfileScope ps("datagen_id_views", (uint64_t)n * 16 + (uint64_t)(hi - lo) * datagen::kLongIdLen, (uint64_t)n);
  hipLaunchKernelGGL(datagen_long_id_pool_kernel, dim3(grid_for(hi - lo, kBlock)), dim3(kBlock), 0, stream(), lo, hi, out_pool);
  PLX_HIP(hipGetLastError());
  if (n <= 0) return;
  hipLaunchKernelGGL(datagen_long_id_views_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), n, seed, strm, lo, hi, reinterpret_cast<ulonglong2*>(out_views));
  PLX_HIP(hipGetLastError());
}

template <class T>
__global__ __launch_bounds__(kBlock) void datagen_uniform_kernel(int64_t n, uint64_t seed, uint32_t strm, int64_t lo, int64_t hi, double scale, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = datagen::uniform_value(seed, strm, (uint64_t)i, lo, hi);
    if constexpr (sizeof(T) == 8 && !dev::is_fp<T>::value) out[i] = (T)v;
    else if constexpr (dev::is_fp<T>::value) out[i] = (T)((double)v * scale);
    else out[i] = (T)v;
  }
}
void datagen_uniform(int dtype, int64_t n, uint64_t seed, uint32_t strm, int64_t lo, int64_t hi, double scale, void* out) {
  if (n <= 0) return;
  ProfileScope ps("datagen_uniform", (uint64_t)n * dtype_width(dtype), (uint64_t)n);
  const int grid = grid_for(n, kBlock * 4);
  switch (dtype) {
    case PLX_I64: hipLaunchKernelGGL((datagen_uniform_kernel<int64_t>), dim3(grid), dim3(kBlock), 0, stream(), n, seed, strm, lo, hi, scale, (int64_t*)out); break;
    case PLX_U32: hipLaunchKernelGGL((datagen_uniform_kernel<uint32_t>), dim3(grid), dim3(kBlock), 0, stream(), n, seed, strm, lo, hi, scale, (uint32_t*)out); break;
    case PLX_F64: hipLaunchKernelGGL((datagen_uniform_kernel<double>), dim3(grid), dim3(kBlock), 0, stream(), n, seed, strm, lo, hi, scale, (double*)out); break;
    default: fail(PLX_ERR_UNSUPPORTED, "datagen_uniform: dtype must be Int64, UInt32 or Float64");
  }
  PLX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void datagen_zipf_kernel(int64_t n, uint64_t seed, uint32_t strm, uint64_t x0_q62, int64_t n_keys, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = datagen::zipf_value(seed, strm, (uint64_t)i, x0_q62, n_keys);
}
void datagen_zipf(int64_t n, uint64_t seed, uint32_t strm, uint64_t x0_q62, int64_t n_keys, int64_t* out) {
  if (n <= 0) return;
  ProfileScope ps("datagen_zipf", (uint64_t)n * 8, (uint64_t)n);
  hipLaunchKernelGGL(datagen_zipf_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, stream(), n, seed, strm, x0_q62, n_keys, out);
  PLX_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace plx
