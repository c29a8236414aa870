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

}  // namespace k
}  // namespace plx
