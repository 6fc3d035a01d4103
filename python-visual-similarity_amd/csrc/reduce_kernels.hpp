// Deterministic reductions shared by the training passes (fisher.hip: EM step, learn.hip: Lloyd / Gram / seeding):
// chunk partial sums are added in chunk order in fp64, single-block sums use a fixed tree -- run-to-run identical.
#pragma once
#include "common.hpp"

namespace pvs {

// off[i] = t0 + min(i * chunk, tn), i = 0 .. nchunks: CSR offsets of fixed-size pseudo-images over rows [t0, t0 + tn)
static __global__ void chunk_offsets_kernel(int64_t* off, int64_t t0, int64_t tn, int chunk, int64_t nchunks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= nchunks) off[i] = t0 + (i * chunk < tn ? i * chunk : tn);
}

// acc[j] (+)= part[0][j] + part[1][j] + ...  in chunk order
template <typename T>
static __global__ __launch_bounds__(256) void reduce_chunks_kernel(const T* __restrict__ part, int64_t nchunks, int64_t len,
                                                                   double* __restrict__ acc, int first) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= len) return;
  double t = first ? 0.0 : acc[j];
  for (int64_t c = 0; c < nchunks; ++c) t += (double)part[c * len + j];
  acc[j] = t;
}

// single block: acc[0] (+)= sum v[0..n)  (thread t takes t, t + 256, ...; fixed tree afterwards)
static __global__ __launch_bounds__(256) void sum_f64_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ acc, int first) {
  __shared__ double sh[256];
  double t = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) t += v[i];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[0] = (first ? 0.0 : acc[0]) + sh[0];
}

}  // namespace pvs
