// Vocabulary training on gfx950: the device side of ImageEncoderBase.learn (reference:
// pyvisim/encoders/_base_encoder.py:311-342, which fits sklearn KMeans / GaussianMixture / PCA on the stacked
// descriptors).  The heavy passes reuse the encode kernels -- the centroid assignment (f32 MFMA) and the raw residual
// sums of the VLAD aggregate for a Lloyd iteration, the fp64 MFMA posterior + moments for an EM iteration (fisher.hip)
// -- and this file adds the reductions around them.  The host keeps only the K x D sized updates and the loop control
// (python-visual-similarity_amd/pvsim/learn.py).
//
//   Lloyd iteration  sklearn/cluster/_kmeans.py:_kmeans_single_lloyd -> _k_means_lloyd.pyx:lloyd_iter_chunked_dense:
//                    labels = argmin_k (|c_k|^2 - 2 x.c_k); new centre = mean of the members.
//   k-means++        sklearn/cluster/_kmeans.py:_kmeans_plusplus (greedy variant: 2 + log K candidates per step).
//   PCA.fit          sklearn/decomposition/_pca.py:_fit_full, svd_solver "covariance_eigh": eigenvectors of the covariance.
//
// Every sum below is formed in a fixed order (chunk partial sums added in chunk order, fixed reduction trees), so a fit
// is run-to-run identical.
#include <algorithm>

#include "common.hpp"
#include "reduce_kernels.hpp"

namespace pvs {

constexpr int LEARN_CHUNK = 4096;  // descriptors per pseudo-image of the aggregate pass (its LDS sort width)

// member counts (integer atomics: order independent) and the number of labels that changed since the last pass
__global__ __launch_bounds__(256) void learn_label_stats_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ prev,
                                                                int64_t total, int K, unsigned long long* __restrict__ counts,
                                                                unsigned long long* __restrict__ changed) {
  extern __shared__ unsigned int lh[];  // [K]
  for (int k = threadIdx.x; k < K; k += 256) lh[k] = 0u;
  __syncthreads();
  unsigned int diff = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int l = labels[i];
    atomicAdd(&lh[l], 1u);
    if (prev != nullptr && prev[i] != l) ++diff;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    if (lh[k]) atomicAdd(&counts[k], (unsigned long long)lh[k]);
  if (prev == nullptr) diff = 0;
  for (int m = 32; m >= 1; m >>= 1) diff += __shfl_xor(diff, m, 64);
  if ((threadIdx.x & 63) == 0 && diff) atomicAdd(changed, (unsigned long long)diff);
}

__global__ void learn_counts_to_f64_kernel(const unsigned long long* __restrict__ c, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)c[i];
}

// squared distance of every descriptor to its own centre (fp32 per lane, xor tree over the wave) and, per block of
// 64 descriptors, their fp64 sum: the inertia (sklearn/cluster/_k_means_common.pyx:_inertia_dense) and the input of
// the empty-cluster relocation (_relocate_empty_clusters_dense).
__global__ __launch_bounds__(256) void learn_sqdist_kernel(const float* __restrict__ X, int64_t total, int D,
                                                           const int32_t* __restrict__ labels, const float* __restrict__ cent,
                                                           float* __restrict__ sq, double* __restrict__ block_sum) {
  __shared__ float rows[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int j = 0; j < 16; ++j) {
    const int64_t row = r0 + wave * 16 + j;
    float s = 0.f;
    if (row < total) {
      const float* c = cent + (int64_t)labels[row] * D;
      for (int d = lane; d < D; d += 64) {
        const float t = X[row * D + d] - c[d];
        s = fmaf(t, t, s);
      }
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) {
      rows[wave * 16 + j] = s;
      if (row < total && sq != nullptr) sq[row] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int j = 0; j < 64; ++j) t += (double)rows[j];
    block_sum[blockIdx.x] = t;
  }
}

// One Lloyd pass over the descriptors with the centres of `cb`:
//   d_labels[i]            nearest centre (first minimum), as KMeans.predict
//   d_stats[0 .. K*D)      sum over the members of (x - c_k)   (so the new centre is c_k + sum / count)
//   d_stats[K*D .. +K)     member counts
//   d_stats[K*D+K]         inertia  sum_i |x_i - c_label|^2
//   d_stats[K*D+K+1]       number of labels that differ from d_prev_labels (0 when that is null)
int launch_kmeans_step(pvs_ctx* ctx, const pvs_codebook* cb, const float* x, int64_t total, int32_t* d_labels,
                       const int32_t* d_prev_labels, double* d_stats, float* d_sqdist) {
  const int K = cb->K, D = cb->D;
  if (total <= 0) PVS_FAIL(PVS_ERR_INVALID, "k-means needs at least one descriptor");
  if (K > 2048) PVS_FAIL(PVS_ERR_UNSUPPORTED, "K = %d exceeds the device k-means limit (2048)", K);
  const int64_t len = (int64_t)K * D;
  const int64_t rows_per_batch = (int64_t)LEARN_CHUNK * std::max<int64_t>(1, ((int64_t)1 << 30) / (len * 4));
  pvs_norm_params prm{1.0, 2.0, 0.0};
  int first = 1;
  for (int64_t t0 = 0; t0 < total; t0 += rows_per_batch) {
    const int64_t tn = std::min(rows_per_batch, total - t0);
    const int64_t nch = (tn + LEARN_CHUNK - 1) / LEARN_CHUNK;
    const size_t off_b = ((size_t)(nch + 1) * 8 + 255) / 256 * 256;
    char* ws = nullptr;
    PVS_TRY(ws_reserve(ctx, 1, off_b + (size_t)nch * len * 4, reinterpret_cast<void**>(&ws)));
    int64_t* off = reinterpret_cast<int64_t*>(ws);
    float* part = reinterpret_cast<float*>(ws + off_b);
    PVS_TRY(launch_assign(ctx, cb, x + t0 * D, PVS_DESC_F32, tn, D, d_labels + t0));
    hipLaunchKernelGGL(chunk_offsets_kernel, dim3((unsigned)((nch + 256) / 256)), dim3(256), 0, ctx->stream, off, t0, tn,
                       LEARN_CHUNK, nch);
    PVS_TRY(launch_vlad_aggregate(ctx, cb, x, PVS_DESC_F32, D, off, nch, d_labels, prm, part, nullptr, /*raw=*/true));
    hipLaunchKernelGGL(reduce_chunks_kernel<float>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, part, nch, len,
                       d_stats, first);
    PVS_HIP(hipGetLastError());
    first = 0;
  }
  // counts, changed labels, inertia
  const int64_t nblk = (total + 63) / 64;
  const size_t cnt_b = ((size_t)(K + 1) * 8 + 255) / 256 * 256;
  char* ws = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, cnt_b + (size_t)nblk * 8, reinterpret_cast<void**>(&ws)));
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(ws);
  double* bs = reinterpret_cast<double*>(ws + cnt_b);
  PVS_HIP(hipMemsetAsync(cnt, 0, (size_t)(K + 1) * 8, ctx->stream));
  const unsigned hb = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(learn_label_stats_kernel, dim3(hb), dim3(256), (size_t)K * 4, ctx->stream, d_labels, d_prev_labels, total, K,
                     cnt, cnt + K);
  hipLaunchKernelGGL(learn_counts_to_f64_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, cnt, K, d_stats + len);
  hipLaunchKernelGGL(learn_counts_to_f64_kernel, dim3(1), dim3(64), 0, ctx->stream, cnt + K, 1, d_stats + len + K + 1);
  hipLaunchKernelGGL(learn_sqdist_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, x, total, D, d_labels, cb->d_cent,
                     d_sqdist, bs);
  hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(256), 0, ctx->stream, bs, nblk, d_stats + len + K, 1);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ per-label sums
__global__ void learn_square_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] * x[i];
}

// d_out[k][d] = sum over the descriptors labelled k of x_id (square = 0) or x_id**2 squared in fp32 (square = 1): the
// hard-assignment moments a GMM is initialised from (sklearn/mixture/_base.py:_initialize_parameters, init_params="kmeans").
// Same machinery as a Lloyd pass: raw aggregate sums against an all-zero centre table, chunks added in order in fp64.
int launch_label_sums(pvs_ctx* ctx, const float* x, int64_t total, int D, const int32_t* d_labels, int K, int square, double* d_out) {
  if (total <= 0) PVS_FAIL(PVS_ERR_INVALID, "empty input");
  if (K > 2048) PVS_FAIL(PVS_ERR_UNSUPPORTED, "K = %d exceeds the device limit (2048)", K);
  const int64_t len = (int64_t)K * D;
  const int64_t rows_per_batch = (int64_t)LEARN_CHUNK * std::max<int64_t>(1, ((int64_t)1 << 30) / (len * 4));
  float* zero = nullptr;
  PVS_TRY(ws_reserve(ctx, 3, (size_t)len * 4, reinterpret_cast<void**>(&zero)));
  PVS_HIP(hipMemsetAsync(zero, 0, (size_t)len * 4, ctx->stream));
  pvs_codebook cb;
  cb.K = K; cb.D = D; cb.d_cent = zero;
  pvs_norm_params prm{1.0, 2.0, 0.0};
  int first = 1;
  for (int64_t t0 = 0; t0 < total; t0 += rows_per_batch) {
    const int64_t tn = std::min(rows_per_batch, total - t0);
    const int64_t nch = (tn + LEARN_CHUNK - 1) / LEARN_CHUNK;
    const size_t off_b = ((size_t)(nch + 1) * 8 + 255) / 256 * 256;
    char* ws = nullptr;
    PVS_TRY(ws_reserve(ctx, 1, off_b + (size_t)nch * len * 4, reinterpret_cast<void**>(&ws)));
    int64_t* off = reinterpret_cast<int64_t*>(ws);
    float* part = reinterpret_cast<float*>(ws + off_b);
    const float* xb = x + t0 * D;   // the batch's rows; offsets and labels below are relative to it
    if (square) {
      float* sq = nullptr;
      PVS_TRY(ws_reserve(ctx, 4, (size_t)tn * D * 4, reinterpret_cast<void**>(&sq)));
      hipLaunchKernelGGL(learn_square_kernel, dim3(4096), dim3(256), 0, ctx->stream, xb, tn * D, sq);
      xb = sq;
    }
    hipLaunchKernelGGL(chunk_offsets_kernel, dim3((unsigned)((nch + 256) / 256)), dim3(256), 0, ctx->stream, off, (int64_t)0, tn,
                       LEARN_CHUNK, nch);
    PVS_TRY(launch_vlad_aggregate(ctx, &cb, xb, PVS_DESC_F32, D, off, nch, d_labels + t0, prm, part, nullptr, /*raw=*/true));
    hipLaunchKernelGGL(reduce_chunks_kernel<float>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, part, nch, len, d_out, first);
    PVS_HIP(hipGetLastError());
    first = 0;
  }
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ Gram matrix (PCA.fit)
// d_out = [sum_i x_i (D) | sum_i x_i x_i^T (D*D)]  in fp64.  Block = 64x64 tile of the Gram matrix for one chunk of rows
// (upper triangle of tiles only; the host mirrors), 256 threads x (4x4); chunk partials are added in chunk order.
constexpr int GRAM_ROWS = 8192;

__global__ __launch_bounds__(256) void learn_gram_kernel(const float* __restrict__ X, int64_t total, int D, int ntile,
                                                         double* __restrict__ part /*[chunk][ntile*ntile][64*64]*/,
                                                         double* __restrict__ colsum /*[chunk][D]*/) {
  __shared__ double la[32][65], lb[32][65];
  const int ti = blockIdx.x / ntile, tj = blockIdx.x % ntile;
  if (tj < ti) return;
  const int64_t chunk = blockIdx.y;
  const int64_t r0 = chunk * GRAM_ROWS, r1 = r0 + GRAM_ROWS < total ? r0 + GRAM_ROWS : total;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4] = {};
  double cs = 0.0;  // threads 0..63 of diagonal tiles: column sums of tile ti
  for (int64_t r = r0; r < r1; r += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
      const int rr = idx >> 6, c = idx & 63;
      const bool in = r + rr < r1;
      const int da = ti * 64 + c, db = tj * 64 + c;
      la[rr][c] = (in && da < D) ? (double)X[(r + rr) * D + da] : 0.0;
      lb[rr][c] = (in && db < D) ? (double)X[(r + rr) * D + db] : 0.0;
    }
    __syncthreads();
    if (ti == tj && threadIdx.x < 64) {
      for (int rr = 0; rr < 32; ++rr) cs += la[rr][threadIdx.x];
    }
    for (int rr = 0; rr < 32; ++rr) {
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = la[rr][ty * 4 + q];
        b[q] = lb[rr][tx * 4 + q];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
    }
  }
  double* o = part + ((int64_t)chunk * ntile * ntile + blockIdx.x) * 4096;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) o[(ty * 4 + p) * 64 + tx * 4 + q] = acc[p][q];
  if (ti == tj && threadIdx.x < 64 && ti * 64 + (int)threadIdx.x < D) colsum[chunk * D + ti * 64 + threadIdx.x] = cs;
}

__global__ void learn_gram_scatter_kernel(const double* __restrict__ tiles /*[ntile*ntile][4096]*/, int D, int ntile,
                                          double* __restrict__ gram) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * D) return;
  const int i = (int)(idx / D), j = (int)(idx % D);
  const int a = i <= j ? i : j, b = i <= j ? j : i;   // upper-triangle tile holds (a, b)
  gram[idx] = tiles[((int64_t)(a / 64) * ntile + b / 64) * 4096 + (a % 64) * 64 + (b % 64)];
}

int launch_gram(pvs_ctx* ctx, const float* x, int64_t total, int D, double* d_out) {
  if (total <= 0 || D <= 0) PVS_FAIL(PVS_ERR_INVALID, "the Gram matrix needs at least one row");
  const int ntile = (D + 63) / 64;
  const int64_t tl = (int64_t)ntile * ntile * 4096;
  const int64_t nchunk_all = (total + GRAM_ROWS - 1) / GRAM_ROWS;
  const int64_t per_batch = std::max<int64_t>(1, ((int64_t)1 << 30) / (tl * 8));
  double* acc = nullptr;
  PVS_TRY(ws_reserve(ctx, 4, (size_t)tl * 8, reinterpret_cast<void**>(&acc)));
  int first = 1;
  for (int64_t c0 = 0; c0 < nchunk_all; c0 += per_batch) {
    const int64_t nc = std::min(per_batch, nchunk_all - c0);
    char* ws = nullptr;
    PVS_TRY(ws_reserve(ctx, 1, (size_t)nc * (tl + D) * 8, reinterpret_cast<void**>(&ws)));
    double* part = reinterpret_cast<double*>(ws);
    double* cs = part + nc * tl;
    const int64_t r0 = c0 * GRAM_ROWS;
    const int64_t rows = std::min<int64_t>(total - r0, nc * GRAM_ROWS);
    hipLaunchKernelGGL(learn_gram_kernel, dim3((unsigned)(ntile * ntile), (unsigned)nc), dim3(256), 0, ctx->stream, x + r0 * D, rows, D,
                       ntile, part, cs);
    hipLaunchKernelGGL(reduce_chunks_kernel<double>, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, ctx->stream, part, nc, tl, acc, first);
    hipLaunchKernelGGL(reduce_chunks_kernel<double>, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, cs, nc, (int64_t)D, d_out,
                       first);
    PVS_HIP(hipGetLastError());
    first = 0;
  }
  hipLaunchKernelGGL(learn_gram_scatter_kernel, dim3((unsigned)(((int64_t)D * D + 255) / 256)), dim3(256), 0, ctx->stream, acc, D, ntile,
                     d_out + D);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ k-means++ seeding
// For up to 8 candidate centres: d_dist[j][i] = |x_i - cand_j|^2 (fp32) and d_pot[j] = sum_i min(d_mind[i], d_dist[j][i])
// -- the potential the seeding would have if candidate j were taken (sklearn _kmeans_plusplus: "best candidate").
// d_mind null = no centre chosen yet (potential = plain sum of distances).
constexpr int SEED_MAX = 8;

constexpr int SEED_ROWS = 256;   // descriptors per block (8 rounds of 32): at 1024 a pass over 5e5 rows was 512 blocks, two per CU, each a chain of 32 dependent loads (1.8 TB/s)

// Eight lanes share a descriptor (each takes dims 4 l, 4 l + 32, ... as float4: a row's lanes read 128 contiguous bytes per
// step), every lane keeps its partial |x - cand_j|^2 for all candidates in registers, and one 3-step butterfly per candidate
// folds the eight lanes -- 3 cross-lane steps per descriptor and candidate instead of 6 per wave-wide reduction.
__global__ __launch_bounds__(256) void learn_seed_kernel(const float* __restrict__ X, int64_t total, int D, const float* __restrict__ cand,
                                                         int nc, const float* __restrict__ mind, float* __restrict__ dist,
                                                         double* __restrict__ block_pot /*[SEED_MAX][nblk]*/) {
  extern __shared__ float sc[];  // [nc][D] candidates, then [32][SEED_MAX] row results
  float* res = sc + ((nc * D + 1) & ~1);     // 8-B aligned: it holds doubles at the end
  for (int i = threadIdx.x; i < nc * D; i += 256) sc[i] = cand[i];
  __syncthreads();
  const int l8 = threadIdx.x & 7, grp = threadIdx.x >> 3;   // 32 descriptors per round
  const bool vec = (D % 4 == 0) && (reinterpret_cast<uintptr_t>(X) % 16 == 0);
  // every 8-lane group keeps the fp64 potentials of ITS rows (one row per round) in registers: no barrier and no serial sum
  // per round (the loop was bound by them: 156 -> see DESIGN 6b); the 32 group sums are added in group order at the end
  double gp[SEED_MAX];
#pragma unroll
  for (int c = 0; c < SEED_MAX; ++c) gp[c] = 0.0;
  for (int round = 0; round < SEED_ROWS / 32; ++round) {
    const int64_t r0 = (int64_t)blockIdx.x * SEED_ROWS + round * 32;
    if (r0 >= total) break;
    const int64_t row = r0 + grp;
    float s[SEED_MAX];
#pragma unroll
    for (int c = 0; c < SEED_MAX; ++c) s[c] = 0.f;
    if (row < total) {
      if (vec) {
        for (int d = 4 * l8; d < D; d += 32) {
          const float4 xv = *reinterpret_cast<const float4*>(X + row * D + d);
#pragma unroll
          for (int c = 0; c < SEED_MAX; ++c)
            if (c < nc) {
              const float4 cv = *reinterpret_cast<const float4*>(sc + c * D + d);
              float t = xv.x - cv.x; s[c] = fmaf(t, t, s[c]);
              t = xv.y - cv.y; s[c] = fmaf(t, t, s[c]);
              t = xv.z - cv.z; s[c] = fmaf(t, t, s[c]);
              t = xv.w - cv.w; s[c] = fmaf(t, t, s[c]);
            }
        }
      } else {
        for (int d = l8; d < D; d += 8) {
          const float xv = X[row * D + d];
#pragma unroll
          for (int c = 0; c < SEED_MAX; ++c)
            if (c < nc) {
              const float t = xv - sc[c * D + d];
              s[c] = fmaf(t, t, s[c]);
            }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < SEED_MAX; ++c) {
      s[c] += __shfl_xor(s[c], 1, 64);
      s[c] += __shfl_xor(s[c], 2, 64);
      s[c] += __shfl_xor(s[c], 4, 64);
    }
    if (l8 == 0 && row < total) {
      const float md = mind != nullptr ? mind[row] : INFINITY;
#pragma unroll
      for (int c = 0; c < SEED_MAX; ++c)
        if (c < nc) {
          dist[(int64_t)c * total + row] = s[c];
          gp[c] += (double)fminf(md, s[c]);
        }
    }
  }
  double* resd = reinterpret_cast<double*>(res);     // [32 groups][SEED_MAX]
  if (l8 == 0) {
#pragma unroll
    for (int c = 0; c < SEED_MAX; ++c) resd[grp * SEED_MAX + c] = gp[c];
  }
  __syncthreads();
  if (threadIdx.x < SEED_MAX) {
    double pot = 0.0;
    for (int jj = 0; jj < 32; ++jj) pot += resd[jj * SEED_MAX + threadIdx.x];
    block_pot[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = pot;
  }
}

// block c: d_pot[c] = sum of block_pot[c][0..nblk)  (strided partial sums, fixed tree)
__global__ __launch_bounds__(256) void learn_seed_reduce_kernel(const double* __restrict__ block_pot, int64_t nblk, double* __restrict__ pot) {
  __shared__ double sh[256];
  const double* v = block_pot + (int64_t)blockIdx.x * nblk;
  double t = 0.0;
  for (int64_t i = threadIdx.x; i < nblk; i += 256) t += v[i];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) pot[blockIdx.x] = sh[0];
}

int launch_seed_distances(pvs_ctx* ctx, const float* x, int64_t total, int D, const float* d_cand, int n_cand,
                          const float* d_mind, float* d_dist, double* d_pot) {
  if (n_cand < 1 || n_cand > SEED_MAX) PVS_FAIL(PVS_ERR_INVALID, "1..%d seeding candidates per call (got %d)", SEED_MAX, n_cand);
  if (total <= 0) PVS_FAIL(PVS_ERR_INVALID, "seeding needs at least one descriptor");
  const size_t lds = ((size_t)n_cand * D + 2 + 64 * SEED_MAX) * 4;
  if (lds > 64 * 1024) PVS_FAIL(PVS_ERR_UNSUPPORTED, "descriptor dimension %d too large for the seeding kernel", D);
  const int64_t nblk = (total + SEED_ROWS - 1) / SEED_ROWS;
  double* bp = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, (size_t)nblk * SEED_MAX * 8, reinterpret_cast<void**>(&bp)));
  hipLaunchKernelGGL(learn_seed_kernel, dim3((unsigned)nblk), dim3(256), lds, ctx->stream, x, total, D, d_cand, n_cand, d_mind, d_dist, bp);
  hipLaunchKernelGGL(learn_seed_reduce_kernel, dim3(SEED_MAX), dim3(256), 0, ctx->stream, bp, nblk, d_pot);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// Candidate draw of one seeding step, on the device: candidate c falls into block blk[c] (chosen by the host from the
// per-block sums) at the first position whose running fp64 sum of d_mind, started at base[c], reaches target[c]
// (= searchsorted(cumsum(mind), r), sklearn/cluster/_kmeans.py:_kmeans_plusplus); its row is copied to cand[c].
// With `u` (the device-side run, pvs_kmeanspp_run_dev) the block of the draw is found here as well: target = u[c] * pot and
// searchsorted(cumsum(block_sums), target) with the running fp64 sum in block order -- the numbers pvsim/learn.py:_draw_candidates
// forms on the host for the stepwise call.
__global__ __launch_bounds__(64) void learn_pick_kernel(const float* __restrict__ X, int64_t total, int D, const float* __restrict__ mind,
                                                        const int64_t* __restrict__ blk, const double* __restrict__ base,
                                                        const double* __restrict__ target, int64_t* __restrict__ out_idx,
                                                        float* __restrict__ cand, const double* __restrict__ u = nullptr,
                                                        const double* __restrict__ pot = nullptr,
                                                        const double* __restrict__ block_sums = nullptr, int64_t nblk = 0) {
  __shared__ float vals[LEARN_CHUNK];
  __shared__ double part[64];
  __shared__ int64_t s_idx;
  __shared__ int64_t s_blk;
  __shared__ double s_base, s_tgt;
  const int c = blockIdx.x, t = threadIdx.x;
  if (t == 0) {
    if (u != nullptr) {
      const double r = u[c] * pot[0];
      double cum = 0.0, before = 0.0;
      int64_t b = nblk - 1;
      for (int64_t i = 0; i < nblk; ++i) {
        before = cum;
        cum += block_sums[i];
        if (cum >= r) { b = i; break; }
      }
      s_blk = b; s_base = b > 0 ? before : 0.0; s_tgt = r;
    } else {
      s_blk = blk[c]; s_base = base[c]; s_tgt = target[c];
    }
  }
  __syncthreads();
  const int64_t lo = s_blk * LEARN_CHUNK;
  const int cnt = (int)min((int64_t)LEARN_CHUNK, total - lo);
  for (int i = t; i < LEARN_CHUNK; i += 64) vals[i] = i < cnt ? mind[lo + i] : 0.f;   // coalesced; zeros past the end
  __syncthreads();
  // the block's running sum in a fixed order: 64 runs of 64 consecutive entries, then the runs in order
  constexpr int RUN = LEARN_CHUNK / 64;
  double p = 0.0;
  for (int i = 0; i < RUN; ++i) p += (double)vals[t * RUN + i];
  part[t] = p;
  __syncthreads();
  if (t == 0) {
    double run = s_base;
    int sub = 63;
    for (int q = 0; q < 64; ++q) {
      if (run + part[q] >= s_tgt) { sub = q; break; }
      run += part[q];
    }
    int pos = min(sub * RUN + RUN - 1, cnt - 1);
    for (int i = 0; i < RUN; ++i) {
      run += (double)vals[sub * RUN + i];
      if (run >= s_tgt) { pos = sub * RUN + i; break; }
    }
    pos = min(pos, cnt - 1);
    s_idx = lo + pos;
    out_idx[c] = lo + pos;
  }
  __syncthreads();
  const int64_t row = s_idx;
  for (int d = t; d < D; d += 64) cand[(int64_t)c * D + d] = X[row * D + d];
}

int launch_seed_pick(pvs_ctx* ctx, const float* x, int64_t total, int D, const float* d_mind, const int64_t* d_blk,
                     const double* d_base, const double* d_target, int n_cand, int64_t* d_idx, float* d_cand) {
  hipLaunchKernelGGL(learn_pick_kernel, dim3((unsigned)n_cand), dim3(64), 0, ctx->stream, x, total, D, d_mind, d_blk, d_base, d_target,
                     d_idx, d_cand);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// d_mind = min(d_mind, d_dist) and the fp64 sums of d_mind over blocks of LEARN_CHUNK entries (the host samples the next
// candidates from these: block by cumulative sum, then the position inside the block)
// With `block_pot` (the device-side run) every block first forms the candidates' potentials from the distance pass's block sums --
// the reduction of learn_seed_reduce_kernel, addition for addition -- and takes the first minimum (np.argmin); block 0 records
// the winner's descriptor index and potential.
__global__ __launch_bounds__(256) void learn_min_update_kernel(float* __restrict__ mind, const float* __restrict__ dist, int64_t total,
                                                               double* __restrict__ block_sums, const double* __restrict__ block_pot = nullptr,
                                                               int64_t nblk_seed = 0, int trials = 0, const int64_t* __restrict__ idx = nullptr,
                                                               double* __restrict__ pot = nullptr, int64_t* __restrict__ indices = nullptr, int c = 0) {
  __shared__ double sh[256];
  __shared__ double s_pots[8];
  if (block_pot != nullptr) {
    for (int j = 0; j < trials; ++j) {
      const double* v = block_pot + (int64_t)j * nblk_seed;
      double t = 0.0;
      for (int64_t i = threadIdx.x; i < nblk_seed; i += 256) t += v[i];
      sh[threadIdx.x] = t;
      __syncthreads();
      for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
        __syncthreads();
      }
      if (threadIdx.x == 0) s_pots[j] = sh[0];
      __syncthreads();
    }
    int best = 0;
    for (int j = 1; j < trials; ++j)
      if (s_pots[j] < s_pots[best]) best = j;
    dist += (int64_t)best * total;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      pot[0] = s_pots[best];
      indices[c] = idx[best];
    }
  }
  const int64_t b0 = (int64_t)blockIdx.x * LEARN_CHUNK;
  double t = 0.0;
  for (int i = threadIdx.x; i < LEARN_CHUNK; i += 256) {
    const int64_t r = b0 + i;
    if (r < total) {
      const float v = dist != nullptr ? fminf(mind[r], dist[r]) : mind[r];
      mind[r] = v;
      t += (double)v;
    }
  }
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}

int launch_min_update(pvs_ctx* ctx, float* d_mind, const float* d_dist, int64_t total, double* d_block_sums) {
  if (total <= 0) return PVS_OK;
  const int64_t nblk = (total + LEARN_CHUNK - 1) / LEARN_CHUNK;
  hipLaunchKernelGGL(learn_min_update_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_mind, d_dist, total, d_block_sums,
                     static_cast<const double*>(nullptr), (int64_t)0, 0, static_cast<const int64_t*>(nullptr), static_cast<double*>(nullptr),
                     static_cast<int64_t*>(nullptr), 0);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ---- greedy k-means++ without host round trips (pvs_kmeanspp_run_dev): three launches per step -- draw (learn_pick_kernel finds the
// block of each target itself), distances (learn_seed_kernel), choice + running-minimum update (learn_min_update_kernel)
__global__ void learn_copy_row_kernel(const float* __restrict__ X, int D, const int64_t* __restrict__ indices, int c, float* __restrict__ cand) {
  const int64_t row = indices[c];
  for (int d = threadIdx.x; d < D; d += blockDim.x) cand[d] = X[row * D + d];
}

int launch_kmeanspp_run(pvs_ctx* ctx, const float* x, int64_t total, int D, int n_clusters, int trials, const double* d_uniform,
                        float* d_mind, float* d_dist, float* d_cand, double* d_block_sums, char* d_small, int64_t* d_indices) {
  // d_small: pots[8] | pot | idx[8]
  double* d_pots = reinterpret_cast<double*>(d_small);
  double* d_pot = d_pots + 8;
  int64_t* d_idx = reinterpret_cast<int64_t*>(d_pot + 1);
  const int64_t nblk = (total + LEARN_CHUNK - 1) / LEARN_CHUNK, nblk_seed = (total + SEED_ROWS - 1) / SEED_ROWS;
  const size_t lds = ((size_t)trials * D + 2 + 64 * SEED_MAX) * 4;
  if (lds > 64 * 1024) PVS_FAIL(PVS_ERR_UNSUPPORTED, "descriptor dimension %d too large for the seeding kernel", D);
  double* bp = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, (size_t)nblk_seed * SEED_MAX * 8, reinterpret_cast<void**>(&bp)));
  // first centre: its distances are the running minima
  hipLaunchKernelGGL(learn_copy_row_kernel, dim3(1), dim3(64), 0, ctx->stream, x, D, d_indices, 0, d_cand);
  PVS_TRY(launch_seed_distances(ctx, x, total, D, d_cand, 1, nullptr, d_dist, d_pots));
  PVS_HIP(hipMemcpyAsync(d_pot, d_pots, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  PVS_TRY(launch_min_update(ctx, d_mind, d_dist, total, d_block_sums));
  for (int c = 1; c < n_clusters; ++c) {
    hipLaunchKernelGGL(learn_pick_kernel, dim3((unsigned)trials), dim3(64), 0, ctx->stream, x, total, D, d_mind, static_cast<const int64_t*>(nullptr),
                       static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), d_idx, d_cand,
                       d_uniform + (size_t)(c - 1) * trials, d_pot, d_block_sums, nblk);
    hipLaunchKernelGGL(learn_seed_kernel, dim3((unsigned)nblk_seed), dim3(256), lds, ctx->stream, x, total, D, d_cand, trials, d_mind, d_dist, bp);
    hipLaunchKernelGGL(learn_min_update_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_mind, d_dist, total, d_block_sums, bp,
                       nblk_seed, trials, d_idx, d_pot, d_indices, c);
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

}  // namespace pvs
