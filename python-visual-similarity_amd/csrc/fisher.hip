// Fisher-vector encode on gfx950 (K4 posterior, K5 moments + gradients + normalisation) and the PCA prologue.
//
// Reference semantics (paths relative to the reference root):
//   K4  pyvisim/encoders/fisher_vector.py:99 -> sklearn GaussianMixture.predict_proba, 'diag'
//       (sklearn/mixture/_gaussian_mixture.py:495-512, sklearn/mixture/_base.py:513-538):
//         logp_ik = const_k + x_i.(mu_k prec_k) - 0.5 (x_i**2).prec_k ;  gamma = exp(logp - logsumexp_k)
//       fp64 tables, X**2 squared in X's dtype (fp32) first.
//   K5  fisher_vector.py:102-129: s0 = mean_i gamma, s1 = gamma^T X / n, s2 = gamma^T X**2 / n; gradients wrt
//       (pi, mu, sigma^2) with the analytic diagonal normalisation; sign|v|^p; GLOBAL L2 + eps.
//       Output layout [d_pi (K) | d_mu (K*D, k-major) | d_sigma (K*D)].
//   PCA vlad.py:89-90 / fisher_vector.py:91-92 -> sklearn PCA.transform: X @ comp^T - mean @ comp^T (fp32).
//
// The reference computes this path in fp64 and returns fp64; the covariance floor (1e-6 -> precision 1e6)
// makes the log-density a sum of large cancelling terms, so the device path keeps fp64 arithmetic
// (MI355X runs fp64 at half the fp32 rate -- cheaper than losing 3 digits).  K <= 256 runs on v_mfma_f64_16x16x4
// (posterior as one GEMM over [x | x**2], moments as gamma^T [x | x**2]); larger K uses the vector-FMA kernels.
#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "desc_load.hpp"
#include "reduce_kernels.hpp"

namespace pvs {

// ------------------------------------------------------------------------------------ materialise RootSIFT
template <int KIND>
__global__ __launch_bounds__(256) void materialise_kernel(const void* __restrict__ X, int64_t total, int D,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += load1<KIND>(X, row, D, d);
  s = wave_sum_xor(s, 64);
  for (int d = lane; d < D; d += 64) {
    const float v = load1<KIND>(X, row, D, d);
    out[row * D + d] = DescTraits<KIND>::rootsift ? RootsiftRow<KIND>(s)(v) : v;
  }
}

static int materialise_f32(pvs_ctx* ctx, const void*& d_desc, int& kind, int64_t total, int D) {
  constexpr int ws_slot = 4;  // never aliases the host-API staging slot 0
  if (kind == PVS_DESC_F32 || total <= 0) return PVS_OK;
  float* buf = nullptr;
  PVS_TRY(ws_reserve(ctx, ws_slot, (size_t)total * D * sizeof(float), reinterpret_cast<void**>(&buf)));
  const dim3 grid((unsigned)((total + 3) / 4));
  if (kind == PVS_DESC_U8_ROOTSIFT)
    hipLaunchKernelGGL(materialise_kernel<PVS_DESC_U8_ROOTSIFT>, grid, dim3(256), 0, ctx->stream, d_desc, total, D, buf);
  else
    hipLaunchKernelGGL(materialise_kernel<PVS_DESC_F32_ROOTSIFT>, grid, dim3(256), 0, ctx->stream, d_desc, total, D, buf);
  PVS_HIP(hipGetLastError());
  d_desc = buf;
  kind = PVS_DESC_F32;
  return PVS_OK;
}

// plain fp32 rows of any descriptor kind (RootSIFT applied) into a caller buffer: the training entry points work on these
int launch_materialise(pvs_ctx* ctx, const void* d_desc, int kind, int64_t total, int D, float* d_out) {
  if (total <= 0) return PVS_OK;
  const dim3 grid((unsigned)((total + 3) / 4));
  switch (kind) {
    case PVS_DESC_F32: hipLaunchKernelGGL(materialise_kernel<PVS_DESC_F32>, grid, dim3(256), 0, ctx->stream, d_desc, total, D, d_out); break;
    case PVS_DESC_F32_ROOTSIFT: hipLaunchKernelGGL(materialise_kernel<PVS_DESC_F32_ROOTSIFT>, grid, dim3(256), 0, ctx->stream, d_desc, total, D, d_out); break;
    case PVS_DESC_U8_ROOTSIFT: hipLaunchKernelGGL(materialise_kernel<PVS_DESC_U8_ROOTSIFT>, grid, dim3(256), 0, ctx->stream, d_desc, total, D, d_out); break;
    default: PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ PCA.transform
// out[i][c] = sum_d x[i][d] comp[c][d] - off[c]   (fp32).  64 rows x 64 comps per block, 4x4 per thread.
__global__ __launch_bounds__(256) void pca_kernel(const float* __restrict__ X, int64_t total, int Din,
                                                  const float* __restrict__ comp, const float* __restrict__ off,
                                                  int C, float* __restrict__ out) {
  constexpr int TS = 64, KS = 16;
  __shared__ float sx[KS][TS + 1];
  __shared__ float sc[KS][TS + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t r0 = (int64_t)blockIdx.y * TS;
  const int c0 = blockIdx.x * TS;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < Din; k0 += KS) {
    for (int idx = threadIdx.x; idx < TS * KS; idx += 256) {
      const int r = idx / KS, c = idx % KS;
      const int k = k0 + c;
      sx[c][r] = (r0 + r < total && k < Din) ? X[(r0 + r) * Din + k] : 0.f;
      sc[c][r] = (c0 + r < C && k < Din) ? comp[(int64_t)(c0 + r) * Din + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = sx[c][ty * 4 + u]; bv[u] = sc[c][tx * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(av[u], bv[v], acc[u][v]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int64_t r = r0 + ty * 4 + u;
      const int c = c0 + tx * 4 + v;
      if (r < total && c < C) out[r * C + c] = acc[u][v] - off[c];
    }
}

int launch_pca(pvs_ctx* ctx, const pvs_pca* p, const void* d_desc, int kind, int64_t total, float* d_out) {
  if (total <= 0) return PVS_OK;
  const void* x = d_desc;
  int k = kind;
  PVS_TRY(materialise_f32(ctx, x, k, total, p->Din));
  ScopedTimer tm(ctx, T_MISC);
  dim3 grid((unsigned)((p->C + 63) / 64), (unsigned)((total + 63) / 64));
  hipLaunchKernelGGL(pca_kernel, grid, dim3(256), 0, ctx->stream, static_cast<const float*>(x), total, p->Din,
                     p->d_comp, p->d_off, p->C, d_out);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ K4 posterior
// Block: PR descriptors x all K clusters.  Thread t owns clusters t, t+256, ...; the descriptor slab and its
// fp32 squares sit in LDS as fp64 pairs (broadcast reads); the (mu*prec, prec) tables are streamed from L2
// in [d][k] order so that consecutive threads read consecutive addresses.
constexpr int POST_ROWS = 32;
constexpr int POST_DK = 64;      // dims staged per step
constexpr int POST_KMAX = 1024;  // clusters per block pass = 256 threads x 4

struct PostArgs {
  const float* X;
  int64_t total;
  int D, ld, K;
  const double* mupT;   // [D][K]
  const double* precT;  // [D][K]
  const double* cst;    // [K]
  double* resp;         // [total][K]
  double* lse;          // [total] or null (training)
};

__global__ __launch_bounds__(256) void gmm_posterior_kernel(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* xs = reinterpret_cast<double2*>(smem);                    // [POST_DK][POST_ROWS]  (x, x*x)
  double* lp = reinterpret_cast<double*>(xs + POST_DK * POST_ROWS);  // [POST_ROWS][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * POST_ROWS;
  const int nkb = (a.K + 255) / 256;

  for (int kb = 0; kb < nkb; ++kb) {
    const int k = kb * 256 + tid;
    const bool kv = k < a.K;
    double acc[POST_ROWS];
#pragma unroll
    for (int r = 0; r < POST_ROWS; ++r) acc[r] = 0.0;
    for (int d0 = 0; d0 < a.D; d0 += POST_DK) {
      __syncthreads();
      for (int idx = tid; idx < POST_DK * POST_ROWS; idx += 256) {
        const int r = idx / POST_DK, dd = idx % POST_DK;  // consecutive threads -> consecutive dims of a row
        const int d = d0 + dd;
        float x = 0.f;
        if (r0 + r < a.total && d < a.D) x = a.X[(r0 + r) * a.ld + d];
        const float x2 = x * x;                           // squared in fp32, as X**2 on an fp32 array
        xs[dd * POST_ROWS + r] = make_double2((double)x, (double)x2);
      }
      __syncthreads();
      const int dn = min(POST_DK, a.D - d0);
      if (kv) {
        for (int dd = 0; dd < dn; ++dd) {
          const double m = a.mupT[(int64_t)(d0 + dd) * a.K + k];
          const double p = -0.5 * a.precT[(int64_t)(d0 + dd) * a.K + k];
#pragma unroll
          for (int r = 0; r < POST_ROWS; ++r) {
            const double2 v = xs[dd * POST_ROWS + r];
            acc[r] = fma(v.x, m, fma(v.y, p, acc[r]));
          }
        }
      }
    }
    if (kv) {
      const double c = a.cst[k];
#pragma unroll
      for (int r = 0; r < POST_ROWS; ++r) lp[r * a.K + k] = acc[r] + c;
    }
  }
  __syncthreads();
  // softmax over k, one wave per row (scipy logsumexp: max, log-sum-exp, subtract, exp)
  for (int r = wave; r < POST_ROWS; r += 4) {
    if (r0 + r >= a.total) continue;
    double mx = -INFINITY;
    for (int k = lane; k < a.K; k += 64) mx = fmax(mx, lp[r * a.K + k]);
    for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
    double s = 0.0;
    for (int k = lane; k < a.K; k += 64) s += exp(lp[r * a.K + k] - mx);
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    const double lse = mx + log(s);
    for (int k = lane; k < a.K; k += 64) a.resp[(r0 + r) * a.K + k] = exp(lp[r * a.K + k] - lse);
    if (a.lse != nullptr && lane == 0) a.lse[r0 + r] = lse;
  }
}

__global__ void transpose_tables_kernel(const double* __restrict__ prec, const double* __restrict__ mup, int K, int D,
                                        double* __restrict__ precT, double* __restrict__ mupT) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)K * D) return;
  const int k = (int)(i / D), d = (int)(i % D);
  precT[(int64_t)d * K + k] = prec[i];
  mupT[(int64_t)d * K + k] = mup[i];
}

static int posterior_on(pvs_ctx* ctx, const pvs_gmm* g, const float* x, int ld, int64_t total, double* d_resp,
                        double* tabT, double* d_lse = nullptr) {
  if (g->K > POST_KMAX) PVS_FAIL(PVS_ERR_UNSUPPORTED, "GMM with K = %d components exceeds the kernel limit (%d)", g->K,
                                POST_KMAX);
  const int64_t kd = (int64_t)g->K * g->D;
  double* precT = tabT;
  double* mupT = tabT + kd;
  hipLaunchKernelGGL(transpose_tables_kernel, dim3((unsigned)((kd + 255) / 256)), dim3(256), 0, ctx->stream, g->d_prec,
                     g->d_mup, g->K, g->D, precT, mupT);
  PostArgs a{x, total, g->D, ld, g->K, mupT, precT, g->d_const, d_resp, d_lse};
  const size_t lds = (size_t)POST_DK * POST_ROWS * sizeof(double2) + (size_t)POST_ROWS * g->K * sizeof(double);
  PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_posterior_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ScopedTimer tm(ctx, T_FPOST);
  hipLaunchKernelGGL(gmm_posterior_kernel, dim3((unsigned)((total + POST_ROWS - 1) / POST_ROWS)), dim3(256), lds,
                     ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ fp64 MFMA building block
// v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15] (one f64 per lane each),
// C/D: 4 f64 per lane, col = lane&15, row = (lane>>4) + 4*reg.  Operand chunks sit in LDS as [row][KC+1] doubles
// (the +1 makes the 16 rows of a fragment read hit distinct bank pairs).
typedef double f64x4_t __attribute__((ext_vector_type(4)));
constexpr int F64_KC = 16, F64_KCP = F64_KC + 1;

template <int MI, int NI, int KC = F64_KC>
__device__ __forceinline__ void f64_chunk_mma(const double* la, const double* lb, int arow0, int bcol0, int lane,
                                              f64x4_t (&acc)[MI][NI]) {
  constexpr int KCP = KC + 1;
  const int r = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KC / 4; ++ks) {
    double a[MI], b[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = la[(arow0 + 16 * mi + r) * KCP + 4 * ks + kk];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = lb[(bcol0 + 16 * ni + r) * KCP + 4 * ks + kk];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------ K4 posterior on fp64 MFMA
// logp[i][k] = const_k + [x_i | x_i**2] . [mu prec | -0.5 prec]_k      (one fp64 GEMM with inner dimension 2D)
// Block = 32 descriptors x 256 cluster columns (4 waves x 64); K <= 256 (padded clusters get const = -inf).
constexpr int PM_ROWS = 128, PM_COLS = 256, PM_THREADS = 512;

struct PostMArgs {
  const float* X;
  int64_t total;
  int D, ld, K;
  const double* tab2;  // [K][2D]: mu*prec | -0.5*prec
  const double* cst;   // [K]
  double* resp;        // [total][K]
  double* lse;         // [total] log sum_k exp(logp) per descriptor (training: the EM lower bound), or null
};

// 8 waves: wave = (wm, wn), wm = wave >> 2 owns descriptors [64 wm, +64), wn = wave & 3 owns clusters [64 wn, +64)
__global__ __launch_bounds__(PM_THREADS, 2) void gmm_posterior_mfma_kernel(PostMArgs a) {
  // double-buffered operand chunks: the global loads of chunk c+1 are in flight while chunk c runs on the MFMA pipe
  extern __shared__ __attribute__((aligned(16))) char pm_smem[];
  double* const la0 = reinterpret_cast<double*>(pm_smem);            // [2][PM_ROWS * F64_KCP]
  double* const lb0 = la0 + 2 * PM_ROWS * F64_KCP;                   // [2][PM_COLS * F64_KCP]
  double (*red)[4] = reinterpret_cast<double (*)[4]>(lb0 + 2 * PM_COLS * F64_KCP);   // [PM_ROWS][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t r0 = (int64_t)blockIdx.x * PM_ROWS;
  const int kd = 2 * a.D;
  f64x4_t acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f64x4_t{0.0, 0.0, 0.0, 0.0};

  // thread t stages inner index jj = t & 15 of rows (t >> 4) + 32 q: A rows q < 4 (descriptors), B rows q < 8 (clusters)
  const int jj = tid & 15, rr = tid >> 4;
  float xa[PM_ROWS / 32];
  double tb[PM_COLS / 32];
  auto fetch = [&](int k0) {
    const int j = k0 + jj;
#pragma unroll
    for (int q = 0; q < PM_ROWS / 32; ++q) {
      const int r = rr + 32 * q;
      xa[q] = (r0 + r < a.total && j < kd) ? a.X[(r0 + r) * a.ld + (j < a.D ? j : j - a.D)] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < PM_COLS / 32; ++q) {
      const int c = rr + 32 * q;
      tb[q] = (c < a.K && j < kd) ? a.tab2[(int64_t)c * kd + j] : 0.0;
    }
  };
  auto stash = [&](int k0, int buf) {
    const bool sq = k0 + jj >= a.D;   // X**2 squared in fp32 first (sklearn _gaussian_mixture.py:500)
    double* la = la0 + buf * (PM_ROWS * F64_KCP);
    double* lb = lb0 + buf * (PM_COLS * F64_KCP);
#pragma unroll
    for (int q = 0; q < PM_ROWS / 32; ++q) la[(rr + 32 * q) * F64_KCP + jj] = sq ? (double)(xa[q] * xa[q]) : (double)xa[q];
#pragma unroll
    for (int q = 0; q < PM_COLS / 32; ++q) lb[(rr + 32 * q) * F64_KCP + jj] = tb[q];
  };
  fetch(0);
  stash(0, 0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < kd; k0 += F64_KC) {
    const bool more = k0 + F64_KC < kd;
    if (more) fetch(k0 + F64_KC);
    f64_chunk_mma<4, 4>(la0 + buf * (PM_ROWS * F64_KCP), lb0 + buf * (PM_COLS * F64_KCP), wm * 64, wn * 64, lane, acc);
    if (more) stash(k0 + F64_KC, buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- softmax over clusters (scipy logsumexp: max, log-sum-exp, subtract, exp).  This lane holds, for column
  // c = lane&15 of each of the wave's 4 cluster tiles, the 16 descriptor rows 64 wm + 16 mi + 4 reg + (lane>>4).
  const int col = lane & 15, rq = lane >> 4;
  double cst[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int k = wn * 64 + 16 * ni + col;
    cst[ni] = k < a.K ? a.cst[k] : -INFINITY;
  }
  // (no per-lane copy of the 16 row maxima is kept across the phases: 32 registers that pushed the kernel over 256 and into
  // scratch -- the one kernel of the EM chain with a private segment; the maxima are re-read from LDS where they are used)
  double (*red2)[4] = reinterpret_cast<double (*)[4]>(reinterpret_cast<double*>(red) + PM_ROWS * 4);   // [PM_ROWS][4] partial sums
  double* const rmax = reinterpret_cast<double*>(red2) + PM_ROWS * 4;                                  // [PM_ROWS] row maxima
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double m = -INFINITY;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        acc[mi][ni][r] += cst[ni];
        m = fmax(m, acc[mi][ni][r]);
      }
      for (int s = 8; s >= 1; s >>= 1) m = fmax(m, __shfl_xor(m, s, 64));  // over the 16 columns of this row group
      if (col == 0) red[wm * 64 + 16 * mi + 4 * r + rq][wn] = m;
    }
  __syncthreads();
  // e = exp(logp - max) is formed once and kept in the accumulators; gamma = e / sum (within 2 ulp of scipy's
  // exp(logp - logsumexp)); the row's log-sum-exp itself is only needed by training
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 64 + 16 * mi + 4 * r + rq;
      const double m = fmax(fmax(red[row][0], red[row][1]), fmax(red[row][2], red[row][3]));
      double s = 0.0;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        acc[mi][ni][r] = exp(acc[mi][ni][r] - m);
        s += acc[mi][ni][r];
      }
      for (int q = 8; q >= 1; q >>= 1) s += __shfl_xor(s, q, 64);
      if (col == 0) {
        red2[row][wn] = s;
        if (wn == 0) rmax[row] = m;
      }
    }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 64 + 16 * mi + 4 * r + rq;
      const double sum = ((red2[row][0] + red2[row][1]) + red2[row][2]) + red2[row][3];
      const double rs = 1.0 / sum;
      if (r0 + row < a.total) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int k = wn * 64 + 16 * ni + col;
          if (k < a.K) a.resp[(r0 + row) * a.K + k] = acc[mi][ni][r] * rs;
        }
        if (a.lse != nullptr && wn == 0 && col == 0) a.lse[r0 + row] = rmax[row] + log(sum);
      }
    }
}

// ------------------------------------------------------------------------------------ K5 on fp64 MFMA (fused)
// Per image:  S[k][c] = sum_i gamma[i][k] * Z[i][c]  (inner dimension n_i), then the gradients, the power norm and
// this block's share of the global norm -- the raw sums never leave the registers.
// Block = (image, 32 dims), 8 waves: wave wm owns cluster rows [32 wm, +32) (MM_MI = 2 row tiles of 16) and the 64 columns
// Z = [x_d (32) | x_d**2 (32)] of dims d0 + [0, 32), so that the lane holding S1[k][d] (column tile ni) also holds S2[k][d]
// (tile ni + 2).  s0[k] = sum_i gamma[i][k] is summed by thread k from the staged gamma^T chunks (descriptor order).
// Shape, as measured at configs[2] (profiles/r03_fisher_moments_ablation.txt):
//   rounds 1-2  one workgroup of 8 waves x (64 clusters x 64 columns) on 64 dims, chunks of 16, 104 KB LDS, 256 registers:
//               2 waves per SIMD, all in the same phase                                                    19.6 ms, pipe 56 % busy
//   round 3 (a) two workgroups of 4 such waves on 32 dims, chunks of 8, 46 KB each                         19.0 ms
//           (b) 8 waves x (32 clusters x 64 columns): 64 accumulator registers, <= 128 in all, so TWO workgroups = 4 waves
//               per SIMD share a CU                                                                        17.8 ms, pipe 60 % busy
// fp64 vector instructions and fp64 MFMAs run on the same units, so the epilogue (35 fp64 instructions per output) adds to the
// loop whatever the occupancy.  The dim blocks of an image are mapped to ONE XCD, next to each other in its dispatch order, so
// that the image's gamma rows (400 KB at configs[2], read by every dim block) are shared in that XCD's L2 (23.0 GB fetched beyond
// L2 for 6.6 GB of operands; 28.6 GB with 64-dim blocks dealt round-robin).
struct MomMArgs {
  const float* X;
  int D, ld, K;
  const int64_t* offsets;
  const double* resp;   // [total][K]
  const double* w;
  const double* mu;
  const double* cov;
  const double* inv_mu;   // [K][D]  1 / (sqrt(w_k) sqrt(cov_kd))
  const double* inv_sg;   // [K][D]  1 / (sqrt(2) sqrt(w_k) cov_kd)
  double power;
  int norm_mode;
  double norm_p;
  void* out;
  int out_f64;
  double* partial;      // [n_images][dblocks]
  int dblocks;
  double* raw_s;        // RAW: [n_images][K][2D]  sum_i gamma x | sum_i gamma x**2
  double* raw_s0;       // RAW: [n_images][K]      sum_i gamma
  int resp_ld, k0;      // gamma row stride and first cluster of this launch (RAW over a mixture of more than 256 components)
  int fold;             // != 0 (one workgroup per image): the workgroup walks ALL dim blocks of its image and divides by the norm itself
  double eps;           // fold: added to the norm before dividing
  int64_t n_img;        // images (RAW: descriptor chunks) of this launch
};

__device__ __forceinline__ double power_norm64(double v, double p);
__device__ __forceinline__ double norm_term64(double v, int mode, double p);
template <typename T>
__device__ __forceinline__ void store_out(void* out, int f64, int64_t i, T v) {
  if (f64) static_cast<double*>(out)[i] = (double)v;
  else static_cast<float*>(out)[i] = (float)v;
}

constexpr int MM_MI = 2;                                   // 16-cluster row tiles per wave
constexpr int MM_THREADS = 64 * (256 / (16 * MM_MI)), MM_DIMS = 32, MM_KC = 8, MM_KCP = MM_KC + 1;

// The fused epilogue is instantiated per (power, norm) case: with p and the norm order as run-time values every one of
// its 64 unrolled outputs carried an inlined pow() (18k instructions, 110 KB of code -- twice the instruction cache --
// and the kernel ran at 2.3x the time of its own MFMA loop).  PM: 0 p == 1, 1 p == 0.5, 2 any p.  NM: 2 L2, 1 abs-based
// (L1 / inf), 0 general order.  The general cases call out-of-line helpers.
__device__ __attribute__((noinline)) double power_norm64_slow(double v, double p);
__device__ __attribute__((noinline)) double norm_pow64_slow(double av, double p);
template <int PM>
__device__ __forceinline__ double pnorm_t(double v, double p) {
  if constexpr (PM == 0) {
    return v;
  } else if constexpr (PM == 1) {
    const double m = sqrt(fabs(v));
    return v > 0.0 ? m : (v < 0.0 ? -m : (v == 0.0 ? 0.0 * m : v));
  } else {
    return power_norm64_slow(v, p);
  }
}
template <int NM>
__device__ __forceinline__ double nterm_t(double v, double p) {
  if constexpr (NM == 2) return v * v;
  else if constexpr (NM == 1) return fabs(v);
  else return norm_pow64_slow(fabs(v), p);
}

// RAW = leave the sums as they are (one EM M-step's sufficient statistics per descriptor chunk) instead of the Fisher epilogue
template <bool RAW, int PM = 2, int NM = 0, bool OUT64 = true, bool FOLD = false>
__global__ __launch_bounds__(MM_THREADS, 4) void fisher_moments_mfma_kernel(MomMArgs a) {
  // double-buffered chunks (global loads of chunk c+1 in flight during the MFMAs of chunk c)
  extern __shared__ __attribute__((aligned(16))) char mm_smem[];
  double* const la0 = reinterpret_cast<double*>(mm_smem);   // [2][256 * MM_KCP]  gamma^T chunk: [k][i]
  double* const lb0 = la0 + 2 * 256 * MM_KCP;               // [2][64 * MM_KCP]   Z chunk: [c][i], c = (x: 0..31 | x**2: 32..63)
  double* const s0s = lb0 + 2 * 64 * MM_KCP;                // [256]
  double* const red = s0s + 256;                            // [8]
  // a.fold (non-RAW, one workgroup per image; measurement variant, see fisher_batch): this workgroup walks all dim blocks of its
  // image, keeps the norm term of each (added in block order, exactly as fisher_scale_kernel adds the partials) and finally
  // divides its own outputs, in place of the separate scale pass.
  constexpr bool fold = !RAW && FOLD;   // (a.fold says the same at run time: the launcher picks the instantiation)
  // Otherwise workgroup b runs on XCD b % 8 (round-robin dispatch), as that XCD's (b / 8)-th: images are dealt to the XCDs in
  // groups of 8 and an XCD walks the dim blocks of its image before it moves to the next group.
  int img, db_first, db_last;
  if (fold) {
    img = blockIdx.x;
    db_first = 0;
    db_last = a.dblocks;
  } else {
    const int64_t seq = (int64_t)(blockIdx.x >> 3);
    const int64_t im = (seq / a.dblocks) * 8 + (blockIdx.x & 7);
    if (im >= a.n_img) return;
    img = (int)im;
    db_first = (int)(seq % a.dblocks);
    db_last = db_first + 1;
  }
  double norm_total = 0.0;   // thread 0 only
  for (int db = db_first; db < db_last; ++db) {
  // the thread index is re-read behind an opaque barrier in every round: otherwise every per-thread index and address of the
  // body is hoisted out of the dim-block loop and the kernel (128 accumulator registers) spills 76 registers
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wm = tid >> 6;
  const int dblk = db * MM_DIMS, d0 = dblk;
  const int64_t row0 = a.offsets[img];
  const int n = (int)(a.offsets[img + 1] - row0);
  const int K = a.K, D = a.D;
  const int64_t L = (int64_t)K + 2 * (int64_t)K * D;
  f64x4_t acc[MM_MI][4];
#pragma unroll
  for (int mi = 0; mi < MM_MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f64x4_t{0.0, 0.0, 0.0, 0.0};
  double s0 = 0.0;

  // staging: gamma: thread t takes cluster k = t & 255 of the chunk's descriptors (t >> 8) + GS q (consecutive threads ->
  // consecutive clusters); Z: the first 256 threads: dim dd = t & 31 of descriptor t >> 5
  constexpr int GS = MM_THREADS / 256, GQ = MM_KC / GS;
  const int gk = tid & 255, gi = tid >> 8, zd = tid & 31, zi = tid >> 5;
  const bool zt = tid < 32 * MM_KC;
  static_assert(32 * MM_KC <= MM_THREADS && MM_KC % GS == 0 && MM_KC % 4 == 0, "staging layout");
  double gv[GQ];
  float zv = 0.f;
  auto fetch = [&](int i0) {
#pragma unroll
    for (int q = 0; q < GQ; ++q)
      gv[q] = (i0 + gi + GS * q < n && gk < K) ? a.resp[(row0 + i0 + gi + GS * q) * a.resp_ld + a.k0 + gk] : 0.0;
    if (zt) zv = (i0 + zi < n && dblk + zd < D) ? a.X[(row0 + i0 + zi) * a.ld + dblk + zd] : 0.f;
  };
  auto stash = [&](int buf) {
    double* la = la0 + buf * (256 * MM_KCP);
    double* lb = lb0 + buf * (64 * MM_KCP);
#pragma unroll
    for (int q = 0; q < GQ; ++q) la[gk * MM_KCP + gi + GS * q] = gv[q];
    if (zt) {
      lb[zd * MM_KCP + zi] = (double)zv;
      lb[(zd + 32) * MM_KCP + zi] = (double)(zv * zv);   // np.power(descriptors, 2) in fp32 (fisher_vector.py:104)
    }
  };
  if (n > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int i0 = 0; i0 < n; i0 += MM_KC) {
    const bool more = i0 + MM_KC < n;
    if (more) fetch(i0 + MM_KC);
    const double* la = la0 + buf * (256 * MM_KCP);
    if (tid < 256) {
#pragma unroll
      for (int ii = 0; ii < MM_KC; ++ii) s0 += la[tid * MM_KCP + ii];   // zeros past n; descriptor order
    }
    f64_chunk_mma<MM_MI, 4, MM_KC>(la, lb0 + buf * (64 * MM_KCP), wm * (16 * MM_MI), 0, lane, acc);
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (tid < 256) s0s[tid] = s0;
  __syncthreads();

  if constexpr (RAW) {
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MM_MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = wm * (16 * MM_MI) + 16 * mi + 4 * r + rq;
        if (k >= K) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int d = d0 + 16 * ni + col;
          if (d >= D) continue;
          double* o = a.raw_s + ((int64_t)img * K + k) * 2 * D;
          o[d] = acc[mi][ni][r];
          o[D + d] = acc[mi][ni + 2][r];
        }
      }
    if (db == 0 && tid < K) a.raw_s0[(int64_t)img * K + tid] = s0s[tid];
    return;
  }

  // Epilogue arithmetic: fp64 divisions and square roots cost tens of instructions each, and this epilogue is as long
  // as the MFMA loop if written literally.  The per-(k,d) divisors 1/(sqrt(w) sqrt(cov)) and 1/(sqrt(2) sqrt(w) cov)
  // come from tables built once per GMM, and /n becomes *(1/n): <= 2 ulp of fp64 from the literal formula.
  using OutT = std::conditional_t<OUT64, double, float>;
  OutT* const out = static_cast<OutT*>(a.out) + (int64_t)img * L;
  const bool is_max = a.norm_mode == 3;
  double part = 0.0;
  const int col = lane & 15, rq = lane >> 4;
  const double dn = (double)(n > 0 ? n : 1), rdn = 1.0 / dn;
#pragma unroll
  for (int mi = 0; mi < MM_MI; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = wm * (16 * MM_MI) + 16 * mi + 4 * r + rq;
      const double pp_sum = s0s[k < K ? k : 0] * rdn;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int d = d0 + 16 * ni + col;
        const bool live = k < K && d < D;
        const int64_t kd_i = live ? (int64_t)k * D + d : 0;      // clamped: the loads below are unconditional
        const double mu = a.mu[kd_i], cv = a.cov[kd_i], imu = a.inv_mu[kd_i], isg = a.inv_sg[kd_i];
        const double pp_x = acc[mi][ni][r] * rdn, pp_x2 = acc[mi][ni + 2][r] * rdn;
        double d_mu = pp_x - pp_sum * mu;
        double d_sg = ((-pp_x2 - pp_sum * (mu * mu)) + pp_sum * cv) + (2.0 * pp_x) * mu;
        d_mu = pnorm_t<PM>(d_mu * imu, a.power);
        d_sg = pnorm_t<PM>(d_sg * isg, a.power);
        if (n == 0) d_mu = d_sg = 0.0;   // empty image: zero row (the reference divides by zero; fenced quirk)
        if (live) {
          out[K + kd_i] = (OutT)d_mu;
          out[K + (int64_t)K * D + kd_i] = (OutT)d_sg;
          const double t1 = nterm_t<NM>(d_mu, a.norm_p), t2 = nterm_t<NM>(d_sg, a.norm_p);
          part = is_max ? fmax(part, fmax(t1, t2)) : part + (t1 + t2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one (mi, r) group at a time: with every group's table loads hoisted to the top the kernel spills at 128 registers
    }
  if (db == 0 && tid < K) {   // d_pi (fisher_vector.py:107,117)
    const double w = a.w[tid];
    const double d_pi = n > 0 ? pnorm_t<PM>((s0s[tid] / dn - w) / sqrt(w), a.power) : 0.0;
    out[tid] = (OutT)d_pi;
    const double t = nterm_t<NM>(d_pi, a.norm_p);
    part = is_max ? fmax(part, t) : part + t;
  }
  // deterministic block reduction of the norm term
  for (int m = 32; m >= 1; m >>= 1) {
    const double o = __shfl_xor(part, m, 64);
    part = is_max ? fmax(part, o) : part + o;
  }
  if (lane == 0) red[wm] = part;
  __syncthreads();
  if (tid == 0) {
    double t = red[0];
    for (int wv = 1; wv < MM_THREADS / 64; ++wv) t = is_max ? fmax(t, red[wv]) : t + red[wv];
    if (fold) norm_total = is_max ? fmax(norm_total, t) : norm_total + t;
    else a.partial[(int64_t)img * a.dblocks + db] = t;
  }
  __syncthreads();   // the staging buffers, s0s and red are reused by the next dim block
  }   // dim blocks
  if (fold) {
    // the image's norm, as fisher_scale_kernel forms it; then every thread divides its share of the row the workgroup wrote
    const int tid = threadIdx.x;
    if (tid == 0) {
      const double nrm = a.norm_mode == 2 ? sqrt(norm_total) : (a.norm_mode == 0 ? pow(norm_total, 1.0 / a.norm_p) : norm_total);
      red[0] = nrm + a.eps;
    }
    __syncthreads();   // also makes this workgroup's own global stores visible to all of its threads
    const double den = red[0];
    using OutT2 = std::conditional_t<OUT64, double, float>;
    OutT2* const row = static_cast<OutT2*>(a.out) + (int64_t)img * ((int64_t)a.K + 2 * (int64_t)a.K * a.D);
    const int64_t Lr = (int64_t)a.K + 2 * (int64_t)a.K * a.D;
    // 16-byte accesses, four in flight per thread before the first division (one workgroup has to cover the latency of its own
    // 1-2 MB row: element-at-a-time this tail cost more than the scale pass it replaces)
    constexpr int V = 16 / (int)sizeof(OutT2);
    using VecT = std::conditional_t<OUT64, double2, float4>;
    int64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
      const int64_t nvec = Lr / V;
      VecT* const rv = reinterpret_cast<VecT*>(row);
      int64_t i = tid;
      for (; i + 3 * MM_THREADS < nvec; i += 4 * MM_THREADS) {
        VecT v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = rv[i + u * MM_THREADS];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          OutT2* e = reinterpret_cast<OutT2*>(&v[u]);
#pragma unroll
          for (int c = 0; c < V; ++c) e[c] = (OutT2)((double)e[c] / den);
          rv[i + u * MM_THREADS] = v[u];
        }
      }
      for (; i < nvec; i += MM_THREADS) {
        VecT v = rv[i];
        OutT2* e = reinterpret_cast<OutT2*>(&v);
#pragma unroll
        for (int c = 0; c < V; ++c) e[c] = (OutT2)((double)e[c] / den);
        rv[i] = v;
      }
      done = nvec * V;
    }
    for (int64_t i = done + tid; i < Lr; i += MM_THREADS) row[i] = (OutT2)((double)row[i] / den);
  }
}

constexpr size_t MM_LDS = (size_t)(2 * 256 * MM_KCP + 2 * 64 * MM_KCP + 256 + 8) * sizeof(double);
template <bool RAW, int PM, int NM, bool OUT64>
static int launch_moments(pvs_ctx* ctx, const MomMArgs& m) {
  auto k = fisher_moments_mfma_kernel<RAW, PM, NM, OUT64, false>;
  if constexpr (!RAW) {
    if (m.fold) k = fisher_moments_mfma_kernel<RAW, PM, NM, OUT64, true>;
  }
  PVS_TRY(ensure_lds(ctx, reinterpret_cast<const void*>(k), MM_LDS));
  // one workgroup per (image, dim block), images padded to groups of 8 (one per XCD); fold: one workgroup per image
  const int64_t nblk = m.fold ? m.n_img : (m.n_img + 7) / 8 * 8 * m.dblocks;
  if (nblk > 0x7fffffffLL) PVS_FAIL(PVS_ERR_INVALID, "Fisher moments: %lld workgroups in one launch", (long long)nblk);
  hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(MM_THREADS), MM_LDS, ctx->stream, m);
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ K5 moments + gradients
constexpr int MOM_KB = 16;  // clusters per block
constexpr int MOM_IC = 32;  // descriptors staged per step

struct MomArgs {
  const float* X;
  int D, ld, K;
  const int64_t* offsets;
  const double* resp;  // [total][K]
  const double* w;
  const double* mu;
  const double* cov;
  double power;
  int norm_mode;       // 0 general p, 1 L1, 2 L2, 3 +inf
  double norm_p;
  void* out;           // [n_images][K + 2KD]
  int out_f64;
  double* partial;     // [n_images][blocks_per_image]
  int blocks_per_image;
  int dblocks;
  const double* S;     // PRE: [n_images][K][2D] raw sums from fisher_moments_mfma_kernel
  const double* S0;    // PRE: [n_images][K]
};

__device__ __forceinline__ double power_norm64(double v, double p) {
  if (p == 1.0) return v;
  const double av = fabs(v);
  const double m = (p == 0.5) ? sqrt(av) : pow(av, p);
  return v > 0.0 ? m : (v < 0.0 ? -m : (v == 0.0 ? 0.0 * m : v));
}
__device__ __forceinline__ double norm_term64(double v, int mode, double p) {
  const double av = fabs(v);
  return mode == 2 ? v * v : (mode == 0 ? pow(av, p) : av);
}
__device__ __attribute__((noinline)) double power_norm64_slow(double v, double p) { return power_norm64(v, p); }
__device__ __attribute__((noinline)) double norm_pow64_slow(double av, double p) { return pow(av, p); }

// PRE = false: the sums are accumulated here with vector fp64 FMAs (any K);  PRE = true: the sums were produced by
// the fp64-MFMA kernels and this kernel is only the gradient / power-norm / norm-partial epilogue.
template <int BT, bool PRE>
__global__ __launch_bounds__(BT) void fisher_moments_kernel(MomArgs a) {
  __shared__ double sg[MOM_IC][MOM_KB];
  __shared__ double red[BT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.z, kb = blockIdx.y, db = blockIdx.x;
  const int64_t row0 = a.offsets[img];
  const int n = (int)(a.offsets[img + 1] - row0);
  const int d = db * BT + tid;
  const bool dv = d < a.D;
  const int k0 = kb * MOM_KB;
  const int K = a.K, D = a.D;
  const int64_t L = (int64_t)K + 2 * (int64_t)K * D;

  double s0[MOM_KB], s1[MOM_KB], s2[MOM_KB];
#pragma unroll
  for (int j = 0; j < MOM_KB; ++j) s0[j] = s1[j] = s2[j] = 0.0;

  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < MOM_KB; ++j) {
      const int k = k0 + j;
      if (k < K) {
        s0[j] = a.S0[(int64_t)img * K + k];
        if (dv) {
          s1[j] = a.S[((int64_t)img * K + k) * (2 * D) + d];
          s2[j] = a.S[((int64_t)img * K + k) * (2 * D) + D + d];
        }
      }
    }
  }
  for (int i0 = 0; !PRE && i0 < n; i0 += MOM_IC) {
    __syncthreads();
    for (int idx = tid; idx < MOM_IC * MOM_KB; idx += BT) {
      const int ii = idx / MOM_KB, j = idx % MOM_KB;
      sg[ii][j] = (i0 + ii < n && k0 + j < K) ? a.resp[(row0 + i0 + ii) * K + k0 + j] : 0.0;
    }
    __syncthreads();
    const int in = min(MOM_IC, n - i0);
    for (int ii = 0; ii < in; ++ii) {
      float xf = 0.f;
      if (dv) xf = a.X[(row0 + i0 + ii) * a.ld + d];
      const double x = (double)xf, x2 = (double)(xf * xf);  // np.power(descriptors, 2) in fp32 (fisher_vector.py:104)
#pragma unroll
      for (int j = 0; j < MOM_KB; ++j) {
        const double gm = sg[ii][j];
        s0[j] += gm;
        s1[j] = fma(gm, x, s1[j]);
        s2[j] = fma(gm, x2, s2[j]);
      }
    }
  }

  double part = 0.0;
  if (n > 0) {
    const double dn = (double)n;
#pragma unroll
    for (int j = 0; j < MOM_KB; ++j) {
      const int k = k0 + j;
      const bool kin = k < K;
      const double w = kin ? a.w[k] : 1.0, sw = sqrt(w);
      const double pp_sum = s0[j] / dn;
      if (dv && kin) {
        const double mu = a.mu[(int64_t)k * D + d], cv = a.cov[(int64_t)k * D + d];
        const double pp_x = s1[j] / dn, pp_x2 = s2[j] / dn;
        double d_mu = pp_x - pp_sum * mu;
        double d_sg = ((-pp_x2 - pp_sum * (mu * mu)) + pp_sum * cv) + (2.0 * pp_x) * mu;
        d_mu = d_mu / (sw * sqrt(cv));
        d_sg = d_sg / ((sqrt(2.0) * sw) * cv);
        d_mu = power_norm64(d_mu, a.power);
        d_sg = power_norm64(d_sg, a.power);
        const int64_t o_mu = (int64_t)img * L + K + (int64_t)k * D + d;
        const int64_t o_sg = o_mu + (int64_t)K * D;
        if (a.out_f64) {
          static_cast<double*>(a.out)[o_mu] = d_mu;
          static_cast<double*>(a.out)[o_sg] = d_sg;
        } else {
          static_cast<float*>(a.out)[o_mu] = (float)d_mu;
          static_cast<float*>(a.out)[o_sg] = (float)d_sg;
        }
        const double t1 = norm_term64(d_mu, a.norm_mode, a.norm_p), t2 = norm_term64(d_sg, a.norm_mode, a.norm_p);
        part = a.norm_mode == 3 ? fmax(part, fmax(t1, t2)) : part + (t1 + t2);
      }
      if (db == 0 && tid == j && kin) {
        double d_pi = (pp_sum - w) / sw;
        d_pi = power_norm64(d_pi, a.power);
        const int64_t o = (int64_t)img * L + k;
        if (a.out_f64) static_cast<double*>(a.out)[o] = d_pi;
        else static_cast<float*>(a.out)[o] = (float)d_pi;
        const double t = norm_term64(d_pi, a.norm_mode, a.norm_p);
        part = a.norm_mode == 3 ? fmax(part, t) : part + t;
      }
    }
  } else {
    // empty image: zero row (the reference divides by zero here; fenced quirk)
#pragma unroll
    for (int j = 0; j < MOM_KB; ++j) {
      const int k = k0 + j;
      const bool kin = k < K;
      if (dv && kin) {
        const int64_t o_mu = (int64_t)img * L + K + (int64_t)k * D + d;
        const int64_t o_sg = o_mu + (int64_t)K * D;
        if (a.out_f64) { static_cast<double*>(a.out)[o_mu] = 0.0; static_cast<double*>(a.out)[o_sg] = 0.0; }
        else { static_cast<float*>(a.out)[o_mu] = 0.f; static_cast<float*>(a.out)[o_sg] = 0.f; }
      }
      if (db == 0 && tid == j && kin) {
        if (a.out_f64) static_cast<double*>(a.out)[(int64_t)img * L + k] = 0.0;
        else static_cast<float*>(a.out)[(int64_t)img * L + k] = 0.f;
      }
    }
  }
  // deterministic block reduction of the norm term
  for (int m = 32; m >= 1; m >>= 1) {
    const double o = __shfl_xor(part, m, 64);
    part = a.norm_mode == 3 ? fmax(part, o) : part + o;
  }
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) {
    double t = red[0];
    for (int w = 1; w < BT / 64; ++w) t = a.norm_mode == 3 ? fmax(t, red[w]) : t + red[w];
    a.partial[(int64_t)img * a.blocks_per_image + kb * a.dblocks + db] = t;
  }
}

__global__ __launch_bounds__(256) void fisher_scale_kernel(void* out, int out_f64, int64_t L, const double* partial,
                                                           int blocks_per_image, int norm_mode, double norm_p,
                                                           double eps) {
  __shared__ double s_den;
  const int img = blockIdx.y;
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int b = 0; b < blocks_per_image; ++b) {
      const double v = partial[(int64_t)img * blocks_per_image + b];
      t = norm_mode == 3 ? fmax(t, v) : t + v;
    }
    const double nrm = norm_mode == 2 ? sqrt(t) : (norm_mode == 0 ? pow(t, 1.0 / norm_p) : t);
    s_den = nrm + eps;
  }
  __syncthreads();
  const double den = s_den;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
    if (out_f64) static_cast<double*>(out)[(int64_t)img * L + i] /= den;
    else static_cast<float*>(out)[(int64_t)img * L + i] = (float)((double)static_cast<float*>(out)[(int64_t)img * L + i] / den);
  }
}

__global__ void build_tab2_kernel(const double* __restrict__ prec, const double* __restrict__ mup, int K, int D,
                                  double* __restrict__ tab2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)K * D) return;
  const int k = (int)(i / D), d = (int)(i % D);
  tab2[(int64_t)k * 2 * D + d] = mup[i];
  tab2[(int64_t)k * 2 * D + D + d] = -0.5 * prec[i];
}

static int posterior_mfma_on(pvs_ctx* ctx, const pvs_gmm* g, const float* x, int ld, int64_t total, double* d_resp,
                             double* tab2, double* d_lse = nullptr) {
  const int64_t kd = (int64_t)g->K * g->D;
  hipLaunchKernelGGL(build_tab2_kernel, dim3((unsigned)((kd + 255) / 256)), dim3(256), 0, ctx->stream, g->d_prec, g->d_mup,
                     g->K, g->D, tab2);
  PostMArgs a{x, total, g->D, ld, g->K, tab2, g->d_const, d_resp, d_lse};
  constexpr size_t lds = (size_t)(2 * PM_ROWS * F64_KCP + 2 * PM_COLS * F64_KCP + PM_ROWS * 4 + PM_ROWS * 4 + PM_ROWS) * sizeof(double);   // operands | row maxima per wave column | partial sums | row maxima
  PVS_TRY(ensure_lds(ctx, reinterpret_cast<const void*>(gmm_posterior_mfma_kernel), lds));
  ScopedTimer tm(ctx, T_FPOST);
  hipLaunchKernelGGL(gmm_posterior_mfma_kernel, dim3((unsigned)((total + PM_ROWS - 1) / PM_ROWS)), dim3(PM_THREADS), lds, ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_gmm_posterior(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int kind, int ld, int64_t total,
                         double* d_resp) {
  if (total <= 0) return PVS_OK;
  const void* x = d_desc;
  int k = kind;
  PVS_TRY(materialise_f32(ctx, x, k, total, g->D));
  double* tab = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, (size_t)2 * g->K * g->D * sizeof(double), reinterpret_cast<void**>(&tab)));
  if (g->K <= PM_COLS) return posterior_mfma_on(ctx, g, static_cast<const float*>(x), ld, total, d_resp, tab);
  return posterior_on(ctx, g, static_cast<const float*>(x), ld, total, d_resp, tab);
}

// one batch of images (offsets are absolute descriptor rows; the batch's descriptors are rows [t0, t0 + tn))
static int fisher_batch(pvs_ctx* ctx, const pvs_gmm* g, const float* x, int ld, const int64_t* d_offsets, int64_t img0,
                        int64_t n_img, int64_t t0, int64_t tn, const pvs_norm_params& prm, void* d_out, int out_f64,
                        char* ws, size_t tab_b, size_t resp_b) {
  const int K = g->K, D = g->D;
  const bool mfma = K <= PM_COLS;
  const int bt = D <= 128 ? 128 : 256;
  const int dblocks = mfma ? (D + MM_DIMS - 1) / MM_DIMS : (D + bt - 1) / bt;
  const int kblocks = mfma ? 1 : (K + MOM_KB - 1) / MOM_KB;
  const int bpi = dblocks * kblocks;
  double* tab = reinterpret_cast<double*>(ws);
  double* resp = reinterpret_cast<double*>(ws + tab_b);
  double* partial = reinterpret_cast<double*>(ws + tab_b + resp_b);
  const int64_t L = (int64_t)K + 2 * (int64_t)K * D;
  double* resp_abs = resp - t0 * K;  // kernels index responsibilities by ABSOLUTE descriptor row
  if (tn > 0) {
    if (mfma) PVS_TRY(posterior_mfma_on(ctx, g, x + t0 * ld, ld, tn, resp, tab));
    else PVS_TRY(posterior_on(ctx, g, x + t0 * ld, ld, tn, resp, tab));
  }
  const double ord = prm.norm_order;
  const int norm_mode = std::isinf(ord) ? 3 : (ord == 2.0 ? 2 : (ord == 1.0 ? 1 : 0));
  void* out_b = out_f64 ? static_cast<void*>(static_cast<double*>(d_out) + img0 * L)
                        : static_cast<void*>(static_cast<float*>(d_out) + img0 * L);
  ScopedTimer tm(ctx, T_FMOM);
  bool folded = false;
  if (mfma) {
    MomMArgs m{x, D, ld, K, d_offsets + img0, resp_abs, g->d_w, g->d_mu, g->d_cov, g->d_inv_mu, g->d_inv_sg, prm.power_norm_weight, norm_mode, ord,
               out_b, out_f64, partial, dblocks, nullptr, nullptr, K, 0, 0, 0.0, n_img};
    const int pm = prm.power_norm_weight == 1.0 ? 0 : (prm.power_norm_weight == 0.5 ? 1 : 2);
    const int nm = norm_mode == 2 ? 2 : (norm_mode == 0 ? 0 : 1);
    // PVS_OPT_FISHER_SCALE = 2: one workgroup per image walks the image's dim blocks and divides the row by its norm itself (no
    // second pass).  configs[2], alternating on one box (profiles/r03_fisher_scale_in_kernel.txt): 23.3 against 23.7 ms (f32 rows),
    // 26.0 against 26.3 (f64 rows) -- but every dim block then re-reads the image's gamma rows from beyond L2 (63.7 + 16.7 GB
    // fetched + written against 31.4 + 16.7 GB for moments + scale pass, profiles/r03_fisher_scale_pmc.txt): 1.6 % of the time for
    // 1.7 x the bytes, so the default stays one workgroup per (image, dim block) with the image's blocks on one XCD, and a second pass.
    const bool fold = ctx->opt[PVS_OPT_FISHER_SCALE] == 2;
    m.fold = fold ? 1 : 0;
    m.eps = prm.epsilon;
    folded = fold;
#define PVS_MOM(PMV, NMV)                                                                             \
  do {                                                                                                \
    if (out_f64) PVS_TRY((launch_moments<false, PMV, NMV, true>(ctx, m)));                            \
    else PVS_TRY((launch_moments<false, PMV, NMV, false>(ctx, m)));                                   \
  } while (0)
    if (pm == 1 && nm == 2) PVS_MOM(1, 2);        // the reference's defaults: p = 0.5, L2
    else if (pm == 0 && nm == 2) PVS_MOM(0, 2);
    else if (pm == 1 && nm == 1) PVS_MOM(1, 1);
    else if (pm == 0 && nm == 1) PVS_MOM(0, 1);
    else if (nm == 2) PVS_MOM(2, 2);
    else if (nm == 1) PVS_MOM(2, 1);
    else PVS_MOM(2, 0);
#undef PVS_MOM
  } else {
    MomArgs a{};
    a.X = x; a.D = D; a.ld = ld; a.K = K; a.offsets = d_offsets + img0; a.resp = resp_abs;
    a.w = g->d_w; a.mu = g->d_mu; a.cov = g->d_cov; a.power = prm.power_norm_weight;
    a.norm_mode = norm_mode; a.norm_p = ord; a.out = out_b; a.out_f64 = out_f64; a.partial = partial;
    a.blocks_per_image = bpi; a.dblocks = dblocks;
    dim3 grid((unsigned)dblocks, (unsigned)kblocks, (unsigned)n_img);
    if (bt == 128) hipLaunchKernelGGL((fisher_moments_kernel<128, false>), grid, dim3(128), 0, ctx->stream, a);
    else hipLaunchKernelGGL((fisher_moments_kernel<256, false>), grid, dim3(256), 0, ctx->stream, a);
  }
  PVS_HIP(hipGetLastError());
  if (folded) return PVS_OK;
  const unsigned sx = (unsigned)std::min<int64_t>((L + 255) / 256, 64);
  hipLaunchKernelGGL(fisher_scale_kernel, dim3(sx, (unsigned)n_img), dim3(256), 0, ctx->stream, out_b, out_f64, L, partial, bpi,
                     norm_mode, ord, prm.epsilon);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------ training: one EM pass
// E-step + sufficient statistics of one EM iteration (sklearn/mixture/_base.py:_e_step, _gaussian_mixture.py:
// _estimate_gaussian_parameters): d_stats = [s0 (K) | per k: sum gamma x (D), sum gamma x**2 (D)] and
// d_stats[K + 2KD] = sum_i log p(x_i).  The descriptors go through in batches of fixed-size chunks; every chunk's sums
// are formed by the Fisher moments kernel (RAW) and the chunks are added in order, in fp64.
int launch_gmm_em_step(pvs_ctx* ctx, const pvs_gmm* g, const float* x, int ld, int64_t total, double* d_stats) {
  const int K = g->K, D = g->D;
  if (K > POST_KMAX) PVS_FAIL(PVS_ERR_UNSUPPORTED, "GMM training on the device supports at most %d components (got %d)", POST_KMAX, K);
  if (total <= 0) PVS_FAIL(PVS_ERR_INVALID, "GMM training needs at least one descriptor");
  constexpr int CHUNK = 2048;
  const int nslab = (K + PM_COLS - 1) / PM_COLS;   // the moments kernel takes 256 components at a time
  const int64_t len = (int64_t)K * 2 * D;
  const size_t tab_b = ((size_t)2 * K * D * 8 + 255) / 256 * 256;
  const int64_t rows_per_batch = std::max<int64_t>(CHUNK, (((int64_t)2 << 30) / ((int64_t)K * 8)) / CHUNK * CHUNK);
  const int dblocks = (D + MM_DIMS - 1) / MM_DIMS;
  int first = 1;
  for (int64_t t0 = 0; t0 < total; t0 += rows_per_batch) {
    const int64_t tn = std::min(rows_per_batch, total - t0);
    const int64_t nch = (tn + CHUNK - 1) / CHUNK;
    const size_t resp_b = ((size_t)tn * K * 8 + 255) / 256 * 256;
    const size_t lse_b = ((size_t)tn * 8 + 255) / 256 * 256;
    const size_t off_b = ((size_t)(nch + 1) * 8 + 255) / 256 * 256;
    const int64_t slab_len = (int64_t)PM_COLS * 2 * D;
    const size_t raw_b = ((size_t)nch * slab_len * 8 + 255) / 256 * 256;
    const size_t s0_b = ((size_t)nch * PM_COLS * 8 + 255) / 256 * 256;
    char* ws = nullptr;
    PVS_TRY(ws_reserve(ctx, 1, tab_b + resp_b + lse_b + off_b + raw_b + s0_b, reinterpret_cast<void**>(&ws)));
    double* tab = reinterpret_cast<double*>(ws);
    double* resp = reinterpret_cast<double*>(ws + tab_b);
    double* lse = reinterpret_cast<double*>(ws + tab_b + resp_b);
    int64_t* off = reinterpret_cast<int64_t*>(ws + tab_b + resp_b + lse_b);
    double* raw = reinterpret_cast<double*>(ws + tab_b + resp_b + lse_b + off_b);
    double* raw0 = reinterpret_cast<double*>(ws + tab_b + resp_b + lse_b + off_b + raw_b);
    if (K <= PM_COLS) PVS_TRY(posterior_mfma_on(ctx, g, x + t0 * ld, ld, tn, resp, tab, lse));
    else PVS_TRY(posterior_on(ctx, g, x + t0 * ld, ld, tn, resp, tab, lse));
    hipLaunchKernelGGL(chunk_offsets_kernel, dim3((unsigned)((nch + 256) / 256)), dim3(256), 0, ctx->stream, off, t0, tn, CHUNK, nch);
    for (int sl = 0; sl < nslab; ++sl) {
      const int k0 = sl * PM_COLS, ks = std::min(PM_COLS, K - k0);
      {
        ScopedTimer tm(ctx, T_FMOM);
        MomMArgs m{x, D, ld, ks, off, resp - t0 * K, g->d_w, g->d_mu, g->d_cov, g->d_inv_mu, g->d_inv_sg, 1.0, 2, 2.0,
                   nullptr, 1, nullptr, dblocks, raw, raw0, K, k0, 0, 0.0, nch};
        PVS_TRY((launch_moments<true, 2, 0, true>(ctx, m)));
      }
      const int64_t sl_len = (int64_t)ks * 2 * D;
      hipLaunchKernelGGL(reduce_chunks_kernel<double>, dim3((unsigned)((ks + 255) / 256)), dim3(256), 0, ctx->stream, raw0, nch, (int64_t)ks,
                         d_stats + k0, first);
      hipLaunchKernelGGL(reduce_chunks_kernel<double>, dim3((unsigned)((sl_len + 255) / 256)), dim3(256), 0, ctx->stream, raw, nch, sl_len,
                         d_stats + K + (int64_t)k0 * 2 * D, first);
    }
    hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(256), 0, ctx->stream, lse, tn, d_stats + K + len, first);
    PVS_HIP(hipGetLastError());
    first = 0;
  }
  return PVS_OK;
}

int launch_fisher(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int kind, int ld, const int64_t* d_offsets,
                  int64_t n_images, int64_t total, const pvs_norm_params& prm, void* d_out, int out_f64) {
  if (n_images <= 0) return PVS_OK;
  const void* xv = d_desc;
  int k = kind;
  PVS_TRY(materialise_f32(ctx, xv, k, total, g->D));
  const float* x = static_cast<const float*>(xv);
  const int K = g->K, D = g->D;
  // Images go through in batches so that the responsibilities (fp64, n x K) and raw moment sums (K x 2D per image)
  // stay within ~3 GiB of workspace.  Batch boundaries need the CSR offsets on the host.
  std::vector<int64_t> off((size_t)n_images + 1);
  PVS_HIP(hipMemcpyAsync(off.data(), d_offsets, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  // the offsets index workspaces sized from `total`: a bad table is an argument error, never an out-of-bounds access
  if (off[0] != 0) PVS_FAIL(PVS_ERR_INVALID, "offsets[0] must be 0 (got %lld)", (long long)off[0]);
  for (int64_t i = 0; i < n_images; ++i)
    if (off[i + 1] < off[i]) PVS_FAIL(PVS_ERR_INVALID, "offsets must be non-decreasing (image %lld)", (long long)i);
  if (off[n_images] != total)
    PVS_FAIL(PVS_ERR_INVALID, "offsets[n_images] = %lld does not match total_desc = %lld", (long long)off[n_images], (long long)total);
  const size_t per_img = 4096;
  const size_t budget = (size_t)3 << 30;
  const int bt = D <= 128 ? 128 : 256;
  const int bpi = ((D + bt - 1) / bt) * ((K + MOM_KB - 1) / MOM_KB);
  int64_t i0 = 0;
  while (i0 < n_images) {
    int64_t i1 = i0;
    size_t bytes = 0;
    while (i1 < n_images && i1 - i0 < 65535) {
      const size_t add = per_img + (size_t)(off[i1 + 1] - off[i1]) * K * 8;
      if (i1 > i0 && bytes + add > budget) break;
      bytes += add;
      ++i1;
    }
    const int64_t n_img = i1 - i0, t0 = off[i0], tn = off[i1] - off[i0];
    const size_t tab_b = ((size_t)2 * K * D * 8 + 255) / 256 * 256;
    const size_t resp_b = ((size_t)std::max<int64_t>(tn, 1) * K * 8 + 255) / 256 * 256;
    const int dbl = K <= PM_COLS ? (D + MM_DIMS - 1) / MM_DIMS : bpi;
    const size_t part_b = (size_t)n_img * std::max(dbl, bpi) * 8;
    char* ws = nullptr;
    PVS_TRY(ws_reserve(ctx, 1, tab_b + resp_b + part_b, reinterpret_cast<void**>(&ws)));
    PVS_TRY(fisher_batch(ctx, g, x, ld, d_offsets, i0, n_img, t0, tn, prm, d_out, out_f64, ws, tab_b, resp_b));
    i0 = i1;
  }
  return PVS_OK;
}

}  // namespace pvs
