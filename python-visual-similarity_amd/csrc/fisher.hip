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
// (MI355X runs fp64 at half the fp32 rate -- cheaper than losing 3 digits).  Round-1 kernels use vector
// fp64 FMA with LDS-broadcast operands; v_mfma_f64_16x16x4 is the planned upgrade.
#include "common.hpp"
#include "desc_load.hpp"

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
    out[row * D + d] = DescTraits<KIND>::rootsift ? rootsift_apply(v, s) : v;
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
                        double* tabT) {
  if (g->K > POST_KMAX) PVS_FAIL(PVS_ERR_UNSUPPORTED, "GMM with K = %d components exceeds the kernel limit (%d)", g->K,
                                POST_KMAX);
  const int64_t kd = (int64_t)g->K * g->D;
  double* precT = tabT;
  double* mupT = tabT + kd;
  hipLaunchKernelGGL(transpose_tables_kernel, dim3((unsigned)((kd + 255) / 256)), dim3(256), 0, ctx->stream, g->d_prec,
                     g->d_mup, g->K, g->D, precT, mupT);
  PostArgs a{x, total, g->D, ld, g->K, mupT, precT, g->d_const, d_resp};
  const size_t lds = (size_t)POST_DK * POST_ROWS * sizeof(double2) + (size_t)POST_ROWS * g->K * sizeof(double);
  PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_posterior_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ScopedTimer tm(ctx, T_FPOST);
  hipLaunchKernelGGL(gmm_posterior_kernel, dim3((unsigned)((total + POST_ROWS - 1) / POST_ROWS)), dim3(256), lds,
                     ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_gmm_posterior(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int kind, int ld, int64_t total,
                         double* d_resp) {
  if (total <= 0) return PVS_OK;
  const void* x = d_desc;
  int k = kind;
  PVS_TRY(materialise_f32(ctx, x, k, total, g->D));
  double* tabT = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, (size_t)2 * g->K * g->D * sizeof(double), reinterpret_cast<void**>(&tabT)));
  return posterior_on(ctx, g, static_cast<const float*>(x), ld, total, d_resp, tabT);
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

template <int BT>
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

  for (int i0 = 0; i0 < n; i0 += MOM_IC) {
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

int launch_fisher(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int kind, int ld, const int64_t* d_offsets,
                  int64_t n_images, int64_t total, const pvs_norm_params& prm, void* d_out, int out_f64) {
  if (n_images <= 0) return PVS_OK;
  if (n_images > 65535) PVS_FAIL(PVS_ERR_UNSUPPORTED, "fisher: at most 65535 images per call (got %lld); batch the call",
                                (long long)n_images);
  const void* x = d_desc;
  int k = kind;
  PVS_TRY(materialise_f32(ctx, x, k, total, g->D));
  const int K = g->K, D = g->D;
  const int bt = D <= 128 ? 128 : 256;
  const int dblocks = (D + bt - 1) / bt, kblocks = (K + MOM_KB - 1) / MOM_KB;
  const int bpi = dblocks * kblocks;
  // workspace: transposed tables | responsibilities | norm partials
  const size_t tab_b = ((size_t)2 * K * D * 8 + 255) / 256 * 256;
  const size_t resp_b = ((size_t)std::max<int64_t>(total, 1) * K * 8 + 255) / 256 * 256;
  const size_t part_b = (size_t)n_images * bpi * 8;
  char* ws = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, tab_b + resp_b + part_b, reinterpret_cast<void**>(&ws)));
  double* tabT = reinterpret_cast<double*>(ws);
  double* resp = reinterpret_cast<double*>(ws + tab_b);
  double* partial = reinterpret_cast<double*>(ws + tab_b + resp_b);
  if (total > 0) PVS_TRY(posterior_on(ctx, g, static_cast<const float*>(x), ld, total, resp, tabT));

  MomArgs a{};
  a.X = static_cast<const float*>(x); a.D = D; a.ld = ld; a.K = K; a.offsets = d_offsets; a.resp = resp;
  a.w = g->d_w; a.mu = g->d_mu; a.cov = g->d_cov; a.power = prm.power_norm_weight;
  const double ord = prm.norm_order;
  a.norm_mode = std::isinf(ord) ? 3 : (ord == 2.0 ? 2 : (ord == 1.0 ? 1 : 0));
  a.norm_p = ord; a.out = d_out; a.out_f64 = out_f64; a.partial = partial; a.blocks_per_image = bpi; a.dblocks = dblocks;
  const int64_t L = (int64_t)K + 2 * (int64_t)K * D;
  {
    ScopedTimer tm(ctx, T_FMOM);
    dim3 grid((unsigned)dblocks, (unsigned)kblocks, (unsigned)n_images);
    if (bt == 128) hipLaunchKernelGGL(fisher_moments_kernel<128>, grid, dim3(128), 0, ctx->stream, a);
    else hipLaunchKernelGGL(fisher_moments_kernel<256>, grid, dim3(256), 0, ctx->stream, a);
    PVS_HIP(hipGetLastError());
    const unsigned sx = (unsigned)std::min<int64_t>((L + 255) / 256, 64);
    hipLaunchKernelGGL(fisher_scale_kernel, dim3(sx, (unsigned)n_images), dim3(256), 0, ctx->stream, d_out, out_f64, L,
                       partial, bpi, a.norm_mode, a.norm_p, prm.epsilon);
    PVS_HIP(hipGetLastError());
  }
  return PVS_OK;
}

}  // namespace pvs
