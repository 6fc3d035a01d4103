// Cosine similarity on gfx950: row norms + the N x M similarity GEMM.
//
// Reference semantics: pyvisim/_utils.py:312-330 -> sklearn.metrics.pairwise.cosine_similarity
// (sklearn/metrics/pairwise.py:1683-1738): rows L2-normalised (zero norms -> 1, i.e. zero rows stay
// zero), then Xn @ Yn^T; fp32 iff both operands are fp32.  Here the 1/||.|| factors are applied to the
// accumulator in the epilogue (same value up to fp32 rounding; tolerance stated in the tests).
//
// K6 fp32: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), 128x128 block tile, BK = 32, both operands are
// K-major so tiles stream HBM -> LDS with 16-B-per-lane LDS-DMA (global_load_lds_dwordx4), double
// buffered; LDS image XOR-swizzled on the SOURCE address + matching XOR on the ds_read_b128 (conflict
// free); block -> tile map is XCD-aware so the 64 tiles resident on one XCD share 8 A and 8 B panels.
#include "common.hpp"

namespace pvs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------- row norms
__global__ __launch_bounds__(256) void row_inv_norms_kernel(const float* __restrict__ x, int64_t rows, int64_t L,
                                                            float* __restrict__ inv, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * L;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (vec) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int64_t i = lane; i < (L >> 2); i += 64) {
      const float4 v = p4[i];
      s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
    }
  } else {
    for (int64_t i = lane; i < L; i += 64) s0 = fmaf(p[i], p[i], s0);
  }
  float s = (s0 + s1) + (s2 + s3);
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) inv[row] = s > 0.f ? 1.f / sqrtf(s) : 1.f;
}

int launch_row_inv_norms(pvs_ctx* ctx, const float* d_x, int64_t rows, int64_t L, float* d_inv) {
  if (rows <= 0) return PVS_OK;
  const int vec = (L % 4 == 0) && (reinterpret_cast<uintptr_t>(d_x) % 16 == 0);
  ScopedTimer tm(ctx, T_MISC);
  hipLaunchKernelGGL(row_inv_norms_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, d_x, rows, L,
                     d_inv, vec);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// ------------------------------------------------------------------------------------- K6 fp32 MFMA
constexpr int GT_M = 128, GT_N = 128, GT_K = 32;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_KBLOCK = 1024;  // k-values per MFMA accumulation chain (see kernel)
constexpr int TILE_BYTES = GT_M * GT_K * 4;  // 16 KiB per operand per stage

struct GemmArgs {
  const float* A;
  const float* B;
  int64_t M, N, L;
  const float* inva;
  const float* invb;
  float* out;
  int64_t ldo;
  int tiles_m, tiles_n;
};

// LDS image of one operand tile: [128 rows][8 chunks of 16 B]; position (r, c) holds the row's global
// chunk c ^ ((r >> 1) & 7).  One wave-instruction of LDS-DMA writes 64 x 16 B = 8 consecutive rows.
__device__ __forceinline__ void stage_tile(const float* __restrict__ base, int64_t nrows, int64_t L, int64_t row0,
                                           int64_t k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 32 * wave + 8 * q + (lane >> 3);
    const int c = lane & 7;
    const int gc = c ^ ((r >> 1) & 7);
    int64_t grow = row0 + r;
    grow = grow < nrows ? grow : nrows - 1;  // clamp: rows past the edge are computed and discarded
    const int64_t kc = k0 + 4 * gc;
    const float* src = kc < L ? base + grow * L + kc : g_zero16;
    // LDS destination = wave-uniform base + lane * 16 B; the per-lane SOURCE address carries the swizzle
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + (32 * wave + 8 * q) * (GT_K * 4)),
                                     16, 0, 0);
  }
}

__device__ __forceinline__ float4 lds_frag(const char* lds_tile, int row, int cc) {
  return *reinterpret_cast<const float4*>(lds_tile + (row * 8 + (cc ^ ((row >> 1) & 7))) * 16);
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void cosine_gemm_f32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A tile | B tile]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware, grouped tile order (bijective for any grid size)
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, pos = bid >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int grp = lin / per_group, rem = lin - grp * per_group;
  const int first_m = grp * GM;
  const int gsz = min(GM, g.tiles_m - first_m);
  const int tm = first_m + rem % gsz, tn = rem / gsz;
  const int64_t m0 = (int64_t)tm * GT_M, n0 = (int64_t)tn * GT_N;

  // Two-level summation: the MFMA chain (a k-ordered fp32 fma chain) runs over at most GEMM_KBLOCK k-values,
  // then its tile is added into `tot`.  A single 32768-long chain drifts ~4e-6 relative (random-walk
  // rounding); blocked at 1024 the error is ~1.6e-7, the level of a blocked BLAS sgemm (the reference).
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; tot[a][b][r] = 0.f; }

  const int nk = (int)((g.L + GT_K - 1) / GT_K);
  stage_tile(g.A, g.M, g.L, m0, 0, smem, wave, lane);
  stage_tile(g.B, g.N, g.L, n0, 0, smem + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * (2 * TILE_BYTES);
    char* nxt = smem + ((kt + 1) & 1) * (2 * TILE_BYTES);
    if (kt + 1 < nk) {
      stage_tile(g.A, g.M, g.L, m0, (int64_t)(kt + 1) * GT_K, nxt, wave, lane);
      stage_tile(g.B, g.N, g.L, n0, (int64_t)(kt + 1) * GT_K, nxt + TILE_BYTES, wave, lane);
    }
    const char* la = cur;
    const char* lb = cur + TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cc = 2 * t + h;
      const float4 a0 = lds_frag(la, wm * 64 + i, cc);
      const float4 a1 = lds_frag(la, wm * 64 + 32 + i, cc);
      const float4 b0 = lds_frag(lb, wn * 64 + i, cc);
      const float4 b1 = lds_frag(lb, wn * 64 + 32 + i, cc);
#define PVS_MF(AV, BV, ACC)                                                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, BV.x, ACC, 0, 0, 0);      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, BV.y, ACC, 0, 0, 0);      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, BV.z, ACC, 0, 0, 0);      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, BV.w, ACC, 0, 0, 0);
      PVS_MF(a0, b0, acc[0][0]) PVS_MF(a0, b1, acc[0][1]) PVS_MF(a1, b0, acc[1][0]) PVS_MF(a1, b1, acc[1][1])
#undef PVS_MF
    }
    if ((kt & (GEMM_KBLOCK / GT_K - 1)) == GEMM_KBLOCK / GT_K - 1) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          tot[a][b] += acc[a][b];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] += tot[a][b];

  // ---- epilogue: scale by 1/(||a|| ||b||); C/D layout col = lane&31, row = (reg&3)+8*(reg>>2)+4*h
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t n = n0 + wn * 64 + 32 * ni + i;
      const float sb = (n < g.N && g.invb) ? g.invb[n] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M && n < g.N) {
          const float sa = g.inva ? g.inva[m] : 1.f;
          g.out[m * g.ldo + n] = acc[mi][ni][r] * sa * sb;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------- generic tiled fallback
// Any dtype/alignment (fp64 operands of the Fisher path; odd L).  64x64 tile, 4x4 outputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void cosine_gemm_generic_kernel(const T* __restrict__ A, int64_t M,
                                                                  const T* __restrict__ B, int64_t N, int64_t L,
                                                                  const T* __restrict__ inva,
                                                                  const T* __restrict__ invb, T* __restrict__ out,
                                                                  int64_t ldo) {
  constexpr int TS = 64, KS = 16;
  __shared__ T sa[KS][TS + 1];
  __shared__ T sb[KS][TS + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * TS, n0 = (int64_t)blockIdx.x * TS;
  T acc[4][4] = {};
  for (int64_t k0 = 0; k0 < L; k0 += KS) {
    for (int idx = threadIdx.x; idx < TS * KS; idx += 256) {
      const int r = idx / KS, c = idx % KS;
      const int64_t k = k0 + c;
      sa[c][r] = (m0 + r < M && k < L) ? A[(m0 + r) * L + k] : T(0);
      sb[c][r] = (n0 + r < N && k < L) ? B[(n0 + r) * L + k] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      T av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = sa[c][ty * 4 + u]; bv[u] = sb[c][tx * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fma(av[u], bv[v], acc[u][v]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int64_t m = m0 + ty * 4 + u, n = n0 + tx * 4 + v;
      if (m < M && n < N) out[m * ldo + n] = acc[u][v] * (inva ? inva[m] : T(1)) * (invb ? invb[n] : T(1));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void row_inv_norms_generic_kernel(const T* __restrict__ x, int64_t rows, int64_t L,
                                                                    T* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  T s = 0;
  for (int64_t i = lane; i < L; i += 64) s = fma(x[row * L + i], x[row * L + i], s);
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) inv[row] = s > T(0) ? T(1) / sqrt(s) : T(1);
}

int launch_cosine_f32(pvs_ctx* ctx, const float* A, int64_t M, const float* B, int64_t N, int64_t L,
                      const float* inva, const float* invb, float* out, int64_t ldo) {
  if (M <= 0 || N <= 0) return PVS_OK;
  if (L <= 0) PVS_FAIL(PVS_ERR_INVALID, "cosine: L must be positive");
  const bool fast = (L % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(B) % 16 == 0);
  ScopedTimer tm(ctx, T_GEMM);
  if (fast) {
    GemmArgs g{A, B, M, N, L, inva, invb, out, ldo, (int)((M + GT_M - 1) / GT_M), (int)((N + GT_N - 1) / GT_N)};
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n;
    if (nwg > 0x7fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "cosine: too many tiles for one launch");
    const size_t lds = 4 * TILE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
      PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_gemm_f32_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(cosine_gemm_f32_kernel, dim3((unsigned)nwg), dim3(GEMM_THREADS), lds, ctx->stream, g);
  } else {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(cosine_gemm_generic_kernel<float>, grid, dim3(256), 0, ctx->stream, A, M, B, N, L, inva, invb,
                       out, ldo);
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_cosine_f64(pvs_ctx* ctx, const double* A, int64_t M, const double* B, int64_t N, int64_t L, double* out) {
  if (M <= 0 || N <= 0) return PVS_OK;
  double* inv = nullptr;
  PVS_TRY(ws_reserve(ctx, 1, (size_t)(M + N) * sizeof(double), reinterpret_cast<void**>(&inv)));
  ScopedTimer tm(ctx, T_GEMM);
  hipLaunchKernelGGL(row_inv_norms_generic_kernel<double>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, ctx->stream, A,
                     M, L, inv);
  hipLaunchKernelGGL(row_inv_norms_generic_kernel<double>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, B,
                     N, L, inv + M);
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
  hipLaunchKernelGGL(cosine_gemm_generic_kernel<double>, grid, dim3(256), 0, ctx->stream, A, M, B, N, L, inv, inv + M,
                     out, N);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

}  // namespace pvs
