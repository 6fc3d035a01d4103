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
#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "gemm_mfma.hpp"
#include "gemm_f64.hpp"
#include "gemm_f16_8ph.hpp"
#include "gemm_f16_2lvl.hpp"

namespace pvs {

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
// Kernel: gemm_mfma.hpp (128x128 tile, 4 waves, 2 stages, 2 workgroups per CU).  The host builds the tile list:
// 8x8 super-tiles for L2 panel sharing, upper triangle only when A == B (SYMM), and -- when the last round of
// workgroups would be mostly empty -- hands the remaining tiles to a deterministic split-K tail.
using GemmMain = GemmCfg<128, 128, 2, 2, 2, false>;   // exact fp32: 2 workgroups per CU
using GemmHalf = GemmCfg<256, 256, 2, 4, 2, true>;    // fp16 operands: 1 workgroup (8 waves) per CU
using GemmPre = GemmCfg<128, 128, 2, 2, 2, true>;     // fp16 operands, chains of 1024 k summed in fp32 (bounded error: prefilter)
// MODEL: 0 exact fp32, 1 fp16 256x256 single-level, 2 fp16 128x128 two-level
template <int MODEL> struct GemmModel;
template <> struct GemmModel<0> { using Cfg = GemmMain; static constexpr bool F16 = false, TWO = true; static constexpr int BM = 128, WN = 2, PER_CU = 2; };
template <> struct GemmModel<1> { using Cfg = GemmHalf; static constexpr bool F16 = true, TWO = false; static constexpr int BM = 256, WN = 4, PER_CU = 1; };
template <> struct GemmModel<2> { using Cfg = GemmPre; static constexpr bool F16 = true, TWO = true; static constexpr int BM = 128, WN = 2, PER_CU = 2; };

using GemmPlan = pvs_ctx::GemmPlanSlot;   // lives in the context (one device, one stream): never shared between contexts

static int build_plan(pvs_ctx* ctx, int which, int tiles_m, int tiles_n, bool symm, int slots, GemmPlan** out) {
  GemmPlan& P = ctx->gemm_plan[which];
  const int key[4] = {tiles_m, tiles_n, symm ? 1 : 0, slots};
  if (memcmp(P.key, key, sizeof(key)) != 0) {
    std::vector<GemmTile> t;
    if (symm) {
      const int TS = (tiles_m + 7) / 8;
      for (int si = 0; si < TS; ++si)
        for (int sj = si; sj < TS; ++sj)
          for (int w = 0; w < 64; ++w) {
            const int tm = si * 8 + (w & 7), tn = sj * 8 + (w >> 3);
            if (tm < tiles_m && tn < tiles_m && tn >= tm) t.push_back({tm, tn});
          }
    } else {
      for (int g0 = 0; g0 < tiles_m; g0 += 8) {
        const int gsz = std::min(8, tiles_m - g0);
        for (int tn = 0; tn < tiles_n; ++tn)
          for (int dm = 0; dm < gsz; ++dm) t.push_back({g0 + dm, tn});
      }
    }
    const int total = (int)t.size();
    // How to cut the work into launches.  One tile takes the same time T whether the chip is full or not, so a problem of
    // few tiles (a rank's share in the multi-GPU scheme, a query batch) is fast only when its tiles are split along K into
    // s slices (whole 1024-k chains each; the reduce adds the chain images in chain order, so the scores do not change).
    // Cost in units of T:  rounds(blocks) * (1/s + eps) + the reduce's traffic;  candidates: full rounds unsplit + the last
    // partial round split (A), or every tile split (B).
    const double eps = 0.02, red = 3.0e-4;
    auto rounds = [&](long blocks) { return (double)((blocks + slots - 1) / slots); };
    const int full = total / slots, rem = total % slots;
    double best = rounds(total);            // nothing split
    P.n_main = total;
    P.n_tail = 0;
    P.splitk = 1;
    for (int sk = 2; sk <= 32; sk *= 2) {
      if (rem > 0) {
        const double a = full + rounds((long)rem * sk) * (1.0 / sk + eps) + red * rem;
        if (a < best - 1e-9) { best = a; P.n_main = total - rem; P.n_tail = rem; P.splitk = sk; }
      }
      if (total < 4 * slots) {
        const double b = rounds((long)total * sk) * (1.0 / sk + eps) + red * total;
        if (b < best - 1e-9) { best = b; P.n_main = 0; P.n_tail = total; P.splitk = sk; }
      }
    }
    if (P.cap < t.size()) {
      PVS_HIP(hipStreamSynchronize(ctx->stream));
      if (P.d_tiles) PVS_HIP(hipFree(P.d_tiles));
      P.d_tiles = nullptr;
      P.cap = 0;
      P.key[0] = -1;
      const size_t cap = t.size() + t.size() / 4 + 64;
      PVS_HIP(hipMalloc(&P.d_tiles, cap * sizeof(GemmTile)));
      P.cap = cap;
    }
    PVS_HIP(hipMemcpyAsync(P.d_tiles, t.data(), t.size() * sizeof(GemmTile), hipMemcpyHostToDevice, ctx->stream));
    PVS_HIP(hipStreamSynchronize(ctx->stream));  // `t` is pageable host memory going out of scope
    memcpy(P.key, key, sizeof(key));
  }
  *out = &P;
  return PVS_OK;
}

template <bool SYMM, int MODEL, bool DUAL = false>
static int launch_gemm_mfma(pvs_ctx* ctx, GemmArgs g, const GemmPlan& plan) {
  using GM = GemmModel<MODEL>;
  using Cfg = typename GM::Cfg;
  constexpr bool F16 = GM::F16;
  constexpr int BM = GM::BM, WM = 2, WN = GM::WN;
  // fp16 256x256: single-level accumulation (input rounding dominates); LDS-DMA interleaved into the MFMA phase pays
  // only in the general (non-symmetric) order (measured 3.62 vs 3.91 ms; symmetric 2.19 vs 1.98 ms)
  constexpr bool TWO = GM::TWO, ILV = MODEL == 1 && !SYMM;
  auto kfull = gemm_mfma_kernel<BM, BM, WM, WN, 2, SYMM, 2, GEMM_MODE_FULL, false, F16, TWO, ILV, DUAL>;
  auto kpart = gemm_mfma_kernel<BM, BM, WM, WN, 2, SYMM, 2, GEMM_MODE_PARTIAL, false, F16, TWO, ILV, DUAL>;
  auto kred = gemm_mfma_kernel<BM, BM, WM, WN, 2, SYMM, 2, GEMM_MODE_REDUCE, false, F16, TWO, ILV, DUAL>;
  for (const void* k : {reinterpret_cast<const void*>(kfull), reinterpret_cast<const void*>(kpart),
                        reinterpret_cast<const void*>(kred)})
    PVS_TRY(ensure_lds(ctx, k, Cfg::LDS_BYTES));
  if (plan.n_main > 0) {
    g.tile_base = 0;
    if constexpr (MODEL == 1) {
      // fp16 256 x 256: the full rounds run on the 8-phase schedule with v_mfma_f32_16x16x32_f16 (gemm_f16_8ph.hpp; measured
      // 1.45 PFLOP/s against 1.12-1.19 for the two-stage kernel, profiles/r03_fp16_gemm_8phase_control_*.txt); the split-K
      // tail of a partly filled last round stays on the two-stage kernel below (same tile list, same epilogue)
      auto k8 = gemm_f16_8ph_kernel<SYMM, true>;
      PVS_TRY(ensure_lds(ctx, reinterpret_cast<const void*>(k8), G8_LDS_BYTES));
      hipLaunchKernelGGL(k8, dim3((unsigned)plan.n_main), dim3(512), G8_LDS_BYTES, ctx->stream, g);
    } else {
      hipLaunchKernelGGL(kfull, dim3((unsigned)plan.n_main), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
    }
  }
  if (plan.n_tail > 0) {
    // partial images per tile: one per 1024-k chain (exact fp32 path) or one per slice (fp16 path)
    const int bk = F16 ? 64 : 32;
    const int nk = (int)((g.L + bk - 1) / bk);
    const int nparts = TWO ? (nk + GEMM_KBLOCK / bk - 1) / (GEMM_KBLOCK / bk) : plan.splitk;
    const size_t bytes = (size_t)plan.n_tail * nparts * BM * BM * sizeof(float);
    if (bytes > ((size_t)1 << 30)) {
      // very long rows (e.g. Fisher vectors): the chain images would not fit a sane workspace -> unsplit launch
      g.tile_base = plan.n_main;
      hipLaunchKernelGGL(kfull, dim3((unsigned)plan.n_tail), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
    } else {
      float* part = nullptr;
      PVS_TRY(ws_reserve(ctx, 4, bytes, reinterpret_cast<void**>(&part)));
      g.tile_base = plan.n_main;
      g.splitk = std::min(plan.splitk, nparts);
      g.nparts = nparts;
      g.partial = part;
      hipLaunchKernelGGL(kpart, dim3((unsigned)(plan.n_tail * g.splitk)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
      hipLaunchKernelGGL(kred, dim3((unsigned)plan.n_tail), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
    }
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// shared front end of the two MFMA paths
template <int MODEL>
static int cosine_mfma(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, const float* inva,
                       const float* invb, float* out, int64_t ldo, float* out_t = nullptr, int64_t ldt = 0, int64_t ld = 0,
                       int accumulate = 0) {
  constexpr int BT = GemmModel<MODEL>::BM;
  const int tiles_m = (int)((M + BT - 1) / BT), tiles_n = (int)((N + BT - 1) / BT);
  if ((int64_t)tiles_m * tiles_n > 0x3fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "cosine: too many tiles for one launch");
  // self-similarity: same operand, same norms -> only the upper triangle is computed, the rest mirrored
  const bool symm = (A == B) && (M == N) && (inva == invb) && out_t == nullptr;
  GemmPlan* plan = nullptr;
  if constexpr (MODEL == 2) {
    // the bounded-error prefilter on problems of several full rounds of 256 x 128 tiles (a query block against a corpus
    // panel): gemm_f16_2lvl.hpp, 1.13 against 0.93 PFLOP/s (profiles/r03_fp16_gemm_two_level_256x128.txt); whole tiles only, so
    // small problems and the symmetric case keep the 128 x 128 kernel with its split-K tail
    const int t256 = (int)((M + 255) / 256), t128 = (int)((N + 127) / 128);
    if (!symm && out_t == nullptr && (int64_t)t256 * t128 >= (int64_t)4 * ctx->num_cu && (int64_t)t256 * t128 <= 0x3fffffffLL &&
        L <= (int64_t)8 * 1024 * 1024) {
      PVS_TRY(build_plan(ctx, 4, t256, t128, false, ctx->num_cu, &plan));
      GemmArgs g{};
      g.A = A; g.B = B; g.M = M; g.N = N; g.L = L; g.lda = ld ? ld : L; g.ldb = ld ? ld : L; g.inva = inva; g.invb = invb;
      g.accumulate = accumulate;
      g.out = out; g.ldo = ldo; g.tiles = static_cast<const GemmTile*>(plan->d_tiles); g.splitk = 1; g.tile_base = 0;
      PVS_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(const_cast<float**>(&g.zero16)), HIP_SYMBOL(g_zero16)));
      auto k2 = gemm_f16_2lvl_kernel<false>;
      PVS_TRY(ensure_lds(ctx, reinterpret_cast<const void*>(k2), G2_LDS_BYTES));
      hipLaunchKernelGGL(k2, dim3((unsigned)(plan->n_main + plan->n_tail)), dim3(512), G2_LDS_BYTES, ctx->stream, g);
      PVS_HIP(hipGetLastError());
      return PVS_OK;
    }
  }
  PVS_TRY(build_plan(ctx, MODEL, tiles_m, tiles_n, symm, ctx->num_cu * GemmModel<MODEL>::PER_CU, &plan));
  GemmArgs g{};
  g.A = A; g.B = B; g.M = M; g.N = N; g.L = L; g.lda = ld ? ld : L; g.ldb = ld ? ld : L; g.inva = inva; g.invb = invb;
  g.accumulate = accumulate;
  g.out = out; g.ldo = ldo; g.out_t = out_t; g.ldt = ldt; g.tiles = static_cast<const GemmTile*>(plan->d_tiles); g.splitk = 1;
  PVS_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(const_cast<float**>(&g.zero16)), HIP_SYMBOL(g_zero16)));
  if (symm) return launch_gemm_mfma<true, MODEL>(ctx, g, *plan);
  if constexpr (MODEL == 0) {
    if (out_t) return launch_gemm_mfma<false, 0, true>(ctx, g, *plan);
  }
  return launch_gemm_mfma<false, MODEL>(ctx, g, *plan);
}

// ------------------------------------------------------------------------------------- fp32 -> fp16 encodings
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float4* __restrict__ src, int64_t n4, uint2* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    union { _Float16 h[4]; uint2 u; } o;
    o.h[0] = (_Float16)v.x; o.h[1] = (_Float16)v.y; o.h[2] = (_Float16)v.z; o.h[3] = (_Float16)v.w;  // round-to-nearest-even
    dst[i] = o.u;
  }
}
__global__ void f32_to_f16_tail_kernel(const float* __restrict__ src, int64_t start, int64_t n, _Float16* __restrict__ dst) {
  const int64_t i = start + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (_Float16)src[i];
}

int launch_f32_to_f16(pvs_ctx* ctx, const float* src, int64_t n, void* dst) {
  if (n <= 0) return PVS_OK;
  ScopedTimer tm(ctx, T_MISC);
  const bool vec = reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 8 == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  if (n4 > 0) {
    const unsigned grid = (unsigned)std::min<int64_t>((n4 + 255) / 256, (int64_t)ctx->num_cu * 16);
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const float4*>(src), n4,
                       reinterpret_cast<uint2*>(dst));
  }
  if (n4 * 4 < n)
    hipLaunchKernelGGL(f32_to_f16_tail_kernel, dim3((unsigned)((n - n4 * 4 + 255) / 256)), dim3(256), 0, ctx->stream, src,
                       n4 * 4, n, reinterpret_cast<_Float16*>(dst));
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_cosine_f16(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, const float* inva,
                      const float* invb, float* out, int64_t ldo) {
  if (M <= 0 || N <= 0) return PVS_OK;
  if (L <= 0 || L % 8 != 0 || reinterpret_cast<uintptr_t>(A) % 16 || reinterpret_cast<uintptr_t>(B) % 16)
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "fp16 cosine needs 16-B aligned rows (L %% 8 == 0), got L = %lld", (long long)L);
  if (L > (int64_t)8 * 1024 * 1024) PVS_FAIL(PVS_ERR_UNSUPPORTED, "fp16 cosine: L too large");
  ScopedTimer tm(ctx, T_GEMM);
  return cosine_mfma<1>(ctx, A, M, B, N, L, inva, invb, out, ldo);
}

// fp16 operands with the accumulation error bounded by chains of 1024 k (the prefilter of the exact filtered top-k)
int launch_cosine_f16_bounded(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, const float* inva,
                              const float* invb, float* out, int64_t ldo) {
  if (M <= 0 || N <= 0) return PVS_OK;
  if (L <= 0 || L % 8 != 0 || reinterpret_cast<uintptr_t>(A) % 16 || reinterpret_cast<uintptr_t>(B) % 16)
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "fp16 cosine needs 16-B aligned rows (L %% 8 == 0), got L = %lld", (long long)L);
  ScopedTimer tm(ctx, T_GEMM);
  // Rows longer than 32768 go through in segments accumulated in the output: over thousands of k-tiles the workgroups of an
  // XCD drift apart, their panels stop sharing L2 and one long launch runs at a tenth of the rate (measured at L = 262,400).
  constexpr int64_t SEG = 32768;
  const char* a = static_cast<const char*>(A);
  const char* b = static_cast<const char*>(B);
  for (int64_t k0 = 0; k0 < L; k0 += SEG)
    PVS_TRY(cosine_mfma<2>(ctx, a + k0 * 2, M, A == B ? a + k0 * 2 : b + k0 * 2, N, std::min(SEG, L - k0), inva, invb, out, ldo, nullptr, 0, L,
                           k0 > 0 ? 1 : 0));
  return PVS_OK;
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
      if (m < M && n < N) out[m * ldo + n] = acc[u][v] * ((inva ? inva[m] : T(1)) * (invb ? invb[n] : T(1)));   // sa*sb commutes: out[m][n] == out[n][m] bitwise for A == B
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
  // MFMA path: 16-B aligned rows, and per-lane byte offsets inside a 128-row tile must fit 32 bits
  const bool fast = (L % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (L <= (int64_t)8 * 1024 * 1024);
  ScopedTimer tm(ctx, T_GEMM);
  if (fast) {
    PVS_TRY(cosine_mfma<0>(ctx, A, M, B, N, L, inva, invb, out, ldo));
  } else {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(cosine_gemm_generic_kernel<float>, grid, dim3(256), 0, ctx->stream, A, M, B, N, L, inva, invb,
                       out, ldo);
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_cosine_f32_dual(pvs_ctx* ctx, const float* A, int64_t M, const float* B, int64_t N, int64_t L, const float* inva,
                           const float* invb, float* out, int64_t ldo, float* out_t, int64_t ldt) {
  if (M <= 0 || N <= 0) return PVS_OK;
  if (!((L % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0) &&
        (L <= (int64_t)8 * 1024 * 1024)))
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "dual-output cosine needs 16-B aligned rows (L %% 4 == 0)");
  ScopedTimer tm(ctx, T_GEMM);
  return cosine_mfma<0>(ctx, A, M, B, N, L, inva, invb, out, ldo, out_t, ldt);
}

// ------------------------------------------------------------------------------------- float64 on the f64 matrix pipe
int launch_row_inv_norms_f64(pvs_ctx* ctx, const double* d_x, int64_t rows, int64_t L, double* d_inv) {
  if (rows <= 0) return PVS_OK;
  ScopedTimer tm(ctx, T_MISC);
  hipLaunchKernelGGL(row_inv_norms_generic_kernel<double>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, d_x, rows, L,
                     d_inv);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

template <bool SYMM>
static int launch_gemm_f64(pvs_ctx* ctx, GemmArgsF64 g, const GemmPlan& plan) {
  using Cfg = GemmCfgF64;
  auto kfull = gemm_f64_kernel<SYMM, GEMM_MODE_FULL>;
  auto kpart = gemm_f64_kernel<SYMM, GEMM_MODE_PARTIAL>;
  auto kred = gemm_f64_kernel<SYMM, GEMM_MODE_REDUCE>;
  for (const void* k : {reinterpret_cast<const void*>(kfull), reinterpret_cast<const void*>(kpart), reinterpret_cast<const void*>(kred)})
    PVS_TRY(ensure_lds(ctx, k, Cfg::LDS_BYTES));
  if (plan.n_main > 0) {
    g.tile_base = 0;
    hipLaunchKernelGGL(kfull, dim3((unsigned)plan.n_main), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
  }
  if (plan.n_tail > 0) {
    const int nk = (int)((g.L + Cfg::BK - 1) / Cfg::BK);
    const int sk = std::max(1, std::min(plan.splitk, nk));
    const size_t bytes = (size_t)plan.n_tail * sk * Cfg::PART_ELEMS * sizeof(double);
    g.tile_base = plan.n_main;
    if (sk == 1 || bytes > ((size_t)1 << 30)) {
      hipLaunchKernelGGL(kfull, dim3((unsigned)plan.n_tail), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
    } else {
      double* part = nullptr;
      PVS_TRY(ws_reserve(ctx, 4, bytes, reinterpret_cast<void**>(&part)));
      g.splitk = sk;
      g.nparts = sk;
      g.partial = part;
      hipLaunchKernelGGL(kpart, dim3((unsigned)(plan.n_tail * sk)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
      hipLaunchKernelGGL(kred, dim3((unsigned)plan.n_tail), dim3(Cfg::THREADS), Cfg::LDS_BYTES, ctx->stream, g);
    }
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

// out[m*ldo + n] = (A_m . B_n) * inva[m] * invb[n], all float64, device pointers; inva / invb may be null (= 1)
int launch_cosine_f64_dev(pvs_ctx* ctx, const double* A, int64_t M, const double* B, int64_t N, int64_t L, const double* inva,
                          const double* invb, double* out, int64_t ldo) {
  if (M <= 0 || N <= 0) return PVS_OK;
  if (L <= 0) PVS_FAIL(PVS_ERR_INVALID, "cosine: L must be positive");
  // MFMA path: 16-B aligned rows (even L), per-lane byte offsets inside a 128-row tile must fit 32 bits
  const bool fast = (L % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0) &&
                    (L <= (int64_t)4 * 1024 * 1024);
  ScopedTimer tm(ctx, T_GEMM);
  if (!fast) {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(cosine_gemm_generic_kernel<double>, grid, dim3(256), 0, ctx->stream, A, M, B, N, L, inva, invb, out, ldo);
    PVS_HIP(hipGetLastError());
    return PVS_OK;
  }
  const int tiles_m = (int)((M + 127) / 128), tiles_n = (int)((N + 127) / 128);
  if ((int64_t)tiles_m * tiles_n > 0x3fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "cosine: too many tiles for one launch");
  const bool symm = (A == B) && (M == N) && (inva == invb);
  GemmPlan* plan = nullptr;
  PVS_TRY(build_plan(ctx, 3, tiles_m, tiles_n, symm, ctx->num_cu * 2, &plan));
  GemmArgsF64 g{};
  g.A = A; g.B = B; g.M = M; g.N = N; g.L = L; g.lda = L; g.ldb = L; g.inva = inva; g.invb = invb; g.out = out; g.ldo = ldo;
  g.tiles = static_cast<const GemmTile*>(plan->d_tiles);
  g.splitk = 1;
  PVS_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(const_cast<float**>(&g.zero16)), HIP_SYMBOL(g_zero16)));
  return symm ? launch_gemm_f64<true>(ctx, g, *plan) : launch_gemm_f64<false>(ctx, g, *plan);
}

// host-API form: norms of both operands (workspace slot 1), then the GEMM; out is M x N, ld = N
int launch_cosine_f64(pvs_ctx* ctx, const double* A, int64_t M, const double* B, int64_t N, int64_t L, double* out) {
  if (M <= 0 || N <= 0) return PVS_OK;
  double* inv = nullptr;
  const bool same = (A == B && M == N);
  PVS_TRY(ws_reserve(ctx, 1, (size_t)(M + N) * sizeof(double), reinterpret_cast<void**>(&inv)));
  PVS_TRY(launch_row_inv_norms_f64(ctx, A, M, L, inv));
  if (!same) PVS_TRY(launch_row_inv_norms_f64(ctx, B, N, L, inv + M));
  return launch_cosine_f64_dev(ctx, A, M, B, N, L, inv, same ? inv : inv + M, out, N);
}

}  // namespace pvs
