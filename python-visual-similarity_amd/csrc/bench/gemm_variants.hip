// Variant harness for the exact-fp32 cosine GEMM (not part of libpvsim_hip.so).
//   hipcc -O3 --offload-arch=gfx950 -o gemm_variants gemm_variants.hip && ./gemm_variants [N] [L]
// Random normal operands (never zeros: zero operands raise the clock and read high), A == B (self-similarity),
// every variant checked against fp64 dot products on sampled entries, 1 warm-up + 3 timed launches each.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <cstring>

#define PVS_GEMM_DBG 1
#include "../gemm_mfma.hpp"
#include "../gemm_f16_8ph.hpp"
#include "../gemm_f16_2lvl.hpp"

using namespace pvs;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_normal(float* x, int64_t n, uint64_t seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
  const float u1 = ((z >> 40) + 1) * (1.0f / 16777217.0f), u2 = ((z >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
  // sparse like a VLAD row: ~40 % exact zeros
  x[i] = (z & 7) < 3 ? 0.f : sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.01f;
}

__global__ void inv_norms(const float* x, int64_t rows, int64_t L, int64_t ld, float* inv) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int64_t i = lane; i < L; i += 64) s = fmaf(x[row * ld + i], x[row * ld + i], s);
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) inv[row] = s > 0.f ? 1.f / sqrtf(s) : 1.f;
}

__global__ void ref_samples(const float* x, const float* inv, int64_t L, int64_t ld, const int* sm, const int* sn, int ns, double* out) {
  const int s = blockIdx.x;
  if (s >= ns) return;
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (int64_t i = lane; i < L; i += 64) acc += (double)x[(int64_t)sm[s] * ld + i] * (double)x[(int64_t)sn[s] * ld + i];
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) out[s] = acc * (double)inv[sm[s]] * (double)inv[sn[s]];
}

__global__ void to_half(const float* x, _Float16* y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (_Float16)x[i];
}
__global__ void ref_samples_h(const _Float16* x, const float* inv, int64_t L, int64_t ld, const int* sm, const int* sn,
                              int ns, double* out) {
  const int s = blockIdx.x;
  if (s >= ns) return;
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (int64_t i = lane; i < L; i += 64) acc += (double)(float)x[(int64_t)sm[s] * ld + i] * (double)(float)x[(int64_t)sn[s] * ld + i];
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) out[s] = acc * (double)inv[sm[s]] * (double)inv[sn[s]];
}

// [row block of 128][k-tile][128 rows x 64 halfs, 16-B chunk g of row r at chunk g ^ ((r >> 1) & 7)], zero padded
__global__ void to_tiled(const _Float16* x, int64_t rows, int64_t L, int64_t ld, _Float16* t, int64_t nk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-B chunk of the tiled array per thread
  const int64_t nchunks = ((rows + 127) / 128) * nk * 128 * 8;
  if (i >= nchunks) return;
  const int p = (int)(i & 7), rr = (int)((i >> 3) & 127);
  const int64_t bt = i >> 10, kt = bt % nk, blk = bt / nk;
  const int g = p ^ ((rr >> 1) & 7);
  const int64_t r = blk * 128 + rr, k = kt * 64 + 8 * g;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (r < rows && k + 8 <= L) v = *reinterpret_cast<const uint4*>(x + r * ld + k);
  reinterpret_cast<uint4*>(t)[i] = v;
}

__device__ __attribute__((aligned(16))) float d_zero16[4] = {0, 0, 0, 0};
static int64_t g_ld = 0;  // operand row stride (floats)
static int g_dbg = 0;     // ablation bits for stamped builds

static std::vector<GemmTile> tile_list(int tm_n, int tn_n, bool symm) {
  std::vector<GemmTile> t;
  if (symm) {
    const int TS = (tm_n + 7) / 8;
    for (int si = 0; si < TS; ++si)
      for (int sj = si; sj < TS; ++sj)
        for (int w = 0; w < 64; ++w) {
          const int tm = si * 8 + (w & 7), tn = sj * 8 + (w >> 3);
          if (tm < tm_n && tn < tm_n && tn >= tm) t.push_back({tm, tn});
        }
  } else {
    for (int g0 = 0; g0 < tm_n; g0 += 8)
      for (int tn = 0; tn < tn_n; ++tn)
        for (int dm = 0; dm < std::min(8, tm_n - g0); ++dm) t.push_back({g0 + dm, tn});
  }
  return t;
}

// TAILK > 1: the tiles of the last partial round (slots = 512) go through the split-K tail
template <int BM, int BN, int WM, int WN, int STAGES, bool SYMM, int OCC, bool STAMP = false, bool F16 = false,
          bool TWO = true, bool ILV = false, int LW = WM * WN, bool PP = false>
static void run(const char* name, const void* A, const float* inv, int64_t N, int64_t L, float* out, int ns,
                const double* ref_h, const int* sm_h, const int* sn_h, int tailk) {
  using Cfg = GemmCfg<BM, BN, WM, WN, STAGES, F16, LW>;
  constexpr int LDSB = PP ? RING_LDS_BYTES : Cfg::LDS_BYTES;
  GemmArgs g{};
  g.A = A; g.B = A; g.M = N; g.N = N; g.L = L; g.lda = g_ld; g.ldb = g_ld; g.inva = inv; g.invb = inv; g.out = out;
  g.ldo = N; g.splitk = 1; g.dbg = g_dbg;
  CK(hipGetSymbolAddress((void**)&g.zero16, HIP_SYMBOL(d_zero16)));
  std::vector<GemmTile> t = tile_list((int)((N + BM - 1) / BM), (int)((N + BN - 1) / BN), SYMM);
  GemmTile* d_t; CK(hipMalloc(&d_t, t.size() * sizeof(GemmTile)));
  CK(hipMemcpy(d_t, t.data(), t.size() * sizeof(GemmTile), hipMemcpyHostToDevice));
  g.tiles = d_t;
  const int slots = 256 * (OCC * 256 / Cfg::THREADS), total = (int)t.size();
  int n_main = total, n_tail = 0;
  if (tailk > 1 && total > slots && total % slots) { n_tail = total % slots; n_main = total - n_tail; }
  float* part = nullptr;
  if (n_tail) CK(hipMalloc(&part, (size_t)n_tail * (TWO ? (size_t)((L + 1023) / 1024) : (size_t)tailk) * BM * BN * 4));
  auto kf = gemm_mfma_kernel<BM, BN, WM, WN, STAGES, SYMM, OCC, GEMM_MODE_FULL, STAMP, F16, TWO, ILV, false, LW, PP>;
  auto kp = gemm_mfma_kernel<BM, BN, WM, WN, STAGES, SYMM, OCC, GEMM_MODE_PARTIAL, false, F16, TWO, ILV, false, LW, PP>;
  auto kr = gemm_mfma_kernel<BM, BN, WM, WN, STAGES, SYMM, OCC, GEMM_MODE_REDUCE, false, F16, TWO, ILV, false, LW, PP>;
  for (const void* k : {(const void*)kf, (const void*)kp, (const void*)kr})
    CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  if (STAMP) CK(hipMalloc(&g.stamps, (size_t)n_main * 64));
  auto launch = [&]() {
    GemmArgs a = g;
    a.tile_base = 0;
    hipLaunchKernelGGL(kf, dim3((unsigned)n_main), dim3(Cfg::THREADS), LDSB, 0, a);
    if (n_tail) {
      a.tile_base = n_main; a.splitk = tailk; a.nparts = TWO ? (int)((L + 1023) / 1024) : tailk; a.partial = part;
      hipLaunchKernelGGL(kp, dim3((unsigned)(n_tail * tailk)), dim3(Cfg::THREADS), LDSB, 0, a);
      hipLaunchKernelGGL(kr, dim3((unsigned)n_tail), dim3(Cfg::THREADS), LDSB, 0, a);
    }
  };
  CK(hipMemset(out, 0xff, (size_t)N * N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms); sum += ms;
  }
  double maxerr = 0.0;
  for (int s = 0; s < ns; ++s) {
    float a, b;
    CK(hipMemcpy(&a, out + (int64_t)sm_h[s] * N + sn_h[s], 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&b, out + (int64_t)sn_h[s] * N + sm_h[s], 4, hipMemcpyDeviceToHost));
    maxerr = fmax(maxerr, fmax(fabs((double)a - ref_h[s]), fabs((double)b - ref_h[s])));
    if (a != b) maxerr = fmax(maxerr, 1.0);  // symmetric inputs must give bitwise symmetric outputs
  }
  const double flop_alg = 2.0 * (double)N * (double)N * (double)L, flop_exec = 2.0 * BM * BN * (double)L * total;
  printf("%-26s tiles %5d (+%d x splitK %d)  avg %8.3f ms  best %8.3f  executed %7.2f TF/s  algorithmic %7.2f TF/s  maxerr %.2e %s\n",
         name, n_main, n_tail, n_tail ? tailk : 1, sum / 3, best, flop_exec / (sum / 3 * 1e-3) / 1e12,
         flop_alg / (sum / 3 * 1e-3) / 1e12, maxerr, maxerr < (F16 ? 2e-5 : 2e-6) ? "ok" : "FAIL");
  if (STAMP) {
    std::vector<unsigned long long> h((size_t)n_main * 8);
    CK(hipMemcpy(h.data(), g.stamps, (size_t)n_main * 64, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc; double seg[4] = {0, 0, 0, 0};
    for (int b = 0; b < n_main; ++b) {
      const double dt = (double)(h[8 * b + 1] - h[8 * b]), dr = (double)(h[8 * b + 3] - h[8 * b + 2]);
      if (dr > 0) { clk.push_back(dt / dr * 0.1); cyc.push_back(dt); }
      for (int q = 0; q < 4; ++q) seg[q] += (double)h[8 * b + 4 + q];
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double nkt = (double)n_main * (double)((L + Cfg::BK - 1) / Cfg::BK);
    printf("   [stamped build] clock %.3f GHz; block loop %.1f cyc per k-tile; per k-tile: seg0 %.0f seg1 %.0f"
           " seg2 %.0f seg3 %.0f  (k-loop: vmcnt / barrier / issue / reads+MFMA;  ping-pong: reads / barrier / lgkm+MFMA / barrier)\n", clk[clk.size() / 2], cyc[cyc.size() / 2] / (double)((L + Cfg::BK - 1) / Cfg::BK), seg[0] / nkt,
           seg[1] / nkt, seg[2] / nkt, seg[3] / nkt);
    CK(hipFree(g.stamps));
  }
  fflush(stdout);
  CK(hipFree(d_t)); if (part) CK(hipFree(part));
}

// the 8-phase 256 x 256 fp16 kernel (gemm_f16_8ph.hpp); general tile order or symmetric (upper triangle + mirror)
template <bool SYMM, bool M16, bool STAMP = false, bool TILED = false>
static float run8(const char* name, const void* A, const float* inv, int64_t N, int64_t L, float* out, int ns,
                  const double* ref_h, const int* sm_h, const int* sn_h, int reps = 3) {
  GemmArgs g{};
  g.A = A; g.B = A; g.M = N; g.N = N; g.L = L; g.lda = g_ld; g.ldb = g_ld; g.inva = inv; g.invb = inv; g.out = out;
  g.ldo = N; g.splitk = 1;
  CK(hipGetSymbolAddress((void**)&g.zero16, HIP_SYMBOL(d_zero16)));
  std::vector<GemmTile> t = tile_list((int)((N + 255) / 256), (int)((N + 255) / 256), SYMM);
  GemmTile* d_t; CK(hipMalloc(&d_t, t.size() * sizeof(GemmTile)));
  CK(hipMemcpy(d_t, t.data(), t.size() * sizeof(GemmTile), hipMemcpyHostToDevice));
  g.tiles = d_t;
  const int total = (int)t.size();
  auto kf = gemm_f16_8ph_kernel<SYMM, M16, STAMP, TILED>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_BYTES));
  if (STAMP) CK(hipMalloc(&g.stamps, (size_t)total * 64));
  auto launch = [&]() { hipLaunchKernelGGL(kf, dim3((unsigned)total), dim3(512), G8_LDS_BYTES, 0, g); };
  CK(hipMemset(out, 0xff, (size_t)N * N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int it = 0; it < reps; ++it) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms); sum += ms;
  }
  double maxerr = 0.0;
  for (int s = 0; s < ns; ++s) {
    float a, b;
    CK(hipMemcpy(&a, out + (int64_t)sm_h[s] * N + sn_h[s], 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&b, out + (int64_t)sn_h[s] * N + sm_h[s], 4, hipMemcpyDeviceToHost));
    maxerr = fmax(maxerr, fmax(fabs((double)a - ref_h[s]), fabs((double)b - ref_h[s])));
    if (a != b) maxerr = fmax(maxerr, 1.0);
  }
  const double flop_alg = 2.0 * (double)N * (double)N * (double)L, flop_exec = 2.0 * 256 * 256 * (double)L * total;
  printf("%-30s tiles %5d  avg %8.3f ms  best %8.3f  executed %7.2f TF/s  algorithmic %7.2f TF/s  maxerr %.2e %s\n", name, total,
         sum / reps, best, flop_exec / (sum / reps * 1e-3) / 1e12, flop_alg / (sum / reps * 1e-3) / 1e12, maxerr, maxerr < 2e-5 ? "ok" : "FAIL");
  if (STAMP) {
    std::vector<unsigned long long> h((size_t)total * 8);
    CK(hipMemcpy(h.data(), g.stamps, (size_t)total * 64, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < total; ++b) {
      const double dt = (double)(h[8 * b + 1] - h[8 * b]), dr = (double)(h[8 * b + 3] - h[8 * b + 2]);
      if (dr > 0) { clk.push_back(dt / dr * 0.1); cyc.push_back(dt); }
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    printf("   [stamped build] in-kernel clock %.3f GHz (median over workgroups); k-loop %.1f cycles per k-tile (MFMA floor 1024)\n",
           clk[clk.size() / 2], cyc[cyc.size() / 2] / (double)((L + 63) / 64));
    CK(hipFree(g.stamps));
  }
  fflush(stdout);
  CK(hipFree(d_t));
  return sum / reps;
}

// the two-level 256 x 128 kernel (gemm_f16_2lvl.hpp): general tile order
template <bool STAMP = false>
static float run2(const char* name, const void* A, const float* inv, int64_t N, int64_t L, float* out, int ns,
                  const double* ref_h, const int* sm_h, const int* sn_h, int reps = 3) {
  GemmArgs g{};
  g.A = A; g.B = A; g.M = N; g.N = N; g.L = L; g.lda = g_ld; g.ldb = g_ld; g.inva = inv; g.invb = inv; g.out = out;
  g.ldo = N; g.splitk = 1;
  CK(hipGetSymbolAddress((void**)&g.zero16, HIP_SYMBOL(d_zero16)));
  std::vector<GemmTile> t = tile_list((int)((N + 255) / 256), (int)((N + 127) / 128), false);
  GemmTile* d_t; CK(hipMalloc(&d_t, t.size() * sizeof(GemmTile)));
  CK(hipMemcpy(d_t, t.data(), t.size() * sizeof(GemmTile), hipMemcpyHostToDevice));
  g.tiles = d_t;
  const int total = (int)t.size();
  auto kf = gemm_f16_2lvl_kernel<STAMP>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
  if (STAMP) CK(hipMalloc(&g.stamps, (size_t)total * 64));
  auto launch = [&]() { hipLaunchKernelGGL(kf, dim3((unsigned)total), dim3(512), G2_LDS_BYTES, 0, g); };
  CK(hipMemset(out, 0xff, (size_t)N * N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int it = 0; it < reps; ++it) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms); sum += ms;
  }
  double maxerr = 0.0;
  for (int s = 0; s < ns; ++s) {
    float a, b;
    CK(hipMemcpy(&a, out + (int64_t)sm_h[s] * N + sn_h[s], 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&b, out + (int64_t)sn_h[s] * N + sm_h[s], 4, hipMemcpyDeviceToHost));
    maxerr = fmax(maxerr, fmax(fabs((double)a - ref_h[s]), fabs((double)b - ref_h[s])));
  }
  const double flop = 2.0 * 256 * 128 * (double)L * total;
  printf("%-30s tiles %5d  avg %8.3f ms  best %8.3f  executed %7.2f TF/s  maxerr %.2e %s\n", name, total, sum / reps, best,
         flop / (sum / reps * 1e-3) / 1e12, maxerr, maxerr < 2e-5 ? "ok" : "FAIL");
  if (STAMP) {
    std::vector<unsigned long long> h((size_t)total * 8);
    CK(hipMemcpy(h.data(), g.stamps, (size_t)total * 64, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < total; ++b) {
      const double dt = (double)(h[8 * b + 1] - h[8 * b]), dr = (double)(h[8 * b + 3] - h[8 * b + 2]);
      if (dr > 0) { clk.push_back(dt / dr * 0.1); cyc.push_back(dt); }
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    printf("   [stamped build] in-kernel clock %.3f GHz (median over workgroups); k-loop %.1f cycles per k-tile (MFMA floor 512 per wave, 1024 per SIMD)\n",
           clk[clk.size() / 2], cyc[cyc.size() / 2] / (double)((L + 63) / 64));
    CK(hipFree(g.stamps));
  }
  fflush(stdout);
  CK(hipFree(d_t));
  return sum / reps;
}

// Which scalar recurrence reproduces the f32 MFMA accumulation bit for bit?  (needed by an exact re-scoring kernel)
static void chain_check(const float* A, const float* inv, const float* out, int64_t N, int64_t L, const int* sm, const int* sn, int ns) {
  std::vector<float> ra(L), rb(L);
  int match[6] = {0, 0, 0, 0, 0, 0};
  for (int s = 0; s < ns; ++s) {
    const int i = sm[s], j = sn[s];
    float si, sj, got;
    CK(hipMemcpy(ra.data(), A + (int64_t)i * g_ld, L * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rb.data(), A + (int64_t)j * g_ld, L * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&si, inv + i, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&sj, inv + j, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&got, out + (int64_t)i * N + j, 4, hipMemcpyDeviceToHost));
    float res[6];
    for (int v = 0; v < 6; ++v) {
      float tot = 0.f;
      for (int64_t c0 = 0; c0 < L; c0 += 1024) {
        float acc = 0.f;
        for (int64_t b8 = c0; b8 < c0 + 1024 && b8 < L; b8 += 8)
          for (int e = 0; e < 4; ++e) {
            const int64_t k1 = b8 + e, k2 = b8 + 4 + e;
            const float a1 = ra[k1], b1 = rb[k1], a2 = ra[k2], b2 = rb[k2];
            switch (v) {
              case 0: acc = fmaf(a1, b1, acc); acc = fmaf(a2, b2, acc); break;          // h = 0 then h = 1, fused
              case 1: acc = fmaf(a2, b2, acc); acc = fmaf(a1, b1, acc); break;          // h = 1 then h = 0
              case 2: acc = acc + fmaf(a2, b2, a1 * b1); break;                          // dot2 then add
              case 3: acc = acc + (a1 * b1); acc = acc + (a2 * b2); break;               // unfused
              case 4: acc = (float)((double)acc + (double)a1 * b1 + (double)a2 * b2); break;   // exact dot2, one rounding
              case 5: acc = fmaf(a1, b1, fmaf(a2, b2, 0.f)) + acc; break;
            }
          }
        tot += acc;
      }
      res[v] = tot * (si * sj);
      match[v] += (res[v] == got);
    }
    if (s < 4) printf("  sample (%d,%d): device %.9g  v0 %.9g v1 %.9g v2 %.9g v3 %.9g v4 %.9g v5 %.9g\n", i, j, got, res[0], res[1], res[2], res[3], res[4], res[5]);
  }
  printf("chain check over %d samples: bitwise matches v0 %d v1 %d v2 %d v3 %d v4 %d v5 %d\n", ns, match[0], match[1], match[2], match[3], match[4], match[5]);
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 8189, L = argc > 2 ? atoll(argv[2]) : 32768;
  g_ld = L + (argc > 3 ? atoll(argv[3]) : 0);
  float *A, *inv, *out;
  CK(hipMalloc(&A, (size_t)N * g_ld * 4)); CK(hipMalloc(&inv, (size_t)N * 4)); CK(hipMalloc(&out, (size_t)N * N * 4));
  hipLaunchKernelGGL(fill_normal, dim3((unsigned)(((int64_t)N * g_ld + 255) / 256)), dim3(256), 0, 0, A, N * g_ld, 1234ull);
  hipLaunchKernelGGL(inv_norms, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, 0, A, N, L, g_ld, inv);
  const int ns = 48;
  std::vector<int> sm(ns), sn(ns);
  srand(7);
  for (int s = 0; s < ns; ++s) { sm[s] = rand() % N; sn[s] = s < 8 ? sm[s] : rand() % N; }
  sm[8] = 0; sn[8] = (int)N - 1; sm[9] = (int)N - 1; sn[9] = (int)N - 1; sm[10] = 127; sn[10] = 128; sm[11] = 255; sn[11] = 256;
  int *d_sm, *d_sn; double* d_ref;
  CK(hipMalloc(&d_sm, ns * 4)); CK(hipMalloc(&d_sn, ns * 4)); CK(hipMalloc(&d_ref, ns * 8));
  CK(hipMemcpy(d_sm, sm.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sn, sn.data(), ns * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ref_samples, dim3(ns), dim3(64), 0, 0, A, inv, L, g_ld, d_sm, d_sn, ns, d_ref);
  std::vector<double> ref(ns);
  CK(hipMemcpy(ref.data(), d_ref, ns * 8, hipMemcpyDeviceToHost));
  printf("N=%lld L=%lld ld=%lld\n", (long long)N, (long long)L, (long long)g_ld);
#define RUN(BM, BN, WM, WN, ST, SY, OCC, TK) run<BM, BN, WM, WN, ST, SY, OCC>(#BM "x" #BN " w" #WM "x" #WN " st" #ST " symm" #SY, A, inv, N, L, out, ns, ref.data(), sm.data(), sn.data(), TK)
#define RUNH(BM, BN, WM, WN, ST, SY, OCC, ILV, TK) run<BM, BN, WM, WN, ST, SY, OCC, false, true, false, ILV>("f16 " #BM "x" #BN " w" #WM "x" #WN " st" #ST " symm" #SY " ilv" #ILV, A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), TK)
  const int which = argc > 4 ? atoi(argv[4]) : 0;
  if (which == 0 || which == 1) {
    RUN(128, 128, 2, 2, 2, false, 2, 1);
    if (argc > 5 && !strcmp(argv[5], "chain")) { RUN(128, 128, 2, 2, 2, false, 2, 1); chain_check(A, inv, out, N, L, sm.data(), sn.data(), ns); return 0; }
    run<128, 128, 2, 2, 2, false, 2, true>("128x128 stamped", A, inv, N, L, out, ns, ref.data(), sm.data(), sn.data(), 1);
    RUN(128, 128, 2, 2, 2, true, 2, 16);
  }
  if (which == 0 || which == 2) {
    _Float16* A16; CK(hipMalloc(&A16, (size_t)N * g_ld * 2));
    hipLaunchKernelGGL(to_half, dim3((unsigned)(((int64_t)N * g_ld + 255) / 256)), dim3(256), 0, 0, A, A16, N * g_ld);
    hipLaunchKernelGGL(ref_samples_h, dim3(ns), dim3(64), 0, 0, A16, inv, L, g_ld, d_sm, d_sn, ns, d_ref);
    std::vector<double> refh(ns);
    CK(hipMemcpy(refh.data(), d_ref, ns * 8, hipMemcpyDeviceToHost));
    if (which == 2 && argc > 5) {   // ping-pong A/B, interleaved rounds in one process
      for (int round = 0; round < atoi(argv[5]); ++round) {
        RUNH(256, 256, 2, 4, 2, false, 2, true, 1);
        run<256, 256, 2, 4, 2, false, 2, false, true, false, false, 8, true>("f16 256x256 ping-pong", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
        run<256, 256, 2, 4, 2, true, 2, false, true, false, false, 8, true>("f16 256x256 ping-pong symm", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 8);
        RUNH(256, 256, 2, 4, 2, true, 2, true, 8);
      }
      for (int d : {0, 2, 6}) {
        g_dbg = d;
        char nm[64]; snprintf(nm, sizeof nm, "f16 ping-pong dbg%d", d);
        run<256, 256, 2, 4, 2, false, 2, false, true, false, false, 8, true>(nm, A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
        snprintf(nm, sizeof nm, "f16 ping-pong stamped dbg%d", d);
        run<256, 256, 2, 4, 2, false, 2, true, true, false, false, 8, true>(nm, A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
      }
      g_dbg = 0;
      run<256, 256, 2, 4, 2, false, 2, true, true, false, true, 8, false>("f16 256x256 ilv stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
      return 0;
    }
    RUNH(256, 256, 2, 4, 2, false, 2, false, 1);
    run<256, 256, 2, 4, 2, false, 2, false, true, false, false, 4>("f16 256x256 LW4", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
    run<256, 256, 2, 4, 2, false, 2, true, true, false, false, 4>("f16 256x256 LW4 stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
    run<256, 256, 2, 4, 2, true, 2, false, true, false, false, 4>("f16 256x256 LW4 symm", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 8);
    run<256, 256, 2, 4, 2, false, 2, false, true, false, true, 4>("f16 256x256 LW4 ilv", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
    RUNH(256, 256, 2, 4, 2, false, 2, true, 1);
    run<256, 256, 2, 4, 2, false, 2, true, true, false, false>("f16 256x256 stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
    RUNH(128, 128, 2, 2, 2, false, 2, false, 1);
    RUNH(128, 128, 2, 2, 2, false, 2, true, 1);
    RUNH(256, 256, 2, 4, 2, true, 2, false, 8);
    RUNH(256, 256, 2, 4, 2, true, 2, true, 8);
  }
  if (which == 3) {   // the 8-phase control (guide's template) beside the shipped schedules, interleaved rounds in ONE process
    _Float16* A16; CK(hipMalloc(&A16, (size_t)N * g_ld * 2));
    hipLaunchKernelGGL(to_half, dim3((unsigned)(((int64_t)N * g_ld + 255) / 256)), dim3(256), 0, 0, A, A16, N * g_ld);
    hipLaunchKernelGGL(ref_samples_h, dim3(ns), dim3(64), 0, 0, A16, inv, L, g_ld, d_sm, d_sn, ns, d_ref);
    std::vector<double> refh(ns);
    CK(hipMemcpy(refh.data(), d_ref, ns * 8, hipMemcpyDeviceToHost));
    const int rounds = argc > 5 ? atoi(argv[5]) : 3;
    const int64_t nk_t = (L + 63) / 64, tiled_halfs = ((N + 127) / 128) * nk_t * 128 * 64;
    _Float16* T16; CK(hipMalloc(&T16, (size_t)tiled_halfs * 2));
    hipLaunchKernelGGL(to_tiled, dim3((unsigned)((tiled_halfs / 8 + 255) / 256)), dim3(256), 0, 0, A16, N, L, g_ld, T16, nk_t);
    CK(hipDeviceSynchronize());
    for (int round = 0; round < rounds; ++round) {
      RUNH(256, 256, 2, 4, 2, false, 2, true, 1);
      run<256, 256, 2, 4, 2, false, 2, false, true, false, false, 8, true>("f16 256x256 ping-pong", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
      run8<false, true>("f16 8-phase 16x16x32", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
      run8<false, false>("f16 8-phase 32x32x16", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
      RUNH(256, 256, 2, 4, 2, true, 2, false, 8);
      run8<true, true>("f16 8-phase 16x16x32 symm", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
      run8<true, false>("f16 8-phase 32x32x16 symm", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
      run8<false, true, false, true>("f16 8-phase 16x16x32 TILED", T16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
    }
    run8<false, true, true, true>("f16 8-phase 16x16x32 TILED stamped", T16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
    run8<false, true, true>("f16 8-phase 16x16x32 stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
    run8<false, false, true>("f16 8-phase 32x32x16 stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
    run<256, 256, 2, 4, 2, false, 2, true, true, false, true>("f16 256x256 shipped stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
  }
  if (which == 4) {   // the two-level (chains of 1024 k) prefilter GEMM: 128 x 128 two-stage kernel vs the 256 x 128 phase-scheduled one
    _Float16* A16; CK(hipMalloc(&A16, (size_t)N * g_ld * 2));
    hipLaunchKernelGGL(to_half, dim3((unsigned)(((int64_t)N * g_ld + 255) / 256)), dim3(256), 0, 0, A, A16, N * g_ld);
    hipLaunchKernelGGL(ref_samples_h, dim3(ns), dim3(64), 0, 0, A16, inv, L, g_ld, d_sm, d_sn, ns, d_ref);
    std::vector<double> refh(ns);
    CK(hipMemcpy(refh.data(), d_ref, ns * 8, hipMemcpyDeviceToHost));
    const int rounds = argc > 5 ? atoi(argv[5]) : 3;
    for (int round = 0; round < rounds; ++round) {
      run<128, 128, 2, 2, 2, false, 2, false, true, true, false>("f16 128x128 two-level (shipped)", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data(), 1);
      run2<false>("f16 256x128 two-level phases", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
      run8<false, true>("f16 8-phase 16x16x32 (1 level)", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
    }
    run2<true>("f16 256x128 two-level stamped", A16, inv, N, L, out, ns, refh.data(), sm.data(), sn.data());
  }
  return 0;
}
