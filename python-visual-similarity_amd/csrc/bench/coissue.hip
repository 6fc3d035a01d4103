// Microbenchmark: do VALU instructions issue under a running MFMA on gfx950?
//   mode 0: MFMAs only (32x32x16 f16, 4 independent accumulators)   mode 1: VALU only (fma chains)
//   mode 2: one MFMA followed by NV independent VALU fmas, same totals as mode 0 + mode 1
//   mode 3: wave-specialised: even waves run mode 0, odd waves mode 1 (needs >= 2 waves per SIMD)
// Build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue ; run: ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP>
__device__ __forceinline__ float vop(float x, float seed) {
  if (OP == 0) return __builtin_fmaf(x, 1.0001f, seed);
  if (OP == 1) return __builtin_amdgcn_fmed3f(x, seed, 3.f);
  if (OP == 2) return __int_as_float(__float_as_int(x) + 12345);
  if (OP == 3) return (float)(_Float16)x + 0.f;      // cvt + cvt
  if (OP == 5) return __builtin_fmaxf(x, seed);       // v_max_f32
  if (OP == 6) return __int_as_float((__float_as_int(x) & ~31) | (__float_as_int(seed) & 31) | 3);   // v_and_or_b32 class
  if (OP == 7) {                                       // the packed-key scan step: and_or, max, med3
    const float key = __int_as_float((__float_as_int(x) & ~31) | 5);
    const float b = __builtin_fmaxf(seed, key);
    return __builtin_amdgcn_fmed3f(seed, b, key);
  }
  if (OP == 8) return x < seed ? seed : x;             // cmp + cndmask
  return x < seed ? seed : x + 1.f;                   // cmp + cndmask + add
}
template <int MODE, int NV, int OP>
__global__ void k(float* out, int iters, float seed, long long* clk) {
  const long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f16x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(seed + threadIdx.x); b[q] = (_Float16)(seed * 0.5f); }
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = seed + q;
  const int wave = threadIdx.x >> 6;
  constexpr bool do_m = MODE == 0 || MODE == 2;
  constexpr bool do_v = MODE == 1 || MODE == 2;
  if (MODE == 3) {      // wave-specialised, each role in its own loop
    if ((wave & 4) == 0) {
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    } else {
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int t = 0; t < 4 * NV; ++t) v[t % 8] = vop<OP>(v[t % 8], seed);
    }
  }
  for (int i = 0; i < (MODE == 3 ? 0 : iters); ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (do_m) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
      if (do_v) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q % 8] = vop<OP>(v[q % 8], seed);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int MODE, int NV, int OP = 0>
static void run(const char* name, int threads, int iters) {
  float* out;
  hipMalloc(&out, 1024 * 1024 * 4);
  long long* clk; hipMalloc(&clk, 16); long long h[2];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NV, OP><<<256, threads>>>(out, 10, 1.f, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NV, OP><<<256, threads>>>(out, iters, 1.f, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: waves/SIMD * iters * 4 groups
  const double groups = (double)(threads / 256) * iters * 4;
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-34s threads %4d  NV %2d  %8.3f ms  %6.1f ns  %6.1f shader cycles per group per SIMD  (shader clock %.0f MHz)\n", name, threads, NV, ms,
         ms * 1e6 / groups, (double)h[0] / groups, (double)h[0] / ((double)h[1] / 100.0));
  hipFree(out);
}

int main() {
  const int it = 200000;
  run<0, 0>("MFMA only", 256, it);
  run<0, 0>("MFMA only", 512, it);
  run<1, 6, 0>("VALU only fma", 256, it);
  run<3, 6, 0>("split waves: MFMA | fma", 512, it);
  run<1, 6, 1>("VALU only med3", 256, it);
  run<3, 6, 1>("split waves: MFMA | med3", 512, it);
  run<1, 6, 2>("VALU only iadd", 256, it);
  run<3, 6, 2>("split waves: MFMA | iadd", 512, it);
  run<1, 6, 3>("VALU only cvt16 x2 + add", 256, it);
  run<3, 6, 3>("split waves: MFMA | cvt", 512, it);
  run<1, 6, 4>("VALU only cmp+cndmask+add", 256, it);
  run<3, 6, 4>("split waves: MFMA | cmp/cndmask", 512, it);
  run<2, 6, 2>("same wave MFMA + iadd", 256, it);
  run<2, 6, 1>("same wave MFMA + med3", 256, it);
  // round 2: the same-wave interleave (1 MFMA : NV VALU) for every op class, one and two waves per SIMD, and fewer fillers
  run<2, 6, 0>("same wave MFMA + 6 fma", 256, it);
  run<2, 6, 0>("same wave MFMA + 6 fma", 512, it);
  run<2, 3, 0>("same wave MFMA + 3 fma", 256, it);
  run<2, 3, 0>("same wave MFMA + 3 fma", 512, it);
  run<2, 6, 4>("same wave MFMA + 6 (cmp,cndmask,add)", 256, it);
  run<2, 2, 4>("same wave MFMA + 2 (cmp,cndmask,add)", 256, it);
  run<2, 2, 4>("same wave MFMA + 2 (cmp,cndmask,add)", 512, it);
  run<2, 6, 3>("same wave MFMA + 6 (cvt,cvt,add)", 256, it);
  run<2, 6, 1>("same wave MFMA + med3", 512, it);
  run<1, 6, 5>("VALU only max", 256, it);
  run<3, 6, 5>("split waves: MFMA | max", 512, it);
  run<2, 6, 5>("same wave MFMA + 6 max", 256, it);
  run<2, 6, 5>("same wave MFMA + 6 max", 512, it);
  run<1, 6, 6>("VALU only and_or", 256, it);
  run<2, 6, 6>("same wave MFMA + 6 and_or", 256, it);
  run<2, 6, 6>("same wave MFMA + 6 and_or", 512, it);
  run<1, 2, 7>("VALU only 2 x (and_or,max,med3)", 256, it);
  run<2, 2, 7>("same wave MFMA + 2 x (and_or,max,med3)", 256, it);
  run<2, 2, 7>("same wave MFMA + 2 x (and_or,max,med3)", 512, it);
  run<2, 1, 7>("same wave MFMA + 1 x (and_or,max,med3)", 512, it);
  run<1, 3, 8>("VALU only 3 x (cmp,cndmask)", 256, it);
  run<2, 3, 8>("same wave MFMA + 3 x (cmp,cndmask)", 256, it);
  run<2, 3, 8>("same wave MFMA + 3 x (cmp,cndmask)", 512, it);
  run<1, 6, 0>("VALU only fma", 512, it);
  run<1, 6, 4>("VALU only cmp+cndmask+add", 512, it);
  return 0;
}
