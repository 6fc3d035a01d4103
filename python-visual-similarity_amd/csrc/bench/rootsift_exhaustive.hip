// Exhaustive check of cheaper RootSIFT element formulas for uint8 descriptors: the value sqrt(raw / (s + 1e-7f)) depends on
// (raw, s) only, raw in 0..255, s = the row sum, an integer in 0..128*255 -- 8.4 M pairs, all compared here with the IEEE
// expression (correctly rounded division and square root, what NumPy computes).
// Build: hipcc -O3 --offload-arch=gfx950 rootsift_exhaustive.hip -o rootsift_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float ref(float raw, float s) { return sqrtf(raw / (s + 1e-7f)); }

// division: q0 = raw * r, one or two fma refinements with r = fl(1 / d) (IEEE, once per row)
template <int DIVSTEPS>
__device__ __forceinline__ float fast_div(float raw, float d, float r) {
  float q = raw * r;
#pragma unroll
  for (int i = 0; i < DIVSTEPS; ++i) {
    const float e = __builtin_fmaf(-q, d, raw);
    q = __builtin_fmaf(e, r, q);
  }
  return q;
}
// square root from the hardware reciprocal square root + Newton steps
template <int MODE>
__device__ __forceinline__ float fast_sqrt(float q) {
  if (q == 0.f) return 0.f;
  const float rs = __builtin_amdgcn_rsqf(q);
  float g = q * rs, h = 0.5f * rs;
  if (MODE >= 2) {
    const float r1 = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r1, g);
    h = __builtin_fmaf(h, r1, h);
  }
  if (MODE >= 1) {
    const float e = __builtin_fmaf(-g, g, q);
    g = __builtin_fmaf(e, h, g);
  }
  if (MODE >= 3) {
    const float e = __builtin_fmaf(-g, g, q);
    g = __builtin_fmaf(e, h, g);
  }
  return g;
}

template <int DIVSTEPS, int MODE>
__global__ void check(unsigned long long* bad, unsigned long long* bad_div) {
  const int s_i = blockIdx.x;            // 0 .. 32640
  const int raw_i = threadIdx.x;         // 0 .. 255
  const float s = (float)s_i, raw = (float)raw_i;
  if (raw_i > s_i) return;               // an element cannot exceed its row sum
  const float d = s + 1e-7f;
  const float r = 1.0f / d;
  const float q = fast_div<DIVSTEPS>(raw, d, r);
  const float want_q = raw / d;
  if (__float_as_uint(q) != __float_as_uint(want_q)) atomicAdd(bad_div, 1ull);
  const float got = fast_sqrt<MODE>(q);
  if (__float_as_uint(got) != __float_as_uint(ref(raw, s))) atomicAdd(bad, 1ull);
}

template <int DIVSTEPS, int MODE>
static void run() {
  unsigned long long *d, h[2] = {0, 0};
  (void)hipMalloc(&d, 16);
  (void)hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  check<DIVSTEPS, MODE><<<128 * 255 + 1, 256>>>(d, d + 1);
  (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("division: %d fma steps -> %llu quotients differ;  sqrt mode %d -> %llu results differ (of ~8.3 M pairs)\n", DIVSTEPS, h[1], MODE, h[0]);
  (void)hipFree(d);
}

int main() {
  run<0, 1>(); run<1, 1>(); run<2, 1>();
  run<1, 0>(); run<1, 2>(); run<1, 3>(); run<2, 2>(); run<2, 3>();
  return 0;
}
