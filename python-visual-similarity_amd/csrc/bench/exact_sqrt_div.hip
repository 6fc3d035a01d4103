// Are the short forms of sqrt and of division by a per-row constant used in the VLAD normalisation epilogue (desc_load.hpp:
// sqrt_rn, DivByRow) the IEEE results, bit for bit?
//   sqrt: ALL 2^32 bit patterns against sqrtf (correctly rounded, what NumPy computes).
//   division a / b with r = fl(1 / b) formed once (IEEE): q0 = fl(a r), e = fma(-q0, b, a) (exact), q = fma(e, r, q0) -- correctly
//   rounded when nothing under- or overflows (Markstein).  Checked here on 2^36 random pairs in the guarded range + edge
//   patterns (b with an all-ones significand, a = k b, a = 0, results next to a rounding boundary).
// Build: make (csrc/Makefile) -> bench/exact_sqrt_div;  `exact_sqrt_div quick` runs an eighth of the division pairs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "../desc_load.hpp"

using namespace pvs;

__global__ void check_sqrt(unsigned long long* bad, unsigned long long* fast_taken) {
  const uint32_t hi = blockIdx.x;   // 2^20 blocks x 4096 patterns
  unsigned long long b = 0, f = 0;
  for (int i = 0; i < 16; ++i) {
    const uint32_t bits = (hi << 12) | (uint32_t)(i * 256 + threadIdx.x);
    const float x = __uint_as_float(bits);
    const float want = sqrtf(x);
    const float got = sqrt_rn(x);
    const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
    b += same ? 0 : 1;
    f += sqrt_rn_fast_range(x) ? 1 : 0;
  }
  for (int m = 32; m >= 1; m >>= 1) { b += __shfl_xor(b, m, 64); f += __shfl_xor(f, m, 64); }
  if ((threadIdx.x & 63) == 0) { if (b) atomicAdd(bad, b); atomicAdd(fast_taken, f); }
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
  return z;
}

// mode 0: random significands, exponents of a in [-64, 64], of b in [-40, 40];  mode 1: b with an all-ones / all-zeros / one-bit
// significand;  mode 2: a = fl(k b) +- 1 ulp (quotients next to integers and half-integers);  mode 3: a of every magnitude
// incl. denormals and zero against guarded b (the guard must send what it cannot do to the IEEE division);  mode 4: the
// unguarded call of the normalisation epilogue
__global__ void check_div(int mode, uint64_t seed, unsigned long long* bad, unsigned long long* fast_taken) {
  unsigned long long bcount = 0, f = 0;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < 256; ++it) {
    const uint64_t z = mix(gid * 256 + it + seed * 0x9E3779B97F4A7C15ull), z2 = mix(z + 0x1234567);
    uint32_t am = (uint32_t)z & 0x7fffff, bm = (uint32_t)(z >> 23) & 0x7fffff;
    int ae = (int)((z >> 46) % 129) - 64, be = (int)((z >> 54) % 81) - 40;
    const uint32_t sgn = (uint32_t)(z2 & 1) << 31;
    if (mode == 1) {
      const int pick = (int)(z2 >> 1) % 4;
      bm = pick == 0 ? 0x7fffff : (pick == 1 ? 0 : (pick == 2 ? (1u << ((z2 >> 8) % 23)) : (0x7fffff ^ (1u << ((z2 >> 8) % 23)))));
    }
    float b = __uint_as_float(((uint32_t)(be + 127) << 23) | bm);
    float a = __uint_as_float(sgn | ((uint32_t)(ae + 127) << 23) | am);
    if (mode == 2) {
      const float k = (float)((z2 >> 8) % 4096) * 0.5f;
      a = k * b;
      const int d = (int)((z2 >> 24) % 5) - 2;
      a = __uint_as_float(__float_as_uint(a) + d);
      if (!(a == a) || fabsf(a) > 3e38f) a = b;
    }
    if (mode == 3) {
      const int e3 = (int)((z2 >> 8) % 256);    // every exponent field incl. 0 (denormals / zero) and 255 (inf / NaN)
      a = __uint_as_float(sgn | ((uint32_t)e3 << 23) | ((z2 >> 40) % 3 == 0 ? 0u : am));
    }
    bool guard = true;
    if (mode == 4) {   // the unguarded use: a == +0, or 2^-75 <= |a| <= |b| (an element of a vector whose norm is b, after sqrt)
      const int e4 = (int)((z2 >> 8) % (be + 75 + 1)) - 75;        // exponent of a in [-75, be]
      a = __uint_as_float(sgn | ((uint32_t)(e4 + 127) << 23) | am);
      if (fabsf(a) > b) a = __builtin_copysignf(b, a);
      if ((z2 >> 40) % 16 == 0) a = 0.f;
      guard = false;
    }
    const DivByRow dv(b);
    const float got = dv(a, guard), want = a / b;
    const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
    bcount += same ? 0 : 1;
    f += dv.fast ? 1 : 0;
  }
  for (int m = 32; m >= 1; m >>= 1) { bcount += __shfl_xor(bcount, m, 64); f += __shfl_xor(f, m, 64); }
  if ((threadIdx.x & 63) == 0) { if (bcount) atomicAdd(bad, bcount); atomicAdd(fast_taken, f); }
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && argv[1][0] == 'q';   // the test suite's run: all 2^32 square roots, an eighth of the division pairs
  unsigned long long *d, h[2];
  (void)hipMalloc(&d, 16);
  (void)hipMemset(d, 0, 16);
  check_sqrt<<<1 << 20, 256>>>(d, d + 1);
  (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("sqrt_rn vs sqrtf over all 2^32 bit patterns: %llu differ (short form taken for %llu patterns)\n", h[0], h[1]);
  unsigned long long total_bad = h[0];
  for (int mode = 0; mode < 5; ++mode) {
    (void)hipMemset(d, 0, 16);
    const int rounds = ((mode == 0 || mode == 4) ? 64 : 8) / (quick ? 8 : 1);
    for (int r = 0; r < rounds; ++r) check_div<<<1 << 14, 256>>>(mode, (uint64_t)mode * 1000 + r, d, d + 1);
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double n = (double)rounds * (1 << 14) * 256.0 * 256.0;
    printf("DivByRow vs IEEE division, mode %d: %.3g pairs, %llu differ (short form taken for %.1f %%)\n", mode, n, h[0], 100.0 * h[1] / n);
    total_bad += h[0];
  }
  (void)hipFree(d);
  printf(total_bad == 0 ? "ALL EQUAL\n" : "MISMATCHES\n");
  return total_bad == 0 ? 0 : 1;
}
