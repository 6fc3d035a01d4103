// Descriptor-row access shared by the encode kernels.  A "descriptor kind" says how rows are stored
// in HBM and whether the RootSIFT tail (pyvisim/features/_features.py:112-114) is fused into the load:
//     d /= (sum_j d_j + 1e-7);  d = sqrt(d)            (fp32; IEEE-correct / and sqrt, hipcc default)
// For integer-valued SIFT rows (OpenCV's output) every partial sum is an exact integer < 2^24, so the
// row sum does not depend on the summation order and the fused transform is bit-identical to NumPy.
#pragma once
#ifndef PVS_SQRT_RN_MODE
#define PVS_SQRT_RN_MODE 1
#endif
#include "common.hpp"

namespace pvs {

template <int KIND>
struct DescTraits;
template <>
struct DescTraits<PVS_DESC_F32> {
  using elem = float;
  static constexpr bool rootsift = false;
};
template <>
struct DescTraits<PVS_DESC_F32_ROOTSIFT> {
  using elem = float;
  static constexpr bool rootsift = true;
};
template <>
struct DescTraits<PVS_DESC_U8_ROOTSIFT> {
  using elem = uint8_t;
  static constexpr bool rootsift = true;
};

__device__ __forceinline__ float rootsift_apply(float raw, float row_sum) {
  return sqrtf(raw / (row_sum + 1e-7f));
}

// The RootSIFT transform of one row: sqrt(raw / (row_sum + 1e-7f)) per element, correctly rounded division and square root
// (what NumPy computes).  For uint8 rows the value depends only on (raw, row_sum), raw in 0..255 <= row_sum <= 128 * 255, and a
// shorter sequence gives the same bits for EVERY such pair: one IEEE reciprocal per row, then q = raw r refined by one
// fma pair, g = q rsq(q) refined by one fma pair -- 10 instructions per element instead of ~22.  Checked exhaustively on the
// device against the IEEE expression (csrc/bench/rootsift_exhaustive.hip: 0 of 8.3 M pairs differ).
template <int KIND>
struct RootsiftRow {
  float d, r;
  __device__ __forceinline__ explicit RootsiftRow(float row_sum)
      : d(row_sum + 1e-7f), r(KIND == PVS_DESC_U8_ROOTSIFT ? 1.0f / (row_sum + 1e-7f) : 0.f) {}
  // the same two numbers, computed once per row by the assignment pass (uint8 rows: the row sum is an exact integer)
  __device__ __forceinline__ RootsiftRow(float d_, float r_) : d(d_), r(r_) {}
  __device__ __forceinline__ float operator()(float raw) const {
    if (KIND == PVS_DESC_U8_ROOTSIFT && d <= 32640.5f) {   // the checked domain (D <= 128); longer rows take the IEEE path
      float q = raw * r;
      q = __builtin_fmaf(__builtin_fmaf(-q, d, raw), r, q);
      const float rs = __builtin_amdgcn_rsqf(q);
      float g = q * rs;
      const float h = 0.5f * rs;
      g = __builtin_fmaf(__builtin_fmaf(-g, g, q), h, g);
      return q == 0.f ? 0.f : g;
    } else {
      return sqrtf(raw / d);
    }
  }
};

// ---- the normalisation epilogue's square root and division, in fewer instructions than the compiler's IEEE sequences and
// with the same bits (bench/exact_sqrt_div.hip: sqrt_rn == sqrtf on all 2^32 bit patterns; DivByRow == IEEE division on
// 1.4e11 pairs incl. the edge patterns).  hipcc's sqrtf / division carry denormal scaling and special-case fix-ups (14 and
// 10 instructions); at K D = 32768 outputs per image these two were a third of the aggregate kernel's instruction stream.
__device__ __forceinline__ bool sqrt_rn_fast_range(float x) { return x == 0.f || (x >= 0x1p-100f && x <= 0x1p+100f); }
template <int MODE = PVS_SQRT_RN_MODE>
__device__ __forceinline__ float sqrt_rn(float x) {
  if (!sqrt_rn_fast_range(x)) return sqrtf(x);   // denormal, huge, negative, inf, NaN: the IEEE sequence (rare: per-lane branch)
  const float rs = __builtin_amdgcn_rsqf(fmaxf(x, 0x1p-126f));   // (x = 0: g = 0 * rs = 0, residual 0 -> 0)
  float g = x * rs, h = 0.5f * rs;
  if (MODE >= 2) {
    const float r1 = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r1, g);
    h = __builtin_fmaf(h, r1, h);
  }
  const float e = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(e, h, g);
}

// a / b for many a and one b: r = fl(1 / b) once, then q0 = fl(a r), e = a - q0 b (exact in an fma), q = fl(q0 + e r): the
// correctly rounded quotient when no intermediate leaves the normal range (Markstein).  `fast` is decided per divisor; the
// caller guarantees for the dividends of a fast divisor: |a| <= 2^40 |b| ... in short |a / b| in [2^-120, 2^80] or a == +0.
// (a == -0 would give +0: callers with possibly negative zeros or unbounded dividends use operator() with guard = true.)
struct DivByRow {
  float b, r;
  bool fast;
  __device__ __forceinline__ explicit DivByRow(float den) : b(den), r(1.0f / den), fast(den >= 0x1p-40f && den <= 0x1p+40f) {}
  __device__ __forceinline__ float operator()(float a, bool guard = true) const {
    const float aa = __builtin_fabsf(a);
    if (fast && (!guard || (aa >= 0x1p-64f && aa <= 0x1p+64f))) {
      const float q0 = a * r;
      const float e = __builtin_fmaf(-q0, b, a);
      return __builtin_fmaf(e, r, q0);
    }
    return a / b;
  }
};

// 4 consecutive elements starting at column d (d % 4 == 0, row 16-B aligned for f32 / 4-B for u8).
template <int KIND>
__device__ __forceinline__ float4 load4(const void* base, int64_t row, int ld, int d) {
  if constexpr (KIND == PVS_DESC_U8_ROOTSIFT) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(base) + row * ld + d);
    return make_float4(float(w & 0xffu), float((w >> 8) & 0xffu), float((w >> 16) & 0xffu), float(w >> 24));
  } else {
    return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + row * ld + d);
  }
}

template <int KIND>
__device__ __forceinline__ float load1(const void* base, int64_t row, int ld, int d) {
  if constexpr (KIND == PVS_DESC_U8_ROOTSIFT) {
    return float(static_cast<const uint8_t*>(base)[row * ld + d]);
  } else {
    return static_cast<const float*>(base)[row * ld + d];
  }
}

__device__ __forceinline__ float wave_sum_xor(float v, int width) {
  // butterfly over `width` consecutive lanes (width = 32 or 64); same tree every run -> deterministic
  for (int m = width >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max_xor(float v, int width) {
  for (int m = width >> 1; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// The same butterfly over 32 consecutive lanes (lane i pairs with i ^ 16, 8, 4, 2, 1 in that order, so the fp32 result is
// bit-identical to wave_sum_xor(v, 32)), with the exchanges done by ds_swizzle (xor 16) and DPP row operations instead of five
// ds_bpermute round trips: ~150 instead of ~600 cycles of dependent latency.
template <int M>
__device__ __forceinline__ int lane_xor_i(int v) {
  if constexpr (M == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                       // and 0x1f, or 0, xor 0x10
  else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);   // row_ror:8
  else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                   // row_shl:4 into banks 0, 2
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);                          // row_shr:4 into banks 1, 3
  } else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  else return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);                         // quad_perm [1,0,3,2]
}
template <int M>
__device__ __forceinline__ float lane_xor_f(float v) { return __int_as_float(lane_xor_i<M>(__float_as_int(v))); }

__device__ __forceinline__ float half_sum_xor(float v) {
  v += lane_xor_f<16>(v);
  v += lane_xor_f<8>(v);
  v += lane_xor_f<4>(v);
  v += lane_xor_f<2>(v);
  v += lane_xor_f<1>(v);
  return v;
}
__device__ __forceinline__ float half_max_xor(float v) {
  v = fmaxf(v, lane_xor_f<16>(v));
  v = fmaxf(v, lane_xor_f<8>(v));
  v = fmaxf(v, lane_xor_f<4>(v));
  v = fmaxf(v, lane_xor_f<2>(v));
  v = fmaxf(v, lane_xor_f<1>(v));
  return v;
}


// ---- VLAD normalisation pieces shared by the gather, stream and fused kernels (one definition: the same bits everywhere)
// SHORT = false: the compiler's branch-free IEEE square root instead of sqrt_rn (same bits; the fused kernel's epilogue loop,
// whose four chains per lane interleave, is 30 % slower with the short form's fallback branch per value -- measured)
template <bool SHORT = true>
__device__ __forceinline__ float power_norm(float v, float p) {
  // np.sign(v) * np.abs(v) ** p  (vlad.py:106); p == 1 and p == 0.5 take exact paths
  if (p == 1.f) return v;
  const float a = fabsf(v);
  if (p == 0.5f) {
    // v > 0: sqrt, v < 0: -sqrt, v == +-0: +0, NaN: itself (sums only ever hold quiet NaNs, for which v + 0 is v)
    const float m = SHORT ? sqrt_rn(a) : sqrtf(a);
    return a > 0.f ? __builtin_copysignf(m, v) : v + 0.f;
  }
  const float m = powf(a, p);
  return v > 0.f ? m : (v < 0.f ? -m : (v == 0.f ? 0.f * m : v));
}

__device__ __forceinline__ float norm_accum(float v, int mode, float p) {
  const float a = fabsf(v);
  return mode == 2 ? v * v : (mode == 1 ? a : (mode == 3 ? a : powf(a, p)));
}

}  // namespace pvs
