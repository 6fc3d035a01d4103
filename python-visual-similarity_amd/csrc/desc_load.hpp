// Descriptor-row access shared by the encode kernels.  A "descriptor kind" says how rows are stored
// in HBM and whether the RootSIFT tail (pyvisim/features/_features.py:112-114) is fused into the load:
//     d /= (sum_j d_j + 1e-7);  d = sqrt(d)            (fp32; IEEE-correct / and sqrt, hipcc default)
// For integer-valued SIFT rows (OpenCV's output) every partial sum is an exact integer < 2^24, so the
// row sum does not depend on the summation order and the fused transform is bit-identical to NumPy.
#pragma once
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

}  // namespace pvs
