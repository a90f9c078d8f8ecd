// The two-f16-piece split of f32 operands shared by the split-operand conv kernels (conv_wide.hip, conv_os.hip) and by
// the producers of their inputs (reduce_rows in conv.hip): one definition, so that a row split by its producer is
// bit-identical to the same row split by a consumer.
#pragma once
#include <stdint.h>

// power of two that moves the largest |x| of a row (given as the integer max of the |x| bit patterns, monotone in the
// magnitude) into [2^14, 2^15); biased exponent clamped to [20, 240]: zero / denormal rows get a finite scale
__device__ __forceinline__ float dgr_row_scale_of(uint32_t max_abs_bits) {
  const uint32_t e = min(max(max_abs_bits >> 23, 20u), 240u);
  return max_abs_bits == 0 ? 1.f : __builtin_bit_cast(float, (268u - e) << 23);
}
// s x = h + m (+ <= 2^-22 |s x|): two f16 pieces by round-to-nearest; sx is a power of two
__device__ __forceinline__ void dgr_split2(float x, float sx, _Float16 &h, _Float16 &m) {
  const float xs = x * sx;
  h = (_Float16)xs;
  m = (_Float16)(xs - (float)h);
}
__device__ __forceinline__ float dgr_inv_pow2(float s) {   // 1 / s for a normal power of two
  return __builtin_bit_cast(float, 0x7f000000u - __builtin_bit_cast(uint32_t, s));
}
// byte offset of channel c (a multiple of 4) of piece p (0 = h, 1 = m) inside a split row (conv_wide.hip: per 64-channel
// block 128 bytes of h, then 128 bytes of m)
__device__ __forceinline__ int dgr_split_row_offset(int c, int p) { return (c >> 6) * 256 + p * 128 + (c & 63) * 2; }
