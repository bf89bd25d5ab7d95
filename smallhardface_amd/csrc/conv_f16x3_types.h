// Split-fp16 convolution kernels: vector types, the hi / lo split, the MFMA wrapper, conv1_1's K-slot map.
// (part of the one translation unit conv_f16x3.hip: see its header for the arithmetic and the kernel map)
#pragma once
#include "conv_common.h"

namespace shf {


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef F16X3_DMA_LATE
#define F16X3_DMA_LATE 1   // 1: the early-finishing waves 0-3 issue the next stage's weight DMA after their MFMAs
#endif
#ifndef F16X3_CONV_MID
#define F16X3_CONV_MID 1   // 1: split the next halo tile to fp16 hi/lo in the middle of the last stage's MFMAs
#endif

namespace f16x3 {
constexpr int KC = 32;      // input channels per chunk
constexpr int ROWB = 144;   // bytes per LDS row (pixel or cout)
constexpr int TH = 16, TW = 16, HTW = TW + 2, HTH = TH + 2, HP = HTH * HTW;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
}  // namespace f16x3

// conv1_1 on the matrix cores (fused first pair): which tap (ci * 9 + ky * 3 + kx; -1: none, zero weight) K slot
// (k-step kk, half-wave kh, element j) of the 27 -> 32 padded reduction multiplies.  Chosen so that a lane's sixteen patch
// reads are base(kk, kh) + a compile-time offset: kk = 0 is input channel kh at kernel positions 0..7 (the half-waves' taps
// lie one channel plane apart), kk = 1 holds channel 2 -- half-wave kh reads column kx = kh of the three kernel rows in
// j = 0..2 (the taps lie one pixel apart), column 2 in j = 3..5 -- and position 8 of channels 0 / 1 in j = 6 / 7; the
// slots half-wave 1 has no tap for read the patch one pixel further (a finite value) against a zero weight.
__host__ __device__ constexpr int first_conv_slot_tap(int kk, int kh, int j) {
  if (kk == 0) return kh * 9 + j;
  if (j < 3) return 18 + j * 3 + kh;
  if (kh != 0) return -1;
  if (j < 6) return 18 + (j - 3) * 3 + 2;
  return (j - 6) * 9 + 8;
}

// bf16 mode (BF = true kernels; conv mode "bf16"): ONE product per fp32 product on v_mfma_f32_32x32x16_bf16, operands
// rounded to bf16 (8 mantissa bits, fp32's exponent range: no fp16 range guard, no activation exponent).  The 16-bit
// "hi" halves of the LDS rows / weight packs then hold bf16 bit patterns and the "lo" halves are never read.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
template <bool BF>
__device__ __forceinline__ f32x16 mma16(const half8 a, const half8 b, const f32x16 c) {
  if constexpr (BF)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ _Float16 bf16_as_half(float x) { return __builtin_bit_cast(_Float16, (__bf16)x); }
// two floats -> their bf16 bit patterns in one register (low half = a).  (hipcc 7.2 lowers a VECTOR float2 -> bf16x2
// conversion to v_cvt_pk_bf16_f32 with the first element in both source slots -- the odd element is lost: tools/diag_bf16.py
// -- so the instruction is spelled out.)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// four floats -> four bf16 bit patterns in a half4
__device__ __forceinline__ void bf16x4_of(const float4 v, _Float16 (&h)[4]) {
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  const h2_ a = __builtin_bit_cast(h2_, pk_bf16(v.x, v.y)), b = __builtin_bit_cast(h2_, pk_bf16(v.z, v.w));
  h[0] = a[0]; h[1] = a[1]; h[2] = b[0]; h[3] = b[1];
}

// x -> (hi, lo) for four values: v_cvt_pk_f16_f32 for the hi pairs, conv_split_lo for the lo pairs (2 VALU ops per value;
// same results as the scalar form)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4 v, half4& hi, half4& lo) {
  const f32x2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
  const half2v h01 = __builtin_convertvector(x01, half2v), h23 = __builtin_convertvector(x23, half2v);
  const half2v l01 = conv_split_lo(x01, h01), l23 = conv_split_lo(x23, h23);   // (conv_common.h: v_fma_mixlo / mixhi_f16)
  hi = half4{h01[0], h01[1], h23[0], h23[1]};
  lo = half4{l01[0], l01[1], l23[0], l23[1]};
}
template <bool BF>
__device__ __forceinline__ void split4t(const float4 v, half4& hi, half4& lo) {
  if constexpr (BF) {
    _Float16 h[4];
    bf16x4_of(v, h);
    hi = half4{h[0], h[1], h[2], h[3]};
    lo = half4{(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
  } else {
    split4(v, hi, lo);
  }
}


}  // namespace shf
