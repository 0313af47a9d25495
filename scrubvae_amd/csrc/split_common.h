// Shared by the split-precision implicit-GEMM translation units (gemm_bf16s.hip: forward / data-gradient gather kernels;
// wgrad_bf16s.hip: weight-gradient kernels): piece splitting, MFMA wrappers, the LDS chunk swizzle.
#pragma once
#include "gemm_common.h"
#include <type_traits>

namespace svae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
// {hi16(b), hi16(a)}: the truncated-bf16 pair (a in the low half = lower k)
__device__ __forceinline__ unsigned pack_hi(float a, float b) { return __builtin_amdgcn_perm(f2u(b), f2u(a), 0x07060302u); }
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float residual(float x) { return x - u2f(f2u(x) & 0xffff0000u); }  // exact

// split 4 consecutive-k floats into P pieces of 4 bf16 (8 bytes each)
template <int P>
__device__ __forceinline__ void split4(float4 v, uint2 (&out)[P]) {
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (p == P - 1) {
      out[p].x = pack_rne(x[0], x[1]);
      out[p].y = pack_rne(x[2], x[3]);
    } else {
      out[p].x = pack_hi(x[0], x[1]);
      out[p].y = pack_hi(x[2], x[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = residual(x[i]);
    }
  }
}

// ---- fp16 pieces (H = true, P = 2): x = h0 + h1 + O(2^-22 x), h0 = fp16(x) (11 significant bits), h1 = fp16(x - h0): two pieces hold
// 22 of the 24 bits of an fp32 against 16 for two bf16 pieces, so the THREE cross products h0*g0 + h0*g1 + h1*g0 are accurate to
// ~2^-22 per product -- fp32-class accuracy at half the matrix-core work of the 6-product bf16 split (the fp16 and bf16 MFMAs run at
// the same rate).  fp16 has a narrow exponent range: operands above 65504 saturate (round-toward-zero conversion: no infinities)
// and second pieces below 6e-5 go subnormal (absolute error <= 3e-8, negligible next to O(1) activations); the WEIGHTS, whose
// second pieces would sit there, are scaled by 2^10 when they are split and the accumulators by 2^-10 in the epilogue (both exact).
// Used for the forward pass only (precision "f16x3b3"): gradients span too many decades for unscaled fp16.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16_WSCALE = 1024.f, F16_OSCALE = 1.f / 1024.f;
__device__ __forceinline__ unsigned pack_h2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ float h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ void split4h(float4 v, uint2 (&out)[2]) {
  const unsigned a = pack_h2(v.x, v.y), b = pack_h2(v.z, v.w);
  out[0].x = a;
  out[0].y = b;
  const float r0 = v.x - h2f((unsigned short)(a & 0xffffu)), r1 = v.y - h2f((unsigned short)(a >> 16));
  const float r2 = v.z - h2f((unsigned short)(b & 0xffffu)), r3 = v.w - h2f((unsigned short)(b >> 16));
  out[1].x = pack_h2(r0, r1);
  out[1].y = pack_h2(r2, r3);
}
template <int P, bool H>
__device__ __forceinline__ void split4x(float4 v, uint2 (&out)[P]) {
  if constexpr (H) { static_assert(P == 2, "fp16 pieces: two"); split4h(v, out); }
  else split4<P>(v, out);
}

// 16-byte chunk c of LDS row `row` (64-byte rows = 32 bf16) sits at chunk position c ^ swz(row):
// the ds_read_b128 operand fetch of 32 consecutive rows is then conflict-free without padding
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc += sum over kept cross products of the pieces (small terms first)
template <int P>
__device__ __forceinline__ f32x16 mfma_split(const uint4 (&a)[P], const uint4 (&b)[P], f32x16 acc) {
#pragma unroll
  for (int s = P - 1; s >= 0; --s)
#pragma unroll
    for (int i = s; i >= 0; --i) acc = mfma_bf16(a[i], b[s - i], acc);
  return acc;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;

__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int P, bool H>
__device__ __forceinline__ f32x16 mfma_splitx(const uint4 (&a)[P], const uint4 (&b)[P], f32x16 acc) {
  if constexpr (H) {
#pragma unroll
    for (int s = P - 1; s >= 0; --s)
#pragma unroll
      for (int i = s; i >= 0; --i) acc = mfma_f16(a[i], b[s - i], acc);
    return acc;
  } else {
    return mfma_split<P>(a, b, acc);
  }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_bf16(uint4 a, uint4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mfma16_split2(const uint4 (&a)[2], const uint4 (&b)[2], f32x4v acc) {  // small terms first
  acc = mfma16_bf16(a[1], b[0], acc);
  acc = mfma16_bf16(a[0], b[1], acc);
  acc = mfma16_bf16(a[0], b[0], acc);
  return acc;
}

}  // namespace svae
