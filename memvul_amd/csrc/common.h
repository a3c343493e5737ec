// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libmemvul_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define MV_HIDDEN 768
#define MV_HEADS 12
#define MV_HEAD_DIM 64
#define MV_INTER 3072
#define MV_PROJ 512
#define MV_WAVE 64

// Sum / max over the 64 lanes of a wave (all lanes receive the result).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Row of the 32x32 MFMA C/D fragment held in accumulator register r by a lane of half `hi`
// (= lane >> 5); the column is lane & 31 (cdna_hip_programming.md §3).
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// XCD-aware, bijective remap of a 1-D block id: the hardware dispatches block b to XCD b % 8, so give
// each XCD one contiguous chunk of the logical tile sequence (neighbouring tiles share operand panels
// in that XCD's private L2).  cdna_hip_programming.md §5 "XCD swizzle must be bijective".
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// exact-erf GELU (HF "gelu", BertIntermediate): gelu(x) = x Phi(x) = 0.5 x (1 + erf(x / sqrt 2)).
// erfc(z), z = |x|/sqrt 2 >= 0, by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7):
//   erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z)
// and gelu(x) = max(x, 0) - 0.5 |x| erfc(z) (for x >= 0: x - x (1 - Phi); for x < 0: x Phi): no cancellation, no
// branches or selects, one v_rcp + one v_exp, and every other operation is an fma/mul that hipcc pairs into
// v_pk_fma_f32 / v_pk_mul_f32 when two values are processed together (gelu_erf2): 9 VALU issues per element
// instead of ~14 for the select form and ~45 with two divergent branches for ocml erff (the FFN-1 epilogue is
// VALU-bound: profiles/r01_e_*).  The result is stored as fp16 (rel. step 4.9e-4), so 1.5e-7 is noise.
// gelu_erf and gelu_erf2 perform the same IEEE operations in the same order -> identical bits.
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2_t gelu_erf2(float2_t x) {
  const float2_t az = __builtin_elementwise_abs(x);
  const float2_t d = __builtin_elementwise_fma(az, (float2_t)(0.3275911f * 0.70710678118654752440f), (float2_t)(1.0f));
  float2_t t;
  t.x = __builtin_amdgcn_rcpf(d.x);
  t.y = __builtin_amdgcn_rcpf(d.y);
  float2_t p = __builtin_elementwise_fma(t, (float2_t)(1.061405429f), (float2_t)(-1.453152027f));
  p = __builtin_elementwise_fma(t, p, (float2_t)(1.421413741f));
  p = __builtin_elementwise_fma(t, p, (float2_t)(-0.284496736f));
  p = __builtin_elementwise_fma(t, p, (float2_t)(0.254829592f));
  p = p * t;
  float2_t m = az * az;
  m = m * (float2_t)(-0.5f * 1.44269504088896340736f);  // -z^2 log2(e)
  float2_t e;
  e.x = __builtin_amdgcn_exp2f(m.x);
  e.y = __builtin_amdgcn_exp2f(m.y);
  const float2_t pe = p * e;                             // erfc(|x| / sqrt 2)
  const float2_t ha = az * (float2_t)(0.5f);
  const float2_t s = __builtin_elementwise_fma(x, (float2_t)(0.5f), ha);  // max(x, 0), exactly
  return __builtin_elementwise_fma(-ha, pe, s);
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float az = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(az, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  p = __builtin_fmaf(t, p, 1.421413741f);
  p = __builtin_fmaf(t, p, -0.284496736f);
  p = __builtin_fmaf(t, p, 0.254829592f);
  p = p * t;
  float m = az * az;
  m = m * (-0.5f * 1.44269504088896340736f);
  const float pe = p * __builtin_amdgcn_exp2f(m);
  const float ha = az * 0.5f;
  const float s = __builtin_fmaf(x, 0.5f, ha);
  return __builtin_fmaf(-ha, pe, s);
}
