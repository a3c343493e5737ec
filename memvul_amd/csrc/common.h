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
// and gelu(x) = 0.5 x erfc(z) for x < 0, x - 0.5 x erfc(z) for x >= 0: no cancellation, no branches, one
// v_rcp + one v_exp (ocml erff costs ~45 VALU with two divergent branches per element, which made the FFN-1
// epilogue as long as its main loop).  The result is stored as fp16 (rel. step 4.9e-4), so 1.5e-7 is noise.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
  const float q = 0.5f * x * poly * e;  // 0.5 x erfc(|x|/sqrt2)
  return x >= 0.f ? x - q : q;
}
