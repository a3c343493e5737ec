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

// ---- MV_F16X8 planes: OCP e4m3 of x 2^MV_X8_ACT_SHIFT (hi8) and of (x - fp16(x)) 2^(11 + MV_X8_ACT_SHIFT) (lo8).
// v_cvt_scalef32_pk_fp8_f32 converts src / scale (the power of two rides in the instruction: no multiply) but, like
// v_cvt_pk_fp8_f32, does NOT saturate — overflow gives NaN (tools/cvt_fp8_probe.hip on the MI355X; hip_fp8.h clamps too) —
// so each value is clamped to the format's +-448 / 2^shift first.  Activations use ONE static shift: |x| up to 112 keeps
// its hi8 / lo8 in range; larger values only lose the correction term of that element (graceful: fp16-level accuracy there).
#define MV_X8_ACT_SHIFT 2
typedef short shortx2_t __attribute__((ext_vector_type(2)));
// four values (already inside the format's range after the scale) -> one dword of e4m3(v / scale)
__device__ __forceinline__ uint32_t pack_fp8x4_scaled(float a, float b, float c, float d, float scale) {
  shortx2_t v = {0, 0};
  v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, a, b, scale, false);
  v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, c, d, scale, true);
  return __builtin_bit_cast(uint32_t, v);
}
// four consecutive values -> (hi8 dword, lo8 dword).  ONE clamp per value (to +-448 / 2^shift = +-112) serves both planes: the
// rounding residual of a clamped value c is at most 2^-5 (c below 128), i.e. 2^8 after the lo plane's 2^(11 + shift) — inside
// e4m3's range — and a clamped-away value (exactly +-112, representable) has residual 0: it loses its correction term, nothing else.
__device__ __forceinline__ void x8_planes4(float v0, float v1, float v2, float v3, uint32_t& hi8, uint32_t& lo8) {
  constexpr float SH = 1.0f / (float)(1 << MV_X8_ACT_SHIFT), SL = 1.0f / (float)(2048 << MV_X8_ACT_SHIFT), BOUND = 448.f * SH;
  v0 = __builtin_amdgcn_fmed3f(v0, -BOUND, BOUND); v1 = __builtin_amdgcn_fmed3f(v1, -BOUND, BOUND);
  v2 = __builtin_amdgcn_fmed3f(v2, -BOUND, BOUND); v3 = __builtin_amdgcn_fmed3f(v3, -BOUND, BOUND);
  hi8 = pack_fp8x4_scaled(v0, v1, v2, v3, SH);
  lo8 = pack_fp8x4_scaled(v0 - (float)(half_t)v0, v1 - (float)(half_t)v1, v2 - (float)(half_t)v2, v3 - (float)(half_t)v3, SL);
}
// the hi8 plane alone (a consumer that sweeps only the weight-side term never reads lo8): clamped form and the common in-range form
__device__ __forceinline__ uint32_t x8_hi4(float v0, float v1, float v2, float v3) {
  constexpr float SH = 1.0f / (float)(1 << MV_X8_ACT_SHIFT), BOUND = 448.f * SH;
  return pack_fp8x4_scaled(__builtin_amdgcn_fmed3f(v0, -BOUND, BOUND), __builtin_amdgcn_fmed3f(v1, -BOUND, BOUND),
                           __builtin_amdgcn_fmed3f(v2, -BOUND, BOUND), __builtin_amdgcn_fmed3f(v3, -BOUND, BOUND), SH);
}
__device__ __forceinline__ uint32_t x8_hi4_in_range(float v0, float v1, float v2, float v3) {
  return pack_fp8x4_scaled(v0, v1, v2, v3, 1.0f / (float)(1 << MV_X8_ACT_SHIFT));
}
// The common path: every value inside the range (|v| <= 112, checked per block by the caller through x8_absmax4 / x8_any_out_of_range, which
// falls back to x8_planes4 for a block that is not): no clamps — and without them the fp16 rounding below is the SAME expression as the one the
// producer's fp16 output store needs, so hipcc computes it once (v_cvt_pk_f16_f32 + v_cvt_f32_f16 instead of a second v_cvt_f16_f32 per value).
// In range the two functions give identical bits.
__device__ __forceinline__ void x8_planes4_in_range(float v0, float v1, float v2, float v3, uint32_t& hi8, uint32_t& lo8) {
  constexpr float SH = 1.0f / (float)(1 << MV_X8_ACT_SHIFT), SL = 1.0f / (float)(2048 << MV_X8_ACT_SHIFT);
  hi8 = pack_fp8x4_scaled(v0, v1, v2, v3, SH);
  lo8 = pack_fp8x4_scaled(v0 - (float)(half_t)v0, v1 - (float)(half_t)v1, v2 - (float)(half_t)v2, v3 - (float)(half_t)v3, SL);
}

// The same with the fp16 roundings taken from the PACKED fp16 words the producer has already formed for its fp16 output (p01 = fp16(v0) | fp16(v1) << 16,
// p23 likewise): hipcc turns {(half)a, (half)b} into one vector fptrunc (v_cvt_pk_f16_f32) that it does not unify with the scalar roundings above, so
// FFN-1's VALU-bound epilogue would convert every value twice.  Identical bits.
__device__ __forceinline__ void x8_planes4_in_range_packed(float v0, float v1, float v2, float v3, uint32_t p01, uint32_t p23, uint32_t& hi8, uint32_t& lo8) {
  constexpr float SH = 1.0f / (float)(1 << MV_X8_ACT_SHIFT), SL = 1.0f / (float)(2048 << MV_X8_ACT_SHIFT);
  // residual = v - fp16(v) as ONE v_fma_mix_f32 per value (fp16 half of the packed word x (-1) + v: exact, the same bits as convert + subtract); the
  // asm also keeps the packed words opaque (instcombine would fold an extraction back into fpext(fptrunc(v)) and round every value twice again)
  float l0, l1, l2, l3;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_fma_mix_f32 %0, %4, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %4, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %2, %5, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %5, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
      : "v"(p01), "v"(p23), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#else
  l0 = l1 = l2 = l3 = 0.f;
#endif
  hi8 = pack_fp8x4_scaled(v0, v1, v2, v3, SH);
  lo8 = pack_fp8x4_scaled(l0, l1, l2, l3, SL);
}

// Saturation accounting (mv_x8_saturation): an element beyond +-112 keeps fp16 accuracy but loses its correction term — silently,
// unless somebody counts.  The producers fold |v| of everything they convert into ONE per-lane running maximum (two v_max3_f32 per
// four values, no scalar state: a first form that counted with v_cmp + s_bcnt1 per value cost the FFN-1 epilogue 30 us in SGPR spills)
// and test it once per block of values; only a block that holds an out-of-range element (never, on the models measured) is
// re-counted exactly and added to the handle's device counter.
#define MV_X8_ACT_BOUND (448.f / (float)(1 << MV_X8_ACT_SHIFT))
__device__ __forceinline__ float x8_absmax4(float m, float v0, float v1, float v2, float v3) {
  m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v0)), __builtin_fabsf(v1));
  return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v2)), __builtin_fabsf(v3));
}
__device__ __forceinline__ bool x8_any_out_of_range(float m) { return __builtin_amdgcn_ballot_w64(m > MV_X8_ACT_BOUND) != 0; }
__device__ __forceinline__ int x8_count4(float v0, float v1, float v2, float v3) {  // wave-wide count (the rare path)
  return __builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_fabsf(v0) > MV_X8_ACT_BOUND)) +
         __builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_fabsf(v1) > MV_X8_ACT_BOUND)) +
         __builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_fabsf(v2) > MV_X8_ACT_BOUND)) +
         __builtin_popcountll(__builtin_amdgcn_ballot_w64(__builtin_fabsf(v3) > MV_X8_ACT_BOUND));
}
__device__ __forceinline__ void x8_sat_add(unsigned long long* counter, int n) {  // 64-bit: one element per token and layer cannot wrap it
  if (counter && n && (threadIdx.x & 63) == 0) atomicAdd(counter, (unsigned long long)n);
}

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

// exact-erf GELU (HF "gelu", BertIntermediate): gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|).
// log2 Phi(-a), a >= 0, is smooth enough that a degree-7 polynomial Q (tools/fit_gelu_tail.py: Chebyshev least
// squares on [0, 6.5]) gives Phi(-a) = exp2(Q(a)) to 9e-6 relative, i.e. the GELU to 7e-7 absolute — three orders
// below the fp16 rounding of the stored value (rel. step 4.9e-4) — with ONE transcendental per element; Q's leading
// coefficient is negative and Q falls monotonically beyond the fit interval, so no clamp is needed
// (|x| exp2(Q(|x|)) < 3e-10 for |x| > 6.5).  No cancellation, no branches or selects; every operation but v_exp is an
// fma/mul that hipcc pairs into v_pk_fma_f32 / v_pk_mul_f32 when two values are processed together (gelu_erf2):
// 12 VALU issues + 2 v_exp per PAIR of elements (the previous Abramowitz-Stegun 7.1.26 erfc form needed 14 + 4
// transcendentals; the FFN-1 epilogue is VALU-bound with the matrix pipe idle, profiles/r01_g_*).
// gelu_erf and gelu_erf2 perform the same IEEE operations in the same order -> identical bits.
#define MV_GELU_Q0 (-0.9999971389770508f)
#define MV_GELU_Q1 (-1.1512190103530884f)
#define MV_GELU_Q2 (-0.45855340361595154f)
#define MV_GELU_Q3 (-0.05386830121278763f)
#define MV_GELU_Q4 (0.00846320204436779f)
#define MV_GELU_Q5 (-0.0009194298181682825f)
#define MV_GELU_Q6 (6.024034883012064e-05f)
#define MV_GELU_Q7 (-1.7698473584459862e-06f)
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2_t gelu_erf2(float2_t x) {
  const float2_t az = __builtin_elementwise_abs(x);
  float2_t q = __builtin_elementwise_fma(az, (float2_t)(MV_GELU_Q7), (float2_t)(MV_GELU_Q6));
  q = __builtin_elementwise_fma(az, q, (float2_t)(MV_GELU_Q5));
  q = __builtin_elementwise_fma(az, q, (float2_t)(MV_GELU_Q4));
  q = __builtin_elementwise_fma(az, q, (float2_t)(MV_GELU_Q3));
  q = __builtin_elementwise_fma(az, q, (float2_t)(MV_GELU_Q2));
  q = __builtin_elementwise_fma(az, q, (float2_t)(MV_GELU_Q1));
  q = __builtin_elementwise_fma(az, q, (float2_t)(MV_GELU_Q0));
  float2_t e;
  e.x = __builtin_amdgcn_exp2f(q.x);  // Phi(-|x|)
  e.y = __builtin_amdgcn_exp2f(q.y);
  const float2_t ha = az * (float2_t)(0.5f);
  const float2_t s = __builtin_elementwise_fma(x, (float2_t)(0.5f), ha);  // max(x, 0), exactly
  return __builtin_elementwise_fma(-az, e, s);
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float az = fabsf(x);
  float q = __builtin_fmaf(az, MV_GELU_Q7, MV_GELU_Q6);
  q = __builtin_fmaf(az, q, MV_GELU_Q5);
  q = __builtin_fmaf(az, q, MV_GELU_Q4);
  q = __builtin_fmaf(az, q, MV_GELU_Q3);
  q = __builtin_fmaf(az, q, MV_GELU_Q2);
  q = __builtin_fmaf(az, q, MV_GELU_Q1);
  q = __builtin_fmaf(az, q, MV_GELU_Q0);
  const float e = __builtin_amdgcn_exp2f(q);
  const float ha = az * 0.5f;
  const float s = __builtin_fmaf(x, 0.5f, ha);
  return __builtin_fmaf(-az, e, s);
}

// Virtual LayerNorm statistics ("vstats", gemm_pp.h): three (sum, sum of squares) pairs per row — one per 256-column tile
// of the residual GEMM whose epilogue produced the row (the embedding kernel writes one pair and two zeros).  Every
// consumer (the RAW GEMM epilogues, the residual GEMMs' accumulator init, cls_gather_kernel) turns them into (mean, rstd)
// with this function, so there is no statistics kernel between the GEMMs: var = E[x^2] - mean^2 in fp32 (rows are O(1),
// clamped at 0), rstd by v_rsq_f32 (1 ulp).
__device__ __forceinline__ float2 ln_from_partials(float2 p0, float2 p1, float2 p2, float eps) {
#pragma clang fp contract(off)
  const float s1 = (p0.x + p1.x) + p2.x, s2 = (p0.y + p1.y) + p2.y;
  const float mean = s1 * (1.0f / MV_HIDDEN);
  const float var = fmaxf(s2 * (1.0f / MV_HIDDEN) - mean * mean, 0.f);
  float2 st;
  st.x = mean;
  st.y = __builtin_amdgcn_rsqf(var + eps);
  return st;
}
