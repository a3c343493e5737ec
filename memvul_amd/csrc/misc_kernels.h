// Memory-bound and small fp32 kernels of the hot path: embeddings + LayerNorm (K1), LayerNorm of the
// residual stream (the LN half of K4/K6), pooler + header (K7/K8), anchor match + best anchor / top-k
// (K9/K10).  SURVEY.md §2a.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// One wave per token row of 768 fp32: lane owns elements 4*lane + 256*i .. +3, i = 0..2
// (three coalesced 1-KiB float4 sweeps per row).
// ---------------------------------------------------------------------------------------------
// W32: write the normalised fp32 row; otherwise only the fp16 copy (the GEMM operand) and, when `stats` is given,
// (mean, rstd) of the row: the consumer of the fp32 stream (gemm_pp PP_RESLN) then normalises the raw row itself
// with exactly the operations below — (x - mean) * rstd, one fma with gamma / beta — so both routes give the same bits.
template <bool W32>
__device__ __forceinline__ void ln_row_store(const float4 (&x)[3], const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, int lane,
                                             float* __restrict__ out32, half_t* __restrict__ out16,
                                             float* __restrict__ stats) {
  // every operation below is spelled out (explicit fma, contraction off): hipcc otherwise contracts the variance
  // sum differently in different instantiations of this template, and the two routes of the residual stream
  // (normalised here, or by the PP_RESLN consumer from the statistics) would differ in the last bit of rstd
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  const float mean = wave_sum(s) * (1.0f / MV_HIDDEN);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a = x[i].x - mean, b = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
    v = __builtin_fmaf(a, a, v);
    v = __builtin_fmaf(b, b, v);
    v = __builtin_fmaf(c, c, v);
    v = __builtin_fmaf(d, d, v);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(v) * (1.0f / MV_HIDDEN) + eps);
  if (stats && lane == 0) {
    float2 st;
    st.x = mean; st.y = rstd;
    *(float2*)stats = st;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = 4 * lane + 256 * i;
    const float4 g = *(const float4*)(gamma + c);
    const float4 bb = *(const float4*)(beta + c);
    float4 y;
    y.x = __builtin_fmaf((x[i].x - mean) * rstd, g.x, bb.x);
    y.y = __builtin_fmaf((x[i].y - mean) * rstd, g.y, bb.y);
    y.z = __builtin_fmaf((x[i].z - mean) * rstd, g.z, bb.z);
    y.w = __builtin_fmaf((x[i].w - mean) * rstd, g.w, bb.w);
    if constexpr (W32) *(float4*)(out32 + c) = y;
    half4_t h;
    h[0] = (half_t)y.x; h[1] = (half_t)y.y; h[2] = (half_t)y.z; h[3] = (half_t)y.w;
    *(half4_t*)(out16 + c) = h;
  }
}

// K1: x = LN(word[id] + pos[s] + type[0])   (HF BertEmbeddings; token-type ids are all zero in this
// path, custom_PTM_embedder.py:199-202).  ids are [B][S_in] (0-padded), the engine row pitch is Sp.
// RAWOUT (virtual LayerNorm, gemm_pp.h): x32 <- the un-normalised sum, x16 <- its fp16 copy, stats <- the row's vstats
// (the exact two-pass mean and variance, expressed as one (sum, sum of squares) pair); xlo: the stream as two fp16 planes
// (gemm_pp PP_RESLN3), x8: + its fp8 planes (MV_F16X8).
// `pitch` = ints between the rows of ids (>= S_in: a length-bucketed sweep reads only the first S_in columns of wider rows).
// lens != nullptr (MV_F16X8, round 6 "special rows"): the LAST token of every sequence ([SEP]) is computed in row 1 and token 1 in the last token's row —
// everything downstream is per-row or attention (a sum over the keys, whatever their order), positions enter through the position embedding added HERE
// and the mask only asks "row < len", so only the [CLS] row's result matters and it does not change — which puts the two tokens trained BERT heads use
// as attention sinks at the FIXED rows 0 and 1 of each sequence: the rows whose roundings can reach the pooler un-averaged (gemm_pp.h "special rows").
// sp_lo: 2^11 x the low parts (x - fp16(x)) of those two rows, compact [2 b + row][768] fp16: the A operand of the first QKV projection's row term.
template <bool RAWOUT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, int pitch, int S_in, int Sp, int n_tok,
                                                       int vocab, const float* __restrict__ wemb,
                                                       const float* __restrict__ pemb, const float* __restrict__ temb,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ x32, half_t* __restrict__ x16,
                                                       float* __restrict__ stats, half_t* __restrict__ xlo, uint8_t* __restrict__ x8,
                                                       unsigned long long* __restrict__ x8_sat, const int32_t* __restrict__ lens = nullptr,
                                                       half_t* __restrict__ sp_lo = nullptr) {
  float vmax8 = 0.f;  // MV_F16X8: max |stream value| of the row's share (saturation accounting, common.h)
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_tok) return;
  const int b = t / Sp, s = t - b * Sp;
  int ss = s;  // the token position this row holds
  if (lens) {
    const int len = lens[b];
    if (len >= 3) ss = (s == 1) ? len - 1 : (s == len - 1) ? 1 : s;
  }
  int id = (ss < S_in) ? ids[(size_t)b * pitch + ss] : 0;
  id = (id < 0 || id >= vocab) ? 0 : id;
  const float* w = wemb + (size_t)id * MV_HIDDEN;
  const float* p = pemb + (size_t)(ss < S_in ? ss : 0) * MV_HIDDEN;  // columns >= S_in are engine padding (always masked)
  float4 x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = 4 * lane + 256 * i;
    const float4 a = *(const float4*)(w + c), q = *(const float4*)(p + c), r = *(const float4*)(temb + c);
    x[i].x = a.x + q.x + r.x; x[i].y = a.y + q.y + r.y; x[i].z = a.z + q.z + r.z; x[i].w = a.w + q.w + r.w;
  }
  if constexpr (RAWOUT) {
#pragma clang fp contract(off)
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) sm += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    const float mean = wave_sum(sm) * (1.0f / MV_HIDDEN);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float a = x[i].x - mean, bb = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
      v = __builtin_fmaf(a, a, v); v = __builtin_fmaf(bb, bb, v); v = __builtin_fmaf(c, c, v); v = __builtin_fmaf(d, d, v);
    }
    const float var = wave_sum(v) * (1.0f / MV_HIDDEN);
    if (lane < 3) {
      float2 st;
      st.x = lane == 0 ? mean * MV_HIDDEN : 0.f;
      st.y = lane == 0 ? (var + mean * mean) * MV_HIDDEN : 0.f;
      *(float2*)(stats + 6 * (size_t)t + 2 * lane) = st;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = 4 * lane + 256 * i;
      half4_t h;
      h[0] = (half_t)x[i].x; h[1] = (half_t)x[i].y; h[2] = (half_t)x[i].z; h[3] = (half_t)x[i].w;
      *(half4_t*)(x16 + (size_t)t * MV_HIDDEN + c) = h;
      if (x8) {  // MV_F16X8: [lo8 | hi8] planes of the raw stream (the A8 operand of the first QKV GEMM, gemm_pp.h)
        uint32_t h8, l8;
        x8_planes4(x[i].x, x[i].y, x[i].z, x[i].w, h8, l8);
        vmax8 = x8_absmax4(vmax8, x[i].x, x[i].y, x[i].z, x[i].w);
        *(uint32_t*)(x8 + (size_t)t * (2 * MV_HIDDEN) + c) = l8;
        *(uint32_t*)(x8 + (size_t)t * (2 * MV_HIDDEN) + MV_HIDDEN + c) = h8;
      }
      if (sp_lo && s < 2) {
        half4_t l;
        l[0] = (half_t)((x[i].x - (float)h[0]) * 2048.0f); l[1] = (half_t)((x[i].y - (float)h[1]) * 2048.0f);
        l[2] = (half_t)((x[i].z - (float)h[2]) * 2048.0f); l[3] = (half_t)((x[i].w - (float)h[3]) * 2048.0f);
        *(half4_t*)(sp_lo + (size_t)(2 * b + s) * MV_HIDDEN + c) = l;
      }
      if (xlo) {  // two-plane raw stream (gemm_pp PP_RESLN3): lo = fp16(x - hi) instead of the fp32 row
        half4_t l;
        l[0] = (half_t)(x[i].x - (float)h[0]); l[1] = (half_t)(x[i].y - (float)h[1]);
        l[2] = (half_t)(x[i].z - (float)h[2]); l[3] = (half_t)(x[i].w - (float)h[3]);
        *(half4_t*)(xlo + (size_t)t * MV_HIDDEN + c) = l;
      } else if (!x8) {
        *(float4*)(x32 + (size_t)t * MV_HIDDEN + c) = x[i];
      }
    }
    if (x8 && x8_any_out_of_range(vmax8)) {  // rare: count exactly
      int n = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) n += x8_count4(x[i].x, x[i].y, x[i].z, x[i].w);
      x8_sat_add(x8_sat, n);
    }
  } else {
    (void)vmax8;
    ln_row_store<true>(x, gamma, beta, eps, lane, x32 + (size_t)t * MV_HIDDEN, x16 + (size_t)t * MV_HIDDEN, nullptr);
    // the embedding output IS the normalised stream: identity statistics for a PP_RESLN consumer (with gamma = 1, beta = 0)
    if (stats && lane == 0) {
      float2 st;
      st.x = 0.f; st.y = 1.f;
      *(float2*)(stats + 2 * (size_t)t) = st;
    }
  }
}

// LayerNorm of the residual stream (the GEMM epilogue already added bias + residual).
// W32 = true : x32 <- LN(x32) in place, x16 <- fp16(LN(x32)).
// W32 = false: x32 is left as the raw (pre-LN) stream; x16 <- fp16(LN(x32)), stats[t] <- (mean, rstd).
template <bool W32>
__global__ __launch_bounds__(256) void ln_kernel(float* __restrict__ x32, half_t* __restrict__ x16, int n_tok,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                 float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_tok) return;
  float* row = x32 + (size_t)t * MV_HIDDEN;
  float4 x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = *(const float4*)(row + 4 * lane + 256 * i);
  ln_row_store<W32>(x, gamma, beta, eps, lane, row, x16 + (size_t)t * MV_HIDDEN, W32 ? nullptr : stats + 2 * (size_t)t);
}

// Raw stream -> fp32 rows (only the un-pruned last layer needs them: its final LayerNorm kernel reads fp32).  MV_F16: hi + lo fp16 planes.  MV_F16X8 (sp_lo != null;
// gemm.h GemmArgs::out16b): a row's low part is the lo8 plane of x8 (rows [lo8 (768) | hi8 (768)]), a special row's (2^11 x) the compact sp_lo [2 b + row][768].
__global__ __launch_bounds__(256) void hilo_to_f32_kernel(const half_t* __restrict__ hi, const half_t* __restrict__ lo, size_t n4, float* __restrict__ out,
                                                          const half_t* __restrict__ sp_lo = nullptr, int Sp = 0, const uint8_t* __restrict__ x8 = nullptr) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const half4_t h = *(const half4_t*)(hi + 4 * i);
  float l[4];
  if (!sp_lo) {
    const half4_t t = *(const half4_t*)(lo + 4 * i);
    l[0] = (float)t[0]; l[1] = (float)t[1]; l[2] = (float)t[2]; l[3] = (float)t[3];
  } else {
    const size_t row = (4 * i) / MV_HIDDEN, col = 4 * i - row * MV_HIDDEN;
    const size_t b = row / Sp, s = row - b * Sp;
    if (s < 2) {
      const half4_t t = *(const half4_t*)(sp_lo + (2 * b + s) * MV_HIDDEN + col);
      l[0] = (float)t[0] * (1.0f / 2048.0f); l[1] = (float)t[1] * (1.0f / 2048.0f); l[2] = (float)t[2] * (1.0f / 2048.0f); l[3] = (float)t[3] * (1.0f / 2048.0f);
    } else {
      constexpr float SLO = 1.0f / (float)(2048 << MV_X8_ACT_SHIFT);
      const uint32_t w = *(const uint32_t*)(x8 + row * (2 * MV_HIDDEN) + col);
      const float2_t p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(w, 1.0f, false), q = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(w, 1.0f, true);
      l[0] = p.x * SLO; l[1] = p.y * SLO; l[2] = q.x * SLO; l[3] = q.y * SLO;
    }
  }
  float4 y;
  y.x = (float)h[0] + l[0]; y.y = (float)h[1] + l[1]; y.z = (float)h[2] + l[2]; y.w = (float)h[3] + l[3];
  *(float4*)(out + 4 * i) = y;
}

// Last-layer pruning (only token 0 of each issue report reaches the pooler, model_memory.py:99): gather the
// [CLS] rows of the fp32 stream and of the fp16 GEMM operand into compact [B][768] buffers.  With `stats` the
// stream holds raw (pre-LN) rows and is normalised here (same operations as ln_row_store); vstats != 0: `stats` holds
// the rows' vstats (virtual LayerNorm) instead of (mean, rstd).
__global__ __launch_bounds__(256) void cls_gather_kernel(const float* __restrict__ x32, const half_t* __restrict__ x16, int Sp,
                                                         int B, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ c32, half_t* __restrict__ c16, int raw16,
                                                         const half_t* __restrict__ xlo, int vstats, float eps,
                                                         const half_t* __restrict__ sp_lo = nullptr) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const size_t t = (size_t)b * Sp;
  // MV_F16X8 (sp_lo != null): the stream has no lo fp16 plane; the [CLS] row's low part lives in the compact sp_lo (2^11 x), row 2 b
  const half_t* lo_row = sp_lo ? sp_lo + (size_t)(2 * b) * MV_HIDDEN : xlo + t * MV_HIDDEN;
  const float lo_scale = sp_lo ? 1.0f / 2048.0f : 1.0f;
  float mean = 0.f, rstd = 1.f;
  if (stats && vstats) {
    const float2* p = (const float2*)(stats + 6 * t);
    const float2 st = ln_from_partials(p[0], p[1], p[2], eps);
    mean = st.x; rstd = st.y;
  } else if (stats) {
    mean = stats[2 * t]; rstd = stats[2 * t + 1];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = 4 * lane + 256 * i;
    float4 y;
    if (xlo || sp_lo) {  // raw stream as hi + a low part (MV_F16: the lo plane; MV_F16X8: the [CLS] row's compact low part)
      const half4_t hh = *(const half4_t*)(x16 + t * MV_HIDDEN + c), ll = *(const half4_t*)(lo_row + c);
      y.x = __builtin_fmaf((float)ll[0], lo_scale, (float)hh[0]); y.y = __builtin_fmaf((float)ll[1], lo_scale, (float)hh[1]);
      y.z = __builtin_fmaf((float)ll[2], lo_scale, (float)hh[2]); y.w = __builtin_fmaf((float)ll[3], lo_scale, (float)hh[3]);
    } else {
      y = *(const float4*)(x32 + t * MV_HIDDEN + c);
    }
    if (stats) {
      const float4 g = *(const float4*)(gamma + c);
      const float4 bb = *(const float4*)(beta + c);
      y.x = __builtin_fmaf((y.x - mean) * rstd, g.x, bb.x);
      y.y = __builtin_fmaf((y.y - mean) * rstd, g.y, bb.y);
      y.z = __builtin_fmaf((y.z - mean) * rstd, g.z, bb.z);
      y.w = __builtin_fmaf((y.w - mean) * rstd, g.w, bb.w);
    }
    *(float4*)(c32 + (size_t)b * MV_HIDDEN + c) = y;
    if (raw16) {  // x16 holds the raw stream (virtual LayerNorm): the fp16 operand is rounded from the normalised row
      half4_t h;
      h[0] = (half_t)y.x; h[1] = (half_t)y.y; h[2] = (half_t)y.z; h[3] = (half_t)y.w;
      *(half4_t*)(c16 + (size_t)b * MV_HIDDEN + c) = h;
    } else {
      *(half4_t*)(c16 + (size_t)b * MV_HIDDEN + c) = *(const half4_t*)(x16 + t * MV_HIDDEN + c);
    }
  }
}

// cls_aside: which 256-row tiles of a pass keep the default form — flags[tm] = 1 when a sequence that owns rows of tile tm has fewer than
// min_len tokens (rows past the last sequence belong to nobody).  Once per pass (the lengths do not change between the layers).
__global__ __launch_bounds__(256) void cls_tile_flags_kernel(const int32_t* __restrict__ lens, int B, int Sp, int min_len, int ntile,
                                                             int32_t* __restrict__ flags) {
  const int tm = blockIdx.x * 256 + threadIdx.x;
  if (tm >= ntile) return;
  const int b0 = (tm * 256) / Sp, b1 = (tm * 256 + 255) / Sp;
  int f = 0;
  for (int b = b0; b <= b1 && b < B; ++b) f |= lens[b] < min_len;
  flags[tm] = f;
}

// K7, K8 (model_memory.py:99-102): u = relu(W_h tanh(W_p h[:,0] + b_p) + b_h), all fp32, as two launches of one
// dense kernel: out[b][n] = act(sum_k x[b][k] W^T[k][n] + bias[n]), K = 768, on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, MI355X_MICROARCH.md "f32-input MFMA").  A workgroup owns
// one 32 x 32 output tile; its 8 waves split K (96 each, ascending k inside a wave; all of a wave's operand loads are
// issued up front — the kernel is latency-bound: 29 -> ~12 us per launch against 4 waves x 192 with 4 steps in flight)
// and the eight partial tiles are added in wave order through LDS, so the result is deterministic.  The weights are
// stored transposed ([k][n]): a B-operand load is two 128-byte row pieces; grid = B / 32 x N / 32 (192 + 128
// workgroups at B = 256).
typedef float floatx16_t __attribute__((ext_vector_type(16)));
// ACT 0: tanh (BertPooler), 1: ReLU (header FeedForward).  The [CLS] tail of the pruned last layer in the precise compute
// dtype (MV_F16X8) runs on the same kernel in full fp32 — its rows feed the pooler directly, so their operand rounding is not
// attenuated by later layers: 2: identity (Q projection), 3: exact-erf GELU (FFN-1), 4: + residual `res` (output projection,
// FFN-2; `res` may alias `out`: every element is read and written by the same thread).  KT = the contraction length.
template <int ACT, int KT = MV_HIDDEN>
__global__ __launch_bounds__(512) void dense768_kernel(const float* __restrict__ x, size_t row_stride, int B,
                                                       const float* __restrict__ WT, const float* __restrict__ bias, int N,
                                                       float* out, const float* res = nullptr) {
  constexpr int NW = 8, KW = KT / NW;  // waves, k per wave
  static_assert(KW % 96 == 0, "a wave walks its K share in chunks of 96");
  __shared__ float part[NW][16][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int h = lane >> 5, l31 = lane & 31;
  const int row = (b0 + l31 < B) ? b0 + l31 : B - 1;  // rows past B: computed on a valid row, never stored
  const float* xr = x + (size_t)row * row_stride + wave * KW;
  const float* wc = WT + (size_t)(wave * KW + h) * N + n0 + l31;
  floatx16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
  for (int kc = 0; kc < KW; kc += 96) {
    // every operand load of the chunk first (24 x 16 B of the row, 48 dwords of W^T: 144 VGPRs), then its 48 MFMAs: left to itself hipcc 7.2
    // schedules load -> s_waitcnt vmcnt(0) -> MFMA pairs, i.e. ~40 dependent L2 round trips per chunk (22 us per launch at K = 768 instead of ~8)
    float4 xa[24];
    float wv[48];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      xa[2 * j] = *(const float4*)(xr + kc + 8 * j);
      xa[2 * j + 1] = *(const float4*)(xr + kc + 8 * j + 4);
    }
#pragma unroll
    for (int j = 0; j < 48; ++j) wv[j] = wc[(size_t)(kc + 2 * j) * N];
    __builtin_amdgcn_sched_barrier(0);  // the scheduler moves nothing across this line
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const float4 a0 = xa[2 * j], a1 = xa[2 * j + 1];
      // lanes 0-31 carry k0 + 2 j, lanes 32-63 k0 + 2 j + 1
      const float s0 = h ? a0.y : a0.x, s1 = h ? a0.w : a0.z, s2 = h ? a1.y : a1.x, s3 = h ? a1.w : a1.z;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, wv[4 * j + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, wv[4 * j + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s2, wv[4 * j + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s3, wv[4 * j + 3], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 512 * j, r = idx >> 6, ln = idx & 63;
    float v = part[0][r][ln];
#pragma unroll
    for (int q = 1; q < NW; ++q) v += part[q][r][ln];
    const int orow = b0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), ocol = n0 + (ln & 31);
    if (orow < B) {
      v += bias[ocol];
      if constexpr (ACT == 0) v = tanhf(v);
      else if constexpr (ACT == 1) v = fmaxf(v, 0.f);
      else if constexpr (ACT == 3) v = gelu_erf(v);
      else if constexpr (ACT == 4) v += res[(size_t)orow * N + ocol];
      out[(size_t)orow * N + ocol] = v;
    }
  }
}

// K9 + K10 (anchor match, softmax_2, best anchor / top-k): match_topk.h
