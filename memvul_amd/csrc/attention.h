// Attention argument block and the single-query attention of the pruned last layer.  The full self-attention kernel is
// attention_v2.h (HF BertSelfAttention as invoked from custom_PTM_embedder.py:228):
// ctx = softmax(Q K^T / 8 + (1-mask)*-10000) V.
//
// Layout in HBM (written by the QKV GEMM epilogue): Q,K [B][12][S][64] fp16 (Q pre-scaled by 1/8),
// V^T [B][12][64][S] fp16; output ctx [B*S][768] fp16 (head h at columns 64h..64h+63).
// (The round-1 kernel that staged a head's whole K / V^T per workgroup — the A/B yardstick of rounds 1-2 — is
// git history (rounds 1-2); padded lengths 320 / 448 now run as 384 / 512 through attention_v2's 128-key chunks.)
#pragma once
#include "common.h"

struct AttnArgs {
  const half_t* q;
  const half_t* k;
  const half_t* vt;
  const int32_t* lens;  // [B] real tokens per row; keys >= len get the additive -10000 mask
  half_t* ctx;          // [B*S][768]
  int S;                // padded length: 64, 128, 192, 256, 384 or 512
  int B;
  uint8_t* ctx8;        // MV_F16X8 (attention_v2_kernel<.., X8 = 1>): [B*S][1536] = [lo8 (768) | hi8 (768)] planes of ctx (gemm_pp.h)
  unsigned long long* x8_sat; // MV_F16X8: device counter of context elements beyond the fp8 planes' range (common.h x8_planes4)
  const half_t* vt_lo;  // attention_v2_kernel<.., VLO = 1> (MV_F16X8, padded length <= 128): V^T's second fp16 plane, fp16(V - fp16(V)), same layout as vt
  const half_t *q_lo, *k_lo;  // the same for Q and K
  // MV_F16X8, round 6 "special rows" (rows 0 and 1 of every sequence hold its [CLS] and [SEP] token: misc_kernels.h embed_ln_kernel):
  const half_t* vlo_sp;  // 2^11 x the low parts of V of those two keys, [b 12 + head][64 dims][2] fp16 (gemm_pp.h GemmArgs::vlo_sp): O += p[:, 0..1] V_lo[0..1] —
                         // with attention sinks the sink token's V reaches every row's context un-averaged, so its fp16 storage alone costs 1.2e-3 on the logits
  unsigned long long* conc;  // X8, the concentration monitor (mv_attention_concentration): [0] = max over every (sequence, head, launch) of the [CLS] query row's collision
                             // mass on ORDINARY keys, sum_{j >= 2} p[0][j]^2, as float bits (>= f^2 when one token that is neither [CLS] nor [SEP] holds the share f of that
                             // row's attention), [1] = the number of (sequence, head, launch) items where it exceeds 0.25, [2] = the number of items looked at (sequences of >= 16 tokens).  The special rows cover sinks on the two
                             // delimiter tokens; a sink on an ordinary token is outside the measured envelope of the default form (profiles/r06_*_sink_envelope.txt)
  int lo8_min_len;       // X8: sequences of at least this many tokens get the hi8 plane of ctx8 alone (0: every sequence gets both planes): in the [CLS]-row form the
                         // output projection sweeps the weight-side term only and never reads a long sequence's lo8 plane (50 MB of 201 MB the launch writes)
  half_t* sp_lo_out;     // 2^11 x the low parts of the CONTEXT of those two rows, compact [2 b + row][768] fp16: the A operand of the output projection's row term
};

// Last encoder layer: only the [CLS] query (token 0) of each issue report is consumed downstream
// (BertPooler takes hidden[:, 0], model_memory.py:99), so its attention is one query row per (batch row, head):
// scores over the S keys, softmax, one V^T-weighted sum.  One wave per (b, h); HBM-bound (reads the layer's K and
// V^T once: 2 x B x 12 x S x 128 B), so every load instruction covers eight whole 128-B lines: lane = (row
// lane >> 3, 16-B chunk lane & 7) for the K rows (8 keys per instruction) and for the V^T rows (8 head dims x 64
// keys per instruction); the 8 lanes of a row combine their partial sums with three xor-shuffles.
// Numerics mirror attention_kernel: fp16 q, k, v and fp16-rounded P, fp32 scores / statistics / accumulation,
// additive -10000 on padded keys.
// q: [Bpad][768] fp32 (the Q projection of the gathered [CLS] rows, 1/8 already folded into W_q), ctx: [Bpad][768] fp16
// (or ctx32: the same rows in fp32).
__global__ __launch_bounds__(256) void attention_cls_kernel(const float* __restrict__ q, const half_t* __restrict__ k,
                                                            const half_t* __restrict__ vt, const int32_t* __restrict__ lens,
                                                            half_t* __restrict__ ctx, int S, int nbh, float* __restrict__ ctx32 = nullptr,
                                                            const half_t* __restrict__ vlo_sp = nullptr) {
  __shared__ float ps[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= nbh) return;  // no workgroup barrier below: waves are independent
  const int b = bh / MV_HEADS, h = bh - b * MV_HEADS;
  const int len = lens[b];
  const int c = lane & 7, sub = lane >> 3;
  float* pw = ps[wave];
  float qv[8];  // this lane's 8 dims of the fp16-rounded query
  {
    const float4* qp = (const float4*)(q + (size_t)b * MV_HIDDEN + h * MV_HEAD_DIM + 8 * c);
    const float4 t0 = qp[0], t1 = qp[1];
    qv[0] = (float)(half_t)t0.x; qv[1] = (float)(half_t)t0.y; qv[2] = (float)(half_t)t0.z; qv[3] = (float)(half_t)t0.w;
    qv[4] = (float)(half_t)t1.x; qv[5] = (float)(half_t)t1.y; qv[6] = (float)(half_t)t1.z; qv[7] = (float)(half_t)t1.w;
  }
  // ---- scores -> LDS (key = key0 + sub)
  const half_t* kb = k + (size_t)bh * S * MV_HEAD_DIM + (size_t)sub * MV_HEAD_DIM + 8 * c;
#pragma unroll 8
  for (int key0 = 0; key0 < S; key0 += 8) {
    const half8_t kk = *(const half8_t*)(kb + (size_t)key0 * MV_HEAD_DIM);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = __builtin_fmaf(qv[e], (float)kk[e], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (c == 0) pw[key0 + sub] = s + ((key0 + sub >= len) ? -10000.0f : 0.0f);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- softmax statistics; P (fp16-rounded) back to LDS
  float mx = -3.0e38f;
  for (int key = lane; key < S; key += 64) mx = fmaxf(mx, pw[key]);
  mx = wave_max(mx);
  float psum = 0.f;
  for (int key = lane; key < S; key += 64) {
    const float p = __expf(pw[key] - mx);
    psum += p;
    pw[key] = (float)(half_t)p;
  }
  const float inv = 1.0f / wave_sum(psum);
  __builtin_amdgcn_wave_barrier();
  // ---- o[d] = sum_key P[key] V^T[d][key]: lane = (dim 8 db + sub, keys kb0 + 8 c .. + 7)
  float acc[8];
#pragma unroll
  for (int db = 0; db < 8; ++db) acc[db] = 0.f;
  const half_t* vb = vt + ((size_t)bh * MV_HEAD_DIM + sub) * S + 8 * c;
  for (int kb0 = 0; kb0 < S; kb0 += 64) {
    const float4 p0 = *(const float4*)(pw + kb0 + 8 * c), p1 = *(const float4*)(pw + kb0 + 8 * c + 4);
    const float pr[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      const half8_t vv = *(const half8_t*)(vb + (size_t)(8 * db) * S + kb0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[db] = __builtin_fmaf(pr[e], (float)vv[e], acc[db]);
    }
  }
  float out = 0.f;
#pragma unroll
  for (int db = 0; db < 8; ++db) {
    float t = acc[db];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 4, 64);
    out = (c == db) ? t : out;  // lane (sub, c) keeps dim 8 c + sub
  }
  if (vlo_sp) {  // + p[0] V_lo[0] + p[1] V_lo[1]: the special rows' V as hi + lo (AttnArgs::vlo_sp; the values are 2^11 x the low parts)
    const half2_t l2 = *(const half2_t*)(vlo_sp + ((size_t)bh * MV_HEAD_DIM + 8 * c + sub) * 2);
    out = __builtin_fmaf(__builtin_fmaf(pw[0], (float)l2[0], pw[1] * (float)l2[1]), 1.0f / 2048.0f, out);
  }
  if (ctx32) ctx32[(size_t)b * MV_HIDDEN + h * MV_HEAD_DIM + 8 * c + sub] = out * inv;  // the fp32 [CLS] tail of MV_F16X8
  else ctx[(size_t)b * MV_HIDDEN + h * MV_HEAD_DIM + 8 * c + sub] = (half_t)(out * inv);
}
