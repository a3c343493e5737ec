// Self-attention for one (batch row, head) per workgroup column (HF BertSelfAttention
// as invoked from custom_PTM_embedder.py:228): ctx = softmax(Q K^T / 8 + (1-mask)*-10000) V.
//
// Layout in HBM (written by the QKV GEMM epilogue): Q,K [B][12][S][64] fp16 (Q pre-scaled by 1/8),
// V^T [B][12][64][S] fp16; output ctx [B*S][768] fp16 (head h at columns 64h..64h+63).
//
// Structure (cdna_hip_programming.md Appendix B "swapped QK^T", v_mfma_f32_32x32x16_f16):
//   * a workgroup stages the head's whole K ([S][64], rows padded to 144 B) and V^T ([64][S], rows
//     padded by 8 B) into LDS once (S <= 512: 70 KB at S=256, 137 KB at S=512); both paddings make the
//     fragment reads bank-conflict-free (K: ds_read_b128, 16 distinct rows -> 9r mod 16 distinct slots;
//     V^T: ds_read_b64, 32 rows -> 17d mod 32 distinct 8-B slots).
//   * each wave owns 32 query rows and walks the keys 64 at a time with an exact fp32 online softmax.
//     It computes S^T = K Q^T, so a lane holds, for ONE query (lane & 31), 16 key scores per 32-key
//     fragment: the row max / row sum are lane-local plus one exchange with lane ^ 32, and the
//     exponentiated scores, packed to fp16 in register order, ARE the B operand of the P·V MFMA
//     (O^T = V^T P^T) — no LDS round trip for P.  The k-slot pairing of the two operands is by
//     (lane >> 5, element j); the V^T fragment is read with the same key order the score registers have:
//     key = base + (j & 3) + 8 (j >> 2) + 4 (lane >> 5).
#pragma once
#include "common.h"

struct AttnArgs {
  const half_t* q;
  const half_t* k;
  const half_t* vt;
  const int32_t* lens;  // [B] real tokens per row; keys >= len get the additive -10000 mask
  half_t* ctx;          // [B*S][768]
  int S;                // padded length, multiple of 64, <= 512
  int B;
  unsigned long long* clk;  // optional (development probe, tools/attn_probe.hip): per-workgroup s_memtime ticks [total, waiting at the hand-over]
  half_t* ctx_lo;       // optional (split-operand mode, attention_kernel only): fp16(ctx - fp16(ctx)), same layout as ctx
};

#define ATT_KROW 144                         // bytes per K row in LDS (128 + 16 pad)
#define ATT_VROW(S) (2 * (S) + 8)            // bytes per V^T row in LDS
#define ATT_LDS_BYTES(S) ((S) * ATT_KROW + 64 * ATT_VROW(S))

template <int NW>  // waves per workgroup; the workgroup covers 32*NW query rows
__global__ __launch_bounds__(NW * 64, 2) void attention_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = a.S;
  const int qblocks = (S + 32 * NW - 1) / (32 * NW);
  const int bh = blockIdx.x / qblocks;          // b * 12 + h
  const int qb = blockIdx.x - bh * qblocks;
  const int b = bh / MV_HEADS, h = bh - b * MV_HEADS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  char* ldsK = smem;
  char* ldsV = smem + S * ATT_KROW;
  const int vrow = ATT_VROW(S);

  // ---- stage K and V^T (16 B per thread per step)
  {
    const half_t* gk = a.k + (size_t)bh * S * MV_HEAD_DIM;
    for (int c = tid; c < S * 8; c += NW * 64) {  // chunk c: row c/8, 16-B chunk c%8
      const half8_t v = *(const half8_t*)(gk + (size_t)c * 8);
      *(half8_t*)(ldsK + (c >> 3) * ATT_KROW + (c & 7) * 16) = v;
    }
    const half_t* gv = a.vt + (size_t)bh * MV_HEAD_DIM * S;
    const int cpr = S >> 3;  // 16-B chunks per V^T row
    for (int c = tid; c < 64 * cpr; c += NW * 64) {
      const int d = c / cpr, kc = c - d * cpr;
      const half8_t v = *(const half8_t*)(gv + (size_t)d * S + kc * 8);
      half4_t lo, hi4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = v[e]; hi4[e] = v[4 + e]; }
      *(half4_t*)(ldsV + d * vrow + kc * 16) = lo;
      *(half4_t*)(ldsV + d * vrow + kc * 16 + 8) = hi4;
    }
  }

  const int q0 = (qb * NW + wave) * 32;  // first query row of this wave
  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + ql][16 kk + 8 hi .. +7]
  half8_t qf[4];
  {
    const int qrow = (q0 + ql < S) ? (q0 + ql) : (S - 1);
    const half_t* gq = a.q + ((size_t)bh * S + qrow) * MV_HEAD_DIM + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const half8_t*)(gq + kk * 16);
  }
  __syncthreads();
  if (q0 >= S) return;  // (after the barrier) this wave has no query rows

  const int len = a.lens[b];
  floatx16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  for (int kb = 0; kb < S; kb += 64) {
    // ---- S^T fragments: st[t][r] = score(key = kb + 32 t + mfma32_row(r, hi), query = q0 + ql)
    floatx16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
      const char* krow = ldsK + (kb + t * 32 + ql) * ATT_KROW + hi * 16;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const half8_t kf = *(const half8_t*)(krow + kk * 32);
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], st[t], 0, 0, 0);
      }
    }
    if (kb + 64 > len) {  // wave-uniform: this key block contains padded keys
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb + t * 32 + mfma32_row(r, hi) >= len) st[t][r] += -10000.0f;
    }
    // ---- online softmax (fp32 statistics); both half-waves keep identical m_run
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    half8_t pf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __expf(st[t][r] - m_new);
        psum += p;
        pf[t][r >> 3][r & 7] = (half_t)p;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T[d][q] += V^T[d][keys] P^T[keys][q]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const char* vbase = ldsV + (dt * 32 + ql) * vrow + (kb + 4 * hi) * 2;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const char* vp = vbase + (t * 32 + u * 16) * 2;
          const half4_t v0 = *(const half4_t*)(vp);       // keys base + 4 hi + 0..3
          const half4_t v1 = *(const half4_t*)(vp + 16);  // keys base + 8 + 4 hi + 0..3
          half8_t vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[t][u], o[dt], 0, 0, 0);
        }
    }
  }
  // ---- normalise and store: lane holds O^T[d = 32 dt + mfma32_row(r,hi)][q0 + ql]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  half_t* dst = a.ctx + ((size_t)b * S + q0 + ql) * MV_HIDDEN + h * MV_HEAD_DIM;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      half4_t v4, l4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = o[dt][4 * rg + e] * inv;
        v4[e] = (half_t)x;
        l4[e] = (half_t)(x - (float)v4[e]);
      }
      *(half4_t*)(dst + dt * 32 + 8 * rg + 4 * hi) = v4;
      if (a.ctx_lo) *(half4_t*)(a.ctx_lo + (dst - a.ctx) + dt * 32 + 8 * rg + 4 * hi) = l4;
    }
}

// Last encoder layer: only the [CLS] query (token 0) of each issue report is consumed downstream
// (BertPooler takes hidden[:, 0], model_memory.py:99), so its attention is one query row per (batch row, head):
// scores over the S keys, softmax, one V^T-weighted sum.  One wave per (b, h); HBM-bound (reads the layer's K and
// V^T once: 2 x B x 12 x S x 128 B), so every load instruction covers eight whole 128-B lines: lane = (row
// lane >> 3, 16-B chunk lane & 7) for the K rows (8 keys per instruction) and for the V^T rows (8 head dims x 64
// keys per instruction); the 8 lanes of a row combine their partial sums with three xor-shuffles.
// Numerics mirror attention_kernel: fp16 q, k, v and fp16-rounded P, fp32 scores / statistics / accumulation,
// additive -10000 on padded keys.
// q: [Bpad][768] fp32 (the Q projection of the gathered [CLS] rows, 1/8 already folded into W_q), ctx: [Bpad][768] fp16.
__global__ __launch_bounds__(256) void attention_cls_kernel(const float* __restrict__ q, const half_t* __restrict__ k,
                                                            const half_t* __restrict__ vt, const int32_t* __restrict__ lens,
                                                            half_t* __restrict__ ctx, int S, int nbh) {
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
  ctx[(size_t)b * MV_HIDDEN + h * MV_HEAD_DIM + 8 * c + sub] = (half_t)(out * inv);
}
