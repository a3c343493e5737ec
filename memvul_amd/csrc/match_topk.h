// K9 + K10 fused (model_memory.py:135-147; BASELINE.json configs[4]): anchor match, softmax_2 and best-anchor / top-k
// selection in ONE pass over the anchor bank — P(same) [B, G] never goes through HBM unless the caller asks for it.
//
//   logits[b,g,c] = W_a[c] . u_b  +  W_b[c] . v_g  +  W_c[c] . |u_b - v_g|        (W_m = [W_a | W_b | W_c], bias-free, l.73,141)
//   p = softmax_2(logits)                                                          (l.142)
//   g* = argmax_g p[b,g,same]  (first maximal g, l.144-145);  top-k: k rounds of it
//
// The reference materialises the [B, G, 1536] concatenation (3.1 GB at B = 512, G = 1000).  Here:
//   * W_a . u_b is hoisted (once per issue report: lane-parallel partial sums + a fixed-order wave reduction),
//     W_b . v_g is one fma chain per anchor (lane = anchor), so the inner loop is  d = u - v;  acc_c += W_c[c] * |d|
//     = 3 VALU operations per (b, g, feature) instead of the 5 of the plain form (the |.| is a source modifier);
//   * a workgroup = 4 issue reports x GC anchors (lane = anchor); the anchor chunk is staged through LDS MI features at
//     a time by coalesced 16-byte loads (row stride MI + 4 floats: the lane = anchor ds_read_b128 is conflict-free), the
//     next step's loads are in flight while the current one is consumed; the issue-report rows and the four weight rows
//     W_b[c], W_c[c] sit in LDS and are read as wave-uniform (broadcast) float4;
//   * the chunk's P(same) values go to LDS and one wave per row runs k rounds of (value desc, index asc) arg-max; with one
//     chunk (the 124-anchor CWE memory) these ARE the results; otherwise per-chunk candidate lists [B][chunks][k]
//     (8 B k per chunk) are merged by topk_merge_kernel.
// Algorithmic HBM bytes: 4 (B P + G P) + 8 B k = 3.1 MB at B = 256, G = 1000, k = 10, all L2-resident; the
// bound is the fp32 vector ALU: 3 B G P operations = 0.39 G lane-ops (5.0 us at 78.6 T lane-op/s).
// Every (b, g) result is computed by the same instruction sequence wherever it lands in the grid: results do not depend
// on B, on the chunking or on RB (tested), and mv_match / mv_forward / mv_topk / the resident sweep share this kernel.
#pragma once
#include "common.h"

#define MK_KMAX 64

struct MatchArgs {  // (u, v, W_m travel as separate `const __restrict__` kernel arguments: provably read-only -> W_m by scalar loads)
  int B, G, same_idx, k, nchunk;
  float *logits, *probs, *psame;      // optional full outputs: [B,G,2], [B,G,2], [B,G]
  float* best;                        // [B,2]  p[b, g*, :]      (k >= 1, final when nchunk == 1)
  int32_t* best_idx;                  // [B]
  float* topk_p;                      // [B,k]  (optional)
  int32_t* topk_idx;
  float *part_p, *part_q;             // nchunk > 1: candidates [B][nchunk][k]: P(same), P(other)
  int32_t* part_i;
};

// rank key: NaN (non-finite weights upstream) ranks above every probability, like torch.argmax treats it
__device__ __forceinline__ float mk_key(float x) { return x != x ? 2.0f : x; }

// k rounds of arg-max over candidates held NJ per lane (candidate j of lane l = slot l + 64 j), order (key desc, index asc).
// A candidate is ONE 64-bit word  (key bits << 32) | ~index  (keys are >= 0 as floats, so their bit patterns order like the
// values; ~index makes the lower index win a tie), so a round is a 64-bit max: 6 xor-shuffle stages of two dwords instead
// of three values per stage (the selection is shuffle-latency-bound).  emit(round, slot) gets the winner's slot, or -1 when
// the candidates are exhausted; it is called by every lane with the same arguments.
template <int NJ, typename F>
__device__ __forceinline__ void mk_select(const float (&key)[NJ], const int (&gidx)[NJ], int k, int lane, F emit) {
  unsigned long long cand[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    cand[j] = gidx[j] == 0x7fffffff ? 0ull : ((unsigned long long)__float_as_uint(fmaxf(key[j], 0.0f)) << 32) | (unsigned)(~gidx[j]);
  unsigned long long prev = ~0ull;
  for (int round = 0; round < k; ++round) {
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const unsigned long long c = cand[j];
      best = (c < prev && c > best) ? c : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(best, off, 64);
      best = o > best ? o : best;
    }
    int slot = -1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) slot = (best != 0ull && cand[j] == best) ? lane + 64 * j : slot;
    // the winner is unique (indices differ): exactly one lane holds it; everyone learns the slot with one more shuffle round
    const unsigned long long has = __ballot(slot >= 0);
    const int src = has ? __ffsll((long long)has) - 1 : 0;
    slot = __shfl(slot, src, 64);
    emit(round, best == 0ull ? -1 : slot);
    prev = best == 0ull ? 0ull : best;
  }
}

// RB issue reports per row group, GC anchors per workgroup chunk (64 per wave), MI features staged per step.
// Waves: AW = GC / 64 anchor waves x RW = 4 / AW row groups; a workgroup covers RW * RB issue reports x GC anchors.
//   <4, 256, 32>  large banks: 4 rows x 256 anchors, 16 steps, 53 KB LDS / 126 VGPRs (3 workgroups per CU)
//   <2, 128, 64>  the 124-anchor CWE memory: 4 rows x 128 anchors, 8 steps
// Measured (tools/match_probe.hip, profiles/r02_e_match_probe.txt): ONE workgroup alone takes 24 us (128 anchors) / 40 us
// (256 anchors) — the pass is bound by the per-wave instruction chain (about 65 VALU + 9 LDS issues per 4 features:
// 8 k instructions per wave), not by bytes: B = 256, G = 124: 24 us (round 1: 43 + 6); B = 256, G = 1000, k = 10:
// 42 + 13 us merge (round 1: 73 + 11); B = 512: 57 + 13 us.
template <int RB, int GC, int MI>
__global__ __launch_bounds__(256) void match_topk_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const float* __restrict__ Wm, MatchArgs a) {
  constexpr int AW = GC / 64, RW = 4 / AW, NR = RW * RB, STRIDE = MI + 4;  // row stride = 4 mod 64 floats: conflict-free b128
  static_assert(AW * RW == 4 && (MI % 4) == 0 && MV_PROJ % MI == 0, "wave split");
  __shared__ __attribute__((aligned(16))) float sv[GC * STRIDE];      // anchor chunk x MI features; later P(same) / P(other) [2][NR][GC]
  // wave-uniform operands, interleaved per feature quad: [W_b[0] | W_b[1] | W_c[0] | W_c[1] | u_0 | .. | u_{NR-1}] x float4,
  // so that one base address + immediate offsets serve every broadcast read of a step (separate arrays cost a
  // v_mov + s_add per read: the pass was instruction-bound on address arithmetic)
  constexpr int XQ = 4 + NR;
  __shared__ __attribute__((aligned(16))) float sx[(MV_PROJ / 4) * XQ * 4];
  __shared__ float sa[NR][2];                                         // W_a[c] . u_r
  static_assert(2 * NR * GC <= GC * STRIDE, "P(same) / P(other) reuse the staging buffer");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int aw = w % AW, rw = w / AW;
  const int g0 = blockIdx.x * GC, b0 = blockIdx.y * NR;
  // ---- first anchor chunk in flight, then the small operands -> LDS
  constexpr int NST = GC * MI / 4 / 256;  // float4 per thread per step
  // (plain unrolled loops on a local array: behind lambdas the array stayed an alloca that the backend "promoted" to LDS
  //  — 64 KB more LDS and every staged value bounced through it)
  typedef float mk_f4 __attribute__((ext_vector_type(4)));  // (an ext-vector, not HIP's float4 struct-with-union: SROA splits it)
  mk_f4 stage[NST];
#define MK_LOAD_CHUNK(I0)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < NST; ++j) {                                                               \
    const int e = tid + 256 * j, r = e / (MI / 4), c4 = e % (MI / 4);                                             \
    /* rows past G read the last anchor (never ranked, never stored): an `in range ? load : 0` select makes hipcc */ \
    /* branch around every load and wait for each in turn (cdna_hip_programming.md §5 trap (c))                    */ \
    const int gr = g0 + r < a.G ? g0 + r : a.G - 1;                                                               \
    stage[j] = *(const mk_f4*)(v + (size_t)gr * MV_PROJ + (I0) + 4 * c4);                                        \
  }
#define MK_STORE_CHUNK()                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NST; ++j) {                                                               \
    const int e = tid + 256 * j, r = e / (MI / 4), c4 = e % (MI / 4);                                             \
    *(mk_f4*)(sv + r * STRIDE + 4 * c4) = stage[j];                                                               \
  }
  MK_LOAD_CHUNK(0)
  for (int e = tid; e < NR * (MV_PROJ / 4); e += 256) {  // rows past B repeat the last valid one; never stored
    const int r = e / (MV_PROJ / 4), c4 = e % (MV_PROJ / 4);
    const int b = b0 + r < a.B ? b0 + r : a.B - 1;
    *(float4*)(sx + (c4 * XQ + 4 + r) * 4) = *(const float4*)(u + (size_t)b * MV_PROJ + 4 * c4);
  }
  for (int e = tid; e < 4 * (MV_PROJ / 4); e += 256) {
    const int c = e / (MV_PROJ / 4), c4 = e % (MV_PROJ / 4);
    const int row = c == 0 ? 1 : c == 1 ? 4 : c == 2 ? 2 : 5;  // W_m rows: [W_a | W_b | W_c] of class 0, then of class 1
    *(float4*)(sx + (c4 * XQ + c) * 4) = *(const float4*)(Wm + (size_t)row * MV_PROJ + 4 * c4);
  }
  __syncthreads();
  // ---- hoisted W_a . u_r: wave w takes rows r = w, w + 4, ..; lane-parallel partial sums (ascending i), fixed-order reduce
  for (int r = w; r < NR; r += 4) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < MV_PROJ / 64; ++j) {
      const int i = lane + 64 * j;
      const float uu = sx[((i >> 2) * XQ + 4 + r) * 4 + (i & 3)];
      s0 = fmaf(Wm[lane + 64 * j], uu, s0);
      s1 = fmaf(Wm[3 * MV_PROJ + lane + 64 * j], uu, s1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64);
      s1 += __shfl_xor(s1, off, 64);
    }
    if (lane == 0) { sa[r][0] = s0; sa[r][1] = s1; }
  }
  float d0[RB], d1[RB], bv0 = 0.f, bv1 = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) d0[r] = d1[r] = 0.f;
  const float* myrow = sv + (64 * aw + lane) * STRIDE;
#pragma unroll  // fully: `stage` then has only compile-time indices and no loop-carried copy (it stays in registers)
  for (int i0 = 0; i0 < MV_PROJ; i0 += MI) {
    MK_STORE_CHUNK()
    __syncthreads();
    if (i0 + MI < MV_PROJ) { MK_LOAD_CHUNK(i0 + MI) }  // in flight while this chunk is consumed
#pragma unroll 4
    for (int q = 0; q < MI / 4; ++q) {
      const float4 vv = *(const float4*)(myrow + 4 * q);
      const float vx[4] = {vv.x, vv.y, vv.z, vv.w};
      const float4* xq = (const float4*)sx + (size_t)(i0 / 4 + q) * XQ;  // wave-uniform: LDS broadcast reads
      const float4 b0v = xq[0], b1v = xq[1], c0v = xq[2], c1v = xq[3];
      const float wb0[4] = {b0v.x, b0v.y, b0v.z, b0v.w}, wb1[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
      const float wc0[4] = {c0v.x, c0v.y, c0v.z, c0v.w}, wc1[4] = {c1v.x, c1v.y, c1v.z, c1v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bv0 = fmaf(wb0[e], vx[e], bv0);
        bv1 = fmaf(wb1[e], vx[e], bv1);
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float4 uu = xq[4 + rw * RB + r];
        const float ux[4] = {uu.x, uu.y, uu.z, uu.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dd = fabsf(ux[e] - vx[e]);
          d0[r] = fmaf(wc0[e], dd, d0[r]);
          d1[r] = fmaf(wc1[e], dd, d1[r]);
        }
      }
    }
    __syncthreads();  // every wave is done with this chunk before the next one overwrites it
  }
  // ---- logits, softmax_2, optional full outputs; P(same) / P(other) of the chunk -> LDS
  float* sp = sv;             // [NR][GC]
  float* sq = sv + NR * GC;   // [NR][GC]
  const int gl = 64 * aw + lane, g = g0 + gl;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int rl = rw * RB + r, b = b0 + rl;
    const float l0 = (sa[rl][0] + bv0) + d0[r];
    const float l1 = (sa[rl][1] + bv1) + d1[r];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.0f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    const float ps = a.same_idx == 0 ? p0 : p1, pq = a.same_idx == 0 ? p1 : p0;
    sp[rl * GC + gl] = g < a.G ? ps : -1.0f;
    sq[rl * GC + gl] = pq;
    if (b < a.B && g < a.G) {
      const size_t o = ((size_t)b * a.G + g) * 2;
      if (a.logits) { a.logits[o] = l0; a.logits[o + 1] = l1; }
      if (a.probs) { a.probs[o] = p0; a.probs[o + 1] = p1; }
      if (a.psame) a.psame[(size_t)b * a.G + g] = ps;
    }
  }
  if (a.k <= 0) return;
  __syncthreads();
  // ---- selection: wave w ranks rows w, w + 4, ...
  for (int r = w; r < NR; r += 4) {
    const int b = b0 + r;
    if (b >= a.B) continue;
    float key[AW];
    int gi[AW];
#pragma unroll
    for (int j = 0; j < AW; ++j) {
      const int gg = g0 + lane + 64 * j;
      key[j] = mk_key(sp[r * GC + lane + 64 * j]);
      gi[j] = gg < a.G ? gg : 0x7fffffff;
    }
    mk_select<AW>(key, gi, a.k, lane, [&](int round, int slot) {
      if (lane != 0) return;
      const float ps = slot >= 0 ? sp[r * GC + slot] : -1.0f, pq = slot >= 0 ? sq[r * GC + slot] : -1.0f;
      const int gw = slot >= 0 ? g0 + slot : 0x7fffffff;
      if (a.nchunk > 1) {
        const size_t o = ((size_t)b * a.nchunk + blockIdx.x) * a.k + round;
        a.part_p[o] = ps; a.part_q[o] = pq; a.part_i[o] = gw;
      } else {
        if (a.topk_p) { a.topk_p[(size_t)b * a.k + round] = ps; a.topk_idx[(size_t)b * a.k + round] = gw; }
        if (round == 0) {
          if (a.best_idx) a.best_idx[b] = gw;
          if (a.best) { a.best[2 * b + a.same_idx] = ps; a.best[2 * b + 1 - a.same_idx] = pq; }
        }
      }
    });
  }
}

#undef MK_LOAD_CHUNK
#undef MK_STORE_CHUNK

// G > 256: merge the per-chunk candidate lists of an issue report (each the chunk's top k, so their union holds the
// global top k); one wave per issue report, candidates 16 per lane per pass.
__global__ __launch_bounds__(256) void topk_merge_kernel(MatchArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int n = a.nchunk * a.k;
  const float* pp = a.part_p + (size_t)b * n;
  const float* pq = a.part_q + (size_t)b * n;
  const int32_t* pi = a.part_i + (size_t)b * n;
  constexpr int NJ = 16;  // 1024 candidates per pass (nchunk * k <= 1024 is enforced by the host: G <= 4096 at k = 64)
  float key[NJ];
  int gi[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    key[j] = c < n ? mk_key(pp[c]) : -1.0f;
    gi[j] = c < n ? pi[c] : 0x7fffffff;
  }
  mk_select<NJ>(key, gi, a.k, lane, [&](int round, int slot) {
    if (lane != 0) return;
    const float ps = slot >= 0 ? pp[slot] : -1.0f, q = slot >= 0 ? pq[slot] : -1.0f;
    const int gw = slot >= 0 ? pi[slot] : 0x7fffffff;
    if (a.topk_p) { a.topk_p[(size_t)b * a.k + round] = ps; a.topk_idx[(size_t)b * a.k + round] = gw; }
    if (round == 0) {
      if (a.best_idx) a.best_idx[b] = gw;
      if (a.best) { a.best[2 * b + a.same_idx] = ps; a.best[2 * b + 1 - a.same_idx] = q; }
    }
  });
}
