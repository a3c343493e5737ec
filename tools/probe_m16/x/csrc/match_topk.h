// K9 + K10 fused (model_memory.py:135-147; BASELINE.json configs[4]): anchor match, softmax_2 and best-anchor / top-k
// selection in ONE pass over the anchor bank — P(same) [B, G] never goes through HBM unless the caller asks for it.
//
//   logits[b,g,c] = W_a[c] . u_b  +  W_b[c] . v_g  +  W_c[c] . |u_b - v_g|        (W_m = [W_a | W_b | W_c], bias-free, l.73,141)
//   p = softmax_2(logits)                                                          (l.142)
//   g* = argmax_g p[b,g,same]  (first maximal g, l.144-145);  top-k: k rounds of it
//
// The reference materialises the [B, G, 1536] concatenation (3.1 GB at B = 512, G = 1000).  Here:
//   * two classes: P depends only on delta = logit_0 - logit_1, so the loop keeps ONE chain per (b, g) with the
//     class-difference weights; the class-0 chain is added only when the caller wants the logits (LOGITS build:
//     logit_1 = logit_0 - delta), so every entry point derives P from the same delta;
//   * W_a . u_b is hoisted (once per issue report: lane-parallel partial sums + a fixed-order wave reduction),
//     W_b . v_g is one fma chain per anchor (lane = anchor), so the inner loop is  d = u - v;  acc += W_c * |d|
//     = 2 VALU operations per (b, g, feature) (3 with logits) instead of the 5 of the plain form (|.| is a source modifier);
//   * a workgroup = 4 issue reports x GC anchors (lane = anchor); the anchor chunk is staged through LDS MI features at
//     a time by coalesced 16-byte loads (row stride MI + 4 floats: the lane = anchor ds_read_b128 is conflict-free), two
//     chunks ahead; the issue-report rows and the weight rows sit in LDS, are read as ONE dword per lane (feature l & 3 of
//     the quad) and broadcast inside each 4-lane quad by the DPP operand of the VALU instruction itself;
//   * the chunk's P(same) values go to LDS and one wave per row runs k rounds of (value desc, index asc) arg-max; with one
//     chunk (the 124-anchor CWE memory) these ARE the results; otherwise per-chunk candidate lists [B][chunks][k]
//     (8 B k per chunk) are merged by topk_merge_kernel.
// Algorithmic HBM bytes: 4 (B P + G P) + 8 B k = 3.1 MB at B = 256, G = 1000, k = 10, all L2-resident; the
// bound is the vector ALU's ISSUE rate: 2 B G P / 64 = 4.1 M wave-instructions at B = 256, G = 1000; tools/valu_rate.hip
// measures 2.45 ns per wave-instruction at one wave per SIMD on the loaded chip and ~1.9 ns for the 4-cycle DPP forms at
// two or more (plain fp32 forms: ~1.1 ns; v_pk_fma_f32 ~2.0 ns, i.e. no faster per value) -> 4.1 M / 1024 SIMDs x 1.9 ns = 7.6 us.
// Every (b, g) result is computed by the same instruction sequence wherever it lands in the grid: results do not depend
// on B or on the chunking (tested), and mv_match / mv_forward / mv_topk / the resident sweep share this kernel.
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
  unsigned long long* clk;            // tools/match_probe.hip only: s_memtime stamps of workgroup (0,0) [start, loop, logits, end]
};

// rank key: NaN (non-finite weights upstream) ranks above every probability, like torch.argmax treats it
__device__ __forceinline__ float mk_key(float x) { return x != x ? 2.0f : x; }

// Wave-wide max of one 64-bit word per lane, every lane gets the result.  DPP row shifts + row broadcasts (the gfx9 reduction
// ladder: lane 63 ends up with the maximum) instead of six xor-shuffles: a 64-bit `__shfl_xor` is two ds_bpermute round trips
// through the LDS crossbar, and the dependent chain of twelve made a selection round ~1 300 cycles (tools/match_probe.hip).
__device__ __forceinline__ unsigned long long mk_wave_max64(unsigned long long x) {
#define MK_MAX_STEP(CTRL, ROWS)                                                                \
  {                                                                                            \
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);                                 \
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, ROWS, 0xf, false); \
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, ROWS, 0xf, false); \
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;                        \
    x = o > x ? o : x;                                                                         \
  }
  MK_MAX_STEP(0x111, 0xf)  // row_shr:1
  MK_MAX_STEP(0x112, 0xf)  // row_shr:2
  MK_MAX_STEP(0x114, 0xf)  // row_shr:4
  MK_MAX_STEP(0x118, 0xf)  // row_shr:8   -> lane 15 of each row of 16 holds the row's maximum
  MK_MAX_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  MK_MAX_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef MK_MAX_STEP
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// Wave-wide sum (same ladder, every lane gets the total): the order of the additions is fixed by the ladder, the same for every row.
__device__ __forceinline__ float mk_wave_sum(float x) {
#define MK_SUM_STEP(CTRL, ROWS) x += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, ROWS, 0xf, false));
  MK_SUM_STEP(0x111, 0xf)
  MK_SUM_STEP(0x112, 0xf)
  MK_SUM_STEP(0x114, 0xf)
  MK_SUM_STEP(0x118, 0xf)
  MK_SUM_STEP(0x142, 0xa)
  MK_SUM_STEP(0x143, 0xc)
#undef MK_SUM_STEP
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x), 63));
}

// k rounds of arg-max over candidates held NJ per lane (candidate j of lane l = slot l + 64 j), order (key desc, index asc).
// A candidate is ONE 64-bit word  (key bits << 32) | ~index  (keys are >= 0 as floats, so their bit patterns order like the
// values; ~index makes the lower index win a tie), so a round is one 64-bit wave maximum.  emit(round, slot) gets the winner's
// slot, or -1 when the candidates are exhausted; it is called by every lane with the same arguments.
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
    best = mk_wave_max64(best);
    int slot = -1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) slot = (best != 0ull && cand[j] == best) ? lane + 64 * j : slot;
    // the winner is unique (indices differ): exactly one lane holds it; everyone reads the slot from that lane
    const unsigned long long has = __ballot(slot >= 0);
    const int src = has ? __ffsll((long long)has) - 1 : 0;
    slot = __builtin_amdgcn_readlane(slot, src);
    emit(round, best == 0ull ? -1 : slot);
    prev = best == 0ull ? 0ull : best;
  }
}

// G > 256: merge the per-chunk candidate lists of issue report b (each the chunk's top k, so their union holds the global
// top k); one wave, NJ candidates per lane (64 NJ >= nchunk * k).  Everything a winner needs stays in its lane's registers:
// a round ends without a dependent global load.
template <int NJ>
__device__ __forceinline__ void mk_merge_row(const MatchArgs& a, int b, int lane) {
  const int n = a.nchunk * a.k;
  const float* pp = a.part_p + (size_t)b * n;
  const float* pq = a.part_q + (size_t)b * n;
  const int32_t* pi = a.part_i + (size_t)b * n;
  float key[NJ], pv[NJ], qv[NJ];
  int gi[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    const int cc = c < n ? c : 0;
    pv[j] = pp[cc]; qv[j] = pq[cc];
    key[j] = c < n ? mk_key(pv[j]) : -1.0f;
    gi[j] = c < n ? pi[cc] : 0x7fffffff;
  }
  mk_select<NJ>(key, gi, a.k, lane, [&](int round, int slot) {
    if (lane != (slot >= 0 ? (slot & 63) : 0)) return;  // the winner's lane writes the round's outputs
    float ps = -1.0f, q = -1.0f;
    int gw = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (slot >= 0 && (slot >> 6) == j) { ps = pv[j]; q = qv[j]; gw = gi[j]; }
    if (a.topk_p) { a.topk_p[(size_t)b * a.k + round] = ps; a.topk_idx[(size_t)b * a.k + round] = gw; }
    if (round == 0) {
      if (a.best_idx) a.best_idx[b] = gw;
      if (a.best) { a.best[2 * b + a.same_idx] = ps; a.best[2 * b + 1 - a.same_idx] = q; }
    }
  });
}
__device__ __forceinline__ void mk_merge_row_any(const MatchArgs& a, int b, int lane) {  // nchunk * k <= 1024 (host-enforced)
  const int n = a.nchunk * a.k;
  if (n <= 64) mk_merge_row<1>(a, b, lane);
  else if (n <= 256) mk_merge_row<4>(a, b, lane);
  else mk_merge_row<16>(a, b, lane);
}

// RB issue reports per row group, GC anchors per workgroup chunk (64 per wave), MI features staged per step.
// Waves: AW = GC / 64 anchor waves x RW row groups; a workgroup covers RW * RB issue reports x GC anchors.
//   <2, 256, 32, L, 2>  large banks: 8 waves, 4 rows x 256 anchors, 16 steps (two waves per SIMD already at 256 workgroups)
//   <2, 128, 64, L, 2>  the 124-anchor CWE memory: 4 waves, 4 rows x 128 anchors, 8 steps
// Measured (tools/match_probe.hip, profiles/r02_i_match_probe.txt): B = 256, G = 124: 15.7 us (round 1: 43 + 6; start of
// round 2: 24); B = 256, G = 1000, k = 10: 24.7 + 5.0 us merge (round 1: 73 + 11; start of round 2: 42 + 13).  What the
// probe's in-kernel stamps and ablations ruled out on the way is recorded at the main loop below.
// P = the embedding width the matcher runs on: 512 (header output, every reference config) or 768 (use_header = False: the
// pooler output itself, model_memory.py:69-73).
template <int RB, int GC, int MI, int LOGITS, int RW, int P = MV_PROJ>
__global__ __launch_bounds__(GC * RW) void match_topk_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const float* __restrict__ Wm, MatchArgs a) {
  constexpr int AW = GC / 64, NW = AW * RW, NT = 64 * NW, NR = RW * RB, STRIDE = MI + 4;  // row stride = 4 mod 64 floats: conflict-free b128
  static_assert((NW == 4 || NW == 8) && (MI % 4) == 0 && P % MI == 0 && (GC * MI / 4) % NT == 0, "wave split");
  __shared__ __attribute__((aligned(16))) float sv[GC * STRIDE];      // anchor chunk x MI features; later P(same) / P(other) [2][NR][GC]
  // Two classes: the probabilities depend only on delta = logit_0 - logit_1, so the loop accumulates ONE chain per (row, anchor)
  // with the class-difference weights (2 instructions per (row, feature) instead of 3); the class-0 chain is added only when the
  // caller wants the logits themselves (LOGITS: logit_1 = logit_0 - delta), so every entry point derives P from the same delta.
  // wave-uniform operands, interleaved per feature quad: [W_b delta | W_c delta | W_b[0] | W_c[0] | u_0 | .. | u_{NR-1}] x float4,
  // so that one base address + immediate offsets serve every read of a step
  constexpr int XQ = 4 + NR;
  __shared__ __attribute__((aligned(16))) float sx[(P / 4) * XQ * 4];
  __shared__ float sa[NR][2];                                         // (W_a[0] - W_a[1]) . u_r, W_a[0] . u_r
  static_assert(2 * NR * GC <= GC * STRIDE, "P(same) / P(other) reuse the staging buffer");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int aw = w % AW, rw = w / AW;
  const int g0 = blockIdx.x * GC, b0 = blockIdx.y * NR;
  const bool stamp = a.clk && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  if (stamp) a.clk[0] = __builtin_amdgcn_s_memtime();
  // ---- anchor chunks: registers -> LDS, two chunks ahead (a chunk's compute is shorter than the latency of its successor's loads)
  constexpr int NST = GC * MI / 4 / NT;  // float4 per thread per step
  // (plain unrolled loops on a local array: behind lambdas the array stayed an alloca that the backend "promoted" to LDS
  //  — 64 KB more LDS and every staged value bounced through it)
  typedef float mk_f4 __attribute__((ext_vector_type(4)));  // (an ext-vector, not HIP's float4 struct-with-union: SROA splits it)
  mk_f4 stage[2][NST];
#define MK_LOAD_CHUNK(ST, I0)                                                                                       \
  _Pragma("unroll") for (int j = 0; j < NST; ++j) {                                                               \
    const int e = tid + NT * j, r = e / (MI / 4), c4 = e % (MI / 4);                                              \
    /* rows past G read the last anchor (never ranked, never stored): an `in range ? load : 0` select makes hipcc */ \
    /* branch around every load and wait for each in turn (cdna_hip_programming.md §5 trap (c))                    */ \
    const int gr = g0 + r < a.G ? g0 + r : a.G - 1;                                                               \
    stage[ST][j] = *(const mk_f4*)(v + (size_t)gr * P + (I0) + 4 * c4);                                    \
  }
#define MK_STORE_CHUNK(ST)                                                                                        \
  _Pragma("unroll") for (int j = 0; j < NST; ++j) {                                                               \
    const int e = tid + NT * j, r = e / (MI / 4), c4 = e % (MI / 4);                                              \
    *(mk_f4*)(sv + r * STRIDE + 4 * c4) = stage[ST][j];                                                           \
  }
  MK_LOAD_CHUNK(0, 0)
  MK_LOAD_CHUNK(1, MI)
  for (int e = tid; e < NR * (P / 4); e += NT) {  // rows past B repeat the last valid one; never stored
    const int r = e / (P / 4), c4 = e % (P / 4);
    const int b = b0 + r < a.B ? b0 + r : a.B - 1;
    *(float4*)(sx + (c4 * XQ + 4 + r) * 4) = *(const float4*)(u + (size_t)b * P + 4 * c4);
  }
  for (int e = tid; e < 4 * (P / 4); e += NT) {  // slots: W_b delta, W_c delta, W_b[0], W_c[0]  (delta = class 0 - class 1)
    const int c = e / (P / 4), c4 = e % (P / 4);
    const int row = (c & 1) ? 2 : 1;  // W_m rows: [W_a | W_b | W_c] of class 0, then of class 1
    float4 ww = *(const float4*)(Wm + (size_t)row * P + 4 * c4);
    if (c < 2) {
      const float4 w1 = *(const float4*)(Wm + (size_t)(row + 3) * P + 4 * c4);
      ww.x -= w1.x; ww.y -= w1.y; ww.z -= w1.z; ww.w -= w1.w;
    }
    *(float4*)(sx + (c4 * XQ + c) * 4) = ww;
  }
  __syncthreads();
  // ---- hoisted W_a . u_r: wave w takes rows r = w, w + NW, ..; lane-parallel partial sums (ascending i), fixed-order reduce
  for (int r = w; r < NR; r += NW) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < P / 64; ++j) {
      const int i = lane + 64 * j;
      const float uu = sx[((i >> 2) * XQ + 4 + r) * 4 + (i & 3)];
      const float wa0 = Wm[lane + 64 * j];
      s0 = fmaf(wa0 - Wm[3 * P + lane + 64 * j], uu, s0);  // delta chain
      s1 = fmaf(wa0, uu, s1);                                     // class-0 chain (LOGITS)
    }
    s0 = mk_wave_sum(s0);
    s1 = mk_wave_sum(s1);
    if (lane == 0) { sa[r][0] = s0; sa[r][1] = s1; }
  }
  // ---- main loop.  What tools/match_probe.hip (in-kernel stamps, ablations) and tools/valu_rate.hip showed: at one wave per
  // SIMD (B = 256 x G = 1000 is 256 workgroups) EVERY instruction of the wave — VALU, s_nop, s_waitcnt, SALU, LDS issue — costs
  // one ~5-cycle issue slot (2.45 ns on the loaded chip), so the loop time is its instruction count; with two or more waves
  // per SIMD the plain fp32 forms issue in 2 cycles and the DPP / packed forms in 4.  Tried and measured: pair-packed
  // v_pk_add / v_pk_fma (no |x| modifier: a v_and per value, and no faster per value): no gain; software-pipelined LDS reads
  // and a deeper chunk prefetch alone: no gain (the reads were never the stall); scalar loads into SGPRs: 1.4x slower (they
  // miss the shared scalar cache and can only be waited for with lgkmcnt(0)).  What pays is fewer instructions: the class
  // delta chain (-1 of 3 per step), one asm statement per quad (hipcc follows each inline-asm statement with an s_nop and
  // re-derives its waits), uniform operands as ONE dword per lane broadcast by the DPP operand (quad_perm:[e,e,e,e]) instead
  // of four broadcast VGPRs per ds_read_b128, and eight waves per workgroup so that two waves share a SIMD.
  // One (row, feature) step = v_sub_dpp + v_fmac_dpp with |x| on the plain operand.
  struct Uni { float wbd, wcd, wb0, wc0, u[RB]; };
  const float* xl = sx + (lane & 3);
  const float* myrow = sv + (64 * aw + lane) * STRIDE;
#define MK_READ(Q, VV, S)                                                                  \
  {                                                                                        \
    const int qq = (Q) < MI / 4 ? (Q) : MI / 4 - 1; /* the read past the chunk re-reads */ \
    const float* xq = xl + (size_t)(i0 / 4 + qq) * (XQ * 4);                               \
    VV = *(const mk_f4*)(myrow + 4 * qq);                                                  \
    S.wbd = xq[0]; S.wcd = xq[4];                                                          \
    if constexpr (LOGITS) { S.wb0 = xq[8]; S.wc0 = xq[12]; } else { S.wb0 = S.wc0 = 0.f; } \
    _Pragma("unroll") for (int r = 0; r < RB; ++r) S.u[r] = xq[(4 + rw * RB + r) * 4];     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
  float dd[RB], d0[RB], bvd = 0.f, bv0 = 0.f;  // delta chains; class-0 chains (LOGITS)
#pragma unroll
  for (int r = 0; r < RB; ++r) dd[r] = d0[r] = 0.f;
  // One asm statement per feature quad: hipcc follows every inline-asm statement with an `s_nop` and re-derives its waits per
  // statement, and at one wave per SIMD EVERY instruction of the wave — s_nop, s_waitcnt, SALU, LDS issue — takes a 4-cycle
  // issue slot of its own (the probe: ~90 instructions per quad at 4 cycles each, whatever their kind).
#define MK_DPP(E) " quad_perm:[" #E "," #E "," #E "," #E "] row_mask:0xf bank_mask:0xf\n\t"
#define MK_B(E) "v_fmac_f32_dpp %[bvd], %[wbd], %[v" #E "]" MK_DPP(E)
#define MK_B0(E) "v_fmac_f32_dpp %[bv0], %[wb0], %[v" #E "]" MK_DPP(E)
#define MK_S(E, R) "v_sub_f32_dpp %[t" #R "], %[u" #R "], %[v" #E "]" MK_DPP(E)
#define MK_F(E, R) "v_fmac_f32_dpp %[a" #R "], %[wcd], |%[t" #R "]|" MK_DPP(E)
#define MK_F0(E, R) "v_fmac_f32_dpp %[b" #R "], %[wc0], |%[t" #R "]|" MK_DPP(E)
  // (per feature: the subtractions of all rows first, then the fma chains: no instruction reads a result of the one before it)
#define MK_E4(E) MK_S(E, 0) MK_S(E, 1) MK_S(E, 2) MK_S(E, 3) MK_B(E) MK_F(E, 0) MK_F(E, 1) MK_F(E, 2) MK_F(E, 3)
#define MK_E4L(E) MK_E4(E) MK_B0(E) MK_F0(E, 0) MK_F0(E, 1) MK_F0(E, 2) MK_F0(E, 3)
#define MK_E2(E) MK_S(E, 0) MK_S(E, 1) MK_B(E) MK_F(E, 0) MK_F(E, 1)
#define MK_E2L(E) MK_E2(E) MK_B0(E) MK_F0(E, 0) MK_F0(E, 1)
#define MK_OPS4(VV, S)                                                                                                      \
  : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [bvd] "+v"(bvd), [bv0] "+v"(bv0), [a0] "+v"(dd[0]), \
    [a1] "+v"(dd[1]), [a2] "+v"(dd[2]), [a3] "+v"(dd[3]), [b0] "+v"(d0[0]), [b1] "+v"(d0[1]), [b2] "+v"(d0[2]),         \
    [b3] "+v"(d0[3])                                                                                                   \
  : [wbd] "v"(S.wbd), [wcd] "v"(S.wcd), [wb0] "v"(S.wb0), [wc0] "v"(S.wc0), [u0] "v"(S.u[0]), [u1] "v"(S.u[1]),         \
    [u2] "v"(S.u[2]), [u3] "v"(S.u[3]), [v0] "v"(VV.x), [v1] "v"(VV.y), [v2] "v"(VV.z), [v3] "v"(VV.w)
#define MK_OPS2(VV, S)                                                                                                      \
  : [t0] "=&v"(t0), [t1] "=&v"(t1), [bvd] "+v"(bvd), [bv0] "+v"(bv0), [a0] "+v"(dd[0]), [a1] "+v"(dd[1]),            \
    [b0] "+v"(d0[0]), [b1] "+v"(d0[1])                                                                                 \
  : [wbd] "v"(S.wbd), [wcd] "v"(S.wcd), [wb0] "v"(S.wb0), [wc0] "v"(S.wc0), [u0] "v"(S.u[0]), [u1] "v"(S.u[1]),         \
    [v0] "v"(VV.x), [v1] "v"(VV.y), [v2] "v"(VV.z), [v3] "v"(VV.w)
#define MK_QUAD(VV, S)                                                                 \
  {                                                                                    \
    float t0, t1, t2, t3;                                                              \
    if constexpr (RB == 4 && LOGITS) {                                                 \
      asm(MK_E4L(0) MK_E4L(1) MK_E4L(2) MK_E4L(3) MK_OPS4(VV, S));                            \
    } else if constexpr (RB == 4) {                                                    \
      asm(MK_E4(0) MK_E4(1) MK_E4(2) MK_E4(3) MK_OPS4(VV, S));                                \
    } else if constexpr (LOGITS) {                                                     \
      asm(MK_E2L(0) MK_E2L(1) MK_E2L(2) MK_E2L(3) MK_OPS2(VV, S));                            \
    } else {                                                                           \
      asm(MK_E2(0) MK_E2(1) MK_E2(2) MK_E2(3) MK_OPS2(VV, S));                                \
    }                                                                                  \
    (void)t2; (void)t3;                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  }
  static_assert(RB == 4 || RB == 2, "the quad statement is written out for 4 and for 2 rows per wave");
#pragma unroll  // fully: `stage` then has only compile-time indices and no loop-carried copy (it stays in registers)
  for (int i0 = 0; i0 < P; i0 += MI) {
    if ((i0 / MI) & 1) { MK_STORE_CHUNK(1) } else { MK_STORE_CHUNK(0) }
    __syncthreads();
    if (i0 + 2 * MI < P) {  // in flight while this chunk and the next are consumed
      if ((i0 / MI) & 1) { MK_LOAD_CHUNK(1, i0 + 2 * MI) } else { MK_LOAD_CHUNK(0, i0 + 2 * MI) }
    }
    if (stamp && i0 == 0) a.clk[1] = __builtin_amdgcn_s_memtime();
    // operands one quad AHEAD of their use (two register sets, scheduling barriers): left alone hipcc sinks every read next to
    // its first use and each quad then exposes the LDS latency
    mk_f4 vA, vB;
    Uni sA, sB;
    MK_READ(0, vA, sA)
#pragma unroll 1
    for (int q = 0; q < MI / 4; q += 2) {
      MK_READ(q + 1, vB, sB)
      MK_QUAD(vA, sA)
      MK_READ(q + 2, vA, sA)
      MK_QUAD(vB, sB)
    }
    __syncthreads();  // every wave is done with this chunk before the next one overwrites it
  }
#undef MK_READ
#undef MK_QUAD
#undef MK_B
#undef MK_B0
#undef MK_S
#undef MK_F
#undef MK_F0
#undef MK_E4
#undef MK_E4L
#undef MK_E2
#undef MK_E2L
#undef MK_OPS4
#undef MK_OPS2
#undef MK_DPP
  if (stamp) a.clk[2] = __builtin_amdgcn_s_memtime();
  // ---- logits, softmax_2, optional full outputs; P(same) / P(other) of the chunk -> LDS
  float* sp = sv;             // [NR][GC]
  float* sq = sv + NR * GC;   // [NR][GC]
  const int gl = 64 * aw + lane, g = g0 + gl;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int rl = rw * RB + r, b = b0 + rl;
    const float dl = (sa[rl][0] + bvd) + dd[r];  // logit_0 - logit_1
    const float l0 = (sa[rl][1] + bv0) + d0[r];  // (LOGITS only)
    const float l1 = l0 - dl;
    const float ed = expf(-fabsf(dl));            // softmax_2 = (1, e^-|d|) / (1 + e^-|d|), the larger class first
    const float inv = 1.0f / (1.0f + ed);
    const float p0 = dl >= 0.f ? inv : ed * inv, p1 = dl >= 0.f ? ed * inv : inv;
    const float ps = a.same_idx == 0 ? p0 : p1, pq = a.same_idx == 0 ? p1 : p0;
    sp[rl * GC + gl] = g < a.G ? ps : -1.0f;
    sq[rl * GC + gl] = pq;
    if (b < a.B && g < a.G) {
      const size_t o = ((size_t)b * a.G + g) * 2;
      if constexpr (LOGITS) { if (a.logits) { a.logits[o] = l0; a.logits[o + 1] = l1; } }
      if (a.probs) { a.probs[o] = p0; a.probs[o + 1] = p1; }
      if (a.psame) a.psame[(size_t)b * a.G + g] = ps;
    }
  }
  if (a.k <= 0) return;
  __syncthreads();
  if (stamp) a.clk[3] = __builtin_amdgcn_s_memtime();
  // ---- selection: wave w ranks rows w, w + NW, ...
  for (int r = w; r < NR; r += NW) {
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
    // lane `round` remembers the round's winner; the k results are then fetched and written by k lanes at once (a round does not
    // wait for an LDS read + three global stores of lane 0)
    int won = -1;
    mk_select<AW>(key, gi, a.k, lane, [&](int round, int slot) { won = lane == round ? slot : won; });
    if (lane < a.k) {
      const int round = lane, slot = won;
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
    }
  }
  if (stamp) a.clk[4] = __builtin_amdgcn_s_memtime();
}

#undef MK_LOAD_CHUNK
#undef MK_STORE_CHUNK

// One wave per issue report.  (Merging inside match_topk_kernel — the chunk that arrives last at a per-row-group counter
// does it — was measured and dropped: the device-scope fences that make the candidate lists visible across the 8 XCDs' L2s
// turned 26 + 5 us into 85 us at B = 256, G = 1000.)
template <int NJ>
__global__ __launch_bounds__(256) void topk_merge_kernel(MatchArgs a) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b < a.B) mk_merge_row<NJ>(a, b, threadIdx.x & 63);
}

// nchunk * k <= 1024 is enforced by the host (G <= 4096 at k = 64)
inline void launch_topk_merge(const MatchArgs& a, hipStream_t stream) {
  const int n = a.nchunk * a.k;
  const dim3 grid((a.B + 3) / 4), block(256);
  if (n <= 64) hipLaunchKernelGGL(topk_merge_kernel<1>, grid, block, 0, stream, a);
  else if (n <= 256) hipLaunchKernelGGL(topk_merge_kernel<4>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(topk_merge_kernel<16>, grid, block, 0, stream, a);
}
