// Persistent self-attention for padded lengths S <= 256 (HF BertSelfAttention as invoked from
// custom_PTM_embedder.py:228): ctx = softmax(Q K^T / 8 + (1 - mask) * -10000) V.  Same HBM layouts as attention.h
// (Q, K [B][12][S][64] fp16 with 1/8 folded into Q; V^T [B][12][64][S] fp16; ctx [B*S][768] fp16).
//
// The kernel is HBM-bound at the bench shape (per (batch row, head): 32 KB K + 32 KB V^T + 32 KB Q in, 32 KB out;
// 400 MB per launch), so the structure is built around keeping the memory pipe busy while the matrix pipe works:
//   * one workgroup = S/32 waves (32 query rows each) walks a strided list of (batch row, head) items; K and V^T of
//     item i+1 arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces) into the other half of a 2-deep LDS ring
//     while item i is computed; the Q fragments of item i+1 are prefetched into registers; one `s_waitcnt vmcnt(0)`
//     + barrier per item hands the ring over (each K / V^T byte is read from HBM exactly once).
//   * K ([S][128 B]) and V^T (per 64-key block [64 d][128 B]) sit in LDS unpadded; 16-B chunk c of row r lives at
//     slot c ^ ((r >> 1) & 7), applied on the per-lane DMA source address and on the ds_read_b128 (guide rule 21).
//   * swapped QK^T (S^T = K Q^T) over the WHOLE key range at once: a lane holds every score of one query
//     (32 NKB... 128 fp32 registers at S = 256), so the softmax is a plain two-pass one — no running max, no
//     rescaling of O — with exp2 and log2(e) folded into one fma.
//   * the K rows of a 32-key fragment are fed to the MFMA in the order pi(i) = i with bits 2 and 3 swapped, which
//     makes the 8 scores a lane packs into one P^T k-slot group 8 CONSECUTIVE keys: the matching V^T fragment is one
//     ds_read_b128 (attention.h needs two ds_read_b64 for the native order).
//   * the normalised O tile (fp16) is kept in 16 registers across the item boundary and written out at the start of
//     the NEXT item: through the K half of the other ring slot (free, and about to be refilled by this same wave's
//     own DMA pieces — rows 32 w .. 32 w + 31 both times, so no barrier is needed) as a [S][128 B] image read back as
//     whole 128-B rows -> full-line global stores that have the whole item to complete.  One barrier per item.
//   * NCH = 3 / 4 (S = 384 / 512, instantiated with NKB = 2): a work unit is (batch row, head, block of 128 queries)
//     and its keys arrive as NCH chunks of 128 through the same ring; every chunk after the first rescales O and the row
//     sums by exp2(m_old - m_new) (online softmax at chunk granularity), so 256 < S <= 512 keeps the LDS-DMA pipeline
//     with 64 score registers per lane and two 4-wave workgroups per CU (engine.hip: attention_v2_kernel<2, S / 128>).
#pragma once
#include "attention.h"
#include "gemm_pp.h"  // glds16, pack_h2, x8_planes4

#define ATT2_BUF_BYTES(NKB) ((NKB) * 16384)
#define ATT2_LDS_BYTES(NKB) (2 * ATT2_BUF_BYTES(NKB))
#define ATT2_LDS_BYTES_VLO(NKB) (2 * (NKB) * 32768)  // VLO: + the V^T and K lo planes in every ring slot

// X8 1 (MV_F16X8, gemm_pp.h): the context is also written as fp8 planes [lo8 (768) | hi8 (768)] per token row to
// AttnArgs::ctx8 — e4m3 of (O - fp16(O)) 2^(11 + s) and of O 2^s, the A8 operand of the output projection's correction sweep
// (16 more registers across the unit boundary, a second pass through the O image); a separate instantiation.
// (The timing ablations of rounds 1-2 — no Q loads / O stores / DMA / exp / MFMA / fragment reads — were retired: git history before round 5.)
// VLO 1 (MV_F16X8, padded length <= 128: NKB <= 2, NCH = 1): Q, K, V and P as hi + lo fp16 — S^T += K_lo Q_hi + K_hi Q_lo, O^T += V_lo P_hi + V_hi P_lo on top
// of the hi x hi products.  What is left of the precise mode's error is the fp16 storage of Q, K, V and P, averaged by attention over the keys:
// ~ 1 / sqrt(keys), so short sequences feel it most (profiles/r05_f_length_envelope.txt: 9.4e-4 on the logits at 8 tokens against 2.3e-4 at 256) — and
// there the second planes are nearly free: the lo planes of K and V^T (written by the QKV projection's epilogue, GemmArgs::k_lo / vt_lo) ride through the
// ring next to K and V^T, Q's lo fragments are prefetched with Q's, P_lo = fp16(p - fp16(p)) is formed with the packing.
template <int NKB, int NCH = 1, int X8 = 0, int VLO = 0>  // chunk = 64 NKB keys = 2 NKB waves x 32 queries; padded length S = 64 NKB NCH
__global__ __launch_bounds__(NKB * 128) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_v2_kernel(AttnArgs a, int nunits) {
  static_assert(!VLO || (X8 && NCH == 1 && NKB <= 2), "the two-plane V / P path serves the precise mode's short passes");
  constexpr int S = NKB * 64;        // keys per chunk = queries per unit
  constexpr int ST = S * NCH;        // padded sequence length (row pitch of V^T, rows per head of Q / K)
  constexpr int NT = 2 * NKB;        // 32-key score fragments per chunk
  constexpr int BUF = VLO ? NKB * 32768 : ATT2_BUF_BYTES(NKB), VOFF = NKB * 8192, VLOFF = NKB * 16384, KLOFF = NKB * 24576;
  constexpr float LOG2E = 1.44269504088896340736f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;

  // ---- LDS-DMA geometry: wave w moves K pieces 4w..4w+3 and V^T pieces 4w..4w+3 of a chunk (1 KiB = 8 rows x 128 B)
  uint32_t srcK[2], srcV[2];  // per-lane byte offsets inside the chunk's K / V^T block for even / odd pieces
  {
    const int rl = lane >> 3;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int c = (lane & 7) ^ (((rl >> 1) + 4 * x) & 7);  // row = 8 (piece & 7) + rl -> (row >> 1) & 7
      srcK[x] = (uint32_t)(rl * 128 + c * 16);
      srcV[x] = (uint32_t)(rl * (2 * ST) + c * 16);
    }
  }
  // work-list position u -> unit (bh, qb): head bh, query block qb; chunk j of its keys = keys j S .. j S + S - 1.
  // NCH > 1: the NCH query blocks of a head read the same K / V^T, so they are placed 8 list positions apart — the
  // workgroups that hold them at the same time are 8 apart too, i.e. on the same XCD (round-robin dispatch), and the
  // second reader finds the chunk in that XCD's L2.  Positions are permuted inside groups of 8 NCH; a tail shorter
  // than a group keeps the plain order.
  const int nfull = nunits - nunits % (8 * NCH);
  auto unit_bh = [&](int u) -> int {
    if (NCH == 1 || u >= nfull) return u / NCH;
    const int g = u / (8 * NCH), r = u - g * (8 * NCH);
    return 8 * g + (r & 7);
  };
  auto unit_qb = [&](int u) -> int {
    if (NCH == 1) return 0;
    if (u >= nfull) return u % NCH;
    return (u % (8 * NCH)) >> 3;
  };
  auto issue_chunk = [&](int u, int j, int pb) {
    const int bh = unit_bh(u);
    const char* kg = (const char*)(a.k + ((size_t)bh * ST + (size_t)j * S) * MV_HEAD_DIM);
    const char* vg = (const char*)(a.vt + (size_t)bh * MV_HEAD_DIM * ST + (size_t)j * S);
    char* kb = smem + pb * BUF;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int p = 4 * wave + x;
      glds16((const half_t*)(kg + p * 1024 + srcK[x & 1]), kb + p * 1024);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int p = 4 * wave + x;  // 64-key block p >> 3, head dims 8 (p & 7) ..+7
      glds16((const half_t*)(vg + (size_t)(8 * (p & 7)) * (2 * ST) + (p >> 3) * 128 + srcV[x & 1]), kb + VOFF + p * 1024);
    }
    if constexpr (VLO) {
      const char* vlg = (const char*)(a.vt_lo + (size_t)bh * MV_HEAD_DIM * ST + (size_t)j * S);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int p = 4 * wave + x;
        glds16((const half_t*)(vlg + (size_t)(8 * (p & 7)) * (2 * ST) + (p >> 3) * 128 + srcV[x & 1]), kb + VLOFF + p * 1024);
      }
      const char* klg = (const char*)(a.k_lo + ((size_t)bh * ST + (size_t)j * S) * MV_HEAD_DIM);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int p = 4 * wave + x;
        glds16((const half_t*)(klg + p * 1024 + srcK[x & 1]), kb + KLOFF + p * 1024);
      }
    }
  };
  auto load_q = [&](int u, half8_t (&qf)[4]) {  // B operand of S^T = K Q^T: lane holds Q[qb S + 32 wave + ql][16 kk + 8 hi ..+7]
    const int bh = unit_bh(u), qb = unit_qb(u);
    const half_t* gq = a.q + ((size_t)bh * ST + qb * S + 32 * wave + ql) * MV_HEAD_DIM + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const half8_t*)(gq + kk * 16);
  };
  // special rows (AttnArgs::vlo_sp): the A operand of ONE more MFMA per 32 head dims — lane (dim 32 dt + ql, hi = 0) holds {V_lo[key 0][dim], V_lo[key 1][dim], 0 x 6}
  // (2^11 x), lanes hi = 1 (keys 8 .. 15 of the k-slot group) zeros — against the P^T fragment of keys 0 .. 15
  constexpr bool SPV = X8 && !VLO;
  auto load_vsp = [&](int u, uint32_t (&v)[2]) {
    const uint32_t* g = (const uint32_t*)a.vlo_sp + (size_t)unit_bh(u) * MV_HEAD_DIM + ql;
    v[0] = g[0];  // (raw: the zeros of the hi = 1 lanes are selected where the operand is USED — any use here makes hipcc wait for the load, and for the
    v[1] = g[32];  //  K / V^T / Q loads in flight before it, in the middle of the unit: +15 us per launch)
  };
  auto load_q_lo = [&](int u, half8_t (&qf)[4]) {  // VLO: the same fragments of Q's lo plane
    const int bh = unit_bh(u), qb = unit_qb(u);
    const half_t* gq = a.q_lo + ((size_t)bh * ST + qb * S + 32 * wave + ql) * MV_HEAD_DIM + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const half8_t*)(gq + kk * 16);
  };

  // ---- fragment read offsets
  const int pq = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);  // pi(ql): bits 2 and 3 swapped
  uint32_t koff[4], voff[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    koff[x] = (uint32_t)(pq * 128 + (((2 * x + hi) ^ ((pq >> 1) & 7)) << 4));
    voff[x] = (uint32_t)(ql * 128 + (((2 * x + hi) ^ ((ql >> 1) & 7)) << 4));
  }
  // O image (reuses the K half of the ring): row q = 32 wave + ql, 16-B slot s at s ^ (q & 7)
  const uint32_t o_wr = (uint32_t)((32 * wave + ql) * 128 + 8 * hi);
  const uint32_t o_rd = (uint32_t)((32 * wave + (lane >> 3)) * 128 + (((lane & 7) ^ (lane >> 3)) << 4));

  const int first = blockIdx.x, stride = gridDim.x;
  if (first >= nunits) return;
  half8_t qf[4], qn[4];
  half8_t qfl[VLO ? 4 : 1], qnl[VLO ? 4 : 1];  // VLO: Q's lo fragments, prefetched like Q's
  uint32_t vs[2] = {0u, 0u}, vsn[2] = {0u, 0u};  // SPV: this unit's / the next unit's special-row V_lo operand, prefetched like Q
  constexpr bool spv = SPV;  // (the engine hands every X8 launch that is not a two-plane short pass the special rows' V_lo: no run-time test, no branch around the prefetch)
  issue_chunk(first, 0, 0);
  load_q(first, qn);
  if constexpr (spv) load_vsp(first, vsn);
  if constexpr (VLO) load_q_lo(first, qnl);
  int len_n = a.lens[unit_bh(first) / MV_HEADS];  // prefetched like Q: a VGPR-destination load must never be waited for mid-unit

  uint32_t opk[2][4][2];  // normalised O^T of the previous unit, fp16 pairs: [dt][rg] = dims 32 dt + 8 rg + 4 hi ..+3
  uint32_t op8[X8 ? 2 : 1][4][2];  // X8: [dt][rg][0] = lo8, [1] = hi8 of the same four dims, one byte each
  auto flush_plane = [&](int u, char* kb, const uint32_t (&pk)[2][4][2], half_t* base) {
    // ---- O(u) -> LDS image (this wave's 32 rows) -> whole-row global stores
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        u32x2 v;
        v[0] = pk[dt][rg][0]; v[1] = pk[dt][rg][1];
        *(u32x2*)(kb + (o_wr ^ (uint32_t)(((4 * dt + rg) ^ (ql & 7)) << 4))) = v;
      }
    const int bh = unit_bh(u), qb = unit_qb(u);
    const int b = bh / MV_HEADS, h = bh - b * MV_HEADS;
    half_t* dst = base + ((size_t)b * ST + qb * S + 32 * wave + (lane >> 3)) * MV_HIDDEN + h * MV_HEAD_DIM + 8 * (lane & 7);
    u32x4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) v[it] = *(const u32x4*)(kb + o_rd + it * 1024);
#pragma unroll
    for (int it = 0; it < 4; ++it) *(u32x4*)(dst + (size_t)(8 * it) * MV_HIDDEN) = v[it];
  };
  // X8: the fp8 planes through the same image: row q = [lo8 of dims 0..63 | hi8 of dims 0..63], i.e. 16-B slot
  // 2 dt + (rg >> 1) (+ 4 for hi8) at byte 8 (rg & 1) + 4 hi; read back as whole rows, stored as two 64-B segments per row
  auto flush_x8 = [&](int u, char* kb, int len_u) {
    const uint32_t o_wr8 = (uint32_t)((32 * wave + ql) * 128 + 4 * hi);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          *(uint32_t*)(kb + (o_wr8 ^ (uint32_t)(((4 * pl + 2 * dt + (rg >> 1)) ^ (ql & 7)) << 4)) + 8 * (rg & 1)) = op8[dt][rg][pl];
    const int bh = unit_bh(u), qb = unit_qb(u);
    const int b = bh / MV_HEADS, h = bh - b * MV_HEADS;
    const int slot = lane & 7;
    uint8_t* dst = a.ctx8 + ((size_t)b * ST + qb * S + 32 * wave + (lane >> 3)) * (2 * MV_HIDDEN) + (slot >> 2) * MV_HIDDEN +
                   h * MV_HEAD_DIM + 16 * (slot & 3);
    u32x4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) v[it] = *(const u32x4*)(kb + o_rd + it * 1024);
    // (AttnArgs::lo8_min_len: a sequence long enough for the [CLS]-row form keeps no lo8 plane: lanes of slots 0 .. 3 — the lo8 half of the image rows — store nothing)
    if (!(a.lo8_min_len > 0 && len_u >= a.lo8_min_len && slot < 4)) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *(u32x4*)(dst + (size_t)(8 * it) * (2 * MV_HIDDEN)) = v[it];
    }
    // special rows (AttnArgs::sp_lo_out): the low parts of the context rows of queries 0 and 1 of the sequence, compact, taken from the lo8 bytes this pass has
    // just laid into the image (rows 0 and 1 of wave 0 in query block 0; 32 lanes x one dword): e4m3((x - fp16(x)) 2^(11 + shift)) -> fp16, 2^11 x the low part.
    // (Formed from the fp32 context in the unit's last phase — by wave 0 alone, with every other wave waiting at the hand-over — it cost the launch 17 us.)
    if (a.sp_lo_out && wave == 0 && qb == 0 && lane < 32) {
      const int r = lane >> 4, w = lane & 15;
      const uint32_t l8 = *(const uint32_t*)(kb + r * 128 + ((((uint32_t)(w >> 2)) ^ (uint32_t)r) << 4) + (w & 3) * 4);
      constexpr float SC = 1.0f / (float)(1 << MV_X8_ACT_SHIFT);
      const float2_t p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(l8, 1.0f, false), q = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(l8, 1.0f, true);
      half4_t y;
      y[0] = (half_t)(p.x * SC); y[1] = (half_t)(p.y * SC); y[2] = (half_t)(q.x * SC); y[3] = (half_t)(q.y * SC);
      *(half4_t*)(a.sp_lo_out + (size_t)(2 * b + r) * MV_HIDDEN + h * MV_HEAD_DIM + 4 * w) = y;
    }
  };
  auto flush_o = [&](int u, char* kb, int len_u) {
    flush_plane(u, kb, opk, a.ctx);
    if constexpr (X8) {
      // the image rows are wave-private and LDS executes a wave's instructions in order: the second pass's writes may
      // follow the first plane's reads directly (hipcc waits for the read results before the global stores use them)
      flush_x8(u, kb, len_u);
    }
  };

  int pb = 0, prev = -1, len = 0, len_prev = 0;
  floatx16 o[2];
  float m_run = 0.f, l_run = 0.f;  // NCH > 1: running row maximum / this lane's share of the running row sum
  float c_run = 0.f, csp_run = 0.f;  // X8, wave 0: this lane's share of sum e^2 over the keys / over keys 0 and 1 (AttnArgs::conc)
  for (int unit = first; unit < nunits; unit += stride) {
    const int nxt = unit + stride;
#pragma unroll 1  // one body for every chunk: unrolled, the two copies of a 200-register body spill
    for (int j = 0; j < NCH; ++j, pb ^= 1) {
      char* kb = smem + pb * BUF;
      // ---- hand-over: this chunk's K / V^T (and, for j = 0, Q) have landed for every wave, and every wave has left
      // the other ring half (its last reads were the previous chunk's PV)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      if (j == 0) {
        // the prefetched registers are consumed HERE (hipcc would otherwise put its own `s_waitcnt vmcnt(0)` in front
        // of their first use, i.e. after the next chunk's DMA has been issued, and drain it)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]), "+v"(len_n));
        if constexpr (VLO) asm volatile("" : "+v"(qnl[0]), "+v"(qnl[1]), "+v"(qnl[2]), "+v"(qnl[3]));
        if constexpr (SPV) asm volatile("" : "+v"(vsn[0]), "+v"(vsn[1]));
#endif
        len = __builtin_amdgcn_readfirstlane(len_n);
        // previous unit's O through the K half of the OTHER ring slot: rows 32 wave .. + 31 are exactly the rows this
        // wave's own K pieces of the next chunk will overwrite, so the only ordering needed is this wave's lgkmcnt(0)
        if (prev >= 0) {
          flush_o(prev, smem + (pb ^ 1) * BUF, len_prev);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
        vs[0] = vsn[0]; vs[1] = vsn[1];
        if constexpr (VLO) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) qfl[kk] = qnl[kk];
        }
      }
      if (j + 1 < NCH) issue_chunk(unit, j + 1, pb ^ 1);
      else if (nxt < nunits) issue_chunk(nxt, 0, pb ^ 1);

      // ---- S^T = K Q^T over the chunk: st[t][r] = score(query ql, key j S + 32 t + 16 (r >> 3) + 8 hi + 4 ((r >> 2) & 1) + (r & 3))
      floatx16 st[NT];
      {
        // fragments of key block t + 1 are requested before the four MFMAs of block t (sched_barrier pins the order:
        // left alone, hipcc serialises read -> wait -> MFMA through one register set)
        half8_t kf[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[0][kk] = *(const half8_t*)(kb + koff[kk]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NT) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kf[(t + 1) & 1][kk] = *(const half8_t*)(kb + (t + 1) * 4096 + koff[kk]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t & 1][kk], qf[kk], st[t], 0, 0, 0);
          if constexpr (VLO) {  // + K_hi Q_lo + K_lo Q_hi (the small terms after the large one; K's lo fragments read here: the short passes are not LDS-latency-bound)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t & 1][kk], qfl[kk], st[t], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const half8_t kl = *(const half8_t*)(kb + KLOFF + t * 4096 + koff[kk]);
              st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf[kk], st[t], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (len < (j + 1) * S) {  // wave-uniform branch: padded keys get the reference's additive mask
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");  // keeps hipcc from if-converting the block into 128 always-executed selects
#endif
        const int thr = len - j * S - 8 * hi;  // key(t, r) = j S + 32 t + 16 (r >> 3) + 4 ((r >> 2) & 1) + (r & 3) + 8 hi
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            st[t][r] += (32 * t + 16 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3) >= thr) ? -10000.0f : 0.0f;
      }
      float mx = st[0][0];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
      {  // the other 8-key half of every k-slot group lives in lane ^ 32: one v_permlane32_swap gives both halves
        const auto sw = __builtin_amdgcn_permlane32_swap(f2u(mx), f2u(mx), false, false);
        mx = fmaxf(u2f(sw[0]), u2f(sw[1]));
      }
      float alpha = 1.0f;  // NCH > 1, j > 0: what the previous chunks' O and row sum are rescaled by
      if (j > 0) {
        const float m_new = fmaxf(m_run, mx);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        mx = m_new;
      }
      m_run = mx;
      const float nm = -mx * LOG2E;
      float ps4[4] = {0.f, 0.f, 0.f, 0.f};  // four independent partial sums (a single chain is 128 dependent adds)
      half8_t pf[NT][2];
      half8_t pfl[VLO ? NT : 1][2];  // VLO: P's second fp16 plane
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_fmaf(st[t][r], LOG2E, nm);
          const float p = __builtin_amdgcn_exp2f(e);
          ps4[r & 3] += p;
          const half_t ph = (half_t)p;
          pf[t][r >> 3][r & 7] = ph;
          if constexpr (VLO) pfl[t][r >> 3][r & 7] = (half_t)(p - (float)ph);
        }
      {
        const float psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
        l_run = (j > 0) ? __builtin_fmaf(l_run, alpha, psum) : psum;
      }
      if constexpr (X8) {
        // concentration monitor (AttnArgs::conc): sum_j p[q][j]^2 of this lane's query over this chunk's keys, from the packed fp16 probabilities, by the first
        // wave of the sequence's first query block alone (64 v_dot2_f32_f16: lanes 0 and 32 hold query 0 = the [CLS] row)
        if (a.conc && wave == 0 && unit_qb(unit) == 0) {
          float s2 = 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const half2_t h2 = {pf[t][u][2 * e], pf[t][u][2 * e + 1]};
                s2 = __builtin_amdgcn_fdot2(h2, h2, s2, false);
              }
          const float e0 = (float)pf[0][0][0], e1 = (float)pf[0][0][1];  // keys 0 and 1 (lanes of hi = 0, first chunk)
          const float sp = (j == 0 && hi == 0) ? __builtin_fmaf(e0, e0, e1 * e1) : 0.f;
          if constexpr (NCH == 1) {  // the whole row is here: finish now (nothing of it lives across the P V phase: two more live values cost the S = 256 build 150 B of scratch)
            const auto cs = __builtin_amdgcn_permlane32_swap(f2u(s2 - sp), f2u(s2 - sp), false, false);
            const auto ls = __builtin_amdgcn_permlane32_swap(f2u(l_run), f2u(l_run), false, false);
            const float linv = 1.0f / (u2f(ls[0]) + u2f(ls[1]));
            const float coll = (u2f(cs[0]) + u2f(cs[1])) * linv * linv;
            if (lane == 0 && len >= 16) {  // query 0 of the sequence (a sequence of a handful of tokens concentrates by construction: not what is looked for)
              atomicMax(a.conc, (unsigned long long)f2u(fmaxf(coll, 0.f)));
              if (coll > 0.25f) atomicAdd(a.conc + 1, 1ull);
              atomicAdd(a.conc + 2, 1ull);
            }
          } else {
            c_run = (j > 0) ? __builtin_fmaf(c_run, alpha * alpha, s2) : s2;
            csp_run = (j > 0) ? __builtin_fmaf(csp_run, alpha * alpha, sp) : sp;
          }
        }
      }
      // next unit's Q fragments: issued in the unit's last chunk (the score registers are dead), they land under its PV phase
      if (j == NCH - 1 && nxt < nunits) {
        if constexpr (VLO) load_q_lo(nxt, qnl);
        load_q(nxt, qn);
        if constexpr (spv) load_vsp(nxt, vsn);
        len_n = a.lens[unit_bh(nxt) / MV_HEADS];
      }
      // ---- O^T[d][q] (+)= V^T[d][keys] P^T[keys][q]
      if (j == 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        if constexpr (SPV) {
          // special rows: O^T starts from V_lo[keys 0, 1]^T P^T[keys 0, 1] — keys 0 .. 15 are the first k-slot group of the first score fragment.  The V_lo operand
          // is 2^11 x the low parts (normal fp16 numbers); the product is scaled back in fp32 before the sum over all keys is added on top.  (Scaling the P
          // fragment by 2^-11 in fp16 instead — four packed multiplies for 32 — turns probabilities below 2^-3 into subnormals: measured 6.3e-4 instead of
          // 3.5e-4 on a golden, for 2 us.)  As the accumulator's START value the term costs no live range (added after the P V loop it cost the S = 256
          // instantiation 70 VGPRs and 320 B of scratch); in all +3 .. 4 us per launch at S = 256 (profiles/r06_b_*).
          if constexpr (spv) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              const u32x4 av = {hi ? 0u : vs[dt], 0u, 0u, 0u};
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, av), pf[0][0], o[dt], 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 16; ++r) o[dt][r] *= 1.0f / 2048.0f;
            }
          }
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      {
        half8_t vf[2][4];  // [parity][2 dt + u] of one 32-key fragment t
        auto read_v = [&](int t, half8_t (&dstf)[4]) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              dstf[2 * dt + u] = *(const half8_t*)(kb + VOFF + (t >> 1) * 8192 + dt * 4096 + voff[2 * (t & 1) + u]);
        };
        read_v(0, vf[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NT) read_v(t + 1, vf[(t + 1) & 1]);
          half8_t vl[VLO ? 4 : 1];  // VLO: the lo plane's fragments of key block t
          if constexpr (VLO) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
              for (int u = 0; u < 2; ++u)
                vl[2 * dt + u] = *(const half8_t*)(kb + VLOFF + (t >> 1) * 8192 + dt * 4096 + voff[2 * (t & 1) + u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t & 1][2 * dt + u], pf[t][u], o[dt], 0, 0, 0);
              if constexpr (VLO) {
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[2 * dt + u], pf[t][u], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t & 1][2 * dt + u], pfl[t][u], o[dt], 0, 0, 0);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (j == NCH - 1) {
        const auto sw = __builtin_amdgcn_permlane32_swap(f2u(l_run), f2u(l_run), false, false);
        const float inv = 1.0f / (u2f(sw[0]) + u2f(sw[1]));
        if constexpr (X8 && NCH > 1) {
          if (a.conc && wave == 0 && unit_qb(unit) == 0) {
            const float own = c_run - csp_run;  // ordinary keys of this lane's half
            const auto cs = __builtin_amdgcn_permlane32_swap(f2u(own), f2u(own), false, false);
            const float coll = (u2f(cs[0]) + u2f(cs[1])) * inv * inv;
            if (lane == 0 && len >= 16) {  // query 0 of the sequence (a sequence of a handful of tokens concentrates by construction: not what is looked for)
              atomicMax(a.conc, (unsigned long long)f2u(fmaxf(coll, 0.f)));
              if (coll > 0.25f) atomicAdd(a.conc + 1, 1ull);
              atomicAdd(a.conc + 2, 1ull);
            }
          }
        }
        float vmax8 = 0.f;  // MV_F16X8: max |context value| of the unit (saturation accounting, common.h)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float x0 = o[dt][4 * rg + 0] * inv, x1 = o[dt][4 * rg + 1] * inv, x2 = o[dt][4 * rg + 2] * inv, x3 = o[dt][4 * rg + 3] * inv;
            opk[dt][rg][0] = pack_h2(x0, x1);
            opk[dt][rg][1] = pack_h2(x2, x3);
            if constexpr (X8) {
              x8_planes4_in_range_packed(x0, x1, x2, x3, opk[dt][rg][0], opk[dt][rg][1], op8[dt][rg][1], op8[dt][rg][0]);
              vmax8 = x8_absmax4(vmax8, x0, x1, x2, x3);
            }
          }
        if constexpr (X8) {
          if (x8_any_out_of_range(vmax8)) {  // rare: redo the unit's planes with the clamps, count exactly
            int n = 0;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const float x0 = o[dt][4 * rg + 0] * inv, x1 = o[dt][4 * rg + 1] * inv, x2 = o[dt][4 * rg + 2] * inv, x3 = o[dt][4 * rg + 3] * inv;
                x8_planes4(x0, x1, x2, x3, op8[dt][rg][1], op8[dt][rg][0]);
                n += x8_count4(x0, x1, x2, x3);
              }
            x8_sat_add(a.x8_sat, n);
          }
        }
        prev = unit;
        len_prev = len;
      }
    }
  }
  // ---- last unit's O: the K half of the slot no DMA was issued into (nothing reads it any more)
  flush_o(prev, smem + pb * BUF, len_prev);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}
