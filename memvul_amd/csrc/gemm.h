// MFMA GEMM for the BERT projections:
//   C[M,N] = A[M,K] (fp16) x W[N,K]^T (fp16, torch Linear.weight layout) + bias, fp32 accumulate,
// with the reference's elementwise work fused into the epilogue.
//
// Common to both tile shapes below (v_mfma_f32_32x32x16_f16, wave64):
//   * operand tiles are [rows][64 halfs] = 128-B rows in LDS, double-buffered, filled by
//     global_load_lds_dwordx4 (16 B per lane, 1 KiB per wave-instruction, LDS image lane-linear), so the
//     bank-conflict swizzle is applied on the per-lane SOURCE address and again on the ds_read_b128
//     (cdna_hip_programming.md §5.4 rule 21): 16-B chunk c of row r lives at chunk slot c ^ ((r >> 1) & 7);
//     with that, the four 16-lane groups of a ds_read_b128 fragment read (32 rows x one chunk) hit 16
//     distinct bank slots.
//   * blockIdx -> tile through xcd_remap() + a grouped raster: the logical tile sequence is
//     (column group of GN tiles) > tile_m > tile_n-in-group and each XCD takes one contiguous run of it.
//     The W panels of a group (GN x rows x K x 2 B, ~1.5 MB) stay in that XCD's 4 MiB L2 while it sweeps the
//     A row panels; with the plain N-fastest order W (4.7 MB for FFN-1) thrashed the L2 and FETCH_SIZE showed
//     1.5 GB per FFN-1 launch against 105 MB of operands (profiles/r01_a_pmc_hbm_v1.txt).
//   * the accumulation order over K is the same in every variant (ascending 16-wide MFMA steps), so the
//     128^2 and 256^2 kernels produce bit-identical results.
//
// gemm128: 256 threads = 4 waves as 2(M) x 2(N), 64x64 per wave, 2 x 32 KB LDS => two workgroups per CU;
//          one barrier per K-step ("step-3" structure of the guide).  The mid-size path (passes too small to fill the
//          chip with 256^2 tiles); the bench-scale path is gemm_pp.h, the skinny-M path ([CLS] tail) gemm_ring below.
//          (The round-1/2 library also carried a 256^2 one-tile-per-workgroup kernel, a register-staged 128^2 form and a
//          sweep of ring geometries: retired (git history), A/B records in profiles/r01_d_gemm_ablation.txt.)
#pragma once
#include "common.h"

enum { EPI_F32 = 0, EPI_QKV = 1, EPI_GELU = 2, EPI_RES = 3 };

struct GemmArgs {
  const half_t* A;    // [Mpad][K]
  const half_t* W;    // [N][K]
  const float* bias;  // [N]
  int M;              // rows to compute (multiple of the M tile)
  int Mreal;          // rows that exist (scatter epilogues skip the rest)
  int N, K;
  int GN;             // raster group width in tiles (divides N / tile)
  float* outf;        // EPI_F32: [M][N]
  half_t* out16;      // EPI_GELU: [M][N]; PP_RESLN3: the hi plane of the raw stream (the next consumer's fp16 operand)
  half_t* out16b;     // PP_RESLN3, MV_F16: the lo plane, fp16(r - hi).  MV_F16X8 (round 6) has no lo fp16 plane: the residual's low part is read back from the lo8 plane of
                      // out8 (in place: r ~= hi + lo8 2^-(11 + shift), 2^-15 of the element), and the SPECIAL rows' — whose stream reaches the pooler un-averaged — from
                      // sp_lo_out (hi + lo, compact): 5 of the 19 bytes per element the two residual GEMMs of a layer moved.  Round 5's lo8 stream applied to EVERY row
                      // cost 3.0 -> 3.5e-4 on the median; model with the special rows kept as hi + lo (scripts/r06_stream_model.py): the two-plane stream's error
  float* xres;        // EPI_RES: [M][N] residual stream, updated in place
  half_t* q;          // EPI_QKV: [B][12][S][64]  (W_q, b_q pre-scaled by 1/8)
  half_t* k;          //          [B][12][S][64]
  half_t* vt;         //          [B][12][64][S]
  int S;              // padded sequence length (multiple of 64)
  int col0;           // EPI_QKV / PP_QK: packed-QKV column of this launch's first output column (0, or 768 = K,V only)
  const float* lnstats;  // gemm_pp: the rows' "vstats" [M][3][2] of the LayerNorm in front of this GEMM (common.h ln_from_partials);
                         // lng / lnb: that LayerNorm's gamma, beta [768] (PP_RESLN3)
  const float* lng;
  const float* lnb;
  float* lnpart;         // PP_RESLN3 (N = 768): vstats of the NEW raw rows, [M][3][2] = (sum, sum of squares) per row and 256-column tile
  float ln_eps;          // LayerNorm epsilon of the vstats consumers
  // MV_F16X8 (gemm_pp.h X8): the fp8 (OCP e4m3) correction sweep  2^-s (A_lo8 W_hi8 + A_hi8 W_lo8)  over a virtual K of 2 K
  const uint8_t* A8;     // [Mpad][2 K]: row = [lo8 (K bytes) | hi8 (K bytes)] of the A operand, pre-scaled by 2^(11 + sa) / 2^sa
  const uint8_t* W8;     // [N][2 K]:    row = [hi8 | lo8] of W, pre-scaled by 2^sw / 2^(11 + sw)
  uint8_t* out8;         // PP_GELU / PP_RESLN3: [M][2 N] = [lo8 | hi8] planes of this GEMM's output (the next GEMM's A8)
  int x8_scale;          // E8M0 byte of 2^-(11 + sa + sw), replicated in the four bytes (the other scale operand is 1.0)
  int x8_terms;          // 0 / 2: both first-order terms; 1: only A_hi8 W_lo8 (the second halves of the rows: K bytes, K / 128 K-tiles);
                         // 3 (PP_QK): per Q / K / V block by x8_aside_mask
  int x8_aside_mask;     // x8_terms == 3: bit 0 / 1 / 2 set = the Q / K / V block sweeps both terms, clear = the weight-side term only
  half_t* vt_lo;         // PP_QK X8, short passes (padded length <= 128): second fp16 plane of V^T, fp16(V - fp16(V)), same layout as vt (attention_v2.h VLO);
  half_t *q_lo, *k_lo;   // the same for Q and K (set together with vt_lo)
  unsigned long long* x8_sat;  // MV_F16X8 producers: device counter of activation elements the fp8 planes' +-112 clamp changed (common.h x8_planes4)
  int raster_mode;       // gemm_pp: 0 = column group > tile_m > tile_n; 1 = "A-stationary": consecutive persistent iterations of a workgroup
                         // keep its tile_m and walk the column groups (the XCD's A panels stay in its L2 across the whole N sweep)
  int out8_hi_only;      // gemm_pp X8 producers (PP_GELU / PP_RESLN3), cls_aside: the consumer of out8 sweeps the weight-side term only — write the hi8
                         // plane alone (the special rows' low parts travel through sp_lo_out)
  const int* tile_both;  // gemm_pp X8, cls_aside: [M / 256] flags, non-zero = a sequence of that row tile is shorter than MEMVUL_CLS_ASIDE_MIN_LEN: the tile runs
                         // the default form (both terms where x8_terms = 1, no cls_corr, full planes); cls_tile_flags_kernel writes them once per pass
  const float* cls_corr; // gemm_pp X8, the row term of the SPECIAL ROWS (round 6; engine.hip): [2 ceil(M / S)][N] fp32 = 2^11 x the A-side first-order term
                         // A_lo W_hi^T of rows b S ([CLS]) and b S + 1 ([SEP]: the embedding kernel computes the last token there) of every sequence, from a
                         // skinny fp16 GEMM in front of this launch; added to those rows' accumulators before the epilogue in every tile whose sweep
                         // carried the weight-side term only
  half_t* sp_lo_out;     // gemm_pp X8 producers (PP_GELU / PP_RESLN3): 2^11 x the low parts (v - fp16(v)) of the special rows of THIS launch's output, compact
                         // [2 b + row][N] fp16 — the A operand of the next GEMM's row term (no gather launch, no lo8 rows kept for it)
  half_t* vlo_sp;        // PP_QK X8: 2^11 x the low parts of V of the special rows, [b 12 + head][64 dims][2 rows] fp16: attention adds p[:, 0..1] V_lo[0..1]
                         // (attention_v2.h): with attention sinks V of the sink token reaches every row's context un-averaged
};

// logical tile index -> (tile_m, tile_n) under the grouped raster
__device__ __forceinline__ void raster(int t, int tm_count, int tn_count, int GN, int& tile_m, int& tile_n) {
  const int per_group = tm_count * GN;
  const int g = t / per_group;
  const int r = t - g * per_group;
  tile_m = r / GN;
  tile_n = g * GN + (r - tile_m * GN);
}

// Epilogue of one 32x32 accumulator fragment whose top-left element is (mb, nb): for register r a
// half-wave covers 32 consecutive columns of one row (col = lane & 31, row = mfma32_row(r, hi)).
template <int EPI>
__device__ __forceinline__ void epilogue_frag(const GemmArgs& a, const floatx16& acc, int mb, int nb, int lane) {
  const int hi = lane >> 5;
  const int n = nb + (lane & 31);
  // the bias is already in the accumulators (init_frag): every kernel of this library starts the K sum from
  // the bias, so all tile shapes / orientations produce the same bits for a given output element
  if constexpr (EPI == EPI_F32) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a.outf[(size_t)(mb + mfma32_row(r, hi)) * a.N + n] = acc[r];
  } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a.out16[(size_t)(mb + mfma32_row(r, hi)) * a.N + n] = (half_t)gelu_erf(acc[r]);
  } else if constexpr (EPI == EPI_RES) {  // N == 768 (the residual stream)
    const int lane_off = 4 * hi * MV_HIDDEN + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* rowbase = a.xres + (size_t)(mb + (r & 3) + 8 * (r >> 2)) * MV_HIDDEN + nb;
      rowbase[lane_off] = acc[r];
    }
  } else {  // EPI_QKV: n in [0, 2304) = which * 768 + head * 64 + d; a fragment never straddles a head half
    const int which = (nb + a.col0) / MV_HIDDEN;
    const int hn = nb + a.col0 - which * MV_HIDDEN;
    const int head = hn >> 6;
    const int d = (hn & 63) + (lane & 31);
    const int b = mb / a.S;  // S % 64 == 0 and mb % 32 == 0: one batch row per fragment
    const int s_base = mb - b * a.S;
    if (which < 2) {
      half_t* dst = (which == 0 ? a.q : a.k) + ((size_t)(b * MV_HEADS + head) * a.S) * MV_HEAD_DIM + d;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = mfma32_row(r, hi);
        if (mb + rr < a.Mreal) dst[(size_t)(s_base + rr) * MV_HEAD_DIM] = (half_t)acc[r];
      }
    } else {
      half_t* dst = a.vt + ((size_t)(b * MV_HEADS + head) * MV_HEAD_DIM + d) * a.S;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 8 * rg + 4 * hi;  // rows rr..rr+3 are registers 4*rg..4*rg+3
        half4_t v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = (half_t)acc[4 * rg + e];
        if (mb + rr < a.Mreal) *(half4_t*)(dst + s_base + rr) = v4;
      }
    }
  }
}

// Accumulator init for one fragment.  EPI_RES: start from bias + residual instead of zero — the 16 dword
// loads per fragment are issued before the main loop (their latency overlaps the first LDS-DMA stage), so the
// epilogue is store-only; doing the read-modify-write after the loop cost 125-160 us per launch (latency-bound
// batches of dword loads at one workgroup per CU).  The tile is owned by this workgroup, so in-place is safe.
template <int EPI>
__device__ __forceinline__ void init_frag(const GemmArgs& a, floatx16& acc, int mb, int nb, int lane) {
  if constexpr (EPI == EPI_RES) {
    // wave-uniform row base (SGPR) + one 32-bit per-lane offset: no per-load 64-bit VGPR address math
    const int lane_off = 4 * (lane >> 5) * MV_HIDDEN + (lane & 31);
    const float bias = a.bias[nb + (lane & 31)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* rowbase = a.xres + (size_t)(mb + (r & 3) + 8 * (r >> 2)) * MV_HIDDEN + nb;
      acc[r] = rowbase[lane_off] + bias;
    }
  } else {
    const float bias = a.bias ? a.bias[nb + (lane & 31)] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias;
  }
}

__device__ __forceinline__ void glds16(const half_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// =================================================================================================
#define G128_TILE_BYTES (128 * 64 * 2)        // 16 KiB per operand tile
#define G128_LDS_BYTES (4 * G128_TILE_BYTES)  // 2 buffers x (A + W)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  int tile_m, tile_n;
  raster(xcd_remap(blockIdx.x, gridDim.x), a.M >> 7, a.N >> 7, a.GN, tile_m, tile_n);
  const int m0 = tile_m << 7, n0 = tile_n << 7;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = a.K;

  // staging: wave w fills slabs w*4 .. w*4+3 (8 rows x 128 B each) of both operand tiles.
  // lane -> (row = slab*8 + lane/8, chunk slot = lane%8); source chunk = slot ^ ((row>>1)&7).
  const half_t* srcA[4];
  const half_t* srcW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((row >> 1) & 7);
    srcA[i] = a.A + (size_t)(m0 + row) * K + sc * 8;
    srcW[i] = a.W + (size_t)(n0 + row) * K + sc * 8;
  }
  auto stage_glds = [&](int buf, int kt) {
    char* baseA = smem + buf * (2 * G128_TILE_BYTES) + wave * 4096;
    char* baseW = baseA + G128_TILE_BYTES;
    const int koff = kt * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(srcA[i] + koff, baseA + i * 1024);
      glds16(srcW[i] + koff, baseW + i * 1024);
    }
  };

  const int swz = (lane >> 1) & 7;  // ((row >> 1) & 7) with row = 32*x + (lane & 31)
  const int arow = (wm * 64 + (lane & 31)) * 128;
  const int brow = (wn * 64 + (lane & 31)) * 128;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) init_frag<EPI>(a, acc[i][j], m0 + wm * 64 + i * 32, n0 + wn * 64 + j * 32, lane);

  auto compute = [&](int buf) {
    const char* baseA = smem + buf * (2 * G128_TILE_BYTES);
    const char* baseW = baseA + G128_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = ((kk * 2 + hi) ^ swz) << 4;
      half8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *(const half8_t*)(baseA + arow + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const half8_t*)(baseW + brow + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = K / 64;
  stage_glds(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage_glds(cur ^ 1, kt + 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      epilogue_frag<EPI>(a, acc[i][j], m0 + wm * 64 + i * 32, n0 + wn * 64 + j * 32, lane);
}

// =================================================================================================
// gemm_ring: the same MFMA mainloop over a RING of LDS stages with counted `s_waitcnt vmcnt(N)`, so that
// STAGES-1 operand stages are in flight across barriers (cdna_hip_programming.md T3/T4).  PMC on gemm256
// (profiles/r01_c_pmc_*): 0 bank conflicts, LDS 10 % busy, MFMA 28-34 % busy, waves parked ~40 % of the time:
// with one 64-KB stage in flight per CU the K-loop runs at ~12 B/cycle/CU of operand delivery — latency-bound.
//   tile  = (WAVES_M * FR_M * 32) x (WAVES_N * FR_N * 32), K-step BK in {32, 64}
//   stage = [A rows | W rows] x BK halfs, 1-KiB slabs dealt round-robin to the waves
//   loop  : wait(stage kt landed, newer stages stay in flight) -> barrier -> issue stage kt+STAGES-1 into the
//           slot of tile kt-1 (every wave finished reading it before the barrier) -> MFMAs on stage kt.
// The raw s_barrier is fenced with `asm volatile("" ::: "memory")` so no LDS access moves across it.
template <int BK>
struct RingGeom {
  static constexpr int ROW_BYTES = BK * 2;
  static constexpr int ROWS_PER_SLAB = 1024 / ROW_BYTES;      // 8 (BK=64) or 16 (BK=32)
  static constexpr int CHUNKS = ROW_BYTES / 16;               // 8 or 4
  static constexpr int LANES_PER_ROW = CHUNKS;
  __device__ static __forceinline__ int swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, int FR_M, int FR_N, int WAVES_M, int WAVES_N, int BK, int STAGES, int MIN_WAVES_PER_SIMD>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MIN_WAVES_PER_SIMD) void gemm_ring_kernel(GemmArgs a) {
  using Gm = RingGeom<BK>;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * FR_M * 32, BN = WAVES_N * FR_N * 32;
  constexpr int STAGE_BYTES = (BM + BN) * Gm::ROW_BYTES;
  constexpr int SLABS = STAGE_BYTES / 1024;
  static_assert(SLABS % NW == 0, "slabs must deal evenly to the waves");
  constexpr int G = SLABS / NW;  // LDS-DMA instructions per wave per stage
  constexpr int KK = BK / 16;    // MFMA k-substeps per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  int tile_m, tile_n;
  raster(xcd_remap(blockIdx.x, gridDim.x), a.M / BM, a.N / BN, a.GN, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int K = a.K;

  floatx16 acc[FR_M][FR_N];
#pragma unroll
  for (int i = 0; i < FR_M; ++i)
#pragma unroll
    for (int j = 0; j < FR_N; ++j) init_frag<EPI>(a, acc[i][j], m0 + (wm * FR_M + i) * 32, n0 + (wn * FR_N + j) * 32, lane);

  // slab s = wave + i*NW covers stage rows [s*RPS, (s+1)*RPS): rows < BM belong to A, the rest to W
  const half_t* src[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int row = (wave + i * NW) * Gm::ROWS_PER_SLAB + lane / Gm::LANES_PER_ROW;  // row inside the stage image
    const int sc = (lane % Gm::CHUNKS) ^ Gm::swz(row);  // BM % 32 == 0: the swizzle key is the same in A- and W-local rows
    src[i] = (row < BM) ? a.A + (size_t)(m0 + row) * K + sc * 8 : a.W + (size_t)(n0 + row - BM) * K + sc * 8;
  }
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * STAGE_BYTES + wave * 1024;
    const int koff = kt * BK;
#pragma unroll
    for (int i = 0; i < G; ++i) glds16(src[i] + koff, base + i * NW * 1024);
  };

  const int swzl = Gm::swz(lane & 31);
  const int arow = ((wm * FR_M) * 32 + (lane & 31)) * Gm::ROW_BYTES;
  const int brow = (BM + (wn * FR_N) * 32 + (lane & 31)) * Gm::ROW_BYTES;
  auto compute = [&](int slot) {
    const char* base = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int coff = ((kk * 2 + hi) ^ swzl) << 4;
      half8_t fa[FR_M], fb[FR_N];
#pragma unroll
      for (int j = 0; j < FR_N; ++j) fb[j] = *(const half8_t*)(base + brow + j * 32 * Gm::ROW_BYTES + coff);
#pragma unroll
      for (int i = 0; i < FR_M; ++i) fa[i] = *(const half8_t*)(base + arow + i * 32 * Gm::ROW_BYTES + coff);
#pragma unroll
      for (int i = 0; i < FR_M; ++i)
#pragma unroll
        for (int j = 0; j < FR_N; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s);
  int slot = 0, fill = STAGES - 1;  // slot of tile kt; slot to fill next (= slot of tile kt-1)
  for (int kt = 0; kt < nk; ++kt) {
    const int newer = nk - 1 - kt;  // stages issued after tile kt that may stay in flight (at most STAGES-2)
    if (STAGES >= 5 && newer >= 3) wait_vmcnt<(STAGES >= 5 ? 3 : 0) * G>();
    else if (STAGES >= 4 && newer >= 2) wait_vmcnt<(STAGES >= 4 ? 2 : 0) * G>();
    else if (STAGES >= 3 && newer >= 1) wait_vmcnt<(STAGES >= 3 ? 1 : 0) * G>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the slot about to be refilled are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < nk) stage(fill, kt + STAGES - 1);
    compute(slot);
    asm volatile("" ::: "memory");
    slot = (slot + 1 == STAGES) ? 0 : slot + 1;
    fill = (fill + 1 == STAGES) ? 0 : fill + 1;
  }

#pragma unroll
  for (int i = 0; i < FR_M; ++i)
#pragma unroll
    for (int j = 0; j < FR_N; ++j)
      epilogue_frag<EPI>(a, acc[i][j], m0 + (wm * FR_M + i) * 32, n0 + (wn * FR_N + j) * 32, lane);
}
