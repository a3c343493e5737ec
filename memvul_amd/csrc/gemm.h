// MFMA GEMM for the BERT projections:
//   C[M,N] = A[M,K] (fp16) x W[N,K]^T (fp16, torch Linear.weight layout) + bias, fp32 accumulate,
// with the reference's elementwise work fused into the epilogue.
//
// v1 structure ("128^2 tile, LDS-DMA, one barrier per K-step", cdna_hip_programming.md §5):
//   * 256 threads = 4 waves as 2(M) x 2(N); each wave owns a 64x64 sub-tile = 2x2 fragments of
//     v_mfma_f32_32x32x16_f16 (64 accumulator VGPRs).
//   * K-step 64: the A and W tiles are both [128 rows][64 halfs] = 128-B rows in LDS, double-buffered
//     (2 x 32 KB => two workgroups per CU).  They are filled by global_load_lds_dwordx4 (16 B per lane,
//     1 KiB per wave-instruction, LDS image lane-linear), so the bank-conflict swizzle is applied on the
//     per-lane SOURCE address and again on the ds_read_b128 (rule 21): 16-B chunk c of row r lives at
//     chunk slot c ^ ((r >> 1) & 7); with that, the four 16-lane groups of a ds_read_b128 fragment read
//     (32 rows x one chunk) hit 16 distinct bank slots.
//   * blockIdx -> tile through xcd_remap(): each XCD sweeps a contiguous run of tiles, N fastest, so the
//     A row panel and the (small) W matrix stay in that XCD's L2.
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
  float* outf;        // EPI_F32: [M][N]
  half_t* out16;      // EPI_GELU: [M][N]
  float* xres;        // EPI_RES: [M][N] residual stream, updated in place
  half_t* q;          // EPI_QKV: [B][12][S][64]  (W_q, b_q pre-scaled by 1/8)
  half_t* k;          //          [B][12][S][64]
  half_t* vt;         //          [B][12][64][S]
  int S;              // padded sequence length (multiple of 64)
};

#define G128_BM 128
#define G128_BN 128
#define G128_BK 64
#define G128_TILE_BYTES (128 * 64 * 2)          // 16 KiB per operand tile
#define G128_LDS_BYTES (4 * G128_TILE_BYTES)    // 2 buffers x (A + W)

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int tn = a.N >> 7;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tn;
  const int tile_n = tile - tile_m * tn;
  const int m0 = tile_m << 7, n0 = tile_n << 7;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = a.K;

  // ---- staging: wave w fills slabs w*4 .. w*4+3 (8 rows x 128 B each) of both operand tiles.
  // lane -> (row = slab*8 + lane/8, chunk slot = lane%8); source chunk = slot ^ ((row>>1)&7).
  const half_t* srcA[4];
  const half_t* srcW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slab = wave * 4 + i;
    const int row = slab * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((row >> 1) & 7);
    srcA[i] = a.A + (size_t)(m0 + row) * K + sc * 8;
    srcW[i] = a.W + (size_t)(n0 + row) * K + sc * 8;
  }

  auto stage_glds = [&](int buf, int kt) {
    char* baseA = smem + buf * (2 * G128_TILE_BYTES);
    char* baseW = baseA + G128_TILE_BYTES;
    const int koff = kt * G128_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int slab = wave * 4 + i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + koff),
                                       (__attribute__((address_space(3))) void*)(baseA + slab * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[i] + koff),
                                       (__attribute__((address_space(3))) void*)(baseW + slab * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside an operand tile)
  const int swz = (lane >> 1) & 7;  // ((row >> 1) & 7) with row = 32*x + (lane & 31)
  const int arow = (wm * 64 + (lane & 31)) * 128;
  const int brow = (wn * 64 + (lane & 31)) * 128;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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

  const int nk = K / G128_BK;
  if constexpr (GLDS) {
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
  } else {
    // register staging: global -> VGPR -> ds_write_b128 (same LDS image as the LDS-DMA path)
    half8_t ra[4], rw[4];
    auto gload = [&](int kt) {
      const int koff = kt * G128_BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *(const half8_t*)(srcA[i] + koff);
        rw[i] = *(const half8_t*)(srcW[i] + koff);
      }
    };
    auto lwrite = [&](int buf) {
      char* baseA = smem + buf * (2 * G128_TILE_BYTES);
      char* baseW = baseA + G128_TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int slab = wave * 4 + i;
        *(half8_t*)(baseA + slab * 1024 + lane * 16) = ra[i];
        *(half8_t*)(baseW + slab * 1024 + lane * 16) = rw[i];
      }
    };
    gload(0);
    lwrite(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) gload(kt + 1);
      compute(cur);
      if (kt + 1 < nk) lwrite(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue straight from the accumulator layout: for register r a half-wave covers 32
  // consecutive columns of one row (col = lane & 31, row = mfma32_row(r, hi)).
  const int col_in = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mb = m0 + wm * 64 + i * 32;  // first row of this 32-row fragment
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + col_in;
      const float bias = a.bias ? a.bias[n] : 0.f;
      if constexpr (EPI == EPI_F32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + mfma32_row(r, hi);
          a.outf[(size_t)m * a.N + n] = acc[i][j][r] + bias;
        }
      } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + mfma32_row(r, hi);
          a.out16[(size_t)m * a.N + n] = (half_t)gelu_erf(acc[i][j][r] + bias);
        }
      } else if constexpr (EPI == EPI_RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + mfma32_row(r, hi);
          float* p = a.xres + (size_t)m * a.N + n;
          *p = acc[i][j][r] + bias + *p;
        }
      } else {  // EPI_QKV
        const int which = n0 / MV_HIDDEN;                          // 0 q, 1 k, 2 v (block-uniform)
        const int hn0 = n0 - which * MV_HIDDEN + wn * 64;          // first column of this wave inside [0,768)
        const int head = hn0 >> 6;                                 // wave-uniform
        const int d = j * 32 + col_in;
        const int b = mb / a.S;                                    // S % 64 == 0 and mb % 32 == 0: one b per fragment
        const int s_base = mb - b * a.S;
        if (which < 2) {
          half_t* dst = (which == 0 ? a.q : a.k) + ((size_t)(b * MV_HEADS + head) * a.S) * MV_HEAD_DIM + d;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = mfma32_row(r, hi);
            if (mb + rr < a.Mreal) dst[(size_t)(s_base + rr) * MV_HEAD_DIM] = (half_t)(acc[i][j][r] + bias);
          }
        } else {
          half_t* dst = a.vt + ((size_t)(b * MV_HEADS + head) * MV_HEAD_DIM + d) * a.S;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int rr = 8 * rg + 4 * hi;  // rows rr..rr+3 are registers 4*rg..4*rg+3
            half4_t v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = (half_t)(acc[i][j][4 * rg + e] + bias);
            if (mb + rr < a.Mreal) *(half4_t*)(dst + s_base + rr) = v4;
          }
        }
      }
    }
  }
}
