// Development probe (not part of the product library): does a TWO-workgroups-per-CU persistent GEMM keep the main-loop rate of
// gemm_pp while its tile phases (accumulator init / epilogue: HBM at ~25 GB/s per CU, matrix pipes idle) overlap the other
// workgroup's main loop?  gemm_pp runs ONE 8-wave workgroup per CU (256 x 256 tile, 128 KB operand ring): 19 - 43 % of a tile's
// time is outside the main loop and every attempt to overlap it inside one workgroup failed for lack of LDS.
//
// gemm_pp2 (prototype): 256 threads = 4 waves, tile 128 (M) x 256 (N), wave w -> columns 64 w .. + 63 of all 128 rows (the same
// 128 x 64 wave tile as gemm_pp: 128 accumulator VGPRs, 12 ds_read_b128 per 16 MFMAs); K step 32: a stage = A [128][64 B] +
// W [256][64 B] = 24 KB, three stages = 72 KB -> two workgroups per CU; one barrier per K step (16 MFMAs per wave); LDS-DMA
// fragments register-double-buffered across the barrier; staging with the bank swizzle on the per-lane source address (16-B chunk c of row r at slot c ^ ((r >> 2) & 3)).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_pp2_probe.hip -o tools/gemm_pp2_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "legacy/gemm.h"
#include "legacy/gemm_pp.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define PP2_STAGE 24576
#define PP2_NSTG 3
#define PP2_LDS (PP2_NSTG * PP2_STAGE)

// EPI 0: no stores (main loop alone); 1: out16 = fp16(A W^T + bias), row-per-lane stores (16 B pieces)
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_pp2_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int K = a.K, nk = K >> 5;
  const int tm_count = a.M >> 7, tn_count = a.N >> 8;
  const int ntiles = tm_count * tn_count;
  const int G = gridDim.x;
  const int bslot = xcd_remap(blockIdx.x, G);
  if (bslot >= ntiles) return;

  // ---- staging: a stage = 24 pieces of 1 KiB (16 rows x 64 B): 0..7 A rows 16 p.., 8..23 W rows 16 (p - 8)..; wave w issues
  // pieces w, w + 4, .., w + 20.  Lane L of a piece lands at +16 L: row L >> 2, slot L & 3 -> source chunk slot ^ ((row >> 2) & 3)
  const uint32_t lane_src = (uint32_t)((lane >> 2) * K * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
  int i_it = 0, i_k = 0;  // issue cursor: tile iteration and K step
  const char *iA = nullptr, *iW = nullptr;
  auto set_issue_tile = [&](int it) {
    const int L = it * G + bslot;
    if (L < ntiles) {
      int tm, tn;
      raster(L, tm_count, tn_count, a.GN, tm, tn);
      iA = (const char*)a.A + (size_t)tm * 128 * K * 2;
      iW = (const char*)a.W + (size_t)tn * 256 * K * 2;
    }
  };
  set_issue_tile(0);
  auto issue_stage = [&](int stg) {
    char* dst = smem + stg * PP2_STAGE;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int p = wave + 4 * i;  // wave-uniform
      const char* src = (p < 8 ? iA + (size_t)(16 * p) * K * 2 : iW + (size_t)(16 * (p - 8)) * K * 2) + i_k * 64 + lane_src;
      glds16((const half_t*)src, dst + p * 1024);
    }
    if (++i_k == nk) { i_k = 0; set_issue_tile(++i_it); }
  };

  // ---- fragment read offsets inside a stage (A at +0, W at +8192)
  uint32_t offA[2], offW[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint32_t c = (uint32_t)(((2 * kk + hi) ^ ((l31 >> 2) & 3)) << 4);
    offA[kk] = (uint32_t)(l31 * 64) + c;
    offW[kk] = 8192u + (uint32_t)((64 * wave + l31) * 64) + c;
  }

  // total K steps of this workgroup
  int my_tiles = 0;
  for (int L = bslot; L < ntiles; L += G) ++my_tiles;
  const int total = my_tiles * nk;
  // Fragments are register-double-buffered ACROSS the barrier: step g's MFMAs run on the set read during step g - 1 while
  // the set of step g + 1 is read.  Ring: stage g + 1 (landed, being read), g + 2 and g + 3 in flight (slot of g + 3 = slot of g,
  // whose reads ended before this step's barrier).
  issue_stage(0);
  if (total > 1) issue_stage(1);
  if (total > 2) issue_stage(2);
  half8_t Af[2][4][2], Wf[2][2][2];
  auto read_frags = [&](int gstep, int set) {
    const char* sb = smem + (gstep % PP2_NSTG) * PP2_STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) Wf[set][j][kk] = *(const half8_t*)(sb + offW[kk] + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) Af[set][i][kk] = *(const half8_t*)(sb + offA[kk] + i * 2048);
    }
  };
  // stage 0 landed for everyone, then its fragments
  if (total > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (total > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, 0);

  floatx16 acc[4][2];
  int g = 0;
  auto step = [&](auto setc) {
    constexpr int set = decltype(setc)::value;
    // stage g + 1 has landed for this wave once at most the 6 pieces of stage g + 2 are outstanding
    if (g + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave has finished reading stage g
    __builtin_amdgcn_sched_barrier(0);
    if (g + 3 < total) issue_stage((g + 3) % PP2_NSTG);
    if (g + 1 < total) read_frags(g + 1, set ^ 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[set][j][kk], Af[set][i][kk], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    ++g;
  };
  for (int it = 0; it < my_tiles; ++it) {
    const int L = it * G + bslot;
    int tile_m, tile_n;
    raster(L, tm_count, tn_count, a.GN, tile_m, tile_n);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = 0; k < nk; k += 2) {  // nk is even (K % 64 == 0): the register sets alternate with compile-time indices
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    // ---- epilogue
    if constexpr (EPI == 1) {
      const int mw = tile_m << 7, nw = (tile_n << 8) + wave * 64;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 bv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bv[gq] = *(const float4*)(a.bias + nw + 32 * j + 8 * gq + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          half_t* rowptr = a.out16 + (size_t)(mw + i * 32 + l31) * a.N + nw + 32 * j;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            half4_t h;
            h[0] = (half_t)(acc[i][j][4 * gq + 0] + bv[gq].x); h[1] = (half_t)(acc[i][j][4 * gq + 1] + bv[gq].y);
            h[2] = (half_t)(acc[i][j][4 * gq + 2] + bv[gq].z); h[3] = (half_t)(acc[i][j][4 * gq + 3] + bv[gq].w);
            *(half4_t*)(rowptr + 8 * gq + 4 * hi) = h;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) keep_live(acc[i][j]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------------------------------------
// gemm_pp3 (prototype): ONE wave per SIMD with a 128 x 128 wave tile — 256 accumulator registers (the unified 512-register
// file: hipcc puts the MFMA C/D operands in AGPRs), 4 waves = 2 (M) x 2 (N) on a 256 x 256 tile, K tile 64, two 64 KB stages
// (as gemm_pp).  32 fragment reads per 64 MFMAs instead of 24 per 32: a third fewer LDS bytes per FLOP; the fragments of K
// sub-step kk + 1 are read while the 16 MFMAs of sub-step kk run (the wave has nobody to hand the pipe to).
// Measured variants of this pipeline (main loop alone, FFN-1 / QKV / FFN-2 / output-projection shapes):
//   this form (vmcnt(0) + barrier at the K-tile boundary, one K tile in flight)                         1.00 - 1.08 PF
//   arrival barrier moved before the last sub-step + first fragments of the next K tile read under it    0.95 - 1.02 PF
//   four-deep ring of half-K sub-stages with 64-byte rows (prefetch distance 3, barrier mid sub-step)    0.83 - 0.89 PF
//     (half-line DMA rows double the vector-memory requests; gemm_pp2 above shares that layout)
// Timing ablations of this form (EPI 2 / 3): without the operand DMA the same loop runs at 1.31 - 1.44 PF; with the DMA issued but
// never waited for it is back at the shipped 1.00 - 1.13 PF — neither the wait nor the latency costs the 25 %, the LDS-DMA WRITES
// do (64 KB per K tile at the LDS's 64 - 85 B/clk store rate = ~40 % of the LDS's cycles, shared with the fragment reads).
// i.e. the K-tile boundary bubble is not what holds it back; with two 64 KB stages only ONE K tile (64 KB per CU) can be in
// flight for the ~1.1 us of a step, which is exactly the demand at 1.2 PF (58 GB/s per CU) with no slack — gemm_pp keeps the
// same 64 KB in flight but in 16 KB half-tiles issued every interval.  Next: stage units of 16 KB (A / W half-tiles, as
// gemm_pp) in a 10-unit ring (2.5 K tiles) so that 1.5 K tiles are in flight.
#define PP3_STAGE 65536
#define PP3_LDS (2 * PP3_STAGE)
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_pp3_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int K = a.K, nk = K >> 6;
  const int tm_count = a.M >> 8, tn_count = a.N >> 8;
  const int ntiles = tm_count * tn_count;
  const int G = gridDim.x;
  const int bslot = xcd_remap(blockIdx.x, G);
  if (bslot >= ntiles) return;
  // staging: a stage = 64 pieces of 1 KiB (8 rows x 128 B): 0..31 A rows 8 p.., 32..63 W rows; wave w issues pieces w + 4 i.
  // lane L: row L >> 3, slot L & 7 -> source chunk slot ^ ((row_in_tile >> 1) & 7), row_in_tile = 8 p + (L >> 3)
  uint32_t lane_src[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) lane_src[x] = (uint32_t)((lane >> 3) * K * 2 + (((lane & 7) ^ (((lane >> 4) + 4 * x) & 7)) << 4));
  int i_it = 0, i_k = 0;
  const char *iA = nullptr, *iW = nullptr;
  auto set_issue_tile = [&](int it) {
    const int L = it * G + bslot;
    if (L < ntiles) {
      int tm, tn;
      raster(L, tm_count, tn_count, a.GN, tm, tn);
      iA = (const char*)a.A + (size_t)tm * 256 * K * 2;
      iW = (const char*)a.W + (size_t)tn * 256 * K * 2;
    }
  };
  set_issue_tile(0);
  auto issue_stage = [&](int stg) {
    char* dst = smem + stg * PP3_STAGE;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = wave + 4 * i;  // piece parity = wave parity
      const char* src = (p < 32 ? iA + (size_t)(8 * p) * K * 2 : iW + (size_t)(8 * (p - 32)) * K * 2) + i_k * 128;
      glds16((const half_t*)(src + ((wave & 1) ? lane_src[1] : lane_src[0])), dst + p * 1024);
    }
    if (++i_k == nk) { i_k = 0; set_issue_tile(++i_it); }
  };
  const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
  const uint32_t rowA = (uint32_t)((wr * 128 + l31) * 128), rowW = 32768u + (uint32_t)((wc * 128 + l31) * 128);
  int my_tiles = 0;
  for (int L = bslot; L < ntiles; L += G) ++my_tiles;
  const int total = my_tiles * nk;
  issue_stage(0);
  floatx16 acc[4][4];
  int g = 0;
  for (int it = 0; it < my_tiles; ++it) {
    const int L = it * G + bslot;
    int tile_m, tile_n;
    raster(L, tm_count, tn_count, a.GN, tile_m, tile_n);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = 0; k < nk; ++k, ++g) {
      if (EPI != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K tile g has landed for this wave (the only thing in flight)
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // EPI 3 (timing ablation, wrong results): the DMA is issued but a whole extra K tile may stay in flight
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave has left K tile g - 1 (the slot of g + 1)
      __builtin_amdgcn_sched_barrier(0);
      if (EPI != 2 && g + 1 < total) issue_stage((g + 1) & 1);  // EPI 2 (timing ablation, wrong results): no operand DMA after the first tile
      const char* sb = smem + (g & 1) * PP3_STAGE;
      // Fragment reads as inline asm with their waits placed by hand: hipcc's own schedule reads one fragment, waits for it
      // and issues four MFMAs, sixteen times per K tile — with one wave per SIMD every one of those waits is exposed.  Here the
      // eight reads of sub-step kk + 1 are issued, then `lgkmcnt(8)` (LDS returns in order: sub-step kk's eight are in) and
      // only then the sixteen MFMAs of sub-step kk run, under which the newer reads complete.
      half8_t Af[2][4], Wf[2][4];
      const uint32_t sbo = (uint32_t)((g & 1) * PP3_STAGE);
#define PP3_READ(KK, SET, TAIL)                                                                                          \
  {                                                                                                                      \
    const uint32_t c = (((uint32_t)(2 * (KK) + hi)) ^ swz) << 4;                                                         \
    const uint32_t aa = sbo + rowA + c, ww = sbo + rowW + c;                                                             \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\t"          \
                 "ds_read_b128 %3, %8 offset:12288\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:4096\n\t"         \
                 "ds_read_b128 %6, %9 offset:8192\n\tds_read_b128 %7, %9 offset:12288\n\t" TAIL                          \
                 : "=&v"(Af[SET][0]), "=&v"(Af[SET][1]), "=&v"(Af[SET][2]), "=&v"(Af[SET][3]), "=&v"(Wf[SET][0]),         \
                   "=&v"(Wf[SET][1]), "=&v"(Wf[SET][2]), "=&v"(Wf[SET][3])                                                \
                 : "v"(aa), "v"(ww)                                                                                      \
                 : "memory");                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
#define PP3_MMA(SET)                                                                                                     \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                            \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[SET][j], Af[SET][i], acc[i][j], 0, 0, 0);                     \
  __builtin_amdgcn_sched_barrier(0);
      PP3_READ(0, 0, "")
      PP3_READ(1, 1, "s_waitcnt lgkmcnt(8)")
      PP3_MMA(0)
      PP3_READ(2, 0, "s_waitcnt lgkmcnt(8)")
      PP3_MMA(1)
      PP3_READ(3, 1, "s_waitcnt lgkmcnt(8)")
      PP3_MMA(0)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      PP3_MMA(1)
#undef PP3_READ
#undef PP3_MMA
    }
    if constexpr (EPI == 1) {
      const int mw = (tile_m << 8) + wr * 128, nw = (tile_n << 8) + wc * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 bv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bv[gq] = *(const float4*)(a.bias + nw + 32 * j + 8 * gq + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          half_t* rowptr = a.out16 + (size_t)(mw + i * 32 + l31) * a.N + nw + 32 * j;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            half4_t h;
            h[0] = (half_t)(acc[i][j][4 * gq + 0] + bv[gq].x); h[1] = (half_t)(acc[i][j][4 * gq + 1] + bv[gq].y);
            h[2] = (half_t)(acc[i][j][4 * gq + 2] + bv[gq].z); h[3] = (half_t)(acc[i][j][4 * gq + 3] + bv[gq].w);
            *(half4_t*)(rowptr + 8 * gq + 4 * hi) = h;
          }
        }
      }
    } else {  // keep the accumulators live without pinning 256 of them into VGPRs: one conditional store that never happens
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 12345.678f) a.out16[tid] = (half_t)sum;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------------------------------------
// gemm_pp4 (prototype): gemm_pp3 with the W operand NOT staged through the LDS: every lane loads its B-operand fragments
// (row = output column 32 j + lane & 31, 16 bytes of K) straight from global memory / L2 into registers, one K tile ahead
// (64 VGPRs per K tile, two sets: the one-wave-per-SIMD form has the registers).  The LDS carries A only: 32 KB per K tile
// (half the LDS-DMA writes), 16 fragment reads per 64 MFMAs instead of 32.  The two waves that share a column half load the
// same W bytes (L2 / L1 traffic for W doubles).
// Measured (main loop alone): 0.93 - 1.02 PF with fragment-major W (0.61 - 0.64 PF with row-major W: 64 scattered 16-B pieces per
// load instruction) against gemm_pp3's 1.00 - 1.11 and gemm_pp's 1.22 - 1.28 PF on the same box: 64 KB of W per K tile through the
// vector-memory path into VGPRs (64 B/clk per CU) costs more than the LDS-DMA writes it saves.
#define PP4_STAGE 32768
#define PP4_LDS (2 * PP4_STAGE)
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_pp4_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int K = a.K, nk = K >> 6;
  const int tm_count = a.M >> 8, tn_count = a.N >> 8;
  const int ntiles = tm_count * tn_count;
  const int G = gridDim.x;
  const int bslot = xcd_remap(blockIdx.x, G);
  if (bslot >= ntiles) return;
  uint32_t lane_src[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) lane_src[x] = (uint32_t)((lane >> 3) * K * 2 + (((lane & 7) ^ (((lane >> 4) + 4 * x) & 7)) << 4));
  int i_it = 0, i_k = 0;
  const char *iA = nullptr, *iW = nullptr;
  auto set_issue_tile = [&](int it) {
    const int L = it * G + bslot;
    if (L < ntiles) {
      int tm, tn;
      raster(L, tm_count, tn_count, a.GN, tm, tn);
      iA = (const char*)a.A + (size_t)tm * 256 * K * 2;
      // fragment-major W (a.W2, prepared once like the folded weights): block (32 columns nb, 16 k kb) = 64 lanes x 16 B contiguous,
      // so a wave-load is one contiguous KiB (row-major W: 64 scattered 16-B pieces per instruction, TA-bound at 0.64 PF)
      iW = (const char*)a.W2 + ((size_t)(tn * 8 + wc * 4) * (K >> 4)) * 1024 + lane * 16;
    }
  };
  set_issue_tile(0);
  half8_t Wg[2][4][4];  // [set][kk][j]
  auto issue_stage = [&](int stg, auto setc) {
    constexpr int set = decltype(setc)::value;
    char* dst = smem + stg * PP4_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = wave + 4 * i;  // A pieces 0..31 (8 rows x 128 B); piece parity = wave parity
      const char* src = iA + (size_t)(8 * p) * K * 2 + i_k * 128;
      glds16((const half_t*)(src + ((wave & 1) ? lane_src[1] : lane_src[0])), dst + p * 1024);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) Wg[set][kk][j] = *(const half8_t*)(iW + ((size_t)j * (K >> 4) + i_k * 4 + kk) * 1024);
    if (++i_k == nk) { i_k = 0; set_issue_tile(++i_it); }
  };
  const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
  const uint32_t rowA = (uint32_t)((wr * 128 + l31) * 128);
  int my_tiles = 0;
  for (int L = bslot; L < ntiles; L += G) ++my_tiles;
  const int total = my_tiles * nk;
  issue_stage(0, std::integral_constant<int, 0>{});
  floatx16 acc[4][4];
  int g = 0;
  half8_t Af[2][4];
#define PP4_READ(KK, SET, TAIL)                                                                                          \
  {                                                                                                                      \
    const uint32_t aa = sbo + rowA + ((((uint32_t)(2 * (KK) + hi)) ^ swz) << 4);                                         \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"          \
                 "ds_read_b128 %3, %4 offset:12288\n\t" TAIL                                                              \
                 : "=&v"(Af[SET][0]), "=&v"(Af[SET][1]), "=&v"(Af[SET][2]), "=&v"(Af[SET][3])                             \
                 : "v"(aa)                                                                                               \
                 : "memory");                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
#define PP4_MMA(SET, WS, KK)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                            \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wg[WS][KK][j], Af[SET][i], acc[i][j], 0, 0, 0);                  \
  __builtin_amdgcn_sched_barrier(0);
  auto step = [&](auto wsc) {
    constexpr int ws = decltype(wsc)::value;  // W register set of K tile g
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K tile g: A has landed in the LDS, W in this lane's registers
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < total) issue_stage((g + 1) & 1, std::integral_constant<int, ws ^ 1>{});
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t sbo = (uint32_t)((g & 1) * PP4_STAGE);
    PP4_READ(0, 0, "")
    PP4_READ(1, 1, "s_waitcnt lgkmcnt(4)")
    PP4_MMA(0, ws, 0)
    PP4_READ(2, 0, "s_waitcnt lgkmcnt(4)")
    PP4_MMA(1, ws, 1)
    PP4_READ(3, 1, "s_waitcnt lgkmcnt(4)")
    PP4_MMA(0, ws, 2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    PP4_MMA(1, ws, 3)
    ++g;
  };
  for (int it = 0; it < my_tiles; ++it) {
    const int L = it * G + bslot;
    int tile_m, tile_n;
    raster(L, tm_count, tn_count, a.GN, tile_m, tile_n);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = 0; k < nk; k += 2) {  // nk is even at the probe's shapes: the W register sets alternate with compile-time indices
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    if constexpr (EPI == 1) {
      const int mw = (tile_m << 8) + wr * 128, nw = (tile_n << 8) + wc * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 bv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bv[gq] = *(const float4*)(a.bias + nw + 32 * j + 8 * gq + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          half_t* rowptr = a.out16 + (size_t)(mw + i * 32 + l31) * a.N + nw + 32 * j;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            half4_t h;
            h[0] = (half_t)(acc[i][j][4 * gq + 0] + bv[gq].x); h[1] = (half_t)(acc[i][j][4 * gq + 1] + bv[gq].y);
            h[2] = (half_t)(acc[i][j][4 * gq + 2] + bv[gq].z); h[3] = (half_t)(acc[i][j][4 * gq + 3] + bv[gq].w);
            *(half4_t*)(rowptr + 8 * gq + 4 * hi) = h;
          }
        }
      }
    } else {  // keep the accumulators live without pinning 256 of them into VGPRs: one conditional store that never happens
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 12345.678f) a.out16[tid] = (half_t)sum;
    }
  }
#undef PP4_READ
#undef PP4_MMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void fill_h(half_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (half_t)(((hash32((uint32_t)i * 2654435761u + seed) >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale);
}
__global__ void fill_f(float* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = ((hash32((uint32_t)i * 2654435761u + seed) >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale;
}
// W [N][K] row-major -> fragment-major: block (nb = n / 32, kb = k / 16) holds lane (hi = (k % 16) / 8, l = n % 32) x 8 halfs
__global__ void to_fragment_major(const half_t* W, half_t* Wf, int N, int K) {
  const size_t n8 = (size_t)N * K / 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const size_t blk = i >> 6;
    const int kb = (int)(blk % (K >> 4)), nb = (int)(blk / (K >> 4));
    const int n = nb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
    *(half8_t*)(Wf + i * 8) = *(const half8_t*)(W + (size_t)n * K + k);
  }
}
__global__ void cmp_h(const half_t* x, const half_t* y, size_t n, unsigned* maxbits, unsigned long long* bad) {
  float mx = 0.f; unsigned long long nb = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = fabsf((float)x[i] - (float)y[i]);
    if (!(d <= 1e-3f * fmaxf(1.f, fabsf((float)x[i])))) nb++;
    mx = fmaxf(mx, d == d ? d : 1e30f);
  }
  atomicMax(maxbits, __float_as_uint(mx));
  if (nb) atomicAdd(bad, nb);
}

static int choose_gn(int tn, int gn_max) { int g = 1; for (int d = 1; d <= gn_max && d <= tn; ++d) if (tn % d == 0) g = d; return g; }

int main() {
  const int M = 65536;
  half_t *A, *W, *Wfm, *o1, *o2; float* bias; unsigned* mb; unsigned long long* bad;
  CK(hipMalloc(&A, (size_t)M * 3072 * 2)); CK(hipMalloc(&W, (size_t)3072 * 3072 * 2)); CK(hipMalloc(&Wfm, (size_t)3072 * 3072 * 2)); CK(hipMalloc(&o1, (size_t)M * 3072 * 2)); CK(hipMalloc(&o2, (size_t)M * 3072 * 2));
  CK(hipMalloc(&bias, 3072 * 4)); CK(hipMalloc(&mb, 4)); CK(hipMalloc(&bad, 8));
  hipLaunchKernelGGL(fill_h, dim3(4096), dim3(256), 0, 0, A, (size_t)M * 3072, 1u, 1.0f);
  hipLaunchKernelGGL(fill_h, dim3(4096), dim3(256), 0, 0, W, (size_t)3072 * 3072, 2u, 0.05f);
  hipLaunchKernelGGL(fill_f, dim3(16), dim3(256), 0, 0, bias, (size_t)3072, 3u, 0.5f);
  CK(hipDeviceSynchronize());
  int ncu = 256;
  { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
  auto k_ref = gemm_pp_kernel<PP_F16, 4, 0, 1, 1>;
  auto k_ref_noepi = gemm_pp_kernel<PP_F16, 4, PP_ABL_NOEPI, 1, 1>;
  CK(hipFuncSetAttribute((const void*)k_ref, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)k_ref_noepi, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm_pp2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP2_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP2_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP3_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP3_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, PP3_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PP3_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP4_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_pp4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP4_LDS));
  { int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_pp2_kernel<1>, 256, PP2_LDS)); printf("gemm_pp2: %d workgroups per CU (occupancy query)\n", nb); }
  struct Shape { const char* name; int N, K; };
  const Shape shapes[] = {{"FFN-1 (N 3072, K 768)", 3072, 768}, {"QKV (N 2304, K 768)", 2304, 768}, {"FFN-2 (N 768, K 3072)", 768, 3072}, {"out-proj (N 768, K 768)", 768, 768}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& s : shapes) {
    GemmArgs a{};
    a.A = A; a.W = W; a.bias = bias; a.M = M; a.Mreal = M; a.N = s.N; a.K = s.K;
    GemmArgs r = a; r.out16 = o1; r.GN = choose_gn(s.N / 256, 4);
    GemmArgs n = a; n.out16 = o2; n.GN = choose_gn(s.N / 256, 4);
    hipLaunchKernelGGL(to_fragment_major, dim3(2048), dim3(256), 0, 0, W, Wfm, s.N, s.K);
    n.W2 = Wfm;
    const int tiles_ref = (M / 256) * (s.N / 256), tiles_new = (M / 128) * (s.N / 256);
    auto run_ref = [&](bool epi) { hipLaunchKernelGGL(epi ? k_ref : k_ref_noepi, dim3(std::min(tiles_ref, ncu)), dim3(512), PP_LDS_BYTES, 0, r); };
    auto run_new = [&](bool epi) {
      if (epi) hipLaunchKernelGGL(gemm_pp2_kernel<1>, dim3(std::min(tiles_new, 2 * ncu)), dim3(256), PP2_LDS, 0, n);
      else hipLaunchKernelGGL(gemm_pp2_kernel<0>, dim3(std::min(tiles_new, 2 * ncu)), dim3(256), PP2_LDS, 0, n);
    };
    auto run_pp3 = [&](bool epi) {
      if (epi) hipLaunchKernelGGL(gemm_pp3_kernel<1>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP3_LDS, 0, n);
      else hipLaunchKernelGGL(gemm_pp3_kernel<0>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP3_LDS, 0, n);
    };
    {  // gemm_pp3 against the reference first
      CK(hipMemset(o1, 0, (size_t)M * s.N * 2)); CK(hipMemset(o2, 0xff, (size_t)M * s.N * 2));
      run_ref(true); run_pp3(true);
      CK(hipDeviceSynchronize());
      CK(hipMemset(mb, 0, 4)); CK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(cmp_h, dim3(2048), dim3(256), 0, 0, o1, o2, (size_t)M * s.N, mb, bad);
      unsigned mbh; unsigned long long badh; CK(hipMemcpy(&mbh, mb, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&badh, bad, 8, hipMemcpyDeviceToHost));
      float mx; memcpy(&mx, &mbh, 4);
      printf("%-26s gemm_pp3 check: max|diff| %.3e, bad %llu %s\n", s.name, mx, badh, badh ? "FAIL" : "OK");
      CK(hipMemset(o2, 0xff, (size_t)M * s.N * 2));
      hipLaunchKernelGGL(gemm_pp4_kernel<1>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP4_LDS, 0, n);
      CK(hipDeviceSynchronize());
      CK(hipMemset(mb, 0, 4)); CK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(cmp_h, dim3(2048), dim3(256), 0, 0, o1, o2, (size_t)M * s.N, mb, bad);
      CK(hipMemcpy(&mbh, mb, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&badh, bad, 8, hipMemcpyDeviceToHost));
      memcpy(&mx, &mbh, 4);
      printf("%-26s gemm_pp4 check: max|diff| %.3e, bad %llu %s\n", s.name, mx, badh, badh ? "FAIL" : "OK");
    }
    CK(hipMemset(o1, 0, (size_t)M * s.N * 2)); CK(hipMemset(o2, 0xff, (size_t)M * s.N * 2));
    run_ref(true); run_new(true);
    CK(hipDeviceSynchronize());
    CK(hipMemset(mb, 0, 4)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(cmp_h, dim3(2048), dim3(256), 0, 0, o1, o2, (size_t)M * s.N, mb, bad);
    unsigned mbh; unsigned long long badh; CK(hipMemcpy(&mbh, mb, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&badh, bad, 8, hipMemcpyDeviceToHost));
    float mx; memcpy(&mx, &mbh, 4);
    printf("%-26s check: max|diff| %.3e, bad %llu %s\n", s.name, mx, badh, badh ? "FAIL" : "OK");
    const double fl = 2.0 * M * s.N * s.K;
    for (int which = 0; which < 10; ++which) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) { if (which == 0) run_ref(true); else if (which == 1) run_ref(false); else if (which == 2) run_new(true); else if (which == 3) run_new(false); else if (which == 4) run_pp3(true); else if (which == 5) run_pp3(false); else if (which == 6) hipLaunchKernelGGL(gemm_pp3_kernel<2>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP3_LDS, 0, n); else if (which == 7) hipLaunchKernelGGL(gemm_pp3_kernel<3>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP3_LDS, 0, n); else if (which == 8) hipLaunchKernelGGL(gemm_pp4_kernel<1>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP4_LDS, 0, n); else hipLaunchKernelGGL(gemm_pp4_kernel<0>, dim3(std::min(tiles_ref, ncu)), dim3(256), PP4_LDS, 0, n); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 5);
      }
      const char* nm[] = {"gemm_pp  256x256 1 wg/CU", "gemm_pp  no epilogue", "gemm_pp2 128x256 2 wg/CU", "gemm_pp2 no epilogue", "gemm_pp3 128x128 wave tile", "gemm_pp3 no epilogue", "gemm_pp3 no epi, no DMA", "gemm_pp3 no epi, late wait", "gemm_pp4 W from L2 to regs", "gemm_pp4 no epilogue"};
      printf("   %-28s %8.1f us  %7.1f TF\n", nm[which], best * 1e3, fl / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
