// NOT part of the library (kept as a record, see the result at the end of this comment).
// Attention-output projection (K4a) as TWO workgroups per CU: `gemm_pp2_resln3_kernel`.
//
// The residual GEMMs read and write the raw stream through the GEMM (PP_RESLN3, gemm_pp.h): per 256 x 256 tile 256 KB of loads
// in the accumulator init and 256 KB of stores in the epilogue, at the ~25 GB/s a CU can pull (256 CUs x 25 GB/s = the HBM rate:
// all workgroups of gemm_pp run their phases together) — ~18 us per tile with the matrix pipes idle.  With K = 768 the main loop
// of a tile is only 24 us, so the output projection spends 43 % of its time in those phases (149 us per launch against 65 us of
// main loops and 83 us of HBM time).  Nothing inside ONE workgroup can overlap them; a second, independent
// workgroup on the same CU can: while one is in its I/O phases the other owns the matrix pipes.
//
//   * 256 threads = 4 waves, tile 128 (M) x 256 (N), wave w -> columns 64 w .. + 63 of all 128 rows: the same 128 x 64 wave tile,
//     fragment geometry, K order (ascending 16-wide MFMA steps) and epilogue arithmetic as gemm_pp -> bit-identical results;
//   * K step 32: a stage = A [128 rows][64 B] + W [256 rows][64 B] = 24 KB, three stages (72 KB) + 4 x 2 KB transposition
//     scratch = 80 KB per workgroup, two per CU; LDS-DMA staging, 16-B chunk c of row r at slot c ^ ((r >> 2) & 3) (applied on
//     the per-lane source address and on the fragment read: conflict-free for ds_read_b128's lane groups); one barrier per K step;
//   * its main loop alone is SLOWER than gemm_pp's (0.9 vs 1.2 PF, tools/gemm_pp2_probe.hip: the two workgroups do not share
//     their A tiles) — it pays only where the I/O phases dominate, i.e. for this GEMM, not for FFN-2 (K = 3072);
//   * bias / gamma / beta of the tile's 256 columns (3 KB) sit in the ring slot that is free between two tiles (the stage the
//     last K step consumed: it is refilled by the next tile's first step, after the accumulator init that reads the image).
//
// PP_RESLN3 semantics (gemm_pp.h): residual tile = hi + lo fp16 planes (a.out16 / a.out16b, updated in place), normalised with
// the rows' vstats (a.lnstats) and a.lng / a.lnb while initialising the accumulators; epilogue writes the new raw stream as
// hi, lo planes and the rows' vstats of this tile's 256 columns (a.lnpart, slot tile_n).  N == 768, K % 64 == 0, M % 128 == 0.
//
// RESULT (round 2, wired into engine.hip behind MEMVUL_PP2 for one same-box A/B, results bit-compatible with the parity suite):
// the output projection went from 146 - 149 us (gemm_pp, one workgroup per CU) to 173 us, 20.4 k -> 19.9 k issue reports/s.  The
// phases do not overlap the way the byte count suggests: this form's main loop is LDS-bound (fragment reads + LDS-DMA writes,
// tools/gemm_pp2_probe.hip) and the other workgroup's accumulator init / epilogue go through the LDS as well (the transposition
// scratch), so the two compete for the same unit instead of using different ones.  Removed from the library again.
#pragma once
#include "legacy/gemm_pp.h"

#define PP2_STAGE 24576
#define PP2_NSTG 3
#define PP2_LDS_SCR (PP2_NSTG * PP2_STAGE)   // 4 waves x 2 KiB wave-private transposition scratch
#define PP2_LDS_BYTES (PP2_LDS_SCR + 4 * 2048)  // 81,920: two workgroups per CU

// One float4 of each of the three column images (bias, gamma at +1 KiB, beta at +2 KiB).
__device__ __forceinline__ void pp2_read_bgb1(uint32_t addr, float4& bi, float4& ga, float4& be) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(bi), "=&v"(ga), "=&v"(be)
               : "v"(addr)
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

__global__ __launch_bounds__(256, 2) void gemm_pp2_resln3_kernel(GemmArgs a) {
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
  // fragment read offsets inside a stage (A at +0, W at +8192)
  uint32_t offA[2], offW[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint32_t c = (uint32_t)(((2 * kk + hi) ^ ((l31 >> 2) & 3)) << 4);
    offA[kk] = (uint32_t)(l31 * 64) + c;
    offW[kk] = 8192u + (uint32_t)((64 * wave + l31) * 64) + c;
  }
  int my_tiles = 0;
  for (int L = bslot; L < ntiles; L += G) ++my_tiles;
  const int total = my_tiles * nk;

  // ---- wave-private scratch addresses (as gemm_pp's COAL path)
  const uint32_t scr = (uint32_t)(PP2_LDS_SCR + wave * 2048);
  const uint32_t sf = (uint32_t)((l31 >> 2) & 3);
  const uint32_t scr_c = scr + (lane >> 2) * 64 + ((((uint32_t)lane & 3) ^ (((uint32_t)lane >> 4) & 3)) << 4);
  const uint32_t wbase = scr + l31 * 64 + hi * 8 + (sf << 4);
  const int crow = lane >> 2, cchunk = lane & 3;  // coalesced layout: row (+16 for the second access), 16-B chunk

  floatx16 acc[4][2];
  // the two fp16 planes of fragment pair i of the residual tile at (mw0, nw0) by full-line loads (16 rows x 64 B per instruction),
  // parked in the accumulator registers they will be transposed into: [i][j] registers 0-3 / 4-7 = hi rows crow / crow + 16,
  // 8-11 / 12-15 = lo
  auto park_residual = [&](int i, int mw0, int nw0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const half_t* src = (pl ? a.out16b : a.out16) + (size_t)(mw0 + i * 32 + x * 16 + crow) * MV_HIDDEN + nw0 + j * 32 + 8 * cchunk;
          const float4 t = *(const float4*)src;
          acc[i][j][8 * pl + 4 * x + 0] = t.x; acc[i][j][8 * pl + 4 * x + 1] = t.y;
          acc[i][j][8 * pl + 4 * x + 2] = t.z; acc[i][j][8 * pl + 4 * x + 3] = t.w;
        }
  };
  // bias / gamma / beta of tile column block tn -> the 3 KB image at the start of ring slot `slot` (one float of each per thread;
  // inline asm: a compiler-visible LDS store would first wait for the LDS-DMA in flight)
  auto write_bgb = [&](int slot, int tn) {
    const uint32_t addr = (uint32_t)(slot * PP2_STAGE + tid * 4);
    const int col = tn * 256 + tid;
    const float b = a.bias ? a.bias[col] : 0.f, g = a.lng[col], be = a.lnb[col];
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:1024\n\tds_write_b32 %0, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                 ::"v"(addr), "v"(b), "v"(g), "v"(be)
                 : "memory");
  };

  // ---- prologue: the first tile's column image goes into slot 2 (stage 2 is issued by the first K step, after the init)
  {
    int tm, tn;
    raster(bslot, tm_count, tn_count, a.GN, tm, tn);
    write_bgb(2 % PP2_NSTG, tn);
  }
  issue_stage(0);
  if (total > 1) issue_stage(1);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();  // the column image is complete
  __builtin_amdgcn_sched_barrier(0);

  int g = 0;
  for (int it = 0; it < my_tiles; ++it) {
    const int L = it * G + bslot;
    int tile_m, tile_n;
    raster(L, tm_count, tn_count, a.GN, tile_m, tile_n);
    const bool has_next = it + 1 < my_tiles;
    int next_m = 0, next_n = 0;
    if (has_next) raster(L + G, tm_count, tn_count, a.GN, next_m, next_n);
    const int mw = tile_m << 7;
    const int nw = (tile_n << 8) + wave * 64;

    // ---- accumulator init: LN(residual) + bias, the residual lines parked in the accumulator registers (requested during the
    // previous tile's epilogue; here for the workgroup's first tile)
    {
      float2 lnp[4][3], lnst[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2* pp = (const float2*)(a.lnstats + 6 * (size_t)(mw + i * 32 + l31));
        lnp[i][0] = pp[0]; lnp[i][1] = pp[1]; lnp[i][2] = pp[2];
      }
      if (it == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) park_residual(i, mw, nw);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) lnst[i] = ln_from_partials(lnp[i][0], lnp[i][1], lnp[i][2], a.ln_eps);
      const uint32_t baddr = (uint32_t)(((g + 2) % PP2_NSTG) * PP2_STAGE + (wave * 64 + 4 * hi) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma clang fp contract(off)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          u32x4 p[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) p[q][e] = f2u(acc[i][j][4 * q + e]);
          u32x2 oh[4], ol[4];
          scr_f16_rev2(scr_c, p[0], p[1], p[2], p[3], wbase, wbase ^ 16u, wbase ^ 32u, wbase ^ 48u, oh, ol);
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            float4 bi, ga, be;
            pp2_read_bgb1(baddr + j * 128 + gq * 32, bi, ga, be);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t wh = oh[gq][e >> 1], wl = ol[gq][e >> 1];  // scalar copies before the bit casts (see f2u)
              const half2_t h2 = __builtin_bit_cast(half2_t, wh);
              const half2_t l2 = __builtin_bit_cast(half2_t, wl);
              const float r = (float)h2[e & 1] + (float)l2[e & 1];
              const float t = (r - lnst[i].x) * lnst[i].y;
              acc[i][j][4 * gq + e] = __builtin_fmaf(t, ((const float*)&ga)[e], ((const float*)&be)[e]) + ((const float*)&bi)[e];
            }
          }
        }
      }
    }

    // ---- main loop
    for (int k = 0; k < nk; ++k, ++g) {
      // stage g has landed for this wave once at most the 6 pieces of stage g + 1 are outstanding
      if (g + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave has left stage g - 1 (= the slot of stage g + 2) and, at
      __builtin_amdgcn_sched_barrier(0);  // a tile's first step, its accumulator init (the column image lives in that slot)
      if (g + 2 < total) issue_stage((g + 2) % PP2_NSTG);
      const char* sb = smem + (g % PP2_NSTG) * PP2_STAGE;
      half8_t Af[4][2], Wf[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 2; ++j) Wf[j][kk] = *(const half8_t*)(sb + offW[kk] + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) Af[i][kk] = *(const half8_t*)(sb + offA[kk] + i * 2048);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[j][kk], Af[i][kk], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue.  vstats of the new raw rows over this tile's 256 columns: each wave reduces its 64 columns (lane = token row,
    // the two half-waves hold disjoint column sets) and parks the 128 pairs in its scratch; after a workgroup barrier — which also
    // means every wave is past its last fragment reads: the slot of the last K step is free — wave 0 adds the four shares in wave
    // order and writes slot tile_n of the rows' three pairs, and all threads write the NEXT tile's column image into the free
    // slot.  A second barrier keeps the scratch intact until it has been read.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          s1 += v;
          s2 = __builtin_fmaf(v, v, s2);
        }
      const auto t1 = __builtin_amdgcn_permlane32_swap(f2u(s1), f2u(s1), false, false);
      const auto t2 = __builtin_amdgcn_permlane32_swap(f2u(s2), f2u(s2), false, false);
      float2 st;
      st.x = u2f(t1[0]) + u2f(t1[1]);
      st.y = u2f(t2[0]) + u2f(t2[1]);
      const uint32_t waddr = scr + (uint32_t)(i * 32 + l31) * 8;  // both half-waves write the same pair
      asm volatile("ds_write_b64 %0, %1" ::"v"(waddr), "v"(st) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) write_bgb((g + 2) % PP2_NSTG, next_n);  // g = the next tile's first step: its free slot is (g + 2) % 3 = (g - 1) % 3
    if (wave == 0) {  // rows i * 32 + l31: the lower half-wave takes i = 0, 1, the upper one i = 2, 3
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int row = (2 * hi + ii) * 32 + l31;
        const uint32_t raddr = (uint32_t)PP2_LDS_SCR + (uint32_t)row * 8;
        float2 q0, q1, q2, q3;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\tds_read_b64 %2, %4 offset:4096\n\t"
                     "ds_read_b64 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                     : "v"(raddr)
                     : "memory");
        float2 st;
        st.x = ((q0.x + q1.x) + q2.x) + q3.x;
        st.y = ((q0.y + q1.y) + q2.y) + q3.y;
        *(float2*)(a.lnpart + ((size_t)(mw + row) * 3 + tile_n) * 2) = st;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
      half_t* obase = a.out16 + (size_t)(mw + crow) * a.N + nw + 8 * cchunk;
      const size_t rstride = a.N, istride = (size_t)32 * a.N, jstride = 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        half_t* ob = obase + i * istride;
        u32x2 d[2][4];
        u32x4 o[4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            d[j][gq][0] = pack_h2(acc[i][j][4 * gq + 0], acc[i][j][4 * gq + 1]);
            d[j][gq][1] = pack_h2(acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]);
          }
        scr_f16x2(wbase, wbase ^ 16u, wbase ^ 32u, wbase ^ 48u, d[0], d[1], scr_c, o);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          half_t* op = ob + j * jstride;
          *(u32x4*)op = o[2 * j];
          *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
        }
        // second plane: lo = fp16(r - hi), same addresses in the lo buffer
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const float v0 = acc[i][j][4 * gq + 0], v1 = acc[i][j][4 * gq + 1], v2 = acc[i][j][4 * gq + 2], v3 = acc[i][j][4 * gq + 3];
            d[j][gq][0] = pack_h2(v0 - (float)(half_t)v0, v1 - (float)(half_t)v1);
            d[j][gq][1] = pack_h2(v2 - (float)(half_t)v2, v3 - (float)(half_t)v3);
          }
        scr_f16x2(wbase, wbase ^ 16u, wbase ^ 32u, wbase ^ 48u, d[0], d[1], scr_c, o);
        half_t* ol = ob + (a.out16b - a.out16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          half_t* op = ol + j * jstride;
          *(u32x4*)op = o[2 * j];
          *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
        }
        // fragment pair i is out: request pair i of the NEXT tile's residual into its registers
        if (has_next) park_residual(i, next_m << 7, (next_n << 8) + wave * 64);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}
