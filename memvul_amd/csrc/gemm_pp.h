// Persistent ping-pong MFMA GEMM for the BERT projections at bench scale:
//   C[M,N] = A[M,K] (fp16) x W[N,K]^T (fp16, torch Linear.weight layout) + bias, fp32 accumulate,
// with the reference's elementwise work fused (HF BertSelfAttention / BertSelfOutput / BertIntermediate /
// BertOutput as invoked from custom_PTM_embedder.py:228).
//
// Structure (cdna_hip_programming.md §5 "256^2 8-phase template", T1-T5; MI355X_MICROARCH.md "Two waves per SIMD"):
//   * 256x256x64 tile, 512 threads = 8 waves as 2(M) x 4(N), 128x64 of C per wave = 8 token blocks x 4 column blocks of 16 x 16
//     (128 accumulator VGPRs), **v_mfma_f32_16x16x32_f16**: these kernels are power-bound and this shape costs less
//     energy per FLOP than 32x32x16 (half the accumulator traffic): chip-wide 1.82-1.85 PF against 1.46-1.50
//     (tools/mfma_shape_probe.hip), +6 % on every launch; the 32x32x16 form of this file is in the history (git show 5bc1d9a^:memvul_amd/csrc/gemm_pp.h).
//     ONE workgroup per CU, grid = #CUs, each workgroup walks a strided list of output tiles (persistent): the K-tile stream
//     never drains between output tiles, so the next tile's operands are already in flight while the epilogue stores.
//   * ONE s_barrier per phase; the first M-half of the workgroup (waves 0-3) runs [MFMA(j), read fragments(j+1)] and
//     the second (waves 4-7: one wave of each half per SIMD) [read fragments(j), MFMA(j)] inside the same barrier
//     interval, so each SIMD's matrix pipe is handed from one wave to the other in the middle of the interval.
//   * A K-tile is 4 phases = the 4 quadrants (64 tokens x 32 columns: 16 MFMAs) of the wave's C block in the order (a0,b0) (a0,b1)
//     (a1,b1) (a1,b0): each phase reads at most ONE operand half-tile (8 or 4 ds_read_b128 per wave).  Operand lane (row = lane & 15,
//     kg = lane >> 4) holds the 8 halves k = 32 kk + 8 kg .. + 7 of its row = 16-byte chunk 4 kk + kg of the 128-byte K-tile row.
//   * Operand half-tiles (128 rows x 128 B = 16 KiB = 2 LDS-DMA instructions per wave) are the unit of staging: LDS
//     holds two K-tiles x {a0,a1,b0,b1} = 128 KiB; F = 4 half-tiles are kept in flight across every barrier (half-tile
//     H is issued in interval H-3-F, is landed for every wave at the barrier that ends interval H-3, and its LDS region
//     was last read in interval H-8 or H-9) with one counted `s_waitcnt vmcnt(2 F)` per interval; loads stay in
//     flight across barriers (raw s_barrier, never __syncthreads).
//   * LDS image of a half-tile is lane-linear per DMA instruction (1 KiB = 8 rows x 128 B); the bank swizzle
//     (16-B chunk c of row r at slot c ^ ((r >> 1) & 7)) is applied on the per-lane SOURCE address and on the
//     ds_read_b128 (rule 21).  The DMA uses the SGPR-base + 32-bit-VGPR-offset form.
//   * Orientation: C^T fragments (W rows as the MFMA A operand): acc[tb][cb] (4 registers) = token 16 tb + (lane & 15), columns
//     16 cb + 4 (lane >> 4) .. + 3; every 32 x 32 block (4 accumulators) goes through a wave-private LDS transposition so that
//     each global store / residual load instruction covers 16 rows x 64 contiguous bytes.
//   * Tile order: logical tile sequence = (column group of GN tiles) > tile_m > tile_n-in-group; per persistent
//     iteration the 256 concurrent tiles are consecutive in it and each XCD takes a contiguous run of 32 (the
//     group's W panels stay in that XCD's L2 while it sweeps the A row panels).
//
// Three kernel kinds (the encoder layer's four GEMMs; what rounds 1-2 also carried — fp32-stream epilogues, the two-barrier
// schedule, timing ablations — was retired (git history before round 5; A/B records in profiles/):
//   PP_QK   (RAW)  Q, K (head-major) and V^T in ONE launch: the V tiles go through the wave's LDS image transposed.
//   PP_GELU (RAW)  FFN-1 + exact-erf GELU.
//   PP_RESLN3      attention-output projection / FFN-2: + bias + LayerNorm(residual), in place on the raw stream.
// "Virtual LayerNorm": no LayerNorm kernel between the GEMMs.  By linearity
//   W LN(r) + b = rstd * (W'' r) + b',  W''[n][k] = W[n][k] gamma[k] - mean_k(W[n][.] gamma[.]),  b' = b + W beta
// (the row mean of r drops out against the row-centred weights), so a RAW consumer takes the raw stream rounded to fp16 as
// its A operand and the folded weights W'' (prepared once on the host), starts its accumulators from zero and applies
// fma(rstd_row, acc, b'_col) in the epilogue; its next tile's row statistics and bias' arrive by seven LDS-DMA pieces.  The
// producer (PP_RESLN3) normalises the residual tile it loads anyway while initialising the accumulators and writes the new raw
// stream as TWO fp16 planes  hi = fp16(r)  (the operand of the RAW consumers),  lo = fp16(r - hi)  (r ~= hi + lo to 2^-22)
// plus the rows' "vstats": per row and 256-column tile the (sum, sum of squares); every consumer turns the three pairs of a
// row into (mean, rstd) itself (common.h ln_from_partials).  (X8: no lo fp16 plane — the low part is the lo8 plane of the stream's fp8 planes, the special
// rows' the compact sp_lo_out: park_residual, gemm.h GemmArgs::out16b.)
//
// X8 = 1 (compute dtype MV_F16X8, "precise"): every GEMM adds a SECOND sweep (x8_terms = 1: its weight-side half only) on the fp8 matrix path into the same fp32
// accumulators:  A W ~= A_hi W_hi + 2^-s (A_lo8 W_hi8 + A_hi8 W_lo8),  A_hi = fp16(A), A_lo8 = e4m3((A - A_hi) 2^(11 + sa)),
// A_hi8 = e4m3(A_hi 2^sa), W likewise with its own shift sw, s = 11 + sa + sw — the first-order correction terms of the
// split-operand product, which only need ~4 significant bits, as ONE v_mfma_scale_f32_16x16x128_f8f6f4 sweep (OCP e4m3, uniform
// E8M0 scales = the exact power of two 2^-s) over a virtual K of 2 K: the fp8 operands are rows [lo8 (K bytes) | hi8 (K bytes)]
// for A and [hi8 | lo8] for W, so row pitch (2 K bytes), K-tile width (128 bytes) and K-tile count (K / 64) equal the fp16
// sweep's and the staging code is shared; an fp8 K-tile covers 128 products per row pair in the matrix-pipe time the fp16
// tile needs for 64.  Cost: 2x the main loop (a three-sweep fp16 split: 3x); error: operand rounding 2^-12 -> ~2^-15.5
// (oracle/precision_model.py "f16x8").  The producers (PP_GELU, PP_RESLN3; embedding, attention) write the [lo8 | hi8] planes.
// [CLS]-row form (round 5, the default; engine.hip cls_aside): only token 0 of a sequence reaches the pooler (model_memory.py:99) and every other row's
// A-operand rounding reaches it through attention, averaged over the keys — so every launch sweeps the weight-side term only (x8_terms = 1: K / 128 fp8
// K-tiles; the Q block of the QKV projection keeps both), the A-side term A_lo W_hi^T of the [CLS] rows alone arrives through GemmArgs::cls_corr (a skinny fp16
// GEMM over those B rows in front of the launch) and is added to those rows' accumulators at the start of the epilogue, and the producers write the hi8
// plane alone (GemmArgs::out8_hi_only; a 32-row block that holds a [CLS] row keeps its lo8 row).  Row tiles of sequences too short for it (GemmArgs::tile_both)
// run the both-terms form bit for bit.  +14 % issue reports/s at the both-terms form's trained-like logit error (profiles/r05_j*, r05_k*).
// PP_QK X8 with GemmArgs::vt_lo set (passes of padded length <= 128): second fp16 planes of Q, K and V^T for the two-plane attention (attention_v2.h VLO).
#pragma once
#include "common.h"
#include <type_traits>
#include "gemm.h"

enum { PP_QK = 1, PP_GELU = 3, PP_RESLN3 = 8 };
constexpr int PP_F = 4;  // half-tiles kept in flight across each barrier

#define PP_LDS_A 0           // [par][a][wr][64 rows][128 B]
#define PP_LDS_B 65536       // [par][b][wc][32 rows][128 B]
#define PP_LDS_BIAS 131072   // PP_RESLN3: bias | gamma | beta, 3 x 768 fp32
#define PP_LDS_SCR (PP_LDS_BIAS + MV_INTER * 4)  // 8 waves x 2 KiB: wave-private transposition scratch
#define PP_LDS_BYTES (PP_LDS_SCR + 8 * 2048)     // 159,744 of 163,840
// RAW kernels re-partition everything above the operand ring: [2][256] bias' of the tile | [2][256 rows][3][sum, sumsq] | scratch
#define PP_LDS_BIAS_T PP_LDS_BIAS                  // 2 x 1 KiB
#define PP_LDS_STATS (PP_LDS_BIAS + 2048)          // 2 x 6 KiB
#define PP_LDS_SCR_RAW (PP_LDS_STATS + 2 * 6144)   // 8 waves x 2 KiB
#define PP_LDS_RSTD (PP_LDS_SCR_RAW + 8 * 2048)    // 2 x 256 rstd of the tile's rows (computed once per workgroup, in the main loop)
#define PP_LDS_BYTES_RAW (PP_LDS_RSTD + 2048)      // 163,840 = all of the CU's LDS

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  const half2_t h = {(half_t)a, (half_t)b};
  return __builtin_bit_cast(uint32_t, h);
}

// An fp32 result that is about to be rounded to fp16 goes through this first: left alone, hipcc fuses `(half_t)fmaf(a, b, c)` into v_fma_mix{lo,hi}_f16 (ONE
// rounding) for some elements of an unrolled epilogue and keeps v_fma_f32 + v_cvt (TWO roundings) for others, so the same row gave different fp16 bits in
// token block tb and in tb + 4 — 1 ulp in ~2^-13 of the elements — and a row's bits depended on which 64-row group of its tile it sat in (passes of
// padded length 64 / 192; round 6, scripts/r06_perm_probe*.py).  Every store of these kernels rounds the fp32 value the source names.
__device__ __forceinline__ void pin_f32x4(float& a, float& b, float& c, float& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#endif
}

// (hipcc/ROCm 7.2: __builtin_bit_cast applied directly to an ext_vector ELEMENT expression reads element 0 —
// always go through a scalar copy)
__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __uint_as_float(x); }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));

// Wave-private transposition through a [32 rows][64 B] LDS image (16-B chunk c of row r at slot c ^ ((r >> 2) & 3)):
// the C/D layout gives a lane 8 contiguous bytes of a row; stored directly every wave-store touches 64 scattered pieces (~64 cycles
// of address processing per instruction, 3.6 us per 256^2 tile), through the image each store covers 16 rows x 64 contiguous
// bytes.  Inline asm keeps these LDS accesses out of hipcc's LDS-DMA alias bookkeeping (it would put `s_waitcnt vmcnt(0)` before
// them); LDS executes a wave's instructions in order, so the read-after-write needs no wait, only the read results do (guide
// §5.7 form i).  Two blocks (j = 0, 1) per statement: the second's writes may follow the first's reads without a wait in between.
__device__ __forceinline__ void scr_f16x2(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const u32x2 (&da)[4],
                                          const u32x2 (&db)[4], uint32_t r, u32x4 (&o)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_write_b64 %4, %8\n\tds_write_b64 %5, %9\n\tds_write_b64 %6, %10\n\tds_write_b64 %7, %11\n\t"
      "ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:1024\n\t"
      "ds_write_b64 %4, %12\n\tds_write_b64 %5, %13\n\tds_write_b64 %6, %14\n\tds_write_b64 %7, %15\n\t"
      "ds_read_b128 %2, %16\n\tds_read_b128 %3, %16 offset:1024\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(da[0]), "v"(da[1]), "v"(da[2]), "v"(da[3]), "v"(db[0]), "v"(db[1]),
        "v"(db[2]), "v"(db[3]), "v"(r)
      : "memory");
#endif
}
// Two fp8 planes (MV_F16X8: lo8, then hi8) of one 32-row x 64-column block through the wave's [32 rows][64 B] image, 16x16 C/D layout:
// the lane's dword (tbl, s) = token 16 tbl + m16, columns 16 s + 4 q4 .. + 3 goes to slot s of row 16 tbl + m16 at byte 4 q4;
// w0..w3 = the lane's (swizzled) slot addresses of row m16, da / db index 4 tbl + s.  The read side is the fp16 one: rows lane >> 2 and + 16, 16-B chunk lane & 3.
__device__ __forceinline__ void scr_f8x2(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const uint32_t (&da)[8],
                                           const uint32_t (&db)[8], uint32_t r, u32x4 (&o)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_write_b32 %4, %8\n\tds_write_b32 %5, %9\n\tds_write_b32 %6, %10\n\tds_write_b32 %7, %11\n\t"
      "ds_write_b32 %4, %12 offset:1024\n\tds_write_b32 %5, %13 offset:1024\n\tds_write_b32 %6, %14 offset:1024\n\tds_write_b32 %7, %15 offset:1024\n\t"
      "ds_read_b128 %0, %24\n\tds_read_b128 %1, %24 offset:1024\n\t"
      "ds_write_b32 %4, %16\n\tds_write_b32 %5, %17\n\tds_write_b32 %6, %18\n\tds_write_b32 %7, %19\n\t"
      "ds_write_b32 %4, %20 offset:1024\n\tds_write_b32 %5, %21 offset:1024\n\tds_write_b32 %6, %22 offset:1024\n\tds_write_b32 %7, %23 offset:1024\n\t"
      "ds_read_b128 %2, %24\n\tds_read_b128 %3, %24 offset:1024\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(da[0]), "v"(da[1]), "v"(da[2]), "v"(da[3]), "v"(da[4]), "v"(da[5]), "v"(da[6]),
        "v"(da[7]), "v"(db[0]), "v"(db[1]), "v"(db[2]), "v"(db[3]), "v"(db[4]), "v"(db[5]), "v"(db[6]), "v"(db[7]), "v"(r)
      : "memory");
#endif
}
// ONE fp8 plane of the block (the first half of scr_f8x2): o[0] = rows lane >> 2, o[1] = rows + 16
__device__ __forceinline__ void scr_f8x1(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const uint32_t (&da)[8], uint32_t r, u32x4 (&o)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_write_b32 %2, %6\n\tds_write_b32 %3, %7\n\tds_write_b32 %4, %8\n\tds_write_b32 %5, %9\n\t"
      "ds_write_b32 %2, %10 offset:1024\n\tds_write_b32 %3, %11 offset:1024\n\tds_write_b32 %4, %12 offset:1024\n\tds_write_b32 %5, %13 offset:1024\n\t"
      "ds_read_b128 %0, %14\n\tds_read_b128 %1, %14 offset:1024\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1])
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(da[0]), "v"(da[1]), "v"(da[2]), "v"(da[3]), "v"(da[4]), "v"(da[5]), "v"(da[6]), "v"(da[7]), "v"(r)
      : "memory");
#endif
}
// The V block of a merged Q,K,V launch in the 16x16 C/D layout (lane = token 16 tbl + m16, registers = 4 consecutive head dims
// 16 cbl + 4 q4 + e of unit k = 2 tbl + cbl) through the image TRANSPOSED - image rows = head dims, image columns = tokens: 16 two-byte
// writes per fragment at t0 (tbl = 0) / t1 (tbl = 1) + 1024 cbl + 64 e (t = q4 * 256 + swizzled token chunk + 2 (m16 & 7)).
__device__ __forceinline__ void scr_f16x2_t(uint32_t t0, uint32_t t1, const u32x2 (&da)[4], const u32x2 (&db)[4], uint32_t r, u32x4 (&o)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t a00 = da[0][0], a01 = da[0][1], a10 = da[1][0], a11 = da[1][1], a20 = da[2][0], a21 = da[2][1], a30 = da[3][0],
                 a31 = da[3][1];
  const uint32_t b00 = db[0][0], b01 = db[0][1], b10 = db[1][0], b11 = db[1][1], b20 = db[2][0], b21 = db[2][1], b30 = db[3][0],
                 b31 = db[3][1];
#define MV_T16(A, LO, HI, OFF)                                                                            \
  "ds_write_b16 " A ", " LO " offset:" #OFF "+0\n\tds_write_b16_d16_hi " A ", " LO " offset:" #OFF "+64\n\t" \
  "ds_write_b16 " A ", " HI " offset:" #OFF "+128\n\tds_write_b16_d16_hi " A ", " HI " offset:" #OFF "+192\n\t"
  asm volatile(
      MV_T16("%4", "%6", "%7", 0) MV_T16("%4", "%8", "%9", 1024) MV_T16("%5", "%10", "%11", 0) MV_T16("%5", "%12", "%13", 1024)
      "ds_read_b128 %0, %22\n\tds_read_b128 %1, %22 offset:1024\n\t"
      MV_T16("%4", "%14", "%15", 0) MV_T16("%4", "%16", "%17", 1024) MV_T16("%5", "%18", "%19", 0) MV_T16("%5", "%20", "%21", 1024)
      "ds_read_b128 %2, %22\n\tds_read_b128 %3, %22 offset:1024\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(t0), "v"(t1), "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(a20), "v"(a21), "v"(a30), "v"(a31), "v"(b00), "v"(b01),
        "v"(b10), "v"(b11), "v"(b20), "v"(b21), "v"(b30), "v"(b31), "v"(r)
      : "memory");
#undef MV_T16
#endif
}
// The reverse direction for two fp16 planes of one 32 x 32 block: coalesced 16-byte pieces (rows lane >> 2 and + 16, chunk
// lane & 3) are written into the image, the four C/D-layout units (tbl, cbl) of the lane (addresses r0..r3) are read back.
__device__ __forceinline__ void scr_f16_rev2(uint32_t wc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, uint32_t r0,
                                             uint32_t r1, uint32_t r2, uint32_t r3, u32x2 (&oa)[4], u32x2 (&ob)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_write_b128 %8, %9\n\tds_write_b128 %8, %10 offset:1024\n\t"
      "ds_read_b64 %0, %13\n\tds_read_b64 %1, %14\n\tds_read_b64 %2, %15\n\tds_read_b64 %3, %16\n\t"
      "ds_write_b128 %8, %11\n\tds_write_b128 %8, %12 offset:1024\n\t"
      "ds_read_b64 %4, %13\n\tds_read_b64 %5, %14\n\tds_read_b64 %6, %15\n\tds_read_b64 %7, %16\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(oa[0]), "=&v"(oa[1]), "=&v"(oa[2]), "=&v"(oa[3]), "=&v"(ob[0]), "=&v"(ob[1]), "=&v"(ob[2]), "=&v"(ob[3])
      : "v"(wc), "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(r0), "v"(r1), "v"(r2), "v"(r3)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}


// PP_RESLN3 accumulator init: one float4 of each of the bias, gamma and beta images (gamma at +3072 B, beta at +6144 B).
__device__ __forceinline__ void lds_read_bgb1(uint32_t addr, float4& bi, float4& ga, float4& be) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:3072\n\tds_read_b128 %2, %3 offset:6144\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(bi), "=&v"(ga), "=&v"(be)
               : "v"(addr)
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// One dword at a wave-uniform address through the scalar cache.  hipcc reads such a value with global_load + v_readfirstlane (the kernel also
// stores, so it cannot prove the location unclobbered) and waits for it with `s_waitcnt vmcnt(0)` — which here would drain the whole LDS-DMA
// operand pipeline at every use (GemmArgs::tile_both read that way: FFN-1 +22 us); s_load_dword only waits on lgkmcnt.
__device__ __forceinline__ int sload_i32(const int* p) {
  int v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
#endif
  return v;
}

// logical tile L of a persistent grid of G workgroups -> (tile_m, tile_n).  Mode 0: the grouped raster of gemm.h (all tile_m of a
// column group, then the next group: an A panel is fetched once per group).  Mode 1 (host: only when tm_count * GN % G == 0): window
// w = L / G of the sequence is (block b = w / ngroups of G consecutive (tile_m, tile_n-in-group) positions, column group g = w % ngroups),
// so a workgroup — and with the XCD remap a whole XCD's contiguous run of 32 positions — meets the SAME tile_m in ngroups consecutive
// iterations: its A panels can stay in that XCD's L2 across the whole N sweep while the W panels of the groups pass through.
__device__ __forceinline__ void raster_pp(const GemmArgs& a, int L, int G, int tm_count, int tn_count, int& tile_m, int& tile_n) {
  if (a.raster_mode == 1) {
    const int ngroups = tn_count / a.GN;
    const int w = L / G, r = L - w * G;
    const int b = w / ngroups, g = w - b * ngroups;
    const int p = b * G + r;
    tile_m = p / a.GN;
    tile_n = g * a.GN + (p - tile_m * a.GN);
  } else {
    raster(L, tm_count, tn_count, a.GN, tile_m, tile_n);
  }
}

template <int EPI, int RAW = 0, int X8 = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs a) {
  static_assert(EPI == PP_QK || EPI == PP_GELU || EPI == PP_RESLN3, "kernel kinds of the encoder layer");
  static_assert(RAW == (EPI != PP_RESLN3), "PP_QK / PP_GELU consume the raw stream (virtual LayerNorm), PP_RESLN3 produces it");
  constexpr bool IS_RES = (EPI == PP_RESLN3);
  static_assert(X8 == 0 || X8 == 1, "X8 = 1: MV_F16X8 (the fp8 correction sweep)");
  constexpr int WAITN = 2 * PP_F;
  constexpr int LDS_SCR = RAW ? PP_LDS_SCR_RAW : PP_LDS_SCR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int K = a.K, nk0 = K >> 6;                 // K-tiles of one sweep
  constexpr int nseg = X8 ? 2 : 1;                  // X8: the fp16 sweep, then the fp8 correction sweep
  // K-tiles of the fp8 sweep of a tile: K / 64 for both first-order terms (virtual K = 2 K), K / 128 when only the weight-side term
  // A_hi8 W_lo8 is swept (the SECOND halves of both operand rows, byte offset K).  a.x8_terms: 0 / 2 = both terms everywhere,
  // 1 = weight-side only, 3 = per Q / K / V block of a PP_QK launch by a.x8_aside_mask (bit `which` set = both terms).  The QKV
  // projection sweeps the A-side term for ONE of its three blocks only (Q): the A-operand rounding is common to Q, K and V, and
  // restoring it in one of them already brings the trained-like logit error back to the three-block level — measured on the MI355X
  // over the goldens l12_trained_s256 / _ragged and the reference's own 12-layer run (profiles/r04_d_qkv_aside_errors.txt), max:
  // all three 3.8e-4, Q only 4.2e-4, K only 4.7e-4, V only 4.4e-4, none 8.1e-4; oracle/precision_model.py predicts the same on the
  // first case (7.6e-4 / 2.9e-4 / 3.8e-4 for none / Q / all).  QKV launch 388 -> 335 us (310 with none), +2.9 % issue reports/s.
  // a.tile_both (engine.hip cls_aside): per 256-row tile, non-zero = a sequence in it is too short for the [CLS]-row form (the other rows' A-side
  // rounding reaches the [CLS] row averaged over the keys): such a tile sweeps BOTH terms where x8_terms asks for the weight-side one only, takes
  // nothing from cls_corr and keeps its lo8 planes — i.e. it runs the default form bit for bit, whatever the rest of the pass does.
  auto short_tile = [&](int tm) -> bool { return a.tile_both && sload_i32(a.tile_both + tm) != 0; };
  auto both_terms = [&](int tm, int tn) -> bool {
    if (a.x8_terms == 1) return short_tile(tm);
    if (a.x8_terms != 3) return true;
    const int which = (tn * 256 + a.col0) / MV_HIDDEN;  // 0 = Q, 1 = K, 2 = V
    return (a.x8_aside_mask >> which) & 1;
  };
  int i_nk8 = X8 ? nk0 : 0;    // of the tile being STAGED (issue cursor)
  size_t i_off8 = 0;
  const int tm_count = a.M >> 8, tn_count = a.N >> 8;
  const int ntiles = tm_count * tn_count;
  const int G = gridDim.x;
  const int bslot = xcd_remap(blockIdx.x, G);

  // ---- bias -> LDS (once per workgroup; RAW kernels: per tile, issue_stats)
  if constexpr (!RAW) {
    float* lb = (float*)(smem + PP_LDS_BIAS);
    for (int n = tid; n < a.N; n += 512) lb[n] = a.bias ? a.bias[n] : 0.f;
    for (int n = tid; n < MV_HIDDEN; n += 512) {  // N == 768: gamma at [768, 1536), beta at [1536, 2304) of the same image
      lb[MV_HIDDEN + n] = a.lng[n];
      lb[2 * MV_HIDDEN + n] = a.lnb[n];
    }
  }

  // ---- staging geometry: wave w fills slabs 2w, 2w+1 (8 rows x 128 B each) of every half-tile
  uint32_t offA[2], offB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rho = (2 * wave + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rho >> 1) & 7);
    const int rowA = (rho >> 6) * 128 + (rho & 63);
    const int rowB = (rho >> 5) * 64 + (rho & 31);
    offA[j] = (uint32_t)(rowA * K + c * 8) * 2u;
    offB[j] = (uint32_t)(rowB * K + c * 8) * 2u;
  }
  // issue cursor (wave-uniform): output tile being staged, its operand panels and K-tile index
  int i_it = 0, i_kt = 0, i_seg = 0;
  const char* iA = (const char*)a.A;
  const char* iW = (const char*)a.W;
  size_t tA = 0, tW = 0;  // X8: byte offsets of the issue tile's operand panels (the fp8 panels have the same row pitch, 2 K bytes)
  auto set_issue_seg = [&]() {  // X8, sweep 0: A_hi W_hi (fp16), 1: [A_lo8 | A_hi8] x [W_hi8 | W_lo8] (fp8)
    iA = (i_seg ? (const char*)a.A8 + i_off8 : (const char*)a.A) + tA;
    iW = (i_seg ? (const char*)a.W8 + i_off8 : (const char*)a.W) + tW;
  };
  auto set_issue_tile = [&](int it) {
    const int L = it * G + bslot;
    if (L < ntiles) {  // past the end: keep staging the last valid tile (never read, keeps the vmcnt ledger exact)
      int tm, tn;
      raster_pp(a, L, G, tm_count, tn_count, tm, tn);
      if constexpr (X8) {
        tA = (size_t)tm * 256 * K * 2;
        tW = (size_t)tn * 256 * K * 2;
        const bool both = both_terms(tm, tn);
        i_nk8 = both ? nk0 : nk0 >> 1;
        i_off8 = both ? 0 : (size_t)K;
      } else {
        iA = (const char*)a.A + (size_t)tm * 256 * K * 2;
        iW = (const char*)a.W + (size_t)tn * 256 * K * 2;
      }
    }
    if constexpr (X8) set_issue_seg();
  };
  set_issue_tile(0);
  // kind: 0 = a0, 1 = a1, 2 = b0, 3 = b1; issue order per K-tile: b0, a0, b1, a1 (= read order)
  auto issue = [&](auto kindc, auto parc) {
    constexpr int kind = decltype(kindc)::value;
    constexpr int par = decltype(parc)::value;
    const char* src;
    char* dst;
    if constexpr (kind < 2) {
      src = iA + (size_t)(kind * 64) * K * 2 + i_kt * 128;
      dst = smem + PP_LDS_A + par * 32768 + kind * 16384 + wave * 2048;
      glds16((const half_t*)(src + offA[0]), dst);
      glds16((const half_t*)(src + offA[1]), dst + 1024);
    } else {
      src = iW + (size_t)((kind - 2) * 32) * K * 2 + i_kt * 128;
      dst = smem + PP_LDS_B + par * 32768 + (kind - 2) * 16384 + wave * 2048;
      glds16((const half_t*)(src + offB[0]), dst);
      glds16((const half_t*)(src + offB[1]), dst + 1024);
    }
    if constexpr (kind == 1) {  // last half-tile of this K-tile: advance the cursor
      if (++i_kt == ((X8 && i_seg) ? i_nk8 : nk0)) {
        i_kt = 0;
        if constexpr (X8) {
          if (++i_seg == nseg) {
            i_seg = 0;
            set_issue_tile(++i_it);
          } else {
            set_issue_seg();
          }
        } else {
          set_issue_tile(++i_it);
        }
      }
    }
  };
  // psi-th half-tile of the stream (psi = phase + 1): psi % 4 -> kind, (psi / 4) & 1 -> LDS parity
  auto issue_psi = [&](auto psic) {
    constexpr int psi = decltype(psic)::value;
    constexpr int q = psi & 3;
    constexpr int kind = (q == 0) ? 2 : (q == 1) ? 0 : (q == 2) ? 3 : 1;
    issue(std::integral_constant<int, kind>{}, std::integral_constant<int, (psi >> 2) & 1>{});
  };

  // ---- fragment read addresses: operand row lane & 15 of a 16-row block, 16-byte chunk 4 kk + (lane >> 4) of its 128-byte K-tile row
  // (conflict-free on the unchanged half-tile image: each ds_read_b128 lane group covers the 16 slots mod 256 B once)
  const int m16 = lane & 15, q4 = lane >> 4;
  int rdA[2], rdB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int o = m16 * 128 + (((4 * kk + q4) ^ ((m16 >> 1) & 7)) << 4);
    rdA[kk] = PP_LDS_A + wr * 8192 + o;
    rdB[kk] = PP_LDS_B + wc * 4096 + o;
  }
  // fragment registers: A sub-tile = 4 token blocks x 2 K-steps, a W sub-tile = 2 column blocks x 2 K-steps.  X8 builds hold the
  // two chunks (q4, 4 + q4) of a row as ONE 8-dword value: the fp8 instruction (16x16x128) takes them as its 32 K-bytes — the same
  // assignment for both operands, so every byte column meets its partner once — and the fp16 sweep uses the halves as K-steps 0 / 1.
  half8_t Xf[X8 ? 1 : 4][X8 ? 1 : 2], Wx[X8 ? 1 : 2][X8 ? 1 : 2], Wy[X8 ? 1 : 2][X8 ? 1 : 2];
  intx8 Xp[X8 ? 4 : 1], Wxp[X8 ? 2 : 1], Wyp[X8 ? 2 : 1];
  auto ld16 = [&](int off) -> intx4 { return *(const intx4*)(smem + off); };
  auto read_a = [&](int par, int asub) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int o = par * 32768 + asub * 16384 + t4 * 2048;
      if constexpr (X8) {
        Xp[t4] = __builtin_shufflevector(ld16(rdA[0] + o), ld16(rdA[1] + o), 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) Xf[t4][kk] = *(const half8_t*)(smem + rdA[kk] + o);
      }
    }
  };
  auto read_b = [&](half8_t (&Wf)[X8 ? 1 : 2][X8 ? 1 : 2], intx8 (&Wp)[X8 ? 2 : 1], int par, int bsub) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int o = par * 32768 + bsub * 16384 + c2 * 2048;
      if constexpr (X8) {
        Wp[c2] = __builtin_shufflevector(ld16(rdB[0] + o), ld16(rdB[1] + o), 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) Wf[c2][kk] = *(const half8_t*)(smem + rdB[kk] + o);
      }
    }
  };

  floatx4 acc[8][4];  // [token block of 16][column block of 16]
  // the two fp16 planes of the 32 x 32 block i (token rows) x j (columns) of the residual tile at (mw0, nw0) by full-line loads
  // (16 rows x 64 B per instruction), parked in the four accumulators of that block: acc[2 i + pl][2 j + x] = plane pl (hi, lo),
  // rows crow + 16 x — the accumulator init transposes them into the C/D layout
  // X8 (MV_F16X8): the raw stream has NO lo fp16 plane.  The rows' low part is the lo8 plane of the stream's fp8 planes (out8, in place: rows [lo8 (768) | hi8 (768)];
  // 16 rows x 32 B per instruction), parked in registers 0, 1 of the lo slot — except the SPECIAL rows' (rows 0, 1 of a sequence = lanes crow < 2 of block row 0 or 2),
  // which take 2^11 x their low parts from the compact sp_lo_out (16 B; what the previous residual GEMM or the embedding kernel left there: a tile reads its
  // entries before it writes them); the accumulator init turns both into what a lo fp16 plane would hold
  auto park_residual = [&](int i, int mw0, int nw0) {
    const int crow = lane >> 2, cchunk = lane & 3;
    bool sp_c = false;  // this lane's row of block row i (x = 0) is a special row
    size_t sp_off = 0;
    if constexpr (X8) {
      if ((i & 1) == 0) {
        const int row0 = mw0 + 32 * i, b = row0 / a.S;
        sp_c = row0 < a.Mreal && b * a.S == row0 && crow < 2;
        sp_off = (size_t)(2 * b + crow) * MV_HIDDEN;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const size_t row = (size_t)(mw0 + i * 32 + x * 16 + crow);
          if constexpr (X8) {
            if (pl == 1) {
              float4 t;  // (every register written on both sides: with different register sets per side hipcc indexes the accumulator array dynamically = scratch)
              if ((i & 1) == 0 && x == 0 && sp_c) {
                t = *(const float4*)(a.sp_lo_out + sp_off + nw0 + j * 32 + 8 * cchunk);
              } else {
                const float2 u = *(const float2*)(a.out8 + row * (2 * MV_HIDDEN) + nw0 + j * 32 + 8 * cchunk);
                t.x = u.x; t.y = u.y; t.z = 0.f; t.w = 0.f;
              }
              acc[2 * i + 1][2 * j + x][0] = t.x; acc[2 * i + 1][2 * j + x][1] = t.y; acc[2 * i + 1][2 * j + x][2] = t.z; acc[2 * i + 1][2 * j + x][3] = t.w;
              continue;
            }
          }
          const float4 t = *(const float4*)((pl ? a.out16b : a.out16) + row * MV_HIDDEN + nw0 + j * 32 + 8 * cchunk);
          acc[2 * i + pl][2 * j + x][0] = t.x; acc[2 * i + pl][2 * j + x][1] = t.y;
          acc[2 * i + pl][2 * j + x][2] = t.z; acc[2 * i + pl][2 * j + x][3] = t.w;
        }
  };
  // One quadrant (64 tokens x 32 columns) of a K-tile: 16 MFMAs of 16x16x32 (fp16) or 8 of 16x16x128 (the fp8 correction sweep).
  auto mma_quadrant = [&](auto asubc, auto bc, auto f8c, const half8_t (&Wf)[X8 ? 1 : 2][X8 ? 1 : 2], const intx8 (&Wp)[X8 ? 2 : 1]) {
    constexpr int asub = decltype(asubc)::value;
    constexpr int b = decltype(bc)::value;
    if constexpr (X8) {
      if constexpr (decltype(f8c)::value) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
            acc[asub * 4 + t4][b * 2 + c2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Wp[c2], Xp[t4], acc[asub * 4 + t4][b * 2 + c2], 0, 0, 0,
                                                                                            a.x8_scale, 0, 0x7f7f7f7f);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const intx4 w4 = kk ? __builtin_shufflevector(Wp[c2], Wp[c2], 4, 5, 6, 7) : __builtin_shufflevector(Wp[c2], Wp[c2], 0, 1, 2, 3);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
              const intx4 x4 = kk ? __builtin_shufflevector(Xp[t4], Xp[t4], 4, 5, 6, 7) : __builtin_shufflevector(Xp[t4], Xp[t4], 0, 1, 2, 3);
              acc[asub * 4 + t4][b * 2 + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, w4), __builtin_bit_cast(half8_t, x4),
                                                                                   acc[asub * 4 + t4][b * 2 + c2], 0, 0, 0);
            }
          }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
            acc[asub * 4 + t4][b * 2 + c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[c2][kk], Xf[t4][kk], acc[asub * 4 + t4][b * 2 + c2], 0, 0, 0);
    }
  };

  // ---- building blocks: fragment set / quadrant of phase s (P = s & 3, LDS parity = s >> 2)
  auto read_set = [&](auto sc) {
    constexpr int s = decltype(sc)::value & 7;
    constexpr int P = s & 3, par = s >> 2;
    if constexpr (P == 0) { read_a(par, 0); read_b(Wx, Wxp, par, 0); }
    if constexpr (P == 1) read_b(Wy, Wyp, par, 1);
    if constexpr (P == 2) read_a(par, 1);
  };
  auto mma_set = [&](auto sc, auto f8c) {
    constexpr int P = decltype(sc)::value & 3;
    if constexpr (P == 0) mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, f8c, Wx, Wxp);
    if constexpr (P == 1) mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, f8c, Wy, Wyp);
    if constexpr (P == 2) mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, f8c, Wy, Wyp);
    if constexpr (P == 3) mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, f8c, Wx, Wxp);
  };
  auto interval = [&](auto sc, auto grpc, auto f8c, bool last_of_tile = false) {
    constexpr int s = decltype(sc)::value;
    constexpr int grp = decltype(grpc)::value;
    if constexpr (grp == 0) {  // matrix pipe first, then the NEXT phase's fragments
      __builtin_amdgcn_s_setprio(1);
      mma_set(sc, f8c);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // the next OUTPUT tile's first fragments are read after its accumulator init instead (run_tiles): keeping
      // 48 fragment VGPRs live across the epilogue + init costs more than one exposed LDS read per tile
      if (!(s == 7 && last_of_tile)) read_set(std::integral_constant<int, (s + 1) & 7>{});
      issue_psi(std::integral_constant<int, (s + 3 + PP_F) & 7>{});
    } else {  // this phase's fragments first, then the matrix pipe as the other half releases it
      read_set(sc);
      issue_psi(std::integral_constant<int, (s + 3 + PP_F) & 7>{});
      __builtin_amdgcn_sched_barrier(0);
      mma_set(sc, f8c);
    }
    if constexpr (X8) {
      // pin this interval's accumulators here: the scaled-MFMA builtin is a pure function to hipcc's IR passes, which otherwise
      // sink four intervals' worth of them across the barriers into one block (20 + 0 + 0 ... per interval instead of 4 each,
      // 18 fragment tuples live at once, 400-600 bytes of scratch per lane)
#if defined(__HIP_DEVICE_COMPILE__)
      constexpr int P = s & 3, a0 = (P >= 2) ? 4 : 0, bq = (P == 1 || P == 2) ? 2 : 0;  // the quadrant of phase P (mma_set)
      asm volatile("" : "+v"(acc[a0][bq]), "+v"(acc[a0 + 1][bq]), "+v"(acc[a0 + 2][bq]), "+v"(acc[a0 + 3][bq]), "+v"(acc[a0][bq + 1]),
                   "+v"(acc[a0 + 1][bq + 1]), "+v"(acc[a0 + 2][bq + 1]), "+v"(acc[a0 + 3][bq + 1]));
#endif
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- RAW: vstats of the 256 token rows (6 KiB) and the 256 bias' values (1 KiB) of persistent iteration `itn` -> LDS
  // images itn & 1.  Seven 1-KiB LDS-DMA pieces, two per wave 0..3, riding in the same vmcnt ledger: a piece only pushes
  // OLDER operand pieces out of a counted wait's window, so every wait stays conservative.  (tm, tn) = the tile's
  // coordinates (computed at the START of the previous tile: the raster's integer divisions stay out of the main loop).
  auto issue_stats = [&](int itn, int tm, int tn) {
    if (itn * G + bslot < ntiles && wave < 4) {
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int piece = 2 * wave + pc;  // 0..5: vstats, 6: bias', 7: none
        if (piece < 6) {
          const char* src = (const char*)a.lnstats + (size_t)tm * 6144 + piece * 1024 + lane * 16;
          glds16((const half_t*)src, smem + PP_LDS_STATS + (itn & 1) * 6144 + piece * 1024);
        } else if (piece == 6) {
          const char* src = (const char*)a.bias + (size_t)tn * 1024 + lane * 16;
          glds16((const half_t*)src, smem + PP_LDS_BIAS_T + (itn & 1) * 1024);
        }
      }
    }
  };
  if constexpr (RAW) {
    int tm = 0, tn = 0;
    if (bslot < ntiles) raster_pp(a, bslot, G, tm_count, tn_count, tm, tn);
    issue_stats(0, tm, tn);
  }

  // ---- prologue: half-tiles psi = 0 .. 2 + F in flight, psi 0 .. 2 landed
  issue_psi(std::integral_constant<int, 0>{});
  issue_psi(std::integral_constant<int, 1>{});
  issue_psi(std::integral_constant<int, 2>{});
  issue_psi(std::integral_constant<int, 3>{});
  issue_psi(std::integral_constant<int, 4>{});
  issue_psi(std::integral_constant<int, 5>{});
  issue_psi(std::integral_constant<int, 6>{});
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");  // lgkmcnt: the bias image writes
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  auto run_tiles = [&](auto grpc) {
  for (int it = 0;; ++it) {
    const int L = it * G + bslot;
    if (L >= ntiles) break;
    int tile_m, tile_n;
    raster_pp(a, L, G, tm_count, tn_count, tile_m, tile_n);
    int next_m = 0, next_n = 0;  // RAW: the workgroup's next tile (issue_stats); PP_RESLN3: its residual tile is requested
    const bool has_next = L + G < ntiles;  // during this tile's epilogue
    if (has_next) raster_pp(a, L + G, G, tm_count, tn_count, next_m, next_n);
    const int mw = (tile_m << 8) + wr * 128;  // first token row of this wave
    const int nw = (tile_n << 8) + wc * 64;   // first output column of this wave
    bool tile_short = false;                  // cls_aside: this row tile's sequence keeps the default form (read once per tile)
    if constexpr (X8) tile_short = short_tile(tile_m);
    const bool tile_both_terms = X8 && (a.x8_terms == 1 ? tile_short : both_terms(tile_m, tile_n));

    // ---- accumulator init: zero (RAW) or bias + LayerNorm(residual).  The images are read with inline-asm ds_reads: hipcc would put
    // `s_waitcnt vmcnt(0)` in front of a compiler-visible LDS load here (LDS-DMA in flight) and drain the operand pipeline once per tile.
    {
      const uint32_t baddr = (uint32_t)(PP_LDS_BIAS + (nw + 4 * q4) * 4);  // + 64 cb: this lane's four columns of column block cb
      // scratch addresses of this lane: C/D-layout units (token 16 tbl + m16 of a 32 x 32 block, columns 16 cbl + 4 q4 .. + 3 = slot
      // 2 cbl + (q4 >> 1), half q4 & 1 of the [32 rows][64 B] image) and coalesced layout (row = lane >> 2)
      const uint32_t scr = (uint32_t)(LDS_SCR + wave * 2048);
      const uint32_t sf = (uint32_t)((m16 >> 2) & 3);
      const uint32_t scr_c = scr + (lane >> 2) * 64 + ((((uint32_t)lane & 3) ^ (((uint32_t)lane >> 4) & 3)) << 4);
      if constexpr (IS_RES) {
        // residual tile by full-line loads (16 rows x 64 B per instruction), parked in the accumulators and transposed into the C/D layout
        float2 lnst[8];    // (mean, rstd) of this lane's eight token rows 16 tb + m16
        float2 lnp[8][3];  // the rows' vstats, loaded here and turned into (mean, rstd) only AFTER the residual tile's loads are issued
                           // (in source order hipcc waits for these loads first and the two memory latencies add up: +1.3 us per tile)
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
          const float2* pp = (const float2*)(a.lnstats + 6 * (size_t)(mw + tb * 16 + m16));
          lnp[tb][0] = pp[0]; lnp[tb][1] = pp[1]; lnp[tb][2] = pp[2];
        }
        const uint32_t ub = scr + m16 * 64 + (q4 & 1) * 8;  // unit (tbl, cbl) at ub + 1024 tbl, slot (2 cbl + (q4 >> 1)) ^ sf
        const uint32_t u00 = ub + ((((uint32_t)(q4 >> 1)) ^ sf) << 4), u01 = ub + (((2u + (uint32_t)(q4 >> 1)) ^ sf) << 4);
        // (tiles after the workgroup's first: the lines were requested block row by block row during the PREVIOUS tile's epilogue, each
        //  as soon as its accumulators had been stored — park_residual — so this tile's read phase runs under the last one's writes)
        if (it == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) park_residual(i, mw, nw);
        }
        if constexpr (X8) {
          // the parked lo8 bytes (two registers: 8 columns) / compact special-row parts (four registers, 2^11 x) -> the four registers of packed fp16 a lo plane
          // would hold: e4m3 x 2^-(11 + shift) and fp16 x 2^-11 are exact in fp16 (subnormals included).  (The whole tile in front of the loop below: converting block
          // row by block row inside it, in place or on the way into the transposition, costs 0.8 - 1 KB of scratch per lane with hipcc 7.2.)
          const half2_t s8 = {(half_t)(1.0f / (float)(2048 << MV_X8_ACT_SHIFT)), (half_t)(1.0f / (float)(2048 << MV_X8_ACT_SHIFT))};
          const half2_t s11 = {(half_t)(1.0f / 2048.0f), (half_t)(1.0f / 2048.0f)};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bool sp_c = false;
            if ((i & 1) == 0) {
              const int row0 = mw + 32 * i, sb0 = row0 / a.S;
              sp_c = row0 < a.Mreal && sb0 * a.S == row0 && (lane >> 2) < 2;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int x = 0; x < 2; ++x) {
                floatx4& l = acc[2 * i + 1][2 * j + x];
                const uint32_t w0 = f2u(l[0]), w1 = f2u(l[1]), w2 = f2u(l[2]), w3 = f2u(l[3]);
                half2_t c0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, false) * s8, c1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, true) * s8;
                half2_t c2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, false) * s8, c3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, true) * s8;
                if ((i & 1) == 0 && x == 0 && sp_c) {
                  c0 = __builtin_bit_cast(half2_t, w0) * s11; c1 = __builtin_bit_cast(half2_t, w1) * s11;
                  c2 = __builtin_bit_cast(half2_t, w2) * s11; c3 = __builtin_bit_cast(half2_t, w3) * s11;
                }
                l[0] = u2f(__builtin_bit_cast(uint32_t, c0)); l[1] = u2f(__builtin_bit_cast(uint32_t, c1));
                l[2] = u2f(__builtin_bit_cast(uint32_t, c2)); l[3] = u2f(__builtin_bit_cast(uint32_t, c3));
              }
          }
        }
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) lnst[tb] = ln_from_partials(lnp[tb][0], lnp[tb][1], lnp[tb][2], a.ln_eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma clang fp contract(off)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            u32x4 p[4];  // p[2 pl + x] = plane pl, rows crow + 16 x (park_residual)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
              for (int e = 0; e < 4; ++e) p[qq][e] = f2u(acc[2 * i + (qq >> 1)][2 * j + (qq & 1)][e]);
            u32x2 oh[4], ol[4];  // unit k = 2 tbl + cbl
            scr_f16_rev2(scr_c, p[0], p[1], p[2], p[3], u00, u01, u00 + 1024u, u01 + 1024u, oh, ol);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int tbl = k >> 1, cbl = k & 1;
              float4 bi, ga, be;
              lds_read_bgb1(baddr + (2 * j + cbl) * 64, bi, ga, be);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t wh = oh[k][e >> 1];  // scalar copies before the bit casts (see f2u)
                const half2_t h2 = __builtin_bit_cast(half2_t, wh);
                const uint32_t wl = ol[k][e >> 1];
                const half2_t l2 = __builtin_bit_cast(half2_t, wl);
                const float r = (float)h2[e & 1] + (float)l2[e & 1];
                const float t = (r - lnst[2 * i + tbl].x) * lnst[2 * i + tbl].y;
                acc[2 * i + tbl][2 * j + cbl][e] = __builtin_fmaf(t, ((const float*)&ga)[e], ((const float*)&be)[e]) + ((const float*)&bi)[e];
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int tb = 0; tb < 8; ++tb)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            acc[tb][cb][0] = 0.f; acc[tb][cb][1] = 0.f; acc[tb][cb][2] = 0.f; acc[tb][cb][3] = 0.f;
          }
      }
    }

    if constexpr (decltype(grpc)::value == 0) read_set(std::integral_constant<int, 0>{});
    auto two_ktiles = [&](auto f8c, bool last) {
      interval(std::integral_constant<int, 0>{}, grpc, f8c);
      interval(std::integral_constant<int, 1>{}, grpc, f8c);
      interval(std::integral_constant<int, 2>{}, grpc, f8c);
      interval(std::integral_constant<int, 3>{}, grpc, f8c);
      interval(std::integral_constant<int, 4>{}, grpc, f8c);
      interval(std::integral_constant<int, 5>{}, grpc, f8c);
      interval(std::integral_constant<int, 6>{}, grpc, f8c);
      interval(std::integral_constant<int, 7>{}, grpc, f8c, last);
    };
    for (int kt = 0; kt < nk0; kt += 2) {
      if constexpr (RAW) {  // every wave has left the previous tile's epilogue (>= 8 barriers ago): its stats image is free
        if (kt == 2) {
          issue_stats(it + 1, next_m, next_n);
          // rstd of THIS tile's 256 rows -> LDS, once per workgroup (waves 4..7, one row per lane; in the epilogue, per column
          // wave, the three pairs + rsq of four rows per lane cost 0.4 us per tile)
          if (wave >= 4) {
            const int row = (wave - 4) * 64 + lane;
            const uint32_t saddr = (uint32_t)(PP_LDS_STATS + (it & 1) * 6144 + row * 24);
            float2 p0, p1, p2;
            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2)
                         : "v"(saddr)
                         : "memory");
            const float rs = ln_from_partials(p0, p1, p2, a.ln_eps).y;
            const uint32_t waddr = (uint32_t)(PP_LDS_RSTD + (it & 1) * 1024 + row * 4);
            asm volatile("ds_write_b32 %0, %1" ::"v"(waddr), "v"(rs) : "memory");
          }
        }
      }
      two_ktiles(std::false_type{}, !X8 && kt + 2 >= nk0);
    }
    if constexpr (X8) {  // the correction sweep: the same intervals on the fp8 matrix path
      const int nk = nk0 + (tile_both_terms ? nk0 : nk0 >> 1);  // K-tiles of THIS output tile
      for (int kt = nk0; kt < nk; kt += 2) two_ktiles(std::true_type{}, kt + 2 >= nk);
    }

    // ---- epilogue (store only): every 32 x 32 block (i, j) = accumulators (2 i + tbl, 2 j + cbl) goes through the wave's [32][64 B] image
    {
      const uint32_t scr = (uint32_t)(LDS_SCR + wave * 2048);
      const uint32_t sf = (uint32_t)((m16 >> 2) & 3);
      const uint32_t scr_c = scr + (lane >> 2) * 64 + ((((uint32_t)lane & 3) ^ (((uint32_t)lane >> 4) & 3)) << 4);
      const int crow = lane >> 2, cchunk = lane & 3;  // coalesced layout: row (+16 for the second read), 16-B chunk
      int cls_b[2] = {-1, -1};  // X8: the sequence whose rows 0, 1 are this wave's rows mw + 64 j, + 1, or -1 (computed once per tile)
      if constexpr (X8) {
        // Row term of the special rows (GemmArgs::cls_corr; engine.hip): where the sweep above carried the weight-side correction term only, the A-side
        // term A_lo W_hi^T is added here for the TWO rows per sequence whose rounding can reach the pooler un-averaged — row b S, the [CLS] token, and
        // row b S + 1, where the embedding kernel computes the [SEP] token (the attention sinks of trained BERT heads) — from a skinny fp16 GEMM over
        // those rows (2^11 x the term, so that its operands stay normal fp16 numbers).  S % 64 == 0 and mw % 128 == 0: of this wave's 128 rows only
        // mw (+ 1) and mw + 64 (+ 1) can be such rows = token blocks 0 and 4, lanes m16 < 2.
        if (a.cls_corr || a.sp_lo_out || a.vlo_sp) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = mw + 64 * j, b = row / a.S;  // wave-uniform
            if (row < a.Mreal && b * a.S == row) cls_b[j] = b;
          }
        }
        if (a.cls_corr && !tile_both_terms) {  // (a tile whose sweep carried both terms has it already)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (cls_b[j] >= 0 && m16 < 2) {
              const float* cp = a.cls_corr + (size_t)(2 * cls_b[j] + m16) * a.N + nw + 4 * q4;
              floatx4 c[4];  // all four loads in flight before the first use: ONE exposed memory latency per tile that holds such a row
#pragma unroll
              for (int cb = 0; cb < 4; ++cb) c[cb] = *(const floatx4*)(cp + 16 * cb);
#if defined(__HIP_DEVICE_COMPILE__)
              asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
#endif
#pragma unroll
              for (int cb = 0; cb < 4; ++cb) {
                acc[4 * j][cb][0] = __builtin_fmaf(c[cb][0], 1.0f / 2048.0f, acc[4 * j][cb][0]);
                acc[4 * j][cb][1] = __builtin_fmaf(c[cb][1], 1.0f / 2048.0f, acc[4 * j][cb][1]);
                acc[4 * j][cb][2] = __builtin_fmaf(c[cb][2], 1.0f / 2048.0f, acc[4 * j][cb][2]);
                acc[4 * j][cb][3] = __builtin_fmaf(c[cb][3], 1.0f / 2048.0f, acc[4 * j][cb][3]);
              }
            }
          }
        }
      }
      if constexpr (IS_RES) {
        // vstats of the new raw rows: (sum, sum of squares) over this TILE's 256 columns.  Each wave reduces its 64 columns (a token
        // row's 64 values sit in the four lanes m16 + 16 q4), parks the 128 pairs in its own scratch, and after a workgroup barrier
        // the first column wave of each M-half adds the four shares in wave order (deterministic) and writes slot tile_n of the
        // rows' three pairs.  A second barrier keeps the scratch intact until it has been read (inline asm: see the accumulator init).
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = acc[tb][cb][e];
              s1 += v;
              s2 = __builtin_fmaf(v, v, s2);
            }
          // q4 pairs (0,1) / (2,3): rows of 16 lanes swapped; then the two halves of the wave
          const auto a1 = __builtin_amdgcn_permlane16_swap(f2u(s1), f2u(s1), false, false);
          const auto a2 = __builtin_amdgcn_permlane16_swap(f2u(s2), f2u(s2), false, false);
          const float h1 = u2f(a1[0]) + u2f(a1[1]), h2 = u2f(a2[0]) + u2f(a2[1]);
          const auto t1 = __builtin_amdgcn_permlane32_swap(f2u(h1), f2u(h1), false, false);
          const auto t2 = __builtin_amdgcn_permlane32_swap(f2u(h2), f2u(h2), false, false);
          float2 st;
          st.x = u2f(t1[0]) + u2f(t1[1]);
          st.y = u2f(t2[0]) + u2f(t2[1]);
          const uint32_t waddr = scr + (uint32_t)(tb * 16 + m16) * 8;  // the four lanes of a row write the same pair
          asm volatile("ds_write_b64 %0, %1" ::"v"(waddr), "v"(st) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (wc == 0) {  // rows i * 32 + l31 of this M-half: the lower half-wave takes i = 0, 1, the upper one i = 2, 3
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int row = (2 * hi + ii) * 32 + l31;
            const uint32_t raddr = (uint32_t)(LDS_SCR + wr * 4 * 2048) + (uint32_t)row * 8;
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
      }
      {
        // unit (tbl, cbl) of a block: token 16 tbl + m16, columns 16 cbl + 4 q4 .. + 3 -> image row 16 tbl + m16, slot 2 cbl + (q4 >> 1),
        // half q4 & 1.  fp16: 8-byte units; fp8 planes: 4-byte units at byte 4 q4 of slot cbl (columns 16 cbl .. + 15 of a 32-column half)
        const uint32_t ub = scr + m16 * 64 + (q4 & 1) * 8;
        const uint32_t u00 = ub + ((((uint32_t)(q4 >> 1)) ^ sf) << 4), u01 = ub + (((2u + (uint32_t)(q4 >> 1)) ^ sf) << 4);
        const uint32_t ub8 = scr + m16 * 64 + q4 * 4;  // + ((slot ^ sf) << 4), slot = 2 j + cbl of the 64-byte plane row
        half_t* obase;    // pointer of (row crow, chunk cchunk) of block (i = 0, j = 0)
        size_t rstride;   // elements between image rows in the output
        size_t istride;   // elements between i blocks (32 token rows)
        size_t jstride;   // elements between j blocks
        bool live = true;
        bool vtile = false;  // PP_QK: this tile belongs to the V block and is stored as V^T
        ptrdiff_t lo_delta = 0;  // PP_QK X8, short passes: elements from this tile's output buffer to its second (lo) plane
        // V^T image: rows = head dims 16 cbl + 4 q4 + e, columns = tokens 16 tbl + m16 (2 bytes), 16-B chunk (token >> 3) ^ q4
        const uint32_t tb0 = scr + q4 * 256 + (m16 & 7) * 2;
        const uint32_t tt0 = tb0 + ((((uint32_t)(m16 >> 3)) ^ (uint32_t)q4) << 4), tt1 = tb0 + (((2u + (uint32_t)(m16 >> 3)) ^ (uint32_t)q4) << 4);
        if constexpr (EPI == PP_GELU || IS_RES) {
          obase = a.out16 + (size_t)(mw + crow) * a.N + nw + 8 * cchunk;
          rstride = a.N; istride = (size_t)32 * a.N; jstride = 32;
        } else {  // PP_QK: one head per wave; S % 64 == 0 so a 128-row block may span two batch rows
          // columns [0,768) -> Q, [768,1536) -> K, [1536,2304) -> V^T (a merged launch); col0 = 768: K (and V) only
          const int colg = nw + a.col0;
          const int which = colg >= 2 * MV_HIDDEN ? 2 : (colg >= MV_HIDDEN ? 1 : 0);
          const int head = (colg - which * MV_HIDDEN) >> 6;
          vtile = which == 2;
          if (vtile) {  // image rows = head dims, image columns = tokens (scr_f16x2_t)
            obase = a.vt + ((size_t)head * MV_HEAD_DIM + crow) * a.S + 8 * cchunk;
            rstride = a.S; istride = 0; jstride = (size_t)32 * a.S;
            if (a.vt_lo) lo_delta = a.vt_lo - a.vt;
          } else {
            obase = (which ? a.k : a.q) + (size_t)head * a.S * MV_HEAD_DIM + (size_t)crow * MV_HEAD_DIM + 8 * cchunk;
            rstride = MV_HEAD_DIM; istride = 0; jstride = 32;  // the batch-row part is added per i below
            if (a.vt_lo) lo_delta = which ? a.k_lo - a.k : a.q_lo - a.q;
          }
        }
        // RAW: bias' of this lane's columns (per-tile LDS image: one float4 per column block) and rstd of its eight token rows
        // (from the vstats image), applied as fma(rstd, acc, bias')
        float4 rbv[4];
        float rrs[8];
        if constexpr (RAW) {
          const uint32_t baddr = (uint32_t)(PP_LDS_BIAS_T + (it & 1) * 1024 + (wc * 64 + 4 * q4) * 4);
          const uint32_t saddr = (uint32_t)(PP_LDS_RSTD + (it & 1) * 1024 + (wr * 128 + m16) * 4);  // this wave's 128 rows
          asm volatile(
              "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:64\n\tds_read_b128 %2, %12 offset:128\n\t"
              "ds_read_b128 %3, %12 offset:192\n\t"
              "ds_read_b32 %4, %13\n\tds_read_b32 %5, %13 offset:64\n\tds_read_b32 %6, %13 offset:128\n\t"
              "ds_read_b32 %7, %13 offset:192\n\tds_read_b32 %8, %13 offset:256\n\tds_read_b32 %9, %13 offset:320\n\t"
              "ds_read_b32 %10, %13 offset:384\n\tds_read_b32 %11, %13 offset:448\n\ts_waitcnt lgkmcnt(0)"
              : "=&v"(rbv[0]), "=&v"(rbv[1]), "=&v"(rbv[2]), "=&v"(rbv[3]), "=&v"(rrs[0]), "=&v"(rrs[1]), "=&v"(rrs[2]), "=&v"(rrs[3]),
                "=&v"(rrs[4]), "=&v"(rrs[5]), "=&v"(rrs[6]), "=&v"(rrs[7])
              : "v"(baddr), "v"(saddr)
              : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        (void)ub8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mb = mw + i * 32;
          half_t* ob = obase + i * istride;
          if constexpr (EPI == PP_QK) {
            live = mb < a.Mreal;
            const int b = mb / a.S, s0 = mb - b * a.S;
            if (!vtile) ob = obase + ((size_t)b * MV_HEADS * a.S + s0) * MV_HEAD_DIM;
            else ob = obase + (size_t)b * MV_HEADS * MV_HEAD_DIM * a.S + s0;
          }
          u32x2 d[2][4];  // [j][k = 2 tbl + cbl]
          u32x4 o[4];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int tb = 2 * i + (k >> 1), cb = 2 * j + (k & 1);
              float v0 = acc[tb][cb][0], v1 = acc[tb][cb][1], v2 = acc[tb][cb][2], v3 = acc[tb][cb][3];
              if constexpr (RAW) {  // lane = token row 16 tb + m16, registers = 4 consecutive columns
                v0 = __builtin_fmaf(rrs[tb], v0, rbv[cb].x); v1 = __builtin_fmaf(rrs[tb], v1, rbv[cb].y);
                v2 = __builtin_fmaf(rrs[tb], v2, rbv[cb].z); v3 = __builtin_fmaf(rrs[tb], v3, rbv[cb].w);
              }
              if constexpr (EPI == PP_GELU) {
                float2_t a01, a23;
                a01.x = v0; a01.y = v1; a23.x = v2; a23.y = v3;
                a01 = gelu_erf2(a01);
                a23 = gelu_erf2(a23);
                v0 = a01.x; v1 = a01.y; v2 = a23.x; v3 = a23.y;
                if constexpr (X8) {  // the fp8 planes below are taken from the activated values
                  acc[tb][cb][0] = v0; acc[tb][cb][1] = v1; acc[tb][cb][2] = v2; acc[tb][cb][3] = v3;
                }
              }
              pin_f32x4(v0, v1, v2, v3);
              d[j][k][0] = pack_h2(v0, v1);
              d[j][k][1] = pack_h2(v2, v3);
            }
          if (EPI == PP_QK && vtile) scr_f16x2_t(tt0, tt1, d[0], d[1], scr_c, o);
          else scr_f16x2(u00, u01, u00 + 1024u, u01 + 1024u, d[0], d[1], scr_c, o);
          if (live) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              half_t* op = ob + j * jstride;
              *(u32x4*)op = o[2 * j];
              *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
            }
          }
          if constexpr (EPI == PP_QK && X8) {
            // V of the special rows as hi + lo (GemmArgs::vlo_sp): lane = token 16 tb + m16 (tb = 2 i), registers = head dims 16 cb + 4 q4 + e of this wave's head
            if (a.vlo_sp && vtile && (i & 1) == 0 && cls_b[i >> 1] >= 0 && m16 < 2) {
              const int head = (nw + a.col0 - 2 * MV_HIDDEN) >> 6;
              half_t* vp = a.vlo_sp + ((size_t)(cls_b[i >> 1] * MV_HEADS + head) * MV_HEAD_DIM + 4 * q4) * 2 + m16;
#pragma unroll
              for (int cb = 0; cb < 4; ++cb) {
                float vv[4] = {__builtin_fmaf(rrs[2 * i], acc[2 * i][cb][0], rbv[cb].x), __builtin_fmaf(rrs[2 * i], acc[2 * i][cb][1], rbv[cb].y),
                               __builtin_fmaf(rrs[2 * i], acc[2 * i][cb][2], rbv[cb].z), __builtin_fmaf(rrs[2 * i], acc[2 * i][cb][3], rbv[cb].w)};
                pin_f32x4(vv[0], vv[1], vv[2], vv[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) vp[(16 * cb + e) * 2] = (half_t)((vv[e] - (float)(half_t)vv[e]) * 2048.0f);
              }
            }
            // short passes (a.vt_lo set: padded length <= 128): Q, K and V^T as TWO fp16 planes each — what is left of the precise mode's error is the
            // fp16 storage of Q, K, V and P, which attention averages over the keys, so short sequences feel it most (profiles/r05_f_length_envelope.txt);
            // the attention kernel of these passes adds the first-order terms K_lo Q_hi + K_hi Q_lo and V_lo P_hi + V_hi P_lo (attention_v2.h VLO).
            // Same values as the first pass, recomputed.
            if (a.vt_lo) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int tb = 2 * i + (k >> 1), cb = 2 * j + (k & 1);
                  float v0 = __builtin_fmaf(rrs[tb], acc[tb][cb][0], rbv[cb].x), v1 = __builtin_fmaf(rrs[tb], acc[tb][cb][1], rbv[cb].y);
                  float v2 = __builtin_fmaf(rrs[tb], acc[tb][cb][2], rbv[cb].z), v3 = __builtin_fmaf(rrs[tb], acc[tb][cb][3], rbv[cb].w);
                  pin_f32x4(v0, v1, v2, v3);
                  d[j][k][0] = pack_h2(v0 - (float)(half_t)v0, v1 - (float)(half_t)v1);
                  d[j][k][1] = pack_h2(v2 - (float)(half_t)v2, v3 - (float)(half_t)v3);
                }
              if (vtile) scr_f16x2_t(tt0, tt1, d[0], d[1], scr_c, o);
              else scr_f16x2(u00, u01, u00 + 1024u, u01 + 1024u, d[0], d[1], scr_c, o);
              if (live) {
                half_t* ol = ob + lo_delta;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  half_t* op = ol + j * jstride;
                  *(u32x4*)op = o[2 * j];
                  *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
                }
              }
            }
          }
          if constexpr (IS_RES && !X8) {  // second plane: lo = fp16(r - hi), same addresses in the lo buffer (X8: the lo8 plane below is the stream's low part)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int tb = 2 * i + (k >> 1), cb = 2 * j + (k & 1);
                const float v0 = acc[tb][cb][0], v1 = acc[tb][cb][1], v2 = acc[tb][cb][2], v3 = acc[tb][cb][3];
                d[j][k][0] = pack_h2(v0 - (float)(half_t)v0, v1 - (float)(half_t)v1);
                d[j][k][1] = pack_h2(v2 - (float)(half_t)v2, v3 - (float)(half_t)v3);
              }
            scr_f16x2(u00, u01, u00 + 1024u, u01 + 1024u, d[0], d[1], scr_c, o);
            half_t* ol = ob + (a.out16b - a.out16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              half_t* op = ol + j * jstride;
              *(u32x4*)op = o[2 * j];
              *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
            }
          }
          if constexpr (X8 && (IS_RES || EPI == PP_GELU)) {
            // MV_F16X8: the [lo8 | hi8] planes of this output (the A8 operand of the next GEMM's correction sweep): rows of
            // 2 N bytes, lo8 of column n at byte n, hi8 at byte N + n; one 16-B store per lane, plane and 16-row half
            uint32_t dl[8], dh[8];  // [4 tbl + 2 j + cbl]: the dword of token 16 tbl + m16, columns 32 j + 16 cbl + 4 q4 .. + 3
            float vmax8 = 0.f;      // max |value| of the block (saturation accounting, common.h)
            const uint32_t w8 = ub8 + (sf << 4);
            uint8_t* o8 = a.out8 + (size_t)(mb + crow) * (2 * a.N) + nw + 16 * cchunk;
            // GemmArgs::out8_hi_only (engine.hip cls_aside: the consumer sweeps the weight-side term only): no lo8 plane
            bool hi_only = false;
            if constexpr (X8) hi_only = a.out8_hi_only && !tile_short;
            // the special rows' low parts, compact (GemmArgs::sp_lo_out): token block 2 i of an even block row holds rows mw + 32 i + m16
            if (a.sp_lo_out && (i & 1) == 0 && cls_b[i >> 1] >= 0 && m16 < 2) {
              half_t* sp = a.sp_lo_out + (size_t)(2 * cls_b[i >> 1] + m16) * a.N + nw + 4 * q4;
#pragma unroll
              for (int cb = 0; cb < 4; ++cb) {
                half4_t l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float v = acc[2 * i][cb][e];
                  l[e] = (half_t)((v - (float)(half_t)v) * 2048.0f);
                }
                *(half4_t*)(sp + 16 * cb) = l;
              }
            }
#pragma unroll
            for (int tbl = 0; tbl < 2; ++tbl)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int cbl = 0; cbl < 2; ++cbl) {
                  const int tb = 2 * i + tbl, cb = 2 * j + cbl, u = 4 * tbl + 2 * j + cbl;
                  if (hi_only) {
                    dh[u] = x8_hi4_in_range(acc[tb][cb][0], acc[tb][cb][1], acc[tb][cb][2], acc[tb][cb][3]);
                  } else if constexpr (EPI == PP_GELU) {  // d[j][2 tbl + cbl] still holds this unit's packed fp16 output words
                    const uint32_t p01 = d[j][2 * tbl + cbl][0], p23 = d[j][2 * tbl + cbl][1];
                    x8_planes4_in_range_packed(acc[tb][cb][0], acc[tb][cb][1], acc[tb][cb][2], acc[tb][cb][3], p01, p23, dh[u], dl[u]);
                  } else {
                    x8_planes4_in_range(acc[tb][cb][0], acc[tb][cb][1], acc[tb][cb][2], acc[tb][cb][3], dh[u], dl[u]);
                  }
                  vmax8 = x8_absmax4(vmax8, acc[tb][cb][0], acc[tb][cb][1], acc[tb][cb][2], acc[tb][cb][3]);
                }
            if (x8_any_out_of_range(vmax8)) {  // rare (never on the models measured): redo this block with the clamps and count its out-of-range elements
              int n = 0;
#pragma unroll
              for (int tbl = 0; tbl < 2; ++tbl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                  for (int cbl = 0; cbl < 2; ++cbl) {
                    const int tb = 2 * i + tbl, cb = 2 * j + cbl;
                    x8_planes4(acc[tb][cb][0], acc[tb][cb][1], acc[tb][cb][2], acc[tb][cb][3], dh[4 * tbl + 2 * j + cbl], dl[4 * tbl + 2 * j + cbl]);
                    n += x8_count4(acc[tb][cb][0], acc[tb][cb][1], acc[tb][cb][2], acc[tb][cb][3]);
                  }
              x8_sat_add(a.x8_sat, n);
            }
            if (hi_only) {
              scr_f8x1(w8, w8 ^ 16u, w8 ^ 32u, w8 ^ 48u, dh, scr_c, o);
              *(u32x4*)(o8 + a.N) = o[0];
              *(u32x4*)(o8 + a.N + (size_t)16 * (2 * a.N)) = o[1];
            } else {
              scr_f8x2(w8, w8 ^ 16u, w8 ^ 32u, w8 ^ 48u, dl, dh, scr_c, o);
              *(u32x4*)o8 = o[0];
              *(u32x4*)(o8 + (size_t)16 * (2 * a.N)) = o[1];
              *(u32x4*)(o8 + a.N) = o[2];
              *(u32x4*)(o8 + a.N + (size_t)16 * (2 * a.N)) = o[3];
            }
          }
          if constexpr (IS_RES) {  // block row i is out: request block row i of the NEXT tile's residual into its registers
            if (has_next) park_residual(i, (next_m << 8) + wr * 128, (next_n << 8) + wc * 64);
          }
        }
      }
    }
  }
  };  // run_tiles
  if (wr == 0) run_tiles(std::integral_constant<int, 0>{});
  else run_tiles(std::integral_constant<int, 1>{});

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}
