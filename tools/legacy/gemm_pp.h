// Persistent ping-pong MFMA GEMM for the BERT projections at bench scale:
//   C[M,N] = A[M,K] (fp16) x W[N,K]^T (fp16, torch Linear.weight layout) + bias, fp32 accumulate,
// with the reference's elementwise work fused (HF BertSelfAttention / BertSelfOutput / BertIntermediate /
// BertOutput as invoked from custom_PTM_embedder.py:228).
//
// Structure (cdna_hip_programming.md §5 "256^2 8-phase template", T1-T5; MI355X_MICROARCH.md "Two waves per SIMD"):
//   * 256x256x64 tile, 512 threads = 8 waves as 2(M) x 4(N), 128x64 of C per wave (128 accumulator VGPRs),
//     v_mfma_f32_32x32x16_f16, ONE workgroup per CU, grid = #CUs, each workgroup walks a strided list of
//     output tiles (persistent): the K-tile stream never drains between output tiles, so the next tile's
//     operands are already in flight while the epilogue stores.
//   * The two M-halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run the same phase
//     sequence ONE s_barrier apart: while one wave of a SIMD issues its 8 MFMAs of a phase (256 matrix-pipe
//     cycles) its partner reads the next fragments from LDS and issues its LDS-DMA, then they swap.  Every
//     s_barrier is workgroup-wide; a phase is [ds_read + DMA issue] barrier [MFMA] barrier.
//   * A K-tile is 4 phases = the 4 quadrants (64 x 32) of the wave's C block in the order (a0,b0) (a0,b1)
//     (a1,b1) (a1,b0'): each phase reads at most ONE operand half-tile (8 or 4 ds_read_b128 per wave), and
//     the 4th reads b0 of the NEXT K-tile into the register set b1 just vacated (the two W register sets swap
//     roles every K-tile; the loop is unrolled by two K-tiles so this is static).
//   * Operand half-tiles (128 rows x 64 halfs = 16 KiB = 2 LDS-DMA instructions per wave) are the unit of
//     staging: LDS holds two K-tiles x {a0,a1,b0,b1} = 128 KiB, half-tile h(phi) is read in phase phi only and
//     re-issued for the K-tile two ahead DIST phases before its read (DIST <= 6: a region is rewritten no
//     sooner than two phases after its last read, which covers the one-barrier skew between the wave halves).
//     One counted `s_waitcnt vmcnt(2 (DIST-1))` per phase retires exactly the half-tile the NEXT phase reads;
//     loads stay in flight across barriers (raw s_barrier, never __syncthreads), 2 DIST KiB x 8 per CU.
//   * LDS image of a half-tile is lane-linear per DMA instruction (1 KiB = 8 rows x 128 B); the bank swizzle
//     (16-B chunk c of row r at slot c ^ ((r >> 1) & 7)) is applied on the per-lane SOURCE address and on the
//     ds_read_b128 (rule 21).  The DMA uses the SGPR-base + 32-bit-VGPR-offset form.
//   * Orientation: SWAP computes C^T fragments (W rows as the MFMA A operand), so a lane holds 4 CONSECUTIVE
//     output columns of one token row per register group -> 16-byte epilogue stores (fp32 directly, fp16 after
//     one v_permlane32_swap per dword, T21).  The V^T epilogue uses the other orientation (lane = head dim,
//     registers = consecutive tokens) for the same reason.
//   * bias is staged once per workgroup into LDS and the accumulators START from it (and, for the residual
//     epilogue, from x + bias), so the epilogue has no loads; non-residual kernels have no VGPR-destination
//     VMEM load anywhere in the persistent loop (hipcc would drain the DMA queue with vmcnt(0) at each one).
//   * Tile order: logical tile sequence = (column group of GN tiles) > tile_m > tile_n-in-group; per persistent
//     iteration the 256 concurrent tiles are consecutive in it and each XCD takes a contiguous run of 32 (the
//     group's W panels stay in that XCD's L2 while it sweeps the A row panels).
//   * K accumulation order per output element is ascending 16-wide MFMA steps, the same as gemm128/gemm256.
#pragma once
#include "common.h"
#include <type_traits>

#include "gemm.h"

enum { PP_F32 = 0, PP_QK = 1, PP_VT = 2, PP_GELU = 3, PP_RES = 4, PP_F16 = 5, PP_RESLN = 6, PP_RESLN2 = 7, PP_RESLN3 = 8 };
// PP_RESLN: PP_RES whose residual tile is the PRE-LayerNorm stream: the accumulators start from
//   LN(x) + bias = fma((x - mean) * rstd, gamma, beta) + bias   (row statistics from ln_kernel<stats>),
// the same IEEE operations ln_row_store performs, so the result equals PP_RES on a normalised stream bit for bit
// while the LayerNorm kernel no longer writes the fp32 stream back (201 MB per LayerNorm at the bench shape).
// PP_RESLN2 + RAW consumers ("virtual LayerNorm": no LayerNorm kernel at all).  By linearity
//   W LN(r) + b = rstd * (W'' r) + b',  W''[n][k] = W[n][k] gamma[k] - mean_k(W[n][.] gamma[.]),  b' = b + W beta
// (the row mean of r drops out against the row-centred weights), so a consumer GEMM (RAW = 1: PP_QK — Q, K and V^T in
// one launch — and PP_GELU; PP_VT, the separate V^T launch of round 1, survives for tools/gemm_bench.hip only) takes the RAW stream rounded to fp16 as its A operand, the folded weights W'' (prepared once on the
// host), starts its accumulators from zero and applies  fma(rstd_row, acc, b'_col)  in the epilogue; the 256 rows'
// statistics of the workgroup's next tile arrive by six LDS-DMA pieces (and its 256 bias' values by a seventh) into
// 2 x 6 KiB (2 x 1 KiB) images.  The producer (PP_RESLN2 = PP_RESLN that ALSO writes the fp16 copy of the raw stream and the
// rows' "vstats": per row and 256-column tile the (sum, sum of squares), the four column waves' shares added in wave
// order through LDS) replaces the LayerNorm kernel's second pass over the stream, and every consumer turns the three
// pairs of a row into (mean, rstd) itself (common.h ln_from_partials) — no statistics kernel between the GEMMs
// (round 1 / early round 2 ran a 6 us ln_finalize launch after every residual GEMM: 22 launches per pass).
// PP_RESLN3 = PP_RESLN2 with the raw stream kept as TWO fp16 planes instead of fp32 + an fp16 copy:
//   hi = fp16(r)  (exactly the operand the RAW consumers read),  lo = fp16(r - hi),  r ~= hi + lo to 2^-22 relative
// (fp32 carries 2^-24).  The residual tile is read as hi + lo (same bytes as fp32) and written as hi, lo: 100 MB less
// per launch than fp32 + fp16 copy at the bench shape, and no fp32 transposition pass in the epilogue.
// timing ablations (wrong results by construction; cdna_hip_programming.md §5.4 rules 8/17)
enum { PP_ABL_NODMA = 1, PP_ABL_NOMFMA = 2, PP_ABL_NOREAD = 4, PP_ABL_NOEPI = 8, PP_ABL_NOPRIO = 16, PP_ABL_NOSTAGGER = 32,
       PP_ABL_CLK = 64, PP_ABL_B34 = 128 };  // B34 (timing only): PP_RES moves 3/4 of its residual bytes (every 4th line neither loaded nor stored)  // CLK (tools/gemm_bench.hip): per-wave s_memtime sums of accumulator init / main loop / epilogue -> a.clk[(wg 8 + wave) 4 ..]

#define PP_LDS_A 0           // [par][a][wr][64 rows][128 B]
#define PP_LDS_B 65536       // [par][b][wc][32 rows][128 B]
#define PP_LDS_BIAS 131072   // [N] fp32 (N <= 3072)
#define PP_LDS_SCR (PP_LDS_BIAS + MV_INTER * 4)  // 8 waves x 2 KiB: wave-private transposition scratch
#define PP_LDS_BYTES (PP_LDS_SCR + 8 * 2048)     // 159,744 of 163,840
// RAW kernels re-partition everything above the operand ring: [2][256] bias' of the tile | [2][256 rows][3][sum, sumsq] | scratch
#define PP_LDS_BIAS_T PP_LDS_BIAS                  // 2 x 1 KiB
#define PP_LDS_STATS (PP_LDS_BIAS + 2048)          // 2 x 6 KiB
#define PP_LDS_SCR_RAW (PP_LDS_STATS + 2 * 6144)   // 8 waves x 2 KiB
#define PP_LDS_RSTD (PP_LDS_SCR_RAW + 8 * 2048)    // 2 x 256 rstd of the tile's rows (computed once per workgroup, in the main loop)
#define PP_LDS_BYTES_RAW (PP_LDS_RSTD + 2048)      // 163,840 = all of the CU's LDS

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  half2_t h;
  h[0] = (half_t)a;
  h[1] = (half_t)b;
  return __builtin_bit_cast(uint32_t, h);
}

// 16 fp32 of one 32x32 fragment in the "4 consecutive elements per register group" orientation ->
// two 16-byte stores per lane: after the swaps lanes 0-31 hold elements 16p..16p+7 and lanes 32-63 hold
// 16p+8..16p+15 of the fragment's 32-wide contiguous axis.  `rowptr` = this lane's row start (fp16).
template <typename T>
__device__ __forceinline__ void keep_live(const T& v) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass of hipcc parses kernel bodies too and rejects the VGPR constraint
  asm volatile("" ::"v"(v));
#endif
}

template <typename F>
__device__ __forceinline__ void store_frag_f16(const floatx16& v, half_t* rowptr, int hi, F f) {
  uint32_t d[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    d[g][0] = pack_h2(f(v[4 * g + 0]), f(v[4 * g + 1]));
    d[g][1] = pack_h2(f(v[4 * g + 2]), f(v[4 * g + 3]));
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    auto rx = __builtin_amdgcn_permlane32_swap(d[2 * p][0], d[2 * p + 1][0], false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(d[2 * p][1], d[2 * p + 1][1], false, false);
    uint4 o;
    o.x = rx[0]; o.y = ry[0]; o.z = rx[1]; o.w = ry[1];
    *(uint4*)(rowptr + 16 * p + 8 * hi) = o;
  }
}

// (hipcc/ROCm 7.2: __builtin_bit_cast applied directly to an ext_vector ELEMENT expression reads element 0 —
// always go through a scalar copy)
__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __uint_as_float(x); }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Wave-private transposition through a [32 rows][64 B] LDS image (16-B chunk c of row r at slot c ^ ((r >> 2) & 3)):
// the MFMA C/D layout gives a lane 8 or 16 contiguous bytes of ONE row (lane = row), so storing it directly makes
// every wave-store touch 64 scattered 16-B pieces (measured: ~64 cycles of address processing per instruction,
// 3.6 us per 256^2 tile).  Through the image each store covers 16 rows x 64 contiguous bytes.  Inline asm keeps
// these LDS accesses out of hipcc's LDS-DMA alias bookkeeping (it would put `s_waitcnt vmcnt(0)` before them and
// wait for the epilogue's own stores); LDS executes a wave's instructions in order, so the read-after-write
// needs no wait, only the read results do (same statement, guide §5.7 form i).
// Two fp16 fragments (j = 0, 1) per statement: LDS executes a wave's instructions in order, so the second
// fragment's writes may follow the first one's reads into the same image without a wait in between.
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
// The V block of a merged Q,K,V launch (PP_QK orientation: lane = token, registers = 4 consecutive head dims) goes
// through the image TRANSPOSED - image rows = head dims, image columns = tokens - so that the read side and the
// stores are the V^T ones: 16 two-byte writes per fragment (row 8 g + 4 hi + e at a0 / a1 = a0 ^ 32 plus
// 512 g + 64 e; the slot swizzle (row >> 2) & 3 = (2 g + hi) & 3 alternates between hi and hi ^ 2).
__device__ __forceinline__ void scr_f16x2_t(uint32_t a0, uint32_t a1, const u32x2 (&da)[4], const u32x2 (&db)[4], uint32_t r,
                                            u32x4 (&o)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t a00 = da[0][0], a01 = da[0][1], a10 = da[1][0], a11 = da[1][1], a20 = da[2][0], a21 = da[2][1], a30 = da[3][0],
                 a31 = da[3][1];
  const uint32_t b00 = db[0][0], b01 = db[0][1], b10 = db[1][0], b11 = db[1][1], b20 = db[2][0], b21 = db[2][1], b30 = db[3][0],
                 b31 = db[3][1];
#define MV_T16(A, LO, HI, OFF)                                                                            \
  "ds_write_b16 " A ", " LO " offset:" #OFF "+0\n\tds_write_b16_d16_hi " A ", " LO " offset:" #OFF "+64\n\t" \
  "ds_write_b16 " A ", " HI " offset:" #OFF "+128\n\tds_write_b16_d16_hi " A ", " HI " offset:" #OFF "+192\n\t"
  asm volatile(
      MV_T16("%4", "%6", "%7", 0) MV_T16("%5", "%8", "%9", 512) MV_T16("%4", "%10", "%11", 1024) MV_T16("%5", "%12", "%13", 1536)
      "ds_read_b128 %0, %22\n\tds_read_b128 %1, %22 offset:1024\n\t"
      MV_T16("%4", "%14", "%15", 0) MV_T16("%5", "%16", "%17", 512) MV_T16("%4", "%18", "%19", 1024) MV_T16("%5", "%20", "%21", 1536)
      "ds_read_b128 %2, %22\n\tds_read_b128 %3, %22 offset:1024\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(a0), "v"(a1), "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(a20), "v"(a21), "v"(a30), "v"(a31), "v"(b00), "v"(b01),
        "v"(b10), "v"(b11), "v"(b20), "v"(b21), "v"(b30), "v"(b31), "v"(r)
      : "memory");
#undef MV_T16
#endif
}
// The reverse direction for two fp16 planes of one 32 x 32 fragment: coalesced 16-byte pieces (rows lane >> 2 and + 16, chunk
// lane & 3) are written into the image, the C/D-layout units (row lane & 31, columns 8 g + 4 hi .. + 3) are read back.
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
// Four fp32 rounds (32 rows x 16 columns each) per statement: writes at (wa, wb), reads at (ra, rb).
__device__ __forceinline__ void scr_f32x4(uint32_t wa, uint32_t wb, const u32x4 (&d)[8], uint32_t ra, uint32_t rb,
                                          u32x4 (&o)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_write_b128 %8, %10\n\tds_write_b128 %9, %11\n\tds_read_b128 %0, %18\n\tds_read_b128 %1, %19\n\t"
      "ds_write_b128 %8, %12\n\tds_write_b128 %9, %13\n\tds_read_b128 %2, %18\n\tds_read_b128 %3, %19\n\t"
      "ds_write_b128 %8, %14\n\tds_write_b128 %9, %15\n\tds_read_b128 %4, %18\n\tds_read_b128 %5, %19\n\t"
      "ds_write_b128 %8, %16\n\tds_write_b128 %9, %17\n\tds_read_b128 %6, %18\n\tds_read_b128 %7, %19\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(wa), "v"(wb), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(ra),
        "v"(rb)
      : "memory");
#endif
}

// PP_RESLN accumulator init: the 4 float4 (columns 8 g + 4 hi .. + 3, g = 0..3, of one 32-column block) of the
// bias, gamma and beta images (gamma at +3072 B, beta at +6144 B of the bias image) in one statement.
__device__ __forceinline__ void lds_read_bgb(uint32_t addr, float4 (&bi)[4], float4 (&ga)[4], float4 (&be)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:32\n\tds_read_b128 %2, %12 offset:64\n\t"
      "ds_read_b128 %3, %12 offset:96\n\tds_read_b128 %4, %12 offset:3072\n\tds_read_b128 %5, %12 offset:3104\n\t"
      "ds_read_b128 %6, %12 offset:3136\n\tds_read_b128 %7, %12 offset:3168\n\tds_read_b128 %8, %12 offset:6144\n\t"
      "ds_read_b128 %9, %12 offset:6176\n\tds_read_b128 %10, %12 offset:6208\n\tds_read_b128 %11, %12 offset:6240\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(bi[0]), "=&v"(bi[1]), "=&v"(bi[2]), "=&v"(bi[3]), "=&v"(ga[0]), "=&v"(ga[1]), "=&v"(ga[2]), "=&v"(ga[3]),
        "=&v"(be[0]), "=&v"(be[1]), "=&v"(be[2]), "=&v"(be[3])
      : "v"(addr)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// One float4 of each of the three images (register-lean form of lds_read_bgb for PP_RESLN3).
__device__ __forceinline__ void lds_read_bgb1(uint32_t addr, float4& bi, float4& ga, float4& be) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:3072\n\tds_read_b128 %2, %3 offset:6144\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(bi), "=&v"(ga), "=&v"(be)
               : "v"(addr)
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// SCHED 0: two barriers per phase, the M-halves one barrier apart (DIST = issue distance in phases, 2..6).
// SCHED 1: ONE barrier per phase; the first M-half runs [MFMA(j), read fragments(j+1)] and the second
//          [read fragments(j), MFMA(j)] inside the same barrier interval, so each SIMD's matrix pipe is handed from
//          one wave to the other in the middle of the interval without a barrier in between (DIST = F, the number
//          of half-tiles kept in flight across each barrier, 2..4: half-tile H is issued in interval H-3-F, is
//          landed for every wave at the barrier that ends interval H-3, and its LDS region was last read in
//          interval H-8 or H-9).
// COAL 1: epilogue stores (and the residual loads) go through the wave-private LDS transposition above.
// X2 1: split-operand mode (GemmArgs::nseg == 3): three K sweeps A_hi W_hi + A_lo W_hi + A_hi W_lo, and PP_GELU also writes
//       the lo plane of its output.  A separate instantiation: the plain kernels keep their register allocation.
template <int EPI, int DIST, int ABL, int SCHED = 0, int COAL = 0, int RAW = 0, int X2 = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs a) {
  static_assert(!RAW || (COAL == 1 && (EPI == PP_QK || EPI == PP_GELU)), "RAW: the fp16-output kernels of the transposed path, token row per lane");
  static_assert(SCHED == 0 ? (DIST >= 2 && DIST <= 6) : (DIST >= 2 && DIST <= 4), "half-tile issue distance");
  constexpr bool SWAP = (EPI != PP_VT);
  constexpr bool IS_RESLN = (EPI == PP_RESLN || EPI == PP_RESLN2 || EPI == PP_RESLN3);
  constexpr bool HILO = (EPI == PP_RESLN3);        // raw stream as two fp16 planes
  constexpr bool EMITS = (EPI == PP_RESLN2 || EPI == PP_RESLN3);  // partial row sums + fp16 operand for the RAW consumers
  constexpr bool IS_RES = (EPI == PP_RES || IS_RESLN);
  static_assert(!IS_RESLN || COAL == 1, "the LayerNorm-fused residual init is written for the transposed (COAL) path");
  constexpr int WAITN = SCHED == 0 ? 2 * (DIST - 1) : 2 * DIST;
  constexpr int LDS_SCR = RAW ? PP_LDS_SCR_RAW : PP_LDS_SCR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int K = a.K, nk0 = K >> 6;                 // K-tiles of one sweep
  constexpr int nseg = X2 ? 3 : 1;                  // split-operand mode: three sweeps (GemmArgs::nseg)
  const int nk = nk0 * nseg;                        // K-tiles per output tile
  const int tm_count = a.M >> 8, tn_count = a.N >> 8;
  const int ntiles = tm_count * tn_count;
  const int G = gridDim.x;
  const int bslot = xcd_remap(blockIdx.x, G);
  const unsigned long long clk0 = a.clk ? __builtin_amdgcn_s_memtime() : 0ull;

  // ---- optional start-up stagger: all workgroups of a launch otherwise run their tiles in lockstep, so their
  // accumulator-init loads and epilogue stores hit HBM in bursts (every CU at once) with the matrix pipes idle, and
  // HBM idles during the main loops.  (Waves 1..7 wait for wave 0 at the prologue barrier.)
  // stagger < 0: TWO phase groups instead of a random spread — odd slots start -stagger x 8128 cycles late, so half of
  // every XCD's workgroups sit in their main loops while the other half runs its epilogue / accumulator init (each phase
  // group keeps sharing its operand panels in L2 at the same moment, which the random spread destroyed).
  if (a.stagger != 0 && wave == 0) {
    const int n = a.stagger > 0 ? (int)((uint32_t)(bslot * 2654435761u) >> 16) % (a.stagger + 1) : ((bslot & 1) ? -a.stagger : 0);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- bias -> LDS (once per workgroup; RAW kernels: per tile, issue_stats)
  if constexpr (!RAW) {
    float* lb = (float*)(smem + PP_LDS_BIAS);
    for (int n = tid; n < a.N; n += 512) lb[n] = a.bias ? a.bias[n] : 0.f;
    if constexpr (IS_RESLN) {  // N == 768: gamma at [768, 1536), beta at [1536, 2304) of the same image
      for (int n = tid; n < MV_HIDDEN; n += 512) {
        lb[MV_HIDDEN + n] = a.lng[n];
        lb[2 * MV_HIDDEN + n] = a.lnb[n];
      }
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
  size_t tA = 0, tW = 0;  // X2: byte offsets of the issue tile's operand panels
  auto set_issue_seg = [&]() {  // X2, sweep 0: A_hi W_hi, 1: A_lo W_hi, 2: A_hi W_lo
    iA = (const char*)(i_seg == 1 ? a.A2 : a.A) + tA;
    iW = (const char*)(i_seg == 2 ? a.W2 : a.W) + tW;
  };
  auto set_issue_tile = [&](int it) {
    const int L = it * G + bslot;
    if (L < ntiles) {  // past the end: keep staging the last valid tile (never read, keeps the vmcnt ledger exact)
      int tm, tn;
      raster(L, tm_count, tn_count, a.GN, tm, tn);
      if constexpr (X2) {
        tA = (size_t)tm * 256 * K * 2;
        tW = (size_t)tn * 256 * K * 2;
      } else {
        iA = (const char*)a.A + (size_t)tm * 256 * K * 2;
        iW = (const char*)a.W + (size_t)tn * 256 * K * 2;
      }
    }
    if constexpr (X2) set_issue_seg();
  };
  set_issue_tile(0);
  // kind: 0 = a0, 1 = a1, 2 = b0, 3 = b1; issue order per K-tile: b0, a0, b1, a1 (= read order)
  auto issue = [&](auto kindc, auto parc) {
    constexpr int kind = decltype(kindc)::value;
    constexpr int par = decltype(parc)::value;
    if constexpr (!(ABL & PP_ABL_NODMA)) {
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
    }
    if constexpr (kind == 1) {  // last half-tile of this K-tile: advance the cursor
      if (++i_kt == nk0) {
        i_kt = 0;
        if constexpr (X2) {
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

  // ---- fragment read addresses
  const int swz = (lane >> 1) & 7;
  int rdA[4], rdB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int o = l31 * 128 + (((ks * 2 + hi) ^ swz) << 4);
    rdA[ks] = PP_LDS_A + wr * 8192 + o;
    rdB[ks] = PP_LDS_B + wc * 4096 + o;
  }
  half8_t Xf[2][4], Wx[4], Wy[4];
  auto read_a = [&](int par, int asub) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        Xf[ii][ks] = *(const half8_t*)(smem + rdA[ks] + par * 32768 + asub * 16384 + ii * 4096);
  };
  auto read_b = [&](half8_t (&Wf)[4], int par, int bsub) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) Wf[ks] = *(const half8_t*)(smem + rdB[ks] + par * 32768 + bsub * 16384);
  };
  if constexpr (ABL & PP_ABL_NOREAD) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Xf[0][ks] = *(const half8_t*)(a.A + (size_t)l31 * K + ks * 16 + hi * 8);
      Xf[1][ks] = *(const half8_t*)(a.A + (size_t)(32 + l31) * K + ks * 16 + hi * 8);
      Wx[ks] = *(const half8_t*)(a.W + (size_t)l31 * K + ks * 16 + hi * 8);
      Wy[ks] = *(const half8_t*)(a.W + (size_t)(32 + l31) * K + ks * 16 + hi * 8);
    }
  }

  floatx16 acc[4][2];
  // PP_RESLN3: the two fp16 planes of fragment pair i of the residual tile at (mw0, nw0) by full-line loads (16 rows x 64 B per
  // instruction), parked in the accumulator registers they will be transposed into: [i][j] registers 0-3 / 4-7 = hi rows
  // crow / crow + 16, 8-11 / 12-15 = lo
  auto park_residual = [&](int i, int mw0, int nw0) {
    const int crow = lane >> 2, cchunk = lane & 3;
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
  auto mma_quadrant = [&](auto asubc, auto bc, const half8_t (&Wf)[4]) {
    constexpr int asub = decltype(asubc)::value;
    constexpr int b = decltype(bc)::value;
    if constexpr (ABL & PP_ABL_NOMFMA) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        keep_live(Wf[ks]);
        keep_live(Xf[0][ks]);
        keep_live(Xf[1][ks]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          if constexpr (SWAP)
            acc[asub * 2 + ii][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[ks], Xf[ii][ks], acc[asub * 2 + ii][b], 0, 0, 0);
          else
            acc[asub * 2 + ii][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Xf[ii][ks], Wf[ks], acc[asub * 2 + ii][b], 0, 0, 0);
        }
    }
  };

  // One phase.  s = phase index inside the 8-phase (two K-tile) loop body; P = s & 3, par = s >> 2.
  // Even K-tile: b0 in Wx, b1 -> Wy, next b0 -> Wy.  Odd K-tile: b0 in Wy, b1 -> Wx, next b0 -> Wx.
  auto phase = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int P = s & 3, par = s >> 2;
    // ---- read section
    if constexpr (!(ABL & PP_ABL_NOREAD)) {
      if constexpr (P == 0) read_a(par, 0);
      if constexpr (P == 1) { if constexpr (par == 0) read_b(Wy, par, 1); else read_b(Wx, par, 1); }
      if constexpr (P == 2) read_a(par, 1);
      if constexpr (P == 3) { if constexpr (par == 0) read_b(Wy, par ^ 1, 0); else read_b(Wx, par ^ 1, 0); }
    }
    issue_psi(std::integral_constant<int, (s + 1 + DIST) & 7>{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix section
    if constexpr (!(ABL & PP_ABL_NOPRIO)) __builtin_amdgcn_s_setprio(1);
    if constexpr (P == 0) { if constexpr (par == 0) mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, Wx);
                            else mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, Wy); }
    if constexpr (P == 1) { if constexpr (par == 0) mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, Wy);
                            else mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, Wx); }
    if constexpr (P == 2) { if constexpr (par == 0) mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, Wy);
                            else mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, Wx); }
    if constexpr (P == 3) { if constexpr (par == 0) mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, Wx);
                            else mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, Wy); }
    if constexpr (!(ABL & PP_ABL_NOPRIO)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- SCHED 1 building blocks: fragment set / quadrant of phase s (P = s & 3, LDS parity = s >> 2)
  auto read_set = [&](auto sc) {
    constexpr int s = decltype(sc)::value & 7;
    constexpr int P = s & 3, par = s >> 2;
    if constexpr (!(ABL & PP_ABL_NOREAD)) {
      if constexpr (P == 0) { read_a(par, 0); read_b(Wx, par, 0); }
      if constexpr (P == 1) read_b(Wy, par, 1);
      if constexpr (P == 2) read_a(par, 1);
    }
  };
  auto mma_set = [&](auto sc) {
    constexpr int P = decltype(sc)::value & 3;
    if constexpr (P == 0) mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, Wx);
    if constexpr (P == 1) mma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, Wy);
    if constexpr (P == 2) mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, Wy);
    if constexpr (P == 3) mma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, Wx);
  };
  auto interval = [&](auto sc, auto grpc, bool last_of_tile = false) {
    constexpr int s = decltype(sc)::value;
    constexpr int grp = decltype(grpc)::value;
    if constexpr (grp == 0) {  // matrix pipe first, then the NEXT phase's fragments
      if constexpr (!(ABL & PP_ABL_NOPRIO)) __builtin_amdgcn_s_setprio(1);
      mma_set(sc);
      if constexpr (!(ABL & PP_ABL_NOPRIO)) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // the next OUTPUT tile's first fragments are read after its accumulator init instead (run_tiles): keeping
      // 48 fragment VGPRs live across the epilogue + init costs more than one exposed LDS read per tile
      if (!(s == 7 && last_of_tile)) read_set(std::integral_constant<int, (s + 1) & 7>{});
      issue_psi(std::integral_constant<int, (s + 3 + DIST) & 7>{});
    } else {  // this phase's fragments first, then the matrix pipe as the other half releases it
      read_set(sc);
      issue_psi(std::integral_constant<int, (s + 3 + DIST) & 7>{});
      __builtin_amdgcn_sched_barrier(0);
      mma_set(sc);
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
    if (bslot < ntiles) raster(bslot, tm_count, tn_count, a.GN, tm, tn);
    issue_stats(0, tm, tn);
  }

  // ---- prologue.  SCHED 0: half-tiles psi = 0 .. DIST in flight, psi 0 (b0 of K-tile 0) and 1 (a0) landed.
  //               SCHED 1: psi = 0 .. 2 + F in flight, psi 0 .. 2 landed.
  constexpr int NPRO = SCHED == 0 ? DIST + 1 : DIST + 3;
  issue_psi(std::integral_constant<int, 0>{});
  issue_psi(std::integral_constant<int, 1>{});
  issue_psi(std::integral_constant<int, 2>{});
  if constexpr (NPRO > 3) issue_psi(std::integral_constant<int, 3>{});
  if constexpr (NPRO > 4) issue_psi(std::integral_constant<int, 4>{});
  if constexpr (NPRO > 5) issue_psi(std::integral_constant<int, 5>{});
  if constexpr (NPRO > 6) issue_psi(std::integral_constant<int, 6>{});
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");  // lgkmcnt: the bias image writes
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  constexpr bool STAGGER = !(ABL & PP_ABL_NOSTAGGER);
  if constexpr (SCHED == 0) {
    if constexpr (!(ABL & PP_ABL_NOREAD)) read_b(Wx, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAGGER) {
      if (wr == 1) __builtin_amdgcn_s_barrier();  // the second M-half runs one barrier behind the first
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  unsigned long long c_init = 0ull, c_main = 0ull, c_epi = 0ull;
  auto run_tiles = [&](auto grpc) {
  for (int it = 0;; ++it) {
    const int L = it * G + bslot;
    if (L >= ntiles) break;
    unsigned long long tA = 0ull, tB = 0ull, tC = 0ull;
    if constexpr (ABL & PP_ABL_CLK) tA = __builtin_amdgcn_s_memtime();
    int tile_m, tile_n;
    raster(L, tm_count, tn_count, a.GN, tile_m, tile_n);
    int next_m = 0, next_n = 0;  // RAW: the workgroup's next tile (issue_stats); PP_RESLN3: its residual tile is requested
    const bool has_next = L + G < ntiles;  // during this tile's epilogue
    if constexpr (RAW || HILO) {
      if (has_next) raster(L + G, tm_count, tn_count, a.GN, next_m, next_n);
    }
    (void)next_m; (void)next_n; (void)has_next;
    const int mw = (tile_m << 8) + wr * 128;  // first token row of this wave
    const int nw = (tile_n << 8) + wc * 64;   // first output column of this wave

    // ---- accumulator init: bias (+ residual).  The bias image is read with inline-asm ds_reads: hipcc would
    // put `s_waitcnt vmcnt(0)` in front of a compiler-visible LDS load here (LDS-DMA in flight) and drain the
    // operand pipeline once per output tile.  Loads and their lgkmcnt wait are one statement (guide §5.7 form i).
    if constexpr (SWAP) {
      float4 bv[2][4];
      const uint32_t baddr = (uint32_t)(PP_LDS_BIAS + (nw + 4 * hi) * 4);
      if constexpr (!RAW)  // (RAW: accumulators start from zero; bias' is applied in the epilogue from the per-tile image)
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\t"
          "ds_read_b128 %3, %8 offset:96\n\tds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\t"
          "ds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(bv[0][0]), "=&v"(bv[0][1]), "=&v"(bv[0][2]), "=&v"(bv[0][3]), "=&v"(bv[1][0]), "=&v"(bv[1][1]),
            "=&v"(bv[1][2]), "=&v"(bv[1][3])
          : "v"(baddr)
          : "memory");
      // scratch addresses of this lane: MFMA layout (row = lane & 31) and coalesced layout (row = lane >> 2)
      const uint32_t scr = (uint32_t)(LDS_SCR + wave * 2048);
      const uint32_t sf = (uint32_t)((l31 >> 2) & 3);
      const uint32_t scr_m32 = scr + l31 * 64 + (((uint32_t)hi ^ sf) << 4);          // fp32 rounds: chunk 2 gg + hi -> ^ (gg * 32)
      const uint32_t scr_c = scr + (lane >> 2) * 64 + ((((uint32_t)lane & 3) ^ (((uint32_t)lane >> 4) & 3)) << 4);
      if constexpr (IS_RES && COAL) {
        // residual tile by full-line loads (16 rows x 64 B per instruction), transposed into the C/D layout
        // (the raw lines are parked in the accumulator registers they will be transposed into)
        float2 lnst[4];  // PP_RESLN: (mean, rstd) of this lane's four token rows
        float2 lnp[4][3];  // virtual LayerNorm: the rows' vstats, loaded here and turned into (mean, rstd) only AFTER the
                           // residual tile's loads are issued (in source order hipcc waits for these loads first and the
                           // two memory latencies add up: +1.3 us per tile)
        if constexpr (IS_RESLN) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (EMITS) {
              const float2* pp = (const float2*)(a.lnstats + 6 * (size_t)(mw + i * 32 + l31));
              lnp[i][0] = pp[0]; lnp[i][1] = pp[1]; lnp[i][2] = pp[2];
            } else {
              lnst[i] = *(const float2*)(a.lnstats + 2 * (size_t)(mw + i * 32 + l31));
            }
          }
        }
        auto finish_stats = [&]() {
          if constexpr (EMITS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lnst[i] = ln_from_partials(lnp[i][0], lnp[i][1], lnp[i][2], a.ln_eps);
          }
        };
        (void)lnp;
        if constexpr (HILO) {
          // the two fp16 planes of the raw stream by full-line loads (16 rows x 64 B per instruction), parked in the
          // accumulator registers: [i][j] registers 0-3 / 4-7 = hi rows crow / crow + 16, 8-11 / 12-15 = lo
          const uint32_t wbase = scr + l31 * 64 + hi * 8 + (sf << 4);
          // (tiles after the workgroup's first: the lines were requested fragment pair by fragment pair during the PREVIOUS
          //  tile's epilogue, each as soon as its accumulator registers had been stored — park_residual below — so the read
          //  phase of this tile runs under the write phase of the last one instead of after it)
          if (it == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) park_residual(i, mw, nw);
          }
          finish_stats();
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
              for (int g = 0; g < 4; ++g) {
                float4 bi, ga, be;
                lds_read_bgb1(baddr + j * 128 + g * 32, bi, ga, be);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const uint32_t wh = oh[g][e >> 1], wl = ol[g][e >> 1];  // scalar copies before the bit casts (see f2u)
                  const half2_t h2 = __builtin_bit_cast(half2_t, wh);
                  const half2_t l2 = __builtin_bit_cast(half2_t, wl);
                  const float r = (float)h2[e & 1] + (float)l2[e & 1];
                  const float t = (r - lnst[i].x) * lnst[i].y;
                  acc[i][j][4 * g + e] = __builtin_fmaf(t, ((const float*)&ga)[e], ((const float*)&be)[e]) + ((const float*)&bi)[e];
                }
              }
            }
          }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int x = 0; x < 2; ++x) {
                if constexpr (ABL & PP_ABL_B34) { if (j == 1 && x == 1) continue; }
                const float4 t = *(const float4*)(a.xres + (size_t)(mw + i * 32 + x * 16 + (lane >> 2)) * MV_HIDDEN + nw + j * 32 + h * 16 + 4 * (lane & 3));
                acc[i][j][4 * (2 * h + x) + 0] = t.x; acc[i][j][4 * (2 * h + x) + 1] = t.y;
                acc[i][j][4 * (2 * h + x) + 2] = t.z; acc[i][j][4 * (2 * h + x) + 3] = t.w;
              }
        finish_stats();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 d[8], o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)  // q = (j, h, x): registers 4 q .. 4 q + 3 of fragment pair i
#pragma unroll
            for (int e = 0; e < 4; ++e) d[q][e] = f2u(acc[i][q >> 2][4 * (q & 3) + e]);
          scr_f32x4(scr_c, scr_c + 1024, d, scr_m32, scr_m32 ^ 32u, o);
          if constexpr (IS_RESLN) {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              float4 bi[4], ga[4], be[4];
              lds_read_bgb(baddr + j * 128, bi, ga, be);
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float t = (u2f(o[4 * j + g][e]) - lnst[i].x) * lnst[i].y;
                  acc[i][j][4 * g + e] =
                      __builtin_fmaf(t, ((const float*)&ga[g])[e], ((const float*)&be[g])[e]) + ((const float*)&bi[g])[e];
                }
            }
          } else {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][q >> 2][4 * (q & 3) + e] = u2f(o[q][e]) + ((const float*)&bv[q >> 2][q & 3])[e];
          }
        }
        }  // !HILO
      } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (RAW) {
              acc[i][j][4 * g + 0] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
            } else if constexpr (IS_RES) {
              const float4 xv = *(const float4*)(a.xres + (size_t)(mw + i * 32 + l31) * MV_HIDDEN + nw + j * 32 + 8 * g + 4 * hi);
              acc[i][j][4 * g + 0] = xv.x + bv[j][g].x; acc[i][j][4 * g + 1] = xv.y + bv[j][g].y;
              acc[i][j][4 * g + 2] = xv.z + bv[j][g].z; acc[i][j][4 * g + 3] = xv.w + bv[j][g].w;
            } else {
              acc[i][j][4 * g + 0] = bv[j][g].x; acc[i][j][4 * g + 1] = bv[j][g].y;
              acc[i][j][4 * g + 2] = bv[j][g].z; acc[i][j][4 * g + 3] = bv[j][g].w;
            }
          }
      }
      (void)scr_m32; (void)scr_c;
    } else {  // lane = output column
      float b0, b1;
      const uint32_t baddr = (uint32_t)(PP_LDS_BIAS + (nw + l31) * 4);
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(b0), "=&v"(b1)
                   : "v"(baddr)
                   : "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][0][r] = RAW ? 0.f : b0;
          acc[i][1][r] = RAW ? 0.f : b1;
        }
    }

    if constexpr (ABL & PP_ABL_CLK) tB = __builtin_amdgcn_s_memtime();
    if constexpr (SCHED == 1 && decltype(grpc)::value == 0) read_set(std::integral_constant<int, 0>{});
    for (int kt = 0; kt < nk; kt += 2) {
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
      if constexpr (SCHED == 0) {
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        phase(std::integral_constant<int, 3>{});
        phase(std::integral_constant<int, 4>{});
        phase(std::integral_constant<int, 5>{});
        phase(std::integral_constant<int, 6>{});
        phase(std::integral_constant<int, 7>{});
      } else {
        interval(std::integral_constant<int, 0>{}, grpc);
        interval(std::integral_constant<int, 1>{}, grpc);
        interval(std::integral_constant<int, 2>{}, grpc);
        interval(std::integral_constant<int, 3>{}, grpc);
        interval(std::integral_constant<int, 4>{}, grpc);
        interval(std::integral_constant<int, 5>{}, grpc);
        interval(std::integral_constant<int, 6>{}, grpc);
        interval(std::integral_constant<int, 7>{}, grpc, kt + 2 >= nk);
      }
    }

    if constexpr (ABL & PP_ABL_CLK) tC = __builtin_amdgcn_s_memtime();
    // ---- epilogue (store only)
    if constexpr (ABL & PP_ABL_NOEPI) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) keep_live(acc[i][j]);
    } else if constexpr (COAL) {
      // ---- coalesced epilogue: every 32x32 fragment goes through the wave's [32][64 B] LDS image
      const uint32_t scr = (uint32_t)(LDS_SCR + wave * 2048);
      const uint32_t sf = (uint32_t)((l31 >> 2) & 3);
      const uint32_t scr_c = scr + (lane >> 2) * 64 + ((((uint32_t)lane & 3) ^ (((uint32_t)lane >> 4) & 3)) << 4);
      const int crow = lane >> 2, cchunk = lane & 3;  // coalesced layout: row (+16 for the second read), 16-B chunk
      if constexpr (EMITS) {
        // vstats of the new raw rows: (sum, sum of squares) over this TILE's 256 columns.  Each wave reduces its 64 columns
        // (lane = token row, the two half-waves hold disjoint column sets), parks the 128 pairs in its own scratch, and after
        // a workgroup barrier the first column wave of each M-half adds the four shares in wave order (deterministic) and
        // writes slot tile_n of the rows' three pairs.  A second barrier keeps the scratch intact until it has been read.
        // (inline asm: compiler-visible LDS accesses would wait for the LDS-DMA in flight, see the accumulator init)
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
      if constexpr ((EPI == PP_F32 || IS_RES) && !HILO) {
        const uint32_t scr_m32 = scr + l31 * 64 + (((uint32_t)hi ^ sf) << 4);
        float* obase = (IS_RES ? a.xres : a.outf) + (size_t)(mw + crow) * a.N + nw + 4 * cchunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 d[8], o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)  // q = (j, h, gg): C/D registers 4 q .. 4 q + 3 of fragment pair i
#pragma unroll
            for (int e = 0; e < 4; ++e) d[q][e] = f2u(acc[i][q >> 2][4 * (q & 3) + e]);
          scr_f32x4(scr_m32, scr_m32 ^ 32u, d, scr_c, scr_c + 1024, o);
#pragma unroll
          for (int q = 0; q < 8; ++q) {  // q = (j, h, x): rows 16 x + crow, columns 32 j + 16 h + 4 cchunk
            float* op = obase + (size_t)(i * 32 + (q & 1) * 16) * a.N + (q >> 2) * 32 + ((q >> 1) & 1) * 16;
            if constexpr (ABL & PP_ABL_B34) { if ((q >> 2) == 1 && (q & 1) == 1) continue; }
            *(u32x4*)op = o[q];
            if constexpr (EPI == PP_RESLN2) {
              if (a.raw) {  // fp16 copy straight from the transposed fp32 image: 8 bytes per lane, 32-byte row pieces
                half_t* hp = a.out16 + (size_t)(mw + crow + i * 32 + (q & 1) * 16) * a.N + nw + 4 * cchunk + (q >> 2) * 32 + ((q >> 1) & 1) * 16;
                u32x2 hv;
                hv[0] = pack_h2(u2f(o[q][0]), u2f(o[q][1]));
                hv[1] = pack_h2(u2f(o[q][2]), u2f(o[q][3]));
                *(u32x2*)hp = hv;
              }
            }
          }
        }
      }
      if constexpr (!(EPI == PP_F32 || IS_RES) || EMITS)
      if (EPI != PP_RESLN2 || !a.raw) {
        // fp16 outputs: 8-B units (4 values) of the C/D layout -> chunk g, half hi of the row
        // (PP_RESLN2: the fp16 copy of the raw stream, the A operand of the next RAW consumer)
        const uint32_t wbase = scr + l31 * 64 + hi * 8 + (sf << 4);
        half_t* obase;    // pointer of (row crow, chunk cchunk) of fragment (i = 0, j = 0)
        size_t rstride;   // elements between image rows in the output
        size_t istride;   // elements between i blocks (32 C/D rows along the register axis or the lane axis)
        size_t jstride;   // elements between j blocks
        bool live = true;
        bool vtile = false;  // PP_QK: this tile belongs to the V block and is stored as V^T
        const uint32_t tbase = scr + hi * 256 + (((uint32_t)(l31 >> 3) ^ (uint32_t)hi) << 4) + (uint32_t)(l31 & 7) * 2;
        if constexpr (EPI == PP_F16 || EPI == PP_GELU || EMITS) {
          obase = a.out16 + (size_t)(mw + crow) * a.N + nw + 8 * cchunk;
          rstride = a.N; istride = (size_t)32 * a.N; jstride = 32;
        } else if constexpr (EPI == PP_QK) {  // one head per wave; S % 64 == 0 so a 128-row block may span two batch rows
          // columns [0,768) -> Q, [768,1536) -> K, [1536,2304) -> V^T (a merged launch); col0 = 768: K (and V) only
          const int colg = nw + a.col0;
          const int which = colg >= 2 * MV_HIDDEN ? 2 : (colg >= MV_HIDDEN ? 1 : 0);
          const int head = (colg - which * MV_HIDDEN) >> 6;
          vtile = which == 2;
          if (vtile) {  // image rows = head dims, image columns = tokens (scr_f16x2_t)
            obase = a.vt + ((size_t)head * MV_HEAD_DIM + crow) * a.S + 8 * cchunk;
            rstride = a.S; istride = 0; jstride = (size_t)32 * a.S;
          } else {
            obase = (which ? a.k : a.q) + (size_t)head * a.S * MV_HEAD_DIM + (size_t)crow * MV_HEAD_DIM + 8 * cchunk;
            rstride = MV_HEAD_DIM; istride = 0; jstride = 32;  // the batch-row part is added per i below
          }
        } else {  // PP_VT: image rows = head dims, image columns = tokens
          const int head = nw >> 6;
          obase = a.vt + ((size_t)head * MV_HEAD_DIM + crow) * a.S + 8 * cchunk;
          rstride = a.S; istride = 0; jstride = (size_t)32 * a.S;
        }
        // RAW: bias' of this wave's columns (per-tile LDS image) and rstd of its token rows (from the vstats image, one row
        // per lane and i), applied as fma(rstd, acc, bias')
        float4 rbv[2][4];
        float rrs[4];
        if constexpr (RAW) {
          const uint32_t baddr = (uint32_t)(PP_LDS_BIAS_T + (it & 1) * 1024 + (wc * 64 + 4 * hi) * 4);
          const uint32_t saddr = (uint32_t)(PP_LDS_RSTD + (it & 1) * 1024 + (wr * 128 + l31) * 4);  // this wave's 128 rows
          asm volatile(
              "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:32\n\tds_read_b128 %2, %12 offset:64\n\t"
              "ds_read_b128 %3, %12 offset:96\n\tds_read_b128 %4, %12 offset:128\n\tds_read_b128 %5, %12 offset:160\n\t"
              "ds_read_b128 %6, %12 offset:192\n\tds_read_b128 %7, %12 offset:224\n\t"
              "ds_read_b32 %8, %13\n\tds_read_b32 %9, %13 offset:128\n\tds_read_b32 %10, %13 offset:256\n\t"
              "ds_read_b32 %11, %13 offset:384\n\ts_waitcnt lgkmcnt(0)"
              : "=&v"(rbv[0][0]), "=&v"(rbv[0][1]), "=&v"(rbv[0][2]), "=&v"(rbv[0][3]), "=&v"(rbv[1][0]), "=&v"(rbv[1][1]),
                "=&v"(rbv[1][2]), "=&v"(rbv[1][3]), "=&v"(rrs[0]), "=&v"(rrs[1]), "=&v"(rrs[2]), "=&v"(rrs[3])
              : "v"(baddr), "v"(saddr)
              : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        (void)rbv; (void)rrs;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mb = mw + i * 32;
          half_t* ob = obase + i * istride;
          const float rrs_i = RAW ? rrs[i] : 1.f;  // RAW: rstd of token row 32 i + l31
          (void)rrs_i;
          if constexpr (EPI == PP_QK || EPI == PP_VT) {
            live = mb < a.Mreal;
            const int b = mb / a.S, s0 = mb - b * a.S;
            if (EPI == PP_QK && !vtile) ob = obase + ((size_t)b * MV_HEADS * a.S + s0) * MV_HEAD_DIM;
            else ob = obase + (size_t)b * MV_HEADS * MV_HEAD_DIM * a.S + s0;
          }
          u32x2 d[2][4];
          u32x4 o[4];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v0 = acc[i][j][4 * g + 0], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
              if constexpr (RAW) {  // lane = token row 32 i + l31, registers = 4 consecutive columns
                v0 = __builtin_fmaf(rrs_i, v0, rbv[j][g].x); v1 = __builtin_fmaf(rrs_i, v1, rbv[j][g].y);
                v2 = __builtin_fmaf(rrs_i, v2, rbv[j][g].z); v3 = __builtin_fmaf(rrs_i, v3, rbv[j][g].w);
              }
              if constexpr (EPI == PP_GELU) {
                float2_t a01, a23;
                a01.x = v0; a01.y = v1; a23.x = v2; a23.y = v3;
                a01 = gelu_erf2(a01);
                a23 = gelu_erf2(a23);
                v0 = a01.x; v1 = a01.y; v2 = a23.x; v3 = a23.y;
                if constexpr (X2) {  // split-operand mode: the lo plane below is taken from the activated values
                  acc[i][j][4 * g + 0] = v0; acc[i][j][4 * g + 1] = v1; acc[i][j][4 * g + 2] = v2; acc[i][j][4 * g + 3] = v3;
                }
              }
              d[j][g][0] = pack_h2(v0, v1);
              d[j][g][1] = pack_h2(v2, v3);
            }
          if (EPI == PP_QK && vtile) scr_f16x2_t(tbase, tbase ^ 32u, d[0], d[1], scr_c, o);
          else scr_f16x2(wbase, wbase ^ 16u, wbase ^ 32u, wbase ^ 48u, d[0], d[1], scr_c, o);
          if (live) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              half_t* op = ob + j * jstride;
              *(u32x4*)op = o[2 * j];
              *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
            }
          }
          if constexpr (HILO || (X2 && EPI == PP_GELU)) {  // second plane: lo = fp16(r - hi), same addresses in the lo buffer
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float v0 = acc[i][j][4 * g + 0], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                d[j][g][0] = pack_h2(v0 - (float)(half_t)v0, v1 - (float)(half_t)v1);
                d[j][g][1] = pack_h2(v2 - (float)(half_t)v2, v3 - (float)(half_t)v3);
              }
            scr_f16x2(wbase, wbase ^ 16u, wbase ^ 32u, wbase ^ 48u, d[0], d[1], scr_c, o);
            half_t* ol = ob + (a.out16b - a.out16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              half_t* op = ol + j * jstride;
              *(u32x4*)op = o[2 * j];
              *(u32x4*)(op + 16 * rstride) = o[2 * j + 1];
            }
            if constexpr (HILO) {  // fragment pair i is out: request pair i of the NEXT tile's residual into its registers
              if (has_next) park_residual(i, (next_m << 8) + wr * 128, (next_n << 8) + wc * 64);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mb = mw + i * 32;  // 32-row block: wave-uniform
        if constexpr (EPI == PP_F32 || IS_RES) {
          float* base = (IS_RES ? a.xres : a.outf) + (size_t)(mb + l31) * a.N + nw + 4 * hi;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float4 v;
              v.x = acc[i][j][4 * g + 0]; v.y = acc[i][j][4 * g + 1]; v.z = acc[i][j][4 * g + 2]; v.w = acc[i][j][4 * g + 3];
              *(float4*)(base + j * 32 + 8 * g) = v;
            }
        } else if constexpr (EPI == PP_F16 || EPI == PP_GELU) {
          half_t* rowptr = a.out16 + (size_t)(mb + l31) * a.N + nw;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (EPI == PP_GELU) store_frag_f16(acc[i][j], rowptr + j * 32, hi, [](float x) { return gelu_erf(x); });
            else store_frag_f16(acc[i][j], rowptr + j * 32, hi, [](float x) { return x; });
          }
        } else if constexpr (EPI == PP_QK) {  // N = 1536: columns [0,768) -> Q, [768,1536) -> K; 64 columns of a wave = one head
          if (mb < a.Mreal) {
            const int which = (nw + a.col0) >= MV_HIDDEN;
            const int head = (nw + a.col0 - which * MV_HIDDEN) >> 6;
            const int b = mb / a.S, s0 = mb - b * a.S;
            half_t* rowptr = (which ? a.k : a.q) + ((size_t)(b * MV_HEADS + head) * a.S + s0 + l31) * MV_HEAD_DIM;
#pragma unroll
            for (int j = 0; j < 2; ++j) store_frag_f16(acc[i][j], rowptr + j * 32, hi, [](float x) { return x; });
          }
        } else {  // PP_VT: N = 768 (the V block); lane = head dim, registers = consecutive tokens
          if (mb < a.Mreal) {
            const int head = nw >> 6;
            const int b = mb / a.S, s0 = mb - b * a.S;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              half_t* rowptr = a.vt + ((size_t)(b * MV_HEADS + head) * MV_HEAD_DIM + j * 32 + l31) * a.S + s0;
              store_frag_f16(acc[i][j], rowptr, hi, [](float x) { return x; });
            }
          }
        }
      }
    }
    if constexpr (ABL & PP_ABL_CLK) {
      c_init += tB - tA;
      c_main += tC - tB;
      c_epi += __builtin_amdgcn_s_memtime() - tC;
    }
  }

  };  // run_tiles
  if constexpr (SCHED == 1 && STAGGER) {
    if (wr == 0) run_tiles(std::integral_constant<int, 0>{});
    else run_tiles(std::integral_constant<int, 1>{});
  } else {
    run_tiles(std::integral_constant<int, 1>{});
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
  if constexpr (ABL & PP_ABL_CLK) {
    if (a.clk && (tid & 63) == 0) {
      unsigned long long* c = a.clk + (size_t)(blockIdx.x * 8 + wave) * 4;
      c[0] = __builtin_amdgcn_s_memtime() - clk0;
      c[1] = c_init;
      c[2] = c_main;
      c[3] = c_epi;
    }
  } else {
    if (a.clk && tid == 0) a.clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
  }
  if constexpr (SCHED == 0 && STAGGER) {
    if (wr == 0) __builtin_amdgcn_s_barrier();
  }
}
