// Round-4 probe (not product): what the MX-scaled matrix instruction and the fp16 -> fp6/bf6 pack conversions cost and do on gfx950.
//   (1) issue cycles per v_mfma_scale_f32_16x16x128_f8f6f4 by operand format pair (s_memtime, one or two waves per SIMD, every CU busy)
//   (2) cycles per v_cvt_scalef32_pk32_{fp6,bf6}_f16 and per byte-permute "bf8 from the top byte" group, beside nothing
//   (3) the conversions' semantics: scale direction, rounding, saturation (table dumped for the host to compare with a model)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_mx/mx_probe.hip -o tools/probe_mx/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half32 __attribute__((ext_vector_type(32)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int FA, int FB>
__global__ __launch_bounds__(512) void mfma_rate(const intx8* in, float* out, long long* ticks, int iters) {
  intx8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  floatx4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (floatx4){0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], FA, FB, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ __launch_bounds__(512) void mfma16_rate(const intx8* in, float* out, long long* ticks, int iters) {
  intx8 a8 = in[threadIdx.x & 63], b8 = in[64 + (threadIdx.x & 63)];
  half8 a = __builtin_bit_cast(half8, __builtin_shufflevector(a8, a8, 0, 1, 2, 3));
  half8 b = __builtin_bit_cast(half8, __builtin_shufflevector(b8, b8, 0, 1, 2, 3));
  floatx4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (floatx4){0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// mixed stream as the interleaved precise GEMM would issue it: per group 16 fp16 MFMAs + NQ scaled MFMAs of format F on the same accumulators
template <int F, int NQ>
__global__ __launch_bounds__(512) void mix_rate(const intx8* in, float* out, long long* ticks, int iters) {
  intx8 a8 = in[threadIdx.x & 63], b8 = in[64 + (threadIdx.x & 63)];
  half8 a = __builtin_bit_cast(half8, __builtin_shufflevector(a8, a8, 0, 1, 2, 3));
  half8 b = __builtin_bit_cast(half8, __builtin_shufflevector(b8, b8, 0, 1, 2, 3));
  floatx4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (floatx4){0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 7], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i & 7], F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>  // 0: pk32_fp6_f16, 1: pk32_bf6_f16, 2: top-byte bf8 of 32 halves with round-to-nearest (16 v_pk_add_u16 + 8 v_perm_b32), 3: pk_fp8_f16 x16
__global__ __launch_bounds__(256) void cvt_rate(const half32* in, unsigned* out, long long* ticks, int iters, float scale) {
  half32 v = in[threadIdx.x & 63];
  unsigned accum = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
      u6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
      accum += r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4] ^ r[5];
    } else if constexpr (KIND == 1) {
      u6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, scale);
      accum += r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4] ^ r[5];
    } else if constexpr (KIND == 2) {
      typedef unsigned u16v __attribute__((ext_vector_type(16)));
      u16v w = __builtin_bit_cast(u16v, v);
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        unsigned x = w[i], y = w[i + 1];
        asm volatile("v_pk_add_u16 %0, %0, %2\n\tv_pk_add_u16 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(0x00800080u));
        accum += __builtin_amdgcn_perm(y, x, 0x07050301u);
      }
    } else {
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      typedef short short2v __attribute__((ext_vector_type(2)));
      typedef unsigned u16v __attribute__((ext_vector_type(16)));
      u16v w = __builtin_bit_cast(u16v, v);
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        short2v r = {0, 0};
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(half2v, (unsigned)w[i]), scale, false);
        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(half2v, (unsigned)w[i + 1]), scale, true);
        accum += __builtin_bit_cast(unsigned, r);
      }
    }
    v[it & 31] += (_Float16)accum;  // keep the loop from being hoisted
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = accum;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void cvt_table(const half32* in, u6* o_fp6, u6* o_bf6, float scale) {
  half32 v = in[threadIdx.x];
  o_fp6[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
  o_bf6[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, scale);
}
// one MFMA with known operands: lane l supplies fp6 codes; checks K-slot <-> lane mapping and the scale operand (E8M0 byte per lane, opsel byte 0)
template <int F>
__global__ void mfma_sem(const intx8* a, const intx8* b, floatx4* c, const int* sa, const int* sb) {
  floatx4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, F, F, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}

static double fp6_e2m3(unsigned c) { int s = c >> 5, e = (c >> 3) & 3, m = c & 7; double v = e ? (1 + m / 8.0) * std::ldexp(1.0, e - 1) : m / 8.0; return s ? -v : v; }
static double bf6_e3m2(unsigned c) { int s = c >> 5, e = (c >> 2) & 7, m = c & 3; double v = e ? (1 + m / 4.0) * std::ldexp(1.0, e - 3) : m / 4.0 * 0.25; return s ? -v : v; }
static unsigned get6(const unsigned* w, int i) { int bit = 6 * i; unsigned long long x = w[bit >> 5] | ((unsigned long long)(bit / 32 + 1 < 6 ? w[bit / 32 + 1] : 0) << 32); return (x >> (bit & 31)) & 63; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  std::vector<int> hin(128 * 8);
  srand(7);
  for (auto& x : hin) x = (rand() & 0x3f3f3f3f) | 0x10101010;  // finite, mid-range in every 8- and 6-bit reading
  intx8* din; float* dout; long long* dt;
  CK(hipMalloc(&din, hin.size() * 4)); CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, 4 * 512 * 4096)); CK(hipMalloc(&dt, 8 * 8 * 4096));
  const int iters = 2000;
  auto report = [&](const char* name, int nblk, int nthr, double per_iter) {
    std::vector<long long> t(nblk * nthr / 64);
    hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : t) s += x;
    // readcyclecounter = s_memtime: 100 MHz constant clock on gfx9 -> report ns too via events
    printf("  %-44s ticks/iter-group %.2f (avg over %zu waves) -> per instr %.3f ticks\n", name, s / t.size() / iters, t.size(), s / t.size() / iters / per_iter);
  };
#define RUNK(NAME, KERN, NTHR, PER)                                                       \
  do {                                                                                     \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                           \
    hipLaunchKernelGGL(KERN, dim3(cus), dim3(NTHR), 0, 0, din, dout, dt, 10);              \
    hipDeviceSynchronize(); hipEventRecord(e0);                                            \
    hipLaunchKernelGGL(KERN, dim3(cus), dim3(NTHR), 0, 0, din, dout, dt, iters);           \
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    double n_instr = (double)iters * PER * (NTHR / 64) * cus;                              \
    printf("%-34s thr %d: %.3f ms, %.2f ns per instr per SIMD-wave-slot, chip %.1f G instr/s\n", NAME, NTHR, ms, ms * 1e6 / ((double)iters * PER * (NTHR / 256.0)), n_instr / ms / 1e6); \
    report(NAME, cus, NTHR, PER);                                                          \
  } while (0)
  for (int nthr : {256, 512}) {
    RUNK("fp16 16x16x32", mfma16_rate, nthr, 8);
    RUNK("scaled fp8 x fp8", (mfma_rate<0, 0>), nthr, 8);
    RUNK("scaled bf8 x fp8", (mfma_rate<1, 0>), nthr, 8);
    RUNK("scaled fp6 x fp6", (mfma_rate<2, 2>), nthr, 8);
    RUNK("scaled bf6 x bf6", (mfma_rate<3, 3>), nthr, 8);
    RUNK("scaled bf6 x fp6", (mfma_rate<3, 2>), nthr, 8);
    RUNK("scaled fp4 x fp4", (mfma_rate<4, 4>), nthr, 8);
    RUNK("scaled fp8 x fp6", (mfma_rate<0, 2>), nthr, 8);
    RUNK("scaled fp8 x fp4", (mfma_rate<0, 4>), nthr, 8);
    RUNK("mix 16 f16 + 8 fp8", (mix_rate<0, 8>), nthr, 24);
    RUNK("mix 16 f16 + 8 fp6", (mix_rate<2, 8>), nthr, 24);
    RUNK("mix 16 f16 + 8 bf6", (mix_rate<3, 8>), nthr, 24);
  }
  // conversions
  {
    std::vector<_Float16> hv(64 * 32);
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = (_Float16)(((rand() % 2001) - 1000) / 256.0f);
    half32* dv; unsigned* du;
    CK(hipMalloc(&dv, hv.size() * 2)); CK(hipMemcpy(dv, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&du, 4 * 256 * cus));
#define RUNC(NAME, KIND)                                                                    \
  do {                                                                                       \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                             \
    hipLaunchKernelGGL(cvt_rate<KIND>, dim3(cus), dim3(256), 0, 0, dv, du, dt, 10, 1.0f);    \
    hipDeviceSynchronize(); hipEventRecord(e0);                                              \
    hipLaunchKernelGGL(cvt_rate<KIND>, dim3(cus), dim3(256), 0, 0, dv, du, dt, iters, 1.0f); \
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("%-40s %.3f ms -> %.2f ns per 32 values per wave\n", NAME, ms, ms * 1e6 / iters);  \
  } while (0)
    RUNC("v_cvt_scalef32_pk32_fp6_f16", 0);
    RUNC("v_cvt_scalef32_pk32_bf6_f16", 1);
    RUNC("top-byte bf8, RNE add + perm (24 VALU)", 2);
    RUNC("16 x v_cvt_scalef32_pk_fp8_f16", 3);
    // semantics table: lane l holds values v_i = (i - 16 + l / 64) * 2^(l % 5 - 2) ... simple ramps, scales 1, 2, 0.5
    std::vector<_Float16> tv(64 * 32);
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 32; ++i) tv[l * 32 + i] = (_Float16)((i - 16 + (l & 7) / 8.0f) * std::ldexp(1.0f, (l >> 3) - 4));
    CK(hipMemcpy(dv, tv.data(), tv.size() * 2, hipMemcpyHostToDevice));
    u6 *o1, *o2; CK(hipMalloc(&o1, 64 * 24)); CK(hipMalloc(&o2, 64 * 24));
    for (float scale : {1.0f, 4.0f, 0.25f}) {
      hipLaunchKernelGGL(cvt_table, dim3(1), dim3(64), 0, 0, dv, o1, o2, scale);
      std::vector<unsigned> h1(64 * 6), h2(64 * 6);
      CK(hipMemcpy(h1.data(), o1, 64 * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, 64 * 24, hipMemcpyDeviceToHost));
      printf("scale %g: lane: in -> fp6 value (x scale) | bf6 value (x scale)\n", scale);
      for (int l : {0, 3, 8, 21, 34, 45, 63})
        for (int i : {0, 5, 13, 15, 16, 17, 19, 24, 31}) {
          double in = (double)tv[l * 32 + i];
          printf("   l%2d i%2d  in % .6f  fp6 % .6f  bf6 % .6f\n", l, i, in, fp6_e2m3(get6(&h1[l * 6], i)) * scale, bf6_e3m2(get6(&h2[l * 6], i)) * scale);
        }
    }
  }
  // MFMA semantics: A row r (lane r + 16 g holds its K-slots 32 g .. 32 g + 31) = fp6 code of 1.0 at slot k0(r), B col c all 1.0 -> D[r][c] = scaleA * scaleB
  {
    auto pack6 = [](std::vector<unsigned>& w, int i, unsigned code) { int bit = 6 * i; w[bit >> 5] |= code << (bit & 31); if ((bit & 31) > 26) w[(bit >> 5) + 1] |= code >> (32 - (bit & 31)); };
    std::vector<unsigned> ha(64 * 8, 0), hb(64 * 8, 0);
    std::vector<int> hsa(64), hsb(64);
    const unsigned ONE = 0x08;  // e2m3 1.0 = exp 1, mant 0
    for (int l = 0; l < 64; ++l) {
      std::vector<unsigned> wa(8, 0), wb(8, 0);
      int r = l & 15, g = l >> 4;
      // A: row r has 1.0 in slot (r % 32) of lane-group g = r / 4 % 4 only -> D[r][c] = scale_a(lane r + 16 g) * B[c][that slot] * scale_b(...)
      if (g == ((r >> 2) & 3)) pack6(wa, (2 * r + 1) & 31, ONE);
      for (int i = 0; i < 32; ++i) pack6(wb, i, ONE);  // B all ones
      for (int j = 0; j < 8; ++j) { ha[l * 8 + j] = wa[j]; hb[l * 8 + j] = wb[j]; }
      hsa[l] = 127 + (r & 3) + 4 * g * 0;  // row-dependent scale 2^(r&3)
      hsb[l] = 127 - (l & 15 ? 0 : 1) + (g == 2 ? 3 : 0);  // col 0: 0.5; lane group 2: x8 (only visible where A's nonzero slot is in group 2: rows 8..11)
    }
    intx8 *da, *db; floatx4* dc; int *dsa, *dsb;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dc, 64 * 16)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_sem<2>, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    std::vector<float> hc(64 * 4);
    CK(hipMemcpy(hc.data(), dc, 64 * 16, hipMemcpyDeviceToHost));
    printf("MFMA semantics (builtin operand order a, b): lane l reg e = D[?]: first operand = rows? print lanes 0..63 reg 0..3\n");
    for (int l = 0; l < 64; l += 1) printf("  l%2d: %g %g %g %g\n", l, hc[l * 4], hc[l * 4 + 1], hc[l * 4 + 2], hc[l * 4 + 3]);
  }
  return 0;
}
