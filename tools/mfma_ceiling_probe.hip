// Probe (GPU box): what the matrix pipes deliver when NOTHING else happens — v_mfma_f32_32x32x16_f16 from registers, no LDS, no
// memory traffic in the loop — with every CU busy, against a handful of CUs busy.  The gap between the two is the chip's power
// budget at this instruction (the effective clock falls under a full-chip MFMA load), i.e. the ceiling any GEMM main loop on
// all 256 CUs runs under; operand values matter (toggling), so random and zero operands are both measured.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_ceiling_probe.hip -o tools/mfma_ceiling_probe && tools/mfma_ceiling_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(const half8* ops, float* out, int iters, unsigned long long* ticks) {
  const int tid = threadIdx.x;
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = ops[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 2; ++i) b[i] = ops[(blockIdx.x * 512 + tid) * 6 + 4 + i];
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f16v){0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i & 1], a[i & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 1234.5f) out[0] = s;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int maxwg = 2 * ncu;
  std::vector<_Float16> h((size_t)maxwg * 512 * 6 * 8);
  half8* d; float* o; unsigned long long* tk;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 64); hipMalloc(&tk, maxwg * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("CUs %d; one workgroup = 8 waves (2 per SIMD), 8 independent accumulators per wave\n", ncu);
  printf("%-28s %10s %12s %12s %14s\n", "case", "wall us", "TFLOP/s", "x CUs/busy PF", "implied GHz");
  for (int zero = 0; zero < 2; ++zero) {
    srand(2021);
    for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int wgs : {8, 64, ncu, 2 * ncu}) {
      const int iters = 20000;
      k<8><<<wgs, 512>>>(d, o, 2000, tk);  // warm-up
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<8><<<wgs, 512>>>(d, o, iters, tk);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> t(wgs);
      hipMemcpy(t.data(), tk, wgs * 8, hipMemcpyDeviceToHost);
      double tmax = 0; for (auto v : t) tmax = v > tmax ? v : tmax;
      const double flops = (double)wgs * 8 * iters * 8 * 2.0 * 32 * 32 * 16;
      const double tf = flops / (ms * 1e-3) / 1e12;
      char name[64]; snprintf(name, sizeof name, "%s operands, %d WGs", zero ? "zero" : "random", wgs);
      printf("%-28s %10.1f %12.1f %12.3f %14.3f\n", name, ms * 1e3, tf, tf / (wgs < ncu ? wgs : ncu) * ncu / 1e3, tf * 1e12 / ((wgs < ncu ? wgs : ncu) * 4096.0) / 1e9);  // 4 SIMDs x 32768 FLOP / 32 cycles per CU and cycle
      (void)tmax;
    }
  }
  return 0;
}
