// Probe (GPU box): does the ORDER in which a wave feeds operand fragments to the matrix pipe change the power-limited rate?
// Same MFMA count, same random fragments (4 A x 2 B per wave), 8 accumulators acc[a][b], all 256 CUs busy, three issue orders:
//   0  both operands change at every MFMA        (A0B0 A1B1 A2B0 A3B1 A0B1 A1B0 A2B1 A3B0)
//   1  one operand shared by consecutive MFMAs   (A0B0 A0B1 A1B1 A1B0 A2B0 A2B1 A3B1 A3B0: Gray order)
//   2  the same pair eight times                 (A0B0 x 8, accumulators still distinct)
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_toggle_probe.hip -o tools/mfma_toggle_probe && tools/mfma_toggle_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define M(A, B, C) acc[C] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[B], a[A], acc[C], 0, 0, 0)
template <int MODE>
__global__ __launch_bounds__(512) void k(const half8* ops, float* out, int iters) {
  const int tid = threadIdx.x;
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = ops[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 2; ++i) b[i] = ops[(blockIdx.x * 512 + tid) * 6 + 4 + i];
  f16v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f16v){0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { M(0, 0, 0); M(1, 1, 3); M(2, 0, 4); M(3, 1, 7); M(0, 1, 1); M(1, 0, 2); M(2, 1, 5); M(3, 0, 6); }
    if (MODE == 1) { M(0, 0, 0); M(0, 1, 1); M(1, 1, 3); M(1, 0, 2); M(2, 0, 4); M(2, 1, 5); M(3, 1, 7); M(3, 0, 6); }
    if (MODE == 2) { M(0, 0, 0); M(0, 0, 1); M(0, 0, 2); M(0, 0, 3); M(0, 0, 4); M(0, 0, 5); M(0, 0, 6); M(0, 0, 7); }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 1234.5f) out[0] = s;
}
template <int MODE> static double run(const half8* d, float* o, int wgs, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wgs, 512>>>(d, o, 2000); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<wgs, 512>>>(d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)wgs * 8 * iters * 8 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
}
int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<_Float16> h((size_t)ncu * 512 * 6 * 8);
  srand(2021);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  half8* d; float* o; hipMalloc(&d, h.size() * 2); hipMalloc(&o, 64);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep)
    printf("all %d CUs, random fp16 operands: both operands change %.0f TF | one shared (Gray order) %.0f TF | same pair %.0f TF\n", ncu,
           run<0>(d, o, ncu, 20000), run<1>(d, o, ncu, 20000), run<2>(d, o, ncu, 20000));
  return 0;
}
