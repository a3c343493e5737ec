// Probe (GPU box): is the chip-wide, power-limited MFMA rate the same for the two fp16 shapes?  v_mfma_f32_32x32x16_f16 (8 accumulators
// of 16 registers) against v_mfma_f32_16x16x32_f16 (32 accumulators of 4 registers): same FLOPs per instruction (32768), same
// register footprint, random operands, all CUs busy, no memory traffic in the loop.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe && tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const half8* ops, float* out, int iters) {
  const int tid = threadIdx.x;
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = ops[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 2; ++i) b[i] = ops[(blockIdx.x * 512 + tid) * 6 + 4 + i];
  float s = 0;
  if constexpr (SHAPE == 32) {
    f16v acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f16v){0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[i & 1], a[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f4v acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f4v){0};
    for (int it = 0; it < iters; ++it) {  // 16 instructions of 16384 FLOP = the 8 x 32768 of the other branch; 32 accumulators in two halves
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i & 1], a[(i >> 1) & 3], acc[i], 0, 0, 0);
      ++it;
#pragma unroll
      for (int i = 16; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i & 1], a[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  if (s == 1234.5f) out[0] = s;
}
template <int SHAPE> static double run(const half8* d, float* o, int wgs, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE><<<wgs, 512>>>(d, o, 2000); hipDeviceSynchronize();
  hipEventRecord(e0); k<SHAPE><<<wgs, 512>>>(d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)wgs * 8 * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12;  // 8 x 32768 (or 16 x 16384) FLOP per wave and iteration
}
int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<_Float16> h((size_t)ncu * 512 * 6 * 8);
  srand(2021);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  half8* d; float* o; hipMalloc(&d, h.size() * 2); hipMalloc(&o, 64);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep)
    printf("random fp16 operands: 8 CUs  32x32x16 %.1f TF  16x16x32 %.1f TF | all %d CUs  32x32x16 %.0f TF  16x16x32 %.0f TF\n",
           run<32>(d, o, 8, 20000), run<16>(d, o, 8, 20000), ncu, run<32>(d, o, ncu, 20000), run<16>(d, o, ncu, 20000));
  return 0;
}
