// Development probe (not part of the product library): attention_v2_kernel<4> at the bench shape on random data with the
// s_memtime instrumentation on: how much of a workgroup's time is spent waiting at the per-item hand-over
// (s_waitcnt vmcnt(0) + barrier = next item's K / V^T not landed yet, or waves out of step).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attn_probe.hip -o tools/attn_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "legacy/gemm.h"
#include "legacy/gemm_pp.h"
#include "legacy/attention_v2.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill_h(half_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = (half_t)(((x >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale);
  }
}

int main() {
  const int B = 256, S = 256, items = B * 12;
  const size_t n = (size_t)B * 12 * S * 64;
  half_t *q, *k, *vt, *ctx;
  int* lens;
  unsigned long long* clk;
  CK(hipMalloc(&q, n * 2)); CK(hipMalloc(&k, n * 2)); CK(hipMalloc(&vt, n * 2)); CK(hipMalloc(&ctx, n * 2));
  CK(hipMalloc(&lens, B * 4)); CK(hipMalloc(&clk, 512 * 8 * 3 * 8));
  std::vector<int> hl(B, S);
  CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_h, dim3(2048), dim3(256), 0, 0, q, n, 1u, 0.5f);
  hipLaunchKernelGGL(fill_h, dim3(2048), dim3(256), 0, 0, k, n, 2u, 1.0f);
  hipLaunchKernelGGL(fill_h, dim3(2048), dim3(256), 0, 0, vt, n, 3u, 1.0f);
  CK(hipFuncSetAttribute((const void*)attention_v2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES(4)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int probe = 0; probe < 2; ++probe) {
    AttnArgs a{q, k, vt, lens, ctx, S, B, probe ? clk : nullptr};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((attention_v2_kernel<4>), dim3(256), dim3(512), ATT2_LDS_BYTES(4), 0, a, items);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((attention_v2_kernel<4>), dim3(256), dim3(512), ATT2_LDS_BYTES(4), 0, a, items);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("attention_v2<4> B=256 S=256 %s: %.1f us per launch\n", probe ? "instrumented" : "plain", ms * 100.0f);
    if (probe) {
      std::vector<unsigned long long> h(256 * 8 * 3);
      CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
      double tot = 0, wait = 0, vm = 0;
      for (int i = 0; i < 256 * 8; ++i) { tot += (double)h[3 * i]; wait += (double)h[3 * i + 1]; vm += (double)h[3 * i + 2]; }
      const double nw = 256 * 8;
      printf("  per wave (12 items): %.0f ticks total; at the hand-over %.0f (%.1f %%) = %.0f in s_waitcnt vmcnt(0) + %.0f at the barrier; tick rate %.2f GHz\n",
             tot / nw, wait / nw, 100.0 * wait / tot, vm / nw, (wait - vm) / nw, tot / nw / (ms * 100.0f * 1e-6) / 1e9);
      printf("  workgroup 0, per wave [total wait vm]:");
      for (int w = 0; w < 8; ++w) printf(" [%llu %llu %llu]", h[3 * w], h[3 * w + 1], h[3 * w + 2]);
      printf("\n");
    }
  }
  // where the time goes: the same launch with the transcendental and / or the matrix instructions taken out
  auto time_abl = [&](auto kern, const char* what) -> int {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES(4)));
    AttnArgs a{q, k, vt, lens, ctx, S, B, nullptr};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), ATT2_LDS_BYTES(4), 0, a, items);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), ATT2_LDS_BYTES(4), 0, a, items);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("attention_v2<4> %-28s %.1f us per launch\n", what, ms * 100.0f);
    return 0;
  };
  if (time_abl(attention_v2_kernel<4, 1, 0>, "as shipped:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 1>, "no v_exp_f32:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 2>, "no MFMA:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 3>, "no v_exp_f32, no MFMA:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 4>, "no LDS fragment reads:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 7>, "none of the three:")) return 2;
  // memory streams (round 2): which of the four costs what
  if (time_abl(attention_v2_kernel<4, 1, 8>, "Q loads from one cached tile:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 16>, "O stores to one cached tile:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 32>, "no K / V^T LDS-DMA:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 24>, "Q and O cached:")) return 2;
  if (time_abl(attention_v2_kernel<4, 1, 56>, "no memory streams at all:")) return 2;
  // working set vs the 256 MB Infinity Cache: back-to-back launches over the first Bs batch rows re-read the same
  // Q / K / V^T (0.29 MB per row and head in, 0.1 MB out), which only stay cached when they fit
  for (int Bs : {256, 192, 128, 64}) {
    AttnArgs a{q, k, vt, lens, ctx, S, Bs, nullptr};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((attention_v2_kernel<4>), dim3(256), dim3(512), ATT2_LDS_BYTES(4), 0, a, Bs * 12);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((attention_v2_kernel<4>), dim3(256), dim3(512), ATT2_LDS_BYTES(4), 0, a, Bs * 12);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mb = Bs * 12 * 4.0 * 32768 / 1e6;
    printf("attention_v2<4> B=%3d (%.0f MB touched per launch): %.1f us per launch = %.2f TB/s\n", Bs, mb, ms * 100.0f, mb / (ms * 100.0f));
  }
  return 0;
}
