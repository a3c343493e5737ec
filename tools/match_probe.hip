// Stand-alone timing probe of the fused matcher (not part of the product library).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/match_probe.hip -o tools/match_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../memvul_amd/csrc/match_topk.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

int main() {
  const int Bmax = 1024, Gmax = 1024;
  float *u, *v, *Wm, *best, *tp, *pp, *pq, *ps;
  int32_t *bi, *ti, *pi;
  CK(hipMalloc(&u, Bmax * 512 * 4)); CK(hipMalloc(&v, Gmax * 512 * 4)); CK(hipMalloc(&Wm, 6 * 512 * 4));
  CK(hipMalloc(&best, Bmax * 8)); CK(hipMalloc(&bi, Bmax * 4)); CK(hipMalloc(&tp, Bmax * 64 * 4)); CK(hipMalloc(&ti, Bmax * 64 * 4));
  CK(hipMalloc(&pp, Bmax * 1024 * 4)); CK(hipMalloc(&pq, Bmax * 1024 * 4)); CK(hipMalloc(&pi, Bmax * 1024 * 4)); CK(hipMalloc(&ps, (size_t)Bmax * Gmax * 4));
  std::vector<float> h(Gmax * 512);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f;
  CK(hipMemcpy(u, h.data(), Bmax * 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), Gmax * 512 * 4, hipMemcpyHostToDevice));
  for (auto& x : h) x = (x - 0.5f) * 0.05f;
  CK(hipMemcpy(Wm, h.data(), 6 * 512 * 4, hipMemcpyHostToDevice));
  unsigned long long* clk; CK(hipMalloc(&clk, 128)); CK(hipMemset(clk, 0, 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { int B, G, k, want_ps, alt; };
  const Case cases[] = {{256, 124, 1, 0, 0}, {256, 124, 1, 0, 1}, {256, 124, 0, 0, 0}, {4, 124, 1, 0, 0}, {4, 124, 1, 0, 1}, {1024, 124, 1, 0, 0}, {1024, 124, 1, 0, 1},
                        {256, 1000, 10, 0, 0}, {256, 1000, 10, 0, 1}, {256, 1000, 10, 0, 2}, {256, 1000, 0, 0, 0}, {256, 1000, 0, 0, 1}, {512, 1000, 10, 0, 0}, {512, 1000, 10, 0, 1},
                        {4, 256, 10, 0, 0}, {4, 256, 10, 0, 1}};
  for (const Case& c : cases) {
    MatchArgs a{};
    a.B = c.B; a.G = c.G; a.same_idx = 0; a.k = c.k;
    const bool small = c.G <= 128;
    a.nchunk = (c.G + (small ? 127 : 255)) / (small ? 128 : 256);
    a.best = best; a.best_idx = bi; a.topk_p = tp; a.topk_idx = ti; a.part_p = pp; a.part_q = pq; a.part_i = pi;
    a.psame = c.want_ps ? ps : nullptr;  // small: alt 0 delta-only / 1 with logits; large: alt 0 = 4 waves x 4 rows, 1 = 8 waves x 2 rows, 2 = the latter with logits
    auto run = [&]() {
      const dim3 grid(small ? 1 : a.nchunk, (c.B + 3) / 4);
      if (small && !c.alt) hipLaunchKernelGGL((match_topk_kernel<2, 128, 64, 0, 2>), grid, dim3(256), 0, 0, u, v, Wm, a);
      else if (small) hipLaunchKernelGGL((match_topk_kernel<2, 128, 64, 1, 2>), grid, dim3(256), 0, 0, u, v, Wm, a);
      else if (c.alt == 0) hipLaunchKernelGGL((match_topk_kernel<4, 256, 32, 0, 1>), grid, dim3(256), 0, 0, u, v, Wm, a);
      else if (c.alt == 1) hipLaunchKernelGGL((match_topk_kernel<2, 256, 32, 0, 2>), grid, dim3(512), 0, 0, u, v, Wm, a);
      else hipLaunchKernelGGL((match_topk_kernel<2, 256, 32, 1, 2>), grid, dim3(512), 0, 0, u, v, Wm, a);
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    float best_ms = 1e9f, tot = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) run();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best_ms = ms < best_ms ? ms : best_ms; tot += ms;
    }
    float mrg = 0.f;
    if (a.nchunk > 1 && c.k > 0) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) launch_topk_merge(a, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&mrg, e0, e1));
    }
    {  // one more launch with the in-kernel stamps (s_memtime: 100 MHz)
      a.clk = clk; run(); CK(hipDeviceSynchronize());
      unsigned long long t[10]; CK(hipMemcpy(t, clk, 80, hipMemcpyDeviceToHost));
      const double tot = (double)((c.k ? t[4] : t[2]) - t[0]);  // (s_memtime ticks: ~0.4 ns here; shares are what is reported)
      printf("   in-kernel shares: prologue %.0f %%  loop %.0f %%  logits %.0f %%  select %.0f %%\n", 100 * (t[1] - t[0]) / tot,
             100 * (t[2] - t[1]) / tot, c.k ? 100 * (t[3] - t[2]) / tot : 0.0, c.k ? 100 * (t[4] - t[3]) / tot : 0.0);
      a.clk = nullptr;
    }
    printf("MATCH alt=%d B=%4d G=%4d k=%2d psame=%d grid=%dx%d : %.2f us per launch back-to-back (best of 5 x 20), merge %.2f us\n", c.alt, c.B, c.G, c.k, c.want_ps,
           a.nchunk, (c.B + 3) / 4, best_ms / 20 * 1e3, mrg / 20 * 1e3);
  }
  return 0;
}
