// Probe (GPU box): what v_cvt_scalef32_pk_fp8_f32 computes — scale direction and overflow behaviour — against the plain
// v_cvt_pk_fp8_f32 behind a clamp that common.h pack_fp8x4 uses.  Prints a table; no assertion.
//   hipcc --offload-arch=gfx950 -O2 tools/cvt_fp8_probe.hip -o tools/cvt_fp8_probe && tools/cvt_fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef short shortx2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int n, float scale, unsigned* plain, unsigned* scaled, unsigned* raw) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = x[i];
  float c = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, c, v, false);
  plain[i] = (unsigned)v & 0xff;
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, a, r, false);  // no clamp: what does overflow give?
  raw[i] = (unsigned)r & 0xff;
  shortx2 o = {0, 0};
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, a, scale, false);
  scaled[i] = (unsigned)(unsigned short)o[0] & 0xff;
}
static float dec(unsigned b) {
  int e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? m * std::ldexp(1.f, -9) : (8 + m) * std::ldexp(1.f, e - 10);
  if (e == 15 && m == 7) v = NAN;
  return (b & 0x80) ? -v : v;
}
int main() {
  std::vector<float> x = {0.f, 1.f, 1.0625f, 1.1875f, 3.3f, -7.7f, 100.f, 447.f, 448.f, 449.f, 480.f, 1000.f, -5000.f, 1e30f, 0.001f, 0.002f, 0.0009f, 2.5e-3f,
                          INFINITY, NAN};
  int n = (int)x.size();
  float *dx; unsigned *dp, *ds, *dr;
  hipMalloc(&dx, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 0.25f, 4.0f, 1.0f / 8192.f}) {
    k<<<1, 64>>>(dx, n, scale, dp, ds, dr);
    std::vector<unsigned> p(n), s(n), r(n);
    hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    printf("scale operand = %g\n  %14s | clamp+cvt_pk (bits val) | cvt_pk no clamp | cvt_scalef32 (bits val) | val/x\n", scale, "x");
    for (int i = 0; i < n; ++i)
      printf("  %14g | %02x %10g | %02x %10g | %02x %10g | %g\n", x[i], p[i], dec(p[i]), r[i], dec(r[i]), s[i], dec(s[i]), x[i] != 0 ? dec(s[i]) / x[i] : 0.f);
  }
  return 0;
}
