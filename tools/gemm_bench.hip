// Stand-alone development probe for the GEMM kernels (not part of the product library): correctness of
// gemm_pp against the validated gemm256 epilogues, then interleaved A/B timing rounds of variants/ablations
// at the production shapes.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o tools/gemm_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include "legacy/gemm.h"
#include "legacy/gemm_pp.h"

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "HIP error %s at %s:%d: %s\n", #x, __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void fill_h(half_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (half_t)(((hash32((uint32_t)i * 2654435761u + seed) >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale);
}
__global__ void fill_f(float* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = ((hash32((uint32_t)i * 2654435761u + seed) >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale;
}
// max |a-b| and count of |a-b| > tol * max(1,|a|)
template <typename T>
__global__ void cmp_k(const T* a, const T* b, size_t n, float tol, unsigned* maxbits, unsigned long long* bad) {
  float mx = 0.f;
  unsigned long long nb = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = (float)a[i], y = (float)b[i];
    const float d = fabsf(x - y);
    if (!(d <= tol * fmaxf(1.f, fabsf(x)))) nb++;
    mx = fmaxf(mx, d == d ? d : 1e30f);
  }
  atomicMax(maxbits, __float_as_uint(mx));
  if (nb) atomicAdd(bad, nb);
}

static hipStream_t st;
static int NCU = 256;
static unsigned* d_maxbits;
static unsigned long long* d_bad;

template <typename T>
bool compare(const char* what, const T* a, const T* b, size_t n, float tol) {
  CK(hipMemsetAsync(d_maxbits, 0, 4, st));
  CK(hipMemsetAsync(d_bad, 0, 8, st));
  hipLaunchKernelGGL((cmp_k<T>), dim3(2048), dim3(256), 0, st, a, b, n, tol, d_maxbits, d_bad);
  unsigned mb;
  unsigned long long bad;
  CK(hipMemcpyAsync(&mb, d_maxbits, 4, hipMemcpyDeviceToHost, st));
  CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  float mx;
  memcpy(&mx, &mb, 4);
  printf("CHECK %-28s n=%zu max|diff|=%.3e bad(>%g rel)=%llu %s\n", what, n, mx, tol, bad, bad == 0 ? "OK" : "FAIL");
  fflush(stdout);
  return bad == 0;
}

static int choose_gn(int tn, int gn_max) {
  int g = 1;
  for (int d = 1; d <= gn_max && d <= tn; ++d)
    if (tn % d == 0) g = d;
  return g;
}

template <int EPI>
void run_old(GemmArgs a) {
  a.GN = choose_gn(a.N / 256, 4);
  hipLaunchKernelGGL((gemm256_kernel<EPI>), dim3((a.M / 256) * (a.N / 256)), dim3(512), G256_LDS_BYTES, st, a);
}
static int g_grid_override = 0;
static bool g_only_phases = false;  // argv[4] == "phases": only the s_memtime phase timeline at the end
template <int EPI, int DIST, int ABL, int SCHED = 0, int COAL = 0>
void run_pp(GemmArgs a) {
  auto kern = gemm_pp_kernel<EPI, DIST, ABL, SCHED, COAL>;
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
    attr = true;
  }
  a.GN = choose_gn(a.N / 256, 4);
  const int tiles = (a.M / 256) * (a.N / 256);
  hipLaunchKernelGGL(kern, dim3(g_grid_override ? g_grid_override : std::min(tiles, NCU)), dim3(512), PP_LDS_BYTES, st, a);
}

struct Variant {
  std::string name;
  double flops;
  std::function<void()> fn;
  std::vector<float> ms;
};

static void time_group(const char* title, std::vector<Variant>& vs, int rounds, int reps, FILE* js) {
  if (g_only_phases) return;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto& v : vs) { v.fn(); }  // warm
  CK(hipStreamSynchronize(st));
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) v.fn();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / reps);
    }
  CK(hipGetLastError());
  printf("== %s\n", title);
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float mn = v.ms.front(), med = v.ms[v.ms.size() / 2];
    printf("TIME %-34s min %8.1f us  med %8.1f us  %7.1f TF (med)\n", v.name.c_str(), mn * 1e3, med * 1e3, v.flops / (med * 1e-3) / 1e12);
    if (js) fprintf(js, "{\"group\":\"%s\",\"variant\":\"%s\",\"min_us\":%.2f,\"med_us\":%.2f,\"tflops_med\":%.1f}\n", title, v.name.c_str(), mn * 1e3, med * 1e3, v.flops / (med * 1e-3) / 1e12);
  }
  fflush(stdout);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  g_only_phases = argc > 4 && !strcmp(argv[4], "phases");
  const int S = 256;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  NCU = prop.multiProcessorCount;
  printf("device %s CUs=%d M=%d\n", prop.name, NCU, M);
  CK(hipStreamCreate(&st));
  CK(hipMalloc(&d_maxbits, 4));
  CK(hipMalloc(&d_bad, 8));
  system("mkdir -p gpurun_out");
  FILE* js = fopen("gpurun_out/gemm_bench.jsonl", "w");

  half_t *A768, *A3072, *Wqkv, *Wo, *W1, *W2;
  float *bias, *of_ref, *of_new, *xr_ref, *xr_new, *xr_src;
  half_t *o16_ref, *o16_new, *q_ref, *k_ref, *vt_ref, *q_new, *k_new, *vt_new;
  CK(hipMalloc(&A768, (size_t)M * 768 * 2));
  CK(hipMalloc(&A3072, (size_t)M * 3072 * 2));
  CK(hipMalloc(&Wqkv, (size_t)2304 * 768 * 2));
  CK(hipMalloc(&Wo, (size_t)768 * 768 * 2));
  CK(hipMalloc(&W1, (size_t)3072 * 768 * 2));
  CK(hipMalloc(&W2, (size_t)768 * 3072 * 2));
  CK(hipMalloc(&bias, 3072 * 4));
  CK(hipMalloc(&of_ref, (size_t)M * 2304 * 4));
  CK(hipMalloc(&of_new, (size_t)M * 2304 * 4));
  CK(hipMalloc(&xr_ref, (size_t)M * 768 * 4));
  CK(hipMalloc(&xr_new, (size_t)M * 768 * 4));
  CK(hipMalloc(&xr_src, (size_t)M * 768 * 4));
  CK(hipMalloc(&o16_ref, (size_t)M * 3072 * 2));
  CK(hipMalloc(&o16_new, (size_t)M * 3072 * 2));
  const size_t qn = (size_t)M * 768;
  CK(hipMalloc(&q_ref, qn * 2)); CK(hipMalloc(&k_ref, qn * 2)); CK(hipMalloc(&vt_ref, qn * 2));
  CK(hipMalloc(&q_new, qn * 2)); CK(hipMalloc(&k_new, qn * 2)); CK(hipMalloc(&vt_new, qn * 2));
  auto fh = [&](half_t* p, size_t n, uint32_t seed, float sc) { hipLaunchKernelGGL(fill_h, dim3(4096), dim3(256), 0, st, p, n, seed, sc); };
  auto ff = [&](float* p, size_t n, uint32_t seed, float sc) { hipLaunchKernelGGL(fill_f, dim3(4096), dim3(256), 0, st, p, n, seed, sc); };
  fh(A768, (size_t)M * 768, 1, 1.0f);
  fh(A3072, (size_t)M * 3072, 2, 1.0f);
  fh(Wqkv, (size_t)2304 * 768, 3, 0.05f);
  fh(Wo, (size_t)768 * 768, 4, 0.05f);
  fh(W1, (size_t)3072 * 768, 5, 0.05f);
  fh(W2, (size_t)768 * 3072, 6, 0.03f);
  ff(bias, 3072, 7, 0.5f);
  ff(xr_src, (size_t)M * 768, 8, 1.0f);
  CK(hipStreamSynchronize(st));

  auto base = [&](const half_t* A, const half_t* W, int N, int K) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.M = M; g.Mreal = M; g.N = N; g.K = K; g.S = S;
    return g;
  };
  bool ok = true;
  // ---- correctness -----------------------------------------------------------------------------
  {  // F32, N=2304 K=768
    GemmArgs g = base(A768, Wqkv, 2304, 768);
    g.outf = of_ref; run_old<EPI_F32>(g);
    g.outf = of_new; CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 6, 0>(g);
    ok &= compare("f32 N2304 K768 D6", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 4, 0>(g);
    ok &= compare("f32 N2304 K768 D4", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 2, 0>(g);
    ok &= compare("f32 N2304 K768 D2", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 6, PP_ABL_NOSTAGGER>(g);
    ok &= compare("f32 N2304 K768 D6 nostagger", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 4, 0, 1>(g);
    ok &= compare("f32 N2304 K768 S1 F4", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 4, 0, 1, 1>(g);
    ok &= compare("f32 N2304 K768 S1 F4 COAL", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 3, 0, 1>(g);
    ok &= compare("f32 N2304 K768 S1 F3", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 2, 0, 1>(g);
    ok &= compare("f32 N2304 K768 S1 F2", of_ref, of_new, (size_t)M * 2304, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 2304 * 4, st)); run_pp<PP_F32, 4, PP_ABL_NOSTAGGER, 1>(g);
    ok &= compare("f32 N2304 K768 S1 F4 nostagger", of_ref, of_new, (size_t)M * 2304, 1e-4f);
  }
  {  // F32, N=768 K=3072
    GemmArgs g = base(A3072, W2, 768, 3072);
    g.outf = of_ref; run_old<EPI_F32>(g);
    g.outf = of_new; CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 768 * 4, st)); run_pp<PP_F32, 6, 0>(g);
    ok &= compare("f32 N768 K3072 D6", of_ref, of_new, (size_t)M * 768, 1e-4f);
    CK(hipMemsetAsync(of_new, 0xff, (size_t)M * 768 * 4, st)); run_pp<PP_F32, 4, 0, 1>(g);
    ok &= compare("f32 N768 K3072 S1 F4", of_ref, of_new, (size_t)M * 768, 1e-4f);
  }
  {  // GELU
    GemmArgs g = base(A768, W1, 3072, 768);
    g.out16 = o16_ref; run_old<EPI_GELU>(g);
    g.out16 = o16_new; CK(hipMemsetAsync(o16_new, 0xff, (size_t)M * 3072 * 2, st)); run_pp<PP_GELU, 6, 0>(g);
    ok &= compare("gelu N3072 K768 D6", o16_ref, o16_new, (size_t)M * 3072, 2e-3f);
    CK(hipMemsetAsync(o16_new, 0xff, (size_t)M * 3072 * 2, st)); run_pp<PP_GELU, 4, 0, 1>(g);
    ok &= compare("gelu N3072 K768 S1 F4", o16_ref, o16_new, (size_t)M * 3072, 2e-3f);
    CK(hipMemsetAsync(o16_new, 0xff, (size_t)M * 3072 * 2, st)); run_pp<PP_GELU, 4, 0, 1, 1>(g);
    ok &= compare("gelu N3072 K768 S1 F4 COAL", o16_ref, o16_new, (size_t)M * 3072, 2e-3f);
  }
  {  // RES (both K)
    GemmArgs g = base(A768, Wo, 768, 768);
    CK(hipMemcpyAsync(xr_ref, xr_src, (size_t)M * 768 * 4, hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(xr_new, xr_src, (size_t)M * 768 * 4, hipMemcpyDeviceToDevice, st));
    g.xres = xr_ref; run_old<EPI_RES>(g);
    g.xres = xr_new; run_pp<PP_RES, 6, 0>(g);
    ok &= compare("res N768 K768 D6", xr_ref, xr_new, (size_t)M * 768, 1e-4f);
    GemmArgs h = base(A3072, W2, 768, 3072);
    CK(hipMemcpyAsync(xr_ref, xr_src, (size_t)M * 768 * 4, hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(xr_new, xr_src, (size_t)M * 768 * 4, hipMemcpyDeviceToDevice, st));
    h.xres = xr_ref; run_old<EPI_RES>(h);
    h.xres = xr_new; run_pp<PP_RES, 6, 0>(h);
    ok &= compare("res N768 K3072 D6", xr_ref, xr_new, (size_t)M * 768, 1e-4f);
    CK(hipMemcpyAsync(xr_new, xr_src, (size_t)M * 768 * 4, hipMemcpyDeviceToDevice, st));
    h.xres = xr_new; run_pp<PP_RES, 4, 0, 1>(h);
    ok &= compare("res N768 K3072 S1 F4", xr_ref, xr_new, (size_t)M * 768, 1e-4f);
    CK(hipMemcpyAsync(xr_new, xr_src, (size_t)M * 768 * 4, hipMemcpyDeviceToDevice, st));
    h.xres = xr_new; run_pp<PP_RES, 4, 0, 1, 1>(h);
    ok &= compare("res N768 K3072 S1 F4 COAL", xr_ref, xr_new, (size_t)M * 768, 1e-4f);
  }
  {  // QKV
    GemmArgs g = base(A768, Wqkv, 2304, 768);
    g.q = q_ref; g.k = k_ref; g.vt = vt_ref; run_old<EPI_QKV>(g);
    CK(hipMemsetAsync(q_new, 0xff, qn * 2, st)); CK(hipMemsetAsync(k_new, 0xff, qn * 2, st)); CK(hipMemsetAsync(vt_new, 0xff, qn * 2, st));
    GemmArgs a1 = base(A768, Wqkv, 1536, 768);
    a1.q = q_new; a1.k = k_new; run_pp<PP_QK, 6, 0>(a1);
    GemmArgs a2 = base(A768, Wqkv + (size_t)1536 * 768, 768, 768);
    a2.bias = bias + 1536; a2.vt = vt_new; run_pp<PP_VT, 6, 0>(a2);
    ok &= compare("qkv q", q_ref, q_new, qn, 2e-3f);
    ok &= compare("qkv k", k_ref, k_new, qn, 2e-3f);
    ok &= compare("qkv vt", vt_ref, vt_new, qn, 2e-3f);
    CK(hipMemsetAsync(q_new, 0xff, qn * 2, st)); CK(hipMemsetAsync(k_new, 0xff, qn * 2, st)); CK(hipMemsetAsync(vt_new, 0xff, qn * 2, st));
    run_pp<PP_QK, 4, 0, 1, 1>(a1); run_pp<PP_VT, 4, 0, 1, 1>(a2);
    ok &= compare("qkv q S1 COAL", q_ref, q_new, qn, 2e-3f);
    ok &= compare("qkv k S1 COAL", k_ref, k_new, qn, 2e-3f);
    ok &= compare("qkv vt S1 COAL", vt_ref, vt_new, qn, 2e-3f);
  }
  printf("CORRECTNESS %s\n", ok ? "ALL OK" : "FAILED");
  if (js) fprintf(js, "{\"correct\":%s}\n", ok ? "true" : "false");

  // ---- timing ----------------------------------------------------------------------------------
  {
    GemmArgs g = base(A768, W1, 3072, 768);
    g.out16 = o16_new;
    const double fl = 2.0 * M * 3072.0 * 768.0;
    std::vector<Variant> vs;
    vs.push_back({"old256 gelu", fl, [=] { run_old<EPI_GELU>(g); }, {}});
    vs.push_back({"pp S0 D6 gelu", fl, [=] { run_pp<PP_GELU, 6, 0>(g); }, {}});
    vs.push_back({"pp S1 F4 gelu", fl, [=] { run_pp<PP_GELU, 4, 0, 1>(g); }, {}});
    vs.push_back({"pp S1 F4 gelu COAL", fl, [=] { run_pp<PP_GELU, 4, 0, 1, 1>(g); }, {}});
    vs.push_back({"pp S0 D6 gelu COAL", fl, [=] { run_pp<PP_GELU, 6, 0, 0, 1>(g); }, {}});
    vs.push_back({"pp S1 F4 f16 (no gelu)", fl, [=] { run_pp<PP_F16, 4, 0, 1>(g); }, {}});
    vs.push_back({"pp S1 F4 f16 COAL", fl, [=] { run_pp<PP_F16, 4, 0, 1, 1>(g); }, {}});
    vs.push_back({"pp S1 F4 ABL noepi", fl, [=] { run_pp<PP_GELU, 4, PP_ABL_NOEPI, 1>(g); }, {}});
    time_group("ffn1 M x3072 x768 (gelu)", vs, rounds, reps, js);
  }
  {
    GemmArgs g = base(A768, Wqkv, 2304, 768);
    g.q = q_ref; g.k = k_ref; g.vt = vt_ref;
    GemmArgs a1 = base(A768, Wqkv, 1536, 768);
    a1.q = q_new; a1.k = k_new;
    GemmArgs a2 = base(A768, Wqkv + (size_t)1536 * 768, 768, 768);
    a2.bias = bias + 1536; a2.vt = vt_new;
    const double fl = 2.0 * M * 2304.0 * 768.0;
    std::vector<Variant> vs;
    vs.push_back({"old256 qkv", fl, [=] { run_old<EPI_QKV>(g); }, {}});
    vs.push_back({"pp qk+vt D6", fl, [=] { run_pp<PP_QK, 6, 0>(a1); run_pp<PP_VT, 6, 0>(a2); }, {}});
    vs.push_back({"pp S1 F4 qk+vt", fl, [=] { run_pp<PP_QK, 4, 0, 1>(a1); run_pp<PP_VT, 4, 0, 1>(a2); }, {}});
    vs.push_back({"pp S1 F4 qk+vt COAL", fl, [=] { run_pp<PP_QK, 4, 0, 1, 1>(a1); run_pp<PP_VT, 4, 0, 1, 1>(a2); }, {}});
    vs.push_back({"pp qk only D6 (2/3 flops)", fl * 2 / 3, [=] { run_pp<PP_QK, 6, 0>(a1); }, {}});
    vs.push_back({"pp vt only D6 (1/3 flops)", fl / 3, [=] { run_pp<PP_VT, 6, 0>(a2); }, {}});
    time_group("qkv M x2304 x768", vs, rounds, reps, js);
  }
  {
    GemmArgs g = base(A768, Wo, 768, 768);
    g.xres = xr_new;
    const double fl = 2.0 * M * 768.0 * 768.0;
    std::vector<Variant> vs;
    vs.push_back({"old256 res", fl, [=] { run_old<EPI_RES>(g); }, {}});
    vs.push_back({"pp res D6", fl, [=] { run_pp<PP_RES, 6, 0>(g); }, {}});
    vs.push_back({"pp S1 F4 res", fl, [=] { run_pp<PP_RES, 4, 0, 1>(g); }, {}});
    vs.push_back({"pp S1 F4 res COAL", fl, [=] { run_pp<PP_RES, 4, 0, 1, 1>(g); }, {}});
    time_group("attn-out M x768 x768 (res)", vs, rounds, reps, js);
  }
  {
    GemmArgs g = base(A3072, W2, 768, 3072);
    g.xres = xr_new;
    const double fl = 2.0 * M * 768.0 * 3072.0;
    std::vector<Variant> vs;
    vs.push_back({"old256 res", fl, [=] { run_old<EPI_RES>(g); }, {}});
    vs.push_back({"pp res D6", fl, [=] { run_pp<PP_RES, 6, 0>(g); }, {}});
    vs.push_back({"pp S1 F4 res", fl, [=] { run_pp<PP_RES, 4, 0, 1>(g); }, {}});
    vs.push_back({"pp S1 F4 res COAL", fl, [=] { run_pp<PP_RES, 4, 0, 1, 1>(g); }, {}});
    time_group("ffn2 M x768 x3072 (res)", vs, rounds, reps, js);
  }
  {  // residual GEMMs: what the accumulator-init loads and the epilogue stores cost (lockstep bursts)
    for (int K : {768, 3072}) {
      GemmArgs g = base(K == 768 ? A768 : A3072, K == 768 ? Wo : W2, 768, K);
      g.xres = xr_new;
      g.outf = of_new;
      const double fl = 2.0 * M * 768.0 * K;
      std::vector<Variant> vs;
      vs.push_back({"res  (loads + stores)", fl, [=] { run_pp<PP_RES, 4, 0, 1, 1>(g); }, {}});
      vs.push_back({"res noepi (loads only)", fl, [=] { run_pp<PP_RES, 4, PP_ABL_NOEPI, 1, 1>(g); }, {}});
      vs.push_back({"f32  (stores only)", fl, [=] { run_pp<PP_F32, 4, 0, 1, 1>(g); }, {}});
      vs.push_back({"f32 noepi (neither)", fl, [=] { run_pp<PP_F32, 4, PP_ABL_NOEPI, 1, 1>(g); }, {}});
      char title[64];
      snprintf(title, sizeof title, "residual GEMM M x768 x%d: memory phases", K);
      time_group(title, vs, rounds, reps, js);
    }
  }
  {  // per-tile time of one workgroup as a function of how many workgroups run (memory-burst contention)
    GemmArgs g = base(A768, W1, 3072, 768);
    g.out16 = o16_new;
    GemmArgs r = base(A768, Wo, 768, 768);
    r.xres = xr_new;
    GemmArgs r2 = base(A3072, W2, 768, 3072);
    r2.xres = xr_new;
    for (int grid : {256, 64, 8}) {
      std::vector<Variant> vs;
      const double fl = 2.0 * M * 3072.0 * 768.0;
      auto G = [grid](std::function<void()> f) { return [=] { g_grid_override = grid; f(); g_grid_override = 0; }; };
      vs.push_back({"gelu S1F4 COAL", fl, G([=] { run_pp<PP_GELU, 4, 0, 1, 1>(g); }), {}});
      vs.push_back({"f16 S1F4 COAL", fl, G([=] { run_pp<PP_F16, 4, 0, 1, 1>(g); }), {}});
      vs.push_back({"noepi S1F4", fl, G([=] { run_pp<PP_GELU, 4, PP_ABL_NOEPI, 1>(g); }), {}});
      vs.push_back({"res K768 S1F4 COAL", 2.0 * M * 768.0 * 768.0, G([=] { run_pp<PP_RES, 4, 0, 1, 1>(r); }), {}});
      vs.push_back({"res K3072 S1F4 COAL", 2.0 * M * 768.0 * 3072.0, G([=] { run_pp<PP_RES, 4, 0, 1, 1>(r2); }), {}});
      char title[64];
      snprintf(title, sizeof title, "grid=%d (us x grid/256 = per-256-normalised)", grid);
      time_group(title, vs, 2, 1, js);
    }
  }
  {  // where a workgroup's time goes: s_memtime sums of accumulator init / main loop / epilogue per wave (PP_ABL_CLK)
    unsigned long long* dclk;
    CK(hipMalloc(&dclk, 256 * 8 * 4 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    GemmArgs f1 = base(A768, W1, 3072, 768);
    f1.out16 = o16_new;
    f1.clk = dclk;
    GemmArgs ro = base(A768, Wo, 768, 768);
    ro.xres = xr_new;
    ro.clk = dclk;
    GemmArgs r2 = base(A3072, W2, 768, 3072);
    r2.xres = xr_new;
    r2.clk = dclk;
    struct Case { const char* name; std::function<void()> fn; int tiles; };
    std::vector<Case> cases = {
        {"ffn1 gelu  M x3072 x768 ", [=] { run_pp<PP_GELU, 4, PP_ABL_CLK, 1, 1>(f1); }, (M / 256) * 12},
        {"ffn1 f16 (no gelu)       ", [=] { run_pp<PP_F16, 4, PP_ABL_CLK, 1, 1>(f1); }, (M / 256) * 12},
        {"attn-out res M x768 x768 ", [=] { run_pp<PP_RES, 4, PP_ABL_CLK, 1, 1>(ro); }, (M / 256) * 3},
        {"ffn2 res   M x768 x3072", [=] { run_pp<PP_RES, 4, PP_ABL_CLK, 1, 1>(r2); }, (M / 256) * 3},
        {"attn-out res, 3/4 of the bytes", [=] { run_pp<PP_RES, 4, PP_ABL_CLK | PP_ABL_B34, 1, 1>(ro); }, (M / 256) * 3},
        {"ffn2 res, 3/4 of the bytes    ", [=] { run_pp<PP_RES, 4, PP_ABL_CLK | PP_ABL_B34, 1, 1>(r2); }, (M / 256) * 3},
        {"attn-out res (again)     ", [=] { run_pp<PP_RES, 4, PP_ABL_CLK, 1, 1>(ro); }, (M / 256) * 3},
        {"ffn2 res (again)         ", [=] { run_pp<PP_RES, 4, PP_ABL_CLK, 1, 1>(r2); }, (M / 256) * 3},
    };
    std::vector<Case> more;
    for (int sg : {7}) {  // start-up stagger: hash(workgroup) % (sg + 1) sleeps of ~8k cycles
      if (getenv("GEMM_BENCH_NO_STAGGER")) break;
      GemmArgs ros = ro, r2s = r2, f1s = f1;
      ros.stagger = r2s.stagger = f1s.stagger = sg;
      static char names[9][48];
      static int ni = 0;
      snprintf(names[ni], 48, "attn-out res, stagger %d  ", sg);
      more.push_back({names[ni++], [=] { run_pp<PP_RES, 4, PP_ABL_CLK, 1, 1>(ros); }, (M / 256) * 3});
      snprintf(names[ni], 48, "ffn2 res, stagger %d      ", sg);
      more.push_back({names[ni++], [=] { run_pp<PP_RES, 4, PP_ABL_CLK, 1, 1>(r2s); }, (M / 256) * 3});
      snprintf(names[ni], 48, "ffn1 gelu, stagger %d     ", sg);
      more.push_back({names[ni++], [=] { run_pp<PP_GELU, 4, PP_ABL_CLK, 1, 1>(f1s); }, (M / 256) * 12});
    }
    for (auto& c : more) cases.push_back(c);
    for (auto& c : cases) {
      for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) { CK(hipMemsetAsync(dclk, 0, 256 * 8 * 4 * 8, st)); CK(hipEventRecord(e0, st)); }
        c.fn();
        if (rep == 2) CK(hipEventRecord(e1, st));
      }
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> hc(256 * 8 * 4);
      CK(hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost));
      double tot = 0, ini = 0, mn = 0, ep = 0;
      const int nw = NCU * 8;
      for (int i = 0; i < nw; ++i) { tot += (double)hc[4 * i]; ini += (double)hc[4 * i + 1]; mn += (double)hc[4 * i + 2]; ep += (double)hc[4 * i + 3]; }
      const double us_per_tick = ms * 1e3 / (tot / nw);
      const double tiles_per_wg = (double)c.tiles / NCU;
      printf("PHASES %s wall %7.1f us, %.1f tiles per workgroup; per tile: init %5.2f us  main loop %6.2f us  epilogue %5.2f us  (per launch: %5.1f / %6.1f / %5.1f us; %.0f %% in the main loop)\n",
             c.name, ms * 1e3, tiles_per_wg, ini / nw * us_per_tick / tiles_per_wg, mn / nw * us_per_tick / tiles_per_wg,
             ep / nw * us_per_tick / tiles_per_wg, ini / nw * us_per_tick, mn / nw * us_per_tick, ep / nw * us_per_tick, 100.0 * mn / (ini + mn + ep));
      printf("       workgroup 0 waves [init main epi] ticks:");
      for (int w = 0; w < 8; ++w) printf(" [%llu %llu %llu]", hc[4 * w + 1], hc[4 * w + 2], hc[4 * w + 3]);
      printf("\n");
    }
    CK(hipFree(dclk));
  }
  {  // effective shader clock: s_memtime ticks of each workgroup's whole run / wall time of the launch
    unsigned long long* dclk;
    CK(hipMalloc(&dclk, 256 * 8));
    GemmArgs g = base(A768, W1, 3072, 768);
    g.out16 = o16_new;
    g.clk = dclk;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256, 64, 8}) {
      if (g_only_phases) break;
      for (int which = 0; which < 3; ++which) {
        g_grid_override = grid;
        CK(hipMemsetAsync(dclk, 0, 256 * 8, st));
        for (int rep = 0; rep < 3; ++rep) {
          if (rep == 2) CK(hipEventRecord(e0, st));
          if (which == 0) run_pp<PP_GELU, 4, PP_ABL_NOEPI, 1>(g);
          else if (which == 1) run_pp<PP_GELU, 4, 0, 1>(g);
          else run_pp<PP_GELU, 4, 13, 1>(g);
          if (rep == 2) CK(hipEventRecord(e1, st));
        }
        CK(hipEventSynchronize(e1));
        g_grid_override = 0;
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> hc(256);
        CK(hipMemcpy(hc.data(), dclk, 256 * 8, hipMemcpyDeviceToHost));
        double sum = 0, mx = 0;
        for (int i = 0; i < grid; ++i) { sum += (double)hc[i]; mx = std::max(mx, (double)hc[i]); }
        printf("CLOCK grid=%3d %-22s wall %8.1f us  ticks avg %.0f max %.0f  -> %.3f GHz (max ticks / wall)\n", grid,
               which == 0 ? "noepi" : which == 1 ? "gelu" : "mfma-only skeleton", ms * 1e3, sum / grid, mx, mx / (ms * 1e-3) / 1e9);
      }
    }
  }
  if (js) fclose(js);
  return ok ? 0 : 1;
}
