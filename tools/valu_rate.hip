// VALU issue-rate microbenchmark at one wave per SIMD: cycles per instruction for plain / DPP / packed fp32 forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, b = 0.5f, c = 0.25f;
  float2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, pb = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) asm volatile(REP16("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (KIND == 1) asm volatile(REP16("v_fmac_f32_dpp %0, %4, %5 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %2, %4, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %4, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (KIND == 2) asm volatile(REP16("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1\n\t") : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pb));
    if (KIND == 3) asm volatile(REP16("v_fma_f32 %0, |%4|, %5, %0\n\tv_sub_f32 %1, %4, %1\n\tv_fma_f32 %2, |%4|, %5, %2\n\tv_sub_f32 %3, %5, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (KIND == 4) asm volatile(REP16("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (KIND == 5) asm volatile(REP16("v_fmac_f32_dpp %0, %4, |%5| quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %1, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %2, %4, |%5| quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %3, %4, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y;
}
template <int KIND>
void run(const char* name, int blocks, int threads, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s blocks=%4d threads=%3d : %.3f ns per wave-instruction\n", name, blocks, threads, ms * 1e6 / (iters * 64.0));
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  for (int threads : {256}) for (int blocks : {1, 256, 1024}) {  // 1 wave per SIMD on one CU / on every CU; 4 waves per SIMD
    run<0>("v_fma_f32 (VOP3)", blocks, threads, out);
    run<4>("v_fmac_f32 (VOP2)", blocks, threads, out);
    run<1>("v_fmac_f32_dpp", blocks, threads, out);
    run<5>("v_fmac/v_sub dpp |x|", blocks, threads, out);
    run<3>("v_fma |x| + v_sub", blocks, threads, out);
    run<2>("v_pk_fma_f32", blocks, threads, out);
  }
  return 0;
}
