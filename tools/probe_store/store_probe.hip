// Round-4 probe (not product): does an epilogue-like store stream run faster when every wave-store covers 8 FULL 128-byte lines instead of 16 half
// lines (16 rows x 64 B: what gemm_pp.h's [32 rows][64 B] transposition image gives)?  Same bytes, same instruction count, 256 workgroups x 8 waves, each
// wave writing its own 128-row x 64-column fp16 sub-tile of a [M][N] matrix tile after tile (the persistent GEMM's output pattern), nothing else running.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_store/store_probe.hip -o tools/probe_store/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: 16 rows x 64 B per instruction; 1: 8 rows x 128 B per instruction; 2: 4 rows x 256 B (row = 128 fp16 columns: two waves' columns merged)
__global__ __launch_bounds__(512) void store_kernel(uint8_t* out, int M, int N, int tiles_per_wg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int tn_count = N / 256;
  u32x4 v = {(unsigned)tid, (unsigned)blockIdx.x, 3u, 4u};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int L = t * gridDim.x + blockIdx.x;
    const int tm = L / tn_count, tn = L % tn_count;
    const size_t row0 = (size_t)tm * 256 + wr * 128;
    const size_t colb = ((size_t)tn * 256 + wc * 64) * 2;  // byte offset of the wave's 64 fp16 columns
    if (MODE == 0) {
      for (int i = 0; i < 8; ++i)      // 8 blocks of 16 rows
        for (int j = 0; j < 2; ++j) {  // two 64-byte halves of the 128-byte row piece
          uint8_t* p = out + (row0 + i * 16 + (lane >> 2)) * (size_t)N * 2 + colb + j * 64 + (lane & 3) * 16;
          *(u32x4*)p = v;
        }
    } else if (MODE == 1) {
      for (int i = 0; i < 16; ++i) {   // 16 blocks of 8 rows x 128 B
        uint8_t* p = out + (row0 + i * 8 + (lane >> 3)) * (size_t)N * 2 + colb + (lane & 7) * 16;
        *(u32x4*)p = v;
      }
    } else {
      // wave pairs (wc even/odd) write 256-byte pieces: this wave takes rows of parity (wc & 1) of the pair's 128 x 128-column area
      const size_t colb2 = ((size_t)tn * 256 + (wc >> 1) * 128) * 2;
      for (int i = 0; i < 16; ++i) {   // 16 instructions of 4 rows x 256 B
        uint8_t* p = out + (row0 + i * 8 + (wc & 1) * 4 + (lane >> 4)) * (size_t)N * 2 + colb2 + (lane & 15) * 16;
        *(u32x4*)p = v;
      }
    }
  }
}

int main() {
  const int M = 65536;
  for (int N : {3072, 768}) {
    uint8_t* d;
    const size_t bytes = (size_t)M * N * 2;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    const int tiles = (M / 256) * (N / 256), grid = 256, per = tiles / grid;
    for (int rep = 0; rep < 2; ++rep)
      for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(512), 0, 0, d, M, N, per);
          else if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 0, 0, d, M, N, per);
          else hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(512), 0, 0, d, M, N, per);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("N=%4d mode %d (%s): %.1f us per %zu MB = %.2f TB/s\n", N, mode, mode == 0 ? "16 rows x 64 B" : mode == 1 ? "8 rows x 128 B" : "4 rows x 256 B",
               ms * 100, bytes >> 20, bytes / (ms * 1e-4) / 1e12);
      }
    hipFree(d);
  }
  return 0;
}
