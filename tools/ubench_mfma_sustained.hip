// Sustained f32-MFMA rate: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators, 3 waves per SIMD, launched
// repeatedly for ~1 s; prints TF/s per launch so clock/power throttling under sustained matrix load is visible.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_sustained.hip -o /tmp/mfma_sustained && /tmp/mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 768 * 256 * sizeof(float));
  const int grid = 768, iters = 6000;                     // ~10 ms per launch
  hipEvent_t e[64];
  for (int i = 0; i < 64; ++i) hipEventCreate(&e[i]);
  hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e[0]);
  for (int l = 1; l < 64; ++l) {
    hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e[l]);
  }
  hipDeviceSynchronize();
  const double flop = (double)grid * 4 * iters * 32 * 4096;
  for (int l = 1; l < 64; l += 3) {
    float ms;
    hipEventElapsedTime(&ms, e[l - 1], e[l]);
    float t0;
    hipEventElapsedTime(&t0, e[0], e[l]);
    printf("t=%7.1f ms  launch %2d: %.3f ms  %.1f TF/s\n", t0, l, ms, flop / ms / 1e9);
  }
  return 0;
}
