// MFMA issue-rate microbenchmark (gfx950): hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Measures how many SIMD cycles one v_mfma_f32_16x16x4_f32 / 32x32x2_f32 costs alone and with VALU, packed-VALU, v_exp or LDS reads
// interleaved in the same wave, at 1 and 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMA stream with VALU work interleaved in the same wave (NV fma per MFMA)
template <int NV>
__global__ __launch_bounds__(256) void k16v(float* out, int iters, float a, float b) {
  f32x4 acc[3];
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = __builtin_fmaf(v[q], b, a);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// MFMA stream with LDS reads interleaved (ND ds_read per MFMA, b32 or b128), results consumed by later MFMAs as operands
template <int ND, bool B128>
__global__ __launch_bounds__(256) void k16d(float* out, int iters, float a, float b) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a;
  __syncthreads();
  f32x4 acc[3];
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float x = a;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < ND; ++q) {
          if (B128) { f32x4 t = *reinterpret_cast<volatile f32x4*>(&lds[(lane * 4 + 256 * ((u + q + i) & 7)) & 4095]); x = t[0]; }
          else x = *reinterpret_cast<volatile float*>(&lds[(lane + 64 * ((u + q + i) & 31)) & 4095]);
        }
      }
  }
  float s = x;
  for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMA + transcendental
template <int NT>
__global__ __launch_bounds__(256) void k16t(float* out, int iters, float a, float b) {
  f32x4 acc[3];
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a * 0.01f * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NT; ++q) v[q] = __builtin_amdgcn_exp2f(v[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// packed fp32 VALU
template <int NV>
__global__ __launch_bounds__(256) void k16p(float* out, int iters, float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x4 acc[3];
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, a - i};
  const f32x2 bb = {b, b}, aa = {a, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = __builtin_elementwise_fma(v[q], bb, aa);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int blocks_per_cu, int nacc, double flop_per_mfma, int extra_per = 0) {
  float* out;
  hipMalloc(&out, 256 * 64 * 256 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)grid * 4 * iters * 8 * nacc;
  printf("%-28s waves/SIMD=%d nacc=%d : %.3f ms  %.1f TF/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, blocks_per_cu, nacc, ms,
         mfmas * flop_per_mfma / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / (256.0 * 4)));
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 3; w += 2) {
    run("16x16x4 + 0 valu", k16v<0>, w, 3, 2048);
    run("16x16x4 + 4 valu", k16v<4>, w, 3, 2048);
    run("16x16x4 + 2 pk_fma", k16p<2>, w, 3, 2048);
    run("16x16x4 + 4 pk_fma", k16p<4>, w, 3, 2048);
    run("16x16x4 + 1 exp", k16t<1>, w, 3, 2048);
    run("16x16x4 + 2 exp", k16t<2>, w, 3, 2048);
    run("16x16x4 + 1 ds_b32", k16d<1, false>, w, 3, 2048);
    run("16x16x4 + 2 ds_b32", k16d<2, false>, w, 3, 2048);
    run("16x16x4 + 1 ds_b128", k16d<1, true>, w, 3, 2048);
  }
  return 0;
}
