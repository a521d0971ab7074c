// gather-copy bandwidth vs contiguous piece size, attention-like access: per window 144 tokens (2 z x 6 h x 12 w),
// token row stride 3C floats; each workgroup reads PIECE floats of q, k, v per token and writes PIECE floats of o.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PIECE>   // floats per token piece: 32 (one head), 64, 96, 192
__global__ __launch_bounds__(192) void gather_copy(const float* __restrict__ qkv, float* __restrict__ out, int Z, int H, int W,
                                                   int C, int nLon, int nHw, int pieces_per_row) {
  const int b = blockIdx.x;
  const int piece = b % pieces_per_row, rest = b / pieces_per_row;
  const int l = rest % nLon, t = rest / nLon;
  const int zwin = t / nHw, hwin = t % nHw;
  constexpr int V4 = PIECE / 4;                    // float4 per piece
  f32x4 acc = {0, 0, 0, 0};
  for (int f = threadIdx.x; f < 144 * V4; f += 192) {
    const int n = f / V4, c4 = (f % V4) * 4;
    const int zi = n / 72, r = n % 72, hi = r / 12, wi = r % 12;
    const int z = 2 * zwin + zi, h = 6 * hwin + hi, w = 12 * l + wi;
    if (h >= H) continue;
    const size_t tok = ((size_t)z * H + h) * W + w;
    const float* src = qkv + tok * 3 * C + piece * PIECE + c4;
    const f32x4 q = *reinterpret_cast<const f32x4*>(src);
    const f32x4 k = *reinterpret_cast<const f32x4*>(src + C);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 2 * C);
    acc = q + k + v;
    *reinterpret_cast<f32x4*>(out + tok * C + piece * PIECE + c4) = acc;
  }
}
template <int PIECE>
void run(int Z, int H, int W, int C) {
  const size_t N = (size_t)Z * H * W;
  float *qkv, *out;
  hipMalloc(&qkv, N * 3 * C * 4);
  hipMalloc(&out, N * C * 4);
  hipMemset(qkv, 0, N * 3 * C * 4);
  const int nLon = W / 12, nHw = (H + 5) / 6, nZw = Z / 2;
  const int ppr = C / PIECE;
  const int grid = nLon * nHw * nZw * ppr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(gather_copy<PIECE>, dim3(grid), dim3(192), 0, 0, qkv, out, Z, H, W, C, nLon, nHw, ppr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(gather_copy<PIECE>, dim3(grid), dim3(192), 0, 0, qkv, out, Z, H, W, C, nLon, nHw, ppr);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("C=%d piece=%4d B  grid=%6d  %.3f ms  %.0f GB/s\n", C, PIECE * 4, grid, ms, N * 4.0 * C * 4 / ms / 1e6);
  hipFree(qkv); hipFree(out);
}
int main() {
  run<32>(8, 181, 360, 192); run<64>(8, 181, 360, 192); run<96>(8, 181, 360, 192); run<192>(8, 181, 360, 192);
  run<32>(8, 91, 180, 384); run<64>(8, 91, 180, 384); run<128>(8, 91, 180, 384); run<384>(8, 91, 180, 384);
  return 0;
}
