// EarthAttention3D.forward on an already PARTITIONED tensor (reference models/layers.py:360-421 between linear1 and linear2):
// the module's own calling convention -- x_window (nLon, types, 144, C) and an explicit mask tensor -- kept for callers that
// use the module outside EarthSpecificBlock.  NOT on the hot path: the block's forward / backward go through the fused kernels
// of attn_f32.hip / attn_bf16.hip, which fold partition, shift, padding and the closed-form mask into addressing.  Every
// window slot is an ordinary token here (pad slots included) and the mask is whatever tensor the caller passes.
//
// One (window, head) per 256-thread workgroup; K and V of the head in LDS (36 KB), one query row per thread (144 active),
// scores recomputed in the second pass instead of stored (plain VALU dot products: 3 x 144 x 144 x 32 FMA per workgroup).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void attn_windows_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ esb,
                                                               const float* __restrict__ mask, long long mask_lon_stride,
                                                               float* __restrict__ out, int types, int heads, int C) {
  __shared__ f32x4 Ks[PANGU_WTOK * 8], Vs[PANGU_WTOK * 8];
  const int head = blockIdx.x % heads;
  const int win = blockIdx.x / heads;              // l * types + t
  const int t = win % types, l = win / types;
  const size_t row0 = (size_t)win * PANGU_WTOK;
  const int C3 = 3 * C;
  for (int i = threadIdx.x; i < PANGU_WTOK * 8; i += 256) {
    const int r = i >> 3, c = i & 7;
    const float* src = qkv + (row0 + r) * C3 + head * PANGU_HEAD_DIM + c * 4;
    Ks[i] = *reinterpret_cast<const f32x4*>(src + C);
    Vs[i] = *reinterpret_cast<const f32x4*>(src + 2 * C);
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= PANGU_WTOK) return;
  const float scale = 0.17677669529663687f;        // 32 ** -0.5 (layers.py:285, :374)
  f32x4 q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    q[c] = *reinterpret_cast<const f32x4*>(qkv + (row0 + i) * C3 + head * PANGU_HEAD_DIM + c * 4) * scale;
  const float* brow = esb + (((size_t)t * heads + head) * PANGU_WTOK + i) * PANGU_WTOK;
  const float* mrow = mask ? mask + (size_t)l * mask_lon_stride + ((size_t)t * PANGU_WTOK + i) * PANGU_WTOK : nullptr;
  auto score = [&](int j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 k = Ks[j * 8 + c];
      s = fmaf(q[c][0], k[0], s); s = fmaf(q[c][1], k[1], s); s = fmaf(q[c][2], k[2], s); s = fmaf(q[c][3], k[3], s);
    }
    s += brow[j];
    if (mrow) s += mrow[j];
    return s;
  };
  float m = -INFINITY;
  for (int j = 0; j < PANGU_WTOK; ++j) m = fmaxf(m, score(j));
  f32x4 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sum = 0.f;
  for (int j = 0; j < PANGU_WTOK; ++j) {
    const float p = expf(score(j) - m);
    sum += p;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += Vs[j * 8 + c] * p;
  }
  const float inv = 1.0f / sum;
  float* dst = out + (row0 + i) * C + head * PANGU_HEAD_DIM;
#pragma unroll
  for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(dst + c * 4) = acc[c] * inv;
}

}  // namespace

extern "C" int pangu_attn_windows_fwd(pangu_stream_t stream, const float* qkv, const float* esb, const float* mask,
                                      long long mask_lon_stride, float* out, int n_lon, int types, int heads, int C) {
  if (!qkv || !esb || !out) return PANGU_E_NULL;
  if (n_lon <= 0 || types <= 0 || heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  if (mask && mask_lon_stride != 0 && mask_lon_stride != (long long)types * PANGU_WTOK * PANGU_WTOK) return PANGU_E_ARG;
  const long long blocks = (long long)n_lon * types * heads;
  if (blocks > 0x7FFFFFFFll) return PANGU_E_RANGE;
  hipLaunchKernelGGL(attn_windows_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, qkv, esb, mask,
                     mask_lon_stride, out, types, heads, C);
  return pangu_launch_status();
}
