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

// Backward of the kernel above (autograd of reference layers.py:368-415 for the module taken on its own).  One workgroup per
// (window type, head) walks the n_lon longitude windows, so d_esb[t][head] = sum_l dS accumulates in LDS (82 KB) without atomics.
// Per window: pass 1, one query row per thread -- log-sum-exp, delta = sum_j P dP, dq = scale * sum_j dS k_j, dS into the LDS
// tile; pass 2, one key per thread -- dk_j = sum_i dS q_i (q pre-scaled), dv_j = sum_i P dO_i, with P recomputed from the
// row statistics pass 1 left in LDS.  Scores are recomputed instead of stored (plain VALU dot products; not on the hot path).
__global__ __launch_bounds__(256) void attn_windows_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ esb,
                                                                   const float* __restrict__ mask, long long mask_lon_stride,
                                                                   const float* __restrict__ dout, float* __restrict__ dqkv,
                                                                   float* __restrict__ d_esb, int n_lon, int types, int heads,
                                                                   int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* Qs = reinterpret_cast<f32x4*>(smem_raw);            // [144][8] scaled q
  f32x4* Ks = Qs + PANGU_WTOK * 8;
  f32x4* Vs = Ks + PANGU_WTOK * 8;
  f32x4* Gs = Vs + PANGU_WTOK * 8;                           // dO
  float* lse_s = reinterpret_cast<float*>(Gs + PANGU_WTOK * 8);
  float* del_s = lse_s + PANGU_WTOK;
  float* dB = del_s + PANGU_WTOK;                            // [144][144] sum over the longitude windows of dS
  const int head = blockIdx.x % heads, t = blockIdx.x / heads;
  const int C3 = 3 * C, tid = threadIdx.x;
  const float scale = 0.17677669529663687f;
  for (int i = tid; i < PANGU_WTOK * PANGU_WTOK; i += 256) dB[i] = 0.f;
  const float* bias_t = esb + ((size_t)t * heads + head) * PANGU_WTOK * PANGU_WTOK;
  for (int l = 0; l < n_lon; ++l) {
    const size_t row0 = ((size_t)l * types + t) * PANGU_WTOK;
    const float* mask_t = mask ? mask + (size_t)l * mask_lon_stride + (size_t)t * PANGU_WTOK * PANGU_WTOK : nullptr;
    __syncthreads();                                         // the previous window's readers are done
    for (int i = tid; i < PANGU_WTOK * 8; i += 256) {
      const int r = i >> 3, c = i & 7;
      const float* src = qkv + (row0 + r) * C3 + head * PANGU_HEAD_DIM + c * 4;
      Qs[i] = *reinterpret_cast<const f32x4*>(src) * scale;
      Ks[i] = *reinterpret_cast<const f32x4*>(src + C);
      Vs[i] = *reinterpret_cast<const f32x4*>(src + 2 * C);
      Gs[i] = *reinterpret_cast<const f32x4*>(dout + (row0 + r) * C + head * PANGU_HEAD_DIM + c * 4);
    }
    __syncthreads();
    auto dot = [](const f32x4* a, const f32x4* b) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 x = a[c], y = b[c];
        s = fmaf(x[0], y[0], s); s = fmaf(x[1], y[1], s); s = fmaf(x[2], y[2], s); s = fmaf(x[3], y[3], s);
      }
      return s;
    };
    if (tid < PANGU_WTOK) {                                  // ---- pass 1: query row i = tid
      const int i = tid;
      const float* brow = bias_t + (size_t)i * PANGU_WTOK;
      const float* mrow = mask_t ? mask_t + (size_t)i * PANGU_WTOK : nullptr;
      auto score = [&](int j) { return dot(Qs + i * 8, Ks + j * 8) + brow[j] + (mrow ? mrow[j] : 0.f); };
      float m = -INFINITY;
      for (int j = 0; j < PANGU_WTOK; ++j) m = fmaxf(m, score(j));
      float sum = 0.f, delta = 0.f;
      for (int j = 0; j < PANGU_WTOK; ++j) {
        const float p = expf(score(j) - m);
        sum += p;
        delta += p * dot(Gs + i * 8, Vs + j * 8);
      }
      const float lse = m + logf(sum);
      delta /= sum;
      f32x4 dq[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) dq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < PANGU_WTOK; ++j) {
        const float p = expf(score(j) - lse);
        const float ds = p * (dot(Gs + i * 8, Vs + j * 8) - delta);
        dB[i * PANGU_WTOK + j] += ds;
#pragma unroll
        for (int c = 0; c < 8; ++c) dq[c] += Ks[j * 8 + c] * ds;
      }
      lse_s[i] = lse;
      del_s[i] = delta;
      float* dst = dqkv + (row0 + i) * C3 + head * PANGU_HEAD_DIM;
#pragma unroll
      for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(dst + c * 4) = dq[c] * scale;
    }
    __syncthreads();
    if (tid < PANGU_WTOK) {                                  // ---- pass 2: key j = tid
      const int j = tid;
      f32x4 dk[8], dv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { dk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[c] = dk[c]; }
      for (int i = 0; i < PANGU_WTOK; ++i) {
        const float s = dot(Qs + i * 8, Ks + j * 8) + bias_t[(size_t)i * PANGU_WTOK + j] +
                        (mask_t ? mask_t[(size_t)i * PANGU_WTOK + j] : 0.f);
        const float p = expf(s - lse_s[i]);
        const float ds = p * (dot(Gs + i * 8, Vs + j * 8) - del_s[i]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          dk[c] += Qs[i * 8 + c] * ds;                       // Qs holds scale * q
          dv[c] += Gs[i * 8 + c] * p;
        }
      }
      float* dst = dqkv + (row0 + j) * C3 + head * PANGU_HEAD_DIM;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        *reinterpret_cast<f32x4*>(dst + C + c * 4) = dk[c];
        *reinterpret_cast<f32x4*>(dst + 2 * C + c * 4) = dv[c];
      }
    }
  }
  __syncthreads();
  float* out_t = d_esb + ((size_t)t * heads + head) * PANGU_WTOK * PANGU_WTOK;
  for (int i = tid; i < PANGU_WTOK * PANGU_WTOK; i += 256) out_t[i] = dB[i];
}

}  // namespace

extern "C" int pangu_attn_windows_bwd(pangu_stream_t stream, const float* qkv, const float* esb, const float* mask,
                                      long long mask_lon_stride, const float* dout, float* dqkv, float* d_esb, int n_lon,
                                      int types, int heads, int C) {
  if (!qkv || !esb || !dout || !dqkv || !d_esb) return PANGU_E_NULL;
  if (n_lon <= 0 || types <= 0 || heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  if (mask && mask_lon_stride != 0 && mask_lon_stride != (long long)types * PANGU_WTOK * PANGU_WTOK) return PANGU_E_ARG;
  const size_t shm = (size_t)4 * PANGU_WTOK * 8 * sizeof(f32x4) + 2 * PANGU_WTOK * sizeof(float) +
                     (size_t)PANGU_WTOK * PANGU_WTOK * sizeof(float);
  PANGU_ENSURE_DYN_LDS(attn_windows_bwd_f32_kernel, shm);
  hipLaunchKernelGGL(attn_windows_bwd_f32_kernel, dim3((unsigned)(types * heads)), dim3(256), shm, (hipStream_t)stream, qkv, esb,
                     mask, mask_lon_stride, dout, dqkv, d_esb, n_lon, types, heads, C);
  return pangu_launch_status();
}

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
