// Backward of the HBM-bound row kernels (fp32): LayerNorm-residual, down/up-sample LayerNorm with their
// gather/scatter addressing, and the patch-recovery scatter.  One wave per row; LayerNorm statistics are
// recomputed from the saved pre-norm row (one extra wave reduction) instead of being stored.
// dgamma/dbeta: per-lane register accumulators over the rows a wave visits, reduced across the workgroup's
// 4 waves in LDS, one atomicAdd per column per workgroup.
#include "common.h"

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int ROWS_PER_BLOCK = 4;

typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ inline u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ inline f32x4 ld4(const u16* p) {
  const u32x2 u = *reinterpret_cast<const u32x2*>(p);
  return f32x4{__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xFFFF0000u),
               __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xFFFF0000u)};
}
__device__ inline void st4(u16* p, f32x4 v) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// y: pre-norm row (in), g: upstream gradient row already multiplied by the branch scale (in) -> dy (out, in y)
template <int NV>
__device__ inline void row_ln_bwd(f32x4 (&y)[NV], const f32x4 (&g)[NV], int nvec, int lane, int C,
                                  const float* __restrict__ gamma, f32x4 (&dg)[NV], f32x4 (&db)[NV]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) s += (y[i][0] + y[i][1]) + (y[i][2] + y[i][3]);
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float d = y[i][c] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(q) / C + LN_EPS);
  f32x4 gg[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * (lane + 64 * i));
      y[i] = (y[i] - mean) * rstd;                  // xhat
      gg[i] = g[i] * gm;
      dg[i] += g[i] * y[i];
      db[i] += g[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) { s1 += gg[i][c]; s2 += gg[i][c] * y[i][c]; }
    }
  const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) y[i] = (gg[i] - m1 - y[i] * m2) * rstd;
}

template <int NV>
__device__ inline void flush_param_grads(const f32x4 (&dg)[NV], const f32x4 (&db)[NV], int nvec, int lane, int wave,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* red) {
  // red: [2][4 waves][4*nvec] floats of LDS
  const int C = 4 * nvec;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) {
      *reinterpret_cast<f32x4*>(&red[wave * C + 4 * (lane + 64 * i)]) = dg[i];
      *reinterpret_cast<f32x4*>(&red[(4 + wave) * C + 4 * (lane + 64 * i)]) = db[i];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(&dgamma[c], (red[c] + red[C + c]) + (red[2 * C + c] + red[3 * C + c]));
    atomicAdd(&dbeta[c], (red[4 * C + c] + red[5 * C + c]) + (red[6 * C + c] + red[7 * C + c]));
  }
}

template <int NV, typename T>
__global__ __launch_bounds__(256) void ln_residual_bwd_kernel(const T* __restrict__ dout, int lddo,
                                                              const T* __restrict__ yin,
                                                              const float* __restrict__ gamma, T* __restrict__ dy,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int N, int C, float branch_scale) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 2;
  f32x4 dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = dg[i]; }
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < N; row += gridDim.x * ROWS_PER_BLOCK) {
    f32x4 y[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) {
        y[i] = ld4(yin + (size_t)row * C + 4 * (lane + 64 * i));
        g[i] = ld4(dout + (size_t)row * lddo + 4 * (lane + 64 * i)) * branch_scale;
      }
    row_ln_bwd<NV>(y, g, nvec, lane, C, gamma, dg, db);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) st4(dy + (size_t)row * C + 4 * (lane + 64 * i), y[i]);
  }
  flush_param_grads<NV>(dg, db, nvec, lane, wave, dgamma, dbeta, red);
}

// bf16 fast path of the post-norm residual backward for C % 8 == 0, C <= 512 (the model's 192 / 384): 16-B loads (8 channels
// per lane), LPR lanes per row (32 -> two rows per wave at C <= 256), gamma and the dgamma/dbeta accumulators in registers,
// both inputs of two row groups requested together before any arithmetic, persistent grid.  Same math as row_ln_bwd.
template <int LPR, int UNR>
__global__ __launch_bounds__(256) void ln_residual_bwd_bf16_v8_kernel(const u16* __restrict__ dout, int lddo,
                                                                      const u16* __restrict__ yin,
                                                                      const float* __restrict__ gamma, u16* __restrict__ dy,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      int N, int C, float branch_scale) {
  constexpr int RPW = 64 / LPR, GROUPS = 4 * RPW;      // UNR row groups in flight per wave (each row is a chain of 3 cross-lane reductions)
  __shared__ float red[2 * GROUPS * (LPR * 8)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const bool act = l * 8 < C;
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    gm[c] = act ? gamma[l * 8 + c] : 0.f;
    dg[c] = 0.f;
    db[c] = 0.f;
  }
  const float inv_c = 1.0f / C;
  const int rows_per_block = 4 * RPW * UNR;
  for (int base = blockIdx.x * rows_per_block + wave * RPW * UNR; base < N; base += gridDim.x * rows_per_block) {
    u32x4 yv[UNR], gv[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int row = base + u * RPW + sub;
      ok[u] = act && row < N;
      yv[u] = u32x4{0u, 0u, 0u, 0u};
      gv[u] = u32x4{0u, 0u, 0u, 0u};
      if (ok[u]) {
        yv[u] = *reinterpret_cast<const u32x4*>(yin + (size_t)row * C + l * 8);
        gv[u] = *reinterpret_cast<const u32x4*>(dout + (size_t)row * lddo + l * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float y[8], g[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        y[2 * c] = __builtin_bit_cast(float, yv[u][c] << 16);
        y[2 * c + 1] = __builtin_bit_cast(float, yv[u][c] & 0xFFFF0000u);
        g[2 * c] = __builtin_bit_cast(float, gv[u][c] << 16) * branch_scale;
        g[2 * c + 1] = __builtin_bit_cast(float, gv[u][c] & 0xFFFF0000u) * branch_scale;
      }
      float s = ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
      s = group_sum<LPR>(s);
      const float mean = s * inv_c;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) { const float d = y[c] - mean; q += d * d; }
      if (!act) q = 0.f;
      q = group_sum<LPR>(q);
      const float rstd = rsqrtf(q * inv_c + LN_EPS);
      float gg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        y[c] = act ? (y[c] - mean) * rstd : 0.f;        // xhat
        gg[c] = g[c] * gm[c];
        dg[c] += g[c] * y[c];
        db[c] += g[c];
        s1 += gg[c];
        s2 += gg[c] * y[c];
      }
      s1 = group_sum<LPR>(s1);
      s2 = group_sum<LPR>(s2);
      const float m1 = s1 * inv_c, m2 = s2 * inv_c;
      if (ok[u]) {
        u32x4 o4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          o4[c] = pack_bf16x2((gg[2 * c] - m1 - y[2 * c] * m2) * rstd, (gg[2 * c + 1] - m1 - y[2 * c + 1] * m2) * rstd);
        const int row = base + u * RPW + sub;
        *reinterpret_cast<u32x4*>(dy + (size_t)row * C + l * 8) = o4;
      }
    }
  }
  // flush the per-lane parameter gradients: [GROUPS][C] through LDS, one atomic per channel and workgroup
  const int grp = wave * RPW + sub;
  constexpr int CW = LPR * 8;
  if (act) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      red[grp * CW + l * 8 + c] = dg[c];
      red[(GROUPS + grp) * CW + l * 8 + c] = db[c];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int gq = 0; gq < GROUPS; ++gq) {
      a += red[gq * CW + c];
      b += red[(GROUPS + gq) * CW + c];
    }
    atomicAdd(&dgamma[c], a);
    atomicAdd(&dbeta[c], b);
  }
}

// The same kernel with 16 channels (two 16-B loads per tensor) per lane and 32 lanes per row: C = 384 rows take 24 of the 32
// lanes, TWO rows per wave instead of one row on 48 of 64 lanes -- half the instructions per row (the C = 384 launches of the
// 8-channel form ran at 3.9 TB/s against 5.3 at C = 192, where it already had two rows per wave).  C % 16 == 0, 256 < C <= 512.
template <int UNR>
__global__ __launch_bounds__(256) void ln_residual_bwd_bf16_v16_kernel(const u16* __restrict__ dout, int lddo,
                                                                       const u16* __restrict__ yin,
                                                                       const float* __restrict__ gamma, u16* __restrict__ dy,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                       int N, int C, float branch_scale) {
  constexpr int LPR = 32, RPW = 2, GROUPS = 8, CW = LPR * 16;
  __shared__ float red[2 * GROUPS * CW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const bool act = l * 16 < C;
  float gm[16], dg[16], db[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    gm[c] = act ? gamma[l * 16 + c] : 0.f;
    dg[c] = 0.f;
    db[c] = 0.f;
  }
  const float inv_c = 1.0f / C;
  const int rows_per_block = 4 * RPW * UNR;
  for (int base = blockIdx.x * rows_per_block + wave * RPW * UNR; base < N; base += gridDim.x * rows_per_block) {
    u32x4 ya[UNR], yb[UNR], ga[UNR], gb[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int row = base + u * RPW + sub;
      ok[u] = act && row < N;
      ya[u] = u32x4{0u, 0u, 0u, 0u};
      yb[u] = ya[u]; ga[u] = ya[u]; gb[u] = ya[u];
      if (ok[u]) {
        const u16* yp = yin + (size_t)row * C + l * 16;
        const u16* gp = dout + (size_t)row * lddo + l * 16;
        ya[u] = *reinterpret_cast<const u32x4*>(yp);
        yb[u] = *reinterpret_cast<const u32x4*>(yp + 8);
        ga[u] = *reinterpret_cast<const u32x4*>(gp);
        gb[u] = *reinterpret_cast<const u32x4*>(gp + 8);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float y[16], g[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        y[2 * c] = __builtin_bit_cast(float, ya[u][c] << 16);
        y[2 * c + 1] = __builtin_bit_cast(float, ya[u][c] & 0xFFFF0000u);
        y[8 + 2 * c] = __builtin_bit_cast(float, yb[u][c] << 16);
        y[9 + 2 * c] = __builtin_bit_cast(float, yb[u][c] & 0xFFFF0000u);
        g[2 * c] = __builtin_bit_cast(float, ga[u][c] << 16) * branch_scale;
        g[2 * c + 1] = __builtin_bit_cast(float, ga[u][c] & 0xFFFF0000u) * branch_scale;
        g[8 + 2 * c] = __builtin_bit_cast(float, gb[u][c] << 16) * branch_scale;
        g[9 + 2 * c] = __builtin_bit_cast(float, gb[u][c] & 0xFFFF0000u) * branch_scale;
      }
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; c += 4) s += (y[c] + y[c + 1]) + (y[c + 2] + y[c + 3]);
      s = group_sum<LPR>(s);
      const float mean = s * inv_c;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { const float d = y[c] - mean; q += d * d; }
      if (!act) q = 0.f;
      q = group_sum<LPR>(q);
      const float rstd = rsqrtf(q * inv_c + LN_EPS);
      float gg[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        y[c] = act ? (y[c] - mean) * rstd : 0.f;        // xhat
        gg[c] = g[c] * gm[c];
        dg[c] += g[c] * y[c];
        db[c] += g[c];
        s1 += gg[c];
        s2 += gg[c] * y[c];
      }
      s1 = group_sum<LPR>(s1);
      s2 = group_sum<LPR>(s2);
      const float m1 = s1 * inv_c, m2 = s2 * inv_c;
      if (ok[u]) {
        u32x4 o0, o1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o0[c] = pack_bf16x2((gg[2 * c] - m1 - y[2 * c] * m2) * rstd, (gg[2 * c + 1] - m1 - y[2 * c + 1] * m2) * rstd);
          o1[c] = pack_bf16x2((gg[8 + 2 * c] - m1 - y[8 + 2 * c] * m2) * rstd, (gg[9 + 2 * c] - m1 - y[9 + 2 * c] * m2) * rstd);
        }
        const int row = base + u * RPW + sub;
        u16* op = dy + (size_t)row * C + l * 16;
        *reinterpret_cast<u32x4*>(op) = o0;
        *reinterpret_cast<u32x4*>(op + 8) = o1;
      }
    }
  }
  const int grp = wave * RPW + sub;
  if (act) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      red[grp * CW + l * 16 + c] = dg[c];
      red[(GROUPS + grp) * CW + l * 16 + c] = db[c];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int gq = 0; gq < GROUPS; ++gq) {
      a += red[gq * CW + c];
      b += red[(GROUPS + gq) * CW + c];
    }
    atomicAdd(&dgamma[c], a);
    atomicAdd(&dbeta[c], b);
  }
}

// ---- bf16 fast paths of the two resampling LayerNorm backwards (round 3; forward twins in rowops_bf16.hip): 16-B accesses, gamma
// and the dgamma / dbeta accumulators in registers, several rows in flight per wave, persistent grid.  Same math as row_ln_bwd.
__device__ inline void unpack8(const u32x4 v, float* f) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f[2 * c] = __builtin_bit_cast(float, v[c] << 16);
    f[2 * c + 1] = __builtin_bit_cast(float, v[c] & 0xFFFF0000u);
  }
}

// Down-sampling backward: one wave per row of 4C channels, 16 channels of ONE source token per lane (C % 16 == 0, C <= 256)
template <int UNR>
__global__ __launch_bounds__(256) void downsample_ln_bwd_bf16_v16_kernel(const u16* __restrict__ dout, const u16* __restrict__ x,
                                                                         int ldx, const float* __restrict__ gamma,
                                                                         u16* __restrict__ dx, float* __restrict__ dgamma,
                                                                         float* __restrict__ dbeta, int Z, int H, int W, int C,
                                                                         const u16* __restrict__ add) {
  __shared__ float red[2 * 4 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lpq = C >> 4;
  const bool act = lane < 4 * lpq;
  const int quad = act ? lane / lpq : 0, c0 = (lane - quad * lpq) * 16;
  const int H2 = (H + 1) / 2, W2 = W / 2, C4 = 4 * C, N2 = Z * H2 * W2;
  float gm[16], dg[16], db[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    gm[c] = act ? gamma[lane * 16 + c] : 0.f;
    dg[c] = 0.f;
    db[c] = 0.f;
  }
  const float inv_c = 1.0f / C4;
  for (int base = (blockIdx.x * 4 + wave) * UNR; base < N2; base += gridDim.x * 4 * UNR) {
    u32x4 ya[UNR], yb[UNR], ga[UNR], gb[UNR];
    long long src[UNR];                                  // element offset of this lane's 16 channels in dx, -1 = padded row / idle
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int row = base + u;
      ya[u] = u32x4{0u, 0u, 0u, 0u};
      yb[u] = ya[u]; ga[u] = ya[u]; gb[u] = ya[u];
      src[u] = -1;
      if (act && row < N2) {
        const int w2 = row % W2, h2 = (row / W2) % H2, z = row / (W2 * H2);
        const int h = 2 * h2 + (quad >> 1), w = 2 * w2 + (quad & 1);
        const u16* gp = dout + (size_t)row * C4 + lane * 16;
        ga[u] = *reinterpret_cast<const u32x4*>(gp);
        gb[u] = *reinterpret_cast<const u32x4*>(gp + 8);
        if (h < H) {
          const size_t tok = (size_t)(z * H + h) * W + w;
          const u16* p = x + tok * ldx + c0;
          ya[u] = *reinterpret_cast<const u32x4*>(p);
          yb[u] = *reinterpret_cast<const u32x4*>(p + 8);
          src[u] = (long long)(tok * C + c0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float y[16], g[16];
      unpack8(ya[u], y); unpack8(yb[u], y + 8);
      unpack8(ga[u], g); unpack8(gb[u], g + 8);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; c += 4) s += (y[c] + y[c + 1]) + (y[c + 2] + y[c + 3]);
      const float mean = wave_sum(s) * inv_c;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { const float d = y[c] - mean; q += d * d; }
      if (!act) q = 0.f;
      const float rstd = rsqrtf(wave_sum(q) * inv_c + LN_EPS);
      float gg[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        y[c] = act ? (y[c] - mean) * rstd : 0.f;         // xhat (a padded source token holds zeros, i.e. xhat = -mean * rstd: it is part of the row)
        gg[c] = g[c] * gm[c];
        dg[c] += g[c] * y[c];
        db[c] += g[c];
        s1 += gg[c];
        s2 += gg[c] * y[c];
      }
      const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
      if (src[u] >= 0) {
        u32x4 o0, o1;
        float ad[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) ad[c] = 0.f;
        if (add) {                                       // a second gradient of the same tokens (the skip connection's): summed here
          unpack8(*reinterpret_cast<const u32x4*>(add + src[u]), ad);
          unpack8(*reinterpret_cast<const u32x4*>(add + src[u] + 8), ad + 8);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o0[c] = pack_bf16x2((gg[2 * c] - m1 - y[2 * c] * m2) * rstd + ad[2 * c], (gg[2 * c + 1] - m1 - y[2 * c + 1] * m2) * rstd + ad[2 * c + 1]);
          o1[c] = pack_bf16x2((gg[8 + 2 * c] - m1 - y[8 + 2 * c] * m2) * rstd + ad[8 + 2 * c],
                              (gg[9 + 2 * c] - m1 - y[9 + 2 * c] * m2) * rstd + ad[9 + 2 * c]);
        }
        *reinterpret_cast<u32x4*>(dx + src[u]) = o0;
        *reinterpret_cast<u32x4*>(dx + src[u] + 8) = o1;
      }
    }
  }
  // per-lane parameter gradients -> [4 waves][4C] through LDS, one atomic per channel and workgroup
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    red[wave * 1024 + lane * 16 + c] = dg[c];
    red[4096 + wave * 1024 + lane * 16 + c] = db[c];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C4; c += 256) {
    atomicAdd(&dgamma[c], (red[c] + red[1024 + c]) + (red[2048 + c] + red[3072 + c]));
    atomicAdd(&dbeta[c], (red[4096 + c] + red[5120 + c]) + (red[6144 + c] + red[7168 + c]));
  }
}

// Up-sampling backward: LPR = 32 lanes per fine row of Co <= 256 channels, two rows per wave, UNR row pairs in flight.  The
// iteration space includes the cropped fine rows h >= H (reference layers.py:488-489): their quarter of dy receives zeros.
template <int UNR>
__global__ __launch_bounds__(256) void upsample_ln_bwd_bf16_v8_kernel(const u16* __restrict__ dout, const u16* __restrict__ yin,
                                                                      const float* __restrict__ gamma, u16* __restrict__ dy,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      int Z, int H2, int W2, int H, int Co) {
  constexpr int LPR = 32, RPW = 2, GROUPS = 8, CW = LPR * 8;
  __shared__ float red[2 * GROUPS * CW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const bool act = l * 8 < Co;
  const int Wf = 2 * W2, Hf = 2 * H2, Nall = Z * Hf * Wf;
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    gm[c] = act ? gamma[l * 8 + c] : 0.f;
    dg[c] = 0.f;
    db[c] = 0.f;
  }
  const float inv_c = 1.0f / Co;
  for (int base = (blockIdx.x * 4 + wave) * RPW * UNR; base < Nall; base += gridDim.x * 4 * RPW * UNR) {
    u32x4 yv[UNR], gv[UNR];
    long long src[UNR];
    bool real[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = base + u * RPW + sub;
      yv[u] = u32x4{0u, 0u, 0u, 0u};
      gv[u] = yv[u];
      src[u] = -1;
      real[u] = false;
      if (act && r < Nall) {
        const int w = r % Wf, h = (r / Wf) % Hf, z = r / (Wf * Hf);
        src[u] = (long long)(((size_t)(z * H2 + (h >> 1)) * W2 + (w >> 1)) * (4 * Co) + ((h & 1) * 2 + (w & 1)) * Co + l * 8);
        real[u] = h < H;
        if (real[u]) {
          yv[u] = *reinterpret_cast<const u32x4*>(yin + src[u]);
          gv[u] = *reinterpret_cast<const u32x4*>(dout + (((size_t)z * H + h) * Wf + w) * Co + l * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float y[8], g[8];
      unpack8(yv[u], y);
      unpack8(gv[u], g);
      const float s = group_sum<LPR>(((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7])));
      const float mean = s * inv_c;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) { const float d = y[c] - mean; q += d * d; }
      if (!act) q = 0.f;
      const float rstd = rsqrtf(group_sum<LPR>(q) * inv_c + LN_EPS);
      float gg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        y[c] = (act && real[u]) ? (y[c] - mean) * rstd : 0.f;
        gg[c] = g[c] * gm[c];
        dg[c] += g[c] * y[c];
        db[c] += g[c];
        s1 += gg[c];
        s2 += gg[c] * y[c];
      }
      const float m1 = group_sum<LPR>(s1) * inv_c, m2 = group_sum<LPR>(s2) * inv_c;
      if (src[u] >= 0) {
        u32x4 o = {0u, 0u, 0u, 0u};
        if (real[u]) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            o[c] = pack_bf16x2((gg[2 * c] - m1 - y[2 * c] * m2) * rstd, (gg[2 * c + 1] - m1 - y[2 * c + 1] * m2) * rstd);
        }
        *reinterpret_cast<u32x4*>(dy + src[u]) = o;
      }
    }
  }
  const int grp = wave * RPW + sub;
  if (act) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      red[grp * CW + l * 8 + c] = dg[c];
      red[(GROUPS + grp) * CW + l * 8 + c] = db[c];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Co; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int gq = 0; gq < GROUPS; ++gq) {
      a += red[gq * CW + c];
      b += red[(GROUPS + gq) * CW + c];
    }
    atomicAdd(&dgamma[c], a);
    atomicAdd(&dbeta[c], b);
  }
}

template <int NV, typename T>
__global__ __launch_bounds__(256) void downsample_ln_bwd_kernel(const T* __restrict__ dout,
                                                                const T* __restrict__ x, int ldx,
                                                                const float* __restrict__ gamma, T* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int Z, int H, int W, int C, const T* __restrict__ add) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H2 = (H + 1) / 2, W2 = W / 2, C4 = 4 * C, nvec = C4 >> 2, cvec = C >> 2;
  const int N2 = Z * H2 * W2;
  f32x4 dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = dg[i]; }
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < N2; row += gridDim.x * ROWS_PER_BLOCK) {
    const int w2 = row % W2, h2 = (row / W2) % H2, z = row / (W2 * H2);
    f32x4 y[NV], g[NV];
    size_t src[NV];
    bool real[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = lane + 64 * i;
      if (f < nvec) {
        const int quad = f / cvec, c4 = f - quad * cvec;
        const int h = 2 * h2 + (quad >> 1), w = 2 * w2 + (quad & 1);
        real[i] = h < H;
        src[i] = (size_t)(z * H + h) * W + w;
        y[i] = real[i] ? ld4(x + src[i] * ldx + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        g[i] = ld4(dout + (size_t)row * C4 + 4 * f);
        src[i] = src[i] * C + 4 * c4;
      }
    }
    row_ln_bwd<NV>(y, g, nvec, lane, C4, gamma, dg, db);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec && real[i]) st4(dx + src[i], add ? y[i] + ld4(add + src[i]) : y[i]);
  }
  flush_param_grads<NV>(dg, db, nvec, lane, wave, dgamma, dbeta, red);
}

template <int NV, typename T>
__global__ __launch_bounds__(256) void upsample_ln_bwd_kernel(const T* __restrict__ dout,
                                                              const T* __restrict__ yin,
                                                              const float* __restrict__ gamma, T* __restrict__ dy,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int Z, int H2, int W2, int H, int Co) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Wf = 2 * W2, Hf = 2 * H2, nvec = Co >> 2;
  const int Nall = Z * Hf * Wf;         // includes the cropped rows h >= H, which receive zero gradient
  f32x4 dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { dg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; db[i] = dg[i]; }
  for (int r = blockIdx.x * ROWS_PER_BLOCK + wave; r < Nall; r += gridDim.x * ROWS_PER_BLOCK) {
    const int w = r % Wf, h = (r / Wf) % Hf, z = r / (Wf * Hf);
    const size_t src = ((size_t)(z * H2 + (h >> 1)) * W2 + (w >> 1)) * (4 * Co) + ((h & 1) * 2 + (w & 1)) * Co;
    f32x4 y[NV], g[NV];
    if (h >= H) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nvec) st4(dy + src + 4 * (lane + 64 * i), f32x4{0.f, 0.f, 0.f, 0.f});
      continue;
    }
    const size_t orow = ((size_t)z * H + h) * Wf + w;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) {
        y[i] = ld4(yin + src + 4 * (lane + 64 * i));
        g[i] = ld4(dout + orow * Co + 4 * (lane + 64 * i));
      }
    row_ln_bwd<NV>(y, g, nvec, lane, Co, gamma, dg, db);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) st4(dy + src + 4 * (lane + 64 * i), y[i]);
  }
  flush_param_grads<NV>(dg, db, nvec, lane, wave, dgamma, dbeta, red);
}

// gradient of patch_recover_scatter: gather the field gradients back into GEMM-output layout
constexpr int EMB_TOK = 64;

// One block = 64 tokens of one (z-slab, latitude-group): for every output column (variable, pz, ph) the token's four pw values are
// four CONSECUTIVE longitudes of one plane row -- one 16-B load per (column run, token), one 16-B LDS write (tile rows of 164
// floats: 16-B aligned, and the 8-lane groups of a ds_write_b128 land on all 32 banks) -- and the transposed tile leaves as 8
// columns per thread (16-B stores in bf16).  (Round 3: the scalar form of this kernel ran at 1.7 TB/s.)
constexpr int GB_LD = 164;

template <typename T>
__global__ __launch_bounds__(256) void patch_recover_gather_bwd_kernel(const float* __restrict__ d_output,
                                                                       const float* __restrict__ d_output_surface,
                                                                       T* __restrict__ dy_upper,
                                                                       T* __restrict__ dy_surface, int LAT, int LON,
                                                                       int H4, int W4, int chunks) {
  __shared__ __attribute__((aligned(16))) float tile[EMB_TOK * GB_LD];
  const int chunk = blockIdx.x % chunks, h4 = (blockIdx.x / chunks) % H4, zp = blockIdx.x / (chunks * H4);
  const int w0 = chunk * EMB_TOK;
  const int ntok = min(EMB_TOK, W4 - w0);
  const int tid = threadIdx.x;
  const int ncol = zp == 0 ? 64 : 160;
  const size_t plane = (size_t)LAT * LON;
  const int nrun = ncol / 4;
  const int tk = tid & 63;
  for (int run = tid >> 6; run < nrun; run += 4) {
    int v, pz, ph;
    if (zp == 0) { v = run >> 2; pz = 0; ph = run & 3; } else { v = run >> 3; pz = (run >> 2) & 1; ph = run & 3; }
    const int lat = 4 * h4 + ph;
    const int lev = 2 * (zp - 1) + pz;
    const bool valid = lat < LAT && (zp == 0 || lev < 13) && tk < ntok;
    const float* src = zp == 0 ? d_output_surface + v * plane + (size_t)lat * LON
                               : d_output + ((size_t)v * 13 + lev) * plane + (size_t)lat * LON;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (valid) val = *reinterpret_cast<const f32x4*>(src + 4 * (w0 + tk));
    *reinterpret_cast<f32x4*>(&tile[tk * GB_LD + run * 4]) = val;
  }
  __syncthreads();
  T* dst = zp == 0 ? dy_surface + ((size_t)h4 * W4 + w0) * 64 : dy_upper + (((size_t)(zp - 1) * H4 + h4) * W4 + w0) * 160;
  const int cpr = ncol / 8;                                // 8-column pieces per token row
  for (int i = tid; i < ntok * cpr; i += 256) {
    const int t = i / cpr, c8 = (i - t * cpr) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(&tile[t * GB_LD + c8]);
    const f32x4 b = *reinterpret_cast<const f32x4*>(&tile[t * GB_LD + c8 + 4]);
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<u32x4*>(dst + (size_t)t * ncol + c8) =
          u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    } else {
      *reinterpret_cast<f32x4*>(dst + (size_t)t * ncol + c8) = a;
      *reinterpret_cast<f32x4*>(dst + (size_t)t * ncol + c8 + 4) = b;
    }
  }
}

int row_grid(int rows) {
  int blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return blocks < 2048 ? blocks : 2048;        // 8 workgroups per CU: fewer, fatter register accumulators
}

}  // namespace

#define PANGU_NV_DISPATCH(C_, KERNEL, T, ...)                                                                        \
  do {                                                                                                               \
    const size_t shm = (size_t)8 * (C_) * sizeof(float);                                                             \
    if ((C_) <= 256) hipLaunchKernelGGL((KERNEL<1, T>), g, b, shm, s, __VA_ARGS__);                                  \
    else if ((C_) <= 512) hipLaunchKernelGGL((KERNEL<2, T>), g, b, shm, s, __VA_ARGS__);                             \
    else hipLaunchKernelGGL((KERNEL<4, T>), g, b, shm, s, __VA_ARGS__);                                              \
  } while (0)

extern "C" int pangu_ln_residual_bwd(pangu_stream_t stream, const float* dout, int lddo, const float* y,
                                     const float* gamma, float* dy, float* dgamma, float* dbeta, int N, int C,
                                     float branch_scale) {
  if (!dout || !y || !gamma || !dy || !dgamma || !dbeta) return PANGU_E_NULL;
  if (N <= 0 || C <= 0 || (C & 3) || C > 1024 || lddo < C || (lddo & 3)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(N)), b(256);
  PANGU_NV_DISPATCH(C, ln_residual_bwd_kernel, float, dout, lddo, y, gamma, dy, dgamma, dbeta, N, C, branch_scale);
  return pangu_launch_status();
}

extern "C" int pangu_downsample_ln_bwd(pangu_stream_t stream, const float* dout, const float* x, int ldx,
                                       const float* gamma, float* dx, float* dgamma, float* dbeta, int Z, int H, int W,
                                       int C, const float* dx_add) {
  if (!dout || !x || !gamma || !dx || !dgamma || !dbeta) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || (W & 1) || (C & 3) || 4 * C > 1024 || ldx < C || (ldx & 3)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(Z * ((H + 1) / 2) * (W / 2))), b(256);
  PANGU_NV_DISPATCH(4 * C, downsample_ln_bwd_kernel, float, dout, x, ldx, gamma, dx, dgamma, dbeta, Z, H, W, C, dx_add);
  return pangu_launch_status();
}

extern "C" int pangu_upsample_ln_bwd(pangu_stream_t stream, const float* dout, const float* y, const float* gamma,
                                     float* dy, float* dgamma, float* dbeta, int Z, int H2, int W2, int H, int Co) {
  if (!dout || !y || !gamma || !dy || !dgamma || !dbeta) return PANGU_E_NULL;
  if (Z <= 0 || H2 <= 0 || W2 <= 0 || H <= 0 || H > 2 * H2 || (Co & 3) || Co > 1024) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(Z * 2 * H2 * 2 * W2)), b(256);
  PANGU_NV_DISPATCH(Co, upsample_ln_bwd_kernel, float, dout, y, gamma, dy, dgamma, dbeta, Z, H2, W2, H, Co);
  return pangu_launch_status();
}

extern "C" int pangu_patch_recover_gather_bwd(pangu_stream_t stream, const float* d_output,
                                              const float* d_output_surface, float* dy_upper, float* dy_surface,
                                              int LAT, int LON) {
  if (!d_output || !d_output_surface || !dy_upper || !dy_surface) return PANGU_E_NULL;
  if (LAT <= 0 || LON <= 0 || (LON & 3)) return PANGU_E_SHAPE;
  const int H4 = (LAT + 3) / 4, W4 = LON / 4, chunks = (W4 + EMB_TOK - 1) / EMB_TOK;
  hipLaunchKernelGGL(patch_recover_gather_bwd_kernel<float>, dim3(8 * H4 * chunks), dim3(256), 0, (hipStream_t)stream, d_output,
                     d_output_surface, dy_upper, dy_surface, LAT, LON, H4, W4, chunks);
  return pangu_launch_status();
}

// ---- bf16 I/O variants (dout / saved activations / outputs bf16; statistics, dgamma, dbeta fp32) ----------------

extern "C" int pangu_ln_residual_bwd_bf16(pangu_stream_t stream, const void* dout, int lddo, const void* y,
                                          const float* gamma, void* dy, float* dgamma, float* dbeta, int N, int C,
                                          float branch_scale) {
  if (!dout || !y || !gamma || !dy || !dgamma || !dbeta) return PANGU_E_NULL;
  if (N <= 0 || C <= 0 || (C & 3) || C > 1024 || lddo < C || (lddo & 3)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(N)), b(256);
  if ((C & 7) == 0 && C <= 512 && (lddo & 7) == 0) {
    constexpr int unr = 2;      // row groups in flight (4 measured level)
#define PANGU_LNB(LPR_, UNR_)                                                                                              \
  do {                                                                                                                    \
    const int rpb = 4 * (64 / LPR_) * UNR_, blocks = (N + rpb - 1) / rpb;                                                 \
    hipLaunchKernelGGL((ln_residual_bwd_bf16_v8_kernel<LPR_, UNR_>), dim3(blocks < 2048 ? blocks : 2048), b, 0, s,        \
                       (const u16*)dout, lddo, (const u16*)y, gamma, (u16*)dy, dgamma, dbeta, N, C, branch_scale);        \
  } while (0)
    constexpr bool v16 = true;      // 16 channels per lane, two rows per wave at C = 384 (8 per lane: +0.2 ms per training step)
    if (C <= 256) {
      PANGU_LNB(32, unr);
    } else if (v16 && (C & 15) == 0 && (lddo & 7) == 0) {
      const int blocks = (N + 15) / 16;
      hipLaunchKernelGGL(ln_residual_bwd_bf16_v16_kernel<2>, dim3(blocks < 2048 ? blocks : 2048), b, 0, s, (const u16*)dout, lddo,
                         (const u16*)y, gamma, (u16*)dy, dgamma, dbeta, N, C, branch_scale);
    } else {
      PANGU_LNB(64, unr);
    }
#undef PANGU_LNB
    return pangu_launch_status();
  }
  PANGU_NV_DISPATCH(C, ln_residual_bwd_kernel, u16, (const u16*)dout, lddo, (const u16*)y, gamma, (u16*)dy, dgamma, dbeta, N,
                    C, branch_scale);
  return pangu_launch_status();
}

extern "C" int pangu_downsample_ln_bwd_bf16(pangu_stream_t stream, const void* dout, const void* x, int ldx,
                                            const float* gamma, void* dx, float* dgamma, float* dbeta, int Z, int H, int W,
                                            int C, const void* dx_add) {
  if (!dout || !x || !gamma || !dx || !dgamma || !dbeta) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || (W & 1) || (C & 3) || 4 * C > 1024 || ldx < C || (ldx & 3)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(Z * ((H + 1) / 2) * (W / 2))), b(256);
  constexpr bool fast = true;      // 16-B fast path where the shape allows (the generic one-row-per-wave kernel below otherwise)
  if (fast && (C & 15) == 0 && C <= 256 && (ldx & 7) == 0) {
    const int rows = Z * ((H + 1) / 2) * (W / 2), blocks = (rows + 7) / 8;
    hipLaunchKernelGGL(downsample_ln_bwd_bf16_v16_kernel<2>, dim3(blocks < 2048 ? blocks : 2048), b, 0, s, (const u16*)dout,
                       (const u16*)x, ldx, gamma, (u16*)dx, dgamma, dbeta, Z, H, W, C, (const u16*)dx_add);
    return pangu_launch_status();
  }
  PANGU_NV_DISPATCH(4 * C, downsample_ln_bwd_kernel, u16, (const u16*)dout, (const u16*)x, ldx, gamma, (u16*)dx, dgamma,
                    dbeta, Z, H, W, C, (const u16*)dx_add);
  return pangu_launch_status();
}

extern "C" int pangu_upsample_ln_bwd_bf16(pangu_stream_t stream, const void* dout, const void* y, const float* gamma,
                                          void* dy, float* dgamma, float* dbeta, int Z, int H2, int W2, int H, int Co) {
  if (!dout || !y || !gamma || !dy || !dgamma || !dbeta) return PANGU_E_NULL;
  if (Z <= 0 || H2 <= 0 || W2 <= 0 || H <= 0 || H > 2 * H2 || (Co & 3) || Co > 1024) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(Z * 2 * H2 * 2 * W2)), b(256);
  constexpr bool fast = true;      // 16-B fast path where the shape allows (the generic one-row-per-wave kernel below otherwise)
  if (fast && (Co & 7) == 0 && Co <= 256) {
    const int rows = Z * 2 * H2 * 2 * W2, blocks = (rows + 15) / 16;
    hipLaunchKernelGGL(upsample_ln_bwd_bf16_v8_kernel<2>, dim3(blocks < 2048 ? blocks : 2048), b, 0, s, (const u16*)dout,
                       (const u16*)y, gamma, (u16*)dy, dgamma, dbeta, Z, H2, W2, H, Co);
    return pangu_launch_status();
  }
  PANGU_NV_DISPATCH(Co, upsample_ln_bwd_kernel, u16, (const u16*)dout, (const u16*)y, gamma, (u16*)dy, dgamma, dbeta, Z, H2,
                    W2, H, Co);
  return pangu_launch_status();
}

extern "C" int pangu_patch_recover_gather_bwd_bf16(pangu_stream_t stream, const float* d_output,
                                                   const float* d_output_surface, void* dy_upper, void* dy_surface,
                                                   int LAT, int LON) {
  if (!d_output || !d_output_surface || !dy_upper || !dy_surface) return PANGU_E_NULL;
  if (LAT <= 0 || LON <= 0 || (LON & 3)) return PANGU_E_SHAPE;
  const int H4 = (LAT + 3) / 4, W4 = LON / 4, chunks = (W4 + EMB_TOK - 1) / EMB_TOK;
  hipLaunchKernelGGL(patch_recover_gather_bwd_kernel<u16>, dim3(8 * H4 * chunks), dim3(256), 0, (hipStream_t)stream,
                     d_output, d_output_surface, (u16*)dy_upper, (u16*)dy_surface, LAT, LON, H4, W4, chunks);
  return pangu_launch_status();
}
