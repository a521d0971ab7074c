// HBM-bound row kernels of the Pangu-Weather path (fp32): post-norm residual, down/up-sample
// gather + LayerNorm, patch-embed gather, patch-recover scatter.  One wave per token row, float4 lanes,
// wave-shuffle reductions; all permute/pad/crop steps of the reference are address arithmetic.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int ROWS_PER_BLOCK = 4;   // 4 waves of 64

// LayerNorm of one row held as NV float4 per lane (lane i owns float4 i, i+64, ..); C = 4*nvec valid float4.
template <int NV>
__device__ inline void row_layernorm(f32x4 (&v)[NV], int nvec, int lane, int C, const float* gamma,
                                     const float* beta, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float d = v[i][c] - mean; q += d * d; }
    }
  rstd = rsqrtf(wave_sum(q) / C + LN_EPS);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * (lane + 64 * i));
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * (lane + 64 * i));
      v[i] = (v[i] - mean) * rstd * gm + bt;
    }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_residual_kernel(const float* __restrict__ y,
                                                          const float* __restrict__ shortcut, int lds,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          int ldo, float* __restrict__ mean_rstd, int N, int C,
                                                          float branch_scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 2;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < N; row += gridDim.x * ROWS_PER_BLOCK) {
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) v[i] = *reinterpret_cast<const f32x4*>(y + (size_t)row * C + 4 * (lane + 64 * i));
    float mean, rstd;
    row_layernorm<NV>(v, nvec, lane, C, gamma, beta, mean, rstd);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(shortcut + (size_t)row * lds + 4 * (lane + 64 * i));
        *reinterpret_cast<f32x4*>(out + (size_t)row * ldo + 4 * (lane + 64 * i)) = sc + branch_scale * v[i];
      }
    if (mean_rstd && lane == 0) { mean_rstd[2 * (size_t)row] = mean; mean_rstd[2 * (size_t)row + 1] = rstd; }
  }
}

// DownSample: out row (z,h2,w2) = LN( [x(z,2h2,2w2), x(z,2h2,2w2+1), x(z,2h2+1,2w2), x(z,2h2+1,2w2+1)] ), 4C wide
template <int NV>
__global__ __launch_bounds__(256) void downsample_ln_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            float* __restrict__ mean_rstd, int Z, int H, int W, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H2 = (H + 1) / 2, W2 = W / 2, C4 = 4 * C, nvec = C4 >> 2, cvec = C >> 2;
  const int N2 = Z * H2 * W2;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < N2; row += gridDim.x * ROWS_PER_BLOCK) {
    const int w2 = row % W2, h2 = (row / W2) % H2, z = row / (W2 * H2);
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = lane + 64 * i;
      if (f < nvec) {
        const int quad = f / cvec, c4 = f - quad * cvec;     // quad = dh*2 + dw
        const int h = 2 * h2 + (quad >> 1), w = 2 * w2 + (quad & 1);
        v[i] = h < H ? *reinterpret_cast<const f32x4*>(x + ((size_t)(z * H + h) * W + w) * ldx + 4 * c4)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    float mean, rstd;
    row_layernorm<NV>(v, nvec, lane, C4, gamma, beta, mean, rstd);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) *reinterpret_cast<f32x4*>(out + (size_t)row * C4 + 4 * (lane + 64 * i)) = v[i];
    if (mean_rstd && lane == 0) { mean_rstd[2 * (size_t)row] = mean; mean_rstd[2 * (size_t)row + 1] = rstd; }
  }
}

// UpSample: out token (z, h, w) (h < H, w < 2*W2) = LN( y[(z, h/2, w/2)][ (h&1)*2Co + (w&1)*Co + 0..Co ) )
template <int NV>
__global__ __launch_bounds__(256) void upsample_ln_kernel(const float* __restrict__ y,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          float* __restrict__ mean_rstd, int Z, int H2, int W2, int H,
                                                          int Co) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Wf = 2 * W2, nvec = Co >> 2;
  const int N = Z * H * Wf;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < N; row += gridDim.x * ROWS_PER_BLOCK) {
    const int w = row % Wf, h = (row / Wf) % H, z = row / (Wf * H);
    const float* src = y + ((size_t)(z * H2 + (h >> 1)) * W2 + (w >> 1)) * (4 * Co) + ((h & 1) * 2 + (w & 1)) * Co;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) v[i] = *reinterpret_cast<const f32x4*>(src + 4 * (lane + 64 * i));
    float mean, rstd;
    row_layernorm<NV>(v, nvec, lane, Co, gamma, beta, mean, rstd);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) *reinterpret_cast<f32x4*>(out + (size_t)row * Co + 4 * (lane + 64 * i)) = v[i];
    if (mean_rstd && lane == 0) { mean_rstd[2 * (size_t)row] = mean; mean_rstd[2 * (size_t)row + 1] = rstd; }
  }
}

// ---- patch embed gather --------------------------------------------------------------------------------
// One workgroup per (plane zp, patch row h4, chunk of 64 patch columns).  zp = 0: surface (7 ch x 4 x 4 = 112
// columns), zp = 1..7: upper-air level pair (6 ch x 2 x 4 x 4 = 192 columns).  Reads are 1 KB contiguous
// longitude runs per (channel, level, latitude); rows are assembled in LDS and written out whole.
constexpr int EMB_TOK = 64;

__global__ __launch_bounds__(256) void patch_embed_gather_kernel(
    const float* __restrict__ input, const float* __restrict__ input_surface, const float* __restrict__ s_mean,
    const float* __restrict__ s_std, const float* __restrict__ u_mean, const float* __restrict__ u_std,
    const float* __restrict__ maps, const float* __restrict__ const_h, float* __restrict__ a_surface,
    float* __restrict__ a_upper, int LAT, int LON, int H4, int W4, int chunks, int levels_reversed) {
  constexpr int TLD = 196;               // tile row stride in floats: 16-B aligned, 8 tokens x 4 dwords cover the 32 banks
  __shared__ __attribute__((aligned(16))) float tile[EMB_TOK * TLD];
  const int chunk = blockIdx.x % chunks, h4 = (blockIdx.x / chunks) % H4, zp = blockIdx.x / (chunks * H4);
  const int w0 = chunk * EMB_TOK;
  const int ntok = min(EMB_TOK, W4 - w0);
  const int tid = threadIdx.x;
  const int ncol = zp == 0 ? 112 : 192;
  const int nrun = ncol / 4;            // (c, [pz,] ph) combinations; each is a run of ntok float4 (the 4 longitudes of a token)
  const size_t plane = (size_t)LAT * LON;
  for (int run = tid >> 6; run < nrun; run += 4) {
    // decode run -> channel c, level offset pz, lat offset ph
    int c, pz, ph;
    if (zp == 0) { c = run >> 2; pz = 0; ph = run & 3; } else { c = run >> 3; pz = (run >> 2) & 1; ph = run & 3; }
    const int lat = 4 * h4 + ph;
    const float* src = nullptr;
    float mean = 0.f, sd = 1.f;
    bool valid = lat < LAT;
    if (zp == 0) {
      if (c < 4) { src = input_surface + c * plane + (size_t)lat * LON; mean = s_mean[c]; sd = s_std[c]; }
      else { src = maps + (size_t)(c - 4) * (4 * H4) * LON + (size_t)lat * LON; valid = true; }   // maps are pre-padded
    } else {
      const int lev = 2 * (zp - 1) + pz;
      valid = valid && lev < 13;
      if (valid) {
        if (c < 5) {
          // levels_reversed: the field is stored as the reader finds it on disk (level axis ascending) and the reversal of
          // reference era5_data/utils_data.py:117 is this address -- logical level `lev` lives in plane 12 - lev
          src = input + ((size_t)c * 13 + (levels_reversed ? 12 - lev : lev)) * plane + (size_t)lat * LON;
          mean = u_mean[(12 - lev) * 5 + c]; sd = u_std[(12 - lev) * 5 + c];
        } else {
          src = const_h + (size_t)lev * plane + (size_t)lat * LON;
        }
      }
    }
    const bool norm = (zp == 0) ? (c < 4) : (c < 5);
    for (int tk = (tid & 63); tk < ntok; tk += 64) {      // LON % 4 == 0: a token's 4 longitudes are one aligned float4
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        v = *reinterpret_cast<const f32x4*>(src + 4 * (w0 + tk));
        if (norm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) / sd;
        }
      }
      *reinterpret_cast<f32x4*>(&tile[tk * TLD + run * 4]) = v;
    }
  }
  __syncthreads();
  float* dst = zp == 0 ? a_surface + ((size_t)h4 * W4 + w0) * 112
                       : a_upper + (((size_t)(zp - 1) * H4 + h4) * W4 + w0) * 192;
  const int c4n = ncol / 4;
  for (int i = tid; i < ntok * c4n; i += 256) {
    const int tk = i / c4n, c4 = i - tk * c4n;
    reinterpret_cast<f32x4*>(dst)[i] = *reinterpret_cast<const f32x4*>(&tile[tk * TLD + c4 * 4]);
  }
}

// ---- patch embed gather, adjoint w.r.t. the raw fields ---------------------------------------------------------
// d_input[c][lev][lat][lon] = da_upper[token][c*32 + pz*16 + ph*4 + pw] / upper_std(lev, c) and the surface twin: the transpose of
// the gather above for the columns that came from `input` / `input_surface` -- the caller forms only those columns of dA (the
// first 160 of 192 / 64 of 112: the constant maps, const_h and the zero padding have no field behind them) -- divided by the
// std the forward divided by (autograd of `(x - mean) / std`, reference models/layers.py:48-55,71-76).  One workgroup per (plane zp, patch row h4, chunk of 64 patch columns), like the
// gather; rows of da are read whole into LDS, the fields are written in 1 KB longitude runs.
__global__ __launch_bounds__(256) void patch_embed_gather_bwd_kernel(
    const float* __restrict__ da_surface, const float* __restrict__ da_upper, const float* __restrict__ s_std,
    const float* __restrict__ u_std, float* __restrict__ d_input, float* __restrict__ d_input_surface, int LAT, int LON,
    int H4, int W4, int chunks, int levels_reversed) {
  __shared__ float tile[EMB_TOK * 161];
  const int chunk = blockIdx.x % chunks, h4 = (blockIdx.x / chunks) % H4, zp = blockIdx.x / (chunks * H4);
  const int w0 = chunk * EMB_TOK;
  const int ntok = min(EMB_TOK, W4 - w0);
  const int tid = threadIdx.x;
  const int ncol = zp == 0 ? 64 : 160;        // the A-matrix columns that came from the fields (4 x 16 of 112, 5 x 32 of 192)
  const float* src = zp == 0 ? da_surface + ((size_t)h4 * W4 + w0) * 64
                             : da_upper + (((size_t)(zp - 1) * H4 + h4) * W4 + w0) * 160;
  for (int i = tid; i < ntok * ncol; i += 256) {
    const int tk = i / ncol, col = i - tk * ncol;
    tile[tk * 161 + col] = src[i];
  }
  __syncthreads();
  const size_t plane = (size_t)LAT * LON;
  const int nrun = ncol / 4;
  for (int run = tid >> 6; run < nrun; run += 4) {
    int v, pz, ph;
    if (zp == 0) { v = run >> 2; pz = 0; ph = run & 3; } else { v = run >> 3; pz = (run >> 2) & 1; ph = run & 3; }
    const int lat = 4 * h4 + ph;
    if (lat >= LAT) continue;
    float* dst;
    float sd;
    if (zp == 0) {
      dst = d_input_surface + v * plane + (size_t)lat * LON;
      sd = s_std[v];
    } else {
      const int lev = 2 * (zp - 1) + pz;
      if (lev >= 13) continue;
      dst = d_input + ((size_t)v * 13 + (levels_reversed ? 12 - lev : lev)) * plane + (size_t)lat * LON;
      sd = u_std[(12 - lev) * 5 + v];
    }
    for (int i = (tid & 63); i < 4 * ntok; i += 64) dst[4 * w0 + i] = tile[(i >> 2) * 161 + run * 4 + (i & 3)] / sd;
  }
}

// ---- patch recover scatter -------------------------------------------------------------------------------
// DENORM (rollout, reference era5_data/utils_data.py:324-330 `normBackData` folded in): every element also leaves in physical
// units, phys = out * std + mean of its (variable, level) plane -- the same two roundings as the reference's expression
// (a multiply, then an add: no fused multiply-add) -- into a second pair of fields: the next step's input buffers.
template <bool DENORM>
__global__ __launch_bounds__(256) void patch_recover_scatter_kernel(const float* __restrict__ y_upper,
                                                                    const float* __restrict__ y_surface,
                                                                    float* __restrict__ output,
                                                                    float* __restrict__ output_surface, int LAT,
                                                                    int LON, int H4, int W4, int chunks,
                                                                    float* __restrict__ phys, float* __restrict__ phys_surface,
                                                                    const float* __restrict__ u_mean,
                                                                    const float* __restrict__ u_std,
                                                                    const float* __restrict__ s_mean,
                                                                    const float* __restrict__ s_std) {
  __shared__ float tile[EMB_TOK * 161];
  const int chunk = blockIdx.x % chunks, h4 = (blockIdx.x / chunks) % H4, zp = blockIdx.x / (chunks * H4);
  const int w0 = chunk * EMB_TOK;
  const int ntok = min(EMB_TOK, W4 - w0);
  const int tid = threadIdx.x;
  const int ncol = zp == 0 ? 64 : 160;
  const float* src = zp == 0 ? y_surface + ((size_t)h4 * W4 + w0) * 64
                             : y_upper + (((size_t)(zp - 1) * H4 + h4) * W4 + w0) * 160;
  for (int i = tid; i < ntok * ncol; i += 256) {
    const int tk = i / ncol, col = i - tk * ncol;
    tile[tk * 161 + col] = src[i];
  }
  __syncthreads();
  const size_t plane = (size_t)LAT * LON;
  const int nrun = ncol / 4;
  for (int run = tid >> 6; run < nrun; run += 4) {
    int v, pz, ph;
    if (zp == 0) { v = run >> 2; pz = 0; ph = run & 3; } else { v = run >> 3; pz = (run >> 2) & 1; ph = run & 3; }
    const int lat = 4 * h4 + ph;
    if (lat >= LAT) continue;
    float* dst;
    float* dst2 = nullptr;
    float mn = 0.f, sd = 1.f;
    if (zp == 0) {
      dst = output_surface + v * plane + (size_t)lat * LON;
      if (DENORM) { dst2 = phys_surface + v * plane + (size_t)lat * LON; mn = s_mean[v]; sd = s_std[v]; }
    } else {
      const int lev = 2 * (zp - 1) + pz;
      if (lev >= 13) continue;
      dst = output + ((size_t)v * 13 + lev) * plane + (size_t)lat * LON;
      if (DENORM) { dst2 = phys + ((size_t)v * 13 + lev) * plane + (size_t)lat * LON; mn = u_mean[v * 13 + lev]; sd = u_std[v * 13 + lev]; }
    }
    for (int i = (tid & 63); i < 4 * ntok; i += 64) {
      const float val = tile[(i >> 2) * 161 + run * 4 + (i & 3)];
      dst[4 * w0 + i] = val;
      if (DENORM) {
        float p = val * sd;
        asm volatile("" : "+v"(p));        // the product is rounded before the add, as torch's `out * std + mean` rounds it
        dst2[4 * w0 + i] = p + mn;         // (hipcc contracts even __fadd_rn(__fmul_rn(..)) into one v_fma_f32)
      }
    }
  }
}

int row_grid(int rows) {
  constexpr int cap = 8192;
  int blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return blocks < cap ? blocks : cap;
}

}  // namespace

extern "C" int pangu_ln_residual_fwd(pangu_stream_t stream, const float* y, const float* shortcut, int lds,
                                     const float* gamma, const float* beta, float* out, int ldo, float* mean_rstd,
                                     int N, int C, float branch_scale) {
  if (!y || !shortcut || !gamma || !beta || !out) return PANGU_E_NULL;
  if (N <= 0 || C <= 0 || (C & 3) || C > 1024 || lds < C || ldo < C || (lds & 3) || (ldo & 3)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(row_grid(N)), b(256);
  if (C <= 256) hipLaunchKernelGGL(ln_residual_kernel<1>, g, b, 0, s, y, shortcut, lds, gamma, beta, out, ldo, mean_rstd, N, C, branch_scale);
  else if (C <= 512) hipLaunchKernelGGL(ln_residual_kernel<2>, g, b, 0, s, y, shortcut, lds, gamma, beta, out, ldo, mean_rstd, N, C, branch_scale);
  else hipLaunchKernelGGL(ln_residual_kernel<4>, g, b, 0, s, y, shortcut, lds, gamma, beta, out, ldo, mean_rstd, N, C, branch_scale);
  return pangu_launch_status();
}

extern "C" int pangu_downsample_ln_fwd(pangu_stream_t stream, const float* x, int ldx, const float* gamma,
                                       const float* beta, float* out, float* mean_rstd, int Z, int H, int W, int C) {
  if (!x || !gamma || !beta || !out) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || (W & 1) || (C & 3) || 4 * C > 1024 || ldx < C || (ldx & 3)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int rows = Z * ((H + 1) / 2) * (W / 2);
  dim3 g(row_grid(rows)), b(256);
  if (4 * C <= 256) hipLaunchKernelGGL(downsample_ln_kernel<1>, g, b, 0, s, x, ldx, gamma, beta, out, mean_rstd, Z, H, W, C);
  else if (4 * C <= 512) hipLaunchKernelGGL(downsample_ln_kernel<2>, g, b, 0, s, x, ldx, gamma, beta, out, mean_rstd, Z, H, W, C);
  else hipLaunchKernelGGL(downsample_ln_kernel<4>, g, b, 0, s, x, ldx, gamma, beta, out, mean_rstd, Z, H, W, C);
  return pangu_launch_status();
}

extern "C" int pangu_upsample_ln_fwd(pangu_stream_t stream, const float* y, const float* gamma, const float* beta,
                                     float* out, float* mean_rstd, int Z, int H2, int W2, int H, int Co) {
  if (!y || !gamma || !beta || !out) return PANGU_E_NULL;
  if (Z <= 0 || H2 <= 0 || W2 <= 0 || H <= 0 || H > 2 * H2 || (Co & 3) || Co > 1024) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int rows = Z * H * 2 * W2;
  dim3 g(row_grid(rows)), b(256);
  if (Co <= 256) hipLaunchKernelGGL(upsample_ln_kernel<1>, g, b, 0, s, y, gamma, beta, out, mean_rstd, Z, H2, W2, H, Co);
  else if (Co <= 512) hipLaunchKernelGGL(upsample_ln_kernel<2>, g, b, 0, s, y, gamma, beta, out, mean_rstd, Z, H2, W2, H, Co);
  else hipLaunchKernelGGL(upsample_ln_kernel<4>, g, b, 0, s, y, gamma, beta, out, mean_rstd, Z, H2, W2, H, Co);
  return pangu_launch_status();
}

extern "C" int pangu_patch_embed_gather(pangu_stream_t stream, const float* input, const float* input_surface,
                                        const float* surface_mean, const float* surface_std,
                                        const float* upper_mean, const float* upper_std, const float* maps,
                                        const float* const_h, float* a_surface, float* a_upper, int LAT, int LON,
                                        int levels_reversed) {
  if (!input || !input_surface || !surface_mean || !surface_std || !upper_mean || !upper_std || !maps || !const_h ||
      !a_surface || !a_upper)
    return PANGU_E_NULL;
  if (LAT <= 0 || LON <= 0 || (LON & 3)) return PANGU_E_SHAPE;
  const int H4 = (LAT + 3) / 4, W4 = LON / 4, chunks = (W4 + EMB_TOK - 1) / EMB_TOK;
  hipLaunchKernelGGL(patch_embed_gather_kernel, dim3(8 * H4 * chunks), dim3(256), 0, (hipStream_t)stream, input,
                     input_surface, surface_mean, surface_std, upper_mean, upper_std, maps, const_h, a_surface, a_upper,
                     LAT, LON, H4, W4, chunks, levels_reversed != 0);
  return pangu_launch_status();
}

extern "C" int pangu_patch_embed_gather_bwd(pangu_stream_t stream, const float* da_surface, const float* da_upper,
                                            const float* surface_std, const float* upper_std, float* d_input,
                                            float* d_input_surface, int LAT, int LON, int levels_reversed) {
  if (!da_surface || !da_upper || !surface_std || !upper_std || !d_input || !d_input_surface) return PANGU_E_NULL;
  if (LAT <= 0 || LON <= 0 || (LON & 3)) return PANGU_E_SHAPE;
  const int H4 = (LAT + 3) / 4, W4 = LON / 4, chunks = (W4 + EMB_TOK - 1) / EMB_TOK;
  hipLaunchKernelGGL(patch_embed_gather_bwd_kernel, dim3(8 * H4 * chunks), dim3(256), 0, (hipStream_t)stream, da_surface,
                     da_upper, surface_std, upper_std, d_input, d_input_surface, LAT, LON, H4, W4, chunks, levels_reversed != 0);
  return pangu_launch_status();
}

extern "C" int pangu_patch_recover_scatter(pangu_stream_t stream, const float* y_upper, const float* y_surface,
                                           float* output, float* output_surface, int LAT, int LON) {
  if (!y_upper || !y_surface || !output || !output_surface) return PANGU_E_NULL;
  if (LAT <= 0 || LON <= 0 || (LON & 3)) return PANGU_E_SHAPE;
  const int H4 = (LAT + 3) / 4, W4 = LON / 4, chunks = (W4 + EMB_TOK - 1) / EMB_TOK;
  hipLaunchKernelGGL(patch_recover_scatter_kernel<false>, dim3(8 * H4 * chunks), dim3(256), 0, (hipStream_t)stream, y_upper,
                     y_surface, output, output_surface, LAT, LON, H4, W4, chunks, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr);
  return pangu_launch_status();
}

extern "C" int pangu_patch_recover_scatter_denorm(pangu_stream_t stream, const float* y_upper, const float* y_surface,
                                                  float* output, float* output_surface, float* phys, float* phys_surface,
                                                  const float* upper_mean, const float* upper_std, const float* surface_mean,
                                                  const float* surface_std, int LAT, int LON) {
  if (!y_upper || !y_surface || !output || !output_surface || !phys || !phys_surface || !upper_mean || !upper_std ||
      !surface_mean || !surface_std)
    return PANGU_E_NULL;
  if (LAT <= 0 || LON <= 0 || (LON & 3)) return PANGU_E_SHAPE;
  const int H4 = (LAT + 3) / 4, W4 = LON / 4, chunks = (W4 + EMB_TOK - 1) / EMB_TOK;
  hipLaunchKernelGGL(patch_recover_scatter_kernel<true>, dim3(8 * H4 * chunks), dim3(256), 0, (hipStream_t)stream, y_upper,
                     y_surface, output, output_surface, LAT, LON, H4, W4, chunks, phys, phys_surface, upper_mean, upper_std,
                     surface_mean, surface_std);
  return pangu_launch_status();
}

// ---- latitude-weighted evaluation sums (reference era5_data/score.py:92-105,123-135) ------------------------------
// For every (sample, channel) plane [H][W]: S0 = sum w (p-t)^2, S1 = sum w p t, S2 = sum w p^2, S3 = sum w t^2, with the
// per-latitude weight vector w[H] supplied by the caller.  One workgroup per (plane, 8-row slab); fp32 atomics.
namespace {
constexpr int SC_ROWS = 8;
__global__ __launch_bounds__(256) void lat_weighted_sums_kernel(const float* __restrict__ pred,
                                                                const float* __restrict__ target,
                                                                const float* __restrict__ w, float* __restrict__ out,
                                                                int H, int W, int slabs) {
  __shared__ float red[4][4];
  const int plane = blockIdx.x / slabs, slab = blockIdx.x - plane * slabs;
  const size_t base = (size_t)plane * H * W;
  const int h0 = slab * SC_ROWS, h1 = min(H, h0 + SC_ROWS);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int w4 = W >> 2;
  for (int h = h0; h < h1; ++h) {
    const float wt = w[h];
    const f32x4* pr = reinterpret_cast<const f32x4*>(pred + base + (size_t)h * W);
    const f32x4* tr = reinterpret_cast<const f32x4*>(target + base + (size_t)h * W);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = threadIdx.x; i < w4; i += 256) {
      const f32x4 p = pr[i], t = tr[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = p[c] - t[c];
        a0 += d * d; a1 += p[c] * t[c]; a2 += p[c] * p[c]; a3 += t[c] * t[c];
      }
    }
    s0 += wt * a0; s1 += wt * a1; s2 += wt * a2; s3 += wt * a3;
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; }
  __syncthreads();
  if (threadIdx.x < 4)
    atomicAdd(&out[plane * 4 + threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
}  // namespace

extern "C" int pangu_lat_weighted_sums(pangu_stream_t stream, const float* pred, const float* target,
                                       const float* lat_weight, float* out, int planes, int H, int W) {
  if (!pred || !target || !lat_weight || !out) return PANGU_E_NULL;
  if (planes <= 0 || H <= 0 || W <= 0 || (W & 3)) return PANGU_E_SHAPE;
  const int slabs = (H + SC_ROWS - 1) / SC_ROWS;
  hipLaunchKernelGGL(lat_weighted_sums_kernel, dim3(planes * slabs), dim3(256), 0, (hipStream_t)stream, pred, target,
                     lat_weight, out, H, W, slabs);
  return pangu_launch_status();
}
