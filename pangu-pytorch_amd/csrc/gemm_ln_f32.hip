// fp32 projection GEMM with the post-norm residual fused into its epilogue (inference path), gfx950:
//   out[M,192] = shortcut + branch_scale * (LayerNorm(A[M,K] @ W[192,K]^T + bias) * gamma + beta)      (layers.py:250-251)
// Only for N = 192, where the 128 x 192 tile of gemm_f32.hip (TN = 3) already spans the whole row: the main loop below IS
// that kernel's (exact-f32 v_mfma_f32_32x32x2_f32, BK = 16, permuted-k b128 fragments, 3 workgroups per CU); the epilogue
// transposes each 32x32 accumulator tile through its wave-private LDS patch as before, but keeps the row-major float4s in
// registers (they replace the accumulators one tile at a time), reduces (sum, sum of squares) of each row over the 8 lanes
// that share it with DPP adds, meets the partner wave's half row in a 2-KB LDS table behind ONE barrier, then normalises,
// applies gamma / beta / branch scale, adds the shortcut (read as the same 16-B row segments) and stores.
// Saves the branch's HBM round trip: the standalone LN-residual kernel is HBM-bound (3 passes over N x C).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int LDS_LD = 20;   // padded row (floats)
constexpr int TN = 3;
constexpr int BN = 64 * TN;  // = N = 192
constexpr float LN_EPS = 1e-5f;

// sum over the 8 lanes that share lane >> 3 (DPP quad_perm x2 + row_half_mirror), result in every lane
__device__ inline float oct_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 3) void gemm_ln_residual_f32_kernel(const float* __restrict__ A, int lda,
                                                                      const float* __restrict__ W,
                                                                      const float* __restrict__ bias,
                                                                      const float* __restrict__ shortcut, int lds_sc,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float* __restrict__ C,
                                                                      int ldc, int M, int K, float branch_scale) {
  __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDS_LD];

  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // global -> register staging assignment: float4 index f = tid + 256*i, row = f>>2, kq = f&3
  const int ld_row = tid >> 2, ld_kq = tid & 3;
  const float* a_ptr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = m0 + ld_row + 64 * i;
    r = r < M ? r : M - 1;
    a_ptr[i] = A + (size_t)r * lda + ld_kq * 4;
  }
  const float* w_ptr[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) w_ptr[i] = W + (size_t)(ld_row + 64 * i) * K + ld_kq * 4;
  const int st_a = ld_row * LDS_LD + ld_kq * 4;   // + 64*LDS_LD*i

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[2], rw[TN];
#pragma unroll
  for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i]);
#pragma unroll
  for (int i = 0; i < TN; ++i) rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i]);
  {
    float* As = smem[0];
    float* Ws = As + BM * LDS_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&As[st_a + 64 * LDS_LD * i]) = ra[i];
#pragma unroll
    for (int i = 0; i < TN; ++i) *reinterpret_cast<f32x4*>(&Ws[st_a + 64 * LDS_LD * i]) = rw[i];
  }
  __syncthreads();

  const int KT = K / BK;
  const int rd_a = (wm * 64 + lr) * LDS_LD + lh * 8;
  const int rd_w = (wn * 32 * TN + lr) * LDS_LD + lh * 8;

  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;
    if (more) {
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (kt + 1) * BK);
#pragma unroll
      for (int i = 0; i < TN; ++i) rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + (kt + 1) * BK);
    }
    const float* As = smem[kt & 1];
    const float* Ws = As + BM * LDS_LD;
    f32x4 fa0[2], fw0[TN], fa1[2], fw1[TN];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa0[i] = *reinterpret_cast<const f32x4*>(&As[rd_a + i * 32 * LDS_LD]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fw0[j] = *reinterpret_cast<const f32x4*>(&Ws[rd_w + j * 32 * LDS_LD]);
#pragma unroll
    for (int i = 0; i < 2; ++i) fa1[i] = *reinterpret_cast<const f32x4*>(&As[rd_a + i * 32 * LDS_LD + 4]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fw1[j] = *reinterpret_cast<const f32x4*>(&Ws[rd_w + j * 32 * LDS_LD + 4]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][s], fw0[j][s], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][s], fw1[j][s], acc[i][j], 0, 0, 0);
    if (more) {
      float* An = smem[(kt + 1) & 1];
      float* Wn = An + BM * LDS_LD;
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&An[st_a + 64 * LDS_LD * i]) = ra[i];
#pragma unroll
      for (int i = 0; i < TN; ++i) *reinterpret_cast<f32x4*>(&Wn[st_a + 64 * LDS_LD * i]) = rw[i];
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // After the transposition lane (er = lane>>3, ec = 4*(lane&7)) holds, for it = 0..3, y[row 32i + er + 8it][col 32j + ec..+3].
  constexpr int EP_LD = 36;
  float* ep = &smem[0][0] + wave * (32 * EP_LD);
  float* stats = &smem[0][0] + 4 * (32 * EP_LD);             // [2 wm][2 wn][64 rows][sum, sumsq]
  const int er = lane >> 3, ec = (lane & 7) * 4;
  f32x4 y[2][TN][4];
  float s[2][4], q[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int it = 0; it < 4; ++it) { s[i][it] = 0.f; q[i][it] = 0.f; }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * 32 * TN + j * 32 + ec;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP_LD + lr] = acc[i][j][r];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(er + 8 * it) * EP_LD + ec]);
        v += bv;
        y[i][j][it] = v;
        s[i][it] += (v[0] + v[1]) + (v[2] + v[3]);
        q[i][it] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
  }
  // row statistics: 8 lanes share a row (this wave's 96 columns), the partner wave holds the other 96
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float ss = oct_sum(s[i][it]), qq = oct_sum(q[i][it]);
      if ((lane & 7) == 0) {
        float* st = stats + (((wm * 2 + wn) * 64) + i * 32 + er + 8 * it) * 2;
        st[0] = ss;
        st[1] = qq;
      }
    }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      C, 0, (int)(((size_t)(M - 1) * ldc + BN) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(shortcut), 0, (int)(((size_t)(M - 1) * lds_sc + BN) * sizeof(float)), 0x00020000);
  constexpr float INV_C = 1.0f / BN;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rloc = i * 32 + er + 8 * it;
      const float* s0 = stats + ((wm * 2 + 0) * 64 + rloc) * 2;
      const float* s1 = stats + ((wm * 2 + 1) * 64 + rloc) * 2;
      const float mean = (s0[0] + s1[0]) * INV_C;
      const float rstd = rsqrtf(fmaxf((s0[1] + s1[1]) * INV_C - mean * mean, 0.f) + LN_EPS);
      const unsigned row = (unsigned)(m0 + wm * 64 + rloc);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * 32 * TN + j * 32 + ec;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + col);
        const f32x4 sc = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, (int)((row * (unsigned)lds_sc + (unsigned)col) * 4u), 0, 0));
        const f32x4 v = sc + branch_scale * ((y[i][j][it] - mean) * rstd * gm + bt);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rsrc,
                                               (int)((row * (unsigned)ldc + (unsigned)col) * 4u), 0, 0);
      }
    }
}

}  // namespace

int pangu_linear_ln_f32_dma(hipStream_t s, const float* A, int lda, const float* W, const float* bias, const float* shortcut,
                            int lds, const float* gamma, const float* beta, float* out, int ldo, int M, int N, int K,
                            float branch_scale);      // gemm_ln_f32_dma.hip

extern "C" int pangu_linear_ln_residual_fwd(pangu_stream_t stream, const float* A, int lda, const float* W, const float* bias,
                                            const float* shortcut, int lds, const float* gamma, const float* beta, float* out,
                                            int ldo, int M, int N, int K, float branch_scale) {
  if (!A || !W || !shortcut || !gamma || !beta || !out) return PANGU_E_NULL;
  if (M <= 0 || K <= 0 || (K % BK) != 0 || lda < K || (lda & 3) || ldo < N || (ldo & 3) || lds < N || (lds & 3))
    return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lda, 4) || !pangu_fits_u32(M, ldo, 4) || !pangu_fits_u32(M, lds, 4)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  // default: the LDS-DMA kernels (gemm_ln_f32_dma.hip: N = 192 and N = 384); PANGU_LN_DMA=0 keeps the register-staged N = 192 one
  static const int dma = getenv("PANGU_LN_DMA") ? atoi(getenv("PANGU_LN_DMA")) : 1;
  if (dma && (N == 192 || N == 384))
    return pangu_linear_ln_f32_dma(s, A, lda, W, bias, shortcut, lds, gamma, beta, out, ldo, M, N, K, branch_scale);
  if (N != BN) return PANGU_E_SHAPE;                       // the tile must span the whole row
  dim3 g((M + BM - 1) / BM), blk(256);
  if (bias)
    hipLaunchKernelGGL(gemm_ln_residual_f32_kernel<true>, g, blk, 0, s, A, lda, W, bias, shortcut, lds, gamma, beta, out, ldo, M,
                       K, branch_scale);
  else
    hipLaunchKernelGGL(gemm_ln_residual_f32_kernel<false>, g, blk, 0, s, A, lda, W, bias, shortcut, lds, gamma, beta, out, ldo, M,
                       K, branch_scale);
  return pangu_launch_status();
}
