// fp32 projection GEMM for gfx950: C[M,N] = act(A[M,K] @ W[N,K]^T + bias), exact-f32 MFMA.
//
// Shape regime of this path: M = 65k..521k tokens, N,K in {64..1536} -> output-tile parallel, no split-K.
// Tile: 128 x (64*TN) per 256-thread workgroup (4 waves as 2x2, each 64 x 32*TN = 2 x TN MFMA 32x32 tiles),
// BK = 16, LDS double-buffered with register prefetch (one barrier per K-step).  Occupancy is the strongest lever measured
// on this kernel: 3 workgroups per CU for TN = 3 (168 VGPRs), 4 for TN <= 2 (<= 128 VGPRs, 4 x 40 KB = all of the LDS), which
// is worth +4-5 % on the N = 384 shapes; 192-row tiles at 2 workgroups per CU lose 2-8 %.
// MFMA: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD = the chip's f32 matrix peak, 157 TF).  Both operands are
// K-contiguous in memory, so each lane reads its 8 k-values of a K-step as two ds_read_b128: the MFMA's
// k index is permuted (lane half h owns k = 8h..8h+7) identically for A and W, which leaves the sum unchanged.
// LDS rows are padded to 20 floats: conflict-free for the ds_read_b128 lane groups.
// Block -> tile map is XCD-aware: workgroups that share an A row-panel (the n-tiles of one m-tile) are
// consecutive on ONE XCD, so the panel is fetched from HBM once and re-read from that XCD's L2.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int LDS_LD = 20;   // padded row (floats)
#ifndef GEMM_WAVES_PER_SIMD
#define GEMM_WAVES_PER_SIMD 3
#endif

template <int TN, int ACT, bool HAS_BIAS>
__global__ __launch_bounds__(256, TN <= 2 ? 4 : GEMM_WAVES_PER_SIMD) void gemm_tn_f32_kernel(const float* __restrict__ A, int lda,
                                                             const float* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ C, int ldc, int M, int N, int K,
                                                             int m_tiles, int n_tiles, float* __restrict__ aux) {
  constexpr int BN = 64 * TN;
  __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDS_LD];

  // XCD-aware tile assignment (blocks b, b+8, b+16.. share an XCD; they walk the n-tiles of one m-tile).
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int m_tile = (local / n_tiles) * 8 + xcd;
  const int n_tile = local % n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

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
  for (int i = 0; i < TN; ++i) {
    int r = n0 + ld_row + 64 * i;
    r = r < N ? r : N - 1;
    w_ptr[i] = W + (size_t)r * K + ld_kq * 4;
  }
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
    // fragments in two halves (k-steps 0-3, then 4-7); the second half is fetched before the first half's MFMAs issue,
    // so only one LDS round trip per K-step is exposed to this wave
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

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // Each 32x32 tile is transposed through a wave-private LDS patch (aliasing the staging buffers, free after
  // the loop's last barrier) so that a lane stores 16 B and a wave-instruction covers 8 rows x 128 B:
  // 24 global_store_dwordx4 per lane instead of 96 dword stores (the store tail is issue-bound).
  constexpr int EP_LD = 36;
  float* ep = &smem[0][0] + wave * (32 * EP_LD);
  const int er = lane >> 3, ec = (lane & 7) * 4;
  // Branch-free tail handling: stores go through a buffer descriptor whose range check drops rows >= M (and
  // lanes whose column is >= N are pointed past the range).  Divergent branches around the stores would make
  // hipcc serialise them with s_waitcnt vmcnt(0).
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      C, 0, (int)(((size_t)(M - 1) * ldc + N) * sizeof(float)), 0x00020000);
  // aux [M][N] dense: GELU -> optional pre-activation output (zero-sized range when absent: stores dropped);
  // GELU_BWD -> saved pre-activation input
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      aux, 0, aux ? (int)((size_t)M * N * sizeof(float)) : 0, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * 32 * TN + j * 32 + ec;
    const bool col_ok = col < N;            // N % 4 == 0 is enforced by the launcher
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bias + (col_ok ? col : 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP_LD + lr] = acc[i][j][r];
      const int row0 = m0 + wm * 64 + i * 32 + er;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(er + 8 * it) * EP_LD + ec]);
        v += bv;
        const unsigned xoff = col_ok ? ((unsigned)(row0 + 8 * it) * (unsigned)N + (unsigned)col) * 4u : 0xFFFFFFFFu;
        if (ACT == PANGU_ACT_GELU) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), x_rsrc, (int)xoff, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = gelu_erf(v[c]);
        }
        if (ACT == PANGU_ACT_GELU_BWD) {
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)xoff, 0, 0));
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] *= gelu_erf_grad(x[c]);
        }
        if (ACT == PANGU_ACT_ADD) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)xoff, 0, 0));
        const unsigned off = col_ok ? ((unsigned)(row0 + 8 * it) * (unsigned)ldc + (unsigned)col) * 4u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rsrc, (int)off, 0, 0);
      }
    }
  }
}

template <int TN>
int launch_tn(hipStream_t s, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M,
              int N, int K, int act, float* aux) {
  constexpr int BN = 64 * TN;
  const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
  const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  dim3 g(grid), blk(256);
#define PANGU_GEMM_LAUNCH(ACT, HB) \
  hipLaunchKernelGGL((gemm_tn_f32_kernel<TN, ACT, HB>), g, blk, 0, s, A, lda, W, bias, C, ldc, M, N, K, m_tiles, n_tiles, aux)
  if (act == PANGU_ACT_GELU) {
    if (bias) PANGU_GEMM_LAUNCH(PANGU_ACT_GELU, true); else PANGU_GEMM_LAUNCH(PANGU_ACT_GELU, false);
  } else if (act == PANGU_ACT_GELU_BWD) {
    if (bias) PANGU_GEMM_LAUNCH(PANGU_ACT_GELU_BWD, true); else PANGU_GEMM_LAUNCH(PANGU_ACT_GELU_BWD, false);
  } else if (act == PANGU_ACT_ADD) {
    if (bias) PANGU_GEMM_LAUNCH(PANGU_ACT_ADD, true); else PANGU_GEMM_LAUNCH(PANGU_ACT_ADD, false);
  } else {
    if (bias) PANGU_GEMM_LAUNCH(PANGU_ACT_NONE, true); else PANGU_GEMM_LAUNCH(PANGU_ACT_NONE, false);
  }
#undef PANGU_GEMM_LAUNCH
  return pangu_launch_status();
}

}  // namespace

int pangu_linear_f32_dma(hipStream_t s, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M,
                         int N, int K, int act, float* aux, int tn);      // gemm_f32_dma.hip

extern "C" int pangu_linear_fwd(pangu_stream_t stream, const float* A, int lda, const float* W, const float* bias,
                                float* C, int ldc, int M, int N, int K, int act, float* aux) {
  if (!A || !W || !C) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || (N & 3) || lda < K || ldc < N || (lda & 3) || (ldc & 3))
    return PANGU_E_SHAPE;
  if (act != PANGU_ACT_NONE && act != PANGU_ACT_GELU && act != PANGU_ACT_GELU_BWD && act != PANGU_ACT_ADD) return PANGU_E_ARG;
  if ((act == PANGU_ACT_GELU_BWD || act == PANGU_ACT_ADD) && !aux) return PANGU_E_NULL;
  if (!pangu_fits_u32(M, lda, 4) || !pangu_fits_u32(M, ldc, 4)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  // Tile width (measured per shape on MI355X, tools/bench_kernels.py): the LDS-DMA kernel (gemm_f32_dma.hip) with 192 columns
  // wherever N is a multiple of 192 or fits one tile (N = 160: 17 % padded MFMAs still beat three 64-wide tiles; four workgroups
  // per CU, +3.5-11 % over the register-staged kernel), 128-column LDS-DMA tiles for N = 384 (five workgroups per CU; 192-wide
  // tiles would leave 2.67 rounds of workgroups); the register-staged kernel below for the other widths (N = 64, 128, ..).
  if (N == 384) return pangu_linear_f32_dma(s, A, lda, W, bias, C, ldc, M, N, K, act, aux, 2);
  if (N % 192 == 0 || (N > 128 && N < 192)) return pangu_linear_f32_dma(s, A, lda, W, bias, C, ldc, M, N, K, act, aux, 3);
  if (N % 128 == 0) return launch_tn<2>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
  return launch_tn<1>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
}
