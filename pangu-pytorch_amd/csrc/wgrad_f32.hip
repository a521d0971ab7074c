// fp32 weight-gradient GEMM for gfx950:  dW[N,K] += dC[M,N]^T @ A[M,K],  db[N] += colsum(dC).
//
// The contraction runs over the token dimension M (65k..521k) while the output is small (<= 1536 x 384), so the
// grid is (output tiles) x (M splits): each 256-thread workgroup owns one 64*TNN(n) x 64*TK(k) output tile and a
// contiguous slab of tokens, accumulates it with v_mfma_f32_32x32x2_f32 and adds the tile to dW with no-return
// fp32 atomics (each wave-instruction adds two 128-B row segments: the full-rate atomic shape).
// Both operands are read exactly as they lie in memory (token-major rows): a 16-token slab of dC and of A is
// staged in LDS as [token][column]; MFMA fragments are ds_read_b32 with consecutive lanes on consecutive
// columns (conflict-free), lane half h supplying token 2s+h of k-step s.
// All loads go through range-checked buffer descriptors (rows >= M read as 0, no divergent branches).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int WG_BM = 16;    // tokens per K-step

// TNN x TK = 32x32 accumulators per wave (2x2 waves): <2,*> for N % 128 == 0, <3,*> for N % 192 == 0 (N = 192 would
// leave every second 128-row tile half empty)
template <int TNN, int TK>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const float* __restrict__ dC, int lddc,
                                                           const float* __restrict__ A, int lda,
                                                           float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                           int K, int n_tiles, int k_tiles, int rows_per_split) {
  constexpr int WG_BN = 64 * TNN;                    // output rows (n) per tile
  constexpr int BKC = 64 * TK;                       // output columns (k) per tile
  constexpr int D_LD = WG_BN + 4, A_LD = BKC + 4;    // +4 keeps rows 16-B aligned and staggers the two lane halves
  __shared__ __attribute__((aligned(16))) float smem[2][WG_BM * (D_LD + A_LD)];

  // XCD-aware order (blocks b, b+8, b+16.. share an XCD and its L2): the output tiles of ONE token slab run
  // back to back on one XCD, so each dC / A slab is fetched from HBM once and re-read from that L2.
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int tile = local % (n_tiles * k_tiles), split = (local / (n_tiles * k_tiles)) * 8 + xcd;
  const int n_tile = tile / k_tiles, k_tile = tile - n_tile * k_tiles;
  const int n0 = n_tile * WG_BN, k0 = k_tile * BKC;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dC), 0, (int)(((size_t)(M - 1) * lddc + N) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(float)), 0x00020000);

  // staging assignment.  dC slab: 16 x WG_BN floats = 256*TNN float4; A slab: 16 x BKC = 256*TK float4.
  const int d_per_row = 16 * TNN;
  int d_row[TNN], d_c4[TNN];
  bool d_ok[TNN];
#pragma unroll
  for (int i = 0; i < TNN; ++i) {
    const int f = tid + 256 * i;
    d_row[i] = f / d_per_row;
    d_c4[i] = f - d_row[i] * d_per_row;
    d_ok[i] = n0 + d_c4[i] * 4 < N;
  }
  const int a_per_row = 16 * TK;
  int a_row[TK], a_c4[TK];
  bool a_ok[TK];
#pragma unroll
  for (int i = 0; i < TK; ++i) {
    const int f = tid + 256 * i;
    a_row[i] = f / a_per_row;
    a_c4[i] = f - a_row[i] * a_per_row;
    a_ok[i] = k0 + a_c4[i] * 4 < K;
  }

  auto load_d = [&](int m, int i) -> f32x4 {
    const unsigned off = d_ok[i] ? ((unsigned)(m + d_row[i]) * (unsigned)lddc + (unsigned)(n0 + d_c4[i] * 4)) * 4u : 0xFFFFFFFFu;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, (int)off, 0, 0));
  };
  auto load_a = [&](int m, int i) -> f32x4 {
    const unsigned off = a_ok[i] ? ((unsigned)(m + a_row[i]) * (unsigned)lda + (unsigned)(k0 + a_c4[i] * 4)) * 4u : 0xFFFFFFFFu;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0));
  };

  f32x16 acc[TNN][TK];
#pragma unroll
  for (int i = 0; i < TNN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 dbacc[TNN];
#pragma unroll
  for (int i = 0; i < TNN; ++i) dbacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // rows >= m_end inside the last slab belong to the next split: mask them through the row bound
  f32x4 rd[TNN], ra[TK];
  auto fetch = [&](int m) {
#pragma unroll
    for (int i = 0; i < TNN; ++i) {
      rd[i] = load_d(m, i);
      if (m + d_row[i] >= m_end) rd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TK; ++i) ra[i] = load_a(m, i);
  };
  auto stash = [&](int buf) {
    float* Ds = smem[buf];
    float* As = Ds + WG_BM * D_LD;
#pragma unroll
    for (int i = 0; i < TNN; ++i) {
      *reinterpret_cast<f32x4*>(&Ds[d_row[i] * D_LD + d_c4[i] * 4]) = rd[i];
      dbacc[i] += rd[i];
    }
#pragma unroll
    for (int i = 0; i < TK; ++i) *reinterpret_cast<f32x4*>(&As[a_row[i] * A_LD + a_c4[i] * 4]) = ra[i];
  };

  fetch(m_begin);
  stash(0);
  __syncthreads();
  const int steps = (m_end - m_begin + WG_BM - 1) / WG_BM;
  for (int st = 0; st < steps; ++st) {
    const bool more = st + 1 < steps;
    if (more) fetch(m_begin + (st + 1) * WG_BM);
    const float* Ds = smem[st & 1];
    const float* As = Ds + WG_BM * D_LD;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float fd[TNN], fa[TK];
#pragma unroll
      for (int i = 0; i < TNN; ++i) fd[i] = Ds[(2 * s + lh) * D_LD + wn * 32 * TNN + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < TK; ++j) fa[j] = As[(2 * s + lh) * A_LD + wk * 32 * TK + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < TNN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[i], fa[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash((st + 1) & 1);
    __syncthreads();
  }

  // dW tile: C/D layout col = lane&31 (k), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (n)
#pragma unroll
  for (int j = 0; j < TK; ++j) {
    const int kc = k0 + wk * 32 * TK + j * 32 + lr;
#pragma unroll
    for (int i = 0; i < TNN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 * TNN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < N && kc < K) atomicAdd(&dW[(size_t)n * K + kc], acc[i][j][r]);
      }
  }
  // bias gradient: column sums of the dC slab, from the staging registers (k-tile 0 only)
  if (db != nullptr && k_tile == 0) {
    float* red = &smem[0][0];               // [WG_BN columns], LDS atomics
    __syncthreads();
    if (tid < WG_BN) red[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TNN; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) atomicAdd(&red[d_c4[i] * 4 + c], dbacc[i][c]);
    __syncthreads();
    if (tid < WG_BN && n0 + tid < N) atomicAdd(&db[n0 + tid], red[tid]);
  }
}

template <int TNN, int TK>
int launch_wgrad(hipStream_t s, const float* dC, int lddc, const float* A, int lda, float* dW, float* db, int M, int N,
                 int K) {
  constexpr int WG_BN = 64 * TNN, BKC = 64 * TK;
  const int n_tiles = (N + WG_BN - 1) / WG_BN, k_tiles = (K + BKC - 1) / BKC;
  const int tiles = n_tiles * k_tiles;
  // ~3 workgroups per CU slot-pair: enough M-splits to fill 256 CUs x 2, slabs a multiple of the K-step
  constexpr int target = 768;    // measured sweep: 768 best
  int split = ((target + tiles - 1) / tiles + 7) & ~7;              // equal share per XCD
  int rows = ((M + split - 1) / split + WG_BM - 1) / WG_BM * WG_BM;
  if (rows < 8 * WG_BM) rows = 8 * WG_BM;
  split = ((M + rows - 1) / rows + 7) & ~7;                         // grid padded to whole XCD rounds (empty slabs exit)
  hipLaunchKernelGGL((wgrad_f32_kernel<TNN, TK>), dim3(tiles * split), dim3(256), 0, s, dC, lddc, A, lda, dW, db, M, N, K,
                     n_tiles, k_tiles, rows);
  return pangu_launch_status();
}

}  // namespace

// wgrad_f32_dma.hip: LDS-DMA variant for N % (64*TNN) == 0, K % 192 == 0; returns 1 when the shape is not covered
int pangu_linear_wgrad_f32_dma(hipStream_t s, const float* dC, int lddc, const float* A, int lda, float* dW, float* db,
                               int M, int N, int K, int tnn, int target, float* ws, size_t ws_bytes);

extern "C" int pangu_linear_wgrad(pangu_stream_t stream, const float* dC, int lddc, const float* A, int lda, float* dW,
                                  float* db, int M, int N, int K) {
  return pangu_linear_wgrad_ws(stream, dC, lddc, A, lda, dW, db, M, N, K, nullptr, 0);
}

extern "C" int pangu_linear_wgrad_ws(pangu_stream_t stream, const float* dC, int lddc, const float* A, int lda, float* dW,
                                     float* db, int M, int N, int K, float* workspace, long long workspace_bytes) {
  if (!dC || !A || !dW) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (N & 3) || (K & 3) || lddc < N || lda < K || (lddc & 3) || (lda & 3))
    return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lddc, 4) || !pangu_fits_u32(M, lda, 4)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  // measured (tools/bench_kernels.py wgrad): 192-row tiles win where 128-row tiles would be part empty (N = 192, 576, 160)
  // and at N = 1152 (12 instead of 18 tiles per slab); 128-row tiles win at N = 384, 768, 1536
  const bool wide = (N % 192 == 0 && N % 384 != 0) || (N > 128 && N < 192) || N == 1152;
  constexpr int dma_target = 768;    // measured sweep 768 / 1024 / 1536 / 2048: 768 best (1536 within 1 %)
  {
    const bool ws_ok = workspace != nullptr && workspace_bytes > 0 && (reinterpret_cast<size_t>(workspace) & 15) == 0;
    const int rc = pangu_linear_wgrad_f32_dma(s, dC, lddc, A, lda, dW, db, M, N, K, wide ? 3 : 2, dma_target,
                                              ws_ok ? workspace : nullptr, ws_ok ? (size_t)workspace_bytes : 0);
    if (rc != -1000) return rc;          // -1000 = PANGU_WGRAD_NOT_COVERED (wgrad_f32_dma.hip): fall through
  }
  if (wide) {
    if (K % 192 == 0) return launch_wgrad<3, 3>(s, dC, lddc, A, lda, dW, db, M, N, K);
    if (K % 128 == 0) return launch_wgrad<3, 2>(s, dC, lddc, A, lda, dW, db, M, N, K);
    return launch_wgrad<3, 1>(s, dC, lddc, A, lda, dW, db, M, N, K);
  }
  if (K % 192 == 0) return launch_wgrad<2, 3>(s, dC, lddc, A, lda, dW, db, M, N, K);
  if (K % 128 == 0) return launch_wgrad<2, 2>(s, dC, lddc, A, lda, dW, db, M, N, K);
  return launch_wgrad<2, 1>(s, dC, lddc, A, lda, dW, db, M, N, K);
}
