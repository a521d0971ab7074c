// fp32 projection GEMM, LDS-DMA variant for the 128 x 192 tile (TN = 3): C[M,N] = act(A[M,K] @ W[N,K]^T + bias).
//
// Same MFMA scheme as gemm_f32.hip (exact-f32 v_mfma_f32_32x32x2_f32, permuted-k b128 fragments, 2 x 2 waves of 64 x 96),
// but built for OCCUPANCY, the strongest lever measured on this chip: the register-staged kernel needs 168 VGPRs and
// 51 KB of LDS (three workgroups per CU); here the operands travel L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no
// staging registers, no ds_write pass) into an UNPADDED 64-byte-row image whose 16-B chunks are XOR-swizzled on the
// SOURCE side (conflict-free ds_read_b128 lane groups), the fragments of the two k-halves share registers, and a ring of
// two 20-KB stages leaves 40 KB per workgroup: FOUR workgroups per CU at <= 128 VGPRs.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128;
constexpr int BK = 16;       // floats per K-step = one 64-byte LDS row

// byte offset of logical 16-B chunk `chunk` of row `row` (64-byte rows, chunk XOR F[(row>>2)&3], F = {0,2,3,1})
__device__ inline int kswz64(int row, int chunk) {
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((chunk ^ f) << 4);
}

// TN = 3: 128 x 192 tile, 40 KB ring, 127 VGPRs -> four workgroups per CU; TN = 2: 128 x 128 tile, 32 KB ring -> five
template <int TN, int ACT, bool HAS_BIAS>
__global__ __launch_bounds__(256, TN == 3 ? 4 : 5) void gemm_tn_f32_dma_kernel(const float* __restrict__ A, int lda,
                                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                                 float* __restrict__ C, int ldc, int M, int N, int K,
                                                                 int m_tiles, int n_tiles, float* __restrict__ aux) {
  constexpr int BN = 64 * TN;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * 64;          // bytes per ring slot
  constexpr int LPS = ROWS / 64;            // LDS-DMA instructions per wave and stage (16 rows of 64 B each)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  // XCD-aware tile assignment (blocks b, b+8, b+16.. share an XCD; they walk the n-tiles of one m-tile).
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int m_tile = (local / n_tiles) * 8 + xcd;
  const int n_tile = local % n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(W), 0, (int)((size_t)N * K * sizeof(float)), 0x00020000);
  // DMA instruction q = i*4 + wave (i < LPS) fills rows 16q .. 16q+15 (1 KB): this lane fills (row 16q + lane>>2, physical
  // chunk lane&3) and therefore fetches logical chunk (lane&3) ^ F(row).  Rows >= M / >= N are out of range: zeros.
  unsigned voff[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = 16 * (i * 4 + wave) + (lane >> 2);
    const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
    const int c = (lane & 3) ^ f;
    voff[i] = row < BM ? ((unsigned)(m0 + row) * (unsigned)lda + c * 4) * 4u
                       : ((unsigned)(n0 + row - BM) * (unsigned)K + c * 4) * 4u;
  }
  auto issue = [&](int kt) {
    unsigned char* base = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int q = i * 4 + wave;
      auto dst = (__attribute__((address_space(3))) void*)(base + q * 1024);
      if (16 * q < BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)voff[i], kt * BK * 4, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, (int)voff[i], kt * BK * 4, 0, 0);
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment byte offsets: lane (lr, lh) owns k = 8lh .. 8lh+7 of row (.. + lr): logical chunks 2lh (half 0), 2lh+1 (half 1)
  int off_a[2][2], off_w[TN][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) off_a[i][h] = kswz64(wm * 64 + i * 32 + lr, 2 * lh + h);
#pragma unroll
    for (int j = 0; j < TN; ++j) off_w[j][h] = BM * 64 + kswz64(wn * 32 * TN + j * 32 + lr, 2 * lh + h);
  }

  const int KT = K / BK;
  issue(0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of step kt has landed
    __builtin_amdgcn_s_barrier();                          // ... and everybody's; slot (kt+1)&1 is free
    asm volatile("" ::: "memory");
    if (kt + 1 < KT) issue(kt + 1);
    const unsigned char* St = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // the two k-halves share the fragment registers
      f32x4 fa[2], fw[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(St + off_a[i][h]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fw[j] = *reinterpret_cast<const f32x4*>(St + off_w[j][h]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fw[j][s], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                                         // every wave is done with the ring before the epilogue reuses it

  // epilogue (as gemm_f32.hip): each 32x32 tile is transposed through a wave-private LDS patch -> 16-B row segments
  constexpr int EP_LD = 36;
  float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      C, 0, (int)(((size_t)(M - 1) * ldc + N) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      aux, 0, aux ? (int)((size_t)M * N * sizeof(float)) : 0, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * 32 * TN + j * 32 + ec;
    const bool col_ok = col < N;
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
        // non-temporal: the 0.2-1.6 GB output is written once and read by the NEXT kernel; streaming it past L2 keeps the A / W
        // panels resident (+2-4 % per launch, +0.7 % on the forward)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rsrc, (int)off, 0, 2);
      }
    }
  }
}

}  // namespace

template <int TN>
int launch_dma(hipStream_t s, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K,
               int act, float* aux) {
  constexpr int BN = 64 * TN;
  const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
  const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  dim3 g(grid), blk(256);
#define PANGU_DMA_LAUNCH(ACT, HB) \
  hipLaunchKernelGGL((gemm_tn_f32_dma_kernel<TN, ACT, HB>), g, blk, 0, s, A, lda, W, bias, C, ldc, M, N, K, m_tiles, n_tiles, aux)
  if (act == PANGU_ACT_GELU) {
    if (bias) PANGU_DMA_LAUNCH(PANGU_ACT_GELU, true); else PANGU_DMA_LAUNCH(PANGU_ACT_GELU, false);
  } else if (act == PANGU_ACT_GELU_BWD) {
    if (bias) PANGU_DMA_LAUNCH(PANGU_ACT_GELU_BWD, true); else PANGU_DMA_LAUNCH(PANGU_ACT_GELU_BWD, false);
  } else if (act == PANGU_ACT_ADD) {
    if (bias) PANGU_DMA_LAUNCH(PANGU_ACT_ADD, true); else PANGU_DMA_LAUNCH(PANGU_ACT_ADD, false);
  } else {
    if (bias) PANGU_DMA_LAUNCH(PANGU_ACT_NONE, true); else PANGU_DMA_LAUNCH(PANGU_ACT_NONE, false);
  }
#undef PANGU_DMA_LAUNCH
  return pangu_launch_status();
}

// gemm_f32.hip dispatches here: tn = 3 for its 192-wide-tile shapes, 2 for the 128-wide ones
int pangu_linear_f32_dma(hipStream_t s, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M,
                         int N, int K, int act, float* aux, int tn) {
  if (tn == 3) return launch_dma<3>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
  return launch_dma<2>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
}
