// fp32 projection GEMM + post-norm residual in one launch, LDS-DMA variant (inference path):
//   out[M,N] = shortcut + branch_scale * (LayerNorm(A[M,K] @ W[N,K]^T + bias) * gamma + beta)           (layers.py:250-251)
// for N = 192 (WNW = 2: 4 waves, 128 x 192 tile, 40-KB ring, four workgroups per CU) and N = 384 (WNW = 4: 8 waves,
// 128 x 384 tile, 64-KB ring, two workgroups per CU): either way 16 waves per CU at <= 128 VGPRs, the occupancy of the plain
// LDS-DMA GEMM (gemm_f32_dma.hip), whose main loop this is -- and the tile spans the whole row, so the LayerNorm statistics
// never leave the workgroup.
// Epilogue, one 32-row group at a time (register budget): the wave's three 32x32 accumulator tiles go through its LDS patch
// and come back as row-major float4s (+ bias), per-row (sum, sum of squares) are reduced over the 8 lanes sharing a row with
// DPP adds, the WNW partial rows meet in a small LDS table behind one barrier, then normalise / gamma / beta / branch scale /
// shortcut and 16-B stores of whole 128-B row segments.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 16;       // floats per K-step = one 64-byte LDS row
constexpr float LN_EPS = 1e-5f;

__device__ inline int kswz64(int row, int chunk) {
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((chunk ^ f) << 4);
}
// sum over the 8 lanes that share lane >> 3 (DPP quad_perm x2 + row_half_mirror), result in every lane
__device__ inline float oct_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <int WNW, bool HAS_BIAS>
__global__ __launch_bounds__(128 * WNW, 4) void gemm_ln_residual_f32_dma_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ shortcut, int lds_sc, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ C, int ldc, int M, int K, float branch_scale) {
  constexpr int NW = 2 * WNW;               // waves
  constexpr int BN = 96 * WNW;              // = N
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * 64;
  constexpr int LPS = ROWS / 16 / NW;       // LDS-DMA instructions per wave and stage
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int lr = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(W), 0, (int)((size_t)BN * K * sizeof(float)), 0x00020000);
  unsigned voff[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = 16 * (i * NW + wave) + (lane >> 2);
    const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
    const int c = (lane & 3) ^ f;
    voff[i] = row < BM ? ((unsigned)(m0 + row) * (unsigned)lda + c * 4) * 4u : ((unsigned)(row - BM) * (unsigned)K + c * 4) * 4u;
  }
  auto issue = [&](int kt) {
    unsigned char* base = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int q = i * NW + wave;
      auto dst = (__attribute__((address_space(3))) void*)(base + q * 1024);
      if (16 * q < BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)voff[i], kt * BK * 4, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, (int)voff[i], kt * BK * 4, 0, 0);
    }
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int off_a[2][2], off_w[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) off_a[i][h] = kswz64(wm * 64 + i * 32 + lr, 2 * lh + h);
#pragma unroll
    for (int j = 0; j < 3; ++j) off_w[j][h] = BM * 64 + kswz64(wn * 96 + j * 32 + lr, 2 * lh + h);
  }

  const int KT = K / BK;
  issue(0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < KT) issue(kt + 1);
    const unsigned char* St = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 fa[2], fw[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(St + off_a[i][h]);
#pragma unroll
      for (int j = 0; j < 3; ++j) fw[j] = *reinterpret_cast<const f32x4*>(St + off_w[j][h]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fw[j][s], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                                         // every wave is done with the ring before the epilogue reuses it

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  constexpr int EP_LD = 36;
  float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
  float* stats = reinterpret_cast<float*>(smem) + NW * (32 * EP_LD);     // [2 wm][WNW][64 rows][sum, sumsq]
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      C, 0, (int)(((size_t)(M - 1) * ldc + BN) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(shortcut), 0, (int)(((size_t)(M - 1) * lds_sc + BN) * sizeof(float)), 0x00020000);
  constexpr float INV_C = 1.0f / BN;
  // Register budget of this epilogue: 96 accumulator registers + the 128-VGPR cap of four workgroups per CU leave 32 for everything
  // else, so the transposed rows of a 32-row group (48 registers) can only live in the registers its own accumulators free one
  // 32 x 32 tile at a time, while the OTHER row group's 48 accumulators stay live: a few values go through scratch (round 3: 20-22
  // registers; round 4, with the column-block loop outermost and bias / gamma / beta re-loaded per (row group, column block) instead
  // of hoisted: 8-15), all of it outside the MFMA loop.  Measured alternative (round 4): two transposition passes (statistics, then
  // normalise from the accumulators again) keep the accumulators live through both passes and spill 48-61.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x4 y[3][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int col = wn * 96 + j * 32 + ec;
      asm volatile("" : "+v"(col));
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP_LD + lr] = acc[i][j][r];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(er + 8 * it) * EP_LD + ec]);
        v += bv;
        y[j][it] = v;
        s[it] += (v[0] + v[1]) + (v[2] + v[3]);
        q[it] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float ss = oct_sum(s[it]), qq = oct_sum(q[it]);
      if ((lane & 7) == 0) {
        float* st = stats + (((wm * WNW + wn) * 64) + i * 32 + er + 8 * it) * 2;
        st[0] = ss;
        st[1] = qq;
      }
    }
    __syncthreads();
    float mean[4], rstd[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rloc = i * 32 + er + 8 * it;
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int w = 0; w < WNW; ++w) {
        const float* st = stats + ((wm * WNW + w) * 64 + rloc) * 2;
        ts += st[0];
        tq += st[1];
      }
      mean[it] = ts * INV_C;
      rstd[it] = rsqrtf(fmaxf(tq * INV_C - mean[it] * mean[it], 0.f) + LN_EPS);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int col = wn * 96 + j * 32 + ec;
      asm volatile("" : "+v"(col));
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const unsigned row = (unsigned)(m0 + wm * 64 + i * 32 + er + 8 * it);
        const f32x4 sc = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, (int)((row * (unsigned)lds_sc + (unsigned)col) * 4u), 0, 0));
        const f32x4 v = sc + branch_scale * ((y[j][it] - mean[it]) * rstd[it] * gm + bt);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rsrc,
                                               (int)((row * (unsigned)ldc + (unsigned)col) * 4u), 0, 0);
      }
    }
  }
}

template <int WNW>
int launch_ln_dma(hipStream_t s, const float* A, int lda, const float* W, const float* bias, const float* shortcut, int lds,
                  const float* gamma, const float* beta, float* out, int ldo, int M, int K, float branch_scale) {
  dim3 g((M + BM - 1) / BM), blk(128 * WNW);
  if (bias)
    hipLaunchKernelGGL((gemm_ln_residual_f32_dma_kernel<WNW, true>), g, blk, 0, s, A, lda, W, bias, shortcut, lds, gamma, beta,
                       out, ldo, M, K, branch_scale);
  else
    hipLaunchKernelGGL((gemm_ln_residual_f32_dma_kernel<WNW, false>), g, blk, 0, s, A, lda, W, bias, shortcut, lds, gamma, beta,
                       out, ldo, M, K, branch_scale);
  return pangu_launch_status();
}

}  // namespace

extern "C" int pangu_linear_ln_residual_fwd(pangu_stream_t stream, const float* A, int lda, const float* W, const float* bias,
                                            const float* shortcut, int lds, const float* gamma, const float* beta, float* out,
                                            int ldo, int M, int N, int K, float branch_scale) {
  if (!A || !W || !shortcut || !gamma || !beta || !out) return PANGU_E_NULL;
  if (M <= 0 || K <= 0 || (K % BK) != 0 || lda < K || (lda & 3) || ldo < N || (ldo & 3) || lds < N || (lds & 3))
    return PANGU_E_SHAPE;
  if (N != 192 && N != 384) return PANGU_E_SHAPE;          // the tile must span the whole row
  if (!pangu_fits_u32(M, lda, 4) || !pangu_fits_u32(M, ldo, 4) || !pangu_fits_u32(M, lds, 4)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  if (N == 192) return launch_ln_dma<2>(s, A, lda, W, bias, shortcut, lds, gamma, beta, out, ldo, M, K, branch_scale);
  return launch_ln_dma<4>(s, A, lda, W, bias, shortcut, lds, gamma, beta, out, ldo, M, K, branch_scale);
}
