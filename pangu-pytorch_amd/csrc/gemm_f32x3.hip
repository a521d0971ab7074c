// fp32 projection GEMM on the bf16 matrix pipe with split ("bf16x3") products, gfx950 — OPT-IN fast path.
//
// C[M,N] = act(A[M,K] @ W[N,K]^T + bias) with fp32 inputs, fp32 output, fp32 accumulation.  Each fp32 operand is split
// on the fly into hi = bf16(x), lo = bf16(x - hi) (x = hi + lo up to 2^-17 |x|) and every product is evaluated as
// hi*hi + hi*lo + lo*hi with three v_mfma_f32_16x16x32_bf16 (the dropped lo*lo term is < 2^-16 relative).  That is
// 3/16 of the matrix cycles of the exact-fp32 MFMA path (gemm_f32.hip), which turns the projections from MFMA-bound
// into HBM-bound; the price is ~1e-5 instead of ~1e-7 relative error per dot product.  It is NOT the default: the
// default fp32 path stays on v_mfma_f32_32x32x2_f32 (exact).  Measured parity of the whole forward with this path
// is reported by bench.py ("f32x3_forward") and asserted in tests/test_gpu_f32x3.py.
//
// Structure = gemm_bf16.hip: 128 x (64*TN) tile, 4 waves 2x2, swapped operands (a lane owns 4 consecutive output
// columns), XOR-swizzled [row][64 bf16] LDS image where a row holds 32 hi values followed by the 32 lo values of one
// 32-wide K-step, register staging one step ahead, 16-row epilogue groups through LDS -> whole-row 16-B stores.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int XBM = 128;
constexpr int XBK = 32;        // fp32 k-values per step (= 32 hi + 32 lo bf16 per LDS row)

__device__ inline int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 8 fp32 -> 8 hi bf16 (one 16-B chunk) + 8 lo bf16
__device__ inline void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  unsigned h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const u16 hb = __builtin_bit_cast(u16, (__bf16)v[e]);
    const float hf = __builtin_bit_cast(float, (unsigned)hb << 16);
    h[e] = hb;
    l[e] = __builtin_bit_cast(u16, (__bf16)(v[e] - hf));
  }
  hi = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  lo = u32x4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}

template <int TN, int ACT, bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void gemm_tn_f32x3_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ C, int ldc, int M, int N, int K,
                                                               int m_tiles, int n_tiles, float* __restrict__ aux) {
  constexpr int BN = 64 * TN;
  constexpr int STAGE = (XBM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int m_tile = (local / n_tiles) * 8 + xcd;
  const int n_tile = local % n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * XBM, n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lc = lane & 15, lg = lane >> 4;

  // staging task f = tid + 256 i: row f>>2 (0 .. XBM+BN-1), k-chunk f&3 (8 fp32 = 32 B of the source row)
  constexpr int NT = (XBM + BN) * 4 / 256;                 // 2 + TN tasks per thread
  const float* src[NT];
  int dst[NT];
  const int kq = tid & 3;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int row = (tid >> 2) + 64 * i;
    if (row < XBM) {
      int r = m0 + row;
      r = r < M ? r : M - 1;
      src[i] = A + (size_t)r * lda + kq * 8;
      dst[i] = row;
    } else {
      int r = n0 + row - XBM;
      r = r < N ? r : N - 1;
      src[i] = W + (size_t)r * K + kq * 8;
      dst[i] = row;
    }
  }
  const int KT = (K + XBK - 1) / XBK;
  f32x4 st0[NT], st1[NT];
  auto fetch = [&](int kt) {
    const bool in = kt * XBK + kq * 8 < K;                   // K % 8 == 0
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (in) {
        st0[i] = *reinterpret_cast<const f32x4*>(src[i] + (size_t)kt * XBK);
        st1[i] = *reinterpret_cast<const f32x4*>(src[i] + (size_t)kt * XBK + 4);
      } else {
        st0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        st1[i] = st0[i];
      }
    }
  };
  auto stash = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      u32x4 hi, lo;
      split8(st0[i], st1[i], hi, lo);
      const int row = dst[i] < XBM ? dst[i] : dst[i] - XBM;
      unsigned char* img = base + (dst[i] < XBM ? 0 : XBM * 128);
      *reinterpret_cast<u32x4*>(img + swz(row, kq)) = hi;           // chunks 0..3: hi of k 0..31
      *reinterpret_cast<u32x4*>(img + swz(row, 4 + kq)) = lo;       // chunks 4..7: lo of k 0..31
    }
  };

  f32x4 acc[4][2 * TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;
    if (more) fetch(kt + 1);
    const unsigned char* As = smem + (kt & 1) * STAGE;
    const unsigned char* Ws = As + XBM * 128;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(As + swz(wm * 64 + i * 16 + lc, lg));
      al[i] = *reinterpret_cast<const bf16x8*>(As + swz(wm * 64 + i * 16 + lc, 4 + lg));
    }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(Ws + swz(wn * 32 * TN + j * 16 + lc, lg));
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(Ws + swz(wn * 32 * TN + j * 16 + lc, 4 + lg));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // small terms first, then the dominant one (D[n][m], swapped operands)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[i], acc[i][j], 0, 0, 0);
      }
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane (lg, lc) of tile (i, j) holds C[m = wm*64 + 16i + lc][n = wn*32TN + 16j + 4lg + r]
  const int wave_n0 = n0 + wn * 32 * TN;
  const int wave_m0 = m0 + wm * 64;
  constexpr int ROWB = 32 * TN * 4;                        // payload bytes per patch row (fp32)
  constexpr int EP_LD = ROWB + 16;
  constexpr int CPR = ROWB / 16;
  unsigned char* ep = smem + wave * (16 * EP_LD);
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      C, 0, (int)(((size_t)(M - 1) * ldc + N) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      aux, 0, aux ? (int)((size_t)M * N * sizeof(float)) : 0, 0x00020000);
  f32x4 bv[2 * TN];
#pragma unroll
  for (int j = 0; j < 2 * TN; ++j) {
    const int col = wave_n0 + j * 16 + lg * 4;
    bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) bv[j] = *reinterpret_cast<const f32x4*>(bias + (col < N ? col : 0));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (ACT == PANGU_ACT_GELU_BWD) {       // stage the saved pre-activation rows of this group (coalesced)
#pragma unroll
      for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
        const int f = lane + 64 * it;
        const int row = f / CPR, ch = f % CPR;
        const int col = wave_n0 + ch * 4;
        if (f < 16 * CPR) {
          const unsigned off = col < N ? ((unsigned)(wave_m0 + i * 16 + row) * (unsigned)N + (unsigned)col) * 4u : 0xFFFFFFFFu;
          *reinterpret_cast<u32x4*>(ep + row * EP_LD + ch * 16) = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)off, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
      f32x4 v = acc[i][j] + bv[j];
      f32x4* slot = reinterpret_cast<f32x4*>(ep + lc * EP_LD + (j * 16 + lg * 4) * 4);
      if (ACT == PANGU_ACT_GELU_BWD) {
        const f32x4 x = *slot;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] *= gelu_erf_grad(x[c]);
      }
      if (ACT == PANGU_ACT_GELU) {
        const int col = wave_n0 + j * 16 + lg * 4;
        const unsigned xo = col < N ? ((unsigned)(wave_m0 + i * 16 + lc) * (unsigned)N + (unsigned)col) * 4u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), x_rsrc, (int)xo, 0, 0);   // dropped when aux == NULL
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = gelu_erf(v[c]);
      }
      *slot = v;
    }
#pragma unroll
    for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
      const int f = lane + 64 * it;
      const int row = f / CPR, ch = f % CPR;
      const int col = wave_n0 + ch * 4;
      if (f < 16 * CPR) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * EP_LD + ch * 16);
        const unsigned off = col < N ? ((unsigned)(wave_m0 + i * 16 + row) * (unsigned)ldc + (unsigned)col) * 4u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, (int)off, 0, 0);
      }
    }
  }
}

template <int TN>
int launch_x3(hipStream_t s, const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N,
              int K, int act, float* aux) {
  constexpr int BN = 64 * TN;
  const int m_tiles = (M + XBM - 1) / XBM, n_tiles = (N + BN - 1) / BN;
  const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  const size_t shm = 2 * (size_t)(XBM + BN) * 128;
  dim3 g(grid), blk(256);
#define PANGU_X3(ACT, HB)                                                                                             \
  do {                                                                                                                \
    auto kern = gemm_tn_f32x3_kernel<TN, ACT, HB>;                                                                    \
    PANGU_ENSURE_DYN_LDS(kern, shm);                                                                                  \
    hipLaunchKernelGGL(kern, g, blk, shm, s, A, lda, W, bias, C, ldc, M, N, K, m_tiles, n_tiles, aux);                \
  } while (0)
  if (act == PANGU_ACT_GELU) {
    if (bias) PANGU_X3(PANGU_ACT_GELU, true); else PANGU_X3(PANGU_ACT_GELU, false);
  } else if (act == PANGU_ACT_GELU_BWD) {
    if (bias) PANGU_X3(PANGU_ACT_GELU_BWD, true); else PANGU_X3(PANGU_ACT_GELU_BWD, false);
  } else {
    if (bias) PANGU_X3(PANGU_ACT_NONE, true); else PANGU_X3(PANGU_ACT_NONE, false);
  }
#undef PANGU_X3
  return pangu_launch_status();
}

}  // namespace

extern "C" int pangu_linear_fwd_f32x3(pangu_stream_t stream, const float* A, int lda, const float* W, const float* bias,
                                      float* C, int ldc, int M, int N, int K, int act, float* aux) {
  if (!A || !W || !C) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 3) || lda < K || ldc < N || (lda & 3) || (ldc & 3)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lda, 4) || !pangu_fits_u32(M, ldc, 4)) return PANGU_E_RANGE;
  if (act != PANGU_ACT_NONE && act != PANGU_ACT_GELU && act != PANGU_ACT_GELU_BWD) return PANGU_E_ARG;
  if (act == PANGU_ACT_GELU_BWD && !aux) return PANGU_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (N % 192 == 0 || (N > 128 && N < 192)) return launch_x3<3>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
  if (N % 128 == 0) return launch_x3<2>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
  return launch_x3<1>(s, A, lda, W, bias, C, ldc, M, N, K, act, aux);
}
