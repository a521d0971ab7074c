// bf16 projection GEMM for gfx950: C[M,N] = act(A[M,K] @ W[N,K]^T + bias), bf16 operands, fp32 accumulate.
//
// At this path's shapes (M = 65k..521k tokens, N,K <= 1536) a bf16 projection moves ~1 GB for ~0.15 TFLOP: it sits on
// the HBM/MFMA ridge, so the kernel is built to stream: 128 x (64*TN) tile per 256-thread workgroup (4 waves 2x2,
// wave tile 64 x 32*TN), BK = 64, two workgroups per CU (2 x 80 KB LDS) so one workgroup's barrier / epilogue hides
// under the other's MFMAs.
//   MFMA: v_mfma_f32_16x16x32_bf16 with SWAPPED operands (W fragment as A, activation fragment as B), so an
//   accumulator lane holds 4 consecutive output COLUMNS of one row: bias/GELU act on float4s and the bf16 result
//   packs to 8 bytes; the tile is then transposed through LDS and leaves as whole 16-B row segments.
//   LDS image: [row][64 bf16] (128-B rows), 16-B chunk index XOR-swizzled with (row>>1)&7: conflict-free for the
//   ds_read_b128 lane groups (each group holds 16 distinct rows with chunks c / c^1) and for the staging writes.
//   Staging: global_load_dwordx4 -> registers -> ds_write_b128 one K-step ahead (double-buffered LDS).
#include "common.h"
#include <stdlib.h>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef PANGU_GELU_BWD_H_PATCH
#define PANGU_GELU_BWD_H_PATCH 1      // the GELU-backward epilogue's h output leaves through the LDS patch as whole 16-B row segments (0: 8-B
                                      // pieces straight from the MFMA layout; measured 0.57 vs 0.67 ms at C = 192, 0.36 vs 0.41 at C = 384)
#endif
// GEMM_ABLATE (tools/ablate_gemm.py, timing only, LDS-DMA kernel): 1 no in-loop requests, 2 and no barrier, 3 and no fragment reads,
// 4 everything but the epilogue, 5 = 1 without the epilogue
constexpr int BBM = 128;
constexpr int BBK = 64;

__device__ inline u16 f2bf(float f) {   // round-to-nearest-even, NaN-preserving via the hardware convert
  return __builtin_bit_cast(u16, (__bf16)f);
}
__device__ inline unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }

__device__ inline int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }   // byte offset

template <int TN, int ACT, bool HAS_BIAS, bool OUT_F32>
__device__ __forceinline__ void bf16_epilogue(f32x4 (&acc)[4][2 * TN], unsigned char* smem, const float* __restrict__ bias,
                                              void* __restrict__ Cv, int ldc, int M, int N, u16* __restrict__ aux, int m0,
                                              int n0, int wave, int lane, u16* __restrict__ aux2 = nullptr) {
  const int wm = wave >> 1, wn = wave & 1;
  const int lc = lane & 15, lg = lane >> 4;
  // ---- epilogue.  lane (lg, lc) of tile (i, j) holds C[m = wm*64 + 16i + lc][n = wn*32TN + 16j + 4lg + r], r = 0..3
  const int wave_n0 = n0 + wn * 32 * TN;
  const int wave_m0 = m0 + wm * 64;
  if (OUT_F32) {
    float* C = reinterpret_cast<float*>(Cv);
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        C, 0, (int)(((size_t)(M - 1) * ldc + N) * sizeof(float)), 0x00020000);
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
      const int col = wave_n0 + j * 16 + lg * 4;
      const bool ok = col < N;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bias + (ok ? col : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = acc[i][j] + bv;
        if (ACT == PANGU_ACT_GELU) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = gelu_erf(v[c]);
        }
        const unsigned off = ok ? ((unsigned)(wave_m0 + i * 16 + lc) * (unsigned)ldc + (unsigned)col) * 4u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rsrc, (int)off, 0, 0);
      }
    }
    return;
  }
  // bf16 output: per-wave [64][32*TN] bf16 patch in LDS (row stride EP_LD bytes), written as 8-B packed quads,
  // read back as 16-B row segments -> buffer stores (rows >= M and columns >= N dropped by the range check)
  constexpr int ROWB = 32 * TN * 2;                        // bytes of payload per patch row
  constexpr int EP_LD = ROWB + 16;
  unsigned char* ep = smem + wave * (64 * EP_LD);
  u16* C = reinterpret_cast<u16*>(Cv);
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      C, 0, (int)(((size_t)(M - 1) * ldc + N) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      aux, 0, aux ? (int)((size_t)M * N * sizeof(u16)) : 0, 0x00020000);
  constexpr int CPR = ROWB / 16;                           // 16-B chunks per patch row
  const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      aux2, 0, ACT == PANGU_ACT_GELU_BWD_H ? (int)((size_t)M * N * sizeof(u16)) : 0, 0x00020000);
  if (ACT == PANGU_ACT_GELU_BWD || ACT == PANGU_ACT_GELU_BWD_H || ACT == PANGU_ACT_ADD) {
    // stage the saved pre-activation (or addend) patch first (coalesced 16-B loads), so gelu' can be applied in the MFMA layout
#pragma unroll
    for (int it = 0; it < CPR; ++it) {
      const int f = lane + 64 * it, row = f / CPR, ch = f - row * CPR;
      const int col = wave_n0 + ch * 8;
      const unsigned off = col < N ? ((unsigned)(wave_m0 + row) * (unsigned)N + (unsigned)col) * 2u : 0xFFFFFFFFu;
      *reinterpret_cast<u32x4*>(ep + row * EP_LD + ch * 16) = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)off, 0, 0);
    }
  }
  // 16-row groups: finish (bias/activation/pack -> LDS) one group, then read it back as whole 16-B row segments and
  // store it, so the stores of group i overlap the activation VALU work of group i+1.
  f32x4 bv[2 * TN];
#pragma unroll
  for (int j = 0; j < 2 * TN; ++j) {
    const int col = wave_n0 + j * 16 + lg * 4;
    bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) bv[j] = *reinterpret_cast<const f32x4*>(bias + (col < N ? col : 0));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    [[maybe_unused]] u32x2 hp[2 * TN];                                      // PANGU_GELU_BWD_H_PATCH: h of this row group, until the patch rows are free
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
      const int col = wave_n0 + j * 16 + lg * 4;
      f32x4 v = acc[i][j] + bv[j];
      unsigned char* slot = ep + (i * 16 + lc) * EP_LD + (j * 16 + lg * 4) * 2;
      if (ACT == PANGU_ACT_GELU_BWD) {
        const u32x2 xp = *reinterpret_cast<const u32x2*>(slot);
        v[0] *= gelu_erf_grad_lp(__builtin_bit_cast(float, xp[0] << 16));
        v[1] *= gelu_erf_grad_lp(__builtin_bit_cast(float, xp[0] & 0xFFFF0000u));
        v[2] *= gelu_erf_grad_lp(__builtin_bit_cast(float, xp[1] << 16));
        v[3] *= gelu_erf_grad_lp(__builtin_bit_cast(float, xp[1] & 0xFFFF0000u));
      }
      if (ACT == PANGU_ACT_GELU_BWD_H) {
        // gelu'(x) = Phi(x) + x phi(x) and h = gelu(x) = x Phi(x) share Phi: h leaves as an 8-B piece per lane (the 4 lanes
        // of a row make one 32-B segment), like the pre-activation of the forward epilogue
        const u32x2 xp = *reinterpret_cast<const u32x2*>(slot);
        const f32x4 x = {__builtin_bit_cast(float, xp[0] << 16), __builtin_bit_cast(float, xp[0] & 0xFFFF0000u),
                         __builtin_bit_cast(float, xp[1] << 16), __builtin_bit_cast(float, xp[1] & 0xFFFF0000u)};
        f32x4 hh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float phi_c = 0.5f * (1.0f + erf_poly(x[c] * 0.70710678118654752440f));
          hh[c] = x[c] * phi_c;
          v[c] *= phi_c + x[c] * 0.3989422804014327f * __expf(-0.5f * x[c] * x[c]);
        }
#if PANGU_GELU_BWD_H_PATCH
        hp[j] = u32x2{pack2(hh[0], hh[1]), pack2(hh[2], hh[3])};
#else
        const unsigned ho = col < N ? ((unsigned)(wave_m0 + i * 16 + lc) * (unsigned)N + (unsigned)col) * 2u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(hh[0], hh[1]), pack2(hh[2], hh[3])}, h_rsrc, (int)ho, 0, 0);
#endif
      }
      if (ACT == PANGU_ACT_ADD) {
        const u32x2 xp = *reinterpret_cast<const u32x2*>(slot);
        v[0] += __builtin_bit_cast(float, xp[0] << 16);
        v[1] += __builtin_bit_cast(float, xp[0] & 0xFFFF0000u);
        v[2] += __builtin_bit_cast(float, xp[1] << 16);
        v[3] += __builtin_bit_cast(float, xp[1] & 0xFFFF0000u);
      }
      if (ACT == PANGU_ACT_GELU) {
        if (aux) {                                          // pre-activation out (rounded to bf16, as it will be re-read)
          const unsigned xo = col < N ? ((unsigned)(wave_m0 + i * 16 + lc) * (unsigned)N + (unsigned)col) * 2u : 0xFFFFFFFFu;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])}, x_rsrc, (int)xo, 0, 0);
        }
        v = gelu_erf_lp4(v);
      }
      *reinterpret_cast<u32x2*>(slot) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
    }
    // read back rows 16i .. 16i+15 of the patch: 16 * CPR chunks of 16 B
#pragma unroll
    for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
      const int f = lane + 64 * it;
      const int row = i * 16 + f / CPR, ch = f % CPR;
      const int col = wave_n0 + ch * 8;
      if (f < 16 * CPR) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * EP_LD + ch * 16);
        const unsigned off = col < N ? ((unsigned)(wave_m0 + row) * (unsigned)ldc + (unsigned)col) * 2u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, (int)off, 0, 2);
      }
    }
#if PANGU_GELU_BWD_H_PATCH
    if (ACT == PANGU_ACT_GELU_BWD_H) {                     // h through the same patch rows: whole 16-B row segments instead of 8-B pieces
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j)
        *reinterpret_cast<u32x2*>(ep + (i * 16 + lc) * EP_LD + (j * 16 + lg * 4) * 2) = hp[j];
#pragma unroll
      for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
        const int f = lane + 64 * it;
        const int row = i * 16 + f / CPR, ch = f % CPR;
        const int col = wave_n0 + ch * 8;
        if (f < 16 * CPR) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * EP_LD + ch * 16);
          const unsigned off = col < N ? ((unsigned)(wave_m0 + row) * (unsigned)N + (unsigned)col) * 2u : 0xFFFFFFFFu;
          __builtin_amdgcn_raw_buffer_store_b128(v, h_rsrc, (int)off, 0, 0);
        }
      }
    }
#endif
  }
}

template <int TN, int ACT, bool HAS_BIAS, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(const u16* __restrict__ A, int lda,
                                                              const u16* __restrict__ W, const float* __restrict__ bias,
                                                              void* __restrict__ Cv, int ldc, int M, int N, int K,
                                                              int m_tiles, int n_tiles, u16* __restrict__ aux, u16* __restrict__ aux2) {
  constexpr int BN = 64 * TN;
  constexpr int STAGE = (BBM + BN) * 128;                 // bytes per LDS stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int m_tile = (local / n_tiles) * 8 + xcd;
  const int n_tile = local % n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BBM, n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lc = lane & 15, lg = lane >> 4;

  // staging: (BBM + BN) rows x 8 chunks of 16 B; chunk id f = tid + 256*i -> row f>>3, chunk f&7
  constexpr int NCH = (BBM + BN) * 8 / 256;               // 4 + 2*TN chunks per thread
  const u16* src[NCH];
  int dst[NCH];
  const int kchunk = tid & 7;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int row = (tid >> 3) + 32 * i;                   // 0 .. BBM+BN-1
    if (row < BBM) {
      int r = m0 + row;
      r = r < M ? r : M - 1;
      src[i] = A + (size_t)r * lda + kchunk * 8;
    } else {
      int r = n0 + row - BBM;
      r = r < N ? r : N - 1;
      src[i] = W + (size_t)r * K + kchunk * 8;
    }
    dst[i] = (row < BBM ? swz(row, kchunk) : BBM * 128 + swz(row - BBM, kchunk));
  }
  const int KT = (K + BBK - 1) / BBK;
  u32x4 stg[NCH];
  auto fetch = [&](int kt) {
    const bool in = kt * BBK + kchunk * 8 < K;             // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      stg[i] = in ? *reinterpret_cast<const u32x4*>(src[i] + (size_t)kt * BBK) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(smem + buf * STAGE + dst[i]) = stg[i];
  };

  f32x4 acc[4][2 * TN];                                    // [m tile 16][n tile 16], lane: 4 consecutive n of row m=lc
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
    const unsigned char* Ws = As + BBM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fw[2 * TN];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(As + swz(wm * 64 + i * 16 + lc, kk * 4 + lg));
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j)
        fw[j] = *reinterpret_cast<const bf16x8*>(Ws + swz(wn * 32 * TN + j * 16 + lc, kk * 4 + lg));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }
  bf16_epilogue<TN, ACT, HAS_BIAS, OUT_F32>(acc, smem, bias, Cv, ldc, M, N, aux, m0, n0, wave, lane, aux2);
}

// ---- direct-to-LDS variant -------------------------------------------------------------------------------------------
// Same tile and fragment scheme, but the operands travel HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (no staging VGPRs,
// no ds_write pass), BK = 32 and a FOUR-stage LDS ring (4 x 20 KB, still two workgroups per CU): loads run three K-steps
// ahead of the MFMAs behind counted s_waitcnt vmcnt(N) and ONE raw s_barrier per step.  An LDS-DMA wave-instruction
// writes 1 KB linearly (16 rows x 64 B), so the bank swizzle is applied on the SOURCE side: the lane that fills
// physical chunk p of row r fetches logical chunk p ^ F(r).  Rows >= M / >= N fall outside the descriptor's range
// and arrive as zeros.
constexpr int GBK = 32;

__device__ inline int kswz64(int row, int chunk) {      // 64-byte rows, 4 chunks: F = {0,2,3,1}[(row>>2)&3]
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((chunk ^ f) << 4);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// RING = LDS ring depth: 4 (80 KB, two workgroups per CU, loads three K-steps ahead) or 2 (40 KB ring, 53 KB with the
// epilogue patches: THREE workgroups per CU at <= 168 VGPRs -- there are no staging registers -- loads one K-step ahead)
template <int TN, int ACT, bool HAS_BIAS, bool OUT_F32, int RING>
__global__ __launch_bounds__(256, RING == 2 ? 3 : 2) void gemm_tn_bf16_glds_kernel(const u16* __restrict__ A, int lda,
                                                                   const u16* __restrict__ W, const float* __restrict__ bias,
                                                                   void* __restrict__ Cv, int ldc, int M, int N, int K,
                                                                   int m_tiles, int n_tiles, u16* __restrict__ aux, u16* __restrict__ aux2) {
  constexpr int BN = 64 * TN;
  constexpr int ROWS = BBM + BN;
  constexpr int STAGE = ROWS * 64;                        // bytes per ring slot
  constexpr int LPS = ROWS / 64;                          // LDS-DMA instructions per wave and stage (16 rows each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int m_tile = (local / n_tiles) * 8 + xcd;
  const int n_tile = local % n_tiles;
  if (m_tile >= m_tiles) return;
  const int m0 = m_tile * BBM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lc = lane & 15, lg = lane >> 4;

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(W), 0, (int)((size_t)N * K * sizeof(u16)), 0x00020000);
  // instruction q = i*4 + wave (i < LPS) fills rows 16q .. 16q+15; this lane fills (row 16q + lane>>2, chunk lane&3)
  unsigned voff[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = 16 * (i * 4 + wave) + (lane >> 2);
    const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
    const int c = (lane & 3) ^ f;                           // logical chunk this physical slot must receive
    voff[i] = row < BBM ? ((unsigned)(m0 + row) * (unsigned)lda + c * 8) * 2u
                        : ((unsigned)(n0 + row - BBM) * (unsigned)K + c * 8) * 2u;
  }
  auto issue = [&](int kt) {
    unsigned char* base = smem + (kt % RING) * STAGE;
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int q = i * 4 + wave;
      auto dst = (__attribute__((address_space(3))) void*)(base + q * 1024);
      if (16 * q < BBM) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)voff[i], kt * GBK * 2, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, (int)voff[i], kt * GBK * 2, 0, 0);
    }
  };

  f32x4 acc[4][2 * TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = K / GBK;
#pragma unroll
  for (int s0 = 0; s0 < RING - 1; ++s0)
    if (s0 < KT) issue(s0);
  for (int kt = 0; kt < KT; ++kt) {
    const int rem = KT - 1 - kt;                           // newer steps already in flight: min(rem, RING - 2)
    if (RING >= 4 && rem >= 2) wait_vmcnt<2 * LPS>();
    else if (RING >= 3 && rem >= 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
#ifdef GEMM_ABLATE
    if (GEMM_ABLATE < 2 || GEMM_ABLATE == 4)
#endif
    __builtin_amdgcn_s_barrier();                          // step kt landed for every wave; slot (kt-1)%4 is free
    asm volatile("" ::: "memory");
#ifdef GEMM_ABLATE
    if (GEMM_ABLATE == 4)
#endif
    if (kt + RING - 1 < KT) issue(kt + RING - 1);
    const unsigned char* As = smem + (kt % RING) * STAGE;
    const unsigned char* Ws = As + BBM * 64;
    bf16x8 fa[4], fw[2 * TN];
#ifdef GEMM_ABLATE
    if (GEMM_ABLATE == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(fa[i][0]) : "v"(kt));
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(fw[j][0]) : "v"(kt));
    } else {
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + kswz64(wm * 64 + i * 16 + lc, lg));
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(Ws + kswz64(wn * 32 * TN + j * 16 + lc, lg));
#ifdef GEMM_ABLATE
    }
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
  }
  __syncthreads();                                         // every wave is done with the ring before the epilogue reuses it
#ifdef GEMM_ABLATE
  if (GEMM_ABLATE == 4 || GEMM_ABLATE == 5) {              // timing only: no epilogue (one store keeps the accumulators alive)
    float t = 0.f;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 2 * TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) reinterpret_cast<u16*>(Cv)[0] = 1;
    return;
  }
#endif
  bf16_epilogue<TN, ACT, HAS_BIAS, OUT_F32>(acc, smem, bias, Cv, ldc, M, N, aux, m0, n0, wave, lane, aux2);
}

template <int TN, bool OUT_F32>
int launch_bf16(hipStream_t s, const u16* A, int lda, const u16* W, const float* bias, void* C, int ldc, int M, int N,
                int K, int act, u16* aux, u16* aux2 = nullptr) {
  constexpr int BN = 64 * TN;
  const int m_tiles = (M + BBM - 1) / BBM, n_tiles = (N + BN - 1) / BN;
  const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  // Staging pipelines, A/B-measured on MI355X at this model's shapes (tools/bench_kernels.py gemm_bf16): the LDS-DMA ring of 2
  // runs THREE workgroups per CU (no staging registers, 53 KB of LDS) and wins 7-13 % on the K = 384 / 768 shapes; for
  // K >= 1024 the register-staged BK = 64 kernel (two workgroups per CU, half the barriers) is 4 % ahead; the ring of 4
  // equals the register-staged kernel everywhere.  PANGU_BF16_GLDS = 0 / 1 / 2 forces register staging / ring of 4 / ring of 2.
  const bool glds = K < 1024 && (K % GBK == 0);      // (a ring of 4 measured equal to register staging: removed)
  const size_t epi = OUT_F32 ? 0 : (size_t)4 * 64 * (32 * TN * 2 + 16);
  const size_t ring = (size_t)2 * (BBM + BN) * 64;
  const size_t shm = glds ? (ring > epi ? ring : epi) : 2 * (size_t)(BBM + BN) * 128;
  dim3 g(grid), blk(256);
#define PANGU_BGEMM(ACT, HB)                                                                                          \
  do {                                                                                                                \
    if (glds) {                                                                                                       \
      auto kern = gemm_tn_bf16_glds_kernel<TN, ACT, HB, OUT_F32, 2>;                                                  \
      PANGU_ENSURE_DYN_LDS(kern, shm);                                                                                \
      hipLaunchKernelGGL(kern, g, blk, shm, s, A, lda, W, bias, C, ldc, M, N, K, m_tiles, n_tiles, aux, aux2);        \
      break;                                                                                                          \
    }                                                                                                                 \
    auto kern = gemm_tn_bf16_kernel<TN, ACT, HB, OUT_F32>;                                                            \
    PANGU_ENSURE_DYN_LDS(kern, shm);                                                                                  \
    hipLaunchKernelGGL(kern, g, blk, shm, s, A, lda, W, bias, C, ldc, M, N, K, m_tiles, n_tiles, aux, aux2);          \
  } while (0)
  if (act == PANGU_ACT_GELU) {
    if (bias) PANGU_BGEMM(PANGU_ACT_GELU, true); else PANGU_BGEMM(PANGU_ACT_GELU, false);
  } else if (act == PANGU_ACT_GELU_BWD_H) {
    if constexpr (OUT_F32) return PANGU_E_ARG; else PANGU_BGEMM(PANGU_ACT_GELU_BWD_H, false);
  } else if (act == PANGU_ACT_GELU_BWD) {
    if (OUT_F32) return PANGU_E_ARG;
    PANGU_BGEMM(PANGU_ACT_GELU_BWD, false);
  } else if (act == PANGU_ACT_ADD) {
    if (OUT_F32) return PANGU_E_ARG;
    if (bias) PANGU_BGEMM(PANGU_ACT_ADD, true); else PANGU_BGEMM(PANGU_ACT_ADD, false);
  } else {
    if (bias) PANGU_BGEMM(PANGU_ACT_NONE, true); else PANGU_BGEMM(PANGU_ACT_NONE, false);
  }
#undef PANGU_BGEMM
  return pangu_launch_status();
}

}  // namespace

int pangu_linear_ws_bf16(hipStream_t s, const void* A, int lda, const void* W, const float* bias, void* C, int ldc, int M,
                         int N, int K, int act, void* aux, int out_f32, void* aux2 = nullptr);      // gemm_ws_bf16.hip

extern "C" int pangu_linear_gelu_bwd_bf16(pangu_stream_t stream, const void* A, int lda, const void* W, void* dpre, int ldc,
                                          int M, int N, int K, const void* pre, void* h) {
  if (!A || !W || !dpre || !pre) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 7) || lda < K || ldc < N || (lda & 7) || (ldc & 7)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lda, 2) || !pangu_fits_u32(M, ldc, 2) || !pangu_fits_u32(M, N, 2)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  const int act = h ? PANGU_ACT_GELU_BWD_H : PANGU_ACT_GELU_BWD;
  // K <= 192: the weights-stationary kernel re-reads dm once per 192-column slice (4x at N = 768) next to the pre-activation
  // read and the dpre / h writes; measured (tools/bench_kernels.py mlp_train, MI355X) the tiled LDS-DMA kernel is 7-13 % ahead
  // on this epilogue (0.677 vs 0.729 ms with h, 0.479 vs 0.553 without), so it is the default; PANGU_BF16_WS_GELU_BWD=1 = the
  // weights-stationary kernel
  const u16* a = (const u16*)A;
  const u16* w = (const u16*)W;
  u16* x = (u16*)const_cast<void*>(pre);
  // with h the epilogue moves three (tokens x N) tensors per tile: 128 x 128 tiles (116 VGPRs, 37 KB of patches: four workgroups
  // per CU cover its load -> gelu' -> store chain) beat 128 x 192 (three per CU): 0.53 vs 0.57 ms at C = 192, 0.34 vs 0.36 at
  // C = 384 (tools/bench_kernels.py mlp_train); without h the two are level.  PANGU_GELU_BWD_TN=3 / 2 forces one.
  const bool tn2 = h != nullptr;
  if (tn2 && N % 128 == 0) return launch_bf16<2, false>(s, a, lda, w, nullptr, dpre, ldc, M, N, K, act, x, (u16*)h);
  if ((N % 192 == 0) || (N > 128 && N < 192)) return launch_bf16<3, false>(s, a, lda, w, nullptr, dpre, ldc, M, N, K, act, x, (u16*)h);
  if (N % 128 == 0) return launch_bf16<2, false>(s, a, lda, w, nullptr, dpre, ldc, M, N, K, act, x, (u16*)h);
  return launch_bf16<1, false>(s, a, lda, w, nullptr, dpre, ldc, M, N, K, act, x, (u16*)h);
}

extern "C" int pangu_linear_fwd_bf16(pangu_stream_t stream, const void* A, int lda, const void* W, const float* bias,
                                     void* C, int ldc, int M, int N, int K, int act, void* aux, int out_dtype) {
  if (!A || !W || !C) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 7) || lda < K || ldc < N || (lda & 7) || (ldc & 3)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lda, 2) || !pangu_fits_u32(M, ldc, 4)) return PANGU_E_RANGE;
  if (act != PANGU_ACT_NONE && act != PANGU_ACT_GELU && act != PANGU_ACT_GELU_BWD && act != PANGU_ACT_ADD) return PANGU_E_ARG;
  if (act == PANGU_ACT_GELU_BWD && (!aux || bias)) return PANGU_E_ARG;
  if (act == PANGU_ACT_ADD && !aux) return PANGU_E_ARG;
  if (out_dtype != PANGU_BF16 && out_dtype != PANGU_F32) return PANGU_E_DTYPE;
  if (out_dtype == PANGU_BF16 && (ldc & 7)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  // K <= 192: weights-stationary barrier-free kernel (gemm_ws_bf16.hip; not for the GELU-backward epilogue: the tiled LDS-DMA
  // kernel is 7-13 % ahead there, see pangu_linear_gelu_bwd_bf16)
  if (M >= 4096 && act != PANGU_ACT_ADD && act != PANGU_ACT_GELU_BWD) {
    const int rc = pangu_linear_ws_bf16(s, A, lda, W, bias, C, ldc, M, N, K, act, aux, out_dtype == PANGU_F32);
    if (rc != PANGU_E_SHAPE) return rc;
  }
  const u16* a = (const u16*)A;
  const u16* w = (const u16*)W;
  u16* x = (u16*)aux;
  const bool wide = (N % 192 == 0) || (N > 128 && N < 192);
  if (out_dtype == PANGU_F32) {
    if (wide) return launch_bf16<3, true>(s, a, lda, w, bias, C, ldc, M, N, K, act, x);
    if (N % 128 == 0) return launch_bf16<2, true>(s, a, lda, w, bias, C, ldc, M, N, K, act, x);
    return launch_bf16<1, true>(s, a, lda, w, bias, C, ldc, M, N, K, act, x);
  }
  if (wide) return launch_bf16<3, false>(s, a, lda, w, bias, C, ldc, M, N, K, act, x);
  if (N % 128 == 0) return launch_bf16<2, false>(s, a, lda, w, bias, C, ldc, M, N, K, act, x);
  return launch_bf16<1, false>(s, a, lda, w, bias, C, ldc, M, N, K, act, x);
}
