// Weights-stationary bf16 projection GEMM for gfx950 — the K <= 384 projections of this model (qkv, proj, mlp.linear1,
// patch embed / recover, upsample) where M = 65k..521k tokens stream past a weight slice that fits in LDS.
//
// The tiled kernel (gemm_bf16.hip) spends most of its time parked at the one-barrier-per-K-step rendezvous: with 24-48
// MFMAs (0.4-0.8k cycles) between barriers its MFMA pipe is busy 20-36 % (rocprofv3 PMC).  Here nothing is shared
// between waves in the main loop, so there is NO barrier in it:
//   * a persistent 8-wave workgroup (one per CU) copies its weight slice W[n0 .. n0+BNW) x K (<= 72 KB) into LDS ONCE
//     (16-B chunks XOR-swizzled: conflict-free ds_read_b128 fragments) and then walks row tiles of 256 tokens;
//   * each wave owns 32 token rows x all BNW columns: its activation fragments (B operand, 16 B = 8 consecutive k of
//     one row per lane) go straight from global memory to VGPRs — all K/32 steps of a tile are requested up front,
//     and the next tile's are requested before the current tile's epilogue — while the weight fragments (A operand)
//     are re-read from the resident LDS image;
//   * v_mfma_f32_16x16x32_bf16 with swapped operands, so a lane owns 4 consecutive output columns: float4 bias / GELU,
//     8-byte packed bf16, then a 16-row LDS patch turns them into whole 16-B row segments for the stores.
#include "common.h"
#include <stdlib.h>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WS_WAVES = 8;
constexpr int WS_BM = 32 * WS_WAVES;          // 256 token rows per workgroup tile

__device__ inline u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ inline unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }

// byte offset of 16-B chunk `chunk` of weight row `row` (row pitch = KMAX*2 bytes, a multiple of 128)
// Row pitch 384 B (KMAX 192, = 128 mod 256): XOR the low 3 chunk bits with (row>>1)&7; pitch 768 B (KMAX 384, = 0 mod
// 256): XOR the low 4 chunk bits with row&15.  Either way the 16 rows of a ds_read_b128 lane group (which read chunks
// c and c^1) land on 16 distinct 16-B slots of the 256-B bank row.
template <int KMAX>
__device__ inline int wswz(int row, int chunk) {
  if constexpr ((KMAX * 2) % 256 == 0) return row * (KMAX * 2) + ((chunk ^ (row & 15)) << 4);
  else return row * (KMAX * 2) + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BNW, int KMAX, int ACT, bool HAS_BIAS, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_ws_bf16_kernel(const u16* __restrict__ A, int lda, const u16* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ Cv,
                                                              int ldc, int M, int N, int K, int n_slices, int m_tiles,
                                                              u16* __restrict__ aux, u16* __restrict__ aux2) {
  constexpr int NT = BNW / 16;                 // 16-column tiles per wave row-block
  constexpr int KS = KMAX / 32;                // K-steps of one MFMA (32 k-values)
  constexpr int ROWB = BNW * (OUT_F32 ? 4 : 2);
  constexpr int EP_LD = ROWB + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wl = smem;                                   // [BNW][KMAX] bf16, swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* bias_s = reinterpret_cast<float*>(smem + BNW * KMAX * 2);          // [BNW] fp32, resident like W
  unsigned char* ep = smem + BNW * KMAX * 2 + BNW * 4 + wave * (16 * EP_LD);
  const int lc = lane & 15, lg = lane >> 4;

  const int slice = blockIdx.x % n_slices;
  const int wg_in_slice = blockIdx.x / n_slices, wgs_per_slice = gridDim.x / n_slices;
  const int n0 = slice * BNW;
  const int ksteps = K / 32;                                  // <= KS

  // ---- resident weight slice: BNW rows x K/8 chunks (rows >= N and chunks >= K/8 are zero-filled)
  {
    const int cpr = KMAX / 8;
    for (int f = tid; f < BNW * cpr; f += 512) {
      const int row = f / cpr, ch = f - row * cpr;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (n0 + row < N && ch * 8 < K) v = *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + row) * K + ch * 8);
      *reinterpret_cast<u32x4*>(Wl + wswz<KMAX>(row, ch)) = v;
    }
  }
  if (tid < BNW) bias_s[tid] = (HAS_BIAS && n0 + tid < N) ? bias[n0 + tid] : 0.f;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      Cv, 0, (int)(((size_t)(M - 1) * ldc + N) * (OUT_F32 ? 4 : 2)), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      aux, 0, aux ? (int)((size_t)M * N * sizeof(u16)) : 0, 0x00020000);

  // activation fragments of one tile: [k-step][row sub-tile]; rows past M read as zeros (range-checked descriptor)
  u32x4 af[KS][2];
  auto fetch = [&](int tile) {
    const int m = tile * WS_BM + wave * 32;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const unsigned off = ((unsigned)(m + mt * 16 + lc) * (unsigned)lda + (unsigned)(ks * 32 + lg * 8)) * 2u;
        af[ks][mt] = ks < ksteps ? __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0) : u32x4{0u, 0u, 0u, 0u};
      }
  };

  int tile = wg_in_slice;
  if (tile < m_tiles) fetch(tile);
  for (; tile < m_tiles; tile += wgs_per_slice) {
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < ksteps) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[ks][0]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[ks][1]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wl + wswz<KMAX>(j * 16 + lc, ks * 4 + lg));
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a0, acc[0][j], 0, 0, 0);     // D[n][m]
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a1, acc[1][j], 0, 0, 0);
        }
      }
    }
    const int m_cur = tile * WS_BM + wave * 32;
    if (tile + wgs_per_slice < m_tiles) fetch(tile + wgs_per_slice);      // next tile's loads fly under the epilogue

    // ---- epilogue: lane (lg, lc) of (mt, j) holds C[m_cur + 16 mt + lc][n0 + 16 j + 4 lg + r]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      constexpr int CPR = ROWB / 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + j * 16 + lg * 4;
        f32x4 v = acc[mt][j];
        if (HAS_BIAS) v += *reinterpret_cast<const f32x4*>(bias_s + j * 16 + lg * 4);   // LDS: no vmcnt coupling with the in-flight next-tile loads
        if (ACT == PANGU_ACT_GELU) {
          if (aux) {
            const unsigned xo = col < N ? ((unsigned)(m_cur + mt * 16 + lc) * (unsigned)N + (unsigned)col) * 2u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])}, x_rsrc, (int)xo, 0, 0);
          }
          v = gelu_erf_lp4(v);
        }
        if (OUT_F32) *reinterpret_cast<f32x4*>(ep + lc * EP_LD + (j * 16 + lg * 4) * 4) = v;
        else *reinterpret_cast<u32x2*>(ep + lc * EP_LD + (j * 16 + lg * 4) * 2) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
      }
#pragma unroll
      for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
        const int f = lane + 64 * it, row = f / CPR, ch = f % CPR;
        const int col = n0 + ch * (OUT_F32 ? 4 : 8);
        if (f < 16 * CPR) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * EP_LD + ch * 16);
          const unsigned off = col < N ? ((unsigned)(m_cur + mt * 16 + row) * (unsigned)ldc + (unsigned)col) * (OUT_F32 ? 4u : 2u)
                                       : 0xFFFFFFFFu;
          __builtin_amdgcn_raw_buffer_store_b128(v, c_rsrc, (int)off, 0, 2);
        }
      }
    }
  }
}

template <int BNW, int KMAX, bool OUT_F32>
int launch_ws(hipStream_t s, const u16* A, int lda, const u16* W, const float* bias, void* C, int ldc, int M, int N, int K,
              int act, u16* aux, u16* aux2) {
  const int n_slices = (N + BNW - 1) / BNW;
  const int m_tiles = (M + WS_BM - 1) / WS_BM;
  int per_slice = 256 / n_slices;                       // one persistent workgroup per CU
  if (per_slice < 1) per_slice = 1;
  if (per_slice > m_tiles) per_slice = m_tiles;
  const int grid = per_slice * n_slices;
  const size_t shm = (size_t)BNW * KMAX * 2 + BNW * 4 + (size_t)WS_WAVES * 16 * (BNW * (OUT_F32 ? 4 : 2) + 16);
#define PANGU_WS(ACT, HB)                                                                                             \
  do {                                                                                                                \
    auto kern = gemm_ws_bf16_kernel<BNW, KMAX, ACT, HB, OUT_F32>;                                                     \
    PANGU_ENSURE_DYN_LDS(kern, shm);                                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, s, A, lda, W, bias, C, ldc, M, N, K, n_slices, m_tiles, aux, aux2); \
  } while (0)
  if (act == PANGU_ACT_GELU) {
    if (bias) PANGU_WS(PANGU_ACT_GELU, true); else PANGU_WS(PANGU_ACT_GELU, false);
  } else {
    if (bias) PANGU_WS(PANGU_ACT_NONE, true); else PANGU_WS(PANGU_ACT_NONE, false);
  }
#undef PANGU_WS
  return pangu_launch_status();
}

}  // namespace

// Internal entry used by pangu_linear_fwd_bf16's dispatcher (same argument meaning); returns PANGU_E_SHAPE when the shape is
// not one this kernel covers.
int pangu_linear_ws_bf16(hipStream_t s, const void* A, int lda, const void* W, const float* bias, void* C, int ldc, int M,
                         int N, int K, int act, void* aux, int out_f32, void* aux2) {
  // Measured (tools/bench_kernels.py gemm_bf16, MI355X): with K <= 192 a 192-column slice is resident and the kernel beats
  // the tiled one by 5-25 %; with K = 384 only 96 columns fit (activations re-read twice as often) and it LOSES 20-40 %
  // (instantiation removed in round 4), as it does on the GELU-backward epilogue (7-13 %): those stay on the tiled kernel.
  if ((K & 31) || K > 192 || (lda & 7) || (N & 7)) return PANGU_E_SHAPE;
  if (out_f32) return PANGU_E_SHAPE;                        // fp32 patch of a 192-wide slice does not fit next to W
  if (act != PANGU_ACT_NONE && act != PANGU_ACT_GELU) return PANGU_E_SHAPE;
  return launch_ws<192, 192, false>(s, (const u16*)A, lda, (const u16*)W, bias, C, ldc, M, N, K, act, (u16*)aux, (u16*)aux2);
}
