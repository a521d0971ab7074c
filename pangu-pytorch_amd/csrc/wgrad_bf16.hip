// bf16 weight-gradient GEMM for gfx950:  dW[N,K] (fp32) += dC[M,N]^T @ A[M,K],  db[N] += colsum(dC), bf16 operands.
//
// Contraction over the token dimension M: both operands are stored token-major, but v_mfma_f32_16x16x32_bf16 wants 8
// contraction indices (tokens) per lane for one output row/column.  Instead of transposing in registers, the 64-token
// slabs of dC and A are copied into LDS exactly as they lie in memory ([token][column], straight 16-B chunk copies) and
// the fragments are fetched with gfx950's transposing LDS read ds_read_b64_tr_b16 (4 tokens x 16 columns per 16-lane
// group, column-major out): two reads give a lane its 8 tokens {4g..4g+3} and {16+4g..16+4g+3} of one column (the MFMA
// k index is permuted identically for both operands).  Row strides of 288 / 416 / 160 B (= 32 B mod 256) x this token
// assignment make every transposing read conflict-free in its 32-lane half.
// Grid = (128 x 64*TK output tiles) x (M splits); each workgroup adds its tile to dW with fp32 no-return atomics.
#include "common.h"
#include <stdlib.h>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int WB_N = 128;      // output rows (n) per tile
constexpr int D_LD = 288;      // bytes per token row of the dC slab image (256 payload)

__device__ inline bf16x8 tr_frag(const unsigned char* img, int ld, int row0, int col0, int lg, int lc) {
  const unsigned char* p = img + (row0 + 4 * lg + (lc >> 2)) * ld + (col0 + 4 * (lc & 3)) * 2;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 16 * ld));
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// WB_M = tokens per K-step: 64 (TK <= 2: 74 KB of LDS, two workgroups per CU) or 32 (TK = 3: 45 KB)
template <int TK, int WB_M>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const u16* __restrict__ dC, int lddc, const u16* __restrict__ A,
                                                            int lda, float* __restrict__ dW, float* __restrict__ db, int M,
                                                            int N, int K, int n_tiles, int k_tiles, int rows_per_split) {
  constexpr int BKC = 64 * TK;
  constexpr int A_LD = BKC * 2 + (TK == 3 ? 32 : (TK == 2 ? 32 : 32));     // 416 / 288 / 160 bytes
  constexpr int STAGE = WB_M * (D_LD + A_LD);
  constexpr int ACH = BKC / 8;                              // 16-B chunks per A row
  constexpr int NA = WB_M * ACH / 256;                      // A chunks per thread
  constexpr int ND = WB_M / 16;                             // dC chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware order (blocks b, b+8, b+16.. share an XCD and its L2): the output tiles of ONE token slab run
  // back to back on one XCD, so each dC / A slab is fetched from HBM once and re-read from that L2.
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int tile = local % (n_tiles * k_tiles), split = (local / (n_tiles * k_tiles)) * 8 + xcd;
  const int n_tile = tile / k_tiles, k_tile = tile - n_tile * k_tiles;
  const int n0 = n_tile * WB_N, k0 = k_tile * BKC;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lc = lane & 15, lg = lane >> 4;

  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(dC), 0, (int)(((size_t)(M - 1) * lddc + N) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(u16)), 0x00020000);

  // staging: dC slab 64 rows x 16 chunks (4 per thread: row = (tid>>4) + 16 i, chunk tid&15),
  //          A slab 64 rows x ACH chunks (NA per thread: f = tid + 256 i -> row f / ACH, chunk f % ACH)
  const int d_row = tid >> 4, d_ch = tid & 15;
  const bool d_ok = n0 + d_ch * 8 < N;
  int a_row[NA], a_ch[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = tid + 256 * i;
    a_row[i] = f / ACH;
    a_ch[i] = f - a_row[i] * ACH;
    a_ok[i] = k0 + a_ch[i] * 8 < K;
  }
  u32x4 rd[ND], ra[NA];
  f32x4 dbacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  auto fetch = [&](int m) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const unsigned off = d_ok ? ((unsigned)(m + d_row + 16 * i) * (unsigned)lddc + (unsigned)(n0 + d_ch * 8)) * 2u : 0xFFFFFFFFu;
      rd[i] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, (int)off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const unsigned off = a_ok[i] ? ((unsigned)(m + a_row[i]) * (unsigned)lda + (unsigned)(k0 + a_ch[i] * 8)) * 2u : 0xFFFFFFFFu;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0);
    }
  };
  auto stash = [&](int buf) {
    unsigned char* Ds = smem + buf * STAGE;
    unsigned char* As = Ds + WB_M * D_LD;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      *reinterpret_cast<u32x4*>(Ds + (d_row + 16 * i) * D_LD + d_ch * 16) = rd[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {      // column sums for the bias gradient: columns 8 d_ch .. 8 d_ch + 7
        dbacc[0][c] += __builtin_bit_cast(float, (c & 1) ? (rd[i][c >> 1] & 0xFFFF0000u) : (rd[i][c >> 1] << 16));
        dbacc[1][c] += __builtin_bit_cast(float, (c & 1) ? (rd[i][2 + (c >> 1)] & 0xFFFF0000u) : (rd[i][2 + (c >> 1)] << 16));
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(As + a_row[i] * A_LD + a_ch[i] * 16) = ra[i];
  };

  f32x4 acc[4][2 * TK];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(m_begin);
  stash(0);
  __syncthreads();
  const int steps = (m_end - m_begin + WB_M - 1) / WB_M;
  for (int st = 0; st < steps; ++st) {
    const bool more = st + 1 < steps;
    if (more) fetch(m_begin + (st + 1) * WB_M);
    const unsigned char* Ds = smem + (st & 1) * STAGE;
    const unsigned char* As = Ds + WB_M * D_LD;
#pragma unroll
    for (int kk = 0; kk < WB_M / 32; ++kk) {
      bf16x8 fd[4], fa[2 * TK];
#pragma unroll
      for (int i = 0; i < 4; ++i) fd[i] = tr_frag(Ds, D_LD, kk * 32, wn * 64 + i * 16, lg, lc);
#pragma unroll
      for (int j = 0; j < 2 * TK; ++j) fa[j] = tr_frag(As, A_LD, kk * 32, wk * 32 * TK + j * 16, lg, lc);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], fa[j], acc[i][j], 0, 0, 0);     // D[n][k]
    }
    if (more) stash((st + 1) & 1);
    __syncthreads();
  }

  // lane (lg, lc) of tile (i, j): dW[n = n0 + wn*64 + 16i + 4lg + r][k = k0 + wk*32TK + 16j + lc]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TK; ++j) {
      const int kc = k0 + wk * 32 * TK + j * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + lg * 4 + r;
        if (n < N && kc < K) atomicAdd(&dW[(size_t)n * K + kc], acc[i][j][r]);
      }
    }
  if (db != nullptr && k_tile == 0) {
    float* red = reinterpret_cast<float*>(smem);        // [16 row groups][128 columns]
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // dbacc[0][c] = column 8 d_ch + {0,1,2,3}[c] ... see stash(): element order within the 16-B chunk
      red[d_row * WB_N + d_ch * 8 + c] = dbacc[0][c];
      red[d_row * WB_N + d_ch * 8 + 4 + c] = dbacc[1][c];
    }
    __syncthreads();
    if (tid < WB_N) {
      float v = 0.f;
#pragma unroll
      for (int gq = 0; gq < 16; ++gq) v += red[gq * WB_N + tid];
      if (n0 + tid < N) atomicAdd(&db[n0 + tid], v);
    }
  }
}

template <int TK>
int launch_wgrad_bf16(hipStream_t s, const u16* dC, int lddc, const u16* A, int lda, float* dW, float* db, int M, int N,
                      int K) {
  constexpr int BKC = 64 * TK;
  constexpr int WB_M = TK == 3 ? 32 : 64;
  const int n_tiles = (N + WB_N - 1) / WB_N, k_tiles = (K + BKC - 1) / BKC;
  const int tiles = n_tiles * k_tiles;
  // ONE round of the 512 resident workgroups (2 per CU), as many M-splits as fit: every workgroup ends with 98 KB of fp32
  // atomics into dW, which costs 10-45 % of the kernel at these shapes, so fewer, longer workgroups win as long as no
  // second (partial) round appears (measured sweep, tools/bench_kernels.py wgrad_bf16).
  int split = (512 / tiles) & ~7;   // multiple of 8: equal share per XCD
  if (split < 8) split = 8;
  int rows = ((M + split - 1) / split + 63) / 64 * 64;
  if (rows < 256) rows = 256;
  split = ((M + rows - 1) / rows + 7) & ~7;                         // grid padded to whole XCD rounds (empty slabs exit)
  const size_t shm = 2 * (size_t)WB_M * (D_LD + BKC * 2 + 32);
  auto kern = wgrad_bf16_kernel<TK, WB_M>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  hipLaunchKernelGGL(kern, dim3(tiles * split), dim3(256), shm, s, dC, lddc, A, lda, dW, db, M, N, K, n_tiles, k_tiles, rows);
  return pangu_launch_status();
}

}  // namespace

// wgrad_bf16_dma.hip: LDS-DMA variant for K % 192 == 0; returns 1 when the shape is not covered
int pangu_linear_wgrad_bf16_dma(hipStream_t s, const unsigned short* dC, int lddc, const unsigned short* A, int lda,
                                float* dW, float* db, int M, int N, int K, int target, float* ws, size_t ws_bytes);

extern "C" int pangu_linear_wgrad_bf16(pangu_stream_t stream, const void* dC, int lddc, const void* A, int lda, float* dW,
                                       float* db, int M, int N, int K) {
  return pangu_linear_wgrad_bf16_ws(stream, dC, lddc, A, lda, dW, db, M, N, K, nullptr, 0);
}

extern "C" int pangu_linear_wgrad_bf16_ws(pangu_stream_t stream, const void* dC, int lddc, const void* A, int lda, float* dW,
                                          float* db, int M, int N, int K, float* workspace, long long workspace_bytes) {
  if (!dC || !A || !dW) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || lddc < N || lda < K || (lddc & 7) || (lda & 7)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lddc, 2) || !pangu_fits_u32(M, lda, 2)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  const u16* d = (const u16*)dC;
  const u16* a = (const u16*)A;
  // LDS-DMA kernel for the large products (tools/bench_kernels.py wgrad_bf16, tools/ablate_wgrad.py; round 4, with a workspace):
  // -14...-24 % at the qkv / MLP / down- and up-sampling shapes (>= 77 GFLOP) and -5 % at the C = 384 projections (38 GFLOP, N a
  // multiple of 384: one 12-wave tile column); the C = 192 projections and the patch embedding (38 GFLOP, N = 192: 4-wave tiles)
  // end in their partial tiles sooner than the extra workgroups pay (0.119 against 0.104 ms) and stay on the register-staged
  // kernel, like everything without a workspace below 100 GFLOP (fp32 atomic tail).
  constexpr int dma_target = 768;
  const bool ws_ok = workspace != nullptr && workspace_bytes > 0 && (reinterpret_cast<size_t>(workspace) & 15) == 0;
  const double flop = 2.0 * M * N * K;
  if (flop >= (ws_ok ? 6.0e10 : 1.0e11) || (ws_ok && N % 384 == 0 && flop >= 3.0e10)) {
    const int rc = pangu_linear_wgrad_bf16_dma(s, d, lddc, a, lda, dW, db, M, N, K, dma_target, ws_ok ? workspace : nullptr,
                                               ws_ok ? (size_t)workspace_bytes : 0);
    if (rc != 1) return rc;
  }
  if (K % 192 == 0) return launch_wgrad_bf16<3>(s, d, lddc, a, lda, dW, db, M, N, K);
  if (K % 128 == 0) return launch_wgrad_bf16<2>(s, d, lddc, a, lda, dW, db, M, N, K);
  return launch_wgrad_bf16<1>(s, d, lddc, a, lda, dW, db, M, N, K);
}
