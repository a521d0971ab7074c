// bf16 weight-gradient GEMM for gfx950:  dW[N,K] (fp32) += dC[M,N]^T @ A[M,K],  db[N] += colsum(dC), bf16 operands.
//
// Contraction over the token dimension M: both operands are stored token-major, but v_mfma_f32_16x16x32_bf16 wants 8
// consecutive contraction indices (tokens) per lane.  The staging loader therefore TRANSPOSES on the fly: a thread
// loads an 8-token x 8-column block (eight 16-B loads, coalesced across threads along the columns), transposes it in
// registers with v_perm_b32, and writes eight 16-B chunks "column c, tokens t..t+7" into the same swizzled
// [row][64] LDS image the forward GEMM uses; fragment reads are then plain conflict-free ds_read_b128.
// Grid = (128 x 64*TK output tiles) x (M splits); each workgroup adds its tile to dW with fp32 no-return atomics.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int WB_N = 128;      // output rows (n) per tile
constexpr int WB_M = 64;       // tokens per K-step

__device__ inline int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 8x8 transpose of 16-bit elements: r[i] = row i (8 elements in 4 dwords) -> c[j] = column j (rows 0..7)
__device__ inline void transpose8x8(const u32x4 (&r)[8], u32x4 (&c)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int p = 0; p < 4; ++p)
      c[j][p] = __builtin_amdgcn_perm(r[2 * p + 1][j >> 1], r[2 * p][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
}

template <int TK>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const u16* __restrict__ dC, int lddc, const u16* __restrict__ A,
                                                            int lda, float* __restrict__ dW, float* __restrict__ db, int M,
                                                            int N, int K, int n_tiles, int k_tiles, int rows_per_split) {
  constexpr int BKC = 64 * TK;
  constexpr int STAGE = (WB_N + BKC) * 128;
  constexpr int NBLK = (WB_N + BKC) / 8 * (WB_M / 8);     // 8x8 blocks per K-step: (16 + 8TK) * 8
  constexpr int BPT = (NBLK + 255) / 256;                  // blocks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tile = blockIdx.x % (n_tiles * k_tiles), split = blockIdx.x / (n_tiles * k_tiles);
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

  // block assignment: block b = tid + 256*i.  b < 128: dC block (token block mb = b>>4, column chunk b&15), i.e.
  // threads 0..127 at i = 0 (waves 0,1: wave-uniform); every other block is an A block
  // (b' = b-128: mb = b' / (8TK), column chunk b' % (8TK)).  Descriptor choice is made wave-uniform explicitly.
  const bool wave_d = __builtin_amdgcn_readfirstlane(tid) < 128;
  bool live[BPT], col_ok[BPT];
  int mb[BPT], cc[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int b = tid + 256 * i;
    live[i] = b < NBLK;
    if (i == 0 && b < 128) { mb[i] = b >> 4; cc[i] = b & 15; col_ok[i] = n0 + cc[i] * 8 < N; }
    else { const int bb = b - 128; mb[i] = bb / (8 * TK); cc[i] = bb - mb[i] * (8 * TK); col_ok[i] = k0 + cc[i] * 8 < K; }
  }

  u32x4 blk[BPT][8];
  float dbacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbacc[j] = 0.f;

  auto fetch = [&](int m) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const bool from_d = (i == 0) && wave_d;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const unsigned row = (unsigned)(m + mb[i] * 8 + r);
        if (from_d) {
          const unsigned off = col_ok[i] ? (row * (unsigned)lddc + (unsigned)(n0 + cc[i] * 8)) * 2u : 0xFFFFFFFFu;
          blk[i][r] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, (int)off, 0, 0);
        } else {
          const unsigned off = (col_ok[i] && live[i]) ? (row * (unsigned)lda + (unsigned)(k0 + cc[i] * 8)) * 2u : 0xFFFFFFFFu;
          blk[i][r] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0);
        }
      }
    }
  };
  auto stash = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const bool from_d = (i == 0) && wave_d;
      u32x4 col[8];
      transpose8x8(blk[i], col);
      if (live[i]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *reinterpret_cast<u32x4*>(base + (from_d ? 0 : WB_N * 128) + swz(cc[i] * 8 + j, mb[i])) = col[j];
      }
      if (from_d && db != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float sm = 0.f;
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            sm += __builtin_bit_cast(float, col[j][p] << 16);
            sm += __builtin_bit_cast(float, col[j][p] & 0xFFFF0000u);
          }
          dbacc[j] += sm;
        }
      }
    }
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
    const unsigned char* As = Ds + WB_N * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fd[4], fa[2 * TK];
#pragma unroll
      for (int i = 0; i < 4; ++i) fd[i] = *reinterpret_cast<const bf16x8*>(Ds + swz(wn * 64 + i * 16 + lc, kk * 4 + lg));
#pragma unroll
      for (int j = 0; j < 2 * TK; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(As + swz(wk * 32 * TK + j * 16 + lc, kk * 4 + lg));
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
    float* red = reinterpret_cast<float*>(smem);        // [8 token blocks][128 columns]
    __syncthreads();
    if (tid < 128) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(tid >> 4) * WB_N + (tid & 15) * 8 + j] = dbacc[j];
    }
    __syncthreads();
    if (tid < WB_N) {
      float v = 0.f;
#pragma unroll
      for (int gq = 0; gq < 8; ++gq) v += red[gq * WB_N + tid];
      if (n0 + tid < N) atomicAdd(&db[n0 + tid], v);
    }
  }
}

template <int TK>
int launch_wgrad_bf16(hipStream_t s, const u16* dC, int lddc, const u16* A, int lda, float* dW, float* db, int M, int N,
                      int K) {
  constexpr int BKC = 64 * TK;
  const int n_tiles = (N + WB_N - 1) / WB_N, k_tiles = (K + BKC - 1) / BKC;
  const int tiles = n_tiles * k_tiles;
  int split = (1024 + tiles - 1) / tiles;
  int rows = ((M + split - 1) / split + WB_M - 1) / WB_M * WB_M;
  if (rows < 4 * WB_M) rows = 4 * WB_M;
  split = (M + rows - 1) / rows;
  const size_t shm = 2 * (size_t)(WB_N + BKC) * 128;
  auto kern = wgrad_bf16_kernel<TK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles * split), dim3(256), shm, s, dC, lddc, A, lda, dW, db, M, N, K, n_tiles, k_tiles, rows);
  return pangu_launch_status();
}

}  // namespace

extern "C" int pangu_linear_wgrad_bf16(pangu_stream_t stream, const void* dC, int lddc, const void* A, int lda, float* dW,
                                       float* db, int M, int N, int K) {
  if (!dC || !A || !dW) return PANGU_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || lddc < N || lda < K || (lddc & 7) || (lda & 7)) return PANGU_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const u16* d = (const u16*)dC;
  const u16* a = (const u16*)A;
  if (K % 192 == 0) return launch_wgrad_bf16<3>(s, d, lddc, a, lda, dW, db, M, N, K);
  if (K % 128 == 0) return launch_wgrad_bf16<2>(s, d, lddc, a, lda, dW, db, M, N, K);
  return launch_wgrad_bf16<1>(s, d, lddc, a, lda, dW, db, M, N, K);
}
