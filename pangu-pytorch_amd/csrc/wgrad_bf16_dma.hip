// bf16 weight-gradient GEMM for gfx950, LDS-DMA variant:  dW[N,K] (fp32) += dC[M,N]^T @ A[M,K],  db[N] += colsum(dC).
//
// Same decomposition as wgrad_bf16.hip (128 x 192 output tile and one token slab per 256-thread workgroup,
// v_mfma_f32_16x16x32_bf16 fed by transposing LDS reads of token-major slabs, fp32 no-return atomics into dW, XCD-aware
// slab order), with the slabs travelling L2 -> LDS by LDS-DMA: no staging registers, no ds_write pass, so more
// workgroups fit a CU (the 16-cycle bf16 MFMAs leave a K-step of a few hundred cycles between barriers: what covers the
// barrier and LDS latency is other workgroups).
// The DMA writes lane-linear 16-B pieces, so rows cannot be padded; instead the 16-B chunks are XOR-swizzled on the
// SOURCE side (a lane fetches the logical chunk that belongs at its physical position):
//   dC slab rows of 256 B (= one bank period): physical chunk = logical ^ 2*(row & 7)   -> the 8 token rows a 32-lane
//     half reads (two 4-row blocks, 32 B per row) land on 8 different 32-B bank groups;
//   A slab rows of 384 B (odd rows start half a period later): physical chunk = logical ^ 2*((row >> 1) & 3).
// Both make every ds_read_b64_tr_b16 of the fragment reads conflict-free in its 32-lane half.
// The bias gradient is one more MFMA column: dC^T @ ones (k-tile-0 workgroups), free next to the 24 tile MFMAs.
// Only K % 192 == 0 (every block linear); other shapes stay on wgrad_bf16.hip.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int WB_N = 128;      // output rows (n) per tile
constexpr int BKC = 192;       // output columns (k) per tile
constexpr int D_ROW = WB_N * 2, A_ROW = BKC * 2;          // slab row bytes: 256 / 384

__device__ inline int d_swz(int row, int chunk) { return row * D_ROW + ((chunk ^ (2 * (row & 7))) << 4); }
__device__ inline int a_swz(int row, int chunk) { return row * A_ROW + ((chunk ^ (2 * ((row >> 1) & 3))) << 4); }

// 8-token fragment of column col0 + lc: tokens {row0 + 4lg + e} and {row0 + 16 + 4lg + e}.  Lane 4q+p of a 16-lane group
// hands the transposing read the address of token row q, columns col0 + 4p .. +3 (8 bytes inside chunk col0/8 + (p>>1)).
template <bool IS_A>
__device__ inline bf16x8 tr_frag(const unsigned char* img, int row0, int col0, int lg, int lc) {
  const int row = row0 + 4 * lg + (lc >> 2);
  const int ch = (col0 >> 3) + ((lc & 3) >> 1), sub = 8 * (lc & 1);
  const unsigned char* pa = img + (IS_A ? a_swz(row, ch) : d_swz(row, ch)) + sub;
  const unsigned char* pb = img + (IS_A ? a_swz(row + 16, ch) : d_swz(row + 16, ch)) + sub;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// WB_M = tokens per K-step: 32 (20 KB per ring slot, 157 VGPRs: three workgroups per CU).  Measured and dropped: 64-token
// steps at two workgroups per CU (no better than the register-staged kernel), four workgroups per CU (128-VGPR cap:
// spills in the K loop, 3x slower)
// TWO_STAGE: the workgroup's 128 x 192 partial tile goes to its token slab's slice of a workspace with PLAIN stores (free
// next to the MFMAs: the stamps of the atomic form show 15-50 % of the kernel in its fp32 atomic tail -- 75 MB of adds per
// launch at the chip's 1.3 TB/s atomic rate) and wgrad_reduce_kernel sums the slices into dW.
template <int WB_M, int MIN_WGS, bool TWO_STAGE>
__global__ __launch_bounds__(256, MIN_WGS) void wgrad_bf16_dma_kernel(
    const u16* __restrict__ dC, int lddc, const u16* __restrict__ A, int lda, float* __restrict__ dW,
    float* __restrict__ db, int M, int N, int K, int n_tiles, int k_tiles, int rows_per_split) {
  constexpr int D_BYTES = WB_M * D_ROW, STAGE = WB_M * (D_ROW + A_ROW);
  constexpr int ND = D_BYTES / 4096, NA = WB_M * A_ROW / 4096;       // DMA instructions per wave and K-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware order (blocks b, b+8, b+16.. share an XCD and its L2): the output tiles of ONE token slab run
  // back to back on one XCD, so each dC / A slab is fetched from HBM once and re-read from that L2.
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int tile = local % (n_tiles * k_tiles), split = (local / (n_tiles * k_tiles)) * 8 + xcd;
  const int n_tile = tile / k_tiles, k_tile = tile - n_tile * k_tiles;
  const int n0 = n_tile * WB_N, k0 = k_tile * BKC;
  const int m_begin = split * rows_per_split;          // a multiple of 64
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int lc = lane & 15, lg = lane >> 4;

  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(dC), 0, (int)(((size_t)(M - 1) * lddc + N) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(u16)), 0x00020000);

  // DMA instruction 4i + wave of a slab fills its LDS bytes [1024 (4i+wave), +1024): lane l fills physical 16-B piece
  // f = 64 (4i+wave) + l = (row, physical chunk) and fetches the logical chunk that the swizzle puts there.
  unsigned d_off[ND], a_off[NA];
  bool d_ok[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int f = (4 * i + wave) * 64 + lane;
    const int row = f >> 4, ch = (f & 15) ^ (2 * (row & 7));
    d_ok[i] = n0 + ch * 8 < N;                         // N = 192: the second 128-column tile is half empty
    d_off[i] = ((unsigned)row * (unsigned)lddc + (unsigned)(n0 + ch * 8)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = (4 * i + wave) * 64 + lane;
    const int row = f / 24, ch = (f - row * 24) ^ (2 * ((row >> 1) & 3));
    a_off[i] = ((unsigned)row * (unsigned)lda + (unsigned)(k0 + ch * 8)) * 2u;
  }
  const unsigned d_step = (unsigned)WB_M * (unsigned)lddc * 2u, a_step = (unsigned)WB_M * (unsigned)lda * 2u;
  unsigned d_m = (unsigned)m_begin * (unsigned)lddc * 2u, a_m = (unsigned)m_begin * (unsigned)lda * 2u;
  auto issue = [&](int st) {
    unsigned char* base = smem + (st & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      auto dst = (__attribute__((address_space(3))) void*)(base + (4 * i + wave) * 1024);
      // the token offset is part of the range-checked VGPR offset: rows >= M (and columns >= N) read as zeros
      const unsigned off = d_ok[i] ? d_off[i] + d_m : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, dst, 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      auto dst = (__attribute__((address_space(3))) void*)(base + D_BYTES + (4 * i + wave) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)(a_off[i] + a_m), 0, 0, 0);
    }
    d_m += d_step;
    a_m += a_step;
  };

  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient: dC^T @ ones as one more MFMA column (waves with wk == 0 of the k-tile-0 workgroups)
  const bool want_db = db != nullptr && k_tile == 0 && wk == 0;
  f32x4 dbacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dbacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const short one = (short)0x3F80;                     // bf16 1.0
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  const int steps = (m_end - m_begin + WB_M - 1) / WB_M;
  issue(0);
  for (int st = 0; st < steps; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of slab st has landed
    __builtin_amdgcn_s_barrier();                          // ... and everybody's; slot (st+1)&1 is free
    asm volatile("" ::: "memory");
    if (st + 1 < steps) issue(st + 1);
    const unsigned char* Ds = smem + (st & 1) * STAGE;
    const unsigned char* As = Ds + D_BYTES;
#pragma unroll
    for (int kk = 0; kk < WB_M / 32; ++kk) {
      bf16x8 fd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fd[i] = tr_frag<false>(Ds, kk * 32, wn * 64 + i * 16, lg, lc);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const bf16x8 fa = tr_frag<true>(As, kk * 32, wk * 96 + j * 16, lg, lc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], fa, acc[i][j], 0, 0, 0);     // D[n][k]
      }
      if (want_db) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dbacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], ones, dbacc[i], 0, 0, 0);
      }
    }
  }

  // lane (lg, lc) of tile (i, j): dW[n = n0 + wn*64 + 16i + 4lg + r][k = k0 + wk*96 + 16j + lc]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int kc = k0 + wk * 96 + j * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + lg * 4 + r;
        if (n < N) {
          if (TWO_STAGE) dW[((size_t)split * N + n) * K + kc] = acc[i][j][r];      // dW = the workspace here
          else atomicAdd(&dW[(size_t)n * K + kc], acc[i][j][r]);
        }
      }
    }
  if (want_db && lc == 0) {                             // every column of dC^T @ ones holds the column sum
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + lg * 4 + r;
        if (n < N) atomicAdd(&db[n], dbacc[i][r]);
      }
  }
}

// dW[e] += sum over the token slabs s in this block's chunk of ws[s][e]: 16 independent 16-B loads per thread, then one
// round of atomics per chunk (n_valid / 16 adders per element instead of n_valid)
constexpr int RED_CHUNK = 32;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const f32x4* __restrict__ ws, float* __restrict__ dW, int nk4,
                                                           int n_valid) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nk4) return;
  const int s0 = blockIdx.y * RED_CHUNK, s1 = min(n_valid, s0 + RED_CHUNK);
  const f32x4* p = ws + (size_t)s0 * nk4 + e;
  f32x4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int sidx = s0;
  for (; sidx + 8 <= s1; sidx += 8) {          // eight independent 16-B loads in flight per thread
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += __builtin_nontemporal_load(p + (size_t)u * nk4);
    p += (size_t)8 * nk4;
  }
  for (; sidx < s1; ++sidx) {
    acc[0] += __builtin_nontemporal_load(p);
    p += nk4;
  }
  const f32x4 a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  float* d = dW + (size_t)e * 4;
  atomicAdd(d, a[0]); atomicAdd(d + 1, a[1]); atomicAdd(d + 2, a[2]); atomicAdd(d + 3, a[3]);
}

}  // namespace

// dW[nk] += sum of the n_valid slab slices of the workspace (shared with the fp32 weight-gradient kernel, wgrad_f32_dma.hip)
void pangu_wgrad_reduce(hipStream_t s, const float* ws, float* dW, int nk, int n_valid) {
  const int nk4 = nk / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nk4 + 255) / 256, (n_valid + RED_CHUNK - 1) / RED_CHUNK), dim3(256), 0, s,
                     reinterpret_cast<const f32x4*>(ws), dW, nk4, n_valid);
}

namespace {

template <int WB_M, int MIN_WGS>
int launch(hipStream_t s, const u16* dC, int lddc, const u16* A, int lda, float* dW, float* db, int M, int N, int K,
           int target, float* ws, size_t ws_bytes) {
  const int n_tiles = (N + WB_N - 1) / WB_N, k_tiles = K / BKC;
  const int tiles = n_tiles * k_tiles;
  int split = (target / tiles) & ~7;                                // multiple of 8: equal share per XCD
  if (split < 8) split = 8;
  int rows = ((M + split - 1) / split + 63) / 64 * 64;
  if (rows < 256) rows = 256;
  split = ((M + rows - 1) / rows + 7) & ~7;                         // grid padded to whole XCD rounds (empty slabs exit)
  const int n_valid = (M + rows - 1) / rows;                        // slabs that hold tokens (the others write nothing)
  const size_t shm = 2 * (size_t)WB_M * (D_ROW + A_ROW);
  const bool two_stage = ws != nullptr && (size_t)n_valid * N * K * sizeof(float) <= ws_bytes && n_valid > 1;
  if (two_stage) {
    auto kern = wgrad_bf16_dma_kernel<WB_M, MIN_WGS, true>;
    PANGU_ENSURE_DYN_LDS(kern, shm);
    hipLaunchKernelGGL(kern, dim3(tiles * split), dim3(256), shm, s, dC, lddc, A, lda, ws, db, M, N, K, n_tiles, k_tiles, rows);
    pangu_wgrad_reduce(s, ws, dW, N * K, n_valid);
  } else {
    auto kern = wgrad_bf16_dma_kernel<WB_M, MIN_WGS, false>;
    PANGU_ENSURE_DYN_LDS(kern, shm);
    hipLaunchKernelGGL(kern, dim3(tiles * split), dim3(256), shm, s, dC, lddc, A, lda, dW, db, M, N, K, n_tiles, k_tiles, rows);
  }
  return pangu_launch_status();
}

}  // namespace

// -> PANGU_OK when launched, 1 when the shape is not covered (the caller falls back to the register-staged kernel)
int pangu_linear_wgrad_bf16_dma(hipStream_t s, const unsigned short* dC, int lddc, const unsigned short* A, int lda,
                                float* dW, float* db, int M, int N, int K, int target, float* ws, size_t ws_bytes) {
  if (K % BKC != 0) return 1;
  // the VGPR byte offset of the last slab's rows (up to M + 63, plus one row of columns) must not wrap 32 bits
  if (((size_t)M + 128) * (size_t)lddc * 2u >= 0xFFFFFFFFull || ((size_t)M + 128) * (size_t)lda * 2u >= 0xFFFFFFFFull) return 1;
  return launch<32, 3>(s, dC, lddc, A, lda, dW, db, M, N, K, target, ws, ws_bytes);
}
