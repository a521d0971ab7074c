// fp32 weight-gradient GEMM for gfx950, LDS-DMA variant:  dW[N,K] += dC[M,N]^T @ A[M,K],  db[N] += colsum(dC).
//
// Same decomposition as wgrad_f32.hip (one 64*TNN x 64*TK output tile and one token slab per 256-thread workgroup,
// v_mfma_f32_32x32x2_f32, fp32 no-return atomics into dW, XCD-aware slab order), built for OCCUPANCY like
// gemm_f32_dma.hip: the 16-token slabs of dC and A travel L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, no
// staging registers, no ds_write pass).  The token-major slab image is exactly what the DMA writes (lane-linear 16-B
// pieces), unpadded: ds_read_b32 fragments are served per 32-lane half, each half one token row, consecutive lanes on
// consecutive columns -- conflict-free at any row length.  <2,3>: 40 KB of LDS and <= 128 VGPRs: FOUR workgroups per
// CU (the register-staged kernel: three); <3,3>: 48 KB, three (two).
// The bias gradient (column sums of dC) is read back from the LDS slab by the k-tile-0 workgroups.
// Only for N % (64*TNN) == 0 and K % (64*TK) == 0 (no column range checks); rows >= M are out of the buffer range and
// read as zeros.  Everything else stays on wgrad_f32.hip.
#include "common.h"

void pangu_wgrad_reduce(hipStream_t s, const float* ws, float* dW, int nk, int n_valid);      // wgrad_bf16_dma.hip

namespace {

constexpr int WG_BM = 16;    // tokens per K-step

// TWO_STAGE: the partial tile goes to its token slab's slice of a workspace with plain stores and pangu_wgrad_reduce sums the
// slices into dW (see wgrad_bf16_dma.hip: plain stores are free next to the MFMAs, the atomic tail is not)
template <int TNN, int TK, bool TWO_STAGE>
__global__ __launch_bounds__(256, (TNN * TK <= 6) ? 4 : 3) void wgrad_f32_dma_kernel(
    const float* __restrict__ dC, int lddc, const float* __restrict__ A, int lda, float* __restrict__ dW,
    float* __restrict__ db, int M, int N, int K, int n_tiles, int k_tiles, int rows_per_split) {
  constexpr int WG_BN = 64 * TNN;                    // output rows (n) per tile
  constexpr int BKC = 64 * TK;                       // output columns (k) per tile
  constexpr int D_F4 = WG_BN / 4, A_F4 = BKC / 4;    // float4 per slab row
  constexpr int D_BYTES = WG_BM * WG_BN * 4, STAGE = WG_BM * (WG_BN + BKC) * 4;
  constexpr int ND = TNN, NA = TK;                   // DMA instructions per wave and K-step: dC 4*TNN chunks, A 4*TK
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  // XCD-aware order (blocks b, b+8, b+16.. share an XCD and its L2): the output tiles of ONE token slab run
  // back to back on one XCD, so each dC / A slab is fetched from HBM once and re-read from that L2.
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int tile = local % (n_tiles * k_tiles), split = (local / (n_tiles * k_tiles)) * 8 + xcd;
  const int n_tile = tile / k_tiles, k_tile = tile - n_tile * k_tiles;
  const int n0 = n_tile * WG_BN, k0 = k_tile * BKC;
  const int m_begin = split * rows_per_split;        // a multiple of WG_BM
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dC), 0, (int)(((size_t)(M - 1) * lddc + N) * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(float)), 0x00020000);

  // DMA chunk q of a slab = its 16-B pieces 64q .. 64q+63 in row-major order (1 KB of LDS); wave w issues chunks
  // 4i + w.  Byte offsets relative to the slab's first token; the token offset of the K-step is added per step (it must
  // be part of the range-checked VGPR offset: rows >= M have to read as zeros).
  unsigned d_off[ND], a_off[NA];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int f = (4 * i + wave) * 64 + lane;
    d_off[i] = ((unsigned)(f / D_F4) * (unsigned)lddc + (unsigned)(n0 + (f % D_F4) * 4)) * 4u;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = (4 * i + wave) * 64 + lane;
    a_off[i] = ((unsigned)(f / A_F4) * (unsigned)lda + (unsigned)(k0 + (f % A_F4) * 4)) * 4u;
  }
  const unsigned d_step = (unsigned)WG_BM * (unsigned)lddc * 4u, a_step = (unsigned)WG_BM * (unsigned)lda * 4u;
  unsigned d_m = (unsigned)m_begin * (unsigned)lddc * 4u, a_m = (unsigned)m_begin * (unsigned)lda * 4u;
  auto issue = [&](int st) {
    unsigned char* base = smem + (st & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      auto dst = (__attribute__((address_space(3))) void*)(base + (4 * i + wave) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, dst, 16, (int)(d_off[i] + d_m), 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      auto dst = (__attribute__((address_space(3))) void*)(base + D_BYTES + (4 * i + wave) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)(a_off[i] + a_m), 0, 0, 0);
    }
    d_m += d_step;
    a_m += a_step;
  };

  f32x16 acc[TNN][TK];
#pragma unroll
  for (int i = 0; i < TNN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient (k-tile-0 workgroups): column sums of the dC slab read back from LDS.  128-column tiles: thread t sums
  // column t % 128 over token rows 8*(t / 128) .. +7; 192-column tiles: threads 0..191 sum all 16 rows.
  const bool want_db = db != nullptr && k_tile == 0;
  constexpr int DB_ROWS = (WG_BN == 128) ? 8 : 16;
  const int db_col = (WG_BN == 128) ? (tid & 127) : tid;
  const int db_row0 = (WG_BN == 128) ? 8 * (tid >> 7) : 0;
  float dbacc = 0.f;

  const int steps = (m_end - m_begin + WG_BM - 1) / WG_BM;
  issue(0);
  for (int st = 0; st < steps; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of slab st has landed
    __builtin_amdgcn_s_barrier();                          // ... and everybody's; slot (st+1)&1 is free
    asm volatile("" ::: "memory");
    if (st + 1 < steps) issue(st + 1);
    const float* Ds = reinterpret_cast<const float*>(smem + (st & 1) * STAGE);
    const float* As = Ds + WG_BM * WG_BN;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float fd[TNN], fa[TK];
#pragma unroll
      for (int i = 0; i < TNN; ++i) fd[i] = Ds[(2 * s + lh) * WG_BN + wn * 32 * TNN + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < TK; ++j) fa[j] = As[(2 * s + lh) * BKC + wk * 32 * TK + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < TNN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[i], fa[j], acc[i][j], 0, 0, 0);
    }
    if (want_db) {
      if (WG_BN == 128 || tid < WG_BN) {
#pragma unroll
        for (int r = 0; r < DB_ROWS; ++r) dbacc += Ds[(db_row0 + r) * WG_BN + db_col];
      }
    }
  }

  // dW tile: C/D layout col = lane&31 (k), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (n).  ONE running row pointer per 32 x 32 tile,
  // made opaque per tile: left to itself the compiler forms all 144 element addresses (64-bit) up front, next to the 144 accumulator
  // registers -- 65-66 registers through scratch in the <3,3> instantiation at its 168-VGPR cap (rounds 2-3)
  float* tile0 = dW + ((size_t)(TWO_STAGE ? (size_t)split * N : 0) + n0 + wn * 32 * TNN + 4 * lh) * K + k0 + wk * 32 * TK + lr;
#pragma unroll
  for (int j = 0; j < TK; ++j) {
#pragma unroll
    for (int i = 0; i < TNN; ++i) {
      float* p = tile0 + (size_t)(i * 32) * K + j * 32;
      asm volatile("" : "+v"(p));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (TWO_STAGE) *p = acc[i][j][4 * q + e];                            // dW = the workspace here
          else atomicAdd(p, acc[i][j][4 * q + e]);
          p += (e < 3) ? K : 5 * K;                                            // rows 8q + e -> 8q + e + 1 ... -> 8(q + 1)
        }
      }
    }
  }
  if (want_db && (WG_BN == 128 || tid < WG_BN)) atomicAdd(&db[n0 + db_col], dbacc);
}

template <int TNN, int TK>
int launch_wgrad_dma(hipStream_t s, const float* dC, int lddc, const float* A, int lda, float* dW, float* db, int M, int N,
                     int K, int target, float* ws, size_t ws_bytes) {
  constexpr int WG_BN = 64 * TNN, BKC = 64 * TK;
  const int n_tiles = N / WG_BN, k_tiles = K / BKC;
  const int tiles = n_tiles * k_tiles;
  int split = ((target + tiles - 1) / tiles + 7) & ~7;              // equal share per XCD
  int rows = ((M + split - 1) / split + WG_BM - 1) / WG_BM * WG_BM;
  if (rows < 8 * WG_BM) rows = 8 * WG_BM;
  split = ((M + rows - 1) / rows + 7) & ~7;                         // grid padded to whole XCD rounds (empty slabs exit)
  const int n_valid = (M + rows - 1) / rows;                        // slabs that hold tokens (the others write nothing)
  if (ws != nullptr && n_valid > 1 && (size_t)n_valid * N * K * sizeof(float) <= ws_bytes) {
    hipLaunchKernelGGL((wgrad_f32_dma_kernel<TNN, TK, true>), dim3(tiles * split), dim3(256), 0, s, dC, lddc, A, lda, ws, db, M,
                       N, K, n_tiles, k_tiles, rows);
    pangu_wgrad_reduce(s, ws, dW, N * K, n_valid);
  } else {
    hipLaunchKernelGGL((wgrad_f32_dma_kernel<TNN, TK, false>), dim3(tiles * split), dim3(256), 0, s, dC, lddc, A, lda, dW, db, M,
                       N, K, n_tiles, k_tiles, rows);
  }
  return pangu_launch_status();
}

}  // namespace

// -> PANGU_OK when launched, PANGU_WGRAD_NOT_COVERED (a value no hipError_t or PANGU_E_* takes) when the shape is not
// covered: the caller falls back to the register-staged kernel; every other non-zero code is a real launch error
constexpr int PANGU_WGRAD_NOT_COVERED = -1000;
int pangu_linear_wgrad_f32_dma(hipStream_t s, const float* dC, int lddc, const float* A, int lda, float* dW, float* db,
                               int M, int N, int K, int tnn, int target, float* ws, size_t ws_bytes) {
  // the VGPR byte offset of the last slab's rows (up to M + 15, plus one row of columns) must not wrap 32 bits
  if (((size_t)M + 32) * (size_t)lddc * 4u >= 0xFFFFFFFFull || ((size_t)M + 32) * (size_t)lda * 4u >= 0xFFFFFFFFull) return PANGU_WGRAD_NOT_COVERED;
  if (K % 192 != 0) return PANGU_WGRAD_NOT_COVERED;
  if (tnn == 3 && N % 192 == 0) return launch_wgrad_dma<3, 3>(s, dC, lddc, A, lda, dW, db, M, N, K, target, ws, ws_bytes);
  if (tnn == 2 && N % 128 == 0) return launch_wgrad_dma<2, 3>(s, dC, lddc, A, lda, dW, db, M, N, K, target, ws, ws_bytes);
  return PANGU_WGRAD_NOT_COVERED;
}
