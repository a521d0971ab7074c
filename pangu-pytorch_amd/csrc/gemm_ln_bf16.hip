// bf16 projection GEMM with the post-norm residual fused into its epilogue (inference path), gfx950:
//   out[M,N] = shortcut[M,N] + LayerNorm(A[M,K] @ W[N,K]^T + bias) * gamma + beta        (reference layers.py:250-251)
// for N = 192 or 384 (the model's two widths): the workgroup tile is 128 rows x the WHOLE row (wave grid 2 x WNW, wave
// tile 64 x 96, WNW = N / 96), so the row statistics never leave the workgroup.  Same operand staging, LDS swizzle and
// swapped-operand MFMA layout as gemm_bf16.hip (a lane owns 4 consecutive output columns of one row).
// Epilogue: the shortcut patch is staged in LDS with coalesced 16-B loads; each wave reduces (sum, sum of squares) of its
// 96 columns in fp32 straight from the accumulators (in-lane + 2 shuffles), the WNW partials of a row meet in a 4-KB LDS
// table behind ONE barrier, then normalise / gamma / beta / residual run in the MFMA layout and the rows leave as 16-B
// segments.  Saves the branch's HBM round trip (write y, read y, read shortcut, write out -> read shortcut, write out):
// the standalone LN-residual kernel is HBM-bound at ~5 TB/s and was 12 % of the bf16 forward.
// The statistics are taken on the fp32 accumulators, i.e. BEFORE the bf16 rounding the unfused path applies to y.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Which widths take the persistent LDS-DMA kernel below (bit 0: N = 384 and N = 192 with K >= 384, bit 1: N = 192 always); 0 = the
// register-staged kernel for everything (tools/ab_lib.sh builds the A/B libraries).
#ifndef PANGU_GEMM_LN_DMA
#define PANGU_GEMM_LN_DMA 1
#endif
#ifndef PANGU_GEMM_LN_RING
// N = 384: ring of the persistent kernel.  2 (default since the second A/B of round 6): TWO slots of 64-channel K-steps (128 KB) --
// half the barriers and the second half-step's fragment reads overlap the first's MFMAs: -6...-9 % against 3 / 4 = three / four slots
// of 32-channel K-steps (96 / 128 KB; level with each other).  profiles/r06_gemm_ln_bf16_ab.md
#define PANGU_GEMM_LN_RING 2
#endif

constexpr int LBM = 128;
constexpr int LBK = 64;
constexpr float LN_EPS = 1e-5f;

__device__ inline unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }
__device__ inline int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }   // byte offset

template <int WNW>
__global__ __launch_bounds__(128 * WNW, 2) void gemm_ln_residual_bf16_kernel(
    const u16* __restrict__ A, int lda, const u16* __restrict__ W, const float* __restrict__ bias,
    const u16* __restrict__ shortcut, const float* __restrict__ gamma, const float* __restrict__ beta, u16* __restrict__ out,
    int ldo, int M, int K) {
  constexpr int NT = 128 * WNW;
  constexpr int BN = 96 * WNW;                             // = N
  constexpr int STAGE = (LBM + BN) * 128;                  // bytes per LDS stage
  constexpr int NCH = (LBM + BN) * 8 / NT;                 // 16-B chunks staged per thread and K-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int m0 = blockIdx.x * LBM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int lc = lane & 15, lg = lane >> 4;

  const u16* src[NCH];
  int dst[NCH];
  const int kchunk = tid & 7;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int row = (tid >> 3) + (NT / 8) * i;              // 0 .. LBM+BN-1
    if (row < LBM) {
      int r = m0 + row;
      r = r < M ? r : M - 1;
      src[i] = A + (size_t)r * lda + kchunk * 8;
    } else {
      src[i] = W + (size_t)(row - LBM) * K + kchunk * 8;
    }
    dst[i] = (row < LBM ? swz(row, kchunk) : LBM * 128 + swz(row - LBM, kchunk));
  }
  const int KT = (K + LBK - 1) / LBK;
  u32x4 stg[NCH];
  auto fetch = [&](int kt) {
    const bool in = kt * LBK + kchunk * 8 < K;             // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      stg[i] = in ? *reinterpret_cast<const u32x4*>(src[i] + (size_t)kt * LBK) : u32x4{0u, 0u, 0u, 0u};
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(smem + buf * STAGE + dst[i]) = stg[i];
  };

  f32x4 acc[4][6];                                         // [m tile 16][n tile 16], lane: 4 consecutive n of row m = lc
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;
    if (more) fetch(kt + 1);
    const unsigned char* As = smem + (kt & 1) * STAGE;
    const unsigned char* Ws = As + LBM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fw[6];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + swz(wm * 64 + i * 16 + lc, kk * 4 + lg));
#pragma unroll
      for (int j = 0; j < 6; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(Ws + swz(wn * 96 + j * 16 + lc, kk * 4 + lg));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue.  lane (lg, lc) of tile (i, j) holds y[m = wm*64 + 16i + lc][n = wn*96 + 16j + 4lg + r], r = 0..3
  constexpr int EP_LD = 96 * 2 + 16;                       // bytes per patch row
  constexpr int CPR = 12;                                  // 16-B chunks per patch row
  unsigned char* ep = smem + wave * (64 * EP_LD);
  float* stats = reinterpret_cast<float*>(smem + 2 * WNW * 64 * EP_LD);      // [2 wm][WNW][64 rows][sum, sumsq]
  const int wave_n0 = wn * 96, wave_m0 = m0 + wm * 64;
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      out, 0, (int)(((size_t)(M - 1) * ldo + BN) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(shortcut), 0, (int)((size_t)M * BN * sizeof(u16)), 0x00020000);
  // shortcut patch [64][96] -> LDS (rows >= M arrive as zeros)
#pragma unroll
  for (int it = 0; it < CPR; ++it) {
    const int f = lane + 64 * it, row = f / CPR, ch = f - row * CPR;
    const unsigned off = ((unsigned)(wave_m0 + row) * (unsigned)BN + (unsigned)(wave_n0 + ch * 8)) * 2u;
    *reinterpret_cast<u32x4*>(ep + row * EP_LD + ch * 16) = __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, (int)off, 0, 0);
  }
  // y = acc + bias (kept in acc), per-row partial statistics of this wave's 96 columns
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + wave_n0 + j * 16 + lg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] += bv;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s += acc[i][j][r];
        q = fmaf(acc[i][j][r], acc[i][j][r], q);
      }
    s += __shfl_xor(s, 16, 64);
    q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    if (lg == 0) {
      float* st = stats + (((wm * WNW + wn) * 64) + i * 16 + lc) * 2;
      st[0] = s;
      st[1] = q;
    }
  }
  __syncthreads();
  f32x4 gv[6], bt[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    gv[j] = *reinterpret_cast<const f32x4*>(gamma + wave_n0 + j * 16 + lg * 4);
    bt[j] = *reinterpret_cast<const f32x4*>(beta + wave_n0 + j * 16 + lg * 4);
  }
  constexpr float INV_C = 1.0f / BN;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < WNW; ++w) {
      const float* st = stats + (((wm * WNW + w) * 64) + i * 16 + lc) * 2;
      s += st[0];
      q += st[1];
    }
    const float mean = s * INV_C;
    const float rstd = rsqrtf(fmaxf(q * INV_C - mean * mean, 0.f) + LN_EPS);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      unsigned char* slot = ep + (i * 16 + lc) * EP_LD + (j * 16 + lg * 4) * 2;
      const u32x2 xp = *reinterpret_cast<const u32x2*>(slot);
      f32x4 v = (acc[i][j] - mean) * rstd * gv[j] + bt[j];
      v[0] += __builtin_bit_cast(float, xp[0] << 16);
      v[1] += __builtin_bit_cast(float, xp[0] & 0xFFFF0000u);
      v[2] += __builtin_bit_cast(float, xp[1] << 16);
      v[3] += __builtin_bit_cast(float, xp[1] & 0xFFFF0000u);
      *reinterpret_cast<u32x2*>(slot) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
    }
    // read back rows 16i .. 16i+15 of the patch: 16 * 12 chunks of 16 B -> whole row segments
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int f = lane + 64 * it;
      const int row = i * 16 + f / CPR, ch = f % CPR;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * EP_LD + ch * 16);
      const unsigned off = ((unsigned)(wave_m0 + row) * (unsigned)ldo + (unsigned)(wave_n0 + ch * 8)) * 2u;
      __builtin_amdgcn_raw_buffer_store_b128(v, o_rsrc, (int)off, 0, 2);
    }
  }
}

// ---- persistent LDS-DMA variant (round 6) ----------------------------------------------------------------------------
// The kernel above owns one 128-row tile per workgroup and stages its operands through registers one K-step ahead: with K = C the
// whole K-loop is 3-6 steps of ~770 MFMA cycles each, i.e. far shorter than one memory latency, and -- N = 384: 128 KB of LDS, ONE
// workgroup per CU -- nothing overlaps a tile's prologue, its per-step load latencies and its epilogue: 22 us per tile, 1.84x the
// HBM floor (profiles/r05_fwd_bf16_issue_table.md: 0.47 of the wave-cycles parked, MFMA busy 0.17).  Here the same tile, wave grid
// and fragment reads run in a PERSISTENT workgroup whose operand ring never drains:
//   * A and W travel L2/HBM -> LDS by LDS-DMA (buffer_load .. lds: no staging registers) in K-steps of 32 through a ring of RING
//     slots; the requests of the NEXT tile's first RING-1 steps are issued under the current tile's last K-steps, so its epilogue
//     (LayerNorm, residual, stores) runs with the next tile's operands already in flight;
//   * the tile's 128 x N shortcut patch (the other HBM stream) is requested into registers at the tile's first K-step -- 12 16-B
//     loads per thread in the layout the output rows leave in -- and is consumed by the epilogue one 16-row group at a time
//     through a 3.3-KB per-wave LDS patch (26 KB instead of the 106 KB the tile-at-once epilogue needs);
//   * every wait is a COUNTED s_waitcnt vmcnt(n): the in-order queue holds, per thread, LPS DMA requests per step, 12 shortcut
//     loads and 12 output stores per tile at known positions.
// LDS: RING x (128 + N) x 64 B ring + 2 WNW x 3328 B patches + 4 KB statistics (N = 384, RING = 3: 127 KB, one 8-wave workgroup
// per CU; N = 192, RING = 3: 75 KB, two 4-wave workgroups per CU).
__device__ inline int kswz64(int row, int chunk) {      // 64-byte rows, 4 chunks: F = {0,2,3,1}[(row>>2)&3]
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((chunk ^ f) << 4);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DBK = channels per K-step: 32 (64-byte ring rows, swizzle kswz64) or 64 (128-byte rows, swizzle swz: half the barriers, the second
// half-step's fragment reads can overlap the first's MFMAs; ring of 2 only -- 2 x 64 KB at N = 384).
// WM = wave rows of 64 tokens: the tile is 64 WM x N (2: 128 rows, the N = 384 form; 4: 256 rows x 192 columns -- one 8-wave workgroup
// per CU at N = 192 as well, the W panel re-streamed half as often).
template <int WNW, int RING, int DBK = 32, int WM = 2>
__global__ __launch_bounds__(64 * WM * WNW, 2) void gemm_ln_residual_bf16_dma_kernel(
    const u16* __restrict__ A, int lda, const u16* __restrict__ W, const float* __restrict__ bias,
    const u16* __restrict__ shortcut, const float* __restrict__ gamma, const float* __restrict__ beta, u16* __restrict__ out,
    int ldo, int M, int K, int m_tiles) {
  static_assert((DBK == 32 && (RING == 3 || RING == 4)) || (DBK == 64 && RING == 2), "ring of three or four K-steps of 32, or two of 64");
  constexpr int ROWB = DBK * 2;                            // bytes per ring row
  constexpr int RPI = 1024 / ROWB;                         // ring rows per LDS-DMA instruction (1 KB)
  constexpr int CH = DBK / 8;                              // 16-B chunks per ring row
  constexpr int NW = WM * WNW;                             // waves
  constexpr int TBM = 64 * WM;                             // token rows per tile
  constexpr int BN = 96 * WNW;                             // = N
  constexpr int ROWS = TBM + BN;
  constexpr int STAGE = ROWS * ROWB;                       // bytes per ring slot
  constexpr int LPS = ROWS / (RPI * NW);                   // LDS-DMA instructions per wave and K-step (RPI rows each)
  static_assert(ROWS % (RPI * NW) == 0, "whole DMA instructions per wave");
  constexpr int EP_LD = 96 * 2 + 16;                       // bytes per patch row
  constexpr int CPR = 12;                                  // 16-B chunks per patch row
  constexpr int NS = 12;                                   // shortcut loads = output stores per thread and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ep = smem + RING * STAGE + (threadIdx.x >> 6) * (16 * EP_LD);
  // the statistics table [WM][WNW][64 rows][sum, sumsq] (4 KB) ALIASES the patches: it is dead (every wave has taken
  // its rows' mean / rstd into registers, second barrier) before the first patch is written -- what lets a ring of four fit
  float* stats = reinterpret_cast<float*>(smem + RING * STAGE);
  static_assert(WM * WNW * 64 * 2 * 4 <= NW * 16 * (96 * 2 + 16), "statistics fit in the patch area");
  float* prm = reinterpret_cast<float*>(smem + RING * STAGE + NW * 16 * EP_LD);      // [bias | gamma | beta][N]: resident

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int lc = lane & 15, lg = lane >> 4;
  const int wave_n0 = wn * 96;

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(W), 0, (int)((size_t)BN * K * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      out, 0, (int)(((size_t)(M - 1) * ldo + BN) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(shortcut), 0, (int)((size_t)M * BN * sizeof(u16)), 0x00020000);

  // DMA instruction q = i * NW + wave fills ring rows RPI q .. RPI q + RPI - 1; this lane fills (row RPI q + lane / CH, physical
  // chunk lane % CH) with the LOGICAL chunk (lane % CH) ^ F(row) (source-side swizzle).  Rows < 128: A (tile-relative; the tile's byte offset is added
  // to the VECTOR offset at issue time -- the descriptor's range check sees vector + immediate offsets only, and rows >= M must
  // arrive as zeros), rows >= 128: W.
  unsigned voff[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = RPI * (i * NW + wave) + lane / CH;
    const int f = DBK == 32 ? ((0x78 >> (((row >> 2) & 3) * 2)) & 3) : ((row >> 1) & 7);
    const int c = (lane % CH) ^ f;
    voff[i] = row < TBM ? ((unsigned)row * (unsigned)lda + c * 8) * 2u : ((unsigned)(row - TBM) * (unsigned)K + c * 8) * 2u;
  }
  const int KT = K / DBK;
  // global step g = it * KT + kt of this workgroup's it-th tile
  auto issue = [&](int g, int tile, int kt) {
    unsigned char* base = smem + (g % RING) * STAGE;
    const unsigned a_base = (unsigned)tile * (unsigned)(TBM * 2) * (unsigned)lda;      // < 2^32: checked by the launcher
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int q = i * NW + wave;
      auto dst = (__attribute__((address_space(3))) void*)(base + q * 1024);
      if (RPI * q < TBM) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)(voff[i] + a_base), kt * DBK * 2, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, (int)voff[i], kt * DBK * 2, 0, 0);
    }
  };

  for (int c = tid; c < BN; c += 64 * NW) {
    prm[c] = bias ? bias[c] : 0.f;
    prm[BN + c] = gamma[c];
    prm[2 * BN + c] = beta[c];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the counted waits below see DMA / shortcut / store traffic only
  __syncthreads();

  int tile = blockIdx.x;
  if (tile >= m_tiles) return;
#pragma unroll
  for (int p0 = 0; p0 < RING - 1; ++p0) issue(p0, tile, p0);          // KT >= RING - 1: checked by the launcher
  bool first = true;
  int g = 0;
  for (; tile < m_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool has_next = next < m_tiles;
    const int m0 = tile * TBM, wave_m0 = m0 + wm * 64;
    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(prm + wave_n0 + j * 16 + lg * 4);
    u32x4 sc[NS];
    for (int kt = 0; kt < KT; ++kt, ++g) {
      // In-order queue of this thread (oldest first) within a tile: DMA(0 .. RING-2) were issued under the PREVIOUS tile's last
      // K-steps, then that tile's NS output stores, then at step k: DMA(k + RING - 1), and at step 0 also the NS shortcut loads.
      // Step k needs DMA(k); NEWER than it (allowed to stay in flight) are
      //   the DMA groups k+1 .. k+RING-2 that exist, the stores iff k <= RING-2 (not in this workgroup's first tile), and the
      //   shortcut loads iff 1 <= k <= RING-1.
      {
        int groups = RING - 2;
        if (!has_next && KT - 1 - kt < groups) groups = KT - 1 - kt;
        const int extra = ((kt <= RING - 2 && !first) ? 1 : 0) + ((kt >= 1 && kt <= RING - 1) ? 1 : 0);
        const int code = groups * 3 + extra;                 // (groups 0..2) x (extra 0..2)
        if (code == 0) wait_vmcnt<0>();
        else if (code == 1) wait_vmcnt<NS>();
        else if (code == 2) wait_vmcnt<2 * NS>();
        else if (code == 3) wait_vmcnt<LPS>();
        else if (code == 4) wait_vmcnt<LPS + NS>();
        else if (code == 5) wait_vmcnt<LPS + 2 * NS>();
        else if (code == 6) wait_vmcnt<2 * LPS>();
        else if (code == 7) wait_vmcnt<2 * LPS + NS>();
        else wait_vmcnt<2 * LPS + 2 * NS>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's fragment reads of step g-1 have returned
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                            // step g landed for every wave; slot (g-1) % RING is free
      asm volatile("" ::: "memory");
      if (kt + RING - 1 < KT) issue(g + RING - 1, tile, kt + RING - 1);
      else if (has_next) issue(g + RING - 1, next, kt + RING - 1 - KT);
      if (kt == 0) {
        // shortcut patch [64 rows][96 columns] of this wave, 16-B chunks in the order the output rows leave in (rows >= M: zeros)
#pragma unroll
        for (int it = 0; it < NS; ++it) {
          const int f = lane + 64 * it, row = f / CPR, ch = f - row * CPR;
          const unsigned off = ((unsigned)(wave_m0 + row) * (unsigned)BN + (unsigned)(wave_n0 + ch * 8)) * 2u;
          sc[it] = __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, (int)off, 0, 0);
        }
      }
      const unsigned char* As = smem + (g % RING) * STAGE;
      const unsigned char* Ws = As + TBM * ROWB;
#pragma unroll
      for (int kk = 0; kk < DBK / 32; ++kk) {
        bf16x8 fa[4], fw[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wm * 64 + i * 16 + lc;
          fa[i] = *reinterpret_cast<const bf16x8*>(As + (DBK == 32 ? kswz64(row, lg) : swz(row, kk * 4 + lg)));
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int row = wn * 96 + j * 16 + lc;
          fw[j] = *reinterpret_cast<const bf16x8*>(Ws + (DBK == 32 ? kswz64(row, lg) : swz(row, kk * 4 + lg)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);   // D[n][m]
      }
    }
    first = false;

    // ---- epilogue.  lane (lg, lc) of tile (i, j) holds y[m = wm*64 + 16i + lc][n = wn*96 + 16j + 4lg + r] (bias included)
    // (the compiler guards each group's first use of the shortcut registers with vmcnt(9..11): in the in-order queue that is a
    // wait for the OLDEST of the next tile's operand requests from the third group on -- requests issued two K-steps and half an
    // epilogue earlier; forcing the registers "landed" inside the K-loop costs a full vmcnt(0) there instead, which is worse)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s += acc[i][j][r];
          q = fmaf(acc[i][j][r], acc[i][j][r], q);
        }
      s += __shfl_xor(s, 16, 64);
      q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lg == 0) {
        float* st = stats + (((wm * WNW + wn) * 64) + i * 16 + lc) * 2;
        st[0] = s;
        st[1] = q;
      }
    }
    // (a raw barrier: __syncthreads() would also wait for the next tile's operand requests -- vmcnt(0); the statistics went
    // through LDS only, and the next tile's statistics are >= KT barriers away)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr float INV_C = 1.0f / BN;
    float mean_[4], rstd_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WNW; ++w) {
        const float* st = stats + (((wm * WNW + w) * 64) + i * 16 + lc) * 2;
        s += st[0];
        q += st[1];
      }
      mean_[i] = s * INV_C;
      rstd_[i] = rsqrtf(fmaxf(q * INV_C - mean_[i] * mean_[i], 0.f) + LN_EPS);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // the table is dead: the patches may overwrite it
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mean = mean_[i], rstd = rstd_[i];
      // the group's shortcut rows -> patch (the previous group's read-back must have returned: same wave, LDS in order)
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int f = lane + 64 * it, row = f / CPR, ch = f - row * CPR;
        *reinterpret_cast<u32x4*>(ep + row * EP_LD + ch * 16) = sc[3 * i + it];
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        unsigned char* slot = ep + lc * EP_LD + (j * 16 + lg * 4) * 2;
        const u32x2 xp = *reinterpret_cast<const u32x2*>(slot);
        const f32x4 gj = *reinterpret_cast<const f32x4*>(prm + BN + wave_n0 + j * 16 + lg * 4);
        const f32x4 bj = *reinterpret_cast<const f32x4*>(prm + 2 * BN + wave_n0 + j * 16 + lg * 4);
        f32x4 v = (acc[i][j] - mean) * rstd * gj + bj;
        v[0] += __builtin_bit_cast(float, xp[0] << 16);
        v[1] += __builtin_bit_cast(float, xp[0] & 0xFFFF0000u);
        v[2] += __builtin_bit_cast(float, xp[1] << 16);
        v[3] += __builtin_bit_cast(float, xp[1] & 0xFFFF0000u);
        *reinterpret_cast<u32x2*>(slot) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
      }
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int f = lane + 64 * it, row = f / CPR, ch = f - row * CPR;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * EP_LD + ch * 16);
        const unsigned off = ((unsigned)(wave_m0 + i * 16 + row) * (unsigned)ldo + (unsigned)(wave_n0 + ch * 8)) * 2u;
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rsrc, (int)off, 0, 2);
      }
    }
  }
}

template <int WNW, int RING, int DBK = 32, int WM = 2>
int launch_ln_dma(hipStream_t s, const u16* A, int lda, const u16* W, const float* bias, const u16* shortcut, const float* gamma,
                  const float* beta, u16* out, int ldo, int M, int K) {
  constexpr int BN = 96 * WNW, TBM = 64 * WM, NW = WM * WNW;
  const int m_tiles = (M + TBM - 1) / TBM;
  const size_t shm = (size_t)RING * (TBM + BN) * (DBK * 2) + (size_t)NW * 16 * (96 * 2 + 16) + (size_t)3 * BN * sizeof(float);
  auto kern = gemm_ln_residual_bf16_dma_kernel<WNW, RING, DBK, WM>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  const int per_cu = (int)(160 * 1024 / shm) < 1 ? 1 : (int)(160 * 1024 / shm);
  int grid = 256 * (per_cu > 2 ? 2 : per_cu);
  if (grid > m_tiles) grid = m_tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), shm, s, A, lda, W, bias, shortcut, gamma, beta, out, ldo, M, K, m_tiles);
  return pangu_launch_status();
}

template <int WNW>
int launch_ln(hipStream_t s, const u16* A, int lda, const u16* W, const float* bias, const u16* shortcut, const float* gamma,
              const float* beta, u16* out, int ldo, int M, int K) {
  constexpr int BN = 96 * WNW;
  const size_t stages = 2 * (size_t)(LBM + BN) * 128;
  const size_t epi = (size_t)2 * WNW * 64 * (96 * 2 + 16) + (size_t)2 * WNW * 64 * 2 * sizeof(float);
  const size_t shm = stages > epi ? stages : epi;
  auto kern = gemm_ln_residual_bf16_kernel<WNW>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  hipLaunchKernelGGL(kern, dim3((M + LBM - 1) / LBM), dim3(128 * WNW), shm, s, A, lda, W, bias, shortcut, gamma, beta, out, ldo,
                     M, K);
  return pangu_launch_status();
}

}  // namespace

extern "C" int pangu_linear_ln_residual_fwd_bf16(pangu_stream_t stream, const void* A, int lda, const void* W,
                                                 const float* bias, const void* shortcut, const float* gamma,
                                                 const float* beta, void* out, int ldo, int M, int N, int K) {
  if (!A || !W || !shortcut || !gamma || !beta || !out) return PANGU_E_NULL;
  if (M <= 0 || K <= 0 || (K & 7) || lda < K || (lda & 7) || ldo < N || (ldo & 7)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, lda, 2) || !pangu_fits_u32(M, ldo, 2)) return PANGU_E_RANGE;
  if (N != 192 && N != 384) return PANGU_E_SHAPE;          // the tile must span the whole row
  hipStream_t s = (hipStream_t)stream;
  // the persistent LDS-DMA kernel: K a multiple of 32 and at least three K-steps, dense A rows addressable through the scalar offset
  // N = 384: always (bit 0).  N = 192: the 256-row-tile form where the K-loop is long enough to pay (K >= 384: -7 % at K = 768; at
  // K = 192 -- the forward's launch -- it measured level with the register-staged kernel, which then stays); bit 1 forces it.
  const bool dma = PANGU_GEMM_LN_DMA && K % 32 == 0 && K >= 96 && pangu_fits_u32(M, lda, 2) &&
                   (N == 384 ? (PANGU_GEMM_LN_DMA & 1) : ((PANGU_GEMM_LN_DMA & 2) || K >= 384));
  if (N == 192) {
    if (dma && K % 64 == 0 && K >= 192)      // 256 x 192 tiles, one 8-wave workgroup per CU, K-steps of 64, ring of two
      return launch_ln_dma<2, 2, 64, 4>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
    if (dma) return launch_ln_dma<2, 3>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
    return launch_ln<2>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
  }
  if (dma) {
    // ring of four (three K-steps in flight: 96 KB per CU) where the K-loop is long enough for its wait pattern, three otherwise
    if (PANGU_GEMM_LN_RING == 2 && K % 64 == 0 && K >= 192)      // K-steps of 64, ring of two (A/B build)
      return launch_ln_dma<4, 2, 64>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
    if (PANGU_GEMM_LN_RING == 4 && K >= 160)
      return launch_ln_dma<4, 4>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
    return launch_ln_dma<4, 3>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
  }
  return launch_ln<4>(s, (const u16*)A, lda, (const u16*)W, bias, (const u16*)shortcut, gamma, beta, (u16*)out, ldo, M, K);
}
