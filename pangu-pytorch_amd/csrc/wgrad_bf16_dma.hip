// bf16 weight-gradient GEMM for gfx950, LDS-DMA variant:  dW[N,K] (fp32) += dC[M,N]^T @ A[M,K],  db[N] += colsum(dC).
//
// Same decomposition as wgrad_bf16.hip (an output tile and one token slab per workgroup, v_mfma_f32_16x16x32_bf16 fed by
// transposing LDS reads of token-major slabs, fp32 partial tiles into a workspace or no-return atomics into dW, XCD-aware
// slab order), with the slabs travelling L2 -> LDS by LDS-DMA: no staging registers, no ds_write pass.
// Every wave owns 64 (n) x 96 (k) of the output; the workgroup is NWN x NWK waves:
//   2 x 2  128 x 192 tile, 4 waves, three workgroups per CU, two ring slots each      (any N, K % 192 == 0)
//   6 x 2  384 x 192 tile, 12 waves, ONE workgroup per CU with a 4-slot ring          (N % 384 == 0)
//   3 x 4  192 x 384 tile, 12 waves, likewise                                          (N % 192 == 0, K % 384 == 0)
// The DMA writes lane-linear 16-B pieces, so rows cannot be padded; instead the 16-B chunks are XOR-swizzled on the
// SOURCE side (a lane fetches the logical chunk that belongs at its physical position):
//   slab rows that are a multiple of 256 B (one bank period): physical chunk = logical ^ 2*(row & 7)   -> the 8 token rows
//     a 32-lane half reads (two 4-row blocks, 32 B per row) land on 8 different 32-B bank groups;
//   slab rows of 384 B (odd rows start half a period later): physical chunk = logical ^ 2*((row >> 1) & 3).
// Both make every ds_read_b64_tr_b16 of the fragment reads conflict-free in its 32-lane half.
// The bias gradient is one more MFMA column: dC^T @ ones (k-tile-0 workgroups), free next to the 24 tile MFMAs.
// Only K % 192 == 0 (every block linear); other shapes stay on wgrad_bf16.hip.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#ifdef PANGU_WGRAD_STAMP
// Diagnostic build only (tools/ablate_wgrad.py): per-wave s_memtime sums over the K-steps: [0] own-DMA wait, [1] barrier,
// [2] DMA issue, [3] first fragments (four dC + one A), [4] the 24 MFMAs with their A-fragment reads, [5] whole kernel, [6] steps.
constexpr int STAMP_WAVES = 16384;
__device__ unsigned long long g_wgrad_stamp[STAMP_WAVES * 8];
__device__ __forceinline__ unsigned long long wg_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define WG_STAMP(v) const unsigned long long v = wg_stamp()
#define WG_ACC(k, a, b) st_sum[k] += (b) - (a)
#else
#define WG_STAMP(v)
#define WG_ACC(k, a, b)
#endif

// Why the 12-wave tiles (round 4): 0.6x the staged bytes and LDS-DMA pieces per FLOP (36 KB and 36 pieces per 384 x 192 x 32
// MACs against 20 KB and 20 pieces per 128 x 192 x 32), and a 4-slot ring keeps three slabs (108 KB) in flight where three 4-wave
// workgroups hold one 20-KB slab each: the waves' wait for their own pieces drops from 360-560 to ~100 cycles per step.
// In-kernel stamps (tools/ablate_wgrad.py, -DPANGU_WGRAD_STAMP), 12-wave step of ~2,150 cycles per wave: barrier 550 (skew: the
// SIMD's three waves share one MFMA pipe, 3 x 384 cycles), 3 piece requests 330, first fragments 335, MFMA loop 815.  Timing-only
// ablations at M = 131,040, N = 1,536, K = 384 (0.164 ms as shipped then): no requests 0.145, and no barrier 0.141, and no
// fragment reads 0.127 -- of which 0.062 is MFMA time at peak: the partial-tile stores + the reduce launch were ~0.05 ms, hence
// the register-layout workspace below (-4..-10 us per launch).
// Measured and dropped: 192 x 192 tiles on six waves (a CU takes ONE six-wave workgroup at 154 VGPRs -- waves 4, 5 land on SIMDs
// 0, 1 -- 0.31 ms against 0.20); 64-token steps on a 2-slot ring and a 3-slot ring (level with this one); first fragments of
// step st + 1 requested at the end of step st across the barrier (level: the step is bounded by the shared MFMA pipe plus the
// barrier skew, not by the fragment latency).
template <int ROWB>
__device__ inline int swz_mask(int row) { return ROWB % 256 == 0 ? 2 * (row & 7) : 2 * ((row >> 1) & 3); }
template <int ROWB>
__device__ inline int swz(int row, int chunk) { return row * ROWB + ((chunk ^ swz_mask<ROWB>(row)) << 4); }

// Fragment reads are inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 BUILTIN that follows an
// LDS-DMA request (it cannot tell the ring slots apart) -- the slab just requested would have to land before this step's
// fragments are read, i.e. no copy / MFMA overlap inside a workgroup at all (round 4: that wait was the kernel's "parked 0.66").
// Asm reads carry no automatic waits, so every use sits behind lds_wait(), which ties the registers to an s_waitcnt lgkmcnt(0).
template <int OFF>
__device__ inline s16x4 lds_tr(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ inline void lds_wait(s16x4& a, s16x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ inline bf16x8 cat(const s16x4 a, const s16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// 8-token fragment of column col0 + lc: tokens {4lg + e} and {16 + 4lg + e} of a 32-token step.  Lane 4q+p of a 16-lane group
// hands the transposing read the address of token row q, columns col0 + 4p .. +3 (8 bytes inside chunk col0/8 + (p>>1)); the
// second read is 16 rows further (same swizzle mask: both masks have period <= 8 rows).
template <int ROWB>
__device__ inline unsigned tr_addr(int col0, int lg, int lc) {
  const int row = 4 * lg + (lc >> 2);
  return (unsigned)(swz<ROWB>(row, (col0 >> 3) + ((lc & 3) >> 1)) + 8 * (lc & 1));
}

// WB_M = tokens per K-step (32), S = ring slots.  Measured and dropped at 2 x 2: 64-token steps at two workgroups per CU (no
// better than the register-staged kernel), four workgroups per CU (128-VGPR cap: spills in the K loop, 3x slower).
// TWO_STAGE: the workgroup's partial tile goes to its token slab's slice of a workspace with PLAIN stores, in MFMA register
// order (the atomic form spends 15-50 % of the kernel in its fp32 atomic tail: 75 MB of adds per launch at the chip's ~1.2 TB/s
// atomic rate, XCD-private lines or not -- tools/ubench_xcd_atomics.hip; the same bytes as stores: 14 us) and
// wgrad_reduce_tiles_kernel sums the slices into dW.
template <int WB_M, int NWN, int NWK, int S, int MIN_WGS, bool TWO_STAGE>
__global__ __launch_bounds__(64 * NWN * NWK, MIN_WGS) void wgrad_bf16_dma_kernel(
    const u16* __restrict__ dC, int lddc, const u16* __restrict__ A, int lda, float* __restrict__ dW,
    float* __restrict__ db, int M, int N, int K, int n_tiles, int k_tiles, int rows_per_split) {
  constexpr int WB_N = 64 * NWN, BKC = 96 * NWK, D_ROW = WB_N * 2, A_ROW = BKC * 2, NW = NWN * NWK;
  constexpr int DCH = D_ROW / 16, ACH = A_ROW / 16;
  constexpr int D_BYTES = WB_M * D_ROW, STAGE = WB_M * (D_ROW + A_ROW);
  constexpr int ND = D_BYTES / 1024 / NW, NA = WB_M * A_ROW / 1024 / NW;       // DMA instructions per wave and K-step
  static_assert(ND * NW * 1024 == D_BYTES && NA * NW * 1024 == WB_M * A_ROW, "slabs must split into whole DMA instructions");
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
  const int wn = wave / NWK, wk = wave - wn * NWK;
  const int lc = lane & 15, lg = lane >> 4;

  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(dC), 0, (int)(((size_t)(M - 1) * lddc + N) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(A), 0, (int)(((size_t)(M - 1) * lda + K) * sizeof(u16)), 0x00020000);

  // DMA instruction NW i + wave of a slab fills its LDS bytes [1024 (NW i + wave), +1024): lane l fills physical 16-B piece
  // f = 64 (NW i + wave) + l = (row, physical chunk) and fetches the logical chunk that the swizzle puts there.
  unsigned d_off[ND], a_off[NA];
  bool d_ok[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int f = (NW * i + wave) * 64 + lane;
    const int row = f / DCH, ch = (f - row * DCH) ^ swz_mask<D_ROW>(row);
    d_ok[i] = n0 + ch * 8 < N;                         // a last tile that is part empty
    d_off[i] = ((unsigned)row * (unsigned)lddc + (unsigned)(n0 + ch * 8)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = (NW * i + wave) * 64 + lane;
    const int row = f / ACH, ch = (f - row * ACH) ^ swz_mask<A_ROW>(row);
    a_off[i] = ((unsigned)row * (unsigned)lda + (unsigned)(k0 + ch * 8)) * 2u;
  }
  const unsigned d_step = (unsigned)WB_M * (unsigned)lddc * 2u, a_step = (unsigned)WB_M * (unsigned)lda * 2u;
  unsigned d_m = (unsigned)m_begin * (unsigned)lddc * 2u, a_m = (unsigned)m_begin * (unsigned)lda * 2u;
  int wslot = 0;                                         // ring slot the next slab goes to
  constexpr int PER = ND + NA;
  // piece p of the next slab (this wave's ND dC pieces, then its NA A pieces)
  auto issue_piece = [&](int p) {                       // p is a constant after unrolling
    unsigned char* base = smem + wslot * STAGE;
    if (p < ND) {
      auto dst = (__attribute__((address_space(3))) void*)(base + (NW * p + wave) * 1024);
      // the token offset is part of the range-checked VGPR offset: rows >= M (and columns >= N) read as zeros
      const unsigned off = d_ok[p < ND ? p : 0] ? d_off[p < ND ? p : 0] + d_m : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, dst, 16, (int)off, 0, 0, 0);
    } else {
      auto dst = (__attribute__((address_space(3))) void*)(base + D_BYTES + (NW * (p - ND) + wave) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst, 16, (int)(a_off[p < ND ? 0 : p - ND] + a_m), 0, 0, 0);
    }
  };
  auto advance = [&]() {
    d_m += d_step;
    a_m += a_step;
    wslot = wslot + 1 == S ? 0 : wslot + 1;
  };
  auto issue = [&]() {
#pragma unroll
    for (int p = 0; p < PER; ++p) issue_piece(p);
    advance();
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

  // per-lane fragment addresses inside a ring slot (loop invariant; the slot base and the constant offsets are added per read)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned d_addr[4], a_addr[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) d_addr[i] = tr_addr<D_ROW>(wn * 64 + i * 16, lg, lc);
#pragma unroll
  for (int j = 0; j < 6; ++j) a_addr[j] = tr_addr<A_ROW>(wk * 96 + j * 16, lg, lc);

  const int steps = (m_end - m_begin + WB_M - 1) / WB_M;
#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < steps) issue();
  int rslot = 0;
#ifdef PANGU_WGRAD_STAMP
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0};
  const unsigned long long st_begin = wg_stamp();
#endif
  for (int st = 0; st < steps; ++st) {
    WG_STAMP(t0);
    // this wave's part of slab st has landed: the S - 2 younger slabs may still be in flight (the last steps just drain)
    if (S > 2 && st + S - 2 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (S - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(t1);
    __builtin_amdgcn_s_barrier();                          // ... and everybody's; the slot read at step st - 1 is free
    asm volatile("" ::: "memory");
    WG_STAMP(t2);
    // Slab st + S - 1 goes to the slot read last at step st - 1.  A piece costs its wave ~100 cycles wherever it is issued
    // (stamps, round 4: here, spread between the MFMA groups, or at the end of the step all give the same step time; with 16
    // of 64 lanes active too: a per-instruction cost, not bytes), ~19 us of a 160-us launch.
    if (st + S - 1 < steps) issue();
    WG_STAMP(t3);
    const unsigned sb = lds0 + (unsigned)(rslot * STAGE);
    rslot = rslot + 1 == S ? 0 : rslot + 1;
#pragma unroll
    for (int kk = 0; kk < WB_M / 32; ++kk) {
      // A fragments run PD column blocks ahead of their MFMAs (an LDS round trip under the CU's load is ~150 cycles, a block's
      // four MFMAs 64; distances 1 .. 4 measure within 2 % of each other: the SIMD's other two waves cover it either way)
      constexpr int PD = 3, NR = PD + 1;
      s16x4 d0[4], d1[4], a0[NR], a1[NR];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d0[i] = lds_tr<0>(sb + d_addr[i] + kk * 32 * D_ROW);
        d1[i] = lds_tr<16 * D_ROW>(sb + d_addr[i] + kk * 32 * D_ROW);
      }
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        a0[j] = lds_tr<D_BYTES>(sb + a_addr[j] + kk * 32 * A_ROW);
        a1[j] = lds_tr<D_BYTES + 16 * A_ROW>(sb + a_addr[j] + kk * 32 * A_ROW);
      }
      bf16x8 fd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(d0[i]), "+v"(d1[i]) : "n"(2 * PD));
        fd[i] = cat(d0[i], d1[i]);
      }
      WG_STAMP(t4);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int cur = j % NR, nxt = (j + PD) % NR;
        if (j + PD < 6) {
          a0[nxt] = lds_tr<D_BYTES>(sb + a_addr[j + PD < 6 ? j + PD : 0] + kk * 32 * A_ROW);
          a1[nxt] = lds_tr<D_BYTES + 16 * A_ROW>(sb + a_addr[j + PD < 6 ? j + PD : 0] + kk * 32 * A_ROW);
        }
        // reads return in order: everything but the fragments of the blocks after j has landed
        const int younger = 2 * ((j + PD < 6 ? j + PD : 5) - j);
        if (younger == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0[cur]), "+v"(a1[cur]));
        else if (younger == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0[cur]), "+v"(a1[cur]));
        else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0[cur]), "+v"(a1[cur]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[cur]), "+v"(a1[cur]));
        const bf16x8 fa = cat(a0[cur], a1[cur]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], fa, acc[i][j], 0, 0, 0);     // D[n][k]
        __builtin_amdgcn_sched_barrier(0);               // the scheduler would sink the MFMAs below the later blocks' waits
      }
      if (want_db) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dbacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], ones, dbacc[i], 0, 0, 0);
      }
      WG_STAMP(t5);
      WG_ACC(0, t0, t1); WG_ACC(1, t1, t2); WG_ACC(2, t2, t3); WG_ACC(3, t3, t4); WG_ACC(4, t4, t5);
    }
  }
#ifdef PANGU_WGRAD_STAMP
  {
    const unsigned long long st_end = wg_stamp();
    const int w = blockIdx.x * NW + wave;
    if (lane == 0 && w < STAMP_WAVES) {
      unsigned long long* d = g_wgrad_stamp + (size_t)w * 8;
      for (int k = 0; k < 5; ++k) d[k] += st_sum[k];
      d[5] += st_end - st_begin;
      d[6] += (unsigned long long)steps;
      d[7] += 1;
    }
  }
#endif

  // lane (lg, lc) of tile (i, j): dW[n = n0 + wn*64 + 16i + 4lg + r][k = k0 + wk*96 + 16j + lc]
  if (TWO_STAGE) {
    // the workspace slice keeps the MFMA register layout: block (tile, wave, i, j) = 64 lanes x 16 B, one fully coalesced
    // 1-KB store per accumulator (the [n][k] layout needed 96 dword stores of four 64-B pieces each per wave);
    // wgrad_reduce_tiles_kernel undoes the permutation when it adds into dW
    f32x4* slice = reinterpret_cast<f32x4*>(dW) + ((size_t)split * (n_tiles * k_tiles) + tile) * (NW * 24 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(acc[i][j], slice + (wave * 24 + i * 6 + j) * 64 + lane);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int kc = k0 + wk * 96 + j * 16 + lc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * 64 + i * 16 + lg * 4 + r;
          if (n < N) atomicAdd(&dW[(size_t)n * K + kc], acc[i][j][r]);
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

// The same reduction for slices in the MFMA register layout (wgrad_bf16_dma_kernel, TWO_STAGE): element e of a slice is
// (tile, wave, accumulator i*6+j, lane), its four floats are rows n .. n+3 of column k.
__global__ __launch_bounds__(256) void wgrad_reduce_tiles_kernel(const f32x4* __restrict__ ws, float* __restrict__ dW, int e4,
                                                                 int n_valid, int N, int K, int k_tiles, int nwn, int nwk) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= e4) return;
  const int s0 = blockIdx.y * RED_CHUNK, s1 = min(n_valid, s0 + RED_CHUNK);
  const f32x4* p = ws + (size_t)s0 * e4 + e;
  f32x4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int sidx = s0;
  for (; sidx + 8 <= s1; sidx += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += __builtin_nontemporal_load(p + (size_t)u * e4);
    p += (size_t)8 * e4;
  }
  for (; sidx < s1; ++sidx) {
    acc[0] += __builtin_nontemporal_load(p);
    p += e4;
  }
  const f32x4 a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  const int lane = e & 63, ij = (e >> 6) % 24, w = (e / (64 * 24)) % (nwn * nwk), tile = e / (64 * 24 * nwn * nwk);
  const int i = ij / 6, j = ij - i * 6, wn = w / nwk, wk = w - wn * nwk, n_tile = tile / k_tiles, k_tile = tile - n_tile * k_tiles;
  const int n = n_tile * 64 * nwn + wn * 64 + i * 16 + (lane >> 4) * 4, k = k_tile * 96 * nwk + wk * 96 + j * 16 + (lane & 15);
  float* d = dW + (size_t)n * K + k;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (n + r < N) atomicAdd(d + (size_t)r * K, a[r]);
}

}  // namespace

// dW[nk] += sum of the n_valid slab slices of the workspace (shared with the fp32 weight-gradient kernel, wgrad_f32_dma.hip)
void pangu_wgrad_reduce(hipStream_t s, const float* ws, float* dW, int nk, int n_valid) {
  const int nk4 = nk / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nk4 + 255) / 256, (n_valid + RED_CHUNK - 1) / RED_CHUNK), dim3(256), 0, s,
                     reinterpret_cast<const f32x4*>(ws), dW, nk4, n_valid);
}

namespace {

// `target` = workgroups of one resident round
template <int WB_M, int NWN, int NWK, int S, int MIN_WGS>
int launch(hipStream_t s, const u16* dC, int lddc, const u16* A, int lda, float* dW, float* db, int M, int N, int K,
           int target, float* ws, size_t ws_bytes) {
  constexpr int WB_N = 64 * NWN, BKC = 96 * NWK;
  const int n_tiles = (N + WB_N - 1) / WB_N, k_tiles = K / BKC;
  const int tiles = n_tiles * k_tiles;
  int split = (target / tiles) & ~7;                                // multiple of 8: equal share per XCD
  if (split < 8) split = 8;
  int rows = ((M + split - 1) / split + 63) / 64 * 64;
  if (rows < 256) rows = 256;
  split = ((M + rows - 1) / rows + 7) & ~7;                         // grid padded to whole XCD rounds (empty slabs exit)
  const int n_valid = (M + rows - 1) / rows;                        // slabs that hold tokens (the others write nothing)
  const size_t shm = (size_t)S * WB_M * 2 * (WB_N + BKC);
  const int e4 = tiles * (WB_N * BKC / 4);                          // 16-B elements of a slice (whole tiles: N padded)
  const bool two_stage = ws != nullptr && (size_t)n_valid * e4 * 16 <= ws_bytes && n_valid > 1;
  if (two_stage) {
    auto kern = wgrad_bf16_dma_kernel<WB_M, NWN, NWK, S, MIN_WGS, true>;
    PANGU_ENSURE_DYN_LDS(kern, shm);
    hipLaunchKernelGGL(kern, dim3(tiles * split), dim3(64 * NWN * NWK), shm, s, dC, lddc, A, lda, ws, db, M, N, K, n_tiles, k_tiles, rows);
    hipLaunchKernelGGL(wgrad_reduce_tiles_kernel, dim3((e4 + 255) / 256, (n_valid + RED_CHUNK - 1) / RED_CHUNK), dim3(256), 0, s,
                       reinterpret_cast<const f32x4*>(ws), dW, e4, n_valid, N, K, k_tiles, NWN, NWK);
  } else {
    auto kern = wgrad_bf16_dma_kernel<WB_M, NWN, NWK, S, MIN_WGS, false>;
    PANGU_ENSURE_DYN_LDS(kern, shm);
    hipLaunchKernelGGL(kern, dim3(tiles * split), dim3(64 * NWN * NWK), shm, s, dC, lddc, A, lda, dW, db, M, N, K, n_tiles, k_tiles, rows);
  }
  return pangu_launch_status();
}

}  // namespace

#ifdef PANGU_WGRAD_STAMP
extern "C" int pangu_wgrad_stamp_read(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  static unsigned long long host[STAMP_WAVES * 8];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wgrad_stamp), sizeof(host));
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (int w = 0; w < STAMP_WAVES; ++w)
    for (int k = 0; k < 8; ++k) out8[k] += host[(size_t)w * 8 + k];
  for (size_t i = 0; i < (size_t)STAMP_WAVES * 8; ++i) host[i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_stamp), host, sizeof(host));
  return 0;
}
#endif

// -> PANGU_OK when launched, 1 when the shape is not covered (the caller falls back to the register-staged kernel)
int pangu_linear_wgrad_bf16_dma(hipStream_t s, const unsigned short* dC, int lddc, const unsigned short* A, int lda,
                                float* dW, float* db, int M, int N, int K, int target, float* ws, size_t ws_bytes) {
  if (K % 192 != 0) return 1;
  // the VGPR byte offset of the last slab's rows (up to M + 63, plus one row of columns) must not wrap 32 bits
  if (((size_t)M + 128) * (size_t)lddc * 2u >= 0xFFFFFFFFull || ((size_t)M + 128) * (size_t)lda * 2u >= 0xFFFFFFFFull) return 1;
  // `target` counts three 4-wave workgroups per CU; the 12-wave tiles run one per CU
  if (N % 384 == 0) return launch<32, 6, 2, 4, 1>(s, dC, lddc, A, lda, dW, db, M, N, K, target / 3, ws, ws_bytes);
  if (N % 192 == 0 && K % 384 == 0) return launch<32, 3, 4, 4, 1>(s, dC, lddc, A, lda, dW, db, M, N, K, target / 3, ws, ws_bytes);
  return launch<32, 2, 2, 2, 3>(s, dC, lddc, A, lda, dW, db, M, N, K, target, ws, ws_bytes);
}
