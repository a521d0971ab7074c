// Earth-specific 3D window attention forward, bf16 operands / fp32 softmax, gfx950.
//
// Same decomposition as attn_f32.hip (one (window, head) per 3-wave workgroup, transposed scores so that the
// probability registers are the B operand of the second product), on v_mfma_f32_16x16x32_bf16:
//   S^T tile [16 keys][16 queries] = K[16][32] . Q^T[32][16]  is ONE MFMA (K = head_dim = 32);
//   O^T[16 d][16 q] += V^T[16 d][32 keys] . P^T[32 keys][16 q]: five MFMAs per d-tile (144 keys = 4.5 k-steps).
// With 16x fewer matrix cycles than fp32 the kernel is bound by its memory/latency structure and the softmax VALU work, so:
// K is staged as [key][32] with a chunk swizzle (conflict-free b128 fragment reads), V is staged TRANSPOSED ([d][144
// keys], 336-B rows); the keys of score tiles 2u, 2u+1 are interleaved (key_of) so that one lane's eight scores are eight
// CONSECUTIVE keys: the bias arrives as one 16-B load per tile pair, the packed probabilities are the PV B-fragment in
// natural key order and the V^T A-fragment is one conflict-free ds_read_b128; scale is folded into the bias add
// (s = acc*scale + bias).  All prologue loads are issued up front (closed-form tokens), there is one barrier, the bias row of
// tile i+1 is prefetched under tile i, and the two heads sharing a token's 128-B line run back to back on one XCD.
#include "common.h"
#include <stdlib.h>

#include "attn_bf16_tile.h"

namespace {

// Latency structure: a workgroup is short (27 KB in, 9 KB out, 99 MFMAs), so its life is a chain of memory round trips
// unless they are overlapped.  Every global load the workgroup needs up front -- the K/V rows it stages, the Q fragments
// of all three query tiles of each wave and the first tile's bias row -- is issued before anything waits (source
// tokens come from the closed form, not through LDS), the bias row of tile i+1 is requested before tile i computes, and
// there is ONE barrier.
template <bool SHIFTED>
__global__ __launch_bounds__(192, 3) void window_attn_bf16_kernel(const u16* __restrict__ qkv,
                                                                  const u16* __restrict__ qkv_bias,
                                                                  const u16* __restrict__ esb, u16* __restrict__ out,
                                                                  float* __restrict__ lse, WinGeom g, int C, int heads,
                                                                  int n_pairs) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[PANGU_WTOK * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[VT_BYTES_PAD];

  // Block order: a bf16 head slice is 64 B, half a cache line, so the two heads that share each token's 128-B line run
  // back to back on ONE XCD (blocks b, b+8, .. share an XCD/L2): unit = (window type, head pair), then the longitude
  // window, then the head within the pair.  The second head's q/k/v reads and the out writes hit lines already in
  // that L2; the unit's two bias tiles stay L2-resident across its nLon windows.
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  int pair, l;
  if (heads & 1) {
    pair = (local / g.nLon) * 8 + xcd;
    l = local % g.nLon;
  } else {
    const int sub = local & 1, wl = local >> 1;
    const int unit = (wl / g.nLon) * 8 + xcd;
    l = wl % g.nLon;
    pair = 2 * unit + sub;
  }
  if (pair >= n_pairs) return;
  const int t = pair / heads, hd = pair - t * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int C3 = 3 * C;
  const u16* bias_tile = esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK;

  // ---- every up-front global load, back to back
  u32x4 kv[3], vv[3];                                      // 144 rows x 4 chunks of 8 bf16 = 3 per thread
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int f = tid + 192 * i, n = f >> 2, ch = f & 3;
    const int tok = win_src_token(g, l, t, n, SHIFTED);
    const u16* src = tok >= 0 ? qkv + (size_t)tok * C3 : qkv_bias;
    kv[i] = *reinterpret_cast<const u32x4*>(src + C + hd * 32 + ch * 8);
    vv[i] = *reinterpret_cast<const u32x4*>(src + 2 * C + hd * 32 + ch * 8);
  }
  int qtok[3];
  bf16x8 qf[3];                                            // Q fragment (B operand): Q[query][d = 8lg .. 8lg+7]
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int qn = (wave + 3 * i) * 16 + lq;
    qtok[i] = win_src_token(g, l, t, qn, SHIFTED);
    qf[i] = *reinterpret_cast<const bf16x8*>((qtok[i] >= 0 ? qkv + (size_t)qtok[i] * C3 : qkv_bias) + hd * 32 + lg * 8);
  }
  BiasRow b0 = load_bias_row(bias_tile, wave * 16 + lq, lg);

  bool zcut = false, hcut = false;
  unsigned long long kz_bits = 0ull, kh_bits = 0ull;
  if (SHIFTED) {
    const int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
    zcut = zwin == g.nZw - 1;
    hcut = hwin == g.nHw - 1;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kn = key_of(j, lg * 4 + r);
        if (kn >= 72) kz_bits |= 1ull << (4 * j + r);
        if (((kn / 12) % 6) < 3) kh_bits |= 1ull << (4 * j + r);
      }
  }

  // ---- stage K ([key][32], swizzled) and V^T ([d][key])
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int f = tid + 192 * i, n = f >> 2, ch = f & 3;
    *reinterpret_cast<u32x4*>(Ks + kswz(n, ch)) = kv[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<u16*>(Vt + vt_off<false>(ch * 8 + 2 * e, n)) = (u16)(vv[i][e] & 0xFFFFu);
      *reinterpret_cast<u16*>(Vt + vt_off<false>(ch * 8 + 2 * e + 1, n)) = (u16)(vv[i][e] >> 16);
    }
  }
  __syncthreads();

  const BiasRow b1 = load_bias_row(bias_tile, (wave + 3) * 16 + lq, lg);
  attn_tile<SHIFTED, false>(Ks, Vt, qf[0], b0, wave * 16 + lq, qtok[0], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
  b0 = load_bias_row(bias_tile, (wave + 6) * 16 + lq, lg);
  attn_tile<SHIFTED, false>(Ks, Vt, qf[1], b1, (wave + 3) * 16 + lq, qtok[1], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
  attn_tile<SHIFTED, false>(Ks, Vt, qf[2], b0, (wave + 6) * 16 + lq, qtok[2], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
}

// ---- QKV projection fused in ----------------------------------------------------------------------------------------
// window_attn_qkv_bf16_kernel: the same (window, head) workgroup first COMPUTES its q, k, v from the window's 144 input
// rows and the head's 96 rows of linear1 (reference layers.py:365-374) instead of reading a (tokens x 3C) qkv tensor that
// a separate GEMM wrote: that tensor (0.6 GB per C = 192 block in bf16, written and read once) never exists.
//   * K-loop over the C input channels in steps of 32: the step's x slice (144 rows x 64 B, gathered with the window's
//     closed-form token indices; zero-pad rows arrive as zeros from the buffer range check, so their q/k/v are the bias,
//     layers.py:192) and weight slice (96 rows x 64 B) travel L2 -> LDS by LDS-DMA into a ring of 3 slots (source-side
//     XOR swizzle, counted vmcnt, one raw barrier per step); wave w owns token tiles 3w..3w+2 and all six 16-row weight
//     tiles: 9 conflict-free ds_read_b128 feed 18 MFMAs per step;
//   * q and k are computed transposed (weights as the A operand: d on the register index, token on the lane), v the
//     other way round (token on the register index, d on the lane): the q accumulators, packed, ARE the B operand of the
//     score product; a lane's eight k values are the d set {4g..4g+3} u {16+4g..16+4g+3} -- the same permuted order in
//     q and k, which a dot product does not see -- and go to the K image as ONE 16-B write; a lane's four v values are
//     four consecutive tokens of one d: one 8-B write into the transposed V image.  No transposes, no shuffles;
//   * the biases are the accumulators' initial values; then the three query tiles of the wave run through attn_tile
//     exactly as in the kernel above.
constexpr int QK_ROWS = PANGU_WTOK + 96;                 // rows of one ring slot: 144 x rows, then q/k/v weight rows
constexpr int QK_SLOT = QK_ROWS * 64;                    // slot bytes at 32 channels per K-step (BK = 64: twice that)

__device__ inline int kswz64(int row, int chunk) {      // 64-byte rows, 4 chunks: F = {0,2,3,1}[(row>>2)&3]
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((chunk ^ f) << 4);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef PANGU_ATTN_STAMP
// Diagnostic build only (tools/ablate_attn.py): s_memtime sums over all waves: [0] K-loop, [1] q/K/V staging + barrier,
// [2] the three attention tiles, [3] whole kernel, [4] waves, [5] prologue up to the first K-step.
constexpr int STAMP_WAVES = 80000;
__device__ unsigned long long g_attn_stamp[STAMP_WAVES * 8];      // per wave: no atomics (they would dominate the timing)
__device__ __forceinline__ unsigned long long attn_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif

// win_src_token (common.h) with the window type already split into (zwin, hwin): the split is a runtime division of a
// workgroup-uniform value that the compiler would redo on the vector ALU for each of the kernel's eight token lookups.
__device__ inline void split_type(const WinGeom& g, int t, int& zwin, int& hwin) {
  zwin = 0;
  hwin = t;
#pragma unroll 1
  for (int k = 1; k < g.nZw; ++k)            // nZw = 4 for the model's 8 levels: three scalar compares
    if (t >= k * g.nHw) { zwin = k; hwin = t - k * g.nHw; }
}
__device__ inline int win_src_token_zh(const WinGeom& g, int l, int zwin, int hwin, int n, int shifted) {
  int zi = n / 72, r = n - zi * 72, hi = r / 12, wi = r - hi * 12;
  int z = 2 * zwin + zi, h = 6 * hwin + hi, w = 12 * l + wi;
  if (shifted) {
    z += 1; if (z >= g.Z) z -= g.Z;
    h += 3; if (h >= g.Hp) h -= g.Hp;
    w += 6; if (w >= g.W) w -= g.W;
  }
  return h >= g.H ? -1 : (z * g.H + h) * g.W + w;
}

// Waves per SIMD the register allocator must leave room for (the second __launch_bounds__ argument).  2 = the compiler's free
// choice (138-156 VGPRs -> three waves per SIMD, four 3-wave workgroups per CU); 4 = capped at 128 VGPRs (five workgroups per CU):
// the round-6 A/B build (tools/ab_lib.sh, profiles/r06_attn_qkv_occ4_ab.md).
#ifndef PANGU_ATTN_QKV_MIN_WAVES
#define PANGU_ATTN_QKV_MIN_WAVES 2
#endif

template <bool SHIFTED, int C>
__global__ __launch_bounds__(192, PANGU_ATTN_QKV_MIN_WAVES) void window_attn_qkv_bf16_kernel(const u16* __restrict__ x, int ldx,
                                                                      const u16* __restrict__ wqkv,
                                                                      const float* __restrict__ bqkv,
                                                                      const u16* __restrict__ esb, u16* __restrict__ out,
                                                                      float* __restrict__ lse, WinGeom g, int n_tok,
                                                                      int heads, int n_pairs) {
  constexpr int QK_RING = 2, BK = 32;        // ring of two 32-channel K-steps (the measured winner; see launch_attn_qkv)
#ifdef PANGU_ATTN_STAMP
  const unsigned long long st0 = attn_stamp();
#endif
  constexpr int KS = C / BK;
  constexpr int CH = BK / 8;                 // 16-B chunks per slot row
  constexpr int ROWB = BK * 2;               // bytes per slot row
  constexpr int SLOT = QK_ROWS * ROWB;
  constexpr int RPI = 1024 / ROWB;           // rows per LDS-DMA instruction (1 KB)
  constexpr int NIW = QK_ROWS / RPI / 3;     // instructions per wave and step
  constexpr int NIX = PANGU_WTOK / RPI;      // the first NIX instructions of a step carry x rows, the rest weight rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the K / V^T images REUSE the ring's memory (they are written after the K-loop, behind a barrier): 45 KB per workgroup
  // with a ring of 3, so three workgroups per CU each keep two K-steps of LDS-DMA in flight -- the loop is bound by the
  // DMA round trip (a step's 15 KB per workgroup), not by its 54 MFMAs
  unsigned char* const ring = smem;                                  // QK_RING slots
  unsigned char* const Ks = smem;                                    // [144][64 B]
  unsigned char* const Vt = Ks + PANGU_WTOK * 64;                    // V^T image (swizzled layout, VT_BYTES_SWZ)

  // Block order: blocks b, b+8, .. share an XCD (its L2).  Every head of a window reads the SAME 144 input rows, so the
  // heads of one window run back to back on one XCD (x leaves HBM once, not `heads` times: PMC FETCH_SIZE 0.67 -> ...
  // GB per launch, profiles/), then the next longitude window of the same type (its `heads` bias tiles stay in that L2).
  // (3-D grid since round 6: x = xcd + 8 * head, y = longitude window, z = type / 8 -- the SAME dispatch order as the former 1-D grid
  // b = xcd + 8 * (head + heads * (l + nLon * z)), x fastest and gridDim.x a multiple of 8, but no runtime division: five uniform
  // 32-bit divisions cost ~110 of the kernel's 1 300 vector instructions per wave)
  const int xcd = blockIdx.x & 7, hd = blockIdx.x >> 3;
  const int l = blockIdx.y;
  const int t = blockIdx.z * 8 + xcd;
  const int pair = t * heads + hd;
  if (pair >= n_pairs) return;
  int zwin, hwin;
  split_type(g, t, zwin, hwin);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const u16* bias_tile = esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(x), 0, (int)(((size_t)(n_tok - 1) * ldx + C) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(wqkv), 0, 3 * C * C * (int)sizeof(u16), 0x00020000);

  // ---- LDS-DMA plan: instruction q covers slot rows RPI q .. RPI q + RPI - 1 (1 KB); q < NIX the x rows, then the weight rows.
  // Wave w issues q = w, w+3, ..  This lane fills (row RPI q + lane / CH, physical chunk lane % CH) with the logical chunk
  // (lane % CH) ^ F(row), F = {0,2,3,1}[(row>>2)&3] (conflict-free b128 reads of the 64-byte rows).
  auto fsw = [](int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; };
  unsigned voff[NIW];
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    const int q = wave + 3 * i;
    const int row = RPI * q + lane / CH;
    const int c = (lane % CH) ^ fsw(row);
    if (q < NIX) {
      const int tok = win_src_token_zh(g, l, zwin, hwin, row, SHIFTED);
      voff[i] = tok >= 0 ? ((unsigned)tok * (unsigned)ldx + c * 8) * 2u : 0x7FFFFFF0u;       // pad row: out of range -> zeros
    } else {
      const int r = row - PANGU_WTOK, which = r >> 5, d = r & 31;
      voff[i] = ((unsigned)(which * C + hd * 32 + d) * (unsigned)C + c * 8) * 2u;
    }
  }
  auto issue = [&](int ks) {
    unsigned char* base = ring + (ks % QK_RING) * SLOT;
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const int q = wave + 3 * i;
      auto dst = (__attribute__((address_space(3))) void*)(base + q * 1024);
      if (q < NIX) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, dst, 16, (int)voff[i], ks * ROWB, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, (int)voff[i], ks * ROWB, 0, 0);
    }
  };
  issue(0);

  // the first tile's bias row and the query-token indices are requested / computed under the K-loop
  const int tile0 = 3 * wave;
  int qtok[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) qtok[i] = win_src_token_zh(g, l, zwin, hwin, (tile0 + i) * 16 + lq, SHIFTED);

  // ---- accumulators: [rt 0,1 = q | 2,3 = k][token tile] transposed (d = 4lg + r on the registers, token on the lane);
  //      [rt 4,5 = v] token 4lg + r on the registers, d = 16(rt-4) + lq on the lane.  Initial value = bias.
  f32x4 acc[6][3];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bqkv + (rt >> 1) * C + hd * 32 + (rt & 1) * 16 + 4 * lg);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[rt][i] = bv;
  }
#pragma unroll
  for (int rt = 4; rt < 6; ++rt) {
    const float bv = bqkv[2 * C + hd * 32 + (rt - 4) * 16 + lq];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[rt][i] = f32x4{bv, bv, bv, bv};
  }

#ifdef PANGU_ATTN_STAMP
  const unsigned long long st1 = attn_stamp();
#endif
  for (int ks = 0; ks < KS; ++ks) {
    wait_vmcnt<0>();                                               // step ks landed
    // every fragment read of the previous step must have RETURNED before this wave releases the barrier: behind it the
    // other waves re-request that ring slot, and an LDS-DMA write can overtake a ds_read that is still queued (seen on
    // MI355X as ~1 wrong workgroup in 3000 when the compiler had sunk the last reads' wait below the barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                  // ... for every wave; the slot of step ks-1 is free
    asm volatile("" ::: "memory");
    if (ks + 1 < KS) issue(ks + 1);
    const unsigned char* slot = ring + (ks % QK_RING) * SLOT;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 fx[3], fw[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = (tile0 + i) * 16 + lq;
        fx[i] = *reinterpret_cast<const bf16x8*>(slot + row * ROWB + (((kk * 4 + lg) ^ fsw(row)) << 4));
      }
#if PANGU_ATTN_QKV_MIN_WAVES >= 4
      // register diet of the 128-VGPR build: the six weight fragments in three pairs instead of all at once (24 -> 8 registers)
#pragma unroll
      for (int rp = 0; rp < 3; ++rp) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = PANGU_WTOK + (2 * rp + h) * 16 + lq;
          fw[h] = *reinterpret_cast<const bf16x8*>(slot + row * ROWB + (((kk * 4 + lg) ^ fsw(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int rt = 2 * rp + h;
            if (rt < 4) acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[h], fx[i], acc[rt][i], 0, 0, 0);
            else acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fw[h], acc[rt][i], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
#else
#pragma unroll
      for (int rt = 0; rt < 6; ++rt) {
        const int row = PANGU_WTOK + rt * 16 + lq;
        fw[rt] = *reinterpret_cast<const bf16x8*>(slot + row * ROWB + (((kk * 4 + lg) ^ fsw(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[rt], fx[i], acc[rt][i], 0, 0, 0);
#pragma unroll
        for (int rt = 4; rt < 6; ++rt) acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fw[rt], acc[rt][i], 0, 0, 0);
      }
#endif
    }
  }

#ifdef PANGU_ATTN_STAMP
  const unsigned long long st2 = attn_stamp();
#endif
  // ---- q fragments (registers), K image and V^T image (LDS, over the ring: every wave must be done reading it)
  BiasRow b0 = load_bias_row(bias_tile, tile0 * 16 + lq, lg);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 qf[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    qf[i] = __builtin_bit_cast(bf16x8, u32x4{pack2(acc[0][i][0], acc[0][i][1]), pack2(acc[0][i][2], acc[0][i][3]),
                                             pack2(acc[1][i][0], acc[1][i][1]), pack2(acc[1][i][2], acc[1][i][3])});
    const int n = (tile0 + i) * 16 + lq;
    *reinterpret_cast<u32x4*>(Ks + kswz(n, lg)) = u32x4{pack2(acc[2][i][0], acc[2][i][1]), pack2(acc[2][i][2], acc[2][i][3]),
                                                        pack2(acc[3][i][0], acc[3][i][1]), pack2(acc[3][i][2], acc[3][i][3])};
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      *reinterpret_cast<u32x2*>(Vt + vt_off<true>(dt * 16 + lq, (tile0 + i) * 16 + 4 * lg)) =
          u32x2{pack2(acc[4 + dt][i][0], acc[4 + dt][i][1]), pack2(acc[4 + dt][i][2], acc[4 + dt][i][3])};
  }

  bool zcut = false, hcut = false;
  unsigned long long kz_bits = 0ull, kh_bits = 0ull;
  if (SHIFTED) {
    zcut = zwin == g.nZw - 1;
    hcut = hwin == g.nHw - 1;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kn = key_of(j, lg * 4 + r);
        if (kn >= 72) kz_bits |= 1ull << (4 * j + r);
        if (((kn / 12) % 6) < 3) kh_bits |= 1ull << (4 * j + r);
      }
  }
  __syncthreads();
#ifdef PANGU_ATTN_STAMP
  const unsigned long long st3 = attn_stamp();
#endif
#if PANGU_ATTN_QKV_MIN_WAVES >= 4
  // the 128-VGPR build: two key halves per tile, ONE bias row live (the next tile's row refills it half by half)
  attn_tile_halves<SHIFTED, true>(Ks, Vt, qf[0], b0, bias_tile + (size_t)((tile0 + 1) * 16 + lq) * PANGU_WTOK, tile0 * 16 + lq, qtok[0],
                                  lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
  attn_tile_halves<SHIFTED, true>(Ks, Vt, qf[1], b0, bias_tile + (size_t)((tile0 + 2) * 16 + lq) * PANGU_WTOK, (tile0 + 1) * 16 + lq,
                                  qtok[1], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
  attn_tile_halves<SHIFTED, true>(Ks, Vt, qf[2], b0, nullptr, (tile0 + 2) * 16 + lq, qtok[2], lq, lg, zcut, hcut, kz_bits, kh_bits, out,
                                  lse, C, heads, hd);
#else
  const BiasRow b1 = load_bias_row(bias_tile, (tile0 + 1) * 16 + lq, lg);
  attn_tile<SHIFTED, true>(Ks, Vt, qf[0], b0, tile0 * 16 + lq, qtok[0], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
  b0 = load_bias_row(bias_tile, (tile0 + 2) * 16 + lq, lg);
  attn_tile<SHIFTED, true>(Ks, Vt, qf[1], b1, (tile0 + 1) * 16 + lq, qtok[1], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
  attn_tile<SHIFTED, true>(Ks, Vt, qf[2], b0, (tile0 + 2) * 16 + lq, qtok[2], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
#endif
#ifdef PANGU_ATTN_STAMP
  const unsigned long long st4 = attn_stamp();
  if (lane == 0 && (int)(blockIdx.x * 3 + wave) < STAMP_WAVES) {
    unsigned long long* d = g_attn_stamp + (size_t)(blockIdx.x * 3 + wave) * 8;
    d[0] = st2 - st1; d[1] = st3 - st2; d[2] = st4 - st3; d[3] = st4 - st0; d[4] = 1ull; d[5] = st1 - st0;
  }
#endif
}


}  // namespace

extern "C" int pangu_window_attn_fwd_bf16(pangu_stream_t stream, const void* qkv, const void* qkv_bias, const void* esb,
                                          void* out, float* lse, int Z, int H, int W, int C, int heads, int shifted) {
  if (!qkv || !qkv_bias || !esb || !out) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  const WinGeom g = make_geom(Z, H, W);
  const int n_pairs = g.types * heads;
  const int grid = (heads & 1) ? ((n_pairs + 7) / 8) * 8 * g.nLon : ((n_pairs / 2 + 7) / 8) * 8 * g.nLon * 2;
  hipStream_t s = (hipStream_t)stream;
  if (shifted)
    hipLaunchKernelGGL(window_attn_bf16_kernel<true>, dim3(grid), dim3(192), 0, s, (const u16*)qkv, (const u16*)qkv_bias,
                       (const u16*)esb, (u16*)out, lse, g, C, heads, n_pairs);
  else
    hipLaunchKernelGGL(window_attn_bf16_kernel<false>, dim3(grid), dim3(192), 0, s, (const u16*)qkv,
                       (const u16*)qkv_bias, (const u16*)esb, (u16*)out, lse, g, C, heads, n_pairs);
  return pangu_launch_status();
}

static int launch_attn_qkv(pangu_stream_t stream, const void* x, int ldx, const void* w_qkv, const float* b_qkv,
                           const void* esb, void* out, float* lse, int Z, int H, int W, int C, int heads, int shifted) {
  if (!x || !w_qkv || !b_qkv || !esb || !out) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM || (C != 192 && C != 384) || ldx < C || (ldx & 7)) return PANGU_E_SHAPE;
  const int n_tok = Z * H * W;
  // the pad-row sentinels of the K-loop (0x7FFFFFF0) and of the side stores must stay out of range
  if (!pangu_fits_u32(n_tok, ldx, 2) || (size_t)n_tok * ldx * 2 >= 0x7FFFFFF0ull) return PANGU_E_RANGE;
  const WinGeom g = make_geom(Z, H, W);
  const int n_pairs = g.types * heads;
  const dim3 grid(8 * heads, g.nLon, (g.types + 7) / 8);
  // ring of 2 x 32 channels, four workgroups per CU.  Measured and removed (round 4; DESIGN.md keeps the numbers): ring of 3 (three
  // workgroups per CU, -5 %), one slot of 64 channels (+-1 %), ring of 2 x 64 channels (two workgroups per CU, -20 %), HG heads of a
  // window per workgroup sharing the x slice (-10..-30 %), window rows register-resident with the heads looped (-8 %), and the
  // training variant with qkv + lse side outputs (+0.3 ms per step): resident workgroups decide, not the pipeline inside one.
  const size_t shm = (size_t)2 * QK_SLOT;                          // >= the K + V^T images (9216 + 12288 B) that reuse it
  hipStream_t s = (hipStream_t)stream;
#define PANGU_QKV_LAUNCH(SH, CC)                                                                                          \
  do {                                                                                                                    \
    auto kern = window_attn_qkv_bf16_kernel<SH, CC>;                                                                      \
    PANGU_ENSURE_DYN_LDS(kern, shm);                                                                                      \
    hipLaunchKernelGGL(kern, grid, dim3(192), shm, s, (const u16*)x, ldx, (const u16*)w_qkv, b_qkv,                 \
                       (const u16*)esb, (u16*)out, lse, g, n_tok, heads, n_pairs);                                        \
  } while (0)
  if (C == 192) {
    if (shifted) PANGU_QKV_LAUNCH(true, 192); else PANGU_QKV_LAUNCH(false, 192);
  } else {
    if (shifted) PANGU_QKV_LAUNCH(true, 384); else PANGU_QKV_LAUNCH(false, 384);
  }
#undef PANGU_QKV_LAUNCH
  return pangu_launch_status();
}

extern "C" int pangu_window_attn_qkv_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_qkv,
                                              const float* b_qkv, const void* esb, void* out, float* lse, int Z, int H, int W,
                                              int C, int heads, int shifted) {
  return launch_attn_qkv(stream, x, ldx, w_qkv, b_qkv, esb, out, lse, Z, H, W, C, heads, shifted);
}

#ifdef PANGU_ATTN_STAMP
extern "C" int pangu_attn_stamp_read(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  static unsigned long long host[STAMP_WAVES * 8];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamp), sizeof(host));
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (int w = 0; w < STAMP_WAVES; ++w)
    for (int k = 0; k < 8; ++k) out8[k] += host[(size_t)w * 8 + k];
  for (size_t i = 0; i < (size_t)STAMP_WAVES * 8; ++i) host[i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), host, sizeof(host));
  return 0;
}
#endif
