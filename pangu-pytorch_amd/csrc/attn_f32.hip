// Earth-specific 3D window attention, fp32, gfx950.
//
// One workgroup (3 waves) per (longitude window l, window type t, head): 144 tokens x 32 dims.
// Roll / pad / partition / reverse / crop are address arithmetic (win_src_token); the shift mask is closed
// form; nothing of size (..,144,144) ever reaches HBM.
//   K, V of the window are staged in LDS as [144][32] floats with the 16-B chunk index XOR-swizzled by a function of
//   the key (conflict-free for the b128 row reads AND the b32 column reads): 37.4 KB per workgroup, so FOUR
//   workgroups = 12 waves = 3 per SIMD share a CU (the padded [144][36] image allowed only 9 waves: 3/2/2/2).
//   Each wave owns 3 query tiles of 16 rows.  Per tile it computes the TRANSPOSED scores
//     S^T[key][query] = K . (scale*Q)^T      with v_mfma_f32_16x16x4_f32 (9 key tiles x 8 k-steps),
//   so a query's 144 scores live in 16 lanes-groups' registers: softmax = in-lane reduce + 2 shuffles,
//   and the probability registers are directly the B operand of the second product
//     O^T[d][query] = V^T[d][key] . P^T[key][query]
//   (the MFMA k index is the accumulator's register index: no LDS round trip, no lane movement).
// Workgroup order is XCD-aware: the nLon windows sharing one (type, head) bias tile (83 KB) are consecutive
// on one XCD, so the 62 MB bias tensor streams from HBM once per block and is re-read from L2.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KV_LD = 32;

// float offset of element (key, d): chunk (d>>2) XOR f(key), f = key bits {1,3,2} -> chunk bits {0,1,2}
__device__ inline int kvoff(int key, int d) {
  const int f = ((key >> 1) & 1) | (((key >> 3) & 1) << 1) | (((key >> 2) & 1) << 2);
  return key * KV_LD + ((((d >> 2) ^ f) << 2) | (d & 3));
}

struct BiasRowF {          // one query row of the bias tile, this lane's 36 keys (key 16j + 4lg + r)
  f32x4 v[9];
};

// COMPACT: the row comes from the paper's compact table of this (type, head) -- 3 312 floats instead of 144 x 144 --
// through the position index of reference layers.py:319-357 in closed form:
//   index(query, key) = (zq + 2 zk) * 828 + (hq + 6 hk) * 23 + (wq - wk + 11),  n = 72 z + 12 h + w.
// The 4 keys 16j + 4lg + r of a lane share (zk, hk) (12 | 4-aligned runs) and their wk rises with r, so their entries are 4
// CONSECUTIVE floats in falling order: one dword-aligned 16-B load, reversed.  Same values in the same registers as the
// expanded tile gives: the two modes are bit-identical (tests/test_gpu_extras.py).
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
constexpr int PANGU_COMPACT_BIAS = 3312;

template <bool COMPACT>
__device__ inline BiasRowF load_bias_row(const float* __restrict__ bias_tile, int qn, int lg) {
  BiasRowF b;
  if (COMPACT) {
    const int zq = qn / 72, hq = (qn / 12) % 6, wq = qn % 12;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int k0 = j * 16 + lg * 4;
      const int zk = k0 / 72, hk = (k0 / 12) % 6, wk = k0 % 12;
      const int idx = (zq + 2 * zk) * 828 + (hq + 6 * hk) * 23 + (wq - wk + 11);      // key k0; key k0 + r: idx - r
      const f32x4 v = *reinterpret_cast<const f32x4_a4*>(bias_tile + idx - 3);
      b.v[j] = f32x4{v[3], v[2], v[1], v[0]};
    }
  } else {
    const float* brow = bias_tile + (size_t)qn * PANGU_WTOK + lg * 4;
#pragma unroll
    for (int j = 0; j < 9; ++j) b.v[j] = *reinterpret_cast<const f32x4*>(brow + j * 16);
  }
  return b;
}

// ---- the three phases of one 16-query tile of one wave -----------------------------------------------------------------
// S^T = bias^T + K (scale Q)^T : 9 key tiles, `s` enters holding the bias row (the accumulators' initial value).
// Key tiles go three at a time so consecutive MFMAs hit different accumulators (16x16x4 f32: 32-cycle issue, 40-cycle
// dependent latency).  lane holds S^T[key = 16j + 4lg + r][query], r = 0..3.
__device__ __forceinline__ void s_chunk(f32x4 (&s)[9], const float* Ksq, const f32x4 q0, const f32x4 q1, int lq, int lg,
                                        int j0) {
  f32x4 k0[3], k1[3];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    k0[jj] = *reinterpret_cast<const f32x4*>(&Ksq[kvoff((j0 + jj) * 16 + lq, lg * 4)]);
    k1[jj] = *reinterpret_cast<const f32x4*>(&Ksq[kvoff((j0 + jj) * 16 + lq, 16 + lg * 4)]);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
      s[j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[jj][ks], q0[ks], s[j0 + jj], 0, 0, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
      s[j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[jj][ks], q1[ks], s[j0 + jj], 0, 0, 0);
}
__device__ __forceinline__ void s_phase(f32x4 (&s)[9], const float* Ksq, const f32x4 q0, const f32x4 q1, int lq, int lg) {
  s_chunk(s, Ksq, q0, q1, lq, lg, 0);
  s_chunk(s, Ksq, q0, q1, lq, lg, 3);
  s_chunk(s, Ksq, q0, q1, lq, lg, 6);
}

// shift mask (closed form, only in the cut window types) and row max
template <bool SHIFTED>
__device__ __forceinline__ float softmax_max(f32x4 (&s)[9], int qn, bool zcut, bool hcut, unsigned long long kz_bits,
                                             unsigned long long kh_bits) {
  float mx = -INFINITY;
  if (SHIFTED) {
    if (zcut || hcut) {
      const bool zq = qn >= 72, hq = ((qn / 12) % 6) < 3;
      // bit (4j + r) of the key-class words: this lane's 36 keys kn = 16j + 4lg + r
      const unsigned long long zsel = zq ? ~kz_bits : kz_bits;      // keys whose z-half differs from the query's
      const unsigned long long hsel = hq ? ~kh_bits : kh_bits;
      const unsigned long long cut = (zcut ? zsel : 0ull) | (hcut ? hsel : 0ull);
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((cut >> (4 * j + r)) & 1ull) s[j][r] += -100.0f;
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[j][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  return mx;
}
// softmax numerators of key tiles [JLO, JHI) in place; returns their partial sum
template <int JLO, int JHI>
__device__ __forceinline__ float softmax_exp(f32x4 (&s)[9], float mx) {
  float sum = 0.f;
  const float nmx = -mx * 1.4426950408889634f;      // exp(s - mx) = exp2(s*log2e - mx*log2e): one fma + v_exp_f32
#pragma unroll
  for (int j = JLO; j < JHI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(fmaf(s[j][r], 1.4426950408889634f, nmx));
      s[j][r] = e;
      sum += e;
    }
  return sum;
}
__device__ __forceinline__ float softmax_sum(float sum) {
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  return sum;
}

// O^T = V^T P^T : two 16-dim tiles, k = key; normalise and store.  lane holds O^T[d = 16*dt + 4lg + r][query].
// The A operands V[key 16j + 4lg + r][d] are b32 column reads of the row-major V image; the eight reads of key tile
// j+1 are issued as a batch before the eight MFMAs of tile j (read just ahead of each MFMA, every MFMA would wait a
// full LDS round trip).
__device__ __forceinline__ void pv_phase(const f32x4 (&s)[9], const float* Vsq, float mx, float sum, int qtok, int lq, int lg,
                                         float* __restrict__ out, float* __restrict__ lse, int C, int heads, int hd) {
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
  float va[4], vb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    va[r] = Vsq[kvoff(lg * 4 + r, lq)];
    vb[r] = Vsq[kvoff(lg * 4 + r, 16 + lq)];
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    float ca[4], cb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { ca[r] = va[r]; cb[r] = vb[r]; }
    if (j < 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = (j + 1) * 16 + lg * 4 + r;
        va[r] = Vsq[kvoff(key, lq)];
        vb[r] = Vsq[kvoff(key, 16 + lq)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[r], s[j][r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cb[r], s[j][r], o1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (qtok >= 0) {
    const float inv = 1.0f / sum;
    o0 *= inv; o1 *= inv;
    float* dst = out + (size_t)qtok * C + hd * 32 + lg * 4;
    *reinterpret_cast<f32x4*>(dst) = o0;
    *reinterpret_cast<f32x4*>(dst + 16) = o1;
    if (lse && lg == 0) lse[(size_t)qtok * heads + hd] = mx + __logf(sum);
  }
}

// K/V fragments are the same for every query tile; an opaque zero offset per use keeps the compiler from holding
// 200+ tile-invariant fragment registers live across the tiles (which would halve the occupancy).
__device__ __forceinline__ const float* opaque(const float* p) {
  int lz = 0;
  asm volatile("" : "+v"(lz));
  return p + lz;
}

// Current tile's softmax (scores in sc) overlapped with the next tile's 72 score MFMAs (into sn) INSIDE one wave: each
// MFMA (8 passes = 32 cycles in the matrix core) is followed in program order by a few independent softmax VALU
// instructions that execute in its shadow; a sched_barrier after every slot pins that order (the scheduler otherwise
// clusters the MFMAs and leaves the VALU work serial behind them).  Chunks: [24 MFMAs | row max], [24 | exp of key
// tiles 0-4.5], [24 | exp of the rest], so each chunk has independent work for both pipes.
template <bool SHIFTED>
__device__ __forceinline__ void softmax_with_next_scores(f32x4 (&sc)[9], f32x4 (&sn)[9], const float* Ks, const f32x4 q0,
                                                         const f32x4 q1, int qn, int lq, int lg, bool zcut, bool hcut,
                                                         unsigned long long kz_bits, unsigned long long kh_bits, float& mx_out,
                                                         float& sum_out) {
  if (SHIFTED) {
    if (zcut || hcut) {
      const bool zq = qn >= 72, hq = ((qn / 12) % 6) < 3;
      const unsigned long long zsel = zq ? ~kz_bits : kz_bits;      // keys whose z-half differs from the query's
      const unsigned long long hsel = hq ? ~kh_bits : kh_bits;
      const unsigned long long cut = (zcut ? zsel : 0ull) | (hcut ? hsel : 0ull);
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((cut >> (4 * j + r)) & 1ull) sc[j][r] += -100.0f;
    }
  }
  float mx = -INFINITY, sum = 0.f;
  const float LOG2E = 1.4426950408889634f;
  float nmx = 0.f;                                            // -mx * log2(e)
#pragma unroll
  for (int chunk = 0; chunk < 3; ++chunk) {
    const int j0 = 3 * chunk;
    const float* Ksq = opaque(Ks);
    f32x4 k0[3], k1[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      k0[jj] = *reinterpret_cast<const f32x4*>(&Ksq[kvoff((j0 + jj) * 16 + lq, lg * 4)]);
      k1[jj] = *reinterpret_cast<const f32x4*>(&Ksq[kvoff((j0 + jj) * 16 + lq, 16 + lg * 4)]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int half = m / 12, ks = (m % 12) / 3, jj = m % 3;
      sn[j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(half ? k1[jj][ks] : k0[jj][ks], half ? q1[ks] : q0[ks],
                                                         sn[j0 + jj], 0, 0, 0);
      if (chunk == 0) {
        if (m < 18) {                                         // row max: two of the 36 scores per slot
          const int e = 2 * m;
          mx = fmaxf(mx, fmaxf(sc[e >> 2][e & 3], sc[(e + 1) >> 2][(e + 1) & 3]));
        }
      } else {
        const int e = 18 * (chunk - 1) + m;                   // softmax numerators: one score per slot (18 per chunk)
        if (m < 18) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sc[e >> 2][e & 3], LOG2E, nmx));
          sc[e >> 2][e & 3] = p;
          sum += p;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (chunk == 0) {
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      nmx = -mx * LOG2E;
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  mx_out = mx;
  sum_out = sum;
}

// Latency structure: every global load the workgroup needs up front -- the K/V rows it stages (source tokens from the
// closed form, not through LDS), the Q fragments of all three query tiles of each wave and the first tile's bias row --
// is issued before anything waits; ONE barrier.  The score MFMAs of tile i+1 are issued slot by slot between the
// softmax VALU instructions of tile i (tools/ubench_mfma_rate.hip: on gfx950 a VALU instruction costs ~3 cycles and a
// v_exp ~12 cycles of the SIMD's MFMA issue time even at 3 waves/SIMD, so the gain is only the removed dependency
// stalls; ablations -- no bias loads, no exp, head-fastest block order -- change nothing, no global loads/stores at all
// gives 0.58 ms of the 0.79 ms at stage 0: the kernel is bound by MFMA issue + its own VALU/LDS instruction stream).
template <bool SHIFTED, bool COMPACT>
__global__ __launch_bounds__(192, 3) void window_attn_f32_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ qkv_bias,
                                                              const float* __restrict__ esb,
                                                              float* __restrict__ out, float* __restrict__ lse,
                                                              WinGeom g, int C, int heads, int n_pairs, int shift) {
  // SHIFTED = the instantiation carries the mask code; `shift` (runtime, <= SHIFTED) = the geometry actually rolled.  The two
  // agree in the product; PANGU_ATTN_F32_TPL=1 runs unshifted blocks on the masked instantiation (A/B of the code generation:
  // profiles/r03 notes)
  __shared__ __attribute__((aligned(16))) float Ks[PANGU_WTOK * KV_LD];
  __shared__ __attribute__((aligned(16))) float Vs[PANGU_WTOK * KV_LD];

  // block -> (pair=(t,head), l): blocks b, b+8, .. share an XCD and walk l for one pair
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int pair = (local / g.nLon) * 8 + xcd;
  const int l = local % g.nLon;
  if (pair >= n_pairs) return;
  const int t = pair / heads, hd = pair - t * heads;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int C3 = 3 * C;
  const float scale = 0.17677669529663687f;   // 32^-0.5, reference layers.py:289
  const float* bias_tile = esb + (size_t)(t * heads + hd) * (COMPACT ? PANGU_COMPACT_BIAS : PANGU_WTOK * PANGU_WTOK);

  // ---- up-front loads.  Q fragment: lane group lg owns dims {4lg..4lg+3} and {16+4lg..16+4lg+3} (the MFMA k index is
  // permuted the same way for K: chunk pairs (lg, lg^1) inside every ds_read_b128 lane group keep the swizzled reads
  // conflict-free)
  int qtok[3];
  f32x4 q0[3], q1[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int qn = (wave + 3 * i) * 16 + lq;
    qtok[i] = win_src_token(g, l, t, qn, shift);
    const float* src = (qtok[i] >= 0 ? qkv + (size_t)qtok[i] * C3 : qkv_bias) + hd * 32 + lg * 4;
    q0[i] = *reinterpret_cast<const f32x4*>(src);
    q1[i] = *reinterpret_cast<const f32x4*>(src + 16);
  }
  f32x4 sA[9], sB[9];
  {
    const BiasRowF b0 = load_bias_row<COMPACT>(bias_tile, wave * 16 + lq, lg);
#pragma unroll
    for (int j = 0; j < 9; ++j) sA[j] = b0.v[j];
  }
  // K and V: 144 rows x 8 float4 each = 6 + 6 per thread, all requested before the first LDS write waits
  {
    f32x4 kv[6], vv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int f = tid + 192 * i, n = f >> 3, c4 = (f & 7) * 4;
      const int tok = win_src_token(g, l, t, n, shift);
      const float* src = tok >= 0 ? qkv + (size_t)tok * C3 : qkv_bias;
      kv[i] = *reinterpret_cast<const f32x4*>(src + C + hd * 32 + c4);
      vv[i] = *reinterpret_cast<const f32x4*>(src + 2 * C + hd * 32 + c4);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int f = tid + 192 * i, n = f >> 3, c4 = (f & 7) * 4;
      *reinterpret_cast<f32x4*>(&Ks[kvoff(n, c4)]) = kv[i];
      *reinterpret_cast<f32x4*>(&Vs[kvoff(n, c4)]) = vv[i];
    }
  }

  bool zcut = false, hcut = false;
  // per-lane key-class bits (bit 4j+r <-> key 16j+4lg+r): kz = key in the upper z-plane, kh = key in rows hi<3
  unsigned long long kz_bits = 0ull, kh_bits = 0ull;
  if (SHIFTED) {
    const int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
    zcut = shift && zwin == g.nZw - 1;
    hcut = shift && hwin == g.nHw - 1;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kn = j * 16 + lg * 4 + r;
        if (kn >= 72) kz_bits |= 1ull << (4 * j + r);
        if (((kn / 12) % 6) < 3) kh_bits |= 1ull << (4 * j + r);
      }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { q0[i] *= scale; q1[i] *= scale; }
  __syncthreads();

  float mx, sum;
  // tile 0 scores; bias row of tile 1 -> sB
  {
    const BiasRowF b1 = load_bias_row<COMPACT>(bias_tile, (wave + 3) * 16 + lq, lg);
    s_phase(sA, opaque(Ks), q0[0], q1[0], lq, lg);
#pragma unroll
    for (int j = 0; j < 9; ++j) sB[j] = b1.v[j];
  }
  // tile 0 softmax  ||  tile 1 scores
  softmax_with_next_scores<SHIFTED>(sA, sB, Ks, q0[1], q1[1], wave * 16 + lq, lq, lg, zcut, hcut, kz_bits, kh_bits, mx, sum);
  {
    const BiasRowF b2 = load_bias_row<COMPACT>(bias_tile, (wave + 6) * 16 + lq, lg);
    pv_phase(sA, opaque(Vs), mx, sum, qtok[0], lq, lg, out, lse, C, heads, hd);
#pragma unroll
    for (int j = 0; j < 9; ++j) sA[j] = b2.v[j];
  }
  // tile 1 softmax  ||  tile 2 scores
  softmax_with_next_scores<SHIFTED>(sB, sA, Ks, q0[2], q1[2], (wave + 3) * 16 + lq, lq, lg, zcut, hcut, kz_bits, kh_bits, mx,
                                    sum);
  pv_phase(sB, opaque(Vs), mx, sum, qtok[1], lq, lg, out, lse, C, heads, hd);
  // tile 2
  mx = softmax_max<SHIFTED>(sA, (wave + 6) * 16 + lq, zcut, hcut, kz_bits, kh_bits);
  sum = softmax_sum(softmax_exp<0, 9>(sA, mx));
  pv_phase(sA, opaque(Vs), mx, sum, qtok[2], lq, lg, out, lse, C, heads, hd);
}

__global__ void window_index_export_kernel(int32_t* out, WinGeom g, int shifted) {
  const int total = g.nLon * g.types * PANGU_WTOK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i % PANGU_WTOK, t = (i / PANGU_WTOK) % g.types, l = i / (PANGU_WTOK * g.types);
    out[i] = win_src_token(g, l, t, n, shifted);
  }
}

__global__ void window_mask_export_kernel(float* out, WinGeom g) {
  const int total = g.types * PANGU_WTOK * PANGU_WTOK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int nj = i % PANGU_WTOK, ni = (i / PANGU_WTOK) % PANGU_WTOK, t = i / (PANGU_WTOK * PANGU_WTOK);
    out[i] = win_mask(g, t, ni, nj);
  }
}

int check_geom(int Z, int H, int W) {
  if (Z <= 0 || H <= 0 || W <= 0) return PANGU_E_SHAPE;
  if (Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  return PANGU_OK;
}

}  // namespace

static int launch_attn_fwd(pangu_stream_t stream, const float* qkv, const float* qkv_bias, const float* esb, float* out,
                           float* lse, int Z, int H, int W, int C, int heads, int shifted, bool compact) {
  if (!qkv || !qkv_bias || !esb || !out) return PANGU_E_NULL;
  if (int e = check_geom(Z, H, W)) return e;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  const WinGeom g = make_geom(Z, H, W);
  const int n_pairs = g.types * heads;
  const int grid = ((n_pairs + 7) / 8) * 8 * g.nLon;
  hipStream_t s = (hipStream_t)stream;
#define PANGU_LAUNCH_ATTN(SH, CP)                                                                                      \
  hipLaunchKernelGGL((window_attn_f32_kernel<SH, CP>), dim3(grid), dim3(192), 0, s, qkv, qkv_bias, esb, out, lse, g, C, \
                     heads, n_pairs, shifted ? 1 : 0)
  if (shifted) {
    if (compact) PANGU_LAUNCH_ATTN(true, true); else PANGU_LAUNCH_ATTN(true, false);
  } else {
    if (compact) PANGU_LAUNCH_ATTN(false, true); else PANGU_LAUNCH_ATTN(false, false);
  }
#undef PANGU_LAUNCH_ATTN
  return pangu_launch_status();
}

extern "C" int pangu_window_attn_fwd(pangu_stream_t stream, const float* qkv, const float* qkv_bias,
                                     const float* esb, float* out, float* lse, int Z, int H, int W, int C,
                                     int heads, int shifted) {
  return launch_attn_fwd(stream, qkv, qkv_bias, esb, out, lse, Z, H, W, C, heads, shifted, false);
}

extern "C" int pangu_window_attn_fwd_compact(pangu_stream_t stream, const float* qkv, const float* qkv_bias,
                                             const float* esb_compact, float* out, float* lse, int Z, int H, int W, int C,
                                             int heads, int shifted) {
  return launch_attn_fwd(stream, qkv, qkv_bias, esb_compact, out, lse, Z, H, W, C, heads, shifted, true);
}

extern "C" int pangu_window_index_export(pangu_stream_t stream, int32_t* out, int Z, int H, int W, int shifted) {
  if (!out) return PANGU_E_NULL;
  if (int e = check_geom(Z, H, W)) return e;
  const WinGeom g = make_geom(Z, H, W);
  hipLaunchKernelGGL(window_index_export_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, g, shifted ? 1 : 0);
  return pangu_launch_status();
}

extern "C" int pangu_window_mask_export(pangu_stream_t stream, float* out, int Z, int H, int W) {
  if (!out) return PANGU_E_NULL;
  if (int e = check_geom(Z, H, W)) return e;
  const WinGeom g = make_geom(Z, H, W);
  hipLaunchKernelGGL(window_mask_export_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, g);
  return pangu_launch_status();
}
