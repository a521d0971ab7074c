// Backward of the Earth-specific window attention, fp32, gfx950 (the single-score-orientation kernel, `v2` below).
//
// One workgroup of 12 waves (9 tile owners + 3 helpers) per (window type t, head); it walks the nLon longitude windows
// that share the bias tile esb[t][head], so the bias gradient d_esb[t][head] = sum_l dS stays in registers and is written
// once: no atomics, no (..,144,144) tensor in HBM.  P is recomputed from the saved log-sum-exp.  The shift mask is
// window-invariant and constant over the 4 tokens a lane holds per tile: one bit per tile, built once before the window loop.
// Zero-pad slots (q/k/v = linear1.bias) send their k/v gradient to dqkv_bias with atomics (2.7 % of slots); every real token
// row of dqkv is written exactly once.  (The round-1 kernel, which computed the scores in both orientations -- 56 instead of 40
// MFMAs per 16x16 tile pair, 2.36 vs 1.93 ms per launch at C = 192 -- was removed in round 4; DESIGN.md keeps its measurements.)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KV_LD = 36;
constexpr int NQ = 9;                      // query tiles = key tiles (16 tokens each)
constexpr int NW = 12;                     // waves per workgroup: 9 tile owners + 3 helpers (three waves on every SIMD)
constexpr int NT = NW * 64;

// streaming accesses (every 128-B head slice of qkv / dO / O is read by exactly one workgroup, every dqkv slice written
// once): with the `nt` hint they do not displace the 83-KB bias tiles the 32 workgroups of an XCD re-read from L2 twice per
// window -- without it half of those re-reads miss (2.0x the algorithmic HBM reads, round-1 PMC tables)
template <bool NTH>
__device__ inline f32x4 ldg4(const float* p) {
  return NTH ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
}
template <bool NTH>
__device__ inline void stg4(float* p, f32x4 v) {
  if (NTH) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NTH>
__device__ inline void stg1(float* p, float v) {
  if (NTH) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ===================================================================================================================
// v2 (round 2): ONE score orientation.  Phase 1: the owner of KEY tile w computes S = Qs K^T and dP = dO V^T with the key
// on the lane for query tiles 0..6 (helper wave 9 + h: query tiles 7, 8 of key tiles 3h..3h+2), so P and dS feed
// dV += P^T dO and dK += dS^T Qs directly; d_esb (this orientation's accumulators: 6 quads per owner, 9 per helper) stays
// in registers over the longitude windows, and dS is written ONCE to a [query][key] fp32 image in LDS.  Phase 2: the owner
// of QUERY tile w computes dQ^T += K^T dS^T from that image.  40 instead of 56 MFMAs per 16x16 tile pair (the round-1
// kernel recomputed S and dP in the transposed orientation for dQ): 3 240 instead of 4 536 v_mfma_f32_16x16x4_f32 per
// window, 264-288 per wave on every SIMD, and the bias tile is read once per window instead of twice.
// LDS (156 KB): K, Qs, dO images (61 KB) + dS image (81 KB); V needs no image (only the owner of a key tile reads it: its
// fragment comes straight from global memory).  A helper hands its partial dK / dV sums over through LDS: the first key
// tile's in a 12-KB region of its own (written at once: frees 16 registers), the other two in the memory of the Qs / dO
// images, which are dead after phase 1 (one extra barrier).
namespace v2 {

constexpr int DS_LD = PANGU_WTOK;          // floats per row of the dS image (2-way conflicts on its 9 + 36 accesses per wave and window: noise)
constexpr int A_SPLIT = 7;                 // owner: query tiles [0, 7); helper: [7, 9) of three key tiles (phase 1: 224 / 192 MFMAs; then the owners' 72 of phase 2)

#ifdef PANGU_ATTN_BWD_STAMP
// Diagnostic build only (tools/ablate_attn_bwd.py f32): per-wave s_memtime sums of the OWNER waves: [0] staging pass,
// [1] phase 1, [2] barriers A + B, [3] dK/dV hand-over + stores, [4] request + phase 2, [5] waves, [6] wait at the top
// barrier, [7] LDS writes of the staging pass + its closing barrier
constexpr int STAMP_WAVES = 12 * 1024;
__device__ unsigned long long g_bwdf_stamp[STAMP_WAVES * 8];
__device__ __forceinline__ unsigned long long bwdf_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define BWDF_STAMP(v) const unsigned long long v = bwdf_stamp()
#else
#define BWDF_STAMP(v)
#endif

template <bool SHIFTED, bool NTH>
__global__ __launch_bounds__(NT) void window_attn_bwd2_f32_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ esb,
    const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ dout,
    float* __restrict__ dqkv, float* __restrict__ dqkv_bias, float* __restrict__ d_esb, WinGeom g, int C, int heads) {
  __shared__ __attribute__((aligned(16))) float Ks[PANGU_WTOK * KV_LD];
  __shared__ __attribute__((aligned(16))) float QGs[2 * PANGU_WTOK * KV_LD];     // Qs (scaled), dO; after phase 1: partial sums of the helpers' key tiles 1, 2
  __shared__ __attribute__((aligned(16))) float dSs[PANGU_WTOK * DS_LD];
  __shared__ __attribute__((aligned(16))) f32x4 part0_s[3 * 4 * 64];             // partial sums of the helpers' key tile 0
  __shared__ __attribute__((aligned(16))) float lse_s[PANGU_WTOK];
  __shared__ __attribute__((aligned(16))) float del_s[PANGU_WTOK];
  __shared__ int tok_s[PANGU_WTOK];
  __shared__ float pad_s[64];            // [2][32]: dK, dV summed over the zero-pad keys of this (type, head)
  float* Qs = QGs;
  float* Gs = QGs + PANGU_WTOK * KV_LD;
  f32x4* part12_s = reinterpret_cast<f32x4*>(QGs);                               // [helper][key tile 1, 2][4][64 lanes]

  const int pair = blockIdx.x;
  const int t = pair / heads, hd = pair - t * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lq = lane & 15, lg = lane >> 4;     // laundered once per window (below): the per-tile LDS / bias addresses derived from
                                          // them must not be hoisted out of the window loop (one register per address at the 168 cap)
  const int C3 = 3 * C;
  const float scale = 0.17677669529663687f;
  constexpr float K_LOG2E = 1.4426950408889634f;
  const float* bias_tile = esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK;

  bool zcut = false, hcut = false;
  if (SHIFTED) {
    const int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
    zcut = zwin == g.nZw - 1;
    hcut = hwin == g.nHw - 1;
  }
  auto masked = [&](int nq, int nk) -> bool {
    const bool zd = (nq >= 72) != (nk >= 72);
    const bool hdiff = (((nq / 12) % 6) < 3) != (((nk / 12) % 6) < 3);
    return (zcut && zd) || (hcut && hdiff);
  };

  if (tid < 64) pad_s[tid] = 0.f;
  const bool owner = wave < NQ;
  const int kt0 = owner ? wave : 3 * (wave - NQ);
  // mask bit i (owner) / (9 - A_SPLIT) kk + (i - A_SPLIT) (helper): query tile i against this lane's key of task kk; window-invariant and the
  // same for the 4 queries 16i + 4lg + r of a lane (the cuts fall on multiples of 12 and at 72)
  unsigned mbits = 0u;
  if (SHIFTED) {
    if (zcut || hcut) {
      if (owner) {
        for (int i = 0; i < A_SPLIT; ++i)
          if (masked(i * 16 + lg * 4, kt0 * 16 + lq)) mbits |= 1u << i;
      } else {
        for (int kk = 0; kk < 3; ++kk)
          for (int i = A_SPLIT; i < NQ; ++i)
            if (masked(i * 16 + lg * 4, (kt0 + kk) * 16 + lq)) mbits |= 1u << ((NQ - A_SPLIT) * kk + i - A_SPLIT);
      }
    }
  }

  const float* bias_l = bias_tile;
#ifdef PANGU_ATTN_BWD_STAMP
  unsigned long long sub_st[2] = {0ull, 0ull};
#endif
  // ---- staging pass of window l (all 768 threads): Qs (scaled), K, dO images, delta = rowsum(dO o O), lse; the V
  // fragments of this wave's NK key tiles come straight from global memory (in flight during the pass)
  // ---- staging of window l in two halves.  request(): ALL its global loads (1 152 16-B chunk slots = 144 tokens x 8
  // chunks: ONE per owner thread, THREE per helper thread -- the helpers have the registers; the lse values; the V
  // fragments of this wave's NK key tiles, which need no LDS image) into registers -- issued BEFORE phase 2 of the previous
  // window (owners; the helpers are idle by then), where the register file has room (dK / dV accumulators and K / V
  // fragments are dead), so the ~4.5 us the gathers take run under phase 2 and the barrier waits.
  // stage(): images, delta = rowsum(dO o O) and lse to LDS.
  f32x4 pq[3], pk_[3], pg[3], po[3];
  float plv[3];
  auto slot_of = [&](int u) { return owner ? tid : NQ * 64 + (tid - NQ * 64) + u * 192; };
  auto request = [&](int l, auto& vfr, auto nk_tag) {
    constexpr int NK = decltype(nk_tag)::value;      // key tiles of this wave = chunk slots of this thread: 1 (owner) or 3 (helper)
    // the slot -> (z, h, w) split of win_src_token is window-invariant: left to itself the compiler hoists it out of the window
    // loop for every slot and keeps the pieces live across it -- in the SHIFTED instantiation (wrap-around terms) 17 registers
    // over the 168 a 12-wave workgroup has, i.e. 13 scratch reloads per window (round 3).  Opaque slot numbers make it
    // recompute them here, where the VALU is idle.
    int lq_o = lq;
    asm volatile("" : "+v"(lq_o));
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int tok = win_src_token(g, l, t, (kt0 + kk) * 16 + lq_o, SHIFTED);
      const float* src = (tok >= 0 ? qkv + (size_t)tok * C3 : qkv_bias) + 2 * C + hd * 32 + lg * 8;
      vfr[kk][0] = ldg4<NTH>(src);
      vfr[kk][1] = ldg4<NTH>(src + 4);
    }
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      int f = slot_of(u);
      asm volatile("" : "+v"(f));
      const int n = f >> 3, c4 = (f & 7) * 4;
      const int tok = win_src_token(g, l, t, n, SHIFTED);
      const float* src = tok >= 0 ? qkv + (size_t)tok * C3 : qkv_bias;
      pq[u] = ldg4<NTH>(src + hd * 32 + c4);
      pk_[u] = ldg4<NTH>(src + C + hd * 32 + c4);
      // pad rows: any valid address (the values are zeroed in stage())
      const size_t go = (size_t)(tok >= 0 ? tok : 0) * C + hd * 32 + c4;
      pg[u] = ldg4<NTH>(dout + go);
      po[u] = ldg4<NTH>(out + go);
      plv[u] = lse[(size_t)(tok >= 0 ? tok : 0) * heads + hd];
    }
  };
  auto stage = [&](int l, auto nk_tag) {
    constexpr int NK = decltype(nk_tag)::value;
    BWDF_STAMP(u0);
    __syncthreads();                               // previous window's LDS reads (images, dS, partial sums) are done
    BWDF_STAMP(u1);
    // the bias tile is the same for every window: stop the compiler from hoisting its loads out of the window loop
    long lz = 0;
    asm volatile("" : "+s"(lz));
    bias_l = bias_tile + lz;
    asm volatile("" : "+v"(lq), "+v"(lg));
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      int f = slot_of(u);
      asm volatile("" : "+v"(f));
      const int n = f >> 3, c4 = (f & 7) * 4;
      const int tok = win_src_token(g, l, t, n, SHIFTED);      // recomputed (the VALU is idle here) rather than kept across phase 2
      const bool real = tok >= 0;
      const f32x4 g4 = real ? pg[u] : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 o4 = real ? po[u] : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&Qs[n * KV_LD + c4]) = pq[u] * scale;
      *reinterpret_cast<f32x4*>(&Ks[n * KV_LD + c4]) = pk_[u];
      *reinterpret_cast<f32x4*>(&Gs[n * KV_LD + c4]) = g4;
      float d = (g4[0] * o4[0] + g4[1] * o4[1]) + (g4[2] * o4[2] + g4[3] * o4[3]);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      if ((f & 7) == 0) {
        // p = exp2(S*log2e - lse*log2e), and -delta as the INITIAL ACCUMULATOR of the dP product (dS = p * (dP - delta)).
        // A pad query's row of P must vanish (its output is discarded): -huge makes exp2(..) = 0
        del_s[n] = -d;
        lse_s[n] = real ? -K_LOG2E * plv[u] : -1e30f;
        tok_s[n] = tok;
      }
    }
    __syncthreads();
#ifdef PANGU_ATTN_BWD_STAMP
    { const unsigned long long u2 = bwdf_stamp(); sub_st[0] += u1 - u0; sub_st[1] += u2 - u1; }
#endif
  };
  // ---- one 16x16 score tile: query tile i against key tile kt (fragments k0, k1, v0, v1); lane: [query 16i + 4lg + r][key kn]
  int& lq_w = lq;
  int& lg_w = lg;
  // `bv` holds this tile's four bias values on entry and the NEXT tile's (key tile ktn, query tile in) on exit: their L2
  // round trip runs under this tile's 32 MFMAs instead of in front of the exp
  auto score_tile = [&](int kt, int i, int ktn, int in, const f32x4& k0, const f32x4& k1, const f32x4& v0, const f32x4& v1,
                        bool msk, f32x4& bv, f32x4& db, f32x4& dv0, f32x4& dv1, f32x4& dk0, f32x4& dk1) {
    // per-tile copies of the lane ids behind an opaque asm: the tile's address arithmetic stays inside the tile (the
    // compiler otherwise computes the bias / LDS addresses of all unrolled tiles up front: 48+ registers)
    int lq = lq_w, lg = lg_w;
    asm volatile("" : "+v"(lq), "+v"(lg));
    const int kn = kt * 16 + lq;
    const int qrow = (i * 16 + lq) * KV_LD + lg * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(&Qs[qrow]);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(&Qs[qrow + 4]);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Gs[qrow]);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Gs[qrow + 4]);
    const f32x4 ls = *reinterpret_cast<const f32x4*>(&lse_s[i * 16 + lg * 4]);      // -lse*log2e
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = *reinterpret_cast<const f32x4*>(&del_s[i * 16 + lg * 4]);      // -delta
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], k0[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[ks], v0[ks], dp, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], k1[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[ks], v1[ks], dp, 0, 0, 0);
    }
    f32x4 p, ds, bvn;
#pragma unroll
    for (int r = 0; r < 4; ++r) bvn[r] = bias_l[(size_t)(in * 16 + lg * 4 + r) * PANGU_WTOK + ktn * 16 + lq];
    const float cm = msk ? -100.0f * K_LOG2E : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = __builtin_amdgcn_exp2f(fmaf(s[r] + bv[r], K_LOG2E, ls[r] + cm));
      ds[r] = p[r] * dp[r];
      dSs[(i * 16 + lg * 4 + r) * DS_LD + kn] = ds[r];
    }
    db += ds;
    // dV[key][d] += P[query][key] dO[query][d];  dK[key][d] += dS[query][key] Qs[query][d]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qn = i * 16 + lg * 4 + r;
      dv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], Gs[qn * KV_LD + lq], dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], Gs[qn * KV_LD + 16 + lq], dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], Qs[qn * KV_LD + lq], dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], Qs[qn * KV_LD + 16 + lq], dk1, 0, 0, 0);
    }
    bv = bvn;
  };
  float* dbase = d_esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK + (size_t)(lg * 4) * PANGU_WTOK + lq;

  if (owner) {
    f32x4 db[A_SPLIT];
#pragma unroll
    for (int j = 0; j < A_SPLIT; ++j) db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv;                                               // bias values of the next tile to run (window-invariant)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias_tile[(size_t)(lg * 4 + r) * PANGU_WTOK + wave * 16 + lq];
#ifdef PANGU_ATTN_BWD_STAMP
    unsigned long long acc_st[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    const __amdgpu_buffer_rsrc_t dq_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(dqkv, 0, (int)((size_t)g.Z * g.H * g.W * C3 * sizeof(float)), 0x00020000);
    f32x4 vfn[1][2];                                        // V fragment of the NEXT window (requested before phase 2)
    request(0, vfn, std::integral_constant<int, 1>{});
    for (int l = 0; l < g.nLon; ++l) {
      BWDF_STAMP(t0);
      stage(l, std::integral_constant<int, 1>{});
      const f32x4 vfr[1][2] = {{vfn[0][0], vfn[0][1]}};
      BWDF_STAMP(t1);
      const int kn = wave * 16 + lq;                        // this lane's key column (phase 1) / query row (phase 2)
      // =========================== phase 1: key tile `wave`, query tiles 0..6 ===========================
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(&Ks[kn * KV_LD + lg * 8]);
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(&Ks[kn * KV_LD + lg * 8 + 4]);
      f32x4 dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
#pragma unroll
      for (int i = 0; i < A_SPLIT; ++i) {
        __builtin_amdgcn_sched_barrier(0);      // keep each tile's loads inside its iteration (VGPR cap 168)
        score_tile(wave, i, wave, (i + 1) % A_SPLIT, k0, k1, vfr[0][0], vfr[0][1], SHIFTED && ((mbits >> i) & 1u), bv, db[i],
                   dv0, dv1, dk0, dk1);
      }
      BWDF_STAMP(t2);
      __syncthreads();                             // A: the dS image is complete; the Qs / dO images are dead
      __syncthreads();                             // B: the helpers' partial sums are in LDS (they write them between A and B)
      BWDF_STAMP(t3);
      {
        const int h = wave / 3, kk = wave - 3 * h;
        const f32x4* src = kk == 0 ? part0_s + h * 4 * 64 + lane : part12_s + ((h * 2 + kk - 1) * 4) * 64 + lane;
        dv0 += src[0]; dv1 += src[64]; dk0 += src[128]; dk1 += src[192];
        // lane: dK/dV[key = 16*wave + 4lg + r][d = 16dt + lq]
        bool any_pad = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ktok = tok_s[wave * 16 + lg * 4 + r];
          if (ktok >= 0) {
            float* dst = dqkv + (size_t)ktok * C3 + hd * 32 + lq;
            stg1<NTH>(dst + C, dk0[r]);
            stg1<NTH>(dst + C + 16, dk1[r]);
            stg1<NTH>(dst + 2 * C, dv0[r]);
            stg1<NTH>(dst + 2 * C + 16, dv1[r]);
          } else {
            any_pad = true;
          }
        }
        // zero-pad keys all carry linear1.bias: sum their gradients over the keys this lane holds, then over the four
        // key groups, then in LDS; ONE global atomic per value at the end (instead of 64 per pad key and window)
        if (__any(any_pad)) {
          float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (tok_s[wave * 16 + lg * 4 + r] < 0) { a0 += dk0[r]; a1 += dk1[r]; b0 += dv0[r]; b1 += dv1[r]; }
          a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64);
          a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); b0 += __shfl_xor(b0, 32, 64); b1 += __shfl_xor(b1, 32, 64);
          if (lg == 0) {
            atomicAdd(&pad_s[lq], a0);
            atomicAdd(&pad_s[16 + lq], a1);
            atomicAdd(&pad_s[32 + lq], b0);
            atomicAdd(&pad_s[48 + lq], b1);
          }
        }
      }
          BWDF_STAMP(t4);
      request(l + 1 < g.nLon ? l + 1 : l, vfn, std::integral_constant<int, 1>{});     // the last one is redundant
      // =========================== phase 2: query tile `wave`: dQ^T[d][query] += K^T[d][key] dS^T[key][query] =========
      {
        f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          __builtin_amdgcn_sched_barrier(0);      // keep each key tile's LDS reads next to their MFMAs (register cap)
          // k index lg of step s <-> key 16j + 4lg + s on both operands
          const f32x4 dsq = *reinterpret_cast<const f32x4*>(&dSs[kn * DS_LD + j * 16 + lg * 4]);
#pragma unroll
          for (int sidx = 0; sidx < 4; ++sidx) {
            const int key = j * 16 + lg * 4 + sidx;
            dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[key * KV_LD + lq], dsq[sidx], dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[key * KV_LD + 16 + lq], dsq[sidx], dq1, 0, 0, 0);
          }
        }
        // lane: dQ^T[d = 16dt + 4lg + r][query = kn]; q was pre-scaled, so dq = scale * dQs
        // unconditional (a pad query's store carries an out-of-range offset and is dropped): a branch around the stores
        // would turn the wait for the prefetched loads at the next stage() into vmcnt(0), i.e. into a wait for these
        // stores' acknowledgements as well
        const int qtok = tok_s[kn];
        const unsigned off = qtok >= 0 ? ((unsigned)qtok * (unsigned)C3 + (unsigned)(hd * 32 + lg * 4)) * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dq0 * scale), dq_rsrc, (int)off, 0, NTH ? 2 : 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dq1 * scale), dq_rsrc, (int)off, 64, NTH ? 2 : 0);
      }
#ifdef PANGU_ATTN_BWD_STAMP
      {
        const unsigned long long t5 = bwdf_stamp();
        acc_st[0] += t1 - t0; acc_st[1] += t2 - t1; acc_st[2] += t3 - t2; acc_st[3] += t4 - t3; acc_st[4] += t5 - t4;
      }
#endif
    }
#ifdef PANGU_ATTN_BWD_STAMP
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long* d = g_bwdf_stamp + (size_t)(blockIdx.x * 12 + wave) * 8;
      for (int k = 0; k < 5; ++k) d[k] = acc_st[k];
      d[5] = 1ull; d[6] = sub_st[0]; d[7] = sub_st[1];
    }
#endif
    // bias gradient: lane holds sum_l dS[query = 16i + 4lg + r][key = 16 wave + lq]
#pragma unroll
    for (int i = 0; i < A_SPLIT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) dbase[(i * 16 + r) * PANGU_WTOK + wave * 16] = db[i][r];
  } else {
    // ============================================ helpers: query tiles 7, 8 of key tiles 3h .. 3h+2 =====================
    const int h = wave - NQ;
    constexpr int HT = NQ - A_SPLIT;               // query tiles per key tile
    f32x4 db[3 * HT];
#pragma unroll
    for (int j = 0; j < 3 * HT; ++j) db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias_tile[(size_t)(A_SPLIT * 16 + lg * 4 + r) * PANGU_WTOK + kt0 * 16 + lq];
    f32x4 vfr[3][2];
    request(0, vfr, std::integral_constant<int, 3>{});
    for (int l = 0; l < g.nLon; ++l) {
      stage(l, std::integral_constant<int, 3>{});
      f32x4 pk[2][4];                              // partial sums of key tiles 1, 2 (key tile 0's go to LDS at once)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int kt = kt0 + kk;
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(&Ks[(kt * 16 + lq) * KV_LD + lg * 8]);
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(&Ks[(kt * 16 + lq) * KV_LD + lg * 8 + 4]);
        f32x4 dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
#pragma unroll
        for (int ii = 0; ii < NQ - A_SPLIT; ++ii) {
          __builtin_amdgcn_sched_barrier(0);
          // next tile in this wave's order: (kk, ii + 1), then (kk + 1, 0), then the next window's (0, 0)
          const int kkn = ii + 1 < NQ - A_SPLIT ? kk : (kk + 1) % 3, iin = ii + 1 < NQ - A_SPLIT ? ii + 1 : 0;
          score_tile(kt, A_SPLIT + ii, kt0 + kkn, A_SPLIT + iin, k0, k1, vfr[kk][0], vfr[kk][1],
                     SHIFTED && ((mbits >> (HT * kk + ii)) & 1u), bv, db[HT * kk + ii], dv0, dv1, dk0, dk1);
        }
        if (kk == 0) {
          f32x4* dst = part0_s + h * 4 * 64 + lane;
          dst[0] = dv0; dst[64] = dv1; dst[128] = dk0; dst[192] = dk1;
        } else {
          pk[kk - 1][0] = dv0; pk[kk - 1][1] = dv1; pk[kk - 1][2] = dk0; pk[kk - 1][3] = dk1;
        }
      }
      __syncthreads();                             // A: the Qs / dO images are dead
#pragma unroll
      for (int kk = 1; kk < 3; ++kk) {
        f32x4* dst = part12_s + ((h * 2 + kk - 1) * 4) * 64 + lane;
        dst[0] = pk[kk - 1][0]; dst[64] = pk[kk - 1][1]; dst[128] = pk[kk - 1][2]; dst[192] = pk[kk - 1][3];
      }
      __syncthreads();                             // B
      request(l + 1 < g.nLon ? l + 1 : l, vfr, std::integral_constant<int, 3>{});     // the helpers are idle from here on
    }
#ifdef PANGU_ATTN_BWD_STAMP
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long* d = g_bwdf_stamp + (size_t)(blockIdx.x * 12 + wave) * 8;
      for (int k = 0; k < 8; ++k) d[k] = 0ull;
    }
#endif
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int ii = 0; ii < NQ - A_SPLIT; ++ii)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dbase[((A_SPLIT + ii) * 16 + r) * PANGU_WTOK + (kt0 + kk) * 16] = db[(NQ - A_SPLIT) * kk + ii][r];
  }
  __syncthreads();
  if (tid < 64 && pad_s[tid] != 0.f) atomicAdd(dqkv_bias + (tid < 32 ? C : 2 * C) + hd * 32 + (tid & 31), pad_s[tid]);
}

}  // namespace v2

}  // namespace

#ifdef PANGU_ATTN_BWD_STAMP
extern "C" int pangu_attn_bwdf_stamp_read(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  static unsigned long long host[v2::STAMP_WAVES * 8];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(v2::g_bwdf_stamp), sizeof(host));
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (int w = 0; w < v2::STAMP_WAVES; ++w)
    for (int k = 0; k < 8; ++k) out8[k] += host[(size_t)w * 8 + k];
  for (size_t i = 0; i < (size_t)v2::STAMP_WAVES * 8; ++i) host[i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(v2::g_bwdf_stamp), host, sizeof(host));
  return 0;
}
#endif

extern "C" int pangu_window_attn_bwd(pangu_stream_t stream, const float* qkv, const float* qkv_bias, const float* esb,
                                     const float* out, const float* lse, const float* dout, float* dqkv,
                                     float* dqkv_bias, float* d_esb, int Z, int H, int W, int C, int heads,
                                     int shifted) {
  if (!qkv || !qkv_bias || !esb || !out || !lse || !dout || !dqkv || !dqkv_bias || !d_esb) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  // 32-bit buffer offsets; the pad-row sentinel 0x80000000 must stay OUTSIDE the (n_tok x 3C) buffers
  if ((size_t)Z * H * W * 3 * C * sizeof(float) >= 0x7FFFFFF0ull) return PANGU_E_RANGE;
  const WinGeom g = make_geom(Z, H, W);
  const int n_pairs = g.types * heads;
  hipStream_t s = (hipStream_t)stream;
#define PANGU_LAUNCH_BWD(SH)                                                                                            \
  hipLaunchKernelGGL((v2::window_attn_bwd2_f32_kernel<SH, true>), dim3(n_pairs), dim3(NT), 0, s, qkv, qkv_bias, esb, out, lse, \
                     dout, dqkv, dqkv_bias, d_esb, g, C, heads)
  if (shifted) PANGU_LAUNCH_BWD(true); else PANGU_LAUNCH_BWD(false);
#undef PANGU_LAUNCH_BWD
  return pangu_launch_status();
}
