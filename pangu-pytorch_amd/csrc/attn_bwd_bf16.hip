// Backward of the Earth-specific window attention, bf16 operands / fp32 softmax + accumulation, gfx950.
//
// Same decomposition as attn_bwd_f32.hip: one 9-wave workgroup per (window type, head) walks the longitude windows and
// keeps d_esb[t][head] = sum_l dS in registers.  A 16x16 score tile is ONE v_mfma_f32_16x16x32_bf16 (K = head_dim = 32).  The
// kernel below computes the scores in ONE orientation (see its header comment); the round-1 kernel (both orientations, transposed
// token-major LDS copies, 66 instead of 48 MFMAs per wave and window: 0.86 vs 0.60 ms at C = 192) was removed in round 4.
#include "common.h"
#include <stdlib.h>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int NW = 9, NT = NW * 64;
constexpr int ROWIMG = PANGU_WTOK * 64;        // bytes of a row-major [144][32] bf16 image

__device__ inline u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ inline unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }

#ifndef PANGU_ATTN_BWD_WIDE
#define PANGU_ATTN_BWD_WIDE 1      // 16-B gradient stores after a v_permlane16_swap exchange (0: two 8-B stores per row and lane)
#endif
// x0 / x1: this lane's four values of d = 4lg.. / 16 + 4lg.. of one token -> the 16-B piece d = {0, 16, 8, 24}[lg] .. +7 of that token
// (all 64 lanes must be active: the swap exchanges whole 16-lane rows)
__device__ inline u32x4 wide16(const f32x4 x0, const f32x4 x1) {
  const auto r0 = __builtin_amdgcn_permlane16_swap(pack2(x0[0], x0[1]), pack2(x1[0], x1[1]), false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(pack2(x0[2], x0[3]), pack2(x1[2], x1[3]), false, false);
  return u32x4{r0[0], r1[0], r0[1], r1[1]};
}
__device__ inline float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ inline float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
__device__ inline float bf1(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ inline int kswz(int row, int chunk) {
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((chunk ^ f) << 4);
}
__device__ inline bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return __builtin_bit_cast(bf16x8, u32x4{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])});
}
// ===================================================================================================================
// v2 (round 2): ONE score orientation.  Wave w owns KEY tile w in phase 1: S = Q K^T and dP = dO V^T with the key on
// the lane, so P and dS (two adjacent query tiles' accumulator quads = one 8-token operand) feed dV^T += dO^T P and
// dK^T += Q^T dS directly; dS (bf16) crosses LDS ONCE as a [key][query] image for phase 2, where wave w owns QUERY tile
// w: dQ^T += K^T dS^T.  The operands whose k index runs over tokens (dO^T, Q^T, K^T, dS^T) are read from the ROW-major
// images with ds_read_b64_tr_b16 (hardware transpose): no transposed copies, no 2-byte scatter writes in the staging
// pass (v1: 24 ds_write_b16 per thread and window, 44 % bank-conflict cycles), half the exp / softmax VALU work and
// 48 instead of 66 MFMAs per wave and window.  d_esb stays in registers over the longitude windows (key tile w x all
// query tiles).  The next window's q/k/v/dO/O rows are requested before the compute phases and written to LDS after
// them (register prefetch): the HBM latency of the staging pass runs under the MFMAs.  Bias tile transposed in LDS
// ([key][query]: one 8-byte read per score tile).
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int DS_LD = 288;                       // bytes per row of the [key][query] bf16 images (dS, bias^T)
constexpr int DSIMG = PANGU_WTOK * DS_LD;
constexpr int DB2_LDS = 4, DB2_REG = 9 - DB2_LDS;
// LDS map (byte offsets from the start of the dynamic region).  Every access below is (one lane-constant base
// register) + (compile-time immediate < 64 KB): the bases are made opaque to the compiler, which otherwise keeps one
// precomputed address register per (image, tile) alive across the window loop -- 40 registers at the 168-VGPR cap
constexpr int L_ROW = 0;                         // lse_s, del_s (float[144] each), tok_s (int[144]), pad_s (float[64])
constexpr int L_IMG = 2048;                      // Kr, Vr, Qr, Gr: row-major [144][32] bf16, 16-B chunks swizzled
constexpr int L_DS = L_IMG + 4 * ROWIMG;         // dS of the current window, [key][query] bf16
constexpr int L_BT = L_DS + DSIMG;               // bias tile transposed, [key][query] bf16
constexpr int L_DB = L_BT + DSIMG;               // d_esb quads that do not fit the register file (lane-private)
constexpr int L_END = L_DB + DB2_LDS * NT * 16;
constexpr int I_K = 0, I_V = ROWIMG, I_Q = 2 * ROWIMG, I_G = 3 * ROWIMG;

template <typename T>
__device__ inline __attribute__((address_space(3))) T* ldsp(unsigned off) {
  return (__attribute__((address_space(3))) T*)(size_t)(off);
}
__device__ inline s16x4 tr16(unsigned off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(ldsp<s16x4>(off));
}
__device__ inline bf16x8 cat8(s16x4 a, s16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ inline unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

#ifdef PANGU_ATTN_BWD_STAMP
// Diagnostic build only (tools/ablate_attn_bwd.py): per-wave s_memtime sums: [0] staging (loop top .. second barrier),
// [1] phase 1, [2] wait at the dS barrier, [3] phase 2, [4] whole kernel, [5] waves, [6] prologue
constexpr int STAMP_WAVES = 9 * 1024;
__device__ unsigned long long g_bwd_stamp[STAMP_WAVES * 8];
__device__ __forceinline__ unsigned long long bwd_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define BWD_STAMP(v) const unsigned long long v = bwd_stamp()
#else
#define BWD_STAMP(v)
#endif

template <bool SHIFTED>
__global__ __launch_bounds__(NT) void window_attn_bwd2_bf16_kernel(
    const u16* __restrict__ qkv, const u16* __restrict__ qkv_bias, const u16* __restrict__ esb,
    const u16* __restrict__ out, const float* __restrict__ lse, const u16* __restrict__ dout, u16* __restrict__ dqkv,
    float* __restrict__ dqkv_bias, float* __restrict__ d_esb, WinGeom g, int C, int heads) {
  BWD_STAMP(st_begin);
#ifdef PANGU_ATTN_BWD_STAMP
  unsigned long long acc_st[4] = {0ull, 0ull, 0ull, 0ull};
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned L0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  int pair = blockIdx.x;
  if (!(heads & 1)) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    pair = 2 * ((local >> 1) * 8 + xcd) + (local & 1);
  }
  if (pair >= g.types * heads) return;
  const int t = pair / heads, hd = pair - t * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int wide_doff = ((lg & 1) << 4) | ((lg >> 1) << 3);      // first d of this lane's 16-B piece after wide16()
  const int C3 = 3 * C;
  const float scale = 0.17677669529663687f;
  constexpr float K_LOG2E = 1.4426950408889634f;
  const float scale2 = scale * K_LOG2E;
  constexpr float K_MASK2 = -100.0f * K_LOG2E;
  const u16* bias_tile = esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK;

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
  // the mask is window-invariant and the same for the 4 queries 16i + 4lg + r of a lane (the cuts fall on multiples of
  // 12 and at 72): one bit per query tile against this lane's key, built once
  const int kn = wave * 16 + lq;
  unsigned mbits = 0u;
  if (SHIFTED) {
    if (zcut || hcut)
      for (int i = 0; i < 9; ++i)
        if (masked(i * 16 + lg * 4, kn)) mbits |= 1u << i;
  }

  if (tid < 64) *ldsp<float>(L0 + L_ROW + 1728 + tid * 4) = 0.f;
  // bias tile -> LDS, transposed: 16-B global reads along the key axis, 2-byte LDS writes (once per workgroup)
  for (int i = tid; i < PANGU_WTOK * (PANGU_WTOK / 8); i += NT) {
    const int qn = i / (PANGU_WTOK / 8), c = i - qn * (PANGU_WTOK / 8);
    const u32x4 b = *reinterpret_cast<const u32x4*>(bias_tile + (size_t)qn * PANGU_WTOK + c * 8);
    const unsigned o = L0 + L_BT + c * 8 * DS_LD + ((((qn >> 2) ^ (2 * (c & 1))) << 3) | ((qn & 3) << 1));   // swizzled piece, see b_kqbt
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *ldsp<u16>(o + (2 * e) * DS_LD) = (u16)(b[e] & 0xFFFFu);
      *ldsp<u16>(o + (2 * e + 1) * DS_LD) = (u16)(b[e] >> 16);
    }
  }

  // ---- lane-constant LDS bases
  const int tq = lq >> 2, tp = lq & 3;      // transposed reads: this lane supplies row tq of its group's block, 4 columns at 4 tp
  // d 0..15 / 16..31 of a row image, rows tb + 4lg + tq (tb a multiple of 16: the swizzle term does not see it)
  const unsigned b_trlo = opaque(L0 + L_IMG + kswz(4 * lg + tq, tp >> 1) + 8 * (tp & 1));
  const unsigned b_trhi = opaque(L0 + L_IMG + kswz(4 * lg + tq, 2 + (tp >> 1)) + 8 * (tp & 1));
  // The [key][query] images have 288-byte rows (72 dwords = 8 mod 32 banks): the 16 lanes of a ds_write_b64 group (16 keys, one
  // 8-byte piece each) landed on 4 bank pairs (4-way: 108 of ~500 LDS-array cycles per wave and window, round-4 bank model
  // tools/lds_banks.py), the 32 lanes of a bias^T ds_read_b64 on 16 (2-way).  Swizzle of the 8-byte piece index inside a row:
  // dS pieces ^ ((key >> 2) & 3), bias^T pieces ^ 2 ((key >> 3) & 1): writes, row reads and transposed reads all conflict-free.
  const unsigned b_trds = opaque(L0 + L_DS + (4 * lg + tq) * DS_LD + wave * 32 + 8 * (tp ^ (lg & 3)));   // dS: key rows, this wave's queries
  const unsigned b_kqds = opaque(L0 + L_DS + kn * DS_LD + 8 * (lg ^ ((lq >> 2) & 3)));     // [key kn][query 16i + 4lg ..] (+ 32 i)
  const unsigned b_kqbt = opaque(L0 + L_BT + kn * DS_LD + 8 * (lg ^ (2 * (lq >> 3))));
  const unsigned b_row = opaque(L0 + L_IMG + kswz(lq, lg));            // row read of token 16x + lq, chunk lg (+ 1024 x)
  const unsigned b_ls = opaque(L0 + L_ROW + lg * 16);                  // lse_s / del_s quads of query tile i (+ 64 i)
  const unsigned b_db = opaque(L0 + L_DB + tid * 16);

  f32x4 dbias[DB2_REG];
#pragma unroll
  for (int j = 0; j < DB2_REG; ++j) dbias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < DB2_LDS; ++j) *ldsp<f32x4>(b_db + j * NT * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- register prefetch of the staging pass: one 16-B chunk (8 dims) of q, k, v, dO, O per thread.
  // Every global access of the window loop is UNCONDITIONAL (pad rows: buffer loads / stores with an out-of-range offset
  // return zero / are dropped), and a window's loads are issued and consumed in the same loop iteration: the compiler can
  // then wait for them with a COUNTED vmcnt that leaves the iteration's six gradient stores in flight (a branch around
  // any of them forces vmcnt(0): the staging pass then also waits for the store acknowledgements, ~2k cycles per window)
  const int n_tok = g.Z * g.H * g.W;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t dq_rsrc = __builtin_amdgcn_make_buffer_rsrc(dqkv, 0, n_tok * C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(dout), 0, n_tok * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(out), 0, n_tok * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lse), 0, n_tok * heads * 4, 0x00020000);
  const int sn = tid >> 2, sch = tid & 3;
  const unsigned b_st = opaque(L0 + L_IMG + kswz(sn, sch));
  const unsigned b_sr = opaque(L0 + L_ROW + sn * 4);
  u32x4 qv, kv, vv, gv, ov;
  int ptok;
  float plse;
  const unsigned ho = hd * 32 + sch * 8;
  auto request = [&](int l) {
    ptok = win_src_token(g, l, t, sn, SHIFTED);
    const u16* src = ptok >= 0 ? qkv : qkv_bias;
    const unsigned so = (ptok >= 0 ? (unsigned)ptok * (unsigned)C3 : 0u) + ho;     // 32-bit element offsets
    qv = *reinterpret_cast<const u32x4*>(src + so);
    kv = *reinterpret_cast<const u32x4*>(src + so + (unsigned)C);
    vv = *reinterpret_cast<const u32x4*>(src + so + 2u * (unsigned)C);
    const unsigned go = ptok >= 0 ? ((unsigned)ptok * (unsigned)C + ho) * 2u : OOB;
    gv = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (int)go, 0, 0);
    ov = __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, (int)go, 0, 0);
    plse = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
        l_rsrc, (int)(ptok >= 0 ? ((unsigned)ptok * (unsigned)heads + (unsigned)hd) * 4u : OOB), 0, 0));
  };
  auto stage = [&]() {
    *ldsp<u32x4>(b_st + I_Q) = qv;
    *ldsp<u32x4>(b_st + I_K) = kv;
    *ldsp<u32x4>(b_st + I_V) = vv;
    *ldsp<u32x4>(b_st + I_G) = gv;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) d += bflo(gv[e]) * bflo(ov[e]) + bfhi(gv[e]) * bfhi(ov[e]);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    if (sch == 0) {
      // row constants as INITIAL ACCUMULATORS: S' = q.k - lse/scale (p = exp2(S' scale log2e + b log2e)), dP - delta.
      // A pad query's row of P must vanish (its output is discarded): -huge makes exp2(..) = 0
      *ldsp<float>(b_sr + 576) = -d;
      *ldsp<float>(b_sr) = ptok >= 0 ? -plse * (1.0f / scale) : -1e30f;
      *ldsp<int>(b_sr + 1152) = ptok;
    }
  };
  request(0);
  BWD_STAMP(st_loop);
  stage();

  for (int l = 0; l < g.nLon; ++l) {
    BWD_STAMP(s0);
    __syncthreads();                              // window l is staged
    request(l + 1 < g.nLon ? l + 1 : l);          // in flight during the two compute phases (the last one is redundant)
    BWD_STAMP(s1);

    // =========================== phase 1: key tile `wave`; S[query][key], key on the lane ===========================
    {
      const bf16x8 kf = *ldsp<bf16x8>(b_row + I_K + wave * 1024);
      const bf16x8 vf = *ldsp<bf16x8>(b_row + I_V + wave * 1024);
      f32x4 dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        // each tile's P / dS quads are packed to bf16 (2 + 2 registers) as soon as they exist; the pair is the 8-token
        // k operand of the second products
        u32x2 ppk[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}}, dsk[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * u + h;
          if (i < 9) {
            const bf16x8 af = *ldsp<bf16x8>(b_row + I_Q + i * 1024);
            const bf16x8 gf = *ldsp<bf16x8>(b_row + I_G + i * 1024);
            const f32x4 s0 = *ldsp<f32x4>(b_ls + i * 64);
            const f32x4 d0 = *ldsp<f32x4>(b_ls + 576 + i * 64);
            const u32x2 bq = *ldsp<u32x2>(b_kqbt + 32 * i);
            const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, kf, s0, 0, 0, 0);     // [query 4lg+r][key lq]
            const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf, d0, 0, 0, 0);    // dP - delta
            const float bb[4] = {bflo(bq[0]), bfhi(bq[0]), bflo(bq[1]), bfhi(bq[1])};
            float cm = 0.f;
            if (SHIFTED) { if ((mbits >> i) & 1u) cm = K_MASK2; }
            f32x4 p, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, fmaf(bb[r], K_LOG2E, cm)));
              ds[r] = p[r] * dp[r];
            }
            if (i < DB2_REG) dbias[i < DB2_REG ? i : 0] += ds;
            else *ldsp<f32x4>(b_db + (i - DB2_REG) * NT * 16) += ds;
            ppk[h] = u32x2{pack2(p[0], p[1]), pack2(p[2], p[3])};
            dsk[h] = u32x2{pack2(ds[0], ds[1]), pack2(ds[2], ds[3])};
            *ldsp<u32x2>(b_kqds + 32 * i) = dsk[h];
          }
        }
        // dV^T[d][key] += dO^T[d][queries] . P ;  dK^T[d][key] += Q^T[d][queries] . dS   (queries of tiles 2u, 2u+1)
        const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{ppk[0][0], ppk[0][1], ppk[1][0], ppk[1][1]});
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsk[0][0], dsk[0][1], dsk[1][0], dsk[1][1]});
        const s16x4 z4 = {0, 0, 0, 0};
        const bool two = u < 4;                   // tile 9 does not exist: its half of the operand is zero
        const bf16x8 g_lo = cat8(tr16(b_trlo + I_G + u * 2048), two ? tr16(b_trlo + I_G + u * 2048 + 1024) : z4);
        const bf16x8 g_hi = cat8(tr16(b_trhi + I_G + u * 2048), two ? tr16(b_trhi + I_G + u * 2048 + 1024) : z4);
        const bf16x8 q_lo = cat8(tr16(b_trlo + I_Q + u * 2048), two ? tr16(b_trlo + I_Q + u * 2048 + 1024) : z4);
        const bf16x8 q_hi = cat8(tr16(b_trhi + I_Q + u * 2048), two ? tr16(b_trhi + I_Q + u * 2048 + 1024) : z4);
        dv0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_lo, pf, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_hi, pf, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_lo, dsf, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_hi, dsf, dk1, 0, 0, 0);
      }
      // lane: dK^T / dV^T [d = 16dt + 4lg + r][key kn]
      const int ktok = *ldsp<int>(L0 + L_ROW + 1152 + kn * 4);
      dk0 *= scale; dk1 *= scale;
#if PANGU_ATTN_BWD_WIDE
      {
        // the token's 64-B head row sits in the wave as 8-B pieces over the four 16-lane rows (x0: d = 4lg.., x1: d = 16 + 4lg..):
        // v_permlane16_swap (odd rows of x0 <-> even rows of x1) leaves EIGHT consecutive d per lane -- rows 0..3: d = 0, 16, 8, 24
        // -- so each gradient row leaves as one 16-B store per lane instead of two 8-B stores (attn_bf16.hip, attn_tile)
        const unsigned dst = ktok >= 0 ? ((unsigned)ktok * (unsigned)C3 + (unsigned)(hd * 32 + wide_doff)) * 2u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(wide16(dk0, dk1), dq_rsrc, (int)dst, 2 * C, 0);
        asm volatile("s_nop 1" ::: "memory");        // store-data WAR hazard hipcc leaves open with an SGPR soffset (mlp_fused_bf16.hip)
        __builtin_amdgcn_raw_buffer_store_b128(wide16(dv0, dv1), dq_rsrc, (int)dst, 4 * C, 0);
        asm volatile("s_nop 1" ::: "memory");
      }
#else
      {
        const unsigned dst = ktok >= 0 ? ((unsigned)ktok * (unsigned)C3 + (unsigned)(hd * 32 + lg * 4)) * 2u : OOB;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(dk0[0], dk0[1]), pack2(dk0[2], dk0[3])}, dq_rsrc, (int)dst, 2 * C, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(dk1[0], dk1[1]), pack2(dk1[2], dk1[3])}, dq_rsrc, (int)dst, 2 * C + 32, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(dv0[0], dv0[1]), pack2(dv0[2], dv0[3])}, dq_rsrc, (int)dst, 4 * C, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(dv1[0], dv1[1]), pack2(dv1[2], dv1[3])}, dq_rsrc, (int)dst, 4 * C + 32, 0);
      }
#endif
      // zero-pad keys all carry linear1.bias: their gradients are summed (lanes of a pad key, then LDS, then ONE
      // global atomic per value at the end) instead of 64 same-address global atomics per pad key and window
      if (__any(ktok < 0)) {
        const float keep = ktok < 0 ? 1.f : 0.f;
        float* pad_s = (float*)(smem + L_ROW + 1728);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a0 = dk0[r] * keep, a1 = dk1[r] * keep, b0 = dv0[r] * keep, b1 = dv1[r] * keep;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64);
            b0 += __shfl_xor(b0, o, 64); b1 += __shfl_xor(b1, o, 64);
          }
          if (lq == 0) {
            atomicAdd(&pad_s[lg * 4 + r], a0);
            atomicAdd(&pad_s[16 + lg * 4 + r], a1);
            atomicAdd(&pad_s[32 + lg * 4 + r], b0);
            atomicAdd(&pad_s[48 + lg * 4 + r], b1);
          }
        }
      }
    }
    BWD_STAMP(s2);
    __syncthreads();                              // the dS image is complete
    BWD_STAMP(s3);

    // =========================== phase 2: query tile `wave`: dQ^T[d][query] += K^T[d][keys] . dS^T[keys][query] ======
    {
      f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
      const s16x4 z4 = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const bool two = u < 4;
        const bf16x8 k_lo = cat8(tr16(b_trlo + I_K + u * 2048), two ? tr16(b_trlo + I_K + u * 2048 + 1024) : z4);
        const bf16x8 k_hi = cat8(tr16(b_trhi + I_K + u * 2048), two ? tr16(b_trhi + I_K + u * 2048 + 1024) : z4);
        const bf16x8 dsf = cat8(tr16(b_trds + u * 32 * DS_LD), two ? tr16(b_trds + u * 32 * DS_LD + 16 * DS_LD) : z4);
        dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_lo, dsf, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, dsf, dq1, 0, 0, 0);
      }
      const int qtok = *ldsp<int>(L0 + L_ROW + 1152 + kn * 4);
#if PANGU_ATTN_BWD_WIDE
      {                       // lane: dQ^T[d = 16dt + 4lg + r][query 16 wave + lq] -> eight consecutive d per lane, one 16-B store
        const unsigned dst = qtok >= 0 ? ((unsigned)qtok * (unsigned)C3 + (unsigned)(hd * 32 + wide_doff)) * 2u : OOB;
        dq0 *= scale; dq1 *= scale;
        __builtin_amdgcn_raw_buffer_store_b128(wide16(dq0, dq1), dq_rsrc, (int)dst, 0, 0);
      }
#else
      {                       // lane: dQ^T[d = 16dt + 4lg + r][query 16 wave + lq]
        const unsigned dst = qtok >= 0 ? ((unsigned)qtok * (unsigned)C3 + (unsigned)(hd * 32 + lg * 4)) * 2u : OOB;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(dq0[0] * scale, dq0[1] * scale), pack2(dq0[2] * scale, dq0[3] * scale)},
                                              dq_rsrc, (int)dst, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(dq1[0] * scale, dq1[1] * scale), pack2(dq1[2] * scale, dq1[3] * scale)},
                                              dq_rsrc, (int)dst, 32, 0);
      }
#endif
    }
    BWD_STAMP(s4);
    __syncthreads();                              // every wave is done with the images and dS of window l
    stage();                                      // window l + 1 (its loads were requested before phase 1)
#ifdef PANGU_ATTN_BWD_STAMP
    {
      const unsigned long long s5 = bwd_stamp();
      acc_st[0] += (s1 - s0) + (s5 - s4); acc_st[1] += s2 - s1; acc_st[2] += s3 - s2; acc_st[3] += s4 - s3;
    }
#endif
  }
#ifdef PANGU_ATTN_BWD_STAMP
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long* d = g_bwd_stamp + (size_t)(blockIdx.x * 9 + wave) * 8;
    d[0] = acc_st[0]; d[1] = acc_st[1]; d[2] = acc_st[2]; d[3] = acc_st[3];
    d[4] = bwd_stamp() - st_begin; d[5] = 1ull; d[6] = st_loop - st_begin;
  }
#endif
  {
    const float pv = tid < 64 ? *ldsp<float>(L0 + L_ROW + 1728 + tid * 4) : 0.f;
    if (tid < 64 && pv != 0.f) atomicAdd(dqkv_bias + (tid < 32 ? C : 2 * C) + hd * 32 + (tid & 31), pv);
  }
  // ---- bias gradient: lane holds sum_l dS[query = 16i + 4lg + r][key = kn]
  float* drow = d_esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK + (size_t)(lg * 4) * PANGU_WTOK + kn;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const f32x4 v = i < DB2_REG ? dbias[i < DB2_REG ? i : 0] : *ldsp<f32x4>(b_db + (i < DB2_REG ? 0 : i - DB2_REG) * NT * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) drow[(i * 16 + r) * PANGU_WTOK] = v[r];
  }
}

}  // namespace

#ifdef PANGU_ATTN_BWD_STAMP
extern "C" int pangu_attn_bwd_stamp_read(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  static unsigned long long host[STAMP_WAVES * 8];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_stamp), sizeof(host));
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (int w = 0; w < STAMP_WAVES; ++w)
    for (int k = 0; k < 8; ++k) out8[k] += host[(size_t)w * 8 + k];
  for (size_t i = 0; i < (size_t)STAMP_WAVES * 8; ++i) host[i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stamp), host, sizeof(host));
  return 0;
}
#endif

extern "C" int pangu_window_attn_bwd_bf16(pangu_stream_t stream, const void* qkv, const void* qkv_bias, const void* esb,
                                          const void* out, const float* lse, const void* dout, void* dqkv,
                                          float* dqkv_bias, float* d_esb, int Z, int H, int W, int C, int heads,
                                          int shifted) {
  if (!qkv || !qkv_bias || !esb || !out || !lse || !dout || !dqkv || !dqkv_bias || !d_esb) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  // 32-bit buffer offsets; the pad-row sentinel 0x80000000 must stay OUTSIDE the (n_tok x 3C) buffers
  if ((size_t)Z * H * W * 3 * C * sizeof(u16) >= 0x7FFFFFF0ull) return PANGU_E_RANGE;
  const WinGeom g = make_geom(Z, H, W);
  const int n_pairs = (heads & 1) ? g.types * heads : ((g.types * heads / 2 + 7) / 8) * 16;    // padded to whole XCD rounds
  hipStream_t s = (hipStream_t)stream;
  const size_t shm2 = L_END;
  PANGU_ENSURE_DYN_LDS(window_attn_bwd2_bf16_kernel<true>, shm2);
  PANGU_ENSURE_DYN_LDS(window_attn_bwd2_bf16_kernel<false>, shm2);
  if (shifted)
    hipLaunchKernelGGL(window_attn_bwd2_bf16_kernel<true>, dim3(n_pairs), dim3(NT), shm2, s, (const u16*)qkv,
                       (const u16*)qkv_bias, (const u16*)esb, (const u16*)out, lse, (const u16*)dout, (u16*)dqkv,
                       dqkv_bias, d_esb, g, C, heads);
  else
    hipLaunchKernelGGL(window_attn_bwd2_bf16_kernel<false>, dim3(n_pairs), dim3(NT), shm2, s, (const u16*)qkv,
                       (const u16*)qkv_bias, (const u16*)esb, (const u16*)out, lse, (const u16*)dout, (u16*)dqkv,
                       dqkv_bias, d_esb, g, C, heads);
  return pangu_launch_status();
}
