// Earth-specific window attention INCLUDING the QKV projection (reference layers.py:365-415), bf16 -- the LONGITUDE-WALKING form
// (VERDICT r4 item 3): one persistent workgroup per (window type, head) keeps what every longitude
// window of that unit shares ON CHIP and walks the nLon windows:
//   * the head's 96 rows of linear1 (q, k, v: 3 x 32 x C bf16 = 36 / 72 KB) live in LDS for the workgroup's whole life, loaded
//     ONCE by LDS-DMA (the (window, head) kernel of attn_bf16.hip streams them again for every window and K-step: 6 of the 15 KB
//     a K-step moves through the L2 -> LDS path);
//   * the workgroup is G independent window PIPELINES of three waves (pipeline p takes windows p, p + G, ..) that share the
//     resident weights;
//   * a wave's K-loop needs no other wave: with the weights resident, the only streamed operand is the x slice of the wave's OWN
//     48 window rows, which goes global -> registers in MFMA-fragment shape (16 B per lane, zero-pad rows arrive as zeros from
//     the buffer range check) -- no LDS ring, no LDS-DMA issue cost, NO barrier in the K-loop;
//   * the three waves of a pipeline meet twice per window (before / after writing the window's K and V^T images, which all three
//     read) at an LDS-counter rendezvous private to the pipeline, so the G pipelines drift freely against each other: one's
//     softmax tiles run beside another's projection MFMAs and a third's loads, as independent workgroups do -- without the
//     per-window cold start (window addressing, mask bits, bias / weight first touch: 5k of the 25-33k cycles a (window, head)
//     workgroup lives);
//   * BIAS = 1: the wave's three 16-query x 144-key bias row blocks (54 VGPRs, bf16) stay in REGISTERS over the walk (G = 2: 256
//     registers per wave); BIAS = 0: they are re-read from L2 per window as in the (window, head) kernel (G = 4: 168 registers).
// attn_tile (scores, softmax, PV, store) is shared with attn_bf16.hip.
#include "common.h"
#include <stdlib.h>

#include "attn_bf16_tile.h"

namespace {

constexpr int WALK_IMG = PANGU_WTOK * 64 + VT_BYTES_SWZ;      // K image + V^T image of one pipeline: 21 504 B

__device__ __forceinline__ int fsw64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }      // F = {0,2,3,1}[(row>>2)&3]

// Rendezvous of the three waves of one pipeline on a monotonic LDS counter (gfx950 has one hardware barrier per workgroup; the
// pipelines must not couple).  Every wave first drains its own LDS traffic (reads returned, writes performed: the LDS executes
// a wave's operations in order), then lane 0 adds 1 and all poll until the three arrivals of this rendezvous are in.
__device__ __forceinline__ void pipe_rendezvous(unsigned* ctr, unsigned& target, int lane) {
  target += 3u;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (true) {
    const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

template <bool SHIFTED, int C, int G, int BIAS>
__global__ __launch_bounds__(192 * G) void window_attn_qkv_walk_bf16_kernel(const u16* __restrict__ x, int ldx,
                                                                            const u16* __restrict__ wqkv,
                                                                            const float* __restrict__ bqkv,
                                                                            const u16* __restrict__ esb, u16* __restrict__ out,
                                                                            float* __restrict__ lse, WinGeom g, int n_tok,
                                                                            int heads) {
  constexpr int KS = C / 32;                   // K-steps of 32 input channels
  constexpr int WSLAB = 96 * 64;               // one K-step of the resident weight image: 96 rows x 64 B
  constexpr int WIMG = KS * WSLAB;
  constexpr int NW = 3 * G;                    // waves
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const wimg = smem;                                       // [KS][96][64 B], 16-B chunks XOR fsw64(row)
  unsigned char* const imgs = smem + WIMG;                                // G x (K image, V^T image)
  float* const bq_s = reinterpret_cast<float*>(imgs + G * WALK_IMG);      // the head's 96 linear1.bias values
  unsigned* const ctrs = reinterpret_cast<unsigned*>(bq_s + 96);          // G rendezvous counters

  // Unit order: blocks b, b + 8, .. share an XCD (its L2): the `heads` units of one window type run side by side on one XCD and
  // walk the same windows at about the same time, so a window's x rows leave HBM once for all heads.
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int hd = local % heads;
  const int t = (local / heads) * 8 + xcd;
  if (t >= g.types) return;
  const int pair = t * heads + hd;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pipe = wave / 3, wg = wave - 3 * pipe;
  const u16* bias_tile = esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK;
  unsigned char* const Ks = imgs + pipe * WALK_IMG;
  unsigned char* const Vt = Ks + PANGU_WTOK * 64;
  unsigned* const ctr = ctrs + pipe;

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(x), 0, (int)(((size_t)(n_tok - 1) * ldx + C) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(wqkv), 0, 3 * C * C * (int)sizeof(u16), 0x00020000);

  // ---- once per workgroup: the head's weight rows -> LDS (KS x 6 pieces of 1 KB by LDS-DMA, source-side swizzle), the head's
  // bias values, the rendezvous counters
  for (int p = wave; p < KS * 6; p += NW) {
    const int ks = p / 6, q = p - 6 * ks;
    const int r = 16 * q + (lane >> 2);                 // image row: which * 32 + d
    const int c = (lane & 3) ^ fsw64(r);
    const int which = r >> 5, d = r & 31;
    const unsigned voff = ((unsigned)(which * C + hd * 32 + d) * (unsigned)C + ks * 32 + c * 8) * 2u;
    auto dst = (__attribute__((address_space(3))) void*)(wimg + ks * WSLAB + q * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, (int)voff, 0, 0, 0);
  }
  if (tid < 96) bq_s[tid] = bqkv[(tid >> 5) * C + hd * 32 + (tid & 31)];
  if (tid < G) ctrs[tid] = 0u;

  bool zcut = false, hcut = false;
  unsigned long long kz_bits_c = 0ull, kh_bits_c = 0ull;
  if (SHIFTED) {
    const int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
    zcut = zwin == g.nZw - 1;
    hcut = hwin == g.nHw - 1;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kn = key_of(j, (lane >> 4) * 4 + r);
        if (kn >= 72) kz_bits_c |= 1ull << (4 * j + r);
        if (((kn / 12) % 6) < 3) kh_bits_c |= 1ull << (4 * j + r);
      }
  }
  const int tile0 = 3 * wg;
  BiasRow br[BIAS ? 3 : 1];
  if (BIAS) {
#pragma unroll
    for (int i = 0; i < 3; ++i) br[BIAS ? i : 0] = load_bias_row(bias_tile, (tile0 + i) * 16 + (lane & 15), lane >> 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned target = 0u;
  const int lane_c = lane;
  for (int l = pipe; l < g.nLon; l += G) {
    // Against the optimiser: everything below that depends only on the lane and the window TYPE is invariant over the walk (LDS
    // fragment addresses, the 36 x 3 mask predicates of the shifted tiles as 64-bit lane masks, ..) and would be hoisted out of it
    // into ~60 VGPRs + ~170 SGPRs that then spill; the lane id and the mask bits are made opaque once per window instead
    int lane_w = lane_c;
    asm volatile("" : "+v"(lane_w));
    const int lq = lane_w & 15, lg = lane_w >> 4;
    unsigned kz_lo = (unsigned)kz_bits_c, kz_hi = (unsigned)(kz_bits_c >> 32), kh_lo = (unsigned)kh_bits_c, kh_hi = (unsigned)(kh_bits_c >> 32);
    asm volatile("" : "+v"(kz_lo), "+v"(kz_hi), "+v"(kh_lo), "+v"(kh_hi));
    const unsigned long long kz_bits = ((unsigned long long)kz_hi << 32) | kz_lo, kh_bits = ((unsigned long long)kh_hi << 32) | kh_lo;
    // ---- the wave's 48 window rows: source tokens (closed form) and the byte offsets of their 16-B fragment pieces
    int qtok[3];
    unsigned xoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      qtok[i] = win_src_token(g, l, t, (tile0 + i) * 16 + lq, SHIFTED);
      xoff[i] = qtok[i] >= 0 ? ((unsigned)qtok[i] * (unsigned)ldx + lg * 8) * 2u : 0x7FFFFFF0u;      // pad row: out of range -> zeros
    }
    // ---- q, k, v of the wave's rows: acc[rt 0,1 = q | 2,3 = k][tile] transposed (d = 4lg + r on the registers, token on the
    // lane); [rt 4,5 = v] token 4lg + r on the registers, d = 16(rt-4) + lq on the lane
    f32x4 acc[6][3];
#pragma unroll
    for (int rt = 0; rt < 6; ++rt)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[rt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PD = G >= 3 ? 1 : 2;             // x fragments requested PD K-steps ahead of their MFMAs (168 registers at G >= 3)
    bf16x8 fx[PD + 1][3];
#pragma unroll
    for (int ks = 0; ks < PD; ++ks)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        fx[ks][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)xoff[i], ks * 64, 0));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __builtin_amdgcn_sched_barrier(0);           // (keeps the unrolled steps' loads where they are written: PD steps ahead)
      if (ks + PD < KS) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          fx[(ks + PD) % (PD + 1)][i] =
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)xoff[i], (ks + PD) * 64, 0));
      }
      const unsigned char* slab = wimg + ks * WSLAB;
      bf16x8 fw[6];
#pragma unroll
      for (int rt = 0; rt < 6; ++rt) {
        const int row = rt * 16 + lq;
        fw[rt] = *reinterpret_cast<const bf16x8*>(slab + row * 64 + ((lg ^ fsw64(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bf16x8 xf = fx[ks % (PD + 1)][i];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[rt], xf, acc[rt][i], 0, 0, 0);
#pragma unroll
        for (int rt = 4; rt < 6; ++rt) acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, fw[rt], acc[rt][i], 0, 0, 0);
      }
    }
    // the first tile's bias rows travel under the rendezvous (BIAS = 0).  The tile does not depend on the window: without the
    // opaque offset the compiler hoists all three row blocks out of the walk AND unpacks them to fp32 there (108 registers)
    int lz = 0;
    asm volatile("" : "+v"(lz));
    const u16* const btile = bias_tile + lz;
    BiasRow b0;
    if (!BIAS) b0 = load_bias_row(btile, tile0 * 16 + lq, lg);
    // ---- + linear1.bias, then q fragments (registers), K image and V^T image (LDS; every wave of the pipeline must be done
    // reading the previous window's images)
    {
      f32x4 bq4[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) bq4[rt] = *reinterpret_cast<const f32x4*>(bq_s + (rt >> 1) * 32 + (rt & 1) * 16 + 4 * lg);
      const float bv0 = bq_s[64 + lq], bv1 = bq_s[64 + 16 + lq];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt][i] += bq4[rt];
        acc[4][i] += f32x4{bv0, bv0, bv0, bv0};
        acc[5][i] += f32x4{bv1, bv1, bv1, bv1};
      }
    }
    pipe_rendezvous(ctr, target, lane);
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
    pipe_rendezvous(ctr, target, lane);
    if (BIAS) {
      // resident rows stay PACKED (bf16 pairs, 54 registers): opaque to the optimiser once per window, or their fp32 unpacking is
      // hoisted out of the walk
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(br[BIAS ? i : 0].p[u][e]));
        asm volatile("" : "+v"(br[BIAS ? i : 0].t[0]), "+v"(br[BIAS ? i : 0].t[1]));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        attn_tile<SHIFTED, true>(Ks, Vt, qf[i], br[BIAS ? i : 0], (tile0 + i) * 16 + lq, qtok[i], lq, lg, zcut, hcut, kz_bits, kh_bits,
                                 out, lse, C, heads, hd);
    } else {
      const BiasRow b1 = load_bias_row(btile, (tile0 + 1) * 16 + lq, lg);
      attn_tile<SHIFTED, true>(Ks, Vt, qf[0], b0, tile0 * 16 + lq, qtok[0], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
      b0 = load_bias_row(btile, (tile0 + 2) * 16 + lq, lg);
      attn_tile<SHIFTED, true>(Ks, Vt, qf[1], b1, (tile0 + 1) * 16 + lq, qtok[1], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
      attn_tile<SHIFTED, true>(Ks, Vt, qf[2], b0, (tile0 + 2) * 16 + lq, qtok[2], lq, lg, zcut, hcut, kz_bits, kh_bits, out, lse, C, heads, hd);
    }
  }
}

template <bool SH, int CC, int G, int BIAS>
static int launch_walk(hipStream_t s, const void* x, int ldx, const void* w_qkv, const float* b_qkv, const void* esb, void* out,
                       float* lse, const WinGeom& g, int n_tok, int heads) {
  const size_t shm = (size_t)96 * CC * 2 + (size_t)G * WALK_IMG + 96 * sizeof(float) + G * sizeof(unsigned);
  if (shm > 160 * 1024) return PANGU_E_SHAPE;
  auto kern = window_attn_qkv_walk_bf16_kernel<SH, CC, G, BIAS>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  const int grid = ((g.types + 7) / 8) * 8 * heads;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(192 * G), shm, s, (const u16*)x, ldx, (const u16*)w_qkv, b_qkv, (const u16*)esb,
                     (u16*)out, lse, g, n_tok, heads);
  return pangu_launch_status();
}

}  // namespace

// variant = 10 * G + BIAS: 40 (four pipelines, bias rows from L2 per window), 21 (two pipelines, bias rows register-resident),
// 20, 30, 31 ..; anything not instantiated -> PANGU_E_SHAPE
extern "C" int pangu_window_attn_qkv_walk_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_qkv,
                                                   const float* b_qkv, const void* esb, void* out, float* lse, int Z, int H, int W,
                                                   int C, int heads, int shifted, int variant) {
  if (!x || !w_qkv || !b_qkv || !esb || !out) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM || (C != 192 && C != 384) || ldx < C || (ldx & 7)) return PANGU_E_SHAPE;
  const int n_tok = Z * H * W;
  if (!pangu_fits_u32(n_tok, ldx, 2) || (size_t)n_tok * ldx * 2 >= 0x7FFFFFF0ull) return PANGU_E_RANGE;
  const WinGeom g = make_geom(Z, H, W);
  hipStream_t s = (hipStream_t)stream;
#define PANGU_WALK(GG, BB)                                                                                                   \
  if (variant == 10 * GG + BB) {                                                                                            \
    if (C == 192) return shifted ? launch_walk<true, 192, GG, BB>(s, x, ldx, w_qkv, b_qkv, esb, out, lse, g, n_tok, heads)  \
                                 : launch_walk<false, 192, GG, BB>(s, x, ldx, w_qkv, b_qkv, esb, out, lse, g, n_tok, heads); \
    return shifted ? launch_walk<true, 384, GG, BB>(s, x, ldx, w_qkv, b_qkv, esb, out, lse, g, n_tok, heads)                \
                   : launch_walk<false, 384, GG, BB>(s, x, ldx, w_qkv, b_qkv, esb, out, lse, g, n_tok, heads);              \
  }
  PANGU_WALK(4, 0)
  PANGU_WALK(3, 0)
  PANGU_WALK(2, 0)
  PANGU_WALK(2, 1)
  PANGU_WALK(1, 1)
#undef PANGU_WALK
  return PANGU_E_SHAPE;
}
