// Fused MLP branch of an EarthSpecificBlock for gfx950, bf16 operands / fp32 accumulation (inference path):
//
//   out[m,:] = x[m,:] + s * ( LayerNorm( GELU(x[m,:] W1^T + b1) W2^T + b2 ) * gamma + beta )
//
// i.e. reference models/layers.py:251 (`x + drop_path(norm2(linear(x)))`) with Mlp.forward (:264-270) inside, in ONE
// launch: the (tokens x 4C) hidden activation never exists in memory (unfused it is written and re-read once per block:
// 1.6 GB of the 3.6 GB a C = 192 block moves in bf16), and the LayerNorm + residual run on the accumulators.
//
// Structure (MI355X: 160 KB LDS, 512 registers per lane at one wave per SIMD):
//  * a workgroup = 4 waves owns 64*T consecutive tokens, wave w the 16*T tokens w*16T .. (T token tiles of 16);
//    the wave's activations x[16T][C] are loaded ONCE, straight from global memory in MFMA-fragment shape, and stay in
//    registers (C/32 * T fragments) for the whole tile -- they are the B operand of the first product;
//  * both products are computed TRANSPOSED with v_mfma_f32_16x16x32_bf16, weights as the A operand:
//        H^T[hidden][token] = W1[hidden][:] . x^T          (K = C)
//        Y^T[c][token]     += W2[c][hidden chunk] . GELU(H^T)[hidden chunk][token]     (K = 32 per chunk)
//    so the accumulator of the first product (hidden on the register index, token on the lane) IS the B operand of the
//    second one after GELU + bf16 packing (the bias is the accumulator's initial value): two 16-row hidden tiles give
//    each lane the eight k values {4g..4g+3} u {16+4g..16+4g+3} of a 32-deep k-step; the W2 fragment is stored in that
//    k order by the host-side packing, so nothing crosses lanes and nothing goes through LDS between the two products;
//  * Y^T (C x 16T per wave, fp32) stays in 4*(C/16)*T accumulator registers over all hidden chunks;
//  * the weights stream L2 -> LDS by LDS-DMA in chunks of 32 hidden units, W1 rows and W2 columns as two streams with
//    a ring of 3 slots each (the host packs both contiguously, layout below), one raw s_barrier per chunk, counted
//    vmcnt (two chunks ahead); every fragment read is a conflict-free ds_read_b128 feeding T MFMAs;
//  * three-stage software pipeline over chunks -- with one wave per SIMD nothing but the wave's own instruction order
//    overlaps VALU, LDS and DMA-issue work with the matrix pipe: iteration i runs the SECOND product of chunk i-2, then
//    the FIRST product of chunk i, and spreads the GELU + pack of chunk i-1 and the LDS-DMA requests of chunks i+2 / i
//    between the MFMAs of both (step boundaries pinned with sched_barrier, MFMA : VALU interleave inside a step with
//    sched_group_barrier);
//  * epilogue per 16-token tile: shortcut rows staged into a per-wave LDS patch (coalesced 16-B loads), LayerNorm
//    statistics in fp32 over the lane's 4*C/16 values + two cross-lane adds, normalise / gamma / beta / scale / residual
//    in the MFMA layout against the patch, rows read back and stored as whole 16-B segments.
//
// Packed weight image (u16 elements) [2][4C/32][32*C]: plane 0 = W1 stream, plane 1 = W2 stream, chunk ch each:
//   W1 chunk: row hr = 0..31 (hidden 32ch+hr), 16-B chunk position p = 0..C/8-1 holds the eight K values
//             8q..8q+7 of that row with q = p ^ (hr & SW), SW = 15 (C = 384) / 7 (C = 192)   (bank swizzle)
//   W2 chunk: plane g = 0..3, row c = 0..C-1, 8 elements j: W2[c][32ch + (j<4 ? 4g+j : 16+4g+j-4)]
#include "common.h"
#include <type_traits>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr float LN_EPS = 1e-5f;

// Development knobs (tools/ablate_mlp.py builds timing variants with -D...; the product uses the defaults).
#ifndef PANGU_MLP_ABLATE
#define PANGU_MLP_ABLATE 0      // timing only: 1 no in-loop weight requests, 2 no GELU, 4 / 8 no first / second product
#endif
#ifndef PANGU_MLP_PD1
#define PANGU_MLP_PD1 2         // fragment-read distance (k-steps ahead of the MFMAs), first product
#endif
#ifndef PANGU_MLP_PD2
#define PANGU_MLP_PD2 4         // the same (row tiles ahead), second product
#endif
#ifndef PANGU_MLP_IGLP
#define PANGU_MLP_IGLP 3        // VALU instructions placed behind each MFMA of a step (0 = leave it to the scheduler)
#endif
constexpr int ABL = PANGU_MLP_ABLATE;
constexpr int PD1 = PANGU_MLP_PD1, PD2 = PANGU_MLP_PD2, IGLP = PANGU_MLP_IGLP;

__device__ inline float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ inline float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// GELU for the bf16 hidden activation: x * sigmoid(x (c0 + c1 x^2)) with (c0, c1) fitted to the EXACT erf form the
// reference uses (nn.GELU(), layers.py:261): max |err| = 2.7e-4 over all x (the degree-6 erf polynomial of the unfused
// bf16 epilogue: 3.7e-4), below the bf16 rounding of the result; 5 plain VALU + v_exp_f32 + v_rcp_f32 per element --
// with one wave per SIMD every VALU instruction costs issue time beside the MFMAs, so the cheapest form wins.
__device__ __forceinline__ float gelu1(float x) {
  const float t = x * x;
  const float w = fmaf(t, -0.06940179f * 1.4426950408889634f, -1.60031416f * 1.4426950408889634f);
  const float e = __builtin_amdgcn_exp2f(x * w);                    // exp(-x (c0 + c1 x^2)); +inf for x << 0 -> result 0
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

template <bool B>
using Flag = std::integral_constant<bool, B>;
template <int N>
using Int = std::integral_constant<int, N>;

template <int C, int T>
__global__ __launch_bounds__(256, 1) void mlp_ln_residual_bf16_kernel(
    const u16* __restrict__ X, int ldx, const u16* __restrict__ Wimg, const float* __restrict__ b1,
    const float* __restrict__ b2, const float* __restrict__ gamma, const float* __restrict__ beta,
    u16* __restrict__ Out, int ldo, int M, float scale) {
  constexpr int HID = 4 * C, NCH = HID / 32, KS = C / 32, RT = C / 16;
  constexpr int WB = 64 * C;                  // bytes of one W1 (or W2) chunk
  constexpr int NS = 3;                       // ring slots per stream
  constexpr int LPS = WB / 4096;              // LDS-DMA instructions per wave, chunk and stream (1 KB each)
  constexpr int LPW = 2 * LPS;                // ... per wave and iteration
  constexpr int SW = C == 384 ? 15 : 7;
  constexpr int PLD = 2 * C + 16;             // bytes per row of an epilogue patch
  constexpr int CPR = C / 8;                  // 16-B chunks per activation row
  constexpr int E = 8 * T;                    // GELU elements per lane and chunk: [tt][ht][r]
  static_assert(KS % LPS == 0 && RT % LPS == 0 && NCH > 4, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ring1 = smem;                      // W1 stream, NS slots
  unsigned char* const ring2 = smem + NS * WB;            // W2 stream, NS slots
  float* const b1s = reinterpret_cast<float*>(smem + 2 * NS * WB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * (64 * T) + wave * (16 * T);

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(X), 0, (int)(((size_t)(M - 1) * ldx + C) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(Wimg), 0, 2 * NCH * WB, 0x00020000);

  // ---- activations of this wave: fragment (ks, tt) = x[m0 + 16tt + lq][32ks + 8lg .. +7]  (rows >= M read as zeros)
  bf16x8 xf[KS][T];
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    const unsigned row = (unsigned)(m0 + 16 * tt + lq);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      xf[ks][tt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                  x_rsrc, (int)((row * (unsigned)ldx + 32 * ks + 8 * lg) * 2u), 0, 0));
  }
  // ---- b1 -> LDS (fp32)
  for (int i = tid; i < HID / 4; i += 256)
    reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(b1)[i];

  // piece i (0 .. LPS-1) of this wave for chunk ch of stream st (0 = W1, 1 = W2) -> ring slot `slot` of that stream
  auto issue_piece = [&](int st, int ch, int slot, int i) {
    const int q = i * 4 + wave;
    auto dst = (__attribute__((address_space(3))) void*)((st ? ring2 : ring1) + slot * WB + q * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane * 16, (st * NCH + ch) * WB + q * 1024, 0, 0);
  };

  f32x4 yacc[RT][T];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int tt = 0; tt < T; ++tt) yacc[rt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 hacc[2][T];          // first product of the current chunk (bias included)
  f32x4 hg[2][T];            // first product of the previous chunk, GELU applied element by element during this iteration
  bf16x8 hf[T];              // packed GELU output of the chunk before that: B operand of the second product

  // One iteration: G2 = second product of chunk ch-2 (operand hf), GE = GELU of chunk ch-1 (hg), G1 = first product of
  // chunk ch (-> hacc); I1 / I2: request W1 chunk ch+2 / W2 chunk ch.  All flags are compile-time (peeled prologue /
  // drain iterations), so the steady-state body is one basic block.
  auto iteration = [&](int ch, int a /* ch % 3 */, auto g2, auto ge, auto g1, auto i1, auto i2) {
    constexpr bool G2 = decltype(g2)::value, GE = decltype(ge)::value, G1 = decltype(g1)::value;
    constexpr bool I1 = decltype(i1)::value && !(ABL & 1), I2 = decltype(i2)::value && !(ABL & 1);
    const int a1 = a + 1 >= NS ? a + 1 - NS : a + 1, a2 = a + 2 >= NS ? a + 2 - NS : a + 2;
    const unsigned char* w1 = ring1 + a * WB;              // W1 chunk ch
    const unsigned char* w2 = ring2 + a1 * WB;             // W2 chunk ch-2  ((ch-2) % 3 == (ch+1) % 3)
    // GELU element e of hg in the order [tt][ht][r]: the first half is done beside the second product, the rest beside
    // the first product
    auto gelu_elems = [&](int e0, int e1) {
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        const int tt = e >> 3, ht = (e >> 2) & 1, r = e & 3;
        if (!(ABL & 2)) hg[ht][tt][r] = gelu1(hg[ht][tt][r]);
      }
    };
    // ---- second product of chunk ch-2: RT steps of T MFMAs, fragment reads PD2 steps ahead
    if constexpr (G2) {
      auto rd = [&](int rt) { return *reinterpret_cast<const bf16x8*>(w2 + lg * (16 * C) + (16 * rt + lq) * 16); };
      bf16x8 fa[RT];
#pragma unroll
      for (int rt = 0; rt < PD2 && rt < RT; ++rt) fa[rt] = rd(rt);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (rt + PD2 < RT) fa[rt + PD2] = rd(rt + PD2);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 8)) {
#pragma unroll
          for (int tt = 0; tt < T; ++tt)
            yacc[rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt], hf[tt], yacc[rt][tt], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(fa[rt]), "v"(hf[0]));
        }
        if constexpr (GE) gelu_elems((E / 2) * rt / RT, (E / 2) * (rt + 1) / RT);
        if (I2 && rt % (RT / LPS) == 0) issue_piece(1, ch, a, rt / (RT / LPS));
        if (IGLP && GE) {
#pragma unroll
          for (int i = 0; i < T; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, IGLP, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if constexpr (GE) gelu_elems(0, E / 2);
      if (I2) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) issue_piece(1, ch, a, i);
      }
    }
    // ---- first product of chunk ch: KS steps of 2T MFMAs, fragment reads PD1 steps ahead; accumulators start from the
    // bias of this lane's hidden rows 4lg .. 4lg+3 of both 16-row tiles
    if constexpr (G1) {
      const f32x4 bv0 = *reinterpret_cast<const f32x4*>(b1s + 32 * ch + 4 * lg);
      const f32x4 bv1 = *reinterpret_cast<const f32x4*>(b1s + 32 * ch + 16 + 4 * lg);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        hacc[0][tt] = bv0;
        hacc[1][tt] = bv1;
      }
      auto rd = [&](int ks, int ht) {
        const int pc = (4 * ks + lg) ^ (lq & SW);
        return *reinterpret_cast<const bf16x8*>(w1 + (16 * ht + lq) * (2 * C) + pc * 16);
      };
      bf16x8 fa[KS][2];
#pragma unroll
      for (int ks = 0; ks < PD1 && ks < KS; ++ks) {
        fa[ks][0] = rd(ks, 0);
        fa[ks][1] = rd(ks, 1);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + PD1 < KS) {
          fa[ks + PD1][0] = rd(ks + PD1, 0);
          fa[ks + PD1][1] = rd(ks + PD1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4)) {
#pragma unroll
          for (int tt = 0; tt < T; ++tt) {
            hacc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][0], xf[ks][tt], hacc[0][tt], 0, 0, 0);
            hacc[1][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][1], xf[ks][tt], hacc[1][tt], 0, 0, 0);
          }
        } else {
          asm volatile("" ::"v"(fa[ks][0]), "v"(fa[ks][1]));
        }
        if constexpr (GE) gelu_elems(E / 2 + (E / 2) * ks / KS, E / 2 + (E / 2) * (ks + 1) / KS);
        if (I1 && ks % (KS / LPS) == 0) issue_piece(0, ch + 2, a2, ks / (KS / LPS));
        if (IGLP && GE) {
#pragma unroll
          for (int i = 0; i < 2 * T; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, IGLP, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if constexpr (GE) gelu_elems(E / 2, E);
    }
    // ---- rotate the pipeline registers: hf <- pack(GELU(chunk ch-1)), hg <- chunk ch
    if constexpr (GE) {
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
        hf[tt] = __builtin_bit_cast(bf16x8, u32x4{pack_bf16x2(hg[0][tt][0], hg[0][tt][1]), pack_bf16x2(hg[0][tt][2], hg[0][tt][3]),
                                                  pack_bf16x2(hg[1][tt][0], hg[1][tt][1]), pack_bf16x2(hg[1][tt][2], hg[1][tt][3])});
    }
    if constexpr (G1) {
#pragma unroll
      for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int tt = 0; tt < T; ++tt) hg[ht][tt] = hacc[ht][tt];
    }
  };
  // top of iteration ch: the request group of iteration ch-2 has landed (this wave's pieces: vmcnt leaves the N newest
  // requests in flight; every wave's: barrier); the slots it re-requests now were last read in iteration ch-1
  auto sync = [&](auto n) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // no fragment read of the slots re-requested next may still be queued
    wait_vmcnt<decltype(n)::value>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using Y = Flag<true>;
  using N_ = Flag<false>;

  // ---- prologue: W1 chunks 0, 1 requested; iterations 0 and 1 have no second product yet
#pragma unroll
  for (int i = 0; i < LPS; ++i) issue_piece(0, 0, 0, i);
#pragma unroll
  for (int i = 0; i < LPS; ++i) issue_piece(0, 1, 1, i);
  sync(Int<LPS>{});                                       // W1 chunk 0 (and x, b1) landed; W1 chunk 1 in flight
  iteration(0, 0, N_{}, N_{}, Y{}, Y{}, Y{});
  sync(Int<LPW>{});
  iteration(1, 1, N_{}, Y{}, Y{}, Y{}, Y{});
  int a = 2;
  for (int ch = 2; ch < NCH - 2; ++ch) {
    sync(Int<LPW>{});
    iteration(ch, a, Y{}, Y{}, Y{}, Y{}, Y{});
    a = a + 1 == NS ? 0 : a + 1;
  }
  // ---- drain: chunks NCH-2, NCH-1 request no W1 any more; then two iterations without a first product
  sync(Int<LPW>{});
  iteration(NCH - 2, (NCH - 2) % NS, Y{}, Y{}, Y{}, N_{}, Y{});
  sync(Int<LPS>{});
  iteration(NCH - 1, (NCH - 1) % NS, Y{}, Y{}, Y{}, N_{}, Y{});
  sync(Int<LPS>{});
  iteration(NCH, NCH % NS, Y{}, Y{}, N_{}, N_{}, N_{});
  sync(Int<0>{});
  iteration(NCH + 1, (NCH + 1) % NS, Y{}, N_{}, N_{}, N_{}, N_{});
  __syncthreads();                               // every wave is done with the rings: the patches may reuse them

  // ---- epilogue: lane holds Y^T[c = 16rt + 4lg + r][token m0 + 16tt + lq]
  unsigned char* patch = smem + wave * (16 * PLD);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      Out, 0, (int)(((size_t)(M - 1) * ldo + C) * sizeof(u16)), 0x00020000);
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    const int tok0 = m0 + 16 * tt;
    // shortcut rows -> patch
#pragma unroll
    for (int it = 0; it < 16 * CPR / 64; ++it) {
      const int f = lane + 64 * it, row = f / CPR, chk = f - row * CPR;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
          x_rsrc, (int)(((unsigned)(tok0 + row) * (unsigned)ldx + chk * 8) * 2u), 0, 0);
      *reinterpret_cast<u32x4*>(patch + row * PLD + chk * 16) = v;
    }
    float s = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      yacc[rt][tt] += *reinterpret_cast<const f32x4*>(b2 + 16 * rt + 4 * lg);
      s += (yacc[rt][tt][0] + yacc[rt][tt][1]) + (yacc[rt][tt][2] + yacc[rt][tt][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float ss = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = yacc[rt][tt][r] - mean;
        ss = fmaf(d, d, ss);
      }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = rsqrtf(ss * (1.0f / C) + LN_EPS);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * rt + 4 * lg);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * rt + 4 * lg);
      unsigned char* slotp = patch + lq * PLD + (16 * rt + 4 * lg) * 2;
      const u32x2 xs = *reinterpret_cast<const u32x2*>(slotp);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf((yacc[rt][tt][r] - mean) * rstd, gm[r], bt[r]) * scale;
      o[0] += bflo(xs[0]); o[1] += bfhi(xs[0]); o[2] += bflo(xs[1]); o[3] += bfhi(xs[1]);
      *reinterpret_cast<u32x2*>(slotp) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
#pragma unroll
    for (int it = 0; it < 16 * CPR / 64; ++it) {
      const int f = lane + 64 * it, row = f / CPR, chk = f - row * CPR;
      const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * PLD + chk * 16);
      const unsigned off = tok0 + row < M ? ((unsigned)(tok0 + row) * (unsigned)ldo + chk * 8) * 2u : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(v, o_rsrc, (int)off, 0, 0);
    }
  }
}

template <int C, int T>
int launch_mlp(hipStream_t s, const u16* x, int ldx, const u16* wimg, const float* b1, const float* b2,
               const float* gamma, const float* beta, u16* out, int ldo, int M, float scale) {
  constexpr size_t ring = (size_t)6 * 64 * C + 4 * C * sizeof(float);
  constexpr size_t epi = (size_t)4 * 16 * (2 * C + 16);
  constexpr size_t shm = ring > epi ? ring : epi;
  static_assert(shm <= 160 * 1024, "LDS");
  auto kern = mlp_ln_residual_bf16_kernel<C, T>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  const int grid = (M + 64 * T - 1) / (64 * T);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, s, x, ldx, wimg, b1, b2, gamma, beta, out, ldo, M, scale);
  return pangu_launch_status();
}

}  // namespace

extern "C" int pangu_mlp_ln_residual_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_packed,
                                              const float* b1, const float* b2, const float* gamma, const float* beta,
                                              void* out, int ldo, int M, int C, float branch_scale) {
  if (!x || !w_packed || !b1 || !b2 || !gamma || !beta || !out) return PANGU_E_NULL;
  if (M <= 0 || ldx < C || ldo < C || (ldx & 7) || (ldo & 7)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, ldx, 2) || !pangu_fits_u32(M, ldo, 2)) return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  if (C == 192)
    return launch_mlp<192, 4>(s, (const u16*)x, ldx, (const u16*)w_packed, b1, b2, gamma, beta, (u16*)out, ldo, M,
                              branch_scale);
  if (C == 384)
    return launch_mlp<384, 2>(s, (const u16*)x, ldx, (const u16*)w_packed, b1, b2, gamma, beta, (u16*)out, ldo, M,
                              branch_scale);
  return PANGU_E_SHAPE;
}
