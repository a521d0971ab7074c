// Fused MLP branch of an EarthSpecificBlock for gfx950, bf16 operands / fp32 accumulation (inference path):
//
//   out[m,:] = x[m,:] + s * ( LayerNorm( GELU(x[m,:] W1^T + b1) W2^T + b2 ) * gamma + beta )
//
// i.e. reference models/layers.py:251 (`x + drop_path(norm2(linear(x)))`) with Mlp.forward (:264-270) inside, in ONE
// launch: the (tokens x 4C) hidden activation never exists in memory (unfused it is written and re-read once per block:
// 1.6 GB of the 3.6 GB a C = 192 block moves in bf16), and the LayerNorm + residual run on the accumulators.
//
// Structure (MI355X: 160 KB LDS, 512 registers per lane at one wave per SIMD):
//  * a workgroup = 4 waves owns 128*T consecutive tokens, wave w the 32*T tokens w*32T .. (T token tiles of 32);
//    the wave's activations x[32T][C] are loaded ONCE, straight from global memory in MFMA-fragment shape, and stay in
//    registers (C/16 * T fragments) for the whole tile -- they are the B operand of the first product;
//  * both products are computed TRANSPOSED with v_mfma_f32_32x32x16_bf16, weights as the A operand:
//        H^T[hidden][token] = W1[hidden][:] . x^T          (K = C)
//        Y^T[c][token]     += W2[c][hidden chunk] . GELU(H^T)[hidden chunk][token]     (K = 32 per chunk = 2 k-steps)
//    so the accumulator of the first product (hidden on the register index, token on the lane) IS the B operand of the
//    second one after GELU + bf16 packing (the bias is the accumulator's initial value): registers 8s..8s+7 are k-step s,
//    element j of lane half h being hidden row 16s + 8(j>>2) + 4h + (j&3); the W2 fragment is stored in that k order by
//    the host-side packing, so nothing crosses lanes and nothing goes through LDS between the two products.
//    The 32x32x16 shape (32 matrix cycles per instruction) is what makes a one-wave-per-SIMD kernel feasible: measured
//    on the 16x16x32 version of this kernel (PMC, gpurun_out/pmc_mlp2), every instruction of the single wave costs
//    >= 4 issue cycles and only 8 of a 16-cycle MFMA's cycles are free for others -- with 3.6 other instructions per
//    MFMA it was issue-bound at 0.39 MFMA-busy; the same work in half as many MFMAs leaves 24 free cycles each;
//  * Y^T (C x 32T per wave, fp32) stays in 16*(C/32)*T accumulator registers over all hidden chunks;
//  * the weights stream L2 -> LDS by LDS-DMA in chunks of 32 hidden units, W1 rows and W2 columns as two streams with
//    a ring of 3 slots each (the host packs both contiguously, layout below), one raw s_barrier per chunk, counted
//    vmcnt (two chunks ahead); every fragment read is a conflict-free ds_read_b128 feeding T MFMAs;
//  * three-stage software pipeline over chunks -- with one wave per SIMD nothing but the wave's own instruction order
//    overlaps VALU, LDS and DMA-issue work with the matrix pipe: iteration i runs the SECOND product of chunk i-2, then
//    the FIRST product of chunk i, and spreads the GELU + pack of chunk i-1 and the LDS-DMA requests of chunks i+2 / i
//    between the MFMAs of both (step boundaries pinned with sched_barrier, MFMA : VALU interleave inside a step with
//    sched_group_barrier);
//  * epilogue per 32-token tile: shortcut rows staged into a per-wave LDS patch (coalesced 16-B loads), LayerNorm
//    statistics in fp32 over the lane's 16*C/32 values + one cross-lane add, normalise / gamma / beta / scale / residual
//    in the MFMA layout against the patch, rows read back and stored as whole 16-B segments.
//
// Packed weight image (u16 elements) [2][4C/32][32*C]: plane 0 = W1 stream, plane 1 = W2 stream, chunk ch each:
//   W1 chunk: row hr = 0..31 (hidden 32ch+hr), 16-B chunk position p = 0..C/8-1 holds the eight K values
//             8q..8q+7 of that row with q = p ^ f(hr), f = hr & 15 (C = 384) / (hr >> 1) & 7 (C = 192)   (bank swizzle)
//   W2 chunk: [s = 0,1][h = 0,1][row c = 0..C-1][8 elements j]: W2[c][32ch + 16s + 8(j>>2) + 4h + (j&3)]
#include "common.h"
#include <stdlib.h>
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
#define PANGU_MLP_PD1 3         // fragment-read distance (k-steps ahead of the MFMAs), first product
#endif
#ifndef PANGU_MLP_PD2
#define PANGU_MLP_PD2 4         // the same (row tiles ahead), second product
#endif
#ifndef PANGU_MLP_IGLP
#define PANGU_MLP_IGLP 4        // VALU instructions placed behind each MFMA of a step (0 = leave it to the scheduler)
#endif
#ifndef PANGU_MLP_NT_X
#define PANGU_MLP_NT_X 0        // cache-policy bits of the once-read activation loads (2 = nt: keep the weight image in L2): A/B in profiles/r03 notes
#endif
#ifndef PANGU_MLP_NT_OUT
#define PANGU_MLP_NT_OUT 0      // ... of the result / side-output stores
#endif
#ifndef PANGU_MLP_NT_SIDE
#define PANGU_MLP_NT_SIDE 0     // ... of the training side outputs (pre, m)
#endif
#ifndef PANGU_MLP_PRE_VARIANT
#define PANGU_MLP_PRE_VARIANT 0 // training variant, how the pre-activation leaves: 0 = 16-B pieces via v_permlane32_swap, 1 = the same with s_nop padding (hazard probe), 2 = 8-B pieces, no exchange
#endif
#ifndef PANGU_MLP_PRE_AT_TOP
#define PANGU_MLP_PRE_AT_TOP 0  // training variant: pre-activation stores at the very top of an iteration instead of under the first fragment reads
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

#ifdef PANGU_MLP_STAMP
// Diagnostic build only (tools/ablate_mlp.py): where a steady-state iteration spends its cycles.  Sums over all waves of
// s_memtime differences: [0] sync (wait + barrier), [1] second product phase, [2] first product phase, [3] iterations,
// [4] whole kernel per wave, [5] waves.
__device__ unsigned long long g_stamp[8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif

template <bool B>
using Flag = std::integral_constant<bool, B>;
template <int N>
using Int = std::integral_constant<int, N>;

// TR (training forward): two side outputs for the backward pass, written from the registers the values live in anyway --
//   Pre[m][4C] (bf16): the first product + b1 BEFORE GELU (what gelu' needs; the backward's data-gradient GEMM also
//                     re-creates h = GELU(pre) from it for the W2 weight gradient, so h is never stored by the forward);
//                     may be null (recompute mode: the backward re-runs the MLP-up GEMM);
//   Mo[m][C]   (bf16): the second product + b2 BEFORE LayerNorm (what the LayerNorm backward normalises again).
template <int C, int T, int NW, int TR>      // TR: 0 inference, 2 training (side outputs Pre + Mo)
__global__ __launch_bounds__(64 * NW, NW / 4) void mlp_ln_residual_bf16_kernel(
    const u16* __restrict__ X, int ldx, const u16* __restrict__ Wimg, const float* __restrict__ b1,
    const float* __restrict__ b2, const float* __restrict__ gamma, const float* __restrict__ beta,
    u16* __restrict__ Out, int ldo, int M, float scale, u16* __restrict__ Pre, int ldp, u16* __restrict__ Mo, int ldm) {
  constexpr int HID = 4 * C, NCH = HID / 32, KS = C / 16, RT = C / 32;
  constexpr int WB = 64 * C;                  // bytes of one W1 (or W2) chunk
  constexpr int NS = 3;                       // ring slots per stream
  // LDS-DMA instructions (1 KB each) per wave: NW = 4: every wave requests LPS pieces of BOTH streams per iteration; NW = 8
  // (two waves per SIMD): waves 0-3 request the W1 stream, waves 4-7 the W2 stream, LPS pieces each
  constexpr int LPS = WB / 4096;
  constexpr int LPW = NW == 4 ? 2 * LPS : LPS;          // ... per wave and iteration
  constexpr int PLD = 2 * C + 16;             // bytes per row of an epilogue patch
  constexpr int CPR = C / 8;                  // 16-B chunks per activation row
  constexpr int E = 16 * T;                   // GELU elements per lane and chunk: [tt][register]
  static_assert(KS % LPS == 0 && (2 * RT) % LPS == 0 && NCH > 4, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [b1 | b2, gamma, beta | W1 ring | W2 ring]; the epilogue's per-wave patches reuse the rings (and may extend past them)
  float* const b1s = reinterpret_cast<float*>(smem);
  float* const eps = b1s + HID;                           // (b2 |) gamma | beta (C floats each) for the epilogue
  constexpr int PRE = (HID + 3 * C) * 4;
  unsigned char* const ring1 = smem + PRE;                // W1 stream, NS slots
  unsigned char* const ring2 = ring1 + NS * WB;           // W2 stream, NS slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * (32 * T * NW) + wave * (32 * T);
  const int role = NW == 4 ? -1 : (wave >> 2);             // NW = 8: which stream this wave requests
  const int fsw = C == 384 ? (lr & 15) : ((lr >> 1) & 7);      // this lane's W1 chunk swizzle

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(X), 0, (int)(((size_t)(M - 1) * ldx + C) * sizeof(u16)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(Wimg), 0, 2 * NCH * WB, 0x00020000);

  // ---- activations of this wave: fragment (ks, tt) = x[m0 + 32tt + lr][16ks + 8lh .. +7]  (rows >= M read as zeros)
  bf16x8 xf[KS][T];
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    const unsigned row = (unsigned)(m0 + 32 * tt + lr);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      xf[ks][tt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                  x_rsrc, (int)((row * (unsigned)ldx + 16 * ks + 8 * lh) * 2u), 0, PANGU_MLP_NT_X));
  }
  // ---- b1 and the epilogue's per-channel vectors -> LDS (fp32): as global loads in the epilogue (144 per lane, each
  // waited for at its use) they cost a quarter of the kernel
  for (int i = tid; i < HID / 4; i += 64 * NW)
    reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(b1)[i];
  for (int i = tid; i < 3 * C / 4; i += 64 * NW) {
    const int which = i / (C / 4), j = i - which * (C / 4);
    reinterpret_cast<f32x4*>(eps)[i] = reinterpret_cast<const f32x4*>(which == 0 ? b2 : which == 1 ? gamma : beta)[j];
  }

  // piece i (0 .. LPS-1) of this wave for chunk ch of stream st (0 = W1, 1 = W2) -> ring slot `slot` of that stream
  auto issue_piece = [&](int st, int ch, int slot, int i) {
    if (NW == 8 && st != role) return;
    const int q = i * 4 + (wave & 3);
    auto dst = (__attribute__((address_space(3))) void*)((st ? ring2 : ring1) + slot * WB + q * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane * 16, (st * NCH + ch) * WB + q * 1024, 0, 0);
  };

#ifdef PANGU_MLP_STAMP
  unsigned long long st_sync = 0, st_b = 0, st_all = 0, st_n = 0, st_last = 0;
  const unsigned long long st_begin = stamp();
#endif
  // second-product accumulators start from b2 of the lane's channels 32rt + (i&3) + 8(i>>2) + 4lh (no bias pass in the epilogue)
  f32x16 yacc[RT][T];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + 32 * rt + 8 * q + 4 * lh);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        yacc[rt][tt][4 * q] = bv[0]; yacc[rt][tt][4 * q + 1] = bv[1]; yacc[rt][tt][4 * q + 2] = bv[2]; yacc[rt][tt][4 * q + 3] = bv[3];
      }
    }
  f32x16 hacc[T];            // first product of the current chunk (bias included)
  float hg[T][16];           // first product of the previous chunk, GELU applied stage by stage during this iteration
  bf16x8 hf[2][T];           // packed GELU output of the chunk before that: B operand (k-step s) of the second product
  float gt[16 * T];          // GELU temporaries, one per element

  // TR: the pre-activation of chunk `chunk` (still untouched in hg: the GELU's last stage overwrites it) leaves as 16-B
  // pieces: lane (token, half h) holds hidden 8q + 4h + r (q = register quad, r = 0..3); two v_permlane32_swap per quad
  // pair give half 0 the eight consecutive hidden values of quad 2qp and half 1 those of quad 2qp + 1: per token row and
  // chunk one contiguous 64-B segment, which the next chunk's stores extend (write-combined in L2).
  const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      Pre, 0, TR == 2 ? (int)(((size_t)(M - 1) * ldp + HID) * sizeof(u16)) : 0, 0x00020000);
  // one address register for all T tiles (tile tt adds a scalar offset); rows past M are dropped by the range check
  const unsigned pre_off = ((unsigned)(m0 + lr) * (unsigned)ldp + 8 * lh) * 2u;
  [[maybe_unused]] const unsigned pre_off8 = ((unsigned)(m0 + lr) * (unsigned)ldp + 4 * lh) * 2u;      // (8-B pieces: PANGU_MLP_PRE_VARIANT 2)
  const int pre_tile = __builtin_amdgcn_readfirstlane(32 * ldp * 2);
  auto store_pre = [&](int chunk) {
    if (TR != 2) return;
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        // quads 2qp, 2qp+1 only: four packed registers live at a time (the kernel sits at the 512-register limit)
        unsigned a0 = pack_bf16x2(hg[tt][8 * qp], hg[tt][8 * qp + 1]), a1 = pack_bf16x2(hg[tt][8 * qp + 2], hg[tt][8 * qp + 3]);
        unsigned b0 = pack_bf16x2(hg[tt][8 * qp + 4], hg[tt][8 * qp + 5]), b1 = pack_bf16x2(hg[tt][8 * qp + 6], hg[tt][8 * qp + 7]);
#if PANGU_MLP_PRE_VARIANT == 2
        // no cross-lane exchange: each lane stores its own two 8-B pieces (hidden 8q + 4h .. +3 of quads 2qp, 2qp+1)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{a0, a1}, p_rsrc, (int)pre_off8, tt * pre_tile + (chunk * 32 + 16 * qp) * 2, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{b0, b1}, p_rsrc, (int)pre_off8, tt * pre_tile + (chunk * 32 + 16 * qp + 8) * 2, 0);
#else
#if PANGU_MLP_PRE_VARIANT == 1
        asm volatile("s_nop 7" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
#endif
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
#if PANGU_MLP_PRE_VARIANT == 1
        asm volatile("s_nop 7" : "+v"(v));
#endif
        __builtin_amdgcn_raw_buffer_store_b128(v, p_rsrc, (int)pre_off, tt * pre_tile + (chunk * 32 + 16 * qp) * 2, PANGU_MLP_NT_SIDE);
        // Write-after-read hazard the compiler does not cover (ROCm 7.2, gfx950): a 16-B buffer store reads its data
        // registers over several cycles, and hipcc pads a following VALU write of those registers (s_nop) only when the
        // store's soffset is an immediate; with soffset in an SGPR (the steady-state loop here) the GELU's first
        // v_pk_mul_f32 re-used v[n:n+1] right behind the store and dword 1 of lanes 12-15 / 28-31 (+32) left with the
        // product's bits (found by the oracle test as isolated wrong elements in columns 18, 19, 26, 27 mod 32).
        asm volatile("s_nop 1" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // One iteration: G2 = second product of chunk ch-2 (operand hf), GE = GELU of chunk ch-1 (hg), G1 = first product of
  // chunk ch (-> hacc); I1 / I2: request W1 chunk ch+2 / W2 chunk ch.  All flags are compile-time (peeled prologue /
  // drain iterations), so the steady-state body is one basic block.
  auto iteration = [&](int ch, int a /* ch % 3 */, auto g2, auto ge, auto g1, auto i1, auto i2) {
    constexpr bool G2 = decltype(g2)::value, GE = decltype(ge)::value, G1 = decltype(g1)::value;
    constexpr bool I1 = decltype(i1)::value && !(ABL & 1), I2 = decltype(i2)::value && !(ABL & 1);
    const int a1 = a + 1 >= NS ? a + 1 - NS : a + 1, a2 = a + 2 >= NS ? a + 2 - NS : a + 2;
    const unsigned char* w1 = ring1 + a * WB;              // W1 chunk ch
    const unsigned char* w2 = ring2 + a1 * WB;             // W2 chunk ch-2  ((ch-2) % 3 == (ch+1) % 3)
    // GELU of hg, STAGE-major: op n = stage * E + element, stages t = x x; w = fma(t, c1, c0); z = x w; e = exp2(z);
    // d = 1 + e; r = rcp(d); x r.  Consecutive ops belong to different elements and an element's next stage comes E ops
    // (several MFMA steps) later: with one wave per SIMD a dependent VALU chain issued back to back stalls on every
    // result (that was 25 % of the wave's cycles, SQ_WAIT_INST_ANY, when each element's seven ops ran in a row).
    constexpr int NOPS = 7 * E, NSTEPS = 2 * RT + KS;
    auto gelu_ops = [&](int n0, int n1) {
#pragma unroll
      for (int n = n0; n < n1; ++n) {
        const int st = n / E, e = n % E;
        if (ABL & 2) continue;
        float& x = hg[e >> 4][e & 15];
        float& t = gt[e];
        if (st == 0) t = x * x;
        else if (st == 1) t = fmaf(t, -0.06940179f * 1.4426950408889634f, -1.60031416f * 1.4426950408889634f);
        else if (st == 2) t = x * t;
        else if (st == 3) t = __builtin_amdgcn_exp2f(t);
        else if (st == 4) t = 1.0f + t;
        else if (st == 5) t = __builtin_amdgcn_rcpf(t);
        else x = x * t;
      }
    };
    // step 0 (right behind the barrier, while the first fragment reads are in flight: ~250 cycles with nothing else to
    // issue) takes the first two stages of every element; the rest is spread evenly over the other steps
    constexpr int FRONT = 2 * E;
    auto gelu_step = [&](int g) {
      if (g > 0) gelu_ops(FRONT + (NOPS - FRONT) * (g - 1) / (NSTEPS - 1), FRONT + (NOPS - FRONT) * g / (NSTEPS - 1));
    };
#if PANGU_MLP_PRE_AT_TOP
    // TR: chunk ch-1's pre-activation leaves first, where the fewest registers are live (no fragments, no GELU temporaries)
    if constexpr (GE) {
      if (TR == 2) {
        store_pre(ch - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#endif
    // ---- second product of chunk ch-2: 2 RT steps (row tile, k-step) of T MFMAs, fragment reads PD2 steps ahead
    if constexpr (G2) {
      constexpr int NSTEP = 2 * RT;
      auto rd = [&](int st) {     // step st = 2 rt + s
        return *reinterpret_cast<const bf16x8*>(w2 + (((st & 1) * 2 + lh) * C + 32 * (st >> 1) + lr) * 16);
      };
      bf16x8 fa[NSTEP];
#pragma unroll
      for (int st = 0; st < PD2 && st < NSTEP; ++st) fa[st] = rd(st);
      if constexpr (GE) {
        __builtin_amdgcn_sched_barrier(0);
        if (TR == 2 && !PANGU_MLP_PRE_AT_TOP) store_pre(ch - 1);      // chunk ch-1's pre-activation leaves under the first reads' flight
        gelu_ops(0, FRONT);                                // under the first reads' flight
      }
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + PD2 < NSTEP) fa[st + PD2] = rd(st + PD2);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 8)) {
#pragma unroll
          for (int tt = 0; tt < T; ++tt)
            yacc[st >> 1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st], hf[st & 1][tt], yacc[st >> 1][tt], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(fa[st]), "v"(hf[0][0]));
        }
        if constexpr (GE) gelu_step(st);
        if (I2 && st % (NSTEP / LPS) == 0) issue_piece(1, ch, a, st / (NSTEP / LPS));
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
      if constexpr (GE) {
        if (TR == 2 && !PANGU_MLP_PRE_AT_TOP) store_pre(ch - 1);
        gelu_ops(0, FRONT + (NOPS - FRONT) * (2 * RT - 1) / (NSTEPS - 1));
      }
      if (I2) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) issue_piece(1, ch, a, i);
      }
    }
#ifdef PANGU_MLP_STAMP
    const unsigned long long t_mid = stamp();
    if (G2 && G1) st_b += t_mid - st_last;
#endif
    // ---- first product of chunk ch: KS steps of T MFMAs, fragment reads PD1 steps ahead; accumulators start from the
    // bias of this lane's hidden rows (i&3) + 8(i>>2) + 4lh
    if constexpr (G1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + 32 * ch + 8 * q + 4 * lh);
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          hacc[tt][4 * q] = bv[0]; hacc[tt][4 * q + 1] = bv[1]; hacc[tt][4 * q + 2] = bv[2]; hacc[tt][4 * q + 3] = bv[3];
        }
      }
      auto rd = [&](int ks) {
        return *reinterpret_cast<const bf16x8*>(w1 + lr * (2 * C) + (((2 * ks + lh) ^ fsw) << 4));
      };
      bf16x8 fa[KS];
#pragma unroll
      for (int ks = 0; ks < PD1 && ks < KS; ++ks) fa[ks] = rd(ks);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + PD1 < KS) fa[ks + PD1] = rd(ks + PD1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4)) {
#pragma unroll
          for (int tt = 0; tt < T; ++tt)
            hacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], xf[ks][tt], hacc[tt], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(fa[ks]));
        }
        if constexpr (GE) gelu_step(2 * RT + ks);
        if (I1 && ks % (KS / LPS) == 0) issue_piece(0, ch + 2, a2, ks / (KS / LPS));
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
      if constexpr (GE) gelu_ops(FRONT + (NOPS - FRONT) * (2 * RT - 1) / (NSTEPS - 1), NOPS);
    }
    // ---- rotate the pipeline registers: hf <- pack(GELU(chunk ch-1)), hg <- chunk ch
    if constexpr (GE) {
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int sk = 0; sk < 2; ++sk)
          hf[sk][tt] = __builtin_bit_cast(
              bf16x8, u32x4{pack_bf16x2(hg[tt][8 * sk], hg[tt][8 * sk + 1]), pack_bf16x2(hg[tt][8 * sk + 2], hg[tt][8 * sk + 3]),
                            pack_bf16x2(hg[tt][8 * sk + 4], hg[tt][8 * sk + 5]), pack_bf16x2(hg[tt][8 * sk + 6], hg[tt][8 * sk + 7])});
    }
    if constexpr (G1) {
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) hg[tt][i] = hacc[tt][i];
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
#ifdef PANGU_MLP_STAMP
  const unsigned long long st_pro = stamp() - st_begin;
#endif
  iteration(0, 0, N_{}, N_{}, Y{}, Y{}, Y{});
  sync(Int<LPW>{});
  iteration(1, 1, N_{}, Y{}, Y{}, Y{}, Y{});
  int a = 2;
  for (int ch = 2; ch < NCH - 2; ++ch) {
#ifdef PANGU_MLP_STAMP
    const unsigned long long t0 = stamp();
#endif
    sync(Int<LPW>{});
#ifdef PANGU_MLP_STAMP
    st_last = stamp();
    st_sync += st_last - t0;
#endif
    iteration(ch, a, Y{}, Y{}, Y{}, Y{}, Y{});
#ifdef PANGU_MLP_STAMP
    st_all += stamp() - t0;
    st_n += 1;
#endif
    a = a + 1 == NS ? 0 : a + 1;
  }
  // ---- drain: chunks NCH-2, NCH-1 request no W1 any more; then two iterations without a first product
  // (NW = 8: the two request streams sit on different waves; the drain simply waits for everything)
  sync(Int<LPW>{});
  iteration(NCH - 2, (NCH - 2) % NS, Y{}, Y{}, Y{}, N_{}, Y{});
  sync(Int<NW == 4 ? LPS : 0>{});
  iteration(NCH - 1, (NCH - 1) % NS, Y{}, Y{}, Y{}, N_{}, Y{});
  sync(Int<NW == 4 ? LPS : 0>{});
  iteration(NCH, NCH % NS, Y{}, Y{}, N_{}, N_{}, N_{});
  sync(Int<0>{});
  iteration(NCH + 1, (NCH + 1) % NS, Y{}, N_{}, N_{}, N_{}, N_{});
  __syncthreads();                               // every wave is done with the rings: the patches may reuse them
#ifdef PANGU_MLP_STAMP
  const unsigned long long st_epi0 = stamp();
#endif

  // ---- epilogue: lane holds Y^T[c = 32rt + (i&3) + 8(i>>2) + 4lh][token m0 + 32tt + lr]
  unsigned char* patch = ring1 + wave * (32 * PLD);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      Out, 0, (int)(((size_t)(M - 1) * ldo + C) * sizeof(u16)), 0x00020000);
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    const int tok0 = m0 + 32 * tt;
    // The shortcut IS this wave's input tile, still in registers as the first product's B fragments: lane (token, h) holds
    // channels 16ks + 8h .. +7, the accumulator layout wants 32rt + 8q + 4h .. +3 -- half of which the partner lane (h ^ 1,
    // same token) holds: two v_permlane32_swap per fragment and (rt, q) reads registers {2(q&1), 2(q&1)+1} of fragment
    // 2rt + (q>>1) in BOTH halves.  (Re-loading the tile instead made every CU read 100 KB at the same moment: the kernel
    // runs its tiles in lockstep, so the epilogue was a chip-wide 25 MB burst, 38k of 175k cycles per tile.)
    u32x4 sc[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 f = __builtin_bit_cast(u32x4, xf[ks][tt]);
      const auto r02 = __builtin_amdgcn_permlane32_swap(f[0], f[2], false, false);
      const auto r13 = __builtin_amdgcn_permlane32_swap(f[1], f[3], false, false);
      sc[ks] = u32x4{r02[0], r13[0], r02[1], r13[1]};
    }
    if (TR) {
      // pre-LayerNorm rows (second product + b2) -> Mo, through the same per-wave patch as the result below
      const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          Mo, 0, (int)(((size_t)(M - 1) * ldm + C) * sizeof(u16)), 0x00020000);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<u32x2*>(patch + lr * PLD + (32 * rt + 8 * q + 4 * lh) * 2) =
              u32x2{pack_bf16x2(yacc[rt][tt][4 * q], yacc[rt][tt][4 * q + 1]),
                    pack_bf16x2(yacc[rt][tt][4 * q + 2], yacc[rt][tt][4 * q + 3])};
#pragma unroll
      for (int it = 0; it < 32 * CPR / 64; ++it) {
        const int f = lane + 64 * it, row = f / CPR, chk = f - row * CPR;
        const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * PLD + chk * 16);
        const unsigned off = tok0 + row < M ? ((unsigned)(tok0 + row) * (unsigned)ldm + chk * 8) * 2u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128(v, m_rsrc, (int)off, 0, PANGU_MLP_NT_SIDE);
      }
    }
    // LayerNorm statistics: four independent partial sums (a single dependent add chain stalls the lone wave on every add)
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) s4[i & 3] += yacc[rt][tt][i];
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = yacc[rt][tt][i] - mean;
        q4[i & 3] = fmaf(d, d, q4[i & 3]);
      }
    float ss = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = rsqrtf(ss * (1.0f / C) + LN_EPS);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * rt + 8 * q + 4 * lh;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(eps + C + c0);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(eps + 2 * C + c0);
        unsigned char* slotp = patch + lr * PLD + c0 * 2;
        const u32x2 xs = u32x2{sc[2 * rt + (q >> 1)][2 * (q & 1)], sc[2 * rt + (q >> 1)][2 * (q & 1) + 1]};
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaf((yacc[rt][tt][4 * q + r] - mean) * rstd, gm[r], bt[r]) * scale;
        o[0] += bflo(xs[0]); o[1] += bfhi(xs[0]); o[2] += bflo(xs[1]); o[3] += bfhi(xs[1]);
        *reinterpret_cast<u32x2*>(slotp) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
      }
#pragma unroll
    for (int it = 0; it < 32 * CPR / 64; ++it) {
      const int f = lane + 64 * it, row = f / CPR, chk = f - row * CPR;
      const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * PLD + chk * 16);
      const unsigned off = tok0 + row < M ? ((unsigned)(tok0 + row) * (unsigned)ldo + chk * 8) * 2u : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(v, o_rsrc, (int)off, 0, PANGU_MLP_NT_OUT);
    }
  }
#ifdef PANGU_MLP_STAMP
  if (lane == 0) {
    atomicAdd(&g_stamp[0], st_sync); atomicAdd(&g_stamp[1], st_b);
    atomicAdd(&g_stamp[2], st_all - st_sync - st_b); atomicAdd(&g_stamp[3], st_n);
    const unsigned long long t_end = stamp();
    atomicAdd(&g_stamp[4], t_end - st_begin); atomicAdd(&g_stamp[5], 1ull);
    atomicAdd(&g_stamp[6], st_pro); atomicAdd(&g_stamp[7], t_end - st_epi0);
  }
#endif
}

template <int C, int T, int NW, int TR = 0>
int launch_mlp(hipStream_t s, const u16* x, int ldx, const u16* wimg, const float* b1, const float* b2,
               const float* gamma, const float* beta, u16* out, int ldo, int M, float scale, u16* pre = nullptr,
               int ldp = 0, u16* mo = nullptr, int ldm = 0) {
  constexpr size_t ring = (size_t)6 * 64 * C;
  constexpr size_t epi = (size_t)NW * 32 * (2 * C + 16);
  constexpr size_t shm = 7 * C * sizeof(float) + (ring > epi ? ring : epi);
  static_assert(shm <= 160 * 1024, "LDS");
  auto kern = mlp_ln_residual_bf16_kernel<C, T, NW, TR>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  const int grid = (M + 32 * T * NW - 1) / (32 * T * NW);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), shm, s, x, ldx, wimg, b1, b2, gamma, beta, out, ldo, M, scale, pre,
                     ldp, mo, ldm);
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
  if (C == 192) {
    // 4 waves x 64 tokens (one wave per SIMD, 512 registers); 8 waves x 32 tokens (two per SIMD, 256 registers) measured the same
    // wall time and was removed in round 4
    return launch_mlp<192, 2, 4>(s, (const u16*)x, ldx, (const u16*)w_packed, b1, b2, gamma, beta, (u16*)out, ldo, M,
                                 branch_scale);
  }
  if (C == 384)
    return launch_mlp<384, 1, 4>(s, (const u16*)x, ldx, (const u16*)w_packed, b1, b2, gamma, beta, (u16*)out, ldo, M,
                                 branch_scale);
  return PANGU_E_SHAPE;
}

// Training forward of the same branch: additionally writes what the backward needs (see the kernel's TR note):
// pre (M x 4C bf16, row stride ldp; may be null = recompute mode) and m (M x C bf16, the pre-LayerNorm value, row stride ldm).
extern "C" int pangu_mlp_ln_residual_train_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_packed,
                                                    const float* b1, const float* b2, const float* gamma,
                                                    const float* beta, void* out, int ldo, void* pre, int ldp, void* m,
                                                    int ldm, int M, int C, float branch_scale) {
  if (!x || !w_packed || !b1 || !b2 || !gamma || !beta || !out || !m || !pre) return PANGU_E_NULL;
  if (M <= 0 || ldx < C || ldo < C || ldm < C || (ldx & 7) || (ldo & 7) || (ldm & 7)) return PANGU_E_SHAPE;
  if (ldp < 4 * C || (ldp & 7)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, ldx, 2) || !pangu_fits_u32(M, ldo, 2) || !pangu_fits_u32(M, ldm, 2) || !pangu_fits_u32(M, ldp, 2))
    return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
#define PANGU_MLP_TR(CC, TT, MODE)                                                                                        \
  launch_mlp<CC, TT, 4, MODE>(s, (const u16*)x, ldx, (const u16*)w_packed, b1, b2, gamma, beta, (u16*)out, ldo, M,        \
                              branch_scale, (u16*)pre, ldp, (u16*)m, ldm)
  if (C == 192) return PANGU_MLP_TR(192, 2, 2);
  if (C == 384) return PANGU_MLP_TR(384, 1, 2);
#undef PANGU_MLP_TR
  return PANGU_E_SHAPE;
}

#ifdef PANGU_MLP_STAMP
extern "C" int pangu_mlp_stamp_read(unsigned long long* out8) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8);
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z));
  return 0;
}
#endif
