// The whole MLP branch of an Earth-specific block in ONE fp32 launch (inference):
//     out = x + s * (LayerNorm(GELU(x W1^T + b1) W2^T + b2) * gamma + beta)          reference layers.py:251 with :264-270 inside
// on true fp32 MFMA (v_mfma_f32_32x32x2_f32).  The (tokens x 4C) hidden activation never reaches memory: unfused it is written by
// the MLP-up launch and read back by the MLP-down launch, 3.2 GB per stage-0 block and 1.6 GB per stage-1/2 block, whose HBM time
// the two MFMA-bound launches only half hide (K = 192 launches at 0.74-0.76 of the fp32 roof).
//
// Shape of the computation (the fp32 twin of mlp_fused_bf16.hip): a workgroup = 4 waves, one per SIMD with up to 512 registers;
// a wave owns 32 tokens.  Both products run TRANSPOSED with the weights as the A operand:
//   H^T [32 hidden][32 tokens] = W1[chunk rows][C] . x^T        (C/2 MFMAs into ONE 16-register accumulator, initial value b1)
//   Y^T [C][32 tokens]        += W2[:, chunk] . GELU(H^T)       (16 MFMAs per 32-channel output tile; C/2 accumulator registers)
// The first product's accumulator layout IS the second product's B operand: lane (token t, half h) holds hidden 8(r/4) + 4h + r%4 in
// register r, so register r feeds the k pair (8(r/4) + r%4, + 4) of one MFMA -- nothing crosses lanes or LDS between the products.
// The input tile lives in registers in the SAME channel <-> register map as Y^T (register 4q + e of k-slab j <-> channel
// 32j + 8q + 4h + e; a dot product does not see the k order), so the shortcut add and the LayerNorm work register against register
// and x is loaded once.
// Weights stream L2 -> LDS by LDS-DMA straight from the parameters' own (out, in) row-major layouts: a SLAB = 32 rows x 32 k fp32
// (4 KB; W1: 32 hidden rows x a 32-channel k range, W2: 32 output channels x the 32 hidden of a chunk), its 16-B pieces XOR-swizzled on
// the source side with (row >> 1) & 7 (conflict-free ds_read_b128 under the real lane groups); a lane's four b128 reads of a slab are
// the A operands of the slab's 16 MFMAs.  A STAGE = 6 slabs (24 KB, 96 MFMAs = 6 144 matrix cycles per wave), ring of 4 stages, one
// barrier per stage placed in the MIDDLE of the previous stage (no wait at a stage boundary), requests two stages ahead.  The GELU of chunk c (exact-erf form, common.h) is issued between the MFMAs of the
// first product of chunk c + 1.
#include "common.h"
#include <type_traits>

namespace {

constexpr int SLAB = 32 * 128;            // bytes
constexpr int STAGE_SLABS = 6;
constexpr int STAGE = STAGE_SLABS * SLAB; // 24 576 B
constexpr int RING = 4;
constexpr float LN_EPS = 1e-5f;
// timing-only ablations (tools/ablate_mlp_f32.sh builds them into scratch/; never in the shipped library)
#ifndef MLP_PIN
#define MLP_PIN 1          // scheduling barrier after every MFMA gap
#endif
#ifndef MLP_NOGELU
#define MLP_NOGELU 0       // 1: skip the GELU (wrong results)
#endif
#ifndef MLP_NODMA
#define MLP_NODMA 0        // 1: no in-loop LDS-DMA requests and no barriers (stale LDS: wrong results)
#endif
#ifndef MLP_DUAL
#define MLP_DUAL 0         // 1: the first product alternates two accumulators (independent MFMA chains), summed per chunk
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int C>
__global__ __launch_bounds__(256) void mlp_ln_residual_f32_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ out, int ldo,
                                                                  int M, float branch_scale) {
  constexpr int HID = 4 * C;
  constexpr int NCH = HID / 32;             // hidden chunks
  constexpr int KJ = C / 32;                // k-slabs of the first product = output tiles of the second
  constexpr int SPP = KJ / STAGE_SLABS;     // stages per product and chunk (1 at C = 192, 2 at C = 384)
  static_assert(KJ % STAGE_SLABS == 0, "C must be a multiple of 192");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ring = smem;                                               // RING stages
  float* const b1_s = reinterpret_cast<float*>(smem + RING * STAGE);              // [4C]
  float* const b2_s = b1_s + HID;                                                 // [C]
  float* const g_s = b2_s + C;                                                    // gamma [C]
  float* const be_s = g_s + C;                                                    // beta [C]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128 + wave * 32;

  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(((size_t)(M - 1) * ldx + C) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(((size_t)(M - 1) * ldo + C) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w1), 0, HID * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w2), 0, HID * C * 4, 0x00020000);

  // ---- the wave's 32 input rows, once, in the channel <-> register map of the accumulators (16-B pieces; rows >= M: zeros)
  f32x4 xr[KJ][4];
  {
    const unsigned row = (unsigned)(m0 + t);
    const unsigned base = row < (unsigned)M ? (row * (unsigned)ldx + 4u * h) * 4u : 0x7FFFFFF0u;
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xr[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(base + (32 * j + 8 * q) * 4), 0, 0));
  }
  for (int i = tid; i < HID; i += 256) b1_s[i] = b1[i];
  for (int i = tid; i < C; i += 256) {
    b2_s[i] = b2[i];
    g_s[i] = gamma[i];
    be_s[i] = beta[i];
  }

  // ---- LDS-DMA plan.  Wave w carries rows 8w .. 8w+7 of every slab: this lane fills (row 8w + lane/8, physical piece lane%8) with
  // the logical piece (lane % 8) ^ ((row >> 1) & 7).  Stage n of the stream: chunk c = n / (2 SPP), part = n % (2 SPP);
  // part < SPP: W1 rows 32c.., k-slabs 6 part ..;  else: W2 output tiles 6 (part - SPP) .., hidden 32c ..
  const int drow = 8 * wave + (lane >> 3);
  const int dpc = (lane & 7) ^ ((drow >> 1) & 7);
  const unsigned voff1 = ((unsigned)drow * C + 4u * dpc) * 4u;          // within W1: + (32c C + 32 j) floats
  const unsigned voff2 = ((unsigned)drow * HID + 4u * dpc) * 4u;        // within W2: + (32 o HID + 32c) floats
  constexpr int NSTAGE = NCH * 2 * SPP;
  // The weight stream is a flat list of stages in the order they are consumed:  P1(0) | for c: { P1(c+1) (if c+1 < NCH), P2(c) }.
  // One LDS-DMA piece (1 KB: this wave's 8 rows of slab k) of stage n:
  auto issue_piece = [&](int n, int k) {
    int kind, c, part;
    if (n < SPP) { kind = 0; c = 0; part = n; }
    else {
      const int m = n - SPP;
      const int full = (NCH - 1) * 2 * SPP;      // every chunk but the last: P1(c+1) parts then P2(c) parts; the last chunk: P2 only
      if (m < full) {
        c = m / (2 * SPP);
        const int r = m % (2 * SPP);
        if (r < SPP) { kind = 0; c = c + 1; part = r; }
        else { kind = 1; part = r - SPP; }
      } else {
        kind = 1; c = NCH - 1; part = m - full;
      }
    }
    auto dst = (__attribute__((address_space(3))) void*)(ring + (n & (RING - 1)) * STAGE + k * SLAB + wave * 1024);
    const int jo = part * STAGE_SLABS + k;
    if (kind == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, dst, 16, (int)voff1, (32 * c * C + 32 * jo) * 4, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, dst, 16, (int)voff2, (32 * jo * HID + 32 * c) * 4, 0, 0);
  };

  // fragment reads of one slab: A operands of its 16 MFMAs (lane (row i = t, half h): pieces 2q + h, q = 0..3)
  const int frow = t * 128;
  const int fsw = (t >> 1) & 7;
  auto frag1 = [&](const unsigned char* slab, int q) {
    return *reinterpret_cast<const f32x4*>(slab + frow + (((2 * q + h) ^ fsw) << 4));
  };

  // ---- accumulators
  f32x16 Y[KJ];
  __syncthreads();                                                       // the parameter vectors are in LDS
#pragma unroll
  for (int o = 0; o < KJ; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b2_s + 32 * o + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) Y[o][4 * q + e] = bv[e];
    }

  // Ring protocol (RING = 4 slots; stage s lives in slot s % 4).  sync(s) runs in the MIDDLE of stage s - 1 (all four waves are
  // then past stage s - 2): wait for this wave's own pieces of stage s (its pieces of stage s + 1 stay in flight), barrier -> stage
  // s is readable by every wave; the six pieces of stage s + 2 are then requested into the slot of stage s - 2, ONE per MFMA gap.  At
  // a stage boundary nothing waits: the first fragments of stage s are read under the last MFMAs of stage s - 1.
  auto sync = [&](int s) {
    if (s >= NSTAGE) return;
    if (s + 1 < NSTAGE) wait_vmcnt<STAGE_SLABS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x4 a[2][4];                 // A operands of the slab in flight / the next one (16 MFMAs each)
  f32x16 Hd;                     // (MLP_DUAL ablation: second accumulator of the first product)
#pragma unroll
  for (int r = 0; r < 16; ++r) Hd[r] = 0.f;
  f32x16 Hc, G;                  // Hc: first-product accumulator of the chunk in flight; G: the previous chunk, GELU'd IN PLACE
  auto h_init = [&](int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1_s + 32 * c + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) Hc[4 * q + e] = bv[e];
    }
  };
  // The exact-erf GELU of common.h (gelu_erf: Abramowitz-Stegun 7.1.26) cut into three groups of five VALU operations, so that one
  // group fits the shadow of one 64-cycle MFMA; (gu, gt, ga) carry a value between its groups.
  float gu, gt, ga;
  auto gelu_group = [&](int v, int grp) {
    // (the empty asm statements pin each group between the scheduling barriers of its MFMA gap: the values are pure arithmetic
    // whose only use is a whole product later, and the optimiser would otherwise gather all 16 chains in one place)
    if (grp == 0) {
      asm volatile("" : "+v"(G[v]));
      gu = G[v] * 0.70710678118654752440f;
      const float ax = fabsf(gu);
      gt = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
      ga = (ax * ax) * -1.4426950408889634f;                         // exp(-ax^2) = exp2(-ax^2 log2 e)
      ga = __builtin_amdgcn_exp2f(ga);
      asm volatile("" : "+v"(gu), "+v"(gt), "+v"(ga));
    } else if (grp == 1) {
      float p = fmaf(1.061405429f, gt, -1.453152027f);
      p = fmaf(p, gt, 1.421413741f);
      p = fmaf(p, gt, -0.284496736f);
      p = fmaf(p, gt, 0.254829592f);
      gt = p * gt;
      asm volatile("" : "+v"(gt));
    } else {
      const float r = fmaf(-gt, ga, 1.0f);
      const float er = copysignf(r, gu);
      const float hx = 0.5f * G[v];
      G[v] = fmaf(hx, er, hx);
      asm volatile("" : "+v"(G[v]));
    }
  };
  // one stage (6 slabs = 96 MFMAs) of the stream, stage number s (runtime); KIND 0: first product into Hc (k-slabs 6 part ..; with
  // gelu_prev the GELU of G runs in its MFMA gaps), KIND 1: second product from G into Y (output tiles 6 part ..).  Every MFMA is
  // followed by AT MOST one small group of other work and a scheduling barrier, so the matrix pipe sees back-to-back issue:
  //   MFMA 16k + q (q < 4): one of the four b128 fragment reads of the NEXT slab;  MFMA 48: sync(s + 1);  MFMAs 49..54: one LDS-DMA
  //   piece of stage s + 2 each;  even MFMAs of a gelu stage: one GELU group
  auto run_stage = [&](auto kind_tag, int part, int s, bool gelu_prev) {
    constexpr int KIND = decltype(kind_tag)::value;
    const unsigned char* base = ring + (s & (RING - 1)) * STAGE;
    const unsigned char* next = ring + ((s + 1) & (RING - 1)) * STAGE;
#pragma unroll
    for (int k = 0; k < STAGE_SLABS; ++k) {
      const int jo = part * STAGE_SLABS + k;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = k * 16 + q * 4 + e;
          if (KIND == 0) {
            if (MLP_DUAL && (m & 1)) Hd = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 1][q][e], xr[jo][q][e], Hd, 0, 0, 0);
            else Hc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 1][q][e], xr[jo][q][e], Hc, 0, 0, 0);
          } else Y[jo] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 1][q][e], G[4 * q + e], Y[jo], 0, 0, 0);
          if (m % 16 < 4) {                        // the next slab's fragments, one read per gap
            if (k + 1 < STAGE_SLABS) a[(k + 1) & 1][m % 16] = frag1(base + (k + 1) * SLAB, m % 16);
            else if (s + 1 < NSTAGE) a[0][m % 16] = frag1(next, m % 16);
          }
          if (!MLP_NODMA) {
            if (m == 48) sync(s + 1);
            if (m > 48 && m <= 48 + STAGE_SLABS && s + 2 < NSTAGE) issue_piece(s + 2, m - 49);
          }
          if (!MLP_NOGELU && KIND == 0 && gelu_prev && part == 0 && (m & 1) == 0 && m / 2 < 48) gelu_group(m / 6, (m / 2) % 3);
          if (MLP_PIN) __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;

#pragma unroll
  for (int k = 0; k < STAGE_SLABS; ++k) issue_piece(0, k);
  if (NSTAGE > 1) {
#pragma unroll
    for (int k = 0; k < STAGE_SLABS; ++k) issue_piece(1, k);
  }
  sync(0);
  if (NSTAGE > 2) {
#pragma unroll
    for (int k = 0; k < STAGE_SLABS; ++k) issue_piece(2, k);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) a[0][q] = frag1(ring, q);
  int sn = 0;                                                            // next stage of the stream
  // ---- P1(0)
  h_init(0);
#pragma unroll
  for (int p = 0; p < SPP; ++p) run_stage(K0{}, p, sn++, false);
  for (int c = 0; c < NCH; ++c) {
    G = Hc;                                                              // chunk c before GELU
    if (MLP_DUAL) {
      G += Hd;
#pragma unroll
      for (int r = 0; r < 16; ++r) Hd[r] = 0.f;
    }
    if (c + 1 < NCH) {
      h_init(c + 1);
#pragma unroll
      for (int p = 0; p < SPP; ++p) run_stage(K0{}, p, sn++, true);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) G[r] = MLP_NOGELU ? G[r] : gelu_erf(G[r]);
    }
#pragma unroll
    for (int p = 0; p < SPP; ++p) run_stage(K1{}, p, sn++, false);
  }

  // ---- epilogue: LayerNorm over the row (this lane's C/2 channels + the partner half's), residual, 16-B stores
  float s1 = 0.f;
#pragma unroll
  for (int o = 0; o < KJ; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) s1 += Y[o][r];
  s1 += __shfl_xor(s1, 32, 64);
  const float mean = s1 * (1.0f / C);
  float s2 = 0.f;
#pragma unroll
  for (int o = 0; o < KJ; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = Y[o][r] - mean;
      s2 = fmaf(d, d, s2);
    }
  s2 += __shfl_xor(s2, 32, 64);
  const float rstd = rsqrtf(s2 * (1.0f / C) + LN_EPS);
  {
    const unsigned row = (unsigned)(m0 + t);
    const unsigned base = row < (unsigned)M ? (row * (unsigned)ldo + 4u * h) * 4u : 0x7FFFFFF0u;
#pragma unroll
    for (int o = 0; o < KJ; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(g_s + 32 * o + 8 * q + 4 * h);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(be_s + 32 * o + 8 * q + 4 * h);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = xr[o][q][e] + branch_scale * ((Y[o][4 * q + e] - mean) * rstd * gm[e] + bt[e]);
        // (the column offset rides in the instruction's immediate field: a 16-B store with an SGPR soffset directly followed by a
        // VALU write of its data registers loses dword 1 of lanes 12-15 / 28-31 on gfx950 -- the compiler hazard of DESIGN section 0b)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, (int)(base + (32 * o + 8 * q) * 4), 0, 0);
        asm volatile("s_nop 1" ::: "memory");
      }
  }
}

template <int C>
int launch_mlp(hipStream_t s, const float* x, int ldx, const float* w1, const float* b1, const float* w2, const float* b2,
               const float* gamma, const float* beta, float* out, int ldo, int M, float branch_scale) {
  const size_t shm = (size_t)RING * STAGE + (size_t)(4 * C + 3 * C) * sizeof(float);
  auto kern = mlp_ln_residual_f32_kernel<C>;
  PANGU_ENSURE_DYN_LDS(kern, shm);
  hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), shm, s, x, ldx, w1, b1, w2, b2, gamma, beta, out, ldo, M, branch_scale);
  return pangu_launch_status();
}

}  // namespace

extern "C" int pangu_mlp_ln_residual_fwd(pangu_stream_t stream, const float* x, int ldx, const float* w1, const float* b1,
                                         const float* w2, const float* b2, const float* gamma, const float* beta, float* out,
                                         int ldo, int M, int C, float branch_scale) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !gamma || !beta || !out) return PANGU_E_NULL;
  // C = 192 only (the stage-0 / stage-3 blocks, whose two K = 192 launches run at 0.74-0.76 of the fp32 roof).  At C = 384 a wave's
  // input tile alone is 192 of the 256 architectural VGPRs (the C = 384 instantiation spills ~120 registers and measured 0.55 of
  // the roof against 0.87 for the two launches it would replace, profiles/r05_mlp_f32_ab.md), so stage 1 / 2 keeps them.
  if (M <= 0 || C != 192 || ldx < C || (ldx & 3) || ldo < C || (ldo & 3)) return PANGU_E_SHAPE;
  if (!pangu_fits_u32(M, ldx, 4) || !pangu_fits_u32(M, ldo, 4) || (size_t)M * ldx * 4 >= 0x7FFFFFF0ull ||
      (size_t)M * ldo * 4 >= 0x7FFFFFF0ull)
    return PANGU_E_RANGE;
  hipStream_t s = (hipStream_t)stream;
  return launch_mlp<192>(s, x, ldx, w1, b1, w2, b2, gamma, beta, out, ldo, M, branch_scale);
}
