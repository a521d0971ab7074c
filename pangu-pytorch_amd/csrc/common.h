// Shared device helpers for the Pangu-Weather gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pangu_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PANGU_WZ 2
#define PANGU_WH 6
#define PANGU_WW 12
#define PANGU_WTOK 144
#define PANGU_HEAD_DIM 32
#define PANGU_PAD_H 5

// Geometry of one attention stage: grid (Z,H,W), padded latitude Hp = H+5, windows (2,6,12).
struct WinGeom {
  int Z, H, W, Hp, nLon, nZw, nHw, types;
};

__host__ __device__ inline WinGeom make_geom(int Z, int H, int W) {
  WinGeom g;
  g.Z = Z; g.H = H; g.W = W; g.Hp = H + PANGU_PAD_H;
  g.nLon = W / PANGU_WW; g.nZw = Z / PANGU_WZ; g.nHw = g.Hp / PANGU_WH; g.types = g.nZw * g.nHw;
  return g;
}

// Flat token index (z*H+h)*W+w that feeds window slot (l, t, n), or -1 for a zero-pad slot.
// Folds view/pad/roll/partition (reference models/layers.py:188-221) into address arithmetic;
// the same token receives the attention output (reverse/roll-back/crop, layers.py:227-247).
__host__ __device__ inline int win_src_token(const WinGeom& g, int l, int t, int n, int shifted) {
  int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
  int zi = n / 72, r = n - zi * 72, hi = r / 12, wi = r - hi * 12;
  int z = 2 * zwin + zi, h = 6 * hwin + hi, w = 12 * l + wi;
  if (shifted) {
    z += 1; if (z >= g.Z) z -= g.Z;
    h += 3; if (h >= g.Hp) h -= g.Hp;
    w += 6; if (w >= g.W) w -= g.W;
  }
  return h >= g.H ? -1 : (z * g.H + h) * g.W + w;
}

// Shifted-window mask (reference models/layers.py:153-181) in closed form: -100 or 0.
__host__ __device__ inline float win_mask(const WinGeom& g, int t, int ni, int nj) {
  int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
  bool zcut = (zwin == g.nZw - 1) && ((ni / 72) != (nj / 72));
  bool hcut = (hwin == g.nHw - 1) && ((((ni / 12) % 6) < 3) != (((nj / 12) % 6) < 3));
  return (zcut || hcut) ? -100.0f : 0.0f;
}

// erf(x), |abs err| < 1.5e-7 (Abramowitz-Stegun 7.1.26), branch-free: 1 rcp + 1 exp + 6 fma.  The exact-erf GELU
// of the reference (nn.GELU(), layers.py:261) is reproduced to ~4e-7 absolute, i.e. at fp32 rounding level of
// the O(1) activations, at a third of the instruction count of the libm erff (which branches per lane).
__device__ inline float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp), not an IEEE division
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// bf16-path variant: erf(x) ~ clamp(x P(x^2)) with a degree-6 minimax P on |x| <= 2.8 (|err| < 1.9e-4, GELU abs error
// < 3.7e-4: below bf16 output rounding), 8 FMAs and no transcendental: the bf16 GEMM epilogue is VALU-bound otherwise.
__device__ inline float erf_poly(float x) {
  const float ax = fminf(fabsf(x), 2.8f);
  const float t = ax * ax;
  float p = fmaf(5.37784878e-06f, t, -0.000177775825f);
  p = fmaf(p, t, 0.00251051432f);
  p = fmaf(p, t, -0.0201338951f);
  p = fmaf(p, t, 0.103594314f);
  p = fmaf(p, t, -0.370308868f);
  p = fmaf(p, t, 1.12730194f);
  return copysignf(fminf(p * ax, 1.0f), x);
}
__device__ inline float gelu_erf_lp(float x) { return 0.5f * x * (1.0f + erf_poly(x * 0.70710678118654752440f)); }

// The same GELU on a float4 with gfx950's packed-fp32 VALU ops (v_pk_mul/fma/add_f32: two lanes-elements per
// instruction): gelu(x) = x (0.5 + h(x/sqrt2)), h = erf/2 = clamp(u P'(u^2), -.5, .5) with P' = P/2 and u clamped to
// +-2.8 (odd in u: no abs / copysign).  7.5 VALU instructions per element instead of 15.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ inline f32x2_t half_erf_poly2(f32x2_t u) {
  f32x2_t uc;
  uc[0] = __builtin_amdgcn_fmed3f(u[0], -2.8f, 2.8f);
  uc[1] = __builtin_amdgcn_fmed3f(u[1], -2.8f, 2.8f);
  const f32x2_t t = uc * uc;
#define PANGU_C2(v) f32x2_t{v, v}
  f32x2_t p = __builtin_elementwise_fma(PANGU_C2(0.5f * 5.37784878e-06f), t, PANGU_C2(0.5f * -0.000177775825f));
  p = __builtin_elementwise_fma(p, t, PANGU_C2(0.5f * 0.00251051432f));
  p = __builtin_elementwise_fma(p, t, PANGU_C2(0.5f * -0.0201338951f));
  p = __builtin_elementwise_fma(p, t, PANGU_C2(0.5f * 0.103594314f));
  p = __builtin_elementwise_fma(p, t, PANGU_C2(0.5f * -0.370308868f));
  p = __builtin_elementwise_fma(p, t, PANGU_C2(0.5f * 1.12730194f));
#undef PANGU_C2
  f32x2_t e = p * uc;
  e[0] = __builtin_amdgcn_fmed3f(e[0], -0.5f, 0.5f);
  e[1] = __builtin_amdgcn_fmed3f(e[1], -0.5f, 0.5f);
  return e;
}
__device__ inline f32x4 gelu_erf_lp4(f32x4 x) {
  const f32x2_t c = {0.70710678118654752440f, 0.70710678118654752440f}, half = {0.5f, 0.5f};
  f32x2_t a = {x[0], x[1]}, b = {x[2], x[3]};
  a = a * (half_erf_poly2(a * c) + half);
  b = b * (half_erf_poly2(b * c) + half);
  return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ inline float gelu_erf_grad_lp(float x) {
  return 0.5f * (1.0f + erf_poly(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ inline float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// two floats -> two bf16 (round-to-nearest-even) in ONE v_cvt_pk_bf16_f32, low half = a
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ inline unsigned pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// Sum over the 16 lanes of a DPP row, result in every lane: four v_add_f32 with DPP operand swizzles (quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) -- no LDS crossbar round trips (ds_bpermute) for these steps.
__device__ inline float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
// Sum over aligned groups of LPR lanes (16, 32 or 64), result in every lane of the group
template <int LPR>
__device__ inline float group_sum(float v) {
  v = row16_sum(v);
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ inline float wave_sum(float v) { return group_sum<64>(v); }

// hipFuncSetAttribute is per (function, DEVICE): remember it per device so that a process driving several GPUs
// (or a model moved to another GPU) works, while the steady state stays one thread-local hipGetDevice per launch.
// Runs on the first launch per device, i.e. during warm-up, outside any later graph capture.
static inline void pangu_ensure_dyn_lds(const void* kern, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(*done_mask & bit)) {
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    *done_mask |= bit;
  }
}
#define PANGU_ENSURE_DYN_LDS(kern, bytes)                                                      \
  do {                                                                                         \
    static unsigned long long pangu_lds_done_ = 0ull;                                          \
    pangu_ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)(bytes), &pangu_lds_done_); \
  } while (0)

// The GEMM-family kernels address their operands through buffer descriptors with 32-bit BYTE offsets (that is what
// makes the range-checked, branch-free tails possible): a matrix whose last row starts at or beyond 4 GB cannot be
// addressed and is refused with PANGU_E_RANGE (the Python layer splits such calls by rows, ops._row_chunks).
static inline bool pangu_fits_u32(long long rows, long long ld, int elem_bytes) {
  return (rows + 256) * ld * elem_bytes < 0xFFFFFFFFll;
}

static inline int pangu_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PANGU_OK : (int)e;
}
