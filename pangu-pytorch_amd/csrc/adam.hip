// Adam for the whole model in ONE launch (reference finetune_fully.py:121: torch.optim.Adam(lr=5e-6, weight_decay=3e-6); the step of
// models/pangu_sample.py:75).  torch's fused Adam walks the 223 tensors in 14 launches of at most 320 workgroups each (its tensor-list
// metadata caps a launch): 1.8 ms for 7.7 GB = 4.3 TB/s, about one workgroup per CU.  Here a device-resident job table describes
// every (param, grad, exp_avg, exp_avg_sq) quadruple and one launch of ~67 000 workgroups streams them.
//
// The arithmetic follows ATen/native/cuda/fused_adam_utils.cuh (adam_math, ADAM_MODE::ORIGINAL, no amsgrad / maximize / grad
// scaling) operation by operation, INCLUDING its mixed precision -- lr, betas, weight_decay, eps are doubles there, so the decay
// term, both moment updates, the step size and the denominator are evaluated in double and rounded to float where that code assigns
// to its float variables -- so the result is bit-identical to torch.optim.Adam(fused=True) (tests/test_gpu_extras.py).
//
// Job table: (n_jobs + 1) rows of 8 int64: [0] param [1] grad (0 = a zero gradient, never read) [2] exp_avg [3] exp_avg_sq (float*)  [4] bf16 image of the updated
// param or 0  [5] n  [6] 0 = use the launch's bias corrections (every tensor at the same step count: the table then only changes
// when a pointer does), else float bits of bias_correction1 | float bits of sqrt(bias_correction2) << 32  [7] first block; the last
// row is a sentinel whose [7] = total blocks.  4096 elements per block.
#include "common.h"

namespace {

constexpr int ADAM_CHUNK = 4096;
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ inline void adam_one(float& param, float g, float& exp_avg, float& exp_avg_sq, double lr, double beta1, double beta2,
                                double weight_decay, double eps, float bc1, float bc2_sqrt) {
  float grad = g;
  if (weight_decay != 0) grad = (float)((double)grad + (double)param * weight_decay);      // grad += param * weight_decay
  exp_avg = (float)(beta1 * (double)exp_avg + (1 - beta1) * (double)grad);
  exp_avg_sq = (float)(beta2 * (double)exp_avg_sq + (1 - beta2) * (double)grad * (double)grad);
  const float step_size = (float)(lr / (double)bc1);
  const float denom = (float)((double)(sqrtf(exp_avg_sq) / bc2_sqrt) + eps);
  param -= step_size * exp_avg / denom;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const long long* __restrict__ jobs, int n_jobs, double lr, double beta1,
                                                         double beta2, double weight_decay, double eps, float bc1_all,
                                                         float bc2s_all) {
  int lo = 0, hi = n_jobs;
  const long long b = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (jobs[(size_t)mid * 8 + 7] <= b) lo = mid; else hi = mid;
  }
  const long long* J = jobs + (size_t)lo * 8;
  float* __restrict__ P = reinterpret_cast<float*>(J[0]);
  const float* __restrict__ G = reinterpret_cast<const float*>(J[1]);
  float* __restrict__ M = reinterpret_cast<float*>(J[2]);
  float* __restrict__ V = reinterpret_cast<float*>(J[3]);
  u16* __restrict__ S = reinterpret_cast<u16*>(J[4]);
  const long long n = J[5];
  // bias corrections: per job when the tensors are at different step counts (row field != 0), else the launch's
  const float bc1 = J[6] ? __builtin_bit_cast(float, (unsigned)(J[6] & 0xFFFFFFFFll)) : bc1_all;
  const float bc2s = J[6] ? __builtin_bit_cast(float, (unsigned)((unsigned long long)J[6] >> 32)) : bc2s_all;
  const long long base = (b - J[7]) * ADAM_CHUNK;
  const bool vec = (n & 3) == 0;
#pragma unroll
  for (int k = 0; k < ADAM_CHUNK / 1024; ++k) {
    const long long i = base + k * 1024 + threadIdx.x * 4;
    if (vec && i + 4 <= n) {
      f32x4 p = *reinterpret_cast<const f32x4*>(P + i);
      const f32x4 g = G ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(G + i)) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 m = *reinterpret_cast<const f32x4*>(M + i);
      f32x4 v = *reinterpret_cast<const f32x4*>(V + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_one(pe, g[e], me, ve, lr, beta1, beta2, weight_decay, eps, bc1, bc2s);
        p[e] = pe; m[e] = me; v[e] = ve;
      }
      *reinterpret_cast<f32x4*>(P + i) = p;
      *reinterpret_cast<f32x4*>(M + i) = m;
      *reinterpret_cast<f32x4*>(V + i) = v;
      if (S) *reinterpret_cast<u32x2*>(S + i) = u32x2{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3])};
    } else {
      for (long long j = i; j < n && j < i + 4; ++j) {
        float pe = P[j], me = M[j], ve = V[j];
        adam_one(pe, G ? G[j] : 0.f, me, ve, lr, beta1, beta2, weight_decay, eps, bc1, bc2s);
        P[j] = pe; M[j] = me; V[j] = ve;
        if (S) S[j] = __builtin_bit_cast(u16, (__bf16)pe);
      }
    }
  }
}

}  // namespace

extern "C" int pangu_adam_step_multi(pangu_stream_t stream, const void* jobs, int n_jobs, long long total_blocks, double lr,
                                     double beta1, double beta2, double weight_decay, double eps, float bias_correction1,
                                     float bias_correction2_sqrt) {
  if (!jobs) return PANGU_E_NULL;
  if (n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7FFFFFFFll) return PANGU_E_SHAPE;
  if (!(lr >= 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps >= 0) || !(weight_decay >= 0)) return PANGU_E_ARG;
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), n_jobs, lr, beta1, beta2, weight_decay, eps, bias_correction1,
                     bias_correction2_sqrt);
  return pangu_launch_status();
}
