// Training loss of the reference in two passes over the fields instead of ~14 torch launches (reference models/pangu_sample.py:61-67
// with the variable weights of era5_data/config.py:45-46):
//   loss = mean(|o - t| * w_u[var]) + 0.25 * mean(|o_s - t_s| * w_s[var])
// o / t: (B, Vu, L, H, W) upper-air fields, o_s / t_s: (B, Vs, H, W) surface fields, fp32.  As torch ops the expression and its
// backward read or write the 286 MB fields fourteen times (sub, abs, mul, mean; sign, three multiplies; twice: 0.9 ms of the
// training step); here the forward reads o and t once (block sums -> one fp64 final sum) and the backward reads them once more and
// writes d_o = sign(o - t) * ((g / n) * w[var]) -- the order torch's autograd multiplies in -- 0.86 GB, HBM-bound.
// The target side of the host-fed step is folded in as well (no extra pass over the 286 MB target):
//   * t_mean / t_std given: the target arrives in PHYSICAL units and is normalised on the fly, t' = (t - mean[var][lev]) / std[var][lev]
//     (reference era5_data/utils_data.py:315-321 `normData`, called at models/pangu_sample.py:57: a subtract, then a true division);
//   * target_levels_reversed: the target is stored as the reader finds it on disk (level axis ascending) and the reversal of
//     era5_data/utils_data.py:117 is an address: logical level lev lives in plane L-1-lev.  Statistics are indexed by LOGICAL level.
// Blocks never straddle a (sample, variable, level) plane, so weight, statistics and the target's plane are block-uniform.
#include "common.h"

namespace {

constexpr int LOSS_CHUNK = 8192;      // elements per block, inside ONE (sample, variable) plane: the weight is block-uniform

struct LossGeom {
  long long plane_u, plane_s;         // elements of one (sample, variable, level) plane / one (sample, variable) surface plane
  int chunks_u, chunks_s;             // blocks per plane
  int planes_u, planes_s;             // B * Vu * L, B * Vs
  int Vu, Vs, L;
  int t_rev;                          // the target's level axis is stored reversed
};

struct TargetStats {                  // null = the target is already normalised
  const float* mean_u; const float* std_u;      // [Vu][L], logical level order
  const float* mean_s; const float* std_s;      // [Vs]
};

// block b -> which field, which variable, [begin, end) inside the plane, the plane's offset in out (base) and in target (base_t),
// and the index of the plane's statistics
__device__ inline void locate(const LossGeom& g, int b, bool& surface, int& var, long long& begin, long long& end,
                              long long& base, long long& base_t, int& stat) {
  const int nb_u = g.planes_u * g.chunks_u;
  surface = b >= nb_u;
  const int bb = surface ? b - nb_u : b;
  const int chunks = surface ? g.chunks_s : g.chunks_u;
  const long long plane = surface ? g.plane_s : g.plane_u;
  const int p = bb / chunks, c = bb - p * chunks;
  if (surface) {
    var = p % g.Vs;
    stat = var;
    base_t = (long long)p * plane;
  } else {
    const int lev = p % g.L;
    var = (p / g.L) % g.Vu;
    stat = var * g.L + lev;
    base_t = (long long)(p - lev + (g.t_rev ? g.L - 1 - lev : lev)) * plane;
  }
  base = (long long)p * plane;
  begin = (long long)c * LOSS_CHUNK;
  end = begin + LOSS_CHUNK < plane ? begin + LOSS_CHUNK : plane;
}

__global__ __launch_bounds__(256) void l1_loss_partial_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                              const float* __restrict__ os, const float* __restrict__ ts,
                                                              const float* __restrict__ wu, const float* __restrict__ ws,
                                                              float* __restrict__ partial, LossGeom g, TargetStats st) {
  bool surface; int var, stat; long long begin, end, base, base_t;
  locate(g, blockIdx.x, surface, var, begin, end, base, base_t, stat);
  const float* __restrict__ a = (surface ? os : o) + base;
  const float* __restrict__ b = (surface ? ts : t) + base_t;
  const bool nrm = st.mean_u != nullptr;
  const float mn = nrm ? (surface ? st.mean_s[stat] : st.mean_u[stat]) : 0.f;
  const float sd = nrm ? (surface ? st.std_s[stat] : st.std_u[stat]) : 1.f;
  auto tgt = [nrm, mn, sd](float y) { return nrm ? (y - mn) / sd : y; };
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = ((base | base_t | begin) & 3) == 0;
  if (vec) {
#pragma unroll
    for (int k = 0; k < LOSS_CHUNK / 1024; ++k) {
      const long long i = begin + k * 1024 + threadIdx.x * 4;
      if (i + 4 <= end) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + i), y = *reinterpret_cast<const f32x4*>(b + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += fabsf(x[e] - tgt(y[e]));
      } else {
        for (long long j = i; j < end; ++j) acc[0] += fabsf(a[j] - tgt(b[j]));
      }
    }
  } else {
    for (long long i = begin + threadIdx.x; i < end; i += 256) acc[0] += fabsf(a[i] - tgt(b[i]));
  }
  float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * (surface ? ws[var] : wu[var]);
}

// loss[0] = total, loss[1] = upper mean, loss[2] = surface mean
__global__ __launch_bounds__(256) void l1_loss_final_kernel(const float* __restrict__ partial, float* __restrict__ loss, int nb_u,
                                                            int nb_s, double n_u, double n_s) {
  double su = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < nb_u; i += 256) su += (double)partial[i];
  for (int i = threadIdx.x; i < nb_s; i += 256) ss += (double)partial[nb_u + i];
  __shared__ double ru[256], rs[256];
  ru[threadIdx.x] = su;
  rs[threadIdx.x] = ss;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ru[threadIdx.x] += ru[threadIdx.x + off];
      rs[threadIdx.x] += rs[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float lu = (float)(ru[0] / n_u), ls = (float)(rs[0] / n_s);
    loss[1] = lu;
    loss[2] = ls;
    loss[0] = lu + ls * 0.25f;
  }
}

__global__ __launch_bounds__(256) void l1_loss_bwd_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                          const float* __restrict__ os, const float* __restrict__ ts,
                                                          const float* __restrict__ wu, const float* __restrict__ ws,
                                                          const float* __restrict__ grad, float* __restrict__ d_o,
                                                          float* __restrict__ d_os, LossGeom g, TargetStats st, float inv_nu,
                                                          float inv_ns) {
  bool surface; int var, stat; long long begin, end, base, base_t;
  locate(g, blockIdx.x, surface, var, begin, end, base, base_t, stat);
  const float* __restrict__ a = (surface ? os : o) + base;
  const float* __restrict__ b = (surface ? ts : t) + base_t;
  float* __restrict__ d = (surface ? d_os : d_o) + base;
  const bool nrm = st.mean_u != nullptr;
  const float mn = nrm ? (surface ? st.mean_s[stat] : st.mean_u[stat]) : 0.f;
  const float sd = nrm ? (surface ? st.std_s[stat] : st.std_u[stat]) : 1.f;
  auto tgt = [nrm, mn, sd](float y) { return nrm ? (y - mn) / sd : y; };
  // torch's autograd: d(mean) = g * (1 / n) (true division by a host scalar is a multiply by its fp32 reciprocal), times the
  // variable weight, times sign(o - t); the surface term's incoming gradient is g * 0.25
  const float gr = grad[0];
  float c = surface ? ((gr * 0.25f) * inv_ns) : (gr * inv_nu);
  asm volatile("" : "+v"(c));                       // keep the two multiplies apart (no re-association through the weight)
  c = c * (surface ? ws[var] : wu[var]);
  auto sgn = [c](float x) { return x > 0.f ? c : (x < 0.f ? -c : c * 0.f); };       // NaN -> NaN, like torch.sign's product
  const bool vec = ((base | base_t | begin) & 3) == 0;
  if (vec) {
#pragma unroll
    for (int k = 0; k < LOSS_CHUNK / 1024; ++k) {
      const long long i = begin + k * 1024 + threadIdx.x * 4;
      if (i + 4 <= end) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + i), y = *reinterpret_cast<const f32x4*>(b + i);
        *reinterpret_cast<f32x4*>(d + i) =
            f32x4{sgn(x[0] - tgt(y[0])), sgn(x[1] - tgt(y[1])), sgn(x[2] - tgt(y[2])), sgn(x[3] - tgt(y[3]))};
      } else {
        for (long long j = i; j < end; ++j) d[j] = sgn(a[j] - tgt(b[j]));
      }
    }
  } else {
    for (long long i = begin + threadIdx.x; i < end; i += 256) d[i] = sgn(a[i] - tgt(b[i]));
  }
}

bool make_loss_geom(LossGeom& g, int B, int Vu, long long plane_u, int Vs, long long plane_s, int levels, int t_rev) {
  if (B <= 0 || Vu <= 0 || Vs <= 0 || plane_u <= 0 || plane_s <= 0 || levels <= 0 || plane_u % levels) return false;
  g.plane_u = plane_u / levels; g.plane_s = plane_s;
  g.chunks_u = (int)((g.plane_u + LOSS_CHUNK - 1) / LOSS_CHUNK);
  g.chunks_s = (int)((plane_s + LOSS_CHUNK - 1) / LOSS_CHUNK);
  g.planes_u = B * Vu * levels; g.planes_s = B * Vs;
  g.Vu = Vu; g.Vs = Vs; g.L = levels; g.t_rev = t_rev != 0;
  const long long blocks = (long long)g.planes_u * g.chunks_u + (long long)g.planes_s * g.chunks_s;
  return blocks > 0 && blocks < (1ll << 30);
}

bool make_stats(TargetStats& st, const float* mu, const float* su, const float* ms, const float* ss) {
  st = TargetStats{mu, su, ms, ss};
  const int n = (mu != nullptr) + (su != nullptr) + (ms != nullptr) + (ss != nullptr);
  return n == 0 || n == 4;            // all four or none
}

}  // namespace

extern "C" long long pangu_weighted_l1_loss_blocks(int B, int Vu, long long plane_u, int Vs, long long plane_s, int levels) {
  LossGeom g;
  if (!make_loss_geom(g, B, Vu, plane_u, Vs, plane_s, levels, 0)) return PANGU_E_SHAPE;
  return (long long)g.planes_u * g.chunks_u + (long long)g.planes_s * g.chunks_s;
}

extern "C" int pangu_weighted_l1_loss_fwd(pangu_stream_t stream, const float* out, const float* target, const float* out_surface,
                                          const float* target_surface, const float* w_upper, const float* w_surface,
                                          float* partial, float* loss, int B, int Vu, long long plane_u, int Vs,
                                          long long plane_s, int levels, int target_levels_reversed, const float* t_mean_upper,
                                          const float* t_std_upper, const float* t_mean_surface, const float* t_std_surface) {
  if (!out || !target || !out_surface || !target_surface || !w_upper || !w_surface || !partial || !loss) return PANGU_E_NULL;
  LossGeom g;
  TargetStats st;
  if (!make_loss_geom(g, B, Vu, plane_u, Vs, plane_s, levels, target_levels_reversed)) return PANGU_E_SHAPE;
  if (!make_stats(st, t_mean_upper, t_std_upper, t_mean_surface, t_std_surface)) return PANGU_E_NULL;
  const int nb_u = g.planes_u * g.chunks_u, nb_s = g.planes_s * g.chunks_s;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_loss_partial_kernel, dim3(nb_u + nb_s), dim3(256), 0, s, out, target, out_surface, target_surface, w_upper,
                     w_surface, partial, g, st);
  hipLaunchKernelGGL(l1_loss_final_kernel, dim3(1), dim3(256), 0, s, partial, loss, nb_u, nb_s, (double)B * Vu * (double)plane_u,
                     (double)g.planes_s * (double)plane_s);
  return pangu_launch_status();
}

extern "C" int pangu_weighted_l1_loss_bwd(pangu_stream_t stream, const float* out, const float* target, const float* out_surface,
                                          const float* target_surface, const float* w_upper, const float* w_surface,
                                          const float* grad, float* d_out, float* d_out_surface, int B, int Vu,
                                          long long plane_u, int Vs, long long plane_s, int levels, int target_levels_reversed,
                                          const float* t_mean_upper, const float* t_std_upper, const float* t_mean_surface,
                                          const float* t_std_surface) {
  if (!out || !target || !out_surface || !target_surface || !w_upper || !w_surface || !grad || !d_out || !d_out_surface) return PANGU_E_NULL;
  LossGeom g;
  TargetStats st;
  if (!make_loss_geom(g, B, Vu, plane_u, Vs, plane_s, levels, target_levels_reversed)) return PANGU_E_SHAPE;
  if (!make_stats(st, t_mean_upper, t_std_upper, t_mean_surface, t_std_surface)) return PANGU_E_NULL;
  const int nb = g.planes_u * g.chunks_u + g.planes_s * g.chunks_s;
  const float inv_nu = 1.0f / (float)((double)B * Vu * (double)plane_u), inv_ns = 1.0f / (float)((double)g.planes_s * (double)plane_s);
  hipLaunchKernelGGL(l1_loss_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, out, target, out_surface, target_surface,
                     w_upper, w_surface, grad, d_out, d_out_surface, g, st, inv_nu, inv_ns);
  return pangu_launch_status();
}
