// Weight-shadow refresh for gfx950: every bf16 derived copy of the fp32 master weights in ONE launch.
//
// The bf16 paths run on bf16 copies of the parameters: plain casts (projection weights, the Earth-specific bias tables: 94 % of
// the bytes), transposed casts (the W operand of the input-gradient GEMMs) and the packed chunk image of the fused MLP kernel
// (a fixed permutation of the two Mlp weights).  After an optimizer step all of them are stale.  Re-made one tensor at a time
// with torch ops that is ~240 launches of 3-7 us (1.6 ms of a 47 ms training step, 10 ms of host time); here a device-resident
// job table describes them all and one launch streams the 1.1 GB of fp32 weights once (HBM-bound: 1.56 GB).
//
// Job table: n_jobs + 1 rows of 8 int64 (row n_jobs is a sentinel carrying the total block count):
//   [0] src0 (fp32)  [1] src1 (fp32, gather only)  [2] dst (bf16)  [3] idx (int32, gather only)  [4] n0  [5] n1  [6] mode
//   [7] first block of the job
//   mode 0  cast:       dst[i] = bf16(src0[i]),  i < n0                               4096 elements per block
//   mode 1  transpose:  dst[c][r] = bf16(src0[r][c]),  r < n0, c < n1                 64 x 64 tile per block
//   mode 2  gather:     e = idx[i]; dst[i] = bf16(e < n0 ? src0[e] : src1[e - n0]),  i < n1      4096 elements per block
#include "common.h"

namespace {

typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int SH_CHUNK = 4096;

__device__ inline u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }      // round-to-nearest-even, as torch's cast

__global__ __launch_bounds__(256) void shadow_refresh_bf16_kernel(const long long* __restrict__ jobs, int n_jobs) {
  __shared__ float tile[64][65];
  // binary search: the job whose block range holds blockIdx.x (uniform: scalar loads)
  int lo = 0, hi = n_jobs;
  const long long b = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (jobs[(size_t)mid * 8 + 7] <= b) lo = mid; else hi = mid;
  }
  const long long* J = jobs + (size_t)lo * 8;
  const float* __restrict__ src0 = reinterpret_cast<const float*>(J[0]);
  const float* __restrict__ src1 = reinterpret_cast<const float*>(J[1]);
  u16* __restrict__ dst = reinterpret_cast<u16*>(J[2]);
  const int* __restrict__ idx = reinterpret_cast<const int*>(J[3]);
  const long long n0 = J[4], n1 = J[5];
  const int mode = (int)J[6];
  const long long chunk = b - J[7];
  const int t = threadIdx.x;

  if (mode == 0) {
    const long long base = chunk * SH_CHUNK;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long i = base + h * 2048 + t * 8;
      if (i + 8 <= n0) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src0 + i);
        const f32x4 c = *reinterpret_cast<const f32x4*>(src0 + i + 4);
        *reinterpret_cast<u32x4*>(dst + i) = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(c[0], c[1]),
                                                   pack_bf16x2(c[2], c[3])};
      } else {
        for (long long k = i; k < n0 && k < i + 8; ++k) dst[k] = f2bf(src0[k]);
      }
    }
  } else if (mode == 1) {
    const long long tiles_c = (n1 + 63) / 64;
    const long long r0 = (chunk / tiles_c) * 64, c0 = (chunk % tiles_c) * 64;
    const int ty = t >> 4, tx = t & 15;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long long r = r0 + ty + 16 * p, c = c0 + tx * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < n0) {
        if (c + 4 <= n1 && (n1 & 3) == 0) v = *reinterpret_cast<const f32x4*>(src0 + r * n1 + c);
        else
          for (int e = 0; e < 4; ++e)
            if (c + e < n1) v[e] = src0[r * n1 + c + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[ty + 16 * p][tx * 4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long long c = c0 + ty + 16 * p, r = r0 + tx * 4;      // dst row c (a source column), dst columns r .. r+3
      if (c < n1) {
        const int lc = ty + 16 * p;
        if (r + 4 <= n0 && (n0 & 3) == 0) {
          *reinterpret_cast<u32x2*>(dst + c * n0 + r) =
              u32x2{pack_bf16x2(tile[tx * 4][lc], tile[tx * 4 + 1][lc]), pack_bf16x2(tile[tx * 4 + 2][lc], tile[tx * 4 + 3][lc])};
        } else {
          for (int e = 0; e < 4; ++e)
            if (r + e < n0) dst[c * n0 + r + e] = f2bf(tile[tx * 4 + e][lc]);
        }
      }
    }
  } else {
    const long long base = chunk * SH_CHUNK;
#pragma unroll 4
    for (int k = 0; k < SH_CHUNK / 256; ++k) {
      const long long i = base + k * 256 + t;
      if (i < n1) {
        const long long e = idx[i];
        dst[i] = f2bf(e < n0 ? src0[e] : src1[e - n0]);
      }
    }
  }
}

}  // namespace

extern "C" int pangu_shadow_refresh_bf16(pangu_stream_t stream, const void* jobs, int n_jobs, long long total_blocks) {
  if (!jobs) return PANGU_E_NULL;
  if (n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7FFFFFFFll) return PANGU_E_SHAPE;
  hipLaunchKernelGGL(shadow_refresh_bf16_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), n_jobs);
  return pangu_launch_status();
}
