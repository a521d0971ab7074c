// Multi-GPU rehearsal on ONE GPU: a device-to-device copy confined to a FIXED number of workgroups, launched
// on a side stream next to the backward kernels -- a stand-in for the HBM traffic and the CUs an RCCL all-reduce of the same bucket
// occupies on its own stream (RCCL runs a few tens of persistent workgroups, one per channel, not a grid that fills the chip).
// What it measures: an upper bound on how much the bucketed collective slows the compute kernels it overlaps with.  It is not part
// of the product path: dist.FlatGradSync(rehearse=..) and bench.py --rehearse-collective are the only callers.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void traffic_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i + k * 256 < n16 ? __builtin_nontemporal_load(src + i + k * 256) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * 256 < n16) __builtin_nontemporal_store(v[k], dst + i + k * 256);
  }
}

}  // namespace

extern "C" int pangu_traffic_copy(pangu_stream_t stream, const void* src, void* dst, long long bytes, int workgroups) {
  if (!src || !dst) return PANGU_E_NULL;
  if (bytes <= 0 || (bytes & 15) || workgroups <= 0 || ((size_t)src & 15) || ((size_t)dst & 15)) return PANGU_E_SHAPE;
  hipLaunchKernelGGL(traffic_copy_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst,
                     (size_t)bytes / 16);
  return pangu_launch_status();
}
