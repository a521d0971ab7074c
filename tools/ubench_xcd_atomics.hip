// micro-benchmark: 256 workgroups x 768 threads each add / store a 295-KB fp32 tile (96 floats per thread, lane-contiguous)
//   mode 0: plain stores into a private slice per workgroup (75 MB)          -- what the two-stage epilogue does
//   mode 1: atomic adds into ONE buffer per tile (8 tiles, shared by all XCDs) -- the old atomic tail
//   mode 2: atomic adds into a buffer per (XCC_ID, tile)                       -- XCD-local accumulation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int TILE = 384 * 192, TPB = 768, PER = TILE / TPB;   // 96
__global__ __launch_bounds__(TPB) void k(float* buf, int mode, int tiles, unsigned* xcc_seen) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15;
  const int tile = (blockIdx.x >> 3) % tiles;
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc;
  float* dst = mode == 0 ? buf + (size_t)blockIdx.x * TILE : mode == 1 ? buf + (size_t)tile * TILE : buf + ((size_t)xcc * tiles + tile) * TILE;
  const float v = 1.0f + threadIdx.x * 1e-6f;
#pragma unroll 8
  for (int i = 0; i < PER; ++i) {
    float* p = dst + i * TPB + threadIdx.x;
    if (mode == 0) __builtin_nontemporal_store(v, p);
    else atomicAdd(p, v);
  }
}
int main() {
  const int WGS = 256, tiles = 8;
  float* buf; unsigned* seen;
  CK(hipMalloc(&buf, (size_t)WGS * TILE * 4));
  CK(hipMalloc(&seen, WGS * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(buf, 0, (size_t)WGS * TILE * 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(WGS), dim3(TPB), 0, 0, buf, mode, tiles, seen);
    CK(hipDeviceSynchronize());
    CK(hipMemset(buf, 0, (size_t)WGS * TILE * 4));
    CK(hipEventRecord(a));
    const int R = 20;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k, dim3(WGS), dim3(TPB), 0, 0, buf, mode, tiles, seen);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // check a value
    std::vector<float> h(4); CK(hipMemcpy(h.data(), buf, 16, hipMemcpyDeviceToHost));
    printf("mode %d: %.1f us per launch (%.2f TB/s of 75.5 MB), buf[0] = %.3f\n", mode, ms / R * 1e3, 75.5e6 / (ms / R * 1e-3) / 1e12, h[0]);
  }
  std::vector<unsigned> hs(WGS); CK(hipMemcpy(hs.data(), seen, WGS * 4, hipMemcpyDeviceToHost));
  printf("XCC_ID of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hs[i]); printf("\n");
  int bad = 0; for (int i = 0; i < WGS; ++i) bad += (hs[i] != hs[i & 7]);
  printf("blocks whose XCC_ID differs from block (i & 7)'s: %d\n", bad);
  return 0;
}
