// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_ID: bits 5:4 = SIMD, 11:8 = CU, 3:0 = wave slot)
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/wave_simd.hip -o /tmp/wave_simd && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  extern __shared__ float smem[];
  unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  smem[threadIdx.x] = 0.f;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipLaunchKernelGGL(probe, dim3(64), dim3(512), 140 * 1024, 0, d);
  unsigned h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 6; ++b) {
    printf("wg %d:", b);
    for (int w = 0; w < 8; ++w) printf("  w%d simd %u cu %u slot %u |", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
    printf("\n");
  }
  int hist[4][4] = {};
  for (int b = 0; b < 64; ++b) for (int w = 0; w < 8; ++w) hist[w & 3][(h[b * 8 + w] >> 4) & 3]++;
  for (int r = 0; r < 4; ++r) printf("wave %% 4 == %d: simd histogram %d %d %d %d\n", r, hist[r][0], hist[r][1], hist[r][2], hist[r][3]);
  return 0;
}
