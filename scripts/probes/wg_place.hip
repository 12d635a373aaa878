// Which workgroups of a (256, 2) grid of 256-thread workgroups with 75 KB of LDS (two per CU: the fused per-sample
// kernels' launch) share a CU?  HW_ID: bits 11:8 CU, 12 SH, 15:13 SE; XCC_ID (hwreg 20): bits 3:0.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/wg_place.hip -o /tmp/wg_place && /tmp/wg_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out) {
  extern __shared__ float smem[];
  const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
  const int b = blockIdx.x + gridDim.x * blockIdx.y;
  if (threadIdx.x == 0) { out[2 * b] = id; out[2 * b + 1] = xcc; }
  smem[threadIdx.x] = 0.f;
  // stay resident long enough for the whole grid to be placed
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {}
}
int main() {
  const int GX = 256, GY = 2;
  unsigned* d; (void)hipMalloc(&d, GX * GY * 8);
  (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 75 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(GX, GY), dim3(256), 75 * 1024, 0, d);
  std::vector<unsigned> h(GX * GY * 2); (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < GX * GY; ++b) {
    const unsigned id = h[2 * b], x = h[2 * b + 1] & 15;
    cu[(x << 16) | (((id >> 13) & 7) << 8) | (((id >> 12) & 1) << 4) | ((id >> 8) & 15)].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu) among %d workgroups\n", cu.size(), GX * GY);
  int same_net = 0, diff_net = 0, other = 0, shown = 0;
  for (auto& kv : cu) {
    if (kv.second.size() == 2) { if (kv.second[0] / GX == kv.second[1] / GX) ++same_net; else ++diff_net; } else ++other;
    if (shown++ < 12) { printf("  xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15);
      for (int b : kv.second) printf("  (x %3d, y %d)", b % GX, b / GX); printf("\n"); }
  }
  printf("CUs holding two workgroups of the same y: %d, of different y: %d, other counts: %d\n", same_net, diff_net, other);
  printf("workgroup -> xcc of the first 16: ");
  for (int b = 0; b < 16; ++b) printf("%u ", h[2 * b + 1] & 15);
  printf("\n");
  return 0;
}
