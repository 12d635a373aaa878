// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 (what a perfect fp32 MFMA kernel could reach
// on this box, at the clock the card actually holds).  hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
  v16f acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float x = a + threadIdx.x * 1e-6f, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int CHAINS>
static void run(int wgs, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = double(wgs) * 4 * iters * 4 * CHAINS * 32.0 * 32 * 2 * 2;
    printf("chains=%d wgs=%d iters=%d  %.1f us  %.1f TFLOP/s\n", CHAINS, wgs, iters, ms * 1e3, flops / ms * 1e-9);
  }
}
int main() {
  float* d; hipMalloc(&d, 4096);
  // durations from ~30 us to ~5 ms, 1 and 2 workgroups per CU
  for (int wgs : {256, 512, 1024})
    for (int iters : {64, 256, 4096}) { run<1>(wgs, iters, d); run<2>(wgs, iters, d); run<4>(wgs, iters, d); }
  return 0;
}
