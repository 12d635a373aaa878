// Microbenchmark 2: what slows a 3-accumulator MFMA stream (the row-split wgrad inner loop) with 1 or 2 waves / SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_mix2.hip -o scripts/mfma_mix2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
#define SB __builtin_amdgcn_sched_barrier(0)
// NACC accumulators per wave; per MFMA: LD ds_read_b32 (1 = yes), VA v_add count, SA scalar op count
template <int NT, int NACC, int LD, int VA, int SA>
__global__ __launch_bounds__(NT) void k(float* out, int iters, int stride, int lim) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = 1.f + i * 1e-6f;
  __syncthreads();
  v16f acc[NACC];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float* p = lds + (threadIdx.x & 63);
  float a0[NACC], a1[NACC], b = 1.f, dummy = 0.f;
#pragma unroll
  for (int c = 0; c < NACC; ++c) { a0[c] = p[c * 64]; a1[c] = p[c * 64 + 1]; }
  int off = 0, cnt = 0, s2 = 1;
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int c = 0; c < NACC; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b, acc[c], 0, 0, 0);
      if (LD) a1[c] = p[off + c * 64];
#pragma unroll
      for (int v = 0; v < VA; ++v) dummy += b;
#pragma unroll
      for (int q = 0; q < SA; ++q) { cnt += stride; if (cnt == lim) cnt = s2; s2 ^= cnt; }
      SB;
    }
    off = (off + stride + (cnt & 1)) & 4095;
#pragma unroll
    for (int c = 0; c < NACC; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b, acc[c], 0, 0, 0);
      if (LD) a0[c] = p[off + c * 64];
#pragma unroll
      for (int v = 0; v < VA; ++v) dummy += b;
#pragma unroll
      for (int q = 0; q < SA; ++q) { cnt += stride; if (cnt == lim) cnt = s2; s2 ^= cnt; }
      SB;
    }
    off = (off + stride + (cnt & 1)) & 4095;
  }
  float s = dummy + cnt;
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int NT, int NACC, int LD, int VA, int SA>
static void run(float* d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 6000, wgs = 256;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NT, NACC, LD, VA, SA>), dim3(wgs), dim3(NT), 0, 0, d, iters, 128, 77777);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double waves_per_simd = NT / 256.0;
  const double cyc = best * 1e-3 * 2.4e9 / (iters * (double)NACC * waves_per_simd);
  printf("waves/SIMD %.0f  acc %d  lds %d  valu %d  salu %d : %.1f cycles per MFMA (pipe)\n", waves_per_simd, NACC, LD,
         VA, SA, cyc);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  run<256, 3, 0, 0, 0>(d); run<512, 3, 0, 0, 0>(d); run<1024, 3, 0, 0, 0>(d);
  run<512, 3, 1, 0, 0>(d);
  run<512, 3, 1, 1, 0>(d);
  run<512, 3, 1, 1, 2>(d);
  run<512, 3, 1, 1, 4>(d);
  run<512, 3, 1, 1, 8>(d);
  run<512, 3, 0, 0, 8>(d);
  run<512, 3, 0, 2, 0>(d);
  run<1024, 3, 1, 1, 4>(d);
  run<1024, 3, 1, 1, 8>(d);
  run<512, 9, 1, 1, 2>(d);
  run<256, 9, 1, 1, 2>(d);
  return 0;
}
