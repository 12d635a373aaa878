// Microbenchmark: v_mfma_f32_32x32x2_f32 rate of ONE wave per SIMD when other instructions sit between the MFMAs.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_mix.hip -o scripts/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
#define SB __builtin_amdgcn_sched_barrier(0)
// MODE 0: 9 MFMAs back to back; 1: one ds_read_b32 after each MFMA (operand of the next iteration);
// 2: ds_read + 2 scalar ops; 3: all 9 reads in a block before the MFMAs; 4: ds_read_b32 into unused regs;
// 5: one v_add after each MFMA (no LDS)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.f + i * 1e-6f;
  __syncthreads();
  v16f acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float* p = lds + (threadIdx.x & 63);
  float a[9], n[9], b = 1.f, dummy = 0.f;
#pragma unroll
  for (int c = 0; c < 9; ++c) a[c] = p[c * 64];
  int off = 0, cnt = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int c = 0; c < 9; ++c) n[c] = p[off + c * 64];
      SB;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b, acc[c], 0, 0, 0);
      if (MODE == 1 || MODE == 2 || MODE == 4) n[c] = p[off + c * 64];
      if (MODE == 2) { cnt += stride; if (cnt == 77) cnt = 0; }
      if (MODE == 5) dummy += b;
      SB;
    }
    off = (off + stride) & 4095;
    if (MODE == 2) off = (off + cnt) & 4095;
    if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int c = 0; c < 9; ++c) a[c] = n[c];
    } else if (MODE == 4) {
#pragma unroll
      for (int c = 0; c < 9; ++c) dummy += n[c];
    }
  }
  float s = dummy;
#pragma unroll
  for (int c = 0; c < 9; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE>
static void run(float* d, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000, wgs = 256;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters, 128);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double cyc = best * 1e-3 * 2.4e9 / (iters * 9.0);
  printf("mode %d (%s): %.1f us, %.1f cycles per MFMA at 2.4 GHz\n", MODE, name, best * 1e3, cyc);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  run<0>(d, "mfma only");
  run<1>(d, "mfma + ds_read each, used next iter");
  run<2>(d, "mfma + ds_read + salu");
  run<3>(d, "reads in a block, then mfmas");
  run<4>(d, "mfma + ds_read each, unused");
  run<5>(d, "mfma + v_add each");
  return 0;
}
