// Launch floor on this machine: back-to-back launches of (a) an empty kernel, (b) a kernel that dirties some MB,
// (c) chains handing over through memory, timed with HIP events over many launches.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_empty() {}
__global__ void k_lds() { extern __shared__ float s[]; if (threadIdx.x == 9999) s[0] = 1.f; }
__global__ void k_write(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (float)i;
}
__global__ void k_rw(const float* a, float* b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i] + 1.f;
}
template <class F> static float time_us(F f, int reps, hipStream_t st) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) f();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *a, *b; const long N = 16 << 20; CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  const int R = 2000;
  printf("empty, 1 workgroup:                         %.2f us / launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, R, st));
  printf("empty, 512 workgroups x 256:                %.2f us / launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st); }, R, st));
  printf("empty, 512 x 256, 72 KB LDS:                %.2f us / launch\n", time_us([&] { hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), 72 * 1024, st); }, R, st));
  printf("empty, 2048 x 256:                          %.2f us / launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st); }, R, st));
  for (long mb : {1L, 8L, 32L}) {
    const long n = mb << 18;
    printf("write %2ld MB (512 x 256):                    %.2f us / launch\n", mb, time_us([&] { hipLaunchKernelGGL(k_write, dim3(512), dim3(256), 0, st, a, n); }, R, st));
    printf("read+write %2ld MB, ping-pong chain:           %.2f us / launch\n", mb, time_us([&] { hipLaunchKernelGGL(k_rw, dim3(512), dim3(256), 0, st, a, b, n); hipLaunchKernelGGL(k_rw, dim3(512), dim3(256), 0, st, b, a, n); }, R / 2, st) / 2);
  }
  // the same empty launches captured in a graph of 10 kernel nodes
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("graph of 10 empty 512 x 256 launches:       %.2f us / launch\n", time_us([&] { hipGraphLaunch(ge, st); }, 500, st) / 10);
  return 0;
}
