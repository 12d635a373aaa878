// How fast can a launch pull ROWS x 4 KB rows when every workgroup walks its NT rows 128 B (one 32-float line) at a time --
// the bank access pattern of the pair_exp kernels -- against 512 B / 1 KB / 4 KB per row visit?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// workgroup b owns rows [b*NT, (b+1)*NT); per step it reads SEG bytes of each row; a 16-byte piece per thread and load,
// PIECES loads in flight per thread and step
template <int SEG, int DEPTH>
__global__ __launch_bounds__(256) void k_rows(const float4* __restrict__ p, int NT, float* out) {
  const int tid = threadIdx.x;
  constexpr int TPR = SEG / 16;                 // threads per row segment
  const int rows_per_pass = 256 / TPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long row0 = (long long)blockIdx.x * NT;
  // DEPTH row visits in flight per thread (requested together, summed afterwards)
  const int visits = (NT + rows_per_pass - 1) / rows_per_pass * (4096 / SEG);
  const int per_seg = (NT + rows_per_pass - 1) / rows_per_pass;
  for (int v0 = 0; v0 < visits; v0 += DEPTH) {
    float4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int vi = v0 + d < visits ? v0 + d : visits - 1;
      const int seg = vi / per_seg, r = tid / TPR + (vi - seg * per_seg) * rows_per_pass;
      v[d] = p[(row0 + (r < NT ? r : NT - 1)) * 256 + seg * TPR + (tid % TPR)];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
  }
  if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}
int main(int argc, char** argv) {
  const long long ROWS = argc > 1 ? atoll(argv[1]) : 20480;   // 20480 rows = 84 MB
  float4* p; float* out; CK(hipMalloc(&p, ROWS * 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(p, 0, ROWS * 4096));
  float4* scratch; CK(hipMalloc(&scratch, 512ll << 20)); 
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int NT : {32, 128}) {
    const int grid = (int)(ROWS / NT);
    auto run = [&](auto kern, const char* name) {
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        CK(hipMemsetAsync(scratch, it, 512ll << 20, st));   // push the rows out of the caches
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, (const float4*)p, NT, out);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("NT=%3d rows per workgroup (%4d workgroups), %s per row visit: %7.1f us = %5.2f TB/s\n", NT, grid, name, best * 1e3f, ROWS * 4096.0 / (best * 1e-3) / 1e12);
    };
    run(k_rows<128, 1>, "128 B x1 "); run(k_rows<128, 4>, "128 B x4 "); run(k_rows<128, 8>, "128 B x8 "); run(k_rows<128, 16>, "128 B x16");
    run(k_rows<4096, 8>, "4 KB  x8 "); run(k_rows<4096, 16>, "4 KB  x16");
  }
  return 0;
}
