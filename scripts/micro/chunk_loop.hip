// What does one "chunk" of a staged MFMA loop cost on this machine when a SIMD holds ONE wave?  (the pattern of
// loss_dfeat_lds_kernel's long problem, pair_exp_wide_kernel's lines, the fused forward's conv0 chunks: stage -> barrier
// -> fragment reads -> a few MFMAs, measured at 0.8-1.8 us per chunk where the MFMAs need 0.2.)
// One workgroup of 256 threads per CU, ITER chunks; the ingredients can be switched off one by one.
//   hipcc --offload-arch=gfx950 -O3 chunk_loop.hip -o chunk_loop && ./chunk_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

// WR: five ds_write_b128 per thread and chunk; BAR: the barrier; RD: 32 ds_read_b32 per lane and chunk; NM: MFMAs per
// wave and chunk; CH: independent accumulator chains; KIND 0: v_mfma_f32_16x16x4_f32 (32 cycles), 1: v_mfma_f32_32x32x16_bf16
template <int WR, int BAR, int RD, int NM, int CH, int KIND>
__global__ __launch_bounds__(256) void k_chunk(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2][64 * 96];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x4 acc4[4] = {};
  f32x16 acc16[4] = {};
  float4 v = make_float4(tid, 1.f, 2.f, 3.f);
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    float* st = lds[it & 1];
    if (WR) {
#pragma unroll
      for (int q = 0; q < 5; ++q) *(float4*)(st + ((tid + 256 * q) % 1536) * 4) = v;
    }
    if (BAR) __syncthreads();
    float av[16], bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (RD) { av[k] = st[k * 64 + lane]; bv[k] = st[3072 + k * 80 + lane]; }
      else { av[k] = v.x + k; bv[k] = v.y + k; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (KIND == 0) acc4[m % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m & 15], bv[m & 15], acc4[m % CH], 0, 0, 0);
      else {
        bf16x8 a8, b8;
        __builtin_memcpy(&a8, &av[(m & 3) * 4], 16); __builtin_memcpy(&b8, &bv[(m & 3) * 4], 16);
        acc16[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc16[m % CH], 0, 0, 0);
      }
    }
    if (NM == 0) { keep += av[it & 15] + bv[(it + 1) & 15]; }
    v.x += 1.f;
  }
  float r = keep;
  for (int c = 0; c < 4; ++c) { r += acc4[c][0] + acc16[c][0]; }
  if (r == 123.456f) out[0] = r;
}
template <class K> static void run(K kern, const char* name, hipStream_t st, float* out) {
  const int iters = 512;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, st, out, iters);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-78s %6.0f ns per chunk\n", name, best * 1e6f / iters);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float* out; CK(hipMalloc(&out, 64));
  run(k_chunk<1, 1, 1, 16, 2, 0>, "writes + barrier + 32 reads + 16 MFMA 16x16x4 f32 in 2 chains (the dfeat chunk)", st, out);
  run(k_chunk<1, 1, 1, 16, 4, 0>, "   ... in 4 chains", st, out);
  run(k_chunk<1, 1, 1, 16, 1, 0>, "   ... in 1 chain", st, out);
  run(k_chunk<0, 1, 1, 16, 2, 0>, "   without the LDS writes", st, out);
  run(k_chunk<1, 0, 1, 16, 2, 0>, "   without the barrier", st, out);
  run(k_chunk<1, 1, 0, 16, 2, 0>, "   without the LDS reads", st, out);
  run(k_chunk<1, 1, 1, 0, 2, 0>, "   without the MFMAs", st, out);
  run(k_chunk<0, 0, 0, 16, 2, 0>, "16 MFMA 16x16x4 f32 alone, 2 chains", st, out);
  run(k_chunk<0, 0, 0, 16, 4, 0>, "16 MFMA 16x16x4 f32 alone, 4 chains", st, out);
  run(k_chunk<0, 0, 0, 16, 1, 0>, "16 MFMA 16x16x4 f32 alone, 1 chain", st, out);
  run(k_chunk<0, 0, 0, 12, 2, 1>, "12 MFMA 32x32x16 bf16 alone, 2 chains", st, out);
  run(k_chunk<0, 0, 0, 12, 1, 1>, "12 MFMA 32x32x16 bf16 alone, 1 chain", st, out);
  run(k_chunk<0, 1, 0, 0, 1, 0>, "the barrier alone", st, out);
  run(k_chunk<1, 1, 1, 0, 1, 0>, "writes + barrier + reads (no MFMA)", st, out);
  return 0;
}
