// Probe (MI355X): ds_read_b64_tr_b16 operand fetch for v_mfma_f32_32x32x16_bf16 from a k-major LDS plane
// [position k][64 channels] bf16 with the half-swap swizzle used by wgrad3b_kernel.  D = A^T B for one K16 step,
// checked against the host.      hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __host__ inline int plane_off(int pos, int ch) {   // byte offset
  return pos * 128 + (((ch >> 5) ^ ((pos >> 1) & 1)) * 64) + (ch & 31) * 2;
}
__global__ void k(const uint16_t* a, const uint16_t* b, float* d, int shift) {
  __shared__ __attribute__((aligned(16))) uint16_t pa[64 * 64], pb[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) {
    const int pos = i >> 6, ch = i & 63;
    pa[plane_off(pos, ch) >> 1] = a[i];
    pb[plane_off(pos, ch) >> 1] = b[i];
  }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int it = 1, ct = 0;      // tiles: A channels 32..63, B channels 0..31
  const int row = 8 * (g >> 1) + q;
  // the swizzle bit of this lane's rows: (pos >> 1) & 1 = (q >> 1) & 1 when the block base is a multiple of 4
  const int sw = (q >> 1) & 1;
  const int abyte = row * 128 + ((it ^ sw) * 64) + 32 * (g & 1) + 8 * p;
  const int bbyte = row * 128 + ((ct ^ sw) * 64) + 32 * (g & 1) + 8 * p;
  const char* A = (const char*)pa + shift * 128;
  const char* B = (const char*)pb;
  s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(A + abyte));
  s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(A + abyte + 512));
  s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(B + bbyte));
  s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(B + bbyte + 512));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
static float bf(uint16_t u) { uint32_t x = (uint32_t)u << 16; float f; memcpy(&f, &x, 4); return f; }
int main() {
  uint16_t ha[64 * 64], hb[64 * 64];
  for (int i = 0; i < 64 * 64; ++i) { ha[i] = 0x3f80 + ((i * 2654435761u >> 7) % 61); hb[i] = 0x3f00 + ((i * 40503u + (i >> 6) * 97u) % 113); }
  uint16_t *da, *db; float* dd; float hd[32 * 32];
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  for (int shift = 0; shift <= 16; shift += 8) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, shift);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      double s = 0;
      for (int kk = 0; kk < 16; ++kk) s += (double)bf(ha[(kk + shift) * 64 + 32 + m]) * bf(hb[kk * 64 + n]);
      worst = fmax(worst, fabs(s - hd[m * 32 + n]));
    }
    printf("shift %d: max |err| = %g  (d[0]=%g)\n", shift, worst, hd[0]);
  }
  return 0;
}
