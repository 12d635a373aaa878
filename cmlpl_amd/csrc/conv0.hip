// conv0: 1x1 convolution C -> 64 over the band-major patch tensor (tools/models.py:102,132).
//
//   conv0_fwd_kernel  : a0[pix][co] = sum_c xn[c][pix] * W[co][c] + b[co]
//                       A operand straight from HBM: for a fixed band c, 32 consecutive pixels of the
//                       NCHW tensor are one coalesced 128-B segment = the MFMA A fragment, so the input
//                       is read exactly once and never staged.  Output is written pixel-major
//                       (channel-last), the layout the 3x3 kernels consume.
//   conv0_wgrad_kernel: dW[c][co] = sum_pix xn[c][pix] * da0[pix][co]  (split over samples, partials reduced)
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ xn, const float* __restrict__ w,
                                                        const float* __restrict__ b, long long pstride,
                                                        float* __restrict__ a0, int n, int C, int HW) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // wT[Cp][65]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int net = blockIdx.y;
  const int Cp = (C + 1) & ~1;
  const float* W = w + (long long)net * pstride;
  for (int i = tid; i < C * 64; i += 256) {
    const int co = i / C, c = i - co * C;
    smem[c * 65 + co] = W[i];
  }
  if (Cp != C) for (int i = tid; i < 64; i += 256) smem[C * 65 + i] = 0.f;
  __syncthreads();

  const long long M = (long long)n * HW;
  const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + l31;
  const bool valid = m < M;
  const long long mm = valid ? m : 0;
  const int sample = (int)(mm / HW), pix = (int)(mm - (long long)sample * HW);
  const float* ap = xn + ((long long)net * n + sample) * C * HW + pix;
  f32x16 acc0 = zero16(), acc1 = zero16();
  const int KK = Cp >> 1;
#pragma unroll 8
  for (int kk = 0; kk < KK; ++kk) {
    const int c = 2 * kk + hh;
    const float a = (c < C) ? ap[(long long)c * HW] : 0.f;
    const float b0 = smem[c * 65 + l31], b1 = smem[c * 65 + 32 + l31];
    acc0 = mfma32(a, b0, acc0);
    acc1 = mfma32(a, b1, acc1);
  }
  const float* bias = b + (long long)net * pstride;
  const float bv0 = bias[l31], bv1 = bias[32 + l31];
  const long long mbase = ((long long)blockIdx.x * 4 + wave) * 32;
  float* out = a0 + (long long)net * M * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long mr = mbase + acc_row(r, lane);
    if (mr < M) {
      out[mr * 64 + l31] = acc0[r] + bv0;
      out[mr * 64 + 32 + l31] = acc1[r] + bv1;
    }
  }
}

hipError_t launch_conv0_fwd(int nets, int n, int C, int HW, const float* xn, const float* w, const float* b,
                            long long pstride, float* a0, hipStream_t st) {
  const long long M = (long long)n * HW;
  const size_t lds = (size_t)(((C + 1) & ~1) + 1) * 65 * 4;
  dim3 grid((unsigned)((M + 127) / 128), nets);
  hipLaunchKernelGGL(conv0_fwd_kernel, grid, dim3(256), lds, st, xn, w, b, pstride, a0, n, C, HW);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight gradient.  One workgroup walks `SPG` samples; per sample the [C][HW] slab of xn is staged
// in LDS (row stride odd => conflict-free column reads), da0 rows come straight from HBM/L2.
// wave w: co tile = w&1, band tiles (w>>1), (w>>1)+2, ...   (MAXT tiles of 32 bands per wave)
// ------------------------------------------------------------------------------------------
constexpr int C0_MAXT = 4;   // up to 8 band tiles = 256 input channels

__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ xn, const float* __restrict__ da0,
                                                          float* __restrict__ part, int n, int C, int HW, int G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // xs[Ct][HWp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int net = blockIdx.y, g = blockIdx.x;
  const int NT = (C + 31) >> 5, Ct = NT * 32;
  const int HWp = (HW + 2) | 1;             // odd, >= HW+1 (one zero pad column for odd HW)
  const int SPG = (n + G - 1) / G;
  const int sbeg = g * SPG, send = (sbeg + SPG < n) ? sbeg + SPG : n;
  const int ct = wave & 1, it0 = wave >> 1;
  f32x16 acc[C0_MAXT];
#pragma unroll
  for (int t = 0; t < C0_MAXT; ++t) acc[t] = zero16();
  float dbacc = 0.f;
  // rows >= C and columns >= HW stay zero for the whole kernel
  for (int i = tid; i < Ct * HWp; i += 256) smem[i] = 0.f;

  for (int s = sbeg; s < send; ++s) {
    __syncthreads();
    const float* xs = xn + ((long long)net * n + s) * C * HW;
    for (int i = tid; i < C * HW; i += 256) {
      const int c = i / HW, p = i - c * HW;
      smem[c * HWp + p] = xs[i];
    }
    __syncthreads();
    const float* brow = da0 + ((long long)net * n + s) * HW * 64 + ct * 32 + l31;
    const int pairs = (HW + 1) >> 1;
#pragma unroll 4
    for (int t = 0; t < pairs; ++t) {
      const int p = 2 * t + hh;
      const float b = (p < HW) ? brow[(long long)p * 64] : 0.f;
      dbacc += b;
#pragma unroll
      for (int q = 0; q < C0_MAXT; ++q) {
        const int tile = it0 + 2 * q;
        if (tile < NT) acc[q] = mfma32(smem[(tile * 32 + l31) * HWp + p], b, acc[q]);
      }
    }
  }
  // partial layout [c][co] (+ 64 db)
  float* pp = part + ((long long)net * G + g) * ((long long)Ct * 64 + 64);
#pragma unroll
  for (int q = 0; q < C0_MAXT; ++q) {
    const int tile = it0 + 2 * q;
    if (tile < NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pp[(tile * 32 + acc_row(r, lane)) * 64 + ct * 32 + l31] = acc[q][r];
    }
  }
  if (it0 == 0) {
    const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
    if (hh == 0) pp[(long long)Ct * 64 + ct * 32 + l31] = tot;
  }
}

__global__ void conv0_wgrad_reduce_kernel(const float* __restrict__ part, int G, int C, int Ct,
                                          float* __restrict__ dW, float* __restrict__ db, long long grad_ns) {
  const int net = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int PS = Ct * 64 + 64;
  if (e >= PS) return;
  const float* p = part + (long long)net * G * PS + e;
  float s0 = 0.f, s1 = 0.f;
  int g = 0;
  for (; g + 1 < G; g += 2) { s0 += p[(size_t)g * PS]; s1 += p[(size_t)(g + 1) * PS]; }
  if (g < G) s0 += p[(size_t)g * PS];
  const float sum = s0 + s1;
  if (e < Ct * 64) {
    const int c = e >> 6, co = e & 63;
    if (c < C) dW[(long long)net * grad_ns + co * C + c] = sum;
  } else {
    db[(long long)net * grad_ns + (e - Ct * 64)] = sum;
  }
}

int plan_conv0_wgrad_G(int n, int C, int HW) {
  (void)C; (void)HW;
  int G = n < 128 ? n : 128;     // per net; 2 nets -> up to 256 workgroups
  return G < 1 ? 1 : G;
}

hipError_t launch_conv0_wgrad(int nets, int n, int C, int HW, const float* xn, const float* da0, float* part,
                              float* dW, float* db, long long grad_ns, hipStream_t st) {
  const int NT = (C + 31) / 32, Ct = NT * 32;
  if (NT > 2 * C0_MAXT) return hipErrorInvalidValue;
  const int HWp = (HW + 2) | 1;
  const size_t lds = (size_t)Ct * HWp * 4;
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)conv0_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_MAX);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int G = plan_conv0_wgrad_G(n, C, HW);
  hipLaunchKernelGGL(conv0_wgrad_kernel, dim3(G, nets), dim3(256), lds, st, xn, da0, part, n, C, HW, G);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int PS = Ct * 64 + 64;
  hipLaunchKernelGGL(conv0_wgrad_reduce_kernel, dim3((PS + 255) / 256, nets), dim3(256), 0, st,
                     (const float*)part, G, C, Ct, dW, db, grad_ns);
  return hipGetLastError();
}

}  // namespace cmlpl
