// Dense contractions of the spectral branch / classifier / loss gradients on the fp32 MFMA.
//
//   spe_fwd_kernel : y = relu(sn . Wspe^T + b)   (feat_spe + ReLU, tools/models.py:142-143)
//   gemm_tn_kernel : C[i][j] = scale * sum_r A[r][i] * B[r][j]  (+ column sums of A)
//                    both operands are read straight from HBM/L2 as MFMA fragments: with the
//                    reduction index r as the slow dimension, 32 consecutive i (or j) are one
//                    coalesced 128-B segment.  Used for dW_spe, dW_cls and the two contrastive
//                    feature gradients (G^T.f_w, G.f_s).
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

// workgroup = 32 rows x 128 outputs; operands staged in LDS with odd row stride.
__global__ __launch_bounds__(256) void spe_fwd_kernel(const float* __restrict__ sn, const float* __restrict__ w,
                                                      const float* __restrict__ b, long long pstride,
                                                      float* __restrict__ y, int n, int bands) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int net = blockIdx.z, r0 = blockIdx.x * 32, o0 = blockIdx.y * 128;
  const int BP = (bands + 2) | 1;                   // odd stride, >= bands+1 (zero pad for odd bands)
  float* As = smem;                                 // [32][BP]
  float* Bs = smem + 32 * BP;                       // [128][BP]
  const float* S = sn + (long long)net * n * bands;
  const float* W = w + (long long)net * pstride;
  for (int i = tid; i < 32 * BP; i += 256) {
    const int r = i / BP, k = i - r * BP;
    As[i] = (k < bands && r0 + r < n) ? S[(long long)(r0 + r) * bands + k] : 0.f;
  }
  for (int i = tid; i < 128 * BP; i += 256) {
    const int o = i / BP, k = i - o * BP;
    Bs[i] = (k < bands) ? W[(long long)(o0 + o) * bands + k] : 0.f;
  }
  __syncthreads();
  f32x16 acc = zero16();
  const float* ar = As + l31 * BP + hh;
  const float* br = Bs + (wave * 32 + l31) * BP + hh;
  const int KK = (bands + 1) >> 1;
#pragma unroll 4
  for (int kk = 0; kk < KK; ++kk) acc = mfma32(ar[2 * kk], br[2 * kk], acc);
  const int o = o0 + wave * 32 + l31;
  const float bv = b[(long long)net * pstride + o];
  float* Y = y + (long long)net * n * FD;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + acc_row(r, lane);
    if (row < n) Y[(long long)row * FD + o] = fmaxf(acc[r] + bv, 0.f);
  }
}

hipError_t launch_spe_fwd(int nets, int n, int bands, const float* sn, const float* w, const float* b,
                          long long pstride, float* y, hipStream_t st) {
  const int BP = (bands + 2) | 1;
  const size_t lds = (size_t)160 * BP * 4;
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)spe_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_MAX);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  dim3 grid((n + 31) / 32, FD / 128, nets);
  hipLaunchKernelGGL(spe_fwd_kernel, grid, dim3(256), lds, st, sn, w, b, pstride, y, n, bands);
  return hipGetLastError();
}

// one wave per 32x32 output tile; 4 waves of a workgroup share the M tile (consecutive N tiles)
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int bz = blockIdx.z;
  const int mt = blockIdx.y, nt = blockIdx.x * 4 + wave;
  const int NT = (g.N + 31) >> 5;
  if (nt >= NT) return;
  const float* A = g.A + (long long)bz * g.a_bstride;
  const float* B = g.B + (long long)bz * g.b_bstride;
  const int i = mt * 32 + l31, j = nt * 32 + l31;
  const bool iv = i < g.M, jv = j < g.N;
  const float* ap = A + (iv ? i : 0);
  const float* bp = B + (jv ? j : 0);
  f32x16 acc = zero16();
  float asum = 0.f;
  const int R = g.R, pairs = (R + 1) >> 1;
#pragma unroll 8
  for (int t = 0; t < pairs; ++t) {
    const int r = 2 * t + hh;
    const bool rv = r < R;
    const float a = (rv && iv) ? ap[(long long)r * g.lda] : 0.f;
    const float b = (rv && jv) ? bp[(long long)r * g.ldb] : 0.f;
    asum += a;
    acc = mfma32(a, b, acc);
  }
  float* C = g.C + (long long)bz * g.c_bstride;
  if (jv) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + acc_row(r, lane);
      if (row < g.M) C[(long long)row * g.ldc + j] = acc[r] * g.scale;
    }
  }
  if (g.bias != nullptr && nt == 0) {
    const float tot = asum + __shfl_xor(asum, 32, 64);
    if (hh == 0 && iv) g.bias[(long long)bz * g.bias_bstride + i] = tot * g.scale;
  }
}

hipError_t launch_gemm_tn(const GemmTN& g, hipStream_t st) {
  const int MT = (g.M + 31) / 32, NT = (g.N + 31) / 32;
  dim3 grid((NT + 3) / 4, MT, g.batches);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, st, g);
  return hipGetLastError();
}

}  // namespace cmlpl
