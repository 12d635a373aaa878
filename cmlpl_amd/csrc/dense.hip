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
#include "gemm_tn.hpp"

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
  staged_copy<8, float>(32 * BP, tid,
      [&](int i) { const int r = i / BP, k = i - r * BP; const bool ok = (k < bands) && (r0 + r < n);
                   const float v = S[ok ? (long long)(r0 + r) * bands + k : 0]; return ok ? v : 0.f; },
      [&](int i, float v) { As[i] = v; });
  staged_copy<16, float>(128 * BP, tid,
      [&](int i) { const int o = i / BP, k = i - o * BP; const bool ok = k < bands;
                   const float v = W[(long long)(o0 + o) * bands + (ok ? k : 0)]; return ok ? v : 0.f; },
      [&](int i, float v) { Bs[i] = v; });
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
    if (row < n) Y[(long long)row * FD + o] = relu_nan(acc[r] + bv);
  }
}

hipError_t launch_spe_fwd(int nets, int n, int bands, const float* sn, const float* w, const float* b,
                          long long pstride, float* y, hipStream_t st) {
  const int BP = (bands + 2) | 1;
  const size_t lds = (size_t)160 * BP * 4;
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static DevOnce attr_once;
  {
    hipError_t e = ensure_max_lds(attr_once, spe_fwd_kernel);
    if (e != hipSuccess) return e;
  }
  dim3 grid((n + 31) / 32, FD / 128, nets);
  hipLaunchKernelGGL(spe_fwd_kernel, grid, dim3(256), lds, st, sn, w, b, pstride, y, n, bands);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN2 t) {
  __shared__ GemmTNShared sh;
  gemm_tn_block(t, (int)blockIdx.x, sh);
}

static int gemm_blocks(const GemmTN& g) { return gemm_tn_blocks(g); }

hipError_t launch_gemm_tn(const GemmTN& g, hipStream_t st) {
  GemmTN2 t;
  t.p[0] = g; t.p[1] = g; t.nblk0 = gemm_blocks(g);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(t.nblk0), dim3(256), 0, st, t);
  return hipGetLastError();
}

hipError_t launch_gemm_tn2(const GemmTN& g0, const GemmTN& g1, hipStream_t st) {
  GemmTN2 t;
  t.p[0] = g0; t.p[1] = g1; t.nblk0 = gemm_blocks(g0);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(t.nblk0 + gemm_blocks(g1)), dim3(256), 0, st, t);
  return hipGetLastError();
}

}  // namespace cmlpl
