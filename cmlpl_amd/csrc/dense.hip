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

// One workgroup per 32x32 output tile; the reduction index r is split over the 4 waves (each takes a
// contiguous quarter), every wave keeps 16 operand pairs (32 loads) in flight, and the four partial
// accumulators are folded through LDS.  These GEMMs are tiny (R = batch rows): what matters is the
// number of dependent memory round trips per wave, which this shape cuts to R/128.
constexpr int GT_DEPTH = 16;

struct GemmTN2 { GemmTN p[2]; int nblk0; };

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN2 t) {
  __shared__ float red[3][16][64];
  __shared__ float ared[4][64];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // linear block id -> (problem, nt, mt, batch)
  const int pi = ((int)blockIdx.x >= t.nblk0) ? 1 : 0;
  const GemmTN g = t.p[pi];
  int b = (int)blockIdx.x - (pi ? t.nblk0 : 0);
  const int NTg = (g.N + 31) >> 5, MTg = (g.M + 31) >> 5;
  const int nt = b % NTg; b /= NTg;
  const int mt = b % MTg;
  const int bz = b / MTg;
  const float* A = g.A + (long long)bz * g.a_bstride;
  const float* B = g.B + (long long)bz * g.b_bstride;
  const int i = mt * 32 + l31, j = nt * 32 + l31;
  const bool iv = i < g.M, jv = j < g.N;
  const float* ap = A + (iv ? i : 0);
  const float* bp = B + (jv ? j : 0);
  const int R = g.R;
  const int pairs = (R + 1) >> 1;
  const int ppw = (pairs + 3) >> 2;                    // pairs per wave
  const int t0 = wave * ppw, t1 = (t0 + ppw < pairs) ? t0 + ppw : pairs;
  f32x16 acc = zero16();
  float asum = 0.f;
  for (int tb = t0; tb < t1; tb += GT_DEPTH) {
    float av[GT_DEPTH], bv[GT_DEPTH];
#pragma unroll
    for (int q = 0; q < GT_DEPTH; ++q) {
      const int r = 2 * (tb + q) + hh;
      const bool rv = (tb + q < t1) && (r < R);
      const int rc = rv ? r : 0;
      const float a = ap[(long long)rc * g.lda], b = bp[(long long)rc * g.ldb];
      av[q] = (rv && iv) ? a : 0.f;
      bv[q] = (rv && jv) ? b : 0.f;
    }
#pragma unroll
    for (int q = 0; q < GT_DEPTH; ++q) {
      asum += av[q];
      acc = mfma32(av[q], bv[q], acc);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  ared[wave][lane] = asum;
  __syncthreads();
  if (wave == 0) {
    float* C = g.C + (long long)bz * g.c_bstride;
    if (jv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + acc_row(r, lane);
        float v = (((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane]) * g.scale;
        if (g.bias_in != nullptr) v += g.bias_in[(long long)bz * g.bias_in_bstride + j];
        if (g.relu) v = relu_nan(v);
        if (row < g.M) C[(long long)row * g.ldc + j] = v;
      }
    }
    if (g.bias != nullptr && nt == 0) {
      float tot = (ared[0][lane] + ared[1][lane]) + (ared[2][lane] + ared[3][lane]);
      tot += __shfl_xor(tot, 32, 64);
      if (hh == 0 && iv) g.bias[(long long)bz * g.bias_bstride + i] = tot * g.scale;
    }
  }
}

static int gemm_blocks(const GemmTN& g) { return ((g.M + 31) / 32) * ((g.N + 31) / 32) * g.batches; }

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
