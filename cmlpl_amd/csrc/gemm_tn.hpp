// The small transposed-A GEMM block shared by several launches (dense.hip: spectral forward and weight gradients;
// loss.hip: the contrastive feature gradients + the scalar block; conv0.hip: weight-gradient reduce + GEMMs).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

// One workgroup per 32x32 output tile; the reduction index r is split over the 4 waves (each takes a
// contiguous quarter), every wave keeps 16 operand pairs (32 loads) in flight, and the four partial
// accumulators are folded through LDS.  These GEMMs are tiny (R = batch rows): what matters is the
// number of dependent memory round trips per wave, which this shape cuts to R/128.
constexpr int GT_DEPTH = 16;

struct GemmTN2 { GemmTN p[2]; int nblk0; };
struct GemmTNShared { __attribute__((aligned(16))) float red[3][16][64]; float ared[4][64]; };

__device__ __forceinline__ void gemm_tn_block(const GemmTN2& t, int bid, GemmTNShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // linear block id -> (problem, nt, mt, batch)
  const int pi = (bid >= t.nblk0) ? 1 : 0;
  const GemmTN g = t.p[pi];
  int b = bid - (pi ? t.nblk0 : 0);
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
    // segmented B (rows spread over rank-major blocks): one division per batch, then carried (rows advance by 2)
    int seg = 0, off = 0;
    if (g.b_seg_rows > 0) { const int r0 = 2 * tb + hh; seg = r0 / g.b_seg_rows; off = r0 - seg * g.b_seg_rows; }
#pragma unroll
    for (int q = 0; q < GT_DEPTH; ++q) {
      const int r = 2 * (tb + q) + hh;
      const bool rv = (tb + q < t1) && (r < R);
      const int rc = rv ? r : 0;
      const long long bo = g.b_seg_rows > 0 ? (rv ? (long long)seg * g.b_seg_stride + (long long)off * g.ldb : 0)
                                            : (long long)rc * g.ldb;
      const float a = ap[(long long)rc * g.lda], b = bp[bo];
      av[q] = (rv && iv) ? a : 0.f;
      bv[q] = (rv && jv) ? b : 0.f;
      off += 2;
      while (g.b_seg_rows > 0 && off >= g.b_seg_rows) { off -= g.b_seg_rows; ++seg; }
    }
#pragma unroll
    for (int q = 0; q < GT_DEPTH; ++q) {
      asum += av[q];
      acc = mfma32(av[q], bv[q], acc);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sh.red[wave - 1][r][lane] = acc[r];
  }
  sh.ared[wave][lane] = asum;
  __syncthreads();
  if (wave == 0) {
    float* C = g.C + (long long)bz * g.c_bstride;
    if (jv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + acc_row(r, lane);
        float v = (((acc[r] + sh.red[0][r][lane]) + sh.red[1][r][lane]) + sh.red[2][r][lane]) * g.scale;
        if (g.bias_in != nullptr) v += g.bias_in[(long long)bz * g.bias_in_bstride + j];
        if (g.relu) v = relu_nan(v);
        if (row < g.M) C[(long long)row * g.ldc + j] = v;
      }
    }
    if (g.bias != nullptr && nt == 0) {
      float tot = (sh.ared[0][lane] + sh.ared[1][lane]) + (sh.ared[2][lane] + sh.ared[3][lane]);
      tot += __shfl_xor(tot, 32, 64);
      if (hh == 0 && iv) g.bias[(long long)bz * g.bias_bstride + i] = tot * g.scale;
    }
  }
}

inline int gemm_tn_blocks(const GemmTN& g) { return ((g.M + 31) / 32) * ((g.N + 31) / 32) * g.batches; }

}  // namespace cmlpl
