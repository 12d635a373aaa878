// NT-Xent contrastive loss of the reference's tools.models.ContrastiveLoss (tools/models.py:14-39):
//   z = normalize([emb_i ; emb_j])   S = z z^T (cosine similarity)   positives on the +-B diagonals
//   loss = mean_a( -log( exp(S[a,p(a)]/T) / sum_{b != a} exp(S[a,b]/T) ) )
// forward + analytic backward.  S and the gradient products run on the fp32 MFMA through gemm_tn_kernel
// (k-major copies zT make both operands coalesced); the row pass is one wavefront per row with the
// denominator reduced by wavefront shuffles.
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

struct NtxArgs {
  const float* ei; const float* ej; int B, D; float T;
  float *z, *zT, *norm, *S, *G, *GT, *rowloss, *dz1, *dz2, *loss, *gi, *gj;
};

// one wave per row: z = x / max(||x||, 1e-12) (F.normalize), plus the k-major copy zT[d][r]
__global__ __launch_bounds__(256) void ntx_normalize_kernel(NtxArgs a) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6), N2 = 2 * a.B, D = a.D;
  if (r >= N2) return;
  const float* x = (r < a.B) ? a.ei + (long long)r * D : a.ej + (long long)(r - a.B) * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[d]; ss = fmaf(v, v, ss); }
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  if (lane == 0) a.norm[r] = nrm;
  for (int d = lane; d < D; d += 64) {
    const float v = x[d] / nrm;
    a.z[(long long)r * D + d] = v;
    a.zT[(long long)d * N2 + r] = v;
  }
}

// one wave per row a: denominator over b != a by shuffles, loss_a, G = dL/dS (and its transpose)
__global__ __launch_bounds__(256) void ntx_rows_kernel(NtxArgs a) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6), N2 = 2 * a.B;
  if (r >= N2) return;
  const float* Sr = a.S + (long long)r * N2;
  const int p = (r < a.B) ? r + a.B : r - a.B;
  float den = 0.f;
  for (int b = lane; b < N2; b += 64) den += (b != r) ? expf(Sr[b] / a.T) : 0.f;
  den = wave_sum(den);
  const float sp = Sr[p];
  if (lane == 0) a.rowloss[r] = -logf(expf(sp / a.T) / den);
  const float sc = 1.f / (a.T * (float)N2);
  for (int b = lane; b < N2; b += 64) {
    const float g = (b == r) ? 0.f : (expf(Sr[b] / a.T) / den - (b == p ? 1.f : 0.f)) * sc;
    a.G[(long long)r * N2 + b] = g;
    a.GT[(long long)b * N2 + r] = g;
  }
}

// one wave per row: dz = dz1 + dz2 (= (G + G^T) z), back through the normalisation; row 0's wave also sums the loss
__global__ __launch_bounds__(256) void ntx_embgrad_kernel(NtxArgs a) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6), N2 = 2 * a.B, D = a.D;
  if (r >= N2) return;
  const float* z = a.z + (long long)r * D;
  const float* t1 = a.dz1 + (long long)r * D;
  const float* t2 = a.dz2 + (long long)r * D;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot = fmaf(z[d], t1[d] + t2[d], dot);
  dot = wave_sum(dot);
  const float nrm = a.norm[r];
  float* g = (r < a.B) ? a.gi + (long long)r * D : a.gj + (long long)(r - a.B) * D;
  for (int d = lane; d < D; d += 64) g[d] = ((t1[d] + t2[d]) - z[d] * dot) / nrm;
  if (r == 0) {
    float s = 0.f;
    for (int b = lane; b < N2; b += 64) s += a.rowloss[b];
    s = wave_sum(s);
    if (lane == 0) a.loss[0] = s / (float)N2;
  }
}

size_t ntxent_ws_floats(int B, int D) {
  const size_t N2 = 2 * (size_t)B;
  return 4 * N2 * D + 3 * N2 * N2 + 2 * N2 + 64;
}

hipError_t launch_ntxent(const float* ei, const float* ej, int B, int D, float T, float* loss, float* gi, float* gj,
                         float* ws, hipStream_t st) {
  const size_t N2 = 2 * (size_t)B;
  NtxArgs a;
  a.ei = ei; a.ej = ej; a.B = B; a.D = D; a.T = T; a.loss = loss; a.gi = gi; a.gj = gj;
  a.z = ws; ws += N2 * D;
  a.zT = ws; ws += N2 * D;
  a.dz1 = ws; ws += N2 * D;
  a.dz2 = ws; ws += N2 * D;
  a.S = ws; ws += N2 * N2;
  a.G = ws; ws += N2 * N2;
  a.GT = ws; ws += N2 * N2;
  a.norm = ws; ws += N2;
  a.rowloss = ws;
  const dim3 rows((unsigned)((N2 + 3) / 4));
  hipLaunchKernelGGL(ntx_normalize_kernel, rows, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  GemmTN g;
  g.A = a.zT; g.lda = (int)N2; g.M = (int)N2; g.B = a.zT; g.ldb = (int)N2; g.N = (int)N2; g.R = D;
  g.C = a.S; g.ldc = (int)N2; g.a_bstride = g.b_bstride = g.c_bstride = 0; g.bias = nullptr; g.bias_bstride = 0;
  g.bias_in = nullptr; g.bias_in_bstride = 0; g.relu = 0; g.batches = 1; g.scale = 1.f;
  if ((e = launch_gemm_tn(g, st)) != hipSuccess) return e;                  // S = z z^T
  hipLaunchKernelGGL(ntx_rows_kernel, rows, dim3(256), 0, st, a);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  GemmTN h = g;
  g.A = a.GT; g.B = a.z; g.ldb = D; g.N = D; g.R = (int)N2; g.C = a.dz1; g.ldc = D;   // dz1 = G z
  h.A = a.G;  h.B = a.z; h.ldb = D; h.N = D; h.R = (int)N2; h.C = a.dz2; h.ldc = D;   // dz2 = G^T z
  if ((e = launch_gemm_tn2(g, h, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(ntx_embgrad_kernel, rows, dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace cmlpl
