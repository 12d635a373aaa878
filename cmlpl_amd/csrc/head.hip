// Classifier head of BaseNet2 (tools/models.py:141,144-150): flatten + concat + dropout + Linear,
// and the L2 normalisation of the spectral feature (Normalize, models.py:87-90, no epsilon).
// One wavefront per sample: the 64*H4*W4 + 1024 wide row lives in that wave's LDS slice and the
// K class dot products are reduced with wavefront shuffles.
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

struct HeadFwdArgs {
  const float* p2; const float* y; const float* dropmask; float* dropgen;
  const float* wc; const float* bc; long long pstride;
  float* catd; float* ynorm; float* logits; float* feat;
  float dropout_p; int train; uint64_t seed, step;
  int n, HW4, K;
};

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = blockIdx.y, sample = blockIdx.x * 4 + wave;
  if (sample >= a.n) return;  // whole wave; no block-level sync below
  const int SF = a.HW4 * 64, F = SF + FD, K = a.K;
  float* row = smem + (size_t)wave * F;
  const long long rs = (long long)net * a.n + sample;
  const float* p2 = a.p2 + rs * SF;
  const float* y = a.y + rs * FD;
  float* catd = a.catd + rs * F;
  // 0 = none, 1 = explicit mask, 2 = generate (Philox) and record for the backward pass
  const int dmode = (!a.train || a.dropout_p <= 0.f) ? 0 : (a.dropmask != nullptr ? 1 : 2);
  const float* dm = (dmode == 1) ? a.dropmask + rs * F : nullptr;
  float* dg = (dmode == 2) ? a.dropgen + rs * F : nullptr;
  const float keep_scale = 1.0f / (1.0f - a.dropout_p);
  float ss = 0.f;
  for (int f = lane; f < F; f += 64) {
    float v;
    if (f < SF) {
      const int c = f / a.HW4, hw = f - c * a.HW4;   // canonical NCHW flatten order (x.view, models.py:141)
      v = p2[hw * 64 + c];
    } else {
      v = y[f - SF];
      ss += v * v;
    }
    if (dmode == 1) v *= dm[f];
    else if (dmode == 2) {
      const unsigned long long e = (unsigned long long)sample * F + f;
      const float4 u = philox_uniform4(a.seed, a.step, STREAM_DROPOUT + net, e >> 2);
      const float uu = ((e & 3) == 0) ? u.x : ((e & 3) == 1) ? u.y : ((e & 3) == 2) ? u.z : u.w;
      const float mlt = (uu >= a.dropout_p) ? keep_scale : 0.f;
      dg[f] = mlt;
      v *= mlt;
    }
    row[f] = v;
    catd[f] = v;
  }
  ss = wave_sum(ss);
  const float norm = sqrtf(ss);
  if (lane == 0) a.ynorm[rs] = norm;
  float* feat = a.feat + rs * FD;
  for (int j = lane; j < FD; j += 64) feat[j] = y[j] / norm;

  const float* wc = a.wc + (long long)net * a.pstride;
  const float* bc = a.bc + (long long)net * a.pstride;
  for (int kc = 0; kc < K; kc += 4) {
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const float* w0 = wc + (long long)(kc) * F;
    const float* w1 = wc + (long long)((kc + 1 < K) ? kc + 1 : K - 1) * F;
    const float* w2 = wc + (long long)((kc + 2 < K) ? kc + 2 : K - 1) * F;
    const float* w3 = wc + (long long)((kc + 3 < K) ? kc + 3 : K - 1) * F;
    for (int f = lane; f < F; f += 64) {
      const float c = row[f];
      acc0 = fmaf(c, w0[f], acc0);
      acc1 = fmaf(c, w1[f], acc1);
      acc2 = fmaf(c, w2[f], acc2);
      acc3 = fmaf(c, w3[f], acc3);
    }
    acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2); acc3 = wave_sum(acc3);
    if (lane == 0) {
      float* lo = a.logits + rs * K;
      lo[kc] = acc0 + bc[kc];
      if (kc + 1 < K) lo[kc + 1] = acc1 + bc[kc + 1];
      if (kc + 2 < K) lo[kc + 2] = acc2 + bc[kc + 2];
      if (kc + 3 < K) lo[kc + 3] = acc3 + bc[kc + 3];
    }
  }
}

hipError_t launch_head_fwd(int nets, int n, int HW4, int K, const float* p2, const float* y, const float* dropmask,
                           float* dropgen, float dropout_p, int train, uint64_t seed, uint64_t step,
                           const float* wc, const float* bc, long long pstride,
                           float* catd, float* ynorm, float* logits, float* feat, hipStream_t st) {
  HeadFwdArgs a;
  a.p2 = p2; a.y = y; a.dropmask = dropmask; a.dropgen = dropgen; a.wc = wc; a.bc = bc; a.pstride = pstride;
  a.catd = catd; a.ynorm = ynorm; a.logits = logits; a.feat = feat;
  a.dropout_p = dropout_p; a.train = train; a.seed = seed; a.step = step; a.n = n; a.HW4 = HW4; a.K = K;
  const size_t lds = (size_t)4 * (HW4 * 64 + FD) * 4;
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)head_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_MAX);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(head_fwd_kernel, dim3((n + 3) / 4, nets), dim3(256), lds, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// backward: dcat = (dlogits . Wc) * dropmask ; spatial part scattered back to [hw][c];
// spectral part joined with the gradient through the L2 normalisation and the ReLU mask:
//   dy = (y > 0) * ( dcat_y + (dfeat - feat * <feat, dfeat>) / ||y|| )
// ------------------------------------------------------------------------------------------
struct HeadBwdArgs {
  const float* dlogits; const float* dfeat; const float* dropmask; const float* wc; long long pstride;
  const float* y; const float* ynorm;
  float* dy; float* dp2;
  int n, HW4, K;
};

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
  __shared__ float dls[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = blockIdx.y, sample = blockIdx.x * 4 + wave;
  if (sample >= a.n) return;
  const int SF = a.HW4 * 64, F = SF + FD, K = a.K;
  const long long rs = (long long)net * a.n + sample;
  dls[wave][lane] = (lane < K) ? a.dlogits[rs * K + lane] : 0.f;
  const float* y = a.y + rs * FD;
  const float* df = (a.dfeat != nullptr) ? a.dfeat + rs * FD : nullptr;
  const float norm = a.ynorm[rs];
  float dot = 0.f;   // <feat, dfeat> with feat = y / ||y|| (same division as the forward pass)
  if (df != nullptr) {
    for (int j = lane; j < FD; j += 64) dot = fmaf(y[j] / norm, df[j], dot);
    dot = wave_sum(dot);
  }
  const float* wc = a.wc + (long long)net * a.pstride;
  const float* dm = (a.dropmask != nullptr) ? a.dropmask + rs * F : nullptr;
  float* dp2 = a.dp2 + rs * SF;
  float* dy = a.dy + rs * FD;
  __builtin_amdgcn_wave_barrier();
  for (int f = lane; f < F; f += 64) {
    float dc = 0.f;
    for (int k = 0; k < K; ++k) dc = fmaf(dls[wave][k], wc[(long long)k * F + f], dc);
    if (dm != nullptr) dc *= dm[f];
    if (f < SF) {
      const int c = f / a.HW4, hw = f - c * a.HW4;
      dp2[hw * 64 + c] = dc;
    } else {
      const int j = f - SF;
      float g = dc;
      if (df != nullptr) g += (df[j] - (y[j] / norm) * dot) / norm;
      dy[j] = (y[j] > 0.f) ? g : 0.f;
    }
  }
}

hipError_t launch_head_bwd(int nets, int n, int HW4, int K, const float* dlogits, const float* dfeat,
                           const float* dropmask, const float* wc, long long pstride,
                           const float* y, const float* ynorm,
                           float* dy, float* dp2, hipStream_t st) {
  HeadBwdArgs a;
  a.dlogits = dlogits; a.dfeat = dfeat; a.dropmask = dropmask; a.wc = wc; a.pstride = pstride;
  a.y = y; a.ynorm = ynorm; a.dy = dy; a.dp2 = dp2; a.n = n; a.HW4 = HW4; a.K = K;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((n + 3) / 4, nets), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace cmlpl
