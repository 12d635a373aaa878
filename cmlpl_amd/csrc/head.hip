// Classifier head of BaseNet2 (tools/models.py:141,144-150): flatten + concat + dropout + Linear,
// and the L2 normalisation of the spectral feature (Normalize, models.py:87-90, no epsilon).
// One 256-thread workgroup per sample: the 64*H4*W4 + 1024 wide row is built once in LDS with all
// loads of a thread in flight together; each wavefront then owns a subset of the K classes and reduces
// its dot products with wavefront shuffles (no block-level reduction on the logits path).
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

struct HeadFwdArgs {
  const float* p2; const float* y; const float* dropmask; float* dropgen;
  const float* wc; const float* bc; long long pstride;
  float* catd; float* ynorm; float* logits; float* feat;
  float dropout_p; int train; uint64_t seed, step;
  int n, HW4, K;
  int nlab, lab0, unl_base;   // local row -> GLOBAL sample index (Philox key independent of sharding)
  DynRef dyn;                 // the step counter from device memory (graph replay), or null
};

constexpr int HEAD_MAXQ4 = 3;   // ceil(F / 1024) <= 3  (F <= 3072)

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // row[F] + red[4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, sample = blockIdx.x;
  const int SF = a.HW4 * 64, F = SF + FD, K = a.K;
  float* row = smem;
  float* red = smem + F;
  const long long rs = (long long)net * a.n + sample;
  const float* p2 = a.p2 + rs * SF;
  const float* y = a.y + rs * FD;
  float* catd = a.catd + rs * F;
  // 0 = none, 1 = explicit mask, 2 = generate (Philox) and record for the backward pass
  const int dmode = (!a.train || a.dropout_p <= 0.f) ? 0 : (a.dropmask != nullptr ? 1 : 2);
  const float* dm = (dmode == 1) ? a.dropmask + rs * F : nullptr;
  float* dg = (dmode == 2) ? a.dropgen + rs * F : nullptr;
  const float keep_scale = 1.0f / (1.0f - a.dropout_p);
  // thread t owns the 4 consecutive row elements f = 1024*q + 4*t .. +3 (F is a multiple of 4): float4 traffic
  // to catd / dropmask / LDS, and ONE Philox block per 4 dropout decisions
  const int NQ = (F + 1023) >> 10;
  float4 v[HEAD_MAXQ4];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < HEAD_MAXQ4; ++q) {
    const int f0 = 1024 * q + 4 * tid;
    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < NQ && f0 < F) {
      if (f0 >= SF) {
        v[q] = *(const float4*)(y + (f0 - SF));
      } else {   // canonical NCHW flatten order (x.view, models.py:141): f = c*HW4 + hw
        float t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = f0 + j, c = f / a.HW4, hw = f - c * a.HW4;
          t4[j] = p2[hw * 64 + c];
        }
        v[q] = make_float4(t4[0], t4[1], t4[2], t4[3]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < HEAD_MAXQ4; ++q) {
    const int f0 = 1024 * q + 4 * tid;
    if (q < NQ && f0 < F) {
      float4 x = v[q];
      if (f0 >= SF) ss += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
      if (dmode == 1) {
        const float4 m4 = *(const float4*)(dm + f0);
        x.x *= m4.x; x.y *= m4.y; x.z *= m4.z; x.w *= m4.w;
      } else if (dmode == 2) {
        const unsigned long long gs = (sample < a.nlab) ? a.lab0 + sample : a.unl_base + (sample - a.nlab);
        uint64_t rstep = a.step;
        const cmlpl_dyn* dynr = dyn_row(a.dyn);
        if (dynr != nullptr) rstep = (uint64_t)uni64((long long)dynr->step);
        const float4 u = philox_uniform4(a.seed, rstep, STREAM_DROPOUT + net, (gs * F + f0) >> 2);
        float4 m4;
        m4.x = (u.x >= a.dropout_p) ? keep_scale : 0.f; m4.y = (u.y >= a.dropout_p) ? keep_scale : 0.f;
        m4.z = (u.z >= a.dropout_p) ? keep_scale : 0.f; m4.w = (u.w >= a.dropout_p) ? keep_scale : 0.f;
        *(float4*)(dg + f0) = m4;
        x.x *= m4.x; x.y *= m4.y; x.z *= m4.z; x.w *= m4.w;
      }
      *(float4*)(row + f0) = x;
      *(float4*)(catd + f0) = x;
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  if (a.feat != nullptr) {     // (null: feat_norm_kernel formed the embeddings and the norm behind the spectral branch)
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (tid == 0) a.ynorm[rs] = norm;
    float* feat = a.feat + rs * FD;
    // the spectral part of this thread's slice is still in v[] (pre-dropout)
#pragma unroll
    for (int q = 0; q < HEAD_MAXQ4; ++q) {
      const int f0 = 1024 * q + 4 * tid;
      if (q < NQ && f0 >= SF && f0 < F) {
        float4 o = v[q];
        o.x /= norm; o.y /= norm; o.z /= norm; o.w /= norm;
        *(float4*)(feat + (f0 - SF)) = o;
      }
    }
  }
  // logits: wave w takes the quarter [w*F4, (w+1)*F4) of the row for ALL classes (8 accumulators at a time,
  // every weight load independent), then the four partial dot products meet in LDS
  const float* wc = a.wc + (long long)net * a.pstride;
  const float* bc = a.bc + (long long)net * a.pstride;
  float* part = red + 4;                       // [4 waves][64 classes]
  const int F4 = (F + 3) >> 2, f0 = wave * F4, f1 = (f0 + F4 < F) ? f0 + F4 : F;
  for (int kc = 0; kc < K; kc += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // four features x eight classes per lane and trip: 32 weight loads in flight (one feature per trip made the
    // 2624-wide row of the 20 x 20 window a chain of 11 memory round trips per class chunk: 17.7 us per launch)
    for (int fb = f0 + lane; fb < f1; fb += 256) {
      float x[4], wv[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = fb + 64 * i;
        const bool ok = f < f1;
        x[i] = ok ? row[f] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[i][j] = wc[(long long)((kc + j < K) ? kc + j : K - 1) * F + (ok ? f : f0)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(x[i], wv[i][j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = wave_sum(acc[j]);
      if (lane == 0 && kc + j < K) part[wave * 64 + kc + j] = t;
    }
  }
  __syncthreads();
  if (tid < K) a.logits[rs * K + tid] = ((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) + bc[tid];
}

hipError_t launch_head_fwd(int nets, int n, int HW4, int K, const float* p2, const float* y, const float* dropmask,
                           float* dropgen, float dropout_p, int train, uint64_t seed, uint64_t step,
                           int nlab, int lab0, int unl_base,
                           const float* wc, const float* bc, long long pstride,
                           float* catd, float* ynorm, float* logits, float* feat, hipStream_t st, DynRef dyn) {
  HeadFwdArgs a;
  a.dyn = dyn;
  a.p2 = p2; a.y = y; a.dropmask = dropmask; a.dropgen = dropgen; a.wc = wc; a.bc = bc; a.pstride = pstride;
  a.catd = catd; a.ynorm = ynorm; a.logits = logits; a.feat = feat;
  a.dropout_p = dropout_p; a.train = train; a.seed = seed; a.step = step; a.n = n; a.HW4 = HW4; a.K = K;
  a.nlab = nlab; a.lab0 = lab0; a.unl_base = unl_base;
  const int F = HW4 * 64 + FD;
  if (F > 1024 * HEAD_MAXQ4 || (F & 3)) return hipErrorInvalidValue;
  const size_t lds = (size_t)(F + 4 + 256) * 4;
  hipLaunchKernelGGL(head_fwd_kernel, dim3(n, nets), dim3(256), lds, st, a);
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
  __shared__ float dls[64];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, sample = blockIdx.x;
  const int SF = a.HW4 * 64, F = SF + FD, K = a.K;
  const long long rs = (long long)net * a.n + sample;
  if (tid < 64) dls[tid] = (tid < K) ? a.dlogits[rs * K + tid] : 0.f;
  const float* y = a.y + rs * FD;
  const float* df = (a.dfeat != nullptr) ? a.dfeat + rs * FD : nullptr;
  const float norm = a.ynorm[rs];
  // this thread's 4 spectral elements j = tid + 256*q
  float yv[4], dv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    yv[q] = y[tid + 256 * q];
    dv[q] = (df != nullptr) ? df[tid + 256 * q] : 0.f;
  }
  float dot = 0.f;   // <feat, dfeat> with feat = y / ||y|| (same division as the forward pass)
#pragma unroll
  for (int q = 0; q < 4; ++q) dot = fmaf(yv[q] / norm, dv[q], dot);
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  const float* wc = a.wc + (long long)net * a.pstride;
  const float* dm = (a.dropmask != nullptr) ? a.dropmask + rs * F : nullptr;
  float* dp2 = a.dp2 + rs * SF;
  float* dy = a.dy + rs * FD;
  // four row elements x sixteen classes per thread and trip: every load of a trip (64 classifier weights, the dropout
  // multipliers, the spectral values) is requested before the first is used (one element per trip with its K strided
  // loads in groups of four was a chain of ~30 round trips on the 2624-wide row: 14.7 us per launch)
  for (int fb = tid; fb < F; fb += 1024) {
    float dc[4] = {0.f, 0.f, 0.f, 0.f}, dmv[4], yj[4], dfj[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = fb + 256 * i;
      const bool ok = f < F, spec = ok && f >= SF;
      dmv[i] = (dm != nullptr && ok) ? dm[f] : 1.f;
      yj[i] = spec ? y[f - SF] : 0.f;
      dfj[i] = (spec && df != nullptr) ? df[f - SF] : 0.f;
    }
    for (int kc = 0; kc < K; kc += 16) {
      float wv[4][16];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = fb + 256 * i;
        const float* wf = wc + (f < F ? f : 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) wv[i][j] = wf[(long long)((kc + j < K) ? kc + j : K - 1) * F];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float dl = dls[kc + j < 64 ? kc + j : 63];          // zero beyond K
#pragma unroll
        for (int i = 0; i < 4; ++i) dc[i] = fmaf((kc + j < K) ? dl : 0.f, wv[i][j], dc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = fb + 256 * i;
      if (f >= F) continue;
      const float d = dc[i] * dmv[i];
      if (f < SF) {
        const int c = f / a.HW4, hw = f - c * a.HW4;
        dp2[hw * 64 + c] = d;
      } else {
        float g = d;
        if (df != nullptr) g += (dfj[i] - (yj[i] / norm) * dot) / norm;
        dy[f - SF] = relu_open(yj[i]) ? g : 0.f;
      }
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
  hipLaunchKernelGGL(head_bwd_kernel, dim3(n, nets), dim3(256), 0, st, a);
  return hipGetLastError();
}

// The feature-gradient part of dy, added behind a backward head that ran WITHOUT it (dfeat == null):
//   dy += relu'(y) * (dfeat - feat <feat, dfeat>) / ||y||          (the L2 normalisation's backward, models.py:145-146)
// dy is linear in dfeat, so the head can run as soon as dlogits exist -- the convolution backward needs nothing else
// (tools/models.py:144-150) -- while the contrastive feature gradients are still being formed (on a second stream) or
// reduce-scattered (data parallelism); this launch joins them in front of feat_spe's weight-gradient GEMM.  Element for
// element the arithmetic of conv3_bwd_head / head_bwd_kernel (thread t: elements t + 256 q, dot by fmaf over q, lanes by
// wave_sum, waves as (0 + 1) + (2 + 3)); the sum d + x is rounded once either way, so head + fix-up equals the head that
// was handed dfeat bit for bit.  Every row is visited (labelled rows carry dfeat = 0: a NaN row then spreads exactly as
// 0 * NaN does in the reference's autograd).
__global__ __launch_bounds__(256) void dy_fixup_kernel(const float* __restrict__ y, const float* __restrict__ ynorm,
                                                       const float* __restrict__ dfeat, float* __restrict__ dy, int n) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long rs = (long long)blockIdx.y * n + blockIdx.x;
  const float norm = ynorm[rs];
  float yv[4], dv[4], dyv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    yv[q] = y[rs * FD + tid + 256 * q];
    dv[q] = dfeat[rs * FD + tid + 256 * q];
    dyv[q] = dy[rs * FD + tid + 256 * q];
  }
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) dot = fmaf(yv[q] / norm, dv[q], dot);
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // (a closed ReLU holds dy = 0 whatever arrives; an open one -- NaN included -- takes the sum)
    const float g = dyv[q] + (dv[q] - (yv[q] / norm) * dot) / norm;
    dy[rs * FD + tid + 256 * q] = relu_open(yv[q]) ? g : 0.f;
  }
}

hipError_t launch_dy_fixup(int nets, int n, const float* y, const float* ynorm, const float* dfeat, float* dy,
                           hipStream_t st) {
  hipLaunchKernelGGL(dy_fixup_kernel, dim3(n, nets), dim3(256), 0, st, y, ynorm, dfeat, dy, n);
  return hipGetLastError();
}

}  // namespace cmlpl
