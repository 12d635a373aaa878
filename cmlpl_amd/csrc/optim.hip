// torch.optim.Adam.step (train.py:131-132,268,272) as one elementwise launch over the live
// parameters of both networks (flat buffers; dead tensors feat_ss* are never touched, matching
// the reference where their grad is None).
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

__global__ void adam_kernel(float* __restrict__ params, long long pstride, const float* __restrict__ grads,
                            long long gstride, float* __restrict__ m, float* __restrict__ v, long long live,
                            float w1, float b2, float w2, float step_size, float bc2_sqrt, float eps,
                            float* __restrict__ packed, PackInfo pi) {
  const int net = blockIdx.y;
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= live) return;
  float* p = params + (long long)net * pstride + i4;
  const float* g = grads + (long long)net * gstride + i4;
  float* mm = m + (long long)net * pstride + i4;
  float* vv = v + (long long)net * pstride + i4;
  const float4 gv = *(const float4*)g;
  float4 mv = *(const float4*)mm, sv = *(const float4*)vv, pv = *(const float4*)p;
  const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
  float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {sv.x, sv.y, sv.z, sv.w}, pa[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ma[q] = ma[q] + w1 * (ga[q] - ma[q]);                 // exp_avg.lerp_(grad, 1-beta1)
    va[q] = va[q] * b2 + (w2 * ga[q]) * ga[q];            // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(va[q]) / bc2_sqrt + eps;
    pa[q] = pa[q] - step_size * (ma[q] / denom);          // param.addcdiv_(exp_avg, denom, -step_size)
  }
  *(float4*)mm = make_float4(ma[0], ma[1], ma[2], ma[3]);
  *(float4*)vv = make_float4(va[0], va[1], va[2], va[3]);
  *(float4*)p = make_float4(pa[0], pa[1], pa[2], pa[3]);
  // the 3x3 kernels read re-packed weights (split-bf16 MFMA fragments, forward and transposed+flipped; kernels.hpp):
  // refresh them here instead of a separate launch (same mapping as pack_weights_kernel)
  if (packed != nullptr) {
    float* pkn = packed + (long long)net * pi.stride;
    const int which = (i4 >= pi.off_w1 && i4 < pi.off_w1 + PACK_CONV) ? 0 : (i4 >= pi.off_w2 && i4 < pi.off_w2 + PACK_CONV) ? 2 : -1;
    if (which >= 0) {
      const int e0 = (int)(i4 - (which == 0 ? pi.off_w1 : pi.off_w2));
      uint16_t* bf = (uint16_t*)(pkn + pack_off_b3(pi.C, pi.bands, which));        // forward fragment set
      uint16_t* bd = (uint16_t*)(pkn + pack_off_b3(pi.C, pi.bands, which + 1));    // transposed + flipped (dgrad)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q, kw = e % 3, kh = (e / 3) % 3, ci = (e / 9) & 63, co = e / 576;
        uint32_t pcs[3];
        b3_split(pa[q], pcs);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          bf[conv_b3_index(kh * 3 + kw, ci, co, pc)] = (uint16_t)pcs[pc];
          bd[conv_b3_index((2 - kh) * 3 + (2 - kw), co, ci, pc)] = (uint16_t)pcs[pc];
        }
        // conv2 also as 16x16x4 fp32 B fragments (forward tail of the fused kernel)
        if (which == 2) pkn[pack_off_frag() + conv2_frag_index(kh * 3 + kw, co, ci)] = pa[q];
      }
    } else if (i4 >= pi.off_w0 && i4 < pi.off_w0 + 64LL * pi.C) {      // conv0.weight[co][c] -> w0T[c][co]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long e = i4 + q - pi.off_w0;
        if (e < 64LL * pi.C) {
          const int co = (int)(e / pi.C), c = (int)(e - (long long)co * pi.C);
          pkn[pack_off_w0t() + c * 64 + co] = pa[q];
          uint32_t pcs[3];
          b3_split(pa[q], pcs);
          uint16_t* wb = (uint16_t*)(pkn + pack_off_w0b3(pi.C, pi.bands));
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) wb[conv_b3_index(0, c, co, pc)] = (uint16_t)pcs[pc];
        }
      }
    } else if (i4 >= pi.off_ws && i4 < pi.off_ws + 1024LL * pi.bands) { // feat_spe.weight[o][band] -> wsT[band][o]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long e = i4 + q - pi.off_ws;
        if (e < 1024LL * pi.bands) { const int o = (int)(e / pi.bands), band = (int)(e - (long long)o * pi.bands); pkn[pack_off_wst(pi.C) + (long long)band * 1024 + o] = pa[q]; }
      }
    }
  }
}

hipError_t launch_adam(int nets, float* params, long long pstride, const float* grads, long long gstride,
                       float* m, float* v, long long live, long long t, float lr, float b1, float b2, float eps,
                       float* packed, const PackInfo& pi, hipStream_t st) {
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const long long n4 = (live + 3) / 4;
  dim3 grid((unsigned)((n4 + 255) / 256), nets);
  hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, st, params, pstride, grads, gstride, m, v, live,
                     (float)(1.0 - (double)b1), b2, (float)(1.0 - (double)b2), step_size, bc2_sqrt, eps, packed, pi);
  return hipGetLastError();
}

}  // namespace cmlpl
