// torch.optim.Adam.step (train.py:131-132,268,272) as one elementwise launch over the live
// parameters of both networks (flat buffers; dead tensors feat_ss* are never touched, matching
// the reference where their grad is None).
#include "common.hpp"
#include "kernels.hpp"

#ifndef CMLPL_ABL
#define CMLPL_ABL 0
#endif

namespace cmlpl {

struct AdamArgs {
  float* params; long long pstride; const float* grads; long long gstride; float* m; float* v; long long live;
  float w1, b2, w2, step_size, bc2_sqrt, eps;
  float* packed; PackInfo pi;
  int nb_elem;                     // blocks of the elementwise part; the 3x3 weight chunks follow
};

__device__ __forceinline__ float adam_update(float& mm, float& vv, float p, float g, const AdamArgs& a) {
  mm = mm + a.w1 * (g - mm);                             // exp_avg.lerp_(grad, 1-beta1)
  vv = vv * a.b2 + (a.w2 * g) * g;                       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
  return p - a.step_size * (mm / denom);                 // param.addcdiv_(exp_avg, denom, -step_size)
}

// A 3x3 weight tensor, four output channels (4 x 576 consecutive elements) per workgroup: Adam on 9 elements per
// thread (coalesced), the updated values meet in LDS, and the split-bf16 fragment sets are written as whole 16-byte
// (forward: 8 consecutive ci of one (co, tap)) and 8-byte (data gradient: these 4 co of one (ci, tap)) pieces
// instead of 54 scattered 2-byte stores per thread -- same contents as pack_weights_kernel.
__device__ __forceinline__ void adam_conv_chunk(const AdamArgs& a, int chunk, int net, float* lds) {
  const int which = (chunk >= 16) ? 2 : 0, co0 = (chunk & 15) * 4, tid = threadIdx.x;
  const long long off = (which == 0 ? a.pi.off_w1 : a.pi.off_w2) + (long long)co0 * 576;
  float* p = a.params + (long long)net * a.pstride + off;
  const float* g = a.grads + (long long)net * a.gstride + off;
  float* mm = a.m + (long long)net * a.pstride + off;
  float* vv = a.v + (long long)net * a.pstride + off;
  // every load of the nine elements first: written load -> update -> store element by element, the stores to m / v / p
  // may alias the next element's loads as far as the compiler can tell, and the nine become a chain of dependent
  // memory round trips -- for the 64 chunk workgroups that chain WAS the kernel's duration
  float gq[9], mq[9], vq[9], pq[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int i = tid + 256 * q;
    gq[q] = g[i]; mq[q] = mm[i]; vq[q] = vv[i]; pq[q] = p[i];
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int i = tid + 256 * q;
    const float pn = adam_update(mq[q], vq[q], pq[q], gq[q], a);
    mm[i] = mq[q]; vv[i] = vq[q]; p[i] = pn;
    lds[i] = pn;
  }
  if (a.packed == nullptr) return;
  __syncthreads();
  float* pkn = a.packed + (long long)net * a.pi.stride;
  uint4* bf = (uint4*)(pkn + pack_off_b3(a.pi.C, a.pi.bands, which));          // forward set, as 8-bf16 pieces
  uint2* bd = (uint2*)(pkn + pack_off_b3(a.pi.C, a.pi.bands, which + 1));      // data-gradient set, as 4-bf16 pieces
  // forward: item = (co, tap, k-step kq, half h) -> ci = 16 kq + 8 h .. + 7
  for (int it = tid; it < 4 * 9 * 8; it += 256) {
    const int kqh = it & 7, tap = (it >> 3) % 9, cl = it / 72, co = co0 + cl;
    uint32_t pc[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) b3_split(lds[cl * 576 + (kqh * 8 + j) * 9 + tap], pc[j]);
#pragma unroll
    for (int pcs = 0; pcs < 3; ++pcs)
      bf[conv_b3_index(tap, kqh * 8, co, pcs) >> 3] =
          make_uint4(pc[0][pcs] | (pc[1][pcs] << 16), pc[2][pcs] | (pc[3][pcs] << 16),
                     pc[4][pcs] | (pc[5][pcs] << 16), pc[6][pcs] | (pc[7][pcs] << 16));
  }
  // data gradient (k = co, n = ci, tap flipped): item = (ci, tap) -> these four co
  for (int it = tid; it < 64 * 9; it += 256) {
    const int tap = it % 9, ci = it / 9;
    float w4[4];
    uint32_t pc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) { w4[j] = lds[j * 576 + ci * 9 + tap]; b3_split(w4[j], pc[j]); }
#pragma unroll
    for (int pcs = 0; pcs < 3; ++pcs)
      bd[conv_b3_index(8 - tap, co0, ci, pcs) >> 2] = make_uint2(pc[0][pcs] | (pc[1][pcs] << 16), pc[2][pcs] | (pc[3][pcs] << 16));
    // conv2 also as 16x16x4 fp32 B fragments (forward tail of the fused kernel): n = co, k = ci
    if (which == 2) *(float4*)(pkn + pack_off_frag() + conv2_frag_index(tap, co0, ci)) = make_float4(w4[0], w4[1], w4[2], w4[3]);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 576];
  const int net = blockIdx.y;
  if ((int)blockIdx.x >= a.nb_elem) { adam_conv_chunk(a, (int)blockIdx.x - a.nb_elem, net, lds); return; }
  const PackInfo& pi = a.pi;
  float* packed = a.packed;
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= a.live) return;
  // the 3x3 weights are handled by the chunk workgroups (they re-pack through LDS)
  if ((i4 >= pi.off_w1 && i4 < pi.off_w1 + PACK_CONV) || (i4 >= pi.off_w2 && i4 < pi.off_w2 + PACK_CONV)) return;
  float* p = a.params + (long long)net * a.pstride + i4;
  const float* g = a.grads + (long long)net * a.gstride + i4;
  float* mm = a.m + (long long)net * a.pstride + i4;
  float* vv = a.v + (long long)net * a.pstride + i4;
  const float4 gv = *(const float4*)g;
  float4 mv = *(const float4*)mm, sv = *(const float4*)vv, pv = *(const float4*)p;
  const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
  float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {sv.x, sv.y, sv.z, sv.w}, pa[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) pa[q] = adam_update(ma[q], va[q], pa[q], ga[q], a);
  *(float4*)mm = make_float4(ma[0], ma[1], ma[2], ma[3]);
  *(float4*)vv = make_float4(va[0], va[1], va[2], va[3]);
  *(float4*)p = make_float4(pa[0], pa[1], pa[2], pa[3]);
  // the thin weights' k-major / fragment copies (same mapping as pack_weights_kernel)
  if (packed != nullptr) {
    float* pkn = packed + (long long)net * pi.stride;
    if (i4 >= pi.off_w0 && i4 < pi.off_w0 + 64LL * pi.C) {      // conv0.weight[co][c] -> w0T[c][co]
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
    }
    // (feat_spe.weight has no packed copy any more: the spectral kernels read the canonical tensor, dense.hip)
  }
}

hipError_t launch_adam(int nets, float* params, long long pstride, const float* grads, long long gstride,
                       float* m, float* v, long long live, long long t, float lr, float b1, float b2, float eps,
                       float* packed, const PackInfo& pi, hipStream_t st) {
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const long long n4 = (live + 3) / 4;
  AdamArgs a;
  a.params = params; a.pstride = pstride; a.grads = grads; a.gstride = gstride; a.m = m; a.v = v; a.live = live;
  a.w1 = (float)(1.0 - (double)b1); a.b2 = b2; a.w2 = (float)(1.0 - (double)b2); a.step_size = step_size;
  a.bc2_sqrt = bc2_sqrt; a.eps = eps; a.packed = (CMLPL_ABL == 40) ? nullptr : packed; a.pi = pi;   // (40: timing of the update alone)
  a.nb_elem = (int)((n4 + 255) / 256);
  dim3 grid((unsigned)(a.nb_elem + 32), nets);          // + 16 four-channel chunks of conv1.weight, 16 of conv2.weight
  hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace cmlpl
