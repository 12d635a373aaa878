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
  DynRef dyn;                      // the two bias-correction scalars from device memory (graph replay), or null
};

__device__ __forceinline__ float adam_update(float& mm, float& vv, float p, float g, const AdamArgs& a) {
  mm = mm + a.w1 * (g - mm);                             // exp_avg.lerp_(grad, 1-beta1)
  vv = vv * a.b2 + (a.w2 * g) * g;                       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
  return p - a.step_size * (mm / denom);                 // param.addcdiv_(exp_avg, denom, -step_size)
}
// (this launch runs behind the cursor's advance: its row is the one before the cursor)
__device__ __forceinline__ void adam_dyn(AdamArgs& a) {
  const cmlpl_dyn* d = dyn_row(a.dyn, -1);
  if (d != nullptr) {
    a.step_size = __int_as_float(uni32(__float_as_int(d->adam_step_size)));
    a.bc2_sqrt = __int_as_float(uni32(__float_as_int(d->adam_bc2_sqrt)));
  }
}

// A 3x3 weight tensor in chunks of (four output channels) x (sixteen input channels) = 4 segments of 144 consecutive
// elements per workgroup, 128 chunks per tensor pair and network: Adam on <= 3 elements per thread, the updated values
// meet in LDS, and the split-bf16 fragment sets are written as whole 16-byte (forward: 8 consecutive ci of one (co, tap))
// and 8-byte (data gradient: these 4 co of one (ci, tap)) pieces instead of scattered 2-byte stores -- same contents as
// pack_weights_kernel.  (Round 3: chunks of 4 x 64 input channels, 32 workgroups per network, each with nine elements
// per thread and three passes over the packing items, were the long pole of the launch: 10 us.)
constexpr int ADAM_CHUNKS = 2 * 16 * 4;          // (conv1 | conv2) x output-channel quads x input-channel blocks
__device__ __forceinline__ void adam_conv_chunk(const AdamArgs& a, int chunk, int net, float* lds) {
  const int which = (chunk >= 64) ? 2 : 0, co0 = ((chunk >> 2) & 15) * 4, ci0 = (chunk & 3) * 16, tid = threadIdx.x;
  const long long off = (which == 0 ? a.pi.off_w1 : a.pi.off_w2) + (long long)co0 * 576 + ci0 * 9;
  float* p = a.params + (long long)net * a.pstride + off;
  const float* g = a.grads + (long long)net * a.gstride + off;
  float* mm = a.m + (long long)net * a.pstride + off;
  float* vv = a.v + (long long)net * a.pstride + off;
  // every load first: with load -> update -> store element by element the stores to m / v / p may alias the next
  // element's loads as far as the compiler can tell, and the elements become a chain of dependent memory round trips
  float gq[3], mq[3], vq[3], pq[3];
  int gi[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int e = tid + 256 * q, j = e / 144, r = e - j * 144;      // segment (output channel), element in it
    gi[q] = (e < 576) ? j * 576 + r : -1;
    const int i = gi[q] < 0 ? 0 : gi[q];
    gq[q] = g[i]; mq[q] = mm[i]; vq[q] = vv[i]; pq[q] = p[i];
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (gi[q] < 0) continue;
    const float pn = adam_update(mq[q], vq[q], pq[q], gq[q], a);
    mm[gi[q]] = mq[q]; vv[gi[q]] = vq[q]; p[gi[q]] = pn;
    lds[tid + 256 * q] = pn;                                          // [4 co][16 ci][9 taps]
  }
  if (a.packed == nullptr) return;
  __syncthreads();
  float* pkn = a.packed + (long long)net * a.pi.stride;
  if (tid < 72) {
    // forward set, as 8-bf16 pieces: item = (co, tap, half h) -> ci = ci0 + 8 h .. + 7
    uint4* bf = (uint4*)(pkn + pack_off_b3(a.pi.C, a.pi.bands, which));
    const int h = tid & 1, tap = (tid >> 1) % 9, cl = tid / 18, co = co0 + cl;
    uint32_t pc[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) b3_split(lds[cl * 144 + (h * 8 + j) * 9 + tap], pc[j]);
#pragma unroll
    for (int pcs = 0; pcs < 3; ++pcs)
      bf[conv_b3_index(tap, ci0 + h * 8, co, pcs) >> 3] =
          make_uint4(pc[0][pcs] | (pc[1][pcs] << 16), pc[2][pcs] | (pc[3][pcs] << 16),
                     pc[4][pcs] | (pc[5][pcs] << 16), pc[6][pcs] | (pc[7][pcs] << 16));
  } else if (tid < 72 + 144) {
    // data-gradient set (k = co, n = ci, tap flipped), as 4-bf16 pieces: item = (ci, tap) -> these four co
    uint2* bd = (uint2*)(pkn + pack_off_b3(a.pi.C, a.pi.bands, which + 1));
    const int it = tid - 72, tap = it % 9, cil = it / 9, ci = ci0 + cil;
    uint32_t pc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) b3_split(lds[j * 144 + cil * 9 + tap], pc[j]);
#pragma unroll
    for (int pcs = 0; pcs < 3; ++pcs)
      bd[conv_b3_index(8 - tap, co0, ci, pcs) >> 2] = make_uint2(pc[0][pcs] | (pc[1][pcs] << 16), pc[2][pcs] | (pc[3][pcs] << 16));
  }
  // ... and as TWO fp16 pieces (kernels.hpp: pack_off_h2; same items, same threads): forward set 16-byte pieces,
  // data-gradient set 8-byte pieces; a weight outside fp16's range at the packing scale raises the network's flag
  static_assert(H2_WEXP == 13, "h2_split_w scales by 2^13");
  bool bad = false;
  if (tid < 72) {
    uint4* hf = (uint4*)(pkn + pack_off_h2(a.pi.C, a.pi.bands, which));
    const int h = tid & 1, tap = (tid >> 1) % 9, cl = tid / 18, co = co0 + cl;
    uint32_t pc[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) bad |= h2_split_w(lds[cl * 144 + (h * 8 + j) * 9 + tap], pc[j]);
#pragma unroll
    for (int pcs = 0; pcs < 2; ++pcs)
      hf[conv_h2_index(tap, ci0 + h * 8, co, pcs) >> 3] =
          make_uint4(pc[0][pcs] | (pc[1][pcs] << 16), pc[2][pcs] | (pc[3][pcs] << 16),
                     pc[4][pcs] | (pc[5][pcs] << 16), pc[6][pcs] | (pc[7][pcs] << 16));
  } else if (tid < 72 + 144) {
    uint2* hd = (uint2*)(pkn + pack_off_h2(a.pi.C, a.pi.bands, which + 1));
    const int it = tid - 72, tap = it % 9, cil = it / 9, ci = ci0 + cil;
    uint32_t pc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) bad |= h2_split_w(lds[j * 144 + cil * 9 + tap], pc[j]);
#pragma unroll
    for (int pcs = 0; pcs < 2; ++pcs)
      hd[conv_h2_index(8 - tap, co0, ci, pcs) >> 2] = make_uint2(pc[0][pcs] | (pc[1][pcs] << 16), pc[2][pcs] | (pc[3][pcs] << 16));
  }
  if (bad) atomicOr((unsigned int*)(pkn + pack_off_h2flag(a.pi.C, a.pi.bands)), 1u);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[768];
  const int net = blockIdx.y;
  adam_dyn(a);
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

void adam_bias_scalars(float lr, float b1, float b2, long long t, float* step_size, float* bc2_sqrt) {
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
  *step_size = (float)((double)lr / bc1); *bc2_sqrt = (float)sqrt(bc2);
}

hipError_t launch_adam(int nets, float* params, long long pstride, const float* grads, long long gstride,
                       float* m, float* v, long long live, long long t, float lr, float b1, float b2, float eps,
                       float* packed, const PackInfo& pi, hipStream_t st, DynRef dyn) {
  float step_size, bc2_sqrt;
  adam_bias_scalars(lr, b1, b2, t, &step_size, &bc2_sqrt);
  const long long n4 = (live + 3) / 4;
  AdamArgs a;
  a.dyn = dyn;
  a.params = params; a.pstride = pstride; a.grads = grads; a.gstride = gstride; a.m = m; a.v = v; a.live = live;
  a.w1 = (float)(1.0 - (double)b1); a.b2 = b2; a.w2 = (float)(1.0 - (double)b2); a.step_size = step_size;
  a.bc2_sqrt = bc2_sqrt; a.eps = eps; a.packed = (CMLPL_ABL == 40) ? nullptr : packed; a.pi = pi;   // (40: timing of the update alone)
  a.nb_elem = (int)((n4 + 255) / 256);
  dim3 grid((unsigned)(a.nb_elem + ADAM_CHUNKS), nets);  // + the (4 co x 16 ci) chunks of conv1.weight and conv2.weight
  hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace cmlpl
