// Input augmentation + batch concat (train.py:157-158,163-164,170-174,181-184):
//   xn[net] = cat(XPl, XPu) + sigma*N(0,1)      sn[net] = cat(Xl, Xu) + sigma*N(0,1)
// The reference draws the noise on the CPU generator and copies it over PCIe every step; here it is
// Philox4x32-10 + Box-Muller in registers (or explicit noise tensors in parity mode).
// HBM-bound elementwise kernel: reads each source once, writes one noisy copy per network.
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

struct AugArgs {
  const float* srcl[2]; const float* srcu[2];     // [0] = XP, [1] = X
  const float* noise[8];                          // reference draw order, or all null
  float* dst[2];                                  // xn, sn
  long long nl[2], nu[2];                         // labelled / unlabelled element counts
  float sigma; int nets; int explicit_noise; uint64_t seed, step;
};

__global__ void augment_kernel(AugArgs a) {
  const int seg = blockIdx.y;                 // 0: XP net0, 1: XP net1, 2: X net0, 3: X net1
  const int t = seg >> 1, net = seg & 1;
  if (net >= a.nets) return;
  const long long N = a.nl[t] + a.nu[t];
  const long long e0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e0 >= N) return;
  float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool need_noise = a.sigma != 0.f;
  if (need_noise && !a.explicit_noise)
    nz = philox_normal4(a.seed, a.step, (t == 0 ? STREAM_NOISE_XP : STREAM_NOISE_X) + net, (uint64_t)(e0 >> 2));
  const float* nl_ptr = a.noise[2 * net + t];         // XPl/net: 0,2 ; Xl/net: 1,3
  const float* nu_ptr = a.noise[4 + 2 * net + t];     // XPu/net: 4,6 ; Xu/net: 5,7
  float* dst = a.dst[t] + (long long)net * N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long long e = e0 + q;
    if (e < N) {
      const bool lab = e < a.nl[t];
      const float x = lab ? a.srcl[t][e] : a.srcu[t][e - a.nl[t]];
      float z = (q == 0) ? nz.x : (q == 1) ? nz.y : (q == 2) ? nz.z : nz.w;
      if (need_noise && a.explicit_noise) z = lab ? nl_ptr[e] : nu_ptr[e - a.nl[t]];
      dst[e] = need_noise ? x + z * a.sigma : x;
    }
  }
}

hipError_t launch_augment(int nets, long long nl_xp, long long nu_xp, long long nl_x, long long nu_x,
                          const float* xpl, const float* xl, const float* xpu, const float* xu,
                          const float* const* noise8, float sigma, uint64_t seed, uint64_t step,
                          float* xn, float* sn, hipStream_t st) {
  AugArgs a;
  a.srcl[0] = xpl; a.srcl[1] = xl; a.srcu[0] = xpu; a.srcu[1] = xu;
  for (int i = 0; i < 8; ++i) a.noise[i] = noise8 ? noise8[i] : nullptr;
  a.dst[0] = xn; a.dst[1] = sn;
  a.nl[0] = nl_xp; a.nu[0] = nu_xp; a.nl[1] = nl_x; a.nu[1] = nu_x;
  a.sigma = sigma; a.nets = nets; a.explicit_noise = noise8 != nullptr; a.seed = seed; a.step = step;
  const long long N = nl_xp + nu_xp;
  dim3 grid((unsigned)(((N + 3) / 4 + 255) / 256), 4);
  hipLaunchKernelGGL(augment_kernel, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace cmlpl
