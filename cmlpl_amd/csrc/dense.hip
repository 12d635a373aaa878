// Dense contractions of the spectral branch / classifier / loss gradients on the fp32 MFMA.
//
//   spe_fwd_kernel : y = relu(sn . Wspe^T + b)   (feat_spe + ReLU, tools/models.py:142-143)
//   gemm_tn_kernel : C[i][j] = scale * sum_r A[r][i] * B[r][j]  (+ column sums of A)
//                    both operands are read straight from HBM/L2 as MFMA fragments: with the
//                    reduction index r as the slow dimension, 32 consecutive i (or j) are one
//                    coalesced 128-B segment.  Used for dW_spe, dW_cls and the two contrastive
//                    feature gradients (G^T.f_w, G.f_s).
#include <stdlib.h>

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

// Spectral branch of the forward with the augmentation fused in (train.py:158,164,171,182 + tools/models.py:142-143):
//   sn = x + sigma * N(0,1)   (raw labelled / unlabelled spectra; counter hash or the explicit draws of parity mode)
//   y  = relu(sn . Wspe^T + b)
// Same shape as gemm_tn_block -- one workgroup per 32 x 32 output tile, the band contraction split over the four
// waves, every operand of a wave in flight at once (one memory round trip) -- but the A fragment is FORMED in
// registers: lane (row i, band k) loads the raw element and adds its noise.  A hash call yields the four normals of
// bands 4g..4g+3 and a lane's bands are k = hh, 2 + hh, 4 + hh, ...: one call per two k-steps.  The column-tile-0
// workgroups also write sn (row-major) for the weight gradient.  Replaces the augmentation launch (+ its transposed
// copy) and the GEMM launch.  (A first version staged the augmented rows in LDS and walked the contraction
// serially per wave: slower than the two launches it replaced.)
// NW = waves per workgroup (template parameter); a wave takes at most 16 k-pairs: four waves cover bands <= 128,
// eight waves 256 (B4: 200 -- with four waves of 32 k-pairs each the launch took 19.4 us against 9.9 for 103 bands)
constexpr int SPE_MAXP = 16;   // MFMA steps (bands per lane half) of a wave
constexpr int SPE_GPH = 4;     // groups of four bands per lane half

struct SpeArgs {
  XSrc xs; const float* w; const float* bias; long long p_ns;      // feat_spe.weight [1024][bands] / .bias, canonical
  float* y; float* sn; int n, bands;
  const long long* labels; float* labels_f; int bt;        // optional: labels as float for the exchange buffer (labels by xs.lab_idx)
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void spe_fused_kernel(SpeArgs a) {
  __shared__ float red[NW - 1][16][64];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x, mt = blockIdx.y, net = blockIdx.z;
  if (a.labels_f != nullptr && nt == 0 && mt == 0 && net == 0)
    for (int q = tid; q < a.bt; q += 64 * NW) a.labels_f[q] = (float)a.labels[xsrc_index(a.xs, q)];
  // Band order of the contraction.  An MFMA step multiplies ONE band per lane half; which band is ours to choose, as
  // long as both operands agree.  The bands are cut into groups of four (= one noise call, = one 16-byte load of the
  // row); a wave takes 2 * SPE_GPH consecutive groups, its lower lane half the first SPE_GPH of them, the upper half
  // the rest: a lane then walks CONSECUTIVE bands -- 16-byte loads of its row instead of one 4-byte load per band from
  // 32 different rows per instruction, and every normal of a noise call is used by the lane that formed it.
  const int bands = a.bands, G = (bands + 3) >> 2;
  const int gpw = 2 * ((G + 2 * NW - 1) / (2 * NW));            // groups per wave (even), <= 2 * SPE_GPH
  const int g0 = wave * gpw + hh * (gpw >> 1);                   // this lane half's first group
  const int i = mt * 32 + l31, j = nt * 32 + l31;
  const bool iv = i < a.n;
  const int ic = iv ? i : 0;
  const float* x = xsrc_row(a.xs, net, ic, bands);
  const float* nz = (a.xs.sigma != 0.f) ? xsrc_noise_row(a.xs, net, ic, bands) : nullptr;
  // this lane's output row of feat_spe.weight: walked along the bands like the input row, 16 bytes at a time (round 2
  // kept a k-major copy wsT for 128-byte reads across the lanes; with consecutive bands per lane it is not needed, and
  // the optimizer no longer scatters 2 x 105k four-byte stores into it every step)
  const float* wrow = a.w + (long long)net * a.p_ns + (long long)j * bands;
  const float bias = a.bias[(long long)net * a.p_ns + j];      // requested with the rows, used after the fold
  float av[SPE_MAXP], bv[SPE_MAXP], zv[SPE_MAXP];
#pragma unroll
  for (int gq = 0; gq < SPE_GPH; ++gq) {
    const int g = g0 + gq, k0 = 4 * g;
    const bool gv = gq < (gpw >> 1) && g < G;                    // (uniform per lane half)
    // the row's four bands of this group: one 16-byte load where the whole group exists (rows are only 4-byte aligned:
    // dword-aligned multi-dword loads are fine), band by band for the ragged last group
    float xa[4] = {0.f, 0.f, 0.f, 0.f}, za[4] = {0.f, 0.f, 0.f, 0.f}, wa[4] = {0.f, 0.f, 0.f, 0.f};
    if (gv && k0 + 3 < bands) {
      const float4 v = *(const float4*)(x + k0), w4 = *(const float4*)(wrow + k0);
      xa[0] = v.x; xa[1] = v.y; xa[2] = v.z; xa[3] = v.w;
      wa[0] = w4.x; wa[1] = w4.y; wa[2] = w4.z; wa[3] = w4.w;
      if (nz != nullptr) { const float4 z = *(const float4*)(nz + k0); za[0] = z.x; za[1] = z.y; za[2] = z.z; za[3] = z.w; }
    } else if (gv) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e < bands) { xa[e] = x[k0 + e]; wa[e] = wrow[k0 + e]; if (nz != nullptr) za[e] = nz[k0 + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool kv = gv && k0 + e < bands;
      av[4 * gq + e] = (kv && iv) ? xa[e] : 0.f;
      bv[4 * gq + e] = kv ? wa[e] : 0.f;
      zv[4 * gq + e] = za[e];                                     // parity mode: the reference's own draws
    }
  }
  if (a.xs.sigma != 0.f && nz == nullptr) {                     // in-kernel noise (uniform branch, pure ALU)
    const uint64_t gs = xsrc_global_sample(a.xs, ic);
    const uint64_t rstep = xsrc_step(a.xs);
#pragma unroll
    for (int gq = 0; gq < SPE_GPH; ++gq) {
      const float4 z = noise_normal4(a.xs.seed, rstep, STREAM_NOISE_X + net, noise_ctr(gs, (uint32_t)(g0 + gq)));
      zv[4 * gq] = z.x; zv[4 * gq + 1] = z.y; zv[4 * gq + 2] = z.z; zv[4 * gq + 3] = z.w;
    }
  }
  f32x16 acc = zero16();
  float vq[SPE_MAXP];
#pragma unroll
  for (int q = 0; q < SPE_MAXP; ++q) {
    const int k = 4 * g0 + q;
    const bool kv = (q >> 2) < (gpw >> 1) && k < bands;
    vq[q] = (kv && iv) ? fmaf(zv[q], a.xs.sigma, av[q]) : 0.f;
    if (q < 2 * gpw) acc = mfma32(vq[q], bv[q], acc);             // uniform
  }
  if (nt == 0 && iv) {                                            // the augmented row, for the weight gradient
    float* snr = a.sn + ((long long)net * a.n + i) * bands;
#pragma unroll
    for (int gq = 0; gq < SPE_GPH; ++gq) {
      const int g = g0 + gq, k0 = 4 * g;
      if (gq < (gpw >> 1) && g < G) {
        if (k0 + 3 < bands) *(float4*)(snr + k0) = make_float4(vq[4 * gq], vq[4 * gq + 1], vq[4 * gq + 2], vq[4 * gq + 3]);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k0 + e < bands) snr[k0 + e] = vq[4 * gq + e];
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    float* Y = a.y + (long long)net * a.n * FD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + acc_row(r, lane);
      float v = acc[r];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];     // fixed order: waves 1, 2, ...
      v += bias;
      if (row < a.n) Y[(long long)row * FD + j] = relu_nan(v);
    }
  }
}

bool spe_fused_ok(int bands) {
  const bool off = switches().fuse_spe == 0;
  return !off && bands <= 256;
}

hipError_t launch_spe_fused(int nets, int n, int bands, const XSrc& xs, const float* w,
                            const float* bias, long long p_ns, float* y, float* sn, const long long* labels,
                            float* labels_f, int bt, hipStream_t st) {
  if (!spe_fused_ok(bands)) return hipErrorInvalidValue;
  SpeArgs a;
  a.labels = labels; a.labels_f = labels != nullptr ? labels_f : nullptr; a.bt = bt;
  a.xs = xs; a.w = w; a.bias = bias; a.p_ns = p_ns; a.y = y; a.sn = sn; a.n = n; a.bands = bands;
  if (bands <= 128) hipLaunchKernelGGL(spe_fused_kernel<4>, dim3(FD / 32, (n + 31) / 32, nets), dim3(256), 0, st, a);
  else              hipLaunchKernelGGL(spe_fused_kernel<8>, dim3(FD / 32, (n + 31) / 32, nets), dim3(512), 0, st, a);
  return hipGetLastError();
}

// L2 normalisation of the spectral feature in a launch of its own (Normalize, tools/models.py:87-90,145-146: no epsilon,
// an all-zero row gives 0 / 0 = NaN as in the reference): feat = y / ||y||, ynorm = ||y||, one workgroup per (network,
// row).  The embeddings depend on the spectral branch alone, so a step that wants them EARLY -- the similarity products
// beside the convolutions, the all-gather of a sharded step under them -- forms them here, right behind spe_fused_kernel;
// the per-sample forward then skips its own normalisation (FwdTail::feat == null).  Same arithmetic, element for
// element, as conv3_fwd_tail: thread t squares elements 4t .. 4t+3 as (x^2 + y^2) + (z^2 + w^2), lanes folded by
// wave_sum, waves as (0 + 1) + (2 + 3) -- the two producers of `feat` are bit-identical.
__global__ __launch_bounds__(256) void feat_norm_kernel(const float* __restrict__ y, float* __restrict__ ynorm,
                                                        float* __restrict__ feat, long long feat_ns, int n) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = blockIdx.y, row = blockIdx.x;
  const long long rs = (long long)net * n + row;
  const float4 y4 = *(const float4*)(y + rs * FD + 4 * tid);
  float ss = (y4.x * y4.x + y4.y * y4.y) + (y4.z * y4.z + y4.w * y4.w);
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  if (tid == 0) ynorm[rs] = norm;
  float4 o = y4;
  o.x /= norm; o.y /= norm; o.z /= norm; o.w /= norm;
  *(float4*)(feat + (long long)net * feat_ns + (long long)row * FD + 4 * tid) = o;
}

hipError_t launch_feat_norm(int nets, int n, const float* y, float* ynorm, float* feat, long long feat_ns, hipStream_t st) {
  hipLaunchKernelGGL(feat_norm_kernel, dim3(n, nets), dim3(256), 0, st, y, ynorm, feat, feat_ns, n);
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
