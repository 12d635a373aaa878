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
// SPE_MAXP = k-pairs per wave (template parameter): 16 covers bands <= 4 * 2 * 16 = 128, 32 covers 256 (B4: 200)

struct SpeArgs {
  XSrc xs; const float* wsT; long long wsT_ns; const float* bias; long long p_ns;
  float* y; float* sn; int n, bands;
  const long long* labels; float* labels_f; int bt;        // optional: labels as float for the exchange buffer
};

template <int SPE_MAXP>
__global__ __launch_bounds__(256) void spe_fused_kernel(SpeArgs a) {
  __shared__ float red[3][16][64];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x, mt = blockIdx.y, net = blockIdx.z;
  if (a.labels_f != nullptr && nt == 0 && mt == 0 && net == 0)
    for (int q = tid; q < a.bt; q += 256) a.labels_f[q] = (float)a.labels[q];
  // the waves split the bands by whole groups of four (= two k-steps), so a noise call is never shared across waves
  const int bands = a.bands, pairs = (bands + 1) >> 1, ppw = 2 * ((((bands + 3) >> 2) + 3) >> 2);
  const int t0 = wave * ppw, t1 = (t0 + ppw < pairs) ? t0 + ppw : pairs;
  const int i = mt * 32 + l31, j = nt * 32 + l31;
  const bool iv = i < a.n;
  const int ic = iv ? i : 0;
  const float* x = xsrc_row(a.xs, net, ic, bands);
  const float* nz = (a.xs.sigma != 0.f) ? xsrc_noise_row(a.xs, net, ic, bands) : nullptr;
  const float* bp = a.wsT + (long long)net * a.wsT_ns + j;
  float av[SPE_MAXP], bv[SPE_MAXP], zv[SPE_MAXP];
#pragma unroll
  for (int q = 0; q < SPE_MAXP; ++q) {
    const int k = 2 * (t0 + q) + hh;
    const bool kv = (t0 + q < t1) && (k < bands);
    const int kc = kv ? k : 0;
    const float xa = x[kc], wb = bp[(long long)kc * FD];
    av[q] = (kv && iv) ? xa : 0.f;
    bv[q] = kv ? wb : 0.f;
    zv[q] = (nz != nullptr && kv) ? nz[kc] : 0.f;          // parity mode: the reference's own draws
  }
  if (a.xs.sigma != 0.f && nz == nullptr) {                 // in-kernel noise (uniform branch, pure ALU)
    const uint64_t gs = xsrc_global_sample(a.xs, ic);
#pragma unroll
    for (int q = 0; q < SPE_MAXP; q += 2) {
      // k-steps t0+q and t0+q+1 (t0 + q even): bands k = 4g + hh and k + 2 of the same group g
      const int k = 2 * (t0 + q) + hh;
      const float4 z = noise_normal4(a.xs.seed, a.xs.step, STREAM_NOISE_X + net, noise_ctr(gs, (uint32_t)(k >> 2)));
      zv[q] = hh ? z.y : z.x;
      zv[q + 1] = hh ? z.w : z.z;
    }
  }
  f32x16 acc = zero16();
#pragma unroll
  for (int q = 0; q < SPE_MAXP; ++q) {
    const int k = 2 * (t0 + q) + hh;
    const bool kv = (t0 + q < t1) && (k < bands);
    const float v = (kv && iv) ? fmaf(zv[q], a.xs.sigma, av[q]) : 0.f;
    if (nt == 0 && kv && iv) a.sn[((long long)net * a.n + i) * bands + k] = v;
    if (t0 + q < t1) acc = mfma32(v, bv[q], acc);            // uniform
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    const float bias = a.bias[(long long)net * a.p_ns + j];
    float* Y = a.y + (long long)net * a.n * FD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + acc_row(r, lane);
      const float v = (((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane]) + bias;
      if (row < a.n) Y[(long long)row * FD + j] = relu_nan(v);
    }
  }
}

bool spe_fused_ok(int bands) {
  static const bool off = getenv("CMLPL_FUSE_SPE") && atoi(getenv("CMLPL_FUSE_SPE")) == 0;
  return !off && bands <= 256;
}

hipError_t launch_spe_fused(int nets, int n, int bands, const XSrc& xs, const float* wsT, long long wsT_ns,
                            const float* bias, long long p_ns, float* y, float* sn, const long long* labels,
                            float* labels_f, int bt, hipStream_t st) {
  if (!spe_fused_ok(bands)) return hipErrorInvalidValue;
  SpeArgs a;
  a.labels = labels; a.labels_f = labels != nullptr ? labels_f : nullptr; a.bt = bt;
  a.xs = xs; a.wsT = wsT; a.wsT_ns = wsT_ns; a.bias = bias; a.p_ns = p_ns; a.y = y; a.sn = sn; a.n = n; a.bands = bands;
  if (bands <= 128) hipLaunchKernelGGL(spe_fused_kernel<16>, dim3(FD / 32, (n + 31) / 32, nets), dim3(256), 0, st, a);
  else              hipLaunchKernelGGL(spe_fused_kernel<32>, dim3(FD / 32, (n + 31) / 32, nets), dim3(256), 0, st, a);
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
