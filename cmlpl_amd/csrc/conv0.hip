// conv0: 1x1 convolution C -> 64 over the band-major patch tensor (tools/models.py:102,132).
//
//   conv0_fwd_kernel  : a0[pix][co] = sum_c xn[c][pix] * W[co][c] + b[co]
//                       A operand straight from HBM: for a fixed band c, 32 consecutive pixels of the
//                       NCHW tensor are one coalesced 128-B segment = the MFMA A fragment, so the input
//                       is read exactly once and never staged.  Each lane keeps 16 band loads in flight
//                       (double-buffered chunks).  Output is written pixel-major (channel-last), the
//                       layout the 3x3 kernels consume.
//   conv0_wgrad_kernel: dW[c][co] = sum_pix xn[c][pix] * da0[pix][co]  (one sample slab per workgroup
//                       pass, partials reduced deterministically by partial_reduce_kernel)
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_tn.hpp"

namespace cmlpl {

constexpr int C0_KMAX = 128;    // k-steps of 2 bands: up to 256 input channels

// A[pixel][band] fragments come straight from HBM (32 consecutive pixels of one band of the NCHW input = one
// 128-B segment) and EVERY band load of a lane is issued before anything is consumed: at ~2 waves per SIMD
// the kernel is bound by the number of sequential memory round trips, and this makes it one.  The k-major
// weight copy w0T (L2-resident, maintained by Adam) is copied straight into LDS meanwhile.
template <int KCAP>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ xn, const float* __restrict__ w0t,
                                                        long long w0t_ns, const float* __restrict__ b,
                                                        long long pstride, float* __restrict__ a0, int n, int C,
                                                        int HW) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cp][64]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y;
  const int Cp = (C + 1) & ~1, KK = Cp >> 1;
  const long long M = (long long)n * HW;
  const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + l31;
  const long long mm = (m < M) ? m : 0;
  const int sample = (int)(mm / HW), pix = (int)(mm - (long long)sample * HW);
  const float* ap = xn + ((long long)net * n + sample) * C * HW + pix;
  float av[KCAP];
#pragma unroll
  for (int q = 0; q < KCAP; ++q) {
    const int c = 2 * q + hh;
    const bool ok = (q < KK) && (c < C);
    const float v = ap[(long long)(ok ? c : 0) * HW];
    av[q] = ok ? v : 0.f;
  }
  const float4* wsrc = (const float4*)(w0t + (long long)net * w0t_ns);
  staged_copy<8, float4>(Cp * 16, tid, [&](int i) { return wsrc[i]; }, [&](int i, float4 v) { ((float4*)wl)[i] = v; });
  __syncthreads();
  f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
  for (int q = 0; q < KCAP; ++q) {
    if (q < KK) {   // uniform
      const int c = 2 * q + hh;
      acc0 = mfma32(av[q], wl[c * 64 + l31], acc0);
      acc1 = mfma32(av[q], wl[c * 64 + 32 + l31], acc1);
    }
  }
  const float* bias = b + (long long)net * pstride;
  const float bv0 = bias[l31], bv1 = bias[32 + l31];
  const long long mbase = ((long long)blockIdx.x * 4 + wave) * 32;
  float* out = a0 + (long long)net * M * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long mr = mbase + acc_row(r, lane);
    if (mr < M) {
      out[mr * 64 + l31] = acc0[r] + bv0;
      out[mr * 64 + 32 + l31] = acc1[r] + bv1;
    }
  }
}

hipError_t launch_conv0_fwd(int nets, int n, int C, int HW, const float* xn, const float* w0t, long long w0t_ns,
                            const float* b, long long pstride, float* a0, hipStream_t st) {
  const long long M = (long long)n * HW;
  const int KK = ((C + 1) & ~1) / 2;
  if (KK > C0_KMAX) return hipErrorInvalidValue;
  const size_t lds = (size_t)2 * KK * 64 * 4;
  dim3 grid((unsigned)((M + 127) / 128), nets);
  if (KK <= 32)      hipLaunchKernelGGL((conv0_fwd_kernel<32>), grid, dim3(256), lds, st, xn, w0t, w0t_ns, b, pstride, a0, n, C, HW);
  else if (KK <= 64) hipLaunchKernelGGL((conv0_fwd_kernel<64>), grid, dim3(256), lds, st, xn, w0t, w0t_ns, b, pstride, a0, n, C, HW);
  else               hipLaunchKernelGGL((conv0_fwd_kernel<128>), grid, dim3(256), lds, st, xn, w0t, w0t_ns, b, pstride, a0, n, C, HW);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// conv0a_fwd_kernel: the patch AUGMENTATION and conv0 in one launch, on the split-bf16 MFMA -- the general path's forward
// for windows the per-sample fused kernels do not reach (the reference's own 20 x 20 x 60, tools/models.py:102,127,132;
// train.py:157,163,170,181).  Round 5's general path ran augment_kernel (raw rows -> xn, 74 MB at 128 + 128 rows of P)
// and conv0_fwd_kernel (xn -> a0 on the f32-input MFMA: 60 steps of 64 cycles per wave, 23 us of matrix pipe in a 38-us
// launch) back to back: 53 us.  Here a workgroup takes 128 consecutive pixels of ONE sample (window sizes that are
// multiples of 8 pixels, so that a pair of 16-byte groups -- the unit of the noise generator, elements 8c .. 8c + 7 of
// the sample's [C][HW] block -- is eight pixels of one band):
//   phase 1  every thread walks (band, 8-pixel pair) items of the tile: two 16-byte loads of the raw row, one
//            eight-normal hash call (or the reference's own draws), the augmented values to LDS [band][132] and -- for
//            conv0's weight gradient in the backward pass -- to xn in HBM, 16 bytes at a time;
//   phase 2  wave w = pixel tile 32 w .. 32 w + 31, both output-channel tiles: per k-step of 16 bands eight ds_read_b32
//            down the bands (consecutive lanes = consecutive pixels), the split into three bf16 pieces, twelve MFMAs
//            against conv0's packed split fragments (kernels.hpp: pack_off_w0b3, the set the fused forward uses), read
//            from L2 a step ahead;
//   a0 + bias goes out pixel-major / channel-last.
// Same element values as augment_kernel forms (same counters), same products as conv3_stage's conv0.
// ------------------------------------------------------------------------------------------
constexpr int C0A_SP = 132;     // LDS row stride in floats (128 pixels + 4: 16-byte aligned rows)
struct Conv0aArgs {
  XSrc xs; const float* w0b3; long long w0b3_ns; const float* b; long long pstride;
  float* a0; float* xn;         // xn may be null (no backward follows: inference)
  int n, C, HW;
};

template <int KQMAX>
__global__ __launch_bounds__(256, KQMAX <= 4 ? 4 : 2) void conv0a_fwd_kernel(Conv0aArgs a) {
  extern __shared__ __attribute__((aligned(16))) float slab[];   // [16 KQ0][C0A_SP]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.y, net = blockIdx.z, p0 = blockIdx.x * 128;
  const int C = a.C, HW = a.HW, KQ0 = (C + 15) >> 4;
  const int npx = (HW - p0 < 128) ? HW - p0 : 128, np8 = npx >> 3;      // pixels of this tile (a multiple of 8)
  const long long per = (long long)C * HW;
  const float* xrow = xsrc_row(a.xs, net, s, per);
  const float sigma = a.xs.sigma;
  const float* nzrow = (sigma != 0.f) ? xsrc_noise_row(a.xs, net, s, per) : nullptr;
  const uint64_t gs = xsrc_global_sample(a.xs, s), rstep = xsrc_step(a.xs);
  float* xnrow = (a.xn != nullptr) ? a.xn + ((long long)net * a.n + s) * per : nullptr;
  const uint4* wq = (const uint4*)(a.w0b3 + (long long)net * a.w0b3_ns) + lane;
  // ---- phase 1: items it = tid, tid + 256, ...: (band c = it / np8, pair j = it % np8) -> pixels p0 + 8 j .. + 7
  constexpr int NI = (KQMAX * 16 * 16 + 255) / 256;                   // items per thread at most (16 pairs per band)
  const int items = C * np8;
  const uint32_t mg = (np8 <= 1) ? 0u : (uint32_t)((0x100000000ULL + (uint32_t)np8 - 1) / (uint32_t)np8);
  float4 v0[NI], v1[NI];
  int cc[NI], jj[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int it = tid + 256 * q;
    const int itc = it < items ? it : 0;
    cc[q] = mg == 0u ? itc : (int)__umulhi((uint32_t)itc, mg);
    jj[q] = itc - cc[q] * np8;
    const float* src = xrow + (long long)cc[q] * HW + p0 + 8 * jj[q];
    v0[q] = *(const float4*)src; v1[q] = *(const float4*)(src + 4);
  }
  if (sigma != 0.f) {
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const long long e = (long long)cc[q] * HW + p0 + 8 * jj[q];     // first element of the pair (a multiple of 8)
      float4 z0, z1;
      if (nzrow != nullptr) { z0 = *(const float4*)(nzrow + e); z1 = *(const float4*)(nzrow + e + 4); }
      else noise_normal8(a.xs.seed, rstep, STREAM_NOISE_XP + net, gs, (uint32_t)(e >> 3), z0, z1);
      v0[q].x = fmaf(z0.x, sigma, v0[q].x); v0[q].y = fmaf(z0.y, sigma, v0[q].y);
      v0[q].z = fmaf(z0.z, sigma, v0[q].z); v0[q].w = fmaf(z0.w, sigma, v0[q].w);
      v1[q].x = fmaf(z1.x, sigma, v1[q].x); v1[q].y = fmaf(z1.y, sigma, v1[q].y);
      v1[q].z = fmaf(z1.z, sigma, v1[q].z); v1[q].w = fmaf(z1.w, sigma, v1[q].w);
    }
  }
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    if (tid + 256 * q < items) {
      float* d = slab + cc[q] * C0A_SP + 8 * jj[q];
      *(float4*)d = v0[q]; *(float4*)(d + 4) = v1[q];
      if (xnrow != nullptr) {
        float* g = xnrow + (long long)cc[q] * HW + p0 + 8 * jj[q];
        *(float4*)g = v0[q]; *(float4*)(g + 4) = v1[q];
      }
    }
  }
  // conv0's split fragments of the first two k-steps: requested here (the tile's values have left the registers), they
  // land under the zero fill and the barrier -- requested at kernel start they cost 48 registers through phase 1 and the
  // kernel one workgroup per SIMD pair less
  uint4 bw[2][6];
#pragma unroll
  for (int kq = 0; kq < 2; ++kq)
#pragma unroll
    for (int i = 0; i < 6; ++i) bw[kq][i] = wq[(((kq < KQ0 ? kq : 0) * 3 + (i >> 1)) * 2 + (i & 1)) * 64];
  // bands C .. 16 KQ0 - 1 meet zero weights but must be finite; so must the pixels of a partial tile
  for (int i = C * C0A_SP + tid; i < KQ0 * 16 * C0A_SP; i += 256) slab[i] = 0.f;
  if (npx < 128)
    for (int i = tid; i < C * (128 - npx); i += 256) { const int c = i / (128 - npx); slab[c * C0A_SP + npx + (i - c * (128 - npx))] = 0.f; }
  __syncthreads();
  // ---- phase 2
  if (32 * wave >= npx) return;                                         // (uniform) a tile without pixels
  f32x16 z0 = zero16(), z1 = zero16();
  const float* ap = slab + 8 * hh * C0A_SP + 32 * wave + l31;
  float rn[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rn[j] = ap[j * C0A_SP];
#pragma unroll
  for (int kq = 0; kq < KQMAX; ++kq) {
    if (kq < KQ0) {                                                     // uniform
      uint4 A1, A2, A3;
      a_split(make_float4(rn[0], rn[1], rn[2], rn[3]), make_float4(rn[4], rn[5], rn[6], rn[7]), A1, A2, A3);
      if (kq + 1 < KQ0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rn[j] = ap[((kq + 1) * 16 + j) * C0A_SP];
      }
      const uint4 (&b)[6] = bw[kq & 1];
      z0 = mfma_b3(A1, A2, A3, b[0], b[2], b[4], z0);
      z1 = mfma_b3(A1, A2, A3, b[1], b[3], b[5], z1);
      if (kq + 2 < KQ0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) bw[kq & 1][i] = wq[(((kq + 2) * 3 + (i >> 1)) * 2 + (i & 1)) * 64];
      }
    }
  }
  const float* bias = a.b + (long long)net * a.pstride;
  const float bv0 = bias[l31], bv1 = bias[32 + l31];
  float* out = a.a0 + (((long long)net * a.n + s) * HW + p0 + 32 * wave) * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = acc_row(r, lane);
    if (32 * wave + m < npx) {
      out[(long long)m * 64 + l31] = z0[r] + bv0;
      out[(long long)m * 64 + 32 + l31] = z1[r] + bv1;
    }
  }
}

bool conv0a_ok(int C, int HW) {
  return switches().conv0a != 0 && C >= 1 && C <= 128 && (HW & 7) == 0 && HW >= 8;
}

hipError_t launch_conv0a_fwd(int nets, int n, int C, int HW, const XSrc& xs, const float* w0b3, long long w0b3_ns,
                             const float* b, long long pstride, float* a0, float* xn, hipStream_t st) {
  if (!conv0a_ok(C, HW)) return hipErrorInvalidValue;
  Conv0aArgs a;
  a.xs = xs; a.w0b3 = w0b3; a.w0b3_ns = w0b3_ns; a.b = b; a.pstride = pstride; a.a0 = a0; a.xn = xn;
  a.n = n; a.C = C; a.HW = HW;
  const int KQ0 = (C + 15) / 16;
  const size_t lds = (size_t)KQ0 * 16 * C0A_SP * 4;
  static DevOnce attr_once;
  hipError_t e = ensure_max_lds(attr_once, conv0a_fwd_kernel<4>, conv0a_fwd_kernel<8>);
  if (e != hipSuccess) return e;
  dim3 grid((HW + 127) / 128, n, nets);
  if (KQ0 <= 4) hipLaunchKernelGGL((conv0a_fwd_kernel<4>), grid, dim3(256), lds, st, a);
  else          hipLaunchKernelGGL((conv0a_fwd_kernel<8>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// weight gradient.  One workgroup walks its samples; per sample the [C][HW] slab of xn is staged
// in LDS (row stride odd => conflict-free column reads), da0 rows come straight from HBM/L2 in
// double-buffered batches of 8 pixel pairs.
// wave w: co tile = w&1, band tiles (w>>1), (w>>1)+2, ...   (C0_MAXT tiles of 32 bands per wave)
// ------------------------------------------------------------------------------------------
constexpr int C0_MAXT = 4;    // up to 8 band tiles = 256 input channels
constexpr int C0_PB = 8;      // pixel pairs per batch (double-buffered); larger batches cost VGPRs -> occupancy (measured: 28 -> 54 us)

// DMA = true (odd HW): the LDS slab is a LINEAR copy of the sample's [C][HW] block (row stride HW is odd, so the
// column reads are conflict-free without padding) and is filled by global_load_lds_dwordx4 -- ~C*HW/256 wave
// instructions per workgroup instead of C*HW/256 scalar load + divide + ds_write per THREAD.
typedef __attribute__((address_space(3))) void c0_lds_void;
typedef __attribute__((address_space(1))) const void c0_gbl_void;
// PS > 1 (no DMA): a sample's pixels are walked in PS equal ranges, one slab of [Ct][HW / PS] at a time -- the 20 x 20
// window's whole slab is 103 KB, i.e. ONE workgroup per CU whose staging and MFMA phases cannot overlap with anything;
// half slabs let three workgroups share a CU.
// zstat (or null): the step's per-sample maxima table ([kind][2][n]: conv3x3.hip Conv3Args::hstat) + the networks' weight
// range flags: a sample whose conv1 gradient image was zero everywhere (a row no loss term reaches) has da0 = 0 -- it is not
// walked (unless its activations were not finite, or the weights were: 0 x inf stays NaN).
template <bool DMA>
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ xn, const float* __restrict__ da0,
                                                          float* __restrict__ part, int n, int C, int HWfull, int G,
                                                          int PS, const uint32_t* __restrict__ zstat,
                                                          const uint32_t* __restrict__ h2flag, long long h2flag_ns) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // xs[Ct][HWp] (+ 64 zero floats)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, g = blockIdx.x;
  const int NT = (C + 31) >> 5, Ct = NT * 32;
  const int HW = HWfull / PS;                  // pixels per slab (PS == 1 with DMA)
  const int HWp = DMA ? HW : ((HW + 2) | 1);   // odd; without DMA >= HW+1 (one zero pad column for odd HW)
  const int SPG = (n + G - 1) / G;
  const int sbeg = g * SPG, send = (sbeg + SPG < n) ? sbeg + SPG : n;
  const int ct = wave & 1, it0 = wave >> 1;
  f32x16 acc[C0_MAXT];
#pragma unroll
  for (int t = 0; t < C0_MAXT; ++t) acc[t] = zero16();
  float dbacc = 0.f;
  // rows >= C and columns >= HW stay zero for the whole kernel
  // (with DMA the slab rows [0, C) are overwritten in full by every sample: zero only the pad rows and the tail)
  for (int i = (DMA ? C * HW : 0) + tid; i < Ct * HWp + 64; i += 256) smem[i] = 0.f;
  const int pairs = (HW + 1) >> 1;

  const bool zflag = zstat != nullptr && h2flag[(long long)net * h2flag_ns] == 0u;
  for (int su = sbeg * PS; su < send * PS; ++su) {
    const int s = su / PS, p0 = (su - s * PS) * HW;              // sample, first pixel of this slab
    if (zflag && zstat[(2 * 2 + net) * n + s] == 0u && (zstat[(0 * 2 + net) * n + s] >> 23) < 255u) continue;   // (uniform)
    const float* brow = da0 + (((long long)net * n + s) * HWfull + p0) * 64 + ct * 32 + l31;
    float bq[4][C0_PB];             // ring of four batches: three requested ahead of the one being multiplied
    auto fetch = [&](float (&buf)[C0_PB], int t0) {
#pragma unroll
      for (int q = 0; q < C0_PB; ++q) {
        const int p = 2 * (t0 + q) + hh;
        buf[q] = brow[(long long)(p < HW ? p : 0) * 64];   // raw (masked at use): a select here would wait for the load
      }
    };
    // in flight while the slab is staged (three batches ahead of the one being multiplied)
    fetch(bq[0], 0); fetch(bq[1], C0_PB); fetch(bq[2], 2 * C0_PB);
    __syncthreads();
    const float* xs = xn + ((long long)net * n + s) * C * HWfull + p0;
    if (DMA) {
      // (the pixel past the end of an odd row is the first element of the next row: finite, and its B operand
      // is zero; the row after the last one is the zeroed tail)
      const int nf4 = (C * HW) >> 2;
      for (int q = wave; q * 64 < nf4; q += 4) {
        const int f = q * 64 + lane;
        if (f < nf4)
          __builtin_amdgcn_global_load_lds((c0_gbl_void*)(xs + 4 * f), (c0_lds_void*)(smem + q * 256), 16, 0, 0);
      }
      const int rem = C * HW - 4 * nf4;
      if (tid < rem) smem[4 * nf4 + tid] = xs[4 * nf4 + tid];
    } else {
      if ((HW & 3) == 0) {
        // rows of whole 16-byte pieces (e.g. the reference's 20 x 20 window): one load, one divide per FOUR elements
        const int H4 = HW >> 2;
        staged_copy<8, float4>(C * H4, tid,
                               [&](int i) { const int c = i / H4, q = i - c * H4; return *(const float4*)(xs + (long long)c * HWfull + 4 * q); },
                               [&](int i, float4 v) {
                                 const int c = i / H4, p = 4 * (i - c * H4);
                                 float* d = smem + c * HWp + p;
                                 d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                               });
      } else {
        staged_copy<16, float>(C * HW, tid, [&](int i) { const int c = i / HW; return xs[(long long)c * HWfull + (i - c * HW)]; },
                               [&](int i, float v) { const int c = i / HW, p = i - c * HW; smem[c * HWp + p] = v; });
      }
    }
    __syncthreads();
    for (int t1 = 0; t1 < pairs; t1 += 4 * C0_PB) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t0 = t1 + u * C0_PB;                       // workgroup-uniform
        if (t0 + 3 * C0_PB < pairs) fetch(bq[(u + 3) & 3], t0 + 3 * C0_PB);
        if (t0 < pairs) {
          // the batch's gradient values (zero past the last pixel: the tail batch multiplies zeros instead of branching
          // per pixel pair), then per band tile ALL its activation reads before its MFMAs: with a branch in front of
          // every MFMA each one waited out its own LDS read (~190 cycles per 64-cycle MFMA; 74 -> 52 us per launch at
          // 60 x 20 x 20.  Requesting the NEXT batch's reads before this batch's MFMAs as well gained nothing more.)
          float bv[C0_PB];
#pragma unroll
          for (int q = 0; q < C0_PB; ++q) {
            const int p = 2 * (t0 + q) + hh;
            bv[q] = (t0 + q < pairs && p < HW) ? bq[u][q] : 0.f;
            dbacc += bv[q];
          }
#pragma unroll
          for (int k = 0; k < C0_MAXT; ++k) {
            const int tile = it0 + 2 * k;
            if (tile < NT) {                                 // wave-uniform, the same for the whole kernel
              const float* xr = smem + (tile * 32 + l31) * HWp + 2 * t0 + hh;   // (reads past the row end stay inside the slab + tail)
              float av[C0_PB];
#pragma unroll
              for (int q = 0; q < C0_PB; ++q) av[q] = xr[2 * q];
              __builtin_amdgcn_sched_barrier(0);             // (else the reads are sunk to their MFMAs again)
#pragma unroll
              for (int q = 0; q < C0_PB; ++q) acc[k] = mfma32(av[q], bv[q], acc[k]);
            }
          }
        }
      }
    }
  }
  // partial layout [c][co] (+ 64 db)
  const int Cw = conv0_partial_rows(C);
  float* pp = part + ((long long)net * G + g) * ((long long)Cw * 64 + 64);
#pragma unroll
  for (int k = 0; k < C0_MAXT; ++k) {
    const int tile = it0 + 2 * k;
    if (tile < NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile * 32 + acc_row(r, lane);
        if (row < Cw) pp[row * 64 + ct * 32 + l31] = acc[k][r];
      }
    }
  }
  if (it0 == 0) {
    const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
    if (hh == 0) pp[(long long)Cw * 64 + ct * 32 + l31] = tot;
  }
}

// out[map(e)] = sum_g part[g][e]: 64 elements x 4 slices of g per block, 8 loads in flight per thread,
// fixed summation order (bit-reproducible, unlike float atomics).  Up to 3 tensors per launch.
// mode 0: conv0  e = c*64+co -> dW[co*C + c] (c < C), tail 64 -> db
// mode 1: conv3x3 e = s*4096 + ci*64 + co -> dW[co*576 + ci*9 + s], tail 64 -> db
// One block = 256 consecutive elements x 4 slices of g: a thread owns 4 consecutive elements (one 16-B load per
// partial row, a wave reads 1 KiB of a row at a time -- the 4-B-per-lane version read 256-B pieces of rows that lie
// tens of KB apart and reached 3.1 TB/s), 8 loads in flight, fixed summation order.
__device__ __forceinline__ void partial_reduce_block(const ReduceTable& t, int bx, int net, float (*red)[64]) {
  int pi = 0;
  if (t.count > 1 && bx >= t.p[1].blk0) pi = 1;
  if (t.count > 2 && bx >= t.p[2].blk0) pi = 2;
  const ReduceProb pr = t.p[pi];
  const int tid = threadIdx.x, el = tid & 63, sl = tid >> 6;
  // graph replay: this launch is where the device-side step cursor advances (no workgroup of this launch reads it;
  // the launches before it took their row at the old value, Adam behind it takes the row before the new one)
  if (t.dyn_cursor != nullptr && bx == 0 && net == 0 && tid == 0) {
    const int c = *t.dyn_cursor + 1;
    *t.dyn_cursor = c;
    const uint4* src = (const uint4*)(t.dyn_table + c);          // the next step's row becomes the working copy
    uint4* dst = (uint4*)t.dyn_table;
    const uint4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
    dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
  }
  const int G = pr.G, PS = pr.PS;                      // PS is a multiple of 4 (64-float tail, 4096-float taps)
  const int e0 = ((bx - pr.blk0) * 64 + el) * 4;
  const bool ev = e0 < PS;
  const float* p = pr.part + (long long)net * G * PS + (ev ? e0 : 0);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  for (int g0 = sl; g0 < G; g0 += 32) {
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int g = g0 + 4 * q;
      const float4 x = *(const float4*)(p + (size_t)(g < G ? g : 0) * PS);
      v[q] = (g < G) ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#define CMLPL_ADD4(F) s0.F += (v[0].F + v[1].F) + (v[2].F + v[3].F); s1.F += (v[4].F + v[5].F) + (v[6].F + v[7].F);
    CMLPL_ADD4(x) CMLPL_ADD4(y) CMLPL_ADD4(z) CMLPL_ADD4(w)
#undef CMLPL_ADD4
  }
  float4* red4 = (float4*)&red[0][0];                  // [4 slices][64] float4 = 4 KB
  red4[sl * 64 + el] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  if (sl == 0 && ev) {
    const float4 a = red4[el], b = red4[64 + el], c = red4[128 + el], d = red4[192 + el];
    const float sum[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                          (a.w + b.w) + (c.w + d.w)};
    const int body = PS - 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j;
      if (e >= body) {
        pr.db[(long long)net * t.grad_ns + (e - body)] = sum[j];
      } else if (pr.mode == 0) {
        const int c2 = e >> 6, co = e & 63;
        if (c2 < pr.C) pr.dW[(long long)net * t.grad_ns + co * pr.C + c2] = sum[j];
      } else {
        const int s = e >> 12, ci = (e >> 6) & 63, co = e & 63;
        pr.dW[(long long)net * t.grad_ns + co * 576 + ci * 9 + s] = sum[j];
      }
    }
  }
}

__global__ __launch_bounds__(256) void partial_reduce_kernel(ReduceTable t) {
  __shared__ __attribute__((aligned(16))) float red[16][64];
  partial_reduce_block(t, (int)blockIdx.x, (int)blockIdx.y, red);
}

// The weight-gradient reduce and the two small weight-gradient GEMMs (classifier, feat_spe) are independent and
// both short: one launch, reduce blocks first, GEMM tiles after them.
__global__ __launch_bounds__(256) void reduce_gemm_kernel(ReduceTable t, int reduce_blocks, GemmTN2 g) {
  __shared__ GemmTNShared sh;
  const int bid = (int)blockIdx.x;
  if (bid < reduce_blocks) partial_reduce_block(t, bid % t.total_blocks, bid / t.total_blocks, (float (*)[64])&sh.red[0][0][0]);   // 4 KB of the 12 KB
  else gemm_tn_block(g, bid - reduce_blocks, sh);
}

hipError_t launch_reduce_gemm(int nets, const ReduceTable& t, const GemmTN& g0, const GemmTN& g1, hipStream_t st) {
  GemmTN2 g;
  g.p[0] = g0; g.p[1] = g1; g.nblk0 = gemm_tn_blocks(g0);
  const int rb = t.total_blocks * nets, gb = g.nblk0 + gemm_tn_blocks(g1);
  hipLaunchKernelGGL(reduce_gemm_kernel, dim3(rb + gb), dim3(256), 0, st, t, rb, g);
  return hipGetLastError();
}

void reduce_table_add(ReduceTable& t, const float* part, int G, int PS, int mode, int C, float* dW, float* db) {
  ReduceProb& p = t.p[t.count++];
  p.part = part; p.dW = dW; p.db = db; p.G = G; p.PS = PS; p.mode = mode; p.C = C; p.blk0 = t.total_blocks;
  p.el = 256;
  t.total_blocks += (PS + 255) / 256;
}

hipError_t launch_partial_reduce(int nets, const ReduceTable& t, hipStream_t st) {
  hipLaunchKernelGGL(partial_reduce_kernel, dim3(t.total_blocks, nets), dim3(256), 0, st, t);
  return hipGetLastError();
}

int conv0_partial_size(int C) { return conv0_partial_rows(C) * 64 + 64; }

int plan_conv0_wgrad_G(int n, int C, int HW) {
  (void)C; (void)HW;
  int G = n < 256 ? n : 256;     // per net: one sample per workgroup up to 2 x 256 workgroups
  return G < 1 ? 1 : G;
}

hipError_t launch_conv0_wgrad(int nets, int n, int C, int HW, const float* xn, const float* da0, float* part,
                              hipStream_t st, const uint32_t* zstat, const uint32_t* h2flag, long long h2flag_ns) {
  if (switches().zero_skip == 0 || h2flag == nullptr) zstat = nullptr;
  const int NT = (C + 31) / 32, Ct = NT * 32;
  if (NT > 2 * C0_MAXT) return hipErrorInvalidValue;
  const bool dma_off = switches().conv0_dma == 0;
  const bool dma = (HW & 1) && !dma_off;
  // pixel ranges per sample: halves (of whole 16-byte groups) when the whole slab would leave one workgroup per CU
  const int force_ps = switches().conv0_ps;
  int PS = (!dma && (HW & 7) == 0 && ((size_t)Ct * ((HW + 2) | 1) + 64) * 4 > LDS_MAX / 2) ? 2 : 1;
  if (force_ps == 1 || (force_ps == 2 && !dma && (HW & 7) == 0)) PS = force_ps;
  const int HWs = HW / PS;
  const int HWp = dma ? HW : ((HWs + 2) | 1);
  const size_t lds = ((size_t)Ct * HWp + 64) * 4;
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static DevOnce attr_once;
  {
    hipError_t e = ensure_max_lds(attr_once, conv0_wgrad_kernel<false>, conv0_wgrad_kernel<true>);
    if (e != hipSuccess) return e;
  }
  const int G = plan_conv0_wgrad_G(n, C, HW);
  if (dma) hipLaunchKernelGGL(conv0_wgrad_kernel<true>, dim3(G, nets), dim3(256), lds, st, xn, da0, part, n, C, HW, G, 1, zstat, h2flag, h2flag_ns);
  else     hipLaunchKernelGGL(conv0_wgrad_kernel<false>, dim3(G, nets), dim3(256), lds, st, xn, da0, part, n, C, HW, G, PS, zstat, h2flag, h2flag_ns);
  return hipGetLastError();
}

}  // namespace cmlpl
