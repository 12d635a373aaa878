// 3x3 / pad 1 / 64->64 convolutions of BaseNet2 (tools/models.py:104-107,134-140) as
// LDS-staged implicit GEMMs on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   conv3x3_kernel<FWD>   : z = conv(in)+bias+in ; r = relu(z) ; out = avgpool2(r) ; mask = (r>0) nibble
//                           (models.py:134-136 / 138-140, fused)
//   conv3x3_kernel<DGRAD> : dz = mask * upsample(dpool)/4 (avgpool+relu backward, formed while staging)
//                           out = conv_transpose(dz) + dz      (residual branch adds dz itself)
//   wgrad3_kernel         : dW[s][ci][co] = sum_pix in[pix+s][ci] * dz[pix][co], db[co] = sum dz
//
// Data layout: activations are pixel-major / channel-last  [net][sample][pixel][64] so that
// the 64 channels of a pixel are one 256-B line; the padded image of S samples sits in LDS
// with a zero border, so the 9 taps are pure address offsets.  Weights are re-packed once per
// step (pack_weights_kernel) to [tap][ci/4][co][4] so that both MFMA operands are ds_read_b128.
#include <stdlib.h>

#ifndef CMLPL_ABL
#define CMLPL_ABL 0
#endif

#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

constexpr int CS = 68;  // LDS pixel stride in floats (64 + 4): conflict-free ds_read_b128 over 16 pixels

// ------------------------------------------------------------------------------------------
// weight packing: canonical W[co][ci][kh][kw] ->
//   fwd  : wf[s][q][co][r] = W[co][4q+r][kh][kw]            s = kh*3+kw
//   dgrad: wd[s][q][ci][r] = W[4q+r][ci][2-kh][2-kw]        (transposed + flipped)
// ------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ params, long long pstride, PackInfo pi,
                                    float* __restrict__ packed) {
  const int net = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= pi.stride) return;
  const float* P = params + (long long)net * pstride;
  float v;
  if (e < PACK_PER_NET) {
    const int which = (int)(e / PACK_CONV);          // 0 c1 fwd, 1 c1 dgrad, 2 c2 fwd, 3 c2 dgrad
    const int i = (int)(e - (long long)which * PACK_CONV);
    const int r = i & 3, oc = (i >> 2) & 63, q = (i >> 8) & 15, s = i >> 12;
    const int kh = s / 3, kw = s - kh * 3;
    const float* W = P + ((which < 2) ? pi.off_w1 : pi.off_w2);
    const int k = 4 * q + r;
    if ((which & 1) == 0) v = W[((oc * 64 + k) * 3 + kh) * 3 + kw];                    // co=oc, ci=k
    else                  v = W[((k * 64 + oc) * 3 + (2 - kh)) * 3 + (2 - kw)];        // co=k, ci=oc
  } else if (e < pack_off_wst(pi.C)) {
    const int i = (int)(e - pack_off_w0t()), c = i >> 6, co = i & 63;
    v = (c < pi.C) ? P[pi.off_w0 + (long long)co * pi.C + c] : 0.f;
  } else {
    const long long i = e - pack_off_wst(pi.C);
    const int band = (int)(i >> 10), o = (int)(i & 1023);
    v = P[pi.off_ws + (long long)o * pi.bands + band];
  }
  packed[(long long)net * pi.stride + e] = v;
}

hipError_t launch_pack_weights(int nets, const float* params, long long pstride, const PackInfo& pi, float* packed,
                               hipStream_t st) {
  dim3 grid((unsigned)((pi.stride + 255) / 256), nets);
  hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, st, params, pstride, pi, packed);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------
#if CMLPL_ABL == 9
// phase timeline instrumentation (ablation build only): constant-rate 100 MHz stamps per workgroup
__device__ unsigned long long g_stamps[3][2048][8];
#define STAMP(MODE_, i) do { if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 2048) \
    g_stamps[MODE_][blockIdx.x + gridDim.x * blockIdx.y][i] = wall_clock64(); } while (0)
extern "C" int cmlpl_abl_read_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#else
#define STAMP(MODE_, i) do {} while (0)
#endif

struct Conv3Args {
  const float* in; const uint8_t* mask_in; const float* wpk; const float* bias;
  float* out; uint8_t* mask_out;
  long long in_ns, mask_in_ns, wpk_ns, bias_ns, out_ns, mask_out_ns;  // per-net strides (elements)
  int n, H, W, S;
};

// The 9-tap main loop.  NTA = number of this wave's M tiles that carry real pixels (wave-uniform, so
// the loop body is branch-free and the compiler can hoist the ds_read_b128 of step kk+1 above the
// MFMAs of step kk).  Tap weights go global -> registers (prefetched one tap ahead) -> LDS.
template <int MTW, int NTA>
__device__ __forceinline__ void conv3_taps(const float* __restrict__ img, float* __restrict__ wbuf,
                                           const float4* __restrict__ wg, float4 w0, float4 w1, float4 w2,
                                           float4 w3, const int (&abase)[MTW], f32x16 (&acc)[MTW][2], int PW,
                                           int tid, int l31, int hh) {
  float4* wl = (float4*)wbuf;
  const float* bbase = wbuf + (hh * 64 + l31) * 4;
#pragma unroll 1
  for (int s = 0; s < 9; ++s) {
#if CMLPL_ABL == 6
    if (s == 0) {
#endif
    __syncthreads();  // everyone done with wbuf of tap s-1 (and, for s == 0, the staged image is complete)
    wl[tid] = w0; wl[tid + 256] = w1; wl[tid + 512] = w2; wl[tid + 768] = w3;
    __syncthreads();
#if CMLPL_ABL == 6
    }
#endif
    if (s + 1 < 9) {
      const float4* wn = wg + (s + 1) * 1024 + tid;
      w0 = wn[0]; w1 = wn[256]; w2 = wn[512]; w3 = wn[768];
    }
    if (NTA > 0) {
      const int kh = s / 3, kw = s - kh * 3;
      const float* ib = img + ((kh - 1) * PW + (kw - 1)) * CS;
      // software pipeline: the ds_read_b128 of step kk+1 are issued BEFORE the MFMAs of step kk and the
      // order is pinned with sched_barrier (left alone, the scheduler sinks the reads below the MFMAs
      // and every step then eats a full LDS round trip on a wave that is alone on its SIMD)
      float4 b0 = *(const float4*)(bbase), b1 = *(const float4*)(bbase + 128);
      float4 av[NTA > 0 ? NTA : 1];
#pragma unroll
      for (int t = 0; t < NTA; ++t) av[t] = *(const float4*)(ib + abase[t]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float4 nb0 = b0, nb1 = b1, nav[NTA > 0 ? NTA : 1];
#pragma unroll
        for (int t = 0; t < NTA; ++t) nav[t] = av[t];
        if (kk < 7) {
          nb0 = *(const float4*)(bbase + (kk + 1) * 512);
          nb1 = *(const float4*)(bbase + (kk + 1) * 512 + 128);
#pragma unroll
          for (int t = 0; t < NTA; ++t) nav[t] = *(const float4*)(ib + abase[t] + (kk + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#if CMLPL_ABL == 5
#pragma unroll
        for (int t = 0; t < NTA; ++t) asm volatile("" :: "v"(av[t].x), "v"(av[t].w), "v"(b0.x), "v"(b1.w));
#else
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
          acc[t][0] = mfma32(av[t].x, b0.x, acc[t][0]);
          acc[t][1] = mfma32(av[t].x, b1.x, acc[t][1]);
          acc[t][0] = mfma32(av[t].y, b0.y, acc[t][0]);
          acc[t][1] = mfma32(av[t].y, b1.y, acc[t][1]);
          acc[t][0] = mfma32(av[t].z, b0.z, acc[t][0]);
          acc[t][1] = mfma32(av[t].z, b1.z, acc[t][1]);
          acc[t][0] = mfma32(av[t].w, b0.w, acc[t][0]);
          acc[t][1] = mfma32(av[t].w, b1.w, acc[t][1]);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        b0 = nb0; b1 = nb1;
#pragma unroll
        for (int t = 0; t < NTA; ++t) av[t] = nav[t];
      }
    }
  }
}

// Everything a 3x3 workgroup does before its tap loop: zero-bordered LDS image of S samples (FWD: the
// activation; DGRAD: dz = mask * upsample(dpool) / 4 formed on the fly), output-pixel LUT, and the tap-0
// weights requested early so their latency overlaps the staging.
struct Conv3Ctx {
  int tid, lane, l31, hh, wave, net, s0, H, W, HW, PW, IMG, H2, W2, P2, RO, CO, PX, S, npx;
  float* img; float* wbuf; int* lut; const float4* wg;
  float4 wp0, wp1, wp2, wp3;
};

template <int MODE>
__device__ __forceinline__ void conv3_stage(const Conv3Args& a, float* smem, int lut_entries, Conv3Ctx& c) {
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, s0 = blockIdx.x * a.S;
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2, IMG = (H + 2) * PW;
  const int H2 = H >> 1, W2 = W >> 1, P2 = H2 * W2;
  const int RO = (MODE == 0) ? 2 * H2 : H, CO = (MODE == 0) ? 2 * W2 : W;
  const int PX = RO * CO, S = a.S, npx = S * PX;
  float* img = smem;                       // [S][IMG][CS]
  float* wbuf = img + (size_t)S * IMG * CS;  // [16][64][4]
  int* lut = (int*)(wbuf + 4096);          // padded-image position of output pixel m
  {  // zero the padded images (border must be zero; interior overwritten below)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* p = (float4*)img;
    const int tot = S * IMG * (CS / 4);
    for (int i = tid; i < tot; i += 256) p[i] = z;
  }
  for (int m = tid; m < lut_entries; m += 256) {
    const int mm = (m < npx) ? m : 0;
    const int s = mm / PX, rem = mm - s * PX, r = rem / CO, c = rem - r * CO;
    lut[m] = s * IMG + (r + 1) * PW + (c + 1);
  }
  const float4* wg = (const float4*)(a.wpk + (long long)net * a.wpk_ns);
  // tap-0 weights: issued now so the HBM/L2 latency overlaps the image staging below
  c.wp0 = wg[tid]; c.wp1 = wg[tid + 256]; c.wp2 = wg[tid + 512]; c.wp3 = wg[tid + 768];
  __syncthreads();

  if (MODE == 0) {
    const float* src = a.in + (long long)net * a.in_ns;
    staged_copy<8, float4>(S * HW * 16, tid,
        [&](int idx) {
          const int c4 = idx & 15, p = idx >> 4, s = p / HW, pix = p - s * HW, sample = s0 + s;
          const bool ok = sample < a.n;
          const float4 v = *(const float4*)(src + ((size_t)(ok ? sample : s0) * HW + pix) * 64 + c4 * 4);
          return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        },
        [&](int idx, float4 v) {
          const int c4 = idx & 15, p = idx >> 4, s = p / HW, pix = p - s * HW, h = pix / W, w = pix - h * W;
          *(float4*)(img + (size_t)(s * IMG + (h + 1) * PW + w + 1) * CS + c4 * 4) = v;
        });
  } else {
    const float* dp = a.in + (long long)net * a.in_ns;
    const uint8_t* mk = a.mask_in + (long long)net * a.mask_in_ns;
    struct DM { float4 d; uint32_t m; };
    // one (pooled pixel, 4 channels) item feeds the 4 full-resolution positions of its 2x2 window
    staged_copy<8, DM>(S * P2 * 16, tid,
        [&](int idx) {
          const int c4 = idx & 15, pp = idx >> 4, s = pp / P2, q = pp - s * P2, sample = s0 + s;
          const bool ok = sample < a.n;
          const size_t g = ((size_t)(ok ? sample : s0) * P2 + q) * 64 + c4 * 4;
          DM r;
          r.d = *(const float4*)(dp + g);
          r.m = ok ? *(const uint32_t*)(mk + g) : 0u;
          return r;
        },
        [&](int idx, DM r) {
          const int c4 = idx & 15, pp = idx >> 4, s = pp / P2, q = pp - s * P2, ph = q / W2, pw = q - ph * W2;
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            float4 v;
            v.x = ((r.m >> sub) & 1u) ? r.d.x * 0.25f : 0.f;
            v.y = ((r.m >> (8 + sub)) & 1u) ? r.d.y * 0.25f : 0.f;
            v.z = ((r.m >> (16 + sub)) & 1u) ? r.d.z * 0.25f : 0.f;
            v.w = ((r.m >> (24 + sub)) & 1u) ? r.d.w * 0.25f : 0.f;
            const int h = 2 * ph + (sub >> 1), w = 2 * pw + (sub & 1);
            *(float4*)(img + (size_t)(s * IMG + (h + 1) * PW + w + 1) * CS + c4 * 4) = v;
          }
        });
  }

  c.tid = tid; c.lane = lane; c.l31 = l31; c.hh = hh; c.wave = wave; c.net = net; c.s0 = s0;
  c.H = H; c.W = W; c.HW = HW; c.PW = PW; c.IMG = IMG; c.H2 = H2; c.W2 = W2; c.P2 = P2;
  c.RO = RO; c.CO = CO; c.PX = PX; c.S = S; c.npx = npx;
  c.img = img; c.wbuf = wbuf; c.lut = lut; c.wg = wg;
}

// avgpool2 + ReLU-mask epilogue shared by the forward kernels (img holds relu(z) at the pixel centres)
__device__ __forceinline__ void conv3_pool_store(const Conv3Args& a, const Conv3Ctx& c) {
  float* out = a.out + (long long)c.net * a.out_ns;
  uint8_t* mo = a.mask_out + (long long)c.net * a.mask_out_ns;
  const int tot = c.S * c.P2 * 64;
  for (int idx = c.tid; idx < tot; idx += 256) {
    const int co = idx & 63, pp = idx >> 6;
    const int s = pp / c.P2, q = pp - s * c.P2, ph = q / c.W2, pw = q - ph * c.W2;
    const int sample = c.s0 + s;
    if (sample < a.n) {
      const float* p = c.img + (size_t)(s * c.IMG + (2 * ph + 1) * c.PW + 2 * pw + 1) * CS + co;
      const float v00 = p[0], v01 = p[CS], v10 = p[c.PW * CS], v11 = p[c.PW * CS + CS];
      const size_t g = ((size_t)sample * c.P2 + q) * 64 + co;
      out[g] = (v00 + v01 + v10 + v11) * 0.25f;
      mo[g] = (uint8_t)((v00 > 0.f ? 1 : 0) | (v01 > 0.f ? 2 : 0) | (v10 > 0.f ? 4 : 0) | (v11 > 0.f ? 8 : 0));
    }
  }
}

template <int MODE, int MTW>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Conv3Ctx c;
  STAMP(MODE, 0);
  conv3_stage<MODE>(a, smem, MTW * 128, c);
  STAMP(MODE, 1);
  const int tid = c.tid, lane = c.lane, l31 = c.l31, hh = c.hh, wave = c.wave, net = c.net, s0 = c.s0;
  const int HW = c.HW, PW = c.PW, S = c.S, PX = c.PX, npx = c.npx;
  float* img = c.img; float* wbuf = c.wbuf; int* lut = c.lut; const float4* wg = c.wg;
  const float4 wp0 = c.wp0, wp1 = c.wp1, wp2 = c.wp2, wp3 = c.wp3;
  const int MT = (npx + 31) >> 5;
  int abase[MTW];
  f32x16 acc[MTW][2];
#pragma unroll
  for (int t = 0; t < MTW; ++t) {
    abase[t] = lut[(wave + 4 * t) * 32 + l31] * CS + 4 * hh;
    acc[t][0] = zero16();
    acc[t][1] = zero16();
  }

  // tiles t <= MTW-2 are always active; only the last one may be missing for some waves (wave-uniform)
  if (wave + 4 * (MTW - 1) < MT) conv3_taps<MTW, MTW>(img, wbuf, wg, wp0, wp1, wp2, wp3, abase, acc, PW, tid, l31, hh);
  else                           conv3_taps<MTW, MTW - 1>(img, wbuf, wg, wp0, wp1, wp2, wp3, abase, acc, PW, tid, l31, hh);
  __syncthreads();  // all MFMA reads of img are done; the epilogue overwrites it in place
  STAMP(MODE, 2);

  if (MODE == 0) {
    const float* bias = a.bias + (long long)net * a.bias_ns;
    const float bv0 = bias[l31], bv1 = bias[32 + l31];
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < MT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tile * 32 + acc_row(r, lane);
          if (m < npx) {
            float* p = img + (size_t)lut[m] * CS;
            const float v0 = acc[t][0][r] + bv0 + p[l31];
            const float v1 = acc[t][1][r] + bv1 + p[32 + l31];
            p[l31] = fmaxf(v0, 0.f);
            p[32 + l31] = fmaxf(v1, 0.f);
          }
        }
      }
    }
    __syncthreads();
    conv3_pool_store(a, c);
  } else {
    float* out = a.out + (long long)net * a.out_ns;
    const int nvalid = (a.n - s0 < S ? a.n - s0 : S) * PX;  // rows that map to real samples
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < MT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tile * 32 + acc_row(r, lane);
          if (m < nvalid) {
            const float* p = img + (size_t)lut[m] * CS;
            float* o = out + ((size_t)s0 * HW + m) * 64;
            o[l31] = acc[t][0][r] + p[l31];
            o[32 + l31] = acc[t][1][r] + p[32 + l31];
          }
        }
      }
    }
  }
  STAMP(MODE, 3);
}


// Small-map variant (all output pixels of the workgroup fit ONE 32-row tile, e.g. the 5x5 / 4x4 maps of
// conv2 at 11x11 windows): instead of one busy wave and three idle ones, wave w takes output-channel half
// (w & 1) and input-channel half (w >> 1); the two K halves are folded through LDS before the epilogue.
template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(Conv3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Conv3Ctx c;
  conv3_stage<MODE>(a, smem, 128, c);
  const int tid = c.tid, lane = c.lane, l31 = c.l31, hh = c.hh, wave = c.wave, net = c.net;
  const int nt = wave & 1, kh2 = wave >> 1;
  float* img = c.img; float* wbuf = c.wbuf; const int* lut = c.lut;
  const int abase = lut[l31] * CS + 4 * hh;
  f32x16 acc = zero16();
  {
    float4 w0 = c.wp0, w1 = c.wp1, w2 = c.wp2, w3 = c.wp3;
    float4* wl = (float4*)wbuf;
    const float* bbase = wbuf + (hh * 64 + nt * 32 + l31) * 4;
#pragma unroll 1
    for (int s = 0; s < 9; ++s) {
      __syncthreads();
      wl[tid] = w0; wl[tid + 256] = w1; wl[tid + 512] = w2; wl[tid + 768] = w3;
      __syncthreads();
      if (s + 1 < 9) {
        const float4* wn = c.wg + (s + 1) * 1024 + tid;
        w0 = wn[0]; w1 = wn[256]; w2 = wn[512]; w3 = wn[768];
      }
      const int khh = s / 3, kww = s - khh * 3;
      const float* ib = img + ((khh - 1) * c.PW + (kww - 1)) * CS + abase;
      float4 av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = kh2 * 4 + q;
        av[q] = *(const float4*)(ib + kk * 8);
        bv[q] = *(const float4*)(bbase + kk * 512);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc = mfma32(av[q].x, bv[q].x, acc);
        acc = mfma32(av[q].y, bv[q].y, acc);
        acc = mfma32(av[q].z, bv[q].z, acc);
        acc = mfma32(av[q].w, bv[q].w, acc);
      }
    }
  }
  __syncthreads();                       // tap loop done everywhere: wbuf becomes the K-half exchange
  float* xch = wbuf + nt * 1024;         // [16][64] per output-channel half
  if (kh2 == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (kh2 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += xch[r * 64 + lane];
    if (MODE == 0) {
      const float bv = (a.bias + (long long)net * a.bias_ns)[nt * 32 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, lane);
        if (m < c.npx) {
          float* p = img + (size_t)lut[m] * CS + nt * 32 + l31;
          *p = fmaxf(acc[r] + bv + *p, 0.f);
        }
      }
    } else {
      float* out = a.out + (long long)net * a.out_ns;
      const int nvalid = (a.n - c.s0 < c.S ? a.n - c.s0 : c.S) * c.PX;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, lane);
        if (m < nvalid) {
          const float* p = img + (size_t)lut[m] * CS + nt * 32 + l31;
          out[((size_t)c.s0 * c.HW + m) * 64 + nt * 32 + l31] = acc[r] + *p;
        }
      }
    }
  }
  if (MODE == 0) {
    __syncthreads();
    conv3_pool_store(a, c);
  }
}

static size_t conv3_lds(int S, int H, int W, int MTW) {
  return ((size_t)S * (H + 2) * (W + 2) * CS + 4096 + (size_t)MTW * 128) * 4;
}

// Pick samples-per-workgroup S.  Cost model: MFMA tile-times queued on the busiest SIMD (workgroups
// on a CU share its 4 SIMDs, wave w of every workgroup lands on a different SIMD) plus a fixed
// per-workgroup staging/drain overhead that is hidden when a second workgroup is co-resident.
bool plan_conv3(int mode, int H, int W, int rows, Conv3Plan* p) {
  const int PX = (mode == 0) ? (2 * (H / 2)) * (2 * (W / 2)) : H * W;
  if (PX <= 0) return false;
  static const int force_s = getenv("CMLPL_CONV3_S") ? atoi(getenv("CMLPL_CONV3_S")) : 0;
  double best = 1e30;
  bool ok = false;
  for (int S = 1; S <= 16; ++S) {
    const int MT = (S * PX + 31) / 32, MTW = (MT + 3) / 4;
    if (MTW > 4) break;
    const bool split = (MT == 1);          // one tile: the 4 waves split (co half, ci half) instead
    const size_t lds = conv3_lds(S, H, W, MTW);
    if (lds > LDS_MAX) break;
    const long long wgs = (rows + S - 1) / S;
    const int resident = (int)(LDS_MAX / lds) < 4 ? (int)(LDS_MAX / lds) : 4;    // workgroups per CU
    const long long per_cu = (wgs + 255) / 256;                                    // workgroups queued per CU
    const long long waves_deep = (per_cu + resident - 1) / resident;               // sequential rounds
    const double mfma = (double)per_cu * (split ? 0.25 : MTW);                     // tile-times on the busiest SIMD
    const double overhead = 0.45 * (double)waves_deep + (resident > 1 ? 0.0 : 0.15 * MTW);
    const double cost = mfma + overhead;
    if (force_s ? (S == force_s) : (cost < best - 1e-9)) {
      best = cost; p->S = S; p->MTW = split ? 0 : MTW; p->lds = lds; ok = true;
    }
  }
  return ok;
}

template <int MODE, int MTW>
static hipError_t launch_conv3_t(const Conv3Args& a, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_kernel<MODE, MTW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((conv3x3_kernel<MODE, MTW>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_conv3(int mode, int nets, int n, int H, int W, const float* in, const uint8_t* mask_in,
                        const float* wpk, long long wpk_ns, const float* bias, long long bias_ns,
                        float* out, uint8_t* mask_out, hipStream_t st) {
  Conv3Plan pl;
  if (!plan_conv3(mode, H, W, nets * n, &pl)) return hipErrorInvalidValue;
  const int HW = H * W, P2 = (H / 2) * (W / 2);
  Conv3Args a;
  a.in = in; a.mask_in = mask_in; a.wpk = wpk; a.bias = bias; a.out = out; a.mask_out = mask_out;
  a.wpk_ns = wpk_ns; a.bias_ns = bias_ns;
  if (mode == 0) { a.in_ns = (long long)n * HW * 64; a.out_ns = (long long)n * P2 * 64; a.mask_out_ns = a.out_ns; a.mask_in_ns = 0; }
  else           { a.in_ns = (long long)n * P2 * 64; a.mask_in_ns = a.in_ns; a.out_ns = (long long)n * HW * 64; a.mask_out_ns = 0; }
  a.n = n; a.H = H; a.W = W; a.S = pl.S;
  dim3 grid((n + pl.S - 1) / pl.S, nets);
#define CMLPL_DISPATCH(M)                                                      \
  switch (pl.MTW) {                                                            \
    case 0: {                                                                  \
      static bool attr0 = false;                                               \
      if (!attr0) {                                                            \
        hipError_t e0 = hipFuncSetAttribute((const void*)conv3x3_small_kernel<M>,                          \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);      \
        if (e0 != hipSuccess) return e0;                                       \
        attr0 = true;                                                          \
      }                                                                        \
      hipLaunchKernelGGL((conv3x3_small_kernel<M>), grid, dim3(256), pl.lds, st, a);                        \
      return hipGetLastError();                                                \
    }                                                                          \
    case 1: return launch_conv3_t<M, 1>(a, grid, pl.lds, st);                  \
    case 2: return launch_conv3_t<M, 2>(a, grid, pl.lds, st);                  \
    case 3: return launch_conv3_t<M, 3>(a, grid, pl.lds, st);                  \
    default: return launch_conv3_t<M, 4>(a, grid, pl.lds, st);                 \
  }
  if (mode == 0) { CMLPL_DISPATCH(0) } else { CMLPL_DISPATCH(1) }
#undef CMLPL_DISPATCH
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
struct Wgrad3Args {
  const float* in; const float* dpool; const uint8_t* mask; float* part;
  long long in_ns, dpool_ns, part_ns;
  int n, H, W, RU, U, G;
};

// CSPL = 1: one workgroup produces all 64 output channels (wave = (co tile, ci tile), 9 taps each).
// CSPL = 2: the output channels are split over two workgroups (blockIdx.z = co half); a workgroup then stages
//           only its half of dz (LDS ~53 KB instead of 131 KB for 11x11 maps), so 2-3 workgroups share a CU and
//           one's staging / partial write-out overlaps another's MFMAs; wave = (ci tile, tap parity), 5 or 4 taps.
template <int CSPL>
__global__ __launch_bounds__(256) void wgrad3_kernel(Wgrad3Args a) {
  constexpr int DC = 64 / CSPL;            // dz channels staged by this workgroup
  constexpr int NS = (CSPL == 1) ? 9 : 5;  // tap slots per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, g = blockIdx.x;
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2;
  const int H2 = H >> 1, W2 = W >> 1, P2 = H2 * W2, RO = 2 * H2, CO = 2 * W2;
  const int RU = a.RU, U = a.U;
  const int IMGU = (RU + 2) * PW;          // padded rows of one unit
  const int DU = RU * CO;                  // dz slots per unit
  const int D = U * DU;                    // dz slots per pass
  const int UPS = (RO + RU - 1) / RU;      // units per sample
  const int NU = a.n * UPS;
  const int UPG = (NU + a.G - 1) / a.G;
  const int ubeg = g * UPG, uend = (ubeg + UPG < NU) ? ubeg + UPG : NU;

  float* img = smem;                        // [U][IMGU][64]
  float* dz = img + (size_t)U * IMGU * 64;  // [D][DC]

  for (int i = tid; i < U * IMGU * 64; i += 256) img[i] = 0.f;
  const float* src = a.in + (long long)net * a.in_ns;
  const float* dp = a.dpool + (long long)net * a.dpool_ns;
  const uint8_t* mk = a.mask + (long long)net * a.dpool_ns;

  const int ct = (CSPL == 1) ? (wave & 1) : (int)blockIdx.z;
  const int it = (CSPL == 1) ? (wave >> 1) : (wave & 1);
  const int wh = (CSPL == 1) ? 0 : (wave >> 1);
  f32x16 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = zero16();
  float dbacc = 0.f;
  int shoff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tap = (CSPL == 1) ? s : (2 * s + wh < 9 ? 2 * s + wh : 0);   // inactive slot: any valid address
    shoff[s] = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * 64;
  }
  const bool last_active = (CSPL == 1) || (wh == 0);      // slot NS-1 exists only for tap parity 0 (wave-uniform)

  for (int ub = ubeg; ub < uend; ub += U) {
    __syncthreads();  // previous pass finished reading img/dz
#if CMLPL_ABL == 3
    if (ub >= 0) goto staged;
#endif
    // stage the input rows (row0-1 .. row0+RU) of each unit; rows outside the image are zero
    staged_copy<8, float4>(U * (RU + 2) * W * 16, tid,
        [&](int idx) {
          const int c4 = idx & 15, p = idx >> 4;
          const int u = p / ((RU + 2) * W), rem = p - u * (RU + 2) * W, ir = rem / W, w = rem - ir * W;
          const int uid = ub + u;
          const int uc = (uid < uend) ? uid : ubeg;
          const int sample = uc / UPS, j = uc - sample * UPS, row = j * RU - 1 + ir;
          const bool ok = (uid < uend) && row >= 0 && row < H;
          const float4 v = *(const float4*)(src + ((size_t)sample * HW + (ok ? row : 0) * W + w) * 64 + c4 * 4);
          return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        },
        [&](int idx, float4 v) {
          const int c4 = idx & 15, p = idx >> 4;
          const int u = p / ((RU + 2) * W), rem = p - u * (RU + 2) * W, ir = rem / W, w = rem - ir * W;
          *(float4*)(img + (size_t)(u * IMGU + ir * PW + w + 1) * 64 + c4 * 4) = v;
        });
    // stage dz = mask * dpool / 4 for the unit's output rows (one pooled item -> its 2x2 window)
    {
      struct DM { float4 d; uint32_t m; };
      const int RUh = RU >> 1;
      constexpr int C4N = 16 / CSPL;       // float4 chunks of this workgroup's channel range
      staged_copy<8, DM>(U * RUh * W2 * C4N, tid,
          [&](int idx) {
            const int c4 = idx % C4N, p = idx / C4N;
            const int u = p / (RUh * W2), rem = p - u * RUh * W2, rh = rem / W2, pw = rem - rh * W2;
            const int uid = ub + u;
            const int uc = (uid < uend) ? uid : ubeg;
            const int sample = uc / UPS, j = uc - sample * UPS, prow = j * RUh + rh;   // pooled row
            const bool ok = (uid < uend) && prow < H2;
            const size_t gi = ((size_t)sample * P2 + (ok ? prow : 0) * W2 + pw) * 64 + (CSPL == 1 ? 0 : ct * 32) + c4 * 4;
            DM r;
            r.d = *(const float4*)(dp + gi);
            r.m = ok ? *(const uint32_t*)(mk + gi) : 0u;
            return r;
          },
          [&](int idx, DM r) {
            const int c4 = idx % C4N, p = idx / C4N;
            const int u = p / (RUh * W2), rem = p - u * RUh * W2, rh = rem / W2, pw = rem - rh * W2;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
              float4 v;
              v.x = ((r.m >> sub) & 1u) ? r.d.x * 0.25f : 0.f;
              v.y = ((r.m >> (8 + sub)) & 1u) ? r.d.y * 0.25f : 0.f;
              v.z = ((r.m >> (16 + sub)) & 1u) ? r.d.z * 0.25f : 0.f;
              v.w = ((r.m >> (24 + sub)) & 1u) ? r.d.w * 0.25f : 0.f;
              const int d = u * DU + (2 * rh + (sub >> 1)) * CO + 2 * pw + (sub & 1);
              *(float4*)(dz + (size_t)d * DC + c4 * 4) = v;
            }
          });
    }
#if CMLPL_ABL == 3
  staged:
#endif
    __syncthreads();
    // main loop over pixel pairs (c, c+1) of each staged output row; CO is even, so a pair never
    // straddles a row and every address is affine in (row, c): no lookup, operands of pair t+1 are
    // fetched while the 9 MFMAs of pair t run.  lane half hh takes pixel c+hh of the pair.
    const int rows = U * RU, cpr = CO >> 1;
    const float* arow0 = img + it * 32 + l31 + (PW + 1 + hh) * 64;   // (r+1)*PW + (c+1) with r = c = 0
    const float* brow0 = dz + (CSPL == 1 ? ct * 32 : 0) + l31 + hh * DC;
    float an[NS], bn;
    {
#pragma unroll
      for (int s = 0; s < NS; ++s) an[s] = arow0[shoff[s]];
      bn = brow0[0];
    }
    int u = 0, r = 0, cp = 0;
    const int pairs = rows * cpr;
    float ac[NS], bc;
#pragma unroll
    for (int s = 0; s < NS; ++s) ac[s] = an[s];
    bc = bn;
    for (int t = 0; t < pairs; ++t) {
      // advance (u, r, cp) and fetch the next pair (the last fetch re-reads pair 0: harmless)
      if (++cp == cpr) { cp = 0; if (++r == RU) { r = 0; ++u; } }
      const int un = (t + 1 < pairs) ? u : 0, rn = (t + 1 < pairs) ? r : 0, cn = (t + 1 < pairs) ? cp : 0;
      const float* ap = arow0 + (un * IMGU + rn * PW + 2 * cn) * 64;
      const float* bp = brow0 + ((un * RU + rn) * CO + 2 * cn) * DC;
#if CMLPL_ABL != 2
#pragma unroll
      for (int s = 0; s < NS; ++s) an[s] = ap[shoff[s]];
      bn = bp[0];
#else
      asm volatile("" :: "v"(ap), "v"(bp));
#endif
      __builtin_amdgcn_sched_barrier(0);   // reads of pair t+1 stay above the MFMAs of pair t
      dbacc += bc;
#if CMLPL_ABL != 1
#pragma unroll
      for (int s = 0; s < NS - 1; ++s) acc[s] = mfma32(ac[s], bc, acc[s]);
      if (last_active) acc[NS - 1] = mfma32(ac[NS - 1], bc, acc[NS - 1]);
#else
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("" :: "v"(ac[s]), "v"(bc));
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NS; ++s) ac[s] = an[s];
      bc = bn;
    }
  }

  float* part = a.part + (long long)net * a.part_ns + (size_t)g * PART3;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tap = (CSPL == 1) ? s : 2 * s + wh;
    if (tap < 9) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = it * 32 + acc_row(r, lane);
        part[tap * 4096 + ci * 64 + ct * 32 + l31] = acc[s][r];
      }
    }
  }
  // bias gradient: one wave per output-channel tile summed its B operand; fold the two pixel parities
  if (it == 0 && wh == 0) {
    const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
    if (hh == 0) part[9 * 4096 + ct * 32 + l31] = tot;
  }
}

// Pipelined variant (the default whenever a stage of U units needs <= NI float4 per thread): LDS holds TWO
// stages; while the 9-tap MFMA loop runs on stage g, the rows of stage g+1 are already in flight global ->
// registers, and are written to the other LDS half after the loop.  One barrier per stage.  A workgroup has a
// single wave per SIMD here (4 waves, 144 accumulator registers each), so nothing else would hide the staging
// latency: measured on B2/256 conv1 the unpipelined kernel spends 10 of its 59 us staging with the MFMA pipe idle.
template <int NI, int ND>
__global__ __launch_bounds__(256) void wgrad3p_kernel(Wgrad3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, g = blockIdx.x;
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2;
  const int H2 = H >> 1, W2 = W >> 1, P2 = H2 * W2, RO = 2 * H2, CO = 2 * W2;
  const int RU = a.RU, U = a.U;
  const int IMGU = (RU + 2) * PW, DU = RU * CO, D = U * DU;
  const int UPS = (RO + RU - 1) / RU;
  const int NU = a.n * UPS;
  const int UPG = (NU + a.G - 1) / a.G;
  const int ubeg = g * UPG, uend = (ubeg + UPG < NU) ? ubeg + UPG : NU;
  const int IMGF = U * IMGU * 64, BUF = IMGF + D * 64;     // floats per stage buffer: [img | dz]
  STAMP(2, 0);

  {  // borders of both buffers must be zero; interiors are rewritten every stage
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* p4 = (float4*)smem;
    for (int i = tid; i < (2 * BUF) >> 2; i += 256) p4[i] = z;
  }
  const float* src = a.in + (long long)net * a.in_ns;
  const float* dp = a.dpool + (long long)net * a.dpool_ns;
  const uint8_t* mk = a.mask + (long long)net * a.dpool_ns;
  const int ct = wave & 1, it = wave >> 1;

  // stage items of this thread: the (unit, row, column, channel chunk) decode does not depend on the stage;
  // the (sample, unit-in-sample) pair of every item is carried from stage to stage (no divisions in the loop)
  const int RUh = RU >> 1;
  const int itot = U * (RU + 2) * W * 16, dtot = U * RUh * W2 * 16;
  const int qU = U / UPS, rU = U - qU * UPS;
  int i_u[NI], i_ir[NI], i_g[NI], i_l[NI], i_smp[NI], i_j[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int idx = tid + q * 256;
    const int id = idx < itot ? idx : 0;
    const int c4 = id & 15, p = id >> 4;
    const int u = p / ((RU + 2) * W), rem = p - u * (RU + 2) * W, ir = rem / W, w = rem - ir * W;
    i_u[q] = idx < itot ? u : -1; i_ir[q] = ir; i_g[q] = w * 64 + c4 * 4;
    i_l[q] = (u * IMGU + ir * PW + w + 1) * 64 + c4 * 4;
    i_smp[q] = (ubeg + u) / UPS; i_j[q] = (ubeg + u) - i_smp[q] * UPS;
  }
  int d_u[ND], d_rh[ND], d_g[ND], d_l[ND], d_smp[ND], d_j[ND];
#pragma unroll
  for (int q = 0; q < ND; ++q) {
    const int idx = tid + q * 256;
    const int id = idx < dtot ? idx : 0;
    const int c4 = id & 15, p = id >> 4;
    const int u = p / (RUh * W2), rem = p - u * RUh * W2, rh = rem / W2, pw = rem - rh * W2;
    d_u[q] = idx < dtot ? u : -1; d_rh[q] = rh; d_g[q] = pw * 64 + c4 * 4;
    d_l[q] = (u * DU + 2 * rh * CO + 2 * pw) * 64 + c4 * 4;
    d_smp[q] = (ubeg + u) / UPS; d_j[q] = (ubeg + u) - d_smp[q] * UPS;
  }
  struct DM { float4 d; uint32_t m; };
  float4 pi[NI];
  DM pd[ND];
  // issue(ub) must be called for ub = ubeg, ubeg + U, ... in order (it advances the carried indices)
  auto issue = [&](int ub) {
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int row = i_j[q] * RU - 1 + i_ir[q];
      const bool ok = (i_u[q] >= 0) && (ub + i_u[q] < uend) && row >= 0 && row < H;
      const float4 v = *(const float4*)(src + ((size_t)(ok ? i_smp[q] : ubeg / UPS) * HW + (ok ? row : 0) * W) * 64 + i_g[q]);
      pi[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      i_smp[q] += qU; i_j[q] += rU;
      if (i_j[q] >= UPS) { i_j[q] -= UPS; ++i_smp[q]; }
    }
#pragma unroll
    for (int q = 0; q < ND; ++q) {
      const int prow = d_j[q] * RUh + d_rh[q];
      const bool ok = (d_u[q] >= 0) && (ub + d_u[q] < uend) && prow < H2;
      const size_t gi = ((size_t)(ok ? d_smp[q] : ubeg / UPS) * P2 + (ok ? prow : 0) * W2) * 64 + d_g[q];
      pd[q].d = *(const float4*)(dp + gi);
      pd[q].m = ok ? *(const uint32_t*)(mk + gi) : 0u;
      d_smp[q] += qU; d_j[q] += rU;
      if (d_j[q] >= UPS) { d_j[q] -= UPS; ++d_smp[q]; }
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int q = 0; q < NI; ++q)
      if (i_u[q] >= 0) *(float4*)(buf + i_l[q]) = pi[q];
#pragma unroll
    for (int q = 0; q < ND; ++q) {
      if (d_u[q] >= 0) {
        float* dzb = buf + IMGF + d_l[q];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          float4 v;
          v.x = ((pd[q].m >> sub) & 1u) ? pd[q].d.x * 0.25f : 0.f;
          v.y = ((pd[q].m >> (8 + sub)) & 1u) ? pd[q].d.y * 0.25f : 0.f;
          v.z = ((pd[q].m >> (16 + sub)) & 1u) ? pd[q].d.z * 0.25f : 0.f;
          v.w = ((pd[q].m >> (24 + sub)) & 1u) ? pd[q].d.w * 0.25f : 0.f;
          *(float4*)(dzb + ((sub >> 1) * CO + (sub & 1)) * 64) = v;
        }
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) acc[s] = zero16();
  float dbacc = 0.f;

  if (ubeg < uend) issue(ubeg);
  __syncthreads();                 // zero fill complete
  if (ubeg < uend) commit(smem);
  __syncthreads();
  STAMP(2, 1);
  const int rows = U * RU, cpr = CO >> 1, pairs = rows * cpr;   // rows is even, so pairs is even
  // Operand addressing.  Pixel pair (row r, columns 2cp, 2cp+1) of unit u; lane half hh takes column 2cp+hh.
  // Tap (kh, kw) of that pixel sits at padded position ((r + kh) * PW + 2cp + hh + kw): three row pointers and
  // the immediate offsets 0 / 64 / 128 floats.  The running offset advances by 128 floats per pair, plus a row
  // jump at the end of a row and a unit jump (the two halo rows) at the end of a unit; dz is contiguous.
  const int rowjump = PW * 64 - cpr * 128, unitjump = 2 * PW * 64, rstride = PW * 64;
  int cur = 0;
  for (int ub = ubeg; ub < uend; ub += U) {
    const bool more = ub + U < uend;            // workgroup-uniform
    if (more) issue(ub + U);                    // global loads in flight across the MFMA loop below
    const float* buf = smem + cur * BUF;
    const float* a_base = buf + it * 32 + l31 + hh * 64;
    const float* b_base = buf + IMGF + ct * 32 + l31 + hh * 64;
    float a0[9], a1[9], b0, b1;                 // ping-pong operand sets: no register copies in the loop
    int aoff = 0, boff = 0, cp = 0, r = 0;
    // One region per MFMA, fenced by sched_barrier: the MFMA of the current pair, ONE LDS read of the next pair
    // and a slice of the address bookkeeping of the pair after that.  With a single wave per SIMD nothing else
    // keeps the MFMA pipe fed, and the wave issues in order: any block of non-MFMA instructions longer than one
    // MFMA (64 cycles) is a bubble.  Measured per pair: 1025 cycles with reads and bookkeeping in a block in
    // front of the MFMAs, 740 with only the reads interleaved, against 576 cycles of MFMA.
#define WG3_REGION(S, CA, CB, NA, PTR, OFF, EXTRA)                              \
    NA[S] = PTR[OFF];                                                           \
    acc[S] = mfma32(CA[S], CB, acc[S]);                                         \
    EXTRA;                                                                      \
    __builtin_amdgcn_sched_barrier(0);
    // P* = operand pointers of the pair loaded in this half, Q* = those of the following pair (formed here)
#define WG3_HALF(CA, CB, NA, NB, P0, P1, P2, PB, Q0, Q1, Q2, QB, TN)            \
    {                                                                           \
      WG3_REGION(0, CA, CB, NA, P0, 0,   (aoff += 128, boff += 128, ++cp))      \
      WG3_REGION(1, CA, CB, NA, P0, 64,  { if (cp == cpr) { cp = 0; aoff += rowjump; ++r; } }) \
      WG3_REGION(2, CA, CB, NA, P0, 128, { if (r == RU) { r = 0; aoff += unitjump; } })        \
      WG3_REGION(3, CA, CB, NA, P1, 0,   { if ((TN) >= pairs) { aoff = 0; boff = 0; } }) \
      WG3_REGION(4, CA, CB, NA, P1, 64,  Q0 = a_base + aoff)                    \
      WG3_REGION(5, CA, CB, NA, P1, 128, Q1 = Q0 + rstride)                     \
      WG3_REGION(6, CA, CB, NA, P2, 0,   Q2 = Q1 + rstride)                     \
      WG3_REGION(7, CA, CB, NA, P2, 64,  QB = b_base + boff)                    \
      NB = PB[0];                                                               \
      WG3_REGION(8, CA, CB, NA, P2, 128, dbacc += CB)                           \
    }
    const float *x0, *x1, *x2, *xb, *y0, *y1, *y2, *yb;
    {
      const float* p0 = a_base;
      const float* p1 = p0 + rstride;
      const float* p2 = p1 + rstride;
      a0[0] = p0[0]; a0[1] = p0[64]; a0[2] = p0[128];
      a0[3] = p1[0]; a0[4] = p1[64]; a0[5] = p1[128];
      a0[6] = p2[0]; a0[7] = p2[64]; a0[8] = p2[128];
      b0 = b_base[0];
      // pointers of pair 1
      aoff = 128; boff = 128; cp = 1;
      if (cp == cpr) { cp = 0; aoff += rowjump; ++r; }
      if (r == RU) { r = 0; aoff += unitjump; }
      x0 = a_base + aoff; x1 = x0 + rstride; x2 = x1 + rstride; xb = b_base + boff;
    }
    for (int t = 0; t < pairs; t += 2) {
      WG3_HALF(a0, b0, a1, b1, x0, x1, x2, xb, y0, y1, y2, yb, t + 2)   // MFMA pair t, load t+1, address t+2
      WG3_HALF(a1, b1, a0, b0, y0, y1, y2, yb, x0, x1, x2, xb, t + 3)   // MFMA pair t+1, load t+2, address t+3
    }
#undef WG3_REGION
#undef WG3_HALF
    if (more) commit(smem + (cur ^ 1) * BUF);
    __syncthreads();   // stage g fully read by every wave, stage g+1 fully written
    cur ^= 1;
  }
  STAMP(2, 2);

  float* part = a.part + (long long)net * a.part_ns + (size_t)g * PART3;
#pragma unroll
  for (int s = 0; s < 9; ++s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = it * 32 + acc_row(r, lane);
      part[s * 4096 + ci * 64 + ct * 32 + l31] = acc[s][r];
    }
  }
  if (it == 0) {
    const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
    if (hh == 0) part[9 * 4096 + ct * 32 + l31] = tot;
  }
  STAMP(2, 3);
}

static size_t wgrad3_lds(int RU, int U, int W, int cspl = 1) {
  const int PW = W + 2, CO = 2 * (W / 2);
  const size_t D = (size_t)U * RU * CO;
  return ((size_t)U * (RU + 2) * PW * 64 + D * (64 / cspl)) * 4;
}

bool plan_wgrad3(int n, int H, int W, Wgrad3Plan* p) {
  const int RO = 2 * (H / 2);
  if (RO <= 0) return false;
  // rows per unit: the largest even divisor-friendly RU that fits with U = 1
  int RU = 0;
  for (int cand = RO; cand >= 2; cand -= 2) {
    if (wgrad3_lds(cand, 1, W) > LDS_MAX) continue;
    if (RU == 0) RU = cand;                       // largest that fits
    if (RO % cand == 0) { RU = cand; break; }     // prefer an exact split
  }
  if (RU == 0) return false;
  const int UPS = (RO + RU - 1) / RU;
  const long long NU = (long long)n * UPS;
  static const int force_u = getenv("CMLPL_WGRAD3_U") ? atoi(getenv("CMLPL_WGRAD3_U")) : 0;
  int U = 1;
  while (U < 8 && wgrad3_lds(RU, U + 1, W) <= LDS_MAX && (NU + U) / (U + 1) >= 128) ++U;
  if (force_u > 0 && wgrad3_lds(RU, force_u, W) <= LDS_MAX) U = force_u;
  // Experiment (CMLPL_WGRAD3_CSPL=2): split the output channels over two workgroups and walk the units one at
  // a time so that 2-3 workgroups are co-resident per CU.  Measured on B2/256: 67.5 us vs 59.0 us for conv1 --
  // the image is staged twice and that costs more than the overlap buys -- so it is off by default.
  static const int force_c = getenv("CMLPL_WGRAD3_CSPL") ? atoi(getenv("CMLPL_WGRAD3_CSPL")) : 0;
  p->cspl = 1;
  if (force_c == 2 && wgrad3_lds(RU, 1, W, 2) <= LDS_MAX) {
    p->cspl = 2;
    U = 1;
  }
  // one pass per workgroup when that still fills the chip; never more workgroups than passes
  long long G = (NU + U - 1) / U;
  if (p->cspl == 2) { G = (NU + 1) / 2; if (G > 128) G = 128; }   // 2 co-halves x 2 nets x 128 = 512 workgroups
  if (G > 256) G = 256;                            // per net; 2 nets -> 512 WGs
  p->RU = RU; p->U = U; p->G = (int)G; p->lds = wgrad3_lds(RU, U, W, p->cspl);
  p->NI = 0;
  // pipelined kernel: units of 2 output rows, U of them per stage so that a stage carries >= 16 pixel pairs
  // (amortises the per-stage barrier) but needs at most 8 float4 of prefetch registers per thread
  static const int pipe = getenv("CMLPL_WGRAD3_PIPE") ? atoi(getenv("CMLPL_WGRAD3_PIPE")) : 1;
  static const int force_pg = getenv("CMLPL_WGRAD3_PG") ? atoi(getenv("CMLPL_WGRAD3_PG")) : 0;
  static const int force_pu = getenv("CMLPL_WGRAD3_PU") ? atoi(getenv("CMLPL_WGRAD3_PU")) : 0;
  const int CO = 2 * (W / 2);
  if (pipe && p->cspl == 1 && CO >= 2) {
    auto ni_of = [&](int ru, int u) { return (u * (ru + 2) * W * 16 + 255) / 256; };
    auto nd_of = [&](int ru, int u) { return (u * (ru / 2) * (W / 2) * 16 + 255) / 256; };
    auto fits = [&](int ru, int u) {
      return 2 * wgrad3_lds(ru, u, W) <= LDS_MAX && ni_of(ru, u) <= 16 && nd_of(ru, u) <= 2;
    };
    // rows per unit: the largest even divisor of the output rows whose double buffer fits (a whole sample for
    // 11x11 windows: two stages per workgroup instead of five)
    int RUp = 0;
    for (int cand = RO; cand >= 2; cand -= 2)
      if (RO % cand == 0 && fits(cand, 1)) { RUp = cand; break; }
    if (RUp > 0) {
      const int ppu = RUp * CO / 2;                 // pixel pairs per unit
      int Up = 1;
      while (Up * ppu < 16 && fits(RUp, Up + 1)) ++Up;
      if (force_pu > 0 && fits(RUp, force_pu)) Up = force_pu;
      // one workgroup per CU across both networks; whole samples per workgroup
      long long Gp = n < 128 ? n : 128;
      if (force_pg > 0) Gp = force_pg < n ? force_pg : n;
      const long long spg = (n + Gp - 1) / Gp;     // samples per workgroup
      Gp = (n + spg - 1) / spg;
      const int NI = ni_of(RUp, Up);
      p->RU = RUp; p->U = Up; p->G = (int)Gp; p->lds = 2 * wgrad3_lds(RUp, Up, W);
      p->NI = NI <= 4 ? 4 : NI <= 6 ? 6 : NI <= 8 ? 8 : NI <= 10 ? 10 : NI <= 12 ? 12 : 16;
      p->ND = nd_of(RUp, Up);
    }
  }
  return true;
}

hipError_t launch_wgrad3(int nets, int n, int H, int W, const float* in, const float* dpool, const uint8_t* mask,
                         float* part, hipStream_t st) {
  Wgrad3Plan pl;
  if (!plan_wgrad3(n, H, W, &pl)) return hipErrorInvalidValue;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_MAX);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)wgrad3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  Wgrad3Args a;
  a.in = in; a.dpool = dpool; a.mask = mask; a.part = part;
  a.in_ns = (long long)n * H * W * 64;
  a.dpool_ns = (long long)n * (H / 2) * (W / 2) * 64;
  a.part_ns = (long long)pl.G * PART3;
  a.n = n; a.H = H; a.W = W; a.RU = pl.RU; a.U = pl.U; a.G = pl.G;
  if (pl.NI > 0) {
#define WG3P_CASE(NI_, ND_)                                                                          \
    if (pl.NI == NI_ && pl.ND == ND_) {                                                              \
      static bool attr_p = false;                                                                    \
      if (!attr_p) {                                                                                 \
        hipError_t e = hipFuncSetAttribute((const void*)wgrad3p_kernel<NI_, ND_>,                    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); \
        if (e != hipSuccess) return e;                                                               \
        attr_p = true;                                                                               \
      }                                                                                              \
      hipLaunchKernelGGL((wgrad3p_kernel<NI_, ND_>), dim3(pl.G, nets), dim3(256), pl.lds, st, a);    \
      return hipGetLastError();                                                                      \
    }
    WG3P_CASE(4, 1) WG3P_CASE(6, 1) WG3P_CASE(8, 1) WG3P_CASE(10, 1) WG3P_CASE(12, 1) WG3P_CASE(16, 1)
    WG3P_CASE(4, 2) WG3P_CASE(6, 2) WG3P_CASE(8, 2) WG3P_CASE(10, 2) WG3P_CASE(12, 2) WG3P_CASE(16, 2)
#undef WG3P_CASE
    return hipErrorInvalidValue;
  } else if (pl.cspl == 2) hipLaunchKernelGGL(wgrad3_kernel<2>, dim3(pl.G, nets, 2), dim3(256), pl.lds, st, a);
  else              hipLaunchKernelGGL(wgrad3_kernel<1>, dim3(pl.G, nets), dim3(256), pl.lds, st, a);
  return hipGetLastError();
}

int wgrad3_G(int n, int H, int W) {
  Wgrad3Plan pl;
  return plan_wgrad3(n, H, W, &pl) ? pl.G : 0;
}

}  // namespace cmlpl
