// 3x3 / pad 1 / 64->64 convolutions of BaseNet2 (tools/models.py:104-107,134-140) and the per-sample fused
// forward / backward kernels built around them, on the bf16 MFMA with fp32 operands taken as three exact bf16 pieces
// ("fp32 on the bf16 MFMA", below: v_mfma_f32_32x32x16_bf16, six products per fp32 product, fp32 accumulate).
//
//   conv3x3_kernel<0>     : z = conv(in)+bias+in ; r = relu(z) ; out = avgpool2(r) ; mask = (r>0) nibble
//                           (models.py:134-136 / 138-140, fused), S samples per workgroup, tap weights staged in LDS
//   conv3x3_kernel<1>     : dz = mask * upsample(dpool)/4 (avgpool+relu backward, formed while staging)
//                           out = conv_transpose(dz) + dz      (residual branch adds dz itself)
//   conv3x3_kernel<2,1,1> : the WHOLE forward of one sample in one workgroup: input slab in 16-band chunks global ->
//                           registers -> + noise -> LDS, conv0 1x1 per chunk as it lands, conv1 (barrier-free tap
//                           loop, weight fragments L2 -> registers), ReLU / pool, conv2 + pool + concat / dropout /
//                           classifier / L2-norm (conv3_fwd_tail)
//   conv3x3_kernel<3,1,1> : the whole data-gradient chain of one sample: head backward + conv2 data gradient
//                           (conv3_bwd_head), conv1 data gradient, conv0 weight-gradient partial (band passes)
//   conv3x3_small_kernel  : maps of at most 32 pixels (one M tile shared by the four waves)
//   (the 3x3 weight gradients live in wgrad3x3.hip)
//
// Data layout: activations are pixel-major / channel-last  [net][sample][pixel][64] so that the 64 channels of a
// pixel are one 256-B line; the padded image of S samples sits in LDS with a zero border (pixel stride 68 floats),
// so the 9 taps are pure address offsets.  Weights are kept re-packed by the optimizer (kernels.hpp, PACK_*) as
// ready-made split-bf16 B fragments [tap][k16][piece][n tile][lane][8 bf16]: a fragment is one 16-byte read.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// Ablation / timeline builds (scripts/conv_timeline.py): -DCMLPL_ABL=n removes one ingredient
// of a kernel (results are then wrong on purpose) or adds per-workgroup phase stamps (9).  0 = the product.
#ifndef CMLPL_ABL
#define CMLPL_ABL 0
#endif
// Tap loops stay rolled (UNR = 1).  Fully unrolling them was tried for the fused kernels: -1.4 us for the data
// gradient in one build, but the register allocation of the unrolled loop is fragile (another build of the same
// source went to 512 registers + scratch and 27 us slower); forming the slab's augmentation noise inside the
// unrolled data-gradient loop, a piece per tap, bought nothing either: with the cheap generator the phase behind
// the loop is bound by the slab DMA, which cannot start before the loop ends (the slab aliases the image).
#define CMLPL_UNR_F 1

#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

constexpr int CS = 68;  // LDS pixel stride in floats (64 + 4): conflict-free ds_read_b128 over 16 pixels

// ------------------------------------------------------------------------------------------
// weight packing: canonical W[co][ci][kh][kw] -> the fragment sets described in kernels.hpp (PACK_*)
// ------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ params, long long pstride, PackInfo pi,
                                    float* __restrict__ packed) {
  const int net = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= pi.stride) return;
  const float* P = params + (long long)net * pstride;
  float v;
  if (e < 4LL * PACK_B3) {                           // split-bf16 fragment sets: two bf16 per float slot
    const int which = (int)(e / PACK_B3);            // 0 c1 fwd, 1 c1 dgrad, 2 c2 fwd, 3 c2 dgrad
    const int i = (int)(e - (long long)which * PACK_B3) * 2;     // bf16 index (inverse of conv_b3_index), j even
    const int j = i & 7, l31 = (i >> 3) & 31, h = (i >> 8) & 1, nt = (i >> 9) & 1, rest = i >> 10;
    const int pc = rest % 3, kq = (rest / 3) & 3, tap = rest / 12;
    const int kh = tap / 3, kw = tap - kh * 3, n = nt * 32 + l31;
    const float* W = P + ((which < 2) ? pi.off_w1 : pi.off_w2);
    uint32_t out = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = kq * 16 + h * 8 + j + d;
      const float w = ((which & 1) == 0) ? W[((n * 64 + k) * 3 + kh) * 3 + kw]                 // co=n, ci=k
                                         : W[((k * 64 + n) * 3 + (2 - kh)) * 3 + (2 - kw)];     // co=k, ci=n
      uint32_t pcs[3];
      b3_split(w, pcs);
      out |= pcs[pc] << (16 * d);
    }
    v = __uint_as_float(out);
  } else if (e < PACK_PER_NET) {
    v = 0.f;                                           // (former fp32 16x16x4 fragments of conv2: unused region)
  } else if (e < pack_off_wst(pi.C)) {
    const int i = (int)(e - pack_off_w0t()), c = i >> 6, co = i & 63;
    v = (c < pi.C) ? P[pi.off_w0 + (long long)co * pi.C + c] : 0.f;
  } else if (e < pack_off_w0b3(pi.C, pi.bands)) {
    v = 0.f;                                           // (former k-major copy of feat_spe.weight: unused region)
  } else if (e >= pack_off_h2flag(pi.C, pi.bands)) {
    return;                                            // the flag words: zeroed by the launcher, set below
  } else if (e >= pack_off_h2(pi.C, pi.bands, 0)) {    // conv1 as two fp16 pieces (kernels.hpp: pack_off_h2): two fp16 per float slot
    const long long r = e - pack_off_h2(pi.C, pi.bands, 0);
    const int which = (int)(r / PACK_H2);              // 0 / 2: conv1 / conv2 forward (n = co, k = ci), 1 / 3: data gradient (n = ci, k = co, taps flipped)
    const int i = (int)(r - (long long)which * PACK_H2) * 2;
    const int j = i & 7, l31 = (i >> 3) & 31, h = (i >> 8) & 1, nt = (i >> 9) & 1, rest = i >> 10;
    const int pc = rest & 1, kq = (rest >> 1) & 3, tap = rest >> 3;
    const int kh = tap / 3, kw = tap - kh * 3, n = nt * 32 + l31;
    const float* W = P + ((which < 2) ? pi.off_w1 : pi.off_w2);
    uint32_t out = 0;
    bool bad = false;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = kq * 16 + h * 8 + j + d;
      const float w = ((which & 1) == 0) ? W[((n * 64 + k) * 3 + kh) * 3 + kw] : W[((k * 64 + n) * 3 + (2 - kh)) * 3 + (2 - kw)];
      uint32_t pcs[2];
      bad |= h2_split_w(w, pcs);
      out |= pcs[pc] << (16 * d);
    }
    if (bad) atomicOr((unsigned int*)(packed + (long long)net * pi.stride + pack_off_h2flag(pi.C, pi.bands)), 1u);
    v = __uint_as_float(out);
  } else {                                           // conv0 as split-bf16 fragments: conv_b3_index(0, k = band, n = co, piece)
    const int i = (int)(e - pack_off_w0b3(pi.C, pi.bands)) * 2;
    const int j = i & 7, l31 = (i >> 3) & 31, h = (i >> 8) & 1, nt = (i >> 9) & 1, rest = i >> 10;
    const int pc = rest % 3, kq = rest / 3, n = nt * 32 + l31;
    uint32_t out = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = kq * 16 + h * 8 + j + d;
      uint32_t pcs[3];
      b3_split(k < pi.C ? P[pi.off_w0 + (long long)n * pi.C + k] : 0.f, pcs);
      out |= pcs[pc] << (16 * d);
    }
    v = __uint_as_float(out);
  }
  packed[(long long)net * pi.stride + e] = v;
}

hipError_t launch_pack_weights(int nets, const float* params, long long pstride, const PackInfo& pi, float* packed,
                               hipStream_t st) {
  dim3 grid((unsigned)((pi.stride + 255) / 256), nets);
  for (int net = 0; net < nets; ++net) {               // the two-piece fp16 sets' range flag: cleared by a FULL pack only
    hipError_t e = hipMemsetAsync(packed + (long long)net * pi.stride + pack_off_h2flag(pi.C, pi.bands), 0, 64, st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, st, params, pstride, pi, packed);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------
#if CMLPL_ABL == 9 || (CMLPL_ABL >= 20 && CMLPL_ABL != 26)
// phase timeline instrumentation (ablation build only): constant-rate 100 MHz stamps per workgroup
__device__ unsigned long long g_stamps[3][2048][16];
#ifndef CMLPL_STAMP_MIN_H             // launches on maps of fewer rows leave no stamps (conv1 of the general path is followed
#define CMLPL_STAMP_MIN_H 0           // by conv2 on the pooled map): ABL_FLAGS=-DCMLPL_STAMP_MIN_H=20 bash scripts/build_abl.sh 9
#endif
#define STAMP(MODE_, i) do { if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 2048) \
    g_stamps[MODE_][blockIdx.x + gridDim.x * blockIdx.y][i] = wall_clock64(); } while (0)
#define STAMPG(MODE_, i) do { if (MODE >= 2 || CMLPL_STAMP_MIN_H == 0 || a.H >= CMLPL_STAMP_MIN_H) STAMP(MODE_, i); } while (0)
extern "C" int cmlpl_abl_read_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#else
#define STAMP(MODE_, i) do {} while (0)
#define STAMPG(MODE_, i) do {} while (0)
#endif

struct Conv3Args {
  const float* in; const uint8_t* mask_in; const float* wpk; const float* bias;
  float* out; uint8_t* mask_out;
  long long in_ns, mask_in_ns, wpk_ns, bias_ns, out_ns, mask_out_ns;  // per-net strides (elements)
  int n, H, W, S;
  // MODE 2 (conv0 fused into the conv1 forward): the input patches come from `xs` (raw rows + noise on the fly);
  // a0 is produced here
  const float* w0t; const float* b0; float* a0out;
  long long w0t_ns, b0_ns;
  int C;
  float* xn_out;                                  // MODE 2, optional: the augmented rows [net][n][C*HW], kept for the backward pass
  // MODE 3 (conv0 weight gradient fused into the conv1 data gradient): da0 never leaves the workgroup; the input
  // slab is re-formed from `xs` (same noise as the forward: counter-based)
  float* part0; long long part0_ns;
  int bp;                                         // bands per pass of the fused conv0 weight gradient (>= C: one pass)
  XSrc xs;
  // TAIL (forward): conv2 + ReLU + avgpool + flatten/concat + dropout + classifier + L2-norm for the same sample
  // (tools/models.py:137-152), in the workgroup that has just pooled conv1's output
  const float* w2f; long long w2f_ns;            // conv2 forward split-bf16 fragment set (kernels.hpp: pack_off_b3(.., 2))
  const float* b2; const float* wc; const float* bc; long long p_ns;   // conv2.bias, classifier.weight / .bias
  const float* yin;                               // spectral branch output relu(feat_spe(x)) [net][n][1024]
  const float* dropmask; float* dropgen; float* catd; float* ynorm; float* logits; float* feat;
  float* p2out; uint8_t* m2out;
  float dropout_p; int train, K;
  // HEAD (backward, MODE 3 + TAIL): classifier / L2-norm / spectral-ReLU backward and the conv2 data gradient of
  // the same sample run first; their output dp1 feeds the conv1 data gradient through LDS
  const float* dlogits; const float* dfeat; const float* hmask;   // hmask: dropout multiplier rows or null
  const float* ynrm; const uint8_t* m2in; const float* w2d; long long w2d_ns;
  float* dy; float* dp2out; float* dp1out;
  // INFER (forward, TAIL == 2): whole-image inference straight from the scene cube (tools/hyper_tools.py:226-243,416-437):
  // sample s is pixel pix0 + s (row-major) of the band-last cube [crows][ccols][C]; its H x W window is gathered through
  // the mirror index of ExtractPatches while the slab chunks are staged -- no patch tensor exists; the argmax goes to
  // labels_out[s] (logits optional)
  const float* cube; int crows, ccols; long long pix0; long long* labels_out;
  // x / d for the per-item index arithmetic of the staging and pooling loops, as one multiply (fdiv below): d = H W, W,
  // (H / 2)(W / 2), W / 2 -- a 32-bit division by a run-time value is ~40 vector instructions, and those loops did two to
  // four of them per 16-byte item (set by conv3_set_magics on the host)
  uint32_t mg_hw, mg_w, mg_p2, mg_w2;
  // H2X kernels (conv1's tap loop on two fp16 pieces): this launch's two-piece weight set, the networks' range flags, and
  // the LDS word (float index into the dynamic allocation) in which the staging leaves the image's largest magnitude
  const float* wpk16; long long wpk16_ns; const uint32_t* h2flag; long long h2flag_ns; int maxslot;
  int h2_noskip;      // measurement aid (CMLPL_F16X2=4): an all-zero image runs the two-piece loop instead of skipping it
  int pairshift;      // see wg_decode
  int hkind;          // general kernels (MODE 0 / 1): which statistic this launch's image is (0 a0, 1 p1, 2 / 3 conv1's / conv2's gradient operand)
  // statistics for the two-piece WEIGHT-GRADIENT kernel (wgrad3x3.hip), [kind][2 networks][n samples] words: every
  // workgroup leaves its sample's largest magnitude (float bits) of a0, p1 (forward) and of conv1's / conv2's masked
  // up-sampled pooled gradients (backward) -- plain stores, every slot rewritten every step; null = not collected
  uint32_t* hstat;
};

// x / d == umulhi(x, ceil(2^32 / d)) for x d < 2^32 (every index here is below 2^16); d = 1 is flagged by 0
static inline uint32_t fdiv_magic(int d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ULL + (uint32_t)d - 1) / (uint32_t)d); }
__device__ __forceinline__ int fdiv(int x, uint32_t mg) { return mg == 0u ? x : (int)__umulhi((uint32_t)x, mg); }
static inline void conv3_set_magics(Conv3Args& a) {
  const int H2 = a.H >> 1, W2 = a.W >> 1;
  a.mg_hw = fdiv_magic(a.H * a.W); a.mg_w = fdiv_magic(a.W); a.mg_p2 = fdiv_magic(H2 * W2); a.mg_w2 = fdiv_magic(W2);
}

// Workgroup -> (network, first sample).  (Measured, round 3: numbering the workgroups so that the two a CU holds belong
// to the same network -- hoping their weight-fragment streams would meet in L1 -- changed nothing: 0.2129 vs 0.2128 ms.)
// pairshift (fused backward, S == 1): network 1's workgroup b takes sample (b + n / 2) % n, so that the two workgroups a CU
// holds (b of network 0 and b of network 1, by launch order) are a labelled and an unlabelled row where the batch is
// [labelled ; unlabelled] halves -- an unlabelled row under the confidence threshold has an all-zero gradient image and
// skips most of its work: paired with a working row instead of with its own twin it leaves that row the CU.
__device__ __forceinline__ void wg_decode(const Conv3Args& a, int& net, int& s0) {
  net = (int)blockIdx.y; s0 = (int)blockIdx.x * a.S;
  if (a.pairshift != 0 && net == 1) { s0 += a.n >> 1; if (s0 >= a.n) s0 -= a.n; }
}
// Inference from the cube: workgroup b takes pixel (b % 8) * ceil(n / 8) + b / 8 of the launch's range, so that each XCD
// (own L2, workgroups dealt round-robin) walks one contiguous eighth in raster order and overlapping windows are re-read
// from ITS L2 (what made raster-ordered patch extraction 25-30 % faster, augment.hip); -1: past the range.
__device__ __forceinline__ int wg_infer_sample(const Conv3Args& a) {
  const int b = (int)blockIdx.x, per = (a.n + 7) >> 3;
  const int s = (b & 7) * per + (b >> 3);
  return ((b >> 3) < per && s < a.n) ? s : -1;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: A lane l holds A[i = l&15][k = l>>4]; B lane l holds B[k = l>>4][j = l&15];
// D reg r of lane l is D[row = 4*(l>>4) + r][col = l&15].  Exact fp32, 32 cycles.
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Fused data gradient (MODE 3): the sample's [C][HW] rows AS THE FORWARD SAW THEM -- the augmented rows the fused
// forward left in the workspace (cmlpl_forward: xn_save), or a caller's pre-augmented tensor -- go to LDS as a LINEAR
// copy by global_load_lds_dwordx4 (1 KiB per wave-instruction, every piece in flight at once).  A RANGE of the slab --
// `nfl` floats from float4 offset `f4base` of the sample's block, the whole block unless the bands take several
// passes -- lands at `slab`; the caller's next __syncthreads() publishes it (the barrier waits for the DMA).
// (Rounds 2-3 could also re-form the forward's noise here from the raw rows; since the forward stores the augmented
// rows that path was dead, and it is gone: this kernel reads plain rows only, by batch row -- no index lists.)
constexpr int SLAB_MAXQ = 16;   // pieces per wave: range <= 4 waves * 16 * 256 floats
constexpr int SLAB_RING = 7;    // forward: chunks (16 bands each) resident in LDS per pass
constexpr int SLAB_WIN = 4;     // forward: chunks in flight global -> registers per wave
constexpr int SLAB_NUP = 7;     // forward: chunks whose noise is formed before the chunk loop starts = SLAB_RING, all of them
                                // (measured: forming the later chunks' noise inside the loop, beside the MFMAs, took 0.8 us longer)
static_assert(SLAB_NUP == SLAB_RING, "the forward forms the noise of every chunk of a pass up front");
typedef __attribute__((address_space(3))) void slab_lds_void;
typedef __attribute__((address_space(1))) const void slab_gbl_void;
struct SlabRange { const float* xs; int nfl; };
__device__ __forceinline__ SlabRange slab_range(const XSrc& x, int net, int s, int per, int f4base, int nfl) {
  SlabRange r;
  r.xs = (s < x.nlab ? x.lab[net] + (long long)s * per : x.unl[net] + (long long)(s - x.nlab) * per) + 4LL * f4base;
  r.nfl = nfl;
  return r;
}
template <int NW = 4>
__device__ __forceinline__ void slab_issue(const SlabRange& r, float* slab, int wave, int lane) {
  const int nf4 = r.nfl >> 2;
#pragma unroll
  for (int k = 0; k < SLAB_MAXQ * 4 / NW; ++k) {
    const int q = wave + NW * k;                          // wave-uniform
    if (q * 64 < nf4) {
      const int f = q * 64 + lane;
      if (f < nf4) __builtin_amdgcn_global_load_lds((slab_gbl_void*)(r.xs + 4 * f), (slab_lds_void*)(slab + q * 256), 16, 0, 0);
    }
  }
}
// the last, partial float4 group of the range (C * HW need not be a multiple of 4)
__device__ __forceinline__ void slab_tail(const SlabRange& r, float* slab, int tid) {
  const int nf4 = r.nfl >> 2, rem = r.nfl - 4 * nf4;
  if (tid < rem) slab[4 * nf4 + tid] = r.xs[4 * nf4 + tid];
}

// ---- "fp32 on the bf16 MFMA" ------------------------------------------------------------------------------------
// gfx950 runs the f32-input MFMA at 1/16 of the bf16 rate.  The 3x3 convolutions therefore take every fp32 operand
// as THREE bf16 pieces, x = x1 + x2 + x3 exactly (b3_split: successive truncations, 8 significant bits each), and
// form a . b as the six products whose weight is >= 2^-16,
//      a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Each bf16 x bf16 product is exact in fp32; what is dropped
// (a2 b3 + a3 b2 + a3 b3, with |x2| < 2^-7 |x|, |x3| < 2^-15 |x|) is below 2^-21 |a b| in the worst case and
// 0.7 * 2^-24 |a b| on average -- less than the rounding of one fp32 multiply (tests/test_split_bf16_math.py) -- and
// the accumulator is rounded 6 times per 16 k (once per MFMA) where the f32-input MFMA rounds it 8 times.  Six bf16
// MFMAs of 32 cycles replace eight f32 MFMAs of 64: 2.7x fewer matrix-pipe cycles for the same fp32-grade result
// (tests/test_gpu_ops.py measures the error of both against an fp64 reference).
// The weights are split once per step (Adam writes the fragment sets, kernels.hpp: conv_b3_index); activations are
// split in registers right after their ds_read_b128 (the LDS image stays fp32: a split image would not leave room
// for two workgroups per CU).
constexpr int WBUF = 6144;      // floats: one tap's weight fragments (4 k-steps x 3 pieces x 2 n tiles x 1 KiB)
constexpr int TAPW = WBUF / 4;  // float4 per tap
// a thread's share of one tap's weights on their way global -> LDS (named members: an array member ends up in scratch)
struct TapRegs { float4 w0, w1, w2, w3, w4, w5; };
__device__ __forceinline__ TapRegs tap_fetch(const float4* wg, int tap, int tid) {
  TapRegs t;
  const float4* wn = wg + tap * TAPW + tid;
  t.w0 = wn[0]; t.w1 = wn[256]; t.w2 = wn[512]; t.w3 = wn[768]; t.w4 = wn[1024]; t.w5 = wn[1280];
  return t;
}
__device__ __forceinline__ void tap_put(float4* wl, const TapRegs& t, int tid) {
  wl[tid] = t.w0; wl[tid + 256] = t.w1; wl[tid + 512] = t.w2; wl[tid + 768] = t.w3; wl[tid + 1024] = t.w4;
  wl[tid + 1280] = t.w5;
}

// the same for 512-thread workgroups: three 16-byte pieces per thread
struct TapRegs8 { float4 w0, w1, w2; };
__device__ __forceinline__ TapRegs8 tap_fetch8(const float4* wg, int tap, int tid) {
  TapRegs8 t;
  const float4* wn = wg + tap * TAPW + tid;
  t.w0 = wn[0]; t.w1 = wn[512]; t.w2 = wn[1024];
  return t;
}
__device__ __forceinline__ void tap_put8(float4* wl, const TapRegs8& t, int tid) {
  wl[tid] = t.w0; wl[tid + 512] = t.w1; wl[tid + 1024] = t.w2;
}

// The 9-tap main loop: 36 k-steps of 16 input channels.  NTA = number of this wave's M tiles that carry real pixels
// (wave-uniform, so the loop body is branch-free).  Tap weights go global -> registers (prefetched one tap ahead)
// -> LDS.  The steps are software-pipelined across the taps:
//   step k issues the LDS reads of step k+1's weights (inside a tap) and of step k+2's raw activations, runs its
//   12 MFMAs per tile, and between them splits step k+1's raw activations (landed during step k-1) into bf16 pieces;
//   sched_group_barrier pins the interleave (4 MFMAs, then 1 MFMA : 6 VALU) -- left alone the scheduler puts each
//   step's 44 split instructions in front of its MFMAs and the matrix pipe idles meanwhile.
// Only a tap's first weight read is exposed (it cannot be issued before the tap's barrier).
struct NoSide { __device__ __forceinline__ void operator()(int) const {} };
struct ASplit { uint4 p1, p2, p3; };

// LDS float offset of k-step `step` (tap = step / 4, 16 channels (step & 3) * 16) relative to the output pixel
__device__ __forceinline__ int tap_step_off(int step, int PW) {
  const int s = step >> 2, kh = s / 3, kw = s - kh * 3;
  return ((kh - 1) * PW + (kw - 1)) * CS + (step & 3) * 16;
}

// one k-step (KQ = its position inside the tap) for NTA tiles

// HALF (several tiles per wave only): 1 / 2 = the wave's LAST active tile is computed for output-channel tile 0 / 1 only
// (its other half belongs to another wave: see the eight-wave general kernels' tile assignment in conv3x3_kernel)
template <int NTA, int KQ, int MTW, int HALF = 0>
__device__ __forceinline__ void tap_step(const float* __restrict__ img, const uint4* __restrict__ bl,
                                         const int (&abase)[MTW], f32x16 (&acc)[MTW][2], ASplit (&cur)[NTA],
                                         float4 (&rn0)[NTA], float4 (&rn1)[NTA], uint4 (&b)[6], int step, int PW) {
  uint4 nb[6];
  float4 rnn0[NTA], rnn1[NTA];
#pragma unroll
  for (int i = 0; i < 6; ++i) nb[i] = b[i];
  if (KQ < 3) {
#pragma unroll
    for (int i = 0; i < 6; ++i) nb[i] = bl[((KQ + 1) * 6 + i) * 64];
  }
  {
    const int o2 = tap_step_off(step + 2 < 36 ? step + 2 : 35, PW);
#pragma unroll
    for (int t = 0; t < NTA; ++t) {
      rnn0[t] = *(const float4*)(img + abase[t] + o2);
      rnn1[t] = *(const float4*)(img + abase[t] + o2 + 4);
    }
  }
  ASplit nxt[NTA];
  if constexpr (NTA == 1) {
    acc[0][0] = mfma_b3(cur[0].p1, cur[0].p2, cur[0].p3, b[0], b[2], b[4], acc[0][0]);
    acc[0][1] = mfma_b3(cur[0].p1, cur[0].p2, cur[0].p3, b[1], b[3], b[5], acc[0][1]);
    a_split(rn0[0], rn1[0], nxt[0].p1, nxt[0].p2, nxt[0].p3);
  } else {
    // Several tiles per wave (the general path: 20 x 20 and 15 x 15 windows, ONE wave per SIMD): the interleave is
    // written out by hand -- one MFMA, then one element of the next step's split (4 vector instructions) or three of
    // its pack instructions, with plain scheduling fences between them.  (Left to the compiler, a tile's 44 split
    // instructions sit in front of its twelve MFMAs and the matrix pipe idles meanwhile: 110 us for conv1's forward at
    // 20 x 20 against a matrix-pipe floor of 52; sched_group_barrier patterns, what the one-tile waves use, take
    // minutes to compile at this many groups.)  Same MFMAs in the same order per accumulator: bit-identical results.
#define CMLPL_SPLIT1(j) { u0[j] = __float_as_uint(v[j]); const float r1_ = v[j] - __uint_as_float(u0[j] & 0xffff0000u); \
                          u1[j] = __float_as_uint(r1_); u2[j] = __float_as_uint(r1_ - __uint_as_float(u1[j] & 0xffff0000u)); }
#define CMLPL_FENCE __builtin_amdgcn_sched_barrier(0);
    constexpr int NFULL = NTA - (HALF != 0 ? 1 : 0);
#pragma unroll
    for (int t = 0; t < NFULL; ++t) {
      const uint4 A1 = cur[t].p1, A2 = cur[t].p2, A3 = cur[t].p3;
      const float v[8] = {rn0[t].x, rn0[t].y, rn0[t].z, rn0[t].w, rn1[t].x, rn1[t].y, rn1[t].z, rn1[t].w};
      uint32_t u0[8], u1[8], u2[8];
      f32x16 c0 = acc[t][0], c1 = acc[t][1];
      CMLPL_FENCE
      c0 = mfma_b16(A1, b[4], c0); CMLPL_FENCE CMLPL_SPLIT1(0) CMLPL_FENCE
      c0 = mfma_b16(A2, b[2], c0); CMLPL_FENCE CMLPL_SPLIT1(1) CMLPL_FENCE
      c0 = mfma_b16(A3, b[0], c0); CMLPL_FENCE CMLPL_SPLIT1(2) CMLPL_FENCE
      c0 = mfma_b16(A1, b[2], c0); CMLPL_FENCE CMLPL_SPLIT1(3) CMLPL_FENCE
      c0 = mfma_b16(A2, b[0], c0); CMLPL_FENCE CMLPL_SPLIT1(4) CMLPL_FENCE
      c0 = mfma_b16(A1, b[0], c0); CMLPL_FENCE CMLPL_SPLIT1(5) CMLPL_FENCE
      c1 = mfma_b16(A1, b[5], c1); CMLPL_FENCE CMLPL_SPLIT1(6) CMLPL_FENCE
      c1 = mfma_b16(A2, b[3], c1); CMLPL_FENCE CMLPL_SPLIT1(7) CMLPL_FENCE
      c1 = mfma_b16(A3, b[1], c1); CMLPL_FENCE
      nxt[t].p1 = make_uint4(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]), hi_pair(u0[4], u0[5]), hi_pair(u0[6], u0[7]));
      CMLPL_FENCE
      c1 = mfma_b16(A1, b[3], c1); CMLPL_FENCE
      nxt[t].p2 = make_uint4(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]), hi_pair(u1[4], u1[5]), hi_pair(u1[6], u1[7]));
      CMLPL_FENCE
      c1 = mfma_b16(A2, b[1], c1); CMLPL_FENCE
      nxt[t].p3 = make_uint4(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]), hi_pair(u2[4], u2[5]), hi_pair(u2[6], u2[7]));
      CMLPL_FENCE
      c1 = mfma_b16(A1, b[1], c1); CMLPL_FENCE
      acc[t][0] = c0; acc[t][1] = c1;
    }
    if constexpr (HALF != 0) {   // the shared tile: six MFMAs of ONE channel tile, in the order a full tile runs them
      constexpr int t = NTA - 1, h = HALF == 2 ? 1 : 0;
      const uint4 A1 = cur[t].p1, A2 = cur[t].p2, A3 = cur[t].p3;
      const float v[8] = {rn0[t].x, rn0[t].y, rn0[t].z, rn0[t].w, rn1[t].x, rn1[t].y, rn1[t].z, rn1[t].w};
      uint32_t u0[8], u1[8], u2[8];
      f32x16 ch = acc[t][h];
      CMLPL_FENCE
      ch = mfma_b16(A1, b[4 + h], ch); CMLPL_FENCE CMLPL_SPLIT1(0) CMLPL_SPLIT1(1) CMLPL_FENCE
      ch = mfma_b16(A2, b[2 + h], ch); CMLPL_FENCE CMLPL_SPLIT1(2) CMLPL_SPLIT1(3) CMLPL_FENCE
      ch = mfma_b16(A3, b[0 + h], ch); CMLPL_FENCE CMLPL_SPLIT1(4) CMLPL_SPLIT1(5) CMLPL_FENCE
      ch = mfma_b16(A1, b[2 + h], ch); CMLPL_FENCE CMLPL_SPLIT1(6) CMLPL_SPLIT1(7) CMLPL_FENCE
      ch = mfma_b16(A2, b[0 + h], ch); CMLPL_FENCE
      nxt[t].p1 = make_uint4(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]), hi_pair(u0[4], u0[5]), hi_pair(u0[6], u0[7]));
      nxt[t].p2 = make_uint4(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]), hi_pair(u1[4], u1[5]), hi_pair(u1[6], u1[7]));
      CMLPL_FENCE
      ch = mfma_b16(A1, b[0 + h], ch); CMLPL_FENCE
      nxt[t].p3 = make_uint4(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]), hi_pair(u2[4], u2[5]), hi_pair(u2[6], u2[7]));
      CMLPL_FENCE
      acc[t][h] = ch;
    }
#undef CMLPL_SPLIT1
#undef CMLPL_FENCE
  }
  // pin the interleave: the reads first, four MFMAs while they (and nothing else) are outstanding, then one MFMA
  // per six split instructions
  // (one tile per wave only: with more tiles the compiler's own order is kept -- the group solver's compile time grows
  // steeply with the number of groups, and those kernels are not the headline path)
  if (NTA == 1) {
    __builtin_amdgcn_sched_group_barrier(0x100, (KQ < 3 ? 6 : 0) + 2 * NTA, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    SchedInterleave<NTA == 1 ? 8 : 0>::run();
  }
#pragma unroll
  for (int t = 0; t < NTA; ++t) { cur[t] = nxt[t]; rn0[t] = rnn0[t]; rn1[t] = rnn1[t]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = nb[i];
}

// `side(s)` runs once per tap right after the tap's weights are queued: the fused forward drains its deferred
// a0 stores there, two rows per tap, instead of bursting them in front of the loop.
// (W = TapRegs: 256 threads, six 16-byte pieces each; TapRegs8: 512 threads, three each -- the eight-wave general kernels)
template <int MTW, int NTA, class Side = NoSide, int UNR = 1, class W = TapRegs, int HALF = 0>
__device__ __forceinline__ void conv3_taps(const float* __restrict__ img, float* __restrict__ wbuf,
                                           const float4* __restrict__ wg, W w, const int (&abase)[MTW],
                                           f32x16 (&acc)[MTW][2], int PW, int tid, int lane, Side side = Side()) {
  float4* wl = (float4*)wbuf;
  const uint4* bl = (const uint4*)wbuf + lane;
  constexpr int NT = NTA > 0 ? NTA : 1;
  ASplit cur[NT];                 // step k, split
  float4 rn0[NT], rn1[NT];        // step k+1, raw
#pragma unroll UNR
  for (int s = 0; s < 9; ++s) {
#if CMLPL_ABL == 20         // ablation: tap-0 weights for every tap (no re-staging, no barriers) -- wrong results
    if (s == 0) {
#endif
    __syncthreads();  // everyone done with wbuf of tap s-1 (and, for s == 0, the staged image is complete)
    if constexpr (sizeof(W) == sizeof(TapRegs)) {
      tap_put(wl, w, tid);
      __syncthreads();
      if (s + 1 < 9) w = tap_fetch(wg, s + 1, tid);
    } else {
      tap_put8(wl, w, tid);
      __syncthreads();
      if (s + 1 < 9) w = tap_fetch8(wg, s + 1, tid);
    }
#if CMLPL_ABL == 20
    }
#endif
    side(s);
    if constexpr (NTA > 0) {
      uint4 b[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) b[i] = bl[i * 64];
      if (s == 0) {               // pipeline fill: steps 0 and 1
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
          const float* p0 = img + abase[t] + tap_step_off(0, PW);
          a_split(*(const float4*)p0, *(const float4*)(p0 + 4), cur[t].p1, cur[t].p2, cur[t].p3);
          const float* p1 = img + abase[t] + tap_step_off(1, PW);
          rn0[t] = *(const float4*)p1; rn1[t] = *(const float4*)(p1 + 4);
        }
      }
      tap_step<NTA, 0, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 0, PW);
      tap_step<NTA, 1, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 1, PW);
      tap_step<NTA, 2, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 2, PW);
      tap_step<NTA, 3, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 3, PW);
    }
  }
}

// Tap loop of the per-sample fused kernels (S = 1, at most four M tiles), WITHOUT weight staging and without
// barriers: wave w = (pixel half mh = w >> 1, channel half kh = w & 1) multiplies k-steps 2kh, 2kh+1 of every tap
// (32 of the 64 input channels) into M tiles 2mh, 2mh+1 x both N tiles.  A weight fragment is then needed by two
// waves only, so each wave takes its twelve fragments per tap straight from L2 into registers, a tap ahead -- the
// 24 KiB-per-tap LDS copy and its two workgroup barriers per tap (3.7 us / 4.9 us of the forward / backward tap
// loops, CMLPL_ABL=20) are gone.  The two channel halves meet once, after the loop, through LDS: wave w keeps tile
// w and hands the other tile of its pair to wave w ^ 1 (see conv3_ks_fold).
// Unit = (tile t, k-step q): two ds_read_b128 of raw activations, their split, 12 MFMAs; units are pipelined as in
// tap_step (raw reads two units ahead, split one unit ahead between the MFMAs).
// TPW = M tiles per wave: 2 in the four-wave workgroups (wave = (pixel half, channel half), two workgroups per CU), 1 in
// the eight-wave workgroups (wave = (pixel tile, channel half): ONE sample-net per CU, two waves per SIMD, half the
// MFMAs, splits and fragment reads per wave -- what a rank of a data-parallel job launches when its sample-net
// workgroups do not exceed the CUs).  Units of a tap: I = 0 .. 2 TPW - 1, tile t = I % TPW, k-step q = I / TPW.
template <int I, int TPW>
__device__ __forceinline__ void ks_unit(const float* __restrict__ img, const int (&abase)[TPW], f32x16 (&acc)[TPW][2],
                                        ASplit& cur, float4& rn0, float4& rn1, const uint4 (&bq)[2][6], int s,
                                        int kh, int PW) {
  constexpr int U = 2 * TPW;
  constexpr int t = I % TPW, q = I / TPW;
  // raw reads of unit u + 2
  float4 rnn0, rnn1;
  {
    constexpr int I2 = (I + 2) % U;
    constexpr int t2 = I2 % TPW, q2 = I2 / TPW;
    const int s2 = (I + 2 >= U) ? (s + 1 < 9 ? s + 1 : 8) : s;
    const int kh2 = s2 / 3, kw2 = s2 - kh2 * 3;
    const float* p = img + abase[t2] + ((kh2 - 1) * PW + (kw2 - 1)) * CS + (2 * kh + q2) * 16;
    rnn0 = *(const float4*)p; rnn1 = *(const float4*)(p + 4);
  }
  acc[t][0] = mfma_b3(cur.p1, cur.p2, cur.p3, bq[q][0], bq[q][2], bq[q][4], acc[t][0]);
  acc[t][1] = mfma_b3(cur.p1, cur.p2, cur.p3, bq[q][1], bq[q][3], bq[q][5], acc[t][1]);
  ASplit nxt;
  a_split(rn0, rn1, nxt.p1, nxt.p2, nxt.p3);
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  SchedInterleave<8>::run();
  cur = nxt; rn0 = rnn0; rn1 = rnn1;
}

// one tap of conv3_taps_ks: request the next tap's fragments into `nb`, run this tap's units on `bq`
template <int TPW, class Side>
__device__ __forceinline__ void ks_tap(const float* __restrict__ img, const uint4* __restrict__ wq, const int (&abase)[TPW],
                                       f32x16 (&acc)[TPW][2], ASplit& cur, float4& rn0, float4& rn1, const uint4 (&bq)[2][6],
                                       uint4 (&nb)[2][6], int s, int kh, int PW, bool active, Side& side) {
  if (s + 1 < 9) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 6; ++i)
        nb[q][i] = wq[(size_t)(s + 1) * TAPW + (((2 * kh + q) * 3 + (i >> 1)) * 2 + (i & 1)) * 64];
  }
  side(s);
  if (active) {        // wave-uniform: a pixel half without real pixels (HW <= 64) only keeps the barriers' company
    ks_unit<0, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW);
    ks_unit<1, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW);
    if constexpr (TPW == 2) {
      ks_unit<2, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW);
      ks_unit<3, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW);
    }
  }
}

template <int TPW, class Side = NoSide>
__device__ __forceinline__ void conv3_taps_ks(const float* __restrict__ img, const uint4* __restrict__ wq,
                                              const int (&abase)[TPW], f32x16 (&acc)[TPW][2], int PW, int wave, bool active,
                                              Side side = Side()) {
  const int kh = wave & 1;
  // this wave's fragments of a tap: [k-step q][piece * 2 + n tile]; two sets that swap roles from tap to tap (taps in
  // pairs: copying the set that was fetched ahead into the set in use was 24 64-bit moves per tap and wave)
  uint4 ba[2][6], bb[2][6];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 6; ++i) ba[q][i] = wq[(((2 * kh + q) * 3 + (i >> 1)) * 2 + (i & 1)) * 64];
  ASplit cur;
  float4 rn0, rn1;
  {  // pipeline fill: units 0 and 1 of tap 0 (the staged image is complete: the caller's barrier)
    const float* p0 = img + abase[0] + (-PW - 1) * CS + (2 * kh) * 16;
    a_split(*(const float4*)p0, *(const float4*)(p0 + 4), cur.p1, cur.p2, cur.p3);
    // unit 1: the second tile's first k-step (TPW == 2) or this tile's second k-step (TPW == 1)
    const float* p1 = img + abase[TPW - 1] + (-PW - 1) * CS + (2 * kh + (TPW == 1 ? 1 : 0)) * 16;
    rn0 = *(const float4*)p1; rn1 = *(const float4*)(p1 + 4);
  }
#pragma unroll 1
  for (int s = 0; s < 8; s += 2) {
    ks_tap<TPW>(img, wq, abase, acc, cur, rn0, rn1, ba, bb, s, kh, PW, active, side);
    ks_tap<TPW>(img, wq, abase, acc, cur, rn0, rn1, bb, ba, s + 1, kh, PW, active, side);
  }
  ks_tap<TPW>(img, wq, abase, acc, cur, rn0, rn1, ba, bb, 8, kh, PW, active, side);
}

// conv3_taps_ks on TWO fp16 pieces (common.hpp: h_split / mfma_h2): the same walk -- wave = (pixel half, channel half),
// fragments L2 -> registers a tap ahead, raw activations two units ahead -- with a unit of two ds_read_b128, a 32-
// instruction split (scale, truncate to fp16, exact residual, truncate) and SIX MFMAs where the three-piece bf16 unit has
// 48 and twelve; eight fragments per tap and wave instead of twelve.  `sc` = the sample's power of two (its largest
// magnitude lands in [2^14, 2^15)); the caller multiplies the folded accumulators by 1 / (sc 2^H2_WEXP).
constexpr int TAPH = 4 * 2 * 2 * 64;     // uint4 per tap of a two-piece set: k-steps x pieces x n tiles x lanes
struct HSplit { uint4 p1, p2; };
template <int I, int TPW>
__device__ __forceinline__ void ks_unit_h(const float* __restrict__ img, const int (&abase)[TPW], f32x16 (&acc)[TPW][2],
                                          HSplit& cur, float4& rn0, float4& rn1, const uint4 (&bq)[2][4], int s,
                                          int kh, int PW, float sc) {
  constexpr int U = 2 * TPW;
  constexpr int t = I % TPW, q = I / TPW;
  float4 rnn0, rnn1;
  {
    constexpr int I2 = (I + 2) % U;
    constexpr int t2 = I2 % TPW, q2 = I2 / TPW;
    const int s2 = (I + 2 >= U) ? (s + 1 < 9 ? s + 1 : 8) : s;
    const int kh2 = s2 / 3, kw2 = s2 - kh2 * 3;
    const float* p = img + abase[t2] + ((kh2 - 1) * PW + (kw2 - 1)) * CS + (2 * kh + q2) * 16;
    rnn0 = *(const float4*)p; rnn1 = *(const float4*)(p + 4);
  }
  acc[t][0] = mfma_h2(cur.p1, cur.p2, bq[q][0], bq[q][2], acc[t][0]);
  acc[t][1] = mfma_h2(cur.p1, cur.p2, bq[q][1], bq[q][3], acc[t][1]);
  HSplit nxt;
  h_split(rn0, rn1, sc, nxt.p1, nxt.p2);
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  SchedInterleave<6>::run();
  cur = nxt; rn0 = rnn0; rn1 = rnn1;
}
template <int TPW, class Side>
__device__ __forceinline__ void ks_tap_h(const float* __restrict__ img, const uint4* __restrict__ wq, const int (&abase)[TPW],
                                         f32x16 (&acc)[TPW][2], HSplit& cur, float4& rn0, float4& rn1, const uint4 (&bq)[2][4],
                                         uint4 (&nb)[2][4], int s, int kh, int PW, bool active, float sc, Side& side) {
  if (s + 1 < 9) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        nb[q][i] = wq[(size_t)(s + 1) * TAPH + (((2 * kh + q) * 2 + (i >> 1)) * 2 + (i & 1)) * 64];
  }
  side(s);
  if (active) {
    ks_unit_h<0, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW, sc);
    ks_unit_h<1, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW, sc);
    if constexpr (TPW == 2) {
      ks_unit_h<2, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW, sc);
      ks_unit_h<3, TPW>(img, abase, acc, cur, rn0, rn1, bq, s, kh, PW, sc);
    }
  }
}
template <int TPW, class Side = NoSide>
__device__ __forceinline__ void conv3_taps_ks_h(const float* __restrict__ img, const uint4* __restrict__ wq,
                                                const int (&abase)[TPW], f32x16 (&acc)[TPW][2], int PW, int wave, bool active,
                                                float sc, Side side = Side()) {
  const int kh = wave & 1;
  uint4 ba[2][4], bb[2][4];                // [k-step q][piece * 2 + n tile]
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) ba[q][i] = wq[(((2 * kh + q) * 2 + (i >> 1)) * 2 + (i & 1)) * 64];
  HSplit cur;
  float4 rn0, rn1;
  {
    const float* p0 = img + abase[0] + (-PW - 1) * CS + (2 * kh) * 16;
    h_split(*(const float4*)p0, *(const float4*)(p0 + 4), sc, cur.p1, cur.p2);
    const float* p1 = img + abase[TPW - 1] + (-PW - 1) * CS + (2 * kh + (TPW == 1 ? 1 : 0)) * 16;
    rn0 = *(const float4*)p1; rn1 = *(const float4*)(p1 + 4);
  }
#pragma unroll 1
  for (int s = 0; s < 8; s += 2) {
    ks_tap_h<TPW>(img, wq, abase, acc, cur, rn0, rn1, ba, bb, s, kh, PW, active, sc, side);
    ks_tap_h<TPW>(img, wq, abase, acc, cur, rn0, rn1, bb, ba, s + 1, kh, PW, active, sc, side);
  }
  ks_tap_h<TPW>(img, wq, abase, acc, cur, rn0, rn1, ba, bb, 8, kh, PW, active, sc, side);
}
// The LDS-staged tap loop (conv3_taps) on two fp16 pieces, for the eight-wave general kernels (windows beyond 256 pixels:
// the reference's 20 x 20): a tap's fragments are 16 KiB (4 k-steps x 2 pieces x 2 n tiles x 1 KiB) -- two 16-byte pieces
// per thread global -> registers -> LDS, a tap ahead --, a k-step of a tile is six MFMAs and a 32-instruction split.
struct TapRegs8H { float4 w0, w1; };
__device__ __forceinline__ TapRegs8H tap_fetch8h(const float4* wg, int tap, int tid) {
  TapRegs8H t;
  const float4* wn = wg + tap * TAPH + tid;
  t.w0 = wn[0]; t.w1 = wn[512];
  return t;
}
__device__ __forceinline__ void tap_put8h(float4* wl, const TapRegs8H& t, int tid) { wl[tid] = t.w0; wl[tid + 512] = t.w1; }

template <int NTA, int KQ, int MTW, int HALF = 0>
__device__ __forceinline__ void tap_step_h(const float* __restrict__ img, const uint4* __restrict__ bl,
                                           const int (&abase)[MTW], f32x16 (&acc)[MTW][2], HSplit (&cur)[NTA],
                                           float4 (&rn0)[NTA], float4 (&rn1)[NTA], uint4 (&b)[4], int step, int PW, float sc) {
  uint4 nb[4];                     // [piece * 2 + n tile] of the next k-step
  float4 rnn0[NTA], rnn1[NTA];
#pragma unroll
  for (int i = 0; i < 4; ++i) nb[i] = b[i];
  if (KQ < 3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nb[i] = bl[((KQ + 1) * 4 + i) * 64];
  }
  {
    const int o2 = tap_step_off(step + 2 < 36 ? step + 2 : 35, PW);
#pragma unroll
    for (int t = 0; t < NTA; ++t) {
      rnn0[t] = *(const float4*)(img + abase[t] + o2);
      rnn1[t] = *(const float4*)(img + abase[t] + o2 + 4);
    }
  }
  HSplit nxt[NTA];
  // one MFMA, then one PAIR of the next step's split (scale, truncate, exact residual, truncate: 8 vector instructions),
  // with plain scheduling fences between them -- as the three-piece loop for several tiles per wave does it
#define CMLPL_HPAIR(j) { const float a_ = v[2 * (j)] * sc, b_ = v[2 * (j) + 1] * sc; \
                         const f16x2v p_ = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(a_, b_)); \
                         h1[j] = __builtin_bit_cast(uint32_t, p_); \
                         h2[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a_ - (float)p_[0], b_ - (float)p_[1])); }
#define CMLPL_FENCE __builtin_amdgcn_sched_barrier(0);
  constexpr int NFULL = NTA - (HALF != 0 ? 1 : 0);
#pragma unroll
  for (int t = 0; t < NFULL; ++t) {
    const uint4 H1 = cur[t].p1, H2 = cur[t].p2;
    const float v[8] = {rn0[t].x, rn0[t].y, rn0[t].z, rn0[t].w, rn1[t].x, rn1[t].y, rn1[t].z, rn1[t].w};
    uint32_t h1[4], h2[4];
    f32x16 c0 = acc[t][0], c1 = acc[t][1];
    CMLPL_FENCE
    c0 = mfma_h16(H2, b[0], c0); CMLPL_FENCE CMLPL_HPAIR(0) CMLPL_FENCE
    c0 = mfma_h16(H1, b[2], c0); CMLPL_FENCE CMLPL_HPAIR(1) CMLPL_FENCE
    c0 = mfma_h16(H1, b[0], c0); CMLPL_FENCE CMLPL_HPAIR(2) CMLPL_FENCE
    c1 = mfma_h16(H2, b[1], c1); CMLPL_FENCE CMLPL_HPAIR(3) CMLPL_FENCE
    c1 = mfma_h16(H1, b[3], c1); CMLPL_FENCE
    nxt[t].p1 = make_uint4(h1[0], h1[1], h1[2], h1[3]); nxt[t].p2 = make_uint4(h2[0], h2[1], h2[2], h2[3]);
    CMLPL_FENCE
    c1 = mfma_h16(H1, b[1], c1); CMLPL_FENCE
    acc[t][0] = c0; acc[t][1] = c1;
  }
  if constexpr (HALF != 0) {     // the shared tile: three MFMAs of ONE channel tile, in the order a full tile runs them
    constexpr int t = NTA - 1, h = HALF == 2 ? 1 : 0;
    const uint4 H1 = cur[t].p1, H2 = cur[t].p2;
    const float v[8] = {rn0[t].x, rn0[t].y, rn0[t].z, rn0[t].w, rn1[t].x, rn1[t].y, rn1[t].z, rn1[t].w};
    uint32_t h1[4], h2[4];
    f32x16 ch = acc[t][h];
    CMLPL_FENCE
    ch = mfma_h16(H2, b[0 + h], ch); CMLPL_FENCE CMLPL_HPAIR(0) CMLPL_HPAIR(1) CMLPL_FENCE
    ch = mfma_h16(H1, b[2 + h], ch); CMLPL_FENCE CMLPL_HPAIR(2) CMLPL_HPAIR(3) CMLPL_FENCE
    ch = mfma_h16(H1, b[0 + h], ch); CMLPL_FENCE
    nxt[t].p1 = make_uint4(h1[0], h1[1], h1[2], h1[3]); nxt[t].p2 = make_uint4(h2[0], h2[1], h2[2], h2[3]);
    acc[t][h] = ch;
  }
#undef CMLPL_HPAIR
#undef CMLPL_FENCE
#pragma unroll
  for (int t = 0; t < NTA; ++t) { cur[t] = nxt[t]; rn0[t] = rnn0[t]; rn1[t] = rnn1[t]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = nb[i];
}

template <int MTW, int NTA, int HALF = 0>
__device__ __forceinline__ void conv3_taps_h(const float* __restrict__ img, float* __restrict__ wbuf,
                                             const float4* __restrict__ wg, TapRegs8H w, const int (&abase)[MTW],
                                             f32x16 (&acc)[MTW][2], int PW, int tid, int lane, float sc) {
  float4* wl = (float4*)wbuf;
  const uint4* bl = (const uint4*)wbuf + lane;
  constexpr int NT = NTA > 0 ? NTA : 1;
  HSplit cur[NT];
  float4 rn0[NT], rn1[NT];
#pragma unroll 1
  for (int s = 0; s < 9; ++s) {
    __syncthreads();  // everyone done with wbuf of tap s-1
    tap_put8h(wl, w, tid);
    __syncthreads();
    if (s + 1 < 9) w = tap_fetch8h(wg, s + 1, tid);
    if constexpr (NTA > 0) {
      uint4 b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = bl[i * 64];
      if (s == 0) {               // pipeline fill: steps 0 and 1
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
          const float* p0 = img + abase[t] + tap_step_off(0, PW);
          h_split(*(const float4*)p0, *(const float4*)(p0 + 4), sc, cur[t].p1, cur[t].p2);
          const float* p1 = img + abase[t] + tap_step_off(1, PW);
          rn0[t] = *(const float4*)p1; rn1[t] = *(const float4*)(p1 + 4);
        }
      }
      tap_step_h<NTA, 0, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 0, PW, sc);
      tap_step_h<NTA, 1, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 1, PW, sc);
      tap_step_h<NTA, 2, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 2, PW, sc);
      tap_step_h<NTA, 3, MTW, HALF>(img, bl, abase, acc, cur, rn0, rn1, b, s * 4 + 3, PW, sc);
    }
  }
}

// the staging's share of the two-piece path: a thread's largest |value| -> the workgroup's, in the LDS word `slot` (zeroed
// at kernel start; published by the barrier in front of the tap loop).  Magnitudes are compared as their BIT PATTERNS
// (sign cleared): the order of the non-negative floats, with every NaN above infinity -- an image that holds a NaN is
// "out of range" (it takes the three-piece loop, where the NaN travels as in fp32), never "all zero".
__device__ __forceinline__ uint32_t mag_max(uint32_t m, float v) {
  const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
  return b > m ? b : m;
}
__device__ __forceinline__ uint32_t mag_max4(uint32_t m, const float4& v) {
  return mag_max(mag_max(mag_max(mag_max(m, v.x), v.y), v.z), v.w);
}
__device__ __forceinline__ uint32_t wave_mag_max(uint32_t mx) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)mx, o, 64); mx = x > mx ? x : mx; }
  return mx;
}
__device__ __forceinline__ void h2_publish_max(uint32_t mx, float* slot) {
  mx = wave_mag_max(mx);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)slot, mx);
}

// Tap loop of the EIGHT-wave per-sample workgroups (one workgroup per CU).  Measured with conv3_taps_ks at eight waves
// (round 5): a wave = (pixel tile, channel half) still needs its twelve fragments per tap, so eight waves pull 864 KB of
// weight fragments per sample through the CU's vector-memory path where four pull 432 -- the loop got SLOWER (11.2 us
// against 9.9 for a four-wave workgroup alone on its CU, matrix-pipe floor 6.6), and with nothing else resident every
// tap waits out its own L2 round trip.  With the CU to itself the workgroup has the LDS for what two co-resident
// workgroups could not afford: every tap's 24 KiB of fragments go global -> registers -> LDS ONCE per workgroup (three
// 16-byte pieces per thread, requested two taps ahead), double-buffered (the tap-weight buffer and the fold's exchange
// region), ONE barrier per tap; a wave reads a k-step's six fragments a unit ahead of their MFMAs.
// one unit (tile t = I % TPW, k-step q = I / TPW) with this unit's six fragments in `b`; on exit `b` holds the next
// unit's (read from this tap's LDS copy when the k-step changes inside the tap)
template <int I, int TPW>
__device__ __forceinline__ void ks8_unit(const float* __restrict__ img, const uint4* __restrict__ fl, const int (&abase)[TPW],
                                         f32x16 (&acc)[TPW][2], ASplit& cur, float4& rn0, float4& rn1, uint4 (&b)[6], int s,
                                         int kh, int PW) {
  constexpr int U = 2 * TPW;
  constexpr int t = I % TPW, q = I / TPW;
  constexpr bool next_q = (I + 1 < U) && ((I + 1) / TPW != q);
  uint4 nb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) nb[i] = b[i];
  if constexpr (next_q) {
#pragma unroll
    for (int i = 0; i < 6; ++i) nb[i] = fl[((2 * kh + q + 1) * 6 + i) * 64];
  }
  float4 rnn0, rnn1;
  {
    constexpr int I2 = (I + 2) % U;
    constexpr int t2 = I2 % TPW, q2 = I2 / TPW;
    const int s2 = (I + 2 >= U) ? (s + 1 < 9 ? s + 1 : 8) : s;
    const int kh2 = s2 / 3, kw2 = s2 - kh2 * 3;
    const float* p = img + abase[t2] + ((kh2 - 1) * PW + (kw2 - 1)) * CS + (2 * kh + q2) * 16;
    rnn0 = *(const float4*)p; rnn1 = *(const float4*)(p + 4);
  }
  acc[t][0] = mfma_b3(cur.p1, cur.p2, cur.p3, b[0], b[2], b[4], acc[t][0]);
  acc[t][1] = mfma_b3(cur.p1, cur.p2, cur.p3, b[1], b[3], b[5], acc[t][1]);
  ASplit nxt;
  a_split(rn0, rn1, nxt.p1, nxt.p2, nxt.p3);
  __builtin_amdgcn_sched_group_barrier(0x100, (next_q ? 6 : 0) + 2, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  SchedInterleave<8>::run();
  cur = nxt; rn0 = rnn0; rn1 = rnn1;
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = nb[i];
}

// One tap: its units on the fragments in `cb`, then the NEXT tap's fragments (register set `w`, requested two taps ago)
// into the other buffer -- free since every wave passed the barrier that ended tap s - 1 -- the request for tap s + 3 into
// the same set, and the barrier that ends the tap.
template <int TPW, class Side>
__device__ __forceinline__ void ks8_tap(const float* __restrict__ img, const float* cb, float* nbuf,
                                        const float4* __restrict__ wg, TapRegs8& w, const int (&abase)[TPW],
                                        f32x16 (&acc)[TPW][2], ASplit& cur, float4& rn0, float4& rn1, int s, int kh, int PW,
                                        int tid, int lane, bool active, Side& side) {
  side(s);
  if (active) {
    const uint4* fl = (const uint4*)cb + lane;
    uint4 b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = fl[((2 * kh) * 6 + i) * 64];
    ks8_unit<0, TPW>(img, fl, abase, acc, cur, rn0, rn1, b, s, kh, PW);
    ks8_unit<1, TPW>(img, fl, abase, acc, cur, rn0, rn1, b, s, kh, PW);
    if constexpr (TPW == 2) {
      ks8_unit<2, TPW>(img, fl, abase, acc, cur, rn0, rn1, b, s, kh, PW);
      ks8_unit<3, TPW>(img, fl, abase, acc, cur, rn0, rn1, b, s, kh, PW);
    }
  }
  if (s + 1 < 9) tap_put8((float4*)nbuf, w, tid);
  if (s + 3 < 9) w = tap_fetch8(wg, s + 3, tid);
  __syncthreads();   // tap s is done everywhere (its buffer is free), tap s + 1's fragments are complete
}

// buf0 holds tap 0's fragments and `wa` tap 1's (in registers) on entry, published by the caller's barrier.  Two
// register sets that take turns (taps in pairs), so that a tap's fragments have two tap bodies to arrive.
template <int TPW, class Side = NoSide>
__device__ __forceinline__ void conv3_taps_lds8(const float* __restrict__ img, float* buf0, float* buf1,
                                                const float4* __restrict__ wg, TapRegs8 wa, const int (&abase)[TPW],
                                                f32x16 (&acc)[TPW][2], int PW, int tid, int wave, int lane, bool active,
                                                Side side = Side()) {
  const int kh = wave & 1;
  TapRegs8 wb = tap_fetch8(wg, 2, tid);
  ASplit cur;
  float4 rn0, rn1;
  {  // pipeline fill: units 0 and 1 of tap 0
    const float* p0 = img + abase[0] + (-PW - 1) * CS + (2 * kh) * 16;
    a_split(*(const float4*)p0, *(const float4*)(p0 + 4), cur.p1, cur.p2, cur.p3);
    const float* p1 = img + abase[TPW - 1] + (-PW - 1) * CS + (2 * kh + (TPW == 1 ? 1 : 0)) * 16;
    rn0 = *(const float4*)p1; rn1 = *(const float4*)(p1 + 4);
  }
#pragma unroll 1
  for (int s = 0; s < 8; s += 2) {
    ks8_tap<TPW>(img, buf0, buf1, wg, wa, abase, acc, cur, rn0, rn1, s, kh, PW, tid, lane, active, side);
    ks8_tap<TPW>(img, buf1, buf0, wg, wb, abase, acc, cur, rn0, rn1, s + 1, kh, PW, tid, lane, active, side);
  }
  ks8_tap<TPW>(img, buf0, buf1, wg, wa, abase, acc, cur, rn0, rn1, 8, kh, PW, tid, lane, active, side);
}

// After conv3_taps_ks: fold the two channel halves.  The caller's barrier before (every wave out of its tap loop) and
// the last barrier here (x free again) are part of the protocol.
//   TPW == 2: wave w keeps tile w = 2 (w >> 1) + kh (acc[kh]) and gives acc[kh ^ 1] to its partner w ^ 1, one n tile at
//             a time through `x` (16 KiB: [4 waves][16][64] floats); out[nt] = the kept tile's n tile nt.
//   TPW == 1: both waves of a pair hold the same tile; wave w keeps ITS n tile kh and gives n tile kh ^ 1, in one
//             round through `x` ([8 waves][16][64] floats); out[0] = (tile w >> 1, n tile kh).
template <int TPW>
__device__ __forceinline__ void conv3_ks_fold(f32x16 (&acc)[TPW][2], f32x16 (&out)[TPW], float* x, int wave, int lane) {
  const int kh = wave & 1;
  if constexpr (TPW == 2) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x16 give = kh ? acc[0][nt] : acc[1][nt];
      f32x16 own = kh ? acc[1][nt] : acc[0][nt];
#pragma unroll
      for (int r = 0; r < 16; ++r) x[(wave * 16 + r) * 64 + lane] = give[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) own[r] += x[((wave ^ 1) * 16 + r) * 64 + lane];
      out[nt] = own;
      __syncthreads();
    }
  } else {
    const f32x16 give = kh ? acc[0][0] : acc[0][1];
    f32x16 own = kh ? acc[0][1] : acc[0][0];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[(wave * 16 + r) * 64 + lane] = give[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) own[r] += x[((wave ^ 1) * 16 + r) * 64 + lane];
    out[0] = own;
    __syncthreads();
  }
}

// Everything a 3x3 workgroup does before its tap loop: zero-bordered LDS image of S samples (FWD: the
// activation; DGRAD: dz = mask * upsample(dpool) / 4 formed on the fly), output-pixel LUT, and the tap-0
// weights requested early so their latency overlaps the staging.
struct Conv3Ctx {
  int tid, lane, l31, hh, wave, net, s0, H, W, HW, PW, IMG, H2, W2, P2, RO, CO, PX, S, npx;
  float* img; float* wbuf; int* lut; const float4* wg;
  TapRegs wp;
  TapRegs8 wp8;           // eight-wave workgroups: tap 1's fragments (tap 0's are in LDS when conv3_stage returns)
  TapRegs8H wp8h;         // H2X general kernels: tap 0's two-piece fragments (requested beside wp8: which loop runs is known later)
};

// NW = waves of the workgroup: 4 (every MODE), or 8 for the per-sample kernels (MODE >= 2) when one workgroup has a CU
// to itself (see ks_unit); NT = its threads, TPW = M tiles per wave in the tap loop.
template <int MODE, int NW = 4, int TPW = 8 / NW, bool CUBE = false, bool H2X = false>
__device__ __forceinline__ void conv3_stage(const Conv3Args& a, float* smem, int lut_entries, Conv3Ctx& c,
                                            const float* dp_lds = nullptr, const uint32_t* mpre = nullptr) {
  constexpr int NT = 64 * NW;
  static_assert(!CUBE || MODE == 2, "the cube source feeds the fused forward");
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int net, s0;
  wg_decode(a, net, s0);
  if constexpr (CUBE) s0 = wg_infer_sample(a);          // (>= 0: the kernel returned otherwise)
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2, IMG = (H + 2) * PW;
  const int H2 = H >> 1, W2 = W >> 1, P2 = H2 * W2;
  const int RO = !(MODE & 1) ? 2 * H2 : H, CO = !(MODE & 1) ? 2 * W2 : W;
  const int PX = RO * CO, S = a.S, npx = S * PX;
  float* img = smem;                       // [S][IMG][CS]
  float* wbuf = img + (size_t)S * IMG * CS;  // one tap's weight fragments
  int* lut = (int*)(wbuf + WBUF);          // padded-image position of output pixel m
  // (fused backward head, dp_lds != null: image zero fill and LUT were done before the head, under its loads)
  if (MODE != 2 && dp_lds == nullptr) {  // zero the padded images (border must be zero; interior overwritten below)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* p = (float4*)img;
    const int tot = S * IMG * (CS / 4);
    for (int i = tid; i < tot; i += NT) p[i] = z;
  }
  if (MODE != 2 && dp_lds == nullptr) {
    for (int m = tid; m < lut_entries; m += NT) {
      const int mm = (m < npx) ? m : 0;
      const int s = mm / PX, rem = mm - s * PX, r = rem / CO, c = rem - r * CO;
      lut[m] = s * IMG + (r + 1) * PW + (c + 1);
    }
  }
  const float4* wg = (const float4*)(a.wpk + (long long)net * a.wpk_ns);
  // tap-0 weights: issued now so the HBM/L2 latency overlaps the image staging below
  if constexpr (MODE < 2 && NW == 4) c.wp = tap_fetch(wg, 0, tid);   // (the four-wave per-sample kernels fetch their fragments themselves)
  if constexpr (NW == 8 && (MODE == 3 || MODE < 2)) c.wp8 = tap_fetch8(wg, 0, tid);
  if constexpr (H2X && NW == 8 && MODE < 2) c.wp8h = tap_fetch8h((const float4*)(a.wpk16 + (long long)net * a.wpk16_ns), 0, tid);
  __syncthreads();

  if (MODE == 2) {
    // conv0 (1x1, C -> 64, tools/models.py:102,132) fused in: a0 of this workgroup's sample (S == 1) is computed
    // here, written into the LDS image (the conv1 input) and to HBM (the backward pass reads it).  The image and
    // tap-weight regions are idle until then, so the sample's whole [C][HW] slab is copied into them linearly by
    // global_load_lds_dwordx4 -- every piece in flight at once, ONE wait -- and only afterwards is the region
    // re-initialised as the zero-bordered image.
    // The product runs on the split-bf16 MFMA: wave = (output-channel tile nt, pixel half mh: M tiles 2mh, 2mh+1);
    // A[pixel][band] = slab[band][pixel] (eight ds_read_b32 per tile and k-step of 16 bands, consecutive lanes =
    // consecutive pixels, split in registers); B = this wave's share of the conv0 weight fragments (kernels.hpp:
    // pack_off_w0b3), requested from L2 at kernel start -- nothing is shared between the waves, so no LDS copy.
    // The slab is walked in CHUNKS of 16 bands (= one k-step of the conv0 MFMA): a chunk goes global -> registers
    // (SLAB_WIN chunks in flight per wave, 16-byte loads, 1 KiB per wave-instruction) -> + sigma * noise -> its own
    // LDS slot, and as soon as a chunk is complete (one barrier) its twelve MFMAs per wave run while the later chunks
    // are still landing -- round 2 landed the whole slab by LDS-DMA first (9 us with the noise) and only then
    // multiplied (4 us); the conv0 work now hides under the arrival of the data.  (LDS-DMA would not do here: the
    // compiler makes every LDS access behind an outstanding global_load_lds wait for ALL of them.)  The noise of
    // every chunk of a pass is formed up front, while the first loads are in flight.  Up to SLAB_RING chunks are
    // resident per pass; C > 16 * SLAB_RING (B4: 200 bands) takes further passes through the same slots, the
    // accumulators carried.
    const int C = a.C, KQ0 = (C + 15) >> 4;
    const int CH4 = 4 * HW;                               // float4 per chunk (16 bands x HW floats)
    const int PPW = (CH4 + NT - 1) / NT;                  // float4 per lane and chunk: 1 (HW <= 16 NW) or 2 (HW <= 32 NW)
    const int SLOT = PPW * NT * 4;                        // floats per LDS slot (>= 16 * HW)
    float* slab = smem;                                   // [SLAB_RING][SLOT], aliases img | wbuf | lut
    // wave = pixel tile (32 pixels), BOTH output-channel tiles: the band values of a pixel are split into bf16 pieces
    // once (not once per output-channel tile, as with wave = (channel tile, pixel half))
    const int nfl = C * HW, nf4 = CUBE ? (1 << 30) : (nfl >> 2), rem = CUBE ? 0 : (nfl & 3);
    const float* xrow = CUBE ? a.cube : xsrc_row(a.xs, net, s0, nfl);
    const float sigma = CUBE ? 0.f : a.xs.sigma;
    const float* nzrow = (sigma != 0.f) ? xsrc_noise_row(a.xs, net, s0, nfl) : nullptr;
    const uint64_t gsample = xsrc_global_sample(a.xs, s0);
    const uint64_t rstep = xsrc_step(a.xs);               // counter of the random streams (launch argument, or the device-side row)
    // The augmented rows also go to HBM (16-byte stores from the registers that feed the LDS slots): the backward pass
    // lands them by DMA instead of regenerating the noise -- forming 12,463 normals per sample-net costs ~5 us of vector
    // work per workgroup, the longest single item of that kernel's second half, while these stores ride on an idle HBM.
    float* xnrow = (a.xn_out != nullptr) ? a.xn_out + ((long long)net * a.n + s0) * (long long)nfl : nullptr;
    const uint4* wq0 = (const uint4*)(a.w0t + (long long)net * a.w0t_ns) + lane;
    // the last, partial float4 group of the slab (C * HW need not be a multiple of 4): threads 0 .. rem-1
    float tailv = 0.f;
    if constexpr (!CUBE) if (tid < rem) {
      tailv = xrow[4 * nf4 + tid];
      if (sigma != 0.f) {
        float zt;
        if (nzrow != nullptr) zt = nzrow[4 * nf4 + tid];
        else {
          const float4 t = noise_normal4p(a.xs.seed, rstep, STREAM_NOISE_XP + net, gsample, (uint32_t)nf4);
          zt = tid == 0 ? t.x : tid == 1 ? t.y : t.z;
        }
        tailv = fmaf(zt, sigma, tailv);
      }
    }
    // four waves: wave = pixel tile, BOTH output-channel tiles (C0N = 2); eight waves: wave = (pixel tile, channel tile)
    // -- the same (tile, channel tiles) a wave owns after conv1's fold, see the kernel's epilogue
    constexpr int C0N = TPW;
    const int pt = (TPW == 2) ? wave : (wave >> 1), nt0 = (TPW == 2) ? 0 : (wave & 1);
    f32x16 z[C0N];
#pragma unroll
    for (int i = 0; i < C0N; ++i) z[i] = zero16();
    const int pix0 = (pt * 32 + l31 < HW) ? pt * 32 + l31 : HW - 1;
    for (int c0 = 0; c0 < KQ0; c0 += SLAB_RING) {         // uniform; one pass up to 112 bands
      const int nch = (KQ0 - c0 < SLAB_RING) ? KQ0 - c0 : SLAB_RING;
      // (opaque copy: keeps the compiler from hoisting this pass body's per-lane offsets out of the loop and holding
      // them in registers across it)
      int CH4l = CH4;
      asm volatile("" : "+s"(CH4l));
      const int HWl = CH4l >> 2;
      // item k of this lane inside a chunk (16-byte group index): two neighbouring groups where a lane has two (TPW == 2),
      // so that one eight-normal hash call serves both; item_g0 = the wave's first group (uniform)
      auto item_g = [&](int k) { return TPW == 2 ? 2 * (wave * 64 + lane) + k : wave * 64 + lane; };
      auto item_g0 = [&](int k) { return TPW == 2 ? 2 * wave * 64 + k : wave * 64; };
      if (c0 > 0) __syncthreads();                        // every wave is done reading the previous pass's slots
      uint4 bw[2][3 * C0N];                               // conv0 weight fragments of two chunks (window): [n tile][piece]
      float4 dv[SLAB_WIN][TPW], nzv[SLAB_RING][TPW];      // (a lane's float4 pieces of a chunk: TPW = 512 / NT at most)
      // (Measured and dropped, round 4: network 1 walking the band chunks DOWNWARDS so that the two networks of a sample,
      // which read the same raw rows on the same XCD, would find each other's first half in L2 -- no change, 0.1913 ms
      // either way: the slab's arrival is not what the prologue waits for.)
      auto pch = [&](int kq) { return c0 + kq; };
      auto fetch_b = [&](int kq, uint4 (&b)[3 * C0N]) {
        const int kp = pch(kq);
#pragma unroll
        for (int nt = 0; nt < C0N; ++nt)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) b[3 * nt + pc] = wq0[((kp * 3 + pc) * 2 + nt0 + nt) * 64];
      };
      // CUBE: item g of a chunk = (window pixel g >> 2, bands 16 kq + 4 (g & 3) ..): four consecutive bands of ONE cube
      // pixel (the cube is band-last), found through the mirror index of ExtractPatches; cofs = that pixel's float offset
      int cofs[TPW];
      if constexpr (CUBE) {
        const long long pix = a.pix0 + s0;
        const int pr = (int)(pix / a.ccols), pc = (int)(pix - (long long)pr * a.ccols), hwin = W >> 1;
        const int magicp = (65536 + W - 1) / W;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int g = item_g(k), pw = (g < CH4l ? g : 0) >> 2;
          const int wi = (pw * magicp) >> 16, wj = pw - wi * W;
          int rr = pr + wi - (H >> 1), cc = pc + wj - hwin;
          rr = rr < 0 ? -rr - 1 : (rr >= a.crows ? 2 * a.crows - 1 - rr : rr);
          cc = cc < 0 ? -cc - 1 : (cc >= a.ccols ? 2 * a.ccols - 1 - cc : cc);
          cofs[k] = (rr * a.ccols + cc) * C + 4 * (g & 3);
        }
      }
      // this lane's float4 k of chunk kq: local index g inside the chunk, global index gg inside the slab
      auto fetch_d = [&](int kq, float4 (&d)[TPW]) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int g = item_g(k), gg = pch(kq) * CH4l + g;
          if constexpr (CUBE) {       // four 4-byte loads, each clamped (bands past C read as zero: their weights are zero, the values must be finite)
            const bool ok = g < CH4l;
            const int b0 = 16 * pch(kq) + 4 * (g & 3);
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool v = ok && b0 + e < C;
              const float t = xrow[v ? cofs[k] + 16 * pch(kq) + e : 0];
              x[e] = v ? t : 0.f;
            }
            d[k] = make_float4(x[0], x[1], x[2], x[3]);
          } else {
            const bool ok = g < CH4l && gg < nf4;
            const float4 v = *(const float4*)(xrow + 4 * (ok ? gg : 0));
            d[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      };
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) if (kq < nch) fetch_b(kq, bw[kq]);
#pragma unroll
      for (int kq = 0; kq < SLAB_WIN; ++kq) if (kq < nch) fetch_d(kq, dv[kq]);
      if (!CUBE && c0 + nch == KQ0) {
        // the bands beyond C of the last chunk meet zero weights, but must be finite
        const int used = nfl - (KQ0 - 1) * 16 * HWl;
        float* sl = slab + (nch - 1) * SLOT;
        for (int i = used + tid; i < 16 * HWl; i += NT) sl[i] = 0.f;
      }
      // The noise of a chunk's elements (pure vector work: a hash + Box-Muller per four normals, ~380 cycles per call and
      // wave: 5 us per workgroup pair, the largest item of this prologue).  The first SLAB_NUP chunks' noise is formed
      // up front, under the latency of the first loads (SLAB_NUP == SLAB_RING: all of a pass's chunks).
      // (Two bodies behind ONE uniform branch, not a branch per piece: with "load the reference's draw OR generate" inside
      // one lambda both paths write the same registers, and the compiler guards the generated values' write with a
      // vmcnt(0) against the possibly outstanding load -- which also waits for every chunk load in flight, 14 times.)
      auto noise_live = [&](int kq, int k) { return sigma != 0.f && kq < nch && item_g0(k) < CH4l; };   // uniform
      if (nzrow != nullptr) {                             // parity mode: the reference's own draws
#pragma unroll
        for (int kq = 0; kq < SLAB_NUP; ++kq)
#pragma unroll
          for (int k = 0; k < TPW; ++k) {
            const int g = item_g(k), gg = pch(kq) * CH4l + g;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (noise_live(kq, k)) z = *(const float4*)(nzrow + 4 * ((g < CH4l && gg < nf4) ? gg : 0));
            nzv[kq][k] = z;
          }
      } else if constexpr (TPW == 2) {
        // a lane's two items are NEIGHBOURS (groups gg, gg + 1 with gg even: CH4 is even): one hash call gives both
        // (noise_normal8: eight normals per call)
#pragma unroll
        for (int kq = 0; kq < SLAB_NUP; ++kq) {
          const int gg = pch(kq) * CH4l + item_g(0);
          float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f), z1 = z0;
          if (noise_live(kq, 0)) noise_normal8(a.xs.seed, rstep, STREAM_NOISE_XP + net, gsample, (uint32_t)gg >> 1, z0, z1);
          nzv[kq][0] = z0; nzv[kq][1] = z1;
        }
      } else {
#pragma unroll
        for (int kq = 0; kq < SLAB_NUP; ++kq) {
          const int gg = pch(kq) * CH4l + item_g(0);
          float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          if (noise_live(kq, 0)) z = noise_normal4p(a.xs.seed, rstep, STREAM_NOISE_XP + net, gsample, (uint32_t)gg);
          nzv[kq][0] = z;
        }
      }
      if (c0 == 0 && CMLPL_ABL != 25) STAMP(0, 4);
      // One chunk AHEAD: chunk kq + 1 is put into LDS (noise added) and published before the MFMAs of chunk kq are
      // issued, and its operand reads go out in front of them -- the split of chunk kq + 1 then runs while the matrix
      // pipe works on chunk kq (a chunk at a time, every wave walked read -> split -> MFMA -> barrier in series).
      auto put_chunk = [&](int kq) {                      // registers -> (+ noise) -> LDS slot kq
        float* sl = slab + kq * SLOT;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int g = item_g(k), gg = pch(kq) * CH4l + g;
          if constexpr (CUBE) {     // (band quad, pixel) -> four rows of the slot [16 bands][HW]
            if (g < CH4l) {
              const float4 v = dv[kq % SLAB_WIN][k];
              float* d4 = sl + (4 * (g & 3)) * HWl + (g >> 2);
              d4[0] = v.x; d4[HWl] = v.y; d4[2 * HWl] = v.z; d4[3 * HWl] = v.w;
            }
          } else
          if (g < CH4l && gg < nf4) {
            float4 v = dv[kq % SLAB_WIN][k];
            if (sigma != 0.f) {
              v.x = fmaf(nzv[kq][k].x, sigma, v.x); v.y = fmaf(nzv[kq][k].y, sigma, v.y);
              v.z = fmaf(nzv[kq][k].z, sigma, v.z); v.w = fmaf(nzv[kq][k].w, sigma, v.w);
            }
            *(float4*)(sl + 4 * g) = v;
            if (xnrow != nullptr) *(float4*)(xnrow + 4 * gg) = v;
          }
        }
        if constexpr (!CUBE) if (pch(kq) == KQ0 - 1 && tid < rem) {
          sl[4 * nf4 - (KQ0 - 1) * 16 * HWl + tid] = tailv;
          if (xnrow != nullptr) xnrow[4 * nf4 + tid] = tailv;
        }
      };
      float rn[8];                                        // raw operand values of the chunk ahead
      auto read_chunk = [&](int kq) {
        const float* ap0 = slab + kq * SLOT + hh * 8 * HWl + pix0;
#pragma unroll
        for (int j = 0; j < 8; ++j) rn[j] = ap0[j * HWl];
      };
      uint4 A1, A2, A3;
      put_chunk(0);
      if (SLAB_WIN < nch) fetch_d(SLAB_WIN, dv[0]);
      __syncthreads();                                    // chunk 0 complete in LDS
      if (c0 == 0 && CMLPL_ABL != 25) STAMP(0, 12);
      read_chunk(0);
      a_split(make_float4(rn[0], rn[1], rn[2], rn[3]), make_float4(rn[4], rn[5], rn[6], rn[7]), A1, A2, A3);
#pragma unroll
      for (int kq = 0; kq < SLAB_RING; ++kq) {
        if (kq < nch) {                                   // uniform
          if (kq + 1 < nch) {
            put_chunk(kq + 1);
            if (kq + 1 + SLAB_WIN < nch) fetch_d(kq + 1 + SLAB_WIN, dv[(kq + 1) % SLAB_WIN]);
            __syncthreads();                              // chunk kq + 1 complete in LDS
            read_chunk(kq + 1);
          }
          if (c0 == 0 && kq == 3 && CMLPL_ABL != 25) STAMP(0, 13);
#pragma unroll
          for (int nt = 0; nt < C0N; ++nt)
            z[nt] = mfma_b3(A1, A2, A3, bw[kq % 2][3 * nt], bw[kq % 2][3 * nt + 1], bw[kq % 2][3 * nt + 2], z[nt]);
          if (kq + 2 < nch) fetch_b(kq + 2, bw[kq % 2]);
          if (kq + 1 < nch)
            a_split(make_float4(rn[0], rn[1], rn[2], rn[3]), make_float4(rn[4], rn[5], rn[6], rn[7]), A1, A2, A3);
        }
      }
    }
    __syncthreads();                                      // every wave is done with the slab
    if (CMLPL_ABL != 25) STAMP(0, 5);
    if constexpr (NW == 8) c.wp8 = tap_fetch8(wg, 0, tid);  // (its L2 round trip runs under the border fill / LUT / a0 write)
    {  // now the region becomes the zero-bordered image (the interior is written just below) and the LUT
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const int nbp = 2 * PW + 2 * H;                     // border pixels: top row, bottom row, left / right columns
      for (int i = tid; i < nbp * (CS / 4); i += NT) {
        const int bp = i / (CS / 4), f = i - bp * (CS / 4);
        int pos;
        if (bp < PW) pos = bp;
        else if (bp < 2 * PW) pos = (H + 1) * PW + (bp - PW);
        else { const int k = bp - 2 * PW; pos = (1 + (k >> 1)) * PW + ((k & 1) ? W + 1 : 0); }
        ((float4*)img)[pos * (CS / 4) + f] = z;
      }
      for (int m = tid; m < lut_entries; m += NT) {
        const int mm = (m < npx) ? m : 0;
        const int r = mm / CO, cc = mm - r * CO;
        lut[m] = (r + 1) * PW + (cc + 1);
      }
    }
    if (CMLPL_ABL != 25) STAMP(0, 6);
    const float* b0 = a.b0 + (long long)net * a.b0_ns;
    float bv[C0N];
#pragma unroll
    for (int nt = 0; nt < C0N; ++nt) bv[nt] = b0[32 * (nt0 + nt) + l31];
    const int magic = (65536 + W - 1) / W;                // m / W == (m * magic) >> 16 for every m of the map (checked on the host)
    uint32_t hmx = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = pt * 32 + acc_row(r, lane);
      if (m < HW) {
        const int h = (m * magic) >> 16, w = m - h * W;
        float* d = img + (size_t)((h + 1) * PW + w + 1) * CS + 32 * nt0 + l31;
#pragma unroll
        for (int nt = 0; nt < C0N; ++nt) {
          const float v = z[nt][r] + bv[nt];
          d[32 * nt] = v;
          if constexpr (H2X) hmx = mag_max(hmx, v);
        }
      }
    }
    if constexpr (H2X) h2_publish_max(hmx, smem + a.maxslot);
    if constexpr (NW == 8) {                              // tap 0's fragments into the tap-weight buffer, tap 1's requested
      tap_put8((float4*)wbuf, c.wp8, tid);
      c.wp8 = tap_fetch8(wg, 1, tid);
    }

    __syncthreads();                                      // the LUT (and the image) are complete
  } else if (MODE == 0) {
    const float* src = a.in + (long long)net * a.in_ns;
    uint32_t hmx = 0u;
    staged_copy<8, float4, NT>(S * HW * 16, tid,
        [&](int idx) {
          const int c4 = idx & 15, p = idx >> 4, s = fdiv(p, a.mg_hw), pix = p - s * HW, sample = s0 + s;
          const bool ok = sample < a.n;
          const float4 v = *(const float4*)(src + ((size_t)(ok ? sample : s0) * HW + pix) * 64 + c4 * 4);
          return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        },
        [&](int idx, float4 v) {
          const int c4 = idx & 15, p = idx >> 4, s = fdiv(p, a.mg_hw), pix = p - s * HW, h = fdiv(pix, a.mg_w), w = pix - h * W;
          *(float4*)(img + (size_t)(s * IMG + (h + 1) * PW + w + 1) * CS + c4 * 4) = v;
          if constexpr (H2X) hmx = mag_max4(hmx, v);
        });
    if constexpr (H2X) h2_publish_max(hmx, smem + a.maxslot);
  } else if (dp_lds != nullptr) {
    // fused backward head (S == 1): the pooled gradient was produced by this workgroup and waits in LDS, the ReLU
    // mask words were fetched at kernel start; one (pooled pixel, 4 channels) item -> its 2x2 window
    uint32_t hmx = 0u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = tid + NT * q;
      if (idx < P2 * 16) {
        const int c4 = idx & 15, pp = idx >> 4, ph = fdiv(pp, a.mg_w2), pw = pp - ph * W2;
        const float4 d = *(const float4*)(dp_lds + pp * 64 + c4 * 4);
        const uint32_t m = mpre[q];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          float4 v;
          v.x = ((m >> sub) & 1u) ? d.x * 0.25f : 0.f;
          v.y = ((m >> (8 + sub)) & 1u) ? d.y * 0.25f : 0.f;
          v.z = ((m >> (16 + sub)) & 1u) ? d.z * 0.25f : 0.f;
          v.w = ((m >> (24 + sub)) & 1u) ? d.w * 0.25f : 0.f;
          const int h = 2 * ph + (sub >> 1), w = 2 * pw + (sub & 1);
          *(float4*)(img + (size_t)((h + 1) * PW + w + 1) * CS + c4 * 4) = v;
          if constexpr (H2X) hmx = mag_max4(hmx, v);
        }
      }
    }
    if constexpr (H2X) h2_publish_max(hmx, smem + a.maxslot);
    if constexpr (NW == 8) {       // (the tap-weight buffer held the head's dz2 image, dead since the head's last barrier)
      tap_put8((float4*)wbuf, c.wp8, tid);
      c.wp8 = tap_fetch8(wg, 1, tid);
    }
  } else {
    const float* dp = a.in + (long long)net * a.in_ns;
    const uint8_t* mk = a.mask_in + (long long)net * a.mask_in_ns;
    struct DM { float4 d; uint32_t m; };
    uint32_t hmx = 0u;
    // one (pooled pixel, 4 channels) item feeds the 4 full-resolution positions of its 2x2 window
    staged_copy<8, DM, NT>(S * P2 * 16, tid,
        [&](int idx) {
          const int c4 = idx & 15, pp = idx >> 4, s = fdiv(pp, a.mg_p2), q = pp - s * P2, sample = s0 + s;
          const bool ok = sample < a.n;
          const size_t g = ((size_t)(ok ? sample : s0) * P2 + q) * 64 + c4 * 4;
          DM r;
          r.d = *(const float4*)(dp + g);
          r.m = ok ? *(const uint32_t*)(mk + g) : 0u;
          return r;
        },
        [&](int idx, DM r) {
          const int c4 = idx & 15, pp = idx >> 4, s = fdiv(pp, a.mg_p2), q = pp - s * P2, ph = fdiv(q, a.mg_w2), pw = q - ph * W2;
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            float4 v;
            v.x = ((r.m >> sub) & 1u) ? r.d.x * 0.25f : 0.f;
            v.y = ((r.m >> (8 + sub)) & 1u) ? r.d.y * 0.25f : 0.f;
            v.z = ((r.m >> (16 + sub)) & 1u) ? r.d.z * 0.25f : 0.f;
            v.w = ((r.m >> (24 + sub)) & 1u) ? r.d.w * 0.25f : 0.f;
            const int h = 2 * ph + (sub >> 1), w = 2 * pw + (sub & 1);
            *(float4*)(img + (size_t)(s * IMG + (h + 1) * PW + w + 1) * CS + c4 * 4) = v;
            if constexpr (H2X) hmx = mag_max4(hmx, v);
          }
        });
    if constexpr (H2X) h2_publish_max(hmx, smem + a.maxslot);
  }

  c.tid = tid; c.lane = lane; c.l31 = l31; c.hh = hh; c.wave = wave; c.net = net; c.s0 = s0;
  c.H = H; c.W = W; c.HW = HW; c.PW = PW; c.IMG = IMG; c.H2 = H2; c.W2 = W2; c.P2 = P2;
  c.RO = RO; c.CO = CO; c.PX = PX; c.S = S; c.npx = npx;
  c.img = img; c.wbuf = wbuf; c.lut = lut; c.wg = wg;
}

// avgpool2 + ReLU-mask epilogue shared by the forward kernels (img holds relu(z) at the pixel centres)
template <int NT = 256, bool STORE = true, bool STAT = false>
__device__ __forceinline__ void conv3_pool_store(const Conv3Args& a, const Conv3Ctx& c, float* img2 = nullptr, float* statslot = nullptr) {
  uint32_t pmx = 0u;
  float* out = STORE ? a.out + (long long)c.net * a.out_ns : nullptr;
  uint8_t* mo = STORE ? a.mask_out + (long long)c.net * a.mask_out_ns : nullptr;
  // one (pooled pixel, 4 channels) item per thread and pass: four ds_read_b128, one 16-B and one 4-B store
  const int tot = c.S * c.P2 * 16;
  for (int idx = c.tid; idx < tot; idx += NT) {
    const int c4 = idx & 15, pp = idx >> 4;
    const int s = fdiv(pp, a.mg_p2), q = pp - s * c.P2, ph = fdiv(q, a.mg_w2), pw = q - ph * c.W2;
    const int sample = c.s0 + s;
    if (sample < a.n) {
      const float* p = c.img + (size_t)(s * c.IMG + (2 * ph + 1) * c.PW + 2 * pw + 1) * CS + c4 * 4;
      const float4 v00 = *(const float4*)p, v01 = *(const float4*)(p + CS);
      const float4 v10 = *(const float4*)(p + c.PW * CS), v11 = *(const float4*)(p + c.PW * CS + CS);
      const size_t g = ((size_t)sample * c.P2 + q) * 64 + c4 * 4;
      float4 o;
      o.x = (v00.x + v01.x + v10.x + v11.x) * 0.25f;
      o.y = (v00.y + v01.y + v10.y + v11.y) * 0.25f;
      o.z = (v00.z + v01.z + v10.z + v11.z) * 0.25f;
      o.w = (v00.w + v01.w + v10.w + v11.w) * 0.25f;
      if constexpr (STORE) *(float4*)(out + g) = o;
      if constexpr (STAT) pmx = mag_max4(pmx, o);
      // fused tail: the pooled map also becomes the zero-bordered conv2 input image, in LDS
      if (img2 != nullptr) *(float4*)(img2 + (size_t)((ph + 1) * (c.W2 + 2) + pw + 1) * CS + c4 * 4) = o;
#define CMLPL_NIB(A, B, C, D) ((uint32_t)((relu_open(A) ? 1 : 0) | (relu_open(B) ? 2 : 0) | (relu_open(C) ? 4 : 0) | (relu_open(D) ? 8 : 0)))
      const uint32_t m = CMLPL_NIB(v00.x, v01.x, v10.x, v11.x) | (CMLPL_NIB(v00.y, v01.y, v10.y, v11.y) << 8) |
                         (CMLPL_NIB(v00.z, v01.z, v10.z, v11.z) << 16) | (CMLPL_NIB(v00.w, v01.w, v10.w, v11.w) << 24);
#undef CMLPL_NIB
      if constexpr (STORE) *(uint32_t*)(mo + g) = m;
    }
  }
  if constexpr (STAT) { if (statslot != nullptr) h2_publish_max(pmx, statslot); }
}

// The rest of BaseNet2.forward for this workgroup's sample (S == 1), entered right after conv1's pooled map p1 has
// been written to HBM (backward needs it) and, zero-bordered, to img2 (LDS, in the dead tap-weight buffer):
//   conv2 3x3 + bias + residual + ReLU + avgpool (models.py:137-140) on the split-bf16 MFMA, 16x16x32 shape: the 4x4
//   output pixels that the floor-pooling keeps are ONE 16-row tile; wave w owns output channels 16w..16w+15.  The
//   pooled map is split ONCE into three bf16 planes [pixel][64 ch] in LDS (all four waves need the same A operand);
//   A fragments are ds_read_b128 from the planes, B fragments 16-byte loads of the conv2 forward split set in L2
//   (conv_b3_index: a lane's 8 consecutive ci of one co are contiguous there too), the next tap's six in flight while
//   this tap's twelve MFMAs run.  108 MFMAs of 16 cycles per wave where the f32-input 16x16x4 took 144 of 32.
//   No LDS weight staging, no block barrier in the loop.
//   flatten (NCHW order) + concat with the spectral branch + dropout + classifier, and the L2-normalised spectral
//   feature (models.py:141-152) -- head_fwd_kernel's math on the row this workgroup already holds.
// Requires H4 == W4 == 2 and (H2+2)*(W2+2)*CS <= 4096 floats (windows 8..11).
// Eight-wave workgroups: conv2 and the head stay on waves 0..3 (their chains are one sample's latency, not issue
// slots); waves 4..7 share the strided LDS passes and keep the barriers' company (the same number of barriers on both
// paths: s_barrier counts arrivals).
// argmax of a logits row held in LDS (torch.max(outputs, 1), hyper_tools.py:430: the first index of the maximum; a NaN is
// the maximum and the first one wins), one thread
__device__ __forceinline__ long long logits_argmax(const float* v, int K) {
  int best = 0;
  float bv = v[0];
  for (int k = 1; k < K; ++k) {
    const float x = v[k];
    if (bv == bv && (x != x || x > bv)) { best = k; bv = x; }
  }
  return (long long)best;
}

template <int NW = 4, bool INFER = false>
__device__ __forceinline__ void conv3_fwd_tail(const Conv3Args& a, const Conv3Ctx& c, float* smem) {
  constexpr int NT = 64 * NW;
  const int tid = c.tid, lane = c.lane, wave = c.wave, net = c.net, sample = c.s0;
  const int PW2 = c.W2 + 2;
  const float* img2 = c.wbuf;
  float* row = smem;                         // [F] head input row (the conv1 image is dead)
  const int SF = 256, F = SF + FD, K = a.K;
  float* red = smem + F;                     // [4] + [4][64]
  const long long rs = (long long)net * a.n + sample;
  // ---- the pooled map as three bf16 planes [pixel][64 ch] (pixel stride 36 dwords: the 16-byte reads of the 16 output
  // pixels land two-way on the banks at worst), behind the head row in the dead image region
  constexpr int PS2 = 36;
  const int NPX2 = (c.H2 + 2) * PW2, PLN = NPX2 * PS2;
  uint32_t* pl = (uint32_t*)(smem + 2048);
  auto split_planes = [&]() {
    for (int it = tid; it < NPX2 * 16; it += NT) {
      const int px = it >> 4, c4 = it & 15;
      const float4 v = *(const float4*)(img2 + (size_t)px * CS + 4 * c4);
      const float x[4] = {v.x, v.y, v.z, v.w};
      uint32_t u0[4], u1[4], u2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u0[q] = __float_as_uint(x[q]);
        const float r1 = x[q] - __uint_as_float(u0[q] & 0xffff0000u);
        u1[q] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1[q] & 0xffff0000u);
        u2[q] = __float_as_uint(r2);
      }
      uint32_t* d = pl + px * PS2 + 2 * c4;
      *(uint2*)(d) = make_uint2(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]));
      *(uint2*)(d + PLN) = make_uint2(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]));
      *(uint2*)(d + 2 * PLN) = make_uint2(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]));
    }
  };
  static_assert(!(INFER && NW == 8), "inference from the cube: four-wave or eight-tile kernels");
  if (NW == 8 && wave >= 4) {                // (uniform) the barriers of the path below, one for one
    __syncthreads();
    split_planes();
    __syncthreads();
    __syncthreads();
    __syncthreads();
    __syncthreads();
    return;
  }
  // ---- loads issued up front: tap-0 B fragments, this thread's slice of the spectral row
  const int j = lane & 15, kg = lane >> 4;
  // B fragment (tap, k-step ks of 32 ci, piece p) of this wave's 16 output channels: uint4 index
  //   (((tap*4 + 2 ks + (kg >> 1)) * 3 + p) * 2 + (wave >> 1)) * 64 + (kg & 1) * 32 + 16 (wave & 1) + j
  const uint4* wq = (const uint4*)(a.w2f + (long long)net * a.w2f_ns) + ((kg >> 1) * 6 + (wave >> 1)) * 64 + (kg & 1) * 32 +
                    16 * (wave & 1) + j;
  // ring of C2_AHEAD + 1 fragment sets [ks][piece]: tap t's twelve MFMAs are ~0.1 us of matrix pipe, an L2 round trip is
  // several times that -- with ONE tap of look-ahead every tap waited out most of a round trip (9 in a row: 5 of the
  // 6 us this convolution took); the loop below is fully unrolled, so the ring indices are compile-time constants
  constexpr int C2_AHEAD = 3;
  uint4 bq[C2_AHEAD + 1][6];
#pragma unroll
  for (int t0 = 0; t0 < C2_AHEAD; ++t0)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) bq[t0][3 * ks + pc] = wq[((t0 * 4 + 2 * ks) * 3 + pc) * 128];
  const float4 y4 = *(const float4*)(a.yin + rs * FD + 4 * tid);
  const float bias2 = (a.b2 + (long long)net * a.p_ns)[wave * 16 + j];
  const float bcv = (a.bc + (long long)net * a.p_ns)[tid < K ? tid : 0];   // (needed by the very last statement: not behind the last barrier)
  // ... and what the head will want after conv2, so that its L2 round trips run under conv2's: the classifier rows of
  // the first 16 classes (wave w owns features [w F/4, (w+1) F/4) of the row, 5 per lane) and the dropout multipliers
  // of this thread's two float4 of the row (counter-based: they depend on nothing computed here)
  const float* wc = a.wc + (long long)net * a.p_ns;
  const int F4 = F >> 2, fb = wave * F4;   // 320 features per wave = 5 per lane
  float wv0[16][5];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float* wr = wc + (long long)(q < K ? q : K - 1) * F + fb + lane;
#pragma unroll
    for (int i = 0; i < 5; ++i) wv0[q][i] = wr[64 * i];     // (loading only the classes that exist, behind uniform branches, was 1 us slower)
  }
  // 0 = none, 1 = explicit mask, 2 = generate (Philox) and record for the backward pass
  const int dmode = (!a.train || a.dropout_p <= 0.f) ? 0 : (a.dropmask != nullptr ? 1 : 2);
  const float keep_scale = 1.0f / (1.0f - a.dropout_p);
  float4 dm4[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int f0 = 1024 * q + 4 * tid;
    dm4[q] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (f0 < F) {
      if (dmode == 1) {
        dm4[q] = *(const float4*)(a.dropmask + rs * F + f0);
      } else if (dmode == 2) {
        const unsigned long long gs = xsrc_global_sample(a.xs, sample);
        const float4 u = philox_uniform4(a.xs.seed, xsrc_step(a.xs), STREAM_DROPOUT + net, (gs * F + f0) >> 2);
        float4 m4;
        m4.x = (u.x >= a.dropout_p) ? keep_scale : 0.f; m4.y = (u.y >= a.dropout_p) ? keep_scale : 0.f;
        m4.z = (u.z >= a.dropout_p) ? keep_scale : 0.f; m4.w = (u.w >= a.dropout_p) ? keep_scale : 0.f;
        *(float4*)(a.dropgen + rs * F + f0) = m4;
        dm4[q] = m4;
      }
    }
  }
  if (CMLPL_ABL == 25) STAMP(0, 4);          // (tail-detail timeline build: the prologue's stamp slots are re-used)
  __syncthreads();                           // img2 interior complete; every thread is done pooling from img
  if (CMLPL_ABL == 25) STAMP(0, 5);
  *(float4*)(row + SF + 4 * tid) = y4;       // spectral part of the head row (pre-dropout); row aliases the dead img
  split_planes();
  __syncthreads();
  if (CMLPL_ABL == 25) STAMP(0, 6);
  // ---- conv2: pixel i = lane & 15 = (oh, ow) = (i >> 2, i & 3); lane group kg holds ci 8 kg .. 8 kg + 7 of a k-step
  const uint32_t* ap0 = pl + (size_t)(((j >> 2) + 1) * PW2 + (j & 3) + 1) * PS2 + 4 * kg;
  f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  auto mm = [](const uint4& x, const uint4& y, f32x4v cc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), cc, 0, 0, 0);
  };
  // A fragments (three planes x two k-steps = six ds_read_b128) are read a whole tap AHEAD of their MFMAs: written as
  // read -> multiply per k-step, every MFMA group sat behind an LDS round trip (54 waits, ~2 us of this 4.5-us loop)
  uint4 af[2][6];
  auto read_a = [&](int tap, uint4 (&x)[6]) {
    const uint32_t* ap = ap0 + ((tap / 3 - 1) * PW2 + (tap % 3 - 1)) * PS2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      x[3 * ks] = *(const uint4*)(ap + 16 * ks); x[3 * ks + 1] = *(const uint4*)(ap + PLN + 16 * ks);
      x[3 * ks + 2] = *(const uint4*)(ap + 2 * PLN + 16 * ks);
    }
  };
  read_a(0, af[0]);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap + C2_AHEAD < 9) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bq[(tap + C2_AHEAD) % (C2_AHEAD + 1)][3 * ks + pc] = wq[(((tap + C2_AHEAD) * 4 + 2 * ks) * 3 + pc) * 128];
    }
    if (tap + 1 < 9) read_a(tap + 1, af[(tap + 1) & 1]);
    const uint4 (&bcur)[6] = bq[tap % (C2_AHEAD + 1)];
    const uint4 (&acur)[6] = af[tap & 1];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 a1 = acur[3 * ks], a2 = acur[3 * ks + 1], a3 = acur[3 * ks + 2];
      const uint4 b1 = bcur[3 * ks], b2 = bcur[3 * ks + 1], b3 = bcur[3 * ks + 2];
      // the six products of weight >= 2^-16, two accumulator chains (a dependent MFMA waits for its predecessor)
      acc0 = mm(a1, b3, acc0); acc1 = mm(a2, b2, acc1); acc0 = mm(a3, b1, acc0);
      acc1 = mm(a1, b2, acc1); acc0 = mm(a2, b1, acc0); acc1 = mm(a1, b1, acc1);
    }
    // pin the order: this tap's loads and reads first, then its twelve MFMAs
    __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);    // VMEM reads (none in the last taps: the group is then empty)
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);    // DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);   // MFMA
  }
  STAMP(0, 8);
  // ---- conv2 epilogue: lane (co = 16 wave + j, output row oh = kg) holds the 4 pixels ow = 0..3 of that row
  {
    const int co = wave * 16 + j;
    const float* res = img2 + (size_t)((kg + 1) * PW2 + 1) * CS + co;   // p1 at (oh, ow = 0): the residual branch
    float r_[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) r_[r] = relu_nan((acc0[r] + acc1[r]) + bias2 + res[r * CS]);
    const float s0 = r_[0] + r_[1], s1 = r_[2] + r_[3];
    const float t0 = s0 + __shfl_xor(s0, 16, 64), t1 = s1 + __shfl_xor(s1, 16, 64);   // rows oh ^ 1
    const int dh = kg & 1;
    uint32_t n0 = (relu_open(r_[0]) ? 1u : 0u) | (relu_open(r_[1]) ? 2u : 0u);
    uint32_t n1 = (relu_open(r_[2]) ? 1u : 0u) | (relu_open(r_[3]) ? 2u : 0u);
    n0 <<= 2 * dh; n1 <<= 2 * dh;
    n0 |= (uint32_t)__shfl_xor((int)n0, 16, 64); n1 |= (uint32_t)__shfl_xor((int)n1, 16, 64);
    if (dh == 0) {
      const int ph = kg >> 1;
      const float o0 = t0 * 0.25f, o1 = t1 * 0.25f;
      if constexpr (!INFER) {
        float* p2 = a.p2out + (rs * 4 + ph * 2) * 64 + co;
        uint8_t* m2 = a.m2out + (rs * 4 + ph * 2) * 64 + co;
        p2[0] = o0; p2[64] = o1;
        m2[0] = (uint8_t)n0; m2[64] = (uint8_t)n1;
      }
      row[co * 4 + ph * 2] = o0;             // canonical flatten order f = c * HW4 + hw (x.view, models.py:141)
      row[co * 4 + ph * 2 + 1] = o1;
    }
  }
  if (CMLPL_ABL == 25) STAMP(0, 12);
  // ---- head
  float ss = (y4.x * y4.x + y4.y * y4.y) + (y4.z * y4.z + y4.w * y4.w);
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();                           // row[0..256) complete, red[] written
  if (CMLPL_ABL == 25) STAMP(0, 13);
  if constexpr (!INFER) {
    if (a.feat != nullptr) {                 // (null: feat_norm_kernel formed the embeddings behind the spectral branch)
      const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
      if (tid == 0) a.ynorm[rs] = norm;
      float4 o = y4;
      o.x /= norm; o.y /= norm; o.z /= norm; o.w /= norm;
      *(float4*)(a.feat + rs * FD + 4 * tid) = o;
    }
    float* catd = a.catd + rs * F;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f0 = 1024 * q + 4 * tid;
      if (f0 < F) {
        float4 x = *(const float4*)(row + f0);
        if (dmode != 0) { x.x *= dm4[q].x; x.y *= dm4[q].y; x.z *= dm4[q].z; x.w *= dm4[q].w; }
        *(float4*)(catd + f0) = x;
        *(float4*)(row + f0) = x;            // same thread re-writes what it read
      }
    }
  }
  __syncthreads();
  STAMP(0, 9);
  // logits: wave w takes the quarter [w*F4, (w+1)*F4) of the row for ALL classes (16 at a time), then the four
  // partial dot products meet in LDS
  float* part = red + 4;                     // [4 waves][64 classes]
  float xr[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xr[i] = row[fb + lane + 64 * i];
  {
    // all sixteen reductions in ONE basic block, step by step (the same sums as wave_sum): behind a `q < K` branch each
    // class's six dependent cross-lane steps ran alone -- nine chains in a row, ~2 us of this sample's critical path
    float t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) acc = fmaf(xr[i], wv0[q][i], acc);
      t[q] = acc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] += __shfl_xor(t[q], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (q < K) part[wave * 64 + q] = t[q];
    }
  }
  for (int kc = 16; kc < K; kc += 16) {      // more than 16 classes: further chunks
    // 16 classes x 5 features: all 80 weight loads are issued before the first is used (one L2 round trip per
    // chunk; a load -> fma -> load loop here cost ~7 us per workgroup)
    float wv[16][5];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float* wr = wc + (long long)((kc + q < K) ? kc + q : K - 1) * F + fb + lane;
#pragma unroll
      for (int i = 0; i < 5; ++i) wv[q][i] = wr[64 * i];
    }
    float t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) acc = fmaf(xr[i], wv[q][i], acc);
      t[q] = acc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] += __shfl_xor(t[q], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (kc + q < K) part[wave * 64 + kc + q] = t[q];
    }
  }
  __syncthreads();
  if constexpr (INFER) {     // inference: the label (and, when asked for, the logits) of pixel pix0 + sample
    if (tid < K) {
      const float lg = ((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) + bcv;
      part[256 + tid] = lg;
      if (a.logits != nullptr) a.logits[rs * K + tid] = lg;
    }
    __syncthreads();
    if (tid == 0) a.labels_out[rs] = logits_argmax(part + 256, K);
  } else {
    if (tid < K) a.logits[rs * K + tid] = ((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) + bcv;
  }
}

// The first part of the backward pass of this workgroup's sample (S == 1), run in front of the conv1 data gradient:
//   head: dcat = (dlogits . Wc) * dropout multiplier; its spatial part is the pooled conv2 gradient dp2, its
//         spectral part joins the gradient through the L2 normalisation and the spectral ReLU (head_bwd_kernel's
//         math): dy = relu'(y) * (dcat_y + (dfeat - feat <feat, dfeat>) / ||y||)
//   conv2 data gradient on the 16x16x4 fp32 MFMA: dz2 = mask2 * upsample(dp2) / 4 as a zero-bordered LDS image,
//         dp1 = conv_transpose(dz2) + dz2 for the H2 x W2 pixels (two 16-row tiles), wave w = input channels
//         16w..16w+15, B fragments ready-made in L2.
// dy, dp2 and dp1 also go to HBM (the weight-gradient kernels read them); dp1 stays in LDS (returned region) for the
// conv1 staging.  LDS use: img2d in the tap-weight buffer, everything else behind the LUT (dead before the tap loop).
// Eight-wave workgroups: the head and the conv2 data gradient stay on waves 0..3 (first version); waves 4..7 fetch their
// share of conv1's mask words and keep the barriers' company.
template <int NW = 4>
__device__ __forceinline__ const float* conv3_bwd_head(const Conv3Args& a, float* smem, uint32_t (&mpre)[2]) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int net, sample;
  wg_decode(a, net, sample);
  const int H2 = a.H >> 1, W2 = a.W >> 1, P2 = H2 * W2, PW2 = W2 + 2;
  const int SF = 256, F = SF + FD, K = a.K;
  float* img2 = smem + (size_t)(a.H + 2) * (a.W + 2) * CS;        // = wbuf: [(H2+2)*(W2+2)][CS] <= 4096 floats
  float* ext = img2 + WBUF + 128;                                  // behind the LUT
  float* dp1s = ext;                                               // [P2][64]
  float* dp2s = dp1s + P2 * 64;                                    // [4][64]  pooled conv2 gradient [hw][c]
  float* dls = dp2s + 256;                                         // [64]
  float* red = dls + 64;                                           // [4]
  const long long rs = (long long)net * a.n + sample;
  // conv1's ReLU-mask words of this thread's staging items (consumed after the conv2 data gradient)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + NT * q;
    mpre[q] = (idx < P2 * 16) ? *(const uint32_t*)(a.mask_in + (long long)net * a.mask_in_ns + ((size_t)sample * P2 + (idx >> 4)) * 64 + (idx & 15) * 4) : 0u;
  }
  if (NW == 8 && wave >= 4) {                // (uniform) the barriers of the path below, one for one
    __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
    return dp1s;
  }
  const int l31 = lane & 31, hh = lane >> 5;
  // conv2 data gradient: wave = (output-channel tile nt, half kh2 of every tap's 64 input channels)
  const int nt = wave & 1, kh2 = wave >> 1;
  // ---- loads up front: tap-0 B fragments of the conv2 data gradient, this thread's spectral elements
  const uint4* wq = (const uint4*)(a.w2d + (long long)net * a.w2d_ns) + lane;
  // (a ring of C2_AHEAD + 1 fragment sets, as in conv3_fwd_tail: a tap is 12 MFMAs, far less than an L2 round trip)
  constexpr int C2_AHEAD = 3;
  uint4 bq[C2_AHEAD + 1][6];
#pragma unroll
  for (int t0 = 0; t0 < C2_AHEAD; ++t0)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) bq[t0][3 * q + pc] = wq[(size_t)t0 * TAPW + (((kh2 * 2 + q) * 3 + pc) * 2 + nt) * 64];
  if (tid < 64) dls[tid] = (tid < K) ? a.dlogits[rs * K + tid] : 0.f;
  const float* y = a.yin + rs * FD;
  const float* df = (a.dfeat != nullptr) ? a.dfeat + rs * FD : nullptr;
  const float norm = a.ynrm[rs];
  float yv[4], dv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    yv[q] = y[tid + 256 * q];
    dv[q] = (df != nullptr) ? df[tid + 256 * q] : 0.f;
  }
  // conv2's ReLU-mask word of this thread's dz2 item (threads 0..63: pooled pixel tid >> 4, channels 4 (tid & 15)..)
  const uint32_t m2pre = (tid < 64) ? *(const uint32_t*)(a.m2in + (rs * 4 + (tid >> 4)) * 64 + (tid & 15) * 4) : 0u;
  const float* wc = a.wc + (long long)net * a.p_ns;
  const float* dm = (a.hmask != nullptr) ? a.hmask + rs * F : nullptr;
  float dmv[5];
#pragma unroll
  for (int it = 0; it < 5; ++it) dmv[it] = (dm != nullptr) ? dm[tid + 256 * it] : 1.f;
  // classifier weights of this thread's 5 row elements f = tid + 256 it, 16 classes at a time: every load of a
  // chunk is in flight before the first is used
  float wv0[16][5];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float* wr = wc + (long long)(q < K ? q : K - 1) * F + tid;
#pragma unroll
    for (int it = 0; it < 5; ++it) wv0[q][it] = wr[256 * it];
  }
  {  // while those loads fly: the whole tap-weight buffer becomes the zero-bordered dz2 image, and the conv1 image
     // region (free until the conv1 staging) gets its zero fill and output-pixel LUT
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 1024; i += 256) ((float4*)img2)[i] = z4;
    const int PW = a.W + 2, IMG = (a.H + 2) * PW;
    for (int i = tid; i < IMG * (CS / 4); i += 256) ((float4*)smem)[i] = z4;
    int* lut = (int*)(img2 + WBUF);
    const int HWl = a.H * a.W;
    for (int m = tid; m < 128; m += 256) {
      const int mm = (m < HWl) ? m : 0;
      const int r = mm / a.W, cc = mm - r * a.W;
      lut[m] = (r + 1) * PW + (cc + 1);
    }
  }
  float dot = 0.f;   // <feat, dfeat> with feat = y / ||y|| (same division as the forward pass)
#pragma unroll
  for (int q = 0; q < 4; ++q) dot = fmaf(yv[q] / norm, dv[q], dot);
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  float* dp2g = a.dp2out + rs * SF;
  float* dyg = a.dy + rs * FD;
  float dc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float dl = dls[q];                                       // zero beyond K
#pragma unroll
    for (int it = 0; it < 5; ++it) dc[it] = fmaf(dl, wv0[q][it], dc[it]);
  }
  for (int kc = 16; kc < K; kc += 16) {                            // more than 16 classes: further chunks
    float wv[16][5];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float* wr = wc + (long long)(kc + q < K ? kc + q : K - 1) * F + tid;
#pragma unroll
      for (int it = 0; it < 5; ++it) wv[q][it] = wr[256 * it];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float dl = dls[kc + q];
#pragma unroll
      for (int it = 0; it < 5; ++it) dc[it] = fmaf(dl, wv[q][it], dc[it]);
    }
  }
#pragma unroll
  for (int it = 0; it < 5; ++it) {                                 // F = 1280 = 5 x 256, f = tid + 256 it
    const int f = tid + 256 * it;
    const float d = (dm != nullptr) ? dc[it] * dmv[it] : dc[it];
    if (it == 0) {                                                 // f < SF: spatial part, f = c * 4 + hw
      const int c = f >> 2, hw = f & 3;
      dp2s[hw * 64 + c] = d;
      dp2g[hw * 64 + c] = d;
    } else {                                                       // spectral element j = f - 256 = tid + 256 (it - 1)
      const float yj = yv[it - 1];
      float g = d;
      if (df != nullptr) g += (dv[it - 1] - (yj / norm) * dot) / norm;
      dyg[tid + 256 * (it - 1)] = relu_open(yj) ? g : 0.f;
    }
  }
  STAMP(1, 10);
  __syncthreads();                                                 // dp2s complete, img2 zeroed
  STAMP(1, 4);
  if (tid < 64) {   // dz2 = mask2 * upsample(dp2) / 4: (pooled pixel, 4 channels) -> its 2x2 window
    const int c4 = tid & 15, pp = tid >> 4, ph = pp >> 1, pw = pp & 1;
    const float4 d = *(const float4*)(dp2s + pp * 64 + c4 * 4);
    const uint32_t m = m2pre;
    uint32_t zmx = 0u;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      float4 v;
      v.x = ((m >> sub) & 1u) ? d.x * 0.25f : 0.f;
      v.y = ((m >> (8 + sub)) & 1u) ? d.y * 0.25f : 0.f;
      v.z = ((m >> (16 + sub)) & 1u) ? d.z * 0.25f : 0.f;
      v.w = ((m >> (24 + sub)) & 1u) ? d.w * 0.25f : 0.f;
      const int h = 2 * ph + (sub >> 1), w = 2 * pw + (sub & 1);
      *(float4*)(img2 + (size_t)((h + 1) * PW2 + w + 1) * CS + c4 * 4) = v;
      zmx = mag_max4(zmx, v);
    }
    if (a.hstat != nullptr) {       // (wave 0 whole: the batch-level maximum of conv2's gradient operand, for its weight gradient)
      zmx = wave_mag_max(zmx);
      if (tid == 0) { a.hstat[(3 * 2 + net) * a.n + sample] = zmx; ((uint32_t*)smem)[a.maxslot + 1] = zmx; }
    }
  }
  __syncthreads();
  // a gradient image that is zero everywhere (a row no loss term reaches through the convolutions) meets finite weights
  // (range flag clear): conv2's data gradient is zero -- its MFMA loop is skipped, the epilogue writes the zeros
  const bool z2skip = a.hstat != nullptr && a.h2_noskip == 0 && ((const volatile uint32_t*)smem)[a.maxslot + 1] == 0u &&
                      a.h2flag[(long long)net * a.h2flag_ns] == 0u;
  STAMP(1, 5);
  // ---- conv2 data gradient, split-bf16 on the 32x32x16 MFMA: the P2 <= 32 output pixels are ONE tile (row p = l31,
  // rows >= P2 read the zero corner); this wave multiplies its half of the input channels of every tap into its
  // output-channel tile.  Nothing is shared between the waves, so the B fragments go L2 -> registers directly
  // (six 1 KiB loads per tap, the next tap's in flight) and the loop has no barrier.
  // (p / W2 by multiplication: a 32-bit division by a run-time value is ~40 vector instructions, and the fold below did
  //  sixteen of them per lane)
  const int mg2 = (65536 + W2 - 1) / W2;                           // p / W2 == (p * mg2) >> 16 for p < 4096 / W2 ... (p W2 < 65536)
  const int phA = (l31 * mg2) >> 16;
  const int posA = l31 < P2 ? (phA + 1) * PW2 + (l31 - phA * W2) + 1 : 0;
  const float* ap = img2 + (size_t)posA * CS + kh2 * 32 + hh * 8;
  f32x16 acc = zero16();
  // Units u = (tap, k-step q) of 16 input channels, software-pipelined as in ks_unit: unit u's two ds_read_b128 are
  // issued two units ahead, its split runs between the six MFMAs of unit u - 1 (pinned: left alone, the scheduler
  // sinks each read to just in front of its split and every k-step waits out an LDS round trip).
  auto raw_ptr = [&](int u) {
    const int tap = u >> 1;
    const int toff = ((tap / 3 - 1) * PW2 + (tap % 3 - 1)) * CS;
    // the dummy rows (p >= P2) sit at the zero corner: a negative tap offset would leave the image, keep them there
    return ((l31 < P2) ? ap + toff : ap) + 16 * (u & 1);
  };
  ASplit cur;
  float4 rn0, rn1;
  if (!z2skip) {
  {
    const float* p0 = raw_ptr(0);
    a_split(*(const float4*)p0, *(const float4*)(p0 + 4), cur.p1, cur.p2, cur.p3);
    const float* p1 = raw_ptr(1);
    rn0 = *(const float4*)p1; rn1 = *(const float4*)(p1 + 4);
  }
#pragma unroll
  for (int u = 0; u < 18; ++u) {
    const int tap = u >> 1, q = u & 1;
    if (q == 0 && tap + C2_AHEAD < 9) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bq[(tap + C2_AHEAD) % (C2_AHEAD + 1)][3 * q2 + pc] =
              wq[(size_t)(tap + C2_AHEAD) * TAPW + (((kh2 * 2 + q2) * 3 + pc) * 2 + nt) * 64];
    }
    if (q == 0) __builtin_amdgcn_sched_barrier(0);        // the loads stay in front of the pinned groups of this tap's units
    float4 rnn0 = rn0, rnn1 = rn1;
    if (u + 2 < 18) {
      const float* p2 = raw_ptr(u + 2);
      rnn0 = *(const float4*)p2; rnn1 = *(const float4*)(p2 + 4);
    }
    const uint4 (&bcur)[6] = bq[tap % (C2_AHEAD + 1)];
    acc = mfma_b3(cur.p1, cur.p2, cur.p3, bcur[3 * q], bcur[3 * q + 1], bcur[3 * q + 2], acc);
    ASplit nxt = cur;
    if (u + 1 < 18) a_split(rn0, rn1, nxt.p1, nxt.p2, nxt.p3);
    if (u + 2 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    SchedInterleave<6>::run();                             // (6 VALU, 1 MFMA) x 6; the rest of the split behind them
    cur = nxt; rn0 = rnn0; rn1 = rnn1;
  }
  }
  STAMP(1, 6);
  // ---- fold the two channel halves through LDS (the dz2 image region, once every wave is done reading it) and
  // finish: lane (ci = 32 nt + l31) holds pixels p = acc_row(r); the residual branch adds dz2 itself
  {
    const int ci = nt * 32 + l31;
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = acc_row(r, lane), ph = (p * mg2) >> 16;
      res[r] = (kh2 == 0 && p < P2) ? img2[(size_t)((ph + 1) * PW2 + (p - ph * W2) + 1) * CS + ci] : 0.f;
    }
    __syncthreads();                                               // every wave has finished reading img2
    STAMP(1, 7);
    float* xch = img2 + nt * 1024;                                 // [16][64] per output-channel tile
    if (kh2 == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    STAMP(1, 8);
    if (kh2 == 0) {
      float* dp1g = a.dp1out + rs * (long long)P2 * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = acc_row(r, lane);
        if (p < P2) {
          const float v = (acc[r] + xch[r * 64 + lane]) + res[r];
          dp1s[p * 64 + ci] = v;
          dp1g[p * 64 + ci] = v;
        }
      }
    }
  }
  __syncthreads();                                                 // dp1s complete; img2 (= wbuf) is free again
  STAMP(1, 11);
  return dp1s;
}

// ------------------------------------------------------------------------------------------
// Generalised tail / head of the eight-wave, eight-tile per-sample kernels (windows of 129 .. 256 pixels: BASELINE
// configs[4]'s 15 x 15 x 48): final pooled maps H4 x W4 of up to 12 pixels (conv3_fwd_tail / conv3_bwd_head above are
// the 2 x 2 case, hand-fitted to four waves), pooled conv1 maps of up to 64 pixels, classifier rows of any length
// F = 64 H4 W4 + 1024 taken as 16-byte groups (one group per thread).
// ------------------------------------------------------------------------------------------
// conv2 3x3 + bias + residual + ReLU + avgpool (models.py:137-140) on the 16x16x32 split-bf16 MFMA with the output pixels
// in WINDOW-major order: pixel i of tile t is sub-pixel (i & 3) of pooling window 4 t + (i >> 2), so a lane's four
// accumulator registers are one 2 x 2 window -- pooling and the ReLU nibble need no shuffle.  wave = (16 output channels
// w & 3, tiles (w >> 2), (w >> 2) + 2, ..).  Then flatten / concat / dropout / classifier / L2-norm (models.py:141-152):
// thread t owns the 16-byte group t of the head row (its dropout multipliers, its classifier weights), the eight waves'
// partial logits meet in LDS.
template <bool INFER = false>
__device__ __forceinline__ void conv3_fwd_tail_g(const Conv3Args& a, const Conv3Ctx& c, float* smem) {
  constexpr int NT = 512;
  const int tid = c.tid, lane = c.lane, wave = c.wave, net = c.net, sample = c.s0;
  const int H2 = c.H2, W2 = c.W2, PW2 = W2 + 2, H4 = H2 >> 1, W4 = W2 >> 1, P4 = H4 * W4;
  const int SF = 64 * P4, F = SF + FD, K = a.K, F4 = F >> 2;
  const float* img2 = c.wbuf;                // zero-bordered pooled conv1 map [(H2+2)*(W2+2)][CS]
  float* row = smem;                         // [F] head row (the conv1 image is dead); F <= 1792
  float* red = smem + 1792;                  // [8] + [8 waves][64 classes]
  float* part = red + 8;
  constexpr int PS2 = 36;
  const int NPX2 = (H2 + 2) * PW2, PLN = NPX2 * PS2;
  uint32_t* pl = (uint32_t*)(smem + 2560);   // three bf16 planes [pixel][PS2 dwords]
  float* xch = (float*)(pl + 3 * PLN);       // conv2's k-half exchange [4 channel groups][3 tiles][4][64] (behind the planes)
  const long long rs = (long long)net * a.n + sample;
  // conv2: wave = (16 output channels cog = w & 3, k-step ksh = w >> 2 of every tap's two), ALL (up to three) 16-row tiles:
  // a tap's three B fragments are loaded once and serve every tile; the two k-steps meet through LDS once, at the end
  const int cog = wave & 3, ksh = wave >> 2;
  const int j = lane & 15, kg = lane >> 4;
  const uint4* wq = (const uint4*)(a.w2f + (long long)net * a.w2f_ns) + ((2 * ksh + (kg >> 1)) * 6 + (cog >> 1)) * 64 +
                    (kg & 1) * 32 + 16 * (cog & 1) + j;
  constexpr int C2_AHEAD = 3;
  uint4 bq[C2_AHEAD + 1][3];
#pragma unroll
  for (int t0 = 0; t0 < C2_AHEAD; ++t0)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) bq[t0][pc] = wq[((t0 * 4) * 3 + pc) * 128];
  // ---- loads up front: the spectral row (threads 0..255), conv2's bias, this thread's dropout multipliers
  float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) y4 = *(const float4*)(a.yin + rs * FD + 4 * tid);
  const float bias2 = (a.b2 + (long long)net * a.p_ns)[cog * 16 + j];
  const float bcv = (a.bc + (long long)net * a.p_ns)[tid < K ? tid : 0];
  const float* wc = a.wc + (long long)net * a.p_ns;
  const bool grp = tid < F4;                 // this thread owns the 16-byte group tid of the row
  const int f0 = 4 * (grp ? tid : 0);
  const int dmode = (!a.train || a.dropout_p <= 0.f) ? 0 : (a.dropmask != nullptr ? 1 : 2);
  const float keep_scale = 1.0f / (1.0f - a.dropout_p);
  float4 dm4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (grp) {
    if (dmode == 1) {
      dm4 = *(const float4*)(a.dropmask + rs * F + f0);
    } else if (dmode == 2) {
      const unsigned long long gs = xsrc_global_sample(a.xs, sample);
      const float4 u = philox_uniform4(a.xs.seed, xsrc_step(a.xs), STREAM_DROPOUT + net, (gs * F + f0) >> 2);
      dm4.x = (u.x >= a.dropout_p) ? keep_scale : 0.f; dm4.y = (u.y >= a.dropout_p) ? keep_scale : 0.f;
      dm4.z = (u.z >= a.dropout_p) ? keep_scale : 0.f; dm4.w = (u.w >= a.dropout_p) ? keep_scale : 0.f;
      *(float4*)(a.dropgen + rs * F + f0) = dm4;
    }
  }
  __syncthreads();                           // img2 interior complete; every thread is done pooling from img
  if (tid < 256) *(float4*)(row + SF + 4 * tid) = y4;
  for (int it = tid; it < NPX2 * 16; it += NT) {   // the pooled map as three bf16 planes
    const int px = it >> 4, c4 = it & 15;
    const float4 v = *(const float4*)(img2 + (size_t)px * CS + 4 * c4);
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint32_t u0[4], u1[4], u2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u0[q] = __float_as_uint(x[q]);
      const float r1 = x[q] - __uint_as_float(u0[q] & 0xffff0000u);
      u1[q] = __float_as_uint(r1);
      u2[q] = __float_as_uint(r1 - __uint_as_float(u1[q] & 0xffff0000u));
    }
    uint32_t* d = pl + px * PS2 + 2 * c4;
    *(uint2*)(d) = make_uint2(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]));
    *(uint2*)(d + PLN) = make_uint2(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]));
    *(uint2*)(d + 2 * PLN) = make_uint2(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]));
  }
  {  // the spectral row's squared norm (waves 0..3 hold it)
    float ss = (y4.x * y4.x + y4.y * y4.y) + (y4.z * y4.z + y4.w * y4.w);
    ss = wave_sum(ss);
    if (lane == 0 && wave < 4) red[wave] = ss;
  }
  __syncthreads();
  auto mm = [](const uint4& x, const uint4& y, f32x4v cc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), cc, 0, 0, 0);
  };
  const int co = cog * 16 + j;
  // A rows of this lane, window-major: sub-pixel (j & 3) of window 4 t + (j >> 2) of tile t; windows past the map
  // multiply a real pixel's operands (results dropped)
  const uint32_t* ap0[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int wa = 4 * t + (j >> 2), wac = wa < P4 ? wa : 0;
    const int wph = (wac * ((65536 + W4 - 1) / W4)) >> 16;          // wac / W4 (a 32-bit division costs ~40 instructions)
    const int oh = 2 * wph + ((j >> 1) & 1), ow = 2 * (wac - wph * W4) + (j & 1);
    ap0[t] = pl + (size_t)((oh + 1) * PW2 + ow + 1) * PS2 + 4 * kg + 16 * ksh;
  }
  f32x4v acc0[3], acc1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) { acc0[t] = {0.f, 0.f, 0.f, 0.f}; acc1[t] = {0.f, 0.f, 0.f, 0.f}; }
  uint4 af[2][9];
  auto read_a = [&](int tap, uint4 (&x)[9]) {
    const int toff = ((tap / 3 - 1) * PW2 + (tap % 3 - 1)) * PS2;
#pragma unroll
    for (int t = 0; t < 3; ++t) {            // (all three tiles whatever P4: straight-line code; a tile past the map costs 18 MFMAs)
      const uint32_t* ap = ap0[t] + toff;
      x[3 * t] = *(const uint4*)(ap); x[3 * t + 1] = *(const uint4*)(ap + PLN); x[3 * t + 2] = *(const uint4*)(ap + 2 * PLN);
    }
  };
  read_a(0, af[0]);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap + C2_AHEAD < 9) {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        bq[(tap + C2_AHEAD) % (C2_AHEAD + 1)][pc] = wq[(((tap + C2_AHEAD) * 4) * 3 + pc) * 128];
    }
    if (tap + 1 < 9) read_a(tap + 1, af[(tap + 1) & 1]);
    const uint4 (&bcur)[3] = bq[tap % (C2_AHEAD + 1)];
    const uint4 (&acur)[9] = af[tap & 1];
    const uint4 b1 = bcur[0], b2 = bcur[1], b3 = bcur[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const uint4 a1 = acur[3 * t], a2 = acur[3 * t + 1], a3 = acur[3 * t + 2];
      acc0[t] = mm(a1, b3, acc0[t]); acc1[t] = mm(a2, b2, acc1[t]); acc0[t] = mm(a3, b1, acc0[t]);
      acc1[t] = mm(a1, b2, acc1[t]); acc0[t] = mm(a2, b1, acc0[t]); acc1[t] = mm(a1, b1, acc1[t]);
    }
    // pin the order: this tap's loads and reads first, then its eighteen MFMAs
    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);    // VMEM reads (none in the last taps: the group is then empty)
    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);    // DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);   // MFMA
  }
  STAMP(0, 8);
  // first classes' classifier weights of this thread's group (their round trip runs under the fold and the epilogue)
  constexpr int KC = 24;
  float4 wv[KC];
#pragma unroll
  for (int q = 0; q < KC; ++q) wv[q] = *(const float4*)(wc + (long long)(q < K ? q : K - 1) * F + f0);
  // fold the two k-steps: waves ksh = 1 hand (acc0 + acc1) of every tile to their partner
  if (ksh == 1) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xch[((cog * 3 + t) * 4 + r) * 64 + lane] = acc0[t][r] + acc1[t][r];
  }
  __syncthreads();
  if (ksh == 0) {
    // epilogue: lane (co, kg) holds window 4 t + kg of tile t, registers = its four sub-pixels (dh, dw) = (r >> 1, r & 1)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int win = 4 * t + kg;
      if (win < P4) {
        const int ph = (win * ((65536 + W4 - 1) / W4)) >> 16, pw = win - ph * W4;
        float r_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float res = img2[(size_t)((2 * ph + (r >> 1) + 1) * PW2 + 2 * pw + (r & 1) + 1) * CS + co];   // residual branch
          r_[r] = relu_nan(((acc0[t][r] + acc1[t][r]) + xch[((cog * 3 + t) * 4 + r) * 64 + lane]) + bias2 + res);
        }
        const float o = ((r_[0] + r_[1]) + (r_[2] + r_[3])) * 0.25f;
        const uint32_t nib = (relu_open(r_[0]) ? 1u : 0u) | (relu_open(r_[1]) ? 2u : 0u) | (relu_open(r_[2]) ? 4u : 0u) |
                             (relu_open(r_[3]) ? 8u : 0u);
        if constexpr (!INFER) {
          a.p2out[(rs * P4 + win) * 64 + co] = o;
          a.m2out[(rs * P4 + win) * 64 + co] = (uint8_t)nib;
        }
        row[co * P4 + win] = o;              // canonical flatten order f = c * P4 + hw (x.view, models.py:141)
      }
    }
  }
  __syncthreads();                           // row[0 .. SF) complete
  STAMP(0, 9);
  if constexpr (!INFER) {
    if (a.feat != nullptr) {                 // (null: feat_norm_kernel formed the embeddings behind the spectral branch)
      const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
      if (tid == 0) a.ynorm[rs] = norm;
      if (tid < 256) {
        float4 o = y4;
        o.x /= norm; o.y /= norm; o.z /= norm; o.w /= norm;
        *(float4*)(a.feat + rs * FD + 4 * tid) = o;
      }
    }
  }
  float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (grp) {
    x4 = *(const float4*)(row + f0);
    if (dmode != 0) { x4.x *= dm4.x; x4.y *= dm4.y; x4.z *= dm4.z; x4.w *= dm4.w; }
    if constexpr (!INFER) *(float4*)(a.catd + rs * F + f0) = x4;
  }
  // logits: KC classes at a time, every weight of a chunk requested before the first is used
  for (int kc = 0; kc < K; kc += KC) {
    if (kc > 0) {
#pragma unroll
      for (int q = 0; q < KC; ++q) wv[q] = *(const float4*)(wc + (long long)(kc + q < K ? kc + q : K - 1) * F + f0);
    }
    float t[KC];                             // all reductions of a chunk in one basic block, step by step (see conv3_fwd_tail)
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      const float acc = (x4.x * wv[q].x + x4.y * wv[q].y) + (x4.z * wv[q].z + x4.w * wv[q].w);
      t[q] = grp ? acc : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < KC; ++q) t[q] += __shfl_xor(t[q], o, 64);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < KC; ++q)
        if (kc + q < K) part[wave * 64 + kc + q] = t[q];
    }
  }
  __syncthreads();
  if (tid < K) {
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sacc += part[w * 64 + tid];
    const float lg = sacc + bcv;
    if (!INFER || a.logits != nullptr) a.logits[rs * K + tid] = lg;
    if constexpr (INFER) red[520 + tid] = lg;                         // (behind the partials: red[8 + 512 ..])
  }
  if constexpr (INFER) {
    __syncthreads();
    if (tid == 0) a.labels_out[rs] = logits_argmax(red + 520, K);
  }
}

// Head backward + conv2 data gradient of the eight-tile kernels (see conv3_bwd_head for the math).  Thread t owns the
// 16-byte group t of the head row; the conv2 data gradient covers up to 64 pooled pixels: wave = (pixel tile w >> 2,
// input-channel tile (w >> 1) & 1, half (w & 1) of every tap's 64 output channels).
__device__ __forceinline__ const float* conv3_bwd_head_g(const Conv3Args& a, float* smem, uint32_t (&mpre)[2]) {
  constexpr int NT = 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int net, sample;
  wg_decode(a, net, sample);
  const int H2 = a.H >> 1, W2 = a.W >> 1, P2 = H2 * W2, PW2 = W2 + 2;
  const int H4 = H2 >> 1, W4 = W2 >> 1, P4 = H4 * W4;
  const int SF = 64 * P4, F = SF + FD, K = a.K, F4 = F >> 2;
  const int PW = a.W + 2, IMG = (a.H + 2) * PW;
  float* img2 = smem + (size_t)IMG * CS;                           // = wbuf: dz2 image [(H2+2)*(W2+2)][CS] <= WBUF floats
  float* ext = img2 + WBUF + 256;                                  // behind the LUT (256 entries)
  float* dp1s = ext;                                               // [P2][64]
  float* dp2s = dp1s + P2 * 64;                                    // [P4][64]  pooled conv2 gradient [hw][c]
  float* dls = dp2s + P4 * 64;                                     // [64]
  float* red = dls + 64;                                           // [8]
  const long long rs = (long long)net * a.n + sample;
  const int l31 = lane & 31, hh = lane >> 5;
  const int mtile = wave >> 2, nt = (wave >> 1) & 1, kh2 = wave & 1;
#pragma unroll
  for (int q = 0; q < 2; ++q) {       // conv1's ReLU-mask words of this thread's staging items
    const int idx = tid + NT * q;
    mpre[q] = (idx < P2 * 16) ? *(const uint32_t*)(a.mask_in + (long long)net * a.mask_in_ns + ((size_t)sample * P2 + (idx >> 4)) * 64 + (idx & 15) * 4) : 0u;
  }
  const uint4* wq = (const uint4*)(a.w2d + (long long)net * a.w2d_ns) + lane;
  constexpr int C2_AHEAD = 3;
  if (tid < 64) dls[tid] = (tid < K) ? a.dlogits[rs * K + tid] : 0.f;
  const bool grp = tid < F4;
  const int f0 = 4 * (grp ? tid : 0);
  const bool spec = grp && f0 >= SF;                                // (SF is a multiple of 4: a group is all spatial or all spectral)
  const int j0 = spec ? f0 - SF : 0;
  const float norm = a.ynrm[rs];
  float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f), df4 = y4;
  if (spec) {
    y4 = *(const float4*)(a.yin + rs * FD + j0);
    if (a.dfeat != nullptr) df4 = *(const float4*)(a.dfeat + rs * FD + j0);
  }
  const uint32_t m2pre = (tid < P4 * 16) ? *(const uint32_t*)(a.m2in + (rs * P4 + (tid >> 4)) * 64 + (tid & 15) * 4) : 0u;
  const float* wc = a.wc + (long long)net * a.p_ns;
  float4 dm4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (a.hmask != nullptr && grp) dm4 = *(const float4*)(a.hmask + rs * F + f0);
  constexpr int KC = 24;                                            // classes per chunk: every weight of a chunk in flight at once
  float4 wv[KC];
#pragma unroll
  for (int q = 0; q < KC; ++q) wv[q] = *(const float4*)(wc + (long long)(q < K ? q : K - 1) * F + f0);
  {  // while those loads fly: zero-bordered dz2 image, conv1 image zero fill, output-pixel LUT
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < WBUF / 4; i += NT) ((float4*)img2)[i] = z4;
    for (int i = tid; i < IMG * (CS / 4); i += NT) ((float4*)smem)[i] = z4;
    int* lut = (int*)(img2 + WBUF);
    const int HWl = a.H * a.W;
    for (int m = tid; m < 256; m += NT) {
      const int mm = (m < HWl) ? m : 0;
      const int r = mm / a.W, cc = mm - r * a.W;
      lut[m] = (r + 1) * PW + (cc + 1);
    }
  }
  float dot = (y4.x / norm) * df4.x + (y4.y / norm) * df4.y + (y4.z / norm) * df4.z + (y4.w / norm) * df4.w;
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kc = 0; kc < K; kc += KC) {
    if (kc > 0) {
#pragma unroll
      for (int q = 0; q < KC; ++q) wv[q] = *(const float4*)(wc + (long long)(kc + q < K ? kc + q : K - 1) * F + f0);
    }
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      const float dl = (kc + q < K) ? dls[kc + q] : 0.f;
      dc.x = fmaf(dl, wv[q].x, dc.x); dc.y = fmaf(dl, wv[q].y, dc.y);
      dc.z = fmaf(dl, wv[q].z, dc.z); dc.w = fmaf(dl, wv[q].w, dc.w);
    }
  }
  if (grp) {
    const float d[4] = {dc.x * dm4.x, dc.y * dm4.y, dc.z * dm4.z, dc.w * dm4.w};
    if (!spec) {                                                     // spatial part: f = c * P4 + hw
      float* dp2g = a.dp2out + rs * SF;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = f0 + e, cc = (f * ((65536 + P4 - 1) / P4)) >> 16, hw = f - cc * P4;    // f / P4, f < 64 P4 <= 768
        dp2s[hw * 64 + cc] = d[e];
        dp2g[hw * 64 + cc] = d[e];
      }
    } else {
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, dv[4] = {df4.x, df4.y, df4.z, df4.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = d[e];
        if (a.dfeat != nullptr) g += (dv[e] - (yv[e] / norm) * dot) / norm;
        o[e] = relu_open(yv[e]) ? g : 0.f;
      }
      *(float4*)(a.dy + rs * FD + j0) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  STAMP(1, 10);
  // the conv2 data gradient's first fragment sets (requested here: their round trip runs under the dz2 staging and its
  // barriers; at kernel start they would sit in registers through the head's arithmetic)
  uint4 bq[C2_AHEAD + 1][6];
#pragma unroll
  for (int t0 = 0; t0 < C2_AHEAD; ++t0)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) bq[t0][3 * q + pc] = wq[(size_t)t0 * TAPW + (((kh2 * 2 + q) * 3 + pc) * 2 + nt) * 64];
  __syncthreads();                                                 // dp2s complete, img2 zeroed
  STAMP(1, 4);
  uint32_t zmx = 0u;
  if (tid < P4 * 16) {   // dz2 = mask2 * upsample(dp2) / 4: (pooled pixel, 4 channels) -> its 2x2 window
    const int c4 = tid & 15, pp = tid >> 4, ph = pp / W4, pw = pp - ph * W4;
    const float4 d = *(const float4*)(dp2s + pp * 64 + c4 * 4);
    const uint32_t m = m2pre;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      float4 v;
      v.x = ((m >> sub) & 1u) ? d.x * 0.25f : 0.f;
      v.y = ((m >> (8 + sub)) & 1u) ? d.y * 0.25f : 0.f;
      v.z = ((m >> (16 + sub)) & 1u) ? d.z * 0.25f : 0.f;
      v.w = ((m >> (24 + sub)) & 1u) ? d.w * 0.25f : 0.f;
      const int h = 2 * ph + (sub >> 1), w = 2 * pw + (sub & 1);
      *(float4*)(img2 + (size_t)((h + 1) * PW2 + w + 1) * CS + c4 * 4) = v;
      zmx = mag_max4(zmx, v);
    }
  }
  // (two-piece weight gradient: this sample's largest |dz2| -- whole waves take part in the reduction; the second word of
  //  the statistics slot was zeroed at kernel start, in front of the barrier above)
  if (a.hstat != nullptr && (tid >> 6) * 64 < P4 * 16) h2_publish_max(zmx, smem + a.maxslot + 1);
  __syncthreads();
  if (a.hstat != nullptr && tid == 0) a.hstat[(3 * 2 + net) * a.n + sample] = ((const volatile uint32_t*)smem)[a.maxslot + 1];
  STAMP(1, 5);
  // ---- conv2 data gradient (split-bf16, 32x32x16): this wave's pixel tile x input-channel tile x output-channel half
  const int pA = mtile * 32 + l31;
  const int mg2 = (65536 + W2 - 1) / W2;                           // p / W2 == (p * mg2) >> 16 (p W2 < 65536)
  const int phA = (pA * mg2) >> 16;
  const int posA = pA < P2 ? (phA + 1) * PW2 + (pA - phA * W2) + 1 : 0;
  const float* ap = img2 + (size_t)posA * CS + kh2 * 32 + hh * 8;
  f32x16 acc = zero16();
  auto raw_ptr = [&](int u) {
    const int tap = u >> 1;
    const int toff = ((tap / 3 - 1) * PW2 + (tap % 3 - 1)) * CS;
    return ((pA < P2) ? ap + toff : ap) + 16 * (u & 1);
  };
  ASplit cur;
  float4 rn0, rn1;
  {
    const float* p0 = raw_ptr(0);
    a_split(*(const float4*)p0, *(const float4*)(p0 + 4), cur.p1, cur.p2, cur.p3);
    const float* p1 = raw_ptr(1);
    rn0 = *(const float4*)p1; rn1 = *(const float4*)(p1 + 4);
  }
#pragma unroll
  for (int u = 0; u < 18; ++u) {
    const int tap = u >> 1, q = u & 1;
    if (q == 0 && tap + C2_AHEAD < 9) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bq[(tap + C2_AHEAD) % (C2_AHEAD + 1)][3 * q2 + pc] =
              wq[(size_t)(tap + C2_AHEAD) * TAPW + (((kh2 * 2 + q2) * 3 + pc) * 2 + nt) * 64];
    }
    if (q == 0) __builtin_amdgcn_sched_barrier(0);
    float4 rnn0 = rn0, rnn1 = rn1;
    if (u + 2 < 18) {
      const float* p2 = raw_ptr(u + 2);
      rnn0 = *(const float4*)p2; rnn1 = *(const float4*)(p2 + 4);
    }
    const uint4 (&bcur)[6] = bq[tap % (C2_AHEAD + 1)];
    acc = mfma_b3(cur.p1, cur.p2, cur.p3, bcur[3 * q], bcur[3 * q + 1], bcur[3 * q + 2], acc);
    ASplit nxt = cur;
    if (u + 1 < 18) a_split(rn0, rn1, nxt.p1, nxt.p2, nxt.p3);
    if (u + 2 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    SchedInterleave<6>::run();
    cur = nxt; rn0 = rnn0; rn1 = rnn1;
  }
  STAMP(1, 6);
  {
    const int ci = nt * 32 + l31;
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = mtile * 32 + acc_row(r, lane), ph = (p * mg2) >> 16;
      res[r] = (kh2 == 0 && p < P2) ? img2[(size_t)((ph + 1) * PW2 + (p - ph * W2) + 1) * CS + ci] : 0.f;
    }
    __syncthreads();                                               // every wave has finished reading img2
    STAMP(1, 7);
    float* xch = img2 + (mtile * 2 + nt) * 1024;                   // [16][64] per (pixel tile, channel tile)
    if (kh2 == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    STAMP(1, 8);
    if (kh2 == 0) {
      float* dp1g = a.dp1out + rs * (long long)P2 * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = mtile * 32 + acc_row(r, lane);
        if (p < P2) {
          const float v = (acc[r] + xch[r * 64 + lane]) + res[r];
          dp1s[p * 64 + ci] = v;
          dp1g[p * 64 + ci] = v;
        }
      }
    }
  }
  __syncthreads();                                                 // dp1s complete; img2 (= wbuf) is free again
  STAMP(1, 11);
  return dp1s;
}

// (the per-sample kernels, MODE >= 2, count on two waves per SIMD, at most 256 registers: two four-wave workgroups per
//  CU, or -- NW = 8 -- one eight-wave workgroup that has the CU to itself)
//  TPW = pixel tiles per wave of the tap loop: NW * TPW / 2 tiles of 32 pixels -- 4 (windows up to 128 pixels: NW = 4 with
//  TPW = 2, or NW = 8 with TPW = 1) or 8 (NW = 8, TPW = 2: windows up to 256 pixels, e.g. BASELINE configs[4]'s 15 x 15)
//  KSG (MODE 0 / 1, one sample per workgroup, at most four pixel tiles): the GENERAL forward / data-gradient kernel with
//  the per-sample kernels' barrier-free tap loop (conv3_taps_ks: wave = (pixel half, channel half), fragments L2 ->
//  registers a tap ahead, the halves folded through LDS) instead of the LDS-staged weights with two workgroup barriers
//  per tap -- what conv2 takes on windows the fused kernels do not reach (the reference's 20 x 20: 10 x 10 maps).
//  H2X (MODE 2 / 3 with tail / head, four waves): conv1's tap loop on TWO fp16 pieces (conv3_taps_ks_h) whenever the
//  sample's image and the network's weights are inside the ranges that scheme needs -- else, workgroup-uniformly, the
//  three-piece bf16 loop, which has no range conditions.
template <int MODE, int MTW, int TAIL = 0, int NW = 4, int TPW = 8 / NW, bool KSG = false, bool H2X = false>
__global__ __launch_bounds__(64 * NW, ((MODE >= 2 || KSG) && NW == 4 ? 2 : 1)) void conv3x3_kernel(Conv3Args a) {
  static_assert(!KSG || (MODE < 2 && MTW == 1 && TAIL == 0 && NW == 4 && TPW == 2), "KSG: four waves, S = 1, four tiles");
  static_assert(!H2X || (MODE >= 2 && TAIL == 1 && (NW == 8 || TPW == 2)) || (MODE < 2 && TAIL == 0 && (KSG || NW == 8)),
                "H2X: the per-sample kernels with tail / head; the general kernels with one sample per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if constexpr (H2X) { if (threadIdx.x == 0) { smem[a.maxslot] = 0.f; smem[a.maxslot + 1] = 0.f; } }
  constexpr int NT = 64 * NW;
  constexpr int KMT = NW * TPW / 2;        // pixel tiles of the per-sample kernels
  constexpr int LUTN = (MODE >= 2) ? KMT * 32 : MTW * NW * 32;
  constexpr bool BIG = (KMT == 8);         // the generalised tail / head (final maps up to 12 pooled pixels)
  constexpr bool INFER = (MODE == 2 && TAIL == 2);   // forward from the scene cube, eval, argmax out: nothing kept for a backward
  if constexpr (INFER) { if (wg_infer_sample(a) < 0) return; }
  if (CMLPL_ABL == 26) return;             // ablation: the launch itself (grid, LDS allocation, end of kernel) and nothing else
  Conv3Ctx c;
  STAMPG(MODE & 1, 0);
  const float* dp_lds = nullptr;
  uint32_t mpre[2] = {0u, 0u};
  if constexpr (MODE == 3 && TAIL) {
    if constexpr (BIG) dp_lds = conv3_bwd_head_g(a, smem, mpre);
    else dp_lds = conv3_bwd_head<NW>(a, smem, mpre);
  }
  conv3_stage<MODE, NW, TPW, INFER, H2X>(a, smem, LUTN, c, dp_lds, mpre);
  STAMPG(MODE & 1, 1);
  const int tid = c.tid, lane = c.lane, l31 = c.l31, hh = c.hh, wave = c.wave, net = c.net, s0 = c.s0;
  const int HW = c.HW, PW = c.PW, S = c.S, PX = c.PX, npx = c.npx;
  float* img = c.img; float* wbuf = c.wbuf; int* lut = c.lut; const float4* wg = c.wg;
  const TapRegs wp = c.wp;
  const int MT = (npx + 31) >> 5;
  int abase[MTW];
  f32x16 acc[MTW][2];
  // per-sample fused kernels (S == 1, MTW == 1): the barrier-free tap loop, see conv3_taps_ks
  constexpr bool KS = (MODE >= 2) || KSG;
  // General kernels: slot t of wave w holds pixel tile w + NW t.  Eight waves sit two to a SIMD (w and w + 4), so when
  // the last round has 1 or 5 tiles, SIMD 0 carries one tile more than the others (20 x 20 windows: 13 tiles = 4, 3, 3,
  // 3): that round's last tile is then SHARED by its wave (output-channel tile 0) and the next wave (channel tile 1, in
  // the slot that would be empty) -- 3.5, 3.5, 3, 3.  nmask = which channel tiles of the last slot this wave computes.
  const int RL = MT - NW * (MTW - 1);
  const bool nsplit = !KS && NW == 8 && MTW >= 2 && (RL == 1 || RL == 5);
  int last_tile = wave + NW * (MTW - 1), nmask = 3;
  if (nsplit && wave == RL - 1) nmask = 1;
  else if (nsplit && wave == RL) { last_tile = MT - 1; nmask = 2; }
  auto slot_tile = [&](int t) { return t == MTW - 1 ? last_tile : wave + NW * t; };
  if constexpr (!KS) {
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      abase[t] = lut[slot_tile(t) * 32 + l31] * CS + 8 * hh;
      acc[t][0] = zero16();
      acc[t][1] = zero16();
    }
  }
  int ab2[TPW];
  f32x16 acc2[TPW][2];
  if (KS) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      ab2[t] = lut[(TPW * (wave >> 1) + t) * 32 + l31] * CS + 8 * hh;
      acc2[t][0] = zero16();
      acc2[t][1] = zero16();
    }
  }
  const uint4* wq = (const uint4*)wg + lane;
  // H2X: the two-piece loop is taken when the network's weights fit fp16 at the packing scale (flag word, set by the
  // packing kernels) and the image's largest magnitude is an ordinary number: scale 2^(14 - floor(log2 max)) puts it in
  // [2^14, 2^15); the folded accumulators are multiplied by hinv = 1 / (scale 2^H2_WEXP).  (Both exact powers of two.)
  // An image that is ZERO everywhere (the data gradient of a row no loss term reaches: a masked-out unlabelled row -- a
  // third of a step's workgroups in the backward) meets finite weights (flag clear): its products are exact zeros on
  // either loop; the backward skips the tap loop (hzero), the forward runs the two-piece one at scale 1.
  const uint4* wq16 = nullptr;
  bool h2on = false, hzero = false;
  float hsc = 1.f, hinv = 1.f;
  auto h2_decide = [&]() {
    if constexpr (H2X) {
      const uint32_t mxb = __builtin_amdgcn_readfirstlane(((const volatile uint32_t*)smem)[a.maxslot]);
      const uint32_t e = mxb >> 23;
      const uint32_t flag = a.h2flag[(long long)net * a.h2flag_ns];
      hzero = flag == 0u && mxb == 0u && (MODE & 1) != 0;      // (a zero GRADIENT image: its products are skipped)
      h2on = flag == 0u && ((e >= 40u && e <= 200u) || mxb == 0u);
      hsc = mxb == 0u ? 1.f : __uint_as_float((268u - e) << 23);
      hinv = mxb == 0u ? 1.f : __uint_as_float((e - 27u - (uint32_t)(H2_WEXP - 13)) << 23);
      wq16 = (const uint4*)(a.wpk16 + (long long)net * a.wpk16_ns) + lane;
      if (a.hstat != nullptr && tid == 0) a.hstat[((MODE == 2 ? 0 : MODE == 3 ? 2 : a.hkind) * 2 + net) * a.n + s0] = mxb;
    }
  };
  float* x8 = (float*)(lut + LUTN);        // eight waves: [8][16][64] floats behind the LUT (second tap buffer, then the fold's exchange)
  const bool ks_active = (wave >> 1) * TPW * 32 < npx;
  // What a wave of the per-sample kernels owns after the fold of the two channel halves -- and, in the forward, already
  // in the conv0 stage: TPW == 2: pixel tile `wave`, both channel tiles; TPW == 1: pixel tile wave >> 1, channel tile
  // wave & 1.  own[i] = channel tile nt0 + i of pixel tile `ot`.
  const int ot = (TPW == 2) ? wave : (wave >> 1), nt0 = (TPW == 2) ? 0 : (wave & 1);
  f32x16 own[TPW];

  // tiles t <= MTW-2 are always active; only the last one may be missing for some waves (wave-uniform)
  STAMPG(MODE & 1, 14);
  if constexpr (INFER) {
    if constexpr (NW == 8) conv3_taps_lds8<TPW>(img, wbuf, x8, wg, c.wp8, ab2, acc2, PW, tid, wave, lane, ks_active);
    else conv3_taps_ks<TPW>(img, wq, ab2, acc2, PW, wave, ks_active);
  } else if (MODE == 2) {
    // a0 goes to HBM (the backward pass reads it) from the LDS image itself, which is read-only during the tap
    // loop: tap s copies items NT s + tid (pixel, 16-byte channel chunk) -- coalesced 16-byte stores, no registers
    // held across the loop.
    float* a0g = a.a0out + ((long long)net * a.n + s0) * (long long)HW * 64;
    const int magicW = (65536 + c.W - 1) / c.W;          // m / W == (m * magic) >> 16 for every m of the map (checked on the host)
    auto side = [&](int s) {
      const int idx = s * NT + tid, m = idx >> 4, c4 = idx & 15;
      if (s < KMT * 512 / NT && m < HW) {
        const int h = (m * magicW) >> 16, w = m - h * c.W;
        *(float4*)(a0g + (size_t)m * 64 + c4 * 4) = *(const float4*)(img + (size_t)((h + 1) * PW + w + 1) * CS + c4 * 4);
      }
    };
    // (eight waves on two fp16 pieces: a wave's eight fragments per tap straight from L2 as in the four-wave loop -- per CU
    //  the same 64 KiB per tap as two four-wave workgroups pull; the LDS copy of tap 0 staged above is then not used)
    if constexpr (H2X) h2_decide();
    if (h2on) conv3_taps_ks_h<TPW>(img, wq16, ab2, acc2, PW, wave, ks_active, hsc, side);
    else if constexpr (NW == 8) conv3_taps_lds8<TPW>(img, wbuf, x8, wg, c.wp8, ab2, acc2, PW, tid, wave, lane, ks_active, side);
    else conv3_taps_ks<TPW>(img, wq, ab2, acc2, PW, wave, ks_active, side);
  } else if constexpr (KS) {
    __syncthreads();   // the staged image is complete (conv3_taps has this barrier in front of its first tap)
    if constexpr (H2X) h2_decide();
    if (h2on) { if (!hzero || a.h2_noskip) conv3_taps_ks_h<TPW>(img, wq16, ab2, acc2, PW, wave, ks_active, hsc); }
    else if constexpr (NW == 8) conv3_taps_lds8<TPW>(img, wbuf, x8, wg, c.wp8, ab2, acc2, PW, tid, wave, lane, ks_active);
    else conv3_taps_ks<TPW>(img, wq, ab2, acc2, PW, wave, ks_active);
  }
  else if constexpr (NW == 8) {   // the general kernels with eight waves (one workgroup per CU: 20 x 20 windows): two waves per SIMD
    if constexpr (H2X) { __syncthreads(); h2_decide(); }      // (the staged image's maximum is complete behind this barrier)
    if (h2on) {
      if (!hzero || a.h2_noskip) {
        const float4* wg16 = (const float4*)(a.wpk16 + (long long)net * a.wpk16_ns);
        if (last_tile >= MT) conv3_taps_h<MTW, MTW - 1>(img, wbuf, wg16, c.wp8h, abase, acc, PW, tid, lane, hsc);
        else if (nmask == 3) conv3_taps_h<MTW, MTW>(img, wbuf, wg16, c.wp8h, abase, acc, PW, tid, lane, hsc);
        else if constexpr (MTW >= 2) {
          if (nmask == 1) conv3_taps_h<MTW, MTW, 1>(img, wbuf, wg16, c.wp8h, abase, acc, PW, tid, lane, hsc);
          else            conv3_taps_h<MTW, MTW, 2>(img, wbuf, wg16, c.wp8h, abase, acc, PW, tid, lane, hsc);
        }
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[t][0][r] *= hinv; acc[t][1][r] *= hinv; }
      }
    }
    else if (last_tile >= MT) conv3_taps<MTW, MTW - 1, NoSide, 1, TapRegs8>(img, wbuf, wg, c.wp8, abase, acc, PW, tid, lane);
    else if (nmask == 3) conv3_taps<MTW, MTW, NoSide, 1, TapRegs8>(img, wbuf, wg, c.wp8, abase, acc, PW, tid, lane);
    else if constexpr (MTW >= 2) {
      if (nmask == 1) conv3_taps<MTW, MTW, NoSide, 1, TapRegs8, 1>(img, wbuf, wg, c.wp8, abase, acc, PW, tid, lane);
      else            conv3_taps<MTW, MTW, NoSide, 1, TapRegs8, 2>(img, wbuf, wg, c.wp8, abase, acc, PW, tid, lane);
    }
  }
  else if (wave + 4 * (MTW - 1) < MT) conv3_taps<MTW, MTW>(img, wbuf, wg, wp, abase, acc, PW, tid, lane);
  else                           conv3_taps<MTW, MTW - 1>(img, wbuf, wg, wp, abase, acc, PW, tid, lane);
  STAMPG(MODE & 1, 15);
  __syncthreads();  // all MFMA reads of img are done; the epilogue overwrites it in place
  // (eight waves: the exchange needs [8][16][64] floats = 32 KiB, more than the tap-weight buffer: its own region behind
  //  the LUT -- where the backward's head kept its hand-off buffers, dead since the staging; one workgroup per CU, LDS
  //  is plentiful: conv3_ks8_lds)
  if constexpr (KS) conv3_ks_fold<TPW>(acc2, own, NW == 8 ? x8 : wbuf, wave, lane);
  if constexpr (H2X) {
    if (h2on) {
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) own[i][r] *= hinv;
    }
  }
  STAMPG(MODE & 1, 2);

  if (!(MODE & 1)) {
    const float* bias = a.bias + (long long)net * a.bias_ns;
    const float bv0 = bias[l31], bv1 = bias[32 + l31];
    if constexpr (KS) {
      // this wave's tile `ot`, channel tiles nt0 .. nt0 + TPW - 1: three passes -- positions, residual reads, writes
      if (ot < MT) {
        int pos[16];
        float xr[TPW][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = ot * 32 + acc_row(r, lane);
          pos[r] = lut[m < npx ? m : 0] * CS + 32 * nt0 + l31;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < TPW; ++i) xr[i][r] = img[pos[r] + 32 * i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = ot * 32 + acc_row(r, lane);
          if (m < npx) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
              img[pos[r] + 32 * i] = relu_nan(own[i][r] + ((nt0 + i) ? bv1 : bv0) + xr[i][r]);
          }
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int tile = slot_tile(t);
      const int nm = (t == MTW - 1) ? nmask : 3;
      if (tile < MT) {
        // three passes -- positions, residual reads, writes -- so that the 16 rows' LDS round trips overlap
        // (a row-by-row loop is a chain of dependent lut -> read -> write trips: 6 us per workgroup here)
        int pos[16];
        float x0[16], x1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tile * 32 + acc_row(r, lane);
          pos[r] = lut[m < npx ? m : 0] * CS;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { x0[r] = img[pos[r] + l31]; x1[r] = img[pos[r] + 32 + l31]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tile * 32 + acc_row(r, lane);
          if (m < npx) {
            if (nm & 1) img[pos[r] + l31] = relu_nan(acc[t][0][r] + bv0 + x0[r]);
            if (nm & 2) img[pos[r] + 32 + l31] = relu_nan(acc[t][1][r] + bv1 + x1[r]);
          }
        }
      }
    }
    }
    if (TAIL) {   // the tap-weight buffer is dead: it becomes the zero-bordered conv2 input image
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = tid; i < (BIG ? WBUF / 4 : 1024); i += NT) ((float4*)wbuf)[i] = z4;
    }
    STAMP(0, 10);
    __syncthreads();
    STAMP(0, 11);
    // (H2X: the pooled map's largest magnitude goes into the second statistics word of the LDS slot -- dead since the tap
    //  loop was chosen -- and from there, behind the tail's barriers, into the batch-level word the weight gradient reads)
    conv3_pool_store<NT, !INFER, H2X>(a, c, TAIL ? wbuf : nullptr, (H2X && a.hstat != nullptr) ? smem + a.maxslot + 1 : nullptr);
    STAMP(0, 7);
    if constexpr (TAIL != 0) {
      if constexpr (BIG) conv3_fwd_tail_g<INFER>(a, c, smem);
      else conv3_fwd_tail<NW, INFER>(a, c, smem);
    }
    if constexpr (H2X) {
      if (a.hstat != nullptr && tid == 0) a.hstat[(1 * 2 + net) * a.n + s0] = ((const volatile uint32_t*)smem)[a.maxslot + 1];
    }
  } else if (MODE == 3) {
    // conv0 weight gradient fused in (S == 1, MTW == 1):  dW0[c][co] = sum_pix xn[c][pix] * da0[pix][co].
    // This wave's da0 tile = accumulators + dz (the residual branch); it goes to LDS as the B operand [pix][64]
    // next to the sample's input slab, which global_load_lds_dwordx4 copies linearly into the now dead image
    // region.  wave = band tile (A rows c = 32 wave + l31, conflict-free: row stride HW is odd or the reads are b32
    // over consecutive c), both co tiles.  The partial has conv0_wgrad_kernel's layout.  Up to a.bp bands fit beside
    // da0 (two workgroups per CU): more bands (B4: 200) take further PASSES over [bp][HW] ranges of the slab through
    // the same region, da0 staying where it is.
    const int C = a.C, BP = a.bp;
    float vv[TPW][16];                                    // da0 of tile `ot`, channel tiles nt0 ..
    {
      int pos[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ot * 32 + acc_row(r, lane);
        pos[r] = lut[m < PX ? m : 0] * CS + 32 * nt0 + l31;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < TPW; ++i) vv[i][r] = img[pos[r] + 32 * i];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < TPW; ++i) vv[i][r] += own[i][r];
    }
    __syncthreads();                                      // image and LUT are dead from here on
    float* slab = smem;                                   // [min(C, BP)][HW]
    float* dal = smem + (C < BP ? C : BP) * HW;           // [HW + 1][64]
    const int Cw = conv0_partial_rows(C);                 // rows of the partial (kernels.hpp)
    float* pp = a.part0 + (long long)net * a.part0_ns + (size_t)s0 * ((size_t)Cw * 64 + 64);
    // a sample whose gradient image was zero everywhere (hzero) has da0 = 0: its partial is zero -- neither the slab nor
    // the MFMAs are needed, the stores below write the zeros
    // (not when the sample's activations were not finite -- the forward left their largest magnitude: 0 times a
    //  non-finite input is NaN in fp32, and stays so)
    bool zskip = false;
    if constexpr (H2X) {
      if (a.hstat != nullptr) zskip = hzero && a.h2_noskip == 0 && (a.hstat[(0 * 2 + net) * a.n + s0] >> 23) < 255u;
    }
    SlabRange rg = slab_range(a.xs, net, s0, C * HW, 0, (C < BP ? C : BP) * HW);
    if (!zskip) slab_issue<NW>(rg, slab, wave, lane);     // the forward's input again (first pass)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = ot * 32 + acc_row(r, lane);
      if (m < HW) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) dal[m * 64 + 32 * (nt0 + i) + l31] = vv[i][r];
      }
    }
    if (tid < 64) dal[HW * 64 + tid] = 0.f;               // the pixel past the end of an odd map
    STAMP(1, 12);
#pragma unroll 1
    for (int cb = 0; cb < C; cb += BP) {                  // uniform; one pass unless C > BP
      const int nb = (C - cb < BP) ? C - cb : BP;
      // (an opaque copy of HW: left alone, the compiler hoists the ~64 per-lane operand offsets of the MFMA block out
      // of this loop and keeps them in registers across it -- the kernel then no longer fits two waves per SIMD)
      int HWl = HW;
      asm volatile("" : "+s"(HWl));
      if (cb > 0) {
        __syncthreads();                                  // every wave is done reading the previous pass's rows
        rg = slab_range(a.xs, net, s0, C * HW, (cb * HW) >> 2, nb * HW);   // (BP * HW is a multiple of 4)
        if (!zskip) slab_issue<NW>(rg, slab, wave, lane);
      }
      if (!zskip) slab_tail(rg, slab, tid);
      if (cb == 0) STAMP(1, 13);
      __syncthreads();                                    // range landed (the barrier waits for the DMA), da0 complete
      // dW0 tile of this wave on the split-bf16 MFMA: k-steps of 16 pixels; lane (row = band c, half h) takes pixels
      // 16 kq + 8h .. + 7 of its slab row, lane (col = co, half h) the same pixels of da0; both are split in registers
      // (132 VALU instructions per 12 MFMAs).  Pixels >= HW of the last step read the zero row behind da0.
      if constexpr (BIG) {
        // Eight waves, up to eight band tiles -- but BASELINE configs[4] has 48 bands: with wave = band tile six waves
        // would idle while two walk fifteen k-steps.  By the band tiles of this pass (uniform):
        //   <= 2 tiles: wave = (band tile w >> 2, co tile (w >> 1) & 1, k-steps of parity w & 1), the two parities meet in LDS
        //   <= 4 tiles: wave = (band tile w >> 1, co tile w & 1)
        //   else      : wave = band tile, both co tiles
        const int NBT = (nb + 31) >> 5;
        const int NS = (HWl + 15) >> 4;
        float* xw = dal + (size_t)(HWl + 1) * 64;           // [8 waves][16][64] + [8][64]: the k-parity exchange (behind da0)
        auto wtile = [&](auto nct_c, int btile, int cn0, int kq0, int kqs, bool pair) {
          constexpr int NCT = decltype(nct_c)::value;
          const bool has_tile = btile * 32 < nb;
          const bool has_db = cb == 0 && btile == 0;        // the waves of band tile 0 also sum the bias gradient of their co tile(s)
          f32x16 g[NCT];
#pragma unroll
          for (int i = 0; i < NCT; ++i) g[i] = zero16();
          float dbacc[NCT];
#pragma unroll
          for (int i = 0; i < NCT; ++i) dbacc[i] = 0.f;
          if (has_tile || has_db) {
            const int crow = btile * 32 + l31;
            const float* ap = slab + (size_t)crow * HWl + 8 * hh;
            const float* bp = dal + 32 * cn0 + l31;
#pragma unroll 1
            for (int kq = kq0; kq < NS; kq += kqs) {        // (rolled: three variants of sixteen unrolled k-steps spilled addresses)
              if (!zskip) {
                float ra[8], rb[NCT][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int k = kq * 16 + 8 * hh + j;
                  const int kc = k < HWl ? k : HWl;         // the zero row
                  ra[j] = ap[kq * 16 + j];
#pragma unroll
                  for (int i = 0; i < NCT; ++i) rb[i][j] = bp[kc * 64 + 32 * i];
                }
                uint4 A1, A2, A3, P1, P2, P3;
                a_split(make_float4(ra[0], ra[1], ra[2], ra[3]), make_float4(ra[4], ra[5], ra[6], ra[7]), A1, A2, A3);
#pragma unroll
                for (int i = 0; i < NCT; ++i) {
                  a_split(make_float4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]), make_float4(rb[i][4], rb[i][5], rb[i][6], rb[i][7]), P1, P2, P3);
                  g[i] = mfma_b3(A1, A2, A3, P1, P2, P3, g[i]);
#pragma unroll
                  for (int j = 0; j < 8; ++j) dbacc[i] += rb[i][j];
                }
              }
            }
          }
          if (pair) {                                       // (uniform; NCT == 1) odd k-parity hands its tile and bias sum to the even one
            if (wave & 1) {
#pragma unroll
              for (int r = 0; r < 16; ++r) xw[(wave * 16 + r) * 64 + lane] = g[0][r];
              xw[8 * 1024 + wave * 64 + lane] = dbacc[0];
            }
            __syncthreads();
            if (wave & 1) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) g[0][r] += xw[((wave + 1) * 16 + r) * 64 + lane];
            dbacc[0] += xw[8 * 1024 + (wave + 1) * 64 + lane];
          }
          if (has_tile) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cc = btile * 32 + acc_row(r, lane);
              if (cc < nb || (C <= BP && cc < Cw)) {
#pragma unroll
                for (int i = 0; i < NCT; ++i) pp[(size_t)(cb + cc) * 64 + 32 * (cn0 + i) + l31] = g[i][r];
              }
            }
          }
          if (has_db) {
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
              const float tot = dbacc[i] + __shfl_xor(dbacc[i], 32, 64);
              if (hh == 0) pp[(size_t)Cw * 64 + 32 * (cn0 + i) + l31] = tot;
            }
          }
        };
        if (NBT <= 2) wtile(std::integral_constant<int, 1>(), wave >> 2, (wave >> 1) & 1, wave & 1, 2, true);
        else if (NBT <= 4) wtile(std::integral_constant<int, 1>(), wave >> 1, wave & 1, 0, 1, false);
        else wtile(std::integral_constant<int, 2>(), wave, 0, 0, 1, false);
      } else {
      // four waves: wave = band tile, both co tiles; eight waves: wave = (band tile, co tile)
      const int btile = (TPW == 2) ? wave : (wave >> 1), cn0 = (TPW == 2) ? 0 : (wave & 1);
      const bool has_tile = btile * 32 < nb;              // uniform: this wave's band tile exists in this pass
      const bool has_db = cb == 0 && wave < 2;            // waves 0 / 1 also sum the bias gradient of co tile 0 / 1 (first pass)
      if (has_tile || has_db) {
        f32x16 g[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) g[i] = zero16();
        float dbacc = 0.f;
        const int crow = btile * 32 + l31;                // band cb + crow of this lane (rows >= nb: finite garbage, not stored)
        const float* ap = slab + (size_t)crow * HWl + 8 * hh;
        const float* bp = dal + 32 * cn0 + l31;
        const int NS = (HWl + 15) >> 4;
#pragma unroll
        for (int kq = 0; kq < 2 * KMT; ++kq) {            // HW <= 32 KMT (conv3_fused_bwd_ok)
          if (kq < NS && !zskip) {                        // uniform
            float ra[8], rb[TPW][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int k = kq * 16 + 8 * hh + j;
              const int kc = k < HWl ? k : HWl;           // the zero row
              ra[j] = ap[kq * 16 + j];
#pragma unroll
              for (int i = 0; i < TPW; ++i) rb[i][j] = bp[kc * 64 + 32 * i];
            }
            uint4 A1, A2, A3, P1, P2, P3;
            a_split(make_float4(ra[0], ra[1], ra[2], ra[3]), make_float4(ra[4], ra[5], ra[6], ra[7]), A1, A2, A3);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
              a_split(make_float4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]), make_float4(rb[i][4], rb[i][5], rb[i][6], rb[i][7]), P1, P2, P3);
              g[i] = mfma_b3(A1, A2, A3, P1, P2, P3, g[i]);
            }
            // waves 0 / 1 sum co tile 0 / 1 for the bias gradient (fixed order: j, then k-step, then the two halves)
#pragma unroll
            for (int j = 0; j < 8; ++j) dbacc += (TPW == 2 && wave != 0) ? rb[TPW - 1][j] : rb[0][j];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cc = btile * 32 + acc_row(r, lane);
          // single pass: the partial's rows (C rounded up to 4; rows >= C are dropped by the reduce); several passes: only
          // this pass's bands (the rows behind them belong to the next pass)
          if (has_tile && (cc < nb || (C <= BP && cc < Cw))) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) pp[(size_t)(cb + cc) * 64 + 32 * (cn0 + i) + l31] = g[i][r];
          }
        }
        if (has_db) {
          const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
          if (hh == 0) pp[(size_t)Cw * 64 + wave * 32 + l31] = tot;
        }
      }
      }
    }
  } else if constexpr (KS) {      // (MODE 1 with the barrier-free tap loop: this wave's tile `ot`, both channel tiles, after the fold)
    float* out = a.out + (long long)net * a.out_ns;
    const int nvalid = (a.n - s0 < S ? a.n - s0 : S) * PX;
    if (ot < MT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ot * 32 + acc_row(r, lane);
        if (m < nvalid) {
          const float* p = img + (size_t)lut[m] * CS;
          float* o = out + ((size_t)s0 * HW + m) * 64;
#pragma unroll
          for (int i = 0; i < TPW; ++i) o[32 * (nt0 + i) + l31] = own[i][r] + p[32 * (nt0 + i) + l31];
        }
      }
    }
  } else {
    float* out = a.out + (long long)net * a.out_ns;
    const int nvalid = (a.n - s0 < S ? a.n - s0 : S) * PX;  // rows that map to real samples
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int tile = slot_tile(t);
      const int nm = (t == MTW - 1) ? nmask : 3;
      if (tile < MT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tile * 32 + acc_row(r, lane);
          if (m < nvalid) {
            const float* p = img + (size_t)lut[m] * CS;
            float* o = out + ((size_t)s0 * HW + m) * 64;
            if (nm & 1) o[l31] = acc[t][0][r] + p[l31];
            if (nm & 2) o[32 + l31] = acc[t][1][r] + p[32 + l31];
          }
        }
      }
    }
  }
  STAMPG(MODE & 1, 3);
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
  const int abase = lut[l31] * CS + 8 * hh;
  f32x16 acc = zero16();
  {
    TapRegs w = c.wp;
    float4* wl = (float4*)wbuf;
    const uint4* bl = (const uint4*)wbuf + lane;
#pragma unroll 1
    for (int s = 0; s < 9; ++s) {
      __syncthreads();
      tap_put(wl, w, tid);
      __syncthreads();
      if (s + 1 < 9) w = tap_fetch(c.wg, s + 1, tid);
      const int khh = s / 3, kww = s - khh * 3;
      const float* ib = img + ((khh - 1) * c.PW + (kww - 1)) * CS + abase;
      float4 av[4];
      uint4 bv[6];
#pragma unroll
      for (int q = 0; q < 2; ++q) {          // this wave's two k-steps of 16
        const int kq = kh2 * 2 + q;
        av[2 * q] = *(const float4*)(ib + kq * 16);
        av[2 * q + 1] = *(const float4*)(ib + kq * 16 + 4);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bv[3 * q + pc] = bl[((kq * 3 + pc) * 2 + nt) * 64];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint4 A1, A2, A3;
        a_split(av[2 * q], av[2 * q + 1], A1, A2, A3);
        acc = mfma_b3(A1, A2, A3, bv[3 * q], bv[3 * q + 1], bv[3 * q + 2], acc);
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
          *p = relu_nan(acc[r] + bv + *p);
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

static size_t conv3_lds(int S, int H, int W, int MTW, int NW = 4) {
  return ((size_t)S * (H + 2) * (W + 2) * CS + WBUF + (size_t)MTW * NW * 32) * 4;
}

// Pick samples-per-workgroup S.  Cost model: MFMA tile-times queued on the busiest SIMD (workgroups
// on a CU share its 4 SIMDs, wave w of every workgroup lands on a different SIMD) plus a fixed
// per-workgroup staging/drain overhead that is hidden when a second workgroup is co-resident.
bool plan_conv3(int mode, int H, int W, int rows, Conv3Plan* p) {
  const int PX = (mode == 0) ? (2 * (H / 2)) * (2 * (W / 2)) : H * W;
  if (PX <= 0) return false;
  const int force_s = switches().conv3_s;
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
      best = cost; p->S = S; p->MTW = split ? 0 : MTW; p->lds = lds; p->nw = 4; ok = true;
      // one sample per workgroup, one tile per wave, two workgroups per CU: the barrier-free tap loop (KSG)
      p->ks = (!split && S == 1 && MTW == 1 && resident >= 2 && switches().conv3_ks != 0) ? 1 : 0;
      // One workgroup per CU and several tiles per wave (20 x 20 windows: a 131 KB image, 13 pixel tiles): EIGHT waves
      // -- two per SIMD, so that a wave's LDS reads, barriers and stores have something to hide under (round 5;
      // CMLPL_CONV3_NW8=0: four waves as before)
      const size_t lds8 = conv3_lds(S, H, W, (MT + 7) / 8, 8);
      if (!split && resident == 1 && MTW >= 2 && lds8 <= LDS_MAX && switches().conv3_nw8 != 0) {
        p->nw = 8; p->MTW = (MT + 7) / 8; p->lds = lds8; p->ks = 0;
      }
    }
  }
  return ok;
}

template <int MODE, int MTW, int NW = 4>
static hipError_t launch_conv3_t(const Conv3Args& a, dim3 grid, size_t lds, hipStream_t st) {
  static DevOnce attr_once;
  {
    hipError_t e = ensure_max_lds(attr_once, conv3x3_kernel<MODE, MTW, 0, NW>);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((conv3x3_kernel<MODE, MTW, 0, NW>), grid, dim3(64 * NW), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_conv3(int mode, int nets, int n, int H, int W, const float* in, const uint8_t* mask_in,
                        const float* wpk, long long wpk_ns, const float* bias, long long bias_ns,
                        float* out, uint8_t* mask_out, hipStream_t st, const Conv3H2* h2) {
  Conv3Plan pl;
  if (!plan_conv3(mode, H, W, nets * n, &pl)) return hipErrorInvalidValue;
  const int HW = H * W, P2 = (H / 2) * (W / 2);
  Conv3Args a;
  a.in = in; a.mask_in = mask_in; a.wpk = wpk; a.bias = bias; a.out = out; a.mask_out = mask_out;
  a.wpk_ns = wpk_ns; a.bias_ns = bias_ns;
  if (mode == 0) { a.in_ns = (long long)n * HW * 64; a.out_ns = (long long)n * P2 * 64; a.mask_out_ns = a.out_ns; a.mask_in_ns = 0; }
  else           { a.in_ns = (long long)n * P2 * 64; a.mask_in_ns = a.in_ns; a.out_ns = (long long)n * HW * 64; a.mask_out_ns = 0; }
  a.n = n; a.H = H; a.W = W; a.S = pl.S;
  conv3_set_magics(a);
  a.w0t = nullptr; a.b0 = nullptr; a.a0out = nullptr; a.w0t_ns = a.b0_ns = 0; a.C = 0; a.xn_out = nullptr;
  a.xs = XSrc(); a.part0 = nullptr; a.part0_ns = 0; a.bp = 0;
  a.wpk16 = nullptr; a.wpk16_ns = 0; a.h2flag = nullptr; a.h2flag_ns = 0; a.maxslot = 0; a.h2_noskip = switches().f16x2 == 4 || switches().zero_skip == 0; a.hstat = nullptr; a.hkind = 0; a.pairshift = 0;
  dim3 grid((n + pl.S - 1) / pl.S, nets);
  // the two-piece tap loops (one sample per workgroup: the barrier-free loop, or eight waves with staged tap weights)
  const bool h2x = h2 != nullptr && conv3_h2x_general(mode, H, W, nets * n);
  if (h2x) {
    a.wpk16 = h2->wpk16; a.wpk16_ns = h2->wpk16_ns; a.h2flag = h2->h2flag; a.h2flag_ns = h2->wpk16_ns; a.hstat = h2->hstat;
    a.hkind = h2->kind; a.maxslot = (int)(pl.lds / 4);
    const size_t lds = pl.lds + 64;
#define CMLPL_H2_LAUNCH(...)                                                                       \
    { static DevOnce attr_h;                                                                         \
      hipError_t eh = ensure_max_lds(attr_h, conv3x3_kernel<__VA_ARGS__>);                           \
      if (eh != hipSuccess) return eh;                                                               \
      hipLaunchKernelGGL((conv3x3_kernel<__VA_ARGS__>), grid, dim3(pl.ks ? 256 : 512), lds, st, a);  \
      return hipGetLastError(); }
    if (pl.ks) { if (mode == 0) CMLPL_H2_LAUNCH(0, 1, 0, 4, 2, true, true) else CMLPL_H2_LAUNCH(1, 1, 0, 4, 2, true, true) }
    if (pl.MTW == 1) { if (mode == 0) CMLPL_H2_LAUNCH(0, 1, 0, 8, 1, false, true) else CMLPL_H2_LAUNCH(1, 1, 0, 8, 1, false, true) }
    if (mode == 0) CMLPL_H2_LAUNCH(0, 2, 0, 8, 1, false, true) else CMLPL_H2_LAUNCH(1, 2, 0, 8, 1, false, true)
#undef CMLPL_H2_LAUNCH
  }
  if (pl.ks) {
    static DevOnce attr_ks;
    hipError_t e = ensure_max_lds(attr_ks, conv3x3_kernel<0, 1, 0, 4, 2, true>, conv3x3_kernel<1, 1, 0, 4, 2, true>);
    if (e != hipSuccess) return e;
    if (mode == 0) hipLaunchKernelGGL((conv3x3_kernel<0, 1, 0, 4, 2, true>), grid, dim3(256), pl.lds, st, a);
    else           hipLaunchKernelGGL((conv3x3_kernel<1, 1, 0, 4, 2, true>), grid, dim3(256), pl.lds, st, a);
    return hipGetLastError();
  }
  if (pl.nw == 8) {   // (MTW 1 .. 2: at most 16 pixel tiles in a 160 KB image)
    if (pl.MTW > 2) return hipErrorInvalidValue;
    if (mode == 0) return pl.MTW == 1 ? launch_conv3_t<0, 1, 8>(a, grid, pl.lds, st) : launch_conv3_t<0, 2, 8>(a, grid, pl.lds, st);
    return pl.MTW == 1 ? launch_conv3_t<1, 1, 8>(a, grid, pl.lds, st) : launch_conv3_t<1, 2, 8>(a, grid, pl.lds, st);
  }
#define CMLPL_DISPATCH(M)                                                      \
  switch (pl.MTW) {                                                            \
    case 0: {                                                                  \
      static DevOnce attr0;                                                    \
      hipError_t e0 = ensure_max_lds(attr0, conv3x3_small_kernel<M>);          \
      if (e0 != hipSuccess) return e0;                                         \
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

// conv0 fused into the conv1 forward (MODE 2).  Possible when the plain forward plan is one sample per workgroup
// with one pixel tile per wave (H*W <= 128) and two workgroups still fit a CU with the slab + conv0 weights in LDS.
// LDS of the fused kernel: the plain forward's regions, or slab [Cp][HW] + 64 + weights [Cp][64] if that is larger
static size_t conv3_fused_lds(int H, int W, int C, size_t plain, int NW = 4) {
  const int KQ0 = (C + 15) / 16, ring = KQ0 < SLAB_RING ? KQ0 : SLAB_RING;
  const int NT = 64 * NW;
  const size_t slot = (size_t)((4 * H * W + NT - 1) / NT) * NT * 4;       // floats per chunk slot
  const size_t need = ring * slot * 4;
  return need > plain ? need : plain;
}

// Eight-wave per-sample workgroups (one per CU): what a launch takes when its sample-net workgroups do not exceed the
// CUs (a rank's shard of a data-parallel job: 64 + 64 rows of both networks = 256 workgroups) -- a four-wave workgroup
// alone on a CU runs one wave per SIMD, nothing to issue while it waits: 39.5 / 33.6 us for half the rows of a 52.5 /
// 46.3 us launch (round 4).  CMLPL_KS8 = 0 / 1: never / at every size.  LDS: the plain regions + the [8][16][64] fold
// exchange behind the LUT.
static bool conv3_ks8(int rows) {
  const int m = switches().ks8;
  return m == 0 ? false : (m == 1 ? true : rows <= device_cus());
}
static size_t conv3_ks8_lds(size_t plain) { return plain + (size_t)8 * 16 * 64 * 4; }

// Windows of 129 .. 256 pixels (BASELINE configs[4]: 15 x 15 x 48): the eight-wave, eight-tile per-sample kernels
// conv3x3_kernel<2 | 3, 1, 1, 8, 2> with the generalised tail / head (final pooled maps of up to 12 pixels, pooled
// conv1 maps of up to 64), ONE workgroup per CU whatever the grid.  CMLPL_FUSE_BIG=0: the general kernels instead.
struct BigGeom { int HW, IMG, H2, W2, P2, NPX2, P4; };
static bool conv3_big_geom(int H, int W, BigGeom* g) {
  g->HW = H * W; g->IMG = (H + 2) * (W + 2); g->H2 = H / 2; g->W2 = W / 2; g->P2 = g->H2 * g->W2;
  g->NPX2 = (g->H2 + 2) * (g->W2 + 2); g->P4 = (g->H2 / 2) * (g->W2 / 2);
  if (switches().fuse_big == 0 || switches().fuse_conv0 == 0 || switches().fuse_tail == 0) return false;
  if (g->HW <= 128 || g->HW > 256 || g->P2 > 64 || g->P4 < 1 || g->P4 > 12) return false;
  if ((size_t)g->NPX2 * CS > WBUF) return false;                                  // the pooled map / dz2 image in the tap-weight buffer
  if ((size_t)2560 + 3 * (size_t)g->NPX2 * 36 + 3072 > (size_t)g->IMG * CS) return false;   // head row + partials + bf16 planes + conv2's exchange in the dead image
  for (int m = 0; m < 256; ++m)                                                   // the magic-number divide of the kernel
    if (((m * ((65536 + W - 1) / W)) >> 16) != m / W) return false;
  return true;
}
static size_t conv3_big_plain(const BigGeom& g) { return ((size_t)g.IMG * CS + WBUF + 256 + 8192) * 4; }
static size_t conv3_big_fwd_lds(const BigGeom& g, int C) {
  const int KQ0 = (C + 15) / 16, ring = KQ0 < SLAB_RING ? KQ0 : SLAB_RING;
  const size_t slot = (size_t)((4 * g.HW + 511) / 512) * 512 * 4;                  // floats per chunk slot
  const size_t need = ring * slot * 4, plain = conv3_big_plain(g);
  return need > plain ? need : plain;
}
static int conv3_big_bp(const BigGeom& g, int C) {
  const long long avail = (long long)(LDS_MAX / 4) - (long long)(g.HW + 1) * 64;   // floats left for the slab rows
  long long bp = avail / g.HW;
  if (bp > 256) bp = 256;
  if (bp * g.HW > 4 * SLAB_MAXQ * 256) bp = 4 * SLAB_MAXQ * 256 / g.HW;
  if (C <= bp) return C;
  return (int)(bp & ~31LL);                                                        // whole band tiles per pass
}
static size_t conv3_big_bwd_lds(const BigGeom& g, int C) {
  const int bp = conv3_big_bp(g, C);
  const size_t rows = (size_t)((bp + 31) / 32) * 32;                               // band rows a partial tile reads (garbage rows are dropped)
  const size_t slab = (size_t)bp * g.HW + (size_t)(g.HW + 1) * 64 + 8 * 1024 + 8 * 64,      // + the conv0 weight gradient's k-parity exchange
               reach = rows * g.HW + 256;
  const size_t need = (slab > reach ? slab : reach) * 4, plain = conv3_big_plain(g);
  return need > plain ? need : plain;
}
static bool conv3_big_fwd_ok(int H, int W, int C, BigGeom* g) {
  return C >= 1 && conv3_big_geom(H, W, g) && conv3_big_fwd_lds(*g, C) <= LDS_MAX;
}
static bool conv3_big_bwd_ok(int H, int W, int C, BigGeom* g) {
  if (switches().fuse_conv0_bwd == 0 || C < 1 || C > 256 || !conv3_big_geom(H, W, g)) return false;
  if (conv3_big_bp(*g, C) < 32 && conv3_big_bp(*g, C) < C) return false;
  // the head's hand-off buffers behind the LUT: dp1s [P2][64] + dp2s [P4][64] + dls [64] + red [8] share the fold exchange
  if ((size_t)g->P2 * 64 + (size_t)g->P4 * 64 + 64 + 8 > 8192) return false;
  return conv3_big_bwd_lds(*g, C) <= LDS_MAX;
}

bool conv3_fused_ok(int H, int W, int C, int rows) {
  const bool off = switches().fuse_conv0 == 0;
  if (off || C < 1) return false;
  BigGeom bg;
  if (conv3_big_fwd_ok(H, W, C, &bg)) return true;
  Conv3Plan pl;
  if (!plan_conv3(0, H, W, rows, &pl)) return false;
  if (pl.S != 1 || pl.MTW != 1 || H * W > 128) return false;
  for (int m = 0; m < 128; ++m)                       // the magic-number divide of the kernel
    if (((m * ((65536 + W - 1) / W)) >> 16) != m / W) return false;
  return 2 * conv3_fused_lds(H, W, C, pl.lds) <= LDS_MAX;
}

bool conv3_fused_tail_ok(int H, int W, int C, int rows, int K) {
  const bool off = switches().fuse_tail == 0;
  const int H2 = H / 2, W2 = W / 2;
  BigGeom bg;
  if (conv3_big_fwd_ok(H, W, C, &bg)) return K >= 1 && K <= 64;
  return !off && conv3_fused_ok(H, W, C, rows) && H2 / 2 == 2 && W2 / 2 == 2 && (H2 + 2) * (W2 + 2) * CS <= 4096 &&
         K >= 1 && K <= 64;
}

hipError_t launch_conv3_fused(int nets, int n, int C, int H, int W, const XSrc& xs, const float* w0t, long long w0t_ns,
                              const float* b0, long long b0_ns, float* a0out, const float* wpk, long long wpk_ns,
                              const float* bias, long long bias_ns, float* out, uint8_t* mask_out,
                              const FwdTail* tail, hipStream_t st, float* xn_out) {
  Conv3Plan pl;
  if (!conv3_fused_ok(H, W, C, nets * n) || !plan_conv3(0, H, W, nets * n, &pl)) return hipErrorInvalidValue;
  const int HW = H * W, P2 = (H / 2) * (W / 2);
  BigGeom bg;
  const bool big = conv3_big_fwd_ok(H, W, C, &bg);
  if (big && tail == nullptr) return hipErrorInvalidValue;          // (the eight-tile kernels exist with their tail only)
  Conv3Args a;
  a.in = nullptr; a.mask_in = nullptr; a.wpk = wpk; a.bias = bias; a.out = out; a.mask_out = mask_out;
  a.wpk_ns = wpk_ns; a.bias_ns = bias_ns;
  a.in_ns = 0; a.mask_in_ns = 0; a.out_ns = (long long)n * P2 * 64; a.mask_out_ns = a.out_ns;
  a.n = n; a.H = H; a.W = W; a.S = 1;
  conv3_set_magics(a);
  a.w0t = w0t; a.w0t_ns = w0t_ns; a.b0 = b0; a.b0_ns = b0_ns; a.a0out = a0out; a.C = C; a.xn_out = xn_out;
  a.xs = xs; a.part0 = nullptr; a.part0_ns = 0; a.bp = 0;
  a.wpk16 = nullptr; a.wpk16_ns = 0; a.h2flag = nullptr; a.h2flag_ns = 0; a.maxslot = 0; a.h2_noskip = switches().f16x2 == 4 || switches().zero_skip == 0; a.hstat = nullptr; a.hkind = 0; a.pairshift = 0;
  (void)HW;
  if (tail != nullptr) {
    if (!conv3_fused_tail_ok(H, W, C, nets * n, tail->K)) return hipErrorInvalidValue;
    a.w2f = tail->w2f; a.w2f_ns = tail->w2f_ns; a.b2 = tail->b2; a.wc = tail->wc; a.bc = tail->bc; a.p_ns = tail->p_ns;
    a.yin = tail->y; a.dropmask = tail->dropmask; a.dropgen = tail->dropgen; a.catd = tail->catd; a.ynorm = tail->ynorm;
    a.logits = tail->logits; a.feat = tail->feat; a.p2out = tail->p2; a.m2out = tail->m2;
    a.dropout_p = tail->dropout_p; a.train = tail->train; a.K = tail->K;
    static DevOnce attr_once;
    hipError_t e = ensure_max_lds(attr_once, conv3x3_kernel<2, 1, 1>, conv3x3_kernel<2, 1, 1, 8>, conv3x3_kernel<2, 1, 1, 8, 2>);
    if (e != hipSuccess) return e;
    const bool h2x = (switches().f16x2 == 1 || switches().f16x2 == 2 || switches().f16x2 == 4) && tail->w1h != nullptr && tail->h2flag != nullptr;
    if (h2x) { a.wpk16 = tail->w1h; a.wpk16_ns = tail->w1h_ns; a.h2flag = tail->h2flag; a.h2flag_ns = tail->w1h_ns; a.hstat = tail->hstat; }
    if (big) {
      const size_t ldsb = conv3_big_fwd_lds(bg, C);
      if (h2x && ldsb + 64 <= LDS_MAX) {
        static DevOnce attr_hb;
        hipError_t eh = ensure_max_lds(attr_hb, conv3x3_kernel<2, 1, 1, 8, 2, false, true>);
        if (eh != hipSuccess) return eh;
        a.maxslot = (int)(ldsb / 4);
        hipLaunchKernelGGL((conv3x3_kernel<2, 1, 1, 8, 2, false, true>), dim3(n, nets), dim3(512), ldsb + 64, st, a);
        return hipGetLastError();
      }
      a.hstat = nullptr;
      hipLaunchKernelGGL((conv3x3_kernel<2, 1, 1, 8, 2>), dim3(n, nets), dim3(512), ldsb, st, a);
      return hipGetLastError();
    }
    if (conv3_ks8(nets * n)) {
      const size_t lds8 = conv3_fused_lds(H, W, C, conv3_ks8_lds(pl.lds), 8);
      if (lds8 > LDS_MAX) return hipErrorInvalidValue;
      if (h2x && lds8 + 64 <= LDS_MAX) {
        static DevOnce attr_h8;
        hipError_t eh = ensure_max_lds(attr_h8, conv3x3_kernel<2, 1, 1, 8, 1, false, true>);
        if (eh != hipSuccess) return eh;
        a.maxslot = (int)(lds8 / 4);
        hipLaunchKernelGGL((conv3x3_kernel<2, 1, 1, 8, 1, false, true>), dim3(n, nets), dim3(512), lds8 + 64, st, a);
        return hipGetLastError();
      }
      hipLaunchKernelGGL((conv3x3_kernel<2, 1, 1, 8>), dim3(n, nets), dim3(512), lds8, st, a);
      return hipGetLastError();
    }
    const size_t lds = conv3_fused_lds(H, W, C, pl.lds);
    if (h2x && 2 * (lds + 64) <= LDS_MAX) {
      // conv1's taps on two fp16 pieces; one more LDS word (the image's largest magnitude) behind everything else
      static DevOnce attr_h;
      hipError_t eh = ensure_max_lds(attr_h, conv3x3_kernel<2, 1, 1, 4, 2, false, true>);
      if (eh != hipSuccess) return eh;
      a.maxslot = (int)(lds / 4);
      hipLaunchKernelGGL((conv3x3_kernel<2, 1, 1, 4, 2, false, true>), dim3(n, nets), dim3(256), lds + 64, st, a);
      return hipGetLastError();
    }
    hipLaunchKernelGGL((conv3x3_kernel<2, 1, 1>), dim3(n, nets), dim3(256), lds, st, a);
    return hipGetLastError();
  }
  return launch_conv3_t<2, 1>(a, dim3(n, nets), conv3_fused_lds(H, W, C, pl.lds), st);
}

// Whole-image inference from the scene cube (cmlpl_infer_cube): the fused eval forward with the cube gather as its slab
// source -- n consecutive pixels from pix0, one network; y = relu(feat_spe(spectrum)) of the same pixels comes from the
// spectral launch in front.  Shapes: whatever the per-sample forward with its tail takes (four-wave kernels up to 128
// window pixels, eight-tile kernels up to 256).
// (one sample per workgroup whatever the training planner would pick for this window at this batch: its S > 1 choices
//  are a throughput trade for the multi-sample kernels, not a limit of the per-sample one)
static bool conv3_infer_small_ok(int H, int W, int C, size_t* lds) {
  const int H2 = H / 2, W2 = W / 2;
  if (switches().fuse_conv0 == 0 || switches().fuse_tail == 0 || C < 1 || H * W > 128 || H < 8 || W < 8) return false;
  if (H2 / 2 != 2 || W2 / 2 != 2 || (H2 + 2) * (W2 + 2) * CS > 4096) return false;
  for (int m = 0; m < 128; ++m)
    if (((m * ((65536 + W - 1) / W)) >> 16) != m / W) return false;
  *lds = conv3_fused_lds(H, W, C, conv3_lds(1, H, W, 1));
  return 2 * *lds <= LDS_MAX;
}
bool conv3_infer_ok(int H, int W, int C, int K) {
  BigGeom bg;
  size_t lds;
  return H == W && K >= 1 && K <= 64 && (conv3_big_fwd_ok(H, W, C, &bg) || conv3_infer_small_ok(H, W, C, &lds));
}

hipError_t launch_conv3_infer(int n, int C, int H, int W, const float* cube, int crows, int ccols, long long pix0,
                              const float* w0t, const float* b0, const float* wpk, const float* bias, const FwdTail& t,
                              long long* labels_out, hipStream_t st) {
  if (!conv3_infer_ok(H, W, C, t.K) || n < 1 || !cube || !labels_out) return hipErrorInvalidValue;
  if ((long long)crows * ccols * C >= (1LL << 31) || W / 2 > crows || W / 2 > ccols) return hipErrorInvalidValue;   // (32-bit offsets; one mirror fold)
  BigGeom bg;
  const bool big = conv3_big_fwd_ok(H, W, C, &bg);
  size_t lds_small = 0;
  if (!big && !conv3_infer_small_ok(H, W, C, &lds_small)) return hipErrorInvalidValue;
  Conv3Args a;
  memset(&a, 0, sizeof(a));
  a.wpk = wpk; a.bias = bias;
  a.n = n; a.H = H; a.W = W; a.S = 1;
  conv3_set_magics(a);
  a.w0t = w0t; a.b0 = b0; a.C = C;
  a.w2f = t.w2f; a.b2 = t.b2; a.wc = t.wc; a.bc = t.bc; a.yin = t.y; a.logits = t.logits; a.K = t.K;
  a.train = 0; a.dropout_p = 0.f;
  a.cube = cube; a.crows = crows; a.ccols = ccols; a.pix0 = pix0; a.labels_out = labels_out;
  static DevOnce attr_once;
  hipError_t e = ensure_max_lds(attr_once, conv3x3_kernel<2, 1, 2>, conv3x3_kernel<2, 1, 2, 8, 2>);
  if (e != hipSuccess) return e;
  const dim3 grid(8 * ((n + 7) / 8), 1);
  if (big) hipLaunchKernelGGL((conv3x3_kernel<2, 1, 2, 8, 2>), grid, dim3(512), conv3_big_fwd_lds(bg, C), st, a);
  else hipLaunchKernelGGL((conv3x3_kernel<2, 1, 2>), grid, dim3(256), lds_small, st, a);
  return hipGetLastError();
}

// conv0 weight gradient fused into the conv1 data gradient (MODE 3): same shape conditions as the fused forward, and
// two workgroups per CU with slab [bp][HW] + da0 [HW+1][64] in LDS, bp = bands per pass (at most four band tiles).
// All C bands in one pass when they fit (B2: 103), else the largest multiple of four that does (B4: 104 + 96).
static int conv3_bwd_bp(int H, int W, int C) {
  const int HW = H * W;
  const long long avail = (long long)(LDS_MAX / 2 / 4) - (long long)(HW + 1) * 64;     // floats left for the slab rows
  long long bp = avail / HW;
  if (bp > 128) bp = 128;
  if ((long long)bp * HW > 4 * SLAB_MAXQ * 256) bp = 4 * SLAB_MAXQ * 256 / HW;
  if (C <= bp) return C;
  return (int)(bp & ~3LL);
}
static size_t conv3_fused_bwd_lds(int H, int W, int C, size_t plain) {
  const int bp = conv3_bwd_bp(H, W, C);
  const size_t need = ((size_t)(bp > 0 ? bp : 0) * H * W + (size_t)(H * W + 1) * 64) * 4;
  return need > plain ? need : plain;
}

bool conv3_fused_bwd_ok(int H, int W, int C, int rows) {
  const bool off = switches().fuse_conv0 == 0;
  const bool offb = switches().fuse_conv0_bwd == 0;
  if (off || offb || C < 1 || C > 256) return false;
  BigGeom bg;
  if (conv3_big_bwd_ok(H, W, C, &bg)) return true;
  Conv3Plan pl;
  if (!plan_conv3(1, H, W, rows, &pl)) return false;
  if (pl.S != 1 || pl.MTW != 1 || H * W > 128 || conv3_bwd_bp(H, W, C) < 32) return false;
  // band rows up to 127 are read (garbage rows are dropped later) and must stay inside the allocation
  const size_t lds = conv3_fused_bwd_lds(H, W, C, pl.lds);
  if ((size_t)128 * H * W * 4 > lds) return false;
  return 2 * lds <= LDS_MAX;
}

bool conv3_fused_head_ok(int H, int W, int C, int rows, int K) {
  const bool off = switches().fuse_tail == 0;
  const int H2 = H / 2, W2 = W / 2;
  BigGeom bg;
  if (conv3_big_bwd_ok(H, W, C, &bg)) return K >= 1 && K <= 64;
  if (off || !conv3_fused_bwd_ok(H, W, C, rows) || H2 / 2 != 2 || W2 / 2 != 2 || (H2 + 2) * (W2 + 2) * CS > 4096 ||
      H2 * W2 > 32 || K < 1 || K > 64)
    return false;
  // LDS behind the LUT: dp1s [P2][64] + dp2s [256] + dls [64] + red [4]
  Conv3Plan pl;
  if (!plan_conv3(1, H, W, rows, &pl)) return false;
  const size_t need = ((size_t)(H + 2) * (W + 2) * CS + WBUF + 128 + (size_t)H2 * W2 * 64 + 256 + 64 + 4) * 4;
  return need <= conv3_fused_bwd_lds(H, W, C, pl.lds);
}

hipError_t launch_conv3_fused_bwd(int nets, int n, int C, int H, int W, const float* dpool, const uint8_t* mask,
                                  const float* wpk, long long wpk_ns, const XSrc& xs, float* part0, long long part0_ns,
                                  const BwdHead* head, hipStream_t st) {
  Conv3Plan pl;
  if (!conv3_fused_bwd_ok(H, W, C, nets * n) || !plan_conv3(1, H, W, nets * n, &pl)) return hipErrorInvalidValue;
  // slab_range() reads plain rows by batch row: no noise, no index lists (the rows the forward saw, api.hip)
  if (xs.sigma != 0.f || xs.sel.lab_idx != nullptr || xs.sel.unl_idx != nullptr) return hipErrorInvalidValue;
  const int P2 = (H / 2) * (W / 2);
  BigGeom bg;
  const bool big = conv3_big_bwd_ok(H, W, C, &bg);
  if (big && head == nullptr) return hipErrorInvalidValue;          // (the eight-tile kernels exist with their head only)
  Conv3Args a;
  a.in = dpool; a.mask_in = mask; a.wpk = wpk; a.bias = nullptr; a.out = nullptr; a.mask_out = nullptr;
  a.wpk_ns = wpk_ns; a.bias_ns = 0;
  a.in_ns = (long long)n * P2 * 64; a.mask_in_ns = a.in_ns; a.out_ns = 0; a.mask_out_ns = 0;
  a.n = n; a.H = H; a.W = W; a.S = 1;
  conv3_set_magics(a);
  a.w0t = nullptr; a.b0 = nullptr; a.a0out = nullptr; a.w0t_ns = a.b0_ns = 0; a.C = C; a.xn_out = nullptr;
  a.xs = xs; a.part0 = part0; a.part0_ns = part0_ns; a.bp = big ? conv3_big_bp(bg, C) : conv3_bwd_bp(H, W, C);
  a.wpk16 = nullptr; a.wpk16_ns = 0; a.h2flag = nullptr; a.h2flag_ns = 0; a.maxslot = 0; a.h2_noskip = switches().f16x2 == 4 || switches().zero_skip == 0; a.hstat = nullptr; a.hkind = 0; a.pairshift = 0;
  if (head != nullptr) {
    if (!conv3_fused_head_ok(H, W, C, nets * n, head->K)) return hipErrorInvalidValue;
    a.dlogits = head->dlogits; a.dfeat = head->dfeat; a.hmask = head->mask; a.wc = head->wc; a.p_ns = head->p_ns;
    a.yin = head->y; a.ynrm = head->ynorm; a.m2in = head->m2; a.w2d = head->w2d; a.w2d_ns = head->w2d_ns;
    a.dy = head->dy; a.dp2out = head->dp2; a.dp1out = head->dp1; a.K = head->K;
    static DevOnce attr_once;
    hipError_t e = ensure_max_lds(attr_once, conv3x3_kernel<3, 1, 1>, conv3x3_kernel<3, 1, 1, 8>, conv3x3_kernel<3, 1, 1, 8, 2>);
    if (e != hipSuccess) return e;
    const bool h2x = (switches().f16x2 == 1 || switches().f16x2 == 3 || switches().f16x2 == 4) && head->w1h != nullptr && head->h2flag != nullptr;
    if (h2x) { a.wpk16 = head->w1h; a.wpk16_ns = head->w1h_ns; a.h2flag = head->h2flag; a.h2flag_ns = head->w1h_ns; a.hstat = head->hstat; }
    a.pairshift = (nets == 2 && switches().bwd_pair != 0) ? 1 : 0;
    if (big) {
      const size_t ldsb = conv3_big_bwd_lds(bg, C);
      if (h2x && ldsb + 64 <= LDS_MAX) {
        static DevOnce attr_hb;
        hipError_t eh = ensure_max_lds(attr_hb, conv3x3_kernel<3, 1, 1, 8, 2, false, true>);
        if (eh != hipSuccess) return eh;
        a.maxslot = (int)(ldsb / 4);
        hipLaunchKernelGGL((conv3x3_kernel<3, 1, 1, 8, 2, false, true>), dim3(n, nets), dim3(512), ldsb + 64, st, a);
        return hipGetLastError();
      }
      a.hstat = nullptr;
      hipLaunchKernelGGL((conv3x3_kernel<3, 1, 1, 8, 2>), dim3(n, nets), dim3(512), ldsb, st, a);
      return hipGetLastError();
    }
    if (conv3_ks8(nets * n)) {
      const size_t lds8 = conv3_fused_bwd_lds(H, W, C, conv3_ks8_lds(pl.lds));
      if (lds8 > LDS_MAX) return hipErrorInvalidValue;
      if (h2x && lds8 + 64 <= LDS_MAX) {
        static DevOnce attr_h8;
        hipError_t eh = ensure_max_lds(attr_h8, conv3x3_kernel<3, 1, 1, 8, 1, false, true>);
        if (eh != hipSuccess) return eh;
        a.maxslot = (int)(lds8 / 4);
        hipLaunchKernelGGL((conv3x3_kernel<3, 1, 1, 8, 1, false, true>), dim3(n, nets), dim3(512), lds8 + 64, st, a);
        return hipGetLastError();
      }
      hipLaunchKernelGGL((conv3x3_kernel<3, 1, 1, 8>), dim3(n, nets), dim3(512), lds8, st, a);
      return hipGetLastError();
    }
    const size_t lds = conv3_fused_bwd_lds(H, W, C, pl.lds);
    if (h2x && 2 * (lds + 64) <= LDS_MAX) {
      static DevOnce attr_h;
      hipError_t eh = ensure_max_lds(attr_h, conv3x3_kernel<3, 1, 1, 4, 2, false, true>);
      if (eh != hipSuccess) return eh;
      a.maxslot = (int)(lds / 4);
      hipLaunchKernelGGL((conv3x3_kernel<3, 1, 1, 4, 2, false, true>), dim3(n, nets), dim3(256), lds + 64, st, a);
      return hipGetLastError();
    }
    hipLaunchKernelGGL((conv3x3_kernel<3, 1, 1>), dim3(n, nets), dim3(256), lds, st, a);
    return hipGetLastError();
  }
  return launch_conv3_t<3, 1>(a, dim3(n, nets), conv3_fused_bwd_lds(H, W, C, pl.lds), st);
}

// do the fused forward AND backward launches of this shape / batch take the two-piece kernels (which also collect the
// batch statistics the two-piece weight gradient needs)?  Mirrors the launchers below.
bool conv3_h2x_both(int H, int W, int C, int rows, int K) {
  if (switches().f16x2 != 1 && switches().f16x2 != 4) return false;
  BigGeom bg;
  if (!conv3_fused_tail_ok(H, W, C, rows, K) || !conv3_fused_head_ok(H, W, C, rows, K)) return false;
  if (conv3_big_fwd_ok(H, W, C, &bg) || conv3_big_bwd_ok(H, W, C, &bg))           // the eight-tile kernels (windows of 129 .. 256 pixels)
    return conv3_big_fwd_ok(H, W, C, &bg) && conv3_big_bwd_ok(H, W, C, &bg) && conv3_big_fwd_lds(bg, C) + 64 <= LDS_MAX &&
           conv3_big_bwd_lds(bg, C) + 64 <= LDS_MAX;
  Conv3Plan pf, pb;
  if (!plan_conv3(0, H, W, rows, &pf) || !plan_conv3(1, H, W, rows, &pb)) return false;
  if (conv3_ks8(rows))
    return conv3_fused_lds(H, W, C, conv3_ks8_lds(pf.lds), 8) + 64 <= LDS_MAX && conv3_fused_bwd_lds(H, W, C, conv3_ks8_lds(pb.lds)) + 64 <= LDS_MAX;
  return 2 * (conv3_fused_lds(H, W, C, pf.lds) + 64) <= LDS_MAX && 2 * (conv3_fused_bwd_lds(H, W, C, pb.lds) + 64) <= LDS_MAX;
}

// would launch_conv3 run this map on a two-piece kernel?  (one sample per workgroup: the barrier-free loop of at most four
// tiles, or the eight-wave kernels with one or two tiles per wave; LDS for one more word)
bool conv3_h2x_general(int mode, int H, int W, int rows) {
  if (switches().f16x2 != 1 && switches().f16x2 != 4) return false;
  Conv3Plan pl;
  if (!plan_conv3(mode, H, W, rows, &pl) || pl.S != 1) return false;
  if (pl.ks) return 2 * (pl.lds + 64) <= LDS_MAX;
  return pl.nw == 8 && pl.MTW >= 1 && pl.MTW <= 2 && pl.lds + 64 <= LDS_MAX;
}

}  // namespace cmlpl
