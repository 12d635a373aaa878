// The loss block of the CMLPL step (train.py:191-266), forward + analytic backward, and the
// memory-bank write (train.py:223-237).
//
// Every kernel works on a ROW SHARD: inputs (logits, feats, labels, probabilities) are the GLOBAL
// batch, outputs are produced for the local labelled rows [lab0, lab0+nlab) and local unlabelled rows
// [unl0, unl0+nunl) only.  One GPU = the full range; under data parallelism each rank owns a slice and
// the host code in cmlpl_amd/distributed.py exchanges what crosses ranks.
//
//   phase 1
//     pair_exp_kernel   : exp(f_a . f_b^T / T) tiles on the fp32 MFMA for the three products
//                         fU_w x bank0 (train.py:213), fU_s x bank1 (:217), fU_s x fU_w (:246,257),
//                         local rows x all columns; bank tiles are reduced on the fly to row sums and
//                         E.bank_probs partials (the [btu,Q] similarity matrix is never materialised)
//     loss_rows_kernel  : one wavefront per local sample, lanes = classes: CE (:191-194), softmax
//                         (:203,209), smoothing (:214-219), threshold masks (:220-228), mutual soft-CE
//                         (:239-242) and d/dlogits of all of it
//   phase 2  (needs the smoothed probabilities of ALL unlabelled rows)
//     graph_loss_kernel : one wavefront per local unlabelled row: similarity softmax denominator by
//                         wavefront shuffles (:247), pseudo-label graph Q/Qn (:249-256), contrastive
//                         loss (:260-265) and dL/d(sim logits) -> G, G^T
//     finalize_kernel   : bank write of the GLOBAL batch (modulo Q) + this shard's share of the logged
//                         scalars (:266,270,274-278); shares are additive across shards
//     gemm_tn           : dfeat_s(local rows) = G . fU_w ; dfeat_w(all rows, partial) = G^T . fU_s(local)
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"
#include "gemm_tn.hpp"

namespace cmlpl {

enum { RL_CLS_S = 0, RL_CLS_W, RL_ACC, RL_CON_S, RL_CON_W, RL_CTR, RL_NPOS, RL_NNEG, RL_COUNT };

size_t loss_ws_floats(int nlab, int nunl, int btu_g, int K, int Q) {
  const size_t CT = (Q + 31) / 32, RL = (size_t)(nlab > nunl ? nlab : nunl);
  size_t f = 0;
  f += 2 * CT * nunl;                 // rs_part
  f += 2 * CT * nunl * K;             // ep_part
  f += 3 * (size_t)nunl * btu_g;      // Smat, GT, G
  f += 2 * (size_t)nunl;              // masks
  f += RL_COUNT * RL;                 // rowloss
  return (f + 63) & ~(size_t)63;
}

void loss_ws_carve(LossArgs& a, float* ws) {
  const size_t CT = (a.Q + 31) / 32, RL = (size_t)(a.nlab > a.nunl ? a.nlab : a.nunl);
  const size_t nunl = a.nunl, btu = a.btu;
  a.rs_part = ws; ws += 2 * CT * nunl;
  a.ep_part = ws; ws += 2 * CT * nunl * a.K;
  a.Smat = ws; ws += nunl * btu;
  a.GT = ws; ws += nunl * btu;
  a.G = ws; ws += nunl * btu;
  a.masks = ws; ws += 2 * nunl;
  a.rowloss = ws; ws += RL_COUNT * RL;
}

// probabilities of global unlabelled row g: which = 0 p_w, 1 p_s (smoothed), 2 p_w0, 3 p_s0.
// Layout is shard-major [btu / pshard][4][pshard][K] -- exactly what an all-gather of per-rank
// [4][nunl][K] blocks produces (one GPU: pshard = btu, i.e. plain [4][btu][K]).
__device__ __forceinline__ const float* prob_row(const LossArgs& a, int which, int g) {
  const int sh = g / a.pshard, l = g - sh * a.pshard;
  return a.probs_g + (((long long)sh * 4 + which) * a.pshard + l) * a.K;
}

// Global rows by (network, labelled/unlabelled, index): one buffer in plain mode, or the rank-major blocks of the
// all-gather in packed mode (no re-ordering copy: the block of rank r holds that rank's rows [labelled ; unlabelled]).
__device__ __forceinline__ const float* feat_lab(const LossArgs& a, int net, int g) {
  if (a.recv_f == nullptr) return a.feat + ((long long)net * (a.bt + a.btu) + g) * FD;
  const int r = g / a.bt_l, i = g - r * a.bt_l, n_l = a.bt_l + a.btu_l;
  return a.recv_f + r * a.pack_f + ((long long)net * n_l + i) * FD;
}
__device__ __forceinline__ const float* feat_unl(const LossArgs& a, int net, int g) {
  if (a.recv_f == nullptr) return a.feat + ((long long)net * (a.bt + a.btu) + a.bt + g) * FD;
  const int r = g / a.btu_l, i = g - r * a.btu_l, n_l = a.bt_l + a.btu_l;
  return a.recv_f + r * a.pack_f + ((long long)net * n_l + a.bt_l + i) * FD;
}
// logits: every reader asks for rows of THIS shard only (packed mode keeps them local: g - lab0 / g - unl0 is the local row)
__device__ __forceinline__ const float* logit_lab(const LossArgs& a, int net, int g) {
  if (a.recv_f == nullptr) return a.logits + ((long long)net * (a.bt + a.btu) + g) * a.K;
  return a.logits_loc + ((long long)net * (a.bt_l + a.btu_l) + (g - a.lab0)) * a.K;
}
__device__ __forceinline__ const float* logit_unl(const LossArgs& a, int net, int g) {
  if (a.recv_f == nullptr) return a.logits + ((long long)net * (a.bt + a.btu) + a.bt + g) * a.K;
  return a.logits_loc + ((long long)net * (a.bt_l + a.btu_l) + a.bt_l + (g - a.unl0)) * a.K;
}
__device__ __forceinline__ int label_of(const LossArgs& a, int g) {
  if (a.recv_f == nullptr) return (int)a.labels[rowsel_index(a.sel, true, g)];
  const int r = g / a.bt_l, i = g - r * a.bt_l, n_l = a.bt_l + a.btu_l;
  return (int)(a.recv_f[r * a.pack_f + 2LL * n_l * FD + i] + 0.5f);
}

// the step scalars of the loss block: launch arguments, or the device-side row (graph replay)
// (value first, then the override: written as `d ? d->smooth : a.smooth` the compiler selects between the two
//  ADDRESSES, the kernel-argument struct escapes into a generic pointer and is copied to scratch -- 352 bytes per lane
//  and 15 -> 30 us for pair_exp16_kernel)
__device__ __forceinline__ int loss_smooth(const LossArgs& a) {
  int v = a.smooth;
  const cmlpl_dyn* d = dyn_row(a.sel.dyn);
  if (d != nullptr) v = uni32(d->smooth);
  return v;
}

// One workgroup per 32x32 tile of one product; the 1024-long contraction is split over the 4 waves (256
// each = 8 lines of 128 B per row).  Loading MFMA fragments straight from memory would touch 32 different
// 128-B lines per instruction and use a quarter of each (measured: ~500 MB of L1<->L2 traffic per launch,
// waves 66 % stalled).  Instead each instruction fetches WHOLE lines -- 8 rows x 128 B -- into a small
// per-wave LDS tile ([32 rows][32+4 floats], no block barrier: only the owning wave touches it), from
// which the fragments are read as ds_read_b128; the next step's loads are already in flight in registers.
constexpr int PT = 36;              // per-wave tile row stride in floats (32 + 4)

__global__ __launch_bounds__(256) void pair_exp_kernel(LossArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 32 * PT];  // 4 waves x (A,B) tiles; later reused as red[4][16][64]
  float (*red)[16][64] = (float (*)[16][64])lds;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = blockIdx.z;
  if (prob < 2 && !loss_smooth(a)) return;
  const int btu = a.btu, nunl = a.nunl, K = a.K;
  const float* A = feat_unl(a, prob == 0 ? 1 : 0, a.unl0);        // local rows (one rank's block: contiguous)
  const float* B = (prob == 0) ? a.bank_f[0] : (prob == 1) ? a.bank_f[1] : nullptr;   // prob 2: fU_w, all rows
  const int NB = (prob < 2) ? a.Q : btu;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  if (c0 >= NB || r0 >= nunl) return;
  const int jb = c0 + l31;
  float* tA = lds + wave * (2 * 32 * PT);        // 2 x 1152 floats per wave
  float* tB = tA + 32 * PT;
  // loader role: lane -> (row group r8 = lane>>3, 16-byte chunk c8 = lane&7); load j covers rows 8j + r8
  const int r8 = lane >> 3, c8 = lane & 7;
  // (named registers, not arrays: loop-carried arrays end up in scratch)
  const long long ko = wave * 256 + c8 * 4;
#define CMLPL_ROWPTR(base, r, lim) ((base) + (long long)((r) < (lim) ? (r) : 0) * FD + ko)
#define CMLPL_BROW(r) ((B != nullptr ? B + (long long)((r) < NB ? (r) : 0) * FD : feat_unl(a, 1, (r) < NB ? (r) : 0)) + ko)
  const float* la0 = CMLPL_ROWPTR(A, r0 + r8, nunl);
  const float* la1 = CMLPL_ROWPTR(A, r0 + 8 + r8, nunl);
  const float* la2 = CMLPL_ROWPTR(A, r0 + 16 + r8, nunl);
  const float* la3 = CMLPL_ROWPTR(A, r0 + 24 + r8, nunl);
  const float* lb0 = CMLPL_BROW(c0 + r8);
  const float* lb1 = CMLPL_BROW(c0 + 8 + r8);
  const float* lb2 = CMLPL_BROW(c0 + 16 + r8);
  const float* lb3 = CMLPL_BROW(c0 + 24 + r8);
#undef CMLPL_BROW
#undef CMLPL_ROWPTR
  float4 a0 = *(const float4*)la0, a1 = *(const float4*)la1, a2 = *(const float4*)la2, a3 = *(const float4*)la3;
  float4 b0 = *(const float4*)lb0, b1 = *(const float4*)lb1, b2 = *(const float4*)lb2, b3 = *(const float4*)lb3;
  float* wA = tA + r8 * PT + c8 * 4;
  float* wB = tB + r8 * PT + c8 * 4;
  const float* rA = tA + l31 * PT + hh * 16;     // half-wave hh owns floats [16hh, 16hh+16) of the line
  const float* rB = tB + l31 * PT + hh * 16;
  f32x16 acc = zero16();
#pragma unroll 1
  for (int ln = 0; ln < 8; ++ln) {               // 8 lines of 32 floats
    *(float4*)(wA) = a0; *(float4*)(wA + 8 * PT) = a1; *(float4*)(wA + 16 * PT) = a2; *(float4*)(wA + 24 * PT) = a3;
    *(float4*)(wB) = b0; *(float4*)(wB + 8 * PT) = b1; *(float4*)(wB + 16 * PT) = b2; *(float4*)(wB + 24 * PT) = b3;
    if (ln + 1 < 8) {
      const int o = (ln + 1) * 32;
      a0 = *(const float4*)(la0 + o); a1 = *(const float4*)(la1 + o);
      a2 = *(const float4*)(la2 + o); a3 = *(const float4*)(la3 + o);
      b0 = *(const float4*)(lb0 + o); b1 = *(const float4*)(lb1 + o);
      b2 = *(const float4*)(lb2 + o); b3 = *(const float4*)(lb3 + o);
    }
    const float4 x0 = *(const float4*)(rA), x1 = *(const float4*)(rA + 4), x2 = *(const float4*)(rA + 8),
                 x3 = *(const float4*)(rA + 12);
    const float4 y0 = *(const float4*)(rB), y1 = *(const float4*)(rB + 4), y2 = *(const float4*)(rB + 8),
                 y3 = *(const float4*)(rB + 12);
#define CMLPL_M4(PA, PB)                                                            \
    acc = mfma32(PA.x, PB.x, acc); acc = mfma32(PA.y, PB.y, acc);                   \
    acc = mfma32(PA.z, PB.z, acc); acc = mfma32(PA.w, PB.w, acc);
    CMLPL_M4(x0, y0) CMLPL_M4(x1, y1) CMLPL_M4(x2, y2) CMLPL_M4(x3, y3)
#undef CMLPL_M4
  }
  __syncthreads();                               // every wave is done with its staging tile: reuse as `red`
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // wave w finishes accumulator rows 4w .. 4w+3
  const bool jv = jb < NB;
  float e[4];
  int irow[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = wave * 4 + q;
    const float u = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
    e[q] = jv ? expf(u / a.T) : 0.f;
    irow[q] = r0 + acc_row(r, lane);
  }
  if (prob == 2) {
    if (jv) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (irow[q] < nunl) a.Smat[(long long)irow[q] * btu + jb] = e[q];
    }
    return;
  }
  const int CT = (a.Q + 31) >> 5, ctile = c0 >> 5;
  float* rs = a.rs_part + ((long long)prob * CT + ctile) * nunl;
  float* ep = a.ep_part + ((long long)prob * CT + ctile) * nunl * K;
  if (K <= 32) {
    // row sums and E . bank_probs as a small LDS product instead of 5 shuffles per (row, class): the tile E
    // [32][33] and the probability tile [32 cols][33] go to LDS (`red` is dead after the barrier), each thread
    // then forms whole outputs, columns summed in index order
    __syncthreads();                                   // all reads of `red` are done
    float* ew = lds;                                   // [32][33]
    float* pw = lds + 32 * 33;                         // [32][33]
#pragma unroll
    for (int q = 0; q < 4; ++q) ew[acc_row(wave * 4 + q, lane) * 33 + l31] = e[q];
    for (int i = tid; i < 32 * K; i += 256) {
      const int c = i / K, k = i - c * K;
      pw[c * 33 + k] = (c0 + c < NB) ? a.bank_p[prob][(long long)(c0 + c) * K + k] : 0.f;
    }
    __syncthreads();
    for (int o = tid; o < 32 * (K + 1); o += 256) {
      const int row = o & 31, kk = o >> 5;             // kk == K: the plain row sum
      const float* er = ew + row * 33;
      float sum = 0.f;
      if (kk < K) {
#pragma unroll 8
        for (int c = 0; c < 32; ++c) sum += er[c] * pw[c * 33 + kk];
      } else {
#pragma unroll 8
        for (int c = 0; c < 32; ++c) sum += er[c];
      }
      const int ir = r0 + row;
      if (ir < nunl) {
        if (kk < K) ep[(long long)ir * K + kk] = sum; else rs[ir] = sum;
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float sum = half_sum(e[q]);
    if (l31 == 0 && irow[q] < nunl) rs[irow[q]] = sum;
  }
  const float* bpr = a.bank_p[prob] + (long long)(jv ? jb : 0) * K;
  for (int k = 0; k < K; ++k) {
    const float pv = jv ? bpr[k] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float sum = half_sum(e[q] * pv);
      if (l31 == 0 && irow[q] < nunl) ep[(long long)irow[q] * K + k] = sum;
    }
  }
}

// 16-row variant of pair_exp_kernel for K <= 32 (the default): one workgroup per 16 x 32 tile on v_mfma_f32_16x16x4_f32
// (two column blocks), same K split over the four waves, same per-column-tile partials.  At B2 the 32 x 32 tiling makes
// 336 equal workgroups for 256 CUs -- 80 CUs get two and set the pace; 672 half-size workgroups spread as 3 / 2 per CU.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pair_exp16_kernel(LossArgs a) {
  constexpr int WT = 48 * PT;                                        // per-wave staging: A [16][PT] + B [32][PT]
  __shared__ __attribute__((aligned(16))) float lds[4 * WT];         // later reused as red[4][8][64], then E / p tiles
  float (*red)[8][64] = (float (*)[8][64])lds;
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = blockIdx.z;
  if (prob < 2 && !loss_smooth(a)) return;
  const int btu = a.btu, nunl = a.nunl, K = a.K;
  const float* A = feat_unl(a, prob == 0 ? 1 : 0, a.unl0);        // local rows (one rank's block: contiguous)
  const float* B = (prob == 0) ? a.bank_f[0] : (prob == 1) ? a.bank_f[1] : nullptr;   // prob 2: fU_w, all rows
  const int NB = (prob < 2) ? a.Q : btu;
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 32;
  if (c0 >= NB || r0 >= nunl) return;
  float* tA = lds + wave * WT;
  float* tB = tA + 16 * PT;
  // the bank-probability tile [32 columns][K] of the epilogue, requested now (K <= 32: at most 4 values per thread)
  float pq[4] = {0.f, 0.f, 0.f, 0.f};
  if (prob < 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 256 * q, c = i / K;
      if (i < 32 * K && c0 + c < NB) pq[q] = a.bank_p[prob][(long long)(c0 + c) * K + (i - c * K)];
    }
  }
  // loader role: lane -> (row group r8 = lane>>3, 16-byte chunk c8 = lane&7); load j covers rows 8j + r8
  const int r8 = lane >> 3, c8 = lane & 7;
  const long long ko = wave * 256 + c8 * 4;
#define CMLPL_ROWPTR(base, r, lim) ((base) + (long long)((r) < (lim) ? (r) : 0) * FD + ko)
#define CMLPL_BROW(r) ((B != nullptr ? B + (long long)((r) < NB ? (r) : 0) * FD : feat_unl(a, 1, (r) < NB ? (r) : 0)) + ko)
  const float* la0 = CMLPL_ROWPTR(A, r0 + r8, nunl);
  const float* la1 = CMLPL_ROWPTR(A, r0 + 8 + r8, nunl);
  const float* lb0 = CMLPL_BROW(c0 + r8);
  const float* lb1 = CMLPL_BROW(c0 + 8 + r8);
  const float* lb2 = CMLPL_BROW(c0 + 16 + r8);
  const float* lb3 = CMLPL_BROW(c0 + 24 + r8);
#undef CMLPL_BROW
#undef CMLPL_ROWPTR
  // lines are requested THREE ahead (a ring of four register sets, the loop fully unrolled so that the compiler counts
  // the outstanding loads): a line is only 16 MFMAs per wave, a bank row comes from HBM / the Infinity Cache, and one
  // line of look-ahead made every line wait out most of a memory round trip (8 round trips per workgroup)
  // (named register sets, pasted by macro: arrays indexed by the line number end up in scratch even when unrolled)
#define CMLPL_DECL(S) float4 a0##S, a1##S, b0##S, b1##S, b2##S, b3##S;
  CMLPL_DECL(_p) CMLPL_DECL(_q) CMLPL_DECL(_r) CMLPL_DECL(_s)
#undef CMLPL_DECL
#define CMLPL_LOADLINE(S, LN)                                                                              \
  { const int o_ = (LN) * 32;                                                                              \
    a0##S = *(const float4*)(la0 + o_); a1##S = *(const float4*)(la1 + o_);                                \
    b0##S = *(const float4*)(lb0 + o_); b1##S = *(const float4*)(lb1 + o_);                                \
    b2##S = *(const float4*)(lb2 + o_); b3##S = *(const float4*)(lb3 + o_); }
  CMLPL_LOADLINE(_p, 0) CMLPL_LOADLINE(_q, 1) CMLPL_LOADLINE(_r, 2)
  __builtin_amdgcn_sched_barrier(0);
  float* wA = tA + r8 * PT + c8 * 4;
  float* wB = tB + r8 * PT + c8 * 4;
  const float* rA = tA + l16 * PT + kq * 8;      // lane group kq owns floats [8kq, 8kq+8) of the line
  const float* rB = tB + l16 * PT + kq * 8;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#define CMLPL_M2(XA, YB, ZB)                                                                 \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(XA, YB, acc0, 0, 0, 0);                      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(XA, ZB, acc1, 0, 0, 0);
  // one line: request line LN + 3 into set N (free since line LN - 1), stage set C, multiply
#define CMLPL_LINE(LN, C, N)                                                                 \
  {                                                                                          \
    if ((LN) + 3 < 8) CMLPL_LOADLINE(N, (LN) + 3)                                            \
    __builtin_amdgcn_sched_barrier(0);   /* the scheduler would sink the loads to save registers */ \
    *(float4*)(wA) = a0##C; *(float4*)(wA + 8 * PT) = a1##C;                                 \
    *(float4*)(wB) = b0##C; *(float4*)(wB + 8 * PT) = b1##C;                                 \
    *(float4*)(wB + 16 * PT) = b2##C; *(float4*)(wB + 24 * PT) = b3##C;                      \
    const float4 x0 = *(const float4*)(rA), x1 = *(const float4*)(rA + 4);                   \
    const float4 y0 = *(const float4*)(rB), y1 = *(const float4*)(rB + 4);                   \
    const float4 z0 = *(const float4*)(rB + 16 * PT), z1 = *(const float4*)(rB + 16 * PT + 4); \
    CMLPL_M2(x0.x, y0.x, z0.x) CMLPL_M2(x0.y, y0.y, z0.y) CMLPL_M2(x0.z, y0.z, z0.z) CMLPL_M2(x0.w, y0.w, z0.w) \
    CMLPL_M2(x1.x, y1.x, z1.x) CMLPL_M2(x1.y, y1.y, z1.y) CMLPL_M2(x1.z, y1.z, z1.z) CMLPL_M2(x1.w, y1.w, z1.w) \
  }
  CMLPL_LINE(0, _p, _s) CMLPL_LINE(1, _q, _p) CMLPL_LINE(2, _r, _q) CMLPL_LINE(3, _s, _r)
  CMLPL_LINE(4, _p, _s) CMLPL_LINE(5, _q, _p) CMLPL_LINE(6, _r, _q) CMLPL_LINE(7, _s, _r)
#undef CMLPL_LINE
#undef CMLPL_M2
#undef CMLPL_LOADLINE
  __syncthreads();                               // every wave is done with its staging tile: reuse as `red`
#pragma unroll
  for (int r = 0; r < 4; ++r) { red[wave][r][lane] = acc0[r]; red[wave][4 + r][lane] = acc1[r]; }
  __syncthreads();
  // wave w finishes accumulator registers 2w, 2w+1: register g -> column block g >> 2, row 4 * kq + (g & 3)
  float e[2];
  int irow[2], col[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int g = wave * 2 + q;
    const float u = ((red[0][g][lane] + red[1][g][lane]) + red[2][g][lane]) + red[3][g][lane];
    col[q] = 16 * (g >> 2) + l16;
    irow[q] = 4 * kq + (g & 3);
    e[q] = (c0 + col[q] < NB) ? expf(u / a.T) : 0.f;
  }
  if (prob == 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (c0 + col[q] < NB && r0 + irow[q] < nunl) a.Smat[(long long)(r0 + irow[q]) * btu + c0 + col[q]] = e[q];
    return;
  }
  const int CT = (a.Q + 31) >> 5, ctile = c0 >> 5;
  float* rs = a.rs_part + ((long long)prob * CT + ctile) * nunl;
  float* ep = a.ep_part + ((long long)prob * CT + ctile) * nunl * K;
  // row sums and E . bank_probs as a small LDS product: the tile E [16][33] and the probability tile [32 cols][33] go
  // to LDS (`red` is dead after the barrier), each thread then forms whole outputs, columns summed in index order
  __syncthreads();                                   // all reads of `red` are done
  float* ew = lds;                                   // [16][33]
  float* pw = lds + 16 * 33;                         // [32][33]
#pragma unroll
  for (int q = 0; q < 2; ++q) ew[irow[q] * 33 + col[q]] = e[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid + 256 * q, c = i / K;
    if (i < 32 * K) pw[c * 33 + (i - c * K)] = pq[q];
  }
  __syncthreads();
  for (int o = tid; o < 16 * (K + 1); o += 256) {
    const int row = o & 15, kk = o >> 4;             // kk == K: the plain row sum
    const float* er = ew + row * 33;
    float sum = 0.f;
    if (kk < K) {
#pragma unroll 8
      for (int c = 0; c < 32; ++c) sum += er[c] * pw[c * 33 + kk];
    } else {
#pragma unroll 8
      for (int c = 0; c < 32; ++c) sum += er[c];
    }
    const int ir = r0 + row;
    if (ir < nunl) {
      if (kk < K) ep[(long long)ir * K + kk] = sum; else rs[ir] = sum;
    }
  }
}

// Tall-tile variant for wide products with MORE than 128 local rows (configs[2] whole on one or two GPUs; up to 128
// rows pair_exp_wide_kernel below is faster).  One workgroup = up to 128 local rows x 32 columns; wave w owns rows
// 32w..32w+31 over the WHOLE contraction, so there is no cross-wave reduction and the column tile is read once
// instead of once per 32 rows.  K is walked in 32-float lines staged in LDS (whole 128-B lines per row, row stride
// PT floats), double-buffered, one barrier per line, the next line's loads in flight in registers.  The products run
// on the split-bf16 MFMA (common.hpp): both fragments of a k-step are split in registers, six bf16 MFMAs per product
// (64.5 -> 57.7 us per launch at W = 8; the per-line barrier, not the matrix pipe, sets the pace now).
__global__ __launch_bounds__(256) void pair_exp_tall_kernel(LossArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2][(128 + 32) * PT];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = blockIdx.z;
  if (prob < 2 && !loss_smooth(a)) return;
  const int btu = a.btu, nunl = a.nunl, K = a.K;
  const float* A = feat_unl(a, prob == 0 ? 1 : 0, a.unl0);
  const float* B = (prob == 0) ? a.bank_f[0] : (prob == 1) ? a.bank_f[1] : nullptr;
  const int NB = (prob < 2) ? a.Q : btu;
  const int r0 = blockIdx.y * 128, c0 = blockIdx.x * 32;
  if (c0 >= NB || r0 >= nunl) return;
  const int jb = c0 + l31;
  // loader: thread -> (row lr = tid >> 3 of a 32-row group, 16-byte chunk c8 = tid & 7)
  const int lr = tid >> 3, c8 = tid & 7;
#define CMLPL_ROWPTR(base, r, lim) ((base) + (long long)((r) < (lim) ? (r) : 0) * FD + c8 * 4)
  const float* pa0 = CMLPL_ROWPTR(A, r0 + lr, nunl);
  const float* pa1 = CMLPL_ROWPTR(A, r0 + 32 + lr, nunl);
  const float* pa2 = CMLPL_ROWPTR(A, r0 + 64 + lr, nunl);
  const float* pa3 = CMLPL_ROWPTR(A, r0 + 96 + lr, nunl);
  const float* pb = (B != nullptr ? B + (long long)(c0 + lr < NB ? c0 + lr : 0) * FD
                                  : feat_unl(a, 1, c0 + lr < NB ? c0 + lr : 0)) + c8 * 4;
#undef CMLPL_ROWPTR
  // two register sets: set 0 carries the even lines, set 1 the odd ones, each requested TWO lines before it is
  // written to LDS (one line is only 16 MFMAs per wave -- not enough to cover an L2 round trip)
  float4 va0 = *(const float4*)pa0, va1 = *(const float4*)pa1, va2 = *(const float4*)pa2, va3 = *(const float4*)pa3;
  float4 vb = *(const float4*)pb;
  float4 wa0 = *(const float4*)(pa0 + 32), wa1 = *(const float4*)(pa1 + 32), wa2 = *(const float4*)(pa2 + 32),
         wa3 = *(const float4*)(pa3 + 32);
  float4 wb = *(const float4*)(pb + 32);
  const int wo = lr * PT + c8 * 4;
  f32x16 acc = zero16();
#define CMLPL_LINE(LN, R0, R1, R2, R3, RB)                                          \
  {                                                                                 \
    float* buf = lds[(LN) & 1];                                                     \
    *(float4*)(buf + wo) = R0; *(float4*)(buf + wo + 32 * PT) = R1;                 \
    *(float4*)(buf + wo + 64 * PT) = R2; *(float4*)(buf + wo + 96 * PT) = R3;       \
    *(float4*)(buf + wo + 128 * PT) = RB;                                           \
    if ((LN) + 2 < FD / 32) {                                                       \
      const int o = ((LN) + 2) * 32;                                                \
      R0 = *(const float4*)(pa0 + o); R1 = *(const float4*)(pa1 + o);               \
      R2 = *(const float4*)(pa2 + o); R3 = *(const float4*)(pa3 + o);               \
      RB = *(const float4*)(pb + o);                                                \
    }                                                                               \
    __syncthreads(); /* line staged; the other buffer (read during the previous line) is free */ \
    const float* rA = buf + (wave * 32 + l31) * PT + hh * 8;                        \
    const float* rB = buf + (128 + l31) * PT + hh * 8;                              \
    _Pragma("unroll")                                                               \
    for (int s16 = 0; s16 < 2; ++s16) {   /* split-bf16 MFMA: k = 16 s16 + 8 hh .. + 7 of this lane's row */ \
      uint4 A1, A2, A3, P1, P2, P3;                                                 \
      a_split(*(const float4*)(rA + 16 * s16), *(const float4*)(rA + 16 * s16 + 4), A1, A2, A3);              \
      a_split(*(const float4*)(rB + 16 * s16), *(const float4*)(rB + 16 * s16 + 4), P1, P2, P3);              \
      acc = mfma_b3(A1, A2, A3, P1, P2, P3, acc);                                   \
    }                                                                               \
  }
  // fully unrolled: across a loop back-edge hipcc loses count of the outstanding loads and waits vmcnt(0) before the
  // LDS writes, i.e. for the prefetch it has just issued
#pragma unroll
  for (int ln = 0; ln < FD / 32; ln += 2) {
    CMLPL_LINE(ln, va0, va1, va2, va3, vb)
    CMLPL_LINE(ln + 1, wa0, wa1, wa2, wa3, wb)
  }
#undef CMLPL_LINE
  // epilogue: this wave's 32 rows x 32 columns (lane = column jb, register r = row)
  const bool jv = jb < NB;
  const int rw = r0 + wave * 32;
  if (prob == 2) {
    if (rw >= nunl) return;
    if (jv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ir = rw + acc_row(r, lane);
        if (ir < nunl) a.Smat[(long long)ir * btu + jb] = expf(acc[r] / a.T);
      }
    }
    return;
  }
  const int CT = (a.Q + 31) >> 5, ctile = c0 >> 5;
  float* rs = a.rs_part + ((long long)prob * CT + ctile) * nunl;
  float* ep = a.ep_part + ((long long)prob * CT + ctile) * nunl * K;
  // Row sums and E . bank_probs of this wave's 32 x 32 tile.  Reducing over the 32 columns (= lanes) with
  // shuffles costs 5 ds_bpermute per (row, class): ~800 per wave here.  Instead E and the [32][K] probability
  // tile go to LDS (the staging buffers are free now) and each lane forms whole outputs: a [32x32].[32x(K+1)]
  // product, columns summed in index order.
  __syncthreads();                                       // every wave has finished its last line
  if (rw >= nunl) return;
  float* ew = &lds[0][0] + wave * (2 * 32 * 33);         // [32 rows][33]
  float* pw = ew + 32 * 33;                              // [32 cols][33], K <= 32 (else: shuffle path below)
#pragma unroll
  for (int r = 0; r < 16; ++r) ew[acc_row(r, lane) * 33 + l31] = jv ? expf(acc[r] / a.T) : 0.f;
  if (K <= 32) {
    for (int i = lane; i < 32 * K; i += 64) {
      const int c = i / K, k = i - c * K;
      pw[c * 33 + k] = (c0 + c < NB) ? a.bank_p[prob][(long long)(c0 + c) * K + k] : 0.f;
    }
    // wave-private region, but other lanes wrote what this lane reads: wait for the LDS writes
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    for (int o = lane; o < 32 * (K + 1); o += 64) {
      const int row = o & 31, kk = o >> 5;               // kk == K: the plain row sum
      const float* er = ew + row * 33;
      float sum = 0.f;
      if (kk < K) {
#pragma unroll 8
        for (int c = 0; c < 32; ++c) sum += er[c] * pw[c * 33 + kk];
      } else {
#pragma unroll 8
        for (int c = 0; c < 32; ++c) sum += er[c];
      }
      const int ir = rw + row;
      if (ir < nunl) {
        if (kk < K) ep[(long long)ir * K + kk] = sum; else rs[ir] = sum;
      }
    }
    return;
  }
  const float* bpr = a.bank_p[prob] + (long long)(jv ? jb : 0) * K;
  float e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    e[r] = jv ? expf(acc[r] / a.T) : 0.f;
    const float sum = half_sum(e[r]);
    const int ir = rw + acc_row(r, lane);
    if (l31 == 0 && ir < nunl) rs[ir] = sum;
  }
  for (int k = 0; k < K; ++k) {
    const float pv = jv ? bpr[k] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sum = half_sum(e[r] * pv);
      const int ir = rw + acc_row(r, lane);
      if (l31 == 0 && ir < nunl) ep[(long long)ir * K + k] = sum;
    }
  }
}

// Wide products (data parallelism: the banks hold 10 x the GLOBAL labelled batch, so a rank's local rows meet W x more
// columns -- 84 MB of bank rows at W = 8).  The tall tile above (128 x 32, four waves, one 32 x 32 MFMA tile each) re-reads
// the local rows once per 32 bank rows and splits the same 32 bank columns in each of its waves.
// Ablations of this kernel at W = 8 (57 us as is: 53 without its global loads, 36 without its MFMAs) say what bounds
// both: not HBM but the operand traffic through LDS -- every six MFMAs want three 16-byte fragment reads per lane, and
// neither more waves per SIMD, more workgroups per CU nor deeper prefetch changed the time.  Here:
//   * one workgroup = all local rows (MT = 128, or 64) x NT = 32..128 bank columns, NT chosen on the host so that the
//     launch is ONE round of about one workgroup per CU;
//   * EIGHT waves: MB row blocks x CG column groups x the two halves of every 32-float line (k-half kh: a wave
//     multiplies 16 of a line's 32 k into its 32 rows x 32 NBW columns), so that each SIMD holds two waves whose LDS /
//     vector phases run under each other's MFMAs; the two k-halves are folded through LDS once, after the last line;
//   * the contraction is walked in 32-float lines (whole 128-B lines per row and load instruction), three lines in
//     flight in registers, double-buffered LDS stages, one barrier per line;
//   * a bank line is split into its three bf16 planes ONCE, by the thread that loaded it, on its way into LDS
//     ([plane][column][32 k], row stride 20 dwords: conflict-free ds_read_b128 fragments); a wave splits only its own
//     rows and k-half of the local features (read as fp32 from LDS);
//   * partial row sums / E . bank_probs are written per wave tile (32 NBW columns: LossArgs.ctw), not per 32 columns, so
//     the row kernel folds NBW times fewer partials.
// Same six bf16 products per fp32 product as the convolutions (common.hpp).
// staging helpers: this thread's pieces of a line, registers -> LDS stage
// (A: rebuilt member by member -- storing wa[q] itself is a 16-byte struct copy out of the array, and with it the compiler
//  kept the A sets in scratch, 144 bytes per lane)
template <int NA, int RP>
__device__ __forceinline__ void wide_stage_a(float* st, const float4 (&wa)[NA], int lr, int c8) {
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const float4 t = wa[q];
    *(float4*)(st + (lr + RP * q) * PT + c8 * 4) = make_float4(t.x, t.y, t.z, t.w);
  }
}
// four consecutive k of bank column `col` -> the three bf16 planes ([NTP columns][BPS dwords] each)
template <int NTP, int BPS>
__device__ __forceinline__ void wide_stage_b(uint32_t* bpl, const float4& v, int col, int c8) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t u0[4], u1[4], u2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u0[j] = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(u0[j] & 0xffff0000u);
    u1[j] = __float_as_uint(r1);
    u2[j] = __float_as_uint(r1 - __uint_as_float(u1[j] & 0xffff0000u));
  }
  uint32_t* d = bpl + col * BPS + c8 * 2;
  *(uint2*)(d) = make_uint2(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]));
  *(uint2*)(d + NTP * BPS) = make_uint2(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]));
  *(uint2*)(d + 2 * NTP * BPS) = make_uint2(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]));
}

template <int MB, int CG, int NBW>
struct WideCfg {
  static constexpr int NW = 2 * MB * CG, NTHR = 64 * NW;        // waves (two k-halves), threads
  static constexpr int MT = 32 * MB, NT = 32 * CG * NBW;        // local rows, bank columns per workgroup
  static constexpr int RP = NTHR / 8;                           // rows per loader pass (8 threads x 16 B per line)
  static constexpr int NA = (MT + RP - 1) / RP, NBL = (NT + RP - 1) / RP;   // 16-byte pieces per thread and line
  static constexpr int MTP = NA * RP, NTP = NBL * RP;           // rows held in LDS (>= MT, NT: the loader's passes are whole)
  static constexpr int BPS = 20;                                // dwords per column of a B plane: 16 (32 bf16) + 4 pad
  static constexpr int A_FL = MTP * PT, B_DW = 3 * NTP * BPS, STAGE = A_FL + B_DW;
  static constexpr size_t FOLD = (size_t)MB * CG * NBW * 16 * 64;          // floats: the k-half exchange
  static constexpr size_t EPI = (size_t)MB * CG * 2 * 32 * 33;             // floats: per-wave E / p tiles
  static constexpr size_t LDS_FL = 2 * STAGE > FOLD + EPI ? 2 * STAGE : FOLD + EPI;
};

template <int MB, int CG, int NBW>
__global__ __launch_bounds__(128 * MB * CG) void pair_exp_wide_kernel(LossArgs a) {
  typedef WideCfg<MB, CG, NBW> Cf;
  constexpr int MT = Cf::MT, NT = Cf::NT, RP = Cf::RP, NA = Cf::NA, NBL = Cf::NBL, NTP = Cf::NTP, BPS = Cf::BPS;
  constexpr int A_FL = Cf::A_FL, STAGE = Cf::STAGE;
  constexpr int NL = FD / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][STAGE]; then the k-half exchange and the E / p tiles
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave / (MB * CG), w2 = wave - kh * (MB * CG);   // k-half; (row block, column group)
  const int rb = w2 / CG, cg = w2 - rb * CG;
  const int prob = blockIdx.z;
  if (prob < 2 && !loss_smooth(a)) return;
  const int btu = a.btu, nunl = a.nunl, K = a.K;
  const float* A = feat_unl(a, prob == 0 ? 1 : 0, a.unl0);
  const float* B = (prob == 0) ? a.bank_f[0] : (prob == 1) ? a.bank_f[1] : nullptr;
  const int NB = (prob < 2) ? a.Q : btu;
  const int r0 = blockIdx.y * MT, c0 = blockIdx.x * NT;
  if (c0 >= NB || r0 >= nunl) return;
  // loader: piece q of this thread = row (tid >> 3) + RP q of the A (B) tile, 16-byte chunk c8 = tid & 7 of the line
  // (rows past the tile or the matrix are clamped: they land in LDS rows nobody multiplies)
  const int lr = tid >> 3, c8 = tid & 7;
  const float* pa[NA];
  const float* pb[NBL];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int r = r0 + lr + RP * q;
    pa[q] = A + (long long)((r < nunl && lr + RP * q < MT) ? r : 0) * FD + c8 * 4;
  }
#pragma unroll
  for (int q = 0; q < NBL; ++q) {
    const int c = c0 + lr + RP * q, cc = (c < NB && lr + RP * q < NT) ? c : c0;
    pb[q] = (B != nullptr ? B + (long long)cc * FD : feat_unl(a, 1, cc)) + c8 * 4;
  }
  // three register sets: the set of line x is x % 3, requested three lines before it is staged
  // (named arrays, not one indexed by the line: that one ends up in scratch even when fully unrolled)
  float4 va0[NA], vb0[NBL], va1[NA], vb1[NBL], va2[NA], vb2[NBL];
#pragma unroll
  for (int q = 0; q < NA; ++q) { va0[q] = *(const float4*)(pa[q]); va1[q] = *(const float4*)(pa[q] + 32); va2[q] = *(const float4*)(pa[q] + 64); }
#pragma unroll
  for (int q = 0; q < NBL; ++q) { vb0[q] = *(const float4*)(pb[q]); vb1[q] = *(const float4*)(pb[q] + 32); vb2[q] = *(const float4*)(pb[q] + 64); }
  f32x16 acc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) acc[nb] = zero16();
  // line 0 goes into stage 0 before the loop (both tiles); line 3 is requested into set 0
  {
    wide_stage_a<NA, RP>(lds, va0, lr, c8);
#pragma unroll
    for (int q = 0; q < NBL; ++q) wide_stage_b<NTP, BPS>((uint32_t*)(lds + A_FL), vb0[q], lr + RP * q, c8);
#pragma unroll
    for (int q = 0; q < NA; ++q) va0[q] = *(const float4*)(pa[q] + 96);
#pragma unroll
    for (int q = 0; q < NBL; ++q) vb0[q] = *(const float4*)(pb[q] + 96);
  }
  // One line of the contraction = ONE barrier (line LN is staged; the other stage, read during line LN - 1, is free), then
  // this wave's NBW units (column block nb, its k-half) of six MFMAs each.  Everything else rides in the units' shadow, in
  // program order pinned by scheduling fences (left alone, the compiler stages first, multiplies afterwards and issues
  // the prefetch last): the fragment reads of unit u + 1 in front of unit u's MFMAs; line LN + 1's pieces split and
  // stored into the other stage, one bank piece per unit; the request for line LN + 4 as soon as the set's last piece
  // has been staged.  (Row blocks past the local rows multiply clamped rows like everyone else.)
  constexpr int ULAST = (NBL - 1 < NBW - 1) ? NBL - 1 : NBW - 1;   // unit that stages the set's last bank piece
#define CMLPL_WLINE(LN, WA, WB) {\
    __syncthreads();\
    const float* st = lds + ((LN) & 1) * STAGE;\
    float* sn = lds + (((LN) + 1) & 1) * STAGE;\
    uint32_t* bpn = (uint32_t*)(sn + A_FL);\
    const float* rA = st + (rb * 32 + l31) * PT + 16 * kh + hh * 8;\
    const uint32_t* rB = (const uint32_t*)(st + A_FL) + (cg * NBW * 32 + l31) * BPS + 8 * kh + hh * 4;\
    const float4 x0 = *(const float4*)(rA), x1 = *(const float4*)(rA + 4);\
    uint4 P1 = *(const uint4*)rB, P2 = *(const uint4*)(rB + NTP * BPS), P3 = *(const uint4*)(rB + 2 * NTP * BPS);\
    wide_stage_a<NA, RP>(sn, WA, lr, c8);\
    uint4 A1, A2, A3;\
    a_split(x0, x1, A1, A2, A3);\
    __builtin_amdgcn_sched_barrier(0);\
_Pragma("unroll")\
    for (int u = 0; u < NBW; ++u) {\
      uint4 N1 = P1, N2 = P2, N3 = P3;\
      if (u + 1 < NBW) {\
        const uint32_t* pp = rB + (u + 1) * 32 * BPS;\
        N1 = *(const uint4*)pp; N2 = *(const uint4*)(pp + NTP * BPS); N3 = *(const uint4*)(pp + 2 * NTP * BPS);\
      }\
      acc[u] = mfma_b3(A1, A2, A3, P1, P2, P3, acc[u]);\
_Pragma("unroll")\
      for (int q = 0; q < NBL; ++q)\
        if (q == u || (u == NBW - 1 && q > u)) wide_stage_b<NTP, BPS>(bpn, WB[q], lr + RP * q, c8);\
      if (u == ULAST && (LN) + 4 < NL) {\
        const int o = ((LN) + 4) * 32;\
_Pragma("unroll")\
        for (int q = 0; q < NA; ++q) WA[q] = *(const float4*)(pa[q] + o);\
_Pragma("unroll")\
        for (int q = 0; q < NBL; ++q) WB[q] = *(const float4*)(pb[q] + o);\
      }\
      P1 = N1; P2 = N2; P3 = N3;\
      if (u + 1 < NBW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);\
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);\
      SchedInterleave<5>::run();\
      __builtin_amdgcn_sched_barrier(0);\
    }\
  }
  // Written out 32 times (not a loop under #pragma unroll: beyond ~16 K instructions the compiler unrolls only partly,
  // and across a loop back-edge it loses count of the outstanding loads and waits vmcnt(0) in front of the staging,
  // i.e. for the prefetch it has just issued).  Line L stages line L + 1 from set (L + 1) % 3 and requests line L + 4.
  static_assert(NL == 32, "the line sequence below is written out for FD = 1024");
#define CMLPL_W3(L) CMLPL_WLINE(L, va1, vb1) CMLPL_WLINE((L) + 1, va2, vb2) CMLPL_WLINE((L) + 2, va0, vb0)
  CMLPL_W3(0) CMLPL_W3(3) CMLPL_W3(6) CMLPL_W3(9) CMLPL_W3(12) CMLPL_W3(15) CMLPL_W3(18) CMLPL_W3(21) CMLPL_W3(24) CMLPL_W3(27)
  CMLPL_WLINE(30, va1, vb1) CMLPL_WLINE(31, va2, vb2)
#undef CMLPL_W3
#undef CMLPL_WLINE
  // ---- fold the two k-halves: waves kh = 1 hand their accumulators to their partner through LDS
  const int rw = r0 + rb * 32;
  const int cw = c0 + cg * NBW * 32;                      // first column of this wave
  // the bank-probability tiles of the epilogue ([NBW][32 columns][K] of this wave), requested before the exchange
  constexpr int PQN = 16;                                 // K <= 32: 32 K / 64 values per lane and column block
  float pq[NBW][PQN];
  if (kh == 0 && prob < 2) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int i = 0; i < PQN; ++i) {
        const int idx = lane + 64 * i, c = idx / K, col = cw + nb * 32 + c;
        float v = 0.f;
        if (idx < 32 * K && col < NB) v = a.bank_p[prob][(long long)col * K + (idx - c * K)];
        pq[nb][i] = v;
      }
  }
  __syncthreads();                                        // every wave has finished its last line: the stages are free
  float* xch = lds + (size_t)w2 * NBW * 16 * 64;
  if (kh == 1) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(nb * 16 + r) * 64 + lane] = acc[nb][r];
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] += xch[(nb * 16 + r) * 64 + lane];
  // ---- epilogue: this wave's 32 rows x NBW x 32 columns (lane = column, register r = row acc_row(r))
  if (rw >= nunl || cw >= NB) return;
  if (prob == 2) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int jb = cw + nb * 32 + l31;
      if (jb < NB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ir = rw + acc_row(r, lane);
          if (ir < nunl) a.Smat[(long long)ir * btu + jb] = expf(acc[nb][r] / a.T);
        }
      }
    }
    return;
  }
  // Row sums and E . bank_probs over this wave's columns: per 32-column block E [32][33] and the probability tile
  // [32 columns][33] go to the wave's own LDS region (behind the exchange) and each lane forms whole outputs (row, class |
  // row sum), columns summed in index order, blocks in order -- a [32 x 32 NBW] . [32 NBW x (K + 1)] product without
  // shuffles.
  float* ew = lds + Cf::FOLD + (size_t)w2 * (2 * 32 * 33);
  float* pw = ew + 32 * 33;
  constexpr int ON = (32 * 33 + 63) / 64;                 // outputs per lane: 32 rows x (K + 1 <= 33)
  float outv[ON];
#pragma unroll
  for (int i = 0; i < ON; ++i) outv[i] = 0.f;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const bool jv = cw + nb * 32 + l31 < NB;
#pragma unroll
    for (int r = 0; r < 16; ++r) ew[acc_row(r, lane) * 33 + l31] = jv ? expf(acc[nb][r] / a.T) : 0.f;
#pragma unroll
    for (int i = 0; i < PQN; ++i) {
      const int idx = lane + 64 * i, c = idx / K;
      if (idx < 32 * K) pw[c * 33 + (idx - c * K)] = pq[nb][i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): wave-private region, other lanes wrote what this lane reads
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < ON; ++i) {
      const int o = lane + 64 * i;
      if (o < 32 * (K + 1)) {
        const int row = o & 31, kk = o >> 5;               // kk == K: the plain row sum
        const float* er = ew + row * 33;
        float sum = 0.f;
        if (kk < K) {
#pragma unroll 8
          for (int c = 0; c < 32; ++c) sum += er[c] * pw[c * 33 + kk];
        } else {
#pragma unroll 8
          for (int c = 0; c < 32; ++c) sum += er[c];
        }
        outv[i] += sum;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();                      // every lane is done reading before the next block's tiles land
  }
  const int CT = (a.Q + a.ctw - 1) / a.ctw, ctile = cw / (32 * NBW);
  float* rs = a.rs_part + ((long long)prob * CT + ctile) * nunl;
  float* ep = a.ep_part + ((long long)prob * CT + ctile) * nunl * K;
#pragma unroll
  for (int i = 0; i < ON; ++i) {
    const int o = lane + 64 * i;
    if (o < 32 * (K + 1)) {
      const int row = o & 31, kk = o >> 5, ir = rw + row;
      if (ir < nunl) {
        if (kk < K) ep[(long long)ir * K + kk] = outv[i]; else rs[ir] = outv[i];
      }
    }
  }
}

// Bank write of the GLOBAL batch (train.py:223-236), one workgroup per written row, identical on every shard:
//   bank0 <- [fU_w ; fL_s], [p_w0 ; onehot]     bank1 <- [fU_s ; fL_w], [p_s0 ; onehot]      (rows modulo Q)
// It runs in the loss_rows launch: pair_exp_kernel (the launch before) was the last reader of the banks, and the
// un-smoothed probabilities p_w0 / p_s0 are just the softmax of the logits row, re-formed here with the same
// instructions loss_rows_kernel uses (bit-identical).
__device__ __forceinline__ void bank_write_block(const LossArgs& a, int r) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int btu = a.btu, K = a.K, Q = a.Q;
  int p0 = a.ptr0, p1 = a.ptr1;
  const cmlpl_dyn* dynr = dyn_row(a.sel.dyn);
  if (dynr != nullptr) { p0 = uni32(dynr->ptr[0]); p1 = uni32(dynr->ptr[1]); }
  const int d0 = (p0 + r) % Q, d1 = (p1 + r) % Q;
  const float *s0, *s1;
  if (r < btu) { s0 = feat_unl(a, 1, r); s1 = feat_unl(a, 0, r); }
  else         { s0 = feat_lab(a, 0, r - btu); s1 = feat_lab(a, 1, r - btu); }
  const float4 v0 = ((const float4*)s0)[tid], v1 = ((const float4*)s1)[tid];
  const float NEG = -3.0e38f;
  const bool kv = lane < K;
  float zs = NEG, zw = NEG;
  int yl = -1;
  // packed mode (the sharded step): the logits of other ranks' rows are not here, but their un-smoothed probabilities
  // are -- the row kernel wrote p_w0 / p_s0 (the same instructions as below) into the gathered probabilities
  const bool from_probs = a.recv_f != nullptr;
  if (tid < 64) {                                          // requested before the feature stores, not behind them
    if (r < btu) {
      if (kv) {
        if (from_probs) { zw = prob_row(a, 2, r)[lane]; zs = prob_row(a, 3, r)[lane]; }
        else { zs = logit_unl(a, 0, r)[lane]; zw = logit_unl(a, 1, r)[lane]; }
      }
    } else yl = label_of(a, r - btu);
  }
  ((float4*)(a.bank_fw[0] + (long long)d0 * FD))[tid] = v0;
  ((float4*)(a.bank_fw[1] + (long long)d1 * FD))[tid] = v1;
  if (tid < 64) {
    float q0, q1;
    if (r < btu && from_probs) {
      q0 = zw; q1 = zs;
    } else if (r < btu) {
      const float mxs = wave_max(zs), mxw = wave_max(zw);
      const float es = kv ? expf(zs - mxs) : 0.f, ew = kv ? expf(zw - mxw) : 0.f;
      const float ses = wave_sum(es), sew = wave_sum(ew);
      q0 = ew / sew; q1 = es / ses;                      // "probs" (Base1) -> bank0, "probs1" (Base) -> bank1
    } else {
      q0 = q1 = (lane == yl) ? 1.f : 0.f;
    }
    if (kv) {
      a.bank_pw[0][(long long)d0 * K + lane] = q0;
      a.bank_pw[1][(long long)d1 * K + lane] = q1;
    }
  }
}

__global__ __launch_bounds__(256) void loss_rows_kernel(LossArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bt = a.bt, btu = a.btu, K = a.K;
  const int nlab = a.nlab, nunl = a.nunl, nl = nlab + nunl;
  const int row_blocks = (nl + 3) >> 2;
  if ((int)blockIdx.x >= row_blocks) { bank_write_block(a, (int)blockIdx.x - row_blocks); return; }
  const int idx = blockIdx.x * 4 + wave;
  if (idx >= nl) return;
  const int RL = nlab > nunl ? nlab : nunl;
  const bool kv = lane < K;
  const float NEG = -3.0e38f;
  int smooth = a.smooth;
  float adap_mask = a.adap_mask;
  const cmlpl_dyn* dynr = dyn_row(a.sel.dyn);
  if (dynr != nullptr) { smooth = uni32(dynr->smooth); adap_mask = __int_as_float(uni32(__float_as_int(dynr->adap_mask))); }
  if (idx < nlab) {
    const int il = idx, ig = a.lab0 + il;                 // local / global labelled row
    const int yl = label_of(a, ig);
    const float zin[2] = {kv ? logit_lab(a, 0, ig)[lane] : NEG, kv ? logit_lab(a, 1, ig)[lane] : NEG};   // both before any store
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      const float z = zin[net];
      const float mx = wave_max(z);
      const float ez = kv ? expf(z - mx) : 0.f;
      const float se = wave_sum(ez);
      const float lse = mx + logf(se);
      const float zy = __shfl(z, yl, 64);
      if (kv) a.dlogits[((long long)net * nl + il) * K + lane] = (ez / se - (lane == yl ? 1.f : 0.f)) / (float)bt;
      {  // labelled rows carry no feature gradient (they only enter the banks)
        float4* dz = (float4*)(a.dfeat + ((long long)net * nl + il) * FD);
#pragma unroll
        for (int q = 0; q < 4; ++q) dz[lane + 64 * q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (lane == 0) a.rowloss[(net == 0 ? RL_CLS_S : RL_CLS_W) * RL + il] = lse - zy;
      if (net == 1) {  // torch.max(labeled_output1, 1): first index of the maximum (train.py:194)
        // a NaN logit is the maximum for torch.max, and the first one wins (dead-ReLU regime, models.py:87-90)
        const unsigned long long nanb = __ballot(kv && z != z);
        const unsigned long long bal = nanb ? nanb : __ballot(kv && z == mx);
        const int amax = __ffsll((long long)bal) - 1;
        if (lane == 0) a.rowloss[RL_ACC * RL + il] = (amax == yl) ? 1.f : 0.f;
      }
    }
    return;
  }
  const int i = idx - nlab, ig = a.unl0 + i;               // local / global unlabelled row
  const float zs = kv ? logit_unl(a, 0, ig)[lane] : NEG;
  const float zw = kv ? logit_unl(a, 1, ig)[lane] : NEG;
  // The per-column-tile partials of pair_exp_kernel are requested BEFORE the softmax arithmetic (they do not depend on
  // it; loaded after it, the row paid three memory round trips in a row: logits, row sums, E.p partials).  Lanes are
  // split into 64/KP groups of KP >= K lanes; group gq takes tiles ct = gq, gq+G, ... for class (lane % KP), LRD loads
  // of each product in flight at a time.
  constexpr int LRD = 16;
  const int CT = (a.Q + a.ctw - 1) / a.ctw;
  int KP = 1;
  while (KP < K) KP <<= 1;
  const int G = 64 / KP, gq = lane / KP, kq = lane - gq * KP;
  const bool kqv = kq < K;
  const long long cstride = (long long)nunl * K;
  const float* e0 = a.ep_part + ((long long)0 * CT * nunl + i) * K + (kqv ? kq : 0);
  const float* e1 = a.ep_part + ((long long)1 * CT * nunl + i) * K + (kqv ? kq : 0);
  float t0[LRD], t1[LRD], r0 = 0.f, r1 = 0.f;
  if (smooth) {
#pragma unroll
    for (int q = 0; q < LRD; ++q) {
      const int ct = gq + q * G;
      const bool ok = ct < CT;
      const float x0 = e0[(long long)(ok ? ct : 0) * cstride], x1 = e1[(long long)(ok ? ct : 0) * cstride];
      t0[q] = ok ? x0 : 0.f; t1[q] = ok ? x1 : 0.f;
    }
    if (lane < CT) {
      r0 = a.rs_part[((long long)0 * CT + lane) * nunl + i];
      r1 = a.rs_part[((long long)1 * CT + lane) * nunl + i];
    }
  }
  const float mxs = wave_max(zs), mxw = wave_max(zw);
  const float es = kv ? expf(zs - mxs) : 0.f, ew = kv ? expf(zw - mxw) : 0.f;
  const float ses = wave_sum(es), sew = wave_sum(ew);
  const float sms = es / ses, smw = ew / sew;            // softmax
  const float lsms = zs - mxs - logf(ses), lsmw = zw - mxw - logf(sew);  // log_softmax
  float pw = smw, ps = sms;                              // "probs" (Base1) / "probs1" (Base)
  if (smooth) {
    float rsw = r0, rss = r1;
    for (int ct = lane + 64; ct < CT; ct += 64) {
      rsw += a.rs_part[((long long)0 * CT + ct) * nunl + i];
      rss += a.rs_part[((long long)1 * CT + ct) * nunl + i];
    }
    rsw = wave_sum(rsw); rss = wave_sum(rss);
    float epw = 0.f, eps_ = 0.f;
#pragma unroll
    for (int q = 0; q < LRD; ++q) { epw += t0[q]; eps_ += t1[q]; }
    for (int cb = gq + LRD * G; cb < CT; cb += LRD * G) {      // banks of more than LRD * G column tiles (data parallelism)
#pragma unroll
      for (int q = 0; q < LRD; ++q) {
        const int ct = cb + q * G;
        const bool ok = ct < CT;
        const float x0 = e0[(long long)(ok ? ct : 0) * cstride], x1 = e1[(long long)(ok ? ct : 0) * cstride];
        t0[q] = ok ? x0 : 0.f; t1[q] = ok ? x1 : 0.f;
      }
#pragma unroll
      for (int q = 0; q < LRD; ++q) { epw += t0[q]; eps_ += t1[q]; }
    }
    for (int o = KP; o < 64; o <<= 1) { epw += __shfl_xor(epw, o, 64); eps_ += __shfl_xor(eps_, o, 64); }
    // lanes 0..K-1 (group 0) now hold the full sums for class = lane
    pw = a.alpha * smw + (1.f - a.alpha) * (epw / rsw);
    ps = a.alpha * sms + (1.f - a.alpha) * (eps_ / rss);
  }
  const float mw = (wave_max(kv ? pw : NEG) >= adap_mask) ? 1.f : 0.f;   // "mask"  (train.py:222)
  const float ms = (wave_max(kv ? ps : NEG) >= adap_mask) ? 1.f : 0.f;   // "masks" (train.py:228)
  const float spw = wave_sum(kv ? pw : 0.f), sps = wave_sum(kv ? ps : 0.f);
  const float cs = -wave_sum(kv ? lsms * pw : 0.f) * mw;   // train.py:239
  const float cw = -wave_sum(kv ? lsmw * ps : 0.f) * ms;   // train.py:240
  if (kv) {
    const float scale = a.w_mutual / (float)btu;
    a.dlogits[((long long)nlab + i) * K + lane] = scale * mw * (sms * spw - pw);
    a.dlogits[((long long)nl + nlab + i) * K + lane] = scale * ms * (smw * sps - ps);
    a.probs_l[((long long)0 * nunl + i) * K + lane] = pw;
    a.probs_l[((long long)1 * nunl + i) * K + lane] = ps;
    a.probs_l[((long long)2 * nunl + i) * K + lane] = smw;
    a.probs_l[((long long)3 * nunl + i) * K + lane] = sms;
  }
  if (lane == 0) {
    a.masks[i] = mw;
    a.masks[nunl + i] = ms;
    a.rowloss[RL_CON_S * RL + i] = cs;
    a.rowloss[RL_CON_W * RL + i] = cw;
  }
}

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// N sums at once: one pair of barriers instead of N (the same fold order as block_sum: lanes, then waves 0..3)
template <int N>
__device__ __forceinline__ void block_sum_n(float (&v)[N], float* red, int tid) {
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_sum(v[q]);
  __syncthreads();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < N; ++q) red[(tid >> 6) * N + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = red[q] + red[N + q] + red[2 * N + q] + red[3 * N + q];
}

// One workgroup per local unlabelled row (under data parallelism a row meets W x more columns; a single wavefront
// walking them in three passes was 32 us at W = 8): the 256 threads stride the columns, sums are folded per wave
// and then across the four waves in a fixed order.
__global__ __launch_bounds__(256) void graph_loss_kernel(LossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // gq[btu], pp[btu]
  __shared__ float red[32];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.nunl) { bank_write_block(a, (int)blockIdx.x - a.nunl); return; }   // (packed mode: see launch_loss_graph)
  const int btu = a.btu, K = a.K, i = blockIdx.x;               // local row
  const int ig = a.unl0 + i;                                    // global row (diagonal position)
  const int RL = a.nlab > a.nunl ? a.nlab : a.nunl;
  float* gq = smem;
  float* pp = gq + btu;
  const float* psi = prob_row(a, 1, ig);                        // smoothed p_s ("probs1"), row i
  const float* Srow = a.Smat + (long long)i * btu;
  float sQ = 0.f, sN = 0.f, R = 0.f, npos = 0.f, nneg = 0.f;
  for (int j = tid; j < btu; j += 256) {
    const float* pwj = prob_row(a, 0, j);                       // smoothed p_w ("probs"), row j
    float q0 = 0.f;
    for (int k = 0; k < K; ++k) q0 = fmaf(psi[k], pwj[k], q0);                      // train.py:249
    if (j == ig) q0 = 1.f;                                                           // :250
    const bool isp = q0 >= a.pos_thr, isn = q0 <= a.neg_thr;                        // :251,254
    const float pos = isp ? q0 : 0.f, neg = isn ? (1.f - q0) : 0.f;
    sQ += pos; sN += neg; R += Srow[j];
    npos += isp ? 1.f : 0.f; nneg += isn ? 1.f : 0.f;
    gq[j] = pos; pp[j] = neg;
  }
  {
    float v5[5] = {sQ, sN, R, npos, nneg};
    block_sum_n<5>(v5, red, tid);
    sQ = v5[0]; sN = v5[1]; R = v5[2]; npos = v5[3]; nneg = v5[4];
  }
  float lp = 0.f, ln = 0.f, gp = 0.f;
  const float inv_btu = 1.f / (float)btu;
  for (int j = tid; j < btu; j += 256) {            // each thread revisits exactly the columns it wrote
    const float P = Srow[j] / R;                    // sim_probs (:247)
    const float Qv = gq[j] / sQ;                    // :253
    const float Qn = pp[j] / (sN + 1e-8f);          // :256
    lp -= logf(P) * Qv;                             // :260
    ln += logf(P + 1.f) * Qn;                       // :261
    const float g = (-Qv / P + Qn / (1.f + P)) * inv_btu;   // dL/dP_ij
    gp = fmaf(g, P, gp);
    gq[j] = g; pp[j] = P;
  }
  {
    float v3[3] = {lp, ln, gp};
    block_sum_n<3>(v3, red, tid);
    lp = v3[0]; ln = v3[1]; gp = v3[2];
  }
  const float scale = a.w_contrast / a.T;
  for (int j = tid; j < btu; j += 256) {
    const float Gij = pp[j] * (gq[j] - gp) * scale;   // softmax backward, then d(sim)/d(f.f/T)
    a.G[(long long)i * btu + j] = Gij;                // [nunl][btu]
    a.GT[(long long)j * a.nunl + i] = Gij;            // [btu][nunl]
  }
  if (tid == 0) {
    a.rowloss[RL_CTR * RL + i] = lp + ln;
    a.rowloss[RL_NPOS * RL + i] = npos;
    a.rowloss[RL_NNEG * RL + i] = nneg;
  }
}

// This shard's share of the logged scalars (train.py:266,270,274-278): local sums over GLOBAL counts, additive
// across shards.  One workgroup, appended to the launch of the two feature-gradient GEMMs (it only needs the
// per-row losses of the launches before).
__device__ __forceinline__ void loss_scalars_block(const LossArgs& a, float* red) {
  const int tid = threadIdx.x;
  const int bt = a.bt, btu = a.btu;
  const int RL = a.nlab > a.nunl ? a.nlab : a.nunl;
  // every per-thread partial first (ten independent strided loads, one memory round trip), then ONE folded block sum
  // (round 3 ran ten load -> two-barrier sums in a row: this single workgroup was the long pole of its launch)
  float v[RL_COUNT + 2];
#pragma unroll
  for (int q = 0; q < RL_COUNT; ++q) {
    const int cnt = (q <= RL_ACC) ? a.nlab : a.nunl;
    float s = 0.f;
    for (int i = tid; i < cnt; i += 256) s += a.rowloss[q * RL + i];
    v[q] = s;
  }
  float mws = 0.f, mss = 0.f;
  for (int i = tid; i < a.nunl; i += 256) { mws += a.masks[i]; mss += a.masks[a.nunl + i]; }
  v[RL_COUNT] = mws; v[RL_COUNT + 1] = mss;
  block_sum_n<RL_COUNT + 2>(v, red, tid);
  mws = v[RL_COUNT]; mss = v[RL_COUNT + 1];
  if (tid == 0) {
    const float cls_s = v[RL_CLS_S] / bt, cls_w = v[RL_CLS_W] / bt, acc = v[RL_ACC] / bt;
    const float con_s = v[RL_CON_S] / btu, con_w = v[RL_CON_W] / btu, ctr = v[RL_CTR] / btu;
    int hrow = 0;
    const cmlpl_dyn* dynr = dyn_row(a.sel.dyn);
    if (dynr != nullptr) hrow = dynr->hist_row;
    float* o = a.scalars + 16 * hrow;
    o[0] = ctr;                                              // loss_contrast  (train.py:274)
    o[1] = cls_s + a.w_contrast * ctr + a.w_mutual * con_s;  // total_loss     (:266,275)
    o[2] = cls_s;                                            // (:276)
    o[3] = con_s;                                            // (:277)
    o[4] = acc;                                              // (:278)
    o[5] = cls_w + a.w_contrast * ctr + a.w_mutual * con_w;  // total_loss1    (:270)
    o[6] = cls_w; o[7] = con_w; o[8] = ctr;                  // loss_contrast1 == loss_contrast numerically
    o[9] = mws; o[10] = mss; o[11] = v[RL_NPOS]; o[12] = v[RL_NNEG];
    o[13] = 0.f; o[14] = 0.f; o[15] = 0.f;
  }
}

// The two feature-gradient GEMMs with their operands through LDS: one workgroup per 16 x 64 output tile (wave w: columns
// 16 w .. 16 w + 15 on v_mfma_f32_16x16x4_f32), the contraction walked in chunks of DF_KC rows that are fetched as whole
// 16-byte pieces (A: one row of 16 floats = 64 B per quarter-wave; B: 256 B per row), two chunks in flight in registers,
// double-buffered in LDS, one barrier per chunk.  gemm_tn_block feeds its MFMAs straight from 4-byte global loads -- fine
// while the contraction is one batch of loads long (one GPU: 128 rows), 25 us at eight ranks (1024 rows: a thousand
// dword loads per wave).  No bias / batch / ReLU epilogue: C = A^T . B.
constexpr int DF_KC = 64, DF_AS = 16, DF_BS = 80;       // chunk rows; LDS row strides in floats (16 apart modulo 32 banks: the four k of
                                                         // an MFMA operand read land on disjoint banks two by two)
struct DfeatShared { __attribute__((aligned(16))) float a[2][DF_KC * DF_AS]; __attribute__((aligned(16))) float b[2][DF_KC * DF_BS]; };
inline int dfeat_lds_blocks(const GemmTN& g) { return ((g.M + 15) / 16) * ((g.N + 63) / 64); }

__device__ __forceinline__ void dfeat_lds_block(const GemmTN2& t, int bid, DfeatShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pi = (bid >= t.nblk0) ? 1 : 0;
  const GemmTN g = t.p[pi];
  const int b = bid - (pi ? t.nblk0 : 0);
  const int NTg = (g.N + 63) >> 6;
  const int nt = b % NTg, mt = b / NTg;
  const int m0 = mt * 16, n0 = nt * 64, R = g.R;
  // loader roles: A piece = (row ra = tid >> 2 of the chunk, floats 4 (tid & 3) ..) ; B pieces q = 0..3: row rb = (tid >> 4) + 16 q,
  // floats 4 (tid & 15) ..
  const int ra = tid >> 2, ca = (tid & 3) * 4, rb = tid >> 4, cb = (tid & 15) * 4;
  const bool a_ok = m0 + ca < g.M, b_ok = n0 + cb < g.N;        // (M and N are multiples of 4 where this kernel is used: checked on the host)
  // B rows may be spread over rank-major blocks (packed mode): row r lives in segment r / seg_rows.  One division per piece
  // up front, then (segment, offset) are carried from chunk to chunk (the fetches walk the chunks in order)
  int bseg[4], boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rb + 16 * q;
    bseg[q] = g.b_seg_rows > 0 ? r / g.b_seg_rows : 0;
    boff[q] = g.b_seg_rows > 0 ? r - bseg[q] * g.b_seg_rows : r;
  }
  const float* bcol = g.B + n0 + (b_ok ? cb : 0);
  auto fetch = [&](int k0, float4& va, float4 (&vb)[4]) {   // chunks must be fetched in order: 0, 1, 2, ...
    const int r = k0 + ra;
    const float4 x = *(const float4*)(g.A + (long long)(r < R ? r : 0) * g.lda + (a_ok ? m0 + ca : 0));
    va = (r < R && a_ok) ? x : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = k0 + rb + 16 * q;
      const bool ok = rr < R;
      const long long bo = (long long)bseg[q] * g.b_seg_stride + (long long)boff[q] * g.ldb;
      const float4 y = *(const float4*)(bcol + (ok ? bo : 0));
      vb[q] = (ok && b_ok) ? y : make_float4(0.f, 0.f, 0.f, 0.f);
      boff[q] += DF_KC;
      while (g.b_seg_rows > 0 && boff[q] >= g.b_seg_rows) { boff[q] -= g.b_seg_rows; ++bseg[q]; }
    }
  };
  auto stage = [&](int buf, const float4& va, const float4 (&vb)[4]) {
    *(float4*)(&sh.a[buf][ra * DF_AS + ca]) = make_float4(va.x, va.y, va.z, va.w);
#pragma unroll
    for (int q = 0; q < 4; ++q) *(float4*)(&sh.b[buf][(rb + 16 * q) * DF_BS + cb]) = make_float4(vb[q].x, vb[q].y, vb[q].z, vb[q].w);
  };
  const int NC = (R + DF_KC - 1) / DF_KC;
  // DF_PF chunks in flight in registers (set of chunk c: c % DF_PF; four instead of two changed nothing at 16 chunks)
  constexpr int DF_PF = 2;
  float4 va[DF_PF], vb[DF_PF][4];
#pragma unroll
  for (int c = 0; c < DF_PF; ++c) if (c < NC) fetch(c * DF_KC, va[c], vb[c]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  stage(0, va[0], vb[0]);
  if (DF_PF < NC) fetch(DF_PF * DF_KC, va[0], vb[0]);
  // (sixteen chunks written out per trip: across a loop back-edge hipcc loses count of the outstanding loads and waits
  //  vmcnt(0) in front of the staging, i.e. for the prefetch it has just issued -- a memory round trip per chunk)
  static_assert(16 % DF_PF == 0, "the set of a chunk is a compile-time constant inside a trip");
  for (int cb = 0; cb < NC; cb += 16) {
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) {
      const int c = cb + ci;
      if (c < NC) {                                    // uniform
        __syncthreads();                               // chunk c staged; the other buffer (read during chunk c - 1) is free
        if (c + 1 < NC) {
          stage((ci + 1) & 1, va[(ci + 1) % DF_PF], vb[(ci + 1) % DF_PF]);
          if (c + 1 + DF_PF < NC) fetch((c + 1 + DF_PF) * DF_KC, va[(ci + 1) % DF_PF], vb[(ci + 1) % DF_PF]);
        }
        const float* as = &sh.a[ci & 1][kq * DF_AS + l16];
        const float* bs = &sh.b[ci & 1][kq * DF_BS + wave * 16 + l16];
        // every operand of the chunk first (32 reads in flight), then the MFMAs
        float av[DF_KC / 4], bv[DF_KC / 4];
#pragma unroll
        for (int k4 = 0; k4 < DF_KC / 4; ++k4) { av[k4] = as[(4 * k4) * DF_AS]; bv[k4] = bs[(4 * k4) * DF_BS]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < DF_KC / 4; k4 += 2) {    // two accumulator chains (a dependent MFMA waits for its predecessor)
          acc  = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k4], bv[k4], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k4 + 1], bv[k4 + 1], acc2, 0, 0, 0);
        }
      }
    }
  }
  // D register r of lane l: row 4 (l >> 4) + r, column l & 15
  const int col = n0 + wave * 16 + l16;
  if (col < g.N) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * kq + r;
      if (row < g.M) g.C[(long long)row * g.ldc + col] = (acc[r] + acc2[r]) * g.scale;
    }
  }
}

__global__ __launch_bounds__(256) void loss_dfeat_lds_kernel(GemmTN2 t, int gemm_blocks, LossArgs a) {
  __shared__ DfeatShared sh;
  if ((int)blockIdx.x < gemm_blocks) dfeat_lds_block(t, (int)blockIdx.x, sh);
  else loss_scalars_block(a, &sh.a[0][0]);
}

// dfeat_s / dfeat_w GEMMs (blocks [0, gemm blocks)) + the scalar block (last block)
__global__ __launch_bounds__(256) void loss_dfeat_kernel(GemmTN2 t, int gemm_blocks, LossArgs a) {
  __shared__ GemmTNShared sh;
  if ((int)blockIdx.x < gemm_blocks) gemm_tn_block(t, (int)blockIdx.x, sh);
  else loss_scalars_block(a, &sh.ared[0][0]);
}

template <int MB, int CG, int NBW>
static hipError_t launch_pair_wide(const LossArgs& a, int maxc, hipStream_t st, bool attr_only = false) {
  typedef WideCfg<MB, CG, NBW> Cf;
  constexpr size_t lds = Cf::LDS_FL * 4;
  static DevOnce once;
  hipError_t e = ensure_max_lds(once, pair_exp_wide_kernel<MB, CG, NBW>);
  if (e != hipSuccess || attr_only) return e;
  hipLaunchKernelGGL((pair_exp_wide_kernel<MB, CG, NBW>), dim3((maxc + Cf::NT - 1) / Cf::NT, (a.nunl + Cf::MT - 1) / Cf::MT, 3),
                     dim3(Cf::NTHR), lds, st, a);
  return hipGetLastError();
}

// A captured step picks its pair_exp kernel for the banks' width whatever the smoothing gate says (the gate is read on
// the device), while the eager warm-up step in front of the capture may have run a narrow kernel (gate closed): the
// wide kernels' LDS attribute must not be set for the first time INSIDE the capture, so cmlpl_step_graph_create sets
// all of them before it begins.
hipError_t loss_prepare_capture() {
  const LossArgs a = LossArgs();
  hipError_t e;
  if ((e = launch_pair_wide<4, 1, 1>(a, 0, nullptr, true)) != hipSuccess) return e;
  if ((e = launch_pair_wide<4, 1, 2>(a, 0, nullptr, true)) != hipSuccess) return e;
  if ((e = launch_pair_wide<4, 1, 3>(a, 0, nullptr, true)) != hipSuccess) return e;
  if ((e = launch_pair_wide<4, 1, 4>(a, 0, nullptr, true)) != hipSuccess) return e;
  if ((e = launch_pair_wide<2, 2, 1>(a, 0, nullptr, true)) != hipSuccess) return e;
  return launch_pair_wide<2, 2, 2>(a, 0, nullptr, true);
}

hipError_t launch_loss_phase1(const LossArgs& a_in, hipStream_t st) {
  hipError_t e;
  LossArgs a = a_in;
  a.ctw = 32;
  const int nl = a.nlab + a.nunl;
  // (device-side step scalars: whether the banks are read is decided in the kernels, the grid covers them)
  const bool banks = a.smooth || a.sel.dyn.table != nullptr;
  const int maxc = (banks && a.Q > a.btu) ? a.Q : a.btu;
  const int ctiles = (maxc + 31) / 32;
  // Wide products (data parallelism: >= 4096 columns): pair_exp_wide_kernel up to 128 local rows, pair_exp_tall_kernel
  // beyond.  Narrow ones: 16 x 32 tiles, the contraction split over the waves.  CMLPL_PAIR_WIDE=0 / 1: never / at every
  // size (tests); CMLPL_PAIR_MB, CMLPL_PAIR_NBW force its tile shape; CMLPL_PAIR_TALL=0 / 1: the tall tiles never / always.
  const int wide_mode = switches().pair_wide;   // 0 off, 1 at every size (tests), -1 the planner
  const int force_nbw = switches().pair_nbw;
  const int force_mb = switches().pair_mb;
  const bool pair16 = switches().pair16 != 0;
  // (per rank, configs[2] = 512 + 512 rows over W GPUs, Q = 5120; pair_exp + row kernel: wide 153.7 / 73.0 / 49.9 / 45.0 us at
  //  W = 1 / 2 / 4 / 8 against 115.4 / 69.9 / 55.7 / 55.0 us for the tall tiles: wide up to 128 local rows, tall beyond)
  const int force_tall = switches().pair_tall;
  const bool wide = force_tall <= 0 && wide_mode != 0 && a.K <= 32 && ((ctiles >= 128 && a.nunl <= 128) || wide_mode == 1);
  const bool tall = !wide && (force_tall >= 0 ? force_tall != 0 : (ctiles >= 128 && a.nunl >= 64));
  if (wide) {
    const long long total = (banks ? 2LL * a.Q : 0) + a.btu;                 // columns of the three products
    // 64 x 64 tiles (three workgroups fit a CU) while they make at most one and a half rounds; beyond that 128 x 32 NBW
    // tiles, NBW such that the launch is one round of about one workgroup per CU
    // (measured per rank, B2 128 + 128 rows: W = 4 36.0 us with 64-row tiles / 43.5 with 128-row tiles; W = 8 60.8 / 57.2)
    const long long n64 = ((a.nunl + 63) / 64) * ((total + 63) / 64);
    const int MB = force_mb == 2 || force_mb == 4 ? force_mb : ((a.nunl <= 64 || n64 * 2 <= 3LL * device_cus()) ? 2 : 4);
    const int CG = 4 / MB;
    const int rowblk = (a.nunl + 32 * MB - 1) / (32 * MB);                   // > 1 only beyond 128 local rows
    const long long per = (total * rowblk + device_cus() - 1) / device_cus();   // columns per workgroup for one round
    int NBW = (int)((per + 32 * CG - 1) / (32 * CG));
    const int nbw_max = MB == 4 ? 4 : 2;
    if (MB == 2 && a.nunl > 64) NBW = 1;                                     // the 64 x 64 tiles of the rule above
    if (force_nbw > 0) NBW = force_nbw;
    NBW = NBW < 1 ? 1 : NBW > nbw_max ? nbw_max : NBW;
    a.ctw = 32 * NBW;
    if (MB == 4) e = NBW == 1 ? launch_pair_wide<4, 1, 1>(a, maxc, st) : NBW == 2 ? launch_pair_wide<4, 1, 2>(a, maxc, st)
                   : NBW == 3 ? launch_pair_wide<4, 1, 3>(a, maxc, st) : launch_pair_wide<4, 1, 4>(a, maxc, st);
    else         e = NBW == 1 ? launch_pair_wide<2, 2, 1>(a, maxc, st) : launch_pair_wide<2, 2, 2>(a, maxc, st);
    if (e != hipSuccess) return e;
  } else if (tall) {
    hipLaunchKernelGGL(pair_exp_tall_kernel, dim3(ctiles, (a.nunl + 127) / 128, 3), dim3(256), 0, st, a);
  } else if (a.K <= 32 && pair16) {
    hipLaunchKernelGGL(pair_exp16_kernel, dim3(ctiles, (a.nunl + 15) / 16, 3), dim3(256), 0, st, a);
  } else {
    dim3 g1(ctiles, (a.nunl + 31) / 32, 3);
    hipLaunchKernelGGL(pair_exp_kernel, g1, dim3(256), 0, st, a);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  // + one workgroup per row of the global batch for the bank write (plain mode; packed mode: with the graph launch)
  hipLaunchKernelGGL(loss_rows_kernel, dim3((nl + 3) / 4 + (a.recv_f == nullptr ? a.bt + a.btu : 0)), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_loss_graph(const LossArgs& a, hipStream_t st) {
  const size_t lds = (size_t)2 * a.btu * 4;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  // packed mode: + one workgroup per row of the global batch for the bank write (every rank's probabilities are here now;
  // pair_exp, the last reader of the banks, ran a launch earlier)
  hipLaunchKernelGGL(graph_loss_kernel, dim3(a.nunl + (a.recv_f != nullptr ? a.bt + a.btu : 0)), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_loss_dfeat(const LossArgs& a, hipStream_t st) {
  hipError_t e;
  const int bt = a.bt, btu = a.btu, n = bt + btu;
  // dfeat_s[local rows] = G . fU_w            : C[i][d] = sum_l GT[l][i] * fU_w[l][d]   (R = btu, M = nunl)
  GemmTN g;
  g.A = a.GT; g.lda = a.nunl; g.M = a.nunl; g.R = btu;
  g.b_seg_rows = 0; g.b_seg_stride = 0;
  if (a.recv_f == nullptr) {
    g.B = a.feat + ((long long)n + bt) * FD;
  } else {   // fU_w of all ranks, read from the rank-major blocks: row r -> block r / btu_l
    const long long n_l = a.bt_l + a.btu_l;
    g.B = a.recv_f + (n_l + a.bt_l) * FD;
    g.b_seg_rows = a.btu_l; g.b_seg_stride = a.pack_f;
  }
  g.ldb = FD; g.N = FD;
  g.C = a.dfeat + (long long)a.nlab * FD; g.ldc = FD;
  g.a_bstride = g.b_bstride = g.c_bstride = 0; g.bias = nullptr; g.bias_bstride = 0; g.batches = 1; g.scale = 1.f;
  g.bias_in = nullptr; g.bias_in_bstride = 0; g.relu = 0;
  // dfeat_w[all rows] (this shard's partial) = G^T . fU_s[local] : C[l][d] = sum_i G[i][l] * fU_s[unl0+i][d]
  GemmTN h = g;
  h.b_seg_rows = 0; h.b_seg_stride = 0;
  h.A = a.G; h.lda = btu; h.M = btu; h.R = a.nunl;
  if (a.recv_f == nullptr) {
    h.B = a.feat + ((long long)bt + a.unl0) * FD;
  } else {   // this rank's fU_s rows: one block, contiguous
    const int r = a.unl0 / a.btu_l, i0 = a.unl0 - r * a.btu_l;
    h.B = a.recv_f + r * a.pack_f + (a.bt_l + i0) * FD;
  }
  h.C = a.dfw_part;
  (void)e;
  GemmTN2 t;
  t.p[0] = g; t.p[1] = h;
  // operands through LDS wherever every row is a whole number of 16-byte pieces (measured per rank, B2 128 + 128 rows: 6.7 -> 5.2 us
  // on one GPU, 25.9 -> 22.4 us at W = 8); CMLPL_DFEAT_LDS=0: the direct-load tiles always
  const int lds_mode = switches().dfeat_lds;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const bool lds_ok = (g.M % 4) == 0 && (h.M % 4) == 0 && (g.lda % 4) == 0 && (h.lda % 4) == 0 && (g.b_seg_stride % 4) == 0 &&
                      al16(g.A) && al16(g.B) && al16(h.A) && al16(h.B) && al16(g.C) && al16(h.C);   // 16-byte pieces everywhere
  if (lds_ok && lds_mode != 0) {
    t.nblk0 = dfeat_lds_blocks(g);
    const int gb = t.nblk0 + dfeat_lds_blocks(h);
    hipLaunchKernelGGL(loss_dfeat_lds_kernel, dim3(gb + 1), dim3(256), 0, st, t, gb, a);
    return hipGetLastError();
  }
  t.nblk0 = gemm_tn_blocks(g);
  const int gb = t.nblk0 + gemm_tn_blocks(h);
  hipLaunchKernelGGL(loss_dfeat_kernel, dim3(gb + 1), dim3(256), 0, st, t, gb, a);
  return hipGetLastError();
}

}  // namespace cmlpl
