// Shared device helpers for the CMLPL gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cmlpl.h"

namespace cmlpl {

// Per-step scalars in device memory (cmlpl_dyn, include/cmlpl.h): a step captured in a hipGraph reads what changes
// from step to step -- random-stream counter, Adam step, bank pointers, gates, batch offsets, logging row -- from
// table[*cursor (+ bias)] instead of from its launch arguments.  table == null: the by-value arguments are used.
// table[0] is the WORKING COPY of the current step's row: every launch in front of the weight-gradient reduce reads
// it there (ONE dependent load; through the cursor it was two, and the replayed step ran 2-3 % longer than the eager
// one).  The reduce launch (conv0.hip), which reads none of it, advances the cursor and copies the next row into
// table[0]; Adam, behind it, still needs the finished step's row: table[*cursor - 1] (bias -1).
struct DynRef { const cmlpl_dyn* table; const int* cursor; };
// (by VALUE: a reference to a member of a kernel-argument struct makes hipcc copy the whole struct to scratch --
//  pair_exp16_kernel went from 0 to 352 bytes of scratch per lane and from 15 to 30 us that way)
__device__ __forceinline__ const cmlpl_dyn* dyn_row(DynRef d, int bias = 0) {
  if (d.table == nullptr) return nullptr;
  return bias == 0 ? d.table : d.table + (__builtin_amdgcn_readfirstlane(*d.cursor) + bias);
}
// values read through a row are the same for every lane: say so (they then live in scalar registers whatever kind of
// load fetched them -- without this hipcc 7.2 dies in the backend on "illegal VGPR to SGPR copy" in conv3x3.hip)
__device__ __forceinline__ int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CH = 64;        // conv channels (tools/models.py:102-107)
constexpr int FD = 1024;      // spectral feature width (tools/models.py:119)

// v_mfma_f32_32x32x2_f32: exact fp32 (fmaf chain), 64 cycles/SIMD.
//   A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31]
//   D: reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// torch.relu / threshold_backward semantics (tools/models.py:108,135,139,143): a NaN pre-activation stays NaN
// in the forward and lets the gradient through in the backward (x <= 0 ? 0 : grad).  fmaxf would drop the NaN.
__device__ __forceinline__ float relu_nan(float v) { return (v <= 0.f) ? 0.f : v; }
__device__ __forceinline__ bool relu_open(float r) { return !(r <= 0.f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 32 lanes of each half-wave (lanes sharing l>>5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Staging copy with instruction-level parallelism: each thread first issues UNR independent loads, then
// performs the UNR stores.  At this problem size the chip is far from full, so a staging loop that does
// load -> wait -> store one element at a time is bound by memory latency, not bandwidth.
//   load(idx)  must be safe for every idx in [0, tot) ; store(idx, v) is only called for idx < tot.
template <int UNR, typename T, int NTHREADS = 256, class L, class S>
__device__ __forceinline__ void staged_copy(int tot, int tid, L load, S store) {
  for (int base = 0; base < tot; base += NTHREADS * UNR) {
    T v[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int idx = base + q * NTHREADS + tid;
      v[q] = load(idx < tot ? idx : tot - 1);
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int idx = base + q * NTHREADS + tid;
      if (idx < tot) store(idx, v[q]);
    }
  }
}

// ---------------------------------------------------------------- Philox4x32-10
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __device__ static __forceinline__ uint4 gen(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      // one 64-bit product (v_mad_u64_u32) gives hi and lo: 32-bit multiplies are quarter-rate on CDNA,
      // and this generator is what bounds the augmentation kernel
      const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
      const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
      const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
      k.x += W0; k.y += W1;
    }
    return c;
  }
};
// 4 standard normals from one Philox block (Box-Muller)
__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint64_t step, uint32_t stream, uint64_t idx) {
  uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, (uint32_t)step);
  uint2 k = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
  uint4 r = Philox::gen(c, k);
  const float s = 2.3283064365386963e-10f;  // 2^-32
  float u0 = ((float)r.x + 1.0f) * s, u1 = (float)r.y * s;
  float u2 = ((float)r.z + 1.0f) * s, u3 = (float)r.w * s;
  u0 = fminf(u0, 1.0f); u2 = fminf(u2, 1.0f);
  float r0 = sqrtf(-2.0f * __logf(u0)), r1 = sqrtf(-2.0f * __logf(u2));
  float s0, c0, s1, c1;
  __sincosf(6.283185307179586f * u1, &s0, &c0);
  __sincosf(6.283185307179586f * u3, &s1, &c1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}
// Bulk augmentation noise (6.4 M normals per step at B2): PCG4D (Jarzynski & Olano, "Hash Functions for GPU
// Rendering", JCGT 2020), a counter-based 4 x 32 -> 4 x 32 bit hash -- 12 multiply-adds against the 20 64-bit
// multiplies of Philox4x32-10, which on CDNA are quarter-rate instructions and made the generator, not the memory
// system, the bound of the augmentation (measured: 8-12 us per fused kernel with Philox).  Inputs: the counter
// (group, sample), the stream and the step, each whitened with the seed.  Dropout keeps Philox.
__device__ __forceinline__ uint4 pcg4d(uint4 v) {
  v.x = v.x * 1664525u + 1013904223u; v.y = v.y * 1664525u + 1013904223u;
  v.z = v.z * 1664525u + 1013904223u; v.w = v.w * 1664525u + 1013904223u;
  v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
  v.x ^= v.x >> 16; v.y ^= v.y >> 16; v.z ^= v.z >> 16; v.w ^= v.w >> 16;
  v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
  return v;
}
// 4 standard normals for elements 4g..4g+3 of a sample (Box-Muller on the four hash words)
__device__ __forceinline__ float4 noise_normal4(uint64_t seed, uint64_t step, uint32_t stream, uint64_t ctr) {
  const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  const uint4 r = pcg4d(make_uint4((uint32_t)ctr ^ s0, (uint32_t)(ctr >> 32) ^ s1,
                                   (stream * 0x9E3779B9u) ^ (uint32_t)(step >> 32) ^ (s1 * 0x85EBCA6Bu),
                                   (uint32_t)step ^ (s0 * 0xC2B2AE35u)));
  // Box-Muller on the hardware transcendentals directly: u in (0, 1] -> radius sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
  // (v_log_f32 is log2, u >= 2^-32 is a normal number), angle in REVOLUTIONS (what v_sin_f32 / v_cos_f32 take).
  // The library forms (denormal scaling for log, correctly rounded sqrt, radian range reduction) cost 3x the
  // instructions and buy nothing for augmentation noise.
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = fminf(fmaf((float)r.x, k, k), 1.0f), u2 = fminf(fmaf((float)r.z, k, k), 1.0f);
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  const float a0 = (float)r.y * k, a1 = (float)r.w * k;
  return make_float4(r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0),
                     r1 * __builtin_amdgcn_cosf(a1), r1 * __builtin_amdgcn_sinf(a1));
}
__device__ __forceinline__ float4 philox_uniform4(uint64_t seed, uint64_t step, uint32_t stream, uint64_t idx) {
  uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, (uint32_t)step);
  uint2 k = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
  uint4 r = Philox::gen(c, k);
  const float s = 2.3283064365386963e-10f;
  return make_float4((float)r.x * s, (float)r.y * s, (float)r.z * s, (float)r.w * s);
}

constexpr uint32_t STREAM_NOISE_XP = 0x100;   // + net
constexpr uint32_t STREAM_NOISE_X = 0x200;    // + net
constexpr uint32_t STREAM_DROPOUT = 0x300;    // + net

// Counter of the Philox block that carries the noise of elements 4g .. 4g+3 of one sample's patch (or spectrum):
// (GLOBAL sample index, group) -- the same in the augmentation kernel, the fused forward and the fused data
// gradient (which regenerates the forward's noise instead of reading an augmented copy back from HBM), and
// independent of how the batch is sharded over GPUs.
__device__ __forceinline__ uint64_t noise_ctr(uint64_t global_sample, uint32_t group) {
  return (global_sample << 24) | (uint64_t)group;
}

// The PATCH noise (12,463 normals per sample-net at B2: the largest vector-bound item of the fused forward) takes EIGHT
// normals from one hash call: each 32-bit hash word gives a 16-bit radius uniform u = (hi16 + 1) / 2^16 in (0, 1] and a
// 16-bit angle lo16 / 2^16 revolutions, i.e. two normals per word -- half the hash multiplies per normal and no more
// transcendentals than before.  |z| <= sqrt(2 ln 2^16) = 4.71 (the mass beyond is 2.5e-6: ~16 of a step's 6.4 M draws
// land on the last radius instead of further out); 65,536 radii x 65,536 angles per pair.  The unit is the PAIR of
// 16-byte groups 2c, 2c + 1 of a sample (elements 8c .. 8c + 7), keyed like noise_normal4: counter (global sample, c).
__device__ __forceinline__ void bm16(uint32_t w, float& c, float& s) {
  const float k16 = 1.52587890625e-05f;     // 2^-16
  const float u = fmaf((float)(w >> 16), k16, k16);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
  const float a = (float)(w & 0xffffu) * k16;
  c = r * __builtin_amdgcn_cosf(a); s = r * __builtin_amdgcn_sinf(a);
}
__device__ __forceinline__ uint4 noise_hash(uint64_t seed, uint64_t step, uint32_t stream, uint64_t ctr) {
  const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  return pcg4d(make_uint4((uint32_t)ctr ^ s0, (uint32_t)(ctr >> 32) ^ s1,
                          (stream * 0x9E3779B9u) ^ (uint32_t)(step >> 32) ^ (s1 * 0x85EBCA6Bu),
                          (uint32_t)step ^ (s0 * 0xC2B2AE35u)));
}
// both groups of pair c = gidx >> 1 (gidx even): lo = elements 8c .. 8c + 3, hi = 8c + 4 .. 8c + 7
__device__ __forceinline__ void noise_normal8(uint64_t seed, uint64_t step, uint32_t stream, uint64_t gsample, uint32_t pair,
                                              float4& lo, float4& hi) {
  const uint4 r = noise_hash(seed, step, stream, noise_ctr(gsample, pair));
  bm16(r.x, lo.x, lo.y); bm16(r.y, lo.z, lo.w); bm16(r.z, hi.x, hi.y); bm16(r.w, hi.z, hi.w);
}
// ONE group (gidx) of its pair: the whole hash, half of the Box-Muller work (callers whose two groups are not neighbours)
__device__ __forceinline__ float4 noise_normal4p(uint64_t seed, uint64_t step, uint32_t stream, uint64_t gsample, uint32_t gidx) {
  const uint4 r = noise_hash(seed, step, stream, noise_ctr(gsample, gidx >> 1));
  const bool odd = (gidx & 1u) != 0;
  float4 o;
  bm16(odd ? r.z : r.x, o.x, o.y); bm16(odd ? r.w : r.y, o.z, o.w);
  return o;
}

// Where the patch rows of one network come from (train.py:157-174,181-184): local rows [0, nlab) are rows of
// `lab`, rows [nlab, n) rows of `unl` (the concat is an index computation, not a copy); the augmentation
// x + sigma * N(0,1) is applied on the fly -- from explicit noise tensors (parity mode: the reference's draws)
// or from the Philox stream.  sigma == 0: plain rows (inference, or inputs that are already augmented).
// fp32 value as three bf16 pieces, v = p0 + p1 + p2 EXACTLY: each piece is the truncation (top 16 bits) of what the
// previous ones left, and truncation keeps 8 significant bits, so three pieces hold the 24-bit significand; the
// residuals v - p0 and (v - p0) - p1 are exact in fp32.  (NaN stays NaN in every piece; +-Inf does not occur in
// this path and would turn into NaN.)
__device__ __forceinline__ void b3_split(float v, uint32_t (&hi16)[3]) {
  const uint32_t u0 = __float_as_uint(v);
  const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
  const uint32_t u1 = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
  hi16[0] = u0 >> 16; hi16[1] = u1 >> 16; hi16[2] = __float_as_uint(r2) >> 16;
}

// ---- fp32 as TWO fp16 pieces (conv3x3.hip): x = h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) -- 21-22 significant bits (tests/test_split_f16_math.py) as
// long as the residual stays a normal fp16, which is what the operands' power-of-two scales are for (activations: per
// sample, from the staged image's maximum; weights: 2^H2_WEXP at packing time).  a.w ~ h1 g1 + h2 g1 + h1 g2: three MFMAs.
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_h16(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
}
// a weight at the packing scale -> its two pieces (low 16 bits of pcs[0], pcs[1]); true when it does not fit fp16's range
__device__ __forceinline__ bool h2_split_w(float w, uint32_t (&pcs)[2]) {
  const float x = w * (float)(1 << 13);                       // (H2_WEXP, kernels.hpp)
  const _Float16 g1 = (_Float16)x;
  const _Float16 g2 = (_Float16)(x - (float)g1);
  pcs[0] = (uint32_t)__builtin_bit_cast(uint16_t, g1); pcs[1] = (uint32_t)__builtin_bit_cast(uint16_t, g2);
  return !(fabsf(x) <= 65000.f);                               // (also true for NaN)
}
// eight consecutive fp32 A elements, scaled by the sample's power of two -> the two fp16 A fragments (round toward zero:
// the residual of a truncated first piece is exact and has the sign of x)
__device__ __forceinline__ void h_split(const float4& x0, const float4& x1, float sc, uint4& H1, uint4& H2) {
  const float w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  uint32_t h1[4], h2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = w[2 * j] * sc, b = w[2 * j + 1] * sc;
    const f16x2v p = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(a, b));
    h1[j] = __builtin_bit_cast(uint32_t, p);
    h2[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a - (float)p[0], b - (float)p[1]));
  }
  H1 = make_uint4(h1[0], h1[1], h1[2], h1[3]);
  H2 = make_uint4(h2[0], h2[1], h2[2], h2[3]);
}
// one 32 x 32 x 16 step of the two-piece product; g1, g2 = the weight pieces of this n tile
__device__ __forceinline__ f32x16 mfma_h2(const uint4& H1, const uint4& H2, const uint4& g1, const uint4& g2, f32x16 acc) {
  acc = mfma_h16(H2, g1, acc);
  acc = mfma_h16(H1, g2, acc);
  acc = mfma_h16(H1, g1, acc);
  return acc;
}

// ---- split-bf16 MFMA helpers (conv3x3.hip "fp32 on the bf16 MFMA"; shared with wgrad3x3.hip)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_b16(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// high halves of two dwords -> one dword (lo element in the low half)
__device__ __forceinline__ uint32_t hi_pair(uint32_t lo_el, uint32_t hi_el) {
  return __builtin_amdgcn_perm(hi_el, lo_el, 0x07060302u);
}
// eight consecutive fp32 A elements (k = 8h .. 8h+7 of this lane's row) -> the three bf16 A fragments
__device__ __forceinline__ void a_split(const float4& x0, const float4& x1, uint4& A1, uint4& A2, uint4& A3) {
#if defined(CMLPL_ABL) && CMLPL_ABL == 21         // ablation: no split arithmetic -- wrong results
  A1 = make_uint4(__float_as_uint(x0.x), __float_as_uint(x0.y), __float_as_uint(x0.z), __float_as_uint(x0.w));
  A2 = make_uint4(__float_as_uint(x1.x), __float_as_uint(x1.y), __float_as_uint(x1.z), __float_as_uint(x1.w));
  A3 = A1;
  return;
#endif
#if defined(CMLPL_ABL) && CMLPL_ABL == 60         // ablation: the instruction mix of a TWO-piece fp16 product (x = h1 + 2^-11 h2)
  {                                                 // against the weight fragments as they are -- wrong results, right timing
    const float w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    uint32_t h1[4], h2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      const half2v p = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(w[2 * j], w[2 * j + 1]));
      const float r0 = (w[2 * j] - (float)p[0]) * 2048.f, r1 = (w[2 * j + 1] - (float)p[1]) * 2048.f;
      h1[j] = __builtin_bit_cast(uint32_t, p);
      h2[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
    A1 = make_uint4(h1[0], h1[1], h1[2], h1[3]);
    A2 = make_uint4(h2[0], h2[1], h2[2], h2[3]);
    A3 = A1;
    return;
  }
#endif
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  uint32_t u0[8], u1[8], u2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u0[j] = __float_as_uint(v[j]);
    const float r1 = v[j] - __uint_as_float(u0[j] & 0xffff0000u);
    u1[j] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1[j] & 0xffff0000u);
    u2[j] = __float_as_uint(r2);
  }
  A1 = make_uint4(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]), hi_pair(u0[4], u0[5]), hi_pair(u0[6], u0[7]));
  A2 = make_uint4(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]), hi_pair(u1[4], u1[5]), hi_pair(u1[6], u1[7]));
  A3 = make_uint4(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]), hi_pair(u2[4], u2[5]), hi_pair(u2[6], u2[7]));
}
// one 32 x 32 x 16 step of the split product; b1..b3 = the weight pieces of this n tile
__device__ __forceinline__ f32x16 mfma_b3(const uint4& A1, const uint4& A2, const uint4& A3, const uint4& b1,
                                          const uint4& b2, const uint4& b3, f32x16 acc) {
#if defined(CMLPL_ABL) && CMLPL_ABL == 60
  {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A2), __builtin_bit_cast(f16x8, b1), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A1), __builtin_bit_cast(f16x8, b2), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A1), __builtin_bit_cast(f16x8, b1), acc, 0, 0, 0);
    return acc;
  }
#endif
  acc = mfma_b16(A1, b3, acc);
  acc = mfma_b16(A2, b2, acc);
  acc = mfma_b16(A3, b1, acc);
  acc = mfma_b16(A1, b2, acc);
  acc = mfma_b16(A2, b1, acc);
  acc = mfma_b16(A1, b1, acc);
  return acc;
}


// scheduling pattern of the software-pipelined MFMA loops: N x (six vector instructions, one MFMA), pinned with
// sched_group_barrier (left alone, the compiler puts a step's split instructions in front of its MFMAs)
template <int N> struct SchedInterleave {
  static __device__ __forceinline__ void run() {
    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    SchedInterleave<N - 1>::run();
  }
};
template <> struct SchedInterleave<0> { static __device__ __forceinline__ void run() {} };

// Batches by index (cmlpl_batch.d_lab_idx / d_unl_idx): labelled (unlabelled) batch row r is row idx[off + r] of the
// resident split; idx == null: row r.  The offsets come from the device-side row when one is given (graph replay).
struct RowSel {
  const long long* lab_idx; const long long* unl_idx;
  long long lab_off, unl_off;
  DynRef dyn;
};
__device__ __forceinline__ long long rowsel_index(RowSel s, bool lab, int r) {
  const long long* idx = lab ? s.lab_idx : s.unl_idx;
  if (idx == nullptr) return r;
  const cmlpl_dyn* d = dyn_row(s.dyn);
  long long loff = s.lab_off, uoff = s.unl_off;
  if (d != nullptr) { loff = uni64(d->lab_off); uoff = uni64(d->unl_off); }   // (`lab` may differ between lanes)
  return idx[(lab ? loff : uoff) + r];
}

struct XSrc {
  const float* lab[2]; const float* unl[2];       // per network (the same pointer twice for raw inputs)
  const float* nz_lab[2]; const float* nz_unl[2]; // explicit N(0,1) draws per network, or null
  float sigma; int nlab, philox /* 1 = in-kernel draws (PCG4D noise), 0 = explicit noise tensors */, lab0, unl_base;
  uint64_t seed, step;
  RowSel sel;                                      // batches by index + device-side step scalars (all null: plain rows)
};
// row of the source buffers that batch row s stands for
__device__ __forceinline__ long long xsrc_index(const XSrc& x, int s) {
  const bool lab = s < x.nlab;
  return rowsel_index(x.sel, lab, lab ? s : s - x.nlab);
}
__device__ __forceinline__ uint64_t xsrc_step(const XSrc& x) {
  const cmlpl_dyn* d = dyn_row(x.sel.dyn);
  uint64_t v = x.step;
  if (d != nullptr) v = (uint64_t)uni64((long long)d->step);
  return v;
}
__device__ __forceinline__ const float* xsrc_row(const XSrc& x, int net, int s, long long per) {
  return (s < x.nlab ? x.lab[net] : x.unl[net]) + xsrc_index(x, s) * per;
}
__device__ __forceinline__ const float* xsrc_noise_row(const XSrc& x, int net, int s, long long per) {
  const float* b = s < x.nlab ? x.nz_lab[net] : x.nz_unl[net];
  return b == nullptr ? nullptr : b + (long long)(s < x.nlab ? s : s - x.nlab) * per;
}
__device__ __forceinline__ uint64_t xsrc_global_sample(const XSrc& x, int s) {
  return (uint64_t)(s < x.nlab ? x.lab0 + s : x.unl_base + (s - x.nlab));
}

}  // namespace cmlpl
