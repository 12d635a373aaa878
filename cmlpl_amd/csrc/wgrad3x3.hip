// 3x3 weight gradients of BaseNet2's conv1 / conv2 (tools/models.py:104-107 backward):
//   dW[s][ci][co] = sum_pix in[pix+s][ci] * dz[pix][co], db[co] = sum dz, with dz = mask * upsample(dpool) / 4 formed
//   while staging.  wgrad3b_kernel (split-bf16 MFMA, planes + transposed LDS reads) is the default; wgrad3r_kernel
//   (f32-input MFMA, row-split, LDS-DMA staging) and wgrad3_kernel (general) are the round-2a / round-1 kernels kept
//   as fallbacks and reference points.  Split from conv3x3.hip so that the two heavy translation units compile in
//   parallel.  See conv3x3.hip for the data layout and for "fp32 on the bf16 MFMA".
#include <stdlib.h>

#ifndef CMLPL_ABL
#define CMLPL_ABL 0
#endif

#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

#if CMLPL_ABL == 9 || (CMLPL_ABL >= 20 && CMLPL_ABL != 26)
// phase timeline instrumentation (ablation build only): constant-rate 100 MHz stamps per workgroup; mode 2 of
// scripts/conv_timeline.py (device globals are per translation unit: conv3x3.hip has its own for modes 0 / 1)
__device__ unsigned long long g_wstamps[2048][16];
#define STAMP(MODE_, i) do { if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 2048) \
    g_wstamps[blockIdx.x + gridDim.x * blockIdx.y][i] = wall_clock64(); } while (0)
extern "C" int cmlpl_abl_read_wstamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wstamps), sizeof(g_wstamps));
}
#else
#define STAMP(MODE_, i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
constexpr int WG3B_MAXCPR = 11;   // wider maps: two three-piece stage buffers no longer fit in LDS
struct Wgrad3Args {
  const float* in; const float* dpool; const uint8_t* mask; float* part;
  long long in_ns, dpool_ns, part_ns;
  int n, H, W, RU, U, G, UPG;
  // two-piece fp16 planes (wgrad3b_body<CPR, true>): per network and sample the largest magnitude (float bits) of the
  // activation operand and of the masked up-sampled pooled gradient, left by the fused forward / backward kernels of this
  // step (conv3x3.hip: Conv3Args::hstat, [2 networks][n]); the networks' weight-range flags (two-piece packing, kernels.hpp);
  // null = three-piece bf16 planes
  const uint32_t* stat_a; const uint32_t* stat_g; const uint32_t* h2flag; long long h2flag_ns;
  // ... and for skipping the samples whose gradient operand is zero everywhere (stat_g word 0: a row no loss term reaches):
  // byte offset of the LDS list of the other samples (behind the stage buffers; 0 = no room, nothing is skipped), x / UPS
  // as a multiply
  int slist_off; uint32_t mg_ups;
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
  const int D = U * DU; (void)D;           // dz slots per pass
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

// Row-split variant (default): blockIdx.z = kernel row kh, so a workgroup owns the three taps (kh, 0..2) for ALL
// 64x64 channel pairs and walks three times as many samples as a workgroup that owns all nine taps would.  What
// that buys on B2/256: the per-workgroup partials (and with them the kernel-end write drain and the reduce
// kernel's input) shrink from 37.7 MB to 11.8 MB per launch; no halo rows are staged (tap row kh of output row r
// is input row r + kh - 1, zero outside the image); one A row pointer per output row instead of three.
//   512 threads: wave w = (pixel group kg = w >> 2, ci tile (w >> 1) & 1, co tile w & 1), 3 accumulators each
//   (kw = 0..2).  A stage holds U units (unit = one pooled row = two output rows); pixel group kg takes the units
//   of its parity, so the two waves of a SIMD run independent MFMA streams; the two groups are folded through LDS
//   once, after the last stage.  Stages are double-buffered: while the MFMA loop runs on stage g, the rows of
//   stage g+1 are in flight global -> registers and are written to the other LDS half after the loop.
//   CPR (column pairs per output row) is a template parameter and a whole unit (2 x CPR pixel pairs) is unrolled,
//   so every LDS operand address is a row pointer plus an immediate.  This matters more than anything else here
//   (scripts/mfma_mix*.hip, MI355X): a lone wave issues v_mfma_f32_32x32x2 every 70 cycles, not 64; each VALU
//   instruction between two MFMAs costs ~6-15 cycles; and scalar instructions are a CU-wide ~1/cycle resource
//   that more waves do NOT hide -- the generic (cp, r, unit) bookkeeping, ~22 s_cmp/s_cselect per pixel pair,
//   held the loop at 1.5x its MFMA time.
//   Staging: the activation rows go global -> LDS directly (global_load_lds_dwordx4: one wave-instruction moves
//   4 pixels x 64 channels = 1 KiB, no VGPR round trip, no ds_write), a stage is rows x ceil(W/4) such pieces
//   spread over the eight waves, all bookkeeping scalar.  Only the pooled gradient (it needs the ReLU-mask
//   multiply and the 2x2 upsample) is staged through registers, one item per thread.  With every wave staging
//   its share through registers instead (4 float4 + address arithmetic + ds_write per thread and stage) the
//   same kernel measured 43.5 us on B2/256: ~200 non-MFMA instructions per thread and stage are not hidden by
//   the other wave of the SIMD.  (Also tried and slower: four dedicated loader waves, 49 us.)
constexpr int WG3R_NTK = 4;    // LDS-DMA pieces per wave and stage (planner keeps rows x ceil(W/4) <= 32)
typedef __attribute__((address_space(3))) void wg3r_lds_void;
typedef __attribute__((address_space(1))) const void wg3r_gbl_void;
template <int CPR>
__global__ __launch_bounds__(512) void wgrad3r_kernel(Wgrad3Args a) {
  constexpr int NT = 512, NTK = WG3R_NTK;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.y, g = blockIdx.x, kh = blockIdx.z;
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2;
  const int H2 = H >> 1, W2 = W >> 1, P2 = H2 * W2;
  constexpr int CO = 2 * CPR;
  const int U = a.U;
  const int IMGU = 2 * PW;
  constexpr int DU = 2 * CO;
  const int UPS = H2;                       // units (pooled rows) per sample
  const int NU = a.n * UPS;
  const int UPG = a.UPG;
  const int ubeg = g * UPG, uend = (ubeg + UPG < NU) ? ubeg + UPG : NU;
  const int IMGF = U * IMGU * 64, BUF = IMGF + U * DU * 64;     // floats per stage buffer: [img | dz]
  STAMP(2, 0);

  {  // border columns of both buffers must be zero; interiors are rewritten every stage
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* p4 = (float4*)smem;
    for (int i = tid; i < (2 * BUF) >> 2; i += NT) p4[i] = z;
  }
  const float* src = a.in + (long long)net * a.in_ns;
  const float* dp = a.dpool + (long long)net * a.dpool_ns;
  const uint8_t* mk = a.mask + (long long)net * a.dpool_ns;
  const int ct = wave & 1, it = (wave >> 1) & 1, kg = wave >> 2;

  // LDS-DMA pieces of this wave: piece t = (stage row t / CH, 4-pixel chunk t % CH), t = wave, wave + 8, ...
  // The decode is stage-invariant and wave-uniform; the (sample, unit-in-sample) pair of every piece and of the
  // thread's pooled-gradient item is carried from stage to stage (no divisions per stage; 32-bit offsets, the
  // planner checks that one network's activations stay below 2^31 elements).
  const int CH = (W + 3) >> 2, ntask = 2 * U * CH;
  const int qU = U / UPS, rU = U - qU * UPS;
  int t_u[NTK], t_irk[NTK], t_l[NTK], t_smp[NTK], t_j[NTK], t_lane[NTK];
#pragma unroll
  for (int q = 0; q < NTK; ++q) {
    const int t = wave + 8 * q;
    const int tt = t < ntask ? t : 0;
    const int rowi = tt / CH, ch = tt - rowi * CH, u = rowi >> 1, ir = rowi & 1;
    t_u[q] = t < ntask ? u : (1 << 28); t_irk[q] = ir + kh - 1;
    t_l[q] = (u * IMGU + ir * PW + 1 + 4 * ch) * 64;                      // LDS float offset of the piece
    t_smp[q] = (ubeg + u) / UPS; t_j[q] = (ubeg + u) - t_smp[q] * UPS;
    const int px = 4 * ch + (lane >> 4);                                  // this lane's pixel of the row
    t_lane[q] = px < W ? px * 64 + (lane & 15) * 4 : -1;                   // per-lane source offset (floats)
  }
  const int dtot = U * W2 * 16;          // <= 512 (planner)
  int d_u, d_g, d_l, d_smp, d_j;
  {
    const int id = tid < dtot ? tid : 0;
    const int c4 = id & 15, p = id >> 4;
    const int u = p / W2, pw = p - u * W2;
    d_u = tid < dtot ? u : (1 << 28); d_g = pw * 64 + c4 * 4;
    d_l = (u * DU + 2 * pw) * 64 + c4 * 4;
    d_smp = (ubeg + u) / UPS; d_j = (ubeg + u) - d_smp * UPS;
  }
  float4 pdd;
  uint32_t pdm;
  bool pdok = false;
  float4 dbsum = make_float4(0.f, 0.f, 0.f, 0.f);   // bias gradient of channels 4*(tid&15)..+3 (kernel row 0 only)
  // issue(ub, buf): start the transfer of the stage that begins at unit ub into buf.  Must be called for
  // ub = ubeg, ubeg + U, ... in order (it advances the carried indices).  Rows outside the image or beyond the
  // workgroup's range are zero-filled instead (the buffer holds an older stage).
  auto issue = [&](int ub, float* buf) {
#pragma unroll
    for (int q = 0; q < NTK; ++q) {
      const int row = t_j[q] * 2 + t_irk[q];
      const bool ex = t_u[q] < (1 << 28);                                     // wave-uniform
      const bool ok = ex && (ub + t_u[q] < uend) && row >= 0 && row < H;      // wave-uniform
      if (ok && (CMLPL_ABL != 12 || ub == ubeg)) {
        if (t_lane[q] >= 0)
#if CMLPL_ABL == 10
          __builtin_amdgcn_global_load_lds((wg3r_gbl_void*)(src + (((0 * HW + row * W) << 6) + t_lane[q])),
#else
          __builtin_amdgcn_global_load_lds((wg3r_gbl_void*)(src + (((t_smp[q] * HW + row * W) << 6) + t_lane[q])),
#endif
                                           (wg3r_lds_void*)(buf + t_l[q]), 16, 0, 0);
      } else if (ex) {
        if (t_lane[q] >= 0) *(float4*)(buf + t_l[q] + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      t_smp[q] += qU; t_j[q] += rU;
      if (t_j[q] >= UPS) { t_j[q] -= UPS; ++t_smp[q]; }
    }
    {
      pdok = ub + d_u < uend;
      const int gi = pdok ? ((d_smp * P2 + d_j * W2) << 6) + d_g : 0;
      pdd = *(const float4*)(dp + gi);         // raw: touching the value here would put the wait before the MFMAs
      pdm = *(const uint32_t*)(mk + gi);
      d_smp += qU; d_j += rU;
      if (d_j >= UPS) { d_j -= UPS; ++d_smp; }
    }
  };
  auto commit = [&](float* buf) {
    if (d_u < (1 << 28)) {
      if (!pdok) pdm = 0u;
      float* dzb = buf + IMGF + d_l;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        float4 v;
        v.x = ((pdm >> sub) & 1u) ? pdd.x * 0.25f : 0.f;
        v.y = ((pdm >> (8 + sub)) & 1u) ? pdd.y * 0.25f : 0.f;
        v.z = ((pdm >> (16 + sub)) & 1u) ? pdd.z * 0.25f : 0.f;
        v.w = ((pdm >> (24 + sub)) & 1u) ? pdd.w * 0.25f : 0.f;
        *(float4*)(dzb + ((sub >> 1) * CO + (sub & 1)) * 64) = v;
        dbsum.x += v.x; dbsum.y += v.y; dbsum.z += v.z; dbsum.w += v.w;
      }
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) acc[s] = zero16();

  __syncthreads();                 // zero fill complete
  if (ubeg < uend) { issue(ubeg, smem); commit(smem); }
  __syncthreads();
  STAMP(2, 1);
  // This wave's units of a stage: kg, kg + 2, ...  Pixel pair q of a unit (row r = q / CPR, columns 2cp, 2cp+1
  // with cp = q % CPR; lane half hh takes column 2cp + hh): tap kw is padded position r * PW + 2cp + hh + kw.
  const int rstride = PW * 64, ustep_a = 2 * IMGU * 64;
  constexpr int ustep_b = 2 * DU * 64;
  const int nun = U >> 1;                     // units per wave and stage
  int cur = 0;
  for (int ub = ubeg; ub < uend; ub += U) {
    const bool more = ub + U < uend;            // workgroup-uniform
    const float* buf = smem + cur * BUF;
    const float* a_base = buf + kg * (IMGU * 64) + it * 32 + l31 + hh * 64;
    const float* b_base = buf + IMGF + kg * (DU * 64) + ct * 32 + l31 + hh * 64;
    float av[2][3], bv[2];                      // ping-pong operand sets (indices fold after unrolling)
    av[0][0] = a_base[0]; av[0][1] = a_base[64]; av[0][2] = a_base[128];
    bv[0] = b_base[0];
    const float* pa0 = a_base;
    const float* pb = b_base;
    for (int uu = 0; uu < nun; ++uu) {
#if CMLPL_ABL != 7 && CMLPL_ABL != 8
      if (more && uu == 0) issue(ub + U, smem + (cur ^ 1) * BUF);   // first thing: in flight across the MFMAs below
#endif
      const float* pa1 = pa0 + rstride;
      const int nxt = (uu + 1 < nun) ? uu + 1 : 0;        // after the last unit: re-read unit 0 (unused)
      const float* na0 = a_base + nxt * ustep_a;
      const float* nb = b_base + nxt * ustep_b;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2 * CPR; ++q) {
        const int c = q & 1, n = c ^ 1;
        const int nq = q + 1;
        const float* pA = (nq < 2 * CPR) ? (nq >= CPR ? pa1 : pa0) : na0;
        const float* pB = (nq < 2 * CPR) ? pb : nb;
        const int oA = (nq < 2 * CPR) ? (nq % CPR) * 128 : 0;
        const int oB = (nq < 2 * CPR) ? nq * 128 : 0;
        // tap 0 of the next pair in the same row is tap 2 of this one (the pairs are two columns apart)
        const bool same_row = (nq < 2 * CPR) && (nq % CPR != 0);
        bv[n] = pB[oB];
        av[n][0] = same_row ? av[c][2] : pA[oA];
        acc[0] = mfma32(av[c][0], bv[c], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        av[n][1] = pA[oA + 64];
        acc[1] = mfma32(av[c][1], bv[c], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        av[n][2] = pA[oA + 128];
        acc[2] = mfma32(av[c][2], bv[c], acc[2]);
        __builtin_amdgcn_sched_barrier(0);
      }
      pa0 = na0; pb = nb;
    }
#if CMLPL_ABL != 7 && CMLPL_ABL != 8 && CMLPL_ABL != 11
    if (more) commit(smem + (cur ^ 1) * BUF);
#endif
#if CMLPL_ABL != 8
    __syncthreads();   // stage g fully read by every wave, stage g+1 fully written
#endif
    cur ^= 1;
  }
  STAMP(2, 2);

  // fold pixel group 1 into group 0 through LDS ([4 waves][48][64]); group 0 stores the partial
  float* red = smem + (size_t)(wave & 3) * 48 * 64 + lane;
  if (kg == 1) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(s * 16 + r) * 64] = acc[s][r];
  }
  float4* dbl = (float4*)(smem + 4 * 48 * 64);    // [512] per-thread bias partial sums
  if (kh == 0) dbl[tid] = dbsum;
  __syncthreads();
  float* part = a.part + (long long)net * a.part_ns + (size_t)g * PART3;
  if (kg == 0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = it * 32 + acc_row(r, lane);
        part[(3 * kh + s) * 4096 + ci * 64 + ct * 32 + l31] = acc[s][r] + red[(s * 16 + r) * 64];
      }
    }
  }
  // bias gradient (kernel row 0 only): thread t holds channels 4*(t & 15)..+3; fixed-order sum over the 32 threads
  // of each channel quad
  if (kh == 0 && tid < 64) {
    const int c4 = tid >> 2, e = tid & 3;
    float sum = 0.f;
    for (int k = 0; k < 32; ++k) sum += ((const float*)&dbl[c4 + 16 * k])[e];
    part[9 * 4096 + tid] = sum;
  }
  STAMP(2, 3);
}

// ---- split-bf16 weight gradient (default where it fits): wgrad3r_kernel's decomposition (blockIdx.z = kernel row,
// 512 threads, wave = (pixel group kg, ci tile it, co tile ct), three accumulators kw = 0..2, stages of U pooled
// rows, double-buffered) with both operands as three bf16 pieces on v_mfma_f32_32x32x16_bf16 (see "fp32 on the bf16
// MFMA" above).  The contraction index of this GEMM is the PIXEL, and the activations arrive pixel-major /
// channel-last -- k-major for both operands -- so the pieces are formed ONCE, while staging (global -> registers ->
// split -> LDS), into bf16 planes [position][64 channels], and the MFMA loop fetches its fragments with the
// transposed LDS read (ds_read_b64_tr_b16: 4 positions x 16 channels per 16 lanes), with no VALU work at all.
//   Position order: a stage holds R = 2U image rows; the planes are COLUMN-major, pos = x * R + row, so that
//   (1) the gradient plane dz (columns 0..CO-1 only) is one gapless k range [0, CO * R): no halo slots are
//       multiplied, and
//   (2) tap kw of the same k is the activation position k + kw * R: a constant offset, like every other address in
//       the loop (k-step, piece, half-fragment): one base register per operand, the rest immediates.
//   The 64-B channel halves of a position are swapped where (pos >> 1) & 1, which makes the four rows of a
//   transposed read fall on four different 16-bank groups (conflict-free; probe: scripts/probes/tr_probe.hip).
//   K = CO * R is padded to whole k-steps of 16 with zero gradient rows; pixel group kg takes steps kg, kg+2, ...
constexpr int wg3b_U(int CPR) {      // pooled rows per stage: two buffers of three-piece planes must fit in LDS
  const int CO = 2 * CPR;
  int u = 97 / (2 * CO + 3);
  if (32 / CPR < u) u = 32 / CPR;
  if (u > 16) u = 16;
  return u & ~1;
}
constexpr int wg3b_apos(int CPR) { return (2 * CPR + 3) * 2 * wg3b_U(CPR); }               // activation positions
constexpr int wg3b_kp(int CPR) { return ((2 * CPR * 2 * wg3b_U(CPR) + 15) / 16) * 16; }    // gradient rows (padded)
constexpr int wg3b_buf(int CPR) { return 3 * (wg3b_apos(CPR) + wg3b_kp(CPR)) * 128; }      // bytes per stage buffer
__device__ __forceinline__ int plane_byte(int pos, int ch) {
  return pos * 128 + (((ch >> 5) ^ ((pos >> 1) & 1)) << 6) + (ch & 31) * 2;
}
// four consecutive channels of one position -> the three planes
__device__ __forceinline__ void plane_put(char* plane0, int plane_stride, int byte, const float4& v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t u0[4], u1[4], u2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u0[j] = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(u0[j] & 0xffff0000u);
    u1[j] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1[j] & 0xffff0000u);
    u2[j] = __float_as_uint(r2);
  }
  *(uint2*)(plane0 + byte) = make_uint2(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]));
  *(uint2*)(plane0 + plane_stride + byte) = make_uint2(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]));
  *(uint2*)(plane0 + 2 * plane_stride + byte) = make_uint2(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]));
}
// ... and as TWO fp16 pieces of x * sc (common.hpp: h_split): truncated first piece, exact residual truncated
__device__ __forceinline__ void plane_put_h(char* plane0, int plane_stride, int byte, const float4& v, float sc) {
  const float x[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
  const f16x2v p0 = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
  const f16x2v p1 = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(x[2], x[3]));
  *(uint2*)(plane0 + byte) = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
  *(uint2*)(plane0 + plane_stride + byte) =
      make_uint2(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x[0] - (float)p0[0], x[1] - (float)p0[1])),
                 __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x[2] - (float)p1[0], x[3] - (float)p1[1])));
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// one MFMA operand (8 k per lane) = two transposed reads, 4 positions apart
__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 512));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// HP: both operands as TWO fp16 pieces of (value x its batch-level power-of-two scale) -- two planes each where the
// three-piece scheme has three, three MFMAs per product where it has six, a 12-instruction split per four elements
// where it has 22; the partial is multiplied by 1 / (sa sg) = `inv` (exact) when it is written.
// CP: the units are those of the `nnz` samples listed in `slist` (ascending), dealt to the G workgroups in equal contiguous
// shares: a sample whose gradient operand is zero everywhere contributes nothing and is not read.  An item's address is
// then formed from its unit through the list, stage by stage, instead of being carried.
template <int CPR, bool HP = false, bool CP = false>
__device__ __forceinline__ void wgrad3b_body(const Wgrad3Args& a, const int g, const int net, const int kh, float* smem,
                                             const float sa = 1.f, const float sg = 1.f, const float inv = 1.f,
                                             const uint16_t* slist = nullptr, const int nnz = 0) {
  constexpr int NT = 512, U = wg3b_U(CPR), R = 2 * U, CO = 2 * CPR;
  constexpr int APOS = wg3b_apos(CPR), KP = wg3b_kp(CPR), NST = KP / 16;
  constexpr int NP = HP ? 2 : 3;                                   // pieces = planes per operand
  constexpr int APL = APOS * 128, BPL = KP * 128, BUF = NP * (APOS + KP) * 128;
  static_assert(BUF <= wg3b_buf(CPR), "the launch sizes LDS for three-piece planes");
  constexpr int NRA = (R * (CO + 1) * 16 + NT - 1) / NT;          // activation items per thread and stage
  static_assert(U >= 2 && U * CPR * 16 <= NT, "stage geometry");
  char* lds = (char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, HW = H * W;
  const int H2 = H >> 1, W2 = W >> 1;
  const int UPS = H2;                       // units (pooled rows) per sample
  const int NU = (CP ? nnz : a.n) * UPS;
  const int UPG = CP ? ((NU + a.G - 1) / a.G + U - 1) / U * U : a.UPG;      // (CP: equal shares of what is left, whole stages)
  const int ubeg = (g * UPG < NU) ? g * UPG : NU, uend = (ubeg + UPG < NU) ? ubeg + UPG : NU;
  STAMP(2, 0);
  {  // What staging never writes must read as zero, in both buffers and all three pieces: the activation halo
     // columns x = 0 and x >= W + 1, and the gradient rows of the k padding.  (Everything else is rewritten by every
     // stage, rows outside the image as zeros.)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tail0 = (W + 1) * R, ntail = APOS - tail0;          // positions of the right halo (and beyond)
    const int nz = R + ntail + (KP - CO * R);                      // positions to clear per (buffer, piece)
    for (int i = tid; i < 2 * NP * nz * 8; i += NT) {              // 8 float4 per position
      const int f = i & 7, q = (i >> 3) % nz, bp = (i >> 3) / nz, buf = bp / NP, pc = bp - NP * buf;
      char* base = lds + buf * BUF;
      char* dst = q < R ? base + pc * APL + q * 128
                : q < R + ntail ? base + pc * APL + (tail0 + q - R) * 128
                : base + NP * APL + pc * BPL + (CO * R + q - R - ntail) * 128;
      *(float4*)(dst + f * 16) = z;
    }
  }
  const float* src = a.in + (long long)net * a.in_ns;
  const float* dp = a.dpool + (long long)net * a.dpool_ns;
  const uint8_t* mk = a.mask + (long long)net * a.dpool_ns;
  const int ct = wave & 1, it = (wave >> 1) & 1, kg = wave >> 2;

  // staging items of this thread (stage-invariant decode).  What is carried from stage to stage is each item's element
  // OFFSET and image row, advanced by constants (a stage = U pooled rows = qU samples + rU rows; a row index that runs
  // past its sample wraps into the next one): ~13 vector instructions per item and stage where carrying (sample, row)
  // and re-forming the address cost ~25 -- on a SIMD these add to the MFMA time, they do not hide under it.
  const int qU = U / UPS, rU = U - qU * UPS;
  const int dStep = (qU * HW + 2 * rU * W) * 64, dWrap = (HW - 2 * UPS * W) * 64;
  int a_u[NRA], a_dst[NRA], a_off[NRA], a_row[NRA], a_lim[NRA];
  int a_ir[NRA], a_in[NRA];                 // CP: row of the item inside its unit's row pair (+ kernel row - 1), offset inside an image row
#pragma unroll
  for (int q = 0; q < NRA; ++q) {
    const int t = tid + NT * q;
    const int c4 = t & 15, pr = t >> 4;
    const int rho = pr / W, px = pr - rho * W;
    const bool ex = rho < R;
    const int ir = (rho & 1) + kh - 1;
    a_u[q] = ex ? (rho >> 1) : (1 << 28);
    a_ir[q] = ir; a_in[q] = px * 64 + c4 * 4;
    a_dst[q] = plane_byte((px + 1) * R + (ex ? rho : 0), c4 * 4);
    const int smp = (ubeg + (rho >> 1)) / UPS, j = (ubeg + (rho >> 1)) - smp * UPS;
    a_row[q] = 2 * j + ir;                                 // image row of the item (-1 .. H: outside rows read as zero)
    a_lim[q] = 2 * UPS + ir;
    a_off[q] = (smp * HW + a_row[q] * W) * 64 + px * 64 + c4 * 4;
  }
  int d_u, d_off, d_dst[4];
  {
    const bool ex = tid < U * W2 * 16;
    const int id = ex ? tid : 0;
    const int c4 = id & 15, p = id >> 4;
    const int u = p / W2, pw = p - u * W2;
    d_u = ex ? u : (1 << 28);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) d_dst[sub] = plane_byte((2 * pw + (sub & 1)) * R + 2 * u + (sub >> 1), c4 * 4);
    d_off = (ubeg + u) * (W2 * 64) + pw * 64 + c4 * 4;    // pooled rows of consecutive samples are consecutive: linear
    if constexpr (CP) d_off = pw * 64 + c4 * 4;           // (the offset inside a pooled row; the row comes from the list)
  }
  float4 pa[NRA];
  float4 pdd = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t pdm = 0u;
  float4 dbsum = make_float4(0.f, 0.f, 0.f, 0.f);   // bias gradient of channels 4*(tid&15)..+3 (kernel row 0 only)
  // fetch(ub): the global loads of the stage that begins at unit ub (in order: it advances the carried offsets)
  auto fetch = [&](int ub) {
    const int left = uend - ub;                          // units of this workgroup from ub on (uniform)
    if constexpr (CP) {
#pragma unroll
      for (int q = 0; q < NRA; ++q) {
        const bool in = a_u[q] < left;                     // (a_u of an item that does not exist is huge)
        const int v = in ? ub + a_u[q] : 0;
        const int si = (int)__umulhi((uint32_t)v, a.mg_ups), j = v - si * UPS;
        const int smp = slist[si], row = 2 * j + a_ir[q];
        const bool ok = in && (unsigned)row < (unsigned)H;
        const float4 x = *(const float4*)(src + (ok ? (smp * HW + row * W) * 64 + a_in[q] : 0));
        pa[q] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const bool ok = d_u < left;
      const int v = ok ? ub + d_u : 0;
      const int si = (int)__umulhi((uint32_t)v, a.mg_ups), j = v - si * UPS;
      const int gi = ok ? ((int)slist[si] * UPS + j) * (W2 * 64) + d_off : 0;
      pdd = *(const float4*)(dp + gi);
      pdm = ok ? *(const uint32_t*)(mk + gi) : 0u;
      return;
    }
#pragma unroll
    for (int q = 0; q < NRA; ++q) {
      const bool ok = a_u[q] < left && (unsigned)a_row[q] < (unsigned)H;      // (a_u of an item that does not exist is huge)
      const float4 v = *(const float4*)(src + (ok ? a_off[q] : 0));
      pa[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      a_row[q] += 2 * rU; a_off[q] += dStep;
      if (a_row[q] >= a_lim[q]) { a_row[q] -= 2 * UPS; a_off[q] += dWrap; }
    }
    {
      const bool ok = d_u < left;
      const int gi = ok ? d_off : 0;
      pdd = *(const float4*)(dp + gi);
      pdm = ok ? *(const uint32_t*)(mk + gi) : 0u;
      d_off += U * (W2 * 64);
    }
  };
  // commit(buf): split what fetch() loaded and write the planes of one stage buffer
  auto commit = [&](char* buf) {
#pragma unroll
    for (int q = 0; q < NRA; ++q)
      if (a_u[q] < (1 << 28)) {
        if constexpr (HP) plane_put_h(buf, APL, a_dst[q], pa[q], sa);
        else plane_put(buf, APL, a_dst[q], pa[q]);
      }
    if (d_u < (1 << 28)) {
      // the four positions of a pooled item's 2x2 window carry the SAME value d / 4, each behind its own gate bit:
      // split once, then gate the packed pieces (a 16-bit lane mask per channel) -- 4 x (gate, scale, split) before
      const float q[4] = {pdd.x * 0.25f, pdd.y * 0.25f, pdd.z * 0.25f, pdd.w * 0.25f};
      uint2 p0, p1, p2 = make_uint2(0u, 0u);
      if constexpr (HP) {
        const float x[4] = {q[0] * sg, q[1] * sg, q[2] * sg, q[3] * sg};
        const f16x2v h0 = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
        const f16x2v h1 = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(x[2], x[3]));
        p0 = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
        p1 = make_uint2(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x[0] - (float)h0[0], x[1] - (float)h0[1])),
                        __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x[2] - (float)h1[0], x[3] - (float)h1[1])));
      } else {
        uint32_t u0[4], u1[4], u2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u0[j] = __float_as_uint(q[j]);
          const float r1 = q[j] - __uint_as_float(u0[j] & 0xffff0000u);
          u1[j] = __float_as_uint(r1);
          const float r2 = r1 - __uint_as_float(u1[j] & 0xffff0000u);
          u2[j] = __float_as_uint(r2);
        }
        p0 = make_uint2(hi_pair(u0[0], u0[1]), hi_pair(u0[2], u0[3]));
        p1 = make_uint2(hi_pair(u1[0], u1[1]), hi_pair(u1[2], u1[3]));
        p2 = make_uint2(hi_pair(u2[0], u2[1]), hi_pair(u2[2], u2[3]));
      }
      char* bpl = buf + NP * APL;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        // gate bit of channel j and window position sub: bit 8 j + sub of the mask word -> all-ones / zero
        const uint32_t s0 = (uint32_t)((int32_t)(pdm << (31 - sub)) >> 31), s1 = (uint32_t)((int32_t)(pdm << (23 - sub)) >> 31);
        const uint32_t s2 = (uint32_t)((int32_t)(pdm << (15 - sub)) >> 31), s3 = (uint32_t)((int32_t)(pdm << (7 - sub)) >> 31);
        const uint32_t m01 = (s0 & 0xffffu) | (s1 & 0xffff0000u), m23 = (s2 & 0xffffu) | (s3 & 0xffff0000u);
        *(uint2*)(bpl + d_dst[sub]) = make_uint2(p0.x & m01, p0.y & m23);
        *(uint2*)(bpl + BPL + d_dst[sub]) = make_uint2(p1.x & m01, p1.y & m23);
        if constexpr (!HP) *(uint2*)(bpl + 2 * BPL + d_dst[sub]) = make_uint2(p2.x & m01, p2.y & m23);
      }
      // bias gradient: the value times the number of open gates of its window
      // (a closed window contributes 0 whatever its value, as the per-position gate did)
      const uint32_t g0 = pdm & 0xfu, g1 = pdm & 0xf00u, g2 = pdm & 0xf0000u, g3 = pdm & 0xf000000u;
      dbsum.x += g0 ? q[0] * (float)__popc(g0) : 0.f; dbsum.y += g1 ? q[1] * (float)__popc(g1) : 0.f;
      dbsum.z += g2 ? q[2] * (float)__popc(g2) : 0.f; dbsum.w += g3 ? q[3] * (float)__popc(g3) : 0.f;
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) acc[s] = zero16();
  __syncthreads();                 // zero fill complete
  if (ubeg < uend) { fetch(ubeg); commit(lds); }
  // (Measured, round 3: with all eight waves in lockstep a stage costs MFMAs 1.06 us + split / plane writes 0.85 us +
  // global-load issue and wait 0.55 us, nothing overlapped -- and letting waves 4..7 do their staging BEFORE their
  // MFMAs, with their loads one stage further ahead, made the launch 3 us slower: every wave's own chain is still
  // MFMAs + staging, the order does not shorten it.)
  __syncthreads();
  STAMP(2, 1);
  // this lane's part of a transposed read: 16-lane group gq = lane >> 4 covers k half (gq >> 1) and channel block
  // (gq & 1) of the wave's tile; inside it lane 4q + p addresses row q, channels 4p..4p+3.  The block base is a
  // multiple of 4 positions, so the swizzle bit of the lane's row is (q >> 1) & 1 in every read.
  const int gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int rowb = (8 * (gq >> 1) + qq) * 128 + 32 * (gq & 1) + 8 * pp;
  const int sw = (qq >> 1) & 1;
  const int a_lane = rowb + ((it ^ sw) << 6);
  const int b_lane = NP * APL + rowb + ((ct ^ sw) << 6);
  int cur = 0;
  for (int ub = ubeg; ub < uend; ub += U) {
    const bool more = ub + U < uend;            // workgroup-uniform
    const char* buf = lds + cur * BUF;
    // (ablation builds: 30 = no global fetch after the first stage, 31 = no split / plane writes after the first stage,
    //  32 = no MFMA loop -- wrong results on purpose, timeline only)
    if (more && CMLPL_ABL != 30) fetch(ub + U); // in flight across the MFMAs below
    const char* pa_ = buf + a_lane;
    const char* pb_ = buf + b_lane;
#pragma unroll
    for (int st = 0; st < (NST + 1) / 2; ++st) {
      const int step = 2 * st + kg;             // wave-uniform
      if (step < NST && CMLPL_ABL != 32) {
        const char* pbs = pb_ + step * 2048;
        if constexpr (HP) {
          const f16x8v b1 = __builtin_bit_cast(f16x8v, tr_frag(pbs)), b2 = __builtin_bit_cast(f16x8v, tr_frag(pbs + BPL));
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const char* pas = pa_ + step * 2048 + kw * (R * 128);
            const f16x8v a1 = __builtin_bit_cast(f16x8v, tr_frag(pas)), a2 = __builtin_bit_cast(f16x8v, tr_frag(pas + APL));
            acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc[kw], 0, 0, 0);
            acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc[kw], 0, 0, 0);
            acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[kw], 0, 0, 0);
          }
        } else {
        const bf16x8 b1 = tr_frag(pbs), b2 = tr_frag(pbs + BPL), b3 = tr_frag(pbs + 2 * BPL);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const char* pas = pa_ + step * 2048 + kw * (R * 128);
          const bf16x8 a1 = tr_frag(pas), a2 = tr_frag(pas + APL), a3 = tr_frag(pas + 2 * APL);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[kw], 0, 0, 0);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[kw], 0, 0, 0);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[kw], 0, 0, 0);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[kw], 0, 0, 0);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[kw], 0, 0, 0);
          acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[kw], 0, 0, 0);
        }
        }
      }
    }
    if (more && CMLPL_ABL != 31) commit(lds + (cur ^ 1) * BUF);
    __syncthreads();   // stage g fully read by every wave, stage g+1 fully written
    cur ^= 1;
  }
  STAMP(2, 2);

  // fold pixel group 1 into group 0 through LDS ([4 waves][48][64]); group 0 stores the partial
  float* red = smem + (size_t)(wave & 3) * 48 * 64 + lane;
  if (kg == 1) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(s * 16 + r) * 64] = acc[s][r];
  }
  float4* dbl = (float4*)(smem + 4 * 48 * 64);    // [512] per-thread bias partial sums
  if (kh == 0) dbl[tid] = dbsum;
  __syncthreads();
  float* part = a.part + (long long)net * a.part_ns + (size_t)g * PART3;
  if (kg == 0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = it * 32 + acc_row(r, lane);
        const float v = acc[s][r] + red[(s * 16 + r) * 64];
        part[(3 * kh + s) * 4096 + ci * 64 + ct * 32 + l31] = HP ? v * inv : v;
      }
    }
  }
  // bias gradient (kernel row 0 only): thread t holds channels 4*(t & 15)..+3; fixed-order sum over the 32 threads
  // of each channel quad
  if (kh == 0 && tid < 64) {
    const int c4 = tid >> 2, e = tid & 3;
    float sum = 0.f;
    for (int k = 0; k < 32; ++k) sum += ((const float*)&dbl[c4 + 16 * k])[e];
    part[9 * 4096 + tid] = sum;
  }
  STAMP(2, 3);
}

// three-piece planes, or two-piece ones when this step's fused kernels left both operands' per-sample maxima: the
// workgroup first folds them to the BATCH maxima (the accumulators sum over samples: one scale per operand), and takes the
// two-piece planes when both are ordinary numbers and the network's weights were inside the two-piece range (the flag
// marks a numerically extreme network: everything of it stays on three pieces).  Workgroup-uniform.
template <int CPR>
__device__ __forceinline__ void wgrad3b_run(const Wgrad3Args& a, const int g, const int net, const int kh, float* smem) {
  if (a.stat_a != nullptr && a.stat_g != nullptr) {
    uint32_t* s_stat = (uint32_t*)smem;          // (two words of the plane region, handed back behind the third barrier)
    const int tid = threadIdx.x;
    if (tid < 2) s_stat[tid] = 0u;
    uint32_t ma = 0u, mg = 0u;
    for (int i = tid; i < a.n; i += 512) {
      const uint32_t x = a.stat_a[(long long)net * a.n + i], y = a.stat_g[(long long)net * a.n + i];
      ma = x > ma ? x : ma; mg = y > mg ? y : mg;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const uint32_t x = (uint32_t)__shfl_xor((int)ma, o, 64), y = (uint32_t)__shfl_xor((int)mg, o, 64);
      ma = x > ma ? x : ma; mg = y > mg ? y : mg;
    }
    __syncthreads();
    if ((tid & 63) == 0) { atomicMax(&s_stat[0], ma); atomicMax(&s_stat[1], mg); }
    __syncthreads();
    const uint32_t ea = __builtin_amdgcn_readfirstlane(s_stat[0]) >> 23, eg = __builtin_amdgcn_readfirstlane(s_stat[1]) >> 23;
    __syncthreads();
    const uint32_t flag = a.h2flag != nullptr ? a.h2flag[(long long)net * a.h2flag_ns] : 0u;
    if (flag == 0u && ea >= 40u && ea <= 200u && eg >= 40u && eg <= 200u && ea + eg >= 160u) {
      // scales 2^(14 - floor(log2 max)): the largest magnitude of either operand lands in [2^14, 2^15)
      const float sa = __uint_as_float((268u - ea) << 23), sg = __uint_as_float((268u - eg) << 23);
      const float inv = __uint_as_float((ea + eg - 155u) << 23);
      if (a.slist_off > 0) {
        // the samples with a non-zero gradient operand, in ascending order (a ballot prefix per wave, the waves' counts
        // through LDS): when some are missing, only these are walked
        uint16_t* slist = (uint16_t*)((char*)smem + a.slist_off);
        int* wcnt = (int*)smem + 4;
        const int lane = tid & 63, wave = tid >> 6;
        int nnz = 0;
        for (int i0 = 0; i0 < a.n; i0 += 512) {
          const int i = i0 + tid;
          // (a sample with a zero gradient operand stays in when its activations were not finite: 0 x inf is NaN)
          const bool nzf = i < a.n && (a.stat_g[(long long)net * a.n + i] != 0u || (a.stat_a[(long long)net * a.n + i] >> 23) == 255u);
          const unsigned long long bal = __ballot(nzf);
          if (lane == 0) wcnt[wave] = __popcll(bal);
          __syncthreads();
          int woff = 0, tot = 0;
#pragma unroll
          for (int w = 0; w < 8; ++w) { const int c = wcnt[w]; woff += (w < wave) ? c : 0; tot += c; }
          if (nzf) slist[nnz + woff + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
          nnz += tot;
          __syncthreads();
        }
        if (nnz < a.n) {
          wgrad3b_body<CPR, true, true>(a, g, net, kh, smem, sa, sg, inv, slist, nnz);
          return;
        }
      }
      wgrad3b_body<CPR, true>(a, g, net, kh, smem, sa, sg, inv);
      return;
    }
  }
  wgrad3b_body<CPR, false>(a, g, net, kh, smem);
}

template <int CPR>
__global__ __launch_bounds__(512) void wgrad3b_kernel(Wgrad3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad3b_run<CPR>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}
// conv1's and conv2's weight gradients of one backward pass in ONE launch (one ramp and one drain instead of two).
// A workgroup fills a CU and workgroups go to the XCDs round-robin in launch order, so ALL of conv1's long
// workgroups come first in a 1-D grid (<= one per CU: they all start at once) and conv2's short ones follow into
// the CUs as they free up.  (Interleaved per (net, kernel row) as a 3-D grid would have them, late conv1 workgroups
// queue behind early ones on their XCD: 48.9 us against 38.2 for the two separate launches.)
template <int CPRA, int CPRB>
__global__ __launch_bounds__(512) void wgrad3b_pair_kernel(Wgrad3Args a, Wgrad3Args b, int nets) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int na = a.G * nets * 3;
  int id = (int)blockIdx.x;
  if (id < na) { const int g = id % a.G, r = id / a.G; wgrad3b_run<CPRA>(a, g, r % nets, r / nets, smem); }
  else { id -= na; const int g = id % b.G, r = id / b.G; wgrad3b_run<CPRB>(b, g, r % nets, r / nets, smem); }
}

static size_t wgrad3_lds(int RU, int U, int W, int cspl = 1) {
  const int PW = W + 2, CO = 2 * (W / 2);
  const size_t D = (size_t)U * RU * CO;
  return ((size_t)U * (RU + 2) * PW * 64 + D * (64 / cspl)) * 4;
}


bool plan_wgrad3(int nets, int n, int H, int W, Wgrad3Plan* p, int role) {
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
  const int force_u = switches().wgrad3_u;
  int U = 1;
  while (U < 8 && wgrad3_lds(RU, U + 1, W) <= LDS_MAX && (NU + U) / (U + 1) >= 128) ++U;
  if (force_u > 0 && wgrad3_lds(RU, force_u, W) <= LDS_MAX) U = force_u;
  // Experiment (CMLPL_WGRAD3_CSPL=2): split the output channels over two workgroups and walk the units one at
  // a time so that 2-3 workgroups are co-resident per CU.  Measured on B2/256: 67.5 us vs 59.0 us for conv1 --
  // the image is staged twice and that costs more than the overlap buys -- so it is off by default.
  const int force_c = switches().wgrad3_cspl;
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
  p->rsplit = 0; p->UPG = 0; p->b3 = 0;
  const int CO = 2 * (W / 2);
  // row-split kernel: blockIdx.z = kernel row; units of one pooled row, U (even) per stage
  const int rsp = switches().wgrad3_r;
  const int force_ru = switches().wgrad3_ru;
  const int force_rg = switches().wgrad3_rg;
  if (rsp && p->cspl == 1 && CO >= 2 && (long long)n * H * W * 64 < (1LL << 31)) {
    const int PW = W + 2, cpr = CO / 2;
    const long long NUr = (long long)n * (H / 2);
    const int CUS = device_cus();                              // 256 on an MI355X; fewer on a partitioned or smaller part
    long long Gt = CUS / (3 * nets);                           // one workgroup per CU over (chunks, nets, 3 rows)
    // In the pair launch (wgrad3b_pair_kernel) a workgroup of either map fills a CU, so the first map's workgroups
    // must not take ALL the CUs: with 240 of them the second map's 192 short ones ran almost entirely AFTER the first
    // map's had finished (25 us + 10 us).  The first map gets 3/4 of the CUs (192 workgroups of ten stages instead of
    // 240 of eight at B2), the second map fewer, longer workgroups (their ~4 us of prologue + epilogue weighs less) on
    // the CUs left over, and both end together.  Measured (B2, launch + reduce): (40, 32) groups per network and kernel
    // row 36.5 + 13.3 us, (36, 16) 38.2 + 12.4, (32, 16) 33.8 + 12.2, (32, 12) 33.5 + 12.1, (28, 12) 37.9 + 11.9.  The
    // second map's workgroup count must not exceed the CUs left over: stragglers of a second round run on after the first
    // map has finished (512 + 512 rows on one GPU: 0.680 -> 0.703 ms with 72 workgroups for 64 CUs).
    const int force_pg1 = switches().wgrad3_pg1;
    const int force_pg2 = switches().wgrad3_pg2;
    if (role == 1) { Gt = (CUS * 3 / 4) / (3 * nets); if (force_pg1 > 0) Gt = force_pg1; }
    if (role == 2) {
      long long g1 = (CUS * 3 / 4) / (3 * nets);
      if (force_pg1 > 0) g1 = force_pg1;
      const long long spare = CUS - 3 * nets * g1;
      Gt = spare / (3 * nets);                                  // ONE round on the CUs left over, at every batch size
      if (force_pg2 > 0) Gt = force_pg2;
    }
    if (force_rg > 0) Gt = force_rg;
    if (Gt < 1) Gt = 1;
    const size_t red = (size_t)(4 * 48 * 64 + 512 * 4) * 4;   // fold area + per-thread bias sums
    // split-bf16 variant (default wherever two three-piece stage buffers fit in LDS: every window the 3x3 kernels
    // themselves can hold)
    const int b3on = switches().wgrad3_b3;
    if (b3on && cpr <= WG3B_MAXCPR) {
      const int Ub = wg3b_U(cpr);
      long long upgb = (NUr + Gt - 1) / Gt;
      upgb = ((upgb + Ub - 1) / Ub) * Ub;
      const size_t need = 2 * (size_t)wg3b_buf(cpr);
      p->RU = 2; p->U = Ub; p->UPG = (int)upgb; p->G = (int)((NUr + upgb - 1) / upgb);
      p->lds = need > red ? need : red;
      p->rsplit = cpr; p->b3 = 1;
      return true;
    }
    // f32-input MFMA row-split kernel (round 2a's): instantiated for the two maps of the headline shape only
    // (11 x 11 -> CPR 5, 5 x 5 -> CPR 2), as the reference point for the split-bf16 kernel (CMLPL_WGRAD3_B3=0)
    auto lds_r = [&](int u) {
      const size_t buf2 = 2 * (size_t)u * (2 * PW * 64 + 2 * CO * 64) * 4;
      return buf2 > red ? buf2 : red;
    };
    auto ni_r = [&](int u) { return (2 * u * ((W + 3) / 4) + 7) / 8; };        // LDS-DMA pieces per wave
    auto nd_r = [&](int u) { return (u * (W / 2) * 16 + 511) / 512; };
    auto fits_r = [&](int u) { return lds_r(u) <= LDS_MAX && ni_r(u) <= WG3R_NTK && nd_r(u) <= 1; };
    int Ur = 2 * ((16 + 2 * cpr - 1) / (2 * cpr));           // >= 16 pixel pairs per wave and stage
    if (force_ru > 0) Ur = force_ru & ~1;
    while (Ur > 2 && !fits_r(Ur)) Ur -= 2;
    if ((cpr == 2 || cpr == 5) && Ur >= 2 && fits_r(Ur)) {
      long long upg = (NUr + Gt - 1) / Gt;
      upg = ((upg + Ur - 1) / Ur) * Ur;
      const long long Gr = (NUr + upg - 1) / upg;
      p->RU = 2; p->U = Ur; p->G = (int)Gr; p->UPG = (int)upg; p->lds = lds_r(Ur);
      p->rsplit = cpr;                                          // = template parameter CPR of wgrad3r_kernel
    }
  }
  return true;
}

hipError_t launch_wgrad3(int nets, int n, int H, int W, const float* in, const float* dpool, const uint8_t* mask,
                         float* part, hipStream_t st) {
  Wgrad3Plan pl;
  if (!plan_wgrad3(nets, n, H, W, &pl, 0)) return hipErrorInvalidValue;
  static DevOnce attr_once;
  {
    hipError_t e = ensure_max_lds(attr_once, wgrad3_kernel<1>, wgrad3_kernel<2>);
    if (e != hipSuccess) return e;
  }
  Wgrad3Args a;
  a.in = in; a.dpool = dpool; a.mask = mask; a.part = part;
  a.in_ns = (long long)n * H * W * 64;
  a.dpool_ns = (long long)n * (H / 2) * (W / 2) * 64;
  a.part_ns = (long long)pl.G * PART3;
  a.n = n; a.H = H; a.W = W; a.RU = pl.RU; a.U = pl.U; a.G = pl.G; a.UPG = pl.UPG;
  a.stat_a = nullptr; a.stat_g = nullptr; a.h2flag = nullptr; a.h2flag_ns = 0; a.slist_off = 0; a.mg_ups = 0;
  if (pl.rsplit && pl.b3) {
#define WG3B_CASE(CPR_)                                                                              \
    case CPR_: {                                                                                     \
      static DevOnce attr_b;                                                                         \
      hipError_t e = ensure_max_lds(attr_b, wgrad3b_kernel<CPR_>);                                   \
      if (e != hipSuccess) return e;                                                                 \
      hipLaunchKernelGGL((wgrad3b_kernel<CPR_>), dim3(pl.G, nets, 3), dim3(512), pl.lds, st, a);     \
      return hipGetLastError();                                                                      \
    }
    switch (pl.rsplit) {
      WG3B_CASE(1) WG3B_CASE(2) WG3B_CASE(3) WG3B_CASE(4) WG3B_CASE(5) WG3B_CASE(6) WG3B_CASE(7) WG3B_CASE(8)
      WG3B_CASE(9) WG3B_CASE(10) WG3B_CASE(11)
      default: return hipErrorInvalidValue;
    }
#undef WG3B_CASE
  }
  if (pl.rsplit) {
#define WG3R_CASE(CPR_)                                                                              \
    case CPR_: {                                                                                     \
      static DevOnce attr_r;                                                                         \
      hipError_t e = ensure_max_lds(attr_r, wgrad3r_kernel<CPR_>);                                   \
      if (e != hipSuccess) return e;                                                                 \
      hipLaunchKernelGGL((wgrad3r_kernel<CPR_>), dim3(pl.G, nets, 3), dim3(512), pl.lds, st, a);     \
      return hipGetLastError();                                                                      \
    }
    switch (pl.rsplit) {
      WG3R_CASE(2) WG3R_CASE(5)
      default: return hipErrorInvalidValue;
    }
#undef WG3R_CASE
  }
  if (pl.cspl == 2) hipLaunchKernelGGL(wgrad3_kernel<2>, dim3(pl.G, nets, 2), dim3(256), pl.lds, st, a);
  else              hipLaunchKernelGGL(wgrad3_kernel<1>, dim3(pl.G, nets), dim3(256), pl.lds, st, a);
  return hipGetLastError();
}

static void wgrad3_args(Wgrad3Args& a, const Wgrad3Plan& pl, int n, int H, int W, const float* in,
                        const float* dpool, const uint8_t* mask, float* part) {
  a.in = in; a.dpool = dpool; a.mask = mask; a.part = part;
  a.in_ns = (long long)n * H * W * 64;
  a.dpool_ns = (long long)n * (H / 2) * (W / 2) * 64;
  a.part_ns = (long long)pl.G * PART3;
  a.n = n; a.H = H; a.W = W; a.RU = pl.RU; a.U = pl.U; a.G = pl.G; a.UPG = pl.UPG;
  a.stat_a = nullptr; a.stat_g = nullptr; a.h2flag = nullptr; a.h2flag_ns = 0; a.slist_off = 0; a.mg_ups = 0;
}

// Plans of both 3x3 weight gradients of a backward pass.  *pair: one launch (wgrad3b_pair_kernel) -- then the two plans
// share the CUs between them (roles 1 / 2 above); otherwise each map is planned for a launch of its own.  want_pair:
// the caller would use the pair launch (the fused backward does; the general path launches the two on side streams).
// Every user of the group counts (workspace carving, the reduce table, the launch) goes through here.
static bool wgrad3_pair_instantiated(int ca, int cb) {
  return (ca == 4 && cb == 2) || (ca == 5 && cb == 2) || (ca == 6 && cb == 3) || (ca == 7 && cb == 3) ||
         (ca == 8 && cb == 4) || (ca == 9 && cb == 4) || (ca == 10 && cb == 5);
}
bool plan_wgrad3_both(int nets, int n, int H1, int W1, int H2, int W2, bool want_pair, Wgrad3Plan* p1, Wgrad3Plan* p2,
                      bool* pair) {
  const bool off = switches().wgrad3_pair == 0;
  *pair = false;
  if (!plan_wgrad3(nets, n, H1, W1, p1, 0) || !plan_wgrad3(nets, n, H2, W2, p2, 0)) return false;
  if (want_pair && !off && p1->b3 && p2->b3 && wgrad3_pair_instantiated(p1->rsplit, p2->rsplit)) {
    Wgrad3Plan q1, q2;
    if (plan_wgrad3(nets, n, H1, W1, &q1, 1) && plan_wgrad3(nets, n, H2, W2, &q2, 2) && q1.b3 && q2.b3) {
      *p1 = q1; *p2 = q2; *pair = true;
    }
  }
  return true;
}

// both 3x3 weight gradients of a backward pass: one launch where a pair kernel exists, else two
hipError_t launch_wgrad3_pair(int nets, int n, int H1, int W1, const float* in1, const float* dpool1,
                              const uint8_t* mask1, float* part1, int H2, int W2, const float* in2,
                              const float* dpool2, const uint8_t* mask2, float* part2, bool* merged, hipStream_t st,
                              const uint32_t* hstat, const uint32_t* h2flag, long long h2flag_ns) {
  Wgrad3Plan p1, p2;
  *merged = false;
  bool pair = false;
  if (!plan_wgrad3_both(nets, n, H1, W1, H2, W2, true, &p1, &p2, &pair)) return hipErrorInvalidValue;
  if (pair) {
    Wgrad3Args a, b;
    wgrad3_args(a, p1, n, H1, W1, in1, dpool1, mask1, part1);
    wgrad3_args(b, p2, n, H2, W2, in2, dpool2, mask2, part2);
    if (hstat != nullptr) {     // [kind][2 networks][n]: a0, p1, conv1's gradient operand, conv2's
      a.stat_a = hstat + 0 * 2 * n; a.stat_g = hstat + 2 * 2 * n; b.stat_a = hstat + 1 * 2 * n; b.stat_g = hstat + 3 * 2 * n;
      a.h2flag = b.h2flag = h2flag; a.h2flag_ns = b.h2flag_ns = h2flag_ns;
    }
    size_t lds = p1.lds > p2.lds ? p1.lds : p2.lds;
    // room for the list of samples with a non-zero gradient operand behind the stage buffers: zero samples are skipped
    // (CMLPL_ZERO_SKIP=0: never)
    const size_t lbytes = ((size_t)2 * n + 15) & ~(size_t)15;
    if (hstat != nullptr && switches().zero_skip != 0 && n <= 65535 && H1 / 2 >= 2 && H2 / 2 >= 2 && lds + lbytes <= LDS_MAX) {
      a.slist_off = b.slist_off = (int)lds;
      a.mg_ups = (uint32_t)((0x100000000ULL + (uint32_t)(H1 / 2) - 1) / (uint32_t)(H1 / 2));
      b.mg_ups = (uint32_t)((0x100000000ULL + (uint32_t)(H2 / 2) - 1) / (uint32_t)(H2 / 2));
      lds += lbytes;
    }
    const dim3 grid((p1.G + p2.G) * nets * 3);
#define WG3P_CASE(CA_, CB_)                                                                          \
    if (p1.rsplit == CA_ && p2.rsplit == CB_) {                                                      \
      static DevOnce attr_p;                                                                         \
      hipError_t e = ensure_max_lds(attr_p, wgrad3b_pair_kernel<CA_, CB_>);                          \
      if (e != hipSuccess) return e;                                                                 \
      hipLaunchKernelGGL((wgrad3b_pair_kernel<CA_, CB_>), grid, dim3(512), lds, st, a, b, nets);     \
      *merged = true;                                                                                \
      return hipGetLastError();                                                                      \
    }
    // windows 8..21: (W / 2, W / 4); B2 / B4 = (5, 2), P = (10, 5), B5 = (7, 3)
    WG3P_CASE(4, 2) WG3P_CASE(5, 2) WG3P_CASE(6, 3) WG3P_CASE(7, 3) WG3P_CASE(8, 4) WG3P_CASE(9, 4) WG3P_CASE(10, 5)
#undef WG3P_CASE
  }
  hipError_t e = launch_wgrad3(nets, n, H2, W2, in2, dpool2, mask2, part2, st);
  if (e != hipSuccess) return e;
  return launch_wgrad3(nets, n, H1, W1, in1, dpool1, mask1, part1, st);
}

}  // namespace cmlpl
