// NT-Xent contrastive loss of the reference's tools.models.ContrastiveLoss (tools/models.py:14-39):
//   z = normalize([emb_i ; emb_j])   S = z z^T (cosine similarity)   positives on the +-B diagonals
//   loss = mean_a( -log( exp(S[a,p(a)]/T) / sum_{b != a} exp(S[a,b]/T) ) )
// forward + analytic backward in TWO launches (round 3: five -- normalise, S, rows, two gradient GEMMs, embedding
// gradient -- 0.2 GFLOP spread over five ramps and a k-major copy of z):
//   ntx_sim_kernel  : 16 x 32 tiles of the RAW products x_a . x_b on the fp32 MFMA, the contraction split over the four
//                     waves, whole 128-byte lines staged through per-wave LDS tiles (the shape of loss.hip's
//                     pair_exp16_kernel).  The rows' squared norms are summed from the very elements the tile stages,
//                     so no normalised copy of the embeddings is ever written: S = raw / (|x_a| |x_b|) in the
//                     epilogue, plus the tile's share of every row's denominator.
//   ntx_grad_kernel : one workgroup per (8 rows, 256 embedding columns).  It folds the denominators of ALL rows from the
//                     column-tile partials (a row's gradient needs the other rows' denominators: the one global
//                     dependency of this loss, hence the launch boundary), forms its rows of
//                         W[a][b] = (E[a][b] (1/den_a + 1/den_b) - 2 [b = p(a)]) / (2B T)      (= G + G^T, E = exp(S/T))
//                     in LDS, and streams the embeddings once: dz[a] = sum_b W[a][b] z[b].  The backward of the
//                     normalisation needs z_a . dz_a = sum_b W[a][b] S[a][b] -- a row of numbers the workgroup already
//                     holds, so no column slice waits for another.  Workgroup (0, 0) also sums the loss.
#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

struct NtxArgs {
  const float* ei; const float* ej; int B, D; float T;
  float *S, *rs_part, *nrm, *loss, *gi, *gj;
};

__device__ __forceinline__ const float* ntx_row(const NtxArgs& a, int r) {
  return (r < a.B) ? a.ei + (long long)r * a.D : a.ej + (long long)(r - a.B) * a.D;
}

constexpr int NPT = 36;              // per-wave staging tile row stride in floats (32 + 4)

__global__ __launch_bounds__(256) void ntx_sim_kernel(NtxArgs a) {
  constexpr int WT = 48 * NPT;                                       // per-wave staging: A [16][NPT] + B [32][NPT]
  __shared__ __attribute__((aligned(16))) float lds[4 * WT];         // later reused: red[4][8][64], then the E tile
  __shared__ float ssq[4][48];                                       // per-wave squared-norm partials of the 48 rows
  float (*red)[8][64] = (float (*)[8][64])lds;
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N2 = 2 * a.B, D = a.D;
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 32;
  float* tA = lds + wave * WT;
  float* tB = tA + 16 * NPT;
  // loader role: lane -> (row group r8 = lane >> 3, 16-byte chunk c8 = lane & 7); load j covers rows 8j + r8.  The
  // contraction is cut into 32-float lines; wave w takes lines w, w + 4, ... (any D).
  const int r8 = lane >> 3, c8 = lane & 7;
  const int nlines = (D + 31) >> 5;
  const float* pa[2]; const float* pb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int r = r0 + 8 * j + r8; pa[j] = ntx_row(a, r < N2 ? r : 0); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int r = c0 + 8 * j + r8; pb[j] = ntx_row(a, r < N2 ? r : 0); }
  float* wA = tA + r8 * NPT + c8 * 4;
  float* wB = tB + r8 * NPT + c8 * 4;
  const float* rA = tA + l16 * NPT + kq * 8;     // lane group kq owns floats [8kq, 8kq+8) of the line
  const float* rB = tB + l16 * NPT + kq * 8;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float sq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // (rows are only 4-byte aligned when D is not a multiple of 4: dword-aligned 16-byte loads are fine on this
  //  hardware; the ragged last group of a row is taken element by element)
  auto ld = [&](const float* p, int ln) {
    const int o = ln * 32 + c8 * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ln < nlines && o + 3 < D) v = *(const float4*)(p + o);
    else if (ln < nlines && o < D) { v.x = p[o]; if (o + 1 < D) v.y = p[o + 1]; if (o + 2 < D) v.z = p[o + 2]; }
    return v;
  };
  // (one line of look-ahead; three line sets in flight -- pair_exp16's recipe -- made this kernel SLOWER, 13 -> 17 us at
  //  2B = 256 and 43 -> 81 us at 2B = 1024: measured and reverted, round 4)
  float4 va[2], vb[4], na[2], nb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) va[j] = ld(pa[j], wave);
#pragma unroll
  for (int j = 0; j < 4; ++j) vb[j] = ld(pb[j], wave);
  for (int ln = wave; ln < nlines; ln += 4) {
#pragma unroll
    for (int j = 0; j < 2; ++j) na[j] = ld(pa[j], ln + 4);           // next line in flight
#pragma unroll
    for (int j = 0; j < 4; ++j) nb[j] = ld(pb[j], ln + 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *(float4*)(wA + 8 * j * NPT) = va[j];
      sq[j] += (va[j].x * va[j].x + va[j].y * va[j].y) + (va[j].z * va[j].z + va[j].w * va[j].w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(float4*)(wB + 8 * j * NPT) = vb[j];
      sq[2 + j] += (vb[j].x * vb[j].x + vb[j].y * vb[j].y) + (vb[j].z * vb[j].z + vb[j].w * vb[j].w);
    }
    // (only this wave touches its staging tile: its own writes are ordered before its reads, no block barrier)
    const float4 x0 = *(const float4*)(rA), x1 = *(const float4*)(rA + 4);
    const float4 y0 = *(const float4*)(rB), y1 = *(const float4*)(rB + 4);
    const float4 z0 = *(const float4*)(rB + 16 * NPT), z1 = *(const float4*)(rB + 16 * NPT + 4);
#define CMLPL_M2(XA, YB, ZB)                                                                 \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(XA, YB, acc0, 0, 0, 0);                      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(XA, ZB, acc1, 0, 0, 0);
    CMLPL_M2(x0.x, y0.x, z0.x) CMLPL_M2(x0.y, y0.y, z0.y) CMLPL_M2(x0.z, y0.z, z0.z) CMLPL_M2(x0.w, y0.w, z0.w)
    CMLPL_M2(x1.x, y1.x, z1.x) CMLPL_M2(x1.y, y1.y, z1.y) CMLPL_M2(x1.z, y1.z, z1.z) CMLPL_M2(x1.w, y1.w, z1.w)
#undef CMLPL_M2
#pragma unroll
    for (int j = 0; j < 2; ++j) va[j] = na[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[j] = nb[j];
  }
  // squared norms: the 8 lanes of a row group (c8) hold its chunks of this wave's lines
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float v = sq[j];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (c8 == 0) ssq[wave][8 * j + r8] = v;                          // slot 8j + r8: rows 0..15 = A, 16..47 = B
  }
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
    const float qa = ((ssq[0][irow[q]] + ssq[1][irow[q]]) + ssq[2][irow[q]]) + ssq[3][irow[q]];
    const float qb = ((ssq[0][16 + col[q]] + ssq[1][16 + col[q]]) + ssq[2][16 + col[q]]) + ssq[3][16 + col[q]];
    const float s = u / (fmaxf(sqrtf(qa), 1e-12f) * fmaxf(sqrtf(qb), 1e-12f));     // F.normalize's eps (models.py:24-25)
    const int gr = r0 + irow[q], gc = c0 + col[q];
    const bool ok = gr < N2 && gc < N2;
    if (ok) a.S[(long long)gr * N2 + gc] = s;
    e[q] = (ok && gr != gc) ? expf(s / a.T) : 0.f;                                  // negatives_mask (models.py:31-33)
  }
  if (blockIdx.x == 0 && tid < 16 && r0 + tid < N2)
    a.nrm[r0 + tid] = fmaxf(sqrtf(((ssq[0][tid] + ssq[1][tid]) + ssq[2][tid]) + ssq[3][tid]), 1e-12f);
  __syncthreads();                               // all reads of `red` are done
  float* ew = lds;                               // [16][33]
#pragma unroll
  for (int q = 0; q < 2; ++q) ew[irow[q] * 33 + col[q]] = e[q];
  __syncthreads();
  if (tid < 16 && r0 + tid < N2) {               // this tile's share of the row's denominator, columns in index order
    float sum = 0.f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) sum += ew[tid * 33 + c];
    a.rs_part[(long long)blockIdx.x * N2 + r0 + tid] = sum;
  }
}

constexpr int NTX_R = 8;             // rows per workgroup of the gradient kernel

__global__ __launch_bounds__(256) void ntx_grad_kernel(NtxArgs a, int CT) {
  extern __shared__ __attribute__((aligned(16))) float sm[];         // iden[N2] | W[NTX_R][N2] | red[4][16] | inrm[N2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N2 = 2 * a.B, D = a.D, B = a.B;
  const int r0 = blockIdx.x * NTX_R, d = blockIdx.y * 256 + tid;
  float* iden = sm;
  float* W = sm + N2;
  float* red = W + NTX_R * N2;
  float* inrm = red + 64;                                            // 1 / |x_b| of every row
  const float invT = 1.f / a.T, sc = 1.f / (a.T * (float)N2);
  // denominators of all rows (column tiles summed in index order), and the loss (workgroup (0, 0))
  float lsum = 0.f;
  for (int b = tid; b < N2; b += 256) {
    float den = 0.f;
    for (int c0 = 0; c0 < CT; c0 += 8) {               // eight partials requested together, added in index order
      float pv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) pv[q] = a.rs_part[(long long)(c0 + q < CT ? c0 + q : CT - 1) * N2 + b];
#pragma unroll
      for (int q = 0; q < 8; ++q) den += (c0 + q < CT) ? pv[q] : 0.f;
    }
    iden[b] = 1.f / den;
    inrm[b] = 1.f / a.nrm[b];
    if (blockIdx.x == 0 && blockIdx.y == 0) {
      const int p = (b < B) ? b + B : b - B;
      lsum += logf(den) - a.S[(long long)b * N2 + p] * invT;          // -log(exp(S_ap / T) / den_a)
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
  }
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)N2;
  // this workgroup's rows of W = G + G^T, and z_a . dz_a = sum_b W[a][b] S[a][b]
  float dotp[NTX_R];
#pragma unroll
  for (int i = 0; i < NTX_R; ++i) dotp[i] = 0.f;
  for (int b = tid; b < N2; b += 256) {
    const float idb = iden[b];
    float sv[NTX_R];                                     // the eight similarities first (unconditional loads: behind a
#pragma unroll                                           // branch each one waited out its own round trip)
    for (int i = 0; i < NTX_R; ++i) sv[i] = a.S[(long long)(r0 + i < N2 ? r0 + i : N2 - 1) * N2 + b];
#pragma unroll
    for (int i = 0; i < NTX_R; ++i) {
      const int ra = r0 + i;
      const int rc = ra < N2 ? ra : N2 - 1;
      const int p = (ra < B) ? ra + B : ra - B;
      const float wv = (expf(sv[i] * invT) * (iden[rc] + idb) - (b == p ? 2.f : 0.f)) * sc;
      const float w = (ra < N2 && b != ra) ? wv : 0.f;
      dotp[i] = fmaf(w, sv[i], dotp[i]);
      W[i * N2 + b] = w;
    }
  }
  __syncthreads();                               // (also: red[] has been read)
#pragma unroll
  for (int i = 0; i < NTX_R; ++i) {
    const float v = wave_sum(dotp[i]);
    if (lane == 0) red[wave * 16 + i] = v;
  }
  __syncthreads();
  // dz[a][d] = sum_b W[a][b] x[b][d] / |x_b|: the embeddings stream through once, eight rows in flight per thread
  float acc[NTX_R];
#pragma unroll
  for (int i = 0; i < NTX_R; ++i) acc[i] = 0.f;
  const bool dv = d < D;
  // sixteen rows per trip, the NEXT trip's sixteen loads requested before this trip's arithmetic (the first version
  // loaded eight rows, used them, loaded the next eight: one memory round trip per eight rows, 50 us at 2B = 256 and
  // 260 us at 2B = 1024 for a few MFLOP)
  constexpr int NB = 16;
  const int dc = dv ? d : 0;
  // the rows are two plain arrays (emb_i, emb_j): each is walked with a running offset (the generic row lookup -- a
  // 64-bit select per load -- cost more instructions than the arithmetic it fed)
  auto walk = [&](const float* __restrict__ base, int nrows, int col0) {
    const float* p = base + dc;
    float cur[NB], nxt[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) cur[q] = p[(size_t)(q < nrows ? q : nrows - 1) * D];
    for (int b0 = 0; b0 < nrows; b0 += NB) {
#pragma unroll
      for (int q = 0; q < NB; ++q) { const int b = b0 + NB + q; nxt[q] = p[(size_t)(b < nrows ? b : nrows - 1) * D]; }
      const float* wrow = W + col0 + b0;
      const float* nrow = inrm + col0 + b0;
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const bool ok = b0 + q < nrows;
        const float xv = ok ? cur[q] * nrow[ok ? q : 0] : 0.f;
#pragma unroll
        for (int i = 0; i < NTX_R; ++i) acc[i] = fmaf(wrow[i * N2 + (ok ? q : 0)], xv, acc[i]);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) cur[q] = nxt[q];
    }
  };
  walk(a.ei, B, 0);
  walk(a.ej, B, B);
  if (!dv) return;
#pragma unroll
  for (int i = 0; i < NTX_R; ++i) {
    const int ra = r0 + i;
    if (ra < N2) {
      const float n = a.nrm[ra], z = ntx_row(a, ra)[d] * inrm[ra];
      const float dot = (red[i] + red[16 + i]) + (red[32 + i] + red[48 + i]);
      float* g = (ra < B) ? a.gi + (long long)ra * D : a.gj + (long long)(ra - B) * D;
      g[d] = (acc[i] - z * dot) / n;             // backward of x / max(|x|, eps) (for |x| above eps)
    }
  }
}

// The gradient on the MFMA (D a multiple of 4, 16-byte aligned rows: the case that matters; ntx_grad_kernel above keeps
// the rest).  One workgroup per (16 rows a, 64 embedding columns d) -- 256 of them at 2B = 256, D = 1024 where the
// vector version had 128 and spent 33 us streaming rows through 8 x 256 fused multiply-adds per thread.  The prologue is
// the same (denominators of all rows, this workgroup's rows of W = G + G^T, z_a . dz_a), but W goes to LDS TRANSPOSED and
// already divided by |x_b| -- Wt[b][a] = W[a][b] / |x_b| is the A operand of dz = Wt^T . x, complete before the first
// multiply -- and only the embedding rows are staged: chunks of 64 rows x 64 columns in 16-byte pieces, two chunks in
// flight in registers, double-buffered in LDS, one barrier per chunk; wave w owns columns 16 w .. 16 w + 15 on
// v_mfma_f32_16x16x4_f32.
// NCW = 16-column tiles per wave (a workgroup takes 64 NCW embedding columns), KC = embedding rows per staged chunk.  The
// prologue (denominators of ALL rows, the workgroup's sixteen rows of W) is the same for every column slice of a row
// group, so at large batches the slices are made wide: B = 512 with NCW = 1 ran 64 x 16 workgroups, every one folding
// 32 k denominator partials and forming 16 k exponentials for 2 MFLOP of product, four rounds of one workgroup per CU
// (126 us); NCW = 4 is ONE round of 256 workgroups, W formed four times instead of sixteen (launch_ntxent picks the
// widest slice that still leaves a workgroup per CU).  (Also built and measured, round 5: 64 x 64 tiles in both launches --
// ntx_sim64 / ntx_grad64, W formed chunk by chunk beside the embedding rows, a quarter of the L2 traffic -- 44.9 + 74.7 us
// at 2B = 1024 against 43.1 + 65.4: one four-wave workgroup per CU exposes every stall of its chunk loop; removed.)
constexpr int NTG_R = 16;                                  // rows per workgroup
__host__ __device__ constexpr int ntg_bs(int NCW) { return 64 * NCW + 16; }     // LDS row stride of a chunk

// NW = waves: 8 where the slice is 256 columns wide (ONE workgroup per CU: two waves per SIMD, two column tiles each)
template <int NCW, int KC, int NW = 4>
__global__ __launch_bounds__(64 * NW) void ntx_grad_mfma_kernel(NtxArgs a, int CT) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // iden[N2] | inrm[N2] | red[128] | Wt[N2p][16] | xs[2][KC][BS]
  constexpr int BS = ntg_bs(NCW), WC = 64 * NCW;               // columns per workgroup
  constexpr int NT = 64 * NW, TW = 4 * NCW / NW;              // threads; 16-column tiles per wave
  constexpr int NP = KC * (WC / 4) / NT;                      // 16-byte pieces per thread and chunk
  constexpr int RSTEP = NT / (WC / 4);                        // chunk rows between a thread's pieces
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N2 = 2 * a.B, D = a.D, B = a.B;
  const int N2p = ((N2 + 63) / 64) * 64;                          // rows of Wt, padded to whole chunks of either size (zeros)
  const int r0 = blockIdx.x * NTG_R, d0 = blockIdx.y * WC;
  float* iden = sm;
  float* inrm = iden + N2;
  float* red = inrm + N2;
  float* Wt = sm + (((2 * N2 + 128) + 3) & ~3);                       // [N2p][16], 16-byte aligned
  float* xs = Wt + (size_t)N2p * NTG_R;                             // [2][KC][BS]
  const float invT = 1.f / a.T, sc = 1.f / (a.T * (float)N2);
  // ---- the first two chunks of embedding rows are requested before anything else
  const int rb = tid / (WC / 4), cb = (tid % (WC / 4)) * 4;         // piece q: row rb + RSTEP q of the chunk, floats cb .. cb + 3
  const bool b_ok = d0 + cb < D;
  auto fetch = [&](int k0, float4 (&v)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int r = k0 + rb + RSTEP * q;
      const float4 y = *(const float4*)(ntx_row(a, r < N2 ? r : 0) + (b_ok ? d0 + cb : 0));
      v[q] = (r < N2 && b_ok) ? y : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&](int buf, const float4 (&v)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
      *(float4*)(xs + (size_t)buf * KC * BS + (rb + RSTEP * q) * BS + cb) = make_float4(v[q].x, v[q].y, v[q].z, v[q].w);
  };
  const int NC = N2p / KC;
  float4 v0[NP], v1[NP];
  fetch(0, v0);
  if (NC > 1) fetch(KC, v1);
  // ---- denominators of all rows (column tiles summed in index order), and the loss (workgroup (0, 0))
  float lsum = 0.f;
  for (int b = tid; b < N2; b += NT) {
    float den = 0.f;
    for (int c0 = 0; c0 < CT; c0 += 32) {              // thirty-two partials requested together, added in index order
      float pv[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) pv[q] = a.rs_part[(long long)(c0 + q < CT ? c0 + q : CT - 1) * N2 + b];
#pragma unroll
      for (int q = 0; q < 32; ++q) den += (c0 + q < CT) ? pv[q] : 0.f;
    }
    iden[b] = 1.f / den;
    inrm[b] = 1.f / a.nrm[b];
    if (blockIdx.x == 0 && blockIdx.y == 0) {
      const int p = (b < B) ? b + B : b - B;
      lsum += logf(den) - a.S[(long long)b * N2 + p] * invT;          // -log(exp(S_ap / T) / den_a)
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
  }
  // the sixteen similarities of this thread's first column b (unconditional loads, in flight across the barrier)
  auto load_s = [&](int b, float (&sv)[NTG_R]) {
#pragma unroll
    for (int i = 0; i < NTG_R; ++i) sv[i] = a.S[(long long)(r0 + i < N2 ? r0 + i : N2 - 1) * N2 + (b < N2 ? b : 0)];
  };
  float sv[NTG_R];
  load_s(tid, sv);
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    float l = (red[0] + red[1]) + (red[2] + red[3]);
    if (NW == 8) l += (red[4] + red[5]) + (red[6] + red[7]);
    a.loss[0] = l / (float)N2;
  }
  // ---- this workgroup's rows of W = G + G^T (transposed, scaled by 1 / |x_b|), and z_a . dz_a = sum_b W[a][b] S[a][b]
  float dotp[NTG_R];
#pragma unroll
  for (int i = 0; i < NTG_R; ++i) dotp[i] = 0.f;
  for (int b = tid; b < N2p; b += NT) {
    const bool bv = b < N2;
    const float idb = bv ? iden[b] : 0.f, inb = bv ? inrm[b] : 0.f;
    float sn[NTG_R];                                     // the next column's similarities while this one's exponentials run
    if (b + NT < N2p) load_s(b + NT, sn);
    float wr[NTG_R];
#pragma unroll
    for (int i = 0; i < NTG_R; ++i) {
      const int ra = r0 + i;
      const int rc = ra < N2 ? ra : N2 - 1;
      const int p = (ra < B) ? ra + B : ra - B;
      const float wv = (expf(sv[i] * invT) * (iden[rc] + idb) - (b == p ? 2.f : 0.f)) * sc;
      const float w = (bv && ra < N2 && b != ra) ? wv : 0.f;
      dotp[i] = fmaf(w, sv[i], dotp[i]);
      wr[i] = w * inb;
    }
#pragma unroll
    for (int i = 0; i < NTG_R; i += 4) *(float4*)(Wt + (size_t)b * NTG_R + i) = make_float4(wr[i], wr[i + 1], wr[i + 2], wr[i + 3]);
    if (b + NT < N2p) {
#pragma unroll
      for (int i = 0; i < NTG_R; ++i) sv[i] = sn[i];
    }
  }
  __syncthreads();                               // (also: red[] has been read)
#pragma unroll
  for (int i = 0; i < NTG_R; ++i) {
    const float v = wave_sum(dotp[i]);
    if (lane == 0) red[wave * 16 + i] = v;
  }
  // ---- dz[a][d] = sum_b Wt[b][a] x[b][d]: wave w owns column tiles TW w .. TW w + TW - 1 of the slice
  f32x4 acc[TW], acc2[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) { acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  stage(0, v0);
  if (NC > 2) fetch(2 * KC, v0);
  for (int c = 0; c < NC; ++c) {
    __syncthreads();                             // chunk c staged (and, c = 0: Wt, red complete); the other buffer is free
    if (c + 1 < NC) {
      if ((c & 1) == 0) { stage(1, v1); if (c + 3 < NC) fetch((c + 3) * KC, v1); }
      else              { stage(0, v0); if (c + 3 < NC) fetch((c + 3) * KC, v0); }
    }
    const float* as = Wt + (size_t)(c * KC + kq) * NTG_R + l16;
    const float* bs = xs + (size_t)(c & 1) * KC * BS + kq * BS + wave * (16 * TW) + l16;
    float av[KC / 4];
#pragma unroll
    for (int k4 = 0; k4 < KC / 4; ++k4) av[k4] = as[(4 * k4) * NTG_R];
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      float bv[KC / 4];
#pragma unroll
      for (int k4 = 0; k4 < KC / 4; ++k4) bv[k4] = bs[(4 * k4) * BS + 16 * j];
#pragma unroll
      for (int k4 = 0; k4 < KC / 4; k4 += 2) {
        acc[j]  = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k4], bv[k4], acc[j], 0, 0, 0);
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k4 + 1], bv[k4 + 1], acc2[j], 0, 0, 0);
      }
    }
  }
  // ---- backward of x / max(|x|, eps): D register r of lane l is row 4 (l >> 4) + r, column l & 15
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    const int d = d0 + wave * (16 * TW) + 16 * j + l16;
    if (d < D) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * kq + r, ra = r0 + i;
        if (ra < N2) {
          const float n = a.nrm[ra], z = ntx_row(a, ra)[d] * inrm[ra];
          float dot = (red[i] + red[16 + i]) + (red[32 + i] + red[48 + i]);
          if (NW == 8) dot += (red[64 + i] + red[80 + i]) + (red[96 + i] + red[112 + i]);
          float* g = (ra < B) ? a.gi + (long long)ra * D : a.gj + (long long)(ra - B) * D;
          g[d] = ((acc[j][r] + acc2[j][r]) - z * dot) / n;
        }
      }
    }
  }
}

size_t ntxent_ws_floats(int B, int D) {
  (void)D;
  const size_t N2 = 2 * (size_t)B, CT = (N2 + 31) / 32;
  return N2 * N2 + CT * N2 + N2 + 64;
}

hipError_t launch_ntxent(const float* ei, const float* ej, int B, int D, float T, float* loss, float* gi, float* gj,
                         float* ws, hipStream_t st) {
  const size_t N2 = 2 * (size_t)B;
  const int CT = (int)((N2 + 31) / 32);
  NtxArgs a;
  a.ei = ei; a.ej = ej; a.B = B; a.D = D; a.T = T; a.loss = loss; a.gi = gi; a.gj = gj;
  a.S = ws; ws += N2 * N2;
  a.rs_part = ws; ws += (size_t)CT * N2;
  a.nrm = ws;
  const bool aligned = D % 4 == 0 && (((uintptr_t)ei | (uintptr_t)ej) & 15) == 0;
  hipLaunchKernelGGL(ntx_sim_kernel, dim3(CT, (unsigned)((N2 + 15) / 16)), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // the MFMA gradient wherever the rows are whole 16-byte pieces (CMLPL_NTX_MFMA=0: the vector kernel always); the widest
  // column slice (64 NCW columns per workgroup) that still gives every CU a workgroup and fits LDS
  const size_t N2p = ((N2 + 63) / 64) * 64;
  if (aligned && switches().ntx_mfma != 0) {
    const unsigned rg = (unsigned)((N2 + NTG_R - 1) / NTG_R);
    auto lds_of = [&](int ncw, int kc) { return ((((2 * N2 + 128) + 3) & ~(size_t)3) + N2p * NTG_R + 2 * (size_t)kc * ntg_bs(ncw)) * 4; };
    auto wgs_of = [&](int ncw) { return (size_t)rg * (size_t)((D + 64 * ncw - 1) / (64 * ncw)); };
    const int force = switches().ntx_ncw;
    int ncw = 1;
    if (force == 2 || force == 4) ncw = force;
    else if (force == 0) {
      if (wgs_of(2) >= 256 && lds_of(2, 64) <= LDS_MAX) ncw = 2;
      if (wgs_of(4) >= 256 && lds_of(4, 32) <= LDS_MAX) ncw = 4;
    }
    const size_t lds_m = lds_of(ncw, ncw == 4 ? 32 : 64);
    if (lds_m <= LDS_MAX) {
      const dim3 grid(rg, (unsigned)((D + 64 * ncw - 1) / (64 * ncw)));
      if (ncw == 4) {
        static DevOnce once;
        if ((e = ensure_max_lds(once, ntx_grad_mfma_kernel<4, 32, 8>)) != hipSuccess) return e;
        hipLaunchKernelGGL((ntx_grad_mfma_kernel<4, 32, 8>), grid, dim3(512), lds_m, st, a, CT);
      } else if (ncw == 2) {
        static DevOnce once;
        if ((e = ensure_max_lds(once, ntx_grad_mfma_kernel<2, 64>)) != hipSuccess) return e;
        hipLaunchKernelGGL((ntx_grad_mfma_kernel<2, 64>), grid, dim3(256), lds_m, st, a, CT);
      } else {
        static DevOnce once;
        if ((e = ensure_max_lds(once, ntx_grad_mfma_kernel<1, 64>)) != hipSuccess) return e;
        hipLaunchKernelGGL((ntx_grad_mfma_kernel<1, 64>), grid, dim3(256), lds_m, st, a, CT);
      }
      return hipGetLastError();
    }
  }
  const size_t lds = (2 * N2 + (size_t)NTX_R * N2 + 64) * 4;
  if (lds > 64 * 1024) return hipErrorInvalidValue;        // 2B <= 1600 rows
  hipLaunchKernelGGL(ntx_grad_kernel, dim3((unsigned)((N2 + NTX_R - 1) / NTX_R), (unsigned)((D + 255) / 256)), dim3(256),
                     lds, st, a, CT);
  return hipGetLastError();
}

}  // namespace cmlpl
