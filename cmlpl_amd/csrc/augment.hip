// Input augmentation + batch concat (train.py:157-158,163-164,170-174,181-184):
//   xn[net] = cat(XPl, XPu) + sigma*N(0,1)      sn[net] = cat(Xl, Xu) + sigma*N(0,1)
// The reference draws the noise on the CPU generator and copies it over PCIe every step; here it is
// Philox4x32-10 + Box-Muller in registers (or explicit noise tensors in parity mode).  The Philox
// counter is (4-element group within the sample, GLOBAL sample index, stream, step): a sample gets
// the same noise no matter how the batch is sharded over GPUs.
// HBM-bound elementwise kernel: reads each source once, writes one noisy copy per network.
//
// dist_unpack_kernel: re-orders gathered per-rank [feat | labels] and [logits] blocks into the
// global row order [labelled of all ranks ; unlabelled of all ranks] the loss kernels expect.
#ifndef CMLPL_ABL
#define CMLPL_ABL 0
#endif

#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

struct AugArgs {
  const float* srcl[2]; const float* srcu[2];     // [0] = XP, [1] = X
  const float* noise[8];                          // reference draw order, or all null
  float* dst[2];                                  // xn, sn
  float* snT;                                     // optional [nets][bands][n] copy of sn (k-major for feat_spe)
  int per[2];                                     // elements per sample: C*H*W, bands
  int bt, btu, lab0, unl_base;                    // local rows and their global sample indices
  float sigma; int nets; int explicit_noise; uint64_t seed, step;
  int t0;                                         // first tensor handled by this launch (blockIdx.z = 0)
  const long long* labels; float* labels_f;       // optional: labels as float, for the packed exchange buffer
  RowSel sel;                                     // batches by index / device-side step scalars (common.hpp)
};

__global__ void augment_kernel(AugArgs a) {
  const int t = a.t0 + blockIdx.z;            // 0: XP, 1: X
  const int per = a.per[t], s = blockIdx.y;   // local sample
  // a block covers 1024 consecutive elements of one sample = 256 groups of 4; thread t takes group t (one Philox
  // block = the noise of elements 4g..4g+3: the fused forward / data-gradient kernels use the same counter, so the
  // augmented values are identical whether this kernel or they form them).  The source is read once and written
  // once per network, each network with its own Philox stream.
  const int base = blockIdx.x * 1024 + 4 * threadIdx.x;
  if (a.labels_f != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < a.bt; i += 256) a.labels_f[i] = (float)a.labels[rowsel_index(a.sel, true, i)];
  if (blockIdx.x * 1024 >= per) return;
  const bool lab = s < a.bt;
  const int sl = lab ? s : s - a.bt;
  const float* src = (lab ? a.srcl[t] : a.srcu[t]) + rowsel_index(a.sel, lab, sl) * per;
  uint64_t rstep = a.step;
  const cmlpl_dyn* dynr = dyn_row(a.sel.dyn);
  if (dynr != nullptr) rstep = (uint64_t)uni64((long long)dynr->step);
  const bool need_noise = a.sigma != 0.f;
  const uint64_t gs = (uint64_t)(lab ? a.lab0 + sl : a.unl_base + sl);
  const long long n_all = a.bt + a.btu;
  if ((per & 3) == 0 && !(t == 1 && a.snT != nullptr)) {
    // rows of whole 16-byte groups (60 x 20 x 20, 48 x 15 x 15 patches): one 16-byte load and one 16-byte store per
    // network instead of four 4-byte ones -- the kernel is bound by its memory instructions, not by the generator
    // (20 x 20 x 60: 25 -> us per launch for 74 MB)
    if (base >= per) return;
    const float4 xv = *(const float4*)(src + base);
    // both networks' noise first, then both stores (nets <= 2, written out: a runtime loop over the networks made every
    // iteration wait for its own generator before its store went out)
    const bool two = a.nets > 1;
    float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f), z1 = z0;
    if (need_noise && a.explicit_noise) {
      z0 = *(const float4*)((lab ? a.noise[t] : a.noise[4 + t]) + (long long)sl * per + base);
      if (two) z1 = *(const float4*)((lab ? a.noise[2 + t] : a.noise[6 + t]) + (long long)sl * per + base);
    } else if (need_noise) {
      // (patches: the eight-normals-per-hash generator of the fused forward, one group of a pair here; spectra: four per hash)
      if (t == 0) {
        z0 = noise_normal4p(a.seed, rstep, STREAM_NOISE_XP, gs, (uint32_t)(base >> 2));
        if (two) z1 = noise_normal4p(a.seed, rstep, STREAM_NOISE_XP + 1, gs, (uint32_t)(base >> 2));
      } else {
        z0 = noise_normal4(a.seed, rstep, STREAM_NOISE_X, noise_ctr(gs, (uint32_t)(base >> 2)));
        if (two) z1 = noise_normal4(a.seed, rstep, STREAM_NOISE_X + 1, noise_ctr(gs, (uint32_t)(base >> 2)));
      }
    }
    float4 v0 = xv, v1 = xv;
    if (need_noise) {
      v0.x = fmaf(z0.x, a.sigma, xv.x); v0.y = fmaf(z0.y, a.sigma, xv.y); v0.z = fmaf(z0.z, a.sigma, xv.z); v0.w = fmaf(z0.w, a.sigma, xv.w);
      v1.x = fmaf(z1.x, a.sigma, xv.x); v1.y = fmaf(z1.y, a.sigma, xv.y); v1.z = fmaf(z1.z, a.sigma, xv.z); v1.w = fmaf(z1.w, a.sigma, xv.w);
    }
    float* d0 = a.dst[t] + (long long)s * per + base;
    *(float4*)d0 = v0;
    if (two) *(float4*)(d0 + n_all * per) = v1;
    return;
  }
  float x[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = base + q;
    x[q] = src[e < per ? e : per - 1];
  }
  for (int net = 0; net < a.nets; ++net) {
    float* dst = a.dst[t] + ((long long)net * n_all + s) * per;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (need_noise && !a.explicit_noise) {
      const float4 nz = (t == 0) ? noise_normal4p(a.seed, rstep, STREAM_NOISE_XP + net, gs, (uint32_t)(base >> 2))
                                 : noise_normal4(a.seed, rstep, STREAM_NOISE_X + net, noise_ctr(gs, (uint32_t)(base >> 2)));
      z[0] = nz.x; z[1] = nz.y; z[2] = nz.z; z[3] = nz.w;
    } else if (need_noise) {
      const float* nptr = (lab ? a.noise[2 * net + t] : a.noise[4 + 2 * net + t]) + (long long)sl * per;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = base + q;
        z[q] = nptr[e < per ? e : per - 1];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = base + q;
      if (e < per) {
        const float v = need_noise ? fmaf(z[q], a.sigma, x[q]) : x[q];
        dst[e] = v;
        if (t == 1 && a.snT != nullptr) a.snT[((long long)net * per + e) * n_all + s] = v;
      }
    }
  }
}

hipError_t launch_augment(int which, int nets, int bt, int btu, int per_xp, int per_x, int lab0, int unl_base,
                          const float* xpl, const float* xl, const float* xpu, const float* xu,
                          const float* const* noise8, float sigma, uint64_t seed, uint64_t step,
                          float* xn, float* sn, float* snT, hipStream_t st, const long long* labels, float* labels_f,
                          const RowSel* sel) {
  AugArgs a;
  a.sel = sel != nullptr ? *sel : RowSel();
  a.snT = snT; a.labels = labels; a.labels_f = (labels != nullptr) ? labels_f : nullptr;
  a.srcl[0] = xpl; a.srcl[1] = xl; a.srcu[0] = xpu; a.srcu[1] = xu;
  for (int i = 0; i < 8; ++i) a.noise[i] = noise8 ? noise8[i] : nullptr;
  a.dst[0] = xn; a.dst[1] = sn;
  a.per[0] = per_xp; a.per[1] = per_x; a.bt = bt; a.btu = btu; a.lab0 = lab0; a.unl_base = unl_base;
  a.sigma = sigma; a.nets = nets; a.explicit_noise = noise8 != nullptr; a.seed = seed; a.step = step;
  if (!(which & 3)) return hipSuccess;
  a.t0 = (which & 1) ? 0 : 1;
  const int nz = (which & 3) == 3 ? 2 : 1;
  const int mx = nz == 2 ? (per_xp > per_x ? per_xp : per_x) : ((which & 1) ? per_xp : per_x);
  dim3 grid((mx + 1023) / 1024, bt + btu, nz);
  hipLaunchKernelGGL(augment_kernel, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// recv_f: [W][ 2*n_l*1024 feat | bt_l labels-as-float ], recv_z: [W][2*n_l*K logits], n_l = bt_l + btu_l
__global__ void dist_unpack_kernel(const float* __restrict__ recv_f, const float* __restrict__ recv_z, int W, int bt_l,
                                   int btu_l, int K, float* __restrict__ logits_g, float* __restrict__ feat_g,
                                   long long* __restrict__ labels_g) {
  const int n_l = bt_l + btu_l, bt_g = W * bt_l, n_g = W * n_l;
  const long long pack_f = 2LL * n_l * FD + bt_l, pack_z = 2LL * n_l * K;
  const long long nfeat4 = 2LL * n_g * (FD / 4), nlog = 2LL * n_g * K;
  const long long total = nfeat4 + nlog + bt_g;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (i < nfeat4) {
      const int c4 = (int)(i % (FD / 4));
      const long long row = i / (FD / 4);
      const int net = (int)(row / n_g), g = (int)(row - (long long)net * n_g);
      const int rank = (g < bt_g) ? g / bt_l : (g - bt_g) / btu_l;
      const int loc = (g < bt_g) ? g - rank * bt_l : bt_l + (g - bt_g) - rank * btu_l;
      const float* src = recv_f + rank * pack_f + ((long long)net * n_l + loc) * FD;
      ((float4*)feat_g)[i] = ((const float4*)src)[c4];
    } else if (i < nfeat4 + nlog) {
      const long long j = i - nfeat4;
      const int k = (int)(j % K);
      const long long row = j / K;
      const int net = (int)(row / n_g), g = (int)(row - (long long)net * n_g);
      const int rank = (g < bt_g) ? g / bt_l : (g - bt_g) / btu_l;
      const int loc = (g < bt_g) ? g - rank * bt_l : bt_l + (g - bt_g) - rank * btu_l;
      logits_g[j] = recv_z[rank * pack_z + ((long long)net * n_l + loc) * K + k];
    } else {
      const int g = (int)(i - nfeat4 - nlog);
      const int rank = g / bt_l, loc = g - rank * bt_l;
      labels_g[g] = (long long)(recv_f[rank * pack_f + 2LL * n_l * FD + loc] + 0.5f);
    }
  }
}

hipError_t launch_dist_unpack(const float* recv_f, const float* recv_z, int W, int bt_l, int btu_l, int K, float* logits_g,
                              float* feat_g, long long* labels_g, hipStream_t st) {
  hipLaunchKernelGGL(dist_unpack_kernel, dim3(512), dim3(256), 0, st, recv_f, recv_z, W, bt_l, btu_l, K, logits_g, feat_g,
                     labels_g);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Patch extraction on device (reference tools/hyper_tools.py:35-55 MirrowCut + :226-243 ExtractPatches):
// out[p][ch][i][j] = cube[mirror(r+i-hw)][mirror(c+j-hw)][ch] for pixel k = idx[p] = r*cols + c, hw = w/2,
// symmetric (edge-repeating) mirror.  Pure gather, exact.  One workgroup per patch; the w*w x C tile goes
// through LDS so that the cube is read along its contiguous channel axis and the band-major patch is
// written along its contiguous pixel axis.
//   Gather.  A patch whose columns need no mirroring (all but a 2*hw-wide margin of the image) reads w SPANS of w*C
// consecutive floats, one per patch row (row mirroring only picks another source row): 16-byte loads, each thread's
// eight in flight together, i.e. the whole patch in one round trip.  (Round 3 measurements: stores alone run at
// 5.1 TB/s; pixel by pixel with 4-byte loads -- two wave-instructions per 412-byte pixel, sixteen in flight per wave --
// the gather alone took 149 of the launch's 167 us, however the patches were ordered.)  Patches on the column margin
// keep the pixel-by-pixel path.
//   LDS layout.  C odd: tile[i * RS + j * C + ch], RS = w*C rounded up to 4 floats, so that a span lands with aligned
// 16-byte LDS writes; the scatter's column reads (lanes = consecutive pixels) see stride C (odd) inside a row.  C even:
// tile[(i*w + j) * (C+1) + ch], the padded layout (odd pixel stride); span elements are placed one by one.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void extract_patches_kernel(const float* __restrict__ cube, int rows, int cols, int C,
                                                              int w, const long long* __restrict__ idx, int n,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hw = w >> 1, ww = w * w;
  const bool codd = (C & 1) != 0;
  const int NF = w * C;                               // floats of one patch row (span)
  const int RS = codd ? ((NF + 3) & ~3) : w * (C + 1);   // LDS floats per patch row
  const int CS = codd ? C : C + 1;                    // LDS floats per pixel
  const int n4 = NF >> 2, NI = w * n4, rem = NF & 3;  // 16-byte items per row / per patch; floats beyond them per row
  const float inv4 = 1.0f / (float)(n4 > 0 ? n4 : 1), invC = 1.0f / (float)C;
  constexpr int GB = 8;                               // 16-byte items per thread in one batch (one batch covers w*C <= 16 K floats)
  // (Gather alone and scatter alone take 84 and 80 us for the 8192-patch case -- each 408 MB at ~5 TB/s -- and the
  // launch 139 us: reads + writes together are at the HBM rate.  Persistent workgroups that keep the NEXT patch's loads in
  // flight across this patch's stores were slower, 149 us: two instead of three workgroups per CU for the registers.)
  float4 v[GB];
  float vt = 0.f;
  int r = 0, c = 0;
  bool interior = false;
  // fetch(p): request patch p's spans (first batch and the row tails); margin patches are gathered in put()
  auto fetch = [&](int p) {
    const long long k = idx[p];
    r = (int)(k / cols); c = (int)(k - (long long)r * cols);
    interior = (c - hw >= 0) && (c - hw + w <= cols) && CMLPL_ABL != 50;      // workgroup-uniform
    if (!interior) return;
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int t = tid + 512 * q, tc = t < NI ? t : 0;
      // row of the item: exact for t < 2^22 / n4 (the quotient is at least 0.5 / n4 away from an integer)
      const int i = (int)(((float)tc + 0.5f) * inv4), e4 = tc - i * n4;
      int rr = r + i - hw;
      rr = rr < 0 ? -rr - 1 : (rr >= rows ? 2 * rows - 1 - rr : rr);
      v[q] = *(const float4*)(cube + ((long long)rr * cols + (c - hw)) * C + 4 * e4);
    }
    if (tid < w * rem) {
      const int i = tid / rem, e = (NF & ~3) + (tid - i * rem);
      int rr = r + i - hw;
      rr = rr < 0 ? -rr - 1 : (rr >= rows ? 2 * rows - 1 - rr : rr);
      vt = cube[((long long)rr * cols + (c - hw)) * C + e];
    }
  };
  auto put_item = [&](int i, int e0, const float4& x4) {
    if (codd) {
      *(float4*)(tile + i * RS + e0) = x4;
    } else {
      int j = (int)(((float)e0 + 0.5f) * invC), ch = e0 - j * C;
      const float x[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tile[i * RS + j * CS + ch] = x[e];
        if (++ch == C) { ch = 0; ++j; }
      }
    }
  };
  // put(): the fetched patch -> LDS tile
  auto put = [&]() {
    if (interior) {
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int t = tid + 512 * q;
        if (t < NI) { const int i = (int)(((float)t + 0.5f) * inv4); put_item(i, 4 * (t - i * n4), v[q]); }
      }
      for (int t = 512 * GB + tid; t < NI; t += 512) {            // rows beyond one batch (w*C > 16 K floats): not pipelined
        const int i = (int)(((float)t + 0.5f) * inv4), e4 = t - i * n4;
        int rr = r + i - hw;
        rr = rr < 0 ? -rr - 1 : (rr >= rows ? 2 * rows - 1 - rr : rr);
        put_item(i, 4 * e4, *(const float4*)(cube + ((long long)rr * cols + (c - hw)) * C + 4 * e4));
      }
      if (tid < w * rem) {
        const int i = tid / rem, e = (NF & ~3) + (tid - i * rem);
        const int j = e / C, ch = e - j * C;
        tile[i * RS + j * CS + ch] = vt;
      }
    } else if (CMLPL_ABL != 50) {
      // column margin: a wave takes whole pixels (its (i, j), the mirrored source pixel and its address are
      // wave-uniform), lanes = channels, eight pixels of a wave in flight at once
      constexpr int PB = 8;
      for (int pix0 = wave; pix0 < ww; pix0 += 8 * PB) {
        for (int ch0 = 0; ch0 < C; ch0 += 128) {
          float x[PB][2];
#pragma unroll
          for (int q = 0; q < PB; ++q) {
            const int pix = pix0 + 8 * q, pc = pix < ww ? pix : 0;                      // wave-uniform
            const int i = pc / w, j = pc - i * w;
            int rr = r + i - hw, cc = c + j - hw;
            rr = rr < 0 ? -rr - 1 : (rr >= rows ? 2 * rows - 1 - rr : rr);
            cc = cc < 0 ? -cc - 1 : (cc >= cols ? 2 * cols - 1 - cc : cc);
            const float* src = cube + ((long long)rr * cols + cc) * C;
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int ch = ch0 + 64 * h + lane; x[q][h] = src[ch < C ? ch : 0]; }
          }
#pragma unroll
          for (int q = 0; q < PB; ++q) {
            const int pix = pix0 + 8 * q;
            if (pix < ww) {
              const int i = pix / w, j = pix - i * w;
#pragma unroll
              for (int h = 0; h < 2; ++h) { const int ch = ch0 + 64 * h + lane; if (ch < C) tile[i * RS + j * CS + ch] = x[q][h]; }
            }
          }
        }
      }
    }
  };
  // Workgroups go to the 8 XCDs round-robin in launch order and each XCD has its own 4 MB L2: workgroup b takes patch
  // (b % 8) * ceil(n / 8) + b / 8, so an XCD walks ONE contiguous eighth of the list and, when neighbouring list entries
  // are neighbouring pixels (raster order: whole-image inference; sorted index lists), the overlapping windows of its
  // ~100 resident patches are re-read from its L2 -- provided the output does not wash the cube out of that L2, hence
  // the non-temporal stores below.  8192 patches: random order 134 -> 129 us, sorted 140 -> 96-106 us (3.9-4.3 TB/s
  // written); either change alone does nothing for the sorted list.  (Ordering a random list by image row in a
  // preceding one-workgroup counting sort cost 13 us and won back 5: row order alone is not raster order.  Not kept.)
  const int chunk_ = (n + 7) >> 3;
  const int p = (int)(blockIdx.x & 7) * chunk_ + (int)(blockIdx.x >> 3);
  if (p >= n || (int)(blockIdx.x >> 3) >= chunk_) return;
  fetch(p);
  put();
  __syncthreads();
  // scatter: a wave takes whole bands, lanes = pixels: the band-major patch is written along its contiguous pixel
  // axis (256 consecutive bytes per wave-instruction)
  float* o = out + (long long)p * C * ww;
  for (int px0 = 0; px0 < ww; px0 += 256) {
    int po[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pix = px0 + 64 * q + lane, pc = pix < ww ? pix : 0, i = pc / w;
      po[q] = i * RS + (pc - i * w) * CS;
    }
    for (int ch = wave; ch < C; ch += 8) {
      float* orow = o + (long long)ch * ww;
      float x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = tile[po[q] + ch];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int pix = px0 + 64 * q + lane; if (pix < ww && (CMLPL_ABL != 51 || x[q] == 123.456f)) __builtin_nontemporal_store(x[q], orow + pix); }
    }
  }
}

static size_t extract_lds(int C, int w) {
  const size_t rs = (C & 1) ? (size_t)((w * C + 3) & ~3) : (size_t)w * (C + 1);
  return (size_t)w * rs * 4;
}

hipError_t launch_extract_patches(const float* cube, int rows, int cols, int C, int w, const long long* idx, int n,
                                  float* out, hipStream_t st) {
  const size_t lds = extract_lds(C, w);
  if (lds > LDS_MAX) return hipErrorInvalidValue;
  static DevOnce attr_once;
  {
    hipError_t e = ensure_max_lds(attr_once, extract_patches_kernel);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(extract_patches_kernel, dim3(8 * ((n + 7) / 8)), dim3(512), lds, st, cube, rows, cols, C, w, idx, n, out);
  return hipGetLastError();
}

}  // namespace cmlpl
