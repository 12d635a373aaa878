// loss_helper.py of the reference on gfx950:
//   dequeue_and_enqueue          (:19-36)    per-class FIFO memory bank
//   compute_contra_memobank_loss (:39-219)   class-wise anchor / negative selection + cosine InfoNCE
//   compute_unsupervised_loss    (:242-261)  entropy-percentile-filtered cross entropy
// All of it is gather / reduction work: bound by HBM/L2 latency, no MFMA.  Everything is deterministic
// (ordered compaction, fixed-order sums, no float atomics).
//
// Memory bank of one class: a physical ring [cap][D] with (rows, head); LOGICAL row i -- what the reference's
// `queue[0][i]` holds after its cat + [-size:] -- is ring slot (head + i) % cap.  The reference copies the whole
// bank on every call; the ring writes only the new keys.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace cmlpl {

// ------------------------------------------------------------------------------------------
// per-class selection (loss_helper.py:67-123): ordered index lists
//   list 0: low_valid rows            (prototype mean, :102-106; count decides "valid class", :135)
//   list 1: low-entropy anchor pool   (prob > 0.3 and low_valid, :95,99)
//   list 2: negative keys             (prob < 1, high_valid, rank window, :96,110-123)
// One workgroup per class walks the rows in order; positions come from ballot prefix sums.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mb_select_kernel(const float* __restrict__ prob, const float* __restrict__ label,
                                                        const float* __restrict__ low_mask,
                                                        const float* __restrict__ high_mask, int N, int Nl, int K,
                                                        float delta_p, float delta_n, int low_rank, int high_rank,
                                                        int* __restrict__ lists, int* __restrict__ counts) {
  __shared__ int wtot[3][4];
  __shared__ int base[3];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 3) base[tid] = 0;
  __syncthreads();
  int* l0 = lists + ((size_t)c * 3 + 0) * N;
  int* l1 = lists + ((size_t)c * 3 + 1) * N;
  int* l2 = lists + ((size_t)c * 3 + 2) * N;
  for (int n0 = 0; n0 < N; n0 += 256) {
    const int n = n0 + tid;
    bool f0 = false, f1 = false, f2 = false;
    if (n < N) {
      const float* pr = prob + (size_t)n * K;
      const float pc = pr[c], lab = label[(size_t)n * K + c];
      f0 = (lab * low_mask[n]) != 0.f;
      const bool hv = (lab * high_mask[n]) != 0.f;
      f1 = (pc > delta_p) && f0;
      int rank = 0;                                   // position of class c in the descending sort of the row
      for (int j = 0; j < K; ++j) rank += (pr[j] > pc || (pr[j] == pc && j < c)) ? 1 : 0;
      const bool cm = (n < Nl) ? (rank < low_rank && lab == 0.f) : (rank >= low_rank && rank < high_rank);
      f2 = (pc < delta_n) && hv && cm;
    }
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1), b2 = __ballot(f2);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) { wtot[0][wave] = __popcll(b0); wtot[1][wave] = __popcll(b1); wtot[2][wave] = __popcll(b2); }
    __syncthreads();
    int o0 = base[0], o1 = base[1], o2 = base[2];
    for (int w = 0; w < wave; ++w) { o0 += wtot[0][w]; o1 += wtot[1][w]; o2 += wtot[2][w]; }
    if (f0) l0[o0 + __popcll(b0 & below)] = n;
    if (f1) l1[o1 + __popcll(b1 & below)] = n;
    if (f2) l2[o2 + __popcll(b2 & below)] = n;
    __syncthreads();
    if (tid < 3) base[tid] += wtot[tid][0] + wtot[tid][1] + wtot[tid][2] + wtot[tid][3];
    __syncthreads();
  }
  if (tid < 3) counts[c * 3 + tid] = base[tid];
}

// prototype of every class: mean of the teacher features of its low_valid rows (NaN for an empty class, like
// torch.mean of an empty selection)
__global__ __launch_bounds__(256) void mb_proto_kernel(const float* __restrict__ rep_t, int D, const int* __restrict__ lists,
                                                       const int* __restrict__ counts, int N, float* __restrict__ proto) {
  const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const int cnt = counts[c * 3 + 0];
  const int* l0 = lists + ((size_t)c * 3 + 0) * N;
  float s = 0.f;
  for (int i = 0; i < cnt; ++i) s += rep_t[(size_t)l0[i] * D + d];
  proto[(size_t)c * D + d] = cnt > 0 ? s / (float)cnt : __builtin_nanf("");
}

// enqueue the negative keys of every class into its ring (only the last `cap` of them can survive)
__global__ __launch_bounds__(256) void mb_enqueue_kernel(const float* __restrict__ rep_t, int D, const int* __restrict__ lists,
                                                         const int* __restrict__ counts, int N, float* __restrict__ bank,
                                                         const int* __restrict__ state, const int* __restrict__ caps,
                                                         int cap_stride) {
  const int c = blockIdx.y;
  const int cap = caps[c];
  const int m = counts[c * 3 + 2], rows = state[c * 2 + 0], head = state[c * 2 + 1];
  const int* l2 = lists + ((size_t)c * 3 + 2) * N;
  const int j0 = m > cap ? m - cap : 0;
  const long long tot = (long long)(m - j0) * D;
  float* bc = bank + (size_t)c * cap_stride * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
    const int j = j0 + (int)(e / D), d = (int)(e - (long long)(j - j0) * D);
    const int slot = (int)(((long long)head + rows + j) % cap);
    bc[(size_t)slot * D + d] = rep_t[(size_t)l2[j] * D + d];
  }
}

__global__ void mb_state_kernel(const int* __restrict__ counts, int* __restrict__ state, int K,
                                const int* __restrict__ caps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  const int cap = caps[c];
  const int m = counts[c * 3 + 2], rows = state[c * 2 + 0], head = state[c * 2 + 1];
  const long long total = (long long)rows + m;
  const long long dropped = total > cap ? total - cap : 0;
  state[c * 2 + 0] = (int)(total > cap ? cap : total);
  state[c * 2 + 1] = (int)(((long long)head + dropped) % cap);
}

// ------------------------------------------------------------------------------------------
// InfoNCE of one loop position (loss_helper.py:164-215): one workgroup per query.
//   anchor  = rep[pool[anchor_draw[q]]]                       (gradient flows here only)
//   keys    = [ positive(q) ; bank[(head + neg_draw[q][j]) % cap], j < NN ]
//   logits  = cos(anchor, key) / temp, target 0, mean over queries; `scale` = 1 / valid_seg
// torch.cosine_similarity normalises each vector by max(|x|, 1e-8) and then takes the dot product.
// ------------------------------------------------------------------------------------------
constexpr int MB_MAXKEYS = 128;

__global__ __launch_bounds__(256) void mb_infonce_kernel(const float* __restrict__ rep, int D, const int* __restrict__ pool,
                                                         const long long* __restrict__ anchor_draw,
                                                         const float* __restrict__ pos, long long pos_qstride,
                                                         const float* __restrict__ bank_c, int cap, int head,
                                                         const long long* __restrict__ neg_draw, int NN, float temp,
                                                         float scale, int Qn, float* __restrict__ lossq,
                                                         float* __restrict__ ganchor) {
  __shared__ float s_dot[MB_MAXKEYS], s_kn[MB_MAXKEYS], s_w[MB_MAXKEYS];
  __shared__ const float* s_key[MB_MAXKEYS];
  __shared__ float s_an2[4], s_misc[2];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NK = NN + 1;
  const float* a = rep + (size_t)pool[anchor_draw[q]] * D;
  for (int j = tid; j < NK; j += 256) {
    if (j == 0) s_key[0] = pos + (long long)q * pos_qstride;
    else {
      const long long r = neg_draw[(long long)q * NN + (j - 1)];
      s_key[j] = bank_c + (size_t)(((long long)head + r) % cap) * D;
    }
  }
  {  // |a|^2
    float s = 0.f;
    for (int d = tid; d < D; d += 256) { const float v = a[d]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) s_an2[wave] = s;
  }
  __syncthreads();
  const float an = sqrtf((s_an2[0] + s_an2[1]) + (s_an2[2] + s_an2[3]));
  const float anc = fmaxf(an, 1e-8f);
  for (int j = wave; j < NK; j += 4) {           // one wave per key: a . k and |k|^2
    const float* k = s_key[j];
    float dt = 0.f, kn = 0.f;
    for (int d = lane; d < D; d += 64) { const float kv = k[d]; dt += a[d] * kv; kn += kv * kv; }
    dt = wave_sum(dt); kn = wave_sum(kn);
    if (lane == 0) { s_dot[j] = dt; s_kn[j] = fmaxf(sqrtf(kn), 1e-8f); }
  }
  __syncthreads();
  if (wave == 0) {                               // softmax over the NK logits (NK <= 128: two per lane)
    float l0 = -3.0e38f, l1 = -3.0e38f, c0 = 0.f, c1 = 0.f;
    if (lane < NK) { c0 = s_dot[lane] / (anc * s_kn[lane]); l0 = c0 / temp; }
    if (lane + 64 < NK) { c1 = s_dot[lane + 64] / (anc * s_kn[lane + 64]); l1 = c1 / temp; }
    const float mx = wave_max(fmaxf(l0, l1));
    const float e0 = lane < NK ? expf(l0 - mx) : 0.f, e1 = lane + 64 < NK ? expf(l1 - mx) : 0.f;
    const float se = wave_sum(e0 + e1);
    const float first = __shfl(l0, 0, 64);
    // d loss / d cos_j = (softmax_j - [j == 0]) / temp * scale / Qn
    const float g = scale / ((float)Qn * temp);
    const float w0 = (e0 / se - (lane == 0 ? 1.f : 0.f)) * g, w1 = (e1 / se) * g;
    float wc = 0.f;
    if (lane < NK) { s_w[lane] = w0 / (anc * s_kn[lane]); wc += w0 * c0; }
    if (lane + 64 < NK) { s_w[lane + 64] = w1 / (anc * s_kn[lane + 64]); wc += w1 * c1; }
    wc = wave_sum(wc);
    if (lane == 0) {
      lossq[q] = (mx + logf(se) - first) * scale / (float)Qn;
      s_misc[0] = wc / (anc * anc);              // sum_j w_j cos_j / |a|^2
      s_misc[1] = (an >= 1e-8f) ? 1.f : 0.f;     // below the clamp the normalisation is a constant: no a-term
    }
  }
  __syncthreads();
  const float selfc = s_misc[0] * s_misc[1];
  float* go = ganchor + (size_t)q * D;
  for (int d = tid; d < D; d += 256) {
    float s = 0.f;
    for (int j = 0; j < NK; ++j) s += s_w[j] * s_key[j][d];
    go[d] = s - selfc * a[d];
  }
}

// d rep[row] += sum of the anchor gradients of the queries that drew that row, in query order.  One workgroup per
// query: it proceeds only if no earlier query drew the same row (the "leader"), collects the later duplicates in
// order, and does the single read-modify-write of its row -- distinct leaders own distinct rows, so the result
// does not depend on scheduling.
__global__ __launch_bounds__(256) void mb_scatter_kernel(const float* __restrict__ ganchor, const int* __restrict__ pool,
                                                         const long long* __restrict__ anchor_draw, int Qn, int D,
                                                         float* __restrict__ drep) {
  extern __shared__ int s_mem[];                  // rows[Qn] | dup flags[Qn]
  __shared__ int s_follower;
  int* s_row = s_mem;
  int* s_dup = s_mem + Qn;
  const int q = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_follower = 0;
  for (int p = tid; p < Qn; p += 256) s_row[p] = pool[anchor_draw[p]];
  __syncthreads();
  const int r = s_row[q];
  for (int p = tid; p < Qn; p += 256) {
    const bool same = s_row[p] == r;
    s_dup[p] = (same && p > q) ? 1 : 0;
    if (same && p < q) s_follower = 1;            // benign race: every writer stores 1
  }
  __syncthreads();
  if (s_follower) return;
  for (int d = tid; d < D; d += 256) {
    float s = ganchor[(size_t)q * D + d];
    for (int p = q + 1; p < Qn; ++p)
      if (s_dup[p]) s += ganchor[(size_t)p * D + d];
    drep[(size_t)r * D + d] += s;
  }
}

__global__ void mb_sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];
    out[0] = s;
  }
}

// ------------------------------------------------------------------------------------------
// compute_unsupervised_loss (loss_helper.py:242-261)
//   ws: ent[B] | sel[2] (a[lo], a[hi]) | info[4] (n_valid, kept, loss_sum, thresh)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void us_entropy_kernel(const float* __restrict__ teacher, const long long* __restrict__ target,
                                                         int B, int K, float* __restrict__ ent, int* __restrict__ nvalid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool valid = false;
  if (i < B) {
    const float* t = teacher + (size_t)i * K;
    float mx = -3.0e38f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, t[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(t[k] - mx);
    float e = 0.f;
    for (int k = 0; k < K; ++k) { const float p = expf(t[k] - mx) / se; e -= p * logf(p + 1e-10f); }
    ent[i] = e;
    valid = target[i] != 255;
  }
  const unsigned long long b = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(nvalid, __popcll(b));     // integer count: order-independent
}

// order statistics a[lo], a[hi] of the valid entropies by rank counting (stable: ties by row index).  The B x B
// comparisons are spread over (row block, column slice) workgroups; the partial ranks meet in integer atomics
// (order-independent, so the result is deterministic), and a second tiny launch picks the two entropies.
constexpr int US_SLICES = 16;
__global__ __launch_bounds__(256) void us_rank_kernel(const float* __restrict__ ent, const long long* __restrict__ target, int B,
                                                      int* __restrict__ rank) {
  __shared__ float s_e[256];
  __shared__ int s_v[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float ei = i < B ? ent[i] : 0.f;
  const int per = (((B + US_SLICES - 1) / US_SLICES) + 255) / 256 * 256;
  const int jb = blockIdx.y * per, je = (jb + per < B) ? jb + per : B;
  int cnt = 0;
  for (int j0 = jb; j0 < je; j0 += 256) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    s_e[threadIdx.x] = j < je ? ent[j] : 0.f;
    s_v[threadIdx.x] = (j < je && target[j] != 255) ? 1 : 0;
    __syncthreads();
    const int lim = je - j0 < 256 ? je - j0 : 256;
    for (int t = 0; t < lim; ++t)
      if (s_v[t]) cnt += (s_e[t] < ei || (s_e[t] == ei && j0 + t < i)) ? 1 : 0;
  }
  if (i < B && cnt) atomicAdd(rank + i, cnt);
}

__global__ __launch_bounds__(256) void us_select_kernel(const float* __restrict__ ent, const long long* __restrict__ target, int B,
                                                        double percent, const int* __restrict__ nvalid,
                                                        const int* __restrict__ rank, float* __restrict__ sel) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = *nvalid;
  if (n <= 0 || i >= B || target[i] == 255) return;
  const double vidx = (double)(n - 1) * percent / 100.0;
  const int lo = (int)floor(vidx), hi = lo + 1 < n ? lo + 1 : n - 1;
  const int r = rank[i];
  if (r == lo) sel[0] = ent[i];
  if (r == hi) sel[1] = ent[i];
}

// threshold (numpy 'linear' percentile, evaluated in double like numpy does for a float32 array and a float64
// fraction, then rounded to float32 for the comparison with the float32 entropies), drop mask, kept count, CE sum
__global__ __launch_bounds__(256) void us_mask_kernel(const float* __restrict__ predict, long long* __restrict__ target,
                                                      const float* __restrict__ ent, int B, int K, double percent,
                                                      const int* __restrict__ nvalid, const float* __restrict__ sel,
                                                      float* __restrict__ rowloss, int* __restrict__ kept) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = *nvalid;
  float thr = 3.0e38f;
  if (n > 0) {
    const double vidx = (double)(n - 1) * percent / 100.0;
    const double g = vidx - floor(vidx);
    const double a = (double)sel[0], b = (double)sel[1], diff = b - a;
    double t = a + diff * g;
    if (g >= 0.5) t = b - diff * (1.0 - g);
    if (diff == 0.0) t = a;
    thr = (float)t;
  }
  bool keep = false;
  if (i < B) {
    long long tg = target[i];
    if (tg != 255 && ent[i] >= thr) { tg = 255; target[i] = 255; }
    keep = tg != 255;
    float l = 0.f;
    if (keep) {
      const float* p = predict + (size_t)i * K;
      float mx = -3.0e38f;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, p[k]);
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += expf(p[k] - mx);
      l = mx + logf(se) - p[tg];
    }
    rowloss[i] = l;
  }
  const unsigned long long bm = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && bm) atomicAdd(kept, __popcll(bm));
}

__global__ __launch_bounds__(256) void us_grad_kernel(const float* __restrict__ predict, const long long* __restrict__ target,
                                                      const float* __restrict__ rowloss, int B, int K,
                                                      const int* __restrict__ kept, float* __restrict__ loss,
                                                      float* __restrict__ dpredict) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float kf = (float)(*kept);
  const float weight = (float)B / kf;                    // :256 (inf / NaN when nothing is kept, like the reference)
  if (blockIdx.x == 0) {                                 // fixed-order sum: per-thread strided partials, then in order
    __shared__ float s_part[256];
    float s = 0.f;
    for (int r = threadIdx.x; r < B; r += 256) s += rowloss[r];
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int k = 0; k < 256; ++k) t += s_part[k];
      loss[0] = weight * (t / kf);
    }
  }
  if (i < B) {
    const long long tg = target[i];
    const float* p = predict + (size_t)i * K;
    float* g = dpredict + (size_t)i * K;
    if (tg == 255) {
      for (int k = 0; k < K; ++k) g[k] = 0.f;
    } else {
      float mx = -3.0e38f;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, p[k]);
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += expf(p[k] - mx);
      const float sc = weight / kf;
      for (int k = 0; k < K; ++k) g[k] = sc * (expf(p[k] - mx) / se - (k == (int)tg ? 1.f : 0.f));
    }
  }
}

// ==========================================================================================
// One-pass form of compute_contra_memobank_loss (no host read-back between the stages): three launches.
//   mb_prepare_kernel     one workgroup per class: selection lists + counts, class prototype, enqueue of the
//                         negative keys into the ring, ring state update
//   mb_infonce_all_kernel one workgroup per (query, loop position): which classes are valid and what the loop
//                         position means is worked out ON DEVICE from the counts (the reference reads them back
//                         with .item()); draws either injected or formed in-kernel (Philox, like torch.randint
//                         statistically); EMA blend of the positive with the momentum prototype in-kernel
//   mb_scatter_all_kernel anchor-gradient scatter over ALL positions (duplicate rows summed in (position, query)
//                         order: deterministic) + the fixed-order sum of the per-query losses
// ==========================================================================================
__global__ __launch_bounds__(256) void mb_prepare_kernel(MbPrep a) {
  __shared__ int wtot[3][4];
  __shared__ int base[3];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, Nl = a.Nl, K = a.K, D = a.D;
  if (tid < 3) base[tid] = 0;
  __syncthreads();
  int* l0 = a.lists + ((size_t)c * 3 + 0) * N;
  int* l1 = a.lists + ((size_t)c * 3 + 1) * N;
  int* l2 = a.lists + ((size_t)c * 3 + 2) * N;
  for (int n0 = 0; n0 < N; n0 += 256) {
    const int n = n0 + tid;
    bool f0 = false, f1 = false, f2 = false;
    if (n < N) {
      const float* pr = (n < Nl) ? a.prob_l + (size_t)n * K : a.prob_u + (size_t)(n - Nl) * K;
      const float lab = (n < Nl) ? a.label_l[(size_t)n * K + c] : a.label_u[(size_t)(n - Nl) * K + c];
      const float pc = pr[c];
      f0 = (lab * a.low_mask[n]) != 0.f;
      const bool hv = (lab * a.high_mask[n]) != 0.f;
      f1 = (pc > 0.3f) && f0;
      int rank = 0;                                   // position of class c in the descending sort of the row
      for (int j = 0; j < K; ++j) rank += (pr[j] > pc || (pr[j] == pc && j < c)) ? 1 : 0;
      const bool cm = (n < Nl) ? (rank < 3 && lab == 0.f) : (rank >= 3 && rank < 9);
      f2 = (pc < 1.0f) && hv && cm;
    }
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1), b2 = __ballot(f2);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) { wtot[0][wave] = __popcll(b0); wtot[1][wave] = __popcll(b1); wtot[2][wave] = __popcll(b2); }
    __syncthreads();
    int o0 = base[0], o1 = base[1], o2 = base[2];
    for (int w = 0; w < wave; ++w) { o0 += wtot[0][w]; o1 += wtot[1][w]; o2 += wtot[2][w]; }
    if (f0) l0[o0 + __popcll(b0 & below)] = n;
    if (f1) l1[o1 + __popcll(b1 & below)] = n;
    if (f2) l2[o2 + __popcll(b2 & below)] = n;
    __syncthreads();
    if (tid < 3) base[tid] += wtot[tid][0] + wtot[tid][1] + wtot[tid][2] + wtot[tid][3];
    __syncthreads();
  }
  const int cnt0 = base[0], m = base[2];
  if (tid < 3) a.counts[c * 3 + tid] = base[tid];
  // the lists were written by this workgroup: make them visible to all of its threads
  __threadfence_block();
  __syncthreads();
  // prototype: mean of the teacher features of the low_valid rows (NaN for an empty class, like torch.mean)
  for (int d = tid; d < D; d += 256) {
    float sum = 0.f;
    for (int i = 0; i < cnt0; ++i) sum += a.rep_t[(size_t)l0[i] * D + d];
    a.proto[(size_t)c * D + d] = cnt0 > 0 ? sum / (float)cnt0 : __builtin_nanf("");
  }
  // dequeue_and_enqueue of the negative keys (only the last `cap` of them can survive)
  const int cap = a.caps[c];
  const int rows = a.state[c * 2 + 0], head = a.state[c * 2 + 1];
  const int j0 = m > cap ? m - cap : 0;
  float* bc = a.bank + (size_t)c * a.cap_stride * D;
  for (int j = j0; j < m; ++j) {
    const int slot = (int)(((long long)head + rows + j) % cap);
    const float* src = a.rep_t + (size_t)l2[j] * D;
    for (int d = tid; d < D; d += 256) bc[(size_t)slot * D + d] = src[d];
  }
  __syncthreads();                                    // every thread has read the old state
  if (tid == 0) {
    const long long total = (long long)rows + m;
    const long long dropped = total > cap ? total - cap : 0;
    const int rows_after = (int)(total > cap ? cap : total);
    a.state[c * 2 + 0] = rows_after;
    a.state[c * 2 + 1] = (int)(((long long)head + dropped) % cap);
    if (a.keys_log != nullptr) { a.keys_log[c * 2 + 0] = m; a.keys_log[c * 2 + 1] = rows_after; }
  }
}

// valid classes (count of low_valid rows > 0) in class order -> s_valid[0 .. nvalid); K <= 1024
__device__ __forceinline__ int mb_valid_list(const int* counts, int K, int* s_valid, int* s_cnt, int tid) {
  if (tid == 0) {
    int nv = 0;
    for (int c = 0; c < K; ++c)
      if (counts[c * 3 + 0] > 0) s_valid[nv++] = c;
    *s_cnt = nv;
  }
  __syncthreads();
  return *s_cnt;
}

__global__ __launch_bounds__(256) void mb_infonce_all_kernel(MbLoss a) {
  __shared__ float s_dot[MB_MAXKEYS], s_kn[MB_MAXKEYS], s_w[MB_MAXKEYS];
  __shared__ const float* s_key[MB_MAXKEYS];
  __shared__ float s_an2[4], s_misc[2];
  __shared__ int s_valid[1024], s_nv;
  extern __shared__ __attribute__((aligned(16))) float s_pos[];   // [D]: this query's positive key
  const int q = blockIdx.x, i = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = a.D, NN = a.NN, NK = NN + 1, Q = a.Q;
  const int nv = mb_valid_list(a.counts, a.K, s_valid, &s_nv, tid);
  const int slot = i * Q + q;
  if (nv <= 1 || i >= nv) {                          // loss_helper.py:139-145, or no such loop position
    if (tid == 0) { a.lossq[slot] = 0.f; a.arow[slot] = -1; }
    return;
  }
  const int vc = s_valid[i];
  const int pool_rows = a.counts[i * 3 + 1];         // position i's anchor pool (the reference's quirk, :158-168)
  const int rows = a.state[vc * 2 + 0], head = a.state[vc * 2 + 1], cap = a.caps[vc];
  if (pool_rows == 0 || rows == 0) {                 // :158-172
    if (tid == 0) { a.lossq[slot] = 0.f; a.arow[slot] = -1; }
    return;
  }
  const float scale = 1.f / (float)nv;
  const int* pool = a.lists + ((size_t)i * 3 + 1) * a.N;
  long long ad;
  if (a.anchor_draw != nullptr) {                    // injected (the caller validates the range; clamp for safety)
    ad = a.anchor_draw[(size_t)i * Q + q];
    ad = ad < 0 ? 0 : (ad >= pool_rows ? pool_rows - 1 : ad);
  } else {
    const float4 u = philox_uniform4(a.seed, a.call, 0x400 + i, (uint64_t)q);
    ad = (long long)(u.x * (float)pool_rows);
    if (ad >= pool_rows) ad = pool_rows - 1;
  }
  const int arow = pool[ad];
  const float* av = a.rep + (size_t)arow * D;
  const float* bank_c = a.bank + (size_t)vc * a.cap_stride * D;
  // positive key: the class prototype of POSITION i (:186-192), blended with the momentum prototype (:194-203)
  {
    const float* pr = a.proto + (size_t)i * D;
    const bool blend = a.momentum != nullptr && *a.momentum_on != 0;
    for (int d = tid; d < D; d += 256) {
      float v = pr[d];
      if (blend) v = (1.f - a.ema) * v + a.ema * a.momentum[((size_t)vc * Q + q) * D + d];
      s_pos[d] = v;
      if (a.prototype != nullptr) a.prototype[((size_t)vc * Q + q) * D + d] = v;
    }
  }
  for (int j = tid; j < NK; j += 256) {
    if (j == 0) s_key[0] = s_pos;
    else {
      long long r;
      if (a.neg_draw != nullptr) {
        r = a.neg_draw[(size_t)i * Q * NN + (size_t)q * NN + (j - 1)];
        r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
      } else {
        const int jj = j - 1;
        const float4 u = philox_uniform4(a.seed, a.call, 0x500 + i, (uint64_t)q * 64 + (jj >> 2));
        const float uu = (jj & 3) == 0 ? u.x : (jj & 3) == 1 ? u.y : (jj & 3) == 2 ? u.z : u.w;
        r = (long long)(uu * (float)rows);
        if (r >= rows) r = rows - 1;
      }
      s_key[j] = bank_c + (size_t)(((long long)head + r) % cap) * D;
    }
  }
  {  // |a|^2
    float s = 0.f;
    for (int d = tid; d < D; d += 256) { const float v = av[d]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) s_an2[wave] = s;
  }
  __syncthreads();
  const float an = sqrtf((s_an2[0] + s_an2[1]) + (s_an2[2] + s_an2[3]));
  const float anc = fmaxf(an, 1e-8f);
  // (general kernel: one wave per key, the keys re-read from L2 for the gradient pass; the default where it fits is
  // mb_infonce_fast_kernel below, which keeps them in registers)
  for (int j = wave; j < NK; j += 4) {           // a . k and |k|^2
    const float* k = s_key[j];
    float dt = 0.f, kn = 0.f;
    for (int d = lane; d < D; d += 64) { const float kv = k[d]; dt += av[d] * kv; kn += kv * kv; }
    dt = wave_sum(dt); kn = wave_sum(kn);
    if (lane == 0) { s_dot[j] = dt; s_kn[j] = fmaxf(sqrtf(kn), 1e-8f); }
  }
  __syncthreads();
  if (wave == 0) {                               // softmax over the NK logits (NK <= 128: two per lane)
    float l0 = -3.0e38f, l1 = -3.0e38f, c0 = 0.f, c1 = 0.f;
    if (lane < NK) { c0 = s_dot[lane] / (anc * s_kn[lane]); l0 = c0 / a.temp; }
    if (lane + 64 < NK) { c1 = s_dot[lane + 64] / (anc * s_kn[lane + 64]); l1 = c1 / a.temp; }
    const float mx = wave_max(fmaxf(l0, l1));
    const float e0 = lane < NK ? expf(l0 - mx) : 0.f, e1 = lane + 64 < NK ? expf(l1 - mx) : 0.f;
    const float se = wave_sum(e0 + e1);
    const float first = __shfl(l0, 0, 64);
    const float g = scale / ((float)Q * a.temp);
    const float w0 = (e0 / se - (lane == 0 ? 1.f : 0.f)) * g, w1 = (e1 / se) * g;
    float wc = 0.f;
    if (lane < NK) { s_w[lane] = w0 / (anc * s_kn[lane]); wc += w0 * c0; }
    if (lane + 64 < NK) { s_w[lane + 64] = w1 / (anc * s_kn[lane + 64]); wc += w1 * c1; }
    wc = wave_sum(wc);
    if (lane == 0) {
      a.lossq[slot] = (mx + logf(se) - first) * scale / (float)Q;
      a.arow[slot] = arow;
      s_misc[0] = wc / (anc * anc);
      s_misc[1] = (an >= 1e-8f) ? 1.f : 0.f;
    }
  }
  __syncthreads();
  const float selfc = s_misc[0] * s_misc[1];
  float* go = a.ganchor + (size_t)slot * D;
  for (int d = tid; d < D; d += 256) {
    float s = 0.f;
    for (int j = 0; j < NK; ++j) s += s_w[j] * s_key[j][d];
    go[d] = s - selfc * av[d];
  }
}

// The same computation with the metadata in ONE round trip and the keys in REGISTERS (round 3; the default where it
// fits: D a multiple of 4 and <= 1024, K <= 64, ceil(NK / 4) * ceil(D / 256) <= 16 -- the reference's 50 negatives of
// 256 floats take 13 of the 16 slots).  The kernel above is a chain of ~8 dependent memory round trips per workgroup
// (counts -> ring state -> anchor pool -> anchor row -> four batches of keys), and with the gathered keys kept in LDS
// only three workgroups share a CU: 72 us for 120 MB of gathers.  Here every wave (a) reads the counts, ring states and
// capacities of all classes at once (lane = class) and picks what it needs with a shuffle, (b) forms the addresses of
// ITS keys j = wave, wave + 4, ... itself (lane t draws key t; no table in LDS, no barrier) and requests all of them --
// one 16-byte load per lane and 256 floats of key -- before the anchor's pool entry and row are even known, (c) keeps
// them in registers for the gradient pass: partial sum_j w_j k_j per wave, folded through 4 x D floats of LDS.  Same
// draws (injected or Philox), same per-key dot products bit for bit; |a|^2 and the gradient sum fold in another order.
template <int NCH>
__global__ __launch_bounds__(256) void mb_infonce_fast_kernel(MbLoss a) {
  constexpr int KW = 16 / NCH;                    // keys per wave
  __shared__ float s_dot[MB_MAXKEYS], s_kn[MB_MAXKEYS];
  extern __shared__ __attribute__((aligned(16))) float s_red[];   // [4][D]
  const int q = blockIdx.x, i = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = a.D, D4 = D >> 2, NN = a.NN, NK = NN + 1, Q = a.Q, K = a.K;
  const int slot = i * Q + q;
  // (a) everything the draws depend on, one round trip
  const bool cv = lane < K;
  const int cnt0 = cv ? a.counts[lane * 3] : 0;
  const int st_rows = cv ? a.state[lane * 2] : 0, st_head = cv ? a.state[lane * 2 + 1] : 0, st_cap = cv ? a.caps[lane] : 1;
  const int pool_rows = a.counts[i * 3 + 1];         // position i's anchor pool (the reference's quirk, :158-168)
  unsigned long long vm = __ballot(cv && cnt0 > 0);  // valid classes (loss_helper.py:139-145), in class order
  const int nv = __popcll(vm);
  if (nv <= 1 || i >= nv) {                          // no such loop position
    if (tid == 0) { a.lossq[slot] = 0.f; a.arow[slot] = -1; }
    return;
  }
  for (int s_ = 0; s_ < i; ++s_) vm &= vm - 1;
  const int vc = __ffsll((long long)vm) - 1;
  const int rows = __shfl(st_rows, vc, 64), head = __shfl(st_head, vc, 64), cap = __shfl(st_cap, vc, 64);
  if (pool_rows == 0 || rows == 0) {                 // :158-172
    if (tid == 0) { a.lossq[slot] = 0.f; a.arow[slot] = -1; }
    return;
  }
  const float scale = 1.f / (float)nv;
  const float* bank_c = a.bank + (size_t)vc * a.cap_stride * D;
  // (b) this wave's keys: lane t owns key j = wave + 4 t
  const int jt = wave + 4 * lane;
  int krow = 0;
  if (lane < KW && jt >= 1 && jt < NK) {
    long long r;
    if (a.neg_draw != nullptr) {
      r = a.neg_draw[(size_t)i * Q * NN + (size_t)q * NN + (jt - 1)];
      r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    } else {
      const int jj = jt - 1;
      const float4 u = philox_uniform4(a.seed, a.call, 0x500 + i, (uint64_t)q * 64 + (jj >> 2));
      const float uu = (jj & 3) == 0 ? u.x : (jj & 3) == 1 ? u.y : (jj & 3) == 2 ? u.z : u.w;
      r = (long long)(uu * (float)rows);
      if (r >= rows) r = rows - 1;
    }
    krow = (int)(((long long)head + r) % cap);
  }
  float4 kv[KW][NCH];
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    const int j = wave + 4 * t;                      // wave-uniform
    const float4* k4 = (const float4*)(bank_c + (size_t)__builtin_amdgcn_readlane(krow, t) * D);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int d4 = lane + 64 * c;
      kv[t][c] = (j >= 1 && j < NK && d4 < D4) ? k4[d4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // the anchor (the draw is the same in every lane)
  long long ad;
  if (a.anchor_draw != nullptr) {                    // injected (the caller validates the range; clamp for safety)
    ad = a.anchor_draw[(size_t)i * Q + q];
    ad = ad < 0 ? 0 : (ad >= pool_rows ? pool_rows - 1 : ad);
  } else {
    const float4 u = philox_uniform4(a.seed, a.call, 0x400 + i, (uint64_t)q);
    ad = (long long)(u.x * (float)pool_rows);
    if (ad >= pool_rows) ad = pool_rows - 1;
  }
  const int arow = (a.lists + ((size_t)i * 3 + 1) * a.N)[ad];
  const float4* av4 = (const float4*)(a.rep + (size_t)arow * D);
  float4 a4[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { const int d4 = lane + 64 * c; a4[c] = d4 < D4 ? av4[d4] : make_float4(0.f, 0.f, 0.f, 0.f); }
  // positive key (j = 0, wave 0): the class prototype of POSITION i (:186-192), blended with the momentum prototype (:194-203)
  if (wave == 0) {
    const float4* pr = (const float4*)(a.proto + (size_t)i * D);
    const bool blend = a.momentum != nullptr && *a.momentum_on != 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int d4 = lane + 64 * c;
      if (d4 < D4) {
        float4 v = pr[d4];
        if (blend) {
          const float4 m = ((const float4*)(a.momentum + ((size_t)vc * Q + q) * D))[d4];
          v.x = (1.f - a.ema) * v.x + a.ema * m.x; v.y = (1.f - a.ema) * v.y + a.ema * m.y;
          v.z = (1.f - a.ema) * v.z + a.ema * m.z; v.w = (1.f - a.ema) * v.w + a.ema * m.w;
        }
        kv[0][c] = v;
        if (a.prototype != nullptr) ((float4*)(a.prototype + ((size_t)vc * Q + q) * D))[d4] = v;
      }
    }
  }
  // a . k and |k|^2 of this wave's keys (lane t keeps key t's pair), |a|^2 (every wave forms it: no exchange)
  float mydt = 0.f, mykn = 1.f;
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    const int j = wave + 4 * t;
    if (j < NK) {                                    // wave-uniform
      float dt = 0.f, kn = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 k = kv[t][c], x = a4[c];
        dt += x.x * k.x; kn += k.x * k.x; dt += x.y * k.y; kn += k.y * k.y;
        dt += x.z * k.z; kn += k.z * k.z; dt += x.w * k.w; kn += k.w * k.w;
      }
      dt = wave_sum(dt); kn = wave_sum(kn);
      if (lane == t) { mydt = dt; mykn = fmaxf(sqrtf(kn), 1e-8f); }
    }
  }
  if (lane < KW && jt < NK) { s_dot[jt] = mydt; s_kn[jt] = mykn; }
  float an2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) an2 += (a4[c].x * a4[c].x + a4[c].y * a4[c].y) + (a4[c].z * a4[c].z + a4[c].w * a4[c].w);
  const float an = sqrtf(wave_sum(an2));
  const float anc = fmaxf(an, 1e-8f);
  __syncthreads();
  // softmax over the NK logits (NK <= 128: two per lane), formed by every wave for itself
  float l0 = -3.0e38f, l1 = -3.0e38f, c0 = 0.f, c1 = 0.f, kn0 = 1.f, kn1 = 1.f;
  if (lane < NK) { kn0 = s_kn[lane]; c0 = s_dot[lane] / (anc * kn0); l0 = c0 / a.temp; }
  if (lane + 64 < NK) { kn1 = s_kn[lane + 64]; c1 = s_dot[lane + 64] / (anc * kn1); l1 = c1 / a.temp; }
  const float mx = wave_max(fmaxf(l0, l1));
  const float e0 = lane < NK ? expf(l0 - mx) : 0.f, e1 = lane + 64 < NK ? expf(l1 - mx) : 0.f;
  const float se = wave_sum(e0 + e1);
  const float first = __shfl(l0, 0, 64);
  const float g = scale / ((float)Q * a.temp);
  const float w0 = (e0 / se - (lane == 0 ? 1.f : 0.f)) * g, w1 = (e1 / se) * g;
  float wc = 0.f, sw0 = 0.f, sw1 = 0.f;
  if (lane < NK) { sw0 = w0 / (anc * kn0); wc += w0 * c0; }
  if (lane + 64 < NK) { sw1 = w1 / (anc * kn1); wc += w1 * c1; }
  wc = wave_sum(wc);
  if (tid == 0) {
    a.lossq[slot] = (mx + logf(se) - first) * scale / (float)Q;
    a.arow[slot] = arow;
  }
  const float selfc = (wc / (anc * anc)) * ((an >= 1e-8f) ? 1.f : 0.f);
  // (c) gradient wrt the anchor: this wave's keys from registers, the four partial sums through LDS
  float4 gs[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) gs[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < KW; ++t) {
    const int j = wave + 4 * t;
    if (j < NK) {
      const float w = (j < 64) ? __shfl(sw0, j, 64) : __shfl(sw1, j - 64, 64);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        gs[c].x += w * kv[t][c].x; gs[c].y += w * kv[t][c].y; gs[c].z += w * kv[t][c].z; gs[c].w += w * kv[t][c].w;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) { const int d4 = lane + 64 * c; if (d4 < D4) ((float4*)(s_red + (size_t)wave * D))[d4] = gs[c]; }
  __syncthreads();
  float4* go = (float4*)(a.ganchor + (size_t)slot * D);
  for (int d4 = tid; d4 < D4; d4 += 256) {
    const float4 r0 = ((const float4*)s_red)[d4], r1 = ((const float4*)(s_red + D))[d4];
    const float4 r2 = ((const float4*)(s_red + 2 * (size_t)D))[d4], r3 = ((const float4*)(s_red + 3 * (size_t)D))[d4];
    const float4 x = av4[d4];
    go[d4] = make_float4(((r0.x + r1.x) + (r2.x + r3.x)) - selfc * x.x, ((r0.y + r1.y) + (r2.y + r3.y)) - selfc * x.y,
                         ((r0.z + r1.z) + (r2.z + r3.z)) - selfc * x.z, ((r0.w + r1.w) + (r2.w + r3.w)) - selfc * x.w);
  }
}

// d rep[row] = sum of the anchor gradients of ALL (position, query) slots that drew that row, in slot order
// (deterministic).  One workgroup per ROW of rep: it collects the slots that drew its row by an ordered compaction
// of the slot -> row table and writes its row of drep in full (zeros when nobody drew it).  The extra last
// workgroup sums the per-query losses in fixed order.
__global__ __launch_bounds__(256) void mb_scatter_all_kernel(MbLoss a, int slots) {
  extern __shared__ int s_hit[];                  // ordered list of the slots that drew this row
  __shared__ int wtot[4], s_n;
  const int r = blockIdx.x, tid = threadIdx.x, D = a.D, lane = tid & 63, wave = tid >> 6;
  if (r == a.N) {                                 // total loss, in a fixed order: thread t takes slots t, t + 256, ...,
    __shared__ float s_tot[256];                  // then the 256 partial sums fold as a tree.  (One thread adding all
    float t = 0.f;                                // K * Q values one load after the other WAS the launch: 93 us.)
    for (int i = tid; i < slots; i += 256) t += a.lossq[i];
    s_tot[tid] = t;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (tid < h) s_tot[tid] += s_tot[tid + h];
      __syncthreads();
    }
    if (tid == 0) a.total[0] = s_tot[0];
    return;
  }
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int p0 = 0; p0 < slots; p0 += 256) {       // ordered compaction, 256 slots at a time
    const int p = p0 + tid;
    const bool hit = p < slots && a.arow[p] == r;
    const unsigned long long bal = __ballot(hit);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int o = s_n;
    for (int w = 0; w < wave; ++w) o += wtot[w];
    if (hit) s_hit[o + __popcll(bal & below)] = p;
    __syncthreads();
    if (tid == 0) s_n += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
  }
  const int nh = s_n;
  for (int d = tid; d < D; d += 256) {
    float sum = 0.f;
    int k = 0;
    for (; k + 8 <= nh; k += 8) {                 // eight rows in flight, added in slot order
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = a.ganchor[(size_t)s_hit[k + e] * D + d];
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[e];
    }
    for (; k < nh; ++k) sum += a.ganchor[(size_t)s_hit[k] * D + d];
    a.drep[(size_t)r * D + d] = sum;
  }
}

hipError_t launch_mb_onepass(const MbPrep& pa, const MbLoss& la, hipStream_t st) {
  if (la.NN + 1 > MB_MAXKEYS || la.K > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mb_prepare_kernel, dim3(pa.K), dim3(256), 0, st, pa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  {
    // keys in registers where they fit (mb_infonce_fast_kernel), else the general kernel (keys re-read from L2)
    const int NK = la.NN + 1, nch = (la.D / 4 + 63) / 64;
    const bool fast = la.D % 4 == 0 && la.D >= 4 && la.D <= 1024 && la.K <= 64 && ((NK + 3) / 4) * nch <= 16 &&
                      switches().mb_fast != 0;
    const size_t lds = (size_t)4 * la.D * 4;
    if (fast && nch == 1) hipLaunchKernelGGL(mb_infonce_fast_kernel<1>, dim3(la.Q, la.K), dim3(256), lds, st, la);
    else if (fast && nch == 2) hipLaunchKernelGGL(mb_infonce_fast_kernel<2>, dim3(la.Q, la.K), dim3(256), lds, st, la);
    else if (fast && nch == 3) hipLaunchKernelGGL(mb_infonce_fast_kernel<3>, dim3(la.Q, la.K), dim3(256), lds, st, la);
    else if (fast && nch == 4) hipLaunchKernelGGL(mb_infonce_fast_kernel<4>, dim3(la.Q, la.K), dim3(256), lds, st, la);
    else {
      hipLaunchKernelGGL(mb_infonce_all_kernel, dim3(la.Q, la.K), dim3(256), (size_t)la.D * 4, st, la);
    }
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  const int slots = la.K * la.Q;
  const size_t lds = (size_t)slots * sizeof(int);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mb_scatter_all_kernel, dim3(la.N + 1), dim3(256), lds, st, la, slots);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ launchers
hipError_t launch_mb_select(const float* prob, const float* label, const float* low_mask, const float* high_mask, int N,
                            int Nl, int K, int* lists, int* counts, hipStream_t st) {
  hipLaunchKernelGGL(mb_select_kernel, dim3(K), dim3(256), 0, st, prob, label, low_mask, high_mask, N, Nl, K, 0.3f, 1.0f,
                     3, 9, lists, counts);
  return hipGetLastError();
}

hipError_t launch_mb_proto(const float* rep_t, int D, const int* lists, const int* counts, int N, int K, float* proto,
                           hipStream_t st) {
  hipLaunchKernelGGL(mb_proto_kernel, dim3((D + 255) / 256, K), dim3(256), 0, st, rep_t, D, lists, counts, N, proto);
  return hipGetLastError();
}

hipError_t launch_mb_enqueue(const float* rep_t, int D, const int* lists, const int* counts, int N, int K, float* bank,
                             int* state, const int* caps, int cap_stride, hipStream_t st) {
  hipLaunchKernelGGL(mb_enqueue_kernel, dim3(32, K), dim3(256), 0, st, rep_t, D, lists, counts, N, bank, state, caps,
                     cap_stride);
  hipLaunchKernelGGL(mb_state_kernel, dim3((K + 63) / 64), dim3(64), 0, st, counts, state, K, caps);
  return hipGetLastError();
}

// dequeue_and_enqueue for ONE class with the keys given directly (m known on the host)
__global__ __launch_bounds__(256) void mb_push_kernel(const float* __restrict__ keys, int m, int D, float* __restrict__ bank_c,
                                                      const int* __restrict__ state_c, int cap) {
  const int rows = state_c[0], head = state_c[1];
  const int j0 = m > cap ? m - cap : 0;
  const long long tot = (long long)(m - j0) * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
    const int j = j0 + (int)(e / D), d = (int)(e - (long long)(j - j0) * D);
    const int slot = (int)(((long long)head + rows + j) % cap);
    bank_c[(size_t)slot * D + d] = keys[(size_t)j * D + d];
  }
}

__global__ void mb_push_state_kernel(int m, int* __restrict__ state_c, int cap) {
  const long long total = (long long)state_c[0] + m;
  const long long dropped = total > cap ? total - cap : 0;
  const int head = state_c[1];
  state_c[0] = (int)(total > cap ? cap : total);
  state_c[1] = (int)(((long long)head + dropped) % cap);
}

hipError_t launch_mb_push(const float* keys, int m, int D, float* bank_c, int* state_c, int cap, hipStream_t st) {
  if (m > 0) hipLaunchKernelGGL(mb_push_kernel, dim3(64), dim3(256), 0, st, keys, m, D, bank_c, state_c, cap);
  hipLaunchKernelGGL(mb_push_state_kernel, dim3(1), dim3(1), 0, st, m, state_c, cap);
  return hipGetLastError();
}

hipError_t launch_mb_infonce(const float* rep, int D, const int* pool, const long long* anchor_draw, const float* pos,
                             long long pos_qstride, const float* bank_c, int cap, int head, const long long* neg_draw,
                             int Qn, int NN, float temp, float scale, float* lossq, float* ganchor, float* drep,
                             hipStream_t st) {
  if (NN + 1 > MB_MAXKEYS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mb_infonce_kernel, dim3(Qn), dim3(256), 0, st, rep, D, pool, anchor_draw, pos, pos_qstride, bank_c,
                     cap, head, neg_draw, NN, temp, scale, Qn, lossq, ganchor);
  if (drep)
    hipLaunchKernelGGL(mb_scatter_kernel, dim3(Qn), dim3(256), (size_t)2 * Qn * sizeof(int), st, ganchor, pool,
                       anchor_draw, Qn, D, drep);
  return hipGetLastError();
}

hipError_t launch_mb_sum(const float* v, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(mb_sum_kernel, dim3(1), dim3(64), 0, st, v, n, out);
  return hipGetLastError();
}

// ---- the whole of compute_unsupervised_loss in ONE launch of ONE workgroup (B <= 8192 rows: what a batch of pixel
// logits is on this path; round 3 took a fill and five launches, 0.1 ms for 4096 x 9 numbers).  1024 threads, up to
// eight rows per thread held in registers from the teacher entropy to the gradient.  The two order statistics of the
// percentile come from a radix select over order-preserving integer keys of the valid entropies (four 8-bit passes,
// histograms in LDS by integer atomics: the VALUE at a rank does not depend on how ties are ordered), not from B x B
// rank counting; every per-row formula is the one of the multi-launch kernels above (same instructions, same values:
// the dropped set is exact either way).
constexpr int US1_T = 1024, US1_R = 8;
__device__ __forceinline__ uint32_t us_key(float e) {
  const uint32_t u = __float_as_uint(e);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float us_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// value of rank `rank` (0-based, ascending) among the valid keys; all threads call it, all get the answer
__device__ __forceinline__ uint32_t us_select(const uint32_t (&key)[US1_R], const bool (&valid)[US1_R], int rank,
                                              int* hist, int* sbin) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t pfx = 0u, msk = 0u;
  int rem = rank;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < US1_R; ++q)
      if (valid[q] && (key[q] & msk) == pfx) atomicAdd(&hist[(key[q] >> shift) & 255u], 1);
    __syncthreads();
    if (tid < 64) {                                    // wave 0: lane l owns bins 4l .. 4l+3
      const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const int mine = (c0 + c1) + (c2 + c3);
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
      const int excl = incl - mine;
      const unsigned long long hit = __ballot(incl > rem);
      const int first = __ffsll((long long)hit) - 1;   // (rank < number of valid keys: some lane qualifies)
      if (lane == first) {
        int r = rem - excl, b = 4 * lane;
        if (r >= c0) { r -= c0; ++b; if (r >= c1) { r -= c1; ++b; if (r >= c2) { r -= c2; ++b; } } }
        sbin[0] = b; sbin[1] = r;
      }
    }
    __syncthreads();
    pfx |= (uint32_t)sbin[0] << shift; msk |= 255u << shift; rem = sbin[1];
    __syncthreads();                                   // sbin / hist are rewritten by the next pass
  }
  return pfx;
}

// a row of K <= 16 logits in registers: every load of the row requested at once (with K a run-time number the per-k
// loops re-load the row for every pass and the passes become chains of dependent L1 round trips)
constexpr int US1_K = 16;
struct UsRow { float v[US1_K]; };
__device__ __forceinline__ UsRow us_load_row(const float* __restrict__ p, int K) {
  UsRow r;
#pragma unroll
  for (int k = 0; k < US1_K; ++k) r.v[k] = p[k < K ? k : 0];
  return r;
}
// max, sum of exp(. - max) over the K real entries (same operation order as the per-k loops of the kernels above)
__device__ __forceinline__ void us_max_se(const UsRow& r, int K, float& mx, float& se) {
  mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < US1_K; ++k) if (k < K) mx = fmaxf(mx, r.v[k]);
  se = 0.f;
#pragma unroll
  for (int k = 0; k < US1_K; ++k) if (k < K) se += expf(r.v[k] - mx);
}

template <bool REGROWS>
__global__ __launch_bounds__(US1_T) void us_onewg_kernel(const float* __restrict__ predict, long long* __restrict__ target,
                                                         const float* __restrict__ teacher, int B, int K, double percent,
                                                         float* __restrict__ loss, float* __restrict__ dpredict) {
  __shared__ int hist[256];
  __shared__ int sbin[2];
  __shared__ int scnt[3];
  __shared__ uint32_t smin[US1_T / 64];
  __shared__ float sred[US1_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ent[US1_R];
  uint32_t key[US1_R];
  bool valid[US1_R];
  long long tg[US1_R];
  if (tid < 3) scnt[tid] = 0;
  __syncthreads();
  int nv = 0;
#pragma unroll
  for (int q = 0; q < US1_R; ++q) {
    const int i = tid + US1_T * q;
    ent[q] = 0.f; valid[q] = false; tg[q] = 255;
    if (i < B) {
      const float* t = teacher + (size_t)i * K;
      float e = 0.f;
      if (REGROWS) {
        const UsRow r = us_load_row(t, K);
        tg[q] = target[i];
        float mx, se;
        us_max_se(r, K, mx, se);
#pragma unroll
        for (int k = 0; k < US1_K; ++k) if (k < K) { const float p = expf(r.v[k] - mx) / se; e -= p * logf(p + 1e-10f); }
      } else {
        float mx = -3.0e38f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, t[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(t[k] - mx);
        for (int k = 0; k < K; ++k) { const float p = expf(t[k] - mx) / se; e -= p * logf(p + 1e-10f); }
        tg[q] = target[i];
      }
      ent[q] = e;
      valid[q] = tg[q] != 255;
      nv += valid[q] ? 1 : 0;
    }
    key[q] = us_key(ent[q]);
  }
  if (nv) atomicAdd(&scnt[0], nv);
  __syncthreads();
  const int n = scnt[0];
  float thr = 3.0e38f;
  if (n > 0) {                                          // uniform
    const double vidx = (double)(n - 1) * percent / 100.0;
    const int lo = (int)floor(vidx), hi = lo + 1 < n ? lo + 1 : n - 1;
    const uint32_t klo = us_select(key, valid, lo, hist, sbin);
    // a[hi] without a second select: it equals a[lo] when more than lo + 1 valid keys are <= a[lo], else it is the
    // smallest valid key above a[lo]
    uint32_t khi = klo;
    if (hi != lo) {
      int le = 0;
      uint32_t mn = 0xffffffffu;
#pragma unroll
      for (int q = 0; q < US1_R; ++q)
        if (valid[q]) { if (key[q] <= klo) ++le; else mn = key[q] < mn ? key[q] : mn; }
      if (le) atomicAdd(&scnt[2], le);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)mn, o, 64); mn = v < mn ? v : mn; }
      if (lane == 0) smin[wave] = mn;
      __syncthreads();
      if (scnt[2] <= hi) {
        khi = 0xffffffffu;
        for (int w = 0; w < US1_T / 64; ++w) khi = smin[w] < khi ? smin[w] : khi;
      }
    }
    const float vlo = us_unkey(klo), vhi = us_unkey(khi);
    const double g = vidx - floor(vidx);
    const double a = (double)vlo, b = (double)vhi, diff = b - a;
    double t = a + diff * g;
    if (g >= 0.5) t = b - diff * (1.0 - g);
    if (diff == 0.0) t = a;
    thr = (float)t;
  }
  float part = 0.f;
  int nk = 0;
#pragma unroll
  for (int q = 0; q < US1_R; ++q) {
    const int i = tid + US1_T * q;
    if (i < B) {
      if (tg[q] != 255 && ent[q] >= thr) { tg[q] = 255; target[i] = 255; }
      if (tg[q] != 255) {
        const float* p = predict + (size_t)i * K;
        float mx, se, pt;
        if (REGROWS) {
          const UsRow r = us_load_row(p, K);
          us_max_se(r, K, mx, se);
          pt = 0.f;
#pragma unroll
          for (int k = 0; k < US1_K; ++k) if (k == (int)tg[q]) pt = r.v[k];
        } else {
          mx = -3.0e38f;
          for (int k = 0; k < K; ++k) mx = fmaxf(mx, p[k]);
          se = 0.f;
          for (int k = 0; k < K; ++k) se += expf(p[k] - mx);
          pt = p[tg[q]];
        }
        part += mx + logf(se) - pt;
        ++nk;
      }
    }
  }
  if (nk) atomicAdd(&scnt[1], nk);
  part = wave_sum(part);                                // fixed order: rows by q, lanes by the xor tree, waves in order
  if (lane == 0) sred[wave] = part;
  __syncthreads();
  const float kf = (float)scnt[1];
  const float weight = (float)B / kf;                   // :256 (inf / NaN when nothing is kept, like the reference)
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < US1_T / 64; ++w) t += sred[w];
    loss[0] = weight * (t / kf);
  }
  const float sc = weight / kf;
#pragma unroll
  for (int q = 0; q < US1_R; ++q) {
    const int i = tid + US1_T * q;
    if (i < B) {
      const float* p = predict + (size_t)i * K;
      float* g = dpredict + (size_t)i * K;
      if (tg[q] == 255) {
        for (int k = 0; k < K; ++k) g[k] = 0.f;
      } else if (REGROWS) {
        const UsRow r = us_load_row(p, K);
        float mx, se;
        us_max_se(r, K, mx, se);
#pragma unroll
        for (int k = 0; k < US1_K; ++k)
          if (k < K) g[k] = sc * (expf(r.v[k] - mx) / se - (k == (int)tg[q] ? 1.f : 0.f));
      } else {
        float mx = -3.0e38f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, p[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(p[k] - mx);
        for (int k = 0; k < K; ++k) g[k] = sc * (expf(p[k] - mx) / se - (k == (int)tg[q] ? 1.f : 0.f));
      }
    }
  }
}

// ---- three launches for 1024 < B <= 8192 rows (one workgroup alone is compute-bound there: 4096 x 9 logits cost one
// CU 32 us of exp / log / divide issue): row entropies on the grid, the radix select + threshold + kept count in ONE
// workgroup (nothing but keys: 10 us), then drop / CE / gradient on the grid with the loss summed by the last
// workgroup to arrive (integer ticket, block partials in index order: deterministic, no float atomics).
struct UsMid { float thr; int kept; int ticket; int pad; };

__global__ __launch_bounds__(256) void us_entropy2_kernel(const float* __restrict__ teacher, int B, int K,
                                                          float* __restrict__ ent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const float* t = teacher + (size_t)i * K;
  float e = 0.f;
  if (K <= US1_K) {
    const UsRow r = us_load_row(t, K);
    float mx, se;
    us_max_se(r, K, mx, se);
#pragma unroll
    for (int k = 0; k < US1_K; ++k) if (k < K) { const float p = expf(r.v[k] - mx) / se; e -= p * logf(p + 1e-10f); }
  } else {
    float mx = -3.0e38f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, t[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(t[k] - mx);
    for (int k = 0; k < K; ++k) { const float p = expf(t[k] - mx) / se; e -= p * logf(p + 1e-10f); }
  }
  ent[i] = e;
}

__global__ __launch_bounds__(US1_T) void us_select1_kernel(const float* __restrict__ ent, const long long* __restrict__ target,
                                                           int B, double percent, UsMid* __restrict__ mid) {
  __shared__ int hist[256];
  __shared__ int sbin[2];
  __shared__ int scnt[3];
  __shared__ uint32_t smin[US1_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ev[US1_R];
  uint32_t key[US1_R];
  bool valid[US1_R];
  if (tid < 3) scnt[tid] = 0;
  __syncthreads();
  int nv = 0;
#pragma unroll
  for (int q = 0; q < US1_R; ++q) {                      // every load of the thread's rows first
    const int i = tid + US1_T * q;
    ev[q] = ent[i < B ? i : 0];
    valid[q] = (i < B) && target[i < B ? i : 0] != 255;
  }
#pragma unroll
  for (int q = 0; q < US1_R; ++q) { key[q] = us_key(ev[q]); nv += valid[q] ? 1 : 0; }
  if (nv) atomicAdd(&scnt[0], nv);
  __syncthreads();
  const int n = scnt[0];
  float thr = 3.0e38f;
  if (n > 0) {                                          // uniform
    const double vidx = (double)(n - 1) * percent / 100.0;
    const int lo = (int)floor(vidx), hi = lo + 1 < n ? lo + 1 : n - 1;
    const uint32_t klo = us_select(key, valid, lo, hist, sbin);
    uint32_t khi = klo;
    if (hi != lo) {                                     // a[hi]: a[lo] again, or the smallest key above it
      int le = 0;
      uint32_t mn = 0xffffffffu;
#pragma unroll
      for (int q = 0; q < US1_R; ++q)
        if (valid[q]) { if (key[q] <= klo) ++le; else mn = key[q] < mn ? key[q] : mn; }
      if (le) atomicAdd(&scnt[2], le);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)mn, o, 64); mn = v < mn ? v : mn; }
      if (lane == 0) smin[wave] = mn;
      __syncthreads();
      if (scnt[2] <= hi) {
        khi = 0xffffffffu;
        for (int w = 0; w < US1_T / 64; ++w) khi = smin[w] < khi ? smin[w] : khi;
      }
    }
    const float vlo = us_unkey(klo), vhi = us_unkey(khi);
    const double g = vidx - floor(vidx);
    const double a = (double)vlo, b = (double)vhi, diff = b - a;
    double t = a + diff * g;
    if (g >= 0.5) t = b - diff * (1.0 - g);
    if (diff == 0.0) t = a;
    thr = (float)t;
  }
  int nk = 0;
#pragma unroll
  for (int q = 0; q < US1_R; ++q) nk += (valid[q] && !(ev[q] >= thr)) ? 1 : 0;     // rows that stay
  if (nk) atomicAdd(&scnt[1], nk);
  __syncthreads();
  if (tid == 0) { mid->thr = thr; mid->kept = scnt[1]; mid->ticket = 0; mid->pad = 0; }
}

__global__ __launch_bounds__(256) void us_apply_kernel(const float* __restrict__ predict, long long* __restrict__ target,
                                                       const float* __restrict__ ent, int B, int K, UsMid* __restrict__ mid,
                                                       float* __restrict__ bpart, float* __restrict__ loss,
                                                       float* __restrict__ dpredict) {
  __shared__ float sred[4];
  __shared__ int slast;
  const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  const float thr = mid->thr, kf = (float)mid->kept;
  const float weight = (float)B / kf;                   // :256 (inf / NaN when nothing is kept, like the reference)
  const float sc = weight / kf;
  float l = 0.f;
  if (i < B) {
    long long tg = target[i];
    if (tg != 255 && ent[i] >= thr) { tg = 255; target[i] = 255; }
    const float* p = predict + (size_t)i * K;
    float* g = dpredict + (size_t)i * K;
    if (tg == 255) {
      for (int k = 0; k < K; ++k) g[k] = 0.f;
    } else if (K <= US1_K) {
      const UsRow r = us_load_row(p, K);
      float mx, se;
      us_max_se(r, K, mx, se);
      float pt = 0.f;
#pragma unroll
      for (int k = 0; k < US1_K; ++k) if (k == (int)tg) pt = r.v[k];
      l = mx + logf(se) - pt;
#pragma unroll
      for (int k = 0; k < US1_K; ++k)
        if (k < K) g[k] = sc * (expf(r.v[k] - mx) / se - (k == (int)tg ? 1.f : 0.f));
    } else {
      float mx = -3.0e38f;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, p[k]);
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += expf(p[k] - mx);
      l = mx + logf(se) - p[tg];
      for (int k = 0; k < K; ++k) g[k] = sc * (expf(p[k] - mx) / se - (k == (int)tg ? 1.f : 0.f));
    }
  }
  l = wave_sum(l);
  if ((tid & 63) == 0) sred[tid >> 6] = l;
  __syncthreads();
  if (tid == 0) {
    // publish this workgroup's partial (release at agent scope; the explicit wait keeps the compiler from letting the
    // ticket overtake the write-back: MI355X guide, "compiler hazard"), take a ticket; the last arriver sums
    bpart[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = atomicAdd(&mid->ticket, 1);
    slast = (t == (int)gridDim.x - 1) ? 1 : 0;
    if (slast) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float tot = 0.f;
      for (int b = 0; b < (int)gridDim.x; ++b) tot += __builtin_nontemporal_load(bpart + b);   // index order
      loss[0] = weight * (tot / kf);
    }
  }
}

size_t unsup_ws_bytes(int B) { return ((size_t)3 * B + 16) * 4; }

hipError_t launch_unsup(const float* predict, long long* target, const float* teacher, int B, int K, double percent,
                        float* loss, float* dpredict, void* ws, hipStream_t st) {
  const bool onewg_off = switches().unsup_onewg == 0;
  const bool three_off = switches().unsup_3l == 0;
  if (B > US1_T && B <= US1_T * US1_R && !onewg_off && !three_off) {
    float* ent3 = (float*)ws;                          // [B]
    float* bpart = ent3 + B;                            // [blocks]
    UsMid* mid = (UsMid*)(bpart + B);                   // (16-byte aligned: the workspace is, B floats twice)
    const int nb3 = (B + 255) / 256;
    hipLaunchKernelGGL(us_entropy2_kernel, dim3(nb3), dim3(256), 0, st, teacher, B, K, ent3);
    hipLaunchKernelGGL(us_select1_kernel, dim3(1), dim3(US1_T), 0, st, ent3, target, B, percent, mid);
    hipLaunchKernelGGL(us_apply_kernel, dim3(nb3), dim3(256), 0, st, predict, target, ent3, B, K, mid, bpart, loss, dpredict);
    return hipGetLastError();
  }
  if (B <= US1_T * US1_R && !onewg_off) {
    if (K <= US1_K) hipLaunchKernelGGL(us_onewg_kernel<true>, dim3(1), dim3(US1_T), 0, st, predict, target, teacher, B, K, percent, loss, dpredict);
    else            hipLaunchKernelGGL(us_onewg_kernel<false>, dim3(1), dim3(US1_T), 0, st, predict, target, teacher, B, K, percent, loss, dpredict);
    return hipGetLastError();
  }
  float* ent = (float*)ws;
  float* rowloss = ent + B;
  float* sel = rowloss + B;           // [2]
  int* info = (int*)(sel + 2);        // nvalid, kept
  int* rank = (int*)(sel + 16);       // [B]
  hipError_t e = hipMemsetAsync(sel, 0, ((size_t)B + 16) * 4, st);    // sel, info, rank
  if (e != hipSuccess) return e;
  const int nb = (B + 255) / 256;
  hipLaunchKernelGGL(us_entropy_kernel, dim3(nb), dim3(256), 0, st, teacher, target, B, K, ent, info);
  hipLaunchKernelGGL(us_rank_kernel, dim3(nb, US_SLICES), dim3(256), 0, st, ent, target, B, rank);
  hipLaunchKernelGGL(us_select_kernel, dim3(nb), dim3(256), 0, st, ent, target, B, percent, info, rank, sel);
  hipLaunchKernelGGL(us_mask_kernel, dim3(nb), dim3(256), 0, st, predict, target, ent, B, K, percent, info, sel,
                     rowloss, info + 1);
  hipLaunchKernelGGL(us_grad_kernel, dim3(nb), dim3(256), 0, st, predict, target, rowloss, B, K, info + 1, loss, dpredict);
  return hipGetLastError();
}

}  // namespace cmlpl
