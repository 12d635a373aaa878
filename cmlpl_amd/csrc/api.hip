// C-ABI entry points (include/cmlpl.h): argument checking, workspace carving and the launch
// sequence of one training step.  No allocation, no synchronisation, graph-capturable.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/cmlpl.h"
#include "kernels.hpp"

using namespace cmlpl;

namespace {

inline int64_t up4(int64_t v) { return (v + 3) & ~(int64_t)3; }
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Dims {
  int C, H, W, bands, K, HW, H2, W2, P2, H4, W4, P4, SF, F;
};

bool make_dims(const cmlpl_shape* s, Dims* d) {
  if (!s) return false;
  d->C = s->C; d->H = s->H; d->W = s->W; d->bands = s->bands; d->K = s->K;
  if (d->C < 1 || d->C > 256 || d->H < 4 || d->W < 4 || d->bands < 1 || d->K < 1 || d->K > 64) return false;
  d->HW = d->H * d->W; d->H2 = d->H / 2; d->W2 = d->W / 2; d->P2 = d->H2 * d->W2;
  d->H4 = d->H2 / 2; d->W4 = d->W2 / 2; d->P4 = d->H4 * d->W4;
  d->SF = 64 * d->P4; d->F = d->SF + 1024;
  return d->P4 >= 1;
}

// workspace of the network forward/backward (all sizes in bytes, 256-B aligned regions)
struct NetWs {
  float *a0, *p1, *p2, *y, *ynorm, *catd, *dropgen, *dy, *dp2, *dp1, *da0, *part1, *part2, *part0;
  uint8_t *m1, *m2;
  uint32_t* hstat;      // [4 kinds][2 networks][n] per-sample maxima the fused kernels leave for the two-piece weight gradient
  size_t bytes;
};

// do ALL FOUR general 3x3 launches of a step (conv1 / conv2, forward / data gradient) take two-piece kernels?  Then each
// leaves its samples' image maxima in the statistics table and the weight-gradient pair launch can scale by them.
bool general_h2_stats(const Dims& d, int rows) {
  return !conv3_fused_ok(d.H, d.W, d.C, rows) && !conv3_fused_bwd_ok(d.H, d.W, d.C, rows) &&
         conv3_h2x_general(0, d.H, d.W, rows) && conv3_h2x_general(1, d.H, d.W, rows) &&
         conv3_h2x_general(0, d.H2, d.W2, rows) && conv3_h2x_general(1, d.H2, d.W2, rows);
}

// per-net conv0 weight-gradient partials: one per sample when the fused data-gradient kernel produces them
int conv0_partials(const Dims& d, int nets, int n) {
  return conv3_fused_bwd_ok(d.H, d.W, d.C, nets * n) ? n : plan_conv0_wgrad_G(n, d.C, d.HW);
}

bool carve_net(const Dims& d, int nets, int n, char* base, NetWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += up256(bytes); return p; };
  const size_t N = (size_t)nets * n;
  Wgrad3Plan w1, w2;
  bool wpair = false;
  if (!plan_wgrad3_both(nets, n, d.H, d.W, d.H2, d.W2, true, &w1, &w2, &wpair)) return false;
  Conv3Plan c;
  if (!plan_conv3(0, d.H, d.W, nets * n, &c) || !plan_conv3(1, d.H, d.W, nets * n, &c) ||
      !plan_conv3(0, d.H2, d.W2, nets * n, &c) || !plan_conv3(1, d.H2, d.W2, nets * n, &c)) return false;
  const int G0 = conv0_partials(d, nets, n);
  w->a0 = (float*)take(N * d.HW * 64 * 4);
  w->p1 = (float*)take(N * d.P2 * 64 * 4);
  w->m1 = (uint8_t*)take(N * d.P2 * 64);
  w->p2 = (float*)take(N * d.P4 * 64 * 4);
  w->m2 = (uint8_t*)take(N * d.P4 * 64);
  w->y = (float*)take(N * 1024 * 4);
  w->ynorm = (float*)take(N * 4);
  w->catd = (float*)take(N * d.F * 4);
  w->dropgen = (float*)take(N * d.F * 4);
  w->dy = (float*)take(N * 1024 * 4);
  w->dp2 = (float*)take(N * d.P4 * 64 * 4);
  w->dp1 = (float*)take(N * d.P2 * 64 * 4);
  w->da0 = (float*)take(N * d.HW * 64 * 4);
  w->part1 = (float*)take((size_t)nets * w1.G * PART3 * 4);
  w->part2 = (float*)take((size_t)nets * w2.G * PART3 * 4);
  w->part0 = (float*)take((size_t)nets * G0 * (size_t)conv0_partial_size(d.C) * 4);
  w->hstat = (uint32_t*)take((size_t)4 * 2 * n * 4);
  w->bytes = off;
  return true;
}

// extra regions used only by cmlpl_train_step / cmlpl_loss_fwd_bwd
struct StepWs {
  float *xn, *sn, *dlogits, *dfeat, *probs, *loss;
  size_t bytes;
};

void carve_step(const Dims& d, int n, int bank_rows, char* base, StepWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += up256(bytes); return p; };
  w->xn = (float*)take((size_t)2 * n * d.C * d.HW * 4);
  w->sn = (float*)take((size_t)2 * n * d.bands * 4);
  w->dlogits = (float*)take((size_t)2 * n * d.K * 4);
  w->dfeat = (float*)take((size_t)2 * n * 1024 * 4);
  w->probs = (float*)take((size_t)4 * n * d.K * 4);
  w->loss = (float*)take(loss_ws_floats(n, n, n, d.K, bank_rows > n ? bank_rows : n) * 4);
  w->bytes = off;
}

int chk(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

PackInfo make_pack_info(const Dims& d, const cmlpl_layout_t& L) {
  PackInfo pi;
  pi.stride = L.packed_total; pi.off_w0 = L.param_off[0]; pi.off_w1 = L.param_off[2]; pi.off_w2 = L.param_off[4];
  pi.off_ws = L.param_off[6]; pi.C = d.C; pi.bands = d.bands;
  return pi;
}

// ---- optional per-kernel timing (cmlpl_timing_begin/_end): hipEvent pairs recorded on the launch stream
// around the selected launches.  Off by default; the only process-global state of the library.
struct Timing {
  bool on = false;
  uint32_t mask = 0;
  std::vector<hipEvent_t> ev;
  std::vector<int> ids;
  size_t used = 0;
} g_timing;

template <class F>
int timed(int id, hipStream_t st, F f) {
  Timing& t = g_timing;
  if (t.on && ((t.mask >> id) & 1u) && t.used + 2 <= t.ev.size()) {
    (void)hipEventRecord(t.ev[t.used], st);
    const int rc = f();
    (void)hipEventRecord(t.ev[t.used + 1], st);
    t.ids.push_back(id);
    t.used += 2;
    return rc;
  }
  return f();
}
#define TIMED(id, expr) timed((id), st, [&]() -> int { return (expr); })

}  // namespace

extern "C" {

int cmlpl_abi_version(void) { return CMLPL_ABI_VERSION; }

// hash of the kernel sources + C header this binary was built from (cmlpl_amd/build_ext.py passes it; the marker lets
// the build script read it from the file without loading the library): the Python binding refuses a stale binary
#ifndef CMLPL_SOURCE_HASH
#define CMLPL_SOURCE_HASH "unknown"
#endif
const char* cmlpl_source_hash(void) {
  static const char marked[] = "CMLPL_SOURCE_HASH=" CMLPL_SOURCE_HASH;
  return marked + sizeof("CMLPL_SOURCE_HASH=") - 1;
}

int cmlpl_layout(const cmlpl_shape* shape, cmlpl_layout_t* out) {
  Dims d;
  if (!out) return CMLPL_E_ARG;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  const int64_t numel[CMLPL_NUM_TENSORS] = {
      64LL * d.C, 64, 36864, 64, 36864, 64, 1024LL * d.bands, 1024, (int64_t)d.K * d.F, d.K,
      256LL * 1024, 256, 64LL * 1024, 64, 64LL * 256, 64};
  int64_t off = 0;
  for (int i = 0; i < CMLPL_NUM_TENSORS; ++i) {
    out->param_off[i] = off;
    out->param_numel[i] = numel[i];
    off = up4(off + numel[i]);
    if (i == CMLPL_NUM_LIVE - 1) out->param_live = off;
  }
  out->param_total = off;
  out->packed_total = pack_total(d.C, d.bands);
  out->cls_in = d.F;
  out->reserved = 0;
  return 0;
}

size_t cmlpl_workspace_bytes(const cmlpl_shape* shape, int nets, int n, int bank_rows) {
  Dims d;
  if (!make_dims(shape, &d) || nets < 1 || nets > 2 || n < 1) return 0;
  NetWs nw;
  if (!carve_net(d, nets, n, nullptr, &nw)) return 0;
  StepWs sw;
  carve_step(d, n, bank_rows, nullptr, &sw);
  return nw.bytes + sw.bytes + 256;
}

int cmlpl_pack_weights(const cmlpl_shape* shape, int nets, const float* d_params, int64_t param_stride,
                       float* d_packed, void* stream) {
  cmlpl_layout_t L;
  Dims d;
  int rc = cmlpl_layout(shape, &L);
  if (rc) return rc;
  make_dims(shape, &d);
  if (!d_params || !d_packed || nets < 1 || nets > 2) return CMLPL_E_ARG;
  return chk(launch_pack_weights(nets, d_params, param_stride, make_pack_info(d, L), d_packed, (hipStream_t)stream));
}

int cmlpl_augment(const cmlpl_shape* shape, int nets, int bt, int btu, const float* d_xpl, const float* d_xl,
                  const float* d_xpu, const float* d_xu, const float* const* noise8, float sigma, uint64_t seed,
                  uint64_t step, const cmlpl_shard* shard, float* d_xn, float* d_sn, float* d_snT, void* stream) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  if (nets < 1 || nets > 2 || bt < 0 || btu < 0 || bt + btu < 1 || !d_xn || !d_sn) return CMLPL_E_ARG;
  if ((bt > 0 && (!d_xpl || !d_xl)) || (btu > 0 && (!d_xpu || !d_xu))) return CMLPL_E_ARG;
  if (shard && (shard->nlab != bt || shard->nunl != btu)) return CMLPL_E_ARG;
  const int lab0 = shard ? shard->lab0 : 0, unl_base = shard ? shard->bt_g + shard->unl0 : bt;
  hipStream_t st = (hipStream_t)stream;
  return TIMED(CMLPL_K_AUGMENT, chk(launch_augment(3, nets, bt, btu, d.C * d.HW, d.bands, lab0, unl_base, d_xpl, d_xl,
                            d_xpu, d_xu, noise8, sigma, seed, step, d_xn, d_sn, d_snT, st)));
}

namespace {
// patch rows that are already augmented: [nets][n][C*HW]
XSrc xsrc_plain(const float* d_xn, int nets, int n, long long per, uint64_t seed, uint64_t step,
                const cmlpl_shard* sh, DynRef dyn = DynRef()) {
  XSrc x = XSrc();
  x.sel.dyn = dyn;
  const int nlab = sh ? sh->nlab : n;
  for (int i = 0; i < 2; ++i) {
    x.lab[i] = d_xn + (long long)(i < nets ? i : 0) * n * per;
    x.unl[i] = x.lab[i] + (long long)nlab * per;
  }
  x.nlab = nlab; x.sigma = 0.f;
  x.lab0 = sh ? sh->lab0 : 0; x.unl_base = sh ? sh->bt_g + sh->unl0 : n;   // keys of the dropout stream
  x.seed = seed; x.step = step;
  return x;
}
// raw labelled / unlabelled rows + noise formed in the kernels
RowSel batch_sel(const cmlpl_batch* b, DynRef dyn) {
  RowSel s = RowSel();
  s.lab_idx = (const long long*)b->d_lab_idx; s.unl_idx = (const long long*)b->d_unl_idx; s.dyn = dyn;
  return s;
}
XSrc xsrc_raw(const cmlpl_batch* b, float sigma, uint64_t seed, uint64_t step, const cmlpl_shard* sh,
              DynRef dyn = DynRef()) {
  XSrc x = XSrc();
  x.sel = batch_sel(b, dyn);
  for (int i = 0; i < 2; ++i) {
    x.lab[i] = b->d_xpl; x.unl[i] = b->d_xpu;
    x.nz_lab[i] = b->noise8 ? b->noise8[2 * i] : nullptr;          // reference draw order, see cmlpl_augment
    x.nz_unl[i] = b->noise8 ? b->noise8[4 + 2 * i] : nullptr;
  }
  x.sigma = sigma; x.nlab = b->bt; x.philox = b->noise8 == nullptr;
  x.lab0 = sh ? sh->lab0 : 0; x.unl_base = sh ? sh->bt_g + sh->unl0 : b->bt;
  x.seed = seed; x.step = step;
  return x;
}
// parts: bit 0 = the spectral branch (feat_spe + ReLU; with early_feat also the L2 normalisation -> d_feat, ynorm),
//        bit 1 = the convolution stack + head (-> d_logits; d_feat / ynorm too unless early_feat)
int fwd_core(const Dims& d, const cmlpl_layout_t& L, int nets, int n, const float* d_params, int64_t param_stride,
             const float* d_packed, const XSrc& xs, const XSrc* xspec, float* d_sn_out, const long long* d_labels, float* d_labels_f, const float* d_xn,
             const float* d_sn, const float* d_snT,
             const float* d_dropmask, float dropout_p, int train, uint64_t seed, uint64_t step,
             const cmlpl_shard* shard, float* d_logits, float* d_feat, const NetWs& w, hipStream_t st,
             float* xn_save = nullptr, int parts = 3, bool early_feat = false);
// parts: bit 0 = the data-gradient chain + the 3x3 weight-gradient partials (needs d_dlogits; d_dfeat only when parts == 3),
//        bit 1 = partial reduction + classifier / feat_spe weight gradients (parts == 2: dy first takes its d_dfeat part)
int bwd_core(const Dims& d, const cmlpl_layout_t& L, int nets, int n, const float* d_params, int64_t param_stride,
             const float* d_packed, const XSrc& xs, const float* d_xn, const float* d_sn, const float* d_dropmask,
             float dropout_p, int train, const float* d_dlogits, const float* d_dfeat, float* d_grads,
             int64_t grad_stride, const NetWs& w, hipStream_t st, int* dyn_cursor = nullptr, cmlpl_dyn* dyn_table = nullptr,
             int parts = 3);
}  // namespace

int cmlpl_basenet2_fwd(const cmlpl_shape* shape, int nets, int n, const float* d_params, int64_t param_stride,
                       const float* d_packed, const float* d_xn, const float* d_sn, const float* d_snT,
                       const float* d_dropmask, float dropout_p, int train, uint64_t seed, uint64_t step,
                       const cmlpl_shard* shard,
                       float* d_logits, float* d_feat, void* d_workspace, size_t workspace_bytes, void* stream) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  if (nets < 1 || nets > 2 || n < 1 || !d_params || !d_packed || !d_xn || !d_sn || !d_logits || !d_feat ||
      !d_workspace)
    return CMLPL_E_ARG;
  if (dropout_p < 0.f || dropout_p >= 1.f) return CMLPL_E_ARG;
  NetWs w;
  if (!carve_net(d, nets, n, (char*)d_workspace, &w)) return CMLPL_E_SHAPE;
  if (w.bytes > workspace_bytes) return CMLPL_E_WORKSPACE;
  return fwd_core(d, L, nets, n, d_params, param_stride, d_packed,
                  xsrc_plain(d_xn, nets, n, (long long)d.C * d.HW, seed, step, shard), nullptr, nullptr, nullptr, nullptr, d_xn, d_sn, d_snT, d_dropmask, dropout_p, train, seed, step, shard, d_logits, d_feat, w, (hipStream_t)stream);
}

namespace {
int fwd_core(const Dims& d, const cmlpl_layout_t& L, int nets, int n, const float* d_params, int64_t param_stride,
             const float* d_packed, const XSrc& xs, const XSrc* xspec, float* d_sn_out, const long long* d_labels, float* d_labels_f, const float* d_xn,
             const float* d_sn, const float* d_snT,
             const float* d_dropmask, float dropout_p, int train, uint64_t seed, uint64_t step,
             const cmlpl_shard* shard, float* d_logits, float* d_feat, const NetWs& w, hipStream_t st,
             float* xn_save, int parts, bool early_feat) {
  int rc;
  const long long pk_ns = L.packed_total;
  if (parts & 1) {  // spectral branch (feat_spe + ReLU)
    if (xspec != nullptr) {
      // raw spectra: augmentation + GEMM + bias + ReLU in one launch (also writes the augmented rows for the
      // weight gradient)
      if ((rc = TIMED(CMLPL_K_SPE_FWD, chk(launch_spe_fused(nets, n, d.bands, *xspec, d_params + L.param_off[6],
                                   d_params + L.param_off[7], param_stride, w.y, d_sn_out, d_labels, d_labels_f,
                                   xspec->nlab, st))))) return rc;
    } else {
      // pre-augmented spectra (the nn.Module path): y = relu(sn . W^T + b) from the canonical weight
      if ((rc = TIMED(CMLPL_K_SPE_FWD, chk(launch_spe_fwd(nets, n, d.bands, d_sn, d_params + L.param_off[6],
                                   d_params + L.param_off[7], param_stride, w.y, st))))) return rc;
    }
    // the embeddings right here (they depend on nothing the convolutions compute, tools/models.py:142-146)
    // (timed with the spectral branch it belongs to)
    if (early_feat && (rc = TIMED(CMLPL_K_SPE_FWD, chk(launch_feat_norm(nets, n, w.y, w.ynorm, d_feat, (long long)n * 1024, st)))))
      return rc;
  }
  if (!(parts & 2)) return 0;
  if (early_feat) d_feat = nullptr;           // the head skips its own normalisation
  if (shard && shard->nlab + shard->nunl != n) return CMLPL_E_ARG;
  if (conv3_fused_tail_ok(d.H, d.W, d.C, nets * n, d.K)) {
    // the whole spatial forward + head of a sample in one workgroup: conv0 + conv1 + pool + conv2 + pool + flatten /
    // concat / dropout / classifier / L2-norm.  Needs the spectral branch output (launched above, same stream).
    FwdTail t;
    t.w2f = d_packed + pack_off_b3(d.C, d.bands, 2); t.w2f_ns = pk_ns; t.b2 = d_params + L.param_off[5];
    t.wc = d_params + L.param_off[8]; t.bc = d_params + L.param_off[9]; t.p_ns = param_stride;
    t.y = w.y; t.dropmask = d_dropmask; t.dropgen = w.dropgen; t.catd = w.catd; t.ynorm = w.ynorm;
    t.logits = d_logits; t.feat = d_feat; t.p2 = w.p2; t.m2 = w.m2; t.dropout_p = dropout_p; t.train = train; t.K = d.K;
    t.w1h = d_packed + pack_off_h2(d.C, d.bands, 0); t.w1h_ns = pk_ns; t.h2flag = (const uint32_t*)(d_packed + pack_off_h2flag(d.C, d.bands));
    // every sample's largest |a0| / |p1| (this launch) and gradient operands (the backward's), for the two-piece
    // weight-gradient launch of the same step
    if (conv3_h2x_both(d.H, d.W, d.C, nets * n, d.K)) t.hstat = w.hstat;
    return TIMED(CMLPL_K_CONV1_FWD, chk(launch_conv3_fused(nets, n, d.C, d.H, d.W, xs, d_packed + pack_off_w0b3(d.C, d.bands),
                               pk_ns, d_params + L.param_off[1], param_stride, w.a0, d_packed + pack_off_b3(d.C, d.bands, 0), pk_ns,
                               d_params + L.param_off[3], param_stride, w.p1, w.m1, &t, st, xn_save)));
  }
  // the general 3x3 launches on two fp16 pieces where their plans allow; when all four of a step do, each leaves its
  // samples' image maxima for the two-piece weight gradient (general_h2_stats)
  const uint32_t* h2flag = (const uint32_t*)(d_packed + pack_off_h2flag(d.C, d.bands));
  const bool gen_stats = general_h2_stats(d, nets * n);
  if (conv3_fused_ok(d.H, d.W, d.C, nets * n)) {
    // conv0 + conv1 in one launch, input rows taken where they lie and augmented in LDS: neither an augmented
    // copy of the input nor a0's round trip between the two convolutions touches HBM (a0 is still written once,
    // for the backward pass)
    if ((rc = TIMED(CMLPL_K_CONV1_FWD, chk(launch_conv3_fused(nets, n, d.C, d.H, d.W, xs, d_packed + pack_off_w0b3(d.C, d.bands),
                               pk_ns, d_params + L.param_off[1], param_stride, w.a0, d_packed + pack_off_b3(d.C, d.bands, 0), pk_ns,
                               d_params + L.param_off[3], param_stride, w.p1, w.m1, nullptr, st, xn_save))))) return rc;
  } else {
    if (conv0a_ok(d.C, d.HW) && (d_xn == nullptr || xs.sigma == 0.f)) {
      // the general path: augmentation + conv0 in ONE launch on the split-bf16 MFMA, from the rows `xs` names (raw rows
      // with their noise -> the augmented rows also go to xn_save for the backward; or a caller's pre-augmented tensor)
      if ((rc = TIMED(CMLPL_K_CONV0_FWD, chk(launch_conv0a_fwd(nets, n, d.C, d.HW, xs, d_packed + pack_off_w0b3(d.C, d.bands), pk_ns,
                                     d_params + L.param_off[1], param_stride, w.a0, xn_save, st))))) return rc;
    } else {
      if (!d_xn) return CMLPL_E_ARG;
      if ((rc = TIMED(CMLPL_K_CONV0_FWD, chk(launch_conv0_fwd(nets, n, d.C, d.HW, d_xn, d_packed + pack_off_w0t(), pk_ns,
                                     d_params + L.param_off[1], param_stride, w.a0, st))))) return rc;
    }
    const Conv3H2 h1 = {d_packed + pack_off_h2(d.C, d.bands, 0), pk_ns, h2flag, gen_stats ? w.hstat : nullptr, 0};
    if ((rc = TIMED(CMLPL_K_CONV1_FWD, chk(launch_conv3(0, nets, n, d.H, d.W, w.a0, nullptr, d_packed + pack_off_b3(d.C, d.bands, 0),
                               pk_ns, d_params + L.param_off[3], param_stride, w.p1, w.m1, st, &h1))))) return rc;
  }
  const Conv3H2 h2 = {d_packed + pack_off_h2(d.C, d.bands, 2), pk_ns, h2flag, gen_stats ? w.hstat : nullptr, 1};
  if ((rc = TIMED(CMLPL_K_CONV2_FWD, chk(launch_conv3(0, nets, n, d.H2, d.W2, w.p1, nullptr, d_packed + pack_off_b3(d.C, d.bands, 2),
                             pk_ns, d_params + L.param_off[5], param_stride, w.p2, w.m2, st, &h2))))) return rc;
  const int nlab = shard ? shard->nlab : n, lab0 = shard ? shard->lab0 : 0;
  const int unl_base = shard ? shard->bt_g + shard->unl0 : n;
  return TIMED(CMLPL_K_HEAD_FWD, chk(launch_head_fwd(nets, n, d.P4, d.K, w.p2, w.y, d_dropmask, w.dropgen, dropout_p,
                             train, seed, step, nlab, lab0, unl_base, d_params + L.param_off[8],
                             d_params + L.param_off[9], param_stride, w.catd, w.ynorm, d_logits, d_feat, st,
                             xs.sel.dyn)));
}
}  // namespace

int cmlpl_basenet2_bwd(const cmlpl_shape* shape, int nets, int n, const float* d_params, int64_t param_stride,
                       const float* d_packed, const float* d_xn, const float* d_sn, const float* d_dropmask,
                       float dropout_p, int train, const float* d_dlogits, const float* d_dfeat, float* d_grads,
                       int64_t grad_stride, void* d_workspace, size_t workspace_bytes, void* stream) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  if (nets < 1 || nets > 2 || n < 1 || !d_params || !d_packed || !d_xn || !d_sn || !d_dlogits || !d_grads ||
      !d_workspace)
    return CMLPL_E_ARG;
  NetWs w;
  if (!carve_net(d, nets, n, (char*)d_workspace, &w)) return CMLPL_E_SHAPE;
  if (w.bytes > workspace_bytes) return CMLPL_E_WORKSPACE;
  return bwd_core(d, L, nets, n, d_params, param_stride, d_packed,
                  xsrc_plain(d_xn, nets, n, (long long)d.C * d.HW, 0, 0, nullptr), d_xn, d_sn, d_dropmask, dropout_p, train, d_dlogits, d_dfeat, d_grads, grad_stride, w, (hipStream_t)stream);
}

namespace {
int bwd_core(const Dims& d, const cmlpl_layout_t& L, int nets, int n, const float* d_params, int64_t param_stride,
             const float* d_packed, const XSrc& xs, const float* d_xn, const float* d_sn, const float* d_dropmask,
             float dropout_p, int train, const float* d_dlogits, const float* d_dfeat, float* d_grads,
             int64_t grad_stride, const NetWs& w, hipStream_t st, int* dyn_cursor, cmlpl_dyn* dyn_table, int parts) {
  const float* mask = (!train || dropout_p <= 0.f) ? nullptr : (d_dropmask ? d_dropmask : w.dropgen);
  int rc;
  const bool fused_head = conv3_fused_head_ok(d.H, d.W, d.C, nets * n, d.K);
  // in two parts the head runs WITHOUT the feature gradients (dy is linear in them; launch_dy_fixup adds their share
  // in front of the feat_spe weight-gradient GEMM, bit-identically)
  const float* d_dfeat_head = (parts == 3) ? d_dfeat : nullptr;
  GemmTN gw_cls, gw_spe;
  {
    GemmTN& g = gw_cls;
    g.bias_in = nullptr; g.bias_in_bstride = 0; g.relu = 0;
    // dW_cls[k][f] = sum_n dlogits[n][k] * catd[n][f] ; db_cls[k] = sum_n dlogits[n][k]
    g.A = d_dlogits; g.a_bstride = (long long)n * d.K; g.lda = d.K; g.M = d.K;
    g.B = w.catd; g.b_bstride = (long long)n * d.F; g.ldb = d.F; g.N = d.F;
    g.C = d_grads + L.param_off[8]; g.c_bstride = grad_stride; g.ldc = d.F;
    g.bias = d_grads + L.param_off[9]; g.bias_bstride = grad_stride;
    g.R = n; g.batches = nets; g.scale = 1.f;
    // dW_spe[o][b] = sum_n dy[n][o] * sn[n][b] ; db_spe[o] = sum_n dy[n][o]   (same launch)
    GemmTN& h = gw_spe;
    h = g;
    h.A = w.dy; h.a_bstride = (long long)n * 1024; h.lda = 1024; h.M = 1024;
    h.B = d_sn; h.b_bstride = (long long)n * d.bands; h.ldb = d.bands; h.N = d.bands;
    h.C = d_grads + L.param_off[6]; h.ldc = d.bands;
    h.bias = d_grads + L.param_off[7];
  }
  if (!(parts & 1)) {
  } else if (fused_head) {
    // ONE per-sample launch for the whole data-gradient chain: head backward -> conv2 data gradient -> conv1 data
    // gradient -> conv0 weight-gradient partial.  dp2 / dp1 / dy still go to HBM for the weight-gradient kernels
    // that follow; da0 and the pooled-gradient hand-offs stay on chip.
    BwdHead hd;
    hd.dlogits = d_dlogits; hd.dfeat = d_dfeat_head; hd.mask = mask; hd.wc = d_params + L.param_off[8]; hd.p_ns = param_stride;
    hd.y = w.y; hd.ynorm = w.ynorm; hd.m2 = w.m2; hd.w2d = d_packed + pack_off_b3(d.C, d.bands, 3); hd.w2d_ns = L.packed_total;
    hd.dy = w.dy; hd.dp2 = w.dp2; hd.dp1 = w.dp1; hd.K = d.K;
    hd.w1h = d_packed + pack_off_h2(d.C, d.bands, 1); hd.w1h_ns = L.packed_total;
    hd.h2flag = (const uint32_t*)(d_packed + pack_off_h2flag(d.C, d.bands));
    const bool h2w = conv3_h2x_both(d.H, d.W, d.C, nets * n, d.K);
    if (h2w) hd.hstat = w.hstat;
    if ((rc = TIMED(CMLPL_K_CONV1_DGRAD, chk(launch_conv3_fused_bwd(nets, n, d.C, d.H, d.W, w.dp1, w.m1,
                               d_packed + pack_off_b3(d.C, d.bands, 1), L.packed_total, xs, w.part0,
                               (long long)n * conv0_partial_size(d.C), &hd, st))))) return rc;
    // conv1's and conv2's weight gradients: one launch where the pair kernel exists (timed as the conv1 kernel)
    bool merged = false;
    if ((rc = TIMED(CMLPL_K_CONV1_WGRAD, chk(launch_wgrad3_pair(nets, n, d.H, d.W, w.a0, w.dp1, w.m1, w.part1, d.H2, d.W2,
                                  w.p1, w.dp2, w.m2, w.part2, &merged, st, h2w ? w.hstat : nullptr, hd.h2flag, L.packed_total))))) return rc;
    (void)merged;
  } else {
    // General path (windows the per-sample kernels do not take: P 20x20, B5 15x15): one launch per stage.  Round 4:
    // both 3x3 weight gradients in ONE launch here too (the pair kernel, placed behind conv2's data gradient, which
    // produces conv1's pooled gradient), and the classifier / feat_spe weight-gradient GEMMs ride in the reduce launch --
    // 17 launches where there were 19; the forked side streams of round 1 are gone (they never paid: DESIGN.md section 7).
    // feat is re-formed inside head_bwd as y / ||y|| (the forward's own division), so it is not an input here
    if ((rc = TIMED(CMLPL_K_HEAD_BWD, chk(launch_head_bwd(nets, n, d.P4, d.K, d_dlogits, d_dfeat_head, mask,
                                  d_params + L.param_off[8], param_stride, w.y, w.ynorm, w.dy, w.dp2, st))))) return rc;
    const uint32_t* h2flag = (const uint32_t*)(d_packed + pack_off_h2flag(d.C, d.bands));
    const bool gen_stats = general_h2_stats(d, nets * n);
    const Conv3H2 h3 = {d_packed + pack_off_h2(d.C, d.bands, 3), L.packed_total, h2flag, gen_stats ? w.hstat : nullptr, 3};
    if ((rc = TIMED(CMLPL_K_CONV2_DGRAD, chk(launch_conv3(1, nets, n, d.H2, d.W2, w.dp2, w.m2, d_packed + pack_off_b3(d.C, d.bands, 3),
                               L.packed_total, nullptr, 0, w.dp1, nullptr, st, &h3))))) return rc;
    // (round 6: conv1's data gradient in FRONT of the weight-gradient pair launch -- it leaves the maxima of conv1's
    //  gradient operand that the two-piece weight gradient scales by; neither reads what the other writes)
    if (conv3_fused_bwd_ok(d.H, d.W, d.C, nets * n)) {
      // conv1 data gradient + conv0 weight gradient in one launch: da0 never goes to HBM
      if ((rc = TIMED(CMLPL_K_CONV1_DGRAD, chk(launch_conv3_fused_bwd(nets, n, d.C, d.H, d.W, w.dp1, w.m1,
                                 d_packed + pack_off_b3(d.C, d.bands, 1), L.packed_total, xs, w.part0,
                                 (long long)n * conv0_partial_size(d.C), nullptr, st))))) return rc;
    } else {
      if (!d_xn) return CMLPL_E_ARG;
      const Conv3H2 h1 = {d_packed + pack_off_h2(d.C, d.bands, 1), L.packed_total, h2flag, gen_stats ? w.hstat : nullptr, 2};
      if ((rc = TIMED(CMLPL_K_CONV1_DGRAD, chk(launch_conv3(1, nets, n, d.H, d.W, w.dp1, w.m1, d_packed + pack_off_b3(d.C, d.bands, 1),
                                 L.packed_total, nullptr, 0, w.da0, nullptr, st, &h1))))) return rc;
      if ((rc = TIMED(CMLPL_K_CONV0_WGRAD, chk(launch_conv0_wgrad(nets, n, d.C, d.HW, d_xn, w.da0, w.part0, st, gen_stats ? w.hstat : nullptr, h2flag,
                                                                         L.packed_total)))))
        return rc;
    }
    bool merged = false;
    if ((rc = TIMED(CMLPL_K_CONV1_WGRAD, chk(launch_wgrad3_pair(nets, n, d.H, d.W, w.a0, w.dp1, w.m1, w.part1, d.H2, d.W2,
                                  w.p1, w.dp2, w.m2, w.part2, &merged, st, gen_stats ? w.hstat : nullptr, h2flag,
                                  L.packed_total))))) return rc;
    (void)merged;
  }
  if (!(parts & 2)) return 0;
  if (parts == 2) {
    if (!d_dfeat) return CMLPL_E_ARG;
    if ((rc = TIMED(CMLPL_K_HEAD_BWD, chk(launch_dy_fixup(nets, n, w.y, w.ynorm, d_dfeat, w.dy, st))))) return rc;
  }
  // one launch folds the per-workgroup partials of all three convolutions into the flat gradient
  ReduceTable rt;
  rt.count = 0; rt.total_blocks = 0; rt.grad_ns = grad_stride; rt.dyn_cursor = dyn_cursor; rt.dyn_table = dyn_table;
  Wgrad3Plan wp1, wp2;
  bool wpair = false;
  if (!plan_wgrad3_both(nets, n, d.H, d.W, d.H2, d.W2, true, &wp1, &wp2, &wpair)) return CMLPL_E_SHAPE;
  reduce_table_add(rt, w.part1, wp1.G, PART3, 1, 64, d_grads + L.param_off[2], d_grads + L.param_off[3]);
  reduce_table_add(rt, w.part2, wp2.G, PART3, 1, 64, d_grads + L.param_off[4], d_grads + L.param_off[5]);
  reduce_table_add(rt, w.part0, conv0_partials(d, nets, n), conv0_partial_size(d.C), 0, d.C,
                   d_grads + L.param_off[0], d_grads + L.param_off[1]);
  // the classifier / feat_spe weight-gradient GEMMs ride along (independent, short)
  return TIMED(CMLPL_K_CONV1_WRED, chk(launch_reduce_gemm(nets, rt, gw_cls, gw_spe, st)));
}

// does the step need an augmented copy of the patches in HBM?  (only when a conv0 pass falls back to the unfused kernels)
bool need_xn_copy(const Dims& d, int rows) {
  return !(conv3_fused_ok(d.H, d.W, d.C, rows) && conv3_fused_bwd_ok(d.H, d.W, d.C, rows));
}
bool check_batch(const cmlpl_batch* b) {
  return b && b->bt >= 1 && b->btu >= 1 && b->d_xpl && b->d_xl && b->d_xpu && b->d_xu;
}
}  // namespace

}  // extern "C"
namespace {
int forward_impl(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch, const cmlpl_shard* shard,
                 const float* d_params, const float* d_packed, const float* d_dropmask, int train, uint64_t seed,
                 uint64_t step, float* d_logits, float* d_feat, float* d_labels_f, void* d_workspace,
                 size_t workspace_bytes, void* stream, DynRef dyn, int parts = 3) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  // in two parts (the sharded step) the spectral part forms the embeddings itself: early_feat
  const bool early_feat = parts != 3;
  if (!hp || !check_batch(batch) || !d_params || !d_workspace || ((parts & 2) && (!d_packed || !d_logits)) ||
      ((parts & 1 || !early_feat) && !d_feat))
    return CMLPL_E_ARG;
  if (d_labels_f && !batch->d_labels) return CMLPL_E_ARG;
  if (hp->dropout_p < 0.f || hp->dropout_p >= 1.f) return CMLPL_E_ARG;
  const int n = batch->bt + batch->btu;
  if (shard && (shard->nlab != batch->bt || shard->nunl != batch->btu)) return CMLPL_E_ARG;
  NetWs nw;
  if (!carve_net(d, 2, n, (char*)d_workspace, &nw)) return CMLPL_E_SHAPE;
  StepWs sw;
  carve_step(d, n, n, (char*)d_workspace + nw.bytes, &sw);
  if (nw.bytes + ((char*)sw.dlogits - ((char*)d_workspace + nw.bytes)) > workspace_bytes) return CMLPL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool copy = need_xn_copy(d, 2 * n);
  const int lab0 = shard ? shard->lab0 : 0, unl_base = shard ? shard->bt_g + shard->unl0 : batch->bt;
  int rc;
  // the augmentation launch remains only for what the fused kernels cannot take raw: the patches when a conv0 pass
  // falls back to the unfused kernels, the spectra when the fused spectral kernel does not apply, and the label
  // conversion for the data-parallel exchange buffer
  const bool spe_fused = spe_fused_ok(d.bands);
  // (general path: the augmentation of the patches rides in conv0's launch where conv0a_fwd_kernel takes the window; it
  //  leaves the augmented rows in sw.xn like the fused forward does)
  const bool c0a = copy && !conv3_fused_ok(d.H, d.W, d.C, 2 * n) && conv0a_ok(d.C, d.HW);
  const int which = ((parts & 2) && copy && !c0a ? 1 : 0) | ((parts & 1) && !spe_fused ? 2 : 0);
  if (!(parts & 1)) d_labels_f = nullptr;     // (the labels travel with the spectral part)
  const RowSel sel = batch_sel(batch, dyn);
  if (which &&
      (rc = TIMED(CMLPL_K_AUGMENT, chk(launch_augment(which, 2, batch->bt, batch->btu, d.C * d.HW, d.bands, lab0,
                                unl_base, batch->d_xpl, batch->d_xl, batch->d_xpu, batch->d_xu, batch->noise8,
                                hp->noise_sigma, seed, step, sw.xn, sw.sn, nullptr, st,
                                (const long long*)batch->d_labels, d_labels_f, &sel))))) return rc;
  XSrc xspec = xsrc_raw(batch, hp->noise_sigma, seed, step, shard, dyn);
  for (int i = 0; i < 2; ++i) {       // the spectral rows and their draws (reference order, see cmlpl_augment)
    xspec.lab[i] = batch->d_xl; xspec.unl[i] = batch->d_xu;
    xspec.nz_lab[i] = batch->noise8 ? batch->noise8[2 * i + 1] : nullptr;
    xspec.nz_unl[i] = batch->noise8 ? batch->noise8[4 + 2 * i + 1] : nullptr;
  }
  // (fused per-sample kernels: the forward leaves the rows it saw in sw.xn for cmlpl_backward, which lands them by DMA)
  return fwd_core(d, L, 2, n, d_params, L.param_total, d_packed, xsrc_raw(batch, hp->noise_sigma, seed, step, shard, dyn),
                  spe_fused ? &xspec : nullptr, sw.sn, (const long long*)batch->d_labels, d_labels_f,
                  (copy && !c0a) ? sw.xn : nullptr, sw.sn, nullptr, d_dropmask, hp->dropout_p,
                  train, seed, step, shard, d_logits, d_feat, nw, st,
                  (!copy || c0a) ? sw.xn : nullptr, parts, early_feat);
}
}  // namespace
extern "C" {
int cmlpl_forward_spectral(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                           const cmlpl_shard* shard, const float* d_params, uint64_t seed, uint64_t step, float* d_feat,
                           float* d_labels_f, void* d_workspace, size_t workspace_bytes, void* stream) {
  return forward_impl(shape, hp, batch, shard, d_params, nullptr, nullptr, 1, seed, step, nullptr, d_feat, d_labels_f,
                      d_workspace, workspace_bytes, stream, DynRef(), 1);
}
int cmlpl_forward_spatial(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                          const cmlpl_shard* shard, const float* d_params, const float* d_packed, const float* d_dropmask,
                          int train, uint64_t seed, uint64_t step, float* d_logits, void* d_workspace,
                          size_t workspace_bytes, void* stream) {
  return forward_impl(shape, hp, batch, shard, d_params, d_packed, d_dropmask, train, seed, step, d_logits, nullptr, nullptr,
                      d_workspace, workspace_bytes, stream, DynRef(), 2);
}
int cmlpl_forward(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch, const cmlpl_shard* shard,
                  const float* d_params, const float* d_packed, const float* d_dropmask, int train, uint64_t seed,
                  uint64_t step, float* d_logits, float* d_feat, float* d_labels_f, void* d_workspace,
                  size_t workspace_bytes, void* stream) {
  return forward_impl(shape, hp, batch, shard, d_params, d_packed, d_dropmask, train, seed, step, d_logits, d_feat,
                      d_labels_f, d_workspace, workspace_bytes, stream, DynRef());
}
}  // extern "C"
namespace {
int backward_impl(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch, const cmlpl_shard* shard,
                  const float* d_params, const float* d_packed, const float* d_dropmask, int train, uint64_t seed,
                  uint64_t step, const float* d_dlogits, const float* d_dfeat, float* d_grads, int64_t grad_stride,
                  void* d_workspace, size_t workspace_bytes, void* stream, DynRef dyn, int* dyn_cursor, int parts = 3);
}  // namespace
extern "C" {

int cmlpl_backward_data(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch, const cmlpl_shard* shard,
                        const float* d_params, const float* d_packed, const float* d_dropmask, int train, uint64_t seed,
                        uint64_t step, const float* d_dlogits, void* d_workspace, size_t workspace_bytes, void* stream) {
  return backward_impl(shape, hp, batch, shard, d_params, d_packed, d_dropmask, train, seed, step, d_dlogits, nullptr,
                       nullptr, 0, d_workspace, workspace_bytes, stream, DynRef(), nullptr, 1);
}
int cmlpl_backward_weights(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                           const cmlpl_shard* shard, const float* d_params, const float* d_packed, const float* d_dropmask,
                           int train, uint64_t seed, uint64_t step, const float* d_dlogits, const float* d_dfeat,
                           float* d_grads, int64_t grad_stride, void* d_workspace, size_t workspace_bytes, void* stream) {
  return backward_impl(shape, hp, batch, shard, d_params, d_packed, d_dropmask, train, seed, step, d_dlogits, d_dfeat,
                       d_grads, grad_stride, d_workspace, workspace_bytes, stream, DynRef(), nullptr, 2);
}

int cmlpl_backward(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch, const cmlpl_shard* shard,
                   const float* d_params, const float* d_packed, const float* d_dropmask, int train, uint64_t seed,
                   uint64_t step, const float* d_dlogits, const float* d_dfeat, float* d_grads, int64_t grad_stride,
                   void* d_workspace, size_t workspace_bytes, void* stream) {
  return backward_impl(shape, hp, batch, shard, d_params, d_packed, d_dropmask, train, seed, step, d_dlogits, d_dfeat,
                       d_grads, grad_stride, d_workspace, workspace_bytes, stream, DynRef(), nullptr);
}
}  // extern "C"
namespace {
int backward_impl(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch, const cmlpl_shard* shard,
                  const float* d_params, const float* d_packed, const float* d_dropmask, int train, uint64_t seed,
                  uint64_t step, const float* d_dlogits, const float* d_dfeat, float* d_grads, int64_t grad_stride,
                  void* d_workspace, size_t workspace_bytes, void* stream, DynRef dyn, int* dyn_cursor, int parts) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  if (!hp || !check_batch(batch) || !d_params || !d_packed || !d_dlogits || ((parts & 2) && !d_grads) || !d_workspace)
    return CMLPL_E_ARG;
  const int n = batch->bt + batch->btu;
  NetWs nw;
  if (!carve_net(d, 2, n, (char*)d_workspace, &nw)) return CMLPL_E_SHAPE;
  StepWs sw;
  carve_step(d, n, n, (char*)d_workspace + nw.bytes, &sw);
  if (nw.bytes + ((char*)sw.dlogits - ((char*)d_workspace + nw.bytes)) > workspace_bytes) return CMLPL_E_WORKSPACE;
  const bool copy = need_xn_copy(d, 2 * n);
  // the patches as the forward saw them: sw.xn holds them in [nets][n][C*HW] layout either way -- written by the fused
  // forward (augmented, or plain copies when no noise is added) or, when a conv0 pass fell back to the unfused kernels
  // (`copy`), by the augmentation launch.  The fused data gradient reads plain rows by batch row: it knows neither noise
  // nor index lists, so it must never be handed the raw batch (a 9x9 window at 128 + 128 rows plans an unfused forward
  // and a fused backward: with the raw rows conv0's weight gradient came from un-augmented rows 0..n-1 of the split).
  const XSrc xs = xsrc_plain(sw.xn, 2, n, (long long)d.C * d.HW, seed, step, shard, dyn);
  return bwd_core(d, L, 2, n, d_params, L.param_total, d_packed, xs,
                  copy ? sw.xn : nullptr, sw.sn, d_dropmask, hp->dropout_p, train, d_dlogits, d_dfeat, d_grads,
                  grad_stride, nw, (hipStream_t)stream, dyn_cursor, dyn_cursor ? (cmlpl_dyn*)dyn.table : nullptr, parts);
}
}  // namespace
extern "C" {

namespace {
int fill_loss_args(const Dims& d, const cmlpl_shard* sh, const float* d_logits, const float* d_feat,
                   const int64_t* d_labels, const cmlpl_banks* banks, int smooth, float adap_mask,
                   const cmlpl_hparams* hp, void* ws, size_t ws_bytes, LossArgs* out,
                   const cmlpl_gathered* gth = nullptr, const RowSel* sel = nullptr) {
  if (!sh || !banks || !hp || !ws) return CMLPL_E_ARG;
  if (gth == nullptr && (!d_logits || !d_feat || !d_labels)) return CMLPL_E_ARG;
  if (gth != nullptr && (!gth->d_recv_feat || !gth->d_logits_local || gth->world < 1 || gth->bt_local < 1 || gth->btu_local < 1 ||
                         gth->world * gth->bt_local != sh->bt_g || gth->world * gth->btu_local != sh->btu_g ||
                         sh->nunl != gth->btu_local || sh->unl0 % gth->btu_local != 0 ||
                         (sh->nlab != 0 && (sh->nlab != gth->bt_local || sh->lab0 % gth->bt_local != 0))))
    return CMLPL_E_ARG;
  if (sh->bt_g < 1 || sh->btu_g < 1 || sh->btu_g > 2048 || sh->nlab < 0 || sh->nunl < 1 || sh->lab0 < 0 ||
      sh->unl0 < 0 || sh->lab0 + sh->nlab > sh->bt_g || sh->unl0 + sh->nunl > sh->btu_g)
    return CMLPL_E_ARG;
  if (banks->Q < sh->bt_g + sh->btu_g || !banks->d_feats[0] || !banks->d_feats[1] || !banks->d_probs[0] ||
      !banks->d_probs[1])
    return CMLPL_E_ARG;
  if (loss_ws_floats(sh->nlab, sh->nunl, sh->btu_g, d.K, banks->Q) * 4 > ws_bytes) return CMLPL_E_WORKSPACE;
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = d_logits; a.feat = d_feat; a.labels = d_labels;
  if (sel != nullptr) a.sel = *sel;
  if (gth != nullptr) {
    const long long n_l = gth->bt_local + gth->btu_local;
    a.recv_f = gth->d_recv_feat; a.bt_l = gth->bt_local; a.btu_l = gth->btu_local;
    a.pack_f = 2 * n_l * 1024 + gth->bt_local;
    a.logits_loc = gth->d_logits_local;
  }
  for (int i = 0; i < 2; ++i) {
    a.bank_f[i] = banks->d_feats[i]; a.bank_p[i] = banks->d_probs[i];
    a.bank_fw[i] = banks->d_feats[i]; a.bank_pw[i] = banks->d_probs[i];
  }
  a.Q = banks->Q; a.ptr0 = ((banks->ptr[0] % a.Q) + a.Q) % a.Q; a.ptr1 = ((banks->ptr[1] % a.Q) + a.Q) % a.Q;
  a.bt = sh->bt_g; a.btu = sh->btu_g; a.K = d.K; a.smooth = smooth;
  a.lab0 = sh->lab0; a.nlab = sh->nlab; a.unl0 = sh->unl0; a.nunl = sh->nunl;
  a.adap_mask = adap_mask; a.T = hp->temperature; a.alpha = hp->alpha;
  a.w_contrast = hp->w_contrast; a.w_mutual = hp->w_mutual; a.pos_thr = hp->pos_thr; a.neg_thr = hp->neg_thr;
  loss_ws_carve(a, (float*)ws);
  *out = a;
  return 0;
}
}  // namespace

size_t cmlpl_loss_workspace_bytes(const cmlpl_shape* shape, const cmlpl_shard* shard, int bank_rows) {
  Dims d;
  if (!make_dims(shape, &d) || !shard || shard->nunl < 1 || shard->btu_g < 1 || bank_rows < 1) return 0;
  return loss_ws_floats(shard->nlab, shard->nunl, shard->btu_g, d.K, bank_rows) * 4;
}

int cmlpl_loss_phase1(const cmlpl_shape* shape, const cmlpl_shard* shard, const float* d_logits, const float* d_feat,
                      const int64_t* d_labels, const cmlpl_banks* banks, int smooth, float adap_mask,
                      const cmlpl_hparams* hp, float* d_dlogits, float* d_dfeat, float* d_probs_local,
                      void* d_workspace, size_t workspace_bytes, void* stream) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  LossArgs a;
  int rc = fill_loss_args(d, shard, d_logits, d_feat, d_labels, banks, smooth, adap_mask, hp, d_workspace,
                          workspace_bytes, &a);
  if (rc) return rc;
  if (!d_dlogits || !d_dfeat || !d_probs_local) return CMLPL_E_ARG;
  a.dlogits = d_dlogits; a.dfeat = d_dfeat; a.probs_l = d_probs_local;
  hipStream_t st = (hipStream_t)stream;
  return TIMED(CMLPL_K_LOSS, chk(launch_loss_phase1(a, st)));
}

int cmlpl_loss_phase2(const cmlpl_shape* shape, const cmlpl_shard* shard, const float* d_logits, const float* d_feat,
                      const int64_t* d_labels, const cmlpl_banks* banks, int smooth, float adap_mask,
                      const cmlpl_hparams* hp, const float* d_probs_global, int probs_shard_rows, float* d_scalars,
                      float* d_dfeat, float* d_dfeat_w_partial, void* d_workspace, size_t workspace_bytes,
                      void* stream) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  LossArgs a;
  int rc = fill_loss_args(d, shard, d_logits, d_feat, d_labels, banks, smooth, adap_mask, hp, d_workspace,
                          workspace_bytes, &a);
  if (rc) return rc;
  if (!d_probs_global || !d_scalars || !d_dfeat || !d_dfeat_w_partial || probs_shard_rows < 1 ||
      shard->btu_g % probs_shard_rows != 0)
    return CMLPL_E_ARG;
  a.probs_g = d_probs_global; a.pshard = probs_shard_rows; a.scalars = d_scalars; a.dfeat = d_dfeat;
  a.dfw_part = d_dfeat_w_partial;
  hipStream_t st = (hipStream_t)stream;
  int rc2 = TIMED(CMLPL_K_LOSS2, chk(launch_loss_graph(a, st)));
  if (rc2) return rc2;
  // the two feature-gradient GEMMs + the scalar block (the bank write went with phase 1's row kernel)
  return TIMED(CMLPL_K_LOSS_DFEAT, chk(launch_loss_dfeat(a, st)));
}

int cmlpl_loss_phase1_g(const cmlpl_shape* shape, const cmlpl_shard* shard, const cmlpl_gathered* gathered,
                        const cmlpl_banks* banks, int smooth, float adap_mask, const cmlpl_hparams* hp,
                        float* d_dlogits, float* d_dfeat, float* d_probs_local, void* d_workspace,
                        size_t workspace_bytes, void* stream) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  if (!gathered) return CMLPL_E_ARG;
  LossArgs a;
  int rc = fill_loss_args(d, shard, nullptr, nullptr, nullptr, banks, smooth, adap_mask, hp, d_workspace,
                          workspace_bytes, &a, gathered);
  if (rc) return rc;
  if (!d_dlogits || !d_dfeat || !d_probs_local) return CMLPL_E_ARG;
  a.dlogits = d_dlogits; a.dfeat = d_dfeat; a.probs_l = d_probs_local;
  hipStream_t st = (hipStream_t)stream;
  return TIMED(CMLPL_K_LOSS, chk(launch_loss_phase1(a, st)));
}

int cmlpl_loss_phase2_g(const cmlpl_shape* shape, const cmlpl_shard* shard, const cmlpl_gathered* gathered,
                        const cmlpl_banks* banks, int smooth, float adap_mask, const cmlpl_hparams* hp,
                        const float* d_probs_global, int probs_shard_rows, float* d_scalars, float* d_dfeat,
                        float* d_dfeat_w_partial, void* d_workspace, size_t workspace_bytes, void* stream) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  if (!gathered) return CMLPL_E_ARG;
  LossArgs a;
  int rc = fill_loss_args(d, shard, nullptr, nullptr, nullptr, banks, smooth, adap_mask, hp, d_workspace,
                          workspace_bytes, &a, gathered);
  if (rc) return rc;
  if (!d_probs_global || !d_scalars || !d_dfeat || !d_dfeat_w_partial || probs_shard_rows < 1 ||
      shard->btu_g % probs_shard_rows != 0)
    return CMLPL_E_ARG;
  a.probs_g = d_probs_global; a.pshard = probs_shard_rows; a.scalars = d_scalars; a.dfeat = d_dfeat;
  a.dfw_part = d_dfeat_w_partial;
  hipStream_t st = (hipStream_t)stream;
  int rc2 = TIMED(CMLPL_K_LOSS2, chk(launch_loss_graph(a, st)));
  if (rc2) return rc2;
  return TIMED(CMLPL_K_LOSS_DFEAT, chk(launch_loss_dfeat(a, st)));
}

}  // extern "C"
namespace {
// the whole loss block on one GPU (both phases back to back); sel: labels by index / device-side step scalars
int loss_both(const cmlpl_shape* shape, int bt, int btu, const float* d_logits, const float* d_feat,
              const int64_t* d_labels, const cmlpl_banks* banks, int smooth, float adap_mask,
              const cmlpl_hparams* hp, float* d_scalars, float* d_dlogits, float* d_dfeat,
              float* d_probs, void* d_workspace, size_t workspace_bytes, void* stream, const RowSel* sel) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  if (!d_probs || !d_dfeat || !d_dlogits || !d_scalars) return CMLPL_E_ARG;
  const cmlpl_shard sh = {bt, btu, 0, bt, 0, btu};
  LossArgs a;
  int rc = fill_loss_args(d, &sh, d_logits, d_feat, d_labels, banks, smooth, adap_mask, hp, d_workspace,
                          workspace_bytes, &a, nullptr, sel);
  if (rc) return rc;
  a.dlogits = d_dlogits; a.dfeat = d_dfeat; a.probs_l = d_probs;
  // one GPU: the probabilities are already global, and the column-side gradient goes straight to dfeat[1]
  a.probs_g = d_probs; a.pshard = btu; a.scalars = d_scalars;
  a.dfw_part = d_dfeat + ((size_t)(bt + btu) + bt) * 1024;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = TIMED(CMLPL_K_LOSS, chk(launch_loss_phase1(a, st))))) return rc;
  if ((rc = TIMED(CMLPL_K_LOSS2, chk(launch_loss_graph(a, st))))) return rc;
  return TIMED(CMLPL_K_LOSS_DFEAT, chk(launch_loss_dfeat(a, st)));
}
}  // namespace
extern "C" {
int cmlpl_loss_fwd_bwd(const cmlpl_shape* shape, int bt, int btu, const float* d_logits, const float* d_feat,
                       const int64_t* d_labels, const cmlpl_banks* banks, int smooth, float adap_mask,
                       const cmlpl_hparams* hp, float* d_scalars, float* d_dlogits, float* d_dfeat,
                       float* d_probs, void* d_workspace, size_t workspace_bytes, void* stream) {
  return loss_both(shape, bt, btu, d_logits, d_feat, d_labels, banks, smooth, adap_mask, hp, d_scalars, d_dlogits,
                   d_dfeat, d_probs, d_workspace, workspace_bytes, stream, nullptr);
}

int cmlpl_dist_unpack(const cmlpl_shape* shape, int world, int bt_local, int btu_local, const float* d_gathered_feat,
                      const float* d_gathered_logits, float* d_logits_g, float* d_feat_g, int64_t* d_labels_g, void* stream) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  if (world < 1 || bt_local < 1 || btu_local < 1 || !d_gathered_feat || !d_gathered_logits || !d_logits_g || !d_feat_g ||
      !d_labels_g)
    return CMLPL_E_ARG;
  return chk(launch_dist_unpack(d_gathered_feat, d_gathered_logits, world, bt_local, btu_local, d.K, d_logits_g, d_feat_g,
                                (long long*)d_labels_g, (hipStream_t)stream));
}

int cmlpl_dyn_adam(const cmlpl_hparams* hp, int64_t adam_t, float* step_size, float* bc2_sqrt) {
  if (!hp || adam_t < 1 || !step_size || !bc2_sqrt) return CMLPL_E_ARG;
  adam_bias_scalars(hp->lr, hp->beta1, hp->beta2, adam_t, step_size, bc2_sqrt);
  return 0;
}

}  // extern "C"
namespace {
int adam_impl(const cmlpl_shape* shape, int nets, float* d_params, int64_t param_stride,
              const float* d_grads, int64_t grad_stride, float* d_m, float* d_v, int64_t t,
              const cmlpl_hparams* hp, float* d_packed, void* stream, DynRef dyn) {
  cmlpl_layout_t L;
  Dims d;
  int rc = cmlpl_layout(shape, &L);
  if (rc) return rc;
  make_dims(shape, &d);
  if (nets < 1 || nets > 2 || !d_params || !d_grads || !d_m || !d_v || !hp || t < 1) return CMLPL_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  return TIMED(CMLPL_K_ADAM, chk(launch_adam(nets, d_params, param_stride, d_grads, grad_stride, d_m, d_v,
                            L.param_live, t, hp->lr, hp->beta1, hp->beta2, hp->eps, d_packed,
                            make_pack_info(d, L), st, dyn)));
}
}  // namespace
extern "C" {
int cmlpl_adam_step(const cmlpl_shape* shape, int nets, float* d_params, int64_t param_stride,
                    const float* d_grads, int64_t grad_stride, float* d_m, float* d_v, int64_t t,
                    const cmlpl_hparams* hp, float* d_packed, void* stream) {
  return adam_impl(shape, nets, d_params, param_stride, d_grads, grad_stride, d_m, d_v, t, hp, d_packed, stream, DynRef());
}

int cmlpl_train_step(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_step_io* io, void* stream) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  if (!hp || !io || io->bt < 1 || io->btu < 1 || !io->d_workspace || !io->d_params || !io->d_grads ||
      !io->d_packed || !io->d_logits || !io->d_feat || !io->d_scalars || !io->d_labels)
    return CMLPL_E_ARG;
  if (io->apply_update && (!io->d_m || !io->d_v)) return CMLPL_E_ARG;
  const int n = io->bt + io->btu;
  NetWs nw;
  if (!carve_net(d, 2, n, (char*)io->d_workspace, &nw)) return CMLPL_E_SHAPE;
  StepWs sw;
  carve_step(d, n, io->banks.Q, (char*)io->d_workspace + nw.bytes, &sw);
  if (nw.bytes + sw.bytes > io->workspace_bytes) return CMLPL_E_WORKSPACE;
  const int train = 1;
  int rc;
  if ((io->d_dyn_table == nullptr) != (io->d_dyn_cursor == nullptr)) return CMLPL_E_ARG;
  DynRef dyn;
  dyn.table = io->d_dyn_table; dyn.cursor = io->d_dyn_cursor;
  const cmlpl_shard sh = {io->bt, io->btu, 0, io->bt, 0, io->btu};
  const cmlpl_batch batch = {io->d_xpl, io->d_xl, io->d_xpu, io->d_xu, io->d_labels, io->noise8, io->bt, io->btu,
                             io->d_lab_idx, io->d_unl_idx};
  if ((rc = forward_impl(shape, hp, &batch, &sh, io->d_params, io->d_packed, io->d_dropmask, train, io->seed,
                         io->step, io->d_logits, io->d_feat, nullptr, io->d_workspace, io->workspace_bytes, stream, dyn)))
    return rc;
  const RowSel sel = batch_sel(&batch, dyn);
  if ((rc = loss_both(shape, io->bt, io->btu, io->d_logits, io->d_feat, io->d_labels, &io->banks,
                      io->smooth, io->adap_mask, hp, io->d_scalars, sw.dlogits, sw.dfeat, sw.probs, sw.loss,
                      loss_ws_floats(n, n, n, d.K, io->banks.Q > n ? io->banks.Q : n) * 4, stream, &sel)))
    return rc;
  if ((rc = backward_impl(shape, hp, &batch, &sh, io->d_params, io->d_packed, io->d_dropmask, train, io->seed,
                          io->step, sw.dlogits, sw.dfeat, io->d_grads, L.param_total, io->d_workspace,
                          io->workspace_bytes, stream, dyn, io->d_dyn_cursor))) return rc;
  if (io->apply_update)   // (device-side scalars: the by-value step count is not used, any valid one will do)
    return adam_impl(shape, 2, io->d_params, L.param_total, io->d_grads, L.param_total, io->d_m, io->d_v,
                     (dyn.table != nullptr && io->adam_t < 1) ? 1 : io->adam_t, hp, io->d_packed, stream, dyn);
  return 0;
}

// ---- the step as a replayable hipGraph
struct StepGraph { hipGraph_t graph; hipGraphExec_t exec; };

int cmlpl_step_graph_create(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_step_io* io, void* stream,
                            void** graph_out) {
  if (!io || !graph_out || !io->d_dyn_table || !io->d_dyn_cursor) return CMLPL_E_ARG;
  if (g_timing.on) return CMLPL_E_ARG;                     // (event pairs are not part of the product's graph)
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = loss_prepare_capture();
  if (e != hipSuccess) return (int)e;
  e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return (int)e;
  const int rc = cmlpl_train_step(shape, hp, io, stream);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st, &g);                          // always end the capture, also after a failed launch
  if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return (int)e; }
  StepGraph* sg = new StepGraph{g, ex};
  *graph_out = sg;
  return 0;
}

// one stage of the sharded step, every per-step scalar from the device table (see cmlpl_dist_io)
static int dist_stage_run(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_dist_io* io, int stage, void* stream) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  DynRef dyn;
  dyn.table = io->d_dyn_table; dyn.cursor = io->d_dyn_cursor;
  const int train = 1;
  switch (stage) {
    case CMLPL_STAGE_SPECTRAL:
      return forward_impl(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, nullptr, train, io->seed, 0,
                          nullptr, io->d_feat_l, io->d_labels_f, io->d_workspace, io->workspace_bytes, stream, dyn, 1);
    case CMLPL_STAGE_SPATIAL:
      return forward_impl(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, nullptr, train, io->seed, 0,
                          io->d_logits_l, nullptr, nullptr, io->d_workspace, io->workspace_bytes, stream, dyn, 2);
    case CMLPL_STAGE_PHASE1: {
      LossArgs a;
      RowSel sel = RowSel();
      sel.dyn = dyn;
      int rc = fill_loss_args(d, &io->shard, nullptr, nullptr, nullptr, &io->banks, 1, 0.f, hp, io->d_loss_workspace,
                              io->loss_workspace_bytes, &a, &io->gathered, &sel);
      if (rc) return rc;
      if (!io->d_dlogits || !io->d_dfeat || !io->d_probs_l) return CMLPL_E_ARG;
      a.dlogits = io->d_dlogits; a.dfeat = io->d_dfeat; a.probs_l = io->d_probs_l;
      return chk(launch_loss_phase1(a, (hipStream_t)stream));
    }
    case CMLPL_STAGE_PHASE2: {
      LossArgs a;
      RowSel sel = RowSel();
      sel.dyn = dyn;
      int rc = fill_loss_args(d, &io->shard, nullptr, nullptr, nullptr, &io->banks, 1, 0.f, hp, io->d_loss_workspace,
                              io->loss_workspace_bytes, &a, &io->gathered, &sel);
      if (rc) return rc;
      if (!io->d_probs_g || !io->d_scalars || !io->d_dfeat || !io->d_dfeat_w_partial || io->probs_shard_rows < 1 ||
          io->shard.btu_g % io->probs_shard_rows != 0)
        return CMLPL_E_ARG;
      a.probs_g = io->d_probs_g; a.pshard = io->probs_shard_rows; a.scalars = io->d_scalars; a.dfeat = io->d_dfeat;
      a.dfw_part = io->d_dfeat_w_partial;
      if ((rc = chk(launch_loss_graph(a, (hipStream_t)stream)))) return rc;
      return chk(launch_loss_dfeat(a, (hipStream_t)stream));
    }
    case CMLPL_STAGE_BACKWARD_DATA:
      return backward_impl(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, nullptr, train, io->seed, 0,
                           io->d_dlogits, nullptr, nullptr, 0, io->d_workspace, io->workspace_bytes, stream, dyn, nullptr, 1);
    case CMLPL_STAGE_BACKWARD_WEIGHTS:      // (the reduce launch of this part advances the table's cursor)
      return backward_impl(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, nullptr, train, io->seed, 0,
                           io->d_dlogits, io->d_dfeat, io->d_grads, io->grad_stride, io->d_workspace, io->workspace_bytes,
                           stream, dyn, io->d_dyn_cursor, 2);
    case CMLPL_STAGE_UPDATE:
      return adam_impl(shape, 2, io->d_params, L.param_total, io->d_grads, io->grad_stride, io->d_m, io->d_v, 1, hp,
                       io->d_packed, stream, dyn);
    default:
      return CMLPL_E_ARG;
  }
}

int cmlpl_dist_stage_graph_create(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_dist_io* io, int stage,
                                  void* stream, void** graph_out) {
  if (!shape || !hp || !io || !graph_out || !io->d_dyn_table || !io->d_dyn_cursor || !io->batch.d_lab_idx ||
      !io->batch.d_unl_idx || io->batch.noise8 != nullptr)
    return CMLPL_E_ARG;
  if (g_timing.on) return CMLPL_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = loss_prepare_capture();
  if (e != hipSuccess) return (int)e;
  e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return (int)e;
  const int rc = dist_stage_run(shape, hp, io, stage, stream);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st, &g);
  if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return (int)e; }
  *graph_out = new StepGraph{g, ex};
  return 0;
}

int cmlpl_step_graph_launch(void* graph, void* stream) {
  if (!graph) return CMLPL_E_ARG;
  return chk(hipGraphLaunch(((StepGraph*)graph)->exec, (hipStream_t)stream));
}

int cmlpl_step_graph_destroy(void* graph) {
  if (!graph) return CMLPL_E_ARG;
  StepGraph* sg = (StepGraph*)graph;
  (void)hipGraphExecDestroy(sg->exec);
  (void)hipGraphDestroy(sg->graph);
  delete sg;
  return 0;
}

int cmlpl_extract_patches(const float* d_cube, int rows, int cols, int C, int w, const int64_t* d_pixel_idx, int n,
                          float* d_out, void* stream) {
  if (!d_cube || !d_pixel_idx || !d_out || rows < 1 || cols < 1 || C < 1 || w < 1 || n < 1) return CMLPL_E_ARG;
  if (w / 2 > rows || w / 2 > cols || (size_t)w * w * (C | 1) * 4 > LDS_MAX) return CMLPL_E_SHAPE;
  return chk(launch_extract_patches(d_cube, rows, cols, C, w, (const long long*)d_pixel_idx, n, d_out,
                                    (hipStream_t)stream));
}

size_t cmlpl_infer_workspace_bytes(const cmlpl_shape* shape, int n) {
  Dims d;
  if (!make_dims(shape, &d) || n < 1 || !conv3_infer_ok(d.H, d.W, d.C, d.K)) return 0;   // 0: not a shape cmlpl_infer_cube takes
  return up256((size_t)n * 1024 * 4);                       // y = relu(feat_spe(spectrum)) of the launch's pixels
}

int cmlpl_infer_cube(const cmlpl_shape* shape, const float* d_params, const float* d_packed, const float* d_cube,
                     int rows, int cols, const float* d_spectra, int64_t pixel0, int n, int64_t* d_labels,
                     float* d_logits, void* d_workspace, size_t workspace_bytes, void* stream) {
  Dims d;
  cmlpl_layout_t L;
  if (!make_dims(shape, &d) || cmlpl_layout(shape, &L)) return CMLPL_E_SHAPE;
  if (!d_params || !d_packed || !d_cube || !d_spectra || !d_labels || !d_workspace || rows < 1 || cols < 1 || n < 1 ||
      pixel0 < 0 || pixel0 + n > (int64_t)rows * cols)
    return CMLPL_E_ARG;
  if (!conv3_infer_ok(d.H, d.W, d.C, d.K)) return CMLPL_E_SHAPE;
  if (cmlpl_infer_workspace_bytes(shape, n) > workspace_bytes) return CMLPL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* y = (float*)d_workspace;
  int rc;
  // spectral branch of the range's pixels (plain rows, canonical weight)
  if ((rc = chk(launch_spe_fwd(1, n, d.bands, d_spectra + (long long)pixel0 * d.bands, d_params + L.param_off[6],
                               d_params + L.param_off[7], L.param_total, y, st)))) return rc;
  FwdTail t;
  memset(&t, 0, sizeof(t));
  t.w2f = d_packed + pack_off_b3(d.C, d.bands, 2); t.b2 = d_params + L.param_off[5];
  t.wc = d_params + L.param_off[8]; t.bc = d_params + L.param_off[9];
  t.y = y; t.logits = d_logits; t.K = d.K;
  return chk(launch_conv3_infer(n, d.C, d.H, d.W, d_cube, rows, cols, pixel0, d_packed + pack_off_w0b3(d.C, d.bands),
                                d_params + L.param_off[1], d_packed + pack_off_b3(d.C, d.bands, 0),
                                d_params + L.param_off[3], t, (long long*)d_labels, st));
}

size_t cmlpl_ntxent_workspace_bytes(int B, int D) { return (B < 1 || D < 1) ? 0 : ntxent_ws_floats(B, D) * 4; }

int cmlpl_ntxent_fwd_bwd(const float* d_emb_i, const float* d_emb_j, int B, int D, float temperature, float* d_loss,
                         float* d_grad_i, float* d_grad_j, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!d_emb_i || !d_emb_j || !d_loss || !d_grad_i || !d_grad_j || !d_workspace || B < 1 || D < 1 ||
      !(temperature > 0.f))
    return CMLPL_E_ARG;
  if (ntxent_ws_floats(B, D) * 4 > workspace_bytes) return CMLPL_E_WORKSPACE;
  return chk(launch_ntxent(d_emb_i, d_emb_j, B, D, temperature, d_loss, d_grad_i, d_grad_j, (float*)d_workspace,
                           (hipStream_t)stream));
}

// ---- loss_helper.py
size_t cmlpl_unsup_workspace_bytes(int B) { return B < 1 ? 0 : unsup_ws_bytes(B); }

int cmlpl_unsup_loss(const float* d_predict, int64_t* d_target, const float* d_pred_teacher, int B, int K,
                     double percent, float* d_loss, float* d_dpredict, void* d_workspace, size_t workspace_bytes,
                     void* stream) {
  if (!d_predict || !d_target || !d_pred_teacher || !d_loss || !d_dpredict || !d_workspace || B < 1 || K < 1 ||
      !(percent >= 0.0 && percent <= 100.0))
    return CMLPL_E_ARG;
  if (unsup_ws_bytes(B) > workspace_bytes) return CMLPL_E_WORKSPACE;
  return chk(launch_unsup(d_predict, (long long*)d_target, d_pred_teacher, B, K, percent, d_loss, d_dpredict,
                          d_workspace, (hipStream_t)stream));
}

int cmlpl_memobank_select(const float* d_prob, const float* d_label, const float* d_low_mask, const float* d_high_mask,
                          int N, int n_labeled, int K, int32_t* d_lists, int32_t* d_counts, void* stream) {
  if (!d_prob || !d_label || !d_low_mask || !d_high_mask || !d_lists || !d_counts || N < 1 || n_labeled < 0 ||
      n_labeled > N || K < 1 || K > 1024)
    return CMLPL_E_ARG;
  return chk(launch_mb_select(d_prob, d_label, d_low_mask, d_high_mask, N, n_labeled, K, d_lists, d_counts,
                              (hipStream_t)stream));
}

int cmlpl_memobank_proto(const float* d_rep_teacher, int N, int D, int K, const int32_t* d_lists,
                         const int32_t* d_counts, float* d_proto, void* stream) {
  if (!d_rep_teacher || !d_lists || !d_counts || !d_proto || N < 1 || D < 1 || K < 1) return CMLPL_E_ARG;
  return chk(launch_mb_proto(d_rep_teacher, D, d_lists, d_counts, N, K, d_proto, (hipStream_t)stream));
}

int cmlpl_memobank_enqueue(const float* d_rep_teacher, int N, int D, int K, const int32_t* d_lists,
                           const int32_t* d_counts, float* d_bank, int32_t* d_state, const int32_t* d_capacity,
                           int capacity_stride, void* stream) {
  if (!d_rep_teacher || !d_lists || !d_counts || !d_bank || !d_state || !d_capacity || N < 1 || D < 1 || K < 1 ||
      capacity_stride < 1)
    return CMLPL_E_ARG;
  return chk(launch_mb_enqueue(d_rep_teacher, D, d_lists, d_counts, N, K, d_bank, d_state, d_capacity,
                               capacity_stride, (hipStream_t)stream));
}

int cmlpl_memobank_push(const float* d_keys, int m, int D, float* d_bank_c, int32_t* d_state_c, int capacity,
                        void* stream) {
  if ((m > 0 && !d_keys) || !d_bank_c || !d_state_c || m < 0 || D < 1 || capacity < 1) return CMLPL_E_ARG;
  return chk(launch_mb_push(d_keys, m, D, d_bank_c, d_state_c, capacity, (hipStream_t)stream));
}

int cmlpl_memobank_infonce(const float* d_rep, int N, int D, const int32_t* d_pool, int pool_rows,
                           const int64_t* d_anchor_draw, const float* d_pos, int64_t pos_qstride,
                           const float* d_bank_c, int capacity, int bank_rows, int head, const int64_t* d_neg_draw,
                           int queries, int negatives, float temperature, float scale, float* d_lossq,
                           float* d_ganchor, float* d_drep, void* stream) {
  if (!d_rep || !d_pool || !d_anchor_draw || !d_pos || !d_bank_c || !d_neg_draw || !d_lossq || !d_ganchor ||
      N < 1 || D < 1 || pool_rows < 1 || pool_rows > N || capacity < 1 || bank_rows < 1 || bank_rows > capacity ||
      head < 0 || head >= capacity || queries < 1 || negatives < 1 || negatives > 127 || !(temperature > 0.f))
    return CMLPL_E_ARG;
  return chk(launch_mb_infonce(d_rep, D, d_pool, (const long long*)d_anchor_draw, d_pos, pos_qstride, d_bank_c,
                               capacity, head, (const long long*)d_neg_draw, queries, negatives, temperature, scale,
                               d_lossq, d_ganchor, d_drep, (hipStream_t)stream));
}

int cmlpl_memobank_loss(const cmlpl_memobank_call* c, void* stream) {
  if (!c || !c->d_rep || !c->d_rep_teacher || !c->d_prob_l || !c->d_prob_u || !c->d_label_l || !c->d_label_u ||
      !c->d_low_mask || !c->d_high_mask || !c->d_bank || !c->d_state || !c->d_capacity || !c->d_lists || !c->d_counts ||
      !c->d_proto || !c->d_lossq || !c->d_ganchor || !c->d_arow || !c->d_drep || !c->d_total)
    return CMLPL_E_ARG;
  if (c->N < 1 || c->n_labeled < 0 || c->n_labeled > c->N || c->K < 1 || c->K > 1024 || c->D < 1 || c->queries < 1 ||
      c->negatives < 1 || c->negatives > 127 || c->capacity_stride < 1 || !(c->temperature > 0.f))
    return CMLPL_E_ARG;
  if ((c->d_anchor_draw == nullptr) != (c->d_neg_draw == nullptr)) return CMLPL_E_ARG;
  if (c->d_momentum && (!c->d_momentum_on || !c->d_prototype)) return CMLPL_E_ARG;
  MbPrep p;
  p.prob_l = c->d_prob_l; p.prob_u = c->d_prob_u; p.label_l = c->d_label_l; p.label_u = c->d_label_u;
  p.low_mask = c->d_low_mask; p.high_mask = c->d_high_mask; p.rep_t = c->d_rep_teacher;
  p.N = c->N; p.Nl = c->n_labeled; p.K = c->K; p.D = c->D;
  p.lists = c->d_lists; p.counts = c->d_counts; p.proto = c->d_proto;
  p.bank = c->d_bank; p.state = c->d_state; p.caps = c->d_capacity; p.cap_stride = c->capacity_stride;
  p.keys_log = c->d_keys_log;
  MbLoss l;
  l.rep = c->d_rep; l.N = c->N; l.D = c->D; l.K = c->K; l.Q = c->queries; l.NN = c->negatives;
  l.lists = c->d_lists; l.counts = c->d_counts; l.state = c->d_state; l.proto = c->d_proto;
  l.bank = c->d_bank; l.caps = c->d_capacity; l.cap_stride = c->capacity_stride;
  l.anchor_draw = (const long long*)c->d_anchor_draw; l.neg_draw = (const long long*)c->d_neg_draw;
  l.seed = c->seed; l.call = c->call;
  l.momentum = c->d_momentum; l.momentum_on = c->d_momentum_on; l.ema = c->ema; l.prototype = c->d_prototype;
  l.temp = c->temperature;
  l.lossq = c->d_lossq; l.ganchor = c->d_ganchor; l.arow = c->d_arow; l.drep = c->d_drep; l.total = c->d_total;
  return chk(launch_mb_onepass(p, l, (hipStream_t)stream));
}

int cmlpl_memobank_sum(const float* d_v, int n, float* d_out, void* stream) {
  if (!d_v || !d_out || n < 1) return CMLPL_E_ARG;
  return chk(launch_mb_sum(d_v, n, d_out, (hipStream_t)stream));
}

int cmlpl_timing_begin(uint32_t kernel_mask, int max_launches) {
  Timing& t = g_timing;
  if (t.on || max_launches < 1) return CMLPL_E_ARG;
  t.ev.resize((size_t)max_launches * 2);
  for (auto& e : t.ev) {
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return (int)rc;
  }
  t.ids.clear(); t.used = 0; t.mask = kernel_mask; t.on = true;
  return 0;
}

int cmlpl_timing_end(double* ms_sum, int64_t* launches) {
  Timing& t = g_timing;
  if (!t.on || !ms_sum || !launches) return CMLPL_E_ARG;
  t.on = false;
  for (int i = 0; i < CMLPL_K_COUNT; ++i) { ms_sum[i] = 0.0; launches[i] = 0; }
  int rc = 0;
  for (size_t i = 0; i < t.ids.size(); ++i) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(t.ev[2 * i + 1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]);
    if (e != hipSuccess) { rc = (int)e; break; }
    ms_sum[t.ids[i]] += ms; launches[t.ids[i]] += 1;
  }
  for (auto& e : t.ev) (void)hipEventDestroy(e);
  t.ev.clear(); t.ids.clear(); t.used = 0;
  return rc;
}

int cmlpl_debug_region(const cmlpl_shape* shape, int nets, int n, const char* name, size_t* byte_offset,
                       size_t* bytes) {
  Dims d;
  if (!make_dims(shape, &d)) return CMLPL_E_SHAPE;
  if (nets < 1 || nets > 2 || n < 1 || !name || !byte_offset || !bytes) return CMLPL_E_ARG;
  NetWs w;
  char* base = (char*)(uintptr_t)4096;   // fake base: only offsets are used
  if (!carve_net(d, nets, n, base, &w)) return CMLPL_E_SHAPE;
  const size_t N = (size_t)nets * n;
  struct { const char* nm; const void* p; size_t b; } tab[] = {
      {"a0", w.a0, N * d.HW * 256}, {"p1", w.p1, N * d.P2 * 256}, {"m1", w.m1, N * d.P2 * 64},
      {"p2", w.p2, N * d.P4 * 256}, {"m2", w.m2, N * d.P4 * 64}, {"y", w.y, N * 4096},
      {"ynorm", w.ynorm, N * 4}, {"catd", w.catd, N * d.F * 4}, {"dropgen", w.dropgen, N * d.F * 4},
      {"dy", w.dy, N * 4096}, {"dp2", w.dp2, N * d.P4 * 256}, {"dp1", w.dp1, N * d.P2 * 256},
      {"da0", w.da0, N * d.HW * 256}};
  for (auto& t : tab)
    if (!strcmp(t.nm, name)) { *byte_offset = (size_t)((const char*)t.p - base); *bytes = t.b; return 0; }
  if (!strcmp(name, "xn") && nets == 2) {   // the step's augmented patch rows [2][n][C*H*W] (cmlpl_forward keeps them for cmlpl_backward)
    *byte_offset = w.bytes; *bytes = (size_t)2 * n * d.C * d.HW * 4;
    return 0;
  }
  return CMLPL_E_ARG;
}

int cmlpl_debug_reload_switches(void) {
  switches_table() = read_switches();
  return 0;
}

int cmlpl_debug_two_piece(const cmlpl_shape* shape, int nets, int n) {
  Dims d;
  if (!make_dims(shape, &d) || nets < 1 || nets > 2 || n < 1) return CMLPL_E_SHAPE;
  const int rows = nets * n, f = switches().f16x2;
  int bits = 0;
  if (conv3_fused_tail_ok(d.H, d.W, d.C, rows, d.K) && conv3_fused_head_ok(d.H, d.W, d.C, rows, d.K)) {   // the per-sample kernels
    if (conv3_h2x_both(d.H, d.W, d.C, rows, d.K)) return 1 | 2 | 4;
    return f == 2 ? 1 : f == 3 ? 2 : 0;    // (the switch's forward-only / backward-only modes: weight gradients on three pieces)
  }
  if (conv3_fused_ok(d.H, d.W, d.C, rows) || conv3_fused_bwd_ok(d.H, d.W, d.C, rows)) return 0;   // (mixed paths: switch-forced variants)
  if (conv3_h2x_general(0, d.H, d.W, rows)) bits |= 1;
  if (conv3_h2x_general(1, d.H, d.W, rows)) bits |= 2;
  if (general_h2_stats(d, rows)) bits |= 4;
  if (conv3_h2x_general(0, d.H2, d.W2, rows) && conv3_h2x_general(1, d.H2, d.W2, rows)) bits |= 8;
  return bits;
}

}  // extern "C"
