// The SHARDED training step as ONE host call: the seven stages (api.hip's entry points, in the order and with the
// arguments cmlpl_amd/distributed.py's drive_step gives them) and the four collectives between them, issued from here.
// Why: at a rank's shard the step is bound by the HOST when Python drives it (seven marshalled calls + four collective
// calls leave the device idle 80 of 234 us: profiles/r06_dist_trace_b2_64.txt).  From here the whole step is ~25 kernel /
// RCCL enqueues back to back.
//
// The collectives come in as three plain C functions (cmlpl_collectives): this library does not link RCCL.
// cmlpl_rccl_bind fills them from a librccl the process has already loaded (torch's) and an initialised ncclComm_t.
#include <dlfcn.h>

#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace {

// RCCL's three entry points as this path uses them (float32, sum); ncclDataType_t ncclFloat32 = 7, ncclRedOp_t ncclSum = 0
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_reduce_scatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int NCCL_F32 = 7, NCCL_SUM = 0;

struct RcclCtx {
  void* dl;
  void* comm;
  nccl_all_gather_fn all_gather;
  nccl_reduce_scatter_fn reduce_scatter;
  nccl_all_reduce_fn all_reduce;
};

// an RCCL failure keeps its code, moved out of the way of hipError_t and of CMLPL_E_*
inline int comm_rc(int r) { return r == 0 ? 0 : CMLPL_E_COMM - r; }

int rccl_all_gather(void* ctx, const float* send, float* recv, size_t send_count, void* stream) {
  RcclCtx* c = (RcclCtx*)ctx;
  return comm_rc(c->all_gather(send, recv, send_count, NCCL_F32, c->comm, (hipStream_t)stream));
}
int rccl_reduce_scatter(void* ctx, const float* send, float* recv, size_t recv_count, void* stream) {
  RcclCtx* c = (RcclCtx*)ctx;
  return comm_rc(c->reduce_scatter(send, recv, recv_count, NCCL_F32, NCCL_SUM, c->comm, (hipStream_t)stream));
}
int rccl_all_reduce(void* ctx, float* buf, size_t count, void* stream) {
  RcclCtx* c = (RcclCtx*)ctx;
  return comm_rc(c->all_reduce(buf, buf, count, NCCL_F32, NCCL_SUM, c->comm, (hipStream_t)stream));
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// an exchange that runs beside the next stage: [record on the step's stream] -> side stream waits -> collective ->
// [record on the side stream]; the stage that reads its output makes the step's stream wait for the second event
template <class F>
int issue_async(const cmlpl_collectives* cl, int pair, hipStream_t st, F&& call) {
  if (cl->side_stream == nullptr) return call(st);               // no side stream: in order on the step's stream
  hipEvent_t e0 = (hipEvent_t)cl->events[2 * pair], e1 = (hipEvent_t)cl->events[2 * pair + 1];
  hipStream_t side = (hipStream_t)cl->side_stream;
  int rc;
  if ((rc = hip_rc(hipEventRecord(e0, st)))) return rc;
  if ((rc = hip_rc(hipStreamWaitEvent(side, e0, 0)))) return rc;
  if ((rc = call(side))) return rc;
  return hip_rc(hipEventRecord(e1, side));
}
inline int wait_async(const cmlpl_collectives* cl, int pair, hipStream_t st) {
  if (cl->side_stream == nullptr) return 0;
  return hip_rc(hipStreamWaitEvent(st, (hipEvent_t)cl->events[2 * pair + 1], 0));
}

}  // namespace

extern "C" {

int cmlpl_rccl_bind(const char* librccl_path, void* nccl_comm, cmlpl_collectives* out) {
  if (!librccl_path || !nccl_comm || !out) return CMLPL_E_ARG;
  void* dl = dlopen(librccl_path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) return CMLPL_E_COMM;
  RcclCtx* c = (RcclCtx*)calloc(1, sizeof(RcclCtx));
  if (!c) { dlclose(dl); return CMLPL_E_ARG; }
  c->dl = dl; c->comm = nccl_comm;
  c->all_gather = (nccl_all_gather_fn)dlsym(dl, "ncclAllGather");
  c->reduce_scatter = (nccl_reduce_scatter_fn)dlsym(dl, "ncclReduceScatter");
  c->all_reduce = (nccl_all_reduce_fn)dlsym(dl, "ncclAllReduce");
  if (!c->all_gather || !c->reduce_scatter || !c->all_reduce) { free(c); dlclose(dl); return CMLPL_E_COMM; }
  cmlpl_collectives o;
  o.ctx = c; o.all_gather = rccl_all_gather; o.reduce_scatter = rccl_reduce_scatter; o.all_reduce = rccl_all_reduce;
  o.side_stream = nullptr;
  for (int i = 0; i < 4; ++i) o.events[i] = nullptr;
  hipStream_t side;
  hipError_t e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
  for (int i = 0; i < 4 && e == hipSuccess; ++i) {
    hipEvent_t ev;
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) o.events[i] = ev;
  }
  if (e != hipSuccess) {
    for (int i = 0; i < 4; ++i) if (o.events[i]) (void)hipEventDestroy((hipEvent_t)o.events[i]);
    free(c); dlclose(dl);
    return (int)e;
  }
  o.side_stream = side;
  *out = o;
  return 0;
}

int cmlpl_rccl_unbind(cmlpl_collectives* c) {
  if (!c || !c->ctx) return CMLPL_E_ARG;
  for (int i = 0; i < 4; ++i) if (c->events[i]) (void)hipEventDestroy((hipEvent_t)c->events[i]);
  if (c->side_stream) (void)hipStreamDestroy((hipStream_t)c->side_stream);
  RcclCtx* x = (RcclCtx*)c->ctx;
  dlclose(x->dl);
  free(x);
  c->ctx = nullptr; c->side_stream = nullptr;
  for (int i = 0; i < 4; ++i) c->events[i] = nullptr;
  return 0;
}

int cmlpl_dist_step(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_dist_io* io,
                    const cmlpl_dist_step_args* args, const cmlpl_collectives* coll, void* stream) {
  if (!shape || !hp || !io || !args) return CMLPL_E_ARG;
  if (io->d_dyn_table || io->d_dyn_cursor) return CMLPL_E_ARG;     // (the replayed step: cmlpl_dist_stage_graph_create)
  const cmlpl_gathered& g = io->gathered;
  const int W = g.world, bt_l = g.bt_local, btu_l = g.btu_local, n_l = bt_l + btu_l, K = shape->K;
  if (W < 1 || bt_l < 1 || btu_l < 1 || io->batch.bt != bt_l || io->batch.btu != btu_l || io->shard.nlab != bt_l ||
      io->shard.nunl != btu_l || io->shard.bt_g != bt_l * W || io->shard.btu_g != btu_l * W)
    return CMLPL_E_ARG;
  if (!io->d_feat_l || !io->d_labels_f || !io->d_logits_l || !io->d_dlogits || !io->d_dfeat || !io->d_probs_l ||
      !io->d_probs_g || !io->d_dfeat_w_partial || !io->d_grads || !io->d_scalars || !g.d_recv_feat ||
      g.d_logits_local != io->d_logits_l || args->scalars_row < 0 || io->probs_shard_rows != btu_l)
    return CMLPL_E_ARG;
  // this rank's block of the exchange buffer is ONE piece, [feat | labels]
  const size_t nf = (size_t)2 * n_l * 1024, pack_len = nf + bt_l, np = (size_t)4 * btu_l * K;
  if (io->d_labels_f != io->d_feat_l + nf) return CMLPL_E_ARG;
  float* dfeat_own = io->d_dfeat + ((size_t)n_l + bt_l) * 1024;       // net 1's unlabelled rows: the reduce-scatter's output
  if (coll == nullptr) {
    // no communicator: one rank whose exchange buffers ARE its own blocks (the collectives are identities)
    if (W != 1 || g.d_recv_feat != io->d_feat_l || io->d_probs_g != io->d_probs_l || io->d_dfeat_w_partial != dfeat_own)
      return CMLPL_E_ARG;
  } else if (!coll->all_gather || !coll->reduce_scatter || !coll->all_reduce ||
             (coll->side_stream && (!coll->events[0] || !coll->events[1] || !coll->events[2] || !coll->events[3]))) {
    return CMLPL_E_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int train = 1;
  cmlpl_banks banks = io->banks;
  float* scalars = io->d_scalars + (size_t)16 * args->scalars_row;
  int rc;
  // spectral: the embeddings (+ labels) land in this rank's block; their all-gather runs under the convolutions
  if ((rc = cmlpl_forward_spectral(shape, hp, &io->batch, &io->shard, io->d_params, io->seed, args->step, io->d_feat_l,
                                   io->d_labels_f, io->d_workspace, io->workspace_bytes, stream))) return rc;
  if (coll && (rc = issue_async(coll, 0, st, [&](hipStream_t s) {
        return coll->all_gather(coll->ctx, io->d_feat_l, (float*)g.d_recv_feat, pack_len, s); }))) return rc;
  if ((rc = cmlpl_forward_spatial(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, args->d_dropmask, train,
                                  io->seed, args->step, io->d_logits_l, io->d_workspace, io->workspace_bytes, stream))) return rc;
  if (coll && (rc = wait_async(coll, 0, st))) return rc;
  if ((rc = cmlpl_loss_phase1_g(shape, &io->shard, &g, &banks, args->smooth, args->adap_mask, hp, io->d_dlogits, io->d_dfeat,
                                io->d_probs_l, io->d_loss_workspace, io->loss_workspace_bytes, stream))) return rc;
  if (coll && (rc = coll->all_gather(coll->ctx, io->d_probs_l, (float*)io->d_probs_g, np, stream))) return rc;
  if ((rc = cmlpl_loss_phase2_g(shape, &io->shard, &g, &banks, args->smooth, args->adap_mask, hp, io->d_probs_g, btu_l, scalars,
                                io->d_dfeat, io->d_dfeat_w_partial, io->d_loss_workspace, io->loss_workspace_bytes, stream)))
    return rc;
  // backward of "gather the keys": sum the ranks' partials, keep this rank's rows -- under the data-gradient chain
  if (coll && (rc = issue_async(coll, 1, st, [&](hipStream_t s) {
        return coll->reduce_scatter(coll->ctx, io->d_dfeat_w_partial, dfeat_own, (size_t)btu_l * 1024, s); }))) return rc;
  if ((rc = cmlpl_backward_data(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, args->d_dropmask, train,
                                io->seed, args->step, io->d_dlogits, io->d_workspace, io->workspace_bytes, stream))) return rc;
  if (coll && (rc = wait_async(coll, 1, st))) return rc;
  if ((rc = cmlpl_backward_weights(shape, hp, &io->batch, &io->shard, io->d_params, io->d_packed, args->d_dropmask, train,
                                   io->seed, args->step, io->d_dlogits, io->d_dfeat, io->d_grads, io->grad_stride,
                                   io->d_workspace, io->workspace_bytes, stream))) return rc;
  if (coll && (rc = coll->all_reduce(coll->ctx, io->d_grads, (size_t)2 * io->grad_stride, stream))) return rc;
  if (!args->apply_update) return 0;
  if (!io->d_m || !io->d_v || args->adam_t < 1) return CMLPL_E_ARG;
  cmlpl_layout_t L;
  if ((rc = cmlpl_layout(shape, &L))) return rc;
  return cmlpl_adam_step(shape, 2, io->d_params, L.param_total, io->d_grads, io->grad_stride, io->d_m, io->d_v, args->adam_t,
                         hp, io->d_packed, stream);
}

}  // extern "C"
