/* cmlpl.h -- C ABI of the MI355X-native CMLPL training hot path (libcmlpl_hip.so).
 *
 * The reference (liuli33/CMLPL) has no FFI/plugin interface: its hot path is inline
 * PyTorch in train.py:150-279 and tools/models.py:97-152.  This header is therefore
 * the boundary the reference's Python would bind via ctypes; each
 * entry point names the reference lines it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer named d_* is DEVICE memory
 *     owned by the caller (no ownership transfer, no hidden allocation);
 *   - every call takes a hipStream_t as `void* stream`, is asynchronous and re-entrant
 *     across streams (no implicit synchronisation, graph-capturable);
 *   - return 0 = ok, negative = argument error (CMLPL_E_*), positive = hipError_t;
 *   - every tensor is fp32 and every result is fp32-grade; labels are int64 (torch.long).  The 3x3 / 1x1 convolutions
 *     (forward, data and weight gradients) form each fp32 product from three exact bf16 pieces per operand, six of
 *     the nine piece products (the dropped ones are below 2^-21 |a b| in the worst case, 0.7 * 2^-24 on average),
 *     fp32 accumulation: measured against fp64 at the level of an fp32 fma chain.  An infinite
 *     operand of such a product becomes NaN (a finite fp32 path would keep the infinity); NaN stays NaN.
 *   - the library expects ONE calling thread per process at a time (one process per GPU is the deployment
 *     model); the lazily-set kernel attributes are tracked per device, so engines on several devices of one
 *     process work, but cmlpl_timing_begin/_end state is process-global and not thread-safe.
 *   - `nets` is 1 or 2: kernels are batched over the two networks Base/Base1
 *     (train.py:118-125) through a grid dimension; per-network buffers are
 *     laid out [net][...] with the strides returned by cmlpl_layout().
 */
#ifndef CMLPL_H
#define CMLPL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMLPL_ABI_VERSION 4
#define CMLPL_FEAT_DIM 1024 /* tools/models.py:119 */
#define CMLPL_CONV_CH 64    /* tools/models.py:102-107 */

enum {
  CMLPL_E_ARG = -1,      /* null pointer / bad size */
  CMLPL_E_SHAPE = -2,    /* unsupported shape (e.g. K > 64, window too large for LDS) */
  CMLPL_E_WORKSPACE = -3,/* workspace too small */
  CMLPL_E_COMM = -4      /* cmlpl_dist_step: a collective failed (RCCL's ncclResult_t r > 0 comes back as -4 - r) */
};

/* Shape of one BaseNet2 (tools/models.py:98-128, generalised per SURVEY.md section 0). */
typedef struct cmlpl_shape {
  int32_t C;     /* conv0 input channels (reference literal 60, models.py:102) */
  int32_t H, W;  /* window size (reference: 20x20)                            */
  int32_t bands; /* num_features (models.py:121)                              */
  int32_t K;     /* num_classes  (models.py:127), 1..64                       */
} cmlpl_shape;

/* Hyper-parameters consumed inside the step (train.py:356-379 flags + literals). */
typedef struct cmlpl_hparams {
  float lr, beta1, beta2, eps;      /* torch.optim.Adam defaults, train.py:131-132        */
  float temperature, alpha;         /* train.py:374,371                                    */
  float noise_sigma, dropout_p;     /* train.py:378,377                                    */
  float w_contrast, w_mutual;       /* literals 0.5 / 4, train.py:266,270                  */
  float pos_thr, neg_thr;           /* literals 0.8 / 0.3, train.py:251,254                */
} cmlpl_hparams;

/* Flat parameter buffer of ONE network: the 16 state_dict tensors of BaseNet2 in
 * canonical PyTorch layouts, LIVE tensors first:
 *   0 conv0.weight[64,C,1,1] 1 conv0.bias 2 conv1.weight[64,64,3,3] 3 conv1.bias
 *   4 conv2.weight 5 conv2.bias 6 feat_spe.weight[1024,bands] 7 feat_spe.bias
 *   8 classifier.weight[K,cls_in] 9 classifier.bias | 10..15 feat_ss*, dead (models.py:122-126)
 * Each tensor's offset is a multiple of 4 floats. */
#define CMLPL_NUM_TENSORS 16
#define CMLPL_NUM_LIVE 10
typedef struct cmlpl_layout_t {
  int64_t param_off[CMLPL_NUM_TENSORS];   /* element offsets in the per-net flat buffer  */
  int64_t param_numel[CMLPL_NUM_TENSORS];
  int64_t param_total;                    /* floats per net (all 16 tensors)             */
  int64_t param_live;                     /* floats per net covered by Adam (tensors 0-9) */
  int64_t packed_total;                   /* floats per net of kernel-side re-packed weights (3x3 and conv0 weights as
                                             split-bf16 MFMA fragments, k-major copies of conv0 / feat_spe); opaque to
                                             the caller; written by cmlpl_pack_weights and by cmlpl_adam_step */
  int32_t cls_in;                         /* 64*(H/2/2)*(W/2/2) + 1024                    */
  int32_t reserved;
} cmlpl_layout_t;

int cmlpl_abi_version(void);
/* 16-hex-digit hash of the kernel sources and this header the binary was built from ("unknown" for a build outside
 * cmlpl_amd/build_ext.py).  The Python binding compares it with the hash of the sources next to it and refuses a
 * stale binary (cmlpl_amd/_lib.py); the reference has no counterpart (it has no native code). */
const char* cmlpl_source_hash(void);
int cmlpl_layout(const cmlpl_shape* shape, cmlpl_layout_t* out);

/* Bytes of device workspace needed by the calls below for up to `n` rows per network
 * (n = labelled + unlabelled), `nets` networks and a bank of `bank_rows` rows. */
size_t cmlpl_workspace_bytes(const cmlpl_shape* shape, int nets, int n, int bank_rows);

/* Re-pack the weights of `nets` networks into the kernel-side layouts (every entry of d_packed, pad rows included).
 * Call it once after parameters are set or loaded; cmlpl_adam_step then keeps every non-pad entry current itself
 * (it never touches the zero pad rows this call writes, so a packed buffer must have been through this call once). */
int cmlpl_pack_weights(const cmlpl_shape* shape, int nets, const float* d_params, int64_t param_stride,
                       float* d_packed, void* stream);

/* Row shard of a data-parallel step: the global batch has bt_g labelled and
 * btu_g unlabelled rows; this rank owns labelled rows [lab0, lab0+nlab) and unlabelled rows
 * [unl0, unl0+nunl).  One GPU: {bt, btu, 0, bt, 0, btu} (or NULL where allowed).  In-kernel random
 * streams are keyed by the GLOBAL sample index, so noise/dropout do not depend on the sharding. */
typedef struct cmlpl_shard {
  int32_t bt_g, btu_g;
  int32_t lab0, nlab;
  int32_t unl0, nunl;
} cmlpl_shard;

/* Input augmentation + batch concat: train.py:157-158,163-164,170-171,173-174,181-184.
 *   xn[net] = cat(XPl, XPu) + sigma * N(0,1),  sn[net] = cat(Xl, Xu) + sigma * N(0,1)
 * d_noise: NULL (in-kernel counter-based draws -- PCG4D hash + Box-Muller, keyed by seed/step/stream/global sample) or 8 device pointers in the
 * reference's draw order [XPl/0, Xl/0, XPl/1, Xl/1, XPu/0, Xu/0, XPu/1, Xu/1].
 * With bt == 0 or sigma == 0 it is a plain (concatenating) copy. */
int cmlpl_augment(const cmlpl_shape* shape, int nets, int bt, int btu,
                  const float* d_xpl, const float* d_xl, const float* d_xpu, const float* d_xu,
                  const float* const* noise8, float sigma, uint64_t seed, uint64_t step,
                  const cmlpl_shard* shard /* NULL = one GPU */, float* d_xn, float* d_sn,
                  float* d_snT /* optional [nets][bands][n] copy of sn, k-major for feat_spe */, void* stream);

/* BaseNet2.forward (tools/models.py:130-152) for `nets` networks on rows [n].
 *   d_xn [nets][n][C][H*W], d_sn [nets][n][bands]  (already augmented)
 *   d_dropmask: [nets][n][cls_in] multiplier (0 or 1/(1-p)); NULL with dropout_p == 0 or
 *               train == 0 -> no dropout; NULL with train != 0 and p > 0 -> Philox mask
 *   outputs: d_logits [nets][n][K], d_feat [nets][n][1024]; activations needed by
 *   cmlpl_basenet2_bwd are kept in d_workspace. */
int cmlpl_basenet2_fwd(const cmlpl_shape* shape, int nets, int n,
                       const float* d_params, int64_t param_stride, const float* d_packed,
                       const float* d_xn, const float* d_sn,
                       const float* d_snT /* optional: sn transposed [nets][bands][n] (from cmlpl_augment) */,
                       const float* d_dropmask,
                       float dropout_p, int train, uint64_t seed, uint64_t step,
                       const cmlpl_shard* shard /* NULL = rows are global samples 0..n-1 */,
                       float* d_logits, float* d_feat, void* d_workspace, size_t workspace_bytes,
                       void* stream);

/* Backward of the above w.r.t. the 10 live tensors (autograd of train.py:267,271).
 *   d_dlogits [nets][n][K], d_dfeat [nets][n][1024] (may be NULL = zero)
 *   d_grads   [nets][grad_stride] flat, same offsets as the parameters (tensors 0-9 written). */
int cmlpl_basenet2_bwd(const cmlpl_shape* shape, int nets, int n,
                       const float* d_params, int64_t param_stride, const float* d_packed,
                       const float* d_xn, const float* d_sn,
                       const float* d_dropmask, float dropout_p, int train, /* as given to _fwd */
                       const float* d_dlogits, const float* d_dfeat,
                       float* d_grads, int64_t grad_stride,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* One training batch as the reference holds it (train.py:155-171): labelled and unlabelled rows in their own
 * buffers (the concat of train.py:173-174,183-184 is an index computation in the kernels, not a copy) and the
 * augmentation noise either drawn in-kernel (PCG4D hash + Box-Muller, noise8 == NULL) or given as the 8 tensors of the
 * reference's draw order (see cmlpl_augment). */
typedef struct cmlpl_batch {
  const float* d_xpl; const float* d_xl;   /* [bt][C][H][W], [bt][bands]   */
  const float* d_xpu; const float* d_xu;   /* [btu][C][H][W], [btu][bands] */
  const int64_t* d_labels;                 /* [bt] (may be NULL where no entry point reads it) */
  const float* const* noise8;
  int32_t bt, btu;
  /* ABI 3 -- batches BY INDEX (hsi_loader.py:109-133 hands out rows of XP.npy / X.npy / Y.npy by index; the reference's
   * DataLoader then copies them into a batch tensor): when given, labelled row s of the batch is row d_lab_idx[s] of
   * d_xpl / d_xl / d_labels and unlabelled row s is row d_unl_idx[s] of d_xpu / d_xu -- the kernels read the rows where
   * the resident split lies, no gathered copy of the batch exists.  NULL = rows 0 .. bt-1 / 0 .. btu-1 as before.
   * (The explicit noise tensors and dropout masks of parity mode stay indexed by batch row.) */
  const int64_t* d_lab_idx; const int64_t* d_unl_idx;
} cmlpl_batch;

/* One step's scalars in DEVICE memory (ABI 3): what changes from step to step, for a step captured ONCE in a hipGraph
 * and replayed (SURVEY.md section 7 stage 6; the reference's loop passes them as Python scalars, train.py:146-150,
 * 212,221,234-237).  The caller fills a table of rows ahead of time (e.g. one epoch) and passes it with a device
 * cursor: a launch of cmlpl_train_step with d_dyn_table != NULL takes these values from the table instead of from
 * cmlpl_step_io, and advances the cursor itself (in its weight-gradient reduce launch), so that replaying the captured
 * launch sequence walks the table with no host work per step.
 * Table layout: row 0 is the WORKING COPY of the current step's row (the kernels read it there: one dependent load
 * instead of cursor -> row); the rows of steps 1, 2, .. follow at indices 1, 2, ..  To start (or restart) a run of
 * steps the caller writes its rows at 1 .. k, copies row 1 into row 0 as well and sets *d_dyn_cursor = 1; every step
 * then moves the cursor on and copies the next row into row 0 (so the table needs one row beyond the last step's:
 * k + 2 rows for k steps; the extra row's contents do not matter). */
typedef struct cmlpl_dyn {
  uint64_t step;            /* counter of the in-kernel random streams (cmlpl_step_io.step)            */
  int64_t adam_t;           /* 1-based Adam step                                                       */
  int64_t lab_off, unl_off; /* offsets added to the row number before d_lab_idx / d_unl_idx are read   */
  int32_t ptr[2];           /* bank write pointers BEFORE the step (train.py:234,237)                  */
  int32_t smooth;           /* train.py:212 gate                                                       */
  float adap_mask;          /* train.py:221                                                            */
  int32_t hist_row;         /* the step's scalars go to d_scalars + 16 * hist_row                      */
  float adam_step_size;     /* lr / (1 - beta1^adam_t)   } torch.optim.Adam's bias corrections, formed in double  */
  float adam_bc2_sqrt;      /* sqrt(1 - beta2^adam_t)    } on the host: cmlpl_dyn_adam() fills both               */
  int32_t reserved;
} cmlpl_dyn;
/* the two Adam scalars of a table row, exactly as cmlpl_adam_step forms them from (hp, t) */
int cmlpl_dyn_adam(const cmlpl_hparams* hp, int64_t adam_t, float* step_size, float* bc2_sqrt);

/* Both networks' forward on one batch (train.py:157-189): augmentation + BaseNet2.forward for Base and Base1.
 * Where the window fits the per-sample kernels (H*W <= 128) the patches are augmented inside the fused forward, which
 * also stores the augmented rows in the workspace (2 * n * C * H * W floats written per step); cmlpl_backward lands
 * them for conv0's weight gradient.  It must therefore be given the same batch / seed / step / shard and the SAME
 * workspace (sized by cmlpl_workspace_bytes(shape, 2, bt + btu, bank_rows)), which holds the saved activations and
 * those rows in between -- and NO other cmlpl_forward may run on that workspace between the two (an evaluation or
 * teacher pass in between needs a workspace of its own: it would overwrite the rows the weight gradient reads).
 *   hp: noise_sigma and dropout_p are used.  Outputs as cmlpl_basenet2_fwd / _bwd with nets = 2. */
int cmlpl_forward(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                  const cmlpl_shard* shard /* NULL = one GPU */,
                  const float* d_params /* [2][param_total] */, const float* d_packed /* [2][packed_total] */,
                  const float* d_dropmask, int train, uint64_t seed, uint64_t step,
                  float* d_logits, float* d_feat,
                  float* d_labels_f /* optional [bt]: the labels as float, written for the packed exchange buffer */,
                  void* d_workspace, size_t workspace_bytes, void* stream);
int cmlpl_backward(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                   const cmlpl_shard* shard,
                   const float* d_params, const float* d_packed,
                   const float* d_dropmask, int train, uint64_t seed, uint64_t step,
                   const float* d_dlogits, const float* d_dfeat, float* d_grads, int64_t grad_stride,
                   void* d_workspace, size_t workspace_bytes, void* stream);

/* ABI 4: cmlpl_forward / cmlpl_backward in two parts each, for a step that exchanges data between them (the sharded step
 * of cmlpl_amd/distributed.py).  The split follows what depends on what in tools/models.py:130-152:
 *   cmlpl_forward_spectral : augmented spectra -> feat_spe + ReLU -> L2 normalisation (:142-146) -> d_feat [2][n][1024]
 *                            (+ d_labels_f).  The embeddings depend on NOTHING the convolutions compute: a sharded step
 *                            starts their all-gather here and lets it run under the spatial part.
 *   cmlpl_forward_spatial  : conv0 .. conv2, pooling, concat / dropout / classifier (:132-141,144,147-150) -> d_logits.
 *                            Same workspace, after the spectral part (it reads the spectral ReLU output there).
 *   cmlpl_backward_data    : the data-gradient chain (classifier, conv2, conv1, conv0 partials) and the 3x3 weight-gradient
 *                            partials -- needs d_dlogits ONLY (:144-150): the column-side feature gradient can still be
 *                            in its reduce-scatter.
 *   cmlpl_backward_weights : dy takes its feature-gradient share (through the normalisation's backward; bit-identical to
 *                            the one-part head), partials are reduced, classifier / feat_spe weight gradients -> d_grads.
 * Both halves of a pair take the same batch / seed / step / shard / workspace as the whole call would. */
int cmlpl_forward_spectral(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                           const cmlpl_shard* shard, const float* d_params, uint64_t seed, uint64_t step,
                           float* d_feat, float* d_labels_f /* optional [bt] */,
                           void* d_workspace, size_t workspace_bytes, void* stream);
int cmlpl_forward_spatial(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                          const cmlpl_shard* shard, const float* d_params, const float* d_packed,
                          const float* d_dropmask, int train, uint64_t seed, uint64_t step, float* d_logits,
                          void* d_workspace, size_t workspace_bytes, void* stream);
int cmlpl_backward_data(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                        const cmlpl_shard* shard, const float* d_params, const float* d_packed,
                        const float* d_dropmask, int train, uint64_t seed, uint64_t step, const float* d_dlogits,
                        void* d_workspace, size_t workspace_bytes, void* stream);
int cmlpl_backward_weights(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_batch* batch,
                           const cmlpl_shard* shard, const float* d_params, const float* d_packed,
                           const float* d_dropmask, int train, uint64_t seed, uint64_t step, const float* d_dlogits,
                           const float* d_dfeat, float* d_grads, int64_t grad_stride,
                           void* d_workspace, size_t workspace_bytes, void* stream);

/* State of the two memory banks (train.py:138-145). */
typedef struct cmlpl_banks {
  float* d_feats[2];  /* [Q][1024] queue_feats, queue_feats1 */
  float* d_probs[2];  /* [Q][K]    queue_probs, queue_probs1 */
  int32_t Q;          /* rows, 5*labeled_batch_size*2 (train.py:138) */
  int32_t ptr[2];     /* write pointers BEFORE this step (host keeps train.py:234,237) */
} cmlpl_banks;

/* The loss block, train.py:191-266, forward + analytic backward, plus the bank write
 * (train.py:223-237; rows are written modulo Q).
 *   d_logits [2][n][K], d_feat [2][n][1024]: net 0 = Base ("s"), net 1 = Base1 ("w");
 *   rows [0,bt) labelled, [bt,n) unlabelled; d_labels int64 [bt]
 *   smooth    : train.py:212 gate (epoch > 0 or batch_index > queue_batch)
 *   adap_mask : thr * exp(-0.5*(epoch/num_epochs)^2), train.py:147-148,221
 *   outputs   : d_scalars[16] = {ctr_s,total_s,cls_s,con_s,acc, total_w,cls_w,con_w,ctr_w,
 *                                n_mask_w,n_mask_s,n_pos,n_neg,0,0,0}   (train.py:274-278)
 *               d_dlogits [2][n][K], d_dfeat [2][n][1024]
 *               d_probs [4][btu][K] = {p_w, p_s smoothed, p_w0, p_s0}  (required scratch/output). */
int cmlpl_loss_fwd_bwd(const cmlpl_shape* shape, int bt, int btu,
                       const float* d_logits, const float* d_feat, const int64_t* d_labels,
                       const cmlpl_banks* banks, int smooth, float adap_mask,
                       const cmlpl_hparams* hp,
                       float* d_scalars, float* d_dlogits, float* d_dfeat, float* d_probs,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* The same loss block split at the one point where a data-parallel step must exchange data.
 * Inputs are the GLOBAL logits/feats/labels (rows ordered [labelled of all ranks ; unlabelled of all
 * ranks]); outputs cover this rank's rows only.
 *   phase 1: similarity tiles (local rows x banks / all keys), per-row softmax, CE, smoothing, masks,
 *            mutual loss; writes d_dlogits [2][nlab+nunl][K] and d_probs_local [4][nunl][K]
 *   -- caller all-gathers d_probs_local into d_probs_global (rank-major [W][4][nunl][K]) --
 *   phase 2: pseudo-label graph + contrastive loss for the local rows, bank write of the global batch,
 *            this rank's additive share of d_scalars[16], d_dfeat [2][nlab+nunl][1024] (net-0 rows final;
 *            net-1 rows are to be overwritten by the caller with its reduce-scattered slice of)
 *            d_dfeat_w_partial [btu_g][1024] = this rank's partial of the column-side gradient. */
size_t cmlpl_loss_workspace_bytes(const cmlpl_shape* shape, const cmlpl_shard* shard, int bank_rows);
int cmlpl_loss_phase1(const cmlpl_shape* shape, const cmlpl_shard* shard,
                      const float* d_logits, const float* d_feat, const int64_t* d_labels,
                      const cmlpl_banks* banks, int smooth, float adap_mask, const cmlpl_hparams* hp,
                      float* d_dlogits, float* d_dfeat /* labelled rows are zeroed here */, float* d_probs_local,
                      void* d_workspace, size_t workspace_bytes, void* stream);
int cmlpl_loss_phase2(const cmlpl_shape* shape, const cmlpl_shard* shard,
                      const float* d_logits, const float* d_feat, const int64_t* d_labels,
                      const cmlpl_banks* banks, int smooth, float adap_mask, const cmlpl_hparams* hp,
                      const float* d_probs_global, int probs_shard_rows,
                      float* d_scalars, float* d_dfeat, float* d_dfeat_w_partial,
                      void* d_workspace, size_t workspace_bytes, void* stream);

/* The loss phases of the SHARDED step (ABI 4).  What crosses ranks is read WHERE THE ALL-GATHER LEFT IT (no re-ordering
 * copy): d_recv_feat holds `world` rank-major blocks [ 2*n_l*1024 feat | bt_l labels as float ] (n_l = bt_local +
 * btu_local, per-rank rows [labelled ; unlabelled]) -- the buffer cmlpl_forward_spectral's d_feat and d_labels_f are laid
 * out for, gathered while the convolutions run.  The LOGITS are never gathered: d_logits_local [2][n_l][K] are this
 * rank's rows (cmlpl_forward_spatial's output), the only ones phase 1 reads; the bank write needs the other ranks'
 * un-smoothed probabilities and takes them from the gathered probabilities, so it rides with phase 2.
 * Otherwise identical to cmlpl_loss_phase1 / _phase2. */
typedef struct cmlpl_gathered {
  const float* d_recv_feat;
  const float* d_logits_local;
  int32_t world, bt_local, btu_local;
} cmlpl_gathered;
int cmlpl_loss_phase1_g(const cmlpl_shape* shape, const cmlpl_shard* shard, const cmlpl_gathered* gathered,
                        const cmlpl_banks* banks, int smooth, float adap_mask, const cmlpl_hparams* hp,
                        float* d_dlogits, float* d_dfeat, float* d_probs_local,
                        void* d_workspace, size_t workspace_bytes, void* stream);
int cmlpl_loss_phase2_g(const cmlpl_shape* shape, const cmlpl_shard* shard, const cmlpl_gathered* gathered,
                        const cmlpl_banks* banks, int smooth, float adap_mask, const cmlpl_hparams* hp,
                        const float* d_probs_global, int probs_shard_rows,
                        float* d_scalars, float* d_dfeat, float* d_dfeat_w_partial,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* Inspection aid: re-order gathered blocks into the global row order [labelled of all ranks ; unlabelled of all ranks]:
 * d_gathered_feat = [world][ 2*n_l*1024 feat | bt_l labels as float ] (the step's exchange buffer), d_gathered_logits =
 * [world][2*n_l*K] (the step does not gather logits: a caller that wants the global logits gathers them for this call). */
int cmlpl_dist_unpack(const cmlpl_shape* shape, int world, int bt_local, int btu_local,
                      const float* d_gathered_feat, const float* d_gathered_logits,
                      float* d_logits_g, float* d_feat_g, int64_t* d_labels_g, void* stream);

/* torch.optim.Adam.step for `nets` flat buffers (train.py:268,272); `t` is the 1-based
 * step count.  Also refreshes the packed conv weights when d_packed != NULL. */
int cmlpl_adam_step(const cmlpl_shape* shape, int nets, float* d_params, int64_t param_stride,
                    const float* d_grads, int64_t grad_stride, float* d_m, float* d_v,
                    int64_t t, const cmlpl_hparams* hp, float* d_packed, void* stream);

/* One whole training step, train.py:150-278: augment -> 2x forward -> loss block ->
 * bank write -> 2x backward -> 2x Adam.  d_state buffers persist across steps. */
typedef struct cmlpl_step_io {
  const float* d_xpl; const float* d_xl; const int64_t* d_labels;  /* labelled batch  */
  const float* d_xpu; const float* d_xu;                           /* unlabelled batch */
  const float* const* noise8;   /* NULL = in-kernel draws (PCG4D hash + Box-Muller)     */
  const float* d_dropmask;      /* NULL = Philox4x32-10 mask; else [2][n][cls_in]       */
  float* d_params; float* d_m; float* d_v; float* d_grads; float* d_packed;  /* [2][param_total] (packed: [2][packed_total]) */
  cmlpl_banks banks;
  float* d_scalars;             /* [16] as in cmlpl_loss_fwd_bwd                        */
  float* d_logits; float* d_feat; /* [2][n][K], [2][n][1024] (outputs)                  */
  void* d_workspace; size_t workspace_bytes;
  int32_t bt, btu;
  int32_t smooth; float adap_mask;
  int64_t adam_t;               /* 1-based                                               */
  uint64_t seed, step;          /* key / counter of the in-kernel random streams         */
  int32_t apply_update;         /* 0 = stop after the gradients                          */
  int32_t reserved;
  /* ABI 3 */
  const int64_t* d_lab_idx; const int64_t* d_unl_idx;   /* as in cmlpl_batch; NULL = consecutive rows */
  const cmlpl_dyn* d_dyn_table; int32_t* d_dyn_cursor;   /* NULL = the by-value fields above are used  */
} cmlpl_step_io;

int cmlpl_train_step(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_step_io* io,
                     void* stream);

/* The step as a replayable hipGraph (SURVEY.md section 7 stage 6): _create captures ONE cmlpl_train_step(io) on
 * `stream` (io->d_dyn_table / d_dyn_cursor must be set: everything that changes between steps then lives in device
 * memory) and instantiates it; _launch enqueues one replay = one training step, no other host work.  Run one eager
 * step first (lazily-set kernel attributes cannot be set while capturing).  The handle owns the graph; _destroy frees it. */
int cmlpl_step_graph_create(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_step_io* io, void* stream,
                            void** graph_out);
int cmlpl_step_graph_launch(void* graph, void* stream);
int cmlpl_step_graph_destroy(void* graph);

/* The seven stages of the SHARDED step (spectral | spatial | loss phase 1 | loss phase 2 | backward data | backward
 * weights | update; the collectives run between them, in the caller's hands: cmlpl_amd/distributed.py) as replayable
 * hipGraphs: what cmlpl_forward_spectral, cmlpl_forward_spatial, cmlpl_loss_phase1_g, cmlpl_loss_phase2_g,
 * cmlpl_backward_data, cmlpl_backward_weights and cmlpl_adam_step launch, captured once per stage with every per-step
 * scalar read from the cmlpl_dyn table (as a captured cmlpl_train_step does).  The backward-weights stage advances the
 * cursor (its reduce launch), the update stage reads the finished step's row.  Batches by index only (d_lab_idx /
 * d_unl_idx: this rank's slice starts at the row's lab_off / unl_off); in-kernel random streams only.  Launch / destroy a
 * stage's graph with cmlpl_step_graph_launch / cmlpl_step_graph_destroy. */
typedef struct cmlpl_dist_io {
  cmlpl_batch batch;            /* this rank's rows (bt, btu PER RANK), by index                                   */
  cmlpl_shard shard;            /* this rank's place in the global batch                                           */
  cmlpl_gathered gathered;      /* the gathered embeddings / labels and this rank's logits                          */
  cmlpl_banks banks;            /* (ptr[] unused: the table carries the pointers)                                  */
  float* d_params; float* d_m; float* d_v; float* d_packed;   /* [2][param_total] / [2][packed_total]              */
  float* d_grads; int64_t grad_stride;                         /* the gradient bucket, [2][grad_stride]             */
  float* d_logits_l; float* d_feat_l; float* d_labels_f;       /* this rank's logits; its block of the exchange buffer */
  float* d_dlogits; float* d_dfeat;                            /* [2][n_l][K], [2][n_l][1024]                        */
  float* d_probs_l; const float* d_probs_g; int32_t probs_shard_rows; int32_t reserved;
  float* d_scalars;             /* ring base [hist_rows][16]: the row comes from the table                          */
  float* d_dfeat_w_partial;     /* [btu_g][1024]                                                                    */
  void* d_workspace; size_t workspace_bytes;                   /* cmlpl_workspace_bytes(shape, 2, n_l, Q)           */
  void* d_loss_workspace; size_t loss_workspace_bytes;         /* cmlpl_loss_workspace_bytes                        */
  uint64_t seed;
  cmlpl_dyn* d_dyn_table; int32_t* d_dyn_cursor;
} cmlpl_dist_io;
enum { CMLPL_STAGE_SPECTRAL = 0, CMLPL_STAGE_SPATIAL = 1, CMLPL_STAGE_PHASE1 = 2, CMLPL_STAGE_PHASE2 = 3,
       CMLPL_STAGE_BACKWARD_DATA = 4, CMLPL_STAGE_BACKWARD_WEIGHTS = 5, CMLPL_STAGE_UPDATE = 6 };
int cmlpl_dist_stage_graph_create(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_dist_io* io, int stage,
                                  void* stream, void** graph_out);

/* The SHARDED step as ONE call: the seven stages above, eagerly, with the four collectives issued between them from
 * inside (train.py:150-278 for a rank's shard; the staging of cmlpl_amd/distributed.py's drive_step: embedding
 * all-gather asynchronously behind the spectral stage and waited for in front of phase 1, probability all-gather,
 * reduce-scatter of d_dfeat_w_partial into net 1's unlabelled rows of d_dfeat asynchronously behind phase 2 and waited
 * for in front of the backward-weights stage, all-reduce of the gradient bucket).  A rank's step driven stage by stage
 * from Python is bound by the host (profiles/r06_dist_trace_b2_64.txt); this entry point is ~25 enqueues back to back.
 *   io    : as for the stage graphs, with d_dyn_table = d_dyn_cursor = NULL (scalars by value: `args`); batch by rows
 *           or by index, noise8 / banks.ptr[] as in the eager stage calls; d_labels_f must follow d_feat_l directly (the
 *           rank's block of the exchange buffer [2*n_l*1024 feat | bt_l labels] is sent as one piece), gathered.
 *           d_logits_local == d_logits_l, probs_shard_rows == btu_local; d_scalars = ring base, row args->scalars_row.
 *   coll  : the rank's communicator as three C functions over float32 buffers (sum), the side stream the two
 *           asynchronous exchanges run on and four hipEvent_t created with hipEventDisableTiming (side_stream NULL:
 *           all four collectives in order on `stream`, no events).  Each function
 *           enqueues on `stream` and returns 0 or an error code.  cmlpl_rccl_bind fills one from an initialised
 *           ncclComm_t and the path of the librccl.so the process uses (dlopen: this library does not link RCCL; it
 *           also creates the side stream and the events); _unbind frees what _bind made (not the ncclComm_t).
 *           NULL = one rank whose exchange buffers alias its own blocks (d_recv_feat == d_feat_l, d_probs_g ==
 *           d_probs_l, d_dfeat_w_partial == d_dfeat + (n_l + bt_l) * 1024): no collective is issued. */
typedef struct cmlpl_collectives {
  void* ctx;
  int (*all_gather)(void* ctx, const float* send, float* recv, size_t send_count, void* stream);
  int (*reduce_scatter)(void* ctx, const float* send, float* recv, size_t recv_count, void* stream);
  int (*all_reduce)(void* ctx, float* buf, size_t count, void* stream);
  void* side_stream;            /* hipStream_t */
  void* events[4];              /* hipEvent_t  */
} cmlpl_collectives;
typedef struct cmlpl_dist_step_args {
  uint64_t step;                /* counter of the in-kernel random streams                                          */
  int64_t adam_t;               /* 1-based                                                                          */
  const float* d_dropmask;      /* NULL = Philox mask; else this rank's [2][n_l][cls_in]                            */
  int32_t smooth; float adap_mask;
  int32_t apply_update;         /* 0 = stop after the gradient all-reduce                                           */
  int32_t scalars_row;
} cmlpl_dist_step_args;
int cmlpl_rccl_bind(const char* librccl_path, void* nccl_comm, cmlpl_collectives* out);
int cmlpl_rccl_unbind(cmlpl_collectives* coll);
int cmlpl_dist_step(const cmlpl_shape* shape, const cmlpl_hparams* hp, const cmlpl_dist_io* io,
                    const cmlpl_dist_step_args* args, const cmlpl_collectives* coll, void* stream);

/* Caller-side row N3: w x w patch windows gathered on device from the z-scored / PCA'd
 * scene cube instead of materialising XP.npy (tools/hyper_tools.py:35-55 MirrowCut, :226-243
 * ExtractPatches; for PaviaU that tensor is 19.9 GB).  d_cube [rows][cols][C] f32, d_pixel_idx int64 [n]
 * row-major pixel indices, d_out [n][C][w][w].  Exact (pure gather).  Even w reproduces the reference;
 * odd w (which the reference cannot run) is the centred generalisation. */
int cmlpl_extract_patches(const float* d_cube, int rows, int cols, int C, int w,
                          const int64_t* d_pixel_idx, int n, float* d_out, void* stream);

/* Caller-side rows N1 x N3 joined: whole-image inference straight from the scene cube
 * (tools/hyper_tools.py:416-437 test_whole over the patches of :226-243 ExtractPatches; train.py:291-294).  One network,
 * eval mode (no dropout): pixels pixel0 .. pixel0 + n - 1 (row-major) of d_cube [rows][cols][C] f32 (band-last, the
 * z-scored / PCA'd scene) with window shape->H == shape->W, their spectra d_spectra [rows * cols][bands] (row pixel0 + i
 * is read for pixel i).  The fused forward gathers each window through the mirror index while it stages its slab: no
 * patch tensor exists in HBM.  d_labels [n] int64 = argmax of the logits (first maximum, NaN first: torch.max);
 * d_logits [n][K] optional (null: not written).  d_params / d_packed: ONE network's flat parameters and packed weights
 * (cmlpl_layout / cmlpl_pack_weights with nets = 1).  Workspace: cmlpl_infer_workspace_bytes(shape, n) -- which is 0 for
 * a shape this entry point does not take.
 * Returns CMLPL_E_SHAPE for windows the per-sample forward does not take (more than 256 window pixels, final pooled
 * maps of more than 12 pixels: extract patches and use cmlpl_basenet2_fwd there). */
size_t cmlpl_infer_workspace_bytes(const cmlpl_shape* shape, int n);
int cmlpl_infer_cube(const cmlpl_shape* shape, const float* d_params, const float* d_packed, const float* d_cube,
                     int rows, int cols, const float* d_spectra, int64_t pixel0, int n, int64_t* d_labels,
                     float* d_logits, void* d_workspace, size_t workspace_bytes, void* stream);

/* Caller-side row N4: tools.models.ContrastiveLoss (tools/models.py:14-39) -- NT-Xent over the
 * pairwise cosine similarity of the 2B normalised embeddings, forward + analytic backward.
 * d_emb_i, d_emb_j [B][D]; d_loss [1]; d_grad_i, d_grad_j [B][D] = dLoss/d emb. */
size_t cmlpl_ntxent_workspace_bytes(int B, int D);
int cmlpl_ntxent_fwd_bwd(const float* d_emb_i, const float* d_emb_j, int B, int D, float temperature,
                         float* d_loss, float* d_grad_i, float* d_grad_j,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * loss_helper.py of the reference.  The reference keeps one FIFO per class as a Python list
 * of tensors; here a class's bank is a physical ring d_bank[c][capacity_stride][D] (capacity[c] slots used) with
 * d_state[c] = {rows, head}:
 * LOGICAL row i (what loss_helper.py's queue[0][i] holds) is ring slot (head + i) % capacity.
 * ---------------------------------------------------------------------------------------------- */

/* compute_unsupervised_loss (loss_helper.py:242-261): rows whose teacher entropy reaches the `percent`-th percentile
 * (numpy 'linear') of the valid rows are set to 255 IN d_target; d_loss[0] = (B / kept) * mean-over-kept CE;
 * d_dpredict [B][K] = d loss / d predict. */
size_t cmlpl_unsup_workspace_bytes(int B);
int cmlpl_unsup_loss(const float* d_predict, int64_t* d_target, const float* d_pred_teacher, int B, int K,
                     double percent, float* d_loss, float* d_dpredict, void* d_workspace, size_t workspace_bytes,
                     void* stream);

/* Per-class selection of compute_contra_memobank_loss (loss_helper.py:67-123).  d_prob / d_label [N][K] are the
 * labelled rows followed by the unlabelled ones (n_labeled first), masks [N].  d_lists [K][3][N] receives, in row
 * order: 0 low_valid rows, 1 low-entropy anchor pool, 2 negative-key rows; d_counts [K][3] their lengths. */
int cmlpl_memobank_select(const float* d_prob, const float* d_label, const float* d_low_mask, const float* d_high_mask,
                          int N, int n_labeled, int K, int32_t* d_lists, int32_t* d_counts, void* stream);
/* class prototypes (loss_helper.py:102-106): d_proto[c] = mean of d_rep_teacher over list 0 (NaN when empty) */
int cmlpl_memobank_proto(const float* d_rep_teacher, int N, int D, int K, const int32_t* d_lists,
                         const int32_t* d_counts, float* d_proto, void* stream);
/* dequeue_and_enqueue (loss_helper.py:19-36, call site :126-133) for every class: list 2 of d_rep_teacher is appended,
 * the last d_capacity[c] rows survive (queue_size may differ per class; class c's ring starts at
 * d_bank + c * capacity_stride * D) */
int cmlpl_memobank_enqueue(const float* d_rep_teacher, int N, int D, int K, const int32_t* d_lists,
                           const int32_t* d_counts, float* d_bank, int32_t* d_state, const int32_t* d_capacity,
                           int capacity_stride, void* stream);
/* dequeue_and_enqueue for one class with the keys given directly */
int cmlpl_memobank_push(const float* d_keys, int m, int D, float* d_bank_c, int32_t* d_state_c, int capacity,
                        void* stream);
/* one loop position of loss_helper.py:158-215: `queries` anchors rep[d_pool[d_anchor_draw[q]]], key 0 =
 * d_pos + q * pos_qstride, keys 1.. = logical bank rows d_neg_draw[q][j]; cosine similarity / temperature, CE with
 * target 0.  d_lossq[q] = scale * CE_q / queries; d_ganchor [queries][D]; if d_drep is given the anchor gradients
 * are added into it (rows drawn repeatedly are summed in query order: deterministic).  Indices must be in range. */
int cmlpl_memobank_infonce(const float* d_rep, int N, int D, const int32_t* d_pool, int pool_rows,
                           const int64_t* d_anchor_draw, const float* d_pos, int64_t pos_qstride,
                           const float* d_bank_c, int capacity, int bank_rows, int head, const int64_t* d_neg_draw,
                           int queries, int negatives, float temperature, float scale, float* d_lossq,
                           float* d_ganchor, float* d_drep, void* stream);
/* compute_contra_memobank_loss (loss_helper.py:39-219) in ONE pass -- three launches, no host read-back: selection +
 * prototypes + enqueue per class; the InfoNCE of every (query, loop position) with the valid-class bookkeeping done
 * on device from the counts; anchor-gradient scatter + loss sum.  d_* outputs:
 *   d_lists [K][3][N] i32, d_counts [K][3] i32, d_proto [K][D]: as cmlpl_memobank_select / _proto
 *   d_keys_log [K][2] i32 (optional): (new keys, rows after) per class, for the caller's pointer bookkeeping
 *   d_lossq [K][queries], d_ganchor [K][queries][D], d_arow [K][queries] i32: per (position, query) scratch
 *   d_drep [N][D]: d loss / d rep (written in full);  d_total [1]: the loss
 * Draws: d_anchor_draw [K][queries] / d_neg_draw [K][queries*negatives] int64 indexed by LOOP POSITION replace the two
 * torch.randint calls (:164,:179) -- entries must be in range for the positions that exist; NULL = drawn in-kernel
 * (Philox keyed by seed / call).  d_momentum [K][queries][D] + d_momentum_on (device int: "not all zero", :196) +
 * ema select the prototype blend of :194-203; then d_prototype [K][queries][D] receives `prototype`. */
typedef struct cmlpl_memobank_call {
  const float* d_rep; const float* d_rep_teacher;
  const float* d_prob_l; const float* d_prob_u; const float* d_label_l; const float* d_label_u;
  const float* d_low_mask; const float* d_high_mask;
  int32_t N, n_labeled, K, D, queries, negatives;
  float* d_bank; int32_t* d_state; const int32_t* d_capacity; int32_t capacity_stride;
  const int64_t* d_anchor_draw; const int64_t* d_neg_draw; uint64_t seed, call;
  const float* d_momentum; const int32_t* d_momentum_on; float ema; float* d_prototype;
  float temperature;
  int32_t* d_lists; int32_t* d_counts; float* d_proto; int32_t* d_keys_log;
  float* d_lossq; float* d_ganchor; int32_t* d_arow; float* d_drep; float* d_total;
} cmlpl_memobank_call;
int cmlpl_memobank_loss(const cmlpl_memobank_call* call, void* stream);

/* fixed-order sum of n floats (the loss of all loop positions) */
int cmlpl_memobank_sum(const float* d_v, int n, float* d_out, void* stream);


/* Optional per-launch timing, measured with hipEvent pairs recorded on the launch stream around the
 * selected kernels (bit i of kernel_mask selects CMLPL_K_i).  cmlpl_timing_end synchronises the
 * recorded events and returns, per kernel id, the summed milliseconds and the number of launches.
 * This is the only process-global state in the library; it is off unless _begin was called. */
enum {
  CMLPL_K_AUGMENT = 0, CMLPL_K_CONV0_FWD, CMLPL_K_CONV1_FWD, CMLPL_K_CONV2_FWD, CMLPL_K_SPE_FWD,
  CMLPL_K_HEAD_FWD, CMLPL_K_LOSS, CMLPL_K_HEAD_BWD, CMLPL_K_CLS_WGRAD, CMLPL_K_SPE_WGRAD,
  CMLPL_K_CONV2_DGRAD, CMLPL_K_CONV2_WGRAD, CMLPL_K_CONV2_WRED, CMLPL_K_CONV1_DGRAD, CMLPL_K_CONV1_WGRAD,
  CMLPL_K_CONV1_WRED, CMLPL_K_CONV0_WGRAD, CMLPL_K_ADAM, CMLPL_K_PACK, CMLPL_K_LOSS2, CMLPL_K_LOSS_FIN,
  CMLPL_K_LOSS_DFEAT, CMLPL_K_COUNT
};
int cmlpl_timing_begin(uint32_t kernel_mask, int max_launches);
int cmlpl_timing_end(double* ms_sum /*[CMLPL_K_COUNT]*/, int64_t* launches /*[CMLPL_K_COUNT]*/);

/* Inspection aid (tests, debugging): byte offset and size inside d_workspace of a saved
 * activation of cmlpl_basenet2_fwd/_bwd for (nets, n).  Names: "a0" conv0 output
 * [nets][n][H*W][64] f32; "p1"/"p2" pooled stage outputs [nets][n][P][64] f32; "m1"/"m2" ReLU
 * masks [nets][n][P][64] u8 (bit (h&1)*2+(w&1) = relu(z) > 0 at that pixel of the 2x2 pool
 * window); "y" spectral ReLU output [nets][n][1024] f32; "catd","dropgen" [nets][n][cls_in];
 * "dy","dp2","dp1","da0" backward intermediates; with nets = 2 also "xn": the augmented patch rows [2][n][C*H*W] a
 * step's forward keeps for its backward (cmlpl_forward / cmlpl_train_step).  Returns CMLPL_E_ARG for an unknown name. */
int cmlpl_debug_region(const cmlpl_shape* shape, int nets, int n, const char* name,
                       size_t* byte_offset, size_t* bytes);

/* Test aid: read the CMLPL_* planner switches from the environment again (they are otherwise read once per
 * process).  Lets ONE test process walk several kernel variants; call it between library calls only -- objects
 * created under the old settings (captured graphs, engines with cached plans) must not be used afterwards.
 * The reference has no counterpart (it has no native code). */
int cmlpl_debug_reload_switches(void);

/* Measurement aid (bench.py's roofline accounting): which products of a step on `nets` networks x n rows run as TWO
 * fp16 pieces (three MFMAs per product) under the current switches -- bit 0: conv1's forward tap loop, bit 1: conv1's
 * data-gradient tap loop, bit 2: both weight gradients, bit 3: conv2's forward and data-gradient tap loops (general
 * path only).  What the planners decide; a sample or network outside the scheme's ranges still takes the three-piece loop
 * at run time.  Negative: CMLPL_E_*. */
int cmlpl_debug_two_piece(const cmlpl_shape* shape, int nets, int n);

#ifdef __cplusplus
}
#endif
#endif /* CMLPL_H */
