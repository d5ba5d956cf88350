/* libdeepof_hip -- C ABI of the MI355X-native DeepOF unsupervised-embedding hot path.
 *
 * Drop-in boundary for the reference trainer (mlfpm/deepof v0.9.0, deepof/clustering).  The
 * reference has no native code: the Python functions below are what a maintainer would rebind
 * (through ctypes).  Every entry point
 *   - takes plain device pointers / sizes / a hipStream_t (passed as void*), no torch types;
 *   - never allocates device memory and never synchronises: the caller owns all buffers,
 *     including the workspace whose size dof_vade_workspace_bytes() reports;
 *   - returns 0 (DOF_OK) or a negative error code; dof_last_error_string() explains it.
 *
 * Reference interface each entry replaces (paths relative to /root/reference):
 *   dof_window_gather[_range]   deepof/utils.py:3354-3377 rolling_window +
 *                               deepof/clustering/dataset.py:16-26 reorder_and_reshape, :183-290 _build_hdf5,
 *                               :561-670 _H5BatchIterableDataset.__iter__ (batch fetch)
 *   dof_vade_forward            deepof/clustering/models_new.py:1841-1891 VaDEPT.forward (eval / train mode)
 *   dof_vade_loss_grads         deepof/clustering/training.py:231-309 step_vade + losses.py:567-797
 *                               VadeLoss.forward + loss.backward() (training.py:163)
 *   dof_vqvae_forward/_loss_grads  models_new.py:1575-1635 VQVAEPT.forward, training.py:312-389 step_vqvae_distill
 *   dof_contrastive_views       training.py:2128-2403 _make_augmented_view (+ the four augmentations),
 *                               model_utils_new.py:332-363 recompute_edges, :751-763 slice_time_per_sample
 *   dof_contrastive_encode      models_new.py:2069-2075 ContrastivePT.forward (recurrent encoder)
 *   dof_contrastive_loss        training.py:482-589 step_contrastive_distill (normalise + loss + logs),
 *                               losses.py:35-249 select_contrastive_loss_pt
 *   dof_contrastive_backward    loss.backward() through one view's encoder pass (training.py:163)
 *   dof_turtle_fit_step/_predict  teacher_model.py:43-350 TurtleTeacher (heads inner fit, task encoder, fit, predict)
 *   dof_optimizer_step          training.py:164-166 clip_grad_value_ + optimizer.step(), losses.py:805-833
 *   dof_comm_*, dof_flat_allreduce   training.py:1087-1096, 1321-1330, 1567-1576 DistributedDataParallel wrapping
 *                               (the gradient all-reduce + the initial parameter broadcast; torch.distributed
 *                               init_process_group("nccl") / destroy_process_group for the communicator)
 *   dof_preprocess_tables, dof_preprocess_video_stats, dof_preprocess_fit_global
 *                               deepof/data.py:3773-3916 TableDict.preprocess (scale="standard") up to window
 *                               extraction: utils.py:2425-2566 scale_table, :2665-2792 _pp_pass1_collect_samples,
 *                               :2795-2863 _pp_fit_global_scaler, :2866-2921 _pp_apply_global,
 *                               :2924-3027 _pp_pass2_scale_and_save, :2577-2583 _pp_sanitize_numeric; output in the
 *                               frame-table layout of data.py:2797-2880 get_graph_dataset (node / edge / angle columns)
 */
#ifndef DEEPOF_HIP_H
#define DEEPOF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DOF_ABI_VERSION 17

/* ---- error reporting ----------------------------------------------------------------------
 * Every int entry point returns 0 or a negative code, with the text in dof_last_error_string():
 * -1 DOF_ERR_ARG (bad argument), -2 DOF_ERR_UNSUPPORTED (shape / feature this build has no kernel for; never a host
 * fallback), -3 DOF_ERR_LAUNCH (HIP / RCCL runtime error), -4 DOF_ERR_STATE (call order, e.g. a step before parameters). */
const char* dof_last_error_string(void);
int dof_abi_version(void);

/* ---- window tensor build (SURVEY 8a R0+R1) -------------------------------------------------
 * node_table (rows, 3N) fp32 column blocks [x_1..x_N | y_1..y_N | s_1..s_N]; edge_table (rows, E).
 * Window b covers table rows row_start[b] .. row_start[b]+window-1.  Outputs in the reference's
 * batch layout: x_out (n_windows, window, N, 3), a_out (n_windows, window, E, 1).  All pointers
 * are device pointers.  The _range variant takes row_start[b] = first_row + b*row_step. */
int dof_window_gather(const float* node_table, const float* edge_table, const int64_t* row_start,
                      int64_t n_windows, int32_t window, int32_t n_nodes, int32_t n_edges, float* x_out,
                      float* a_out, void* stream);
int dof_window_gather_range(const float* node_table, const float* edge_table, int64_t first_row, int64_t row_step,
                            int64_t n_windows, int32_t window, int32_t n_nodes, int32_t n_edges, float* x_out,
                            float* a_out, void* stream);
/* bf16 storage of the window batches (BASELINE.json configs[1] "bf16"; SURVEY 8(d): W*(3N+E)*2 bytes written per
 * window): the same gather writing round-to-nearest-even bf16 (+-Inf kept, a NaN stays a NaN: sign and upper payload
 * kept, quiet bit set), eight elements per 16-byte store.  row_start != NULL:
 * the listed start rows (first_row / row_step ignored), NULL: first_row + k * row_step.  W*3N and W*E must be even.
 * dof_widen_bf16 turns a stored batch back into the fp32 tensors the step kernels read (exact). */
int dof_window_gather_bf16(const float* node_table, const float* edge_table, const int64_t* row_start, int64_t first_row,
                           int64_t row_step, int64_t n_windows, int32_t window, int32_t n_nodes, int32_t n_edges,
                           uint16_t* x_out, uint16_t* a_out, void* stream);
int dof_widen_bf16(const uint16_t* in, float* out, int64_t n, void* stream);

/* ---- VaDE (recurrent encoder/decoder) ------------------------------------------------------ */
typedef struct DofVadeDims {
  int32_t batch;      /* windows per step (per rank) */
  int32_t window;     /* T */
  int32_t n_nodes;    /* N, 3 features per node */
  int32_t n_edges;    /* E, 1 feature per edge */
  int32_t latent;     /* L (internal GRU width = min(64, L) in the reference; this build: 4, 5, 6, 8, 10, 12, 16; 7, 9, 14, 20, 24, 32 with the recurrent encoder) */
  int32_t n_clusters; /* K */
  int32_t mc_samples; /* S of the Monte-Carlo KL (reference: 32) */
} DofVadeDims;

typedef struct DofVadePlan DofVadePlan;

/* Host-side plan: parameter layout, workspace carve-up, graph sparsity.  laplacian (N,N),
 * edge_laplacian (E,E), incidence (N,E) are HOST fp32 arrays (the encoder's registered buffers). */
int dof_vade_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                         const float* incidence, DofVadePlan** out);
void dof_vade_plan_destroy(DofVadePlan* plan);

/* Flat fp32 parameter buffer, tensors in the reference's state_dict order and shapes. */
int32_t dof_vade_param_count(const DofVadePlan* plan);
const char* dof_vade_param_name(const DofVadePlan* plan, int32_t i);
int64_t dof_vade_param_offset(const DofVadePlan* plan, int32_t i);
int64_t dof_vade_param_numel(const DofVadePlan* plan, int32_t i);
int64_t dof_vade_param_total(const DofVadePlan* plan);
/* Tensor shape of parameter i into dims4 (returns the rank; 0 = derive it from the name as for the
 * recurrent families, whose shapes follow from (L, N, E, K)). */
int32_t dof_vade_param_shape(const DofVadePlan* plan, int32_t i, int64_t* dims4);

int64_t dof_vade_workspace_bytes(const DofVadePlan* plan);
/* Zero the workspace and upload the plan's tables into it (enqueued on stream).  Call once per
 * workspace before the first forward / step, outside any graph capture. */
int dof_vade_bind(DofVadePlan* plan, void* workspace, void* stream);

/* Diagnostics (ABI v17): where a named intermediate tensor of a TCN plan sits in the caller-owned workspace -- float
 * offset into the bound workspace, and the padded sequence count Sp of its [time][Sp][32] layout.  The parity tests use it
 * to look at the device's own BatchNorm+ReLU pre-activations where a reference fixture names a ReLU-branch candidate
 * (tests/parity_common.py::confirm_flips_on_device).  Names: "<n|e>.<y1|y2|out|bnp1|bnp2>.<block 0..7>" for the node / edge
 * stream's encoder blocks (bnp = the BatchNorm record mean | rstd | scale | shift, 4 x 32 floats), "<n|e>.skip".
 * Returns DOF_ERR_ARG for another name or a plan without a TCN encoder.  The layout is not part of the contract. */
int dof_vade_ws_tensor(const DofVadePlan* plan, const char* name, int64_t* offset_floats, int64_t* padded_sequences);

/* TCN family (models_new.py:376-819: TCNEncoderPT + TCNDecoderPT): same entry points as the recurrent plans,
 * parameters in VaDEPT / VQVAEPT(encoder_type="TCN").state_dict() order with the BatchNorm running_mean /
 * running_var as entries of the flat buffer (skipped by the optimiser).  dof_vade_loss_grads /
 * dof_vqvae_loss_grads and a dof_vade_forward with eps != NULL run the BatchNorms in train mode and update
 * those entries in place (the `const float* params` of these calls is written to for TCN plans). */
int dof_vade_tcn_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                             const float* incidence, DofVadePlan** out);
int dof_vqvae_tcn_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                              const float* incidence, DofVadePlan** out);

/* Transformer family (models_new.py:832-1327: TFMEncoderPT = per-node / per-edge TransformerCorePT (2 post-norm
 * layers, 4 heads, key_dim = min(64, 3N) rounded down to a multiple of 4, ffn 128, last time step) -> CensNet ->
 * BatchNorm MLP head -> train-time batch standardisation; TFMDecoderPT = latent-expand MLP -> repeat + positional
 * encoding -> 2 causal pre-norm layers (8 heads, width 4L, GELU ffn 128) -> Linear(4L -> 3N) -> loc projection):
 * same entry points as the other plans, parameters in VaDEPT / VQVAEPT / ContrastivePT(encoder_type="transformer")
 * .state_dict() order.  The BatchNorm running buffers of encoder.head are entries of the flat buffer as for the TCN
 * family; dof_vade_set_batchnorm_training(0) also switches dropout and the batch standardisation off (module.eval()).
 * This build: key_dim any multiple of 4 up to 64; window <= 64 with the attention working set (one sequence, forward and
 * backward) inside the kernel's LDS and 512-thread budget -- checked here, DOF_ERR_UNSUPPORTED otherwise. */
int dof_vade_tfm_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                             const float* incidence, DofVadePlan** out);
int dof_vqvae_tfm_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                              const float* incidence, DofVadePlan** out);
int dof_contrastive_tfm_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                                    const float* incidence, DofVadePlan** out);
/* Dropout of a transformer plan (nn.Dropout / the attention-weight dropout of scaled_dot_product_attention,
 * models_new.py:884, 898, 906, 937, 1290-1294, 1318): the sites in the order the reference's forward draws them,
 * each a tensor of `numel` elements in the reference's shape ((sequences, T, width) or (sequences, heads, T, T)).
 * Default source: keep iff hash(seed, site, device step counter, element index) >= p * 2^32, evaluated inside the
 * kernels in forward and backward (no mask is stored); the counter advances on every train-mode encoder forward, so a
 * replayed hipGraph draws fresh masks.  inject_masks != NULL (device, one byte per element, sites concatenated at
 * dof_tfm_dropout_site_offset): those keep-masks are used instead (parity tests replay recorded reference draws). */
int32_t dof_tfm_dropout_site_count(const DofVadePlan* plan);
const char* dof_tfm_dropout_site_name(const DofVadePlan* plan, int32_t i);
int64_t dof_tfm_dropout_site_offset(const DofVadePlan* plan, int32_t i);
int64_t dof_tfm_dropout_site_numel(const DofVadePlan* plan, int32_t i);
float dof_tfm_dropout_site_p(const DofVadePlan* plan, int32_t i);
int dof_tfm_set_dropout(DofVadePlan* plan, const uint8_t* inject_masks, uint32_t seed);
/* The device step counter the mask hash reads.  Default (NULL): a slot of the plan's own workspace, zeroed by
 * dof_vade_bind.  A model that owns several plans (batch sizes, the ragged last batch, the two contrastive views)
 * points all of them at ONE caller-owned device counter, so every train-mode encoder forward of the model consumes a
 * distinct counter value -- the reference draws independent masks for every forward pass (F.dropout on the global
 * generator).  The counter is advanced by the train-mode encoder forward of whichever plan runs. */
int dof_tfm_set_dropout_counter(DofVadePlan* plan, uint32_t* device_counter);

/* module.train() / module.eval() for the BatchNorm layers of a TCN plan (default: training).  With training = 0
 * the loss/grad and train-flagged encode entries normalise with the running buffers and leave them untouched --
 * the reference's validation passes (validate_one_epoch_indexed, training.py:190-229, runs under model.eval()). */
int dof_vade_set_batchnorm_training(DofVadePlan* plan, int32_t training);

/* Exclude (trainable = 0) / include parameter i in dof_optimizer_step -- the reference's "this tensor is
 * not in the optimiser" cases (quirk Q11: the lazily built CensNet tensors of the TCN encoders). */
int dof_vade_set_trainable(DofVadePlan* plan, int32_t i, int32_t trainable, void* stream);

/* hyper[]: device fp32 array of DOF_H_COUNT scalars read by the kernels (graph-replay safe). */
enum {
  DOF_H_KLW = 0, DOF_H_LAMBDA_DISTILL, DOF_H_KM_LATENT, DOF_H_KM_LOSS, DOF_H_REPEL_W, DOF_H_REPEL_LS,
  DOF_H_NONEMPTY_W, DOF_H_NONEMPTY_FLOOR, DOF_H_NONEMPTY_P, DOF_H_L1_ACT, DOF_H_DISTILL_T, DOF_H_CONF_W,
  DOF_H_CONF_THR, DOF_H_HAS_TEACHER, DOF_H_LOGVAR_LO, DOF_H_LOGVAR_HI,
  DOF_H_CLIP = 16, DOF_H_WD = 17,
  DOF_H_LR0 = 18,      /* 4 learning rates, one per optimiser segment */
  DOF_H_BC0 = 22,      /* 4 x (1-beta1^t, 1-beta2^t) */
  DOF_H_ACTIVE0 = 30,  /* 4 x "segment has gradients" (0 = frozen / grad None) */
  DOF_H_VQ_BETA = 34,  /* VQ-VAE commitment weight beta */
  DOF_H_TF_W = 36, DOF_H_CAT_W = 37, DOF_H_TEMPORAL_W = 38, DOF_H_SCATTER_W = 39, DOF_H_SCATTER_BETA = 40,
  DOF_H_COUNT = 42
};
/* optimiser segments of the VaDE parameter buffer */
enum { DOF_SEG_ENCODER = 0, DOF_SEG_DECODER = 1, DOF_SEG_GMM = 2, DOF_SEG_HEADS = 3, DOF_SEG_COUNT = 4 };
/* logs[]: device fp32 array of DOF_LOG_COUNT loss terms (keys of step_vade's logs dict). */
enum {
  DOF_LOG_TOTAL = 0, DOF_LOG_RECON, DOF_LOG_KL, DOF_LOG_CAT, DOF_LOG_KMEANS, DOF_LOG_ACTIVITY, DOF_LOG_PRIOR,
  DOF_LOG_DISTILL, DOF_LOG_TFCLUST, DOF_LOG_NONEMPTY, DOF_LOG_TEMPORAL, DOF_LOG_SCATTER, DOF_LOG_REPEL,
  DOF_LOG_KLW,
  DOF_LOG_ENC_REC = 14, DOF_LOG_VQ = 15, DOF_LOG_POPULATED = 16, /* VQ-VAE: enc_rec_loss, vq_loss, populated codes */
  DOF_LOG_POS_SIM = 17, DOF_LOG_NEG_SIM = 18,                    /* contrastive: pos_similarity, neg_similarity */
  DOF_LOG_COUNT = 20
};

/* Forward.  x (B,T,N,3), a (B,T,E,1) device fp32.  prior (K) device.  eps (B,L) device or NULL:
 * NULL = eval mode (z = z_mean), non-NULL = train mode (z = mean + exp(softplus/2)*eps).
 * Outputs (any may be NULL): z_out (B,L) latent, q_out (B,K) soft cluster assignments,
 * zmean_out (B,L), zlogvar_out (B,L) (softplus output), loc_out (B,T,3N) reconstruction mean,
 * enc_out (B,L) encoder output. */
int dof_vade_forward(DofVadePlan* plan, const float* params, const float* prior, const float* x, const float* a,
                     const float* eps, float* z_out, float* q_out, float* zmean_out, float* zlogvar_out,
                     float* loc_out, float* enc_out, void* stream);

/* Forward + VadeLoss + backward: fills grads (same layout as params) and logs[DOF_LOG_COUNT].
 * eps (B,L); eps_mc (S,B,L) (main phase only); tau (B,K) teacher targets for this batch or NULL;
 * teacher (2K): class weights then teacher marginal, or NULL; pretrain != 0 selects the
 * pre-training objective (KL vs N(0,I)), 0 the main objective (MC-KL vs the GMM). */
int dof_vade_loss_grads(DofVadePlan* plan, const float* params, const float* prior, const float* x,
                        const float* a, const float* eps, const float* eps_mc, const float* tau,
                        const float* teacher, const float* hyper, int32_t pretrain, float* grads, float* logs,
                        void* stream);

/* ---- VQ-VAE (recurrent encoder/decoder + codebook of dims.n_clusters codes) ------------------
 * Same plan type / parameter-table / workspace / bind functions as VaDE (dof_vade_param_*,
 * dof_vade_workspace_bytes, dof_vade_bind, dof_vade_plan_destroy); parameters = encoder.*,
 * decoder.*, vq_layer.codebook (L,K) in the reference VQVAEPT state_dict order.
 * Replaces models_new.py:1575-1635 VQVAEPT.forward, :1330-1423 VectorQuantizerPT and
 * training.py:312-389 step_vqvae_distill (+ backward).  Outputs of dof_vqvae_forward (any may be
 * NULL): ze (B,L) encoder output, quant (B,L), soft (B,K) soft counts, idx (B) int32 code indices,
 * loc_q / loc_e (B,T,3N) reconstruction means from the quantised / raw latents. */
int dof_vqvae_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                          const float* incidence, DofVadePlan** out);
int dof_vqvae_forward(DofVadePlan* plan, const float* params, const float* x, const float* a, float* ze_out,
                      float* quant_out, float* soft_out, int32_t* idx_out, float* loc_q_out, float* loc_e_out,
                      void* stream);
/* tau (B,K): teacher targets of this batch or NULL.  With tau the generic distillation head (the last two entries of
 * the parameter buffer, distill_head.fc.{weight (K,L), bias (K)} = the reference's DiscriminativeHead, which shares
 * the model's optimiser) adds hyper[DOF_H_LAMBDA_DISTILL] * soft-CE(head(z_e), sharpened tau) (training.py:344-372). */
int dof_vqvae_loss_grads(DofVadePlan* plan, const float* params, const float* x, const float* a, const float* tau,
                         const float* hyper, float* grads, float* logs, void* stream);

/* ---- Contrastive (recurrent encoder on half windows, two views per window) -----------------------
 * Plan: dims.window = the HALF window the encoder sees (full window // 2); dims.n_clusters /
 * mc_samples are ignored.  Parameters = encoder.* only (ContrastivePT state_dict order); same
 * dof_vade_param_* / workspace / bind / destroy / dof_optimizer_step functions as the other plans.
 * One plan + workspace per view (activations of both views are alive until the backward passes);
 * the two plans share the caller's parameter, gradient and Adam buffers. */
int dof_contrastive_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                                const float* incidence, DofVadePlan** out);

/* Same, with the TCN encoder (models_new.py:376-657 TemporalBlockPT / TCN1DPT / TCNEncoderPT): parameters =
 * ContrastivePT(encoder_type="TCN").state_dict() order, BatchNorm running_mean / running_var included as
 * entries of the flat buffer (never touched by the optimiser; updated in place by a train-mode
 * dof_contrastive_encode, which is why that call may write to `params` for this plan kind). */
int dof_contrastive_tcn_plan_create(const DofVadeDims* dims, const float* laplacian, const float* edge_laplacian,
                                    const float* incidence, DofVadePlan** out);

#define DOF_MAX_ROT 8
/* Resolved random choices of one augmented view (the reference draws them inside
 * _make_augmented_view).  Device arrays unless stated; any pointer may be NULL = that step is off. */
typedef struct DofAugment {
  const int32_t* start;       /* (B) first frame of the view inside the full window (base + shift, clamped) */
  int32_t n_rot;              /* rotations applied in order, <= DOF_MAX_ROT (host scalars below) */
  int32_t rot_pivot[DOF_MAX_ROT];   /* pivot node of each rotation */
  uint64_t rot_nodes[DOF_MAX_ROT];  /* bit n set = node n is rotated about the pivot (n_nodes <= 64) */
  const float* theta;         /* (n_rot, B) rotation angles in radians (0 = sample not rotated) */
  const int32_t* interp_t0;   /* (B) first frame of the linearly interpolated segment */
  const int32_t* interp_len;  /* (B) its length; 0 = none */
  const float* noise;         /* (B, N, 3) offsets added to (x, y, speed) of every frame */
} DofAugment;

/* Builds one view of every full window: x_full (B, t_full, N, 3) -> x_out (B, t_full/2, N, 3) and
 * a_out (B, t_full/2, E, 1) = node distances over edge_index (E,2 int32, device).  aug == NULL gives
 * the reference's un-augmented central view (start = (t_full/2)/2). */
int dof_contrastive_views(const float* x_full, const int32_t* edge_index, int32_t batch, int32_t t_full,
                          int32_t n_nodes, int32_t n_edges, const DofAugment* aug, float* x_out, float* a_out,
                          void* stream);

/* Encoder pass of one view: z_out (B, L).  train != 0 keeps the activations for dof_contrastive_backward. */
int dof_contrastive_encode(DofVadePlan* plan, const float* params, const float* x, const float* a, int32_t train,
                           float* z_out, void* stream);

enum { DOF_SIM_COSINE = 0, DOF_SIM_DOT = 1, DOF_SIM_EUCLIDEAN = 2 /* also "edit" */ };
enum { DOF_CLOSS_NCE = 0, DOF_CLOSS_DCL = 1, DOF_CLOSS_HARD_DCL = 2, DOF_CLOSS_FC = 3 /* top-10 % negatives dropped per row */ };
/* Row-normalises z / z_aug (B, L), evaluates the loss over all B x B pairs, writes d loss / d z and
 * d loss / d z_aug (B, L; either may be NULL together = value only) and logs[DOF_LOG_TOTAL |
 * DOF_LOG_POS_SIM | DOF_LOG_NEG_SIM].  Scratch comes from the plan's workspace. */
/* teacher_tau (B,K) or NULL: as for the VQ-VAE, the distillation head acts on the NORMALISED central embeddings
 * (training.py:553-580); it needs params (head weights) and hyper (lambda, sharpening, confidence weighting), its
 * weight gradients are written by the following dof_contrastive_backward(plan, ..., accumulate = 0). */
int dof_contrastive_loss(DofVadePlan* plan, const float* z, const float* z_aug, int32_t similarity, int32_t loss_fn,
                         float temperature, float tau, float beta, const float* params, const float* teacher_tau,
                         const float* hyper, float* dz, float* dz_aug, float* logs, void* stream);

/* Backward of the plan's last dof_contrastive_encode(train) from dz (B, L): encoder gradients into
 * grads (accumulate == 0: overwritten, else added -- the second view accumulates onto the first). */
int dof_contrastive_backward(DofVadePlan* plan, const float* params, const float* dz, float* grads,
                             int32_t accumulate, void* stream);

/* ---- TURTLE teacher (soft cluster targets tau* from linear heads over several views) ------------------
 * Replaces teacher_model.py:43-350 (TurtleHeads.inner_fit, TaskEncoder.forward, TurtleTeacher.fit / predict).
 * Parameter buffer = TurtleTeacher.state_dict() order: heads.heads.v.{weight (K,d_v), bias (K)} for every view,
 * then task_encoder.projs.v.{weight, bias}.  adam_m / adam_v have the same layout (only the task-encoder part
 * is used).  feats: HOST array of n_views DEVICE pointers, view v = (rows, d_v) fp32 row-major. */
#define DOF_TURTLE_MAX_VIEWS 8
typedef struct DofTurtleDims {
  int32_t batch;       /* rows per outer step */
  int32_t n_views;
  int32_t n_clusters;  /* K <= 64 */
  int32_t view_dim[DOF_TURTLE_MAX_VIEWS];
} DofTurtleDims;
typedef struct DofTurtleHyper {
  float gamma, alpha, delta;        /* marginal-entropy weight, sample-entropy weight, dead-cluster barrier */
  float head_temp, task_temp;
  float inner_lr, head_wd, lr_theta, rho;
  int32_t inner_steps, normalize_feats;
} DofTurtleHyper;
int64_t dof_turtle_param_total(const DofTurtleDims* dims);
int64_t dof_turtle_param_offset(const DofTurtleDims* dims, int32_t task_encoder, int32_t view, int32_t bias);
int64_t dof_turtle_workspace_bytes(const DofTurtleDims* dims);
/* One outer step `step` of `outer_steps` on one batch: tau, all inner SGD steps of every head, the loss and the
 * Adam update of the task encoder.  logs[5] (device) = loss, CE, E[H(tau)], H(marginal), dead-cluster penalty. */
int dof_turtle_fit_step(const DofTurtleDims* dims, const DofTurtleHyper* hyper, const float* const* feats,
                        float* params, float* adam_m, float* adam_v, int32_t step, int32_t outer_steps,
                        void* workspace, float* logs, void* stream);
/* tau (n_rows, K) of the task encoder for any number of rows (dims.batch is ignored). */
int dof_turtle_predict(const DofTurtleDims* dims, float task_temp, const float* const* feats, const float* params,
                       int64_t n_rows, float* tau_out, void* stream);

/* clip_grad_value_(hyper[DOF_H_CLIP]) + Adam(betas 0.9/0.999, eps 1e-8, weight decay hyper[DOF_H_WD]).
 * opt_state: device int32[DOF_SEG_COUNT], the Adam step count t of every optimiser segment (torch.optim.Adam's
 * per-parameter `step`); this call advances the counter of every segment whose hyper[DOF_H_ACTIVE0 + seg] != 0 and
 * derives the bias corrections 1 - beta^t from it on the device, so that a captured step replays without any
 * host-written per-step value (zero the array to restart the optimiser; hyper[DOF_H_BC0 ..] is no longer read).
 * grad_scale multiplies every gradient before clipping: 1 / world after an all-reduce SUM over data-parallel
 * ranks (DDP's gradient averaging, training.py:1567-1576), 1 otherwise. */
int dof_optimizer_step(DofVadePlan* plan, float* params, const float* grads, float* adam_m, float* adam_v,
                       const float* hyper, int32_t* opt_state, float grad_scale, void* stream);

/* Per-step schedule values without the host: hyper[item.hyper_index] = item.scale * item.table[min(*item.cursor,
 * item.len - 1)], then *item.cursor += 1 when item.advance.  table (device fp32, item.len entries) is the whole
 * weight curve of a Dynamic_weight_manager (losses.py:290-351: KL weight, distillation lambda) evaluated per
 * iteration; cursor is a device int32 owned by the caller (its get_weight() / step() pair).  Enqueued at the head of
 * a training step (advance = 1) or of a validation step (advance = 0; scale = 0 switches a term off), it makes the
 * step a pure function of device memory: capturable into a hipGraph and free of host/device races on hyper[]. */
#define DOF_SCHED_MAX_ITEMS 4
typedef struct DofSchedItem {
  const float* table;
  int32_t* cursor;
  int32_t len;
  int32_t hyper_index;
  int32_t advance;
  float scale;
} DofSchedItem;
int dof_schedule_apply(float* hyper, const DofSchedItem* items, int32_t n_items, void* stream);

/* Head of a captured training step in ONE launch: the schedule items of dof_schedule_apply plus the step's Gaussian
 * noise.  bufs[i].out[0 .. n) receives independent N(0,1) draws -- the reparameterisation noise of
 * models_new.py:1741 (torch.randn_like) and the Monte-Carlo samples of losses.py:536 (torch.randn(S, B, D)), which
 * the reference takes from torch's generator one ATen launch each.  Generator: Philox-4x32-10 with key = seed and
 * counter (element / 4, buffer index i, call index, 0); the four 32-bit outputs of a call become four normals by
 * Box-Muller on 24-bit uniforms ((x >> 8) + 0.5) * 2^-24.  rng_state: device int32[2] owned by the caller, zeroed once:
 * [0] = number of calls made with it (read by the launch and advanced on the device when it ends, so a replayed
 * hipGraph draws fresh noise), [1] = scratch.  Same stream of draws for the same (seed, rng_state[0]) on every run. */
#define DOF_NOISE_MAX_BUFS 2
typedef struct DofNoiseBuf {
  float* out;
  int64_t n;
} DofNoiseBuf;
int dof_step_begin(float* hyper, const DofSchedItem* items, int32_t n_items, uint64_t seed, int32_t* rng_state,
                   const DofNoiseBuf* bufs, int32_t n_bufs, void* stream);

/* Running sums of the logged loss terms (step_vade's logs dict averaged over an epoch, training.py:167-181): with
 * accum != NULL (device float64[DOF_LOG_COUNT]) every dof_vade_loss_grads on this plan also adds its logs[] to
 * accum[] inside its last loss kernel; NULL switches that off.  The caller zeroes / reads accum. */
int dof_vade_set_log_accumulator(DofVadePlan* plan, double* accum);

/* ---- pose-table preprocessing -----------------------------------------------
 * Raw merged tables of all videos, concatenated: raw (n_frames, n_cols) float64 row-major (NaN = missing), video v =
 * rows video_off[v] .. video_off[v+1]-1.  One call = TableDict.preprocess(scale="standard") without the windows:
 * size factors (nan-median nose--tail-base length per animal and video), size normalisation, log1p of distances,
 * per-video standardisation, global standardisation fitted on the sampled rows (or taken from `scaler`), clipping
 * of |z| > clip to missing, linear interpolation in time within each video (flat at the ends, 0 for a column with
 * no valid row), fp32 cast, column selection into the resident frame tables dof_window_gather reads:
 * node_out (n_frames, n_node_cols) = [x.. | y.. | speed..], edge_out (n_frames, n_edge_cols), angle_out.
 * All arithmetic is float64 like the reference's pandas / sklearn path.  Column metadata (device int32 arrays):
 *   col_kind[n_cols]        DOF_PP_OTHER .. DOF_PP_ANGLE
 *   size_ref[n_animals][4]  columns of (nose x, nose y, tail-base x, tail-base y) of each animal, -1 when absent
 *   chain_off[n_cols+1], chain[4*k]  size-divisor chain of a column: entries (animal a1, animal a2, same, source
 *                           column) with -1 = "not one of the animals" -> default factor; divisor = prod over entries
 *                           of (same ? s[a1] : combine(s[a1], s[a2])); a column with an empty chain is not divided;
 *                           an entry whose source column is filtered out of a video (col_keep) is skipped there
 *   out_cols[n_node_cols + n_edge_cols + n_angle_cols]   source column of every output column
 * sample_mask (n_frames) uint8 or NULL: rows entering the global fit (NULL = all rows).
 * scaler (n_cols, 2) float64 = (mean, scale) of the global scaler per column (a group's columns hold the same pair):
 * written when dims.fit_global, read otherwise.  size_out (n_videos, n_animals + 1) float64 or NULL: the size
 * factors and the default factor.  video_scaler (n_videos, n_cols, 2) float64 or NULL: per-video (mean, scale). */
#define DOF_PP_OTHER 0
#define DOF_PP_COORD 1
#define DOF_PP_SPEED 2
#define DOF_PP_DIST_INNER 3
#define DOF_PP_DIST_INTRA 4
#define DOF_PP_ANGLE 5
#define DOF_PP_MODE_NONE 0
#define DOF_PP_MODE_PER_COLUMN 1
#define DOF_PP_MODE_GROUPWISE 2
#define DOF_PP_SCALE_STANDARD 0 /* sklearn StandardScaler: (x - mean) / std */
#define DOF_PP_SCALE_MINMAX 1   /* sklearn MinMaxScaler: (x - min) / (max - min); `scaler` rows are (min, range) */
#define DOF_PP_SCALE_ROBUST 2   /* sklearn RobustScaler: (x - median) / (q75 - q25); `scaler` rows are (median, IQR) */
#define DOF_PP_STAT_DOUBLES 5   /* one statistics row: (n, mean, M2, min, max) */
#define DOF_PP_ORDER_DOUBLES 7  /* one order-statistics row: n, then the order statistics at ranks (n-1)/2, n/2,
                                 * floor((n-1)/4), that + 1, floor(3(n-1)/4), that + 1 (upper ranks capped at n-1) */
#define DOF_PP_MAX_COLS 512
#define DOF_PP_MAX_ANIMALS 8
typedef struct DofPreprocDims {
  int64_t n_frames;          /* rows of raw */
  int32_t n_videos, n_cols, n_animals;
  int32_t n_node_cols, n_edge_cols, n_angle_cols;
  int32_t speed_mode, dist_mode, coord_mode; /* DOF_PP_MODE_* */
  int32_t log_distances;
  int32_t inter_scale;       /* 0 mean, 1 geometric mean, 2 default factor (scale_table's inter_scale) */
  int32_t fit_global;        /* 1: fit the global scalers; 0: apply the ones passed in `scaler` */
  double clip;               /* interpolate_normalized; 0 disables clipping */
  int32_t scale_kind;        /* DOF_PP_SCALE_* (scale_table / _pp_make_scaler's `scale`, utils.py:2425, :2570) */
  int32_t reserved;
  /* _pp_filter_low_variance (utils.py:2604): device bytes (n_videos, n_cols), 0 = the column is dropped in that video
   * (absent from size references, size divisions and every statistic; written back as zeros, utils.py:3011-3015);
   * NULL = nothing dropped.  The decisions come from dof_preprocess_raw_moments. */
  const uint8_t* col_keep;
  /* scale_kind robust only: per-video (median, IQR) per column, device (n_videos, n_cols, 2) float64, identity (0, 1)
   * where a column is not scaled per video.  Medians and quantiles are not mergeable statistics, so the robust path is
   * three calls: dof_preprocess_order_stats (per video) -> these rows; again with them -> `scaler`; then
   * dof_preprocess_tables(fit_global = 0). */
  const double* video_scaler_in;
} DofPreprocDims;
int64_t dof_preprocess_workspace_bytes(const DofPreprocDims* dims);
int dof_preprocess_tables(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                          const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                          const int32_t* chain, const int32_t* out_cols, const uint8_t* sample_mask, double* scaler,
                          double* size_out, double* video_scaler, float* node_out, float* edge_out, float* angle_out,
                          void* workspace, void* stream);
/* Videos sharded over ranks (one process per GPU): the only quantity that couples the videos is the global scaler.
 * dof_preprocess_video_stats runs the statistics pass on the local videos and writes ystat_out (n_videos, n_cols,
 * DOF_PP_STAT_DOUBLES) float64 = (n, mean, M2, min, max) of the sampled, per-video-scaled values; after an all-gather of these rows (in global
 * video order) dof_preprocess_fit_global fits the scalers exactly as a single call over all videos would (same merge
 * order, bit-identical), and dof_preprocess_tables(fit_global = 0, scaler) finishes the local videos.  The output
 * column counts of dims are ignored by these two. */
int dof_preprocess_video_stats(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                               const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                               const int32_t* chain, const uint8_t* sample_mask, double* ystat_out, void* workspace,
                               void* stream);
int dof_preprocess_fit_global(const DofPreprocDims* dims, int32_t n_videos_total, const int32_t* col_kind,
                              const double* ystat_all, double* scaler, void* stream);
/* Exact order statistics for scale_kind robust (RobustScaler.fit: np.nanmedian, np.nanpercentile(25, 75), utils.py:2570).
 * video_scaler == NULL: per video, of the size-normalised / log1p'd values of every per-video-scaled column (groupwise
 * sections pool the group's columns): out (n_videos, n_cols, DOF_PP_ORDER_DOUBLES).  video_scaler (n_videos, n_cols, 2)
 * given: over the rows of sample_mask (NULL = all) of ALL videos, of the per-video-scaled values (v - median) / IQR:
 * out (n_cols, DOF_PP_ORDER_DOUBLES).  Columns outside the scaled sections get n = 0 and NaNs.  The caller turns the
 * rows into (median, IQR) with numpy's interpolation rule (O(videos x columns) host work). */
int dof_preprocess_order_stats(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                               const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                               const int32_t* chain, const double* video_scaler, const uint8_t* sample_mask, double* out,
                               void* workspace, void* stream);
/* (n, mean, M2, min, max) of the RAW values of every column of every video: moments_out (n_videos, n_cols,
 * DOF_PP_STAT_DOUBLES) float64.  The input of _pp_filter_low_variance (utils.py:2604: pandas var, ddof = 1, NaNs
 * skipped = M2 / (n - 1)); the column decisions themselves are O(videos x columns) host work.  The output column
 * counts of dims are ignored; workspace as for dof_preprocess_tables. */
int dof_preprocess_raw_moments(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                               const int32_t* col_kind, double* moments_out, void* workspace, void* stream);

/* ---- data-parallel exchange: ONE all-reduce of the flat gradient per step ------------------------
 * RCCL over xGMI, one process per GPU.  librccl is opened at the first dof_comm_* call (dlopen: the library does not
 * link it, a process that never trains data-parallel never loads it); without it the calls fail with
 * DOF_ERR_UNSUPPORTED -- there is no host fallback.  Rank 0 calls dof_comm_unique_id and hands the DOF_COMM_ID_BYTES to
 * the other ranks over any channel it has (the reference's launcher environment: MASTER_ADDR / a file / MPI); every rank
 * then calls dof_comm_create on its device.  dof_flat_allreduce sums `n` floats in place over the ranks and
 * dof_comm_broadcast copies root's buffer to all, both enqueued on `stream` without a host wait: with the step's stream
 * the data-parallel step is [gather + loss + gradients] -> all-reduce -> [clip + Adam with grad_scale = 1 / world] in
 * stream order (and capturable into one hipGraph).  The mean of DistributedDataParallel is grad_scale's job. */
#define DOF_COMM_ID_BYTES 128
typedef struct DofComm DofComm;
int dof_comm_unique_id(void* id_out);
int dof_comm_create(const void* id, int32_t rank, int32_t world, DofComm** out);
int dof_comm_destroy(DofComm* comm);
/* ncclCommAbort: frees the communicator WITHOUT waiting for collectives in flight -- the way out of a collective that
 * hung or failed its self-check (deepof_amd.training.dp_self_check).  ABI v16. */
int dof_comm_abort(DofComm* comm);
int dof_flat_allreduce(DofComm* comm, float* buf, int64_t n, void* stream);
int dof_comm_broadcast(DofComm* comm, float* buf, int64_t n, int32_t root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPOF_HIP_H */
