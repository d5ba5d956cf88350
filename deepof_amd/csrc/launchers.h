// Internal host-side launcher declarations shared between the kernel files and the step
// orchestration (vade.hip).  Not part of the public C ABI (that is include/deepof_hip.h).
#pragma once
#include "dof_rt.h"
#include "deepof_hip.h"

struct DofGruW {  // both directions of one torch.nn.GRU layer (weight_ih_l0[_reverse], ...)
  const float *wih0, *whh0, *bih0, *bhh0;
  const float *wih1, *whh1, *bih1, *bhh1;
};

// ---- k_rnn.hip ------------------------------------------------------------------------------
int dof_launch_enc_conv_fwd_pair(int L, const int F[2], const float* const xin[2], const float* const w[2], float* const xs[2],
                                 float* const c[2], int* const len[2], int T, const int G[2], const int64_t S[2],
                                 const int64_t Sp[2], hipStream_t st);   // 1: launched, 0: use the per-stream launcher
int dof_launch_enc_conv_fwd(int L, int F, const float* xin, const float* w, float* xs, float* c, int* len, int T,
                            int G, int64_t S, int64_t Sp, hipStream_t st);
int dof_launch_gru16_fwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], float* const O[2], int T,
                              const int64_t S[2], const int64_t Sp[2], hipStream_t st);   // 1 = launched, 0 = not selected
int dof_launch_gru16_bwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], const float* const O[2],
                              const float* const dO[2], float* const dX[2], float* const wg_partial[2], int T,
                              const int64_t S[2], const int64_t Sp[2], hipStream_t st);
bool dof_gru_lane_per_unit(int L, int kind);   // latent 4 / 5 / 6 (and 10's hidden-10 layers): lane-per-unit GRU kernels, unit-major gate buffers (as latent 8)
int dof_launch_gru_fwd(int L, int kind, const float* X, const int* len, DofGruW W, float* O, float* GS, int T,
                       int64_t S, int64_t Sp, hipStream_t st);
int dof_launch_gru_bwd(int L, int kind, const int* len, DofGruW W, const float* O, float* GS, const float* dO,
                       const float* dHfin, float* dX, int T, int64_t S, int64_t Sp, hipStream_t st, const float* X = nullptr,
                       float* wg_part0 = nullptr, float* wg_part1 = nullptr);
int dof_gru3_wg_blocks(int L, int kind, int64_t S);
int dof_launch_ln_fwd(int L, int mult, const float* X, const float* gamma, const float* beta, float* Y, int T,
                      int64_t S, int64_t Sp, hipStream_t st);
// fused GRU(16,16) backward + MFMA weight-gradient accumulation (latent 8)
int64_t dof_gru16_wg_floats(int64_t S);
int64_t dof_gru8_wg_floats(int64_t S);
int dof_launch_gru8_bwd_fused(const float* X, const int* len, DofGruW W, const float* O, const float* GS,
                              const float* dHfin, float* dX, float* wg_partial, int T, int64_t S, int64_t Sp,
                              hipStream_t st);
bool dof_gru8m_fwd_selected(int64_t S0, int64_t S1, int T);   // (T: the kernels' 32-bit row offsets bound T * Sp)
int dof_launch_gru8m_fwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], float* const O[2],
                              float* const GS[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st);
int dof_launch_gru8_fwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], float* const O[2],
                             float* const GS[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st);
int dof_launch_gru8_bwd_fused_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], const float* const O[2],
                                   const float* const GS[2], const float* const dHfin[2], float* const dX[2],
                                   float* const wg_partial[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st);
int dof_launch_gru8_wg_finalize(const float* wg_partial, int64_t S, float* g, const int64_t* off, int accumulate,
                                hipStream_t st);
int dof_launch_gru16_wg_finalize_pair(const float* const* wg_partial, const int64_t* S, float* g, const int64_t* const* off,
                                      int accumulate, hipStream_t st, int n = 2);
int dof_launch_gru8_wg_finalize_pair(const float* const wg_partial[2], const int64_t S[2], float* g, const int64_t* const off[2],
                                     int accumulate, hipStream_t st, int T);
int dof_launch_gru16_bwd_fused(const float* X, const int* len, DofGruW W, const float* O, const float* GS,
                               const float* dO, float* dX, float* wg_partial, int T, int64_t S, int64_t Sp,
                               hipStream_t st);
int dof_launch_gru16_wg_finalize(const float* wg_partial, int64_t S, float* g, const int64_t* off, int accumulate,
                                 hipStream_t st);
int64_t dof_ln_bwd_blocks(int T, int64_t S);
int dof_launch_ln_bwd(int L, int mult, const float* X, const float* dY1, const float* dY2, const float* gamma,
                      float* dX, float* partial, int T, int64_t S, int64_t Sp, hipStream_t st);
// frame validity [T][Bp] and valid-frame count per window of the decoder, computed in the same launch (third slice of the grid)
struct DofDecValid {
  const float* x;  // (B, T, C3) input windows, or null: no third slice
  int T, C3;
  int64_t B, Bp;
  float* valid;
  int* len;
};
int dof_launch_enc_final_fwd_pair(int L, const float* const O2[2], const int* const len[2], const float* const gamma[2],
                                  const float* const beta[2], float* const HF[2], float* const Y[2], const float* const cw[2],
                                  float* const dots[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st,
                                  const DofDecValid* dec = nullptr);
int dof_launch_zero(float* p, int64_t n, hipStream_t st);
int dof_launch_relu_merge(const float* act, float* d0, const float* d1, int64_t n, hipStream_t st);
// encoder convolution weight gradient from the first GRU layer's two input gradients (k_enc_conv_wgrad): workgroups per
// stream (0 = channel count not covered: keep k_relu_merge + the k_outer job) and the launch for both streams
int dof_enc_conv_wgrad_blocks(int C1, int64_t S);
int dof_launch_enc_conv_wgrad(int C1, const float* const act[2], const float* const dX[2], const float* const xs[2], const int F[2],
                              int T, const int64_t S[2], const int64_t Sp[2], const int64_t part_off[2], float* partials,
                              hipStream_t st);

// ---- k_tcn.hip ---------------------------------------------------------------------------------
int64_t dof_tcn_row_blocks(int T, int64_t S);   // partial rows written by the row-per-thread kernels
int64_t dof_tcn_conv_waves(int T, int64_t Sp);  // partial rows written by the MFMA convolution (one per wave)
int dof_launch_tcn_in_conv(int F, const float* xin, const float* w, const float* bias, float* xs, float* y,
                           float* partial, int T, int G, int64_t S, int64_t Sp, int dil, hipStream_t st, int records = 0);
int dof_tcn_onepass_stats();                         // always 0 since round 5 (the shifted one-pass sums are no longer selectable)
int dof_tcn_conv32_resident(int T, int64_t Sp);               // 1: the 32 -> 32 convolutions run the time-resident kernel (it can fuse pass 2 of a BatchNorm backward)
int64_t dof_tcn_conv32_partials(int T, int64_t Sp);   // partial rows written by the 32 -> 32 convolution
int dof_launch_tcn_conv_bwd_bn(const float* dy, const float* w, const float* y, const float* bnp, float* g_out,
                               float* partial, float* sums, int T, int dil, int64_t S, int64_t Sp, hipStream_t st,
                               const float* bwd_y = nullptr, const float* bwd_bnp = nullptr,
                               const float* bwd_coef = nullptr, int bwd_store = 1,
                               float* wg_partials = nullptr, int64_t wg_part0 = 0, int64_t wg_part1 = 0);
// 1: the fused data-gradient variants of the time-resident convolution also accumulate the convolution's weight gradient
// (k_tcn_conv_b WGRAD; partial tiles [dof_tcn_conv32_partials][64][65] per tap pair, k_tcn_wgrad_b3's layout)
int dof_tcn_wgrad_fused(int T, int64_t Sp);
int dof_launch_tcn_conv(int reverse, const float* in, const float* w, const float* bias, const float* bnp_in,
                        float* a_out, float* out, float* partial, int accumulate, int T, int dil, int64_t S, int64_t Sp,
                        hipStream_t st, const float* bwd_y = nullptr, const float* bwd_bnp = nullptr,
                        const float* bwd_coef = nullptr, const float* stat_shift = nullptr, int bwd_store = 1,
                        int stat_records = 0);
int dof_tcn_stat_records();
int dof_launch_tcn_stat_merge(const float* partial, int64_t nblk, float* sums, hipStream_t st);
int dof_launch_tcn_stat_merge_fin(const float* partial, int64_t nblk, float* sums, float count, const float* gamma,
                                  const float* beta, float* rmean, float* rvar, float momentum, float* bnp, hipStream_t st);
// BatchNorm backward, `frozen`: the layer normalised with its running statistics (dof_vade_set_batchnorm_training(0)), so
// the batch-mean terms of the train-mode formula vanish -- dx = gamma rstd dy (torch's eval-mode batch_norm backward);
// the kernels get count = infinity, which makes both coefficients they derive from the sums exactly zero
int dof_launch_bn_bwd_sum_fin(const float* partial, int64_t nblk, float* sums, float count, float* dgamma, float* dbeta,
                              int accumulate, float* coef, hipStream_t st, bool frozen = false);
int64_t dof_tcn_bn_bwd1_blocks(int T, int64_t S);
int dof_launch_tcn_bn_stats(const float* y, float* partial, int64_t n_partial, int stride, float* sums, float count,
                            int T, int CT, int64_t S, int64_t Sp, hipStream_t st, const float* shift = nullptr);
int dof_tcn_combine_fold();
int dof_launch_tcn_conv_comb(const float* res, const float* y2, const float* bnp2, float* out_blk, const float* w,
                             const float* bias, float* out, float* partial, int T, int dil, int64_t S, int64_t Sp,
                             hipStream_t st, const float* stat_shift, int stat_records, float* relu_mask_out);
int dof_tcn_combine_fold0();
int dof_launch_tcn_conv_comb0(const float* xs, int F, const float* dsw, const float* dsb, const float* y2, const float* bnp2,
                              float* out_blk, const float* w, const float* bias, float* out, float* partial, int T, int dil,
                              int64_t S, int64_t Sp, hipStream_t st, int stat_records, float* relu_mask_out);
int dof_tcn_tail_fold();
int dof_launch_tcn_conv_tail(const float* dy, const float* w, const float* bwd_y, const float* bwd_bnp, const float* bwd_coef,
                             int bwd_store, const float* tail_src, const float* tail_mask, float* tail_gres,
                             const float* tail_skip, const float* tail_dfeat, const float* y2, const float* bnp2, float* g_out,
                             float* partial, float* sums, int T, int dil, int64_t S, int64_t Sp, hipStream_t st,
                             const float* wg_x = nullptr, float* wg_partials = nullptr, int64_t wg_part0 = 0, int64_t wg_part1 = 0);
// relu_mask_out / tail_mask / mask_out: [T][Sp] words, bit c = out[t][s][c] > 0 (the block output's ReLU mask)
int dof_launch_bn_fwd_fin(const float* sums, float count, const float* gamma, const float* beta, float* rmean,
                          float* rvar, float momentum, int train, float* bnp, int C, hipStream_t st, int shifted = 0);
int dof_launch_bn_bwd_fin(const float* sums, float count, float* dgamma, float* dbeta, int accumulate, float* coef,
                          int C, hipStream_t st, bool frozen = false);
int dof_launch_tcn_combine(const float* y2, const float* bnp2, const float* res, const float* xs, const float* dsw,
                           const float* dsb, float* out, float* skip, float* feat, int first, int T, int F, int CT,
                           int64_t S, int64_t Sp, hipStream_t st, int xs_ch = 0, int skip_last = 0, float* mask_out = nullptr);
int dof_launch_tcn_bn_bwd1(const float* din, const float* y, const float* bnp, float* g, float* partial, float* sums,
                           int blk, const float* out_blk, const float* dfeat, const float* skip, const float* dskip,
                           float* gres, int T, int CT, int64_t S, int64_t Sp, hipStream_t st, int last_step_only = 0);
int dof_tcn_last_block_sparse();
int dof_launch_tcn_bn_bwd2(float* g, const float* y, const float* bnp, const float* coef, int T, int CT, int64_t S,
                           int64_t Sp, hipStream_t st);
int dof_launch_tcn_convg(int reverse, int KC, int NC, const float* in, const float* w, int w_ci, int cin_real,
                         const float* bias, const float* bnp_in, float* a_out, float* out, float* partial,
                         int accumulate, int T, int dil, int64_t S, int64_t Sp, hipStream_t st);
// weight gradient of one 32 -> 32 dilated convolution (k = 4) with both operands staged through LDS; writes its
// partial tiles in the DofOuterJob layout of two 4-tile jobs (taps 0,1 -> part0, taps 2,3 -> part1; bias sums in
// column 64 of part0), so k_outer_finalize consumes them unchanged
struct DofTcnWgrad {
  const float* dy;  // [T][Sp][32] gradient w.r.t. the pre-BatchNorm conv output
  const float* in;  // [T][Sp][32] convolution input
  // lazy operands (all null: dy / in are used as they are).  dy_y: `dy` still holds the FIRST-pass gradient g of the
  // layer's BatchNorm backward; the staging applies pass 2, dy = scale (g - c1 - xhat c2), from the layer's
  // pre-normalisation tensor dy_y, its record dy_bnp and (mean g | mean g xhat) = dy_coef.  in_bnp: `in` is the
  // pre-normalisation tensor of the BatchNorm + ReLU in front of this convolution; the staging applies them.
  const float *dy_y, *dy_bnp, *dy_coef, *in_bnp;
  int dil, nblk, T;
  int64_t Sp;
  int64_t S;  // valid sequences (lazy dy: rows of padded sequences are zero gradients, not pass 2 of a zero)
  int64_t part0, part1;  // float offsets of the two jobs' partial regions ([nblk][64][65] each)
  // cin > 0 (k_tcn_wgrad_in, the first block): `in` is the raw input [T][Sp][cin] (cin = 3 or 1), part0 = the conv1 job's
  // region (tap j in columns 16 j .. 16 j + cin - 1, bias sums in column 64); dy2 != null: the gradient entering the
  // block's 1x1 residual convolution, whose weight / bias gradient goes to part1 (columns 0 .. cin - 1, 64)
  int cin;
  const float* dy2;
};
#define DOF_TCN_WGRAD_MAX_T 25     // k_tcn_wgrad (fp32 pipe) and the 4-sequence chunks of k_tcn_wgrad_b3
#define DOF_TCN_WGRAD_B3_MAX_T 50  // k_tcn_wgrad_b3 with 2-sequence chunks
int dof_tcn_wgrad_max_t(void);     // the longest window the LDS-staged weight-gradient kernel in use takes
int dof_launch_tcn_wgrad(const DofTcnWgrad* descs_dev, int n, int max_nblk, float* partials, hipStream_t st);
int dof_launch_tcn_wgrad_in(const DofTcnWgrad* descs_dev, int n, int max_nblk, float* partials, hipStream_t st);
int dof_launch_head_rms(const float* flat, float* hn, float* rinv, int J, int64_t B, int64_t Bp, hipStream_t st);
int dof_launch_head_dense(const float* in, const float* bnp_in, float* in_norm, const float* w, const float* bias,
                          float* out, float* partial, float* sums, int CI, int CO, int relu, int64_t B, int64_t Bp,
                          hipStream_t st);
int dof_launch_head_dense_bwd(const float* dout, const float* w, float* din, int CI, int CO, int64_t B, int64_t Bp,
                              hipStream_t st);
int dof_launch_head_bn_bwd(const float* g, const float* h, const float* bnp, float* partial, float* sums, float* coef,
                           float* dgamma, float* dbeta, int accumulate, float* dpre, int C, int64_t B, int64_t Bp,
                           hipStream_t st, int relu = 1, bool frozen = false);
int dof_launch_dec_repeat(const float* d2, const float* bnp, float* zrep, int C4, int T, int64_t B, int64_t Bp,
                          hipStream_t st);
int dof_launch_dec_sum_time(const float* dzrep, float* dzf, int C4, int T, int64_t B, int64_t Bp, hipStream_t st);
int dof_launch_tcn_dec_out(const float* skip, const float* wp, const float* bp, const float* x, const float* valid,
                           float* hid, float* loc_out, float* recon_partial, float* dloc, float* dskip, int T, int C3,
                           int train, int64_t B, int64_t Bp, hipStream_t st);
int dof_launch_head_rms_bwd(const float* dhn, const float* hn, const float* rinv, float* dflat, int J, int64_t B,
                            int64_t Bp, hipStream_t st);

// ---- k_tfm.hip (transformer family, models_new.py:832-1327) -----------------------------------
struct DofDrop {           // one dropout site
  const uint8_t* inject;   // keep-mask bytes in the reference's tensor order (parity tests) or null
  const uint32_t* ctr;     // device step counter mixed into the hash (fresh masks on every graph replay)
  uint32_t seed;           // site seed
  uint32_t thresh;         // keep iff hash >= thresh (= p * 2^32)
  float scale;             // 1 / (1 - p); 0 = site disabled (eval mode)
};
enum { DOF_EPI_NONE = 0, DOF_EPI_RELU, DOF_EPI_GELU, DOF_EPI_MUL_RELU, DOF_EPI_MUL_DGELU };
struct DofGemm {           // Y[r][n] (+)= epi(sum_k X[r][k] W[n][k] + bias[n]) over rows r = t*Sp + s, s < S
  const float* X; int ldx;
  const float* W; int ldw; // trans = 0: W[n*ldw + k]; trans = 1: W[k*ldw + n] (data gradients)
  const float* bias;
  float* Y; int ldy;
  const float* aux; int ldaux;  // DOF_EPI_MUL_RELU: saved activation; DOF_EPI_MUL_DGELU: saved pre-activation
  float* aux_out;               // DOF_EPI_GELU: pre-activation out (ld = ldy)
  DofDrop drop; int drop_ld;    // GELU epilogues: dropout over (row, col), reference index (s*T + t)*drop_ld + col
  int K, N, trans, epi, accumulate;
  int T; int64_t S, Sp;
  int its;                      // (set by the launcher) row tiles per wave
};
struct DofAttn {
  const float* qkv;   // [r][3D]: q | k | v, head h at columns h*dh
  float* ao;          // forward out [r][D]
  const float* dao;   // backward in [r][D]
  float* dqkv;        // backward out [r][3D]
  const float* pad;   // [T][Sp] 1 = padded key, or null
  DofDrop drop;
  int T, D, H, causal, nseq;
  float scale;
  int64_t S, Sp;
  int q_last;         // only the last query row (t = T - 1) is evaluated / differentiated (final encoder layer)
};
struct DofLn {         // u = x + drop(h) (h may be null), y = LayerNorm(u) (gamma null: residual add only)
  const float* x; const float* h; float* u; float* y; const float* gamma; const float* beta;
  DofDrop drop; int T; int64_t S, Sp; float eps;
  int t_off, T_idx;  // dropout index of row (t, s): (s * T_idx + t + t_off) -- a launch over the LAST time step of a
                     // longer tensor (T = 1, pointers offset) passes t_off = T_full - 1, T_idx = T_full; 0, 0 = (0, T)
};
struct DofLnBwd {
  const float* dy1; const float* dy2; const float* dres; const float* u; const float* gamma;
  float* du; float* dh; float* partial; DofDrop drop; int T; int64_t S, Sp; float eps;
  int t_off, T_idx;
};
struct DofDecExp {     // TFMDecoderPT.latent_expand; per-window tensors [c][Bp]
  const float* z; const float *w0, *b0, *w1, *b1, *w2, *b2;
  float *a1, *g1, *a2, *g2, *a3, *g3;
  int keep; int64_t B, Bp;
};
int dof_launch_tfm_tick(uint32_t* ctr, hipStream_t st);
int dof_launch_tfm_embed(int F, const float* xin, const float* w, const float* bias, const float* pe, float* xs,
                         float* pad, float* y, const DofDrop& drop, int T, int G, int D, int64_t S, int64_t Sp,
                         hipStream_t st);
int dof_launch_tfm_embed_bwd(int F, const float* xs, const float* w, const float* bias, const float* dy, float* dpre,
                             const DofDrop& drop, int T, int D, int64_t S, int64_t Sp, hipStream_t st);
int dof_launch_tfm_gemm(const DofGemm& g, hipStream_t st);
bool dof_tfm_attn_fits(int T, int D, int H);
int dof_launch_tfm_attn(DofAttn a, int backward, hipStream_t st);
int64_t dof_tfm_ln_blocks(int C, int T, int64_t Sp);
int dof_launch_tfm_add_ln(const DofLn& a, int C, hipStream_t st);
int dof_launch_tfm_ln_bwd(const DofLnBwd& a, int C, hipStream_t st);
int dof_launch_tfm_last(const float* x, float* n2, int T, int D, int64_t S, int64_t Sp, hipStream_t st);
int dof_launch_tfm_last_bwd(const float* dn2, float* dx, int T, int D, int64_t S, int64_t Sp, hipStream_t st);
int dof_launch_tfm_bstd(const float* in, float* out, float* stat, int L, int standardize, int64_t B, int64_t Bp,
                        hipStream_t st);
int dof_launch_tfm_bstd_bwd(const float* dy, const float* y, const float* stat, float* dx, int L, int standardize,
                            int64_t B, int64_t Bp, hipStream_t st);
int dof_launch_tfm_dec_expand(int L, const DofDecExp& a, hipStream_t st);
int dof_launch_tfm_dec_expand_bwd(int L, const DofDecExp& a, const float* dg3, float* da3, float* da2, float* da1,
                                  float* dz, hipStream_t st);
int dof_launch_tfm_dec_h0(const float* g3, const float* pe, float* h0, int T, int D, int64_t B, int64_t Bp,
                          hipStream_t st);
int dof_launch_tfm_dec_sum_time(const float* dh0, float* dg3, int T, int D, int64_t B, int64_t Bp, hipStream_t st);
int dof_launch_tfm_dec_logp(const float* loc, int ld, const float* x, const float* valid, float* loc_out,
                            float* recon_partial, float* dloc, int T, int C3, int train, int64_t B, int64_t Bp,
                            hipStream_t st);

// ---- k_reduce.hip ---------------------------------------------------------------------------
// Weight-gradient reductions: out[i][j] = sum_{t,s} A[t][i][s] * B[t+shift][j][s] as fp32 MFMA
// (16x16x4) tiles over SoA operands, many jobs per launch, per-block partials + fixed-order
// finalize (run-to-run deterministic).
struct DofOuterTile {  // one 16-column tile of the B operand
  const float* ptr;
  int64_t t_stride, s_stride, c_stride;  // element strides of the time, sequence and channel axes
  int nc;                      // valid columns (<=16); rest masked to 0
  int shift;                   // B is read at time t+shift (skipped when outside [0,T))
  int pack;                    // 0: column i = channel i.  > 0: several convolution taps share the tile -- column i
                               // = channel i % pack read at time t + shift + i / pack (a k=5 conv over 3 input
                               // channels is one 15-column tile instead of five 3-column ones)
};
struct DofOuterJob {
  const float* a_ptr;
  int64_t a_tstride, a_sstride, a_cstride;
  int a_rows;     // valid A rows (<= 64)
  int T;          // time steps
  int64_t Sp;     // padded sequence count (multiple of 64)
  int n_tiles;    // 1..4
  DofOuterTile tile[4];
  int blk0, nblk;        // block range of this job inside the launch
  int64_t partial_off;   // float offset of this job's partials: [nblk][64][65]  (col 64 = row sums of A)
  // Jobs that read the same operand rows (the row blocks x tile groups of one wide layer) form a group of grp_jobs consecutive
  // jobs with equal nblk (a multiple of 8), first job grp_job0, first block grp_blk0: inside the group's block range the
  // workgroups are dealt out so that the ones an XCD receives back to back (block indices b, b + 8, b + 16, ...) are the
  // group's jobs on the SAME K slice -- they stream the same rows at the same time and share them in that XCD's L2
  // (workgroup v of the group: XCD x = v % 8, q = v / 8 -> job q % grp_jobs, slice (q / grp_jobs) * 8 + x).  0 / 1: no group.
  int grp_job0, grp_jobs, grp_blk0;
};
struct DofFinJob {  // scatter-add of one reduced (rows x cols) block into a gradient tensor
  int job;              // source DofOuterJob index
  int col0;             // first source column (tile*16 + offset), or 64 for the row-sum column
  int rows, cols;       // extent
  int r1, r2;           // source row i maps to dst row i (i < r1) or i - (r2 - r1) (i >= r2); rows in [r1,r2) skipped
  int64_t dst_off;      // float offset into the grad buffer
  int64_t row_stride, col_stride;
  int elem0;            // prefix offset of this fin-job's elements in the finalize launch
  int gate_minor;       // HID > 0: the job's A rows are unit-major (row = unit * 4 + gate; the lane-per-unit GRU kernels
                        // keep a unit's four gate values in one 16-byte word), 0: gate-major (row = gate * HID + unit)
};
#define DOF_OUTER_PARTIAL_FLOATS (64 * 65)

int dof_launch_outer(const DofOuterJob* jobs_dev, int njobs, int total_blocks, float* partials, hipStream_t st);
int dof_launch_outer_finalize(const DofOuterJob* jobs_dev, const DofFinJob* fin_dev, int n_fin, int total_elems,
                              const float* partials, float* grads, int accumulate, hipStream_t st);
// out[dst_off + v] (+)= sum_b partial[b][v]   (fixed order)
int dof_launch_sum_partials(const float* partial, int64_t nblk, int nv, float* out, int accumulate, hipStream_t st);
// up to 4 such reductions in one launch (the step has eight of them, each a launch-latency-bound 5 us kernel)
struct DofSumJobs {
  int n;
  const float* partial[8];
  int64_t nblk[8];
  int nv[8];
  float* out[8];
};
int dof_launch_sum_partials_multi(const DofSumJobs& jobs, int accumulate, hipStream_t st);
// the latent-8 recurrent step's three end-of-step reductions in one launch (k_step_finalize): n16 (2 or 3) first-layer
// weight-gradient partial sets, k_gru8x_bwd's two second-layer sets, the plain partial sums
bool dof_step_finalize_selected(const int64_t S8[2], int T);
int dof_launch_step_finalize(const float* const* wg16, const int64_t* S16, const int64_t* const* off16, int n16,
                             const float* const wg8[2], const int64_t S8[2], const int64_t* const off8[2],
                             const DofSumJobs& sums, float* g, int accumulate, hipStream_t st);

struct DofAdamSeg {  // one contiguous parameter range with its own lr / step count / freeze flag
  int64_t lo, hi;
  int lr_index;   // index into hyper[] of the learning rate
  int bc_index;   // (unused since ABI 9: the bias corrections come from the device-side step counters)
  int active_index;  // index into hyper[] of the 0/1 "has gradient" flag
};
int dof_launch_clip_adam(float* params, const float* grads, float* m, float* v, const float* hyper,
                         const DofAdamSeg* segs_dev, int nseg, int64_t total, int clip_index, const float* mask,
                         int* opt_state, int* ticket, float grad_scale, hipStream_t st);
struct DofSchedItems {  // by-value kernel argument of k_schedule_apply
  int n;
  DofSchedItem item[DOF_SCHED_MAX_ITEMS];
};
struct DofNoiseArgs {  // by-value kernel argument of k_step_begin
  float* out[2];
  int64_t n[2];
  int64_t quads0;  // ceil(n[0] / 4): Philox calls of buffer 0
  uint32_t key0, key1;
  int* state;      // device int32[2]: step counter, ticket; null = no noise
};
int dof_launch_step_begin(float* hyper, const DofSchedItems& items, const DofNoiseArgs& noise, hipStream_t st);
int dof_launch_schedule_apply(float* hyper, const DofSchedItems& items, hipStream_t st);

// ---- k_graph_latent.hip ----------------------------------------------------------------------
struct DofTriplets {  // CSR of (partner row m, other-stream element o, coefficient) grouped by a key row
  const int* ptr;     // [G+1]
  const int* m;       // partner index in the SAME stream
  const int* o;       // index in the OTHER stream (whose dot-product weights the pair)
  const int* r;       // output row (used by the by-m / by-o orderings)
  const float* coef;  // Laplacian entry
};
