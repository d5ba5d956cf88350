// CensNet graph layer, GMM latent space, and the batch-level VaDE loss terms
// (SURVEY.md section 8a rows R4, R5, R6, R8, R9).
//
// Reference semantics restated here:
//   * CensNetConvPT._propagate_nodes/_propagate_edges  /root/reference/deepof/clustering/censNetConv_pt.py:92-136
//     out_v = relu(((T diag(Xe.pe) T^T) * Lap_v) Xv Kv + bv)  -- evaluated through the sparsity of
//     the Laplacians: a host-built list of (row r, partner m, other-stream element o, coef=Lap[r][m])
//     triplets, exact for any adjacency;
//   * RecurrentEncoderPT tail (flatten+concat -> Linear)    models_new.py:163-181
//   * GaussianMixtureLatentPT                               models_new.py:1724-1791
//   * compute_kmeans_loss_pt (fp64 spectrum of the Gram)    losses.py:257-287
//   * VadeLoss terms                                        losses.py:567-797
#include "dof_rt.h"
#include "launchers.h"
#include "deepof_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// CensNet
// ---------------------------------------------------------------------------------------------
struct CensStream {
  const float* X;      // [D][Sp]  block outputs of this stream, s = b*G + g
  const float* dots;   // [Sp_other] dot products of the OTHER stream (weights the pairs)
  DofTriplets tri;     // grouped by output row r
  const float* kern;   // (D, L)
  const float* bias;   // (L)
  float* Y;            // [D][Sp]  propagated features (saved for d kern)
  float* Z;            // [L][Sp]  relu output (saved for the mask)
  int G, G_other;
  int64_t S, Sp;
  int flat_row0;       // first row of this stream inside flat [(N+E)L][Bp]
};

template <int D>
__global__ void __launch_bounds__(256) k_cens_dots(const float* __restrict__ X, const float* __restrict__ p,
                                                   float* __restrict__ dots, int64_t S, int64_t Sp) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < D; ++c) acc = fmaf(X[(int64_t)c * Sp + s], p[c], acc);
  dots[s] = acc;
}

// the same with the width as a run-time value (TCN: 32 channels, transformer: key_dim); same summation order
__global__ void __launch_bounds__(256) k_cens_dots_rt(const float* __restrict__ X, const float* __restrict__ p,
                                                      float* __restrict__ dots, int D, int64_t S, int64_t Sp) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float acc = 0.0f;
  for (int c = 0; c < D; ++c) acc = fmaf(X[(int64_t)c * Sp + s], p[c], acc);
  dots[s] = acc;
}

template <int L, int D>
__global__ void __launch_bounds__(256) k_cens_fwd(CensStream A, CensStream Bs, float* __restrict__ flat, int64_t Bp) {
  const CensStream& P = blockIdx.y ? Bs : A;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= P.S) return;
  const int64_t b = s / P.G;
  const int r = (int)(s - b * P.G);
  float y[D];
#pragma unroll
  for (int c = 0; c < D; ++c) y[c] = 0.0f;
  for (int e = P.tri.ptr[r]; e < P.tri.ptr[r + 1]; ++e) {
    const float w = P.tri.coef[e] * P.dots[b * P.G_other + P.tri.o[e]];
    const int64_t sm = b * P.G + P.tri.m[e];
#pragma unroll
    for (int c = 0; c < D; ++c) y[c] = fmaf(w, P.X[(int64_t)c * P.Sp + sm], y[c]);
  }
#pragma unroll
  for (int c = 0; c < D; ++c) P.Y[(int64_t)c * P.Sp + s] = y[c];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float acc = P.bias[l];
#pragma unroll
    for (int c = 0; c < D; ++c) acc = fmaf(y[c], P.kern[c * L + l], acc);
    acc = acc < 0.0f ? 0.0f : acc;   // torch.relu: a NaN stays a NaN (a sequence with every key masked, transformer cores)
    P.Z[(int64_t)l * P.Sp + s] = acc;
    flat[(int64_t)(P.flat_row0 + r * L + l) * Bp + b] = acc;
  }
}

struct CensBwdStream {
  const float* X;       // [D][Sp] this stream's inputs
  const float* dots;    // [Sp_other]
  const float* Z;       // [L][Sp]
  const float* kern;    // (D,L)
  const float* pw;      // (D) this stream's dot-product weights (used by the OTHER stream's update)
  float* dZ;            // [L][Sp] out: grad at pre-activation
  float* dY;            // [D][Sp] out
  DofTriplets by_m;     // this stream's update, grouped by partner m  (fields: r, o, coef)
  DofTriplets oth_by_o; // the OTHER stream's update, grouped by o = element of THIS stream (fields: r, m, coef)
  const float* X_oth;   // [D][Sp_other]
  const float* dY_oth;  // [D][Sp_other]
  float* dX;            // [D][Sp] out: grad wrt this stream's block output
  float* dd;            // [Sp]    out: grad wrt this stream's dot products
  int G, G_other;
  int64_t S, Sp, Sp_other;
  int flat_row0;
  // recurrent encoder: the block output is the LayerNorm of the final GRU2 state -- its backward runs right here on the
  // row the thread holds (k_ln_bwd<D, true>'s arithmetic), dX then is the gradient at the LayerNorm INPUT
  const float* ln_x = nullptr;      // [D][Sp] LayerNorm input (the final hidden states) or null
  const float* ln_gamma = nullptr;  // (D)
  float* ln_partial = nullptr;      // [cdiv(S, 256)][2 D] (sum dy xhat | sum dy) per workgroup
};

template <int L, int D>
__global__ void __launch_bounds__(256) k_cens_bwd1(CensBwdStream A, CensBwdStream Bs, const float* __restrict__ dflat,
                                                   int64_t Bp) {
  const CensBwdStream& P = blockIdx.y ? Bs : A;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= P.S) return;
  const int64_t b = s / P.G;
  const int r = (int)(s - b * P.G);
  float dz[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float g = dflat[(int64_t)(P.flat_row0 + r * L + l) * Bp + b];
    dz[l] = P.Z[(int64_t)l * P.Sp + s] > 0.0f ? g : 0.0f;
    P.dZ[(int64_t)l * P.Sp + s] = dz[l];
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) acc = fmaf(P.kern[c * L + l], dz[l], acc);
    P.dY[(int64_t)c * P.Sp + s] = acc;
  }
}

template <int L, int D>
__global__ void __launch_bounds__(256) k_cens_bwd2(CensBwdStream A, CensBwdStream Bs) {
  const CensBwdStream& P = blockIdx.y ? Bs : A;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = s < P.S;
  if (!live && !P.ln_x) return;
  float dx[D];
#pragma unroll
  for (int c = 0; c < D; ++c) dx[c] = 0.0f;
  if (live) {
    const int64_t b = s / P.G;
    const int m = (int)(s - b * P.G);
    // (a) through the propagated features of THIS stream: Y[r] += coef * dots[o] * X[m]
    for (int e = P.by_m.ptr[m]; e < P.by_m.ptr[m + 1]; ++e) {
      const float w = P.by_m.coef[e] * P.dots[b * P.G_other + P.by_m.o[e]];
      const int64_t sr = b * P.G + P.by_m.r[e];
#pragma unroll
      for (int c = 0; c < D; ++c) dx[c] = fmaf(w, P.dY[(int64_t)c * P.Sp + sr], dx[c]);
    }
    // (b) through this element's dot product, which weights pairs of the OTHER stream's update
    float dd = 0.0f;
    for (int e = P.oth_by_o.ptr[m]; e < P.oth_by_o.ptr[m + 1]; ++e) {
      const int64_t sr = b * P.G_other + P.oth_by_o.r[e];
      const int64_t sm = b * P.G_other + P.oth_by_o.m[e];
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < D; ++c) acc = fmaf(P.X_oth[(int64_t)c * P.Sp_other + sm], P.dY_oth[(int64_t)c * P.Sp_other + sr], acc);
      dd = fmaf(P.oth_by_o.coef[e], acc, dd);
    }
    P.dd[s] = dd;
#pragma unroll
    for (int c = 0; c < D; ++c) dx[c] = fmaf(dd, P.pw[c], dx[c]);
  }
  if (!P.ln_x) {
#pragma unroll
    for (int c = 0; c < D; ++c) P.dX[(int64_t)c * P.Sp + s] = dx[c];
    return;
  }
  // ---- LayerNorm backward of the row (dy = dx): dX = rstd (g - mean g - xhat mean(g xhat)), g = dy gamma
  float vals[2 * D];
#pragma unroll
  for (int c = 0; c < 2 * D; ++c) vals[c] = 0.0f;
  if (live) {
    float x[D];
    float mean = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      x[c] = P.ln_x[(int64_t)c * P.Sp + s];
      mean += x[c];
    }
    mean *= (1.0f / D);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      x[c] -= mean;
      var = fmaf(x[c], x[c], var);
    }
    const float rstd = rsqrtf(var * (1.0f / D) + 1e-3f);
    float mg = 0.0f, mgx = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      x[c] *= rstd;  // xhat
      const float g = dx[c] * P.ln_gamma[c];
      mg += g;
      mgx = fmaf(g, x[c], mgx);
      vals[c] = dx[c] * x[c];
      vals[D + c] = dx[c];
    }
    mg *= (1.0f / D);
    mgx *= (1.0f / D);
#pragma unroll
    for (int c = 0; c < D; ++c) P.dX[(int64_t)c * P.Sp + s] = rstd * (dx[c] * P.ln_gamma[c] - mg - x[c] * mgx);
  }
  if ((int64_t)blockIdx.x * blockDim.x < P.S) dof_block_colsum<2 * D>(vals, P.ln_partial + (int64_t)blockIdx.x * 2 * D);
}

// ---------------------------------------------------------------------------------------------
// Latent forward: final dense, mean / softplus(log-var) heads, reparameterisation, GMM posterior.
// ---------------------------------------------------------------------------------------------
struct LatentFwdArgs {
  const float* flat;  // [J][Bp]
  int J;
  const float *wf, *bf, *wm, *bm, *ws, *bs;  // final_dense (L,J); encoder_mean (L,L); encoder_log_var (L,L)
  const float *gmm_means, *gmm_log_vars, *prior;  // (K,L), (K,L), (K)
  const float* eps;   // (B,L) row-major or null (eval: z = mean)
  float *enc, *mu, *pre, *sv, *z;  // [L][Bp]
  float *q, *qn;      // [K][Bp] posterior, clamp+renormalised posterior
  float *z_out, *q_out, *mu_out, *sv_out, *enc_out;  // reference-layout exports (B,L)/(B,K) or null
  // k_latent_fwd_w: the Gram Z^T Z of the workgroup's 16 windows, [workgroup][L][L] (what k_kmeans_eig sums), or null
  float* gram_partial = nullptr;
  int K;
  int64_t B, Bp;
};

// encoder.final_dense: enc[l][b] = bf[l] + sum_j wf[l][j] flat[j][b]; thread = (b, l = blockIdx.y)
__global__ void __launch_bounds__(256) k_final_dense(const float* __restrict__ flat, const float* __restrict__ wf_,
                                                     const float* __restrict__ bf_, float* __restrict__ enc, int J,
                                                     int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int l = blockIdx.y;
  const dof_cfp wf = dof_cw(wf_) + (int64_t)l * J;
  float a0 = dof_cw(bf_)[l], a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int j = 0;
  // 16 activations in flight per thread (the kernel is 32 workgroups of pure load latency at batch 1024); the four
  // partial sums and their order are those of the 4-wide loop
  for (; j + 15 < J; j += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = flat[(int64_t)(j + u) * Bp + b];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      a0 = fmaf(wf[j + u], v[u], a0);
      a1 = fmaf(wf[j + u + 1], v[u + 1], a1);
      a2 = fmaf(wf[j + u + 2], v[u + 2], a2);
      a3 = fmaf(wf[j + u + 3], v[u + 3], a3);
    }
  }
  for (; j + 3 < J; j += 4) {
    a0 = fmaf(wf[j], flat[(int64_t)j * Bp + b], a0);
    a1 = fmaf(wf[j + 1], flat[(int64_t)(j + 1) * Bp + b], a1);
    a2 = fmaf(wf[j + 2], flat[(int64_t)(j + 2) * Bp + b], a2);
    a3 = fmaf(wf[j + 3], flat[(int64_t)(j + 3) * Bp + b], a3);
  }
  for (; j < J; ++j) a0 = fmaf(wf[j], flat[(int64_t)j * Bp + b], a0);
  enc[(int64_t)l * Bp + b] = (a0 + a1) + (a2 + a3);
}

// d flat[j][b] = sum_l wf[l][j] denc[l][b]; thread = (b, j = blockIdx.y)
template <int L>
__global__ void __launch_bounds__(256) k_final_dense_bwd(const float* __restrict__ denc, const float* __restrict__ wf_,
                                                         float* __restrict__ dflat, int J, int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int j = blockIdx.y;
  const dof_cfp wf = dof_cw(wf_);
  float acc = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) acc = fmaf(wf[l * J + j], denc[(int64_t)l * Bp + b], acc);
  dflat[(int64_t)j * Bp + b] = acc;
}

constexpr int kLatentMaxKL = 1024;  // K * L entries of the per-component constants staged in LDS (K <= 128 at L = 8)

template <int L>
__global__ void __launch_bounds__(256) k_latent_fwd(LatentFwdArgs A) {
  // per (component, dimension) constants of the posterior: 1 / sd and the component's additive term
  // log(prior + 1e-9) - sum_d (log sd + log(2 pi) / 2), once per workgroup instead of an expf, a logf and a division
  // per (window, component, dimension)
  __shared__ float s_inv[kLatentMaxKL];
  __shared__ float s_const[kLatentMaxKL / 4];
  const bool staged = A.K * L <= kLatentMaxKL;
  if (staged) {
    for (int e = threadIdx.x; e < A.K * L; e += 256) s_inv[e] = 1.0f / fmaxf(expf(0.5f * A.gmm_log_vars[e]), 1e-3f);
    for (int c = threadIdx.x; c < A.K; c += 256) {
      float acc = logf(A.prior[c] + 1e-9f);
      for (int d = 0; d < L; ++d) acc += -logf(fmaxf(expf(0.5f * A.gmm_log_vars[c * L + d]), 1e-3f)) - 0.9189385332046727f;
      s_const[c] = acc;
    }
    __syncthreads();
  }
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  const dof_cfp wm = dof_cw(A.wm), bm = dof_cw(A.bm), wsv = dof_cw(A.ws),
                bsv = dof_cw(A.bs), gmeans = dof_cw(A.gmm_means), glv = dof_cw(A.gmm_log_vars), prior = dof_cw(A.prior);
  float enc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) enc[l] = A.enc[(int64_t)l * A.Bp + b];
  float z[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float m = bm[l], p = bsv[l];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      m = fmaf(wm[l * L + k], enc[k], m);
      p = fmaf(wsv[l * L + k], enc[k], p);
    }
    const float sv = dof_softplus(p);
    z[l] = A.eps ? fmaf(expf(0.5f * sv), A.eps[b * L + l], m) : m;
    A.mu[(int64_t)l * A.Bp + b] = m;
    A.pre[(int64_t)l * A.Bp + b] = p;
    A.sv[(int64_t)l * A.Bp + b] = sv;
    A.z[(int64_t)l * A.Bp + b] = z[l];
    if (A.z_out) A.z_out[b * L + l] = z[l];
    if (A.mu_out) A.mu_out[b * L + l] = m;
    if (A.sv_out) A.sv_out[b * L + l] = sv;
    if (A.enc_out) A.enc_out[b * L + l] = enc[l];
  }
  // posterior: softmax_c(log(prior+1e-9) + sum_d log N(z_d; m_cd, max(exp(l_cd/2),1e-3)))
  const float HALF_LOG_2PI = 0.9189385332046727f;
  float mx = -INFINITY;
  for (int c = 0; c < A.K; ++c) {
    float lg;
    if (staged) {
      lg = s_const[c];
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float u = (z[d] - gmeans[c * L + d]) * s_inv[c * L + d];
        lg = fmaf(-0.5f * u, u, lg);
      }
    } else {
      lg = logf(prior[c] + 1e-9f);
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float sd = fmaxf(expf(0.5f * glv[c * L + d]), 1e-3f);
        const float u = (z[d] - gmeans[c * L + d]) / sd;
        lg += -0.5f * u * u - logf(sd) - HALF_LOG_2PI;
      }
    }
    A.q[(int64_t)c * A.Bp + b] = lg;
    mx = fmaxf(mx, lg);
  }
  float sum = 0.0f;
  for (int c = 0; c < A.K; ++c) {
    const float e = expf(A.q[(int64_t)c * A.Bp + b] - mx);
    A.q[(int64_t)c * A.Bp + b] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  float csum = 0.0f;
  for (int c = 0; c < A.K; ++c) {
    const float qv = A.q[(int64_t)c * A.Bp + b] * inv;
    A.q[(int64_t)c * A.Bp + b] = qv;
    if (A.q_out) A.q_out[b * A.K + c] = qv;
    csum += fmaxf(qv, 1e-8f);
  }
  const float cinv = 1.0f / csum;
  for (int c = 0; c < A.K; ++c) A.qn[(int64_t)c * A.Bp + b] = fmaxf(A.q[(int64_t)c * A.Bp + b], 1e-8f) * cinv;
}

constexpr int kLatRows = 16;  // windows per workgroup of the row kernels

// k_latent_fwd with the work of a window spread over a 16-lane DPP row.  These kernels run one wavefront per SIMD
// (1,024 windows), so their duration is the instruction count of a lane times four cycles: nothing is replicated.
// Lane j of a row owns components j, j + 16, ... (NC per lane, K <= 16 NC), latent dimension j mod L, and every
// 16th input of encoder.final_dense (evaluated here when A.flat is set -- one launch instead of two); row sums and
// maxima are four DPP steps, single values travel by row broadcast.
constexpr int kWfMax = 4096;  // J * L entries of the final_dense weight staged in LDS (else read from memory)

template <int L, int NC>
__global__ void __launch_bounds__(256) k_latent_fwd_w(LatentFwdArgs A) {
  constexpr int KMAX = 16 * NC;
  __shared__ float s_inv[KMAX * L], s_gm[KMAX * L], s_const[KMAX];
  __shared__ float s_zz[kLatRows][L * L];   // z z^T of each window (gram_partial)
  __shared__ float s_wm[L * L], s_ws[L * L], s_bm[L], s_bs[L];
  __shared__ float s_wf[kWfMax];  // [input][l]
  const int K = A.K;
  const bool wf_lds = A.flat && A.J * L <= kWfMax;
  for (int e = threadIdx.x; e < K * L; e += 256) {
    s_inv[e] = 1.0f / fmaxf(expf(0.5f * A.gmm_log_vars[e]), 1e-3f);
    s_gm[e] = A.gmm_means[e];
  }
  for (int c = threadIdx.x; c < K; c += 256) {
    float acc = logf(A.prior[c] + 1e-9f);
    for (int d = 0; d < L; ++d) acc += -logf(fmaxf(expf(0.5f * A.gmm_log_vars[c * L + d]), 1e-3f)) - 0.9189385332046727f;
    s_const[c] = acc;
  }
  if (threadIdx.x < L * L) {
    s_wm[threadIdx.x] = A.wm[threadIdx.x];
    s_ws[threadIdx.x] = A.ws[threadIdx.x];
  }
  if (threadIdx.x < L) {
    s_bm[threadIdx.x] = A.bm[threadIdx.x];
    s_bs[threadIdx.x] = A.bs[threadIdx.x];
  }
  if (wf_lds)
    for (int e = threadIdx.x; e < A.J * L; e += 256) {
      const int l = e / A.J;
      s_wf[(e - l * A.J) * L + l] = A.wf[e];
    }
  __syncthreads();
  const int j = (int)threadIdx.x & 15, row = (int)threadIdx.x >> 4;
  const int l = j % L;
  const int64_t b_raw = (int64_t)blockIdx.x * kLatRows + row;
  const bool live = b_raw < A.B;
  const int64_t b = live ? b_raw : A.B - 1;  // idle rows shadow the last window (DPP reads every lane)
  float enc[L];
  if (A.flat) {
    float part[L];
#pragma unroll
    for (int k = 0; k < L; ++k) part[k] = 0.0f;
    for (int jj = j; jj < A.J; jj += 16) {
      const float f = A.flat[(int64_t)jj * A.Bp + b];
#pragma unroll
      for (int k = 0; k < L; ++k) part[k] = fmaf(wf_lds ? s_wf[jj * L + k] : A.wf[k * A.J + jj], f, part[k]);
    }
#pragma unroll
    for (int k = 0; k < L; ++k) enc[k] = dof_row16_sum(part[k]) + A.bf[k];
  } else {
#pragma unroll
    for (int k = 0; k < L; ++k) enc[k] = A.enc[(int64_t)k * A.Bp + b];
  }
  float m = s_bm[l], p = s_bs[l], enc_l = 0.0f;
#pragma unroll
  for (int k = 0; k < L; ++k) {
    m = fmaf(s_wm[l * L + k], enc[k], m);
    p = fmaf(s_ws[l * L + k], enc[k], p);
    if (k == l) enc_l = enc[k];
  }
  const float sv = dof_softplus(p);
  const float zl = A.eps ? fmaf(expf(0.5f * sv), A.eps[b * L + l], m) : m;
  if (live && j < L) {
    if (A.flat) A.enc[(int64_t)l * A.Bp + b] = enc_l;
    A.mu[(int64_t)l * A.Bp + b] = m;
    A.pre[(int64_t)l * A.Bp + b] = p;
    A.sv[(int64_t)l * A.Bp + b] = sv;
    A.z[(int64_t)l * A.Bp + b] = zl;
    if (A.z_out) A.z_out[b * L + l] = zl;
    if (A.mu_out) A.mu_out[b * L + l] = m;
    if (A.sv_out) A.sv_out[b * L + l] = sv;
    if (A.enc_out) A.enc_out[b * L + l] = enc_l;
  }
  float z[L];
  dof_static_for<L>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    z[d] = dof_gbcast<d, 16>(zl);
  });
  if (A.gram_partial) {  // (round 5) the k-means term's Gram, 16 windows here, the workgroups' tiles summed by k_kmeans_eig:
    // the separate reduction launch over z (k_outer, ~17 us at C2) is gone.  Row sums in window order.
    if (j < L) {
#pragma unroll
      for (int d = 0; d < L; ++d) s_zz[row][l * L + d] = live ? zl * z[d] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L * L; e += 256) {
      float acc = 0.0f;
#pragma unroll
      for (int r = 0; r < kLatRows; ++r) acc += s_zz[r][e];
      A.gram_partial[(int64_t)blockIdx.x * (L * L) + e] = acc;
    }
  }
  // posterior: softmax_c(log(prior+1e-9) + sum_d log N(z_d; m_cd, max(exp(l_cd/2),1e-3)))
  float lg[NC], mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = j + 16 * i;
    if (c < K) {
      float acc = s_const[c];
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float u = (z[d] - s_gm[c * L + d]) * s_inv[c * L + d];
        acc = fmaf(-0.5f * u, u, acc);
      }
      lg[i] = acc;
    } else {
      lg[i] = -INFINITY;
    }
    mx = fmaxf(mx, lg[i]);
  }
  mx = dof_row16_max(mx);
  float e[NC], sum = 0.0f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    e[i] = j + 16 * i < K ? expf(lg[i] - mx) : 0.0f;
    sum += e[i];
  }
  const float inv = 1.0f / dof_row16_sum(sum);
  float csum = 0.0f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    e[i] *= inv;
    if (j + 16 * i < K) csum += fmaxf(e[i], 1e-8f);
  }
  const float cinv = 1.0f / dof_row16_sum(csum);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = j + 16 * i;
    if (c < K && live) {
      A.q[(int64_t)c * A.Bp + b] = e[i];
      if (A.q_out) A.q_out[b * K + c] = e[i];
      A.qn[(int64_t)c * A.Bp + b] = fmaxf(e[i], 1e-8f) * cinv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Gram spectrum ("k-means" loss): value and the matrix Pm with dLoss/dZ = Z * Pm.
//   loss = w * mean_i sqrt(max(lambda_i(Z^T Z / B), 1e-9)) ;  dLoss/dZ = w/(L*B) * Z * G^{-1/2}
// One thread, cyclic Jacobi in fp64 on the L x L Gram (formed in fp32 like the reference).
// ---------------------------------------------------------------------------------------------
// partial != null: the Gram's partial tiles ([nblk][64][65] of the weight-gradient reduction, one job: row_stride 65,
// tile_stride DOF_OUTER_PARTIAL_FLOATS; or k_latent_fwd_w's [nblk][L][L]: row_stride L, tile_stride L L) are reduced here
// first -- k_outer_finalize's arithmetic per element (64 lanes stride over the tiles, fixed butterfly), 16 wavefronts x
// 4 elements -- and also written to gram_sum: the finalize launch in front of this kernel is gone.
struct KmeansEigArgs {   // (an argument set with partial == null and gram_sum == null: "not in this launch")
  float* gram_sum; const float* partial; int nblk; const float* hyper; int64_t B; float* km_out; float* Pm;
  int row_stride, tile_stride;
};
template <int L>
__device__ __forceinline__ void kmeans_eig_body(float* __restrict__ gram_sum, const float* __restrict__ partial, int nblk,
                                                const float* __restrict__ hyper, int64_t B,
                                                float* __restrict__ km_out /*[0]=weighted loss*/,
                                                float* __restrict__ Pm /*[L][L]*/, int row_stride, int tile_stride) {
  if (partial) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int e = wv; e < L * L; e += nw) {
      const float* __restrict__ p = partial + (int64_t)(e / L) * row_stride + (e % L);
      float acc = 0.0f;
      for (int b = lane; b < nblk; b += 64) acc += p[(int64_t)b * tile_stride];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
      if (lane == 0) gram_sum[e] = acc;
    }
    __syncthreads();  // (the same workgroup reads gram_sum below: global writes of a workgroup are visible to it after the barrier)
  }
  if (threadIdx.x != 0) return;
  const float w_lat = hyper[DOF_H_KM_LATENT];
  const float w_loss = hyper[DOF_H_KM_LOSS];
  if (!(w_lat > 0.0f) || w_loss == 0.0f) {  // value and gradient are multiplied by both weights
    km_out[0] = 0.0f;
    for (int i = 0; i < L * L; ++i) Pm[i] = 0.0f;
    return;
  }
  // (LDS, not per-thread arrays: dynamically indexed private arrays are scratch memory, which every wavefront of the launches
  // this body is merged into would be sized for)
  __shared__ double a[L][L], v[L][L], inv[L];
  for (int i = 0; i < L; ++i)
    for (int j = 0; j < L; ++j) {
      a[i][j] = (double)(gram_sum[i * L + j] / (float)B);
      v[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int i = 0; i < L; ++i)
    for (int j = i + 1; j < L; ++j) a[i][j] = a[j][i] = 0.5 * (a[i][j] + a[j][i]);
  double diag2 = 0.0;
  for (int i = 0; i < L; ++i) diag2 += a[i][i] * a[i][i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < L; ++i)
      for (int j = i + 1; j < L; ++j) off += a[i][j] * a[i][j];
    if (off <= 1e-30 * diag2 || off < 1e-300) break;
    for (int p = 0; p < L; ++p)
      for (int q = p + 1; q < L; ++q) {
        if (fabs(a[p][q]) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < L; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < L; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < L; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  double val = 0.0;
  for (int i = 0; i < L; ++i) {
    const double lam = fabs(a[i][i]);  // singular values of a symmetric matrix = |eigenvalues|
    val += sqrt(lam > 1e-9 ? lam : 1e-9);
    inv[i] = lam > 1e-9 ? 1.0 / sqrt(lam) : 0.0;
  }
  const double w = (double)w_lat;
  km_out[0] = (float)(w * val / L) * w_loss;
  const double scale = (double)w_lat * (double)w_loss / ((double)L * (double)B);
  for (int i = 0; i < L; ++i)
    for (int j = 0; j < L; ++j) {
      double acc = 0.0;
      for (int k = 0; k < L; ++k) acc += v[i][k] * inv[k] * v[j][k];
      Pm[i * L + j] = (float)(scale * acc);
    }
}
template <int L>
__global__ void __launch_bounds__(1024) k_kmeans_eig(float* __restrict__ gram_sum, const float* __restrict__ partial, int nblk,
                                                    const float* __restrict__ hyper, int64_t B, float* __restrict__ km_out,
                                                    float* __restrict__ Pm, int row_stride, int tile_stride) {
  if (blockIdx.x != 0) return;
  kmeans_eig_body<L>(gram_sum, partial, nblk, hyper, B, km_out, Pm, row_stride, tile_stride);
}

// ---------------------------------------------------------------------------------------------
// Batch statistics.  Block c < K: sum_b qn[b,c] and sum_b qn[b,c] z[b,:].  Block K: activity,
// pretrain KL and distillation class-weight sums.  One block per quantity => fixed summation order.
// ---------------------------------------------------------------------------------------------
struct StatsArgs {
  const float *qn, *z, *mu, *sv;  // SoA
  const float* tau;               // (B,K) row-major teacher targets or null
  const float* class_weight;      // (K)
  const float* hyper;
  float* stats;                   // [K][3L+1]: sum qn | sum qn z | sum qn mu | sum qn mu^2 ; then [4] scalars
  int K;
  int64_t B, Bp;
};

template <int L>
__device__ __forceinline__ void batch_stats_body(const StatsArgs& A, const int blk) {
  // workgroups 0..K-1: one component each; K: activity + pretrain KL; K+1: distillation class weights;
  // K+2: temporal cohesion (three short dependent chains side by side instead of one long one)
  constexpr int SW = 3 * L + 1;
  __shared__ float s_lt[32][256];
  const int c = blk;
  float vals[SW];
#pragma unroll
  for (int i = 0; i < SW; ++i) vals[i] = 0.0f;
  if (c < A.K) {
#pragma unroll 4
    for (int64_t b = threadIdx.x; b < A.B; b += 256) {
      const float qv = A.qn[(int64_t)c * A.Bp + b];
      vals[0] += qv;
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float m = A.mu[(int64_t)d * A.Bp + b];
        vals[1 + d] = fmaf(qv, A.z[(int64_t)d * A.Bp + b], vals[1 + d]);
        vals[1 + L + d] = fmaf(qv, m, vals[1 + L + d]);
        vals[1 + 2 * L + d] = fmaf(qv * m, m, vals[1 + 2 * L + d]);
      }
    }
  } else if (c == A.K) {
#pragma unroll 4
    for (int64_t b = threadIdx.x; b < A.B; b += 256) {
      float act = 0.0f, kl = 0.0f;
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float s = A.sv[(int64_t)d * A.Bp + b];
        const float m = A.mu[(int64_t)d * A.Bp + b];
        act += fabsf(s);
        const float sc = fminf(fmaxf(s, -4.0f), 2.0f);
        kl += m * m + expf(sc) - 1.0f - sc;
      }
      vals[0] += act;
      vals[1] += 0.5f * kl / L;
    }
  } else if (c == A.K + 1) {
    if (A.tau) {  // w_class_b = sum_c sharpen(tau)[c] * class_weight[c]
      const float Ts = A.hyper[DOF_H_DISTILL_T];
      const bool cache = A.K <= 32;  // the scaled logs of this window through LDS: one logf per entry instead of two
      for (int64_t b = threadIdx.x; b < A.B; b += 256) {
        float mx = -INFINITY;
        for (int k = 0; k < A.K; ++k) {
          const float lt = logf(fmaxf(A.tau[b * A.K + k], 1e-8f)) / Ts;
          if (cache) s_lt[k][threadIdx.x] = lt;
          mx = fmaxf(mx, lt);
        }
        float se = 0.0f, sw = 0.0f;
        for (int k = 0; k < A.K; ++k) {
          const float lt = cache ? s_lt[k][threadIdx.x] : logf(fmaxf(A.tau[b * A.K + k], 1e-8f)) / Ts;
          const float e = expf(lt - mx);
          se += e;
          sw = fmaf(e, A.class_weight[k], sw);
        }
        vals[2] += sw / se;
      }
    }
  } else {
    for (int64_t b = threadIdx.x; b + 1 < A.B; b += 256) {  // temporal cohesion: sum_c |qn[b+1,c] - qn[b,c]|
      float tv = 0.0f;
      for (int k = 0; k < A.K; ++k) tv += fabsf(A.qn[(int64_t)k * A.Bp + b + 1] - A.qn[(int64_t)k * A.Bp + b]);
      vals[3] += tv;
    }
  }
  __shared__ float out[SW];
  dof_block_colsum<SW>(vals, out);
  __syncthreads();
  if (threadIdx.x < SW) {
    if (c < A.K) A.stats[c * SW + threadIdx.x] = out[threadIdx.x];
    else if (c == A.K && threadIdx.x < 2) A.stats[A.K * SW + threadIdx.x] = out[threadIdx.x];
    else if (c > A.K && (int)threadIdx.x == c - A.K + 1) A.stats[A.K * SW + threadIdx.x] = out[threadIdx.x];
  }
}

// ---------------------------------------------------------------------------------------------
// Monte-Carlo KL against the GMM prior (main phase), losses.py:525-545.  One launch: a workgroup owns 8 windows,
// the 32 lanes of a half-wave stride over the S samples of one window.  Per (sample, window): z_s = mu + eps * sd,
// log q(z_s), the K component logits (constants of the mixture staged in LDS once per workgroup: mean, 1 / var
// with the clamped log-variance, log prior - sum_d (log 2 pi + lv) / 2), their log-sum-exp and
// d log p / d z_s.  The sample sums the backward pass needs (sum_s dlogp/dz and sum_s dlogp/dz * eps per window
// and dimension; sum of log q - log p per workgroup) are reduced over the half-wave by a fixed butterfly and
// written directly -- the per-sample gradient tensor never exists.  z_s and the log-sum-exp are kept for
// k_gmm_grads.  (Round 1: k_mckl_fwd + k_block_sum + k_mckl_reduce, 30 us at B = 1024, S = 32, with the
// component constants re-derived through three expf per (sample, window, component, dimension).)
// ---------------------------------------------------------------------------------------------
struct McklArgs {
  const float *mu, *sv;       // [L][Bp]
  const float* eps_mc;        // (S,B,L) row-major
  const float *gmm_means, *gmm_log_vars, *prior;
  const float* hyper;
  float* partial;             // [gridDim.x] per-workgroup sums of log q - log p
  float* lse;                 // [S][Bp]
  float* zs;                  // (S,B,L) row-major, like eps_mc
  float* gsum;                // [2L][Bp]: sum_s dlogp/dz | sum_s dlogp/dz * eps
  int K, S;
  int64_t B, Bp;
};

constexpr int kMcklWindows = 8;  // windows per workgroup

template <int L>
__device__ __forceinline__ float gmm_logp_c(const float* zs, const float* means, const float* log_vars, float lo,
                                            float hi, int c) {
  const float LOG_2PI = 1.8378770664093453f;
  float acc = 0.0f;
#pragma unroll
  for (int d = 0; d < L; ++d) {
    const float lv = fminf(fmaxf(log_vars[c * L + d], lo), hi);
    const float df = zs[d] - means[c * L + d];
    acc += LOG_2PI + lv + df * df * expf(-lv);
  }
  return -0.5f * acc;
}

template <int L>
__device__ __forceinline__ void mckl_body(const McklArgs& A, const int blk) {
  __shared__ float s_m[kLatentMaxKL], s_iv[kLatentMaxKL], s_c[kLatentMaxKL / 4];
  __shared__ float s_term[kMcklWindows];
  const float LOG_2PI = 1.8378770664093453f;
  const float lo = A.hyper[DOF_H_LOGVAR_LO], hi = A.hyper[DOF_H_LOGVAR_HI];
  const int K = A.K;
  const bool staged = K * L <= kLatentMaxKL;
  if (staged) {
    for (int e = threadIdx.x; e < K * L; e += 256) {
      s_m[e] = A.gmm_means[e];
      s_iv[e] = expf(-fminf(fmaxf(A.gmm_log_vars[e], lo), hi));
    }
    for (int c = threadIdx.x; c < K; c += 256) {
      float acc = 0.0f;
      for (int d = 0; d < L; ++d) acc += LOG_2PI + fminf(fmaxf(A.gmm_log_vars[c * L + d], lo), hi);
      s_c[c] = logf(fmaxf(A.prior[c], 1e-8f)) - 0.5f * acc;
    }
    __syncthreads();
  }
  const int bl = (int)threadIdx.x >> 5, sl = (int)threadIdx.x & 31;
  const int64_t b = (int64_t)blk * kMcklWindows + bl;
  const bool live = b < A.B;
  float acc[2 * L + 1];
#pragma unroll
  for (int i = 0; i < 2 * L + 1; ++i) acc[i] = 0.0f;
  if (live) {
    float mu[L], sd[L], sc[L];
#pragma unroll
    for (int d = 0; d < L; ++d) {
      sc[d] = fminf(fmaxf(A.sv[(int64_t)d * A.Bp + b], -4.0f), 2.0f);
      sd[d] = expf(0.5f * sc[d]);
      mu[d] = A.mu[(int64_t)d * A.Bp + b];
    }
    for (int smp = sl; smp < A.S; smp += 32) {
      const int64_t item = (int64_t)smp * A.B + b;
      float e[L], zs[L];
      if constexpr (L % 4 == 0) {
        dof_ld_row<L>(A.eps_mc + item * L, e);
      } else {
#pragma unroll
        for (int d = 0; d < L; ++d) e[d] = A.eps_mc[item * L + d];
      }
      float logq = 0.0f;
#pragma unroll
      for (int d = 0; d < L; ++d) {
        zs[d] = fmaf(e[d], sd[d], mu[d]);
        logq += LOG_2PI + sc[d] + e[d] * e[d];
      }
      logq *= -0.5f;
      if constexpr (L % 4 == 0) {
        dof_st_row<L>(A.zs + item * L, zs);
      } else {
#pragma unroll
        for (int d = 0; d < L; ++d) A.zs[item * L + d] = zs[d];
      }
      auto logit = [&](int c) -> float {
        if (!staged) return logf(fmaxf(A.prior[c], 1e-8f)) + gmm_logp_c<L>(zs, A.gmm_means, A.gmm_log_vars, lo, hi, c);
        float q = 0.0f;
#pragma unroll
        for (int d = 0; d < L; ++d) {
          const float df = zs[d] - s_m[c * L + d];
          q = fmaf(df * df, s_iv[c * L + d], q);
        }
        return fmaf(-0.5f, q, s_c[c]);
      };
      float mx = -INFINITY;
      for (int c = 0; c < K; ++c) mx = fmaxf(mx, logit(c));
      float se = 0.0f;
      float g[L];
#pragma unroll
      for (int d = 0; d < L; ++d) g[d] = 0.0f;
      for (int c = 0; c < K; ++c) {
        const float w = expf(logit(c) - mx);
        se += w;
#pragma unroll
        for (int d = 0; d < L; ++d) {
          const float m = staged ? s_m[c * L + d] : A.gmm_means[c * L + d];
          const float iv = staged ? s_iv[c * L + d] : expf(-fminf(fmaxf(A.gmm_log_vars[c * L + d], lo), hi));
          g[d] = fmaf(w, -(zs[d] - m) * iv, g[d]);
        }
      }
      const float lse = mx + logf(se);
      A.lse[(int64_t)smp * A.Bp + b] = lse;
      acc[0] += logq - lse;
      const float inv = 1.0f / se;
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float gd = g[d] * inv;
        acc[1 + d] += gd;
        acc[1 + L + d] = fmaf(gd, e[d], acc[1 + L + d]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2 * L + 1; ++i) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m);
  }
  if (sl == 0) {
    s_term[bl] = acc[0];
    if (live) {
#pragma unroll
      for (int d = 0; d < 2 * L; ++d) A.gsum[(int64_t)d * A.Bp + b] = acc[1 + d];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < kMcklWindows; ++i) t += s_term[i];
    A.partial[blk] = t;
  }
}
// (E.gram_sum != null: one more workgroup, the last, is the k-means term's eigen-solver -- independent of the statistics, a
// single-thread fp64 Jacobi of ~5 us that used to be a launch of its own)
template <int L>
__global__ void __launch_bounds__(256) k_batch_stats(StatsArgs A, KmeansEigArgs E, int n_stats) {
  if ((int)blockIdx.x < n_stats) batch_stats_body<L>(A, (int)blockIdx.x);
  else if (E.gram_sum) kmeans_eig_body<L>(E.gram_sum, E.partial, E.nblk, E.hyper, E.B, E.km_out, E.Pm, E.row_stride, E.tile_stride);
}
template <int L>
__global__ void __launch_bounds__(256) k_mckl(McklArgs A) {
  mckl_body<L>(A, (int)blockIdx.x);
}
// Both in one launch (they are independent and each is a few dozen workgroups of latency): workgroups [0, n_stats) take the
// batch statistics, the rest the Monte-Carlo KL term.
template <int L>
__global__ void __launch_bounds__(256) k_stats_mckl(StatsArgs SA, McklArgs MA, int n_stats, KmeansEigArgs E, int n_mckl) {
  if ((int)blockIdx.x < n_stats) batch_stats_body<L>(SA, (int)blockIdx.x);
  else if ((int)blockIdx.x < n_stats + n_mckl) mckl_body<L>(MA, (int)blockIdx.x - n_stats);
  else if (E.gram_sum) kmeans_eig_body<L>(E.gram_sum, E.partial, E.nblk, E.hyper, E.B, E.km_out, E.Pm, E.row_stride, E.tile_stride);
}

// ---------------------------------------------------------------------------------------------
// Loss assembly (one thread): every scalar of VadeLoss + the small tensors the backward needs.
// ---------------------------------------------------------------------------------------------
struct LossMidArgs {
  const float* stats;          // from k_batch_stats
  const float* recon_partial;  // [n_recon] per-block sums of -log_prob
  int n_recon;
  const float* mckl_partial;   // [n_mckl] per-block sums of (log q - log p) or null (pretrain)
  int n_mckl;
  const float* km;             // [1] weighted k-means term
  const float* teacher_marginal;  // (K) or null
  const float* hyper;
  float* dqbar;                // [K]  d(nonempty + cat)/d mean_b qn[b,c]
  float* dcen;                 // [K][L] d(repel)/d centroid, pre-divided by pi_b[c]
  float* dscat;                // [K][2L+1] scatter term: d/dP_c | d/dM_cd | d/dS2_cd
  float* scal;                 // [8]: 0 klscale(main: klw*flag/(S*B)), 1 wcls_mean, 2 distill_sum(filled later)
  float* logs;                 // [DOF_LOG_COUNT]
  int K, L, S, T, pretrain;
  int64_t B;
};

// 64-lane sum of a partial array (lanes stride over it, fixed-shape butterfly: deterministic)
__device__ __forceinline__ float dof_wave_sum_array(const float* __restrict__ p, int n) {
  float acc = 0.0f;
  for (int i = (int)(threadIdx.x & 63); i < n; i += 64) acc += p[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  return acc;
}

__device__ __forceinline__ float dof_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__global__ void k_loss_mid(LossMidArgs A) {
  // launched as ONE wavefront.  Lane c owns mixture component c (c, c + 64, ...): the per-component terms and the
  // K x K repulsion run side by side, scalars come from 64-lane butterflies (fixed shape: deterministic).  The form
  // with all of it on lane 0 took 17 us at K = 10 -- as long as the whole posterior backward.
  __shared__ float s_cen[kLatentMaxKL];  // soft centroids [K][L]
  __shared__ float s_mass[kLatentMaxKL / 4];
  const int lane = (int)threadIdx.x;
  float recon = dof_wave_sum_array(A.recon_partial, A.n_recon);
  const float mckl_sum = (!A.pretrain && A.mckl_partial) ? dof_wave_sum_array(A.mckl_partial, A.n_mckl) : 0.0f;
  const int K = A.K, L = A.L;
  const int SW = 3 * L + 1;
  const float* H = A.hyper;
  const float Bf = (float)A.B;
  recon /= (Bf * (float)A.T);
  const float* sc = A.stats + K * SW;
  const float activity = H[DOF_H_L1_ACT] * sc[0] / Bf;
  const float klw = H[DOF_H_KLW];
  float kl, scal0;
  if (A.pretrain) {
    kl = klw * sc[1] / Bf;
    scal0 = 0.0f;
  } else {
    const float raw = mckl_sum / ((float)A.S * Bf);
    kl = klw * fmaxf(raw, 0.0f);
    scal0 = raw > 0.0f ? klw / ((float)A.S * Bf) : 0.0f;
  }
  const bool staged = K * L <= kLatentMaxKL;
  if (staged) {
    for (int c = lane; c < K; c += 64) {
      const float pc = fmaxf(A.stats[c * SW], 1e-8f);
      s_mass[c] = pc;
      for (int d = 0; d < L; ++d) s_cen[c * L + d] = A.stats[c * SW + 1 + d] / pc;
    }
    __syncthreads();
  }
  auto mass = [&](int c) { return staged ? s_mass[c] : fmaxf(A.stats[c * SW], 1e-8f); };
  auto cen = [&](int c, int d) { return staged ? s_cen[c * L + d] : A.stats[c * SW + 1 + d] / fmaxf(A.stats[c * SW], 1e-8f); };
  // nonempty floor on the batch marginal
  float nonempty = 0.0f;
  const float nw = H[DOF_H_NONEMPTY_W], base_floor = H[DOF_H_NONEMPTY_FLOOR], pw = H[DOF_H_NONEMPTY_P];
  const bool has_teacher = A.teacher_marginal && H[DOF_H_HAS_TEACHER] != 0.0f;
  const float wcat = A.pretrain ? 0.0f : H[DOF_H_CAT_W];
  float cat = 0.0f, qs = 0.0f, pbar = 0.0f;
  for (int c = lane; c < K; c += 64) {
    const float qm = A.stats[c * SW] / Bf;
    float fl = base_floor;
    if (has_teacher) fl = fmaxf(fl, 0.9f * A.teacher_marginal[c]);
    const float under = fmaxf(fl - qm, 0.0f);
    float g = 0.0f;
    if (nw > 0.0f && under > 0.0f) {
      nonempty += powf(under, pw);
      g = -nw * pw * powf(under, pw - 1.0f);
    }
    if (wcat > 0.0f) {  // KLDivLoss(batchmean) of log(mean q + 1e-9) vs uniform, divided by K (losses.py:354-359)
      const float u = 1.0f / (float)K;
      cat += u * (logf(u) - logf(qm + 1e-9f));
      g += -wcat * (u / (float)K) / (qm + 1e-9f);
    }
    A.dqbar[c] = g;
    qs += A.stats[c * SW];
    pbar += fmaxf(A.stats[c * SW], 1e-8f);
  }
  nonempty = nw * dof_wave_sum(nonempty);
  cat = dof_wave_sum(cat) * (wcat / (float)K);
  qs = dof_wave_sum(qs);
  pbar = dof_wave_sum(pbar) / (float)K;
  // repulsion between soft centroids (q detached): lane c against every other component, in component order
  float repel = 0.0f;
  const float rw = H[DOF_H_REPEL_W];
  if (rw > 0.0f) {
    const float ls = H[DOF_H_REPEL_LS];
    const float den = fmaxf(1e-9f, 2.0f * ls * ls);
    const float norm = (float)(K * K - K > 1 ? K * K - K : 1);
    float ksum = 0.0f;
    auto rows = [&](auto lp_c) {  // LP = register width of a latent vector (8 up to latent 8, else 16 / 64)
      constexpr int LP = decltype(lp_c)::value;
      for (int c = lane; c < K; c += 64) {
        const float pc = mass(c);
        float acc[LP];
#pragma unroll
        for (int d = 0; d < LP; ++d) acc[d] = 0.0f;
        for (int e = 0; e < K; ++e) {
          if (e == c) continue;
          float d2 = 0.0f, df[LP];
#pragma unroll
          for (int d = 0; d < LP; ++d) {
            df[d] = d < L ? cen(c, d) - cen(e, d) : 0.0f;
            d2 += df[d] * df[d];
          }
          const float kv = expf(-d2 / den);
          ksum += kv;
#pragma unroll
          for (int d = 0; d < LP; ++d) acc[d] += (rw / norm) * 2.0f * kv * (-2.0f * df[d] / den) / pc;
        }
#pragma unroll
        for (int d = 0; d < LP; ++d)
          if (d < L) A.dcen[c * L + d] = acc[d];
      }
    };
    if (L <= 8) rows(std::integral_constant<int, 8>{});
    else if (L <= 16) rows(std::integral_constant<int, 16>{});
    else rows(std::integral_constant<int, 64>{});
    repel = rw * dof_wave_sum(ksum) / norm;
  } else {
    for (int i = lane; i < K * L; i += 64) A.dcen[i] = 0.0f;
  }
  // -(q * log(1/K)).sum(-1).mean(); q rows sum to one after renormalisation
  const float prior_loss = A.pretrain ? 0.0f : logf((float)(K > 1 ? K : 1)) * qs / Bf;
  // ---- optional main-phase regularisers (reference default weight 0): temporal, scatter
  float temporal = 0.0f, scatter = 0.0f;
  const float eta = A.pretrain ? 0.0f : H[DOF_H_SCATTER_W];
  if (!A.pretrain) {
    const float rho = H[DOF_H_TEMPORAL_W];
    if (rho > 0.0f && A.B > 1) temporal = rho * sc[3] / (Bf - 1.0f);
  }
  if (eta > 0.0f) {
    const float beta = H[DOF_H_SCATTER_BETA];
    float wa = 0.0f;  // sum_e w_e A_e
    for (int c = lane; c < K; c += 64) {
      const float pc = fmaxf(A.stats[c * SW], 1e-8f);
      const float w = powf(pc / pbar, -beta);
      float a_c = 0.0f;
      for (int d = 0; d < L; ++d) {
        const float mu = A.stats[c * SW + 1 + L + d] / pc;
        a_c += A.stats[c * SW + 1 + 2 * L + d] / pc - mu * mu;
      }
      wa += w * a_c;
    }
    const float wa_sum = dof_wave_sum(wa);
    const float norm = eta / ((float)K * (float)L);
    scatter = norm * wa_sum;
    for (int c = lane; c < K; c += 64) {
      const float praw = A.stats[c * SW];
      const float pc = fmaxf(praw, 1e-8f);
      const float w = powf(pc / pbar, -beta);
      float a_c = 0.0f, dp = 0.0f;
      for (int d = 0; d < L; ++d) {
        const float m = A.stats[c * SW + 1 + L + d], s2 = A.stats[c * SW + 1 + 2 * L + d];
        const float mu = m / pc;
        a_c += s2 / pc - mu * mu;
        dp += -s2 / (pc * pc) + 2.0f * m * m / (pc * pc * pc);
        A.dscat[c * (2 * L + 1) + 1 + d] = norm * w * (-2.0f * mu / pc);
        A.dscat[c * (2 * L + 1) + 1 + L + d] = norm * w / pc;
      }
      // through the (clamped) cluster mass: scatter itself, its weight w_c, and the mean mass in every w_e
      A.dscat[c * (2 * L + 1)] = praw > 1e-8f ? norm * (w * dp - beta * w * a_c / pc + beta * wa_sum / ((float)K * pbar)) : 0.0f;
    }
  } else {
    for (int i = lane; i < K * (2 * L + 1); i += 64) A.dscat[i] = 0.0f;
  }
  if (lane != 0) return;
  A.scal[0] = scal0;
  A.scal[1] = fmaxf(sc[2] / Bf, 1e-8f);
  float* lg = A.logs;
  lg[DOF_LOG_RECON] = recon;
  lg[DOF_LOG_KL] = kl;
  lg[DOF_LOG_KMEANS] = A.km[0];
  lg[DOF_LOG_ACTIVITY] = activity;
  lg[DOF_LOG_PRIOR] = prior_loss;
  lg[DOF_LOG_NONEMPTY] = nonempty;
  lg[DOF_LOG_REPEL] = repel;
  lg[DOF_LOG_CAT] = cat;
  lg[DOF_LOG_TFCLUST] = 0.0f;  // filled by k_loss_total from the per-sample sums
  lg[DOF_LOG_TEMPORAL] = temporal;
  lg[DOF_LOG_SCATTER] = scatter;
  lg[DOF_LOG_KLW] = klw;
  lg[DOF_LOG_DISTILL] = 0.0f;  // filled by k_loss_total once the per-sample CE sums exist
  lg[DOF_LOG_TOTAL] = recon + kl + nonempty + prior_loss + A.km[0] + activity + repel + cat + temporal + scatter;
}

struct LossTotalArgs {   // (logs == null: "not in this launch")
  const float* distill_partial; const float* tf_partial; int n; const float* hyper; int64_t B; int pretrain; float* logs;
  double* accum;
};
// one wavefront (threads 0 .. 63 of the calling workgroup); lane i owns logs[i] (and its running fp64 sum when the caller keeps one)
__device__ __forceinline__ void loss_total_body(const float* __restrict__ distill_partial, const float* __restrict__ tf_partial, int n,
                                                const float* __restrict__ hyper, int64_t B, int pretrain, float* __restrict__ logs,
                                                double* __restrict__ accum) {
  const float s = dof_wave_sum_array(distill_partial, n);
  const float t = dof_wave_sum_array(tf_partial, n);
  const int i = (int)threadIdx.x;
  if (i >= DOF_LOG_COUNT) return;
  const float d = hyper[DOF_H_LAMBDA_DISTILL] * s / (float)B;
  const float tf = pretrain ? 0.0f : -hyper[DOF_H_TF_W] * t / (float)B;
  float v = logs[i];
  if (i == DOF_LOG_DISTILL) v = d;
  if (i == DOF_LOG_TFCLUST) v = tf;
  if (i == DOF_LOG_TOTAL) v += d + tf;
  if (i == DOF_LOG_DISTILL || i == DOF_LOG_TFCLUST || i == DOF_LOG_TOTAL) logs[i] = v;
  if (accum) accum[i] += (double)v;
}
__global__ void k_loss_total(const float* __restrict__ distill_partial, const float* __restrict__ tf_partial, int n,
                             const float* __restrict__ hyper, int64_t B, int pretrain, float* __restrict__ logs,
                             double* __restrict__ accum) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  loss_total_body(distill_partial, tf_partial, n, hyper, B, pretrain, logs, accum);
}

// ---------------------------------------------------------------------------------------------
// Latent backward, thread = b.
// ---------------------------------------------------------------------------------------------
struct LatentBwdArgs {
  // forward state
  const float *enc, *mu, *pre, *sv, *z, *q, *qn;
  const float* eps;        // (B,L)
  const float* eps_mc;     // (S,B,L) or null
  const float* mckl_gsum;  // [2L][Bp] sample sums of dlogp/dz and dlogp/dz*eps, or null
  const float* dz_dec;     // [2][L][Bp] from the decoder
  const float *wf, *wm, *ws, *gmm_means, *gmm_log_vars;
  const float *Pm, *dcen, *dqbar, *dscat, *scal, *hyper;
  const float* tau;        // (B,K) or null
  const float* class_weight;
  // outputs
  float* dmu_dpre;         // [2L][Bp]
  float* denc;             // [L][Bp]
  float* dlogit;           // [K][Bp]
  float* dflat;            // [J][Bp]
  float* dlogp2;           // [K][Bp] tf_cluster path: grad wrt the clamped-variance component log-likelihoods
  float* distill_partial;  // [nblk]
  float* tf_partial;       // [nblk] per-block sums of sum_c qn*softmax(logp)
  int J, K, S, pretrain;
  int64_t B, Bp;
};

template <int L>
__global__ void __launch_bounds__(256) k_latent_bwd(LatentBwdArgs A) {
  // 1 / sd^2 of every (component, dimension), once per workgroup (was an expf + a division per window and entry)
  __shared__ float s_isd2[kLatentMaxKL];
  constexpr int kTbMax = 32;
  __shared__ float s_tb[kTbMax][256];
  const bool staged = A.K * L <= kLatentMaxKL;
  if (staged) {
    for (int e = threadIdx.x; e < A.K * L; e += 256) {
      const float sd = fmaxf(expf(0.5f * A.gmm_log_vars[e]), 1e-3f);
      s_isd2[e] = 1.0f / (sd * sd);
    }
    __syncthreads();
  }
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < A.B;
  float ce_w = 0.0f, tf_w = 0.0f;
  if (live) {
    const dof_cfp H = dof_cw(A.hyper);
    const dof_cfp wm = dof_cw(A.wm), wsv = dof_cw(A.ws), gmeans = dof_cw(A.gmm_means),
                  glv = dof_cw(A.gmm_log_vars);
    const float Bf = (float)A.B;
    const int K = A.K;
    float z[L], dz[L];
#pragma unroll
    for (int d = 0; d < L; ++d) {
      z[d] = A.z[(int64_t)d * A.Bp + b];
      dz[d] = A.dz_dec[(int64_t)d * A.Bp + b] + A.dz_dec[(int64_t)(L + d) * A.Bp + b];
    }
    // ---- gradient wrt the renormalised posterior qn
    const float lam = H[DOF_H_LAMBDA_DISTILL];
    const bool distill = A.tau && lam > 0.0f;
    float w_total = 1.0f, tmx = 0.0f, tse = 1.0f;
    const float Ts = H[DOF_H_DISTILL_T];
    if (distill) {
      tmx = -INFINITY;
      for (int c = 0; c < K; ++c) tmx = fmaxf(tmx, logf(fmaxf(A.tau[b * K + c], 1e-8f)) / Ts);
      tse = 0.0f;
      float sw = 0.0f, cmax = 0.0f;
      for (int c = 0; c < K; ++c) {
        const float e = expf(logf(fmaxf(A.tau[b * K + c], 1e-8f)) / Ts - tmx);
        tse += e;
        sw = fmaf(e, A.class_weight[c], sw);
        cmax = fmaxf(cmax, e);
      }
      w_total = (sw / tse) / A.scal[1];
      if (H[DOF_H_CONF_W] != 0.0f) {
        const float thr = H[DOF_H_CONF_THR];
        w_total *= fminf(fmaxf((cmax / tse - thr) / fmaxf(1e-6f, 1.0f - thr), 0.0f), 1.0f);
      }
    }
    // ---- optional main-phase terms that touch qn: temporal cohesion, scatter, tf-cluster (default weight 0)
    const float rho = A.pretrain ? 0.0f : H[DOF_H_TEMPORAL_W];
    const float eta = A.pretrain ? 0.0f : H[DOF_H_SCATTER_W];
    const float wtf = A.pretrain ? 0.0f : H[DOF_H_TF_W];
    const float lo = H[DOF_H_LOGVAR_LO], hi = H[DOF_H_LOGVAR_HI];
    float mu_b[L];
#pragma unroll
    for (int d = 0; d < L; ++d) mu_b[d] = A.mu[(int64_t)d * A.Bp + b];
    float pl_mx = -INFINITY, pl_se = 1.0f;
    if (wtf != 0.0f) {  // pl = softmax_c log N(z; m_c, clamp(l_c)) with the 1e-3 std floor (losses.py:547-564)
      for (int c = 0; c < K; ++c) {
        float lg = 0.0f;
#pragma unroll
        for (int d = 0; d < L; ++d) {
          const float sd = fmaxf(expf(0.5f * fminf(fmaxf(glv[c * L + d], lo), hi)), 1e-3f);
          const float u = (z[d] - gmeans[c * L + d]) / sd;
          lg += -0.5f * u * u - logf(sd);
        }
        A.dlogp2[(int64_t)c * A.Bp + b] = lg;
        pl_mx = fmaxf(pl_mx, lg);
      }
      pl_se = 0.0f;
      for (int c = 0; c < K; ++c) pl_se += expf(A.dlogp2[(int64_t)c * A.Bp + b] - pl_mx);
    }
    auto extra_dqn = [&](int c, float qn_c) -> float {
      float g = 0.0f;
      if (rho != 0.0f && A.B > 1) {
        const float sc = rho / ((float)A.B - 1.0f);
        if (b > 0) {
          const float df = qn_c - A.qn[(int64_t)c * A.Bp + b - 1];
          g += sc * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
        }
        if (b + 1 < A.B) {
          const float df = A.qn[(int64_t)c * A.Bp + b + 1] - qn_c;
          g -= sc * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
        }
      }
      if (eta != 0.0f) {
        const float* ds = A.dscat + c * (2 * L + 1);
        g += ds[0];
#pragma unroll
        for (int d = 0; d < L; ++d) g += ds[1 + d] * mu_b[d] + ds[1 + L + d] * mu_b[d] * mu_b[d];
      }
      if (wtf != 0.0f) g += -(wtf / Bf) * expf(A.dlogp2[(int64_t)c * A.Bp + b] - pl_mx) / pl_se;
      return g;
    };
    // sharpened teacher probabilities of this window: computed once (three transcendentals each), kept in LDS for the
    // two passes when K <= 32
    const bool tb_cached = distill && K <= kTbMax;
    if (tb_cached)
      for (int c = 0; c < K; ++c) s_tb[c][threadIdx.x] = expf(logf(fmaxf(A.tau[b * K + c], 1e-8f)) / Ts - tmx) / tse;
    auto teacher_prob = [&](int c) -> float {
      return tb_cached ? s_tb[c][threadIdx.x] : expf(logf(fmaxf(A.tau[b * K + c], 1e-8f)) / Ts - tmx) / tse;
    };
    // pass 1: dot = sum_c dqn[c]*qn[c] ; csum = sum_c max(q,1e-8)
    float dot = 0.0f, csum = 0.0f, ce = 0.0f, tf_sum = 0.0f, pl_dot = 0.0f;
    for (int c = 0; c < K; ++c) {
      const float qn = A.qn[(int64_t)c * A.Bp + b];
      float g = A.dqbar[c] / Bf + extra_dqn(c, qn);
      if (distill) {
        const float tb = teacher_prob(c);
        ce -= tb * logf(fmaxf(qn, 1e-8f));
        if (qn >= 1e-8f) g -= (lam / Bf) * w_total * tb / qn;
      }
      if (wtf != 0.0f) {
        const float pl = expf(A.dlogp2[(int64_t)c * A.Bp + b] - pl_mx) / pl_se;
        tf_sum = fmaf(qn, pl, tf_sum);
        pl_dot = fmaf(-(wtf / Bf) * qn, pl, pl_dot);  // sum_c dpl[c]*pl[c]
      }
      dot = fmaf(g, qn, dot);
      csum += fmaxf(A.q[(int64_t)c * A.Bp + b], 1e-8f);
    }
    ce_w = distill ? w_total * ce : 0.0f;
    tf_w = tf_sum;
    // pass 2: dq through clamp+renormalise, accumulate softmax inner product
    float sdot = 0.0f;
    for (int c = 0; c < K; ++c) {
      const float qn = A.qn[(int64_t)c * A.Bp + b];
      const float q = A.q[(int64_t)c * A.Bp + b];
      float g = A.dqbar[c] / Bf + extra_dqn(c, qn);
      if (distill && qn >= 1e-8f) {
        const float tb = teacher_prob(c);
        g -= (lam / Bf) * w_total * tb / qn;
      }
      const float dq = q >= 1e-8f ? (g - dot) / csum : 0.0f;
      A.dlogit[(int64_t)c * A.Bp + b] = dq;  // temporarily dq
      sdot = fmaf(dq, q, sdot);
    }
    // pass 3: dlogit, posterior path into z, repel path, tf-cluster path
    for (int c = 0; c < K; ++c) {
      const float q = A.q[(int64_t)c * A.Bp + b];
      const float dl = q * (A.dlogit[(int64_t)c * A.Bp + b] - sdot);
      A.dlogit[(int64_t)c * A.Bp + b] = dl;
      const float qn = A.qn[(int64_t)c * A.Bp + b];
      float dl2 = 0.0f;
      if (wtf != 0.0f) {  // softmax backward of pl with d loss / d pl[c] = -(w/B) qn[c]
        const float pl = expf(A.dlogp2[(int64_t)c * A.Bp + b] - pl_mx) / pl_se;
        dl2 = pl * (-(wtf / Bf) * qn - pl_dot);
      }
#pragma unroll
      for (int d = 0; d < L; ++d) {
        float isd2;
        if (staged) {
          isd2 = s_isd2[c * L + d];
        } else {
          const float sd = fmaxf(expf(0.5f * glv[c * L + d]), 1e-3f);
          isd2 = 1.0f / (sd * sd);
        }
        dz[d] = fmaf(dl, -(z[d] - gmeans[c * L + d]) * isd2, dz[d]);
        dz[d] = fmaf(qn, A.dcen[c * L + d], dz[d]);
        if (wtf != 0.0f) {
          const float sd2 = fmaxf(expf(0.5f * fminf(fmaxf(glv[c * L + d], lo), hi)), 1e-3f);
          dz[d] = fmaf(dl2, -(z[d] - gmeans[c * L + d]) / (sd2 * sd2), dz[d]);
        }
      }
      if (wtf != 0.0f) A.dlogp2[(int64_t)c * A.Bp + b] = dl2;  // consumed by k_gmm_grads
    }
    // k-means (Gram spectrum): dZ = Z * Pm
#pragma unroll
    for (int d = 0; d < L; ++d) {
      float acc = 0.0f;
#pragma unroll
      for (int e = 0; e < L; ++e) acc = fmaf(z[e], A.Pm[e * L + d], acc);
      dz[d] += acc;
    }
    // ---- through the reparameterisation, activity L1, KL
    const float klw = H[DOF_H_KLW];
    const float act = H[DOF_H_L1_ACT] / Bf;
    float dmu[L], dpre[L];
#pragma unroll
    for (int d = 0; d < L; ++d) {
      const float s = A.sv[(int64_t)d * A.Bp + b];
      const float m = A.mu[(int64_t)d * A.Bp + b];
      float dm = dz[d];
      if (eta != 0.0f) {  // scatter term acts on z_mean directly: sum_c qn (dM_cd + 2 dS2_cd mu)
        for (int c = 0; c < K; ++c) {
          const float* dsc = A.dscat + c * (2 * L + 1);
          dm = fmaf(A.qn[(int64_t)c * A.Bp + b], dsc[1 + d] + 2.0f * dsc[1 + L + d] * m, dm);
        }
      }
      float ds = dz[d] * A.eps[b * L + d] * 0.5f * expf(0.5f * s);
      ds += s > 0.0f ? act : (s < 0.0f ? -act : 0.0f);
      const float sc = fminf(fmaxf(s, -4.0f), 2.0f);
      const bool pass = (s >= -4.0f) && (s <= 2.0f);
      if (A.pretrain) {
        dm = fmaf(klw / (Bf * L), m, dm);
        if (pass) ds = fmaf(klw / (Bf * L), 0.5f * (expf(sc) - 1.0f), ds);
      } else {
        const float ksc = A.scal[0];
        if (ksc != 0.0f) {
          const float gz = A.mckl_gsum[(int64_t)d * A.Bp + b];
          const float gze = A.mckl_gsum[(int64_t)(L + d) * A.Bp + b];
          dm = fmaf(-ksc, gz, dm);
          if (pass) ds += ksc * (-gze * 0.5f * expf(0.5f * sc) - 0.5f * (float)A.S);
        }
      }
      dmu[d] = dm;
      dpre[d] = ds * dof_sigmoid(A.pre[(int64_t)d * A.Bp + b]);
      A.dmu_dpre[(int64_t)d * A.Bp + b] = dm;
      A.dmu_dpre[(int64_t)(L + d) * A.Bp + b] = dpre[d];
    }
    float denc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      float acc = 0.0f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        acc = fmaf(wm[l * L + k], dmu[l], acc);
        acc = fmaf(wsv[l * L + k], dpre[l], acc);
      }
      denc[k] = acc;
      A.denc[(int64_t)k * A.Bp + b] = acc;
    }
  }
  float v2[2] = {ce_w, tf_w};
  float out2[2] = {0.0f, 0.0f};
  __shared__ float o2[2];
  dof_block_colsum<2>(v2, o2);
  __syncthreads();
  (void)out2;
  if (threadIdx.x == 0) {
    A.distill_partial[blockIdx.x] = o2[0];
    A.tf_partial[blockIdx.x] = o2[1];
  }
}

// ---------------------------------------------------------------------------------------------
// Latent backward with the components across lanes: 16 lanes per window, lane j owns components j, j + 16, ...
// (NC per lane, K <= 16 NC), sums over the components are 4-step butterflies inside the 16-lane row, the
// per-window tail (Gram spectrum, reparameterisation, the two dense layers) runs replicated in the row.  Same
// arithmetic as k_latent_bwd (which stays for K > 32); at K = 10 the dependent chain per thread shrinks from ~3 K
// component visits to 3.  Also emits the data gradient of encoder.final_dense (dflat) when asked to.
// ---------------------------------------------------------------------------------------------

template <int L, int NC>
__global__ void __launch_bounds__(256) k_latent_bwd_w(LatentBwdArgs A) {
  constexpr int KMAX = 16 * NC;
  __shared__ float s_isd2[KMAX * L], s_gm[KMAX * L], s_dcen[KMAX * L];
  __shared__ float s_wm[L * L], s_ws[L * L], s_pm[L * L];
  __shared__ float s_wf[kWfMax];  // [input][l]
  const int K = A.K;
  const bool wf_lds = A.dflat && A.J * L <= kWfMax;
  for (int e = threadIdx.x; e < K * L; e += 256) {
    const float sd = fmaxf(expf(0.5f * A.gmm_log_vars[e]), 1e-3f);
    s_isd2[e] = 1.0f / (sd * sd);
    s_gm[e] = A.gmm_means[e];
    s_dcen[e] = A.dcen[e];
  }
  if (threadIdx.x < L * L) {
    s_wm[threadIdx.x] = A.wm[threadIdx.x];
    s_ws[threadIdx.x] = A.ws[threadIdx.x];
    s_pm[threadIdx.x] = A.Pm[threadIdx.x];
  }
  if (wf_lds)
    for (int e = threadIdx.x; e < A.J * L; e += 256) {
      const int l = e / A.J;
      s_wf[(e - l * A.J) * L + l] = A.wf[e];
    }
  __syncthreads();
  const int j = (int)threadIdx.x & 15, row = (int)threadIdx.x >> 4;
  const int l = j % L;  // the latent dimension this lane owns in the per-window tail
  const int64_t b_raw = (int64_t)blockIdx.x * kLatRows + row;
  const bool live = b_raw < A.B;
  const int64_t b = live ? b_raw : A.B - 1;  // idle rows shadow the last window (DPP reads every lane)
  const float* H = A.hyper;
  const float Bf = (float)A.B;
  const float rho = A.pretrain ? 0.0f : H[DOF_H_TEMPORAL_W];
  const float eta = A.pretrain ? 0.0f : H[DOF_H_SCATTER_W];
  const float wtf = A.pretrain ? 0.0f : H[DOF_H_TF_W];
  const float lo = H[DOF_H_LOGVAR_LO], hi = H[DOF_H_LOGVAR_HI];
  const float z_l = A.z[(int64_t)l * A.Bp + b];
  const float mu_l = A.mu[(int64_t)l * A.Bp + b];
  float z[L], mu_b[L];
  dof_static_for<L>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    z[d] = dof_gbcast<d, 16>(z_l);
    mu_b[d] = eta != 0.0f ? dof_gbcast<d, 16>(mu_l) : 0.0f;  // only the scatter term needs every dimension's mean
  });
  int cc[NC];
  bool valid[NC];
  float q[NC], qn[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    valid[i] = j + 16 * i < K;
    cc[i] = valid[i] ? j + 16 * i : 0;
    q[i] = A.q[(int64_t)cc[i] * A.Bp + b];
    qn[i] = A.qn[(int64_t)cc[i] * A.Bp + b];
  }
  // ---- sharpened teacher probabilities and the window's distillation weight
  const float lam = H[DOF_H_LAMBDA_DISTILL];
  const bool distill = A.tau && lam > 0.0f;
  float w_total = 1.0f;
  float tb[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) tb[i] = 0.0f;
  if (distill) {
    const float Ts = H[DOF_H_DISTILL_T];
    float lt[NC], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      lt[i] = valid[i] ? logf(fmaxf(A.tau[b * K + cc[i]], 1e-8f)) / Ts : -INFINITY;
      mx = fmaxf(mx, lt[i]);
    }
    mx = dof_row16_max(mx);
    float se = 0.0f, sw = 0.0f, cmax = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      tb[i] = valid[i] ? expf(lt[i] - mx) : 0.0f;
      se += tb[i];
      sw = fmaf(tb[i], valid[i] ? A.class_weight[cc[i]] : 0.0f, sw);
      cmax = fmaxf(cmax, tb[i]);
    }
    se = dof_row16_sum(se);
    sw = dof_row16_sum(sw);
    cmax = dof_row16_max(cmax);
#pragma unroll
    for (int i = 0; i < NC; ++i) tb[i] /= se;
    w_total = (sw / se) / A.scal[1];
    if (H[DOF_H_CONF_W] != 0.0f) {
      const float thr = H[DOF_H_CONF_THR];
      w_total *= fminf(fmaxf((cmax / se - thr) / fmaxf(1e-6f, 1.0f - thr), 0.0f), 1.0f);
    }
  }
  // ---- optional main-phase terms that touch qn: temporal cohesion, scatter, tf-cluster (default weight 0)
  float pl[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) pl[i] = 0.0f;
  if (wtf != 0.0f) {  // pl = softmax_c log N(z; m_c, clamp(l_c)) with the 1e-3 std floor (losses.py:547-564)
    float lg[NC], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      lg[i] = 0.0f;
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float sd = fmaxf(expf(0.5f * fminf(fmaxf(A.gmm_log_vars[cc[i] * L + d], lo), hi)), 1e-3f);
        const float u = (z[d] - s_gm[cc[i] * L + d]) / sd;
        lg[i] += -0.5f * u * u - logf(sd);
      }
      if (!valid[i]) lg[i] = -INFINITY;
      mx = fmaxf(mx, lg[i]);
    }
    mx = dof_row16_max(mx);
    float se = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      pl[i] = valid[i] ? expf(lg[i] - mx) : 0.0f;
      se += pl[i];
    }
    se = dof_row16_sum(se);
#pragma unroll
    for (int i = 0; i < NC; ++i) pl[i] /= se;
  }
  // ---- d loss / d qn per component, then through clamp + renormalise and the softmax
  float g[NC];
  float dot = 0.0f, csum = 0.0f, ce = 0.0f, tf_sum = 0.0f, pl_dot = 0.0f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = cc[i];
    float gi = A.dqbar[c] / Bf;
    if (rho != 0.0f && A.B > 1) {
      const float sc = rho / ((float)A.B - 1.0f);
      if (b > 0) {
        const float df = qn[i] - A.qn[(int64_t)c * A.Bp + b - 1];
        gi += sc * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
      }
      if (b + 1 < A.B) {
        const float df = A.qn[(int64_t)c * A.Bp + b + 1] - qn[i];
        gi -= sc * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
      }
    }
    if (eta != 0.0f) {
      const float* ds = A.dscat + c * (2 * L + 1);
      gi += ds[0];
#pragma unroll
      for (int d = 0; d < L; ++d) gi += ds[1 + d] * mu_b[d] + ds[1 + L + d] * mu_b[d] * mu_b[d];
    }
    if (wtf != 0.0f) gi += -(wtf / Bf) * pl[i];
    if (distill) {
      if (valid[i]) ce -= tb[i] * logf(fmaxf(qn[i], 1e-8f));
      if (qn[i] >= 1e-8f) gi -= (lam / Bf) * w_total * tb[i] / qn[i];
    }
    if (!valid[i]) gi = 0.0f;
    g[i] = gi;
    if (valid[i]) {
      tf_sum = fmaf(qn[i], pl[i], tf_sum);
      pl_dot = fmaf(-(wtf / Bf) * qn[i], pl[i], pl_dot);  // sum_c dpl[c]*pl[c]
      dot = fmaf(gi, qn[i], dot);
      csum += fmaxf(q[i], 1e-8f);
    }
  }
  dot = dof_row16_sum(dot);
  csum = dof_row16_sum(csum);
  if (distill) ce = dof_row16_sum(ce);
  if (wtf != 0.0f) {
    tf_sum = dof_row16_sum(tf_sum);
    pl_dot = dof_row16_sum(pl_dot);
  }
  float dq[NC], sdot = 0.0f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    dq[i] = (valid[i] && q[i] >= 1e-8f) ? (g[i] - dot) / csum : 0.0f;
    sdot = fmaf(dq[i], valid[i] ? q[i] : 0.0f, sdot);
  }
  sdot = dof_row16_sum(sdot);
  float part[L];
#pragma unroll
  for (int d = 0; d < L; ++d) part[d] = 0.0f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = cc[i];
    const float dl = valid[i] ? q[i] * (dq[i] - sdot) : 0.0f;
    const float qv = valid[i] ? qn[i] : 0.0f;
    float dl2 = 0.0f;
    if (wtf != 0.0f && valid[i]) dl2 = pl[i] * (-(wtf / Bf) * qn[i] - pl_dot);  // softmax backward of pl
    if (valid[i] && live) {
      A.dlogit[(int64_t)c * A.Bp + b] = dl;
      if (wtf != 0.0f) A.dlogp2[(int64_t)c * A.Bp + b] = dl2;  // consumed by k_gmm_grads
    }
#pragma unroll
    for (int d = 0; d < L; ++d) {
      const float df = z[d] - s_gm[c * L + d];
      part[d] = fmaf(dl, -df * s_isd2[c * L + d], part[d]);
      part[d] = fmaf(qv, s_dcen[c * L + d], part[d]);
      if (wtf != 0.0f) {
        const float sd2 = fmaxf(expf(0.5f * fminf(fmaxf(A.gmm_log_vars[c * L + d], lo), hi)), 1e-3f);
        part[d] = fmaf(dl2, -df / (sd2 * sd2), part[d]);
      }
    }
  }
  // the lane's own dimension from here on: decoder gradient + the row sums of the component paths
  float dz_l = A.dz_dec[(int64_t)l * A.Bp + b] + A.dz_dec[(int64_t)(L + l) * A.Bp + b];
#pragma unroll
  for (int d = 0; d < L; ++d) {
    const float t = dof_row16_sum(part[d]);
    if (d == l) dz_l += t;
  }
  // k-means (Gram spectrum): dZ = Z * Pm
  {
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < L; ++e) acc = fmaf(z[e], s_pm[e * L + l], acc);
    dz_l += acc;
  }
  // ---- through the reparameterisation, activity L1, KL
  const float klw = H[DOF_H_KLW];
  const float act = H[DOF_H_L1_ACT] / Bf;
  const float s = A.sv[(int64_t)l * A.Bp + b];
  float dm = dz_l;
  if (eta != 0.0f) {  // scatter term acts on z_mean directly: sum_c qn (dM_cd + 2 dS2_cd mu)
#pragma unroll
    for (int d = 0; d < L; ++d) {
      float sc_part = 0.0f;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const float* dsc = A.dscat + cc[i] * (2 * L + 1);
        if (valid[i]) sc_part = fmaf(qn[i], dsc[1 + d] + 2.0f * dsc[1 + L + d] * mu_b[d], sc_part);
      }
      const float t = dof_row16_sum(sc_part);
      if (d == l) dm += t;
    }
  }
  float ds = dz_l * A.eps[b * L + l] * 0.5f * expf(0.5f * s);
  ds += s > 0.0f ? act : (s < 0.0f ? -act : 0.0f);
  const float sc = fminf(fmaxf(s, -4.0f), 2.0f);
  const bool pass = (s >= -4.0f) && (s <= 2.0f);
  if (A.pretrain) {
    dm = fmaf(klw / (Bf * L), mu_l, dm);
    if (pass) ds = fmaf(klw / (Bf * L), 0.5f * (expf(sc) - 1.0f), ds);
  } else {
    const float ksc = A.scal[0];
    if (ksc != 0.0f) {
      const float gz = A.mckl_gsum[(int64_t)l * A.Bp + b];
      const float gze = A.mckl_gsum[(int64_t)(L + l) * A.Bp + b];
      dm = fmaf(-ksc, gz, dm);
      if (pass) ds += ksc * (-gze * 0.5f * expf(0.5f * sc) - 0.5f * (float)A.S);
    }
  }
  const float dpre_l = ds * dof_sigmoid(A.pre[(int64_t)l * A.Bp + b]);
  if (live && j < L) {
    A.dmu_dpre[(int64_t)l * A.Bp + b] = dm;
    A.dmu_dpre[(int64_t)(L + l) * A.Bp + b] = dpre_l;
  }
  // encoder_mean / encoder_log_var data gradient: denc[k] = sum_l' wm[l', k] dmu[l'] + ws[l', k] dpre[l'], lane owns k = l
  float denc_l = 0.0f;
  dof_static_for<L>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    denc_l = fmaf(s_wm[d * L + l], dof_gbcast<d, 16>(dm), denc_l);
    denc_l = fmaf(s_ws[d * L + l], dof_gbcast<d, 16>(dpre_l), denc_l);
  });
  if (live && j < L) A.denc[(int64_t)l * A.Bp + b] = denc_l;
  if (A.dflat) {  // encoder.final_dense data gradient: dflat[jj] = sum_l wf[l, jj] denc[l]; lane owns every 16th input
    float denc[L];
    dof_static_for<L>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      denc[d] = dof_gbcast<d, 16>(denc_l);
    });
    if (live)
      for (int jj = j; jj < A.J; jj += 16) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < L; ++k) acc = fmaf(wf_lds ? s_wf[jj * L + k] : A.wf[k * A.J + jj], denc[k], acc);
        A.dflat[(int64_t)jj * A.Bp + b] = acc;
      }
  }
  float v2[2] = {(live && j == 0 && distill) ? w_total * ce : 0.0f, (live && j == 0) ? tf_sum : 0.0f};
  __shared__ float o2[2];
  dof_block_colsum<2>(v2, o2);
  __syncthreads();
  if (threadIdx.x == 0) {
    A.distill_partial[blockIdx.x] = o2[0];
    A.tf_partial[blockIdx.x] = o2[1];
  }
}

// ---------------------------------------------------------------------------------------------
// GMM parameter gradients.  Workgroup (c, y): posterior path (sum over b) + MC-KL path (sum over (s,b)) of component c
// over its share of the windows / samples; everything that depends on the component only (1 / sd, 1 / var with the
// clamped log-variance, the additive constant of its logit) is derived once per thread.  The MC path reads the z_s
// and log-sum-exp that k_mckl kept.  Per-workgroup partial sums -> k_sum_partials.
// ---------------------------------------------------------------------------------------------
struct GmmGradArgs {
  const float *z, *dlogit;           // [L][Bp], [K][Bp]
  const float* dlogp2;               // [K][Bp] tf-cluster path (clamped log-variances) or unused
  const float *zs, *lse;             // (S,B,L), [S][Bp] from k_mckl (main phase) or null
  const float *gmm_means, *gmm_log_vars, *prior, *scal, *hyper;
  float* partial;                    // [gridDim.y][2*K*L]: means (K,L) then log-vars (K,L)
  int K, S, pretrain;
  int64_t B, Bp;
};

// (LT.logs != null: workgroup (K, 0) -- one past the components -- finishes the step's logged totals, k_loss_total's
// one-wavefront job: both wait for k_latent_bwd only)
template <int L>
__global__ void __launch_bounds__(256) k_gmm_grads(GmmGradArgs A, LossTotalArgs LT) {
  if ((int)blockIdx.x >= A.K) {
    if (LT.logs && blockIdx.y == 0 && threadIdx.x < 64)
      loss_total_body(LT.distill_partial, LT.tf_partial, LT.n, LT.hyper, LT.B, LT.pretrain, LT.logs, LT.accum);
    return;
  }
  const int c = blockIdx.x;
  float vals[2 * L];
#pragma unroll
  for (int i = 0; i < 2 * L; ++i) vals[i] = 0.0f;
  float m[L], lvraw[L];
#pragma unroll
  for (int d = 0; d < L; ++d) {
    m[d] = A.gmm_means[c * L + d];
    lvraw[d] = A.gmm_log_vars[c * L + d];
  }
  const int64_t tstride = (int64_t)gridDim.y * 256;
  const int64_t t0 = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const bool use_tf = !A.pretrain && A.hyper[DOF_H_TF_W] != 0.0f;
  const float tlo = A.hyper[DOF_H_LOGVAR_LO], thi = A.hyper[DOF_H_LOGVAR_HI];
  if (t0 < A.B) {
    float isd[L];
    bool free_sd[L];  // the 1e-3 floor on the standard deviation is inactive
#pragma unroll
    for (int d = 0; d < L; ++d) {
      const float e = expf(0.5f * lvraw[d]);
      isd[d] = 1.0f / fmaxf(e, 1e-3f);
      free_sd[d] = e >= 1e-3f;
    }
    for (int64_t b = t0; b < A.B; b += tstride) {
      const float dl = A.dlogit[(int64_t)c * A.Bp + b];
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float u = (A.z[(int64_t)d * A.Bp + b] - m[d]) * isd[d];
        vals[d] = fmaf(dl, u * isd[d], vals[d]);
        // d/d log_var of [-u^2/2 - log sd] = (u^2 - 1) * 0.5, only while the 1e-3 floor is inactive
        if (free_sd[d]) vals[L + d] = fmaf(dl, 0.5f * (u * u - 1.0f), vals[L + d]);
      }
    }
    if (use_tf) {
      bool free_lv[L];
#pragma unroll
      for (int d = 0; d < L; ++d) {
        const float e = expf(0.5f * fminf(fmaxf(lvraw[d], tlo), thi));
        isd[d] = 1.0f / fmaxf(e, 1e-3f);
        free_lv[d] = e >= 1e-3f && lvraw[d] >= tlo && lvraw[d] <= thi;
      }
      for (int64_t b = t0; b < A.B; b += tstride) {
        const float dl2 = A.dlogp2[(int64_t)c * A.Bp + b];
#pragma unroll
        for (int d = 0; d < L; ++d) {
          const float u = (A.z[(int64_t)d * A.Bp + b] - m[d]) * isd[d];
          vals[d] = fmaf(dl2, u * isd[d], vals[d]);
          if (free_lv[d]) vals[L + d] = fmaf(dl2, 0.5f * (u * u - 1.0f), vals[L + d]);
        }
      }
    }
  }
  const float ksc = A.pretrain ? 0.0f : A.scal[0];
  if (ksc != 0.0f) {
    const float LOG_2PI = 1.8378770664093453f;
    float iv[L];
    bool free_lv[L];
    float cst = 0.0f;
#pragma unroll
    for (int d = 0; d < L; ++d) {
      const float lv = fminf(fmaxf(lvraw[d], tlo), thi);
      iv[d] = expf(-lv);
      free_lv[d] = lvraw[d] >= tlo && lvraw[d] <= thi;
      cst += LOG_2PI + lv;
    }
    cst = logf(fmaxf(A.prior[c], 1e-8f)) - 0.5f * cst;
    const int64_t n = (int64_t)A.S * A.B;
    for (int64_t i = t0; i < n; i += tstride) {
      const int smp = (int)(i / A.B);
      const int64_t b = i - (int64_t)smp * A.B;
      float zs[L];
      if constexpr (L % 4 == 0) {
        dof_ld_row<L>(A.zs + i * L, zs);
      } else {
#pragma unroll
        for (int d = 0; d < L; ++d) zs[d] = A.zs[i * L + d];
      }
      float q = 0.0f, df[L];
#pragma unroll
      for (int d = 0; d < L; ++d) {
        df[d] = zs[d] - m[d];
        q = fmaf(df[d] * df[d], iv[d], q);
      }
      // responsibility of component c for this sample; the loss has -log p: gradient = -ksc * r * d(log N)/d(param)
      const float w = -ksc * expf(fmaf(-0.5f, q, cst) - A.lse[(int64_t)smp * A.Bp + b]);
#pragma unroll
      for (int d = 0; d < L; ++d) {
        vals[d] = fmaf(w, df[d] * iv[d], vals[d]);
        if (free_lv[d]) vals[L + d] = fmaf(w, 0.5f * (df[d] * df[d] * iv[d] - 1.0f), vals[L + d]);
      }
    }
  }
  __shared__ float out[2 * L];
  dof_block_colsum<2 * L>(vals, out);
  __syncthreads();
  float* dst = A.partial + (int64_t)blockIdx.y * 2 * A.K * L;
  if (threadIdx.x < L) dst[c * L + threadIdx.x] = out[threadIdx.x];
  else if (threadIdx.x < 2 * L) dst[A.K * L + c * L + threadIdx.x - L] = out[threadIdx.x];
}

}  // namespace
