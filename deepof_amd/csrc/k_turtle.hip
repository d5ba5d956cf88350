// TURTLE teacher: soft cluster targets tau* that every view can predict linearly.
//
// Reference semantics restated here (/root/reference/deepof/clustering/teacher_model.py):
//   * TaskEncoder :112-149      tau = softmax_k( mean_v (W_v f_v + c_v) / T_task )
//   * TurtleHeads :43-109       per view: M plain-SGD steps (lr, weight decay on weights AND biases) of
//                               soft-CE( (H_v normalize(f_v) + h_v) / T_head , tau )         (tau detached)
//   * TurtleTeacher.fit :240-350  loss(tau) = mean_v soft-CE(head_v logits [detached], tau) + alpha E_b[H(tau_b)]
//                               + gamma_t relu(log K - H(mean_b tau)) + delta_t sum_k relu(floor - mean_b tau_k^2)/(floor K)
//                               [+ rho mean_b |tau_{b+1} - tau_b|_1 on odd steps];  Adam(lr_theta) on (W, c)
// The problem is tiny and dense (B ~ 2048 rows, d <= 64 features, K ~ 10 clusters) but the reference spends
// 100 optimiser steps x views x ~6 framework ops per OUTER step on it.  Here an inner step is two short launches
// over 256-row tiles (all views at once, features resident in L2), an outer step 4 more.
#include <cmath>
#include <cstring>

#include "dof_rt.h"
#include "deepof_hip.h"

namespace {

struct TtView {
  const float* f;   // (B, d) features of this view
  float* fn;        // (B, d) row-normalised copy (workspace)
  int d;
  int64_t task_w, task_b, head_w, head_b;  // float offsets into the parameter buffer
};

struct TtArgs {
  TtView v[DOF_TURTLE_MAX_VIEWS];
  int V, K, B;
  float* params;
  float *adam_m, *adam_v;
  float* tau;       // (B, K)
  float* dtau;      // (B, K)
  float* dlogit;    // (B, K)
  float* partial;   // (nblk, 2K + 4)
  float* inner_partial;  // (views, nblk, maxel) tile partials of the inner-step gradient
  int maxel;
  float* scal;      // cm[K], cu[K], misc
  float* logs;
  int nblk;
  // hyper
  float gamma_t, alpha, delta_t, head_temp, task_temp, inner_lr, head_wd, lr_theta, rho, bc1, bc2;
  int inner_steps, normalize, smooth;
};

constexpr int TT_MAXK = 64;

// tau of rows [0, n) + (when fn != null) the row-normalised features the heads see
__global__ void __launch_bounds__(256) k_tt_prepare(TtArgs A, int64_t n_rows, float* __restrict__ tau_out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_rows) return;
  float lg[TT_MAXK];
  for (int k = 0; k < A.K; ++k) lg[k] = 0.0f;
  for (int v = 0; v < A.V; ++v) {
    const TtView& w = A.v[v];
    const float* __restrict__ f = w.f + b * w.d;
    float n2 = 0.0f;
    for (int i = 0; i < w.d; ++i) n2 = fmaf(f[i], f[i], n2);
    for (int k = 0; k < A.K; ++k) {
      float acc = A.params[w.task_b + k];
      const float* __restrict__ wr = A.params + w.task_w + (int64_t)k * w.d;
      for (int i = 0; i < w.d; ++i) acc = fmaf(wr[i], f[i], acc);
      lg[k] += acc / A.task_temp;
    }
    if (w.fn) {
      const float inv = A.normalize ? 1.0f / fmaxf(sqrtf(n2), 1e-12f) : 1.0f;
      for (int i = 0; i < w.d; ++i) w.fn[b * w.d + i] = f[i] * inv;
    }
  }
  const float invV = 1.0f / (float)(A.V > 0 ? A.V : 1);
  float mx = -INFINITY;
  for (int k = 0; k < A.K; ++k) {
    lg[k] *= invV;
    mx = fmaxf(mx, lg[k]);
  }
  float se = 0.0f;
  for (int k = 0; k < A.K; ++k) {
    lg[k] = expf(lg[k] - mx);
    se += lg[k];
  }
  for (int k = 0; k < A.K; ++k) tau_out[b * A.K + k] = lg[k] / se;
}

// log-softmax of one head for one row; returns through lp[]
__device__ __forceinline__ void tt_head_logp(const float* __restrict__ H, const float* __restrict__ h,
                                             const float* __restrict__ fn, int d, int K, float inv_temp, float* lp) {
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) {
    float acc = h[k];
    const float* __restrict__ wr = H + (int64_t)k * d;
    for (int i = 0; i < d; ++i) acc = fmaf(wr[i], fn[i], acc);
    lp[k] = acc * inv_temp;
    mx = fmaxf(mx, lp[k]);
  }
  float se = 0.0f;
  for (int k = 0; k < K; ++k) se += expf(lp[k] - mx);
  const float lse = mx + logf(se);
  for (int k = 0; k < K; ++k) lp[k] -= lse;
}

// One inner SGD step of every head = two launches:
//   k_tt_inner_grad  (grid: views x 256-row tiles)  the tile's normalised features, tau rows and the head's weights
//                    go to LDS; a thread per row forms the soft-CE coefficients (softmax * sum tau - tau) / (T B),
//                    then a thread per weight element reduces the tile -> partial[view][tile][element]
//   k_tt_inner_step  (grid: views)  sums the tile partials in fixed order and applies SGD with weight decay.
// The views' features (B x d, <= 256 KB) stay in L2 across the M steps; 2 M short launches per outer step replace
// the reference's M x views x ~6 framework ops, and every sum has a fixed order (run-to-run reproducible).
constexpr int TT_TILE = 256;
constexpr int TT_MAXW = 8192;  // K * (d + 1) floats of one head

__global__ void __launch_bounds__(TT_TILE) k_tt_inner_grad(TtArgs A, float* __restrict__ partial, int ntile, int maxel) {
  __shared__ float Hs[TT_MAXW];
  __shared__ float cf[TT_TILE][TT_MAXK + 1];
  const TtView w = A.v[blockIdx.x];
  const int tile = blockIdx.y, K = A.K, d = w.d, B = A.B, tid = threadIdx.x;
  const int nel = K * (d + 1);
  for (int e = tid; e < nel; e += TT_TILE) {
    const int k = e / (d + 1), i = e - k * (d + 1);
    Hs[e] = i < d ? A.params[w.head_w + (int64_t)k * d + i] : A.params[w.head_b + k];
  }
  __syncthreads();
  const int b = tile * TT_TILE + tid;
  const float inv_temp = 1.0f / A.head_temp;
  if (b < B) {
    float lp[TT_MAXK];
    const float* __restrict__ fn = w.fn + (int64_t)b * d;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float acc = Hs[k * (d + 1) + d];
      const float* wr = Hs + k * (d + 1);
      for (int i = 0; i < d; ++i) acc = fmaf(wr[i], fn[i], acc);
      lp[k] = acc * inv_temp;
      mx = fmaxf(mx, lp[k]);
    }
    float se = 0.0f, st = 0.0f;
    for (int k = 0; k < K; ++k) {
      lp[k] = expf(lp[k] - mx);
      se += lp[k];
      st += fminf(fmaxf(A.tau[(int64_t)b * K + k], 1e-8f), 1.0f);
    }
    const float sc = inv_temp / (float)B;
    for (int k = 0; k < K; ++k) {
      const float tk = fminf(fmaxf(A.tau[(int64_t)b * K + k], 1e-8f), 1.0f);
      cf[tid][k] = (lp[k] / se * st - tk) * sc;
    }
  } else {
    for (int k = 0; k < K; ++k) cf[tid][k] = 0.0f;
  }
  __syncthreads();
  const int r0 = tile * TT_TILE, nr = (B - r0) < TT_TILE ? (B - r0) : TT_TILE;
  float* __restrict__ pout = partial + ((int64_t)blockIdx.x * ntile + tile) * maxel;
  for (int e = tid; e < nel; e += TT_TILE) {
    const int k = e / (d + 1), i = e - k * (d + 1);
    float acc = 0.0f;
    if (i < d) {
      const float* __restrict__ col = w.fn + (int64_t)r0 * d + i;
      for (int r = 0; r < nr; ++r) acc = fmaf(cf[r][k], col[(int64_t)r * d], acc);
    } else {
      for (int r = 0; r < nr; ++r) acc += cf[r][k];
    }
    pout[e] = acc;
  }
}

__global__ void __launch_bounds__(256) k_tt_inner_step(TtArgs A, const float* __restrict__ partial, int ntile, int maxel) {
  const TtView w = A.v[blockIdx.x];
  const int K = A.K, d = w.d, nel = K * (d + 1);
  for (int e = threadIdx.x; e < nel; e += 256) {
    float g = 0.0f;
    for (int t = 0; t < ntile; ++t) g += partial[((int64_t)blockIdx.x * ntile + t) * maxel + e];
    const int k = e / (d + 1), i = e - k * (d + 1);
    float* p = i < d ? A.params + w.head_w + (int64_t)k * d + i : A.params + w.head_b + k;
    *p -= A.inner_lr * (g + A.head_wd * *p);
  }
}

// Per-row part of the outer loss and of d loss / d tau; block sums of the batch-level statistics.
// partial row: [0,K) sum tau, [K,2K) sum clamp(tau,1e-8)^2, 2K: sum CE, 2K+1: sum H(tau_b), 2K+2: sum |tau_{b+1}-tau_b|_1
__global__ void __launch_bounds__(256) k_tt_rows(TtArgs A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = A.K, B = A.B;
  float tau[TT_MAXK], dt[TT_MAXK];
  float ce = 0.0f, hs = 0.0f, sm = 0.0f;
  const bool live = b < B;
  if (live) {
    for (int k = 0; k < K; ++k) {
      tau[k] = A.tau[(int64_t)b * K + k];
      dt[k] = 0.0f;
    }
    const float invVB = 1.0f / ((float)A.V * (float)B);
    for (int v = 0; v < A.V; ++v) {
      const TtView& w = A.v[v];
      float lp[TT_MAXK];
      tt_head_logp(A.params + w.head_w, A.params + w.head_b, w.fn + (int64_t)b * w.d, w.d, K, 1.0f / A.head_temp, lp);
      for (int k = 0; k < K; ++k) {
        const float tk = fminf(fmaxf(tau[k], 1e-8f), 1.0f);
        ce -= tk * lp[k];
        if (tau[k] >= 1e-8f && tau[k] <= 1.0f) dt[k] -= lp[k] * invVB;  // clamp passes the gradient inside its range
      }
    }
    ce /= (float)A.V;
    for (int k = 0; k < K; ++k) {
      const float t9 = fmaxf(tau[k], 1e-9f);
      const float lt = logf(t9);
      hs -= t9 * lt;
      if (tau[k] >= 1e-9f) dt[k] -= A.alpha * (lt + 1.0f) / (float)B;
    }
    if (A.smooth && B > 1) {  // rho * mean over the B-1 neighbour pairs of |tau_{b+1} - tau_b|_1
      const float c = A.rho / (float)(B - 1);
      for (int k = 0; k < K; ++k) {
        if (b + 1 < B) {
          const float df = A.tau[(int64_t)(b + 1) * K + k] - tau[k];
          sm += fabsf(df);
          dt[k] -= c * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
        }
        if (b > 0) {
          const float df = tau[k] - A.tau[(int64_t)(b - 1) * K + k];
          dt[k] += c * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
        }
      }
    }
    for (int k = 0; k < K; ++k) A.dtau[(int64_t)b * K + k] = dt[k];
  }
  // block sums (K <= 64 -> one value at a time through the 1-value column sum)
  float* prow = A.partial + (int64_t)blockIdx.x * (2 * K + 4);
  for (int k = 0; k < K; ++k) {
    float v1[1] = {live ? tau[k] : 0.0f};
    dof_block_colsum<1>(v1, prow + k);
    const float t8 = live ? fmaxf(tau[k], 1e-8f) : 0.0f;
    float v2[1] = {t8 * t8};
    dof_block_colsum<1>(v2, prow + K + k);
  }
  float v3[1] = {live ? ce : 0.0f};
  dof_block_colsum<1>(v3, prow + 2 * K);
  float v4[1] = {live ? hs : 0.0f};
  dof_block_colsum<1>(v4, prow + 2 * K + 1);
  float v5[1] = {live ? sm : 0.0f};
  dof_block_colsum<1>(v5, prow + 2 * K + 2);
}

// batch-level terms: marginal-entropy gap and dead-cluster barrier -> per-cluster gradient coefficients + loss logs
__global__ void __launch_bounds__(64) k_tt_scalars(TtArgs A) {
  const int k = threadIdx.x, K = A.K;
  __shared__ float sh[TT_MAXK];
  const float B = (float)A.B;
  float m = 0.0f, u = 0.0f;
  if (k < K) {
    for (int j = 0; j < A.nblk; ++j) {
      m += A.partial[(int64_t)j * (2 * K + 4) + k];
      u += A.partial[(int64_t)j * (2 * K + 4) + K + k];
    }
    m /= B;
    u /= B;
  }
  const float m9 = fmaxf(m, 1e-9f);
  sh[k < TT_MAXK ? k : 0] = 0.0f;
  __syncthreads();
  if (k < K) sh[k] = -m9 * logf(m9);
  __syncthreads();
  float hm = 0.0f;
  for (int j = 0; j < K; ++j) hm += sh[j];
  const float gap = logf((float)K) - hm;
  const float floor_ = fmaxf(1e-4f, 0.1f / (float)K);
  __syncthreads();
  if (k < K) sh[k] = fmaxf(floor_ - u, 0.0f);
  __syncthreads();
  float dead = 0.0f;
  for (int j = 0; j < K; ++j) dead += sh[j];
  dead /= floor_ * (float)K;
  if (k < K) {
    // d/d tau[b][k] of the two batch-level terms: cm[k] + cu[k] * clamp(tau, 1e-8)
    A.scal[k] = (gap > 0.0f && m >= 1e-9f) ? A.gamma_t * (logf(m9) + 1.0f) / B : 0.0f;
    A.scal[K + k] = (u < floor_) ? -A.delta_t / (floor_ * (float)K) * 2.0f / B : 0.0f;
  }
  if (k == 0) {
    float ce = 0.0f, hs = 0.0f, sm = 0.0f;
    for (int j = 0; j < A.nblk; ++j) {
      ce += A.partial[(int64_t)j * (2 * K + 4) + 2 * K];
      hs += A.partial[(int64_t)j * (2 * K + 4) + 2 * K + 1];
      sm += A.partial[(int64_t)j * (2 * K + 4) + 2 * K + 2];
    }
    ce /= B;
    hs /= B;
    float loss = ce + A.alpha * hs + A.gamma_t * fmaxf(gap, 0.0f) + A.delta_t * dead;
    if (A.smooth && A.B > 1) loss += A.rho * sm / (B - 1.0f);
    A.logs[0] = loss; A.logs[1] = ce; A.logs[2] = hs; A.logs[3] = hm; A.logs[4] = dead;
  }
}

// d loss / d (task logits): softmax backward of the complete d loss / d tau, scaled by 1 / (V T_task)
__global__ void __launch_bounds__(256) k_tt_dlogit(TtArgs A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  const int K = A.K;
  float tau[TT_MAXK], dt[TT_MAXK], dot = 0.0f;
  for (int k = 0; k < K; ++k) {
    tau[k] = A.tau[(int64_t)b * K + k];
    dt[k] = A.dtau[(int64_t)b * K + k] + A.scal[k];
    if (tau[k] >= 1e-8f) dt[k] += A.scal[K + k] * tau[k];
    dot = fmaf(tau[k], dt[k], dot);
  }
  const float sc = 1.0f / ((float)A.V * A.task_temp);
  for (int k = 0; k < K; ++k) A.dlogit[(int64_t)b * K + k] = tau[k] * (dt[k] - dot) * sc;
}

constexpr int TT_THREADS = 1024;
// d W_v = dlogit^T f_v (raw features), d c_v = column sums; Adam on the task encoder of view blockIdx.x
__global__ void __launch_bounds__(TT_THREADS) k_tt_theta(TtArgs A) {
  __shared__ float red[TT_THREADS];
  const TtView w = A.v[blockIdx.x];
  const int K = A.K, d = w.d, B = A.B, tid = threadIdx.x;
  const int nel = K * (d + 1);
  int nchunk = TT_THREADS / nel;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > 16) nchunk = 16;
  const int rows_per = (B + nchunk - 1) / nchunk;
  for (int e0 = 0; e0 < nel; e0 += TT_THREADS / nchunk) {
    const int slot = tid / nchunk, ch = tid - slot * nchunk;
    const int e = e0 + slot;
    float acc = 0.0f;
    const bool on = slot < TT_THREADS / nchunk && e < nel;
    const int k = on ? e / (d + 1) : 0, i = on ? e - k * (d + 1) : 0;
    if (on) {
      const int b0 = ch * rows_per, b1 = (b0 + rows_per) < B ? (b0 + rows_per) : B;
      for (int b = b0; b < b1; ++b)
        acc = fmaf(A.dlogit[(int64_t)b * K + k], i < d ? w.f[(int64_t)b * d + i] : 1.0f, acc);
    }
    red[tid] = acc;
    __syncthreads();
    if (on && ch == 0) {
      float g = 0.0f;
      for (int c = 0; c < nchunk; ++c) g += red[slot * nchunk + c];
      const int64_t pi = i < d ? w.task_w + (int64_t)k * d + i : w.task_b + k;
      const float mi = 0.9f * A.adam_m[pi] + 0.1f * g;
      const float vi = 0.999f * A.adam_v[pi] + 0.001f * g * g;
      A.adam_m[pi] = mi;
      A.adam_v[pi] = vi;
      A.params[pi] -= (A.lr_theta / A.bc1) * mi / (sqrtf(vi) / sqrtf(A.bc2) + 1e-8f);
    }
    __syncthreads();
  }
}

int64_t tt_total(const DofTurtleDims* D) {
  int64_t n = 0;
  for (int v = 0; v < D->n_views; ++v) n += 2LL * D->n_clusters * (D->view_dim[v] + 1);
  return n;
}

int tt_check(const DofTurtleDims* D, const char* who) {
  if (!D || D->n_views < 1 || D->n_views > DOF_TURTLE_MAX_VIEWS || D->n_clusters < 2 || D->n_clusters > TT_MAXK ||
      D->batch < 2) {
    dof_set_error("%s: bad dims (views 1..%d, clusters 2..%d, batch >= 2)", who, DOF_TURTLE_MAX_VIEWS, TT_MAXK);
    return DOF_ERR_ARG;
  }
  for (int v = 0; v < D->n_views; ++v)
    if (D->view_dim[v] < 1 || (int64_t)D->n_clusters * (D->view_dim[v] + 1) > TT_MAXW) {
      dof_set_error("%s: view %d has %d features (K*(d+1) must be <= %d)", who, v, D->view_dim[v], TT_MAXW);
      return DOF_ERR_UNSUPPORTED;
    }
  return DOF_OK;
}

// workspace carve-up (floats): fn_v, inner-step tile partials, tau, dtau, dlogit, partial, scal
void tt_layout(const DofTurtleDims* D, float* ws, TtArgs* A, int64_t* total) {
  int64_t cur = 0;
  auto take = [&](int64_t n) {
    int64_t o = cur;
    cur += (n + 63) / 64 * 64;
    return o;
  };
  const int64_t B = D->batch, K = D->n_clusters;
  for (int v = 0; v < D->n_views; ++v) {
    const int64_t o1 = take(B * D->view_dim[v]);
    if (ws) A->v[v].fn = ws + o1;
  }
  const int nblk = (int)((B + 255) / 256);
  int64_t maxel = 0;
  for (int v = 0; v < D->n_views; ++v) maxel = maxel > K * (D->view_dim[v] + 1) ? maxel : K * (D->view_dim[v] + 1);
  const int64_t oi = take((int64_t)D->n_views * nblk * maxel);
  const int64_t ot = take(B * K), od = take(B * K), ol = take(B * K), op = take((int64_t)nblk * (2 * K + 4)), os = take(2 * K + 8);
  if (ws) {
    A->inner_partial = ws + oi; A->maxel = (int)maxel;
    A->tau = ws + ot; A->dtau = ws + od; A->dlogit = ws + ol; A->partial = ws + op; A->scal = ws + os; A->nblk = nblk;
  }
  if (total) *total = cur;
}

void tt_views(const DofTurtleDims* D, const float* const* feats, TtArgs* A) {
  // parameter order = TurtleTeacher.state_dict(): heads.heads.v.{weight,bias} for all v, then task_encoder.projs.v.*
  int64_t off = 0;
  for (int v = 0; v < D->n_views; ++v) {
    A->v[v].d = D->view_dim[v];
    A->v[v].f = feats ? feats[v] : nullptr;
    A->v[v].head_w = off; off += (int64_t)D->n_clusters * D->view_dim[v];
    A->v[v].head_b = off; off += D->n_clusters;
  }
  for (int v = 0; v < D->n_views; ++v) {
    A->v[v].task_w = off; off += (int64_t)D->n_clusters * D->view_dim[v];
    A->v[v].task_b = off; off += D->n_clusters;
  }
  A->V = D->n_views; A->K = D->n_clusters; A->B = D->batch;
}

}  // namespace

extern "C" int64_t dof_turtle_param_total(const DofTurtleDims* dims) { return dims ? tt_total(dims) : 0; }

extern "C" int64_t dof_turtle_param_offset(const DofTurtleDims* dims, int32_t task, int32_t view, int32_t bias) {
  if (!dims || view < 0 || view >= dims->n_views) return -1;
  TtArgs A;
  tt_views(dims, nullptr, &A);
  return task ? (bias ? A.v[view].task_b : A.v[view].task_w) : (bias ? A.v[view].head_b : A.v[view].head_w);
}

extern "C" int64_t dof_turtle_workspace_bytes(const DofTurtleDims* dims) {
  if (!dims) return 0;
  int64_t n = 0;
  tt_layout(dims, nullptr, nullptr, &n);
  return n * 4;
}

extern "C" int dof_turtle_fit_step(const DofTurtleDims* dims, const DofTurtleHyper* hp, const float* const* feats,
                                   float* params, float* adam_m, float* adam_v, int32_t step, int32_t outer_steps,
                                   void* workspace, float* logs, void* stream) {
  if (tt_check(dims, "dof_turtle_fit_step") != DOF_OK) return DOF_ERR_ARG;
  if (!hp || !feats || !params || !adam_m || !adam_v || !workspace || !logs || outer_steps < 1 || step < 0) {
    dof_set_error("dof_turtle_fit_step: null / bad argument");
    return DOF_ERR_ARG;
  }
  TtArgs A;
  memset(&A, 0, sizeof(A));
  tt_views(dims, feats, &A);
  tt_layout(dims, static_cast<float*>(workspace), &A, nullptr);
  A.params = params; A.adam_m = adam_m; A.adam_v = adam_v; A.logs = logs;
  const double frac = 1.0 - (double)step / (double)outer_steps;
  A.gamma_t = (float)(hp->gamma * frac);
  const double dscale = 0.6 + 0.4 * frac;
  A.delta_t = (float)(hp->delta * (dscale > 0.5 ? dscale : 0.5));
  A.alpha = hp->alpha; A.head_temp = hp->head_temp; A.task_temp = hp->task_temp; A.inner_lr = hp->inner_lr;
  A.head_wd = hp->head_wd; A.lr_theta = hp->lr_theta; A.rho = hp->rho; A.inner_steps = hp->inner_steps;
  A.normalize = hp->normalize_feats; A.smooth = (step % 2) != 0 && hp->rho > 0.0f;
  A.bc1 = (float)(1.0 - pow(0.9, step + 1)); A.bc2 = (float)(1.0 - pow(0.999, step + 1));
  hipStream_t st = (hipStream_t)stream;
  DOF_LAUNCH(k_tt_prepare, (dof_cdiv(A.B, 256)), (256), st, A, (int64_t)A.B, A.tau);
  for (int m = 0; m < A.inner_steps; ++m) {
    DOF_LAUNCH(k_tt_inner_grad, ((unsigned)A.V, (unsigned)A.nblk), (TT_TILE), st, A, A.inner_partial, A.nblk, A.maxel);
    DOF_LAUNCH(k_tt_inner_step, ((unsigned)A.V), (256), st, A, (const float*)A.inner_partial, A.nblk, A.maxel);
  }
  DOF_LAUNCH(k_tt_rows, ((unsigned)A.nblk), (256), st, A);
  DOF_LAUNCH(k_tt_scalars, (1), (64), st, A);
  DOF_LAUNCH(k_tt_dlogit, ((unsigned)A.nblk), (256), st, A);
  DOF_LAUNCH(k_tt_theta, ((unsigned)A.V), (TT_THREADS), st, A);
  return dof_check_launch("dof_turtle_fit_step");
}

extern "C" int dof_turtle_predict(const DofTurtleDims* dims, float task_temp, const float* const* feats,
                                  const float* params, int64_t n_rows, float* tau_out, void* stream) {
  if (tt_check(dims, "dof_turtle_predict") != DOF_OK) return DOF_ERR_ARG;
  if (!feats || !params || !tau_out || n_rows < 0 || !(task_temp > 0.0f)) {
    dof_set_error("dof_turtle_predict: null / bad argument");
    return DOF_ERR_ARG;
  }
  if (n_rows == 0) return DOF_OK;
  TtArgs A;
  memset(&A, 0, sizeof(A));
  tt_views(dims, feats, &A);
  A.params = const_cast<float*>(params);
  A.task_temp = task_temp;
  DOF_LAUNCH(k_tt_prepare, (dof_cdiv(n_rows, 256)), (256), (hipStream_t)stream, A, n_rows, tau_out);
  return dof_check_launch("dof_turtle_predict");
}
