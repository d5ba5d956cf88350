// Vector-quantised latent.
//
// Reference semantics restated here:
//   * VectorQuantizerPT.get_code_indices / forward   /root/reference/deepof/clustering/models_new.py:1358-1423
//       d = |x|^2 + |c_k|^2 - 2 x.c_k ; idx = argmin_k d ; quantised = codebook[:, idx] (no straight-through) ;
//       soft counts = (1/d)^2 row-normalised ; vq_loss = beta*mse(sg[q], x) + mse(q, sg[x])
//   * step_vqvae_distill                              /root/reference/deepof/clustering/training.py:312-389
//       total = -mean log p(x | dec(q)) - mean log p(x | dec(z_e)) + float(vq_loss) + float(kmeans_loss)
//       (the VQ / k-means terms are detached Python floats: they shift the value, not the gradient -- Q9)
// Codebook (L,K) with K up to a few thousand: one thread per window walks the codes (B*K*L ~ 1.7e7
// FMAs at C3), reading the wave-uniform codebook through the scalar cache.
#include "dof_rt.h"

namespace {

struct VqFwdArgs {
  const float* ze;        // [L][Bp] encoder output
  const float* codebook;  // (L,K)
  float* quant;           // [L][Bp] quantised latents (decoder input of pass 1)
  int* idx;               // [Bp]
  float* sq_partial;      // [nblk] per-block sums of |q - z_e|^2
  float *soft_out, *ze_out, *quant_out;  // reference-layout exports (B,K), (B,L), (B,L) or null
  int32_t* idx_out;       // (B) or null
  int K;
  int64_t B, Bp;
};

template <int L>
__global__ void __launch_bounds__(256) k_vq_fwd(VqFwdArgs A) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float sq[1] = {0.0f};
  if (b < A.B) {
    const dof_cfp cb = dof_cw(A.codebook);
    float ze[L];
    float x2 = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      ze[l] = A.ze[(int64_t)l * A.Bp + b];
      x2 = fmaf(ze[l], ze[l], x2);
    }
    int best = 0;
    float dbest = INFINITY, inv_sum = 0.0f;
    for (int k = 0; k < A.K; ++k) {
      float c2 = 0.0f, dot = 0.0f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float c = cb[l * A.K + k];
        c2 = fmaf(c, c, c2);
        dot = fmaf(ze[l], c, dot);
      }
      const float d = (x2 + c2) - 2.0f * dot;
      if (d < dbest) {
        dbest = d;
        best = k;
      }
      const float inv = 1.0f / d;
      inv_sum += inv * inv;
      if (A.soft_out) A.soft_out[b * A.K + k] = inv * inv;
    }
    if (A.soft_out) {
      const float r = 1.0f / inv_sum;
      for (int k = 0; k < A.K; ++k) A.soft_out[b * A.K + k] *= r;
    }
    A.idx[b] = best;
    if (A.idx_out) A.idx_out[b] = best;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float q = A.codebook[l * A.K + best];
      A.quant[(int64_t)l * A.Bp + b] = q;
      if (A.quant_out) A.quant_out[b * L + l] = q;
      if (A.ze_out) A.ze_out[b * L + l] = ze[l];
      const float df = q - ze[l];
      sq[0] = fmaf(df, df, sq[0]);
    }
  }
  dof_block_colsum<1>(sq, A.sq_partial + blockIdx.x);
}

// d codebook[:, k] = sum over windows assigned to code k of d loss / d quantised   (block = code k)
template <int L>
__global__ void __launch_bounds__(256) k_vq_codebook_grad(const float* __restrict__ dzdec /*[2][L][Bp]*/,
                                                          const int* __restrict__ idx, float* __restrict__ g_codebook,
                                                          float* __restrict__ pop, int K, int64_t B, int64_t Bp) {
  const int k = blockIdx.x;
  float vals[L + 1];
#pragma unroll
  for (int l = 0; l <= L; ++l) vals[l] = 0.0f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    if (idx[b] != k) continue;
#pragma unroll
    for (int l = 0; l < L; ++l) vals[l] += dzdec[(int64_t)l * Bp + b] + dzdec[(int64_t)(L + l) * Bp + b];
    vals[L] += 1.0f;
  }
  __shared__ float out[L + 1];
  dof_block_colsum<L + 1>(vals, out);
  __syncthreads();
  if (threadIdx.x < L) g_codebook[threadIdx.x * K + k] = out[threadIdx.x];
  if (threadIdx.x == L) pop[k] = out[L];
}

template <int L>
__global__ void __launch_bounds__(256) k_vq_denc(const float* __restrict__ dzdec, const float* __restrict__ dzh,
                                                 float* __restrict__ denc, int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float v = dzdec[(int64_t)l * Bp + b] + dzdec[(int64_t)(L + l) * Bp + b];
    if (dzh) v += dzh[(int64_t)l * Bp + b];  // distillation head on z_e
    denc[(int64_t)l * Bp + b] = v;
  }
}

struct VqLossArgs {
  const float *recon_q, *recon_e;  // per-block sums of -log p for the two decoder passes
  int n_recon;
  const float* sq_partial;
  int n_sq;
  const float* pop;  // [K] windows per code
  const float* km;   // weighted Gram-spectrum value
  const float* hyper;
  const float* distill_partial;  // per-block sums of the distillation term or null
  int n_distill;
  float* logs;
  int K, L, T;
  int64_t B;
};

__global__ void k_vq_loss(VqLossArgs A) {
  // one wavefront: partial sums by all 64 lanes, the rest on lane 0
  const float rq = dof_wave_sum_array(A.recon_q, A.n_recon), re = dof_wave_sum_array(A.recon_e, A.n_recon);
  const float sq = dof_wave_sum_array(A.sq_partial, A.n_sq);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float bt = (float)A.B * (float)A.T;
  const float enc_rec = rq / bt, rec = re / bt;
  const float vq = (A.hyper[DOF_H_VQ_BETA] + 1.0f) * sq / ((float)A.B * (float)A.L);
  int populated = 0;
  for (int k = 0; k < A.K; ++k) populated += A.pop[k] > 0.0f ? 1 : 0;
  for (int i = 0; i < DOF_LOG_COUNT; ++i) A.logs[i] = 0.0f;
  A.logs[DOF_LOG_ENC_REC] = enc_rec;
  A.logs[DOF_LOG_RECON] = rec;
  A.logs[DOF_LOG_VQ] = vq;
  A.logs[DOF_LOG_KMEANS] = A.km[0];
  A.logs[DOF_LOG_POPULATED] = (float)populated;
  float dist = 0.0f;
  if (A.distill_partial)
    for (int i = 0; i < A.n_distill; ++i) dist += A.distill_partial[i];
  A.logs[DOF_LOG_DISTILL] = dist;
  A.logs[DOF_LOG_TOTAL] = enc_rec + rec + vq + A.km[0] + dist;
}

}  // namespace
