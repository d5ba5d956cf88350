// Temporal convolutional encoder and the BatchNorm MLP head.
//
// Reference semantics restated here (/root/reference/deepof/clustering/models_new.py):
//   * TemporalBlockPT :376-443  causal pad -> Conv1d(k=4, dilation d, bias) -> BatchNorm1d(eps 1e-3, batch
//     statistics in train mode, momentum 0.1) -> ReLU, twice; residual (1x1 conv on the first block);
//     out = ReLU(a2 + res); skip = a2
//   * TCN1DPT :446-506          8 blocks, dilations 1,2,4,8,1,2,4,8; ReLU(sum of skips) at the last time step
//   * TCNEncoderPT head :593-657  x / max(rms(x), 1) -> clamp +-1e4 -> Linear -> ReLU -> BN(momentum 0.01) ->
//     Linear -> ReLU -> BN -> Linear
//
// Mapping.  Activations are channel-minor [t][s][32] (s = window*G + group, padded to 64), so one time
// step of 16 neighbouring sequences is a contiguous 2 KB run.  A 32->32 dilated convolution is a GEMM with
// K = 4 taps x 32 channels: one wave owns 16 (t, s) rows, keeps the whole 128x32 weight matrix in 64 VGPRs
// as the B operand of v_mfma_f32_16x16x4_f32 and streams the four tap rows as the A operand (each lane
// loads 8 consecutive channels = 32 B, four lanes cover a 128-byte row).  BatchNorm needs statistics over
// every sequence and time step, i.e. a grid-wide reduction between a convolution and its activation: the
// convolution writes the pre-normalisation tensor plus per-wave channel sums, a tiny kernel turns the sums
// into per-channel (scale, shift), and the CONSUMER applies scale/shift + ReLU while loading -- the
// activated tensor makes no extra HBM round trip in the forward pass.  The same kernel with the taps
// reversed and the weights transposed is the data-gradient; weight gradients go through the strided MFMA
// reduction in k_reduce.hip.
#include <cstdlib>
#include "dof_rt.h"
#include "launchers.h"

#define TRY_RC(x) do { int _rc = (x); if (_rc != DOF_OK) return _rc; } while (0)

namespace {

constexpr int TC = 32;  // conv_filters
constexpr int TK = 4;   // kernel_size

// per-layer BatchNorm record bnp[4][C]: batch (or running) mean, rstd, scale = gamma*rstd, shift = beta - mean*scale.
// Applied as y*scale + shift -- the form ATen's CPU batch_norm uses (alpha/beta "linear and constant terms"), so
// the rounding pattern matches the reference's; measured against an fp64 evaluation both forms sit at the
// reference's own fp32 noise level.
#define BNP_MEAN(p, C, c) (p)[(c)]
#define BNP_RSTD(p, C, c) (p)[(C) + (c)]
#define BNP_SCALE(p, C, c) (p)[2 * (C) + (c)]
#define BNP_SHIFT(p, C, c) (p)[3 * (C) + (c)]
#define BN_APPLY(p, C, c, y) fmaf((y), BNP_SCALE(p, C, c), BNP_SHIFT(p, C, c))

// ---------------------------------------------------------------------------------------------
// Block 0, conv1: scrambled read of the window tensor + Conv1d(F -> 32, k=4, dilation d) + bias.
// ---------------------------------------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(256) k_tcn_in_conv(const float* __restrict__ xin,  // (B,T,G,F) reference layout
                                                     const float* __restrict__ w,    // (32,F,4)
                                                     const float* __restrict__ bias, float* __restrict__ xs,  // [T][Sp][F]
                                                     float* __restrict__ y,          // [T][Sp][32]
                                                     float* __restrict__ partial,    // [nblk][64] sums, or [nblk][3][32] records
                                                     int T, int G, int64_t S, int64_t Sp, int dil, int rec) {
  // rec: the block's channel statistics as one mergeable (n, mean, M2) record per channel (k_tcn_stat_merge), from sums
  // about the block's own mean -- the layer then needs no second pass over y (k_tcn_var)
  __shared__ float kshift[TC], bsum[2 * TC];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float st[2 * TC];
#pragma unroll
  for (int c = 0; c < 2 * TC; ++c) st[c] = 0.0f;
  if (i < (int64_t)T * S) {
    const int to = (int)(i / S);
    const int64_t s = i - (int64_t)to * S;
    const int64_t b = s / G;
    const int g = (int)(s - b * G);
    const float* __restrict__ win = xin + b * (int64_t)T * G * F;
    const dof_cfp wc = dof_cw(w);
    float rows[TK][F];
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      const int tt = to - (TK - 1 - k) * dil;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        float v = 0.0f;
        if (tt >= 0) {  // y[b,g,tt,f] = x[b, t, cc] with cc*T + t = (f*T + tt)*G + g   (models_new.py:616-619)
          const int lin = (f * T + tt) * G + g;
          const int cc = lin / T;
          v = win[(int64_t)(lin - cc * T) * G * F + cc];
        }
        rows[k][f] = v;
      }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) xs[ACT(to, f, F, Sp, s)] = rows[TK - 1][f];
    float out[TC];
#pragma unroll
    for (int o = 0; o < TC; ++o) {
      float acc = dof_cw(bias)[o];
#pragma unroll
      for (int f = 0; f < F; ++f)
#pragma unroll
        for (int k = 0; k < TK; ++k) acc = fmaf(wc[(o * F + f) * TK + k], rows[k][f], acc);
      out[o] = acc;
      st[o] = acc;
      st[TC + o] = acc * acc;
    }
    dof_st_row<TC>(y + ACT(to, 0, TC, Sp, s), out);
  }
  if (!rec) {
    dof_block_colsum<2 * TC>(st, partial + (int64_t)blockIdx.x * 2 * TC);
    return;
  }
  // shift = the block's own mean (a first 32-value block sum; the values stay in registers): with the block's first row as
  // the shift the ill-conditioned 6-window fixture lost a factor 6 against the centred second pass
  // (encoder.node_tcn.blocks.0.conv1.weight 0.81 against 0.13 on a scale of 162, tests/parity_common.py::run_vqvae_tcn_check)
  const int64_t left = (int64_t)T * S - (int64_t)blockIdx.x * blockDim.x;
  const float nrows = (float)(left < (int64_t)blockDim.x ? left : (int64_t)blockDim.x);
  dof_block_colsum<TC>(st, bsum);
  __syncthreads();
  if (threadIdx.x < TC) kshift[threadIdx.x] = bsum[threadIdx.x] / nrows;
  __syncthreads();
  const bool live = i < (int64_t)T * S;
#pragma unroll
  for (int o = 0; o < TC; ++o) {
    const float dv = live ? st[o] - kshift[o] : 0.0f;
    st[o] = dv;
    st[TC + o] = dv * dv;
  }
  dof_block_colsum<2 * TC>(st, bsum);
  __syncthreads();
  if (threadIdx.x < TC) {
    const int c = threadIdx.x;
    const float n = nrows;
    const float dm = bsum[c] / n;
    float* r = partial + (int64_t)blockIdx.x * 3 * TC;
    r[c] = n;
    r[TC + c] = kshift[c] + dm;
    r[2 * TC + c] = fmaxf(bsum[TC + c] - bsum[c] * dm, 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------
// 32 -> 32 dilated convolution on the matrix cores.
//   REVERSE = false: y[t] = bias + sum_j W[:, :, j] in[t - (3-j) d]           (+ channel sums of y, y^2)
//   REVERSE = true : out[t] (+)= sum_j W[:, :, j]^T in[t + (3-j) d]           (data gradient)
//   BN_IN: the loaded rows are pre-normalisation values; scale/shift + ReLU is applied on the fly and the
//          activated own row (tap offset 0) is stored to a_out for the backward pass.
// ---------------------------------------------------------------------------------------------
struct TcnConvArgs {
  const float* in;      // [T][Sp][32]
  const float* w;       // (32,32,4)
  const float* bias;    // (32) or null
  const float* bnp_in;  // BatchNorm record of the producer (BN_IN)
  float* a_out;         // [T][Sp][32] activated input (BN_IN, forward only) or null
  float* out;           // [T][Sp][32]
  float* partial;       // [n_waves][64] channel sums (forward / fused backward) or null
  const float* fuse_y;    // FUSE_BN: pre-normalisation output of the BatchNorm+ReLU in front of this (reverse) conv's result
  const float* fuse_bnp;  // FUSE_BN: its record
  const float* bwd_y;     // BWD2 (k_tcn_conv_t): pre-normalisation tensor of the BatchNorm whose pass-1 gradient `in` holds
  const float* bwd_bnp;   // BWD2: its record
  const float* bwd_coef;  // BWD2: (mean g | mean g * xhat) of that BatchNorm
  int bwd_store;          // BWD2: 1 = write dy back over `in` (0: the weight-gradient kernel applies pass 2 itself)
  const float* stat_shift;  // forward k_tcn_conv_t: per-channel shift K of the channel sums (sum (y - K) | sum (y - K)^2), or null (see k_bn_fwd_fin)
  // COMB of block 1 (k_tcn_conv_b<.., DS0>): `in` = the raw input rows xs [T][Sp][ds_F] and the residual of block 0 is its
  // 1 x 1 downsample convolution ds_w (32, ds_F) / ds_b (32), computed while the tile is staged
  const float* ds_w = nullptr;
  const float* ds_b = nullptr;
  int ds_F = 0;
  // TAIL (k_tcn_conv_t, conv1's data gradient of block b + 1): the backward of block b's tail in the epilogue
  const float* tail_src = nullptr;    // gradient already waiting at block b's output (the residual branch of block b + 1)
  float* tail_gres = nullptr;         // masked gradient = what enters block b's residual branch
  const float* tail_skip = nullptr;   // final skip-sum (mask of the last-step feature gradient)
  const float* tail_dfeat = nullptr;  // [32][Sp] gradient of the last-step features
  // round 4: the ReLU mask of a block output as one word per (t, s) row (bit c = out[t][s][c] > 0), [T][Sp] -- written by the
  // COMB convolution that computes the output, read by the TAIL convolution instead of the output tensor itself (1 / 32 of it)
  uint32_t* relu_mask_out = nullptr;
  const uint32_t* tail_mask = nullptr;
  // forward k_tcn_conv_t: 1 = the workgroup's channel statistics leave as mergeable (n | mean | M2) records,
  // partial[workgroup][3][32] (see k_tcn_stat_merge), instead of plain / shifted sums
  int stat_records = 0;
  // round 6 (k_tcn_conv_b WGRAD): the convolution's weight gradient accumulated in the same launch -- wg_x = the forward
  // input of the convolution when the epilogue cannot recompute it (TAIL: the previous block's output), wg_partials + wg_part0 /
  // wg_part1 = the partial-tile regions of taps 0, 1 / 2, 3 ([workgroup][64][65], k_tcn_wgrad_b3's layout)
  const float* wg_x = nullptr;
  float* wg_partials = nullptr;
  int64_t wg_part0 = 0, wg_part1 = 0;
  int T, dil, accumulate;
  int64_t S, Sp;
};

// FUSE_BN (reverse only): the result is the gradient entering a ReLU(BatchNorm(y)) -- the first pass of that
// BatchNorm's backward (mask by the activation, channel sums of g and g * xhat) runs in the epilogue, so the
// gradient is written once, already masked, instead of written, re-read and rewritten by k_tcn_bn_bwd1.
template <bool REVERSE, bool BN_IN, bool FUSE_BN = false>
__global__ void __launch_bounds__(256) k_tcn_conv(TcnConvArgs A) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int i = lane & 15, kk = lane >> 4;
  // B operand: k-step (tap j, q) covers input channels {kk*8 + q}; lane (kk, col) holds the weight that
  // multiplies channel kk*8+q of tap j for output column ct*16+col.
  float wr[TK][8][2];
#pragma unroll
  for (int j = 0; j < TK; ++j)
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int cin = kk * 8 + q, col = ct * 16 + i;
        wr[j][q][ct] = REVERSE ? A.w[(cin * TC + col) * TK + j] : A.w[(col * TC + cin) * TK + j];
      }
  float sc[8], sh[8];
  if (BN_IN) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      sc[q] = BNP_SCALE(A.bnp_in, TC, kk * 8 + q);
      sh[q] = BNP_SHIFT(A.bnp_in, TC, kk * 8 + q);
    }
  }
  const float b0 = (!REVERSE && A.bias) ? A.bias[i] : 0.0f;
  const float b1 = (!REVERSE && A.bias) ? A.bias[16 + i] : 0.0f;
  float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
  const int64_t tiles_per_t = A.Sp / 16;
  const int64_t n_tiles = (int64_t)A.T * tiles_per_t;
  for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
    const int t = (int)(tile / tiles_per_t);
    const int64_t s0 = (tile - (int64_t)t * tiles_per_t) * 16;
    dof_f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int tt = REVERSE ? t + (TK - 1 - j) * A.dil : t - (TK - 1 - j) * A.dil;
      float a[8];
      if (tt >= 0 && tt < A.T) {  // wave-uniform
        dof_ld_row<8>(A.in + ACT(tt, kk * 8, TC, A.Sp, s0 + i), a);
        if (BN_IN) {
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] = fmaxf(fmaf(a[q], sc[q], sh[q]), 0.0f);
          if (!REVERSE && j == TK - 1 && A.a_out && s0 + i < A.S) dof_st_row<8>(A.a_out + ACT(t, kk * 8, TC, A.Sp, s0 + i), a);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wr[j][q][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wr[j][q][1], acc1, 0, 0, 0);
        }
      }
    }
    // D layout: lane holds rows kk*4 + r, column i (of column tile 0 / 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t s = s0 + kk * 4 + r;
      if (s < A.S) {
        float* o = A.out + ACT(t, 0, TC, A.Sp, s);
        float v0 = acc0[r], v1 = acc1[r];
        if (REVERSE && A.accumulate) {
          v0 += o[i];
          v1 += o[16 + i];
        }
        if (FUSE_BN) {
          const float* yr = A.fuse_y + ACT(t, 0, TC, A.Sp, s);
          const float y0 = yr[i], y1 = yr[16 + i];
          v0 = BN_APPLY(A.fuse_bnp, TC, i, y0) > 0.0f ? v0 : 0.0f;
          v1 = BN_APPLY(A.fuse_bnp, TC, 16 + i, y1) > 0.0f ? v1 : 0.0f;
          s1[0] += v0; s2[0] = fmaf(v0, (y0 - BNP_MEAN(A.fuse_bnp, TC, i)) * BNP_RSTD(A.fuse_bnp, TC, i), s2[0]);
          s1[1] += v1;
          s2[1] = fmaf(v1, (y1 - BNP_MEAN(A.fuse_bnp, TC, 16 + i)) * BNP_RSTD(A.fuse_bnp, TC, 16 + i), s2[1]);
        }
        o[i] = v0;
        o[16 + i] = v1;
        if (!REVERSE) {
          s1[0] += v0; s2[0] = fmaf(v0, v0, s2[0]);
          s1[1] += v1; s2[1] = fmaf(v1, v1, s2[1]);
        }
      }
    }
  }
  if ((!REVERSE || FUSE_BN) && A.partial) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      s1[ct] += __shfl_xor(s1[ct], 16); s1[ct] += __shfl_xor(s1[ct], 32);
      s2[ct] += __shfl_xor(s2[ct], 16); s2[ct] += __shfl_xor(s2[ct], 32);
    }
    if (kk == 0) {
      float* p = A.partial + (int64_t)wave * 2 * TC;
      p[i] = s1[0]; p[16 + i] = s1[1];
      p[TC + i] = s2[0]; p[TC + 16 + i] = s2[1];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Time-resident form of the same convolution for windows of up to TCT_T steps (the layout is [t][s][32], so the
// four taps of an output row lie T-strided apart and k_tcn_conv fetches every input row four times -- 1.5 GB per
// layer at batch 8192, from the Infinity Cache at best).  Here a workgroup owns 16 sequences for ALL time steps:
// the (T x 16 x 32) input tile is staged in LDS once with coalesced 16-byte loads (50 KB at T = 25, three
// workgroups per CU overlap their load and MFMA phases), the producer's BatchNorm + ReLU (BN_IN) or the second
// pass of a BatchNorm backward (BWD2: dy = scale (g - mean g - xhat mean(g xhat)), written back in place for the
// weight-gradient kernel) is applied once per element while staging instead of once per tap, and wavefront w
// computes the output rows t = w, w + 4, ...  The 16-byte chunks of a row are XOR-swizzled by the sequence index
// so that the ds_read_b128 lane groups hit 16 distinct slots.  The MFMA operands are swapped against k_tcn_conv
// (A = weights, B = input rows): D[channel][sequence] leaves every lane with four consecutive channels of ONE
// sequence, i.e. a 16-byte store (and 16-byte loads of the epilogue operands) instead of four scattered dwords.
// ---------------------------------------------------------------------------------------------
constexpr int TCT_T = 25;

// Chan / Golub / LeVeque update of (n, mean, M2) by a second record; an empty record leaves the other unchanged
__device__ __forceinline__ void dof_stat_merge(float& n, float& mean, float& m2, float nb, float mb, float qb) {
  if (nb == 0.0f) return;
  if (n == 0.0f) {
    n = nb; mean = mb; m2 = qb;
    return;
  }
  const float nt = n + nb, d = mb - mean;
  mean = fmaf(d, nb / nt, mean);
  m2 = m2 + qb + d * d * (n * nb / nt);
  n = nt;
}

// workgroup records partial[nblk][3][32] -> sums = (n mean | M2) per channel, what k_bn_fwd_fin expects of the two-pass
// statistics.  One workgroup per channel: 256 strided runs merged sequentially (3 records each at 768 workgroups, all
// loaded before the first merge), then a fixed tree.
// fin (gamma != null): the channel's BatchNorm record and running buffers straight from the merged record -- k_bn_fwd_fin's
// train branch on (sums[c], sums[C + c]) = (n mean, M2), one launch instead of two per layer.
struct StatFinArgs {
  float count;
  const float *gamma, *beta;
  float *rmean, *rvar;
  float momentum;
  float* bnp;
};
__global__ void __launch_bounds__(256) k_tcn_stat_merge(const float* __restrict__ partial, int nblk, float* __restrict__ sums,
                                                        StatFinArgs F) {
  __shared__ float rn[256], rm[256], rq[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  float n = 0.0f, mean = 0.0f, m2 = 0.0f;
  for (int b0 = tid; b0 < nblk; b0 += 4 * 256) {
    float pn[4], pm[4], pq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + u * 256;
      const float* p = partial + (int64_t)(b < nblk ? b : 0) * 3 * TC;
      pn[u] = b < nblk ? p[c] : 0.0f;
      pm[u] = p[TC + c];
      pq[u] = p[2 * TC + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) dof_stat_merge(n, mean, m2, pn[u], pm[u], pq[u]);
  }
  rn[tid] = n; rm[tid] = mean; rq[tid] = m2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      float a = rn[tid], b = rm[tid], q = rq[tid];
      dof_stat_merge(a, b, q, rn[tid + w], rm[tid + w], rq[tid + w]);
      rn[tid] = a; rm[tid] = b; rq[tid] = q;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float s1 = rn[0] * rm[0], m2 = rq[0];
    sums[c] = s1;
    sums[TC + c] = m2;
    if (F.gamma) {
      const float mean = s1 / F.count, var = m2 / F.count;
      F.rmean[c] = (1.0f - F.momentum) * F.rmean[c] + F.momentum * mean;
      F.rvar[c] = (1.0f - F.momentum) * F.rvar[c] + F.momentum * var * (F.count / fmaxf(F.count - 1.0f, 1.0f));
      const float rstd = 1.0f / sqrtf(var + 1e-3f);
      const float scale = F.gamma[c] * rstd;
      BNP_MEAN(F.bnp, TC, c) = mean;
      BNP_RSTD(F.bnp, TC, c) = rstd;
      BNP_SCALE(F.bnp, TC, c) = scale;
      BNP_SHIFT(F.bnp, TC, c) = F.beta[c] - mean * scale;
    }
  }
}

// Backward twin: the channel sums (sum g | sum g xhat) of a 32-channel layer from its producer's per-workgroup partials
// ([nblk][64], k_sum_partials' arithmetic for both values of the channel) AND k_bn_bwd_fin's step on them: gamma / beta
// gradients and the two batch means pass 2 needs -- one launch instead of two per layer.
__global__ void __launch_bounds__(256) k_bn_bwd_sum_fin(const float* __restrict__ partial, int64_t nblk, float count,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                        float* __restrict__ coef, float* __restrict__ sums) {
  __shared__ float red[2][256];
  const int c = blockIdx.x;
  float a0 = 0.0f, a1 = 0.0f;
  for (int64_t b = threadIdx.x; b < nblk; b += 256) {
    a0 += partial[b * 2 * TC + c];
    a1 += partial[b * 2 * TC + TC + c];
  }
  red[0][threadIdx.x] = a0;
  red[1][threadIdx.x] = a1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float sg = red[0][0], sgx = red[1][0];
    sums[c] = sg;
    sums[TC + c] = sgx;
    dbeta[c] = accumulate ? dbeta[c] + sg : sg;
    dgamma[c] = accumulate ? dgamma[c] + sgx : sgx;
    coef[c] = sg / count;
    coef[TC + c] = sgx / count;
  }
}

// OR over the eight lanes of a half row (the lanes holding the 16-byte chunks of one [32]-channel row): every lane ends with it
__device__ __forceinline__ uint32_t tct_or8(uint32_t w) {
  w |= __builtin_bit_cast(uint32_t, dof_dpp_perm<0xB1>(__builtin_bit_cast(float, w)));
  w |= __builtin_bit_cast(uint32_t, dof_dpp_perm<0x4E>(__builtin_bit_cast(float, w)));
  w |= __builtin_bit_cast(uint32_t, dof_dpp_perm<0x141>(__builtin_bit_cast(float, w)));
  return w;
}
// a row's ReLU mask word from the four values of chunk ch held by each of its eight lanes: bit c = value of channel c > 0
__device__ __forceinline__ uint32_t tct_row_mask(const float* e, int ch) {
  const uint32_t nib = (e[0] > 0.0f ? 1u : 0u) | (e[1] > 0.0f ? 2u : 0u) | (e[2] > 0.0f ? 4u : 0u) | (e[3] > 0.0f ? 8u : 0u);
  return tct_or8(nib << (4 * ch));
}

// NS = 16: sequences s, s + 1 of a time step share a 256-byte bank row, the chunk swizzle by s >> 1 spreads the 16 lanes of a
// ds_read_b128 pass over the 16 bank groups.  NS = 8 (two time steps per MFMA column block): the lanes of a pass are 8
// sequences of row tt and the same 8 of row tt + 1 -- the row parity goes into the swizzle's top bit.
template <int NS>
__device__ __forceinline__ int tct_slot(int t, int sq, int chunk) {
  if (NS == 16) return (t * 16 + sq) * 8 + (chunk ^ ((sq >> 1) & 7));
  return (t * 8 + sq) * 8 + (chunk ^ (((sq >> 1) & 3) | ((t & 1) << 2)));
}

// TAIL (with REVERSE, FUSE_BN, BWD2; round 3): the convolution is conv1's data gradient of block b + 1, its result plus
// tail_src is the complete gradient at block b's output, and the epilogue runs the backward of block b's tail on it --
// mask by the block output, store the residual-branch gradient, add the last-step feature gradient, then (FUSE_BN) the
// first pass of BatchNorm2's backward of block b.  The gradient at the block output is never written and k_tcn_bn_bwd1_w's
// pass over it (3 reads, 2 writes per element) shrinks to 2 more reads here.
// COMB (forward, round 3): the input tile is the previous block's OUTPUT, computed while staging from that block's conv2
// result and residual input -- out = ReLU(ReLU(BN2(y2)) + res) with `in` = res, bwd_y = y2, bnp_in = BatchNorm2's record --
// and written to a_out for the backward pass: k_tcn_combine's pass over (y2, res, out) becomes one more read here.
// NS (round 4): sequences per workgroup.  16 for T <= 25; 8 for T <= 50 -- the same 52 KB tile holds 8 sequences x 50 steps,
// an MFMA column block is 8 sequences x 2 CONSECUTIVE output rows (lane i: sequence i & 7, row t0 + (i >> 3)), a tap that is
// outside the window for one of the two rows only is zeroed per lane (the wave-uniform skip needs both outside).
template <bool REVERSE, bool BN_IN, bool FUSE_BN, bool BWD2, bool TAIL = false, bool COMB = false, int NS = 16>
__global__ void __launch_bounds__(256, 3) k_tcn_conv_t(TcnConvArgs A) {
  static_assert(NS == 16 || NS == 8, "16 sequences x 25 steps or 8 sequences x 50 steps");
  constexpr int TPC = 16 / NS;         // output rows per MFMA column block
  constexpr int TS = 256 / (NS * 8);   // time steps per staging pass of the 256 threads
  static_assert(!TAIL || (REVERSE && FUSE_BN && BWD2), "the tail epilogue extends the fused data-gradient variant");
  static_assert(!COMB || (!REVERSE && !BN_IN && !FUSE_BN && !BWD2), "the combine-on-load variant is a plain forward convolution");
  __shared__ float4 tile[(TCT_T + 1) * 16 * 8];  // + one row for the unconditional staging of an odd T
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int sl = i & (NS - 1), tsub = i / NS;  // the lane's sequence of the group and its row of the column block
  // wavefront -> (output-channel half ct, time parity): 32 of the 128 x 32 weights per lane.
  // A operand: k-step (tap j, q) multiplies input channel (q < 4 ? 0 : 16) + kk*4 + (q & 3) -- the 16-byte chunks
  // kk and kk + 4 of a staged row; lane (kk, i) holds the weight of that channel for output channel ct*16 + i.
  const int ct = wv & 1, tpar = wv >> 1;
  float wr[TK][8];
#pragma unroll
  for (int j = 0; j < TK; ++j)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cin = (q < 4 ? 0 : 16) + kk * 4 + (q & 3), col = ct * 16 + i;
      wr[j][q] = REVERSE ? A.w[(cin * TC + col) * TK + j] : A.w[(col * TC + cin) * TK + j];
    }
  // staging: thread -> (time parity, sequence, 16-byte chunk); its four channels are fixed
  const int half = threadIdx.x / (NS * 8), sq = (threadIdx.x % (NS * 8)) >> 3, ch = threadIdx.x & 7;
  // epilogue constants: output channels ct*16 + kk*4 + r (FUSE_BN: the BatchNorm record waits in LDS)
  __shared__ float4 frec[FUSE_BN ? 4 * TC / 4 : 1];
  if (FUSE_BN) {
    if (threadIdx.x < 4 * TC / 4) frec[threadIdx.x] = reinterpret_cast<const float4*>(A.fuse_bnp)[threadIdx.x];
  }
  float bias[4], kshift[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bias[r] = (!REVERSE && A.bias) ? A.bias[ct * 16 + kk * 4 + r] : 0.0f;
    kshift[r] = (!REVERSE && A.stat_shift) ? A.stat_shift[ct * 16 + kk * 4 + r] : 0.0f;
  }
  float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float n_rows = 0.0f;  // stat_records: rows this lane has summed; its sums are taken about the first one
  const int T = A.T;
  const int64_t n_groups = A.Sp / NS;
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t s0 = grp * NS;
    // per-thread BatchNorm constants of the staging phase, (re)loaded per group: they are dead during the MFMA phase
    // (the fence keeps the compiler from hoisting them out of the loop into registers that phase needs)
    DOF_MEM_FENCE();
    float k0[4], k1[4];                        // BN_IN: scale, shift of the producer's BatchNorm
    float bm[4], br[4], bs[4], c1[4], c2[4];  // BWD2: mean, rstd, scale, mean g, mean g xhat
    if (BN_IN || COMB) {
      dof_ld_row<4>(A.bnp_in + 2 * TC + ch * 4, k0);
      dof_ld_row<4>(A.bnp_in + 3 * TC + ch * 4, k1);
    }
    if (BWD2) {
      dof_ld_row<4>(A.bwd_bnp + ch * 4, bm);
      dof_ld_row<4>(A.bwd_bnp + TC + ch * 4, br);
      dof_ld_row<4>(A.bwd_bnp + 2 * TC + ch * 4, bs);
      dof_ld_row<4>(A.bwd_coef + ch * 4, c1);
      dof_ld_row<4>(A.bwd_coef + TC + ch * 4, c2);
    }
    // ---- stage the group's rows: two time steps per pass over the 256 threads, a batch of loads in flight.  The
    // loads are unconditional (steps past T re-read step T - 1 and land in LDS rows nobody reads): a predicate
    // around them would serialise the batch on vmcnt(0).
    constexpr int NP = (TCT_T * TPC + TS - 1) / TS, NBATCH = COMB ? 2 : BWD2 ? 3 : 7;
    const bool srow = s0 + sq < A.S;
    // 32-bit element offsets (the launcher checks T * Sp * 32 < 2^31): SGPR base + one VGPR per address
    const uint32_t row_stride = (uint32_t)A.Sp * TC;
    const uint32_t st_base = (uint32_t)(s0 + sq) * TC + ch * 4;
#pragma unroll
    for (int n0 = 0; n0 < NP; n0 += NBATCH) {
      float4 v[NBATCH], yv[NBATCH];
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        if (n0 + u < NP) {
          const int t = TS * (n0 + u) + half;
          const uint32_t off = st_base + (uint32_t)(t < T ? t : T - 1) * row_stride;
          v[u] = *reinterpret_cast<const float4*>(A.in + off);
          if (BWD2 || COMB) yv[u] = *reinterpret_cast<const float4*>(A.bwd_y + off);
        }
      }
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        if (n0 + u < NP) {
          const int t = TS * (n0 + u) + half;
          float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          if (BN_IN) {
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = fmaxf(fmaf(e[c], k0[c], k1[c]), 0.0f);
          }
          if (COMB) {
            const float y4[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = fmaxf(fmaxf(fmaf(y4[c], k0[c], k1[c]), 0.0f) + e[c], 0.0f);
          }
          if (BWD2) {
            const float y4[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float xh = (y4[c] - bm[c]) * br[c];
              e[c] = bs[c] * (e[c] - c1[c] - xh * c2[c]);
            }
          }
          const float4 w4 = make_float4(e[0], e[1], e[2], e[3]);
          tile[tct_slot<NS>(t, sq, ch)] = w4;
          if (COMB) {  // the row's 32 sign bits for the TAIL convolution of the backward pass
            const uint32_t wbits = tct_row_mask(e, ch);
            if (ch == 0 && srow && t < T) A.relu_mask_out[(uint32_t)t * (uint32_t)A.Sp + (uint32_t)(s0 + sq)] = wbits;
          }
          if (srow && t < T) {
            const uint32_t off = st_base + (uint32_t)t * row_stride;
            if ((BN_IN || COMB) && !REVERSE && A.a_out) *reinterpret_cast<float4*>(A.a_out + off) = w4;
            if (BWD2 && A.bwd_store) *reinterpret_cast<float4*>(const_cast<float*>(A.in) + off) = w4;
          }
        }
      }
    }
    __syncthreads();
    // ---- output rows t = tpar, tpar + 2, ... (NS = 8: row pairs 2 tpar, 2 tpar + 4, ...) of channel half ct; the
    // epilogue's global operand (the accumulation target or the BatchNorm input of FUSE_BN) is requested two rows ahead
    const int64_t s = s0 + sl;
    const bool ok_s = s < A.S;
    constexpr bool PRE = REVERSE;  // the reverse variants read one epilogue operand (accumulate XOR FUSE_BN)
    const float* pre_src = FUSE_BN ? A.fuse_y : (const float*)A.out;
    const bool pre_on = PRE && ok_s && (FUSE_BN || A.accumulate);
    const uint32_t ep_base = (uint32_t)s * TC + ct * 16 + kk * 4;
    const uint32_t pre_base = (uint32_t)(pre_on ? s : s0) * TC + ct * 16 + kk * 4;  // padded lanes read a valid row and ignore it
    float4 p0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), p1 = p0;
    float4 ts0 = p0;    // TAIL: row of tail_src, requested one output row ahead (register budget) ...
    uint32_t tm0 = 0u;  // ... and the block output's ReLU mask word of that row (round 4: tail_mask instead of the tensor)
    const int tl0 = tpar * TPC + tsub;  // the lane's first output row; its next ones are 2 TPC apart
    const uint32_t mk_base = (uint32_t)(pre_on ? s : s0);
    if (PRE && (FUSE_BN || A.accumulate)) {
      const uint32_t o0 = pre_base + (uint32_t)(tl0 < T ? tl0 : T - 1) * row_stride;
      const uint32_t o1 = pre_base + (uint32_t)(tl0 + 2 * TPC < T ? tl0 + 2 * TPC : T - 1) * row_stride;
      p0 = *reinterpret_cast<const float4*>(pre_src + o0);
      p1 = *reinterpret_cast<const float4*>(pre_src + o1);
      if (TAIL) {
        ts0 = *reinterpret_cast<const float4*>(A.tail_src + o0);
        tm0 = A.tail_mask[mk_base + (uint32_t)(tl0 < T ? tl0 : T - 1) * (uint32_t)A.Sp];
      }
    }
    for (int t0 = tpar * TPC; t0 < T; t0 += 2 * TPC) {
      const int t = t0 + tsub;
      const bool ok = ok_s && (TPC == 1 || t < T);
      const uint32_t off = ep_base + (uint32_t)t * row_stride;
      const float4 pc = p0, tsc = ts0;
      const uint32_t tmc = tm0;
      if (PRE && (FUSE_BN || A.accumulate)) {
        const uint32_t o2 = pre_base + (uint32_t)(t + 4 * TPC < T ? t + 4 * TPC : T - 1) * row_stride;
        p0 = p1;
        p1 = *reinterpret_cast<const float4*>(pre_src + o2);
        if (TAIL) {
          const uint32_t o1n = pre_base + (uint32_t)(t + 2 * TPC < T ? t + 2 * TPC : T - 1) * row_stride;
          ts0 = *reinterpret_cast<const float4*>(A.tail_src + o1n);
          tm0 = A.tail_mask[mk_base + (uint32_t)(t + 2 * TPC < T ? t + 2 * TPC : T - 1) * (uint32_t)A.Sp];
        }
      }
      dof_f32x4 acc = {bias[0], bias[1], bias[2], bias[3]};
      // all valid taps' rows are requested from LDS before the first MFMA (validity is wave-uniform; NS = 8: of either row)
      float4 lo[TK], hi[TK];
      bool tap[TK], mine[TK];
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const int sh = REVERSE ? (TK - 1 - j) * A.dil : -(TK - 1 - j) * A.dil;
        const int tu = t0 + sh;  // the tap's row for the column block's first output row
        tap[j] = (tu >= 0 && tu < T) || (TPC == 2 && tu + 1 >= 0 && tu + 1 < T);
        const int tt = t + sh;
        mine[j] = TPC == 1 || (tt >= 0 && tt < T);
        if (tap[j]) {
          const int tc = TPC == 1 ? tt : (tt < 0 ? 0 : tt < T ? tt : T - 1);
          lo[j] = tile[tct_slot<NS>(tc, sl, kk)];
          hi[j] = tile[tct_slot<NS>(tc, sl, kk + 4)];
        }
      }
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        if (tap[j]) {
          float a[8] = {lo[j].x, lo[j].y, lo[j].z, lo[j].w, hi[j].x, hi[j].y, hi[j].z, hi[j].w};
          if (TPC == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = mine[j] ? a[q] : 0.0f;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][q], a[q], acc, 0, 0, 0);
        }
      }
      if (ok) {
        float v0[4] = {acc[0], acc[1], acc[2], acc[3]};
        if (REVERSE && !FUSE_BN && A.accumulate) {
          v0[0] += pc.x; v0[1] += pc.y; v0[2] += pc.z; v0[3] += pc.w;
        }
        if (TAIL) {
          const float sv[4] = {tsc.x, tsc.y, tsc.z, tsc.w};
          const uint32_t nib = tmc >> (ct * 16 + kk * 4);  // bit r: out[t][s][ct*16 + kk*4 + r] > 0
#pragma unroll
          for (int r = 0; r < 4; ++r) v0[r] = ((nib >> r) & 1u) != 0u ? v0[r] + sv[r] : 0.0f;
          *reinterpret_cast<float4*>(A.tail_gres + off) = make_float4(v0[0], v0[1], v0[2], v0[3]);
          if (A.tail_dfeat && t == T - 1) {
            const float4 sk = *reinterpret_cast<const float4*>(A.tail_skip + off);
            const float sk4[4] = {sk.x, sk.y, sk.z, sk.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
              v0[r] += sk4[r] > 0.0f ? A.tail_dfeat[(int64_t)(ct * 16 + kk * 4 + r) * A.Sp + s] : 0.0f;
          }
        }
        if (FUSE_BN) {
          const float ya[4] = {pc.x, pc.y, pc.z, pc.w};
          const int cw = ct * 4 + kk;
          const float4 m4 = frec[cw], r4 = frec[TC / 4 + cw], sc4 = frec[2 * TC / 4 + cw], sh4 = frec[3 * TC / 4 + cw];
          const float fm[4] = {m4.x, m4.y, m4.z, m4.w}, fr[4] = {r4.x, r4.y, r4.z, r4.w};
          const float fsc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, fsh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v0[r] = fmaf(ya[r], fsc[r], fsh[r]) > 0.0f ? v0[r] : 0.0f;
            s1[r] += v0[r];
            s2[r] = fmaf(v0[r], (ya[r] - fm[r]) * fr[r], s2[r]);
          }
        }
        *reinterpret_cast<float4*>(A.out + off) = make_float4(v0[0], v0[1], v0[2], v0[3]);
        if (!REVERSE) {
          if (A.stat_records && n_rows == 0.0f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kshift[r] = v0[r];
          }
          n_rows += 1.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dv = v0[r] - kshift[r];
            s1[r] += dv;
            s2[r] = fmaf(dv, dv, s2[r]);
          }
        }
      }
    }
    __syncthreads();
  }
  // channel sums of the workgroup: partial[workgroup][64] = (sum | second sum).  A wavefront covers one channel half
  // (ct) and one time parity; the two wavefronts of a half are added through LDS in a fixed order.
  if (!REVERSE && A.partial && A.stat_records) {
    // One-pass statistics without a reference value: every lane's sums are about ITS first output value (no cancellation
    // beyond the spread of the data), turned into (n, mean, M2) and merged pairwise with Chan's update -- 16 lanes of a
    // row, the two wavefronts of a channel half, then the workgroups (k_tcn_stat_merge) -- always in the same order.
    float* rec = reinterpret_cast<float*>(tile);  // [wavefront][3][16]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float n = n_rows, mean = 0.0f, m2 = 0.0f;
      if (n > 0.0f) {
        const float d = s1[r] / n;
        mean = kshift[r] + d;
        m2 = fmaxf(s2[r] - s1[r] * d, 0.0f);
      }
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        const float nb = __shfl_xor(n, m), mb = __shfl_xor(mean, m), qb = __shfl_xor(m2, m);
        dof_stat_merge(n, mean, m2, nb, mb, qb);
      }
      if (i == 0) {
        rec[(wv * 3 + 0) * 16 + kk * 4 + r] = n;
        rec[(wv * 3 + 1) * 16 + kk * 4 + r] = mean;
        rec[(wv * 3 + 2) * 16 + kk * 4 + r] = m2;
      }
    }
    __syncthreads();
    if (threadIdx.x < TC) {  // channel c: wavefronts h and h + 2 hold its half
      const int c = threadIdx.x, h = c >> 4, cl = c & 15;
      float n = rec[(h * 3 + 0) * 16 + cl], mean = rec[(h * 3 + 1) * 16 + cl], m2 = rec[(h * 3 + 2) * 16 + cl];
      dof_stat_merge(n, mean, m2, rec[((h + 2) * 3 + 0) * 16 + cl], rec[((h + 2) * 3 + 1) * 16 + cl], rec[((h + 2) * 3 + 2) * 16 + cl]);
      float* out = A.partial + (int64_t)blockIdx.x * 3 * TC;
      out[c] = n;
      out[TC + c] = mean;
      out[2 * TC + c] = m2;
    }
  } else if ((!REVERSE || FUSE_BN) && A.partial) {
    float* wsum = reinterpret_cast<float*>(tile);  // [wavefront][32]: (sum 16 | second sum 16) of its channel half
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a1 = dof_row16_sum(s1[r]), a2 = dof_row16_sum(s2[r]);
      if (i == 0) {
        wsum[wv * 32 + kk * 4 + r] = a1;
        wsum[wv * 32 + 16 + kk * 4 + r] = a2;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * TC) {  // value v = which * 32 + channel; channel half ct = wavefronts ct and ct + 2
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31, h = c >> 4, cl = c & 15;
      A.partial[(int64_t)blockIdx.x * 2 * TC + threadIdx.x] = wsum[h * 32 + which * 16 + cl] + wsum[(h + 2) * 32 + which * 16 + cl];
    }
  }
}

#include "k_tcn_b3.inc.h"

// Generic variant for the 64-filter decoder TCN: KC input channels per tap, NC output channels, the whole
// (4 KC) x NC weight matrix staged ONCE per workgroup in LDS in B-operand order (64 KB for 64 x 64), so an
// MFMA's B value is one conflict-free ds_read_b32.  cin_real < KC (decoder block 0: 4L inputs) is zero-padded.
template <bool REVERSE, bool BN_IN, int KC, int NC>
__global__ void __launch_bounds__(256) k_tcn_convg(TcnConvArgs A, int cin_real, int w_ci) {
  constexpr int KS = KC / 4, NT = NC / 16;
  __shared__ float wl[TK * KS * NT * 64];
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int i = lane & 15, kk = lane >> 4;
  for (int e = threadIdx.x; e < TK * KS * NT * 64; e += blockDim.x) {
    const int ln = e & 63, ct = (e >> 6) % NT, ks = ((e >> 6) / NT) % KS, j = (e >> 6) / (NT * KS);
    const int ch = (ln >> 4) * KS + ks, n = ct * 16 + (ln & 15);
    // forward: W[o = n][c = ch][j] (c < cin_real); reverse: W[o = ch][c = n][j] (c = n < cin_real)
    float v;
    if (REVERSE) v = n < cin_real ? A.w[((int64_t)ch * w_ci + n) * TK + j] : 0.0f;
    else v = ch < cin_real ? A.w[((int64_t)n * w_ci + ch) * TK + j] : 0.0f;
    wl[e] = v;
  }
  __syncthreads();
  float sc[KS], sh[KS];
  if (BN_IN) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      sc[q] = BNP_SCALE(A.bnp_in, KC, kk * KS + q);
      sh[q] = BNP_SHIFT(A.bnp_in, KC, kk * KS + q);
    }
  }
  float bias[NT], s1[NT], s2[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    bias[ct] = (!REVERSE && A.bias) ? A.bias[ct * 16 + i] : 0.0f;
    s1[ct] = s2[ct] = 0.0f;
  }
  const int64_t tiles_per_t = A.Sp / 16;
  const int64_t n_tiles = (int64_t)A.T * tiles_per_t;
  for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
    const int t = (int)(tile / tiles_per_t);
    const int64_t s0 = (tile - (int64_t)t * tiles_per_t) * 16;
    dof_f32x4 acc[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const dof_f32x4 bv = {bias[ct], bias[ct], bias[ct], bias[ct]};
      acc[ct] = bv;
    }
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int tt = REVERSE ? t + (TK - 1 - j) * A.dil : t - (TK - 1 - j) * A.dil;
      if (tt >= 0 && tt < A.T) {
        float a[KS];
        dof_ld_row<KS>(A.in + ACT(tt, kk * KS, KC, A.Sp, s0 + i), a);
        if (BN_IN) {
#pragma unroll
          for (int q = 0; q < KS; ++q) a[q] = fmaxf(fmaf(a[q], sc[q], sh[q]), 0.0f);
          if (!REVERSE && j == TK - 1 && A.a_out && s0 + i < A.S) dof_st_row<KS>(A.a_out + ACT(t, kk * KS, KC, A.Sp, s0 + i), a);
        }
#pragma unroll
        for (int q = 0; q < KS; ++q)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wl[((j * KS + q) * NT + ct) * 64 + lane], acc[ct], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t s = s0 + kk * 4 + r;
      if (s < A.S) {
        float* o = A.out + ACT(t, 0, NC, A.Sp, s);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          float v = acc[ct][r];
          if (REVERSE && A.accumulate) v += o[ct * 16 + i];
          o[ct * 16 + i] = v;
          if (!REVERSE) {
            s1[ct] += v;
            s2[ct] = fmaf(v, v, s2[ct]);
          }
        }
      }
    }
  }
  if (!REVERSE && A.partial) {
    float* p = A.partial + (int64_t)wave * 2 * NC;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      s1[ct] += __shfl_xor(s1[ct], 16); s1[ct] += __shfl_xor(s1[ct], 32);
      s2[ct] += __shfl_xor(s2[ct], 16); s2[ct] += __shfl_xor(s2[ct], 32);
      if (kk == 0) {
        p[ct * 16 + i] = s1[ct];
        p[NC + ct * 16 + i] = s2[ct];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm bookkeeping (shared by the TCN layers, C = 32, and the head, C = 2L / L)
// ---------------------------------------------------------------------------------------------
// sums[2][C] = (sum x, sum (x - mean)^2) over `count` samples  ->  bnp; train: running buffers updated in place.
// (The second moment is taken about the mean in a second pass over the tensor: E[x^2] - mean^2 in fp32 loses
//  the variance to cancellation as soon as |mean| >> std, and the reference's two-pass variance does not.)
// Shifted single-pass statistics (shifted = 1; the time-resident convolution ran with stat_shift = the layer's running
// mean K, read here before it is updated -- no hidden state: the same parameters and inputs give the same bits).
// sums = (S1 = sum (y - K) | S2 = sum (y - K)^2 | M2 of the centred second pass).  mean = K + S1 / n and
// n var = S2 - S1^2 / n; the subtraction costs log2(S2 / (n var)) bits, so the one-pass value is only used while
// S1^2 / n <= S2 / 100 (|mean - K| <= 0.1 standard deviations -- where a fit spends its time once the running mean has
// caught up with the batch mean: the cancellation then takes < 0.02 bits and the accumulation error of S2 is that of
// the centred pass).  Otherwise -- the first steps of a fit, freshly loaded statistics -- k_tcn_var / k_tcn_var_sum,
// which evaluate the same predicate and return at once when it holds for all their channels, have left the two-pass
// M2 in sums[2C + c].
__device__ __forceinline__ bool bn_shift_ok(float s1, float s2, float count) { return s1 * (s1 / count) <= 0.01f * s2; }

__global__ void __launch_bounds__(64) k_bn_fwd_fin(const float* __restrict__ sums, float count,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                   int train, float* __restrict__ bnp, int C, int shifted) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (train && shifted) {
    const float s1 = sums[c], s2 = sums[C + c], d = s1 / count;
    mean = rmean[c] + d;
    var = bn_shift_ok(s1, s2, count) ? fmaxf(s2 - s1 * d, 0.0f) / count : sums[2 * C + c] / count;
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var * (count / fmaxf(count - 1.0f, 1.0f));
  } else if (train) {
    mean = sums[c] / count;
    var = sums[C + c] / count;
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var * (count / fmaxf(count - 1.0f, 1.0f));
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const float rstd = 1.0f / sqrtf(var + 1e-3f);
  const float scale = gamma[c] * rstd;
  BNP_MEAN(bnp, C, c) = mean;
  BNP_RSTD(bnp, C, c) = rstd;
  BNP_SCALE(bnp, C, c) = scale;
  BNP_SHIFT(bnp, C, c) = beta[c] - mean * scale;
}

// second pass of the batch statistics: channel sums of (y - mean)^2, mean = sums[c] / count; row-per-thread,
// 32-channel half per blockIdx.y; partial[h][nblk][32]
// shift != null (shifted statistics, see bn_shift_ok): the pass is only needed for channels whose one-pass variance is
// ill-conditioned -- the workgroup returns at once when none of its 32 channels is; mean = shift + S1 / count
__global__ void __launch_bounds__(256) k_tcn_var(const float* __restrict__ y, const float* __restrict__ sums, float count,
                                                 float* __restrict__ partial, int T, int CT, int64_t S, int64_t Sp,
                                                 const float* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * TC;
  if (shift) {
    __shared__ int need;
    if (threadIdx.x == 0) need = 0;
    __syncthreads();
    if (threadIdx.x < TC && !bn_shift_ok(sums[c0 + threadIdx.x], sums[CT + c0 + threadIdx.x], count)) need = 1;
    __syncthreads();
    if (!need) return;
  }
  float st[TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) st[c] = 0.0f;
  if (i < (int64_t)T * S) {
    const int t = (int)(i / S);
    const int64_t s = i - (int64_t)t * S;
    float v[TC];
    dof_ld_row<TC>(y + ACT(t, c0, CT, Sp, s), v);
    const float rc = 1.0f / count;
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const float dv = v[c] - (shift ? shift[c0 + c] + sums[c0 + c] * rc : sums[c0 + c] * rc);
      st[c] = dv * dv;
    }
  }
  dof_block_colsum<TC>(st, partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * TC);
}

// partial[h][nblk][32] -> sums[CT + h*32 + c]
// shifted (count > 0): channels with a well-conditioned one-pass variance are skipped, the result goes to sums[2 CT + ch]
__global__ void __launch_bounds__(256) k_tcn_var_sum(const float* __restrict__ partial, int64_t nblk, int CT,
                                                     float* __restrict__ sums, float shifted_count) {
  __shared__ float red[256];
  const int ch = blockIdx.x, h = ch / TC, c = ch - h * TC;
  if (shifted_count > 0.0f && bn_shift_ok(sums[ch], sums[CT + ch], shifted_count)) return;
  const float* src = partial + (int64_t)h * nblk * TC + c;
  float acc = 0.0f;
  for (int64_t b = threadIdx.x; b < nblk; b += 256) acc += src[b * TC];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[(shifted_count > 0.0f ? 2 * CT : CT) + ch] = red[0];
}

// the same for [c][Bp] head tensors: sums[C + c] = sum_b (h[c][b] - mean_c)^2 ; one workgroup per channel
__global__ void __launch_bounds__(256) k_head_var(const float* __restrict__ h, float* __restrict__ sums, float count, int C,
                                                  int64_t B, int64_t Bp) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const float mean = sums[c] / count;
  float acc = 0.0f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float dv = h[(int64_t)c * Bp + b] - mean;
    acc = fmaf(dv, dv, acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[C + c] = red[0];
}

// sums[2][C] = (sum g, sum g*xhat): gamma / beta gradients and the two batch means of the input gradient
__global__ void __launch_bounds__(64) k_bn_bwd_fin(const float* __restrict__ sums, float count,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                   int accumulate, float* __restrict__ coef, int C) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const float sg = sums[c], sgx = sums[C + c];
  dbeta[c] = accumulate ? dbeta[c] + sg : sg;
  dgamma[c] = accumulate ? dgamma[c] + sgx : sgx;
  coef[c] = sg / count;
  coef[C + c] = sgx / count;
}

// ---------------------------------------------------------------------------------------------
// Block tail: a2 = ReLU(BN2(y2)); res = previous block output (or the 1x1 conv of the raw input on block 0);
// out = ReLU(a2 + res); skip-sum += a2; on the last block the last-step features ReLU(skip-sum) leave in
// the [c][s] layout the CensNet kernels read.
// ---------------------------------------------------------------------------------------------
struct TcnCombineArgs {
  const float* y2;
  const float* bnp2;
  const float* res;    // [T][Sp][CT] previous block output, or null on block 0
  const float* xs;     // [T][Sp][xs_ch] raw input (block 0): F real channels in rows of xs_ch floats
  int xs_ch;
  const float *dsw, *dsb;  // (CT,F,1), (CT)
  float* out;          // [T][Sp][CT] or null (encoder's last block: unused by the reference)
  float* skip;         // [T][Sp][CT] running sum
  float* feat;         // [32][Sp] or null
  int first, T, F, CT;
  int skip_last;       // 1: only the last time step of the skip-sum is kept (the encoder reads nothing else of it)
  int t0;              // first time step of the launch (the encoder's last block has no `out`: t0 = T - 1)
  uint32_t* mask_out = nullptr;  // CT = 32 with `out`: [T][Sp] words, bit c = out[t][s][c] > 0 (what k_tcn_conv_t TAIL reads)
  int64_t S, Sp;
};

// one 16-byte word per lane: lanes along the channels of a row and on to the next rows (grid.y = time step)
__global__ void __launch_bounds__(256) k_tcn_combine(TcnCombineArgs A) {
  const int CT = A.CT, q = CT >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A.S * q) return;
  const int t = blockIdx.y + A.t0;
  const bool do_skip = !A.skip_last || t == A.T - 1;
  const int64_t s = e / q;
  const int c0 = (int)(e - s * q) * 4;
  const int64_t off = ACT(t, c0, CT, A.Sp, s);
  float y[4], r[4], sk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  dof_ld_row<4>(A.y2 + off, y);
  if (A.res) {
    dof_ld_row<4>(A.res + off, r);
  } else {
    float xin[64];
    for (int f = 0; f < A.F; ++f) xin[f] = A.xs[ACT(t, f, A.xs_ch, A.Sp, s)];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = A.dsb[c0 + c];
      for (int f = 0; f < A.F; ++f) acc = fmaf(A.dsw[(c0 + c) * A.F + f], xin[f], acc);
      r[c] = acc;
    }
  }
  if (!A.first && do_skip) dof_ld_row<4>(A.skip + off, sk);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float a2 = fmaxf(BN_APPLY(A.bnp2, CT, c0 + c, y[c]), 0.0f);
    sk[c] = A.first ? a2 : sk[c] + a2;
    r[c] = fmaxf(a2 + r[c], 0.0f);
  }
  if (do_skip) dof_st_row<4>(A.skip + off, sk);
  if (A.out) dof_st_row<4>(A.out + off, r);
  if (A.mask_out) {  // CT = 32: the eight lanes of a row
    const uint32_t wbits = tct_row_mask(r, c0 >> 2);
    if (c0 == 0) A.mask_out[(int64_t)t * A.Sp + s] = wbits;
  }
  if (A.feat && t == A.T - 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) A.feat[(int64_t)(c0 + c) * A.Sp + s] = fmaxf(sk[c], 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------
// Backward through ReLU + BatchNorm of one layer, pass 1: g = upstream * [a > 0], channel sums of g, g*xhat.
//   upstream = din (+ block-level terms when `blk`): din masked by the block output, plus the gradient of
//   the last-step features at t = T-1; the masked din is also the residual-branch gradient (stored to gres).
// ---------------------------------------------------------------------------------------------
struct TcnBnBwd1Args {
  const float* din;     // [T][Sp][CT] or null
  const float* y;       // pre-normalisation tensor of this layer
  const float* bnp;
  float* g;             // [T][Sp][CT] out
  float* partial;       // [halves][nblk][64]
  // block-level (BN2) extras
  const float* out_blk; // block output (mask of din) or null
  const float* dfeat;   // [32][Sp] gradient of the last-step features (encoder) or null
  const float* skip;    // final skip-sum (mask of dfeat)
  const float* dskip;   // [T][Sp][CT] gradient of the skip-sum at every step, already masked (decoder) or null
  float* gres;          // [T][Sp][CT] masked din (gradient entering the residual branch) or null
  int t0;               // first time step of the launch (the encoder's last block: T - 1, grid.y = 1)
  int blk, T, CT;
  int64_t S, Sp;
};

// One 16-byte word per lane (lanes along the channels of a row and on to the next rows, see k_tcn_bn_bwd2; the
// row-per-thread form it replaces ran at 4.2 TB/s): a workgroup owns 256 rows of one time step in CT / 4 passes; partial[nblk][2 CT] = (sum g | sum g xhat).
__global__ void __launch_bounds__(256) k_tcn_bn_bwd1_w(TcnBnBwd1Args A) {
  __shared__ float red[256][8];
  const int CT = A.CT, q = CT >> 2, rpp = 256 / q;
  const int sub = threadIdx.x / q, c0 = (threadIdx.x % q) * 4;
  const int t = blockIdx.y + A.t0;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
  float mean[4], rstd[4], sc[4], sh[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mean[c] = BNP_MEAN(A.bnp, CT, c0 + c);
    rstd[c] = BNP_RSTD(A.bnp, CT, c0 + c);
    sc[c] = BNP_SCALE(A.bnp, CT, c0 + c);
    sh[c] = BNP_SHIFT(A.bnp, CT, c0 + c);
  }
  for (int pass = 0; pass < q; ++pass) {
    const int64_t s = (int64_t)blockIdx.x * 256 + pass * rpp + sub;
    if (s >= A.S) continue;
    const int64_t off = ACT(t, c0, CT, A.Sp, s);
    float d[4] = {0.0f, 0.0f, 0.0f, 0.0f}, y[4];
    if (A.din) dof_ld_row<4>(A.din + off, d);
    if (A.blk) {
      if (A.out_blk) {
        float o[4];
        dof_ld_row<4>(A.out_blk + off, o);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = o[c] > 0.0f ? d[c] : 0.0f;
      }
      if (A.gres) dof_st_row<4>(A.gres + off, d);
      if (A.dfeat && t == A.T - 1) {
        float sk[4];
        dof_ld_row<4>(A.skip + off, sk);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] += sk[c] > 0.0f ? A.dfeat[(int64_t)(c0 + c) * A.Sp + s] : 0.0f;
      }
      if (A.dskip) {
        float ds[4];
        dof_ld_row<4>(A.dskip + off, ds);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] += ds[c];
      }
    }
    dof_ld_row<4>(A.y + off, y);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float g = fmaf(y[c], sc[c], sh[c]) > 0.0f ? d[c] : 0.0f;
      d[c] = g;
      acc[c] += g;
      acc[4 + c] = fmaf(g, (y[c] - mean[c]) * rstd[c], acc[4 + c]);
    }
    dof_st_row<4>(A.g + off, d);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) red[threadIdx.x][c] = acc[c];
  __syncthreads();
  if ((int)threadIdx.x < 2 * CT) {
    const int which = threadIdx.x / CT, ch = threadIdx.x - which * CT;
    float sum = 0.0f;
    for (int k = ch >> 2; k < 256; k += q) sum += red[k][which * 4 + (ch & 3)];
    A.partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * CT + threadIdx.x] = sum;
  }
}

// pass 2: dy = scale * (g - mean(g) - xhat * mean(g*xhat)), in place
// One 16-byte word per lane, lanes along the channels of a row and on to the next rows: a wavefront's load covers
// 1 KB of contiguous memory (the row-per-thread form touched 64 different 128-byte lines per instruction).
__global__ void __launch_bounds__(256) k_tcn_bn_bwd2(float* __restrict__ g, const float* __restrict__ y,
                                                     const float* __restrict__ bnp, const float* __restrict__ coef,
                                                     int T, int CT, int64_t S, int64_t Sp) {
  const int q = CT >> 2;  // 16-byte words per row (8 or 16: divides the block size, so a lane's channels are fixed)
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * q) return;
  const int t = blockIdx.y;
  const int64_t s = e / q;
  const int c0 = (int)(e - s * q) * 4;
  float d[4], yv[4];
  float* gp = g + ACT(t, c0, CT, Sp, s);
  dof_ld_row<4>(gp, d);
  dof_ld_row<4>(y + ACT(t, c0, CT, Sp, s), yv);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float xh = (yv[c] - BNP_MEAN(bnp, CT, c0 + c)) * BNP_RSTD(bnp, CT, c0 + c);
    d[c] = BNP_SCALE(bnp, CT, c0 + c) * (d[c] - coef[c0 + c] - xh * coef[CT + c0 + c]);
  }
  dof_st_row<4>(gp, d);
}

// ---------------------------------------------------------------------------------------------
// Head: [c][Bp] tensors, one thread per window
// ---------------------------------------------------------------------------------------------
// hn = x / max(rms(x), 1), clamped to +-1e4; rinv[b] = 1 / max(rms, 1); flag[b] = rms > 1
__global__ void __launch_bounds__(256) k_head_rms(const float* __restrict__ flat, float* __restrict__ hn,
                                                  float* __restrict__ rinv, int J, int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float ss = 0.0f;
  for (int j = 0; j < J; ++j) {
    const float v = flat[(int64_t)j * Bp + b];
    ss = fmaf(v, v, ss);
  }
  const float rms = sqrtf(ss / (float)J);
  const float r = 1.0f / fmaxf(rms, 1.0f);
  rinv[b] = rms > 1.0f ? r : -1.0f;  // sign = "the scale is constant" marker for the backward pass
  // clamp(+-1e4), then nan_to_num(nan = 0) (models_new.py:653-655, 1154-1156): a window with a NaN feature -- e.g. a
  // sequence whose every key is masked in the transformer cores -- has a NaN rms, so its whole row becomes zeros
  // (fmaxf / fminf alone would turn the NaNs into -1e4)
  const bool poisoned = rms != rms;   // (fmaxf(NaN, 1) is 1: the reference divides by the NaN and every entry goes to 0)
  for (int j = 0; j < J; ++j) {
    const float v = flat[(int64_t)j * Bp + b] * r;
    hn[(int64_t)j * Bp + b] = (poisoned || v != v) ? 0.0f : fminf(fmaxf(v, -1e4f), 1e4f);
  }
}

// out[o][b] = act(bias[o] + sum_i W[o][i] * bn(in[i][b])); thread = (b, o = blockIdx.y); optional channel sums
__global__ void __launch_bounds__(256) k_head_dense(const float* __restrict__ in, const float* __restrict__ bnp_in,
                                                    float* __restrict__ in_norm, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                    float* __restrict__ partial, int CI, int relu, int64_t B,
                                                    int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  float st[2] = {0.0f, 0.0f};
  if (b < B) {
    const dof_cfp wr = dof_cw(w) + (int64_t)o * CI;
    float acc = dof_cw(bias)[o];
    for (int i = 0; i < CI; ++i) {
      float v = in[(int64_t)i * Bp + b];
      if (bnp_in) {
        v = BN_APPLY(bnp_in, CI, i, v);
        if (o == 0) in_norm[(int64_t)i * Bp + b] = v;
      }
      acc = fmaf(wr[i], v, acc);
    }
    if (relu) acc = fmaxf(acc, 0.0f);
    out[(int64_t)o * Bp + b] = acc;
    st[0] = acc;
    st[1] = acc * acc;
  }
  if (partial) dof_block_colsum<2>(st, partial + ((int64_t)o * gridDim.x + blockIdx.x) * 2);
}

// partial[(o*nblk + blk)*2 + k] -> sums[k*C + o]   (fixed order)
__global__ void __launch_bounds__(64) k_head_sum(const float* __restrict__ partial, int nblk, float* __restrict__ sums,
                                                 int C) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float a = 0.0f, q = 0.0f;
  for (int k = 0; k < nblk; ++k) {
    a += partial[((int64_t)c * nblk + k) * 2];
    q += partial[((int64_t)c * nblk + k) * 2 + 1];
  }
  sums[c] = a;
  sums[C + c] = q;
}

// din[i][b] = sum_o W[o][i] dout[o][b]; thread = (b, i = blockIdx.y)
__global__ void __launch_bounds__(256) k_head_dense_bwd(const float* __restrict__ dout, const float* __restrict__ w,
                                                        float* __restrict__ din, int CI, int CO, int64_t B,
                                                        int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int i = blockIdx.y;
  float acc = 0.0f;
  for (int o = 0; o < CO; ++o) acc = fmaf(dof_cw(w)[(int64_t)o * CI + i], dout[(int64_t)o * Bp + b], acc);
  din[(int64_t)i * Bp + b] = acc;
}

// BatchNorm backward over the batch, pass 1: sums of g and g*xhat (h = BN input = post-ReLU activation)
__global__ void __launch_bounds__(256) k_head_bn_bwd1(const float* __restrict__ g, const float* __restrict__ h,
                                                      const float* __restrict__ bnp, float* __restrict__ partial,
                                                      int C, int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  float st[2] = {0.0f, 0.0f};
  if (b < B) {
    const float gv = g[(int64_t)c * Bp + b];
    st[0] = gv;
    st[1] = gv * (h[(int64_t)c * Bp + b] - BNP_MEAN(bnp, C, c)) * BNP_RSTD(bnp, C, c);
  }
  dof_block_colsum<2>(st, partial + ((int64_t)c * gridDim.x + blockIdx.x) * 2);
}

// pass 2 + ReLU of the producing Linear: dpre = scale*(g - c1 - xhat*c2) * [h > 0]
__global__ void __launch_bounds__(256) k_head_bn_bwd2(const float* __restrict__ g, const float* __restrict__ h,
                                                      const float* __restrict__ bnp, const float* __restrict__ coef,
                                                      float* __restrict__ dpre, int C, int relu, int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int c = blockIdx.y;
  const float hv = h[(int64_t)c * Bp + b];
  const float xh = (hv - BNP_MEAN(bnp, C, c)) * BNP_RSTD(bnp, C, c);
  const float d = BNP_SCALE(bnp, C, c) * (g[(int64_t)c * Bp + b] - coef[c] - xh * coef[C + c]);
  dpre[(int64_t)c * Bp + b] = (!relu || hv > 0.0f) ? d : 0.0f;
}

// backward of hn = x * r(x): dflat = r * (dhn - hn * (dhn . hn) / J) when rms > 1, else dhn
__global__ void __launch_bounds__(256) k_head_rms_bwd(const float* __restrict__ dhn, const float* __restrict__ hn,
                                                      const float* __restrict__ rinv, float* __restrict__ dflat, int J,
                                                      int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float r = rinv[b];
  if (r < 0.0f) {
    for (int j = 0; j < J; ++j) dflat[(int64_t)j * Bp + b] = dhn[(int64_t)j * Bp + b];
    return;
  }
  float dot = 0.0f;
  for (int j = 0; j < J; ++j) dot = fmaf(dhn[(int64_t)j * Bp + b], hn[(int64_t)j * Bp + b], dot);
  dot /= (float)J;
  for (int j = 0; j < J; ++j)
    dflat[(int64_t)j * Bp + b] = r * (dhn[(int64_t)j * Bp + b] - hn[(int64_t)j * Bp + b] * dot);
}

// ---------------------------------------------------------------------------------------------
// TCN decoder ends (models_new.py:713-819): repeat the normalised latent features over time; output head
// ---------------------------------------------------------------------------------------------
// zrep[t][b][c] = BN2(d2)[c][b] for every t (channels c >= C4 are zero padding up to 32)
template <int ZC>  // channels per zrep row: 32 (4 L <= 32) or 64 (latent 16)
__global__ void __launch_bounds__(256) k_dec_repeat(const float* __restrict__ d2, const float* __restrict__ bnp,
                                                    float* __restrict__ zrep, int C4, int T, int64_t B, int64_t Bp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * B) return;
  const int t = (int)(i / B);
  const int64_t b = i - (int64_t)t * B;
  float row[ZC];
#pragma unroll
  for (int c = 0; c < ZC; ++c)
    row[c] = c < C4 ? BN_APPLY(bnp, C4, c, d2[(int64_t)c * Bp + b]) : 0.0f;
  dof_st_row<ZC>(zrep + ACT(t, 0, ZC, Bp, b), row);
}

// dzf[c][b] = sum_t dzrep[t][b][c]
template <int ZC>
__global__ void __launch_bounds__(256) k_dec_sum_time(const float* __restrict__ dzrep, float* __restrict__ dzf, int C4,
                                                      int T, int64_t B, int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc[ZC];
#pragma unroll
  for (int c = 0; c < ZC; ++c) acc[c] = 0.0f;
  for (int t = 0; t < T; ++t) {
    float row[ZC];
    dof_ld_row<ZC>(dzrep + ACT(t, 0, ZC, Bp, b), row);
#pragma unroll
    for (int c = 0; c < ZC; ++c) acc[c] += row[c];
  }
#pragma unroll
  for (int c = 0; c < ZC; ++c)
    if (c < C4) dzf[(int64_t)c * Bp + b] = acc[c];
}

// hidden = ReLU(skip-sum) at every step -> loc = Linear(64 -> 3N) -> unit-variance Normal log-prob on valid
// frames (NaN on masked ones, Q3) and its gradient back to the skip-sum.  Same conventions as k_dec_tail.
struct TcnDecOutArgs {
  const float* skip;   // [T][Bp][64]
  const float *wp, *bp;  // (C3,64), (C3)
  const float* x;      // (B,T,C3)
  const float* valid;  // [T][Bp]
  float* hid;          // [T][Bp][64]
  float* loc_out;      // (B,T,C3) or null
  float* recon_partial;
  float* dloc;         // [T][Bp][C3]
  float* dskip;        // [T][Bp][64]
  int T, C3, train;
  int64_t B, Bp;
};

__global__ void __launch_bounds__(256) k_tcn_dec_out(TcnDecOutArgs A) {
  constexpr int CH = 64;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float nll[1] = {0.0f};
  if (i < (int64_t)A.T * A.B) {
    const int t = (int)(i / A.B);
    const int64_t b = i - (int64_t)t * A.B;
    const dof_cfp wp = dof_cw(A.wp), bp = dof_cw(A.bp);
    float h[CH], dh[CH];
    dof_ld_row<CH>(A.skip + ACT(t, 0, CH, A.Bp, b), h);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      h[c] = fmaxf(h[c], 0.0f);
      dh[c] = 0.0f;
    }
    if (A.train) dof_st_row<CH>(A.hid + ACT(t, 0, CH, A.Bp, b), h);
    const bool ok = A.valid[(int64_t)t * A.Bp + b] != 0.0f;
    const float inv_bt = 1.0f / ((float)A.B * (float)A.T);
    const float* __restrict__ xr = A.x + (b * A.T + t) * A.C3;
    float sq = 0.0f;
    for (int j = 0; j < A.C3; ++j) {
      float loc = bp[j];
#pragma unroll
      for (int c = 0; c < CH; ++c) loc = fmaf(wp[j * CH + c], h[c], loc);
      if (loc != loc) loc = 0.0f;
      loc = fminf(fmaxf(loc, -1e6f), 1e6f);
      if (A.loc_out) A.loc_out[(b * A.T + t) * A.C3 + j] = loc;
      const float df = xr[j] - loc;
      sq = fmaf(df, df, sq);
      if (A.train) {
        const float dl = ok ? -df * inv_bt : NAN;
        A.dloc[ACT(t, j, A.C3, A.Bp, b)] = dl;
#pragma unroll
        for (int c = 0; c < CH; ++c) dh[c] = fmaf(wp[j * CH + c], dl, dh[c]);
      }
    }
    const float LOG_2PI = 1.8378770664093453f;
    nll[0] = ok ? 0.5f * sq + 0.5f * (float)A.C3 * LOG_2PI : NAN;
    if (A.train) {
#pragma unroll
      for (int c = 0; c < CH; ++c) dh[c] = h[c] > 0.0f ? dh[c] : 0.0f;
      dof_st_row<CH>(A.dskip + ACT(t, 0, CH, A.Bp, b), dh);
    }
  }
  dof_block_colsum<1>(nll, A.recon_partial + blockIdx.x);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int64_t dof_tcn_row_blocks(int T, int64_t S) { return dof_cdiv((int64_t)T * S, 256); }
int64_t dof_tcn_conv_waves(int T, int64_t Sp) {
  const int64_t tiles = (int64_t)T * (Sp / 16);
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return blocks * 4;
}

int dof_launch_tcn_in_conv(int F, const float* xin, const float* w, const float* bias, float* xs, float* y,
                           float* partial, int T, int G, int64_t S, int64_t Sp, int dil, hipStream_t st, int records) {
  const unsigned nb = (unsigned)dof_tcn_row_blocks(T, S);
  if (F == 3) {
    DOF_LAUNCH((k_tcn_in_conv<3>), (nb), (256), st, xin, w, bias, xs, y, partial, T, G, S, Sp, dil, records);
  } else if (F == 1) {
    DOF_LAUNCH((k_tcn_in_conv<1>), (nb), (256), st, xin, w, bias, xs, y, partial, T, G, S, Sp, dil, records);
  } else {
    dof_set_error("features per group %d not supported (3 or 1)", F);
    return DOF_ERR_UNSUPPORTED;
  }
  return dof_check_launch("k_tcn_in_conv");
}

// Workgroups of the time-resident convolution: one per 16 sequences (T <= 25) or per 8 sequences (T <= 50, round 4), at most
// three resident per CU (52 KB of LDS each).  DOF_TCN_RESIDENT_MAX_T=25 restores round 3's limit (A/B measurements: windows
// of 26 .. 50 steps then take k_tcn_conv's four fetches per input row and none of the folds).
static int tct_max_t() {
  static const int v = [] {
    const char* e = getenv("DOF_TCN_RESIDENT_MAX_T");
    const int m = e ? atoi(e) : 2 * TCT_T;
    return m < 2 * TCT_T ? (m < TCT_T ? 0 : TCT_T) : 2 * TCT_T;
  }();
  return v;
}
static bool tct_fits(int T, int64_t Sp) { return T <= tct_max_t() && (int64_t)T * Sp * TC < ((int64_t)1 << 31); }
// Round 6: the time-resident convolutions on the bf16 matrix pipe (k_tcn_conv_b, three exact pieces per operand), 8 / 4
// sequences per workgroup, two workgroups per CU.  DOF_TCN_CONV_B3=0: round 5's fp32-MFMA kernels (A/B measurements).
static int tcn_conv_b3() {
  static const int on = [] {
    const char* e = getenv("DOF_TCN_CONV_B3");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return on;
}
static int tct_ns(int T) { return tcn_conv_b3() ? (T <= TCT_T ? 8 : 4) : (T <= TCT_T ? 16 : 8); }
static unsigned tct_blocks(int T, int64_t Sp) {
  const int64_t groups = Sp / tct_ns(T), cap = tcn_conv_b3() ? 256 : 768;   // (k_tcn_conv_b: one 512-thread workgroup per CU)
  return (unsigned)(groups < cap ? groups : cap);
}
// ... with the convolution's weight gradient accumulated by the two fused data-gradient variants of the backward pass
// (DOF_TCN_WGRAD_FUSED=0: k_tcn_wgrad_b3 as in round 5)
int dof_tcn_wgrad_fused(int T, int64_t Sp) {
  static const int on = [] {
    const char* e = getenv("DOF_TCN_WGRAD_FUSED");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return (on && tcn_conv_b3() && tct_fits(T, Sp)) ? 1 : 0;
}
// k_tcn_conv_t / k_tcn_conv_b <R, BN, FUSE, BWD2, TAIL, COMB> at the group size T asks for
#define TCT_LAUNCH(R, BN, FUSE, BWD2, TAIL, COMB)                                                         \
  do {                                                                                                    \
    if (tcn_conv_b3()) {                                                                                  \
      if (tct_ns(A.T) == 8) DOF_LAUNCH((k_tcn_conv_b<R, BN, FUSE, BWD2, TAIL, COMB, false, 8>), (nbt), (512), st, A); \
      else DOF_LAUNCH((k_tcn_conv_b<R, BN, FUSE, BWD2, TAIL, COMB, false, 4>), (nbt), (512), st, A);       \
    } else if (tct_ns(A.T) == 16) DOF_LAUNCH((k_tcn_conv_t<R, BN, FUSE, BWD2, TAIL, COMB, 16>), (nbt), (256), st, A); \
    else DOF_LAUNCH((k_tcn_conv_t<R, BN, FUSE, BWD2, TAIL, COMB, 8>), (nbt), (256), st, A);                \
  } while (0)
// the fused data-gradient + weight-gradient variants (A.wg_partials set)
// (NCW = 4 compute wavefronts everywhere.  Eight -- 768 threads, two compute wavefronts per SIMD -- were built and measured:
//  the variants without the weight gradient gained 7 - 11 % in isolation (forward BN_IN 204 -> 181 us) and nothing in the C4 step
//  (50.8 against 50.2 ms); the fused variants spill 67 - 158 registers under the 168-register cap -- conv2's, put on a register
//  diet (one operand set, one accumulator: still 20 spilled), ran 524 us against 387.  EXPERIMENTS.md, round 6.)
#define TCT_LAUNCH_WG(TAIL)                                                                               \
  do {                                                                                                    \
    if (tct_ns(A.T) == 8) DOF_LAUNCH((k_tcn_conv_b<true, false, true, true, TAIL, false, true, 8>), (nbt), (512), st, A); \
    else DOF_LAUNCH((k_tcn_conv_b<true, false, true, true, TAIL, false, true, 4>), (nbt), (512), st, A);   \
  } while (0)
// One-pass (shifted) BatchNorm statistics (bn_shift_ok) of the time-resident convolutions: round 2's default, opt-in in
// rounds 3 / 4 (DOF_TCN_ONEPASS=1), no longer selectable since round 5 (the mergeable records below are the default and
// pass the same goldens; the kernels keep the shift argument, always null).  The reason it lost the default: measured elementwise against a reference golden whose running means equal the batch means
// (tests/golden/vade_tcn14_onepass.npz, every channel on the one-pass form) the gradients of 107 of 200 tensors leave the
// standard bar, by up to 4.1 x (2e-3 of the tensor scale), while the centred second pass stays within 0.1 x of it on the
// same fixture -- the refreshed running variances agree with the reference to 1e-7 either way, the loss terms to 1.5e-7.
// The default path must meet the parity bar; the 4 % of the C4 step the second pass costs is the price.
int dof_tcn_onepass_stats() { return 0; }
int dof_tcn_conv32_resident(int T, int64_t Sp) { return tct_fits(T, Sp) ? 1 : 0; }
// Mergeable one-pass statistics of the time-resident forward convolutions (default; DOF_TCN_STAT_RECORDS=0: sum pass +
// centred second pass over the tensor)
int dof_tcn_stat_records() {
  static const int on = [] {
    const char* e = getenv("DOF_TCN_STAT_RECORDS");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return on;
}
int dof_launch_tcn_stat_merge(const float* partial, int64_t nblk, float* sums, hipStream_t st) {
  StatFinArgs F = {};
  DOF_LAUNCH(k_tcn_stat_merge, (TC), (256), st, partial, (int)nblk, sums, F);
  return dof_check_launch("k_tcn_stat_merge");
}
// ... and the layer's BatchNorm record + running buffers (train mode) in the same launch
int dof_launch_tcn_stat_merge_fin(const float* partial, int64_t nblk, float* sums, float count, const float* gamma,
                                  const float* beta, float* rmean, float* rvar, float momentum, float* bnp, hipStream_t st) {
  StatFinArgs F;
  F.count = count; F.gamma = gamma; F.beta = beta; F.rmean = rmean; F.rvar = rvar; F.momentum = momentum; F.bnp = bnp;
  DOF_LAUNCH(k_tcn_stat_merge, (TC), (256), st, partial, (int)nblk, sums, F);
  return dof_check_launch("k_tcn_stat_merge_fin");
}
int dof_launch_bn_bwd_sum_fin(const float* partial, int64_t nblk, float* sums, float count, float* dgamma, float* dbeta,
                              int accumulate, float* coef, hipStream_t st, bool frozen) {
  if (frozen) count = __builtin_inff();   // coefficients = sums / count = 0: eval-mode backward (launchers.h)
  DOF_LAUNCH(k_bn_bwd_sum_fin, (TC), (256), st, partial, nblk, count, dgamma, dbeta, accumulate, coef, sums);
  return dof_check_launch("k_bn_bwd_sum_fin");
}
int64_t dof_tcn_bn_bwd1_blocks(int T, int64_t S) { return (int64_t)dof_cdiv(S, 256) * T; }
// The encoder's last block: BatchNorm2's backward pass 1 on the last time step only, residual-branch gradient = a zero tensor
// (tcn_encoder_backward).  DOF_TCN_LAST_BLOCK_SPARSE=0: the full-size pass of rounds 2 - 5.
int dof_tcn_last_block_sparse() {
  static const int on = [] {
    const char* e = getenv("DOF_TCN_LAST_BLOCK_SPARSE");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return on;
}
int64_t dof_tcn_conv32_partials(int T, int64_t Sp) {
  return dof_tcn_conv32_resident(T, Sp) ? (int64_t)tct_blocks(T, Sp) : dof_tcn_conv_waves(T, Sp);
}

int dof_launch_tcn_conv(int reverse, const float* in, const float* w, const float* bias, const float* bnp_in,
                        float* a_out, float* out, float* partial, int accumulate, int T, int dil, int64_t S, int64_t Sp,
                        hipStream_t st, const float* bwd_y, const float* bwd_bnp, const float* bwd_coef,
                        const float* stat_shift, int bwd_store, int stat_records) {
  TcnConvArgs A;
  A.bwd_store = bwd_store;
  A.stat_records = (stat_records && !reverse && dof_tcn_conv32_resident(T, Sp)) ? 1 : 0;
  A.in = in; A.w = w; A.bias = bias; A.bnp_in = bnp_in; A.a_out = a_out; A.out = out; A.partial = partial;
  A.fuse_y = nullptr; A.fuse_bnp = nullptr;
  A.bwd_y = bwd_y; A.bwd_bnp = bwd_bnp; A.bwd_coef = bwd_coef;
  A.stat_shift = stat_shift;
  if (stat_shift && (reverse || !dof_tcn_conv32_resident(T, Sp))) {
    dof_set_error("k_tcn_conv: shifted channel sums need the forward time-resident kernel");
    return DOF_ERR_UNSUPPORTED;
  }
  A.T = T; A.dil = dil; A.accumulate = accumulate; A.S = S; A.Sp = Sp;
  if (dof_tcn_conv32_resident(T, Sp)) {
    const unsigned nbt = tct_blocks(T, Sp);
    if (reverse && bwd_y) {
      TCT_LAUNCH(true, false, false, true, false, false);
    } else if (reverse) {
      TCT_LAUNCH(true, false, false, false, false, false);
    } else if (bnp_in) {
      TCT_LAUNCH(false, true, false, false, false, false);
    } else {
      TCT_LAUNCH(false, false, false, false, false, false);
    }
    return dof_check_launch("k_tcn_conv_t");
  }
  if (bwd_y) {
    dof_set_error("k_tcn_conv: the fused BatchNorm-backward pass needs the time-resident kernel (T <= %d)", tct_max_t());
    return DOF_ERR_UNSUPPORTED;
  }
  const unsigned nb = (unsigned)(dof_tcn_conv_waves(T, Sp) / 4);
  if (reverse) {
    DOF_LAUNCH((k_tcn_conv<true, false>), (nb), (256), st, A);
  } else if (bnp_in) {
    DOF_LAUNCH((k_tcn_conv<false, true>), (nb), (256), st, A);
  } else {
    DOF_LAUNCH((k_tcn_conv<false, false>), (nb), (256), st, A);
  }
  return dof_check_launch("k_tcn_conv");
}

// data gradient of a 32 -> 32 convolution fused with the first backward pass of the BatchNorm + ReLU that produced
// the convolution's input: g = conv^T(dy) * [BN(y) > 0] into g_out, channel sums (sum g | sum g * xhat) into sums
// bwd_y != null (time-resident kernel only): dy still holds the pass-1 gradient of ITS BatchNorm; pass 2 is applied
// while staging and written back in place.
int dof_launch_tcn_conv_bwd_bn(const float* dy, const float* w, const float* y, const float* bnp, float* g_out,
                               float* partial, float* sums, int T, int dil, int64_t S, int64_t Sp, hipStream_t st,
                               const float* bwd_y, const float* bwd_bnp, const float* bwd_coef, int bwd_store,
                               float* wg_partials, int64_t wg_part0, int64_t wg_part1) {
  TcnConvArgs A;
  A.bwd_store = bwd_store;
  if (wg_partials && !(bwd_y && dof_tcn_wgrad_fused(T, Sp))) {
    dof_set_error("k_tcn_conv_bwd_bn: the fused weight gradient needs k_tcn_conv_b and a lazy BatchNorm2 gradient");
    return DOF_ERR_UNSUPPORTED;
  }
  A.wg_partials = wg_partials; A.wg_part0 = wg_part0; A.wg_part1 = wg_part1;
  A.in = dy; A.w = w; A.bias = nullptr; A.bnp_in = nullptr; A.a_out = nullptr; A.out = g_out; A.partial = partial;
  A.fuse_y = y; A.fuse_bnp = bnp;
  A.bwd_y = bwd_y; A.bwd_bnp = bwd_bnp; A.bwd_coef = bwd_coef;
  A.stat_shift = nullptr;
  A.T = T; A.dil = dil; A.accumulate = 0; A.S = S; A.Sp = Sp;
  if (dof_tcn_conv32_resident(T, Sp)) {
    const unsigned nbt = tct_blocks(T, Sp);
    if (wg_partials) {
      TCT_LAUNCH_WG(false);
    } else if (bwd_y) {
      TCT_LAUNCH(true, false, true, true, false, false);
    } else {
      TCT_LAUNCH(true, false, true, false, false, false);
    }
    if (int rc = dof_check_launch("k_tcn_conv_t_bwd_bn")) return rc;
    return sums ? dof_launch_sum_partials(partial, (int64_t)nbt, 2 * TC, sums, 0, st) : DOF_OK;
  }
  if (bwd_y) {
    dof_set_error("k_tcn_conv_bwd_bn: the fused BatchNorm-backward pass needs the time-resident kernel (T <= %d)", tct_max_t());
    return DOF_ERR_UNSUPPORTED;
  }
  const int64_t waves = dof_tcn_conv_waves(T, Sp);
  DOF_LAUNCH((k_tcn_conv<true, false, true>), ((unsigned)(waves / 4)), (256), st, A);
  if (int rc = dof_check_launch("k_tcn_conv_bwd_bn")) return rc;
  return sums ? dof_launch_sum_partials(partial, waves, 2 * TC, sums, 0, st) : DOF_OK;  // null: the caller's k_bn_bwd_sum_fin reduces them
}

// forward conv1 of block b + 1 with block b's tail (out = ReLU(ReLU(BN2(y2)) + res), written to out_blk) computed while staging
int dof_tcn_combine_fold() {
  static const int on = [] {
    const char* e = getenv("DOF_TCN_COMBINE_FOLD");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return on;
}
int dof_launch_tcn_conv_comb(const float* res, const float* y2, const float* bnp2, float* out_blk, const float* w,
                             const float* bias, float* out, float* partial, int T, int dil, int64_t S, int64_t Sp,
                             hipStream_t st, const float* stat_shift, int stat_records, float* relu_mask_out) {
  if (!dof_tcn_conv32_resident(T, Sp)) {
    dof_set_error("k_tcn_conv_comb: needs the time-resident kernel (T <= %d)", tct_max_t());
    return DOF_ERR_UNSUPPORTED;
  }
  TcnConvArgs A;
  A.bwd_store = 0;
  A.in = res; A.w = w; A.bias = bias; A.bnp_in = bnp2; A.a_out = out_blk; A.out = out; A.partial = partial;
  A.fuse_y = nullptr; A.fuse_bnp = nullptr;
  A.bwd_y = y2; A.bwd_bnp = nullptr; A.bwd_coef = nullptr;
  A.relu_mask_out = reinterpret_cast<uint32_t*>(relu_mask_out);
  A.stat_shift = stat_shift;
  A.stat_records = stat_records ? 1 : 0;
  A.T = T; A.dil = dil; A.accumulate = 0; A.S = S; A.Sp = Sp;
  const unsigned nbt = tct_blocks(T, Sp);
  TCT_LAUNCH(false, false, false, false, false, true);
  return dof_check_launch("k_tcn_conv_t_comb");
}

// ... of block 1: block 0's residual is the 1 x 1 convolution of the raw input, computed from xs while staging (k_tcn_conv_b
// only; k_tcn_combine's own arithmetic: bias, then one fmaf per input channel in order)
int dof_tcn_combine_fold0() { return (dof_tcn_combine_fold() && tcn_conv_b3()) ? 1 : 0; }
int dof_launch_tcn_conv_comb0(const float* xs, int F, const float* dsw, const float* dsb, const float* y2, const float* bnp2,
                              float* out_blk, const float* w, const float* bias, float* out, float* partial, int T, int dil,
                              int64_t S, int64_t Sp, hipStream_t st, int stat_records, float* relu_mask_out) {
  if (!dof_tcn_conv32_resident(T, Sp) || !tcn_conv_b3() || F < 1 || F > 3) {
    dof_set_error("k_tcn_conv_comb0: needs the time-resident bf16-piece kernel and 1 .. 3 input channels (T %d, F %d)", T, F);
    return DOF_ERR_UNSUPPORTED;
  }
  TcnConvArgs A;
  A.bwd_store = 0;
  A.in = xs; A.w = w; A.bias = bias; A.bnp_in = bnp2; A.a_out = out_blk; A.out = out; A.partial = partial;
  A.fuse_y = nullptr; A.fuse_bnp = nullptr;
  A.bwd_y = y2; A.bwd_bnp = nullptr; A.bwd_coef = nullptr;
  A.relu_mask_out = reinterpret_cast<uint32_t*>(relu_mask_out);
  A.stat_shift = nullptr;
  A.stat_records = stat_records ? 1 : 0;
  A.ds_w = dsw; A.ds_b = dsb; A.ds_F = F;
  A.T = T; A.dil = dil; A.accumulate = 0; A.S = S; A.Sp = Sp;
  const unsigned nbt = tct_blocks(T, Sp);
  if (tct_ns(A.T) == 8) DOF_LAUNCH((k_tcn_conv_b<false, false, false, false, false, true, false, 8, 4, true>), (nbt), (512), st, A);
  else DOF_LAUNCH((k_tcn_conv_b<false, false, false, false, false, true, false, 4, 4, true>), (nbt), (512), st, A);
  return dof_check_launch("k_tcn_conv_b_comb0");
}

// conv1's data gradient of block b + 1 with the backward of block b's tail and the first pass of block b's BatchNorm2
// backward in its epilogue (k_tcn_conv_t TAIL): dy = pass-1 gradient of block b + 1's BatchNorm1 (pass 2 applied while
// staging), tail_src = the gradient waiting at block b's output, g_out = block b's g2, sums = its channel sums
int dof_tcn_tail_fold() {
  static const int on = [] {
    const char* e = getenv("DOF_TCN_TAIL_FOLD");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  return on;
}
int dof_launch_tcn_conv_tail(const float* dy, const float* w, const float* bwd_y, const float* bwd_bnp, const float* bwd_coef,
                             int bwd_store, const float* tail_src, const float* tail_mask, float* tail_gres,
                             const float* tail_skip, const float* tail_dfeat, const float* y2, const float* bnp2, float* g_out,
                             float* partial, float* sums, int T, int dil, int64_t S, int64_t Sp, hipStream_t st,
                             const float* wg_x, float* wg_partials, int64_t wg_part0, int64_t wg_part1) {
  if (wg_partials && !(wg_x && dof_tcn_wgrad_fused(T, Sp))) {
    dof_set_error("k_tcn_conv_tail: the fused weight gradient needs k_tcn_conv_b and the convolution's input tensor");
    return DOF_ERR_UNSUPPORTED;
  }
  if (!dof_tcn_conv32_resident(T, Sp) || !bwd_y || !(tail_mask || wg_partials)) {
    dof_set_error("k_tcn_conv_tail: needs the time-resident kernel (T <= %d), a lazy BatchNorm1 gradient and the block output's mask words", tct_max_t());
    return DOF_ERR_UNSUPPORTED;
  }
  TcnConvArgs A;
  A.bwd_store = bwd_store;
  A.in = dy; A.w = w; A.bias = nullptr; A.bnp_in = nullptr; A.a_out = nullptr; A.out = g_out; A.partial = partial;
  A.fuse_y = y2; A.fuse_bnp = bnp2;
  A.bwd_y = bwd_y; A.bwd_bnp = bwd_bnp; A.bwd_coef = bwd_coef;
  A.stat_shift = nullptr;
  A.tail_src = tail_src; A.tail_gres = tail_gres; A.tail_skip = tail_skip; A.tail_dfeat = tail_dfeat;
  A.tail_mask = reinterpret_cast<const uint32_t*>(tail_mask);
  A.wg_x = wg_x; A.wg_partials = wg_partials; A.wg_part0 = wg_part0; A.wg_part1 = wg_part1;
  A.T = T; A.dil = dil; A.accumulate = 0; A.S = S; A.Sp = Sp;
  const unsigned nbt = tct_blocks(T, Sp);
  if (wg_partials) TCT_LAUNCH_WG(true);
  else TCT_LAUNCH(true, false, true, true, true, false);
  if (int rc = dof_check_launch("k_tcn_conv_t_tail")) return rc;
  return sums ? dof_launch_sum_partials(partial, (int64_t)nbt, 2 * TC, sums, 0, st) : DOF_OK;
}

int dof_launch_bn_fwd_fin(const float* sums, float count, const float* gamma, const float* beta, float* rmean,
                          float* rvar, float momentum, int train, float* bnp, int C, hipStream_t st, int shifted) {
  DOF_LAUNCH(k_bn_fwd_fin, (1), (64), st, sums, count, gamma, beta, rmean, rvar, momentum, train, bnp, C, shifted);
  return dof_check_launch("k_bn_fwd_fin");
}

int dof_launch_bn_bwd_fin(const float* sums, float count, float* dgamma, float* dbeta, int accumulate, float* coef,
                          int C, hipStream_t st, bool frozen) {
  if (frozen) count = __builtin_inff();
  DOF_LAUNCH(k_bn_bwd_fin, (1), (64), st, sums, count, dgamma, dbeta, accumulate, coef, C);
  return dof_check_launch("k_bn_bwd_fin");
}

int dof_launch_tcn_combine(const float* y2, const float* bnp2, const float* res, const float* xs, const float* dsw,
                           const float* dsb, float* out, float* skip, float* feat, int first, int T, int F, int CT,
                           int64_t S, int64_t Sp, hipStream_t st, int xs_ch, int skip_last, float* mask_out) {
  TcnCombineArgs A;
  A.mask_out = (out && CT == 32) ? reinterpret_cast<uint32_t*>(mask_out) : nullptr;
  A.skip_last = skip_last;
  A.t0 = (skip_last && !out) ? T - 1 : 0;
  A.xs_ch = xs_ch > 0 ? xs_ch : F;
  A.y2 = y2; A.bnp2 = bnp2; A.res = res; A.xs = xs; A.dsw = dsw; A.dsb = dsb; A.out = out; A.skip = skip;
  A.feat = feat; A.first = first; A.T = T; A.F = F; A.CT = CT; A.S = S; A.Sp = Sp;
  DOF_LAUNCH(k_tcn_combine, (dof_cdiv(S * (CT / 4), 256), (unsigned)(T - A.t0)), (256), st, A);
  return dof_check_launch("k_tcn_combine");
}

// pass 1 of a layer's BatchNorm backward + the reduction of its channel sums into sums[2][CT]
int dof_launch_tcn_bn_bwd1(const float* din, const float* y, const float* bnp, float* g, float* partial, float* sums,
                           int blk, const float* out_blk, const float* dfeat, const float* skip, const float* dskip,
                           float* gres, int T, int CT, int64_t S, int64_t Sp, hipStream_t st, int last_step_only) {
  TcnBnBwd1Args A;
  A.din = din; A.y = y; A.bnp = bnp; A.g = g; A.partial = partial; A.blk = blk; A.out_blk = out_blk; A.dfeat = dfeat;
  A.skip = skip; A.dskip = dskip; A.gres = gres; A.T = T; A.CT = CT; A.S = S; A.Sp = Sp;
  A.t0 = last_step_only ? T - 1 : 0;
  const unsigned nbx = dof_cdiv(S, 256), nby = last_step_only ? 1u : (unsigned)T;
  DOF_LAUNCH(k_tcn_bn_bwd1_w, (nbx, nby), (256), st, A);
  if (int rc = dof_check_launch("k_tcn_bn_bwd1_w")) return rc;
  return sums ? dof_launch_sum_partials(partial, (int64_t)nbx * nby, 2 * CT, sums, 0, st) : DOF_OK;
}

int dof_launch_tcn_bn_bwd2(float* g, const float* y, const float* bnp, const float* coef, int T, int CT, int64_t S,
                           int64_t Sp, hipStream_t st) {
  DOF_LAUNCH(k_tcn_bn_bwd2, (dof_cdiv(S * (CT / 4), 256), (unsigned)T), (256), st, g, y, bnp, coef, T, CT, S, Sp);
  return dof_check_launch("k_tcn_bn_bwd2");
}

// generic convolution (LDS weights): (KC, NC) in {(32,64), (64,64), (64,32)}; ci = input channels of the FORWARD
// convolution's weight tensor (its middle dimension), cin_real <= padded KC / NC
int dof_launch_tcn_convg(int reverse, int KC, int NC, const float* in, const float* w, int w_ci, int cin_real,
                         const float* bias, const float* bnp_in, float* a_out, float* out, float* partial,
                         int accumulate, int T, int dil, int64_t S, int64_t Sp, hipStream_t st) {
  TcnConvArgs A;
  A.in = in; A.w = w; A.bias = bias; A.bnp_in = bnp_in; A.a_out = a_out; A.out = out; A.partial = partial;
  A.fuse_y = nullptr; A.fuse_bnp = nullptr;
  A.bwd_y = nullptr; A.bwd_bnp = nullptr; A.bwd_coef = nullptr;
  A.stat_shift = nullptr; A.bwd_store = 1;
  A.T = T; A.dil = dil; A.accumulate = accumulate; A.S = S; A.Sp = Sp;
  const unsigned nb = (unsigned)(dof_tcn_conv_waves(T, Sp) / 4);
#define CONVG(R, BN, K, N) DOF_LAUNCH((k_tcn_convg<R, BN, K, N>), (nb), (256), st, A, cin_real, w_ci)
  if (!reverse && !bnp_in && KC == 32 && NC == 64) CONVG(false, false, 32, 64);
  else if (!reverse && !bnp_in && KC == 64 && NC == 64) CONVG(false, false, 64, 64);
  else if (!reverse && bnp_in && KC == 64 && NC == 64) CONVG(false, true, 64, 64);
  else if (reverse && KC == 64 && NC == 64) CONVG(true, false, 64, 64);
  else if (reverse && KC == 64 && NC == 32) CONVG(true, false, 64, 32);
  else {
    dof_set_error("tcn conv variant (reverse %d, bn %d, %d -> %d) not built", reverse, bnp_in != nullptr, KC, NC);
    return DOF_ERR_UNSUPPORTED;
  }
#undef CONVG
  return dof_check_launch("k_tcn_convg");
}

namespace {
// ---------------------------------------------------------------------------------------------------------
// Weight gradient of a 32 -> 32 dilated convolution: dW[o][c][j] = sum_{t,s} dy[t][s][o] * in[t - (3-j) d][s][c],
// db[o] = sum dy.  The generic k_outer streams every MFMA operand from global memory with scalar dword loads and
// reads the input once per tap (4x) and dy once per 4-tile job (2x): 770 us per convolution at B = 8192 (24 % MFMA
// utilisation, issue-stalled on the loads).  Here a workgroup stages all T time steps of a few (4) sequences of both
// tensors in LDS with coalesced 16-byte loads (each tensor is read from HBM exactly once), wavefront j owns tap j:
// per time step 4 MFMAs (2 x 2 tiles, one k-slice of 4 sequences) against 4 conflict-free ds_reads; 26 KB of LDS
// per workgroup, so six workgroups per CU overlap each other's load and MFMA phases.
template <int NSEQ>  // sequences per staged chunk: 4 (one MFMA k-slice) or 8
__global__ void __launch_bounds__(256, 6) k_tcn_wgrad(const DofTcnWgrad* __restrict__ descs, float* __restrict__ partials) {
  __shared__ float sx[DOF_TCN_WGRAD_MAX_T][NSEQ][33];
  __shared__ float sd[DOF_TCN_WGRAD_MAX_T][NSEQ][33];
  const DofTcnWgrad D = descs[blockIdx.y];
  if ((int)blockIdx.x >= D.nblk || D.cin != 0) return;
  const int T = D.T;
  const int64_t Sp = D.Sp;
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int shift = -(3 - j) * D.dil;
  dof_f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float rs0 = 0.0f, rs1 = 0.0f;
  const int64_t chunks = Sp / NSEQ;
  constexpr int F4 = NSEQ * 8;  // float4 per time step of one tensor
  const int cth = (threadIdx.x & 7) * 4;  // a thread stages the same four channels in every pass (256 % 8 == 0)
  for (int64_t ch = blockIdx.x; ch < chunks; ch += D.nblk) {
    __syncthreads();
    // lazy operands: the per-channel constants are (re)read per chunk so that they do not occupy registers during the
    // MFMA phase (six workgroups per CU leave 80 VGPRs per lane)
    DOF_MEM_FENCE();
    float xs[4], xh[4], ka[4], kb[4], kc[4], bm[4];
    if (D.in_bnp) {
      dof_ld_row<4>(D.in_bnp + 2 * 32 + cth, xs);
      dof_ld_row<4>(D.in_bnp + 3 * 32 + cth, xh);
    }
    if (D.dy_y) {  // dy = scale (g - c1 - (y - mean) rstd c2) = ka g + kb (y - mean) + kc
      float br[4], c1[4], c2[4];
      dof_ld_row<4>(D.dy_bnp + cth, bm);
      dof_ld_row<4>(D.dy_bnp + 32 + cth, br);
      dof_ld_row<4>(D.dy_bnp + 2 * 32 + cth, ka);
      dof_ld_row<4>(D.dy_coef + cth, c1);
      dof_ld_row<4>(D.dy_coef + 32 + cth, c2);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        kb[k] = -ka[k] * br[k] * c2[k];
        kc[k] = -ka[k] * c1[k];
      }
    }
    for (int idx = threadIdx.x; idx < T * F4; idx += 256) {
      const int t = idx / F4, r = idx - t * F4, s = r >> 3, c = (r & 7) * 4;
      const int64_t off = ((int64_t)t * Sp + ch * NSEQ + s) * 32 + c;
      const float4 vx = *reinterpret_cast<const float4*>(D.in + off);
      const float4 vd = *reinterpret_cast<const float4*>(D.dy + off);
      float ex[4] = {vx.x, vx.y, vx.z, vx.w}, ed[4] = {vd.x, vd.y, vd.z, vd.w};
      if (D.in_bnp) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ex[k] = fmaxf(fmaf(ex[k], xs[k], xh[k]), 0.0f);
      }
      if (D.dy_y) {
        const float4 vy = *reinterpret_cast<const float4*>(D.dy_y + off);
        const float ey[4] = {vy.x, vy.y, vy.z, vy.w};
        const bool valid = ch * NSEQ + s < D.S;
#pragma unroll
        for (int k = 0; k < 4; ++k) ed[k] = valid ? fmaf(ka[k], ed[k], fmaf(kb[k], ey[k] - bm[k], kc[k])) : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sx[t][s][c + k] = ex[k];
        sd[t][s][c + k] = ed[k];
      }
    }
    __syncthreads();
    for (int t = shift < 0 ? -shift : 0; t < T; ++t) {
      const int tb = t + shift;
#pragma unroll
      for (int kk = 0; kk < NSEQ / 4; ++kk) {
        const float a0 = sd[t][q + 4 * kk][i], a1 = sd[t][q + 4 * kk][16 + i];
        const float b0 = sx[tb][q + 4 * kk][i], b1 = sx[tb][q + 4 * kk][16 + i];
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        rs0 += a0;  // tap 3 has shift 0, i.e. sees every time step exactly once: its row sums are the bias gradient
        rs1 += a1;
      }
    }
  }
  // D layout of mfma_f32_16x16x4: lane holds rows (lane>>4)*4 + r, column lane&15
  float* out = partials + (j < 2 ? D.part0 : D.part1) + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) out[(mt * 16 + q * 4 + r4) * 65 + (j & 1) * 32 + nt * 16 + i] = acc[mt][nt][r4];
  if (j == 3) {
    rs0 += __shfl_xor(rs0, 16);
    rs0 += __shfl_xor(rs0, 32);
    rs1 += __shfl_xor(rs1, 16);
    rs1 += __shfl_xor(rs1, 32);
    if (q == 0) {
      float* b = partials + D.part0 + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
      b[i * 65 + 64] = rs0;
      b[(16 + i) * 65 + 64] = rs1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe with fp32-exact products (round 3).  k_tcn_wgrad above runs at 82 % of
// the fp32 MFMA rate, which is 1/16 of the bf16 rate.  Every staged fp32 operand is split once into three bf16 pieces
// hi + mid + lo = the value EXACTLY (truncation: 8 + 8 + 8 significand bits), written as three bf16 planes; a product
// a b is then the six piece products down to 2^-24 relative (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid -- the three
// dropped terms are below 2^-24 |a b|), each exact in the fp32 accumulator's input, so the result carries the rounding of
// an fp32 fmaf chain in a different order: six v_mfma_f32_32x32x16_bf16 replace sixteen v_mfma_f32_16x16x4_f32 (2.7 x).
// GEMM view per tap j: dW_j (32 out x 32 in) = sum over k = (t, s) of dy[k][o] x[k + 4 shift_j][c], k time-major over the 4
// sequences of a chunk, padded to a multiple of 16; A rows = out channels, B columns = in channels, both K-contiguous in
// LDS ([plane][channel][k] bf16; x with 16 zero elements in front: a tap's first K-step reaches at most 12 elements before
// the window, K-steps entirely before it are skipped).  48 KB of LDS: 3 workgroups per CU; the next chunk's global loads
// are issued before the MFMA phase of the current one.  Same partial-tile layout as k_tcn_wgrad.
constexpr int WB_DSTR = 120, WB_XSTR = 132, WB_XP = 16;  // row strides (bf16 elements): 240 bytes (16-byte reads, 16 lanes a pass) / 264 bytes (8-byte reads, 32 lanes a pass): conflict-free
__global__ void __launch_bounds__(256, 3) k_tcn_wgrad_b3(const DofTcnWgrad* __restrict__ descs, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint16_t sd16[3][32][WB_DSTR];
  __shared__ __attribute__((aligned(16))) uint16_t sx16[3][32][WB_XSTR];
  const DofTcnWgrad D = descs[blockIdx.y];
  if ((int)blockIdx.x >= D.nblk || D.cin != 0) return;
  const int T = D.T;
  const int64_t Sp = D.Sp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // zero once: the K padding of dy (k >= 4 T) and the borders of x are never written again
  for (int i = tid; i < 3 * 32 * WB_DSTR / 2; i += 256) reinterpret_cast<uint32_t*>(&sd16[0][0][0])[i] = 0u;
  for (int i = tid; i < 3 * 32 * WB_XSTR / 2; i += 256) reinterpret_cast<uint32_t*>(&sx16[0][0][0])[i] = 0u;
  // staging: thread = (time step, 4-channel group); it holds that row piece of the chunk's 4 sequences.  Windows of 26 .. 50
  // steps (round 4): chunks of nq = 2 sequences, thread = (PAIR of time steps, channel group) -- its four rows are (t, s) =
  // (2 st_t + (q >> 1), q & 1), so the K index nq t + s = 4 st_t + q and everything behind the loads is the same code.
  const int st_t = tid >> 3, cg = (tid & 7) * 4;
  const int nq = T > DOF_TCN_WGRAD_MAX_T ? 2 : 4, tstep = 4 / nq;
  const bool stager = st_t * tstep < T;
  float xs[4] = {1.0f, 1.0f, 1.0f, 1.0f}, xh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float ka[4], kb[4], kc[4], bm[4];
  if (D.in_bnp) {
    dof_ld_row<4>(D.in_bnp + 2 * 32 + cg, xs);
    dof_ld_row<4>(D.in_bnp + 3 * 32 + cg, xh);
  }
  if (D.dy_y) {  // dy = scale (g - c1 - (y - mean) rstd c2) = ka g + kb (y - mean) + kc
    float br[4], c1[4], c2[4];
    dof_ld_row<4>(D.dy_bnp + cg, bm);
    dof_ld_row<4>(D.dy_bnp + 32 + cg, br);
    dof_ld_row<4>(D.dy_bnp + 2 * 32 + cg, ka);
    dof_ld_row<4>(D.dy_coef + cg, c1);
    dof_ld_row<4>(D.dy_coef + 32 + cg, c2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      kb[k] = -ka[k] * br[k] * c2[k];
      kc[k] = -ka[k] * c1[k];
    }
  }
  const int tap = (wave + (int)blockIdx.x) & 3;  // rotated per workgroup: the taps' K-step counts differ (skipped steps)
  const int shift = -(3 - tap) * D.dil;
  const int KS = (nq * T + 15) / 16, ks0 = (-shift) * nq / 16;
  dof_f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int64_t chunks = Sp / nq;
  float4 rx[4], rd[4], ry[4];
  // row q of the thread: nq = 4: (st_t, sequence q), consecutive rows; nq = 2: (2 st_t + (q >> 1), sequence q & 1) -- a second
  // time step past an odd T re-reads step T - 1 and is zeroed below
  const bool t1 = nq == 4 || 2 * st_t + 1 < T;
  const int64_t tstride = (nq == 2 && t1) ? Sp * 32 : 0;  // q = 2, 3 of a pair: the same two sequences one time step on
  auto fetch = [&](int64_t ch) {  // unconditional loads (clamped chunk): a predicate around them would serialise the batch
    const int64_t chc = ch < chunks ? ch : chunks - 1;
    const int64_t base = ((int64_t)(stager ? st_t * tstep : 0) * Sp + chc * nq) * 32 + cg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t o = base + (nq == 4 ? q : (q & 1)) * 32 + (q >= 2 ? tstride : 0);
      rx[q] = *reinterpret_cast<const float4*>(D.in + o);
      rd[q] = *reinterpret_cast<const float4*>(D.dy + o);
      if (D.dy_y) ry[q] = *reinterpret_cast<const float4*>(D.dy_y + o);
    }
  };
  fetch(blockIdx.x);
  const int am = lane & 31, kg = lane >> 5;
  const bool b4 = ((nq * shift) & 3) != 0;
  for (int64_t ch = blockIdx.x; ch < chunks; ch += D.nblk) {
    __syncthreads();  // the previous chunk's MFMA phase (and the zero fill) is done with the planes
    if (stager) {
      float ex[4][4], ed[4][4];  // [sequence][channel]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vx[4] = {rx[q].x, rx[q].y, rx[q].z, rx[q].w}, vd[4] = {rd[q].x, rd[q].y, rd[q].z, rd[q].w};
        const float vy[4] = {ry[q].x, ry[q].y, ry[q].z, ry[q].w};
        const bool row = q < 2 || t1;  // nq = 2: the pair's second step exists
        const bool valid = row && ch * nq + (nq == 4 ? q : (q & 1)) < D.S;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ex[q][k] = !row ? 0.0f : D.in_bnp ? fmaxf(fmaf(vx[k], xs[k], xh[k]), 0.0f) : vx[k];
          ed[q][k] = D.dy_y ? (valid ? fmaf(ka[k], vd[k], fmaf(kb[k], vy[k] - bm[k], kc[k])) : 0.0f) : (row ? vd[k] : 0.0f);
          rs[k] += ed[q][k];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float vd[4] = {ed[0][k], ed[1][k], ed[2][k], ed[3][k]}, vx[4] = {ex[0][k], ex[1][k], ex[2][k], ex[3][k]};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const uint64_t wd = (uint64_t)dof_pack_hi16(vd[0], vd[1]) | ((uint64_t)dof_pack_hi16(vd[2], vd[3]) << 32);
          const uint64_t wx = (uint64_t)dof_pack_hi16(vx[0], vx[1]) | ((uint64_t)dof_pack_hi16(vx[2], vx[3]) << 32);
          *reinterpret_cast<uint64_t*>(&sd16[p][cg + k][4 * st_t]) = wd;
          *reinterpret_cast<uint64_t*>(&sx16[p][cg + k][WB_XP + 4 * st_t]) = wx;
          if (p < 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              vd[q] = dof_bf16_rest(vd[q]);
              vx[q] = dof_bf16_rest(vx[q]);
            }
          }
        }
      }
    }
    fetch(ch + D.nblk);  // lands during the MFMA phase
    __syncthreads();
    for (int ks = ks0; ks < KS; ++ks) {
      const int ka0 = ks * 16 + kg * 8, kb0 = WB_XP + ka0 + nq * shift;
      const dof_bf16x8 ah = dof_ld_bf16x8(&sd16[0][am][ka0]), amid = dof_ld_bf16x8(&sd16[1][am][ka0]),
                       al = dof_ld_bf16x8(&sd16[2][am][ka0]);
      dof_bf16x8 bh, bmid, bl;
      if (b4) {  // 2-sequence chunks at an odd time shift: the B rows start on a 4-byte boundary only
        bh = dof_ld_bf16x8_a4(&sx16[0][am][kb0]); bmid = dof_ld_bf16x8_a4(&sx16[1][am][kb0]); bl = dof_ld_bf16x8_a4(&sx16[2][am][kb0]);
      } else {
        bh = dof_ld_bf16x8(&sx16[0][am][kb0]); bmid = dof_ld_bf16x8(&sx16[1][am][kb0]); bl = dof_ld_bf16x8(&sx16[2][am][kb0]);
      }
      acc = DOF_MFMA_32x32x16_BF16(ah, bl, acc);
      acc = DOF_MFMA_32x32x16_BF16(al, bh, acc);
      acc = DOF_MFMA_32x32x16_BF16(amid, bmid, acc);
      acc = DOF_MFMA_32x32x16_BF16(ah, bmid, acc);
      acc = DOF_MFMA_32x32x16_BF16(amid, bh, acc);
      acc = DOF_MFMA_32x32x16_BF16(ah, bh, acc);
    }
  }
  float* out = partials + (tap < 2 ? D.part0 : D.part1) + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
#pragma unroll
  for (int v = 0; v < 16; ++v) out[(8 * (v / 4) + 4 * kg + v % 4) * 65 + (tap & 1) * 32 + am] = acc[v];
  // bias gradient: channel sums of dy over the stagers of a channel group (fixed order)
  __syncthreads();
  float* red = reinterpret_cast<float*>(&sd16[0][0][0]);  // [32 time rows][32 channels]
  if (tid < 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[st_t * 32 + cg + k] = stager ? rs[k] : 0.0f;
  }
  __syncthreads();
  if (tid < 32) {
    float b = 0.0f;
    for (int t = 0; t < 32; ++t) b += red[t * 32 + tid];
    partials[D.part0 + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS + tid * 65 + 64] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The first block's weight gradients (round 3): conv1 reads the raw input (3 or 1 channels per sequence, dilation 1) and
// the block's residual branch is a 1 x 1 convolution of the same input.  Through the generic reduction they cost a pass
// that normalises conv1's gradient in place (k_tcn_bn_bwd2) plus k_outer's operand stream over two 32-channel gradient
// tensors.  Here a lane owns four channels of one sequence and walks its T steps: 16-byte loads of the two gradient rows
// straight from HBM (pass 2 of BatchNorm1's backward applied on load, like k_tcn_wgrad), the input row shared by the
// eight lanes of the sequence and kept in a four-step register window, and the (4 taps + 1) x cin outer products on the
// vector pipe (a 32 x 15 result does not need the matrix pipe).  No LDS staging, so occupancy is register-bound and the
// loads of several steps are in flight; the sequences of a wavefront are added by a fixed butterfly, the wavefronts in
// order, into partial tiles of k_outer's layout.
template <int F>
__device__ __forceinline__ void tcn_wgrad_in_body(const DofTcnWgrad& D, float* __restrict__ partials, float* red) {
  constexpr int NV = 5 * F + 2;  // values per channel: 4 F conv taps, conv bias, F residual weights, its bias
  // lane = (sequence of the wave's eight) * 8 + channel quad: a wavefront reads 1 KB of contiguous gradient rows per
  // 16-byte load instruction, the eight lanes of a sequence share its input row
  const int T = D.T, tid = threadIdx.x;
  const int64_t Sp = D.Sp;
  const int c4 = (tid & 7) * 4, sl = tid >> 3;
  float ka[4], kb[4], kc[4], bm[4];
  const bool lazy = D.dy_y != nullptr, two = D.dy2 != nullptr;
  if (lazy) {  // dy = scale (g - c1 - (y - mean) rstd c2) = ka g + kb (y - mean) + kc
    float br[4], c1[4], c2[4];
    dof_ld_row<4>(D.dy_bnp + c4, bm);
    dof_ld_row<4>(D.dy_bnp + 32 + c4, br);
    dof_ld_row<4>(D.dy_bnp + 2 * 32 + c4, ka);
    dof_ld_row<4>(D.dy_coef + c4, c1);
    dof_ld_row<4>(D.dy_coef + 32 + c4, c2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      kb[k] = -ka[k] * br[k] * c2[k];
      kc[k] = -ka[k] * c1[k];
    }
  }
  float acc[4][NV];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[c][v] = 0.0f;
  for (int64_t s = (int64_t)blockIdx.x * 64 + sl; s < Sp; s += (int64_t)D.nblk * 64) {
    const bool valid = s < D.S;
    float xw[4][F];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int f = 0; f < F; ++f) xw[j][f] = 0.0f;
#pragma unroll 5
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)t * Sp + s;
      float a[4], b2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      dof_ld_row<4>(D.dy + row * 32 + c4, a);
      if (lazy) {
        float y[4];
        dof_ld_row<4>(D.dy_y + row * 32 + c4, y);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = valid ? fmaf(ka[k], a[k], fmaf(kb[k], y[k] - bm[k], kc[k])) : 0.0f;
      }
      if (two) dof_ld_row<4>(D.dy2 + row * 32 + c4, b2);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        xw[0][f] = xw[1][f];
        xw[1][f] = xw[2][f];
        xw[2][f] = xw[3][f];
        xw[3][f] = D.in[row * F + f];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int f = 0; f < F; ++f) acc[c][j * F + f] = fmaf(a[c], xw[j][f], acc[c][j * F + f]);
        acc[c][4 * F] += a[c];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[c][4 * F + 1 + f] = fmaf(b2[c], xw[3][f], acc[c][4 * F + 1 + f]);
        acc[c][5 * F + 1] += b2[c];
      }
    }
  }
  // the eight sequences of a wavefront (fixed butterfly), then the eight wavefronts in order
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float x = acc[c][v];
      x += __shfl_xor(x, 8);
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      acc[c][v] = x;
    }
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int v = 0; v < NV; ++v) red[(wave * 32 + c4 + c) * NV + v] = acc[c][v];
  }
  __syncthreads();
  float* p0 = partials + D.part0 + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
  float* p1 = partials + D.part1 + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
  for (int e = tid; e < 32 * NV; e += 512) {
    const int oo = e / NV, v = e - oo * NV;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += red[(k * 32 + oo) * NV + v];
    if (v < 4 * F) p0[oo * 65 + (v / F) * 16 + (v % F)] = sum;
    else if (v == 4 * F) p0[oo * 65 + 64] = sum;
    else if (two) {
      if (v < 5 * F + 1) p1[oo * 65 + (v - 4 * F - 1)] = sum;
      else p1[oo * 65 + 64] = sum;
    }
  }
}

// both streams' descriptors in one launch (node: 3 input channels, edge: 1), so that the two fill the chip together
__global__ void __launch_bounds__(512) k_tcn_wgrad_in(const DofTcnWgrad* __restrict__ descs, float* __restrict__ partials) {
  __shared__ float red[8 * 32 * 17];
  const DofTcnWgrad D = descs[blockIdx.y];
  if ((int)blockIdx.x >= D.nblk) return;
  if (D.cin == 3) tcn_wgrad_in_body<3>(D, partials, red);
  else if (D.cin == 1) tcn_wgrad_in_body<1>(D, partials, red);
}

}  // namespace

// DOF_TCN_WGRAD_FP32=1: the fp32-MFMA kernel (A/B measurements)
static int dof_tcn_wgrad_fp32() {
  static const int v = [] {
    const char* e = getenv("DOF_TCN_WGRAD_FP32");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return v;
}

int dof_tcn_wgrad_max_t(void) {
  if (dof_tcn_wgrad_fp32()) return DOF_TCN_WGRAD_MAX_T;
  return tct_max_t() < DOF_TCN_WGRAD_B3_MAX_T ? DOF_TCN_WGRAD_MAX_T : DOF_TCN_WGRAD_B3_MAX_T;
}
int dof_launch_tcn_wgrad(const DofTcnWgrad* descs_dev, int n, int max_nblk, float* partials, hipStream_t st) {
  if (n <= 0) return DOF_OK;
  if (dof_tcn_wgrad_fp32()) DOF_LAUNCH((k_tcn_wgrad<4>), ((unsigned)max_nblk, (unsigned)n), (256), st, descs_dev, partials);
  else DOF_LAUNCH(k_tcn_wgrad_b3, ((unsigned)max_nblk, (unsigned)n), (256), st, descs_dev, partials);
  return dof_check_launch("k_tcn_wgrad");
}
// the descriptors with cin = 3 / 1 (first blocks) among the same table: n_in of them, after the 32-channel ones
int dof_launch_tcn_wgrad_in(const DofTcnWgrad* descs_dev, int n, int max_nblk, float* partials, hipStream_t st) {
  if (n <= 0) return DOF_OK;
  DOF_LAUNCH(k_tcn_wgrad_in, ((unsigned)max_nblk, (unsigned)n), (512), st, descs_dev, partials);
  return dof_check_launch("k_tcn_wgrad_in");
}

int dof_launch_head_rms(const float* flat, float* hn, float* rinv, int J, int64_t B, int64_t Bp, hipStream_t st) {
  DOF_LAUNCH(k_head_rms, (dof_cdiv(B, 256)), (256), st, flat, hn, rinv, J, B, Bp);
  return dof_check_launch("k_head_rms");
}

int dof_launch_head_dense(const float* in, const float* bnp_in, float* in_norm, const float* w, const float* bias,
                          float* out, float* partial, float* sums, int CI, int CO, int relu, int64_t B, int64_t Bp,
                          hipStream_t st) {
  const unsigned nb = dof_cdiv(B, 256);
  DOF_LAUNCH(k_head_dense, (nb, (unsigned)CO), (256), st, in, bnp_in, in_norm, w, bias, out, partial, CI, relu, B, Bp);
  if (partial) {
    DOF_LAUNCH(k_head_sum, (1), (64), st, (const float*)partial, (int)nb, sums, CO);
    DOF_LAUNCH(k_head_var, ((unsigned)CO), (256), st, (const float*)out, sums, (float)B, CO, B, Bp);
  }
  return dof_check_launch("k_head_dense");
}

int dof_launch_head_dense_bwd(const float* dout, const float* w, float* din, int CI, int CO, int64_t B, int64_t Bp,
                              hipStream_t st) {
  DOF_LAUNCH(k_head_dense_bwd, (dof_cdiv(B, 256), (unsigned)CI), (256), st, dout, w, din, CI, CO, B, Bp);
  return dof_check_launch("k_head_dense_bwd");
}

int dof_launch_head_bn_bwd(const float* g, const float* h, const float* bnp, float* partial, float* sums, float* coef,
                           float* dgamma, float* dbeta, int accumulate, float* dpre, int C, int64_t B, int64_t Bp,
                           hipStream_t st, int relu, bool frozen) {
  const unsigned nb = dof_cdiv(B, 256);
  DOF_LAUNCH(k_head_bn_bwd1, (nb, (unsigned)C), (256), st, g, h, bnp, partial, C, B, Bp);
  DOF_LAUNCH(k_head_sum, (1), (64), st, (const float*)partial, (int)nb, sums, C);
  DOF_LAUNCH(k_bn_bwd_fin, (1), (64), st, (const float*)sums, frozen ? __builtin_inff() : (float)B, dgamma, dbeta, accumulate, coef, C);
  DOF_LAUNCH(k_head_bn_bwd2, (nb, (unsigned)C), (256), st, g, h, bnp, (const float*)coef, dpre, C, relu, B, Bp);
  return dof_check_launch("k_head_bn_bwd");
}

int dof_launch_head_rms_bwd(const float* dhn, const float* hn, const float* rinv, float* dflat, int J, int64_t B,
                            int64_t Bp, hipStream_t st) {
  DOF_LAUNCH(k_head_rms_bwd, (dof_cdiv(B, 256)), (256), st, dhn, hn, rinv, dflat, J, B, Bp);
  return dof_check_launch("k_head_rms_bwd");
}

int dof_launch_dec_repeat(const float* d2, const float* bnp, float* zrep, int C4, int T, int64_t B, int64_t Bp,
                          hipStream_t st) {
  if (C4 > 32) DOF_LAUNCH((k_dec_repeat<64>), (dof_cdiv((int64_t)T * B, 256)), (256), st, d2, bnp, zrep, C4, T, B, Bp);
  else DOF_LAUNCH((k_dec_repeat<32>), (dof_cdiv((int64_t)T * B, 256)), (256), st, d2, bnp, zrep, C4, T, B, Bp);
  return dof_check_launch("k_dec_repeat");
}

int dof_launch_dec_sum_time(const float* dzrep, float* dzf, int C4, int T, int64_t B, int64_t Bp, hipStream_t st) {
  if (C4 > 32) DOF_LAUNCH((k_dec_sum_time<64>), (dof_cdiv(B, 256)), (256), st, dzrep, dzf, C4, T, B, Bp);
  else DOF_LAUNCH((k_dec_sum_time<32>), (dof_cdiv(B, 256)), (256), st, dzrep, dzf, C4, T, B, Bp);
  return dof_check_launch("k_dec_sum_time");
}

int dof_launch_tcn_dec_out(const float* skip, const float* wp, const float* bp, const float* x, const float* valid,
                           float* hid, float* loc_out, float* recon_partial, float* dloc, float* dskip, int T, int C3,
                           int train, int64_t B, int64_t Bp, hipStream_t st) {
  TcnDecOutArgs A;
  A.skip = skip; A.wp = wp; A.bp = bp; A.x = x; A.valid = valid; A.hid = hid; A.loc_out = loc_out;
  A.recon_partial = recon_partial; A.dloc = dloc; A.dskip = dskip; A.T = T; A.C3 = C3; A.train = train; A.B = B; A.Bp = Bp;
  DOF_LAUNCH(k_tcn_dec_out, (dof_cdiv((int64_t)T * B, 256)), (256), st, A);
  return dof_check_launch("k_tcn_dec_out");
}

// Batch statistics of one TCN layer from the convolution's channel-sum partials (rows of `stride` floats, the first
// CT of which are the sums of y) + a centred second pass over y; leaves sums[2][CT] for dof_launch_bn_fwd_fin.
// shift != null: the partials are shifted sums (k_tcn_conv_t with stat_shift = shift); the second pass only runs for
// ill-conditioned channels and sums needs 3 CT floats (dof_launch_bn_fwd_fin with shifted = 1).
int dof_launch_tcn_bn_stats(const float* y, float* partial, int64_t n_partial, int stride, float* sums, float count,
                            int T, int CT, int64_t S, int64_t Sp, hipStream_t st, const float* shift) {
  TRY_RC(dof_launch_sum_partials(partial, n_partial, stride, sums, 0, st));
  const unsigned nb = (unsigned)dof_tcn_row_blocks(T, S);
  DOF_LAUNCH(k_tcn_var, (nb, (unsigned)(CT / TC)), (256), st, y, (const float*)sums, count, partial, T, CT, S, Sp, shift);
  DOF_LAUNCH(k_tcn_var_sum, ((unsigned)CT), (256), st, (const float*)partial, (int64_t)nb, CT, sums,
             shift ? count : 0.0f);
  return dof_check_launch("k_tcn_var");
}
