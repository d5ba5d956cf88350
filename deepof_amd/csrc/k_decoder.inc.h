// Recurrent decoder tail + reconstruction likelihood.
//
// Reference semantics restated here:
//   * validity mask / lengths from all-zero INPUT frames     /root/reference/deepof/clustering/models_new.py:330-332
//   * Conv1d(4L->2L, k=5, same, no bias) + ReLU + LayerNorm   models_new.py:365-369
//   * ProbabilisticDecoderPT: loc = Linear(2L->3N); Independent(Normal(loc,1)) scaled by the
//     mask: log_prob = -0.5||x-loc||^2 - 3N/2 log(2pi) on valid frames, NaN on masked frames
//     (scale 0)                                               models_new.py:686-710
//   * reconstruction loss = -mean_{b,t} log_prob              losses.py:586
// The bidirectional GRUs / LayerNorms in front of this tail are the shared kernels of k_rnn.hip.
#include "dof_rt.h"
#include "launchers.h"

namespace {


// thread = (b, t): frame validity; k_dec_len then counts per window
__global__ void __launch_bounds__(256) k_dec_valid(const float* __restrict__ x, int T, int C3, int64_t B, int64_t Bp,
                                                   float* __restrict__ valid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int64_t b = i / T;
  const int t = (int)(i - b * T);
  const float* __restrict__ row = x + i * C3;
  bool any = false;
  for (int j = 0; j < C3; ++j) any |= (row[j] != 0.0f);
  valid[(int64_t)t * Bp + b] = any ? 1.0f : 0.0f;
}

__global__ void __launch_bounds__(256) k_dec_len(const float* __restrict__ valid, int T, int64_t B, int64_t Bp,
                                                 int* __restrict__ len) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0;
  for (int t = 0; t < T; ++t) n += valid[(int64_t)t * Bp + b] != 0.0f ? 1 : 0;
  len[b] = n;
}

// both at once (the recurrent decoder needs the lengths too): one 64-lane group per window, lane = time step, the count is
// a reduction over the group -- one launch instead of two 7 us ones
__global__ void __launch_bounds__(256) k_dec_valid_len(const float* __restrict__ x, int T, int C3, int64_t B, int64_t Bp,
                                                       float* __restrict__ valid, int* __restrict__ len) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float n = 0.0f;
  if (b < B) {
    for (int t = lane; t < T; t += 64) {
      const float* __restrict__ row = x + (b * T + t) * C3;
      bool any = false;
      for (int j = 0; j < C3; ++j) any |= (row[j] != 0.0f);
      valid[(int64_t)t * Bp + b] = any ? 1.0f : 0.0f;
      n += any ? 1.0f : 0.0f;
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
  if (b < B && lane == 0) len[b] = (int)n;
}

struct DecTailArgs {
  const float* n2;     // [T][4L][Bp]
  const float *wc, *g3, *b3, *wp, *bp;  // conv (2L,4L,5), norm3, loc_projection (3N,2L),(3N)
  const float* x;      // (B,T,3N)
  const float* valid;  // [T][Bp]
  float* cv;           // [T][2L][Bp] relu(conv)
  float* n3;           // [T][2L][Bp]
  float* loc_out;      // (B,T,3N) or null
  float* recon_partial;  // [nblk]
  float* dloc;         // [T][3N][Bp]   (train)
  float* dcv;          // [T][2L][Bp]   (train) grad at conv pre-activation
  float* ln3_partial;  // [nblk][4L]    (train)
  int T, C3, train;
  int64_t B, Bp;
};

// CONV = false: the convolution already ran on the matrix pipe (five shifted GEMMs, vade.hip: dec_conv_gemm) and left its
// pre-activation in A.cv; this kernel then starts at the ReLU.  Thread-per-row, the convolution is 2 L x 4 L x 5 FMAs per
// row with scalar weight loads: 2.0 ms of the 13 ms step at latent 32 (profiles/r04_latent32_grum_kernel_stats.md).
template <int L, bool CONV = true>
__global__ void __launch_bounds__(256) k_dec_tail(DecTailArgs A) {
  constexpr int CI = 4 * L, CO = 2 * L;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < (int64_t)A.T * A.B;
  float vals[2 * CO];
#pragma unroll
  for (int c = 0; c < 2 * CO; ++c) vals[c] = 0.0f;
  float nll[1] = {0.0f};
  if (live) {
    const int t = (int)(i / A.B);
    const int64_t b = i - (int64_t)t * A.B;
    const dof_cfp wc = dof_cw(A.wc), g3 = dof_cw(A.g3), b3 = dof_cw(A.b3), wp = dof_cw(A.wp), bp = dof_cw(A.bp);
    float cv[CO];
    if constexpr (CONV) {
#pragma unroll
      for (int o = 0; o < CO; ++o) cv[o] = 0.0f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int ts = t + k - 2;
        if (ts < 0 || ts >= A.T) continue;
        float xin[CI];
        dof_ld_row<CI>(A.n2 + ACT(ts, 0, CI, A.Bp, b), xin);
#pragma unroll
        for (int c = 0; c < CI; ++c) {
#pragma unroll
          for (int o = 0; o < CO; ++o) cv[o] = fmaf(wc[(o * CI + c) * 5 + k], xin[c], cv[o]);
        }
      }
    } else {
      (void)wc;
      dof_ld_row<CO>(A.cv + ACT(t, 0, CO, A.Bp, b), cv);
    }
    float mean = 0.0f;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      cv[o] = cv[o] > 0.0f ? cv[o] : 0.0f;
      mean += cv[o];
    }
    dof_st_row<CO>(A.cv + ACT(t, 0, CO, A.Bp, b), cv);
    mean *= (1.0f / CO);
    float var = 0.0f;
    float xh[CO], n3[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      xh[o] = cv[o] - mean;
      var = fmaf(xh[o], xh[o], var);
    }
    const float rstd = rsqrtf(var * (1.0f / CO) + 1e-3f);
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      xh[o] *= rstd;
      n3[o] = fmaf(xh[o], g3[o], b3[o]);
    }
    dof_st_row<CO>(A.n3 + ACT(t, 0, CO, A.Bp, b), n3);
    const bool ok = A.valid[(int64_t)t * A.Bp + b] != 0.0f;
    const float inv_bt = 1.0f / ((float)A.B * (float)A.T);
    const float* __restrict__ xr = A.x + (b * A.T + t) * A.C3;
    float dn3[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) dn3[o] = 0.0f;
    float sq = 0.0f;
    for (int j = 0; j < A.C3; ++j) {
      float loc = bp[j];
#pragma unroll
      for (int o = 0; o < CO; ++o) loc = fmaf(wp[j * CO + o], n3[o], loc);
      if (loc != loc) loc = 0.0f;  // nan_to_num(nan=0, +-inf -> +-1e6)
      loc = fminf(fmaxf(loc, -1e6f), 1e6f);
      if (A.loc_out) A.loc_out[(b * A.T + t) * A.C3 + j] = loc;
      const float df = xr[j] - loc;
      sq = fmaf(df, df, sq);
      if (A.train) {
        const float dl = ok ? -df * inv_bt : NAN;
        A.dloc[ACT(t, j, A.C3, A.Bp, b)] = dl;
#pragma unroll
        for (int o = 0; o < CO; ++o) dn3[o] = fmaf(wp[j * CO + o], dl, dn3[o]);
      }
    }
    const float LOG_2PI = 1.8378770664093453f;
    nll[0] = ok ? 0.5f * sq + 0.5f * (float)A.C3 * LOG_2PI : NAN;
    if (A.train) {
      float mg = 0.0f, mgx = 0.0f;
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float g = dn3[o] * g3[o];
        mg += g;
        mgx = fmaf(g, xh[o], mgx);
        vals[o] = dn3[o] * xh[o];
        vals[CO + o] = dn3[o];
      }
      mg *= (1.0f / CO);
      mgx *= (1.0f / CO);
      float drow[CO];
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float d = rstd * (dn3[o] * g3[o] - mg - xh[o] * mgx);
        drow[o] = cv[o] > 0.0f ? d : 0.0f;
      }
      dof_st_row<CO>(A.dcv + ACT(t, 0, CO, A.Bp, b), drow);
    }
  }
  dof_block_colsum<1>(nll, A.recon_partial + blockIdx.x);
  if (A.train) dof_block_colsum<2 * CO>(vals, A.ln3_partial + (int64_t)blockIdx.x * 2 * CO);
}

// d n2[t][c] = sum_k sum_o wc[o][c][k] * dcv[t-k+2][o]
// Conv1d weights (CO, CI, 5) -> five (CO, CI) matrices, one per tap: the operand layout the GEMM kernel stages
__global__ void __launch_bounds__(256) k_dec_conv_taps(const float* __restrict__ wc, float* __restrict__ taps, int n_oc) {
  const int e = blockIdx.x * 256 + threadIdx.x;   // (o * CI + c)
  if (e >= n_oc) return;
#pragma unroll
  for (int k = 0; k < 5; ++k) taps[k * n_oc + e] = wc[e * 5 + k];
}

template <int L>
__global__ void __launch_bounds__(256) k_dec_conv_bwd(const float* __restrict__ dcv, const float* __restrict__ wc,
                                                      float* __restrict__ dn2, int T, int64_t B, int64_t Bp) {
  constexpr int CI = 4 * L, CO = 2 * L;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * B) return;
  const int t = (int)(i / B);
  const int64_t b = i - (int64_t)t * B;
  const dof_cfp wcc = dof_cw(wc);
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.0f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int ts = t - k + 2;
    if (ts < 0 || ts >= T) continue;
    float drow[CO];
    dof_ld_row<CO>(dcv + ACT(ts, 0, CO, Bp, b), drow);
#pragma unroll
    for (int o = 0; o < CO; ++o) {
#pragma unroll
      for (int c = 0; c < CI; ++c) acc[c] = fmaf(wcc[(o * CI + c) * 5 + k], drow[o], acc[c]);
    }
  }
  dof_st_row<CI>(dn2 + ACT(t, 0, CI, Bp, b), acc);
}

// ---------------------------------------------------------------------------------------------
// Decoder tail for latent 8, one LANE per output channel: the 16 lanes of a DPP row own one (t, b) row (conv output
// channel o = lane), a workgroup 16 rows.  The thread-per-row form above runs 400 wavefronts of ~4000 dependent FMAs at
// batch 1024 (51 us, pure latency); this form runs 6400 wavefronts of ~300.  Conv and projection weights are staged in
// LDS once per workgroup (transposed so that a lane's reads are conflict-free 16-byte words), the five input rows of the
// convolution too (broadcast reads); the 42-wide projection is spread over the 16 lanes (outputs j = lane + 16 m), its
// backward gathers the row's d loc through LDS.  Same arithmetic per element; the sums over channels run in a
// different order than in the row-per-thread form (fp32 rounding only).
// ---------------------------------------------------------------------------------------------
constexpr int kTailMaxC3 = 96;
constexpr int kTailIters = 4;  // row groups of 16 per workgroup of k_dec_tail_w (vade.hip sizes tail_blocks by 16 * kTailIters)

__global__ void __launch_bounds__(256) k_dec_tail_w(DecTailArgs A) {
  constexpr int L = 8, CI = 4 * L, CO = 2 * L, RB = 16, PS = 20;  // PS: padded row stride of the projection weights
  __shared__ float wcs[5 * (CI / 4) * CO * 4];   // [k][c4][o][4]
  __shared__ float xs[RB * 5 * CI];              // [row][k][c]
  __shared__ float wps[kTailMaxC3 * PS];         // [j][o] (stride PS)
  __shared__ float dls[RB * kTailMaxC3];         // [row][j]
  __shared__ float red[RB][2 * CO + 1];
  const int tid = threadIdx.x, o = tid & 15, g = tid >> 4;
  const int C3 = A.C3;
  for (int e = tid; e < 5 * CI * CO; e += 256) {  // e = (o' * CI + c) * 5 + k  (the parameter's own order)
    const int k = e % 5, c = (e / 5) % CI, oo = e / (5 * CI);
    wcs[((k * (CI / 4) + (c >> 2)) * CO + oo) * 4 + (c & 3)] = A.wc[e];
  }
  for (int e = tid; e < C3 * CO; e += 256) wps[(e / CO) * PS + (e % CO)] = A.wp[e];
  const int64_t rows = (int64_t)A.T * A.B;
  auto gsum = [&](float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
  };
  const float g3o = A.g3[o], b3o = A.b3[o];
  const float inv_bt = 1.0f / ((float)A.B * (float)A.T);
  float nll[1] = {0.0f};
  float vals0 = 0.0f, vals1 = 0.0f;
  for (int it = 0; it < kTailIters; ++it) {  // the staged weights serve kTailIters x 16 rows
    const int64_t r0 = ((int64_t)blockIdx.x * kTailIters + it) * RB;
    __syncthreads();  // weights staged / previous iteration done with xs and dls
    for (int e = tid; e < RB * 5 * (CI / 4); e += 256) {  // float4 units of the staged input rows
      const int c4 = e % (CI / 4), k = (e / (CI / 4)) % 5, rr = e / (5 * (CI / 4));
      const int64_t r = r0 + rr;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (r < rows) {
        const int t = (int)(r / A.B);
        const int64_t b = r - (int64_t)t * A.B;
        const int ts = t + k - 2;
        if (ts >= 0 && ts < A.T) v = *reinterpret_cast<const float4*>(A.n2 + ACT(ts, 4 * c4, CI, A.Bp, b));
      }
      *reinterpret_cast<float4*>(xs + (rr * 5 + k) * CI + 4 * c4) = v;
    }
    __syncthreads();
    const int64_t r = r0 + g;
    const bool live = r < rows;
    const int t = live ? (int)(r / A.B) : 0;
    const int64_t b = live ? r - (int64_t)t * A.B : 0;
    // conv + ReLU
    float cvo = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int c4 = 0; c4 < CI / 4; ++c4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + (g * 5 + k) * CI + 4 * c4);
        const float4 wv = *reinterpret_cast<const float4*>(wcs + ((k * (CI / 4) + c4) * CO + o) * 4);
        cvo = fmaf(wv.x, xv.x, cvo); cvo = fmaf(wv.y, xv.y, cvo); cvo = fmaf(wv.z, xv.z, cvo); cvo = fmaf(wv.w, xv.w, cvo);
      }
    cvo = cvo > 0.0f ? cvo : 0.0f;
    if (live) A.cv[ACT(t, o, CO, A.Bp, b)] = cvo;
    // LayerNorm(eps 1e-3)
    const float mean = gsum(cvo) * (1.0f / CO);
    float xh = cvo - mean;
    const float rstd = rsqrtf(gsum(xh * xh) * (1.0f / CO) + 1e-3f);
    xh *= rstd;
    const float n3o = fmaf(xh, g3o, b3o);
    if (live) A.n3[ACT(t, o, CO, A.Bp, b)] = n3o;
    float n3all[CO];
    dof_static_for<CO>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      n3all[k] = dof_gbcast<k, 16>(n3o);
    });
    // projection + Normal(loc, 1) log-prob; lane o owns outputs j = o + 16 m
    const bool ok = live && A.valid[(int64_t)t * A.Bp + b] != 0.0f;
    const float* __restrict__ xr = A.x + (b * A.T + t) * C3;
    float sq = 0.0f;
    for (int j = o; j < C3; j += 16) {
      float loc = A.bp[j];
#pragma unroll
      for (int q4 = 0; q4 < CO / 4; ++q4) {
        const float4 wv = *reinterpret_cast<const float4*>(wps + j * PS + 4 * q4);
        loc = fmaf(wv.x, n3all[4 * q4], loc); loc = fmaf(wv.y, n3all[4 * q4 + 1], loc);
        loc = fmaf(wv.z, n3all[4 * q4 + 2], loc); loc = fmaf(wv.w, n3all[4 * q4 + 3], loc);
      }
      if (loc != loc) loc = 0.0f;
      loc = fminf(fmaxf(loc, -1e6f), 1e6f);
      float dl = 0.0f;
      if (live) {
        if (A.loc_out) A.loc_out[(b * A.T + t) * C3 + j] = loc;
        const float df = xr[j] - loc;
        sq = fmaf(df, df, sq);
        dl = ok ? -df * inv_bt : NAN;
        if (A.train) A.dloc[ACT(t, j, C3, A.Bp, b)] = dl;
      }
      dls[g * kTailMaxC3 + j] = dl;
    }
    sq = gsum(sq);
    const float LOG_2PI = 1.8378770664093453f;
    if (live && o == 0) nll[0] += ok ? 0.5f * sq + 0.5f * (float)C3 * LOG_2PI : NAN;
    __syncthreads();
    if (A.train) {
      float dn3 = 0.0f;
      for (int j = 0; j < C3; ++j) dn3 = fmaf(wps[j * PS + o], dls[g * kTailMaxC3 + j], dn3);
      const float gg = dn3 * g3o;
      const float mg = gsum(gg) * (1.0f / CO), mgx = gsum(gg * xh) * (1.0f / CO);
      const float d = rstd * (gg - mg - xh * mgx);
      if (live) {
        A.dcv[ACT(t, o, CO, A.Bp, b)] = cvo > 0.0f ? d : 0.0f;
        vals0 += dn3 * xh;
        vals1 += dn3;
      }
    }
  }
  dof_block_colsum<1>(nll, A.recon_partial + blockIdx.x);
  if (A.train) {
    red[g][o] = vals0;
    red[g][CO + o] = vals1;
    __syncthreads();
    if (tid < 2 * CO) {
      float acc = 0.0f;
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) acc += red[rr][tid];
      A.ln3_partial[(int64_t)blockIdx.x * 2 * CO + tid] = acc;
    }
  }
}

// d n2[t][c] = sum_k sum_o wc[o][c][k] * dcv[t-k+2][o], one lane per input channel c (32 lanes per row, 8 rows per
// workgroup), weights and the five gradient rows staged in LDS
__global__ void __launch_bounds__(256) k_dec_conv_bwd_w(const float* __restrict__ dcv, const float* __restrict__ wc,
                                                        float* __restrict__ dn2, int T, int64_t B, int64_t Bp) {
  constexpr int L = 8, CI = 4 * L, CO = 2 * L, RB = 8;
  __shared__ float wcs[5 * (CO / 4) * CI * 4];  // [k][o4][c][4]
  __shared__ float ds[RB * 5 * CO];             // [row][k][o]
  const int tid = threadIdx.x, c = tid & 31, g = tid >> 5;
  for (int e = tid; e < 5 * CI * CO; e += 256) {
    const int k = e % 5, cc = (e / 5) % CI, oo = e / (5 * CI);
    wcs[((k * (CO / 4) + (oo >> 2)) * CI + cc) * 4 + (oo & 3)] = wc[e];
  }
  const int64_t rows = (int64_t)T * B;
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  for (int e = tid; e < RB * 5 * (CO / 4); e += 256) {
    const int o4 = e % (CO / 4), k = (e / (CO / 4)) % 5, rr = e / (5 * (CO / 4));
    const int64_t r = r0 + rr;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r < rows) {
      const int t = (int)(r / B);
      const int64_t b = r - (int64_t)t * B;
      const int ts = t - k + 2;
      if (ts >= 0 && ts < T) v = *reinterpret_cast<const float4*>(dcv + ACT(ts, 4 * o4, CO, Bp, b));
    }
    *reinterpret_cast<float4*>(ds + (rr * 5 + k) * CO + 4 * o4) = v;
  }
  __syncthreads();
  const int64_t r = r0 + g;
  if (r >= rows) return;
  const int t = (int)(r / B);
  const int64_t b = r - (int64_t)t * B;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int o4 = 0; o4 < CO / 4; ++o4) {
      const float4 dv = *reinterpret_cast<const float4*>(ds + (g * 5 + k) * CO + 4 * o4);
      const float4 wv = *reinterpret_cast<const float4*>(wcs + ((k * (CO / 4) + o4) * CI + c) * 4);
      acc = fmaf(wv.x, dv.x, acc); acc = fmaf(wv.y, dv.y, acc); acc = fmaf(wv.z, dv.z, acc); acc = fmaf(wv.w, dv.w, acc);
    }
  dn2[ACT(t, c, CI, Bp, b)] = acc;
}

// TCN decoder, block 0: gradient of the 1x1 residual conv (4L -> 64) back to the repeated input,
// dzrep[t][b][f] += sum_c dsw[c][f] * gres[t][b][c]   (zrep is [T][Bp][ZC], ZC = 32 or 64; gres [T][Bp][64])
__global__ void __launch_bounds__(256) k_dec_ds_bwd(const float* __restrict__ gres, const float* __restrict__ dsw,
                                                    float* __restrict__ dzrep, int C4, int ZC, int T, int64_t B,
                                                    int64_t Bp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * B) return;
  const int t = (int)(i / B);
  const int64_t b = i - (int64_t)t * B;
  float g[64];
  dof_ld_row<64>(gres + ACT(t, 0, 64, Bp, b), g);
  float* o = dzrep + ACT(t, 0, ZC, Bp, b);
  for (int f = 0; f < C4; ++f) {
    float acc = o[f];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc = fmaf(dof_cw(dsw)[c * C4 + f], g[c], acc);
    o[f] = acc;
  }
}

}  // namespace
