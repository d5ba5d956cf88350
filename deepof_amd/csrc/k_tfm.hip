// Transformer encoder / decoder family -- hand-written gfx950 kernels.
//
// Reference modules replaced (paths relative to /root/reference/deepof/clustering):
//   models_new.py:832-840    sinusoidal_positional_encoding   (table built by the host at bind)
//   models_new.py:843-890    MultiHeadAttentionPT             k_tfm_gemm (q|k|v in one GEMM) + k_tfm_attn_fwd/bwd
//   models_new.py:893-919    TransformerEncoderLayerPT        k_tfm_add_ln_fwd / k_tfm_ln_bwd around the GEMMs
//   models_new.py:922-982    TransformerCorePT                k_tfm_embed(_bwd), k_tfm_last_fwd/bwd
//   models_new.py:1160-1162  train-time batch standardisation k_tfm_bstd_fwd/bwd
//   models_new.py:1167-1267  TFMDecoderPT                     k_tfm_dec_expand_fwd/bwd, k_tfm_dec_h0, k_tfm_dec_logp
//   models_new.py:1270-1327  CausalSelfAttentionLayer         same GEMM / attention / LayerNorm kernels (causal, GELU)
//
// Layout: every per-time-step tensor is [t][s][C] (row r = t*Sp + s, channel-minor, Sp = sequences padded to 64;
// rows with s >= S are never written and stay zero from the bind, which the weight-gradient reductions rely on).
// All dense layers are one kernel: rows x K times K x N on v_mfma_f32_16x16x4_f32 (exact fp32), the weight matrix
// staged once per workgroup in LDS in B-operand order, A operands as 16-byte row loads (the k index is permuted
// consistently on both operands so that a lane's float4 feeds four consecutive MFMAs).  Attention on windows <= 64
// and head sizes 1..16: one thread per (sequence, head, query), q/k/v of a few sequences staged in LDS, the backward pass
// recomputes the probabilities (nothing but q, k, v and the output is stored); longer windows: k_tfm_attn_*_long.
// Dropout keep-masks are a counter-based hash of (site seed, device step counter, element index in the reference's
// tensor order) evaluated where needed in forward and backward -- or read from an injected byte mask (parity tests).
#include <cmath>

#include "dof_rt.h"
#include "launchers.h"

#define TRY_RC(x) do { int _rc = (x); if (_rc != DOF_OK) return _rc; } while (0)

namespace {

constexpr int kGemmLds = 12288;  // floats: KC * NT * 256 <= kGemmLds
constexpr int kAttLds = 16384;   // floats: the attention kernels' LDS image of nseq sequences (large class, 64 KB)
// small classes (round 5): the kernels are latency-bound at one or two workgroups per CU; at C2's shape a 24 KB (forward, two
// sequences) / 36 KB (backward, two sequences) image runs 4 - 6 workgroups per CU: the transformer step 6.80 -> 6.37 ms.  A launch
// takes the small class when at least one sequence fits it.
constexpr int kAttLdsFwdSmall = 6144, kAttLdsBwdSmall = 9216;
constexpr int kAttLongStats = 12288;   // floats: 3 per (head, row) of one sequence in k_tfm_attn_bwd_long (heads x window <= 4096)

__device__ __forceinline__ float drop_scale(const DofDrop& d, uint32_t ctr, int64_t idx) {
  if (d.scale == 0.0f) return 1.0f;
  if (d.inject) return d.inject[idx] ? d.scale : 0.0f;
  uint32_t h = (uint32_t)idx * 0x9E3779B1u ^ ((uint32_t)((uint64_t)idx >> 32) * 0x85EBCA77u) ^ d.seed ^ (ctr * 0xC2B2AE3Du);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h >= d.thresh ? d.scale : 0.0f;
}
__device__ __forceinline__ uint32_t drop_ctr(const DofDrop& d) { return d.ctr ? *d.ctr : 0u; }

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---------------------------------------------------------------------------------------------
// embedding: scrambled read of the window tensor (models_new.py:1108-1111), Linear(F -> D) + ReLU, * sqrt(D), + PE,
// dropout.  thread = (row, channel quad)
// ---------------------------------------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(256) k_tfm_embed(const float* __restrict__ xin, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ pe,
                                                   float* __restrict__ xs, float* __restrict__ pad, float* __restrict__ y,
                                                   DofDrop drop, float sqrt_d, int T, int G, int D, int64_t S, int64_t Sp) {
  const int Q = D / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * S * Q) return;
  const int q = (int)(i % Q);
  const int64_t rs = i / Q;
  const int t = (int)(rs / S);
  const int64_t s = rs - (int64_t)t * S;
  const int64_t b = s / G;
  const int g = (int)(s - b * G);
  const float* __restrict__ win = xin + b * (int64_t)T * G * F;
  float xv[F];
  bool any = false;
#pragma unroll
  for (int f = 0; f < F; ++f) {  // y[b,g,t,f] = x[b, t', cc] with cc*T + t' = (f*T + t)*G + g
    const int lin = (f * T + t) * G + g;
    const int cc = lin / T;
    xv[f] = win[(int64_t)(lin - cc * T) * G * F + cc];
    any |= xv[f] != 0.0f;
  }
  if (q == 0) {
#pragma unroll
    for (int f = 0; f < F; ++f) xs[ACT(t, f, F, Sp, s)] = xv[f];
    pad[(int64_t)t * Sp + s] = any ? 0.0f : 1.0f;
  }
  const uint32_t ctr = drop_ctr(drop);
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * q + j;
    float acc = bias[c];
#pragma unroll
    for (int f = 0; f < F; ++f) acc = fmaf(w[c * F + f], xv[f], acc);
    acc = fmaxf(acc, 0.0f) * sqrt_d + pe[t * D + c];
    o[j] = acc * drop_scale(drop, ctr, (s * T + t) * (int64_t)D + c);
  }
  *reinterpret_cast<float4*>(y + ACT(t, 4 * q, D, Sp, s)) = make_float4(o[0], o[1], o[2], o[3]);
}

// gradient entering the embedding's pre-activation (A operand of its weight-gradient job)
template <int F>
__global__ void __launch_bounds__(256) k_tfm_embed_bwd(const float* __restrict__ xs, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ dy,
                                                       float* __restrict__ dpre, DofDrop drop, float sqrt_d, int T,
                                                       int D, int64_t S, int64_t Sp) {
  const int Q = D / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * S * Q) return;
  const int q = (int)(i % Q);
  const int64_t rs = i / Q;
  const int t = (int)(rs / S);
  const int64_t s = rs - (int64_t)t * S;
  float xv[F];
#pragma unroll
  for (int f = 0; f < F; ++f) xv[f] = xs[ACT(t, f, F, Sp, s)];
  const uint32_t ctr = drop_ctr(drop);
  const float4 g4 = *reinterpret_cast<const float4*>(dy + ACT(t, 4 * q, D, Sp, s));
  const float g[4] = {g4.x, g4.y, g4.z, g4.w};
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * q + j;
    float acc = bias[c];
#pragma unroll
    for (int f = 0; f < F; ++f) acc = fmaf(w[c * F + f], xv[f], acc);
    o[j] = acc > 0.0f ? g[j] * drop_scale(drop, ctr, (s * T + t) * (int64_t)D + c) * sqrt_d : 0.0f;
  }
  *reinterpret_cast<float4*>(dpre + ACT(t, 4 * q, D, Sp, s)) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------------------------
// rows x K  @  K x N  on the matrix cores
// ---------------------------------------------------------------------------------------------
template <int NTMAX, int EPI>
__global__ void __launch_bounds__(256) k_tfm_gemm(DofGemm A) {
  __shared__ float wl[kGemmLds];
  // NT = the template's tile count: tiles beyond N hold zeros and are multiplied unconditionally (a per-tile branch
  // inside the MFMA loop made the compiler shuffle the accumulators through thousands of register copies)
  constexpr int NT = NTMAX;
  const int K = A.K, N = A.N;
  const int KC = (K + 15) / 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < KC * 4 * NT * 64; e += 256) {
    const int lane = e & 63;
    int rest = e >> 6;
    const int nt = rest % NT;
    rest /= NT;
    const int m = rest & 3, jc = rest >> 2;
    const int n = nt * 16 + (lane & 15), k = jc * 16 + 4 * (lane >> 4) + m;
    float v = 0.0f;
    if (n < N && k < K) v = A.trans ? A.W[(int64_t)k * A.ldw + n] : A.W[(int64_t)n * A.ldw + k];
    wl[e] = v;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int64_t tiles = (int64_t)A.T * (A.Sp / 16);
  [[maybe_unused]] const uint32_t ctr = drop_ctr(A.drop);
  // A operands: four 16-column chunks (one float4 per lane each) are in flight together, and the next group -- or the
  // first group of the wave's next tile -- is requested before the current group's MFMAs issue (PMC of the
  // load-then-multiply form: 58 % of the wave-cycles parked on vmcnt, MFMA pipe 25 % busy)
  const int G = (KC + 3) / 4;
  auto load_group = [&](const float* __restrict__ xr, int g, float4 (&buf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jc = g * 4 + j;
      buf[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (jc < KC && jc * 16 + 4 * lg < K) buf[j] = *reinterpret_cast<const float4*>(xr + jc * 16);
    }
  };
  const int ITS = A.its;  // row tiles per wave (1..4): small problems spread over more workgroups
  auto tile_of = [&](int it) { return (int64_t)blockIdx.x * (4 * ITS) + wave + 4 * it; };
  auto tile_live = [&](int it) {
    const int64_t tl = tile_of(it);
    return it < ITS && tl < tiles && (tl * 16) % A.Sp < A.S;
  };
  float4 cur[4], nxt[4];
  int it = 0;
  while (it < ITS && !tile_live(it)) ++it;
  if (it < ITS) load_group(A.X + (tile_of(it) * 16 + li) * A.ldx + 4 * lg, 0, cur);
  while (it < ITS) {
    const int64_t row0 = tile_of(it) * 16;
    const int64_t s0 = row0 % A.Sp;
    int it_next = it + 1;
    while (it_next < ITS && !tile_live(it_next)) ++it_next;
    dof_f32x4 acc[NTMAX];
#pragma unroll
    for (int n = 0; n < NTMAX; ++n) acc[n] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float* __restrict__ xr = A.X + (row0 + li) * A.ldx + 4 * lg;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) load_group(xr, g + 1, nxt);
      else if (it_next < ITS) load_group(A.X + (tile_of(it_next) * 16 + li) * A.ldx + 4 * lg, 0, nxt);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (g * 4 + j < KC) {  // (wave-uniform)
          const float av[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
          const float* __restrict__ wrow = wl + ((g * 4 + j) * 4 * NT) * 64 + lane;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < NTMAX; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], wrow[(m * NT + n) * 64], acc[n], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    it = it_next;
    // epilogue: lane (li, lg) holds rows row0 + 4 lg + r (r < 4), column n * 16 + li of tile n
    const int64_t rbase = row0 + 4 * lg;
    float* __restrict__ ybase = A.Y + rbase * A.ldy + li;
    const int nrow = (int)(A.S - (s0 + 4 * lg));  // valid rows of this lane's four (<= 0: none)
    [[maybe_unused]] const float* __restrict__ xbase = nullptr;
    if constexpr (EPI == DOF_EPI_MUL_RELU || EPI == DOF_EPI_MUL_DGELU) xbase = A.aux + rbase * A.ldaux + li;
    [[maybe_unused]] float* __restrict__ pbase = nullptr;
    if constexpr (EPI == DOF_EPI_GELU) pbase = A.aux_out + rbase * A.ldy + li;
    [[maybe_unused]] const int t = (int)(row0 / A.Sp);
#pragma unroll
    for (int n = 0; n < NTMAX; ++n) {
      const int col = n * 16 + li;
      if (col >= N) continue;
      const float bv = A.bias ? A.bias[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= nrow) continue;
        float v = acc[n][r] + bv;
        if constexpr (EPI == DOF_EPI_RELU) v = fmaxf(v, 0.0f);
        if constexpr (EPI == DOF_EPI_GELU) {
          pbase[r * A.ldy + n * 16] = v;
          v = gelu_f(v) * drop_scale(A.drop, ctr, ((s0 + 4 * lg + r) * A.T + t) * (int64_t)A.drop_ld + col);
        }
        if constexpr (EPI == DOF_EPI_MUL_RELU) v = xbase[r * A.ldaux + n * 16] > 0.0f ? v : 0.0f;
        if constexpr (EPI == DOF_EPI_MUL_DGELU)
          v *= dgelu_f(xbase[r * A.ldaux + n * 16]) *
               drop_scale(A.drop, ctr, ((s0 + 4 * lg + r) * A.T + t) * (int64_t)A.drop_ld + col);
        float* __restrict__ dst = ybase + r * A.ldy + n * 16;
        *dst = A.accumulate ? *dst + v : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------
// stage q|k|v rows ([t][s][3D]) of `nseq` neighbouring sequences: dst[t][seq][W]
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src, int W, int T, int nseq,
                                           int64_t s0, int64_t S, int64_t Sp, int tid, int nthr) {
  const int n4 = nseq * W / 4;
  // four loads in flight per thread before the first LDS store (the one-at-a-time form paid a memory latency per 16 bytes)
  for (int e0 = tid; e0 < T * n4; e0 += 4 * nthr) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * nthr;
      v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (e < T * n4) {
        const int t = e / n4, j = e - t * n4;
        if (s0 + (j * 4) / W < S) v[u] = *reinterpret_cast<const float4*>(src + ((int64_t)t * Sp + s0) * W + j * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * nthr;
      if (e < T * n4) {
        const int t = e / n4, j = e - t * n4;
        *reinterpret_cast<float4*>(dst + (int64_t)t * nseq * W + j * 4) = v[u];
      }
    }
  }
}

template <int DH, int TMAX, int LDSF>
__global__ void __launch_bounds__(512) k_tfm_attn_fwd(DofAttn A) {
  __shared__ float sm[LDSF];
  const int T = A.T, D = A.D, H = A.H, nseq = A.nseq, W = 3 * D;
  const int64_t s0 = (int64_t)blockIdx.x * nseq;
  const int tid = threadIdx.x, nthr = blockDim.x;
  float* __restrict__ sq = sm;
  float* __restrict__ spad = sm + T * nseq * W;
  stage_rows(sq, A.qkv, W, T, nseq, s0, A.S, A.Sp, tid, nthr);
  for (int e = tid; e < nseq * T; e += nthr) {
    const int seq = e / T, t = e - seq * T;
    spad[e] = (A.pad && s0 + seq < A.S) ? A.pad[(int64_t)t * A.Sp + s0 + seq] : 0.0f;
  }
  __syncthreads();
  const int tq = tid % T, h = (tid / T) % H, seq = tid / (T * H);
  if (seq >= nseq || s0 + seq >= A.S) return;
  if (A.q_last && tq != T - 1) return;
  const float scale = A.scale;
  float q[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) q[d] = sq[(tq * nseq + seq) * W + h * DH + d] * scale;
  float p[TMAX];
  float mx = -INFINITY;
#pragma unroll
  for (int tk = 0; tk < TMAX; ++tk) {
    float s = -INFINITY;
    if (tk < T && !(A.causal && tk > tq) && spad[seq * T + tk] == 0.0f) {
      const float* __restrict__ kr = sq + (tk * nseq + seq) * W + D + h * DH;
      s = 0.0f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], kr[d], s);
    }
    p[tk] = s;
    mx = fmaxf(mx, s);
  }
  float l = 0.0f;
#pragma unroll
  for (int tk = 0; tk < TMAX; ++tk) {
    p[tk] = tk < T ? __expf(p[tk] - mx) : 0.0f;  // all keys masked: exp(-inf + inf) = NaN, as the reference's softmax
    l += p[tk];
  }
  const float inv = 1.0f / l;
  const uint32_t ctr = drop_ctr(A.drop);
  const int64_t dbase = (((s0 + seq) * H + h) * T + tq) * (int64_t)T;
  float o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.0f;
#pragma unroll
  for (int tk = 0; tk < TMAX; ++tk) {
    if (tk >= T) continue;
    const float pw = p[tk] * inv * drop_scale(A.drop, ctr, dbase + tk);
    const float* __restrict__ vr = sq + (tk * nseq + seq) * W + 2 * D + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = fmaf(pw, vr[d], o[d]);
  }
  float* __restrict__ dst = A.ao + ((int64_t)tq * A.Sp + s0 + seq) * D + h * DH;
#pragma unroll
  for (int d = 0; d < DH; ++d) dst[d] = o[d];
}

template <int DH, int TMAX, int LDSF>
__global__ void __launch_bounds__(512) k_tfm_attn_bwd(DofAttn A) {
  __shared__ float sm[LDSF];
  const int T = A.T, D = A.D, H = A.H, nseq = A.nseq, W = 3 * D;
  const int64_t s0 = (int64_t)blockIdx.x * nseq;
  const int tid = threadIdx.x, nthr = blockDim.x;
  float* __restrict__ sq = sm;
  float* __restrict__ sdo = sq + T * nseq * W;
  float* __restrict__ spad = sdo + T * nseq * D;
  float* __restrict__ sst = spad + nseq * T;  // [seq][h][t][3]: row max, 1 / row sum, sum_k P dP
  stage_rows(sq, A.qkv, W, T, nseq, s0, A.S, A.Sp, tid, nthr);
  stage_rows(sdo, A.dao, D, T, nseq, s0, A.S, A.Sp, tid, nthr);
  for (int e = tid; e < nseq * T; e += nthr) {
    const int seq = e / T, t = e - seq * T;
    spad[e] = (A.pad && s0 + seq < A.S) ? A.pad[(int64_t)t * A.Sp + s0 + seq] : 0.0f;
  }
  __syncthreads();
  const int tx = tid % T, h = (tid / T) % H, seq = tid / (T * H);
  const bool live = seq < nseq && s0 + seq < A.S;
  const float scale = A.scale;
  const uint32_t ctr = drop_ctr(A.drop);
  if (live && (!A.q_last || tx == T - 1)) {  // ---- phase A: thread = query row tx
    float q[DH], go[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      q[d] = sq[(tx * nseq + seq) * W + h * DH + d] * scale;
      go[d] = sdo[(tx * nseq + seq) * D + h * DH + d];
    }
    float p[TMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int tk = 0; tk < TMAX; ++tk) {
      float s = -INFINITY;
      if (tk < T && !(A.causal && tk > tx) && spad[seq * T + tk] == 0.0f) {
        const float* __restrict__ kr = sq + (tk * nseq + seq) * W + D + h * DH;
        s = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s = fmaf(q[d], kr[d], s);
      }
      p[tk] = s;
      mx = fmaxf(mx, s);
    }
    float l = 0.0f;
#pragma unroll
    for (int tk = 0; tk < TMAX; ++tk) {
      p[tk] = tk < T ? __expf(p[tk] - mx) : 0.0f;
      l += p[tk];
    }
    const float inv = 1.0f / l;
    const int64_t dbase = (((s0 + seq) * H + h) * T + tx) * (int64_t)T;
    float dp[TMAX];
    float drow = 0.0f;
#pragma unroll
    for (int tk = 0; tk < TMAX; ++tk) {
      dp[tk] = 0.0f;
      if (tk >= T) continue;
      p[tk] *= inv;
      const float* __restrict__ vr = sq + (tk * nseq + seq) * W + 2 * D + h * DH;
      float a = 0.0f;
#pragma unroll
      for (int d = 0; d < DH; ++d) a = fmaf(go[d], vr[d], a);
      dp[tk] = a * drop_scale(A.drop, ctr, dbase + tk);
      drow = fmaf(p[tk], dp[tk], drow);
    }
    float dq[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.0f;
#pragma unroll
    for (int tk = 0; tk < TMAX; ++tk) {
      if (tk >= T) continue;
      const float ds = p[tk] * (dp[tk] - drow);
      const float* __restrict__ kr = sq + (tk * nseq + seq) * W + D + h * DH;
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
    }
    float* __restrict__ dst = A.dqkv + ((int64_t)tx * A.Sp + s0 + seq) * W + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = dq[d] * scale;
    float* __restrict__ st = sst + ((seq * H + h) * T + tx) * 3;
    st[0] = mx; st[1] = inv; st[2] = drow;
  }
  __syncthreads();
  if (!live) return;
  {  // ---- phase B: thread = key / value row tx
    float kk[DH], vv[DH], dk[DH], dv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      kk[d] = sq[(tx * nseq + seq) * W + D + h * DH + d];
      vv[d] = sq[(tx * nseq + seq) * W + 2 * D + h * DH + d];
      dk[d] = 0.0f;
      dv[d] = 0.0f;
    }
    const bool key_masked = spad[seq * T + tx] != 0.0f;
    for (int tq = (A.q_last ? T - 1 : (A.causal ? tx : 0)); tq < T; ++tq) {
      const float* __restrict__ st = sst + ((seq * H + h) * T + tq) * 3;
      const float* __restrict__ qr = sq + (tq * nseq + seq) * W + h * DH;
      const float* __restrict__ gr = sdo + (tq * nseq + seq) * D + h * DH;
      float s = 0.0f, a = 0.0f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        s = fmaf(qr[d], kk[d], s);
        a = fmaf(gr[d], vv[d], a);
      }
      float pw = key_masked ? 0.0f : __expf(s * scale - st[0]) * st[1];
      if (key_masked && !(st[0] > -INFINITY)) pw = NAN;  // a row with every key masked is NaN in the reference
      const float ks = drop_scale(A.drop, ctr, (((s0 + seq) * H + h) * T + tq) * (int64_t)T + tx);
      const float ds = pw * (a * ks - st[2]) * scale;
      const float pd = pw * ks;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        dk[d] = fmaf(ds, qr[d], dk[d]);
        dv[d] = fmaf(pd, gr[d], dv[d]);
      }
    }
    float* __restrict__ dst = A.dqkv + ((int64_t)tx * A.Sp + s0 + seq) * W + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      dst[D + d] = dk[d];
      dst[2 * D + d] = dv[d];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Attention for windows the LDS-resident kernels above do not take (T > 64, or a window x width product beyond their
// 64 KB; round 5).  One sequence per workgroup; a thread owns (row, head) pairs and walks the other rows straight from
// global memory (a sequence's q | k | v rows are T * 3 D floats -- 98 KB at T = 128, D = 64 -- read by the workgroup's own
// threads only, i.e. from its L1 / L2), with a running maximum instead of a row of probabilities in registers.  Same
// arithmetic per element as the resident kernels (scaled q, fmaf chains over the head width, __expf, 1 / sum, the
// dropout factor on the normalised probability), the sums over keys in key order.  No window limit; the model's default
// windows never come here.
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ void __launch_bounds__(256) k_tfm_attn_fwd_long(DofAttn A) {
  const int T = A.T, D = A.D, H = A.H, W = 3 * D;
  const int64_t s = blockIdx.x;
  if (s >= A.S) return;
  const float scale = A.scale;
  const uint32_t ctr = drop_ctr(A.drop);
  auto row = [&](int t) { return A.qkv + ((int64_t)t * A.Sp + s) * W; };
  auto masked = [&](int tq, int tk) { return (A.causal && tk > tq) || (A.pad && A.pad[(int64_t)tk * A.Sp + s] != 0.0f); };
  for (int item = threadIdx.x; item < T * H; item += blockDim.x) {
    const int tq = item % T, h = item / T;
    if (A.q_last && tq != T - 1) continue;
    float q[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) q[d] = row(tq)[h * DH + d] * scale;
    float mx = -INFINITY;
    for (int tk = 0; tk < T; ++tk) {
      if (masked(tq, tk)) continue;
      const float* __restrict__ kr = row(tk) + D + h * DH;
      float sc = 0.0f;
#pragma unroll
      for (int d = 0; d < DH; ++d) sc = fmaf(q[d], kr[d], sc);
      mx = fmaxf(mx, sc);
    }
    float l = 0.0f;
    for (int tk = 0; tk < T; ++tk) {
      float sc = -INFINITY;
      if (!masked(tq, tk)) {
        const float* __restrict__ kr = row(tk) + D + h * DH;
        sc = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sc = fmaf(q[d], kr[d], sc);
      }
      l += __expf(sc - mx);   // all keys masked: exp(-inf + inf) = NaN, as the reference's softmax
    }
    const float inv = 1.0f / l;
    const int64_t dbase = ((s * H + h) * T + tq) * (int64_t)T;
    float o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.0f;
    for (int tk = 0; tk < T; ++tk) {
      float sc = -INFINITY;
      const float* __restrict__ kr = row(tk) + D + h * DH;
      if (!masked(tq, tk)) {
        sc = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sc = fmaf(q[d], kr[d], sc);
      }
      const float pw = __expf(sc - mx) * inv * drop_scale(A.drop, ctr, dbase + tk);
      const float* __restrict__ vr = kr + D;
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(pw, vr[d], o[d]);
    }
    float* __restrict__ dst = A.ao + ((int64_t)tq * A.Sp + s) * D + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = o[d];
  }
}

// backward: phase A (thread = query row): row maximum, 1 / row sum, sum_k P dP, then dQ; the three row statistics go to
// LDS ([h][t][3], 12 B per row and head).  Phase B (thread = key / value row): dK, dV over the query rows.
template <int DH>
__global__ void __launch_bounds__(256) k_tfm_attn_bwd_long(DofAttn A) {
  __shared__ float sst[kAttLongStats];
  const int T = A.T, D = A.D, H = A.H, W = 3 * D;
  const int64_t s = blockIdx.x;
  if (s >= A.S) return;
  const float scale = A.scale;
  const uint32_t ctr = drop_ctr(A.drop);
  auto row = [&](int t) { return A.qkv + ((int64_t)t * A.Sp + s) * W; };
  auto grow = [&](int t) { return A.dao + ((int64_t)t * A.Sp + s) * D; };
  auto padded = [&](int tk) { return A.pad && A.pad[(int64_t)tk * A.Sp + s] != 0.0f; };
  for (int item = threadIdx.x; item < T * H; item += blockDim.x) {
    const int tx = item % T, h = item / T;
    if (A.q_last && tx != T - 1) continue;
    float q[DH], go[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      q[d] = row(tx)[h * DH + d] * scale;
      go[d] = grow(tx)[h * DH + d];
    }
    auto score = [&](int tk) {
      float sc = -INFINITY;
      if (!(A.causal && tk > tx) && !padded(tk)) {
        const float* __restrict__ kr = row(tk) + D + h * DH;
        sc = 0.0f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sc = fmaf(q[d], kr[d], sc);
      }
      return sc;
    };
    float mx = -INFINITY;
    for (int tk = 0; tk < T; ++tk) mx = fmaxf(mx, score(tk));
    float l = 0.0f;
    for (int tk = 0; tk < T; ++tk) l += __expf(score(tk) - mx);
    const float inv = 1.0f / l;
    const int64_t dbase = ((s * H + h) * T + tx) * (int64_t)T;
    auto dprob = [&](int tk) {
      const float* __restrict__ vr = row(tk) + 2 * D + h * DH;
      float a = 0.0f;
#pragma unroll
      for (int d = 0; d < DH; ++d) a = fmaf(go[d], vr[d], a);
      return a * drop_scale(A.drop, ctr, dbase + tk);
    };
    float drow = 0.0f;
    for (int tk = 0; tk < T; ++tk) drow = fmaf(__expf(score(tk) - mx) * inv, dprob(tk), drow);
    float dq[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.0f;
    for (int tk = 0; tk < T; ++tk) {
      const float ds = __expf(score(tk) - mx) * inv * (dprob(tk) - drow);
      const float* __restrict__ kr = row(tk) + D + h * DH;
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
    }
    float* __restrict__ dst = A.dqkv + ((int64_t)tx * A.Sp + s) * W + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = dq[d] * scale;
    float* __restrict__ st = sst + (h * T + tx) * 3;
    st[0] = mx; st[1] = inv; st[2] = drow;
  }
  __syncthreads();
  for (int item = threadIdx.x; item < T * H; item += blockDim.x) {
    const int tx = item % T, h = item / T;
    float kk[DH], vv[DH], dk[DH], dv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      kk[d] = row(tx)[D + h * DH + d];
      vv[d] = row(tx)[2 * D + h * DH + d];
      dk[d] = 0.0f;
      dv[d] = 0.0f;
    }
    const bool key_masked = padded(tx);
    for (int tq = (A.q_last ? T - 1 : (A.causal ? tx : 0)); tq < T; ++tq) {
      const float* __restrict__ st = sst + (h * T + tq) * 3;
      const float* __restrict__ qr = row(tq) + h * DH;
      const float* __restrict__ gr = grow(tq) + h * DH;
      float sc = 0.0f, a = 0.0f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        sc = fmaf(qr[d], kk[d], sc);
        a = fmaf(gr[d], vv[d], a);
      }
      float pw = key_masked ? 0.0f : __expf(sc * scale - st[0]) * st[1];
      if (key_masked && !(st[0] > -INFINITY)) pw = NAN;  // a row with every key masked is NaN in the reference
      const float ks = drop_scale(A.drop, ctr, ((s * H + h) * T + tq) * (int64_t)T + tx);
      const float ds = pw * (a * ks - st[2]) * scale;
      const float pd = pw * ks;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        dk[d] = fmaf(ds, qr[d], dk[d]);
        dv[d] = fmaf(pd, gr[d], dv[d]);
      }
    }
    float* __restrict__ dst = A.dqkv + ((int64_t)tx * A.Sp + s) * W + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      dst[D + d] = dk[d];
      dst[2 * D + d] = dv[d];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// residual + dropout + LayerNorm(eps 1e-6): u = x + drop(h), y = LN(u).  LPR lanes own one row (a 16-byte word each).
// ---------------------------------------------------------------------------------------------
template <int C>
struct LnGeom {
  static constexpr int LPR = C <= 32 ? 8 : 16;
  static constexpr int RPB = 256 / LPR;  // rows per block pass
  static constexpr int PASSES = 16;
};

template <int C>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int m = 1; m < LnGeom<C>::LPR; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int C>
__global__ void __launch_bounds__(256) k_tfm_add_ln_fwd(DofLn A) {
  constexpr int LPR = LnGeom<C>::LPR, RPB = LnGeom<C>::RPB, PASSES = LnGeom<C>::PASSES;
  const int j = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const bool lane_on = 4 * j < C;
  const uint32_t ctr = drop_ctr(A.drop);
  const int64_t rows = (int64_t)A.T * A.Sp;
  float4 gm = make_float4(0, 0, 0, 0), bt = gm;
  if (A.gamma && lane_on) {
    gm = *reinterpret_cast<const float4*>(A.gamma + 4 * j);
    bt = *reinterpret_cast<const float4*>(A.beta + 4 * j);
  }
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t row = ((int64_t)blockIdx.x * PASSES + ps) * RPB + rl;
    const int64_t s = row % A.Sp;
    const bool on = row < rows && s < A.S && lane_on;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (on) {
      const float4 x4 = *reinterpret_cast<const float4*>(A.x + row * C + 4 * j);
      v[0] = x4.x; v[1] = x4.y; v[2] = x4.z; v[3] = x4.w;
      if (A.h) {
        const float4 h4 = *reinterpret_cast<const float4*>(A.h + row * C + 4 * j);
        const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
        const int t = (int)(row / A.Sp) + A.t_off;
        const int64_t base = (s * (A.T_idx ? A.T_idx : A.T) + t) * (int64_t)C + 4 * j;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaf(hv[c], drop_scale(A.drop, ctr, base + c), v[c]);
      }
      if (A.u) *reinterpret_cast<float4*>(A.u + row * C + 4 * j) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (!A.gamma) continue;  // (uniform) plain residual add
    const float mean = row_sum<C>(v[0] + v[1] + v[2] + v[3]) * (1.0f / C);
    float d[4], sq = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      d[c] = lane_on ? v[c] - mean : 0.0f;
      sq = fmaf(d[c], d[c], sq);
    }
    const float rstd = rsqrtf(row_sum<C>(sq) * (1.0f / C) + A.eps);
    if (on) {
      const float g[4] = {gm.x, gm.y, gm.z, gm.w}, b[4] = {bt.x, bt.y, bt.z, bt.w};
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = fmaf(d[c] * rstd, g[c], b[c]);
      *reinterpret_cast<float4*>(A.y + row * C + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// backward: g = dy1 (+ dy2) is the gradient of y = LN(u); du = LN'(g) (+ dres, a gradient arriving at u directly);
// du -> A.du, du * keep-scale -> A.dh; per-block sums of (g * xhat | g) -> partial[blk][2C].  gamma == null: no
// LayerNorm (plain residual add): du = dres.
template <int C>
__global__ void __launch_bounds__(256) k_tfm_ln_bwd(DofLnBwd A) {
  constexpr int LPR = LnGeom<C>::LPR, RPB = LnGeom<C>::RPB, PASSES = LnGeom<C>::PASSES;
  __shared__ float red[RPB][2 * C + 1];
  const int j = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const bool lane_on = 4 * j < C;
  const uint32_t ctr = drop_ctr(A.drop);
  const int64_t rows = (int64_t)A.T * A.Sp;
  float4 gm = make_float4(0, 0, 0, 0);
  if (A.gamma && lane_on) gm = *reinterpret_cast<const float4*>(A.gamma + 4 * j);
  const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
  float pg[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0};
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t row = ((int64_t)blockIdx.x * PASSES + ps) * RPB + rl;
    const int64_t s = row % A.Sp;
    const bool on = row < rows && s < A.S && lane_on;
    float du[4] = {0, 0, 0, 0};
    if (A.gamma) {
      float u[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
      if (on) {
        const float4 u4 = *reinterpret_cast<const float4*>(A.u + row * C + 4 * j);
        u[0] = u4.x; u[1] = u4.y; u[2] = u4.z; u[3] = u4.w;
        const float4 a4 = *reinterpret_cast<const float4*>(A.dy1 + row * C + 4 * j);
        g[0] = a4.x; g[1] = a4.y; g[2] = a4.z; g[3] = a4.w;
        if (A.dy2) {
          const float4 b4 = *reinterpret_cast<const float4*>(A.dy2 + row * C + 4 * j);
          g[0] += b4.x; g[1] += b4.y; g[2] += b4.z; g[3] += b4.w;
        }
      }
      const float mean = row_sum<C>(u[0] + u[1] + u[2] + u[3]) * (1.0f / C);
      float xh[4], sq = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xh[c] = lane_on ? u[c] - mean : 0.0f;
        sq = fmaf(xh[c], xh[c], sq);
      }
      const float rstd = rsqrtf(row_sum<C>(sq) * (1.0f / C) + A.eps);
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xh[c] *= rstd;
        pg[c] = fmaf(g[c], xh[c], pg[c]);
        pb[c] += g[c];
        g[c] *= gmv[c];
        s1 += g[c];
        s2 = fmaf(g[c], xh[c], s2);
      }
      s1 = row_sum<C>(s1) * (1.0f / C);
      s2 = row_sum<C>(s2) * (1.0f / C);
#pragma unroll
      for (int c = 0; c < 4; ++c) du[c] = rstd * (g[c] - s1 - xh[c] * s2);
    }
    if (on) {
      if (A.dres) {
        const float4 r4 = *reinterpret_cast<const float4*>(A.dres + row * C + 4 * j);
        du[0] += r4.x; du[1] += r4.y; du[2] += r4.z; du[3] += r4.w;
      }
      if (A.du) *reinterpret_cast<float4*>(A.du + row * C + 4 * j) = make_float4(du[0], du[1], du[2], du[3]);
      if (A.dh) {
        const int t = (int)(row / A.Sp) + A.t_off;
        const int64_t base = (s * (A.T_idx ? A.T_idx : A.T) + t) * (int64_t)C + 4 * j;
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = du[c] * drop_scale(A.drop, ctr, base + c);
        *reinterpret_cast<float4*>(A.dh + row * C + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  if (!A.partial) return;
  if (lane_on) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      red[rl][4 * j + c] = pg[c];
      red[rl][C + 4 * j + c] = pb[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    float acc = 0.0f;
#pragma unroll 8
    for (int r = 0; r < RPB; ++r) acc += red[r][threadIdx.x];
    A.partial[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = acc;
  }
}

// last time step of the final layer -> CensNet input [D][Sp]; and its adjoint (full [r][D] gradient, zero elsewhere)
__global__ void __launch_bounds__(256) k_tfm_last_fwd(const float* __restrict__ x, float* __restrict__ n2, int T, int D,
                                                      int64_t S, int64_t Sp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * D) return;
  const int64_t s = i / D;
  const int c = (int)(i - s * D);
  n2[(int64_t)c * Sp + s] = x[ACT(T - 1, c, D, Sp, s)];
}
__global__ void __launch_bounds__(256) k_tfm_last_bwd(const float* __restrict__ dn2, float* __restrict__ dx, int T, int D,
                                                      int64_t S, int64_t Sp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * Sp * D) return;
  const int c = (int)(i % D);
  const int64_t r = i / D;
  const int64_t s = r % Sp;
  dx[i] = (r / Sp == T - 1 && s < S) ? dn2[(int64_t)c * Sp + s] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// batch standardisation of the encoder output in train mode (models_new.py:1160-1162): one block per channel.
// stat[c] = (mean, 1 / max(std, 0.1), std >= 0.1)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) k_tfm_bstd_fwd(const float* __restrict__ in, float* __restrict__ out,
                                                      float* __restrict__ stat, int standardize, int64_t B, int64_t Bp) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const float* __restrict__ x = in + (int64_t)c * Bp;
  if (!standardize) {  // eval mode (or a batch of one): identity
    for (int64_t b = threadIdx.x; b < B; b += 256) out[(int64_t)c * Bp + b] = x[b];
    return;
  }
  float a = 0.0f;
  for (int64_t b = threadIdx.x; b < B; b += 256) a += x[b];
  const float mean = block_sum256(a, red) / (float)B;
  a = 0.0f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float d = x[b] - mean;
    a = fmaf(d, d, a);
  }
  const float sd = sqrtf(block_sum256(a, red) / (float)(B - 1));
  const float inv = 1.0f / fmaxf(sd, 0.1f);
  for (int64_t b = threadIdx.x; b < B; b += 256) out[(int64_t)c * Bp + b] = (x[b] - mean) * inv;
  if (threadIdx.x == 0) {
    stat[3 * c] = mean;
    stat[3 * c + 1] = inv;
    stat[3 * c + 2] = sd >= 0.1f ? 1.0f : 0.0f;
  }
}

__global__ void __launch_bounds__(256) k_tfm_bstd_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ stat, float* __restrict__ dx,
                                                      int standardize, int64_t B, int64_t Bp) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const float* __restrict__ g = dy + (int64_t)c * Bp;
  if (!standardize) {
    for (int64_t b = threadIdx.x; b < B; b += 256) dx[(int64_t)c * Bp + b] = g[b];
    return;
  }
  const float* __restrict__ yh = y + (int64_t)c * Bp;
  float a = 0.0f, d = 0.0f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    a += g[b];
    d = fmaf(g[b], yh[b], d);
  }
  const float gm = block_sum256(a, red) / (float)B;
  const float gy = stat[3 * c + 2] != 0.0f ? block_sum256(d, red) / (float)(B - 1) : 0.0f;
  const float inv = stat[3 * c + 1];
  for (int64_t b = threadIdx.x; b < B; b += 256) dx[(int64_t)c * Bp + b] = inv * (g[b] - gm - yh[b] * gy);
}

// ---------------------------------------------------------------------------------------------
// decoder front: latent-expand MLP (three Linear + GELU, per-window [c][Bp] tensors), repeat over time + PE
// ---------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256) k_tfm_dec_expand_fwd(DofDecExp A) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  const int64_t Bp = A.Bp;
  float z[L], g1[L], g2[2 * L];
#pragma unroll
  for (int i = 0; i < L; ++i) z[i] = A.z[(int64_t)i * Bp + b];
#pragma unroll
  for (int o = 0; o < L; ++o) {
    float acc = A.b0[o];
#pragma unroll
    for (int i = 0; i < L; ++i) acc = fmaf(A.w0[o * L + i], z[i], acc);
    g1[o] = gelu_f(acc);
    if (A.keep) { A.a1[(int64_t)o * Bp + b] = acc; A.g1[(int64_t)o * Bp + b] = g1[o]; }
  }
#pragma unroll
  for (int o = 0; o < 2 * L; ++o) {
    float acc = A.b1[o];
#pragma unroll
    for (int i = 0; i < L; ++i) acc = fmaf(A.w1[o * L + i], g1[i], acc);
    g2[o] = gelu_f(acc);
    if (A.keep) { A.a2[(int64_t)o * Bp + b] = acc; A.g2[(int64_t)o * Bp + b] = g2[o]; }
  }
#pragma unroll
  for (int o = 0; o < 4 * L; ++o) {
    float acc = A.b2[o];
#pragma unroll
    for (int i = 0; i < 2 * L; ++i) acc = fmaf(A.w2[o * 2 * L + i], g2[i], acc);
    if (A.keep) A.a3[(int64_t)o * Bp + b] = acc;
    A.g3[(int64_t)o * Bp + b] = gelu_f(acc);
  }
}

// dg3 [4L][Bp] -> da3, da2, da1 (stored over A.a3 / a2 / a1's gradient twins) and dz [L][Bp]
template <int L>
__global__ void __launch_bounds__(256) k_tfm_dec_expand_bwd(DofDecExp A, const float* __restrict__ dg3,
                                                            float* __restrict__ da3, float* __restrict__ da2,
                                                            float* __restrict__ da1, float* __restrict__ dz) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  const int64_t Bp = A.Bp;
  float d3[4 * L], d2[2 * L], d1[L];
#pragma unroll
  for (int o = 0; o < 4 * L; ++o) {
    d3[o] = dg3[(int64_t)o * Bp + b] * dgelu_f(A.a3[(int64_t)o * Bp + b]);
    da3[(int64_t)o * Bp + b] = d3[o];
  }
#pragma unroll
  for (int i = 0; i < 2 * L; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int o = 0; o < 4 * L; ++o) acc = fmaf(A.w2[o * 2 * L + i], d3[o], acc);
    d2[i] = acc * dgelu_f(A.a2[(int64_t)i * Bp + b]);
    da2[(int64_t)i * Bp + b] = d2[i];
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int o = 0; o < 2 * L; ++o) acc = fmaf(A.w1[o * L + i], d2[o], acc);
    d1[i] = acc * dgelu_f(A.a1[(int64_t)i * Bp + b]);
    da1[(int64_t)i * Bp + b] = d1[i];
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int o = 0; o < L; ++o) acc = fmaf(A.w0[o * L + i], d1[o], acc);
    dz[(int64_t)i * Bp + b] = acc;
  }
}

// h0[t][b][c] = g3[c][b] + PE[t][c]
__global__ void __launch_bounds__(256) k_tfm_dec_h0(const float* __restrict__ g3, const float* __restrict__ pe,
                                                    float* __restrict__ h0, int T, int D, int64_t B, int64_t Bp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * B * D) return;
  const int c = (int)(i % D);
  const int64_t r = i / D;
  const int t = (int)(r / B);
  const int64_t b = r - (int64_t)t * B;
  h0[ACT(t, c, D, Bp, b)] = g3[(int64_t)c * Bp + b] + pe[t * D + c];
}
// dg3[c][b] = sum_t dh0[t][b][c]
__global__ void __launch_bounds__(256) k_tfm_dec_sum_time(const float* __restrict__ dh0, float* __restrict__ dg3, int T,
                                                          int D, int64_t B, int64_t Bp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int c = (int)(i % D);
  const int64_t b = i / D;
  float acc = 0.0f;
  for (int t = 0; t < T; ++t) acc += dh0[ACT(t, c, D, Bp, b)];
  dg3[(int64_t)c * Bp + b] = acc;
}

// Independent(Normal(loc, 1)) log-prob against the window, masked frames -> NaN; thread = (t, b)
__global__ void __launch_bounds__(256) k_tfm_dec_logp(const float* __restrict__ loc, int ld, const float* __restrict__ x,
                                                      const float* __restrict__ valid, float* __restrict__ loc_out,
                                                      float* __restrict__ recon_partial, float* __restrict__ dloc,
                                                      int T, int C3, int train, int64_t B, int64_t Bp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float nll[1] = {0.0f};
  if (i < (int64_t)T * B) {
    const int t = (int)(i / B);
    const int64_t b = i - (int64_t)t * B;
    const bool ok = valid[(int64_t)t * Bp + b] != 0.0f;
    const float inv_bt = 1.0f / ((float)B * (float)T);
    const float* __restrict__ xr = x + (b * T + t) * C3;
    const float* __restrict__ lr = loc + ((int64_t)t * Bp + b) * ld;
    float sq = 0.0f;
    for (int j = 0; j < C3; ++j) {
      float l = lr[j];
      if (l != l) l = 0.0f;
      l = fminf(fmaxf(l, -1e6f), 1e6f);
      if (loc_out) loc_out[(b * T + t) * C3 + j] = l;
      const float df = xr[j] - l;
      sq = fmaf(df, df, sq);
      if (train) dloc[((int64_t)t * Bp + b) * ld + j] = ok ? -df * inv_bt : NAN;
    }
    const float LOG_2PI = 1.8378770664093453f;
    nll[0] = ok ? 0.5f * sq + 0.5f * (float)C3 * LOG_2PI : NAN;
  }
  dof_block_colsum<1>(nll, recon_partial + blockIdx.x);
}

__global__ void k_tfm_tick(uint32_t* ctr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += 1u;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int dof_launch_tfm_tick(uint32_t* ctr, hipStream_t st) {
  DOF_LAUNCH(k_tfm_tick, (1), (64), st, ctr);
  return dof_check_launch("k_tfm_tick");
}

int dof_launch_tfm_embed(int F, const float* xin, const float* w, const float* bias, const float* pe, float* xs,
                         float* pad, float* y, const DofDrop& drop, int T, int G, int D, int64_t S, int64_t Sp,
                         hipStream_t st) {
  const unsigned nb = dof_cdiv((int64_t)T * S * (D / 4), 256);
  const float sq = (float)std::sqrt((double)D);
  if (F == 3) {
    DOF_LAUNCH((k_tfm_embed<3>), (nb), (256), st, xin, w, bias, pe, xs, pad, y, drop, sq, T, G, D, S, Sp);
  } else if (F == 1) {
    DOF_LAUNCH((k_tfm_embed<1>), (nb), (256), st, xin, w, bias, pe, xs, pad, y, drop, sq, T, G, D, S, Sp);
  } else {
    dof_set_error("features per group %d not supported (3 or 1)", F);
    return DOF_ERR_UNSUPPORTED;
  }
  return dof_check_launch("k_tfm_embed");
}

int dof_launch_tfm_embed_bwd(int F, const float* xs, const float* w, const float* bias, const float* dy, float* dpre,
                             const DofDrop& drop, int T, int D, int64_t S, int64_t Sp, hipStream_t st) {
  const unsigned nb = dof_cdiv((int64_t)T * S * (D / 4), 256);
  const float sq = (float)std::sqrt((double)D);
  if (F == 3) {
    DOF_LAUNCH((k_tfm_embed_bwd<3>), (nb), (256), st, xs, w, bias, dy, dpre, drop, sq, T, D, S, Sp);
  } else {
    DOF_LAUNCH((k_tfm_embed_bwd<1>), (nb), (256), st, xs, w, bias, dy, dpre, drop, sq, T, D, S, Sp);
  }
  return dof_check_launch("k_tfm_embed_bwd");
}

int dof_launch_tfm_gemm(const DofGemm& gin, hipStream_t st) {
  const int KC = (gin.K + 15) / 16, NT = (gin.N + 15) / 16;
  const int NTM = NT <= 2 ? 2 : NT <= 3 ? 3 : NT <= 4 ? 4 : NT <= 6 ? 6 : NT <= 8 ? 8 : 12;
  if (KC * NTM * 256 > kGemmLds || NT > 12 || (gin.ldx & 3)) {
    dof_set_error("tfm gemm: K %d x N %d (ldx %d) not supported", gin.K, gin.N, gin.ldx);
    return DOF_ERR_UNSUPPORTED;
  }
  const int64_t tiles = (int64_t)gin.T * (gin.Sp / 16);
  DofGemm g = gin;
  g.its = tiles >= 4 * 4 * 1024 ? 4 : tiles >= 2 * 4 * 1024 ? 2 : 1;  // >= ~1024 workgroups when the problem allows
  const unsigned nb = dof_cdiv(tiles, 4 * g.its);
#define GEMM_EPI(NTV)                                                                                          \
  switch (g.epi) {                                                                                             \
    case DOF_EPI_NONE: DOF_LAUNCH((k_tfm_gemm<NTV, DOF_EPI_NONE>), (nb), (256), st, g); break;                 \
    case DOF_EPI_RELU: DOF_LAUNCH((k_tfm_gemm<NTV, DOF_EPI_RELU>), (nb), (256), st, g); break;                 \
    case DOF_EPI_GELU: DOF_LAUNCH((k_tfm_gemm<NTV, DOF_EPI_GELU>), (nb), (256), st, g); break;                 \
    case DOF_EPI_MUL_RELU: DOF_LAUNCH((k_tfm_gemm<NTV, DOF_EPI_MUL_RELU>), (nb), (256), st, g); break;         \
    case DOF_EPI_MUL_DGELU: DOF_LAUNCH((k_tfm_gemm<NTV, DOF_EPI_MUL_DGELU>), (nb), (256), st, g); break;       \
    default: dof_set_error("tfm gemm: unknown epilogue %d", g.epi); return DOF_ERR_ARG;                        \
  }
  switch (NTM) {
    case 2: GEMM_EPI(2) break;
    case 3: GEMM_EPI(3) break;
    case 4: GEMM_EPI(4) break;
    case 6: GEMM_EPI(6) break;
    case 8: GEMM_EPI(8) break;
    default: GEMM_EPI(12) break;
  }
#undef GEMM_EPI
  return dof_check_launch("k_tfm_gemm");
}

// sequences per workgroup of the attention kernels (LDS and thread budget)
static int attn_nseq(const DofAttn& a, bool bwd, int budget = kAttLds) {
  const int per_seq = a.T * (bwd ? 4 : 3) * a.D + a.T + (bwd ? 3 * a.H * a.T : 0);
  int n = budget / per_seq;
  const int by_threads = 512 / (a.H * a.T);
  if (n > by_threads) n = by_threads;
  if (!bwd && n > 4) n = 4;
  return n;
}

// head sizes 1 .. 16: key_dim = min(64, 3 N) rounded down to a multiple of its 4 heads (models_new.py:1013-1019) is any
// multiple of 4 in 4 .. 64; the decoder's 8 heads of width 4 L / 8 (latent 4, 6, 8, 16: 2, 3, 4, 8)
// does a (window, width, heads) attention fit the kernels' LDS / thread budget, forward AND backward?  (plan creation
// asks, so that an unsupported shape fails there and not at the first backward pass)
// (resident kernels: windows <= 64 that fit their LDS; otherwise the long-window pair, whose only bound is its LDS table of
// three row statistics per head and row)
static bool attn_resident(const DofAttn& a, bool bwd) { return a.T <= 64 && attn_nseq(a, bwd) >= 1; }
bool dof_tfm_attn_fits(int T, int D, int H) {
  DofAttn a = {};
  a.T = T; a.D = D; a.H = H;
  if (D % H != 0 || D / H > 16) return false;
  return (attn_resident(a, false) && attn_resident(a, true)) || 3 * H * T <= kAttLongStats;
}

#define ATTN_DISPATCH(NAME, A, nb, nt, LDSF)                                                       \
  do {                                                                                        \
    const int dh = (A).D / (A).H;                                                             \
    if ((A).T <= 32) {                                                                        \
      if (dh == 1) DOF_LAUNCH((NAME<1, 32, LDSF>), (nb), (nt), st, A);                              \
      else if (dh == 2) DOF_LAUNCH((NAME<2, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 3) DOF_LAUNCH((NAME<3, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 4) DOF_LAUNCH((NAME<4, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 5) DOF_LAUNCH((NAME<5, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 6) DOF_LAUNCH((NAME<6, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 7) DOF_LAUNCH((NAME<7, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 8) DOF_LAUNCH((NAME<8, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 9) DOF_LAUNCH((NAME<9, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 10) DOF_LAUNCH((NAME<10, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 11) DOF_LAUNCH((NAME<11, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 12) DOF_LAUNCH((NAME<12, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 13) DOF_LAUNCH((NAME<13, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 14) DOF_LAUNCH((NAME<14, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 15) DOF_LAUNCH((NAME<15, 32, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 16) DOF_LAUNCH((NAME<16, 32, LDSF>), (nb), (nt), st, A);                        \
      else { dof_set_error("attention head size %d not supported", dh); return DOF_ERR_UNSUPPORTED; } \
    } else {                                                                                  \
      if (dh == 1) DOF_LAUNCH((NAME<1, 64, LDSF>), (nb), (nt), st, A);                              \
      else if (dh == 2) DOF_LAUNCH((NAME<2, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 3) DOF_LAUNCH((NAME<3, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 4) DOF_LAUNCH((NAME<4, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 5) DOF_LAUNCH((NAME<5, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 6) DOF_LAUNCH((NAME<6, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 7) DOF_LAUNCH((NAME<7, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 8) DOF_LAUNCH((NAME<8, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 9) DOF_LAUNCH((NAME<9, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 10) DOF_LAUNCH((NAME<10, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 11) DOF_LAUNCH((NAME<11, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 12) DOF_LAUNCH((NAME<12, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 13) DOF_LAUNCH((NAME<13, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 14) DOF_LAUNCH((NAME<14, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 15) DOF_LAUNCH((NAME<15, 64, LDSF>), (nb), (nt), st, A);                        \
      else if (dh == 16) DOF_LAUNCH((NAME<16, 64, LDSF>), (nb), (nt), st, A);                        \
      else { dof_set_error("attention head size %d not supported", dh); return DOF_ERR_UNSUPPORTED; } \
    }                                                                                         \
  } while (0)

#define ATTN_LONG_CASE(NAME, DHV) case DHV: DOF_LAUNCH((NAME<DHV>), ((unsigned)a.S), (256), st, a); break
#define ATTN_LONG_DISPATCH(NAME)                                                                                  \
  switch (a.D / a.H) {                                                                                            \
    ATTN_LONG_CASE(NAME, 1); ATTN_LONG_CASE(NAME, 2); ATTN_LONG_CASE(NAME, 3); ATTN_LONG_CASE(NAME, 4);           \
    ATTN_LONG_CASE(NAME, 5); ATTN_LONG_CASE(NAME, 6); ATTN_LONG_CASE(NAME, 7); ATTN_LONG_CASE(NAME, 8);           \
    ATTN_LONG_CASE(NAME, 9); ATTN_LONG_CASE(NAME, 10); ATTN_LONG_CASE(NAME, 11); ATTN_LONG_CASE(NAME, 12);        \
    ATTN_LONG_CASE(NAME, 13); ATTN_LONG_CASE(NAME, 14); ATTN_LONG_CASE(NAME, 15); ATTN_LONG_CASE(NAME, 16);       \
    default: dof_set_error("attention head size %d not supported", a.D / a.H); return DOF_ERR_UNSUPPORTED;        \
  }

int dof_launch_tfm_attn(DofAttn a, int backward, hipStream_t st) {
  if (a.D % a.H) {
    dof_set_error("attention: width %d not divisible by %d heads", a.D, a.H);
    return DOF_ERR_UNSUPPORTED;
  }
  a.scale = 1.0f / std::sqrt((float)(a.D / a.H));
  // one choice for the pair: the backward kernel recomputes what the forward kernel computed, in the same order
  if (!(attn_resident(a, false) && attn_resident(a, true))) {
    if (3 * a.H * a.T > kAttLongStats) {
      dof_set_error("attention: %d heads x window %d exceed the long-window kernel's statistics table (%d rows)", a.H, a.T,
                    kAttLongStats / 3);
      return DOF_ERR_UNSUPPORTED;
    }
    a.nseq = 1;
    if (backward) { ATTN_LONG_DISPATCH(k_tfm_attn_bwd_long) } else { ATTN_LONG_DISPATCH(k_tfm_attn_fwd_long) }
    return dof_check_launch("k_tfm_attn_long");
  }
  const int n_small = attn_nseq(a, backward != 0, backward ? kAttLdsBwdSmall : kAttLdsFwdSmall);
  a.nseq = n_small >= 1 ? n_small : attn_nseq(a, backward != 0);
  const unsigned nb = dof_cdiv(a.S, a.nseq);
  const unsigned nt = (unsigned)((a.nseq * a.H * a.T + 63) / 64 * 64);
  if (backward) {
    if (n_small >= 1) ATTN_DISPATCH(k_tfm_attn_bwd, a, nb, nt, kAttLdsBwdSmall);
    else ATTN_DISPATCH(k_tfm_attn_bwd, a, nb, nt, kAttLds);
  } else {
    if (n_small >= 1) ATTN_DISPATCH(k_tfm_attn_fwd, a, nb, nt, kAttLdsFwdSmall);
    else ATTN_DISPATCH(k_tfm_attn_fwd, a, nb, nt, kAttLds);
  }
  return dof_check_launch("k_tfm_attn");
}

int64_t dof_tfm_ln_blocks(int C, int T, int64_t Sp) {
  const int rpb = (C <= 32 ? 32 : 16) * 16;
  return dof_cdiv((int64_t)T * Sp, rpb);
}

#define LN_DISPATCH(NAME, C, nb, A)                                          \
  do {                                                                       \
    switch (C) {                                                             \
      case 4: DOF_LAUNCH((NAME<4>), (nb), (256), st, A); break;            \
      case 8: DOF_LAUNCH((NAME<8>), (nb), (256), st, A); break;            \
      case 12: DOF_LAUNCH((NAME<12>), (nb), (256), st, A); break;            \
      case 16: DOF_LAUNCH((NAME<16>), (nb), (256), st, A); break;            \
      case 20: DOF_LAUNCH((NAME<20>), (nb), (256), st, A); break;            \
      case 24: DOF_LAUNCH((NAME<24>), (nb), (256), st, A); break;            \
      case 28: DOF_LAUNCH((NAME<28>), (nb), (256), st, A); break;            \
      case 32: DOF_LAUNCH((NAME<32>), (nb), (256), st, A); break;            \
      case 36: DOF_LAUNCH((NAME<36>), (nb), (256), st, A); break;            \
      case 40: DOF_LAUNCH((NAME<40>), (nb), (256), st, A); break;            \
      case 44: DOF_LAUNCH((NAME<44>), (nb), (256), st, A); break;            \
      case 48: DOF_LAUNCH((NAME<48>), (nb), (256), st, A); break;            \
      case 52: DOF_LAUNCH((NAME<52>), (nb), (256), st, A); break;            \
      case 56: DOF_LAUNCH((NAME<56>), (nb), (256), st, A); break;            \
      case 60: DOF_LAUNCH((NAME<60>), (nb), (256), st, A); break;            \
      case 64: DOF_LAUNCH((NAME<64>), (nb), (256), st, A); break;            \
      default: dof_set_error("LayerNorm width %d not supported", C); return DOF_ERR_UNSUPPORTED; \
    }                                                                        \
  } while (0)

int dof_launch_tfm_add_ln(const DofLn& a, int C, hipStream_t st) {
  LN_DISPATCH(k_tfm_add_ln_fwd, C, ((unsigned)dof_tfm_ln_blocks(C, a.T, a.Sp)), a);
  return dof_check_launch("k_tfm_add_ln_fwd");
}
int dof_launch_tfm_ln_bwd(const DofLnBwd& a, int C, hipStream_t st) {
  LN_DISPATCH(k_tfm_ln_bwd, C, ((unsigned)dof_tfm_ln_blocks(C, a.T, a.Sp)), a);
  return dof_check_launch("k_tfm_ln_bwd");
}

int dof_launch_tfm_last(const float* x, float* n2, int T, int D, int64_t S, int64_t Sp, hipStream_t st) {
  DOF_LAUNCH(k_tfm_last_fwd, (dof_cdiv(S * D, 256)), (256), st, x, n2, T, D, S, Sp);
  return dof_check_launch("k_tfm_last_fwd");
}
int dof_launch_tfm_last_bwd(const float* dn2, float* dx, int T, int D, int64_t S, int64_t Sp, hipStream_t st) {
  DOF_LAUNCH(k_tfm_last_bwd, (dof_cdiv((int64_t)T * Sp * D, 256)), (256), st, dn2, dx, T, D, S, Sp);
  return dof_check_launch("k_tfm_last_bwd");
}

int dof_launch_tfm_bstd(const float* in, float* out, float* stat, int L, int standardize, int64_t B, int64_t Bp,
                        hipStream_t st) {
  DOF_LAUNCH(k_tfm_bstd_fwd, ((unsigned)L), (256), st, in, out, stat, standardize, B, Bp);
  return dof_check_launch("k_tfm_bstd_fwd");
}
int dof_launch_tfm_bstd_bwd(const float* dy, const float* y, const float* stat, float* dx, int L, int standardize,
                            int64_t B, int64_t Bp, hipStream_t st) {
  DOF_LAUNCH(k_tfm_bstd_bwd, ((unsigned)L), (256), st, dy, y, stat, dx, standardize, B, Bp);
  return dof_check_launch("k_tfm_bstd_bwd");
}

int dof_launch_tfm_dec_expand(int L, const DofDecExp& a, hipStream_t st) {
  const unsigned nb = dof_cdiv(a.B, 256);
  switch (L) {
    case 4: DOF_LAUNCH((k_tfm_dec_expand_fwd<4>), (nb), (256), st, a); break;
    case 5: DOF_LAUNCH((k_tfm_dec_expand_fwd<5>), (nb), (256), st, a); break;
    case 6: DOF_LAUNCH((k_tfm_dec_expand_fwd<6>), (nb), (256), st, a); break;
    case 8: DOF_LAUNCH((k_tfm_dec_expand_fwd<8>), (nb), (256), st, a); break;
    case 10: DOF_LAUNCH((k_tfm_dec_expand_fwd<10>), (nb), (256), st, a); break;
    case 12: DOF_LAUNCH((k_tfm_dec_expand_fwd<12>), (nb), (256), st, a); break;
    case 16: DOF_LAUNCH((k_tfm_dec_expand_fwd<16>), (nb), (256), st, a); break;
    default: dof_set_error("latent_dim %d not supported", L); return DOF_ERR_UNSUPPORTED;
  }
  return dof_check_launch("k_tfm_dec_expand_fwd");
}
int dof_launch_tfm_dec_expand_bwd(int L, const DofDecExp& a, const float* dg3, float* da3, float* da2, float* da1,
                                  float* dz, hipStream_t st) {
  const unsigned nb = dof_cdiv(a.B, 256);
  switch (L) {
    case 4: DOF_LAUNCH((k_tfm_dec_expand_bwd<4>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    case 5: DOF_LAUNCH((k_tfm_dec_expand_bwd<5>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    case 6: DOF_LAUNCH((k_tfm_dec_expand_bwd<6>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    case 8: DOF_LAUNCH((k_tfm_dec_expand_bwd<8>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    case 10: DOF_LAUNCH((k_tfm_dec_expand_bwd<10>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    case 12: DOF_LAUNCH((k_tfm_dec_expand_bwd<12>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    case 16: DOF_LAUNCH((k_tfm_dec_expand_bwd<16>), (nb), (256), st, a, dg3, da3, da2, da1, dz); break;
    default: dof_set_error("latent_dim %d not supported", L); return DOF_ERR_UNSUPPORTED;
  }
  return dof_check_launch("k_tfm_dec_expand_bwd");
}
int dof_launch_tfm_dec_h0(const float* g3, const float* pe, float* h0, int T, int D, int64_t B, int64_t Bp,
                          hipStream_t st) {
  DOF_LAUNCH(k_tfm_dec_h0, (dof_cdiv((int64_t)T * B * D, 256)), (256), st, g3, pe, h0, T, D, B, Bp);
  return dof_check_launch("k_tfm_dec_h0");
}
int dof_launch_tfm_dec_sum_time(const float* dh0, float* dg3, int T, int D, int64_t B, int64_t Bp, hipStream_t st) {
  DOF_LAUNCH(k_tfm_dec_sum_time, (dof_cdiv(B * D, 256)), (256), st, dh0, dg3, T, D, B, Bp);
  return dof_check_launch("k_tfm_dec_sum_time");
}
int dof_launch_tfm_dec_logp(const float* loc, int ld, const float* x, const float* valid, float* loc_out,
                            float* recon_partial, float* dloc, int T, int C3, int train, int64_t B, int64_t Bp,
                            hipStream_t st) {
  DOF_LAUNCH(k_tfm_dec_logp, (dof_cdiv((int64_t)T * B, 256)), (256), st, loc, ld, x, valid, loc_out, recon_partial, dloc,
             T, C3, train, B, Bp);
  return dof_check_launch("k_tfm_dec_logp");
}
