// Recurrent encoder/decoder kernels.
//
// Data layout: time-major activations are channel-minor X[t][s][c] (ACT() in dof_rt.h; Sp =
// sequence count padded to 64); per-window tensors are [c][s].  Hidden sizes are 8..32 -- far too
// small for MFMA tiles to pay in fp32 -- so the recurrences run on the VALU with the state in VGPRs.
//
// Reference semantics restated here:
//   * tf_style_group_reshape scramble  /root/reference/deepof/clustering/models_new.py:120-138
//   * RecurrentBlockPT                 models_new.py:217-278  (Conv1d k=5 'same' no-bias + ReLU;
//     length = number of non-zero conv rows, the FIRST `length` steps are processed (packed
//     semantics), outputs beyond are 0; LayerNorm eps=1e-3 over ALL steps; final hidden of GRU2)
//   * GRU cell = torch.nn.GRU (gates r,z,n; h' = (1-z)*n + z*h)
#include <cstdlib>
#include "dof_rt.h"
#include "launchers.h"
#include "k_sum_partials.inc.h"

namespace {


// ---------------------------------------------------------------------------------------------
// Encoder stage 1: scrambled read + Conv1d(F -> C1, k=5, same, no bias) + ReLU + mask/length.
// ---------------------------------------------------------------------------------------------
template <int C1, int F>
__global__ void __launch_bounds__(256) k_enc_conv_fwd(const float* __restrict__ xin,  // (B,T,G,F) reference layout
                                                      const float* __restrict__ w,    // (C1,F,5)
                                                      float* __restrict__ xs,         // [T][Sp][F] scrambled copy
                                                      float* __restrict__ c,          // [T][Sp][C1]
                                                      int* __restrict__ len, int T, int G, int64_t S, int64_t Sp) {
  // thread = (output time, sequence): T x more parallelism than one thread per sequence; the
  // per-sequence length (number of non-zero conv rows) is an integer atomic count (order-free).
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * S) return;
  const int to = (int)(i / S);
  const int64_t s = i - (int64_t)to * S;
  const int64_t b = s / G;
  const int g = (int)(s - b * G);
  const float* __restrict__ win = xin + b * (int64_t)T * G * F;
  const dof_cfp wc = dof_cw(w);
  float rows[5][F];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int tt = to + k - 2;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float v = 0.0f;
      if (tt >= 0 && tt < T) {
        // y[b,g,tt,f] = x[b, t, cc] with cc*T + t = (f*T + tt)*G + g   (models_new.py:131-137)
        const int lin = (f * T + tt) * G + g;
        const int cc = lin / T;
        const int t = lin - cc * T;
        v = win[(int64_t)t * G * F + cc];
      }
      rows[k][f] = v;
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) xs[ACT(to, f, F, Sp, s)] = rows[2][f];
  bool nz = false;
  float crow[C1];
#pragma unroll
  for (int o = 0; o < C1; ++o) {
    float acc = 0.0f;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int k = 0; k < 5; ++k) acc = fmaf(wc[(o * F + f) * 5 + k], rows[k][f], acc);
    acc = acc > 0.0f ? acc : 0.0f;
    nz |= (acc != 0.0f);
    crow[o] = acc;
  }
  dof_st_row<C1>(c + ACT(to, 0, C1, Sp, s), crow);
  if (nz) atomicAdd(len + s, 1);
}

// The same stage with the window staged in LDS: one workgroup per window.  The scramble
// y[b,g,tt,f] = x[b,t,cc] with cc*T + t = (f*T + tt)*G + g is a plain transpose of the window's (T x G*F) matrix --
// y_flat[cc*T + t] = x[t][cc] -- so the window is read once with coalesced loads and written to LDS transposed; the
// conv taps then read consecutive LDS words.  No integer division per tap, no scattered global reads (the direct
// kernel above spent 1.17 ms on C5's node stream against a 0.11 ms HBM floor), and the per-sequence lengths are
// counted in LDS and stored once (no global atomics).
constexpr int ENC_CONV_LDS = 12288;  // floats: largest window (T*G*F) staged; larger ones take the direct kernel
template <int C1, int F>
__device__ __forceinline__ void enc_conv_fwd_lds_body(const float* __restrict__ xin, const float* __restrict__ w,
                                                      float* __restrict__ xs, float* __restrict__ c, int* __restrict__ len,
                                                      int T, int G, int64_t S, int64_t Sp, float* sy, int* cnt) {
  const int64_t b = blockIdx.x;
  const int GF = G * F, n = T * GF;
  const float* __restrict__ win = xin + b * (int64_t)n;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    const int t = idx / GF, cc = idx - t * GF;
    sy[cc * T + t] = win[idx];
  }
  if ((int)threadIdx.x < G) cnt[threadIdx.x] = 0;
  __syncthreads();
  const dof_cfp wc = dof_cw(w);
  for (int item = threadIdx.x; item < T * G; item += 256) {  // consecutive items = consecutive sequences of one time step
    const int to = item / G, g = item - to * G;
    const int64_t s = b * G + g;
    float rows[5][F];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int tt = to + k - 2;
#pragma unroll
      for (int f = 0; f < F; ++f) rows[k][f] = (tt >= 0 && tt < T) ? sy[(f * T + tt) * G + g] : 0.0f;
    }
#pragma unroll
    for (int f = 0; f < F; ++f) xs[ACT(to, f, F, Sp, s)] = rows[2][f];
    bool nz = false;
    float crow[C1];
#pragma unroll
    for (int o = 0; o < C1; ++o) {
      float acc = 0.0f;
#pragma unroll
      for (int f = 0; f < F; ++f)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc = fmaf(wc[(o * F + f) * 5 + k], rows[k][f], acc);
      acc = acc > 0.0f ? acc : 0.0f;
      nz |= (acc != 0.0f);
      crow[o] = acc;
    }
    dof_st_row<C1>(c + ACT(to, 0, C1, Sp, s), crow);
    if (nz) atomicAdd(&cnt[g], 1);  // integer LDS atomic: order-free
  }
  __syncthreads();
  if ((int)threadIdx.x < G) len[b * G + threadIdx.x] = cnt[threadIdx.x];
}
template <int C1, int F>
__global__ void __launch_bounds__(256) k_enc_conv_fwd_lds(const float* __restrict__ xin, const float* __restrict__ w,
                                                          float* __restrict__ xs, float* __restrict__ c,
                                                          int* __restrict__ len, int T, int G, int64_t S, int64_t Sp) {
  __shared__ float sy[ENC_CONV_LDS];
  __shared__ int cnt[256];
  enc_conv_fwd_lds_body<C1, F>(xin, w, xs, c, len, T, G, S, Sp, sy, cnt);
}
// both encoder streams (node: 3 features per group, edge: 1) in one launch, blockIdx.y = stream: the two launches were
// 21 + 15 us of one-pass-and-a-tail workgroups at C2; side by side the tails overlap
struct EncConvFwdArgs {
  const float* xin; const float* w; float* xs; float* c; int* len;
  int G; int64_t S, Sp;
};
// LDSF: floats of the staged window (classes 2048 / 6144 / ENC_CONV_LDS): the kernel is latency-bound and a C2 window is 1,050
// floats -- with the 48 KB array it ran three workgroups per CU
template <int C1, int LDSF>
__global__ void __launch_bounds__(256) k_enc_conv_fwd_lds_pair(EncConvFwdArgs A0, EncConvFwdArgs A1, int T) {
  __shared__ float sy[LDSF];
  __shared__ int cnt[256];
  if (blockIdx.y == 0) {
    if ((int64_t)blockIdx.x * A0.G < A0.S) enc_conv_fwd_lds_body<C1, 3>(A0.xin, A0.w, A0.xs, A0.c, A0.len, T, A0.G, A0.S, A0.Sp, sy, cnt);
  } else {
    if ((int64_t)blockIdx.x * A1.G < A1.S) enc_conv_fwd_lds_body<C1, 1>(A1.xin, A1.w, A1.xs, A1.c, A1.len, T, A1.G, A1.S, A1.Sp, sy, cnt);
  }
}

// ---------------------------------------------------------------------------------------------
// GRU forward.  Thread = (sequence, direction).  Saves gates (r, z, n, W_hn h + b_hn) for bwd.
// ---------------------------------------------------------------------------------------------
template <int IN, int HID, bool BCAST>
__global__ void __launch_bounds__(256) k_gru_fwd(const float* __restrict__ X,  // [T][IN][Sp] (or [IN][Sp] if BCAST)
                                                 const int* __restrict__ len,
                                                 // weights as separate noalias kernel arguments: lets the
                                                 // compiler read them through the scalar cache (s_load)
                                                 const float* __restrict__ wih0, const float* __restrict__ whh0,
                                                 const float* __restrict__ bih0, const float* __restrict__ bhh0,
                                                 const float* __restrict__ wih1, const float* __restrict__ whh1,
                                                 const float* __restrict__ bih1, const float* __restrict__ bhh1,
                                                 float* __restrict__ O,    // [T][2*HID][Sp]
                                                 float* __restrict__ GS,   // [2][T][4*HID][Sp] or null (inference)
                                                 int T, int64_t S, int64_t Sp) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int dir = blockIdx.y;
  // Stage this direction's weights in LDS once per workgroup.  Every lane then reads the same
  // address (LDS broadcast, conflict-free ds_read_b128 = 4 weights per instruction), many reads
  // in flight -- measured 3-5x faster here than scalar-cache loads, whose ~300-cycle round trip
  // is fully exposed at <= 1 wave per SIMD (batch 1024 gives only ~450 waves per launch).
  __shared__ __attribute__((aligned(16))) float wih[3 * HID * IN];
  __shared__ __attribute__((aligned(16))) float whh[3 * HID * HID];
  __shared__ __attribute__((aligned(16))) float bih[3 * HID];
  __shared__ __attribute__((aligned(16))) float bhh[3 * HID];
  {
    const float* __restrict__ g_wih = dir ? wih1 : wih0;
    const float* __restrict__ g_whh = dir ? whh1 : whh0;
    const float* __restrict__ g_bih = dir ? bih1 : bih0;
    const float* __restrict__ g_bhh = dir ? bhh1 : bhh0;
    for (int i = threadIdx.x; i < 3 * HID * IN; i += blockDim.x) wih[i] = g_wih[i];
    for (int i = threadIdx.x; i < 3 * HID * HID; i += blockDim.x) whh[i] = g_whh[i];
    for (int i = threadIdx.x; i < 3 * HID; i += blockDim.x) {
      bih[i] = g_bih[i];
      bhh[i] = g_bhh[i];
    }
  }
  __syncthreads();
  if (s >= S) return;
  float* __restrict__ gs = GS ? GS + (int64_t)dir * T * 4 * HID * Sp : nullptr;
  const int n = len[s];
  float h[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) h[j] = 0.0f;
  float x[IN];
  float gi[3 * HID];
  if (BCAST) {
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = X[(int64_t)k * Sp + s];
#pragma unroll
    for (int j = 0; j < 3 * HID; ++j) {
      float acc = bih[j];
#pragma unroll
      for (int k = 0; k < IN; ++k) acc = fmaf(wih[j * IN + k], x[k], acc);
      gi[j] = acc;
    }
  }
  for (int step = 0; step < n; ++step) {
    const int t = dir ? (n - 1 - step) : step;
    if (!BCAST) {
#pragma unroll
      for (int k = 0; k < IN; ++k) x[k] = X[ACT(t, k, IN, Sp, s)];
    }
    float hn[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      float ar, az, an, ahn = bhh[2 * HID + j];
      if (BCAST) {
        ar = gi[j] + bhh[j];
        az = gi[HID + j] + bhh[HID + j];
        an = gi[2 * HID + j];
      } else {
        ar = bih[j] + bhh[j];
        az = bih[HID + j] + bhh[HID + j];
        an = bih[2 * HID + j];
#pragma unroll
        for (int k = 0; k < IN; ++k) {
          ar = fmaf(wih[j * IN + k], x[k], ar);
          az = fmaf(wih[(HID + j) * IN + k], x[k], az);
          an = fmaf(wih[(2 * HID + j) * IN + k], x[k], an);
        }
      }
#pragma unroll
      for (int k = 0; k < HID; ++k) {
        ar = fmaf(whh[j * HID + k], h[k], ar);
        az = fmaf(whh[(HID + j) * HID + k], h[k], az);
        ahn = fmaf(whh[(2 * HID + j) * HID + k], h[k], ahn);
      }
      const float r = dof_sigmoid(ar);
      const float z = dof_sigmoid(az);
      const float nn = dof_tanh(fmaf(r, ahn, an));
      hn[j] = fmaf(z, h[j] - nn, nn);
      if (gs) {
        gs[ACT(t, j, 4 * HID, Sp, s)] = r;
        gs[ACT(t, HID + j, 4 * HID, Sp, s)] = z;
        gs[ACT(t, 2 * HID + j, 4 * HID, Sp, s)] = nn;
        gs[ACT(t, 3 * HID + j, 4 * HID, Sp, s)] = ahn;
      }
    }
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      h[j] = hn[j];
      O[ACT(t, dir * HID + j, 2 * HID, Sp, s)] = hn[j];
    }
  }
  for (int t = n; t < T; ++t) {
#pragma unroll
    for (int j = 0; j < HID; ++j) O[ACT(t, dir * HID + j, 2 * HID, Sp, s)] = 0.0f;
    if (gs) {
#pragma unroll
      for (int j = 0; j < 4 * HID; ++j) gs[ACT(t, j, 4 * HID, Sp, s)] = 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GRU backward.  Consumes the saved gates, overwrites them in place with the pre-activation
// gate gradients dG = [d r_pre, d z_pre, d n_pre, d(W_hn h + b_hn)] (input of the weight-grad
// reduction), writes dX per direction.
// ---------------------------------------------------------------------------------------------
template <int IN, int HID, bool BCAST>
__global__ void __launch_bounds__(256) k_gru_bwd(const int* __restrict__ len,
                                                 const float* __restrict__ wih0, const float* __restrict__ whh0,
                                                 const float* __restrict__ wih1, const float* __restrict__ whh1,
                                                 const float* __restrict__ O,      // [T][2*HID][Sp] fwd outputs
                                                 float* __restrict__ GS,           // [2][T][4*HID][Sp] gates -> dG
                                                 const float* __restrict__ dO,     // [T][2*HID][Sp] or null
                                                 const float* __restrict__ dHfin,  // [2*HID][Sp] or null
                                                 float* __restrict__ dX,  // [2][T][IN][Sp]  (BCAST: [2][IN][Sp])
                                                 int T, int64_t S, int64_t Sp) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int dir = blockIdx.y;
  __shared__ __attribute__((aligned(16))) float wih[3 * HID * IN];   // LDS-staged weights, see k_gru_fwd
  __shared__ __attribute__((aligned(16))) float whh[3 * HID * HID];
  {
    const float* __restrict__ g_wih = dir ? wih1 : wih0;
    const float* __restrict__ g_whh = dir ? whh1 : whh0;
    for (int i = threadIdx.x; i < 3 * HID * IN; i += blockDim.x) wih[i] = g_wih[i];
    for (int i = threadIdx.x; i < 3 * HID * HID; i += blockDim.x) whh[i] = g_whh[i];
  }
  __syncthreads();
  if (s >= S) return;
  float* __restrict__ gs = GS + (int64_t)dir * T * 4 * HID * Sp;
  float* __restrict__ dx_out = dX + (int64_t)dir * (BCAST ? 1 : T) * IN * Sp;
  const int n = len[s];
  float dh[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) dh[j] = (dHfin && n > 0) ? dHfin[(int64_t)(dir * HID + j) * Sp + s] : 0.0f;
  float dxacc[IN];
  if (BCAST) {
#pragma unroll
    for (int k = 0; k < IN; ++k) dxacc[k] = 0.0f;
  }
  for (int step = n - 1; step >= 0; --step) {
    const int t = dir ? (n - 1 - step) : step;
    const int tp = dir ? t + 1 : t - 1;  // time index holding h_{prev}
    float dg[4 * HID];
    float dhn[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      const float r = gs[ACT(t, j, 4 * HID, Sp, s)];
      const float z = gs[ACT(t, HID + j, 4 * HID, Sp, s)];
      const float nn = gs[ACT(t, 2 * HID + j, 4 * HID, Sp, s)];
      const float ahn = gs[ACT(t, 3 * HID + j, 4 * HID, Sp, s)];
      const float hp = (step > 0) ? O[ACT(tp, dir * HID + j, 2 * HID, Sp, s)] : 0.0f;
      float dht = dh[j];
      if (dO) dht += dO[ACT(t, dir * HID + j, 2 * HID, Sp, s)];
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp - nn);
      dhn[j] = dht * z;
      const float dnp = dn * (1.0f - nn * nn);
      dg[j] = dnp * ahn * r * (1.0f - r);
      dg[HID + j] = dz * z * (1.0f - z);
      dg[2 * HID + j] = dnp;
      dg[3 * HID + j] = dnp * r;
    }
#pragma unroll
    for (int j = 0; j < 4 * HID; ++j) gs[ACT(t, j, 4 * HID, Sp, s)] = dg[j];
    // dh_prev += W_hh^T [dr, dz, d(ahn)]
#pragma unroll
    for (int j = 0; j < HID; ++j) {
#pragma unroll
      for (int k = 0; k < HID; ++k) {
        dhn[k] = fmaf(whh[j * HID + k], dg[j], dhn[k]);
        dhn[k] = fmaf(whh[(HID + j) * HID + k], dg[HID + j], dhn[k]);
        dhn[k] = fmaf(whh[(2 * HID + j) * HID + k], dg[3 * HID + j], dhn[k]);
      }
    }
#pragma unroll
    for (int j = 0; j < HID; ++j) dh[j] = dhn[j];
    // dx = W_ih^T [dr, dz, dn]
    float dx[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) dx[k] = 0.0f;
#pragma unroll
    for (int j = 0; j < 3 * HID; ++j) {
#pragma unroll
      for (int k = 0; k < IN; ++k) dx[k] = fmaf(wih[j * IN + k], dg[j], dx[k]);
    }
    if (BCAST) {
#pragma unroll
      for (int k = 0; k < IN; ++k) dxacc[k] += dx[k];
    } else {
#pragma unroll
      for (int k = 0; k < IN; ++k) dx_out[ACT(t, k, IN, Sp, s)] = dx[k];
    }
  }
  if (BCAST) {
#pragma unroll
    for (int k = 0; k < IN; ++k) dx_out[(int64_t)k * Sp + s] = dxacc[k];
  } else {
    for (int t = n; t < T; ++t)
#pragma unroll
      for (int k = 0; k < IN; ++k) dx_out[ACT(t, k, IN, Sp, s)] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// Generic GRU with a sequence spread over FOUR adjacent lanes (latent 16: hidden sizes 32 and 16).  The one-thread-per-
// sequence kernels above run 25,600 sequences of a C2 batch as 800 wavefronts -- less than one per SIMD -- each a serial
// chain of 3 HID (IN + HID) LDS-fed FMAs per step (6,144 at HID = IN = 32: 735 us forward, 1,141 us backward).  Here
// lane q of a quad owns units [q HID/4, (q + 1) HID/4) of its sequence: four times the wavefronts, a quarter of the chain,
// and the quad all-gathers h (forward) or the gate gradients (backward) once per step.  Every sum runs over the same
// index order as in k_gru_fwd / k_gru_bwd, so the results are bitwise the same; same buffers, gate-major rows.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void dof_quad_allgather(const float (&mine)[N], float (&full)[4 * N]) {
  const int base = (int)(threadIdx.x & 63) & ~3;
#pragma unroll
  for (int src = 0; src < 4; ++src)
#pragma unroll
    for (int u = 0; u < N; ++u) full[src * N + u] = __shfl(mine[u], base + src);
}

template <int IN, int HID, bool BCAST>
__global__ void __launch_bounds__(256) k_gruq_fwd(const float* __restrict__ X, const int* __restrict__ len,
                                                  const float* __restrict__ wih0, const float* __restrict__ whh0,
                                                  const float* __restrict__ bih0, const float* __restrict__ bhh0,
                                                  const float* __restrict__ wih1, const float* __restrict__ whh1,
                                                  const float* __restrict__ bih1, const float* __restrict__ bhh1,
                                                  float* __restrict__ O, float* __restrict__ GS, int T, int64_t S,
                                                  int64_t Sp) {
  static_assert(HID % 4 == 0 && IN % 4 == 0, "quad split");
  constexpr int UQ = HID / 4;
  const int q = threadIdx.x & 3;
  const int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int dir = blockIdx.y;
  __shared__ __attribute__((aligned(16))) float wih[3 * HID * IN];
  __shared__ __attribute__((aligned(16))) float whh[3 * HID * HID];
  __shared__ __attribute__((aligned(16))) float bih[3 * HID];
  __shared__ __attribute__((aligned(16))) float bhh[3 * HID];
  {
    const float* __restrict__ g_wih = dir ? wih1 : wih0;
    const float* __restrict__ g_whh = dir ? whh1 : whh0;
    const float* __restrict__ g_bih = dir ? bih1 : bih0;
    const float* __restrict__ g_bhh = dir ? bhh1 : bhh0;
    for (int i = threadIdx.x; i < 3 * HID * IN; i += blockDim.x) wih[i] = g_wih[i];
    for (int i = threadIdx.x; i < 3 * HID * HID; i += blockDim.x) whh[i] = g_whh[i];
    for (int i = threadIdx.x; i < 3 * HID; i += blockDim.x) {
      bih[i] = g_bih[i];
      bhh[i] = g_bhh[i];
    }
  }
  __syncthreads();
  // (quads of sequences past the end keep running on the last sequence without storing: the all-gathers are wave-wide)
  const bool live = s < S;
  const int64_t sr = live ? s : S - 1;
  float* __restrict__ gs = GS ? GS + (int64_t)dir * T * 4 * HID * Sp : nullptr;
  const int n = len[sr];
  int nmax = n;  // the longest sequence of the wavefront bounds the loop
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(nmax, m);
    nmax = o > nmax ? o : nmax;
  }
  float h[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) h[j] = 0.0f;
  float x[IN];
  float gi[3 * UQ];
  if (BCAST) {
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = X[(int64_t)k * Sp + sr];
#pragma unroll
    for (int u = 0; u < UQ; ++u)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int j = g * HID + q * UQ + u;
        float acc = bih[j];
#pragma unroll
        for (int k = 0; k < IN; ++k) acc = fmaf(wih[j * IN + k], x[k], acc);
        gi[g * UQ + u] = acc;
      }
  }
  for (int step = 0; step < nmax; ++step) {
    const bool act = step < n;
    const int t = dir ? (n - 1 - step) : step;
    const int tl = act ? t : 0;
    if (!BCAST) dof_ld_row<IN>(X + ACT(tl, 0, IN, Sp, sr), x);
    float hmine[UQ];
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      DOF_MEM_FENCE();  // one unit's weight reads at a time (hoisted across units they take every register)
      const int j = q * UQ + u;
      float ar, az, an, ahn = bhh[2 * HID + j];
      if (BCAST) {
        ar = gi[u] + bhh[j];
        az = gi[UQ + u] + bhh[HID + j];
        an = gi[2 * UQ + u];
      } else {
        ar = bih[j] + bhh[j];
        az = bih[HID + j] + bhh[HID + j];
        an = bih[2 * HID + j];
#pragma unroll
        for (int k = 0; k < IN; ++k) {
          ar = fmaf(wih[j * IN + k], x[k], ar);
          az = fmaf(wih[(HID + j) * IN + k], x[k], az);
          an = fmaf(wih[(2 * HID + j) * IN + k], x[k], an);
        }
      }
#pragma unroll
      for (int k = 0; k < HID; ++k) {
        ar = fmaf(whh[j * HID + k], h[k], ar);
        az = fmaf(whh[(HID + j) * HID + k], h[k], az);
        ahn = fmaf(whh[(2 * HID + j) * HID + k], h[k], ahn);
      }
      const float r = dof_sigmoid(ar);
      const float z = dof_sigmoid(az);
      const float nn = dof_tanh(fmaf(r, ahn, an));
      hmine[u] = fmaf(z, h[j] - nn, nn);
      if (act && live) {
        O[ACT(t, dir * HID + j, 2 * HID, Sp, s)] = hmine[u];
        if (gs) {
          gs[ACT(t, j, 4 * HID, Sp, s)] = r;
          gs[ACT(t, HID + j, 4 * HID, Sp, s)] = z;
          gs[ACT(t, 2 * HID + j, 4 * HID, Sp, s)] = nn;
          gs[ACT(t, 3 * HID + j, 4 * HID, Sp, s)] = ahn;
        }
      }
    }
    dof_quad_allgather<UQ>(hmine, h);
  }
  if (!live) return;
  for (int t = n; t < T; ++t) {
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const int j = q * UQ + u;
      O[ACT(t, dir * HID + j, 2 * HID, Sp, s)] = 0.0f;
      if (gs) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gs[ACT(t, g * HID + j, 4 * HID, Sp, s)] = 0.0f;
      }
    }
  }
}

template <int IN, int HID, bool BCAST>
__global__ void __launch_bounds__(256) k_gruq_bwd(const int* __restrict__ len, const float* __restrict__ wih0,
                                                  const float* __restrict__ whh0, const float* __restrict__ wih1,
                                                  const float* __restrict__ whh1, const float* __restrict__ O,
                                                  float* __restrict__ GS, const float* __restrict__ dO,
                                                  const float* __restrict__ dHfin, float* __restrict__ dX, int T,
                                                  int64_t S, int64_t Sp) {
  static_assert(HID % 4 == 0 && IN % 4 == 0, "quad split");
  constexpr int UQ = HID / 4, XQ = IN / 4;
  const int q = threadIdx.x & 3;
  const int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const int dir = blockIdx.y;
  __shared__ __attribute__((aligned(16))) float wih[3 * HID * IN];
  __shared__ __attribute__((aligned(16))) float whh[3 * HID * HID];
  {
    const float* __restrict__ g_wih = dir ? wih1 : wih0;
    const float* __restrict__ g_whh = dir ? whh1 : whh0;
    for (int i = threadIdx.x; i < 3 * HID * IN; i += blockDim.x) wih[i] = g_wih[i];
    for (int i = threadIdx.x; i < 3 * HID * HID; i += blockDim.x) whh[i] = g_whh[i];
  }
  __syncthreads();
  const bool live = s < S;
  const int64_t sr = live ? s : S - 1;
  float* __restrict__ gs = GS + (int64_t)dir * T * 4 * HID * Sp;
  float* __restrict__ dx_out = dX + (int64_t)dir * (BCAST ? 1 : T) * IN * Sp;
  const int n = len[sr];
  int nmax = n;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(nmax, m);
    nmax = o > nmax ? o : nmax;
  }
  float dh[UQ];  // gradient w.r.t. the lane's own units of h
#pragma unroll
  for (int u = 0; u < UQ; ++u) dh[u] = (dHfin && n > 0) ? dHfin[(int64_t)(dir * HID + q * UQ + u) * Sp + sr] : 0.0f;
  float dxacc[XQ];
#pragma unroll
  for (int k = 0; k < XQ; ++k) dxacc[k] = 0.0f;
  for (int step = nmax - 1; step >= 0; --step) {
    const bool act = step < n;
    const int t = dir ? (n - 1 - step) : step;
    const int tp = dir ? t + 1 : t - 1;
    float dgm[4 * UQ];  // own units: [gate][u]
    float dhn[UQ];
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const int j = q * UQ + u;
      float r = 0.0f, z = 0.0f, nn = 0.0f, ahn = 0.0f, hp = 0.0f, dht = 0.0f;
      if (act) {
        r = gs[ACT(t, j, 4 * HID, Sp, sr)];
        z = gs[ACT(t, HID + j, 4 * HID, Sp, sr)];
        nn = gs[ACT(t, 2 * HID + j, 4 * HID, Sp, sr)];
        ahn = gs[ACT(t, 3 * HID + j, 4 * HID, Sp, sr)];
        hp = (step > 0) ? O[ACT(tp, dir * HID + j, 2 * HID, Sp, sr)] : 0.0f;
        dht = dh[u];
        if (dO) dht += dO[ACT(t, dir * HID + j, 2 * HID, Sp, sr)];
      }
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp - nn);
      dhn[u] = dht * z;
      const float dnp = dn * (1.0f - nn * nn);
      dgm[u] = dnp * ahn * r * (1.0f - r);
      dgm[UQ + u] = dz * z * (1.0f - z);
      dgm[2 * UQ + u] = dnp;
      dgm[3 * UQ + u] = dnp * r;
      if (act && live) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gs[ACT(t, g * HID + j, 4 * HID, Sp, s)] = dgm[g * UQ + u];
      }
    }
    // the whole gate-gradient vector in every lane of the quad: dg[gate][unit], unit = src * UQ + u
    float dgf[16 * UQ];
    dof_quad_allgather<4 * UQ>(dgm, dgf);
    auto dg = [&](int g, int j) -> float { return dgf[(j / UQ) * 4 * UQ + g * UQ + (j % UQ)]; };
    // dh_prev of the lane's own units: W_hh^T [dr, dz, d(ahn)], unit order as in k_gru_bwd
#pragma unroll
    for (int j = 0; j < HID; ++j) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        const int k = q * UQ + u;
        dhn[u] = fmaf(whh[j * HID + k], dg(0, j), dhn[u]);
        dhn[u] = fmaf(whh[(HID + j) * HID + k], dg(1, j), dhn[u]);
        dhn[u] = fmaf(whh[(2 * HID + j) * HID + k], dg(3, j), dhn[u]);
      }
    }
    if (act) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) dh[u] = dhn[u];
    }
    // dx of the lane's own input channels: W_ih^T [dr, dz, dn]
    float dx[XQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) dx[k] = 0.0f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int j = 0; j < HID; ++j) {
#pragma unroll
        for (int k = 0; k < XQ; ++k) dx[k] = fmaf(wih[(g * HID + j) * IN + q * XQ + k], dg(g, j), dx[k]);
      }
    if (act && live) {
      if (BCAST) {
#pragma unroll
        for (int k = 0; k < XQ; ++k) dxacc[k] += dx[k];
      } else {
#pragma unroll
        for (int k = 0; k < XQ; ++k) dx_out[ACT(t, q * XQ + k, IN, Sp, s)] = dx[k];
      }
    }
  }
  if (!live) return;
  if (BCAST) {
#pragma unroll
    for (int k = 0; k < XQ; ++k) dx_out[(int64_t)(q * XQ + k) * Sp + s] = dxacc[k];
  } else {
    for (int t = n; t < T; ++t)
#pragma unroll
      for (int k = 0; k < XQ; ++k) dx_out[ACT(t, q * XQ + k, IN, Sp, s)] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// GEMM-shaped GRU for the wider layers (latent 16: (32,32), (64 -> 16); latent 32: (64,64), (128 -> 32)).  The forward
// recurrence is k_grumx_fwd (k_grumx.inc.h: bf16 matrix pipe, three-piece operands; round 4's fp32-MFMA k_grum_fwd took
// 650 us per encoder stream at (64, 64), 351 us now).  Gates are saved gate-major ([t][s][4 HID], the generic layout of
// k_gru_fwd / k_gruq_fwd), so the generic weight-gradient jobs apply; k_grum_bwd below is the matching backward, still on
// v_mfma_f32_16x16x4_f32.  What bounds these layers now is the traffic of the saved gates and gate gradients (734 MB per
// stream and pass at latent 32), not arithmetic: the latent-8 treatment -- recompute in the backward kernel, weight
// gradients fused -- is what is left to do here.
// ---------------------------------------------------------------------------------------------
// Backward of that layer from the saved gates: per step the gate gradients of the lane's own units (VALU), then
//   dh_{t-1} = dht * z + W_hh^T [g_r, g_z, g_h],   dx_t = W_ih^T [g_r, g_z, g_n]
// on the matrix pipe with the TRANSPOSED weights staged in operand order (row tile = 16 output indices, K block (gate, m, r) =
// the gate gradient of unit 16 m + 4 k + r: again the lane's own value), 3 (HID / 4) (HID / 16 + IN / 16) MFMAs per step.  The
// gate gradients replace the gates in GS (gate-major) for the generic weight-gradient jobs, dX leaves as 16-byte stores.
template <int IN, int HID>
__global__ void __launch_bounds__(256) k_grum_bwd(const int* __restrict__ len, const float* __restrict__ wih0,
                                                  const float* __restrict__ whh0, const float* __restrict__ wih1,
                                                  const float* __restrict__ whh1, const float* __restrict__ O,
                                                  float* __restrict__ GS, const float* __restrict__ dO,
                                                  const float* __restrict__ dHfin, float* __restrict__ dX, int T, int64_t S,
                                                  int64_t Sp) {
  static_assert(HID % 16 == 0 && IN % 16 == 0, "row tiles");
  constexpr int MT = HID / 16, XT = IN / 16, KH = HID / 4, KG = 3 * KH;
  __shared__ float th[MT * KG * 64];   // [output tile][K block (gate, unit slot)][lane]: W_hh^T, gates r, z, hn
  __shared__ float tx[XT * KG * 64];   // W_ih^T, gates r, z, n
  const int dir = blockIdx.y;
  {
    const float* __restrict__ g_wih = dir ? wih1 : wih0;
    const float* __restrict__ g_whh = dir ? whh1 : whh0;
    for (int e = threadIdx.x; e < MT * KG * 64; e += 256) {
      const int ln = e & 63, kb = (e >> 6) % KG, mo = (e >> 6) / KG;
      const int g = kb / KH, q = kb % KH, unit = 16 * (q >> 2) + 4 * (ln >> 4) + (q & 3);
      th[e] = g_whh[(g * HID + unit) * HID + 16 * mo + (ln & 15)];
    }
    for (int e = threadIdx.x; e < XT * KG * 64; e += 256) {
      const int ln = e & 63, kb = (e >> 6) % KG, mo = (e >> 6) / KG;
      const int g = kb / KH, q = kb % KH, unit = 16 * (q >> 2) + 4 * (ln >> 4) + (q & 3);
      tx[e] = g_wih[(g * HID + unit) * IN + 16 * mo + (ln & 15)];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, b = lane >> 4;
  const int64_t s = ((int64_t)blockIdx.x * 4 + wave) * 16 + j;
  if (((int64_t)blockIdx.x * 4 + wave) * 16 >= S) return;
  const bool live = s < S;
  const int64_t sr = live ? s : S - 1;
  float* __restrict__ gs = GS + (int64_t)dir * T * 4 * HID * Sp;
  float* __restrict__ dx_out = dX + (int64_t)dir * T * IN * Sp;
  const int n = live ? len[sr] : 0;
  int nmax = n;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(nmax, m);
    nmax = o > nmax ? o : nmax;
  }
  float dh[KH];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      dh[4 * m + r] = (dHfin && n > 0) ? dHfin[(int64_t)(dir * HID + 16 * m + 4 * b + r) * Sp + sr] : 0.0f;
  const float* __restrict__ thl = th + lane;
  const float* __restrict__ txl = tx + lane;
  for (int step = nmax - 1; step >= 0; --step) {
    const bool act = step < n;
    const int t = act ? (dir ? (n - 1 - step) : step) : 0;
    const int tp = (act && step > 0) ? (dir ? t + 1 : t - 1) : 0;
    float g_r[KH], g_z[KH], g_n[KH], g_h[KH];
    dof_f32x4 d_h[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float r4[4], z4[4], n4[4], a4[4], hp4[4], do4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      dof_ld_row<4>(gs + ACT(t, 16 * m + 4 * b, 4 * HID, Sp, sr), r4);
      dof_ld_row<4>(gs + ACT(t, HID + 16 * m + 4 * b, 4 * HID, Sp, sr), z4);
      dof_ld_row<4>(gs + ACT(t, 2 * HID + 16 * m + 4 * b, 4 * HID, Sp, sr), n4);
      dof_ld_row<4>(gs + ACT(t, 3 * HID + 16 * m + 4 * b, 4 * HID, Sp, sr), a4);
      dof_ld_row<4>(O + ACT(tp, dir * HID + 16 * m + 4 * b, 2 * HID, Sp, sr), hp4);
      if (dO) dof_ld_row<4>(dO + ACT(t, dir * HID + 16 * m + 4 * b, 2 * HID, Sp, sr), do4);   // (wave-uniform condition)
      float dg4[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hp = (act && step > 0) ? hp4[r] : 0.0f;
        const float dht = act ? dh[4 * m + r] + do4[r] : 0.0f;
        const float dn = dht * (1.0f - z4[r]);
        const float dz = dht * (hp - n4[r]);
        const float dnp = dn * (1.0f - n4[r] * n4[r]);
        g_r[4 * m + r] = act ? dnp * a4[r] * r4[r] * (1.0f - r4[r]) : 0.0f;
        g_z[4 * m + r] = act ? dz * z4[r] * (1.0f - z4[r]) : 0.0f;
        g_n[4 * m + r] = act ? dnp : 0.0f;
        g_h[4 * m + r] = act ? dnp * r4[r] : 0.0f;
        d_h[m][r] = act ? dht * z4[r] : 0.0f;
        dg4[0][r] = g_r[4 * m + r]; dg4[1][r] = g_z[4 * m + r]; dg4[2][r] = g_n[4 * m + r]; dg4[3][r] = g_h[4 * m + r];
      }
      if (act) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dof_st_row<4>(gs + ACT(t, g * HID + 16 * m + 4 * b, 4 * HID, Sp, s), dg4[g]);
      }
    }
    dof_f32x4 d_x[XT];
#pragma unroll
    for (int mo = 0; mo < XT; ++mo) d_x[mo] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < KH; ++q) {
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        d_h[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(thl[(mo * KG + q) * 64], g_r[q], d_h[mo], 0, 0, 0);
        d_h[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(thl[(mo * KG + KH + q) * 64], g_z[q], d_h[mo], 0, 0, 0);
        d_h[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(thl[(mo * KG + 2 * KH + q) * 64], g_h[q], d_h[mo], 0, 0, 0);
      }
#pragma unroll
      for (int mo = 0; mo < XT; ++mo) {
        d_x[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(txl[(mo * KG + q) * 64], g_r[q], d_x[mo], 0, 0, 0);
        d_x[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(txl[(mo * KG + KH + q) * 64], g_z[q], d_x[mo], 0, 0, 0);
        d_x[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(txl[(mo * KG + 2 * KH + q) * 64], g_n[q], d_x[mo], 0, 0, 0);
      }
    }
    if (act) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[4 * m + r] = d_h[m][r];
#pragma unroll
      for (int mo = 0; mo < XT; ++mo) {
        const float v[4] = {d_x[mo][0], d_x[mo][1], d_x[mo][2], d_x[mo][3]};
        dof_st_row<4>(dx_out + ACT(t, 16 * mo + 4 * b, IN, Sp, s), v);
      }
    }
  }
  if (!live) return;
  const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int t = n; t < T; ++t)
#pragma unroll
    for (int mo = 0; mo < XT; ++mo) dof_st_row<4>(dx_out + ACT(t, 16 * mo + 4 * b, IN, Sp, s), zero4);
}

// ---------------------------------------------------------------------------------------------
// GRU, weight-stationary form (latent 8: hidden sizes 16 and 8).  One LANE per hidden unit: the
// 16 (or 8) lanes of a DPP row own the units of one (sequence, direction), keep their three gate
// rows of W_ih / W_hh in VGPRs for the whole sequence (no weight traffic inside the time loop) and
// exchange x_t / h_{t-1} with row_newbcast DPP moves.  16x (8x) more wavefronts than the
// one-thread-per-sequence form above, which at batch 1024 was pure latency (<= 1 wave per SIMD).
// Same saved-gate / dG buffers and the same arithmetic; the generic kernels remain the path for
// other latent sizes.
// ---------------------------------------------------------------------------------------------
// (one launch may serve the node and the edge stream of an encoder: blockIdx.z picks the argument set)
struct Gru8Args {
  const float* X;
  const int* len;
  const float *wih0, *whh0, *wih1, *whh1, *bih0, *bhh0, *bih1, *bhh1;
  const float *O, *GS, *dHfin;
  float *dX, *wg_partial, *Oout, *GSout;
  int64_t S, Sp;
  int nblk;  // workgroups per direction of this stream (the partial layout's stride)
};
template <int IN, int HID, bool BCAST>
__global__ void __launch_bounds__(256) k_gru3_fwd(Gru8Args A0, Gru8Args A1, int T) {
  const Gru8Args& AA = blockIdx.z ? A1 : A0;
  const float* __restrict__ X = AA.X;
  const int* __restrict__ len = AA.len;
  const float *__restrict__ wih0 = AA.wih0, *__restrict__ whh0 = AA.whh0, *__restrict__ bih0 = AA.bih0,
              *__restrict__ bhh0 = AA.bhh0, *__restrict__ wih1 = AA.wih1, *__restrict__ whh1 = AA.whh1,
              *__restrict__ bih1 = AA.bih1, *__restrict__ bhh1 = AA.bhh1;
  float *__restrict__ O = AA.Oout, *__restrict__ GS = AA.GSout;
  const int64_t S = AA.S, Sp = AA.Sp;
  // Round 6: hidden sizes that are not a DPP group width (4, 5, 6, 10, 12: latent 4 / 5 / 6) take the next one -- 8 or 16
  // lanes per (sequence, direction), the lanes u >= HID idle: zero weights, a hidden value that stays zero, no stores.
  // (The thread-per-sequence kernels these sizes ran before are pure latency at C2's 28,672 sequences: 155 / 199 us for a
  // GRU(8 -> 8) forward / backward launch, profiles/r06_latent4_kernel_stats_before.md.)
  constexpr int G = HID <= 8 ? 8 : 16;
  static_assert(HID <= 16, "lane-per-unit recurrence: at most 16 hidden units");
  const int ul = threadIdx.x % G;
  const bool unit = ul < HID;
  const int u = unit ? ul : 0;   // idle lanes address unit 0 and multiply by zero
  const int64_t s = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  const int dir = blockIdx.y;
  if (s >= S) return;  // whole groups leave together
  const float* __restrict__ wih = dir ? wih1 : wih0;
  const float* __restrict__ whh = dir ? whh1 : whh0;
  const float* __restrict__ bih = dir ? bih1 : bih0;
  const float* __restrict__ bhh = dir ? bhh1 : bhh0;
  const float live_w = unit ? 1.0f : 0.0f;
  float wr[IN], wz[IN], wn[IN], hr[HID], hz[HID], hnw[HID];
#pragma unroll
  for (int k = 0; k < IN; ++k) {
    wr[k] = wih[u * IN + k] * live_w;
    wz[k] = wih[(HID + u) * IN + k] * live_w;
    wn[k] = wih[(2 * HID + u) * IN + k] * live_w;
  }
#pragma unroll
  for (int k = 0; k < HID; ++k) {
    hr[k] = whh[u * HID + k] * live_w;
    hz[k] = whh[(HID + u) * HID + k] * live_w;
    hnw[k] = whh[(2 * HID + u) * HID + k] * live_w;
  }
  float br = (bih[u] + bhh[u]) * live_w, bz = (bih[HID + u] + bhh[HID + u]) * live_w, bin = bih[2 * HID + u] * live_w;
  const float bhn = bhh[2 * HID + u] * live_w;
  if (BCAST) {  // input constant over time: fold W_ih x into the biases once
#pragma unroll
    for (int k = 0; k < IN; ++k) {
      const float xk = X[(int64_t)k * Sp + s];
      br = fmaf(wr[k], xk, br);
      bz = fmaf(wz[k], xk, bz);
      bin = fmaf(wn[k], xk, bin);
    }
  }
  float* __restrict__ gs = GS ? GS + (int64_t)dir * T * 4 * HID * Sp : nullptr;
  const int n = len[s];
  float h = 0.0f;
  // The input half of the gates (W_ih x_t + b) does not depend on the recurrence: it is computed one step ahead, in the
  // shadow of the dependent W_hh h_{t-1} chain and of the sigmoid / tanh latencies of the current step, and x of the
  // step after that is already in flight (rocprof: 39 % issue stalls + 33 % memory waits in the one-step-at-a-time
  // form, profiles/r02_gru_pmc.md).  Same operations in the same order per gate: bitwise the same result.
  float in_r = br, in_z = bz, in_n = bin;
  float xu_next = 0.0f;
  constexpr bool WIDE = !BCAST && IN != G;
  float xrow_next[WIDE ? IN : 4];
  auto load_x = [&](int step) {
    if (!BCAST && step < n) {
      const int t = dir ? (n - 1 - step) : step;
      if constexpr (WIDE) dof_ld_row<IN>(X + ACT(t, 0, IN, Sp, s), xrow_next);
      else xu_next = X[ACT(t, u, IN, Sp, s)];   // (IN == G here: every lane is a unit)
    }
  };
  auto input_half = [&]() {  // gates' input half from the x held in xu_next / xrow_next
    in_r = br; in_z = bz; in_n = bin;
    if constexpr (WIDE) {  // wider input than the group: every lane holds the (group-uniform) input row
#pragma unroll
      for (int k = 0; k < IN; ++k) {
        in_r = fmaf(wr[k], xrow_next[k], in_r);
        in_z = fmaf(wz[k], xrow_next[k], in_z);
        in_n = fmaf(wn[k], xrow_next[k], in_n);
      }
    } else if constexpr (!BCAST) {
      const float xu = xu_next;
      dof_static_for<IN>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const float b = dof_gbcast<k, G>(xu);
        in_r = fmaf(wr[k], b, in_r);
        in_z = fmaf(wz[k], b, in_z);
        in_n = fmaf(wn[k], b, in_n);
      });
    }
  };
  load_x(0);
  input_half();
  load_x(1);
  for (int step = 0; step < n; ++step) {
    const int t = dir ? (n - 1 - step) : step;
    float ar = in_r, az = in_z, an = in_n, ahn = bhn;
    dof_static_for<HID>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const float b = dof_gbcast<k, G>(h);
      ar = fmaf(hr[k], b, ar);
      az = fmaf(hz[k], b, az);
      ahn = fmaf(hnw[k], b, ahn);
    });
    if (step + 1 < n) input_half();   // next step's input half (x already loaded)
    load_x(step + 2);
    const float r = dof_sigmoid(ar);
    const float z = dof_sigmoid(az);
    const float nn = dof_tanh(fmaf(r, ahn, an));
    h = unit ? fmaf(z, h - nn, nn) : 0.0f;   // (an idle lane: sigmoid(0) h = h / 2 would stay 0 anyway; made explicit)
    if (unit) {
      O[ACT(t, dir * HID + u, 2 * HID, Sp, s)] = h;
      if (gs) {  // unit-major gate buffer: (r, z, n, W_hn h + b_hn) of unit u are one 16-byte word
        const float gate4[4] = {r, z, nn, ahn};
        dof_st_row<4>(gs + ACT(t, 4 * u, 4 * HID, Sp, s), gate4);
      }
    }
  }
  if (!unit) return;
  for (int t = n; t < T; ++t) {
    O[ACT(t, dir * HID + u, 2 * HID, Sp, s)] = 0.0f;
    if (gs) {
      const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      dof_st_row<4>(gs + ACT(t, 4 * u, 4 * HID, Sp, s), zero4);
    }
  }
}

#include "k_grum16.inc.h"   // Gru16mStream, k_gru16x_fwd / _bwd, k_gru8x_fwd / _bwd (latent 8)
#include "k_grumx.inc.h"    // k_grumx_fwd / _bwd (latent 16 / 32)


// WG (round 6, the padded lane groups' layers): the layer's weight gradients in the same launch, the way k_gru16_bwd_fused
// does it for latent 8.  A wavefront's lanes are (unit, sequence) pairs -- the operand layout of v_mfma_f32_16x16x4_f32 --
// so dW_ih[g] += dG_g x_t^T and dW_hh[g] += dG_g h_{t-1}^T are 3 (M + 1) matrix instructions per step straight from the
// registers that hold dG, x_t and h_{t-1} (the matrix pipe idles beside the VALU recurrence; the generic k_outer job spent
// the same instructions in a pass of its own: 239 us of the 1.16 ms latent-6 step).  Groups of 8 lanes put two sequences
// into a tile's sixteen rows: the two diagonal 8 x 8 blocks of the result are the sums, the off-diagonal ones (products
// across sequences) are dropped.  dG is not written.  The workgroup's sums go out as ONE partial tile in k_outer's layout
// (rows 4 u + gate, columns = the job's operand tiles, column 64 = row sums), so k_outer_finalize's fixed-order sum
// applies unchanged.  The time loop runs over the whole window for every lane (the matrix instructions ignore EXEC): a
// finished or absent sequence holds zero operands.  Cost: 24 - 48 accumulator registers take the kernel from 3 to 2
// wavefronts per SIMD and the matrix instructions' issue slots sit inside the dependent chain of a step (measured at
// latent 6: GRU(12 -> 12) 93 -> 106 us, GRU(24 -> 6) 52 -> 93 us per launch, against k_outer 239 -> 36 us; the step 1.145 ->
// 1.088 ms, latent 4 0.830 -> 0.751 ms).  __launch_bounds__(256, 3) spills 50 - 155 registers: not used.
template <int IN, int HID, bool BCAST, bool WG = false>
__global__ void __launch_bounds__(256) k_gru3_bwd(const int* __restrict__ len, const float* __restrict__ wih0,
                                                  const float* __restrict__ whh0, const float* __restrict__ wih1,
                                                  const float* __restrict__ whh1, const float* __restrict__ O,
                                                  float* __restrict__ GS, const float* __restrict__ dO,
                                                  const float* __restrict__ dHfin, float* __restrict__ dX, int T,
                                                  int64_t S, int64_t Sp, const float* __restrict__ X,
                                                  float* __restrict__ wg_part0, float* __restrict__ wg_part1) {
  // (round 6: padded lane groups, see k_gru3_fwd -- lanes u >= HID hold zero gate gradients; a lane owns the input columns
  //  M u .. M u + M - 1 that exist)
  constexpr int G = HID <= 8 ? 8 : 16;
  constexpr int M = (IN + G - 1) / G;
  constexpr int NXT = (IN + 15) / 16;   // WG: operand tiles of x in the k_outer job (the h tile follows them)
  static_assert(HID <= 16, "lane-per-unit recurrence: at most 16 hidden units");
  static_assert(!WG || (!BCAST && NXT + 1 <= 4), "fused weight gradient: a time-varying input of at most 48 columns");
  __shared__ float red[WG ? 64 * 65 : 1];
  const int ul = threadIdx.x % G;
  const bool unit = ul < HID;
  const int u = unit ? ul : 0;
  const int64_t s = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  const int dir = blockIdx.y;
  const bool valid = s < S;
  if (!WG && !valid) return;
  const float* __restrict__ wih = dir ? wih1 : wih0;
  const float* __restrict__ whh = dir ? whh1 : whh0;
  // transposed views: column u of W_hh, columns M ul + m of W_ih
  float tr[HID], tz[HID], tn[HID];
  float xr[M][HID], xz[M][HID], xn[M][HID];
  bool col[M];
#pragma unroll
  for (int m = 0; m < M; ++m) col[m] = M * ul + m < IN;
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    tr[j] = unit ? whh[j * HID + u] : 0.0f;
    tz[j] = unit ? whh[(HID + j) * HID + u] : 0.0f;
    tn[j] = unit ? whh[(2 * HID + j) * HID + u] : 0.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int c = col[m] ? M * ul + m : 0;
      xr[m][j] = col[m] ? wih[j * IN + c] : 0.0f;
      xz[m][j] = col[m] ? wih[(HID + j) * IN + c] : 0.0f;
      xn[m][j] = col[m] ? wih[(2 * HID + j) * IN + c] : 0.0f;
    }
  }
  float* __restrict__ gs = GS + (int64_t)dir * T * 4 * HID * Sp;
  float* __restrict__ dx_out = dX + (int64_t)dir * (BCAST ? 1 : T) * IN * Sp;
  const int n = valid ? len[s] : 0;
  const float dhfin = (dHfin && n > 0 && unit) ? dHfin[(int64_t)(dir * HID + u) * Sp + s] : 0.0f;
  float dh = WG ? 0.0f : dhfin;   // WG: added at the lane's own last step (the loop starts at the window's)
  float dxacc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) dxacc[m] = 0.0f;
  dof_f32x4 accx[WG ? M : 1][3], acch[3];
  float rsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    acch[g] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < (WG ? M : 1); ++m) accx[m][g] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // loads of step - PF are issued before the arithmetic of step (see k_gru16_bwd_fused)
  constexpr int PF = 3;
  float nx_gate[PF][4], nx_hp[PF], nx_do[PF], nx_x[PF][WG ? M : 1];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    nx_gate[d][0] = nx_gate[d][1] = nx_gate[d][2] = nx_gate[d][3] = 0.0f;
    nx_hp[d] = nx_do[d] = 0.0f;
#pragma unroll
    for (int m = 0; m < (WG ? M : 1); ++m) nx_x[d][m] = 0.0f;
  }
  auto issue_loads = [&](auto slot_c, int step) {
    constexpr int slot = decltype(slot_c)::value;
    if (step >= 0 && step < n) {
      const int t = dir ? (n - 1 - step) : step;
      const int tp = dir ? t + 1 : t - 1;
      dof_ld_row<4>(gs + ACT(t, 4 * u, 4 * HID, Sp, s), nx_gate[slot]);
      nx_hp[slot] = (step > 0) ? O[ACT(tp, dir * HID + u, 2 * HID, Sp, s)] : 0.0f;
      nx_do[slot] = (dO && unit) ? dO[ACT(t, dir * HID + u, 2 * HID, Sp, s)] : 0.0f;   // (idle lane: dht = 0, every gate gradient 0)
      if constexpr (WG) {
        if constexpr (M == 4 && IN == 4 * G) {
          dof_ld_row<4>(X + ACT(t, 4 * ul, IN, Sp, s), nx_x[slot]);
        } else {
#pragma unroll
          for (int m = 0; m < M; ++m) nx_x[slot][m] = col[m] ? X[ACT(t, M * ul + m, IN, Sp, s)] : 0.0f;
        }
      }
    }
  };
  auto do_step = [&](auto slot_c, int step) {
    constexpr int slot = decltype(slot_c)::value;
    const bool active = !WG || step < n;
    const int t = dir ? (n - 1 - step) : step;
    const float r = nx_gate[slot][0], z = nx_gate[slot][1], nn = nx_gate[slot][2], ahn = nx_gate[slot][3];
    const float hp = nx_hp[slot];
    const float dht = dh + nx_do[slot] + ((WG && step == n - 1) ? dhfin : 0.0f);
    float xv[WG ? M : 1];
#pragma unroll
    for (int m = 0; m < (WG ? M : 1); ++m) xv[m] = nx_x[slot][m];
    issue_loads(slot_c, step - PF);
    const float dn = dht * (1.0f - z);
    const float dz = dht * (hp - nn);
    const float dnp = dn * (1.0f - nn * nn);
    const float g_r = dnp * ahn * r * (1.0f - r);
    const float g_z = dz * z * (1.0f - z);
    const float g_n = dnp;
    const float g_h = dnp * r;
    if constexpr (WG) {
      // an inactive step's slot holds the zeros it was created with or an already consumed step: dht = 0 makes every gate
      // gradient 0 for a lane that has not reached its sequence yet; the operands are cleared explicitly all the same
      const float a_r = active ? g_r : 0.0f, a_z = active ? g_z : 0.0f, a_n = active ? g_n : 0.0f, a_h = active ? g_h : 0.0f;
      const float bh = (active && unit) ? hp : 0.0f;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float bx = (active && col[m]) ? xv[m] : 0.0f;
        accx[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r, bx, accx[m][0], 0, 0, 0);
        accx[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_z, bx, accx[m][1], 0, 0, 0);
        accx[m][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_n, bx, accx[m][2], 0, 0, 0);
      }
      acch[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r, bh, acch[0], 0, 0, 0);
      acch[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_z, bh, acch[1], 0, 0, 0);
      acch[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_h, bh, acch[2], 0, 0, 0);
      rsum[0] += a_r; rsum[1] += a_z; rsum[2] += a_n; rsum[3] += a_h;
    } else {
      const float dg4[4] = {g_r, g_z, g_n, g_h};
      if (unit) dof_st_row<4>(gs + ACT(t, 4 * u, 4 * HID, Sp, s), dg4);
    }
    float dhp = dht * z;
    float dx[M];
#pragma unroll
    for (int m = 0; m < M; ++m) dx[m] = 0.0f;
    dof_static_for<HID>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const float b_r = dof_gbcast<j, G>(g_r);
      const float b_z = dof_gbcast<j, G>(g_z);
      const float b_n = dof_gbcast<j, G>(g_n);
      const float b_h = dof_gbcast<j, G>(g_h);
      dhp = fmaf(tr[j], b_r, dhp);
      dhp = fmaf(tz[j], b_z, dhp);
      dhp = fmaf(tn[j], b_h, dhp);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        dx[m] = fmaf(xr[m][j], b_r, dx[m]);
        dx[m] = fmaf(xz[m][j], b_z, dx[m]);
        dx[m] = fmaf(xn[m][j], b_n, dx[m]);
      }
    });
    dh = (unit && active) ? dhp : 0.0f;
    if (BCAST) {
#pragma unroll
      for (int m = 0; m < M; ++m) dxacc[m] += dx[m];
    } else if (!active) {
      // (WG: a step in front of the lane's sequence -- nothing to store)
    } else if (M == 4 && IN == 4 * G) {  // lane u owns input columns 4u .. 4u+3: one 16-byte store
      dof_st_row<4>(dx_out + ACT(t, 4 * ul, IN, Sp, s), dx);
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m)
        if (col[m]) dx_out[ACT(t, M * ul + m, IN, Sp, s)] = dx[m];
    }
  };
  const int nloop = WG ? T : n;
  dof_static_for<PF>([&](auto d) { issue_loads(d, nloop - 1 - decltype(d)::value); });
  for (int step = nloop - 1; step >= 0; step -= PF) {
    dof_static_for<PF>([&](auto d) {
      const int st = step - decltype(d)::value;
      if (st >= 0) do_step(d, st);
    });
  }
  if (valid) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (!col[m]) continue;
      if (BCAST) {
        dx_out[(int64_t)(M * ul + m) * Sp + s] = dxacc[m];
      } else {
        for (int t = n; t < T; ++t) dx_out[ACT(t, M * ul + m, IN, Sp, s)] = 0.0f;
      }
    }
  }
  if constexpr (WG) {
    // the workgroup's partial tile: the four wavefronts in order through one LDS tile (deterministic)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 64 * 65; e += 256) red[e] = 0.0f;
    // row sums over the wavefront's sequences (lanes that share a unit)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = rsum[g];
#pragma unroll
      for (int d = G; d < 64; d <<= 1) v += __shfl_xor(v, d);
      rsum[g] = v;
    }
    __syncthreads();
    // D of v_mfma_f32_16x16x4: the lane holds rows (lane >> 4) 4 + r, column lane & 15
    const int dj = lane & 15;
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
        if (lane < G && unit) {
#pragma unroll
          for (int g = 0; g < 4; ++g) red[(4 * ul + g) * 65 + 64] += rsum[g];
        }
#pragma unroll
        for (int half = 0; half < (G == 8 ? 2 : 1); ++half) {   // groups of 8: the two diagonal blocks, one after the other
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int di = (lane >> 4) * 4 + r4;
            const bool mine = G == 16 || ((di >> 3) == half && (dj >> 3) == half);
            const int ui = G == 16 ? di : (di & 7), uj = G == 16 ? dj : (dj & 7);
            if (mine && ui < HID) {
#pragma unroll
              for (int m = 0; m < M; ++m) {
                const int c = M * uj + m;
                if (c < IN) {
#pragma unroll
                  for (int g = 0; g < 3; ++g) red[(4 * ui + g) * 65 + c] += accx[m][g][r4];
                }
              }
              if (uj < HID) {
                red[(4 * ui + 0) * 65 + 16 * NXT + uj] += acch[0][r4];
                red[(4 * ui + 1) * 65 + 16 * NXT + uj] += acch[1][r4];
                red[(4 * ui + 3) * 65 + 16 * NXT + uj] += acch[2][r4];
              }
            }
          }
        }
      }
      __syncthreads();
    }
    float* __restrict__ outp = (dir ? wg_part1 : wg_part0) + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
    for (int e = threadIdx.x; e < 4 * HID * 65; e += 256) {
      const int col_e = e % 65;
      if (col_e < 16 * NXT + HID || col_e == 64) outp[e] = red[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GRU backward (IN = HID = 16) with the weight-gradient reduction fused in.
// In the lane-per-unit mapping a wavefront holds 4 sequences x 16 units, which IS the operand
// layout of v_mfma_f32_16x16x4_f32 (lane = (row i = unit, k = sequence)): the outer products
//   dW_ih[g] += dG_g (16 units x 4 seq) . X^T (4 seq x 16 inputs),  dW_hh[g] += dG_g . Hprev^T
// are 6 MFMAs per step straight out of the registers that already hold dG, x_t and h_{t-1}; the
// matrix pipe runs beside the VALU recurrence.  dG is never written to HBM and the separate
// k_outer pass over it disappears.  The time loop is wave-uniform (MFMA ignores EXEC), finished
// sequences contribute zeros.  Per-workgroup partial tiles -> k_gru16_wg_finalize (fixed order).
// ---------------------------------------------------------------------------------------------
#ifndef GRU16_WG_FLOATS
#define GRU16_WG_FLOATS (6 * 256 + 4 * 16)
#endif
__global__ void __launch_bounds__(256) k_gru16_bwd_fused(
    const float* __restrict__ X, const int* __restrict__ len, const float* __restrict__ wih0,
    const float* __restrict__ whh0, const float* __restrict__ wih1, const float* __restrict__ whh1,
    const float* __restrict__ O, const float* __restrict__ GS, const float* __restrict__ dO,
    float* __restrict__ dX, float* __restrict__ wg_partial, int T, int64_t S, int64_t Sp) {
  constexpr int HID = 16, IN = 16, G = 16;
  __shared__ float red[4][6][256];
  __shared__ float bred[256][5];
  const int u = threadIdx.x & 15;
  const int64_t s = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const float* __restrict__ wih = dir ? wih1 : wih0;
  const float* __restrict__ whh = dir ? whh1 : whh0;
  float tr[HID], tz[HID], tn[HID], xr[HID], xz[HID], xn[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    tr[j] = whh[j * HID + u];
    tz[j] = whh[(HID + j) * HID + u];
    tn[j] = whh[(2 * HID + j) * HID + u];
    xr[j] = wih[j * IN + u];
    xz[j] = wih[(HID + j) * IN + u];
    xn[j] = wih[(2 * HID + j) * IN + u];
  }
  const float* __restrict__ gs = GS + (int64_t)dir * T * 4 * HID * Sp;
  float* __restrict__ dx_out = dX + (int64_t)dir * T * IN * Sp;
  const int n = in_range ? len[s] : 0;
  float dh = 0.0f;
  dof_f32x4 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[a] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float sb_r = 0.0f, sb_z = 0.0f, sb_n = 0.0f, sb_h = 0.0f;
  // Software pipeline, PF steps deep: the loads of step - PF (saved gates, h_{t-1}, x_t, dO_t: none of them depends
  // on the recurrence) are issued before the arithmetic of step.  One step is ~230 VALU instructions (~0.3 us); a
  // load from these 100 MB-class buffers, whose time steps lie megabytes apart, takes ~0.5 us -- measured: 0.85 us
  // per step on the decoder's lightly loaded launch without prefetch distance (profiles/r02_gru_pmc.md: 52 % of
  // the wave-cycles parked on memory at 2 waves per SIMD).  Register slots are static: the loop is unrolled by PF.
  constexpr int PF = 3;
  float nx_gate[PF][4], nx_hp[PF], nx_xu[PF], nx_do[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    nx_gate[d][0] = nx_gate[d][1] = nx_gate[d][2] = nx_gate[d][3] = 0.0f;
    nx_hp[d] = nx_xu[d] = nx_do[d] = 0.0f;
  }
  auto issue_loads = [&](auto slot_c, int step) {
    constexpr int slot = decltype(slot_c)::value;
    if (step >= 0 && step < n) {
      const int t = dir ? (n - 1 - step) : step;
      const int tp = dir ? t + 1 : t - 1;
      dof_ld_row<4>(gs + ACT(t, 4 * u, 4 * HID, Sp, s), nx_gate[slot]);
      nx_hp[slot] = (step > 0) ? O[ACT(tp, dir * HID + u, 2 * HID, Sp, s)] : 0.0f;
      nx_xu[slot] = X[ACT(t, u, IN, Sp, s)];
      nx_do[slot] = dO ? dO[ACT(t, dir * HID + u, 2 * HID, Sp, s)] : 0.0f;
    }
  };
  auto do_step = [&](auto slot_c, int step) {
    constexpr int slot = decltype(slot_c)::value;
    const bool act = step < n;
    const int t = dir ? (n - 1 - step) : step;
    float g_r = 0.0f, g_z = 0.0f, g_n = 0.0f, g_h = 0.0f, hp = 0.0f, xu = 0.0f, dht = 0.0f, z = 0.0f;
    const float r = nx_gate[slot][0], zc = nx_gate[slot][1], nn = nx_gate[slot][2], ahn = nx_gate[slot][3];
    const float hp_c = nx_hp[slot], xu_c = nx_xu[slot], do_c = nx_do[slot];
    issue_loads(slot_c, step - PF);
    if (act) {
      z = zc;
      hp = hp_c;
      xu = xu_c;
      dht = dh + do_c;
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp - nn);
      const float dnp = dn * (1.0f - nn * nn);
      g_r = dnp * ahn * r * (1.0f - r);
      g_z = dz * z * (1.0f - z);
      g_n = dnp;
      g_h = dnp * r;
    }
    // weight-gradient tiles: rows = unit (this lane), k = sequence (lane >> 4), cols = input / hidden index
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_r, xu, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_z, xu, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_n, xu, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_r, hp, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_z, hp, acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_h, hp, acc[5], 0, 0, 0);
    sb_r += g_r; sb_z += g_z; sb_n += g_n; sb_h += g_h;
    float dhp = dht * z;
    float dx = 0.0f;
    dof_static_for<HID>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const float b_r = dof_gbcast<j, G>(g_r);
      const float b_z = dof_gbcast<j, G>(g_z);
      const float b_n = dof_gbcast<j, G>(g_n);
      const float b_h = dof_gbcast<j, G>(g_h);
      dhp = fmaf(tr[j], b_r, dhp);
      dhp = fmaf(tz[j], b_z, dhp);
      dhp = fmaf(tn[j], b_h, dhp);
      dx = fmaf(xr[j], b_r, dx);
      dx = fmaf(xz[j], b_z, dx);
      dx = fmaf(xn[j], b_n, dx);
    });
    if (act) {
      dh = dhp;
      dx_out[ACT(t, u, IN, Sp, s)] = dx;
    }
  };
  dof_static_for<PF>([&](auto d) { issue_loads(d, T - 1 - decltype(d)::value); });
  for (int step = T - 1; step >= 0; step -= PF) {   // wave-uniform trip count (MFMA ignores EXEC)
    dof_static_for<PF>([&](auto d) {
      const int st = step - decltype(d)::value;
      if (st >= 0) do_step(d, st);
    });
  }
  if (in_range)
    for (int t = n; t < T; ++t) dx_out[ACT(t, u, IN, Sp, s)] = 0.0f;
  // ---- workgroup reduction of the 4 waves' tiles and of the bias sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) red[wave][a][((lane >> 4) * 4 + r4) * 16 + (lane & 15)] = acc[a][r4];
  bred[threadIdx.x][0] = sb_r; bred[threadIdx.x][1] = sb_z; bred[threadIdx.x][2] = sb_n; bred[threadIdx.x][3] = sb_h;
  __syncthreads();
  float* __restrict__ out = wg_partial + ((int64_t)dir * gridDim.x + blockIdx.x) * GRU16_WG_FLOATS;
  for (int e = threadIdx.x; e < 6 * 256; e += 256) {
    const int a = e >> 8, c = e & 255;
    out[e] = (red[0][a][c] + red[1][a][c]) + (red[2][a][c] + red[3][a][c]);
  }
  if (threadIdx.x < 64) {  // (gate, unit): sum over the 16 sequences of the workgroup
    const int gate = threadIdx.x >> 4, unit = threadIdx.x & 15;
    float acc_b = 0.0f;
    for (int g = 0; g < 16; ++g) acc_b += bred[g * 16 + unit][gate];
    out[6 * 256 + threadIdx.x] = acc_b;
  }
}


// grads of a (16,16) GRU layer from the per-workgroup partials [dir][nblk][GRU16_WG_FLOATS].  Workgroup = 8
// neighbouring values x 32 row slices: lane (slice, value) adds rows slice, slice + 32, ... (four running sums), so a
// wave-wide load touches 8 rows x 32 contiguous bytes; the 32 slice sums of a value are added in slice order through
// LDS.  (First form: one workgroup per value with lanes striding over the rows -- 2.9 M scattered dword loads, 64
// cache lines per load instruction, 18 us at 14,336 sequences.  A two-stage form with a last-workgroup ticket was
// measured at 55 us: the device-scope fences write back an L2 full of the backward kernel's output.)
constexpr int kWgVals = 8;
static_assert(GRU16_WG_FLOATS % kWgVals == 0, "value groups");
// (one launch may serve two layers -- the node and the edge stream of an encoder: blockIdx.z picks the argument set)
struct WgFinArgs {
  const float* wg_partial;
  int nblk;
  float* g[8];  // wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1
};
__device__ __forceinline__ void gru16_wg_finalize_body(const WgFinArgs& A, int bx, int dir, int accumulate, float (*red)[kWgVals + 1]) {
  const float* __restrict__ wg_partial = A.wg_partial;
  const int nblk = A.nblk;
  float *g_wih0 = A.g[0], *g_whh0 = A.g[1], *g_bih0 = A.g[2], *g_bhh0 = A.g[3], *g_wih1 = A.g[4], *g_whh1 = A.g[5],
        *g_bih1 = A.g[6], *g_bhh1 = A.g[7];
  const int vi = (int)threadIdx.x & (kWgVals - 1), slice = (int)threadIdx.x / kWgVals;
  const float* __restrict__ p = wg_partial + (int64_t)dir * nblk * GRU16_WG_FLOATS + bx * kWgVals + vi;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int r = slice;
  for (; r + 96 < nblk; r += 128) {
    a0 += p[(int64_t)r * GRU16_WG_FLOATS];
    a1 += p[(int64_t)(r + 32) * GRU16_WG_FLOATS];
    a2 += p[(int64_t)(r + 64) * GRU16_WG_FLOATS];
    a3 += p[(int64_t)(r + 96) * GRU16_WG_FLOATS];
  }
  for (; r < nblk; r += 32) a0 += p[(int64_t)r * GRU16_WG_FLOATS];
  red[slice][vi] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x >= kWgVals) return;
  float val = 0.0f;
#pragma unroll
  for (int k = 0; k < 32; ++k) val += red[k][vi];
  const int v = bx * kWgVals + vi;
  float* g_wih = dir ? g_wih1 : g_wih0;
  float* g_whh = dir ? g_whh1 : g_whh0;
  float* g_bih = dir ? g_bih1 : g_bih0;
  float* g_bhh = dir ? g_bhh1 : g_bhh0;
  if (v < 6 * 256) {
    const int tile = v >> 8, row = (v & 255) >> 4, col = v & 15;
    float* d = tile < 3 ? &g_wih[(tile * 16 + row) * 16 + col] : &g_whh[((tile - 3) * 16 + row) * 16 + col];
    *d = accumulate ? *d + val : val;
  } else {
    const int gate = (v - 6 * 256) >> 4, unit = v & 15;  // gates: r, z, n(input side), hn(hidden side)
    float* d0 = gate < 2 ? &g_bih[gate * 16 + unit] : (gate == 2 ? &g_bih[32 + unit] : &g_bhh[32 + unit]);
    *d0 = accumulate ? *d0 + val : val;
    if (gate < 2) {
      float* d1 = &g_bhh[gate * 16 + unit];
      *d1 = accumulate ? *d1 + val : val;
    }
  }
}
__global__ void __launch_bounds__(256) k_gru16_wg_finalize(WgFinArgs A0, WgFinArgs A1, WgFinArgs A2, int accumulate) {
  __shared__ float red[32][kWgVals + 1];
  gru16_wg_finalize_body(blockIdx.z == 0 ? A0 : blockIdx.z == 1 ? A1 : A2, (int)blockIdx.x, (int)blockIdx.y, accumulate, red);
}

// ---------------------------------------------------------------------------------------------
// GRU backward (IN = 32, HID = 8: the second encoder layer of latent 8) with the weight-gradient reduction fused in,
// the twin of k_gru16_bwd_fused.  Eight lanes own one (sequence, direction), so a wavefront holds 8 sequences x 8
// units: lane = seq * 8 + u, i.e. MFMA row index lane & 15 = (p, u) with p = the parity of the sequence, k index
// lane >> 4 = the sequence pair.  With A = a gate gradient of the lane's own (u, sequence) and B = an input value of
// the lane's own sequence (column (p', u') <-> input channel 4 u' + m of tile m, or hidden unit u'), the product
//   D[(p, u)][(p', u')] = sum_pairs g[u, 2k + p] x[4u' + m, 2k + p']
// carries the wanted outer products in its two diagonal 8 x 8 blocks (p = p'); the off-diagonal blocks are wasted
// matrix-pipe work that nobody waits for (15 MFMAs per step beside ~150 VALU instructions).  dG is never written
// (92 MB per stream at batch 1024) and the generic reduction's pass over dG, the layer input and the hidden states
// (the largest part of the step's last k_outer launch) disappears.  Per-workgroup partials [dir][nblk][GRU8_WG_FLOATS]:
// 15 tiles of 8 x 8 (W_ih: gates r, z, n x input tiles m = 0..3; W_hh: r, z, hn) + 4 x 8 bias sums.
// ---------------------------------------------------------------------------------------------
#define GRU8_WG_FLOATS (15 * 64 + 4 * 8)
__global__ void __launch_bounds__(256, 2) k_gru8_bwd_fused(Gru8Args A0, Gru8Args A1, int T) {
  const Gru8Args& AA = blockIdx.z ? A1 : A0;
  if ((int)blockIdx.x >= AA.nblk) return;
  const float* __restrict__ X = AA.X;
  const int* __restrict__ len = AA.len;
  const float *__restrict__ wih0 = AA.wih0, *__restrict__ whh0 = AA.whh0, *__restrict__ wih1 = AA.wih1,
              *__restrict__ whh1 = AA.whh1;
  const float *__restrict__ O = AA.O, *__restrict__ GS = AA.GS, *__restrict__ dHfin = AA.dHfin;
  float *__restrict__ dX = AA.dX, *__restrict__ wg_partial = AA.wg_partial;
  const int64_t S = AA.S, Sp = AA.Sp;
  constexpr int HID = 8, IN = 32, G = 8, M = 4;
  __shared__ float red[4][15][128];
  __shared__ float bred[256][5];
  // W_ih^T for the input gradient: 96 values per unit column would not fit beside the 60 accumulator registers at two
  // waves per SIMD, so they wait in LDS ([u][j][gate][m], rows padded to 100 floats: the eight rows a ds_read_b128
  // touches start in distinct bank groups) and are read inside the step (24 reads per step beside ~150 VALU ops)
  __shared__ float wx[HID][100];
  const int u = threadIdx.x & 7;
  const int64_t s = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const int64_t sr = in_range ? s : S - 1;  // lanes of sequences past the end read a valid row and contribute zeros
  const float* __restrict__ wih = dir ? wih1 : wih0;
  const float* __restrict__ whh = dir ? whh1 : whh0;
  float tr[HID], tz[HID], tn[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    tr[j] = whh[j * HID + u];
    tz[j] = whh[(HID + j) * HID + u];
    tn[j] = whh[(2 * HID + j) * HID + u];
  }
  for (int e = threadIdx.x; e < HID * 96; e += 256) {
    const int uu = e / 96, rem = e - uu * 96, j = rem / 12, gm = rem - j * 12, g = gm >> 2, m = gm & 3;
    wx[uu][rem] = wih[(g * HID + j) * IN + M * uu + m];
  }
  __syncthreads();
  const float* __restrict__ gs = GS + (int64_t)dir * T * 4 * HID * Sp;
  float* __restrict__ dx_out = dX + (int64_t)dir * T * IN * Sp;
  const int n = in_range ? len[s] : 0;
  float dh = (dHfin && n > 0) ? dHfin[(int64_t)(dir * HID + u) * Sp + s] : 0.0f;
  dof_f32x4 acc[15];
#pragma unroll
  for (int a = 0; a < 15; ++a) acc[a] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float sb_r = 0.0f, sb_z = 0.0f, sb_n = 0.0f, sb_h = 0.0f;
  constexpr int PF = 3;  // loads of step - PF are issued before the arithmetic of step (see k_gru16_bwd_fused)
  float nx_gate[PF][4], nx_hp[PF], nx_x[PF][M];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    nx_gate[d][0] = nx_gate[d][1] = nx_gate[d][2] = nx_gate[d][3] = 0.0f;
    nx_hp[d] = 0.0f;
#pragma unroll
    for (int m = 0; m < M; ++m) nx_x[d][m] = 0.0f;
  }
  auto issue_loads = [&](auto slot_c, int step) {
    constexpr int slot = decltype(slot_c)::value;
    if (step >= 0 && step < n) {
      const int t = dir ? (n - 1 - step) : step;
      const int tp = dir ? t + 1 : t - 1;
      dof_ld_row<4>(gs + ACT(t, 4 * u, 4 * HID, Sp, sr), nx_gate[slot]);
      nx_hp[slot] = (step > 0) ? O[ACT(tp, dir * HID + u, 2 * HID, Sp, sr)] : 0.0f;
      dof_ld_row<4>(X + ACT(t, M * u, IN, Sp, sr), nx_x[slot]);
    }
  };
  auto do_step = [&](auto slot_c, int step) {
    constexpr int slot = decltype(slot_c)::value;
    DOF_MEM_FENCE();  // keeps the LDS weight reads inside the step (hoisted, they would take the registers back)
    const bool act = step < n;
    const int t = dir ? (n - 1 - step) : step;
    float g_r = 0.0f, g_z = 0.0f, g_n = 0.0f, g_h = 0.0f, hp = 0.0f, dht = 0.0f, z = 0.0f;
    float xv[M] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float r = nx_gate[slot][0], zc = nx_gate[slot][1], nn = nx_gate[slot][2], ahn = nx_gate[slot][3];
    const float hp_c = nx_hp[slot];
    const float x_c[M] = {nx_x[slot][0], nx_x[slot][1], nx_x[slot][2], nx_x[slot][3]};
    issue_loads(slot_c, step - PF);
    if (act) {
      z = zc;
      hp = hp_c;
#pragma unroll
      for (int m = 0; m < M; ++m) xv[m] = x_c[m];
      dht = dh;
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp - nn);
      const float dnp = dn * (1.0f - nn * nn);
      g_r = dnp * ahn * r * (1.0f - r);
      g_z = dz * z * (1.0f - z);
      g_n = dnp;
      g_h = dnp * r;
    }
    // weight-gradient tiles (the time loop is wave-uniform: MFMA ignores EXEC, inactive lanes multiply zeros)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_r, xv[m], acc[m], 0, 0, 0);
      acc[4 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_z, xv[m], acc[4 + m], 0, 0, 0);
      acc[8 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_n, xv[m], acc[8 + m], 0, 0, 0);
    }
    acc[12] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_r, hp, acc[12], 0, 0, 0);
    acc[13] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_z, hp, acc[13], 0, 0, 0);
    acc[14] = __builtin_amdgcn_mfma_f32_16x16x4f32(g_h, hp, acc[14], 0, 0, 0);
    sb_r += g_r; sb_z += g_z; sb_n += g_n; sb_h += g_h;
    float dhp = dht * z;
    float dx[M] = {0.0f, 0.0f, 0.0f, 0.0f};
    dof_static_for<HID>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const float b_r = dof_gbcast<j, G>(g_r);
      const float b_z = dof_gbcast<j, G>(g_z);
      const float b_n = dof_gbcast<j, G>(g_n);
      const float b_h = dof_gbcast<j, G>(g_h);
      dhp = fmaf(tr[j], b_r, dhp);
      dhp = fmaf(tz[j], b_z, dhp);
      dhp = fmaf(tn[j], b_h, dhp);
      float wr4[4], wz4[4], wn4[4];
      dof_ld_row<4>(&wx[u][j * 12], wr4);
      dof_ld_row<4>(&wx[u][j * 12 + 4], wz4);
      dof_ld_row<4>(&wx[u][j * 12 + 8], wn4);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        dx[m] = fmaf(wr4[m], b_r, dx[m]);
        dx[m] = fmaf(wz4[m], b_z, dx[m]);
        dx[m] = fmaf(wn4[m], b_n, dx[m]);
      }
    });
    if (act) {
      dh = dhp;
      dof_st_row<4>(dx_out + ACT(t, M * u, IN, Sp, s), dx);
    }
  };
  dof_static_for<PF>([&](auto d) { issue_loads(d, T - 1 - decltype(d)::value); });
  for (int step = T - 1; step >= 0; step -= PF) {  // wave-uniform trip count
    dof_static_for<PF>([&](auto d) {
      const int st = step - decltype(d)::value;
      if (st >= 0) do_step(d, st);
    });
  }
  if (in_range) {
    const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int t = n; t < T; ++t) dof_st_row<4>(dx_out + ACT(t, M * u, IN, Sp, s), zero4);
  }
  // ---- workgroup reduction: the diagonal blocks of the 4 waves' tiles, and the bias sums
  // D layout: lane holds rows (lane >> 4) * 4 + r4 = (p, u), column lane & 15 = (p', u')
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, pcol = col >> 3, ucol = col & 7;
#pragma unroll
  for (int a = 0; a < 15; ++a)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int row = (lane >> 4) * 4 + r4;
      if ((row >> 3) == pcol) red[wave][a][(pcol * 8 + (row & 7)) * 8 + ucol] = acc[a][r4];
    }
  bred[threadIdx.x][0] = sb_r; bred[threadIdx.x][1] = sb_z; bred[threadIdx.x][2] = sb_n; bred[threadIdx.x][3] = sb_h;
  __syncthreads();
  float* __restrict__ out = wg_partial + ((int64_t)dir * AA.nblk + blockIdx.x) * GRU8_WG_FLOATS;
  for (int e = threadIdx.x; e < 15 * 64; e += 256) {
    const int a = e >> 6, c = e & 63;
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[w][a][c] + red[w][a][64 + c];
    out[e] = v;
  }
  if (threadIdx.x < 32) {  // (gate, unit): sum over the 32 sequences of the workgroup
    const int gate = threadIdx.x >> 3, unit = threadIdx.x & 7;
    float acc_b = 0.0f;
    for (int g = 0; g < 32; ++g) acc_b += bred[g * 8 + unit][gate];
    out[15 * 64 + threadIdx.x] = acc_b;
  }
}

// grads of the (32 -> 8) GRU layer from the per-workgroup partials [dir][nblk][GRU8_WG_FLOATS] (same slicing as
// k_gru16_wg_finalize).  Value v < 960: tile = v / 64 (0-3: r x m, 4-7: z x m, 8-11: n x m, 12-14: hidden r, z, hn),
// row u = (v % 64) / 8, column u' = v % 8: W_ih[gate * 8 + u][4 u' + m] or W_hh[gate * 8 + u][u'].
static_assert(GRU8_WG_FLOATS % 8 == 0, "value groups");
__global__ void __launch_bounds__(256) k_gru8_wg_finalize(WgFinArgs A0, WgFinArgs A1, int accumulate) {
  const WgFinArgs& A = blockIdx.z ? A1 : A0;
  const float* __restrict__ wg_partial = A.wg_partial;
  const int nblk = A.nblk;
  float *g_wih0 = A.g[0], *g_whh0 = A.g[1], *g_bih0 = A.g[2], *g_bhh0 = A.g[3], *g_wih1 = A.g[4], *g_whh1 = A.g[5],
        *g_bih1 = A.g[6], *g_bhh1 = A.g[7];
  constexpr int NV = 8;
  __shared__ float red[32][NV + 1];
  const int dir = blockIdx.y;
  const int vi = (int)threadIdx.x & (NV - 1), slice = (int)threadIdx.x / NV;
  const float* __restrict__ p = wg_partial + (int64_t)dir * nblk * GRU8_WG_FLOATS + blockIdx.x * NV + vi;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int r = slice;
  for (; r + 96 < nblk; r += 128) {
    a0 += p[(int64_t)r * GRU8_WG_FLOATS];
    a1 += p[(int64_t)(r + 32) * GRU8_WG_FLOATS];
    a2 += p[(int64_t)(r + 64) * GRU8_WG_FLOATS];
    a3 += p[(int64_t)(r + 96) * GRU8_WG_FLOATS];
  }
  for (; r < nblk; r += 32) a0 += p[(int64_t)r * GRU8_WG_FLOATS];
  red[slice][vi] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x >= NV) return;
  float val = 0.0f;
#pragma unroll
  for (int k = 0; k < 32; ++k) val += red[k][vi];
  const int v = blockIdx.x * NV + vi;
  float* g_wih = dir ? g_wih1 : g_wih0;
  float* g_whh = dir ? g_whh1 : g_whh0;
  float* g_bih = dir ? g_bih1 : g_bih0;
  float* g_bhh = dir ? g_bhh1 : g_bhh0;
  if (v < 15 * 64) {
    const int tile = v >> 6, uu = (v & 63) >> 3, uc = v & 7;
    float* d = tile < 12 ? &g_wih[((tile >> 2) * 8 + uu) * 32 + 4 * uc + (tile & 3)]
                         : &g_whh[((tile - 12) * 8 + uu) * 8 + uc];
    *d = accumulate ? *d + val : val;
  } else {
    const int gate = (v - 15 * 64) >> 3, unit = v & 7;  // gates: r, z, n (input side), hn (hidden side)
    float* d0 = gate < 2 ? &g_bih[gate * 8 + unit] : (gate == 2 ? &g_bih[16 + unit] : &g_bhh[16 + unit]);
    *d0 = accumulate ? *d0 + val : val;
    if (gate < 2) {
      float* d1 = &g_bhh[gate * 8 + unit];
      *d1 = accumulate ? *d1 + val : val;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels (eps = 1e-3), thread = (t, s).
// ---------------------------------------------------------------------------------------------
// grads of a (32 -> 8) GRU layer from k_gru8x_bwd's per-wavefront partials [dir][nblk][GRU8X_WG_FLOATS] (the reference's
// tensors in order: weight_ih, weight_hh, then the bias sums of g_r, g_z, g_n, g_h); same reduction shape as
// k_gru16_wg_finalize (8 neighbouring values x 32 row slices per workgroup, slices added in a fixed order)
__device__ __forceinline__ void gru8x_wg_finalize_body(const WgFinArgs& A, int bx, int dir, int accumulate, float (*red)[kWgVals + 1]) {
  const float* __restrict__ wg_partial = A.wg_partial;
  const int nblk = A.nblk;
  constexpr int NV = 8;
  static_assert(GRU8X_WG_FLOATS % NV == 0 && NV == kWgVals, "value groups");
  const int vi = (int)threadIdx.x & (NV - 1), slice = (int)threadIdx.x / NV;
  const float* __restrict__ p = wg_partial + (int64_t)dir * nblk * GRU8X_WG_FLOATS + bx * NV + vi;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int r = slice;
  for (; r + 96 < nblk; r += 128) {
    a0 += p[(int64_t)r * GRU8X_WG_FLOATS];
    a1 += p[(int64_t)(r + 32) * GRU8X_WG_FLOATS];
    a2 += p[(int64_t)(r + 64) * GRU8X_WG_FLOATS];
    a3 += p[(int64_t)(r + 96) * GRU8X_WG_FLOATS];
  }
  for (; r < nblk; r += 32) a0 += p[(int64_t)r * GRU8X_WG_FLOATS];
  red[slice][vi] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x >= NV) return;
  float val = 0.0f;
#pragma unroll
  for (int k = 0; k < 32; ++k) val += red[k][vi];
  const int v = bx * NV + vi;
  float* g_wih = A.g[dir ? 4 : 0];
  float* g_whh = A.g[dir ? 5 : 1];
  float* g_bih = A.g[dir ? 6 : 2];
  float* g_bhh = A.g[dir ? 7 : 3];
  auto put = [&](float* d) { *d = accumulate ? *d + val : val; };
  if (v < 24 * 32) put(&g_wih[v]);
  else if (v < 24 * 32 + 24 * 8) put(&g_whh[v - 24 * 32]);
  else {
    const int kind = (v - (24 * 32 + 24 * 8)) >> 3, unit = v & 7;   // g_r, g_z, g_n (input side), g_h (hidden side)
    if (kind < 2) { put(&g_bih[kind * 8 + unit]); put(&g_bhh[kind * 8 + unit]); }
    else if (kind == 2) put(&g_bih[16 + unit]);
    else put(&g_bhh[16 + unit]);
  }
}
__global__ void __launch_bounds__(256) k_gru8x_wg_finalize(WgFinArgs A0, WgFinArgs A1, int accumulate) {
  __shared__ float red[32][kWgVals + 1];
  gru8x_wg_finalize_body(blockIdx.z ? A1 : A0, (int)blockIdx.x, (int)blockIdx.y, accumulate, red);
}

// The end-of-step reductions of the latent-8 recurrent step in ONE launch (round 5): the first layer's weight-gradient tiles
// (two encoder streams + the decoder's second layer), the second layer's (two streams) and the plain partial sums
// (LayerNorm weights / biases, mixture parameters) were three launches of 7 - 16 us one after another, each a few hundred
// workgroups of latency-bound strided sums; side by side they take what the longest takes.  Block ranges:
// [0, n16) k_gru16_wg_finalize's blocks (x fastest, then direction, then layer), [n16, n16 + n8) k_gru8x_wg_finalize's,
// the rest k_sum_partials_multi's (one output value each).  Same arithmetic and summation order as the three kernels.
struct StepFinArgs {
  WgFinArgs a16[3];
  WgFinArgs a8[2];
  DofSumJobs sums;
  int n16, n8;   // block counts of the first two ranges
};
__global__ void __launch_bounds__(256) k_step_finalize(StepFinArgs A, int accumulate) {
  __shared__ float red[32][kWgVals + 1];
  __shared__ float red1[256];
  constexpr int NX16 = GRU16_WG_FLOATS / kWgVals, NX8 = GRU8X_WG_FLOATS / kWgVals;
  int b = (int)blockIdx.x;
  if (b < A.n16) {
    gru16_wg_finalize_body(A.a16[b / (2 * NX16)], b % NX16, (b / NX16) & 1, accumulate, red);
    return;
  }
  b -= A.n16;
  if (b < A.n8) {
    gru8x_wg_finalize_body(A.a8[b / (2 * NX8)], b % NX8, (b / NX8) & 1, accumulate, red);
    return;
  }
  dof_sum_partials_multi_body(A.sums, b - A.n8, accumulate, red1);
}

template <int C>
__global__ void __launch_bounds__(256) k_ln_fwd(const float* __restrict__ X, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ Y, int T,
                                                int64_t S, int64_t Sp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * S) return;
  const int t = (int)(i / S);
  const int64_t s = i - (int64_t)t * S;
  float x[C];
  dof_ld_row<C>(X + ACT(t, 0, C, Sp, s), x);
  float mean = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) mean += x[c];
  mean *= (1.0f / C);
  float var = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float d = x[c] - mean;
    var = fmaf(d, d, var);
  }
  const float rstd = rsqrtf(var * (1.0f / C) + 1e-3f);
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = fmaf((x[c] - mean) * rstd, dof_cw(gamma)[c], dof_cw(beta)[c]);
  dof_st_row<C>(Y + ACT(t, 0, C, Sp, s), x);
}

// The same LayerNorm with one 16-byte word per lane: the C / 4 lanes of a row are neighbours (a wavefront's load
// covers 1 KB of contiguous memory instead of 64 separate 128-byte lines), row sums by xor-shuffles.  C / 4 must be
// a power of two (latent 8: 4 or 8 lanes per row).
template <int C>
__global__ void __launch_bounds__(256) k_ln_fwd_w(const float* __restrict__ X, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ Y, int64_t S,
                                                  int64_t Sp) {
  constexpr int Q = C / 4;
  static_assert((Q & (Q - 1)) == 0 && Q <= 64, "lanes per row must be a power of two");
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * Q) return;  // whole rows leave together (Q divides the wavefront)
  const int t = blockIdx.y;
  const int64_t s = e / Q;
  const int c0 = (int)(e - s * Q) * 4;
  float x[4];
  dof_ld_row<4>(X + ACT(t, c0, C, Sp, s), x);
  float sum = (x[0] + x[1]) + (x[2] + x[3]);
#pragma unroll
  for (int m = 1; m < Q; m <<= 1) sum += __shfl_xor(sum, m);
  const float mean = sum * (1.0f / C);
  float var = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    x[c] -= mean;
    var = fmaf(x[c], x[c], var);
  }
#pragma unroll
  for (int m = 1; m < Q; m <<= 1) var += __shfl_xor(var, m);
  const float rstd = rsqrtf(var * (1.0f / C) + 1e-3f);
#pragma unroll
  for (int c = 0; c < 4; ++c) x[c] = fmaf(x[c] * rstd, dof_cw(gamma)[c0 + c], dof_cw(beta)[c0 + c]);
  dof_st_row<4>(Y + ACT(t, c0, C, Sp, s), x);
}

// dX = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dY*gamma; per-block partial dgamma/dbeta.
// PW = true: per-window tensors [c][s] (the encoder's final LayerNorm, T = 1).
template <int C, bool PW>
__global__ void __launch_bounds__(256) k_ln_bwd(const float* __restrict__ X, const float* __restrict__ dY1,
                                                const float* __restrict__ dY2, const float* __restrict__ gamma,
                                                float* __restrict__ dX, float* __restrict__ partial,  // [nblk][2C]
                                                int T, int64_t S, int64_t Sp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < (int64_t)T * S;
#define LNIDX(t, c, s) (PW ? ((int64_t)(c) * Sp + (s)) : ACT(t, c, C, Sp, s))
  float vals[2 * C];
#pragma unroll
  for (int c = 0; c < 2 * C; ++c) vals[c] = 0.0f;
  if (live) {
    const int t = (int)(i / S);
    const int64_t s = i - (int64_t)t * S;
    float x[C], dy[C];
    if (PW) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        x[c] = X[LNIDX(t, c, s)];
        dy[c] = dY1[LNIDX(t, c, s)];
        if (dY2) dy[c] += dY2[LNIDX(t, c, s)];
      }
    } else {
      dof_ld_row<C>(X + ACT(t, 0, C, Sp, s), x);
      dof_ld_row<C>(dY1 + ACT(t, 0, C, Sp, s), dy);
      if (dY2) {
        float d2[C];
        dof_ld_row<C>(dY2 + ACT(t, 0, C, Sp, s), d2);
#pragma unroll
        for (int c = 0; c < C; ++c) dy[c] += d2[c];
      }
    }
    float mean = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) mean += x[c];
    mean *= (1.0f / C);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      x[c] -= mean;
      var = fmaf(x[c], x[c], var);
    }
    const float rstd = rsqrtf(var * (1.0f / C) + 1e-3f);
    float mg = 0.0f, mgx = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      x[c] *= rstd;  // xhat
      const float g = dy[c] * dof_cw(gamma)[c];
      mg += g;
      mgx = fmaf(g, x[c], mgx);
      vals[c] = dy[c] * x[c];
      vals[C + c] = dy[c];
    }
    mg *= (1.0f / C);
    mgx *= (1.0f / C);
    float dxr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dxr[c] = rstd * (dy[c] * dof_cw(gamma)[c] - mg - x[c] * mgx);
    if (PW) {
#pragma unroll
      for (int c = 0; c < C; ++c) dX[LNIDX(t, c, s)] = dxr[c];
    } else {
      dof_st_row<C>(dX + ACT(t, 0, C, Sp, s), dxr);
    }
  }
  dof_block_colsum<2 * C>(vals, partial + (int64_t)blockIdx.x * 2 * C);
#undef LNIDX
}

// LayerNorm backward, one 16-byte word per lane (see k_ln_fwd_w): a workgroup still owns 256 consecutive (t, s) rows
// -- Q passes of 256 / Q rows -- so the [nblk][2C] partial layout of the dgamma / dbeta reduction is unchanged.
template <int C>
__global__ void __launch_bounds__(256) k_ln_bwd_w(const float* __restrict__ X, const float* __restrict__ dY1,
                                                  const float* __restrict__ dY2, const float* __restrict__ gamma,
                                                  float* __restrict__ dX, float* __restrict__ partial,  // [nblk][2C]
                                                  int T, int64_t S, int64_t Sp) {
  constexpr int Q = C / 4, RPP = 256 / Q;  // lanes per row, rows per pass
  static_assert((Q & (Q - 1)) == 0, "lanes per row must be a power of two");
  __shared__ float red[256][8];
  const int sub = threadIdx.x / Q, c0 = (threadIdx.x % Q) * 4;
  float gam[4], acc[8];
#pragma unroll
  for (int c = 0; c < 4; ++c) gam[c] = dof_cw(gamma)[c0 + c];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
  const int64_t n_rows = (int64_t)T * S;
  for (int pass = 0; pass < Q; ++pass) {
    const int64_t i = (int64_t)blockIdx.x * 256 + pass * RPP + sub;
    const bool live = i < n_rows;
    const int t = live ? (int)(i / S) : 0;
    const int64_t s = live ? i - (int64_t)t * S : 0;
    float x[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dy[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
      dof_ld_row<4>(X + ACT(t, c0, C, Sp, s), x);
      dof_ld_row<4>(dY1 + ACT(t, c0, C, Sp, s), dy);
      if (dY2) {
        float d2[4];
        dof_ld_row<4>(dY2 + ACT(t, c0, C, Sp, s), d2);
#pragma unroll
        for (int c = 0; c < 4; ++c) dy[c] += d2[c];
      }
    }
    float sum = (x[0] + x[1]) + (x[2] + x[3]);
#pragma unroll
    for (int m = 1; m < Q; m <<= 1) sum += __shfl_xor(sum, m);
    const float mean = sum * (1.0f / C);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      x[c] -= mean;
      var = fmaf(x[c], x[c], var);
    }
#pragma unroll
    for (int m = 1; m < Q; m <<= 1) var += __shfl_xor(var, m);
    const float rstd = rsqrtf(var * (1.0f / C) + 1e-3f);
    float mg = 0.0f, mgx = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      x[c] *= rstd;  // xhat
      const float g = dy[c] * gam[c];
      mg += g;
      mgx = fmaf(g, x[c], mgx);
      acc[c] = fmaf(dy[c], x[c], acc[c]);
      acc[4 + c] += dy[c];
    }
#pragma unroll
    for (int m = 1; m < Q; m <<= 1) {
      mg += __shfl_xor(mg, m);
      mgx += __shfl_xor(mgx, m);
    }
    mg *= (1.0f / C);
    mgx *= (1.0f / C);
    if (live) {
      float dxr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) dxr[c] = rstd * (dy[c] * gam[c] - mg - x[c] * mgx);
      dof_st_row<4>(dX + ACT(t, c0, C, Sp, s), dxr);
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) red[threadIdx.x][c] = acc[c];
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) {  // value v: [dgamma (C) | dbeta (C)]; channel ch is held by the lanes tid % Q == ch / 4
    const int which = threadIdx.x / C, ch = threadIdx.x - which * C;
    float sum = 0.0f;
    for (int k = ch >> 2; k < 256; k += Q) sum += red[k][which * 4 + (ch & 3)];
    partial[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = sum;
  }
}

// ---------------------------------------------------------------------------------------------
// Encoder block tail: gather the final hidden state of GRU2 ([fwd @ len-1, bwd @ 0]) + LayerNorm.
// ---------------------------------------------------------------------------------------------
// + the CensNet dot product of the row with the stream's weight vector (k_cens_dots' arithmetic, same order), so that the
// graph layer's first kernel has its operand without a launch of its own.  One launch serves both streams (blockIdx.y).
struct EncFinalArgs {
  const float* O2;
  const int* len;
  const float *gamma, *beta;
  float *HF, *Y;
  const float* cw;  // CensNet weight vector of this stream (2H) or null
  float* dots;      // [S] or null
  int64_t S, Sp;
};
// blockIdx.y == 2: the decoder's frame-validity mask and valid-frame counts of the same batch (they depend on the input
// windows only): one 64-lane group per window, lane = time step -- a launch of its own otherwise.
template <int C>  // C = 2*H
__global__ void __launch_bounds__(256) k_enc_final_fwd(EncFinalArgs A0, EncFinalArgs A1, int T, DofDecValid V) {
  if (blockIdx.y == 2) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float n = 0.0f;
    if (b < V.B) {
      for (int t = lane; t < V.T; t += 64) {
        const float* __restrict__ row = V.x + (b * V.T + t) * V.C3;
        bool any = false;
        for (int j = 0; j < V.C3; ++j) any |= (row[j] != 0.0f);
        V.valid[(int64_t)t * V.Bp + b] = any ? 1.0f : 0.0f;
        n += any ? 1.0f : 0.0f;
      }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
    if (b < V.B && lane == 0) V.len[b] = (int)n;
    return;
  }
  const EncFinalArgs& A = blockIdx.y ? A1 : A0;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, Sp = A.Sp;
  if (s >= A.S) return;
  const int n = A.len[s];
  float x[C];
  float mean = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int t = (c < C / 2) ? n - 1 : 0;
    x[c] = n > 0 ? A.O2[ACT(t, c, C, Sp, s)] : 0.0f;
    A.HF[(int64_t)c * Sp + s] = x[c];
    mean += x[c];
  }
  mean *= (1.0f / C);
  float var = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float d = x[c] - mean;
    var = fmaf(d, d, var);
  }
  const float rstd = rsqrtf(var * (1.0f / C) + 1e-3f);
  float dot = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float y = fmaf((x[c] - mean) * rstd, A.gamma[c], A.beta[c]);
    A.Y[(int64_t)c * Sp + s] = y;
    if (A.cw) dot = fmaf(y, A.cw[c], dot);
  }
  if (A.dots) A.dots[s] = dot;
  (void)T;
}

__global__ void __launch_bounds__(256) k_zero_f32(float* __restrict__ p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.0f;
}

__global__ void __launch_bounds__(256) k_zero_int(int* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the encoder convolution (Conv1d k = 5 "same", F = 3 / 1 input channels -> C1 = 2L channels, no
// bias) straight from the first GRU layer's two input gradients (round 4):
//   dW[o][f][k] = sum_{t,s} (c[t][s][o] > 0) (dXf + dXb)[t][s][o] * xs[t + k - 2][s][f]
// Before: k_relu_merge read c, dXf, dXb and wrote the merged gradient (4 x 46 MB per stream at C2), then the generic
// k_outer job read it back with scalar operand loads.  Here thread (channel quad, sequence) walks its sequence's T
// steps: three 16-byte loads per step (a wavefront = 1 KB of contiguous rows per load), the five taps' input rows in a
// sliding register window, 4 x 5F accumulators; nothing is written but the workgroup's partial tile in k_outer's
// layout (so k_outer_finalize's fixed-order sum applies unchanged).  Both streams in one launch (blockIdx.y).
// ---------------------------------------------------------------------------------------------
struct EncConvWgArgs {
  const float* act; const float* d0; const float* d1; const float* xs;
  int64_t S, Sp, part_off;
  int nblk, F;
};
template <int C1, int F>
__device__ __forceinline__ void enc_conv_wgrad_body(const EncConvWgArgs& A, int T, float* __restrict__ partials, float* red) {
  // channel quads (padded to a lane-group size QP: latent 6 = 3 quads in groups of 4, the fourth lane idles), sequence
  // slots per workgroup, values per channel
  constexpr int Q = C1 / 4, QP = Q <= 2 ? Q : Q <= 4 ? 4 : 8, SL = 256 / QP, NV = 5 * F;
  static_assert(C1 % 4 == 0 && Q <= 8, "channel quads");
  const int tid = threadIdx.x;
  const int c4 = (tid % QP) * 4, sl = (tid % QP) < Q ? tid / QP : SL;
  float acc[4][NV];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[c][v] = 0.0f;
  if (sl < SL) {
    for (int64_t s = (int64_t)blockIdx.x * SL + sl; s < A.S; s += (int64_t)A.nblk * SL) {
      float xw[5][F];   // xs at t - 2 .. t + 2
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int f = 0; f < F; ++f) xw[k][f] = (k >= 3 && k - 3 < T) ? A.xs[((int64_t)(k - 3) * A.Sp + s) * F + f] : 0.0f;
#pragma unroll 5
      for (int t = 0; t < T; ++t) {
        const int64_t row = (int64_t)t * A.Sp + s;
        float m[4], a[4], b[4];
        dof_ld_row<4>(A.act + row * C1 + c4, m);
        dof_ld_row<4>(A.d0 + row * C1 + c4, a);
        dof_ld_row<4>(A.d1 + row * C1 + c4, b);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int f = 0; f < F; ++f) xw[k][f] = xw[k + 1][f];
#pragma unroll
        for (int f = 0; f < F; ++f) xw[4][f] = t + 2 < T ? A.xs[((int64_t)(t + 2) * A.Sp + s) * F + f] : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float g = m[c] > 0.0f ? a[c] + b[c] : 0.0f;
#pragma unroll
          for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int f = 0; f < F; ++f) acc[c][k * F + f] = fmaf(g, xw[k][f], acc[c][k * F + f]);
        }
      }
    }
  }
  // the sequence slots of a wavefront (fixed butterfly over the lanes that share a channel quad), then the four
  // wavefronts in order
  const int wave = tid >> 6, lane = tid & 63;
  {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float x = acc[c][v];
#pragma unroll
        for (int d = QP; d < 64; d <<= 1) x += __shfl_xor(x, d);
        acc[c][v] = x;
      }
    if (lane < Q) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int v = 0; v < NV; ++v) red[(wave * C1 + c4 + c) * NV + v] = acc[c][v];
    }
    __syncthreads();
    float* p0 = partials + A.part_off + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
    for (int e = tid; e < C1 * NV; e += 256) {
      const int oo = e / NV, v = e - oo * NV;
      float sum = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += red[(k * C1 + oo) * NV + v];
      p0[oo * 65 + v] = sum;   // tile 0, column k * F + f (the packed five-tap tile of the k_outer job)
    }
  }
}
template <int C1>
__global__ void __launch_bounds__(256) k_enc_conv_wgrad(EncConvWgArgs A0, EncConvWgArgs A1, int T, float* __restrict__ partials) {
  __shared__ float red[4 * C1 * 15];
  const EncConvWgArgs& A = blockIdx.y ? A1 : A0;
  if ((int)blockIdx.x >= A.nblk) return;
  if (A.F == 3) enc_conv_wgrad_body<C1, 3>(A, T, partials, red);
  else enc_conv_wgrad_body<C1, 1>(A, T, partials, red);
}

// dc = (dXf + dXb) * (c > 0)   (ReLU mask of the encoder conv; in place into dXf)
__global__ void __launch_bounds__(256) k_relu_merge(const float* __restrict__ act, float* __restrict__ d0,
                                                    const float* __restrict__ d1, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    d0[i] = act[i] > 0.0f ? d0[i] + d1[i] : 0.0f;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
#define DOF_DISPATCH_L(L, CALL)                            \
  switch (L) {                                             \
    case 4: { constexpr int LL = 4; CALL; } break;         \
    case 5: { constexpr int LL = 5; CALL; } break;         \
    case 6: { constexpr int LL = 6; CALL; } break;         \
    case 7: { constexpr int LL = 7; CALL; } break;         \
    case 8: { constexpr int LL = 8; CALL; } break;         \
    case 9: { constexpr int LL = 9; CALL; } break;         \
    case 10: { constexpr int LL = 10; CALL; } break;       \
    case 12: { constexpr int LL = 12; CALL; } break;       \
    case 14: { constexpr int LL = 14; CALL; } break;       \
    case 16: { constexpr int LL = 16; CALL; } break;       \
    case 20: { constexpr int LL = 20; CALL; } break;       \
    case 24: { constexpr int LL = 24; CALL; } break;       \
    case 32: { constexpr int LL = 32; CALL; } break;       \
    default:                                               \
      dof_set_error("latent_dim %d not supported by this build (4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 32)", (int)(L)); \
      return DOF_ERR_UNSUPPORTED;                          \
  }

// the node (F = 3) and the edge (F = 1) stream's encoder convolution in one launch; returns 1 when launched, 0 when the
// shapes ask for the per-stream launcher (a window that does not fit the LDS staging), < 0 on error
int dof_launch_enc_conv_fwd_pair(int L, const int F[2], const float* const xin[2], const float* const w[2], float* const xs[2],
                                 float* const c[2], int* const len[2], int T, const int G[2], const int64_t S[2],
                                 const int64_t Sp[2], hipStream_t st) {
  if (F[0] != 3 || F[1] != 1) return 0;
  EncConvFwdArgs A[2];
  unsigned nwin = 0;
  for (int k = 0; k < 2; ++k) {
    if (!((int64_t)T * G[k] * F[k] <= ENC_CONV_LDS && G[k] <= 256 && S[k] % G[k] == 0)) return 0;
    A[k].xin = xin[k]; A[k].w = w[k]; A[k].xs = xs[k]; A[k].c = c[k]; A[k].len = len[k]; A[k].G = G[k]; A[k].S = S[k]; A[k].Sp = Sp[k];
    if ((unsigned)(S[k] / G[k]) > nwin) nwin = (unsigned)(S[k] / G[k]);
  }
  int64_t need = 0;
  for (int k = 0; k < 2; ++k) need = (int64_t)T * G[k] * F[k] > need ? (int64_t)T * G[k] * F[k] : need;
  if (need <= 2048) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd_lds_pair<2 * LL, 2048>), (nwin, 2), (256), st, A[0], A[1], T));
  } else if (need <= 6144) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd_lds_pair<2 * LL, 6144>), (nwin, 2), (256), st, A[0], A[1], T));
  } else {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd_lds_pair<2 * LL, ENC_CONV_LDS>), (nwin, 2), (256), st, A[0], A[1], T));
  }
  return dof_check_launch("k_enc_conv_fwd_lds_pair") == DOF_OK ? 1 : DOF_ERR_LAUNCH;
}
int dof_launch_enc_conv_fwd(int L, int F, const float* xin, const float* w, float* xs, float* c, int* len, int T,
                            int G, int64_t S, int64_t Sp, hipStream_t st) {
  const unsigned nb = dof_cdiv((int64_t)T * S, 256);
  if ((int64_t)T * G * F <= ENC_CONV_LDS && G <= 256 && S % G == 0) {  // one workgroup per window, window staged in LDS
    const unsigned nwin = (unsigned)(S / G);
    if (F == 3) {
      DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd_lds<2 * LL, 3>), (nwin), (256), st, xin, w, xs, c, len, T, G, S, Sp));
    } else if (F == 1) {
      DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd_lds<2 * LL, 1>), (nwin), (256), st, xin, w, xs, c, len, T, G, S, Sp));
    } else {
      dof_set_error("features per group %d not supported (3 or 1)", F);
      return DOF_ERR_UNSUPPORTED;
    }
    return dof_check_launch("k_enc_conv_fwd_lds");
  }
  DOF_LAUNCH(k_zero_int, (dof_cdiv(S, 256)), (256), st, len, S);  // (a memset node here faulted under hipGraph replay)
  if (F == 3) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd<2 * LL, 3>), (nb), (256), st, xin, w, xs, c, len, T, G, S, Sp));
  } else if (F == 1) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_conv_fwd<2 * LL, 1>), (nb), (256), st, xin, w, xs, c, len, T, G, S, Sp));
  } else {
    dof_set_error("features per group %d not supported (3 or 1)", F);
    return DOF_ERR_UNSUPPORTED;
  }
  return dof_check_launch("k_enc_conv_fwd");
}

// The (16, 16) layer of a launch with >= 8,192 sequences (the encoder streams) takes the matrix-pipe pair
// k_gru16x_fwd / k_gru16x_bwd (no saved gates); shorter launches (the decoder: one sequence per window) stay on the
// lane-per-unit pair, whose 16x more wavefronts fill the chip there (measured at 1,024 sequences: 16 + 21 us against
// 20 + 52 us).  Forward and backward launchers take the same decision from the same S.
// DOF_GRU_MFMA_MIN_S overrides the threshold (0: always -- how the parity tests run the goldens through these kernels).
bool dof_gru16_mfma(int64_t S, int T) {
  static int64_t min_s = -1;
  if (min_s < 0) {
    const char* e = getenv("DOF_GRU_MFMA_MIN_S");
    min_s = e ? atoll(e) : 8192;
  }
  // the kernels address a [T][Sp][<= 32] tensor by 32-bit byte offsets (DofRowWalk): larger launches stay on the lane-per-unit pair
  if (T > 0 && (int64_t)T * dof_pad64(S) * 32 * 4 >= ((int64_t)1 << 31)) return false;
  return S >= min_s;
}

// the GEMM-shaped recurrence of the wider layers (k_grumx_fwd / k_grum_bwd): same size rule
bool dof_grum_selected(int64_t S) {
  // (the alternative here is the quad-split kernel, 290 us for the decoder's 1,024 sequences of a (32, 32) layer: the
  // matrix-pipe form wins from a few hundred sequences on, unlike the (16, 16) layer whose alternative is lane-per-unit)
  return dof_gru16_mfma(S, 0) || S >= 512;
}

static Gru16mStream gru16m_stream(const float* X, const int* len, DofGruW W, float* O, float* GS, const float* dO, float* dX,
                                  float* wg_partial, int64_t S, int64_t Sp) {
  Gru16mStream a;
  a.X = X; a.len = len;
  a.wih0 = W.wih0; a.whh0 = W.whh0; a.bih0 = W.bih0; a.bhh0 = W.bhh0;
  a.wih1 = W.wih1; a.whh1 = W.whh1; a.bih1 = W.bih1; a.bhh1 = W.bhh1;
  a.O = O; a.GS = GS; a.dO = dO; a.dX = dX; a.wg_partial = wg_partial; a.S = S; a.Sp = Sp;
  return a;
}

// The encoder's two streams in one launch (see Gru16mStream).  Returns 1 when the pair was launched, 0 when the caller
// has to launch the layers one by one (the matrix-pipe kernels are not selected for these sizes), < 0 on error.
int dof_launch_gru16_fwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], float* const O[2], int T,
                              const int64_t S[2], const int64_t Sp[2], hipStream_t st) {
  if (!(dof_gru16_mfma(S[0], T) && dof_gru16_mfma(S[1], T))) return 0;
  const Gru16mStream a = gru16m_stream(X[0], len[0], W[0], O[0], nullptr, nullptr, nullptr, nullptr, S[0], Sp[0]);
  const Gru16mStream b = gru16m_stream(X[1], len[1], W[1], O[1], nullptr, nullptr, nullptr, nullptr, S[1], Sp[1]);
  const int64_t smax = S[0] > S[1] ? S[0] : S[1];
  DOF_LAUNCH((k_gru16x_fwd<3, false>), (dof_cdiv(smax, 16), 2, 2), (64), st, a, b, T);
  return dof_check_launch("k_gru16x_fwd (pair)") == DOF_OK ? 1 : DOF_ERR_LAUNCH;
}

int dof_launch_gru16_bwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], const float* const O[2],
                              const float* const dO[2], float* const dX[2], float* const wg_partial[2], int T,
                              const int64_t S[2], const int64_t Sp[2], hipStream_t st) {
  if (!(dof_gru16_mfma(S[0], T) && dof_gru16_mfma(S[1], T))) return 0;
  const Gru16mStream a = gru16m_stream(X[0], len[0], W[0], const_cast<float*>(O[0]), nullptr, dO[0], dX[0], wg_partial[0], S[0], Sp[0]);
  const Gru16mStream b = gru16m_stream(X[1], len[1], W[1], const_cast<float*>(O[1]), nullptr, dO[1], dX[1], wg_partial[1], S[1], Sp[1]);
  const int64_t smax = S[0] > S[1] ? S[0] : S[1];
  if (dO[0] && dO[1]) DOF_LAUNCH((k_gru16x_bwd<true>), (dof_cdiv(smax, 64), 2, 2), (256), st, a, b, T);
  else if (!dO[0] && !dO[1]) DOF_LAUNCH((k_gru16x_bwd<false>), (dof_cdiv(smax, 64), 2, 2), (256), st, a, b, T);
  else { dof_set_error("dof_launch_gru16_bwd_pair: the two streams must both have (or both lack) an output gradient"); return DOF_ERR_ARG; }
  return dof_check_launch("k_gru16x_bwd (pair)") == DOF_OK ? 1 : DOF_ERR_LAUNCH;
}

// kind: 0 = (IN=2L,HID=2L) enc gru1 / dec gru2 ; 1 = (IN=4L,HID=L) enc gru2 ; 2 = (IN=L,HID=L, broadcast input) dec gru1
#define GRU3_W W.wih0, W.whh0, W.bih0, W.bhh0, W.wih1, W.whh1, W.bih1, W.bhh1
static Gru8Args gru3_fwd_args(const float* X, const int* len, const DofGruW& W, float* O, float* GS, int64_t S, int64_t Sp) {
  Gru8Args A = {};
  A.X = X; A.len = len; A.wih0 = W.wih0; A.whh0 = W.whh0; A.bih0 = W.bih0; A.bhh0 = W.bhh0; A.wih1 = W.wih1; A.whh1 = W.whh1;
  A.bih1 = W.bih1; A.bhh1 = W.bhh1; A.Oout = O; A.GSout = GS; A.S = S; A.Sp = Sp;
  return A;
}
// the same layer on the matrix pipe (k_gru8m_fwd; launches of >= DOF_GRU_MFMA_MIN_S sequences per stream): returns 1 when
// launched, 0 when the caller has to use the lane-per-unit kernels, < 0 on error.  DOF_GRU8_MFMA=0: off (A/B).
bool dof_gru8m_fwd_selected(int64_t S0, int64_t S1, int T) { return dof_gru16_mfma(S0, T) && dof_gru16_mfma(S1, T); }
int dof_launch_gru8m_fwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], float* const O[2],
                              float* const GS[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st) {
  if (!dof_gru8m_fwd_selected(S[0], S[1], T)) return 0;
  const Gru16mStream a = gru16m_stream(X[0], len[0], W[0], O[0], GS[0], nullptr, nullptr, nullptr, S[0], Sp[0]);
  const Gru16mStream b = gru16m_stream(X[1], len[1], W[1], O[1], GS[1], nullptr, nullptr, nullptr, S[1], Sp[1]);
  const int64_t smax = S[0] > S[1] ? S[0] : S[1];
  DOF_LAUNCH((k_gru8x_fwd<4>), (dof_cdiv(smax, 16), 2, 2), (64), st, a, b, T);   // (saves no gates: k_gru8x_bwd recomputes them)
  return dof_check_launch("k_gru8x_fwd (pair)") == DOF_OK ? 1 : DOF_ERR_LAUNCH;
}
// the second encoder layer (32 -> 8, latent 8) of both streams in one launch
int dof_launch_gru8_fwd_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], float* const O[2],
                             float* const GS[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st) {
  const Gru8Args A0 = gru3_fwd_args(X[0], len[0], W[0], O[0], GS[0], S[0], Sp[0]);
  const Gru8Args A1 = gru3_fwd_args(X[1], len[1], W[1], O[1], GS[1], S[1], Sp[1]);
  DOF_LAUNCH((k_gru3_fwd<32, 8, false>), (dof_cdiv(S[0] > S[1] ? S[0] : S[1], 32), 2, 2), (256), st, A0, A1, T);
  return dof_check_launch("k_gru3_fwd");
}
// latent sizes whose GRU layers (hidden 2 L and L <= 16, not 8 / 16 themselves) run the lane-per-unit kernels on padded lane
// groups; their saved gates / gate gradients are unit-major like latent 8's (the weight-gradient jobs' `gate_minor`)
// (kind: 0 = GRU(2L -> 2L), 1 = GRU(4L -> L), 2 = GRU(L -> L) with a time-constant input.)  Latent 10: the two layers of
// hidden size 10; its GRU(20 -> 20) layers take the quad-split kernels (gate-major buffers).
bool dof_gru_lane_per_unit(int L, int kind) {
  const int hid = kind == 0 ? 2 * L : L, in = kind == 0 ? 2 * L : kind == 1 ? 4 * L : L;
  // (input widths above 48 take the weight-gradient jobs' row-block form, which reads gate-major buffers only: 14's
  //  GRU(56 -> 14) layers stay on the thread-per-sequence kernels)
  return (L == 4 || L == 5 || L == 6 || L == 7 || L == 9 || L == 10 || L == 14) && hid <= 16 && in <= 48;
}
// ... and the GRU(2L -> 2L) layers of 10 and 14 the quad-split kernels (hidden 20 / 28: a multiple of 4); 9's (hidden 18) stay on
// the thread-per-sequence kernels
static bool gru_quad_layer(int L, int kind) { return (L >= 12 && L % 4 == 0) || ((L == 10 || L == 14) && kind == 0); }
int dof_launch_gru_fwd(int L, int kind, const float* X, const int* len, DofGruW W, float* O, float* GS, int T,
                       int64_t S, int64_t Sp, hipStream_t st) {
  if (L == 8) {  // weight-stationary kernels: matrix-pipe recurrence (kind 0) / lane per unit
    if (kind == 0 && dof_gru16_mfma(S, T)) {  // (no gates saved: k_gru16x_bwd recomputes them)
      const Gru16mStream a = gru16m_stream(X, len, W, O, nullptr, nullptr, nullptr, nullptr, S, Sp);
      DOF_LAUNCH((k_gru16x_fwd<3, false>), (dof_cdiv(S, 16), 2, 1), (64), st, a, a, T);
    }
    const Gru8Args A = gru3_fwd_args(X, len, W, O, GS, S, Sp);
    if (kind == 0 && dof_gru16_mfma(S, T)) {}
    else if (kind == 0) DOF_LAUNCH((k_gru3_fwd<16, 16, false>), (dof_cdiv(S, 16), 2, 1), (256), st, A, A, T);
    else if (kind == 1) DOF_LAUNCH((k_gru3_fwd<32, 8, false>), (dof_cdiv(S, 32), 2, 1), (256), st, A, A, T);
    else DOF_LAUNCH((k_gru3_fwd<8, 8, true>), (dof_cdiv(S, 32), 2, 1), (256), st, A, A, T);
    return dof_check_launch("k_gru3_fwd");
  }
  if (dof_gru_lane_per_unit(L, kind)) {  // latent 4 / 5 / 6 / 10 (round 6): the lane-per-unit kernels on padded lane groups
    const Gru8Args A = gru3_fwd_args(X, len, W, O, GS, S, Sp);
#define GRU3_FWD(IN_, HID_, BC_) DOF_LAUNCH((k_gru3_fwd<IN_, HID_, BC_>), (dof_cdiv(S, 256 / ((HID_) <= 8 ? 8 : 16)), 2, 1), (256), st, A, A, T)
#define GRU3_FWD_L(LL_) \
  case LL_: \
    if (kind == 0) GRU3_FWD(2 * LL_, 2 * LL_, false); \
    else if (kind == 1) GRU3_FWD(4 * LL_, LL_, false); \
    else GRU3_FWD(LL_, LL_, true); \
    break
    switch (L) {
      GRU3_FWD_L(4); GRU3_FWD_L(5); GRU3_FWD_L(6); GRU3_FWD_L(7);
#define GRU3_FWD_H(LL_) case LL_: if (kind == 1) GRU3_FWD(4 * LL_, LL_, false); else GRU3_FWD(LL_, LL_, true); break
      GRU3_FWD_H(9); GRU3_FWD_H(10);
      case 14: GRU3_FWD(14, 14, true); break;
#undef GRU3_FWD_H
      default: return DOF_ERR_UNSUPPORTED;
    }
#undef GRU3_FWD_L
#undef GRU3_FWD
    return dof_check_launch("k_gru3_fwd");
  }
  if ((L == 16 || L == 32) && kind != 2 && dof_grum_selected(S)) {  // encoder-sized launches: the GEMM-shaped recurrence
    const unsigned nbm = dof_cdiv(S, 128);   // four wavefronts x two tiles of 16 sequences
#define GRUM_FWD(IN_, HID_) DOF_LAUNCH((k_grumx_fwd<IN_, HID_>), (nbm, 2), (256), st, X, len, W.wih0, W.whh0, W.bih0, W.bhh0, W.wih1, W.whh1, W.bih1, W.bhh1, O, GS, T, S, Sp)
    if (L == 16 && kind == 0) GRUM_FWD(32, 32);
    else if (L == 16) GRUM_FWD(64, 16);
    else if (kind == 0) GRUM_FWD(64, 64);
    else GRUM_FWD(128, 32);
#undef GRUM_FWD
    return dof_check_launch("k_grumx_fwd");
  }
  if (gru_quad_layer(L, kind)) {  // a sequence across four lanes (at latent 32 thread-per-sequence took 134 ms per C2-shape step)
    const unsigned nq = dof_cdiv(S * 4, 256);
#define GRUQ_FWD(IN_, HID_, BC_) DOF_LAUNCH((k_gruq_fwd<IN_, HID_, BC_>), (nq, 2), (256), st, X, len, W.wih0, W.whh0, W.bih0, W.bhh0, W.wih1, W.whh1, W.bih1, W.bhh1, O, GS, T, S, Sp)
#define GRUQ_FWD_L(LL_) \
  case LL_: \
    if (kind == 0) GRUQ_FWD(2 * LL_, 2 * LL_, false); \
    else if (kind == 1) GRUQ_FWD(4 * LL_, LL_, false); \
    else GRUQ_FWD(LL_, LL_, true); \
    break
    switch (L) {
      GRUQ_FWD_L(12); GRUQ_FWD_L(16); GRUQ_FWD_L(20); GRUQ_FWD_L(24); GRUQ_FWD_L(32);
      case 10: GRUQ_FWD(20, 20, false); break;
      case 14: GRUQ_FWD(28, 28, false); break;
      default: dof_set_error("GRU: latent_dim %d has no quad-split kernel", L); return DOF_ERR_UNSUPPORTED;
    }
#undef GRUQ_FWD_L
#undef GRUQ_FWD
    return dof_check_launch("k_gruq_fwd");
  }
  const unsigned nb = dof_cdiv(S, 256);
  if (kind == 0) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_gru_fwd<2 * LL, 2 * LL, false>), (nb, 2), (256), st, X, len, W.wih0, W.whh0, W.bih0, W.bhh0, W.wih1, W.whh1, W.bih1, W.bhh1, O, GS, T, S, Sp));
  } else if (kind == 1) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_gru_fwd<4 * LL, LL, false>), (nb, 2), (256), st, X, len, W.wih0, W.whh0, W.bih0, W.bhh0, W.wih1, W.whh1, W.bih1, W.bhh1, O, GS, T, S, Sp));
  } else {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_gru_fwd<LL, LL, true>), (nb, 2), (256), st, X, len, W.wih0, W.whh0, W.bih0, W.bhh0, W.wih1, W.whh1, W.bih1, W.bhh1, O, GS, T, S, Sp));
  }
  return dof_check_launch("k_gru_fwd");
}

// Workgroups per direction of the lane-per-unit backward kernel when it also produces the layer's weight gradients (one
// partial tile each; 0: the layer's gradients are a k_outer job).  DOF_GRU_WGRAD_FUSED=0 keeps the job.
int dof_gru3_wg_blocks(int L, int kind, int64_t S) {
  static const int on = [] {
    const char* e = getenv("DOF_GRU_WGRAD_FUSED");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  if (!on || L == 8 || kind == 2 || !dof_gru_lane_per_unit(L, kind)) return 0;
  const int hid = kind == 0 ? 2 * L : L;
  return (int)dof_cdiv(S, 256 / (hid <= 8 ? 8 : 16));
}
int dof_launch_gru_bwd(int L, int kind, const int* len, DofGruW W, const float* O, float* GS, const float* dO,
                       const float* dHfin, float* dX, int T, int64_t S, int64_t Sp, hipStream_t st, const float* X,
                       float* wg_part0, float* wg_part1) {
  if (L == 8) {
    if (kind == 0) DOF_LAUNCH((k_gru3_bwd<16, 16, false>), (dof_cdiv(S, 16), 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    else if (kind == 1) DOF_LAUNCH((k_gru3_bwd<32, 8, false>), (dof_cdiv(S, 32), 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    else DOF_LAUNCH((k_gru3_bwd<8, 8, true>), (dof_cdiv(S, 32), 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return dof_check_launch("k_gru3_bwd");
  }
  if (dof_gru_lane_per_unit(L, kind)) {
    const bool wg = wg_part0 != nullptr;
    if (wg && (!X || !wg_part1 || dof_gru3_wg_blocks(L, kind, S) == 0)) {
      dof_set_error("GRU backward: fused weight gradient asked for a layer that has none (latent %d, kind %d)", L, kind);
      return DOF_ERR_ARG;
    }
#define GRU3_BWD(IN_, HID_, BC_) DOF_LAUNCH((k_gru3_bwd<IN_, HID_, BC_>), (dof_cdiv(S, 256 / ((HID_) <= 8 ? 8 : 16)), 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp, (const float*)nullptr, (float*)nullptr, (float*)nullptr)
#define GRU3_BWD_W(IN_, HID_) \
  do { \
    if (wg) DOF_LAUNCH((k_gru3_bwd<IN_, HID_, false, true>), (dof_cdiv(S, 256 / ((HID_) <= 8 ? 8 : 16)), 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp, X, wg_part0, wg_part1); \
    else GRU3_BWD(IN_, HID_, false); \
  } while (0)
#define GRU3_BWD_L(LL_) \
  case LL_: \
    if (kind == 0) GRU3_BWD_W(2 * LL_, 2 * LL_); \
    else if (kind == 1) GRU3_BWD_W(4 * LL_, LL_); \
    else GRU3_BWD(LL_, LL_, true); \
    break
    switch (L) {
      GRU3_BWD_L(4); GRU3_BWD_L(5); GRU3_BWD_L(6); GRU3_BWD_L(7);
#define GRU3_BWD_H(LL_) case LL_: if (kind == 1) GRU3_BWD_W(4 * LL_, LL_); else GRU3_BWD(LL_, LL_, true); break
      GRU3_BWD_H(9); GRU3_BWD_H(10);
      case 14: GRU3_BWD(14, 14, true); break;
#undef GRU3_BWD_H
      default: return DOF_ERR_UNSUPPORTED;
    }
#undef GRU3_BWD_L
#undef GRU3_BWD_W
#undef GRU3_BWD
    return dof_check_launch("k_gru3_bwd");
  }
  if ((L == 16 || L == 32) && kind != 2 && dof_grum_selected(S)) {
    const unsigned nbm = dof_cdiv(S, 64);
#define GRUM_BWD(IN_, HID_) DOF_LAUNCH((k_grum_bwd<IN_, HID_>), (nbm, 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp)
    if (L == 16 && kind == 0) GRUM_BWD(32, 32);
    else if (L == 16) GRUM_BWD(64, 16);
    else if (kind == 0) GRUM_BWD(64, 64);
    else GRUM_BWD(128, 32);
#undef GRUM_BWD
    return dof_check_launch("k_grum_bwd");
  }
  if (gru_quad_layer(L, kind)) {
    const unsigned nq = dof_cdiv(S * 4, 256);
#define GRUQ_BWD(IN_, HID_, BC_) DOF_LAUNCH((k_gruq_bwd<IN_, HID_, BC_>), (nq, 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp)
#define GRUQ_BWD_L(LL_) \
  case LL_: \
    if (kind == 0) GRUQ_BWD(2 * LL_, 2 * LL_, false); \
    else if (kind == 1) GRUQ_BWD(4 * LL_, LL_, false); \
    else GRUQ_BWD(LL_, LL_, true); \
    break
    switch (L) {
      GRUQ_BWD_L(12); GRUQ_BWD_L(16); GRUQ_BWD_L(20); GRUQ_BWD_L(24); GRUQ_BWD_L(32);
      case 10: GRUQ_BWD(20, 20, false); break;
      case 14: GRUQ_BWD(28, 28, false); break;
      default: dof_set_error("GRU: latent_dim %d has no quad-split kernel", L); return DOF_ERR_UNSUPPORTED;
    }
#undef GRUQ_BWD_L
#undef GRUQ_BWD
    return dof_check_launch("k_gruq_bwd");
  }
  const unsigned nb = dof_cdiv(S, 256);
  if (kind == 0) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_gru_bwd<2 * LL, 2 * LL, false>), (nb, 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp));
  } else if (kind == 1) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_gru_bwd<4 * LL, LL, false>), (nb, 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp));
  } else {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_gru_bwd<LL, LL, true>), (nb, 2), (256), st, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dHfin, dX, T, S, Sp));
  }
  return dof_check_launch("k_gru_bwd");
}

// mult: channels = mult * L  (2 or 4)
int dof_launch_ln_fwd(int L, int mult, const float* X, const float* gamma, const float* beta, float* Y, int T,
                      int64_t S, int64_t Sp, hipStream_t st) {
  const unsigned nb = dof_cdiv((int64_t)T * S, 256);
  if (L == 8) {  // word-per-lane form (lanes per row a power of two)
    if (mult == 2) DOF_LAUNCH((k_ln_fwd_w<16>), (dof_cdiv(S * 4, 256), (unsigned)T), (256), st, X, gamma, beta, Y, S, Sp);
    else DOF_LAUNCH((k_ln_fwd_w<32>), (dof_cdiv(S * 8, 256), (unsigned)T), (256), st, X, gamma, beta, Y, S, Sp);
    return dof_check_launch("k_ln_fwd_w");
  }
  if (mult == 2) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_ln_fwd<2 * LL>), (nb), (256), st, X, gamma, beta, Y, T, S, Sp));
  } else {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_ln_fwd<4 * LL>), (nb), (256), st, X, gamma, beta, Y, T, S, Sp));
  }
  return dof_check_launch("k_ln_fwd");
}

int64_t dof_gru16_wg_floats(int64_t S) { return 2 * (int64_t)dof_cdiv(S, 16) * GRU16_WG_FLOATS; }

int dof_launch_gru16_bwd_fused(const float* X, const int* len, DofGruW W, const float* O, const float* GS,
                               const float* dO, float* dX, float* wg_partial, int T, int64_t S, int64_t Sp,
                               hipStream_t st) {
  if (dof_gru16_mfma(S, T)) {  // matrix-pipe recurrence, gates recomputed (GS is not read: k_gru16x_fwd saved none)
    const Gru16mStream a = gru16m_stream(X, len, W, const_cast<float*>(O), nullptr, dO, dX, wg_partial, S, Sp);
    if (dO) DOF_LAUNCH((k_gru16x_bwd<true>), (dof_cdiv(S, 64), 2, 1), (256), st, a, a, T);
    else DOF_LAUNCH((k_gru16x_bwd<false>), (dof_cdiv(S, 64), 2, 1), (256), st, a, a, T);
    return dof_check_launch("k_gru16x_bwd");
  }
  DOF_LAUNCH(k_gru16_bwd_fused, (dof_cdiv(S, 16), 2), (256), st, X, len, W.wih0, W.whh0, W.wih1, W.whh1, O, GS, dO, dX,
             wg_partial, T, S, Sp);
  return dof_check_launch("k_gru16_bwd_fused");
}

// g: gradient buffer base; off[8]: offsets of (wih, whh, bih, bhh) x (fwd, reverse) as in the parameter layout
static WgFinArgs wg_fin_args(const float* wg_partial, int nblk, float* g, const int64_t* off) {
  WgFinArgs A;
  A.wg_partial = wg_partial;
  A.nblk = nblk;
  for (int k = 0; k < 8; ++k) A.g[k] = g + off[k];
  return A;
}
int dof_launch_gru16_wg_finalize(const float* wg_partial, int64_t S, float* g, const int64_t* off, int accumulate,
                                 hipStream_t st) {
  const WgFinArgs A = wg_fin_args(wg_partial, (int)dof_cdiv(S, 16), g, off);
  DOF_LAUNCH(k_gru16_wg_finalize, ((unsigned)(GRU16_WG_FLOATS / kWgVals), 2, 1), (256), st, A, A, A, accumulate);
  return dof_check_launch("k_gru16_wg_finalize");
}
// two or three layers (the node and the edge stream's [+ the decoder's]) in one launch
int dof_launch_gru16_wg_finalize_pair(const float* const* wg_partial, const int64_t* S, float* g, const int64_t* const* off,
                                      int accumulate, hipStream_t st, int n) {
  const WgFinArgs A0 = wg_fin_args(wg_partial[0], (int)dof_cdiv(S[0], 16), g, off[0]);
  const WgFinArgs A1 = wg_fin_args(wg_partial[1], (int)dof_cdiv(S[1], 16), g, off[1]);
  const WgFinArgs A2 = n > 2 ? wg_fin_args(wg_partial[2], (int)dof_cdiv(S[2], 16), g, off[2]) : A1;
  DOF_LAUNCH(k_gru16_wg_finalize, ((unsigned)(GRU16_WG_FLOATS / kWgVals), 2, (unsigned)n), (256), st, A0, A1, A2, accumulate);
  return dof_check_launch("k_gru16_wg_finalize");
}
int dof_launch_gru8_wg_finalize_pair(const float* const wg_partial[2], const int64_t S[2], float* g, const int64_t* const off[2],
                                     int accumulate, hipStream_t st, int T) {
  if (dof_gru8m_fwd_selected(S[0], S[1], T)) {   // k_gru8x_bwd's partials: one row per wavefront tile of 16 sequences
    const WgFinArgs A0 = wg_fin_args(wg_partial[0], (int)dof_cdiv(S[0], 16), g, off[0]);
    const WgFinArgs A1 = wg_fin_args(wg_partial[1], (int)dof_cdiv(S[1], 16), g, off[1]);
    DOF_LAUNCH(k_gru8x_wg_finalize, ((unsigned)(GRU8X_WG_FLOATS / 8), 2, 2), (256), st, A0, A1, accumulate);
    return dof_check_launch("k_gru8x_wg_finalize");
  }
  const WgFinArgs A0 = wg_fin_args(wg_partial[0], (int)dof_cdiv(S[0], 32), g, off[0]);
  const WgFinArgs A1 = wg_fin_args(wg_partial[1], (int)dof_cdiv(S[1], 32), g, off[1]);
  DOF_LAUNCH(k_gru8_wg_finalize, ((unsigned)(GRU8_WG_FLOATS / 8), 2, 2), (256), st, A0, A1, accumulate);
  return dof_check_launch("k_gru8_wg_finalize");
}

// k_step_finalize: n16 = 2 or 3 first-layer partial sets, the two second-layer sets of k_gru8x_bwd, the plain sums
bool dof_step_finalize_selected(const int64_t S8[2], int T) { return dof_gru8m_fwd_selected(S8[0], S8[1], T); }
int dof_launch_step_finalize(const float* const* wg16, const int64_t* S16, const int64_t* const* off16, int n16,
                             const float* const wg8[2], const int64_t S8[2], const int64_t* const off8[2],
                             const DofSumJobs& sums, float* g, int accumulate, hipStream_t st) {
  StepFinArgs A;
  for (int k = 0; k < 3; ++k) A.a16[k] = wg_fin_args(wg16[k < n16 ? k : 0], (int)dof_cdiv(S16[k < n16 ? k : 0], 16), g, off16[k < n16 ? k : 0]);
  for (int k = 0; k < 2; ++k) A.a8[k] = wg_fin_args(wg8[k], (int)dof_cdiv(S8[k], 16), g, off8[k]);
  A.sums = sums;
  A.n16 = n16 * 2 * (GRU16_WG_FLOATS / kWgVals);
  A.n8 = 2 * 2 * (GRU8X_WG_FLOATS / kWgVals);
  int total = 0;
  for (int j = 0; j < sums.n; ++j) total += sums.nv[j];
  DOF_LAUNCH(k_step_finalize, ((unsigned)(A.n16 + A.n8 + total)), (256), st, A, accumulate);
  return dof_check_launch("k_step_finalize");
}

static_assert(GRU8X_WG_FLOATS == GRU8_WG_FLOATS, "one partial-row size for both backward kernels of the layer");
// (sized for k_gru8x_bwd: one partial row per wavefront tile of 16 sequences; k_gru8_bwd_fused writes half as many)
int64_t dof_gru8_wg_floats(int64_t S) { return 2 * (int64_t)dof_cdiv(S, 16) * GRU8_WG_FLOATS; }

int dof_launch_gru8_bwd_fused(const float* X, const int* len, DofGruW W, const float* O, const float* GS,
                              const float* dHfin, float* dX, float* wg_partial, int T, int64_t S, int64_t Sp,
                              hipStream_t st) {
  Gru8Args A = {};
  A.X = X; A.len = len; A.wih0 = W.wih0; A.whh0 = W.whh0; A.wih1 = W.wih1; A.whh1 = W.whh1; A.O = O; A.GS = GS;
  A.dHfin = dHfin; A.dX = dX; A.wg_partial = wg_partial; A.S = S; A.Sp = Sp; A.nblk = (int)dof_cdiv(S, 32);
  DOF_LAUNCH(k_gru8_bwd_fused, (dof_cdiv(S, 32), 2, 1), (256), st, A, A, T);
  return dof_check_launch("k_gru8_bwd_fused");
}
// both encoder streams in one launch
int dof_launch_gru8_bwd_fused_pair(const float* const X[2], const int* const len[2], const DofGruW W[2], const float* const O[2],
                                   const float* const GS[2], const float* const dHfin[2], float* const dX[2],
                                   float* const wg_partial[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st) {
  if (dof_gru8m_fwd_selected(S[0], S[1], T)) {   // the forward pass saved no gates: recompute on the matrix pipe
    const Gru16mStream a = gru16m_stream(X[0], len[0], W[0], const_cast<float*>(O[0]), nullptr, dHfin[0], dX[0], wg_partial[0], S[0], Sp[0]);
    const Gru16mStream b = gru16m_stream(X[1], len[1], W[1], const_cast<float*>(O[1]), nullptr, dHfin[1], dX[1], wg_partial[1], S[1], Sp[1]);
    const int64_t smax = S[0] > S[1] ? S[0] : S[1];
    DOF_LAUNCH(k_gru8x_bwd, (dof_cdiv(smax, 16), 2, 2), (64), st, a, b, T);
    return dof_check_launch("k_gru8x_bwd (pair)");
  }
  Gru8Args A[2] = {};
  for (int k = 0; k < 2; ++k) {
    A[k].X = X[k]; A[k].len = len[k]; A[k].wih0 = W[k].wih0; A[k].whh0 = W[k].whh0; A[k].wih1 = W[k].wih1; A[k].whh1 = W[k].whh1;
    A[k].O = O[k]; A[k].GS = GS[k]; A[k].dHfin = dHfin[k]; A[k].dX = dX[k]; A[k].wg_partial = wg_partial[k];
    A[k].S = S[k]; A[k].Sp = Sp[k]; A[k].nblk = (int)dof_cdiv(S[k], 32);
  }
  const unsigned nb = dof_cdiv(S[0] > S[1] ? S[0] : S[1], 32);
  DOF_LAUNCH(k_gru8_bwd_fused, (nb, 2, 2), (256), st, A[0], A[1], T);
  return dof_check_launch("k_gru8_bwd_fused");
}

int dof_launch_gru8_wg_finalize(const float* wg_partial, int64_t S, float* g, const int64_t* off, int accumulate,
                                hipStream_t st) {
  const WgFinArgs A = wg_fin_args(wg_partial, (int)dof_cdiv(S, 32), g, off);
  DOF_LAUNCH(k_gru8_wg_finalize, ((unsigned)(GRU8_WG_FLOATS / 8), 2, 1), (256), st, A, A, accumulate);
  return dof_check_launch("k_gru8_wg_finalize");
}

int64_t dof_ln_bwd_blocks(int T, int64_t S) { return dof_cdiv((int64_t)T * S, 256); }

int dof_launch_ln_bwd(int L, int mult, const float* X, const float* dY1, const float* dY2, const float* gamma,
                      float* dX, float* partial, int T, int64_t S, int64_t Sp, hipStream_t st) {
  const unsigned nb = (unsigned)dof_ln_bwd_blocks(T, S);
  if (L == 8 && T > 1) {  // word-per-lane form
    if (mult == 2) DOF_LAUNCH((k_ln_bwd_w<16>), (nb), (256), st, X, dY1, dY2, gamma, dX, partial, T, S, Sp);
    else DOF_LAUNCH((k_ln_bwd_w<32>), (nb), (256), st, X, dY1, dY2, gamma, dX, partial, T, S, Sp);
    return dof_check_launch("k_ln_bwd_w");
  }
  if (mult == 2 && T == 1) {  // per-window [c][s] tensors (encoder block output)
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_ln_bwd<2 * LL, true>), (nb), (256), st, X, dY1, dY2, gamma, dX, partial, T, S, Sp));
  } else if (mult == 2) {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_ln_bwd<2 * LL, false>), (nb), (256), st, X, dY1, dY2, gamma, dX, partial, T, S, Sp));
  } else {
    DOF_DISPATCH_L(L, DOF_LAUNCH((k_ln_bwd<4 * LL, false>), (nb), (256), st, X, dY1, dY2, gamma, dX, partial, T, S, Sp));
  }
  return dof_check_launch("k_ln_bwd");
}

// both encoder streams' tails (final GRU2 state -> LayerNorm [-> CensNet dot product]) in one launch
int dof_launch_enc_final_fwd_pair(int L, const float* const O2[2], const int* const len[2], const float* const gamma[2],
                                  const float* const beta[2], float* const HF[2], float* const Y[2], const float* const cw[2],
                                  float* const dots[2], int T, const int64_t S[2], const int64_t Sp[2], hipStream_t st,
                                  const DofDecValid* dec) {
  EncFinalArgs A[2];
  for (int k = 0; k < 2; ++k) {
    A[k].O2 = O2[k]; A[k].len = len[k]; A[k].gamma = gamma[k]; A[k].beta = beta[k]; A[k].HF = HF[k]; A[k].Y = Y[k];
    A[k].cw = cw[k]; A[k].dots = dots[k]; A[k].S = S[k]; A[k].Sp = Sp[k];
  }
  unsigned nb = dof_cdiv(S[0] > S[1] ? S[0] : S[1], 256);
  DofDecValid V = {};
  if (dec && dec->x) {
    V = *dec;
    const unsigned nv = dof_cdiv(V.B, 4);
    if (nv > nb) nb = nv;
  }
  DOF_DISPATCH_L(L, DOF_LAUNCH((k_enc_final_fwd<2 * LL>), (nb, V.x ? 3 : 2), (256), st, A[0], A[1], T, V));
  return dof_check_launch("k_enc_final_fwd");
}

// zero-fill as a kernel (graph-replay safe; memset nodes on sub-buffers misbehaved under hipGraph replay)
int dof_launch_zero(float* p, int64_t n, hipStream_t st) {
  unsigned nb = dof_cdiv(n, 256);
  if (nb > 2048) nb = 2048;
  DOF_LAUNCH(k_zero_f32, (nb), (256), st, p, n);
  return dof_check_launch("k_zero_f32");
}

// both encoder streams' convolution weight gradients (C1 = 8, 16 or 32 channels: latent 4, 8, 16), partial tiles of
// nblk[s] workgroups each at partials + part_off[s]
int dof_enc_conv_wgrad_blocks(int C1, int64_t S) {
  if (C1 != 8 && C1 != 12 && C1 != 16 && C1 != 32) return 0;
  const int SL = C1 == 12 ? 64 : 256 / (C1 / 4);   // (latent 6: 12 channels = 3 quads in lane groups of 4)
  const int64_t nb = (S + SL - 1) / SL;
  return (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
}
int dof_launch_enc_conv_wgrad(int C1, const float* const act[2], const float* const dX[2], const float* const xs[2], const int F[2],
                              int T, const int64_t S[2], const int64_t Sp[2], const int64_t part_off[2], float* partials,
                              hipStream_t st) {
  EncConvWgArgs A[2];
  unsigned nb = 0;
  for (int k = 0; k < 2; ++k) {
    A[k].act = act[k]; A[k].d0 = dX[k]; A[k].d1 = dX[k] + (int64_t)T * C1 * Sp[k]; A[k].xs = xs[k];
    A[k].S = S[k]; A[k].Sp = Sp[k]; A[k].part_off = part_off[k]; A[k].nblk = dof_enc_conv_wgrad_blocks(C1, S[k]); A[k].F = F[k];
    if (F[k] != 1 && F[k] != 3) { dof_set_error("encoder conv weight gradient: %d input channels", F[k]); return DOF_ERR_UNSUPPORTED; }
    if ((unsigned)A[k].nblk > nb) nb = (unsigned)A[k].nblk;
  }
  if (C1 == 8) DOF_LAUNCH(k_enc_conv_wgrad<8>, (nb, 2), (256), st, A[0], A[1], T, partials);
  else if (C1 == 12) DOF_LAUNCH(k_enc_conv_wgrad<12>, (nb, 2), (256), st, A[0], A[1], T, partials);
  else if (C1 == 16) DOF_LAUNCH(k_enc_conv_wgrad<16>, (nb, 2), (256), st, A[0], A[1], T, partials);
  else if (C1 == 32) DOF_LAUNCH(k_enc_conv_wgrad<32>, (nb, 2), (256), st, A[0], A[1], T, partials);
  else { dof_set_error("encoder conv weight gradient: %d channels", C1); return DOF_ERR_UNSUPPORTED; }
  return dof_check_launch("k_enc_conv_wgrad");
}
int dof_launch_relu_merge(const float* act, float* d0, const float* d1, int64_t n, hipStream_t st) {
  unsigned nb = dof_cdiv(n, 256);
  if (nb > 4096) nb = 4096;
  DOF_LAUNCH(k_relu_merge, (nb), (256), st, act, d0, d1, n);
  return dof_check_launch("k_relu_merge");
}
