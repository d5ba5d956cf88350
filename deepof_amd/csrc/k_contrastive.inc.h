// Contrastive embedding step: view construction + pairwise losses.
//
// Reference semantics restated here (paths under /root/reference/deepof/clustering):
//   * slice_time_per_sample / recompute_edges            model_utils_new.py:751-763, :332-363
//   * _make_augmented_view and its four augmentations    training.py:2128-2403
//       time-shifted half window -> joint-like rotations of graph branches about a pivot node ->
//       one linearly interpolated segment -> per-node offsets on (x | y, speed) -> edges recomputed
//     The random draws themselves are inputs (DofAugment): the host draws them, as the reference does
//     with torch's generator, so a test can feed the reference, the oracle and this kernel the same ones.
//   * select_contrastive_loss_pt, similarities, nce / dcl / hard_dcl losses       losses.py:35-249
//   * step_contrastive_distill: F.normalize on both embeddings, loss, pos/neg similarity logs
//                                                                                  training.py:482-589
// Views are HBM-bound row copies with a little per-frame math (one thread per output frame, the
// frame's node coordinates staged in LDS, conflict-free [node][thread]).  The loss is an all-pairs
// B x B problem with rows of only L <= 8 floats: every thread owns one row, the other side streams
// through LDS in 256-row tiles, exponentials are evaluated relative to the similarity bound 1 so no
// running maximum is needed (unit-norm inputs).  Three passes (row statistics, d/d z, d/d z_aug)
// recompute the similarities instead of materialising the B x B matrix (256 MB at B = 8192).
#include "dof_rt.h"

namespace {

#define DOF_CL_MAX_NODES 64

// no FMA contraction: matches torch's op-by-op rounding
__device__ __forceinline__ float cl_mul(float a, float b) { return dof_fmul_rn(a, b); }
__device__ __forceinline__ float cl_add(float a, float b) { return dof_fadd_rn(a, b); }

struct ViewArgs {
  const float* x_full;    // (B, Tf, N, 3)
  const int* edge_index;  // (E, 2)
  const int* start;       // (B) first source frame of the view, or null: the central start (half/2)
  int n_rot;
  int rot_pivot[DOF_MAX_ROT];
  unsigned long long rot_mask[DOF_MAX_ROT];
  const float* theta;     // (n_rot, B) radians
  const int* interp_t0;   // (B) or null
  const int* interp_len;  // (B), 0 = no interpolated segment
  const float* noise;     // (B, N, 3) or null
  float* x_out;           // (B, half, N, 3)
  float* a_out;           // (B, half, E, 1)
  int B, Tf, N, E, half;
};

__global__ void __launch_bounds__(64) k_cl_view(ViewArgs A) {
  __shared__ float sx[DOF_CL_MAX_NODES][64];
  __shared__ float sy[DOF_CL_MAX_NODES][64];
  const int lane = threadIdx.x;
  const int64_t gid = (int64_t)blockIdx.x * 64 + lane;
  if (gid >= (int64_t)A.B * A.half) return;  // no block-level barrier below: each thread owns its LDS column
  const int b = (int)(gid / A.half), t = (int)(gid % A.half);
  const int st = A.start ? A.start[b] : A.half / 2;
  const int ln = A.interp_len ? A.interp_len[b] : 0;
  const int t0 = A.interp_t0 ? A.interp_t0[b] : 0;
  const bool seg = ln > 0 && t >= t0 && t < t0 + ln;
  // (1-alpha) * frame(t0-1) + alpha * frame(t0+len) inside the segment.  The reference interpolates the
  // ROTATED end frames; rotating the interpolated frame is the same affine map (theta is per sample, not
  // per frame), so the interpolation is done first and one coordinate buffer suffices.
  float alpha = 0.0f;
  int ta = t, tb = t;
  if (seg) {
    alpha = ((float)t - ((float)t0 - 1.0f)) / (float)(ln + 1);
    alpha = fminf(fmaxf(alpha, 0.0f), 1.0f);
    ta = t0 - 1;
    tb = t0 + ln;
  }
  const float* fa = A.x_full + ((int64_t)b * A.Tf + st + ta) * A.N * 3;
  const float* fb = A.x_full + ((int64_t)b * A.Tf + st + tb) * A.N * 3;
  float* xo = A.x_out + gid * A.N * 3;
  for (int n = 0; n < A.N; ++n) {
    float x = fa[n * 3], y = fa[n * 3 + 1], s = fa[n * 3 + 2];
    if (seg) {
      const float om = 1.0f - alpha;
      x = cl_add(cl_mul(om, x), cl_mul(alpha, fb[n * 3]));
      y = cl_add(cl_mul(om, y), cl_mul(alpha, fb[n * 3 + 1]));
      s = cl_add(cl_mul(om, s), cl_mul(alpha, fb[n * 3 + 2]));
    }
    sx[n][lane] = x;
    sy[n][lane] = y;
    xo[n * 3 + 2] = A.noise ? s + A.noise[((int64_t)b * A.N + n) * 3 + 2] : s;
  }
  for (int r = 0; r < A.n_rot; ++r) {
    const float th = A.theta[(int64_t)r * A.B + b];
    const float c = cosf(th), s = sinf(th);
    const int pv = A.rot_pivot[r];
    const float px = sx[pv][lane], py = sy[pv][lane];
    const unsigned long long m = A.rot_mask[r];
    for (int n = 0; n < A.N; ++n)
      if ((m >> n) & 1ull) {
        const float rx = sx[n][lane] - px, ry = sy[n][lane] - py;
        sx[n][lane] = cl_add(cl_add(cl_mul(rx, c), -cl_mul(ry, s)), px);
        sy[n][lane] = cl_add(cl_add(cl_mul(rx, s), cl_mul(ry, c)), py);
      }
  }
  for (int n = 0; n < A.N; ++n) {
    float x = sx[n][lane], y = sy[n][lane];
    if (A.noise) {
      x += A.noise[((int64_t)b * A.N + n) * 3];
      y += A.noise[((int64_t)b * A.N + n) * 3 + 1];
      sx[n][lane] = x;
      sy[n][lane] = y;
    }
    xo[n * 3] = x;
    xo[n * 3 + 1] = y;
  }
  float* ao = A.a_out + gid * A.E;
  for (int e = 0; e < A.E; ++e) {
    const int i = A.edge_index[2 * e], j = A.edge_index[2 * e + 1];
    const float dx = sx[i][lane] - sx[j][lane], dy = sy[i][lane] - sy[j][lane];
    ao[e] = sqrtf(fmaxf(cl_add(cl_mul(dx, dx), cl_mul(dy, dy)), 1e-12f));
  }
}

// ---------------------------------------------------------------------------------------------
// pairwise losses
// ---------------------------------------------------------------------------------------------
enum { CL_SIM_COSINE = 0, CL_SIM_DOT = 1, CL_SIM_EUCLID = 2 };
enum { CL_LOSS_NCE = 0, CL_LOSS_DCL = 1, CL_LOSS_HARD = 2, CL_LOSS_FC = 3 };

struct ClArgs {
  const float* z;    // (B, L) encoder outputs of the central view
  const float* za;   // (B, L) ... of the augmented view
  float* zn;         // (2, B, L) row-normalised copies (F.normalize, eps 1e-12)
  float* inv;        // (2, B)  1 / max(|z|, 1e-12)
  float* rn;         // (2, B)  cosine: 1 / max(|zn|, 1e-8); otherwise 1
  float* rowstat;    // (B, 4)  s_ii, and the row's weights: dL/ds_ij = ca*e + cb*e^2 (j != i), dL/ds_ii = cd
  float* theta;      // (B) fc: largest similarity kept by the row's top-k elimination (+inf otherwise)
  int fc_keep;       // fc: negatives kept per row = (B-1) - ceil(min(topk, 0.5) * B)
  float* partial;    // (nblk, 3) block sums of loss_i, s_ii, sum_{j != i} s_ij
  float* dz;         // (B, L) d loss / d z
  float* dza;        // (B, L) d loss / d z_aug
  float* logs;       // DOF_LOG_* (total, pos_similarity, neg_similarity)
  const float* dzh;  // (B, L) d (distillation loss) / d zn of the central view, or null
  const float* dh_partial;  // per-block sums of the distillation loss
  int n_dh;                 // ... their count (workgroups of k_distill_head)
  int sim, loss_fn;
  float inv_T, tau, beta;
  int B, nblk;
};

template <int L>
__global__ void __launch_bounds__(256) k_cl_normalize(ClArgs A) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= 2 * (int64_t)A.B) return;
  const float* src = r < A.B ? A.z + r * L : A.za + (r - A.B) * L;
  float v[L], n2 = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    v[l] = src[l];
    n2 = fmaf(v[l], v[l], n2);
  }
  const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
  float m2 = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    v[l] *= inv;
    m2 = fmaf(v[l], v[l], m2);
    A.zn[r * L + l] = v[l];
  }
  A.inv[r] = inv;
  A.rn[r] = A.sim == CL_SIM_COSINE ? 1.0f / fmaxf(sqrtf(m2), 1e-8f) : 1.0f;
}

// similarity of own row x (scale rx) with other row y (scale ry); *aux = s^2/d for the euclidean kernel
template <int L>
__device__ __forceinline__ float cl_sim(int sim, const float* x, const float* y, float rx, float ry, float* aux) {
  if (sim == CL_SIM_EUCLID) {
    float d2 = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float df = x[l] - y[l];
      d2 = fmaf(df, df, d2);
    }
    const float d = sqrtf(fmaxf(d2, 0.0f));
    const float s = 1.0f / (1.0f + d);
    *aux = d > 0.0f ? s * s / d : 0.0f;
    return s;
  }
  float dot = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) dot = fmaf(x[l], y[l], dot);
  *aux = 0.0f;
  return sim == CL_SIM_COSINE ? dot * rx * ry : dot;
}

// order-preserving map float -> uint32 (and back) for the bisection of the fc loss
__device__ __forceinline__ unsigned cl_f2o(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float cl_o2f(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __builtin_bit_cast(float, u);
}

// The all-pairs kernels below give every row of the B x B similarity matrix to a GROUP of CLG neighbouring lanes
// (lane `sub` of the group takes the columns sub, sub + CLG, ... of each staged 256-column tile) and add the lanes'
// partial sums with a fixed xor butterfly.  One thread per row meant 32 workgroups walking 8192 columns each at batch
// 8192 (1.4 - 2.4 ms per kernel on a 256-CU chip); a group of 8 gives 256 workgroups and an 8x shorter walk.
constexpr int CLG = 8;
constexpr int CL_ROWS = 256 / CLG;  // rows per workgroup
__device__ __forceinline__ float cl_group_sum(float v) {
#pragma unroll
  for (int m = 1; m < CLG; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// fc loss (losses.py:176-208): each row drops its k LARGEST negatives.  theta_i = the fc_keep-th smallest
// off-diagonal similarity of row i, found by a 32-step bisection over the ordered bit patterns (every step
// recounts the row; similarities are recomputed, never stored).  A block handles 256 rows so the tiles of the
// other side are shared.
template <int L>
__global__ void __launch_bounds__(256) k_cl_fc_threshold(ClArgs A) {
  __shared__ float ty[256][L + 1];
  __shared__ float tr[256];
  const int tid = threadIdx.x, sub = tid % CLG;
  const int i = blockIdx.x * CL_ROWS + tid / CLG;
  const bool live = i < A.B;
  float x[L], rx = 1.0f;
  if (live) {
#pragma unroll
    for (int l = 0; l < L; ++l) x[l] = A.zn[(int64_t)i * L + l];
    rx = A.rn[i];
  }
  const float* yn = A.zn + (int64_t)A.B * L;
  const float* ryv = A.rn + A.B;
  unsigned lo = 0u, hi = 0xffffffffu;  // smallest v with count(s <= v) >= fc_keep
  for (int it = 0; it < 32; ++it) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    const float fm = cl_o2f(mid);
    float cnt = 0.0f;  // (a float count is exact up to 2^24 columns)
    for (int j0 = 0; j0 < A.B; j0 += 256) {
      __syncthreads();
      if (j0 + tid < A.B) {
#pragma unroll
        for (int l = 0; l < L; ++l) ty[tid][l] = yn[(int64_t)(j0 + tid) * L + l];
        tr[tid] = ryv[j0 + tid];
      }
      __syncthreads();
      if (!live) continue;
      const int nj = A.B - j0 < 256 ? A.B - j0 : 256;
      for (int jj = sub; jj < nj; jj += CLG) {
        float aux;
        const float s = cl_sim<L>(A.sim, x, ty[jj], rx, tr[jj], &aux);
        cnt += (j0 + jj != i && s <= fm) ? 1.0f : 0.0f;
      }
    }
    cnt = cl_group_sum(cnt);
    if (cnt >= (float)A.fc_keep) hi = mid; else lo = mid + 1;
  }
  if (live && sub == 0) A.theta[i] = A.fc_keep > 0 ? cl_o2f(hi) : -INFINITY;
}

template <int L>
__global__ void __launch_bounds__(256) k_cl_rowstats(ClArgs A) {
  __shared__ float ty[256][L + 1];
  __shared__ float tr[256];
  const int tid = threadIdx.x, sub = tid % CLG;
  const int i = blockIdx.x * CL_ROWS + tid / CLG;
  const bool live = i < A.B;
  float x[L], rx = 1.0f;
  if (live) {
#pragma unroll
    for (int l = 0; l < L; ++l) x[l] = A.zn[(int64_t)i * L + l];
    rx = A.rn[i];
  }
  const float* yn = A.zn + (int64_t)A.B * L;
  const float* ryv = A.rn + A.B;
  float a1 = 0.0f, a2 = 0.0f, soff = 0.0f, p = 0.0f;
  const float th = (A.loss_fn == CL_LOSS_FC && live) ? A.theta[i] : INFINITY;
  for (int j0 = 0; j0 < A.B; j0 += 256) {
    __syncthreads();
    if (j0 + tid < A.B) {
#pragma unroll
      for (int l = 0; l < L; ++l) ty[tid][l] = yn[(int64_t)(j0 + tid) * L + l];
      tr[tid] = ryv[j0 + tid];
    }
    __syncthreads();
    if (!live) continue;
    const int nj = A.B - j0 < 256 ? A.B - j0 : 256;
    for (int jj = sub; jj < nj; jj += CLG) {
      float aux;
      const float s = cl_sim<L>(A.sim, x, ty[jj], rx, tr[jj], &aux);
      if (j0 + jj == i) {
        p = s;
      } else if (s <= th) {
        const float e = __expf((s - 1.0f) * A.inv_T);
        a1 += e;
        a2 = fmaf(e, e, a2);
        soff += s;
      }
    }
  }
  a1 = cl_group_sum(a1); a2 = cl_group_sum(a2); soff = cl_group_sum(soff);
  p = cl_group_sum(p);  // exactly one lane of the group met the diagonal
  float out[3] = {0.0f, 0.0f, 0.0f};
  if (live && sub == 0) {
    const float ne = (float)(A.B - 1), invB = 1.0f / (float)A.B;
    const float pos = __expf((p - 1.0f) * A.inv_T);
    float den, ca, cb = 0.0f, cd;
    if (A.loss_fn == CL_LOSS_NCE || A.loss_fn == CL_LOSS_FC) {  // fc: the same form over the kept negatives
      den = pos + a1;
      ca = A.inv_T / den;
      cd = A.inv_T * (pos / den - 1.0f);
    } else {
      const bool hard = A.loss_fn == CL_LOSS_HARD && A.beta != 0.0f;
      const float R = hard ? A.beta * ne * a2 / a1 : a1;
      const float ng = (-A.tau * ne * pos + R) / (1.0f - A.tau);
      const float lo = (A.loss_fn == CL_LOSS_DCL ? ne : 1.0f) * __expf(-2.0f * A.inv_T);
      const bool open = ng >= lo;  // torch.clamp passes the gradient on [min, max]
      den = pos + (open ? ng : lo);
      const float k = open ? A.inv_T / (den * (1.0f - A.tau)) : 0.0f;
      cd = A.inv_T * (pos * (1.0f + (open ? -A.tau * ne / (1.0f - A.tau) : 0.0f)) / den - 1.0f);
      if (hard) {
        ca = -k * A.beta * ne * a2 / (a1 * a1);
        cb = k * 2.0f * A.beta * ne / a1;
      } else {
        ca = k;
      }
    }
    out[0] = __logf(den) - (p - 1.0f) * A.inv_T;
    out[1] = p;
    out[2] = soff;
    float* rs = A.rowstat + (int64_t)i * 4;
    rs[0] = p; rs[1] = ca * invB; rs[2] = cb * invB; rs[3] = cd * invB;
  }
  dof_block_colsum<3>(out, A.partial + (int64_t)blockIdx.x * 3);
}

// COLS = false: thread i owns row i of z (weights of its own row); COLS = true: thread j owns row j of
// z_aug and walks the rows i, whose weights travel with the tile.
template <int L, bool COLS>
__global__ void __launch_bounds__(256) k_cl_grad(ClArgs A) {
  __shared__ float ty[256][L + 1];
  __shared__ float tr[256];
  __shared__ float tw[256][4];
  const int tid = threadIdx.x, sub = tid % CLG;
  const int i = blockIdx.x * CL_ROWS + tid / CLG;
  const bool live = i < A.B;
  const float* own = A.zn + (COLS ? (int64_t)A.B * L : 0);
  const float* oth = A.zn + (COLS ? 0 : (int64_t)A.B * L);
  const float* rown = A.rn + (COLS ? A.B : 0);
  const float* roth = A.rn + (COLS ? 0 : A.B);
  float x[L], rx = 1.0f, ca = 0.0f, cb = 0.0f, cd = 0.0f, th = INFINITY;
  const bool fc = A.loss_fn == CL_LOSS_FC;
  if (live) {
#pragma unroll
    for (int l = 0; l < L; ++l) x[l] = own[(int64_t)i * L + l];
    rx = rown[i];
    if (!COLS) {
      ca = A.rowstat[(int64_t)i * 4 + 1]; cb = A.rowstat[(int64_t)i * 4 + 2]; cd = A.rowstat[(int64_t)i * 4 + 3];
      if (fc) th = A.theta[i];
    }
  }
  float V[L], sx = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) V[l] = 0.0f;
  for (int j0 = 0; j0 < A.B; j0 += 256) {
    __syncthreads();
    if (j0 + tid < A.B) {
#pragma unroll
      for (int l = 0; l < L; ++l) ty[tid][l] = oth[(int64_t)(j0 + tid) * L + l];
      tr[tid] = roth[j0 + tid];
      if (COLS) {
        tw[tid][0] = A.rowstat[(int64_t)(j0 + tid) * 4 + 1];
        tw[tid][1] = A.rowstat[(int64_t)(j0 + tid) * 4 + 2];
        tw[tid][2] = A.rowstat[(int64_t)(j0 + tid) * 4 + 3];
        tw[tid][3] = fc ? A.theta[j0 + tid] : INFINITY;
      }
    }
    __syncthreads();
    if (!live) continue;
    const int nj = A.B - j0 < 256 ? A.B - j0 : 256;
    for (int jj = sub; jj < nj; jj += CLG) {
      float aux;
      const float s = cl_sim<L>(A.sim, x, ty[jj], rx, tr[jj], &aux);
      if (COLS) {
        ca = tw[jj][0]; cb = tw[jj][1]; cd = tw[jj][2]; th = tw[jj][3];
      }
      float w;
      if (j0 + jj == i) {
        w = cd;
      } else if (s > th) {
        w = 0.0f;  // negative eliminated by the row's top-k rule
      } else {
        const float e = __expf((s - 1.0f) * A.inv_T);
        w = e * fmaf(cb, e, ca);
      }
      // d s / d own = vc * other - sc * own   (per similarity; see header)
      float vc, sc;
      if (A.sim == CL_SIM_EUCLID) {
        vc = w * aux; sc = vc;
      } else if (A.sim == CL_SIM_COSINE) {
        vc = w * tr[jj]; sc = w * s;
      } else {
        vc = w; sc = 0.0f;
      }
#pragma unroll
      for (int l = 0; l < L; ++l) V[l] = fmaf(vc, ty[jj][l], V[l]);
      sx += sc;
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) V[l] = cl_group_sum(V[l]);
  sx = cl_group_sum(sx);
  if (!live || sub != 0) return;
  float g[L], gz = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    if (A.sim == CL_SIM_COSINE) g[l] = rx * (V[l] - sx * rx * x[l]);
    else g[l] = V[l] - sx * x[l];
    if (!COLS && A.dzh) g[l] += A.dzh[(int64_t)i * L + l];  // distillation head acts on the normalised central view
    gz = fmaf(g[l], x[l], gz);
  }
  // backward of F.normalize: d z = (g - (g . zn) zn) / max(|z|, eps)
  const float inv = A.inv[(COLS ? A.B : 0) + i];
  float* dst = (COLS ? A.dza : A.dz) + (int64_t)i * L;
#pragma unroll
  for (int l = 0; l < L; ++l) dst[l] = inv * (g[l] - gz * x[l]);
}

__global__ void __launch_bounds__(64) k_cl_finalize(ClArgs A) {
  const int lane = threadIdx.x;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int k = lane; k < A.nblk; k += 64)
    for (int v = 0; v < 3; ++v) acc[v] += A.partial[(int64_t)k * 3 + v];
  for (int v = 0; v < 3; ++v)
    for (int off = 32; off > 0; off >>= 1) acc[v] += __shfl_down(acc[v], off);
  if (lane == 0) {
    const float B = (float)A.B;
    for (int k = 0; k < DOF_LOG_COUNT; ++k) A.logs[k] = 0.0f;
    float dist = 0.0f;
    if (A.dh_partial)
      for (int k = 0; k < A.n_dh; ++k) dist += A.dh_partial[k];
    A.logs[DOF_LOG_DISTILL] = dist;
    A.logs[DOF_LOG_TOTAL] = acc[0] / B + dist;
    A.logs[DOF_LOG_POS_SIM] = acc[1] / B;
    const float per_row = A.loss_fn == CL_LOSS_FC ? (float)A.fc_keep : B - 1.0f;
    A.logs[DOF_LOG_NEG_SIM] = per_row > 0.0f ? acc[2] / (B * per_row) : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// Generic distillation head of the VQ-VAE / contrastive steps (training.py:344-372, 553-580):
//   logits = W z + b ; tau_b sharpened (softmax(log tau / T)) ; optional confidence weight
//   loss = lambda * mean_b w_b * soft-CE(logits_b, tau_b)
// z is read through (row, column) strides so the [L][Bp] encoder output and the (B, L) normalised embeddings both fit.
// ---------------------------------------------------------------------------------------------
struct DistillHeadArgs {
  const float* z;
  int64_t zs_b, zs_l;     // element (b, l) at z[b*zs_b + l*zs_l]
  const float* tau;       // (B, K) teacher targets of this batch
  const float *w, *bias;  // (K, L), (K)
  const float* hyper;     // lambda, sharpening T, confidence weighting
  float* dl;              // (B, K) d loss / d logits
  float* dz;              // d loss / d z, element (b, l) at dz[b*dzs_b + l*dzs_l]
  int64_t dzs_b, dzs_l;
  float* partial;         // per-block sums of lambda * w_b * CE_b / B
  int K;
  int64_t B;
};

template <int L>
__global__ void __launch_bounds__(256) k_distill_head(DistillHeadArgs A) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float part[1] = {0.0f};
  if (b < A.B) {
    const int K = A.K;
    const float lam = A.hyper[DOF_H_LAMBDA_DISTILL], T = A.hyper[DOF_H_DISTILL_T];
    float z[L], lp[64], ts[64];
#pragma unroll
    for (int l = 0; l < L; ++l) z[l] = A.z[b * A.zs_b + l * A.zs_l];
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float acc = A.bias[k];
#pragma unroll
      for (int l = 0; l < L; ++l) acc = fmaf(A.w[k * L + l], z[l], acc);
      lp[k] = acc;
      mx = fmaxf(mx, acc);
    }
    float se = 0.0f;
    for (int k = 0; k < K; ++k) se += expf(lp[k] - mx);
    const float lse = mx + logf(se);
    // sharpened targets
    float tmx = -INFINITY, cmax = 0.0f;
    for (int k = 0; k < K; ++k) {
      ts[k] = A.tau[b * K + k];
      if (T > 0.0f) {
        ts[k] = logf(fmaxf(ts[k], 1e-8f)) / T;
        tmx = fmaxf(tmx, ts[k]);
      }
    }
    if (T > 0.0f) {
      float s2 = 0.0f;
      for (int k = 0; k < K; ++k) {
        ts[k] = expf(ts[k] - tmx);
        s2 += ts[k];
      }
      for (int k = 0; k < K; ++k) ts[k] /= s2;
    }
    for (int k = 0; k < K; ++k) cmax = fmaxf(cmax, ts[k]);
    float wgt = 1.0f;
    if (A.hyper[DOF_H_CONF_W] != 0.0f) {
      const float thr = A.hyper[DOF_H_CONF_THR];
      wgt = fminf(fmaxf((cmax - thr) / fmaxf(1e-6f, 1.0f - thr), 0.0f), 1.0f);
    }
    float ce = 0.0f, st = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float tc = fminf(fmaxf(ts[k], 1e-8f), 1.0f);
      ce -= tc * (lp[k] - lse);
      st += tc;
    }
    const float sc = lam * wgt / (float)A.B;
    part[0] = sc * ce;
    float dzv[L];
#pragma unroll
    for (int l = 0; l < L; ++l) dzv[l] = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float tc = fminf(fmaxf(ts[k], 1e-8f), 1.0f);
      const float g = sc * (expf(lp[k] - lse) * st - tc);
      A.dl[b * K + k] = g;
#pragma unroll
      for (int l = 0; l < L; ++l) dzv[l] = fmaf(A.w[k * L + l], g, dzv[l]);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) A.dz[b * A.dzs_b + l * A.dzs_l] = dzv[l];
  }
  dof_block_colsum<1>(part, A.partial + blockIdx.x);
}

// d W[k][l] = sum_b dl[b][k] z[b][l], d bias[k] = sum_b dl[b][k]; block = cluster k, lane l (lane L = bias)
template <int L>
__global__ void __launch_bounds__(64) k_distill_wgrad(const float* __restrict__ dl, const float* __restrict__ z,
                                                      int64_t zs_b, int64_t zs_l, float* __restrict__ gw,
                                                      float* __restrict__ gb, int K, int64_t B) {
  const int k = blockIdx.x, l = threadIdx.x;
  if (l > L) return;
  float acc = 0.0f;
  for (int64_t b = 0; b < B; ++b) acc = fmaf(dl[b * K + k], l < L ? z[b * zs_b + l * zs_l] : 1.0f, acc);
  if (l < L) gw[k * L + l] = acc;
  else gb[k] = acc;
}

__global__ void __launch_bounds__(256) k_fill_f32(float* __restrict__ p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// (B, L) reference layout <-> [L][Bp] workspace layout
template <int L>
__global__ void __launch_bounds__(256) k_cl_export(const float* __restrict__ enc, float* __restrict__ z, int64_t B,
                                                   int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
#pragma unroll
  for (int l = 0; l < L; ++l) z[b * L + l] = enc[(int64_t)l * Bp + b];
}
template <int L>
__global__ void __launch_bounds__(256) k_cl_import(const float* __restrict__ dz, float* __restrict__ denc, int64_t B,
                                                   int64_t Bp) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
#pragma unroll
  for (int l = 0; l < L; ++l) denc[(int64_t)l * Bp + b] = dz[b * L + l];
}

}  // namespace
