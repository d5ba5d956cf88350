// Pose-table preprocessing on the device, HBM-bound float64 streaming.
//
// Replaces the reference's pandas / sklearn host pipeline (paths under /root/reference/deepof)
//   TableDict.preprocess          data.py:3773-3916   (up to, not including, extract_windows)
//   scale_table                   utils.py:2425-2566  size factors, size normalisation, log1p, per-video scaling
//   _pp_pass1_collect_samples     utils.py:2665-2792  + _pp_fit_global_scaler :2795-2863  global StandardScaler fit
//   _pp_apply_global              utils.py:2866-2921
//   _pp_pass2_scale_and_save      utils.py:2924-3027  clip -> NaN -> interpolate, _pp_sanitize_numeric :2577-2583
// for scale="standard".  The raw tables of every video sit concatenated in HBM as (frames, C) float64; the result
// is written straight into the fp32 frame tables dof_window_gather reads, so neither the scaled tables nor the
// W-fold window blow-up ever exist on the host.
//
// Passes over the raw table (8 B/element each; everything else is O(videos x columns)):
//   k_pp_size     4 columns per animal: hypot(nose - tail base) -> exact nan-median by bitwise bisection
//   k_pp_stats    one pass: shifted sums per (video, strip, column) -> (n, mean, M2), all rows and sampled rows
//   k_pp_edges    first / last valid row per (32-row tile, output column) under the final transform
//   k_pp_finish   transform, clip, interpolate across the tile (neighbours from k_pp_carry), cast, write fp32
// The per-video and the global StandardScaler statistics come from the ONE statistics pass: the per-video
// transform is affine per column, so the statistics of the per-video-standardised samples follow from
// (n, mean, M2) of the sampled rows, merged over videos with Chan's pairwise update in a fixed order
// (run-to-run deterministic).  The complete transform of an element is then u = x * rdiv [-> log1p(max(u,0))],
// z = u * a + b with three float64 coefficients per (video, column).
#include <cmath>
#include <cstring>

#include "dof_rt.h"
#include "deepof_hip.h"

namespace {

constexpr int PP_RS = 256;  // rows per statistics strip
constexpr int PP_TR = 32;   // rows per interpolation tile
constexpr double PP_EPS = 2.220446049250313e-16;

struct PpStat {
  double n, mean, m2;
};

__device__ __forceinline__ bool pp_isnan(double x) { return x != x; }
__device__ __forceinline__ uint64_t pp_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
__device__ __forceinline__ double pp_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}
__device__ __forceinline__ double pp_nanv() { return pp_from_bits(0x7ff8000000000000ull); }

// Chan, Golub, LeVeque pairwise update; entries with n == 0 never contribute their mean
__device__ __forceinline__ PpStat pp_merge(const PpStat a, const PpStat b) {
  if (b.n == 0.0) return a;
  if (a.n == 0.0) return b;
  PpStat r;
  r.n = a.n + b.n;
  const double d = b.mean - a.mean;
  r.mean = a.mean + d * (b.n / r.n);
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
  return r;
}
__device__ __forceinline__ PpStat pp_from_sums(double n, double shift, double s1, double s2) {
  PpStat r;
  r.n = n;
  r.mean = n > 0.0 ? shift + s1 / n : 0.0;
  r.m2 = n > 0.0 ? s2 - s1 * s1 / n : 0.0;
  return r;
}
// StandardScaler's (mean_, scale_) from (n, mean, M2): population variance, near-constant features get scale 1
// (sklearn _is_constant_feature / _handle_zeros_in_scale); nothing seen -> NaN like the 0/0 there
__device__ __forceinline__ void pp_fit(const PpStat s, double* mean, double* scale) {
  if (s.n == 0.0) {
    *mean = pp_nanv();
    *scale = pp_nanv();
    return;
  }
  const double var = s.m2 / s.n;
  const double bound = s.n * PP_EPS * var + (s.n * s.mean * PP_EPS) * (s.n * s.mean * PP_EPS);
  *mean = s.mean;
  *scale = (var <= bound) ? 1.0 : sqrt(var);
}

// video v and unit k (strip / tile of R rows) of flat slot b; slots of video v start at video_off[v]/R + v
__device__ __forceinline__ bool pp_locate(const int64_t* __restrict__ video_off, int V, int R, int64_t b, int* v_out,
                                          int64_t* k_out) {
  int lo = 0, hi = V - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (video_off[mid] / R + mid <= b) lo = mid; else hi = mid - 1;
  }
  const int64_t k = b - (video_off[lo] / R + lo);
  const int64_t len = video_off[lo + 1] - video_off[lo];
  *v_out = lo;
  *k_out = k;
  return k >= 0 && k * R < len;
}
static inline int64_t pp_slots(int64_t n_frames, int V, int R) { return n_frames / R + V + 1; }

__device__ __forceinline__ int pp_mode(int kind, int speed_mode, int dist_mode, int coord_mode) {
  if (kind == DOF_PP_SPEED) return speed_mode;
  if (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA) return dist_mode;
  if (kind == DOF_PP_COORD) return coord_mode;
  return DOF_PP_MODE_NONE;
}

// ---------------------------------------------------------------------------------------------------------
// size factor of animal a in video v: nan-median over the video's rows of hypot(nose - tail base).
// Exact selection without a sort: the order statistics (n-1)/2 and n/2 are built bit by bit from the top
// (non-negative doubles order like their bit patterns), one counting sweep of the L2-resident lengths per bit.
__global__ void __launch_bounds__(256) k_pp_size(const double* __restrict__ raw, const int64_t* __restrict__ video_off,
                                                 const int* __restrict__ size_ref, int C, int A, int64_t F,
                                                 double* __restrict__ hyp, double* __restrict__ s_out) {
  __shared__ long long red[2][256];
  const int a = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
  const int c0 = size_ref[4 * a], c1 = size_ref[4 * a + 1], c2 = size_ref[4 * a + 2], c3 = size_ref[4 * a + 3];
  double* out = s_out + (int64_t)v * (A + 1) + a;
  if (c0 < 0 || c1 < 0 || c2 < 0 || c3 < 0) {
    if (tid == 0) *out = pp_nanv();
    return;
  }
  const int64_t r0 = video_off[v], r1 = video_off[v + 1];
  double* h = hyp + (int64_t)a * F;
  long long nv = 0;
  for (int64_t r = r0 + tid; r < r1; r += 256) {
    const double* row = raw + r * C;
    const double len = hypot(row[c0] - row[c2], row[c1] - row[c3]);
    h[r] = len;
    nv += !pp_isnan(len);
  }
  red[0][tid] = nv;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[0][tid] += red[0][tid + s];
    __syncthreads();
  }
  const long long n = red[0][0];
  __syncthreads();
  if (n == 0) {
    if (tid == 0) *out = pp_nanv();
    return;
  }
  const long long k1 = (n - 1) / 2, k2 = n / 2;
  uint64_t res1 = 0, res2 = 0;
  for (int bit = 62; bit >= 0; --bit) {
    const uint64_t t1 = res1 | (1ull << bit), t2 = res2 | (1ull << bit);
    long long q1 = 0, q2 = 0;
    for (int64_t r = r0 + tid; r < r1; r += 256) {
      const double len = h[r];
      if (!pp_isnan(len)) {
        const uint64_t key = pp_bits(len);
        q1 += key < t1;
        q2 += key < t2;
      }
    }
    red[0][tid] = q1;
    red[1][tid] = q2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        red[0][tid] += red[0][tid + s];
        red[1][tid] += red[1][tid + s];
      }
      __syncthreads();
    }
    if (red[0][0] <= k1) res1 = t1;
    if (red[1][0] <= k2) res2 = t2;
    __syncthreads();
  }
  if (tid == 0) *out = (pp_from_bits(res1) + pp_from_bits(res2)) / 2.0;
}

// default factor (median of the usable factors, else 1), substitution of unusable factors, and the reciprocal
// size divisor of every column of every video
__global__ void __launch_bounds__(256) k_pp_divisors(const int* __restrict__ chain_off, const int* __restrict__ chain,
                                                     int C, int A, int inter_scale, double* __restrict__ s_out,
                                                     double* __restrict__ rdiv) {
  __shared__ double S[DOF_PP_MAX_ANIMALS + 1];
  const int v = blockIdx.x;
  double* sv = s_out + (int64_t)v * (A + 1);
  if (threadIdx.x == 0) {
    double ok[DOF_PP_MAX_ANIMALS];
    int m = 0;
    for (int a = 0; a < A; ++a) {
      const double s = sv[a];
      if (s == s && s > 0.0 && s < INFINITY) {
        int i = m++;
        while (i > 0 && ok[i - 1] > s) { ok[i] = ok[i - 1]; --i; }
        ok[i] = s;
      }
    }
    const double dflt = m == 0 ? 1.0 : (ok[(m - 1) / 2] + ok[m / 2]) / 2.0;
    for (int a = 0; a < A; ++a) {
      const double s = sv[a];
      S[a] = (s == s && s > 0.0 && s < INFINITY) ? s : dflt;
      sv[a] = S[a];
    }
    S[A] = dflt;
    sv[A] = dflt;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double d = 1.0;
    for (int e = chain_off[c]; e < chain_off[c + 1]; ++e) {
      const int a1 = chain[3 * e], a2 = chain[3 * e + 1], same = chain[3 * e + 2];
      const double s1 = S[a1 < 0 ? A : a1], s2 = S[a2 < 0 ? A : a2];
      double s = s1;
      if (!same) s = inter_scale == 0 ? 0.5 * (s1 + s2) : inter_scale == 1 ? sqrt(s1 * s2) : S[A];
      d *= s;
    }
    rdiv[(int64_t)v * C + c] = 1.0 / d;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The statistics pass.  A workgroup = one strip of PP_RS rows of one video; lanes run along the columns
// (a row is one contiguous 8C-byte run, so a wavefront reads 512 contiguous bytes), 4 wavefronts interleave
// the rows.  Per-thread sums are taken about the first value seen (no cancellation), turned into
// (n, mean, M2) and merged 4 -> 1 in LDS; the strip's result is a partial for the fixed-order finalize.
__global__ void __launch_bounds__(256) k_pp_stats(const double* __restrict__ raw, const int64_t* __restrict__ video_off,
                                                  const int* __restrict__ col_kind, const double* __restrict__ rdiv,
                                                  const uint8_t* __restrict__ mask, int V, int C, int log_dist,
                                                  int speed_mode, int dist_mode, int coord_mode,
                                                  PpStat* __restrict__ part_all, PpStat* __restrict__ part_smp) {
  __shared__ PpStat sh[2][4][64];
  int v;
  int64_t strip;
  const bool live = pp_locate(video_off, V, PP_RS, blockIdx.x, &v, &strip);
  if (!live) return;
  const int64_t r0 = video_off[v] + strip * PP_RS;
  const int64_t vend = video_off[v + 1];
  const int64_t r1 = r0 + PP_RS < vend ? r0 + PP_RS : vend;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int cbase = 0; cbase < C; cbase += 64) {
    const int c = cbase + lane;
    double n = 0.0, shift = 0.0, s1 = 0.0, s2 = 0.0, nb = 0.0, s1b = 0.0, s2b = 0.0;
    bool need = false;
    if (c < C) {
      const int kind = col_kind[c];
      need = pp_mode(kind, speed_mode, dist_mode, coord_mode) != DOF_PP_MODE_NONE;
      if (need) {
        const double rd = rdiv[(int64_t)v * C + c];
        const bool lg = log_dist && (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA);
        for (int64_t r = r0 + rg; r < r1; r += 4) {
          double u = raw[r * C + c] * rd;
          if (lg) {
            if (u < 0.0) u = 0.0;
            u = log1p(u);
          }
          if (!pp_isnan(u)) {
            if (n == 0.0) shift = u;
            const double d = u - shift;
            n += 1.0;
            s1 += d;
            s2 += d * d;
            if (mask && mask[r]) {
              nb += 1.0;
              s1b += d;
              s2b += d * d;
            }
          }
        }
      }
    }
    sh[0][rg][lane] = pp_from_sums(n, shift, s1, s2);
    sh[1][rg][lane] = mask ? pp_from_sums(nb, shift, s1b, s2b) : sh[0][rg][lane];
    __syncthreads();
    if (rg < 2 && c < C) {
      PpStat t = sh[rg][0][lane];
      for (int g = 1; g < 4; ++g) t = pp_merge(t, sh[rg][g][lane]);
      (rg == 0 ? part_all : part_smp)[(int64_t)blockIdx.x * C + c] = t;
    }
    __syncthreads();
  }
}

// per video: strips -> columns -> groups; per-video (mean, scale); statistics of the per-video-standardised
// sampled rows (what the global scalers are fitted on)
__global__ void __launch_bounds__(256) k_pp_video_fin(const int64_t* __restrict__ video_off,
                                                      const int* __restrict__ col_kind,
                                                      const PpStat* __restrict__ part_all,
                                                      const PpStat* __restrict__ part_smp, int C, int speed_mode,
                                                      int dist_mode, double* __restrict__ vscale,
                                                      PpStat* __restrict__ ystat) {
  __shared__ PpStat col_all[DOF_PP_MAX_COLS];
  __shared__ PpStat col_smp[DOF_PP_MAX_COLS];
  __shared__ PpStat grp[3];
  const int v = blockIdx.x;
  const int64_t slot0 = video_off[v] / PP_RS + v;
  const int64_t nstrip = (video_off[v + 1] - video_off[v] + PP_RS - 1) / PP_RS;
  for (int c = threadIdx.x; c < C; c += 256) {
    PpStat a = {0.0, 0.0, 0.0}, s = {0.0, 0.0, 0.0};
    for (int64_t k = 0; k < nstrip; ++k) {
      a = pp_merge(a, part_all[(slot0 + k) * C + c]);
      s = pp_merge(s, part_smp[(slot0 + k) * C + c]);
    }
    col_all[c] = a;
    col_smp[c] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int kind = DOF_PP_SPEED + threadIdx.x;  // speed, inner, intra
    PpStat g = {0.0, 0.0, 0.0};
    for (int c = 0; c < C; ++c)
      if (col_kind[c] == kind) g = pp_merge(g, col_all[c]);
    grp[threadIdx.x] = g;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int kind = col_kind[c];
    const int mode = kind == DOF_PP_SPEED ? speed_mode
                     : (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA) ? dist_mode : DOF_PP_MODE_NONE;
    double m = 0.0, s = 1.0;
    if (mode == DOF_PP_MODE_PER_COLUMN) pp_fit(col_all[c], &m, &s);
    if (mode == DOF_PP_MODE_GROUPWISE) pp_fit(grp[kind - DOF_PP_SPEED], &m, &s);
    vscale[((int64_t)v * C + c) * 2] = m;
    vscale[((int64_t)v * C + c) * 2 + 1] = s;
    PpStat y = col_smp[c];
    y.mean = (y.mean - m) / s;
    y.m2 = y.m2 / (s * s);
    if (pp_isnan(y.mean) || pp_isnan(y.m2)) y.n = 0.0;
    ystat[(int64_t)v * C + c] = y;
  }
}

// global scalers: videos -> columns -> groups (speed, inner, intra, coord)
__global__ void __launch_bounds__(256) k_pp_global_fin(const int* __restrict__ col_kind,
                                                       const PpStat* __restrict__ ystat, int V, int C, int speed_mode,
                                                       int dist_mode, int coord_mode, double* __restrict__ scaler) {
  __shared__ PpStat col[DOF_PP_MAX_COLS];
  __shared__ PpStat grp[4];
  for (int c = threadIdx.x; c < C; c += 256) {
    PpStat g = {0.0, 0.0, 0.0};
    for (int v = 0; v < V; ++v) g = pp_merge(g, ystat[(int64_t)v * C + c]);
    col[c] = g;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int kind = DOF_PP_COORD + threadIdx.x;  // coord, speed, inner, intra
    PpStat g = {0.0, 0.0, 0.0};
    for (int c = 0; c < C; ++c)
      if (col_kind[c] == kind) g = pp_merge(g, col[c]);
    grp[threadIdx.x] = g;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int kind = col_kind[c];
    const int mode = pp_mode(kind, speed_mode, dist_mode, coord_mode);
    double m = 0.0, s = 1.0;
    if (mode != DOF_PP_MODE_NONE) {
      const PpStat st = mode == DOF_PP_MODE_PER_COLUMN ? col[c] : grp[kind - DOF_PP_COORD];
      pp_fit(st, &m, &s);  // nothing sampled -> NaN like the reference's 0/0 (such columns hold no value anyway)
    }
    scaler[2 * c] = m;
    scaler[2 * c + 1] = s;
  }
}

// coefficients of the complete element transform: u = x * cf[0] [log1p(max(u, 0))]; z = u * cf[1] + cf[2]
__global__ void __launch_bounds__(256) k_pp_coef(const double* __restrict__ rdiv, const double* __restrict__ vscale,
                                                 const double* __restrict__ scaler, int C, double* __restrict__ coef,
                                                 double* __restrict__ video_scaler) {
  const int v = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int64_t i = (int64_t)v * C + c;
    const double m = vscale[2 * i], s = vscale[2 * i + 1], gm = scaler[2 * c], gs = scaler[2 * c + 1];
    const double p = 1.0 / s, q = 1.0 / gs;
    coef[3 * i] = rdiv[i];
    coef[3 * i + 1] = p * q;
    coef[3 * i + 2] = -(m * p * q + gm * q);
    if (video_scaler) {
      video_scaler[2 * i] = m;
      video_scaler[2 * i + 1] = s;
    }
  }
}

struct PpOutArgs {
  const double* raw;
  const int64_t* video_off;
  const int* col_kind;
  const int* out_cols;
  const double* coef;
  int V, C, n_out, log_dist;
  double clip;
};

// final value of raw element x of column c (coefficients cf); NaN when missing or clipped
__device__ __forceinline__ double pp_value(double x, const double* __restrict__ cf, int kind, int log_dist, double clip) {
  double u = x * cf[0];
  if (log_dist && (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA)) {
    if (u < 0.0) u = 0.0;
    u = log1p(u);
  }
  const double z = u * cf[1] + cf[2];
  const bool clipped = clip > 0.0 && kind >= DOF_PP_COORD && kind <= DOF_PP_DIST_INTRA && fabs(z) > clip;
  return clipped ? pp_nanv() : z;
}

// first / last valid row (relative to the video start, -1 = none) of every output column in every PP_TR-row tile
__global__ void __launch_bounds__(256) k_pp_edges(PpOutArgs A, int* __restrict__ first_v, int* __restrict__ last_v) {
  __shared__ int shf[4][64], shl[4][64];
  int v;
  int64_t tile;
  if (!pp_locate(A.video_off, A.V, PP_TR, blockIdx.x, &v, &tile)) return;
  const int64_t voff = A.video_off[v], vlen = A.video_off[v + 1] - voff;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t t0 = tile * PP_TR;
  for (int jb = 0; jb < A.n_out; jb += 64) {
    const int j = jb + lane;
    int f = -1, l = -1;
    if (j < A.n_out) {
      const int c = A.out_cols[j], kind = A.col_kind[c];
      const double* cf = A.coef + ((int64_t)v * A.C + c) * 3;
      for (int k = 0; k < PP_TR / 4; ++k) {
        const int64_t row = t0 + rg * (PP_TR / 4) + k;
        if (row < vlen) {
          const double z = pp_value(A.raw[(voff + row) * A.C + c], cf, kind, A.log_dist, A.clip);
          if (!pp_isnan(z)) {
            if (f < 0) f = (int)row;
            l = (int)row;
          }
        }
      }
    }
    shf[rg][lane] = f;
    shl[rg][lane] = l;
    __syncthreads();
    if (rg == 0 && j < A.n_out) {
      int ff = -1, ll = -1;
      for (int g = 0; g < 4; ++g) {
        if (ff < 0) ff = shf[g][lane];
        if (shl[g][lane] >= 0) ll = shl[g][lane];
      }
      first_v[(int64_t)blockIdx.x * A.n_out + j] = ff;
      last_v[(int64_t)blockIdx.x * A.n_out + j] = ll;
    }
    __syncthreads();
  }
}

// nearest valid row before / after every tile, per (video, output column): prefix max of the tiles' last valid
// row and suffix min of their first valid row (chunked per thread + a 256-entry scan in LDS)
__global__ void __launch_bounds__(256) k_pp_carry(const int64_t* __restrict__ video_off, int n_out,
                                                  const int* __restrict__ first_v, const int* __restrict__ last_v,
                                                  int* __restrict__ prev_v, int* __restrict__ next_v) {
  __shared__ int sh_max[256], sh_min[256];
  const int j = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
  const int64_t slot0 = video_off[v] / PP_TR + v;
  const int64_t nt = (video_off[v + 1] - video_off[v] + PP_TR - 1) / PP_TR;
  const int64_t chunk = (nt + 255) / 256;
  const int64_t a = tid * chunk, b = a + chunk < nt ? a + chunk : nt;
  const int none = 0x7fffffff;
  int mx = -1, mn = none;
  for (int64_t t = a; t < b; ++t) {
    const int l = last_v[(slot0 + t) * n_out + j], f = first_v[(slot0 + t) * n_out + j];
    if (l > mx) mx = l;
    if (f >= 0 && f < mn) mn = f;
  }
  sh_max[tid] = mx;
  sh_min[tid] = mn;
  __syncthreads();
  int carry = -1, back = none;
  for (int i = 0; i < tid; ++i)
    if (sh_max[i] > carry) carry = sh_max[i];
  for (int i = 255; i > tid; --i)
    if (sh_min[i] < back) back = sh_min[i];
  for (int64_t t = a; t < b; ++t) {
    prev_v[(slot0 + t) * n_out + j] = carry;
    const int l = last_v[(slot0 + t) * n_out + j];
    if (l > carry) carry = l;
  }
  for (int64_t t = b - 1; t >= a; --t) {
    next_v[(slot0 + t) * n_out + j] = back == none ? -1 : back;
    const int f = first_v[(slot0 + t) * n_out + j];
    if (f >= 0 && f < back) back = f;
  }
}

#ifdef DOF_EMU
#define PP_MUL_RN(a, b) ((a) * (b))
#define PP_ADD_RN(a, b) ((a) + (b))
#else
#define PP_MUL_RN(a, b) __dmul_rn((a), (b))
#define PP_ADD_RN(a, b) __dadd_rn((a), (b))
#endif

// One PP_TR-row tile of one video: transform + clip into LDS, fill the gaps per column (numpy.interp's
// slope * (x - x0) + y0 with separately rounded multiply and add, flat beyond the first / last valid row of the
// video, 0 for a column without any), then one coalesced fp32 store of whole frame-table rows.
template <int NO, int TR>
__global__ void __launch_bounds__(256) k_pp_finish(PpOutArgs A, const int* __restrict__ prev_v,
                                                   const int* __restrict__ next_v, int n_node, int n_edge,
                                                   float* __restrict__ node_out, float* __restrict__ edge_out,
                                                   float* __restrict__ angle_out) {
  static_assert(PP_TR % TR == 0, "sub-tile must divide the tile");
  __shared__ double zt[TR][NO + 1];
  int v;
  int64_t tile;
  if (!pp_locate(A.video_off, A.V, PP_TR, blockIdx.x, &v, &tile)) return;
  const int64_t voff = A.video_off[v], vlen = A.video_off[v + 1] - voff;
  const int n_out = A.n_out;
  for (int sub = 0; sub < PP_TR / TR; ++sub) {
    const int64_t t0 = tile * PP_TR + sub * TR;
    if (t0 >= vlen) break;
    const int nrows = (int)(vlen - t0 < TR ? vlen - t0 : TR);
    for (int idx = threadIdx.x; idx < nrows * n_out; idx += 256) {
      const int r = idx / n_out, j = idx - r * n_out;
      const int c = A.out_cols[j];
      zt[r][j] = pp_value(A.raw[(voff + t0 + r) * A.C + c], A.coef + ((int64_t)v * A.C + c) * 3, A.col_kind[c],
                          A.log_dist, A.clip);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_out; j += 256) {
      const int c = A.out_cols[j], kind = A.col_kind[c];
      const double* cf = A.coef + ((int64_t)v * A.C + c) * 3;
      // nearest valid rows outside this sub-tile: inside the tile they are found by walking, outside from the carry
      int64_t p = -1, q_after = -2;
      double pv = 0.0, qv_after = 0.0;
      {
        // rows of earlier sub-tiles of the same tile are not in LDS any more: look them up through the raw table
        int64_t cand = prev_v[(int64_t)blockIdx.x * n_out + j];
        for (int64_t row = t0 - 1; row >= tile * PP_TR; --row) {
          const double z = pp_value(A.raw[(voff + row) * A.C + c], cf, kind, A.log_dist, A.clip);
          if (!pp_isnan(z)) { cand = row; break; }
        }
        p = cand;
        if (p >= 0) pv = pp_value(A.raw[(voff + p) * A.C + c], cf, kind, A.log_dist, A.clip);
      }
      int r = 0;
      while (r < nrows) {
        const double z = zt[r][j];
        if (!pp_isnan(z)) {
          p = t0 + r;
          pv = z;
          ++r;
          continue;
        }
        int e = r;
        while (e < nrows && pp_isnan(zt[e][j])) ++e;
        int64_t q = -1;
        double qv = 0.0;
        if (e < nrows) {
          q = t0 + e;
          qv = zt[e][j];
        } else {
          if (q_after == -2) {
            int64_t cand = next_v[(int64_t)blockIdx.x * n_out + j];
            const int64_t tile_end = (tile + 1) * PP_TR < vlen ? (tile + 1) * PP_TR : vlen;
            for (int64_t row = t0 + nrows; row < tile_end; ++row) {
              const double zz = pp_value(A.raw[(voff + row) * A.C + c], cf, kind, A.log_dist, A.clip);
              if (!pp_isnan(zz)) { cand = row; break; }
            }
            q_after = cand;
            if (cand >= 0) qv_after = pp_value(A.raw[(voff + cand) * A.C + c], cf, kind, A.log_dist, A.clip);
          }
          q = q_after;
          qv = qv_after;
        }
        for (int k = r; k < e; ++k) {
          double val = 0.0;
          if (p >= 0 && q >= 0) {
            const double slope = (qv - pv) / (double)(q - p);
            val = PP_ADD_RN(PP_MUL_RN(slope, (double)(t0 + k - p)), pv);
          } else if (p >= 0) {
            val = pv;
          } else if (q >= 0) {
            val = qv;
          }
          zt[k][j] = val;
        }
        r = e;
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nrows * n_out; idx += 256) {
      const int r = idx / n_out, j = idx - r * n_out;
      const int64_t row = voff + t0 + r;
      const float val = (float)zt[r][j];
      if (j < n_node) node_out[row * n_node + j] = val;
      else if (j < n_node + n_edge) edge_out[row * n_edge + (j - n_node)] = val;
      else angle_out[row * (n_out - n_node - n_edge) + (j - n_node - n_edge)] = val;
    }
    __syncthreads();
  }
}

struct PpWorkspace {
  double *hyp, *sfac, *rdiv, *vscale, *coef;
  PpStat *part_all, *part_smp, *ystat;
  int *first_v, *last_v, *prev_v, *next_v;
  int64_t bytes;
};

PpWorkspace pp_layout(const DofPreprocDims& d, void* base) {
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t n) {
    char* q = p ? p + off : nullptr;
    off += (n + 255) / 256 * 256;
    return q;
  };
  const int n_out = d.n_node_cols + d.n_edge_cols + d.n_angle_cols;
  const int64_t strips = pp_slots(d.n_frames, d.n_videos, PP_RS), tiles = pp_slots(d.n_frames, d.n_videos, PP_TR);
  const int64_t vc = (int64_t)d.n_videos * d.n_cols;
  PpWorkspace w;
  w.hyp = (double*)take((int64_t)(d.n_animals > 0 ? d.n_animals : 1) * d.n_frames * 8);
  w.sfac = (double*)take((int64_t)d.n_videos * (d.n_animals + 1) * 8);
  w.rdiv = (double*)take(vc * 8);
  w.vscale = (double*)take(vc * 16);
  w.coef = (double*)take(vc * 24);
  w.part_all = (PpStat*)take(strips * d.n_cols * (int64_t)sizeof(PpStat));
  w.part_smp = (PpStat*)take(strips * d.n_cols * (int64_t)sizeof(PpStat));
  w.ystat = (PpStat*)take(vc * (int64_t)sizeof(PpStat));
  w.first_v = (int*)take(tiles * n_out * 4);
  w.last_v = (int*)take(tiles * n_out * 4);
  w.prev_v = (int*)take(tiles * n_out * 4);
  w.next_v = (int*)take(tiles * n_out * 4);
  w.bytes = off;
  return w;
}

int pp_check(const DofPreprocDims* d) {
  if (!d) {
    dof_set_error("dof_preprocess: dims is null");
    return DOF_ERR_ARG;
  }
  const int n_out = d->n_node_cols + d->n_edge_cols + d->n_angle_cols;
  if (d->n_frames <= 0 || d->n_videos <= 0 || d->n_cols <= 0 || d->n_animals < 0 || n_out <= 0 || d->n_node_cols < 0 ||
      d->n_edge_cols < 0 || d->n_angle_cols < 0 || d->clip < 0.0) {
    dof_set_error("dof_preprocess: bad dims");
    return DOF_ERR_ARG;
  }
  if (d->n_cols > DOF_PP_MAX_COLS || d->n_animals > DOF_PP_MAX_ANIMALS || n_out > 256 || d->n_frames >= (1ll << 31)) {
    dof_set_error("dof_preprocess: unsupported size (columns <= %d, animals <= %d, output columns <= 256, frames < 2^31)",
                  DOF_PP_MAX_COLS, DOF_PP_MAX_ANIMALS);
    return DOF_ERR_UNSUPPORTED;
  }
  for (int m : {d->speed_mode, d->dist_mode, d->coord_mode})
    if (m < DOF_PP_MODE_NONE || m > DOF_PP_MODE_GROUPWISE) {
      dof_set_error("dof_preprocess: bad standardisation mode %d", m);
      return DOF_ERR_ARG;
    }
  if (d->inter_scale < 0 || d->inter_scale > 2) {
    dof_set_error("dof_preprocess: bad inter_scale %d", d->inter_scale);
    return DOF_ERR_ARG;
  }
  return DOF_OK;
}

}  // namespace

extern "C" int64_t dof_preprocess_workspace_bytes(const DofPreprocDims* dims) {
  if (pp_check(dims) != DOF_OK) return -1;
  return pp_layout(*dims, nullptr).bytes;
}

extern "C" int dof_preprocess_tables(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                                     const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                                     const int32_t* chain, const int32_t* out_cols, const uint8_t* sample_mask,
                                     double* scaler, double* size_out, double* video_scaler, float* node_out,
                                     float* edge_out, float* angle_out, void* workspace, void* stream) {
  const int rc = pp_check(dims);
  if (rc != DOF_OK) return rc;
  const DofPreprocDims& d = *dims;
  if (!raw || !video_off || !col_kind || !chain_off || !out_cols || !scaler || !workspace || (d.n_animals > 0 && !size_ref) ||
      (d.n_node_cols > 0 && !node_out) || (d.n_edge_cols > 0 && !edge_out) || (d.n_angle_cols > 0 && !angle_out)) {
    dof_set_error("dof_preprocess_tables: null pointer argument");
    return DOF_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const PpWorkspace w = pp_layout(d, workspace);
  const int V = d.n_videos, C = d.n_cols, A = d.n_animals;
  const int n_out = d.n_node_cols + d.n_edge_cols + d.n_angle_cols;
  const unsigned strips = (unsigned)pp_slots(d.n_frames, V, PP_RS), tiles = (unsigned)pp_slots(d.n_frames, V, PP_TR);
  if (A > 0) DOF_LAUNCH(k_pp_size, (A, V), (256), st, raw, video_off, size_ref, C, A, d.n_frames, w.hyp, w.sfac);
  DOF_LAUNCH(k_pp_divisors, (V), (256), st, chain_off, chain, C, A, d.inter_scale, w.sfac, w.rdiv);
  DOF_LAUNCH(k_pp_stats, (strips), (256), st, raw, video_off, col_kind, (const double*)w.rdiv, sample_mask, V, C,
             d.log_distances, d.speed_mode, d.dist_mode, d.fit_global ? d.coord_mode : DOF_PP_MODE_NONE, w.part_all,
             w.part_smp);
  DOF_LAUNCH(k_pp_video_fin, (V), (256), st, video_off, col_kind, (const PpStat*)w.part_all, (const PpStat*)w.part_smp, C,
             d.speed_mode, d.dist_mode, w.vscale, w.ystat);
  if (d.fit_global)
    DOF_LAUNCH(k_pp_global_fin, (1), (256), st, col_kind, (const PpStat*)w.ystat, V, C, d.speed_mode, d.dist_mode,
               d.coord_mode, scaler);
  DOF_LAUNCH(k_pp_coef, (V), (256), st, (const double*)w.rdiv, (const double*)w.vscale, (const double*)scaler, C, w.coef,
             video_scaler);
  PpOutArgs oa;
  oa.raw = raw;
  oa.video_off = video_off;
  oa.col_kind = col_kind;
  oa.out_cols = out_cols;
  oa.coef = w.coef;
  oa.V = V;
  oa.C = C;
  oa.n_out = n_out;
  oa.log_dist = d.log_distances;
  oa.clip = d.clip;
  DOF_LAUNCH(k_pp_edges, (tiles), (256), st, oa, w.first_v, w.last_v);
  DOF_LAUNCH(k_pp_carry, (n_out, V), (256), st, video_off, n_out, (const int*)w.first_v, (const int*)w.last_v, w.prev_v,
             w.next_v);
#define PP_FINISH(NO, TR)                                                                                          \
  DOF_LAUNCH((k_pp_finish<NO, TR>), (tiles), (256), st, oa, (const int*)w.prev_v, (const int*)w.next_v, d.n_node_cols, \
             d.n_edge_cols, node_out, edge_out, angle_out)
  if (n_out <= 64) PP_FINISH(64, 32);
  else if (n_out <= 128) PP_FINISH(128, 32);
  else PP_FINISH(256, 16);
#undef PP_FINISH
  if (size_out)
    (void)hipMemcpyAsync(size_out, w.sfac, (size_t)V * (A + 1) * 8, hipMemcpyDeviceToDevice, st);
  return dof_check_launch("dof_preprocess_tables");
}
