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
//   k_pp_hyp      4 columns per animal: hypot(nose - tail base) as an order-preserving key per row
//                 (k_pp_size then selects the exact nan-median per (animal, video) by 8-bit radix selection)
//   k_pp_stats    one pass: shifted sums per (video, strip, column) -> (n, mean, M2), all rows and sampled rows
//   k_pp_finish   output columns only: transform, clip, interpolate inside an 8-row tile, cast, write fp32
//   k_pp_fill     closes the (rare) gaps that reach a tile edge, from the tiles' validity bytes
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
constexpr int PP_TR = 8;    // rows per output tile (one thread owns a column of it)
constexpr double PP_EPS = 2.220446049250313e-16;

struct PpStat {
  double n, mean, m2, mn, mx;  // DOF_PP_STAT_DOUBLES; mn / mx = +inf / -inf while n == 0
};
static_assert(sizeof(PpStat) == DOF_PP_STAT_DOUBLES * sizeof(double), "PpStat is the (videos, columns, 5) exchange row");
__device__ __forceinline__ PpStat pp_empty() { return PpStat{0.0, 0.0, 0.0, INFINITY, -INFINITY}; }

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
  r.mn = b.mn < a.mn ? b.mn : a.mn;
  r.mx = b.mx > a.mx ? b.mx : a.mx;
  return r;
}
__device__ __forceinline__ PpStat pp_from_sums(double n, double shift, double s1, double s2, double mn, double mx) {
  PpStat r;
  r.n = n;
  r.mean = n > 0.0 ? shift + s1 / n : 0.0;
  r.m2 = n > 0.0 ? s2 - s1 * s1 / n : 0.0;
  r.mn = n > 0.0 ? mn : INFINITY;
  r.mx = n > 0.0 ? mx : -INFINITY;
  return r;
}
// StandardScaler's (mean_, scale_) from (n, mean, M2): population variance, near-constant features get scale 1
// (sklearn _is_constant_feature / _handle_zeros_in_scale); nothing seen -> NaN like the 0/0 there
// MinMaxScaler (scale_kind 1): X * scale_ + min_ with scale_ = 1 / range, min_ = -data_min * scale_, i.e. the affine map
// (x - data_min) / range in the same (offset, divisor) form; ranges below 10 eps count as constant (divisor 1)
__device__ __forceinline__ void pp_fit(const PpStat s, int scale_kind, double* mean, double* scale) {
  if (s.n == 0.0) {
    *mean = pp_nanv();
    *scale = pp_nanv();
    return;
  }
  if (scale_kind == DOF_PP_SCALE_MINMAX) {
    const double range = s.mx - s.mn;
    *mean = s.mn;
    *scale = range < 10.0 * PP_EPS ? 1.0 : range;
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
// Exact selection without a sort: non-negative doubles order like their bit patterns, so the order statistics
// (n-1)/2 and n/2 are found digit by digit, 8 bits per sweep, from the highest byte in which the smallest and the
// largest length differ: each sweep builds a 256-bin histogram of the next byte among the values that share the
// prefix found so far (integer LDS atomics on 8 private copies: order-independent, so deterministic).  A video of
// up to 256 x 64 rows keeps its keys in registers (one workgroup per (animal, video) is latency-bound otherwise);
// longer ones sweep the L2-resident keys.
constexpr int PP_KPT = 64;
constexpr unsigned long long PP_NOKEY = ~0ull;  // a NaN bit pattern no hypot() returns: "no value"

// keys of all rows, all animals: one thread per row (the 4 reference coordinates of an animal are 4 scattered
// 8-byte reads of the row -- done once here, by as many workgroups as it takes, not by the one workgroup per
// (animal, video) of the selection below)
// keep (videos, C) bytes or null: a reference column the low-variance filter dropped in the row's video is absent there
// (scale_table then finds no size factor for that animal: "no value" keys)
__global__ void __launch_bounds__(256) k_pp_hyp(const double* __restrict__ raw, const int* __restrict__ size_ref, int C, int A,
                                                int64_t F, const int64_t* __restrict__ video_off, int V,
                                                const uint8_t* __restrict__ keep, unsigned long long* __restrict__ keys) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= F) return;
  const double* row = raw + r * C;
  const uint8_t* kv = nullptr;
  if (keep) {
    int lo = 0, hi = V - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (video_off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    kv = keep + (int64_t)lo * C;
  }
  for (int a = 0; a < A; ++a) {
    const int c0 = size_ref[4 * a], c1 = size_ref[4 * a + 1], c2 = size_ref[4 * a + 2], c3 = size_ref[4 * a + 3];
    if (c0 < 0 || c1 < 0 || c2 < 0 || c3 < 0) continue;
    const bool present = !kv || (kv[c0] && kv[c1] && kv[c2] && kv[c3]);
    const double len = hypot(row[c0] - row[c2], row[c1] - row[c3]);
    keys[(int64_t)a * F + r] = (!present || pp_isnan(len)) ? PP_NOKEY : pp_bits(len);
  }
}

__global__ void __launch_bounds__(256) k_pp_size(const int64_t* __restrict__ video_off, const int* __restrict__ size_ref,
                                                 int A, int64_t F, const unsigned long long* __restrict__ keys,
                                                 double* __restrict__ s_out) {
  __shared__ int hist[2][8][256];
  __shared__ int red[256];
  __shared__ unsigned long long mm[2][256];
  __shared__ unsigned long long sel[2];
  __shared__ int krem[2];
  const int a = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
  const int c0 = size_ref[4 * a], c1 = size_ref[4 * a + 1], c2 = size_ref[4 * a + 2], c3 = size_ref[4 * a + 3];
  double* out = s_out + (int64_t)v * (A + 1) + a;
  if (c0 < 0 || c1 < 0 || c2 < 0 || c3 < 0) {
    if (tid == 0) *out = pp_nanv();
    return;
  }
  const int64_t r0 = video_off[v], r1 = video_off[v + 1];
  const bool in_regs = r1 - r0 <= 256 * PP_KPT;
  const unsigned long long* h = keys + (int64_t)a * F;
  unsigned long long key[PP_KPT];
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < PP_KPT; ++i) {
      const int64_t r = r0 + tid + 256 * i;
      key[i] = r < r1 ? h[r] : PP_NOKEY;
    }
  }
  // f(key) for every key of this thread
  auto sweep = [&](auto f) {
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < PP_KPT; ++i)
        if (key[i] != PP_NOKEY) f(key[i]);
    } else {
      for (int64_t rb = r0 + tid; rb < r1; rb += 256 * 8) {
        unsigned long long k8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) k8[i] = rb + 256 * i < r1 ? h[rb + 256 * i] : PP_NOKEY;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (k8[i] != PP_NOKEY) f(k8[i]);
      }
    }
  };
  int nv = 0;
  unsigned long long kmin = PP_NOKEY, kmax = 0ull;
  sweep([&](unsigned long long k) {
    ++nv;
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
  });
  red[tid] = nv;
  mm[0][tid] = kmin;
  mm[1][tid] = kmax;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[tid] += red[tid + s];
      mm[0][tid] = mm[0][tid + s] < mm[0][tid] ? mm[0][tid + s] : mm[0][tid];
      mm[1][tid] = mm[1][tid + s] > mm[1][tid] ? mm[1][tid + s] : mm[1][tid];
    }
    __syncthreads();
  }
  const int n = red[0];
  if (n == 0) {
    if (tid == 0) *out = pp_nanv();
    return;
  }
  // every key shares the bits above the highest bit in which the smallest and the largest key differ
  const unsigned long long diff = mm[0][0] ^ mm[1][0];
  int top = 0;  // shift of the highest byte that differs
  while (top < 56 && (diff >> (top + 8)) != 0ull) top += 8;
  if (tid == 0) {
    sel[0] = sel[1] = top == 56 ? 0ull : mm[0][0] & (~0ull << (top + 8));
    krem[0] = (n - 1) / 2;
    krem[1] = n / 2;
  }
  const int copy = (tid >> 3) & 7;  // 8 private copies of each histogram spread the hot bins
  for (int shift = top; shift >= 0; shift -= 8) {
    for (int i = tid; i < 2 * 8 * 256; i += 256) (&hist[0][0][0])[i] = 0;
    __syncthreads();
    const unsigned long long p0 = sel[0], p1 = sel[1];
    const bool twin = p0 != p1;  // the two order statistics parted ways: they need their own histograms
    const unsigned long long himask = shift == 56 ? 0ull : ~0ull << (shift + 8);
    sweep([&](unsigned long long k) {
      const int digit = (int)((k >> shift) & 255);
      if ((k & himask) == p0) atomicAdd(&hist[0][copy][digit], 1);
      if (twin && (k & himask) == p1) atomicAdd(&hist[1][copy][digit], 1);
    });
    __syncthreads();
    int t0 = 0, t1 = 0;
    for (int k = 0; k < 8; ++k) {
      t0 += hist[0][k][tid];
      t1 += hist[1][k][tid];
    }
    if (!twin) t1 = t0;
    __syncthreads();
    // inclusive prefix sums over the 256 digits (both statistics at once); the digit whose running count first
    // exceeds the remaining rank is the next byte
    hist[0][0][tid] = t0;
    hist[1][0][tid] = t1;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int u0 = tid >= off ? hist[0][0][tid - off] : 0, u1 = tid >= off ? hist[1][0][tid - off] : 0;
      __syncthreads();
      hist[0][0][tid] += u0;
      hist[1][0][tid] += u1;
      __syncthreads();
    }
    const int k0 = krem[0], k1 = krem[1];
    const int inc0 = hist[0][0][tid], inc1 = hist[1][0][tid];
    __syncthreads();
    if (inc0 - t0 <= k0 && k0 < inc0) {
      krem[0] = k0 - (inc0 - t0);
      sel[0] |= (unsigned long long)tid << shift;
    }
    if (inc1 - t1 <= k1 && k1 < inc1) {
      krem[1] = k1 - (inc1 - t1);
      sel[1] |= (unsigned long long)tid << shift;
    }
    __syncthreads();
  }
  if (tid == 0) *out = (pp_from_bits(sel[0]) + pp_from_bits(sel[1])) / 2.0;
}

// default factor (median of the usable factors, else 1), substitution of unusable factors, and the reciprocal
// size divisor of every column of every video
__global__ void __launch_bounds__(256) k_pp_divisors(const int* __restrict__ chain_off, const int* __restrict__ chain,
                                                     int C, int A, int inter_scale, const uint8_t* __restrict__ keep,
                                                     double* __restrict__ s_out, double* __restrict__ rdiv) {
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
      const int a1 = chain[4 * e], a2 = chain[4 * e + 1], same = chain[4 * e + 2], src = chain[4 * e + 3];
      if (keep && !keep[(int64_t)v * C + src]) continue;  // the column this division comes from is absent in this video
      const double s1 = S[a1 < 0 ? A : a1], s2 = S[a2 < 0 ? A : a2];
      double s = s1;
      if (!same) s = inter_scale == 0 ? 0.5 * (s1 + s2) : inter_scale == 1 ? sqrt(s1 * s2) : S[A];
      d *= s;
    }
    rdiv[(int64_t)v * C + c] = 1.0 / d;
  }
}

// ---------------------------------------------------------------------------------------------------------
// slot -> video table (one binary search per slot here instead of one per workgroup in every pass); the tile
// table also carries the tile's first row and row count: (video or -1, global row, row within the video, rows)
__device__ __forceinline__ void pp_slot_table(int64_t b, const int64_t* __restrict__ video_off, int V, int R, int64_t n_slots,
                                              int* __restrict__ slot_v, int* __restrict__ slot_rec) {
  if (b >= n_slots) return;
  int v;
  int64_t k;
  const bool live = pp_locate(video_off, V, R, b, &v, &k);
  if (slot_v) slot_v[b] = live ? v : -1;
  if (slot_rec) {
    const int64_t voff = video_off[v], left = video_off[v + 1] - voff - k * R;
    slot_rec[4 * b] = live ? v : -1;
    slot_rec[4 * b + 1] = (int)(voff + k * R);
    slot_rec[4 * b + 2] = (int)(k * R);
    slot_rec[4 * b + 3] = live ? (int)(left < R ? left : R) : 0;
  }
}

// Column chunks: runs of <= 64 consecutive columns (of the raw table, or of the output column list `index`) that
// agree on "takes log1p", so a wavefront never pays for the logarithm on behalf of a few lanes.
// chunks[0] = count, then (start, n, log) triples.  One thread per column, two Hillis-Steele scans in LDS
// (run start = prefix max of the flip positions, chunk number = prefix sum of the chunk starts).
__device__ __forceinline__ void pp_chunks(const int* __restrict__ col_kind, const int* __restrict__ index, int n_cols,
                                          int log_dist, int* __restrict__ chunks) {
  __shared__ int lg[DOF_PP_MAX_COLS], a[DOF_PP_MAX_COLS], b[DOF_PP_MAX_COLS];
  const int c = threadIdx.x;
  int mine = 0;
  if (c < n_cols) {
    const int k = col_kind[index ? index[c] : c];
    mine = log_dist && (k == DOF_PP_DIST_INNER || k == DOF_PP_DIST_INTRA);
  }
  lg[c] = mine;
  __syncthreads();
  a[c] = (c < n_cols && (c == 0 || lg[c - 1] != mine)) ? c : -1;  // run starts
  __syncthreads();
  for (int off = 1; off < DOF_PP_MAX_COLS; off <<= 1) {
    const int o = c >= off ? a[c - off] : -1;
    __syncthreads();
    if (o > a[c]) a[c] = o;
    __syncthreads();
  }
  const int run0 = a[c];
  const int start = c < n_cols && ((c - run0) & 63) == 0;  // a chunk starts every 64 columns of a run
  b[c] = start;
  __syncthreads();
  for (int off = 1; off < DOF_PP_MAX_COLS; off <<= 1) {
    const int o = c >= off ? b[c - off] : 0;
    __syncthreads();
    b[c] += o;
    __syncthreads();
  }
  if (c < n_cols) {
    // chunk length: up to the next chunk start or the end of the run
    if (start) {
      int e = c + 1;
      while (e < n_cols && e - c < 64 && lg[e] == mine) ++e;
      const int n = b[c] - 1;
      chunks[1 + 3 * n] = c;
      chunks[2 + 3 * n] = e - c;
      chunks[3 + 3 * n] = mine;
    }
    if (c == n_cols - 1) chunks[0] = b[c];
  }
}

// all index tables of a call in one launch: strip slots, tile slots, raw-column chunks, output-column chunks
__global__ void __launch_bounds__(DOF_PP_MAX_COLS) k_pp_setup(const int64_t* __restrict__ video_off, int V, int64_t strips,
                                                              int64_t tiles, const int* __restrict__ col_kind,
                                                              const int* __restrict__ out_cols, int C, int n_out,
                                                              int log_dist, int* __restrict__ strip_v,
                                                              int* __restrict__ tile_rec, int* __restrict__ chunks,
                                                              int* __restrict__ ochunks) {
  const int64_t nb_s = (strips + DOF_PP_MAX_COLS - 1) / DOF_PP_MAX_COLS, nb_t = (tiles + DOF_PP_MAX_COLS - 1) / DOF_PP_MAX_COLS;
  const int64_t blk = blockIdx.x;
  if (blk < nb_s) pp_slot_table(blk * DOF_PP_MAX_COLS + threadIdx.x, video_off, V, PP_RS, strips, strip_v, nullptr);
  else if (blk < nb_s + nb_t)
    pp_slot_table((blk - nb_s) * DOF_PP_MAX_COLS + threadIdx.x, video_off, V, PP_TR, tiles, nullptr, tile_rec);
  else if (blk == nb_s + nb_t) pp_chunks(col_kind, nullptr, C, log_dist, chunks);
  else pp_chunks(col_kind, out_cols, n_out, log_dist, ochunks);
}

// Reference-accuracy logarithm of c in [1, 2] (argument reduction s = f / (2 + f), fdlibm's e_log minimax
// coefficients, < 1 ulp); only used to fill the 128-entry table below.
__device__ __forceinline__ double pp_log_1to2(double c) {
  double m = c;
  double dk = 0.0;
  if (m > 1.4142135623730951) {
    m *= 0.5;
    dk = 1.0;
  }
  const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 +
                         w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double r = t1 + t2, hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + r) + dk * 1.90821492927058770002e-10)) - f);
}
// tab[i] = (1 / c_i, log c_i), c_i = 1 + (i + 1/2) / 128: call with all threads of the block, then __syncthreads()
__device__ __forceinline__ void pp_log_table_init(double (*tab)[2]) {
  for (int i = threadIdx.x; i < 128; i += blockDim.x) {
    const double c = 1.0 + ((double)i + 0.5) / 128.0;
    tab[i][0] = 1.0 / c;
    tab[i][1] = pp_log_1to2(c);
  }
}
// log1p for x >= 0 (negative distances were clamped): log(u) + (x - (u - 1)) / u with u = fl(1 + x) = 2^k m, m in
// [1, 2): the top 7 mantissa bits pick c_i from the table, r = m / c_i - 1 (|r| <= 2^-8, one fma), log(1 + r) is a
// degree-7 Taylor polynomial (remainder < 2^-67), log u = k ln2 + log c_i + log(1 + r).  Absolute error ~1e-16, a
// third of the instructions of the general-purpose library routine, which made the statistics pass compute-bound.
__device__ __forceinline__ double pp_log1p(double x, const double (*tab)[2]) {
  if (!(x < INFINITY)) return x;  // +inf or NaN
  const double u = 1.0 + x;
  const double c = x - (u - 1.0);
  const uint64_t bits = pp_bits(u);
  const unsigned hi = (unsigned)(bits >> 32);
  const double dk = (double)((int)(hi >> 20) - 1023);
  const int i = (int)((hi >> 13) & 127u);
  const double m = pp_from_bits((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  const double r = fma(m, tab[i][0], -1.0);
  // log(1 + r) = r (1 - r/2 + r^2/3 - ... + r^6/7), Estrin's scheme (three short dependent steps instead of seven)
  const double r2 = r * r, r4 = r2 * r2;
  const double p01 = fma(r, -0.5, 1.0), p23 = fma(r, -1.0 / 4.0, 1.0 / 3.0), p45 = fma(r, -1.0 / 6.0, 1.0 / 5.0);
  const double p = fma(r4, fma(r2, 1.0 / 7.0, p45), fma(r2, p23, p01));
  const double lg = fma(dk, 6.93147180369123816490e-01, tab[i][1] + fma(r, p, dk * 1.90821492927058770002e-10));
  const double res = lg + c * (double)(1.0f / (float)u);
  return x == 0.0 ? 0.0 : res;
}

// ---------------------------------------------------------------------------------------------------------
// The statistics pass.  A workgroup = one strip of PP_RS rows of one video; lanes run along the columns of a chunk
// (a row is one contiguous 8C-byte run, so a wavefront reads up to 512 contiguous bytes).  A chunk narrower than
// 32 columns is packed 2x / 4x ... along the rows so that no wavefront idles most of its lanes; the 4 wavefronts
// interleave the remaining rows, 8 independent loads in flight per lane.  Per-thread sums are taken about the
// first value seen (no cancellation), turned into (n, mean, M2) and merged in LDS in a fixed order; the strip's
// result is a partial for the fixed-order finalize.
template <bool MASKED>
__global__ void __launch_bounds__(256) k_pp_stats(const double* __restrict__ raw, const int64_t* __restrict__ video_off,
                                                  const int* __restrict__ slot_v, const int* __restrict__ chunks,
                                                  const int* __restrict__ col_kind, const double* __restrict__ rdiv,
                                                  const uint8_t* __restrict__ mask, int C, int speed_mode, int dist_mode,
                                                  int coord_mode, int all_cols, const uint8_t* __restrict__ keep,
                                                  PpStat* __restrict__ part_all, PpStat* __restrict__ part_smp) {
  __shared__ PpStat sh[2][4][64];
  __shared__ double logtab[128][2];
  const int v = slot_v[blockIdx.x];
  if (v < 0) return;
  pp_log_table_init(logtab);
  __syncthreads();
  const int64_t voff = video_off[v], vend = video_off[v + 1];
  const int64_t r0 = voff + ((int64_t)blockIdx.x - (voff / PP_RS + v)) * PP_RS;
  const int64_t r1 = r0 + PP_RS < vend ? r0 + PP_RS : vend;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int nchunk = chunks[0];
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cbase = chunks[1 + 3 * ch], ccount = chunks[2 + 3 * ch], lg = chunks[3 + 3 * ch];
    int width = 64;  // lanes per row: smallest power of two >= ccount
    while (width / 2 >= ccount && width > 1) width >>= 1;
    const int pack = 64 / width, sub = lane / width, cl = lane - sub * width;
    const int c = cbase + cl;
    // two interleaved accumulator sets (shorter dependency chains); sums are taken about `shift`, the first value
    // this thread sees (fixed from then on, so no element waits for the previous one)
    double n[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0}, nb[2] = {0.0, 0.0}, s1b[2] = {0.0, 0.0},
           s2b[2] = {0.0, 0.0};
    double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY}, lob[2] = {INFINITY, INFINITY},
           hib[2] = {-INFINITY, -INFINITY};
    double shift = 0.0;
    bool have_shift = false;
    // all_cols: moments of the raw values of every column (the low-variance filter's input)
    const bool mine = cl < ccount && (all_cols || pp_mode(col_kind[c], speed_mode, dist_mode, coord_mode) != DOF_PP_MODE_NONE) &&
                      (!keep || keep[(int64_t)v * C + c]);
    if (mine) {
      const double rd = all_cols ? 1.0 : rdiv[(int64_t)v * C + c];
      const int rstep = 4 * pack;
      for (int64_t rb = r0 + rg * pack + sub; rb < r1; rb += 8 * rstep) {
        double x[8];
        bool in[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t r = rb + rstep * i;
          in[i] = MASKED && r < r1 && mask[r];
          x[i] = r < r1 ? raw[r * C + c] : pp_nanv();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          double u = x[i] * rd;
          if (lg && !all_cols) {
            if (u < 0.0) u = 0.0;
            u = pp_log1p(u, logtab);
          }
          x[i] = u;
        }
        if (!have_shift) {
#pragma unroll
          for (int i = 7; i >= 0; --i)
            if (!pp_isnan(x[i])) {
              shift = x[i];
              have_shift = true;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool ok = !pp_isnan(x[i]);
          const double d = ok ? x[i] - shift : 0.0;
          n[i & 1] += ok ? 1.0 : 0.0;
          s1[i & 1] += d;
          s2[i & 1] = fma(d, d, s2[i & 1]);
          lo[i & 1] = ok && x[i] < lo[i & 1] ? x[i] : lo[i & 1];
          hi[i & 1] = ok && x[i] > hi[i & 1] ? x[i] : hi[i & 1];
          if (MASKED) {
            const bool smp = ok && in[i];
            nb[i & 1] += smp ? 1.0 : 0.0;
            s1b[i & 1] += smp ? d : 0.0;
            s2b[i & 1] += smp ? d * d : 0.0;
            lob[i & 1] = smp && x[i] < lob[i & 1] ? x[i] : lob[i & 1];
            hib[i & 1] = smp && x[i] > hib[i & 1] ? x[i] : hib[i & 1];
          }
        }
      }
    }
    sh[0][rg][lane] = pp_from_sums(n[0] + n[1], shift, s1[0] + s1[1], s2[0] + s2[1], fmin(lo[0], lo[1]), fmax(hi[0], hi[1]));
    sh[1][rg][lane] = MASKED ? pp_from_sums(nb[0] + nb[1], shift, s1b[0] + s1b[1], s2b[0] + s2b[1], fmin(lob[0], lob[1]),
                                            fmax(hib[0], hib[1]))
                             : sh[0][rg][lane];
    __syncthreads();
    if (rg < 2 && lane < ccount) {
      PpStat t = pp_empty();
      for (int g = 0; g < 4; ++g)
        for (int k = 0; k < pack; ++k) t = pp_merge(t, sh[rg][g][k * width + lane]);
      (rg == 0 ? part_all : part_smp)[(int64_t)blockIdx.x * C + c] = t;
    }
    __syncthreads();
  }
}

// one of 4 interleaved sub-sequences (rg) of `count` partials (stride in PpStat units), merged in a fixed order,
// 8 loads in flight
__device__ __forceinline__ PpStat pp_merge_run(const PpStat* __restrict__ base, int64_t count, int64_t stride, int rg) {
  PpStat acc = pp_empty();
  for (int64_t k = rg; k < count; k += 32) {
    PpStat t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t kk = k + 4 * i;
      if (kk < count) t[i] = base[kk * stride]; else t[i].n = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = pp_merge(acc, t[i]);
  }
  return acc;
}

// strips -> (video, column) statistics, all rows and sampled rows: one workgroup per (64-column chunk, video)
__global__ void __launch_bounds__(256) k_pp_video_cols(const int64_t* __restrict__ video_off,
                                                       const PpStat* __restrict__ part_all,
                                                       const PpStat* __restrict__ part_smp, int C,
                                                       PpStat* __restrict__ vcol_all, PpStat* __restrict__ vcol_smp) {
  __shared__ PpStat sh[2][4][64];
  const int v = blockIdx.y, lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t slot0 = video_off[v] / PP_RS + v;
  const int64_t nstrip = (video_off[v + 1] - video_off[v] + PP_RS - 1) / PP_RS;
  PpStat a = pp_empty(), s = pp_empty();
  if (c < C) {
    a = pp_merge_run(part_all + slot0 * C + c, nstrip, C, rg);
    s = pp_merge_run(part_smp + slot0 * C + c, nstrip, C, rg);
  }
  sh[0][rg][lane] = a;
  sh[1][rg][lane] = s;
  __syncthreads();
  if (rg < 2 && c < C) {
    PpStat t = sh[rg][0][lane];
    for (int g = 1; g < 4; ++g) t = pp_merge(t, sh[rg][g][lane]);
    (rg == 0 ? vcol_all : vcol_smp)[(int64_t)v * C + c] = t;
  }
}

// statistics of column groups: wavefront w merges the columns of kind kind0 + w (its 64 lanes stride over the
// columns, then a 6-step tree in LDS); fixed order, run-to-run deterministic.  tree: [4][64] scratch.
__device__ __forceinline__ void pp_group_stats(const PpStat* __restrict__ col, const int* __restrict__ kinds, int C,
                                               int kind0, int n_groups, PpStat (*tree)[64], PpStat* __restrict__ grp) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  PpStat acc = pp_empty();
  if (w < n_groups)
    for (int c = lane; c < C; c += 64)
      if (kinds[c] == kind0 + w) acc = pp_merge(acc, col[c]);
  tree[w][lane] = acc;
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if (lane < s) tree[w][lane] = pp_merge(tree[w][lane], tree[w][lane + s]);
    __syncthreads();
  }
  if (lane == 0 && w < n_groups) grp[w] = tree[w][0];
  __syncthreads();
}

// per video: columns -> groups; per-video (mean, scale); statistics of the per-video-standardised sampled rows
// (what the global scalers are fitted on)
__global__ void __launch_bounds__(256) k_pp_video_fin(const int* __restrict__ col_kind,
                                                      const PpStat* __restrict__ vcol_all,
                                                      const PpStat* __restrict__ vcol_smp, int C, int speed_mode,
                                                      int dist_mode, int scale_kind, double* __restrict__ vscale,
                                                      PpStat* __restrict__ ystat) {
  __shared__ PpStat col_all[DOF_PP_MAX_COLS];
  __shared__ int kinds[DOF_PP_MAX_COLS];
  __shared__ PpStat tree[4][64];
  __shared__ PpStat grp[4];
  const int v = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    col_all[c] = vcol_all[(int64_t)v * C + c];
    kinds[c] = col_kind[c];
  }
  __syncthreads();
  pp_group_stats(col_all, kinds, C, DOF_PP_SPEED, 3, tree, grp);  // speed, inner, intra
  for (int c = threadIdx.x; c < C; c += 256) {
    const int kind = kinds[c];
    const int mode = kind == DOF_PP_SPEED ? speed_mode
                     : (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA) ? dist_mode : DOF_PP_MODE_NONE;
    double m = 0.0, s = 1.0;
    if (mode == DOF_PP_MODE_PER_COLUMN) pp_fit(col_all[c], scale_kind, &m, &s);
    if (mode == DOF_PP_MODE_GROUPWISE) pp_fit(grp[kind - DOF_PP_SPEED], scale_kind, &m, &s);
    vscale[((int64_t)v * C + c) * 2] = m;
    vscale[((int64_t)v * C + c) * 2 + 1] = s;
    PpStat y = vcol_smp[(int64_t)v * C + c];
    y.mean = (y.mean - m) / s;
    y.m2 = y.m2 / (s * s);
    y.mn = (y.mn - m) / s;  // s > 0: the extremes stay the extremes
    y.mx = (y.mx - m) / s;
    if (y.n == 0.0 || pp_isnan(y.mean) || pp_isnan(y.m2)) y = pp_empty();
    ystat[(int64_t)v * C + c] = y;
  }
}

// videos -> per-column statistics of the sampled, per-video-standardised rows: one workgroup per 64 columns
__global__ void __launch_bounds__(256) k_pp_global_cols(const PpStat* __restrict__ ystat, int V, int C,
                                                        PpStat* __restrict__ gcol) {
  __shared__ PpStat sh[4][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  PpStat a = pp_empty();
  if (c < C) a = pp_merge_run(ystat + c, V, C, rg);
  sh[rg][lane] = a;
  __syncthreads();
  if (rg == 0 && c < C) {
    PpStat t = sh[0][lane];
    for (int g = 1; g < 4; ++g) t = pp_merge(t, sh[g][lane]);
    gcol[c] = t;
  }
}

// the global scalers alone, from the (n, mean, M2) rows of ALL videos (one workgroup; the multi-GPU path gathers
// every rank's rows and runs this on each rank): same merge order as k_pp_global_cols + k_pp_coef, so the scalers
// are bit-identical to those of a single-GPU call over the same videos
__global__ void __launch_bounds__(256) k_pp_fit_global(const int* __restrict__ col_kind, const PpStat* __restrict__ ystat, int V,
                                                       int C, int speed_mode, int dist_mode, int coord_mode, int scale_kind,
                                                       double* __restrict__ scaler) {
  __shared__ PpStat col[DOF_PP_MAX_COLS];
  __shared__ int kinds[DOF_PP_MAX_COLS];
  __shared__ PpStat tree[4][64];
  __shared__ PpStat grp[4];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int cbase = 0; cbase < C; cbase += 64) {
    const int c = cbase + lane;
    PpStat a = pp_empty();
    if (c < C) a = pp_merge_run(ystat + c, V, C, rg);
    tree[rg][lane] = a;
    __syncthreads();
    if (rg == 0 && c < C) {
      PpStat t = tree[0][lane];
      for (int g = 1; g < 4; ++g) t = pp_merge(t, tree[g][lane]);
      col[c] = t;
      kinds[c] = col_kind[c];
    }
    __syncthreads();
  }
  pp_group_stats(col, kinds, C, DOF_PP_COORD, 4, tree, grp);
  for (int c = threadIdx.x; c < C; c += 256) {
    const int mode = pp_mode(kinds[c], speed_mode, dist_mode, coord_mode);
    double gm = 0.0, gs = 1.0;
    if (mode != DOF_PP_MODE_NONE)
      pp_fit(mode == DOF_PP_MODE_PER_COLUMN ? col[c] : grp[kinds[c] - DOF_PP_COORD], scale_kind, &gm, &gs);
    scaler[2 * c] = gm;
    scaler[2 * c + 1] = gs;
  }
}

// per video: the global scalers (columns -> groups coord / speed / inner / intra; every workgroup derives the same
// values, workgroup 0 publishes them) and the coefficients of the complete element transform
//   u = x * cf[0] [log1p(max(u, 0))];  z = u * cf[1] + cf[2]
__global__ void __launch_bounds__(256) k_pp_coef(const int* __restrict__ col_kind, const PpStat* __restrict__ gcol,
                                                 const double* __restrict__ rdiv, const double* __restrict__ vscale,
                                                 double* __restrict__ scaler, int fit_global, int C, int speed_mode,
                                                 int dist_mode, int coord_mode, int scale_kind,
                                                 const uint8_t* __restrict__ keep, double* __restrict__ coef,
                                                 double* __restrict__ video_scaler) {
  __shared__ PpStat col[DOF_PP_MAX_COLS];
  __shared__ int kinds[DOF_PP_MAX_COLS];
  __shared__ PpStat tree[4][64];
  __shared__ PpStat grp[4];
  const int v = blockIdx.x;
  if (fit_global) {
    for (int c = threadIdx.x; c < C; c += 256) {
      col[c] = gcol[c];
      kinds[c] = col_kind[c];
    }
    __syncthreads();
    pp_group_stats(col, kinds, C, DOF_PP_COORD, 4, tree, grp);  // coord, speed, inner, intra
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    double gm, gs;
    if (fit_global) {
      const int kind = kinds[c];
      const int mode = pp_mode(kind, speed_mode, dist_mode, coord_mode);
      gm = 0.0;
      gs = 1.0;
      // nothing sampled -> NaN like the reference's 0/0 (such columns hold no value anyway)
      if (mode != DOF_PP_MODE_NONE)
        pp_fit(mode == DOF_PP_MODE_PER_COLUMN ? col[c] : grp[kind - DOF_PP_COORD], scale_kind, &gm, &gs);
      if (v == 0) {
        scaler[2 * c] = gm;
        scaler[2 * c + 1] = gs;
      }
    } else {
      gm = scaler[2 * c];
      gs = scaler[2 * c + 1];
    }
    const int64_t i = (int64_t)v * C + c;
    const double m = vscale[2 * i], s = vscale[2 * i + 1];
    const double p = 1.0 / s, q = 1.0 / gs;
    // a column the low-variance filter dropped in this video comes back as zeros (the reference re-inserts it as a
    // missing column, which _pp_sanitize_numeric fills with 0): u = x * 0, z = u * 0 + 0, gaps interpolate between zeros
    const bool gone = keep && !keep[i];
    coef[3 * i] = gone ? 0.0 : rdiv[i];
    coef[3 * i + 1] = gone ? 0.0 : p * q;
    coef[3 * i + 2] = gone ? 0.0 : -(m * p * q + gm * q);
    if (video_scaler) {
      video_scaler[2 * i] = m;
      video_scaler[2 * i + 1] = s;
    }
  }
}

struct PpOutArgs {
  const double* raw;
  const int64_t* video_off;
  const int* slot_rec;  // (video or -1, global row, row within the video, rows) per PP_TR-row tile slot
  const int* col_kind;
  const int* out_cols;
  const double* coef;
  int64_t n_slots;
  int C, n_out, n_node, n_edge, log_dist;
  double clip;
  float *node_out, *edge_out, *angle_out;
};

// final value of raw element x of column c (coefficients cf); NaN when missing or clipped
__device__ __forceinline__ double pp_value(double x, double cf0, double cf1, double cf2, bool lg, bool clipk, double clip,
                                           const double (*tab)[2]) {
  double u = x * cf0;
  if (lg) {
    if (u < 0.0) u = 0.0;
    u = pp_log1p(u, tab);
  }
  const double z = u * cf1 + cf2;
  return (clipk && fabs(z) > clip) ? pp_nanv() : z;
}

#define PP_MUL_RN(a, b) dof_dmul_rn((a), (b))
#define PP_ADD_RN(a, b) dof_dadd_rn((a), (b))
// numpy.interp between valid rows p < q (values pv, qv): slope * (x - x0) + y0, multiply and add rounded separately;
// flat beyond the first / last valid row of the video, 0 for a column without any valid row
__device__ __forceinline__ double pp_interp(int64_t row, int64_t p, double pv, int64_t q, double qv) {
  if (p >= 0 && q >= 0) {
    const double slope = (qv - pv) / (double)(q - p);
    return PP_ADD_RN(PP_MUL_RN(slope, (double)(row - p)), pv);
  }
  return p >= 0 ? pv : q >= 0 ? qv : 0.0;
}
// column j of the concatenated output -> (table, width, column inside the table)
__device__ __forceinline__ float* pp_out_column(const PpOutArgs& A, int j, int* width) {
  if (j < A.n_node) {
    *width = A.n_node;
    return A.node_out + j;
  }
  if (j < A.n_node + A.n_edge) {
    *width = A.n_edge;
    return A.edge_out + (j - A.n_node);
  }
  *width = A.n_out - A.n_node - A.n_edge;
  return A.angle_out + (j - A.n_node - A.n_edge);
}

// The output pass: pure streaming, no LDS staging, no barrier after the table set-up.  A thread owns one output
// column of one PP_TR(=8)-row tile: 8 independent loads down the column, transform + clip, fp32 stores -- lanes run
// along the output columns, so both the loads (inside one 8C-byte raw row) and the stores (inside one frame-table
// row) of a wavefront touch one contiguous region per row.  A wavefront walks 8 consecutive tiles (64 rows); chunks
// narrower than 32 columns (typically the log1p'd edge columns) take 2 / 4 / 8 tiles at a time.  A tile column with
// missing values closes the gaps that have both neighbours inside the tile from its registers and leaves a validity
// byte (bit r = row r valid) for k_pp_fill, which closes the gaps that reach a tile edge; the memset value 0xFF
// means "all rows valid", so the common case writes no note at all.
__global__ void __launch_bounds__(256) k_pp_finish(PpOutArgs A, const int* __restrict__ ochunks,
                                                   uint8_t* __restrict__ notes) {
  __shared__ double logtab[128][2];
  if (A.log_dist) {
    pp_log_table_init(logtab);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t group0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;  // first of this wavefront's 8 slots
  if (group0 >= A.n_slots) return;
  const int nchunk = ochunks[0];
  for (int ch = 0; ch < nchunk; ++ch) {
    const int jbase = ochunks[1 + 3 * ch], count = ochunks[2 + 3 * ch];
    const bool lg = ochunks[3 + 3 * ch] != 0;
    int width = 64;
    while (width / 2 >= count && width > 8) width >>= 1;
    const int pack = 64 / width, sub = lane / width, cl = lane - sub * width;
    if (cl >= count) continue;
    const int j = jbase + cl, c = A.out_cols[j], kind = A.col_kind[c];
    const bool clipk = A.clip > 0.0 && kind >= DOF_PP_COORD && kind <= DOF_PP_DIST_INTRA;
    int wdt;
    float* const out_col = pp_out_column(A, j, &wdt);
    int v_have = -1;
    double cf0 = 0.0, cf1 = 0.0, cf2 = 0.0;
    for (int it = 0; it < 8; it += pack) {
      const int64_t slot = group0 + it + sub;
      if (slot >= A.n_slots) break;
      const int v = A.slot_rec[4 * slot], nrows = A.slot_rec[4 * slot + 3];
      if (v < 0) continue;
      const int64_t grow = A.slot_rec[4 * slot + 1], t0 = A.slot_rec[4 * slot + 2];
      if (v != v_have) {
        const double* cf = A.coef + ((int64_t)v * A.C + c) * 3;
        cf0 = cf[0];
        cf1 = cf[1];
        cf2 = cf[2];
        v_have = v;
      }
      const double* src = A.raw + grow * A.C + c;
      double z[PP_TR];
#pragma unroll
      for (int r = 0; r < PP_TR; ++r) z[r] = r < nrows ? src[(int64_t)r * A.C] : 0.0;
      unsigned valid = 0;
#pragma unroll
      for (int r = 0; r < PP_TR; ++r) {
        z[r] = pp_value(z[r], cf0, cf1, cf2, lg, clipk, A.clip, logtab);
        valid |= (r >= nrows || !pp_isnan(z[r])) ? 1u << r : 0u;
      }
      if (valid != 0xffu) {
        // gaps with both neighbours inside the tile: forward pass = nearest valid row before, backward pass = after
        int before[PP_TR];
        double before_v[PP_TR];
        int p = -1;
        double pv = 0.0;
#pragma unroll
        for (int r = 0; r < PP_TR; ++r) {
          before[r] = p;
          before_v[r] = pv;
          if ((valid >> r) & 1u) {
            p = r;
            pv = z[r];
          }
        }
        int q = -1;
        double qv = 0.0;
#pragma unroll
        for (int r = PP_TR - 1; r >= 0; --r) {
          if ((valid >> r) & 1u) {
            if (r < nrows) {
              q = r;
              qv = z[r];
            }
          } else if (before[r] >= 0 && q >= 0) {
            z[r] = pp_interp(t0 + r, t0 + before[r], before_v[r], t0 + q, qv);
          }
        }
        notes[(int64_t)j * A.n_slots + slot] = (uint8_t)valid;
      }
      float* dst = out_col + grow * wdt;
#pragma unroll
      for (int r = 0; r < PP_TR; ++r)
        if (r < nrows) dst[(int64_t)r * wdt] = (float)z[r];
    }
  }
}

// Gaps that touch a tile edge, per (output column, video): the nearest valid row before / after a tile comes from
// the tiles' validity bytes (prefix max of the last valid row / suffix min of the first: a 256-wide scan in LDS over
// per-thread chunks of tiles, then short walks inside the chunk); every tile closes its own part of a gap with the
// same (p, q) pair, so a gap spanning many tiles is filled in parallel and with the arithmetic of one numpy.interp
// call.
__global__ void __launch_bounds__(256) k_pp_fill(PpOutArgs A, const uint8_t* __restrict__ notes) {
  __shared__ int sh_max[256], sh_min[256];
  __shared__ double logtab[128][2];
  if (A.log_dist) pp_log_table_init(logtab);
  const int j = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
  const int64_t voff = A.video_off[v], vlen = A.video_off[v + 1] - voff;
  const int64_t nt = (vlen + PP_TR - 1) / PP_TR;
  const uint8_t* N = notes + (int64_t)j * A.n_slots + (voff / PP_TR + v);
  const int64_t chunk = (nt + 255) / 256;
  const int64_t a = tid * chunk < nt ? tid * chunk : nt, b = a + chunk < nt ? a + chunk : nt;
  const int none = 0x7fffffff;
  // validity bits of tile t without the padding bits of the video's last tile
  auto bits_of = [&](int64_t t) -> unsigned {
    const int64_t left = vlen - t * PP_TR;
    return left >= PP_TR ? (unsigned)N[t] : (unsigned)N[t] & ((1u << left) - 1u);
  };
  int mx = -1, mn = none;
  bool work = false;
  for (int64_t t = a; t < b; ++t) {
    if (N[t] != 0xffu) work = true;
    const unsigned m = bits_of(t);
    if (m != 0u) {
      const int l = (int)(t * PP_TR) + (31 - __builtin_clz(m)), f = (int)(t * PP_TR) + __builtin_ctz(m);
      if (l > mx) mx = l;
      if (f < mn) mn = f;
    }
  }
  sh_max[tid] = mx;
  sh_min[tid] = mn;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int m = tid >= off ? sh_max[tid - off] : -1;
    const int n = tid + off < 256 ? sh_min[tid + off] : none;
    __syncthreads();
    if (m > sh_max[tid]) sh_max[tid] = m;
    if (n < sh_min[tid]) sh_min[tid] = n;
    __syncthreads();
  }
  if (!work) return;
  const int carry = tid > 0 ? sh_max[tid - 1] : -1;
  const int back = tid < 255 ? sh_min[tid + 1] : none;
  const int c = A.out_cols[j], kind = A.col_kind[c];
  const double* cf = A.coef + ((int64_t)v * A.C + c) * 3;
  const double cf0 = cf[0], cf1 = cf[1], cf2 = cf[2];
  const bool lg = A.log_dist && (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA);
  const bool clipk = A.clip > 0.0 && kind >= DOF_PP_COORD && kind <= DOF_PP_DIST_INTRA;
  int wdt;
  float* const out_col = pp_out_column(A, j, &wdt);
#define PP_AT(row) pp_value(A.raw[(voff + (row)) * A.C + c], cf0, cf1, cf2, lg, clipk, A.clip, logtab)
  for (int64_t t = a; t < b; ++t) {
    if (N[t] == 0xffu) continue;
    const int64_t t0 = t * PP_TR, tend = t0 + PP_TR < vlen ? t0 + PP_TR : vlen;
    const unsigned me = bits_of(t);
    const int64_t f = me ? t0 + __builtin_ctz(me) : -1, l = me ? t0 + (31 - __builtin_clz(me)) : -1;
    const bool need_p = f != t0, need_q = l != tend - 1;
    if (!need_p && !need_q) continue;  // only interior gaps, already closed
    int64_t p = -1, q = -1;            // nearest valid rows outside the tile
    if (need_p) {
      p = carry;
      for (int64_t u = t - 1; u >= a; --u)
        if (N[u] != 0u) { p = u * PP_TR + (31 - __builtin_clz((unsigned)N[u])); break; }  // u is never the last tile
    }
    if (need_q) {
      q = back == none ? -1 : back;
      for (int64_t u = t + 1; u < b; ++u) {
        const unsigned mu = bits_of(u);
        if (mu != 0u) { q = u * PP_TR + __builtin_ctz(mu); break; }
      }
    }
    const double pv = p >= 0 ? PP_AT(p) : 0.0, qv = q >= 0 ? PP_AT(q) : 0.0;
    if (f < 0) {  // no valid row in the tile: one gap
      for (int64_t row = t0; row < tend; ++row) out_col[(voff + row) * wdt] = (float)pp_interp(row, p, pv, q, qv);
      continue;
    }
    if (need_p) {
      const double fv = PP_AT(f);
      for (int64_t row = t0; row < f; ++row) out_col[(voff + row) * wdt] = (float)pp_interp(row, p, pv, f, fv);
    }
    if (need_q) {
      const double lv = PP_AT(l);
      for (int64_t row = l + 1; row < tend; ++row) out_col[(voff + row) * wdt] = (float)pp_interp(row, l, lv, q, qv);
    }
  }
#undef PP_AT
}

// ---------------------------------------------------------------------------------------------------------
// Exact order statistics for scale = "robust" (sklearn RobustScaler: center = nanmedian, scale = 75th - 25th percentile,
// numpy's linear interpolation between the two neighbouring order statistics).  Populations: per video the values u of a
// column (per-column sections) or of all columns of a group (groupwise sections) -- stage A, the per-video scalers -- and,
// over the sampled rows of all videos, the per-video-scaled values y = (u - center_v) / scale_v -- stage B, the global
// scalers.  Radix selection as in k_pp_size, but grid-wide: the values become order-preserving 64-bit keys, column-major;
// per 8-bit digit one pass builds the histograms of the next byte among the keys that share a target's prefix (LDS
// atomics per workgroup slice, integer atomics into the population's histogram: order-independent, deterministic) and a
// one-workgroup-per-population kernel advances the (at most) six targets: the two neighbours of the median and of the
// 25th / 75th percentile positions.
constexpr int PP_OS_T = 6;  // order statistics per population
__device__ __forceinline__ unsigned long long pp_order_key(double x) {  // ascending doubles <-> ascending keys; NaN -> PP_NOKEY
  if (pp_isnan(x)) return PP_NOKEY;
  const unsigned long long b = pp_bits(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double pp_order_value(unsigned long long k) {
  return pp_from_bits((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k);
}
// representative column of c's population: itself (per-column section), the first column of its group (groupwise), -1
__device__ __forceinline__ int pp_os_rep(const int* __restrict__ kinds, int C, int c, int speed_mode, int dist_mode, int coord_mode) {
  const int mode = pp_mode(kinds[c], speed_mode, dist_mode, coord_mode);
  if (mode == DOF_PP_MODE_NONE) return -1;
  if (mode == DOF_PP_MODE_PER_COLUMN) return c;
  for (int k = 0; k < c; ++k)
    if (kinds[k] == kinds[c]) return k;
  return c;
}

// keys[c][r] of every scaled column: a 64 x 64 tile of the row-major table through LDS (coalesced reads along the
// columns, coalesced writes along the rows).  vscale == null: u; else y = (u - m) / s of the sampled rows (mask null =
// all rows), each operation rounded on its own like RobustScaler.transform
__global__ void __launch_bounds__(256) k_pp_os_keys(const double* __restrict__ raw, const int64_t* __restrict__ video_off, int V,
                                                    const int* __restrict__ col_kind, const double* __restrict__ rdiv,
                                                    const double* __restrict__ vscale, const uint8_t* __restrict__ mask,
                                                    const uint8_t* __restrict__ keep, int C, int64_t F, int log_dist,
                                                    int speed_mode, int dist_mode, int coord_mode,
                                                    unsigned long long* __restrict__ keys) {
  __shared__ unsigned long long tile[64][65];
  __shared__ double logtab[128][2];
  __shared__ int vrow[64];
  pp_log_table_init(logtab);
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  if (threadIdx.x < 64) {  // video of each row of the tile
    const int64_t r = r0 + threadIdx.x;
    int lo = 0, hi = V - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (video_off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    vrow[threadIdx.x] = lo;
  }
  __syncthreads();
  const int cl = threadIdx.x & 63, c = c0 + cl;
  for (int rl = threadIdx.x >> 6; rl < 64; rl += 4) {
    const int64_t r = r0 + rl;
    unsigned long long key = PP_NOKEY;
    if (r < F && c < C) {
      const int v = vrow[rl], kind = col_kind[c];
      const int64_t i = (int64_t)v * C + c;
      const bool on = pp_mode(kind, speed_mode, dist_mode, coord_mode) != DOF_PP_MODE_NONE && (!keep || keep[i]) &&
                      (!vscale || !mask || mask[r]);
      if (on) {
        double u = raw[r * C + c] * rdiv[i];
        if (log_dist && (kind == DOF_PP_DIST_INNER || kind == DOF_PP_DIST_INTRA)) {
          if (u < 0.0) u = 0.0;
          u = pp_log1p(u, logtab);
        }
        if (vscale) u = (u - vscale[2 * i]) / vscale[2 * i + 1];
        key = pp_order_key(u);
      }
    }
    tile[rl][cl] = key;
  }
  __syncthreads();
  const int rl = threadIdx.x & 63;
  for (int k = threadIdx.x >> 6; k < 64; k += 4)
    if (c0 + k < C && r0 + rl < F) keys[(int64_t)(c0 + k) * F + r0 + rl] = tile[rl][k];
}

struct PpOsState {
  unsigned long long prefix[PP_OS_T];
  long long krem[PP_OS_T];
  int rep[PP_OS_T];  // first target with the same prefix: the histogram both read
  int pad[2];
  double n;
};

// one digit pass: workgroup (population, slice).  per_video: population = (video, representative column), its keys are
// the video's rows of the columns of that population; else population = representative column, all rows
__global__ void __launch_bounds__(256) k_pp_os_hist(const unsigned long long* __restrict__ keys,
                                                    const int64_t* __restrict__ video_off, const int* __restrict__ col_kind, int C,
                                                    int64_t F, int per_video, int speed_mode, int dist_mode, int coord_mode,
                                                    int shift, const PpOsState* __restrict__ state, int* __restrict__ hist) {
  __shared__ int h[PP_OS_T][256];
  const int p = blockIdx.x, c = p % C, v = p / C;
  if (pp_os_rep(col_kind, C, c, speed_mode, dist_mode, coord_mode) != c) return;
  const PpOsState st = state[p];
  if (shift < 56 && st.n == 0.0) return;
  for (int i = threadIdx.x; i < PP_OS_T * 256; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const unsigned long long himask = shift == 56 ? 0ull : ~0ull << (shift + 8);
  const int64_t r0 = per_video ? video_off[v] : 0, r1 = per_video ? video_off[v + 1] : F;
  const bool group = pp_mode(col_kind[c], speed_mode, dist_mode, coord_mode) == DOF_PP_MODE_GROUPWISE;
  for (int cc = c; cc < (group ? C : c + 1); ++cc) {
    if (cc != c && col_kind[cc] != col_kind[c]) continue;
    const unsigned long long* __restrict__ col = keys + (int64_t)cc * F;
    for (int64_t rb = r0 + (int64_t)blockIdx.y * 256 + threadIdx.x; rb < r1; rb += (int64_t)gridDim.y * 256 * 4) {
      unsigned long long k4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t r = rb + (int64_t)i * gridDim.y * 256;
        k4[i] = r < r1 ? col[r] : PP_NOKEY;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (k4[i] == PP_NOKEY) continue;
        const int digit = (int)((k4[i] >> shift) & 255);
#pragma unroll
        for (int j = 0; j < PP_OS_T; ++j)
          if (st.rep[j] == j && (k4[i] & himask) == st.prefix[j]) atomicAdd(&h[j][digit], 1);
      }
    }
  }
  __syncthreads();
  int* __restrict__ out = hist + (int64_t)p * PP_OS_T * 256;
  for (int i = threadIdx.x; i < PP_OS_T * 256; i += 256)
    if ((&h[0][0])[i] != 0) atomicAdd(&out[i], (&h[0][0])[i]);
}

// after a digit pass: next byte of every target (prefix sums over the 256 bins of the histogram it reads), the first
// pass also counts the population and places the targets: ranks (n-1)/2, n/2 | floor((n-1)/4), +1 | floor(3(n-1)/4), +1
__global__ void __launch_bounds__(256) k_pp_os_advance(const int* __restrict__ col_kind, int C, int speed_mode, int dist_mode,
                                                       int coord_mode, int shift, PpOsState* __restrict__ state,
                                                       int* __restrict__ hist, double* __restrict__ out) {
  __shared__ long long scan[256];
  __shared__ PpOsState st;
  const int p = blockIdx.x, c = p % C, tid = threadIdx.x;
  if (pp_os_rep(col_kind, C, c, speed_mode, dist_mode, coord_mode) != c) return;
  int* __restrict__ hp = hist + (int64_t)p * PP_OS_T * 256;
  if (tid == 0) st = state[p];
  __syncthreads();
  if (shift < 56 && st.n == 0.0) return;
  for (int j = 0; j < PP_OS_T; ++j) {
    const int src = st.rep[j];
    const long long mine = hp[src * 256 + tid];
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const long long o = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += o;
      __syncthreads();
    }
    if (shift == 56 && j == 0 && tid == 255) {
      const long long n = scan[255];
      st.n = (double)n;
      const long long m = n > 0 ? n - 1 : 0;
      st.krem[0] = m / 2;
      st.krem[1] = n / 2;
      st.krem[2] = m / 4;
      st.krem[3] = m / 4 + 1 < n ? m / 4 + 1 : m;
      st.krem[4] = 3 * m / 4;
      st.krem[5] = 3 * m / 4 + 1 < n ? 3 * m / 4 + 1 : m;
    }
    __syncthreads();
    const long long k = st.krem[j], inc = scan[tid];
    const bool live = st.n > 0.0;
    __syncthreads();  // every thread holds the rank before the owner of the bin rewrites it
    if (live) {
      if (inc - mine <= k && k < inc) {  // exactly one bin
        st.krem[j] = k - (inc - mine);
        st.prefix[j] |= (unsigned long long)tid << shift;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < PP_OS_T * 256; i += 256) hp[i] = 0;
  if (tid == 0) {
    for (int j = 0; j < PP_OS_T; ++j) {
      int r = j;
      for (int i = j - 1; i >= 0; --i)
        if (st.prefix[i] == st.prefix[j]) r = i;
      st.rep[j] = r;
    }
    state[p] = st;
    if (shift == 0 || st.n == 0.0) {
      double* o = out + (int64_t)p * (PP_OS_T + 1);
      o[0] = st.n;
      for (int j = 0; j < PP_OS_T; ++j) o[1 + j] = st.n > 0.0 ? pp_order_value(st.prefix[j]) : pp_nanv();
    }
  }
}

// a population's result copied to the columns that share it (groupwise sections), NaN rows for unscaled columns
__global__ void __launch_bounds__(256) k_pp_os_spread(const int* __restrict__ col_kind, int C, int64_t P, int speed_mode,
                                                      int dist_mode, int coord_mode, double* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int c = (int)(p % C);
  const int r = pp_os_rep(col_kind, C, c, speed_mode, dist_mode, coord_mode);
  if (r == c) return;
  double* o = out + p * (PP_OS_T + 1);
  if (r < 0) {
    o[0] = 0.0;
    for (int j = 0; j < PP_OS_T; ++j) o[1 + j] = pp_nanv();
  } else {
    const double* src = out + (p - c + r) * (PP_OS_T + 1);
    for (int j = 0; j <= PP_OS_T; ++j) o[j] = src[j];
  }
}

struct PpWorkspace {
  double *hyp, *sfac, *rdiv, *vscale, *coef;
  PpStat *part_all, *part_smp, *vcol_all, *vcol_smp, *ystat, *gcol;
  int *strip_v, *tile_rec, *chunks, *ochunks;
  uint8_t* notes;
  int64_t note_bytes;
  unsigned long long* os_keys;  // scale = "robust" only
  PpOsState* os_state;
  int* os_hist;
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
  w.vcol_all = (PpStat*)take(vc * (int64_t)sizeof(PpStat));
  w.vcol_smp = (PpStat*)take(vc * (int64_t)sizeof(PpStat));
  w.ystat = (PpStat*)take(vc * (int64_t)sizeof(PpStat));
  w.gcol = (PpStat*)take((int64_t)d.n_cols * (int64_t)sizeof(PpStat));
  w.note_bytes = tiles * n_out;
  w.notes = (uint8_t*)take(w.note_bytes);
  w.strip_v = (int*)take(strips * 4);
  w.tile_rec = (int*)take(tiles * 16);
  w.chunks = (int*)take((1 + 3 * (int64_t)(2 * d.n_cols + 2)) * 4);
  w.ochunks = (int*)take((1 + 3 * (int64_t)(2 * n_out + 2)) * 4);
  const bool robust = d.scale_kind == DOF_PP_SCALE_ROBUST;
  w.os_keys = (unsigned long long*)take(robust ? d.n_frames * d.n_cols * 8 : 0);
  w.os_state = (PpOsState*)take(robust ? vc * (int64_t)sizeof(PpOsState) : 0);
  w.os_hist = (int*)take(robust ? vc * PP_OS_T * 256 * 4 : 0);
  w.bytes = off;
  return w;
}

int pp_check(const DofPreprocDims* d, bool stats_only = false) {
  if (!d) {
    dof_set_error("dof_preprocess: dims is null");
    return DOF_ERR_ARG;
  }
  const int n_out = d->n_node_cols + d->n_edge_cols + d->n_angle_cols;
  if (d->n_frames <= 0 || d->n_videos <= 0 || d->n_cols <= 0 || d->n_animals < 0 || (n_out <= 0 && !stats_only) || d->n_node_cols < 0 ||
      d->n_edge_cols < 0 || d->n_angle_cols < 0 || d->clip < 0.0) {
    dof_set_error("dof_preprocess: bad dims");
    return DOF_ERR_ARG;
  }
  if (d->n_cols > DOF_PP_MAX_COLS || d->n_animals > DOF_PP_MAX_ANIMALS || n_out > DOF_PP_MAX_COLS || d->n_frames >= (1ll << 31)) {
    dof_set_error("dof_preprocess: unsupported size (columns <= %d, animals <= %d, output columns <= columns, frames < 2^31)",
                  DOF_PP_MAX_COLS, DOF_PP_MAX_ANIMALS);
    return DOF_ERR_UNSUPPORTED;
  }
  for (int m : {d->speed_mode, d->dist_mode, d->coord_mode})
    if (m < DOF_PP_MODE_NONE || m > DOF_PP_MODE_GROUPWISE) {
      dof_set_error("dof_preprocess: bad standardisation mode %d", m);
      return DOF_ERR_ARG;
    }
  if (d->scale_kind < DOF_PP_SCALE_STANDARD || d->scale_kind > DOF_PP_SCALE_ROBUST) {
    dof_set_error("dof_preprocess: bad scale_kind %d", d->scale_kind);
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
  if (pp_check(dims, true) != DOF_OK) return -1;  // no output columns: the statistics-only entry points
  return pp_layout(*dims, nullptr).bytes;
}

namespace {
// stats_only: stop after the per-video statistics and hand out the (n, mean, M2) rows of the sampled,
// per-video-standardised values (what the global scalers are fitted on)
int pp_run(const DofPreprocDims* dims, bool stats_only, const double* raw, const int64_t* video_off, const int32_t* col_kind,
           const int32_t* size_ref, const int32_t* chain_off, const int32_t* chain, const int32_t* out_cols,
           const uint8_t* sample_mask, double* scaler, double* size_out, double* video_scaler, double* ystat_out,
           float* node_out, float* edge_out, float* angle_out, void* workspace, void* stream) {
  const int rc = pp_check(dims, stats_only);
  if (rc != DOF_OK) return rc;
  const DofPreprocDims& d = *dims;
  const bool bad_out = !stats_only && (!out_cols || !scaler || (d.n_node_cols > 0 && !node_out) ||
                                       (d.n_edge_cols > 0 && !edge_out) || (d.n_angle_cols > 0 && !angle_out));
  if (!raw || !video_off || !col_kind || !chain_off || !workspace || (d.n_animals > 0 && !size_ref) || bad_out ||
      (stats_only && !ystat_out)) {
    dof_set_error("dof_preprocess: null pointer argument");
    return DOF_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const PpWorkspace w = pp_layout(d, workspace);
  const int V = d.n_videos, C = d.n_cols, A = d.n_animals;
  const int n_out = stats_only ? 0 : d.n_node_cols + d.n_edge_cols + d.n_angle_cols;
  const int64_t strips = pp_slots(d.n_frames, V, PP_RS), tiles = pp_slots(d.n_frames, V, PP_TR);
  const bool coord_stats = stats_only || d.fit_global;
  DOF_LAUNCH(k_pp_setup, (dof_cdiv(strips, DOF_PP_MAX_COLS) + dof_cdiv(tiles, DOF_PP_MAX_COLS) + 2), (DOF_PP_MAX_COLS), st,
             video_off, V, strips, tiles, col_kind, out_cols, C, n_out, d.log_distances, w.strip_v, w.tile_rec, w.chunks,
             w.ochunks);
  if (!stats_only) (void)hipMemsetAsync(w.notes, 0xFF, (size_t)w.note_bytes, st);
  if (A > 0) {
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(w.hyp);
    DOF_LAUNCH(k_pp_hyp, (dof_cdiv(d.n_frames, 256)), (256), st, raw, size_ref, C, A, d.n_frames, video_off, V, d.col_keep, keys);
    DOF_LAUNCH(k_pp_size, (A, V), (256), st, video_off, size_ref, A, d.n_frames, (const unsigned long long*)keys, w.sfac);
  }
  DOF_LAUNCH(k_pp_divisors, (V), (256), st, chain_off, chain, C, A, d.inter_scale, d.col_keep, w.sfac, w.rdiv);
  const bool robust = d.scale_kind == DOF_PP_SCALE_ROBUST;
  if (robust) {  // order statistics, not moments: both scalers come from dof_preprocess_order_stats through the caller
    if (stats_only || d.fit_global || !d.video_scaler_in) {
      dof_set_error("dof_preprocess: scale_kind robust needs video_scaler_in and the fitted `scaler` (fit_global = 0)");
      return DOF_ERR_ARG;
    }
    (void)hipMemcpyAsync(w.vscale, d.video_scaler_in, (size_t)V * C * 16, hipMemcpyDeviceToDevice, st);
  }
#define PP_STATS(M)                                                                                                  \
  DOF_LAUNCH((k_pp_stats<M>), ((unsigned)strips), (256), st, raw, video_off, (const int*)w.strip_v, (const int*)w.chunks, \
             col_kind, (const double*)w.rdiv, sample_mask, C, d.speed_mode, d.dist_mode,                             \
             coord_stats ? d.coord_mode : DOF_PP_MODE_NONE, 0, d.col_keep, w.part_all, w.part_smp)
  const unsigned col_chunks = dof_cdiv(C, 64);
  if (!robust) {
    if (sample_mask) PP_STATS(true); else PP_STATS(false);
    DOF_LAUNCH(k_pp_video_cols, (col_chunks, V), (256), st, video_off, (const PpStat*)w.part_all, (const PpStat*)w.part_smp, C,
               w.vcol_all, w.vcol_smp);
    DOF_LAUNCH(k_pp_video_fin, (V), (256), st, col_kind, (const PpStat*)w.vcol_all, (const PpStat*)w.vcol_smp, C, d.speed_mode,
               d.dist_mode, d.scale_kind, w.vscale, w.ystat);
  }
#undef PP_STATS
  if (size_out) (void)hipMemcpyAsync(size_out, w.sfac, (size_t)V * (A + 1) * 8, hipMemcpyDeviceToDevice, st);
  if (stats_only) {
    (void)hipMemcpyAsync(ystat_out, w.ystat, (size_t)V * C * sizeof(PpStat), hipMemcpyDeviceToDevice, st);
    return dof_check_launch("dof_preprocess_video_stats");
  }
  if (d.fit_global) DOF_LAUNCH(k_pp_global_cols, (col_chunks), (256), st, (const PpStat*)w.ystat, V, C, w.gcol);
  DOF_LAUNCH(k_pp_coef, (V), (256), st, col_kind, (const PpStat*)w.gcol, (const double*)w.rdiv, (const double*)w.vscale,
             scaler, d.fit_global, C, d.speed_mode, d.dist_mode, d.coord_mode, d.scale_kind, d.col_keep, w.coef, video_scaler);
  PpOutArgs oa;
  oa.raw = raw;
  oa.video_off = video_off;
  oa.slot_rec = w.tile_rec;
  oa.col_kind = col_kind;
  oa.out_cols = out_cols;
  oa.coef = w.coef;
  oa.n_slots = tiles;
  oa.C = C;
  oa.n_out = n_out;
  oa.n_node = d.n_node_cols;
  oa.n_edge = d.n_edge_cols;
  oa.log_dist = d.log_distances;
  oa.clip = d.clip;
  oa.node_out = node_out;
  oa.edge_out = edge_out;
  oa.angle_out = angle_out;
  DOF_LAUNCH(k_pp_finish, (dof_cdiv(tiles, 32)), (256), st, oa, (const int*)w.ochunks, w.notes);
  DOF_LAUNCH(k_pp_fill, (n_out, V), (256), st, oa, (const uint8_t*)w.notes);
  return dof_check_launch("dof_preprocess_tables");
}
}  // namespace

extern "C" int dof_preprocess_tables(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                                     const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                                     const int32_t* chain, const int32_t* out_cols, const uint8_t* sample_mask,
                                     double* scaler, double* size_out, double* video_scaler, float* node_out,
                                     float* edge_out, float* angle_out, void* workspace, void* stream) {
  return pp_run(dims, false, raw, video_off, col_kind, size_ref, chain_off, chain, out_cols, sample_mask, scaler, size_out,
                video_scaler, nullptr, node_out, edge_out, angle_out, workspace, stream);
}

extern "C" int dof_preprocess_video_stats(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                                          const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                                          const int32_t* chain, const uint8_t* sample_mask, double* ystat_out,
                                          void* workspace, void* stream) {
  return pp_run(dims, true, raw, video_off, col_kind, size_ref, chain_off, chain, nullptr, sample_mask, nullptr, nullptr, nullptr,
                ystat_out, nullptr, nullptr, nullptr, workspace, stream);
}

extern "C" int dof_preprocess_fit_global(const DofPreprocDims* dims, int32_t n_videos_total, const int32_t* col_kind,
                                         const double* ystat_all, double* scaler, void* stream) {
  if (!dims || !col_kind || !ystat_all || !scaler || n_videos_total <= 0 || dims->n_cols <= 0 || dims->n_cols > DOF_PP_MAX_COLS) {
    dof_set_error("dof_preprocess_fit_global: bad argument");
    return DOF_ERR_ARG;
  }
  DOF_LAUNCH(k_pp_fit_global, (1), (256), (hipStream_t)stream, col_kind, reinterpret_cast<const PpStat*>(ystat_all),
             n_videos_total, dims->n_cols, dims->speed_mode, dims->dist_mode, dims->coord_mode, dims->scale_kind, scaler);
  return dof_check_launch("dof_preprocess_fit_global");
}

extern "C" int dof_preprocess_order_stats(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                                          const int32_t* col_kind, const int32_t* size_ref, const int32_t* chain_off,
                                          const int32_t* chain, const double* video_scaler, const uint8_t* sample_mask,
                                          double* out, void* workspace, void* stream) {
  const int rc = pp_check(dims, true);
  if (rc != DOF_OK) return rc;
  const DofPreprocDims& d = *dims;
  if (!raw || !video_off || !col_kind || !chain_off || !out || !workspace || (d.n_animals > 0 && !size_ref) ||
      d.scale_kind != DOF_PP_SCALE_ROBUST) {
    dof_set_error("dof_preprocess_order_stats: null pointer argument or scale_kind is not robust");
    return DOF_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const PpWorkspace w = pp_layout(d, workspace);
  const int V = d.n_videos, C = d.n_cols, A = d.n_animals;
  const bool stage_a = video_scaler == nullptr;
  const int coord_mode = stage_a ? DOF_PP_MODE_NONE : d.coord_mode;  // the per-video scaling leaves the coordinates alone
  const int64_t P = stage_a ? (int64_t)V * C : C;
  if (A > 0) {
    unsigned long long* hk = reinterpret_cast<unsigned long long*>(w.hyp);
    DOF_LAUNCH(k_pp_hyp, (dof_cdiv(d.n_frames, 256)), (256), st, raw, size_ref, C, A, d.n_frames, video_off, V, d.col_keep, hk);
    DOF_LAUNCH(k_pp_size, (A, V), (256), st, video_off, size_ref, A, d.n_frames, (const unsigned long long*)hk, w.sfac);
  }
  DOF_LAUNCH(k_pp_divisors, (V), (256), st, chain_off, chain, C, A, d.inter_scale, d.col_keep, w.sfac, w.rdiv);
  DOF_LAUNCH(k_pp_os_keys, (dof_cdiv(d.n_frames, 64), dof_cdiv(C, 64)), (256), st, raw, video_off, V, col_kind,
             (const double*)w.rdiv, video_scaler, sample_mask, d.col_keep, C, d.n_frames, d.log_distances, d.speed_mode,
             d.dist_mode, coord_mode, w.os_keys);
  (void)hipMemsetAsync(w.os_state, 0, (size_t)P * sizeof(PpOsState), st);
  (void)hipMemsetAsync(w.os_hist, 0, (size_t)P * PP_OS_T * 256 * 4, st);
  // slices per population: a few thousand keys per workgroup and pass
  const int64_t rows = stage_a ? d.n_frames / V + 1 : d.n_frames;
  int64_t slices = rows / 8192 + 1;
  if (slices > 128) slices = 128;
  for (int shift = 56; shift >= 0; shift -= 8) {
    DOF_LAUNCH(k_pp_os_hist, ((unsigned)P, (unsigned)slices), (256), st, (const unsigned long long*)w.os_keys, video_off, col_kind,
               C, d.n_frames, stage_a ? 1 : 0, d.speed_mode, d.dist_mode, coord_mode, shift, (const PpOsState*)w.os_state,
               w.os_hist);
    DOF_LAUNCH(k_pp_os_advance, ((unsigned)P), (256), st, col_kind, C, d.speed_mode, d.dist_mode, coord_mode, shift, w.os_state,
               w.os_hist, out);
  }
  DOF_LAUNCH(k_pp_os_spread, (dof_cdiv(P, 256)), (256), st, col_kind, C, P, d.speed_mode, d.dist_mode, coord_mode, out);
  return dof_check_launch("dof_preprocess_order_stats");
}

extern "C" int dof_preprocess_raw_moments(const DofPreprocDims* dims, const double* raw, const int64_t* video_off,
                                          const int32_t* col_kind, double* moments_out, void* workspace, void* stream) {
  const int rc = pp_check(dims, true);
  if (rc != DOF_OK) return rc;
  if (!raw || !video_off || !col_kind || !moments_out || !workspace) {
    dof_set_error("dof_preprocess_raw_moments: null pointer argument");
    return DOF_ERR_ARG;
  }
  const DofPreprocDims& d = *dims;
  hipStream_t st = (hipStream_t)stream;
  const PpWorkspace w = pp_layout(d, workspace);
  const int V = d.n_videos, C = d.n_cols;
  const int64_t strips = pp_slots(d.n_frames, V, PP_RS), tiles = pp_slots(d.n_frames, V, PP_TR);
  DOF_LAUNCH(k_pp_setup, (dof_cdiv(strips, DOF_PP_MAX_COLS) + dof_cdiv(tiles, DOF_PP_MAX_COLS) + 2), (DOF_PP_MAX_COLS), st,
             video_off, V, strips, tiles, col_kind, (const int*)nullptr, C, 0, 0, w.strip_v, w.tile_rec, w.chunks, w.ochunks);
  DOF_LAUNCH((k_pp_stats<false>), ((unsigned)strips), (256), st, raw, video_off, (const int*)w.strip_v, (const int*)w.chunks,
             col_kind, (const double*)nullptr, (const uint8_t*)nullptr, C, 0, 0, 0, 1, (const uint8_t*)nullptr, w.part_all, w.part_smp);
  DOF_LAUNCH(k_pp_video_cols, (dof_cdiv(C, 64), V), (256), st, video_off, (const PpStat*)w.part_all, (const PpStat*)w.part_smp, C,
             w.vcol_all, w.vcol_smp);
  (void)hipMemcpyAsync(moments_out, w.vcol_all, (size_t)V * C * sizeof(PpStat), hipMemcpyDeviceToDevice, st);
  return dof_check_launch("dof_preprocess_raw_moments");
}
