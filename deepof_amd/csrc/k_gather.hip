// Window tensor build (SURVEY.md section 8a rows R0+R1), HBM-bound.
//
// Replaces, on device, the reference's host pipeline
//   rolling_window            /root/reference/deepof/utils.py:3354-3377   (as_strided windows)
//   reorder_and_reshape       /root/reference/deepof/clustering/dataset.py:16-26
//   edge expand_dims + fp32   dataset.py:81, :183-290 (HDF5 materialisation)
// Frame tables stay resident in HBM ((frames, 3N) column blocks [x..|y..|s..] and (frames, E));
// a launch writes the reference-layout batch x (B,W,N,3), a (B,W,E,1) for any list of window
// start rows.  Algorithmic bytes per window: W*(3N+E)*4 written + (3N+E)*4 newly read.
//
// Stores are 16-byte, coalesced, streaming / non-temporal so the 5.6 KB-per-window write stream
// does not evict the small, heavily re-read frame rows from L2; the 4 source words of a store
// come from the (cached) frame rows.
#include <cmath>

#include "dof_rt.h"
#include "deepof_hip.h"

namespace {

constexpr int WB = 16;  // windows per workgroup (a multiple of 4 keeps every workgroup's output 16-byte aligned)
constexpr int WB_SMALL = 4;  // ... for launches of a few thousand windows (a training batch): 4x the workgroups

// exact unsigned division for n < 2^24 by a runtime divisor with a precomputed (rounded-down) reciprocal
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, float rcp_lo) {
  unsigned q = (unsigned)((float)n * rcp_lo);  // never overshoots
  if (n - q * d >= d) ++q;
  return q;
}

// float -> bf16 bits, round to nearest even; +-Inf stay Inf, a NaN keeps its sign and upper payload and is made quiet
// (the rounding add would turn a NaN with low mantissa bits only into Inf and wrap an all-ones NaN to zero)
__device__ __forceinline__ uint32_t bf16_rne(float v) {
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  const uint32_t r = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  return (u & 0x7FFFFFFFu) > 0x7F800000u ? ((u >> 16) | 0x40u) : r;
}
// One 16-byte streaming store of EPV consecutive output elements starting at element e0 (EPV = 4 fp32 or 8 bf16), or the
// tail elements one by one
template <bool OUT16>
__device__ __forceinline__ void store_run(void* __restrict__ out_base, unsigned e0, unsigned total, const float* v) {
  constexpr int EPV = OUT16 ? 8 : 4;
  if (e0 + EPV - 1 < total) {
    if constexpr (OUT16) {
      const uint32_t w[4] = {bf16_rne(v[0]) | (bf16_rne(v[1]) << 16), bf16_rne(v[2]) | (bf16_rne(v[3]) << 16),
                             bf16_rne(v[4]) | (bf16_rne(v[5]) << 16), bf16_rne(v[6]) | (bf16_rne(v[7]) << 16)};
      dof_st_stream4(reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(out_base) + e0), w);
    } else {
      const float f[4] = {v[0], v[1], v[2], v[3]};
      dof_st_stream4(reinterpret_cast<float*>(out_base) + e0, f);
    }
  } else {
    for (int j = 0; j < EPV; ++j)
      if (e0 + j < total) {
        if constexpr (OUT16) reinterpret_cast<uint16_t*>(out_base)[e0 + j] = (uint16_t)bf16_rne(v[j]);
        else reinterpret_cast<float*>(out_base)[e0 + j] = v[j];
      }
  }
}

// Mapping: one workgroup = WB consecutive windows (a 16-byte aligned, 4*WB*W*(3N+E)-byte output
// run), one thread = one 16-byte streaming store at a time.  All per-element index arithmetic is
// 32-bit with reciprocal multiplies (the first version spent its time in 64-bit integer division
// and reached only 25 % of HBM peak); the 64-bit row offsets of the WB windows sit in LDS.
//
// STAGE > 0 (the normal case): the frame rows the workgroup's windows cover -- (WB-1)*step + W rows for
// regularly spaced windows, 40 rows = 9 KB for BASELINE C2 -- are first copied into LDS with coalesced
// loads, and a W*3N-entry table holds the (frame, column) source offset of every element of a window, so a
// 16-byte store costs four table lookups + four LDS reads instead of four strided global gathers (the
// gathers kept the kernel on the texture-address path: 56 % of HBM peak; a pure fill reaches 6.2 TB/s here).
// Workgroups whose windows are too far apart for the staging buffer take the direct path (STAGE == 0 code).
// OUT16: the batch leaves as bf16 (the storage format of BASELINE's bf16 configuration: W*(3N+E)*2 bytes written per
// window), eight elements per 16-byte store.
template <bool INDEXED, int STAGE, int TAB, int WB, bool OUT16 = false>
__global__ void __launch_bounds__(256) k_window_gather(
    const float* __restrict__ node_table, const float* __restrict__ edge_table,
    const int64_t* __restrict__ row_start, int64_t first_row, int64_t row_step, int64_t n_windows,
    int W, int N, int E, float rcp_perwin, float rcp_cols, void* __restrict__ x_out, void* __restrict__ a_out) {
  constexpr int EPV = OUT16 ? 8 : 4;   // elements per 16-byte store
  constexpr int OSZ = OUT16 ? 2 : 4;   // bytes per output element
  __shared__ int64_t row0[WB];
  __shared__ int lrow[WB];
  __shared__ int64_t span[2];
  __shared__ float stage[STAGE > 0 ? STAGE : 1];
  __shared__ int tab[TAB > 0 ? TAB : 1];
  const int C = 3 * N;
  const unsigned per_x = (unsigned)(W * C), per_a = (unsigned)(W * E);
  const int64_t w0 = (int64_t)blockIdx.x * WB;
  const int nwin = (int)((n_windows - w0) < WB ? (n_windows - w0) : WB);
  if ((int)threadIdx.x < nwin)
    row0[threadIdx.x] = INDEXED ? row_start[w0 + threadIdx.x] : first_row + (w0 + threadIdx.x) * row_step;
  __syncthreads();
  bool staged = false;
  if (STAGE > 0) {
    if (threadIdx.x == 0) {
      int64_t lo = row0[0], hi = row0[0];
      for (int i = 1; i < nwin; ++i) {
        lo = row0[i] < lo ? row0[i] : lo;
        hi = row0[i] > hi ? row0[i] : hi;
      }
      span[0] = lo;
      span[1] = hi - lo + W;  // rows covered
    }
    __syncthreads();
    const int64_t rmin = span[0], rows = span[1];
    staged = rows * (C + E) <= STAGE && (int)per_x <= TAB;
    if (staged) {
      const int nnode = (int)rows * C, nedge = (int)rows * E;
      const float* __restrict__ srcn = node_table + rmin * C;
      const float* __restrict__ srce = edge_table + rmin * E;
      for (int i = threadIdx.x; i < nnode; i += 256) stage[i] = srcn[i];
      for (int i = threadIdx.x; i < nedge; i += 256) stage[nnode + i] = srce[i];
      // element o = (t, n, f) of a window  <-  frame t, column f*N + n
      for (unsigned o = threadIdx.x; o < per_x; o += 256) {
        const unsigned t = fast_div(o, (unsigned)C, rcp_cols);
        const unsigned r = o - t * C;
        const unsigned n = (r * 43691u) >> 17;  // r / 3 for r < 2^16
        tab[o] = (int)(t * C + (r - 3 * n) * N + n);
      }
      if ((int)threadIdx.x < nwin) lrow[threadIdx.x] = (int)(row0[threadIdx.x] - rmin);
      __syncthreads();
      // ---- nodes: (window, element) of a thread's next store advance incrementally (1024 floats per pass)
      {
        char* __restrict__ out = reinterpret_cast<char*>(x_out) + w0 * per_x * OSZ;
        const unsigned total = (unsigned)nwin * per_x;
        unsigned e0 = EPV * threadIdx.x;
        unsigned w = fast_div(e0, per_x, rcp_perwin);
        unsigned o = e0 - w * per_x;
        for (; e0 < total; e0 += EPV * 256) {
          float v[EPV];
          unsigned wj = w < (unsigned)nwin ? w : (unsigned)nwin - 1, oj = o;
          int base = lrow[wj] * C;
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            v[j] = stage[base + tab[oj]];
            if (++oj == per_x) {
              oj = 0;
              wj = wj + 1 < (unsigned)nwin ? wj + 1 : wj;
              base = lrow[wj] * C;
            }
          }
          store_run<OUT16>(out, e0, total, v);
          o += EPV * 256;
          while (o >= per_x) {
            o -= per_x;
            ++w;
          }
        }
      }
      // ---- edges: a window is W*E contiguous staged floats
      {
        char* __restrict__ out = reinterpret_cast<char*>(a_out) + w0 * per_a * OSZ;
        const float* __restrict__ se = stage + nnode;
        const unsigned total = (unsigned)nwin * per_a;
        const float rcp_pa = rcp_perwin * ((float)C / (float)E) * 0.999999f;
        for (unsigned e0 = EPV * threadIdx.x; e0 < total; e0 += EPV * 256) {
          float v[EPV];
          unsigned w = fast_div(e0, per_a, rcp_pa);
          unsigned o = e0 - w * per_a;
          int base = lrow[w] * E;
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            v[j] = se[base + o];
            if (++o == per_a) {
              o = 0;
              w = w + 1 < (unsigned)nwin ? w + 1 : w;
              base = lrow[w] * E;
            }
          }
          store_run<OUT16>(out, e0, total, v);
        }
      }
      return;
    }
  }
  // ---- direct path (windows of this workgroup too far apart to stage)
  // ---- nodes: (rows, [x..|y..|s..]) -> (W, N, 3)
  {
    char* __restrict__ out = reinterpret_cast<char*>(x_out) + w0 * per_x * OSZ;
    const unsigned total = (unsigned)nwin * per_x;
    for (unsigned e0 = EPV * threadIdx.x; e0 < total; e0 += EPV * 256) {
      float v[EPV];
      unsigned w = fast_div(e0, per_x, rcp_perwin);
      unsigned o = e0 - w * per_x;
      unsigned t = fast_div(o, (unsigned)C, rcp_cols);
      unsigned r = o - t * C;
      unsigned n = (r * 43691u) >> 17;  // r / 3 for r < 2^16
      unsigned f = r - 3 * n;
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        v[j] = (e0 + j < total) ? node_table[(row0[w] + t) * C + f * N + n] : 0.0f;
        if (++f == 3) { f = 0; ++n; }
        if (++r == (unsigned)C) { r = 0; n = 0; f = 0; ++t; }
        if (++o == per_x) { o = 0; t = 0; ++w; if (w >= (unsigned)nwin) w = nwin - 1; }
      }
      store_run<OUT16>(out, e0, total, v);
    }
  }
  // ---- edges: a window is W*E contiguous source floats
  {
    char* __restrict__ out = reinterpret_cast<char*>(a_out) + w0 * per_a * OSZ;
    const unsigned total = (unsigned)nwin * per_a;
    const float rcp_pa = rcp_perwin * ((float)C / (float)E) * 0.999999f;
    for (unsigned e0 = EPV * threadIdx.x; e0 < total; e0 += EPV * 256) {
      float v[EPV];
      unsigned w = fast_div(e0, per_a, rcp_pa);
      unsigned o = e0 - w * per_a;
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        v[j] = (e0 + j < total) ? edge_table[row0[w] * E + o] : 0.0f;
        if (++o == per_a) { o = 0; ++w; if (w >= (unsigned)nwin) w = nwin - 1; }
      }
      store_run<OUT16>(out, e0, total, v);
    }
  }
}

// largest float not above 1/d (so fast_div's first guess never overshoots)
float rcp_down(unsigned d) {
  float r = 1.0f / (float)d;
  while ((double)r * (double)d > 1.0) r = nextafterf(r, 0.0f);
  return r * 0.9999999f;
}

int launch_gather(const float* node_table, const float* edge_table, const int64_t* row_start, int64_t first_row,
                  int64_t row_step, int64_t n_windows, int W, int N, int E, void* x_out, void* a_out,
                  hipStream_t stream, bool out16 = false) {
  if (!node_table || !edge_table || !x_out || !a_out || n_windows < 0 || W <= 0 || N <= 0 || E <= 0) {
    dof_set_error("dof_window_gather: bad argument");
    return DOF_ERR_ARG;
  }
  if ((int64_t)WB * W * 3 * N >= (1 << 24) || (int64_t)WB * W * E >= (1 << 24)) {
    dof_set_error("dof_window_gather: window too large (W*3N*16 must stay below 2^24)");
    return DOF_ERR_UNSUPPORTED;
  }
  if (out16 && ((W * 3 * N) % 2 != 0 || (W * E) % 2 != 0)) {
    dof_set_error("dof_window_gather_bf16: W*3N and W*E must be even (16-byte aligned bf16 runs)");
    return DOF_ERR_UNSUPPORTED;
  }
  if (n_windows == 0) return DOF_OK;
  // a training batch (1024 windows) is 64 workgroups of 16 windows on a 256-CU chip: use 4 windows per workgroup there
  const bool small = n_windows <= 8192;
  const int wb = small ? WB_SMALL : WB;
  const unsigned blocks = (unsigned)((n_windows + wb - 1) / wb);
  const float rp = rcp_down((unsigned)(W * 3 * N)), rc = rcp_down((unsigned)(3 * N));
  // staging-buffer class by the footprint of wb stride-1 windows (other spacings decide per workgroup)
  const int64_t need = (int64_t)(wb - 1 + W) * (3 * N + E), tabn = (int64_t)W * 3 * N;
#define GATHER1(IDX, ST, TB, WBV)                                                                                        \
  do {                                                                                                                   \
    if (out16)                                                                                                           \
      DOF_LAUNCH((k_window_gather<IDX, ST, TB, WBV, true>), (blocks), (256), stream, node_table, edge_table, row_start,  \
                 first_row, row_step, n_windows, W, N, E, rp, rc, x_out, a_out);                                         \
    else                                                                                                                 \
      DOF_LAUNCH((k_window_gather<IDX, ST, TB, WBV, false>), (blocks), (256), stream, node_table, edge_table, row_start, \
                 first_row, row_step, n_windows, W, N, E, rp, rc, x_out, a_out);                                         \
  } while (0)
#define GATHER(IDX, ST, TB)                                            \
  do {                                                                 \
    if (small) GATHER1(IDX, ST, TB, WB_SMALL); else GATHER1(IDX, ST, TB, WB); \
  } while (0)
  if (need <= 3072 && tabn <= 1280) {
    if (row_start) GATHER(true, 3072, 1280); else GATHER(false, 3072, 1280);
  } else if (need <= 10240 && tabn <= 4608) {
    if (row_start) GATHER(true, 10240, 4608); else GATHER(false, 10240, 4608);
  } else {
    if (row_start) GATHER(true, 0, 0); else GATHER(false, 0, 0);
  }
#undef GATHER1
#undef GATHER
  return dof_check_launch("k_window_gather");
}

}  // namespace

extern "C" int dof_window_gather(const float* node_table, const float* edge_table, const int64_t* row_start,
                                 int64_t n_windows, int32_t window, int32_t n_nodes, int32_t n_edges, float* x_out,
                                 float* a_out, void* stream) {
  if (!row_start && n_windows > 0) {
    dof_set_error("dof_window_gather: row_start is null");
    return DOF_ERR_ARG;
  }
  return launch_gather(node_table, edge_table, row_start, 0, 0, n_windows, window, n_nodes, n_edges, x_out, a_out,
                       (hipStream_t)stream);
}

extern "C" int dof_window_gather_range(const float* node_table, const float* edge_table, int64_t first_row,
                                       int64_t row_step, int64_t n_windows, int32_t window, int32_t n_nodes,
                                       int32_t n_edges, float* x_out, float* a_out, void* stream) {
  return launch_gather(node_table, edge_table, nullptr, first_row, row_step, n_windows, window, n_nodes, n_edges, x_out,
                       a_out, (hipStream_t)stream);
}

extern "C" int dof_window_gather_bf16(const float* node_table, const float* edge_table, const int64_t* row_start,
                                      int64_t first_row, int64_t row_step, int64_t n_windows, int32_t window, int32_t n_nodes,
                                      int32_t n_edges, uint16_t* x_out, uint16_t* a_out, void* stream) {
  return launch_gather(node_table, edge_table, row_start, first_row, row_step, n_windows, window, n_nodes, n_edges, x_out, a_out,
                       (hipStream_t)stream, true);
}

namespace {
// bf16 -> fp32 (exact): the train step's fp32 kernels read a widened copy of a bf16-stored batch (2 + 4 bytes per element,
// 8.6 MB for a C2 batch: the batches cross HBM between the gather and the step at half the bytes)
__global__ void __launch_bounds__(256) k_widen_bf16(const uint16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const uint64_t p = *reinterpret_cast<const uint64_t*>(in + i);
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32);
    float4 v;
    v.x = __builtin_bit_cast(float, lo << 16);
    v.y = __builtin_bit_cast(float, lo & 0xFFFF0000u);
    v.z = __builtin_bit_cast(float, hi << 16);
    v.w = __builtin_bit_cast(float, hi & 0xFFFF0000u);
    *reinterpret_cast<float4*>(out + i) = v;
  } else {
    for (int64_t k = i; k < n; ++k) out[k] = __builtin_bit_cast(float, (uint32_t)in[k] << 16);
  }
}
}  // namespace

extern "C" int dof_widen_bf16(const uint16_t* in, float* out, int64_t n, void* stream) {
  if (!in || !out || n < 0) {
    dof_set_error("dof_widen_bf16: bad argument");
    return DOF_ERR_ARG;
  }
  if (n == 0) return DOF_OK;
  DOF_LAUNCH(k_widen_bf16, (dof_cdiv(dof_cdiv(n, 4), 256)), (256), (hipStream_t)stream, in, out, n);
  return dof_check_launch("k_widen_bf16");
}
