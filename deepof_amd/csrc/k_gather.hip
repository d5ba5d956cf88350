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
// two floats -> one word of two bf16 (first argument in the low half), round to nearest even.  Device: one
// v_cvt_pk_bf16_f32 (gfx950); the emulation build uses the integer form above.
__device__ __forceinline__ uint32_t bf16_pack2(float a, float b) {
#ifdef DOF_EMU
  return bf16_rne(a) | (bf16_rne(b) << 16);
#else
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
#endif
}
// One 16-byte streaming store of EPV consecutive output elements starting at element e0 (EPV = 4 fp32 or 8 bf16), or the
// tail elements one by one
template <bool OUT16>
__device__ __forceinline__ void store_run(void* __restrict__ out_base, unsigned e0, unsigned total, const float* v) {
  constexpr int EPV = OUT16 ? 8 : 4;
  if (e0 + EPV - 1 < total) {
    if constexpr (OUT16) {
      const uint32_t w[4] = {bf16_pack2(v[0], v[1]), bf16_pack2(v[2], v[3]), bf16_pack2(v[4], v[5]), bf16_pack2(v[6], v[7])};
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
// regularly spaced windows, 40 rows = 9 KB for BASELINE C2 -- are first copied into LDS with coalesced loads and
// TRANSPOSED on the way in: a staged node row holds its values in the OUTPUT order (n, f) instead of the table's column
// blocks [x.. | y.. | s..], written with an LDS stride of 3 dwords across neighbouring lanes (conflict-free).  A window
// is then W*3N CONTIGUOUS staged floats, so a 16-byte store is two 8-byte LDS reads of consecutive addresses -- no
// bank conflicts, no offset table.  (History: strided global gathers, 56 % of the HBM peak; rows staged in table order
// + a W*3N-entry offset table, 4 lookups + 4 gathered LDS reads per store: 0.66 of peak for fp32 and 0.49 for the bf16
// output, where PMC showed 62 % / 76 % of all LDS cycles to be bank-conflict cycles and the bf16 kernel LDS-bound --
// 431 of its 460 us; profiles/r04_gather_pmc.md.)
// Workgroups whose windows are too far apart for the staging buffer take the direct path (STAGE == 0 code).
// OUT16: the batch leaves as bf16 (the storage format of BASELINE's bf16 configuration: W*(3N+E)*2 bytes written per
// window), eight elements per 16-byte store.
template <bool INDEXED, int STAGE, int WB, bool OUT16 = false>
__global__ void __launch_bounds__(256) k_window_gather(
    const float* __restrict__ node_table, const float* __restrict__ edge_table,
    const int64_t* __restrict__ row_start, int64_t first_row, int64_t row_step, int64_t n_windows,
    int W, int N, int E, float rcp_perwin, float rcp_cols, float rcp_n, void* __restrict__ x_out, void* __restrict__ a_out) {
  constexpr int EPV = OUT16 ? 8 : 4;   // elements per 16-byte store
  constexpr int OSZ = OUT16 ? 2 : 4;   // bytes per output element
  __shared__ int64_t row0[WB];
  __shared__ int lrow[WB];
  __shared__ int64_t span[2];
  __shared__ __attribute__((aligned(16))) float stage[STAGE > 0 ? STAGE : 1];
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
    // the edge rows are staged behind the node rows at a 4-float boundary, so that `at & 3` / `at & 1` in copy_windows
    // speak about the real LDS address of BOTH tables (rows * C need not be a multiple of 4)
    const int64_t nnode_pad = (rows * C + 3) & ~(int64_t)3;
    staged = nnode_pad + rows * E <= STAGE;
    if (staged) {
      const int nnode = (int)rows * C, nedge = (int)rows * E, edge0 = (int)nnode_pad;
      const float* __restrict__ srcn = node_table + rmin * C;
      const float* __restrict__ srce = edge_table + rmin * E;
      // table element (row r, column f*N + n)  ->  staged position r*C + n*3 + f
      for (int i = threadIdx.x; i < nnode; i += 256) {
        const unsigned r = fast_div((unsigned)i, (unsigned)C, rcp_cols);
        const unsigned c = (unsigned)i - r * C;
        const unsigned f = fast_div(c, (unsigned)N, rcp_n);
        stage[r * C + (c - f * N) * 3 + f] = srcn[i];
      }
      for (int i = threadIdx.x; i < nedge; i += 256) stage[edge0 + i] = srce[i];
      if ((int)threadIdx.x < nwin) lrow[threadIdx.x] = (int)(row0[threadIdx.x] - rmin);
      __syncthreads();
      // one 16-byte store = EPV consecutive elements of window w from element o on; `per` elements per window, rows of
      // `cols` staged floats.  A run inside one window (all but ~EPV / per of them) is EPV contiguous staged floats.
      auto copy_windows = [&](const float* __restrict__ st, unsigned per, int cols, char* __restrict__ out, float rcp_per) {
        const unsigned total = (unsigned)nwin * per;
        const bool even = ((per | (unsigned)cols) & 1u) == 0;   // then every run starts at an even float: 8-byte LDS reads
        unsigned e0 = EPV * threadIdx.x;
        unsigned w = fast_div(e0, per, rcp_per);
        unsigned o = e0 - w * per;
        for (; e0 < total; e0 += EPV * 256) {
          float v[EPV];
          if (o + EPV <= per) {
            const int at = lrow[w] * cols + (int)o;
            const float* __restrict__ src = st + at;
            if ((at & 3) == 0) {   // 16-byte aligned (always, for step-1 windows of an even row width): ds_read_b128,
                                   // conflict-free for lanes 16 bytes apart
#pragma unroll
              for (int j = 0; j < EPV; j += 4) {
                const float4 q = *reinterpret_cast<const float4*>(src + j);
                v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
              }
            } else if (even) {
#pragma unroll
              for (int j = 0; j < EPV; j += 2) {
                const float2 q = *reinterpret_cast<const float2*>(src + j);
                v[j] = q.x; v[j + 1] = q.y;
              }
            } else {
#pragma unroll
              for (int j = 0; j < EPV; ++j) v[j] = src[j];
            }
          } else {   // the run crosses into the next window (or the end of the workgroup's output)
            unsigned wj = w, oj = o;
            int base = lrow[wj] * cols;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
              v[j] = st[base + (int)oj];
              if (++oj == per) {
                oj = 0;
                wj = wj + 1 < (unsigned)nwin ? wj + 1 : wj;
                base = lrow[wj] * cols;
              }
            }
          }
          store_run<OUT16>(out, e0, total, v);
          o += EPV * 256;
          while (o >= per) {
            o -= per;
            ++w;
          }
        }
      };
      copy_windows(stage, per_x, C, reinterpret_cast<char*>(x_out) + w0 * per_x * OSZ, rcp_perwin);
      copy_windows(stage + edge0, per_a, E, reinterpret_cast<char*>(a_out) + w0 * per_a * OSZ,
                   rcp_perwin * ((float)C / (float)E) * 0.999999f);
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
  const float rp = rcp_down((unsigned)(W * 3 * N)), rc = rcp_down((unsigned)(3 * N)), rn = rcp_down((unsigned)N);
  // staging-buffer class by the footprint of wb stride-1 windows (other spacings decide per workgroup)
  const int64_t need = (int64_t)(wb - 1 + W) * (3 * N + E);
#define GATHER1(IDX, ST, WBV)                                                                                        \
  do {                                                                                                               \
    if (out16)                                                                                                       \
      DOF_LAUNCH((k_window_gather<IDX, ST, WBV, true>), (blocks), (256), stream, node_table, edge_table, row_start,  \
                 first_row, row_step, n_windows, W, N, E, rp, rc, rn, x_out, a_out);                                 \
    else                                                                                                             \
      DOF_LAUNCH((k_window_gather<IDX, ST, WBV, false>), (blocks), (256), stream, node_table, edge_table, row_start, \
                 first_row, row_step, n_windows, W, N, E, rp, rc, rn, x_out, a_out);                                 \
  } while (0)
#define GATHER(IDX, ST)                                        \
  do {                                                         \
    if (small) GATHER1(IDX, ST, WB_SMALL); else GATHER1(IDX, ST, WB); \
  } while (0)
  if (need <= 3072) {
    if (row_start) GATHER(true, 3072); else GATHER(false, 3072);
  } else if (need <= 10240) {
    if (row_start) GATHER(true, 10240); else GATHER(false, 10240);
  } else {
    if (row_start) GATHER(true, 0); else GATHER(false, 0);
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
