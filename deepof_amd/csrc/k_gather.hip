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
// Mapping: one thread = one 16-byte store (4 consecutive output floats, coalesced, streaming /
// non-temporal so the 5.6 KB-per-window write stream does not evict the small, heavily re-read
// frame rows from L2); the 4 source words come from the (cached) frame rows.
#include "dof_rt.h"
#include "deepof_hip.h"

namespace {

template <bool INDEXED>
__global__ void __launch_bounds__(256) k_window_gather(
    const float* __restrict__ node_table, const float* __restrict__ edge_table,
    const int64_t* __restrict__ row_start, int64_t first_row, int64_t row_step, int64_t n_windows,
    int W, int N, int E, float* __restrict__ x_out, float* __restrict__ a_out) {
  const int C = 3 * N;
  const int64_t per_win_x = (int64_t)W * C;
  const int64_t per_win_a = (int64_t)W * E;
  const int64_t total_x = n_windows * per_win_x;
  const int64_t total_a = n_windows * per_win_a;
  const int64_t quads_x = (total_x + 3) >> 2;
  const int64_t quads_a = (total_a + 3) >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads_x + quads_a; q += stride) {
    const bool is_x = q < quads_x;
    const int64_t base = (is_x ? q : q - quads_x) << 2;
    const int64_t total = is_x ? total_x : total_a;
    const int64_t per_win = is_x ? per_win_x : per_win_a;
    const int cols = is_x ? C : E;
    const float* __restrict__ table = is_x ? node_table : edge_table;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = base + j;
      float val = 0.0f;
      if (i < total) {
        const int64_t b = i / per_win;
        const int o = (int)(i - b * per_win);
        const int t = o / cols;
        const int r = o - t * cols;
        int src_col = r;
        if (is_x) {
          const int n = r / 3;
          const int f = r - 3 * n;
          src_col = f * N + n;
        }
        const int64_t row = (INDEXED ? row_start[b] : first_row + b * row_step) + t;
        val = table[row * cols + src_col];
      }
      v[j] = val;
    }
    float* __restrict__ out = is_x ? x_out : a_out;
    if (base + 3 < total) {
#ifdef DOF_EMU
      out[base] = v[0]; out[base + 1] = v[1]; out[base + 2] = v[2]; out[base + 3] = v[3];
#else
      dof_f32x4 pack = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(pack, reinterpret_cast<dof_f32x4*>(out + base));
#endif
    } else {
      for (int j = 0; j < 4; ++j)
        if (base + j < total) out[base + j] = v[j];
    }
  }
}

int launch_gather(const float* node_table, const float* edge_table, const int64_t* row_start, int64_t first_row,
                  int64_t row_step, int64_t n_windows, int W, int N, int E, float* x_out, float* a_out,
                  hipStream_t stream) {
  if (!node_table || !edge_table || !x_out || !a_out || n_windows < 0 || W <= 0 || N <= 0 || E <= 0) {
    dof_set_error("dof_window_gather: bad argument");
    return DOF_ERR_ARG;
  }
  if (n_windows == 0) return DOF_OK;
  const int64_t quads = (n_windows * (int64_t)W * (3 * N + E) + 3) / 4 + 1;
  int64_t blocks = (quads + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;  // 16 workgroups per CU, grid-stride the rest
  if (row_start)
    DOF_LAUNCH(k_window_gather<true>, ((unsigned)blocks), (256), stream, node_table, edge_table, row_start, first_row,
               row_step, n_windows, W, N, E, x_out, a_out);
  else
    DOF_LAUNCH(k_window_gather<false>, ((unsigned)blocks), (256), stream, node_table, edge_table, row_start, first_row,
               row_step, n_windows, W, N, E, x_out, a_out);
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
