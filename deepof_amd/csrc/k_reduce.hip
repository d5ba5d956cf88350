// Weight-gradient reductions and the optimiser.
//
//  * k_outer: every dense weight gradient of the model is a tall-skinny product
//        dW[i][j] = sum_{t,s} dG[t][i][s] * X[t+shift][j][s]
//    over SoA operands.  This is the one GEMM-shaped reduction of the step (K = T*S ~ 7e5), so it
//    runs on the matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32, same rounding as an fmaf chain).
//    Operands stream straight from HBM with arbitrary (time, sequence, channel) strides -- the
//    big activations are channel-minor [t][s][c], so a k-slice of 4 sequences x 16 channels is
//    4 contiguous 64-byte segments; one launch carries all jobs of the step.  Each workgroup
//    writes one partial tile; k_outer_finalize adds the partials in a fixed order, so gradients
//    are bitwise reproducible run to run (no float atomics).
//  * k_clip_adam: clip_grad_value_(0.75) + torch.optim.Adam on the flat parameter buffer
//    (/root/reference/deepof/clustering/training.py:162-166, losses.py:817-833).
#include "dof_rt.h"
#include "launchers.h"
#include "k_sum_partials.inc.h"

namespace {

__global__ void __launch_bounds__(256) k_outer(const DofOuterJob* __restrict__ jobs, int njobs,
                                               float* __restrict__ partials) {
  __shared__ float red[64][65];
  int jid = 0;
  for (int j = 0; j < njobs; ++j)
    if ((int)blockIdx.x >= jobs[j].blk0) jid = j;
  int blk = blockIdx.x - jobs[jid].blk0;
  if (jobs[jid].grp_jobs > 1) {   // operand-sharing group: see DofOuterJob
    const DofOuterJob& G0 = jobs[jid];
    const int v = blockIdx.x - G0.grp_blk0, x = v & 7, q = v >> 3;
    jid = G0.grp_job0 + q % G0.grp_jobs;
    blk = (q / G0.grp_jobs) * 8 + x;
  }
  const DofOuterJob& J = jobs[jid];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int T = J.T;
  const int64_t Sp = J.Sp;
  const int64_t chunks = Sp >> 4;  // 16 sequences per unit = 4 MFMA k-slices of 4 sequences
  const int64_t n_units = (int64_t)T * chunks;
  const int MT = (J.a_rows + 15) >> 4;
  const int NT = J.n_tiles;
  const int64_t stride = (int64_t)J.nblk * 4;

  dof_f32x4 acc[4][4];
  float rs[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    rs[a] = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }

  float av[4][4], bv[4][4];      // operands of the unit being multiplied   [tile][k-slice]
  float an[4][4], bn[4][4];      // operands of the next unit, in flight while the MFMAs run
  auto load_unit = [&](int64_t u, float (&A)[4][4], float (&B)[4][4]) {
    const int t = (int)(u / chunks);
    const int64_t s0 = ((u - (int64_t)t * chunks) << 4) + q;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) A[mt][kk] = 0.0f;
      const int row = mt * 16 + i;
      if (mt < MT && row < J.a_rows) {
        const float* __restrict__ ap = J.a_ptr + (int64_t)t * J.a_tstride + (int64_t)row * J.a_cstride + s0 * J.a_sstride;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) A[mt][kk] = ap[(int64_t)(4 * kk) * J.a_sstride];
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) B[nt][kk] = 0.0f;
      if (nt < NT) {
        const DofOuterTile& Bt = J.tile[nt];
        const int tap = Bt.pack > 0 ? i / Bt.pack : 0;
        const int ch = Bt.pack > 0 ? i - tap * Bt.pack : i;
        const int tb = t + Bt.shift + tap;
        if (tb >= 0 && tb < T && i < Bt.nc) {
          const float* __restrict__ bp = Bt.ptr + (int64_t)tb * Bt.t_stride + (int64_t)ch * Bt.c_stride + s0 * Bt.s_stride;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) B[nt][kk] = bp[(int64_t)(4 * kk) * Bt.s_stride];
        }
      }
    }
  };

  int64_t u = (int64_t)blk * 4 + wave;
  if (u < n_units) load_unit(u, av, bv);
  for (; u < n_units; u += stride) {
    const bool more = u + stride < n_units;
    if (more) load_unit(u + stride, an, bn);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < MT) rs[mt] += (av[mt][0] + av[mt][1]) + (av[mt][2] + av[mt][3]);
    // k-slices innermost.  (Making the k-slice the outer loop, so that consecutive MFMAs hit different accumulators,
    // was measured: k_outer -2 % on the TCN jobs, but the C2 step +0.55 % -- three runs each on one box -- so the
    // order stayed.)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < MT) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if (nt < NT) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][kk], bv[nt][kk], acc[mt][nt], 0, 0, 0);
          }
      }
    if (more) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          av[a][kk] = an[a][kk];
          bv[a][kk] = bn[a][kk];
        }
    }
  }
  // Sum the four waves' tiles through one LDS tile (fixed wave order => deterministic).
  // D layout of mfma_f32_16x16x4: lane holds rows (lane>>4)*4 + r, col lane&15.
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float r = rs[mt];
        r += __shfl_xor(r, 16);
        r += __shfl_xor(r, 32);
        if (q == 0) red[mt * 16 + i][64] = (w == 0 ? 0.0f : red[mt * 16 + i][64]) + r;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float* cell = &red[mt * 16 + q * 4 + r4][nt * 16 + i];
            *cell = (w == 0 ? 0.0f : *cell) + acc[mt][nt][r4];
          }
      }
    }
    __syncthreads();
  }
  float* __restrict__ out = partials + J.partial_off + (int64_t)blk * DOF_OUTER_PARTIAL_FLOATS;
  for (int e = threadIdx.x; e < 64 * 65; e += 256) {
    const int row = e / 65, col = e - row * 65;
    if (row < J.a_rows && (col < NT * 16 || col == 64)) out[e] = red[row][col];
  }
}

// Round 6: the same reduction on the bf16 matrix pipe.  k_outer is bound by the fp32 matrix instructions themselves
// (v_mfma_f32_16x16x4_f32: 256 FLOP per cycle and CU -- 16 jobs x 14 us at latent 6, 2 x 1.2 ms at latent 32).  Here a unit is 32
// sequences of one time step: the lane (row i, k group g) loads its row's values of the sequences 8 g .. 8 g + 7, cuts them
// into three bf16 pieces (value = p0 + p1 + p2 exactly) and a product is the six piece products that carry more than 2^-24
// of it, each one v_mfma_f32_16x16x32_bf16 (K = the 32 sequences): 6 x 16 cycles per 16 x 16 tile and 32 sequences against
// 8 x 32.  Same jobs, same partial tiles, same finalize; the sums differ from k_outer's in their order only.
__global__ void __launch_bounds__(256, 2) k_outer_b3(const DofOuterJob* __restrict__ jobs, int njobs,
                                                  float* __restrict__ partials) {
  __shared__ float red[64][65];
  int jid = 0;
  for (int j = 0; j < njobs; ++j)
    if ((int)blockIdx.x >= jobs[j].blk0) jid = j;
  int blk = blockIdx.x - jobs[jid].blk0;
  if (jobs[jid].grp_jobs > 1) {   // operand-sharing group: see DofOuterJob
    const DofOuterJob& G0 = jobs[jid];
    const int v = blockIdx.x - G0.grp_blk0, x = v & 7, q = v >> 3;
    jid = G0.grp_job0 + q % G0.grp_jobs;
    blk = (q / G0.grp_jobs) * 8 + x;
  }
  const DofOuterJob& J = jobs[jid];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int T = J.T;
  const int64_t Sp = J.Sp;
  const int64_t chunks = (Sp + 31) >> 5;  // 32 sequences per unit
  const int64_t n_units = (int64_t)T * chunks;
  const int MT = (J.a_rows + 15) >> 4;
  const int NT = J.n_tiles;
  const int64_t stride = (int64_t)J.nblk * 4;

  dof_f32x4 acc[4][4];
  float rs[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    rs[a] = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  float av[4][8], bv[4][8];   // the unit in flight: [tile][sequence of the lane's k group]
  auto load_a_row = [&](int64_t u, int mt) {
    const int t = (int)(u / chunks);
    const int64_t s0 = ((u - (int64_t)t * chunks) << 5) + 8 * kg;
    const bool in = s0 < Sp;   // (Sp is a multiple of 8: a k group is inside or outside as a whole)
#pragma unroll
    for (int j = 0; j < 8; ++j) av[mt][j] = 0.0f;
    const int row = mt * 16 + i;
    if (row < J.a_rows && in) {
      const float* __restrict__ ap = J.a_ptr + (int64_t)t * J.a_tstride + (int64_t)row * J.a_cstride + s0 * J.a_sstride;
#pragma unroll
      for (int j = 0; j < 8; ++j) av[mt][j] = ap[(int64_t)j * J.a_sstride];
    }
  };
  auto load_a = [&](int64_t u) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < MT) load_a_row(u, mt);
  };
  auto load_b = [&](int64_t u) {
    const int t = (int)(u / chunks);
    const int64_t s0 = ((u - (int64_t)t * chunks) << 5) + 8 * kg;
    const bool in = s0 < Sp;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[nt][j] = 0.0f;
      if (nt < NT) {
        const DofOuterTile& Bt = J.tile[nt];
        const int tap = Bt.pack > 0 ? i / Bt.pack : 0;
        const int ch = Bt.pack > 0 ? i - tap * Bt.pack : i;
        const int tb = t + Bt.shift + tap;
        if (tb >= 0 && tb < T && i < Bt.nc && in) {
          const float* __restrict__ bp = Bt.ptr + (int64_t)tb * Bt.t_stride + (int64_t)ch * Bt.c_stride + s0 * Bt.s_stride;
#pragma unroll
          for (int j = 0; j < 8; ++j) bv[nt][j] = bp[(int64_t)j * Bt.s_stride];
        }
      }
    }
  };
  auto cut = [&](const float (&v)[8], dof_bf16x8 (&p)[3]) {
    const float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
    uint32_t wl[3][2], wh[3][2];
    dof_split3x4(lo, wl);
    dof_split3x4(hi, wh);
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = dof_mk_bf16x8(wl[q][0], wl[q][1], wh[q][0], wh[q][1]);
  };

  int64_t u = (int64_t)blk * 4 + wave;
  if (u < n_units) {
    load_b(u);
    load_a(u);
  }
  for (; u < n_units; u += stride) {
    // B pieces for the whole unit, then one row block of A at a time (its pieces live for 24 instructions): 2 wavefronts per SIMD
    const bool more = u + stride < n_units;
    dof_bf16x8 pb[4][3];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (nt < NT) cut(bv[nt], pb[nt]);
    if (more) load_b(u + stride);   // in flight while the matrix instructions run
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < MT) {
        dof_bf16x8 pa[3];
        rs[mt] += ((av[mt][0] + av[mt][1]) + (av[mt][2] + av[mt][3])) + ((av[mt][4] + av[mt][5]) + (av[mt][6] + av[mt][7]));
        cut(av[mt], pa);
        if (more) load_a_row(u + stride, mt);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if (nt < NT) {
            dof_f32x4 c = acc[mt][nt];
            c = DOF_MFMA_16x16x32_BF16(pa[0], pb[nt][2], c);
            c = DOF_MFMA_16x16x32_BF16(pa[2], pb[nt][0], c);
            c = DOF_MFMA_16x16x32_BF16(pa[1], pb[nt][1], c);
            c = DOF_MFMA_16x16x32_BF16(pa[0], pb[nt][1], c);
            c = DOF_MFMA_16x16x32_BF16(pa[1], pb[nt][0], c);
            c = DOF_MFMA_16x16x32_BF16(pa[0], pb[nt][0], c);
            acc[mt][nt] = c;
          }
      }
  }
  // the four waves' tiles through one LDS tile, in wave order (k_outer's epilogue: same D layout)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float r = rs[mt];
        r += __shfl_xor(r, 16);
        r += __shfl_xor(r, 32);
        if (kg == 0) red[mt * 16 + i][64] = (w == 0 ? 0.0f : red[mt * 16 + i][64]) + r;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float* cell = &red[mt * 16 + kg * 4 + r4][nt * 16 + i];
            *cell = (w == 0 ? 0.0f : *cell) + acc[mt][nt][r4];
          }
      }
    }
    __syncthreads();
  }
  float* __restrict__ out = partials + J.partial_off + (int64_t)blk * DOF_OUTER_PARTIAL_FLOATS;
  for (int e = threadIdx.x; e < 64 * 65; e += 256) {
    const int row = e / 65, col = e - row * 65;
    if (row < J.a_rows && (col < NT * 16 || col == 64)) out[e] = red[row][col];
  }
}

// One 64-lane wavefront per output element: lanes stride over the job's per-workgroup partial
// tiles, then a fixed-shape butterfly adds the 64 lane sums (deterministic).
__global__ void __launch_bounds__(256) k_outer_finalize(const DofOuterJob* __restrict__ jobs,
                                                        const DofFinJob* __restrict__ fin, int n_fin, int total,
                                                        const float* __restrict__ partials, float* __restrict__ grads,
                                                        int accumulate) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (e >= total) return;
  int lo = 0, hi = n_fin - 1;
  while (lo < hi) {  // last fin-job with elem0 <= e
    const int mid = (lo + hi + 1) >> 1;
    if (fin[mid].elem0 <= e) lo = mid; else hi = mid - 1;
  }
  const DofFinJob& F = fin[lo];
  const DofOuterJob& J = jobs[F.job];
  const int local = e - F.elem0;
  const int ri = local / F.cols, ci = local - ri * F.cols;
  int src_row = ri < F.r1 ? ri : ri + (F.r2 - F.r1);
  if (F.gate_minor > 0) src_row = (src_row % F.gate_minor) * 4 + src_row / F.gate_minor;
  const float* __restrict__ p = partials + J.partial_off + (int64_t)src_row * 65 + F.col0 + ci;
  float acc = 0.0f;
  for (int b = lane; b < J.nblk; b += 64) acc += p[(int64_t)b * DOF_OUTER_PARTIAL_FLOATS];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (lane == 0) {
    float* dst = grads + F.dst_off + (int64_t)ri * F.row_stride + (int64_t)ci * F.col_stride;
    *dst = accumulate ? *dst + acc : acc;
  }
}

// one workgroup per output value: strided partial sums + fixed-shape LDS tree (deterministic)
__global__ void __launch_bounds__(256) k_sum_partials(const float* __restrict__ partial, int64_t nblk, int nv,
                                                      float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  const int v = blockIdx.x;
  float acc = 0.0f;
  for (int64_t b = threadIdx.x; b < nblk; b += 256) acc += partial[b * nv + v];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[v] = accumulate ? out[v] + red[0] : red[0];
}

__global__ void __launch_bounds__(256) k_sum_partials_multi(DofSumJobs J, int accumulate) {
  __shared__ float red[256];
  dof_sum_partials_multi_body(J, (int)blockIdx.x, accumulate, red);
}

// clip_grad_value_ + Adam on the flat buffer.  The Adam step count t of every optimiser segment lives on the device
// (opt_state, torch.optim.Adam keeps `step` per parameter the same way), which makes a captured step replayable
// without any host-written per-step value: a segment that receives gradients this step (hyper[active] != 0)
// updates with t + 1; every workgroup derives the bias corrections 1 - beta^(t+1) of the <= 8 segments itself
// (double pow on one lane per segment, through LDS), and the LAST workgroup to finish -- elected by an integer
// ticket, after every workgroup has read the counters -- stores the advanced counters.  (Was a separate one-wave
// k_adam_tick launch in front of this kernel.)
constexpr int kMaxAdamSegs = 8;
__global__ void __launch_bounds__(256) k_clip_adam(float* __restrict__ params, const float* __restrict__ grads,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ hyper,
                                                   const DofAdamSeg* __restrict__ segs, int nseg, int64_t total,
                                                   int clip_index, const float* __restrict__ mask,
                                                   int* opt_state, int* ticket, float grad_scale) {
  __shared__ float s_bc[2 * kMaxAdamSegs];
  __shared__ int s_t[kMaxAdamSegs];
  __shared__ int s_last;
  if ((int)threadIdx.x < nseg) {
    const int k = threadIdx.x;
    int t = opt_state[k];
    if (hyper[segs[k].active_index] != 0.0f) ++t;
    s_t[k] = t;
    if (t < 1) t = 1;
    s_bc[2 * k] = (float)(1.0 - pow(0.9, (double)t));
    s_bc[2 * k + 1] = (float)(1.0 - pow(0.999, (double)t));
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int sg = -1;
  if (i < total && mask[i] != 0.0f) {  // mask 0: the parameter never receives a gradient in the reference (grad None)
    for (int k = 0; k < nseg; ++k)
      if (i >= segs[k].lo && i < segs[k].hi) sg = k;
  }
  if (sg >= 0) {
    const DofAdamSeg S = segs[sg];
    if (hyper[S.active_index] != 0.0f) {  // (0: grad is None in the reference -> parameter skipped)
      const float clip = hyper[clip_index];
      const float wd = hyper[clip_index + 1];
      float g = grads[i] * grad_scale;  // data parallel: the all-reduced SUM times 1 / world = DDP's averaged gradient
      if (clip > 0.0f) g = fminf(fmaxf(g, -clip), clip);
      float p = params[i];
      if (wd != 0.0f) g = fmaf(wd, p, g);
      const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
      const float mi = fmaf(b1, m[i], (1.0f - b1) * g);
      const float vi = fmaf(b2, v[i], (1.0f - b2) * g * g);
      m[i] = mi;
      v[i] = vi;
      const float lr = hyper[S.lr_index];
      const float bc1 = s_bc[2 * sg], bc2 = s_bc[2 * sg + 1];
      const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
      params[i] = p - (lr / bc1) * (mi / denom);
    }
  }
  // every workgroup has consumed opt_state by now (s_t went through the barrier above); the last ticket holder
  // publishes the new counters.  No data travels between workgroups, so the ticket needs no fence.
  if (threadIdx.x == 0) {
    const int tk = atomicAdd(ticket, 1);
    s_last = tk == (int)gridDim.x - 1;
    if (s_last) *ticket = 0;
  }
  __syncthreads();
  if (s_last && (int)threadIdx.x < nseg) opt_state[threadIdx.x] = s_t[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Head of a captured step: schedule items + the step's Gaussian noise in one launch.
// Philox-4x32-10 (Salmon, Moraes, Dror, Shaw 2011): counter (quad index, buffer, step, 0), key = seed; the four
// 32-bit outputs become four N(0,1) values by Box-Muller on 24-bit uniforms in (0, 1).  The step index is a
// device counter read by every workgroup and advanced by the last one to finish (same ticket as k_clip_adam).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dof_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                  uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void __launch_bounds__(256) k_step_begin(float* __restrict__ hyper, DofSchedItems items, DofNoiseArgs N) {
  __shared__ int s_last;
  const uint32_t step = N.state ? (uint32_t)N.state[0] : 0u;
  if (blockIdx.x == 0 && (int)threadIdx.x < items.n) {
    const DofSchedItem it = items.item[threadIdx.x];
    int c = *it.cursor;
    const int at = c < it.len - 1 ? c : it.len - 1;
    hyper[it.hyper_index] = it.scale * it.table[at < 0 ? 0 : at];
    if (it.advance) *it.cursor = c + 1;
  }
  if (!N.state) return;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int buf = gid >= N.quads0 ? 1 : 0;
  const int64_t q = buf ? gid - N.quads0 : gid;
  if (4 * q < N.n[buf]) {
    uint32_t r[4];
    dof_philox4x32_10((uint32_t)q, (uint32_t)buf, step, 0u, N.key0, N.key1, r);
    float g[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-8f;      // 2^-24
      const float u2 = ((float)(r[2 * h + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
      const float rad = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      g[2 * h] = rad * cs;
      g[2 * h + 1] = rad * sn;
    }
    float* __restrict__ o = N.out[buf] + 4 * q;
    const int64_t left = N.n[buf] - 4 * q;
    if (left >= 4 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(g[0], g[1], g[2], g[3]);
    } else {
      for (int k = 0; k < 4 && k < left; ++k) o[k] = g[k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tk = atomicAdd(N.state + 1, 1);
    s_last = tk == (int)gridDim.x - 1;
    if (s_last) {
      N.state[1] = 0;
      N.state[0] = (int)(step + 1u);
    }
  }
}

// hyper[item.hyper_index] = scale * table[min(cursor, len - 1)]; cursor advances when item.advance (one thread per
// item, every item owns its cursor).  See dof_schedule_apply in deepof_hip.h.
__global__ void k_schedule_apply(float* __restrict__ hyper, DofSchedItems items) {
  const int k = threadIdx.x;
  if (k >= items.n) return;
  const DofSchedItem it = items.item[k];
  int c = *it.cursor;
  const int at = c < it.len - 1 ? c : it.len - 1;
  hyper[it.hyper_index] = it.scale * it.table[at < 0 ? 0 : at];
  if (it.advance) *it.cursor = c + 1;
}

}  // namespace

int dof_launch_outer(const DofOuterJob* jobs_dev, int njobs, int total_blocks, float* partials, hipStream_t st) {
  if (total_blocks <= 0) return DOF_OK;
  // DOF_OUTER_B3=0: the fp32 matrix instructions (k_outer) instead of the bf16-piece kernel
  static const int b3 = [] {
    const char* e = getenv("DOF_OUTER_B3");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  if (b3) DOF_LAUNCH(k_outer_b3, ((unsigned)total_blocks), (256), st, jobs_dev, njobs, partials);
  else DOF_LAUNCH(k_outer, ((unsigned)total_blocks), (256), st, jobs_dev, njobs, partials);
  return dof_check_launch("k_outer");
}

int dof_launch_outer_finalize(const DofOuterJob* jobs_dev, const DofFinJob* fin_dev, int n_fin, int total_elems,
                              const float* partials, float* grads, int accumulate, hipStream_t st) {
  if (total_elems <= 0) return DOF_OK;
  DOF_LAUNCH(k_outer_finalize, (dof_cdiv(total_elems, 4)), (256), st, jobs_dev, fin_dev, n_fin, total_elems,
             partials, grads, accumulate);
  return dof_check_launch("k_outer_finalize");
}

int dof_launch_sum_partials(const float* partial, int64_t nblk, int nv, float* out, int accumulate, hipStream_t st) {
  DOF_LAUNCH(k_sum_partials, ((unsigned)nv), (256), st, partial, nblk, nv, out, accumulate);
  return dof_check_launch("k_sum_partials");
}

int dof_launch_sum_partials_multi(const DofSumJobs& jobs, int accumulate, hipStream_t st) {
  int total = 0;
  for (int j = 0; j < jobs.n; ++j) total += jobs.nv[j];
  if (total == 0) return DOF_OK;
  DOF_LAUNCH(k_sum_partials_multi, ((unsigned)total), (256), st, jobs, accumulate);
  return dof_check_launch("k_sum_partials_multi");
}

int dof_launch_clip_adam(float* params, const float* grads, float* m, float* v, const float* hyper,
                         const DofAdamSeg* segs_dev, int nseg, int64_t total, int clip_index, const float* mask,
                         int* opt_state, int* ticket, float grad_scale, hipStream_t st) {
  if (nseg > kMaxAdamSegs) {
    dof_set_error("k_clip_adam: %d optimiser segments, at most %d", nseg, kMaxAdamSegs);
    return DOF_ERR_ARG;
  }
  DOF_LAUNCH(k_clip_adam, (dof_cdiv(total, 256)), (256), st, params, grads, m, v, hyper, segs_dev, nseg, total,
             clip_index, mask, opt_state, ticket, grad_scale);
  return dof_check_launch("k_clip_adam");
}

int dof_launch_step_begin(float* hyper, const DofSchedItems& items, const DofNoiseArgs& noise, hipStream_t st) {
  int64_t quads = 0;
  if (noise.state) quads = (noise.n[0] + 3) / 4 + (noise.n[1] + 3) / 4;
  const unsigned nb = quads > 0 ? dof_cdiv(quads, 256) : 1u;
  DOF_LAUNCH(k_step_begin, (nb), (256), st, hyper, items, noise);
  return dof_check_launch("k_step_begin");
}

int dof_launch_schedule_apply(float* hyper, const DofSchedItems& items, hipStream_t st) {
  DOF_LAUNCH(k_schedule_apply, (1), (64), st, hyper, items);
  return dof_check_launch("k_schedule_apply");
}
