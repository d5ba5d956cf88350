// Matrix-pipe GRU recurrences of the encoder streams at latent 8: GRU(16 -> 16) and GRU(32 -> 8), forward and backward,
// on v_mfma_f32_16x16x32_bf16 with exact three-piece operands (reference:
// /root/reference/deepof/clustering/models_new.py:184-278, torch.nn.GRU).  Included by k_rnn.hip (inside its anonymous
// namespace) and by the development probe tools/probe/gru16_probe.hip, so that what is timed in isolation is the
// product kernel text.
#pragma once
#ifndef GRU16_WG_FLOATS
#define GRU16_WG_FLOATS (6 * 256 + 4 * 16)
#endif
// One launch serves up to two independent layers of the same shape (blockIdx.z: the node and the edge stream of the
// encoder): a stream of 14,336 sequences is only 1.75 wavefronts per SIMD.
struct Gru16mStream {
  const float* X; const int* len;
  const float *wih0, *whh0, *bih0, *bhh0, *wih1, *whh1, *bih1, *bhh1;
  float* O; float* GS;          // forward: outputs, saved gates (or null)
  const float* dO; float* dX; float* wg_partial;   // backward only
  int64_t S, Sp;
};

// ---------------------------------------------------------------------------------------------
// GRU(16 -> 16) forward on the bf16 matrix pipe with EXACT three-piece operands (round 5).  Same lane mapping, loads,
// stores and gate arithmetic as k_gru16x_fwd; what changes is how G = W [x_t ; h_{t-1}] is multiplied.  Measured on
// MI355X (tools/probe/gru16_probe.hip): v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate, costs 40 cycles in these
// dependent chains and does not overlap the gate arithmetic of any wavefront of its SIMD -- the 24 of a step were 2/3 of
// the kernel.  Here every fp32 value is cut into three bf16 pieces (v = p0 + p1 + p2 exactly: 8 + 8 + 8 significand
// bits, dof_split3x4) and the product is the sum of the six piece products that carry more than 2^-24 of it,
//   W v = W0 v0 + (W0 v1 + W1 v0) + (W0 v2 + W2 v0 + W1 v1)      (dropped: W1 v2 + W2 v1 + W2 v2 < 2^-23 |W| |v|),
// each ONE v_mfma_f32_16x16x32_bf16 (K = 32, fp32 accumulation, 1/16 of the fp32 form's time per flop):
//   * gates r, z: K = the lane's own [x(4) ; h(4)] per K-block -- lane (b, j) holds x[4b..4b+3] and h[4b..4b+3] of
//     sequence j, i.e. elements 8b .. 8b+7 of column j of B; A = the same K order of [W_ih | W_hh], three pieces = 12
//     registers per gate; six MFMAs per gate;
//   * gate n keeps its halves apart (n = tanh(W_in x + b_in + r (W_hn h + b_hn))): K = 32 holds TWO piece products of a
//     16-wide operand, A = (Wp | Wq), B = (vr ; vs): (W0|W0)(v0;v1), (W1|W0)(v0;v2), (W2|W1)(v0;v1) -- three MFMAs per
//     half; the input half is issued one step ahead.
// 18 MFMAs x 16 cycles per step instead of 24 x 40, plus ~45 VALU operations for the pieces of x and h.  The smallest
// products are accumulated first.  Deviation from the fp32 kernels: 2^-23-level relative differences of the
// pre-activations (the fp32 MFMA form rounds once per product, ~the same size).
// ---------------------------------------------------------------------------------------------
// Row addressing of the time loops without a 64-bit multiply per access (v_mul_lo_u32 / v_mad_u64_u32 run at a quarter
// of the rate: they were a fifth of the forward kernels' VALU time).  A [T][Sp][C] tensor is addressed by a 32-bit BYTE
// offset from its base (the launchers bound T * Sp * 32 * 4 < 2^31: dof_gru16_mfma): one time step = `str` bytes
// (wave-uniform), the lane's row at t = 0 = `lane` bytes, `rev` = its row n - 1 (row 0 for an empty sequence).  The forward
// direction walks t = step, the reverse direction t = n - 1 - step; `step` is wave-uniform, so step * str is scalar work.
struct DofRowWalk {
  int str, lane, rev;
  __device__ __forceinline__ DofRowWalk(int64_t Sp, int C, int64_t s, int c0, int n)
      : str((int)(Sp * C * 4)), lane((int)((s * C + c0) * 4)), rev(lane + (n > 0 ? n - 1 : 0) * str) {}
  // a row that is VALID to read for every lane -- the step's row where the lane is active, some row of the tensor where
  // it is not (its value is not used); `step` may lie outside [0, T) (prefetches past either end)
  __device__ __forceinline__ int read(int dir, int step, int T) const {
    const int st = step < 0 ? 0 : (step < T ? step : T - 1);
    const int r = rev - st * str;
    return dir ? (r > lane ? r : lane) : lane + st * str;
  }
  // the row before it in processing order (t - 1 forward, t + 1 reverse: h_{t-1} of a backward step); at step 0 -- where
  // the caller substitutes zeros -- still a valid row
  __device__ __forceinline__ int read_prev(int dir, int step, int T) const {
    const int st = step < 1 ? 1 : (step < T ? step : T - 1);
    const int r = rev + str - st * str;
    return dir ? (r > lane ? r : lane) : lane + (st - 1) * str;
  }
  // the row an output of `step` goes to: the step's row where the lane is active, row `step` (which must hold zeros:
  // rows t >= n are zero) where it is not
  __device__ __forceinline__ int write(int dir, int step, bool act) const {
    return (act && dir) ? rev - step * str : lane + step * str;
  }
};
__device__ __forceinline__ const float* dof_at(const float* base, int byte_off) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (uint32_t)byte_off);
}
__device__ __forceinline__ float* dof_at(float* base, int byte_off) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (uint32_t)byte_off);
}

// SAVE: write the (r, z, n, W_hn h + b_hn) words of every unit (the layout k_gru3_fwd<16,16> saves) -- compile-time, so
// that the time loop has NO branch: with a branch around a store (or a load) hipcc can no longer count the outstanding
// memory operations and waits vmcnt(0) every step, i.e. for the previous step's stores.  All stores are unconditional:
// a finished (or out-of-range) lane writes the zeros its rows t >= n need anyway (pad rows hold zeros already).
template <int WPE, bool SAVE>
__global__ void __launch_bounds__(64, WPE) k_gru16x_fwd(Gru16mStream sa, Gru16mStream sb, int T) {
  constexpr int HID = 16, IN = 16;
  const Gru16mStream& A = blockIdx.z ? sb : sa;
  const float* __restrict__ X = A.X;
  const int* __restrict__ len = A.len;
  float* __restrict__ O = A.O;
  float* __restrict__ GS = A.GS;
  const int64_t S = A.S, Sp = A.Sp;
  if ((int64_t)blockIdx.x * 16 >= S) return;   // (the grid covers the longer stream)
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, b = lane >> 4;
  const int64_t s = (int64_t)blockIdx.x * 16 + j;
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const float* __restrict__ wih = dir ? A.wih1 : A.wih0;
  const float* __restrict__ whh = dir ? A.whh1 : A.whh0;
  const float* __restrict__ bih = dir ? A.bih1 : A.bih0;
  const float* __restrict__ bhh = dir ? A.bhh1 : A.bhh0;
  // A operands: row = unit j, K-block b.  Gates r, z: elements [W_ih[.][4b..4b+3] | W_hh[.][4b..4b+3]], piece p.
  dof_bf16x8 a_rz[2][3], a_nx[3], a_hn[3];
  {
    uint32_t pi[3][3][2], ph[3][3][2];   // [gate][piece][word]
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float wi[4], wh[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        wi[q] = wih[(g * HID + j) * IN + 4 * b + q];
        wh[q] = whh[(g * HID + j) * HID + 4 * b + q];
      }
      dof_split3x4(wi, pi[g]);
      dof_split3x4(wh, ph[g]);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int p = 0; p < 3; ++p) a_rz[g][p] = dof_mk_bf16x8(pi[g][p][0], pi[g][p][1], ph[g][p][0], ph[g][p][1]);
    // n halves: (W0 | W0), (W1 | W0), (W2 | W1)
    a_nx[0] = dof_mk_bf16x8(pi[2][0][0], pi[2][0][1], pi[2][0][0], pi[2][0][1]);
    a_nx[1] = dof_mk_bf16x8(pi[2][1][0], pi[2][1][1], pi[2][0][0], pi[2][0][1]);
    a_nx[2] = dof_mk_bf16x8(pi[2][2][0], pi[2][2][1], pi[2][1][0], pi[2][1][1]);
    a_hn[0] = dof_mk_bf16x8(ph[2][0][0], ph[2][0][1], ph[2][0][0], ph[2][0][1]);
    a_hn[1] = dof_mk_bf16x8(ph[2][1][0], ph[2][1][1], ph[2][0][0], ph[2][0][1]);
    a_hn[2] = dof_mk_bf16x8(ph[2][2][0], ph[2][2][1], ph[2][1][0], ph[2][1][1]);
  }
  dof_f32x4 c_r, c_z, c_n, c_hn;  // biases in the D layout: register r <-> unit 4b + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int unit = 4 * b + r;
    c_r[r] = bih[unit] + bhh[unit];
    c_z[r] = bih[HID + unit] + bhh[HID + unit];
    c_n[r] = bih[2 * HID + unit];
    c_hn[r] = bhh[2 * HID + unit];
  }
  float* __restrict__ gs = GS ? GS + (int64_t)dir * T * 4 * HID * Sp : nullptr;
  const int n = in_range ? len[s] : 0;
  float h[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  uint32_t hw[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};   // pieces of h
  // x_t is loaded PF steps ahead into static register slots (the loop is unrolled by PF): a rotating copy made the
  // compiler wait `vmcnt(0)` at every loop back-edge -- one exposed HBM round trip (~1.8 us) per time step.  Loads are
  // unconditional (lanes past the end and finished sequences read a valid row): a branch around a load costs a
  // vmcnt(0) at the join.
  constexpr int PF = 4;
  float xs[PF][4];
  const int64_t sr = in_range ? s : S - 1;
  const DofRowWalk wx(Sp, IN, sr, 4 * b, n), wo(Sp, 2 * HID, s, dir * HID + 4 * b, n);
  auto load_x = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    dof_ld_row<4>(dof_at(X, wx.read(dir, step, T)), xs[slot]);
  };
  uint32_t xw[3][2];   // pieces of the CURRENT step's x
  dof_f32x4 g_n;       // W_in x + b_in of the current step
  auto input_part = [&](const float (&xq)[4]) DOF_INLINE_LAMBDA {
    dof_split3x4(xq, xw);
    g_n = c_n;
    g_n = DOF_MFMA_16x16x32_BF16(a_nx[2], dof_mk_bf16x8(xw[0][0], xw[0][1], xw[1][0], xw[1][1]), g_n);
    g_n = DOF_MFMA_16x16x32_BF16(a_nx[1], dof_mk_bf16x8(xw[0][0], xw[0][1], xw[2][0], xw[2][1]), g_n);
    g_n = DOF_MFMA_16x16x32_BF16(a_nx[0], dof_mk_bf16x8(xw[0][0], xw[0][1], xw[1][0], xw[1][1]), g_n);
  };
  auto do_step = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {   // slot = step % PF holds x of this step; xw / g_n are ready
    constexpr int slot = decltype(slot_c)::value;
    constexpr int next = (slot + 1) % PF;
    dof_f32x4 a_r = c_r, a_z = c_z, a_h = c_hn;
    const dof_f32x4 a_n = g_n;
    const dof_bf16x8 b0 = dof_mk_bf16x8(xw[0][0], xw[0][1], hw[0][0], hw[0][1]);
    const dof_bf16x8 b1 = dof_mk_bf16x8(xw[1][0], xw[1][1], hw[1][0], hw[1][1]);
    const dof_bf16x8 b2 = dof_mk_bf16x8(xw[2][0], xw[2][1], hw[2][0], hw[2][1]);
    const dof_bf16x8 h01 = dof_mk_bf16x8(hw[0][0], hw[0][1], hw[1][0], hw[1][1]);
    const dof_bf16x8 h02 = dof_mk_bf16x8(hw[0][0], hw[0][1], hw[2][0], hw[2][1]);
    // smallest products first; three independent accumulator chains, issued round robin
    a_r = DOF_MFMA_16x16x32_BF16(a_rz[0][0], b2, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(a_rz[1][0], b2, a_z);
    a_h = DOF_MFMA_16x16x32_BF16(a_hn[2], h01, a_h);
    a_r = DOF_MFMA_16x16x32_BF16(a_rz[0][2], b0, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(a_rz[1][2], b0, a_z);
    a_h = DOF_MFMA_16x16x32_BF16(a_hn[1], h02, a_h);
    a_r = DOF_MFMA_16x16x32_BF16(a_rz[0][1], b1, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(a_rz[1][1], b1, a_z);
    a_h = DOF_MFMA_16x16x32_BF16(a_hn[0], h01, a_h);
    a_r = DOF_MFMA_16x16x32_BF16(a_rz[0][0], b1, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(a_rz[1][0], b1, a_z);
    a_r = DOF_MFMA_16x16x32_BF16(a_rz[0][1], b0, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(a_rz[1][1], b0, a_z);
    a_r = DOF_MFMA_16x16x32_BF16(a_rz[0][0], b0, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(a_rz[1][0], b0, a_z);
    input_part(xs[next]);          // next step's pieces of x and its W_in x (x arrived PF - 1 steps ago)
    load_x(slot_c, step + PF);     // this step's slot is free again
    const bool act = step < n;
    float gate16[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rr = dof_sigmoid(a_r[r]);
      const float zz = dof_sigmoid(a_z[r]);
      const float nn = dof_tanh(fmaf(rr, a_h[r], a_n[r]));
      const float hnew = fmaf(zz, h[r] - nn, nn);
      h[r] = act ? hnew : h[r];
      gate16[4 * r] = rr; gate16[4 * r + 1] = zz; gate16[4 * r + 2] = nn; gate16[4 * r + 3] = a_h[r];
    }
    dof_split3x4(h, hw);
    // rows t >= n are zero: a finished lane writes the zero row of time `step` (>= n), which nobody else writes
    float hout[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hout[r] = act ? h[r] : 0.0f;
    dof_st_row<4>(dof_at(O, wo.write(dir, step, act)), hout);
    if constexpr (SAVE) {
      const int t = act ? (dir ? (n - 1 - step) : step) : step;
#pragma unroll
      for (int e = 0; e < 16; ++e) gate16[e] = act ? gate16[e] : 0.0f;
      dof_st_row<16>(gs + ACT(t, 16 * b, 4 * HID, Sp, s), gate16);
    }
    DOF_SCHED_FENCE();   // keep each step's load and stores in the step (the scheduler gathered them at the loop end)
  };
  dof_static_for<PF>([&](auto d) { load_x(d, decltype(d)::value); });
  input_part(xs[0]);
  int step = 0;
  for (; step + PF <= T; step += PF) {  // wave-uniform trip count (MFMA ignores EXEC, finished sequences idle); no branch inside
    dof_static_for<PF>([&](auto d) { do_step(d, step + decltype(d)::value); });
  }
  dof_static_for<PF - 1>([&](auto d) {   // the last T % PF steps
    if (step + decltype(d)::value < T) do_step(d, step + decltype(d)::value);
  });
}

// ---------------------------------------------------------------------------------------------
// GRU(32 -> 8) forward on the bf16 matrix pipe with exact three-piece operands (round 5): the second encoder layer of
// latent 8, the twin of k_gru16x_fwd in round 4's lane mapping.  Lane (b, j) owns units 2b and 2b+1 of sequence j;
// the two 16-row tiles are ordered by owner, D rows 4b + (0, 1, 2, 3) =
//   tile 1: r_{2b}, r_{2b+1}, z_{2b}, z_{2b+1}       tile 2: nx_{2b}, nx_{2b+1}, hn_{2b}, hn_{2b+1}
// (nx = W_in x + b_in, hn = W_hn h + b_hn; the A operand holds zeros where a row does not take that half of [x ; h]).
//   * input part: K-block b = the lane's own x[8b .. 8b+7] (two 16-byte loads), one piece per MFMA: six piece products
//     per tile, issued one step ahead;
//   * hidden part: K = 8 units leaves room for FOUR piece products in one K = 32 instruction -- the lane's own
//     (h0, h1, h2, h0) pairs against (W0, W0, W0, W1), and (h0, h1, 0, 0) against (W2, W1, 0, 0): two MFMAs per tile on the
//     recurrence's critical path.
// 16 MFMAs per step instead of 20 fp32 ones at 2.5 x the time each.  NO gates are saved (round 4 wrote 183 MB of them per
// step at C2; the forward pass was bound by those stores): k_gru8x_bwd recomputes them with the same MFMAs.
// ---------------------------------------------------------------------------------------------
struct Gru8xOperands {   // A operands of one lane, shared by the forward and the backward kernel
  dof_bf16x8 x1[3], x2[3];   // input part, tiles 1 / 2, pieces 0..2
  dof_bf16x8 h1a, h1b, h2a, h2b;
};
__device__ __forceinline__ void dof_gru8x_operands(const float* __restrict__ wih, const float* __restrict__ whh, int lane,
                                                   Gru8xOperands& A) {
  constexpr int HID = 8, IN = 32;
  const int i = lane & 15, kb = lane >> 4;
  const int unit = 2 * (i >> 2) + (i & 1), hi = (i >> 1) & 1;   // hi: z (tile 1) / hn (tile 2)
  float v1[8], v2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v1[e] = wih[((hi ? 1 : 0) * HID + unit) * IN + 8 * kb + e];
    v2[e] = hi ? 0.0f : wih[(2 * HID + unit) * IN + 8 * kb + e];
  }
  uint32_t p1[2][3][2], p2[2][3][2];
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v1[0]), p1[0]);
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v1[4]), p1[1]);
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v2[0]), p2[0]);
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v2[4]), p2[1]);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    A.x1[p] = dof_mk_bf16x8(p1[0][p][0], p1[0][p][1], p1[1][p][0], p1[1][p][1]);
    A.x2[p] = dof_mk_bf16x8(p2[0][p][0], p2[0][p][1], p2[1][p][0], p2[1][p][1]);
  }
  // hidden part: the two hidden units 2 kb, 2 kb + 1 of this K-block
  float h1v[4], h2v[4];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    h1v[m] = whh[((hi ? 1 : 0) * HID + unit) * HID + 2 * kb + m];
    h2v[m] = hi ? whh[(2 * HID + unit) * HID + 2 * kb + m] : 0.0f;
    h1v[2 + m] = 0.0f; h2v[2 + m] = 0.0f;
  }
  uint32_t q1[3][2], q2[3][2];
  dof_split3x4(h1v, q1);   // word [p][0] = pieces p of the two weights
  dof_split3x4(h2v, q2);
  A.h1a = dof_mk_bf16x8(q1[0][0], q1[0][0], q1[0][0], q1[1][0]);   // x (h0, h1, h2, h0)
  A.h1b = dof_mk_bf16x8(q1[2][0], q1[1][0], 0u, 0u);               // x (h0, h1, 0, 0)
  A.h2a = dof_mk_bf16x8(q2[0][0], q2[0][0], q2[0][0], q2[1][0]);
  A.h2b = dof_mk_bf16x8(q2[2][0], q2[1][0], 0u, 0u);
}
// the three pieces of the lane's two hidden values as packed words (h0, h1, h2)
__device__ __forceinline__ void dof_split3x2(float a, float b, uint32_t (&w)[3]) {
  const float ra = dof_bf16_rest(a), rb = dof_bf16_rest(b);
  w[0] = dof_pack_hi16(a, b);
  w[1] = dof_pack_hi16(ra, rb);
  w[2] = dof_pack_hi16(dof_bf16_rest(ra), dof_bf16_rest(rb));
}

template <int WPE>
__global__ void __launch_bounds__(64, WPE) k_gru8x_fwd(Gru16mStream sa, Gru16mStream sb, int T) {
  constexpr int HID = 8, IN = 32;
  const Gru16mStream& A = blockIdx.z ? sb : sa;
  const float* __restrict__ X = A.X;
  const int* __restrict__ len = A.len;
  float* __restrict__ O = A.O;
  const int64_t S = A.S, Sp = A.Sp;
  if ((int64_t)blockIdx.x * 16 >= S) return;   // (the grid covers the longer stream)
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, b = lane >> 4;
  const int64_t s = (int64_t)blockIdx.x * 16 + j;
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const float* __restrict__ bih = dir ? A.bih1 : A.bih0;
  const float* __restrict__ bhh = dir ? A.bhh1 : A.bhh0;
  Gru8xOperands W;
  dof_gru8x_operands(dir ? A.wih1 : A.wih0, dir ? A.whh1 : A.whh0, lane, W);
  dof_f32x4 c1, c2;  // biases in the D layout of this lane's two units
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int unit = 2 * b + m;
    c1[m] = bih[unit] + bhh[unit];
    c1[2 + m] = bih[HID + unit] + bhh[HID + unit];
    c2[m] = bih[2 * HID + unit];
    c2[2 + m] = bhh[2 * HID + unit];
  }
  const int n = in_range ? len[s] : 0;
  float h[2] = {0.0f, 0.0f};
  uint32_t hw[3] = {0u, 0u, 0u};
  constexpr int PF = 4;   // x_t loaded PF steps ahead into static register slots (see k_gru16x_fwd)
  float xs[PF][8];
  const int64_t sr = in_range ? s : S - 1;
  const DofRowWalk wx(Sp, IN, sr, 8 * b, n), wo(Sp, 2 * HID, s, dir * HID + 2 * b, n);
  auto load_x = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    dof_ld_row<8>(dof_at(X, wx.read(dir, step, T)), xs[slot]);
  };
  dof_f32x4 g1, g2;   // input parts of the current step
  auto input_part = [&](const float (&xq)[8]) DOF_INLINE_LAMBDA {
    uint32_t lo[3][2], hi[3][2];
    dof_split3x4(reinterpret_cast<const float(&)[4]>(xq[0]), lo);
    dof_split3x4(reinterpret_cast<const float(&)[4]>(xq[4]), hi);
    const dof_bf16x8 b0 = dof_mk_bf16x8(lo[0][0], lo[0][1], hi[0][0], hi[0][1]);
    const dof_bf16x8 b1 = dof_mk_bf16x8(lo[1][0], lo[1][1], hi[1][0], hi[1][1]);
    const dof_bf16x8 b2 = dof_mk_bf16x8(lo[2][0], lo[2][1], hi[2][0], hi[2][1]);
    g1 = c1; g2 = c2;   // smallest products first
    g1 = DOF_MFMA_16x16x32_BF16(W.x1[0], b2, g1);
    g2 = DOF_MFMA_16x16x32_BF16(W.x2[0], b2, g2);
    g1 = DOF_MFMA_16x16x32_BF16(W.x1[2], b0, g1);
    g2 = DOF_MFMA_16x16x32_BF16(W.x2[2], b0, g2);
    g1 = DOF_MFMA_16x16x32_BF16(W.x1[1], b1, g1);
    g2 = DOF_MFMA_16x16x32_BF16(W.x2[1], b1, g2);
    g1 = DOF_MFMA_16x16x32_BF16(W.x1[0], b1, g1);
    g2 = DOF_MFMA_16x16x32_BF16(W.x2[0], b1, g2);
    g1 = DOF_MFMA_16x16x32_BF16(W.x1[1], b0, g1);
    g2 = DOF_MFMA_16x16x32_BF16(W.x2[1], b0, g2);
    g1 = DOF_MFMA_16x16x32_BF16(W.x1[0], b0, g1);
    g2 = DOF_MFMA_16x16x32_BF16(W.x2[0], b0, g2);
  };
  auto do_step = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int next = (slot + 1) % PF;
    dof_f32x4 a1 = g1, a2 = g2;
    const dof_bf16x8 ha = dof_mk_bf16x8(hw[0], hw[1], hw[2], hw[0]);
    const dof_bf16x8 hb = dof_mk_bf16x8(hw[0], hw[1], 0u, 0u);
    a1 = DOF_MFMA_16x16x32_BF16(W.h1b, hb, a1);
    a2 = DOF_MFMA_16x16x32_BF16(W.h2b, hb, a2);
    a1 = DOF_MFMA_16x16x32_BF16(W.h1a, ha, a1);
    a2 = DOF_MFMA_16x16x32_BF16(W.h2a, ha, a2);
    input_part(xs[next]);          // next step's input part (its x arrived PF - 1 steps ago)
    load_x(slot_c, step + PF);     // this step's slot is free again
    const bool act = step < n;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float rr = dof_sigmoid(a1[m]);
      const float zz = dof_sigmoid(a1[2 + m]);
      const float nn = dof_tanh(fmaf(rr, a2[2 + m], a2[m]));
      const float hnew = fmaf(zz, h[m] - nn, nn);
      h[m] = act ? hnew : h[m];
    }
    dof_split3x2(h[0], h[1], hw);
    // rows t >= n are zero: a finished lane writes the zero row of time `step` (>= n), which nobody else writes
    dof_st_pair(dof_at(O, wo.write(dir, step, act)), act ? h[0] : 0.0f, act ? h[1] : 0.0f);
    DOF_SCHED_FENCE();
  };
  dof_static_for<PF>([&](auto d) { load_x(d, decltype(d)::value); });
  input_part(xs[0]);
  int step = 0;
  for (; step + PF <= T; step += PF) {  // wave-uniform trip count (MFMA ignores EXEC, finished sequences idle); no branch inside
    dof_static_for<PF>([&](auto d) { do_step(d, step + decltype(d)::value); });
  }
  dof_static_for<PF - 1>([&](auto d) {   // the last T % PF steps
    if (step + decltype(d)::value < T) do_step(d, step + decltype(d)::value);
  });
}

// ---------------------------------------------------------------------------------------------
// GRU(32 -> 8) backward on the bf16 matrix pipe, gates RECOMPUTED (round 5): the twin of k_gru8x_fwd; replaces
// k_gru8_bwd_fused + the saved gates for the large launches (round 4's pair moved 640 MB of gates per C2 step).
// Lane (b, j) owns units 2b, 2b+1 of sequence j.  Per step:
//   * recompute: k_gru8x_fwd's 16 MFMAs in its order (bitwise the same r, z, n, hn) from x_t and h_{t-1};
//   * gate gradients of the lane's two units on the VALU; its eight values (g_r, g_z, g_n, g_h) x 2 ARE K-block b of
//     the transposed products: K = 8 units x 4 gradient kinds = 32, one piece per MFMA;
//   * [dh ; dx] = 40 output rows = three 16-row tiles ordered by owner -- tile A rows 4b + (0..3) = dh_{2b}, dh_{2b+1},
//     dx_{8b}, dx_{8b+1}; tile B = dx_{8b+2 .. 8b+5}; tile C = dx_{8b+6}, dx_{8b+7}, 0, 0 -- so a lane receives the dh of
//     its own units and its 32 contiguous bytes of dX: 18 MFMAs (six piece products per tile);
//   * weight gradients: contraction over the 16 sequences.  The step's 72 values per sequence go through a 7.5 KB LDS
//     image transposed and cut into pieces ([piece][value][sequence] bf16, written as 2-byte stores of register high
//     halves), and return as K = 32 = (two piece products) x (16 sequences) operands: row tiles [g_r ; g_z], [g_n ; g_h]
//     x column tiles x_0..15, x_16..31, [h ; 0], three MFMAs each = 18 into six accumulator tiles (round 5, first form:
//     24 v_mfma_f32_16x16x4_f32 through an fp32 tile -- a third of the kernel's time).
// The layer has no per-step output gradient: dh starts from dHfin (the final hidden state's gradient).
// Per-wavefront partials [dir][tile][GRU8X_WG_FLOATS]: the reference's tensors in order -- weight_ih (24 x 32),
// weight_hh (24 x 8), then the bias sums of g_r, g_z, g_n, g_h -> k_gru8x_wg_finalize.
// ---------------------------------------------------------------------------------------------
#define GRU8X_WG_FLOATS (24 * 32 + 24 * 8 + 4 * 8)
__global__ void __launch_bounds__(64, 2) k_gru8x_bwd(Gru16mStream st_a, Gru16mStream st_b, int T) {
  constexpr int HID = 8, IN = 32;
  // the step's values TRANSPOSED for the weight gradients: [piece][g_r(8) g_z(8) g_n(8) g_h(8) | x(32) | h(8) | 0(8)][sequence]
  // as bf16 -- a lane's MFMA operand (eight consecutive sequences of one row) is 16 contiguous bytes
  __shared__ __attribute__((aligned(16))) uint16_t tp[3][80][16];
  // the operands of tiles B and C (dx only: nothing waits for them) stay in LDS: 24 registers fewer
  __shared__ __attribute__((aligned(16))) uint32_t tbc[6][64][4];
  const Gru16mStream& A = blockIdx.z ? st_b : st_a;
  const float* __restrict__ X = A.X;
  const int* __restrict__ len = A.len;
  const float* __restrict__ O = A.O;
  const float* __restrict__ dHfin = A.dO;   // (final hidden state's gradient [2 * HID][Sp], or null)
  float* __restrict__ dX = A.dX;
  float* __restrict__ wg_partial = A.wg_partial;
  const int64_t S = A.S, Sp = A.Sp;
  if ((int64_t)blockIdx.x * 16 >= S) return;   // (the grid covers the longer stream)
  const unsigned nblk_own = (unsigned)((S + 15) / 16);
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, b = lane >> 4;
  const int64_t s = (int64_t)blockIdx.x * 16 + j;
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const float* __restrict__ wih = dir ? A.wih1 : A.wih0;
  const float* __restrict__ whh = dir ? A.whh1 : A.whh0;
  const float* __restrict__ bih = dir ? A.bih1 : A.bih0;
  const float* __restrict__ bhh = dir ? A.bhh1 : A.bhh0;
  Gru8xOperands W;
  dof_gru8x_operands(wih, whh, lane, W);
  // transposed operands: row i = 4 bb + sl of tiles A, B, C (see above); K-block kb, element e <-> unit 2 kb + (e & 1),
  // gradient kind e >> 1 (r, z, n, h)
  dof_bf16x8 tA[3];
  {
    const int bb = j >> 2, sl = j & 3;
    auto elem = [&](int tl, int e) DOF_INLINE_LAMBDA -> float {
      const int u = 2 * b + (e & 1), kind = e >> 1;
      // target of this row: dh index (tile A, sl < 2) or dx channel
      int dhi = -1, dxc = -1;
      if (tl == 0) { if (sl < 2) dhi = 2 * bb + sl; else dxc = 8 * bb + sl - 2; }
      else if (tl == 1) dxc = 8 * bb + 2 + sl;
      else if (sl < 2) dxc = 8 * bb + 6 + sl;
      if (dhi >= 0) return kind == 2 ? 0.0f : whh[((kind == 3 ? 2 : kind) * HID + u) * HID + dhi];
      if (dxc >= 0) return kind == 3 ? 0.0f : wih[(kind * HID + u) * IN + dxc];
      return 0.0f;
    };
#pragma unroll
    for (int tl = 0; tl < 3; ++tl) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = elem(tl, e);
      uint32_t lo[3][2], hi[3][2];
      dof_split3x4(reinterpret_cast<const float(&)[4]>(v[0]), lo);
      dof_split3x4(reinterpret_cast<const float(&)[4]>(v[4]), hi);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (tl == 0) tA[p] = dof_mk_bf16x8(lo[p][0], lo[p][1], hi[p][0], hi[p][1]);
        else {
          uint32_t* __restrict__ dst = tbc[3 * (tl - 1) + p][lane];
          dst[0] = lo[p][0]; dst[1] = lo[p][1]; dst[2] = hi[p][0]; dst[3] = hi[p][1];
        }
      }
    }
  }
  dof_f32x4 c1, c2;  // biases in the D layout of this lane's two units
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int unit = 2 * b + m;
    c1[m] = bih[unit] + bhh[unit];
    c1[2 + m] = bih[HID + unit] + bhh[HID + unit];
    c2[m] = bih[2 * HID + unit];
    c2[2 + m] = bhh[2 * HID + unit];
  }
  float* __restrict__ dx_out = dX + (int64_t)dir * T * IN * Sp;
  const int n = in_range ? len[s] : 0;
  float dh[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) dh[m] = (dHfin && n > 0) ? dHfin[(int64_t)(dir * HID + 2 * b + m) * Sp + s] : 0.0f;
  dof_f32x4 acc[6];   // [g_r;g_z] x (x_lo, x_hi, h), [g_n;g_h] x (x_lo, x_hi, h)
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[a] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float sb[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // bias sums of the lane's own (g_r, g_z, g_n, g_h) x 2 units
  // operand addresses: K = 32 holds two piece products of the 16 sequences -- K-block b < 2: first piece, else the second;
  // sequences 8 (b & 1) .. + 7.  A configurations (p0|p0), (p1|p0), (p2|p1); B configurations (p0|p1), (p0|p2).
  if (lane < 8 * 16 / 2) reinterpret_cast<uint32_t*>(&tp[0][72][0])[lane] = 0u;          // zero rows 72..79 of every piece
  if (lane < 8 * 16 / 2) reinterpret_cast<uint32_t*>(&tp[1][72][0])[lane] = 0u;
  if (lane < 8 * 16 / 2) reinterpret_cast<uint32_t*>(&tp[2][72][0])[lane] = 0u;
  const int half = 8 * (b & 1);
  const uint16_t* __restrict__ pa00 = &tp[0][j][half];
  const uint16_t* __restrict__ pa10 = &tp[b < 2 ? 1 : 0][j][half];
  const uint16_t* __restrict__ pa21 = &tp[b < 2 ? 2 : 1][j][half];
  const uint16_t* __restrict__ pb01 = &tp[b < 2 ? 0 : 1][j][half];
  const uint16_t* __restrict__ pb02 = &tp[b < 2 ? 0 : 2][j][half];
  constexpr int PF = 2;
  float nx_x[PF][8], nx_h[PF][2];
  const int64_t sr = in_range ? s : S - 1;  // idle lanes read a valid row
  const DofRowWalk wx(Sp, IN, sr, 8 * b, n), wh(Sp, 2 * HID, sr, dir * HID + 2 * b, n), wdx(Sp, IN, s, 8 * b, n);
  auto issue_loads = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    dof_ld_row<8>(dof_at(X, wx.read(dir, step, T)), nx_x[slot]);
    const float2 hv = *reinterpret_cast<const float2*>(dof_at(O, wh.read_prev(dir, step, T)));
    nx_h[slot][0] = hv.x; nx_h[slot][1] = hv.y;
  };
  auto TBC = [&](int o) DOF_INLINE_LAMBDA { return dof_ld_bf16x8(reinterpret_cast<const uint16_t*>(&tbc[o][lane][0])); };
  auto do_step = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    DOF_MEM_FENCE();  // keeps the LDS operand reads inside the step (hoisted out of the loop they take the registers back)
    const bool act = step < n;
    float xv[8], hp[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = nx_x[slot][e];
#pragma unroll
    for (int m = 0; m < 2; ++m) hp[m] = step > 0 ? nx_h[slot][m] : 0.0f;   // (wave-uniform condition)
    issue_loads(slot_c, step - PF);
    // ---- gates, recomputed exactly as k_gru8x_fwd computes them
    dof_f32x4 a1 = c1, a2 = c2;
    {
      uint32_t lo[3][2], hi[3][2], hw[3];
      dof_split3x4(reinterpret_cast<const float(&)[4]>(xv[0]), lo);
      dof_split3x4(reinterpret_cast<const float(&)[4]>(xv[4]), hi);
      dof_split3x2(hp[0], hp[1], hw);
      const dof_bf16x8 b0 = dof_mk_bf16x8(lo[0][0], lo[0][1], hi[0][0], hi[0][1]);
      const dof_bf16x8 b1 = dof_mk_bf16x8(lo[1][0], lo[1][1], hi[1][0], hi[1][1]);
      const dof_bf16x8 b2 = dof_mk_bf16x8(lo[2][0], lo[2][1], hi[2][0], hi[2][1]);
      a1 = DOF_MFMA_16x16x32_BF16(W.x1[0], b2, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.x2[0], b2, a2);
      a1 = DOF_MFMA_16x16x32_BF16(W.x1[2], b0, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.x2[2], b0, a2);
      a1 = DOF_MFMA_16x16x32_BF16(W.x1[1], b1, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.x2[1], b1, a2);
      a1 = DOF_MFMA_16x16x32_BF16(W.x1[0], b1, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.x2[0], b1, a2);
      a1 = DOF_MFMA_16x16x32_BF16(W.x1[1], b0, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.x2[1], b0, a2);
      a1 = DOF_MFMA_16x16x32_BF16(W.x1[0], b0, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.x2[0], b0, a2);
      const dof_bf16x8 ha = dof_mk_bf16x8(hw[0], hw[1], hw[2], hw[0]);
      const dof_bf16x8 hb = dof_mk_bf16x8(hw[0], hw[1], 0u, 0u);
      a1 = DOF_MFMA_16x16x32_BF16(W.h1b, hb, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.h2b, hb, a2);
      a1 = DOF_MFMA_16x16x32_BF16(W.h1a, ha, a1);
      a2 = DOF_MFMA_16x16x32_BF16(W.h2a, ha, a2);
    }
    float g8[8];   // g_r(2), g_z(2), g_n(2), g_h(2) of the lane's units
    dof_f32x4 d_a0 = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f}, d_a1 = d_a0, d_b = d_a0, d_c = d_a0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float rr = dof_sigmoid(a1[m]);
      const float z = dof_sigmoid(a1[2 + m]);
      const float hn = a2[2 + m];
      const float nn = dof_tanh(fmaf(rr, hn, a2[m]));
      const float dht = act ? dh[m] : 0.0f;   // zero in the idle steps: every product below is then zero
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp[m] - nn);
      const float dnp = dn * (1.0f - nn * nn);
      g8[m] = dnp * hn * rr * (1.0f - rr);
      g8[2 + m] = dz * z * (1.0f - z);
      g8[4 + m] = dnp;
      g8[6 + m] = dnp * rr;
      d_a0[m] = dht * z;
    }
    // ---- [dh ; dx]: transposed operands x the pieces of the lane's own gate gradients
    {
      uint32_t lo[3][2], hi[3][2];
      dof_split3x4(reinterpret_cast<const float(&)[4]>(g8[0]), lo);
      dof_split3x4(reinterpret_cast<const float(&)[4]>(g8[4]), hi);
      const dof_bf16x8 b0 = dof_mk_bf16x8(lo[0][0], lo[0][1], hi[0][0], hi[0][1]);
      const dof_bf16x8 b1 = dof_mk_bf16x8(lo[1][0], lo[1][1], hi[1][0], hi[1][1]);
      const dof_bf16x8 b2 = dof_mk_bf16x8(lo[2][0], lo[2][1], hi[2][0], hi[2][1]);
      // tile A carries dh (the recurrence waits for it): two chains of three, first
      d_a1 = DOF_MFMA_16x16x32_BF16(tA[0], b2, d_a1);
      d_a0 = DOF_MFMA_16x16x32_BF16(tA[0], b1, d_a0);
      d_a1 = DOF_MFMA_16x16x32_BF16(tA[2], b0, d_a1);
      d_a0 = DOF_MFMA_16x16x32_BF16(tA[1], b0, d_a0);
      d_a1 = DOF_MFMA_16x16x32_BF16(tA[1], b1, d_a1);
      d_a0 = DOF_MFMA_16x16x32_BF16(tA[0], b0, d_a0);
      d_b = DOF_MFMA_16x16x32_BF16(TBC(0), b2, d_b);
      d_c = DOF_MFMA_16x16x32_BF16(TBC(3), b2, d_c);
      d_b = DOF_MFMA_16x16x32_BF16(TBC(2), b0, d_b);
      d_c = DOF_MFMA_16x16x32_BF16(TBC(5), b0, d_c);
      d_b = DOF_MFMA_16x16x32_BF16(TBC(1), b1, d_b);
      d_c = DOF_MFMA_16x16x32_BF16(TBC(4), b1, d_c);
      d_b = DOF_MFMA_16x16x32_BF16(TBC(0), b1, d_b);
      d_c = DOF_MFMA_16x16x32_BF16(TBC(3), b1, d_c);
      d_b = DOF_MFMA_16x16x32_BF16(TBC(1), b0, d_b);
      d_c = DOF_MFMA_16x16x32_BF16(TBC(4), b0, d_c);
      d_b = DOF_MFMA_16x16x32_BF16(TBC(0), b0, d_b);
      d_c = DOF_MFMA_16x16x32_BF16(TBC(3), b0, d_c);
    }
    // ---- weight gradients: the step's values go to LDS transposed, piece by piece (the top half of v, of its
    // remainder and of the remainder's remainder ARE the three bf16 pieces: 2-byte stores of register high halves), and
    // come back as K = (two piece products) x (16 sequences) operands: three MFMAs per 16 x 16 tile
    DOF_WAVE_LDS_ORDER();  // this step's writes after the previous step's reads (one wavefront owns the tile)
    {
      auto put3 = [&](int idx, float v) DOF_INLINE_LAMBDA {
        const float r1 = dof_bf16_rest(v), r2 = dof_bf16_rest(r1);
        tp[0][idx][j] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
        tp[1][idx][j] = (uint16_t)(__builtin_bit_cast(uint32_t, r1) >> 16);
        tp[2][idx][j] = (uint16_t)(__builtin_bit_cast(uint32_t, r2) >> 16);
      };
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        put3(8 * (e >> 1) + 2 * b + (e & 1), g8[e]);
        put3(32 + 8 * b + e, xv[e]);
        sb[e] += g8[e];
      }
      put3(64 + 2 * b, hp[0]);
      put3(64 + 2 * b + 1, hp[1]);
    }
    DOF_WAVE_LDS_ORDER();
    {
      constexpr int RT = 16 * 16;   // elements between row tiles
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {   // row tiles [g_r ; g_z], [g_n ; g_h]; smallest products first
        const dof_bf16x8 a21 = dof_ld_bf16x8(pa21 + rt * RT), a10 = dof_ld_bf16x8(pa10 + rt * RT), a00 = dof_ld_bf16x8(pa00 + rt * RT);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {   // column tiles x_0..15, x_16..31, [h ; 0]
          const dof_bf16x8 b01 = dof_ld_bf16x8(pb01 + (2 + ct) * RT), b02 = dof_ld_bf16x8(pb02 + (2 + ct) * RT);
          dof_f32x4 c = acc[3 * rt + ct];
          c = DOF_MFMA_16x16x32_BF16(a21, b01, c);
          c = DOF_MFMA_16x16x32_BF16(a10, b02, c);
          c = DOF_MFMA_16x16x32_BF16(a00, b01, c);
          acc[3 * rt + ct] = c;
        }
      }
    }
    // rows t >= n of dX are zero: an idle lane writes the (zero) row of time `step`
#pragma unroll
    for (int m = 0; m < 2; ++m) dh[m] = act ? d_a0[m] + d_a1[m] : dh[m];
    const float dx8[8] = {d_a0[2] + d_a1[2], d_a0[3] + d_a1[3], d_b[0], d_b[1], d_b[2], d_b[3], d_c[0], d_c[1]};
    dof_st_row<8>(dof_at(dx_out, wdx.write(dir, step, act)), dx8);
    DOF_SCHED_FENCE();
  };
  // slot of step st = (T - 1 - st) % PF
  dof_static_for<PF>([&](auto d) { issue_loads(d, T - 1 - decltype(d)::value); });
  int step = T - 1;
  for (; step - (PF - 1) >= 0; step -= PF) {   // wave-uniform trip count (MFMA ignores EXEC); no branch inside
    dof_static_for<PF>([&](auto d) { do_step(d, step - decltype(d)::value); });
  }
  dof_static_for<PF - 1>([&](auto d) {   // the last T % PF steps
    if (step - decltype(d)::value >= 0) do_step(d, step - decltype(d)::value);
  });
  // ---- partials of this wavefront's 16 sequences, in the reference's tensor order
  float* __restrict__ out = wg_partial + ((int64_t)dir * nblk_own + blockIdx.x) * GRU8X_WG_FLOATS;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * b + r;   // row of a row tile: [g_r ; g_z] or [g_n ; g_h]
    // weight_ih rows: r gate 0..7, z gate 8..15, n gate 16..23; columns j (x_lo) and 16 + j (x_hi)
    out[row * IN + j] = acc[0][r];
    out[row * IN + 16 + j] = acc[1][r];
    if (row < 8) {
      out[(16 + row) * IN + j] = acc[3][r];
      out[(16 + row) * IN + 16 + j] = acc[4][r];
    }
    if (j < 8) {
      out[24 * IN + row * HID + j] = acc[2][r];                       // weight_hh rows of gates r, z
      if (row >= 8) out[24 * IN + (16 + row - 8) * HID + j] = acc[5][r];   // g_h x h: the n gate's rows
    }
  }
  // bias sums [r, z, n, h][unit]: over the 16 sequences of the lane group (lanes of a DPP row share b)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = dof_row16_sum(sb[e]);
    if (j == 0) out[24 * IN + 24 * HID + 8 * (e >> 1) + 2 * b + (e & 1)] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// GRU(16 -> 16) backward on the bf16 matrix pipe with exact three-piece operands (round 5): the twin of k_gru16x_fwd.
// A workgroup = four wavefronts, each an independent tile of 16 (sequence, direction) pairs in k_gru16x_fwd's lane
// mapping; the 24 MFMA weight operands (4 words per lane: 96 registers if held) live in LDS and are shared by the four.
// Per step and wavefront:
//   * recompute: k_gru16x_fwd's 18 MFMAs in its order (bitwise the same r, z, n) from x_t and h_{t-1} -- issued for
//     step t - 1 in the same scheduling region as the dh chain of step t (nothing in it depends on the recurrence);
//   * gate gradients on the VALU (inactive steps need no selects: with dO zeroed for them every gate gradient is
//     exactly zero, dh stays zero and the dX row written is the zero row);
//   * [dh | dx] parts of gates r, z: K-block b = the lane's own [g_r(4) ; g_z(4)], A = the same K order of the transposed
//     [W_r | W_z] rows, six piece products each; the n gate's halves (W_hn^T g_h for dh, W_in^T g_n for dx) pack two piece
//     products into K = 32 like the forward kernel's n halves: 18 MFMAs, each of dh and dx in three independent
//     accumulators (chains of 3) summed on the VALU;
//   * weight gradients: contraction over the 16 sequences -- the step's 96 values per sequence go through a 9 KB LDS
//     image transposed and cut into pieces ([piece][value][sequence] bf16, 2-byte stores of register high halves) and
//     return as K = 32 = (two piece products) x (16 sequences) operands: 18 MFMAs into six accumulator tiles.
// 54 v_mfma_f32_16x16x32_bf16 per step against round 4's 72 v_mfma_f32_16x16x4_f32 at 2.5 x the time each.  Measured on
// MI355X (tools/probe/gru16_probe.hip, both streams of C2): 215 us (round 4) -> 173 us.  The time loop has no branch
// (stores unconditional, see k_gru16x_fwd); HAS_DO: the layer receives dO (compile time).
// ---------------------------------------------------------------------------------------------
template <bool HAS_DO>
__global__ void __launch_bounds__(256, 2) k_gru16x_bwd(Gru16mStream st_a, Gru16mStream st_b, int T) {
  constexpr int HID = 16, IN = 16;
  // The 24 MFMA weight operands (4 words per lane) live in LDS, shared by the four wavefronts of the workgroup (each an
  // independent tile of 16 sequences): in registers they were 96 of the kernel's ~330 and held it to one wavefront per
  // SIMD.  [operand][lane]: a wave-wide read is 1 KB contiguous.
  __shared__ __attribute__((aligned(16))) uint32_t wop[24][64][4];
  // per wavefront: the step's values TRANSPOSED for the weight gradients, [piece][g_r g_z g_n g_h x h_prev (16 each)][sequence]
  // as bf16 -- a lane's MFMA operand (eight consecutive sequences of one row) is 16 contiguous bytes
  __shared__ __attribute__((aligned(16))) uint16_t tps[4][3][96][16];
  __shared__ __attribute__((aligned(16))) float cbias[4][64][4];   // c_r, c_z, c_n, c_hn of every lane (accumulator seeds)
  const Gru16mStream& A = blockIdx.z ? st_b : st_a;
  const float* __restrict__ X = A.X;
  const int* __restrict__ len = A.len;
  const float* __restrict__ O = A.O;
  const float* __restrict__ dO = A.dO;
  float* __restrict__ dX = A.dX;
  float* __restrict__ wg_partial = A.wg_partial;
  const int64_t S = A.S, Sp = A.Sp;
  if ((int64_t)blockIdx.x * 64 >= S) return;   // (the grid covers the longer stream; whole workgroups leave)
  const unsigned nblk_own = (unsigned)((S + 15) / 16);   // partial rows of THIS stream: one per wavefront tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, b = lane >> 4;
  const int64_t tile_idx = (int64_t)blockIdx.x * 4 + wave;
  const int64_t s = tile_idx * 16 + j;
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const float* __restrict__ wih = dir ? A.wih1 : A.wih0;
  const float* __restrict__ whh = dir ? A.whh1 : A.whh0;
  const float* __restrict__ bih = dir ? A.bih1 : A.bih0;
  const float* __restrict__ bhh = dir ? A.bhh1 : A.bhh0;
  // operands 0-5: forward [W_ih | W_hh] of gates r, z, three pieces each; 6-8 / 9-11: the n gate's input / hidden halves
  // (W0|W0), (W1|W0), (W2|W1); 12-14 / 15-17: transposed [W_r | W_z] rows for dh / dx (row = hidden / input index j,
  // K-block b = gate units 4b .. 4b+3); 18-20 / 21-23: W_hn^T / W_in^T.  Wavefront w prepares operands 6w .. 6w+5.
  {
    auto put = [&](int o, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) DOF_INLINE_LAMBDA {
      wop[o][lane][0] = w0; wop[o][lane][1] = w1; wop[o][lane][2] = w2; wop[o][lane][3] = w3;
    };
    const bool transposed = wave >= 2;
    auto rows = [&](const float* __restrict__ w, int g, uint32_t (&out)[3][2]) DOF_INLINE_LAMBDA {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = transposed ? w[(g * HID + 4 * b + q) * 16 + j] : w[(g * HID + j) * 16 + 4 * b + q];
      dof_split3x4(v, out);
    };
    if ((wave & 1) == 0) {   // gates r, z: (input | hidden) per piece (forward), (r | z) per piece (transposed)
      uint32_t i0[3][2], i1[3][2], h0[3][2], h1[3][2];
      rows(wih, 0, i0); rows(wih, 1, i1); rows(whh, 0, h0); rows(whh, 1, h1);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (!transposed) {
          put(p, i0[p][0], i0[p][1], h0[p][0], h0[p][1]);
          put(3 + p, i1[p][0], i1[p][1], h1[p][0], h1[p][1]);
        } else {
          put(12 + p, h0[p][0], h0[p][1], h1[p][0], h1[p][1]);
          put(15 + p, i0[p][0], i0[p][1], i1[p][0], i1[p][1]);
        }
      }
    } else {                 // gate n: two piece products per operand
      uint32_t i2[3][2], h2[3][2];
      rows(wih, 2, i2); rows(whh, 2, h2);
      const int oi = transposed ? 21 : 6, oh = transposed ? 18 : 9;
      put(oi + 0, i2[0][0], i2[0][1], i2[0][0], i2[0][1]);
      put(oi + 1, i2[1][0], i2[1][1], i2[0][0], i2[0][1]);
      put(oi + 2, i2[2][0], i2[2][1], i2[1][0], i2[1][1]);
      put(oh + 0, h2[0][0], h2[0][1], h2[0][0], h2[0][1]);
      put(oh + 1, h2[1][0], h2[1][1], h2[0][0], h2[0][1]);
      put(oh + 2, h2[2][0], h2[2][1], h2[1][0], h2[1][1]);
    }
  }
  if (wave == 0) {   // biases in the D layout (register r <-> unit 4b + r): seeds of the recompute accumulators
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = 4 * b + r;
      cbias[0][lane][r] = bih[unit] + bhh[unit];
      cbias[1][lane][r] = bih[HID + unit] + bhh[HID + unit];
      cbias[2][lane][r] = bih[2 * HID + unit];
      cbias[3][lane][r] = bhh[2 * HID + unit];
    }
  }
  __syncthreads();
  if (tile_idx * 16 >= S) return;   // (a wavefront without sequences; no workgroup barrier below)
  uint16_t (*tp)[96][16] = tps[wave];
  auto W = [&](int o) DOF_INLINE_LAMBDA { return dof_ld_bf16x8(reinterpret_cast<const uint16_t*>(&wop[o][lane][0])); };
  float* __restrict__ dx_out = dX + (int64_t)dir * T * IN * Sp;
  const int n = in_range ? len[s] : 0;
  float dh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  dof_f32x4 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[a] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float sb[4][4];  // bias sums [r, z, n, h][unit 4b + r] of the lane's own sequence
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) sb[g][r] = 0.0f;
  // operand addresses: K = 32 holds two piece products of the 16 sequences -- K-block b < 2: first piece, else the second;
  // sequences 8 (b & 1) .. + 7.  A configurations (p0|p0), (p1|p0), (p2|p1); B configurations (p0|p1), (p0|p2).
  const int half = 8 * (b & 1);
  const uint16_t* __restrict__ pa00 = &tp[0][j][half];
  const uint16_t* __restrict__ pa10 = &tp[b < 2 ? 1 : 0][j][half];
  const uint16_t* __restrict__ pa21 = &tp[b < 2 ? 2 : 1][j][half];
  const uint16_t* __restrict__ pb01 = &tp[b < 2 ? 0 : 1][j][half];
  const uint16_t* __restrict__ pb02 = &tp[b < 2 ? 0 : 2][j][half];
  constexpr int PF = 2;   // x_t, h_{t-1}, dO_t loaded PF steps ahead into static register slots (see k_gru16x_fwd)
  float nx_x[PF][4], nx_h[PF][4], nx_d[PF][4];
  const int64_t sr = in_range ? s : S - 1;  // idle lanes read a valid row
  const DofRowWalk wx(Sp, IN, sr, 4 * b, n), wh(Sp, 2 * HID, sr, dir * HID + 4 * b, n), wdx(Sp, IN, s, 4 * b, n);
  auto issue_loads = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    dof_ld_row<4>(dof_at(X, wx.read(dir, step, T)), nx_x[slot]);
    dof_ld_row<4>(dof_at(O, wh.read_prev(dir, step, T)), nx_h[slot]);
    if constexpr (HAS_DO) dof_ld_row<4>(dof_at(dO, wh.read(dir, step, T)), nx_d[slot]);
    else nx_d[slot][0] = nx_d[slot][1] = nx_d[slot][2] = nx_d[slot][3] = 0.0f;
  };
  // activations of the step the chain works on next: r, z, n, W_hn h + b_hn, and h_{t-1} with the zero of step 0
  float act_r[4], act_z[4], act_n[4], act_hn[4], act_hp[4];
  auto recompute = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {   // from slot (step's x, h_{t-1}); k_gru16x_fwd's MFMAs in its order
    constexpr int slot = decltype(slot_c)::value;
    float hp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hp[q] = step > 0 ? nx_h[slot][q] : 0.0f;   // (wave-uniform condition)
    uint32_t xw[3][2], hw[3][2];
    dof_split3x4(nx_x[slot], xw);
    dof_split3x4(hp, hw);
    auto seed = [&](int g) DOF_INLINE_LAMBDA {
      float v[4];
      dof_ld_row<4>(&cbias[g][lane][0], v);
      return dof_f32x4{v[0], v[1], v[2], v[3]};
    };
    dof_f32x4 a_r = seed(0), a_z = seed(1), a_n = seed(2), a_hn_ = seed(3);
    const dof_bf16x8 x01 = dof_mk_bf16x8(xw[0][0], xw[0][1], xw[1][0], xw[1][1]);
    const dof_bf16x8 x02 = dof_mk_bf16x8(xw[0][0], xw[0][1], xw[2][0], xw[2][1]);
    a_n = DOF_MFMA_16x16x32_BF16(W(8), x01, a_n);
    a_n = DOF_MFMA_16x16x32_BF16(W(7), x02, a_n);
    a_n = DOF_MFMA_16x16x32_BF16(W(6), x01, a_n);
    const dof_bf16x8 b0 = dof_mk_bf16x8(xw[0][0], xw[0][1], hw[0][0], hw[0][1]);
    const dof_bf16x8 b1 = dof_mk_bf16x8(xw[1][0], xw[1][1], hw[1][0], hw[1][1]);
    const dof_bf16x8 b2 = dof_mk_bf16x8(xw[2][0], xw[2][1], hw[2][0], hw[2][1]);
    const dof_bf16x8 h01 = dof_mk_bf16x8(hw[0][0], hw[0][1], hw[1][0], hw[1][1]);
    const dof_bf16x8 h02 = dof_mk_bf16x8(hw[0][0], hw[0][1], hw[2][0], hw[2][1]);
    a_r = DOF_MFMA_16x16x32_BF16(W(0), b2, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(W(3), b2, a_z);
    a_hn_ = DOF_MFMA_16x16x32_BF16(W(11), h01, a_hn_);
    a_r = DOF_MFMA_16x16x32_BF16(W(2), b0, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(W(5), b0, a_z);
    a_hn_ = DOF_MFMA_16x16x32_BF16(W(10), h02, a_hn_);
    a_r = DOF_MFMA_16x16x32_BF16(W(1), b1, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(W(4), b1, a_z);
    a_hn_ = DOF_MFMA_16x16x32_BF16(W(9), h01, a_hn_);
    a_r = DOF_MFMA_16x16x32_BF16(W(0), b1, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(W(3), b1, a_z);
    a_r = DOF_MFMA_16x16x32_BF16(W(1), b0, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(W(4), b0, a_z);
    a_r = DOF_MFMA_16x16x32_BF16(W(0), b0, a_r);
    a_z = DOF_MFMA_16x16x32_BF16(W(3), b0, a_z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      act_r[r] = dof_sigmoid(a_r[r]);
      act_z[r] = dof_sigmoid(a_z[r]);
      act_hn[r] = a_hn_[r];
      act_n[r] = dof_tanh(fmaf(act_r[r], a_hn_[r], a_n[r]));
      act_hp[r] = hp[r];
    }
  };
  auto chain = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {   // the dh recurrence of `step` (its activations are in act_*)
    constexpr int slot = decltype(slot_c)::value;
    const bool act = step < n;
    float g_r[4], g_z[4], g_n[4], g_h[4], xv[4], hp[4];
    dof_f32x4 dh_a, dh_b = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f}, dh_c = dh_b, dx_a = dh_b, dx_b = dh_b, dx_c = dh_b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rr = act_r[r], z = act_z[r], nn = act_n[r], hn = act_hn[r];
      hp[r] = act_hp[r];
      xv[r] = nx_x[slot][r];
      const float dht = dh[r] + (act ? nx_d[slot][r] : 0.0f);   // zero in the idle steps: every product below is then zero
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp[r] - nn);
      const float dnp = dn * (1.0f - nn * nn);
      g_r[r] = dnp * hn * rr * (1.0f - rr);
      g_z[r] = dz * z * (1.0f - z);
      g_n[r] = dnp;
      g_h[r] = dnp * rr;
      dh_a[r] = dht * z;
    }
    {
      uint32_t rw[3][2], zw[3][2], nw[3][2], gw[3][2];
      dof_split3x4(g_r, rw);
      dof_split3x4(g_z, zw);
      dof_split3x4(g_h, gw);
      const dof_bf16x8 b0 = dof_mk_bf16x8(rw[0][0], rw[0][1], zw[0][0], zw[0][1]);
      const dof_bf16x8 b1 = dof_mk_bf16x8(rw[1][0], rw[1][1], zw[1][0], zw[1][1]);
      const dof_bf16x8 b2 = dof_mk_bf16x8(rw[2][0], rw[2][1], zw[2][0], zw[2][1]);
      const dof_bf16x8 h01 = dof_mk_bf16x8(gw[0][0], gw[0][1], gw[1][0], gw[1][1]);
      const dof_bf16x8 h02 = dof_mk_bf16x8(gw[0][0], gw[0][1], gw[2][0], gw[2][1]);
      // the recurrence waits for dh: its three chains first
      dh_b = DOF_MFMA_16x16x32_BF16(W(12), b2, dh_b);
      dh_c = DOF_MFMA_16x16x32_BF16(W(20), h01, dh_c);
      dh_a = DOF_MFMA_16x16x32_BF16(W(12), b1, dh_a);
      dh_b = DOF_MFMA_16x16x32_BF16(W(14), b0, dh_b);
      dh_c = DOF_MFMA_16x16x32_BF16(W(19), h02, dh_c);
      dh_a = DOF_MFMA_16x16x32_BF16(W(13), b0, dh_a);
      dh_b = DOF_MFMA_16x16x32_BF16(W(13), b1, dh_b);
      dh_c = DOF_MFMA_16x16x32_BF16(W(18), h01, dh_c);
      dh_a = DOF_MFMA_16x16x32_BF16(W(12), b0, dh_a);
      dof_split3x4(g_n, nw);
      const dof_bf16x8 n01 = dof_mk_bf16x8(nw[0][0], nw[0][1], nw[1][0], nw[1][1]);
      const dof_bf16x8 n02 = dof_mk_bf16x8(nw[0][0], nw[0][1], nw[2][0], nw[2][1]);
      dx_b = DOF_MFMA_16x16x32_BF16(W(15), b2, dx_b);
      dx_c = DOF_MFMA_16x16x32_BF16(W(23), n01, dx_c);
      dx_a = DOF_MFMA_16x16x32_BF16(W(15), b1, dx_a);
      dx_b = DOF_MFMA_16x16x32_BF16(W(17), b0, dx_b);
      dx_c = DOF_MFMA_16x16x32_BF16(W(22), n02, dx_c);
      dx_a = DOF_MFMA_16x16x32_BF16(W(16), b0, dx_a);
      dx_b = DOF_MFMA_16x16x32_BF16(W(16), b1, dx_b);
      dx_c = DOF_MFMA_16x16x32_BF16(W(21), n01, dx_c);
      dx_a = DOF_MFMA_16x16x32_BF16(W(15), b0, dx_a);
    }
    // ---- weight gradients (see k_gru8x_bwd): the step's values go to LDS transposed, piece by piece, and come back as
    // K = (two piece products) x (16 sequences) operands: three MFMAs per 16 x 16 tile, 18 per step
    DOF_WAVE_LDS_ORDER();  // this step's writes after the previous step's reads (one wavefront owns the image)
    {
      auto put3 = [&](int idx, float v) DOF_INLINE_LAMBDA {
        const float r1 = dof_bf16_rest(v), r2 = dof_bf16_rest(r1);
        tp[0][idx][j] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
        tp[1][idx][j] = (uint16_t)(__builtin_bit_cast(uint32_t, r1) >> 16);
        tp[2][idx][j] = (uint16_t)(__builtin_bit_cast(uint32_t, r2) >> 16);
      };
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        put3(4 * b + r, g_r[r]); put3(16 + 4 * b + r, g_z[r]); put3(32 + 4 * b + r, g_n[r]); put3(48 + 4 * b + r, g_h[r]);
        put3(64 + 4 * b + r, xv[r]); put3(80 + 4 * b + r, hp[r]);
        sb[0][r] += g_r[r]; sb[1][r] += g_z[r]; sb[2][r] += g_n[r]; sb[3][r] += g_h[r];
      }
    }
    DOF_WAVE_LDS_ORDER();
    {
      constexpr int RT = 16 * 16;   // elements between row tiles
      // acc: [g_r, g_z, g_n] x x, then [g_r, g_z, g_h] x h_prev
      dof_static_for<4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;   // row tile: g_r, g_z, g_n, g_h
        const dof_bf16x8 a21 = dof_ld_bf16x8(pa21 + g * RT), a10 = dof_ld_bf16x8(pa10 + g * RT), a00 = dof_ld_bf16x8(pa00 + g * RT);
        if constexpr (g != 3) {   // x columns
          const dof_bf16x8 b01 = dof_ld_bf16x8(pb01 + 4 * RT), b02 = dof_ld_bf16x8(pb02 + 4 * RT);
          dof_f32x4 c = acc[g];
          c = DOF_MFMA_16x16x32_BF16(a21, b01, c);
          c = DOF_MFMA_16x16x32_BF16(a10, b02, c);
          c = DOF_MFMA_16x16x32_BF16(a00, b01, c);
          acc[g] = c;
        }
        if constexpr (g != 2) {   // h_prev columns
          constexpr int ah = g == 3 ? 5 : 3 + g;
          const dof_bf16x8 b01 = dof_ld_bf16x8(pb01 + 5 * RT), b02 = dof_ld_bf16x8(pb02 + 5 * RT);
          dof_f32x4 c = acc[ah];
          c = DOF_MFMA_16x16x32_BF16(a21, b01, c);
          c = DOF_MFMA_16x16x32_BF16(a10, b02, c);
          c = DOF_MFMA_16x16x32_BF16(a00, b01, c);
          acc[ah] = c;
        }
      });
    }
    // rows t >= n of dX are zero: an idle lane writes the (zero) row of time `step`
    float dx4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dh[r] = dh_a[r] + (dh_b[r] + dh_c[r]);
      dx4[r] = dx_a[r] + (dx_b[r] + dx_c[r]);
    }
    dof_st_row<4>(dof_at(dx_out, wdx.write(dir, step, act)), dx4);
  };
  // slot of step st = (T - 1 - st) % PF
  dof_static_for<PF>([&](auto d) { issue_loads(d, T - 1 - decltype(d)::value); });
  recompute(std::integral_constant<int, 0>{}, T - 1);
  auto do_step = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int next = (slot + 1) % PF;
    chain(slot_c, step);
    recompute(std::integral_constant<int, next>{}, step - 1);   // (step 0: a discarded recompute of valid rows)
    issue_loads(slot_c, step - PF);
    DOF_SCHED_FENCE();
  };
  int step = T - 1;
  for (; step - (PF - 1) >= 0; step -= PF) {   // wave-uniform trip count (MFMA ignores EXEC); no branch inside
    dof_static_for<PF>([&](auto d) { do_step(d, step - decltype(d)::value); });
  }
  dof_static_for<PF - 1>([&](auto d) {   // the last T % PF steps
    if (step - decltype(d)::value >= 0) do_step(d, step - decltype(d)::value);
  });
  // ---- partials of this wavefront's 16 sequences: tiles [a][row = unit][col], then the bias sums [gate][unit]
  float* __restrict__ out = wg_partial + ((int64_t)dir * nblk_own + tile_idx) * GRU16_WG_FLOATS;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[a * 256 + (4 * b + r) * 16 + j] = acc[a][r];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = dof_row16_sum(sb[g][r]);  // over the 16 sequences (lanes of a DPP row share b)
      if (j == 0) out[6 * 256 + g * 16 + 4 * b + r] = v;
    }
}

