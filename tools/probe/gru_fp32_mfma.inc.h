// Round 4's fp32-MFMA GRU recurrences (v_mfma_f32_16x16x4_f32), kept for the development probe only
// (tools/probe/gru16_probe.hip times them beside the bf16 three-piece kernels that replaced them in round 5).
// Not part of the product library.  Include after deepof_amd/csrc/k_grum16.inc.h (Gru16mStream).
#pragma once
// ---------------------------------------------------------------------------------------------
// GRU forward on the matrix pipe (IN = HID = 16; round 3).  A wavefront owns 16 (sequence, direction) pairs: lane
// l = (b = l >> 4, j = l & 15) holds units 4b .. 4b+3 of sequence j -- the C/D layout of v_mfma_f32_16x16x4_f32
// (D[4b + r][j] in register r).  One time step is G[unit][seq] = W[unit][k] V[k][seq] with V = [x_t ; h_{t-1}]:
//   * A operand = a 16 x 4 weight tile, resident in one VGPR per (gate, K-block): 24 registers hold W_ih and W_hh
//     (the lane-per-unit form above keeps 96 weights per lane and pays one DPP broadcast per FMA);
//   * B operand = four values of the lane's OWN sequence: K-block q is defined as the indices {q, 4+q, 8+q, 12+q}, so
//     lane (b, j) contributes V[4b + q][j] -- for x the q-th float of the 16-byte piece it loads, for h the q-th of the
//     four units it has just computed.  The recurrence never moves data between lanes.
// 24 MFMAs per step and wavefront (768 matrix-pipe cycles for 16 pairs = 48 per pair and step; the VALU form measures
// 164) beside ~90 VALU instructions of gate arithmetic on four units per lane.  v_mfma_f32_*_f32 multiplies exact
// fp32 and accumulates like an fmaf chain (MI355X_MICROARCH.md), so the gates differ from the lane-per-unit kernel only
// by the summation order over k.  Same saved-gate / output layouts as k_gru3_fwd<16,16>: 16-byte output stores, the
// four units' (r, z, n, W_hn h + b_hn) words are 64 contiguous bytes.
// ---------------------------------------------------------------------------------------------
// One launch serves up to two independent layers of this shape (blockIdx.z: the node and the edge stream of the encoder):
// a stream of 14,336 sequences is only 1.75 wavefronts per SIMD, and a wavefront's MFMA and VALU phases do not overlap,
// so two streams side by side finish in little more than the time of one (measured at 4 x the sequences: 3.4 x the time).

// NT = (sequence, direction) tiles of 16 per wavefront (they share the 24 weight registers; their recurrences are
// independent, so one tile's MFMAs can issue while the other's gate arithmetic runs), WPE = wavefronts per SIMD the
// register allocation is held to.  Round 5: the round-4 form (NT = 1, no bound) compiled to 132 registers = 3 wavefronts
// per SIMD = 3,072 resident wavefronts for the 3,584 of a C2 launch: a second round of 512 wavefronts on an idle chip
// doubled the kernel's time.
template <int NT, int WPE>
__global__ void __launch_bounds__(64, WPE) k_gru16m_fwd(Gru16mStream sa, Gru16mStream sb, int T) {
  constexpr int HID = 16, IN = 16;
  const Gru16mStream& A = blockIdx.z ? sb : sa;
  const float* __restrict__ X = A.X;
  const int* __restrict__ len = A.len;
  float* __restrict__ O = A.O;
  float* __restrict__ GS = A.GS;
  const int64_t S = A.S, Sp = A.Sp;
  if ((int64_t)blockIdx.x * (16 * NT) >= S) return;   // (the grid covers the longer stream)
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, b = lane >> 4;
  const int dir = blockIdx.y;
  const float* __restrict__ wih = dir ? A.wih1 : A.wih0;
  const float* __restrict__ whh = dir ? A.whh1 : A.whh0;
  const float* __restrict__ bih = dir ? A.bih1 : A.bih0;
  const float* __restrict__ bhh = dir ? A.bhh1 : A.bhh0;
  // A tiles: row i = lane & 15 (a unit), k index = lane >> 4; K-block q covers input / hidden indices 4 (lane >> 4) + q
  float aix[3][4], ahh[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      aix[g][q] = wih[(g * HID + j) * IN + 4 * b + q];
      ahh[g][q] = whh[(g * HID + j) * HID + 4 * b + q];
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
  int64_t s[NT], sr[NT];
  int n[NT];
  bool in_range[NT];
  float h[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    s[i] = ((int64_t)blockIdx.x * NT + i) * 16 + j;
    in_range[i] = s[i] < S;
    n[i] = in_range[i] ? len[s[i]] : 0;
    sr[i] = in_range[i] ? s[i] : S - 1;  // lanes past the end (and finished sequences) read a valid row: loads stay
                                         // unconditional (a branch around a load costs a vmcnt(0) at the join)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[i][r] = 0.0f;
  }
  // x_t is loaded PF steps ahead into static register slots (the loop is unrolled by PF): a rotating copy made the
  // compiler wait `vmcnt(0)` at every loop back-edge -- one exposed HBM round trip (~1.8 us) per time step
  constexpr int PF = 4;
  float xs[NT][PF][4];
  auto load_x = [&](auto tile_c, auto slot_c, int step) {
    constexpr int i = decltype(tile_c)::value;
    constexpr int slot = decltype(slot_c)::value;
    const int t = step < n[i] ? (dir ? (n[i] - 1 - step) : step) : 0;
    dof_ld_row<4>(X + ACT(t, 4 * b, IN, Sp, sr[i]), xs[i][slot]);
  };
  // input halves of the gates: they do not depend on the recurrence and are issued one step ahead, so the matrix pipe
  // has independent work while the gate arithmetic of the current step runs on the VALU
  dof_f32x4 g_r[NT], g_z[NT], g_n[NT];
  auto input_half = [&](auto tile_c, const float* xq) {
    constexpr int i = decltype(tile_c)::value;
    g_r[i] = c_r; g_z[i] = c_z; g_n[i] = c_n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g_r[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[0][q], xq[q], g_r[i], 0, 0, 0);
      g_z[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[1][q], xq[q], g_z[i], 0, 0, 0);
      g_n[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[2][q], xq[q], g_n[i], 0, 0, 0);
    }
  };
  auto do_step = [&](auto slot_c, int step) {   // slot = step % PF holds x of this step
    constexpr int slot = decltype(slot_c)::value;
    constexpr int next = (slot + 1) % PF;
    dof_f32x4 a_r[NT], a_z[NT], a_hn[NT], a_n[NT];
    dof_static_for<NT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      a_r[i] = g_r[i]; a_z[i] = g_z[i]; a_hn[i] = c_hn; a_n[i] = g_n[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a_r[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][q], h[i][q], a_r[i], 0, 0, 0);
        a_z[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][q], h[i][q], a_z[i], 0, 0, 0);
        a_hn[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[2][q], h[i][q], a_hn[i], 0, 0, 0);
      }
    });
    dof_static_for<NT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      input_half(ic, xs[i][next]);   // next step's input half (its x arrived PF - 1 steps ago)
      load_x(ic, slot_c, step + PF); // this step's slot is free again
    });
    dof_static_for<NT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const bool act = step < n[i];
      float gate16[16];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rr = dof_sigmoid(a_r[i][r]);
        const float zz = dof_sigmoid(a_z[i][r]);
        const float nn = dof_tanh(fmaf(rr, a_hn[i][r], a_n[i][r]));
        const float hnew = fmaf(zz, h[i][r] - nn, nn);
        h[i][r] = act ? hnew : h[i][r];
        gate16[4 * r] = rr; gate16[4 * r + 1] = zz; gate16[4 * r + 2] = nn; gate16[4 * r + 3] = a_hn[i][r];
      }
      if (act) {
        const int t = dir ? (n[i] - 1 - step) : step;
        dof_st_row<4>(O + ACT(t, dir * HID + 4 * b, 2 * HID, Sp, s[i]), h[i]);
        if (gs) dof_st_row<16>(gs + ACT(t, 16 * b, 4 * HID, Sp, s[i]), gate16);
      }
    });
  };
  dof_static_for<NT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    dof_static_for<PF>([&](auto d) { load_x(ic, d, decltype(d)::value); });
  });
  dof_static_for<NT>([&](auto ic) { input_half(ic, xs[decltype(ic)::value][0]); });
  for (int step = 0; step < T; step += PF) {  // wave-uniform trip count: MFMA ignores EXEC, finished sequences idle
    // (no exit inside the unrolled group: steps >= T are idle steps -- n <= T -- and cost nothing to keep; with exits
    //  the compiler drained the memory counter at the loop header)
    dof_static_for<PF>([&](auto d) { do_step(d, step + decltype(d)::value); });
  }
  const float zero16[16] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NT; ++i)
    if (in_range[i])
      for (int t = n[i]; t < T; ++t) {
        dof_st_row<4>(O + ACT(t, dir * HID + 4 * b, 2 * HID, Sp, s[i]), zero16);
        if (gs) dof_st_row<16>(gs + ACT(t, 16 * b, 4 * HID, Sp, s[i]), zero16);
      }
}


// ---------------------------------------------------------------------------------------------
// GRU forward on the matrix pipe for the second encoder layer of latent 8 (IN = 32, HID = 8; round 4), the twin of
// k_gru16m_fwd.  Eight units per direction fill half a 16-row tile, so two GATES share a tile and the rows are ordered by
// OWNER: lane (b, j) of the D layout (rows 4b .. 4b+3, column j) owns units 2b and 2b+1 of sequence j and receives
//   tile 1 rows 4b + (0, 1, 2, 3) = r_{2b}, r_{2b+1}, z_{2b}, z_{2b+1},   tile 2 = nx_{2b}, nx_{2b+1}, hn_{2b}, hn_{2b+1}
// (nx = W_in x + b_in, hn = W_hn h + b_hn: the A tile holds zeros where a row does not take that half of [x ; h]) -- all
// four pre-activations of a unit in ONE lane, no exchange for the gate arithmetic.  K blocks are again the lane's own
// values: block q of the input half = channels {q, 8+q, 16+q, 24+q} (lane b loads x[8b .. 8b+7]: two 16-byte loads),
// block q of the hidden half = units {q, 2+q, 4+q, 6+q} (the two units the lane has just computed).  16 + 4 MFMAs per
// step for 16 (sequence, direction) pairs; the input half runs one step ahead.  Writes the same hidden-state and
// unit-major gate buffers as k_gru3_fwd<32, 8> (a lane's two units: 32 contiguous bytes), so k_gru8_bwd_fused is
// unchanged; sums over k in another order than the lane-per-unit kernel (ulp-level differences).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_gru8m_fwd(Gru16mStream sa, Gru16mStream sb, int T) {
  constexpr int HID = 8, IN = 32;
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
  // A tiles: row i = lane & 15 -> owner (i >> 2), slot (i & 3): gate half (slot >> 1), unit 2 (i >> 2) + (slot & 1);
  // k index = lane >> 4: input channel 8k + q / hidden unit 2k + q of K block q
  const int arow_unit = 2 * (j >> 2) + (j & 1), arow_hi = (j >> 1) & 1;   // hi: z (tile 1) / hn (tile 2)
  float a1x[8], a2x[8], a1h[2], a2h[2];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a1x[q] = wih[((arow_hi ? 1 : 0) * HID + arow_unit) * IN + 8 * b + q];
    a2x[q] = arow_hi ? 0.0f : wih[(2 * HID + arow_unit) * IN + 8 * b + q];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    a1h[q] = whh[((arow_hi ? 1 : 0) * HID + arow_unit) * HID + 2 * b + q];
    a2h[q] = arow_hi ? whh[(2 * HID + arow_unit) * HID + 2 * b + q] : 0.0f;
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
  float* __restrict__ gs = GS ? GS + (int64_t)dir * T * 4 * HID * Sp : nullptr;
  const int n = in_range ? len[s] : 0;
  float h[2] = {0.0f, 0.0f};
  constexpr int PF = 4;   // x_t loaded PF steps ahead into static register slots (see k_gru16m_fwd)
  float xs[PF][8];
  const int64_t sr = in_range ? s : S - 1;
  auto load_x = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    const int t = step < n ? (dir ? (n - 1 - step) : step) : 0;
    dof_ld_row<8>(X + ACT(t, 8 * b, IN, Sp, sr), xs[slot]);
  };
  dof_f32x4 g1, g2;
  auto input_half = [&](const float* xq) {
    g1 = c1; g2 = c2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1x[q], xq[q], g1, 0, 0, 0);
      g2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2x[q], xq[q], g2, 0, 0, 0);
    }
  };
  auto do_step = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int next = (slot + 1) % PF;
    dof_f32x4 a1 = g1, a2 = g2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1h[q], h[q], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2h[q], h[q], a2, 0, 0, 0);
    }
    input_half(xs[next]);          // next step's input half (its x arrived PF - 1 steps ago)
    load_x(slot_c, step + PF);     // this step's slot is free again
    const bool act = step < n;
    float gate8[8];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float rr = dof_sigmoid(a1[m]);
      const float zz = dof_sigmoid(a1[2 + m]);
      const float hn = a2[2 + m];
      const float nn = dof_tanh(fmaf(rr, hn, a2[m]));
      const float hnew = fmaf(zz, h[m] - nn, nn);
      h[m] = act ? hnew : h[m];
      gate8[4 * m] = rr; gate8[4 * m + 1] = zz; gate8[4 * m + 2] = nn; gate8[4 * m + 3] = hn;
    }
    if (act) {
      const int t = dir ? (n - 1 - step) : step;
      dof_st_pair(O + ACT(t, dir * HID + 2 * b, 2 * HID, Sp, s), h[0], h[1]);
      if (gs) dof_st_row<8>(gs + ACT(t, 8 * b, 4 * HID, Sp, s), gate8);
    }
  };
  dof_static_for<PF>([&](auto d) { load_x(d, decltype(d)::value); });
  input_half(xs[0]);
  for (int step = 0; step < T; step += PF) {  // wave-uniform trip count: MFMA ignores EXEC, finished sequences idle
    dof_static_for<PF>([&](auto d) { do_step(d, step + decltype(d)::value); });
  }
  if (in_range) {
    const float zero8[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int t = n; t < T; ++t) {
      dof_st_pair(O + ACT(t, dir * HID + 2 * b, 2 * HID, Sp, s), 0.0f, 0.0f);
      if (gs) dof_st_row<8>(gs + ACT(t, 8 * b, 4 * HID, Sp, s), zero8);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// GRU backward on the matrix pipe (IN = HID = 16), gates RECOMPUTED (round 3): the twin of k_gru16m_fwd.
// The forward pass of this layer saves no gates any more -- four floats per unit and step, 229 MB written per encoder
// stream and read back here, were the largest traffic item of the C2 step and both kernels ran at the per-CU
// load/store issue limit (~7-10 B/clk/CU), not at any arithmetic limit.  Here a step reads x_t, h_{t-1} and dO_t
// (3 x 16 B per lane), recomputes the gate pre-activations with the SAME 24 MFMAs in the same order as the forward
// kernel (bitwise the same r, z, n), and runs the transposed products on the matrix pipe too:
//   dh_{t-1} = dht * z + W_hh^T [g_r, g_z, g_h],   dx_t = W_ih^T [g_r, g_z, g_n]
// with A = a transposed weight tile (row = the output index, K-block q = gate units {q, 4+q, 8+q, 12+q}) and B = the
// lane's own four gate gradients: again no data moves between lanes, and the result lands in the lane that owns those
// units / input channels (one 16-byte dX store).
// Weight gradients contract over the SEQUENCE index, which is the lane index here: the step's gate gradients, x_t and
// h_{t-1} go through a 6 KB LDS tile ([sequence][unit], written as 16-byte words, read back as MFMA operands with the
// sequence as K) -- 24 more MFMAs into six 16 x 16 accumulator tiles that stay in registers for the whole sequence.
// 72 MFMAs per step and wavefront = 144 matrix-pipe cycles per (sequence, direction) and step (the VALU form: ~230
// VALU instructions per four pairs).  Per-wavefront partials in k_gru16_bwd_fused's layout -> k_gru16_wg_finalize.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_gru16m_bwd(Gru16mStream st_a, Gru16mStream st_b, int T) {
  constexpr int HID = 16, IN = 16;
  __shared__ __attribute__((aligned(16))) float tile[6][16][16];  // [g_r, g_z, g_n, g_h, x, h_prev][sequence][unit]
  const Gru16mStream& A = blockIdx.z ? st_b : st_a;
  const float* __restrict__ X = A.X;
  const int* __restrict__ len = A.len;
  const float* __restrict__ O = A.O;
  const float* __restrict__ dO = A.dO;
  float* __restrict__ dX = A.dX;
  float* __restrict__ wg_partial = A.wg_partial;
  const int64_t S = A.S, Sp = A.Sp;
  if ((int64_t)blockIdx.x * 16 >= S) return;   // (the grid covers the longer stream; whole workgroups leave)
  const unsigned nblk_own = (unsigned)((S + 15) / 16);   // partial rows of THIS stream (gridDim.x may be the other stream's)
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, b = lane >> 4;
  const int64_t s = (int64_t)blockIdx.x * 16 + j;
  const int dir = blockIdx.y;
  const bool in_range = s < S;
  const float* __restrict__ wih = dir ? A.wih1 : A.wih0;
  const float* __restrict__ whh = dir ? A.whh1 : A.whh0;
  const float* __restrict__ bih = dir ? A.bih1 : A.bih0;
  const float* __restrict__ bhh = dir ? A.bhh1 : A.bhh0;
  // forward tiles (row = unit lane & 15, K-block q = indices 4 (lane >> 4) + q) and transposed tiles (row = input /
  // hidden index lane & 15, K-block q = gate units 4 (lane >> 4) + q)
  float aix[3][4], ahh[3][4], tix[3][4], thh[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      aix[g][q] = wih[(g * HID + j) * IN + 4 * b + q];
      ahh[g][q] = whh[(g * HID + j) * HID + 4 * b + q];
      tix[g][q] = wih[(g * HID + 4 * b + q) * IN + j];
      thh[g][q] = whh[(g * HID + 4 * b + q) * HID + j];
    }
  dof_f32x4 c_r, c_z, c_n, c_hn;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int unit = 4 * b + r;
    c_r[r] = bih[unit] + bhh[unit];
    c_z[r] = bih[HID + unit] + bhh[HID + unit];
    c_n[r] = bih[2 * HID + unit];
    c_hn[r] = bhh[2 * HID + unit];
  }
  float* __restrict__ dx_out = dX + (int64_t)dir * T * IN * Sp;
  const int n = in_range ? len[s] : 0;
  float dh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  dof_f32x4 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[a] = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float sb[4][4];  // bias sums [r, z, n, h][unit 4b + r]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) sb[g][r] = 0.0f;
  // operands of a step: x_t, h_{t-1}, dO_t -- none depends on the recurrence; loaded PF steps ahead into static
  // register slots (loop unrolled by PF; see k_gru16m_fwd)
  constexpr int PF = 3;
  float nx_x[PF][4], nx_h[PF][4], nx_d[PF][4];
  const int64_t sr = in_range ? s : S - 1;  // loads are unconditional (a branch around a load costs a vmcnt(0) at the
                                            // join): idle lanes read a valid row and their values are zeroed at use
  auto issue_loads = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    const bool live = step >= 0 && step < n;
    const int t = live ? (dir ? (n - 1 - step) : step) : 0;
    const int tp = (live && step > 0) ? (dir ? t + 1 : t - 1) : 0;
    dof_ld_row<4>(X + ACT(t, 4 * b, IN, Sp, sr), nx_x[slot]);
    dof_ld_row<4>(O + ACT(tp, dir * HID + 4 * b, 2 * HID, Sp, sr), nx_h[slot]);
    if (dO) dof_ld_row<4>(dO + ACT(t, dir * HID + 4 * b, 2 * HID, Sp, sr), nx_d[slot]);   // (wave-uniform condition)
    else nx_d[slot][0] = nx_d[slot][1] = nx_d[slot][2] = nx_d[slot][3] = 0.0f;
  };
  auto do_step = [&](auto slot_c, int step) DOF_INLINE_LAMBDA {
    constexpr int slot = decltype(slot_c)::value;
    const bool act = step < n;
    float xv[4], hp[4], dov[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      xv[q] = act ? nx_x[slot][q] : 0.0f;
      hp[q] = (act && step > 0) ? nx_h[slot][q] : 0.0f;
      dov[q] = act ? nx_d[slot][q] : 0.0f;
    }
    issue_loads(slot_c, step - PF);
    // ---- gates, recomputed exactly as k_gru16m_fwd computes them (input half first, then the hidden half)
    dof_f32x4 a_r = c_r, a_z = c_z, a_n = c_n, a_hn = c_hn;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a_r = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[0][q], xv[q], a_r, 0, 0, 0);
      a_z = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[1][q], xv[q], a_z, 0, 0, 0);
      a_n = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[2][q], xv[q], a_n, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a_r = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][q], hp[q], a_r, 0, 0, 0);
      a_z = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][q], hp[q], a_z, 0, 0, 0);
      a_hn = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[2][q], hp[q], a_hn, 0, 0, 0);
    }
    float g_r[4], g_z[4], g_n[4], g_h[4];
    dof_f32x4 d_h, d_x = dof_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rr = dof_sigmoid(a_r[r]);
      const float z = dof_sigmoid(a_z[r]);
      const float nn = dof_tanh(fmaf(rr, a_hn[r], a_n[r]));
      const float dht = act ? dh[r] + dov[r] : 0.0f;
      const float dn = dht * (1.0f - z);
      const float dz = dht * (hp[r] - nn);
      const float dnp = dn * (1.0f - nn * nn);
      g_r[r] = act ? dnp * a_hn[r] * rr * (1.0f - rr) : 0.0f;
      g_z[r] = act ? dz * z * (1.0f - z) : 0.0f;
      g_n[r] = act ? dnp : 0.0f;
      g_h[r] = act ? dnp * rr : 0.0f;
      d_h[r] = dht * z;
      sb[0][r] += g_r[r]; sb[1][r] += g_z[r]; sb[2][r] += g_n[r]; sb[3][r] += g_h[r];
    }
    // ---- dh_{t-1} and dx_t: transposed tiles x the lane's own gate gradients
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      d_h = __builtin_amdgcn_mfma_f32_16x16x4f32(thh[0][q], g_r[q], d_h, 0, 0, 0);
      d_h = __builtin_amdgcn_mfma_f32_16x16x4f32(thh[1][q], g_z[q], d_h, 0, 0, 0);
      d_h = __builtin_amdgcn_mfma_f32_16x16x4f32(thh[2][q], g_h[q], d_h, 0, 0, 0);
      d_x = __builtin_amdgcn_mfma_f32_16x16x4f32(tix[0][q], g_r[q], d_x, 0, 0, 0);
      d_x = __builtin_amdgcn_mfma_f32_16x16x4f32(tix[1][q], g_z[q], d_x, 0, 0, 0);
      d_x = __builtin_amdgcn_mfma_f32_16x16x4f32(tix[2][q], g_n[q], d_x, 0, 0, 0);
    }
    // ---- weight gradients: contraction over the 16 sequences of the wavefront through the LDS tile
    __syncthreads();  // (one wavefront per workgroup: orders this step's writes after the previous step's reads)
    dof_st_row<4>(&tile[0][j][4 * b], g_r);
    dof_st_row<4>(&tile[1][j][4 * b], g_z);
    dof_st_row<4>(&tile[2][j][4 * b], g_n);
    dof_st_row<4>(&tile[3][j][4 * b], g_h);
    dof_st_row<4>(&tile[4][j][4 * b], xv);
    dof_st_row<4>(&tile[5][j][4 * b], hp);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // K-block q = sequences 4q .. 4q+3; A row = unit j, B column = input / hidden index j
      const float ar_ = tile[0][4 * q + b][j], az_ = tile[1][4 * q + b][j], an_ = tile[2][4 * q + b][j];
      const float ah_ = tile[3][4 * q + b][j], bx_ = tile[4][4 * q + b][j], bh_ = tile[5][4 * q + b][j];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar_, bx_, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(az_, bx_, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(an_, bx_, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar_, bh_, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(az_, bh_, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah_, bh_, acc[5], 0, 0, 0);
    }
    if (act) {
      const int t = dir ? (n - 1 - step) : step;
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = d_h[r];
      const float dx4[4] = {d_x[0], d_x[1], d_x[2], d_x[3]};
      dof_st_row<4>(dx_out + ACT(t, 4 * b, IN, Sp, s), dx4);
    }
  };
  // the loop starts at the first multiple of PF >= T: steps >= T (>= n) are idle, so that the unrolled group needs no
  // exit (with exits the compiler drained the memory counter at the loop header)
  const int top = (T + PF - 1) / PF * PF - 1;
  dof_static_for<PF>([&](auto d) { issue_loads(d, top - decltype(d)::value); });
  for (int step = top; step >= 0; step -= PF) {  // wave-uniform trip count (MFMA ignores EXEC)
    dof_static_for<PF>([&](auto d) { do_step(d, step - decltype(d)::value); });
  }
  if (in_range) {
    const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int t = n; t < T; ++t) dof_st_row<4>(dx_out + ACT(t, 4 * b, IN, Sp, s), zero4);
  }
  // ---- partials of this wavefront's 16 sequences: tiles [a][row = unit][col], then the bias sums [gate][unit]
  float* __restrict__ out = wg_partial + ((int64_t)dir * nblk_own + blockIdx.x) * GRU16_WG_FLOATS;
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

