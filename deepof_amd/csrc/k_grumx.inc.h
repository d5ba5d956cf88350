// GEMM-shaped GRU recurrences of the wider layers (latent 16: (32, 32), (64 -> 16); latent 32: (64, 64), (128 -> 32)) on
// v_mfma_f32_16x16x32_bf16 with exact three-piece operands (round 5; reference:
// /root/reference/deepof/clustering/models_new.py:184-278, torch.nn.GRU).  Included by k_rnn.hip inside its anonymous
// namespace.  k_grumx_fwd replaces round 4's k_grum_fwd, the same recurrence on v_mfma_f32_16x16x4_f32: at (64, 64) that
// issued 384 fp32 MFMAs per step and 16 sequences (15 - 20 ns each, nothing overlapping them: tools/probe), here a step
// is 288 bf16 MFMAs of ~9.5 ns (profiles/r05_mfma_rate_probe.txt).  The backward kernel (k_grum_bwd, k_rnn.hip) is still
// the fp32 form.
//
// Layout, as in k_grum_*: a workgroup = four wavefronts of ONE direction; lane (b, j) owns units 16 m + 4 b + r (m < HID / 16,
// r < 4) of sequence j -- the D layout of the 16 x 16 MFMA for row tile m.  One step is G[unit][seq] = W[unit][k] V[k][seq],
// V = [x_t ; h_{t-1}], with the K blocks made of the lane's OWN values: lane b loads x[b IN/4 .. (b + 1) IN/4) and has just
// computed the hidden units {16 m + 4 b + r}; a K block of the bf16 instruction is 8 of those values per lane -- IN / 32
// blocks of x, ceil(HID / 32) blocks of h (zero-padded), never mixed, so the n gate keeps W_in x and W_hn h apart.  Every
// value is cut into three bf16 pieces (dof_split3x4) and a product is the six piece products that carry more than 2^-24 of
// it (k_grum16.inc.h).  The A operands -- every (gate, row tile, K block, piece): 144 KB at (64, 64) -- are staged once per
// workgroup in LDS in operand order; one read of 16 bytes per lane feeds TWO MFMAs, because a wavefront carries two tiles of
// 16 sequences (at one tile per wavefront the four SIMDs of a CU would ask the LDS for exactly its peak 256 B / clk).
#pragma once

template <int IN, int HID>
struct DofGrumxDims {
  static_assert(HID % 16 == 0 && IN % 32 == 0, "row tiles of 16 units; 8 input values per lane and K block");
  static constexpr int MT = HID / 16, XL = IN / 4, KH = HID / 4;
  static constexpr int NXB = XL / 8, NHB = (KH + 7) / 8, NKB = NXB + NHB;
  static constexpr int NT = 2;   // tiles of 16 sequences per wavefront
};

// forward operand (gate g, row tile m, K block kb) of lane `ln`, before the cut: eight weights
template <int IN, int HID>
__device__ __forceinline__ void dof_grumx_fwd_weights(const float* __restrict__ wih, const float* __restrict__ whh, int g, int m,
                                                      int kb, int ln, float (&v)[8]) {
  using D = DofGrumxDims<IN, HID>;
  const int row = g * HID + 16 * m + (ln & 15), kq = ln >> 4;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (kb < D::NXB) {
      v[e] = wih[row * IN + kq * D::XL + 8 * kb + e];
    } else {
      const int q = 8 * (kb - D::NXB) + e;   // the lane group's own hidden value q: unit 16 (q / 4) + 4 kq + q % 4
      v[e] = q < D::KH ? whh[row * HID + 16 * (q >> 2) + 4 * kq + (q & 3)] : 0.0f;
    }
  }
}
__device__ __forceinline__ void dof_grumx_put(uint32_t (*dst)[4], int ln, const float (&v)[8], int stride_ops) {
  // the three pieces of eight values as three operands (dst, dst + stride_ops, dst + 2 stride_ops), 4 words per lane
  uint32_t lo[3][2], hi[3][2];
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v[0]), lo);
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v[4]), hi);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    uint32_t* o = dst[(int64_t)p * stride_ops * 64 + ln];
    o[0] = lo[p][0]; o[1] = lo[p][1]; o[2] = hi[p][0]; o[3] = hi[p][1];
  }
}
__device__ __forceinline__ void dof_grumx_pieces(const float* v8, dof_bf16x8 (&bq)[3]) {
  uint32_t lo[3][2], hi[3][2];
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v8[0]), lo);
  dof_split3x4(reinterpret_cast<const float(&)[4]>(v8[4]), hi);
#pragma unroll
  for (int p = 0; p < 3; ++p) bq[p] = dof_mk_bf16x8(lo[p][0], lo[p][1], hi[p][0], hi[p][1]);
}

template <int IN, int HID>
__global__ void __launch_bounds__(256, 1) k_grumx_fwd(const float* __restrict__ X, const int* __restrict__ len,
                                                      const float* __restrict__ wih0, const float* __restrict__ whh0,
                                                      const float* __restrict__ bih0, const float* __restrict__ bhh0,
                                                      const float* __restrict__ wih1, const float* __restrict__ whh1,
                                                      const float* __restrict__ bih1, const float* __restrict__ bhh1,
                                                      float* __restrict__ O, float* __restrict__ GS, int T, int64_t S,
                                                      int64_t Sp) {
  using D = DofGrumxDims<IN, HID>;
  constexpr int MT = D::MT, XL = D::XL, KH = D::KH, NXB = D::NXB, NKB = D::NKB, NT = D::NT;
  constexpr int NOPS = 3 * MT * NKB;   // (gate, row tile, K block); x3 pieces
  // [piece][(g * MT + m) * NKB + kb][lane][4 words]
  __shared__ __attribute__((aligned(16))) uint32_t wop[3 * NOPS][64][4];
  __shared__ __attribute__((aligned(16))) float bl[4 * HID];  // r: b_ih + b_hh | z: b_ih + b_hh | n: b_ih | hn: b_hh
  const int dir = blockIdx.y;
  {
    const float* __restrict__ g_wih = dir ? wih1 : wih0;
    const float* __restrict__ g_whh = dir ? whh1 : whh0;
    const float* __restrict__ g_bih = dir ? bih1 : bih0;
    const float* __restrict__ g_bhh = dir ? bhh1 : bhh0;
    for (int e = threadIdx.x; e < NOPS * 64; e += 256) {
      const int ln = e & 63, op = e >> 6;
      const int kb = op % NKB, gm = op / NKB;
      float v[8];
      dof_grumx_fwd_weights<IN, HID>(g_wih, g_whh, gm / MT, gm % MT, kb, ln, v);
      dof_grumx_put(wop[op], ln, v, NOPS);
    }
    for (int u = threadIdx.x; u < HID; u += 256) {
      bl[u] = g_bih[u] + g_bhh[u];
      bl[HID + u] = g_bih[HID + u] + g_bhh[HID + u];
      bl[2 * HID + u] = g_bih[2 * HID + u];
      bl[3 * HID + u] = g_bhh[2 * HID + u];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, b = lane >> 4;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * NT;
  if (tile0 * 16 >= S) return;   // whole wavefronts leave (no barrier below)
  int64_t s[NT], sr[NT];
  int n[NT];
  bool live[NT];
  int nmax = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    s[i] = (tile0 + i) * 16 + j;
    live[i] = s[i] < S;
    sr[i] = live[i] ? s[i] : S - 1;
    n[i] = live[i] ? len[sr[i]] : 0;
    nmax = n[i] > nmax ? n[i] : nmax;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {   // the longest sequence of the wavefront bounds the loop (MFMA ignores EXEC)
    const int o = __shfl_xor(nmax, m);
    nmax = o > nmax ? o : nmax;
  }
  float* __restrict__ gs = GS ? GS + (int64_t)dir * T * 4 * HID * Sp : nullptr;
  float h[NT][(D::NHB * 8)];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int q = 0; q < (D::NHB * 8); ++q) h[i][q] = 0.0f;
  float xa[NT][XL], xb[NT][XL];   // x of the current / the next step (loads unconditional: idle lanes read a valid row)
  auto load_x = [&](int step, float (&dst)[NT][XL]) DOF_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int t = step < n[i] ? (dir ? (n[i] - 1 - step) : step) : 0;
      dof_ld_row<XL>(X + ACT(t, b * XL, IN, Sp, sr[i]), dst[i]);
    }
  };
  load_x(0, xa);
  auto Wop = [&](int op, int p) DOF_INLINE_LAMBDA {
    return dof_ld_bf16x8(reinterpret_cast<const uint16_t*>(&wop[p * NOPS + op][lane][0]));
  };
  for (int step = 0; step < nmax; ++step) {
    DOF_MEM_FENCE();   // keeps the operand reads inside the step
    load_x(step + 1, xb);
    dof_f32x4 a_r[NT][MT], a_z[NT][MT], a_n[NT][MT], a_h[NT][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float c[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) dof_ld_row<4>(&bl[g * HID + 16 * m + 4 * b], c[g]);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        a_r[i][m] = dof_f32x4{c[0][0], c[0][1], c[0][2], c[0][3]};
        a_z[i][m] = dof_f32x4{c[1][0], c[1][1], c[1][2], c[1][3]};
        a_n[i][m] = dof_f32x4{c[2][0], c[2][1], c[2][2], c[2][3]};
        a_h[i][m] = dof_f32x4{c[3][0], c[3][1], c[3][2], c[3][3]};
      }
    }
    dof_static_for<NKB>([&](auto kbc) {
      constexpr int kb = decltype(kbc)::value;
      constexpr bool xblk = kb < NXB;
      dof_bf16x8 bq[NT][3];
#pragma unroll
      for (int i = 0; i < NT; ++i) dof_grumx_pieces(xblk ? &xa[i][8 * kb] : &h[i][8 * (kb - NXB)], bq[i]);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        dof_bf16x8 A[3][3];   // [gate][piece]
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int p = 0; p < 3; ++p) A[g][p] = Wop((g * MT + m) * NKB + kb, p);
        // smallest products first; six independent accumulators (3 gates x 2 tiles) between dependent MFMAs
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            a_r[i][m] = DOF_MFMA_16x16x32_BF16(A[0][PA[k]], bq[i][PB[k]], a_r[i][m]);
            a_z[i][m] = DOF_MFMA_16x16x32_BF16(A[1][PA[k]], bq[i][PB[k]], a_z[i][m]);
            if constexpr (xblk) a_n[i][m] = DOF_MFMA_16x16x32_BF16(A[2][PA[k]], bq[i][PB[k]], a_n[i][m]);
            else a_h[i][m] = DOF_MFMA_16x16x32_BF16(A[2][PA[k]], bq[i][PB[k]], a_h[i][m]);
          }
      }
    });
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool act = step < n[i];
      const int t = dir ? (n[i] - 1 - step) : step;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float hn4[4], r4[4], z4[4], n4[4], a4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float rr = dof_sigmoid(a_r[i][m][r]);
          const float zz = dof_sigmoid(a_z[i][m][r]);
          const float nn = dof_tanh(fmaf(rr, a_h[i][m][r], a_n[i][m][r]));
          const float hnew = fmaf(zz, h[i][4 * m + r] - nn, nn);
          h[i][4 * m + r] = act ? hnew : h[i][4 * m + r];
          hn4[r] = hnew; r4[r] = rr; z4[r] = zz; n4[r] = nn; a4[r] = a_h[i][m][r];
        }
        if (act) {
          dof_st_row<4>(O + ACT(t, dir * HID + 16 * m + 4 * b, 2 * HID, Sp, s[i]), hn4);
          if (gs) {
            dof_st_row<4>(gs + ACT(t, 16 * m + 4 * b, 4 * HID, Sp, s[i]), r4);
            dof_st_row<4>(gs + ACT(t, HID + 16 * m + 4 * b, 4 * HID, Sp, s[i]), z4);
            dof_st_row<4>(gs + ACT(t, 2 * HID + 16 * m + 4 * b, 4 * HID, Sp, s[i]), n4);
            dof_st_row<4>(gs + ACT(t, 3 * HID + 16 * m + 4 * b, 4 * HID, Sp, s[i]), a4);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int k = 0; k < XL; ++k) xa[i][k] = xb[i][k];
  }
  const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (!live[i]) continue;
    for (int t = n[i]; t < T; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        dof_st_row<4>(O + ACT(t, dir * HID + 16 * m + 4 * b, 2 * HID, Sp, s[i]), zero4);
        if (gs) {
#pragma unroll
          for (int g = 0; g < 4; ++g) dof_st_row<4>(gs + ACT(t, g * HID + 16 * m + 4 * b, 4 * HID, Sp, s[i]), zero4);
        }
      }
  }
}
