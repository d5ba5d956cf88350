// Round 6: the time-resident 32 -> 32 convolution of k_tcn.hip on the bf16 matrix pipe with exact three-piece operands,
// and the convolution's WEIGHT GRADIENT accumulated in the same launch.  Included by k_tcn.hip inside its namespace.
//
// What changes against k_tcn_conv_t (same arguments, same epilogues, same results to fp32 rounding):
//   * the staged tile is cut ONCE, while it is staged, into three bf16 planes (value = p0 + p1 + p2 exactly, dof_split3x4);
//     a product is the six piece products that carry more than 2^-24 of it, each one
//     v_mfma_f32_16x16x32_bf16 whose K = the 32 input channels of a tap: 24 matrix instructions per 16 x 16 output tile
//     against 32 v_mfma_f32_16x16x4_f32, at ~1/2 of the time each;
//   * a tile is NS = 8 sequences (windows <= 25 steps) or 4 (<= 50): an MFMA column block is NS sequences x 16 / NS
//     consecutive output rows.  Image = 224 (row, sequence) pairs x 32 channels x 3 planes of bf16 = 42 KB; rows behind the
//     window are staged as zeros, so a tap that leaves the window reads the zero row T instead of being masked per lane;
//   * LOADER and COMPUTE wavefronts (one 512-thread workgroup per CU, two images): wavefronts 4 - 7 stage tile n + 1
//     (BatchNorm / BatchNorm-backward arithmetic, the cut, the LDS writes) into one image while wavefronts 0 - 3 run the
//     matrix phase and the epilogues of tile n on the other; the loaders' global loads are issued one tile further ahead
//     (tile n + 2's loads replace a pass's registers as soon as the pass is written), the compute wavefronts request a
//     column block's epilogue operands one round ahead.  PMC of the single-role form (every wavefront staging, then
//     computing, two workgroups per CU): 60 - 75 % of the wave cycles parked in s_waitcnt / s_barrier;
//   * a compute wavefront owns BOTH channel halves of its column block (the B operand is read from LDS once), wavefront w
//     takes the column blocks 4 r + w of round r; a lane's quad of half ct is 16 bytes of the sequence's 128-byte row, the
//     four lanes of a sequence cover a contiguous 64-byte half row per instruction;
//   * WGRAD (the two data-gradient variants of the backward pass, k_tcn_conv_t's <true,false,true,true,*>): the kernel has
//     dy = the staged tile on chip and the convolution's forward input x on its way through the epilogue -- conv2: x =
//     ReLU(BN1(y1)), recomputed from the row of y1 the BatchNorm-backward epilogue loads anyway; conv1 (TAIL): x = the
//     previous block's output, one extra read that also replaces the mask words.  dW_j[o][c] = sum over (t, s) of
//     dy[t + (3 - j) d][s][o] x[t][s][c] is a GEMM whose K runs over (t, s): the column block's x rows go through a small
//     LDS ring (cut into pieces), a round's four column blocks are exchanged at ONE barrier, and compute wavefront j
//     accumulates tap j on v_mfma_f32_32x32x16_bf16 with both operands read TRANSPOSED from their channel-minor images
//     (ds_read_b64_tr_b16).  The separate weight-gradient kernel (k_tcn_wgrad_b3: three more reads of three tensors per
//     convolution, 66 GB of the C4 step's 262) is not launched for these convolutions; the partial tiles keep its layout.
//
// LDS images.  Plane p of a tile: [row = t NS + s][32 channels] bf16, 64 bytes per row; the 16-byte chunk c of a row sits
// at chunk c ^ 2 ((row >> 2) & 1): with it the sixteen lanes of every ds_read_b128 service group (lane l reads chunk l >> 4
// of row R + (l & 15)) hit sixteen different 16-byte slots of the 256-byte bank row for every R that is a multiple of 4
// (brute-forced over the four groups of the microarchitecture guide's table; SQ_LDS_BANK_CONFLICT = 0 measured).  The
// transposing reads take four whole consecutive rows per 32 lanes: conflict-free under any in-row permutation.  Ring slot
// plane: [column 16][32 channels], 8-byte chunk c8 at c8 ^ ((column >> 1) & 7) (the epilogue's ds_write_b64 of sixteen
// columns, same chunk, then spreads over all banks).
// development probe (tools/probe/tcn_conv_probe.hip compiles this file with TCN_PROBE_STAMPS): cycle stamps of one tile of
// workgroup 0 -- slot 32 w + 8 r + {0 round start, 1 matrix phase issued, 2 epilogue + requests issued, 3 past the round barrier,
// 4 weight-gradient phase issued} for compute wavefront w, 128 + 16 (w - 4) + {0 start, 1 staged, 2 loads issued, 3 past the
// tile barrier} for loader wavefront w
#ifdef TCN_PROBE_STAMPS
__device__ unsigned long long g_tcn_stamps[256];
#define TCN_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && k == TCN_PROBE_STAMPS && (slot) < 256 && ((slot) >= 128 || (threadIdx.x >> 6) < 4)) g_tcn_stamps[(slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define TCN_STAMP(slot) ((void)0)
#endif
constexpr int TB_ROWS = 224;
constexpr int TB_PLANE = TB_ROWS * 32;  // bf16 elements per plane
constexpr int TB_IMG = 3 * TB_PLANE;    // ... per image
constexpr int TB_RING = 16 * 32;        // bf16 elements per ring slot plane

__device__ __forceinline__ int tb_off(int row, int chunk) { return row * 32 + ((chunk ^ (((row >> 2) & 1) << 1)) << 3); }
__device__ __forceinline__ int tb_ring_off(int col, int c8) { return col * 32 + ((c8 ^ ((col >> 1) & 7)) << 2); }
__device__ __forceinline__ void tb_st8(uint16_t* p, uint32_t lo, uint32_t hi) {
  *reinterpret_cast<uint64_t*>(p) = (uint64_t)lo | ((uint64_t)hi << 32);
}

// NCW = compute wavefronts: 4 (512 threads, <= 256 registers) or 8 (768 threads, <= 168 registers: two compute wavefronts
// per SIMD share its matrix pipe and hide each other's epilogue)
// DS0 (COMB of block 1): the residual of block 0 is not a stored tensor but the 1 x 1 convolution of the raw input -- the
// loaders read the row's ds_F input values instead of 32 channels and evaluate it (k_tcn_combine's arithmetic); block 0's
// full-size combine launch (409 us at C4) is then a last-step-only one like the other blocks'
template <bool REVERSE, bool BN_IN, bool FUSE_BN, bool BWD2, bool TAIL, bool COMB, bool WGRAD, int NS, int NCW = 4, bool DS0 = false>
__global__ void __launch_bounds__(64 * (NCW + 4), NCW == 8 ? 3 : 2) k_tcn_conv_b(TcnConvArgs A) {
  static_assert(NS == 8 || NS == 4, "8 sequences x 25 steps or 4 sequences x 50 steps");
  static_assert(NCW == 4 || NCW == 8, "four or eight compute wavefronts");
  constexpr int MAXR = 16 / NCW;      // rounds of a tile (at most 13 column blocks)
  constexpr int TPC = 16 / NS;        // output rows per MFMA column block
  constexpr int TS = 256 / (NS * 8);  // time steps per staging pass of the 256 loader threads
  constexpr int NP = TB_ROWS / NS / TS;
  constexpr bool TWO = BWD2 || COMB;  // the staging reads two tensors
  static_assert(!TAIL || (REVERSE && FUSE_BN && BWD2), "the tail epilogue extends the fused data-gradient variant");
  static_assert(!COMB || (!REVERSE && !BN_IN && !FUSE_BN && !BWD2), "the combine-on-load variant is a plain forward convolution");
  static_assert(!WGRAD || (REVERSE && FUSE_BN && BWD2), "the weight gradient rides on the fused data-gradient variants");
  static_assert(!DS0 || COMB, "the downsample residual belongs to the combine-on-load variant");
  __shared__ __attribute__((aligned(16))) uint16_t img[2 * TB_IMG];
  __shared__ __attribute__((aligned(16))) uint16_t ring[WGRAD ? 2 * NCW * 3 * TB_RING : 8];
  // the A operands (weights): [tap][piece][channel half][lane] x 16 bytes, every compute wavefront reads the same 24 KB (in
  // registers they cost 96 VGPRs per lane, which left no room to request LDS operands ahead of the matrix instructions)
  __shared__ __attribute__((aligned(16))) uint16_t wlds[TK * 3 * 2 * 64 * 8];
  __shared__ float4 frec[FUSE_BN ? 4 * TC / 4 : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool loader = wv >= NCW;  // wave-uniform role
  const int T = A.T;
  const int n_cb = (T + TPC - 1) / TPC, n_rounds = (n_cb + NCW - 1) / NCW;
  const int np_run = T / TS + 1;  // staging passes that reach row T (the zero row); T < NP TS is the launcher's condition
  const int64_t n_groups = A.Sp / NS;
  const uint32_t row_stride = (uint32_t)A.Sp * TC;  // 32-bit element offsets (the launcher checks T * Sp * 32 < 2^31)
  if (FUSE_BN) {
    if (threadIdx.x < 4 * TC / 4) frec[threadIdx.x] = reinterpret_cast<const float4*>(A.fuse_bnp)[threadIdx.x];
  }
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // WGRAD, loaders: the thread's channel sums of dy (bias gradient)
  float s1[2][4], s2[2][4], kshift[2][4];   // compute: channel sums of the lane's output values
  float n_rows = 0.0f;                     // stat_records: rows this lane has summed; its sums are taken about the first one
  dof_f32x16 accw, accw2;                  // WGRAD, compute: tap wv of the weight gradient, 32 x 32, in two partial sums
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[ct][r] = s2[ct][r] = kshift[ct][r] = 0.0f;
#pragma unroll
  for (int v = 0; v < 16; ++v) accw[v] = accw2[v] = 0.0f;

  if (loader) {
    // =========================================== loader wavefronts ===========================================
    // thread -> (time step of the pass, sequence, 8-byte chunk); its four channels are fixed
    const int lt = threadIdx.x - 64 * NCW;
    const int tq = lt / (NS * 8), sq = (lt % (NS * 8)) >> 3, ch = lt & 7;
    float k0[4], k1[4];                        // BN_IN / COMB: scale, shift of the producer's BatchNorm
    float bm[4], br[4], bs[4], c1[4], c2[4];  // BWD2: mean, rstd, scale, mean g, mean g xhat
    if (BN_IN || COMB) {
      dof_ld_row<4>(A.bnp_in + 2 * TC + ch * 4, k0);
      dof_ld_row<4>(A.bnp_in + 3 * TC + ch * 4, k1);
    }
    float dw[4][3], db[4];   // DS0: the thread's four rows of the downsample convolution (absent input channels: weight 0)
    const int dsF = DS0 ? A.ds_F : 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      db[c] = DS0 ? A.ds_b[ch * 4 + c] : 0.0f;
#pragma unroll
      for (int f = 0; f < 3; ++f) dw[c][f] = (DS0 && f < dsF) ? A.ds_w[(ch * 4 + c) * dsF + f] : 0.0f;
    }
    if (BWD2) {
      dof_ld_row<4>(A.bwd_bnp + ch * 4, bm);
      dof_ld_row<4>(A.bwd_bnp + TC + ch * 4, br);
      dof_ld_row<4>(A.bwd_bnp + 2 * TC + ch * 4, bs);
      dof_ld_row<4>(A.bwd_coef + ch * 4, c1);
      dof_ld_row<4>(A.bwd_coef + TC + ch * 4, c2);
    }
    float4 v[NP], yv[TWO ? NP : 1];
    // the loads of pass u of tile g (unconditional: steps past T re-read step T - 1 and are staged as zeros; a tile past
    // the end re-reads the last one and is never staged)
    auto issue = [&](int u, int64_t g) DOF_INLINE_LAMBDA {
      const int64_t gc = g < n_groups ? g : n_groups - 1;
      const int t = TS * u + tq;
      const uint32_t off = (uint32_t)(gc * NS + sq) * TC + ch * 4 + (uint32_t)(t < T ? t : T - 1) * row_stride;
      if (DS0) {   // the row's ds_F raw input values (an absent channel re-reads channel 0 and meets a zero weight)
        const uint32_t xo = ((uint32_t)(gc * NS + sq) + (uint32_t)(t < T ? t : T - 1) * (uint32_t)A.Sp) * (uint32_t)dsF;
        v[u] = make_float4(A.in[xo], A.in[xo + (dsF > 1 ? 1 : 0)], A.in[xo + (dsF > 2 ? 2 : 0)], 0.0f);
      } else {
        v[u] = *reinterpret_cast<const float4*>(A.in + off);
      }
      if (TWO) yv[TWO ? u : 0] = *reinterpret_cast<const float4*>(A.bwd_y + off);
    };
    // pass u of tile g from its registers into image `buf`
    auto stage = [&](int u, int64_t g, int buf) DOF_INLINE_LAMBDA {
      const int64_t s0 = g * NS;
      const int t = TS * u + tq;
      const bool live = s0 + sq < A.S && t < T;
      float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      if (DS0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float acc = db[c];
          acc = fmaf(dw[c][0], v[u].x, acc);
          acc = fmaf(dw[c][1], v[u].y, acc);
          acc = fmaf(dw[c][2], v[u].z, acc);
          e[c] = acc;
        }
      }
      if (BN_IN) {
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] = fmaxf(fmaf(e[c], k0[c], k1[c]), 0.0f);
      }
      if (COMB) {
        const float4 yq = yv[TWO ? u : 0];
        const float y4[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] = fmaxf(fmaxf(fmaf(y4[c], k0[c], k1[c]), 0.0f) + e[c], 0.0f);
      }
      if (BWD2) {
        const float4 yq = yv[TWO ? u : 0];
        const float y4[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float xh = (y4[c] - bm[c]) * br[c];
          e[c] = bs[c] * (e[c] - c1[c] - xh * c2[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) e[c] = live ? e[c] : 0.0f;
      if (WGRAD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) rs[c] += e[c];
      }
      uint32_t pw[3][2];
      dof_split3x4(e, pw);
      const int row = t * NS + sq, eo = buf * TB_IMG + tb_off(row, ch >> 1) + (ch & 1) * 4;
#pragma unroll
      for (int p = 0; p < 3; ++p) tb_st8(&img[p * TB_PLANE + eo], pw[p][0], pw[p][1]);
      if (COMB) {  // the row's 32 sign bits for the TAIL convolution of the backward pass
        const uint32_t wbits = tct_row_mask(e, ch);
        if (ch == 0 && live && A.relu_mask_out) A.relu_mask_out[(uint32_t)t * (uint32_t)A.Sp + (uint32_t)(s0 + sq)] = wbits;
      }
      if (live) {
        const uint32_t off = (uint32_t)(s0 + sq) * TC + ch * 4 + (uint32_t)t * row_stride;
        const float4 w4 = make_float4(e[0], e[1], e[2], e[3]);
        if ((BN_IN || COMB) && !REVERSE && A.a_out) *reinterpret_cast<float4*>(A.a_out + off) = w4;
        if (BWD2 && A.bwd_store) *reinterpret_cast<float4*>(const_cast<float*>(A.in) + off) = w4;
      }
    };
    // Iteration k of the workgroup (k = 0 .. tiles): the loaders stage tile k from their registers into image k & 1 and
    // then issue tile k + 1's loads (a whole iteration passes before they are read); the compute wavefronts work on tile
    // k - 1.  WGRAD: behind each of the compute wavefronts' round barriers loader wavefront j accumulates tap j of the
    // weight gradient over the round's four column blocks -- the matrix work of a tile is shared between the two roles
    // (compute: 4 x 48 convolution instructions + epilogues; loaders: staging + 4 x 24 weight-gradient instructions).
    // The seven passes are straight-line code: a branch between two passes makes the compiler's wait-count pass merge
    // states at the join, and with loads and the staging's stores on one counter that merge becomes s_waitcnt vmcnt(0)
    // in front of EVERY load (measured: the seven loads of a tile took 10,000 cycles to issue, one after the other).
    // Passes beyond the window (T < 24) stage zero rows nobody reads.
    const int tapw = wv - NCW;                   // WGRAD: this wavefront's tap
    const int shw = (TK - 1 - tapw) * A.dil;
    const int g = lane >> 4, q = lane & 15, kh = g >> 1, mh = g & 1;
    // The round's four column blocks, straight-line (a block past the last one, or a tap that lies behind the window,
    // multiplies the zero row: nothing to skip); the products alternate between two accumulators (a v_mfma_f32_32x32x16
    // that waits for its own previous result costs its full latency) and block w2 + 1's twelve transposing reads are
    // requested before block w2's six matrix instructions.
    auto wgrad_round = [&](int r, const uint16_t* im) DOF_INLINE_LAMBDA {
      constexpr bool WDB = NCW == 4;   // operands one block ahead (NCW = 8: the 168-register cap leaves room for one set only)
      dof_bf16x8 wa[WDB ? 2 : 1][3], wb[WDB ? 2 : 1][3];
      auto wrequest = [&](int w2, int slot_i) DOF_INLINE_LAMBDA {
        const int t02 = (NCW * r + w2) * TPC;
        const uint16_t* slot = &ring[((r & 1) * NCW + w2) * 3 * TB_RING];
        uint32_t aw[3][4], bw[3][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kcol = 8 * kh + 4 * h + (q >> 2);  // the K index (column of the block) whose run this lane supplies
          const int tt = t02 + kcol / NS + shw;
          const int row = (tt < T ? tt : T) * NS + kcol % NS;
          // A row m = 16 mh + (lane & 15) = output channel m: the run of lane q covers the four channels 16 mh + 4 (q & 3) .. + 3 =
          // 16-byte chunk 2 mh + ((q & 3) >> 1), its half q & 1
          const int ea = tb_off(row, mh * 2 + ((q & 3) >> 1)) + (q & 1) * 4;
          const int eb = tb_ring_off(kcol, mh * 4 + (q & 3));
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            dof_lds_tr16(&im[p * TB_PLANE + ea], aw[p][2 * h], aw[p][2 * h + 1]);
            dof_lds_tr16(&slot[p * TB_RING + eb], bw[p][2 * h], bw[p][2 * h + 1]);
          }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          wa[slot_i][p] = dof_mk_bf16x8(aw[p][0], aw[p][1], aw[p][2], aw[p][3]);
          wb[slot_i][p] = dof_mk_bf16x8(bw[p][0], bw[p][1], bw[p][2], bw[p][3]);
        }
      };
      if (WDB) wrequest(0, 0);
#pragma unroll
      for (int w2 = 0; w2 < NCW; ++w2) {
        const int c = WDB ? (w2 & 1) : 0;
        if (WDB) {
          if (w2 + 1 < NCW) wrequest(w2 + 1, c ^ 1);
        } else {
          wrequest(w2, 0);
        }
        DOF_SCHED_FENCE();
        if (WDB) {
          accw = DOF_MFMA_32x32x16_BF16(wa[c][0], wb[c][2], accw);
          accw2 = DOF_MFMA_32x32x16_BF16(wa[c][2], wb[c][0], accw2);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][1], wb[c][1], accw);
          accw2 = DOF_MFMA_32x32x16_BF16(wa[c][0], wb[c][1], accw2);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][1], wb[c][0], accw);
          accw2 = DOF_MFMA_32x32x16_BF16(wa[c][0], wb[c][0], accw2);
        } else {   // one accumulator: the SIMD's two compute wavefronts fill the matrix pipe between these
          accw = DOF_MFMA_32x32x16_BF16(wa[c][0], wb[c][2], accw);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][2], wb[c][0], accw);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][1], wb[c][1], accw);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][0], wb[c][1], accw);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][1], wb[c][0], accw);
          accw = DOF_MFMA_32x32x16_BF16(wa[c][0], wb[c][0], accw);
        }
        DOF_SCHED_FENCE();
      }
    };
    const int64_t g0 = blockIdx.x, gs = gridDim.x;
#pragma unroll
    for (int u = 0; u < NP; ++u) issue(u, g0);
    int k = 0;
    for (int64_t grp = g0;; grp += gs, ++k) {
      const bool have = grp < n_groups;  // tile k exists (the last iteration only lets the compute wavefronts finish)
      TCN_STAMP(128 + 16 * (wv - NCW) + 0);
      if (have) {
#pragma unroll
        for (int u = 0; u < NP; ++u) stage(u, grp, k & 1);
      }
      TCN_STAMP(128 + 16 * (wv - NCW) + 1);
      // tile k + 1's loads (a tile past the end re-reads the last one).  WGRAD: behind the first round barrier -- the CU's
      // vector-memory queue is full at this point of a tile and issuing fourteen loads takes ~2,000 cycles, which the
      // compute wavefronts would otherwise spend waiting at that barrier
      if (have && !(WGRAD && k >= 1)) {
#pragma unroll
        for (int u = 0; u < NP; ++u) issue(u, grp + gs);
      }
      if (WGRAD && k >= 1) {
        const uint16_t* im = &img[((k - 1) & 1) * TB_IMG];
        for (int r = 0; r < n_rounds; ++r) {
          __syncthreads();  // the round's four ring slots are complete (double-buffered: the next round writes the other set)
          if (r == 0 && have) {
#pragma unroll
            for (int u = 0; u < NP; ++u) issue(u, grp + gs);
          }
          wgrad_round(r, im);
          TCN_STAMP(128 + 16 * (wv - NCW) + 4 + r);
        }
      }
      TCN_STAMP(128 + 16 * (wv - NCW) + 2);
      __syncthreads();  // tile k - 1 is consumed, tile k is staged
      TCN_STAMP(128 + 16 * (wv - NCW) + 3);
      if (!have) break;
    }
  } else {
    // =========================================== compute wavefronts ===========================================
    const int i = lane & 15, kk = lane >> 4;
    const int sl = i % NS, tsub = i / NS;  // the lane's sequence of the tile and its row of the column block
    // A operands: for tap j, piece p, channel half ct the 16 x 32 weight block of output channels ct 16 + (lane & 15), input
    // channels kk 8 .. + 7.  Compute wavefront j cuts tap j's.
    if (wv < TK) {
      const int j = wv;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float v0[4], v1[4];
        const int col = ct * 16 + i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c0 = kk * 8 + e, c1 = kk * 8 + 4 + e;
          v0[e] = REVERSE ? A.w[(c0 * TC + col) * TK + j] : A.w[(col * TC + c0) * TK + j];
          v1[e] = REVERSE ? A.w[(c1 * TC + col) * TK + j] : A.w[(col * TC + c1) * TK + j];
        }
        uint32_t s0w[3][2], s1w[3][2];
        dof_split3x4(v0, s0w);
        dof_split3x4(v1, s1w);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          uint16_t* dst = &wlds[(((j * 3 + p) * 2 + ct) * 64 + lane) * 8];
          tb_st8(dst, s0w[p][0], s0w[p][1]);
          tb_st8(dst + 4, s1w[p][0], s1w[p][1]);
        }
      }
    }
    if (WGRAD) {  // the ring is read before every slot has been written once (short last round): no NaN patterns in it
      for (int e = lane + 64 * wv; e < 2 * NCW * 3 * TB_RING / 2; e += 64 * NCW) reinterpret_cast<uint32_t*>(ring)[e] = 0u;
    }
    const uint16_t* wl = &wlds[lane * 8];
    // the lane's output channels: ct 16 + kk 4 + q
    float bias[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bias[ct][q] = (!REVERSE && A.bias) ? A.bias[ct * 16 + kk * 4 + q] : 0.0f;
        kshift[ct][q] = (!REVERSE && A.stat_shift) ? A.stat_shift[ct * 16 + kk * 4 + q] : 0.0f;
      }
    const float* pre_src = FUSE_BN ? A.fuse_y : (const float*)A.out;
    const bool pre_on = REVERSE && (FUSE_BN || A.accumulate);
    const int64_t g0 = blockIdx.x, gs = gridDim.x;
    __syncthreads();  // iteration 0: image 0 is being staged (and frec is loaded)
    int k = 1;
    for (int64_t grp = g0; grp < n_groups; grp += gs, ++k) {
      const int64_t s0 = grp * NS;
      const uint16_t* im = &img[((k - 1) & 1) * TB_IMG];
      const int64_t s = s0 + sl;
      const bool ok_s = s < A.S;
      // epilogue operands of the lane's four column blocks: two quads per tensor and round.  Straight-line
      // code inside the tile -- a value carried around a loop edge costs the compiler a copy, i.e. a wait right behind the load --
      // and unconditional loads from clamped addresses (a predicate would merge old and new values: the same copies).
      // TAIL: the wavefront whose column block holds the last time step also needs that row of the skip sum and the feature
      // gradient of its sequences
      float4 skl[2];
      float dfe[2][4];
      const bool own_last = TAIL && A.tail_dfeat != nullptr && wv == ((n_cb - 1) % NCW);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        skl[ct] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int q = 0; q < 4; ++q) dfe[ct][q] = 0.0f;
      }
      constexpr int PD = (TAIL && WGRAD && MAXR > 3) ? 3 : MAXR;   // rounds requested ahead (three tensors x four rounds do not fit the registers)
      float4 pre4[PD][2], tsv4[PD][2], xo4[PD][2];
      uint32_t tmw4[PD];
#pragma unroll
      for (int r = 0; r < PD; ++r) {
        tmw4[r] = 0u;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) pre4[r][ct] = tsv4[r][ct] = xo4[r][ct] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
      auto prefetch = [&](int r) DOF_INLINE_LAMBDA {
        float4 (&pre)[2] = pre4[r % PD];
        float4 (&tsv)[2] = tsv4[r % PD];
        float4 (&xo)[2] = xo4[r % PD];
        uint32_t& tmw = tmw4[r % PD];
        const int cb = NCW * r + wv;
        const int t = (cb < n_cb ? cb : n_cb - 1) * TPC + tsub;
        const uint32_t sv = (uint32_t)(ok_s ? s : s0);  // padded lanes read a valid row and ignore it
        const uint32_t tv = (uint32_t)(t < T ? t : T - 1);
        const uint32_t off = sv * TC + kk * 4 + tv * row_stride;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (pre_on) pre[ct] = *reinterpret_cast<const float4*>(pre_src + off + ct * 16);
          if (TAIL) tsv[ct] = *reinterpret_cast<const float4*>(A.tail_src + off + ct * 16);
          if (TAIL && WGRAD) xo[ct] = *reinterpret_cast<const float4*>(A.wg_x + off + ct * 16);
        }
        if (TAIL && !WGRAD) tmw = A.tail_mask[sv + tv * (uint32_t)A.Sp];
      };
      auto round = [&](int r) DOF_INLINE_LAMBDA {
        const float4 (&pre)[2] = pre4[r % PD];
        const float4 (&tsv)[2] = tsv4[r % PD];
        const float4 (&xo)[2] = xo4[r % PD];
        const uint32_t tmw = tmw4[r % PD];
        const int cb = NCW * r + wv;
        TCN_STAMP(32 * wv + 8 * r + 0);
        if (cb < n_cb) {
          const int t0 = cb * TPC, t = t0 + tsub;
          const bool ok = ok_s && t < T;
          dof_f32x4 acc[2];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[ct] = dof_f32x4{bias[ct][0], bias[ct][1], bias[ct][2], bias[ct][3]};
          // Straight-line over the taps: a lane whose row leaves the window under a tap reads the zero row (no branch inside:
          // a branch would cut the matrix phase into basic blocks and the LDS requests could not be placed ahead of the
          // previous tap's matrix instructions); tap j + 1's nine operands (three pieces of B, three pieces x two halves of A)
          // are requested before tap j's twelve matrix instructions.
          constexpr bool ADB = NCW == 4;   // NCW = 8: the weights of a tap are requested with the tap (one register set)
          dof_bf16x8 bq[2][3], aq[ADB ? 2 : 1][3][2];
          auto request_a = [&](int j, int slot) DOF_INLINE_LAMBDA {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              aq[slot][p][0] = dof_ld_bf16x8_16(&wl[((j * 3 + p) * 2 + 0) * 512]);
              aq[slot][p][1] = dof_ld_bf16x8_16(&wl[((j * 3 + p) * 2 + 1) * 512]);
            }
          };
          auto request = [&](int j, int slot) DOF_INLINE_LAMBDA {
            const int sh = REVERSE ? (TK - 1 - j) * A.dil : -(TK - 1 - j) * A.dil;
            const int tt = t + sh;
            const int row = ((tt >= 0 && tt < T) ? tt : T) * NS + sl;
            const int eo = tb_off(row, kk);
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[slot][p] = dof_ld_bf16x8_16(&im[p * TB_PLANE + eo]);
            if (ADB) request_a(j, slot);
          };
          // the taps that reach into the window for some row of the block are the last NT ones (wave-uniform); one
          // straight-line instance per count (a dilation-8 block of a 25-step window has 2.1 of 4 on average)
          auto taps = [&](auto ntc) DOF_INLINE_LAMBDA {
            constexpr int NT = decltype(ntc)::value;
            request(TK - NT, 0);
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
              const int c = jj & 1;
              if (!ADB) request_a(TK - NT + jj, 0);
              if (jj + 1 < NT) request(TK - NT + jj + 1, c ^ 1);
              DOF_SCHED_FENCE();  // (left alone the scheduler sinks every request to just in front of its use: ~12 exposed LDS round trips per block)
              // small terms first; the two channel halves are independent accumulator chains
              acc[0] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][0][0], bq[c][2], acc[0]);
              acc[1] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][0][1], bq[c][2], acc[1]);
              acc[0] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][2][0], bq[c][0], acc[0]);
              acc[1] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][2][1], bq[c][0], acc[1]);
              acc[0] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][1][0], bq[c][1], acc[0]);
              acc[1] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][1][1], bq[c][1], acc[1]);
              acc[0] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][0][0], bq[c][1], acc[0]);
              acc[1] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][0][1], bq[c][1], acc[1]);
              acc[0] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][1][0], bq[c][0], acc[0]);
              acc[1] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][1][1], bq[c][0], acc[1]);
              acc[0] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][0][0], bq[c][0], acc[0]);
              acc[1] = DOF_MFMA_16x16x32_BF16(aq[ADB ? c : 0][0][1], bq[c][0], acc[1]);
              DOF_SCHED_FENCE();
            }
          };
          // REVERSE: tap j reads row t + (3 - j) d: inside for the block's first row iff (3 - j) d < T - t0;
          // forward: row t - (3 - j) d: inside for the block's last row iff (3 - j) d <= t0 + TPC - 1
          const int reach = REVERSE ? (T - t0 + A.dil - 1) / A.dil : (t0 + TPC - 1) / A.dil + 1;
          switch (reach < TK ? reach : TK) {
            case 1: taps(std::integral_constant<int, 1>{}); break;
            case 2: taps(std::integral_constant<int, 2>{}); break;
            case 3: taps(std::integral_constant<int, 3>{}); break;
            default: taps(std::integral_constant<int, 4>{}); break;
          }
          TCN_STAMP(32 * wv + 8 * r + 1);
          // ---- epilogue: lane = sequence sl at row t, output channels ct 16 + kk 4 + q
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const uint32_t off = (uint32_t)s * TC + ct * 16 + kk * 4 + (uint32_t)t * row_stride;
            float xw[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // WGRAD: the convolution's forward input at (t, s), this lane's channels
            if (ok) {
              float v0[4] = {acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]};
              const float pc[4] = {pre[ct].x, pre[ct].y, pre[ct].z, pre[ct].w};
              if (REVERSE && !FUSE_BN && A.accumulate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v0[q] += pc[q];
              }
              if (TAIL) {
                const float sv[4] = {tsv[ct].x, tsv[ct].y, tsv[ct].z, tsv[ct].w};
                if (WGRAD) {
                  xw[0] = xo[ct].x; xw[1] = xo[ct].y; xw[2] = xo[ct].z; xw[3] = xo[ct].w;
#pragma unroll
                  for (int q = 0; q < 4; ++q) v0[q] = xw[q] > 0.0f ? v0[q] + sv[q] : 0.0f;
                } else {
                  const uint32_t nib = tmw >> (ct * 16 + kk * 4);  // bit q: out[t][s][ct*16 + kk*4 + q] > 0
#pragma unroll
                  for (int q = 0; q < 4; ++q) v0[q] = ((nib >> q) & 1u) != 0u ? v0[q] + sv[q] : 0.0f;
                }
                *reinterpret_cast<float4*>(A.tail_gres + off) = make_float4(v0[0], v0[1], v0[2], v0[3]);
                if (own_last && t == T - 1) {   // (requested at the top of the tile: a load here would be waited for on the spot)
                  const float sk4[4] = {skl[ct].x, skl[ct].y, skl[ct].z, skl[ct].w};
#pragma unroll
                  for (int q = 0; q < 4; ++q) v0[q] += sk4[q] > 0.0f ? dfe[ct][q] : 0.0f;
                }
              }
              if (FUSE_BN) {
                const int cw = ct * 4 + kk;
                const float4 m4 = frec[cw], r4 = frec[TC / 4 + cw], sc4 = frec[2 * TC / 4 + cw], sh4 = frec[3 * TC / 4 + cw];
                const float fm[4] = {m4.x, m4.y, m4.z, m4.w}, fr[4] = {r4.x, r4.y, r4.z, r4.w};
                const float fsc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, fsh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float act = fmaf(pc[q], fsc[q], fsh[q]);
                  if (WGRAD && !TAIL) xw[q] = fmaxf(act, 0.0f);  // conv2's input a1 = ReLU(BN1(y1)), the forward's expression
                  v0[q] = act > 0.0f ? v0[q] : 0.0f;
                  s1[ct][q] += v0[q];
                  s2[ct][q] = fmaf(v0[q], (pc[q] - fm[q]) * fr[q], s2[ct][q]);
                }
              }
              *reinterpret_cast<float4*>(A.out + off) = make_float4(v0[0], v0[1], v0[2], v0[3]);
              if (!REVERSE) {
                if (A.stat_records && n_rows == 0.0f) {
#pragma unroll
                  for (int q = 0; q < 4; ++q) kshift[ct][q] = v0[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float dv = v0[q] - kshift[ct][q];
                  s1[ct][q] += dv;
                  s2[ct][q] = fmaf(dv, dv, s2[ct][q]);
                }
              }
            }
            if (WGRAD) {  // every lane writes its four channels (zeros for padded sequences / rows behind the window)
              uint32_t pw[3][2];
              dof_split3x4(xw, pw);
              uint16_t* slot = &ring[((r & 1) * NCW + wv) * 3 * TB_RING];
              const int eo = tb_ring_off(i, ct * 4 + kk);
#pragma unroll
              for (int p = 0; p < 3; ++p) tb_st8(&slot[p * TB_RING + eo], pw[p][0], pw[p][1]);
            }
          }
          if (!REVERSE && ok) n_rows += 1.0f;
        }
        TCN_STAMP(32 * wv + 8 * r + 2);
        if (r + PD < MAXR) prefetch(r + PD);   // (this round's registers are free)
        if (WGRAD) {
          __syncthreads();  // the round's four ring slots are complete: the loader wavefronts take the weight-gradient phase
          TCN_STAMP(32 * wv + 8 * r + 3);
        }
      };
      // Every round's operands are requested here, at the top of the tile: a CU's vector-memory path serves its wavefronts'
      // requests in order, so a request issued behind the loaders' 57 KB of the next tile waits for all of it (measured: 5,000
      // - 10,000 cycles from request to use when a round's operands were requested one round ahead).
#pragma unroll
      for (int r = 0; r < PD; ++r) prefetch(r);
      if (TAIL && own_last) {
        const uint32_t sv = (uint32_t)(ok_s ? s : s0);
        const uint32_t off = sv * TC + kk * 4 + (uint32_t)(T - 1) * row_stride;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          skl[ct] = *reinterpret_cast<const float4*>(A.tail_skip + off + ct * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) dfe[ct][q] = A.tail_dfeat[(int64_t)(ct * 16 + kk * 4 + q) * A.Sp + sv];
        }
      }
      // at most MAXR rounds (13 column blocks), nested so that every request dominates its use
      if (n_rounds > 0) {
        round(0);
        if (n_rounds > 1) {
          round(1);
          if constexpr (MAXR > 2) {
            if (n_rounds > 2) {
              round(2);
              if (n_rounds > 3) round(3);
            }
          }
        }
      }
      TCN_STAMP(32 * wv + 31);
      __syncthreads();  // tile k - 1 is consumed, tile k is staged
      TCN_STAMP(32 * wv + 30);
    }
  }
  // ---- the workgroup's sums (every wavefront is past its last tile barrier: both images are free)
  float* scratch = reinterpret_cast<float*>(img);
  const int i = lane & 15, kk = lane >> 4;
  if (WGRAD) {
    if (loader) {
      // tap (wv - 4)'s 32 x 32 tile.  D register v of lane l = row 8 (v / 4) + 4 (l >> 5) + v % 4, column l & 31; row m of the A
      // operand was output channel 16 (m >> 4) ... in the image's order: m = 16 mh + i <-> channel (i >> 2) 8 + mh 4 + (i & 3);
      // the columns (B operand, the ring's channel order kk 8 + ct 4 + q at 8-byte chunk kk 2 + ct) are channels in natural order
      const int tap = wv - NCW;
      float* outp = A.wg_partials + (tap < 2 ? A.wg_part0 : A.wg_part1) + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int oc = 8 * (v / 4) + 4 * (lane >> 5) + v % 4;
        outp[oc * 65 + (tap & 1) * 32 + (lane & 31)] = accw[v] + accw2[v];
      }
      // bias gradient: channel sums of dy over the staging threads of a channel quad, fixed order
      const int lt = threadIdx.x - 64 * NCW;
#pragma unroll
      for (int c = 0; c < 4; ++c) scratch[(lt >> 3) * 32 + (lt & 7) * 4 + c] = rs[c];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      float b = 0.0f;
      for (int k = 0; k < 32; ++k) b += scratch[k * 32 + threadIdx.x];
      A.wg_partials[A.wg_part0 + (int64_t)blockIdx.x * DOF_OUTER_PARTIAL_FLOATS + threadIdx.x * 65 + 64] = b;
    }
    __syncthreads();
  }
  if (!REVERSE && A.partial && A.stat_records) {
    // one-pass statistics without a reference value (see k_tcn_conv_t): every lane's sums are about ITS first output
    // value, turned into (n, mean, M2) and merged with Chan's update -- the 16 lanes of a row, the four compute wavefronts
    // in order, then the workgroups (k_tcn_stat_merge)
    float* rec = scratch;  // [wavefront][3][32]
    if (!loader) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float n = n_rows, mean = 0.0f, m2 = 0.0f;
          if (n > 0.0f) {
            const float d = s1[ct][q] / n;
            mean = kshift[ct][q] + d;
            m2 = fmaxf(s2[ct][q] - s1[ct][q] * d, 0.0f);
          }
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            const float nb = __shfl_xor(n, m), mb = __shfl_xor(mean, m), qb = __shfl_xor(m2, m);
            dof_stat_merge(n, mean, m2, nb, mb, qb);
          }
          if (i == 0) {
            const int c = ct * 16 + kk * 4 + q;
            rec[(wv * 3 + 0) * 32 + c] = n;
            rec[(wv * 3 + 1) * 32 + c] = mean;
            rec[(wv * 3 + 2) * 32 + c] = m2;
          }
        }
    }
    __syncthreads();
    if (threadIdx.x < TC) {
      const int c = threadIdx.x;
      float n = rec[c], mean = rec[32 + c], m2 = rec[64 + c];
#pragma unroll
      for (int w = 1; w < NCW; ++w) dof_stat_merge(n, mean, m2, rec[(w * 3 + 0) * 32 + c], rec[(w * 3 + 1) * 32 + c], rec[(w * 3 + 2) * 32 + c]);
      float* out = A.partial + (int64_t)blockIdx.x * 3 * TC;
      out[c] = n;
      out[TC + c] = mean;
      out[2 * TC + c] = m2;
    }
  } else if ((!REVERSE || FUSE_BN) && A.partial) {
    float* wsum = scratch;  // [wavefront][64]: (sum 32 | second sum 32)
    if (!loader) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a1 = dof_row16_sum(s1[ct][q]), a2 = dof_row16_sum(s2[ct][q]);
          if (i == 0) {
            wsum[wv * 64 + ct * 16 + kk * 4 + q] = a1;
            wsum[wv * 64 + 32 + ct * 16 + kk * 4 + q] = a2;
          }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * TC) {
      float acc = wsum[threadIdx.x];
#pragma unroll
      for (int w = 1; w < NCW; ++w) acc += wsum[64 * w + threadIdx.x];
      A.partial[(int64_t)blockIdx.x * 2 * TC + threadIdx.x] = acc;
    }
  }
}
